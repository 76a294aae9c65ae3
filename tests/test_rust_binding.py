"""The Rust side of the boundary cannot be compiled in this image (no cargo / rustc), so it is checked mechanically:
rust/takgpu-sys/src/lib.rs against include/takgpu.h — every symbol, struct, field, field order, type, arity, enum value and
constant — and every FFI call / struct literal in rust/takgpu/src against those declarations."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gen_rust_sys as gen  # noqa: E402

LIB_RS = os.path.join(ROOT, "rust", "takgpu-sys", "src", "lib.rs")
CRATE = os.path.join(ROOT, "rust", "takgpu", "src")


def _parse_lib_rs():
    """independent reading of the Rust file: consts, structs (fields in order), extern functions (argument types in order)"""
    text = re.sub(r"//[^\n]*", "", open(LIB_RS).read())
    consts = {m.group(1): (m.group(2), int(m.group(3))) for m in re.finditer(r"pub const (\w+): (\w+) = (-?\d+);", text)}
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", text, flags=re.S):
        fields = re.findall(r"pub (?:r#)?(\w+): ([^,\n]+),", m.group(2))
        structs[m.group(1)] = fields
    ext = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    funcs = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)(?: -> ([^;]+))?;", ext, flags=re.S):
        args = [a.split(":", 1) for a in m.group(2).split(",") if a.strip()]
        funcs[m.group(1)] = ([(a.strip().replace("r#", ""), t.strip()) for a, t in args], (m.group(3) or "()").strip())
    aliases = dict(re.findall(r"pub type (\w+) = ([^;]+);", text))
    return consts, structs, funcs, aliases


def test_lib_rs_is_the_generated_binding_of_the_header():
    assert open(LIB_RS).read() == gen.generate(gen.parse_header()), "rust/takgpu-sys/src/lib.rs is stale: python scripts/gen_rust_sys.py"
    assert open(gen.INTEGRATION).read() == gen.integration_text(gen.parse_header()), \
        "INTEGRATION.md quotes a stale number of entry points: python scripts/gen_rust_sys.py"


def test_every_declaration_matches_the_header():
    h = gen.parse_header()
    consts, structs, funcs, aliases = _parse_lib_rs()
    known = {n for n, _ in h["enums"]} | {n for n, _ in h["structs"]} | {n for n, _ in h["aliases"]} | set(h["opaque"]) | {n for n, _, _ in h["fnptrs"]}
    # constants and enum values
    for name, val in h["defines"]:
        assert consts[name][1] == val, name
    for ename, items in h["enums"]:
        assert aliases[ename] == "c_int"
        for k, v in items:
            assert consts[k] == (ename, v), k
    # structs: same fields, same order, same types, same array lengths
    assert set(structs) == {n for n, _ in h["structs"]} | set(h["opaque"])
    for sname, fields in h["structs"]:
        want = [(f, (f"[{gen.rust_type(t, known)}; {arr}]" if arr else gen.rust_type(t, known))) for f, t, arr in fields]
        assert structs[sname] == want, sname
    # functions: every symbol of the header, same arity, same argument and return types
    assert set(funcs) == {n for n, _, _ in h["functions"]}
    for name, ret, args in h["functions"]:
        rargs, rret = funcs[name]
        assert len(rargs) == len(args), name
        assert [t for _, t in rargs] == [gen.rust_type(ct, known) for ct, _ in args], name
        assert [a for a, _ in rargs] == [an for _, an in args], name
        want_ret = gen.rust_type(ret, known)
        assert rret == ("()" if want_ret == "c_void" else want_ret), name
    # the library exports exactly these (tak_amd.engine.ABI_SYMBOLS is what tests/test_abi.py checks against the .so)
    from tak_amd.engine import ABI_SYMBOLS

    assert set(ABI_SYMBOLS) == set(funcs)


def _call_args(text, start):
    """the argument list of the call whose '(' is at text[start]: top-level comma count → arity"""
    depth, i, commas, empty = 0, start, 0, True
    while True:
        c = text[i]
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return 0 if empty else commas + 1
        elif depth == 1:
            if c == ",":
                commas += 1
            elif not c.isspace():
                empty = False
        i += 1


def test_the_safe_crate_calls_what_the_header_declares():
    h = gen.parse_header()
    arity = {n: len(a) for n, _, a in h["functions"]}
    fields = {n: [f for f, _, _ in fs] for n, fs in h["structs"]}
    consts = {n for n, _ in h["defines"]} | {k for _, items in h["enums"] for k, _ in items}
    used = set()
    for fname in sorted(os.listdir(CRATE)):
        text = re.sub(r"//[^\n]*", "", open(os.path.join(CRATE, fname)).read())
        for m in re.finditer(r"sys::(tg_\w+)\s*\(", text):
            name = m.group(1)
            assert name in arity, f"{fname}: {name} is not declared in takgpu.h"
            n = _call_args(text, m.end() - 1)
            # a trailing comma inside a multi-line call adds no argument
            assert n == arity[name], f"{fname}: {name} called with {n} arguments, the header declares {arity[name]}"
            used.add(name)
        for m in re.finditer(r"sys::(Tg\w+)\s*\{([^{}]*)\}", text):
            name, body = m.group(1), m.group(2)
            assert name in fields, f"{fname}: struct {name} is not declared in takgpu.h"
            got = re.findall(r"(\w+)\s*:(?!:)", body)
            assert ".." not in body and sorted(got) == sorted(fields[name]), f"{fname}: literal of {name} has fields {got}, header {fields[name]}"
        for m in re.finditer(r"sys::(TG_\w+)", text):
            assert m.group(1) in consts, f"{fname}: constant {m.group(1)} is not in takgpu.h"
    # the seams of SURVEY.md §8(b) are all bound by the safe layer
    for need in ("tg_engine_create", "tg_policy_eval", "tg_selfplay_create", "tg_selfplay_step", "tg_selfplay_drain", "tg_train",
                 "tg_train_commit", "tg_pit", "tg_train_comm_init", "tg_train_set_allreduce", "tg_net_set_tensor", "tg_net_finalize"):
        assert need in used, need
