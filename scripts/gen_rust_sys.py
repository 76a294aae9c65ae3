#!/usr/bin/env python3
"""include/takgpu.h → rust/takgpu-sys/src/lib.rs (raw `extern "C"` binding of every symbol, struct, enum value and constant).

The header is the single source of truth of the C ABI; this script is its only reader that writes Rust.  It also exports
`parse_header()` — the tiny C-declaration parser tests/test_rust_binding.py uses to compare the header with the
committed lib.rs field by field, so a header edit without a regenerated binding fails the CPU test suite.

    python scripts/gen_rust_sys.py            # rewrite rust/takgpu-sys/src/lib.rs and the entry-point count in INTEGRATION.md
    python scripts/gen_rust_sys.py --check    # exit 1 if the committed file differs
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "takgpu.h")
OUT = os.path.join(ROOT, "rust", "takgpu-sys", "src", "lib.rs")

SCALARS = {
    "int": "c_int", "int32_t": "i32", "uint8_t": "u8", "int8_t": "i8", "uint16_t": "u16", "uint32_t": "u32",
    "uint64_t": "u64", "int64_t": "i64", "float": "f32", "double": "f64", "size_t": "usize", "char": "c_char",
    "void": "c_void", "short": "i16",
}


RUST_KEYWORDS = {"move", "fn", "type", "ref", "in", "match", "loop", "box", "mod", "use", "where", "impl", "trait", "self", "super"}


def ident(name):
    """C identifier → Rust identifier (raw form where it collides with a keyword)"""
    return "r#" + name if name in RUST_KEYWORDS else name


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def rust_type(ctype, known):
    """'const float*' → '*const f32'; 'TgEngine**' → '*mut *mut TgEngine'; scalars and known typedef names by value"""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ").split()
    const = "const" in t
    base = [w for w in t if w not in ("const", "struct", "enum")]
    assert len(base) == 1, ctype
    name = base[0]
    r = SCALARS.get(name, name)
    assert name in SCALARS or name in known, f"unknown C type {name!r} in {ctype!r}"
    for i in range(stars):
        r = ("*const " if (const and i == 0) else "*mut ") + r
    return r


def split_decl(decl):
    """'const float* d_buf' → ('const float*', 'd_buf', None); 'uint64_t stack[25]' → ('uint64_t', 'stack', 25)"""
    m = re.match(r"^(.*?)(\w+)\s*(?:\[(\d+)\])?$", decl.strip())
    assert m, decl
    return m.group(1).strip(), m.group(2), int(m.group(3)) if m.group(3) else None


def parse_header(path=HEADER):
    """→ dict(defines=[(name, value)], enums=[(name, [(item, value)])], structs=[(name, [(field, ctype, array)])],
    aliases=[(name, ctype)], opaque=[name], fnptrs=[(name, ret, [(ctype, argname)])], functions=[(name, ret, [(ctype, argname)])])"""
    text = strip_comments(open(path).read())
    text = re.sub(r"\bTG_API\s+", "", text)  # the export attribute of the 60-odd entry points: not part of a signature
    out = dict(defines=[], enums=[], structs=[], aliases=[], opaque=[], fnptrs=[], functions=[])
    for m in re.finditer(r"^#define\s+(TG_[A-Z0-9_]+)\s+(.+)$", text, flags=re.M):
        name, val = m.group(1), m.group(2).strip()
        if "(" in name or name.startswith("TG_META"):
            continue
        mm = re.fullmatch(r"\(?\s*(-?\d+)\s*(?:<<\s*(\d+))?\s*\)?", val)
        if mm:
            out["defines"].append((name, int(mm.group(1)) << int(mm.group(2) or 0)))
    for m in re.finditer(r"typedef\s+enum\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        items, nxt = [], 0
        for part in m.group(2).split(","):
            part = part.strip()
            if not part:
                continue
            if "=" in part:
                k, v = [x.strip() for x in part.split("=")]
                nxt = int(v, 0)
            else:
                k = part
            items.append((k, nxt))
            nxt += 1
        out["enums"].append((m.group(1), items))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            first, *rest = [p.strip() for p in stmt.split(",")]
            ctype, fname, arr = split_decl(first)
            fields.append((fname, ctype, arr))
            for r in rest:  # `uint8_t a, b, c;`
                _, fname2, arr2 = split_decl(ctype + " " + r)
                fields.append((fname2, ctype, arr2))
        out["structs"].append((m.group(1), fields))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+\1\s*;", text):
        out["opaque"].append(m.group(1))
    for m in re.finditer(r"typedef\s+(\w+)\s+(\w+)\s*;", text):
        if m.group(1) not in ("struct", "enum"):
            out["aliases"].append((m.group(2), m.group(1)))
    for m in re.finditer(r"typedef\s+([\w\s\*]+?)\(\s*\*\s*(\w+)\s*\)\s*\((.*?)\)\s*;", text, flags=re.S):
        args = [split_decl(a)[:2] for a in " ".join(m.group(3).split()).split(",")]
        out["fnptrs"].append((m.group(2), m.group(1).strip(), args))
    body = re.sub(r"typedef\s+(enum|struct)\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
    body = re.sub(r"typedef[^;]*;", "", body)
    for m in re.finditer(r"^\s*([\w\s\*]+?)\b(tg_\w+)\s*\(([^;{}]*?)\)\s*;", body, flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        alist = [] if args in ("", "void") else [split_decl(a)[:2] for a in args.split(",")]
        out["functions"].append((name, ret, alist))
    return out


def generate(h):
    known = {n for n, _ in h["enums"]} | {n for n, _ in h["structs"]} | {n for n, _ in h["aliases"]} | set(h["opaque"]) | \
        {n for n, _, _ in h["fnptrs"]}
    L = []
    w = L.append
    w("//! Raw binding of `include/takgpu.h` (libtakgpu.so, the MI355X batched Tak self-play engine).")
    w("//!")
    w("//! GENERATED by `scripts/gen_rust_sys.py` from the header — do not edit; `tests/test_rust_binding.py` fails when this")
    w("//! file and the header disagree in any symbol, struct field, field order, type, arity or constant.")
    w("//! NOT COMPILED IN THE BUILD IMAGE of this repository (no cargo / rustc there): written against the header only.")
    w("#![allow(non_camel_case_types, non_upper_case_globals, clippy::too_many_arguments)]")
    w("")
    w("use std::os::raw::{c_char, c_int, c_void};")
    w("")
    for name, val in h["defines"]:
        w(f"pub const {name}: i32 = {val};")
    w("")
    for name, ctype in h["aliases"]:
        w(f"pub type {name} = {rust_type(ctype, known)};")
    for name in h["opaque"]:
        w("#[repr(C)]")
        w(f"pub struct {name} {{")
        w("    _private: [u8; 0],")
        w("}")
    w("")
    for name, items in h["enums"]:
        w(f"/// C enum `{name}`: passed and returned as `c_int`")
        w(f"pub type {name} = c_int;")
        for k, v in items:
            w(f"pub const {k}: {name} = {v};")
        w("")
    for name, fields in h["structs"]:
        w("#[repr(C)]")
        w("#[derive(Clone, Copy, Debug)]")
        w(f"pub struct {name} {{")
        for fname, ctype, arr in fields:
            rt = rust_type(ctype, known)
            w(f"    pub {ident(fname)}: {f'[{rt}; {arr}]' if arr else rt},")
        w("}")
        w("")
    for name, ret, args in h["fnptrs"]:
        a = ", ".join(f"{ident(an)}: {rust_type(ct, known)}" for ct, an in args)
        r = rust_type(ret, known)
        w(f"pub type {name} = Option<unsafe extern \"C\" fn({a}){'' if r == 'c_void' else ' -> ' + r}>;")
    w("")
    w('#[link(name = "takgpu")]')
    w('extern "C" {')
    for name, ret, args in h["functions"]:
        a = ", ".join(f"{ident(an)}: {rust_type(ct, known)}" for ct, an in args)
        r = rust_type(ret, known)
        w(f"    pub fn {name}({a}){'' if r == 'c_void' else ' -> ' + r};")
    w("}")
    return "\n".join(L) + "\n"


INTEGRATION = os.path.join(ROOT, "INTEGRATION.md")
COUNT_RE = re.compile(r"(<!-- abi-entry-points -->)\d+(<!-- /abi-entry-points -->)")


def integration_text(h, path=INTEGRATION):
    """INTEGRATION.md with the number of entry points taken from the header (the figure between the abi-entry-points markers)"""
    text = open(path).read()
    assert COUNT_RE.search(text), "INTEGRATION.md has lost its <!-- abi-entry-points --> markers"
    return COUNT_RE.sub(lambda m: f"{m.group(1)}{len(h['functions'])}{m.group(2)}", text)


def main():
    h = parse_header()
    text = generate(h)
    doc = integration_text(h)
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text and open(INTEGRATION).read() == doc else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    open(INTEGRATION, "w").write(doc)
    print(f"wrote {OUT} ({text.count(chr(10))} lines); INTEGRATION.md: {len(h['functions'])} entry points")


if __name__ == "__main__":
    main()
