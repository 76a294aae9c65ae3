#!/usr/bin/env python3
"""Extract the known-answer vectors held by the reference's own tests into tests/golden/reference_kats.json.

Reads (as text) /root/reference/{tak/tests/{perft,wins,tps,symm}.rs, alpha-tak/src/repr/tests.rs,
alpha-tak/src/search/move_map.rs}.  Only DATA is kept: PTN move lists, expected counts / results /
strings / planes, seeds, and the sha256 of the legacy 5×5 move table (not the table itself).
Run in the build container (the reference is not present on the GPU box); the JSON is committed.
"""
import hashlib
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def strings(block):
    return re.findall(r'"([^"]*)"', block)


def split_tests(src):
    """yield (name, body) for every `fn name(...) ... {` at top level following #[test]"""
    out = []
    for m in re.finditer(r"#\[test\]\s*fn\s+(\w+)\s*\([^)]*\)[^{]*\{", src):
        start = m.end()
        depth, i = 1, start
        while depth and i < len(src):
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        out.append((m.group(1), src[start:i - 1]))
    return out


def num(s):
    return int(s.replace("_", ""))


kats = {"source": "ViliamVadocz/tak test vectors (data only)", "perft": [], "wins": [], "seeds": []}

# ---- tak/tests/perft.rs ---------------------------------------------------------------------
for name, body in split_tests(read("tak/tests/perft.rs")):
    live = "\n".join(l for l in body.split("\n") if not l.strip().startswith("//"))
    m = re.search(r"Game::<(\d)>::from_ptn_moves\(&\[(.*?)\]\)", live, re.S)
    if m:
        n, moves = int(m.group(1)), strings(m.group(2))
        for d, v in re.findall(r"perf_count\(&game,\s*(\d+)\),\s*([\d_]+)\)", live):
            kats["perft"].append({"name": name, "n": n, "moves": moves, "depth": int(d), "count": num(v)})
    else:
        for n, d, v in re.findall(r"perf_count\(&Game::<(\d)>::default\(\),\s*(\d+)\),\s*([\d_]+)\)", live):
            kats["perft"].append({"name": name, "n": int(n), "moves": [], "depth": int(d), "count": num(v)})

# ---- tak/tests/wins.rs ----------------------------------------------------------------------
RES = {("White", "true"): "WhiteRoad", ("White", "false"): "WhiteFlat", ("Black", "true"): "BlackRoad", ("Black", "false"): "BlackFlat"}
for name, body in split_tests(read("tak/tests/wins.rs")):
    m = re.search(r"Game::<(\d)>::from_ptn_moves\(&\[(.*?)\]\)", body, re.S)
    n, moves = int(m.group(1)), strings(m.group(2))
    half_komi = 0
    # walk statements in order: `game.half_komi = k;` then asserts
    for stmt in re.finditer(r"game\.half_komi\s*=\s*(-?\d+)|assert_eq!\(game\.result\(\),\s*GameResult::(\w+)\s*\{(.*?)\}\)", body, re.S):
        if stmt.group(1) is not None:
            half_komi = int(stmt.group(1))
            continue
        kind, inner = stmt.group(2), stmt.group(3)
        if kind == "Winner":
            color = re.search(r"Color::(\w+)", inner).group(1)
            road = re.search(r"road:\s*(\w+)", inner).group(1)
            res = RES[(color, road)]
        else:
            res = "DrawReversible" if "true" in inner else "Draw"
        kats["wins"].append({"name": name, "n": n, "moves": moves, "half_komi": half_komi, "result": res})

# ---- tak/tests/tps.rs -----------------------------------------------------------------------
src = read("tak/tests/tps.rs")
for name, body in split_tests(src):
    if name == "complicated_board":
        m = re.search(r"Game::<(\d)>::from_ptn_moves\(&\[(.*?)\]\)", body, re.S)
        s = re.search(r'tps\.to_string\(\),\s*"(.*?)"\s*\)', body, re.S).group(1)
        s = re.sub(r"\\\n\s*", "", s)  # rust line continuation
        kats["tps"] = {"n": int(m.group(1)), "moves": strings(m.group(2)), "tps": s}
kats["seeds"] = sorted({int(x) for x in re.findall(r"tps_consistency\((\d+)\)", src)} |
                       {int(x) for x in re.findall(r"symmetrical_boards\((\d+)\)", read("tak/tests/symm.rs"))})

# ---- alpha-tak/src/repr/tests.rs ------------------------------------------------------------
for name, body in split_tests(read("alpha-tak/src/repr/tests.rs")):
    if name == "complicated_board":
        m = re.search(r"Game::<(\d)>::from_ptn_moves\(&\[(.*?)\]\)", body, re.S)
        block = re.search(r"Tensor::of_slice\(&\[(.*?)\]\)\.view\(\[(\d+),\s*5,\s*5\]\)", body, re.S)
        cells = re.findall(r"\b([xo])\b", re.sub(r"//.*", "", block.group(1)))
        planes = int(block.group(2))
        assert len(cells) == planes * 25, len(cells)
        kats["repr"] = {
            "n": int(m.group(1)), "moves": strings(m.group(2)), "to_move_perspective": "White",
            "planes": planes, "bits": "".join("1" if c == "x" else "0" for c in cells),
            "note": "board_repr(&board, Color::White): first `planes` channels as row-major 5x5 bit strings; remaining board channels are zero",
        }

# ---- alpha-tak/src/search/move_map.rs -------------------------------------------------------
mm = read("alpha-tak/src/search/move_map.rs")
table = strings(mm[mm.index("const POSSIBLE_MOVES_IN_5S"):])
assert len(table) == 1575, len(table)
kats["legacy5"] = {
    "count": len(table),
    "sha256_newline_joined": hashlib.sha256("\n".join(table).encode()).hexdigest(),
    "spot": {str(i): table[i] for i in (0, 1, 2, 74, 75, 76, 77, 105, 500, 1000, 1574)},
}

with open(OUT, "w") as f:
    json.dump(kats, f, indent=1)
print("wrote", os.path.normpath(OUT), {k: (len(v) if isinstance(v, (list, dict)) else v) for k, v in kats.items()})
