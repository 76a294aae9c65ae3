#!/usr/bin/env python3
"""How much of the training step's wall time has which kernels running: reads a rocprofv3 --kernel-trace CSV of
`scripts/train_step_ab.py N --driver` and prints, for the second half of the trace (steady state), the wall time, the time with an
MFMA-bound kernel (conv / weight gradient / GEMM) running, with two of them running, with only HBM-bound kernels running, idle, and
the average duration per kernel name.   python scripts/probes/train_overlap.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

MFMA = ("k_conv_halo", "k_wgrad_halo", "k_wgrad<", "k_gemm", "k_fc_ring", "k_conv_pos", "k_conv3x3")

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ev.sort()
t_lo = ev[len(ev) // 2][0]
ev = [x for x in ev if x[0] >= t_lo]
t_hi = max(x[1] for x in ev)
pts = []
for s, e, n in ev:
    heavy = any(m in n for m in MFMA)
    pts.append((s, 1, heavy))
    pts.append((e, -1, heavy))
pts.sort()
acc = defaultdict(int)
nh = nl = 0
last = t_lo
for t, d, heavy in pts:
    key = ("2+ MFMA kernels" if nh >= 2 else "1 MFMA kernel" + (" + HBM-bound" if nl else "") if nh == 1 else "HBM-bound only" if nl else "idle")
    acc[key] += t - last
    last = t
    if heavy:
        nh += d
    else:
        nl += d
wall = t_hi - t_lo
print(f"wall {wall / 1e6:.2f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:32s} {v / 1e6:8.2f} ms  {100 * v / wall:5.1f} %")
dur = defaultdict(list)
for s, e, n in ev:
    dur[n.replace("void ", "").replace("tg::", "").split("(")[0][:60]].append(e - s)
print("kernel                                                        calls   avg us   total ms")
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"  {n:60s} {len(d):5d} {sum(d) / len(d) / 1e3:8.1f} {sum(d) / 1e6:9.2f}")
