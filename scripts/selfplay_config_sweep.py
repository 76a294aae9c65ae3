#!/usr/bin/env python3
"""self_play_parallel on game counts / rollout counts / schedules the test suite does not name, engine against oracle with the hash
evaluator: identical statistics after every ply, identical examples in identical order.  Prints one line per configuration, exits 1 on
a difference."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tak_amd  # noqa: E402
from oracle import oracle as orc  # noqa: E402

bad = 0
cases = []
for n in (3, 4, 5, 6):
    for games, rollouts in ((1, 33), (3, 5), (7, 17), (65, 9), (130, 5)):
        cases.append((n, games, rollouts, 0, 2, 1, 0))
cases += [(5, 33, 12, 40, 3, 2, 0), (6, 5, 40, 0, 0, 1, 7), (4, 9, 20, 25, 4, 1, 3), (5, 257, 3, 0, 2, 1, 0), (6, 64, 6, 100, 2, 1, 0)]
for (n, games, rollouts, total, komi, noise_plies_div, seed) in cases:
    kw = dict(rollouts=rollouts, noise_plies=6 // noise_plies_div, exploit_plies=4, noise_alpha=0.3, noise_ratio=0.25, komi=komi, total_games=total)
    head_e = tak_amd.HEAD_FC5 if n == 5 else tak_amd.HEAD_CONV
    head_o = orc.HEAD_FC5 if n == 5 else orc.HEAD_CONV
    try:
        e = tak_amd.Engine(n, evaluator=tak_amd.EVAL_HASH, max_batch=max(games, 64), policy_head=head_e)
        e.selfplay_create(games, arena_nodes=1 << 13, seed=seed, max_examples=1 << 15, **kw)
        sp = orc.SelfPlay(n, games, head=head_o, evaluator=orc.EVAL_HASH, seed=seed, **kw)
        ok = True
        plies = 40 if n >= 5 else 60
        for step in range(plies):
            e.selfplay_step(1)
            sp.step(1)
            a, b = e.selfplay_stats(), sp.stats()
            if a != b:
                ok = False
                print("   stats differ at ply", step, a, b)
                break
        if ok:
            g, o = e.selfplay_drain(1 << 15), sp.drain(1 << 15)
            ok = all(np.array_equal(x, y) for x, y in zip(g, o)) and np.array_equal(e.search_states(), sp.states()[0])
        print(("ok  " if ok else "BAD ") + f"n={n} games={games} rollouts={rollouts} total={total} komi={komi} seed={seed}: "
              f"{a['games_finished']} games finished, {a['examples']} examples", flush=True)
        bad += not ok
        e.close()
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print(f"EXC n={n} games={games} rollouts={rollouts}: {ex!r}"[:300], flush=True)
print("bad:", bad)
sys.exit(1 if bad else 0)
