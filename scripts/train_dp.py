#!/usr/bin/env python3
"""Config C5 rehearsal: self-play + data-parallel training step, one process per GPU.

    python scripts/train_dp.py --gpus 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P scripts/train_dp.py --gpus N

Every rank plays its own shard of games (no collective), trains on its own examples, and every optimiser step
all-reduces the flat gradient buffer over RCCL inside libtakgpu (tg_train_comm_init); torch.distributed only carries
the 128-byte unique id, the barrier and the timing reduction.  Rank 0 prints one JSON line per phase.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--board", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--rollouts", type=int, default=32)
    ap.add_argument("--examples", type=int, default=10000, help="examples per rank")
    ap.add_argument("--chunk", type=int, default=500)
    ap.add_argument("--chunks-in-step", type=int, default=20)
    args = ap.parse_args()

    import torch

    import tak_amd
    import torch_ref
    from tak_amd import dist as tdist

    rank, world, local_rank = tdist.env_rank()
    if not torch.cuda.is_available():
        print("train_dp.py: no GPU visible — the engine has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = tdist.init("nccl", rank, world, device=torch.device("cuda", local_rank))

    head = "fc5" if args.board == 5 else "conv"
    net = torch_ref.make_net(args.board, args.blocks, args.filters, head, seed=0, randomize_bn=False)  # same weights on every rank
    eng = tak_amd.Engine(args.board, res_blocks=args.blocks, filters=args.filters, evaluator=tak_amd.EVAL_RESNET, max_batch=args.games,
                         device=local_rank)
    eng.load_state_dict(torch_ref.abi_tensors(net))
    eng.train_create(chunk_size=args.chunk, chunks_in_step=args.chunks_in_step)
    uid = tdist.broadcast_unique_id(dist, tak_amd.comm_unique_id, device="cuda")
    eng.train_comm_init(rank, world, uid)

    eng.selfplay_create(args.games, arena_nodes=1 << 13, seed=0, rollouts=args.rollouts, max_examples=4 * args.examples,
                        slot_base=tdist.slot_base(rank, args.games))
    got = [np.zeros((0,), tak_amd.engine.EXAMPLE_HEADER), np.zeros((0, eng.sb), np.uint8), np.zeros((0, 512), np.uint16), np.zeros((0, 512), np.uint32)]
    while len(got[0]) < args.examples:
        eng.selfplay_step(4)
        eng.sync()
        got = [np.concatenate([a, b]) for a, b in zip(got, eng.selfplay_drain(args.examples))]
    n_use = (args.examples // args.chunk) * args.chunk  # every rank runs the same number of whole chunks
    hdr, states, moves, visits = [a[:n_use] for a in got]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lp, lz, steps = eng.train(states, hdr["n_moves"], moves, visits, hdr["result"], seed=rank)
    eng.sync()
    dt_local = time.perf_counter() - t0
    dt, positions = tdist.reduce_time_and_count(dist, dt_local, n_use * 8, device="cuda")
    eng.train_commit()
    if rank == 0:
        print(json.dumps({"metric": "training positions/s (forward + backward + Adam, 8-fold augmented)", "value": positions / dt, "n_gpus": world,
                          "seconds": dt, "chunks_per_rank": n_use // args.chunk, "optimizer_steps": steps, "loss_p": lp, "loss_z": lz,
                          "config": {"workload": f"{args.board}x{args.board}, {args.blocks}x{args.filters} net, chunk {args.chunk} examples x 8, "
                                                 f"{args.chunks_in_step} chunks per step, gradient all-reduce over RCCL per step"}}), flush=True)
    # identical parameters on every rank after the all-reduced steps
    w = eng.train_get_tensor("value.weight", (1, args.filters * args.board * args.board))
    if dist is not None:
        t = torch.from_numpy(w).cuda()
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "parameters diverged across ranks"
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
