#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native Tak self-play engine.

Metric (BASELINE.json): MCTS node-expansions / second, 5×5 Tak, 400 sims/move, 4096 concurrent games
per GPU, 6-block × 64-filter resnet with the FC-1575 policy head (config C2), random-init weights,
synthetic self-play (no dataset exists for this path).  One "step" = one ply of self_play_parallel for
every game on the GPU: opening / instant-win scan / root evaluation + Dirichlet noise / 400 lock-step
rollouts (virtual_rollout → one batched network forward → devirtualize_path) / move choice, example
emission, tree reuse and game recycling.  Everything runs on the GPU with states resident in HBM.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts the N ranks itself, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W
    python bench.py [--gpus N] --train                     (config C5 alone: the data-parallel training step)

Self-play shards by game with no data-path collective: rank r owns games
[r·4096, (r+1)·4096) with their own RNG streams; RCCL is used only for the barrier / max-over-ranks
timing.  Config C5 (`extra.train_c5`, every N): the 10-block x 128-filter network plays, then trains on its own
examples with the reference's chunking (500 examples x 8 symmetries, 20 chunks per optimiser step,
alpha-tak/src/model/network.rs:89-96); with N > 1 every step all-reduces the 32.3 MB flat gradient buffer over RCCL
inside libtakgpu — the only collective of the build.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact f32


def make_weights(n, blocks, filters, head, seed=0):
    """Random-init weights of the named topology (PyTorch default init, fresh BatchNorm), tch layout."""
    import torch_ref

    net = torch_ref.make_net(n, blocks, filters, head, seed=seed, randomize_bn=False)
    return net, torch_ref.abi_tensors(net)


def cpu_baseline(args, net, seconds_budget=22.0, dummy_budget=8.0):
    """The CPU port (oracle/: scalar MCTS + AoS rules, PyTorch-CPU fp32 network) on the host cores, on a bounded sample of
    the same workload: `cpu_games` of the 4096 games, as many whole plies (1 + rollouts lock-step iterations each) as fit
    in ≈ `seconds_budget`.  `value` excludes the first ply (cold caches, empty trees).  A second, shorter sample with the
    reference's DummyNet (alpha-tak/src/search/tests.rs:29-34: uniform policy, eval 0) times board + MCTS alone."""
    import torch

    import torch_ref
    from oracle import oracle as orc

    n = args.board
    head = orc.HEAD_FC5 if args.head == "fc5" else orc.HEAD_CONV
    # the GPU box gives one GPU's worth of host cores (16); PyTorch's default of one thread per visible core
    # (128+) oversubscribes them on these small convolutions
    torch.set_num_threads(min(args.cpu_threads, os.cpu_count() or 1))
    threads = int(torch.get_num_threads())

    def py_eval(states):
        return torch_ref.forward(net, orc.encode(n, states))

    def sample(sp, budget, max_plies):
        sp.set_threads(threads)  # the per-game MCTS phases under OpenMP, the network under PyTorch's pool
        marks = [(0.0, 0)]
        t0 = time.perf_counter()
        while True:
            sp.step(1)
            marks.append((time.perf_counter() - t0, sp.stats()["expansions"]))
            per_ply = marks[-1][0] / (len(marks) - 1)
            if marks[-1][0] + per_ply > budget or len(marks) - 1 >= max_plies:
                return marks

    marks = sample(orc.SelfPlay(n, args.cpu_games, head=head, py_eval=py_eval, seed=args.seed, rollouts=args.rollouts), seconds_budget, 64)
    (t1, e1), (tn, en) = marks[1], marks[-1]
    steady = (en - e1) / (tn - t1) if len(marks) > 2 else en / tn
    dummy = sample(orc.SelfPlay(n, args.games, head=head, evaluator=orc.EVAL_DUMMY, seed=args.seed, rollouts=args.rollouts), dummy_budget, 16)
    return {
        "value": steady,
        "unit": "node-expansions/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{args.cpu_games} of the {args.games} games, {len(marks) - 1} plies = {en} expansions in {tn:.1f} s ({en / tn:.0f}/s with the cold "
                  f"first ply, `value` without it); oracle scalar MCTS (OpenMP over games, {threads} threads) + PyTorch-CPU fp32 "
                  f"{args.blocks}x{args.filters} net ({threads} threads)",
        "board_and_mcts_only": {
            "value": dummy[-1][1] / dummy[-1][0], "unit": "node-expansions/s", "cores": threads,
            "sample": f"all {args.games} games, {len(dummy) - 1} ply(ies) = {dummy[-1][1]} expansions in {dummy[-1][0]:.1f} s with the reference's DummyNet "
                      f"(uniform policy, eval 0): rules + tree work alone, no network",
        },
    }


def measured_deviation(args, tensors, device, count=4096, max_ply=150):
    """max deviation of the split-bf16 forward from the exact-f32 forward on `count` positions reached by random play through the
    engine's own rules kernels (no checker involved) — the figure `alt_precision` quotes"""
    import tak_amd

    head = tak_amd.HEAD_FC5 if args.head == "fc5" else tak_amd.HEAD_CONV
    out = []
    st = None
    for precision in ("f32", "bf16x3"):
        e = tak_amd.Engine(args.board, res_blocks=args.blocks, filters=args.filters, policy_head=head, evaluator=tak_amd.EVAL_RESNET,
                           max_batch=count, device=device)
        e.set_precision(precision)
        e.load_state_dict(tensors)
        if st is None:
            st = np.zeros((count, e.sb), np.uint8)
            hdr = e.sb - 16
            stones = {3: 10, 4: 15, 5: 21, 6: 30}[args.board]
            caps = 1 if args.board >= 5 else 0
            st[:, hdr + 0] = args.board
            st[:, hdr + 4], st[:, hdr + 5], st[:, hdr + 6], st[:, hdr + 7] = stones, caps, stones, caps
            st[:, hdr + 8] = 4
            rng = np.random.default_rng(args.seed)
            target = np.arange(count) % (max_ply + 1)  # position i stops at ply ≈ i mod 151 (earlier if its game ends)
            for ply in range(max_ply):
                moves, counts = e.movegen(st)
                pick = (rng.random(count) * np.maximum(counts, 1)).astype(np.int64)
                nxt, status = e.play(st, moves[np.arange(count), pick])
                res = e.result(nxt)
                go = (target > ply) & (status == 0) & (res == 0) & (counts > 0)
                st[go] = nxt[go]
            plies = st[:, hdr + 2].astype(np.int32) | (st[:, hdr + 3].astype(np.int32) << 8)
        out.append(e.policy_eval(st))
        e.close()
    (p0, v0), (p1, v1) = out
    return {"policy_rel": float((np.abs(p1 - p0) / p0).max()), "policy_abs": float(np.abs(p1 - p0).max()),
            "eval_abs": float(np.abs(v1 - v0).max()), "gate": 1e-4, "positions": count,
            "plies": {"min": int(plies.min()), "median": int(np.median(plies)), "max": int(plies.max())}}


def tak_amd_supports_bf16x3(args):
    return (args.board == 5 and args.filters in (64, 128)) or (args.board == 6 and args.filters == 128)


def switches_set():
    """the TG_* A/B switches that are ON in this process's environment, as libtakgpu itself reads them (tg_debug_switches; no GPU needed)"""
    import tak_amd

    return tak_amd.debug_switches()


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (one per GPU), exactly as
    `torch.distributed.run --nproc-per-node N` would.  This parent never initialises HIP — it imports neither torch nor the
    engine — so nothing that has touched the GPU forks or execs.  Rank 0's JSON line is relayed to stdout, every
    other stream goes to stderr; the exit code is non-zero if any rank fails, and the first failure ends the others (by
    their exact PIDs)."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))

    def relay(pipe):  # the JSON line to stdout; whatever libraries print on rank 0's stdout (gloo's connection notes) to stderr
        for line in iter(pipe.readline, b""):
            dst = sys.stdout.buffer if line.lstrip().startswith(b"{") else sys.stderr.buffer
            dst.write(line)
            dst.flush()

    t = threading.Thread(target=relay, args=(procs[0].stdout,), daemon=True)
    t.start()
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()
        time.sleep(0.05)
    t.join(timeout=5)
    return rc


class Watchdog:
    """Bounds a phase that contains collectives which have never run at this world size: if it does not finish in time,
    `on_timeout` runs (rank 0 prints the line it already has) and the process leaves without waiting for the GPU."""

    def __init__(self, seconds, on_timeout, exit_code=3):
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.on_timeout = on_timeout
        self.exit_code = exit_code

    def _fire(self):
        try:
            self.on_timeout()
        finally:
            os._exit(self.exit_code)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


_LINE = threading.Lock()  # taken by whoever prints this process's ONE JSON line; never released


def print_line(obj):
    """print the JSON line unless this process has already printed one (the C5 phase's watcher thread or watchdog may have) → printed?"""
    if not _LINE.acquire(blocking=False):
        return False
    print(json.dumps(obj), flush=True)
    return True


def run_c5_phase(args, rank, world, out, phase, want_result=False, dist=None):
    """Config C5's phase — the only part of the bench with a data-path collective — under the rules that keep a bad day
    visible: an exception is caught (rank 0's headline line survives, marked `train_c5_failed`), and at N > 1
      * a watchdog bounds the phase: if a collective never completes, rank 0 prints the line it has (marked) and EVERY rank leaves
        with EXIT_C5_FAILED straight from the timer thread (os._exit: a fresh exit, nothing is exec'ed);
      * a rank != 0 that raises inside the phase — also where no agreement protects it: inside tg_train, between two reductions —
        posts the failure on the rendezvous store and waits for rank 0's acknowledgement before it leaves (tak_amd.dist.FailureBoard);
        rank 0 polls the store from a thread while its main thread may sit in the all-reduce the failed rank never joins, prints its
        line (marked, naming the rank) and leaves: the launcher ends all ranks at the first non-zero exit, and the headline is out
        by then.  → (out, failed[, result])"""
    from tak_amd import dist as tdist

    progress = {"stage": "not started"}  # the phase reports where it is (train_c5's `note`): a timeout names the stage it hung in

    def print_marked(error):
        if rank != 0:
            return
        where = {"error": error, "stage": progress["stage"], **{k: v for k, v in progress.items() if k != "stage"}}
        if out is not None:
            marked = dict(out, train_c5_failed=True)
            marked["extra"] = dict(out.get("extra", {}), train_c5=where)
            print_line(marked)
        elif args.train:  # C5 alone: there is no headline yet — say so in the one line
            print_line(failed_train_line(world, error, where))

    def give_up():
        print(f"bench.py: rank {rank}: the C5 phase did not finish within {args.train_timeout:.0f} s", file=sys.stderr, flush=True)
        print_marked(f"timed out after {args.train_timeout:.0f} s (world {world}) in stage '{progress['stage']}'")

    board = tdist.FailureBoard(dist) if world > 1 else None
    stop = threading.Event()

    def watch_peers():  # rank 0 only
        while not stop.wait(0.25):
            msg = board.posted()
            if msg:
                print(f"bench.py: rank 0: {msg} — leaving the C5 phase", file=sys.stderr, flush=True)
                print_marked(f"another rank failed inside the phase ({msg})")
                board.acknowledge()
                sys.stdout.flush()
                os._exit(EXIT_C5_FAILED)

    failed = False
    watcher = None
    try:
        if world > 1:
            if rank == 0 and board.store is not None:
                watcher = threading.Thread(target=watch_peers, daemon=True)
                watcher.start()
            # (rank 0 first, so that its line is out before the launcher sees another rank's exit code and ends the rest)
            with Watchdog(args.train_timeout + (0.0 if rank == 0 else 15.0), give_up, EXIT_C5_FAILED):
                c5 = phase(progress)
        else:
            c5 = phase(progress)
    except Exception as ex:
        if args.train and world == 1:
            raise
        c5, failed = {"error": repr(ex)}, True
        print(f"bench.py: rank {rank}: config C5 failed: {ex!r}", file=sys.stderr, flush=True)
        if world > 1 and rank != 0:
            board.post(rank, repr(ex))
            board.wait_acknowledged(20.0)  # rank 0 prints first (or is past the phase already and prints in finish())
    finally:
        stop.set()
        # the watcher is either idle (it leaves at its next wait) or about to print the marked line and end the process itself:
        # the main thread does not go on to print a second line beside it
        if watcher is not None:
            watcher.join(timeout=30.0)
    if rank == 0 and out is not None and not args.train:
        out.setdefault("extra", {})["train_c5"] = c5
        if failed:
            out["train_c5_failed"] = True
    return (out, failed, c5) if want_result else (out, failed)


def failed_train_line(world, error, where=None):
    """the JSON line of `bench.py --train` when the phase failed: no value, marked"""
    return {"metric": "training positions/s (forward + backward + Adam; 8-fold augmented examples)", "value": None, "per_gpu_value": None,
            "unit": "positions/s", "n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "train_c5_failed": True, "train_c5": where or {"error": error}}


def finish(rank, world, out, c5_failed, dist):
    """rank 0 prints THE line; then a clean shutdown — or, if config C5 failed at N > 1, a non-zero exit on every rank (the
    headline is out, but a data-parallel run whose collective phase failed must not look green)"""
    if rank == 0:
        print_line(out)  # (skipped if the C5 phase's watcher or watchdog has already printed the marked line)
    if c5_failed and world > 1:
        sys.stdout.flush()
        if rank != 0:
            time.sleep(3.0)  # rank 0's line first: the launcher ends the other ranks at the first non-zero exit
        os._exit(EXIT_C5_FAILED)  # (no destroy_process_group: the other ranks may be gone)
    if dist is not None:
        dist.destroy_process_group()


EXIT_C5_FAILED = 4  # N > 1: the C5 phase (the only collective data path) hung or raised; the headline line was still printed
C5_FLOPS_FWD = 161_689_600  # 5x5, 10 blocks x 128 filters, FC-1575 head, 2·MAC per position (forward)


def train_c5(args, rank, world, local_rank, dist, backend, barrier_fn, progress=None):
    """BASELINE config C5 on this rank's GPU: the 10-block x 128-filter network plays 5x5 self-play (games sharded by rank, no
    collective), then trains data-parallel on the examples it just produced: tg_train = Network::train (shuffle, chunks of 500
    examples x 8 symmetries, an Adam step every 20 chunks).  With world > 1 every optimiser step all-reduces the flat
    gradient buffer (8.08 M f32 = 32.3 MB) over RCCL inside libtakgpu (tg_train_comm_init; the 128-byte unique id travels
    over torch.distributed); under TAK_BENCH_BACKEND=gloo (several ranks on one card) the same reduction goes through the
    host hook.  Returns the dict rank 0 reports (identical timing discipline: barrier, sync, max over ranks)."""
    import torch

    import tak_amd
    from tak_amd import dist as tdist

    n, blocks, filters = 5, args.train_blocks, args.train_filters
    games = args.train_games
    dev = "cuda" if backend == "nccl" else "cpu"

    progress = {} if progress is None else progress
    stages = tdist.Stages(dist, rank, dev)

    def stage(name, fn):  # rank-local work, then all ranks agree that it succeeded everywhere; the phase's watchdog names the stage it hung in
        progress["stage"] = name
        return stages.run(name, fn)

    def setup():
        net, weights = make_weights(n, blocks, filters, "fc5", seed=args.seed)  # the same weights on every rank
        e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, policy_head=tak_amd.HEAD_FC5, evaluator=tak_amd.EVAL_RESNET,
                           max_batch=games, device=local_rank)
        e.load_state_dict(weights)
        e.train_create(chunk_size=args.train_chunk, chunks_in_step=args.train_chunks_in_step)
        return e

    eng = stage("engine setup", setup)
    transport = "single rank"
    if world > 1:
        # (collective from here on: the watchdog's line names this stage if the communicator never forms)
        progress["stage"] = "communicator set-up (ncclGetUniqueId broadcast, ncclCommInitRank inside libtakgpu)" if backend == "nccl" else "host all-reduce hook"
        if backend == "nccl":
            uid = tdist.broadcast_unique_id(dist, tak_amd.comm_unique_id, device="cuda")
            eng.train_comm_init(rank, world, uid)
            transport = "RCCL ncclAllReduce(sum, f32) inside libtakgpu"
        else:
            eng.train_set_allreduce(tdist.host_allreduce_hook(dist), world)
            transport = f"host all-reduce through torch.distributed/{backend} (ranks share a card)"

    def communicator():
        info = eng.train_comm_info()
        # what RCCL itself says, checked on every rank: a communicator that saw fewer ranks than the launcher started would
        # still "work" (and train on a fraction of the examples)
        if world > 1 and backend == "nccl" and (info["nccl_count"] != world or info["nccl_rank"] != rank):
            raise RuntimeError(f"rank {rank}: RCCL communicator reports {info['nccl_count']} ranks / rank {info['nccl_rank']}, expected {world} / {rank}")
        return info

    rccl = stage("communicator check", communicator)

    # One float through the reduction the optimiser step will use — RCCL's first collective on this communicator (ring / tree set-up over
    # xGMI) or the host hook — BEFORE anything of a training step is enqueued: "RCCL could not form a ring on this box" ends here, in a
    # stage of its own, and reads differently from "our step hung".  Not a Stages.run: the call itself is the collective.
    progress["stage"] = "reduction preflight (tg_train_comm_preflight: 4 bytes through " + transport + ")"
    preflight_ms = eng.train_comm_preflight()
    progress["preflight_ms_rank0"] = preflight_ms
    preflight = tdist.per_rank_times(dist, preflight_ms, scale=1.0)

    # 1. self-play on the C5 network at the headline search settings (2 plies timed after 1 warm-up ply)
    def warm_up():
        eng.selfplay_create(games, arena_nodes=args.arena, seed=args.seed, rollouts=args.rollouts, max_examples=1 << 14,
                            slot_base=tdist.slot_base(rank, games))
        eng.selfplay_step(1)
        eng.sync()
        return eng.selfplay_stats()

    def two_plies():
        t0 = time.perf_counter()
        eng.selfplay_step(2)
        eng.sync()
        return time.perf_counter() - t0, eng.selfplay_stats()["expansions"]

    s0 = stage("self-play warm-up", warm_up)
    dt_sp_local, exp1 = stage("timed self-play", two_plies)
    dt_sp, sp_total = tdist.reduce_time_and_count(dist, dt_sp_local, exp1 - s0["expansions"], device=dev)

    # 2. examples for the training step: the same network at a few rollouts per move until every rank holds enough
    need = args.train_steps * args.train_chunk * args.train_chunks_in_step

    def examples():
        eng.selfplay_create(games, arena_nodes=1 << 12, seed=args.seed + 1, rollouts=args.train_example_rollouts,
                            max_examples=max(1 << 16, 4 * need), slot_base=tdist.slot_base(rank, games))
        got = None
        t_gen = time.perf_counter()
        while got is None or len(got[0]) < need:
            eng.selfplay_step(8)
            new = eng.selfplay_drain(2 * need)
            got = new if got is None else [np.concatenate([a, b]) for a, b in zip(got, new)]
            if time.perf_counter() - t_gen > 120:
                raise RuntimeError(f"example generation: {len(got[0])} of {need} examples after 120 s")
        return time.perf_counter() - t_gen, [a[:need] for a in got]

    t_gen, (hdr, states, moves, visits) = stage("example generation", examples)

    # 3. the training step(s), timed
    progress["stage"] = "tg_train (data-parallel optimiser steps; the gradient all-reduce of every step)"
    barrier_fn(eng)
    t0 = time.perf_counter()
    lp, lz, steps = eng.train(states, hdr["n_moves"], moves, visits, hdr["result"], seed=args.seed + rank)
    eng.sync()
    dt_local = time.perf_counter() - t0
    barrier_fn(eng)
    progress["stage"] = "after tg_train (reductions of the timings, parameter comparison, tg_train_commit)"
    dt, positions = tdist.reduce_time_and_count(dist, dt_local, need * 8, device=dev)
    step_ms = tdist.per_rank_times(dist, dt_local / max(steps, 1))
    sp_ms = tdist.per_rank_times(dist, dt_sp_local / 2)
    reporting = tdist.ranks_reporting(dist, device=dev)
    ar_ms, ar_n = eng.train_comm_stats()
    # identical parameters on every rank after the all-reduced steps
    w = eng.train_get_tensor("value.weight", (1, filters * n * n))
    same = True
    if dist is not None:
        t = torch.from_numpy(w).to(dev)
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same = bool(torch.equal(lo, hi))
    eng.train_commit()
    eng.close()
    flops = 3 * C5_FLOPS_FWD if (blocks, filters) == (10, 128) else None  # forward + data gradient + weight gradient
    per_gpu = positions / dt / world
    return {
        "metric": "training positions/s (forward + backward + Adam; 8-fold augmented examples)", "value": positions / dt, "unit": "positions/s",
        "per_gpu_value": per_gpu, "n_gpus": world, "ranks_reporting": reporting, "seconds": dt, "optimizer_steps": steps, "ms_per_optimizer_step": 1000.0 * dt / max(steps, 1),
        "ms_per_optimizer_step_by_rank": step_ms,  # every rank's own time; `value` uses the slowest
        "positions_per_rank": need * 8, "loss_p": lp, "loss_z": lz,
        "frac_of_f32_mfma_peak": (per_gpu * flops / 1e12 / F32_MFMA_PEAK_TFLOPS) if flops else None,
        "gradient_allreduce": {"transport": transport, "bytes": None if world == 1 else int(eng_param_bytes(blocks, filters, n)),
                               "count": ar_n, "ms_per_step_rank0": (ar_ms / ar_n) if ar_n else 0.0,
                               # the 4-byte reduction before the first chunk, wall clock per rank (RCCL: includes the communicator's
                               # first-collective set-up); 0.0 on a single rank
                               "preflight_ms": preflight["max"], "preflight_ms_by_rank": preflight,
                               # rank 0's view of the communicator (every rank checked its own above): ncclCommCount,
                               # ncclCommUserRank, ncclGetVersion, and the file ncclAllReduce was bound from
                               "rccl": rccl},
        "parameters_identical_on_all_ranks": same,
        "selfplay_c5net": {"value": sp_total / dt_sp, "unit": "node-expansions/s", "ms_per_step": 1000.0 * dt_sp / 2, "ms_per_step_by_rank": sp_ms,
                           "games_per_gpu": games, "sims_per_move": args.rollouts},
        "example_generation_s": t_gen,
        "workload": f"BASELINE config C5: 5x5 Tak, {blocks}-block x {filters}-filter resnet (fc5 head): self-play, then Network::train on its own examples — "
                    f"{need} examples/rank, chunks of {args.train_chunk} x 8 symmetries, an Adam step every {args.train_chunks_in_step} chunks, "
                    f"gradients all-reduced once per step ({transport}); host buffers in, so the chunk uploads (1.6 MB each) are inside the time",
    }


def eng_param_bytes(blocks, filters, n):
    """bytes of the flat gradient buffer (trainable parameters, f32) of a 5x5 fc5-head network"""
    cin = (n + 2 + 6) * 2 + 2 + 2 * {3: 10, 4: 15, 5: 21, 6: 30}[n] + 2 * (1 if n >= 5 else 0)
    conv = lambda o, i: o * i * 9 + o + 2 * o  # weight + bias + BN gamma/beta
    p = conv(filters, cin) + 2 * blocks * conv(filters, filters)
    p += (filters * n * n) * 1575 + 1575 + filters * n * n + 1
    return 4 * p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--rollouts", type=int, default=400, help="sims per move")
    ap.add_argument("--board", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--filters", type=int, default=64)
    ap.add_argument("--head", default="fc5", choices=["fc5", "conv"])
    ap.add_argument("--arena", type=int, default=0,
                    help="average MCTS node budget per game (all trees share one pool); 0 = sized by the engine from the free device memory")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra lines of the JSON (config C3, 16 384 games, the 120-ply sustained run)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-games", type=int, default=256)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3"],
                    help="tower arithmetic: exact f32 MFMA (default, the parity path) or split-bf16 (3 bf16 MFMAs per product)")
    ap.add_argument("--no-alt-precision", action="store_true", help="skip the second run on the bf16x3 path")
    ap.add_argument("--profile-every", type=int, default=8, help="time the tower convs of every k-th forward (0 = off)")
    ap.add_argument("--train", action="store_true",
                    help="report config C5 alone (data-parallel training step on the 10x128 network) as the JSON line")
    ap.add_argument("--no-train", action="store_true", help="skip extra.train_c5")
    ap.add_argument("--train-steps", type=int, default=2, help="optimiser steps timed (each = chunks-in-step chunks)")
    ap.add_argument("--train-chunk", type=int, default=500, help="examples per chunk (CHUNK_SIZE, network.rs:17)")
    ap.add_argument("--train-chunks-in-step", type=int, default=20, help="chunks per optimiser step (CHUNKS_IN_STEP, network.rs:18)")
    ap.add_argument("--train-blocks", type=int, default=10)
    ap.add_argument("--train-filters", type=int, default=128)
    ap.add_argument("--train-games", type=int, default=4096, help="concurrent games per GPU that produce the training examples")
    ap.add_argument("--train-example-rollouts", type=int, default=16, help="sims per move while producing the training examples")
    ap.add_argument("--train-timeout", type=float, default=300.0,
                    help="N > 1: seconds the C5 phase (first RCCL all-reduce inside libtakgpu) may take before the headline line is printed without it")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="launcher plumbing only, for machines without a GPU: rendezvous, barrier and the max / sum reductions "
                         "of the N ranks, no engine, no measurement ('value' is null)")
    ap.add_argument("--rehearse-hang-rank", type=int, default=-1,
                    help="with --rehearse-launch: this rank never joins the rehearsed collective (the watchdog must end the run, non-zero)")
    ap.add_argument("--rehearse-fail-rank", type=int, default=-1,
                    help="with --rehearse-launch: this rank raises before the rehearsed collective (every rank must leave at once, non-zero)")
    ap.add_argument("--rehearse-fail-in-collective-rank", type=int, default=-1,
                    help="with --rehearse-launch: this rank raises BEHIND the agreement, while the others wait in the rehearsed collective "
                         "(rank 0 must learn it from the store, print its line and every rank must leave, non-zero, long before the watchdog)")
    args = ap.parse_args()

    from tak_amd import dist as tdist

    # dmabuf IPC (RCCL between processes needs it on this driver): set before anything initialises the GPU, so the ranks behave
    # the same whether launch_ranks or torch.distributed.run started them
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "RANK" not in os.environ:
        # invoked directly: become the launcher (before anything in this process touches the GPU)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank, world, local_rank = tdist.env_rank()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    backend = os.environ.get("TAK_BENCH_BACKEND", "nccl")
    if args.rehearse_launch:
        # no engine, no GPU: what is exercised is launch_ranks / the torchrun environment, the rendezvous and the reductions
        dist = tdist.init("gloo", rank, world)
        if dist is not None:
            dist.barrier()
        dt, total = tdist.reduce_time_and_count(dist, 1.0 + rank, 1 + rank)
        devices = tdist.gather(dist, tdist.device_record(rank))
        t0 = time.perf_counter()
        tdist.ranks_reporting(dist)  # the rehearsal's stand-in for the reduction preflight: one small all-reduce, timed per rank
        preflight = tdist.per_rank_times(dist, time.perf_counter() - t0)
        out = {"metric": "launcher rehearsal (no measurement)", "value": None, "per_gpu_value": None, "unit": "node-expansions/s", "n_gpus": world,
               "ranks_reporting": tdist.ranks_reporting(dist), "steps": args.steps, "warmup": args.warmup, "rehearsal": True,
               "max_over_ranks": dt, "sum_over_ranks": total,
               # the self-describing fields of a real run, through the same code: per-rank device records, per-rank times, preflight
               "ms_per_step_by_rank": tdist.per_rank_times(dist, 1.0 + rank),
               "config": {"devices": devices, "devices_distinct": tdist.check_devices(devices, "gloo"), "backend": "gloo",
                          "switches_set": sorted(set(sum(tdist.gather(dist, switches_set()), [])))},
               "preflight_ms_by_rank": preflight}
        if args.rehearse_hang_rank >= 0 or args.rehearse_fail_rank >= 0 or args.rehearse_fail_in_collective_rank >= 0:
            # the two ways config C5 can go wrong at N > 1, acted out with the code the real phase uses: a rank that fails before
            # the collective (Stages: every rank leaves, nobody waits) and a collective that never completes (Watchdog)
            def phase(progress):
                progress["stage"] = "rehearsed stage"

                def local():
                    if rank == args.rehearse_fail_rank:
                        raise RuntimeError("rehearsed local failure")
                tdist.Stages(dist, rank).run("rehearsed stage", local)
                progress["stage"] = "rehearsed collective"
                if rank == args.rehearse_hang_rank:
                    time.sleep(3600.0)
                if rank == args.rehearse_fail_in_collective_rank:
                    raise RuntimeError("rehearsed failure between two reductions")  # behind the agreement, where only the board helps
                tdist.reduce_min(dist, 1.0)  # the "gradient all-reduce" the hanging / failed rank never joins
                return {"rehearsed": True}
            out, c5_failed = run_c5_phase(args, rank, world, out if rank == 0 else None, phase, dist=dist)
            finish(rank, world, out, c5_failed, dist)
            return
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    import torch

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible — the engine has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    # TAK_BENCH_BACKEND=gloo rehearses the N>1 path with several ranks on ONE card (RCCL refuses two ranks per device)
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = tdist.init(backend, rank, world, device=torch.device("cuda", local_rank) if backend == "nccl" else None)

    import tak_amd

    # who sits where: every rank's card as ITS engine sees it (tg_device_info), gathered; two ranks on one card under the RCCL
    # backend end the run here, on every rank, before anything is measured
    probe = tak_amd.Engine(args.board, evaluator=tak_amd.EVAL_DUMMY, max_batch=1, device=local_rank)
    devices = tdist.gather(dist, tdist.device_record(rank, probe))
    probe.close()
    try:
        distinct = tdist.check_devices(devices, backend)
    except RuntimeError as ex:
        print(f"bench.py: rank {rank}: {ex}", file=sys.stderr, flush=True)
        sys.exit(2)
    switches = sorted(set(sum(tdist.gather(dist, switches_set()), [])))
    launch_config = {"devices": devices, "devices_distinct": distinct, "backend": backend if world > 1 else None,
                     # TG_* A/B switches that are ON in some rank's environment (tg_debug_switches): [] on a measured run
                     "switches_set": switches}
    if switches:
        print(f"bench.py: rank {rank}: A/B switches are set: {switches} — the line does not describe the default path", file=sys.stderr, flush=True)

    net, tensors = make_weights(args.board, args.blocks, args.filters, args.head, seed=args.seed)
    steps_total = args.steps + args.warmup

    def barrier(eng):
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def run(precision, profile_every, cfg=None, steps=None, warmup=None, drain_every=0, rollouts=None):
        """W untimed + K timed plies of self-play on a fresh engine → (seconds, expansions, evals, profile).  cfg =
        (board, blocks, filters, head, games, weights) overrides the command line (the extra lines); drain_every > 0
        fetches the finished examples every so many plies inside the timed region, as a training loop would."""
        board, blocks, filters, head, games, weights = cfg or (args.board, args.blocks, args.filters, args.head, args.games, tensors)
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        eng = tak_amd.Engine(board, res_blocks=blocks, filters=filters,
                             policy_head=tak_amd.HEAD_FC5 if head == "fc5" else tak_amd.HEAD_CONV,
                             evaluator=tak_amd.EVAL_RESNET, max_batch=games, device=local_rank)
        if precision != "f32":
            eng.set_precision(precision)
        eng.load_state_dict(weights)
        ring = games * (min(steps + warmup, drain_every + 8 if drain_every else steps + warmup) + 2)
        eng.selfplay_create(games, arena_nodes=args.arena, seed=args.seed, rollouts=args.rollouts if rollouts is None else rollouts,
                            max_examples=max(1 << 14, ring), slot_base=tdist.slot_base(rank, games))
        for _ in range(warmup):
            eng.selfplay_step(1)
        barrier(eng)
        s0 = eng.selfplay_stats()
        if profile_every:
            eng.profile_enable(profile_every)
        drained = 0
        t0 = time.perf_counter()
        for k in range(steps):
            eng.selfplay_step(1)
            if drain_every and (k + 1) % drain_every == 0:
                drained += len(eng.selfplay_drain(games * (drain_every + 2))[0])
        eng.sync()
        torch.cuda.synchronize()
        dt_local = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        prof = eng.profile_read() if profile_every else None
        if profile_every:
            eng.profile_enable(0)
        s1 = eng.selfplay_stats()
        eng.close()
        run.last = {"games_finished": s1["games_finished"], "examples": s1["examples"], "drained": drained, "dropped_examples": s1["dropped_examples"]}
        return dt_local, s1["expansions"] - s0["expansions"], s1["evals"] - s0["evals"], prof

    out = None
    if not args.train:
        dt_local, expansions, evals, prof = run(args.precision, args.profile_every)
        dt, total_exp = tdist.reduce_time_and_count(dist, dt_local, expansions, device="cuda" if backend == "nccl" else "cpu")
        reporting = tdist.ranks_reporting(dist, device="cuda" if backend == "nccl" else "cpu")
        by_rank = tdist.per_rank_times(dist, dt_local / max(args.steps, 1))

        if rank == 0:
            out = {
                "metric": f"MCTS node-expansions/sec ({args.board}x{args.board} Tak, {args.rollouts} sims/move, {args.games} games/GPU)",
                "value": total_exp / dt,          # the whole job: all ranks' expansions over the slowest rank's time
                "per_gpu_value": total_exp / dt / world,
                "unit": "node-expansions/s",
                "n_gpus": world,
                "ranks_reporting": reporting,     # ranks whose counts are in `value` (a SUM all-reduce of 1 per rank)
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": 1000.0 * dt / max(args.steps, 1),
                "ms_per_step_by_rank": by_rank,   # every rank's own time (ms); `ms_per_step` is the slowest (a straggler shows here)
                "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": None,
                "dtype": "f32" if args.precision == "f32" else "bf16x3 (split f32: 3 bf16 MFMAs per product, f32 accumulate)",
                "data": "synthetic",
                "config": {
                    "workload": f"{args.board}x{args.board} Tak self-play, {args.games} concurrent games/GPU, {args.rollouts} sims/move, "
                                f"{args.blocks}-block x {args.filters}-filter resnet ({args.head} policy head), random-init weights, 1 step = 1 ply of all games",
                    "games_per_gpu": args.games, "sims_per_move": args.rollouts, "board": args.board,
                    "parallelism": f"games sharded x{world}, no data-path collective",
                    "expansions_timed": total_exp, "network_evals_rank0": evals,
                    **launch_config,
                },
            }
            if prof and prof["conv_launches"]:
                avg_ms = prof["conv_ms"] / prof["conv_launches"]
                achieved = prof["conv_flops"] / (avg_ms * 1e-3) / 1e12
                traffic = None  # not measured in this run: read from the committed counter pass of the same kernel
                pmc = os.path.join(ROOT, "profiles", "pmc_conv.json")
                if os.path.exists(pmc):
                    try:
                        traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                    except Exception:
                        traffic = None
                if args.precision == "f32":
                    kernel, peak = "k_tower_halo (fused conv0 + residual tower on a halo LDS image, f32 MFMA 16x16x4; one launch = 1+2R 3x3 convs)", F32_MFMA_PEAK_TFLOPS
                else:  # three bf16 MFMA passes per algorithmic product: the ceiling for algorithmic FLOPs is a third of the bf16 peak
                    kernel, peak = "k_tower_s3 (fused tower, 3 x bf16 MFMA 16x16x32 per product; peak = 2500 TFLOP/s dense bf16 / 3)", 2500.0 / 3
                out["roofline"] = {
                    "bound": "mfma", "kernel": kernel,
                    "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "traffic": traffic,
                    "traffic_source": "profiles/pmc_conv.json: a separate rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ pass over the same kernel "
                                      "(scripts/collect_evidence.sh), committed — not collected in this run" if traffic is not None else None,
                    "avg_launch_ms": avg_ms, "launches_timed": prof["conv_launches"],
                    "flops_per_launch": prof["conv_flops"], "rows_per_launch": prof["conv_rows"],
                    # what the MFMA pipe actually issued: the constant input planes (reserves, colour, fcd: 46 of the 72) enter
                    # layer 0 as a per-position bias, so fewer MFMAs run than the algorithmic count — `frac` can exceed this
                    "executed_flops_per_launch": prof.get("conv_flops_executed", prof["conv_flops"]),
                    "achieved_executed": prof.get("conv_flops_executed", prof["conv_flops"]) / (avg_ms * 1e-3) / 1e12,
                    "frac_executed": prof.get("conv_flops_executed", prof["conv_flops"]) / (avg_ms * 1e-3) / 1e12 / peak,
                    "forward_ms": prof["forward_ms"] / max(prof["forwards"], 1),
                }
            if world == 1 and args.precision == "f32" and not args.no_alt_precision and tak_amd_supports_bf16x3(args):
                # the same workload on the split-bf16 tower / policy FC (3 bf16 MFMAs per product, f32 accumulate): measured
                # deviation from the f32 forward ≤ 1.1e-5 relative on the policy, ≤ 5e-6 on the eval (tests/test_gpu_net.py),
                # inside the 1e-4 of the reference comparison; trees differ from the f32 run only through those last bits
                try:
                    dt2, exp2, _, prof2 = run("bf16x3", args.profile_every)
                    alt = {"precision": "bf16x3", "value": exp2 / dt2, "unit": "node-expansions/s", "ms_per_step": 1000.0 * dt2 / max(args.steps, 1),
                           "max_deviation_vs_f32_forward": measured_deviation(args, tensors, local_rank)}
                    if prof2 and prof2["conv_launches"]:
                        avg2 = prof2["conv_ms"] / prof2["conv_launches"]
                        alt["tower_avg_launch_ms"] = avg2
                        alt["tower_f32_equivalent_tflops"] = prof2["conv_flops"] / (avg2 * 1e-3) / 1e12
                        alt["tower_bf16_mfma_frac_of_peak"] = 3 * prof2["conv_flops"] / (avg2 * 1e-3) / 1e12 / 2500.0
                    out["alt_precision"] = alt
                except Exception as ex:
                    out["alt_precision"] = {"error": repr(ex)}
            if world == 1 and args.precision == "f32" and not args.no_extras:
                # more lines than the headline, same engine, same timing discipline (fresh engine, warm-up ply, sync on both sides):
                # the other single-GPU BASELINE config, the north star's "≥ 10 k concurrent games", and a long run with drains
                extras = {}
                try:
                    dt3, exp3, _, _ = run("f32", 0, steps=120, warmup=0, drain_every=10)
                    extras["sustained_120_plies"] = {"value": exp3 / dt3, "unit": "node-expansions/s", "plies": 120, "seconds": dt3,
                                                     "what": "the headline config from ply 0 for 120 plies, finished examples drained every 10 plies", **run.last}
                    dt4, exp4, _, _ = run("f32", 0, cfg=(args.board, args.blocks, args.filters, args.head, 4 * args.games, tensors), steps=2, warmup=1)
                    extras["games_x4"] = {"value": exp4 / dt4, "unit": "node-expansions/s", "games": 4 * args.games, "ms_per_step": 1000.0 * dt4 / 2}
                    if (args.board, args.blocks, args.filters) == (5, 6, 64):
                        net3, w3 = make_weights(6, 10, 128, "conv", seed=args.seed)
                        dt5, exp5, _, p5 = run("f32", 1, cfg=(6, 10, 128, "conv", args.games, w3), steps=2, warmup=1)
                        c3 = {"value": exp5 / dt5, "unit": "node-expansions/s", "ms_per_step": 1000.0 * dt5 / 2,
                              "workload": f"BASELINE config C3: 6x6 Tak, {args.games} games, {args.rollouts} sims/move, 10-block x 128-filter resnet, conv policy head",
                              "fp32_ceiling": F32_MFMA_PEAK_TFLOPS * 1e12 / 240_795_648}
                        if p5 and p5["conv_launches"]:
                            t5 = p5["conv_ms"] / p5["conv_launches"] * 1e-3
                            c3["tower_frac_of_f32_mfma_peak"] = p5["conv_flops"] / t5 / 1e12 / F32_MFMA_PEAK_TFLOPS
                            c3["tower_frac_executed"] = p5.get("conv_flops_executed", p5["conv_flops"]) / t5 / 1e12 / F32_MFMA_PEAK_TFLOPS
                            c3["tower_avg_launch_ms"] = t5 * 1e3
                        extras["config_c3"] = c3
                        # The reference's OWN constants, unchanged (train/src/self_play.rs:10-12,94: 32 lock-step games, ROLLOUTS = 10 000;
                        # alpha-tak/src/model/net6.rs:16-17: 6x6, 16 blocks x 128 filters, conv head): one leaf per game → a 32-position
                        # forward per iteration, the latency-bound end of the engine.  What a maintainer who changes no constant gets.
                        net6, w6 = make_weights(6, 16, 128, "conv", seed=args.seed)
                        dt6, exp6, _, p6 = run("f32", 16, cfg=(6, 16, 128, "conv", 32, w6), steps=2, warmup=1, rollouts=10_000)
                        rc = {"value": exp6 / dt6, "unit": "node-expansions/s", "ms_per_step": 1000.0 * dt6 / 2, "us_per_iteration": 1e6 * dt6 / max(exp6 / 32, 1),
                              "games": 32, "sims_per_move": 10_000,
                              "workload": "the reference's constants: 6x6 Tak, 32 lock-step games, 10 000 rollouts per move, Net6 = 16-block x 128-filter resnet, "
                                          "conv policy head; 2 plies timed after 1 warm-up ply"}
                        if p6 and p6["forwards"]:
                            rc["forward_us_at_32_leaves"] = 1000.0 * p6["forward_ms"] / p6["forwards"]  # HIP events around one network forward
                            rc["tree_and_launch_us_per_iteration"] = rc["us_per_iteration"] - rc["forward_us_at_32_leaves"]
                        extras["reference_constants"] = rc
                except Exception as ex:
                    extras["error"] = repr(ex)
                out["extra"] = extras
    # config C5 on every rank (the only part of the bench with a data-path collective)
    c5_failed = False
    if args.train or not (args.no_train or args.no_extras or args.precision != "f32"):
        out, c5_failed, c5 = run_c5_phase(args, rank, world, out, lambda progress: train_c5(args, rank, world, local_rank, dist, backend, barrier, progress), want_result=True, dist=dist)
        if rank == 0 and args.train and (c5_failed or "error" in c5):
            out = failed_train_line(world, c5.get("error"))
        elif rank == 0 and args.train:
            out = {
                "metric": c5["metric"], "value": c5["value"], "per_gpu_value": c5["value"] / world, "unit": c5["unit"], "n_gpus": world,
                "ranks_reporting": c5["ranks_reporting"], "steps": c5["optimizer_steps"], "warmup": 0,
                "ms_per_step": c5["ms_per_optimizer_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic (examples from the network's own self-play)",
                "ms_per_step_by_rank": c5["ms_per_optimizer_step_by_rank"],
                "config": {"workload": c5["workload"], "parallelism": f"data parallel x{world}, one gradient all-reduce per optimiser step", **launch_config},
                "roofline": {"bound": "mfma", "kernel": "whole training step (forward + data gradients + weight gradients + BatchNorm + Adam)",
                             "achieved": None if c5["frac_of_f32_mfma_peak"] is None else c5["frac_of_f32_mfma_peak"] * F32_MFMA_PEAK_TFLOPS,
                             "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": c5["frac_of_f32_mfma_peak"], "traffic": None},
                "train_c5": c5,
            }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and not args.train:
            try:
                out["cpu_baseline"] = cpu_baseline(args, net)
            except Exception as ex:  # the checker failing must not hide the GPU number
                out["cpu_baseline"] = {"error": repr(ex)}
    finish(rank, world, out, c5_failed, dist)


if __name__ == "__main__":
    main()
