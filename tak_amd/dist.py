"""Multi-GPU plumbing for sharded self-play: one process per GPU, games sharded by slot, no data-path
collective.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU in tests)
is used only for the barrier and for reducing the timing / counters that rank 0 reports."""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def slot_base(rank, games_per_rank):
    """Global index of a rank's first game: RNG streams are keyed by the global slot, so the union of the
    shards' games is the same set of games whatever the number of ranks."""
    return rank * games_per_rank


def init(backend, rank, world, device=None):
    if world <= 1:
        return None
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    kw = {}
    if device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def reduce_time_and_count(dist, dt_local, count_local, device="cpu"):
    """(max over ranks of the elapsed time, sum over ranks of the work count)"""
    if dist is None:
        return dt_local, count_local
    import torch

    t = torch.tensor([dt_local], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(count_local)], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c.item())


def ranks_reporting(dist, device="cpu"):
    """how many ranks took part in the reductions behind the reported total (a SUM all-reduce of 1 per rank): the line says so
    itself instead of leaving it to be inferred from WORLD_SIZE"""
    if dist is None:
        return 1
    import torch

    t = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def reduce_min(dist, value, device="cpu"):
    if dist is None:
        return value
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t.item()


def gather(dist, obj):
    """[obj of rank 0, obj of rank 1, …] on every rank (all_gather_object; a one-element list without a process group)"""
    if dist is None:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def device_record(rank, engine=None):
    """what one rank reports about the card it sits on (config.devices of the bench line): the engine's own view through
    tg_device_info — HIP ordinal, PCI bus id, name, CU count — plus host and pid; without an engine (launcher rehearsals on a
    machine without a GPU) the device fields are None"""
    import socket

    rec = {"rank": rank, "hip_device": None, "pci_bus_id": None, "name": None, "cu_count": None, "host": socket.gethostname(), "pid": os.getpid()}
    if engine is not None:
        info = engine.device_info()
        rec.update({k: info[k] for k in ("hip_device", "pci_bus_id", "name", "cu_count")})
        rec["arch"] = info["arch"]
    return rec


def devices_distinct(records):
    """True when no two ranks of one host report the same PCI bus id (ranks without a device record — rehearsals — count as distinct)"""
    seen = set()
    for r in records:
        if r.get("pci_bus_id") is None:
            continue
        key = (r.get("host"), r["pci_bus_id"])
        if key in seen:
            return False
        seen.add(key)
    return True


def check_devices(records, backend):
    """One rank per GPU is the launch contract (reference train/src/self_play.rs:98,102-104: one shard per process): under the RCCL
    backend two ranks on one card are an error — `ranks_reporting` alone cannot show it, the whole-job figure would count one
    card twice.  Under gloo (several ranks share a card on purpose: one-GPU rehearsals) the line just says devices_distinct = false."""
    distinct = devices_distinct(records)
    if backend == "nccl" and not distinct:
        by = {}
        for r in records:
            by.setdefault((r.get("host"), r.get("pci_bus_id")), []).append(r["rank"])
        shared = {f"{k[1]}": v for k, v in by.items() if len(v) > 1}
        raise RuntimeError(f"ranks share a GPU under the RCCL backend: {shared} (PCI bus id → ranks); check HIP_VISIBLE_DEVICES / LOCAL_RANK")
    return distinct


def per_rank_times(dist, seconds, scale=1000.0):
    """every rank's own time beside the max the headline is computed from: a straggler is otherwise invisible under the
    max-reduce → {"all": [ms of rank 0, …], "min", "max", "fastest_rank", "slowest_rank"}"""
    all_ms = [scale * float(t) for t in gather(dist, float(seconds))]
    lo, hi = min(all_ms), max(all_ms)
    return {"all": all_ms, "min": lo, "max": hi, "fastest_rank": all_ms.index(lo), "slowest_rank": all_ms.index(hi)}


class Stages:
    """Rank-local pieces of work in front of a collective.  `run(name, fn)` executes fn on this rank, then all ranks agree
    (one MIN all-reduce of an ok flag — which is also the barrier in front of the next piece): the rank that failed re-raises its
    own error, every other rank raises "another rank failed" — so nobody enters the next collective (a communicator rendezvous,
    tg_train's gradient all-reduce) to wait there for a rank that will never arrive."""

    def __init__(self, dist, rank, device="cpu"):
        self.dist, self.rank, self.device = dist, rank, device

    def agree(self, name, error=None):
        ok = reduce_min(self.dist, 0.0 if error else 1.0, device=self.device)
        if error:
            raise error
        if ok < 1.0:
            raise RuntimeError(f"rank {self.rank}: another rank failed during {name}; leaving before the next collective")

    def run(self, name, fn):
        try:
            result, failure = fn(), None
        except Exception as ex:  # noqa: BLE001 — re-raised by agree() on this rank, reported to the others
            result, failure = None, ex
        self.agree(name, failure)
        return result


class FailureBoard:
    """A side channel beside the collectives, for the phase whose collectives may hang: the rendezvous store torch.distributed
    already runs (hosted by rank 0's process, served by its own thread — reachable while rank 0's main thread sits in a collective).
    A rank that fails where no agreement protects it (inside tg_train, between two reductions) posts here and then waits for
    rank 0's acknowledgement before it exits non-zero: the launcher (launch_ranks, torch.distributed.run) ends every rank at the
    first non-zero exit, and rank 0 — still waiting in the all-reduce for the rank that failed — must have printed its headline
    line by then.  Rank 0 polls the board from a thread while the phase runs (bench.run_c5_phase)."""

    KEY, ACK = "tak_c5_failure", "tak_c5_rank0_printed"

    def __init__(self, dist):
        self.store = None
        if dist is not None:
            try:
                from torch.distributed.distributed_c10d import _get_default_store

                self.store = _get_default_store()
            except Exception:  # noqa: BLE001 — no store: the watchdog alone bounds the phase
                self.store = None

    def post(self, rank, what):
        if self.store is not None:
            self.store.set(self.KEY, f"rank {rank}: {what}")

    def posted(self):
        """the posted message, or None"""
        if self.store is None:
            return None
        try:
            return self.store.get(self.KEY).decode() if self.store.check([self.KEY]) else None
        except Exception:  # noqa: BLE001
            return None

    def acknowledge(self):
        if self.store is not None:
            self.store.set(self.ACK, "1")

    def wait_acknowledged(self, seconds):
        import time

        t0 = time.time()
        while self.store is not None and time.time() - t0 < seconds:
            try:
                if self.store.check([self.ACK]):
                    return True
            except Exception:  # noqa: BLE001 — rank 0 (the store's host) is gone: nothing left to wait for
                return False
            time.sleep(0.1)
        return False


def training_shard(n_examples, rank, world, chunk_size):
    """[begin, end) of a rank's examples for data-parallel training: every rank gets the same number of WHOLE chunks
    (the gradient all-reduce inside tg_train_chunk's optimiser step must be entered equally often on every rank);
    the remainder is dropped, as Network::train's chunks_exact does (alpha-tak/src/model/network.rs:53)."""
    chunks = n_examples // chunk_size
    per_rank = chunks // world
    begin = rank * per_rank * chunk_size
    return begin, begin + per_rank * chunk_size


def broadcast_unique_id(dist, make_id, device="cpu"):
    """Rank 0 creates the 128-byte RCCL unique id (tak_amd.comm_unique_id), every rank receives it."""
    import torch

    if dist is None:
        return make_id()
    buf = torch.zeros(128, dtype=torch.uint8, device=device)
    if dist.get_rank() == 0:
        buf.copy_(torch.frombuffer(bytearray(make_id()), dtype=torch.uint8))
    dist.broadcast(buf, src=0)
    return bytes(buf.cpu().numpy().tobytes())


def host_allreduce_hook(dist):
    """A reduction for Engine.train_set_allreduce that sums over the ranks of a torch.distributed group on the HOST
    (gloo): device buffer → host, all_reduce(SUM), host → device.  For ranks that share one GPU (RCCL refuses duplicate
    devices) and for CPU-side rehearsals of the data-parallel path; real multi-GPU training uses tg_train_comm_init (RCCL)."""
    import torch

    from . import engine as eng

    def fn(d_buf, count, stream):
        host = eng.device_to_host(d_buf, count, stream)
        t = torch.from_numpy(host)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        eng.host_to_device(d_buf, t.numpy())
        return 0

    return fn
