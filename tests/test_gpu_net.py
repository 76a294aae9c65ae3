"""GPU parity of the network forward (through the C ABI) against PyTorch-CPU fp32 (the same ATen ops
tch-rs calls).  Tolerance from BASELINE.json's north_star: |Δ| ≤ 1e-4 on policy and eval, fp32."""
import glob
import os

import numpy as np
import pytest

import torch_ref

pytestmark = pytest.mark.gpu
TOL = 1e-4
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "net*.npz")))


def _engine(n, blocks, filters, head, max_batch=256):
    import tak_amd

    return tak_amd.Engine(n, res_blocks=blocks, filters=filters, policy_head=tak_amd.HEAD_FC5 if head == "fc5" else tak_amd.HEAD_CONV,
                          evaluator=tak_amd.EVAL_RESNET, max_batch=max_batch)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_fixture(path, orc):
    z = np.load(path)
    n, blocks, filters, head_i, seed = [int(v) for v in z["meta"]]
    head = "fc5" if head_i == 0 else "conv"
    net = torch_ref.make_net(n, blocks, filters, head, seed=seed)
    # the regenerated weights reproduce the committed outputs on CPU (same PyTorch build)
    p_cpu, v_cpu = torch_ref.forward(net, orc.encode(n, z["states"]))
    assert np.abs(p_cpu - z["policy"]).max() < 1e-6 and np.abs(v_cpu - z["eval"]).max() < 1e-6
    e = _engine(n, blocks, filters, head)
    e.load_state_dict(torch_ref.abi_tensors(net))
    p, v = e.policy_eval(z["states"])
    assert np.abs(p - z["policy"]).max() <= TOL
    assert np.abs(v - z["eval"]).max() <= TOL
    assert np.abs(p.sum(1) - 1).max() < 1e-5
    # forward_mcts on caller-encoded planes gives the same numbers (to the last bits: the fused towers take the constant
    # planes of a packed state as a per-position bias, arbitrary caller planes as data — another summation order in layer 0)
    p2, v2 = e.forward_mcts(orc.encode(n, z["states"]))
    assert np.abs(p - p2).max() <= 1e-6 and np.abs(v - v2).max() <= 1e-6
    assert np.abs(p2 - z["policy"]).max() <= TOL and np.abs(v2 - z["eval"]).max() <= TOL
    e.close()


@pytest.mark.parametrize("n,blocks,filters,head,batch", [
    (5, 6, 64, "fc5", 300),     # BASELINE config C2 topology
    (6, 10, 128, "conv", 130),  # config C3 topology
    (5, 10, 128, "fc5", 77),    # config C5 topology
    (5, 2, 64, "conv", 64),     # conv head on 5x5
    (3, 1, 32, "conv", 50),     # DummyNet-sized board (search/tests.rs)
    (5, 0, 128, "fc5", 100),    # no residual block: conv0 and the heads alone (the split tower without an exchange)
    (6, 0, 128, "conv", 40),
])
def test_config_topologies_vs_torch(orc, n, blocks, filters, head, batch):
    net = torch_ref.make_net(n, blocks, filters, head, seed=n * 100 + blocks)
    sts = orc.random_positions(n, batch, seed=5, max_plies=80 if n >= 5 else 12, half_komi=4)
    p_ref, v_ref = torch_ref.forward(net, orc.encode(n, sts))
    e = _engine(n, blocks, filters, head, max_batch=128)  # forces chunking for batch > 128
    e.load_state_dict(torch_ref.abi_tensors(net))
    p, v = e.policy_eval(sts)
    assert np.abs(p - p_ref).max() <= TOL, np.abs(p - p_ref).max()
    assert np.abs(v - v_ref).max() <= TOL, np.abs(v - v_ref).max()
    # relative check on the policy too (probabilities are ~1e-3, so 1e-4 absolute alone is lax)
    assert (np.abs(p - p_ref) / p_ref).max() < 5e-4
    # per-position results do not depend on the batch they are evaluated in
    p1, v1 = e.policy_eval(sts[3:4])
    assert np.array_equal(p1[0], p[3]) and v1[0] == v[3]
    e.close()


@pytest.mark.parametrize("n,blocks,filters,head,batch", [
    (5, 6, 64, "fc5", 300),     # C2
    (6, 10, 128, "conv", 70),   # C3
    (5, 10, 128, "fc5", 77),    # C5 network
    (5, 2, 64, "conv", 45),     # conv head on 5x5 (two head passes of two channel groups)
    (5, 1, 128, "conv", 9),     # one partial workgroup
    (5, 0, 128, "fc5", 100),    # no residual block (round 6: the halo form of the split-bf16 tower mishandled it at ≥ 256 positions)
])
def test_bf16x3_tower_within_tolerance(orc, n, blocks, filters, head, batch):
    """The split-bf16 tower (TG_PRECISION_BF16X3): same 1e-4 gate as the exact path, and its measured deviation."""
    net = torch_ref.make_net(n, blocks, filters, head, seed=n * 100 + blocks)
    sts = orc.random_positions(n, batch, seed=5, max_plies=80, half_komi=4)
    p_ref, v_ref = torch_ref.forward(net, orc.encode(n, sts))
    e = _engine(n, blocks, filters, head, max_batch=128)
    e.set_precision("bf16x3")
    e.load_state_dict(torch_ref.abi_tensors(net))
    p, v = e.policy_eval(sts)
    assert np.abs(p - p_ref).max() <= TOL and np.abs(v - v_ref).max() <= TOL
    # observed: ≈ 1e-5 relative on the policy, a few 1e-6 on the eval (16 mantissa bits per operand)
    assert (np.abs(p - p_ref) / p_ref).max() < 1e-4, (np.abs(p - p_ref) / p_ref).max()
    assert np.abs(v - v_ref).max() < 2e-5, np.abs(v - v_ref).max()
    # batch independence and the planes entry point
    p1, v1 = e.policy_eval(sts[3:4])
    assert np.array_equal(p1[0], p[3]) and v1[0] == v[3]
    p2, v2 = e.forward_mcts(orc.encode(n, sts))  # (planes entry: every plane through layer 0 — another summation order)
    assert np.abs(p - p2).max() <= 2e-5 and np.abs(v - v2).max() <= 2e-5
    assert np.abs(p2 - p_ref).max() <= TOL and np.abs(v2 - v_ref).max() <= TOL
    # back to the exact path on the same engine
    e.set_precision("f32")
    e.load_state_dict(torch_ref.abi_tensors(net))
    pf, vf = e.policy_eval(sts)
    assert np.abs(pf - p_ref).max() <= TOL and not np.array_equal(pf, p)
    print(f"bf16x3 vs torch: policy rel {(np.abs(p - p_ref) / p_ref).max():.2e}, eval abs {np.abs(v - v_ref).max():.2e}; "
          f"f32: policy rel {(np.abs(pf - p_ref) / p_ref).max():.2e}, eval abs {np.abs(vf - v_ref).max():.2e}")
    e.close()


def test_weight_errors():
    import tak_amd

    e = _engine(5, 1, 32, "fc5")
    with pytest.raises(tak_amd.TgError) as ei:
        e.policy_eval(np.zeros((1, 256), np.uint8))
    assert ei.value.code == -7  # not finalized
    net = torch_ref.make_net(5, 1, 32, "fc5")
    t = torch_ref.abi_tensors(net)
    del t["res0.bn2.running_var"]
    with pytest.raises(tak_amd.TgError) as ei:
        e.load_state_dict(t)
    assert ei.value.code == -4 and "res0.bn2.running_var" in str(ei.value)
    assert e.policy_eval(np.zeros((0, 256), np.uint8))[0].shape == (0, 1575)  # empty batch is fine (net5.rs:121)
    e.close()


@pytest.mark.parametrize("n,head", [(5, "fc5"), (6, "conv")])
def test_default_init_matches_tch_defaults_and_torch_forward(orc, n, head):
    """Network::default() (net5.rs:29-73): tg_net_init_random draws tch's default initialisers; the tensors read back
    through tg_net_get_tensor have those distributions, and the forward on them equals PyTorch's on the same tensors."""
    import tak_amd

    F, R = 32, 2
    e = _engine(n, R, F, head)
    e.init_random(seed=5)
    P = tak_amd.policy_size(n, tak_amd.HEAD_FC5 if head == "fc5" else tak_amd.HEAD_CONV)
    cin = tak_amd.input_channels(n)
    shapes = {"conv0.weight": (F, cin, 3, 3), "conv0.bias": (F,), "bn0.weight": (F,), "bn0.bias": (F,), "bn0.running_mean": (F,),
              "bn0.running_var": (F,), "value.weight": (1, F * n * n), "value.bias": (1,)}
    for i in range(R):
        for c in ("conv1", "conv2"):
            shapes[f"res{i}.{c}.weight"] = (F, F, 3, 3)
            shapes[f"res{i}.{c}.bias"] = (F,)
        for b in ("bn1", "bn2"):
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                shapes[f"res{i}.{b}.{leaf}"] = (F,)
    if head == "fc5":
        shapes["policy.weight"], shapes["policy.bias"] = (P, F * n * n), (P,)
    else:
        shapes["policy.weight"], shapes["policy.bias"] = (P // (n * n), F, 3, 3), (P // (n * n),)
    t = {k: e.get_tensor(k, s) for k, s in shapes.items()}
    w = t["res1.conv2.weight"]
    bound = 1.0 / np.sqrt(F * 9)
    assert np.abs(w).max() <= bound and abs(w.mean()) < 0.05 * bound and abs(w.std() - bound / np.sqrt(3)) < 0.03 * bound
    assert np.all(t["res0.conv1.bias"] == 0) and np.all(t["bn0.bias"] == 0) and np.all(t["res1.bn2.running_var"] == 1)
    g = np.concatenate([t[k] for k in t if k.endswith(("bn0.weight", "bn1.weight", "bn2.weight"))])
    assert 0 <= g.min() and g.max() <= 1 and abs(g.mean() - 0.5) < 0.08
    pb = t["policy.bias"]
    if head == "fc5":
        lb = 1.0 / np.sqrt(F * n * n)
        assert np.abs(pb).max() <= lb and pb.std() > 0.4 * lb and np.abs(t["policy.weight"]).max() <= lb
    else:
        assert np.all(pb == 0)
    assert not np.array_equal(t["res0.conv1.weight"], t["res0.conv2.weight"])
    # another seed → other values; the same seed → the same values
    e2 = _engine(n, R, F, head)
    e2.init_random(seed=6)
    assert not np.array_equal(e2.get_tensor("conv0.weight", shapes["conv0.weight"]), t["conv0.weight"])
    e2.init_random(seed=5)
    assert np.array_equal(e2.get_tensor("conv0.weight", shapes["conv0.weight"]), t["conv0.weight"])
    e2.close()
    # forward on these tensors = PyTorch forward on the same tensors
    net = torch_ref.make_net(n, R, F, head)
    torch_ref.load_abi_tensors(net, t)
    states = orc.random_positions(n, 24, seed=3, max_plies=40, half_komi=4)
    pol, ev = e.policy_eval(states)
    rp, rv = torch_ref.forward(net, orc.encode(n, states))
    assert np.abs(pol - rp).max() <= TOL and np.abs(ev - rv).max() <= TOL
    e.close()


def test_c_host_runs_selfplay_through_the_abi(tmp_path):
    """examples/selfplay_host.c: a C99 program against takgpu.h + libtakgpu.so only (perft KAT, policy_eval, self-play,
    drain, example text)"""
    import subprocess

    import test_abi

    exe = test_abi._build_c_host(tmp_path)
    r = subprocess.run([exe, "128", "24", "70"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "perft(5x5, depth 3) = 43320" in r.stdout and r.stdout.strip().endswith("OK")


@pytest.mark.parametrize("n,blocks,filters,head,sizes", [
    (5, 6, 64, "fc5", (1, 200, 300, 700, 1500, 2500)),   # C2: every positions-per-workgroup variant of the tower + both FC kernels
    # C5 network shape; ≤ 128 positions of a 128-filter network run k_tower_split (round 6: a position split over 8 / 4 / 2
    # workgroups by channel tile, slices exchanged through global memory between layers): 1 … 32, 33 … 64 and 65 … 128 positions
    (5, 2, 128, "fc5", (1, 17, 32, 33, 64, 100, 128, 129, 255, 300, 600, 1100)),
    (6, 2, 128, "conv", (5, 32, 40, 64, 65, 128, 129, 260, 520)), # C3 shape, conv head
    (5, 2, 64, "conv", (40, 500, 1100, 2100)),            # conv head on 5×5
    (5, 0, 128, "fc5", (9, 128, 300)),                    # no residual block: the split tower's layer loop without an exchange
    (6, 1, 128, "conv", (3, 64, 65, 300)),                # one block, 6×6: three layers, two exchanges
])
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_result_does_not_depend_on_the_batch_size(orc, n, blocks, filters, head, sizes, precision):
    """Small batches run instantiations with fewer positions per workgroup (and a barrier-free FC) so that a 32-leaf call does
    not take as long as a 4096-leaf one, large ones the halo-image towers (k_tower_halo, k_tower_s3_halo); the per-element
    arithmetic is the same, so a position's outputs must be the same BITS whatever batch it is evaluated in."""
    net = torch_ref.make_net(n, blocks, filters, head, seed=11)
    total = max(sizes)
    sts = orc.random_positions(n, 64, seed=9, max_plies=60, half_komi=4)
    sts = np.concatenate([sts] * ((total + 63) // 64))[:total]
    e = _engine(n, blocks, filters, head, max_batch=total)
    if precision != "f32":
        e.set_precision(precision)
    e.load_state_dict(torch_ref.abi_tensors(net))
    p_ref, v_ref = e.policy_eval(sts)
    assert np.array_equal(p_ref[:64], p_ref[64:128]) and np.abs(p_ref.sum(1) - 1).max() < 1e-5
    for k in sizes[:-1]:
        p, v = e.policy_eval(sts[:k])
        assert np.array_equal(p, p_ref[:k]) and np.array_equal(v, v_ref[:k]), k
    e.close()


def test_split_towers_of_several_engines_at_once(orc):
    """k_tower_split's workgroups wait for each other (a position's 8 siblings meet at a counter between two layers).  Three engines on
    three host threads — three streams, three sets of counters and exchange buffers — launch it at the same time, 150 forwards each at
    batch sizes that fill the chip several times over between them: nothing hangs (the waits are bounded and a bound reached is an error),
    and every forward returns the bits of a forward that ran alone."""
    import threading

    n, blocks, filters, head = 5, 4, 128, "fc5"
    net = torch_ref.make_net(n, blocks, filters, head, seed=21)
    tensors = torch_ref.abi_tensors(net)
    sts = orc.random_positions(n, 200, seed=12, max_plies=60, half_komi=4)[:128]
    engines = []
    for _ in range(3):
        e = _engine(n, blocks, filters, head, max_batch=128)
        e.load_state_dict(tensors)
        engines.append(e)
    sizes = (128, 96, 64, 33, 128, 7)
    alone = {k: engines[0].policy_eval(sts[:k]) for k in set(sizes)}
    errors = []

    def worker(e, shift):
        try:
            for it in range(150):
                k = sizes[(it + shift) % len(sizes)]
                p, v = e.policy_eval(sts[:k])
                if not (np.array_equal(p, alone[k][0]) and np.array_equal(v, alone[k][1])):
                    errors.append(f"engine {shift}, forward {it}, batch {k}: bits differ from the forward that ran alone")
                    return
        except Exception as ex:  # noqa: BLE001 — reported by the main thread
            errors.append(repr(ex))

    threads = [threading.Thread(target=worker, args=(e, i)) for i, e in enumerate(engines)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a forward did not return"
    assert not errors, errors[:3]
    for e in engines:
        e.close()


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("n,blocks,filters,head,batch", [(5, 2, 64, "fc5", 2500), (6, 1, 128, "conv", 700)])
def test_planes_entry_equals_states_entry_at_full_batch(orc, n, blocks, filters, head, batch, precision):
    """tg_forward_mcts (caller-encoded planes, net5.rs:121-131) runs the halo towers' other staging path: same bits as
    tg_policy_eval on the packed states (whose planes the tower encodes itself)."""
    net = torch_ref.make_net(n, blocks, filters, head, seed=4)
    sts = orc.random_positions(n, 64, seed=2, max_plies=70, half_komi=4)
    sts = np.concatenate([sts] * ((batch + 63) // 64))[:batch]
    e = _engine(n, blocks, filters, head, max_batch=batch)
    if precision != "f32":
        e.set_precision(precision)
    e.load_state_dict(torch_ref.abi_tensors(net))
    p, v = e.policy_eval(sts)
    p2, v2 = e.forward_mcts(orc.encode(n, sts))
    # the towers (exact f32 and split-bf16 alike) take a packed state's constant planes (reserves, colour, fcd) as a per-position
    # bias and run layer 0 over the board planes only; caller-encoded planes are arbitrary data and go through layer 0 whole:
    # the same sums in another order, equal to the last bits but not bit for bit — both within 1e-4 of PyTorch
    tol = 1e-6 if precision == "f32" else 2e-5
    assert np.abs(p - p2).max() <= tol and np.abs(v - v2).max() <= tol
    p_ref, v_ref = torch_ref.forward(net, orc.encode(n, sts[:256]))
    for pp, vv in ((p, v), (p2, v2)):
        assert np.abs(pp[:256] - p_ref).max() <= TOL and np.abs(vv[:256] - v_ref).max() <= TOL
    e.close()


@pytest.mark.parametrize("n,blocks,filters,head", [(5, 2, 32, "fc5"), (6, 2, 32, "conv")])
def test_checkpoint_archive_loads_into_the_engine(orc, tmp_path, n, blocks, filters, head):
    """N4: a tch VarStore archive (Network::save, net5.rs:95-104) written by tak_amd.checkpoint → read back → tg_net_set_tensor →
    tg_policy_eval: the same outputs as the engine fed the source tensors directly, and as PyTorch on them; and
    tg_net_get_tensor returns what was loaded (what Network::save would write).  Still not a file written by tch itself."""
    import tak_amd
    from tak_amd import checkpoint

    net = torch_ref.make_net(n, blocks, filters, head, seed=21)
    tensors = torch_ref.abi_tensors(net)
    path = str(tmp_path / "net.model")
    checkpoint.save_tch_varstore(path, tensors, blocks)
    loaded = checkpoint.load_tch_varstore(path, blocks)
    sts = orc.random_positions(n, 96, seed=5, max_plies=40, half_komi=4)
    outs = []
    for src in (tensors, loaded):
        e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, policy_head=tak_amd.HEAD_FC5 if head == "fc5" else tak_amd.HEAD_CONV,
                           evaluator=tak_amd.EVAL_RESNET, max_batch=128)
        e.load_state_dict(src)
        outs.append(e.policy_eval(sts))
        if src is loaded:
            for k, v in tensors.items():
                assert np.array_equal(e.get_tensor(k, v.shape), v), k
        e.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    p_ref, v_ref = torch_ref.forward(net, orc.encode(n, sts))
    assert np.abs(outs[1][0] - p_ref).max() <= 1e-4 and np.abs(outs[1][1] - v_ref).max() <= 1e-4


@pytest.mark.parametrize("stem,n,blocks,filters", [("tch_net4_conv_1x32", 4, 1, 32), ("tch_net6_conv_1x32", 6, 1, 32)])
def test_archive_written_by_libtorch_loads_into_the_engine(stem, n, blocks, filters):
    """N4 against the real container: tests/golden/tch_*.model were written by libtorch's OutputArchive (write + save_to: the
    calls behind tch's VarStore::save, tests/golden/tch_archive_writer.cpp), with tch-style variable names.  Archive →
    tak_amd.checkpoint → tg_net_set_tensor → tg_policy_eval must give the outputs recorded with PyTorch-CPU on the same
    weights (≤ 1e-4).  The variable NAMES remain the recalled tch convention (tch is not vendored in the reference)."""
    import os

    import tak_amd
    from tak_amd import checkpoint

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    loaded = checkpoint.load_tch_varstore(os.path.join(golden, stem + ".model"), blocks)
    exp = np.load(os.path.join(golden, stem + ".expected.npz"))
    e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, policy_head=tak_amd.HEAD_CONV, evaluator=tak_amd.EVAL_RESNET, max_batch=64)
    e.load_state_dict(loaded)
    p, v = e.policy_eval(exp["states"])
    assert np.abs(p - exp["policy"]).max() <= 1e-4 and np.abs(v - exp["value"]).max() <= 1e-4
    e.close()
