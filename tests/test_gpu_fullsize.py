"""BASELINE.json's full sizes (4096 concurrent games, the C2 network) checked through size-independent
properties, plus bit-exact comparison of a slice against the oracle."""
import numpy as np
import pytest

import posgen
import torch_ref

pytestmark = pytest.mark.gpu
G = 4096
G4 = 16384  # bench.py's extra.games_x4 and the north star's "≥ 10 k concurrent games": the widest engine a line of the bench reports


@pytest.fixture(scope="module")
def c2(orc):
    import tak_amd

    # (round 5: randomised BatchNorm affine parameters and running statistics — k_tower_halo and the constant-planes-as-bias table
    # meet a non-trivial fold against PyTorch directly, on the full batch)
    net = torch_ref.make_net(5, 6, 64, "fc5", seed=0, randomize_bn=True)
    tensors = torch_ref.abi_tensors(net)
    e = tak_amd.Engine(5, res_blocks=6, filters=64, evaluator=tak_amd.EVAL_RESNET, max_batch=G)
    e.load_state_dict(tensors)
    yield e, net, tensors
    e.close()


def _roots(orc, count):
    """`count` DISTINCT ongoing 5×5 positions (round 6; rounds 1 – 5 tiled ≈ 2 930 to 4096)"""
    return posgen.distinct_positions(orc, 5, count, seed=21, max_plies=60)


def test_policy_eval_full_batch_properties(c2, orc):
    e, net, _ = c2
    sts = _roots(orc, G)
    p, v = e.policy_eval(sts)
    assert np.abs(p.sum(1) - 1).max() < 1e-5 and (p > 0).all() and (np.abs(v) <= 1).all()
    assert len({s.tobytes() for s in sts}) == G
    # a position's output does not depend on the batch slot it sits in (tiles are batch-position independent): the same 4096
    # positions in another order → the same rows, bit for bit
    perm = np.random.default_rng(5).permutation(G)
    p2, v2 = e.policy_eval(sts[perm])
    assert np.array_equal(p2, p[perm]) and np.array_equal(v2, v[perm])
    # ALL 4096 rows of the full batch — the batch the benchmarked kernels (k_tower_halo, k_fc_ring) run on — against PyTorch
    # fp32 (north_star tolerance 1e-4), in chunks so the CPU reference stays in cache
    worst_p = worst_v = 0.0
    for lo in range(0, G, 512):
        p_ref, v_ref = torch_ref.forward(net, orc.encode(5, sts[lo : lo + 512]))
        worst_p = max(worst_p, float(np.abs(p[lo : lo + 512] - p_ref).max()))
        worst_v = max(worst_v, float(np.abs(v[lo : lo + 512] - v_ref).max()))
    print(f"c2_f32 (randomised BatchNorm fold): worst |dp| {worst_p:.3e}, worst |dv| {worst_v:.3e} over all {G} rows")
    assert worst_p <= 1e-4 and worst_v <= 1e-4, (worst_p, worst_v)


def _all_rows_against_pytorch(orc, n, blocks, filters, head, precision, rows, seed, chunk=512):
    """policy_eval on a FULL batch of 4096 positions (the kernel instantiations the benchmarks run) against PyTorch fp32 on
    `rows` (all 4096, or every 4th): worst |Δ| of policy and eval (north_star: ≤ 1e-4).  BatchNorm's affine parameters and running
    statistics are randomised (round 5), so the fold is not the identity."""
    import tak_amd

    net = torch_ref.make_net(n, blocks, filters, head, seed=seed, randomize_bn=True)
    e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, evaluator=tak_amd.EVAL_RESNET, max_batch=G)
    if precision != "f32":
        e.set_precision(precision)
    e.load_state_dict(torch_ref.abi_tensors(net))
    sts = posgen.distinct_positions(orc, n, G, seed=seed + 40, max_plies=70)
    assert len({s.tobytes() for s in sts}) == G
    p, v = e.policy_eval(sts)
    e.close()
    assert np.abs(p.sum(1) - 1).max() < 2e-5 and (p > 0).all() and (np.abs(v) <= 1).all()
    idx = np.arange(0, G, G // rows)
    worst_p = worst_v = 0.0
    for lo in range(0, len(idx), chunk):
        sel = idx[lo : lo + chunk]
        p_ref, v_ref = torch_ref.forward(net, orc.encode(n, sts[sel]))
        worst_p = max(worst_p, float(np.abs(p[sel] - p_ref).max()))
        worst_v = max(worst_v, float(np.abs(v[sel] - v_ref).max()))
    return worst_p, worst_v


@pytest.mark.parametrize("name,n,blocks,filters,head,precision,rows", [
    # the reference's own shipped topologies — the networks existing weights would load into — on the full-batch kernels, exact f32,
    # ALL 4096 rows: Net5 = 5×5, 8 blocks × 128 filters, FC head (alpha-tak/src/model/net5.rs:16-17, 76-111) and
    # Net6 = 6×6, 16 blocks × 128 filters, conv head (net6.rs:16-17, 75-109)
    ("net5_reference_8x128_f32", 5, 8, 128, "fc5", "f32", 4096),
    ("net6_reference_16x128_f32", 6, 16, 128, "conv", "f32", 4096),
    # the C5 network's full-batch tower k_tower_halo<…,37> and its ring FC, exact f32: ALL 4096 rows
    ("c5net_f32", 5, 10, 128, "fc5", "f32", 4096),
    # the split-bf16 towers k_tower_s3_halo and k_fc_s3b on the C2 network: ALL 4096 rows
    ("c2_bf16x3", 5, 6, 64, "fc5", "bf16x3", 4096),
    # the C5 network on the split-bf16 path (the 8-wave k_tower_s3_halo instantiation): every 4th row
    ("c5net_bf16x3", 5, 10, 128, "fc5", "bf16x3", 1024),
    # C3 (6×6, 10×128, conv-251 head inside the split tower): 1024 rows of the full batch
    ("c3_bf16x3", 6, 10, 128, "conv", "bf16x3", 1024),
])
def test_full_batch_rows_against_pytorch(orc, name, n, blocks, filters, head, precision, rows):
    worst_p, worst_v = _all_rows_against_pytorch(orc, n, blocks, filters, head, precision, rows, seed=3, chunk=512 if n == 5 else 128)
    print(f"{name}: worst |dp| {worst_p:.3e}, worst |dv| {worst_v:.3e} over {rows} rows of a 4096-position batch")
    assert worst_p <= 1e-4 and worst_v <= 1e-4, (name, worst_p, worst_v)


def test_search_full_size_invariants_and_slice_parity(c2, orc):
    import tak_amd

    e, net, tensors = c2
    iters = 24
    sts = _roots(orc, G)
    e.search_create(G, arena_nodes=1 << 13)
    e.search_reset(sts)
    e.search_run(iters)
    r = e.search_root()
    exp, ev = e.search_counters()
    assert exp == G * iters and ev <= exp
    assert (r["root_visits"] == iters).all()              # every rollout adds exactly one real visit to its root
    cs = np.array([r["visits"][g, : r["counts"][g]].sum() for g in range(G)])
    assert (cs == iters - 1).all()                         # the first visit expands the root itself
    assert np.array_equal(r["counts"], orc.movegen(5, sts)[1])
    assert (np.abs(r["q"]) <= 1.0 + 1e-6).all() and (np.abs(r["root_q"]) <= 1.0 + 1e-6).all()
    # 16 of the 4096 games, whole trees, bit for bit against the oracle fed by the same network kernels
    ev_eng = tak_amd.Engine(5, res_blocks=6, filters=64, evaluator=tak_amd.EVAL_RESNET, max_batch=64)
    ev_eng.load_state_dict(tensors)
    pick = np.arange(0, G, G // 16)
    s = orc.Search(5, head=orc.HEAD_FC5, py_eval=lambda st: ev_eng.policy_eval(st))
    s.reset(sts[pick])
    s.run(iters)
    for k, g in enumerate(pick):
        a, b = e.search_dump(int(g)), s.dump(k)
        assert len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in a.dtype.names), g
        assert (a["virtual_visits"] == 0).all()
    ev_eng.close()


def test_selfplay_full_size_two_plies(c2, orc):
    e, _, _ = c2
    rollouts = 16
    e.selfplay_create(G, arena_nodes=1 << 13, seed=9, rollouts=rollouts, max_examples=G * 4)
    e.selfplay_step(2)
    st = e.selfplay_stats()
    assert st["plies"] == 2 and st["games_finished"] == 0 and st["instant_wins"] == 0
    assert st["expansions"] == 2 * G * (rollouts + 1) == st["evals"]   # nothing is terminal at plies 2-3 of 5x5
    states = e.search_states()
    plies = states[:, 256 - 16 + 2].astype(int) | (states[:, 256 - 16 + 3].astype(int) << 8)
    assert (plies == 4).all()                                           # a1, far corner, two searched moves
    # stone conservation: reserves + stones on the board are constant
    heights = (states[:, 200:225] & 63).sum(1)
    reserves = states[:, 256 - 16 + 4 : 256 - 16 + 8].astype(int).sum(1)
    assert (heights + reserves == 44).all()
    # both opening corners occur (RNG purpose 1) and games differ (noise + sampling)
    assert len({s.tobytes() for s in states}) > G // 2


def test_sixteen_thousand_games_forward_search_and_selfplay(orc):
    """The width `extra.games_x4` of the bench line runs at — ONE engine with max_batch = games = 16 384 on the C2 network (other grids
    of k_tower_halo / k_fc_ring: 1024 workgroups, another XCD dealing; another pool geometry) — held to what the 4096-game tests
    hold: tg_policy_eval on all 16 384 DISTINCT rows against PyTorch fp32 (≤ 1e-4), 16 whole trees of a 24-iteration search bit for
    bit against the oracle, and two self-play plies with the counter and stone-conservation invariants.
    Reference semantics: train/src/self_play.rs:181-210 (one leaf per game into one batch), alpha-tak/src/model/network.rs:26-35."""
    import tak_amd

    net = torch_ref.make_net(5, 6, 64, "fc5", seed=0, randomize_bn=True)
    tensors = torch_ref.abi_tensors(net)
    e = tak_amd.Engine(5, res_blocks=6, filters=64, evaluator=tak_amd.EVAL_RESNET, max_batch=G4)
    e.load_state_dict(tensors)
    sts = _roots(orc, G4)
    assert len({s.tobytes() for s in sts}) == G4
    p, v = e.policy_eval(sts)
    assert np.abs(p.sum(1) - 1).max() < 1e-5 and (p > 0).all() and (np.abs(v) <= 1).all()
    worst_p = worst_v = 0.0
    for lo in range(0, G4, 1024):
        p_ref, v_ref = torch_ref.forward(net, orc.encode(5, sts[lo : lo + 1024]))
        worst_p = max(worst_p, float(np.abs(p[lo : lo + 1024] - p_ref).max()))
        worst_v = max(worst_v, float(np.abs(v[lo : lo + 1024] - v_ref).max()))
    print(f"c2_f32 at 16384 rows: worst |dp| {worst_p:.3e}, worst |dv| {worst_v:.3e} over all {G4} distinct rows")
    assert worst_p <= 1e-4 and worst_v <= 1e-4, (worst_p, worst_v)
    # the 4096-row batch of the same positions returns the same bits: a row does not know how wide its batch is
    p4, v4 = e.policy_eval(sts[5000 : 5000 + G])
    assert np.array_equal(p4, p[5000 : 5000 + G]) and np.array_equal(v4, v[5000 : 5000 + G])
    del p, p4
    # search: 16 384 trees, 24 lock-step iterations
    iters = 24
    e.search_create(G4, arena_nodes=1 << 12)
    e.search_reset(sts)
    e.search_run(iters)
    r = e.search_root()
    exp, ev = e.search_counters()
    assert exp == G4 * iters and ev <= exp
    assert (r["root_visits"] == iters).all()
    cs = np.array([r["visits"][g, : r["counts"][g]].sum() for g in range(G4)])
    assert (cs == iters - 1).all()
    assert np.array_equal(r["counts"], orc.movegen(5, sts)[1])
    assert (np.abs(r["q"]) <= 1.0 + 1e-6).all() and (np.abs(r["root_q"]) <= 1.0 + 1e-6).all()
    ev_eng = tak_amd.Engine(5, res_blocks=6, filters=64, evaluator=tak_amd.EVAL_RESNET, max_batch=64)
    ev_eng.load_state_dict(tensors)
    pick = np.arange(0, G4, G4 // 16) + np.arange(16) * 61  # spread over the workgroups AND over the slots inside one
    s = orc.Search(5, head=orc.HEAD_FC5, py_eval=lambda st: ev_eng.policy_eval(st))
    s.reset(sts[pick])
    s.run(iters)
    for k, g in enumerate(pick):
        a, b = e.search_dump(int(g)), s.dump(k)
        assert len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in a.dtype.names), g
        assert (a["virtual_visits"] == 0).all()
    ev_eng.close()
    # self-play: two plies of all 16 384 games
    rollouts = 16
    e.selfplay_create(G4, arena_nodes=1 << 12, seed=9, rollouts=rollouts, max_examples=G4 * 4)
    e.selfplay_step(2)
    st = e.selfplay_stats()
    assert st["plies"] == 2 and st["games_finished"] == 0 and st["instant_wins"] == 0 and st["aborted_games"] == 0 and st["dropped_examples"] == 0
    assert st["expansions"] == 2 * G4 * (rollouts + 1) == st["evals"]
    states = e.search_states()
    plies = states[:, 256 - 16 + 2].astype(int) | (states[:, 256 - 16 + 3].astype(int) << 8)
    assert (plies == 4).all()
    heights = (states[:, 200:225] & 63).sum(1)
    reserves = states[:, 256 - 16 + 4 : 256 - 16 + 8].astype(int).sum(1)
    assert (heights + reserves == 44).all()
    assert len({s.tobytes() for s in states}) > G4 // 4   # (two opening corners × two searched plies: a few thousand outcomes)
    # the shard property at this width: slots 4096 … 8191 of this engine are the games an engine with slot_base = 4096 plays
    e2 = tak_amd.Engine(5, res_blocks=6, filters=64, evaluator=tak_amd.EVAL_RESNET, max_batch=G)
    e2.load_state_dict(tensors)
    e2.selfplay_create(G, arena_nodes=1 << 12, seed=9, rollouts=rollouts, max_examples=G * 4, slot_base=G)
    e2.selfplay_step(2)
    assert np.array_equal(e2.search_states(), states[G : 2 * G])
    e2.close()
    e.close()


def test_board_pass_million_positions(orc):
    import tak_amd

    n = 5
    base = orc.random_positions(n, 6000, seed=3, max_plies=120, half_komi=4)
    base = base[orc.result(n, base) == 0][:4096]
    mv, cnt = orc.movegen(n, base)
    moves = mv[np.arange(len(base)), (np.arange(len(base)) * 7) % cnt]
    reps = (1 << 20) // len(base)
    e = tak_amd.Engine(n, evaluator=tak_amd.EVAL_DUMMY, max_batch=1024)
    ms, out_states, res, counts = e.board_pass_bench(np.tile(base, (reps, 1)), np.tile(moves, reps), reps=2)
    o_states, o_status = orc.play(n, base, moves)
    assert not o_status.any()
    # every copy of a position gives the same, oracle-exact answer wherever it sits in the 2^20 batch
    out_states = out_states.reshape(reps, len(base), -1)
    assert (out_states == o_states[None]).all()
    assert (res.reshape(reps, -1) == orc.result(n, o_states)[None]).all()
    ong = orc.result(n, o_states) == 0
    assert (counts.reshape(reps, -1)[:, ong] == orc.movegen(n, o_states)[1][ong][None]).all()
    e.close()


def test_training_chunk_full_size_properties(orc):
    """Config C5's network at the reference chunk size (500 examples × 8 symmetries = 4000 positions): one forward /
    backward / Adam step through size-independent properties."""
    import tak_amd

    n, blocks, filters = 5, 10, 128
    net = torch_ref.make_net(n, blocks, filters, "fc5", seed=0, randomize_bn=False)
    e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, evaluator=tak_amd.EVAL_RESNET, max_batch=64)
    e.load_state_dict(torch_ref.abi_tensors(net))
    e.train_create(chunk_size=500, chunks_in_step=1)
    sts = orc.random_positions(n, 1500, seed=33, max_plies=60, half_komi=4)
    sts = sts[orc.result(n, sts) == 0][:500]
    assert len(sts) == 500
    mv, cnt = orc.movegen(n, sts)
    rng = np.random.default_rng(0)
    visits = np.zeros((500, 512), np.uint32)
    for i in range(500):
        visits[i, : cnt[i]] = rng.integers(1, 40, cnt[i])
    results = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), 500)
    w0 = e.train_get_tensor("res9.conv2.weight", (filters, filters, 3, 3))
    lp, lz, stepped = e.train_chunk(sts, cnt.astype(np.int32), mv, visits, results)
    assert stepped and np.isfinite(lp) and np.isfinite(lz)
    # a random-init policy is not far from uniform over 1575 outputs: cross entropy of the order of ln 1575 = 7.36
    assert 6.5 < lp < 11.0 and 0.0 < lz < 2.5
    w1 = e.train_get_tensor("res9.conv2.weight", (filters, filters, 3, 3))
    d = np.abs(w1 - w0)
    assert 0 < d.max() <= 1.001e-4                      # first Adam step: |Δ| ≤ lr for every element
    assert np.abs(e.train_get_grad("policy.bias", (1575,))).max() == 0.0  # zero_grad after the step
    # the eight symmetric images of an example carry the same value target and the same visit mass: forward_training on
    # the augmented states is invariant under permuting the batch (BatchNorm statistics are order independent up to rounding)
    a_states, pi = e.augment_examples(sts[:64], cnt[:64].astype(np.int32), mv[:64], visits[:64])
    assert np.allclose(pi.sum(1), 1.0, atol=1e-5)
    lp1, v1 = e.train_forward(a_states)
    perm = rng.permutation(len(a_states))
    lp2, v2 = e.train_forward(a_states[perm])
    assert np.abs(lp1[perm] - lp2).max() < 1e-4 and np.abs(v1[perm] - v2).max() < 1e-4
    e.close()


def test_config_c3_full_size(orc):
    """BASELINE config C3 (6×6, 4096 concurrent games, 10-block × 128-filter net, conv-251 policy head) at full width:
    size-independent properties of the forward, the search and two plies of self-play, and a slice of trees bit for bit
    against the oracle fed by the same network kernels."""
    import tak_amd

    n, blocks, filters = 6, 10, 128
    net = torch_ref.make_net(n, blocks, filters, "conv", seed=0, randomize_bn=True)
    tensors = torch_ref.abi_tensors(net)
    e = tak_amd.Engine(n, res_blocks=blocks, filters=filters, evaluator=tak_amd.EVAL_RESNET, max_batch=G)
    e.load_state_dict(tensors)
    sts = posgen.distinct_positions(orc, n, G, seed=5, max_plies=70)
    p, v = e.policy_eval(sts)
    assert p.shape == (G, 9036) and np.abs(p.sum(1) - 1).max() < 2e-5 and (p > 0).all() and (np.abs(v) <= 1).all()
    # ALL 4096 rows of the full batch against PyTorch fp32 (round 4; it was every 4th row)
    idx = np.arange(0, G)
    worst_p = worst_v = 0.0
    for lo in range(0, len(idx), 128):
        sel = idx[lo : lo + 128]
        p_ref, v_ref = torch_ref.forward(net, orc.encode(n, sts[sel]))
        worst_p = max(worst_p, float(np.abs(p[sel] - p_ref).max()))
        worst_v = max(worst_v, float(np.abs(v[sel] - v_ref).max()))
    print(f"c3_f32 (randomised BatchNorm fold): worst |dp| {worst_p:.3e}, worst |dv| {worst_v:.3e} over all {G} rows")
    assert worst_p <= 1e-4 and worst_v <= 1e-4, (worst_p, worst_v)
    p2, v2 = e.policy_eval(np.roll(sts, 1000, axis=0)[:1024])              # the same positions in other slots of another batch size
    assert np.array_equal(p2[1000:], p[:24]) and np.array_equal(v2[1000:], v[:24])
    iters = 12
    e.search_create(G, arena_nodes=1 << 13)
    e.search_reset(sts)
    e.search_run(iters)
    r = e.search_root()
    exp, ev = e.search_counters()
    assert exp == G * iters and ev <= exp and (r["root_visits"] == iters).all()
    assert np.array_equal(r["counts"], orc.movegen(n, sts)[1])
    ev_eng = tak_amd.Engine(n, res_blocks=blocks, filters=filters, evaluator=tak_amd.EVAL_RESNET, max_batch=16)
    ev_eng.load_state_dict(tensors)
    pick = np.arange(0, G, G // 8)
    s = orc.Search(n, head=orc.HEAD_CONV, py_eval=lambda st: ev_eng.policy_eval(st))
    s.reset(sts[pick])
    s.run(iters)
    for k, g in enumerate(pick):
        a, b = e.search_dump(int(g)), s.dump(k)
        assert len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in a.dtype.names), g
    ev_eng.close()
    rollouts = 8
    e.selfplay_create(G, arena_nodes=1 << 13, seed=3, rollouts=rollouts, max_examples=G * 4)
    e.selfplay_step(2)
    st = e.selfplay_stats()
    assert st["plies"] == 2 and st["games_finished"] == 0
    assert st["expansions"] == 2 * G * (rollouts + 1) == st["evals"]
    states = e.search_states()
    hdr = 384 - 16
    plies = states[:, hdr + 2].astype(int) | (states[:, hdr + 3].astype(int) << 8)
    assert (plies == 4).all()
    heights = (states[:, 288:324] & 63).sum(1)
    reserves = states[:, hdr + 4 : hdr + 8].astype(int).sum(1)
    assert (heights + reserves == 62).all()
    e.close()
