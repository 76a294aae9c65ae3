"""Symmetry / Example::to_tensors (tak/src/symm.rs, alpha-tak/src/example.rs:62-78): oracle properties on CPU,
GPU kernel vs oracle on the GPU."""
import numpy as np
import pytest


def _examples(orc, n, count, seed):
    sts = orc.random_positions(n, count * 2, seed=seed, max_plies=60 if n >= 5 else 16, half_komi=4)
    sts = sts[orc.result(n, sts) == 0][:count]
    mv, cnt = orc.movegen(n, sts)
    rng = np.random.default_rng(seed)
    visits = np.zeros((len(sts), 512), np.uint32)
    for i in range(len(sts)):
        visits[i, : cnt[i]] = rng.integers(0, 50, cnt[i])
        visits[i, rng.integers(cnt[i])] += 1
    return sts, cnt.astype(np.int32), mv, visits


@pytest.mark.parametrize("n", [5, 6])
def test_oracle_symmetry_properties(orc, n):
    head = orc.HEAD_FC5 if n == 5 else orc.HEAD_CONV
    sts, cnt, mv, visits = _examples(orc, n, 40, seed=n)
    out, pi = orc.augment(n, head, sts, cnt, mv, visits)
    out = out.reshape(len(sts), 8, -1)
    pi = pi.reshape(len(sts), 8, -1)
    planes = orc.encode(n, out.reshape(-1, out.shape[-1])).reshape(len(sts), 8, -1, n, n)
    for i in range(len(sts)):
        assert np.array_equal(out[i, 0], sts[i])                       # identity first (symm.rs:11)
        assert np.allclose(pi[i].sum(1), 1.0, atol=1e-5)               # every image keeps all the visit mass
        # the 8 images are exactly the dihedral orbit of the encoded planes (independent of takparse's orientation)
        orbit = []
        for k in range(4):
            r = np.rot90(planes[i, 0], k, axes=(1, 2))
            orbit += [r.tobytes(), r[:, :, ::-1].tobytes()]
        assert sorted(p.tobytes() for p in planes[i]) == sorted(orbit)
        # transformed moves are exactly the legal moves of the transformed game, with the same visit multiset
        for k in range(8):
            lm, lc = orc.movegen(n, out[i, k])
            idx = orc.move_index(n, lm[0, : lc[0]])
            assert lc[0] == cnt[i]
            nz = np.nonzero(pi[i, k])[0]
            assert set(nz) <= set(idx.tolist())
            assert sorted(np.round(pi[i, k][nz] * visits[i].sum()).astype(int)) == sorted(v for v in visits[i, : cnt[i]] if v)
        # results agree within the rotation chains (tak/tests/symm.rs:18-24)
        res = orc.result(n, out[i])
        assert len(set(res[:4])) == 1 and len(set(res[4:])) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("n,conv5", [(4, False), (5, False), (6, False), (5, True)])
def test_gpu_augment_matches_oracle(orc, n, conv5):
    import tak_amd

    # conv5: a 5×5 network with the conv head (3075 outputs) — the policy targets are indexed by the conv formula, not the legacy table
    head = orc.HEAD_FC5 if n == 5 and not conv5 else orc.HEAD_CONV
    e = tak_amd.Engine(n, evaluator=tak_amd.EVAL_DUMMY, max_batch=512,
                       policy_head=tak_amd.HEAD_FC5 if n == 5 and not conv5 else tak_amd.HEAD_CONV)
    sts, cnt, mv, visits = _examples(orc, n, 1500, seed=10 + n)   # crosses the 1024-example chunk
    g_states, g_pi = e.augment_examples(sts, cnt, mv, visits)
    o_states, o_pi = orc.augment(n, head, sts, cnt, mv, visits)
    assert np.array_equal(g_states, o_states)
    assert np.array_equal(g_pi.view(np.uint32), o_pi.view(np.uint32))
    e.close()
