"""CPU: the oracle restatement (oracle/rt_oracle.cpp) against fixtures produced by the compiled reference.

Everything here is bit-exact: the oracle reproduces the reference's hit records and, with std::rand in the
reference's draw order, its whole rendered frames (double pixel sums and write_color ints)."""
import numpy as np
import pytest

from tests import fixtures as fx
from tests import oracle_lib as O


def _bytes_equal(a, b):
    return np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 6])  # 6 = cornell smoke: constant_mediums bounded by boxes, instances, a moving sphere
def test_trace_batch_bit_exact(n):
    sc = fx.scene(n)
    rays, xi, hits = fx.trace(n)
    got, _ = O.trace_batch(sc, rays, xi)
    assert (got["prim"] == hits["prim"]).all()
    for f in ("t", "p", "normal", "u", "v", "front_face"):
        assert _bytes_equal(got[f], hits[f]), f
    assert (hits["prim"] >= 0).sum() > 1000
    if n == 6:
        from raytracing_amd import abi
        assert (sc.nodes["kind"][hits["prim"][hits["prim"] >= 0]] == abi.RT_NODE_MEDIUM).sum() > 200


def test_box_corner_edge_batch_bit_exact():
    """final_scene's ground boxes hit exactly through corners and edges (equal-distance candidates: the winner is decided
    by the reference's visiting order and its strict / inclusive interval tests) and next to them."""
    sc = fx.scene(3)
    rays, xi, hits = fx.boxtrace(3)
    got, _ = O.trace_batch(sc, rays, xi)
    assert (got["prim"] == hits["prim"]).all()
    for f in ("t", "p", "normal", "u", "v", "front_face"):
        assert _bytes_equal(got[f], hits[f]), f
    assert len(rays) == 6700 and (hits["prim"] >= 0).sum() > 6000


def test_f32_bounds_module_agrees_with_the_reference_hits():
    """tests/f32_bounds.py re-derives acceptance margins in double precision: on the reference's own hits it must accept
    the reference's primitive at the reference's distance (otherwise its 'explanations' of float mismatches mean nothing)."""
    import math
    from tests import f32_bounds as fb
    for n in (1, 2, 3, 4):
        sc = fx.scene(n)
        rays, xi, hits = fx.trace(n)
        chains = fb.transform_chains(sc)
        idx = np.nonzero(hits["prim"] >= 0)[0][::5]
        for k in idx:
            r = int(hits["prim"][k])
            ok, t, m = fb.margin(sc, r, chains[r], rays[k], math.inf, float(xi[k]))
            assert ok and m >= 0, (n, k)
            assert abs(t - hits["t"][k]) <= 2e-7 * abs(hits["t"][k]), (n, k)  # triangles: the reference rounds t to float
        rows = fb.classify(sc, rays, hits, hits, xi)
        assert all(r["status"] == "same" and r["rel_t"] == 0 and r["dn"] == 0 for r in rows)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 6])
def test_whole_frame_bit_exact(n):
    sc = fx.scene(n)
    cam, seed, sums, by = fx.img(n)
    O.srand(seed)
    acc, cnt = O.render(sc, cam, rng_mode=0)
    assert _bytes_equal(acc, sums)
    k = int(cam["sqrt_spp"][0])
    assert (O.write_color(acc, 1.0 / (k * k)) == by).all()
    assert cnt[5] == sums.shape[0] * sums.shape[1] * k * k


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert [hex(x) for x in O.philox([0, 0, 0, 0], [0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in O.philox([0xffffffff] * 4, [0xffffffff] * 2)] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6",
                                                                             "0x6d5451fd"]
    assert [hex(x) for x in O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == [
        "0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_philox_render_threads_agree():
    sc = fx.scene(2)
    cam, _, _, _ = fx.img(2)
    a1, _ = O.render(sc, cam, seed=5, rng_mode=1, threads=1)
    a4, _ = O.render(sc, cam, seed=5, rng_mode=1, threads=4)
    assert _bytes_equal(a1, a4)
    # sample-range split sums to the whole (what the multi-GPU spp split relies on)
    lo, _ = O.render(sc, cam, seed=5, s_begin=0, s_end=4, rng_mode=1)
    hi, _ = O.render(sc, cam, seed=5, s_begin=4, s_end=9, rng_mode=1)
    np.testing.assert_allclose(lo + hi, a1, rtol=1e-12, atol=1e-12)
