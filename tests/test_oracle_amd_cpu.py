"""The fill-reducing ordering of the reference's solver (VERDICT r05 next #4a) and what its NumericalIssue means (#4b).

`Eigen::SimplicialLDLT<SparseMatrix<float>> solver; solver.compute(S.transpose())` (TightlyCoupledEKF.cpp:577-578) orders with
Eigen's default AMDOrdering before it factors.  Eigen is absent here (third-party, unpinned upstream), so oracle/amd_order.hpp restates
the published algorithm (CSparse cs_amd, which Eigen's Amd.h ports) and the oracle's update factors S^T(P, P) with Eigen's up-looking
sparse loop.  Pinned by construction, not against Eigen: parity unpinned.  What these tests establish:
  * the routine returns a permutation, puts hubs last and dense rows last, and leaves complete graphs and R's 2 x 2 block pattern in
    natural order -- so for the S of this filter (block-diagonal straight from the diagonal prior, numerically dense afterwards) the
    ordering is the identity in every scenario the reference's own tests run, and the HIP path, which factors in natural order, follows it;
  * a pattern that does permute (built by hand: an early row coupled to many later, mutually uncoupled ones) gives a result that differs
    from the natural-order one by rounding, well inside the fp32|fp64 yardstick;
  * the sparse-form factorisation loop gives the dense loop's bits on a dense pattern;
  * NumericalIssue = a pivot that is exactly zero, nothing else."""
import numpy as np
import pytest

import _variants as V
from ekf_vio_amd.sim import Scenario
from oracle import OracleFilter, amd_order


def _fill(pattern, perm):
    """Number of fill entries of a symbolic elimination of `pattern` in the order `perm`."""
    a = np.array(pattern, bool)[np.ix_(perm, perm)]
    a = a | a.T
    n, fill = a.shape[0], 0
    for k in range(n):
        nb = np.nonzero(a[k + 1:, k])[0] + k + 1
        for i in nb:
            new = ~a[nb, i]
            new[nb == i] = False
            fill += int(new.sum())
            a[nb, i] = True
            a[i, nb] = True
    return fill


@pytest.mark.parametrize("keep_diagonal", [True, False])
def test_amd_order_is_a_permutation_that_never_fills_more_than_the_hub_first_order(keep_diagonal):
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 7, 16, 33, 64, 90):
        for density in (0.0, 0.05, 0.3, 1.0):
            a = rng.random((n, n)) < density
            a = a | a.T | np.eye(n, dtype=bool)
            p = amd_order(a, keep_diagonal)
            assert sorted(p.tolist()) == list(range(n)), (n, density, p)
    # an arrow matrix with its hub FIRST fills completely in natural order and not at all behind the ordering (hub last)
    n = 24
    a = np.eye(n, dtype=bool)
    a[0, :] = a[:, 0] = True
    p = amd_order(a, keep_diagonal)
    assert p[-1] == 0 and _fill(a, p) == 0 and _fill(a, np.arange(n)) == (n - 1) * (n - 2) // 2
    # a star's leaves come before its centre, a path is eliminated without fill
    star = np.eye(9, dtype=bool)
    star[4, :] = star[:, 4] = True
    assert amd_order(star, keep_diagonal)[-1] == 4
    path = np.eye(12, dtype=bool) | np.eye(12, k=1, dtype=bool) | np.eye(12, k=-1, dtype=bool)
    assert _fill(path, amd_order(path, keep_diagonal)) == 0


@pytest.mark.parametrize("keep_diagonal", [True, False])
def test_the_patterns_this_filter_produces_keep_their_natural_order(keep_diagonal):
    """Complete graphs of any size (below the dense-row threshold the first pivot absorbs every other node by mass elimination and the
    post-order lists them ascending; above it every node is 'dense' and ordered last, ascending), and the block-diagonal pattern of
    2 x 2 blocks that S has in an update straight from the diagonal prior (test/test_ekf.cpp:66-141)."""
    for n in (2, 4, 16, 17, 18, 60, 100, 101, 102, 103, 104, 204, 512):
        assert np.array_equal(amd_order(np.ones((n, n), bool), keep_diagonal), np.arange(n)), n
    for blocks in (1, 2, 3, 51, 102, 503):
        a = np.kron(np.eye(blocks, dtype=bool), np.ones((2, 2), bool))
        assert np.array_equal(amd_order(a, keep_diagonal), np.arange(2 * blocks)), blocks


def test_dense_rows_are_ordered_last():
    """Rows with more than max(16, 10 sqrt(n)) entries are not eliminated by degree but appended in index order (cs_amd's `dense`)."""
    n = 400  # threshold 200
    a = np.eye(n, dtype=bool)
    for hub in (7, 123):
        a[hub, :] = a[:, hub] = True
    p = amd_order(a)
    assert p[-2:].tolist() == [7, 123]


def _reference_scenario_3(o):
    """test/test_ekf.cpp:66-82: three landmarks, update straight from the diagonal prior with the middle one unmeasured."""
    o.add_new_features(np.array(V.UV3, np.float32))
    z = np.array(V.UV3, np.float32)
    R = np.tile(np.array([1e-3, 0, 0, 1e-3], np.float32), (3, 1))
    return z, R, np.array([1, 0, 1], np.uint8)


def test_the_references_own_update_scenarios_factor_in_natural_order():
    o = OracleFilter(np.float32)
    z, R, p = _reference_scenario_3(o)
    o.update(z, R, p)
    assert np.array_equal(o.last_perm(), np.arange(4))
    o.update(z, R, p)  # (:80: the second call, on the updated -- now coupled -- state)
    assert np.array_equal(o.last_perm(), np.arange(4))
    o.close()
    for N in (30, 103):
        sc = Scenario(N, seed=1)
        o = OracleFilter(np.float32)
        o.add_new_features(sc.initial_features())
        nat = 0
        for z, R, p in sc.frames(6):
            o.process(sc.dt)
            o.update(z, R, p)
            nat += int(np.array_equal(o.last_perm(), np.arange(2 * N)))
        assert nat == 6, (N, nat)  # S is numerically dense once process(dt) has run: the identity every time
        o.close()


def test_sparse_form_factorisation_gives_the_dense_loops_bits():
    sc = Scenario(40, seed=2)
    frames = list(sc.frames(3))  # (once: the scenario's truth advances with every call)
    outs = []
    for general in (0, 1):
        o = OracleFilter(np.float32)
        o.set_ldlt_order(general_path=general)
        o.add_new_features(sc.initial_features())
        for z, R, p in frames:
            o.process(sc.dt)
            o.update(z, R, p)
        outs.append(o.get_state())
        o.close()
    for k in ("base_mu", "feat_mu", "Sigma"):
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_a_pattern_with_a_leading_hub_is_reordered_and_stays_inside_the_yardstick():
    """Block patterns come back in natural order (the post-order walks the assembly tree's roots by index): fresh landmarks appended
    behind a dense block, the usual source of structure in this filter (addNewFeatures leaves their cross-covariances zero, :58-94),
    do NOT permute.  A real permutation needs an early row that couples to many later, mutually uncoupled ones -- built here by hand:
    landmark 0's u coordinate correlated with every other landmark's, nothing else.  The hub then moves back; the result and the
    natural-order one solve the same system and differ by rounding, less than the fp32 result differs from the fp64 one."""
    N = 10
    sc = Scenario(N, seed=7)
    uv = sc.initial_features()
    z, R, p = list(sc.frames(1))[0]
    # fresh landmarks behind a dense block: natural order
    o = OracleFilter(np.float32)
    o.add_new_features(uv[:6])
    o.process(sc.dt)
    o.update(z[:6], R[:6], p[:6])
    o.add_new_features(uv[6:])
    o.update(z, R, p)
    assert np.array_equal(o.last_perm(), np.arange(2 * N))
    o.close()
    res = {}
    for name, kw in (("amd", {}), ("natural", dict(ldlt_amd_order=0)), ("fp64", {})):
        o = OracleFilter(np.float64 if name == "fp64" else np.float32, **kw)
        o.add_new_features(uv)
        st = o.get_state()
        sig = st["Sigma"].copy()
        for i in range(1, N):
            sig[22, 22 + 3 * i] = sig[22 + 3 * i, 22] = 5e-7
        o.set_state({**st, "Sigma": sig})
        assert o.update(z, R, p) == 0
        res[name] = (o.get_state(), o.last_perm().copy())
        o.close()
    m = 2 * N
    assert sorted(res["amd"][1].tolist()) == list(range(m))
    assert not np.array_equal(res["amd"][1], np.arange(m))  # the hub row moves behind the leaves it couples
    assert np.array_equal(res["natural"][1], np.arange(m))
    a, b, c = res["amd"][0], res["natural"][0], res["fp64"][0]
    assert not np.array_equal(a["Sigma"], b["Sigma"])
    for k in ("base_mu", "feat_mu"):
        assert V.maxabs(a[k], b[k]) <= V.maxabs(b[k], c[k]) + 2e-6, k
    assert V.relf(a["Sigma"], b["Sigma"]) <= V.relf(b["Sigma"], c["Sigma"]) + 2e-5


def test_numerical_issue_means_an_exactly_zero_pivot():
    """Eigen's simplicial LDL^T sets NumericalIssue only for `d == 0` (and then abandons the factorisation): bit 1 of the oracle's status.
    A NEGATIVE pivot passes silently upstream: bit 0 alone (what the HIP path reports as EKFVIO_ENUMERIC, the stricter of the two)."""
    o = OracleFilter(np.float32)
    z, R, p = _reference_scenario_3(o)
    st = o.get_state()
    assert o.update(z, R, p) == 0
    # S = 0 exactly: zero prior on the measured landmarks' coordinates and zero measurement noise
    o.set_state({**st, "Sigma": np.zeros_like(st["Sigma"])})
    with np.errstate(all="ignore"):
        info = o.update(z, np.zeros_like(R), p)
    assert info & 2 and info & 1, info
    # an indefinite S: a negative variance on a measured coordinate -> a negative pivot, no exact zero
    sig = st["Sigma"].copy()
    sig[22, 22] = -1.0
    o.set_state({**st, "Sigma": sig})
    info = o.update(z, R, p)
    assert info == 1, info
    o.close()
