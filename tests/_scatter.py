"""How far the reference's fp32 arithmetic itself scatters on one update step.

The first update from the raw prior is ill-conditioned (cond(S) ~ 1e6): the fp32 result moves
by ~1e-3 in the base state when nothing but the ORDER of the landmarks changes (same algorithm,
same data, mathematically identical answer).  A single fp32 run of the oracle can land close to
the fp64 answer by luck, so a bound "k x the oracle's error" is only meaningful against the worst
of several equivalent orderings.  This module measures that: the fp32 oracle is run on landmark
permutations of the same state and the worst forward error against the fp64 answer is returned.
"""
import numpy as np

from oracle import OracleFilter

BASE = 22


def _perm_state(st, perm):
    n = BASE + 3 * len(perm)
    idx = np.concatenate([np.arange(BASE)] + [BASE + 3 * q + np.arange(3) for q in perm])
    assert idx.shape[0] == n
    return dict(base_mu=st["base_mu"].copy(), feat_mu=st["feat_mu"][perm].copy(), last_klt=st["last_klt"][perm].copy(),
                del_flag=st["del_flag"][perm].copy(), Sigma=st["Sigma"][np.ix_(idx, idx)].copy()), idx


def fp32_scatter(st, z, R, passed, s64, nperm=6, seed=0):
    """st: state before the update (fp32 arrays); s64: fp64 oracle state after the update.
    Returns the worst |fp32 - fp64| over `nperm` landmark orderings (the first is the identity):
    dict(mu=base state max-abs, feat=landmark means max-abs, sig=relative Frobenius)."""
    N = st["feat_mu"].shape[0]
    rng = np.random.default_rng(seed)
    worst = dict(mu=0.0, feat=0.0, sig=0.0)
    for k in range(nperm):
        perm = np.arange(N) if k == 0 else rng.permutation(N)
        sp, idx = _perm_state(st, perm)
        o = OracleFilter(np.float32)
        o.set_state(sp)
        o.update(np.asarray(z)[perm], np.asarray(R)[perm], np.asarray(passed)[perm])
        out = o.get_state()
        inv = np.empty_like(perm)
        inv[perm] = np.arange(N)
        iidx = np.empty_like(idx)
        iidx[idx] = np.arange(idx.shape[0])
        feat = out["feat_mu"][inv].astype(np.float64)
        sig = out["Sigma"][np.ix_(iidx, iidx)].astype(np.float64)
        worst["mu"] = max(worst["mu"], float(np.abs(out["base_mu"].astype(np.float64) - s64["base_mu"]).max()))
        worst["feat"] = max(worst["feat"], float(np.abs(feat - s64["feat_mu"]).max()))
        worst["sig"] = max(worst["sig"], float(np.linalg.norm(sig - s64["Sigma"]) / np.linalg.norm(s64["Sigma"])))
    return worst


def backward_yardstick(st, z, R, passed, s64, c=8.0, trials=8, seed=0):
    """Sensitivity of the exact (fp64) update to fp32-level input noise.

    A backward-stable fp32 solve returns the exact gain of a problem whose S = H Sigma H^T + R and
    Sigma H^T are perturbed by a few units of fp32 round-off.  This evaluates the update in fp64
    with S and Sigma H^T multiplied componentwise by (1 + c * 2^-24 * u), u uniform in [-1, 1]
    (S kept symmetric), the rest exact (Joseph form, TightlyCoupledEKF.cpp:586-612), and returns the
    worst change of the result against the unperturbed fp64 state `s64` over `trials` draws:
    "an algorithm with componentwise backward error <= c ulp may be this far off".  For the first
    update from the raw prior (cond(S) ~ 1e6) one ulp already moves the base state by ~2e-4.
    """
    N = st["feat_mu"].shape[0]
    passed = np.asarray(passed).astype(bool)
    idx = np.array([BASE + 3 * i + a for i in range(N) if passed[i] for a in (0, 1)], dtype=np.int64)
    if idx.size == 0:
        return dict(mu=0.0, feat=0.0, sig=0.0)
    m = idx.size
    P = st["Sigma"].astype(np.float64)
    mu = np.concatenate([st["base_mu"], st["feat_mu"].reshape(-1)]).astype(np.float64)
    Rm = np.zeros((m, m))
    r = 0
    for i in range(N):
        if passed[i]:
            Rm[r:r + 2, r:r + 2] = np.asarray(R[i], np.float64).reshape(2, 2)
            r += 2
    zz = np.array([z[i][a] for i in range(N) if passed[i] for a in (0, 1)], np.float64)
    y = zz - mu[idx]
    S = P[np.ix_(idx, idx)] + Rm
    A = np.triu(S) + np.triu(S, 1).T          # the triangle the reference factors: lower(S^T) = upper(S)
    Cm = P[:, idx]
    W = P[idx, :]
    rng = np.random.default_rng(seed)
    u24 = 2.0 ** -24
    worst = dict(mu=0.0, feat=0.0, sig=0.0)
    for _ in range(trials):
        E = rng.uniform(-1, 1, A.shape)
        E = np.triu(E) + np.triu(E, 1).T
        Ap = A * (1 + c * u24 * E)
        Cp = Cm * (1 + c * u24 * rng.uniform(-1, 1, Cm.shape))
        Kg = np.linalg.solve(Ap, Cp.T).T
        mun = mu + Kg @ y
        mun[3:7] /= np.linalg.norm(mun[3:7])
        T = P - Kg @ W
        Pn = T - T[:, idx] @ Kg.T + Kg @ Rm @ Kg.T
        worst["mu"] = max(worst["mu"], float(np.abs(mun[:BASE] - s64["base_mu"]).max()))
        worst["feat"] = max(worst["feat"], float(np.abs(mun[BASE:].reshape(N, 3) - s64["feat_mu"]).max()))
        worst["sig"] = max(worst["sig"], float(np.linalg.norm(Pn - s64["Sigma"]) / np.linalg.norm(s64["Sigma"])))
    return worst
