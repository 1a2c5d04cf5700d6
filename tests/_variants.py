"""How far apart are the x86-64 Eigen builds the reference could have been compiled as?  (VERDICT r04, next #1.)

The oracle restates Eigen's arithmetic under three switches (oracle/ekf_oracle.hpp Config): the SSE2 quaternion product
and 4-float reduction order (`eigen_sse_quat`, default on), float sin/cos (`trig_float`, default off) and `vector /= scalar`
as a reciprocal multiply (`div_reciprocal`, default off).  This module runs the reference's own scenario inputs under all
eight combinations and reports the spread against the default, next to the fp32-vs-fp64 gap of the default that sizes the
parity tolerances.  Used by tests/test_oracle_variants_cpu.py (asserts) and scripts/oracle_variant_spread.py (the table in
profiles/ and DESIGN.md section 5).  TEST INFRASTRUCTURE: imports the oracle.
"""
import itertools

import numpy as np

from ekf_vio_amd.sim import Scenario
from oracle import OracleFilter

UV3 = [[0.1, 0.1], [-0.1, -0.1], [0.1, -0.1]]
VARIANTS = [dict(eigen_sse_quat=s, trig_float=t, div_reciprocal=d) for s, t, d in itertools.product((1, 0), (0, 1), (0, 1))]
DEFAULT = VARIANTS[0]


def tag(v):
    return "sse=%d trigf=%d recip=%d" % (v["eigen_sse_quat"], v["trig_float"], v["div_reciprocal"])


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0


def relf(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def jacobian_case(repeats, dt, variant, dtype=np.float32, generic=False):
    """test/jacobian_test.cpp:34-72: the three points repeated, omega_x = 3.1415, b_dx = 1.  generic = True: a rotated,
    moving, accelerating base state instead (the reference's test has q = identity, where most products of the
    quaternion multiplication are exact zeros and every summation order gives the same bits)."""
    uv = np.array(UV3 * repeats, np.float32)
    o = OracleFilter(dtype, emulate_static_cache=False, **variant)
    o.add_new_features(uv)
    st = o.get_state()
    st["base_mu"][10], st["base_mu"][7] = 3.1415, 1.0
    if generic:
        q = np.array([0.61, -0.37, 0.52, 0.47])
        st["base_mu"][3:7] = (q / np.linalg.norm(q)).astype(np.float32)
        st["base_mu"][0:3] = (0.3, -1.2, 0.7)
        st["base_mu"][7:10] = (0.4, -0.3, 0.9)
        st["base_mu"][10:13] = (0.31, -0.23, 0.52)
        st["base_mu"][13:16] = (0.8, 0.1, -0.6)
    o.set_state(st)
    F = o.linearize(dt)
    o.close()
    return F


def jacobian_spread(repeats, dt, generic=False):
    """max |F_variant - F_default| per variant, and the default's gap to its fp64 evaluation."""
    F0 = jacobian_case(repeats, dt, DEFAULT, generic=generic)
    F64 = jacobian_case(repeats, dt, DEFAULT, np.float64, generic=generic)
    out = {tag(v): maxabs(jacobian_case(repeats, dt, v, generic=generic), F0) for v in VARIANTS[1:]}
    return dict(n=F0.shape[0], fp32_vs_fp64=maxabs(F0, F64), spread=out)


def _state_gap(a, b):
    return dict(mu=maxabs(a["base_mu"], b["base_mu"]), feat=maxabs(a["feat_mu"], b["feat_mu"]), sig=relf(a["Sigma"], b["Sigma"]))


def step_case(N, variant, dtype, seed=5, warm=4):
    """One process(dt) + update at N landmarks (3 / 103 / 503: the sizes of test/test_ekf.cpp:66-141) from a state that
    `warm` fp64 filter steps of the default variant have made dense: every variant starts from the same fp32 state."""
    sc = Scenario(N, seed=seed)
    teacher = OracleFilter(np.float64, **DEFAULT)
    teacher.add_new_features(sc.initial_features())
    frames = list(sc.frames(warm + 1))
    for z, R, p in frames[:warm]:
        teacher.process(sc.dt), teacher.update(z, R, p)
    st = teacher.get_state()
    st32 = {k: (np.asarray(v, np.float32) if v.dtype.kind == "f" else v) for k, v in st.items()}
    o = OracleFilter(dtype, **variant)
    o.set_state(st32)
    z, R, p = frames[warm]
    o.process(sc.dt)
    after_process = o.get_state()
    o.update(z, R, p)
    out = o.get_state()
    o.close(), teacher.close()
    return after_process, out


def step_spread(N):
    p0, u0 = step_case(N, DEFAULT, np.float32)
    p64, u64 = step_case(N, DEFAULT, np.float64)
    rows = {}
    for v in VARIANTS[1:]:
        pv, uv_ = step_case(N, v, np.float32)
        rows[tag(v)] = dict(process=_state_gap(pv, p0), update=_state_gap(uv_, u0))
    return dict(N=N, fp32_vs_fp64=dict(process=_state_gap(p0, p64), update=_state_gap(u0, u64)), spread=rows)


def simulation_case(variant, dtype, N=30, steps=99, dt=0.05, seed=0, b_vel=(-0.1, 0.0, -0.1), omega=(0.0, 0.1, 0.0)):
    """test/analyzeEKFSimulation.cpp:233-244 (scenario :244: v = (-0.1, 0, -0.1), omega = (0, 0.1, 0), N = 30, dt = 0.05,
    99 steps), free-running from the raw prior."""
    sc = Scenario(N, seed=seed, b_vel=b_vel, omega=omega, dt=dt)
    o = OracleFilter(dtype, **variant)
    o.add_new_features(sc.initial_features())
    flagged = 0
    for z, R, p in sc.frames(steps):
        o.process(sc.dt)
        flagged += 1 if o.update(z, R, p) else 0
    st = o.get_state()
    o.close()
    return st, flagged, sc.pos.copy()


def simulation_spread(**kw):
    s0, f0, truth = simulation_case(DEFAULT, np.float32, **kw)
    s64, f64, _ = simulation_case(DEFAULT, np.float64, **kw)
    rows = {}
    for v in VARIANTS[1:]:
        sv, fv, _ = simulation_case(v, np.float32, **kw)
        rows[tag(v)] = dict(_state_gap(sv, s0), flagged=fv, pos_err=maxabs(sv["base_mu"][:3], truth))
    return dict(fp32_vs_fp64=dict(_state_gap(s0, s64), flagged=f0, flagged64=f64, pos_err=maxabs(s0["base_mu"][:3], truth),
                                  pos_err64=maxabs(s64["base_mu"][:3], truth)), spread=rows)
