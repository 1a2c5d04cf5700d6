"""Independent numpy fp64 restatement of the reference filter — TEST INFRASTRUCTURE ONLY.

Written separately from ekf_oracle.hpp (matrix form, rotation matrices instead of the
quaternion sandwich, plain numpy products) so that a misreading of
include/ekf_vio/TightlyCoupledEKF.cpp in one of the two shows up as a disagreement.
Follows: process :96-121, Q :123-174, FD Jacobian :176-325, motion models :328-460,
update :475-628, H map :634-661.
"""
import numpy as np

BASE = 22
DELTA = 1e-3


def _rotmat(q):
    """Rotation matrix of Eigen's q*v formula (valid also for non-unit q: v + 2w(qv x v) + 2 qv x (qv x v))."""
    w, x, y, z = q
    K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]], dtype=np.float64)
    return np.eye(3) + 2 * w * K + 2 * K @ K


def _qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx])


def _dq(omega, dt):
    on = np.linalg.norm(omega)
    if on < 1e-10:
        q = np.array([1.0, *(omega * dt)])
        return q / np.linalg.norm(q)
    th = dt * on
    return np.array([np.cos(th / 2), *(omega / on * np.sin(th / 2))])


def _qinv(q):
    return np.array([q[0], -q[1], -q[2], -q[3]]) / np.dot(q, q)


def convolve_base_state(mu, dt):
    mu = np.asarray(mu, dtype=np.float64)
    pos, quat, vel, om, acc = mu[0:3], mu[3:7], mu[7:10], mu[10:13], mu[13:16]
    out = mu.copy()
    out[0:3] = pos + _rotmat(quat) @ (dt * vel + 0.5 * dt * dt * acc)
    dq = _dq(om, dt)
    Rinv = _rotmat(_qinv(dq))
    out[7:10] = Rinv @ (vel + dt * acc)
    out[13:16] = Rinv @ acc
    out[3:7] = _qmul(quat, dq)
    return out


def convolve_feature(base, feat, dt):
    base = np.asarray(base, dtype=np.float64)
    u, v, rho = np.asarray(feat, dtype=np.float64)
    p = np.array([u / rho, v / rho, 1.0 / rho])
    t = dt * base[7:10] + 0.5 * dt * dt * base[13:16]
    dq = _dq(base[10:13], dt)
    Rinv = _rotmat(np.array([dq[0], -dq[1], -dq[2], -dq[3]]))  # reference builds dq_inv as the conjugate (:431,:439)
    p2 = Rinv @ p - Rinv @ t
    return np.array([p2[0] / p2[2], p2[1] / p2[2], 1.0 / p2[2]])


class NpFilter:
    def __init__(self, depth=0.5, depth_var=100.0, homog_var=1e-5):
        self.depth, self.depth_var, self.homog_var = depth, depth_var, homog_var
        self.base_mu = np.zeros(BASE)
        self.base_mu[3] = 1.0
        self.feat = np.zeros((0, 3))
        self.last_klt = np.zeros((0, 2))
        self.del_flag = np.zeros(0, dtype=np.uint8)
        d = np.zeros(BASE)
        d[7:16] = 30.0
        d[16:22] = 0.5
        self.Sigma = np.diag(d)

    @property
    def n(self):
        return BASE + 3 * len(self.feat)

    def add_new_features(self, uv):
        uv = np.asarray(uv, dtype=np.float64).reshape(-1, 2)
        k = len(uv)
        if k == 0:
            return
        n0 = self.n
        S = np.zeros((n0 + 3 * k, n0 + 3 * k))
        S[:n0, :n0] = self.Sigma
        for f in range(k):
            S[n0 + 3 * f, n0 + 3 * f] = self.homog_var
            S[n0 + 3 * f + 1, n0 + 3 * f + 1] = self.homog_var
            S[n0 + 3 * f + 2, n0 + 3 * f + 2] = self.depth_var
        self.Sigma = S
        self.feat = np.vstack([self.feat, np.column_stack([uv, np.full(k, 1.0 / self.depth)])])
        self.last_klt = np.vstack([self.last_klt, uv])
        self.del_flag = np.concatenate([self.del_flag, np.zeros(k, np.uint8)])

    def process_noise_diag(self, dt):
        q = np.full(self.n, 1e-4 * dt)
        q[7:10] = 0.01 * dt
        q[10:16] = 5 * dt
        q[16:22] = 1e-3 * dt
        return q

    def linearize(self, dt):
        n, N = self.n, len(self.feat)
        F = np.zeros((n, n))
        for j in range(BASE):
            if j <= 15:
                hi, lo = self.base_mu.copy(), self.base_mu.copy()
                hi[j] += DELTA
                lo[j] -= DELTA
                F[:BASE, j] = (convolve_base_state(hi, dt) - convolve_base_state(lo, dt)) / (2 * DELTA)
                if j >= 7:
                    for f in range(N):
                        F[BASE + 3 * f:BASE + 3 * f + 3, j] = (
                            convolve_feature(hi, self.feat[f], dt) - convolve_feature(lo, self.feat[f], dt)) / (2 * DELTA)
            else:
                F[j, j] = 1.0
        for f in range(N):
            for c in range(3):
                hi, lo = self.feat[f].copy(), self.feat[f].copy()
                hi[c] += DELTA
                lo[c] -= DELTA
                F[BASE + 3 * f:BASE + 3 * f + 3, BASE + 3 * f + c] = (
                    convolve_feature(self.base_mu, hi, dt) - convolve_feature(self.base_mu, lo, dt)) / (2 * DELTA)
        return F

    def process(self, dt):
        F = self.linearize(dt)
        self.feat = np.array([convolve_feature(self.base_mu, f, dt) for f in self.feat]).reshape(-1, 3)
        self.base_mu = convolve_base_state(self.base_mu, dt)
        self.Sigma = F @ self.Sigma @ F.T + np.diag(self.process_noise_diag(dt))

    def measurement_map(self, measured):
        idx = []
        for i, mflag in enumerate(measured):
            if mflag:
                idx += [BASE + 3 * i, BASE + 3 * i + 1]
        return np.array(idx, dtype=np.int64)

    def update(self, z, R, passed):
        N, n = len(self.feat), self.n
        z = np.asarray(z, dtype=np.float64).reshape(N, 2)
        R = np.asarray(R, dtype=np.float64).reshape(N, 4)
        passed = np.asarray(passed).astype(bool)
        idx = self.measurement_map(passed)
        m = len(idx)
        H = np.zeros((m, n))
        H[np.arange(m), idx] = 1.0
        Rm = np.zeros((m, m))
        zz = np.zeros(m)
        j = 0
        for i in range(N):
            if passed[i]:
                self.last_klt[i] = z[i]
                zz[j:j + 2] = z[i]
                Rm[j:j + 2, j:j + 2] = R[i].reshape(2, 2).T  # column-major 2x2
                j += 2
            else:
                self.del_flag[i] = 1
        mu = np.concatenate([self.base_mu, self.feat.reshape(-1)])
        if m > 0:
            y = zz - H @ mu
            S = H @ self.Sigma @ H.T + Rm
            K = np.linalg.solve(S.T, (self.Sigma @ H.T).T).T
            IKH = np.eye(n) - K @ H
            self.Sigma = IKH @ self.Sigma @ IKH.T + K @ Rm @ K.T
            mu = mu + K @ y
        mu[3:7] /= np.linalg.norm(mu[3:7])
        self.base_mu = mu[:BASE].copy()
        self.feat = mu[BASE:].reshape(N, 3).copy()
