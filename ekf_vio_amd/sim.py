"""Synthetic moving-camera scenario generator (ROS-free).

Follows the structure of the reference's closed-loop simulation
(test/analyzeEKFSimulation.cpp:10-125): N landmarks with z = depth_mu + N(0, depth_sigma),
x,y = U(-1.5,1.5)*z; ground-truth pose integrated with the filter's own motion model;
perfect projections (q^-1 p - q^-1 pos -> x/z, y/z) as measurements with R = 1e-5*I and
every landmark measured.  The reference seeds cv::RNG(0); OpenCV is not available here, so
the generator uses its own SplitMix64 -> uniform / Box-Muller stream (seeded), which is
part of the workload definition.
"""
import numpy as np

_MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _MASK

    def next_u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * ((self.next_u64() >> 11) * (1.0 / (1 << 53)))

    def gaussian(self, sigma=1.0):
        u1 = max(self.uniform(), 1e-300)
        u2 = self.uniform()
        return sigma * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _rotate(q, v):
    w, x, y, z = q
    qv = np.array([x, y, z])
    uv = 2.0 * np.cross(qv, v)
    return v + w * uv + np.cross(qv, uv)


def _qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx])


def _conj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


class Scenario:
    """Landmarks + ground-truth trajectory; yields per-frame measurements.

    Defaults are config 2 of BASELINE.json / SURVEY.md 8(d): b_vel=(-0.1,0,-0.1),
    omega=(0,0.1,0), accel=0, dt=1/30, depth 0.5 +- 0.01, seed 0.
    """

    def __init__(self, n_landmarks=256, seed=0, depth_mu=0.5, depth_sigma=0.01, b_vel=(-0.1, 0.0, -0.1),
                 b_accel=(0.0, 0.0, 0.0), omega=(0.0, 0.1, 0.0), dt=1.0 / 30.0, meas_var=1e-5):
        rng = SplitMix64(seed)
        pts = np.zeros((n_landmarks, 3))
        for i in range(n_landmarks):
            z = depth_mu + rng.gaussian(depth_sigma)
            pts[i] = (rng.uniform(-1.5, 1.5) * z, rng.uniform(-1.5, 1.5) * z, z)
        self.points = pts
        self.N = n_landmarks
        self.dt = float(np.float32(dt))
        self.pos = np.zeros(3)
        self.quat = np.array([1.0, 0.0, 0.0, 0.0])
        self.vel = np.array(b_vel, dtype=np.float64)
        self.acc = np.array(b_accel, dtype=np.float64)
        self.omega = np.array(omega, dtype=np.float64)
        self.meas_var = meas_var

    def initial_features(self):
        """Initial normalized image coordinates (x/z, y/z), float32, shape (N,2)."""
        return (self.points[:, :2] / self.points[:, 2:3]).astype(np.float32)

    def advance(self):
        """Integrate the truth one dt (analyzeEKFSimulation.cpp:57-84)."""
        dt = self.dt
        self.pos = self.pos + _rotate(self.quat, dt * self.vel + 0.5 * dt * dt * self.acc)
        on = np.linalg.norm(self.omega)
        if on < 1e-10:
            dq = np.array([1.0, *(self.omega * dt)])
            dq /= np.linalg.norm(dq)
        else:
            th = dt * on
            dq = np.array([np.cos(th / 2), *(self.omega / on * np.sin(th / 2))])
        dqi = _conj(dq) / np.dot(dq, dq)
        self.vel = _rotate(dqi, self.vel + dt * self.acc)
        self.acc = _rotate(dqi, self.acc)
        self.quat = _qmul(self.quat, dq)

    def measure(self):
        """(z[N,2], R[N,4] col-major 2x2, pass[N]) for the current truth pose (:101-125)."""
        qi = _conj(self.quat) / np.dot(self.quat, self.quat)
        Rm = np.array([[1 - 2 * (qi[2] ** 2 + qi[3] ** 2), 2 * (qi[1] * qi[2] - qi[0] * qi[3]), 2 * (qi[1] * qi[3] + qi[0] * qi[2])],
                       [2 * (qi[1] * qi[2] + qi[0] * qi[3]), 1 - 2 * (qi[1] ** 2 + qi[3] ** 2), 2 * (qi[2] * qi[3] - qi[0] * qi[1])],
                       [2 * (qi[1] * qi[3] - qi[0] * qi[2]), 2 * (qi[2] * qi[3] + qi[0] * qi[1]), 1 - 2 * (qi[1] ** 2 + qi[2] ** 2)]])
        fp = (self.points - self.pos) @ Rm.T
        z = (fp[:, :2] / fp[:, 2:3]).astype(np.float32)
        R = np.zeros((self.N, 4), dtype=np.float32)
        R[:, 0] = self.meas_var
        R[:, 3] = self.meas_var
        return z, R, np.ones(self.N, dtype=np.uint8)

    def frames(self, count):
        for _ in range(count):
            self.advance()
            yield self.measure()


def translated_sequence(base, frames, dx=-1.4, dy=-0.45):
    """Image sequence of a fronto-parallel textured plane under lateral camera motion: frame i is the 8-bit image `base`
    translated by i * (dx, dy) pixels (bilinear, wrap-around).  Used by the full-loop benchmark and the replay tests
    (the reference ships single test images, no sequence)."""
    base = np.asarray(base).astype(np.float32)
    out = []
    for i in range(frames):
        sx, sy = i * dx, i * dy
        ix, iy = int(np.floor(sx)), int(np.floor(sy))
        fx, fy = sx - ix, sy - iy
        a = np.roll(base, (iy, ix), axis=(0, 1))
        b = np.roll(base, (iy, ix + 1), axis=(0, 1))
        c = np.roll(base, (iy + 1, ix), axis=(0, 1))
        d = np.roll(base, (iy + 1, ix + 1), axis=(0, 1))
        img = (1 - fy) * ((1 - fx) * a + fx * b) + fy * ((1 - fx) * c + fx * d)
        out.append(np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8)))
    return out
