"""Seeded synthetic data for the LIO scan-matching path.

The reference's demo_data / NCLT inputs are external downloads that are not in
the tree; this module is the stand-in: an analytic scene (ground plane +
axis-aligned boxes + perimeter walls), a 64-beam spinning-lidar ray caster and
a surface sampler for prebuilt maps.  Pure numpy, deterministic per seed.
"""
import numpy as np


# ---------------------------------------------------------------------------
# SE(3) helpers (quaternions are (x, y, z, w), Eigen coeffs order)
# ---------------------------------------------------------------------------
def quat_from_rotvec(v):
    v = np.asarray(v, np.float64)
    a = np.linalg.norm(v)
    if a < 1e-300:
        return np.array([0.0, 0.0, 0.0, 1.0])
    ax = v / a
    return np.concatenate([ax * np.sin(a / 2), [np.cos(a / 2)]])


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def quat_angle(a, b):
    """rotation angle (rad) between two unit quaternions"""
    d = abs(float(np.dot(a, b)) / (np.linalg.norm(a) * np.linalg.norm(b)))
    return 2 * np.arccos(min(1.0, d))


# ---------------------------------------------------------------------------
# scene
# ---------------------------------------------------------------------------
class Scene:
    """ground plane z = 0 on [-half, half]^2, `n_boxes` axis-aligned boxes, 4 perimeter walls"""

    def __init__(self, half=100.0, n_boxes=40, seed=1, wall_h=8.0, box_size=(3.0, 20.0), box_h=(2.0, 12.0), keep_clear=6.0):
        rng = np.random.default_rng(seed)
        self.half = float(half)
        self.wall_h = float(wall_h)
        c = rng.uniform(-half * 0.95, half * 0.95, size=(n_boxes, 2))
        s = rng.uniform(box_size[0], box_size[1], size=(n_boxes, 2))
        h = rng.uniform(box_h[0], box_h[1], size=(n_boxes,))
        lo = np.concatenate([c - s / 2, np.zeros((n_boxes, 1))], 1)
        hi = np.concatenate([c + s / 2, h[:, None]], 1)
        # keep a clear disc around the origin so that the sensor is never inside a box
        keep = ~((lo[:, 0] < keep_clear) & (hi[:, 0] > -keep_clear) & (lo[:, 1] < keep_clear) & (hi[:, 1] > -keep_clear))
        self.lo, self.hi = lo[keep], hi[keep]

    # -- surfaces as rectangles: (origin, edge_u, edge_v) ------------------------------------------
    def _rects(self):
        H = self.half
        rects = [(np.array([-H, -H, 0.0]), np.array([2 * H, 0, 0.0]), np.array([0, 2 * H, 0.0]))]  # ground
        wh = self.wall_h
        rects += [(np.array([-H, -H, 0.0]), np.array([2 * H, 0, 0.0]), np.array([0, 0, wh])),
                  (np.array([-H, H, 0.0]), np.array([2 * H, 0, 0.0]), np.array([0, 0, wh])),
                  (np.array([-H, -H, 0.0]), np.array([0, 2 * H, 0.0]), np.array([0, 0, wh])),
                  (np.array([H, -H, 0.0]), np.array([0, 2 * H, 0.0]), np.array([0, 0, wh]))]
        for lo, hi in zip(self.lo, self.hi):
            d = hi - lo
            rects += [(np.array([lo[0], lo[1], hi[2]]), np.array([d[0], 0, 0.0]), np.array([0, d[1], 0.0])),  # top
                      (lo.copy(), np.array([d[0], 0, 0.0]), np.array([0, 0, d[2]])),
                      (np.array([lo[0], hi[1], lo[2]]), np.array([d[0], 0, 0.0]), np.array([0, 0, d[2]])),
                      (lo.copy(), np.array([0, d[1], 0.0]), np.array([0, 0, d[2]])),
                      (np.array([hi[0], lo[1], lo[2]]), np.array([0, d[1], 0.0]), np.array([0, 0, d[2]]))]
        return rects

    def sample_surface(self, n, seed=0, sigma=0.01, chunk=2_000_000):
        """n surface samples (uniform by area) with isotropic Gaussian noise; returns (n, 4) f32 XYZI"""
        rng = np.random.default_rng(seed)
        rects = self._rects()
        O = np.stack([r[0] for r in rects])
        U = np.stack([r[1] for r in rects])
        V = np.stack([r[2] for r in rects])
        area = np.linalg.norm(np.cross(U, V), axis=1)
        p = area / area.sum()
        out = np.empty((n, 4), np.float32)
        for a in range(0, n, chunk):
            m = min(chunk, n - a)
            idx = rng.choice(len(rects), size=m, p=p)
            uv = rng.random((m, 2))
            pts = O[idx] + U[idx] * uv[:, :1] + V[idx] * uv[:, 1:]
            pts += rng.normal(0.0, sigma, size=(m, 3))
            out[a:a + m, :3] = pts.astype(np.float32)
            out[a:a + m, 3] = rng.uniform(0, 255, size=m).astype(np.float32)
        return out

    def raycast(self, origin, dirs, max_range=100.0):
        """first-hit range along unit `dirs` (n,3) from `origin` (3,) or per-ray origins (n,3); inf where nothing is hit"""
        d = np.asarray(dirs, np.float64)
        n = len(d)
        o = np.broadcast_to(np.asarray(origin, np.float64), (n, 3))
        t_best = np.full(n, np.inf)
        with np.errstate(divide="ignore", invalid="ignore"):
            # ground
            tg = -o[:, 2] / d[:, 2]
            hit = (d[:, 2] < 0) & (tg > 0)
            pg = o[:, :2] + d[:, :2] * tg[:, None]
            hit &= (np.abs(pg[:, 0]) <= self.half) & (np.abs(pg[:, 1]) <= self.half)
            t_best = np.where(hit, tg, t_best)
            # perimeter walls (seen from inside)
            for ax in (0, 1):
                for sgn in (-1.0, 1.0):
                    tw = (sgn * self.half - o[:, ax]) / d[:, ax]
                    pw = o + d * tw[:, None]
                    ok = (tw > 0) & (np.abs(pw[:, 1 - ax]) <= self.half) & (pw[:, 2] >= 0) & (pw[:, 2] <= self.wall_h)
                    t_best = np.where(ok & (tw < t_best), tw, t_best)
            # boxes near enough to matter (slab test)
            c = (self.lo + self.hi) / 2
            rad = np.linalg.norm((self.hi - self.lo)[:, :2] / 2, axis=1)
            om = o.mean(0)
            spread = np.linalg.norm(o[:, :2] - om[None, :2], axis=1).max() if n else 0.0
            near = np.linalg.norm(c[:, :2] - om[None, :2], axis=1) - rad < max_range + spread
            inv = 1.0 / d
            for lo, hi in zip(self.lo[near], self.hi[near]):
                t1 = (lo[None, :] - o) * inv
                t2 = (hi[None, :] - o) * inv
                tmin = np.nanmax(np.minimum(t1, t2), axis=1)
                tmax = np.nanmin(np.maximum(t1, t2), axis=1)
                ok = (tmax >= tmin) & (tmin > 0)
                t_best = np.where(ok & (tmin < t_best), tmin, t_best)
        t_best[t_best > max_range] = np.inf
        return t_best


def lidar_dirs(n_beams=64, n_az=1875, fov_deg=(-25.0, 15.0)):
    """unit directions in the lidar frame, beam-major within each azimuth step (firing order);
    returns dirs (n_az*n_beams, 3) and the per-point time offset in [0, 1) of a sweep"""
    el = np.deg2rad(np.linspace(fov_deg[0], fov_deg[1], n_beams))
    az = np.linspace(0.0, 2 * np.pi, n_az, endpoint=False)
    A, E = np.meshgrid(az, el, indexing="ij")
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    t = np.repeat(np.arange(n_az) / n_az, n_beams)
    return d, t


def make_scan(scene, pos, quat, seed=0, n_beams=64, n_az=1875, sigma=0.02, max_range=100.0, blind=0.1, fov_deg=(-25.0, 15.0)):
    """one static-sensor scan from world pose (pos, quat xyzw); returns (body XYZI f32 (n,4), time offset f32 (n,))"""
    rng = np.random.default_rng(seed)
    d_body, t = lidar_dirs(n_beams, n_az, fov_deg)
    R = quat_to_R(np.asarray(quat, np.float64))
    d_world = d_body @ R.T
    r = scene.raycast(pos, d_world, max_range)
    ok = np.isfinite(r)
    r = r + rng.normal(0.0, sigma, size=r.shape)
    ok &= r > blind
    pts = d_body[ok] * r[ok, None]
    inten = rng.uniform(0, 255, size=ok.sum())
    out = np.concatenate([pts, inten[:, None]], 1).astype(np.float32)
    return out, t[ok].astype(np.float32)


def perturb_pose(pos, quat, seed, max_t=0.3, max_deg=2.0):
    rng = np.random.default_rng(seed)
    dt = rng.uniform(-1, 1, 3)
    dt *= rng.uniform(0, max_t) / np.linalg.norm(dt)
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    ang = np.deg2rad(rng.uniform(0, max_deg))
    return np.asarray(pos, np.float64) + dt, quat_mul(np.asarray(quat, np.float64), quat_from_rotvec(ax * ang))


def state_from_pose(pos, quat, grav=(0.0, 0.0, -9.809)):
    """26-double state (pos3 rot4 R_il4 t_il3 vel3 bg3 ba3 grav3) with identity extrinsics"""
    s = np.zeros(26)
    s[0:3] = pos
    s[3:7] = quat
    s[10] = 1.0
    s[23:26] = grav
    return s


# ---------------------------------------------------------------------------
# moving sensor: analytic IMU-body trajectory, IMU samples and motion-distorted sweeps (the inputs of the FastLIO front
# half: fastlio_imu_enqueue / fastlio_pcl_enqueue)
# ---------------------------------------------------------------------------
class Trajectory:
    """Body (IMU) pose over time: at rest for `t_static` seconds (IMU initialisation needs > 100 quiet samples), then a
    smooth drive with yaw / pitch / roll oscillations.  World frame z up, gravity (0, 0, -9.81)."""

    def __init__(self, p0=(0.0, 0.0, 1.8), t_static=1.5, speed=6.0, tau=1.5, sway=1.5, yaw_amp=0.5, pitch_amp=0.03, roll_amp=0.04, heading=0.3):
        self.p0 = np.asarray(p0, np.float64)
        self.t_static, self.speed, self.tau, self.sway = t_static, speed, tau, sway
        self.yaw_amp, self.pitch_amp, self.roll_amp, self.heading = yaw_amp, pitch_amp, roll_amp, heading

    def _u(self, t):
        return np.maximum(0.0, np.asarray(t, np.float64) - self.t_static)

    def pos(self, t):
        u = self._u(t)
        s = self.speed * (u - self.tau * (1.0 - np.exp(-u / self.tau)))  # distance along the heading, s'(0) = 0
        lat = self.sway * (1.0 - np.cos(0.7 * u))
        z = 0.05 * (1.0 - np.cos(2.3 * u))
        c, sn = np.cos(self.heading), np.sin(self.heading)
        return np.stack([self.p0[0] + c * s - sn * lat, self.p0[1] + sn * s + c * lat, self.p0[2] + z], -1)

    def R(self, t):
        u = self._u(t)
        yaw = self.heading + self.yaw_amp * (1.0 - np.cos(0.5 * u))
        pitch = self.pitch_amp * (1.0 - np.cos(1.9 * u))
        roll = self.roll_amp * (1.0 - np.cos(1.3 * u))
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        R = np.empty(np.shape(u) + (3, 3))
        R[..., 0, 0] = cy * cp; R[..., 0, 1] = cy * sp * sr - sy * cr; R[..., 0, 2] = cy * sp * cr + sy * sr
        R[..., 1, 0] = sy * cp; R[..., 1, 1] = sy * sp * sr + cy * cr; R[..., 1, 2] = sy * sp * cr - cy * sr
        R[..., 2, 0] = -sp;     R[..., 2, 1] = cp * sr;                R[..., 2, 2] = cp * cr
        return R

    def quat(self, t):
        """(x, y, z, w) of R(t), scalar t"""
        R = self.R(float(t))
        w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
        return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])

    def imu(self, t, g=9.81):
        """ideal gyro (rad/s, body) and accelerometer (specific force, m/s^2, body) at scalar t"""
        h = 1e-5
        R = self.R(t)
        W = R.T @ (self.R(t + h) - self.R(t - h)) / (2 * h)
        gyr = np.array([W[2, 1] - W[1, 2], W[0, 2] - W[2, 0], W[1, 0] - W[0, 1]]) / 2
        h = 1e-4
        acc_w = (self.pos(t + h) - 2 * self.pos(t) + self.pos(t - h)) / (h * h)
        return gyr, R.T @ (acc_w + np.array([0.0, 0.0, g]))


class FigureEight(Trajectory):
    """SURVEY.md 8d config 3 stand-in for a recorded drive: at rest for `t_static` seconds, then a figure of eight (lemniscate of Gerono,
    x = A sin s, y = B sin s cos s) driven at ~`speed` m/s with the heading along the velocity and small pitch / roll oscillations"""

    def __init__(self, p0=(0.0, 0.0, 1.8), t_static=1.5, speed=5.0, tau=2.0, A=150.0, B=150.0, pitch_amp=0.02, roll_amp=0.03):
        super().__init__(p0=p0, t_static=t_static, speed=speed, tau=tau)
        self.A, self.B = A, B
        self.pitch_amp, self.roll_amp = pitch_amp, roll_amp
        ss = np.linspace(0, 2 * np.pi, 4001)
        self.length = float(np.sum(np.hypot(np.diff(A * np.sin(ss)), np.diff(B * np.sin(ss) * np.cos(ss)))))

    def _s(self, t):
        u = self._u(t)
        d = self.speed * (u - self.tau * (1.0 - np.exp(-u / self.tau)))  # distance-like parameter, zero velocity at the start
        return 2 * np.pi * d / self.length

    def pos(self, t):
        s = self._s(t)
        u = self._u(t)
        z = 0.05 * (1.0 - np.cos(1.1 * u))
        return np.stack([self.p0[0] + self.A * np.sin(s), self.p0[1] + self.B * np.sin(s) * np.cos(s), self.p0[2] + z], -1)

    def R(self, t):
        s, u = self._s(t), self._u(t)
        yaw = np.arctan2(self.B * np.cos(2 * s), self.A * np.cos(s))
        pitch = self.pitch_amp * (1.0 - np.cos(0.9 * u))
        roll = self.roll_amp * (1.0 - np.cos(0.7 * u))
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        R = np.empty(np.shape(u) + (3, 3))
        R[..., 0, 0] = cy * cp; R[..., 0, 1] = cy * sp * sr - sy * cr; R[..., 0, 2] = cy * sp * cr + sy * sr
        R[..., 1, 0] = sy * cp; R[..., 1, 1] = sy * sp * sr + cy * cr; R[..., 1, 2] = sy * sp * cr - cy * sr
        R[..., 2, 0] = -sp;     R[..., 2, 1] = cp * sr;                R[..., 2, 2] = cp * cr
        return R


def make_sweep(scene, traj, t_beg, scan_period=0.1, ext_R=np.eye(3), ext_t=(0.0, 0.0, 0.0), seed=0, n_beams=64, n_az=1875, sigma=0.02,
               max_range=100.0, fov_deg=(-25.0, 15.0)):
    """one sweep of a MOVING spinning lidar: ray i leaves at t_beg + stamp_i from the pose the trajectory has then.
    Returns (lidar-frame XYZI f32 (n,4), stamp_us uint32 (n,)) in firing order; nothing is filtered (blind-zone returns stay)."""
    rng = np.random.default_rng(seed)
    d_l, frac = lidar_dirs(n_beams, n_az, fov_deg)
    stamp_us = np.round(frac * scan_period * 1e6).astype(np.uint32)
    t_az = t_beg + stamp_us[::n_beams].astype(np.float64) * 1e-6  # one pose per azimuth step
    Rw = traj.R(t_az) @ np.asarray(ext_R, np.float64)                 # (n_az, 3, 3) lidar -> world
    ow = traj.pos(t_az) + traj.R(t_az) @ np.asarray(ext_t, np.float64)
    d_w = np.einsum("aij,abj->abi", Rw, d_l.reshape(n_az, n_beams, 3)).reshape(-1, 3)
    o_w = np.repeat(ow, n_beams, axis=0)
    r = scene.raycast(o_w, d_w, max_range)
    ok = np.isfinite(r)
    r = r + rng.normal(0.0, sigma, size=r.shape)
    ok &= r > 0.0
    pts = d_l[ok] * r[ok, None]
    inten = rng.uniform(0, 255, size=ok.sum())
    return np.concatenate([pts, inten[:, None]], 1).astype(np.float32), stamp_us[ok]


def imu_stream(traj, t0, t1, rate=200.0, seed=0, gyr_sigma=0.0, acc_sigma=0.0):
    """IMU samples (stamp, gyr (3,), acc m/s^2 (3,)) for t0 <= t < t1"""
    rng = np.random.default_rng(seed)
    out = []
    k0, k1 = int(np.ceil(t0 * rate - 1e-9)), int(np.ceil(t1 * rate - 1e-9))
    for k in range(k0, k1):
        t = k / rate
        g, a = traj.imu(t)
        out.append((t, g + rng.normal(0, gyr_sigma, 3), a + rng.normal(0, acc_sigma, 3)))
    return out
