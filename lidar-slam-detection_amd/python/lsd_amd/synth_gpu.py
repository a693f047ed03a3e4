"""Synthetic data generated ON the GPU with torch -- bench.py only (the numpy generators of synth.py stay the ones the tests use).

BASELINE.json config 3 ("streaming LIO frontend, incremental map to 1e7 pts") needs thousands of 64 x 1875-ray sweeps of a moving
sensor and config 4 a 5e7-point prebuilt map; numpy ray casting costs ~0.3 s per sweep and ~0.25 s per million surface samples on one
core -- minutes per run.  Here the same scene (synth.Scene: ground, perimeter walls, axis-aligned boxes), the same ray / trajectory model
(synth.make_sweep, synth.Scene.sample_surface) are evaluated with torch tensors on the device, milliseconds per sweep.  The random streams
are torch's, not numpy's: the clouds are NOT bit-identical to synth.py's (tests/test_synth_gpu.py compares the noise-free ranges)."""
import numpy as np

from . import synth


class Lawnmower(synth.Trajectory):
    """Config 3 stand-in for a long recorded drive that covers new ground all the time: at rest for `t_static` s, then rows along x joined by
    semicircles (radius = half the row spacing), driven at `speed` m/s with the heading along the velocity and small pitch / roll oscillations.
    Rows at y = y0 + k * spacing, x in [-half_len, half_len]."""

    def __init__(self, p0=(-440.0, -450.0, 1.8), t_static=1.5, speed=20.0, tau=3.0, half_len=440.0, spacing=100.0, rows=10, pitch_amp=0.01, roll_amp=0.015):
        super().__init__(p0=p0, t_static=t_static, speed=speed, tau=tau)
        self.half_len, self.spacing, self.rows = float(half_len), float(spacing), int(rows)
        self.pitch_amp, self.roll_amp = pitch_amp, roll_amp
        self.r = self.spacing / 2
        self.seg_row = 2 * self.half_len
        self.seg_turn = np.pi * self.r
        self.length = self.rows * self.seg_row + (self.rows - 1) * self.seg_turn

    def _d(self, t):
        u = self._u(t)
        return np.minimum(self.speed * (u - self.tau * (1.0 - np.exp(-u / self.tau))), self.length - 1e-6)  # arc length driven so far

    def _path(self, d):
        """position (x, y) relative to the start of row 0 and heading (yaw) at arc length d"""
        d = np.asarray(d, np.float64)
        per = self.seg_row + self.seg_turn
        k = np.floor(d / per)
        k = np.minimum(k, self.rows - 1)
        e = d - k * per                      # arc length inside (row k + the turn after it)
        fwd = (k % 2 == 0)                   # even rows run towards +x
        on_row = e <= self.seg_row
        sgn = np.where(fwd, 1.0, -1.0)
        x_row = np.where(fwd, e, self.seg_row - e)
        y_row = k * self.spacing
        yaw_row = np.where(fwd, 0.0, np.pi)
        # the turn: centre at the row's end, half a spacing towards the next row; the angle runs from -90 deg (forward rows) to +90 deg
        a = (e - self.seg_row) / self.r      # 0 .. pi
        cx = np.where(fwd, self.seg_row, 0.0)
        cy = k * self.spacing + self.r
        x_turn = cx + sgn * self.r * np.sin(a)
        y_turn = cy - self.r * np.cos(a)
        yaw_turn = np.where(fwd, a, np.pi - a)
        x = np.where(on_row, x_row, x_turn)
        y = np.where(on_row, y_row, y_turn)
        yaw = np.where(on_row, yaw_row, yaw_turn)
        return x, y, yaw

    def pos(self, t):
        x, y, _ = self._path(self._d(t))
        u = self._u(t)
        z = 0.05 * (1.0 - np.cos(1.1 * u))
        return np.stack([self.p0[0] + x, self.p0[1] + y, self.p0[2] + z], -1)

    def R(self, t):
        _, _, yaw = self._path(self._d(t))
        u = self._u(t)
        pitch = self.pitch_amp * (1.0 - np.cos(0.9 * u))
        roll = self.roll_amp * (1.0 - np.cos(0.7 * u))
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        R = np.empty(np.shape(u) + (3, 3))
        R[..., 0, 0] = cy * cp; R[..., 0, 1] = cy * sp * sr - sy * cr; R[..., 0, 2] = cy * sp * cr + sy * sr
        R[..., 1, 0] = sy * cp; R[..., 1, 1] = sy * sp * sr + cy * cr; R[..., 1, 2] = sy * sp * cr - cy * sr
        R[..., 2, 0] = -sp;     R[..., 2, 1] = cp * sr;                R[..., 2, 2] = cp * cr
        return R

    def duration(self):
        """seconds until the last row ends (the drive stops there)"""
        return self.t_static + self.length / self.speed + self.tau


class Sweeper:
    """synth.make_sweep on the device: one sweep of a moving 64-beam lidar, rays cast against the scene's ground, walls and boxes."""

    def __init__(self, scene, traj, device, n_beams=64, n_az=1875, fov_deg=(-25.0, 15.0), max_range=100.0, sigma=0.02, scan_period=0.1, seed=0):
        import torch

        self.torch = torch
        self.scene, self.traj, self.dev = scene, traj, device
        self.n_beams, self.n_az, self.max_range, self.sigma, self.scan_period, self.seed = n_beams, n_az, float(max_range), float(sigma), scan_period, seed
        d_l, frac = synth.lidar_dirs(n_beams, n_az, fov_deg)
        self.stamp_us_np = np.round(frac * scan_period * 1e6).astype(np.uint32)
        self.t_off = self.stamp_us_np[::n_beams].astype(np.float64) * 1e-6
        self.d_l = torch.from_numpy(d_l).to(device)                       # (n, 3) f64
        self.stamp_us = torch.from_numpy(self.stamp_us_np.astype(np.int64)).to(device)
        self.lo = torch.from_numpy(np.ascontiguousarray(scene.lo)).to(device)
        self.hi = torch.from_numpy(np.ascontiguousarray(scene.hi)).to(device)
        self.c_np = (scene.lo + scene.hi) / 2
        self.rad_np = np.linalg.norm((scene.hi - scene.lo)[:, :2] / 2, axis=1)
        self.gen = torch.Generator(device=device)

    def ranges(self, t_beg):
        """noise-free first-hit ranges of the sweep that starts at t_beg (inf where nothing is hit within max_range); (n,) f64 on the device"""
        torch = self.torch
        t_az = t_beg + self.t_off
        Rw = torch.from_numpy(np.ascontiguousarray(self.traj.R(t_az))).to(self.dev)      # (n_az, 3, 3)
        ow_np = self.traj.pos(t_az)
        ow = torch.from_numpy(np.ascontiguousarray(ow_np)).to(self.dev)                  # (n_az, 3)
        d = torch.einsum("aij,abj->abi", Rw, self.d_l.view(self.n_az, self.n_beams, 3)).reshape(-1, 3)
        o = ow.repeat_interleave(self.n_beams, dim=0)
        H, wh = self.scene.half, self.scene.wall_h
        inf = torch.tensor(float("inf"), dtype=torch.float64, device=self.dev)
        # ground
        tg = -o[:, 2] / d[:, 2]
        pg = o[:, :2] + d[:, :2] * tg[:, None]
        hit = (d[:, 2] < 0) & (tg > 0) & (pg[:, 0].abs() <= H) & (pg[:, 1].abs() <= H)
        t_best = torch.where(hit, tg, inf)
        # perimeter walls (seen from inside)
        for ax in (0, 1):
            for sgn in (-1.0, 1.0):
                tw = (sgn * H - o[:, ax]) / d[:, ax]
                pw = o + d * tw[:, None]
                ok = (tw > 0) & (pw[:, 1 - ax].abs() <= H) & (pw[:, 2] >= 0) & (pw[:, 2] <= wh)
                t_best = torch.where(ok & (tw < t_best), tw, t_best)
        # boxes near enough to matter (slab test, all of them at once)
        om = ow_np.mean(0)
        spread = np.linalg.norm(ow_np[:, :2] - om[None, :2], axis=1).max()
        near = np.nonzero(np.linalg.norm(self.c_np[:, :2] - om[None, :2], axis=1) - self.rad_np < self.max_range + spread)[0]
        if len(near):
            idx = torch.from_numpy(near).to(self.dev)
            lo, hi = self.lo[idx], self.hi[idx]                                          # (K, 3)
            inv = 1.0 / d
            t1 = (lo[None, :, :] - o[:, None, :]) * inv[:, None, :]                      # (n, K, 3)
            t2 = (hi[None, :, :] - o[:, None, :]) * inv[:, None, :]
            tlo, thi = torch.minimum(t1, t2), torch.maximum(t1, t2)
            tlo = torch.where(torch.isnan(tlo), -inf, tlo)                               # numpy's nanmax / nanmin
            thi = torch.where(torch.isnan(thi), inf, thi)
            tmin = tlo.max(dim=2).values
            tmax = thi.min(dim=2).values
            tb = torch.where((tmax >= tmin) & (tmin > 0), tmin, inf).min(dim=1).values
            t_best = torch.minimum(t_best, tb)
        return torch.where(t_best > self.max_range, inf, t_best)

    def sweep(self, k, t_beg=None):
        """sweep number k (starting at k * scan_period unless t_beg is given): (lidar-frame XYZI f32 (n, 4), stamp_us uint32 (n,)) as numpy
        arrays on the host, in firing order, nothing filtered -- what synth.make_sweep returns"""
        torch = self.torch
        r = self.ranges(k * self.scan_period if t_beg is None else t_beg)
        self.gen.manual_seed(self.seed + 7919 * k)
        ok = torch.isfinite(r)
        r = r + self.sigma * torch.randn(r.shape, dtype=torch.float64, device=self.dev, generator=self.gen)
        ok &= r > 0.0
        sel = torch.nonzero(ok).squeeze(1)
        pts = self.d_l[sel] * r[sel, None]
        inten = 255.0 * torch.rand(len(sel), dtype=torch.float64, device=self.dev, generator=self.gen)
        out = torch.cat([pts, inten[:, None]], 1).to(torch.float32).cpu().numpy()
        return out, self.stamp_us_np[sel.cpu().numpy()]


def sample_surface(scene, n, device, seed=0, sigma=0.01, chunk=10_000_000):
    """synth.Scene.sample_surface on the device: n surface samples (uniform by area) with isotropic Gaussian noise, XYZI f32 (n, 4) tensor"""
    import torch

    rects = scene._rects()
    O = torch.from_numpy(np.stack([r[0] for r in rects])).to(device)
    U = torch.from_numpy(np.stack([r[1] for r in rects])).to(device)
    V = torch.from_numpy(np.stack([r[2] for r in rects])).to(device)
    area = torch.linalg.norm(torch.linalg.cross(U, V), dim=1)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    out = torch.empty((n, 4), dtype=torch.float32, device=device)
    for a in range(0, n, chunk):
        m = min(chunk, n - a)
        idx = torch.multinomial(area / area.sum(), m, replacement=True, generator=gen)
        uv = torch.rand((m, 2), dtype=torch.float64, device=device, generator=gen)
        pts = O[idx] + U[idx] * uv[:, :1] + V[idx] * uv[:, 1:]
        pts += sigma * torch.randn((m, 3), dtype=torch.float64, device=device, generator=gen)
        out[a:a + m, :3] = pts.to(torch.float32)
        out[a:a + m, 3] = 255.0 * torch.rand(m, dtype=torch.float32, device=device, generator=gen)
    return out


def imu_stream(traj, t0, t1, rate=100.0, seed=0, gyr_sigma=0.0, acc_sigma=0.0, g=9.81):
    """synth.imu_stream vectorised over the samples (the same central differences of the trajectory): (stamps (m,), gyr (m, 3), acc (m, 3))"""
    rng = np.random.default_rng(seed)
    k0, k1 = int(np.ceil(t0 * rate - 1e-9)), int(np.ceil(t1 * rate - 1e-9))
    t = np.arange(k0, k1) / rate
    h = 1e-5
    R = traj.R(t)
    W = np.einsum("nji,njk->nik", R, (traj.R(t + h) - traj.R(t - h)) / (2 * h))
    gyr = np.stack([W[:, 2, 1] - W[:, 1, 2], W[:, 0, 2] - W[:, 2, 0], W[:, 1, 0] - W[:, 0, 1]], 1) / 2
    h = 1e-4
    acc_w = (traj.pos(t + h) - 2 * traj.pos(t) + traj.pos(t - h)) / (h * h)
    acc = np.einsum("nji,nj->ni", R, acc_w + np.array([0.0, 0.0, g]))
    return t, gyr + rng.normal(0, gyr_sigma, gyr.shape), acc + rng.normal(0, acc_sigma, acc.shape)


class _Still(synth.Trajectory):
    """a sensor at rest at (pos, R): the `trajectory` of a static scan"""

    def __init__(self, pos, R):
        self._p, self._R = np.asarray(pos, np.float64), np.asarray(R, np.float64)

    def pos(self, t):
        return np.broadcast_to(self._p, np.shape(t) + (3,)).copy()

    def R(self, t):
        return np.broadcast_to(self._R, np.shape(t) + (3, 3)).copy()


class StaticScanner:
    """synth.make_scan on the device: scans of a sensor at rest (the metric config's scan pool -- 128 scans of 64 x 1875 rays take ~85 s with
    numpy on one core, a second here).  One Sweeper (ray table, scene boxes on the device) serves every pose."""

    def __init__(self, scene, device, n_beams=64, n_az=1875, fov_deg=(-25.0, 15.0), max_range=100.0, sigma=0.02, blind=0.1):
        self.sw = Sweeper(scene, _Still(np.zeros(3), np.eye(3)), device, n_beams=n_beams, n_az=n_az, fov_deg=fov_deg, max_range=max_range, sigma=sigma)
        self.blind = float(blind)

    def scan(self, pos, quat, seed):
        """body-frame XYZI f32 (n, 4) on the DEVICE (a torch tensor): first hits within max_range, Gaussian range noise, returns closer than
        `blind` dropped -- what synth.make_scan returns, with torch's random stream (seeded by `seed`)"""
        sw, torch = self.sw, self.sw.torch
        sw.traj = _Still(pos, synth.quat_to_R(np.asarray(quat, np.float64)))
        r = sw.ranges(0.0)
        sw.gen.manual_seed(int(seed))
        ok = torch.isfinite(r)
        r = r + sw.sigma * torch.randn(r.shape, dtype=torch.float64, device=sw.dev, generator=sw.gen)
        ok &= r > self.blind
        sel = torch.nonzero(ok).squeeze(1)
        pts = sw.d_l[sel] * r[sel, None]
        inten = 255.0 * torch.rand(len(sel), dtype=torch.float64, device=sw.dev, generator=sw.gen)
        return torch.cat([pts, inten[:, None]], 1).to(torch.float32).contiguous()
