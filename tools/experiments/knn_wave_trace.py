import sys, os, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-slam-detection_amd", "python"))
from lsd_amd import lio, synth, capi
scene = synth.Scene(half=100.0, n_boxes=40, seed=1)
mp = scene.sample_surface(10_000_000, seed=2, sigma=0.01)
m = lio.Map(resolution=0.5, max_points=10_500_000, max_voxels=1 << 20, stencil=19)
m.add(mp)
s = lio.Scan(max_raw=1 << 18, max_ds=100000)
pos = np.array([3.0, -2.0, 1.8]); q = synth.quat_from_rotvec([0, 0, 0.3])
raw, _ = synth.make_scan(scene, pos, q, seed=5, fov_deg=(-24.8, 2.0), max_range=150.0)
s.upload(raw); n = s.voxel_downsample(0.5)
st = synth.state_from_pose(pos, q)
for i in range(5):
    ne = lio.linearize(m, s, st, True)
L = capi.lib()
nw = 16384
buf = np.zeros((nw, 5), np.uint64)
L.lio_debug_knn_trace.argtypes = [C.c_void_p, C.c_int]
print("rc", L.lio_debug_knn_trace(buf.ctypes.data_as(C.c_void_p), nw), "n_ds", n, "waves", nw)
t = buf[:, :4].astype(np.int64)
ok = t[:, 0] > 0
t = t[ok]; sm = buf[ok, 4]
t0 = t[:, 0].min()
tick = 10.0  # ns per wall_clock64 tick (100 MHz)
st_, en = (t[:, 0] - t0) * tick / 1e3, (t[:, 3] - t0) * tick / 1e3
print("kernel span us: %.1f" % en.max())
print("start time percentiles us", np.percentile(st_, [0, 10, 25, 50, 75, 90, 99, 100]).round(1))
print("end time percentiles us", np.percentile(en, [0, 10, 25, 50, 75, 90, 99, 100]).round(1))
life = en - st_
print("wave lifetime us", np.percentile(life, [0, 10, 50, 90, 99, 100]).round(1))
for name, a, b in (("probe", 0, 1), ("sweep", 1, 2), ("merge+out", 2, 3)):
    d = (t[:, b] - t[:, a]) * tick / 1e3
    print(name, "us percentiles", np.percentile(d, [10, 50, 90, 99, 100]).round(2))
# concurrency over time
ev = np.concatenate([np.stack([st_, np.ones_like(st_)], 1), np.stack([en, -np.ones_like(en)], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1])
for tt in range(0, int(en.max()) + 1, 3):
    i = np.searchsorted(ev[:, 0], tt)
    print("t=%2d us resident waves %d" % (tt, conc[min(i, len(conc) - 1)]))
r1 = st_ < 5.0
for nm, sel in (("round1", r1), ("round2", ~r1)):
    print(nm, "waves", sel.sum(), "lifetime med %.1f" % np.median(life[sel]), "probe %.2f sweep %.2f merge %.2f" % tuple(np.median((t[sel, b] - t[sel, a]) * tick / 1e3) for a, b in ((0, 1), (1, 2), (2, 3))))
xcc = (sm.astype(np.int64) >> 0)
print("distinct smid", len(np.unique(xcc)))
