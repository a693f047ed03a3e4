"""The rule behind csrc/voxelgrid.hip::monster_component_sum -- the SEQUENTIAL f32 sum s <- fl(s + x_i) of a run of thousands of points computed 512
points per step by integer arithmetic inside the running sum's binade -- restated with numpy and held against the plain loop on adversarial data
(CPU; the device code itself is held against the oracle in tests/test_voxelgrid_monster_gpu.py)."""
import numpy as np

f32 = np.float32


def seq_sum(s, xs):
    s = f32(s)
    for x in xs:
        s = f32(s + f32(x))
    return s


def fast_step(s, xs):
    """(accepted, s_new): while the running sum stays in one binade [2^k, 2^(k+1)) every addition rounds to a multiple of q = 2^(k-23), so with
    s = S q:  fl(s + x) = (S + rne(x / q)) q  unless x / q is an exact tie; accepted iff no tie, every |x_i / q| < 2^21 and every partial
    integer sum strictly inside (2^23, 2^24) with the sign of S"""
    s = f32(s)
    be = int((np.array([s], f32).view(np.uint32)[0] >> 23) & 0xFF)
    if be < 1 or be > 254:
        return False, s
    e = be - 127 - 23
    S0 = int(np.ldexp(np.float64(s), -e))
    xs = np.asarray(xs, f32)
    with np.errstate(over="ignore", under="ignore"):
        r = np.ldexp(xs, np.int32(-e)).astype(f32)
    ok = np.abs(r) < f32(2097152.0)
    rn = np.rint(r).astype(f32)
    if not ok.all() or (np.abs(r - rn) == f32(0.5)).any():
        return False, s
    T = S0 + np.cumsum(rn.astype(np.int64))
    aT = T if S0 > 0 else -T
    if not ((aT > 2 ** 23) & (aT < 2 ** 24)).all():
        return False, s
    return True, f32(np.ldexp(np.float64(T[-1]), e))


def monster_sum(xs, step=512):
    s, fast, slow = f32(0), 0, 0
    for p in range(0, len(xs), step):
        ok, s2 = fast_step(s, xs[p:p + step])
        if ok:
            s, fast = s2, fast + 1
        else:
            s, slow = seq_sum(s, xs[p:p + step]), slow + 1
    return s, fast, slow


def _cases(rng, n):
    yield "positive coordinates of one voxel", rng.uniform(3.5, 4.0, n)
    yield "negative coordinates of one voxel", rng.uniform(-7.5, -7.0, n)
    yield "the voxel at the origin", np.concatenate([[0.0, 1e-30, 1e-42], rng.uniform(0, 0.5, n)])
    yield "straddling zero (cancellation, sign changes)", rng.uniform(-0.3, 0.3, n)
    yield "coarse grid (exact ties)", np.round(rng.uniform(3.5, 4.0, n) * 4096) / 4096
    yield "halves and odd multiples (ties in every step)", 0.5 + rng.integers(0, 2, n) * 2.0 ** -20
    yield "intensity", rng.uniform(0, 255, n)
    yield "integer intensity", np.floor(rng.uniform(0, 256, n))
    yield "two magnitudes", rng.permutation(np.concatenate([rng.uniform(1e-3, 2e-3, n // 2), rng.uniform(1e3, 2e3, n - n // 2)]))
    yield "outliers", np.where(rng.random(n) < 0.01, rng.uniform(-1e6, 1e6, n), rng.uniform(10, 10.5, n))
    yield "tiny", rng.uniform(1e-38, 3e-38, n)
    yield "huge", rng.uniform(1e33, 2e33, n)
    yield "wide dynamic range", rng.normal(0, 1, n) * 10.0 ** rng.integers(-20, 20)


def test_integer_sums_inside_the_binade_equal_the_sequential_f32_sum():
    rng = np.random.default_rng(1)
    accepted = {}
    for trial in range(6):
        n = int(rng.integers(2100, 9000))
        for name, xs in _cases(rng, n):
            xs = np.asarray(xs).astype(f32)
            with np.errstate(over="ignore"):
                a = seq_sum(0, xs)
                b, fast, slow = monster_sum(xs)
            assert a.view(np.uint32) == b.view(np.uint32), (name, trial, a, b)
            accepted[name] = accepted.get(name, 0) + fast / (fast + slow)
    # the runs the kernel is for -- one voxel's coordinates, intensities -- go through the integer path almost always
    for name in ("positive coordinates of one voxel", "negative coordinates of one voxel", "intensity"):
        assert accepted[name] / 6 > 0.4, (name, accepted[name] / 6)  # (runs of 2-9 k points: ~log2(n / 512) + 1 steps cross a binade; the share grows with the run)
