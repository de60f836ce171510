"""The two families of exact-outcome shortcuts in the march, next to the exact slab test
(renderer.cl:153-161) whose result they predict without evaluating it:

  * the per-ray slab filter of Tracer::march ("this distance estimate certainly does not walk
    the volume"; its second half, "the position is certainly inside the clip box", left the product in
    round 5: three floats fewer per ray were worth more than the slab tests it saved), and
  * the inside-by-a-margin test of the estimate (the slab test returns exactly +0).

rm_selftest_filter evaluates, for arbitrary (origin, direction, t, ground term), the shortcuts
and the exact test side by side on the device.  A shortcut may be conservative (say nothing) but
never wrong: over random rays and rays placed within a few ulps of every decision boundary
  filter "no walk"  =>  the exact test does not walk   (not 0 <= t_in < g)
  margin "inside"   =>  the exact test returns exactly +0."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cpu_contract")]  # this module checks against the CPU oracle


def _records():
    import raymarchcl_amd as rm
    from raymarchcl_amd import structs

    base = dict(width=64, height=64, vres=[64, 64, 64], iter=1, mat="metal")
    out = [("default", structs.encode_bytes(rm.render_options(**base)))]
    o = rm.render_options(**base)
    o.update(voxelBoundsMin=[-0.7, -0.31, -0.93], voxelBoundsMax=[0.45, 0.88, 0.52])
    out.append(("asymmetric clip box", structs.encode_bytes(o)))
    o = rm.render_options(**base)
    s = [1.3, 0.8, 1.1]
    o.update(voxelBounds=s, voxelBounds2=[2 * v for v in s], invVoxelScale=[0.5 / v for v in s],
             voxelBoundsMin=[-0.99 * v for v in s], voxelBoundsMax=[0.99 * v for v in s])
    out.append(("anisotropic box", structs.encode_bytes(o)))
    return out


def _rays(rng, lo, hi, count):
    """Random rays + rays whose march distance sits at / around the entry and exit of the box."""
    ro = rng.uniform(-3.5, 3.5, (count, 3))
    ro[: count // 4] = rng.uniform(lo, hi, (count // 4, 3))          # origins inside the box
    rd = rng.normal(size=(count, 3))
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    rd *= rng.choice([1.0, 1.0, 1.0, 0.37, 1.9], (count, 1))         # reflected rays are not unit length
    k = count // 8
    rd[:k, rng.integers(0, 3)] *= 1e-4                               # nearly axis-parallel
    rd[k:2 * k, :] = np.where(rng.random((k, 3)) < 0.4, 0.0, rd[k:2 * k, :])  # exactly axis-parallel
    rd[np.all(rd == 0, axis=1)] = [0.0, -1.0, 0.0]
    with np.errstate(divide="ignore", invalid="ignore"):
        t0, t1 = (lo - ro) / rd, (hi - ro) / rd
    near = np.nanmax(np.minimum(t0, t1), axis=1)
    far = np.nanmin(np.maximum(t0, t1), axis=1)
    t = rng.uniform(0.0, 12.0, count)
    pick = rng.integers(0, 8, count)
    ulps = rng.integers(-40, 41, count).astype(np.int32)
    for sel, base in ((pick == 1, near), (pick == 2, far)):
        b = np.where(np.isfinite(base), base, 1.0).astype(np.float32)
        moved = (b.view(np.int32) + ulps).view(np.float32)
        t = np.where(sel, moved, t)
    # ... somewhere between entry and exit (inside the box): the "inside" shortcuts' territory
    span = np.where(np.isfinite(near) & np.isfinite(far) & (far > near), far - near, 0.0)
    t = np.where(pick >= 6, np.where(np.isfinite(near), near, 0.0) + span * rng.uniform(-0.02, 1.02, count), t)
    # ... and a little before the entry: the ground term then decides whether the estimate walks
    g_consistent = np.minimum((rd[:, 1] * t + ro[:, 1]) + 1.0, 1e5)
    gap = np.where(np.isfinite(near), near - t, 1.0)
    g = np.where(pick == 3, gap * rng.uniform(0.999, 1.001, count), g_consistent)
    g = np.where(pick == 4, rng.uniform(-0.5, 4.0, count), g)
    g = np.where(pick == 5, rng.choice([0.0, -0.0, 1e-30, 1e5], count), g)
    t = np.where(np.isfinite(t), t, 0.5)  # (negative distances occur too: a camera below the ground steps back)
    return np.concatenate([ro, rd, t[:, None], g[:, None]], axis=1).astype(np.float32)


def test_filter_and_margin_shortcuts_never_contradict_the_exact_slab_test(native):
    from raymarchcl_amd import structs

    rng = np.random.default_rng(77)
    total = said_no_walk = margin = walks = 0
    with native.Context(0) as ctx:
        for name, rec in _records():
            lo = np.frombuffer(rec, np.float32, 3, offset=structs.FIELD_OFFSETS["voxelBoundsMin"])
            hi = np.frombuffer(rec, np.float32, 3, offset=structs.FIELD_OFFSETS["voxelBoundsMax"])
            for _ in range(4):
                rays = _rays(rng, lo.astype(np.float64), hi.astype(np.float64), 1 << 18)
                bits = ctx.selftest_filter(rec, rays)
                g = rays[:, 7]
                no_walk, _unused, walk, zero, marg = [(bits >> k) & 1 == 1 for k in range(5)]
                assert not np.any(no_walk & walk), (name, rays[no_walk & walk][:4])
                assert not np.any(marg & ~zero), (name, rays[marg & ~zero][:4])
                total += len(bits)
                said_no_walk += int(no_walk.sum())
                margin += int(marg.sum())
                walks += int(walk.sum())
    # the shortcuts are exercised, not vacuous
    assert said_no_walk > total // 10 and margin > total // 100 and walks > total // 20
