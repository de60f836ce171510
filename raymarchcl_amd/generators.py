"""Input generators for the render path.

* :func:`generate_scatter_offsets` -- the Monte-Carlo scatter table the kernel
  indexes with ``seed & 0x3fff`` (reference generators.clj:8-16).
* :func:`make_gyroid_volume` -- the benchmark byte volume (generators.clj:18-42).
* :func:`make_terrain` -- the alternative test volume (generators.clj:44-60).
* :func:`make_blob_volume` -- seeded procedural stand-in for the mesh volumes
  (bunny/dragon) whose STL sources are not part of the reference tree
  (meshvoxel.clj:12-14 loads them from a user path).

The reference seeds its table from ``System/nanoTime`` so no two reference runs
agree; here the generator is seeded (SplitMix64) so oracle and device see the
same table.  The distribution is the reference's: four iid U(-1,1) floats, the
4-vector normalised to unit length with the norm taken in double.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(seed, count):
    """count 64-bit outputs of SplitMix64 started at ``seed`` (vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform01(seed, count):
    """Doubles in [0,1) with 53 random bits (java.util.Random.nextDouble's range)."""
    return (_splitmix64(seed, count) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def generate_scatter_offsets(num=0x4000, seed=0):
    """Flat float32 array of ``num`` unit 4-vectors (generators.clj:8-16)."""
    u = _uniform01(seed, 4 * num).reshape(num, 4)
    v = (2.0 * u - 1.0).astype(np.float32)  # (float (- (* 2.0 nextDouble) 1.0))
    vd = v.astype(np.float64)
    x, y, z, w = vd[:, 0], vd[:, 1], vd[:, 2], vd[:, 3]
    m = 1.0 / np.sqrt(x * x + y * y + z * z + w * w)
    return (vd * m[:, None]).astype(np.float32).reshape(-1)


def _vres3(vres):
    if isinstance(vres, (int, np.integer)):
        return int(vres), int(vres), int(vres)
    rx, ry, rz = vres
    return int(rx), int(ry), int(rz)


def make_gyroid_volume(vres, z_chunk=32):
    """uint8 volume, x fastest (index z*rx*ry + y*rx + x), generators.clj:27-42.

    Only slabs with ``(z & 0x3f) >= 32`` are filled.  With
    ``v = |cos X sin Z + cos Y sin X + cos Z sin Y| - 1`` at
    ``(X,Y,Z) = (x,y,z)*s + (0.3875,0,0)``, ``s = 0.01*512/rx``: a thin shell
    ``|0.2 - v| < 0.05`` gets byte 64 or 128 in alternating 32-voxel x-stripes,
    ``v > 0.35`` gets 255.
    """
    rx, ry, rz = _vres3(vres)
    vox = np.zeros((rz, ry, rx), dtype=np.uint8)
    scl = 0.01 * (512.0 / rx)
    X = np.arange(rx, dtype=np.float64) * scl + 0.3875
    Y = np.arange(ry, dtype=np.float64) * scl + 0.0
    cx, sx = np.cos(X)[None, None, :], np.sin(X)[None, None, :]
    cy, sy = np.cos(Y)[None, :, None], np.sin(Y)[None, :, None]
    stripe = ((np.arange(rx) & 0x3F) < 32)[None, None, :]
    for z0 in range(0, rz, z_chunk):
        zs = np.arange(z0, min(z0 + z_chunk, rz))
        live = (zs & 0x3F) >= 32
        if not live.any():
            continue
        zs = zs[live]
        Z = zs.astype(np.float64) * scl + 0.0
        cz, sz = np.cos(Z)[:, None, None], np.sin(Z)[:, None, None]
        v = np.abs(cx * sz + cy * sx + cz * sy) - 1.0
        shell = np.abs(0.2 - v) < 0.05
        out = np.where(shell, np.where(stripe, 64, 128), np.where(v > 0.35, 255, 0))
        vox[zs] = out.astype(np.uint8)
    return vox.reshape(-1)


def make_terrain(vres):
    """Alternative test volume (generators.clj:44-60): two thin walls plus
    sine-modulated columns on a 32-voxel grid."""
    rx, ry, rz = _vres3(vres)
    vox = np.zeros((rz, ry, rx), dtype=np.uint8)
    flat = vox.reshape(-1)
    rxy = rx * ry
    ytop = int(ry * 0.666)
    for z in range(4):
        vox[z, :ytop, :] = 64
        xs = np.arange(rx)
        for y in range(ytop):
            flat[xs * rxy + y * rx + (rx - z - 1)] = 64
    for z in range(rz):
        dz = 16 - (z % 32)
        for x in range(rx):
            dx = 16 - (x % 32)
            if dx * dx + dz * dz <= 121:
                y = int(ry * (0.25 + 0.125 * (np.sin(z * 0.02) * np.cos(x * 0.03))))
                vox[z, : y + 1, x] = 255
    return vox.reshape(-1)


def make_blob_volume(vres, seed=7, blobs=160, z_chunk=16, radius=None, device=None):
    """Seeded procedural stand-in for the reference's mesh-derived volumes.

    Union of metaballs inside the unit cube: field ``f = sum r_i^2 / |p-c_i|^2``;
    ``f > 1`` -> 255 (solid), a thin band just outside -> 64/128 in alternating
    x-stripes (mimics the gyroid's material bands).  ``radius=(lo, hi)`` sets the blob radii:
    the default (0.03, 0.08) merges the blobs into one porous mass (93 % solid; what the golden
    fixtures use), (0.01, 0.03) gives separate bodies at ~8 % fill -- the bench's stand-in for
    the reference's mesh-derived volumes.
    """
    rx, ry, rz = _vres3(vres)
    u = _uniform01(seed, 4 * blobs).reshape(blobs, 4)
    c = 0.15 + 0.7 * u[:, :3]
    c[:, 1] = 0.05 + 0.55 * u[:, 1]
    r2 = ((0.03 + 0.05 * u[:, 3]) if radius is None else (radius[0] + (radius[1] - radius[0]) * u[:, 3])) ** 2
    if device is not None:
        return _blob_volume_torch(rx, ry, rz, c, r2, device)
    vox = np.zeros((rz, ry, rx), dtype=np.uint8)
    xs = ((np.arange(rx) + 0.5) / rx)[None, None, :]
    ys = ((np.arange(ry) + 0.5) / ry)[None, :, None]
    stripe = ((np.arange(rx) * 64 // max(rx // 4, 1)) & 0x3F) < 32
    stripe = stripe[None, None, :]
    for z0 in range(0, rz, z_chunk):
        zs = ((np.arange(z0, min(z0 + z_chunk, rz)) + 0.5) / rz)[:, None, None]
        f = np.zeros((zs.shape[0], ry, rx), dtype=np.float32)
        for i in range(blobs):
            d2 = (xs - c[i, 0]) ** 2 + (ys - c[i, 1]) ** 2 + (zs - c[i, 2]) ** 2
            f += (r2[i] / np.maximum(d2, 1e-9)).astype(np.float32)
        out = np.where(f > 1.0, 255, np.where(f > 0.93, np.where(stripe, 64, 128), 0))
        vox[z0 : z0 + zs.shape[0]] = out.astype(np.uint8)
    return vox.reshape(-1)


def _blob_volume_torch(rx, ry, rz, c, r2, device):
    """make_blob_volume's field evaluated by torch on ``device`` (512^3 x 160 blobs takes two
    minutes in numpy).  Same formula; the float32 sums may round differently from the numpy
    loop, which is fine for a stand-in INPUT volume: callers feed the same bytes to every
    consumer."""
    import torch

    dev = torch.device(device)
    xs = ((torch.arange(rx, device=dev, dtype=torch.float64) + 0.5) / rx).float()[None, None, :]
    ys = ((torch.arange(ry, device=dev, dtype=torch.float64) + 0.5) / ry).float()[None, :, None]
    stripe = (((torch.arange(rx, device=dev) * 64 // max(rx // 4, 1)) & 0x3F) < 32)[None, None, :]
    ct = torch.tensor(c, dtype=torch.float32, device=dev)
    rt = torch.tensor(r2, dtype=torch.float32, device=dev)
    out = torch.empty((rz, ry, rx), dtype=torch.uint8, device=dev)
    zc = max(1, (1 << 24) // (rx * ry))
    for z0 in range(0, rz, zc):
        z1 = min(z0 + zc, rz)
        zs = ((torch.arange(z0, z1, device=dev, dtype=torch.float64) + 0.5) / rz).float()[:, None, None]
        f = torch.zeros((z1 - z0, ry, rx), dtype=torch.float32, device=dev)
        for i in range(ct.shape[0]):
            d2 = (xs - ct[i, 0]) ** 2 + (ys - ct[i, 1]) ** 2 + (zs - ct[i, 2]) ** 2
            f += rt[i] / torch.clamp(d2, min=1e-9)
        band = torch.where(stripe, torch.tensor(64, dtype=torch.uint8, device=dev),
                           torch.tensor(128, dtype=torch.uint8, device=dev)).expand_as(f)
        o = torch.where(f > 1.0, torch.tensor(255, dtype=torch.uint8, device=dev),
                        torch.where(f > 0.93, band, torch.tensor(0, dtype=torch.uint8, device=dev)))
        out[z0:z1] = o
    return out.reshape(-1).cpu().numpy()


def make_sdf_volume(vres, kind="gyroid"):
    """float32 distance field [rz, ry, rx] over the volume box [-1, 1]^3 (cell centres) for the
    QUALITY MODE (rm_set_sdf_volume; not a reference feature).  World units, negative inside.

    ``gyroid``: shell of the reference's gyroid function (generators.clj:22-25, same scale as
    make_gyroid_volume) cut to the slabs the byte volume fills, distance approximated by
    value / gradient length (a conservative first-order estimate, scaled by 0.6);
    ``torus``: a torus with a sphere above it (exact distances)."""
    rx, ry, rz = _vres3(vres)
    z, y, x = np.meshgrid((np.arange(rz) + 0.5) / rz * 2 - 1, (np.arange(ry) + 0.5) / ry * 2 - 1,
                          (np.arange(rx) + 0.5) / rx * 2 - 1, indexing="ij")
    if kind == "torus":
        tor = np.sqrt((np.sqrt(x * x + z * z) - 0.55) ** 2 + (y + 0.3) ** 2) - 0.18
        sph = np.sqrt(x * x + (y - 0.25) ** 2 + z * z) - 0.3
        return np.minimum(tor, sph).astype(np.float32)
    if kind != "gyroid":
        raise KeyError(kind)
    k = 2.56  # cells * 0.01 * 512 / res = world * res/2 * that: 256 cells span 2.56 * 2 radians ... per unit
    X, Y, Z = (x + 1) * k + 0.3875, (y + 1) * k, (z + 1) * k
    v = np.cos(X) * np.sin(Z) + np.cos(Y) * np.sin(X) + np.cos(Z) * np.sin(Y)
    gx = -np.sin(X) * np.sin(Z) + np.cos(Y) * np.cos(X)
    gy = -np.sin(Y) * np.sin(X) + np.cos(Z) * np.cos(Y)
    gz = np.cos(X) * np.cos(Z) - np.sin(Z) * np.sin(Y)
    g = np.sqrt(gx * gx + gy * gy + gz * gz) * k + 1e-3
    shell = (np.abs(v) - 0.25) / g * 0.6
    slab = np.abs(((z + 1) * 2) % 1.0 - 0.75) - 0.25  # keep z-slabs like (z & 0x3f) >= 32
    return np.maximum(shell, slab / 2.0).astype(np.float32)
