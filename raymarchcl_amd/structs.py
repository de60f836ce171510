"""Byte layout of the reference's ``TRenderOpts`` (544 B) and its encoder.

The reference derives the layout from the typedef in its kernel source with
thi.ng/structgen (core.clj:25-26) and encodes one struct per pass
(core.clj:99-106).  structgen is un-vendored; the layout below is the one
clang computes for the same typedef (renderer.cl:14-19, 35-78) under OpenCL
alignment rules (float3 occupies 16 B) and is re-derived from the reference
source by tests/test_oracle_vs_reference.py::test_layout_matches_reference_typedef whenever
/root/reference is present (the C side asserts the same offsets statically: csrc/rm_opts.h).

Encoding rules relied on by the reference call sites: little-endian, every
number narrowed to float32 / int32 / uint8, missing keys encode as 0, short
vectors and short arrays are zero padded (e.g. the ``ao`` preset supplies one
of four ``lightColor`` entries, materials.clj:61).
"""
import numpy as np

TRENDEROPTS_SIZE = 544
TMATERIAL_SIZE = 32

TMaterial = np.dtype(
    {
        "names": ["albedo", "r0", "smoothness", "dummy"],
        "formats": [("<f4", 4), "<f4", "<f4", ("<f4", 2)],
        "offsets": [0, 16, 20, 24],
        "itemsize": TMATERIAL_SIZE,
    }
)

_F3 = ("<f4", 4)  # float3 padded to 16 B; lane 3 is padding
_FIELDS = [
    ("eyePos", _F3, 0),
    ("targetPos", _F3, 16),
    ("up", _F3, 32),
    ("voxelBounds", _F3, 48),
    ("voxelBounds2", _F3, 64),
    ("voxelBoundsMin", _F3, 80),
    ("voxelBoundsMax", _F3, 96),
    ("invVoxelScale", _F3, 112),
    ("skyColor1", _F3, 128),
    ("skyColor2", _F3, 144),
    ("voxelRes", ("<i4", 4), 160),
    ("resolution", ("<i4", 2), 176),
    ("invAspect", "<f4", 184),
    ("time", "<f4", 188),
    ("fov", "<f4", 192),
    ("maxIter", "<i4", 196),
    ("maxVoxelIter", "<i4", 200),
    ("maxDist", "<f4", 204),
    ("startDist", "<f4", 208),
    ("eps", "<f4", 212),
    ("aoIter", "<i4", 216),
    ("aoStepDist", "<f4", 220),
    ("aoAmp", "<f4", 224),
    ("voxelSize", "<f4", 228),
    ("groundY", "<f4", 232),
    ("shadowIter", "<i4", 236),
    ("reflectIter", "<i4", 240),
    ("shadowBias", "<f4", 244),
    ("lightScatter", "<f4", 248),
    ("minLightAtt", "<f4", 252),
    ("gamma", "<f4", 256),
    ("exposure", "<f4", 260),
    ("dof", "<f4", 264),
    ("frameBlend", "<f4", 268),
    ("fogPow", "<f4", 272),
    ("flareAmp", "<f4", 276),
    ("mcTableLength", "<i4", 280),
    ("isoVal", "u1", 284),
    ("numLights", "u1", 285),
    ("lightPos", ("<f4", (4, 4)), 288),
    ("lightColor", ("<f4", (4, 4)), 352),
    ("materials", (TMaterial, 4), 416),
]

TRenderOpts = np.dtype(
    {
        "names": [f[0] for f in _FIELDS],
        "formats": [f[1] for f in _FIELDS],
        "offsets": [f[2] for f in _FIELDS],
        "itemsize": TRENDEROPTS_SIZE,
    }
)
assert TRenderOpts.itemsize == TRENDEROPTS_SIZE

FIELD_OFFSETS = {f[0]: f[2] for f in _FIELDS}


def _fill(dst, value):
    """Zero-padded copy of a (possibly shorter) nested sequence into ``dst``."""
    arr = np.asarray(value, dtype=np.float64)
    if arr.ndim == 0:
        dst[...] = arr
        return
    if arr.ndim == 1:
        n = min(arr.shape[0], dst.shape[0])
        dst[:n] = arr[:n]
        return
    for i in range(min(arr.shape[0], dst.shape[0])):
        _fill(dst[i], arr[i])


def encode(opts):
    """Map of option name -> value  ==>  one TRenderOpts record (numpy void
    scalar array of shape ())."""
    rec = np.zeros((), dtype=TRenderOpts)
    for name, _fmt, _off in _FIELDS:
        if name not in opts or opts[name] is None:
            continue
        v = opts[name]
        if name == "materials":
            for i, m in enumerate(v[:4]):
                for k in ("albedo", "r0", "smoothness", "dummy"):
                    if k in m:
                        if k in ("albedo", "dummy"):
                            _fill(rec["materials"][i][k], m[k])
                        else:
                            rec["materials"][i][k] = m[k]
        elif rec[name].ndim:
            _fill(rec[name], v)
        else:
            rec[name] = v
    return rec


def encode_bytes(opts):
    return encode(opts).tobytes()


def decode(buf):
    """544 bytes -> numpy record (for tests and debugging)."""
    return np.frombuffer(bytes(buf), dtype=TRenderOpts, count=1)[0]
