"""oracle -- TEST INFRASTRUCTURE ONLY.

ctypes bindings for
  * ``librm_restate.so``  -- our plain-C CPU restatement of the reference render
    path (oracle/rm_restate.c), buildable anywhere, and
  * ``_ref/libref_oracle.so`` -- the UNMODIFIED reference kernel compiled for
    x86-64 (oracle/Makefile `ref`), only buildable where /root/reference exists, and
  * ``_ref/renderer_gfx950_{fast,default,strict}.hsaco`` -- the same unmodified source
    compiled for gfx950 and linked against ROCm's own OpenCL built-in library (no stand-in;
    oracle/Makefile `ref_gfx950`), launched on the GPU through ``libref_gfx950_runner.so``
    (oracle/ref_gfx950_runner.cpp).

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
bench.py may import this package.  The product package (raymarchcl_amd) never
does.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_CL = "/root/reference/resources/renderer.cl"
TABLE_FLOATS = 0x4000 * 4
OPTS_SIZE = 544


class Stats(ctypes.Structure):
    _fields_ = [
        (n, ctypes.c_uint64)
        for n in (
            "vox_reads",
            "mc_reads",
            "rays",
            "dts_calls",
            "march_steps",
            "ao_calls",
            "primary_hits",
            "oob_material",
        )
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(ref=None):
    """Compile the restatement (always) and the reference build (when the
    reference tree is present, or when ``ref=True`` is forced)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "restate"])
    if ref is None:
        ref = os.path.exists(REFERENCE_CL)
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref", "ref_gfx950"])
    if os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        subprocess.check_call(["make", "-s", "-C", HERE, "runner"])


_restate = None
_refs = {}

_u8p = ctypes.POINTER(ctypes.c_uint8)
_f32p = ctypes.POINTER(ctypes.c_float)
_u32p = ctypes.POINTER(ctypes.c_uint32)


def _ptr(a, ty):
    return a.ctypes.data_as(ty)


def restate_lib():
    global _restate
    if _restate is None:
        path = os.path.join(HERE, "librm_restate.so")
        if not os.path.exists(path):
            build(ref=False)
        lib = ctypes.CDLL(path)
        lib.rmo_render_image.argtypes = [_u8p, _f32p, ctypes.c_void_p, _f32p, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(Stats)]
        lib.rmo_render_image.restype = None
        lib.rmo_render_image_masked.argtypes = lib.rmo_render_image.argtypes + [_u8p]
        lib.rmo_render_image_masked.restype = None
        lib.rmo_tonemap_image.argtypes = [_f32p, ctypes.c_void_p, _u32p, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int]
        lib.rmo_tonemap_image.restype = None
        lib.rmo_render_frame.argtypes = [_u8p, ctypes.c_void_p, _f32p, ctypes.c_int, _f32p, _u32p,
                                         ctypes.c_int, ctypes.c_int, ctypes.POINTER(Stats)]
        lib.rmo_render_frame.restype = None
        lib.rmo_render_frame_ids.argtypes = [_u8p, ctypes.c_void_p, _f32p, ctypes.c_int, _f32p, ctypes.c_int,
                                             ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_int, _u8p]
        lib.rmo_render_frame_ids.restype = None
        lib.rmo_render_sdf_frame.argtypes = [_f32p, ctypes.c_void_p, _f32p, ctypes.c_int, _f32p, _u32p,
                                             ctypes.c_int, ctypes.c_int]
        lib.rmo_render_sdf_frame.restype = None
        lib.rmo_hw_threads.restype = ctypes.c_int
        lib.rmo_set_seed_cast.argtypes = [ctypes.c_int]
        lib.rmo_set_seed_cast.restype = None
        lib.rmo_get_seed_cast.restype = ctypes.c_int
        for name in ("rmo_exp", "rmo_exp2"):
            getattr(lib, name).argtypes = [ctypes.c_float]
            getattr(lib, name).restype = ctypes.c_float
        lib.rmo_pow.argtypes = [ctypes.c_float, ctypes.c_float]
        lib.rmo_pow.restype = ctypes.c_float
        lib.rmo_f2i.argtypes = [ctypes.c_float]
        lib.rmo_f2i.restype = ctypes.c_int32
        lib.rmo_f2u.argtypes = [ctypes.c_float]
        lib.rmo_f2u.restype = ctypes.c_uint32
        lib.rmo_convert_int_sat.argtypes = [ctypes.c_float]
        lib.rmo_convert_int_sat.restype = ctypes.c_int32
        _restate = lib
    return _restate


def _ref_name(fma):
    """fma: False = the oracle build, True = FMA-contracted, "libm" = exp/exp2/pow from libm."""
    return {False: "libref_oracle.so", True: "libref_oracle_fma.so", "libm": "libref_oracle_libm.so"}[fma]


def have_ref(fma=False):
    return os.path.exists(os.path.join(HERE, "_ref", _ref_name(fma)))


def ref_lib(fma=False):
    if fma not in _refs:
        path = os.path.join(HERE, "_ref", _ref_name(fma))
        lib = ctypes.CDLL(path)
        lib.ref_shim_eval.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p]
        lib.ref_shim_eval.restype = ctypes.c_int
        lib.ref_render_image.argtypes = [_u8p, _f32p, ctypes.c_void_p, _f32p, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int]
        lib.ref_render_image.restype = None
        lib.ref_render_image_mt.argtypes = [_u8p, _f32p, ctypes.c_void_p, _f32p, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.ref_render_image_mt.restype = None
        lib.ref_tonemap_image.argtypes = [_f32p, ctypes.c_void_p, _u32p, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int]
        lib.ref_tonemap_image.restype = None
        _refs[fma] = lib
    return _refs[fma]


def _check(vox, mc, opts, pixels):
    assert vox.dtype == np.uint8 and vox.flags.c_contiguous
    assert mc.dtype == np.float32 and mc.flags.c_contiguous
    assert pixels.dtype == np.float32 and pixels.flags.c_contiguous
    assert len(opts) % OPTS_SIZE == 0


def render_image(vox, mc, opts, pixels, n=None, id0=0, id1=None, threads=0, stats=None,
                 undefined_mask=None):
    """Restatement of one RenderImage pass, in place on ``pixels`` (n*4 float32).
    ``undefined_mask`` (uint8[n], optional) gets 1 where the reference's
    behaviour is undefined (out-of-record material index)."""
    _check(vox, mc, opts, pixels)
    n = pixels.size // 4 if n is None else n
    id1 = n if id1 is None else id1
    ob = ctypes.create_string_buffer(bytes(opts), OPTS_SIZE)
    sp = ctypes.byref(stats) if stats is not None else None
    if undefined_mask is None:
        restate_lib().rmo_render_image(_ptr(vox, _u8p), _ptr(mc, _f32p), ob, _ptr(pixels, _f32p),
                                       n, id0, id1, threads, sp)
    else:
        assert undefined_mask.dtype == np.uint8 and undefined_mask.size >= n
        restate_lib().rmo_render_image_masked(_ptr(vox, _u8p), _ptr(mc, _f32p), ob,
                                              _ptr(pixels, _f32p), n, id0, id1, threads, sp,
                                              _ptr(undefined_mask, _u8p))
    return pixels


def tonemap_image(pixels, opts, n=None):
    n = pixels.size // 4 if n is None else n
    argb = np.zeros(n, dtype=np.uint32)
    ob = ctypes.create_string_buffer(bytes(opts)[:OPTS_SIZE], OPTS_SIZE)
    restate_lib().rmo_tonemap_image(_ptr(pixels, _f32p), ob, _ptr(argb, _u32p), n, 0, n)
    return argb


def render_sdf_frame(sdf, opts_array, mc_array, n, threads=0):
    """QUALITY MODE (not the reference's algorithm): the pipeline over a float32 distance
    field [rz, ry, rx].  -> (pixels float32[n*4], argb uint32[n])"""
    iters = len(opts_array) // OPTS_SIZE
    sdf = np.ascontiguousarray(sdf, dtype=np.float32).reshape(-1)
    mc_array = np.ascontiguousarray(mc_array, dtype=np.float32).reshape(-1)
    assert mc_array.size == iters * TABLE_FLOATS
    pixels = np.zeros(n * 4, dtype=np.float32)
    argb = np.zeros(n, dtype=np.uint32)
    ob = ctypes.create_string_buffer(bytes(opts_array), len(opts_array))
    restate_lib().rmo_render_sdf_frame(_ptr(sdf, _f32p), ob, _ptr(mc_array, _f32p), iters,
                                       _ptr(pixels, _f32p), _ptr(argb, _u32p), n, threads)
    return pixels, argb


def render_frame(vox, opts_array, mc_array, n, threads=0, stats=None, tonemap=True):
    """Restatement of the whole pipeline.  opts_array: iter*544 bytes;
    mc_array: float32 [iter, 0x4000*4].  -> (pixels float32[n*4], argb uint32[n] | None)"""
    iters = len(opts_array) // OPTS_SIZE
    mc_array = np.ascontiguousarray(mc_array, dtype=np.float32).reshape(-1)
    assert mc_array.size == iters * TABLE_FLOATS
    pixels = np.zeros(n * 4, dtype=np.float32)
    argb = np.zeros(n, dtype=np.uint32) if tonemap else None
    ob = ctypes.create_string_buffer(bytes(opts_array), len(opts_array))
    restate_lib().rmo_render_frame(_ptr(vox, _u8p), ob, _ptr(mc_array, _f32p), iters,
                                   _ptr(pixels, _f32p),
                                   _ptr(argb, _u32p) if tonemap else None, n, threads,
                                   ctypes.byref(stats) if stats is not None else None)
    return pixels, argb


def render_frame_ids(vox, opts_array, mc_array, n, ids, threads=0, undefined_mask=None):
    """The pipeline for the work-items ``ids`` only (all passes, in order).  -> pixels float32[n*4],
    zero except at the ids.  Duplicate ids are not allowed."""
    iters = len(opts_array) // OPTS_SIZE
    mc_array = np.ascontiguousarray(mc_array, dtype=np.float32).reshape(-1)
    assert mc_array.size == iters * TABLE_FLOATS
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    assert len(np.unique(ids)) == len(ids)
    pixels = np.zeros(n * 4, dtype=np.float32)
    ob = ctypes.create_string_buffer(bytes(opts_array), len(opts_array))
    restate_lib().rmo_render_frame_ids(_ptr(vox, _u8p), ob, _ptr(mc_array, _f32p), iters, _ptr(pixels, _f32p), n,
                                       ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), len(ids), threads,
                                       _ptr(undefined_mask, _u8p) if undefined_mask is not None else None)
    return pixels


def ref_render_image(vox, mc, opts, pixels, n=None, id0=0, id1=None, fma=False):
    """The reference kernel itself (x86 build), in place on ``pixels``."""
    _check(vox, mc, opts, pixels)
    n = pixels.size // 4 if n is None else n
    id1 = n if id1 is None else id1
    ob = ctypes.create_string_buffer(bytes(opts), OPTS_SIZE)
    ref_lib(fma).ref_render_image(_ptr(vox, _u8p), _ptr(mc, _f32p), ob, _ptr(pixels, _f32p), n,
                                  id0, id1)
    return pixels


def ref_render_image_mt(vox, mc, opts, pixels, threads, n=None, id0=0, id1=None, fma=False):
    """ref_render_image spread over `threads` host threads (as a CPU OpenCL device would)."""
    _check(vox, mc, opts, pixels)
    n = pixels.size // 4 if n is None else n
    id1 = n if id1 is None else id1
    ob = ctypes.create_string_buffer(bytes(opts), OPTS_SIZE)
    ref_lib(fma).ref_render_image_mt(_ptr(vox, _u8p), _ptr(mc, _f32p), ob, _ptr(pixels, _f32p), n,
                                     id0, id1, int(threads))
    return pixels


SHIM_OPS = {"convert_float3": 0, "convert_int3_sat": 1, "dot": 2, "exp": 3, "exp2": 4, "pow": 5, "fabs": 6,
            "sqrt": 7, "mad": 8, "mad3": 9, "max": 10, "max3": 11, "min": 12, "min3": 13, "mix3": 14,
            "mix3s": 15, "step": 16, "clamp": 17, "cross": 18, "length": 19, "normalize": 20,
            "get_global_id": 21}


def shim_eval(name, a, b=None, c=None, out_shape=None, out_dtype=np.float32, fma=False):
    """Call built-in ``name`` of the shim the reference object is linked against (test hook)."""
    a = np.ascontiguousarray(a)
    count = a.shape[0]
    out = np.zeros(out_shape if out_shape is not None else a.shape, dtype=out_dtype)
    args = [np.ascontiguousarray(x, dtype=np.float32) if x is not None else None for x in (b, c)]
    rc = ref_lib(fma).ref_shim_eval(SHIM_OPS[name], count, a.ctypes.data,
                                    args[0].ctypes.data if args[0] is not None else None,
                                    args[1].ctypes.data if args[1] is not None else None, out.ctypes.data)
    assert rc == 0
    return out


def ref_tonemap_image(pixels, opts, n=None, fma=False):
    n = pixels.size // 4 if n is None else n
    argb = np.zeros(n, dtype=np.uint32)
    ob = ctypes.create_string_buffer(bytes(opts)[:OPTS_SIZE], OPTS_SIZE)
    ref_lib(fma).ref_tonemap_image(_ptr(pixels, _f32p), ob, _ptr(argb, _u32p), n, 0, n)
    return argb


# ---- seed casts: x86-64 lowering (default) or GPU lowering (saturating) ----
class seed_cast:
    """``with oracle.seed_cast("gpu"):`` -- the restatement evaluates the (uint) seed casts of
    renderer.cl:267,334,471,472 as a GPU device does (saturate) instead of as x86-64 does (wrap)."""

    def __init__(self, mode):
        assert mode in ("x86", "gpu")
        self.mode = mode

    def __enter__(self):
        self.prev = restate_lib().rmo_get_seed_cast()
        restate_lib().rmo_set_seed_cast(1 if self.mode == "gpu" else 0)
        return self

    def __exit__(self, *a):
        restate_lib().rmo_set_seed_cast(self.prev)


# ---- the reference kernel compiled for gfx950 (needs a GPU; tests + tools only) ----
GFX950_BUILDS = ("fast", "default", "strict")
_gfx = {}
_runner = None


def gfx950_path(build):
    assert build in GFX950_BUILDS
    return os.path.join(HERE, "_ref", f"renderer_gfx950_{build}.hsaco")


def have_gfx950_ref(build="fast"):
    return os.path.exists(gfx950_path(build)) and os.path.exists(os.path.join(HERE, "libref_gfx950_runner.so"))


def _gfx950_runner():
    global _runner
    if _runner is None:
        # one HIP runtime per process: bind to the copy torch has loaded, as the product library does
        if os.environ.get("RAYMARCH_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = ctypes.CDLL(os.path.join(HERE, "libref_gfx950_runner.so"))
        lib.refg_last_error.restype = ctypes.c_char_p
        lib.refg_load.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        lib.refg_unload.argtypes = [ctypes.c_void_p]
        lib.refg_unload.restype = None
        lib.refg_render_frame.argtypes = [ctypes.c_void_p, _u8p, ctypes.c_size_t, _f32p, ctypes.c_void_p,
                                          ctypes.c_int, _f32p, ctypes.c_int, _u32p, ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_float)]
        _runner = lib
    return _runner


def gfx950_render_frame(vox, opts_array, mc_array, n, build="fast", local_size=64, tonemap=True,
                        pixels_in=None):
    """The reference kernels themselves (RenderImage per pass, then TonemapImage with record 0) on
    the GPU, from the code object `build`.  -> (pixels float32[n*4], argb | None, kernel_ms)"""
    lib = _gfx950_runner()
    if build not in _gfx:
        h = ctypes.c_void_p()
        if lib.refg_load(gfx950_path(build).encode(), ctypes.byref(h)) != 0:
            raise RuntimeError("gfx950 reference build: " + lib.refg_last_error().decode())
        _gfx[build] = h
    iters = len(opts_array) // OPTS_SIZE
    mc_array = np.ascontiguousarray(mc_array, dtype=np.float32).reshape(-1)
    assert mc_array.size == iters * TABLE_FLOATS
    assert vox.dtype == np.uint8 and vox.flags.c_contiguous
    pixels = np.zeros(n * 4, dtype=np.float32) if pixels_in is None else np.array(pixels_in, dtype=np.float32).reshape(-1)
    argb = np.zeros(n, dtype=np.uint32) if tonemap else None
    ms = ctypes.c_float(0.0)
    ob = ctypes.create_string_buffer(bytes(opts_array), len(opts_array))
    rc = lib.refg_render_frame(_gfx[build], _ptr(vox, _u8p), vox.size, _ptr(mc_array, _f32p), ob, iters,
                               _ptr(pixels, _f32p), 0 if pixels_in is None else 1,
                               _ptr(argb, _u32p) if tonemap else None, n, int(local_size), ctypes.byref(ms))
    if rc != 0:
        raise RuntimeError("gfx950 reference build: " + lib.refg_last_error().decode())
    return pixels, argb, float(ms.value)
