"""ctypes binding of libraymarch_hip.so (include/raymarch_hip.h).

There is deliberately no fallback: if the library has not been built, cannot
be loaded, or no gfx950 device is visible, the calls raise.  The oracle under
/oracle is never imported from here.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# RAYMARCH_LIB: load (and build) another file of this directory instead, e.g. an A/B
# variant built with extra flags by tools/ab_build.py; the product default is the name below
LIB_PATH = os.path.join(HERE, os.path.basename(os.environ.get("RAYMARCH_LIB", "libraymarch_hip.so")))
SOURCES = ["rm_kernels.hip", "rm_accel.hip", "rm_volgen.hip", "rm_api.hip", "rm_host.cpp"]
# -fno-slp-vectorize: packed f32 VALU (v_pk_add_f32 ...) buys nothing on this chip and its
# even-aligned register pairs cost moves and spills in a 64-VGPR kernel (measured -5 % frame time)
# Device libraries: hipcc's own set (spelled out because --hip-device-lib replaces it) PLUS opencl.bc,
# ROCm's OpenCL built-in library: the RM_CONTRACT_GFX950 kernels call the very functions the
# reference kernel links against when ROCm's OpenCL compiler builds it for this chip (csrc/rm_math.hpp)
DEVICE_LIBS = ["opencl.bc", "ocml.bc", "ockl.bc", "oclc_daz_opt_off.bc", "oclc_unsafe_math_off.bc",
               "oclc_finite_only_off.bc", "oclc_correctly_rounded_sqrt_on.bc", "oclc_wavefrontsize64_on.bc",
               "oclc_isa_version_950.bc", "oclc_abi_version_600.bc"]
# -disable-machine-sink / -disable-machine-licm: with these two machine-level code motion passes on,
# the greedy VGPR allocator of this compiler (ROCm 7.2 clang 22) produces instantiations of the frame
# kernel that render wrong pixels (shadow results of the wave-shared phases go missing) depending on
# the register budget -- reproduced, bisected and described in DESIGN_HISTORY.md section 4c
# (reproducer: tools/repro_gpucast_fault.sh of commit c80bec6).  Cost of switching them off: 0-4 % of the frame time.
# (the ROCm tree comes from ROCM_PATH / HIPCC like hipcc's own; the code-object version is spelled out so
#  that it cannot drift away from the oclc_abi_version_600 bitcode named above)
ROCM_PATH = os.environ.get("ROCM_PATH", "/opt/rocm")


def hipcc_flags(rocm_path=None):
    """The product's compiler flags for a ROCm tree at rocm_path (default: ROCM_PATH of this process)."""
    root = ROCM_PATH if rocm_path is None else rocm_path
    return ["--offload-arch=gfx950", "-O2", "-fno-slp-vectorize", "-std=c++17", "-ffp-contract=off", "-fPIC",
            "-shared", "-mcode-object-version=6", "-mllvm", "-disable-machine-sink", "-mllvm", "-disable-machine-licm",
            "--hip-device-lib-path=" + os.path.join(root, "amdgcn", "bitcode")] + \
           ["--hip-device-lib=" + b for b in DEVICE_LIBS]


HIPCC_FLAGS = hipcc_flags()


def kernel_source_files():
    """Names of the files in csrc/ the library is compiled from (what the repository tracks: scratch files, whose names
    start with `_` and which .gitignore keeps out of the history, are not sources)."""
    return sorted(f for f in os.listdir(CSRC) if not f.startswith("_") and os.path.isfile(os.path.join(CSRC, f)))


def _hipcc():
    return os.environ.get("HIPCC", os.path.join(ROCM_PATH, "bin", "hipcc"))

OPTS_BYTES = 544
TABLE_FLOATS = 0x4000 * 4

# every symbol include/raymarch_hip.h declares
EXPORTS = [
    "rm_last_error", "rm_abi_version", "rm_device_count", "rm_create", "rm_create_multi", "rm_num_devices",
    "rm_destroy", "rm_set_stream", "rm_set_seed_cast", "rm_set_contract", "rm_synchronize", "rm_pin_host_buffer",
    "rm_unpin_host_buffer", "rm_set_volume", "rm_set_volume_device",
    "rm_invalidate_volume", "rm_share_volume", "rm_stage_volume_device", "rm_stage_volume", "rm_commit_staged_volume", "rm_frame_device_full", "rm_last_table_build_ms",
    "rm_make_gyroid_volume", "rm_make_terrain_volume", "rm_voxelize_vertices", "rm_voxelize_scatter", "rm_make_heatmap_volume",
    "rm_render_image", "rm_render_image_range", "rm_render_image_counted", "rm_tonemap_image",
    "rm_render_frame", "rm_set_sdf_volume", "rm_render_sdf_frame", "rm_tiles_per_part", "rm_frame_device", "rm_resolve_device",
    "rm_frame_device_argb", "rm_resolve_device_argb", "rm_last_frame_breakdown",
    "rm_check_device_opts", "rm_last_frame_timing", "rm_frame_timing_history", "rm_debug_get_accel", "rm_debug_get_octants", "rm_debug_volume_band", "rm_debug_block_order", "rm_selftest_prims", "rm_selftest_filter",
    "rm_render_options", "rm_compute_eyepos", "rm_make_scatter_table", "rm_make_gyroid_host",
    "rm_vox_save", "rm_vox_info", "rm_vox_load",
]


# rm_set_contract names (include/raymarch_hip.h): whose arithmetic the kernels reproduce
CONTRACTS = {"cpu": 0, "gfx950": 1, "gfx950-strict": 1, "gfx950-default": 2}


def volume_band(opts_record):
    """-> (lo, hi): the part of the image height whose tile rows a frame of this 544-byte record dispatches first
    (rm_debug_volume_band; (0, 0) = none).  Host-side."""
    lo, hi = ctypes.c_double(), ctypes.c_double()
    check(lib().rm_debug_volume_band(bytes(opts_record[:OPTS_BYTES]), ctypes.byref(lo), ctypes.byref(hi)))
    return float(lo.value), float(hi.value)


def block_order(resx, n, passes, tile_first=0, tile_stride=1, xcd_rows=True, xcd_2d=-1, rows_desc=True, band=(0.0, 0.0)):
    """-> int64 array, one entry per hardware workgroup of that frame-kernel launch: tile << 8 | sub-block, or -1 for a
    workgroup that leaves at once (rm_debug_block_order).  Host-side."""
    args = (int(resx), int(n), int(passes), int(tile_first), int(tile_stride), int(bool(xcd_rows)), int(xcd_2d), int(bool(rows_desc)),
            float(band[0]), float(band[1]))
    blocks = lib().rm_debug_block_order(*args, None, 0)
    if blocks < 0:
        check(int(blocks))
    out = np.empty(blocks, np.int64)
    lib().rm_debug_block_order(*args, out.ctypes.data, blocks)
    return out


class RmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libraymarch_hip: {msg} (code {code})")
        self.code = code


class RenderArgs(ctypes.Structure):
    """rm_render_args of include/raymarch_hip.h; NaN = not given."""
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("vres", ctypes.c_int * 3),
                ("iter", ctypes.c_int), ("t", ctypes.c_double), ("eyepos", ctypes.c_double * 3),
                ("targetpos", ctypes.c_double * 3), ("fov_deg", ctypes.c_double), ("dof", ctypes.c_double),
                ("gamma", ctypes.c_double), ("ground_y", ctypes.c_double), ("voxel_size", ctypes.c_double),
                ("mat", ctypes.c_char_p)]


class Counters(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in (
        "vox_reads", "mc_reads", "rays", "dts_calls", "march_steps", "ao_calls", "primary_hits",
        "oob_material")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    files = [os.path.join(CSRC, f) for f in kernel_source_files()]
    files.append(os.path.join(HERE, "..", "include", "raymarch_hip.h"))
    files.append(os.path.abspath(__file__))  # the compiler flags live here
    return any(os.path.getmtime(f) > t for f in files if os.path.isfile(f))


def build(force=False, verbose=False, lint=None):
    """hipcc cross-compiles for gfx950 without a GPU; the .so stays in-tree.  A real compile of the
    PRODUCT library is followed by lint_kernels() (the generated code must be free of the compiler
    fault of DESIGN_HISTORY.md 4c; ~35 s): a library that fails it is removed again and the call raises.
    lint=False skips that (A/B variants: RAYMARCH_LIB names another file)."""
    if not force and not _stale():
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if lint is None:
        lint = os.path.basename(LIB_PATH) == "libraymarch_hip.so" and os.environ.get("RAYMARCH_SKIP_LINT", "0") != "1"
    if lint:
        try:
            lint_kernels()
        except RmError:
            os.remove(LIB_PATH)
            raise
    return LIB_PATH


def lint_kernels():
    """Compile the kernels to gfx950 assembly with the product flags and look for the two shapes
    of the compiler fault DESIGN_HISTORY.md 4c describes: a spill reload or an allocator-inserted copy in a
    block that is entered with exec = 0 (isa_exec_lint.py of this package), and spilled SGPRs.  Raises RmError if either is present --
    a library built from such code renders wrong or faults in some instantiations, silently.
    (build() runs this after every real compile of the product library; tests/test_isa_budget.py too; ~35 s, no GPU.)"""
    import re
    import tempfile

    from . import isa_exec_lint

    hipcc = _hipcc()
    flags = [f for f in HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    text = ""
    with tempfile.TemporaryDirectory() as d:
        for src in SOURCES:
            if not src.endswith(".hip"):
                continue
            out = os.path.join(d, src + ".s")
            subprocess.run([hipcc] + flags + ["--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out],
                           check=True, stderr=subprocess.DEVNULL)
            text += open(out).read() + "\n"
    lines = text.split("\n")
    fatal = [(name, off, ins) for name, lo, hi in isa_exec_lint.kernels(lines)
             for off, ins in isa_exec_lint.dead_vector_instructions(lines, lo, hi)]
    sgpr = [m.group(1) for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*?)\.sgpr_spill_count:\s+(\d+)", text, re.S)
            if int(m.group(2)) > 0]
    if fatal or sgpr:
        raise RmError(-3, f"kernel build hits the compiler fault of DESIGN_HISTORY.md 4c: vector instructions under exec = 0: {fatal[:4]}; "
                          f"kernels with spilled SGPRs: {sgpr[:4]}")
    return len(list(isa_exec_lint.kernels(lines)))


_lib = None
_vp = ctypes.c_void_p
_i = ctypes.c_int


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RmError(-2, f"{LIB_PATH} is not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7
    # (+ HSA runtime).  If this library pulled in /opt/rocm's copy first, torch
    # would later load a second runtime and find "No HIP GPUs".  Importing torch
    # first makes its copy the one our DT_NEEDED libamdhip64.so.7 binds to, so
    # torch tensors, torch streams and these kernels share one runtime.  Set
    # RAYMARCH_NO_TORCH=1 for a torch-free process (binds /opt/rocm's runtime).
    if os.environ.get("RAYMARCH_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = ctypes.CDLL(LIB_PATH)
    L.rm_last_error.restype = ctypes.c_char_p
    L.rm_create.argtypes = [_i, ctypes.POINTER(_vp)]
    L.rm_create_multi.argtypes = [ctypes.POINTER(_i), _i, ctypes.POINTER(_vp)]
    L.rm_num_devices.argtypes = [_vp]
    L.rm_invalidate_volume.argtypes = [_vp]
    L.rm_share_volume.argtypes = [_vp, _vp]
    L.rm_frame_device_full.argtypes = [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]
    L.rm_last_table_build_ms.argtypes = [_vp, ctypes.POINTER(ctypes.c_float)]
    L.rm_destroy.argtypes = [_vp]
    L.rm_destroy.restype = None
    L.rm_set_stream.argtypes = [_vp, _vp]
    L.rm_synchronize.argtypes = [_vp]
    L.rm_set_seed_cast.argtypes = [_vp, _i]
    L.rm_set_contract.argtypes = [_vp, _i]
    L.rm_pin_host_buffer.argtypes = [_vp, _vp, ctypes.c_size_t]
    L.rm_unpin_host_buffer.argtypes = [_vp, _vp]
    L.rm_set_volume.argtypes = [_vp, _vp, _i, _i, _i]
    L.rm_set_volume_device.argtypes = [_vp, _vp, _i, _i, _i]
    L.rm_stage_volume_device.argtypes = [_vp, _vp, _i, _i, _i, _i]
    L.rm_stage_volume.argtypes = [_vp, _vp, _i, _i, _i, _i]
    L.rm_commit_staged_volume.argtypes = [_vp]
    L.rm_make_gyroid_volume.argtypes = [_vp, _i, _i, _i, _vp]
    L.rm_make_terrain_volume.argtypes = [_vp, _i, _i, _i, _vp]
    L.rm_voxelize_vertices.argtypes = [_vp, _vp, ctypes.c_longlong, _i, _i, _vp]
    L.rm_voxelize_scatter.argtypes = [_vp, _vp, ctypes.c_longlong, _i, ctypes.c_ulonglong, _vp]
    L.rm_make_heatmap_volume.argtypes = [_vp, _vp, _i, ctypes.c_double, _vp]
    L.rm_render_image.argtypes = [_vp, _vp, _vp, _vp, _i]
    L.rm_render_image_range.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i]
    L.rm_render_image_counted.argtypes = [_vp, _vp, _vp, _vp, _i, ctypes.POINTER(Counters)]
    L.rm_tonemap_image.argtypes = [_vp, _vp, _vp, _vp, _i]
    L.rm_render_frame.argtypes = [_vp, _vp, _vp, _i, _i, _vp, _vp]
    L.rm_tiles_per_part.argtypes = [_i, _i, _i]
    L.rm_set_sdf_volume.argtypes = [_vp, _vp, _i, _i, _i]
    L.rm_render_sdf_frame.argtypes = [_vp, _vp, _vp, _i, _i, _vp, _vp]
    L.rm_frame_device.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    L.rm_resolve_device.argtypes = [_vp, _vp, _i, _vp, _i, _i, _vp, _vp]
    L.rm_frame_device_argb.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]
    L.rm_resolve_device_argb.argtypes = [_vp, _vp, _i, _i, _i, _vp]
    L.rm_check_device_opts.argtypes = [_vp, _vp, _i, _i, _i]
    L.rm_last_frame_timing.argtypes = [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(_i)]
    L.rm_frame_timing_history.argtypes = [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(_i), _i, ctypes.POINTER(_i)]
    L.rm_selftest_prims.argtypes = [_vp, _i, _vp, _vp, _vp, _i]
    L.rm_selftest_filter.argtypes = [_vp, _vp, _vp, _i, _vp]
    L.rm_debug_get_accel.argtypes = [_vp, _i, _vp, _vp]
    L.rm_debug_get_octants.argtypes = [_vp, _i, _vp]
    L.rm_debug_volume_band.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.rm_debug_block_order.restype = ctypes.c_longlong
    L.rm_debug_block_order.argtypes = [_i] * 8 + [ctypes.c_double, ctypes.c_double, _vp, ctypes.c_longlong]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RmError(rc, lib().rm_last_error().decode("utf-8", "replace"))


def _np(a, dtype, name):
    if not isinstance(a, np.ndarray) or a.dtype != dtype or not a.flags.c_contiguous:
        raise TypeError(f"{name} must be a C-contiguous numpy array of {np.dtype(dtype).name}")
    return a.ctypes.data


# Arithmetic contract applied to every Context created without an explicit one; None = the
# library's own default (RM_CONTRACT_GFX950_DEFAULT).  Test modules that check the CPU-device contract
# against the CPU oracle set this to "cpu" for their duration.
DEFAULT_CONTRACT = None


class Context:
    """Owns one rm_ctx (one device, one stream)."""

    def __init__(self, device_id=0, contract=None):
        """device_id: a HIP ordinal, or a sequence of ordinals for one frame spread over several
        devices (rm_create_multi; ordinals may repeat).  contract: a key of CONTRACTS; None = "gfx950-default" (the library's)."""
        self._h = _vp()
        if isinstance(device_id, (list, tuple)):
            ids = (_i * len(device_id))(*[int(d) for d in device_id])
            check(lib().rm_create_multi(ids, len(device_id), ctypes.byref(self._h)))
        else:
            check(lib().rm_create(int(device_id), ctypes.byref(self._h)))
        self.device_id = device_id
        self.vres = None
        contract = contract or DEFAULT_CONTRACT
        if contract:
            self.set_contract(contract)

    @property
    def num_devices(self):
        return int(lib().rm_num_devices(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib().rm_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- volume -----------------------------------------------------------
    def set_volume(self, vox, vres):
        rx, ry, rz = (int(v) for v in vres)
        p = _np(vox, np.uint8, "vox")
        if vox.size != rx * ry * rz:
            raise ValueError(f"volume has {vox.size} bytes, vres says {rx}x{ry}x{rz}")
        check(lib().rm_set_volume(self._h, p, rx, ry, rz))
        self.vres = (rx, ry, rz)

    def set_volume_device(self, dptr, vres):
        rx, ry, rz = (int(v) for v in vres)
        check(lib().rm_set_volume_device(self._h, dptr, rx, ry, rz))
        self.vres = (rx, ry, rz)

    def stage_volume_device(self, dptr, vres, iso_val=32):
        """Enqueue the tables of the NEXT volume (device bytes, borrowed) on the library's own stream while the resident
        one renders; returns at once (rm_stage_volume_device)."""
        rx, ry, rz = (int(v) for v in vres)
        check(lib().rm_stage_volume_device(self._h, dptr, rx, ry, rz, int(iso_val)))
        self._staged_vres = (rx, ry, rz)

    def stage_volume(self, vox, vres, iso_val=32):
        """The same from host bytes (uint8 array): copied into a buffer of the library's, returns once they are taken."""
        rx, ry, rz = (int(v) for v in vres)
        v = np.ascontiguousarray(vox, dtype=np.uint8).reshape(-1)
        if v.size != rx * ry * rz:
            raise ValueError("volume size does not match vres")
        check(lib().rm_stage_volume(self._h, v.ctypes.data, rx, ry, rz, int(iso_val)))
        self._staged_vres = (rx, ry, rz)

    def commit_staged_volume(self):
        """The staged volume becomes the resident one for everything given to the context's stream from now on."""
        check(lib().rm_commit_staged_volume(self._h))
        self.vres = self._staged_vres

    def share_volume(self, src):
        """Use ``src``'s resident volume and the tables derived from it (same device)."""
        check(lib().rm_share_volume(self._h, src._h))
        self.vres = src.vres

    def invalidate_volume(self):
        """After an in-place edit of a borrowed device volume: derived tables are rebuilt."""
        check(lib().rm_invalidate_volume(self._h))

    def make_gyroid_volume(self, vres, want_host_copy=True):
        """Generate the gyroid benchmark volume on the device; it becomes the resident
        volume.  Returns the bytes (uint8) when want_host_copy."""
        rx, ry, rz = ((int(vres),) * 3 if isinstance(vres, (int, np.integer)) else tuple(int(v) for v in vres))
        out = np.zeros(rx * ry * rz, dtype=np.uint8) if want_host_copy else None
        check(lib().rm_make_gyroid_volume(self._h, rx, ry, rz, out.ctypes.data if want_host_copy else None))
        self.vres = (rx, ry, rz)
        return out

    def make_terrain_volume(self, vres, want_host_copy=True):
        """gen/make-terrain (generators.clj:44-60) on the device -> resident volume."""
        rx, ry, rz = ((int(vres),) * 3 if isinstance(vres, (int, np.integer)) else tuple(int(v) for v in vres))
        out = np.zeros(rx * ry * rz, dtype=np.uint8) if want_host_copy else None
        check(lib().rm_make_terrain_volume(self._h, rx, ry, rz, out.ctypes.data if want_host_copy else None))
        self.vres = (rx, ry, rz)
        return out

    def voxelize_vertices(self, vertices, res, ks=-1, want_host_copy=True):
        """meshvoxel.clj voxelize (ks < 0) / voxelize-ks on the device -> resident res^3 volume."""
        v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        res = int(res)
        out = np.zeros(res ** 3, dtype=np.uint8) if want_host_copy else None
        check(lib().rm_voxelize_vertices(self._h, v.ctypes.data if v.size else None, v.shape[0], res, int(ks),
                                         out.ctypes.data if want_host_copy else None))
        self.vres = (res, res, res)
        return out

    def voxelize_scatter(self, vertices, res, seed=0, want_host_copy=True):
        """meshvoxel.clj voxelize-scatter with seeded draws on the device -> resident res^3 volume."""
        v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        res = int(res)
        out = np.zeros(res ** 3, dtype=np.uint8) if want_host_copy else None
        check(lib().rm_voxelize_scatter(self._h, v.ctypes.data if v.size else None, v.shape[0], res,
                                        ctypes.c_ulonglong(int(seed) & 0xFFFFFFFFFFFFFFFF),
                                        out.ctypes.data if want_host_copy else None))
        self.vres = (res, res, res)
        return out

    def make_heatmap_volume(self, argb, amp, want_host_copy=True):
        """meshvoxel.clj make-heatmap on the device from a square ARGB image (uint32 [res, res])."""
        a = np.ascontiguousarray(argb, dtype=np.uint32)
        if a.ndim != 2 or a.shape[0] != a.shape[1]:
            raise ValueError("argb must be a square 2-D array")
        res = a.shape[0]
        out = np.zeros(res ** 3, dtype=np.uint8) if want_host_copy else None
        check(lib().rm_make_heatmap_volume(self._h, a.ctypes.data, res, float(amp),
                                           out.ctypes.data if want_host_copy else None))
        self.vres = (res, res, res)
        return out

    def set_stream(self, stream_ptr):
        """hipStream_t handle as int; 0 = the legacy default stream (torch's default),
        -1 = back to the context's own stream."""
        check(lib().rm_set_stream(self._h, _vp(stream_ptr & 0xFFFFFFFFFFFFFFFF)))

    def synchronize(self):
        check(lib().rm_synchronize(self._h))

    def pin_host_buffer(self, array):
        """Page-lock a long-lived numpy array that render_frame_into() will be given repeatedly."""
        check(lib().rm_pin_host_buffer(self._h, array.ctypes.data, array.nbytes))

    def unpin_host_buffer(self, array):
        check(lib().rm_unpin_host_buffer(self._h, array.ctypes.data))

    def render_frame_into(self, opts_array, mc_array, n, pixels, argb):
        """render_frame() into caller-owned arrays (e.g. pinned ones): the frame's outputs are not
        allocated per call (the 544-byte records are still marshalled per call)."""
        iters = len(bytes(opts_array)) // OPTS_BYTES
        n = int(n)
        if iters < 1 or np.asarray(mc_array).size != iters * TABLE_FLOATS:
            raise ValueError(f"mc_array must hold {iters} tables of {TABLE_FLOATS} floats")
        if pixels is not None and np.asarray(pixels).size < 4 * n:
            raise ValueError(f"pixels holds {np.asarray(pixels).size} floats, the frame needs {4 * n}")
        if argb is not None and np.asarray(argb).size < n:
            raise ValueError(f"argb holds {np.asarray(argb).size} words, the frame needs {n}")
        check(lib().rm_render_frame(self._h, self._opts(opts_array, iters), _np(mc_array, np.float32, "mc_array"), iters,
                                    n, _np(pixels, np.float32, "pixels") if pixels is not None else None,
                                    _np(argb, np.uint32, "argb") if argb is not None else None))

    def set_contract(self, contract):
        """"gfx950-default" (the library default): the results of the reference kernel as ROCm's OpenCL compiler
        builds it for this GPU with no options; "gfx950-strict" (= "gfx950"): built with -ffp-contract=off and
        correctly rounded divide/sqrt (each checked bit for bit against that build); "cpu": the results of an
        OpenCL CPU device on x86-64 (checked against the CPU oracle) -- include/raymarch_hip.h rm_set_contract."""
        check(lib().rm_set_contract(self._h, CONTRACTS[contract]))

    def set_seed_cast(self, mode):
        """"x86" (default): the undefined (uint) casts of the seed expressions as an OpenCL CPU
        device lowers them; "gpu": as GPU devices do (include/raymarch_hip.h rm_set_seed_cast)."""
        check(lib().rm_set_seed_cast(self._h, {"x86": 0, "gpu": 1}[mode]))

    # -- kernel-level entry points (host buffers) -------------------------
    @staticmethod
    def _opts(opts, count=1):
        b = bytes(opts)
        if len(b) != OPTS_BYTES * count:
            raise ValueError(f"opts must be {OPTS_BYTES * count} bytes, got {len(b)}")
        return ctypes.create_string_buffer(b, len(b))

    def render_image(self, mc, opts, pixels, n=None, id0=None, id1=None, counters=None):
        n = pixels.size // 4 if n is None else n
        pm = _np(mc, np.float32, "mc")
        if mc.size != TABLE_FLOATS:
            raise ValueError("mc must hold 0x4000 float4")
        pp = _np(pixels, np.float32, "pixels")
        if pixels.size < 4 * n:
            raise ValueError("pixels too small")
        ob = self._opts(opts)
        if counters is not None:
            check(lib().rm_render_image_counted(self._h, pm, ob, pp, n, ctypes.byref(counters)))
        elif id0 is None and id1 is None:
            check(lib().rm_render_image(self._h, pm, ob, pp, n))
        else:
            check(lib().rm_render_image_range(self._h, pm, ob, pp, n, id0 or 0,
                                              n if id1 is None else id1))
        return pixels

    def tonemap_image(self, pixels, opts, n=None):
        n = pixels.size // 4 if n is None else n
        argb = np.zeros(n, dtype=np.uint32)
        check(lib().rm_tonemap_image(self._h, _np(pixels, np.float32, "pixels"),
                                     self._opts(bytes(opts)[:OPTS_BYTES]), argb.ctypes.data, n))
        return argb

    # -- quality mode (not reference-equivalent; include/raymarch_hip.h) -----
    def set_sdf_volume(self, sdf, vres):
        rx, ry, rz = (int(v) for v in vres)
        a = np.ascontiguousarray(sdf, dtype=np.float32).reshape(-1)
        if a.size != rx * ry * rz:
            raise ValueError(f"distance field has {a.size} values, vres says {rx}x{ry}x{rz}")
        check(lib().rm_set_sdf_volume(self._h, a.ctypes.data, rx, ry, rz))

    def render_sdf_frame(self, opts_array, mc_array, n, want_pixels=True, want_argb=True):
        iters = len(bytes(opts_array)) // OPTS_BYTES
        mc = np.ascontiguousarray(mc_array, dtype=np.float32).reshape(-1)
        if mc.size != iters * TABLE_FLOATS:
            raise ValueError("mc_array must hold one table per pass")
        px = np.zeros(4 * n, dtype=np.float32) if want_pixels else None
        argb = np.zeros(n, dtype=np.uint32) if want_argb else None
        check(lib().rm_render_sdf_frame(self._h, self._opts(opts_array, iters), mc.ctypes.data, iters, n,
                                        px.ctypes.data if want_pixels else None,
                                        argb.ctypes.data if want_argb else None))
        return px, argb

    def render_frame(self, opts_array, mc_array, n, want_pixels=True, want_argb=True):
        iters = len(bytes(opts_array)) // OPTS_BYTES
        mc_array = np.ascontiguousarray(mc_array, dtype=np.float32).reshape(-1)
        if mc_array.size != iters * TABLE_FLOATS:
            raise ValueError("mc_array must hold one table per pass")
        pixels = np.zeros(4 * n, dtype=np.float32) if want_pixels else None
        argb = np.zeros(n, dtype=np.uint32) if want_argb else None
        check(lib().rm_render_frame(self._h, self._opts(opts_array, iters), mc_array.ctypes.data,
                                    iters, n,
                                    pixels.ctypes.data if want_pixels else None,
                                    argb.ctypes.data if want_argb else None))
        return pixels, argb

    # -- device-resident pipeline ----------------------------------------
    def frame_device(self, d_opts, d_mc, iters, n, width, d_tiles, tile_first=0, tile_stride=1):
        """All passes of partition (tile_first, tile_stride) into its tile-major
        accumulators (device pointers as ints); asynchronous."""
        check(lib().rm_frame_device(self._h, d_opts, d_mc, iters, n, width, tile_first, tile_stride,
                                    d_tiles))

    def frame_device_argb(self, d_opts, d_mc, iters, n, width, d_tiles, d_argb_tiles, tile_first=0, tile_stride=1):
        """frame_device() that also leaves the partition's TonemapImage words, tile-major, in d_argb_tiles
        (the 4-byte-per-pixel exchange unit of frames that want the ARGB image only)."""
        check(lib().rm_frame_device_argb(self._h, d_opts, d_mc, iters, n, width, tile_first, tile_stride,
                                         d_tiles, d_argb_tiles))

    def resolve_device_argb(self, d_argb_tiles_all, parts, n, width, d_argb):
        check(lib().rm_resolve_device_argb(self._h, d_argb_tiles_all, parts, n, width, d_argb))

    def frame_device_full(self, d_opts, d_mc, iters, n, width, d_pixels=None, d_argb=None):
        """The unpartitioned frame, one launch per group of 16 passes (a run of 20-31 passes: one launch): row-major pixels and / or ARGB; asynchronous."""
        check(lib().rm_frame_device_full(self._h, d_opts, d_mc, iters, n, width, d_pixels, d_argb))

    def last_frame_breakdown(self):
        """-> ([share_ms per device], frame_ms): rm_last_frame_breakdown."""
        k = self.num_devices
        shares = (ctypes.c_float * k)()
        total = ctypes.c_float()
        check(lib().rm_last_frame_breakdown(self._h, shares, k, ctypes.byref(total)))
        return [float(v) for v in shares], float(total.value)

    def last_table_build_ms(self):
        ms = ctypes.c_float()
        check(lib().rm_last_table_build_ms(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def resolve_device(self, d_tiles_all, parts, d_opts, n, width, d_pixels=None, d_argb=None):
        check(lib().rm_resolve_device(self._h, d_tiles_all, parts, d_opts, n, width, d_pixels, d_argb))

    def check_device_opts(self, d_opts, iters, n, width):
        check(lib().rm_check_device_opts(self._h, d_opts, iters, n, width))

    def last_frame_timing(self):
        ms = ctypes.c_float()
        k = _i()
        check(lib().rm_last_frame_timing(self._h, ctypes.byref(ms), ctypes.byref(k)))
        return float(ms.value), int(k.value)

    def frame_timing_history(self, max_frames=32):
        """-> [(ms, launches)] of the last frames (oldest first): device time of their render kernels."""
        ms = (ctypes.c_float * max_frames)()
        ln = (_i * max_frames)()
        k = _i()
        check(lib().rm_frame_timing_history(self._h, ms, ln, max_frames, ctypes.byref(k)))
        return [(float(ms[i]), int(ln[i])) for i in range(k.value)]

    def debug_get_accel(self, iso):
        nvox = int(np.prod(self.vres))
        dist = np.zeros(nvox, dtype=np.uint8)
        surf = np.zeros(nvox, dtype=np.uint32)
        check(lib().rm_debug_get_accel(self._h, iso, dist.ctypes.data, surf.ctypes.data))
        return dist, surf

    def debug_get_octants(self, iso):
        """-> uint8 [8, rz, ry, rx]: the directional tables of the resident volume."""
        rx, ry, rz = self.vres
        out = np.zeros(8 * rx * ry * rz, dtype=np.uint8)
        check(lib().rm_debug_get_octants(self._h, iso, out.ctypes.data))
        return out.reshape(8, rz, ry, rx)

    def selftest_filter(self, opts, rays):
        """rays: float32 [n, 8] (origin, direction, t, g) -> uint32 [n] of decision bits (header)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], dtype=np.uint32)
        check(lib().rm_selftest_filter(self._h, self._opts(bytes(opts)[:OPTS_BYTES]), rays.ctypes.data,
                                       rays.shape[0], out.ctypes.data))
        return out

    def selftest_prims(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.zeros(a.size, dtype=np.uint32)
        pb = None
        if b is not None:
            b = np.ascontiguousarray(b, dtype=np.float32)
            pb = b.ctypes.data
        check(lib().rm_selftest_prims(self._h, op, a.ctypes.data, pb, out.ctypes.data, a.size))
        return out


def device_count():
    return int(lib().rm_device_count())


def tiles_per_part(width, n, parts):
    r = int(lib().rm_tiles_per_part(width, n, parts))
    if r < 0:
        check(r)
    return r
