"""raymarchcl_amd -- MI355X-native replacement for the render path of
thi-ng/raymarchcl (RenderImage / TonemapImage and the pass pipeline that drives
them).  See DESIGN.md; the device code is hand-written HIP behind the C ABI in
include/raymarch_hip.h and there is NO CPU fallback: anything that renders
raises if the HIP library or a GPU is missing."""
from . import generators, materials, options, structs, vio  # noqa: F401
from .options import compute_eyepos, render_options  # noqa: F401

__all__ = ["generators", "materials", "options", "structs", "vio", "render_options", "compute_eyepos"]
