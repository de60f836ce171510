"""Host render driver -- the Python mirror of the reference's Clojure host
(/root/reference/src/thi/ng/raymarchcl/core.clj).  Same function names,
argument meaning and call order; the OpenCL plumbing (thi.ng.simplecl) is
replaced by the C ABI of libraymarch_hip.so.

    render_options               core.clj:28-74   (options.py)
    make_render_option_buffer    core.clj:99-106
    update_render_option_buffer  core.clj:108-117
    init_renderer                core.clj:119-148
    make_pipeline                core.clj:76-97
    execute_pipeline             thi.ng.simplecl.ops/execute-pipeline (core.clj:171)
    compute_eyepos               core.clj:150-152 (options.py)
    test_render / test_anim      core.clj:154-213

Differences that are deliberate:
  * the scatter tables are seeded (``mc_seed``), the reference seeds them from
    the wall clock (generators.clj:10);
  * ``init_renderer`` accepts the volume as bytes (``voxels=``) as well as a
    ``.vox`` path (``vname=``); the reference only loads files (core.clj:146);
  * no CPU fallback: everything that renders needs the HIP library + a GPU.
"""
import os

import numpy as np

from . import generators as gen
from . import structs
from . import vio
from .options import compute_eyepos, render_options

__all__ = [
    "render_options", "compute_eyepos", "make_render_option_buffer",
    "update_render_option_buffer", "init_renderer", "make_pipeline", "execute_pipeline",
    "test_render", "test_anim", "argb_to_rgb8", "save_png",
]


def make_render_option_buffer(n, opts, time_step=0.333):
    """``n`` encoded TRenderOpts records, pass i at ``t = i * 0.333``
    (core.clj:99-106).  -> list of 544-byte ``bytes``."""
    return [structs.encode_bytes(render_options(**{**opts, "t": i * time_step})) for i in range(n)]


def update_render_option_buffer(buffers, opts):
    """Re-encode in place for the next animation frame; note the reference's
    0.3333 here versus 0.333 at creation (core.clj:116 vs :105)."""
    fresh = make_render_option_buffer(len(buffers), opts, time_step=0.3333)
    buffers[:] = fresh
    return buffers


def init_renderer(width, height, vres, iter=1, vname=None, voxels=None, mc_seed=1000, device=0,
                  contract="gfx950-default", **args):
    """Build the render state: option records, one scatter table per pass, the
    device context with the volume resident in HBM, and the pipeline
    (core.clj:119-148)."""
    from . import _native

    opts = dict(args, width=width, height=height, vres=vres, iter=iter)
    if voxels is None:
        voxels, file_res = vio.load_volume(vname or "gyroid-sliced-512-s0.01.vox")
        vres3 = file_res
    else:
        vres3 = (vres,) * 3 if isinstance(vres, int) else tuple(vres)
        voxels = np.ascontiguousarray(voxels).view(np.uint8).reshape(-1)
    ctx = _native.Context(device, contract=contract)
    ctx.set_volume(voxels, vres3)
    state = {
        "ctx": ctx,
        "args": opts,
        "opts-buffers": make_render_option_buffer(iter, opts),
        "mc-buffers": [gen.generate_scatter_offsets(0x4000, seed=mc_seed + i) for i in range(iter)],
        "num": width * height,
        "width": width,
        "height": height,
    }
    state["pipeline"] = make_pipeline(state)
    return state


def make_pipeline(state):
    """The declarative step list of core.clj:76-97: write p/v; per pass write
    (opts_i, mc_i) and run RenderImage; run TonemapImage with opts_0; read q."""
    steps = [{"write": ["p-buf", "v-buf"]}]
    for i in range(len(state["opts-buffers"])):
        steps.append({"write": [("opts", i), ("mc", i)]})
        steps.append({"name": "RenderImage", "pass": i, "n": state["num"]})
    steps.append({"write": ["q-buf"]})
    steps.append({"name": "TonemapImage", "opts": 0, "n": state["num"], "read": ["out"]})
    return {"state": state, "steps": steps}


def execute_pipeline(pipeline, final_size=None, want_pixels=False):
    """Run the pipeline on the device; returns the packed ARGB uint32 array
    (what the reference reads back from q-buf), and the float accumulator too
    when ``want_pixels``."""
    st = pipeline["state"]
    n = st["num"] if final_size is None else final_size
    opts = b"".join(st["opts-buffers"])
    mcs = np.concatenate(st["mc-buffers"])
    pixels, argb = st["ctx"].render_frame(opts, mcs, n, want_pixels=want_pixels, want_argb=True)
    return (argb, pixels) if want_pixels else argb


def argb_to_rgb8(argb, width, height):
    a = np.asarray(argb, dtype=np.uint32).reshape(height, width)
    return np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], axis=-1).astype(np.uint8)


def save_png(argb, width, height, path):
    from PIL import Image  # optional dependency, only for file export

    Image.fromarray(argb_to_rgb8(argb, width, height)).save(path)


def test_render(width=640, height=360, iter=1, vres=256, mat="metal", vname=None,
                out_path="foo.png", theta=135, dist=2.25, **opts):
    """One frame to a PNG (core.clj:154-179).  Extra keys (dof, fov, gamma,
    groundY, voxelSize, targetpos, voxels, mc_seed, device, contract) are forwarded."""
    args = dict(width=width, height=height, vres=vres, iter=iter,
                eyepos=compute_eyepos(theta, dist, 0.35), targetpos=[0, -0.4, 0], mat=mat,
                vname=vname)
    args.update(opts)
    state = init_renderer(**args)
    argb = execute_pipeline(state["pipeline"], final_size=state["num"])
    if out_path:
        save_png(argb, width, height, out_path)
    state["ctx"].close()
    return argb


def test_anim(width, height, iter, res, mat, vname=None, out_dir="export", frames=35, **extra):
    """35-frame turntable (core.clj:181-213)."""
    args = dict(width=width, height=height, vres=[res, res, res], iter=iter, mat=mat, vname=vname)
    args.update(extra)
    state = init_renderer(**args)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    out = []

    def lerp(t, a, b):  # m/map-interval t 0 1 a b
        return a + (b - a) * t

    for frame in range(frames):
        t = frame / 35.0
        theta = lerp(t, 0, 350)
        frame_args = dict(state["args"], fov=lerp(t, 115, 115), targetpos=[0, lerp(t, -0.15, -0.15), 0],
                          eyepos=compute_eyepos(theta, lerp(t, 2.25, 2.25), lerp(t, 0.44, 0.45)))
        frame_args = {k: v for k, v in frame_args.items() if k not in ("vname", "voxels", "mc_seed", "device", "contract")}
        update_render_option_buffer(state["opts-buffers"], frame_args)
        argb = execute_pipeline(make_pipeline(state), final_size=state["num"])
        if out_dir:
            save_png(argb, width, height, os.path.join(out_dir, "frame-%04d.png" % frame))
        out.append(argb)
    state["ctx"].close()
    return out
