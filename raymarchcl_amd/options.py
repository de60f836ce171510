"""Scene / camera / march / shading parameters of one render pass.

Restates ``render-options`` (reference core.clj:28-74): a fixed default map,
a handful of call-site keys, and the material preset merged LAST -- so the
preset-owned keys (lightColor, lightPos, materials, numLights, aoAmp,
reflectIter) cannot be overridden from the call site, exactly as in the
reference.  ``compute_eyepos`` restates core.clj:150-152.
"""
import math

from . import materials

_RAD = math.pi / 180.0  # thi.ng.math RAD


def radians(deg):
    return deg * _RAD


def compute_eyepos(theta, dist, y):
    """``(g/rotate-y (vec3 0 y dist) (radians theta))`` -- thi.ng.geom rotates
    (x, z) to (x cos + z sin, z cos - x sin)."""
    t = radians(theta)
    s, c = math.sin(t), math.cos(t)
    x, z = 0.0, float(dist)
    return [x * c + z * s, float(y), z * c - x * s]


def render_options(
    width,
    height,
    vres,
    t=0.0,
    iter=1,
    eyepos=None,
    mat=None,
    fov=None,
    dof=None,
    targetpos=None,
    gamma=None,
    groundY=None,
    voxelSize=None,
    **_ignored,
):
    """-> dict with one entry per TRenderOpts field that the reference sets.

    Unknown keyword arguments are accepted and ignored: the reference
    destructures only the keys above out of a larger map (core.clj:29).
    """
    eps = 0.005
    clip = 0.99
    if isinstance(vres, int):
        vres = [vres, vres, vres]
    vres = [int(v) for v in vres]

    def _or(v, default):  # Clojure `or`: nil/false -> default (0 is truthy)
        return default if v is None or v is False else v

    opts = {
        "aoAmp": 0.2,
        "aoIter": 5,
        "aoStepDist": 0.05,
        "dof": _or(dof, 0.001),
        "eps": eps,
        "exposure": 3.5,
        "eyePos": _or(eyepos, [2, 0, 2]),
        "flareAmp": 0.015,
        "fogPow": 0.05,
        "fov": radians(_or(fov, 90)),
        "frameBlend": 1.0 / iter,
        "gamma": _or(gamma, 1.5),
        "groundY": _or(groundY, 1.05),
        "invAspect": height / width,
        "invVoxelScale": [0.5, 0.5, 0.5],
        "isoVal": 32,
        "lightColor": [50, 50, 50],
        "lightPos": [[-2, 0, -2, 0], [2, 0, 2, 0]],
        "lightScatter": 0.2,
        "maxDist": 30,
        "maxIter": 128,
        "maxVoxelIter": 192,
        "minLightAtt": 0.0,
        "numLights": 2,
        "reflectIter": 0,
        "resolution": [int(width), int(height)],
        "shadowBias": 0.1,
        "shadowIter": 128,
        "skyColor1": [1.8, 1.8, 1.9],
        "skyColor2": [0.1, 0.1, 0.1],
        "startDist": 0.0,
        "targetPos": _or(targetpos, [0, -0.15, 0]),
        "time": t,
        "up": [0, 1, 0],
        "voxelBounds": [1, 1, 1],
        "voxelBounds2": [2, 2, 2],
        "voxelBoundsMax": [clip, clip, clip],
        "voxelBoundsMin": [-clip, -clip, -clip],
        "voxelRes": vres + [vres[0] * vres[1]],
        "voxelSize": _or(voxelSize, 1.0 / vres[0]),
    }
    opts.update(materials.lookup(mat))
    return opts
