"""Material / light presets.

Data restated from the reference's preset map
(/root/reference/src/thi/ng/raymarchcl/materials.clj:3-76).  Each preset
overrides ``lightColor``, optionally ``lightPos``, ``materials[4]``,
``numLights``, ``aoAmp`` and ``reflectIter`` of the render options; material
slot 0 is the ground plane, 1..3 are voxel value bands (<84, <168, >=168).
"""


def _mat(albedo, r0, smoothness):
    return {"albedo": list(albedo), "r0": r0, "smoothness": smoothness}


presets = {
    "orange-stripes": {
        "lightColor": [[28, 18, 8, 0], [8, 18, 28, 0]],
        "lightPos": [[-2, 0, -2, 0], [2, 0, 2, 0]],
        "materials": [
            _mat([1.0, 1.0, 1.0, 1.0], 0.1, 0.9),
            _mat([4.9, 0.9, 0.05, 1.0], 0.01, 0.5),
            _mat([1.9, 1.9, 1.9, 1.0], 0.01, 0.4),
            _mat([0.9, 0.9, 0.9, 1.0], 0.8, 0.1),
        ],
        "numLights": 2,
        "aoAmp": 0.25,
        "reflectIter": 1,
    },
    "metal": {
        "lightColor": [[28, 18, 8, 0], [16, 36, 56, 0]],
        "lightPos": [[0, 2, 0, 0], [3, 0, 3, 0]],
        "materials": [
            _mat([0.01, 0.01, 0.01, 1.0], 0.1, 0.5),
            _mat([1.9, 1.9, 1.9, 1.0], 0.1, 0.5),
            _mat([0.25, 0.27, 0.5, 1.0], 0.7, 0.1),
            _mat([1.0, 1.0, 1.0, 1.0], 0.2, 0.1),
        ],
        "numLights": 2,
        "aoAmp": 0.25,
        "reflectIter": 3,
    },
    "metal2": {
        "lightColor": [[28, 18, 8, 0], [8, 18, 28, 0]],
        "lightPos": [[-2, 0, -2, 0], [2, 0, 2, 0]],
        "materials": [
            _mat([0.0, 0.0, 0.0, 1.0], 0.1, 0.9),
            _mat([1.0, 1.01, 1.075, 1.0], 0.4, 0.7),
            _mat([1.9, 1.9, 1.9, 1.0], 0.4, 0.5),
            _mat([0.9, 0.9, 0.9, 1.0], 0.75, 0.2),
        ],
        "numLights": 2,
        "aoAmp": 0.25,
        "reflectIter": 3,
    },
    "ao": {
        "lightColor": [[50, 50, 50, 0]],
        "materials": [_mat([1.0, 1.0, 1.0, 1.0], 0.0, 1.0) for _ in range(4)],
        "numLights": 1,
        "aoAmp": 0.25,
        "reflectIter": 0,
    },
}


def lookup(mat):
    """``(get presets mat (presets :ao))`` -- unknown names fall back to ``ao``
    (core.clj:74).  Accepts ``"metal"`` or the keyword spelling ``":metal"``."""
    if isinstance(mat, str) and mat.startswith(":"):
        mat = mat[1:]
    return presets.get(mat, presets["ao"])
