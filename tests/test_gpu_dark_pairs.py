"""The shadow marches that are not traced (lighting_wave, RM_DARK_SKIP): a (hit, light) pair whose
diffuse and specular factors are exact zeros does not need its shadow term -- provided the zero
products it adds cannot flip a -0 in the running sums.  The library decides that from the signs of
sky*ao, reflectCol*ao, albedo and lightColor*att; these frames put negative, zero and huge values
into exactly those inputs (and use 1..4 lights, some failing the attenuation test), where a wrong
decision shows up as a sign or NaN difference.  Whole frames, bit for bit against the oracle."""
import numpy as np
import pytest

import scenes

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cpu_contract")]  # this module checks against the CPU oracle

CASES = {
    "four lights, two behind everything": dict(
        numLights=4, lightPos=[[-2, 0, -2, 0], [2, 0, 2, 0], [0, -0.9, 0, 0], [0, 3, 0, 0]],
        lightColor=[[50, 50, 50, 0], [30, 40, 50, 0], [20, 5, 5, 0], [60, 60, 60, 0]]),
    "negative light colour": dict(
        numLights=2, lightColor=[[-50, 50, -0.0, 0], [50, -20, 50, 0]]),
    "negative and zero albedo": dict(
        materials=[dict(albedo=[1, 1, 1, 1], r0=0.2, smoothness=0.5),
                   dict(albedo=[-0.5, 0.0, 0.9, 1], r0=0.1, smoothness=0.9),
                   dict(albedo=[0.0, -0.0, -1.0, 1], r0=0.0, smoothness=0.1),
                   dict(albedo=[0.9, 0.2, -0.1, 1], r0=0.3, smoothness=0.3)]),
    "black and negative sky": dict(skyColor1=[0.0, -0.0, -1.0], skyColor2=[0.0, 0.0, 0.5]),
    "occlusion below zero": dict(aoAmp=3.0, aoStepDist=0.02),
    "attenuation test fails for far lights": dict(
        numLights=3, minLightAtt=0.12, lightPos=[[-2, 0, -2, 0], [2, 0, 2, 0], [0.5, 0.2, 0.5, 0]],
        lightColor=[[50, 50, 50, 0], [50, 50, 50, 0], [5, 5, 5, 0]]),
    "huge light colour": dict(lightColor=[[3e38, 1e30, 50, 0], [3e38, 3e38, 3e38, 0]]),
    "light inside the surface band": dict(lightScatter=0.0, lightPos=[[0.0, 0.0, 0.0, 0], [0.3, -0.2, 0.1, 0]]),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("mat", ["orange-stripes", "metal"])
def test_unlit_pairs_with_signed_inputs(native, oracle_mod, name, mat):
    import raymarchcl_amd as rm
    from raymarchcl_amd import generators as gen, structs

    w, h, it = 48, 40, 2
    vox = scenes.volume("gyroid", 64)
    recs = []
    for i in range(it):
        o = rm.render_options(width=w, height=h, vres=[64] * 3, t=i * 0.333, iter=it, mat=mat,
                              eyepos=rm.compute_eyepos(-45 + 90 * (name > "h"), 2.2, 0.4), targetpos=[0, -0.3, 0])
        o.update(CASES[name])
        recs.append(structs.encode_bytes(o))
    opts = b"".join(recs)
    mc = np.stack([gen.generate_scatter_offsets(0x4000, seed=4000 + i) for i in range(it)])
    n = w * h
    mask = np.zeros(n, np.uint8)
    want = np.zeros(4 * n, np.float32)
    for i in range(it):
        oracle_mod.render_image(vox, mc[i], opts[i * 544:(i + 1) * 544], want, n=n, undefined_mask=mask)
    with native.Context(0) as ctx:
        ctx.set_volume(vox, (64, 64, 64))
        px, _ = ctx.render_frame(opts, mc, n)
    ok = np.repeat(mask == 0, 4)
    a, b = px.view(np.uint32)[ok], want.view(np.uint32)[ok]
    nan = np.isnan(want[ok])
    assert np.array_equal(a[~nan], b[~nan]), int((a[~nan] != b[~nan]).sum())
    assert np.isnan(px[ok][nan]).all()
    assert ok.sum() > 2 * n  # most of the frame is defined
