"""The CPU restatement (oracle/rm_restate.c) against the golden fixtures that
were generated from the reference kernel itself (tests/golden/make_golden.py).
Bit-exact: this is integer-exact float32 arithmetic with a fixed op order."""
import numpy as np
import pytest

import scenes
from conftest import load_golden

NAMES = list(scenes.SCENES)


@pytest.mark.parametrize("name", NAMES)
def test_restatement_matches_reference_fixture(oracle_mod, name):
    g = load_golden(name)
    sc = scenes.build(name)
    # the fixture's inputs are what the seeds regenerate
    assert np.array_equal(g["vox"], sc["vox"])
    assert g["opts"].tobytes() == sc["opts"]
    assert [str(s) for s in g["mc_sha"]] == [scenes.sha(sc["mc"][i]) for i in range(sc["iter"])]
    n = int(g["n"])
    st = oracle_mod.Stats()
    px, argb = oracle_mod.render_frame(g["vox"], g["opts"].tobytes(), sc["mc"], n, stats=st)
    assert st.oob_material == 0, "scene reaches behaviour that is undefined in the reference"
    assert np.array_equal(px.view(np.uint32), g["pixels"].view(np.uint32))
    assert np.array_equal(argb, g["argb"])
    assert not np.isnan(px).any()


def test_fixture_carries_its_table(oracle_mod):
    g = load_golden("c1_orange")
    sc = scenes.build("c1_orange")
    assert np.array_equal(g["mc_full"].view(np.uint32), sc["mc"].view(np.uint32))


def test_config1_full_frame_hash(oracle_mod):
    """BASELINE config 1 (64^3 gyroid, 256x256, 1 spp) at full size."""
    g = load_golden("c1_full")
    sc = scenes.build(dict(scenes.SCENES["c1_orange"], w=256, h=256))
    assert sc["opts"] == g["opts"].tobytes()
    assert scenes.sha(sc["vox"]) == str(g["vox_sha"])
    px, argb = oracle_mod.render_frame(sc["vox"], sc["opts"], sc["mc"], sc["n"])
    mask = np.zeros(sc["n"], np.uint8)
    oracle_mod.render_image(sc["vox"], sc["mc"][0].copy(), sc["opts"][:544],
                            np.zeros(4 * sc["n"], np.float32), undefined_mask=mask)
    undefined = np.nonzero(mask)[0]
    assert undefined.tolist() == g["undefined_ids"].tolist() and undefined.size <= 2
    px.reshape(-1, 4)[undefined] = 0
    argb[undefined] = 0
    assert scenes.sha(px) == str(g["pixels_sha"])
    assert scenes.sha(argb) == str(g["argb_sha"])
    rows = g["rows"]
    assert np.array_equal(px.reshape(256, 256, 4)[rows].view(np.uint32),
                          g["pixels_rows"].view(np.uint32))


def test_id_ranges_compose(oracle_mod):
    """Work-items are independent: rendering [0,a) then [a,n) equals [0,n)."""
    sc = scenes.build("c1_orange")
    n = sc["n"]
    full = np.zeros(4 * n, np.float32)
    oracle_mod.render_image(sc["vox"], sc["mc"][0].copy(), sc["opts"][:544], full)
    parts = np.zeros(4 * n, np.float32)
    for a, b in ((0, 1000), (1000, 1001), (1001, n)):
        oracle_mod.render_image(sc["vox"], sc["mc"][0].copy(), sc["opts"][:544], parts, id0=a, id1=b)
    assert np.array_equal(full.view(np.uint32), parts.view(np.uint32))


def test_frame_blend_recurrence(oracle_mod):
    """Multi-pass accumulation is p <- p + (c_k - p)/iter, not a mean."""
    sc = scenes.build("orange_dof_2spp")
    n, it = sc["n"], sc["iter"]
    px, _ = oracle_mod.render_frame(sc["vox"], sc["opts"], sc["mc"], n, tonemap=False)
    acc = np.zeros(4 * n, np.float32)
    fb = np.float32(1.0 / it)
    for i in range(it):
        one = np.zeros(4 * n, np.float32)
        o = bytearray(sc["opts"][i * 544:(i + 1) * 544])
        o[268:272] = np.float32(1.0).tobytes()  # frameBlend = 1 -> the pass colour itself
        oracle_mod.render_image(sc["vox"], sc["mc"][i].copy(), bytes(o), one)
        acc = (acc + (one - acc) * fb).astype(np.float32)
    acc.reshape(-1, 4)[:, 3] = 1.0
    assert np.array_equal(acc.view(np.uint32), px.view(np.uint32))
