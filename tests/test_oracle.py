"""Pins the CPU oracle (oracle/vpt_oracle.cpp): bit-identity with float32 states produced by the
reference's own renderer (committed fixtures; live runs too where oracle/_ref exists), and the
instructor image check/lowres/03_volume through the stb-compatible JPEG stage."""
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN

CASES = {  # name -> (shader, resolution, samples, bounces); see tests/golden/make_fixtures.py
    "vol_64_1": ("volpathtrace", 64, 1, 64), "vol_64_4": ("volpathtrace", 64, 4, 64),
    "vol_96_16": ("volpathtrace", 96, 16, 64), "path_64_4": ("pathtrace", 64, 4, 4),
    "naive_64_4": ("naive", 64, 4, 4), "eye_64_2": ("eyelight", 64, 2, 4), "normal_64_2": ("normal", 64, 2, 4),
    "texcoord_64_2": ("texcoord", 64, 2, 4), "color_64_2": ("color", 64, 2, 4),
}


@pytest.fixture(scope="module")
def golden_states():
    return np.load(os.path.join(GOLDEN, "03_volume_states.npz"))


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_bit_identical_to_reference_states(vpt, scene03, oracle, golden_states, name):
    shader, res, spp, bounces = CASES[name]
    p = vpt.PathtraceParams(resolution=res, samples=spp, shader=shader, bounces=bounces)
    st = scene03.make_state(p)
    oracle.oracle_render(scene03, p, st, spp, nthreads=4)
    ref_image, ref_rngs = golden_states[name + "_image"], golden_states[name + "_rngs"]
    assert st.samples == spp and (st.hits == spp).all()
    assert np.array_equal(st.rngs, ref_rngs), "RNG streams diverged: draw order differs from the reference"
    assert np.array_equal(st.image.view(np.uint32), ref_image.view(np.uint32)), "float32 radiance sums differ"


def test_oracle_is_schedule_independent_and_resumable(vpt, scene03, oracle):
    """1 thread vs 8 threads, and 4 passes at once vs 1+3: identical (pixels own their streams)."""
    p = vpt.PathtraceParams(resolution=64, samples=4, shader="volpathtrace", bounces=64)
    a, b, c = scene03.make_state(p), scene03.make_state(p), scene03.make_state(p)
    oracle.oracle_render(scene03, p, a, 4, nthreads=1)
    oracle.oracle_render(scene03, p, b, 4, nthreads=8)
    oracle.oracle_render(scene03, p, c, 1, nthreads=3)
    oracle.oracle_render(scene03, p, c, 3, nthreads=5)
    for x in (b, c):
        assert np.array_equal(a.image.view(np.uint32), x.image.view(np.uint32)) and np.array_equal(a.rngs, x.rngs)
    # no-op once state.samples >= params.samples (yocto_pathtrace.cpp:1055)
    before = a.image.copy()
    oracle.oracle_render(scene03, p, a, 2)
    assert a.samples == 4 and np.array_equal(before, a.image)


def test_oracle_live_against_reference_binary(vpt, scene03, oracle):
    """Where the reference build exists (the build container), compare a larger live run."""
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/ref_driver not built here; covered by the committed fixtures")
    from conftest import SCENE_03
    w, h, image, hits, rngs, _ = oracle.reference_render(SCENE_03, "volpathtrace", 200, 6, 64)
    p = vpt.PathtraceParams(resolution=200, samples=6, shader="volpathtrace", bounces=64)
    st = scene03.make_state(p)
    oracle.oracle_render(scene03, p, st, 6)
    assert (st.width, st.height) == (w, h)
    assert np.array_equal(st.rngs, rngs) and np.array_equal(st.image.view(np.uint32), image.view(np.uint32))


def test_oracle_counters_match_survey_event_profile(vpt, scene03, oracle):
    """Per-sample event counts feed the roofline's algorithmic-bytes figure (13.6 scene
    nodes, 89.2 shape nodes, 20.6 instance tests, 14.8 quad tests per sample on 03_volume)."""
    p = vpt.PathtraceParams(resolution=160, samples=1 << 20, shader="volpathtrace", bounces=64)
    st = scene03.make_state(p)
    c = oracle.oracle_render(scene03, p, st, 8, counters=True)
    n = c["samples"]
    assert n == st.width * st.height * 8
    assert 11 < c["scene_nodes"] / n < 16 and 75 < c["shape_nodes"] / n < 105
    assert 17 < c["instance_tests"] / n < 24 and 12 < c["quad_tests"] / n < 18 and c["tri_tests"] == 0


def test_oracle_reproduces_instructor_image_lowres(vpt, scene03, oracle):
    """check/lowres/03_volume_720_256.jpg (scripts/run.sh:3) through get_render -> sRGB8 -> JPEG q75.
    The reference itself (g++/glibc) is at per-channel RMS 6.3-6.9e-4 of this MSVC-built image
    (SURVEY §6); independent noise would be 0.059."""
    from PIL import Image
    p = vpt.PathtraceParams(resolution=720, samples=256, shader="volpathtrace", bounces=64)
    st = scene03.make_state(p)
    oracle.oracle_render(scene03, p, st, 256)
    jpg = vpt.encode_jpeg_q75(vpt.linear_to_srgb8(st.image, st.samples))
    mine = np.asarray(Image.open(io.BytesIO(jpg)).convert("RGB"), np.float32) / 255
    check = np.asarray(Image.open(os.path.join(GOLDEN, "check", "03_volume_720_256.jpg")).convert("RGB"), np.float32) / 255
    assert mine.shape == check.shape == (300, 720, 3)
    rms = np.sqrt(np.mean((mine - check) ** 2, axis=(0, 1)))
    assert (rms < 1.5e-3).all(), rms


# ---- substitute scenes (tests/golden/make_scenes.py): glossy + normal maps (config 1), 144k-triangle mesh with
# ---- two environments and rough subsurface refraction (config 3), voxel-SDF + analytic SDFs + SDF light (config 4)
EXTRA = {  # name -> (scene, shader, resolution, samples, bounces, noimplicit_mis)
    "surf_path_96_4": ("01_surface_min/surface_min.json", "pathtrace", 96, 4, 4, False),
    "surf_normal_96_1": ("01_surface_min/surface_min.json", "normal", 96, 2, 4, False),
    "surf_eye_96_2": ("01_surface_min/surface_min.json", "eyelight", 96, 2, 4, False),
    "head_vol_96_4": ("05_head1ss_sub/head1ss_sub.json", "volpathtrace", 96, 4, 64, False),
    "sdf_implicit_96_4": ("06_gridsdf_synth/gridsdf_synth.json", "implicit", 96, 4, 4, False),
    "sdf_nomis_96_4": ("06_gridsdf_synth/gridsdf_synth.json", "implicit", 96, 4, 4, True),
    "sdf_normal_96_2": ("06_gridsdf_synth/gridsdf_synth.json", "implicit_normal", 96, 2, 4, False),
}


@pytest.fixture(scope="module")
def substitute_states():
    return np.load(os.path.join(GOLDEN, "substitute_states.npz"))


@pytest.mark.parametrize("name", sorted(EXTRA))
def test_oracle_bit_identical_on_substitute_scenes(vpt, oracle, substitute_states, name):
    scene_file, shader, res, spp, bounces, nomis = EXTRA[name]
    scene = vpt.HostScene(os.path.join(GOLDEN, "scenes", scene_file))
    p = vpt.PathtraceParams(resolution=res, samples=spp, shader=shader, bounces=bounces, noimplicit_mis=nomis)
    st = scene.make_state(p)
    oracle.oracle_render(scene, p, st, spp, nthreads=8)
    assert np.array_equal(st.rngs, substitute_states[name + "_rngs"])
    assert np.array_equal(st.image.view(np.uint32), substitute_states[name + "_image"].view(np.uint32))


def test_host_pipeline_matches_reference_on_substitute_scenes(vpt):
    """loader (PLY triangles, binary + text .sdf, shared HDR), BVH build on 144k triangles, light CDFs of two
    environments / an SDF light: FNV hashes equal the reference's."""
    import json
    golden = json.load(open(os.path.join(GOLDEN, "substitute_stats.json")))
    for scene_file, stats in golden.items():
        mine = json.loads(vpt.HostScene(os.path.join(GOLDEN, "scenes", scene_file)).stats())
        assert mine == stats, scene_file
    assert golden["05_head1ss_sub/head1ss_sub.json"]["shapes"][0]["triangles"] == 144046
    assert golden["06_gridsdf_synth/gridsdf_synth.json"]["volumes"][0]["whd"] == [48, 48, 48]
