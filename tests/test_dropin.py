"""The compiled drop-in: oracle/_ref/ref_dropin is the REFERENCE'S application code — its load_scene,
tesselate_surfaces, make_bvh, make_lights, make_state, get_render, save_image, in run_offline's order — with the body of
pathtrace_samples() replaced by the binding stub oracle/ref_dropin_stub.h over libvpt_hip.so.  Built in the build container
by oracle/Makefile from the reference's sources where they lie; the binary travels to the GPU box like ref_driver does."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from cases import CASES, EXTRA
from conftest import GOLDEN, ROOT, SCENE_03

DROPIN = os.path.join(ROOT, "oracle", "_ref", "ref_dropin")
needs_binary = pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/ref_dropin not built (needs /root/reference at build time)")


def _run(scene, shader, res, spp, bounces, tmp_path, nomis=False, output=None):
    state_file = str(tmp_path / "state.bin")
    cmd = [DROPIN, "--scene", scene, "--shader", shader, "--resolution", str(res), "--samples", str(spp), "--bounces", str(bounces),
           "--state", state_file] + (["--noimplicitmis"] if nomis else []) + (["--output", output] if output else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    raw = open(state_file, "rb").read()
    hdr = np.frombuffer(raw, np.int32, 4)
    w, h = int(hdr[1]), int(hdr[2])
    image = np.frombuffer(raw, np.float32, w * h * 4, 16).reshape(h, w, 4).copy()
    hits = np.frombuffer(raw, np.int32, w * h, 16 + w * h * 16).reshape(h, w).copy()
    rngs = np.frombuffer(raw, np.uint64, w * h * 2, 16 + w * h * 20).reshape(h, w, 2).copy()
    return image, hits, rngs, int(hdr[3])


@needs_binary
def test_dropin_without_a_gpu_reports_the_missing_device(vpt, tmp_path):
    if vpt.device_count() > 0:
        pytest.skip("a GPU is visible here")
    r = subprocess.run([DROPIN, "--scene", SCENE_03, "--samples", "1", "--resolution", "32"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
@needs_binary
@pytest.mark.parametrize("name", ["vol_96_16", "sdf_implicit_96_4", "sdfn_normal_128_2", "lobes_path_96_8"])
def test_reference_application_renders_through_the_hip_path(vpt, oracle, tmp_path, name):
    """state after N calls of the (replaced) pathtrace_samples from the reference's own run_offline loop, against the state the
    unmodified reference produced (committed fixtures); SDF scenes exercise the binding's type-tag recovery from the scene JSON"""
    from test_gpu_parity import MIN_03, MIN_EXTRA, _check_against_reference
    if name in CASES:
        shader, res, spp, bounces = CASES[name]
        scene_file, nomis, gold, floors = SCENE_03, False, np.load(os.path.join(GOLDEN, "03_volume_states.npz")), MIN_03[name]
    else:
        rel, shader, res, spp, bounces, nomis = EXTRA[name]
        scene_file, gold, floors = os.path.join(GOLDEN, "scenes", rel), np.load(os.path.join(GOLDEN, "substitute_states.npz")), MIN_EXTRA[name]
    image, hits, rngs, samples = _run(scene_file, shader, res, spp, bounces, tmp_path, nomis)
    scene = vpt.HostScene(scene_file)
    p = vpt.PathtraceParams(resolution=res, samples=spp, shader=shader, bounces=bounces, noimplicit_mis=nomis)
    g = vpt.PathtraceState(image.shape[1], image.shape[0], samples, image, hits, rngs)
    _check_against_reference(oracle, scene, p, spp, g, gold[name + "_image"], gold[name + "_rngs"], "dropin/" + name, *floors)


@pytest.mark.gpu
@needs_binary
def test_reference_application_writes_the_references_jpeg(tmp_path):
    """the reference's own save_image on the drop-in's render against the JPEG its unmodified build wrote"""
    import io
    from PIL import Image
    out = str(tmp_path / "x.jpg")
    _run(SCENE_03, "volpathtrace", 128, 8, 64, tmp_path, output=out)
    gold = os.path.join(GOLDEN, "03_volume_128_8.jpg")
    a = np.asarray(Image.open(out).convert("RGB"), np.float32) / 255
    b = np.asarray(Image.open(gold).convert("RGB"), np.float32) / 255
    rms = np.sqrt(np.mean((a - b) ** 2, axis=(0, 1)))
    print("drop-in JPEG per-channel RMS:", rms, "byte-identical:", open(out, "rb").read() == open(gold, "rb").read())
    assert a.shape == b.shape and (rms <= 2e-3).all()


def test_integration_md_shows_the_compiled_stub_verbatim():
    """what INTEGRATION.md presents as the reference-side binding is the very file oracle/Makefile compiles; no elisions"""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = open(os.path.join(ROOT, "oracle", "ref_dropin_stub.h")).read()
    assert stub in md
    assert "/* ..." not in md and "/*..." not in md
