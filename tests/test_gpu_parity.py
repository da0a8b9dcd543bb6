"""GPU parity tests (run on a real MI355X with -m gpu).  Everything goes through the C-ABI of
libvpt_hip.so; the CPU oracle is only the checker.

Tolerance model.  Integer work (PCG32 streams, hit counts) must be bit-exact.  Radiance is float32
computed with the reference's operation order, but device libm (ocml sinf/cosf/logf/expf/powf/
atan2f/acosf, <= 2 ulp) is not glibc's, so a small fraction of paths takes a different discrete
decision somewhere (a Fresnel coin, a russian-roulette test, a silhouette hit) and that pixel's
stream diverges from then on.  The tests therefore require (a) a large majority of pixels to end with
the oracle's exact RNG state, (b) those pixels to agree to 1e-3 relative, and (c) whole-image
statistics to agree within Monte-Carlo error.  Against the instructor images the bar is BASELINE's:
per-channel RMS <= 2e-3 after the reference's sRGB8 + JPEG q75 stage."""
import io
import os

import numpy as np
import pytest

from cases import EXTRA
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

SHADERS = [("volpathtrace", 64, 4, 64), ("volpathtrace", 96, 16, 64), ("pathtrace", 64, 4, 4), ("naive", 64, 4, 4),
           ("eyelight", 64, 2, 4), ("normal", 64, 2, 4), ("texcoord", 64, 2, 4), ("color", 64, 2, 4)]


def _pair(vpt, scene, dev, oracle, shader, res, spp, bounces, total=None):
    p = vpt.PathtraceParams(resolution=res, samples=total or spp, shader=shader, bounces=bounces)
    g = scene.make_state(p)
    c = g.copy()
    dev.pathtrace_samples(g, p, spp)
    oracle.oracle_render(scene, p, c, spp)
    return g, c


@pytest.mark.parametrize("shader,res,spp,bounces", SHADERS)
def test_gpu_matches_oracle_small(vpt, scene03, dev03, oracle, shader, res, spp, bounces):
    g, c = _pair(vpt, scene03, dev03, oracle, shader, res, spp, bounces)
    assert g.samples == c.samples == spp
    assert np.array_equal(g.hits, c.hits)                       # integer: exact
    same = np.all(g.rngs == c.rngs, axis=-1)                    # pixels that replayed the oracle's paths
    assert same.mean() >= 0.97, f"only {same.mean():.3f} of the pixel streams match the oracle"
    a, b = g.image[same], c.image[same]
    assert np.allclose(a, b, rtol=1e-3, atol=1e-4 * spp), float(np.abs(a - b).max())
    # image-level agreement including the diverged pixels
    assert abs(g.image[..., :3].mean() - c.image[..., :3].mean()) <= 0.02 * c.image[..., :3].mean() + 1e-6


def test_gpu_matches_reference_fixtures(vpt, scene03, dev03):
    """Same check directly against float32 states of the reference's own renderer."""
    gold = np.load(os.path.join(GOLDEN, "03_volume_states.npz"))
    p = vpt.PathtraceParams(resolution=96, samples=16, shader="volpathtrace", bounces=64)
    g = scene03.make_state(p)
    dev03.pathtrace_samples(g, p, 16)
    same = np.all(g.rngs == gold["vol_96_16_rngs"], axis=-1)
    assert same.mean() >= 0.97
    assert np.allclose(g.image[same], gold["vol_96_16_image"][same], rtol=1e-3, atol=2e-3)


def test_preview_branch_samples_equal_one(vpt, scene03, dev03):
    """params.samples == 1: pixel centres, no jitter draws (yocto_pathtrace.cpp:1059-1068)"""
    gold = np.load(os.path.join(GOLDEN, "03_volume_states.npz"))
    p = vpt.PathtraceParams(resolution=64, samples=1, shader="volpathtrace", bounces=64)
    g = scene03.make_state(p)
    dev03.pathtrace_samples(g, p, 1)
    same = np.all(g.rngs == gold["vol_64_1_rngs"], axis=-1)
    assert same.mean() >= 0.97 and np.allclose(g.image[same], gold["vol_64_1_image"][same], rtol=1e-3, atol=1e-4)


def test_batching_and_sample_cap(vpt, scene03, dev03):
    """nsamples batching is an extension: result must equal consecutive single calls, bit for bit;
    the call is a no-op once state.samples >= params.samples (cpp:1055)."""
    p = vpt.PathtraceParams(resolution=64, samples=5, shader="volpathtrace", bounces=64)
    a, b = scene03.make_state(p), scene03.make_state(p)
    dev03.pathtrace_samples(a, p, 8)           # capped at 5
    for _ in range(7):
        dev03.pathtrace_samples(b, p, 1)
    assert a.samples == b.samples == 5 and (a.hits == 5).all()
    assert np.array_equal(a.image.view(np.uint32), b.image.view(np.uint32)) and np.array_equal(a.rngs, b.rngs)


def test_run_to_run_determinism(vpt, scene03, dev03):
    p = vpt.PathtraceParams(resolution=128, samples=8, shader="volpathtrace", bounces=64)
    a, b = scene03.make_state(p), scene03.make_state(p)
    dev03.pathtrace_samples(a, p, 8)
    dev03.pathtrace_samples(b, p, 8)
    assert np.array_equal(a.image.view(np.uint32), b.image.view(np.uint32)) and np.array_equal(a.rngs, b.rngs)


def test_resume_across_backends(vpt, scene03, dev03, oracle):
    """pathtrace_state is a resumable checkpoint: 2 passes on the GPU then 2 on the CPU
    oracle ends (for pixels whose streams agree) where 4 oracle passes end."""
    p = vpt.PathtraceParams(resolution=64, samples=4, shader="volpathtrace", bounces=64)
    mixed, pure = scene03.make_state(p), scene03.make_state(p)
    dev03.pathtrace_samples(mixed, p, 2)
    oracle.oracle_render(scene03, p, mixed, 2)
    oracle.oracle_render(scene03, p, pure, 4)
    same = np.all(mixed.rngs == pure.rngs, axis=-1)
    assert same.mean() >= 0.97 and np.allclose(mixed.image[same], pure.image[same], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("nranks,tile", [(2, 8), (4, 16), (8, 8)])
def test_virtual_ranks_on_one_gpu_are_bit_identical(vpt, scene03, dev03, nranks, tile):
    """The N-GPU image must equal the 1-GPU image bit for bit: run the N ranks'
    launches one after another on this GPU through the device-state API, concatenate their tile
    buffers as all_gather would, resolve, compare."""
    import torch
    p = vpt.PathtraceParams(resolution=200, samples=1 << 20, shader="volpathtrace", bounces=64)
    host = scene03.make_state(p)
    spp = 3
    ref = host.copy()
    dev03.pathtrace_samples(ref, p, spp)
    dev = torch.device("cuda", 0)
    parts = []
    slots = vpt.layout_slots(vpt.VptLayout(host.width, host.height, tile, tile, 0, nranks))
    for r in range(nranks):
        lay = vpt.VptLayout(host.width, host.height, tile, tile, r, nranks)
        img = torch.zeros((slots, 4), dtype=torch.float32, device=dev)
        hit = torch.zeros((slots,), dtype=torch.int32, device=dev)
        rng = torch.zeros((slots, 2), dtype=torch.int64, device=dev)
        vpt.state_upload(lay, host, img.data_ptr(), hit.data_ptr(), rng.data_ptr())
        dev03.render_device(p, lay, spp, img.data_ptr(), hit.data_ptr(), rng.data_ptr(), 0)
        torch.cuda.synchronize()
        # host mirror of the slot map agrees with the device permutation
        idx = vpt.layout_pixel_index(lay)
        back = host.copy()
        vpt.state_download(lay, img.data_ptr(), hit.data_ptr(), rng.data_ptr(), back)
        mine = idx[idx >= 0]
        assert np.array_equal(back.image.reshape(-1, 4)[mine].view(np.uint32), ref.image.reshape(-1, 4)[mine].view(np.uint32))
        assert np.array_equal(back.rngs.reshape(-1, 2)[mine], ref.rngs.reshape(-1, 2)[mine])
        parts.append(img)
    gathered = torch.cat(parts, 0)
    frame = torch.zeros((host.height, host.width, 4), dtype=torch.float32, device=dev)
    vpt.resolve_device(vpt.VptLayout(host.width, host.height, tile, tile, 0, nranks), gathered.data_ptr(), spp, frame.data_ptr())
    torch.cuda.synchronize()
    expect = ref.image * np.float32(np.float32(1) / np.float32(spp))
    assert np.array_equal(frame.cpu().numpy().view(np.uint32), expect.view(np.uint32))


def test_full_size_properties(vpt, scene03, dev03, oracle):
    """BASELINE frame size 1280x533: size-independent properties + a low-res statistical cross-check."""
    p = vpt.PathtraceParams(resolution=1280, samples=1 << 20, shader="volpathtrace", bounces=64)
    g = scene03.make_state(p)
    before = g.rngs.copy()
    dev03.pathtrace_samples(g, p, 8)
    assert (g.width, g.height, g.samples) == (1280, 533, 8)
    assert (g.hits == 8).all() and np.isfinite(g.image).all() and (g.image >= 0).all()
    assert (g.rngs[..., 1] == before[..., 1]).all() and (g.rngs[..., 0] != before[..., 0]).all()  # inc fixed, state moved
    assert (g.image[..., 3] <= 8).all()                       # alpha accumulates 0/1 per sample
    # same scene at 320 wide on the oracle: mean radiance within MC error
    q = vpt.PathtraceParams(resolution=320, samples=1 << 20, shader="volpathtrace", bounces=64)
    c = scene03.make_state(q)
    oracle.oracle_render(scene03, q, c, 8)
    assert abs(g.image[..., :3].mean() / c.image[..., :3].mean() - 1) < 0.03


def _rms_vs_check(vpt, state, check_name):
    from PIL import Image
    jpg = vpt.encode_jpeg_q75(vpt.linear_to_srgb8(state.image, state.samples))
    mine = np.asarray(Image.open(io.BytesIO(jpg)).convert("RGB"), np.float32) / 255
    check = np.asarray(Image.open(os.path.join(GOLDEN, "check", check_name)).convert("RGB"), np.float32) / 255
    assert mine.shape == check.shape
    return np.sqrt(np.mean((mine - check) ** 2, axis=(0, 1)))


def test_instructor_image_lowres(vpt, scene03, dev03):
    """scripts/run.sh:3 — 720(x300) x 256 spp vs check/lowres (independent noise would be 0.059)"""
    p = vpt.PathtraceParams(resolution=720, samples=256, shader="volpathtrace", bounces=64)
    g = scene03.make_state(p)
    dev03.pathtrace_samples(g, p, 256)
    rms = _rms_vs_check(vpt, g, "03_volume_720_256.jpg")
    print("lowres per-channel RMS", rms)
    assert (rms <= 2e-3).all(), rms


def test_instructor_image_highres_target(vpt, scene03, dev03):
    """BASELINE target: tests/03_volume at 1280(x533) x 1024 spp within 2e-3 per-channel RMS of
    check/highres (scripts/run-highres.sh:3); independent noise would be 0.030."""
    p = vpt.PathtraceParams(resolution=1280, samples=1024, shader="volpathtrace", bounces=64)
    g = scene03.make_state(p)
    dev03.pathtrace_samples(g, p, 1024)
    rms = _rms_vs_check(vpt, g, "03_volume_1280_1024.jpg")
    print("highres per-channel RMS", rms)
    assert (rms <= 2e-3).all(), rms


def test_device_errors(vpt, scene03, dev03):
    import ctypes as C
    abi = vpt.PathtraceParams(shader="volpathtrace").to_abi()
    abi.shader = 42
    st = scene03.make_state(vpt.PathtraceParams(resolution=64))
    n = C.c_int(0)
    rc = vpt.hip.vpt_render(dev03.handle, C.byref(abi), 1, st.width, st.height, st.image.ctypes.data, st.hits.ctypes.data,
                            st.rngs.ctypes.data, C.byref(n))
    assert rc == -4 and b"sampler unknown" in vpt.hip.vpt_last_error()
    abi.shader, abi.camera = 0, 7
    rc = vpt.hip.vpt_render(dev03.handle, C.byref(abi), 1, st.width, st.height, st.image.ctypes.data, st.hits.ctypes.data,
                            st.rngs.ctypes.data, C.byref(n))
    assert rc == -1


# ---- substitute scenes (tests/golden/make_scenes.py; cases: tests/cases.py) ----------------------------------------
# Smallest share of pixels that must end with the reference's exact RNG state, set just under what MI355X measures
# (printed by the test; DESIGN.md §2 lists the measured values).  Streams split where a last-bit libm difference
# flips a discrete decision; the subsurface bunny (hundreds of scattering events per path) and the sphere-traced
# SDFs (hundreds of dependent float steps per ray) amplify more than the quad scenes.
MIN_SAME = {
    "surf_path_96_4": 0.95, "surf_normal_96_1": 0.99, "surf_eye_96_2": 0.99, "head_vol_96_4": 0.90,
    "sdf_implicit_96_4": 0.90, "sdf_nomis_96_4": 0.90, "sdf_normal_96_2": 0.99,
    "sdfn_implicit_128_8": 0.90, "sdfn_nomis_128_4": 0.90, "sdfn_normal_128_2": 0.99,
    "lobes_path_96_8": 0.90, "lobes_vol_96_8": 0.90, "lobes_naive_96_4": 0.90, "lobes_eye_96_2": 0.95,
}
assert set(MIN_SAME) == set(EXTRA)


@pytest.mark.parametrize("name", sorted(EXTRA))
def test_gpu_matches_reference_on_substitute_scenes(vpt, oracle, name):
    """Against float32 states produced by the reference's own renderer on the substitute scenes.

    Pixels that consumed exactly the reference's random numbers (same RNG end state) must carry the reference's
    radiance to 2e-3 — with ONE proven exclusion: pixels in which sample_lights_pdf evaluated the pdf of an SDF light
    the ray hit.  The reference takes that light's normal by finite differences at the shading point with a step
    below the float spacing of the coordinates (yocto_pathtrace.cpp:389), so the pdf is a discontinuous function of
    the last bits of the position (tests/test_kat.py::test_sdf_light_pdf_is_ill_conditioned_in_the_reference: a
    1-ulp nudge moves it by more than 2e-3 in 40 % of the cases) and the path weight with it, without changing the
    number of draws.  The oracle (bit-identical to the reference) reports those pixels (flag bit 0); outside them
    the check is strict, inside them only a quantile is required."""
    scene_file, shader, res, spp, bounces, nomis = EXTRA[name]
    gold = np.load(os.path.join(GOLDEN, "substitute_states.npz"))
    scene = vpt.HostScene(os.path.join(GOLDEN, "scenes", scene_file))
    dev = vpt.DeviceScene(scene, 0)
    p = vpt.PathtraceParams(resolution=res, samples=spp, shader=shader, bounces=bounces, noimplicit_mis=nomis)
    g = scene.make_state(p)
    dev.pathtrace_samples(g, p, spp)
    ref_img, ref_rng = gold[name + "_image"], gold[name + "_rngs"]
    assert (g.hits == spp).all()
    same = np.all(g.rngs == ref_rng, axis=-1)
    # the oracle's condition flags for the same render (and once more: it is the reference, bit for bit)
    c = scene.make_state(p)
    flags = np.zeros((c.height, c.width), np.uint8)
    oracle.oracle_render(scene, p, c, spp, flags=flags)
    assert np.array_equal(c.image.view(np.uint32), ref_img.view(np.uint32))
    noisy = (flags & 1) != 0
    close = np.all(np.isclose(g.image, ref_img, rtol=2e-3, atol=2e-3 * spp), axis=-1)
    strict, loose = same & ~noisy, same & noisy
    print(f"{name}: streams identical {same.mean():.4f}; well-conditioned replayed pixels {strict.sum()}, within tolerance "
          f"{close[strict].mean() if strict.any() else 1.0:.5f} (worst abs diff {float(np.abs(g.image - ref_img)[strict].max()) if strict.any() else 0.0:.3g}); "
          f"SDF-light-pdf pixels {loose.sum()}, within tolerance {close[loose].mean() if loose.any() else 1.0:.4f}")
    assert same.mean() >= MIN_SAME[name], same.mean()
    assert close[strict].all(), np.argwhere(strict & ~close)[:10].tolist()
    if loose.any():
        assert close[loose].mean() >= 0.80
    m_g, m_r = g.image[..., :3].mean(), ref_img[..., :3].mean()
    assert abs(m_g - m_r) <= 0.05 * abs(m_r) + 1e-6


def test_reciprocal_shortcut_is_exact_for_every_float(vpt):
    """rcp + one Newton step must equal the IEEE quotient for every operand the kernels feed it (biased
    exponent 1..250); the rest (zeros, denormals, >= 2^124, inf, NaN: 2 * (2^23 + 5 * 2^23) patterns) divide."""
    bad, skipped = vpt.selftest_reciprocal(0)
    assert bad == 0
    assert skipped == 2 * 6 * (1 << 23)


@pytest.mark.parametrize("scene_file,shader,res,spp,bounces", [
    ("03_volume/volume.json", "volpathtrace", 96, 4, 64),
    ("05_head1ss_sub/head1ss_sub.json", "volpathtrace", 64, 2, 16),   # BVH deep enough for the HBM-backed stack variant
    ("01_surface_min/surface_min.json", "pathtrace", 96, 4, 8),
])
def test_streaming_pipeline_is_bit_identical(vpt, tmp_path, scene_file, shader, res, spp, bounces):
    """VPT_PIPELINE=stream (k_begin / k_trace / k_shade over ray queues) must end in exactly the state the
    single-kernel form produces: same arithmetic, same per-pixel draw order, only the schedule differs."""
    import subprocess
    import sys
    path = os.path.join(GOLDEN, "scenes", scene_file)
    out = str(tmp_path / "stream.npz")
    env = dict(os.environ, VPT_PIPELINE="stream")
    subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "render_state.py"), path, shader, str(res), str(spp),
                    str(bounces), out], check=True, env=env, timeout=300)
    got = np.load(out)
    scene = vpt.HostScene(path)
    dev = vpt.DeviceScene(scene, 0)
    p = vpt.PathtraceParams(resolution=res, samples=spp, shader=shader, bounces=bounces)
    st = scene.make_state(p)
    dev.pathtrace_samples(st, p, spp)
    assert np.array_equal(got["rngs"], st.rngs) and np.array_equal(got["hits"], st.hits)
    assert np.array_equal(got["image"].view(np.uint32), st.image.view(np.uint32))


@pytest.mark.parametrize("scene_file", ["03_volume/volume.json", "05_head1ss_sub/head1ss_sub.json"])
def test_light_cdf_index_equals_upper_bound(vpt, scene_file):
    """The guide table + 16-ary levels over a large light CDF must return std::upper_bound's index for CDF
    entries, their float neighbours, both ends of the range and uniform values (2^20 probes per light)."""
    scene = vpt.HostScene(os.path.join(GOLDEN, "scenes", scene_file))
    dev = vpt.DeviceScene(scene, 0)
    kinds = []
    light = 0
    while True:
        try:
            bad, indexed = dev.selftest_light_cdf(light)
        except vpt.VptError:
            break
        assert bad == 0, (light, bad)
        kinds.append(indexed)
        light += 1
    assert light >= 1 and 2 in kinds, kinds   # every test scene has a textured environment light: guide table in use


def test_device_output_stage_matches_host_quantisation(vpt, scene03, dev03):
    """vpt_resolve_srgb8_device (get_render + rgb_to_srgb + float_to_byte on the device) against the host routine
    the parity pipeline uses: same bytes except where ocml's powf and glibc's land on different sides of a
    quantisation step (at most one count, on a tiny share of the channels)."""
    import torch
    p = vpt.PathtraceParams(resolution=320, samples=8, shader="volpathtrace", bounces=8)
    st = scene03.make_state(p)
    dev03.pathtrace_samples(st, p, 8)
    layout = vpt.VptLayout(st.width, st.height, 8, 8, 0, 1)
    slots = vpt.layout_slots(layout)
    d_img = torch.zeros((slots, 4), dtype=torch.float32, device="cuda")
    d_hits = torch.zeros((slots,), dtype=torch.int32, device="cuda")
    d_rng = torch.zeros((slots, 2), dtype=torch.int64, device="cuda")
    vpt.state_upload(layout, st, d_img.data_ptr(), d_hits.data_ptr(), d_rng.data_ptr())
    out = torch.zeros((st.height, st.width, 4), dtype=torch.uint8, device="cuda")
    vpt.resolve_srgb8_device(layout, d_img.data_ptr(), st.samples, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.int32)
    ref = vpt.linear_to_srgb8(st.image, st.samples).astype(np.int32)
    diff = np.abs(got - ref)
    assert diff.max() <= 1
    assert (diff != 0).mean() < 1e-3


from kat_lib import edge_rays as _edge_rays  # noqa: E402


@pytest.mark.parametrize("scene_file,lo,hi", [
    ("03_volume/volume.json", (-0.7, -0.05, -0.45), (0.7, 0.45, 0.45)),
    ("05_head1ss_sub/head1ss_sub.json", (-0.25, -0.1, -0.25), (0.25, 0.4, 0.25)),
])
def test_intersect_is_bit_identical_on_edge_case_rays(vpt, oracle, scene_file, lo, hi):
    """vpt_intersect (the kernels' quad-node traversal) against the oracle's intersect_bvh, ray by ray: instance,
    element, uv and distance must agree bit for bit, also for NaN-prone rays and for single-instance queries."""
    scene = vpt.HostScene(os.path.join(GOLDEN, "scenes", scene_file))
    dev = vpt.DeviceScene(scene, 0)
    rays = _edge_rays(np.random.default_rng(7), lo, hi, 40000)
    for instance in (-1, 0):
        ids, uvt = dev.intersect(rays, instance)
        rids, ruvt = oracle.oracle_intersect(scene, rays, instance)
        assert (ids[:, 0] >= 0).mean() > 0.05   # the batch does hit things
        assert np.array_equal(ids, rids), np.nonzero((ids != rids).any(axis=1))[0][:10]
        assert np.array_equal(uvt.view(np.uint32), ruvt.view(np.uint32)), np.nonzero((uvt.view(np.uint32) != ruvt.view(np.uint32)).any(axis=1))[0][:10]
