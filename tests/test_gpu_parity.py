"""GPU parity tests (run on a real MI355X with -m gpu).  Everything goes through the C-ABI of
libvpt_hip.so; the CPU oracle is only the checker.

Tolerance model.  Integer work (PCG32 streams, hit counts) must be bit-exact.  Radiance is float32 computed with
the reference's operation order; every function on the path is checked against reference-made tables on identical
inputs (tests/test_kat.py: bit-exact where no libm call is involved).  The one thing that is NOT the reference's is
the device libm (ocml sinf / cosf / logf / expf / powf / atan2f / acosf / atanf, within 1-2 ulp of glibc's), and the
reference amplifies a last-bit difference in two ways: it flips a discrete decision (a Fresnel coin, a roulette test,
a silhouette hit, the texel an environment direction falls in) and the pixel's stream diverges from then on; or it
lands in one of the reference's ill-conditioned finite differences — every SDF normal is taken with a step
h = flt_eps * t, at or below the float spacing of the coordinates (yocto_sdfs.cpp:67-89), and sample_lights_pdf takes
an SDF light's normal at the SHADING point (yocto_pathtrace.cpp:389) — and the radiance moves by percents while the
pixel consumes exactly the same random numbers.

Which pixels those are is MEASURED on the reference's own arithmetic instead of assumed: oracle_lib.unstable_pixels
re-renders the case on the oracle (bit-identical to the reference) with every libm result nudged by -1 / 0 / +1 ulp
and marks the pixels whose stream or radiance moves.  The checks are then
 (a) STRICT: a pixel may differ from the reference (other RNG end state, or radiance off by more than 2e-3) only if
     the reference's own value of that pixel moves (by more than 1e-3, or to another stream) under such nudges.
     Two stages: 16 nudge patterns over the whole frame; the few disagreeing pixels those left unexplained (pixels
     that move in a few per cent of the patterns only) get 2048 more patterns each.  A pixel that disagrees and is
     stable under all of them fails the test;
 (b) floors on the share of pixels with identical streams and on the share of stable pixels (so that the exclusion
     cannot swallow the image), both set just under what MI355X measures (the tests print them);
 (c) whole-image statistics within Monte-Carlo error.
Against the instructor images the bar is BASELINE's: per-channel RMS <= 2e-3 after the reference's sRGB8 + JPEG q75
stage."""
import io
import os

import numpy as np
import pytest

from cases import CASES, EXTRA
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

def _check_against_reference(oracle, scene, params, spp, g, ref_img, ref_rng, name, min_same, min_ok, min_stable, pixels=None, deep_rounds=2048):
    """the strict check of the module docstring; g = device state, ref_* = the reference's state.
    pixels: row-major indices of the pixels the reference state holds (a seeded sample of a full-size frame): the check and
    its shares are over those pixels only"""
    assert g.samples == spp and (g.hits == spp).all()                 # integer: exact
    chosen = np.ones(ref_rng.shape[:2], bool)
    if pixels is not None:
        chosen[:] = False
        chosen.reshape(-1)[pixels] = True
    same = np.all(g.rngs == ref_rng, axis=-1)                        # pixels that replayed the reference's paths
    close = np.all(np.isclose(g.image, ref_img, rtol=2e-3, atol=2e-3 * spp), axis=-1)
    fresh = lambda: scene.make_state(params)                         # noqa: E731
    u_stream, u_rad = oracle.unstable_pixels(scene, params, spp, ref_img, ref_rng, fresh, rounds=16, pixels=pixels)
    stable = ~(u_stream | u_rad)
    ok = same & close
    left = np.flatnonzero((chosen & stable & ~ok).reshape(-1))      # disagreeing pixels the 16 frame-wide patterns did not move
    deep = 0
    if 0 < len(left) <= 64:                                          # stage 2: 2048 more patterns on those pixels only
        d_stream, d_rad = oracle.unstable_pixels(scene, params, spp, ref_img, ref_rng, fresh, rounds=deep_rounds, pixels=left, first_seed=1000)
        deep = int((d_stream | d_rad).sum())
        stable &= ~(d_stream | d_rad)
    bad = chosen & stable & ~ok
    rest = chosen & ~stable
    share = lambda m: float(m[chosen].mean())                        # noqa: E731
    print(f"{name}: {int(chosen.sum())} pixels; streams identical {share(same):.4f}; matching the reference {share(ok):.4f}; stable under 1-ulp libm nudges {share(stable):.4f} "
          f"(stream-unstable {share(u_stream):.4f}, radiance-unstable {share(u_rad):.4f}, shown unstable only by the deep stage {deep}); "
          f"unstable pixels matching {ok[rest].mean() if rest.any() else 1.0:.4f}; worst abs diff on stable pixels "
          f"{float(np.abs(g.image - ref_img)[chosen & stable].max()):.3g}")
    assert not bad.any(), (name, int(bad.sum()), np.argwhere(bad)[:10].tolist())
    assert share(same) >= min_same, (name, share(same))
    assert share(ok) >= min_ok, (name, share(ok))
    assert share(stable) >= min_stable, (name, share(stable))
    m_g, m_r = g.image[chosen][..., :3].mean(), ref_img[chosen][..., :3].mean()      # image-level agreement including the diverged pixels
    assert abs(m_g - m_r) <= 0.03 * abs(m_r) + 1e-6


# 03_volume, the reference's own assets (tests/cases.py CASES): floors just under the measured shares
MIN_03 = {  # name -> floors on (identical streams, pixels matching the reference, stable pixels); measured: 1.0000, 1.0000 and
    # 0.9716 / 0.9057 / 0.8479 for the three volumetric cases, 1.0000 otherwise (gpurun_out/r2c/tests.log, 2026-10-04)
    "vol_64_1": (0.998, 0.998, 0.96), "vol_64_4": (0.998, 0.998, 0.89), "vol_96_16": (0.998, 0.998, 0.83),
    "path_64_4": (0.998, 0.998, 0.998), "naive_64_4": (0.998, 0.998, 0.998), "eye_64_2": (0.998, 0.998, 0.998),
    "normal_64_2": (0.998, 0.998, 0.998), "texcoord_64_2": (0.998, 0.998, 0.998), "color_64_2": (0.998, 0.998, 0.998),
}
assert set(MIN_03) == set(CASES)


@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_matches_reference_fixtures(vpt, scene03, dev03, oracle, name):
    """float32 states of the reference's own renderer on tests/03_volume, every shader (samples == 1: the pixel-centre
    preview branch, yocto_pathtrace.cpp:1059-1068)"""
    shader, res, spp, bounces = CASES[name]
    gold = np.load(os.path.join(GOLDEN, "03_volume_states.npz"))
    p = vpt.PathtraceParams(resolution=res, samples=spp, shader=shader, bounces=bounces)
    g = scene03.make_state(p)
    dev03.pathtrace_samples(g, p, spp)
    _check_against_reference(oracle, scene03, p, spp, g, gold[name + "_image"], gold[name + "_rngs"], name, *MIN_03[name])


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
    assert same.mean() >= 0.998 and np.allclose(mixed.image[same], pure.image[same], rtol=1e-3, atol=1e-3)   # measured 1.0000


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


FULL_SIZE = {  # BASELINE.json configs at their full frame sizes: (scene, shader, resolution, bounces, expected (w, h), spp,
    # floors on the sampled pixels' (identical streams, matching the reference, stable) shares: just under the MI355X-measured ones)
    "config2_03_volume_1280": ("03_volume/volume.json", "volpathtrace", 1280, 64, (1280, 533), 8, (0.998, 0.998, 0.88)),     # measured 1.0000 1.0000 0.8954
    "config3_05_head_1280": ("05_head1ss_sub/head1ss_sub.json", "volpathtrace", 1280, 64, (1280, 1280), 4, (0.998, 0.998, 0.995)),   # 1.0000 1.0000 1.0000
    "config4_06_gridsdf_1280": ("06_gridsdf_full/gridsdf_full.json", "implicit", 1280, 4, (1280, 533), 4, (0.994, 0.987, 0.935)),   # 0.9978 0.9924 0.9508
    "config5_03_volume_3840": ("03_volume/volume.json", "volpathtrace", 3840, 64, (3840, 1600), 2, (0.998, 0.998, 0.935)),   # 1.0000 1.0000 0.9491 (gpurun_out/r3b/tests.log, round 3)
}
FULL_SIZE_PIXELS = 4096


def _sampled_pixels(size):
    """a seeded sample of FULL_SIZE_PIXELS pixels of a frame, a quarter of it from the frame's last rows and columns (where ragged
    tiles and the end of the launch order live), as sorted row-major indices"""
    rng = np.random.default_rng(size[0] * 7 + size[1])
    w, hgt = size
    pix = rng.choice(w * hgt, FULL_SIZE_PIXELS * 3 // 4, replace=False)
    edge_y = rng.integers(hgt - 16, hgt, FULL_SIZE_PIXELS // 8) * w + rng.integers(0, w, FULL_SIZE_PIXELS // 8)
    edge_x = rng.integers(0, hgt, FULL_SIZE_PIXELS // 8) * w + rng.integers(w - 16, w, FULL_SIZE_PIXELS // 8)
    return np.unique(np.concatenate([pix, edge_y, edge_x])).astype(np.int32)


@pytest.mark.parametrize("name", sorted(FULL_SIZE))
def test_full_size_properties(vpt, oracle, name):
    """The BASELINE frame sizes: size-independent properties of the whole frame + the STRICT per-pixel check of the module
    docstring on a seeded sample of 4 096 pixels, which the oracle (bit-identical to the reference) renders at the full
    resolution (vpt_oracle_render_pixels): same RNG end state, radiance within 2e-3, unless the reference's own value of
    that pixel moves under 1-ulp libm nudges."""
    scene_file, shader, res, bounces, size, spp, floors = FULL_SIZE[name]
    scene = vpt.HostScene(os.path.join(GOLDEN, "scenes", scene_file))
    dev = vpt.DeviceScene(scene, 0)
    p = vpt.PathtraceParams(resolution=res, samples=1 << 20, shader=shader, bounces=bounces)
    g = scene.make_state(p)
    before = g.rngs.copy()
    dev.pathtrace_samples(g, p, spp)
    assert (g.width, g.height, g.samples) == (*size, spp)
    assert (g.hits == spp).all() and np.isfinite(g.image).all() and (g.image >= 0).all()
    assert (g.rngs[..., 1] == before[..., 1]).all() and (g.rngs[..., 0] != before[..., 0]).all()  # inc fixed, state moved
    assert (g.image[..., 3] <= spp).all()                     # alpha accumulates 0/1 per sample
    # batching is exact at full size too: the same frame in two launches ends in the same state, bit for bit
    h = scene.make_state(p)
    dev.pathtrace_samples(h, p, 1)
    dev.pathtrace_samples(h, p, spp - 1)
    assert np.array_equal(h.image.view(np.uint32), g.image.view(np.uint32)) and np.array_equal(h.rngs, g.rngs)
    # pixel-exact parity on a seeded sample of the full-size frame
    w, hgt = size
    pix = _sampled_pixels(size)
    q = vpt.PathtraceParams(resolution=res, samples=spp, shader=shader, bounces=bounces)
    ref = scene.make_state(q)
    oracle.oracle_render(scene, q, ref, spp, nthreads=0, pixels=pix)
    untouched = np.ones(w * hgt, bool)
    untouched[pix] = False
    assert (ref.image.reshape(-1, 4)[untouched] == 0).all()   # the oracle rendered the sampled pixels only
    _check_against_reference(oracle, scene, q, spp, g, ref.image, ref.rngs, name + f" {w}x{hgt}x{spp}", *floors, pixels=pix)


LONG_CHAIN = {  # configs 3 and 4 at their full frame, 256 samples of a pixel's serial PCG32 chain (their BASELINE spp are 1024 / 512; the
    # cases above hold them to the reference for 4).  Floors on (identical streams, matching, stable) just under the MI355X-measured shares.
    # A pixel's stream survives 256 samples only if none of its ~700 (head) / ~1 500 (sdf) libm-dependent decisions flips, so the
    # identical-stream share is lower than at 4 spp by construction; what the strict check demands is unchanged: a pixel may differ
    # only where the reference's own value moves under 1-ulp libm nudges.
    # (scene, shader, resolution, bounces, (w, h), spp, pixels, floors); the voxel scene's oracle does 0.1 Msamples/s on the sampled pixels,
    # so its case holds 1 024 of them and its second stage 128 patterns: the case stays within two minutes of host time
    "config3_05_head_1280x256": ("05_head1ss_sub/head1ss_sub.json", "volpathtrace", 1280, 64, (1280, 1280), 256, 4096, (0.998, 0.997, 0.99)),   # measured 0.9998 0.9995 0.9978 (gpurun_out/r4a)
    "config4_06_gridsdf_1280x256": ("06_gridsdf_full/gridsdf_full.json", "implicit", 1280, 4, (1280, 533), 256, 1024, (0.90, 0.88, 0.70)),   # measured 0.9297 0.9160 0.7422 (gpurun_out/r4e)
}


@pytest.mark.parametrize("name", sorted(LONG_CHAIN))
def test_long_chains_on_the_big_scenes(vpt, oracle, name):
    """256 samples per pixel on the 144 046-triangle scene (K1, overflow-stack instance) and on the 96^3 + 64^3 voxel scene (K2):
    the full frame on the device in one call, the oracle on a seeded sample of 4 096 of its pixels, the strict per-pixel check."""
    scene_file, shader, res, bounces, size, spp, npix, floors = LONG_CHAIN[name]
    scene = vpt.HostScene(os.path.join(GOLDEN, "scenes", scene_file))
    dev = vpt.DeviceScene(scene, 0)
    q = vpt.PathtraceParams(resolution=res, samples=spp, shader=shader, bounces=bounces)
    g = scene.make_state(q)
    dev.pathtrace_samples(g, q, spp)
    assert (g.width, g.height, g.samples) == (*size, spp)
    assert (g.hits == spp).all() and np.isfinite(g.image).all()
    pix = _sampled_pixels(size)
    pix = pix[:: max(1, len(pix) // npix)][:npix]
    print(f"{name}: device frame done, oracle on {len(pix)} pixels x {spp} spp ...", flush=True)
    ref = scene.make_state(q)
    oracle.oracle_render(scene, q, ref, spp, nthreads=0, pixels=pix)
    _check_against_reference(oracle, scene, q, spp, g, ref.image, ref.rngs, name + f" {size[0]}x{size[1]}x{spp}", *floors, pixels=pix, deep_rounds=128)


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
MIN_EXTRA = {  # name -> floors on (identical streams, pixels matching the reference, stable pixels), just under the measured
    # shares (in the comments; gpurun_out/r2c/tests.log, 2026-10-04)
    "surf_path_96_4": (0.997, 0.997, 0.985),        # 0.9995 0.9995 0.9927
    "surf_normal_96_1": (0.998, 0.998, 0.998), "surf_eye_96_2": (0.998, 0.998, 0.998),   # 1.0000 1.0000 1.0000
    "surf_subdiv_96_4": (0.997, 0.997, 0.985),      # 0.9995 0.9995 0.9930 (round 3: the scene now holds the reference's subdivision cages)
    "subdiv_path_96_4": (0.998, 0.998, 0.995), "subdiv_normal_96_2": (0.998, 0.998, 0.998),   # 1.0000 1.0000 0.9993 / 1.0000
    "sdf_full_implicit_96_4": (0.994, 0.989, 0.94),   # 0.9982 0.9943 0.9523 (gpurun_out/r3b/tests.log, round 3)
    "head_vol_96_4": (0.998, 0.998, 0.995),         # 1.0000 1.0000 0.9999
    "sdf_implicit_96_4": (0.993, 0.985, 0.93),      # 0.9969 0.9909 0.9432
    "sdf_nomis_96_4": (0.993, 0.987, 0.925),        # 0.9966 0.9922 0.9362
    "sdf_normal_96_2": (0.998, 0.998, 0.998),       # 1.0000 1.0000 1.0000
    "sdfn_implicit_128_8": (0.985, 0.97, 0.83),     # 0.9894 0.9761 0.8407
    "sdfn_nomis_128_4": (0.984, 0.975, 0.865),      # 0.9885 0.9820 0.8782
    "sdfn_normal_128_2": (0.998, 0.998, 0.998),     # 1.0000 1.0000 1.0000
    "lobes_path_96_8": (0.998, 0.998, 0.995), "lobes_vol_96_8": (0.998, 0.998, 0.995),   # 1.0000 1.0000 0.9997 / 1.0000
    "lobes_naive_96_4": (0.998, 0.998, 0.995), "lobes_eye_96_2": (0.998, 0.998, 0.998),  # 1.0000 1.0000 0.9997 / 1.0000
}
assert set(MIN_EXTRA) == set(EXTRA)


@pytest.mark.parametrize("name", sorted(EXTRA))
def test_gpu_matches_reference_on_substitute_scenes(vpt, oracle, name):
    """float32 states of the reference's own renderer on the substitute scenes: glossy + normal maps, the 144k-triangle
    subsurface bunny, voxel / analytic SDFs with an SDF light, every sd_* primitive and every BSDF lobe (rough and
    delta), an emissive mesh with a real BVH."""
    scene_file, shader, res, spp, bounces, nomis = EXTRA[name]
    gold = np.load(os.path.join(GOLDEN, "substitute_states.npz"))
    scene = vpt.HostScene(os.path.join(GOLDEN, "scenes", scene_file))
    dev = vpt.DeviceScene(scene, 0)
    p = vpt.PathtraceParams(resolution=res, samples=spp, shader=shader, bounces=bounces, noimplicit_mis=nomis)
    g = scene.make_state(p)
    dev.pathtrace_samples(g, p, spp)
    _check_against_reference(oracle, scene, p, spp, g, gold[name + "_image"], gold[name + "_rngs"], name, *MIN_EXTRA[name])


def test_reciprocal_shortcut_is_exact_for_every_float(vpt):
    """rcp + one Newton step must equal the IEEE quotient for every operand the kernels feed it (biased
    exponent 1..250); the rest (zeros, denormals, >= 2^124, inf, NaN: 2 * (2^23 + 5 * 2^23) patterns) divide."""
    bad, skipped = vpt.selftest_reciprocal(0)
    assert bad == 0
    assert skipped == 2 * 6 * (1 << 23)


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
    # The traversal has two forms (vpt_mesh_kernel.hip.h): a lane walks its own ray while more than 16 rays of its wave need node or
    # leaf work, and sets of up to 16 rays run in "sessions" on four lanes each.  Dense waves (64 rays) start in the first form
    # and drain into the second; the sparse batches keep 12 / 3 / 1 real rays per wave (the other lanes hold rays that miss the scene's
    # box and finish at once), so that sessions carry those rays from their first node on.
    far = np.float32([1e3, 1e3, 1e3, 0.57735027, 0.57735027, 0.57735027])
    lane = np.arange(len(rays)) % 64
    batches = [("dense", rays)]
    for keep, name in ((12, "12 per wave"), (3, "3 per wave"), (1, "1 per wave")):
        sparse = rays.copy()
        sparse[(lane * 7 + 3) % 64 >= keep] = far
        batches.append((name, sparse))
    for name, batch in batches:
        for instance in (-1, 0):
            ids, uvt = dev.intersect(batch, instance)
            rids, ruvt = oracle.oracle_intersect(scene, batch, instance)
            if name == "dense":
                assert (ids[:, 0] >= 0).mean() > 0.05   # the batch does hit things
            assert np.array_equal(ids, rids), (name, instance, np.nonzero((ids != rids).any(axis=1))[0][:10])
            assert np.array_equal(uvt.view(np.uint32), ruvt.view(np.uint32)), (name, instance, np.nonzero((uvt.view(np.uint32) != ruvt.view(np.uint32)).any(axis=1))[0][:10])


def test_intersect_follows_the_reference_through_nan_hits(vpt, oracle, tmp_path):
    """A triangle of denormal size met exactly at its first corner gives det != 0 with 1 / det = inf and u = v = t = 0 * inf = NaN, which
    intersect_triangle's comparisons all let through (yocto_geometry.h:786-819): the reference records a hit with a NaN distance, and from
    then on ray.tmax = NaN accepts every later primitive of the leaf and fails every later box test.  The group form of the leaf phase tests a
    leaf's primitives in parallel, which is the sequential loop only while no NaN is in play: it must notice and hand the leaf back.  Rays
    from the corner (NaN hits, dense and sparse waves) and rays that pass near it must match the oracle: ids exactly, floats bit for bit or NaN for NaN."""
    tiny = 1e-20
    obj = ["v -1 -1 -1", "v 1 -1 -1", "v 1 1 -1", "v -1 1 -1",                   # a wall behind
           "v 0 0 0", f"v {tiny} 0 0", f"v 0 {tiny} 0",                          # the degenerate triangle at the origin
           "v -1 -1 1", "v 1 -1 1", "v 1 1 1", "v -1 1 1"]                       # a wall in front
    faces = ["f 1 2 3", "f 1 3 4", "f 5 6 7", "f 8 9 10", "f 8 10 11"]
    rng = np.random.default_rng(3)
    for k in range(40):                                                           # clutter, so that the BVH has several leaves and levels
        c = rng.uniform(-0.9, 0.9, 3)
        n = len(obj)
        obj += [f"v {c[0]} {c[1]} {c[2]}", f"v {c[0] + 0.1} {c[1]} {c[2]}", f"v {c[0]} {c[1] + 0.1} {c[2]}"]
        faces.append(f"f {n + 1} {n + 2} {n + 3}")
    (tmp_path / "a.obj").write_text("\n".join(obj + faces) + "\n")
    desc = {"asset": {"version": "4.2"}, "cameras": [{"name": "c", "aspect": 1.0}], "materials": [{"name": "m", "type": "matte", "color": [0.5, 0.5, 0.5]}],
            "shapes": [{"name": "s", "uri": "a.obj"}], "instances": [{"name": "i", "shape": 0, "material": 0}]}
    import json
    (tmp_path / "scene.json").write_text(json.dumps(desc))
    scene = vpt.HostScene(str(tmp_path / "scene.json"))
    dev = vpt.DeviceScene(scene, 0)
    n = 64 * 40
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 6), np.float32)
    rays[:, 3:] = d                                                                # all from the corner itself: tvec == 0
    near = rays.copy()
    near[:, :3] = rng.uniform(-0.5, 0.5, size=(n, 3)).astype(np.float32)           # ordinary rays through the same tree
    far = np.float32([1e3, 1e3, 1e3, 0.57735027, 0.57735027, 0.57735027])
    sparse = rays.copy()
    sparse[(np.arange(n) * 7 + 3) % 64 >= 5] = far                                 # five corner rays per wave: group forms from the first node on
    mixed = np.where((np.arange(n) % 3 == 0)[:, None], rays, near)
    saw_nan = 0
    for name, batch in (("corner", rays), ("near", near), ("sparse", sparse), ("mixed", mixed)):
        for instance in (-1, 0):
            ids, uvt = dev.intersect(batch, instance)
            rids, ruvt = oracle.oracle_intersect(scene, batch, instance)
            assert np.array_equal(ids, rids), (name, instance, np.nonzero((ids != rids).any(axis=1))[0][:10])
            same = (uvt.view(np.uint32) == ruvt.view(np.uint32)) | (np.isnan(uvt) & np.isnan(ruvt))
            assert same.all(), (name, instance, np.nonzero(~same.all(axis=1))[0][:10])
            saw_nan += int(np.isnan(ruvt[:, 2]).sum())
    assert saw_nan > 100                                                           # the case this test is about did occur


@pytest.mark.parametrize("scene_file,shader,res,bounces,spp", [
    ("03_volume/volume.json", "volpathtrace", 1280, 64, 16),
    ("05_head1ss_sub/head1ss_sub.json", "volpathtrace", 1280, 64, 4),
    ("03_volume_lobes/volume_lobes.json", "volpathtrace", 640, 64, 8),
    ("01_surface_min/surface_min.json", "pathtrace", 640, 4, 8),
    ("06_gridsdf_full/gridsdf_full.json", "implicit", 1280, 4, 8),
    ("07_sdfunction_synth/sdfunction_synth.json", "implicit", 640, 6, 8),
])
def test_group_forms_do_not_change_a_bit(vpt, monkeypatch, scene_file, shader, res, bounces, spp):
    """Round 4's kernels run a phase that holds few rays on four lanes per ray (K1: node and leaf phases of traverse(); K2: the scene march).
    VPT_NO_GROUP_FORMS=1 (read when a scene handle is created) keeps every ray in its own lane: the whole frame must come out bit for bit the
    same - radiance sums, hit counts, RNG streams - at the BASELINE frame sizes, over batches, on every kind of scene."""
    scene = vpt.HostScene(os.path.join(GOLDEN, "scenes", scene_file))
    p = vpt.PathtraceParams(resolution=res, samples=1 << 20, shader=shader, bounces=bounces)
    with_forms = vpt.DeviceScene(scene, 0)
    monkeypatch.setenv("VPT_NO_GROUP_FORMS", "1")
    without = vpt.DeviceScene(scene, 0)
    monkeypatch.delenv("VPT_NO_GROUP_FORMS")
    a, b = scene.make_state(p), scene.make_state(p)
    with_forms.pathtrace_samples(a, p, spp)
    without.pathtrace_samples(b, p, 1)
    without.pathtrace_samples(b, p, spp - 1)
    assert a.samples == b.samples == spp
    assert np.array_equal(a.image.view(np.uint32), b.image.view(np.uint32)) and np.array_equal(a.rngs, b.rngs) and np.array_equal(a.hits, b.hits)


def _scene_of_triangles(tmp_path):
    """08_subdiv_synth without its two quads (floor, area light): four tesselated shapes, with and without vertex normals / texcoords, textured
    materials, lit by the environment - a scene whose shapes all hold triangles and whose BVHs fit the LDS stack."""
    import json
    src = os.path.join(GOLDEN, "scenes")
    os.makedirs(tmp_path / "scene")
    for name in ("03_volume", "shared_textures"):
        os.symlink(os.path.join(src, name), tmp_path / name)
    os.symlink(os.path.join(src, "08_subdiv_synth", "subdivs"), tmp_path / "scene" / "subdivs")
    with open(os.path.join(src, "08_subdiv_synth", "subdiv_synth.json")) as f:
        d = json.load(f)
    keep = [1, 2, 3, 4]
    d["shapes"] = [d["shapes"][i] for i in keep]
    d["instances"] = [dict(inst, shape=keep.index(inst["shape"])) for inst in d["instances"] if inst["shape"] in keep]
    d["subdivs"] = [dict(sd, shape=keep.index(sd["shape"])) for sd in d["subdivs"]]
    path = tmp_path / "scene" / "triangles.json"
    with open(path, "w") as f:
        json.dump(d, f)
    return str(path)


@pytest.mark.parametrize("which,shader,res,bounces,spp", [
    ("head", "volpathtrace", 640, 64, 4),      # 144 046 triangles, HBM-overflow stack instance
    ("head", "pathtrace", 640, 8, 4),
    ("subdiv", "pathtrace", 640, 8, 8),        # LDS-stack instance; shapes with and without normals / texcoords
    ("subdiv", "volpathtrace", 640, 16, 8),
    ("subdiv", "eyelight", 320, 4, 2),         # a shader without an instance for the short records: reads the general ones of the same scene
])
def test_compact_triangle_records_do_not_change_a_bit(vpt, monkeypatch, tmp_path, which, shader, res, bounces, spp):
    """A scene whose shapes all hold triangles keeps 48-byte leaf records and 64-byte attribute records beside the general 64 / 96-byte ones, and
    the path tracers' instances compiled for them read those (include/vpt.h: vpt_scene_record_bytes).  VPT_NO_COMPACT_TRIANGLES=1 (read when a
    scene handle is created) keeps the general records only: the frame must come out bit for bit the same."""
    scene_file = os.path.join(GOLDEN, "scenes", "05_head1ss_sub/head1ss_sub.json") if which == "head" else _scene_of_triangles(tmp_path)
    scene = vpt.HostScene(scene_file)
    p = vpt.PathtraceParams(resolution=res, samples=1 << 20, shader=shader, bounces=bounces)
    compact = vpt.DeviceScene(scene, 0)
    monkeypatch.setenv("VPT_NO_COMPACT_TRIANGLES", "1")
    general = vpt.DeviceScene(scene, 0)
    monkeypatch.delenv("VPT_NO_COMPACT_TRIANGLES")
    assert compact.record_bytes() == (48, 64) and general.record_bytes() == (64, 96)
    a, b = scene.make_state(p), scene.make_state(p)
    compact.pathtrace_samples(a, p, spp)
    general.pathtrace_samples(b, p, 1)
    general.pathtrace_samples(b, p, spp - 1)
    assert a.samples == b.samples == spp
    assert np.array_equal(a.image.view(np.uint32), b.image.view(np.uint32)) and np.array_equal(a.rngs, b.rngs) and np.array_equal(a.hits, b.hits)
    assert a.hits.sum() > 0


def test_scenes_with_quads_keep_the_general_records(vpt):
    dev = vpt.DeviceScene(vpt.HostScene(os.path.join(GOLDEN, "scenes", "03_volume/volume.json")), 0)
    assert dev.record_bytes() == (64, 96)


def test_intersect_through_compact_records(vpt, oracle, monkeypatch, tmp_path):
    """vpt_intersect takes the 48-byte leaf records on a scene of triangles (both stack variants of the traversal have an instance for them: the
    head scene runs the HBM-overflow one in test_intersect_is_bit_identical_on_edge_case_rays, this scene the LDS one): same hits, bit for bit,
    as through the general records and as the oracle's intersect_bvh - dense waves and waves with three rays (group forms of the leaf phase)."""
    scene = vpt.HostScene(_scene_of_triangles(tmp_path))
    compact = vpt.DeviceScene(scene, 0)
    monkeypatch.setenv("VPT_NO_COMPACT_TRIANGLES", "1")
    general = vpt.DeviceScene(scene, 0)
    monkeypatch.delenv("VPT_NO_COMPACT_TRIANGLES")
    assert compact.record_bytes()[0] == 48 and general.record_bytes()[0] == 64
    rays = _edge_rays(np.random.default_rng(11), np.float32([-0.7, -0.05, -0.3]), np.float32([0.7, 0.35, 0.3]), 20000)
    sparse = rays.copy()
    sparse[(np.arange(len(rays)) * 7 + 3) % 64 >= 3] = np.float32([1e3, 1e3, 1e3, 0.57735027, 0.57735027, 0.57735027])
    hits = 0
    for batch in (rays, sparse):
        for instance in (-1, 3):
            ids, uvt = compact.intersect(batch, instance)
            hits += int((ids[:, 0] >= 0).sum())
            gids, guvt = general.intersect(batch, instance)
            rids, ruvt = oracle.oracle_intersect(scene, batch, instance)
            assert np.array_equal(ids, gids) and np.array_equal(uvt.view(np.uint32), guvt.view(np.uint32))
            assert np.array_equal(ids, rids) and np.array_equal(uvt.view(np.uint32), ruvt.view(np.uint32))
    assert hits > 300   # the batches do hit things
