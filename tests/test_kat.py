"""Known-answer tables.  tests/golden/kat_tables.npz holds, for every case of kat_lib.CASES, a
batch of argument records and what the REFERENCE'S OWN function of that name returned for them
(tests/golden/make_kat.py -> oracle/_ref/ref_tables, which #includes the reference's yocto_pathtrace.cpp to reach its
file-local eval_bsdfcos / sample_lights / ... and calls the public yocto_scene / yocto_bvh / yocto_sdfs API for the rest).

 * CPU: the oracle's functions must reproduce every table bit for bit (same g++, same glibc, same operation order);
 * GPU: vpt_kat() runs the kernels' own device functions on the same records.  Ops without a libm call on the path
   (intersect, spheretrace, eval_sdf*, eval_volume, sd_*, eval_camera, eval_texture) must be bit-exact; the others are
   held to a stated number of float32 ulps (device libm is ocml, the reference's is glibc; both are within 1-2 ulp of
   the true value per call and the lobes chain several calls)."""
import os

import numpy as np
import pytest

import kat_lib as K

CASE_IDS = [c[0] for c in K.CASES]


@pytest.fixture(scope="module")
def tables():
    return np.load(K.TABLES)


@pytest.fixture(scope="module")
def host_scenes(vpt):
    cache = {}

    def get(key):
        if key not in cache:
            cache[key] = vpt.HostScene(K.scene_path(key))
        return cache[key]
    return get


def test_tables_cover_every_lobe_and_every_sdf_primitive(tables):
    """what the verdict of round 1 found unpinned: reflective / transparent / gltfpbr lobes, sd_sphere / sd_bbox / sd_torus /
    sd_capped_cone / sd_plane"""
    lob = tables["lobes_in"]
    assert set(np.unique(lob[:, 0]).astype(int)) == set(range(8))                 # all 8 material types ...
    for t in range(8):
        assert {0.0} < set(np.unique(lob[lob[:, 0] == t, 4]).tolist())            # ... delta and rough
    import json
    sdfs = json.load(open(K.scene_path("07_sdfunction_synth")))["sdfunctions"]
    assert {s["type"] for s in sdfs} == {"box", "bbox", "capped_cone", "plane", "sphere", "torus"}
    fn = tables["sdf_function_sdfn_in"]
    assert set(np.unique(fn[:, 0]).astype(int)) == set(range(len(sdfs)))
    hit = tables["spheretrace_sdfn_out"]
    assert set(np.unique(hit[hit[:, 0] != 0, 3]).astype(int)) >= set(range(len(sdfs))) - {-1}   # every SDF gets hit by some ray


@pytest.mark.parametrize("case", CASE_IDS)
def test_oracle_reproduces_reference_tables(oracle, host_scenes, tables, case):
    _, key, op, iparam = K.CASES[CASE_IDS.index(case)]
    got = K.run_oracle(oracle, host_scenes(key) if key else None, op, iparam, tables[case + "_in"])
    ok = K.bits_equal(got, tables[case + "_out"])
    assert ok.all(), (case, np.argwhere(~ok)[:5].tolist())


def test_tables_are_what_the_reference_returns_now(tables):
    """build container only: the committed tables are regenerated from the reference build and must be unchanged"""
    if not K.have_reference() or not os.path.exists("/root/reference"):
        pytest.skip("oracle/_ref/ref_tables or /root/reference not present; the tables are committed data")
    for name, key, op, iparam in K.CASES[::4]:
        out = K.run_reference(key, op, iparam, tables[name + "_in"])
        assert K.bits_equal(out, tables[name + "_out"]).all(), name


def test_sdf_light_pdf_is_ill_conditioned_in_the_reference(oracle, host_scenes, tables):
    """Root cause of the same-stream / different-radiance pixels of the implicit shader.
    sample_lights_pdf takes an SDF light's normal with eval_sdf_normal(sdf, POSITION, dist) (yocto_pathtrace.cpp:389):
    four taps h = flt_eps * dist around the SHADING point, not around the point on the light.  There the SDF value is
    the distance to the light (~0.1 .. 1), its float spacing is as large as the tap offsets' effect h * |gradient|, so
    the four differences are 0 or +-1 ulp of rounding noise and the 'normal' is one of a handful of directions that
    has nothing to do with the light and flips with the last bit of `position`.  `position` is the previous vertex
    o + d * t, so from the second bounce on it carries the last-bit differences of the device's sin / cos / atan: the
    pdf, and with it the path weight, changes by a factor of order one while the pixel consumes exactly the same
    random numbers (roulette only starts after bounce 3).  Shown here on the oracle (== reference bit for bit, test
    above): the same queries with each coordinate of the position moved by ONE float ulp up or down."""
    sc = host_scenes("06_gridsdf_synth")
    rec = tables["lights_pdf_sdf_in"].copy()
    base = K.run_oracle(oracle, sc, "lights_pdf", 450, rec)[:, 0]
    bumped = rec.copy()
    toward = np.where(np.random.default_rng(5).random((len(rec), 3)) < 0.5, np.inf, -np.inf).astype(np.float32)
    bumped[:, 0:3] = np.nextafter(rec[:, 0:3], toward)
    moved = K.run_oracle(oracle, sc, "lights_pdf", 450, bumped)[:, 0]
    # which queries hit the SDF light: single-SDF sphere trace of the same rays (sdf 1 = the box light of 06_gridsdf)
    import json
    scene_json = json.load(open(K.scene_path("06_gridsdf_synth")))
    emissive = [i for i, m in enumerate(scene_json["materials"]) if any(m.get("emission", [0]))]
    light = [i for i, f in enumerate(scene_json["sdfunctions"]) if f["material"] in emissive][0]
    rays = np.concatenate([rec, np.full((len(rec), 1), float(light), np.float32)], axis=1)
    light_hit = K.run_oracle(oracle, sc, "spheretrace", 450, rays)[:, 0] != 0
    assert light_hit.sum() > 100
    rel = np.abs(moved - base) / np.maximum(np.abs(base), 1e-20)
    # rays that miss the SDF light: the pdf (an environment-map texel) does not depend on the position at all
    assert (rel[~light_hit] == 0).all()
    # rays that hit it: a large share moves by more than the 2e-3 the parity tests allow, some by more than 10 %
    # (a well-conditioned pdf would move by ~1e-7)
    over, big = (rel[light_hit] > 2e-3).mean(), (rel[light_hit] > 0.1).mean()
    print(f"SDF-light hits: {light_hit.sum()}; pdf moves by > 2e-3 under a 1-ulp nudge of the position: {over:.2f}, by > 10 %: {big:.3f}")
    assert over > 0.25 and big > 0.01


# ---- GPU: the kernels' own device functions against the same tables ------------------------------------------------
# Per op with libm on its path: (float32 ulps, relative tolerance, minimum share of outputs within either).  The
# relative tolerance covers outputs that are well-conditioned but not to the ulp: a bilinear environment lookup turns
# one ulp of atan2 / acos (6e-8 of a texture coordinate) into 1.2e-4 of a texel at 2048 texels, i.e. up to ~1e-4 of the
# interpolated radiance; the Henyey-Greenstein denominator 1 + g^2 - 2 g cos cancels to (1 - g)^2 in the forward
# peak (g = 0.9: 150 ulp per ulp of the cosine).  The share is below 1 only where a last-bit difference moves a
# DISCRETE choice and the output jumps: the Fresnel coin of a lobe sample (rnl < F) or the lobe a sampled direction
# lands in (measured: 0.11 % of the lobe outputs).
TOLERANCE = {
    "lobes": (64, 1e-5, 0.995), "media": (16, 2e-5, 1.0), "surface": (8, 0.0, 1.0), "environment": (8, 5e-4, 1.0),
    "sample_lights": (8, 5e-5, 1.0), "lights_pdf": (32, 0.0, 1.0),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASE_IDS)
def test_device_reproduces_reference_tables(vpt, host_scenes, tables, case):
    _, key, op, iparam = K.CASES[CASE_IDS.index(case)]
    dev = vpt.DeviceScene(host_scenes(key or "03_volume"), 0)
    rec, ref = tables[case + "_in"], tables[case + "_out"]
    ops = [op] if op != "lights_pdf" else ["lights_pdf", "lights_pdf_k2"]   # K1's and K2's code path of sample_lights_pdf
    for o in ops:
        got = dev.kat(K.OPS[o][0], rec, iparam)
        if op in K.EXACT_OPS:
            ok = K.bits_equal(got, ref)
            assert ok.all(), (case, o, int((~ok).sum()), np.argwhere(~ok)[:5].tolist())
            continue
        ulps, rel, share = TOLERANCE[op]
        d = K.ulp_distance(got, ref)
        with np.errstate(invalid="ignore"):
            err = np.abs(got - ref)
            # absolute floor 1e-6 for values that are sums / differences near zero (a cancelling dot product has no ulp meaning)
            close = (d <= ulps) | (err <= np.maximum(rel * np.abs(ref), 1e-6 * np.maximum(1.0, np.abs(ref))))
        frac = close.mean()
        worst = d[~close].max() if (~close).any() else d.max()
        print(f"{case}/{o}: within {ulps} ulp or {rel:g} relative: {frac:.5f}; exact: {(d == 0).mean():.4f}; worst ulp distance {worst}")
        assert frac >= share, (case, o, frac, np.argwhere(~close)[:5].tolist())


@pytest.mark.gpu
def test_named_entry_points(vpt, host_scenes, tables):
    """vpt_spheretrace / vpt_eval_lobes (include/vpt_kat.h) are the same ops under their own names"""
    dev = vpt.DeviceScene(host_scenes("07_sdfunction_synth"), 0)
    rec, ref = tables["spheretrace_sdfn_in"], tables["spheretrace_sdfn_out"]
    whole = rec[:, 6] < 0
    ids, t = dev.spheretrace(rec[whole, :6], -1, 450)
    assert np.array_equal(ids[:, 0], ref[whole, 0].astype(np.int32)) and np.array_equal(ids[:, 1:], ref[whole, 2:].astype(np.int32))
    assert np.array_equal(t.view(np.uint32), ref[whole, 1].view(np.uint32))
    import ctypes as C
    lob = np.ascontiguousarray(tables["lobes_in"][:256])
    out = np.zeros((256, 22), np.float32)
    assert vpt.hip.vpt_eval_lobes(dev.handle, 256, lob.ctypes.data, out.ctypes.data) == 0
    assert np.array_equal(out.view(np.uint32), dev.kat(0, lob).view(np.uint32))
    # ids are range-checked on the host: a batch can never index outside the scene's tables
    bad = rec[:4].copy()
    bad[0, 6] = 99
    with pytest.raises(vpt.VptError):
        dev.kat(K.OPS["spheretrace"][0], bad, 450)
    assert C.c_int(0).value == 0
