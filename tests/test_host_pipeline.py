"""CPU tests of the host side of the boundary: scene loading / BVH / lights / state seeding
against vectors produced by the reference itself (tests/golden/make_fixtures.py), the C-ABI
surface, error behaviour, and the output stage (sRGB quantisation + stb-compatible JPEG)."""
import ctypes as C
import io
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, SCENE_03


def test_pcg32_known_answers(vpt, scene03):
    """SURVEY.md §8(c) KATs measured on the reference: per-pixel stream seeds of make_state
    (yocto_pathtrace.cpp:975-978) — integer arithmetic, must be bit-exact."""
    st = scene03.make_state(vpt.PathtraceParams(resolution=1280))
    assert (st.width, st.height) == (1280, 533)          # height from camera aspect 2.4, not 720
    rng = st.rngs.reshape(-1, 2)
    assert int(rng[0, 0]) == 0x915A80C374CCB637 and int(rng[0, 1]) == 0x56710D81
    assert int(rng[1, 0]) == 0x11D3F0758FEC97BB and int(rng[1, 1]) == 0x50EA1B0F
    # inc = (seq << 1) | 1 with seq = rand1i(master, 1<<31)/2 + 1
    seqs = [(int(rng[i, 1]) - 1) // 2 for i in range(4)]
    assert seqs == [725124800, 678759815, 790335339, 839039096]
    st2 = scene03.make_state(vpt.PathtraceParams(resolution=720))
    assert (st2.width, st2.height) == (720, 300)


def test_first_floats_of_pixel_streams(vpt, scene03):
    """rand1f of pixel 0 / 1: 0x1.25745p-3, 0x1.9893ep-1 / 0x1.1d3f4p-1, 0x1.b88944p-1"""
    st = scene03.make_state(vpt.PathtraceParams(resolution=1280))

    def draws(state, inc, n):
        out = []
        for _ in range(n):
            old = state
            state = (old * 6364136223846793005 + inc) & 0xFFFFFFFFFFFFFFFF
            xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
            rot = old >> 59
            u = ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF
            out.append(np.array([(u >> 9) | 0x3F800000], np.uint32).view(np.float32)[0] - np.float32(1))
        return out
    rng = st.rngs.reshape(-1, 2)
    assert draws(int(rng[0, 0]), int(rng[0, 1]), 2) == [float.fromhex("0x1.25745p-3"), float.fromhex("0x1.9893ep-1")]
    assert draws(int(rng[1, 0]), int(rng[1, 1]), 2) == [float.fromhex("0x1.1d3f4p-1"), float.fromhex("0x1.b88944p-1")]


def test_scene_bvh_lights_match_reference_hashes(scene03):
    """Sizes and FNV-1a hashes of every array the hot path reads (vertex data, texels, both BVH
    levels incl. node order, light CDFs) equal the reference's own load_scene/make_bvh/make_lights."""
    golden = json.load(open(os.path.join(GOLDEN, "03_volume_stats.json")))
    mine = json.loads(scene03.stats())
    assert mine == golden
    assert golden["scene_bvh"]["nodes"] == 5 and golden["shapes"][1]["bvh_nodes"] == 3903
    assert golden["lights"][2]["cdf_len"] == 2097152


def test_capi_exports_every_declared_symbol(vpt):
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):          # every header under include/: vpt.h, vpt_kat.h, ...
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", h)).read(), flags=re.S)   # prototypes, not prose
        names |= set(re.findall(r"\b(vpt_[a-z_0-9]+)\s*\(", text))
    names -= {"vpt_status"}
    assert {"vpt_scene_create", "vpt_render", "vpt_render_device", "vpt_resolve_device", "vpt_kat", "vpt_spheretrace"} <= names
    for name in sorted(names):
        assert hasattr(vpt.hip, name), f"libvpt_hip.so does not export {name}"


def test_scene_create_rejects_bad_descriptors(vpt, scene03):
    """validation runs before any device work, so these are CPU tests: no faulting kernels."""
    out = C.c_void_p()
    assert vpt.hip.vpt_scene_create(None, 0, C.byref(out)) == -1
    # corrupt a copy of the descriptor: instance material out of range

    class Desc(C.Structure):
        _fields_ = [("num_cameras", C.c_int32), ("cameras", C.c_void_p), ("num_instances", C.c_int32), ("instances", C.c_void_p)]
    raw = (C.c_char * 512).from_address(scene03.desc)
    buf = C.create_string_buffer(bytes(raw), 512)
    d = Desc.from_buffer(buf)
    inst = np.frombuffer((C.c_char * (56 * d.num_instances)).from_address(d.instances), np.int32).reshape(-1, 14).copy()
    inst[3, 13] = 99  # vpt_instance.material
    d.instances = inst.ctypes.data
    assert vpt.hip.vpt_scene_create(C.addressof(buf), 0, C.byref(out)) == -1
    assert b"instance 3: bad material" in vpt.hip.vpt_last_error()


def test_unknown_shader_is_an_error(vpt):
    with pytest.raises(vpt.VptError, match="sampler unknown"):
        vpt.PathtraceParams(shader="bogus").to_abi()


def test_load_scene_errors(vpt, tmp_path):
    with pytest.raises(vpt.VptError, match="file not found"):
        vpt.HostScene(str(tmp_path / "missing.json"))
    bad = tmp_path / "bad.json"
    bad.write_text('{"asset": {"version": "4.2"}, "cameras": [{"lens": "x"}]}')
    with pytest.raises(vpt.VptError, match="parse error"):
        vpt.HostScene(str(bad))


def test_jpeg_writer_is_byte_identical_to_reference(vpt):
    """The reference's own save_image(.jpg) output for a 128x53x8spp render vs our pipeline from the
    reference's float state: get_render -> rgb_to_srgb -> float_to_byte -> baseline JPEG q75."""
    st = np.load(os.path.join(GOLDEN, "03_volume_128_8_state.npz"))
    w, h, spp = (int(x) for x in st["meta"])
    rgba8 = vpt.linear_to_srgb8(st["image"], spp)
    mine = vpt.encode_jpeg_q75(rgba8)
    ref = open(os.path.join(GOLDEN, "03_volume_128_8.jpg"), "rb").read()
    assert mine == ref


def test_jpeg_roundtrip_decodes(vpt):
    from PIL import Image
    rng = np.random.default_rng(1)
    rgba = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)  # ragged size: edge replication
    img = np.asarray(Image.open(io.BytesIO(vpt.encode_jpeg_q75(rgba))).convert("RGB"))
    assert img.shape == (37, 53, 3)


def test_byte_to_float_in_three_instructions_is_the_division():
    """csrc/vpt_scene.hip.h: byte_to_float(b) = fma(fma(-255, q, b), r, q) with r = fl(1 / 255), q = fl(b * r) stands in for the reference's
    b / 255.0f (yocto_color.h:212-214) in texture fetches.  Checked here in exact rational arithmetic with one correct rounding per operation
    (what the device's v_mul_f32 / v_fma_f32 do): equal to the IEEE quotient for every byte."""
    from fractions import Fraction
    import math

    def rnd(fr):   # round to nearest even float32, exactly
        if fr == 0:
            return np.float32(0)
        a, e = abs(fr), math.floor(math.log2(abs(fr))) - 23
        while a / Fraction(2) ** e >= 2 ** 24:
            e += 1
        while a / Fraction(2) ** e < 2 ** 23:
            e -= 1
        x = a / Fraction(2) ** e
        m = x.numerator // x.denominator
        rem = x - m
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (m & 1)):
            m += 1
        return np.float32(math.copysign(float(m) * 2.0 ** e, fr))

    r = np.float32(1.0) / np.float32(255.0)
    assert r.view(np.uint32) == 0x3B808081
    for b in range(256):
        q = rnd(Fraction(b) * Fraction(float(r)))
        step = rnd(Fraction(b) - 255 * Fraction(float(q)))
        got = rnd(Fraction(float(step)) * Fraction(float(r)) + Fraction(float(q)))
        assert got.view(np.uint32) == (np.float32(b) / np.float32(255.0)).view(np.uint32), b
