"""Known-answer tables: record generators, the three runners (reference = oracle/_ref/ref_tables,
oracle = libvpt_oracle.so, device = vpt_kat of libvpt_hip.so) and the comparison helpers.  Record layouts and op
numbers: include/vpt_kat.h.  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(GOLDEN, "scenes")
REF_TABLES = os.path.join(ROOT, "oracle", "_ref", "ref_tables")
TABLES = os.path.join(GOLDEN, "kat_tables.npz")

OPS = {  # name -> (op, floats in, floats out)
    "lobes": (0, 19, 22), "media": (1, 15, 10), "texture": (2, 4, 4), "camera": (3, 5, 6), "intersect": (4, 7, 5),
    "surface": (5, 7, 24), "environment": (6, 3, 3), "sample_lights": (7, 7, 3), "lights_pdf": (8, 6, 1),
    "lights_pdf_k2": (9, 6, 1), "sdf_scene": (10, 4, 3), "sdf_normal": (11, 6, 3), "spheretrace": (12, 7, 4),
    "volume": (13, 4, 1), "sdf_function": (14, 4, 1),
}
# ops whose arithmetic has no libm call on the path: the device must reproduce the reference bit for bit.
# (surface: eval_material takes a log for the density of refractive / volumetric materials only.)
EXACT_OPS = {"texture", "camera", "intersect", "sdf_scene", "sdf_normal", "spheretrace", "volume", "sdf_function"}

SCENE_FILES = {
    "03_volume": "03_volume/volume.json", "01_surface_min": "01_surface_min/surface_min.json",
    "05_head1ss_sub": "05_head1ss_sub/head1ss_sub.json", "06_gridsdf_synth": "06_gridsdf_synth/gridsdf_synth.json",
    "07_sdfunction_synth": "07_sdfunction_synth/sdfunction_synth.json", "03_volume_lobes": "03_volume_lobes/volume_lobes.json",
}
# (case name, scene key or None, op name, iparam): the committed tables
CASES = [
    ("lobes", None, "lobes", 0), ("media", None, "media", 0),
    ("texture_surface", "01_surface_min", "texture", 0),
    ("camera_sdfn", "07_sdfunction_synth", "camera", 0), ("camera_vol", "03_volume", "camera", 0),
    ("intersect_vol", "03_volume", "intersect", 0), ("intersect_head", "05_head1ss_sub", "intersect", 0),
    ("intersect_lobes", "03_volume_lobes", "intersect", 0),
    ("surface_surface", "01_surface_min", "surface", 0), ("surface_vol", "03_volume", "surface", 0),
    ("surface_head", "05_head1ss_sub", "surface", 0), ("surface_lobes", "03_volume_lobes", "surface", 0),
    ("environment_vol", "03_volume", "environment", 0), ("environment_head", "05_head1ss_sub", "environment", 0),
    ("sample_lights_vol", "03_volume", "sample_lights", 0), ("sample_lights_head", "05_head1ss_sub", "sample_lights", 0),
    ("sample_lights_sdf", "06_gridsdf_synth", "sample_lights", 0), ("sample_lights_lobes", "03_volume_lobes", "sample_lights", 0),
    ("lights_pdf_vol", "03_volume", "lights_pdf", 450), ("lights_pdf_head", "05_head1ss_sub", "lights_pdf", 450),
    ("lights_pdf_sdf", "06_gridsdf_synth", "lights_pdf", 450), ("lights_pdf_sdfn", "07_sdfunction_synth", "lights_pdf", 450),
    ("lights_pdf_lobes", "03_volume_lobes", "lights_pdf", 450),
    ("sdf_scene_sdf", "06_gridsdf_synth", "sdf_scene", 0), ("sdf_scene_sdfn", "07_sdfunction_synth", "sdf_scene", 0),
    ("sdf_normal_sdf", "06_gridsdf_synth", "sdf_normal", 0), ("sdf_normal_sdfn", "07_sdfunction_synth", "sdf_normal", 0),
    ("spheretrace_sdf", "06_gridsdf_synth", "spheretrace", 450), ("spheretrace_sdfn", "07_sdfunction_synth", "spheretrace", 450),
    ("spheretrace_sdfn_64", "07_sdfunction_synth", "spheretrace", 64),
    ("volume_sdf", "06_gridsdf_synth", "volume", 0), ("sdf_function_sdfn", "07_sdfunction_synth", "sdf_function", 0),
]


def scene_path(key):
    return os.path.join(SCENES, SCENE_FILES[key])


# ---- flattened-scene introspection (counts only) through the C-ABI structs ---------------------------------------
class _Frame(C.Structure):
    _fields_ = [("v", C.c_float * 12)]


class _Shape(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_vertices position_offset normal_offset texcoord_offset color_offset num_triangles "
                                         "triangle_offset num_quads quad_offset num_bvh_nodes bvh_node_offset bvh_prim_offset").split()]


class _Instance(C.Structure):
    _fields_ = [("frame", _Frame), ("shape", C.c_int32), ("material", C.c_int32)]


class _Desc(C.Structure):   # the head of vpt_scene_desc (include/vpt.h): the ten typed tables
    _fields_ = sum(([(f"num_{n}", C.c_int32), (n, C.c_void_p)] for n in
                    "cameras instances shapes materials textures environments volumes vol_instances sdfs lights".split()), [])


def scene_info(host_scene):
    d = _Desc.from_address(host_scene.desc)
    inst = (_Instance * d.num_instances).from_address(d.instances) if d.num_instances else []
    shp = (_Shape * d.num_shapes).from_address(d.shapes) if d.num_shapes else []
    return {
        "cameras": d.num_cameras, "instances": d.num_instances, "textures": d.num_textures, "volumes": d.num_volumes,
        "vol_instances": d.num_vol_instances, "sdfs": d.num_sdfs, "lights": d.num_lights,
        "inst_elems": [shp[i.shape].num_triangles or shp[i.shape].num_quads for i in inst],
        "inst_tri": [shp[i.shape].num_triangles != 0 for i in inst],
    }


# ---- record generators -------------------------------------------------------------------------------------------
def _unit(rng, n):
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def _u01(rng, shape):
    """uniform floats the way rand1f makes them (multiples of 2^-23 in [0, 1)), with the ends of the range sprinkled in"""
    v = (rng.integers(0, 1 << 23, size=shape).astype(np.float32) / np.float32(1 << 23)).astype(np.float32)
    flat = v.reshape(-1)
    k = max(1, flat.size // 64)
    flat[rng.integers(0, flat.size, size=k)] = 0.0
    flat[rng.integers(0, flat.size, size=k)] = np.float32(1 - 2.0 ** -23)
    return v


def gen_lobes(rng):
    rough = np.float32([0.0, 0.03 * 0.03, 0.01, 0.09, 0.25, 1.0])
    rows = []
    for t in range(8):
        for r in rough:
            n = 96
            rec = np.zeros((n, 19), np.float32)
            rec[:, 0] = t
            rec[:, 1:4] = rng.uniform(0.02, 1.0, size=(n, 3))
            rec[: n // 8, 1:4] = 0.0                           # black: zero specular (fresnel_schlick early out)
            rec[:, 4] = r
            rec[:, 5] = rng.choice(np.float32([0, 0.3, 1.0]), size=n)
            rec[:, 6] = rng.choice(np.float32([1.5, 1.33, 1.0005, 2.4]), size=n)
            rec[:, 7:10] = _unit(rng, n)
            rec[:, 10:13] = _unit(rng, n)                      # outgoing on either side of the surface
            rec[:, 13] = _u01(rng, n)
            rec[:, 14:16] = _u01(rng, (n, 2))
            rec[:, 16:19] = _unit(rng, n)
            rows.append(rec)
    return np.concatenate(rows)


def gen_media(rng, n=2048):
    rec = np.zeros((n, 15), np.float32)
    rec[:, 0:3] = rng.uniform(0.0, 400.0, size=(n, 3))
    rec[rng.integers(0, n, size=n // 16), rng.integers(0, 3, size=n // 16)] = 0.0   # a zero-density channel
    rec[:, 3] = rng.uniform(1e-3, 2.0, size=n)
    rec[:, 4] = _u01(rng, n)
    rec[:, 5] = _u01(rng, n)
    rec[:, 6] = rng.choice(np.float32([-0.8, 0.0, 0.0005, 0.3, 0.9]), size=n)
    rec[:, 7:10] = _unit(rng, n)
    rec[:, 10:12] = _u01(rng, (n, 2))
    rec[:, 12:15] = _unit(rng, n)
    return rec


def gen_texture(rng, info, n=4096):
    rec = np.zeros((n, 4), np.float32)
    rec[:, 0] = rng.integers(0, info["textures"], size=n)
    rec[:, 1:3] = rng.uniform(-3.0, 3.0, size=(n, 2))
    rec[: n // 8, 1:3] = np.round(rec[: n // 8, 1:3] * 4) / 4     # texel-aligned and integer coordinates, both signs
    rec[:, 3] = rng.integers(0, 2, size=n)
    return rec


def gen_camera(rng, info, n=1024):
    rec = np.zeros((n, 5), np.float32)
    rec[:, 0] = rng.integers(0, info["cameras"], size=n)
    rec[:, 1:5] = _u01(rng, (n, 4))
    return rec


def edge_rays(rng, lo, hi, n):
    """Rays path tracing rarely produces: axis-aligned and one-zero-component directions, origins on the
    coordinate planes box faces tend to lie on (0*inf in the slab test), denormal direction components."""
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    kind = np.arange(n) % 8
    axis = rng.integers(0, 3, size=n)
    sign = rng.choice(np.float32([-1, 1]), size=n)
    for i in range(n):
        k, a = kind[i], axis[i]
        if k == 1:      # axis-aligned direction
            d[i] = 0
            d[i, a] = sign[i]
        elif k == 2:    # one zero component
            d[i, a] = 0
        elif k == 3:    # zero component and the origin on that coordinate plane (grazes box faces at 0)
            d[i, a] = 0
            o[i, a] = 0
        elif k == 4:    # denormal component: 1/d overflows
            d[i, a] = np.float32(1e-40) * sign[i]
        elif k == 5:    # tiny but normal component
            d[i, a] = np.float32(1e-30) * sign[i]
        elif k == 6:    # origin on a plane, direction not in it
            o[i, a] = 0
    return np.concatenate([o, d], axis=1)


def gen_intersect(rng, info, lo, hi, n=6000):
    rays = np.zeros((n, 7), np.float32)
    rays[:, 0:3] = rng.uniform(lo, hi, size=(n, 3))
    rays[:, 3:6] = _unit(rng, n)
    rays[n // 2:, 0:6] = edge_rays(rng, lo, hi, n - n // 2)
    rays[:, 6] = -1
    k = n // 4
    rays[:k, 6] = rng.integers(0, info["instances"], size=k)       # single-instance queries (sample_lights_pdf's kind)
    return rays


def gen_surface(rng, info, n=3072):
    rec = np.zeros((n, 7), np.float32)
    inst = rng.integers(0, info["instances"], size=n)
    rec[:, 0] = inst
    elems = np.array(info["inst_elems"])[inst]
    rec[:, 1] = (rng.random(n) * elems).astype(np.int64)
    uv = _u01(rng, (n, 2))
    tri = np.array(info["inst_tri"])[inst]
    fold = tri & (uv.sum(axis=1) > 1)                              # barycentric coordinates of a triangle
    uv[fold] = (1 - uv[fold]).astype(np.float32)
    rec[:, 2:4] = uv
    rec[:, 4:7] = _unit(rng, n)
    return rec


def gen_dirs(rng, n=2048):
    d = _unit(rng, n)
    d[:6] = np.float32([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])   # poles and the atan2 seam
    return d


def gen_sample_lights(rng, lo, hi, n=3072):
    rec = np.zeros((n, 7), np.float32)
    rec[:, 0:3] = rng.uniform(lo, hi, size=(n, 3))
    rec[:, 3:7] = _u01(rng, (n, 4))
    return rec


def gen_points(rng, lo, hi, n):
    return rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)


# ---- runners -----------------------------------------------------------------------------------------------------
def have_reference():
    return os.path.exists(REF_TABLES)


def run_reference(scene_key, op_name, iparam, records):
    op, si, so = OPS[op_name]
    records = np.ascontiguousarray(records, np.float32)
    assert records.shape[1] == si
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        records.tofile(fin)
        subprocess.check_call([REF_TABLES, scene_path(scene_key) if scene_key else "-", str(op), str(iparam), fin, fout])
        return np.fromfile(fout, np.float32).reshape(records.shape[0], so)


def run_oracle(oracle_lib, host_scene, op_name, iparam, records):
    op, si, so = OPS[op_name]
    records = np.ascontiguousarray(records, np.float32)
    out = np.zeros((records.shape[0], so), np.float32)
    lib = oracle_lib.lib()
    lib.vpt_oracle_kat.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rc = lib.vpt_oracle_kat(host_scene.desc if host_scene is not None else None, op, iparam, records.shape[0],
                            records.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"vpt_oracle_kat({op_name}) failed: {rc}")
    return out


# ---- comparison ----------------------------------------------------------------------------------------------------
def bits_equal(a, b):
    """bitwise equality where NaNs of any payload count as equal (the sign / payload of a NaN is not part of the contract)"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def ulp_distance(a, b):
    """distance in float32 units in the last place (0 for NaN-vs-NaN, huge for NaN-vs-number or a sign flip across zero)"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)

    def key(x):   # monotone integer image of the float line
        i = x.view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7fffffff), i)
    d = np.abs(key(a) - key(b))
    both_nan = np.isnan(a) & np.isnan(b)
    one_nan = np.isnan(a) ^ np.isnan(b)
    return np.where(both_nan, 0, np.where(one_nan, 1 << 40, d))
