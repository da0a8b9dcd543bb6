"""vpt-mi355x: ctypes bindings over the C-ABI of ``include/vpt.h`` (libvpt_hip.so, the HIP
path) and over the host library (libvpt_host.so: scene.json loader, BVH / light / state builders
that mirror ``libs/yocto_pathtrace/yocto_pathtrace.h:119-139`` of the reference).

Nothing in this package computes radiance on the CPU and nothing here touches ``oracle/``: if
the HIP library is missing or no GPU is present, rendering raises ``VptError``.

The directory name contains a dash, so import it with ``vpt_loader.load()`` (repo root) or
``importlib``; inside, everything is ordinary Python.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

SHADER_NAMES = ["volpathtrace", "pathtrace", "naive", "eyelight", "normal", "texcoord", "color",
                "implicit", "implicit_normal"]  # yocto_pathtrace.h:101-103


class VptError(RuntimeError):
    pass


def _lib(name: str) -> C.CDLL:
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise VptError(f"{path} is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


hip = _lib(os.environ.get("VPT_HIP_LIB", "libvpt_hip.so"))    # must load first: libvpt_host.so links against it
# (VPT_HIP_LIB selects an alternative build of the same ABI for A/B experiments)
host = _lib("libvpt_host.so")


class VptParams(C.Structure):  # vpt_params
    _fields_ = [("camera", C.c_int32), ("resolution", C.c_int32), ("shader", C.c_int32),
                ("samples", C.c_int32), ("bounces", C.c_int32), ("noparallel", C.c_int32),
                ("noimplicit_mis", C.c_int32), ("spheretrace_maxiter", C.c_int32)]


class VptLayout(C.Structure):  # vpt_layout
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("tile_w", C.c_int32),
                ("tile_h", C.c_int32), ("rank", C.c_int32), ("nranks", C.c_int32)]


@dataclass
class PathtraceParams:
    """pathtrace_params, yocto_pathtrace.h:87-99 (same names, same defaults)."""
    camera: int = 0
    resolution: int = 720
    shader: str = "pathtrace"
    samples: int = 512
    bounces: int = 4
    noparallel: bool = False
    noimplicit_mis: bool = False
    spheretrace_maxiter: int = 450

    def to_abi(self) -> VptParams:
        if self.shader not in SHADER_NAMES:
            raise VptError("sampler unknown")  # reference: get_shader throws (cpp:947-950)
        return VptParams(self.camera, self.resolution, SHADER_NAMES.index(self.shader), self.samples,
                         self.bounces, int(self.noparallel), int(self.noimplicit_mis),
                         self.spheretrace_maxiter)


@dataclass
class PathtraceState:
    """pathtrace_state, yocto_pathtrace.h:57-64: row-major host arrays."""
    width: int
    height: int
    samples: int = 0
    image: np.ndarray = field(default=None)  # (h, w, 4) float32 running sums
    hits: np.ndarray = field(default=None)   # (h, w) int32
    rngs: np.ndarray = field(default=None)   # (h, w, 2) uint64 {state, inc}

    def copy(self) -> "PathtraceState":
        return PathtraceState(self.width, self.height, self.samples, self.image.copy(), self.hits.copy(),
                              self.rngs.copy())


# ---- prototypes -----------------------------------------------------------------------------
_p = C.c_void_p
hip.vpt_last_error.restype = C.c_char_p
hip.vpt_version.restype = C.c_char_p
hip.vpt_device_count.restype = C.c_int
hip.vpt_scene_create.argtypes = [_p, C.c_int, C.POINTER(_p)]
hip.vpt_scene_destroy.argtypes = [_p]
hip.vpt_scene_destroy.restype = None
hip.vpt_render.argtypes = [_p, C.POINTER(VptParams), C.c_int, C.c_int, C.c_int, _p, _p, _p, C.POINTER(C.c_int)]
hip.vpt_layout_slots.argtypes = [C.POINTER(VptLayout)]
hip.vpt_layout_slots.restype = C.c_int64
hip.vpt_state_upload.argtypes = [C.POINTER(VptLayout), _p, _p, _p, _p, _p, _p, _p]
hip.vpt_state_download.argtypes = [C.POINTER(VptLayout), _p, _p, _p, _p, _p, _p, _p]
hip.vpt_render_device.argtypes = [_p, C.POINTER(VptParams), C.POINTER(VptLayout), C.c_int, _p, _p, _p, _p]
hip.vpt_resolve_device.argtypes = [C.POINTER(VptLayout), _p, C.c_int, _p, _p]
hip.vpt_last_kernel_ms.argtypes = [_p, C.POINTER(C.c_float)]
hip.vpt_intersect.argtypes = [_p, C.c_int, _p, C.c_int, _p, _p]
hip.vpt_build_bvh.argtypes = [C.c_int, _p, C.c_int, _p, C.c_int, C.POINTER(C.c_int), _p]
hip.vpt_last_wave_costs.argtypes = [_p, _p, C.c_int, C.POINTER(C.c_int)]
hip.vpt_scene_record_bytes.argtypes = [_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
hip.vpt_multi_create.argtypes = [_p, C.POINTER(C.c_int), C.c_int, C.POINTER(_p)]
hip.vpt_multi_destroy.argtypes = [_p]
hip.vpt_multi_destroy.restype = None
hip.vpt_multi_device_count.argtypes = [_p]
hip.vpt_multi_render.argtypes = [_p, C.POINTER(VptParams), C.c_int, C.c_int, C.c_int, _p, _p, _p, C.POINTER(C.c_int)]
hip.vpt_multi_get_render.argtypes = [_p, _p]
hip.vpt_multi_uploaded_parts.argtypes = [_p]
hip.vpt_multi_transport.argtypes = [_p]
hip.vpt_multi_transport.restype = C.c_char_p
hip.vpt_multi_set_state.argtypes = [_p, C.c_int, C.c_int, _p, _p, _p, C.c_int]
hip.vpt_multi_get_state.argtypes = [_p, _p, _p, _p, C.POINTER(C.c_int)]
hip.vpt_check_watchdog.argtypes = [_p]
hip.vpt_resolve_srgb8_device.argtypes = [C.POINTER(VptLayout), _p, C.c_int, _p, _p]
hip.vpt_selftest_reciprocal.argtypes = [C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
hip.vpt_selftest_light_cdf.argtypes = [_p, C.c_int, C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]
hip.vpt_kat_strides.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
hip.vpt_kat.argtypes = [_p, C.c_int, C.c_int, C.c_int, _p, _p]
hip.vpt_spheretrace.argtypes = [_p, C.c_int, _p, C.c_int, C.c_int, _p, _p]
hip.vpt_eval_lobes.argtypes = [_p, C.c_int, _p, _p]
host.vpth_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
host.vpth_scene_load.restype = _p
host.vpth_scene_load_ex.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
host.vpth_scene_load_ex.restype = _p
host.vpth_vertex_normals.argtypes = [_p, C.c_int, _p, C.c_int, C.c_int, C.c_int, _p, C.c_char_p, C.c_int]
host.vpth_displace_vertices.argtypes = [_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _p, _p, _p, C.c_int, C.c_int, _p, C.c_char_p, C.c_int]
host.vpth_catmullclark.argtypes = [_p, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, C.POINTER(C.c_int), _p, C.POINTER(C.c_int), C.c_char_p, C.c_int]
host.vpth_scene_free.argtypes = [_p]
host.vpth_scene_free.restype = None
host.vpth_scene_desc.argtypes = [_p]
host.vpth_scene_desc.restype = _p
host.vpth_state_size.argtypes = [_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
host.vpth_make_state.argtypes = [_p, C.c_int, C.c_int, _p, _p, _p]
host.vpth_scene_stats.argtypes = [_p, C.c_char_p, C.c_int]
host.vpth_scene_rebuild_bvh_device.argtypes = [_p, C.c_int, C.c_char_p, C.c_int]
host.vpth_build_bvh_host.argtypes = [_p, C.c_int, _p, C.POINTER(C.c_int), _p]
host.vpth_linear_to_srgb8.argtypes = [C.c_int, C.c_int, _p, C.c_int, _p]
host.vpth_linear_to_srgb8.restype = None
host.vpth_encode_jpeg_q75.argtypes = [C.c_int, C.c_int, _p, _p, C.c_int64]
host.vpth_encode_jpeg_q75.restype = C.c_int64


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise VptError(f"{what} failed ({rc}): {hip.vpt_last_error().decode()}")


def device_count() -> int:
    return hip.vpt_device_count()


BVH_NODE = np.dtype([("bbox_min", np.float32, 3), ("bbox_max", np.float32, 3), ("start", np.int32), ("num", np.int16), ("axis", np.int8),
                     ("internal", np.uint8)])
assert BVH_NODE.itemsize == 32


def build_bvh(bboxes: np.ndarray, device: Optional[int] = 0):
    """build_bvh(bvh, bboxes, false) of the reference over n boxes {min.xyz, max.xyz}: (nodes, primitives).  device = GPU index
    (vpt_build_bvh) or None for the host build the loader uses (same arrays)."""
    bboxes = np.ascontiguousarray(bboxes, np.float32).reshape(-1, 6)
    n = len(bboxes)
    nodes = np.zeros(max(1, 2 * n), BVH_NODE)
    prims = np.zeros(max(1, n), np.int32)
    count = C.c_int()
    if device is None:
        host.vpth_build_bvh_host(bboxes.ctypes.data, n, nodes.ctypes.data, C.byref(count), prims.ctypes.data)
    else:
        _check(hip.vpt_build_bvh(device, bboxes.ctypes.data, n, nodes.ctypes.data, len(nodes), C.byref(count), prims.ctypes.data), "vpt_build_bvh")
    return nodes[:count.value].copy(), prims[:n].copy()


def catmullclark(quads: np.ndarray, verts: np.ndarray, lock_boundary: bool = False, device: Optional[int] = None):
    """one level of the reference's tesselate_catmullclark (yocto_pathtrace.cpp:1119-1226) on an (n, 4) int32 quad array (z == w: a
    triangle) and an (m, 2 | 3) float32 vertex array: (new quads, new vertices).  device: GPU index for the vertex arithmetic."""
    quads = np.ascontiguousarray(quads, np.int32).reshape(-1, 4)
    verts = np.ascontiguousarray(verts, np.float32)
    n, (m, dim) = len(quads), verts.shape
    qo, vo = np.zeros((4 * n, 4), np.int32), np.zeros((m + 5 * n, dim), np.float32)
    nq, nv = C.c_int(), C.c_int()
    err = C.create_string_buffer(512)
    if host.vpth_catmullclark(quads.ctypes.data, n, verts.ctypes.data, m, dim, int(lock_boundary), -1 if device is None else device,
                              qo.ctypes.data, C.byref(nq), vo.ctypes.data, C.byref(nv), err, len(err)) != 0:
        raise VptError(err.value.decode())
    return qo[:nq.value].copy(), vo[:nv.value].copy()


def vertex_normals(positions: np.ndarray, faces: np.ndarray, device: Optional[int] = None) -> np.ndarray:
    """quads_normals ((n, 4) faces; z == w: a triangle) / triangles_normals ((n, 3)) of yocto_shape.cpp:1478-1512 over (m, 3) float32
    positions: area-weighted face normals added per vertex in face order, normalised.  device: GPU index (vpt_vertex_normals)."""
    positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    faces = np.ascontiguousarray(faces, np.int32)
    out = np.zeros_like(positions)
    err = C.create_string_buffer(512)
    if host.vpth_vertex_normals(positions.ctypes.data, len(positions), faces.ctypes.data, len(faces), faces.shape[1], -1 if device is None else device,
                                out.ctypes.data, err, len(err)) != 0:
        raise VptError(err.value.decode())
    return out


def displace_vertices(texels: np.ndarray, linear: bool, displacement: float, positions: np.ndarray, normals: np.ndarray, texcoords: np.ndarray,
                      device: Optional[int] = None) -> np.ndarray:
    """the displacement step of tesselate_surface (yocto_pathtrace.cpp:1259-1265): positions + normals * displacement * (mean(rgb of
    eval_texture(uv, as_linear)) [- 0.5 for uint8 texels]).  texels: (h, w, 4) uint8 or float32.  device: GPU index (vpt_displace_vertices)."""
    texels = np.ascontiguousarray(texels)
    assert texels.ndim == 3 and texels.shape[2] == 4 and texels.dtype in (np.uint8, np.float32)
    positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    texcoords = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
    out = np.zeros_like(positions)
    err = C.create_string_buffer(512)
    if host.vpth_displace_vertices(texels.ctypes.data, texels.shape[1], texels.shape[0], int(texels.dtype == np.float32), int(linear), C.c_float(displacement),
                                   positions.ctypes.data, normals.ctypes.data, texcoords.ctypes.data, len(positions), -1 if device is None else device,
                                   out.ctypes.data, err, len(err)) != 0:
        raise VptError(err.value.decode())
    return out


class HostScene:
    """load_scene + tesselate_surfaces + make_bvh + make_lights, flattened for the C-ABI."""

    def __init__(self, filename: str, bvh_device: Optional[int] = None, tess_device: Optional[int] = None):
        """bvh_device: build the BVHs on that GPU (make_bvh_device / vpt_build_bvh) instead of on the host - same arrays;
        tess_device: the vertex arithmetic of tesselate_surfaces on that GPU (vpt_subdivide_vertices) - same mesh"""
        err = C.create_string_buffer(1024)
        self.handle = host.vpth_scene_load_ex(os.fsencode(filename), -1 if tess_device is None else tess_device, err, len(err))
        if not self.handle:
            raise VptError(err.value.decode())
        self.filename = filename
        if bvh_device is not None and host.vpth_scene_rebuild_bvh_device(self.handle, bvh_device, err, len(err)) != 0:
            raise VptError(err.value.decode())

    @property
    def desc(self) -> int:
        """address of the vpt_scene_desc (valid while this object lives)"""
        return host.vpth_scene_desc(self.handle)

    def stats(self) -> str:
        buf = C.create_string_buffer(1 << 20)
        n = host.vpth_scene_stats(self.handle, buf, len(buf))
        if n < 0:
            raise VptError("stats buffer too small")
        return buf.value.decode()

    def make_state(self, params: PathtraceParams) -> PathtraceState:
        """make_state, yocto_pathtrace.cpp:960-980"""
        w, h = C.c_int(), C.c_int()
        if host.vpth_state_size(self.handle, params.camera, params.resolution, C.byref(w), C.byref(h)) != 0:
            raise VptError("camera index out of range")
        st = PathtraceState(w.value, h.value, 0, np.zeros((h.value, w.value, 4), np.float32),
                            np.zeros((h.value, w.value), np.int32), np.zeros((h.value, w.value, 2), np.uint64))
        if host.vpth_make_state(self.handle, params.camera, params.resolution, st.image.ctypes.data,
                                st.hits.ctypes.data, st.rngs.ctypes.data) != 0:
            raise VptError("make_state failed")
        return st

    def close(self) -> None:
        if getattr(self, "handle", None) and host is not None:   # at interpreter shutdown the module globals may be gone
            host.vpth_scene_free(self.handle)
        self.handle = None

    def __del__(self):
        self.close()


class DeviceScene:
    """vpt_scene: the scene resident in one GPU's HBM."""

    def __init__(self, scene: HostScene, device: int = 0):
        self.host_scene = scene  # keep the flattened arrays alive until the upload finished
        out = _p()
        _check(hip.vpt_scene_create(scene.desc, device, C.byref(out)), "vpt_scene_create")
        self.handle = out
        self.device = device

    # -- the drop-in for pathtrace_samples(): host state in, host state out ---------------------
    def pathtrace_samples(self, state: PathtraceState, params: PathtraceParams, count: int = 1) -> None:
        abi = params.to_abi()
        samples = C.c_int(state.samples)
        for a in (state.image, state.hits, state.rngs):
            assert a.flags["C_CONTIGUOUS"]
        _check(hip.vpt_render(self.handle, C.byref(abi), count, state.width, state.height, state.image.ctypes.data,
                              state.hits.ctypes.data, state.rngs.ctypes.data, C.byref(samples)), "vpt_render")
        state.samples = samples.value

    # -- device-resident state (pointers are raw device addresses, e.g. torch.Tensor.data_ptr()) ----
    def render_device(self, params: PathtraceParams, layout: VptLayout, nsamples: int, d_image: int, d_hits: int,
                      d_rng: int, stream: int = 0) -> None:
        abi = params.to_abi()
        _check(hip.vpt_render_device(self.handle, C.byref(abi), C.byref(layout), nsamples, d_image, d_hits, d_rng,
                                     stream), "vpt_render_device")

    def intersect(self, rays: np.ndarray, instance: int = -1):
        """intersect_bvh for an (n, 6) float32 array of rays {o, d}: returns (ids (n, 2) int32, uvt (n, 3) float32)"""
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[0]
        ids, uvt = np.zeros((n, 2), np.int32), np.zeros((n, 3), np.float32)
        _check(hip.vpt_intersect(self.handle, n, rays.ctypes.data, instance, ids.ctypes.data, uvt.ctypes.data), "vpt_intersect")
        return ids, uvt

    def kat(self, op: int, records: np.ndarray, iparam: int = 0) -> np.ndarray:
        """vpt_kat (include/vpt_kat.h): run known-answer-test op `op` on an (n, in_stride) float32 array"""
        si, so = C.c_int(), C.c_int()
        _check(hip.vpt_kat_strides(op, C.byref(si), C.byref(so)), "vpt_kat_strides")
        records = np.ascontiguousarray(records, np.float32)
        if records.ndim != 2 or records.shape[1] != si.value:
            raise VptError(f"KAT op {op} takes records of {si.value} floats")
        out = np.zeros((records.shape[0], so.value), np.float32)
        _check(hip.vpt_kat(self.handle, op, iparam, records.shape[0], records.ctypes.data, out.ctypes.data), "vpt_kat")
        return out

    def spheretrace(self, rays: np.ndarray, sdf: int = -1, maxiter: int = 450):
        """vpt_spheretrace for an (n, 6) float32 array of rays {o, d}: (ids (n, 3) int32 {hit, instance, sdf}, t (n,) float32)"""
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[0]
        ids, t = np.zeros((n, 3), np.int32), np.zeros((n,), np.float32)
        _check(hip.vpt_spheretrace(self.handle, n, rays.ctypes.data, sdf, maxiter, ids.ctypes.data, t.ctypes.data), "vpt_spheretrace")
        return ids, t

    def selftest_light_cdf(self, light: int, n: int = 1 << 20):
        """(mismatches, indexed) of the light-CDF search structure against the plain binary search (include/vpt.h)"""
        bad, indexed = C.c_ulonglong(0), C.c_int(0)
        _check(hip.vpt_selftest_light_cdf(self.handle, light, n, C.byref(bad), C.byref(indexed)), "vpt_selftest_light_cdf")
        return bad.value, indexed.value

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(hip.vpt_last_kernel_ms(self.handle, C.byref(ms)), "vpt_last_kernel_ms")
        return ms.value

    def record_bytes(self):
        """(leaf record bytes, attribute record bytes) per primitive: (64, 96), or (48, 64) on a scene of triangles (include/vpt.h)"""
        a, b = C.c_int(0), C.c_int(0)
        _check(hip.vpt_scene_record_bytes(self.handle, C.byref(a), C.byref(b)), "vpt_scene_record_bytes")
        return a.value, b.value

    def last_wave_costs(self) -> np.ndarray:
        """ticks (100 MHz) every wave of the last launch ran, indexed by wave (include/vpt.h)"""
        n = C.c_int(0)
        _check(hip.vpt_last_wave_costs(self.handle, None, 0, C.byref(n)), "vpt_last_wave_costs")
        out = np.zeros(n.value, np.uint32)
        _check(hip.vpt_last_wave_costs(self.handle, out.ctypes.data, n.value, C.byref(n)), "vpt_last_wave_costs")
        return out

    def close(self) -> None:
        if getattr(self, "handle", None) and hip is not None:   # see HostScene.close
            hip.vpt_scene_destroy(self.handle)
        self.handle = None

    def __del__(self):
        self.close()


class MultiDeviceScene:
    """vpt_multi: the scene on several GPUs of this process, tiles dealt round-robin (include/vpt.h)."""

    def __init__(self, scene: HostScene, devices):
        self.host_scene = scene
        devs = (C.c_int * len(devices))(*devices)
        out = _p()
        _check(hip.vpt_multi_create(scene.desc, devs, len(devices), C.byref(out)), "vpt_multi_create")
        self.handle = out

    def pathtrace_samples(self, state: PathtraceState, params: PathtraceParams, count: int = 1) -> None:
        """host state in, host state out (the contract of vpt_render); a device's part of the upload is skipped only while `state`'s
        arrays are the very ones the last call downloaded into and their checksum over every word is unchanged (include/vpt.h)"""
        abi = params.to_abi()
        samples = C.c_int(state.samples)
        _check(hip.vpt_multi_render(self.handle, C.byref(abi), count, state.width, state.height, state.image.ctypes.data,
                                    state.hits.ctypes.data, state.rngs.ctypes.data, C.byref(samples)), "vpt_multi_render")
        state.samples = samples.value

    # -- resident state: upload once, render in batches without transfers, download on demand ---------
    def set_state(self, state: PathtraceState) -> None:
        _check(hip.vpt_multi_set_state(self.handle, state.width, state.height, state.image.ctypes.data, state.hits.ctypes.data,
                                       state.rngs.ctypes.data, state.samples), "vpt_multi_set_state")
        self._resident = (state.width, state.height, state.samples)

    def render_resident(self, params: PathtraceParams, count: int = 1) -> int:
        """`count` passes on the state the devices hold; returns the sample count reached"""
        abi = params.to_abi()
        w, h, n = self._resident
        samples = C.c_int(n)
        _check(hip.vpt_multi_render(self.handle, C.byref(abi), count, w, h, None, None, None, C.byref(samples)), "vpt_multi_render")
        self._resident = (w, h, samples.value)
        return samples.value

    def get_state(self, state: PathtraceState) -> None:
        samples = C.c_int(0)
        _check(hip.vpt_multi_get_state(self.handle, state.image.ctypes.data, state.hits.ctypes.data, state.rngs.ctypes.data,
                                       C.byref(samples)), "vpt_multi_get_state")
        state.samples = samples.value

    def transport(self) -> str:
        return hip.vpt_multi_transport(self.handle).decode()

    def uploaded_parts(self) -> int:
        """devices whose part of the caller's arrays the last pathtrace_samples call uploaded (include/vpt.h: the residency rule)"""
        return hip.vpt_multi_uploaded_parts(self.handle)

    def get_render(self, width: int, height: int) -> np.ndarray:
        out = np.zeros((height, width, 4), np.float32)
        _check(hip.vpt_multi_get_render(self.handle, out.ctypes.data), "vpt_multi_get_render")
        return out

    def close(self) -> None:
        if getattr(self, "handle", None) and hip is not None:
            hip.vpt_multi_destroy(self.handle)
        self.handle = None

    def __del__(self):
        self.close()


def layout_slots(layout: VptLayout) -> int:
    n = hip.vpt_layout_slots(C.byref(layout))
    if n < 0:
        raise VptError(hip.vpt_last_error().decode())
    return n


def state_upload(layout: VptLayout, state: PathtraceState, d_image: int, d_hits: int, d_rng: int, stream: int = 0):
    _check(hip.vpt_state_upload(C.byref(layout), state.image.ctypes.data, state.hits.ctypes.data,
                                state.rngs.ctypes.data, d_image, d_hits, d_rng, stream), "vpt_state_upload")


def state_download(layout: VptLayout, d_image: int, d_hits: int, d_rng: int, state: PathtraceState, stream: int = 0):
    _check(hip.vpt_state_download(C.byref(layout), d_image, d_hits, d_rng, state.image.ctypes.data,
                                  state.hits.ctypes.data, state.rngs.ctypes.data, stream), "vpt_state_download")


def resolve_device(layout: VptLayout, d_tiles_all: int, samples: int, d_rows: int, stream: int = 0):
    _check(hip.vpt_resolve_device(C.byref(layout), d_tiles_all, samples, d_rows, stream), "vpt_resolve_device")


def resolve_srgb8_device(layout: VptLayout, d_tiles_all: int, samples: int, d_rgba8: int, stream: int = 0):
    """get_render + rgb_to_srgb + float_to_byte on the device (row-major RGBA8)"""
    _check(hip.vpt_resolve_srgb8_device(C.byref(layout), d_tiles_all, samples, d_rgba8, stream), "vpt_resolve_srgb8_device")


def get_render(state: PathtraceState) -> np.ndarray:
    """get_render, yocto_pathtrace.cpp:1105-1116: image * (1/samples) in float32"""
    return state.image * np.float32(np.float32(1.0) / np.float32(state.samples))


def selftest_reciprocal(device: int = 0):
    """(mismatches, fallbacks) of the kernels' exact-reciprocal shortcut over all 2^32 floats (include/vpt.h)"""
    bad, skipped = C.c_ulonglong(0), C.c_ulonglong(0)
    _check(hip.vpt_selftest_reciprocal(device, C.byref(bad), C.byref(skipped)), "vpt_selftest_reciprocal")
    return bad.value, skipped.value


def linear_to_srgb8(image_sum: np.ndarray, samples: int) -> np.ndarray:
    """save_image's quantisation: rgb_to_srgb then float_to_byte (yocto_color.h:207-231)"""
    h, w, _ = image_sum.shape
    out = np.zeros((h, w, 4), np.uint8)
    src = np.ascontiguousarray(image_sum, np.float32)
    host.vpth_linear_to_srgb8(w, h, src.ctypes.data, samples, out.ctypes.data)
    return out


def encode_jpeg_q75(rgba8: np.ndarray) -> bytes:
    """byte-exact stand-in for the reference's stbi_write_jpg(..., quality 75) (stb_image_write.h:1398-1611)"""
    h, w, _ = rgba8.shape
    src = np.ascontiguousarray(rgba8, np.uint8)
    n = host.vpth_encode_jpeg_q75(w, h, src.ctypes.data, None, 0)
    buf = (C.c_uint8 * n)()
    host.vpth_encode_jpeg_q75(w, h, src.ctypes.data, buf, n)
    return bytes(buf)


def layout_pixel_index(layout: VptLayout) -> np.ndarray:
    """Host mirror of the device's slot -> pixel map (slot_to_pixel, csrc/vpt_kernels.hip.h): for every
    state slot of `layout.rank`, the row-major pixel index j*width+i it holds, or -1 for padding."""
    tw, th, w, h = layout.tile_w, layout.tile_h, layout.width, layout.height
    if tw % 8 or th % 8 or tw < 8 or th < 8:
        raise VptError("tile size must be a multiple of 8x8")
    tiles_x, tiles_y = -(-w // tw), -(-h // th)
    per_tile = tw * th
    local_tiles = -(-(tiles_x * tiles_y) // layout.nranks)
    slot = np.arange(local_tiles * per_tile, dtype=np.int64)
    local_tile, p = slot // per_tile, slot % per_tile
    tile = local_tile * layout.nranks + layout.rank
    ty, tx = tile // tiles_x, tile % tiles_x
    bw = tw // 8
    blk, q = p // 64, p % 64
    by, bx = blk // bw, blk % bw
    px = tx * tw + bx * 8 + (q % 8)
    py = ty * th + by * 8 + (q // 8)
    ok = (tile < tiles_x * tiles_y) & (px < w) & (py < h)
    return np.where(ok, py * w + px, -1)
