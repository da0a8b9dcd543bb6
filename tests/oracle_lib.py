"""Test-side loader for the CHECKERS under oracle/ (never imported by the product):
  * libvpt_oracle.so  — our CPU restatement (oracle/vpt_oracle.cpp), same contract as vpt_render
  * _ref/ref_driver   — the reference's own renderer built by oracle/Makefile (only where it exists)
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libvpt_oracle.so")
REF_DRIVER = os.path.join(ORACLE_DIR, "_ref", "ref_driver")
COUNTER_NAMES = ("samples,scene_nodes,shape_nodes,instance_tests,quad_tests,tri_tests,texel_f32,texel_u8,"
                 "cdf_probes,surface_hits,volume_events,bounces,sdf_evals,voxel_fetches,light_pdf_hops,marches,steps_hit,"
                 "steps_maxiter,steps_escaped,steps_far,light_march_steps").split(",")

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(ORACLE_SO)
        _lib.vpt_oracle_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_void_p]
    return _lib


def oracle_render(host_scene, params, state, nsamples, nthreads=0, counters=False, flags=None, pixels=None, perturb=None):
    """Advance `state` by `nsamples` passes with the CPU oracle. Returns a counter dict if asked.
    flags: (h, w) uint8 array that receives each pixel's condition flags (bit 0: an SDF-light pdf was evaluated on a
    hit; oracle/vpt_oracle.cpp).  pixels: int32 array of row-major pixel indices: render only those.
    perturb: (seed, site_mask): nudge every libm result of the selected classes by -1 / 0 / +1 ulp (oracle/vpt_oracle.cpp);
    may be combined with `pixels`."""
    abi = params.to_abi()
    samples = C.c_int(state.samples)
    cnt = np.zeros(24, np.uint64)
    common = (host_scene.desc, C.addressof(abi), nsamples, state.width, state.height, state.image.ctypes.data,
              state.hits.ctypes.data, state.rngs.ctypes.data, C.byref(samples), nthreads)
    if flags is not None or pixels is not None or perturb is not None:
        assert not counters and (flags is None or (pixels is None and perturb is None))
        if perturb is not None:
            fn = lib().vpt_oracle_render_perturbed
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.POINTER(C.c_int), C.c_int, C.c_uint64, C.c_uint, C.c_void_p, C.c_int]
            if pixels is not None:
                pixels = np.ascontiguousarray(pixels, np.int32)
            rc = fn(*common, int(perturb[0]), int(perturb[1]), pixels.ctypes.data if pixels is not None else None,
                    len(pixels) if pixels is not None else 0)
        elif flags is not None:
            assert flags.dtype == np.uint8 and flags.shape == (state.height, state.width) and flags.flags["C_CONTIGUOUS"]
            fn = lib().vpt_oracle_render_flags
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.POINTER(C.c_int), C.c_int, C.c_void_p]
            rc = fn(*common, flags.ctypes.data)
        else:
            pixels = np.ascontiguousarray(pixels, np.int32)
            fn = lib().vpt_oracle_render_pixels
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_int]
            rc = fn(*common, pixels.ctypes.data, len(pixels))
        if rc != 0:
            raise RuntimeError(f"oracle failed: {rc}")
        state.samples = samples.value
        return None
    rc = lib().vpt_oracle_render(host_scene.desc, C.addressof(abi), nsamples, state.width, state.height,
                                 state.image.ctypes.data, state.hits.ctypes.data, state.rngs.ctypes.data,
                                 C.byref(samples), nthreads, cnt.ctypes.data if counters else None)
    if rc != 0:
        raise RuntimeError(f"oracle failed: {rc}")
    state.samples = samples.value
    return dict(zip(COUNTER_NAMES, (int(x) for x in cnt))) if counters else None


def unstable_pixels(host_scene, params, nsamples, ref_image, ref_rngs, make_state, rounds=16, rtol=1e-3, atol_per_sample=1e-3,
                    pixels=None, first_seed=1):
    """Pixels on which a faithful implementation may differ from the reference, MEASURED on the reference's own
    arithmetic: re-render with every libm result nudged by -1 / 0 / +1 float ulp (`rounds` different pseudo-random
    patterns, oracle/vpt_oracle.cpp) and mark the pixels whose RNG end state changes (a discrete decision flipped)
    or whose radiance sum moves by more than rtol / atol (half of what the parity tests allow the device).
    pixels: restrict the re-renders to these row-major indices (the others come back False).
    Returns (stream_unstable, radiance_unstable) boolean (h, w) arrays."""
    stream = np.zeros(ref_rngs.shape[:2], bool)
    radiance = np.zeros(ref_rngs.shape[:2], bool)
    chosen = np.ones(ref_rngs.shape[:2], bool)
    if pixels is not None:
        chosen[:] = False
        chosen.reshape(-1)[np.asarray(pixels)] = True
    for r in range(first_seed, first_seed + rounds):
        q = make_state()
        oracle_render(host_scene, params, q, nsamples, perturb=(r, 15), pixels=pixels)
        same = np.all(q.rngs == ref_rngs, axis=-1)
        close = np.all(np.isclose(q.image, ref_image, rtol=rtol, atol=atol_per_sample * nsamples), axis=-1)
        stream |= chosen & ~same
        radiance |= chosen & same & ~close
    return stream, radiance


def oracle_intersect(host_scene, rays, instance=-1):
    """The oracle's intersect_bvh for an (n, 6) float32 array of rays: (ids (n, 2) int32, uvt (n, 3) float32)."""
    rays = np.ascontiguousarray(rays, np.float32)
    n = rays.shape[0]
    ids, uvt = np.zeros((n, 2), np.int32), np.zeros((n, 3), np.float32)
    lib().vpt_oracle_intersect.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rc = lib().vpt_oracle_intersect(host_scene.desc, n, rays.ctypes.data, instance, ids.ctypes.data, uvt.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle intersect failed: {rc}")
    return ids, uvt


def have_reference():
    return os.path.exists(REF_DRIVER)


def reference_render(scene_json, shader, resolution, samples, bounces=4, stmaxiter=450, camera=0, noparallel=False,
                     noimplicitmis=False, workdir="/tmp", stats=False, output=None):
    """Run the reference's own renderer; returns (width, height, image, hits, rngs, info[, stats])."""
    state_file = os.path.join(workdir, f"ref_state_{os.getpid()}.bin")
    cmd = [REF_DRIVER, "--scene", scene_json, "--shader", shader, "--resolution", str(resolution), "--samples",
           str(samples), "--bounces", str(bounces), "--stmaxiter", str(stmaxiter), "--camera", str(camera),
           "--state", state_file]
    stats_file = os.path.join(workdir, f"ref_stats_{os.getpid()}.json")
    if stats:
        cmd += ["--stats", stats_file]
    if output:
        cmd += ["--output", output]
    if noparallel:
        cmd.append("--noparallel")
    if noimplicitmis:
        cmd.append("--noimplicitmis")
    out = subprocess.check_output(cmd)
    info = json.loads(out.decode().strip().splitlines()[-1])
    raw = open(state_file, "rb").read()
    os.remove(state_file)
    hdr = np.frombuffer(raw, np.int32, 4)
    assert hdr[0] == 0x53545056
    w, h, n = int(hdr[1]), int(hdr[2]), int(hdr[3])
    off = 16
    image = np.frombuffer(raw, np.float32, w * h * 4, off).reshape(h, w, 4).copy()
    off += w * h * 16
    hits = np.frombuffer(raw, np.int32, w * h, off).reshape(h, w).copy()
    off += w * h * 4
    rngs = np.frombuffer(raw, np.uint64, w * h * 2, off).reshape(h, w, 2).copy()
    res = [w, h, image, hits, rngs, info]
    if stats:
        res.append(json.load(open(stats_file)))
        os.remove(stats_file)
    return res
