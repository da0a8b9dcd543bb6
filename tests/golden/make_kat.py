#!/usr/bin/env python3
"""Regenerates tests/golden/kat_tables.npz: for every case of tests/kat_lib.py CASES, a batch of input records
(seeded numpy) and the outputs of the REFERENCE'S OWN function of that name (oracle/_ref/ref_tables, built from
/root/reference by oracle/Makefile).  Run in the build container only:

    python tests/golden/make_kat.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kat_lib as K  # noqa: E402
import vpt_loader  # noqa: E402

# where the scenes' content lies (world space): ray origins, shading points
BOUNDS = {
    "03_volume": ((-0.7, -0.05, -0.45), (0.7, 0.45, 0.45)), "03_volume_lobes": ((-0.8, -0.05, -0.45), (0.8, 0.45, 0.45)),
    "05_head1ss_sub": ((-0.25, -0.1, -0.25), (0.25, 0.4, 0.25)), "01_surface_min": ((-0.7, -0.05, -0.45), (0.7, 0.45, 0.45)),
    "06_gridsdf_synth": ((-0.6, -0.05, -0.6), (0.6, 0.5, 0.6)), "07_sdfunction_synth": ((-0.6, -0.05, -0.6), (0.6, 0.5, 0.6)),
}


def main():
    assert K.have_reference(), "build oracle/_ref first: make -C oracle ref"
    vpt = vpt_loader.load()   # host library only: scene counts for the generators (nothing is rendered here)
    infos = {key: K.scene_info(vpt.HostScene(K.scene_path(key))) for key in K.SCENE_FILES}
    out = {}
    for idx, (name, key, op, iparam) in enumerate(K.CASES):
        rng = np.random.default_rng(1000 + idx)
        info = infos.get(key)
        lo, hi = BOUNDS.get(key, ((0, 0, 0), (1, 1, 1)))
        if op == "lobes":
            rec = K.gen_lobes(rng)
        elif op == "media":
            rec = K.gen_media(rng)
        elif op == "texture":
            rec = K.gen_texture(rng, info)
        elif op == "camera":
            rec = K.gen_camera(rng, info)
        elif op == "intersect":
            rec = K.gen_intersect(rng, info, lo, hi)
        elif op == "surface":
            rec = K.gen_surface(rng, info)
        elif op == "environment":
            rec = K.gen_dirs(rng)
        elif op == "sample_lights":
            rec = K.gen_sample_lights(rng, lo, hi)
        elif op == "lights_pdf":
            # half of the directions aim at the lights (the reference's own sample_lights from the same points)
            n = 3072
            sl = K.gen_sample_lights(rng, lo, hi, n)
            aimed = K.run_reference(key, "sample_lights", 0, sl)
            rec = np.zeros((n, 6), np.float32)
            rec[:, 0:3] = sl[:, 0:3]
            rec[:, 3:6] = K._unit(rng, n)
            rec[::2, 3:6] = aimed[::2]
        elif op == "sdf_scene":
            n = 4096
            rec = np.zeros((n, 4), np.float32)
            rec[:, 0:3] = K.gen_points(rng, lo, hi, n)
            # a third of the points inside the voxel grids' boxes, so that the trilinear branch runs
            rec[::3, 0:3] = K.gen_points(rng, (-0.45, 0.0, -0.05), (0.4, 0.16, 0.2), len(rec[::3]))
            rec[:, 3] = rng.uniform(1e-4, 3.0, size=n)
        elif op == "spheretrace":
            n = 2048
            cam = K.run_reference(key, "camera", 0, K.gen_camera(rng, info, n))   # the scene's own camera rays
            rec = np.zeros((n, 7), np.float32)
            rec[:, 0:6] = cam
            rec[n // 2:, 0:3] = K.gen_points(rng, lo, hi, n - n // 2)
            rec[n // 2:, 3:6] = K._unit(rng, n - n // 2)
            rec[:, 6] = -1
            rec[-n // 4:, 6] = rng.integers(0, info["sdfs"], size=n // 4)             # single-SDF form (the SDF-light pdf)
        elif op == "sdf_normal":
            # points the shaders evaluate normals at: spheretrace hits of camera and random rays, plus free points
            n = 2048
            cam = K.run_reference(key, "camera", 0, K.gen_camera(rng, info, n))
            rays = np.zeros((n, 7), np.float32)
            rays[:, 0:6] = cam
            rays[:, 6] = -1
            hits = K.run_reference(key, "spheretrace", 450, rays)
            rec = np.zeros((n, 6), np.float32)
            for i in range(n):
                if hits[i, 0] != 0:
                    grid = hits[i, 2] >= 0
                    rec[i, 0], rec[i, 1] = (0, hits[i, 2]) if grid else (1, hits[i, 3])
                    rec[i, 2:5] = rays[i, 0:3] + rays[i, 3:6] * hits[i, 1]
                    rec[i, 5] = hits[i, 1]
                else:
                    kind = int(rng.integers(0, 2))
                    rec[i, 0], rec[i, 1] = kind, rng.integers(0, info["vol_instances"] if kind == 0 else info["sdfs"])
                    rec[i, 2:5] = K.gen_points(rng, lo, hi, 1)[0]
                    rec[i, 5] = rng.uniform(1e-4, 3.0)
        elif op == "volume":
            n = 4096
            rec = np.zeros((n, 4), np.float32)
            rec[:, 0] = rng.integers(0, info["volumes"], size=n)
            rec[:, 1:4] = rng.uniform(-1.2, 1.2, size=(n, 3))
            rec[:64, 1:4] = rng.choice(np.float32([-1, 0, 1]), size=(64, 3))       # cell corners, box faces
        elif op == "sdf_function":
            n = 4096
            rec = np.zeros((n, 4), np.float32)
            rec[:, 0] = rng.integers(0, info["sdfs"], size=n)
            rec[:, 1:4] = rng.uniform(-0.3, 0.3, size=(n, 3))
        else:
            raise SystemExit(f"no generator for {op}")
        rec = np.ascontiguousarray(rec, np.float32)
        res = K.run_reference(key, op, iparam, rec)
        out[name + "_in"], out[name + "_out"] = rec, res
        print(f"{name:22s} {op:14s} n={rec.shape[0]:5d}  finite outputs {np.isfinite(res).mean():.3f}  non-zero {np.mean(res != 0):.3f}")
    np.savez_compressed(K.TABLES, **out)
    print("wrote", K.TABLES, os.path.getsize(K.TABLES) // 1024, "KiB")


if __name__ == "__main__":
    main()
