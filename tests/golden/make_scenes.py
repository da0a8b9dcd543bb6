#!/usr/bin/env python3
"""Builds the SUBSTITUTE scenes for the BASELINE configs whose assets are missing from the reference
(.MISSING_LARGE_BLOBS).  Everything is derived from the reference's own scene JSONs
and data files; only assets that exist are referenced, and the voxel SDFs are generated from analytic
formulas (deterministic, no RNG).  Run from anywhere:  python tests/golden/make_scenes.py

  01_surface_min   tests/01_surface/surface.json with every OBJ/subdiv shape replaced by shapes/sphere.ply
                   (same instances, frames, materials, textures, lights)         -> config 1
  05_head1ss_sub   tests/05_head1ss/head1ss.json with shape1.ply -> 03_volume/shapes/bunny.ply (144k
                   triangles, uniformly scaled frame) and the missing scattering texture dropped -> config 3
  06_gridsdf_synth tests/06_gridsdf/gridsdf.json with sdfs/sackboy.sdf / bunny.sdf generated here
                   (binary 48^3 sphere-union in mm, text 40^3 torus)              -> config 4
"""
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tests"
OUT = os.path.join(HERE, "scenes")


def dump(path, scene):
    with open(path, "w") as f:
        json.dump(scene, f, indent=1)


def surface_min():
    s = json.load(open(os.path.join(REF, "01_surface", "surface.json")))
    s.pop("subdivs")
    for shp in s["shapes"]:
        if shp["uri"].endswith(".obj"):
            shp["uri"] = "../03_volume/shapes/sphere.ply"
        else:
            shp["uri"] = "../03_volume/" + shp["uri"]
    tex = {"floor": "../03_volume/textures/floor.png", "sky": "../03_volume/textures/sky.hdr",
           "uvgrid": "../shared_textures/uvgrid.png", "spot": "../shared_textures/spot.png",
           "bumps-normal": "../shared_textures/bumps-normal.png",
           "bumps-displacement": "../shared_textures/bumps-normal.png"}  # unused by any material after the subdiv drop
    for t in s["textures"]:
        t["uri"] = tex[t["name"]]
    dump(os.path.join(OUT, "01_surface_min", "surface_min.json"), s)


def head1ss_sub():
    s = json.load(open(os.path.join(REF, "05_head1ss", "head1ss.json")))
    s["shapes"][0]["uri"] = "../03_volume/shapes/bunny.ply"
    s["textures"] = [{"name": "texture2", "uri": "../shared_textures/texture2.hdr"}]
    for e in s["environments"]:
        e["emission_tex"] = 0
    s["materials"][0].pop("scattering_tex")
    # bunny bbox is (-0.081,0,-0.062)..(0.081,0.162,0.062); camera1 looks at (0.05,0.30,-0.04) with a 0.38 m field
    s["instances"][0]["frame"] = [1.8, 0, 0, 0, 1.8, 0, 0, 0, 1.8, 0.05, 0.15, -0.04]
    dump(os.path.join(OUT, "05_head1ss_sub", "head1ss_sub.json"), s)


def gridsdf_synth():
    s = json.load(open(os.path.join(REF, "06_gridsdf", "gridsdf.json")))
    s["textures"][0]["uri"] = "../03_volume/textures/sky.hdr"
    s["volumes"] = [{"name": "sackboy", "uri": "sdfs/sackboy_synth.sdf", "binary": True},
                    {"name": "bunny", "uri": "sdfs/bunny_synth.sdf", "binary": False}]
    dump(os.path.join(OUT, "06_gridsdf_synth", "gridsdf_synth.json"), s)
    # grid A: 48^3, cell 3.0 "mm" (instances use scale 0.001 => 0.144 world units); union of two spheres
    n, res = 48, 3.0
    size = n * res
    ax = np.arange(n, dtype=np.float64) / (n - 1) * size
    x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")            # value index = x + y*W + z*W*H
    body = np.sqrt((x - 72) ** 2 + (y - 45) ** 2 + (z - 72) ** 2) - 38
    head = np.sqrt((x - 72) ** 2 + (y - 100) ** 2 + (z - 72) ** 2) - 28
    vol = np.minimum(body, head).astype(np.float32)
    with open(os.path.join(OUT, "06_gridsdf_synth", "sdfs", "sackboy_synth.sdf"), "wb") as f:
        f.write(struct.pack("<iiif", n, n, n, res))
        f.write(np.eye(4, dtype=np.float32).tobytes())          # 4x4 matrix, ignored by the reader
        f.write(np.transpose(vol, (2, 1, 0)).tobytes())         # x fastest
    # grid B: 40^3 text (SDFGen layout: "W H D" / origin / cell size / values), torus, world units
    n, res = 40, 0.0036
    size = n * res
    ax = np.arange(n, dtype=np.float64) / (n - 1) * size
    x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")
    c = size / 2
    q = np.sqrt((x - c) ** 2 + (z - c) ** 2) - 0.045
    vol = (np.sqrt(q ** 2 + (y - 0.03) ** 2) - 0.02).astype(np.float32)
    flat = np.transpose(vol, (2, 1, 0)).ravel()
    with open(os.path.join(OUT, "06_gridsdf_synth", "sdfs", "bunny_synth.sdf"), "w") as f:
        f.write(f"{n} {n} {n}\n0 0 0\n{res!r}\n")
        for i in range(0, len(flat), 8):
            f.write(" ".join(repr(float(v)) for v in flat[i:i + 8]) + "\n")


if __name__ == "__main__":
    surface_min()
    head1ss_sub()
    gridsdf_synth()
    print("scenes written under", OUT)
