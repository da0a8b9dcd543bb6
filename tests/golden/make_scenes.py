#!/usr/bin/env python3
"""Builds the SUBSTITUTE scenes for the BASELINE configs whose assets are missing from the reference
(.MISSING_LARGE_BLOBS).  Everything is derived from the reference's own scene JSONs
and data files; only assets that exist are referenced, and the voxel SDFs are generated from analytic
formulas (deterministic, no RNG).  Run from anywhere:  python tests/golden/make_scenes.py

  01_surface_min   tests/01_surface/surface.json with its three loadable subdivs (the reference's own OBJ cages) and the
                   stripped sphere-displaced.obj replaced by shapes/sphere.ply (shape and displaced cage)  -> config 1
  08_subdiv_synth  hand-made cages for the corner cases of tesselate_surfaces (non-manifold boundary, triangles, no
                   texcoords, smooth = false, float / 8-bit displacement)          -> SURVEY 8(f) row 4
  05_head1ss_sub   tests/05_head1ss/head1ss.json with shape1.ply -> 03_volume/shapes/bunny.ply (144k
                   triangles, uniformly scaled frame) and the missing scattering texture dropped -> config 3
  06_gridsdf_synth tests/06_gridsdf/gridsdf.json with sdfs/sackboy.sdf / bunny.sdf generated here
                   (binary 48^3 sphere-union in mm, text 40^3 torus)              -> config 4
  06_gridsdf_full  the same with the grids at the sizes SURVEY 8(d) names (96^3 + 64^3, both binary: 4.6 MB)  -> config 4's bench workload
  07_sdfunction_synth  tests/07_sdfunction/sdfunction.json (reflective x3, capped cone, torus, box light) with the
                   same generated grids, plus sphere / bbox / plane SDFs and transparent (rough, delta, opacity < 1),
                   delta reflective, refractive and gltfpbr materials: every sd_* primitive and every lobe the
                   implicit shaders can reach                                      -> SURVEY 8(a) rows F18, S6
  03_volume_lobes  tests/03_volume/volume.json with the five spheres' materials replaced by the lobes no reference
                   scene with loadable assets uses on a mesh (delta + rough reflective, rough + delta transparent
                   with opacity < 1, gltfpbr, subsurface) and an emissive sphere (a mesh light with a real BVH:
                   the 100-hop pdf walk of yocto_pathtrace.cpp:363-378)            -> SURVEY 8(a) rows F17-F21
"""
import json
import os
import shutil
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tests"
OUT = os.path.join(HERE, "scenes")


def dump(path, scene):
    with open(path, "w") as f:
        json.dump(scene, f, indent=1)


def surface_min():
    """tests/01_surface as shipped, minus the one asset the checkout lacks: the cages of its subdivs (cube-subdiv 4 levels,
    spot-subdiv-scaled 2, suzanne-subdiv 2: the reference's own OBJ files, copied as data) are tesselated by
    tesselate_surfaces; shape 2 / subdiv 1 (sphere-displaced.obj: stripped, .MISSING_LARGE_BLOBS) take shapes/sphere.ply as
    shape and as cage, keeping the displacement by bumps-displacement.png."""
    s = json.load(open(os.path.join(REF, "01_surface", "surface.json")))
    out = os.path.join(OUT, "01_surface_min")
    for sub in ("shapes", "subdivs"):
        os.makedirs(os.path.join(out, sub), exist_ok=True)
    for shp in s["shapes"]:
        if shp["uri"].endswith("sphere-displaced.obj"):
            shp["uri"] = "../03_volume/shapes/sphere.ply"
        elif shp["uri"].endswith(".obj"):
            shutil.copyfile(os.path.join(REF, "01_surface", shp["uri"]), os.path.join(out, shp["uri"]))
        else:
            shp["uri"] = "../03_volume/" + shp["uri"]
    for sub in s["subdivs"]:
        if sub["uri"].endswith("sphere-displaced.obj"):
            sub["uri"] = "../03_volume/shapes/sphere.ply"
        else:
            shutil.copyfile(os.path.join(REF, "01_surface", sub["uri"]), os.path.join(out, sub["uri"]))
    shutil.copyfile(os.path.join(REF, "01_surface", "textures", "bumps-displacement.png"), os.path.join(OUT, "shared_textures", "bumps-displacement.png"))
    tex = {"floor": "../03_volume/textures/floor.png", "sky": "../03_volume/textures/sky.hdr",
           "uvgrid": "../shared_textures/uvgrid.png", "spot": "../shared_textures/spot.png",
           "bumps-normal": "../shared_textures/bumps-normal.png",
           "bumps-displacement": "../shared_textures/bumps-displacement.png"}
    for t in s["textures"]:
        t["uri"] = tex[t["name"]]
    dump(os.path.join(out, "surface_min.json"), s)


def subdiv_synth():
    """08_subdiv_synth: cages that reach the corners of tesselate_catmullclark / tesselate_surface no reference scene does -
    a non-manifold "bow tie" (two open quads sharing one vertex: that vertex collects FOUR crease contributions, whose float sum
    depends on the order in which the reference's unordered_map lists the boundary edges), an open strip with a triangle
    (z == w faces: /3 face points, three refined quads), a cage without texture coordinates, smooth = false (normals dropped),
    a displacement read from a float texture (no -0.5) and one from an 8-bit texture on a subdivided cage."""
    out = os.path.join(OUT, "08_subdiv_synth")
    os.makedirs(os.path.join(out, "subdivs"), exist_ok=True)
    with open(os.path.join(out, "subdivs", "bowtie.obj"), "w") as f:   # two quads sharing vertex 3, plus a triangle hanging off the second
        f.write("v -0.1 0.02 -0.1\nv 0 0.02 -0.1\nv 0 0.06 0\nv -0.1 0.02 0\nv 0.1 0.02 0\nv 0.1 0.02 0.1\nv 0 0.02 0.1\nv 0.17 0.08 0.05\n"
                "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt 0.5 0.5\n"
                "f 1/1 2/2 3/3 4/4\nf 3/1 5/2 6/3 7/4\nf 5/1 8/5 6/3\n")
    with open(os.path.join(out, "subdivs", "tetra.obj"), "w") as f:    # closed, triangles only, no texcoords / normals
        f.write("v 0 0 0\nv 0.12 0 0\nv 0.06 0 0.1\nv 0.06 0.11 0.04\nf 1 3 2\nf 1 2 4\nf 2 3 4\nf 3 1 4\n")
    with open(os.path.join(out, "subdivs", "strip.obj"), "w") as f:    # an open 3 x 1 strip with shared texcoords, then a pentagon (fanned)
        f.write("v 0 0.01 0\nv 0.05 0.03 0\nv 0.1 0.01 0\nv 0.15 0.03 0\nv 0 0.01 0.06\nv 0.05 0.04 0.06\nv 0.1 0.01 0.06\nv 0.15 0.04 0.06\nv 0.2 0.02 0.03\n"
                "vt 0 0\nvt 0.33 0\nvt 0.66 0\nvt 1 0\nvt 0 1\nvt 0.33 1\nvt 0.66 1\nvt 1 1\nvt 1.2 0.5\n"
                "f 1/1 5/5 6/6 2/2\nf 2/2 6/6 7/7 3/3\nf 3/3 7/7 8/8 4/4\nf 4/4 8/8 9/9\n")
    ident = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    s = {"asset": {"version": "4.2"},
         "cameras": [{"name": "default", "lens": 0.05, "aperture": 0.0, "aspect": 2.0,
                      "frame": [0.8151804208755493, -0.0, 0.579207181930542, 0.16660168766975403, 0.9577393531799316, -0.23447643220424652,
                                -0.5547295212745667, 0.28763750195503235, 0.7807304263114929, -0.75, 0.4, 0.9]}],
         "environments": [{"name": "sky", "emission": [0.5, 0.5, 0.5], "emission_tex": 0}],
         "textures": [{"name": "sky", "uri": "../03_volume/textures/sky.hdr"}, {"name": "uvgrid", "uri": "../shared_textures/uvgrid.png"},
                      {"name": "bumps", "uri": "../shared_textures/bumps-displacement.png"}, {"name": "hdr", "uri": "../shared_textures/texture2.hdr"}],
         "materials": [{"name": "floor", "color": [0.7, 0.7, 0.7], "type": "matte"},
                       {"name": "a", "color": [1, 1, 1], "color_tex": 1, "roughness": 0.2, "type": "glossy"},
                       {"name": "b", "color": [0.7, 0.5, 0.5], "roughness": 0.2, "type": "glossy"},
                       {"name": "c", "color": [0.5, 0.7, 0.5], "type": "matte"},
                       {"name": "d", "color": [0.5, 0.5, 0.7], "roughness": 0.3, "type": "glossy"},
                       {"name": "light", "emission": [20, 20, 20], "type": "matte"}],
         "shapes": [{"name": "floor", "uri": "../03_volume/shapes/floor.ply"}, {"name": "bowtie", "uri": "subdivs/bowtie.obj"},
                    {"name": "tetra", "uri": "subdivs/tetra.obj"}, {"name": "strip", "uri": "subdivs/strip.obj"},
                    {"name": "ball", "uri": "../03_volume/shapes/sphere.ply"}, {"name": "light", "uri": "../03_volume/shapes/arealight1.ply"}],
         "subdivs": [{"name": "bowtie", "shape": 1, "subdivisions": 3, "smooth": True, "uri": "subdivs/bowtie.obj"},
                     {"name": "tetra", "shape": 2, "subdivisions": 2, "smooth": False, "uri": "subdivs/tetra.obj"},
                     {"name": "strip", "shape": 3, "subdivisions": 2, "smooth": True, "displacement": 0.02, "displacement_tex": 2,
                      "uri": "subdivs/strip.obj"},
                     {"name": "ball", "shape": 4, "subdivisions": 0, "smooth": False, "displacement": 0.004, "displacement_tex": 3,
                      "uri": "../03_volume/shapes/sphere.ply"}],
         "instances": [{"name": "floor", "shape": 0, "material": 0},
                       {"name": "bowtie", "frame": ident + [-0.45, 0, 0], "shape": 1, "material": 1},
                       {"name": "tetra", "frame": ident + [-0.2, 0, 0], "shape": 2, "material": 2},
                       {"name": "strip", "frame": ident + [0.05, 0, 0], "shape": 3, "material": 3},
                       {"name": "ball", "frame": ident + [0.4, 0, 0], "shape": 4, "material": 4},
                       {"name": "light", "frame": [0.8944271802902222, -0.0, 0.4472135901451111, 0.27562475204467773, 0.7874992489814758,
                                                   -0.5512495040893555, -0.3521803617477417, 0.6163156628608704, 0.7043607234954834, -0.4, 0.8, 0.8],
                        "shape": 5, "material": 5}]}
    dump(os.path.join(out, "subdiv_synth.json"), s)


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


def gridsdf_full():
    """06_gridsdf_full: the same scene with the grids at the sizes SURVEY 8(d) names for config 4 - 96^3 (sphere union, mm,
    instances scale 0.001) and 64^3 (torus, world units), both binary: 4.6 MB of voxels, the bench workload of config 4."""
    s = json.load(open(os.path.join(REF, "06_gridsdf", "gridsdf.json")))
    s["textures"][0]["uri"] = "../03_volume/textures/sky.hdr"
    s["volumes"] = [{"name": "sackboy", "uri": "sdfs/sackboy_96.sdf", "binary": True},
                    {"name": "bunny", "uri": "sdfs/bunny_64.sdf", "binary": True}]
    out = os.path.join(OUT, "06_gridsdf_full")
    os.makedirs(os.path.join(out, "sdfs"), exist_ok=True)
    dump(os.path.join(out, "gridsdf_full.json"), s)

    def write(name, n, res, field):
        size = n * res
        ax = np.arange(n, dtype=np.float64) / (n - 1) * size
        x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")        # value index = x + y*W + z*W*H
        vol = field(x, y, z, size).astype(np.float32)
        with open(os.path.join(out, "sdfs", name), "wb") as f:
            f.write(struct.pack("<iiif", n, n, n, res))
            f.write(np.eye(4, dtype=np.float32).tobytes())      # 4x4 matrix, ignored by the reader
            f.write(np.transpose(vol, (2, 1, 0)).tobytes())     # x fastest
    write("sackboy_96.sdf", 96, 1.5, lambda x, y, z, size: np.minimum(np.sqrt((x - 72) ** 2 + (y - 45) ** 2 + (z - 72) ** 2) - 38,
                                                                     np.sqrt((x - 72) ** 2 + (y - 100) ** 2 + (z - 72) ** 2) - 28))
    write("bunny_64.sdf", 64, 0.00225, lambda x, y, z, size: np.sqrt((np.sqrt((x - size / 2) ** 2 + (z - size / 2) ** 2) - 0.045) ** 2 + (y - 0.03) ** 2) - 0.02)


def sdfunction_synth():
    s = json.load(open(os.path.join(REF, "07_sdfunction", "sdfunction.json")))
    s["textures"][0]["uri"] = "../03_volume/textures/sky.hdr"      # byte-identical to tests/07_sdfunction/textures/sky.hdr
    s["volumes"] = [{"name": "sackboy", "uri": "../06_gridsdf_synth/sdfs/sackboy_synth.sdf", "binary": True},
                    {"name": "bunny", "uri": "../06_gridsdf_synth/sdfs/bunny_synth.sdf", "binary": False}]
    first = len(s["materials"])
    s["materials"] += [
        {"name": "frosted", "type": "transparent", "color": [0.9, 0.9, 1.0], "roughness": 0.2, "opacity": 0.7},
        {"name": "mirror", "type": "reflective", "color": [0.9, 0.6, 0.3], "roughness": 0},
        {"name": "thin", "type": "transparent", "color": [0.7, 1.0, 0.7], "roughness": 0},
        {"name": "pbr", "type": "gltfpbr", "color": [0.8, 0.3, 0.3], "roughness": 0.3, "metallic": 0.7},
        {"name": "ground", "type": "matte", "color": [0.3, 0.35, 0.4]},
        {"name": "glass", "type": "refractive", "color": [1.0, 1.0, 1.0], "roughness": 0},
    ]

    def at(x, y, z):   # eval_sdf_scene applies the FORWARD frame to the world point (yocto_sdfs.cpp:13): local = p + o
        return [1, 0, 0, 0, 1, 0, 0, 0, 1, -x, -y, -z]

    s["sdfunctions"] += [
        {"name": "sphere1", "type": "sphere", "radius": 0.05, "material": first + 0, "frame": at(-0.1, 0.05, 0.2)},
        {"name": "sphere2", "type": "sphere", "radius": 0.04, "material": first + 1, "frame": at(0.0, 0.04, 0.35)},
        {"name": "sphere3", "type": "sphere", "radius": 0.03, "material": first + 2, "frame": at(-0.2, 0.03, 0.05)},
        {"name": "bbox1", "type": "bbox", "thickness": 0.008, "whd": [0.05, 0.05, 0.05], "material": first + 3,
         "frame": at(-0.25, 0.06, 0.3)},
        {"name": "plane1_far", "type": "plane", "material": first + 4, "frame": at(0, -0.3, 0)},
        {"name": "sphere4", "type": "sphere", "radius": 0.03, "material": first + 5, "frame": at(-0.35, 0.03, -0.2)},
    ]
    os.makedirs(os.path.join(OUT, "07_sdfunction_synth"), exist_ok=True)
    dump(os.path.join(OUT, "07_sdfunction_synth", "sdfunction_synth.json"), s)


def volume_lobes():
    s = json.load(open(os.path.join(REF, "03_volume", "volume.json")))
    for shp in s["shapes"]:
        shp["uri"] = "../03_volume/" + shp["uri"]
    for t in s["textures"]:
        t["uri"] = "../03_volume/" + t["uri"]
    byname = {m["name"]: m for m in s["materials"]}
    for name, new in {
        "glass": {"type": "reflective", "color": [0.9, 0.7, 0.4], "roughness": 0},
        "jade": {"type": "transparent", "color": [0.6, 0.9, 0.6], "roughness": 0.3},
        "smoke": {"type": "gltfpbr", "color": [0.8, 0.3, 0.3], "roughness": 0.4, "metallic": 0.6},
        "cloud": {"type": "transparent", "color": [0.8, 0.8, 1.0], "roughness": 0, "opacity": 0.6},
        "skin": {"type": "subsurface", "color": [0.76, 0.48, 0.23], "roughness": 0.3, "scattering": [0.436, 0.227, 0.131],
                 "scanisotropy": -0.8, "trdepth": 0.001},
    }.items():
        m = byname[name]
        keep = m["name"]
        m.clear()
        m.update({"name": keep, **new})
    s["materials"] += [{"name": "brushed", "type": "reflective", "color": [0.7, 0.7, 0.8], "roughness": 0.35},
                       {"name": "glow", "type": "matte", "emission": [3, 2.5, 2], "color": [0, 0, 0]}]
    n = len(s["materials"])
    s["instances"] += [{"name": "brushed", "frame": [1, 0, 0, 0, 1, 0, 0, 0, 1, 0.6, 0, 0], "shape": 1, "material": n - 2},
                       {"name": "glow", "frame": [1, 0, 0, 0, 1, 0, 0, 0, 1, -0.6, 0, 0.1], "shape": 1, "material": n - 1}]
    os.makedirs(os.path.join(OUT, "03_volume_lobes"), exist_ok=True)
    dump(os.path.join(OUT, "03_volume_lobes", "volume_lobes.json"), s)


if __name__ == "__main__":
    surface_min()
    subdiv_synth()
    head1ss_sub()
    gridsdf_synth()
    gridsdf_full()
    sdfunction_synth()
    volume_lobes()
    print("scenes written under", OUT)
