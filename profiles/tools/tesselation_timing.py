#!/usr/bin/env python3
"""Host against device, stage by stage, for tesselate_surface on the reference's suzanne cage (tests/01_surface, 2 and 5 levels):
    python profiles/tools/tesselation_timing.py            (on a GPU box; writes nothing, prints the table of profiles/r04_tesselation_timing.txt)
Stages: the Catmull-Clark levels (topology on the host in both columns; vertex arithmetic host / vpt_subdivide_vertices), quads_normals
(host / vpt_vertex_normals), displacement by an 8-bit texture (host / vpt_displace_vertices), triangles_normals after it.  Device times
include the PCIe round trip of every call (the entry points are synchronous, host arrays in and out): that is what a caller pays today."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vpt_loader  # noqa: E402


def read_cage(path):
    v, f = [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            v.append([float(x) for x in p[1:4]])
        elif p[0] == "f":
            idx = [int(t.split("/")[0]) - 1 for t in p[1:]]
            f.append(idx if len(idx) == 4 else idx + [idx[-1]])
    return np.array(f, np.int32), np.array(v, np.float32)


def best(fn, reps=5):
    out, times = None, []
    for _ in range(reps):
        t = time.perf_counter()
        out = fn()
        times.append(time.perf_counter() - t)
    return out, min(times) * 1e3


def main():
    vpt = vpt_loader.load()
    cage = os.path.join(ROOT, "tests/golden/scenes/01_surface_min/subdivs/suzanne-subdiv.obj")
    quads0, verts0 = read_cage(cage)
    rng = np.random.default_rng(3)
    tex = rng.integers(0, 256, size=(512, 512, 4), dtype=np.uint8)
    vpt.catmullclark(quads0, verts0, device=0)   # first touch of the device
    print(f"cage {os.path.basename(cage)}: {len(quads0)} faces, {len(verts0)} vertices; ms = best of 5; device columns include the host <-> device copies of each call")
    print(f"{'levels':>6} {'stage':34s} {'vertices':>9} {'host ms':>9} {'device ms':>10}")
    for levels in (2, 5):
        quads, verts = quads0, verts0
        th = td = 0.0
        for _ in range(levels):
            (qh, vh), a = best(lambda: vpt.catmullclark(quads, verts))
            (qd, vd), b = best(lambda: vpt.catmullclark(quads, verts, device=0))
            assert np.array_equal(vh.view(np.uint32), vd.view(np.uint32))
            quads, verts, th, td = qh, vh, th + a, td + b
        print(f"{levels:6d} {'Catmull-Clark levels (topology + vertices)':34s} {len(verts):9d} {th:9.2f} {td:10.2f}")
        nh, a = best(lambda: vpt.vertex_normals(verts, quads))
        nd, b = best(lambda: vpt.vertex_normals(verts, quads, device=0))
        assert np.array_equal(nh.view(np.uint32), nd.view(np.uint32))
        print(f"{levels:6d} {'quads_normals':34s} {len(verts):9d} {a:9.2f} {b:10.2f}")
        uv = rng.uniform(0, 1, size=(len(verts), 2)).astype(np.float32)
        ph, a = best(lambda: vpt.displace_vertices(tex, False, 0.02, verts, nh, uv))
        pd, b = best(lambda: vpt.displace_vertices(tex, False, 0.02, verts, nh, uv, device=0))
        assert np.array_equal(ph.view(np.uint32), pd.view(np.uint32))
        print(f"{levels:6d} {'displacement (512 x 512 8-bit)':34s} {len(verts):9d} {a:9.2f} {b:10.2f}")
        tris = np.concatenate([quads[:, [0, 1, 3]], quads[quads[:, 2] != quads[:, 3]][:, [2, 3, 1]]]).astype(np.int32)
        th2, a = best(lambda: vpt.vertex_normals(ph, tris))
        td2, b = best(lambda: vpt.vertex_normals(ph, tris, device=0))
        assert np.array_equal(th2.view(np.uint32), td2.view(np.uint32))
        print(f"{levels:6d} {'triangles_normals (after it)':34s} {len(verts):9d} {a:9.2f} {b:10.2f}")


if __name__ == "__main__":
    main()
