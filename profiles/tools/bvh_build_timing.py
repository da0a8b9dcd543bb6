#!/usr/bin/env python3
"""vpt_build_bvh (device) against the host build (the reference's algorithm, g++ -O2, one thread): wall time per build
including the transfers of boxes, nodes and primitive order.  Run from the repo root on a GPU box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vpt_loader

vpt = vpt_loader.load()
rng = np.random.default_rng(3)
vpt.build_bvh(np.zeros((64, 6), np.float32), device=0)   # context creation is not part of a build
print(f"{'boxes':>10} {'host ms':>10} {'device ms':>10} {'ratio':>7}  nodes  equal")
for n in (1_000, 10_000, 144_046, 1_000_000, 4_000_000):
    lo = rng.random((n, 3), dtype=np.float32) * 100
    bb = np.concatenate([lo, lo + rng.random((n, 3), dtype=np.float32) * 0.2], axis=1)
    t0 = time.perf_counter()
    host = vpt.build_bvh(bb, device=None)
    t1 = time.perf_counter()
    dev = vpt.build_bvh(bb, device=0)
    t2 = time.perf_counter()
    dev = vpt.build_bvh(bb, device=0)
    t3 = time.perf_counter()
    same = host[0].tobytes() == dev[0].tobytes() and host[1].tobytes() == dev[1].tobytes()
    print(f"{n:10d} {(t1 - t0) * 1e3:10.1f} {(t3 - t2) * 1e3:10.1f} {(t1 - t0) / (t3 - t2):7.1f}  {len(dev[0])}  {same}")
scene_file = os.path.join(ROOT, "tests", "golden", "scenes", "05_head1ss_sub", "head1ss_sub.json")
t0 = time.perf_counter()
s = vpt.HostScene(scene_file)
t1 = time.perf_counter()
err = vpt.C.create_string_buffer(256)
vpt.host.vpth_scene_rebuild_bvh_device(s.handle, 0, err, 256)
t2 = time.perf_counter()
print(f"05_head1ss_sub: load + host make_bvh + lights {1e3 * (t1 - t0):.1f} ms; make_bvh_device + flatten alone {1e3 * (t2 - t1):.1f} ms")
