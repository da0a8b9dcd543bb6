"""the reference's BVH build (build_bvh + split_middle, libs/yocto/yocto_bvh.cpp:411-507) on the device.
vpt_build_bvh must return the reference's arrays themselves - node ids in its creation order, its primitive permutation
(libstdc++'s std::partition), its float bits - so the checks are equalities of bytes:

* every scene: the FNV-1a hashes of all node and primitive arrays after make_bvh_device equal the ones the REFERENCE
  printed for its own build (tests/golden/03_volume_stats.json, substitute_stats.json: made by oracle/_ref/ref_driver
  --stats) - 144 046 triangles in the largest tree;
* synthetic boxes aimed at the build's corner cases, against the host build (itself pinned by those hashes): ranges of
  1..9 boxes around the leaf size, identical centres (split_middle's `csize == 0` halving), a partition that puts
  everything on one side (the `middle == start || middle == end` halving), many equal coordinates (ties in the
  predicate and in the bounds), -0.0 against +0.0 in the bounds (merge() keeps the later of two equal values: the
  sign of a zero tells which), clustered sizes, a degenerate line of boxes;
* a render from the device-built scene is bit-identical to one from the host-built scene.

CPU part (`-m "not gpu"`): the host build against the reference's hashes is tests/test_host_pipeline.py; here only the
checker's own determinism and the argument checks that need no device."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

SCENES = ["03_volume/volume.json", "01_surface_min/surface_min.json", "05_head1ss_sub/head1ss_sub.json", "06_gridsdf_synth/gridsdf_synth.json",
          "07_sdfunction_synth/sdfunction_synth.json", "03_volume_lobes/volume_lobes.json"]


def _reference_stats(scene_file):
    if scene_file.startswith("03_volume/"):
        return json.load(open(os.path.join(GOLDEN, "03_volume_stats.json")))
    return json.load(open(os.path.join(GOLDEN, "substitute_stats.json")))[scene_file]


def _bvh_part(stats):
    return {"scene_bvh": stats["scene_bvh"], "shapes": [{k: s[k] for k in ("bvh_nodes", "bvh_nodes_fnv", "bvh_prims_fnv")} for s in stats["shapes"]]}


def _boxes(rng, n, kind):
    lo = rng.random((n, 3), dtype=np.float32) * 10 - 5
    size = rng.random((n, 3), dtype=np.float32)
    if kind == "uniform":
        pass
    elif kind == "same_centre":      # csize == 0: halves with axis 0, nothing moves (half sizes are multiples of 1/4: exact)
        half = np.floor(size * 8 + 1).astype(np.float32) / 4
        return np.concatenate([np.float32(1.5) - half, np.float32(1.5) + half], axis=1).astype(np.float32)
    elif kind == "grid":             # few distinct coordinates: ties everywhere
        lo = np.floor(lo).astype(np.float32)
        size = np.ones_like(size)
    elif kind == "zeros":            # boxes touching 0 from both sides, with both signs of zero
        lo = np.where(rng.random((n, 3)) < 0.5, np.float32(-0.0), np.float32(0.0)).astype(np.float32) - (rng.random((n, 3)) < 0.3) * size
        size = np.where(rng.random((n, 3)) < 0.5, size, np.float32(0.0)).astype(np.float32)
    elif kind == "line":             # all centres on one axis
        lo[:, 1:] = 0
        size[:, 1:] = 1
    elif kind == "clusters":         # two far clusters of very different population: lopsided partitions
        far = rng.random(n) < 0.03
        lo[far] += 1000
    elif kind == "one_sided":        # one huge box stretches the centroid box: the partition leaves all but one element on a side
        lo[0] = 1e6
    hi = (lo + size).astype(np.float32)
    if kind == "zeros":
        hi = np.where(hi == 0, np.where(rng.random((n, 3)) < 0.5, np.float32(-0.0), np.float32(0.0)), hi).astype(np.float32)
    return np.concatenate([lo, hi], axis=1).astype(np.float32)


def _equal(a, b):
    return a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()


def test_host_build_is_deterministic_and_well_formed(vpt):
    rng = np.random.default_rng(7)
    bb = _boxes(rng, 1000, "uniform")
    nodes, prims = vpt.build_bvh(bb, device=None)
    again = vpt.build_bvh(bb, device=None)
    assert _equal((nodes, prims), again)
    assert sorted(prims.tolist()) == list(range(1000))
    leaves = nodes[nodes["internal"] == 0]
    assert leaves["num"].sum() == 1000 and leaves["num"].max() <= 4
    inner = nodes[nodes["internal"] == 1]
    assert (inner["num"] == 2).all() and len(nodes) == 2 * len(inner) + 1
    # creation order: the children of the k-th internal node popped sit at 1 + 2 k; the root is popped first
    assert nodes[0]["start"] == 1


def test_empty_build_needs_no_device(vpt):
    nodes, prims = vpt.build_bvh(np.zeros((0, 6), np.float32), device=0)   # answered before any HIP call
    host = vpt.build_bvh(np.zeros((0, 6), np.float32), device=None)
    assert len(prims) == 0 and _equal((nodes, prims), host)
    assert len(nodes) == 1 and nodes[0]["internal"] == 0 and nodes[0]["num"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("scene_file", SCENES)
def test_device_build_reproduces_the_references_trees(vpt, scene_file):
    scene = vpt.HostScene(os.path.join(GOLDEN, "scenes", scene_file), bvh_device=0)
    mine = _bvh_part(json.loads(scene.stats()))
    ref = _bvh_part(_reference_stats(scene_file))
    assert mine == ref


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["uniform", "same_centre", "grid", "zeros", "line", "clusters", "one_sided"])
def test_device_build_equals_host_build_on_corner_cases(vpt, kind):
    rng = np.random.default_rng(len(kind) * 131 + ord(kind[0]))
    sizes = list(range(1, 10)) + [16, 17, 31, 64, 100, 257, 1000, 4099, 20000]
    for n in sizes:
        bb = _boxes(rng, n, kind)
        dev = vpt.build_bvh(bb, device=0)
        host = vpt.build_bvh(bb, device=None)
        if not _equal(dev, host):
            dn, hn = dev[0], host[0]
            first = next((i for i in range(min(len(dn), len(hn))) if dn[i].tobytes() != hn[i].tobytes()), None)
            pytest.fail(f"{kind} n={n}: nodes {len(dn)} vs {len(hn)}, first differing node {first}: {dn[first] if first is not None else None} vs "
                        f"{hn[first] if first is not None else None}; primitives equal: {np.array_equal(dev[1], host[1])}")


@pytest.mark.gpu
def test_device_build_argument_errors(vpt):
    import ctypes as C
    bb = np.zeros((8, 6), np.float32)
    nodes = np.zeros(16, vpt.BVH_NODE)
    prims = np.zeros(8, np.int32)
    count = C.c_int()
    assert vpt.hip.vpt_build_bvh(0, bb.ctypes.data, 8, nodes.ctypes.data, 3, C.byref(count), prims.ctypes.data) == -1   # capacity < 2 n - 1
    assert b"capacity" in vpt.hip.vpt_last_error()
    assert vpt.hip.vpt_build_bvh(99, bb.ctypes.data, 8, nodes.ctypes.data, 16, C.byref(count), prims.ctypes.data) == -1
    assert vpt.hip.vpt_build_bvh(0, None, 8, nodes.ctypes.data, 16, C.byref(count), prims.ctypes.data) == -1


@pytest.mark.gpu
def test_render_from_device_built_scene_is_bit_identical(vpt, scene03, dev03):
    built = vpt.HostScene(os.path.join(GOLDEN, "scenes", "03_volume", "volume.json"), bvh_device=0)
    dev = vpt.DeviceScene(built, 0)
    p = vpt.PathtraceParams(resolution=96, samples=4, shader="volpathtrace", bounces=64)
    a, b = scene03.make_state(p), built.make_state(p)
    dev03.pathtrace_samples(a, p, 4)
    dev.pathtrace_samples(b, p, 4)
    assert np.array_equal(a.image.view(np.uint32), b.image.view(np.uint32)) and np.array_equal(a.rngs, b.rngs)
