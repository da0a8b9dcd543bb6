"""Multi-GPU path, rehearsed on CPU: the tile partition of include/vpt.h (vpt_layout) splits the
pixel set into disjoint per-rank slot lists, every rank renders only its own pixels, an
all_gather of the tile buffers + the resolve map reassembles a frame that is BIT-IDENTICAL to the
single-rank render (pixels own their RNG streams).  world_size 2 over gloo; the
per-rank renderer here is the CPU oracle restricted to the rank's pixel list (this is a test of the
sharding logic, not of the kernels — tests/test_gpu_parity.py repeats it on the device with virtual ranks)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, SCENE_03


@pytest.mark.parametrize("nranks,tile", [(1, 8), (2, 8), (3, 16), (8, 8), (4, 32)])
def test_layout_is_a_partition(vpt, nranks, tile):
    w, h = 150, 67  # ragged: neither a multiple of 8
    seen = np.zeros(w * h, np.int32)
    sizes = set()
    for r in range(nranks):
        lay = vpt.VptLayout(w, h, tile, tile, r, nranks)
        idx = vpt.layout_pixel_index(lay)
        sizes.add(len(idx))
        assert len(idx) == vpt.layout_slots(lay) and len(idx) % 64 == 0
        valid = idx[idx >= 0]
        assert len(np.unique(valid)) == len(valid)
        seen[valid] += 1
    assert (seen == 1).all(), "every pixel must belong to exactly one rank"
    assert len(sizes) == 1, "all ranks allocate the same slot count (needed by all_gather)"


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import vpt_loader
    vpt = vpt_loader.load()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = vpt.HostScene(SCENE_03)
    p = vpt.PathtraceParams(resolution=96, samples=3, shader="volpathtrace", bounces=64)
    full = scene.make_state(p)
    lay = vpt.VptLayout(full.width, full.height, 8, 8, rank, world)
    idx = vpt.layout_pixel_index(lay)
    mine = idx[idx >= 0]
    # the rank renders ONLY its own pixels (vpt_oracle_render_pixels): every other pixel of its state must come back
    # exactly as make_state left it - radiance and hit count zero, the RNG stream at its seed
    st = full.copy()
    oracle_lib.oracle_render(scene, p, st, 3, nthreads=2, pixels=mine)
    foreign = np.ones(full.width * full.height, bool)
    foreign[mine] = False
    assert (st.hits.reshape(-1)[mine] == 3).all() and (st.hits.reshape(-1)[foreign] == 0).all(), "a rank touched a foreign pixel"
    assert not st.image.reshape(-1, 4)[foreign].any()
    assert np.array_equal(st.rngs.reshape(-1, 2)[foreign], full.rngs.reshape(-1, 2)[foreign])
    assert (st.rngs.reshape(-1, 2)[mine, 0] != full.rngs.reshape(-1, 2)[mine, 0]).all()
    tiles = np.zeros((len(idx), 4), np.float32)
    tiles[idx >= 0] = st.image.reshape(-1, 4)[mine]
    gathered = [torch.zeros(len(idx), 4) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(tiles))
    if rank == 0:
        frame = np.zeros((full.height * full.width, 4), np.float32)
        for r in range(world):
            ridx = vpt.layout_pixel_index(vpt.VptLayout(full.width, full.height, 8, 8, r, world))
            frame[ridx[ridx >= 0]] = gathered[r].numpy()[ridx >= 0]
        np.save(out_path, frame.reshape(full.height, full.width, 4))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_the_single_rank_frame(vpt, scene03, oracle, tmp_path):
    """each of the two gloo ranks renders only its tiles' pixels (the worker asserts that nothing else moved); the
    gathered frame equals the single-rank render bit for bit"""
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, 29517, out), nprocs=2, join=True)
    p = vpt.PathtraceParams(resolution=96, samples=3, shader="volpathtrace", bounces=64)
    st = scene03.make_state(p)
    oracle.oracle_render(scene03, p, st, 3)
    frame = np.load(out)
    assert np.array_equal(frame.view(np.uint32), st.image.view(np.uint32))


def test_rendering_a_pixel_subset_leaves_the_rest_alone(vpt, scene03, oracle):
    """the checker's own contract: vpt_oracle_render_pixels advances exactly the listed pixels, each as the whole-frame
    render would"""
    p = vpt.PathtraceParams(resolution=64, samples=2, shader="volpathtrace", bounces=64)
    whole, part = scene03.make_state(p), scene03.make_state(p)
    seed = part.copy()
    oracle.oracle_render(scene03, p, whole, 2)
    some = np.arange(0, whole.width * whole.height, 7, dtype=np.int32)
    oracle.oracle_render(scene03, p, part, 2, pixels=some)
    rest = np.ones(whole.width * whole.height, bool)
    rest[some] = False
    assert np.array_equal(part.image.reshape(-1, 4)[some].view(np.uint32), whole.image.reshape(-1, 4)[some].view(np.uint32))
    assert np.array_equal(part.rngs.reshape(-1, 2)[some], whole.rngs.reshape(-1, 2)[some])
    assert not part.image.reshape(-1, 4)[rest].any() and np.array_equal(part.rngs.reshape(-1, 2)[rest], seed.rngs.reshape(-1, 2)[rest])
    assert part.samples == 2
