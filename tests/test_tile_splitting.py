"""Tile splitting (include/vpt.h, vpt_capi.hip): when a layout shares the frame among ranks, the mesh kernels run the
costliest tiles as several partly filled waves so that a launch is not as long as its costliest tile.  It is a change of
schedule only: every pixel keeps its own RNG stream and accumulator, so the state must be bit-identical with and without
it - checked here on one GPU with virtual ranks and with small frames, across the call in which the decision is taken (call 1: pilot + unsplit
launch that measures the tiles; call 2: decision, first split launch; call 3: split launch in measured order)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
SCENE = os.path.join(GOLDEN, "scenes", "03_volume", "volume.json")


def _run(tmp_path, tag, env, res, nranks, rank, spp, calls, scene=SCENE, shader="volpathtrace", bounces=64):
    out = str(tmp_path / f"{tag}.npz")
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "render_rank_state.py"), scene, str(res), str(nranks), str(rank), str(spp), str(calls), out,
                        shader, str(bounces)], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


def _same_state(a, b):
    return (np.array_equal(a["image"].view(np.uint32), b["image"].view(np.uint32)) and np.array_equal(a["rngs"], b["rngs"])
            and np.array_equal(a["hits"], b["hits"]))


@pytest.mark.parametrize("k", [1, 3, 6])
def test_every_tile_split_gives_the_same_state(tmp_path, k):
    """VPT_SPLIT_K forces every tile into 2^k waves (64, 8 and 1 lane per wave at k = 0, 3, 6)"""
    base = _run(tmp_path, "base", {"VPT_SPLIT": "0"}, 400, 2, 1, 16, 3)
    split = _run(tmp_path, f"k{k}", {"VPT_SPLIT_K": str(k)}, 400, 2, 1, 16, 3)
    tiles = int(base["tiles"])
    assert list(base["waves"]) == [tiles] * 3
    assert list(split["waves"]) == [tiles, tiles << k, tiles << k]      # the second call takes the decision
    assert _same_state(base, split)
    assert (base["hits"].max() == 48) and (np.count_nonzero(base["hits"]) > 0.45 * base["hits"].size)


def test_adaptive_split_of_a_shared_frame_gives_the_same_state(tmp_path):
    """default policy on rank 3 of 8 of the bench frame: some tiles are split (more waves than tiles), the state is the unsplit one"""
    base = _run(tmp_path, "base", {"VPT_SPLIT": "0"}, 1280, 8, 3, 32, 3)
    auto = _run(tmp_path, "auto", {}, 1280, 8, 3, 32, 3)
    tiles = int(base["tiles"])
    assert list(base["waves"]) == [tiles] * 3
    assert auto["waves"][0] == tiles and auto["waves"][1] > tiles and auto["waves"][2] == auto["waves"][1]
    assert _same_state(base, auto)
    print("rank 3 of 8, 1280x533x32spp: unsplit", base["ms"], "ms; split", auto["ms"], "ms;", int(auto["waves"][1]), "waves for", tiles, "tiles")
    # timing is printed, not asserted (one run each on a shared GPU: a correctness test must not depend on the clock); measured on
    # MI355X the split launch takes ~0.6 of the unsplit one (profiles/r02_strong_scaling_rehearsal.txt holds the measured series)


def test_a_full_size_frame_on_one_gpu_is_left_alone(tmp_path):
    """1280x533 = 10 720 tiles on 3 072 wave slots: the launch is bound by total work, the policy is not even consulted"""
    one = _run(tmp_path, "one", {}, 1280, 1, 0, 16, 3)
    assert list(one["waves"]) == [int(one["tiles"])] * 3


def test_a_small_frame_on_one_gpu_is_split_and_unchanged(tmp_path):
    """the reference's default resolution (720x300, 3 420 tiles) leaves most wave slots idle behind its costliest tiles"""
    base = _run(tmp_path, "base", {"VPT_SPLIT": "0"}, 720, 1, 0, 32, 3)
    auto = _run(tmp_path, "auto", {}, 720, 1, 0, 32, 3)
    tiles = int(base["tiles"])
    assert list(base["waves"]) == [tiles] * 3 and auto["waves"][0] == tiles and auto["waves"][2] > tiles
    assert _same_state(base, auto)
    print("720x300x32spp on one GPU: unsplit", base["ms"], "ms; split", auto["ms"], "ms")
    # (timing printed, not asserted; measured on MI355X at 256 spp: 212 -> 152 ms, profiles/r02_small_frames_tile_splitting.txt)


@pytest.mark.parametrize("k", [1, 4])
def test_split_tiles_of_the_implicit_kernel_give_the_same_state(tmp_path, k):
    """K2 takes the same lane table (round 4): every tile of the voxel scene as 2^k partly filled waves - whose quorums scale with their pixels
    and whose scene rounds run in the group form - ends in the unsplit state; and the default policy, which finds nothing to gain on a full
    1280-wide frame (K2's partly filled waves are 0.82 / 0.67 / 0.63 as long as the full one: profiles/r04_k2_lane_histogram.txt), leaves it alone"""
    sdf = os.path.join(GOLDEN, "scenes", "06_gridsdf_synth", "gridsdf_synth.json")
    base = _run(tmp_path, "base", {"VPT_K2_SPLIT": "0"}, 320, 1, 0, 16, 3, scene=sdf, shader="implicit", bounces=4)
    split = _run(tmp_path, f"k{k}", {"VPT_SPLIT_K": str(k)}, 320, 1, 0, 16, 3, scene=sdf, shader="implicit", bounces=4)
    tiles = int(base["tiles"])
    assert list(base["waves"]) == [tiles] * 3
    assert list(split["waves"]) == [tiles, tiles << k, tiles << k]
    assert _same_state(base, split)
