#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ by running the REFERENCE'S OWN renderer
(oracle/_ref/ref_driver, built from /root/reference by oracle/Makefile).  Run in the build
container only; the outputs (data: float32 states, stats, one JPEG) are committed so that the
tests can pin the oracle and the host pipeline on machines where the reference does not exist.

    python tests/golden/make_fixtures.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

REF_SCENE = "/root/reference/tests/03_volume/volume.json"
SCENES = os.path.join(HERE, "scenes")
from cases import CASES as _CASES, EXTRA as _EXTRA  # noqa: E402

# substitute scenes for the configs whose assets are missing (tests/golden/make_scenes.py); the reference's
# own loader and renderer read them from here.  (name, scene, shader, resolution, samples, bounces, noimplicitmis)
EXTRA = [(name, *v) for name, v in _EXTRA.items()]
# (name, shader, resolution, samples, bounces)
CASES = [(name, *v) for name, v in _CASES.items()]


def main():
    assert O.have_reference(), "build oracle/_ref first: make -C oracle ref"
    out = {}
    for name, shader, res, spp, bounces in CASES:
        w, h, image, hits, rngs, info = O.reference_render(REF_SCENE, shader, res, spp, bounces)
        out[name + "_image"] = image
        out[name + "_rngs"] = rngs
        out[name + "_meta"] = np.array([w, h, spp, bounces], np.int32)
        print(name, w, h)
    np.savez_compressed(os.path.join(HERE, "03_volume_states.npz"), **out)
    out = {}
    all_stats = {}
    for name, scene, shader, res, spp, bounces, nomis in EXTRA:
        path = os.path.join(SCENES, scene)
        w, h, image, hits, rngs, info, stats = O.reference_render(path, shader, res, spp, bounces, noimplicitmis=nomis, stats=True)
        out[name + "_image"] = image
        out[name + "_rngs"] = rngs
        out[name + "_meta"] = np.array([w, h, spp, bounces, int(nomis)], np.int32)
        stats.pop("state")
        all_stats[scene] = stats
        print(name, w, h)
    np.savez_compressed(os.path.join(HERE, "substitute_states.npz"), **out)
    json.dump(all_stats, open(os.path.join(HERE, "substitute_stats.json"), "w"), indent=1)
    # structural statistics + hashes of the reference's scene / bvh / lights
    *_, stats = O.reference_render(REF_SCENE, "volpathtrace", 64, 1, 4, stats=True)
    stats.pop("state")
    json.dump(stats, open(os.path.join(HERE, "03_volume_stats.json"), "w"), indent=1)
    # the reference's own JPEG of a small render + the state it was made from
    jpg = os.path.join(HERE, "03_volume_128_8.jpg")
    w, h, image, hits, rngs, info = O.reference_render(REF_SCENE, "volpathtrace", 128, 8, 64, output=jpg)
    np.savez_compressed(os.path.join(HERE, "03_volume_128_8_state.npz"), image=image, meta=np.array([w, h, 8], np.int32))
    print("jpeg", w, h, os.path.getsize(jpg))


if __name__ == "__main__":
    main()
