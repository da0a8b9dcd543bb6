"""Helper for the GPU tests: render one configuration in a fresh process (build switches such as VPT_NO_LEAN / VPT_HIP_LIB are
read once per process) and save the resulting pathtrace_state.

  python render_state.py <scene.json> <shader> <resolution> <spp> <bounces> <out.npz>
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vpt_loader  # noqa: E402


def main():
    scene_file, shader, res, spp, bounces, out = sys.argv[1:7]
    vpt = vpt_loader.load()
    scene = vpt.HostScene(scene_file)
    dev = vpt.DeviceScene(scene, 0)
    p = vpt.PathtraceParams(resolution=int(res), samples=int(spp), shader=shader, bounces=int(bounces))
    st = scene.make_state(p)
    dev.pathtrace_samples(st, p, int(spp))
    np.savez(out, image=st.image, hits=st.hits, rngs=st.rngs)


if __name__ == "__main__":
    main()
