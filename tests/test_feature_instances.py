"""Kernel instances per scene-feature set (vpt_scene.hip.h: VPT_FEAT_*): a scene whose lights need neither BVH hops nor SDF
marches runs a mesh-kernel instance compiled without that code, an SDF scene without emissive meshes runs an implicit-kernel
instance without the mesh-light walks.  The instances differ in the code they leave out, never in the arithmetic of what
they keep: the general instance (forced with VPT_NO_LEAN=1) must give the same state bit for bit on the scenes that
normally take the lean one.  (The scenes that need the general instances - 03_volume_lobes with its emissive mesh, the
mesh shaders on the SDF-light scenes - are covered against the reference by test_gpu_parity.py as before.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

CASES = {
    "k1_volpath_03": ("03_volume/volume.json", "volpathtrace", 128, 6, 64),
    "k1_path_surface": ("01_surface_min/surface_min.json", "pathtrace", 128, 4, 8),
    "k1_volpath_head_spill": ("05_head1ss_sub/head1ss_sub.json", "volpathtrace", 96, 2, 32),   # the HBM-backed stack variant
    "k2_implicit_gridsdf": ("06_gridsdf_synth/gridsdf_synth.json", "implicit", 128, 4, 4),
    "k2_implicit_sdfunction": ("07_sdfunction_synth/sdfunction_synth.json", "implicit", 128, 4, 6),
}


def _render(tmp_path, tag, env, scene_file, shader, res, spp, bounces):
    out = str(tmp_path / f"{tag}.npz")
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "render_state.py"), os.path.join(GOLDEN, "scenes", scene_file), shader, str(res), str(spp),
                        str(bounces), out], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("name", sorted(CASES))
def test_general_and_lean_instances_agree_bit_for_bit(tmp_path, name):
    lean = _render(tmp_path, "lean", {}, *CASES[name])
    general = _render(tmp_path, "general", {"VPT_NO_LEAN": "1"}, *CASES[name])
    assert np.array_equal(lean["image"].view(np.uint32), general["image"].view(np.uint32))
    assert np.array_equal(lean["rngs"], general["rngs"]) and np.array_equal(lean["hits"], general["hits"])
    assert lean["image"].any()
