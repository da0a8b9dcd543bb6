"""The C++ host mirror of the reference's renderer API (volumetric-path-tracer_amd/host/vpt_host.h, namespace vpt): a program written
against it - the way a caller of the reference would be - compiled here and run on the GPU box.  Covers what the Python harness
cannot reach: the device-scene cache behind pathtrace_samples, which must notice in-place edits of the scene's small tables
(the cache's fingerprint used to sample 256 bytes of each array)."""
import os
import subprocess

import pytest

from conftest import ROOT, SCENE_03

pytestmark = pytest.mark.gpu
PKG = os.path.join(ROOT, "volumetric-path-tracer_amd")

PROGRAM = r'''
#include <cstdio>
#include <cstring>
#include "vpt_host.h"
using namespace vpt;
static pathtrace_state render(const scene_data& scene, const bvh_scene& bvh, const pathtrace_lights& lights, const pathtrace_params& params) {
  auto state = make_state(scene, params);
  pathtrace_samples(state, scene, bvh, lights, params, params.samples);
  return state;
}
static bool same(const pathtrace_state& a, const pathtrace_state& b) { return memcmp(a.image.data(), b.image.data(), a.image.size() * sizeof(vec4f)) == 0; }
int main(int argc, char** argv) {
  auto scene = scene_data{};
  auto error = string{};
  if (!load_scene(argv[1], scene, error)) return printf("load: %s\n", error.c_str()), 2;
  tesselate_surfaces(scene);
  auto params = pathtrace_params{};
  params.resolution = 96, params.samples = 4, params.shader = pathtrace_shader_type::volpathtrace, params.bounces = 16;   // the edited material is a medium: its colour is its density
  auto bvh = make_bvh(scene, params);
  auto lights = make_lights(scene, params);
  auto first = render(scene, bvh, lights, params);
  auto again = render(scene, bvh, lights, params);
  if (!same(first, again)) return printf("not deterministic\n"), 3;
  // in-place edits far beyond the first 256 bytes of their tables: the fourth material of eight (bytes 252-336 of the table: outside the 256-byte head and tail the old fingerprint sampled)
  auto colour = scene.materials[3].color;
  scene.materials[3].color = {0.9f, 0.1f, 0.1f};
  auto recoloured = render(scene, bvh, lights, params);
  if (same(first, recoloured)) return printf("a material edited in place went unnoticed: stale device scene\n"), 4;
  scene.materials[3].color = colour;
  auto restored = render(scene, bvh, lights, params);
  if (!same(first, restored)) return printf("restoring the material did not restore the image\n"), 5;
  auto lens = scene.cameras[0].lens;
  scene.cameras[0].lens = lens * 1.5f;
  auto zoomed = render(scene, bvh, lights, params);
  if (same(first, zoomed)) return printf("a camera edited in place went unnoticed\n"), 6;
  scene.cameras[0].lens = lens;
  pathtrace_release(scene);
  auto released = render(scene, bvh, lights, params);
  if (!same(first, released)) return printf("release + render differs\n"), 7;
  printf("ok\n");
  return 0;
}
'''


def test_in_place_scene_edits_reach_the_device(tmp_path):
    src = tmp_path / "mirror.cpp"
    src.write_text(PROGRAM)
    exe = tmp_path / "mirror"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(PKG, "host"), "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", PKG, "-lvpt_host", "-lvpt_hip", f"-Wl,-rpath,{PKG}"])
    r = subprocess.run([str(exe), SCENE_03], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout, r.stderr)
