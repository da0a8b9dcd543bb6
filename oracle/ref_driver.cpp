// oracle/ref_driver.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Headless driver that links the *reference's own* renderer (compiled from
// /root/reference by oracle/Makefile into oracle/_ref/) and dumps what the
// parity tests need:
//   * the raw float32 pathtrace_state (image, hits, rng) after N calls of the
//     reference's pathtrace_samples()            -> pins oracle/vpt_oracle.cpp
//   * structural statistics + FNV-1a hashes of the reference's bvh / lights /
//     shape / texture arrays                     -> pins the host pipeline
//   * the reference's own JPEG/PNG output        -> pins the output stage
//   * wall-clock seconds of the render loop      -> cpu_baseline "reference"
//
// Everything here is written for this repo; it only *calls* the reference API
// (load_scene, tesselate_surfaces, make_bvh, make_lights, make_state,
// pathtrace_samples, get_render, save_image: yocto_pathtrace.h:119-139,
// yocto_sceneio.h:90,209) in the order of apps/ypathtrace/ypathtrace.cpp:41-87.
//
// usage: ref_driver --scene S [--shader volpathtrace] [--resolution 720]
//          [--samples 16] [--bounces 4] [--stmaxiter 450] [--camera 0]
//          [--noparallel] [--noimplicitmis] [--state out.bin] [--stats out.json]
//          [--output out.jpg|png] [--threads-report]

#include <yocto/yocto_sceneio.h>
#include <yocto_pathtrace/yocto_pathtrace.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

using namespace yocto;

#ifdef VPT_DROPIN
namespace yocto { extern std::string vpt_dropin_scene_file; }   // the drop-in binding's one hook (oracle/ref_dropin_stub.h)
#endif

static uint64_t fnv1a(const void* data, size_t nbytes) {
  auto p = (const unsigned char*)data;
  auto h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < nbytes; i++) {
    h ^= p[i];
    h *= 0x100000001b3ull;
  }
  return h;
}
template <typename T>
static uint64_t fnv1a(const std::vector<T>& v) {
  return fnv1a(v.data(), v.size() * sizeof(T));
}

static void die(const std::string& msg) {
  fprintf(stderr, "ref_driver: %s\n", msg.c_str());
  exit(1);
}

int main(int argc, const char** argv) {
  auto scene_name = std::string{}, state_name = std::string{},
       stats_name = std::string{}, output = std::string{};
  auto params     = pathtrace_params{};
  params.shader   = pathtrace_shader_type::volpathtrace;
  params.samples  = 16;
  for (auto i = 1; i < argc; i++) {
    auto a    = std::string{argv[i]};
    auto next = [&]() -> std::string {
      if (i + 1 >= argc) die("missing value for " + a);
      return argv[++i];
    };
    if (a == "--scene") scene_name = next();
    else if (a == "--state") state_name = next();
    else if (a == "--stats") stats_name = next();
    else if (a == "--output") output = next();
    else if (a == "--resolution") params.resolution = atoi(next().c_str());
    else if (a == "--samples") params.samples = atoi(next().c_str());
    else if (a == "--bounces") params.bounces = atoi(next().c_str());
    else if (a == "--camera") params.camera = atoi(next().c_str());
    else if (a == "--stmaxiter") params.spheretrace_maxiter = atoi(next().c_str());
    else if (a == "--noparallel") params.noparallel = true;
    else if (a == "--noimplicitmis") params.noimplicit_mis = true;
    else if (a == "--shader") {
      auto name  = next();
      auto found = false;
      for (auto k = 0; k < (int)pathtrace_shader_names.size(); k++)
        if (pathtrace_shader_names[k] == name) {
          params.shader = (pathtrace_shader_type)k;
          found         = true;
        }
      if (!found) die("unknown shader " + name);
    } else die("unknown option " + a);
  }
  if (scene_name.empty()) die("--scene required");

  using clk = std::chrono::steady_clock;
  auto t0   = clk::now();
  auto error = std::string{};
  auto scene = scene_data{};
  if (!load_scene(scene_name, scene, error)) die(error);
#ifdef VPT_DROPIN
  yocto::vpt_dropin_scene_file = scene_name;   // the one line the binding asks of the application (SDF type tags)
#endif
  tesselate_surfaces(scene);
  auto bvh    = make_bvh(scene, params);
  auto lights = make_lights(scene, params);
  auto state  = make_state(scene, params);
  auto t1     = clk::now();
  try {
    for (auto sample = 0; sample < params.samples; sample++)
      pathtrace_samples(state, scene, bvh, lights, params);
  } catch (const std::exception& e) {   // handle_errors of the reference's application (yocto_cli.h:364-390)
    die(e.what());
  }
  auto t2 = clk::now();
  auto setup_s  = std::chrono::duration<double>(t1 - t0).count();
  auto render_s = std::chrono::duration<double>(t2 - t1).count();
  auto nsamples = (double)state.width * state.height * state.samples;
  printf(
      "{\"width\": %d, \"height\": %d, \"samples\": %d, \"setup_s\": %.4f, "
      "\"render_s\": %.4f, \"msamples_per_s\": %.4f, \"threads\": %u}\n",
      state.width, state.height, state.samples, setup_s, render_s,
      nsamples / render_s * 1e-6,
      params.noparallel ? 1u : std::thread::hardware_concurrency());

  if (!state_name.empty()) {
    auto f = fopen(state_name.c_str(), "wb");
    if (!f) die("cannot write " + state_name);
    int32_t hdr[4] = {0x53545056 /*"VPTS"*/, state.width, state.height, state.samples};
    fwrite(hdr, 4, 4, f);
    fwrite(state.image.data(), sizeof(vec4f), state.image.size(), f);
    fwrite(state.hits.data(), sizeof(int), state.hits.size(), f);
    static_assert(sizeof(rng_state) == 16, "rng layout");
    fwrite(state.rngs.data(), sizeof(rng_state), state.rngs.size(), f);
    fclose(f);
  }

  if (!stats_name.empty()) {
    auto f = fopen(stats_name.c_str(), "w");
    if (!f) die("cannot write " + stats_name);
    static_assert(sizeof(bvh_node) == 32, "node layout");
    fprintf(f, "{\n \"scene_bvh\": {\"nodes\": %zu, \"prims\": %zu, \"nodes_fnv\": \"%016llx\", \"prims_fnv\": \"%016llx\"},\n",
        bvh.nodes.size(), bvh.primitives.size(),
        (unsigned long long)fnv1a(bvh.nodes), (unsigned long long)fnv1a(bvh.primitives));
    fprintf(f, " \"shapes\": [\n");
    for (auto i = 0; i < (int)scene.shapes.size(); i++) {
      auto& s = scene.shapes[i];
      auto& b = bvh.shapes[i];
      fprintf(f,
          "  {\"positions\": %zu, \"normals\": %zu, \"texcoords\": %zu, \"colors\": %zu, "
          "\"triangles\": %zu, \"quads\": %zu, \"pos_fnv\": \"%016llx\", \"nrm_fnv\": \"%016llx\", "
          "\"uv_fnv\": \"%016llx\", \"tri_fnv\": \"%016llx\", \"quad_fnv\": \"%016llx\", "
          "\"bvh_nodes\": %zu, \"bvh_nodes_fnv\": \"%016llx\", \"bvh_prims_fnv\": \"%016llx\"}%s\n",
          s.positions.size(), s.normals.size(), s.texcoords.size(), s.colors.size(),
          s.triangles.size(), s.quads.size(), (unsigned long long)fnv1a(s.positions),
          (unsigned long long)fnv1a(s.normals), (unsigned long long)fnv1a(s.texcoords),
          (unsigned long long)fnv1a(s.triangles), (unsigned long long)fnv1a(s.quads),
          b.nodes.size(), (unsigned long long)fnv1a(b.nodes),
          (unsigned long long)fnv1a(b.primitives),
          i + 1 < (int)scene.shapes.size() ? "," : "");
    }
    fprintf(f, " ],\n \"textures\": [\n");
    for (auto i = 0; i < (int)scene.textures.size(); i++) {
      auto& t = scene.textures[i];
      fprintf(f, "  {\"width\": %d, \"height\": %d, \"linear\": %d, \"f_fnv\": \"%016llx\", \"b_fnv\": \"%016llx\"}%s\n",
          t.width, t.height, (int)t.linear, (unsigned long long)fnv1a(t.pixelsf),
          (unsigned long long)fnv1a(t.pixelsb), i + 1 < (int)scene.textures.size() ? "," : "");
    }
    fprintf(f, " ],\n \"volumes\": [\n");
    for (auto i = 0; i < (int)scene.volumes.size(); i++) {
      auto& v = scene.volumes[i];
      fprintf(f, "  {\"whd\": [%d, %d, %d], \"res\": %.9g, \"n\": %zu, \"fnv\": \"%016llx\"}%s\n",
          v.whd.x, v.whd.y, v.whd.z, v.res, v.vol.size(), (unsigned long long)fnv1a(v.vol),
          i + 1 < (int)scene.volumes.size() ? "," : "");
    }
    fprintf(f, " ],\n \"lights\": [\n");
    for (auto i = 0; i < (int)lights.lights.size(); i++) {
      auto& l = lights.lights[i];
      fprintf(f, "  {\"instance\": %d, \"environment\": %d, \"sdf\": %d, \"cdf_len\": %zu, \"cdf_back\": %.9g, \"cdf_fnv\": \"%016llx\"}%s\n",
          l.instance, l.environment, l.sdf, l.elements_cdf.size(),
          l.elements_cdf.empty() ? 0.0f : l.elements_cdf.back(),
          (unsigned long long)fnv1a(l.elements_cdf), i + 1 < (int)lights.lights.size() ? "," : "");
    }
    fprintf(f, " ],\n \"state\": {\"width\": %d, \"height\": %d, \"image_fnv\": \"%016llx\", \"rng_fnv\": \"%016llx\"}\n}\n",
        state.width, state.height, (unsigned long long)fnv1a(state.image),
        (unsigned long long)fnv1a(state.rngs));
    fclose(f);
  }

  if (!output.empty()) {
    if (!save_image(output, get_render(state), error)) die(error);
  }
  return 0;
}
