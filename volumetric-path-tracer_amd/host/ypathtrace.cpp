// ypathtrace — offline renderer CLI with the reference's command line
// (apps/ypathtrace/ypathtrace.cpp:307-337: --scene --output --shader --samples --resolution
// --bounces --noparallel --noimplicitmis --stmaxiter; --interactive is out of scope) and the
// reference's run_offline sequence (:41-87): load, tesselate, bvh, lights, state, N x
// pathtrace_samples, save.  Rendering happens on the GPU through include/vpt.h.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "vpt_host.h"

using namespace vpt;

static void print_fatal(const string& msg) {
  fprintf(stderr, "error: %s\n", msg.c_str());
  exit(1);
}

int main(int argc, const char** argv) {
  auto params   = pathtrace_params{};
  auto filename = string{"scene.json"}, output = string{"out.png"};
  auto batch    = 0;
  auto usage    = [&]() {
    printf("usage: ypathtrace --scene <scene.json> [--output out.png|jpg] [--shader %s ...]\n"
           "         [--samples 1..4096] [--resolution 1..4096] [--bounces 1..128] [--noparallel]\n"
           "         [--noimplicitmis] [--stmaxiter 1..512] [--camera n] [--batch n]\n",
        pathtrace_shader_names[0].c_str());
  };
  for (auto i = 1; i < argc; i++) {
    auto a    = string{argv[i]};
    auto next = [&]() -> string {
      if (i + 1 >= argc) print_fatal("missing value for " + a);
      return argv[++i];
    };
    auto in_range = [&](int v, int lo, int hi) {
      if (v < lo || v > hi) print_fatal("bad value for " + a);
      return v;
    };
    if (a == "--scene") filename = next();
    else if (a == "--output") output = next();
    else if (a == "--samples") params.samples = in_range(atoi(next().c_str()), 1, 4096);
    else if (a == "--resolution") params.resolution = in_range(atoi(next().c_str()), 1, 4096);
    else if (a == "--bounces") params.bounces = in_range(atoi(next().c_str()), 1, 128);
    else if (a == "--stmaxiter") params.spheretrace_maxiter = in_range(atoi(next().c_str()), 1, 512);
    else if (a == "--camera") params.camera = atoi(next().c_str());
    else if (a == "--batch") batch = atoi(next().c_str());
    else if (a == "--noparallel") params.noparallel = true;
    else if (a == "--no-noparallel") params.noparallel = false;
    else if (a == "--noimplicitmis") params.noimplicit_mis = true;
    else if (a == "--no-noimplicitmis") params.noimplicit_mis = false;
    else if (a == "--interactive") print_fatal("--interactive is not supported by the GPU build");
    else if (a == "--shader") {
      auto name  = next();
      auto found = false;
      for (size_t k = 0; k < pathtrace_shader_names.size(); k++)
        if (pathtrace_shader_names[k] == name) params.shader = (pathtrace_shader_type)k, found = true;
      if (!found) print_fatal("bad value for --shader");
    } else if (a == "--help" || a == "-h") {
      usage();
      return 0;
    } else print_fatal("unknown option " + a);
  }
  try {
    auto error = string{};
    auto scene = scene_data{};
    if (!load_scene(filename, scene, error)) print_fatal(error);
    tesselate_surfaces(scene);
    auto bvh    = make_bvh(scene, params);
    auto lights = make_lights(scene, params);
    auto state  = make_state(scene, params);
    auto t0     = std::chrono::steady_clock::now();
    // one launch per `batch` samples (default: all); identical to that many single calls
    if (batch <= 0) batch = params.samples;
    while (state.samples < params.samples) pathtrace_samples(state, scene, bvh, lights, params, batch);
    auto secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("rendered %dx%d x %d spp in %.3f s (%.2f Msamples/s)\n", state.width, state.height, state.samples, secs,
        (double)state.width * state.height * state.samples / secs * 1e-6);
    if (!save_image(output, get_render(state), error)) print_fatal(error);
  } catch (const std::exception& e) {
    print_fatal(e.what());
  }
  return 0;
}
