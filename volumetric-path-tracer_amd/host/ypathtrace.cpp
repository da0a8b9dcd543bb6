// ypathtrace — offline renderer CLI with the reference's command line (apps/ypathtrace/ypathtrace.cpp:307-337 over
// libs/yocto/yocto_cli.cpp): --scene --output --shader --samples --resolution --bounces --noparallel --noimplicitmis
// --stmaxiter, each with the reference's default and range, --config <file.json> (yocto_cli.cpp:912-945: a JSON object of
// option values, the command line wins), --help; errors are reported as the reference's parser words them ("unknown
// option X", "missing value for X", "bad value for X") and end the program with status 1.  --interactive is out of scope.
// The run itself is the reference's run_offline sequence (:41-87): load, tesselate, bvh, lights, state, N x
// pathtrace_samples, save.  Rendering happens on the GPU through include/vpt.h; two extensions: --gpus N (tiles dealt
// round-robin over N GPUs, same image), --batch n (samples per kernel launch; default: all in one, same image) and --gpubvh
// (the BVHs built on the GPU by vpt_build_bvh: the same trees), --gputess (the float32 half of tesselate_surfaces on the GPU by
// vpt_subdivide_vertices: the same meshes).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "vpt_host.h"

using namespace vpt;

namespace {

struct option {
  enum kind_t { string_k, int_k, bool_k, shader_k } kind;
  int         lo, hi;   // int_k: inclusive range (lo > hi: unbounded)
  const char* usage;
};
const std::vector<std::pair<string, option>> options = {
    {"scene", {option::string_k, 1, 0, "Scene filename."}},
    {"output", {option::string_k, 1, 0, "Output filename."}},
    {"interactive", {option::bool_k, 1, 0, "Run interactively."}},
    {"resolution", {option::int_k, 1, 4096, "Image resolution."}},
    {"shader", {option::shader_k, 1, 0, "Shader type."}},
    {"samples", {option::int_k, 1, 4096, "Number of samples."}},
    {"bounces", {option::int_k, 1, 128, "Number of bounces."}},
    {"noparallel", {option::bool_k, 1, 0, "Disable threading."}},
    {"noimplicitmis", {option::bool_k, 1, 0, "Disable MIS on implicit shader"}},
    {"stmaxiter", {option::int_k, 1, 512, "Number of maximum iteration while spheretracing"}},
    {"camera", {option::int_k, 0, 1 << 20, "Camera index. (extension)"}},
    {"gpus", {option::int_k, 1, 64, "GPUs to spread the frame's tiles over. (extension)"}},
    {"batch", {option::int_k, 0, 4096, "Samples per kernel launch, 0 = all. (extension)"}},
    {"gpubvh", {option::bool_k, 1, 0, "Build the BVHs on the GPU: same trees. (extension)"}},
    {"gputess", {option::bool_k, 1, 0, "Subdivision vertex arithmetic on the GPU: same meshes. (extension)"}},
};
const option* find_option(const string& name) {
  for (auto& [n, o] : options)
    if (n == name) return &o;
  return nullptr;
}

string usage() {
  auto text = string{"usage: ypathtrace [options]\nRaytrace scenes.\n\noptions:\n"};
  for (auto& [name, o] : options) {
    auto line = "  --" + name + (o.kind == option::bool_k ? "/--no-" + name : o.kind == option::int_k ? " <integer>" : " <string>");
    line.resize(line.size() < 32 ? 32 : line.size() + 1, ' ');
    text += line + o.usage + "\n";
    if (o.kind == option::shader_k) {
      text += "    with choices: ";
      for (auto& s : pathtrace_shader_names) text += s + (&s == &pathtrace_shader_names.back() ? "\n" : ", ");
    }
  }
  text += "  --help                        Prints help. (false)\n  --config <string>             Load configuration. (\"\")\n";
  return text;
}

[[noreturn]] void cli_error(const string& message) {   // handle_errors / print_fatal of the reference: message, usage, status 1
  fprintf(stderr, "error: %s\n%s", message.c_str(), usage().c_str());
  exit(1);
}
[[noreturn]] void print_fatal(const string& message) {
  fprintf(stderr, "error: %s\n", message.c_str());
  exit(1);
}

// the text of one option value, validated against the option's type and range
void check_value(const string& name, const option& o, const string& text) {
  if (o.kind == option::int_k) {
    auto end = (char*)nullptr;
    auto v   = strtol(text.c_str(), &end, 10);
    if (end == text.c_str() || *end != 0) cli_error("bad value for " + name);
    if (o.lo <= o.hi && (v < o.lo || v > o.hi)) cli_error("bad value for " + name);
  } else if (o.kind == option::bool_k) {
    if (text != "true" && text != "false") cli_error("bad value for " + name);
  } else if (o.kind == option::shader_k) {
    auto found = false;
    for (auto& s : pathtrace_shader_names) found = found || s == text;
    if (!found) cli_error("bad value for " + name);
  }
}

}  // namespace

int main(int argc, const char** argv) {
  auto values = std::map<string, string>{};   // option name -> value text, command line first
  auto config = string{};
  for (auto i = 1; i < argc; i++) {
    auto arg = string{argv[i]};
    if (arg == "--help" || arg == "-h") {
      printf("%s", usage().c_str());
      return 0;
    }
    if (arg == "--config") {
      if (i + 1 >= argc) cli_error("missing value for config");
      config = argv[++i];
      continue;
    }
    if (arg.rfind("--", 0) != 0) cli_error("unknown option " + arg);
    auto name = arg.substr(2);
    auto o    = find_option(name);
    if (!o && name.rfind("no-", 0) == 0 && find_option(name.substr(3)) && find_option(name.substr(3))->kind == option::bool_k) {
      values[name.substr(3)] = "false";
      continue;
    }
    if (!o) cli_error("unknown option " + arg);
    if (o->kind == option::bool_k) {
      values[name] = "true";
      continue;
    }
    if (i + 1 >= argc) cli_error("missing value for " + name);
    values[name] = argv[++i];
  }
  if (!config.empty()) {   // the file's values fill in what the command line left open
    auto members = vector<std::pair<string, string>>{};
    auto error   = string{};
    if (!load_cli_config(config, members, error)) print_fatal(error);
    for (auto& [key, text] : members) {
      if (!find_option(key)) cli_error("unknown option " + key);
      if (!values.count(key)) values[key] = text;
    }
  }
  for (auto& [name, text] : values) check_value(name, *find_option(name), text);

  auto params   = pathtrace_params{};
  auto filename = string{"scene.json"}, output = string{"image.png"};
  auto batch = 0, gpus = 1;
  auto get_int = [&](const char* name, int& v) {
    if (values.count(name)) v = atoi(values[name].c_str());
  };
  auto get_bool = [&](const char* name, bool& v) {
    if (values.count(name)) v = values[name] == "true";
  };
  if (values.count("scene")) filename = values["scene"];
  if (values.count("output")) output = values["output"];
  get_int("resolution", params.resolution), get_int("samples", params.samples), get_int("bounces", params.bounces);
  get_int("stmaxiter", params.spheretrace_maxiter), get_int("camera", params.camera), get_int("batch", batch), get_int("gpus", gpus);
  get_bool("noparallel", params.noparallel), get_bool("noimplicitmis", params.noimplicit_mis);
  auto interactive = false, gpubvh = false, gputess = false;
  get_bool("interactive", interactive), get_bool("gpubvh", gpubvh), get_bool("gputess", gputess);
  if (interactive) print_fatal("--interactive is not supported by the GPU build");
  if (values.count("shader"))
    for (size_t k = 0; k < pathtrace_shader_names.size(); k++)
      if (pathtrace_shader_names[k] == values["shader"]) params.shader = (pathtrace_shader_type)k;

  try {
    auto error = string{};
    auto scene = scene_data{};
    if (!load_scene(filename, scene, error)) print_fatal(error);
    if (params.camera >= (int)scene.cameras.size()) cli_error("bad value for camera");
    if (gputess) tesselate_surfaces_device(scene, 0);
    else tesselate_surfaces(scene);
    auto bvh    = gpubvh ? make_bvh_device(scene, params, 0) : make_bvh(scene, params);
    auto lights = make_lights(scene, params);
    auto state  = make_state(scene, params);
    if (gpus > 1) {
      auto devices = vector<int>{};
      for (auto d = 0; d < gpus; d++) devices.push_back(d);
      pathtrace_set_devices(devices);
    }
    auto t0 = std::chrono::steady_clock::now();
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
