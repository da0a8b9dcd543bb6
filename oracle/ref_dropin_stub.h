// ref_dropin_stub.h — THE REFERENCE-SIDE BINDING: the body a maintainer of edu-rinaldi/Volumetric-Path-Tracer puts in place
// of pathtrace_samples() (libs/yocto_pathtrace/yocto_pathtrace.cpp:1052-1092) to run the hot path on libvpt_hip.so.
// It is written against the REFERENCE'S types (scene_data, bvh_scene, pathtrace_lights, pathtrace_state,
// pathtrace_params) and compiles only inside the reference tree; this repo compiles it for real in
// oracle/ref_dropin_pathtrace.cpp (test infrastructure; INTEGRATION.md shows this file verbatim).  Everything the
// application does before and after the call - load_scene, tesselate_surfaces, make_bvh, make_lights, make_state,
// get_render, save_image, the command line - stays the reference's own code.
//
// One thing the reference's containers do not keep: an analytic SDF is a std::function (sdf_data::f, yocto_scene.h:194-200)
// bound by the loader to its type and parameters (yocto_sceneio.cpp:3684-3730), which cannot be read back.  A scene that
// has "sdfunctions" therefore needs its file name here - one added line in the application, after load_scene:
//     yocto::vpt_dropin_scene_file = filename;
// and the stub reads type and parameters from that JSON again (the reference's own json.hpp).  Scenes without analytic
// SDFs need nothing.
#pragma once
#include <vpt.h>

#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "ext/json.hpp"   // libs/yocto/ext/json.hpp, the JSON library the reference's loader uses

namespace yocto {

inline std::string vpt_dropin_scene_file;   // see above

namespace vpt_dropin {

struct flat_scene {   // backing store of one flattened (scene_data, bvh_scene, pathtrace_lights) triple
  vpt_scene_desc                   d = {};
  std::vector<vpt_camera>          cameras;
  std::vector<vpt_instance>        instances;
  std::vector<vpt_shape>           shapes;
  std::vector<vpt_material>        materials;
  std::vector<vpt_texture>         textures;
  std::vector<vpt_environment>     environments;
  std::vector<vpt_volume>          volumes;
  std::vector<vpt_volume_instance> vol_instances;
  std::vector<vpt_sdf>             sdfs;
  std::vector<vpt_light>           lights;
  std::vector<vec3f>               positions, normals;
  std::vector<vec2f>               texcoords;
  std::vector<vec4f>               colors, texels_f;
  std::vector<vec3i>               triangles;
  std::vector<vec4i>               quads;
  std::vector<vec4b>               texels_b;
  std::vector<float>               voxels, cdf;
  std::vector<bvh_node>            shape_nodes;
  std::vector<int>                 shape_prims;
};

template <typename A, typename B>
inline void put(A& dst, const B& src) {
  static_assert(sizeof(A) == sizeof(B), "layouts must agree");
  std::memcpy(&dst, &src, sizeof(A));
}
template <typename T>
inline void append(std::vector<T>& dst, const std::vector<T>& src) {
  dst.insert(dst.end(), src.begin(), src.end());
}

// type tag and parameters of the analytic SDFs, read back from the scene file (see the header comment)
inline void sdf_parameters(std::vector<vpt_sdf>& sdfs) {
  if (sdfs.empty()) return;
  if (vpt_dropin_scene_file.empty())
    throw std::runtime_error("the scene has sdfunctions: set yocto::vpt_dropin_scene_file to the scene's file name after load_scene");
  auto stream = std::ifstream(vpt_dropin_scene_file);
  auto js     = nlohmann::json::parse(stream, nullptr, false);
  if (js.is_discarded() || !js.contains("sdfunctions") || js["sdfunctions"].size() != sdfs.size())
    throw std::runtime_error("cannot read the sdfunctions of " + vpt_dropin_scene_file);
  static const auto names = std::map<std::string, int>{{"bbox", VPT_SDF_BBOX}, {"box", VPT_SDF_BOX},
      {"capped_cone", VPT_SDF_CAPPED_CONE}, {"plane", VPT_SDF_PLANE}, {"sphere", VPT_SDF_SPHERE}, {"torus", VPT_SDF_TORUS}};
  auto idx = 0;
  for (auto& element : js["sdfunctions"]) {
    auto& o    = sdfs[idx++];
    auto  type = element.value("type", std::string{});
    if (!names.count(type)) throw std::runtime_error("unknown sdfunction type " + type);
    o.type  = names.at(type);
    auto num = [&](const char* key) { return element.value(key, 0.0f); };
    switch (o.type) {   // the lambdas of yocto_sceneio.cpp:3684-3730
      case VPT_SDF_BBOX: {
        auto whd = element.value("whd", std::vector<float>{0, 0, 0});
        o.p[0] = num("thickness"), o.p[1] = whd.at(0), o.p[2] = whd.at(1), o.p[3] = whd.at(2);
      } break;
      case VPT_SDF_CAPPED_CONE: o.p[0] = num("height"), o.p[1] = num("r1"), o.p[2] = num("r2"); break;
      case VPT_SDF_SPHERE: o.p[0] = num("radius"); break;
      case VPT_SDF_TORUS: o.p[0] = num("r1"), o.p[1] = num("r2"); break;
      default: break;   // box: whd is in sdf_data; plane: no parameter
    }
  }
}

inline vpt_scene* upload(const scene_data& scene, const bvh_scene& bvh, const pathtrace_lights& lights) {
  auto f = std::make_unique<flat_scene>();
  if (bvh.shapes.size() != scene.shapes.size()) throw std::invalid_argument("bvh does not belong to this scene");
  for (auto& c : scene.cameras) {
    auto& o = f->cameras.emplace_back();
    put(o.frame, c.frame);
    o.orthographic = c.orthographic, o.lens = c.lens, o.film = c.film, o.aspect = c.aspect, o.focus = c.focus, o.aperture = c.aperture;
  }
  for (auto i = 0; i < (int)scene.shapes.size(); i++) {
    auto& s = scene.shapes[i];
    auto& b = bvh.shapes[i];
    if (!s.points.empty() || !s.lines.empty()) throw std::invalid_argument("point / line shapes are outside the HIP path");
    auto& o           = f->shapes.emplace_back();
    o.num_vertices    = (int)s.positions.size(), o.position_offset = (int)f->positions.size();
    o.normal_offset   = s.normals.empty() ? -1 : (int)f->normals.size();
    o.texcoord_offset = s.texcoords.empty() ? -1 : (int)f->texcoords.size();
    o.color_offset    = s.colors.empty() ? -1 : (int)f->colors.size();
    o.num_triangles = (int)s.triangles.size(), o.triangle_offset = (int)f->triangles.size();
    o.num_quads = s.triangles.empty() ? (int)s.quads.size() : 0, o.quad_offset = (int)f->quads.size();   // triangles win (yocto_bvh.cpp:770-789)
    o.num_bvh_nodes = (int)b.nodes.size(), o.bvh_node_offset = (int)f->shape_nodes.size(), o.bvh_prim_offset = (int)f->shape_prims.size();
    append(f->positions, s.positions), append(f->normals, s.normals), append(f->texcoords, s.texcoords), append(f->colors, s.colors);
    append(f->triangles, s.triangles);
    if (o.num_quads) append(f->quads, s.quads);
    append(f->shape_nodes, b.nodes), append(f->shape_prims, b.primitives);
  }
  for (auto& i : scene.instances) {
    auto& o = f->instances.emplace_back();
    put(o.frame, i.frame);
    o.shape = i.shape, o.material = i.material;
  }
  for (auto& m : scene.materials) {   // material_data -> vpt_material: same fields, same order
    auto& o = f->materials.emplace_back();
    o.type  = (int)m.type;
    put(o.emission, m.emission), put(o.color, m.color), put(o.scattering, m.scattering);
    o.roughness = m.roughness, o.metallic = m.metallic, o.ior = m.ior;
    o.scanisotropy = m.scanisotropy, o.trdepth = m.trdepth, o.opacity = m.opacity;
    o.emission_tex = m.emission_tex, o.color_tex = m.color_tex, o.roughness_tex = m.roughness_tex;
    o.scattering_tex = m.scattering_tex, o.normal_tex = m.normal_tex;
  }
  for (auto& t : scene.textures) {
    auto is_float = !t.pixelsf.empty();
    f->textures.push_back({t.width, t.height, t.linear, is_float, (int64_t)(is_float ? f->texels_f.size() : f->texels_b.size())});
    if (is_float) append(f->texels_f, t.pixelsf);
    else append(f->texels_b, t.pixelsb);
  }
  for (auto& e : scene.environments) {
    auto& o = f->environments.emplace_back();
    put(o.frame, e.frame), put(o.emission, e.emission);
    o.emission_tex = e.emission_tex;
  }
  for (auto& v : scene.volumes) {
    f->volumes.push_back({{v.whd.x, v.whd.y, v.whd.z}, v.res, (int64_t)f->voxels.size()});
    append(f->voxels, v.vol);
  }
  for (auto& i : scene.vol_instances) {
    auto& o = f->vol_instances.emplace_back();
    put(o.frame, i.frame);
    o.volume = i.volume, o.material = i.material, o.scalef = i.scalef;
  }
  for (auto& s : scene.sdfs) {
    auto& o = f->sdfs.emplace_back();
    o       = {};
    put(o.frame, s.frame), put(o.whd, s.whd);
    o.material = s.material;
  }
  sdf_parameters(f->sdfs);
  for (auto& l : lights.lights) {
    f->lights.push_back({l.instance, l.environment, l.sdf, (int)l.elements_cdf.size(), (int64_t)f->cdf.size()});
    append(f->cdf, l.elements_cdf);
  }
  auto& d = f->d;
  d.num_cameras = (int)f->cameras.size(), d.cameras = f->cameras.data();
  d.num_instances = (int)f->instances.size(), d.instances = f->instances.data();
  d.num_shapes = (int)f->shapes.size(), d.shapes = f->shapes.data();
  d.num_materials = (int)f->materials.size(), d.materials = f->materials.data();
  d.num_textures = (int)f->textures.size(), d.textures = f->textures.data();
  d.num_environments = (int)f->environments.size(), d.environments = f->environments.data();
  d.num_volumes = (int)f->volumes.size(), d.volumes = f->volumes.data();
  d.num_vol_instances = (int)f->vol_instances.size(), d.vol_instances = f->vol_instances.data();
  d.num_sdfs = (int)f->sdfs.size(), d.sdfs = f->sdfs.data();
  d.num_lights = (int)f->lights.size(), d.lights = f->lights.data();
  d.num_positions = (int64_t)f->positions.size(), d.positions = (const float*)f->positions.data();
  d.num_normals = (int64_t)f->normals.size(), d.normals = (const float*)f->normals.data();
  d.num_texcoords = (int64_t)f->texcoords.size(), d.texcoords = (const float*)f->texcoords.data();
  d.num_colors = (int64_t)f->colors.size(), d.colors = (const float*)f->colors.data();
  d.num_triangles = (int64_t)f->triangles.size(), d.triangles = (const int32_t*)f->triangles.data();
  d.num_quads = (int64_t)f->quads.size(), d.quads = (const int32_t*)f->quads.data();
  d.num_texels_f = (int64_t)f->texels_f.size(), d.texels_f = (const float*)f->texels_f.data();
  d.num_texels_b = (int64_t)f->texels_b.size(), d.texels_b = (const uint8_t*)f->texels_b.data();
  d.num_voxels = (int64_t)f->voxels.size(), d.voxels = f->voxels.data();
  d.num_light_cdf = (int64_t)f->cdf.size(), d.light_cdf = f->cdf.data();
  static_assert(sizeof(bvh_node) == sizeof(vpt_bvh_node), "bvh_node is the same 32-byte record");
  d.num_scene_bvh_nodes = (int)bvh.nodes.size(), d.scene_bvh_nodes = (const vpt_bvh_node*)bvh.nodes.data();
  d.num_scene_bvh_prims = (int)bvh.primitives.size(), d.scene_bvh_prims = bvh.primitives.data();
  d.num_shape_bvh_nodes = (int64_t)f->shape_nodes.size(), d.shape_bvh_nodes = (const vpt_bvh_node*)f->shape_nodes.data();
  d.num_shape_bvh_prims = (int64_t)f->shape_prims.size(), d.shape_bvh_prims = f->shape_prims.data();
  vpt_scene* out = nullptr;
  if (vpt_scene_create(&d, 0, &out) != VPT_OK) throw std::runtime_error(std::string("vpt_scene_create: ") + vpt_last_error());
  return out;   // the library copied everything to the device: the flattened arrays can go
}

struct device_scenes {   // one upload per scene object, released at program end
  std::map<const scene_data*, vpt_scene*> handles;
  ~device_scenes() {
    for (auto& [scene, handle] : handles) vpt_scene_destroy(handle);
  }
};
inline device_scenes& cache() {
  static auto scenes = device_scenes{};
  return scenes;
}

}  // namespace vpt_dropin

// the drop-in body of pathtrace_samples (yocto_pathtrace.cpp:1052-1092)
void pathtrace_samples(pathtrace_state& state, const scene_data& scene, const bvh_scene& bvh,
    const pathtrace_lights& lights, const pathtrace_params& params) {
  if (state.samples >= params.samples) return;
  auto& device = vpt_dropin::cache().handles[&scene];
  if (!device) device = vpt_dropin::upload(scene, bvh, lights);
  auto p = vpt_params{params.camera, params.resolution, (int)params.shader, params.samples, params.bounces, params.noparallel,
      params.noimplicit_mis, params.spheretrace_maxiter};
  static_assert(sizeof(rng_state) == 16 && sizeof(vec4f) == 16, "pathtrace_state arrays are passed as they are");
  auto rc = vpt_render(device, &p, 1, state.width, state.height, (float*)state.image.data(), state.hits.data(),
      (uint64_t*)state.rngs.data(), &state.samples);
  if (rc == VPT_ERR_UNKNOWN_SHADER) throw std::runtime_error("sampler unknown");   // as get_shader, yocto_pathtrace.cpp:947-950
  if (rc != VPT_OK) throw std::runtime_error(std::string("vpt_render: ") + vpt_last_error());
}

}  // namespace yocto
