// vpt_pathtrace.cpp — load-time builders + the drop-in pathtrace_samples() that forwards
// to the HIP path through the C-ABI (include/vpt.h).  No rendering arithmetic lives here.
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

#include "vpt_host.h"
#include "vpt_hostmath.h"

namespace vpt {

// =============================================================================================
// make_state — per-pixel PCG32 seeding, yocto_pathtrace.cpp:960-980, yocto_sampling.h:184-205
// =============================================================================================
static uint32_t pcg32_next(rng_state& rng) {
  auto old     = rng.state;
  rng.state    = old * 6364136223846793005ULL + rng.inc;
  auto xs      = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  auto rot     = (uint32_t)(old >> 59u);
  return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
static rng_state pcg32_make(uint64_t seed, uint64_t seq) {
  auto rng  = rng_state{};
  rng.state = 0;
  rng.inc   = (seq << 1u) | 1u;
  pcg32_next(rng);
  rng.state += seed;
  pcg32_next(rng);
  return rng;
}

pathtrace_state make_state(const scene_data& scene, const pathtrace_params& params) {
  if (params.camera < 0 || params.camera >= (int)scene.cameras.size())
    throw std::out_of_range{"camera index out of range"};
  auto& camera = scene.cameras[params.camera];
  auto  state  = pathtrace_state{};
  if (camera.aspect >= 1) {
    state.width  = params.resolution;
    state.height = (int)std::round(params.resolution / camera.aspect);
  } else {
    state.height = params.resolution;
    state.width  = (int)std::round(params.resolution * camera.aspect);
  }
  auto npixels = (size_t)state.width * state.height;
  state.image.assign(npixels, {0, 0, 0, 0});
  state.hits.assign(npixels, 0);
  state.rngs.resize(npixels);
  // master stream: make_rng(1301081) uses the default sequence 1 (yocto_sampling.h:146)
  auto master = pcg32_make(1301081, 1);
  for (auto& rng : state.rngs) {
    // rand1i(rng_, 1 << 31): the int argument wraps to 2147483648u in the unsigned modulo
    auto seq = (int)(pcg32_next(master) % 2147483648u) / 2 + 1;
    rng      = pcg32_make(961748941ull, (uint64_t)seq);
  }
  return state;
}

// =============================================================================================
// make_bvh — middle split, <=4 prims per leaf.  yocto_bvh.cpp:411-441, 447-507, 521-611
// =============================================================================================
static void build_nodes(bvh_data& bvh, const vector<bbox3f>& bboxes) {
  const int max_prims = 4;  // yocto_bvh.cpp:444
  auto      n         = (int)bboxes.size();
  bvh.nodes.clear();
  bvh.nodes.reserve((size_t)n * 2);
  bvh.primitives.resize(n);
  for (auto i = 0; i < n; i++) bvh.primitives[i] = i;
  auto centers = vector<vec3f>(n);
  for (auto i = 0; i < n; i++) centers[i] = center(bboxes[i]);

  struct work { int node, start, end; };
  auto todo = vector<work>{{0, 0, n}};
  bvh.nodes.emplace_back();
  auto* prims = bvh.primitives.data();
  while (!todo.empty()) {
    auto [nodeid, start, end] = todo.back();
    todo.pop_back();
    auto box = bbox3f{};
    for (auto i = start; i < end; i++) box = merge(box, bboxes[prims[i]]);
    auto& node       = bvh.nodes[nodeid];
    node.bbox_min[0] = box.min.x, node.bbox_min[1] = box.min.y, node.bbox_min[2] = box.min.z;
    node.bbox_max[0] = box.max.x, node.bbox_max[1] = box.max.y, node.bbox_max[2] = box.max.z;
    if (end - start <= max_prims) {
      node.internal = 0, node.axis = 0;
      node.num   = (int16_t)(end - start);
      node.start = start;
      continue;
    }
    // split_middle: partition at the centre of the centroid box along its largest axis
    auto cbox = bbox3f{};
    for (auto i = start; i < end; i++) cbox = merge(cbox, centers[prims[i]]);
    auto csize = cbox.max - cbox.min;
    auto mid = (start + end) / 2, axis = 0;
    if (!(csize == vec3f{0, 0, 0})) {
      if (csize.x >= csize.y && csize.x >= csize.z) axis = 0;
      if (csize.y >= csize.x && csize.y >= csize.z) axis = 1;
      if (csize.z >= csize.x && csize.z >= csize.y) axis = 2;
      auto split = comp(center(cbox), axis);
      mid = (int)(std::partition(prims + start, prims + end,
                      [&](int prim) { return comp(centers[prim], axis) < split; }) -
                  prims);
      if (mid == start || mid == end) mid = (start + end) / 2;
    }
    node.internal = 1;
    node.axis     = (int8_t)axis;
    node.num      = 2;
    node.start    = (int)bvh.nodes.size();
    auto first    = node.start;  // `node` dangles after emplace_back only if capacity grew: it cannot
    bvh.nodes.emplace_back();
    bvh.nodes.emplace_back();
    todo.push_back({first + 0, start, mid});
    todo.push_back({first + 1, mid, end});
  }
  bvh.nodes.shrink_to_fit();
}

// the nodes of one tree: on the host (build_nodes) or on a GPU (vpt_build_bvh: the same arrays)
static void build_nodes_on(int device, bvh_data& bvh, const vector<bbox3f>& bboxes) {
  if (device < 0) return build_nodes(bvh, bboxes);
  static_assert(sizeof(bbox3f) == 6 * sizeof(float), "bbox layout");
  auto n = (int)bboxes.size(), count = 0;
  bvh.nodes.assign((size_t)std::max(1, 2 * n), bvh_node{});
  bvh.primitives.assign((size_t)n, 0);
  if (vpt_build_bvh(device, (const float*)bboxes.data(), n, bvh.nodes.data(), (int)bvh.nodes.size(), &count, bvh.primitives.data()) != VPT_OK)
    throw std::runtime_error{string{"make_bvh_device: "} + vpt_last_error()};
  bvh.nodes.resize((size_t)count);
  bvh.nodes.shrink_to_fit();
}
bvh_data build_bvh_host(const float* bboxes, int n) {
  auto boxes = vector<bbox3f>((size_t)n);
  if (n > 0) memcpy((void*)boxes.data(), bboxes, (size_t)n * sizeof(bbox3f));
  auto bvh = bvh_data{};
  build_nodes(bvh, boxes);
  return bvh;
}

static bvh_data make_shape_bvh(const shape_data& shape, int device) {
  auto bvh    = bvh_data{};
  auto bboxes = vector<bbox3f>{};
  if (!shape.triangles.empty()) {
    bboxes.resize(shape.triangles.size());
    for (size_t i = 0; i < bboxes.size(); i++) {
      auto& t = shape.triangles[i];
      auto &p0 = shape.positions[t.x], &p1 = shape.positions[t.y], &p2 = shape.positions[t.z];
      bboxes[i] = {vmin(p0, vmin(p1, p2)), vmax(p0, vmax(p1, p2))};
    }
  } else if (!shape.quads.empty()) {
    bboxes.resize(shape.quads.size());
    for (size_t i = 0; i < bboxes.size(); i++) {
      auto& q = shape.quads[i];
      auto &p0 = shape.positions[q.x], &p1 = shape.positions[q.y], &p2 = shape.positions[q.z],
           &p3 = shape.positions[q.w];
      bboxes[i] = {vmin(p0, vmin(p1, vmin(p2, p3))), vmax(p0, vmax(p1, vmax(p2, p3)))};
    }
  }
  build_nodes_on(device, bvh, bboxes);
  return bvh;
}

static bvh_scene make_bvh_on(int device, const scene_data& scene) {
  auto bvh = bvh_data{};
  bvh.shapes.resize(scene.shapes.size());
  for (size_t i = 0; i < scene.shapes.size(); i++) bvh.shapes[i] = make_shape_bvh(scene.shapes[i], device);
  auto bboxes = vector<bbox3f>(scene.instances.size());
  for (size_t i = 0; i < bboxes.size(); i++) {
    auto& instance = scene.instances[i];
    auto& sbvh     = bvh.shapes.at(instance.shape);
    if (sbvh.nodes.empty()) continue;  // invalidb3f
    auto& root = sbvh.nodes[0];
    auto  box  = bbox3f{};  // transform_bbox, yocto_geometry.h:441-452: 8 corners, z fastest
    for (auto cx = 0; cx < 2; cx++)
      for (auto cy = 0; cy < 2; cy++)
        for (auto cz = 0; cz < 2; cz++) {
          auto corner = vec3f{cx ? root.bbox_max[0] : root.bbox_min[0],
              cy ? root.bbox_max[1] : root.bbox_min[1], cz ? root.bbox_max[2] : root.bbox_min[2]};
          box = merge(box, transform_point(instance.frame, corner));
        }
    bboxes[i] = box;
  }
  build_nodes_on(device, bvh, bboxes);
  return bvh;
}
bvh_scene make_bvh(const scene_data& scene, const pathtrace_params&) { return make_bvh_on(-1, scene); }
bvh_scene make_bvh_device(const scene_data& scene, const pathtrace_params&, int device) {
  if (device < 0) throw std::invalid_argument{"make_bvh_device: negative device"};
  return make_bvh_on(device, scene);
}

// tesselate_surfaces: host/vpt_tesselate.cpp

// =============================================================================================
// make_lights — serial float32 running sums.  yocto_pathtrace.cpp:983-1049
// =============================================================================================
pathtrace_lights make_lights(const scene_data& scene, const pathtrace_params&) {
  auto lights = pathtrace_lights{};
  for (auto handle = 0; handle < (int)scene.instances.size(); handle++) {
    auto& instance = scene.instances[handle];
    auto& material = scene.materials.at(instance.material);
    if (material.emission == vec3f{0, 0, 0}) continue;
    auto& shape = scene.shapes.at(instance.shape);
    if (shape.triangles.empty() && shape.quads.empty()) continue;
    auto& light    = lights.lights.emplace_back();
    light.instance = handle;
    auto& cdf      = light.elements_cdf;
    auto& pos      = shape.positions;
    if (!shape.triangles.empty()) {
      cdf.resize(shape.triangles.size());
      for (size_t i = 0; i < cdf.size(); i++) {
        auto& t = shape.triangles[i];
        cdf[i]  = triangle_area(pos[t.x], pos[t.y], pos[t.z]);
        if (i != 0) cdf[i] += cdf[i - 1];
      }
    }
    if (!shape.quads.empty()) {
      cdf.resize(shape.quads.size());
      for (size_t i = 0; i < cdf.size(); i++) {
        auto& q = shape.quads[i];
        cdf[i]  = quad_area(pos[q.x], pos[q.y], pos[q.z], pos[q.w]);
        if (i != 0) cdf[i] += cdf[i - 1];
      }
    }
  }
  for (auto handle = 0; handle < (int)scene.environments.size(); handle++) {
    auto& environment = scene.environments[handle];
    if (environment.emission == vec3f{0, 0, 0}) continue;
    auto& light       = lights.lights.emplace_back();
    light.environment = handle;
    if (environment.emission_tex == invalidid) continue;
    auto& texture = scene.textures.at(environment.emission_tex);
    auto& cdf     = light.elements_cdf;
    cdf.resize((size_t)texture.width * texture.height);
    for (size_t idx = 0; idx < cdf.size(); idx++) {
      auto i = (int)(idx % texture.width), j = (int)(idx / texture.width);
      auto th = (j + 0.5f) * pif / texture.height;
      // lookup_texture(texture, i, j) without as_linear, then max over ALL FOUR channels
      // (yocto_math.h:1824 — alpha = 1 participates; kept on purpose)
      auto value = vec4f{};
      if (!texture.pixelsf.empty()) {
        value = texture.pixelsf[(size_t)j * texture.width + i];
      } else {
        auto b = texture.pixelsb[(size_t)j * texture.width + i];
        value  = {b.x / 255.0f, b.y / 255.0f, b.z / 255.0f, b.w / 255.0f};
      }
      auto m   = fmax_(fmax_(fmax_(value.x, value.y), value.z), value.w);
      cdf[idx] = m * std::sin(th);
      if (idx != 0) cdf[idx] += cdf[idx - 1];
    }
  }
  for (auto handle = 0; handle < (int)scene.sdfs.size(); handle++) {
    auto& sdf      = scene.sdfs[handle];
    auto& material = scene.materials.at(sdf.material);
    if (material.emission == vec3f{0, 0, 0}) continue;
    auto& light        = lights.lights.emplace_back();
    light.sdf          = handle;
    light.elements_cdf = {sdf.whd.x * sdf.whd.y};
  }
  return lights;
}

// =============================================================================================
// flatten to the C-ABI
// =============================================================================================
static vpt_frame to_abi(const frame3f& f) {
  auto r = vpt_frame{};
  std::memcpy(&r, &f, sizeof(r));
  return r;
}

vpt_params to_abi(const pathtrace_params& p) {
  auto r                = vpt_params{};
  r.camera              = p.camera;
  r.resolution          = p.resolution;
  r.shader              = (int)p.shader;
  r.samples             = p.samples;
  r.bounces             = p.bounces;
  r.noparallel          = p.noparallel;
  r.noimplicit_mis      = p.noimplicit_mis;
  r.spheretrace_maxiter = p.spheretrace_maxiter;
  return r;
}

void flatten_scene(flat_scene& flat, const scene_data& scene, const bvh_scene& bvh,
    const pathtrace_lights& lights) {
  if (bvh.shapes.size() != scene.shapes.size())
    throw std::invalid_argument{"bvh does not belong to this scene"};
  for (auto& c : scene.cameras) {
    auto& d = flat.cameras.emplace_back();
    d.frame = to_abi(c.frame), d.orthographic = c.orthographic, d.lens = c.lens, d.film = c.film;
    d.aspect = c.aspect, d.focus = c.focus, d.aperture = c.aperture;
  }
  for (size_t s = 0; s < scene.shapes.size(); s++) {
    auto& shape = scene.shapes[s];
    if (!shape.points.empty())
      throw std::invalid_argument{"point/line shapes are outside the hot-path scope"};
    auto& d           = flat.shapes.emplace_back();
    d.num_vertices    = (int)shape.positions.size();
    d.position_offset = (int)flat.positions.size();
    flat.positions.insert(flat.positions.end(), shape.positions.begin(), shape.positions.end());
    d.normal_offset = shape.normals.empty() ? -1 : (int)flat.normals.size();
    flat.normals.insert(flat.normals.end(), shape.normals.begin(), shape.normals.end());
    d.texcoord_offset = shape.texcoords.empty() ? -1 : (int)flat.texcoords.size();
    flat.texcoords.insert(flat.texcoords.end(), shape.texcoords.begin(), shape.texcoords.end());
    d.color_offset = shape.colors.empty() ? -1 : (int)flat.colors.size();
    flat.colors.insert(flat.colors.end(), shape.colors.begin(), shape.colors.end());
    d.num_triangles = (int)shape.triangles.size(), d.triangle_offset = (int)flat.triangles.size();
    flat.triangles.insert(flat.triangles.end(), shape.triangles.begin(), shape.triangles.end());
    // the reference tests triangles first (yocto_bvh.cpp:770-789): a shape with both keeps only them
    d.num_quads = shape.triangles.empty() ? (int)shape.quads.size() : 0;
    d.quad_offset = (int)flat.quads.size();
    if (d.num_quads) flat.quads.insert(flat.quads.end(), shape.quads.begin(), shape.quads.end());
    auto& sb          = bvh.shapes[s];
    d.num_bvh_nodes   = (int)sb.nodes.size();
    d.bvh_node_offset = (int)flat.shape_nodes.size();
    d.bvh_prim_offset = (int)flat.shape_prims.size();
    flat.shape_nodes.insert(flat.shape_nodes.end(), sb.nodes.begin(), sb.nodes.end());
    flat.shape_prims.insert(flat.shape_prims.end(), sb.primitives.begin(), sb.primitives.end());
  }
  for (auto& i : scene.instances) {
    auto& d = flat.instances.emplace_back();
    d.frame = to_abi(i.frame), d.shape = i.shape, d.material = i.material;
  }
  for (auto& m : scene.materials) {
    auto& d = flat.materials.emplace_back();
    d.type  = (int)m.type;
    std::memcpy(d.emission, &m.emission, 12), std::memcpy(d.color, &m.color, 12);
    d.roughness = m.roughness, d.metallic = m.metallic, d.ior = m.ior;
    std::memcpy(d.scattering, &m.scattering, 12);
    d.scanisotropy = m.scanisotropy, d.trdepth = m.trdepth, d.opacity = m.opacity;
    d.emission_tex = m.emission_tex, d.color_tex = m.color_tex, d.roughness_tex = m.roughness_tex;
    d.scattering_tex = m.scattering_tex, d.normal_tex = m.normal_tex;
  }
  for (auto& t : scene.textures) {
    auto& d = flat.textures.emplace_back();
    d.width = t.width, d.height = t.height, d.linear = t.linear;
    d.is_float = !t.pixelsf.empty();
    if (d.is_float) {
      d.offset = (int64_t)flat.texels_f.size();
      flat.texels_f.insert(flat.texels_f.end(), t.pixelsf.begin(), t.pixelsf.end());
    } else {
      d.offset = (int64_t)flat.texels_b.size();
      flat.texels_b.insert(flat.texels_b.end(), t.pixelsb.begin(), t.pixelsb.end());
    }
  }
  for (auto& e : scene.environments) {
    auto& d = flat.environments.emplace_back();
    d.frame = to_abi(e.frame), d.emission_tex = e.emission_tex;
    std::memcpy(d.emission, &e.emission, 12);
  }
  for (auto& v : scene.volumes) {
    auto& d  = flat.volumes.emplace_back();
    d.whd[0] = v.whd.x, d.whd[1] = v.whd.y, d.whd[2] = v.whd.z, d.res = v.res;
    d.offset = (int64_t)flat.voxels.size();
    flat.voxels.insert(flat.voxels.end(), v.vol.begin(), v.vol.end());
  }
  for (auto& i : scene.vol_instances) {
    auto& d = flat.vol_instances.emplace_back();
    d.frame = to_abi(i.frame), d.volume = i.volume, d.material = i.material, d.scalef = i.scalef;
  }
  for (auto& s : scene.sdfs) {
    auto& d = flat.sdfs.emplace_back();
    d.frame = to_abi(s.frame), d.type = (int)s.type, d.material = s.material;
    std::memcpy(d.whd, &s.whd, 12), std::memcpy(d.p, s.p, 16);
  }
  for (auto& l : lights.lights) {
    auto& d = flat.lights.emplace_back();
    d.instance = l.instance, d.environment = l.environment, d.sdf = l.sdf;
    d.cdf_len = (int)l.elements_cdf.size(), d.cdf_offset = (int64_t)flat.light_cdf.size();
    flat.light_cdf.insert(flat.light_cdf.end(), l.elements_cdf.begin(), l.elements_cdf.end());
  }
  flat.scene_nodes = bvh.nodes, flat.scene_prims = bvh.primitives;

  auto& d = flat.desc;
  d       = {};
#define VPT_SET(count, ptr, vec) d.count = (decltype(d.count))flat.vec.size(), d.ptr = flat.vec.empty() ? nullptr : (decltype(d.ptr))flat.vec.data()
  VPT_SET(num_cameras, cameras, cameras);
  VPT_SET(num_instances, instances, instances);
  VPT_SET(num_shapes, shapes, shapes);
  VPT_SET(num_materials, materials, materials);
  VPT_SET(num_textures, textures, textures);
  VPT_SET(num_environments, environments, environments);
  VPT_SET(num_volumes, volumes, volumes);
  VPT_SET(num_vol_instances, vol_instances, vol_instances);
  VPT_SET(num_sdfs, sdfs, sdfs);
  VPT_SET(num_lights, lights, lights);
  VPT_SET(num_positions, positions, positions);
  VPT_SET(num_normals, normals, normals);
  VPT_SET(num_texcoords, texcoords, texcoords);
  VPT_SET(num_colors, colors, colors);
  VPT_SET(num_triangles, triangles, triangles);
  VPT_SET(num_quads, quads, quads);
  VPT_SET(num_texels_f, texels_f, texels_f);
  VPT_SET(num_texels_b, texels_b, texels_b);
  VPT_SET(num_voxels, voxels, voxels);
  VPT_SET(num_light_cdf, light_cdf, light_cdf);
  VPT_SET(num_scene_bvh_nodes, scene_bvh_nodes, scene_nodes);
  VPT_SET(num_scene_bvh_prims, scene_bvh_prims, scene_prims);
  VPT_SET(num_shape_bvh_nodes, shape_bvh_nodes, shape_nodes);
  VPT_SET(num_shape_bvh_prims, shape_bvh_prims, shape_prims);
#undef VPT_SET
}

// =============================================================================================
// pathtrace_samples — the drop-in.  Device copies of scenes are cached per scene_data address + content fingerprint.
// =============================================================================================
namespace {
// The reference's API has no release hook and keys nothing: a caller may mutate the scene, rebuild bvh / lights or
// allocate another scene at the same address between two calls.  The cached device copy therefore carries a cheap
// fingerprint of what it was made from (addresses and sizes of the containers the flattening reads, a hash over the
// heads of the big arrays); a mismatch rebuilds it.
struct device_entry {
  vpt_multi* handle = nullptr;
  uint64_t   fingerprint = 0;
  ~device_entry() {
    if (handle) vpt_multi_destroy(handle);
  }
};
std::mutex                                                   cache_mutex;
std::map<const scene_data*, std::unique_ptr<device_entry>>& device_cache() {
  static auto cache = std::map<const scene_data*, std::unique_ptr<device_entry>>{};
  return cache;
}
vector<int>& device_list() {
  static auto devices = vector<int>{0};
  return devices;
}
uint64_t mix(uint64_t h, uint64_t v) { return (h ^ v) * 0x100000001b3ull; }
template <typename T>
uint64_t mix_array(uint64_t h, const vector<T>& v) {   // bulk arrays: address, size, head and tail
  h = mix(mix(h, (uint64_t)(uintptr_t)v.data()), v.size());
  auto bytes = (const unsigned char*)v.data();
  auto n     = v.size() * sizeof(T);
  for (size_t i = 0; i < n && i < 256; i++) h = mix(h, bytes[i]);          // head
  for (size_t i = n > 256 ? n - 256 : n; i < n; i++) h = mix(h, bytes[i]);  // tail
  return h;
}
template <typename T>
uint64_t mix_table(uint64_t h, const vector<T>& v) {   // small tables (a few KB): every byte, eight at a time
  h = mix(mix(h, (uint64_t)(uintptr_t)v.data()), v.size());
  auto bytes = (const unsigned char*)v.data();
  auto n     = v.size() * sizeof(T);
  auto i     = (size_t)0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, bytes + i, 8);
    h = mix(h, w);
  }
  for (; i < n; i++) h = mix(h, bytes[i]);
  return h;
}
// What the cached device copy was made from.  The small tables - cameras, instances, materials, environments, volume
// instances, SDFs, light ids - are hashed in full, so an in-place edit of any of them (a moved instance, a changed material or
// camera) rebuilds the copy on the next call, as the reference - which reads the live scene - would show it.  The bulk arrays
// (vertex data, texels, voxels, BVH nodes, light CDFs) are sampled at head and tail only: editing them in place between two calls
// needs pathtrace_release(scene) (vpt_host.h).
uint64_t scene_fingerprint(const scene_data& scene, const bvh_scene& bvh, const pathtrace_lights& lights) {
  auto h = 0xcbf29ce484222325ull;
  h = mix_table(h, scene.cameras), h = mix_table(h, scene.instances), h = mix_table(h, scene.materials);
  h = mix_table(h, scene.environments), h = mix_table(h, scene.vol_instances), h = mix_table(h, scene.sdfs);
  h = mix(h, scene.shapes.size()), h = mix(h, scene.textures.size()), h = mix(h, scene.volumes.size());
  for (auto& s : scene.shapes) h = mix_array(h, s.positions), h = mix(h, s.triangles.size()), h = mix(h, s.quads.size());
  for (auto& t : scene.textures) h = mix(mix(h, (uint64_t)t.width), (uint64_t)t.height), h = mix_array(h, t.pixelsb), h = mix_array(h, t.pixelsf);
  for (auto& v : scene.volumes) h = mix_array(h, v.vol);
  h = mix_array(h, bvh.nodes), h = mix_array(h, bvh.primitives), h = mix(h, bvh.shapes.size());
  for (auto& b : bvh.shapes) h = mix_array(h, b.nodes);
  h = mix(h, lights.lights.size());
  for (auto& l : lights.lights) h = mix(mix(mix(h, (uint64_t)l.instance), (uint64_t)l.environment), (uint64_t)l.sdf), h = mix_array(h, l.elements_cdf);
  return h;
}
}  // namespace

void pathtrace_release(const scene_data& scene) {
  auto lock = std::lock_guard{cache_mutex};
  device_cache().erase(&scene);
}

void pathtrace_set_devices(const vector<int>& devices) {
  if (devices.empty()) throw std::invalid_argument{"empty device list"};
  auto lock = std::lock_guard{cache_mutex};
  device_list() = devices;
  device_cache().clear();
}

void pathtrace_samples(pathtrace_state& state, const scene_data& scene, const bvh_scene& bvh,
    const pathtrace_lights& lights, const pathtrace_params& params, int count) {
  if (state.samples >= params.samples) return;  // reference cpp:1055
  if ((int)params.shader < 0 || (int)params.shader > (int)pathtrace_shader_type::implicit_normal)
    throw std::runtime_error{"sampler unknown"};  // reference cpp:947-950
  auto handle = (vpt_multi*)nullptr;
  {
    auto  lock  = std::lock_guard{cache_mutex};
    auto& entry = device_cache()[&scene];
    auto  print = scene_fingerprint(scene, bvh, lights);
    if (!entry || entry->fingerprint != print) {
      entry.reset();
      auto flat = flat_scene{};
      flatten_scene(flat, scene, bvh, lights);
      auto fresh = std::make_unique<device_entry>();
      if (vpt_multi_create(&flat.desc, device_list().data(), (int)device_list().size(), &fresh->handle) != VPT_OK) {
        device_cache().erase(&scene);
        throw std::runtime_error{string{"vpt_multi_create: "} + vpt_last_error()};
      }
      fresh->fingerprint = print;
      entry              = std::move(fresh);
    }
    handle = entry->handle;
  }
  auto abi = to_abi(params);
  static_assert(sizeof(rng_state) == 16 && sizeof(vec4f) == 16, "state layout");
  if (vpt_multi_render(handle, &abi, count, state.width, state.height, (float*)state.image.data(),
          state.hits.data(), (uint64_t*)state.rngs.data(), &state.samples) != VPT_OK)
    throw std::runtime_error{string{"vpt_render: "} + vpt_last_error()};
}

void pathtrace_samples(pathtrace_state& state, const scene_data& scene, const bvh_scene& bvh,
    const pathtrace_lights& lights, const pathtrace_params& params) {
  pathtrace_samples(state, scene, bvh, lights, params, 1);
}

// get_render, yocto_pathtrace.cpp:1105-1116 (throws like check_image :1095-1102)
void get_render(color_image& image, const pathtrace_state& state) {
  if (image.width != state.width || image.height != state.height)
    throw std::invalid_argument{"image should have the same size"};
  if (!image.linear) throw std::invalid_argument{"expected linear image"};
  auto scale = 1.0f / (float)state.samples;
  for (size_t i = 0; i < state.image.size(); i++) {
    auto& p         = state.image[i];
    image.pixels[i] = {p.x * scale, p.y * scale, p.z * scale, p.w * scale};
  }
}
color_image get_render(const pathtrace_state& state) {
  auto image = color_image{state.width, state.height, true, {}};
  image.pixels.resize((size_t)state.width * state.height);
  get_render(image, state);
  return image;
}

}  // namespace vpt
