// vpt_host_capi.cpp — C entry points over the host library for the Python harness
// (tests/, bench.py): load a scene.json, build bvh + lights, hand out the flattened
// vpt_scene_desc, seed a pathtrace_state, quantise/encode output.  Host-only; no GPU calls.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>

#include "vpt_host.h"

using namespace vpt;

namespace {
struct host_scene {
  scene_data       scene;
  bvh_scene        bvh;
  pathtrace_lights lights;
  std::unique_ptr<flat_scene> flat = std::make_unique<flat_scene>();
};
void set_error(char* err, int errlen, const string& msg) {
  if (err && errlen > 0) snprintf(err, (size_t)errlen, "%s", msg.c_str());
}
uint64_t fnv1a(const void* data, size_t nbytes) {
  auto p = (const unsigned char*)data;
  auto h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < nbytes; i++) h = (h ^ p[i]) * 0x100000001b3ull;
  return h;
}
template <typename T>
uint64_t fnv1a(const vector<T>& v) { return fnv1a(v.data(), v.size() * sizeof(T)); }
}  // namespace

extern "C" {

// tess_device >= 0: the vertex arithmetic of tesselate_surfaces on that GPU (tesselate_surfaces_device: same mesh)
void* vpth_scene_load_ex(const char* filename, int tess_device, char* err, int errlen) {
  try {
    auto h     = std::make_unique<host_scene>();
    auto error = string{};
    if (!load_scene(filename, h->scene, error)) return set_error(err, errlen, error), nullptr;
    if (tess_device >= 0) tesselate_surfaces_device(h->scene, tess_device);
    else tesselate_surfaces(h->scene);
    auto params = pathtrace_params{};
    h->bvh      = make_bvh(h->scene, params);
    h->lights   = make_lights(h->scene, params);
    flatten_scene(*h->flat, h->scene, h->bvh, h->lights);
    return h.release();
  } catch (const std::exception& e) {
    return set_error(err, errlen, e.what()), nullptr;
  }
}
void* vpth_scene_load(const char* filename, char* err, int errlen) { return vpth_scene_load_ex(filename, -1, err, errlen); }
// one level of tesselate_catmullclark on a quad mesh with `dim` floats per vertex (host arithmetic, or GPU `device` >= 0 for the
// vertex half): quads_out has room for 4 * nquads entries, verts_out for (nverts + 4 * nquads + nquads) * dim floats (edges <= 4 per face)
int vpth_catmullclark(const int32_t* quads, int nquads, const float* verts, int nverts, int dim, int lock_boundary, int device, int32_t* quads_out,
    int* nquads_out, float* verts_out, int* nverts_out, char* err, int errlen) {
  try {
    auto q = vector<vec4i>((size_t)nquads);
    memcpy((void*)q.data(), quads, (size_t)nquads * 16);
    for (auto& f : q)
      for (auto v : {f.x, f.y, f.z, f.w})
        if (v < 0 || v >= nverts) throw std::invalid_argument{"face index out of range"};
    auto v = vector<float>(verts, verts + (size_t)nverts * dim);
    if (device < 0) tesselate_catmullclark(q, v, dim, lock_boundary != 0);
    else {
      auto L = subdiv_level{};
      catmullclark_topology(q, nverts, lock_boundary != 0, L);
      auto next = vector<float>((size_t)(L.nv + L.ne + L.nf) * dim);
      auto desc = vpt_subdiv_level{dim, L.nv, L.ne, L.nf, (int)L.tquads.size(), L.edges.data(), &L.faces.data()->x, &L.tquads.data()->x,
          L.valence.data(), L.offsets.data(), L.items.data(), (int64_t)L.items.size()};
      if (vpt_subdivide_vertices(device, &desc, v.data(), next.data()) != VPT_OK) throw std::runtime_error{vpt_last_error()};
      v = std::move(next), q = std::move(L.tquads);
    }
    memcpy(quads_out, q.data(), q.size() * 16), memcpy(verts_out, v.data(), v.size() * 4);
    *nquads_out = (int)q.size(), *nverts_out = (int)(v.size() / (size_t)dim);
    return 0;
  } catch (const std::exception& e) {
    return set_error(err, errlen, e.what()), -1;
  }
}
// quads_normals (corners = 4) / triangles_normals (3) over float3 positions: host loops (device < 0) or vpt_vertex_normals on that GPU
int vpth_vertex_normals(const float* positions, int nverts, const int32_t* faces, int nfaces, int corners, int device, float* normals, char* err, int errlen) {
  try {
    if (corners != 3 && corners != 4) throw std::invalid_argument{"corners must be 3 or 4"};
    for (long long i = 0; i < (long long)nfaces * corners; i++)
      if (faces[i] < 0 || faces[i] >= nverts) throw std::invalid_argument{"face index out of range"};
    auto pos = vector<vec3f>((size_t)nverts);
    memcpy((void*)pos.data(), positions, (size_t)nverts * 12);
    auto out = vertex_normals(pos, faces, nfaces, corners, device);
    memcpy(normals, out.data(), (size_t)nverts * 12);
    return 0;
  } catch (const std::exception& e) {
    return set_error(err, errlen, e.what()), -1;
  }
}
// the displacement step of tesselate_surface over float3 positions / normals and float2 texcoords; texels: uchar4 or float4 by is_float
int vpth_displace_vertices(const void* texels, int width, int height, int is_float, int linear, float displacement, const float* positions,
    const float* normals, const float* texcoords, int nverts, int device, float* out_positions, char* err, int errlen) {
  try {
    auto tex = texture_data{};
    tex.width = width, tex.height = height, tex.linear = linear != 0;
    if (is_float) tex.pixelsf.assign((const vec4f*)texels, (const vec4f*)texels + (size_t)width * height);
    else tex.pixelsb.assign((const vec4b*)texels, (const vec4b*)texels + (size_t)width * height);
    auto pos = vector<vec3f>((size_t)nverts), nrm = vector<vec3f>((size_t)nverts);
    auto uv  = vector<vec2f>((size_t)nverts);
    memcpy((void*)pos.data(), positions, (size_t)nverts * 12), memcpy((void*)nrm.data(), normals, (size_t)nverts * 12);
    memcpy((void*)uv.data(), texcoords, (size_t)nverts * 8);
    auto out = displace_vertices(tex, displacement, pos, nrm, uv, device);
    memcpy(out_positions, out.data(), (size_t)nverts * 12);
    return 0;
  } catch (const std::exception& e) {
    return set_error(err, errlen, e.what()), -1;
  }
}
// rebuild the scene's BVHs on GPU `device` (make_bvh_device) and flatten again; 0 on success
int vpth_scene_rebuild_bvh_device(void* hh, int device, char* err, int errlen) {
  try {
    auto& h = *(host_scene*)hh;
    h.bvh   = make_bvh_device(h.scene, pathtrace_params{}, device);
    auto flat = std::make_unique<flat_scene>();
    flatten_scene(*flat, h.scene, h.bvh, h.lights);
    h.flat = std::move(flat);   // the address vpth_scene_desc handed out before is gone: callers fetch it again
    return 0;
  } catch (const std::exception& e) {
    return set_error(err, errlen, e.what()), -1;
  }
}
// build_bvh over n boxes on the host (the checker of vpt_build_bvh in tests): nodes has room for max(1, 2 n) entries
int vpth_build_bvh_host(const float* bboxes, int n, vpt_bvh_node* nodes, int* num_nodes, int32_t* primitives) {
  auto bvh = build_bvh_host(bboxes, n);
  memcpy(nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(vpt_bvh_node));
  if (n > 0) memcpy(primitives, bvh.primitives.data(), (size_t)n * sizeof(int32_t));
  *num_nodes = (int)bvh.nodes.size();
  return 0;
}
void vpth_scene_free(void* h) { delete (host_scene*)h; }
const vpt_scene_desc* vpth_scene_desc(void* h) { return &((host_scene*)h)->flat->desc; }

// make_state dimensions (yocto_pathtrace.cpp:964-970)
int vpth_state_size(void* h, int camera, int resolution, int* width, int* height) {
  auto& scene = ((host_scene*)h)->scene;
  if (camera < 0 || camera >= (int)scene.cameras.size()) return -1;
  auto aspect       = scene.cameras[camera].aspect;
  if (aspect >= 1) *width = resolution, *height = (int)std::round(resolution / aspect);
  else *height = resolution, *width = (int)std::round(resolution * aspect);
  return 0;
}
// fills caller arrays of w*h entries: image float4 zeros, hits zeros, rng {state, inc}
int vpth_make_state(void* h, int camera, int resolution, float* image, int32_t* hits, uint64_t* rng) {
  try {
    auto params       = pathtrace_params{};
  (void)params;
    params.camera     = camera;
    params.resolution = resolution;
    auto state        = make_state(((host_scene*)h)->scene, params);
    auto n            = state.image.size();
    memset(image, 0, n * 16), memset(hits, 0, n * 4);
    memcpy(rng, state.rngs.data(), n * 16);
    return 0;
  } catch (...) {
    return -1;
  }
}

// same JSON shape as oracle/ref_driver.cpp --stats, so tests can diff the two verbatim
int vpth_scene_stats(void* hh, char* buf, int buflen) {
  auto& h = *(host_scene*)hh;
  auto  s = string{};
  char  tmp[1024];
  auto  add = [&](const char* fmt, auto... args) {
    if constexpr (sizeof...(args) == 0) s += fmt;
    else {
      snprintf(tmp, sizeof(tmp), fmt, args...);
      s += tmp;
    }
  };
  add("{\n \"scene_bvh\": {\"nodes\": %zu, \"prims\": %zu, \"nodes_fnv\": \"%016llx\", \"prims_fnv\": \"%016llx\"},\n",
      h.bvh.nodes.size(), h.bvh.primitives.size(), (unsigned long long)fnv1a(h.bvh.nodes),
      (unsigned long long)fnv1a(h.bvh.primitives));
  add(" \"shapes\": [\n");
  for (size_t i = 0; i < h.scene.shapes.size(); i++) {
    auto& sh = h.scene.shapes[i];
    auto& b  = h.bvh.shapes[i];
    add("  {\"positions\": %zu, \"normals\": %zu, \"texcoords\": %zu, \"colors\": %zu, "
        "\"triangles\": %zu, \"quads\": %zu, \"pos_fnv\": \"%016llx\", \"nrm_fnv\": \"%016llx\", "
        "\"uv_fnv\": \"%016llx\", \"tri_fnv\": \"%016llx\", \"quad_fnv\": \"%016llx\", "
        "\"bvh_nodes\": %zu, \"bvh_nodes_fnv\": \"%016llx\", \"bvh_prims_fnv\": \"%016llx\"}%s\n",
        sh.positions.size(), sh.normals.size(), sh.texcoords.size(), sh.colors.size(), sh.triangles.size(),
        sh.quads.size(), (unsigned long long)fnv1a(sh.positions), (unsigned long long)fnv1a(sh.normals),
        (unsigned long long)fnv1a(sh.texcoords), (unsigned long long)fnv1a(sh.triangles),
        (unsigned long long)fnv1a(sh.quads), b.nodes.size(), (unsigned long long)fnv1a(b.nodes),
        (unsigned long long)fnv1a(b.primitives), i + 1 < h.scene.shapes.size() ? "," : "");
  }
  add(" ],\n \"textures\": [\n");
  for (size_t i = 0; i < h.scene.textures.size(); i++) {
    auto& t = h.scene.textures[i];
    add("  {\"width\": %d, \"height\": %d, \"linear\": %d, \"f_fnv\": \"%016llx\", \"b_fnv\": \"%016llx\"}%s\n",
        t.width, t.height, (int)t.linear, (unsigned long long)fnv1a(t.pixelsf),
        (unsigned long long)fnv1a(t.pixelsb), i + 1 < h.scene.textures.size() ? "," : "");
  }
  add(" ],\n \"volumes\": [\n");
  for (size_t i = 0; i < h.scene.volumes.size(); i++) {
    auto& v = h.scene.volumes[i];
    add("  {\"whd\": [%d, %d, %d], \"res\": %.9g, \"n\": %zu, \"fnv\": \"%016llx\"}%s\n", v.whd.x, v.whd.y,
        v.whd.z, v.res, v.vol.size(), (unsigned long long)fnv1a(v.vol),
        i + 1 < h.scene.volumes.size() ? "," : "");
  }
  add(" ],\n \"lights\": [\n");
  for (size_t i = 0; i < h.lights.lights.size(); i++) {
    auto& l = h.lights.lights[i];
    add("  {\"instance\": %d, \"environment\": %d, \"sdf\": %d, \"cdf_len\": %zu, \"cdf_back\": %.9g, \"cdf_fnv\": \"%016llx\"}%s\n",
        l.instance, l.environment, l.sdf, l.elements_cdf.size(),
        l.elements_cdf.empty() ? 0.0f : l.elements_cdf.back(), (unsigned long long)fnv1a(l.elements_cdf),
        i + 1 < h.lights.lights.size() ? "," : "");
  }
  add(" ]\n}\n");
  if ((int)s.size() + 1 > buflen) return -(int)s.size() - 1;
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}

// output stage: linear float4 sums / samples -> sRGB8 (w*h*4 bytes)
void vpth_linear_to_srgb8(int width, int height, const float* image_sum, int samples, uint8_t* rgba8) {
  auto img   = color_image{width, height, true, {}};
  auto scale = 1.0f / (float)samples;
  img.pixels.resize((size_t)width * height);
  for (size_t i = 0; i < img.pixels.size(); i++)
    img.pixels[i] = {image_sum[4 * i] * scale, image_sum[4 * i + 1] * scale, image_sum[4 * i + 2] * scale,
        image_sum[4 * i + 3] * scale};
  auto ldr = linear_to_srgb8(img);
  memcpy(rgba8, ldr.data(), ldr.size() * 4);
}
// returns the encoded size; call with out == nullptr to query
int64_t vpth_encode_jpeg_q75(int width, int height, const uint8_t* rgba8, uint8_t* out, int64_t outlen) {
  auto px = vector<vec4b>((size_t)width * height);
  memcpy(px.data(), rgba8, px.size() * 4);
  auto bytes = encode_jpeg_q75(width, height, px);
  if (out && outlen >= (int64_t)bytes.size()) memcpy(out, bytes.data(), bytes.size());
  return (int64_t)bytes.size();
}

}  // extern "C"
