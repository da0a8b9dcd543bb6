// vpt_tesselate.cpp — tesselate_surfaces of the reference (libs/yocto_pathtrace/yocto_pathtrace.cpp:1119-1280): Catmull-Clark
// subdivision of face-varying control cages, split_facevarying, quads_to_triangles, displacement, smooth normals.  Load-time
// work on the callers' side of the hot path; its output is the mesh the BVH is built over, so it has to
// be the reference's mesh bit for bit — vertex order, triangle order, float32 positions / normals / texcoords (the parity
// fixtures hold the reference's hashes of all four arrays).
//
// One level is split into a TOPOLOGY step (integers: the edge numbering, the new faces, which vertices are creased, and,
// per new vertex, the ordered list of things whose centroid the reference adds to it) and a VERTEX step (float32: edge and
// face points, the averaging pass, the correction pass).  The reference accumulates `avert[vid] += c` while it walks the
// crease edges and then the new faces in index order; per vertex that is a sum over its incident items in increasing
// index order, which is how it is written here — on the host as a loop over vertices, on the device
// (vpt_subdivide_vertices, csrc/vpt_subdiv.hip) as one thread per vertex.  Same operations in the same order: same bits.
//
// Containers.  The reference numbers edges by first insertion into an unordered_map and lists boundary edges in that
// map's ITERATION order (get_boundary, yocto_shape.cpp:1809-1815).  The numbering is order-independent; the iteration
// order only matters for a vertex that receives three or more crease contributions (a non-manifold boundary), because
// a float sum of two terms commutes.  To be exact there as well, the edge map below is the same container with the same
// hash (yocto_shape.h:376-384) filled in the same order, so with the same libstdc++ it iterates identically.
#include <cmath>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <unordered_map>

#include "vpt_host.h"
#include "vpt_hostmath.h"

namespace vpt {

namespace {

struct edge_key {
  int  x, y;
  bool operator==(const edge_key& o) const { return x == o.x && y == o.y; }
};
struct edge_hash {   // std::hash<vec2i> of yocto_shape.h:376-384
  size_t operator()(const edge_key& v) const {
    static const auto hasher = std::hash<int>();
    auto              h      = (size_t)0;
    h ^= hasher(v.x) + 0x9e3779b9 + (h << 6) + (h >> 2);
    h ^= hasher(v.y) + 0x9e3779b9 + (h << 6) + (h >> 2);
    return h;
  }
};
struct edge_data { int index, nfaces; };
using edge_map = std::unordered_map<edge_key, edge_data, edge_hash>;

void insert_edge(edge_map& emap, int a, int b) {   // yocto_shape.cpp:1781-1794
  auto es = a < b ? edge_key{a, b} : edge_key{b, a};
  auto it = emap.find(es);
  if (it == emap.end()) emap.insert(it, {es, edge_data{(int)emap.size(), 1}});
  else it->second.nfaces += 1;
}
int edge_index(const edge_map& emap, int a, int b) {
  auto es = a < b ? edge_key{a, b} : edge_key{b, a};
  return emap.at(es).index;
}

}  // namespace

// ---- topology of one level ----------------------------------------------------------------------------------------
void catmullclark_topology(const vector<vec4i>& quads, int nv, bool lock_boundary, subdiv_level& L) {
  auto emap = edge_map{};
  for (auto& q : quads) {   // make_edge_map, yocto_shape.cpp:1755-1764
    insert_edge(emap, q.x, q.y);
    insert_edge(emap, q.y, q.z);
    if (q.z != q.w) insert_edge(emap, q.z, q.w);
    insert_edge(emap, q.w, q.x);
  }
  auto ne = (int)emap.size(), nf = (int)quads.size();
  L.nv = nv, L.ne = ne, L.nf = nf;
  L.edges.assign((size_t)ne * 2, 0);
  for (auto& [edge, data] : emap) L.edges[(size_t)data.index * 2] = edge.x, L.edges[(size_t)data.index * 2 + 1] = edge.y;
  L.faces = quads;
  // new faces (cpp:1149-1170)
  L.tquads.clear();
  L.tquads.reserve((size_t)nf * 4);
  for (auto i = 0; i < nf; i++) {
    auto& q = quads[(size_t)i];
    auto  e = [&](int a, int b) { return nv + edge_index(emap, a, b); };
    auto  f = nv + ne + i;
    if (q.z != q.w) {
      L.tquads.push_back({q.x, e(q.x, q.y), f, e(q.w, q.x)});
      L.tquads.push_back({q.y, e(q.y, q.z), f, e(q.x, q.y)});
      L.tquads.push_back({q.z, e(q.z, q.w), f, e(q.y, q.z)});
      L.tquads.push_back({q.w, e(q.w, q.x), f, e(q.z, q.w)});
    } else {
      L.tquads.push_back({q.x, e(q.x, q.y), f, e(q.z, q.x)});
      L.tquads.push_back({q.y, e(q.y, q.z), f, e(q.x, q.y)});
      L.tquads.push_back({q.z, e(q.z, q.x), f, e(q.y, q.z)});
    }
  }
  // boundary edges in the map's iteration order, each split in two (cpp:1172-1177)
  auto tboundary = vector<edge_key>{};
  for (auto& [edge, data] : emap)
    if (data.nfaces < 2) {
      auto mid = nv + data.index;
      tboundary.push_back({edge.x, mid});
      tboundary.push_back({mid, edge.y});
    }
  auto nt = nv + ne + nf;
  L.valence.assign((size_t)nt, 2);
  for (auto& e : tboundary) L.valence[(size_t)e.x] = L.valence[(size_t)e.y] = lock_boundary ? 0 : 1;
  // per new vertex, what the averaging pass adds to it, in the order the reference's loops reach it (cpp:1196-1218):
  //   valence 0 (locked):  the vertex itself, once per appearance in tcrease_verts          -> items are the vertex id
  //   valence 1 (crease):  the midpoint of every crease edge that contains it               -> items are (a, b) pairs
  //   valence 2 (smooth):  the centroid of every new face that contains it, by face index   -> items are face ids
  auto count = vector<int>((size_t)nt, 0);
  if (lock_boundary) {
    for (auto& b : tboundary) count[(size_t)b.x]++, count[(size_t)b.y]++;   // every tcrease_verts entry has valence 0
  } else {
    for (auto& b : tboundary) count[(size_t)b.x]++, count[(size_t)b.y]++;   // both ends of a crease edge have valence 1
  }
  for (auto& q : L.tquads)
    for (auto vid : {q.x, q.y, q.z, q.w})
      if (L.valence[(size_t)vid] == 2) count[(size_t)vid]++;
  L.offsets.assign((size_t)nt + 1, 0);
  for (auto v = 0; v < nt; v++) L.offsets[(size_t)v + 1] = L.offsets[(size_t)v] + count[(size_t)v] * (L.valence[(size_t)v] == 1 ? 2 : 1);
  L.items.assign((size_t)L.offsets[(size_t)nt], 0);
  auto fill = vector<int>(L.offsets.begin(), L.offsets.end() - 1);
  if (lock_boundary) {
    for (auto& b : tboundary)
      for (auto vid : {b.x, b.y}) L.items[(size_t)fill[(size_t)vid]++] = vid;
  } else {
    for (auto& b : tboundary)
      for (auto vid : {b.x, b.y}) L.items[(size_t)fill[(size_t)vid]++] = b.x, L.items[(size_t)fill[(size_t)vid]++] = b.y;
  }
  for (auto i = 0; i < (int)L.tquads.size(); i++) {
    auto& q = L.tquads[(size_t)i];
    for (auto vid : {q.x, q.y, q.z, q.w})
      if (L.valence[(size_t)vid] == 2) L.items[(size_t)fill[(size_t)vid]++] = i;
  }
}

// ---- vertex arithmetic of one level, host form ------------------------------------------------------------------------
namespace {
template <int D>
struct vecD {
  float v[D];
};
template <int D>
vecD<D> add(const vecD<D>& a, const vecD<D>& b) {
  auto r = vecD<D>{};
  for (auto k = 0; k < D; k++) r.v[k] = a.v[k] + b.v[k];
  return r;
}
template <int D>
vecD<D> sub(const vecD<D>& a, const vecD<D>& b) {
  auto r = vecD<D>{};
  for (auto k = 0; k < D; k++) r.v[k] = a.v[k] - b.v[k];
  return r;
}
template <int D>
vecD<D> div(const vecD<D>& a, float b) {
  auto r = vecD<D>{};
  for (auto k = 0; k < D; k++) r.v[k] = a.v[k] / b;
  return r;
}
template <int D>
vecD<D> mul(const vecD<D>& a, float b) {
  auto r = vecD<D>{};
  for (auto k = 0; k < D; k++) r.v[k] = a.v[k] * b;
  return r;
}

template <int D>
void subdivide_vertices_host(const subdiv_level& L, const float* verts_in, float* verts_out) {
  auto vert = (const vecD<D>*)verts_in;
  auto nt   = L.nv + L.ne + L.nf;
  auto tverts = vector<vecD<D>>((size_t)nt);
  for (auto i = 0; i < L.nv; i++) tverts[(size_t)i] = vert[i];
  for (auto i = 0; i < L.ne; i++)   // (vert[e.x] + vert[e.y]) / 2
    tverts[(size_t)(L.nv + i)] = div(add(vert[L.edges[(size_t)i * 2]], vert[L.edges[(size_t)i * 2 + 1]]), 2.0f);
  for (auto i = 0; i < L.nf; i++) {
    auto& q = L.faces[(size_t)i];
    tverts[(size_t)(L.nv + L.ne + i)] = q.z != q.w ? div(add(add(add(vert[q.x], vert[q.y]), vert[q.z]), vert[q.w]), 4.0f)
                                                   : div(add(add(vert[q.x], vert[q.y]), vert[q.z]), 3.0f);
  }
  auto out = (vecD<D>*)verts_out;
  for (auto v = 0; v < nt; v++) {
    auto avert  = vecD<D>{};   // T()
    auto acount = 0;
    auto val    = L.valence[(size_t)v];
    for (auto k = L.offsets[(size_t)v]; k < L.offsets[(size_t)v + 1];) {
      auto c = vecD<D>{};
      if (val == 0) c = tverts[(size_t)L.items[(size_t)k]], k += 1;
      else if (val == 1) c = div(add(tverts[(size_t)L.items[(size_t)k]], tverts[(size_t)L.items[(size_t)k + 1]]), 2.0f), k += 2;
      else {
        auto& q = L.tquads[(size_t)L.items[(size_t)k]];
        c = div(add(add(add(tverts[(size_t)q.x], tverts[(size_t)q.y]), tverts[(size_t)q.z]), tverts[(size_t)q.w]), 4.0f), k += 1;
      }
      avert = add(avert, c);
      acount += 1;
    }
    avert = div(avert, (float)acount);   // 0 / 0 for a vertex no face refers to, as in the reference
    if (val == 2) avert = add(tverts[(size_t)v], mul(sub(avert, tverts[(size_t)v]), 4 / (float)acount));
    out[v] = avert;
  }
}
}  // namespace

void subdivide_vertices(const subdiv_level& L, int dim, const vector<float>& verts, vector<float>& out) {
  out.assign((size_t)(L.nv + L.ne + L.nf) * dim, 0.0f);
  if (dim == 3) subdivide_vertices_host<3>(L, verts.data(), out.data());
  else if (dim == 2) subdivide_vertices_host<2>(L, verts.data(), out.data());
  else throw std::invalid_argument{"tesselate_catmullclark: 2 or 3 floats per vertex"};
}

void tesselate_catmullclark(vector<vec4i>& quads, vector<float>& verts, int dim, bool lock_boundary) {
  if (dim != 2 && dim != 3) throw std::invalid_argument{"tesselate_catmullclark: 2 or 3 floats per vertex"};
  auto L = subdiv_level{};
  catmullclark_topology(quads, (int)(verts.size() / (size_t)dim), lock_boundary, L);
  auto next = vector<float>{};
  subdivide_vertices(L, dim, verts, next);
  verts = std::move(next);
  quads = std::move(L.tquads);
}

// ---- the rest of tesselate_surface ----------------------------------------------------------------------------------------
namespace {
vec3f normalize(const vec3f& a) {
  auto l = length(a);
  return (l != 0) ? a / l : a;
}
vec3f triangle_normal(const vec3f& p0, const vec3f& p1, const vec3f& p2) { return normalize(cross(p1 - p0, p2 - p0)); }
vec3f quad_normal(const vec3f& p0, const vec3f& p1, const vec3f& p2, const vec3f& p3) {
  return normalize(triangle_normal(p0, p1, p3) + triangle_normal(p2, p3, p1));
}
// quads_normals / triangles_normals, yocto_shape.cpp:1478-1512
vector<vec3f> quads_normals(const vector<vec4i>& quads, const vector<vec3f>& positions) {
  auto normals = vector<vec3f>(positions.size(), vec3f{0, 0, 0});
  for (auto& q : quads) {
    auto normal = quad_normal(positions[(size_t)q.x], positions[(size_t)q.y], positions[(size_t)q.z], positions[(size_t)q.w]);
    auto area   = quad_area(positions[(size_t)q.x], positions[(size_t)q.y], positions[(size_t)q.z], positions[(size_t)q.w]);
    normals[(size_t)q.x] = normals[(size_t)q.x] + normal * area;
    normals[(size_t)q.y] = normals[(size_t)q.y] + normal * area;
    normals[(size_t)q.z] = normals[(size_t)q.z] + normal * area;
    if (q.z != q.w) normals[(size_t)q.w] = normals[(size_t)q.w] + normal * area;
  }
  for (auto& normal : normals) normal = normalize(normal);
  return normals;
}
vector<vec3f> triangles_normals(const vector<vec3i>& triangles, const vector<vec3f>& positions) {
  auto normals = vector<vec3f>(positions.size(), vec3f{0, 0, 0});
  for (auto& t : triangles) {
    auto normal = triangle_normal(positions[(size_t)t.x], positions[(size_t)t.y], positions[(size_t)t.z]);
    auto area   = triangle_area(positions[(size_t)t.x], positions[(size_t)t.y], positions[(size_t)t.z]);
    normals[(size_t)t.x] = normals[(size_t)t.x] + normal * area;
    normals[(size_t)t.y] = normals[(size_t)t.y] + normal * area;
    normals[(size_t)t.z] = normals[(size_t)t.z] + normal * area;
  }
  for (auto& normal : normals) normal = normalize(normal);
  return normals;
}

// split_facevarying, yocto_shape.cpp:2597-2649: one vertex per distinct (position, normal, texcoord) index triple, numbered by
// first appearance over the faces
struct triple_hash {   // std::hash<vec3i> of yocto_shape.h:386-395 (only lookups: the numbering does not depend on it)
  size_t operator()(const vec3i& v) const {
    static const auto hasher = std::hash<int>();
    auto              h      = (size_t)0;
    h ^= hasher(v.x) + 0x9e3779b9 + (h << 6) + (h >> 2);
    h ^= hasher(v.y) + 0x9e3779b9 + (h << 6) + (h >> 2);
    h ^= hasher(v.z) + 0x9e3779b9 + (h << 6) + (h >> 2);
    return h;
  }
};
struct triple_eq {
  bool operator()(const vec3i& a, const vec3i& b) const { return a.x == b.x && a.y == b.y && a.z == b.z; }
};
void split_facevarying(shape_data& shape, const subdiv_data& subdiv) {
  auto vert_map = std::unordered_map<vec3i, int, triple_hash, triple_eq>{};
  auto verts    = vector<vec3i>{};
  shape.quads.resize(subdiv.quadspos.size());
  auto comp4 = [](const vec4i& q, int c) { return c == 0 ? q.x : c == 1 ? q.y : c == 2 ? q.z : q.w; };
  for (size_t fid = 0; fid < subdiv.quadspos.size(); fid++) {
    int out[4];
    for (auto c = 0; c < 4; c++) {
      auto v = vec3i{comp4(subdiv.quadspos[fid], c), !subdiv.quadsnorm.empty() ? comp4(subdiv.quadsnorm[fid], c) : -1,
          !subdiv.quadstexcoord.empty() ? comp4(subdiv.quadstexcoord[fid], c) : -1};
      auto it = vert_map.find(v);
      if (it == vert_map.end()) {
        out[c] = (int)vert_map.size();
        vert_map.insert(it, {v, out[c]});
        verts.push_back(v);
      } else out[c] = it->second;
    }
    shape.quads[fid] = {out[0], out[1], out[2], out[3]};
  }
  shape.positions.clear(), shape.normals.clear(), shape.texcoords.clear();
  if (!subdiv.positions.empty()) {
    shape.positions.resize(verts.size());
    for (size_t i = 0; i < verts.size(); i++) shape.positions[i] = subdiv.positions[(size_t)verts[i].x];
  }
  if (!subdiv.normals.empty()) {
    shape.normals.resize(verts.size());
    for (size_t i = 0; i < verts.size(); i++) shape.normals[i] = subdiv.normals[(size_t)verts[i].y];
  }
  if (!subdiv.texcoords.empty()) {
    shape.texcoords.resize(verts.size());
    for (size_t i = 0; i < verts.size(); i++) shape.texcoords[i] = subdiv.texcoords[(size_t)verts[i].z];
  }
}

// eval_texture(texture, uv, as_linear = true) (yocto_scene.cpp:128-161, lookup_texture :112-125): tiled bilinear lookup, 8-bit
// texels through byte_to_float and - unless the texture is linear - srgb_to_rgb (yocto_color.h:212-227)
vec4f lookup_texture_linear(const texture_data& texture, int i, int j) {
  if (!texture.pixelsf.empty()) return texture.pixelsf[(size_t)j * texture.width + i];   // as_linear only converts non-linear textures: float ones are linear
  auto b     = texture.pixelsb[(size_t)j * texture.width + i];
  auto color = vec4f{b.x / 255.0f, b.y / 255.0f, b.z / 255.0f, b.w / 255.0f};
  if (texture.linear) return color;
  auto srgb_to_rgb = [](float srgb) { return (srgb <= 0.04045) ? srgb / 12.92f : std::pow((srgb + 0.055f) / (1.0f + 0.055f), 2.4f); };
  return {srgb_to_rgb(color.x), srgb_to_rgb(color.y), srgb_to_rgb(color.z), color.w};
}
vec4f eval_texture_linear(const texture_data& texture, const vec2f& uv) {
  if (texture.width == 0 || texture.height == 0) return {0, 0, 0, 0};
  auto s = std::fmod(uv.x, 1.0f) * texture.width;
  if (s < 0) s += texture.width;
  auto t = std::fmod(uv.y, 1.0f) * texture.height;
  if (t < 0) t += texture.height;
  auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; };
  auto i = clampi((int)s, 0, texture.width - 1), j = clampi((int)t, 0, texture.height - 1);
  auto ii = (i + 1) % texture.width, jj = (j + 1) % texture.height;
  auto u = s - i, v = t - j;
  auto scale = [](const vec4f& a, float b) { return vec4f{a.x * b, a.y * b, a.z * b, a.w * b}; };
  auto sum   = [](const vec4f& a, const vec4f& b) { return vec4f{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; };
  return sum(sum(sum(scale(scale(lookup_texture_linear(texture, i, j), 1 - u), 1 - v), scale(scale(lookup_texture_linear(texture, i, jj), 1 - u), v)),
                 scale(scale(lookup_texture_linear(texture, ii, j), u), 1 - v)),
      scale(scale(lookup_texture_linear(texture, ii, jj), u), v));
}

void check_cage(const subdiv_data& subdiv) {   // the reference trusts its files; indices reach device memory here
  auto ok = [](const vector<vec4i>& quads, size_t n) {
    for (auto& q : quads)
      for (auto v : {q.x, q.y, q.z, q.w})
        if (v < 0 || (size_t)v >= n) return false;
    return true;
  };
  if (!ok(subdiv.quadspos, subdiv.positions.size()) || !ok(subdiv.quadsnorm, subdiv.normals.size()) || !ok(subdiv.quadstexcoord, subdiv.texcoords.size()) ||
      (!subdiv.quadsnorm.empty() && subdiv.quadsnorm.size() != subdiv.quadspos.size()) ||
      (!subdiv.quadstexcoord.empty() && subdiv.quadstexcoord.size() != subdiv.quadspos.size()))
    throw std::invalid_argument{"subdiv cage with indices out of range"};
}

// where the float work of tesselate_surface runs: the host loops above, or the device kernels of csrc/vpt_subdiv.hip
struct tess_ops {
  std::function<void(vector<vec4i>&, vector<float>&, int, bool)>                           level;             // one Catmull-Clark level on (quads, flat vertex floats)
  std::function<vector<vec3f>(const vector<vec4i>&, const vector<vec3f>&)>                 normals_of_quads;
  std::function<vector<vec3f>(const vector<vec3i>&, const vector<vec3f>&)>                 normals_of_triangles;
  std::function<void(const texture_data&, float, const shape_data&, vector<vec3f>&)>       displace;          // positions out
};
void displace_host(const texture_data& displacement_tex, float displacement, const shape_data& shape, vector<vec3f>& positions) {
  positions.resize(shape.positions.size());
  for (size_t idx = 0; idx < shape.positions.size(); idx++) {
    auto texel = eval_texture_linear(displacement_tex, shape.texcoords[idx]);
    auto disp  = (texel.x + texel.y + texel.z) / 3;   // mean(xyz(.))
    if (!displacement_tex.pixelsb.empty()) disp -= 0.5f;
    positions[idx] = shape.positions[idx] + shape.normals[idx] * displacement * disp;
  }
}
// tesselate_surface, yocto_pathtrace.cpp:1228-1273
void tesselate_surface(shape_data& shape, const subdiv_data& subdiv_, const scene_data& scene, const tess_ops& ops) {
  auto& level = ops.level;
  auto subdiv = subdiv_;
  check_cage(subdiv);
  if (subdiv.subdivisions != 0) {
    auto flat = vector<float>(subdiv.positions.size() * 3);
    memcpy(flat.data(), subdiv.positions.data(), flat.size() * 4);
    for (auto l = 0; l < subdiv.subdivisions; l++) level(subdiv.quadspos, flat, 3, false);
    subdiv.positions.resize(flat.size() / 3);
    memcpy((void*)subdiv.positions.data(), flat.data(), flat.size() * 4);
    flat.assign(subdiv.texcoords.size() * 2, 0.0f);
    memcpy(flat.data(), subdiv.texcoords.data(), flat.size() * 4);
    for (auto l = 0; l < subdiv.subdivisions; l++) level(subdiv.quadstexcoord, flat, 2, true);
    subdiv.texcoords.resize(flat.size() / 2);
    memcpy((void*)subdiv.texcoords.data(), flat.data(), flat.size() * 4);
    if (subdiv.smooth) {
      subdiv.normals   = ops.normals_of_quads(subdiv.quadspos, subdiv.positions);
      subdiv.quadsnorm = subdiv.quadspos;
    } else {
      subdiv.normals   = {};
      subdiv.quadsnorm = {};
    }
  }
  split_facevarying(shape, subdiv);
  shape.triangles.clear();   // quads_to_triangles, yocto_shape.cpp:2565-2573
  shape.triangles.reserve(shape.quads.size() * 2);
  for (auto& q : shape.quads) {
    shape.triangles.push_back({q.x, q.y, q.w});
    if (q.z != q.w) shape.triangles.push_back({q.z, q.w, q.y});
  }
  shape.quads  = {};
  shape.points = {};
  if (subdiv.displacement != 0 && subdiv.displacement_tex >= 0 && !shape.triangles.empty()) {
    if (shape.texcoords.size() != shape.positions.size()) throw std::invalid_argument{"displaced subdiv without texture coordinates"};
    if (shape.normals.empty()) shape.normals = ops.normals_of_triangles(shape.triangles, shape.positions);
    auto& displacement_tex = scene.textures.at((size_t)subdiv.displacement_tex);
    auto  displaced = vector<vec3f>{};
    ops.displace(displacement_tex, subdiv.displacement, shape, displaced);
    shape.positions = std::move(displaced);
    if (subdiv.smooth) shape.normals = ops.normals_of_triangles(shape.triangles, shape.positions);
    else shape.normals = {};
  }
}
}  // namespace

void tesselate_surfaces(scene_data& scene) {
  auto ops = tess_ops{[](vector<vec4i>& quads, vector<float>& verts, int dim, bool lock) { tesselate_catmullclark(quads, verts, dim, lock); },
      quads_normals, triangles_normals, displace_host};
  for (auto& subdiv : scene.subdivs) tesselate_surface(scene.shapes.at((size_t)subdiv.shape), subdiv, scene, ops);
}

// the same stages one at a time (tests, profiles/tools/tesselation_timing.py): device < 0 host, else that GPU
vector<vec3f> vertex_normals(const vector<vec3f>& positions, const int* faces, int num_faces, int corners, int device) {
  if (device < 0) {
    if (corners == 4) return quads_normals(vector<vec4i>((const vec4i*)faces, (const vec4i*)faces + num_faces), positions);
    return triangles_normals(vector<vec3i>((const vec3i*)faces, (const vec3i*)faces + num_faces), positions);
  }
  auto normals = vector<vec3f>(positions.size());
  if (vpt_vertex_normals(device, (int)positions.size(), &positions.data()->x, num_faces, corners, faces, &normals.data()->x) != VPT_OK)
    throw std::runtime_error{string{"vpt_vertex_normals: "} + vpt_last_error()};
  return normals;
}
static void displace_device(int device, const texture_data& tex, float displacement, const shape_data& shape, vector<vec3f>& positions) {
  auto desc = vpt_texture{tex.width, tex.height, tex.linear ? 1 : 0, tex.pixelsf.empty() ? 0 : 1, 0};
  auto texels = tex.pixelsf.empty() ? (const void*)tex.pixelsb.data() : (const void*)tex.pixelsf.data();
  positions.resize(shape.positions.size());
  if (vpt_displace_vertices(device, &desc, texels, displacement, (int)shape.positions.size(), &shape.positions.data()->x, &shape.normals.data()->x,
          &shape.texcoords.data()->x, &positions.data()->x) != VPT_OK)
    throw std::runtime_error{string{"vpt_displace_vertices: "} + vpt_last_error()};
}
vector<vec3f> displace_vertices(const texture_data& tex, float displacement, const vector<vec3f>& positions, const vector<vec3f>& normals,
    const vector<vec2f>& texcoords, int device) {
  if (normals.size() != positions.size() || texcoords.size() != positions.size()) throw std::invalid_argument{"displacement needs a normal and a texture coordinate per vertex"};
  auto shape = shape_data{};
  shape.positions = positions, shape.normals = normals, shape.texcoords = texcoords;
  auto out = vector<vec3f>{};
  if (device < 0) displace_host(tex, displacement, shape, out);
  else displace_device(device, tex, displacement, shape, out);
  return out;
}

void tesselate_surfaces_device(scene_data& scene, int device) {
  auto level = [device](vector<vec4i>& quads, vector<float>& verts, int dim, bool lock) {
    auto L = subdiv_level{};
    catmullclark_topology(quads, (int)(verts.size() / (size_t)dim), lock, L);
    auto next = vector<float>((size_t)(L.nv + L.ne + L.nf) * dim);
    auto desc = vpt_subdiv_level{dim, L.nv, L.ne, L.nf, (int)L.tquads.size(), L.edges.data(), &L.faces.data()->x, &L.tquads.data()->x,
        L.valence.data(), L.offsets.data(), L.items.data(), (int64_t)L.items.size()};
    if (vpt_subdivide_vertices(device, &desc, verts.data(), next.data()) != VPT_OK)
      throw std::runtime_error{string{"vpt_subdivide_vertices: "} + vpt_last_error()};
    verts = std::move(next);
    quads = std::move(L.tquads);
  };
  auto ops = tess_ops{level,
      [device](const vector<vec4i>& quads, const vector<vec3f>& positions) { return vertex_normals(positions, &quads.data()->x, (int)quads.size(), 4, device); },
      [device](const vector<vec3i>& triangles, const vector<vec3f>& positions) { return vertex_normals(positions, &triangles.data()->x, (int)triangles.size(), 3, device); },
      [device](const texture_data& tex, float displacement, const shape_data& shape, vector<vec3f>& positions) { displace_device(device, tex, displacement, shape, positions); }};
  for (auto& subdiv : scene.subdivs) tesselate_surface(scene.shapes.at((size_t)subdiv.shape), subdiv, scene, ops);
}

}  // namespace vpt
