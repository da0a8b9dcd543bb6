// vpt_subdiv.hip — the float32 half of one Catmull-Clark level of the reference's tesselate_catmullclark
// (libs/yocto_pathtrace/yocto_pathtrace.cpp:1119-1226) on the device: SURVEY §8(f) row 4 (load-time callers of the hot path).
//
// The reference builds the refined vertex set (old vertices, one point per edge, one per face), then ACCUMULATES: it walks
// the crease edges and the new faces in index order and adds each one's centroid to the vertices it touches, divides by the
// count and pulls smooth vertices towards the average (`tverts + (avert - tverts) * (4 / count)`).  A float sum depends on
// its order, so the device must add the same terms in the same order.  Per vertex that order is "its incident items by
// increasing index", which the host-side topology step (host/vpt_tesselate.cpp: catmullclark_topology) hands over as a CSR
// list; then every vertex is independent:
//   K_points   one thread per refined vertex: copy / (a + b) / 2 / ((a + b) + c) + d) / 4 (or / 3 for a triangle)
//   K_average  one thread per refined vertex: for its items in order: centroid from the refined points, avert += c;
//              avert /= n; smooth vertices: the correction
// A face centroid is recomputed by each of its four vertices (the same four loads and three adds: same bits) instead of being
// stored once - 4x the arithmetic of the reference on data that streams from L2, no atomics, no ordering problem.
// Both kernels are bound by HBM: per refined vertex 12 (8) B written twice and read ~9 times (valence-4 faces x 4 corners
// hit in cache), ~60 B of index data.  Built with -ffp-contract=off; f32 division is hipcc's correctly rounded one.
//
// Round 4: the rest of tesselate_surface's float work (yocto_pathtrace.cpp:1239, 1256-1271) by the same ordered-list scheme:
//   K_normals   quads_normals / triangles_normals (yocto_shape.cpp:1478-1512): one thread per vertex walks its incident faces in
//               FACE ORDER (a CSR list the host entry point builds: integers), recomputes each face's normal and area from the
//               corner positions - the reference's expressions, operand for operand - and adds normal * area, then normalises;
//   K_displace  positions += normals * displacement * (mean(xyz(eval_texture(tex, uv, as_linear))) - 0.5 for 8-bit textures):
//               one thread per vertex through the render kernels' OWN eval_texture (vpt_scene.hip.h, pinned bit for bit by the
//               known-answer table texture_surface), fed by a one-texture DScene.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "vpt.h"
#include "vpt_scene.hip.h"

int vpt_set_error(int code, const char* fmt, ...);   // vpt_capi.hip

namespace {

template <int D>
struct vecD {
  float v[D];
};
template <int D>
__device__ inline vecD<D> add(const vecD<D>& a, const vecD<D>& b) {
  vecD<D> r;
#pragma unroll
  for (int k = 0; k < D; k++) r.v[k] = a.v[k] + b.v[k];
  return r;
}
template <int D>
__device__ inline vecD<D> divs(const vecD<D>& a, float b) {
  vecD<D> r;
#pragma unroll
  for (int k = 0; k < D; k++) r.v[k] = a.v[k] / b;
  return r;
}

template <int D>
__global__ void __launch_bounds__(256) subdiv_points_kernel(int nv, int ne, int nf, const vecD<D>* __restrict__ vert, const int2* __restrict__ edges,
    const int4* __restrict__ faces, vecD<D>* __restrict__ tverts) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv + ne + nf) return;
  if (i < nv) tverts[i] = vert[i];
  else if (i < nv + ne) {
    int2 e    = edges[i - nv];
    tverts[i] = divs(add(vert[e.x], vert[e.y]), 2.0f);
  } else {
    int4 q    = faces[i - nv - ne];
    tverts[i] = q.z != q.w ? divs(add(add(add(vert[q.x], vert[q.y]), vert[q.z]), vert[q.w]), 4.0f) : divs(add(add(vert[q.x], vert[q.y]), vert[q.z]), 3.0f);
  }
}

template <int D>
__global__ void __launch_bounds__(256) subdiv_average_kernel(int nt, const vecD<D>* __restrict__ tverts, const int4* __restrict__ tquads,
    const int* __restrict__ valence, const int* __restrict__ offsets, const int* __restrict__ items, vecD<D>* __restrict__ out) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nt) return;
  vecD<D> avert;
#pragma unroll
  for (int k = 0; k < D; k++) avert.v[k] = 0.0f;
  int       acount = 0;
  const int val = valence[v], end = offsets[v + 1];
  for (int k = offsets[v]; k < end;) {
    vecD<D> c;
    if (val == 0) c = tverts[items[k]], k += 1;
    else if (val == 1) c = divs(add(tverts[items[k]], tverts[items[k + 1]]), 2.0f), k += 2;
    else {
      int4 q = tquads[items[k]];
      c = divs(add(add(add(tverts[q.x], tverts[q.y]), tverts[q.z]), tverts[q.w]), 4.0f), k += 1;
    }
    avert = add(avert, c);
    acount += 1;
  }
  avert = divs(avert, (float)acount);
  if (val == 2) {
    vecD<D> t = tverts[v];
    float   w = 4 / (float)acount;
#pragma unroll
    for (int k = 0; k < D; k++) avert.v[k] = t.v[k] + (avert.v[k] - t.v[k]) * w;
  }
  out[v] = avert;
}

// ---- normals and displacement ------------------------------------------------------------------------------------------------
__device__ inline float tri_area(f3 p0, f3 p1, f3 p2) { return length(cross(p1 - p0, p2 - p0)) / 2; }   // yocto_geometry.h:506-510
__device__ inline f3 ld_f3(const float* p, int i) { return mk3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
// items[offsets[v] .. offsets[v + 1]): the faces that add to vertex v, once per corner that is v, in face order
__global__ void __launch_bounds__(256) vertex_normals_kernel(int nv, int corners, const float* __restrict__ pos, const int* __restrict__ faces,
    const int* __restrict__ offsets, const int* __restrict__ items, float* __restrict__ normals) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  f3 n = mk3(0, 0, 0);
  for (int k = offsets[v]; k < offsets[v + 1]; k++) {
    const int* f = faces + (long long)corners * items[k];
    f3    p0 = ld_f3(pos, f[0]), p1 = ld_f3(pos, f[1]), p2 = ld_f3(pos, f[2]);
    f3    normal;
    float area;
    if (corners == 4) {   // quad_normal / quad_area, yocto_geometry.h:497-518
      f3 p3  = ld_f3(pos, f[3]);
      normal = normalize(triangle_normal(p0, p1, p3) + triangle_normal(p2, p3, p1));
      area   = tri_area(p0, p1, p3) + tri_area(p2, p3, p1);
    } else {
      normal = triangle_normal(p0, p1, p2);
      area   = tri_area(p0, p1, p2);
    }
    n = n + normal * area;
  }
  n = normalize(n);
  normals[3 * v] = n.x, normals[3 * v + 1] = n.y, normals[3 * v + 2] = n.z;
}
__global__ void __launch_bounds__(256) displace_kernel(DScene sc, int nv, int byte_texture, float displacement, const float* __restrict__ pos,
    const float* __restrict__ nrm, const float* __restrict__ uv, float* __restrict__ out) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  f4    texel = eval_texture(sc, 0, mk2(uv[2 * v], uv[2 * v + 1]), true);
  float disp  = (texel.x + texel.y + texel.z) / 3;   // mean(xyz(.)), yocto_pathtrace.cpp:1262
  if (byte_texture) disp -= 0.5f;
  f3 p = ld_f3(pos, v) + ld_f3(nrm, v) * displacement * disp;
  out[3 * v] = p.x, out[3 * v + 1] = p.y, out[3 * v + 2] = p.z;
}

struct device_buffers {   // freed on every path
  std::vector<void*> all;
  ~device_buffers() {
    for (void* p : all) (void)hipFree(p);
  }
  template <typename T>
  hipError_t put(const T* host, size_t count, T** out) {
    void*      d = nullptr;
    hipError_t e = hipMalloc(&d, count ? count * sizeof(T) : 16);
    if (e != hipSuccess) return e;
    all.push_back(d);
    *out = (T*)d;
    return host && count ? hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice) : hipSuccess;
  }
};

template <int D>
int run_level(const vpt_subdiv_level& L, const float* vertices, float* new_vertices) {
  const int nt = L.num_vertices + L.num_edges + L.num_faces;
  device_buffers buf;
  vecD<D> *d_vert = nullptr, *d_tverts = nullptr, *d_out = nullptr;
  int2*   d_edges = nullptr;
  int4 *  d_faces = nullptr, *d_tquads = nullptr;
  int *   d_val = nullptr, *d_off = nullptr, *d_items = nullptr;
#define TRY(expr)                                                                                      \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return vpt_set_error(VPT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
  } while (0)
  TRY(buf.put((const vecD<D>*)vertices, (size_t)L.num_vertices, &d_vert));
  TRY(buf.put((const vecD<D>*)nullptr, (size_t)nt, &d_tverts));
  TRY(buf.put((const vecD<D>*)nullptr, (size_t)nt, &d_out));
  TRY(buf.put((const int2*)L.edges, (size_t)L.num_edges, &d_edges));
  TRY(buf.put((const int4*)L.faces, (size_t)L.num_faces, &d_faces));
  TRY(buf.put((const int4*)L.new_faces, (size_t)L.num_new_faces, &d_tquads));
  TRY(buf.put(L.valence, (size_t)nt, &d_val));
  TRY(buf.put(L.offsets, (size_t)nt + 1, &d_off));
  TRY(buf.put(L.items, (size_t)L.num_items, &d_items));
  const int blocks = (nt + 255) / 256;
  hipLaunchKernelGGL(subdiv_points_kernel<D>, dim3(blocks), dim3(256), 0, 0, L.num_vertices, L.num_edges, L.num_faces, d_vert, d_edges, d_faces, d_tverts);
  hipLaunchKernelGGL(subdiv_average_kernel<D>, dim3(blocks), dim3(256), 0, 0, nt, d_tverts, d_tquads, d_val, d_off, d_items, d_out);
  TRY(hipGetLastError());
  TRY(hipMemcpy(new_vertices, d_out, (size_t)nt * sizeof(vecD<D>), hipMemcpyDeviceToHost));
#undef TRY
  return VPT_OK;
}

}  // namespace

extern "C" int vpt_subdivide_vertices(int device, const vpt_subdiv_level* level, const float* vertices, float* new_vertices) {
  if (!level) return vpt_set_error(VPT_ERR_INVALID_ARG, "null argument");
  const vpt_subdiv_level& L = *level;
  if ((L.dim != 2 && L.dim != 3) || L.num_vertices < 0 || L.num_edges < 0 || L.num_faces < 0 || L.num_new_faces < 0 || L.num_items < 0 ||
      (long long)L.num_vertices + L.num_edges + L.num_faces > (1ll << 30))
    return vpt_set_error(VPT_ERR_INVALID_ARG, "bad level sizes");
  const int nt = L.num_vertices + L.num_edges + L.num_faces;
  if (nt == 0) return VPT_OK;   // an attribute the cage does not have (no texture coordinates): nothing to refine
  if (!vertices || !new_vertices) return vpt_set_error(VPT_ERR_INVALID_ARG, "null argument");
  if (!L.valence || !L.offsets || (L.num_edges && !L.edges) || (L.num_faces && !L.faces) || (L.num_new_faces && !L.new_faces) || (L.num_items && !L.items))
    return vpt_set_error(VPT_ERR_INVALID_ARG, "null table");
  // every index the kernels will follow is checked here: a fault on the device can take the whole node down
  for (int i = 0; i < 2 * L.num_edges; i++)
    if (L.edges[i] < 0 || L.edges[i] >= L.num_vertices) return vpt_set_error(VPT_ERR_INVALID_ARG, "edge %d refers to vertex %d", i / 2, L.edges[i]);
  for (int i = 0; i < 4 * L.num_faces; i++)
    if (L.faces[i] < 0 || L.faces[i] >= L.num_vertices) return vpt_set_error(VPT_ERR_INVALID_ARG, "face %d refers to vertex %d", i / 4, L.faces[i]);
  for (int i = 0; i < 4 * L.num_new_faces; i++)
    if (L.new_faces[i] < 0 || L.new_faces[i] >= nt) return vpt_set_error(VPT_ERR_INVALID_ARG, "new face %d refers to vertex %d", i / 4, L.new_faces[i]);
  if (L.offsets[0] != 0 || L.offsets[nt] != L.num_items) return vpt_set_error(VPT_ERR_INVALID_ARG, "item offsets do not span the item list");
  for (int v = 0; v < nt; v++) {
    const int val = L.valence[v], a = L.offsets[v], b = L.offsets[v + 1];
    if (val < 0 || val > 2 || a > b || (val == 1 && ((b - a) & 1))) return vpt_set_error(VPT_ERR_INVALID_ARG, "bad item range of vertex %d", v);
    for (int k = a; k < b; k++)
      if (L.items[k] < 0 || L.items[k] >= (val == 2 ? L.num_new_faces : nt)) return vpt_set_error(VPT_ERR_INVALID_ARG, "item %d of vertex %d out of range", k - a, v);
  }
  if (hipSetDevice(device) != hipSuccess) return vpt_set_error(VPT_ERR_NO_DEVICE, "no device %d", device);
  return L.dim == 3 ? run_level<3>(L, vertices, new_vertices) : run_level<2>(L, vertices, new_vertices);
}

// quads_normals (corners = 4; a quad with z == w adds to three vertices) / triangles_normals (corners = 3), yocto_shape.cpp:1478-1512
extern "C" int vpt_vertex_normals(int device, int32_t num_vertices, const float* positions, int32_t num_faces, int32_t corners, const int32_t* faces, float* normals) {
  if (num_vertices < 0 || num_faces < 0 || (corners != 3 && corners != 4) || num_vertices > (1 << 28) || num_faces > (1 << 28))
    return vpt_set_error(VPT_ERR_INVALID_ARG, "bad mesh sizes");
  if (num_vertices == 0) return VPT_OK;
  if (!positions || !normals || (num_faces && !faces)) return vpt_set_error(VPT_ERR_INVALID_ARG, "null argument");
  // topology (integers, host): per vertex the faces that add to it - once per corner, in face order, as the reference's loop reaches them
  std::vector<int> offsets((size_t)num_vertices + 1, 0), items;
  auto adds = [&](const int32_t* f, int c) { return !(corners == 4 && c == 3 && f[2] == f[3]); };   // `if (q.z != q.w)` of quads_normals
  for (int i = 0; i < num_faces; i++)
    for (int c = 0; c < corners; c++) {
      int v = faces[(size_t)corners * i + c];
      if (v < 0 || v >= num_vertices) return vpt_set_error(VPT_ERR_INVALID_ARG, "face %d refers to vertex %d", i, v);
      if (adds(faces + (size_t)corners * i, c)) offsets[(size_t)v + 1]++;
    }
  for (int v = 0; v < num_vertices; v++) offsets[(size_t)v + 1] += offsets[(size_t)v];
  items.resize((size_t)offsets[(size_t)num_vertices]);
  std::vector<int> fill(offsets.begin(), offsets.end() - 1);
  for (int i = 0; i < num_faces; i++)
    for (int c = 0; c < corners; c++)
      if (adds(faces + (size_t)corners * i, c)) items[(size_t)fill[(size_t)faces[(size_t)corners * i + c]]++] = i;
  if (hipSetDevice(device) != hipSuccess) return vpt_set_error(VPT_ERR_NO_DEVICE, "no device %d", device);
  device_buffers buf;
  float *d_pos = nullptr, *d_nrm = nullptr;
  int *  d_faces = nullptr, *d_off = nullptr, *d_items = nullptr;
#define TRY(expr)                                                                                      \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return vpt_set_error(VPT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
  } while (0)
  TRY(buf.put(positions, (size_t)num_vertices * 3, &d_pos));
  TRY(buf.put((const float*)nullptr, (size_t)num_vertices * 3, &d_nrm));
  TRY(buf.put((const int*)faces, (size_t)num_faces * corners, &d_faces));
  TRY(buf.put(offsets.data(), offsets.size(), &d_off));
  TRY(buf.put(items.data(), items.size(), &d_items));
  hipLaunchKernelGGL(vertex_normals_kernel, dim3((num_vertices + 255) / 256), dim3(256), 0, 0, num_vertices, corners, d_pos, d_faces, d_off, d_items, d_nrm);
  TRY(hipGetLastError());
  TRY(hipMemcpy(normals, d_nrm, (size_t)num_vertices * 12, hipMemcpyDeviceToHost));
  return VPT_OK;
}

// the displacement step of tesselate_surface (yocto_pathtrace.cpp:1256-1266); texture->offset is ignored: `texels` points at its first texel
extern "C" int vpt_displace_vertices(int device, const vpt_texture* texture, const void* texels, float displacement, int32_t num_vertices,
    const float* positions, const float* normals, const float* texcoords, float* new_positions) {
  if (!texture || num_vertices < 0 || num_vertices > (1 << 28)) return vpt_set_error(VPT_ERR_INVALID_ARG, "bad argument");
  if (num_vertices == 0) return VPT_OK;
  if (!positions || !normals || !texcoords || !new_positions) return vpt_set_error(VPT_ERR_INVALID_ARG, "null argument");
  const long long ntex = (long long)texture->width * texture->height;
  if (texture->width < 0 || texture->height < 0 || ntex > (1ll << 30) || (ntex > 0 && !texels)) return vpt_set_error(VPT_ERR_INVALID_ARG, "bad texture");
  if (hipSetDevice(device) != hipSuccess) return vpt_set_error(VPT_ERR_NO_DEVICE, "no device %d", device);
  device_buffers buf;
  float *d_pos = nullptr, *d_nrm = nullptr, *d_uv = nullptr, *d_out = nullptr, *d_lut = nullptr;
  vpt_texture* d_tex = nullptr;
  float4*      d_texf = nullptr;
  uchar4*      d_texb = nullptr;
  vpt_texture  t = *texture;
  t.offset = 0;
  // sRGB decode table exactly as vpt_scene_create builds it (byte_to_float then srgb_to_rgb, yocto_color.h:212-227, host powf)
  float lut[256];
  for (int b = 0; b < 256; b++) {
    float srgb = b / 255.0f;
    lut[b]     = (srgb <= 0.04045) ? srgb / 12.92f : std::pow((srgb + 0.055f) / (1.0f + 0.055f), 2.4f);
  }
  TRY(buf.put(positions, (size_t)num_vertices * 3, &d_pos));
  TRY(buf.put(normals, (size_t)num_vertices * 3, &d_nrm));
  TRY(buf.put(texcoords, (size_t)num_vertices * 2, &d_uv));
  TRY(buf.put((const float*)nullptr, (size_t)num_vertices * 3, &d_out));
  TRY(buf.put(lut, 256, &d_lut));
  TRY(buf.put(&t, 1, &d_tex));
  if (t.is_float) TRY(buf.put((const float4*)texels, (size_t)ntex, &d_texf));
  else TRY(buf.put((const uchar4*)texels, (size_t)ntex, &d_texb));
  DScene sc = {};   // the one table eval_texture follows: textures[0] over its texel pool and the decode table
  sc.num_textures = 1, sc.textures = d_tex, sc.texels_f = d_texf, sc.texels_b = d_texb, sc.srgb_lut = d_lut;
  hipLaunchKernelGGL(displace_kernel, dim3((num_vertices + 255) / 256), dim3(256), 0, 0, sc, num_vertices, t.is_float ? 0 : 1, displacement, d_pos, d_nrm, d_uv, d_out);
  TRY(hipGetLastError());
  TRY(hipMemcpy(new_positions, d_out, (size_t)num_vertices * 12, hipMemcpyDeviceToHost));
#undef TRY
  return VPT_OK;
}
