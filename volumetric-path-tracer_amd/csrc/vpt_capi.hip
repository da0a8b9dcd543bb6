// vpt_capi.hip — implementation of the C-ABI in include/vpt.h: validation and upload of the
// flattened scene into the device layout of vpt_device.h, kernel launches, state movement.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see __graft_entry__.py).
// There is NO CPU fallback in this library: without a gfx950 device every compute entry point
// fails with VPT_ERR_NO_DEVICE.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <limits>
#include <type_traits>
#include <queue>
#include <functional>
#include <string>
#include <vector>

#include "vpt_implicit_kernel.hip.h"
#include "vpt_kat_kernels.hip.h"
#ifdef VPT_SPLIT_TUS   // the path tracers' instances of K1 are compiled in vpt_k1_volpath.hip / vpt_k1_path.hip
#include "vpt_k1_instances.hip.h"
VPT_K1_SPLIT_INSTANCES(VPT_K1_DECLARE, K_VOLPATH)
VPT_K1_SPLIT_INSTANCES(VPT_K1_DECLARE, K_PATH)
#endif
#include <rocprim/rocprim.hpp>

// light_prims of the single-leaf mesh lights (vpt_device.h): one thread per (light, primitive of the leaf)
__global__ void vpt_light_setup_kernel(DScene sc, float4* out) {
  int l = blockIdx.x, k = threadIdx.x;
  if (l >= sc.num_lights || k >= 4) return;
  float4 r7 = sc.light_rec[8 * l + 7];
  if ((__float_as_int(r7.w) & 255) != VPT_LIGHT_SMALL_MESH || k >= ((__float_as_int(r7.w) >> 8) & 15)) return;
  const DInstance& inst = sc.instances[sc.lights[l].instance];
  const DShape&    sh   = sc.shapes[inst.shape];
  const float4*    leaf = sc.leaf_prims + 4 * ((long long)sh.leaf_offset + ((~sh.root_ref) >> 4) + k);
  f3 n = eval_element_normal(sc, inst, __float_as_int(leaf[0].w));
  for (int c = 0; c < 4; c++) out[20 * l + 5 * k + c] = leaf[c];
  out[20 * l + 5 * k + 4] = make_float4(n.x, n.y, n.z, __int_as_float(sh.is_triangles ? 1 : 0));
}

// search_light_cdf against the plain binary search on the same CDF: values at, just below and just above CDF
// entries, uniform ones, and the ends of the range; out[0] = mismatches
__global__ void vpt_light_cdf_selftest_kernel(DScene sc, int light_id, int n, unsigned long long* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const vpt_light& light = sc.lights[light_id];
  const float*     cdf   = sc.light_cdf + light.cdf_offset;
  const int        len   = light.cdf_len;
  float back = cdf[len - 1];
  // the wave votes inside search_light_cdf: keep all lanes in, flag the surplus ones instead of returning
  bool     live = i < n;
  unsigned h    = (unsigned)i * 2654435761u + 12345u;
  h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
  float v = cdf[h % (unsigned)len];
  switch (i & 7) {
    case 0: break;
    case 1: v = __uint_as_float(__float_as_uint(v) - (v > 0 ? 1u : 0u)); break;
    case 2: v = __uint_as_float(__float_as_uint(v) + 1u); break;
    case 3: v = 0.0f; break;
    case 4: v = back; break;
    default: v = back * ((h >> 8) * (1.0f / 16777216.0f)); break;
  }
  float r = clampf(v, 0.0f, back - 0.00001f);
  int a = search_light_cdf(sc, light_id, r);
  int lo = 0, cnt = len;   // std::upper_bound, as sample_discrete
  while (cnt > 0) {
    int half = cnt >> 1;
    if (!(r < cdf[lo + half])) lo += half + 1, cnt -= half + 1;
    else cnt = half;
  }
  int b = lo < len ? lo : len - 1;
  if (live && a != b) atomicAdd(&out[0], 1ull);
}

// Launch schedule: a wave's duration varies by +-17 % from one launch to the next on the same tile (it depends on which waves shared its SIMD:
// profiles/r04_k2_lane_histogram.txt), and longest-first scheduling on such estimates ends well above its bound (K2: 225 ms against 203).  The order is
// therefore taken from a running average of the duration PER SAMPLE (weight = samples seen, capped), whose bit pattern - positive floats - is the sort key.
__global__ void vpt_cost_average_kernel(const unsigned* __restrict__ cost, float* __restrict__ avg, unsigned* __restrict__ key, int n, float nsamples, float weight) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float per_sample = (float)cost[i] / nsamples;
  float a = weight > 0 ? (avg[i] * weight + per_sample * nsamples) / (weight + nsamples) : per_sample;
  avg[i] = a;
  key[i] = __float_as_uint(a);
}

// vpt_intersect: one lane per ray through the production traversal (COMPACT: the leaf records of a scene of triangles, as the path tracers read them)
template <bool SPILL, bool COMPACT>
__global__ void __launch_bounds__(VPT_BLOCK, VPT_WAVES_PER_SIMD) vpt_intersect_kernel(DScene sc, int n, const float* rays, int instance,
    int* ids, float* uvt, stack_cfg stack) {
  extern __shared__ int lds_stack[];
  const lane_stack2<SPILL> stk = make_lane_stack<SPILL>(lds_stack, stack);
  int i = blockIdx.x * VPT_BLOCK + threadIdx.x;
  const bool live = i < n;   // the whole wave goes through the query (traverse(): the group forms need every lane); surplus lanes carry no ray
  if (!live) i = 0;
  hit_t h = traverse<COMPACT>(sc, live, mk3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]), instance, stk);
  if (!live) return;
  ids[2 * i] = h.hit ? h.instance : -1, ids[2 * i + 1] = h.hit ? h.element : -1;
  uvt[3 * i] = h.hit ? h.uv.x : 0, uvt[3 * i + 1] = h.hit ? h.uv.y : 0, uvt[3 * i + 2] = h.hit ? h.distance : 0;
}

// all 2^32 operands of rcp_newton (vpt_mesh_kernel.hip.h) against the IEEE quotient; out[0] = mismatches, out[1] = out of range
__global__ void vpt_reciprocal_selftest_kernel(unsigned long long* out) {
  unsigned long long bad = 0, skipped = 0;
  for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += (unsigned long long)gridDim.x * blockDim.x) {
    float x = __uint_as_float((unsigned)b), a = __builtin_fabsf(x);
    if (!rcp_in_range(a, a)) { skipped++; continue; }
    if (__float_as_uint(rcp_newton(x)) != __float_as_uint(1.0f / x)) bad++;
  }
  if (bad) atomicAdd(&out[0], bad);
  if (skipped) atomicAdd(&out[1], skipped);
}

static std::string& g_error_text() {   // the message of the last failure on the calling thread (vpt_last_error)
  thread_local std::string text;
  return text;
}

namespace {

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error_text() = buf;
  return code;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(VPT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));        \
  } while (0)

// host-side float3 helpers for the load-time precomputation (same formulas as the reference)
struct h3 { float x, y, z; };
h3 hcross(h3 a, h3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
float hdot(h3 a, h3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
h3 hmul(h3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
h3 hadd(h3 a, h3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
struct hframe { h3 x, y, z, o; };
hframe to_h(const vpt_frame& f) { return {{f.x[0], f.x[1], f.x[2]}, {f.y[0], f.y[1], f.y[2]}, {f.z[0], f.z[1], f.z[2]}, {f.o[0], f.o[1], f.o[2]}}; }
// inverse(frame3f, non_rigid), yocto_math.h:2948-2956 with inverse(mat3f) = adjoint * (1/det), :2802-2808
hframe hinverse(const hframe& a, bool non_rigid) {
  h3 mx, my, mz;
  if (non_rigid) {
    h3 c0 = hcross(a.y, a.z), c1 = hcross(a.z, a.x), c2 = hcross(a.x, a.y);   // adjoint = transpose{c0,c1,c2}
    float s = 1 / hdot(a.x, hcross(a.y, a.z));
    mx = hmul({c0.x, c1.x, c2.x}, s), my = hmul({c0.y, c1.y, c2.y}, s), mz = hmul({c0.z, c1.z, c2.z}, s);
  } else {
    mx = {a.x.x, a.y.x, a.z.x}, my = {a.x.y, a.y.y, a.z.y}, mz = {a.x.z, a.y.z, a.z.z};
  }
  h3 mo = hadd(hadd(hmul(mx, a.o.x), hmul(my, a.o.y)), hmul(mz, a.o.z));
  return {mx, my, mz, {-mo.x, -mo.y, -mo.z}};
}
void pack_frame(const hframe& f, float4* out) {
  out[0] = make_float4(f.x.x, f.x.y, f.x.z, f.y.x);
  out[1] = make_float4(f.y.y, f.y.z, f.z.x, f.z.y);
  out[2] = make_float4(f.z.z, f.o.x, f.o.y, f.o.z);
}

// Quad nodes (vpt_device.h): one 128-byte record per internal node at even depth, holding the boxes and
// references of its (up to) four grandchildren in the binary BVH: slots 0,1 = children of child 0 (or
// child 0 itself when it is a leaf, slot 1 empty), slots 2,3 likewise for child 1.  Layout: lo.x[4],
// lo.y[4], lo.z[4], hi.x[4], hi.y[4], hi.z[4], ref[4], {axis | axis0 << 2 | axis1 << 4, 0, 0, 0}.
// ref >= 0: quad node (relative to this BVH), ~ref = start << 4 | count: leaf, VPT_NONE_REF: empty slot.
// Returns the reference of the root and its box, appends to `out`; *need = worst-case number of stack
// entries a traversal of this BVH holds at once (three pending siblings per quad level).
constexpr int VPT_NONE_REF = -2147483647 - 1;
int build_quad_nodes(const vpt_bvh_node* nodes, int count, std::vector<float4>& out, float root_box[6], int* need) {
  for (int k = 0; k < 6; k++) root_box[k] = 0;
  *need = 0;
  if (count <= 0) return ~0;   // empty leaf
  for (int k = 0; k < 3; k++) root_box[k] = nodes[0].bbox_min[k], root_box[3 + k] = nodes[0].bbox_max[k];
  auto leaf_code = [&](int i) { return ~((nodes[i].start << 4) | (nodes[i].num & 15)); };
  if (!nodes[0].internal) return leaf_code(0);
  // binary nodes that become quad nodes, in depth-first preorder (a node's subtree stays close to it)
  std::vector<int> quad_of((size_t)count, -1), order, todo{0};
  auto slots_of = [&](int i, int slot[4], int axes[3]) {
    axes[0] = nodes[i].axis, axes[1] = axes[2] = 0;
    for (int side = 0; side < 2; side++) {
      int c = nodes[i].start + side;
      if (nodes[c].internal) slot[2 * side] = nodes[c].start, slot[2 * side + 1] = nodes[c].start + 1, axes[1 + side] = nodes[c].axis;
      else slot[2 * side] = c, slot[2 * side + 1] = -1;
    }
  };
  while (!todo.empty()) {
    int i = todo.back();
    todo.pop_back();
    quad_of[(size_t)i] = (int)order.size();
    order.push_back(i);
    int slot[4], axes[3];
    slots_of(i, slot, axes);
    for (int k = 3; k >= 0; k--)
      if (slot[k] >= 0 && nodes[slot[k]].internal) todo.push_back(slot[k]);
  }
  size_t base = out.size();
  out.resize(base + 8 * order.size());
  std::vector<int> node_need(order.size(), 0);
  for (size_t n = order.size(); n-- > 0;) {   // children come after their parent in preorder: fill bottom-up
    int i = order[n], slot[4], axes[3];
    slots_of(i, slot, axes);
    float box[6][4];
    int   ref[4], present = 0, deepest = 0;
    for (int k = 0; k < 4; k++) {
      for (int c = 0; c < 6; c++) box[c][k] = 0;
      ref[k] = VPT_NONE_REF;
      if (slot[k] < 0) continue;
      const vpt_bvh_node& ch = nodes[slot[k]];
      for (int c = 0; c < 3; c++) box[c][k] = ch.bbox_min[c], box[3 + c][k] = ch.bbox_max[c];
      ref[k] = ch.internal ? quad_of[(size_t)slot[k]] : leaf_code(slot[k]);
      present++;
      if (ch.internal && node_need[(size_t)quad_of[(size_t)slot[k]]] > deepest) deepest = node_need[(size_t)quad_of[(size_t)slot[k]]];
    }
    node_need[n] = present - 1 + deepest;
    float4* q = &out[base + 8 * n];
    for (int c = 0; c < 6; c++) q[c] = make_float4(box[c][0], box[c][1], box[c][2], box[c][3]);
    memcpy(&q[6], ref, 16);
    int meta[4] = {axes[0] | (axes[1] << 2) | (axes[2] << 4), 0, 0, 0};
    memcpy(&q[7], meta, 16);
  }
  *need = node_need[0];
  return 0;
}

int bvh_depth(const vpt_bvh_node* nodes, int count, int root, int depth, int limit) {
  if (depth > limit) return depth;
  const vpt_bvh_node& n = nodes[root];
  if (!n.internal) return depth;
  int a = bvh_depth(nodes, count, n.start, depth + 1, limit), b = bvh_depth(nodes, count, n.start + 1, depth + 1, limit);
  return a > b ? a : b;
}

}  // namespace

// the same for the other translation units of the library (vpt_multi.cpp)
int vpt_set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error_text() = buf;
  return code;
}

struct vpt_scene {
  int                device = 0;
  DScene             d      = {};
  std::vector<void*> allocs;
  int                stack_cap = 16;    // binary-BVH walk of the implicit kernels' mesh-light pdf: refs only
  int                stack_lds4 = 8, stack_spill4 = 0;   // quad-node traversal: (ref, t0) entries in LDS / in HBM
  void*              spill = nullptr;
  long long          spill_lanes = 0;
  // launch schedule of the mesh kernel (sched_cfg): per-wave cost of the last launch, waves by descending cost
  unsigned *d_cost = nullptr, *d_cost_sorted = nullptr, *d_cost_key = nullptr;
  float*    d_cost_avg = nullptr;      // running average of a wave's duration per sample (vpt_cost_average_kernel)
  float     cost_weight = 0;           // samples behind that average (0: none yet)
  int *     d_order = nullptr, *d_iota = nullptr;
  hipEvent_t  ev_order = nullptr;       // recorded after the sort that writes d_order
  hipStream_t order_stream = nullptr;   // the stream that sort ran on

  void*     sort_temp = nullptr;
  size_t    sort_temp_bytes = 0;
  // tile splitting (launch_mesh): tiles whose pixels run as 2^k partly filled waves, so that a launch is not as long as its costliest tile
  int*      d_lane_slot = nullptr;
  long long lane_cap = 0;
  int       split_waves = 0, split_tiles = 0;   // waves of the split launch (0: no table), tiles that were split
  std::vector<int> h_split_k;                   // per tile: it runs as 2^k waves
  bool      full_costs = false;                 // d_cost holds per-tile durations of an unsplit launch over >= 8 samples
  int       wave_slots_k1 = 3072;               // wave slots of the chip for K1 (CUs x 4 SIMDs x 3)
  int       wave_slots_k2 = 5120;               // ... for K2 (x VPT_K2_WAVES)
  bool      split_decided = false;              // the decision for sched_key has been taken (costs of an unsplit launch were available)
  int       last_waves = 0;                     // grid of the last kernel launch (vpt_last_wave_costs)
  long long sched_waves = 0;       // waves the buffers are sized for
  bool      order_valid = false;   // d_order describes the layout of sched_key
  long long sched_key[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // staging for the host-state entry point vpt_render()
  void *s_image = nullptr, *s_hits = nullptr, *s_rng = nullptr;   // tile-major state
  void *r_image = nullptr, *r_hits = nullptr, *r_rng = nullptr;   // row-major mirror
  long long  staged_pixels = 0, staged_slots = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_host0 = nullptr, ev_host1 = nullptr;   // around a host-side pause inside a call (decide_split): not kernel time
  bool       host_pause = false;                       // the last call recorded that pair
  bool       timed = false;
  unsigned*  d_watchdog = nullptr;   // waves of the implicit kernel that gave up (must stay 0; vpt_implicit_kernel.hip.h)
  bool       large_mesh_lights = false;
  int        light_features = 0;      // VPT_FEAT_* bits this scene's lights need from the mesh kernels
  // host mirrors of a few index tables: range checks of the batch entry points (vpt_intersect, vpt_kat)
  std::vector<int> h_slot_of;                          // instance -> scene-BVH primitive slot (-1: not in the scene BVH)
  std::vector<int> h_inst_shape, h_shape_elems, h_shape_elem_offset;
  std::vector<int> h_prim_slot;                        // [shape elem_offset + element] -> slot in leaf_prims / leaf_attrs
};

namespace {

template <typename T>
int upload(vpt_scene* s, const std::vector<T>& host, const T** out) {
  *out = nullptr;
  size_t bytes = host.size() * sizeof(T);
  void*  p     = nullptr;
  HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));   // never hand the kernel a null table
  s->allocs.push_back(p);
  if (bytes) HIP_TRY(hipMemcpy(p, host.data(), bytes, hipMemcpyHostToDevice));
  *out = (const T*)p;
  return VPT_OK;
}
template <typename T>
int upload(vpt_scene* s, const T* host, long long count, const T** out) {
  return upload(s, std::vector<T>(host, host + (host ? count : 0)), out);
}

#define REQUIRE(cond, ...)                                      \
  do {                                                          \
    if (!(cond)) return fail(VPT_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

int check_nodes(const vpt_bvh_node* nodes, long long count, long long nprims, const char* what) {
  for (long long i = 0; i < count; i++) {
    const vpt_bvh_node& n = nodes[i];
    if (n.internal) REQUIRE(n.start > i && (long long)n.start + 1 < count, "%s bvh node %lld: bad children", what, i);
    else REQUIRE(n.start >= 0 && n.num >= 0 && n.num <= 15 && n.start < (1 << 27) && (long long)n.start + n.num <= nprims, "%s bvh node %lld: bad leaf range", what, i);
    REQUIRE(n.axis >= 0 && n.axis <= 2, "%s bvh node %lld: bad axis", what, i);
  }
  return VPT_OK;
}

int validate(const vpt_scene_desc& d) {
  REQUIRE(d.num_cameras > 0 && d.cameras, "scene has no cameras");
#define TABLE(n, p) REQUIRE((n) >= 0 && ((n) == 0 || (p) != nullptr), "table %s is null", #p)
  TABLE(d.num_instances, d.instances); TABLE(d.num_shapes, d.shapes); TABLE(d.num_materials, d.materials);
  TABLE(d.num_textures, d.textures); TABLE(d.num_environments, d.environments); TABLE(d.num_volumes, d.volumes);
  TABLE(d.num_vol_instances, d.vol_instances); TABLE(d.num_sdfs, d.sdfs); TABLE(d.num_lights, d.lights);
  TABLE(d.num_positions, d.positions); TABLE(d.num_normals, d.normals); TABLE(d.num_texcoords, d.texcoords);
  TABLE(d.num_colors, d.colors); TABLE(d.num_triangles, d.triangles); TABLE(d.num_quads, d.quads);
  TABLE(d.num_texels_f, d.texels_f); TABLE(d.num_texels_b, d.texels_b); TABLE(d.num_voxels, d.voxels);
  TABLE(d.num_light_cdf, d.light_cdf); TABLE(d.num_scene_bvh_nodes, d.scene_bvh_nodes);
  TABLE(d.num_scene_bvh_prims, d.scene_bvh_prims); TABLE(d.num_shape_bvh_nodes, d.shape_bvh_nodes);
  TABLE(d.num_shape_bvh_prims, d.shape_bvh_prims);
#undef TABLE
  auto tex_ok = [&](int t) { return t >= -1 && t < d.num_textures; };
  for (int i = 0; i < d.num_shapes; i++) {
    const vpt_shape& s = d.shapes[i];
    REQUIRE(s.num_vertices >= 0 && s.position_offset >= 0 && (long long)s.position_offset + s.num_vertices <= d.num_positions, "shape %d: positions out of range", i);
    REQUIRE(s.normal_offset == -1 || (s.normal_offset >= 0 && (long long)s.normal_offset + s.num_vertices <= d.num_normals), "shape %d: normals out of range", i);
    REQUIRE(s.texcoord_offset == -1 || (s.texcoord_offset >= 0 && (long long)s.texcoord_offset + s.num_vertices <= d.num_texcoords), "shape %d: texcoords out of range", i);
    REQUIRE(s.color_offset == -1 || (s.color_offset >= 0 && (long long)s.color_offset + s.num_vertices <= d.num_colors), "shape %d: colors out of range", i);
    REQUIRE(s.num_triangles >= 0 && s.triangle_offset >= 0 && (long long)s.triangle_offset + s.num_triangles <= d.num_triangles, "shape %d: triangles out of range", i);
    REQUIRE(s.num_quads >= 0 && s.quad_offset >= 0 && (long long)s.quad_offset + s.num_quads <= d.num_quads, "shape %d: quads out of range", i);
    REQUIRE(s.num_triangles == 0 || s.num_quads == 0, "shape %d: both triangles and quads", i);
    long long nel = s.num_triangles ? s.num_triangles : s.num_quads;
    for (long long k = 0; k < 3LL * s.num_triangles; k++) {
      int v = d.triangles[3LL * s.triangle_offset + k];
      REQUIRE(v >= 0 && v < s.num_vertices, "shape %d: triangle vertex index out of range", i);
    }
    for (long long k = 0; k < 4LL * s.num_quads; k++) {
      int v = d.quads[4LL * s.quad_offset + k];
      REQUIRE(v >= 0 && v < s.num_vertices, "shape %d: quad vertex index out of range", i);
    }
    REQUIRE(s.num_bvh_nodes >= 0 && s.bvh_node_offset >= 0 && (long long)s.bvh_node_offset + s.num_bvh_nodes <= d.num_shape_bvh_nodes, "shape %d: bvh nodes out of range", i);
    REQUIRE(s.bvh_prim_offset >= 0 && (long long)s.bvh_prim_offset + nel <= d.num_shape_bvh_prims, "shape %d: bvh prims out of range", i);
    if (int rc = check_nodes(d.shape_bvh_nodes + s.bvh_node_offset, s.num_bvh_nodes, nel, "shape")) return rc;
    for (long long k = 0; k < nel; k++) {
      int e = d.shape_bvh_prims[s.bvh_prim_offset + k];
      REQUIRE(e >= 0 && e < nel, "shape %d: bvh primitive id out of range", i);
    }
  }
  for (int i = 0; i < d.num_instances; i++) {
    REQUIRE(d.instances[i].shape >= 0 && d.instances[i].shape < d.num_shapes, "instance %d: bad shape", i);
    REQUIRE(d.instances[i].material >= 0 && d.instances[i].material < d.num_materials, "instance %d: bad material", i);
  }
  // Texture ids are only dereferenced for materials bound to mesh instances (eval_material with
  // texcoords, yocto_scene.cpp:529); materials used only by SDFs / voxel grids go through the
  // texture-free eval_material(scene,int) (:581) and the reference tolerates dangling ids there
  // (tests/06_gridsdf ships some), so range-check only what the device can read.
  std::vector<char> textured((size_t)d.num_materials, 0);
  for (int i = 0; i < d.num_instances; i++) textured[(size_t)d.instances[i].material] = 1;
  for (int i = 0; i < d.num_materials; i++) {
    const vpt_material& m = d.materials[i];
    REQUIRE(m.type >= 0 && m.type <= VPT_MAT_GLTFPBR, "material %d: bad type", i);
    if (!textured[(size_t)i]) continue;
    REQUIRE(tex_ok(m.emission_tex) && tex_ok(m.color_tex) && tex_ok(m.roughness_tex) && tex_ok(m.scattering_tex) && tex_ok(m.normal_tex), "material %d: texture id out of range", i);
  }
  for (int i = 0; i < d.num_textures; i++) {
    const vpt_texture& t = d.textures[i];
    long long n = (long long)t.width * t.height;
    REQUIRE(t.width >= 0 && t.height >= 0 && t.offset >= 0 && t.offset + n <= (t.is_float ? d.num_texels_f : d.num_texels_b), "texture %d: texels out of range", i);
  }
  for (int i = 0; i < d.num_environments; i++) REQUIRE(tex_ok(d.environments[i].emission_tex), "environment %d: bad texture", i);
  for (int i = 0; i < d.num_volumes; i++) {
    const vpt_volume& v = d.volumes[i];
    REQUIRE(v.whd[0] >= 0 && v.whd[1] >= 0 && v.whd[2] >= 0 && v.offset >= 0 && v.offset + (long long)v.whd[0] * v.whd[1] * v.whd[2] <= d.num_voxels, "volume %d: voxels out of range", i);
    REQUIRE((long long)v.whd[0] * v.whd[1] * v.whd[2] < (1ll << 31), "volume %d: 2^31 voxels or more", i);   // eval_volume indexes a volume with 32-bit arithmetic
  }
  for (int i = 0; i < d.num_vol_instances; i++) {
    REQUIRE(d.vol_instances[i].volume >= 0 && d.vol_instances[i].volume < d.num_volumes, "vol_instance %d: bad volume", i);
    REQUIRE(d.vol_instances[i].material >= 0 && d.vol_instances[i].material < d.num_materials, "vol_instance %d: bad material", i);
  }
  for (int i = 0; i < d.num_sdfs; i++) {
    REQUIRE(d.sdfs[i].type >= 0 && d.sdfs[i].type <= VPT_SDF_TORUS, "sdf %d: bad type", i);
    REQUIRE(d.sdfs[i].material >= 0 && d.sdfs[i].material < d.num_materials, "sdf %d: bad material", i);
  }
  for (int i = 0; i < d.num_lights; i++) {
    const vpt_light& l = d.lights[i];
    REQUIRE(l.instance >= -1 && l.instance < d.num_instances && l.environment >= -1 && l.environment < d.num_environments && l.sdf >= -1 && l.sdf < d.num_sdfs, "light %d: bad reference", i);
    REQUIRE(l.cdf_len >= 0 && l.cdf_offset >= 0 && l.cdf_offset + l.cdf_len <= d.num_light_cdf, "light %d: cdf out of range", i);
    if (l.instance >= 0) {
      const vpt_shape& s = d.shapes[d.instances[l.instance].shape];
      REQUIRE(l.cdf_len == (s.num_triangles ? s.num_triangles : s.num_quads) && l.cdf_len > 0, "light %d: cdf length != element count", i);
    } else if (l.sdf >= 0) {
      REQUIRE(l.cdf_len == 1, "light %d: sdf light needs a 1-entry cdf", i);
    } else if (l.environment >= 0 && d.environments[l.environment].emission_tex >= 0) {
      const vpt_texture& t = d.textures[d.environments[l.environment].emission_tex];
      REQUIRE(l.cdf_len == t.width * t.height && l.cdf_len > 0, "light %d: cdf length != texel count", i);
    }
  }
  if (int rc = check_nodes(d.scene_bvh_nodes, d.num_scene_bvh_nodes, d.num_scene_bvh_prims, "scene")) return rc;
  for (int i = 0; i < d.num_scene_bvh_prims; i++) REQUIRE(d.scene_bvh_prims[i] >= 0 && d.scene_bvh_prims[i] < d.num_instances, "scene bvh: bad instance id");
  {   // the single-instance query of the mesh-light pdf walk enters an instance through its scene-BVH slot
    std::vector<char> in_bvh((size_t)d.num_instances, 0);
    for (int i = 0; i < d.num_scene_bvh_prims; i++) in_bvh[(size_t)d.scene_bvh_prims[i]] = 1;
    for (int i = 0; i < d.num_lights; i++)
      if (d.lights[i].instance >= 0) REQUIRE(in_bvh[(size_t)d.lights[i].instance], "light %d: its instance is not in the scene bvh", i);
  }
  return VPT_OK;
}

int make_dparams(const vpt_params* p, const vpt_layout* l, int nsamples, DParams& out) {
  REQUIRE(p && l, "null params/layout");
  REQUIRE(l->width > 0 && l->height > 0 && l->nranks > 0 && l->rank >= 0 && l->rank < l->nranks, "bad layout");
  REQUIRE(l->width < 32768 && l->height < 32768, "frame side must be below 32768 pixels (the kernels keep a pixel's coordinates in one word; the reference's --resolution ends at 4096)");
  REQUIRE(l->tile_w >= 8 && l->tile_h >= 8 && l->tile_w % 8 == 0 && l->tile_h % 8 == 0, "tile size must be a multiple of 8x8");
  out = {};
  out.camera = p->camera, out.shader = p->shader, out.bounces = p->bounces, out.noimplicit_mis = p->noimplicit_mis;
  out.spheretrace_maxiter = p->spheretrace_maxiter, out.preview = p->samples == 1, out.nsamples = nsamples;
  out.width = l->width, out.height = l->height, out.tile_w = l->tile_w, out.tile_h = l->tile_h;
  out.tiles_x = (l->width + l->tile_w - 1) / l->tile_w, out.tiles_y = (l->height + l->tile_h - 1) / l->tile_h;
  out.rank = l->rank, out.nranks = l->nranks;
  long long tiles = (long long)out.tiles_x * out.tiles_y;
  long long local = (tiles + l->nranks - 1) / l->nranks;   // every rank allocates the same count
  long long slots = local * l->tile_w * l->tile_h;
  REQUIRE(slots < (1LL << 31), "image too large");
  out.nslots = (int)slots;
  return VPT_OK;
}

}  // namespace

extern "C" {

const char* vpt_last_error(void) { return g_error_text().c_str(); }
const char* vpt_version(void) { return "vpt-mi355x 0.1 (gfx950)"; }

int vpt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void vpt_scene_destroy(vpt_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  for (void* p : s->allocs) (void)hipFree(p);
  for (void* p : {s->s_image, s->s_hits, s->s_rng, s->r_image, s->r_hits, s->r_rng})
    if (p) (void)hipFree(p);
  if (s->spill) (void)hipFree(s->spill);
  for (void* p : {(void*)s->d_cost, (void*)s->d_cost_sorted, (void*)s->d_cost_key, (void*)s->d_cost_avg, (void*)s->d_order, (void*)s->d_iota, s->sort_temp, (void*)s->d_lane_slot})
    if (p) (void)hipFree(p);
  if (s->ev_order) (void)hipEventDestroy(s->ev_order);
  if (s->d_watchdog) (void)hipFree(s->d_watchdog);
  for (hipEvent_t e : {s->ev0, s->ev1, s->ev_host0, s->ev_host1})
    if (e) (void)hipEventDestroy(e);
  delete s;
}

int vpt_scene_create(const vpt_scene_desc* desc, int device, vpt_scene** out) {
  if (!desc || !out) return fail(VPT_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (int rc = validate(*desc)) return rc;
  int ndev = vpt_device_count();
  if (ndev <= 0) return fail(VPT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(VPT_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
  (void)hipGetLastError();   // start from a clean slate: an error left behind by an unrelated earlier call is not this call's
  HIP_TRY(hipSetDevice(device));
  const vpt_scene_desc& d = *desc;
  vpt_scene* s = new vpt_scene{};
  s->device    = device;
  struct guard { vpt_scene*& s; ~guard() { if (s) vpt_scene_destroy(s); } } g{s};
  DScene& D = s->d;
  D.num_cameras = d.num_cameras, D.num_instances = d.num_instances, D.num_shapes = d.num_shapes;
  D.num_materials = d.num_materials, D.num_textures = d.num_textures, D.num_environments = d.num_environments;
  D.num_volumes = d.num_volumes, D.num_vol_instances = d.num_vol_instances, D.num_sdfs = d.num_sdfs;
  D.num_lights = d.num_lights, D.num_scene_nodes = d.num_scene_bvh_nodes, D.num_scene_prims = d.num_scene_bvh_prims;
  D.group_forms = getenv("VPT_NO_GROUP_FORMS") ? 0 : 1;   // A/B switch of the tests: the two forms of a phase must give the same bits

  // --- geometry pools in device layout ---------------------------------------------------------
  std::vector<float4> positions((size_t)d.num_positions), normals((size_t)d.num_normals), colors((size_t)d.num_colors);
  std::vector<float2> texcoords((size_t)d.num_texcoords);
  for (long long i = 0; i < d.num_positions; i++) positions[i] = make_float4(d.positions[3 * i], d.positions[3 * i + 1], d.positions[3 * i + 2], 0);
  for (long long i = 0; i < d.num_normals; i++) normals[i] = make_float4(d.normals[3 * i], d.normals[3 * i + 1], d.normals[3 * i + 2], 0);
  for (long long i = 0; i < d.num_colors; i++) colors[i] = make_float4(d.colors[4 * i], d.colors[4 * i + 1], d.colors[4 * i + 2], d.colors[4 * i + 3]);
  for (long long i = 0; i < d.num_texcoords; i++) texcoords[i] = make_float2(d.texcoords[2 * i], d.texcoords[2 * i + 1]);

  std::vector<DShape> shapes((size_t)d.num_shapes);
  std::vector<int4>   elems;
  std::vector<float4> leafs, leaf_attrs, shape_wnodes, scene_wnodes;
  int max_shape_depth = 0, max_shape_need4 = 0;
  for (int i = 0; i < d.num_shapes; i++) {
    const vpt_shape& sh = d.shapes[i];
    DShape& o = shapes[i];
    o = {};
    o.num_nodes = sh.num_bvh_nodes, o.node_offset = sh.bvh_node_offset;
    o.is_triangles = sh.num_triangles != 0;
    o.num_elems = o.is_triangles ? sh.num_triangles : sh.num_quads;
    o.elem_offset = (int)elems.size(), o.leaf_offset = (int)(leafs.size() / 4);
    o.vertex_offset = sh.position_offset, o.normal_offset = sh.normal_offset;
    o.texcoord_offset = sh.texcoord_offset, o.color_offset = sh.color_offset;
    for (int e = 0; e < o.num_elems; e++) {
      if (o.is_triangles) {
        const int32_t* t = d.triangles + 3LL * (sh.triangle_offset + e);
        elems.push_back(make_int4(t[0], t[1], t[2], t[2]));
      } else {
        const int32_t* q = d.quads + 4LL * (sh.quad_offset + e);
        elems.push_back(make_int4(q[0], q[1], q[2], q[3]));
      }
    }
    s->h_shape_elems.push_back(o.num_elems), s->h_shape_elem_offset.push_back(o.elem_offset);
    s->h_prim_slot.resize(elems.size(), -1);
    // leaf records in BVH primitive order: slot k holds element prims[k]'s corners
    for (int k = 0; k < o.num_elems; k++) {
      int  e = d.shape_bvh_prims[sh.bvh_prim_offset + k];
      s->h_prim_slot[(size_t)o.elem_offset + e] = o.leaf_offset + k;
      int4 q = elems[(size_t)o.elem_offset + e];
      for (int c = 0; c < 4; c++) {
        int    v = c == 0 ? q.x : c == 1 ? q.y : c == 2 ? q.z : q.w;
        float4 p = positions[(size_t)sh.position_offset + v];
        int    tag = c == 0 ? e : 0;
        memcpy(&p.w, &tag, 4);
        leafs.push_back(p);
      }
      // the corners' normals, then their texcoords (zeros where the shape has none: never read then)
      float tc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int c = 0; c < 4; c++) {
        int v = c == 0 ? q.x : c == 1 ? q.y : c == 2 ? q.z : q.w;
        leaf_attrs.push_back(sh.normal_offset >= 0 ? normals[(size_t)sh.normal_offset + v] : make_float4(0, 0, 0, 0));
        if (sh.texcoord_offset >= 0) tc[2 * c] = texcoords[(size_t)sh.texcoord_offset + v].x, tc[2 * c + 1] = texcoords[(size_t)sh.texcoord_offset + v].y;
      }
      leaf_attrs.push_back(make_float4(tc[0], tc[1], tc[2], tc[3]));
      leaf_attrs.push_back(make_float4(tc[4], tc[5], tc[6], tc[7]));
    }
    o.wnode_offset = (int)(shape_wnodes.size() / 8);
    int need4 = 0;
    o.root_ref     = build_quad_nodes(d.shape_bvh_nodes + sh.bvh_node_offset, sh.num_bvh_nodes, shape_wnodes, o.root_box, &need4);
    int depth = o.num_nodes ? bvh_depth(d.shape_bvh_nodes + sh.bvh_node_offset, sh.num_bvh_nodes, 0, 0, 4096) : 0;
    o.stack_need = depth + 2;
    if (depth > max_shape_depth) max_shape_depth = depth;
    if (need4 > max_shape_need4) max_shape_need4 = need4;
  }
  int scene_depth = d.num_scene_bvh_nodes ? bvh_depth(d.scene_bvh_nodes, d.num_scene_bvh_nodes, 0, 0, 4096) : 0;
  float scene_box[6];
  int scene_need4 = 0;
  D.scene_root_ref = build_quad_nodes(d.scene_bvh_nodes, d.num_scene_bvh_nodes, scene_wnodes, scene_box, &scene_need4);
  D.scene_root_lo_x = scene_box[0], D.scene_root_lo_y = scene_box[1], D.scene_root_lo_z = scene_box[2];
  D.scene_root_hi_x = scene_box[3], D.scene_root_hi_y = scene_box[4], D.scene_root_hi_z = scene_box[5];
  // stack entries alive at once: one pending sibling per level (+ the two just pushed), scene level
  // entries stay below the entries of the instance being traversed
  int need = (scene_depth + 2) + (max_shape_depth + 2);
  s->stack_cap = ((need > 8 ? need : 8) + 3) & ~3;
  if ((size_t)s->stack_cap * VPT_BLOCK * sizeof(int) > 64 * 1024)
    return fail(VPT_ERR_UNSUPPORTED, "BVH depth %d needs a %d-entry traversal stack; the LDS stack holds 64", need, s->stack_cap);
  // quad-node traversal: worst case = three pending siblings per quad level of the scene BVH plus of the
  // deepest shape BVH, plus one free entry above the top (the branch-free push stores rejected candidates
  // there).  24 entries per lane = 12 KB per wave keep twelve waves on a CU (144 of 160 KB); whatever the
  // worst case needs beyond that lives in HBM (lane_stack2<true>).
  int need4 = scene_need4 + max_shape_need4 + 1;
  // With the mesh kernel's five parked words per lane (vpt_mesh_kernel.hip.h) a wave takes need4 * 512 + 1280 + 8 bytes of LDS,
  // granted in 1 280-byte steps: up to 22 entries twelve waves fit a CU's 160 KB, with 23 or 24 eleven do - still better than the
  // checked push / pop of the HBM-overflow variant (-7 %), which is for deeper trees only (22 entries in LDS, the rest in HBM).
  s->stack_lds4   = need4 < 8 ? 8 : need4 > 24 ? 22 : need4;
  if (const char* e = getenv("VPT_STACK_LDS")) {   // tuning experiments: force a smaller LDS part (the rest spills to HBM)
    int v = atoi(e);
    if (v >= 4 && v < s->stack_lds4) s->stack_lds4 = v;
  }
  s->stack_spill4 = need4 > s->stack_lds4 ? need4 - s->stack_lds4 : 0;
  if (getenv("VPT_DEBUG"))
    fprintf(stderr, "[vpt] binary depth scene %d shape %d; quad stack need scene %d + shape %d + 1 -> %d in LDS + %d in HBM\n",
        scene_depth, max_shape_depth, scene_need4, max_shape_need4, s->stack_lds4, s->stack_spill4);

  for (int i = 0; i < d.num_lights; i++)
    if (d.lights[i].instance >= 0) {
      int ref = shapes[(size_t)d.instances[d.lights[i].instance].shape].root_ref;
      if (ref >= 0 || ((~ref) & 15) > 4) s->large_mesh_lights = true, s->light_features |= VPT_FEAT_LARGE_LIGHTS;
      else s->light_features |= VPT_FEAT_SMALL_LIGHTS;
    } else if (d.lights[i].sdf >= 0) s->light_features |= VPT_FEAT_SDF_LIGHTS;
  std::vector<DInstance> instances((size_t)d.num_instances);
  for (int i = 0; i < d.num_instances; i++) {
    hframe f = to_h(d.instances[i].frame);
    instances[i] = {};
    pack_frame(hinverse(f, true), instances[i].inv);
    pack_frame(f, instances[i].fwd);
    instances[i].shape = d.instances[i].shape, instances[i].material = d.instances[i].material;
    {
      const vpt_shape& sh = d.shapes[d.instances[i].shape];
      instances[i].shape_flags = (sh.num_triangles != 0 ? VPT_SHP_TRIANGLES : 0) | (sh.normal_offset >= 0 ? VPT_SHP_NORMALS : 0) |
                                 (sh.texcoord_offset >= 0 ? VPT_SHP_TEXCOORDS : 0) | (sh.color_offset >= 0 ? VPT_SHP_COLORS : 0);
    }
    instances[i].translation_only = f.x.x == 1 && f.x.y == 0 && f.x.z == 0 && f.y.x == 0 && f.y.y == 1 && f.y.z == 0 &&
                                    f.z.x == 0 && f.z.y == 0 && f.z.z == 1;
  }
  std::vector<float4> enter((size_t)d.num_scene_bvh_prims * 6);
  std::vector<int>    slot_of((size_t)d.num_instances, -1);
  for (int k = 0; k < d.num_scene_bvh_prims; k++) {
    int id = d.scene_bvh_prims[k];
    const DInstance& in = instances[(size_t)id];
    const DShape&    sh = shapes[(size_t)in.shape];
    float4* e = &enter[6 * (size_t)k];
    e[0] = in.inv[0], e[1] = in.inv[1], e[2] = in.inv[2];
    e[3] = make_float4(sh.root_box[0], sh.root_box[1], sh.root_box[2], sh.root_box[3]);
    // the quad nodes of all BVHs live in one array, the scene's first: a level is named by the index of its first node
    int tail[6] = {sh.root_ref, (int)(scene_wnodes.size() / 8) + sh.wnode_offset, sh.leaf_offset, id, in.translation_only, sh.num_nodes};
    e[4] = make_float4(sh.root_box[4], sh.root_box[5], 0, 0);
    memcpy(&e[4].z, &tail[0], 8);
    memcpy(&e[5], &tail[2], 16);
    slot_of[(size_t)id] = k;
  }
  s->h_slot_of = slot_of;
  for (int i = 0; i < d.num_instances; i++) s->h_inst_shape.push_back(d.instances[i].shape);
  std::vector<float4> env_inv((size_t)d.num_environments * 3), sdf_inv((size_t)d.num_sdfs * 3);
  for (int i = 0; i < d.num_environments; i++) pack_frame(hinverse(to_h(d.environments[i].frame), false), &env_inv[3 * (size_t)i]);
  for (int i = 0; i < d.num_sdfs; i++) pack_frame(hinverse(to_h(d.sdfs[i].frame), false), &sdf_inv[3 * (size_t)i]);
  // sRGB decode LUT: byte_to_float then srgb_to_rgb, yocto_color.h:212-227, evaluated with the host powf
  std::vector<float> lut(256);
  for (int b = 0; b < 256; b++) {
    float srgb = b / 255.0f;
    lut[b]     = (srgb <= 0.04045) ? srgb / 12.92f : std::pow((srgb + 0.055f) / (1.0f + 0.055f), 2.4f);
  }

  int rc = VPT_OK;
#define UP(expr) if ((rc = (expr)) != VPT_OK) return rc
  static_assert(sizeof(vpt_bvh_node) == 2 * sizeof(float4), "bvh node = 2 x float4");
  UP(upload(s, (const float4*)d.scene_bvh_nodes, 2LL * d.num_scene_bvh_nodes, &D.scene_nodes));
  UP(upload(s, d.scene_bvh_prims, d.num_scene_bvh_prims, &D.scene_prims));
  UP(upload(s, (const float4*)d.shape_bvh_nodes, 2LL * d.num_shape_bvh_nodes, &D.shape_nodes));
  leafs.resize(leafs.size() + 8, make_float4(0, 0, 0, 0));   // phase B fetches one record ahead of the one it tests
  UP(upload(s, leafs, &D.leaf_prims));
  UP(upload(s, leaf_attrs, &D.leaf_attrs));
  {   // every shape holds triangles: the compact records beside the general ones (vpt_device.h: tri_prims / tri_attrs)
    bool all_triangles = d.num_shapes > 0 && !getenv("VPT_NO_COMPACT_TRIANGLES");
    for (int i = 0; i < d.num_shapes; i++) all_triangles = all_triangles && shapes[i].is_triangles && shapes[i].num_elems > 0;
    D.tri_prims = D.tri_attrs = nullptr;
    if (all_triangles) {
      const size_t slots = leaf_attrs.size() / 6;
      std::vector<float4> tp(3 * slots + 8, make_float4(0, 0, 0, 0)), ta(4 * slots);   // (+ 8: phase B fetches one record ahead, as above)
      for (size_t k = 0; k < slots; k++) {
        const float4 *p = &leafs[4 * k], *a = &leaf_attrs[6 * k];
        for (int c = 0; c < 3; c++) tp[3 * k + c] = p[c], ta[4 * k + c] = a[c];
        ta[4 * k + 0].w = a[4].x, ta[4 * k + 1].w = a[4].y, ta[4 * k + 2].w = a[4].z;
        ta[4 * k + 3] = make_float4(a[4].w, a[5].x, a[5].y, 0);
      }
      UP(upload(s, tp, &D.tri_prims));
      UP(upload(s, ta, &D.tri_attrs));
    }
  }
  {
    const size_t scene_count = scene_wnodes.size();
    if ((scene_count + shape_wnodes.size()) / 8 >= (1ull << 27)) return fail(VPT_ERR_UNSUPPORTED, "more than 2^27 quad nodes");
    std::vector<float4> wnodes = scene_wnodes;   // one allocation: [scene quad nodes][shape quad nodes]
    wnodes.insert(wnodes.end(), shape_wnodes.begin(), shape_wnodes.end());
    UP(upload(s, wnodes, &D.scene_wnodes));
    D.shape_wnodes = D.scene_wnodes + scene_count;
  }
  UP(upload(s, enter, &D.scene_enter));
  UP(upload(s, slot_of, &D.slot_of_instance));
  UP(upload(s, instances, &D.instances));
  UP(upload(s, shapes, &D.shapes));
  UP(upload(s, elems, &D.elems));
  UP(upload(s, positions, &D.positions));
  UP(upload(s, normals, &D.normals));
  UP(upload(s, texcoords, &D.texcoords));
  UP(upload(s, colors, &D.colors));
  UP(upload(s, d.materials, d.num_materials, &D.materials));
  UP(upload(s, d.textures, d.num_textures, &D.textures));
  UP(upload(s, (const float4*)d.texels_f, d.num_texels_f, &D.texels_f));
  UP(upload(s, (const uchar4*)d.texels_b, d.num_texels_b, &D.texels_b));
  UP(upload(s, lut, &D.srgb_lut));
  UP(upload(s, d.environments, d.num_environments, &D.environments));
  UP(upload(s, env_inv, &D.env_inv));
  UP(upload(s, d.lights, d.num_lights, &D.lights));
  {
    const float inf = std::numeric_limits<float>::infinity();
    std::vector<DCdfIndex> index((size_t)d.num_lights);
    std::vector<float>     pool;
    std::vector<int2>      guide;
    for (int i = 0; i < d.num_lights; i++) {
      DCdfIndex& ix = index[(size_t)i];
      ix = {};
      const float* c = d.light_cdf + d.lights[i].cdf_offset;
      long long    n = d.lights[i].cdf_len;
      bool sorted = n > 64;
      for (long long k = 1; sorted && k < n; k++) sorted = c[k - 1] <= c[k];   // false for NaN too
      if (!sorted) continue;
      std::vector<float> level(c, c + n);
      size_t mark = pool.size();
      while (true) {
        if (ix.levels == 8) { ix.levels = 0; break; }   // > 16^8 entries: keep the binary search
        ix.offset[ix.levels++] = (int)pool.size();
        ix.top_count = (int)level.size();
        pool.insert(pool.end(), level.begin(), level.end());
        pool.resize((pool.size() + 15) / 16 * 16 + (ix.levels == 1 ? 16 : 0), inf);   // level 0 is also read 16-wide from any index
        if (level.size() <= 16) break;
        std::vector<float> up((level.size() + 15) / 16);
        for (size_t g = 0; g < up.size(); g++) up[g] = level[std::min(level.size() - 1, 16 * g + 15)];
        level.swap(up);
      }
      if (ix.levels == 0) { pool.resize(mark); continue; }
      // guide table: n/4 buckets over [0, back); bracket = upper_bound of a lower / an upper bound of the bucket's r
      float back = c[n - 1];
      long long M = n / 4;
      float scale = (float)M / back;
      if (!(back > 0) || !std::isfinite(scale) || M < 16) continue;
      ix.guide_offset = (int)guide.size(), ix.guide_buckets = (int)M, ix.guide_scale = scale;
      for (long long b = 0; b < M; b++) {
        // fl(r * scale) in [b, b+1)  =>  r in [b (1 - 2^-24) / scale, (b+1) (1 + 2^-23) / scale]; widened further
        double lo_r = (double)b * (1.0 - 1.0 / 8388608.0) / (double)scale, hi_r = (double)(b + 1) * (1.0 + 1.0 / 4194304.0) / (double)scale;
        float  lf = std::nextafter((float)lo_r, -inf), hf = std::nextafter((float)hi_r, inf);
        int lo = b == 0 ? 0 : (int)(std::upper_bound(c, c + n, lf) - c);
        int hi = b == M - 1 ? (int)n : (int)(std::upper_bound(c, c + n, hf) - c);
        guide.push_back(make_int2(lo, hi));
      }
    }
    UP(upload(s, d.light_cdf, d.num_light_cdf, &D.light_cdf));
    UP(upload(s, index, &D.light_index));
    UP(upload(s, pool, &D.light_index_pool));
    UP(upload(s, guide, &D.light_guide));
  }
  {
    std::vector<float4> rec(8 * (size_t)d.num_lights, make_float4(0, 0, 0, 0));
    for (int i = 0; i < d.num_lights; i++) {
      const vpt_light& l = d.lights[i];
      float4* r = &rec[8 * (size_t)i];
      float   total = l.cdf_len > 0 ? d.light_cdf[l.cdf_offset + l.cdf_len - 1] : 0.0f;
      int     kind = VPT_LIGHT_NONE, count = 0;
      if (l.instance != VPT_INVALID) {
        const DInstance& in = instances[(size_t)l.instance];
        const DShape&    sh = shapes[(size_t)in.shape];
        // a shape whose BVH is one leaf of <= 4 primitives (the reference's bvh_max_prims) is walked inline from the light's own
        // copy of them (light_prims holds four); anything else goes through the traversal
        bool small = sh.root_ref < 0 && ((~sh.root_ref) & 15) <= 4;
        kind  = small ? VPT_LIGHT_SMALL_MESH : VPT_LIGHT_LARGE_MESH;
        count = small ? ((~sh.root_ref) & 15) : 0;
        for (int k = 0; k < 3; k++) r[k] = in.inv[k], r[3 + k] = in.fwd[k];
        r[6] = make_float4(sh.root_box[0], sh.root_box[1], sh.root_box[2], total);
        r[7] = make_float4(sh.root_box[3], sh.root_box[4], sh.root_box[5], 0);
      } else if (l.sdf != VPT_INVALID) {
        kind = VPT_LIGHT_SDF;
      } else if (l.environment != VPT_INVALID && d.environments[l.environment].emission_tex == VPT_INVALID) {
        kind = VPT_LIGHT_ENV_CONST;
      } else if (l.environment != VPT_INVALID) {
        kind = VPT_LIGHT_ENV_TEX;
        const vpt_texture& t = d.textures[d.environments[l.environment].emission_tex];
        for (int k = 0; k < 3; k++) r[k] = env_inv[3 * (size_t)l.environment + k];
        pack_frame(to_h(d.environments[l.environment].frame), &r[3]);
        int dims[2] = {t.width, t.height};
        memcpy(&r[6].x, dims, 8);
        r[6].z = total;
      }
      int tag = kind | (count << 8);
      memcpy(&r[7].w, &tag, 4);
    }
    UP(upload(s, rec, &D.light_rec));
    UP(upload(s, std::vector<float4>(20 * (size_t)d.num_lights, make_float4(0, 0, 0, 0)), &D.light_prims));
  }
  UP(upload(s, d.volumes, d.num_volumes, &D.volumes));
  UP(upload(s, d.voxels, d.num_voxels, &D.voxels));
  UP(upload(s, d.vol_instances, d.num_vol_instances, &D.vol_instances));
  UP(upload(s, d.sdfs, d.num_sdfs, &D.sdfs));
  UP(upload(s, sdf_inv, &D.sdf_inv));
  {
    // SDF evaluation records (vpt_scene.hip.h "SDF records") and the balls the escaping-ray early-out needs.  The
    // constants are folded with the reference's own float operations (yocto_sdfs.cpp:33-38, yocto_sceneio.cpp:3697);
    // the balls are test-independent geometry, computed in double with a 5 % margin.  Only rigid frames get a ball
    // (a scaling frame turns SDF values into something other than world distances): radius -1 switches the early-out off.
    auto rigid = [](const vpt_frame& f) {
      double c[3][3] = {{f.x[0], f.x[1], f.x[2]}, {f.y[0], f.y[1], f.y[2]}, {f.z[0], f.z[1], f.z[2]}};
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          double dp = c[i][0] * c[j][0] + c[i][1] * c[j][1] + c[i][2] * c[j][2];
          if (std::fabs(dp - (i == j ? 1.0 : 0.0)) > 1e-5) return false;
        }
      return true;
    };
    auto identity3 = [](const vpt_frame& f) {
      return f.x[0] == 1 && f.x[1] == 0 && f.x[2] == 0 && f.y[0] == 0 && f.y[1] == 1 && f.y[2] == 0 && f.z[0] == 0 && f.z[1] == 0 && f.z[2] == 1;
    };
    // world position of a local point: the SDFs apply the FORWARD frame to world points (yocto_sdfs.cpp:13), so world = R^T (local - o)
    auto to_world = [](const vpt_frame& f, const double l[3], double w[3]) {
      double v[3] = {l[0] - f.o[0], l[1] - f.o[1], l[2] - f.o[2]};
      w[0] = f.x[0] * v[0] + f.y[0] * v[1] + f.z[0] * v[2];   // rows of R^T = the frame's x, y, z taken component-wise
      w[1] = f.x[1] * v[0] + f.y[1] * v[1] + f.z[1] * v[2];
      w[2] = f.x[2] * v[0] + f.y[2] * v[1] + f.z[2] * v[2];
    };
    struct ball { double c[3], r; };
    std::vector<ball> balls;
    bool all_bounded_rigid = true;
    int  planes = 0;
    std::vector<float4> fn_rec(6 * (size_t)d.num_sdfs, make_float4(0, 0, 0, 0)), grid_rec(7 * (size_t)d.num_vol_instances, make_float4(0, 0, 0, 0));
    for (int i = 0; i < d.num_sdfs; i++) {
      const vpt_sdf& f = d.sdfs[i];
      float4* r = &fn_rec[6 * (size_t)i];
      pack_frame(to_h(f.frame), r);
      r[3] = make_float4(f.p[0], f.p[1], f.p[2], f.p[3]);
      r[4] = make_float4(f.whd[0] * 0.5f, f.whd[1] * 0.5f, f.whd[2] * 0.5f, 0);
      int tag = f.type | ((identity3(f.frame) ? 1 : 0) << 8);
      memcpy(&r[4].w, &tag, 4);
      double lc[3] = {0, 0, 0}, lr = -1;   // local centre / radius of a ball around the shape
      switch (f.type) {
        case VPT_SDF_BOX: lc[0] = f.whd[0] * 0.5, lc[1] = f.whd[1] * 0.5, lc[2] = f.whd[2] * 0.5, lr = std::sqrt(lc[0] * lc[0] + lc[1] * lc[1] + lc[2] * lc[2]); break;
        case VPT_SDF_BBOX: lr = std::sqrt((double)f.p[1] * f.p[1] + (double)f.p[2] * f.p[2] + (double)f.p[3] * f.p[3]) + 2.0 * std::fabs((double)f.p[0]); break;
        case VPT_SDF_SPHERE: lr = std::fabs((double)f.p[0]); break;
        case VPT_SDF_TORUS: lr = std::fabs((double)f.p[0]) + std::fabs((double)f.p[1]); break;
        case VPT_SDF_CAPPED_CONE: lr = std::sqrt((double)f.p[0] * f.p[0] + std::max((double)f.p[1] * f.p[1], (double)f.p[2] * f.p[2])); break;
        default: break;   // plane: unbounded
      }
      r[5] = make_float4(0, 0, 0, -1);
      if (f.type == VPT_SDF_PLANE) planes++;
      else if (lr > 0 && std::isfinite(lr) && rigid(f.frame)) {
        ball b;
        to_world(f.frame, lc, b.c);
        b.r = lr * 1.05 + 1e-6;
        balls.push_back(b);
        r[5] = make_float4((float)b.c[0], (float)b.c[1], (float)b.c[2], (float)b.r);
      } else all_bounded_rigid = false;
    }
    for (int i = 0; i < d.num_vol_instances; i++) {
      const vpt_volume_instance& vi = d.vol_instances[i];
      const vpt_volume&          vol = d.volumes[vi.volume];
      float4* r = &grid_rec[7 * (size_t)i];
      pack_frame(to_h(vi.frame), r);
      // bbox_max = origin + (vol.res * grid_res) * scalef; bbox_size = bbox_max - origin   (yocto_sdfs.cpp:33-36, float)
      float size[3];
      for (int k = 0; k < 3; k++) {
        float origin = vi.frame.o[k], grid_res = (float)vol.whd[k];
        float bbox_max = origin + (vol.res * grid_res) * vi.scalef;
        size[k] = bbox_max - origin;
      }
      r[3] = make_float4(size[0], size[1], size[2], vi.scalef);
      r[4] = make_float4(size[0] * 0.5f, size[1] * 0.5f, size[2] * 0.5f, 0);
      int tr = identity3(vi.frame) ? 1 : 0;
      memcpy(&r[4].w, &tr, 4);
      int dims[3] = {vol.whd[0], vol.whd[1], vol.whd[2]};
      memcpy(&r[5], dims, 12);
      r[5].w = vol.res;
      int off[2] = {(int)(vol.offset & 0xffffffffll), (int)(vol.offset >> 32)};
      memcpy(&r[6], off, 8);
      double lc[3] = {size[0] * 0.5, size[1] * 0.5, size[2] * 0.5}, lr = std::sqrt(lc[0] * lc[0] + lc[1] * lc[1] + lc[2] * lc[2]);
      if (lr > 0 && std::isfinite(lr) && rigid(vi.frame)) {
        ball b;
        to_world(vi.frame, lc, b.c);
        b.r = lr * 1.05 + 1e-6;
        balls.push_back(b);
      } else all_bounded_rigid = false;
    }
    D.sdf_bound_cx = D.sdf_bound_cy = D.sdf_bound_cz = 0, D.sdf_bound_r = -1, D.sdf_num_planes = planes;
    if (all_bounded_rigid && !balls.empty()) {
      double c[3] = {0, 0, 0}, rr = 0;
      for (const ball& b : balls)
        for (int k = 0; k < 3; k++) c[k] += b.c[k] / (double)balls.size();
      for (const ball& b : balls) {
        double dist = std::sqrt((b.c[0] - c[0]) * (b.c[0] - c[0]) + (b.c[1] - c[1]) * (b.c[1] - c[1]) + (b.c[2] - c[2]) * (b.c[2] - c[2]));
        rr = std::max(rr, dist + b.r);
      }
      D.sdf_bound_cx = (float)c[0], D.sdf_bound_cy = (float)c[1], D.sdf_bound_cz = (float)c[2], D.sdf_bound_r = (float)(rr * 1.01);
    }
    if (getenv("VPT_NO_EARLY_OUT")) {   // A/B switch for the experiments of DESIGN.md: the marches then run to the reference's own end
      D.sdf_bound_r = -1;
      for (int i = 0; i < d.num_sdfs; i++) fn_rec[6 * (size_t)i + 5].w = -1;
    }
    UP(upload(s, fn_rec, &D.sdf_fn_rec));
    UP(upload(s, grid_rec, &D.sdf_grid_rec));
  }
  UP(upload(s, d.cameras, d.num_cameras, &D.cameras));
#undef UP
  {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    s->wave_slots_k1 = prop.multiProcessorCount * 4 * VPT_WAVES_PER_SIMD;
    s->wave_slots_k2 = prop.multiProcessorCount * 4 * VPT_K2_WAVES;
  }
  if (d.num_lights > 0) {   // element normals of the single-leaf mesh lights, by the device's own eval_element_normal
    hipLaunchKernelGGL(vpt_light_setup_kernel, dim3(d.num_lights), dim3(64), 0, 0, s->d, const_cast<float4*>(D.light_prims));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
  }
  HIP_TRY(hipEventCreate(&s->ev0));
  HIP_TRY(hipEventCreate(&s->ev1));
  HIP_TRY(hipEventCreate(&s->ev_host0));
  HIP_TRY(hipEventCreate(&s->ev_host1));
  HIP_TRY(hipMalloc((void**)&s->d_watchdog, 4));
  HIP_TRY(hipMemset(s->d_watchdog, 0, 4));
  HIP_TRY(hipEventCreateWithFlags(&s->ev_order, hipEventDisableTiming));
  HIP_TRY(hipDeviceSynchronize());
  *out = s;
  s    = nullptr;   // release the guard
  return VPT_OK;
}

int64_t vpt_layout_slots(const vpt_layout* layout) {
  DParams   pr;
  vpt_params dummy = {};
  if (make_dparams(&dummy, layout, 0, pr) != VPT_OK) return -1;
  return pr.nslots;
}

static int permute(const vpt_layout* layout, int to_tiles, void* t_image, void* t_hits, void* t_rng, void* r_image,
    void* r_hits, void* r_rng, hipStream_t stream) {
  DParams    pr;
  vpt_params dummy = {};
  if (int rc = make_dparams(&dummy, layout, 0, pr)) return rc;
  int blocks = (pr.nslots + 255) / 256;
  hipLaunchKernelGGL(vpt_permute_kernel, dim3(blocks), dim3(256), 0, stream, pr, to_tiles, (float4*)t_image, (int*)t_hits,
      (ulonglong2*)t_rng, (float4*)r_image, (int*)r_hits, (ulonglong2*)r_rng);
  HIP_TRY(hipGetLastError());
  return VPT_OK;
}

// row-major device staging of one frame for the host <-> tile-major conversions: borrowed from the scene handle where there is
// one (vpt_render: allocated once per frame size), else allocated for the call; released on every path
struct row_staging {
  void *image = nullptr, *hits = nullptr, *rng = nullptr;
  bool  owned = false;
  ~row_staging() {
    if (owned)
      for (void* p : {image, hits, rng})
        if (p) (void)hipFree(p);
  }
  int allocate(size_t pixels) {
    owned = true;
    HIP_TRY(hipMalloc(&image, pixels * 16));
    HIP_TRY(hipMalloc(&hits, pixels * 4));
    HIP_TRY(hipMalloc(&rng, pixels * 16));
    return VPT_OK;
  }
};
static int state_upload(const vpt_layout* layout, const float* image_rgba, const int32_t* hits, const uint64_t* rng,
    void* d_image, void* d_hits, void* d_rng, hipStream_t st, row_staging& rows) {
  size_t n = (size_t)layout->width * layout->height;
  if (!rows.image)
    if (int rc = rows.allocate(n)) return rc;
  HIP_TRY(hipMemcpyAsync(rows.image, image_rgba, n * 16, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(rows.hits, hits, n * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(rows.rng, rng, n * 16, hipMemcpyHostToDevice, st));
  if (int rc = permute(layout, 1, d_image, d_hits, d_rng, rows.image, rows.hits, rows.rng, st)) return rc;
  HIP_TRY(hipStreamSynchronize(st));
  return VPT_OK;
}
static int state_download(const vpt_layout* layout, const void* d_image, const void* d_hits, const void* d_rng,
    float* image_rgba, int32_t* hits, uint64_t* rng, hipStream_t st, row_staging& rows, bool rows_hold_the_frame) {
  size_t n = (size_t)layout->width * layout->height;
  if (!rows.image)
    if (int rc = rows.allocate(n)) return rc;
  if (!rows_hold_the_frame) {   // start from the caller's arrays so that pixels owned by other ranks keep their values
    HIP_TRY(hipMemcpyAsync(rows.image, image_rgba, n * 16, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(rows.hits, hits, n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(rows.rng, rng, n * 16, hipMemcpyHostToDevice, st));
  }
  if (int rc = permute(layout, 0, (void*)d_image, (void*)d_hits, (void*)d_rng, rows.image, rows.hits, rows.rng, st)) return rc;
  HIP_TRY(hipMemcpyAsync(image_rgba, rows.image, n * 16, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(hits, rows.hits, n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(rng, rows.rng, n * 16, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return VPT_OK;
}

int vpt_state_upload(const vpt_layout* layout, const float* image_rgba, const int32_t* hits, const uint64_t* rng,
    void* d_image, void* d_hits, void* d_rng, void* stream) {
  if (!layout || !image_rgba || !hits || !rng || !d_image || !d_hits || !d_rng) return fail(VPT_ERR_INVALID_ARG, "null argument");
  if (layout->width <= 0 || layout->height <= 0) return fail(VPT_ERR_INVALID_ARG, "bad layout");
  row_staging rows;
  return state_upload(layout, image_rgba, hits, rng, d_image, d_hits, d_rng, (hipStream_t)stream, rows);
}

int vpt_state_download(const vpt_layout* layout, const void* d_image, const void* d_hits, const void* d_rng,
    float* image_rgba, int32_t* hits, uint64_t* rng, void* stream) {
  if (!layout || !image_rgba || !hits || !rng || !d_image || !d_hits || !d_rng) return fail(VPT_ERR_INVALID_ARG, "null argument");
  if (layout->width <= 0 || layout->height <= 0) return fail(VPT_ERR_INVALID_ARG, "bad layout");
  row_staging rows;
  return state_download(layout, d_image, d_hits, d_rng, image_rgba, hits, rng, (hipStream_t)stream, rows, false);
}

}  // extern "C"

// HBM part of the traversal stacks for a launch of `lanes` lanes (only scenes whose worst case exceeds the LDS part)
static int stack_config(vpt_scene* s, long long lanes, stack_cfg& cfg) {
  if (s->stack_spill4 > 0 && lanes > s->spill_lanes) {
    if (s->spill) (void)hipFree(s->spill);
    s->spill = nullptr, s->spill_lanes = 0;
    HIP_TRY(hipMalloc(&s->spill, (size_t)lanes * (size_t)s->stack_spill4 * sizeof(int2)));
    s->spill_lanes = lanes;
  }
  cfg.cap = s->stack_lds4, cfg.spill = s->stack_spill4, cfg.mem = (int2*)s->spill, cfg.lanes = lanes;
  return VPT_OK;
}

// Buffers of the launch schedule for `waves` waves; a change of layout / camera / shader forgets the measured costs.
static int sched_prepare(vpt_scene* s, long long waves, const long long key[10], hipStream_t st) {
  if (waves > s->sched_waves) {
    for (void** p : {(void**)&s->d_cost, (void**)&s->d_cost_sorted, (void**)&s->d_cost_key, (void**)&s->d_cost_avg, (void**)&s->d_order, (void**)&s->d_iota, &s->sort_temp})
      if (*p) (void)hipFree(*p), *p = nullptr;
    s->sched_waves = 0, s->order_valid = false, s->cost_weight = 0;
    HIP_TRY(hipMalloc((void**)&s->d_cost, waves * 4));
    HIP_TRY(hipMalloc((void**)&s->d_cost_sorted, waves * 4));
    HIP_TRY(hipMalloc((void**)&s->d_cost_key, waves * 4));
    HIP_TRY(hipMalloc((void**)&s->d_cost_avg, waves * 4));
    HIP_TRY(hipMalloc((void**)&s->d_order, waves * 4));
    HIP_TRY(hipMalloc((void**)&s->d_iota, waves * 4));
    HIP_TRY(hipMemset(s->d_cost, 0, waves * 4));   // waves that own no pixel never write theirs
    std::vector<int> iota((size_t)waves);
    for (long long i = 0; i < waves; i++) iota[(size_t)i] = (int)i;
    HIP_TRY(hipMemcpy(s->d_iota, iota.data(), waves * 4, hipMemcpyHostToDevice));
    size_t bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs_desc((void*)nullptr, bytes, s->d_cost, s->d_cost_sorted, s->d_iota, s->d_order, (size_t)waves));
    HIP_TRY(hipMalloc(&s->sort_temp, bytes ? bytes : 16));
    s->sort_temp_bytes = bytes, s->sched_waves = waves;
  }
  if (memcmp(key, s->sched_key, sizeof(s->sched_key)) != 0)
    s->order_valid = false, s->split_decided = false, s->full_costs = false, s->split_waves = 0, s->split_tiles = 0, s->cost_weight = 0, memcpy(s->sched_key, key, sizeof(s->sched_key));
  (void)st;
  return VPT_OK;
}
// order[] for the next launch from the costs the launch just enqueued on `st` will have written
static float vpt_cost_horizon() {   // launches behind the running average (VPT_COST_HORIZON, calibration)
  static const float h = [] { const char* e = getenv("VPT_COST_HORIZON"); return e ? (float)atof(e) : 6.0f; }();
  return h < 1 ? 1 : h;
}
static int sched_update(vpt_scene* s, long long waves, hipStream_t st, int nsamples = 0) {
  // nsamples > 0: d_cost holds the durations of a launch over that many samples: they enter the running average, whose order the next launch takes;
  // nsamples == 0: d_cost holds predictions (a fresh split table): they start a new average
  static const bool averaging = [] { const char* e = getenv("VPT_COST_AVERAGE"); return !e || atoi(e) != 0; }();
  if (nsamples <= 0 || !averaging) s->cost_weight = 0;
  const float n = nsamples > 0 ? (float)nsamples : 1.0f;
  hipLaunchKernelGGL(vpt_cost_average_kernel, dim3((unsigned)((waves + 255) / 256)), dim3(256), 0, st, s->d_cost, s->d_cost_avg, s->d_cost_key, (int)waves, n, s->cost_weight);
  HIP_TRY(hipGetLastError());
  if (nsamples > 0 && averaging) s->cost_weight = std::min(s->cost_weight + n, vpt_cost_horizon() * n);   // the last few launches
  size_t bytes = s->sort_temp_bytes;
  HIP_TRY(rocprim::radix_sort_pairs_desc(s->sort_temp, bytes, s->d_cost_key, s->d_cost_sorted, s->d_iota, s->d_order, (size_t)waves, 0, 32, st));
  HIP_TRY(hipEventRecord(s->ev_order, st));
  s->order_valid = true, s->order_stream = st;
  return VPT_OK;
}
// d_order / d_cost are written on the stream of the previous launch: a launch on another stream waits for that sort
static int sched_wait(vpt_scene* s, hipStream_t st) {
  if (s->order_valid && s->order_stream != st) HIP_TRY(hipStreamWaitEvent(st, s->ev_order, 0));
  return VPT_OK;
}

// everything one vpt_render_device call hands to the kernel launchers
struct launch_ctx {
  vpt_scene*        s;
  const vpt_params* params;
  const DParams&    pr;
  dim3              grid, block;
  hipStream_t       st;
  float4*           img;
  int*              hit;
  ulonglong2*       rng;
  stack_cfg         stack;
};
static void schedule_key(const launch_ctx& L, long long key[10]) {
  const DParams& pr = L.pr;
  long long k[10] = {pr.nslots, pr.width, pr.height, L.params->shader, L.params->camera, L.params->bounces, pr.rank, pr.nranks, pr.tile_w, pr.tile_h};
  memcpy(key, k, sizeof(k));
}

// ---- tile splitting (K1) --------------------------------------------------------------------------------------------
// A wave runs all samples of its 64 pixels one after the other, so a launch cannot be shorter than its costliest tile.
// On one GPU that tile (273 ms of a 280 ms launch on 03_volume) is level with total work / wave slots and nothing is
// gained by shortening it; once the frame is shared among N GPUs the work per GPU falls with N and the chain does not.
// A tile can be run as 2^k waves that hold every 2^k-th pixel in their first 64 >> k lanes: fewer live lanes diverge
// less, the wave's trips get faster (g[k] below, measured on MI355X: DESIGN.md §5) - at 2^k g[k] times the slot time.
// Policy (decide_split): for a range of candidate spans S every tile is split just enough for its waves to fit S and the
// resulting launch is simulated (longest-first list scheduling on the chip's wave slots, durations scaled by how full the
// chip is); the shortest simulated launch wins if it beats the unsplit one by 2 %.  Taken once per layout / shader /
// camera from the per-tile costs of an unsplit launch, when the launch is short of waves: the frame is shared among ranks
// or holds fewer than three tiles per wave slot (1280x533 on one MI355X has 3.5 and never gains).  Pixels keep their own RNG streams and accumulators, so the result does
// not depend on it.
static double split_gain[7] = {1.0, 0.75, 0.57, 0.44, 0.34, 0.27, 0.20};   // duration of a 64 >> k lane wave of a costly tile / its full wave (round 4's
                                                                           // kernel, whose partly filled waves use their empty lanes as helpers: 0.753 / 0.566 / 0.436 / 0.343 measured, was 0.81 / 0.62 / 0.45 / 0.35)
static double split_gain_k2[7] = {1.0, 0.82, 0.67, 0.63, 0.60, 0.58, 0.56};   // the same for K2 (implicit shaders): 0.82 / 0.67 / 0.63 measured on 06_gridsdf_full (profiles/r04_k2_lane_histogram.txt), the rest extrapolated
static double split_load0 = 0.46, split_margin = 0.98;   // load_factor's intercept; a split has to beat the unsplit launch by this factor
static void split_tuning() {   // VPT_SPLIT_TUNE="g1,g2,g3,g4,g5,g6,load0,margin" (calibration runs only)
  static bool once = [] {
    if (const char* e = getenv("VPT_K2_SPLIT_TUNE")) {   // the same for the implicit kernels' table: "g1,g2,g3,g4,g5,g6"
      double v[6];
      if (sscanf(e, "%lf,%lf,%lf,%lf,%lf,%lf", v, v + 1, v + 2, v + 3, v + 4, v + 5) == 6)
        for (int i = 0; i < 6; i++) split_gain_k2[i + 1] = v[i];
    }
    if (const char* e = getenv("VPT_SPLIT_TUNE")) {
      double v[8];
      if (sscanf(e, "%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6, v + 7) == 8) {
        for (int i = 0; i < 6; i++) split_gain[i + 1] = v[i];
        split_load0 = v[6], split_margin = v[7];
      }
    }
    return true;
  }();
  (void)once;
}
static int split_mode() {   // VPT_SPLIT: 0 never, 1 always consider, unset: consider when the launch is short of waves
  static int v = [] { const char* e = getenv("VPT_SPLIT"); return e ? atoi(e) : -1; }();
  return v;
}
static int split_forced_k() {   // VPT_SPLIT_K (calibration): every tile as 2^k waves
  static int v = [] { const char* e = getenv("VPT_SPLIT_K"); return e ? atoi(e) : -1; }();
  return v;
}
// lane table and predicted wave costs for the split factors s->h_split_k; part_cost[t] = expected duration of one wave of tile t
static int build_split_table(vpt_scene* s, const DParams& pr, const std::vector<double>& part_cost, hipStream_t st) {
  const std::vector<int>& k = s->h_split_k;
  const int ntiles = (int)k.size();
  long long waves = 0;
  int       nsplit = 0;
  for (int t = 0; t < ntiles; t++) waves += 1ll << k[t], nsplit += k[t] > 0;
  s->split_waves = 0, s->split_tiles = 0;
  if (nsplit == 0 || waves > (1ll << 24)) return VPT_OK;
  std::vector<int>      table((size_t)waves * VPT_BLOCK, -1);
  std::vector<unsigned> wcost((size_t)waves);
  long long w = 0;
  for (int t = 0; t < ntiles; t++)
    for (int part = 0; part < (1 << k[t]); part++, w++) {
      wcost[(size_t)w] = (unsigned)part_cost[t];
      for (int lane = 0; lane < (VPT_BLOCK >> k[t]); lane++) table[(size_t)w * VPT_BLOCK + lane] = t * VPT_BLOCK + (lane << k[t]) + part;
    }
  if ((long long)table.size() > s->lane_cap) {
    if (s->d_lane_slot) (void)hipFree(s->d_lane_slot), s->d_lane_slot = nullptr;
    s->lane_cap = 0;
    HIP_TRY(hipMalloc((void**)&s->d_lane_slot, table.size() * 4));
    s->lane_cap = (long long)table.size();
  }
  long long key[10];
  memcpy(key, s->sched_key, sizeof(key));
  if (int rc = sched_prepare(s, waves, key, st)) return rc;   // may reallocate d_cost / d_order for the larger wave count
  HIP_TRY(hipMemcpy(s->d_lane_slot, table.data(), table.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->d_cost, wcost.data(), wcost.size() * 4, hipMemcpyHostToDevice));
  s->split_waves = (int)waves, s->split_tiles = nsplit;
  if (getenv("VPT_SPLIT_VERBOSE"))
    fprintf(stderr, "[vpt] split: %d of %d tiles -> %lld waves on %d slots (rank %d of %d)\n", nsplit, ntiles, waves, s->wave_slots_k1, pr.rank, pr.nranks);
  return sched_update(s, waves, st);   // order of the split launch from the predicted costs; measured ones take over afterwards
}
// makespan of longest-first list scheduling of `costs` (any order) on `slots` machines: what the hardware's dispatch of
// the launch in d_order amounts to
static double lpt_makespan(std::vector<double>& costs, int slots) {
  std::sort(costs.begin(), costs.end(), std::greater<double>());
  std::priority_queue<double, std::vector<double>, std::greater<double>> load;
  double span = 0;
  for (size_t i = 0; i < costs.size(); i++) {
    double at = 0;
    if ((int)load.size() >= slots) at = load.top(), load.pop();
    load.push(at + costs[i]);
    span = std::max(span, at + costs[i]);
  }
  return span;
}
// A wave also runs faster when fewer waves share its SIMD: the costliest tile of 03_volume takes 273 ms with all 3 072
// slots busy and 187 ms when 1 340 waves are resident: duration ~ (0.46 + 0.54 * occupancy) * duration at 1
static double load_factor(double waves, int slots) { return split_load0 + (1 - split_load0) * std::min(1.0, waves / slots); }
static int decide_split(vpt_scene* s, const DParams& pr, int ntiles, int slots, hipStream_t st, const double* split_gain = ::split_gain) {
  s->split_decided = true, s->split_waves = 0, s->split_tiles = 0;
  split_tuning();
  HIP_TRY(hipStreamSynchronize(st));
  std::vector<unsigned> cost((size_t)ntiles);
  HIP_TRY(hipMemcpy(cost.data(), s->d_cost, cost.size() * 4, hipMemcpyDeviceToHost));
  std::vector<int>& k = s->h_split_k;
  k.assign((size_t)ntiles, 0);
  double cmax = 0;
  int    live = 0;
  for (unsigned c : cost) cmax = std::max(cmax, (double)c), live += c > 0;
  if (cmax <= 0) return VPT_OK;
  const double measured_at = load_factor(live, slots);   // the costs were measured with `live` waves resident
  std::vector<double> waves;
  auto plan = [&](double S, bool apply) {   // predicted span when every tile is split just enough for its waves to fit S
    waves.clear();
    for (int t = 0; t < ntiles; t++) {
      if (cost[t] == 0) continue;
      int kt = 0;
      while (kt < 6 && cost[t] * split_gain[kt] > S) kt++;
      if (apply) k[t] = kt;
      for (int p = 0; p < (1 << kt); p++) waves.push_back(cost[t] * split_gain[kt]);
    }
    double f = load_factor((double)waves.size(), slots) / measured_at;
    for (double& w : waves) w *= f;
    return lpt_makespan(waves, slots);
  };
  if (split_forced_k() >= 0) {
    for (int t = 0; t < ntiles; t++) k[t] = std::min(split_forced_k(), 6);
  } else {
    double best_S = cmax, best = plan(cmax, false);
    for (int i = 1; i <= 24; i++) {   // candidates from the costliest tile down to its 1-lane duration
      double S = cmax * std::pow(split_gain[6], i / 24.0), span = plan(S, false);
      if (span < best * split_margin) best = span, best_S = S;   // a split has to pay at least 2 %
    }
    plan(best_S, true);
  }
  std::vector<double> part((size_t)ntiles);
  for (int t = 0; t < ntiles; t++) part[t] = cost[t] * split_gain[k[t]];
  return build_split_table(s, pr, part, st);
}

// K1 (mesh shaders).  Longest-wave-first order from the costs of the previous launch on this layout; without
// them a pilot launch over 1/64 of the call's samples (1..16) measures them first - same arithmetic, batching is exact.
template <int K>
static int launch_mesh(const launch_ctx& L) {
  vpt_scene* s = L.s;
#if defined(VPT_EXPERIMENT_ONLY_K2)   // experiment builds (make variant): only the kernels under study are compiled (minutes -> seconds)
  return fail(VPT_ERR_UNSUPPORTED, "this experiment build holds the implicit kernels only");
#else
#if defined(VPT_EXPERIMENT_ONLY_VOLPATH)
  if (K != K_VOLPATH) return fail(VPT_ERR_UNSUPPORTED, "this experiment build holds the volpathtrace kernel only");
#endif
  long long key[10];
  schedule_key(L, key);
  if (int rc = sched_prepare(s, std::max<long long>(L.grid.x, s->split_waves), key, L.st)) return rc;
  if (int rc = sched_wait(s, L.st)) return rc;
  size_t lds = (size_t)s->stack_lds4 * 2 * VPT_BLOCK * sizeof(int) + 5 * VPT_BLOCK * sizeof(float);   // (ref, t0) pairs + the parked words
  int n = L.pr.nsamples, pilot = n / 64 < 1 ? 1 : n / 64 > 16 ? 16 : n / 64;
  int parts[2] = {(!s->order_valid && n >= 16) ? pilot : n, 0};
  parts[1] = n - parts[0];
  const bool may_split = !L.stack.spill && (split_mode() == 1 || split_forced_k() >= 0 ||
                                            (split_mode() < 0 && (L.pr.nranks > 1 || (long long)L.grid.x < 3ll * s->wave_slots_k1)));
  for (int part = 0; part < 2 && parts[part] > 0; part++) {
    DParams pr  = L.pr;
    pr.nsamples = parts[part];
    bool is_pilot = parts[1] > 0 && part == 0;
    // the costs of an unsplit launch over at least 8 samples decide, once, whether tiles are split from now on
    if (may_split && !s->split_decided && s->order_valid && s->full_costs) {
      // the decision waits for the stream and reads costs back on the host: that pause is bracketed by its own event pair and
      // subtracted by vpt_last_kernel_ms (a pilot launch that ran before it in this call stays counted)
      HIP_TRY(hipEventRecord(s->ev_host0, L.st));
      if (int rc = decide_split(s, pr, (int)L.grid.x, s->wave_slots_k1, L.st)) return rc;
      HIP_TRY(hipEventRecord(s->ev_host1, L.st));
      s->host_pause = true;
    }
    dim3 grid = s->split_waves > 0 ? dim3((unsigned)s->split_waves) : L.grid;
    sched_cfg sch = {s->order_valid ? s->d_order : nullptr, s->d_cost, s->split_waves > 0 ? s->d_lane_slot : nullptr};
    // the instance compiled for the features this scene has (vpt_scene.hip.h: VPT_FEAT_*)
    const int need = getenv("VPT_NO_LEAN") ? VPT_FEAT_ALL : s->light_features;
    auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, L.block, lds, L.st, s->d, pr, L.img, L.hit, L.rng, L.stack, sch); };
    auto launch_feat = [&](auto feat) {
      constexpr int F = decltype(feat)::value;
#if defined(VPT_EXPERIMENT_ONLY_VOLPATH)   // experiment builds: the pilot runs the same instance (one kernel less to compile)
      if (L.stack.spill) launch(vpt_mesh_kernel<K, true, F>);
      else launch(vpt_mesh_kernel<K, false, F>);
#else
      if (is_pilot && L.stack.spill) launch(vpt_mesh_pilot_kernel<K, true, F>);
      else if (is_pilot) launch(vpt_mesh_pilot_kernel<K, false, F>);
      else if (L.stack.spill) launch(vpt_mesh_kernel<K, true, F>);
      else launch(vpt_mesh_kernel<K, false, F>);
#endif
    };
    // three instances: single-leaf mesh lights only / + emissive meshes with a BVH / everything (SDF lights too)
#if defined(VPT_EXPERIMENT_ONLY_VOLPATH)
    if (need != VPT_FEAT_SMALL_LIGHTS && (need & (VPT_FEAT_LARGE_LIGHTS | VPT_FEAT_SDF_LIGHTS)) != 0) return fail(VPT_ERR_UNSUPPORTED, "this experiment build holds the lean instance only");
    if (s->d.tri_prims) {
      if (L.stack.spill) launch(vpt_mesh_kernel<K, true, VPT_FEAT_SMALL_LIGHTS | VPT_FEAT_COMPACT_TRIS>);
      else launch(vpt_mesh_kernel<K, false, VPT_FEAT_SMALL_LIGHTS | VPT_FEAT_COMPACT_TRIS>);
    } else launch_feat(std::integral_constant<int, VPT_FEAT_SMALL_LIGHTS>{});
#else
    // (+ the compact-record form of the first for the two path tracers on scenes of triangles; the pilot runs on the general records)
    if ((K == K_VOLPATH || K == K_PATH) && (need & (VPT_FEAT_LARGE_LIGHTS | VPT_FEAT_SDF_LIGHTS)) == 0 && s->d.tri_prims && !is_pilot) {
      if constexpr (K == K_VOLPATH || K == K_PATH) {
        if (L.stack.spill) launch(vpt_mesh_kernel<K, true, VPT_FEAT_SMALL_LIGHTS | VPT_FEAT_COMPACT_TRIS>);
        else launch(vpt_mesh_kernel<K, false, VPT_FEAT_SMALL_LIGHTS | VPT_FEAT_COMPACT_TRIS>);
      }
    } else if ((need & (VPT_FEAT_LARGE_LIGHTS | VPT_FEAT_SDF_LIGHTS)) == 0) launch_feat(std::integral_constant<int, VPT_FEAT_SMALL_LIGHTS>{});
    else if ((need & VPT_FEAT_SDF_LIGHTS) == 0) launch_feat(std::integral_constant<int, VPT_FEAT_SMALL_LIGHTS | VPT_FEAT_LARGE_LIGHTS>{});
    else launch_feat(std::integral_constant<int, VPT_FEAT_ALL>{});
#endif
    if (s->split_waves == 0) s->full_costs = pr.nsamples >= 8;   // d_cost now holds per-tile durations over enough samples (a pilot of a call with >= 512 samples counts)
    s->last_waves = (int)grid.x;
    if (int rc = sched_update(s, grid.x, L.st, pr.nsamples)) return rc;
  }
  return VPT_OK;
#endif
}
// K2 (implicit shaders): same schedule; without costs of a previous launch on this layout a pilot launch over 1/64 of the call's
// samples (1..16) measures them first, as for K1 (in tile order a first call ran at 216 against 302 Msamples/s on 06_gridsdf)
template <int K>
static int launch_implicit(const launch_ctx& L) {
  vpt_scene* s = L.s;
#if defined(VPT_EXPERIMENT_ONLY_VOLPATH)
  return fail(VPT_ERR_UNSUPPORTED, "this experiment build holds the volpathtrace kernel only");
#else
  long long key[10];
  schedule_key(L, key);
  if (int rc = sched_prepare(s, std::max<long long>(L.grid.x, s->split_waves), key, L.st)) return rc;
  if (int rc = sched_wait(s, L.st)) return rc;
  size_t    lds = (size_t)s->stack_cap * VPT_BLOCK * sizeof(int) +                                      // refs-only stack
               (6 * (size_t)s->d.num_sdfs + 7 * (size_t)s->d.num_vol_instances) * sizeof(float4);       // the SDF records
  if (lds > 64 * 1024) return fail(VPT_ERR_UNSUPPORTED, "scene has too many SDFs for the implicit kernel's LDS copy of their records (%d + %d)", s->d.num_sdfs, s->d.num_vol_instances);
  s->last_waves = (int)L.grid.x;
  unsigned long long watchdog_ticks = VPT_K2_WATCHDOG_TICKS;
  if (const char* e = getenv("VPT_K2_WATCHDOG_MS")) watchdog_ticks = strtoull(e, nullptr, 10) * 100000ull;   // tests of the error path
  // the instance for the features this scene's lights have (VPT_FEAT_*): SDF scenes without emissive meshes run one without the mesh-light walks
  const bool lean = (s->light_features & (VPT_FEAT_LARGE_LIGHTS | VPT_FEAT_SMALL_LIGHTS)) == 0 && !getenv("VPT_NO_LEAN");
  int n = L.pr.nsamples, pilot = n / 64 < 1 ? 1 : n / 64 > 16 ? 16 : n / 64;
  int parts[2] = {(!s->order_valid && n >= 16 && !getenv("VPT_K2_NO_PILOT")) ? pilot : n, 0};
  parts[1] = n - parts[0];
  // tile splitting as for K1 (above): K2's launches hold two waves per wave slot at 1280 x 533, so the longest-first schedule ends well
  // above both of its bounds (226 ms against a longest wave of 192 and 191 of work per slot); the costliest tiles as partly filled waves -
  // whose scene rounds run in the group form: four lanes per ray - pack better.  VPT_K2_SPLIT=0 switches it off.
  static const bool k2_split = [] { const char* e = getenv("VPT_K2_SPLIT"); return !e || atoi(e) != 0; }();
  const bool may_split = k2_split && split_mode() != 0;
  for (int part = 0; part < 2 && parts[part] > 0; part++) {
    DParams pr  = L.pr;
    pr.nsamples = parts[part];
    const bool is_pilot = parts[1] > 0 && part == 0;
    if (may_split && !s->split_decided && s->order_valid && s->full_costs) {
      HIP_TRY(hipEventRecord(s->ev_host0, L.st));
      if (int rc = decide_split(s, pr, (int)L.grid.x, s->wave_slots_k2, L.st, split_gain_k2)) return rc;
      HIP_TRY(hipEventRecord(s->ev_host1, L.st));
      s->host_pause = true;
    }
    dim3 grid = s->split_waves > 0 ? dim3((unsigned)s->split_waves) : L.grid;
    sched_cfg sch = {s->order_valid ? s->d_order : nullptr, s->d_cost, s->split_waves > 0 ? s->d_lane_slot : nullptr};
    auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, L.block, lds, L.st, s->d, pr, L.img, L.hit, L.rng, s->stack_cap, sch, s->d_watchdog, watchdog_ticks); };
    if (is_pilot && lean) launch(vpt_render_pilot_kernel<K, VPT_FEAT_SDF_LIGHTS>);
    else if (is_pilot) launch(vpt_render_pilot_kernel<K, VPT_FEAT_ALL>);
    else if (lean) launch(vpt_render_kernel<K, VPT_FEAT_SDF_LIGHTS>);
    else launch(vpt_render_kernel<K, VPT_FEAT_ALL>);
    if (s->split_waves == 0) s->full_costs = pr.nsamples >= 8;
    s->last_waves = (int)grid.x;
    if (int rc = sched_update(s, grid.x, L.st, pr.nsamples)) return rc;
  }
  return VPT_OK;
#endif
}


extern "C" {

int vpt_render_device(vpt_scene* s, const vpt_params* params, const vpt_layout* layout, int nsamples, void* d_image,
    void* d_hits, void* d_rng, void* stream) {
  if (!s || !params || !layout || !d_image || !d_hits || !d_rng) return fail(VPT_ERR_INVALID_ARG, "null argument");
  if (params->shader < 0 || params->shader > VPT_SHADER_IMPLICIT_NORMAL) return fail(VPT_ERR_UNKNOWN_SHADER, "sampler unknown");
  if (params->camera < 0 || params->camera >= s->d.num_cameras) return fail(VPT_ERR_INVALID_ARG, "camera %d out of range", params->camera);
  if (nsamples < 0 || params->bounces < 0) return fail(VPT_ERR_INVALID_ARG, "negative sample/bounce count");
  if (nsamples == 0) return VPT_OK;
  DParams pr;
  if (int rc = make_dparams(params, layout, nsamples, pr)) return rc;
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  dim3   grid((pr.nslots + VPT_BLOCK - 1) / VPT_BLOCK), block(VPT_BLOCK);
  stack_cfg stack;
  if (int rc = stack_config(s, (long long)grid.x * VPT_BLOCK, stack)) return rc;
  auto   img = (float4*)d_image;
  auto   hit = (int*)d_hits;
  auto   rng = (ulonglong2*)d_rng;
  HIP_TRY(hipEventRecord(s->ev0, st));
  s->host_pause = false;
  launch_ctx L = {s, params, pr, grid, block, st, img, hit, rng, stack};
  int rc = VPT_OK;
  switch (params->shader) {
    case VPT_SHADER_VOLPATHTRACE: rc = launch_mesh<K_VOLPATH>(L); break;
    case VPT_SHADER_PATHTRACE: rc = launch_mesh<K_PATH>(L); break;
    case VPT_SHADER_NAIVE: rc = launch_mesh<K_NAIVE>(L); break;
    case VPT_SHADER_EYELIGHT: rc = launch_mesh<K_EYELIGHT>(L); break;
    case VPT_SHADER_NORMAL:
    case VPT_SHADER_TEXCOORD:
    case VPT_SHADER_COLOR: rc = launch_mesh<K_DEBUG>(L); break;
    case VPT_SHADER_IMPLICIT: rc = launch_implicit<K_IMPLICIT>(L); break;
    case VPT_SHADER_IMPLICIT_NORMAL: rc = launch_implicit<K_IMPLICIT_NORMAL>(L); break;
  }
  if (rc != VPT_OK) return rc;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->ev1, st));
  s->timed = true;
  return VPT_OK;
}

int vpt_scene_record_bytes(const vpt_scene* s, int* leaf_bytes, int* attribute_bytes) {
  if (!s || !leaf_bytes || !attribute_bytes) return fail(VPT_ERR_INVALID_ARG, "null argument");
  *leaf_bytes = s->d.tri_prims ? 48 : 64, *attribute_bytes = s->d.tri_attrs ? 64 : 96;
  return VPT_OK;
}

int vpt_last_wave_costs(vpt_scene* s, unsigned* ticks, int capacity, int* count) {
  if (!s || !count || capacity < 0 || (capacity > 0 && !ticks)) return fail(VPT_ERR_INVALID_ARG, "bad argument");
  if (!s->timed) return fail(VPT_ERR_INVALID_ARG, "no launch recorded");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipEventSynchronize(s->ev1));
  long long n = s->last_waves;   // waves of the last launch
  *count = (int)n;
  if (n > capacity) n = capacity;
  if (n > 0) HIP_TRY(hipMemcpy(ticks, s->d_cost, (size_t)n * 4, hipMemcpyDeviceToHost));
  return VPT_OK;
}

// synchronous: waves of the implicit kernel that hit their watchdog since the scene was created (a defect, never a workload)
int vpt_check_watchdog(vpt_scene* s) {
  if (!s) return fail(VPT_ERR_INVALID_ARG, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  unsigned n = 0;
  HIP_TRY(hipMemcpy(&n, s->d_watchdog, 4, hipMemcpyDeviceToHost));
  if (n) return fail(VPT_ERR_HIP, "%u wave(s) of the implicit kernel gave up after their watchdog time: the result is incomplete", n);
  return VPT_OK;
}

int vpt_last_kernel_ms(vpt_scene* s, float* ms) {
  if (!s || !ms) return fail(VPT_ERR_INVALID_ARG, "null argument");
  if (!s->timed) return fail(VPT_ERR_INVALID_ARG, "no launch recorded");
  HIP_TRY(hipEventSynchronize(s->ev1));
  HIP_TRY(hipEventElapsedTime(ms, s->ev0, s->ev1));
  if (s->host_pause) {
    float pause = 0;
    HIP_TRY(hipEventElapsedTime(&pause, s->ev_host0, s->ev_host1));
    *ms -= pause;
  }
  return vpt_check_watchdog(s);
}

int vpt_resolve_device(const vpt_layout* layout, const void* d_tiles_all_ranks, int samples, void* d_image_rowmajor, void* stream) {
  if (!layout || !d_tiles_all_ranks || !d_image_rowmajor || samples <= 0) return fail(VPT_ERR_INVALID_ARG, "bad argument");
  DParams    pr;
  vpt_params dummy = {};
  if (int rc = make_dparams(&dummy, layout, 0, pr)) return rc;
  long long total = (long long)pr.nslots * pr.nranks;
  if (total >= (1LL << 31)) return fail(VPT_ERR_INVALID_ARG, "image too large");
  int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(vpt_resolve_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pr, (const float4*)d_tiles_all_ranks,
      1.0f / (float)samples, (float4*)d_image_rowmajor);
  HIP_TRY(hipGetLastError());
  return VPT_OK;
}

int vpt_resolve_srgb8_device(const vpt_layout* layout, const void* d_tiles_all_ranks, int samples, void* d_rgba8_rowmajor, void* stream) {
  if (!layout || !d_tiles_all_ranks || !d_rgba8_rowmajor || samples <= 0) return fail(VPT_ERR_INVALID_ARG, "bad argument");
  DParams    pr;
  vpt_params dummy = {};
  if (int rc = make_dparams(&dummy, layout, 0, pr)) return rc;
  long long total = (long long)pr.nslots * pr.nranks;
  if (total >= (1LL << 31)) return fail(VPT_ERR_INVALID_ARG, "image too large");
  int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(vpt_resolve_srgb8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pr, (const float4*)d_tiles_all_ranks,
      1.0f / (float)samples, (uchar4*)d_rgba8_rowmajor);
  HIP_TRY(hipGetLastError());
  return VPT_OK;
}

int vpt_render(vpt_scene* s, const vpt_params* params, int nsamples, int width, int height, float* image_rgba,
    int32_t* hits, uint64_t* rng, int* samples_io) {
  if (!s || !params || !image_rgba || !hits || !rng || !samples_io) return fail(VPT_ERR_INVALID_ARG, "null argument");
  if (width <= 0 || height <= 0) return fail(VPT_ERR_INVALID_ARG, "bad image size");
  if (params->shader < 0 || params->shader > VPT_SHADER_IMPLICIT_NORMAL) return fail(VPT_ERR_UNKNOWN_SHADER, "sampler unknown");
  int todo = params->samples - *samples_io;   // no-op once reached, yocto_pathtrace.cpp:1055
  if (nsamples < todo) todo = nsamples;
  if (todo <= 0) return VPT_OK;
  HIP_TRY(hipSetDevice(s->device));
  vpt_layout lay = {width, height, 8, 8, 0, 1};
  long long  slots = vpt_layout_slots(&lay), pixels = (long long)width * height;
  if (slots < 0) return VPT_ERR_INVALID_ARG;
  if (s->staged_slots != slots || s->staged_pixels != pixels) {
    for (void** p : {&s->s_image, &s->s_hits, &s->s_rng, &s->r_image, &s->r_hits, &s->r_rng})
      if (*p) (void)hipFree(*p), *p = nullptr;
    s->staged_slots = s->staged_pixels = 0;
    HIP_TRY(hipMalloc(&s->s_image, (size_t)slots * 16));
    HIP_TRY(hipMalloc(&s->s_hits, (size_t)slots * 4));
    HIP_TRY(hipMalloc(&s->s_rng, (size_t)slots * 16));
    HIP_TRY(hipMalloc(&s->r_image, (size_t)pixels * 16));
    HIP_TRY(hipMalloc(&s->r_hits, (size_t)pixels * 4));
    HIP_TRY(hipMalloc(&s->r_rng, (size_t)pixels * 16));
    s->staged_slots = slots, s->staged_pixels = pixels;
  }
  row_staging rows;   // the handle's buffers: allocated once per frame size, not per call
  rows.image = s->r_image, rows.hits = s->r_hits, rows.rng = s->r_rng;
  if (int rc = state_upload(&lay, image_rgba, hits, rng, s->s_image, s->s_hits, s->s_rng, nullptr, rows)) return rc;
  if (int rc = vpt_render_device(s, params, &lay, todo, s->s_image, s->s_hits, s->s_rng, nullptr)) return rc;
  // the row-major staging still holds the frame that was uploaded, and this single-rank layout owns every pixel
  if (int rc = state_download(&lay, s->s_image, s->s_hits, s->s_rng, image_rgba, hits, rng, nullptr, rows, true)) return rc;
  if (int rc = vpt_check_watchdog(s)) return rc;
  *samples_io += todo;
  return VPT_OK;
}

int vpt_selftest_reciprocal(int device, unsigned long long* mismatches, unsigned long long* fallbacks) {
  if (!mismatches || !fallbacks) return fail(VPT_ERR_INVALID_ARG, "null argument");
  HIP_TRY(hipSetDevice(device));
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, 16));
  HIP_TRY(hipMemset(d, 0, 16));
  hipLaunchKernelGGL(vpt_reciprocal_selftest_kernel, dim3(4096), dim3(256), 0, 0, d);
  unsigned long long h[2] = {0, 0};
  int rc = hipMemcpy(h, d, 16, hipMemcpyDeviceToHost) == hipSuccess ? VPT_OK : fail(VPT_ERR_HIP, "reciprocal self-test failed to run");
  (void)hipFree(d);
  *mismatches = h[0], *fallbacks = h[1];
  return rc;
}

int vpt_intersect(vpt_scene* s, int n, const float* rays, int instance, int32_t* ids, float* uvt) {
  if (!s || !rays || !ids || !uvt || n < 0) return fail(VPT_ERR_INVALID_ARG, "bad argument");
  if (instance < -1 || instance >= s->d.num_instances) return fail(VPT_ERR_INVALID_ARG, "instance %d out of range", instance);
  if (instance >= 0 && s->h_slot_of[(size_t)instance] < 0) return fail(VPT_ERR_INVALID_ARG, "instance %d is not in the scene BVH", instance);
  if (n == 0) return VPT_OK;
  HIP_TRY(hipSetDevice(s->device));
  float* d_rays = nullptr;
  int*   d_ids  = nullptr;
  float* d_uvt  = nullptr;
  auto   release = [&] { (void)hipFree(d_rays), (void)hipFree(d_ids), (void)hipFree(d_uvt); };
  if (hipMalloc((void**)&d_rays, (size_t)n * 24) != hipSuccess || hipMalloc((void**)&d_ids, (size_t)n * 8) != hipSuccess ||
      hipMalloc((void**)&d_uvt, (size_t)n * 12) != hipSuccess || hipMemcpy(d_rays, rays, (size_t)n * 24, hipMemcpyHostToDevice) != hipSuccess) {
    release();
    return fail(VPT_ERR_HIP, "vpt_intersect: device buffers");
  }
  int       blocks = (n + VPT_BLOCK - 1) / VPT_BLOCK;
  stack_cfg stack;
  if (int rc = stack_config(s, (long long)blocks * VPT_BLOCK, stack)) { release(); return rc; }
  size_t lds = (size_t)s->stack_lds4 * 2 * VPT_BLOCK * sizeof(int);
  auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(blocks), dim3(VPT_BLOCK), lds, 0, s->d, n, d_rays, instance, d_ids, d_uvt, stack); };
  if (s->d.tri_prims) {   // a scene of triangles: through the short leaf records, as its path tracers go
    if (stack.spill) launch(vpt_intersect_kernel<true, true>);
    else launch(vpt_intersect_kernel<false, true>);
  } else if (stack.spill) launch(vpt_intersect_kernel<true, false>);
  else launch(vpt_intersect_kernel<false, false>);
  bool ok = hipMemcpy(ids, d_ids, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(uvt, d_uvt, (size_t)n * 12, hipMemcpyDeviceToHost) == hipSuccess;
  release();
  return ok ? VPT_OK : fail(VPT_ERR_HIP, "vpt_intersect failed to run");
}

// ---- known-answer-test entry points (include/vpt_kat.h) ------------------------------------------------------------
static const int k_kat_strides[VPT_KAT_OP_COUNT][2] = {{19, 22}, {15, 10}, {4, 4}, {5, 6}, {7, 5}, {7, 24}, {3, 3}, {7, 3}, {6, 1},
    {6, 1}, {4, 3}, {6, 3}, {7, 4}, {4, 1}, {4, 1}};

int vpt_kat_strides(int op, int* in_stride, int* out_stride) {
  if (op < 0 || op >= VPT_KAT_OP_COUNT || !in_stride || !out_stride) return fail(VPT_ERR_INVALID_ARG, "unknown KAT op %d", op);
  *in_stride = k_kat_strides[op][0], *out_stride = k_kat_strides[op][1];
  return VPT_OK;
}

int vpt_kat(vpt_scene* s, int op, int iparam, int n, const float* in, float* out) {
  if (!s || n < 0 || (n > 0 && (!in || !out))) return fail(VPT_ERR_INVALID_ARG, "bad argument");
  if (op < 0 || op >= VPT_KAT_OP_COUNT) return fail(VPT_ERR_INVALID_ARG, "unknown KAT op %d", op);
  if (n == 0) return VPT_OK;
  const int si = k_kat_strides[op][0], so = k_kat_strides[op][1];
  const DScene& D = s->d;
  // every id a record carries is checked here, so that no batch can index outside the scene's tables
  std::vector<int> aux((size_t)n, 0);
  auto id_ok = [](float v, int count) { return v >= 0 && v < (float)count && v == (float)(int)v; };
  for (int i = 0; i < n; i++) {
    const float* a = in + (size_t)i * si;
    bool ok = true;
    switch (op) {
      case VPT_KAT_LOBES: ok = id_ok(a[0], VPT_MAT_GLTFPBR + 1); break;
      case VPT_KAT_TEXTURE: ok = id_ok(a[0], D.num_textures); break;
      case VPT_KAT_CAMERA: ok = id_ok(a[0], D.num_cameras); break;
      case VPT_KAT_INTERSECT: ok = a[6] == -1.0f || (id_ok(a[6], D.num_instances) && s->h_slot_of[(size_t)a[6]] >= 0); break;
      case VPT_KAT_SURFACE:
        ok = id_ok(a[0], D.num_instances) && id_ok(a[1], s->h_shape_elems[(size_t)s->h_inst_shape[(size_t)a[0]]]);
        if (ok) aux[(size_t)i] = s->h_prim_slot[(size_t)s->h_shape_elem_offset[(size_t)s->h_inst_shape[(size_t)a[0]]] + (size_t)a[1]];
        break;
      case VPT_KAT_SAMPLE_LIGHTS:
      case VPT_KAT_LIGHTS_PDF:
      case VPT_KAT_LIGHTS_PDF_K2: ok = D.num_lights > 0; break;
      case VPT_KAT_SDF_NORMAL: ok = a[0] == 0.0f ? id_ok(a[1], D.num_vol_instances) : (a[0] == 1.0f && id_ok(a[1], D.num_sdfs)); break;
      case VPT_KAT_SPHERETRACE: ok = a[6] == -1.0f || id_ok(a[6], D.num_sdfs); break;
      case VPT_KAT_VOLUME: ok = id_ok(a[0], D.num_volumes); break;
      case VPT_KAT_SDF_FUNCTION: ok = id_ok(a[0], D.num_sdfs); break;
      default: break;
    }
    if (!ok) return fail(VPT_ERR_INVALID_ARG, "KAT op %d record %d: id out of range", op, i);
  }
  if ((op == VPT_KAT_LIGHTS_PDF || op == VPT_KAT_LIGHTS_PDF_K2 || op == VPT_KAT_SPHERETRACE) && (iparam < 0 || iparam > (1 << 20)))
    return fail(VPT_ERR_INVALID_ARG, "KAT op %d: iteration limit %d out of range", op, iparam);
  HIP_TRY(hipSetDevice(s->device));
  float *d_in = nullptr, *d_out = nullptr;
  int*   d_aux = nullptr;
  auto   release = [&] { (void)hipFree(d_in), (void)hipFree(d_out), (void)hipFree(d_aux); };
  if (hipMalloc((void**)&d_in, (size_t)n * si * 4) != hipSuccess || hipMalloc((void**)&d_out, (size_t)n * so * 4) != hipSuccess ||
      hipMalloc((void**)&d_aux, (size_t)n * 4) != hipSuccess || hipMemcpy(d_in, in, (size_t)n * si * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_aux, aux.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) {
    release();
    return fail(VPT_ERR_HIP, "vpt_kat: device buffers");
  }
  int       blocks = (n + VPT_BLOCK - 1) / VPT_BLOCK;
  stack_cfg stack;
  if (int rc = stack_config(s, (long long)blocks * VPT_BLOCK, stack)) { release(); return rc; }
  size_t lds4 = (size_t)s->stack_lds4 * 2 * VPT_BLOCK * sizeof(int), lds2 = (size_t)s->stack_cap * VPT_BLOCK * sizeof(int);
  size_t lds  = lds4 > lds2 ? lds4 : lds2;
  if (stack.spill) hipLaunchKernelGGL(vpt_kat_kernel<true>, dim3(blocks), dim3(VPT_BLOCK), lds, 0, s->d, op, iparam, n, si, so, d_in, d_aux, d_out, stack, s->stack_cap);
  else hipLaunchKernelGGL(vpt_kat_kernel<false>, dim3(blocks), dim3(VPT_BLOCK), lds, 0, s->d, op, iparam, n, si, so, d_in, d_aux, d_out, stack, s->stack_cap);
  bool ok = hipGetLastError() == hipSuccess && hipMemcpy(out, d_out, (size_t)n * so * 4, hipMemcpyDeviceToHost) == hipSuccess;
  release();
  return ok ? VPT_OK : fail(VPT_ERR_HIP, "vpt_kat failed to run");
}

int vpt_spheretrace(vpt_scene* s, int n, const float* rays, int sdf, int maxiter, int32_t* ids, float* t) {
  if (!s || n < 0 || (n > 0 && (!rays || !ids || !t))) return fail(VPT_ERR_INVALID_ARG, "bad argument");
  std::vector<float> in((size_t)n * 7), out((size_t)n * 4);
  for (int i = 0; i < n; i++) {
    memcpy(&in[(size_t)i * 7], rays + (size_t)i * 6, 24);
    in[(size_t)i * 7 + 6] = (float)(sdf < 0 ? -1 : sdf);
  }
  if (int rc = vpt_kat(s, VPT_KAT_SPHERETRACE, maxiter, n, in.data(), out.data())) return rc;
  for (int i = 0; i < n; i++) {
    ids[3 * (size_t)i] = (int)out[4 * (size_t)i], ids[3 * (size_t)i + 1] = (int)out[4 * (size_t)i + 2], ids[3 * (size_t)i + 2] = (int)out[4 * (size_t)i + 3];
    t[i] = out[4 * (size_t)i + 1];
  }
  return VPT_OK;
}

int vpt_eval_lobes(vpt_scene* s, int n, const float* in19, float* out22) { return vpt_kat(s, VPT_KAT_LOBES, 0, n, in19, out22); }

int vpt_selftest_light_cdf(vpt_scene* s, int light, int n, unsigned long long* mismatches, int* indexed) {
  if (!s || !mismatches || !indexed || n <= 0) return fail(VPT_ERR_INVALID_ARG, "bad argument");
  if (light < 0 || light >= s->d.num_lights) return fail(VPT_ERR_INVALID_ARG, "light %d out of range", light);
  HIP_TRY(hipSetDevice(s->device));
  DCdfIndex ix;
  HIP_TRY(hipMemcpy(&ix, s->d.light_index + light, sizeof(ix), hipMemcpyDeviceToHost));
  *indexed = ix.levels > 0 ? (ix.guide_buckets > 0 ? 2 : 1) : 0, *mismatches = 0;
  if (!ix.levels) return VPT_OK;   // short CDFs use the reference's binary search itself
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, 8));
  HIP_TRY(hipMemset(d, 0, 8));
  hipLaunchKernelGGL(vpt_light_cdf_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, s->d, light, n, d);
  int rc = hipMemcpy(mismatches, d, 8, hipMemcpyDeviceToHost) == hipSuccess ? VPT_OK : fail(VPT_ERR_HIP, "light CDF self-test failed to run");
  (void)hipFree(d);
  return rc;
}

#ifdef VPT_WAVE_TIMES
int vpt_debug_wave_times(unsigned long long* out, int nwaves) {
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vpt_wave_times), sizeof(unsigned long long) * 2 * (size_t)nwaves));
  return VPT_OK;
}
int vpt_debug_wave_hw(unsigned* out, int nwaves) {
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vpt_wave_hw), sizeof(unsigned) * (size_t)nwaves));
  return VPT_OK;
}
#endif

#ifdef VPT_K2_STATS
// diagnostic build only: read (and optionally clear) K2's lane statistics (vpt_implicit_kernel.hip.h)
int vpt_debug_k2_stats(unsigned long long* out24, int reset) {
  if (out24) HIP_TRY(hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_k2_stats), sizeof(unsigned long long) * 24));
  if (reset) {
    unsigned long long zero[24] = {};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_k2_stats), zero, sizeof(zero)));
  }
  return VPT_OK;
}
#endif

#ifdef VPT_COUNTERS
// diagnostic build only: read (and optionally clear) the section counters of vpt_mesh_kernel.hip.h
int vpt_debug_hist(unsigned long long* out24, int reset) {
  if (out24) HIP_TRY(hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_vpt_hist), sizeof(unsigned long long) * 24));
  if (reset) {
    unsigned long long zero[24] = {};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_vpt_hist), zero, sizeof(zero)));
  }
  return VPT_OK;
}
int vpt_debug_counts(unsigned long long* out64, int reset) {
  if (out64) HIP_TRY(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_vpt_cnt), sizeof(unsigned long long) * 64));
  if (reset) {
    unsigned long long zero[64] = {};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_vpt_cnt), zero, sizeof(zero)));
  }
  return VPT_OK;
}
#endif

}  // extern "C"
