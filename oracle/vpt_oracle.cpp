// oracle/vpt_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Scalar CPU restatement of the reference's per-sample integrator (everything behind
// pathtrace_samples(), libs/yocto_pathtrace/yocto_pathtrace.cpp:1052-1092) over the flattened
// scene of include/vpt.h.  It is the checker the HIP path is compared with; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product never does.
//
// PINNING: this file is validated against the reference itself (oracle/_ref/ref_driver, built
// from /root/reference by oracle/Makefile): tests/test_oracle_vs_reference.py requires the
// float32 pathtrace_state to be BIT-IDENTICAL on tests/03_volume and the substitute scenes, and
// the committed fixtures under tests/golden/ carry those states to machines without the reference.
//
// Rules that make bit-identity possible:
//  * every RNG draw is an explicit sequential statement in the order g++/MSVC evaluate the
//    reference's call arguments (right-to-left);
//  * float32 arithmetic in the reference's association order; min/max are the ternary forms of
//    yocto_math.h:1355-1356 (NaN-asymmetric); libm calls are the same glibc float functions;
//  * built with -ffp-contract=off and no -march flags (no FMA).
//
// Each function cites the reference lines it follows.

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "vpt.h"
#include "vpt_kat.h"

namespace {

// ------------------------------------------------------------------------------------------------
// math (yocto_math.h)
// ------------------------------------------------------------------------------------------------
const float pif     = (float)3.14159265358979323846;  // :65
const float flt_max = 3.402823466e+38f;
const float flt_eps = 1.1920928955078125e-07f;        // :72
const float ray_eps = 1e-4f;                          // yocto_geometry.h:118

struct v2 { float x, y; };
struct v3 { float x, y, z; };
struct v4 { float x, y, z, w; };
struct m3 { v3 x, y, z; };
struct fr { v3 x, y, z, o; };

inline float fmin_(float a, float b) { return (a < b) ? a : b; }   // :1355
inline float fmax_(float a, float b) { return (a > b) ? a : b; }   // :1356
inline float fabs_(float a) { return a < 0 ? -a : a; }             // :1354
inline float clampf(float a, float lo, float hi) { return fmin_(fmax_(a, lo), hi); }
inline int   clampi(int a, int lo, int hi) { auto m = a > lo ? a : lo; return m < hi ? m : hi; }

inline v3 operator-(v3 a) { return {-a.x, -a.y, -a.z}; }
inline v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline v3 operator*(v3 a, v3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline v3 operator/(v3 a, v3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline v3 operator+(v3 a, float b) { return {a.x + b, a.y + b, a.z + b}; }
inline v3 operator-(v3 a, float b) { return {a.x - b, a.y - b, a.z - b}; }
inline v3 operator*(v3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
inline v3 operator/(v3 a, float b) { return {a.x / b, a.y / b, a.z / b}; }
inline v3 operator+(float a, v3 b) { return {a + b.x, a + b.y, a + b.z}; }
inline v3 operator-(float a, v3 b) { return {a - b.x, a - b.y, a - b.z}; }
inline v3 operator*(float a, v3 b) { return {a * b.x, a * b.y, a * b.z}; }
inline v3& operator+=(v3& a, v3 b) { return a = a + b; }
inline v3& operator*=(v3& a, v3 b) { return a = a * b; }
inline v3& operator*=(v3& a, float b) { return a = a * b; }
inline bool operator==(v3 a, v3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline v2 operator+(v2 a, v2 b) { return {a.x + b.x, a.y + b.y}; }
inline v2 operator-(v2 a, v2 b) { return {a.x - b.x, a.y - b.y}; }
inline v2 operator*(v2 a, float b) { return {a.x * b, a.y * b}; }
inline v2 operator/(v2 a, float b) { return {a.x / b, a.y / b}; }
inline v2 operator-(float a, v2 b) { return {a - b.x, a - b.y}; }
inline v4 operator+(v4 a, v4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline v4 operator*(v4 a, float b) { return {a.x * b, a.y * b, a.z * b, a.w * b}; }

inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // :1608
inline float dot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
inline v3 cross(v3 a, v3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(v3 a) { return std::sqrt(dot(a, a)); }
inline float length(v2 a) { return std::sqrt(dot(a, a)); }
inline v3 normalize(v3 a) { auto l = length(a); return (l != 0) ? a / l : a; }   // :1619
inline float distance_squared(v3 a, v3 b) { return dot(a - b, a - b); }
inline v3 orthonormalize(v3 a, v3 b) { return normalize(a - b * dot(a, b)); }   // :1636
inline v3 reflect(v3 w, v3 n) { return -w + 2 * dot(n, w) * n; }                // :1641
inline v3 refract(v3 w, v3 n, float inv_eta) {                                   // :1644
  auto cosine = dot(n, w);
  auto k      = 1 + inv_eta * inv_eta * (cosine * cosine - 1);
  if (k < 0) return {0, 0, 0};
  return -w * inv_eta + (inv_eta * cosine - std::sqrt(k)) * n;
}
inline v3 vmax(v3 a, float b) { return {fmax_(a.x, b), fmax_(a.y, b), fmax_(a.z, b)}; }
inline v3 vmin3(v3 a, v3 b) { return {fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; }
inline v3 vmax3(v3 a, v3 b) { return {fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
inline v3 vclamp(v3 a, float lo, float hi) { return {clampf(a.x, lo, hi), clampf(a.y, lo, hi), clampf(a.z, lo, hi)}; }
inline v3 vabs(v3 a) { return {fabs_(a.x), fabs_(a.y), fabs_(a.z)}; }
inline v3 vsqrt(v3 a) { return {std::sqrt(a.x), std::sqrt(a.y), std::sqrt(a.z)}; }
inline float lm_exp(float x);
inline float lm_log(float x);
inline v3 vexp(v3 a) { return {lm_exp(a.x), lm_exp(a.y), lm_exp(a.z)}; }
inline v3 vlog(v3 a) { return {lm_log(a.x), lm_log(a.y), lm_log(a.z)}; }
inline float max3(v3 a) { return fmax_(fmax_(a.x, a.y), a.z); }    // :1691
inline float min3(v3 a) { return fmin_(fmin_(a.x, a.y), a.z); }
inline float sum3(v3 a) { return a.x + a.y + a.z; }
inline float mean3(v3 a) { return sum3(a) / 3; }
inline bool finite3(v3 a) { return std::isfinite(a.x) && std::isfinite(a.y) && std::isfinite(a.z); }
inline v3 lerp3(v3 a, v3 b, float u) { return a * (1 - u) + b * u; }
inline v3 xyz(v4 a) { return {a.x, a.y, a.z}; }
inline float comp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

inline v3 mul(const m3& a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // :2775
inline m3 transpose(const m3& a) { return {{a.x.x, a.y.x, a.z.x}, {a.x.y, a.y.y, a.z.y}, {a.x.z, a.y.z, a.z.z}}; }
inline v3 transform_point(const fr& a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.o; }   // :3097
inline v3 transform_vector(const fr& a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline v3 transform_direction(const fr& a, v3 b) { return normalize(transform_vector(a, b)); }
inline v3 transform_direction(const m3& a, v3 b) { return normalize(mul(a, b)); }
inline v3 transform_normal(const fr& a, v3 b) { return normalize(transform_vector(a, b)); }   // rigid (:3111)
// inverse(frame3f, non_rigid): :2948-2956 ; inverse(mat3f) = adjoint * (1/det) :2802-2808
inline fr inverse(const fr& a, bool non_rigid) {
  auto minv = m3{};
  if (non_rigid) {
    auto adj = transpose(m3{cross(a.y, a.z), cross(a.z, a.x), cross(a.x, a.y)});
    auto det = dot(a.x, cross(a.y, a.z));
    auto s   = 1 / det;
    minv     = {adj.x * s, adj.y * s, adj.z * s};
  } else {
    minv = transpose(m3{a.x, a.y, a.z});
  }
  auto o = -mul(minv, a.o);
  return {minv.x, minv.y, minv.z, o};
}
inline m3 basis_fromz(v3 v) {   // :2811-2820
  auto z    = normalize(v);
  auto sign = copysignf(1.0f, z.z);
  auto a    = -1.0f / (sign + z.z);
  auto b    = z.x * z.y * a;
  auto x    = v3{1.0f + sign * z.x * z.x * a, sign * b, -sign * z.x};
  auto y    = v3{b, sign + z.y * z.y * a, -z.y};
  return {x, y, z};
}
inline fr to_fr(const vpt_frame& f) { fr r; std::memcpy(&r, &f, sizeof(r)); return r; }
inline v3 to_v3(const float* p) { return {p[0], p[1], p[2]}; }

// ------------------------------------------------------------------------------------------------
// counters for the algorithmic-bytes model
// ------------------------------------------------------------------------------------------------
enum { C_SAMPLES, C_SCENE_NODES, C_SHAPE_NODES, C_INSTANCE_TESTS, C_QUAD_TESTS, C_TRI_TESTS,
  C_TEXEL_F32, C_TEXEL_U8, C_CDF_PROBES, C_SURFACE_HITS, C_VOLUME_EVENTS, C_BOUNCES, C_SDF_EVALS,
  C_VOXEL_FETCHES, C_LIGHT_PDF_HOPS,
  // sphere-trace profile: marches, their steps by outcome (hit / ran out of iterations / t overflowed), steps taken beyond t = 16
  C_MARCHES, C_STEPS_HIT, C_STEPS_MAXITER, C_STEPS_ESCAPED, C_STEPS_FAR, C_LIGHT_MARCH_STEPS, C_COUNT = 24 };
thread_local uint64_t tl_counters[C_COUNT];
#define COUNT(c) (tl_counters[c]++)
// Per-pixel condition flags (vpt_oracle_render_flags): which numerically ill-conditioned pieces of the reference a
// pixel's paths went through.  Bit 0: sample_lights_pdf evaluated the pdf of an SDF light that the ray hit — the
// reference takes that light's normal by finite differences at the SHADING point with h = flt_eps * distance
// (yocto_pathtrace.cpp:389 -> yocto_sdfs.cpp:67-76), i.e. below the float spacing of the coordinates, so the normal
// is rounding noise and one ulp in the direction changes the pdf by O(1) (tests/test_kat.py demonstrates it).
enum { F_SDF_LIGHT_PDF = 1 };
thread_local unsigned tl_flags;

// ------------------------------------------------------------------------------------------------
// libm behind wrappers.  Normally they ARE the glibc float functions the reference calls (bit-identical results).
// vpt_oracle_render_perturbed() switches on a per-pixel pseudo-random nudge of every result by -1 / 0 / +1 float ulp:
// the only arithmetic in which the HIP path may differ from the reference is its libm (ocml, <= 1-2 ulp from glibc;
// everything else is IEEE-exact and checked bit for bit by the known-answer tables), so "how much does the
// REFERENCE'S OWN pixel move when its libm results move by an ulp" is the measured definition of a pixel on which
// a faithful implementation may differ.  tests/test_gpu_parity.py is strict on every other pixel.
// ------------------------------------------------------------------------------------------------
enum { P_SINCOS = 1, P_ATAN = 2, P_EXPLOG = 4, P_POW = 8 };
thread_local uint64_t tl_pert_state = 0;   // 0: off
thread_local unsigned tl_pert_mask  = 0;
inline float nudge(float r, unsigned cls) {
  if (!tl_pert_state || !(tl_pert_mask & cls) || !std::isfinite(r)) return r;
  tl_pert_state ^= tl_pert_state << 13, tl_pert_state ^= tl_pert_state >> 7, tl_pert_state ^= tl_pert_state << 17;   // xorshift64
  auto pick = (tl_pert_state >> 32) % 3;
  return pick == 0 ? r : std::nextafter(r, pick == 1 ? INFINITY : -INFINITY);
}
inline float lm_sin(float x) { return nudge(std::sin(x), P_SINCOS); }
inline float lm_cos(float x) { return nudge(std::cos(x), P_SINCOS); }
inline float lm_atan(float x) { return nudge(std::atan(x), P_ATAN); }
inline float lm_atan2(float y, float x) { return nudge(std::atan2(y, x), P_ATAN); }
inline float lm_acos(float x) { return nudge(std::acos(x), P_ATAN); }
inline float lm_log(float x) { return nudge(std::log(x), P_EXPLOG); }
inline float lm_exp(float x) { return nudge(std::exp(x), P_EXPLOG); }
inline float lm_pow(float x, float y) { return nudge(std::pow(x, y), P_POW); }

// ------------------------------------------------------------------------------------------------
// rng (yocto_sampling.h:184-222)
// ------------------------------------------------------------------------------------------------
struct rng_t { uint64_t state, inc; };
inline uint32_t advance_rng(rng_t& rng) {
  auto old   = rng.state;
  rng.state  = old * 6364136223846793005ULL + rng.inc;
  auto xs    = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  auto rot   = (uint32_t)(old >> 59u);
  return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
inline float rand1f(rng_t& rng) {
  auto u = (advance_rng(rng) >> 9) | 0x3f800000u;
  float f;
  std::memcpy(&f, &u, 4);
  return f - 1.0f;
}

// ------------------------------------------------------------------------------------------------
// scene access
// ------------------------------------------------------------------------------------------------
struct ray3 { v3 o, d; float tmin, tmax; };
inline ray3 make_ray(v3 o, v3 d) { return {o, d, ray_eps, flt_max}; }
inline v3 ray_point(const ray3& r, float t) { return r.o + r.d * t; }

struct isect { int instance = -1, element = -1; v2 uv = {0, 0}; float distance = 0; bool hit = false; };

struct mpoint {   // material_point, yocto_scene.h:292-304
  int type = VPT_MAT_GLTFPBR;
  v3 emission = {0, 0, 0}, color = {0, 0, 0};
  float opacity = 1, roughness = 0, metallic = 0, ior = 1;
  v3 density = {0, 0, 0}, scattering = {0, 0, 0};
  float scanisotropy = 0, trdepth = 0.01f;
};

using S = const vpt_scene_desc;

inline v3 pos_at(S& s, const vpt_shape& sh, int i) { return to_v3(s.positions + 3 * (int64_t)(sh.position_offset + i)); }
inline v3 nrm_at(S& s, const vpt_shape& sh, int i) { return to_v3(s.normals + 3 * (int64_t)(sh.normal_offset + i)); }
inline v2 uv_at(S& s, const vpt_shape& sh, int i) { auto p = s.texcoords + 2 * (int64_t)(sh.texcoord_offset + i); return {p[0], p[1]}; }
inline v4 col_at(S& s, const vpt_shape& sh, int i) { auto p = s.colors + 4 * (int64_t)(sh.color_offset + i); return {p[0], p[1], p[2], p[3]}; }
inline const int32_t* tri_at(S& s, const vpt_shape& sh, int e) { return s.triangles + 3 * (int64_t)(sh.triangle_offset + e); }
inline const int32_t* quad_at(S& s, const vpt_shape& sh, int e) { return s.quads + 4 * (int64_t)(sh.quad_offset + e); }

// ------------------------------------------------------------------------------------------------
// ray-primitive (yocto_geometry.h:786-868)
// ------------------------------------------------------------------------------------------------
inline bool intersect_triangle(const ray3& ray, v3 p0, v3 p1, v3 p2, v2& uv, float& dist) {
  auto edge1 = p1 - p0, edge2 = p2 - p0;
  auto pvec = cross(ray.d, edge2);
  auto det  = dot(edge1, pvec);
  if (det == 0) return false;
  auto inv_det = 1.0f / det;
  auto tvec = ray.o - p0;
  auto u    = dot(tvec, pvec) * inv_det;
  if (u < 0 || u > 1) return false;
  auto qvec = cross(tvec, edge1);
  auto v    = dot(ray.d, qvec) * inv_det;
  if (v < 0 || u + v > 1) return false;
  auto t = dot(edge2, qvec) * inv_det;
  if (t < ray.tmin || t > ray.tmax) return false;
  uv = {u, v}, dist = t;
  return true;
}
inline bool intersect_quad(const ray3& ray, v3 p0, v3 p1, v3 p2, v3 p3, v2& uv, float& dist) {
  if (p2 == p3) return intersect_triangle(ray, p0, p1, p3, uv, dist);
  auto hit  = false;
  auto tray = ray;
  if (intersect_triangle(tray, p0, p1, p3, uv, dist)) hit = true, tray.tmax = dist;
  if (intersect_triangle(tray, p2, p3, p1, uv, dist)) hit = true, uv = 1 - uv, tray.tmax = dist;
  return hit;
}
inline bool intersect_bbox(const ray3& ray, v3 dinv, const vpt_bvh_node& node) {
  auto it_min = (to_v3(node.bbox_min) - ray.o) * dinv;
  auto it_max = (to_v3(node.bbox_max) - ray.o) * dinv;
  auto tmin = vmin3(it_min, it_max), tmax = vmax3(it_min, it_max);
  auto t0 = fmax_(max3(tmin), ray.tmin);
  auto t1 = fmin_(min3(tmax), ray.tmax);
  t1 *= 1.00000024f;
  return t0 <= t1;
}

// ------------------------------------------------------------------------------------------------
// two-level BVH (yocto_bvh.cpp:699-881)
// ------------------------------------------------------------------------------------------------
bool intersect_shape_bvh(S& s, const vpt_shape& sh, const ray3& ray_, int& element, v2& uv, float& distance) {
  if (sh.num_bvh_nodes == 0) return false;
  auto nodes = s.shape_bvh_nodes + sh.bvh_node_offset;
  auto prims = s.shape_bvh_prims + sh.bvh_prim_offset;
  int stack[128], cur = 0;
  stack[cur++] = 0;
  auto hit = false;
  auto ray = ray_;
  auto dinv  = v3{1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z};
  int dsign[3] = {dinv.x < 0 ? 1 : 0, dinv.y < 0 ? 1 : 0, dinv.z < 0 ? 1 : 0};
  while (cur != 0) {
    auto& node = nodes[stack[--cur]];
    COUNT(C_SHAPE_NODES);
    if (!intersect_bbox(ray, dinv, node)) continue;
    if (node.internal) {
      if (dsign[node.axis] != 0) stack[cur++] = node.start + 0, stack[cur++] = node.start + 1;
      else stack[cur++] = node.start + 1, stack[cur++] = node.start + 0;
    } else if (sh.num_triangles != 0) {
      for (auto idx = node.start; idx < node.start + node.num; idx++) {
        auto t = tri_at(s, sh, prims[idx]);
        COUNT(C_TRI_TESTS);
        if (intersect_triangle(ray, pos_at(s, sh, t[0]), pos_at(s, sh, t[1]), pos_at(s, sh, t[2]), uv, distance))
          hit = true, element = prims[idx], ray.tmax = distance;
      }
    } else if (sh.num_quads != 0) {
      for (auto idx = node.start; idx < node.start + node.num; idx++) {
        auto q = quad_at(s, sh, prims[idx]);
        COUNT(C_QUAD_TESTS);
        if (intersect_quad(ray, pos_at(s, sh, q[0]), pos_at(s, sh, q[1]), pos_at(s, sh, q[2]), pos_at(s, sh, q[3]), uv, distance))
          hit = true, element = prims[idx], ray.tmax = distance;
      }
    }
  }
  return hit;
}
inline ray3 transform_ray(const fr& a, const ray3& b) { return {transform_point(a, b.o), transform_vector(a, b.d), b.tmin, b.tmax}; }

isect intersect_scene_bvh(S& s, const ray3& ray_) {
  auto r = isect{};
  if (s.num_scene_bvh_nodes == 0) return r;
  int stack[128], cur = 0;
  stack[cur++] = 0;
  auto ray = ray_;
  auto dinv  = v3{1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z};
  int dsign[3] = {dinv.x < 0 ? 1 : 0, dinv.y < 0 ? 1 : 0, dinv.z < 0 ? 1 : 0};
  while (cur != 0) {
    auto& node = s.scene_bvh_nodes[stack[--cur]];
    COUNT(C_SCENE_NODES);
    if (!intersect_bbox(ray, dinv, node)) continue;
    if (node.internal) {
      if (dsign[node.axis] != 0) stack[cur++] = node.start + 0, stack[cur++] = node.start + 1;
      else stack[cur++] = node.start + 1, stack[cur++] = node.start + 0;
    } else {
      for (auto idx = node.start; idx < node.start + node.num; idx++) {
        auto& inst    = s.instances[s.scene_bvh_prims[idx]];
        auto  inv_ray = transform_ray(inverse(to_fr(inst.frame), true), ray);
        COUNT(C_INSTANCE_TESTS);
        if (intersect_shape_bvh(s, s.shapes[inst.shape], inv_ray, r.element, r.uv, r.distance))
          r.hit = true, r.instance = s.scene_bvh_prims[idx], ray.tmax = r.distance;
      }
    }
  }
  return r;
}
isect intersect_instance_bvh(S& s, int instance, const ray3& ray) {   // yocto_bvh.cpp:874-881, 1105-1113
  auto  r       = isect{};
  auto& inst    = s.instances[instance];
  auto  inv_ray = transform_ray(inverse(to_fr(inst.frame), true), ray);
  COUNT(C_INSTANCE_TESTS);
  r.hit      = intersect_shape_bvh(s, s.shapes[inst.shape], inv_ray, r.element, r.uv, r.distance);
  r.instance = instance;
  return r;
}

// ------------------------------------------------------------------------------------------------
// textures (yocto_scene.cpp:112-169, yocto_color.h:212-227)
// ------------------------------------------------------------------------------------------------
inline float srgb_to_rgb(float srgb) {
  return (srgb <= 0.04045) ? srgb / 12.92f : std::pow((srgb + 0.055f) / (1.0f + 0.055f), 2.4f);
}
inline v4 lookup_texture(S& s, const vpt_texture& t, int i, int j, bool as_linear) {
  auto color = v4{0, 0, 0, 0};
  auto idx   = t.offset + (int64_t)j * t.width + i;
  if (t.is_float) {
    auto p = s.texels_f + 4 * idx;
    color  = {p[0], p[1], p[2], p[3]};
    COUNT(C_TEXEL_F32);
  } else {
    auto p = s.texels_b + 4 * idx;
    color  = {p[0] / 255.0f, p[1] / 255.0f, p[2] / 255.0f, p[3] / 255.0f};
    COUNT(C_TEXEL_U8);
  }
  if (as_linear && !t.linear) return {srgb_to_rgb(color.x), srgb_to_rgb(color.y), srgb_to_rgb(color.z), color.w};
  return color;
}
v4 eval_texture(S& s, const vpt_texture& t, v2 uv, bool as_linear) {
  if (t.width == 0 || t.height == 0) return {0, 0, 0, 0};
  auto sx = std::fmod(uv.x, 1.0f) * t.width;
  if (sx < 0) sx += t.width;
  auto ty = std::fmod(uv.y, 1.0f) * t.height;
  if (ty < 0) ty += t.height;
  auto i = clampi((int)sx, 0, t.width - 1), j = clampi((int)ty, 0, t.height - 1);
  auto ii = (i + 1) % t.width, jj = (j + 1) % t.height;
  auto u = sx - i, v = ty - j;
  return lookup_texture(s, t, i, j, as_linear) * (1 - u) * (1 - v) + lookup_texture(s, t, i, jj, as_linear) * (1 - u) * v +
         lookup_texture(s, t, ii, j, as_linear) * u * (1 - v) + lookup_texture(s, t, ii, jj, as_linear) * u * v;
}
inline v4 eval_texture(S& s, int texture, v2 uv, bool as_linear) {
  if (texture == VPT_INVALID) return {1, 1, 1, 1};
  return eval_texture(s, s.textures[texture], uv, as_linear);
}

// ------------------------------------------------------------------------------------------------
// shape/instance evaluation (yocto_scene.cpp:279-526, yocto_geometry.h:506-542, 606-640)
// ------------------------------------------------------------------------------------------------
template <typename T>
inline T interpolate_triangle(T p0, T p1, T p2, v2 uv) { return p0 * (1 - uv.x - uv.y) + p1 * uv.x + p2 * uv.y; }
template <typename T>
inline T interpolate_quad(T p0, T p1, T p2, T p3, v2 uv) {
  if (uv.x + uv.y <= 1) return interpolate_triangle(p0, p1, p3, uv);
  return interpolate_triangle(p2, p3, p1, 1 - uv);
}
inline v3 triangle_normal(v3 p0, v3 p1, v3 p2) { return normalize(cross(p1 - p0, p2 - p0)); }
inline v3 quad_normal(v3 p0, v3 p1, v3 p2, v3 p3) { return normalize(triangle_normal(p0, p1, p3) + triangle_normal(p2, p3, p1)); }

v3 eval_position(S& s, const vpt_instance& inst, int element, v2 uv) {
  auto& sh = s.shapes[inst.shape];
  auto  f  = to_fr(inst.frame);
  if (sh.num_triangles != 0) {
    auto t = tri_at(s, sh, element);
    return transform_point(f, interpolate_triangle(pos_at(s, sh, t[0]), pos_at(s, sh, t[1]), pos_at(s, sh, t[2]), uv));
  } else if (sh.num_quads != 0) {
    auto q = quad_at(s, sh, element);
    return transform_point(f, interpolate_quad(pos_at(s, sh, q[0]), pos_at(s, sh, q[1]), pos_at(s, sh, q[2]), pos_at(s, sh, q[3]), uv));
  }
  return {0, 0, 0};
}
v3 eval_element_normal(S& s, const vpt_instance& inst, int element) {
  auto& sh = s.shapes[inst.shape];
  auto  f  = to_fr(inst.frame);
  if (sh.num_triangles != 0) {
    auto t = tri_at(s, sh, element);
    return transform_normal(f, triangle_normal(pos_at(s, sh, t[0]), pos_at(s, sh, t[1]), pos_at(s, sh, t[2])));
  } else if (sh.num_quads != 0) {
    auto q = quad_at(s, sh, element);
    return transform_normal(f, quad_normal(pos_at(s, sh, q[0]), pos_at(s, sh, q[1]), pos_at(s, sh, q[2]), pos_at(s, sh, q[3])));
  }
  return {0, 0, 0};
}
v3 eval_normal(S& s, const vpt_instance& inst, int element, v2 uv) {
  auto& sh = s.shapes[inst.shape];
  if (sh.normal_offset < 0) return eval_element_normal(s, inst, element);
  auto f = to_fr(inst.frame);
  if (sh.num_triangles != 0) {
    auto t = tri_at(s, sh, element);
    return transform_normal(f, normalize(interpolate_triangle(nrm_at(s, sh, t[0]), nrm_at(s, sh, t[1]), nrm_at(s, sh, t[2]), uv)));
  } else if (sh.num_quads != 0) {
    auto q = quad_at(s, sh, element);
    return transform_normal(f, normalize(interpolate_quad(nrm_at(s, sh, q[0]), nrm_at(s, sh, q[1]), nrm_at(s, sh, q[2]), nrm_at(s, sh, q[3]), uv)));
  }
  return {0, 0, 0};
}
v2 eval_texcoord(S& s, const vpt_instance& inst, int element, v2 uv) {
  auto& sh = s.shapes[inst.shape];
  if (sh.texcoord_offset < 0) return uv;
  if (sh.num_triangles != 0) {
    auto t = tri_at(s, sh, element);
    return interpolate_triangle(uv_at(s, sh, t[0]), uv_at(s, sh, t[1]), uv_at(s, sh, t[2]), uv);
  } else if (sh.num_quads != 0) {
    auto q = quad_at(s, sh, element);
    return interpolate_quad(uv_at(s, sh, q[0]), uv_at(s, sh, q[1]), uv_at(s, sh, q[2]), uv_at(s, sh, q[3]), uv);
  }
  return {0, 0};
}
v4 eval_color(S& s, const vpt_instance& inst, int element, v2 uv) {
  auto& sh = s.shapes[inst.shape];
  if (sh.color_offset < 0) return {1, 1, 1, 1};
  if (sh.num_triangles != 0) {
    auto t = tri_at(s, sh, element);
    return interpolate_triangle(col_at(s, sh, t[0]), col_at(s, sh, t[1]), col_at(s, sh, t[2]), uv);
  } else if (sh.num_quads != 0) {
    auto q = quad_at(s, sh, element);
    return interpolate_quad(col_at(s, sh, q[0]), col_at(s, sh, q[1]), col_at(s, sh, q[2]), col_at(s, sh, q[3]), uv);
  }
  return {0, 0, 0, 0};
}
// triangle_tangents_fromuv, yocto_geometry.h:606-629
inline void triangle_tangents_fromuv(v3 p0, v3 p1, v3 p2, v2 uv0, v2 uv1, v2 uv2, v3& tu, v3& tv) {
  auto p = p1 - p0, q = p2 - p0;
  auto sx = uv1.x - uv0.x, sy = uv2.x - uv0.x;
  auto tx = uv1.y - uv0.y, ty = uv2.y - uv0.y;
  auto div = sx * ty - sy * tx;
  if (div != 0) {
    tu = v3{ty * p.x - tx * q.x, ty * p.y - tx * q.y, ty * p.z - tx * q.z} / div;
    tv = v3{sx * q.x - sy * p.x, sx * q.y - sy * p.y, sx * q.z - sy * p.z} / div;
  } else {
    tu = {1, 0, 0}, tv = {0, 1, 0};
  }
}
// eval_element_tangents, yocto_scene.cpp:414-435 (quads always use the (p0,p1,p3) half: uv {0,0})
void eval_element_tangents(S& s, const vpt_instance& inst, int element, v3& tu, v3& tv) {
  auto& sh = s.shapes[inst.shape];
  auto  f  = to_fr(inst.frame);
  tu = {0, 0, 0}, tv = {0, 0, 0};
  if (sh.num_triangles != 0 && sh.texcoord_offset >= 0) {
    auto t = tri_at(s, sh, element);
    triangle_tangents_fromuv(pos_at(s, sh, t[0]), pos_at(s, sh, t[1]), pos_at(s, sh, t[2]), uv_at(s, sh, t[0]), uv_at(s, sh, t[1]), uv_at(s, sh, t[2]), tu, tv);
    tu = transform_direction(f, tu), tv = transform_direction(f, tv);
  } else if (sh.num_quads != 0 && sh.texcoord_offset >= 0) {
    auto q = quad_at(s, sh, element);
    triangle_tangents_fromuv(pos_at(s, sh, q[0]), pos_at(s, sh, q[1]), pos_at(s, sh, q[3]), uv_at(s, sh, q[0]), uv_at(s, sh, q[1]), uv_at(s, sh, q[3]), tu, tv);
    tu = transform_direction(f, tu), tv = transform_direction(f, tv);
  }
}
v3 eval_normalmap(S& s, const vpt_instance& inst, int element, v2 uv) {   // yocto_scene.cpp:437-457
  auto& material = s.materials[inst.material];
  auto  normal   = eval_normal(s, inst, element, uv);
  auto  texcoord = eval_texcoord(s, inst, element, uv);
  if (material.normal_tex != VPT_INVALID) {
    auto normalmap = -1 + 2 * xyz(eval_texture(s, s.textures[material.normal_tex], texcoord, false));
    v3 tu, tv;
    eval_element_tangents(s, inst, element, tu, tv);
    auto fx = orthonormalize(tu, normal);
    auto fy = normalize(cross(normal, fx));
    auto flip_v = dot(fy, tv) < 0;
    normalmap.y *= flip_v ? 1 : -1;
    normal = normalize(fx * normalmap.x + fy * normalmap.y + normal * normalmap.z);
  }
  return normal;
}
v3 eval_shading_normal(S& s, const vpt_instance& inst, int element, v2 uv, v3 outgoing) {   // :476-503
  auto& material = s.materials[inst.material];
  auto  normal   = eval_normal(s, inst, element, uv);
  if (material.normal_tex != VPT_INVALID) normal = eval_normalmap(s, inst, element, uv);
  if (material.type == VPT_MAT_REFRACTIVE) return normal;
  return dot(normal, outgoing) >= 0 ? normal : -normal;
}

const float min_roughness = 0.03f * 0.03f;   // yocto_scene.cpp:191

inline void finish_material(mpoint& point, int type) {
  if (type == VPT_MAT_REFRACTIVE || type == VPT_MAT_VOLUMETRIC || type == VPT_MAT_SUBSURFACE)
    point.density = -vlog(vclamp(point.color, 0.0001f, 1.0f)) / point.trdepth;
  else
    point.density = {0, 0, 0};
  if (point.type == VPT_MAT_MATTE || point.type == VPT_MAT_GLTFPBR || point.type == VPT_MAT_GLOSSY) {
    point.roughness = clampf(point.roughness, min_roughness, 1.0f);
  } else if (type == VPT_MAT_VOLUMETRIC) {
    point.roughness = 0;
  } else {
    if (point.roughness < min_roughness) point.roughness = 0;
  }
}
mpoint eval_material(S& s, const vpt_instance& inst, int element, v2 uv) {   // yocto_scene.cpp:529-579
  auto& m        = s.materials[inst.material];
  auto  texcoord = eval_texcoord(s, inst, element, uv);
  auto emission_tex   = eval_texture(s, m.emission_tex, texcoord, true);
  auto color_shp      = eval_color(s, inst, element, uv);
  auto color_tex      = eval_texture(s, m.color_tex, texcoord, true);
  auto roughness_tex  = eval_texture(s, m.roughness_tex, texcoord, false);
  auto scattering_tex = eval_texture(s, m.scattering_tex, texcoord, true);
  auto point = mpoint{};
  point.type         = m.type;
  point.emission     = to_v3(m.emission) * xyz(emission_tex);
  point.color        = to_v3(m.color) * xyz(color_tex) * xyz(color_shp);
  point.opacity      = m.opacity * color_tex.w * color_shp.w;
  point.metallic     = m.metallic * roughness_tex.z;
  point.roughness    = m.roughness * roughness_tex.y;
  point.roughness    = point.roughness * point.roughness;
  point.ior          = m.ior;
  point.scattering   = to_v3(m.scattering) * xyz(scattering_tex);
  point.scanisotropy = m.scanisotropy;
  point.trdepth      = m.trdepth;
  finish_material(point, m.type);
  return point;
}
mpoint eval_material(S& s, int mat) {   // yocto_scene.cpp:581-619 (no textures)
  auto& m    = s.materials[mat];
  auto point = mpoint{};
  point.type = m.type, point.emission = to_v3(m.emission), point.color = to_v3(m.color);
  point.opacity = m.opacity, point.metallic = m.metallic;
  point.roughness    = m.roughness;
  point.roughness    = point.roughness * point.roughness;
  point.ior = m.ior, point.scattering = to_v3(m.scattering), point.scanisotropy = m.scanisotropy;
  point.trdepth = m.trdepth;
  finish_material(point, m.type);
  return point;
}
inline bool is_delta(const mpoint& m) {   // yocto_scene.cpp:256-264
  return (m.type == VPT_MAT_REFLECTIVE && m.roughness == 0) || (m.type == VPT_MAT_REFRACTIVE && m.roughness == 0) ||
         (m.type == VPT_MAT_TRANSPARENT && m.roughness == 0) || (m.type == VPT_MAT_VOLUMETRIC);
}
inline bool is_volumetric(S& s, const vpt_instance& inst) {   // :249-253, 622-624
  auto t = s.materials[inst.material].type;
  return t == VPT_MAT_REFRACTIVE || t == VPT_MAT_VOLUMETRIC || t == VPT_MAT_SUBSURFACE;
}

// environment (yocto_scene.cpp:634-651)
v3 eval_environment(S& s, v3 direction) {
  auto emission = v3{0, 0, 0};
  for (auto e = 0; e < s.num_environments; e++) {
    auto& env = s.environments[e];
    auto  wl  = transform_direction(inverse(to_fr(env.frame), false), direction);
    auto  texcoord = v2{lm_atan2(wl.z, wl.x) / (2 * pif), lm_acos(clampf(wl.y, -1.0f, 1.0f)) / pif};
    if (texcoord.x < 0) texcoord.x += 1;
    emission += to_v3(env.emission) * xyz(eval_texture(s, env.emission_tex, texcoord, false));
  }
  return emission;
}

// camera (yocto_scene.cpp:67-102)
ray3 eval_camera(const vpt_camera& camera, v2 image_uv, v2 lens_uv) {
  auto film = camera.aspect >= 1 ? v2{camera.film, camera.film / camera.aspect} : v2{camera.film * camera.aspect, camera.film};
  auto frame = to_fr(camera.frame);
  if (!camera.orthographic) {
    auto q  = v3{film.x * (0.5f - image_uv.x), film.y * (image_uv.y - 0.5f), camera.lens};
    auto dc = -normalize(q);
    auto e  = v3{lens_uv.x * camera.aperture / 2, lens_uv.y * camera.aperture / 2, 0};
    auto p  = dc * camera.focus / fabs_(dc.z);
    auto d  = normalize(p - e);
    return make_ray(transform_point(frame, e), transform_direction(frame, d));
  } else {
    auto scale = 1 / camera.lens;
    auto q = v3{film.x * (0.5f - image_uv.x) * scale, film.y * (image_uv.y - 0.5f) * scale, camera.lens};
    auto e = v3{-q.x, -q.y, 0} + v3{lens_uv.x * camera.aperture / 2, lens_uv.y * camera.aperture / 2, 0};
    auto p = v3{-q.x, -q.y, -camera.focus};
    auto d = normalize(p - e);
    return make_ray(transform_point(frame, e), transform_direction(frame, d));
  }
}

// ------------------------------------------------------------------------------------------------
// sampling warps (yocto_sampling.h:251-395)
// ------------------------------------------------------------------------------------------------
inline v3 sample_hemisphere_cos(v3 normal, v2 ruv) {
  auto z = std::sqrt(ruv.y);
  auto r = std::sqrt(1 - z * z);
  auto phi = 2 * pif * ruv.x;
  auto local = v3{r * lm_cos(phi), r * lm_sin(phi), z};
  return transform_direction(basis_fromz(normal), local);
}
inline float sample_hemisphere_cos_pdf(v3 normal, v3 direction) {
  auto cosw = dot(normal, direction);
  return (cosw <= 0) ? 0 : cosw / pif;
}
inline v3 sample_sphere(v2 ruv) {
  auto z = 2 * ruv.y - 1;
  auto r = std::sqrt(clampf(1 - z * z, 0.0f, 1.0f));
  auto phi = 2 * pif * ruv.x;
  return {r * lm_cos(phi), r * lm_sin(phi), z};
}
inline v2 sample_triangle(v2 ruv) { return {1 - std::sqrt(ruv.x), ruv.y * std::sqrt(ruv.x)}; }
inline int sample_uniform(int size, float r) { return clampi((int)(r * size), 0, size - 1); }
inline int sample_discrete(const float* cdf, int n, float r) {   // :385-390
  auto back = cdf[n - 1];
  r = clampf(r * back, (float)0, back - (float)0.00001);
  auto lo = 0, len = n;   // std::upper_bound
  while (len > 0) {
    auto half = len >> 1;
    COUNT(C_CDF_PROBES);
    if (!(r < cdf[lo + half])) lo += half + 1, len -= half + 1;
    else len = half;
  }
  return clampi(lo, 0, n - 1);
}
inline float sample_discrete_pdf(const float* cdf, int idx) { return idx == 0 ? cdf[0] : cdf[idx] - cdf[idx - 1]; }

// ------------------------------------------------------------------------------------------------
// BSDF lobes (yocto_shading.h)
// ------------------------------------------------------------------------------------------------
inline bool same_hemisphere(v3 n, v3 o, v3 i) { return dot(n, o) * dot(n, i) >= 0; }   // :296
inline v3 fresnel_schlick(v3 specular, v3 normal, v3 outgoing) {   // :302-308
  if (specular == v3{0, 0, 0}) return {0, 0, 0};
  auto cosine = dot(normal, outgoing);
  return specular + (1 - specular) * lm_pow(clampf(1 - fabs_(cosine), 0.0f, 1.0f), 5.0f);
}
inline float fresnel_dielectric(float eta, v3 normal, v3 outgoing) {   // :311-331
  auto cosw = fabs_(dot(normal, outgoing));
  auto sin2 = 1 - cosw * cosw;
  auto eta2 = eta * eta;
  auto cos2t = 1 - sin2 / eta2;
  if (cos2t < 0) return 1;
  auto t0 = std::sqrt(cos2t);
  auto t1 = eta * t0;
  auto t2 = eta * cosw;
  auto rs = (cosw - t1) / (cosw + t1);
  auto rp = (t0 - t2) / (t0 + t2);
  return (rs * rs + rp * rp) / 2;
}
inline v3 fresnel_conductor(v3 eta, v3 etak, v3 normal, v3 outgoing) {   // :334-359
  auto cosw = dot(normal, outgoing);
  if (cosw <= 0) return {0, 0, 0};
  cosw = clampf(cosw, (float)-1, (float)1);
  auto cos2 = cosw * cosw;
  auto sin2 = clampf(1 - cos2, (float)0, (float)1);
  auto eta2 = eta * eta, etak2 = etak * etak;
  auto t0 = eta2 - etak2 - sin2;
  auto a2plusb2 = vsqrt(t0 * t0 + 4 * eta2 * etak2);
  auto t1 = a2plusb2 + cos2;
  auto a  = vsqrt((a2plusb2 + t0) / 2);
  auto t2 = 2 * a * cosw;
  auto rs = (t1 - t2) / (t1 + t2);
  auto t3 = cos2 * a2plusb2 + sin2 * sin2;
  auto t4 = t2 * sin2;
  auto rp = rs * (t3 - t4) / (t3 + t4);
  return (rp + rs) / 2;
}
inline v3 eta_to_reflectivity(v3 eta) { return ((eta - 1) * (eta - 1)) / ((eta + 1) * (eta + 1)); }   // :362
inline v3 reflectivity_to_eta(v3 r_) {   // :366-369
  auto r = vclamp(r_, 0.0f, 0.99f);
  return (1 + vsqrt(r)) / (1 - vsqrt(r));
}
inline float microfacet_distribution(float roughness, v3 normal, v3 halfway) {   // :402-417 (ggx)
  auto cosine = dot(normal, halfway);
  if (cosine <= 0) return 0;
  auto roughness2 = roughness * roughness;
  auto cosine2 = cosine * cosine;
  return roughness2 / (pif * (cosine2 * roughness2 + 1 - cosine2) * (cosine2 * roughness2 + 1 - cosine2));
}
inline float microfacet_shadowing1(float roughness, v3 normal, v3 halfway, v3 direction) {   // :420-438
  auto cosine = dot(normal, direction);
  auto cosineh = dot(halfway, direction);
  if (cosine * cosineh <= 0) return 0;
  auto roughness2 = roughness * roughness;
  auto cosine2 = cosine * cosine;
  return 2 * fabs_(cosine) / (fabs_(cosine) + std::sqrt(cosine2 - roughness2 * cosine2 + roughness2));
}
inline float microfacet_shadowing(float roughness, v3 normal, v3 halfway, v3 outgoing, v3 incoming) {
  return microfacet_shadowing1(roughness, normal, halfway, outgoing) * microfacet_shadowing1(roughness, normal, halfway, incoming);
}
inline v3 sample_microfacet(float roughness, v3 normal, v2 rn) {   // :450-463 (ggx)
  auto phi = 2 * pif * rn.x;
  auto theta = lm_atan(roughness * std::sqrt(rn.y / (1 - rn.y)));
  auto local = v3{lm_cos(phi) * lm_sin(theta), lm_sin(phi) * lm_sin(theta), lm_cos(theta)};
  return transform_direction(basis_fromz(normal), local);
}
inline float sample_microfacet_pdf(float roughness, v3 normal, v3 halfway) {   // :466-471
  auto cosine = dot(normal, halfway);
  if (cosine < 0) return 0;
  return microfacet_distribution(roughness, normal, halfway) * cosine;
}
// matte :543-562
inline v3 eval_matte(v3 color, v3 n, v3 o, v3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return {0, 0, 0};
  return color / pif * fabs_(dot(n, i));
}
inline v3 sample_matte(v3 n, v3 o, v2 rn) {
  auto up = dot(n, o) <= 0 ? -n : n;
  return sample_hemisphere_cos(up, rn);
}
inline float sample_matte_pdf(v3 n, v3 o, v3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return 0;
  auto up = dot(n, o) <= 0 ? -n : n;
  return sample_hemisphere_cos_pdf(up, i);
}
// glossy :565-605
inline v3 eval_glossy(v3 color, float ior, float roughness, v3 n, v3 o, v3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return {0, 0, 0};
  auto up = dot(n, o) <= 0 ? -n : n;
  auto F1 = fresnel_dielectric(ior, up, o);
  auto h  = normalize(i + o);
  auto F  = fresnel_dielectric(ior, h, i);
  auto D  = microfacet_distribution(roughness, up, h);
  auto G  = microfacet_shadowing(roughness, up, h, o, i);
  return color * (1 - F1) / pif * fabs_(dot(up, i)) + v3{1, 1, 1} * F * D * G / (4 * dot(up, o) * dot(up, i)) * fabs_(dot(up, i));
}
inline v3 sample_glossy(float ior, float roughness, v3 n, v3 o, float rnl, v2 rn) {
  auto up = dot(n, o) <= 0 ? -n : n;
  if (rnl < fresnel_dielectric(ior, up, o)) {
    auto h = sample_microfacet(roughness, up, rn);
    auto i = reflect(o, h);
    if (!same_hemisphere(up, o, i)) return {0, 0, 0};
    return i;
  }
  return sample_hemisphere_cos(up, rn);
}
inline float sample_glossy_pdf(float ior, float roughness, v3 n, v3 o, v3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return 0;
  auto up = dot(n, o) <= 0 ? -n : n;
  auto h  = normalize(o + i);
  auto F  = fresnel_dielectric(ior, up, o);
  return F * sample_microfacet_pdf(roughness, up, h) / (4 * fabs_(dot(o, h))) + (1 - F) * sample_hemisphere_cos_pdf(up, i);
}
// reflective (color parametrisation) :608-640, 678-698
inline v3 eval_reflective(v3 color, float roughness, v3 n, v3 o, v3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return {0, 0, 0};
  auto up = dot(n, o) <= 0 ? -n : n;
  auto h  = normalize(i + o);
  auto F  = fresnel_conductor(reflectivity_to_eta(color), {0, 0, 0}, h, i);
  auto D  = microfacet_distribution(roughness, up, h);
  auto G  = microfacet_shadowing(roughness, up, h, o, i);
  return F * D * G / (4 * dot(up, o) * dot(up, i)) * fabs_(dot(up, i));
}
inline v3 sample_reflective(float roughness, v3 n, v3 o, v2 rn) {
  auto up = dot(n, o) <= 0 ? -n : n;
  auto h  = sample_microfacet(roughness, up, rn);
  auto i  = reflect(o, h);
  if (!same_hemisphere(up, o, i)) return {0, 0, 0};
  return i;
}
inline float sample_reflective_pdf(float roughness, v3 n, v3 o, v3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return 0;
  auto up = dot(n, o) <= 0 ? -n : n;
  auto h  = normalize(o + i);
  return sample_microfacet_pdf(roughness, up, h) / (4 * fabs_(dot(o, h)));
}
inline v3 eval_reflective_delta(v3 color, v3 n, v3 o, v3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return {0, 0, 0};
  auto up = dot(n, o) <= 0 ? -n : n;
  return fresnel_conductor(reflectivity_to_eta(color), {0, 0, 0}, up, o);
}
inline v3 sample_reflective_delta(v3 n, v3 o) {
  auto up = dot(n, o) <= 0 ? -n : n;
  return reflect(o, up);
}
inline float sample_reflective_delta_pdf(v3 n, v3 o, v3 i) { return dot(n, i) * dot(n, o) <= 0 ? 0.0f : 1.0f; }
// gltfpbr :723-772
inline v3 eval_gltfpbr(v3 color, float ior, float roughness, float metallic, v3 n, v3 o, v3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return {0, 0, 0};
  auto reflectivity = lerp3(eta_to_reflectivity(v3{ior, ior, ior}), color, metallic);
  auto up = dot(n, o) <= 0 ? -n : n;
  auto F1 = fresnel_schlick(reflectivity, up, o);
  auto h  = normalize(i + o);
  auto F  = fresnel_schlick(reflectivity, h, i);
  auto D  = microfacet_distribution(roughness, up, h);
  auto G  = microfacet_shadowing(roughness, up, h, o, i);
  return color * (1 - metallic) * (1 - F1) / pif * fabs_(dot(up, i)) + F * D * G / (4 * dot(up, o) * dot(up, i)) * fabs_(dot(up, i));
}
inline v3 sample_gltfpbr(v3 color, float ior, float roughness, float metallic, v3 n, v3 o, float rnl, v2 rn) {
  auto up = dot(n, o) <= 0 ? -n : n;
  auto reflectivity = lerp3(eta_to_reflectivity(v3{ior, ior, ior}), color, metallic);
  if (rnl < mean3(fresnel_schlick(reflectivity, up, o))) {
    auto h = sample_microfacet(roughness, up, rn);
    auto i = reflect(o, h);
    if (!same_hemisphere(up, o, i)) return {0, 0, 0};
    return i;
  }
  return sample_hemisphere_cos(up, rn);
}
inline float sample_gltfpbr_pdf(v3 color, float ior, float roughness, float metallic, v3 n, v3 o, v3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return 0;
  auto up = dot(n, o) <= 0 ? -n : n;
  auto h  = normalize(o + i);
  auto reflectivity = lerp3(eta_to_reflectivity(v3{ior, ior, ior}), color, metallic);
  auto F = mean3(fresnel_schlick(reflectivity, up, o));
  return F * sample_microfacet_pdf(roughness, up, h) / (4 * fabs_(dot(o, h))) + (1 - F) * sample_hemisphere_cos_pdf(up, i);
}
// transparent :775-867
inline v3 eval_transparent(v3 color, float ior, float roughness, v3 n, v3 o, v3 i) {
  auto up = dot(n, o) <= 0 ? -n : n;
  if (dot(n, i) * dot(n, o) >= 0) {
    auto h = normalize(i + o);
    auto F = fresnel_dielectric(ior, h, o);
    auto D = microfacet_distribution(roughness, up, h);
    auto G = microfacet_shadowing(roughness, up, h, o, i);
    return v3{1, 1, 1} * F * D * G / (4 * dot(up, o) * dot(up, i)) * fabs_(dot(up, i));
  } else {
    auto reflected = reflect(-i, up);
    auto h = normalize(reflected + o);
    auto F = fresnel_dielectric(ior, h, o);
    auto D = microfacet_distribution(roughness, up, h);
    auto G = microfacet_shadowing(roughness, up, h, o, reflected);
    return color * (1 - F) * D * G / (4 * dot(up, o) * dot(up, reflected)) * (fabs_(dot(up, reflected)));
  }
}
inline v3 sample_transparent(float ior, float roughness, v3 n, v3 o, float rnl, v2 rn) {
  auto up = dot(n, o) <= 0 ? -n : n;
  auto h  = sample_microfacet(roughness, up, rn);
  if (rnl < fresnel_dielectric(ior, h, o)) {
    auto i = reflect(o, h);
    if (!same_hemisphere(up, o, i)) return {0, 0, 0};
    return i;
  } else {
    auto reflected = reflect(o, h);
    auto i = -reflect(reflected, up);
    if (same_hemisphere(up, o, i)) return {0, 0, 0};
    return i;
  }
}
inline float sample_transparent_pdf(float ior, float roughness, v3 n, v3 o, v3 i) {
  auto up = dot(n, o) <= 0 ? -n : n;
  if (dot(n, i) * dot(n, o) >= 0) {
    auto h = normalize(i + o);
    return fresnel_dielectric(ior, h, o) * sample_microfacet_pdf(roughness, up, h) / (4 * fabs_(dot(o, h)));
  } else {
    auto reflected = reflect(-i, up);
    auto h = normalize(reflected + o);
    auto d = (1 - fresnel_dielectric(ior, h, o)) * sample_microfacet_pdf(roughness, up, h);
    return d / (4 * fabs_(dot(o, h)));
  }
}
inline v3 eval_transparent_delta(v3 color, float ior, v3 n, v3 o, v3 i) {
  auto up = dot(n, o) <= 0 ? -n : n;
  if (dot(n, i) * dot(n, o) >= 0) return v3{1, 1, 1} * fresnel_dielectric(ior, up, o);
  return color * (1 - fresnel_dielectric(ior, up, o));
}
inline v3 sample_transparent_delta(float ior, v3 n, v3 o, float rnl) {
  auto up = dot(n, o) <= 0 ? -n : n;
  if (rnl < fresnel_dielectric(ior, up, o)) return reflect(o, up);
  return -o;
}
inline float sample_transparent_delta_pdf(float ior, v3 n, v3 o, v3 i) {
  auto up = dot(n, o) <= 0 ? -n : n;
  if (dot(n, i) * dot(n, o) >= 0) return fresnel_dielectric(ior, up, o);
  return 1 - fresnel_dielectric(ior, up, o);
}
// refractive :870-988
inline v3 eval_refractive(float ior, float roughness, v3 n, v3 o, v3 i) {
  auto entering = dot(n, o) >= 0;
  auto up = entering ? n : -n;
  auto rel_ior = entering ? ior : (1 / ior);
  if (dot(n, i) * dot(n, o) >= 0) {
    auto h = normalize(i + o);
    auto F = fresnel_dielectric(rel_ior, h, o);
    auto D = microfacet_distribution(roughness, up, h);
    auto G = microfacet_shadowing(roughness, up, h, o, i);
    return v3{1, 1, 1} * F * D * G / fabs_(4 * dot(n, o) * dot(n, i)) * fabs_(dot(n, i));
  } else {
    auto h = -normalize(rel_ior * i + o) * (entering ? 1.0f : -1.0f);
    auto F = fresnel_dielectric(rel_ior, h, o);
    auto D = microfacet_distribution(roughness, up, h);
    auto G = microfacet_shadowing(roughness, up, h, o, i);
    return v3{1, 1, 1} * fabs_((dot(o, h) * dot(i, h)) / (dot(o, n) * dot(i, n))) * (1 - F) * D * G /
           lm_pow(rel_ior * dot(h, i) + dot(h, o), 2.0f) * fabs_(dot(n, i));
  }
}
inline v3 sample_refractive(float ior, float roughness, v3 n, v3 o, float rnl, v2 rn) {
  auto entering = dot(n, o) >= 0;
  auto up = entering ? n : -n;
  auto h  = sample_microfacet(roughness, up, rn);
  if (rnl < fresnel_dielectric(entering ? ior : (1 / ior), h, o)) {
    auto i = reflect(o, h);
    if (!same_hemisphere(up, o, i)) return {0, 0, 0};
    return i;
  } else {
    auto i = refract(o, h, entering ? (1 / ior) : ior);
    if (same_hemisphere(up, o, i)) return {0, 0, 0};
    return i;
  }
}
inline float sample_refractive_pdf(float ior, float roughness, v3 n, v3 o, v3 i) {
  auto entering = dot(n, o) >= 0;
  auto up = entering ? n : -n;
  auto rel_ior = entering ? ior : (1 / ior);
  if (dot(n, i) * dot(n, o) >= 0) {
    auto h = normalize(i + o);
    return fresnel_dielectric(rel_ior, h, o) * sample_microfacet_pdf(roughness, up, h) / (4 * fabs_(dot(o, h)));
  } else {
    auto h = -normalize(rel_ior * i + o) * (entering ? 1.0f : -1.0f);
    return (1 - fresnel_dielectric(rel_ior, h, o)) * sample_microfacet_pdf(roughness, up, h) * fabs_(dot(h, i)) /
           lm_pow(rel_ior * dot(h, i) + dot(h, o), 2.0f);
  }
}
inline v3 eval_refractive_delta(float ior, v3 n, v3 o, v3 i) {
  if (fabs_(ior - 1) < 1e-3) return dot(n, i) * dot(n, o) <= 0 ? v3{1, 1, 1} : v3{0, 0, 0};
  auto entering = dot(n, o) >= 0;
  auto up = entering ? n : -n;
  auto rel_ior = entering ? ior : (1 / ior);
  if (dot(n, i) * dot(n, o) >= 0) return v3{1, 1, 1} * fresnel_dielectric(rel_ior, up, o);
  return v3{1, 1, 1} * (1 / (rel_ior * rel_ior)) * (1 - fresnel_dielectric(rel_ior, up, o));
}
inline v3 sample_refractive_delta(float ior, v3 n, v3 o, float rnl) {
  if (fabs_(ior - 1) < 1e-3) return -o;
  auto entering = dot(n, o) >= 0;
  auto up = entering ? n : -n;
  auto rel_ior = entering ? ior : (1 / ior);
  if (rnl < fresnel_dielectric(rel_ior, up, o)) return reflect(o, up);
  return refract(o, up, 1 / rel_ior);
}
inline float sample_refractive_delta_pdf(float ior, v3 n, v3 o, v3 i) {
  if (fabs_(ior - 1) < 1e-3) return dot(n, i) * dot(n, o) < 0 ? 1.0f : 0.0f;
  auto entering = dot(n, o) >= 0;
  auto up = entering ? n : -n;
  auto rel_ior = entering ? ior : (1 / ior);
  if (dot(n, i) * dot(n, o) >= 0) return fresnel_dielectric(rel_ior, up, o);
  return (1 - fresnel_dielectric(rel_ior, up, o));
}
// passthrough :1016-1039
inline v3 eval_passthrough(v3 n, v3 o, v3 i) { return dot(n, i) * dot(n, o) >= 0 ? v3{0, 0, 0} : v3{1, 1, 1}; }
inline float sample_passthrough_pdf(v3 n, v3 o, v3 i) { return dot(n, i) * dot(n, o) >= 0 ? 0.0f : 1.0f; }
// media :1047-1102
inline v3 eval_transmittance(v3 density, float distance) { return vexp(-density * distance); }
inline float sample_transmittance(v3 density, float max_distance, float rl, float rd) {
  auto channel  = clampi((int)(rl * 3), 0, 2);
  auto dc       = comp(density, channel);
  auto distance = (dc == 0) ? flt_max : -lm_log(1 - rd) / dc;
  return fmin_(distance, max_distance);
}
inline float sample_transmittance_pdf(v3 density, float distance, float max_distance) {
  if (distance < max_distance) return sum3(density * vexp(-density * distance)) / 3;
  return sum3(vexp(-density * max_distance)) / 3;
}
inline float eval_phasefunction(float anisotropy, v3 outgoing, v3 incoming) {
  auto cosine = -dot(outgoing, incoming);
  auto denom  = 1 + anisotropy * anisotropy - 2 * anisotropy * cosine;
  return (1 - anisotropy * anisotropy) / (4 * pif * denom * std::sqrt(denom));
}
inline v3 sample_phasefunction(float anisotropy, v3 outgoing, v2 rn) {
  auto cos_theta = 0.0f;
  if (fabs_(anisotropy) < 1e-3f) {
    cos_theta = 1 - 2 * rn.y;
  } else {
    auto square = (1 - anisotropy * anisotropy) / (1 + anisotropy - 2 * anisotropy * rn.y);
    cos_theta   = (1 + anisotropy * anisotropy - square * square) / (2 * anisotropy);
  }
  auto sin_theta = std::sqrt(fmax_(0.0f, 1 - cos_theta * cos_theta));
  auto phi = 2 * pif * rn.x;
  auto local = v3{sin_theta * lm_cos(phi), sin_theta * lm_sin(phi), cos_theta};
  return mul(basis_fromz(-outgoing), local);
}

// material dispatch (yocto_pathtrace.cpp:86-255)
inline v3 eval_emission(const mpoint& m, v3 normal, v3 outgoing) { return dot(normal, outgoing) >= 0 ? m.emission : v3{0, 0, 0}; }
v3 eval_bsdfcos(const mpoint& m, v3 n, v3 o, v3 i) {
  if (m.roughness == 0) return {0, 0, 0};
  switch (m.type) {
    case VPT_MAT_MATTE: return eval_matte(m.color, n, o, i);
    case VPT_MAT_GLOSSY: return eval_glossy(m.color, m.ior, m.roughness, n, o, i);
    case VPT_MAT_REFLECTIVE: return eval_reflective(m.color, m.roughness, n, o, i);
    case VPT_MAT_TRANSPARENT: return eval_transparent(m.color, m.ior, m.roughness, n, o, i);
    case VPT_MAT_REFRACTIVE:
    case VPT_MAT_SUBSURFACE: return eval_refractive(m.ior, m.roughness, n, o, i);
    case VPT_MAT_GLTFPBR: return eval_gltfpbr(m.color, m.ior, m.roughness, m.metallic, n, o, i);
    default: return {0, 0, 0};
  }
}
v3 eval_delta(const mpoint& m, v3 n, v3 o, v3 i) {
  if (m.roughness != 0) return {0, 0, 0};
  switch (m.type) {
    case VPT_MAT_REFLECTIVE: return eval_reflective_delta(m.color, n, o, i);
    case VPT_MAT_TRANSPARENT: return eval_transparent_delta(m.color, m.ior, n, o, i);
    case VPT_MAT_REFRACTIVE: return eval_refractive_delta(m.ior, n, o, i);
    case VPT_MAT_VOLUMETRIC: return eval_passthrough(n, o, i);
    default: return {0, 0, 0};
  }
}
v3 sample_bsdfcos(const mpoint& m, v3 n, v3 o, float rnl, v2 rn) {
  if (m.roughness == 0) return {0, 0, 0};
  switch (m.type) {
    case VPT_MAT_MATTE: return sample_matte(n, o, rn);
    case VPT_MAT_GLOSSY: return sample_glossy(m.ior, m.roughness, n, o, rnl, rn);
    case VPT_MAT_REFLECTIVE: return sample_reflective(m.roughness, n, o, rn);
    case VPT_MAT_TRANSPARENT: return sample_transparent(m.ior, m.roughness, n, o, rnl, rn);
    case VPT_MAT_REFRACTIVE:
    case VPT_MAT_SUBSURFACE: return sample_refractive(m.ior, m.roughness, n, o, rnl, rn);
    case VPT_MAT_GLTFPBR: return sample_gltfpbr(m.color, m.ior, m.roughness, m.metallic, n, o, rnl, rn);
    default: return {0, 0, 0};
  }
}
v3 sample_delta(const mpoint& m, v3 n, v3 o, float rnl) {
  if (m.roughness != 0) return {0, 0, 0};
  switch (m.type) {
    case VPT_MAT_REFLECTIVE: return sample_reflective_delta(n, o);
    case VPT_MAT_TRANSPARENT: return sample_transparent_delta(m.ior, n, o, rnl);
    case VPT_MAT_REFRACTIVE: return sample_refractive_delta(m.ior, n, o, rnl);
    case VPT_MAT_VOLUMETRIC: return -o;
    default: return {0, 0, 0};
  }
}
float sample_bsdfcos_pdf(const mpoint& m, v3 n, v3 o, v3 i) {
  if (m.roughness == 0) return 0;
  switch (m.type) {
    case VPT_MAT_MATTE: return sample_matte_pdf(n, o, i);
    case VPT_MAT_GLOSSY: return sample_glossy_pdf(m.ior, m.roughness, n, o, i);
    case VPT_MAT_REFLECTIVE: return sample_reflective_pdf(m.roughness, n, o, i);
    case VPT_MAT_TRANSPARENT: return sample_transparent_pdf(m.ior, m.roughness, n, o, i);
    case VPT_MAT_REFRACTIVE:
    case VPT_MAT_SUBSURFACE: return sample_refractive_pdf(m.ior, m.roughness, n, o, i);
    case VPT_MAT_GLTFPBR: return sample_gltfpbr_pdf(m.color, m.ior, m.roughness, m.metallic, n, o, i);
    default: return 0;
  }
}
float sample_delta_pdf(const mpoint& m, v3 n, v3 o, v3 i) {
  if (m.roughness != 0) return 0;
  switch (m.type) {
    case VPT_MAT_REFLECTIVE: return sample_reflective_delta_pdf(n, o, i);
    case VPT_MAT_TRANSPARENT: return sample_transparent_delta_pdf(m.ior, n, o, i);
    case VPT_MAT_REFRACTIVE: return sample_refractive_delta_pdf(m.ior, n, o, i);
    case VPT_MAT_VOLUMETRIC: return sample_passthrough_pdf(n, o, i);
    default: return 0;
  }
}
inline v3 eval_scattering(const mpoint& m, v3 o, v3 i) { return m.density * m.scattering * eval_phasefunction(m.scanisotropy, i, o); }
inline v3 sample_scattering(const mpoint& m, v3 o, v2 rn) { return sample_phasefunction(m.scanisotropy, o, rn); }
inline float sample_scattering_pdf(const mpoint& m, v3 o, v3 i) { return eval_phasefunction(m.scanisotropy, o, i); }

// ------------------------------------------------------------------------------------------------
// SDF module (yocto_sdfs.h:43-80, yocto_sdfs.cpp:7-127, yocto_pathtrace.cpp:259-307)
// ------------------------------------------------------------------------------------------------
inline float sd_box(v3 p, v3 b) {
  auto d = vabs(p) - b;
  return fmin_(fmax_(d.x, fmax_(d.y, d.z)), 0.0f) + length(vmax(d, 0.0f));
}
inline float sd_bbox(v3 p, v3 b, float e) {
  p      = vabs(p) - b;
  auto q = vabs(p + e) - e;
  return fmin_(fmin_(length(vmax(v3{p.x, q.y, q.z}, 0.0f)) + fmin_(fmax_(p.x, fmax_(q.y, q.z)), 0.0f),
                   length(vmax(v3{q.x, p.y, q.z}, 0.0f)) + fmin_(fmax_(q.x, fmax_(p.y, q.z)), 0.0f)),
      length(vmax(v3{q.x, q.y, p.z}, 0.0f)) + fmin_(fmax_(q.x, fmax_(q.y, p.z)), 0.0f));
}
inline float sd_torus(v3 p, float r1, float r2) { return length(v2{length(v2{p.x, p.z}) - r1, p.y}) - r2; }
inline float sd_capped_cone(v3 p, float h, float r1, float r2) {
  auto q  = v2{length(v2{p.x, p.z}), p.y};
  auto k1 = v2{r2, h};
  auto k2 = v2{r2 - r1, 2.0f * h};
  auto ca = v2{q.x - fmin_(q.x, (q.y < 0.0) ? r1 : r2), fabs_(q.y) - h};
  auto cb = q - k1 + k2 * clampf(dot(k1 - q, k2) / dot(k2, k2), 0.0f, 1.0f);
  float s = (cb.x < 0.0 && ca.y < 0.0) ? -1.0 : 1.0;
  return s * std::sqrt(fmin_(dot(ca, ca), dot(cb, cb)));
}
inline float eval_sdf_function(const vpt_sdf& sdf, v3 p) {   // the std::function bodies, yocto_sceneio.cpp:3684-3730
  COUNT(C_SDF_EVALS);
  switch (sdf.type) {
    case VPT_SDF_BBOX: return sd_bbox(p, v3{sdf.p[1], sdf.p[2], sdf.p[3]}, sdf.p[0]);
    case VPT_SDF_BOX: return sd_box(p - (to_v3(sdf.whd) * 0.5f), to_v3(sdf.whd) * 0.5f);
    case VPT_SDF_CAPPED_CONE: return sd_capped_cone(p, sdf.p[0], sdf.p[1], sdf.p[2]);
    case VPT_SDF_PLANE: return p.y;
    case VPT_SDF_SPHERE: return length(p) - sdf.p[0];
    case VPT_SDF_TORUS: return sd_torus(p, sdf.p[0], sdf.p[1]);
    default: return flt_max;
  }
}
float eval_volume(S& s, const vpt_volume& vol, v3 uvw) {   // yocto_sdfs.cpp:92-127
  auto W = vol.whd[0], H = vol.whd[1], D = vol.whd[2];
  if ((int64_t)W * H * D == 0) return 0;
  float sx = clampf((uvw.x + 1.0f) * 0.5f, 0.0f, 1.0f) * (W - 1);
  float ty = clampf((uvw.y + 1.0f) * 0.5f, 0.0f, 1.0f) * (H - 1);
  float rz = clampf((uvw.z + 1.0f) * 0.5f, 0.0f, 1.0f) * (D - 1);
  auto i = clampi((int)sx, 0, W - 1), j = clampi((int)ty, 0, H - 1), k = clampi((int)rz, 0, D - 1);
  auto ii = (i + 1 < W - 1) ? i + 1 : W - 1, jj = (j + 1 < H - 1) ? j + 1 : H - 1, kk = (k + 1 < D - 1) ? k + 1 : D - 1;
  float u = sx - i, v = ty - j, w = rz - k;
  auto at = [&](int x, int y, int z) { COUNT(C_VOXEL_FETCHES); return s.voxels[vol.offset + x + (int64_t)y * W + (int64_t)z * W * H]; };
  return at(i, j, k) * (1 - u) * (1 - v) * (1 - w) + at(ii, j, k) * u * (1 - v) * (1 - w) + at(i, jj, k) * (1 - u) * v * (1 - w) +
         at(i, j, kk) * (1 - u) * (1 - v) * w + at(i, jj, kk) * (1 - u) * v * w + at(ii, j, kk) * u * (1 - v) * w +
         at(ii, jj, k) * u * v * (1 - w) + at(ii, jj, kk) * u * v * w;
}
float eval_sdf_grid(S& s, const vpt_volume_instance& inst, v3 p, float t) {   // yocto_sdfs.cpp:30-49
  auto& vol = s.volumes[inst.volume];
  COUNT(C_SDF_EVALS);
  auto grid_res = v3{(float)vol.whd[0], (float)vol.whd[1], (float)vol.whd[2]};
  auto origin   = to_v3(inst.frame.o);
  auto bbox_max  = origin + (vol.res * grid_res) * inst.scalef;
  auto bbox_size = (bbox_max - origin);
  auto bbox_dist = sd_box(p - (bbox_size * 0.5f), (bbox_size * 0.5f));
  if (bbox_dist < flt_eps * t) {
    auto uvw = p * 2.f / (bbox_size)-1;
    return eval_volume(s, vol, uvw) * inst.scalef;
  }
  return bbox_dist;
}
struct sdf_result { float result = flt_max; int instance = -1, sdf = -1; };
sdf_result eval_sdf_scene(S& s, v3 p, float t) {   // yocto_sdfs.cpp:7-26
  auto res = sdf_result{};
  for (auto idx = 0; idx < s.num_vol_instances; idx++) {
    auto& inst = s.vol_instances[idx];
    auto  d    = eval_sdf_grid(s, inst, transform_point(to_fr(inst.frame), p), t);
    if (d < res.result) res = {d, idx, -1};
  }
  for (auto idx = 0; idx < s.num_sdfs; idx++) {
    auto& sdf = s.sdfs[idx];
    auto  d   = eval_sdf_function(sdf, transform_point(to_fr(sdf.frame), p));
    if (d < res.result) res = {d, -1, idx};
  }
  return res;
}
v3 eval_sdf_normal_function(const vpt_sdf& sdf, v3 p, float t) {   // yocto_sdfs.cpp:67-76
  const float h = flt_eps * t;
  auto f  = to_fr(sdf.frame);
  auto p1 = transform_point(f, p + v3{1, -1, -1} * h), p2 = transform_point(f, p + v3{-1, -1, 1} * h);
  auto p3 = transform_point(f, p + v3{-1, 1, -1} * h), p4 = transform_point(f, p + v3{1, 1, 1} * h);
  return normalize(v3{1, -1, -1} * eval_sdf_function(sdf, p1) + v3{-1, -1, 1} * eval_sdf_function(sdf, p2) +
                   v3{-1, 1, -1} * eval_sdf_function(sdf, p3) + v3{1, 1, 1} * eval_sdf_function(sdf, p4));
}
v3 eval_sdf_normal_grid(S& s, const vpt_volume_instance& inst, v3 p, float t) {   // yocto_sdfs.cpp:79-89
  const float h = flt_eps * t;
  auto f  = to_fr(inst.frame);
  auto p1 = transform_point(f, p + v3{1, -1, -1} * h), p2 = transform_point(f, p + v3{-1, -1, 1} * h);
  auto p3 = transform_point(f, p + v3{-1, 1, -1} * h), p4 = transform_point(f, p + v3{1, 1, 1} * h);
  return normalize(v3{1, -1, -1} * eval_sdf_grid(s, inst, p1, t) + v3{-1, -1, 1} * eval_sdf_grid(s, inst, p2, t) +
                   v3{-1, 1, -1} * eval_sdf_grid(s, inst, p3, t) + v3{1, 1, 1} * eval_sdf_grid(s, inst, p4, t));
}
struct st_result { bool hit = false; float dist = flt_max; int instance = -1, sdf = -1; };
st_result spheretrace_one(S& s, const ray3& ray, int sdf_handle, int maxiter) {   // yocto_pathtrace.cpp:267-286
  auto  t   = ray.tmin;
  auto& sdf = s.sdfs[sdf_handle];
  for (int i = 0; i < maxiter && t < ray.tmax; ++i) {
    auto p   = ray_point(ray, t);
    auto res = eval_sdf_function(sdf, transform_point(to_fr(sdf.frame), p));
    tl_counters[C_LIGHT_MARCH_STEPS]++;
    if (fabs_(res) < (flt_eps * t)) return {true, t, -1, sdf_handle};
    t += res;
  }
  return {};
}
st_result spheretrace(S& s, const ray3& ray, int maxiter) {   // yocto_pathtrace.cpp:289-307
  auto t = ray.tmin;
  auto i = 0;
  tl_counters[C_MARCHES]++;
  for (; i < maxiter && t < ray.tmax; ++i) {
    auto p   = ray_point(ray, t);
    auto res = eval_sdf_scene(s, p, t);
    if (t > 16) tl_counters[C_STEPS_FAR]++;
    if (fabs_(res.result) < (flt_eps * t)) {
      tl_counters[C_STEPS_HIT] += i + 1;
      return {true, t, res.instance, res.sdf};
    }
    t += res.result;
  }
  tl_counters[i >= maxiter ? C_STEPS_MAXITER : C_STEPS_ESCAPED] += i;
  return {};
}

// ------------------------------------------------------------------------------------------------
// lights (yocto_pathtrace.cpp:312-421)
// ------------------------------------------------------------------------------------------------
v3 sample_lights(S& s, v3 position, float rl, float rel, v2 ruv) {
  auto  light_id = sample_uniform(s.num_lights, rl);
  auto& light    = s.lights[light_id];
  auto  cdf      = s.light_cdf + light.cdf_offset;
  if (light.instance != VPT_INVALID) {
    auto& inst    = s.instances[light.instance];
    auto& sh      = s.shapes[inst.shape];
    auto  element = sample_discrete(cdf, light.cdf_len, rel);
    auto  uv      = (sh.num_triangles != 0) ? sample_triangle(ruv) : ruv;
    auto  lposition = eval_position(s, inst, element, uv);
    return normalize(lposition - position);
  } else if (light.sdf != VPT_INVALID) {
    auto& sdf     = s.sdfs[light.sdf];
    auto  wlightp = transform_point(inverse(to_fr(sdf.frame), false), v3{ruv.x, ruv.y, 1} * to_v3(sdf.whd));
    return normalize(wlightp - position);
  } else if (light.environment != VPT_INVALID) {
    auto& env = s.environments[light.environment];
    if (env.emission_tex != VPT_INVALID) {
      auto& tex = s.textures[env.emission_tex];
      auto  idx = sample_discrete(cdf, light.cdf_len, rel);
      auto  uv  = v2{((idx % tex.width) + 0.5f) / tex.width, ((idx / tex.width) + 0.5f) / tex.height};
      return transform_direction(to_fr(env.frame),
          v3{lm_cos(uv.x * 2 * pif) * lm_sin(uv.y * pif), lm_cos(uv.y * pif), lm_sin(uv.x * 2 * pif) * lm_sin(uv.y * pif)});
    }
    return sample_sphere(ruv);
  }
  return {0, 0, 0};
}
float sample_lights_pdf(S& s, v3 position, v3 direction, int spheretrace_maxiter) {
  auto pdf = 0.0f;
  for (auto l = 0; l < s.num_lights; l++) {
    auto& light = s.lights[l];
    auto  cdf   = s.light_cdf + light.cdf_offset;
    if (light.instance != VPT_INVALID) {
      auto& inst = s.instances[light.instance];
      auto lpdf = 0.0f;
      auto next_position = position;
      for (auto bounce = 0; bounce < 100; bounce++) {
        COUNT(C_LIGHT_PDF_HOPS);
        auto isec = intersect_instance_bvh(s, light.instance, make_ray(next_position, direction));
        if (!isec.hit) break;
        auto lposition = eval_position(s, inst, isec.element, isec.uv);
        auto lnormal   = eval_element_normal(s, inst, isec.element);
        auto area      = cdf[light.cdf_len - 1];
        lpdf += distance_squared(lposition, position) / (fabs_(dot(lnormal, direction)) * area);
        next_position = lposition + direction * 1e-3f;
      }
      pdf += lpdf;
    } else if (light.sdf != VPT_INVALID) {
      auto ray  = make_ray(position, direction);
      auto isec = spheretrace_one(s, ray, light.sdf, spheretrace_maxiter);
      if (isec.hit) {
        tl_flags |= F_SDF_LIGHT_PDF;
        auto lposition = ray_point(ray, isec.dist);
        auto lnormal   = eval_sdf_normal_function(s.sdfs[isec.sdf], position, isec.dist);   // (sic) at `position`
        auto area      = cdf[light.cdf_len - 1];
        pdf += distance_squared(lposition, position) / (fabs_(dot(lnormal, direction)) * area);
      }
    } else if (light.environment != VPT_INVALID) {
      auto& env = s.environments[light.environment];
      if (env.emission_tex != VPT_INVALID) {
        auto& tex = s.textures[env.emission_tex];
        auto  wl  = transform_direction(inverse(to_fr(env.frame), false), direction);
        auto  texcoord = v2{lm_atan2(wl.z, wl.x) / (2 * pif), lm_acos(clampf(wl.y, -1.0f, 1.0f)) / pif};
        if (texcoord.x < 0) texcoord.x += 1;
        auto i = clampi((int)(texcoord.x * tex.width), 0, tex.width - 1);
        auto j = clampi((int)(texcoord.y * tex.height), 0, tex.height - 1);
        auto prob  = sample_discrete_pdf(cdf, j * tex.width + i) / cdf[light.cdf_len - 1];
        auto angle = (2 * pif / tex.width) * (pif / tex.height) * lm_sin(pif * (j + 0.5f) / tex.height);
        pdf += prob / angle;
      } else {
        pdf += 1 / (4 * pif);
      }
    }
  }
  pdf *= (float)1 / (float)s.num_lights;
  return pdf;
}

// ------------------------------------------------------------------------------------------------
// shaders (yocto_pathtrace.cpp:425-930).  Draw order R0: see header.
// ------------------------------------------------------------------------------------------------
struct ctx { S& s; const vpt_params& p; };

// the MIS "next direction" block shared by pathtrace / volpathtrace / implicit (cpp:621-639 etc.)
// returns false when the path must end (`incoming == 0`, cpp:629)
inline bool next_direction_surface(const ctx& c, const mpoint& material, v3 normal, v3 outgoing, v3 position, rng_t& rng,
    v3& weight, v3& incoming, float bsdf_prob, bool mis) {
  incoming = {0, 0, 0};
  if (!is_delta(material)) {
    if (rand1f(rng) < bsdf_prob) {
      auto rn  = v2{};
      rn.x     = rand1f(rng);
      rn.y     = rand1f(rng);
      auto rnl = rand1f(rng);
      incoming = sample_bsdfcos(material, normal, outgoing, rnl, rn);
    } else {
      auto ruv = v2{};
      ruv.x    = rand1f(rng);
      ruv.y    = rand1f(rng);
      auto rel = rand1f(rng);
      auto rl  = rand1f(rng);
      incoming = sample_lights(c.s, position, rl, rel, ruv);
    }
    if (incoming == v3{0, 0, 0}) return false;
    if (mis) {
      weight *= eval_bsdfcos(material, normal, outgoing, incoming) /
                (0.5f * sample_bsdfcos_pdf(material, normal, outgoing, incoming) +
                    0.5f * sample_lights_pdf(c.s, position, incoming, c.p.spheretrace_maxiter));
    } else {
      weight *= eval_bsdfcos(material, normal, outgoing, incoming) / sample_bsdfcos_pdf(material, normal, outgoing, incoming);
    }
  } else {
    auto rnl = rand1f(rng);
    incoming = sample_delta(material, normal, outgoing, rnl);
    weight *= eval_delta(material, normal, outgoing, incoming) / sample_delta_pdf(material, normal, outgoing, incoming);
  }
  return true;
}
// weight check + russian roulette (cpp:676-683); returns false to end the path
inline bool roulette(v3& weight, int bounce, rng_t& rng) {
  if (weight == v3{0, 0, 0} || !finite3(weight)) return false;
  if (bounce > 3) {
    auto rr_prob = fmin_((float)0.99, max3(weight));
    if (rand1f(rng) >= rr_prob) return false;
    weight *= 1 / rr_prob;
  }
  return true;
}

v4 shade_volpathtrace(const ctx& c, const ray3& ray_, rng_t& rng) {   // cpp:565-687
  S& s = c.s;
  auto radiance = v3{0, 0, 0}, weight = v3{1, 1, 1};
  auto ray = ray_;
  auto hit = false;
  auto in_medium = false;   // 1-deep vstack (cpp:644-647)
  auto medium    = mpoint{};
  for (auto bounce = 0; bounce < c.p.bounces; bounce++) {
    COUNT(C_BOUNCES);
    auto isec = intersect_scene_bvh(s, ray);
    if (!isec.hit) {
      radiance += weight * eval_environment(s, ray.d);
      break;
    }
    auto in_volume = false;
    if (in_medium) {
      auto density  = medium.density;
      auto rd       = rand1f(rng);   // 4th argument evaluated first
      auto rl       = rand1f(rng);
      auto distance = sample_transmittance(density, isec.distance, rl, rd);
      weight *= eval_transmittance(density, distance) / sample_transmittance_pdf(density, distance, isec.distance);
      in_volume     = distance < isec.distance;
      isec.distance = distance;
    }
    if (!in_volume) {
      COUNT(C_SURFACE_HITS);
      auto& inst    = s.instances[isec.instance];
      auto outgoing = -ray.d;
      auto position = eval_position(s, inst, isec.element, isec.uv);
      auto normal   = eval_shading_normal(s, inst, isec.element, isec.uv, outgoing);
      auto material = eval_material(s, inst, isec.element, isec.uv);
      if (material.opacity < 1 && rand1f(rng) >= material.opacity) {
        ray = make_ray(position + ray.d * 1e-2f, ray.d);
        bounce -= 1;
        continue;
      }
      if (bounce == 0) hit = true;
      radiance += weight * eval_emission(material, normal, outgoing);
      auto incoming = v3{0, 0, 0};
      if (!next_direction_surface(c, material, normal, outgoing, position, rng, weight, incoming, 0.5f, true)) break;
      if (is_volumetric(s, inst) && dot(normal, outgoing) * dot(normal, incoming) < 0) {
        if (!in_medium) medium = eval_material(s, inst, isec.element, isec.uv), in_medium = true;
        else in_medium = false;
      }
      ray = make_ray(position, incoming);
    } else {
      COUNT(C_VOLUME_EVENTS);
      auto outgoing = -ray.d;
      auto position = ray_point(ray, isec.distance);
      auto& vol     = medium;
      radiance += weight * eval_emission(vol, position, outgoing);   // (sic) position as "normal", cpp:660
      auto incoming = v3{0, 0, 0};
      if (rand1f(rng) < 0.5) {
        auto rn = v2{};
        rn.x    = rand1f(rng);
        rn.y    = rand1f(rng);
        (void)rand1f(rng);   // rnl: drawn, unused (cpp:665)
        incoming = sample_scattering(vol, outgoing, rn);
      } else {
        auto ruv = v2{};
        ruv.x    = rand1f(rng);
        ruv.y    = rand1f(rng);
        auto rel = rand1f(rng);
        auto rl  = rand1f(rng);
        incoming = sample_lights(s, position, rl, rel, ruv);
      }
      weight *= eval_scattering(vol, outgoing, incoming) /
                (0.5f * sample_scattering_pdf(vol, outgoing, incoming) +
                    0.5f * sample_lights_pdf(s, position, incoming, c.p.spheretrace_maxiter));
      ray = make_ray(position, incoming);
    }
    if (!roulette(weight, bounce, rng)) break;
  }
  return {radiance.x, radiance.y, radiance.z, hit ? 1.0f : 0.0f};
}

v4 shade_pathtrace(const ctx& c, const ray3& ray_, rng_t& rng) {   // cpp:690-762
  S& s = c.s;
  auto radiance = v3{0, 0, 0}, weight = v3{1, 1, 1};
  auto ray = ray_;
  auto hit = false;
  for (auto bounce = 0; bounce < c.p.bounces; bounce++) {
    COUNT(C_BOUNCES);
    auto isec = intersect_scene_bvh(s, ray);
    if (!isec.hit) {
      radiance += weight * eval_environment(s, ray.d);
      break;
    }
    COUNT(C_SURFACE_HITS);
    auto& inst    = s.instances[isec.instance];
    auto outgoing = -ray.d;
    auto position = eval_position(s, inst, isec.element, isec.uv);
    auto normal   = eval_shading_normal(s, inst, isec.element, isec.uv, outgoing);
    auto material = eval_material(s, inst, isec.element, isec.uv);
    if (material.opacity < 1 && rand1f(rng) >= material.opacity) {
      ray = make_ray(position + ray.d * 1e-2f, ray.d);
      bounce -= 1;
      continue;
    }
    if (bounce == 0) hit = true;
    radiance += weight * eval_emission(material, normal, outgoing);
    auto incoming = v3{0, 0, 0};
    if (!next_direction_surface(c, material, normal, outgoing, position, rng, weight, incoming, 0.5f, true)) break;
    ray = make_ray(position, incoming);
    if (!roulette(weight, bounce, rng)) break;
  }
  return {radiance.x, radiance.y, radiance.z, hit ? 1.0f : 0.0f};
}

v4 shade_naive(const ctx& c, const ray3& ray_, rng_t& rng) {   // cpp:765-832
  S& s = c.s;
  auto radiance = v3{0, 0, 0}, weight = v3{1, 1, 1};
  auto ray = ray_;
  auto hit = false;
  for (auto bounce = 0; bounce < c.p.bounces; bounce++) {
    COUNT(C_BOUNCES);
    auto isec = intersect_scene_bvh(s, ray);
    if (!isec.hit) {
      radiance += weight * eval_environment(s, ray.d);
      break;
    }
    COUNT(C_SURFACE_HITS);
    auto& inst    = s.instances[isec.instance];
    auto outgoing = -ray.d;
    auto position = eval_position(s, inst, isec.element, isec.uv);
    auto normal   = eval_shading_normal(s, inst, isec.element, isec.uv, outgoing);
    auto material = eval_material(s, inst, isec.element, isec.uv);
    if (material.opacity < 1 && rand1f(rng) >= material.opacity) {
      ray = make_ray(position + ray.d * 1e-2f, ray.d);
      bounce -= 1;
      continue;
    }
    if (bounce == 0) hit = true;
    radiance += weight * eval_emission(material, normal, outgoing);
    auto incoming = v3{0, 0, 0};
    if (material.roughness != 0) {
      auto rn  = v2{};
      rn.x     = rand1f(rng);
      rn.y     = rand1f(rng);
      auto rnl = rand1f(rng);
      incoming = sample_bsdfcos(material, normal, outgoing, rnl, rn);
      if (incoming == v3{0, 0, 0}) break;
      weight *= eval_bsdfcos(material, normal, outgoing, incoming) / sample_bsdfcos_pdf(material, normal, outgoing, incoming);
    } else {
      auto rnl = rand1f(rng);
      incoming = sample_delta(material, normal, outgoing, rnl);
      if (incoming == v3{0, 0, 0}) break;
      weight *= eval_delta(material, normal, outgoing, incoming) / sample_delta_pdf(material, normal, outgoing, incoming);
    }
    if (!roulette(weight, bounce, rng)) break;
    ray = make_ray(position, incoming);
  }
  return {radiance.x, radiance.y, radiance.z, hit ? 1.0f : 0.0f};
}

v4 shade_eyelight(const ctx& c, const ray3& ray_, rng_t& rng) {   // cpp:835-890
  S& s = c.s;
  auto radiance = v3{0, 0, 0}, weight = v3{1, 1, 1};
  auto ray = ray_;
  auto hit = false;
  auto nb  = c.p.bounces > 4 ? c.p.bounces : 4;
  for (auto bounce = 0; bounce < nb; bounce++) {
    COUNT(C_BOUNCES);
    auto isec = intersect_scene_bvh(s, ray);
    if (!isec.hit) {
      radiance += weight * eval_environment(s, ray.d);
      break;
    }
    COUNT(C_SURFACE_HITS);
    auto& inst    = s.instances[isec.instance];
    auto outgoing = -ray.d;
    auto position = eval_position(s, inst, isec.element, isec.uv);
    auto normal   = eval_shading_normal(s, inst, isec.element, isec.uv, outgoing);
    auto material = eval_material(s, inst, isec.element, isec.uv);
    if (material.opacity < 1 && rand1f(rng) >= material.opacity) {
      ray = make_ray(position + ray.d * 1e-2f, ray.d);
      bounce -= 1;
      continue;
    }
    if (bounce == 0) hit = true;
    auto incoming = outgoing;
    radiance += weight * eval_emission(material, normal, outgoing);
    radiance += weight * pif * eval_bsdfcos(material, normal, outgoing, incoming);
    if (!is_delta(material)) break;
    incoming = sample_delta(material, normal, outgoing, rand1f(rng));
    if (incoming == v3{0, 0, 0}) break;
    weight *= eval_delta(material, normal, outgoing, incoming) / sample_delta_pdf(material, normal, outgoing, incoming);
    if (weight == v3{0, 0, 0} || !finite3(weight)) break;
    ray = make_ray(position, incoming);
  }
  return {radiance.x, radiance.y, radiance.z, hit ? 1.0f : 0.0f};
}

v4 shade_normal(const ctx& c, const ray3& ray, rng_t&) {   // cpp:893-904
  auto isec = intersect_scene_bvh(c.s, ray);
  if (!isec.hit) return {0, 0, 0, 0};
  auto n = eval_shading_normal(c.s, c.s.instances[isec.instance], isec.element, isec.uv, -ray.d);
  return {n.x, n.y, n.z, 1};
}
v4 shade_texcoord(const ctx& c, const ray3& ray, rng_t&) {   // cpp:907-917
  auto isec = intersect_scene_bvh(c.s, ray);
  if (!isec.hit) return {0, 0, 0, 0};
  auto t = eval_texcoord(c.s, c.s.instances[isec.instance], isec.element, isec.uv);
  return {t.x, t.y, 0, 1};
}
v4 shade_color(const ctx& c, const ray3& ray, rng_t&) {   // cpp:920-930
  auto isec = intersect_scene_bvh(c.s, ray);
  if (!isec.hit) return {0, 0, 0, 0};
  auto col = eval_material(c.s, c.s.instances[isec.instance], isec.element, isec.uv).color;
  return {col.x, col.y, col.z, 1};
}

inline v3 implicit_normal(S& s, const st_result& isec, v3 position) {   // cpp:451-460
  if (isec.instance != VPT_INVALID) return eval_sdf_normal_grid(s, s.vol_instances[isec.instance], position, isec.dist);
  return eval_sdf_normal_function(s.sdfs[isec.sdf], position, isec.dist);
}
v4 shade_implicit(const ctx& c, const ray3& ray_, rng_t& rng) {   // cpp:425-535
  S& s = c.s;
  auto radiance = v3{0, 0, 0}, weight = v3{1, 1, 1};
  auto ray = ray_;
  for (auto bounce = 0; bounce < c.p.bounces; bounce++) {
    COUNT(C_BOUNCES);
    auto isec = spheretrace(s, ray, c.p.spheretrace_maxiter);
    if (!isec.hit) {
      radiance += weight * eval_environment(s, ray.d);
      break;
    }
    COUNT(C_SURFACE_HITS);
    auto outgoing = -ray.d;
    auto position = ray_point(ray, isec.dist);
    auto normal   = implicit_normal(s, isec, position);
    auto material_handle = isec.instance != VPT_INVALID ? s.vol_instances[isec.instance].material : s.sdfs[isec.sdf].material;
    auto material = eval_material(s, material_handle);
    if (material.opacity < 1 && rand1f(rng) >= material.opacity) {
      ray = make_ray(position + ray.d * 1e-2f, ray.d);
      bounce -= 1;
      continue;
    }
    radiance += weight * eval_emission(material, normal, outgoing);
    auto incoming = v3{0, 0, 0};
    if (!next_direction_surface(c, material, normal, outgoing, position, rng, weight, incoming,
            c.p.noimplicit_mis ? 1.0f : 0.5f, !c.p.noimplicit_mis))
      break;
    ray = make_ray(position, incoming);
    if (!roulette(weight, bounce, rng)) break;
  }
  return {radiance.x, radiance.y, radiance.z, 1};
}
v4 shade_implicit_normal(const ctx& c, const ray3& ray, rng_t&) {   // cpp:538-562
  auto isec = spheretrace(c.s, ray, c.p.spheretrace_maxiter);
  if (!isec.hit) return {0, 0, 0, 0};
  auto position = ray_point(ray, isec.dist);
  auto normal   = implicit_normal(c.s, isec, position);
  // `normal * 0.5 + 0.5` with double literals: vec3f * (float)0.5 + (float)0.5
  normal = normal * 0.5f + 0.5f;
  return {normal.x, normal.y, normal.z, 1};
}

using shader_fn = v4 (*)(const ctx&, const ray3&, rng_t&);
shader_fn get_shader(int shader) {   // cpp:936-952
  switch (shader) {
    case VPT_SHADER_VOLPATHTRACE: return shade_volpathtrace;
    case VPT_SHADER_PATHTRACE: return shade_pathtrace;
    case VPT_SHADER_NAIVE: return shade_naive;
    case VPT_SHADER_EYELIGHT: return shade_eyelight;
    case VPT_SHADER_NORMAL: return shade_normal;
    case VPT_SHADER_TEXCOORD: return shade_texcoord;
    case VPT_SHADER_COLOR: return shade_color;
    case VPT_SHADER_IMPLICIT: return shade_implicit;
    case VPT_SHADER_IMPLICIT_NORMAL: return shade_implicit_normal;
    default: return nullptr;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// entry point: same contract as vpt_render() in include/vpt.h, plus a thread count and optional
// event counters (24 x u64, see enum above).  Threads pull pixel indices from an atomic counter
// like yocto_parallel.h:189-212.
// ------------------------------------------------------------------------------------------------
static int oracle_render(const vpt_scene_desc* desc, const vpt_params* params, int nsamples, int width, int height,
    float* image_rgba, int32_t* hits, uint64_t* rng, int* samples_io, int nthreads, uint64_t* counters, uint8_t* flags,
    const int32_t* pixels, int npixels_listed, uint64_t perturb_seed = 0, unsigned perturb_mask = 0) {
  if (!desc || !params || !image_rgba || !hits || !rng || !samples_io) return VPT_ERR_INVALID_ARG;
  auto shader = get_shader(params->shader);
  if (!shader) return VPT_ERR_UNKNOWN_SHADER;
  if (params->camera < 0 || params->camera >= desc->num_cameras) return VPT_ERR_INVALID_ARG;
  auto& camera = desc->cameras[params->camera];
  auto  c      = ctx{*desc, *params};
  if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
  if (counters) std::memset(counters, 0, sizeof(uint64_t) * C_COUNT);
  auto npixels = pixels ? npixels_listed : width * height;
  for (auto pass = 0; pass < nsamples; pass++) {
    if (*samples_io >= params->samples) break;   // cpp:1055
    *samples_io += 1;
    auto preview = params->samples == 1;         // cpp:1059
    auto next    = std::atomic<int>{0};
    auto merged  = std::vector<uint64_t>(C_COUNT * (size_t)nthreads, 0);
    auto work    = [&](int tid) {
      std::memset(tl_counters, 0, sizeof(tl_counters));
      while (true) {
        auto k = next.fetch_add(1);
        if (k >= npixels) break;
        auto idx = pixels ? pixels[k] : k;
        auto i = idx % width, j = idx / width;
        tl_flags = 0;
        // libm perturbation (off unless asked): seeded by (seed, pixel, pass), so the result does not depend on the threads
        tl_pert_mask  = perturb_mask;
        tl_pert_state = perturb_seed ? (((perturb_seed * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)idx * 0xBF58476D1CE4E5B9ull) ^ ((uint64_t)(*samples_io) << 48)) | 1ull) : 0;
        auto r = rng_t{rng[2 * idx], rng[2 * idx + 1]};
        float u, v;
        if (preview) {
          u = (i + 0.5f) / width, v = (j + 0.5f) / height;
        } else {
          u = (i + rand1f(r)) / width;
          v = (j + rand1f(r)) / height;
        }
        auto lens = v2{};
        lens.x    = rand1f(r);
        lens.y    = rand1f(r);
        auto ray      = eval_camera(camera, {u, v}, lens);
        auto radiance = shader(c, ray, r);
        if (!(std::isfinite(radiance.x) && std::isfinite(radiance.y) && std::isfinite(radiance.z) && std::isfinite(radiance.w)))
          radiance = {0, 0, 0, 0};
        image_rgba[4 * idx + 0] += radiance.x, image_rgba[4 * idx + 1] += radiance.y;
        image_rgba[4 * idx + 2] += radiance.z, image_rgba[4 * idx + 3] += radiance.w;
        hits[idx] += 1;
        rng[2 * idx] = r.state, rng[2 * idx + 1] = r.inc;
        if (flags) flags[idx] |= (uint8_t)tl_flags;
        tl_pert_state = 0;
        COUNT(C_SAMPLES);
      }
      std::memcpy(&merged[C_COUNT * (size_t)tid], tl_counters, sizeof(tl_counters));
    };
    if (nthreads == 1) {
      work(0);
    } else {
      auto threads = std::vector<std::thread>{};
      for (auto t = 0; t < nthreads; t++) threads.emplace_back(work, t);
      for (auto& t : threads) t.join();
    }
    if (counters)
      for (auto t = 0; t < nthreads; t++)
        for (auto k = 0; k < C_COUNT; k++) counters[k] += merged[C_COUNT * (size_t)t + k];
  }
  return VPT_OK;
}

extern "C" int vpt_oracle_render(const vpt_scene_desc* desc, const vpt_params* params, int nsamples, int width,
    int height, float* image_rgba, int32_t* hits, uint64_t* rng, int* samples_io, int nthreads, uint64_t* counters) {
  return oracle_render(desc, params, nsamples, width, height, image_rgba, hits, rng, samples_io, nthreads, counters, nullptr, nullptr, 0);
}
// the same, also OR-ing each pixel's condition flags (F_* above) into flags[width * height]
extern "C" int vpt_oracle_render_flags(const vpt_scene_desc* desc, const vpt_params* params, int nsamples, int width,
    int height, float* image_rgba, int32_t* hits, uint64_t* rng, int* samples_io, int nthreads, uint8_t* flags) {
  return oracle_render(desc, params, nsamples, width, height, image_rgba, hits, rng, samples_io, nthreads, nullptr, flags, nullptr, 0);
}
// The same with every libm result of the shading path nudged by -1 / 0 / +1 float ulp, pseudo-randomly per pixel
// (seed != 0; site_mask: 1 sin/cos, 2 atan/atan2/acos, 4 exp/log, 8 pow; 15 = all).  See the wrappers above.
// `pixels` (optional): only those row-major pixel indices are rendered, as in vpt_oracle_render_pixels.
extern "C" int vpt_oracle_render_perturbed(const vpt_scene_desc* desc, const vpt_params* params, int nsamples, int width,
    int height, float* image_rgba, int32_t* hits, uint64_t* rng, int* samples_io, int nthreads, uint64_t seed, unsigned site_mask,
    const int32_t* pixels, int npixels) {
  if (seed == 0 || (pixels && npixels < 0)) return VPT_ERR_INVALID_ARG;
  for (auto k = 0; pixels && k < npixels; k++)
    if (pixels[k] < 0 || pixels[k] >= width * height) return VPT_ERR_INVALID_ARG;
  return oracle_render(desc, params, nsamples, width, height, image_rgba, hits, rng, samples_io, nthreads, nullptr, nullptr, pixels, npixels, seed, site_mask);
}
// the same for a subset of the frame: only the `npixels` row-major pixel indices listed in `pixels` are rendered
// (each at most once); every other element of the state arrays is left untouched.  What one rank of a tile-sharded
// render does; *samples_io advances as for the whole frame.
extern "C" int vpt_oracle_render_pixels(const vpt_scene_desc* desc, const vpt_params* params, int nsamples, int width,
    int height, float* image_rgba, int32_t* hits, uint64_t* rng, int* samples_io, int nthreads, const int32_t* pixels, int npixels) {
  if (!pixels || npixels < 0) return VPT_ERR_INVALID_ARG;
  for (auto k = 0; k < npixels; k++)
    if (pixels[k] < 0 || pixels[k] >= width * height) return VPT_ERR_INVALID_ARG;
  return oracle_render(desc, params, nsamples, width, height, image_rgba, hits, rng, samples_io, nthreads, nullptr, nullptr, pixels, npixels);
}

// intersect_bvh(bvh, scene, ray) / intersect_bvh(bvh, scene, instance, ray) (yocto_bvh.cpp:1097-1113) for a batch
// of rays {o, d} with the default tmin = 1e-4, tmax = flt_max: the checker for vpt_intersect (include/vpt.h).
// ids: {instance, element} per ray (-1, -1 on a miss); uvt: {u, v, distance} (zeros on a miss).
extern "C" int vpt_oracle_intersect(const vpt_scene_desc* desc, int n, const float* rays, int instance, int32_t* ids, float* uvt) {
  if (!desc || !rays || !ids || !uvt || n < 0) return VPT_ERR_INVALID_ARG;
  if (instance >= desc->num_instances) return VPT_ERR_INVALID_ARG;
  for (auto i = 0; i < n; i++) {
    auto ray = make_ray({rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, {rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]});
    auto r   = instance < 0 ? intersect_scene_bvh(*desc, ray) : intersect_instance_bvh(*desc, instance, ray);
    ids[2 * i] = r.hit ? r.instance : -1, ids[2 * i + 1] = r.hit ? r.element : -1;
    uvt[3 * i] = r.hit ? r.uv.x : 0, uvt[3 * i + 1] = r.hit ? r.uv.y : 0, uvt[3 * i + 2] = r.hit ? r.distance : 0;
  }
  return VPT_OK;
}

extern "C" const char* vpt_oracle_counter_names() {
  return "samples,scene_nodes,shape_nodes,instance_tests,quad_tests,tri_tests,texel_f32,texel_u8,cdf_probes,"
         "surface_hits,volume_events,bounces,sdf_evals,voxel_fetches,light_pdf_hops,marches,steps_hit,steps_maxiter,"
         "steps_escaped,steps_far,light_march_steps";
}

// ------------------------------------------------------------------------------------------------
// Known-answer-test twin of vpt_kat() (include/vpt_kat.h): the same ops and record layouts, evaluated by
// the functions above.  tests/ compare it bit for bit with the tables the reference's own functions
// produced (oracle/ref_tables.cpp -> tests/golden/kat_*.npz).
// ------------------------------------------------------------------------------------------------
namespace {
const int kat_strides[VPT_KAT_OP_COUNT][2] = {{19, 22}, {15, 10}, {4, 4}, {5, 6}, {7, 5}, {7, 24}, {3, 3}, {7, 3}, {6, 1},
    {6, 1}, {4, 3}, {6, 3}, {7, 4}, {4, 1}, {4, 1}};
inline void put3(float* p, v3 v) { p[0] = v.x, p[1] = v.y, p[2] = v.z; }
inline bool is_zero(v3 v) { return v == v3{0, 0, 0}; }
}  // namespace

extern "C" int vpt_oracle_kat(const vpt_scene_desc* desc, int op, int iparam, int n, const float* in, float* out) {
  if (op < 0 || op >= VPT_KAT_OP_COUNT || n < 0 || (n > 0 && (!in || !out))) return VPT_ERR_INVALID_ARG;
  auto scene_free = op == VPT_KAT_LOBES || op == VPT_KAT_MEDIA;
  if (!desc && !scene_free) return VPT_ERR_INVALID_ARG;
  auto si = kat_strides[op][0], so = kat_strides[op][1];
  static const vpt_scene_desc empty = {};
  S& s = desc ? *desc : empty;
  for (auto r = 0; r < n; r++) {
    auto a = in + (size_t)r * si;
    auto o = out + (size_t)r * so;
    for (auto k = 0; k < so; k++) o[k] = 0;
    switch (op) {
      case VPT_KAT_LOBES: {
        auto m      = mpoint{};
        m.type      = (int)a[0];
        m.color     = to_v3(a + 1);
        m.roughness = a[4], m.metallic = a[5], m.ior = a[6];
        auto normal = to_v3(a + 7), outgoing = to_v3(a + 10);
        auto rnl = a[13];
        auto rn  = v2{a[14], a[15]};
        auto alt = to_v3(a + 16);
        auto s_in = sample_bsdfcos(m, normal, outgoing, rnl, rn);
        put3(o, s_in);
        if (!is_zero(s_in)) put3(o + 3, eval_bsdfcos(m, normal, outgoing, s_in)), o[6] = sample_bsdfcos_pdf(m, normal, outgoing, s_in);
        put3(o + 7, eval_bsdfcos(m, normal, outgoing, alt));
        o[10]     = sample_bsdfcos_pdf(m, normal, outgoing, alt);
        auto d_in = sample_delta(m, normal, outgoing, rnl);
        put3(o + 11, d_in);
        if (!is_zero(d_in)) put3(o + 14, eval_delta(m, normal, outgoing, d_in)), o[17] = sample_delta_pdf(m, normal, outgoing, d_in);
        put3(o + 18, eval_delta(m, normal, outgoing, alt));
        o[21] = sample_delta_pdf(m, normal, outgoing, alt);
      } break;
      case VPT_KAT_MEDIA: {
        auto density = to_v3(a);
        auto maxd = a[3], rl = a[4], rd = a[5], g = a[6];
        auto outgoing = to_v3(a + 7);
        auto rn       = v2{a[10], a[11]};
        auto incoming = to_v3(a + 12);
        auto distance = sample_transmittance(density, maxd, rl, rd);
        o[0] = distance, o[1] = sample_transmittance_pdf(density, distance, maxd);
        put3(o + 2, eval_transmittance(density, distance));
        o[5]       = eval_phasefunction(g, outgoing, incoming);
        auto s_dir = sample_phasefunction(g, outgoing, rn);
        put3(o + 6, s_dir);
        o[9] = eval_phasefunction(g, outgoing, s_dir);
      } break;
      case VPT_KAT_TEXTURE: {
        auto t = (int)a[0];
        if (t < 0 || t >= s.num_textures) return VPT_ERR_INVALID_ARG;
        auto c = eval_texture(s, s.textures[t], v2{a[1], a[2]}, a[3] != 0);
        o[0] = c.x, o[1] = c.y, o[2] = c.z, o[3] = c.w;
      } break;
      case VPT_KAT_CAMERA: {
        auto c = (int)a[0];
        if (c < 0 || c >= s.num_cameras) return VPT_ERR_INVALID_ARG;
        auto ray = eval_camera(s.cameras[c], v2{a[1], a[2]}, v2{a[3], a[4]});
        put3(o, ray.o), put3(o + 3, ray.d);
      } break;
      case VPT_KAT_INTERSECT: {
        auto ray  = make_ray(to_v3(a), to_v3(a + 3));
        auto inst = (int)a[6];
        if (inst >= s.num_instances) return VPT_ERR_INVALID_ARG;
        auto isec = inst < 0 ? intersect_scene_bvh(s, ray) : intersect_instance_bvh(s, inst, ray);
        o[0] = isec.hit ? (float)isec.instance : -1.0f, o[1] = isec.hit ? (float)isec.element : -1.0f;
        o[2] = isec.hit ? isec.uv.x : 0, o[3] = isec.hit ? isec.uv.y : 0, o[4] = isec.hit ? isec.distance : 0;
      } break;
      case VPT_KAT_SURFACE: {
        auto inst = (int)a[0], element = (int)a[1];
        if (inst < 0 || inst >= s.num_instances) return VPT_ERR_INVALID_ARG;
        auto& instance = s.instances[inst];
        auto& sh       = s.shapes[instance.shape];
        if (element < 0 || element >= (sh.num_triangles ? sh.num_triangles : sh.num_quads)) return VPT_ERR_INVALID_ARG;
        auto uv = v2{a[2], a[3]};
        auto outgoing = to_v3(a + 4);
        put3(o, eval_position(s, instance, element, uv));   // eval_shading_position == eval_position for triangles / quads (yocto_scene.cpp:460-473)
        put3(o + 3, eval_shading_normal(s, instance, element, uv, outgoing));
        auto m = eval_material(s, instance, element, uv);
        o[6]   = (float)m.type;
        put3(o + 7, m.emission), put3(o + 10, m.color);
        o[13] = m.opacity, o[14] = m.roughness, o[15] = m.metallic, o[16] = m.ior;
        put3(o + 17, m.density), put3(o + 20, m.scattering);
        o[23] = m.scanisotropy;
      } break;
      case VPT_KAT_ENVIRONMENT: put3(o, eval_environment(s, to_v3(a))); break;
      case VPT_KAT_SAMPLE_LIGHTS:
        if (s.num_lights <= 0) return VPT_ERR_INVALID_ARG;
        put3(o, sample_lights(s, to_v3(a), a[3], a[4], v2{a[5], a[6]}));
        break;
      case VPT_KAT_LIGHTS_PDF:
      case VPT_KAT_LIGHTS_PDF_K2:
        if (s.num_lights <= 0) return VPT_ERR_INVALID_ARG;
        o[0] = sample_lights_pdf(s, to_v3(a), to_v3(a + 3), iparam);
        break;
      case VPT_KAT_SDF_SCENE: {
        auto res = eval_sdf_scene(s, to_v3(a), a[3]);
        o[0] = res.result, o[1] = (float)res.instance, o[2] = (float)res.sdf;
      } break;
      case VPT_KAT_SDF_NORMAL: {
        auto kind = (int)a[0], idx = (int)a[1];
        if (kind == 0) {
          if (idx < 0 || idx >= s.num_vol_instances) return VPT_ERR_INVALID_ARG;
          put3(o, eval_sdf_normal_grid(s, s.vol_instances[idx], to_v3(a + 2), a[5]));
        } else {
          if (idx < 0 || idx >= s.num_sdfs) return VPT_ERR_INVALID_ARG;
          put3(o, eval_sdf_normal_function(s.sdfs[idx], to_v3(a + 2), a[5]));
        }
      } break;
      case VPT_KAT_SPHERETRACE: {
        auto ray = make_ray(to_v3(a), to_v3(a + 3));
        auto sdf = (int)a[6];
        if (sdf >= s.num_sdfs) return VPT_ERR_INVALID_ARG;
        auto res = sdf < 0 ? spheretrace(s, ray, iparam) : spheretrace_one(s, ray, sdf, iparam);
        o[0] = res.hit ? 1.0f : 0.0f, o[1] = res.dist, o[2] = (float)res.instance, o[3] = (float)res.sdf;
      } break;
      case VPT_KAT_VOLUME: {
        auto v = (int)a[0];
        if (v < 0 || v >= s.num_volumes) return VPT_ERR_INVALID_ARG;
        o[0] = eval_volume(s, s.volumes[v], to_v3(a + 1));
      } break;
      case VPT_KAT_SDF_FUNCTION: {
        auto f = (int)a[0];
        if (f < 0 || f >= s.num_sdfs) return VPT_ERR_INVALID_ARG;
        o[0] = eval_sdf_function(s.sdfs[f], to_v3(a + 1));
      } break;
    }
  }
  return VPT_OK;
}
