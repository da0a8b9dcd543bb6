// vpt_scene.hip.h — device functions: primitive tests, the binary-node shape traversal behind K2's
// mesh-light pdf walk (K1 walks quad nodes: traverse(), vpt_mesh_kernel.hip.h),
// scene/material/texture/environment evaluation, BSDF lobes, media, light sampling, SDF sphere
// tracing.  Each function names the reference lines whose arithmetic it reproduces
// (libs/yocto/*.h|cpp, libs/yocto_pathtrace/yocto_pathtrace.cpp).
#pragma once
#include "vpt_device.h"
#include "vpt_math.hip.h"

#ifndef VPT_BLOCK
#define VPT_BLOCK 64   // threads per workgroup = one wave64 = one 8x8 pixel tile: a wave that finishes frees its slot at once
#endif


// Diagnostic build (-DVPT_COUNTERS): how often a wave executes each code section and with how many
// active lanes.  Slot 2k counts wave executions, slot 2k+1 the lanes active in them.  Never in the product build.
#ifdef VPT_COUNTERS
__device__ unsigned long long g_vpt_cnt[64];
VPT_DEV void vpt_cnt(int k) {
  unsigned long long m = __builtin_amdgcn_ballot_w64(true);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
    atomicAdd(&g_vpt_cnt[2 * k], 1ull);
    atomicAdd(&g_vpt_cnt[2 * k + 1], (unsigned long long)__popcll(m));
  }
}
#define VPT_CNT(k) vpt_cnt(k)
// the same with a caller-made lane mask (group forms: the lanes of the groups that take part, not the lanes switched on)
VPT_DEV void vpt_cnt_mask(int k, unsigned long long part) {
  unsigned long long m = __builtin_amdgcn_ballot_w64(true);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
    atomicAdd(&g_vpt_cnt[2 * k], 1ull);
    atomicAdd(&g_vpt_cnt[2 * k + 1], (unsigned long long)__popcll(part));
  }
}
#define VPT_CNT_MASK(k, part) vpt_cnt_mask(k, part)
// histogram of a small count (slots 0 .. 23 of g_vpt_hist): rays in a group-form node step
__device__ unsigned long long g_vpt_hist[24];
VPT_DEV void vpt_hist(int n) {
  unsigned long long m = __builtin_amdgcn_ballot_w64(true);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&g_vpt_hist[n < 23 ? n : 23], 1ull);
}
#define VPT_HIST(n) vpt_hist(n)
// wave-level elapsed cycles per section: slot 32 + k of g_vpt_cnt (accumulated in LDS, flushed at kernel end)
__shared__ unsigned long long s_vpt_time[16];
VPT_DEV void vpt_time_add(int k, unsigned long long t0) {
  unsigned long long dt = __builtin_readcyclecounter() - t0;
  unsigned long long m  = __builtin_amdgcn_ballot_w64(true);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&s_vpt_time[k], dt);
}
#define VPT_T0(k) unsigned long long vpt_t0_##k = __builtin_readcyclecounter()
#define VPT_T1(k) vpt_time_add(k, vpt_t0_##k)
#else
#define VPT_CNT(k)
#define VPT_CNT_MASK(k, part)
#define VPT_HIST(n)
#define VPT_T0(k)
#define VPT_T1(k)
#endif
enum { TM_NODES = 0, TM_PRIMS, TM_ENTER, TM_QUERY, TM_TRIP, TM_LIGHTS_PDF, TM_SAMPLE_LIGHTS, TM_SURFACE, TM_VOLUME, TM_GENERATE, TM_KERNEL, TM_CDF, TM_SCATTER_EVAL, TM_MEDIUM, TM_SURF_GEOM, TM_SURF_DELTA };
enum { CNT_NODE = 0, CNT_PRIM, CNT_ENTER, CNT_OUTER, CNT_TRIP, CNT_POP, CNT_MISS, CNT_SURFACE, CNT_VOLUME, CNT_LIGHTS, CNT_GENERATE, CNT_LEAF, CNT_SESSION, CNT_GNODE, CNT_GLEAF, CNT_WNODE };

// ------------------------------------------------------------------------------------------------
// per-lane traversal stack in LDS: entry e of lane t lives at lds[e * VPT_BLOCK + t], so the 64
// lanes of a wave touch 64 consecutive dwords (conflict-free ds_read_b32 / ds_write_b32).
// ------------------------------------------------------------------------------------------------
struct lane_stack {
  int* base;   // &lds[threadIdx.x]
  int  cap;
  VPT_DEV void push(int& sp, int v) const {
    if (sp < cap) base[sp * VPT_BLOCK] = v;   // capacity is sized on the host from the BVH depth
    sp++;
  }
  VPT_DEV int pop(int& sp) const {
    sp--;
    return sp < cap ? base[sp * VPT_BLOCK] : 0;
  }
};

struct ray_t { f3 o, d; float tmin, tmax; };
VPT_DEV ray_t make_ray(f3 o, f3 d) { ray_t r = {o, d, VPT_RAY_EPS, VPT_FLT_MAX}; return r; }
VPT_DEV f3 ray_point(const ray_t& r, float t) { return r.o + r.d * t; }

struct hit_t { int instance, element; f2 uv; float distance; bool hit; int prim; };   // prim: slot in leaf_prims / leaf_attrs (quad-node traversal only)

// intersect_triangle, yocto_geometry.h:786-819 (Moller-Trumbore, no epsilon)
VPT_DEV bool intersect_triangle(f3 ro, f3 rd, float tmin, float tmax, f3 p0, f3 p1, f3 p2, f2& uv, float& dist) {
  f3    edge1 = p1 - p0, edge2 = p2 - p0;
  f3    pvec  = cross(rd, edge2);
  float det   = dot(edge1, pvec);
  if (det == 0) return false;
  float inv_det = rcp_exact(det);   // == 1.0f / det bit for bit (vpt_math.hip.h)
  f3    tvec    = ro - p0;
  float u       = dot(tvec, pvec) * inv_det;
  if (u < 0 || u > 1) return false;
  f3    qvec = cross(tvec, edge1);
  float v    = dot(rd, qvec) * inv_det;
  if (v < 0 || u + v > 1) return false;
  float t = dot(edge2, qvec) * inv_det;
  if (t < tmin || t > tmax) return false;
  uv = mk2(u, v), dist = t;
  return true;
}
// intersect_quad, yocto_geometry.h:822-839
VPT_DEV bool intersect_quad(f3 ro, f3 rd, float tmin, float tmax, f3 p0, f3 p1, f3 p2, f3 p3, f2& uv, float& dist) {
  if (eq3(p2, p3)) return intersect_triangle(ro, rd, tmin, tmax, p0, p1, p3, uv, dist);
  bool hit = false;
  if (intersect_triangle(ro, rd, tmin, tmax, p0, p1, p3, uv, dist)) hit = true, tmax = dist;
  if (intersect_triangle(ro, rd, tmin, tmax, p2, p3, p1, uv, dist)) hit = true, uv = 1 - uv, tmax = dist;
  return hit;
}
// intersect_bbox(ray, dinv, bbox), yocto_geometry.h:858-868
VPT_DEV bool intersect_bbox(f3 ro, f3 dinv, float tmin_, float tmax_, f3 bmin, f3 bmax) {
  f3    it_min = (bmin - ro) * dinv;
  f3    it_max = (bmax - ro) * dinv;
  f3    lo = vmin3(it_min, it_max), hi = vmax3(it_min, it_max);
  float t0 = fmax_(max3(lo), tmin_);
  float t1 = fmin_(min3(hi), tmax_);
  t1 *= 1.00000024f;
  return t0 <= t1;
}

// a primitive slot's records (vpt_device.h: leaf_prims / leaf_attrs, or tri_prims / tri_attrs in the instances compiled for COMPACT)
template <bool COMPACT>
VPT_DEV const float4* leaf_rec(const DScene& sc, int slot) {
  if constexpr (COMPACT) return sc.tri_prims + 3 * (long long)slot;
  else return sc.leaf_prims + 4 * (long long)slot;
}

// shape-level traversal, yocto_bvh.cpp:699-797.  `sp0` is the stack level owned by the caller.
VPT_DEV bool trace_shape(const DScene& sc, const DShape& sh, f3 ro, f3 rd, float tmin, float tmax,
    const lane_stack& stk, int sp0, int& element, f2& uv, float& distance) {
  if (sh.num_nodes == 0) return false;
  const float4* nodes = sc.shape_nodes + 2 * (long long)sh.node_offset;
  const float4* leafs = sc.leaf_prims + 4 * (long long)sh.leaf_offset;
  f3  dinv = mk3(1 / rd.x, 1 / rd.y, 1 / rd.z);
  int sgn  = (dinv.x < 0 ? 1 : 0) | (dinv.y < 0 ? 2 : 0) | (dinv.z < 0 ? 4 : 0);
  int sp   = sp0;
  stk.push(sp, 0);
  bool hit = false;
  while (sp != sp0) {
    int    n  = stk.pop(sp);
    float4 n0 = nodes[2 * n], n1 = nodes[2 * n + 1];
    if (!intersect_bbox(ro, dinv, tmin, tmax, mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y))) continue;
    int start = __float_as_int(n1.z), meta = __float_as_int(n1.w);
    if (meta >> 24) {   // internal: near child on top (yocto_bvh.cpp:744-750)
      int axis = (meta >> 16) & 0xff;
      if ((sgn >> axis) & 1) stk.push(sp, start), stk.push(sp, start + 1);
      else stk.push(sp, start + 1), stk.push(sp, start);
    } else {
      int num = meta & 0xffff;
      for (int k = 0; k < num; k++) {
        const float4* rec = leafs + 4 * (long long)(start + k);
        float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
        if (intersect_quad(ro, rd, tmin, tmax, xyz(r0), xyz(r1), xyz(r2), xyz(r3), uv, distance))
          hit = true, element = __float_as_int(r0.w), tmax = distance;
      }
    }
  }
  return hit;
}

// transform_ray(inverse(frame, true), ray) with the inverse precomputed at scene creation
VPT_DEV void to_instance_space(const DInstance& inst, f3 ro, f3 rd, f3& lo, f3& ld) {
  frame inv = unpack_frame(inst.inv[0], inst.inv[1], inst.inv[2]);
  lo = transform_point(inv, ro), ld = transform_vector(inv, rd);
}

// single-instance query used by the light pdf, yocto_bvh.cpp:874-881
VPT_DEV hit_t trace_instance(const DScene& sc, int instance, f3 o, f3 d, const lane_stack& stk) {
  hit_t r;
  r.instance = instance, r.element = -1, r.uv = mk2(0, 0), r.distance = 0;
  const DInstance& inst = sc.instances[instance];
  f3 lo, ld;
  to_instance_space(inst, o, d, lo, ld);
  r.hit = trace_shape(sc, sc.shapes[inst.shape], lo, ld, VPT_RAY_EPS, VPT_FLT_MAX, stk, 0, r.element, r.uv, r.distance);
  return r;
}

// ------------------------------------------------------------------------------------------------
// textures, yocto_scene.cpp:112-169 ; sRGB decode through the host-built LUT (yocto_color.h:224-227)
// ------------------------------------------------------------------------------------------------
// byte_to_float, yocto_color.h:212-214: b / 255.0f.  The quotient of the IEEE division for every byte, in three instructions instead of
// the eleven of a float32 division: one Newton step on b * fl(1 / 255) with fused multiply-adds (no overflow / underflow cases to guard for
// operands 0 .. 255); equal to b / 255.0f for all 256 bytes (tests/test_host_pipeline.py checks the identity in exact arithmetic)
VPT_DEV float byte_to_float(unsigned char b) {
  const float rcp = __uint_as_float(0x3b808081u);   // fl(1 / 255)
  const float x = (float)b, q = x * rcp;
  return __builtin_fmaf(__builtin_fmaf(-255.0f, q, x), rcp, q);
}
VPT_DEV f4 lookup_texture(const DScene& sc, const vpt_texture& t, int i, int j, bool as_linear) {
  long long idx = t.offset + (long long)j * t.width + i;
  if (t.is_float) {
    float4 v = sc.texels_f[idx];
    return mk4(v.x, v.y, v.z, v.w);   // linear textures are returned untouched
  }
  uchar4 b = sc.texels_b[idx];
  if (as_linear && !t.linear) return mk4(sc.srgb_lut[b.x], sc.srgb_lut[b.y], sc.srgb_lut[b.z], byte_to_float(b.w));
  return mk4(byte_to_float(b.x), byte_to_float(b.y), byte_to_float(b.z), byte_to_float(b.w));
}
// fmod(x, 1.0f) (yocto_scene.cpp:141-144) without ocml's generic remainder loop: for finite x the
// fractional part x - trunc(x) is exact in float32 and fmod's result carries the sign of x (also for a
// zero result), hence the copysign; inf / NaN give NaN in both forms.
VPT_DEV float fmod1(float x) { return copysignf(x - truncf(x), x); }

VPT_DEV f4 eval_texture(const DScene& sc, int texture, f2 uv, bool as_linear) {
  if (texture == VPT_INVALID) return mk4(1, 1, 1, 1);
  const vpt_texture& t = sc.textures[texture];
  if (t.width == 0 || t.height == 0) return mk4(0, 0, 0, 0);
  float s = fmod1(uv.x) * t.width;
  if (s < 0) s += t.width;
  float tt = fmod1(uv.y) * t.height;
  if (tt < 0) tt += t.height;
  int   i = clampi((int)s, 0, t.width - 1), j = clampi((int)tt, 0, t.height - 1);
  int   ii = i + 1 == t.width ? 0 : i + 1, jj = j + 1 == t.height ? 0 : j + 1;   // (i + 1) % width for 0 <= i < width, without the division
  float u = s - i, v = tt - j;
  return lookup_texture(sc, t, i, j, as_linear) * (1 - u) * (1 - v) + lookup_texture(sc, t, i, jj, as_linear) * (1 - u) * v +
         lookup_texture(sc, t, ii, j, as_linear) * u * (1 - v) + lookup_texture(sc, t, ii, jj, as_linear) * u * v;
}

// ------------------------------------------------------------------------------------------------
// shape evaluation, yocto_scene.cpp:279-526 / yocto_geometry.h:506-542
// ------------------------------------------------------------------------------------------------
VPT_DEV f3 tri_lerp(f3 p0, f3 p1, f3 p2, f2 uv) { return p0 * (1 - uv.x - uv.y) + p1 * uv.x + p2 * uv.y; }
VPT_DEV f2 tri_lerp(f2 p0, f2 p1, f2 p2, f2 uv) { return p0 * (1 - uv.x - uv.y) + p1 * uv.x + p2 * uv.y; }
VPT_DEV f4 tri_lerp(f4 p0, f4 p1, f4 p2, f2 uv) { return p0 * (1 - uv.x - uv.y) + p1 * uv.x + p2 * uv.y; }

// Corner selection of interpolate_triangle / interpolate_quad folded into one index shuffle:
// triangles: (x,y,z; uv) — quads: uv.x+uv.y<=1 ? (x,y,w; uv) : (z,w,y; 1-uv)
struct corners_t { int a, b, c; f2 uv; };
VPT_DEV corners_t pick_corners(const DShape& sh, int4 e, f2 uv) {
  corners_t r;
  if (sh.is_triangles) r.a = e.x, r.b = e.y, r.c = e.z, r.uv = uv;
  else if (uv.x + uv.y <= 1) r.a = e.x, r.b = e.y, r.c = e.w, r.uv = uv;
  else r.a = e.z, r.b = e.w, r.c = e.y, r.uv = 1 - uv;
  return r;
}
VPT_DEV f3 triangle_normal(f3 p0, f3 p1, f3 p2) { return normalize(cross(p1 - p0, p2 - p0)); }

struct surf_t {   // everything the shaders need at a surface hit
  f3 position, normal;
  f2 texcoord;
  f4 color_shp;
};

VPT_DEV f3 eval_position(const DScene& sc, const DInstance& inst, int element, f2 uv) {
  const DShape& sh  = sc.shapes[inst.shape];
  int4          e   = sc.elems[sh.elem_offset + element];
  corners_t     c   = pick_corners(sh, e, uv);
  const float4* pos = sc.positions + sh.vertex_offset;
  f3 p = tri_lerp(xyz(pos[c.a]), xyz(pos[c.b]), xyz(pos[c.c]), c.uv);
  return transform_point(unpack_frame(inst.fwd[0], inst.fwd[1], inst.fwd[2]), p);
}
// eval_element_normal, yocto_scene.cpp:305-327 (transform_normal with a rigid frame = direction)
VPT_DEV f3 eval_element_normal(const DScene& sc, const DInstance& inst, int element) {
  const DShape& sh  = sc.shapes[inst.shape];
  int4          e   = sc.elems[sh.elem_offset + element];
  const float4* pos = sc.positions + sh.vertex_offset;
  f3 n;
  if (sh.is_triangles) {
    n = triangle_normal(xyz(pos[e.x]), xyz(pos[e.y]), xyz(pos[e.z]));
  } else {
    f3 p0 = xyz(pos[e.x]), p1 = xyz(pos[e.y]), p2 = xyz(pos[e.z]), p3 = xyz(pos[e.w]);
    n = normalize(triangle_normal(p0, p1, p3) + triangle_normal(p2, p3, p1));
  }
  return transform_direction(unpack_frame(inst.fwd[0], inst.fwd[1], inst.fwd[2]), n);
}
VPT_DEV f3 eval_normal(const DScene& sc, const DInstance& inst, int element, f2 uv) {
  const DShape& sh = sc.shapes[inst.shape];
  if (sh.normal_offset < 0) return eval_element_normal(sc, inst, element);
  int4          e   = sc.elems[sh.elem_offset + element];
  corners_t     c   = pick_corners(sh, e, uv);
  const float4* nrm = sc.normals + sh.normal_offset;
  f3 n = normalize(tri_lerp(xyz(nrm[c.a]), xyz(nrm[c.b]), xyz(nrm[c.c]), c.uv));
  return transform_direction(unpack_frame(inst.fwd[0], inst.fwd[1], inst.fwd[2]), n);
}
VPT_DEV f2 eval_texcoord(const DScene& sc, const DInstance& inst, int element, f2 uv) {
  const DShape& sh = sc.shapes[inst.shape];
  if (sh.texcoord_offset < 0) return uv;
  int4          e  = sc.elems[sh.elem_offset + element];
  corners_t     c  = pick_corners(sh, e, uv);
  const float2* tc = sc.texcoords + sh.texcoord_offset;
  float2 a = tc[c.a], b = tc[c.b], d = tc[c.c];
  return tri_lerp(mk2(a.x, a.y), mk2(b.x, b.y), mk2(d.x, d.y), c.uv);
}
VPT_DEV f4 eval_color(const DScene& sc, const DInstance& inst, int element, f2 uv) {
  const DShape& sh = sc.shapes[inst.shape];
  if (sh.color_offset < 0) return mk4(1, 1, 1, 1);
  int4          e   = sc.elems[sh.elem_offset + element];
  corners_t     c   = pick_corners(sh, e, uv);
  const float4* col = sc.colors + sh.color_offset;
  float4 a = col[c.a], b = col[c.b], d = col[c.c];
  return tri_lerp(mk4(a.x, a.y, a.z, a.w), mk4(b.x, b.y, b.z, b.w), mk4(d.x, d.y, d.z, d.w), c.uv);
}
// normal mapping, yocto_scene.cpp:414-457 + yocto_geometry.h:606-640 (cold: only `normal_tex` materials)
VPT_DEV f3 eval_normalmap(const DScene& sc, const DInstance& inst, int element, f2 uv, f3 normal, int normal_tex) {
  const DShape& sh = sc.shapes[inst.shape];
  f2 texcoord  = eval_texcoord(sc, inst, element, uv);
  f3 normalmap = -1 + 2 * xyz(eval_texture(sc, normal_tex, texcoord, false));
  f3 tu = mk3(0, 0, 0), tv = mk3(0, 0, 0);
  if (sh.texcoord_offset >= 0) {
    int4          e   = sc.elems[sh.elem_offset + element];
    const float4* pos = sc.positions + sh.vertex_offset;
    const float2* tc  = sc.texcoords + sh.texcoord_offset;
    int ia = e.x, ib = e.y, ic = sh.is_triangles ? e.z : e.w;   // quads: always the (p0,p1,p3) half
    f3 p0 = xyz(pos[ia]), p = xyz(pos[ib]) - p0, q = xyz(pos[ic]) - p0;
    float2 t0 = tc[ia], t1 = tc[ib], t2 = tc[ic];
    float sx = t1.x - t0.x, sy = t2.x - t0.x, tx = t1.y - t0.y, ty = t2.y - t0.y;
    float div = sx * ty - sy * tx;
    if (div != 0) {
      tu = mk3(ty * p.x - tx * q.x, ty * p.y - tx * q.y, ty * p.z - tx * q.z) / div;
      tv = mk3(sx * q.x - sy * p.x, sx * q.y - sy * p.y, sx * q.z - sy * p.z) / div;
    } else {
      tu = mk3(1, 0, 0), tv = mk3(0, 1, 0);
    }
    frame f = unpack_frame(inst.fwd[0], inst.fwd[1], inst.fwd[2]);
    tu = transform_direction(f, tu), tv = transform_direction(f, tv);
  }
  f3 fx = orthonormalize(tu, normal);
  f3 fy = normalize(cross(normal, fx));
  bool flip_v = dot(fy, tv) < 0;
  normalmap.y *= flip_v ? 1 : -1;
  return normalize(fx * normalmap.x + fy * normalmap.y + normal * normalmap.z);
}
// eval_shading_normal, yocto_scene.cpp:476-503 (refractive keeps the geometric orientation)
VPT_DEV f3 eval_shading_normal(const DScene& sc, const DInstance& inst, int element, f2 uv, f3 outgoing) {
  const vpt_material& m = sc.materials[inst.material];
  f3 normal = eval_normal(sc, inst, element, uv);
  int ntex = m.normal_tex;
  if (ntex != VPT_INVALID) normal = eval_normalmap(sc, inst, element, uv, normal, ntex);
  if (m.type == VPT_MAT_REFRACTIVE) return normal;
  return dot(normal, outgoing) >= 0 ? normal : -normal;
}

// ------------------------------------------------------------------------------------------------
// material point, yocto_scene.h:292-304, yocto_scene.cpp:529-619
// ------------------------------------------------------------------------------------------------
struct mpoint {
  int   type;
  f3    emission, color;
  float opacity, roughness, metallic, ior;
  f3    density, scattering;
  float scanisotropy;
};
#define VPT_MIN_ROUGHNESS (0.03f * 0.03f)

VPT_DEV void finish_material(mpoint& p, float trdepth) {
  if (p.type == VPT_MAT_REFRACTIVE || p.type == VPT_MAT_VOLUMETRIC || p.type == VPT_MAT_SUBSURFACE) {
    f3 c      = vclamp(p.color, 0.0001f, 1.0f);
    p.density = -mk3(logf(c.x), logf(c.y), logf(c.z)) / trdepth;
  } else {
    p.density = mk3(0, 0, 0);
  }
  if (p.type == VPT_MAT_MATTE || p.type == VPT_MAT_GLTFPBR || p.type == VPT_MAT_GLOSSY) p.roughness = clampf(p.roughness, VPT_MIN_ROUGHNESS, 1.0f);
  else if (p.type == VPT_MAT_VOLUMETRIC) p.roughness = 0;
  else if (p.roughness < VPT_MIN_ROUGHNESS) p.roughness = 0;
}
// eval_material with the shading point's texcoord and shape colour already evaluated
VPT_DEV mpoint eval_material_at(const DScene& sc, const vpt_material& m, f2 texcoord, f4 color_shp) {
  f4 emission_tex = eval_texture(sc, m.emission_tex, texcoord, true);
  f4 color_tex    = eval_texture(sc, m.color_tex, texcoord, true);
  f4 rough_tex    = eval_texture(sc, m.roughness_tex, texcoord, false);
  f4 scatter_tex  = eval_texture(sc, m.scattering_tex, texcoord, true);
  mpoint p;
  p.type         = m.type;
  p.emission     = ld3(m.emission) * xyz(emission_tex);
  p.color        = ld3(m.color) * xyz(color_tex) * xyz(color_shp);
  p.opacity      = m.opacity * color_tex.w * color_shp.w;
  p.metallic     = m.metallic * rough_tex.z;
  p.roughness    = m.roughness * rough_tex.y;
  p.roughness    = p.roughness * p.roughness;
  p.ior          = m.ior;
  p.scattering   = ld3(m.scattering) * xyz(scatter_tex);
  p.scanisotropy = m.scanisotropy;
  finish_material(p, m.trdepth);
  return p;
}
VPT_DEV mpoint eval_material(const DScene& sc, const DInstance& inst, int element, f2 uv) {
  return eval_material_at(sc, sc.materials[inst.material], eval_texcoord(sc, inst, element, uv), eval_color(sc, inst, element, uv));
}
// eval_position + eval_normal + eval_texcoord (yocto_scene.cpp:279-379) of a hit from its primitive slot: the same
// corner choice (pick_corners), the same interpolation, the same values - fetched from leaf_prims / leaf_attrs.
// Only for shapes with vertex normals and without vertex colours (the callers check shape_flags).
template <bool COMPACT = false>
VPT_DEV void eval_surface_slot(const DScene& sc, const DInstance& inst, int prim, f2 uv, f3& position, f3& normal, f2& texcoord) {
  int a, b, c;
  f2  w = uv;
  if (inst.shape_flags & VPT_SHP_TRIANGLES) a = 0, b = 1, c = 2;
  else if (uv.x + uv.y <= 1) a = 0, b = 1, c = 3;
  else a = 2, b = 3, c = 1, w = 1 - uv;
  const float4* P = leaf_rec<COMPACT>(sc, prim);
  const float4* A = COMPACT ? sc.tri_attrs + 4 * (long long)prim : sc.leaf_attrs + 6 * (long long)prim;
  float4 pa = P[a], pb = P[b], pc = P[c], na = A[a], nb = A[b], nc = A[c];
  frame  f = unpack_frame(inst.fwd[0], inst.fwd[1], inst.fwd[2]);
  position = transform_point(f, tri_lerp(xyz(pa), xyz(pb), xyz(pc), w));
  normal   = transform_direction(f, normalize(tri_lerp(xyz(na), xyz(nb), xyz(nc), w)));
  texcoord = uv;
  if constexpr (COMPACT) {
    if (inst.shape_flags & VPT_SHP_TEXCOORDS) {   // a triangle's texcoords ride in the record's spare words
      float4 t3 = A[3];
      texcoord = tri_lerp(mk2(na.w, nb.w), mk2(nc.w, t3.x), mk2(t3.y, t3.z), w);
    }
  } else if (inst.shape_flags & VPT_SHP_TEXCOORDS) {
    const float2* T = (const float2*)(A + 4);
    float2 ta = T[a], tb = T[b], tc = T[c];
    texcoord = tri_lerp(mk2(ta.x, ta.y), mk2(tb.x, tb.y), mk2(tc.x, tc.y), w);
  }
}
VPT_DEV mpoint eval_material_plain(const DScene& sc, int mat) {   // yocto_scene.cpp:581-619
  const vpt_material& m = sc.materials[mat];
  mpoint p;
  p.type = m.type, p.emission = ld3(m.emission), p.color = ld3(m.color), p.opacity = m.opacity;
  p.metallic     = m.metallic;
  p.roughness    = m.roughness;
  p.roughness    = p.roughness * p.roughness;
  p.ior = m.ior, p.scattering = ld3(m.scattering), p.scanisotropy = m.scanisotropy;
  finish_material(p, m.trdepth);
  return p;
}
VPT_DEV bool is_delta(const mpoint& m) {
  return ((m.type == VPT_MAT_REFLECTIVE || m.type == VPT_MAT_REFRACTIVE || m.type == VPT_MAT_TRANSPARENT) && m.roughness == 0) ||
         m.type == VPT_MAT_VOLUMETRIC;
}
VPT_DEV bool is_volumetric_type(int t) { return t == VPT_MAT_REFRACTIVE || t == VPT_MAT_VOLUMETRIC || t == VPT_MAT_SUBSURFACE; }

// eval_environment, yocto_scene.cpp:634-651 (sum over all environments)
VPT_DEV f3 eval_environment(const DScene& sc, f3 direction) {
  f3 emission = mk3(0, 0, 0);
  for (int e = 0; e < sc.num_environments; e++) {
    const vpt_environment& env = sc.environments[e];
    f3 wl = transform_direction(load_frame(sc.env_inv + 3 * e), direction);
    f2 tc = mk2(atan2f(wl.z, wl.x) / (2 * VPT_PI), acosf(clampf(wl.y, -1.0f, 1.0f)) / VPT_PI);
    if (tc.x < 0) tc.x += 1;
    emission = emission + ld3(env.emission) * xyz(eval_texture(sc, env.emission_tex, tc, false));
  }
  return emission;
}

// eval_camera, yocto_scene.cpp:67-102
VPT_DEV ray_t eval_camera(const vpt_camera& cam, f2 image_uv, f2 lens_uv) {
  f2 film = cam.aspect >= 1 ? mk2(cam.film, cam.film / cam.aspect) : mk2(cam.film * cam.aspect, cam.film);
  frame fr = load_frame(cam.frame);
  if (!cam.orthographic) {
    f3 q  = mk3(film.x * (0.5f - image_uv.x), film.y * (image_uv.y - 0.5f), cam.lens);
    f3 dc = -normalize(q);
    f3 e  = mk3(lens_uv.x * cam.aperture / 2, lens_uv.y * cam.aperture / 2, 0);
    f3 p  = dc * cam.focus / fabs_(dc.z);
    f3 d  = normalize(p - e);
    return make_ray(transform_point(fr, e), transform_direction(fr, d));
  } else {
    float scale = 1 / cam.lens;
    f3 q = mk3(film.x * (0.5f - image_uv.x) * scale, film.y * (image_uv.y - 0.5f) * scale, cam.lens);
    f3 e = mk3(-q.x, -q.y, 0) + mk3(lens_uv.x * cam.aperture / 2, lens_uv.y * cam.aperture / 2, 0);
    f3 p = mk3(-q.x, -q.y, -cam.focus);
    f3 d = normalize(p - e);
    return make_ray(transform_point(fr, e), transform_direction(fr, d));
  }
}

// ------------------------------------------------------------------------------------------------
// sampling warps, yocto_sampling.h:251-395
// ------------------------------------------------------------------------------------------------
VPT_DEV f3 sample_hemisphere_cos(f3 normal, f2 ruv) {
  float z = sqrtf(ruv.y), r = sqrtf(1 - z * z), phi = 2 * VPT_PI * ruv.x;
  return transform_direction(basis_fromz(normal), mk3(r * cosf(phi), r * sinf(phi), z));
}
VPT_DEV float sample_hemisphere_cos_pdf(f3 normal, f3 direction) {
  float cosw = dot(normal, direction);
  return (cosw <= 0) ? 0 : cosw / VPT_PI;
}
VPT_DEV f3 sample_sphere(f2 ruv) {
  float z = 2 * ruv.y - 1, r = sqrtf(clampf(1 - z * z, 0.0f, 1.0f)), phi = 2 * VPT_PI * ruv.x;
  return mk3(r * cosf(phi), r * sinf(phi), z);
}
VPT_DEV int sample_uniform(int size, float r) { return clampi((int)(r * size), 0, size - 1); }
// sample_discrete: std::upper_bound over the CDF (yocto_sampling.h:385-390)
VPT_DEV int sample_discrete(const float* cdf, int n, float r) {
  float back = cdf[n - 1];
  r = clampf(r * back, 0.0f, back - 0.00001f);
  int lo = 0, len = n;
  while (len > 0) {
    int half = len >> 1;
    if (!(r < cdf[lo + half])) lo += half + 1, len -= half + 1;
    else len = half;
  }
  return clampi(lo, 0, n - 1);
}
// entries of one 16-wide group that are <= r (the predicate of the binary search above), group given by its first element
VPT_DEV int count_not_above(const float* g, float r) {
  const float4* q = (const float4*)g;
  float4 a = q[0], b = q[1], c = q[2], d = q[3];
  return (int)!(r < a.x) + (int)!(r < a.y) + (int)!(r < a.z) + (int)!(r < a.w) + (int)!(r < b.x) + (int)!(r < b.y) +
         (int)!(r < b.z) + (int)!(r < b.w) + (int)!(r < c.x) + (int)!(r < c.y) + (int)!(r < c.z) + (int)!(r < c.w) +
         (int)!(r < d.x) + (int)!(r < d.y) + (int)!(r < d.z) + (int)!(r < d.w);
}
// std::upper_bound(cdf, cdf + n, r) clamped to n - 1, through the light's guide table and 16-ary levels
// (DCdfIndex, vpt_device.h); r is already scaled and clamped as sample_discrete does
VPT_DEV int search_light_cdf(const DScene& sc, int light_id, float r) {
  const vpt_light& light = sc.lights[light_id];
  const int        n     = light.cdf_len;
  const DCdfIndex& ix    = sc.light_index[light_id];
  if (ix.guide_buckets > 0) {   // bracket from the guide table; short brackets are resolved with one 16-wide fetch
    int  b  = (int)(r * ix.guide_scale);
    int2 lh = sc.light_guide[ix.guide_offset + (b < ix.guide_buckets ? b : ix.guide_buckets - 1)];
    if (__builtin_amdgcn_ballot_w64(lh.y - lh.x > 16) == 0) {
      const float* g = sc.light_index_pool + ix.offset[0] + lh.x;   // level 0 = the CDF, padded with +inf
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < 16; i++) cnt += (int)!(r < g[i]);
      int found = lh.x + cnt;
      return found < n ? found : n - 1;
    }
  }
  int idx = count_not_above(sc.light_index_pool + ix.offset[ix.levels - 1], r);
  if (idx >= ix.top_count) return n - 1;   // no element above r: upper_bound == n, clamped
  // the group found at one level holds an entry > r (its maximum is the entry just passed), so every level
  // below finds its first entry > r inside that group; the +inf padding is never counted
  for (int k = ix.levels - 2; k >= 0; k--) idx = 16 * idx + count_not_above(sc.light_index_pool + ix.offset[k] + 16 * (long long)idx, r);
  return idx;
}
VPT_DEV int sample_light_cdf(const DScene& sc, int light_id, float r) {
  const vpt_light& light = sc.lights[light_id];
  const float*     cdf   = sc.light_cdf + light.cdf_offset;
  const int        n     = light.cdf_len;
  if (sc.light_index[light_id].levels == 0) return sample_discrete(cdf, n, r);
  float back = cdf[n - 1];
  return search_light_cdf(sc, light_id, clampf(r * back, 0.0f, back - 0.00001f));
}

// ------------------------------------------------------------------------------------------------
// BSDF lobes, yocto_shading.h:296-1039
// ------------------------------------------------------------------------------------------------
VPT_DEV bool same_hemisphere(f3 n, f3 o, f3 i) { return dot(n, o) * dot(n, i) >= 0; }
VPT_DEV f3 fresnel_schlick(f3 specular, f3 normal, f3 outgoing) {
  if (is_zero3(specular)) return mk3(0, 0, 0);
  float cosine = dot(normal, outgoing);
  return specular + (1 - specular) * powf(clampf(1 - fabs_(cosine), 0.0f, 1.0f), 5.0f);
}
VPT_DEV float fresnel_dielectric(float eta, f3 normal, f3 outgoing) {
  float cosw = fabs_(dot(normal, outgoing));
  float sin2 = 1 - cosw * cosw, eta2 = eta * eta;
  float cos2t = 1 - sin2 / eta2;
  if (cos2t < 0) return 1;
  float t0 = sqrtf(cos2t), t1 = eta * t0, t2 = eta * cosw;
  float rs = (cosw - t1) / (cosw + t1), rp = (t0 - t2) / (t0 + t2);
  return (rs * rs + rp * rp) / 2;
}
VPT_DEV f3 fresnel_conductor(f3 eta, f3 etak, f3 normal, f3 outgoing) {
  float cosw = dot(normal, outgoing);
  if (cosw <= 0) return mk3(0, 0, 0);
  cosw = clampf(cosw, -1.0f, 1.0f);
  float cos2 = cosw * cosw, sin2 = clampf(1 - cos2, 0.0f, 1.0f);
  f3 eta2 = eta * eta, etak2 = etak * etak;
  f3 t0 = eta2 - etak2 - sin2;
  f3 a2plusb2 = vsqrt(t0 * t0 + 4 * eta2 * etak2);
  f3 t1 = a2plusb2 + cos2;
  f3 a  = vsqrt((a2plusb2 + t0) / 2);
  f3 t2 = 2 * a * cosw;
  f3 rs = (t1 - t2) / (t1 + t2);
  f3 t3 = cos2 * a2plusb2 + sin2 * sin2;
  f3 t4 = t2 * sin2;
  f3 rp = rs * (t3 - t4) / (t3 + t4);
  return (rp + rs) / 2;
}
VPT_DEV f3 eta_to_reflectivity(f3 eta) { return ((eta - 1) * (eta - 1)) / ((eta + 1) * (eta + 1)); }
VPT_DEV f3 reflectivity_to_eta(f3 r_) {
  f3 r = vclamp(r_, 0.0f, 0.99f);
  return (1 + vsqrt(r)) / (1 - vsqrt(r));
}
VPT_DEV float microfacet_distribution(float roughness, f3 normal, f3 halfway) {
  float cosine = dot(normal, halfway);
  if (cosine <= 0) return 0;
  float r2 = roughness * roughness, c2 = cosine * cosine;
  return r2 / (VPT_PI * (c2 * r2 + 1 - c2) * (c2 * r2 + 1 - c2));
}
VPT_DEV float microfacet_shadowing1(float roughness, f3 normal, f3 halfway, f3 direction) {
  float cosine = dot(normal, direction), cosineh = dot(halfway, direction);
  if (cosine * cosineh <= 0) return 0;
  float r2 = roughness * roughness, c2 = cosine * cosine;
  return 2 * fabs_(cosine) / (fabs_(cosine) + sqrtf(c2 - r2 * c2 + r2));
}
VPT_DEV float microfacet_shadowing(float roughness, f3 n, f3 h, f3 o, f3 i) {
  return microfacet_shadowing1(roughness, n, h, o) * microfacet_shadowing1(roughness, n, h, i);
}
VPT_DEV f3 sample_microfacet(float roughness, f3 normal, f2 rn) {
  float phi   = 2 * VPT_PI * rn.x;
  float theta = atanf(roughness * sqrtf(rn.y / (1 - rn.y)));
  f3 local = mk3(cosf(phi) * sinf(theta), sinf(phi) * sinf(theta), cosf(theta));
  return transform_direction(basis_fromz(normal), local);
}
VPT_DEV float sample_microfacet_pdf(float roughness, f3 normal, f3 halfway) {
  float cosine = dot(normal, halfway);
  if (cosine < 0) return 0;
  return microfacet_distribution(roughness, normal, halfway) * cosine;
}
VPT_DEV f3 upn(f3 n, f3 o) { return dot(n, o) <= 0 ? -n : n; }

VPT_DEV f3 eval_matte(f3 color, f3 n, f3 o, f3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return mk3(0, 0, 0);
  return color / VPT_PI * fabs_(dot(n, i));
}
VPT_DEV float sample_matte_pdf(f3 n, f3 o, f3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return 0;
  return sample_hemisphere_cos_pdf(upn(n, o), i);
}
VPT_DEV f3 eval_glossy(f3 color, float ior, float roughness, f3 n, f3 o, f3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return mk3(0, 0, 0);
  f3 up = upn(n, o);
  float F1 = fresnel_dielectric(ior, up, o);
  f3 h = normalize(i + o);
  float F = fresnel_dielectric(ior, h, i);
  float D = microfacet_distribution(roughness, up, h);
  float G = microfacet_shadowing(roughness, up, h, o, i);
  return color * (1 - F1) / VPT_PI * fabs_(dot(up, i)) + mk3(1, 1, 1) * F * D * G / (4 * dot(up, o) * dot(up, i)) * fabs_(dot(up, i));
}
VPT_DEV f3 sample_glossy(float ior, float roughness, f3 n, f3 o, float rnl, f2 rn) {
  f3 up = upn(n, o);
  if (rnl < fresnel_dielectric(ior, up, o)) {
    f3 h = sample_microfacet(roughness, up, rn);
    f3 i = reflect(o, h);
    if (!same_hemisphere(up, o, i)) return mk3(0, 0, 0);
    return i;
  }
  return sample_hemisphere_cos(up, rn);
}
VPT_DEV float sample_glossy_pdf(float ior, float roughness, f3 n, f3 o, f3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return 0;
  f3 up = upn(n, o);
  f3 h  = normalize(o + i);
  float F = fresnel_dielectric(ior, up, o);
  return F * sample_microfacet_pdf(roughness, up, h) / (4 * fabs_(dot(o, h))) + (1 - F) * sample_hemisphere_cos_pdf(up, i);
}
VPT_DEV f3 eval_reflective(f3 color, float roughness, f3 n, f3 o, f3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return mk3(0, 0, 0);
  f3 up = upn(n, o);
  f3 h  = normalize(i + o);
  f3 F  = fresnel_conductor(reflectivity_to_eta(color), mk3(0, 0, 0), h, i);
  float D = microfacet_distribution(roughness, up, h);
  float G = microfacet_shadowing(roughness, up, h, o, i);
  return F * D * G / (4 * dot(up, o) * dot(up, i)) * fabs_(dot(up, i));
}
VPT_DEV f3 sample_reflective(float roughness, f3 n, f3 o, f2 rn) {
  f3 up = upn(n, o);
  f3 h  = sample_microfacet(roughness, up, rn);
  f3 i  = reflect(o, h);
  if (!same_hemisphere(up, o, i)) return mk3(0, 0, 0);
  return i;
}
VPT_DEV float sample_reflective_pdf(float roughness, f3 n, f3 o, f3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return 0;
  f3 up = upn(n, o);
  f3 h  = normalize(o + i);
  return sample_microfacet_pdf(roughness, up, h) / (4 * fabs_(dot(o, h)));
}
VPT_DEV f3 gltf_reflectivity(f3 color, float ior, float metallic) { return lerp3(eta_to_reflectivity(mk3(ior, ior, ior)), color, metallic); }
VPT_DEV f3 eval_gltfpbr(f3 color, float ior, float roughness, float metallic, f3 n, f3 o, f3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return mk3(0, 0, 0);
  f3 refl = gltf_reflectivity(color, ior, metallic);
  f3 up = upn(n, o);
  f3 F1 = fresnel_schlick(refl, up, o);
  f3 h  = normalize(i + o);
  f3 F  = fresnel_schlick(refl, h, i);
  float D = microfacet_distribution(roughness, up, h);
  float G = microfacet_shadowing(roughness, up, h, o, i);
  return color * (1 - metallic) * (1 - F1) / VPT_PI * fabs_(dot(up, i)) + F * D * G / (4 * dot(up, o) * dot(up, i)) * fabs_(dot(up, i));
}
VPT_DEV f3 sample_gltfpbr(f3 color, float ior, float roughness, float metallic, f3 n, f3 o, float rnl, f2 rn) {
  f3 up = upn(n, o);
  f3 refl = gltf_reflectivity(color, ior, metallic);
  if (rnl < mean3(fresnel_schlick(refl, up, o))) {
    f3 h = sample_microfacet(roughness, up, rn);
    f3 i = reflect(o, h);
    if (!same_hemisphere(up, o, i)) return mk3(0, 0, 0);
    return i;
  }
  return sample_hemisphere_cos(up, rn);
}
VPT_DEV float sample_gltfpbr_pdf(f3 color, float ior, float roughness, float metallic, f3 n, f3 o, f3 i) {
  if (dot(n, i) * dot(n, o) <= 0) return 0;
  f3 up = upn(n, o);
  f3 h  = normalize(o + i);
  float F = mean3(fresnel_schlick(gltf_reflectivity(color, ior, metallic), up, o));
  return F * sample_microfacet_pdf(roughness, up, h) / (4 * fabs_(dot(o, h))) + (1 - F) * sample_hemisphere_cos_pdf(up, i);
}
VPT_DEV f3 eval_transparent(f3 color, float ior, float roughness, f3 n, f3 o, f3 i) {
  f3 up = upn(n, o);
  if (dot(n, i) * dot(n, o) >= 0) {
    f3 h = normalize(i + o);
    float F = fresnel_dielectric(ior, h, o), D = microfacet_distribution(roughness, up, h);
    float G = microfacet_shadowing(roughness, up, h, o, i);
    return mk3(1, 1, 1) * F * D * G / (4 * dot(up, o) * dot(up, i)) * fabs_(dot(up, i));
  } else {
    f3 refl = reflect(-i, up);
    f3 h = normalize(refl + o);
    float F = fresnel_dielectric(ior, h, o), D = microfacet_distribution(roughness, up, h);
    float G = microfacet_shadowing(roughness, up, h, o, refl);
    return color * (1 - F) * D * G / (4 * dot(up, o) * dot(up, refl)) * (fabs_(dot(up, refl)));
  }
}
VPT_DEV f3 sample_transparent(float ior, float roughness, f3 n, f3 o, float rnl, f2 rn) {
  f3 up = upn(n, o);
  f3 h  = sample_microfacet(roughness, up, rn);
  if (rnl < fresnel_dielectric(ior, h, o)) {
    f3 i = reflect(o, h);
    if (!same_hemisphere(up, o, i)) return mk3(0, 0, 0);
    return i;
  } else {
    f3 refl = reflect(o, h);
    f3 i = -reflect(refl, up);
    if (same_hemisphere(up, o, i)) return mk3(0, 0, 0);
    return i;
  }
}
VPT_DEV float sample_transparent_pdf(float ior, float roughness, f3 n, f3 o, f3 i) {
  f3 up = upn(n, o);
  if (dot(n, i) * dot(n, o) >= 0) {
    f3 h = normalize(i + o);
    return fresnel_dielectric(ior, h, o) * sample_microfacet_pdf(roughness, up, h) / (4 * fabs_(dot(o, h)));
  } else {
    f3 refl = reflect(-i, up);
    f3 h = normalize(refl + o);
    float d = (1 - fresnel_dielectric(ior, h, o)) * sample_microfacet_pdf(roughness, up, h);
    return d / (4 * fabs_(dot(o, h)));
  }
}
VPT_DEV f3 eval_refractive(float ior, float roughness, f3 n, f3 o, f3 i) {
  bool entering = dot(n, o) >= 0;
  f3 up = entering ? n : -n;
  float rel_ior = entering ? ior : (1 / ior);
  if (dot(n, i) * dot(n, o) >= 0) {
    f3 h = normalize(i + o);
    float F = fresnel_dielectric(rel_ior, h, o), D = microfacet_distribution(roughness, up, h);
    float G = microfacet_shadowing(roughness, up, h, o, i);
    return mk3(1, 1, 1) * F * D * G / fabs_(4 * dot(n, o) * dot(n, i)) * fabs_(dot(n, i));
  } else {
    f3 h = -normalize(rel_ior * i + o) * (entering ? 1.0f : -1.0f);
    float F = fresnel_dielectric(rel_ior, h, o), D = microfacet_distribution(roughness, up, h);
    float G = microfacet_shadowing(roughness, up, h, o, i);
    return mk3(1, 1, 1) * fabs_((dot(o, h) * dot(i, h)) / (dot(o, n) * dot(i, n))) * (1 - F) * D * G /
           powf(rel_ior * dot(h, i) + dot(h, o), 2.0f) * fabs_(dot(n, i));
  }
}
VPT_DEV f3 sample_refractive(float ior, float roughness, f3 n, f3 o, float rnl, f2 rn) {
  bool entering = dot(n, o) >= 0;
  f3 up = entering ? n : -n;
  f3 h  = sample_microfacet(roughness, up, rn);
  if (rnl < fresnel_dielectric(entering ? ior : (1 / ior), h, o)) {
    f3 i = reflect(o, h);
    if (!same_hemisphere(up, o, i)) return mk3(0, 0, 0);
    return i;
  } else {
    f3 i = refract(o, h, entering ? (1 / ior) : ior);
    if (same_hemisphere(up, o, i)) return mk3(0, 0, 0);
    return i;
  }
}
VPT_DEV float sample_refractive_pdf(float ior, float roughness, f3 n, f3 o, f3 i) {
  bool entering = dot(n, o) >= 0;
  f3 up = entering ? n : -n;
  float rel_ior = entering ? ior : (1 / ior);
  if (dot(n, i) * dot(n, o) >= 0) {
    f3 h = normalize(i + o);
    return fresnel_dielectric(rel_ior, h, o) * sample_microfacet_pdf(roughness, up, h) / (4 * fabs_(dot(o, h)));
  } else {
    f3 h = -normalize(rel_ior * i + o) * (entering ? 1.0f : -1.0f);
    return (1 - fresnel_dielectric(rel_ior, h, o)) * sample_microfacet_pdf(roughness, up, h) * fabs_(dot(h, i)) /
           powf(rel_ior * dot(h, i) + dot(h, o), 2.0f);
  }
}
// delta lobes
VPT_DEV f3 eval_refractive_delta(float ior, f3 n, f3 o, f3 i) {
  if (fabs_(ior - 1) < 1e-3f) return dot(n, i) * dot(n, o) <= 0 ? mk3(1, 1, 1) : mk3(0, 0, 0);
  bool entering = dot(n, o) >= 0;
  f3 up = entering ? n : -n;
  float rel_ior = entering ? ior : (1 / ior);
  if (dot(n, i) * dot(n, o) >= 0) return mk3(1, 1, 1) * fresnel_dielectric(rel_ior, up, o);
  return mk3(1, 1, 1) * (1 / (rel_ior * rel_ior)) * (1 - fresnel_dielectric(rel_ior, up, o));
}
VPT_DEV f3 sample_refractive_delta(float ior, f3 n, f3 o, float rnl) {
  if (fabs_(ior - 1) < 1e-3f) return -o;
  bool entering = dot(n, o) >= 0;
  f3 up = entering ? n : -n;
  float rel_ior = entering ? ior : (1 / ior);
  if (rnl < fresnel_dielectric(rel_ior, up, o)) return reflect(o, up);
  return refract(o, up, 1 / rel_ior);
}
VPT_DEV float sample_refractive_delta_pdf(float ior, f3 n, f3 o, f3 i) {
  if (fabs_(ior - 1) < 1e-3f) return dot(n, i) * dot(n, o) < 0 ? 1.0f : 0.0f;
  bool entering = dot(n, o) >= 0;
  f3 up = entering ? n : -n;
  float rel_ior = entering ? ior : (1 / ior);
  if (dot(n, i) * dot(n, o) >= 0) return fresnel_dielectric(rel_ior, up, o);
  return (1 - fresnel_dielectric(rel_ior, up, o));
}

// material dispatch, yocto_pathtrace.cpp:86-236
VPT_DEV f3 eval_emission(f3 emission, f3 normal, f3 outgoing) { return dot(normal, outgoing) >= 0 ? emission : mk3(0, 0, 0); }
VPT_DEV f3 eval_bsdfcos(const mpoint& m, f3 n, f3 o, f3 i) {
  if (m.roughness == 0) return mk3(0, 0, 0);
  switch (m.type) {
    case VPT_MAT_MATTE: return eval_matte(m.color, n, o, i);
    case VPT_MAT_GLOSSY: return eval_glossy(m.color, m.ior, m.roughness, n, o, i);
    case VPT_MAT_REFLECTIVE: return eval_reflective(m.color, m.roughness, n, o, i);
    case VPT_MAT_TRANSPARENT: return eval_transparent(m.color, m.ior, m.roughness, n, o, i);
    case VPT_MAT_REFRACTIVE:
    case VPT_MAT_SUBSURFACE: return eval_refractive(m.ior, m.roughness, n, o, i);
    case VPT_MAT_GLTFPBR: return eval_gltfpbr(m.color, m.ior, m.roughness, m.metallic, n, o, i);
    default: return mk3(0, 0, 0);
  }
}
VPT_DEV f3 sample_bsdfcos(const mpoint& m, f3 n, f3 o, float rnl, f2 rn) {
  if (m.roughness == 0) return mk3(0, 0, 0);
  switch (m.type) {
    case VPT_MAT_MATTE: return sample_hemisphere_cos(upn(n, o), rn);
    case VPT_MAT_GLOSSY: return sample_glossy(m.ior, m.roughness, n, o, rnl, rn);
    case VPT_MAT_REFLECTIVE: return sample_reflective(m.roughness, n, o, rn);
    case VPT_MAT_TRANSPARENT: return sample_transparent(m.ior, m.roughness, n, o, rnl, rn);
    case VPT_MAT_REFRACTIVE:
    case VPT_MAT_SUBSURFACE: return sample_refractive(m.ior, m.roughness, n, o, rnl, rn);
    case VPT_MAT_GLTFPBR: return sample_gltfpbr(m.color, m.ior, m.roughness, m.metallic, n, o, rnl, rn);
    default: return mk3(0, 0, 0);
  }
}
VPT_DEV float sample_bsdfcos_pdf(const mpoint& m, f3 n, f3 o, f3 i) {
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
VPT_DEV f3 eval_delta(const mpoint& m, f3 n, f3 o, f3 i) {
  if (m.roughness != 0) return mk3(0, 0, 0);
  switch (m.type) {
    case VPT_MAT_REFLECTIVE:
      if (dot(n, i) * dot(n, o) <= 0) return mk3(0, 0, 0);
      return fresnel_conductor(reflectivity_to_eta(m.color), mk3(0, 0, 0), upn(n, o), o);
    case VPT_MAT_TRANSPARENT: {
      f3 up = upn(n, o);
      if (dot(n, i) * dot(n, o) >= 0) return mk3(1, 1, 1) * fresnel_dielectric(m.ior, up, o);
      return m.color * (1 - fresnel_dielectric(m.ior, up, o));
    }
    case VPT_MAT_REFRACTIVE: return eval_refractive_delta(m.ior, n, o, i);
    case VPT_MAT_VOLUMETRIC: return dot(n, i) * dot(n, o) >= 0 ? mk3(0, 0, 0) : mk3(1, 1, 1);
    default: return mk3(0, 0, 0);
  }
}
VPT_DEV f3 sample_delta(const mpoint& m, f3 n, f3 o, float rnl) {
  if (m.roughness != 0) return mk3(0, 0, 0);
  switch (m.type) {
    case VPT_MAT_REFLECTIVE: return reflect(o, upn(n, o));
    case VPT_MAT_TRANSPARENT: {
      f3 up = upn(n, o);
      if (rnl < fresnel_dielectric(m.ior, up, o)) return reflect(o, up);
      return -o;
    }
    case VPT_MAT_REFRACTIVE: return sample_refractive_delta(m.ior, n, o, rnl);
    case VPT_MAT_VOLUMETRIC: return -o;
    default: return mk3(0, 0, 0);
  }
}
VPT_DEV float sample_delta_pdf(const mpoint& m, f3 n, f3 o, f3 i) {
  if (m.roughness != 0) return 0;
  switch (m.type) {
    case VPT_MAT_REFLECTIVE: return dot(n, i) * dot(n, o) <= 0 ? 0.0f : 1.0f;
    case VPT_MAT_TRANSPARENT: {
      f3 up = upn(n, o);
      if (dot(n, i) * dot(n, o) >= 0) return fresnel_dielectric(m.ior, up, o);
      return 1 - fresnel_dielectric(m.ior, up, o);
    }
    case VPT_MAT_REFRACTIVE: return sample_refractive_delta_pdf(m.ior, n, o, i);
    case VPT_MAT_VOLUMETRIC: return dot(n, i) * dot(n, o) >= 0 ? 0.0f : 1.0f;
    default: return 0;
  }
}

// media, yocto_shading.h:1047-1102
VPT_DEV f3 vexp3(f3 a) { return mk3(expf(a.x), expf(a.y), expf(a.z)); }
VPT_DEV float sample_transmittance(f3 density, float max_distance, float rl, float rd) {
  int   channel  = clampi((int)(rl * 3), 0, 2);
  float dc       = comp(density, channel);
  float distance = (dc == 0) ? VPT_FLT_MAX : -logf(1 - rd) / dc;
  return fmin_(distance, max_distance);
}
VPT_DEV float sample_transmittance_pdf(f3 density, float distance, float max_distance) {
  if (distance < max_distance) return sum3(density * vexp3(-density * distance)) / 3;
  return sum3(vexp3(-density * max_distance)) / 3;
}
VPT_DEV float eval_phasefunction(float g, f3 outgoing, f3 incoming) {
  float cosine = -dot(outgoing, incoming);
  float denom  = 1 + g * g - 2 * g * cosine;
  return (1 - g * g) / (4 * VPT_PI * denom * sqrtf(denom));
}
VPT_DEV f3 sample_phasefunction(float g, f3 outgoing, f2 rn) {
  float cos_theta = 0.0f;
  if (fabs_(g) < 1e-3f) {
    cos_theta = 1 - 2 * rn.y;
  } else {
    float square = (1 - g * g) / (1 + g - 2 * g * rn.y);
    cos_theta    = (1 + g * g - square * square) / (2 * g);
  }
  float sin_theta = sqrtf(fmax_(0.0f, 1 - cos_theta * cos_theta));
  float phi = 2 * VPT_PI * rn.x;
  return mul(basis_fromz(-outgoing), mk3(sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta));
}

// ------------------------------------------------------------------------------------------------
// SDF module, yocto_sdfs.h:43-80, yocto_sdfs.cpp:7-127, yocto_pathtrace.cpp:259-307
// ------------------------------------------------------------------------------------------------
// The reference's abs / min / max are the ternaries of yocto_math.h:1354-1356 (fabs_, fmin_, fmax_ in vpt_math.hip.h): a compare
// and a select each, plus a hazard wait state where the select takes a negated operand.  In the box distances below they are
// replaced by the hardware's |x| operand modifier and v_min / v_max (v_max3), which return the same float bits for every finite
// input - the only inputs a march produces (p = o + t d with t < flt_max):
//  * |x| and the ternary abs differ for x = -0 only (|x| = +0, the ternary -0); v_max3 and the nested ternaries pick the same VALUE
//    and can only disagree on the sign of a zero maximum; v_min(x, +0) returns -0 where the ternary returns +0 for x = -0;
//  * v_max(x, +0) equals the ternary (x > 0 ? x : 0) for every x (both give +0 for either zero), so the vector under the square
//    root is identical, and so is its length L >= +0;
//  * what is left is the sign of a zero in the "inside" term i: the result is i + L, and (-0) + L == (+0) + L for every L >= +0
//    (L > 0: L; L = +0: +0 by IEEE addition).  With a zero half-extent b the argument is the same one step earlier (|x| - b = ±0).
// Checked bit for bit against the reference's sd_* tables and eval_sdf_scene tables (tests/test_kat.py, tolerance 0).
// One box distance is 6 + 2 selects and compares less per component: the scene of config 4 evaluates six boxes per march step.
#ifndef VPT_SD_HW
#define VPT_SD_HW 1
#endif
VPT_DEV float sd_inside(float a, float b, float c) { return VPT_SD_HW ? hw_min(hw_max3(a, b, c), 0.0f) : fmin_(fmax_(a, fmax_(b, c)), 0.0f); }
VPT_DEV f3 sd_outside(f3 d) { return VPT_SD_HW ? mk3(hw_max(d.x, 0.0f), hw_max(d.y, 0.0f), hw_max(d.z, 0.0f)) : vmaxs(d, 0.0f); }
VPT_DEV f3 sd_abs(f3 a) { return VPT_SD_HW ? mk3(__builtin_fabsf(a.x), __builtin_fabsf(a.y), __builtin_fabsf(a.z)) : vabs(a); }
VPT_DEV float sd_box(f3 p, f3 b) {
  f3 d = sd_abs(p) - b;
  return sd_inside(d.x, d.y, d.z) + length(sd_outside(d));
}
VPT_DEV float sd_bbox(f3 p, f3 b, float e) {
  p    = vabs(p) - b;
  f3 q = vabs(p + e) - e;
  return fmin_(fmin_(length(vmaxs(mk3(p.x, q.y, q.z), 0.0f)) + fmin_(fmax_(p.x, fmax_(q.y, q.z)), 0.0f),
                   length(vmaxs(mk3(q.x, p.y, q.z), 0.0f)) + fmin_(fmax_(q.x, fmax_(p.y, q.z)), 0.0f)),
      length(vmaxs(mk3(q.x, q.y, p.z), 0.0f)) + fmin_(fmax_(q.x, fmax_(q.y, p.z)), 0.0f));
}
VPT_DEV float sd_capped_cone(f3 p, float h, float r1, float r2) {
  f2 q  = mk2(length(mk2(p.x, p.z)), p.y);
  f2 k1 = mk2(r2, h), k2 = mk2(r2 - r1, 2.0f * h);
  f2 ca = mk2(q.x - fmin_(q.x, (q.y < 0.0f) ? r1 : r2), fabs_(q.y) - h);
  f2 cb = q - k1 + k2 * clampf(dot(k1 - q, k2) / dot(k2, k2), 0.0f, 1.0f);
  float s = (cb.x < 0.0f && ca.y < 0.0f) ? -1.0f : 1.0f;
  return s * sqrtf(fmin_(dot(ca, ca), dot(cb, cb)));
}
// SDF records (vpt_device.h, DScene::sdf_fn_rec / sdf_grid_rec): per analytic SDF / per voxel-grid instance, everything
// an evaluation needs behind ONE wave-uniform index, with the constants the reference recomputes at every evaluation
// (half extents, grid box size) folded on the host by the same float operations (vpt_capi.hip).
//   fn   [0..2] forward frame (packed)  [3] p[0..3]  [4] {whd * 0.5, type | translation << 8}  [5] {world centre, radius} (radius < 0: unbounded)
//   grid [0..2] forward frame (packed)  [3] {box size, scalef}  [4] {box size * 0.5, translation}  [5] {W, H, D, res}  [6] {voxel offset lo, hi}
// transform_point(frame, p): for a frame whose rotation part is exactly the identity the products 1 * p.x + 0 * p.y + 0 * p.z
// equal p.x (a zero term never changes a finite sum; only the sign of an exact zero can differ, which every sd_* below
// discards: they take abs / squares / compare with 0), so such frames take three additions.
// Where the records are read from: the scene's tables in HBM (DScene::sdf_fn_rec / sdf_grid_rec), or a copy a kernel made
// in its LDS (K2: the records are wave-uniform and read six times per march step; from HBM each read is a vector load
// with its full latency on the critical path of the step, from LDS a broadcast ds_read).
struct sdf_recs { const float4 *fn, *grid; };
VPT_DEV sdf_recs scene_sdf_recs(const DScene& sc) { sdf_recs r = {sc.sdf_fn_rec, sc.sdf_grid_rec}; return r; }
VPT_DEV f3 sdf_to_local(const float4* rec, bool translation, f3 pw) {
  float4 r2 = rec[2];
  if (translation) return mk3(pw.x + r2.y, pw.y + r2.z, pw.z + r2.w);
  return transform_point(unpack_frame(rec[0], rec[1], r2), pw);
}
// sdf_data::f of the reference (the lambdas of yocto_sceneio.cpp:3684-3730) at a point of the SDF's local frame
VPT_DEV float sdf_fn_local(const float4* rec, f3 p) {
  float4 q = rec[3], h = rec[4];
  switch (__float_as_int(h.w) & 255) {
    case VPT_SDF_BBOX: return sd_bbox(p, mk3(q.y, q.z, q.w), q.x);
    case VPT_SDF_BOX: return sd_box(p - xyz(h), xyz(h));   // sd_box(p - whd * 0.5, whd * 0.5)
    case VPT_SDF_CAPPED_CONE: return sd_capped_cone(p, q.x, q.y, q.z);
    case VPT_SDF_PLANE: return p.y;
    case VPT_SDF_SPHERE: return length(p) - q.x;
    case VPT_SDF_TORUS: return length(mk2(length(mk2(p.x, p.z)) - q.x, p.y)) - q.y;
    default: return VPT_FLT_MAX;
  }
}
VPT_DEV float sdf_fn_world(const sdf_recs& recs, int idx, f3 pw) {   // sdf.f(transform_point(sdf.frame, p)), yocto_sdfs.cpp:21
  const float4* rec = recs.fn + 6 * idx;
  return sdf_fn_local(rec, sdf_to_local(rec, (__float_as_int(rec[4].w) >> 8) & 1, pw));
}
// eval_volume: 8-tap trilinear in the reference's term order, yocto_sdfs.cpp:92-127.  A volume holds fewer than 2^31 voxels
// (checked at vpt_scene_create), so the eight indices are 32-bit arithmetic on top of one 64-bit base.
VPT_DEV float eval_volume(const DScene& sc, const vpt_volume& vol, f3 uvw) {
  int W = vol.whd[0], H = vol.whd[1], D = vol.whd[2];
  if (W == 0 || H == 0 || D == 0) return 0;
  float s = clampf((uvw.x + 1.0f) * 0.5f, 0.0f, 1.0f) * (W - 1);
  float t = clampf((uvw.y + 1.0f) * 0.5f, 0.0f, 1.0f) * (H - 1);
  float r = clampf((uvw.z + 1.0f) * 0.5f, 0.0f, 1.0f) * (D - 1);
  int i = clampi((int)s, 0, W - 1), j = clampi((int)t, 0, H - 1), k = clampi((int)r, 0, D - 1);
  int ii = min(i + 1, W - 1), jj = min(j + 1, H - 1), kk = min(k + 1, D - 1);
  float u = s - i, v = t - j, w = r - k;
  const float* vox = sc.voxels + vol.offset;
  const int WH = W * H, r0 = j * W + k * WH, r1 = jj * W + k * WH, r2 = j * W + kk * WH, r3 = jj * W + kk * WH;
  float v000 = vox[i + r0], v100 = vox[ii + r0];
  float v010 = vox[i + r1], v001 = vox[i + r2];
  float v011 = vox[i + r3], v101 = vox[ii + r2];
  float v110 = vox[ii + r1], v111 = vox[ii + r3];
  return v000 * (1 - u) * (1 - v) * (1 - w) + v100 * u * (1 - v) * (1 - w) + v010 * (1 - u) * v * (1 - w) +
         v001 * (1 - u) * (1 - v) * w + v011 * (1 - u) * v * w + v101 * u * (1 - v) * w + v110 * u * v * (1 - w) +
         v111 * u * v * w;
}
// eval_sdf(volume, instance, p_local, t) behind the instance's transform, yocto_sdfs.cpp:13 + 30-49.  r2 / hf: the record's
// words [2] and [4] (translation, half box size | translation flag).  (Fetching the next instance's words during this
// evaluation was measured twice - 315 = 315 Msamples/s at four waves per SIMD, 331 against 348 at five: the extra registers spill.)
VPT_DEV float sdf_grid_world(const DScene& sc, const float4* rec, float4 r2, float4 hf, f3 pw, float t) {
  f3 p = (__float_as_int(hf.w) & 1) ? mk3(pw.x + r2.y, pw.y + r2.z, pw.z + r2.w) : transform_point(unpack_frame(rec[0], rec[1], r2), pw);
  float  bbox_dist = sd_box(p - xyz(hf), xyz(hf));
  if (bbox_dist < VPT_FLT_EPS * t) {
    float4 sz = rec[3], g = rec[5], o = rec[6];
    vpt_volume vol;
    vol.whd[0] = __float_as_int(g.x), vol.whd[1] = __float_as_int(g.y), vol.whd[2] = __float_as_int(g.z), vol.res = g.w;
    vol.offset = (long long)(unsigned)__float_as_int(o.x) | ((long long)__float_as_int(o.y) << 32);
    f3 uvw = p * 2.f / xyz(sz) - 1;
    return eval_volume(sc, vol, uvw) * sz.w;
  }
  return bbox_dist;
}
VPT_DEV float sdf_grid_world(const DScene& sc, const sdf_recs& recs, int idx, f3 pw, float t) {
  const float4* rec = recs.grid + 7 * idx;
  return sdf_grid_world(sc, rec, rec[2], rec[4], pw, t);
}
struct sdf_hit { float result; int instance, sdf; };
VPT_DEV sdf_hit eval_sdf_scene(const DScene& sc, const sdf_recs& recs, f3 p, float t) {   // yocto_sdfs.cpp:7-26 (first minimum wins ties)
  sdf_hit res = {VPT_FLT_MAX, -1, -1};
  for (int idx = 0; idx < sc.num_vol_instances; idx++) {
    float d = sdf_grid_world(sc, recs, idx, p, t);
    if (d < res.result) res.result = d, res.instance = idx, res.sdf = -1;
  }
  for (int idx = 0; idx < sc.num_sdfs; idx++) {
    float d = sdf_fn_world(recs, idx, p);
    if (d < res.result) res.result = d, res.instance = -1, res.sdf = idx;
  }
  return res;
}
// tetrahedral 4-tap normals, yocto_sdfs.cpp:67-89
VPT_DEV f3 eval_sdf_normal_function(const sdf_recs& recs, int idx, f3 p, float t) {
  float h = VPT_FLT_EPS * t;
  float d1 = sdf_fn_world(recs, idx, p + mk3(1, -1, -1) * h);
  float d2 = sdf_fn_world(recs, idx, p + mk3(-1, -1, 1) * h);
  float d3 = sdf_fn_world(recs, idx, p + mk3(-1, 1, -1) * h);
  float d4 = sdf_fn_world(recs, idx, p + mk3(1, 1, 1) * h);
  return normalize(mk3(1, -1, -1) * d1 + mk3(-1, -1, 1) * d2 + mk3(-1, 1, -1) * d3 + mk3(1, 1, 1) * d4);
}
VPT_DEV f3 eval_sdf_normal_grid(const DScene& sc, const sdf_recs& recs, int idx, f3 p, float t) {
  float h = VPT_FLT_EPS * t;
  float d1 = sdf_grid_world(sc, recs, idx, p + mk3(1, -1, -1) * h, t);
  float d2 = sdf_grid_world(sc, recs, idx, p + mk3(-1, -1, 1) * h, t);
  float d3 = sdf_grid_world(sc, recs, idx, p + mk3(-1, 1, -1) * h, t);
  float d4 = sdf_grid_world(sc, recs, idx, p + mk3(1, 1, 1) * h, t);
  return normalize(mk3(1, -1, -1) * d1 + mk3(-1, -1, 1) * d2 + mk3(-1, 1, -1) * d3 + mk3(1, 1, 1) * d4);
}

// A march can be cut short without changing its outcome once it can only end in a miss.  For a BOUNDED SDF (or set of
// SDFs) inside the ball (c, r): if the point p = o + t d lies outside the ball of radius 2 r, moves away from c
// (d . (p - c) >= 0) and |o - c| < 1e6 r, no later step of the reference's loop can report a hit:
//   * every later point p' has |p' - c| >= |p - c| > 2 r, so its true distance to the content is >= |p' - c| - r > r;
//   * the sd_* functions are exact distances outside their shapes, evaluated with a relative error of a few ulp, so the
//     value the reference computes is >= 0.999 (|p' - c| - r) > 0;
//   * a hit needs |value| < flt_eps * t' with t' = |p' - o| <= |p' - c| + |c - o|; but 0.999 (x - r) >= 1.2e-7 (x + 1e6 r)
//     for every x >= 2 r (the left side is 0.999 r at x = 2 r and grows faster), so the test fails at every step;
//   * hence the loop ends by i == maxiter or by t overflowing: a miss either way — what is returned here at once.
// (The reference itself needs ~130 doublings of t to get an escaping ray past flt_max.)
VPT_DEV bool march_cannot_hit(f3 c, float r, f3 o, f3 p, f3 d) {
  f3 pc = p - c, oc = o - c;
  return dot(pc, pc) > 4 * r * r && dot(pc, d) >= 0 && dot(oc, oc) < 1e12f * r * r;
}
struct st_hit { bool hit; float dist; int instance, sdf; };
VPT_DEV st_hit spheretrace_one(const DScene& sc, f3 ro, f3 rd, int sdf_handle, int maxiter) {   // cpp:267-286
  st_hit r = {false, VPT_FLT_MAX, -1, -1};
  float  t = VPT_RAY_EPS;
  const sdf_recs recs = scene_sdf_recs(sc);
  float4 bound = recs.fn[6 * sdf_handle + 5];
  for (int i = 0; i < maxiter && t < VPT_FLT_MAX; ++i) {
    f3    p   = ro + rd * t;
    float res = sdf_fn_world(recs, sdf_handle, p);
    if (fabs_(res) < (VPT_FLT_EPS * t)) {
      r.hit = true, r.dist = t, r.sdf = sdf_handle;
      return r;
    }
    if (res > bound.w && bound.w > 0 && march_cannot_hit(xyz(bound), bound.w, ro, p, rd)) return r;   // receding from a bounded SDF: a miss
    t += res;
  }
  return r;
}
// spheretrace(scene, ray, maxiter) (cpp:289-307) is K2's scene_march_step (vpt_implicit_kernel.hip.h), one step per trip

// ------------------------------------------------------------------------------------------------
// lights, yocto_pathtrace.cpp:312-421
// ------------------------------------------------------------------------------------------------
// Scene features a kernel instance is compiled for (template parameter FEAT of the mesh kernels): code for a feature the
// scene does not have costs registers in every path (the allocator serves the worst one), so vpt_capi.hip launches the
// instance without it - 03_volume, whose lights are quads and an environment: 622 -> 676 Msamples/s.
enum { VPT_FEAT_COMPACT_TRIS = 8,   // not a light feature: the instance reads DScene::tri_prims / tri_attrs (scenes whose shapes all hold triangles)
       VPT_FEAT_LARGE_LIGHTS = 1,   // emissive meshes with a real BVH: sample_lights_pdf walks them with extra trips (ST_LPDF)
       VPT_FEAT_SDF_LIGHTS   = 2,   // SDF lights: a sphere trace inside sample_lights_pdf
       VPT_FEAT_SMALL_LIGHTS = 4,   // emissive meshes of a single BVH leaf (area-light quads): walked inline from their light records
       VPT_FEAT_ALL          = 7 };
template <int FEAT = VPT_FEAT_ALL>
VPT_DEV f3 sample_lights(const DScene& sc, f3 position, float rl, float rel, f2 ruv) {
  int           light_id = sample_uniform(sc.num_lights, rl);
  const float4* rec      = sc.light_rec + 8 * (long long)light_id;   // vpt_device.h: everything behind the light id
  float4        r6 = rec[6], r7 = rec[7];
  int           kind = __float_as_int(r7.w) & 255;
  // the one CDF search of this call: emissive mesh -> element, textured environment -> texel
  VPT_T0(TM_CDF);
  int pick = (kind == VPT_LIGHT_SMALL_MESH || kind == VPT_LIGHT_LARGE_MESH || kind == VPT_LIGHT_ENV_TEX) ? sample_light_cdf(sc, light_id, rel) : 0;
  VPT_T1(TM_CDF);
  if ((FEAT & VPT_FEAT_SMALL_LIGHTS) && kind == VPT_LIGHT_SMALL_MESH) {
    // eval_position (yocto_scene.cpp:279-303) from the light's own copy of its <= 4 primitives
    const float4* prims = sc.light_prims + 20 * (long long)light_id;
    int k = 0;
    for (int j = 1; j < ((__float_as_int(r7.w) >> 8) & 15); j++)
      if (__float_as_int(prims[5 * j].w) == pick) k = j;
    float4 c0 = prims[5 * k], c1 = prims[5 * k + 1], c2 = prims[5 * k + 2], c3 = prims[5 * k + 3], cn = prims[5 * k + 4];
    bool   tri = __float_as_int(cn.w) != 0;
    f2     uv  = tri ? mk2(1 - sqrtf(ruv.x), ruv.y * sqrtf(ruv.x)) : ruv;
    f3     lp  = tri ? tri_lerp(xyz(c0), xyz(c1), xyz(c2), uv)
                     : (uv.x + uv.y <= 1 ? tri_lerp(xyz(c0), xyz(c1), xyz(c3), uv) : tri_lerp(xyz(c2), xyz(c3), xyz(c1), 1 - uv));
    return normalize(transform_point(unpack_frame(rec[3], rec[4], rec[5]), lp) - position);
  } else if ((FEAT & VPT_FEAT_LARGE_LIGHTS) && kind == VPT_LIGHT_LARGE_MESH) {
    const DInstance& inst = sc.instances[sc.lights[light_id].instance];
    f2  uv      = sc.shapes[inst.shape].is_triangles ? mk2(1 - sqrtf(ruv.x), ruv.y * sqrtf(ruv.x)) : ruv;
    return normalize(eval_position(sc, inst, pick, uv) - position);
  } else if ((FEAT & VPT_FEAT_SDF_LIGHTS) && kind == VPT_LIGHT_SDF) {
    int sdf_id = sc.lights[light_id].sdf;
    const vpt_sdf& sdf = sc.sdfs[sdf_id];
    f3 wlightp  = transform_point(load_frame(sc.sdf_inv + 3 * sdf_id), mk3(ruv.x, ruv.y, 1) * ld3(sdf.whd));
    return normalize(wlightp - position);
  } else if (kind == VPT_LIGHT_ENV_TEX) {
    int tw = __float_as_int(r6.x), th = __float_as_int(r6.y);
    f2  uv = mk2(((pick % tw) + 0.5f) / tw, ((pick / tw) + 0.5f) / th);
    return transform_direction(unpack_frame(rec[3], rec[4], rec[5]),
        mk3(cosf(uv.x * 2 * VPT_PI) * sinf(uv.y * VPT_PI), cosf(uv.y * VPT_PI), sinf(uv.x * 2 * VPT_PI) * sinf(uv.y * VPT_PI)));
  } else if (kind == VPT_LIGHT_ENV_CONST) {
    return sample_sphere(ruv);
  }
  return mk3(0, 0, 0);
}
// sample_lights_pdf (yocto_pathtrace.cpp:353-421) is evaluated per light from the light records: small_light_pdf /
// general_light_pdf / other_light_pdf in vpt_mesh_kernel.hip.h, the SDF-light march of K2 in vpt_implicit_kernel.hip.h.
