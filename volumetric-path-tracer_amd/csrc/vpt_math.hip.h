// vpt_math.hip.h — float32 vector math for the gfx950 kernels.
//
// Numerical contract: the kernels must take the same branch as the
// reference CPU renderer on (almost) every random draw, so every expression keeps the
// reference's association order (yocto_math.h), min/max/abs are the NaN-asymmetric ternary forms
// (yocto_math.h:1354-1356), divisions stay divisions (hipcc's default correctly-rounded f32
// div/sqrt), and the translation unit is built with -ffp-contract=off (no FMA fusion).
#pragma once
#include <hip/hip_runtime.h>

#define VPT_DEV __device__ __forceinline__

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct m33 { f3 x, y, z; };
struct frame { f3 x, y, z, o; };

#define VPT_PI 3.14159274101257324f        /* (float)3.14159265358979323846 */
#define VPT_FLT_MAX 3.402823466e+38f
#define VPT_FLT_EPS 1.1920928955078125e-07f
#define VPT_RAY_EPS 1e-4f

VPT_DEV float fmin_(float a, float b) { return (a < b) ? a : b; }
VPT_DEV float fmax_(float a, float b) { return (a > b) ? a : b; }
VPT_DEV float fabs_(float a) { return a < 0 ? -a : a; }
VPT_DEV float clampf(float a, float lo, float hi) { return fmin_(fmax_(a, lo), hi); }
VPT_DEV int   clampi(int a, int lo, int hi) { int m = a > lo ? a : lo; return m < hi ? m : hi; }

VPT_DEV f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
VPT_DEV f2 mk2(float x, float y) { f2 r = {x, y}; return r; }
VPT_DEV f4 mk4(float x, float y, float z, float w) { f4 r = {x, y, z, w}; return r; }
VPT_DEV f3 xyz(float4 a) { return mk3(a.x, a.y, a.z); }
VPT_DEV f3 xyz(f4 a) { return mk3(a.x, a.y, a.z); }

VPT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
VPT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
VPT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
VPT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
VPT_DEV f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
VPT_DEV f3 operator+(f3 a, float b) { return mk3(a.x + b, a.y + b, a.z + b); }
VPT_DEV f3 operator-(f3 a, float b) { return mk3(a.x - b, a.y - b, a.z - b); }
VPT_DEV f3 operator*(f3 a, float b) { return mk3(a.x * b, a.y * b, a.z * b); }
VPT_DEV f3 operator/(f3 a, float b) { return mk3(a.x / b, a.y / b, a.z / b); }
VPT_DEV f3 operator+(float a, f3 b) { return mk3(a + b.x, a + b.y, a + b.z); }
VPT_DEV f3 operator-(float a, f3 b) { return mk3(a - b.x, a - b.y, a - b.z); }
VPT_DEV f3 operator*(float a, f3 b) { return mk3(a * b.x, a * b.y, a * b.z); }
VPT_DEV bool eq3(f3 a, f3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
VPT_DEV bool is_zero3(f3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
VPT_DEV f2 operator+(f2 a, f2 b) { return mk2(a.x + b.x, a.y + b.y); }
VPT_DEV f2 operator-(f2 a, f2 b) { return mk2(a.x - b.x, a.y - b.y); }
VPT_DEV f2 operator*(f2 a, float b) { return mk2(a.x * b, a.y * b); }
VPT_DEV f2 operator-(float a, f2 b) { return mk2(a - b.x, a - b.y); }
VPT_DEV f4 operator+(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
VPT_DEV f4 operator*(f4 a, float b) { return mk4(a.x * b, a.y * b, a.z * b, a.w * b); }

VPT_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
VPT_DEV float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
VPT_DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
VPT_DEV float length(f3 a) { return sqrtf(dot(a, a)); }
VPT_DEV float length(f2 a) { return sqrtf(dot(a, a)); }
VPT_DEV f3 normalize(f3 a) { float l = length(a); return (l != 0) ? a / l : a; }
VPT_DEV float distance_squared(f3 a, f3 b) { return dot(a - b, a - b); }
VPT_DEV f3 orthonormalize(f3 a, f3 b) { return normalize(a - b * dot(a, b)); }
VPT_DEV f3 reflect(f3 w, f3 n) { return -w + 2 * dot(n, w) * n; }
VPT_DEV f3 refract(f3 w, f3 n, float inv_eta) {
  float cosine = dot(n, w);
  float k      = 1 + inv_eta * inv_eta * (cosine * cosine - 1);
  if (k < 0) return mk3(0, 0, 0);
  return -w * inv_eta + (inv_eta * cosine - sqrtf(k)) * n;
}
VPT_DEV f3 vmaxs(f3 a, float b) { return mk3(fmax_(a.x, b), fmax_(a.y, b), fmax_(a.z, b)); }
VPT_DEV f3 vmin3(f3 a, f3 b) { return mk3(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)); }
VPT_DEV f3 vmax3(f3 a, f3 b) { return mk3(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)); }
VPT_DEV f3 vclamp(f3 a, float lo, float hi) { return mk3(clampf(a.x, lo, hi), clampf(a.y, lo, hi), clampf(a.z, lo, hi)); }
VPT_DEV f3 vabs(f3 a) { return mk3(fabs_(a.x), fabs_(a.y), fabs_(a.z)); }
VPT_DEV f3 vsqrt(f3 a) { return mk3(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)); }
VPT_DEV float max3(f3 a) { return fmax_(fmax_(a.x, a.y), a.z); }
VPT_DEV float min3(f3 a) { return fmin_(fmin_(a.x, a.y), a.z); }
VPT_DEV float sum3(f3 a) { return a.x + a.y + a.z; }
VPT_DEV float mean3(f3 a) { return sum3(a) / 3; }
VPT_DEV bool finite3(f3 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }
VPT_DEV f3 lerp3(f3 a, f3 b, float u) { return a * (1 - u) + b * u; }
VPT_DEV float comp(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

VPT_DEV f3 mul(m33 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
VPT_DEV f3 transform_point(const frame& a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.o; }
VPT_DEV f3 transform_vector(const frame& a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
VPT_DEV f3 transform_direction(const frame& a, f3 b) { return normalize(transform_vector(a, b)); }
VPT_DEV f3 transform_direction(m33 a, f3 b) { return normalize(mul(a, b)); }

// basis_fromz, yocto_math.h:2811-2820
VPT_DEV m33 basis_fromz(f3 v) {
  f3    z    = normalize(v);
  float sign = copysignf(1.0f, z.z);
  float a    = -1.0f / (sign + z.z);
  float b    = z.x * z.y * a;
  m33   r;
  r.x = mk3(1.0f + sign * z.x * z.x * a, sign * b, -sign * z.x);
  r.y = mk3(b, sign + z.y * z.y * a, -z.y);
  r.z = z;
  return r;
}

// frames packed in 3 float4 (see vpt_device.h)
VPT_DEV frame unpack_frame(float4 a, float4 b, float4 c) {
  frame f;
  f.x = mk3(a.x, a.y, a.z), f.y = mk3(a.w, b.x, b.y), f.z = mk3(b.z, b.w, c.x), f.o = mk3(c.y, c.z, c.w);
  return f;
}
VPT_DEV frame load_frame(const float4* p) { return unpack_frame(p[0], p[1], p[2]); }
VPT_DEV frame load_frame(const vpt_frame& v) {
  frame f;
  f.x = mk3(v.x[0], v.x[1], v.x[2]), f.y = mk3(v.y[0], v.y[1], v.y[2]);
  f.z = mk3(v.z[0], v.z[1], v.z[2]), f.o = mk3(v.o[0], v.o[1], v.o[2]);
  return f;
}
VPT_DEV f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// The 100 MHz wall clock (s_memrealtime) WITHOUT telling the compiler that memory changed.  wall_clock64(),
// __builtin_readcyclecounter() and a volatile asm all count as a write to unknown memory, and after one such call the
// compiler turns reads of wave-uniform tables from scalar loads (s_load, SGPR results) into per-lane vector loads
// (checked on a six-line kernel).  This asm is pure in the compiler's eyes; `dep` is any value that must be computed
// before the read (it keeps two reads from being merged and this one from moving above that value), and a reader that
// must stay at the kernel's start parks the result in LDS at once (a store the read cannot sink below).  Measured
// effect on the shipped kernels: none (K1 622 = 622, K2 252 = 255 Msamples/s) - what sits on K2's step was the
// record reads, now staged in LDS; kept because it costs nothing and removes a trap.
VPT_DEV unsigned long long clock_ticks(int dep) {
  unsigned long long t;
  asm("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(dep));
  return t;
}

// PCG32, yocto_sampling.h:184-216 — bit-exact integer arithmetic
struct rng_t { unsigned long long state, inc; };
VPT_DEV unsigned int advance_rng(rng_t& rng) {
  unsigned long long old = rng.state;
  rng.state              = old * 6364136223846793005ULL + rng.inc;
  unsigned int xs        = (unsigned int)(((old >> 18u) ^ old) >> 27u);
  unsigned int rot       = (unsigned int)(old >> 59u);
  return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
VPT_DEV float rand1f(rng_t& rng) { return __uint_as_float((advance_rng(rng) >> 9) | 0x3f800000u) - 1.0f; }

// ------------------------------------------------------------------------------------------------
// Hardware min/max (no sNaN quieting moves around them: the operands below are never NaN)
VPT_DEV float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
VPT_DEV float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
VPT_DEV float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
VPT_DEV float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// 1/x bit-identical to the IEEE quotient 1.0f/x.  v_rcp_f32 followed by one Newton step is correctly
// rounded for EVERY float whose biased exponent is 1..250 (2^-126 <= |x| < 2^124): checked exhaustively on
// gfx950 by vpt_selftest_reciprocal() (tests/test_gpu_parity.py).  Other inputs (0, denormals, huge, inf,
// NaN) take the division; the choice is made per wave so that only one of the two sequences is executed.
VPT_DEV float rcp_newton(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
VPT_DEV bool rcp_in_range(float lo_abs, float hi_abs) { return lo_abs >= 0x1p-126f && hi_abs < 0x1p124f; }
VPT_DEV float rcp_exact(float x) {
  float a = __builtin_fabsf(x);
  if (__builtin_amdgcn_ballot_w64(!rcp_in_range(a, a)) == 0) return rcp_newton(x);
  return 1 / x;
}
VPT_DEV f3 rcp3_exact(f3 d) {
  float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
  if (__builtin_amdgcn_ballot_w64(!rcp_in_range(hw_min3(ax, ay, az), hw_max3(ax, ay, az))) == 0)
    return mk3(rcp_newton(d.x), rcp_newton(d.y), rcp_newton(d.z));
  return mk3(1 / d.x, 1 / d.y, 1 / d.z);
}
