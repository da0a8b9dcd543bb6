// vpt_hostmath.h — the handful of float32 vector helpers the load-time host code needs.
// Semantics follow yocto_math.h exactly where results must be bit-identical to the reference
// (ternary min/max: yocto_math.h:1355-1356; center/merge: yocto_geometry.h:384-401).
// Compile with -ffp-contract=off.
#pragma once
#include <cmath>

#include "vpt_host.h"

namespace vpt {

inline const float pif     = (float)3.14159265358979323846;
inline const float flt_max = 3.402823466e+38f;

inline float fmin_(float a, float b) { return (a < b) ? a : b; }
inline float fmax_(float a, float b) { return (a > b) ? a : b; }

inline vec3f operator+(const vec3f& a, const vec3f& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3f operator-(const vec3f& a, const vec3f& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3f operator*(const vec3f& a, float b) { return {a.x * b, a.y * b, a.z * b}; }
inline vec3f operator/(const vec3f& a, float b) { return {a.x / b, a.y / b, a.z / b}; }
inline bool  operator==(const vec3f& a, const vec3f& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline float dot(const vec3f& a, const vec3f& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3f cross(const vec3f& a, const vec3f& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(const vec3f& a) { return std::sqrt(dot(a, a)); }
inline vec3f vmin(const vec3f& a, const vec3f& b) { return {fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; }
inline vec3f vmax(const vec3f& a, const vec3f& b) { return {fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
inline float comp(const vec3f& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// transform_point, yocto_math.h:3097
inline vec3f transform_point(const frame3f& f, const vec3f& p) {
  return f.x * p.x + f.y * p.y + f.z * p.z + f.o;
}

struct bbox3f {
  vec3f min = {flt_max, flt_max, flt_max};
  vec3f max = {-flt_max, -flt_max, -flt_max};
};
inline bbox3f merge(const bbox3f& a, const vec3f& b) { return {vmin(a.min, b), vmax(a.max, b)}; }
inline bbox3f merge(const bbox3f& a, const bbox3f& b) { return {vmin(a.min, b.min), vmax(a.max, b.max)}; }
inline vec3f  center(const bbox3f& a) { return (a.min + a.max) / 2; }

// triangle_area / quad_area, yocto_geometry.h:506-518
inline float triangle_area(const vec3f& p0, const vec3f& p1, const vec3f& p2) {
  return length(cross(p1 - p0, p2 - p0)) / 2;
}
inline float quad_area(const vec3f& p0, const vec3f& p1, const vec3f& p2, const vec3f& p3) {
  return triangle_area(p0, p1, p3) + triangle_area(p2, p3, p1);
}

}  // namespace vpt
