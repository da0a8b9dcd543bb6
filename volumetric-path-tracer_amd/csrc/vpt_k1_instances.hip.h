// The instances of K1 for the two path tracers (main and pilot kernel, both stack variants, every light-feature set, the short
// triangle records) take most of the library's compile time.  The Makefile's build (-DVPT_SPLIT_TUS) compiles them in translation units of
// their own - vpt_k1_volpath.hip, vpt_k1_path.hip: explicit instantiations - next to vpt_capi.hip, which then only declares them (`make -j`:
// 4.6 -> 2.4 minutes).  Experiment builds (`make variant`) keep everything in one unit: their diagnostic device globals are per unit.
#pragma once
#include "vpt_mesh_kernel.hip.h"
#define VPT_K1_SPLIT_INSTANCES(X, K)                                                                                                        \
  X(vpt_mesh_kernel, K, true, 4) X(vpt_mesh_kernel, K, false, 4) X(vpt_mesh_kernel, K, true, 5) X(vpt_mesh_kernel, K, false, 5)             \
  X(vpt_mesh_kernel, K, true, 7) X(vpt_mesh_kernel, K, false, 7) X(vpt_mesh_kernel, K, true, 12) X(vpt_mesh_kernel, K, false, 12)           \
  X(vpt_mesh_pilot_kernel, K, true, 4) X(vpt_mesh_pilot_kernel, K, false, 4) X(vpt_mesh_pilot_kernel, K, true, 5)                           \
  X(vpt_mesh_pilot_kernel, K, false, 5) X(vpt_mesh_pilot_kernel, K, true, 7) X(vpt_mesh_pilot_kernel, K, false, 7)
#define VPT_K1_DEFINE(NAME, K, S, F) template __global__ void NAME<K, S, F>(DScene, DParams, float4* __restrict__, int* __restrict__, ulonglong2* __restrict__, stack_cfg, sched_cfg);
#define VPT_K1_DECLARE(NAME, K, S, F) extern VPT_K1_DEFINE(NAME, K, S, F)
static_assert((VPT_FEAT_SMALL_LIGHTS | VPT_FEAT_COMPACT_TRIS) == 12 && (VPT_FEAT_SMALL_LIGHTS | VPT_FEAT_LARGE_LIGHTS) == 5 && VPT_FEAT_ALL == 7, "feature sets of the list above");
