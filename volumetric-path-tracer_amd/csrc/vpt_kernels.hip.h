// vpt_kernels.hip.h — what the render kernels share (slot map, roulette, launch schedule) and the elementwise
// state kernels (layout conversion, get_render, output quantisation).  The render kernels themselves:
// vpt_mesh_kernel.hip.h (K1, the seven mesh shaders), vpt_implicit_kernel.hip.h (K2, the two SDF shaders).
#pragma once
#include "vpt_scene.hip.h"

enum { K_VOLPATH = 0, K_PATH = 1, K_NAIVE = 2, K_EYELIGHT = 3, K_DEBUG = 4, K_IMPLICIT = 5, K_IMPLICIT_NORMAL = 6 };

// slot -> pixel for the tile-major layout of include/vpt.h (vpt_layout)
VPT_DEV bool slot_to_pixel(const DParams& pr, int slot, int& px, int& py) {
  int per_tile   = pr.tile_w * pr.tile_h;
  int local_tile = slot / per_tile, p = slot - local_tile * per_tile;
  int tile       = local_tile * pr.nranks + pr.rank;
  if (tile >= pr.tiles_x * pr.tiles_y) return false;
  int ty = tile / pr.tiles_x, tx = tile - ty * pr.tiles_x;
  // inside a tile, pixels are ordered in 8x8 blocks so that a wave covers a square footprint
  int bw = pr.tile_w >> 3, blk = p >> 6, q = p & 63;
  int by = blk / bw, bx = blk - by * bw;
  px = tx * pr.tile_w + bx * 8 + (q & 7);
  py = ty * pr.tile_h + by * 8 + (q >> 3);
  return px < pr.width && py < pr.height;
}

// weight test + russian roulette, yocto_pathtrace.cpp:676-683
VPT_DEV bool survive(f3& weight, int bounce, rng_t& rng) {
  if (is_zero3(weight) || !finite3(weight)) return false;
  if (bounce > 3) {
    float rr_prob = fmin_(0.99f, max3(weight));
    if (rand1f(rng) >= rr_prob) return false;
    weight = weight * (1 / rr_prob);
  }
  return true;
}

// Launch schedule.  A wave's 64 pixels run all their samples in sequence, so a wave's duration is fixed by its
// tile's content and varies 15x across 03_volume; in tile order the launch ends with a third of the GPU idle
// behind a few long waves.  Every wave records its duration; the next launch on the same layout starts the
// waves longest first (order[] = wave indices by descending cost: LPT list scheduling).  Results do not depend
// on the order (pixels are independent), only the makespan does.
struct sched_cfg {
  const int* order;       // blockIdx.x -> wave index, or null: identity
  unsigned*  cost;        // per wave: duration of this launch in 100 MHz ticks, or null
  const int* lane_slot;   // [wave][64] -> state slot of the lane (-1: none), or null: slot = wave * 64 + lane (one tile per wave).
                          // Set when costly tiles run as several partly filled waves (vpt_capi.hip: tile splitting)
};
#ifndef VPT_INSTANCES_TU   // (the translation units that only instantiate render kernels - vpt_k1_instances.hip.h - skip the plain kernels)
// ---- state layout conversion and output resolve ---------------------------------------------
// row-major host-order arrays <-> this rank's tile-major slots (vpt_state_upload / _download)
__global__ void vpt_permute_kernel(DParams pr, int to_tiles, float4* tiles_image, int* tiles_hits, ulonglong2* tiles_rng,
    float4* rows_image, int* rows_hits, ulonglong2* rows_rng) {
  int slot = blockIdx.x * blockDim.x + threadIdx.x;
  int px, py;
  if (slot >= pr.nslots || !slot_to_pixel(pr, slot, px, py)) return;
  long long idx = (long long)py * pr.width + px;
  if (to_tiles) tiles_image[slot] = rows_image[idx], tiles_hits[slot] = rows_hits[idx], tiles_rng[slot] = rows_rng[idx];
  else rows_image[idx] = tiles_image[slot], rows_hits[idx] = tiles_hits[slot], rows_rng[idx] = tiles_rng[slot];
}
// get_render (cpp:1105-1116) over the gathered buffers of all ranks: [nranks][nslots] -> row-major * 1/samples
__global__ void vpt_resolve_kernel(DParams pr, const float4* tiles_all, float scale, float4* rows_image) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;   // global slot over all ranks
  if (g >= pr.nslots * pr.nranks) return;
  DParams q = pr;
  q.rank    = g / pr.nslots;
  int px, py;
  if (!slot_to_pixel(q, g - q.rank * pr.nslots, px, py)) return;
  float4 v = tiles_all[g];
  rows_image[(long long)py * pr.width + px] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
}

// The output stage of a preview on the device: get_render (cpp:1105-1116) followed by rgb_to_srgb
// (yocto_color.h:228-231) and float_to_byte (:207-211, clamp(int(a * 256), 0, 255)); alpha is quantised
// linearly, as save_image does.  powf is ocml's here and glibc's in the reference: a byte can differ by one
// where the curve lands within an ulp of a multiple of 1/256 (the parity pipeline keeps using the host
// routine, vpth_linear_to_srgb8).
__global__ void vpt_resolve_srgb8_kernel(DParams pr, const float4* tiles_all, float scale, uchar4* rows_rgba8) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;   // global slot over all ranks
  if (g >= pr.nslots * pr.nranks) return;
  DParams q = pr;
  q.rank    = g / pr.nslots;
  int px, py;
  if (!slot_to_pixel(q, g - q.rank * pr.nslots, px, py)) return;
  float4 v = tiles_all[g];
  auto curve = [](float rgb) { return (rgb <= 0.0031308f) ? 12.92f * rgb : (1 + 0.055f) * powf(rgb, 1 / 2.4f) - 0.055f; };
  auto quant = [](float a) {
    int b = (int)(a * 256);
    return (unsigned char)(b < 0 ? 0 : (b > 255 ? 255 : b));
  };
  rows_rgba8[(long long)py * pr.width + px] =
      make_uchar4(quant(curve(v.x * scale)), quant(curve(v.y * scale)), quant(curve(v.z * scale)), quant(v.w * scale));
}
#endif
