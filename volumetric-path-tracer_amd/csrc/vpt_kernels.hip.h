// vpt_kernels.hip.h — what the render kernels share (slot map, MIS direction choice, roulette, launch
// schedule), K2 — the kernel of the two SDF shaders — and the elementwise state kernels.
//
// K2 vpt_render_kernel<K_IMPLICIT | K_IMPLICIT_NORMAL>: shade_implicit / shade_implicit_normal
// (yocto_pathtrace.cpp:425-562) with the skeleton K1 uses (vpt_mesh_kernel.hip.h): one workgroup = one
// wave64 = one 8x8 pixel tile of the tile-major state; one lane owns one pixel for the whole launch (all
// `nsamples` passes) and keeps its PCG32 stream, radiance sum and hit count in registers, so HBM state is
// read once and written once per launch; paths are regenerated per lane (the reference's samples x bounces
// loops, cpp:1081-1090 x 441-532, flattened into one loop whose trip is one path vertex = one sphere trace +
// shading), and a pixel's samples are consumed serially from its own stream, so results do not depend on how
// lanes interleave.  Waves start longest first (sched_cfg).
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

// MIS direction choice of shade_implicit (yocto_pathtrace.cpp:488-519; `noimplicit_mis` switches MIS off).
// RNG draw order is the reference's right-to-left argument evaluation.
// Returns false when the path ends (`incoming == 0`).
VPT_DEV bool next_direction(const DScene& sc, const DParams& pr, const mpoint& m, f3 normal, f3 outgoing, f3 position,
    rng_t& rng, f3& weight, f3& incoming, const lane_stack& stk) {
  incoming = mk3(0, 0, 0);
  if (!is_delta(m)) {
    bool  mis  = !pr.noimplicit_mis;
    float coin = rand1f(rng);
    if (coin < (mis ? 0.5f : 1.0f)) {
      f2 rn;
      rn.x      = rand1f(rng);
      rn.y      = rand1f(rng);
      float rnl = rand1f(rng);
      incoming  = sample_bsdfcos(m, normal, outgoing, rnl, rn);
    } else {
      f2 ruv;
      ruv.x     = rand1f(rng);
      ruv.y     = rand1f(rng);
      float rel = rand1f(rng);
      float rl  = rand1f(rng);
      incoming  = sample_lights(sc, position, rl, rel, ruv);
    }
    if (is_zero3(incoming)) return false;
    f3    f   = eval_bsdfcos(m, normal, outgoing, incoming);
    float pdf = sample_bsdfcos_pdf(m, normal, outgoing, incoming);
    if (mis) pdf = 0.5f * pdf + 0.5f * sample_lights_pdf(sc, position, incoming, pr.spheretrace_maxiter, stk);
    weight = weight * (f / pdf);
  } else {
    float rnl = rand1f(rng);
    incoming  = sample_delta(m, normal, outgoing, rnl);
    weight    = weight * (eval_delta(m, normal, outgoing, incoming) / sample_delta_pdf(m, normal, outgoing, incoming));
  }
  return true;
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
  const int* order;   // blockIdx.x -> wave index, or null: identity
  unsigned*  cost;    // per wave: duration of this launch in 100 MHz ticks, or null
};
#ifndef VPT_K2_WAVES
#define VPT_K2_WAVES 4   // measured on 06_gridsdf_synth: 2 -> 95, 3 -> 117, 4 -> 125, 5 -> 116, 6 -> 108, 8 -> 96 Msamples/s
#endif
template <int SH>
__global__ void __launch_bounds__(VPT_BLOCK, VPT_K2_WAVES) vpt_render_kernel(DScene sc, DParams pr, float4* __restrict__ image,
    int* __restrict__ hits, ulonglong2* __restrict__ rngs, int stack_cap, sched_cfg sched) {
  extern __shared__ int lds_stack[];
  lane_stack stk;
  stk.base = lds_stack + threadIdx.x;
  stk.cap  = stack_cap;
  const unsigned long long wave_start = wall_clock64();
  const int wave = sched.order ? sched.order[blockIdx.x] : (int)blockIdx.x;

  int slot = wave * VPT_BLOCK + threadIdx.x;
  int px = 0, py = 0;
  if (slot >= pr.nslots || !slot_to_pixel(pr, slot, px, py)) return;   // padding lanes own no pixel

  // ---- pixel state: one coalesced read, kept in registers for the whole launch ----------------
  float4     acc_in = image[slot];
  f4         acc    = mk4(acc_in.x, acc_in.y, acc_in.z, acc_in.w);
  ulonglong2 r_in   = rngs[slot];
  rng_t      rng    = {r_in.x, r_in.y};
  const vpt_camera& cam = sc.cameras[pr.camera];
  const int nb = pr.bounces;

  // ---- path state --------------------------------------------------------------------------------
  ray_t ray    = make_ray(mk3(0, 0, 0), mk3(0, 0, 1));
  f3    radiance = mk3(0, 0, 0), weight = mk3(1, 1, 1);
  float alpha  = 0;
  int   bounce = 0, sample = 0;
  bool  fresh  = true;

  while (true) {
    if (fresh) {
      if (sample == pr.nsamples) break;
      float u, v;
      if (pr.preview) {
        u = (px + 0.5f) / pr.width, v = (py + 0.5f) / pr.height;
      } else {
        u = (px + rand1f(rng)) / pr.width;
        v = (py + rand1f(rng)) / pr.height;
      }
      f2 lens;
      lens.x = rand1f(rng);
      lens.y = rand1f(rng);
      ray    = eval_camera(cam, mk2(u, v), lens);
      radiance = mk3(0, 0, 0), weight = mk3(1, 1, 1);
      alpha = (SH == K_IMPLICIT) ? 1.0f : 0.0f;
      bounce = 0, fresh = false;
    }

    bool finish = false;
    if constexpr (SH == K_IMPLICIT_NORMAL) {   // cpp:538-562
      st_hit h = spheretrace(sc, ray, pr.spheretrace_maxiter);
      if (h.hit) {
        f3 position = ray_point(ray, h.dist);
        f3 n = h.instance != VPT_INVALID ? eval_sdf_normal_grid(sc, sc.vol_instances[h.instance], position, h.dist)
                                         : eval_sdf_normal_function(sc.sdfs[h.sdf], position, h.dist);
        radiance = n * 0.5f + 0.5f;
        alpha    = 1;
      }
      finish = true;
    } else if (bounce >= nb) {
      finish = true;
    } else {   // shade_implicit, cpp:425-535
      st_hit h = spheretrace(sc, ray, pr.spheretrace_maxiter);
      if (!h.hit) {
        radiance = radiance + weight * eval_environment(sc, ray.d);
        finish   = true;
      } else {
        f3 outgoing = -ray.d;
        f3 position = ray_point(ray, h.dist);
        f3 normal   = h.instance != VPT_INVALID ? eval_sdf_normal_grid(sc, sc.vol_instances[h.instance], position, h.dist)
                                                : eval_sdf_normal_function(sc.sdfs[h.sdf], position, h.dist);
        int    mat  = h.instance != VPT_INVALID ? sc.vol_instances[h.instance].material : sc.sdfs[h.sdf].material;
        mpoint m    = eval_material_plain(sc, mat);
        if (m.opacity < 1 && rand1f(rng) >= m.opacity) {
          ray = make_ray(position + ray.d * 1e-2f, ray.d);   // bounce -= 1; continue
        } else {
          radiance = radiance + weight * eval_emission(m.emission, normal, outgoing);
          f3 incoming;
          if (!next_direction(sc, pr, m, normal, outgoing, position, rng, weight, incoming, stk)) finish = true;
          else {
            ray = make_ray(position, incoming);
            if (!survive(weight, bounce, rng)) finish = true;
            bounce++;
          }
        }
      }
    }

    if (finish) {   // cpp:1087-1089
      f4 rad = mk4(radiance.x, radiance.y, radiance.z, alpha);
      if (!(isfinite(rad.x) && isfinite(rad.y) && isfinite(rad.z) && isfinite(rad.w))) rad = mk4(0, 0, 0, 0);
      acc = acc + rad;
      sample++;
      fresh = true;
    }
  }

  image[slot] = make_float4(acc.x, acc.y, acc.z, acc.w);
  hits[slot] += pr.nsamples;
  ulonglong2 r_out;
  r_out.x = rng.state, r_out.y = rng.inc;
  rngs[slot] = r_out;
  if (sched.cost && threadIdx.x == 0) {
    unsigned long long dt = wall_clock64() - wave_start;
    sched.cost[wave] = dt < 0xffffffffull ? (unsigned)dt : 0xffffffffu;
  }
}

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
