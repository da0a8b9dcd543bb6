// vpt_implicit_kernel.hip.h — K2, the kernel of the two SDF shaders: shade_implicit / shade_implicit_normal
// (yocto_pathtrace.cpp:425-562) over spheretrace (:267-307), eval_sdf_scene / eval_sdf / eval_volume / eval_sdf_normal
// (yocto_sdfs.cpp:7-127) and the analytic sd_* (yocto_sdfs.h:43-80).
//
// Design.  Like K1 (vpt_mesh_kernel.hip.h): one workgroup = one wave64 = one 8x8 pixel tile of the tile-major
// state; one lane owns one pixel for the whole launch (all `nsamples` passes), keeps its PCG32 stream, radiance sum
// and hit count in registers (HBM state is read once and written once per launch) and consumes the pixel's samples
// serially from its own stream, so results do not depend on how lanes interleave; waves start longest first.
//
// What differs is the unit of lockstep.  A sphere trace is a chain of up to `spheretrace_maxiter` (450) dependent
// steps, each an evaluation of every SDF of the scene, and its length varies from 3 steps (a ray into a nearby
// surface) to all 450 (a ray grazing the floor towards the horizon).  With "one path vertex per trip" (the first
// version of this kernel) a wave sat through its longest march at every vertex: 24 % of the VALU lanes did work and
// the kernel was VALU-issue bound at that utilisation (profiles/r02_k2_v1_*).  Here the trip is ONE MARCH STEP:
// every lane carries a small state machine
//     M_SCENE   marching the scene SDF            (t, it)            -> M_HIT | M_MISS
//     M_LIGHT   marching one SDF light for its pdf (lt, lit, lp_light) -> M_LIGHTS
//     M_HIT / M_MISS / M_LIGHTS / M_NEW            wants the shading block
// and the wave alternates between march steps (all lanes that march, whatever vertex / sample they are at) and the
// shading block, which is entered when VPT_K2_SHADE_AT lanes wait for it or nobody marches any more.  A lane's own
// arithmetic — the t sequence of its marches, its draws — is the reference's, step for step; only the interleaving
// across lanes changed.  The MIS light-pdf loop over the lights (cpp:353-421) is resumable like K1's: mesh and
// environment lights are evaluated inline from the light records, an SDF light hands the lane to M_LIGHT.
//
// Pixels are handed out DYNAMICALLY.  A pixel's cost varies by two orders of magnitude (sky: one short march per
// sample; floor towards the horizon: several 450-step marches), and with one fixed pixel per lane a wave's lanes ran
// out of work one after another: 36 % of the VALU lanes worked (profiles/r02_k2_v2_*).  Now the launch is a fixed
// number of resident waves and a lane that has finished its pixel's samples writes the pixel's state back and takes
// the next pixel from a queue (one wave-aggregated atomic per batch of fetching lanes), longest pixel first: every
// pixel records how many trips it kept its lane busy, and the host sorts the queue by that for the next launch on the
// same layout.  A pixel still belongs to exactly one lane for all its samples, so results are unchanged.
#pragma once
#include "vpt_mesh_kernel.hip.h"

#ifndef VPT_K2_WAVES
#define VPT_K2_WAVES 4
#endif
#ifndef VPT_K2_SHADE_AT
#define VPT_K2_SHADE_AT 16   // lanes waiting for the shading block before the wave runs it
#endif
#ifndef VPT_K2_WATCHDOG_TICKS
#define VPT_K2_WATCHDOG_TICKS 30000000000ull   // 300 s: two orders of magnitude above the longest wave of any test workload
#endif
#ifndef VPT_K2_STEPS
#define VPT_K2_STEPS 4       // march steps between two looks at the wave's state
#endif

enum { M_NEW = 0, M_SCENE = 1, M_HIT = 2, M_MISS = 3, M_LIGHT = 4, M_LIGHTS = 5, M_DONE = 6, M_FETCH = 7 };

// one step of spheretrace(scene, ray, maxiter) (yocto_pathtrace.cpp:289-307); returns the lane's next mode
VPT_DEV int scene_march_step(const DScene& sc, f3 ro, f3 rd, int maxiter, float& t, int& it, int& hit_instance, int& hit_sdf) {
  if (!(it < maxiter && t < VPT_FLT_MAX)) return M_MISS;
  sdf_hit res = eval_sdf_scene(sc, ro + rd * t, t);
  if (fabs_(res.result) < (VPT_FLT_EPS * t)) {
    hit_instance = res.instance, hit_sdf = res.sdf;
    return M_HIT;
  }
  t += res.result, it++;
  return M_SCENE;
}
// one step of spheretrace(scene, ray, sdf, maxiter) (:267-286) inside sample_lights_pdf (:382-394) for SDF light `sdf`:
// on a hit adds the light's pdf term (normal at `position`, sic, :389) to `sum`; returns false when the march ended
VPT_DEV bool light_march_step(const DScene& sc, const vpt_sdf& sdf, float area, f3 position, f3 direction, int maxiter, float& lt, int& lit, float& sum) {
  if (!(lit < maxiter && lt < VPT_FLT_MAX)) return false;
  float res = eval_sdf_function(sdf, transform_point(load_frame(sdf.frame), position + direction * lt));
  if (fabs_(res) < (VPT_FLT_EPS * lt)) {
    f3 lposition = position + direction * lt;
    f3 lnormal   = eval_sdf_normal_function(sdf, position, lt);
    sum += distance_squared(lposition, position) / (fabs_(dot(lnormal, direction)) * area);
    return false;
  }
  lt += res, lit++;
  return true;
}
// the lights of sample_lights_pdf that need no march (mesh lights, environments), one light
VPT_DEV float inline_light_pdf(const DScene& sc, int l, int kind, float4 r6, float4 r7, f3 position, f3 direction, const lane_stack& stk) {
  if (kind == VPT_LIGHT_SMALL_MESH) return small_light_pdf(sc, l, r6, r7, position, direction);
  if (kind == VPT_LIGHT_LARGE_MESH) return general_light_pdf(sc, sc.lights[l], position, direction, stk);
  return other_light_pdf(sc, l, kind, r6, position, direction, 0);   // environments (an SDF light never gets here)
}
// sample_lights_pdf through the pieces above, start to end for one query: what the kernel does spread over its trips
// (the known-answer test of K2's code path, vpt_kat_kernels.hip.h)
VPT_DEV float lights_pdf_k2(const DScene& sc, f3 position, f3 direction, int maxiter, const lane_stack& stk) {
  float sum = 0;
  for (int l = 0; l < sc.num_lights; l++) {
    float4 r6 = sc.light_rec[8 * l + 6], r7 = sc.light_rec[8 * l + 7];
    int    kind = __float_as_int(r7.w) & 255;
    if (kind == VPT_LIGHT_SDF) {
      const vpt_light& light = sc.lights[l];
      float lt = VPT_RAY_EPS;
      int   lit = 0;
      while (light_march_step(sc, sc.sdfs[light.sdf], sc.light_cdf[light.cdf_offset + light.cdf_len - 1], position, direction, maxiter, lt, lit, sum)) {}
    } else sum += inline_light_pdf(sc, l, kind, r6, r7, position, direction, stk);
  }
  return sum * ((float)1 / (float)sc.num_lights);
}

template <int SH>
__global__ void __launch_bounds__(VPT_BLOCK, VPT_K2_WAVES) vpt_render_kernel(DScene sc, DParams pr, float4* __restrict__ image,
    int* __restrict__ hits, ulonglong2* __restrict__ rngs, int stack_cap, sched_cfg sched, unsigned* __restrict__ watchdog) {
  extern __shared__ int lds_stack[];
  lane_stack stk;   // binary-node stack: only the pdf walk of an emissive mesh with a real BVH uses it
  stk.base = lds_stack + threadIdx.x;
  stk.cap  = stack_cap;
  const unsigned long long wave_start = wall_clock64();
  // ---- the lane's current pixel: state slot, pixel coordinates, running state (registers) -------------
  int   slot = -1, px = 0, py = 0;
  f4    acc = mk4(0, 0, 0, 0);
  rng_t rng = {0, 0};
  unsigned age = 0;           // trips of the wave loop since the pixel was fetched: its cost for the next launch's order
  bool  queue_empty = false;  // wave-uniform
  const vpt_camera& cam = sc.cameras[pr.camera];
  const int nb = pr.bounces, maxiter = pr.spheretrace_maxiter;
  const bool mis = !pr.noimplicit_mis;

  // ---- path state --------------------------------------------------------------------------------
  int   mode = M_FETCH;
  f3    ro = mk3(0, 0, 0), rd = mk3(0, 0, 1);          // current ray (tmin = 1e-4, tmax = flt_max)
  float t = VPT_RAY_EPS, lt = VPT_RAY_EPS;             // scene march / light march distance
  int   it = 0, lit = 0, hit_instance = -1, hit_sdf = -1;
  f3    radiance = mk3(0, 0, 0), weight = mk3(1, 1, 1);
  float alpha  = 0;
  int   bounce = 0, sample = 0;
  // pending MIS evaluation: weight *= f / (0.5 pdf + 0.5 lights_pdf) is finished after the loop over the lights
  f3    mis_f = mk3(0, 0, 0);
  float mis_pdf = 0, lp_sum = 0;
  int   lp_light = 0;

  while (true) {
    // every wave reaches an exit: the state machine ends when all lanes are M_DONE; should a defect ever keep it from
    // getting there, the wave gives up after VPT_K2_WATCHDOG_TICKS of the 100 MHz clock and the launch reports it
    if (wall_clock64() - wave_start > VPT_K2_WATCHDOG_TICKS) {
      if (threadIdx.x == 0 && watchdog) atomicAdd(watchdog, 1u);
      break;
    }
    unsigned long long marching = __builtin_amdgcn_ballot_w64(mode == M_SCENE || mode == M_LIGHT);
    unsigned long long waiting  = __builtin_amdgcn_ballot_w64(mode == M_NEW || mode == M_HIT || mode == M_MISS || mode == M_LIGHTS || mode == M_FETCH);
    if ((marching | waiting) == 0) break;   // every lane M_DONE
    age++;

    if (marching != 0 && __popcll(waiting) < VPT_K2_SHADE_AT) {
      // ---- march steps ------------------------------------------------------------------------------
      if (__builtin_amdgcn_ballot_w64(mode == M_SCENE) != 0) {
        for (int k = 0; k < VPT_K2_STEPS; k++)
          if (mode == M_SCENE) mode = scene_march_step(sc, ro, rd, maxiter, t, it, hit_instance, hit_sdf);
      }
      // SDF-light marches: cheap steps (one analytic SDF), several per trip; lanes of one light at a time so that the
      // light's record is wave-uniform (scalar loads)
      unsigned long long lm = __builtin_amdgcn_ballot_w64(mode == M_LIGHT);
      while (lm != 0) {
        int  l    = __builtin_amdgcn_readlane(lp_light, __ffsll((long long)lm) - 1);   // the light of the first lane still to serve
        bool mine = mode == M_LIGHT && lp_light == l;
        if (mine) {
          const vpt_light& light = sc.lights[l];
          const vpt_sdf&   sdf   = sc.sdfs[light.sdf];
          float area = sc.light_cdf[light.cdf_offset + light.cdf_len - 1];
          for (int k = 0; k < 4 * VPT_K2_STEPS; k++)
            if (mode == M_LIGHT && !light_march_step(sc, sdf, area, ro, rd, maxiter, lt, lit, lp_sum)) mode = M_LIGHTS, lp_light++;
        }
        lm &= ~__builtin_amdgcn_ballot_w64(mine);
      }
      continue;
    }

    // ---- shading block: the lanes that wait for it ---------------------------------------------------
    bool finish = false, next_vertex = false;   // next_vertex: a path vertex was completed, the new ray is in (ro, rd)
    if (mode == M_MISS) {   // cpp:444-447 / 545
      if constexpr (SH == K_IMPLICIT) radiance = radiance + weight * eval_environment(sc, rd);
      finish = true;
    } else if (mode == M_HIT) {
      f3 position = ro + rd * t;
      f3 normal   = hit_instance != VPT_INVALID ? eval_sdf_normal_grid(sc, sc.vol_instances[hit_instance], position, t)
                                                : eval_sdf_normal_function(sc.sdfs[hit_sdf], position, t);
      if constexpr (SH == K_IMPLICIT_NORMAL) {   // cpp:538-562
        radiance = normal * 0.5f + 0.5f;
        alpha    = 1;
        finish   = true;
      } else {   // shade_implicit, cpp:449-532
        f3     outgoing = -rd;
        int    mat = hit_instance != VPT_INVALID ? sc.vol_instances[hit_instance].material : sc.sdfs[hit_sdf].material;
        mpoint m   = eval_material_plain(sc, mat);
        if (m.opacity < 1 && rand1f(rng) >= m.opacity) {
          ro = position + rd * 1e-2f;   // bounce -= 1; continue
          t = VPT_RAY_EPS, it = 0, mode = M_SCENE;
        } else {
          radiance = radiance + weight * eval_emission(m.emission, normal, outgoing);
          f3 incoming = mk3(0, 0, 0);
          if (!is_delta(m)) {   // RNG draw order: the reference's right-to-left argument evaluation
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
            if (is_zero3(incoming)) finish = true;
            else {
              mis_f   = eval_bsdfcos(m, normal, outgoing, incoming);
              mis_pdf = sample_bsdfcos_pdf(m, normal, outgoing, incoming);
              ro = position, rd = incoming;
              if (mis) lp_sum = 0, lp_light = 0, mode = M_LIGHTS;   // the loop over the lights, below
              else {
                weight = weight * (mis_f / mis_pdf);
                mode = M_SCENE, next_vertex = true;
              }
            }
          } else {
            float rnl = rand1f(rng);
            incoming  = sample_delta(m, normal, outgoing, rnl);
            weight    = weight * (eval_delta(m, normal, outgoing, incoming) / sample_delta_pdf(m, normal, outgoing, incoming));
            ro = position, rd = incoming, mode = M_SCENE, next_vertex = true;
          }
        }
      }
    }
    if constexpr (SH == K_IMPLICIT) {
      if (mode == M_LIGHTS) {   // sample_lights_pdf's loop over the lights, resumable (cpp:353-421)
        while (lp_light < sc.num_lights) {
          float4 r6 = sc.light_rec[8 * lp_light + 6], r7 = sc.light_rec[8 * lp_light + 7];
          int    kind = __float_as_int(r7.w) & 255;
          if (kind == VPT_LIGHT_SDF) {   // needs a march: hand over
            lt = VPT_RAY_EPS, lit = 0, mode = M_LIGHT;
            break;
          }
          lp_sum += inline_light_pdf(sc, lp_light, kind, r6, r7, ro, rd, stk);
          lp_light++;
        }
        if (mode == M_LIGHTS) {   // all lights visited: finish the MIS weight (cpp:505-509)
          float lights_pdf = lp_sum * ((float)1 / (float)sc.num_lights);
          weight = weight * (mis_f / (0.5f * mis_pdf + 0.5f * lights_pdf));
          mode = M_SCENE, next_vertex = true;
        }
      }
      if (next_vertex) {
        t = VPT_RAY_EPS, it = 0;
        if (!survive(weight, bounce, rng)) finish = true;   // cpp:522-529
        bounce++;
        if (bounce >= nb) finish = true;
      }
    }
    if (finish) {   // cpp:1087-1089
      f4 rad = mk4(radiance.x, radiance.y, radiance.z, alpha);
      if (!(isfinite(rad.x) && isfinite(rad.y) && isfinite(rad.z) && isfinite(rad.w))) rad = mk4(0, 0, 0, 0);
      acc = acc + rad;
      sample++;
      mode = M_NEW;
      if (sample == pr.nsamples) {   // the pixel is done: its state goes back to HBM, the lane takes another pixel
        image[slot] = make_float4(acc.x, acc.y, acc.z, acc.w);
        hits[slot] += pr.nsamples;
        ulonglong2 r_out;
        r_out.x = rng.state, r_out.y = rng.inc;
        rngs[slot] = r_out;
        if (sched.cost) sched.cost[slot] = age;
        mode = M_FETCH;
      }
    }
    // ---- pixel queue: one atomic per batch of fetching lanes; slots that hold no pixel (padding of the last tiles) are skipped
    while (true) {
      unsigned long long need = __builtin_amdgcn_ballot_w64(mode == M_FETCH);
      if (need == 0) break;
      if (queue_empty) {
        if (mode == M_FETCH) mode = M_DONE;
        break;
      }
      int n = __popcll(need), base = 0;
      if ((int)threadIdx.x == __ffsll((long long)need) - 1) base = atomicAdd(sched.next, n);
      base = __builtin_amdgcn_readlane(base, __ffsll((long long)need) - 1);
      queue_empty = base + n >= sched.total;
      if (mode == M_FETCH) {
        int idx = base + __popcll(need & ((1ull << threadIdx.x) - 1));
        if (idx >= sched.total) mode = M_DONE;
        else {
          slot = sched.order ? sched.order[idx] : idx;
          if (slot_to_pixel(pr, slot, px, py)) {
            float4     acc_in = image[slot];
            ulonglong2 r_in   = rngs[slot];
            acc = mk4(acc_in.x, acc_in.y, acc_in.z, acc_in.w), rng.state = r_in.x, rng.inc = r_in.y;
            sample = 0, age = 0, mode = M_NEW;
          }
        }
      }
    }
    if (mode == M_NEW) {
      {
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
        ray_t ray = eval_camera(cam, mk2(u, v), lens);
        ro = ray.o, rd = ray.d;
        radiance = mk3(0, 0, 0), weight = mk3(1, 1, 1);
        alpha = (SH == K_IMPLICIT) ? 1.0f : 0.0f;
        bounce = 0, t = VPT_RAY_EPS, it = 0, mode = M_SCENE;
        if (SH == K_IMPLICIT && nb <= 0) mode = M_MISS, weight = mk3(0, 0, 0);   // no bounce allowed: the reference's loop body never runs (radiance 0, alpha 1)
      }
    }
  }

}
