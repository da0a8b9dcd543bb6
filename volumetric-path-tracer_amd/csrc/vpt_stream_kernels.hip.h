// vpt_stream_kernels.hip.h — K1 as a STREAM of two kernels over compact ray queues.
//
// Why (measured): in a one-kernel-per-launch design a lane carries the whole path
// state (~200 VGPRs => 2-3 waves/SIMD) through the BVH walk, which is a chain of dependent
// L2/Infinity-Cache fetches; rocprofv3 showed waves parked on memory ~47 % of their cycles and only
// ~10 of 64 lanes active per VALU instruction.  Splitting the trip "one path vertex" at its natural
// seam gives each half the shape the hardware wants:
//
//   k_trace   one lane = one ray of the queue.  Only the ray, the closest hit and the traversal
//             cursor live in registers (lean => 4+ waves/SIMD hide the fetch latency); while-while
//             traversal over 64-byte wide nodes with (ref, t0) stacks in LDS; the near child stays in a
//             register (no LDS round trip on the critical path).  Reads 32 B, writes 20 B per ray.
//   k_shade   one lane = one path whose query finished: consumes the hit exactly as the reference's
//             loop body does (medium sampling, surface / volume event, MIS with the light-pdf walk,
//             russian roulette), finishes and REGENERATES paths (next sample of the same pixel), and
//             appends the next ray to the other queue with one wave-aggregated atomic.
//
// Queues are COMPACT: a pixel leaves them only when all its samples are done, so waves stay full
// while the image's easy pixels (sky: 1 vertex/sample) finish many iterations before the hard ones
// (glass / skin / smoke: 5+ vertices/sample).  Path state is SoA in HBM, ~280 B of traffic per path
// vertex (<6 % of HBM bandwidth at 500 Msamples/s).  Per-pixel work is still done strictly in the
// reference's order by one lane at a time, so results do not depend on the schedule: this pipeline
// is bit-identical to the single-kernel version (tests/test_gpu_parity.py).
#pragma once
#include "vpt_mesh_kernel.hip.h"

struct DPaths {
  float4* ray_o;    // o.xyz, bounce (int bits)
  float4* ray_d;    // d.xyz, flags (int bits): bit0 = inside a medium
  float4* weight;   // weight.xyz, alpha
  float4* rad;      // radiance.xyz, sample index (int bits)
  float4* med0;     // density.xyz, anisotropy g
  float4* med1;     // scattering.xyz, emission.x
  float2* med2;     // emission.yz
  float4* hit;      // instance, element (int bits), uv.xy     (instance < 0: miss)
  float*  hit_t;    // distance
  int*    queue[2]; // slot indices of paths with a ray in flight
  int*    count;    // count[0], count[1]
};

#define VPT_FLAG_MEDIUM 1

// ---- wave-aggregated append to a queue ------------------------------------------------------------
VPT_DEV void queue_append(int* queue, int* count, bool emit, int slot) {
  unsigned long long mask = __ballot(emit);
  if (mask == 0) return;
  int lane   = __lane_id();
  int leader = __ffsll((long long)mask) - 1;
  int base   = 0;
  if (lane == leader) base = atomicAdd(count, __popcll(mask));
  base = __shfl(base, leader);
  if (emit) queue[base + __popcll(mask & ((1ull << lane) - 1ull))] = slot;
}

// ---- per-path working set of the shade kernel ---------------------------------------------------------
struct path_t {
  f3    ray_o, ray_d, weight, radiance;
  float alpha;
  int   bounce, sample;
  bool  in_medium;
  f3    med_density, med_scattering, med_emission;
  float med_g;
};

// camera ray of the pixel's next sample, yocto_pathtrace.cpp:1059-1068 / 1081-1086
VPT_DEV void generate(const DScene& sc, const DParams& pr, int px, int py, rng_t& rng, path_t& p) {
  const vpt_camera& cam = sc.cameras[pr.camera];
  float u, v;
  if (pr.preview) {
    u = (px + 0.5f) / pr.width, v = (py + 0.5f) / pr.height;
  } else {
    u = (px + rand1f(rng)) / pr.width;
    v = (py + rand1f(rng)) / pr.height;
  }
  f2 lens;
  lens.x  = rand1f(rng);
  lens.y  = rand1f(rng);
  ray_t r = eval_camera(cam, mk2(u, v), lens);
  p.ray_o = r.o, p.ray_d = r.d;
  p.radiance = mk3(0, 0, 0), p.weight = mk3(1, 1, 1);
  p.alpha = 0, p.bounce = 0, p.in_medium = false;
}

// Accumulate finished paths and start the pixel's next samples until one of them has a ray to trace
// (always the first, unless the bounce limit is 0) or the pixel is done.  Returns true if a ray is ready.
template <int SH>
VPT_DEV bool finish_and_regenerate(const DScene& sc, const DParams& pr, int slot, int px, int py, bool finished,
    rng_t& rng, path_t& p, float4* image, int* hits, int nb) {
  bool have_acc = false;
  f4   acc      = mk4(0, 0, 0, 0);
  while (true) {
    if (finished) {   // yocto_pathtrace.cpp:1087-1089
      if (!have_acc) {
        float4 a = image[slot];
        acc = mk4(a.x, a.y, a.z, a.w), have_acc = true;
      }
      f4 rad = mk4(p.radiance.x, p.radiance.y, p.radiance.z, p.alpha);
      if (!(isfinite(rad.x) && isfinite(rad.y) && isfinite(rad.z) && isfinite(rad.w))) rad = mk4(0, 0, 0, 0);
      acc = acc + rad;
      p.sample++;
    }
    bool more = p.sample < pr.nsamples;
    if (more) {
      generate(sc, pr, px, py, rng, p);
      finished = SH != K_DEBUG && p.bounce >= nb;   // zero-bounce corner: the path ends before its first query
      if (finished) continue;
    }
    if (have_acc) image[slot] = make_float4(acc.x, acc.y, acc.z, acc.w);
    if (!more) hits[slot] += pr.nsamples;
    return more;
  }
}

VPT_DEV void store_path(const DPaths& P, int slot, const path_t& p, bool store_medium) {
  P.ray_o[slot]  = make_float4(p.ray_o.x, p.ray_o.y, p.ray_o.z, __int_as_float(p.bounce));
  P.ray_d[slot]  = make_float4(p.ray_d.x, p.ray_d.y, p.ray_d.z, __int_as_float(p.in_medium ? VPT_FLAG_MEDIUM : 0));
  P.weight[slot] = make_float4(p.weight.x, p.weight.y, p.weight.z, p.alpha);
  P.rad[slot]    = make_float4(p.radiance.x, p.radiance.y, p.radiance.z, __int_as_float(p.sample));
  if (store_medium) {
    P.med0[slot] = make_float4(p.med_density.x, p.med_density.y, p.med_density.z, p.med_g);
    P.med1[slot] = make_float4(p.med_scattering.x, p.med_scattering.y, p.med_scattering.z, p.med_emission.x);
    P.med2[slot] = make_float2(p.med_emission.y, p.med_emission.z);
  }
}

// ---- k_begin: first camera ray of every pixel of this rank ------------------------------------------
template <int SH>
__global__ void __launch_bounds__(VPT_BLOCK) vpt_stream_begin(DScene sc, DParams pr, DPaths P, float4* __restrict__ image,
    int* __restrict__ hits, ulonglong2* __restrict__ rngs) {
  int  slot = blockIdx.x * VPT_BLOCK + threadIdx.x;
  int  px = 0, py = 0;
  bool valid = slot < pr.nslots && slot_to_pixel(pr, slot, px, py);
  bool emit  = false;
  if (valid) {
    ulonglong2 r = rngs[slot];
    rng_t  rng = {r.x, r.y};
    path_t p;
    p.sample = 0, p.alpha = 0, p.bounce = 0, p.in_medium = false;
    p.med_density = p.med_scattering = p.med_emission = mk3(0, 0, 0), p.med_g = 0;
    p.ray_o = p.ray_d = p.radiance = mk3(0, 0, 0), p.weight = mk3(1, 1, 1);
    const int nb = (SH == K_EYELIGHT) ? max(pr.bounces, 4) : pr.bounces;
    emit = finish_and_regenerate<SH>(sc, pr, slot, px, py, false, rng, p, image, hits, nb);
    if (emit) store_path(P, slot, p, true);
    r.x = rng.state, r.y = rng.inc;
    rngs[slot] = r;
  }
  queue_append(P.queue[0], P.count + 0, emit, slot);
}

// ---- k_trace: one BVH query per queue entry ------------------------------------------------------------
template <bool SPILL>
__global__ void __launch_bounds__(VPT_BLOCK, 4) vpt_stream_trace(DScene sc, DPaths P, int q, stack_cfg stack) {
  extern __shared__ int lds_stack[];
  const lane_stack2<SPILL> stk = make_lane_stack<SPILL>(lds_stack, stack);
  if (blockIdx.x == 0 && threadIdx.x == 0) P.count[q ^ 1] = 0;   // the queue the next k_shade appends to
  int i = blockIdx.x * VPT_BLOCK + threadIdx.x;
  if (i >= P.count[q]) return;
  int    slot = P.queue[q][i];
  float4 o = P.ray_o[slot], d = P.ray_d[slot];
  hit_t  h = traverse(sc, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), -1, stk);
  P.hit[slot]   = make_float4(__int_as_float(h.hit ? h.instance : -1), __int_as_float(h.element), h.uv.x, h.uv.y);
  P.hit_t[slot] = h.distance;
}

VPT_DEV float lights_pdf(const DScene& sc, const DParams& pr, f3 position, f3 direction, const lane_stack& stk) {
  float pdf = 0.0f;   // sample_lights_pdf, yocto_pathtrace.cpp:353-421
  for (int l = 0; l < sc.num_lights; l++) {
    float4 r6 = sc.light_rec[8 * l + 6], r7 = sc.light_rec[8 * l + 7];
    int    kind = __float_as_int(r7.w) & 255;
    if (kind == VPT_LIGHT_SMALL_MESH) pdf += small_light_pdf(sc, l, r6, r7, position, direction);
    else if (kind == VPT_LIGHT_LARGE_MESH) pdf += general_light_pdf(sc, sc.lights[l], position, direction, stk);
    else pdf += other_light_pdf(sc, l, kind, r6, position, direction, pr.spheretrace_maxiter);
  }
  return pdf * ((float)1 / (float)sc.num_lights);
}

// ---- k_shade: consume the hit of every queue entry, emit the next ray ------------------------------------
template <int SH>
__global__ void __launch_bounds__(VPT_BLOCK) vpt_stream_shade(DScene sc, DParams pr, DPaths P, int q,
    float4* __restrict__ image, int* __restrict__ hits, ulonglong2* __restrict__ rngs, int stack_cap) {
  extern __shared__ int lds_stack[];
  lane_stack stk;   // only used by general_light_pdf
  stk.base = lds_stack + threadIdx.x;
  stk.cap  = stack_cap;
  int  i      = blockIdx.x * VPT_BLOCK + threadIdx.x;
  bool active = i < P.count[q];
  bool emit   = false;
  int  slot   = 0;
  if (active) {
    slot = P.queue[q][i];
    int px, py;
    slot_to_pixel(pr, slot, px, py);
    const int nb = (SH == K_EYELIGHT) ? max(pr.bounces, 4) : pr.bounces;
    // ---- load the path -------------------------------------------------------------------------------
    path_t p;
    float4 ro = P.ray_o[slot], rd = P.ray_d[slot], we = P.weight[slot], ra = P.rad[slot];
    p.ray_o = mk3(ro.x, ro.y, ro.z), p.bounce = __float_as_int(ro.w);
    p.ray_d = mk3(rd.x, rd.y, rd.z), p.in_medium = (__float_as_int(rd.w) & VPT_FLAG_MEDIUM) != 0;
    p.weight = mk3(we.x, we.y, we.z), p.alpha = we.w;
    p.radiance = mk3(ra.x, ra.y, ra.z), p.sample = __float_as_int(ra.w);
    p.med_density = p.med_scattering = p.med_emission = mk3(0, 0, 0), p.med_g = 0;
    bool medium_dirty = false;
    if constexpr (SH == K_VOLPATH) {
      if (p.in_medium) {
        float4 m0 = P.med0[slot], m1 = P.med1[slot];
        float2 m2 = P.med2[slot];
        p.med_density = mk3(m0.x, m0.y, m0.z), p.med_g = m0.w;
        p.med_scattering = mk3(m1.x, m1.y, m1.z), p.med_emission = mk3(m1.w, m2.x, m2.y);
      }
    }
    ulonglong2 r = rngs[slot];
    rng_t  rng = {r.x, r.y};
    float4 hh = P.hit[slot];
    int    h_inst = __float_as_int(hh.x), h_elem = __float_as_int(hh.y);
    f2     h_uv   = mk2(hh.z, hh.w);
    float  h_dist = P.hit_t[slot];

    // ---- the reference's loop body after intersect_bvh (yocto_pathtrace.cpp:580-683 and siblings) ------
    bool finish = false;
    if (h_inst < 0) {
      if constexpr (SH != K_DEBUG) p.radiance = p.radiance + p.weight * eval_environment(sc, p.ray_d);
      finish = true;
    } else if constexpr (SH == K_DEBUG) {   // shade_normal / texcoord / color, cpp:893-930
      const DInstance& inst = sc.instances[h_inst];
      if (pr.shader == VPT_SHADER_NORMAL) p.radiance = eval_shading_normal(sc, inst, h_elem, h_uv, -p.ray_d);
      else if (pr.shader == VPT_SHADER_TEXCOORD) {
        f2 t       = eval_texcoord(sc, inst, h_elem, h_uv);
        p.radiance = mk3(t.x, t.y, 0);
      } else p.radiance = eval_material(sc, inst, h_elem, h_uv).color;
      p.alpha = 1;
      finish  = true;
    } else {
      bool in_volume = false;
      if constexpr (SH == K_VOLPATH) {
        if (p.in_medium) {   // cpp:586-596 — rd is drawn before rl
          float rd_      = rand1f(rng);
          float rl       = rand1f(rng);
          float distance = sample_transmittance(p.med_density, h_dist, rl, rd_);
          p.weight = p.weight * (vexp3(-p.med_density * distance) / sample_transmittance_pdf(p.med_density, distance, h_dist));
          in_volume = distance < h_dist;
          h_dist    = distance;
        }
      }
      if (!in_volume) {
        const DInstance& inst = sc.instances[h_inst];
        f3     outgoing = -p.ray_d;
        f3     position = eval_position(sc, inst, h_elem, h_uv);
        f3     normal   = eval_shading_normal(sc, inst, h_elem, h_uv, outgoing);
        mpoint m        = eval_material(sc, inst, h_elem, h_uv);
        if (m.opacity < 1 && rand1f(rng) >= m.opacity) {
          p.ray_o = position + p.ray_d * 1e-2f;   // bounce -= 1; continue
        } else {
          if (p.bounce == 0) p.alpha = 1;
          p.radiance = p.radiance + p.weight * eval_emission(m.emission, normal, outgoing);
          f3 incoming = mk3(0, 0, 0);
          if constexpr (SH == K_EYELIGHT) {   // cpp:869-886
            incoming   = outgoing;
            p.radiance = p.radiance + p.weight * VPT_PI * eval_bsdfcos(m, normal, outgoing, incoming);
            if (!is_delta(m)) finish = true;
            else {
              incoming = sample_delta(m, normal, outgoing, rand1f(rng));
              if (is_zero3(incoming)) finish = true;
              else {
                p.weight = p.weight * (eval_delta(m, normal, outgoing, incoming) / sample_delta_pdf(m, normal, outgoing, incoming));
                if (is_zero3(p.weight) || !finite3(p.weight)) finish = true;
                else p.ray_o = position, p.ray_d = incoming;
              }
            }
            p.bounce++;
          } else if constexpr (SH == K_NAIVE) {   // cpp:802-828
            if (m.roughness != 0) {
              f2 rn;
              rn.x      = rand1f(rng);
              rn.y      = rand1f(rng);
              float rnl = rand1f(rng);
              incoming  = sample_bsdfcos(m, normal, outgoing, rnl, rn);
              if (is_zero3(incoming)) finish = true;
              else p.weight = p.weight * (eval_bsdfcos(m, normal, outgoing, incoming) / sample_bsdfcos_pdf(m, normal, outgoing, incoming));
            } else {
              incoming = sample_delta(m, normal, outgoing, rand1f(rng));
              if (is_zero3(incoming)) finish = true;
              else p.weight = p.weight * (eval_delta(m, normal, outgoing, incoming) / sample_delta_pdf(m, normal, outgoing, incoming));
            }
            if (!finish) {
              if (!survive(p.weight, p.bounce, rng)) finish = true;
              else p.ray_o = position, p.ray_d = incoming;
            }
            p.bounce++;
          } else {   // pathtrace / volpathtrace, cpp:619-651
            bool vol_boundary = SH == K_VOLPATH && is_volumetric_type(sc.materials[inst.material].type);
            if (!is_delta(m)) {
              if (rand1f(rng) < 0.5f) {
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
                f3    f   = eval_bsdfcos(m, normal, outgoing, incoming);
                float pdf = 0.5f * sample_bsdfcos_pdf(m, normal, outgoing, incoming) + 0.5f * lights_pdf(sc, pr, position, incoming, stk);
                p.weight  = p.weight * (f / pdf);
              }
            } else {
              float rnl = rand1f(rng);
              incoming  = sample_delta(m, normal, outgoing, rnl);
              p.weight  = p.weight * (eval_delta(m, normal, outgoing, incoming) / sample_delta_pdf(m, normal, outgoing, incoming));
            }
            if (!finish) {
              if (vol_boundary && dot(normal, outgoing) * dot(normal, incoming) < 0) {   // cpp:641-648
                if (!p.in_medium) {
                  p.in_medium   = true, medium_dirty = true;
                  p.med_density = m.density, p.med_scattering = m.scattering, p.med_emission = m.emission, p.med_g = m.scanisotropy;
                } else {
                  p.in_medium = false;
                }
              }
              p.ray_o = position, p.ray_d = incoming;
              if (!survive(p.weight, p.bounce, rng)) finish = true;
              p.bounce++;
            }
          }
        }
      } else if constexpr (SH == K_VOLPATH) {   // volume event, cpp:654-673
        f3 outgoing = -p.ray_d;
        f3 position = p.ray_o + p.ray_d * h_dist;
        p.radiance = p.radiance + p.weight * eval_emission(p.med_emission, position, outgoing);   // (sic) cpp:660
        f3 incoming;
        if (rand1f(rng) < 0.5f) {
          f2 rn;
          rn.x = rand1f(rng);
          rn.y = rand1f(rng);
          (void)rand1f(rng);   // rnl is drawn and ignored, cpp:665
          incoming = sample_phasefunction(p.med_g, outgoing, rn);
        } else {
          f2 ruv;
          ruv.x     = rand1f(rng);
          ruv.y     = rand1f(rng);
          float rel = rand1f(rng);
          float rl  = rand1f(rng);
          incoming  = sample_lights(sc, position, rl, rel, ruv);
        }
        f3    f   = p.med_density * p.med_scattering * eval_phasefunction(p.med_g, incoming, outgoing);
        float pdf = 0.5f * eval_phasefunction(p.med_g, outgoing, incoming) + 0.5f * lights_pdf(sc, pr, position, incoming, stk);
        p.weight  = p.weight * (f / pdf);
        p.ray_o = position, p.ray_d = incoming;
        if (!survive(p.weight, p.bounce, rng)) finish = true;
        p.bounce++;
      }
    }
    // top of the reference's loop: the bounce limit ends the path before the next query
    if (!finish && SH != K_DEBUG && p.bounce >= nb) finish = true;

    emit = true;
    if (finish) emit = finish_and_regenerate<SH>(sc, pr, slot, px, py, true, rng, p, image, hits, nb);
    if (emit) store_path(P, slot, p, medium_dirty);
    r.x = rng.state, r.y = rng.inc;
    rngs[slot] = r;
  }
  queue_append(P.queue[q ^ 1], P.count + (q ^ 1), emit, slot);
}
