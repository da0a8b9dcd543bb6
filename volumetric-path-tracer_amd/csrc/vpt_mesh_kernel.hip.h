// vpt_mesh_kernel.hip.h — K1, the production kernel for the mesh shaders (volpathtrace,
// pathtrace, naive, eyelight, normal/texcoord/color).
//
// Design.  Measured on the first two versions: with one lane per pixel, plain
// SIMT control flow leaves ~7 of 64 lanes active per VALU instruction (rocprofv3
// SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU) — lanes of a wave want different code at any moment
// (box tests vs primitive tests vs entering an instance vs five kinds of shading), and the wave pays
// for all of it every trip.  This version makes the divergence explicit and schedules around it:
//
//  * every lane is a small STATE MACHINE (S_GEN ... S_DONE below) whose complete state lives in
//    registers + its LDS stack, including a RESUMABLE two-level BVH traversal;
//  * the wave runs a PHASE SCHEDULER: each turn it ballots the lanes' states, picks the most
//    populated one (wave-uniform, scalar branch) and executes only that phase's code for exactly the
//    lanes that are in it.  Lanes waiting for another phase simply wait; nothing is executed with a
//    handful of lanes unless nothing better exists.  Batching never reorders the work OF A LANE, so
//    results are independent of the schedule (a pixel's samples and every BVH query are still
//    processed strictly in the reference's order);
//  * one lane = one pixel for the whole launch, one wave64 = one 8x8 tile; the pixel's PCG32 stream,
//    radiance sum and hit count stay in registers, HBM state is touched once per launch;
//  * traversal over 64-byte "wide" nodes (both child boxes + refs in one 4 x dwordx4 fetch), leaf
//    primitives as contiguous 64-byte records, (ref, t0) stacks in LDS (entry-major, conflict-free).
//
// Exactness of the wide-node traversal.  The reference pops a node, tests its box against the
// current ray.tmax and only then looks at the children (yocto_bvh.cpp:728-750).  Here a child's box
// is tested when the parent is visited: t0 = max(max3(lo), tmin), t1 = min(min3(hi), tmax)*1.00000024f.
// tmax only shrinks, so a child failing now would also fail at its pop: not pushing it is exact.  A
// child passing now is pushed with t0; at its pop the reference's test with the smaller tmax' equals
// (t0 <= far*k) && (t0 <= tmax'*k) because x -> x*k is monotone; the first factor is known true, so
// the pop test `t0 <= tmax'*k` is the reference's test bit for bit.  Visit order (near child on top,
// yocto_bvh.cpp:744-750) and primitive order inside leaves are unchanged, so even exact ties in
// distance resolve as in the reference.
#pragma once
#include "vpt_kernels.hip.h"

struct lane_stack2 {
  int* base;   // &lds[threadIdx.x]; entry e: ref at base[(2e)*VPT_BLOCK], t0 at base[(2e+1)*VPT_BLOCK]
  int  cap;
  VPT_DEV void push(int& sp, int ref, float t0) const {
    if (sp < cap) base[(2 * sp) * VPT_BLOCK] = ref, base[(2 * sp + 1) * VPT_BLOCK] = __float_as_int(t0);
    sp++;
  }
  VPT_DEV void pop(int& sp, int& ref, float& t0) const {
    sp--;
    int s = sp < cap ? sp : cap - 1;
    ref = base[(2 * s) * VPT_BLOCK], t0 = __int_as_float(base[(2 * s + 1) * VPT_BLOCK]);
  }
};

#define VPT_BOX_K 1.00000024f

// intersect_bbox(ray, dinv, bbox) (yocto_geometry.h:858-868), also returning the entry distance
VPT_DEV bool box_pass(f3 bmin, f3 bmax, f3 o, f3 dinv, float tmin, float tmax, float& t0) {
  f3 it_min = (bmin - o) * dinv, it_max = (bmax - o) * dinv;
  f3 lo = vmin3(it_min, it_max), hi = vmax3(it_min, it_max);
  t0       = fmax_(max3(lo), tmin);
  float t1 = fmin_(min3(hi), tmax);
  t1 *= VPT_BOX_K;
  return t0 <= t1;
}
VPT_DEV int sign_bits(f3 dinv) { return (dinv.x < 0 ? 1 : 0) | (dinv.y < 0 ? 2 : 0) | (dinv.z < 0 ? 4 : 0); }

// pdf of the non-mesh lights (environment / sdf), one light: yocto_pathtrace.cpp:381-417
VPT_DEV float other_light_pdf(const DScene& sc, const vpt_light& light, f3 position, f3 direction, int maxiter) {
  const float* cdf = sc.light_cdf + light.cdf_offset;
  if (light.sdf != VPT_INVALID) {
    st_hit h = spheretrace_one(sc, position, direction, light.sdf, maxiter);
    if (!h.hit) return 0;
    f3 lposition = position + direction * h.dist;
    f3 lnormal   = eval_sdf_normal_function(sc.sdfs[h.sdf], position, h.dist);
    return distance_squared(lposition, position) / (fabs_(dot(lnormal, direction)) * cdf[light.cdf_len - 1]);
  }
  if (light.environment != VPT_INVALID) {
    const vpt_environment& env = sc.environments[light.environment];
    if (env.emission_tex == VPT_INVALID) return 1 / (4 * VPT_PI);
    int tw = sc.textures[env.emission_tex].width, th = sc.textures[env.emission_tex].height;
    f3 wl = transform_direction(load_frame(sc.env_inv + 3 * light.environment), direction);
    f2 tc = mk2(atan2f(wl.z, wl.x) / (2 * VPT_PI), acosf(clampf(wl.y, -1.0f, 1.0f)) / VPT_PI);
    if (tc.x < 0) tc.x += 1;
    int i = clampi((int)(tc.x * tw), 0, tw - 1), j = clampi((int)(tc.y * th), 0, th - 1);
    int idx = j * tw + i;
    float prob  = (idx == 0 ? cdf[0] : cdf[idx] - cdf[idx - 1]) / cdf[light.cdf_len - 1];
    float angle = (2 * VPT_PI / tw) * (VPT_PI / th) * sinf(VPT_PI * (j + 0.5f) / th);
    return prob / angle;
  }
  return 0;
}

// lane states.  Traversal: S_POP (next stack entry / housekeeping), S_PRIM (one leaf primitive),
// S_ENTER (transform the ray into an instance).  Shading: S_SHADE (medium distance sampling, picks
// surface or volume), S_SURF, S_VOL, S_MISS, S_LPOST (one mesh-light pdf hop finished), S_LIGHTS
// (continue sample_lights_pdf's loop over lights / finish the MIS weight).
enum { S_GEN = 0, S_POP, S_PRIM, S_ENTER, S_SHADE, S_SURF, S_VOL, S_MISS, S_LPOST, S_LIGHTS, S_DONE, S_COUNT };

#ifndef VPT_WAVES_PER_SIMD
#define VPT_WAVES_PER_SIMD 2
#endif

template <int SH>
__global__ void __launch_bounds__(VPT_BLOCK, VPT_WAVES_PER_SIMD) vpt_mesh_kernel(DScene sc, DParams pr,
    float4* __restrict__ image, int* __restrict__ hits, ulonglong2* __restrict__ rngs, int stack_cap) {
  extern __shared__ int lds_stack[];
  lane_stack2 stk;
  stk.base = lds_stack + threadIdx.x;
  stk.cap  = stack_cap;

  const int  slot = blockIdx.x * VPT_BLOCK + threadIdx.x;
  int        px = 0, py = 0;
  const bool valid = slot < pr.nslots && slot_to_pixel(pr, slot, px, py);   // padding lanes own no pixel

  f4    acc = mk4(0, 0, 0, 0);
  rng_t rng = {0, 0};
  if (valid) {
    float4 a = image[slot];
    acc      = mk4(a.x, a.y, a.z, a.w);
    ulonglong2 r = rngs[slot];
    rng.state = r.x, rng.inc = r.y;
  }
  const int      nb      = (SH == K_EYELIGHT) ? max(pr.bounces, 4) : pr.bounces;
  constexpr bool HAS_MIS = (SH == K_VOLPATH || SH == K_PATH);
  const float    tmin    = VPT_RAY_EPS;

  // ---- path state ----------------------------------------------------------------------------------
  f3    ray_o = mk3(0, 0, 0), ray_d = mk3(0, 0, 1);
  f3    radiance = mk3(0, 0, 0), weight = mk3(1, 1, 1);
  float alpha  = 0;
  int   bounce = 0, sample = 0;
  int   state  = (valid && pr.nsamples > 0) ? S_GEN : S_DONE;
  bool  in_medium = false;
  f3    med_density = mk3(0, 0, 0), med_scattering = mk3(0, 0, 0), med_emission = mk3(0, 0, 0);
  float med_g = 0;
  // ---- pending MIS evaluation: f / (0.5 pdf + 0.5 sum_lights), finished after the light-pdf walk ------
  f3    mis_f = mk3(0, 0, 0), lp_pos = mk3(0, 0, 0);
  float mis_pdf = 0, lp_sum = 0, lp_cur = 0;
  int   lp_light = 0, lp_hop = 0;
  bool  mis_toggle = false;
  // ---- resumable traversal state ----------------------------------------------------------------------
  f3    co = mk3(0, 0, 0), cd = mk3(0, 0, 1), cinv = mk3(0, 0, 0);   // ray in the current space
  int   csgn = 0, sp = 0, shape_base = -1, pend = 0, cur_inst = -1, enter = -1, leaf = 0;
  int   wn_base = 0, leaf_base = 0;        // current instance's offsets into shape_wnodes / leaf_prims
  bool  q_lpdf = false;                    // the running query is a light-pdf hop (origin lp_pos)
  float tmax = VPT_FLT_MAX;
  int   h_inst = -1, h_elem = -1;          // closest hit so far (h_inst < 0: none); distance == tmax
  f2    h_uv = mk2(0, 0);

  // start a BVH query with ray {origin, ray_d, 1e-4, flt_max}; instance < 0: whole scene
  auto begin_query = [&](f3 origin, int instance) {
    tmax = VPT_FLT_MAX, sp = 0, shape_base = -1, pend = 0, cur_inst = -1, h_inst = -1, h_elem = -1;
    co = origin, cd = ray_d, cinv = mk3(1 / ray_d.x, 1 / ray_d.y, 1 / ray_d.z), csgn = sign_bits(cinv);
    enter = instance;
    if (instance < 0) {
      float t0;
      if (sc.num_scene_nodes && box_pass(mk3(sc.scene_root_lo_x, sc.scene_root_lo_y, sc.scene_root_lo_z),
                                    mk3(sc.scene_root_hi_x, sc.scene_root_hi_y, sc.scene_root_hi_z), co, cinv, tmin, tmax, t0))
        stk.push(sp, sc.scene_root_ref, t0);
      state = S_POP;
    } else {
      state = S_ENTER;
    }
  };
  auto finish_path = [&]() {   // yocto_pathtrace.cpp:1087-1089
    f4 rad = mk4(radiance.x, radiance.y, radiance.z, alpha);
    if (!(isfinite(rad.x) && isfinite(rad.y) && isfinite(rad.z) && isfinite(rad.w))) rad = mk4(0, 0, 0, 0);
    acc = acc + rad;
    sample++;
    state = (sample == pr.nsamples) ? S_DONE : S_GEN;
  };
  auto next_vertex = [&]() {   // top of the reference's bounce loop
    if (SH != K_DEBUG && bounce >= nb) finish_path();
    else q_lpdf = false, begin_query(ray_o, -1);
  };

  while (true) {
    // ---- phase scheduler: run the most populated state (wave-uniform choice) ------------------------
    int pick = S_DONE, best = 0;
#pragma unroll
    for (int s = 0; s < S_DONE; s++) {
      int n = __popcll(__ballot(state == s));
      if (n > best) best = n, pick = s;
    }
    if (best == 0) break;   // every lane is S_DONE
    pick = __builtin_amdgcn_readfirstlane(pick);

    switch (pick) {
      // ======================================================================================= S_POP
      case S_POP:
        if (state == S_POP) {
          if (shape_base >= 0 && sp == shape_base) {   // instance exhausted: back to world space
            shape_base = -1;
            co = q_lpdf ? lp_pos : ray_o, cd = ray_d;
            cinv = mk3(1 / cd.x, 1 / cd.y, 1 / cd.z), csgn = sign_bits(cinv);
          }
          bool ready = true;
          if (shape_base < 0) {
            if (pend & 15) {   // next instance of the scene leaf being visited (yocto_bvh.cpp:852)
              enter = sc.scene_prims[pend >> 4], pend += 15, state = S_ENTER, ready = false;
            } else if (sp == 0) {   // query finished
              state = q_lpdf ? S_LPOST : (h_inst < 0 ? S_MISS : S_SHADE), ready = false;
            }
          }
          if (ready) {
            int   ref;
            float t0;
            stk.pop(sp, ref, t0);
            if (t0 <= tmax * VPT_BOX_K) {   // the reference's pop-time box test (see header)
              if (ref >= 0) {
                const float4* q = (shape_base < 0 ? sc.scene_wnodes : sc.shape_wnodes + 4 * (long long)wn_base) + 4 * (long long)ref;
                float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                int   ref0 = __float_as_int(q3.x), ref1 = __float_as_int(q3.y), axis = __float_as_int(q3.z);
                float ta, tb;
                bool  pa = box_pass(mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), co, cinv, tmin, tmax, ta);
                bool  pb = box_pass(mk3(q1.z, q1.w, q2.x), mk3(q2.y, q2.z, q2.w), co, cinv, tmin, tmax, tb);
                if ((csgn >> axis) & 1) {   // push child 0 then child 1: child 1 is visited first
                  if (pa) stk.push(sp, ref0, ta);
                  if (pb) stk.push(sp, ref1, tb);
                } else {
                  if (pb) stk.push(sp, ref1, tb);
                  if (pa) stk.push(sp, ref0, ta);
                }
              } else if (shape_base < 0) {
                pend = ~ref;   // scene leaf: its instances are entered one after another, in order
              } else {
                leaf = ~ref;   // shape leaf: primitives tested one per S_PRIM turn, in order
                if (leaf & 15) state = S_PRIM;
              }
            }
          }
        }
        break;
      // ====================================================================================== S_PRIM
      case S_PRIM:
        if (state == S_PRIM) {   // intersect_quad on the next primitive of the leaf (yocto_bvh.cpp:770-789)
          const float4* rec = sc.leaf_prims + 4 * ((long long)leaf_base + (leaf >> 4));
          float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
          float  dist;
          f2     uv;
          if (intersect_quad(co, cd, tmin, tmax, xyz(r0), xyz(r1), xyz(r2), xyz(r3), uv, dist))
            h_uv = uv, h_elem = __float_as_int(r0.w), h_inst = cur_inst, tmax = dist;
          leaf += 15;   // start + 1, count - 1
          if ((leaf & 15) == 0) state = S_POP;
        }
        break;
      // ===================================================================================== S_ENTER
      case S_ENTER:
        if (state == S_ENTER) {   // transform_ray(inverse(frame, true), ray) keeps tmin/tmax (yocto_bvh.cpp:853-855)
          const DInstance& inst = sc.instances[enter];
          frame inv = unpack_frame(inst.inv[0], inst.inv[1], inst.inv[2]);
          f3    wo  = q_lpdf ? lp_pos : ray_o;
          co = transform_point(inv, wo), cd = transform_vector(inv, ray_d);
          cinv = mk3(1 / cd.x, 1 / cd.y, 1 / cd.z), csgn = sign_bits(cinv);
          const DShape& sh = sc.shapes[inst.shape];
          cur_inst = enter, enter = -1, shape_base = sp;
          wn_base = sh.wnode_offset, leaf_base = sh.leaf_offset;
          float t0;
          if (sh.num_nodes && box_pass(ld3(sh.root_box), ld3(sh.root_box + 3), co, cinv, tmin, tmax, t0)) stk.push(sp, sh.root_ref, t0);
          state = S_POP;
        }
        break;
      // ======================================================================================= S_GEN
      case S_GEN:
        if (state == S_GEN) {   // yocto_pathtrace.cpp:1059-1068 / 1081-1086
          const vpt_camera& cam = sc.cameras[pr.camera];
          float u, v;
          if (pr.preview) {
            u = (px + 0.5f) / pr.width, v = (py + 0.5f) / pr.height;
          } else {
            u = (px + rand1f(rng)) / pr.width;
            v = (py + rand1f(rng)) / pr.height;
          }
          f2 lens;
          lens.x   = rand1f(rng);
          lens.y   = rand1f(rng);
          ray_t r  = eval_camera(cam, mk2(u, v), lens);
          ray_o = r.o, ray_d = r.d;
          radiance = mk3(0, 0, 0), weight = mk3(1, 1, 1);
          alpha = 0, bounce = 0, in_medium = false;
          next_vertex();
        }
        break;
      // ====================================================================================== S_MISS
      case S_MISS:
        if (state == S_MISS) {
          if constexpr (SH != K_DEBUG) radiance = radiance + weight * eval_environment(sc, ray_d);
          finish_path();
        }
        break;
      // ===================================================================================== S_SHADE
      case S_SHADE:
        if (state == S_SHADE) {
          state = S_SURF;
          if constexpr (SH == K_VOLPATH) {
            if (in_medium) {   // cpp:586-596 — rd is drawn before rl
              float rd       = rand1f(rng);
              float rl       = rand1f(rng);
              float distance = sample_transmittance(med_density, tmax, rl, rd);
              weight = weight * (vexp3(-med_density * distance) / sample_transmittance_pdf(med_density, distance, tmax));
              if (distance < tmax) state = S_VOL;
              tmax = distance;   // intersection.distance = distance
            }
          }
        }
        break;
      // ======================================================================================= S_VOL
      case S_VOL:
        if constexpr (SH == K_VOLPATH) {
          if (state == S_VOL) {   // volume event, cpp:654-673
            f3 outgoing = -ray_d;
            f3 position = ray_o + ray_d * tmax;
            radiance = radiance + weight * eval_emission(med_emission, position, outgoing);   // (sic) cpp:660
            f3 incoming;
            if (rand1f(rng) < 0.5f) {
              f2 rn;
              rn.x = rand1f(rng);
              rn.y = rand1f(rng);
              (void)rand1f(rng);   // rnl is drawn and ignored, cpp:665
              incoming = sample_phasefunction(med_g, outgoing, rn);
            } else {
              f2 ruv;
              ruv.x     = rand1f(rng);
              ruv.y     = rand1f(rng);
              float rel = rand1f(rng);
              float rl  = rand1f(rng);
              incoming  = sample_lights(sc, position, rl, rel, ruv);
            }
            mis_f      = med_density * med_scattering * eval_phasefunction(med_g, incoming, outgoing);
            mis_pdf    = eval_phasefunction(med_g, outgoing, incoming);
            mis_toggle = false;
            ray_o = position, ray_d = incoming;
            lp_sum = 0, lp_light = 0, state = S_LIGHTS;
          }
        }
        break;
      // ====================================================================================== S_SURF
      case S_SURF:
        if (state == S_SURF) {
          const DInstance& inst = sc.instances[h_inst];
          f3 outgoing = -ray_d;
          if constexpr (SH == K_DEBUG) {   // shade_normal / texcoord / color, cpp:893-930
            if (pr.shader == VPT_SHADER_NORMAL) radiance = eval_shading_normal(sc, inst, h_elem, h_uv, outgoing);
            else if (pr.shader == VPT_SHADER_TEXCOORD) {
              f2 t     = eval_texcoord(sc, inst, h_elem, h_uv);
              radiance = mk3(t.x, t.y, 0);
            } else radiance = eval_material(sc, inst, h_elem, h_uv).color;
            alpha = 1;
            finish_path();
          } else {
            f3     position = eval_position(sc, inst, h_elem, h_uv);
            f3     normal   = eval_shading_normal(sc, inst, h_elem, h_uv, outgoing);
            mpoint m        = eval_material(sc, inst, h_elem, h_uv);
            if (m.opacity < 1 && rand1f(rng) >= m.opacity) {
              ray_o = position + ray_d * 1e-2f;   // bounce -= 1; continue
              next_vertex();
            } else {
              if (bounce == 0) alpha = 1;
              radiance = radiance + weight * eval_emission(m.emission, normal, outgoing);
              f3   incoming = mk3(0, 0, 0);
              bool end_path = false;
              if constexpr (SH == K_EYELIGHT) {   // cpp:869-886
                incoming = outgoing;
                radiance = radiance + weight * VPT_PI * eval_bsdfcos(m, normal, outgoing, incoming);
                if (!is_delta(m)) end_path = true;
                else {
                  incoming = sample_delta(m, normal, outgoing, rand1f(rng));
                  if (is_zero3(incoming)) end_path = true;
                  else {
                    weight = weight * (eval_delta(m, normal, outgoing, incoming) / sample_delta_pdf(m, normal, outgoing, incoming));
                    if (is_zero3(weight) || !finite3(weight)) end_path = true;
                    else ray_o = position, ray_d = incoming;
                  }
                }
                bounce++;
                if (end_path) finish_path();
                else next_vertex();
              } else if constexpr (SH == K_NAIVE) {   // cpp:802-828
                if (m.roughness != 0) {
                  f2 rn;
                  rn.x      = rand1f(rng);
                  rn.y      = rand1f(rng);
                  float rnl = rand1f(rng);
                  incoming  = sample_bsdfcos(m, normal, outgoing, rnl, rn);
                  if (is_zero3(incoming)) end_path = true;
                  else weight = weight * (eval_bsdfcos(m, normal, outgoing, incoming) / sample_bsdfcos_pdf(m, normal, outgoing, incoming));
                } else {
                  incoming = sample_delta(m, normal, outgoing, rand1f(rng));
                  if (is_zero3(incoming)) end_path = true;
                  else weight = weight * (eval_delta(m, normal, outgoing, incoming) / sample_delta_pdf(m, normal, outgoing, incoming));
                }
                if (!end_path) {
                  if (!survive(weight, bounce, rng)) end_path = true;
                  else ray_o = position, ray_d = incoming;
                }
                bounce++;
                if (end_path) finish_path();
                else next_vertex();
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
                  if (is_zero3(incoming)) {
                    finish_path();
                  } else {
                    mis_f      = eval_bsdfcos(m, normal, outgoing, incoming);
                    mis_pdf    = sample_bsdfcos_pdf(m, normal, outgoing, incoming);
                    mis_toggle = vol_boundary && dot(normal, outgoing) * dot(normal, incoming) < 0;
                    if (mis_toggle && !in_medium)   // entering: the medium slot is free, fill it now
                      med_density = m.density, med_scattering = m.scattering, med_emission = m.emission, med_g = m.scanisotropy;
                    ray_o = position, ray_d = incoming;
                    lp_sum = 0, lp_light = 0, state = S_LIGHTS;
                  }
                } else {
                  float rnl = rand1f(rng);
                  incoming  = sample_delta(m, normal, outgoing, rnl);
                  weight    = weight * (eval_delta(m, normal, outgoing, incoming) / sample_delta_pdf(m, normal, outgoing, incoming));
                  if (vol_boundary && dot(normal, outgoing) * dot(normal, incoming) < 0) {   // cpp:641-648
                    if (!in_medium) {
                      in_medium   = true;
                      med_density = m.density, med_scattering = m.scattering, med_emission = m.emission, med_g = m.scanisotropy;
                    } else {
                      in_medium = false;
                    }
                  }
                  ray_o = position, ray_d = incoming;
                  bool alive = survive(weight, bounce, rng);
                  bounce++;
                  if (alive) next_vertex();
                  else finish_path();
                }
              }
            }
          }
        }
        break;
      // ===================================================================================== S_LPOST
      case S_LPOST:
        if constexpr (HAS_MIS) {
          if (state == S_LPOST) {   // one hop of the mesh-light pdf loop, cpp:363-378 (position = ray_o, direction = ray_d)
            bool light_done = true;
            if (h_inst >= 0) {
              const vpt_light& light = sc.lights[lp_light];
              const DInstance& inst  = sc.instances[h_inst];
              f3    lposition = eval_position(sc, inst, h_elem, h_uv);
              f3    lnormal   = eval_element_normal(sc, inst, h_elem);
              float area      = sc.light_cdf[light.cdf_offset + light.cdf_len - 1];
              lp_cur += distance_squared(lposition, ray_o) / (fabs_(dot(lnormal, ray_d)) * area);
              lp_pos = lposition + ray_d * 1e-3f;
              lp_hop++;
              light_done = lp_hop >= 100;
            }
            if (light_done) lp_sum += lp_cur, lp_light++, state = S_LIGHTS;
            else q_lpdf = true, begin_query(lp_pos, sc.lights[lp_light].instance);
          }
        }
        break;
      // ==================================================================================== S_LIGHTS
      case S_LIGHTS:
        if constexpr (HAS_MIS) {
          if (state == S_LIGHTS) {   // sample_lights_pdf's loop over lights, resumable (cpp:353-421)
            bool handed_over = false;
            while (lp_light < sc.num_lights) {
              const vpt_light& light = sc.lights[lp_light];
              if (light.instance != VPT_INVALID) {   // needs BVH hops: hand over to the traversal states
                lp_cur = 0, lp_hop = 0, lp_pos = ray_o, q_lpdf = true;
                begin_query(lp_pos, light.instance);
                handed_over = true;
                break;
              }
              lp_sum += other_light_pdf(sc, light, ray_o, ray_d, pr.spheretrace_maxiter);
              lp_light++;
            }
            if (!handed_over) {   // all lights visited: finish the MIS weight (cpp:630-634 / 668-671)
              float lights_pdf = lp_sum * ((float)1 / (float)sc.num_lights);
              weight = weight * (mis_f / (0.5f * mis_pdf + 0.5f * lights_pdf));
              if (mis_toggle) in_medium = !in_medium;   // cpp:642-648
              bool alive = survive(weight, bounce, rng);
              bounce++;
              if (alive) next_vertex();
              else finish_path();
            }
          }
        }
        break;
      default: break;
    }
  }

  if (valid) {
    image[slot] = make_float4(acc.x, acc.y, acc.z, acc.w);
    hits[slot] += pr.nsamples;
    ulonglong2 r_out;
    r_out.x = rng.state, r_out.y = rng.inc;
    rngs[slot] = r_out;
  }
}
