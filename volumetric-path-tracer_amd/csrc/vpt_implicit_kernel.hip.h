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
// Tried and not kept: handing pixels out dynamically (a fixed number of resident waves, a lane that has finished its
// pixel takes the next one from a queue, longest pixel first).  121 Msamples/s against 131 with one fixed pixel per lane
// and the longest-wave-first launch order: the launch is bound by its heaviest PIXELS — a floor pixel near the horizon
// marches ~700 dependent steps per sample, 128 samples in sequence, ~0.5 s of the 0.67 s launch — not by idle lanes,
// and neighbouring pixels in one wave march alike, which the queue gives up.  What shortens that chain is a cheaper
// step (SDF records with the per-evaluation constants folded, three additions for translation-only frames) and fewer
// steps: a march that can only end in a miss — the ray has left the ball around everything it could hit and recedes
// from it — ends at once (march_cannot_hit, vpt_scene.hip.h; the reference spends ~130 doublings of t on each).
// Also tried and not kept: skipping, per lane, the SDFs whose last value minus the distance marched since (they are
// 1-Lipschitz) still exceeds an upper bound of the scene minimum — exact, five of six SDFs skipped on a floor-grazing
// march — 166 against 174 Msamples/s: a wave evaluates an SDF as soon as ONE lane needs it, and the table costs more
// than the few whole-wave skips save (longest wave 388 against 413 ms, but the mean wave 169 against 161 ms).
// Round 3, also not kept: letting the SDF-light marches ride along in the scene-march rounds (a lane in M_LIGHT evaluates its
// light's SDF at its own point inside the loop over the analytic SDFs, with the instructions the scene lanes execute anyway).
// The light rounds (21 % of the wave time at 14 of 64 lanes) disappear, but every scene step then carries the light lanes'
// bookkeeping (exit tests, receding test, the second set of march registers): 301 against 297 Msamples/s on 06_gridsdf_full, 118
// against 124 on 07_sdfunction_synth (profiles/r03_k2_experiments.txt).
// Round 4: GROUP FORM of the scene march (kept).  A quarter of the scene-march rounds hold <= 16 marching rays and a fifth <= 8
// (profiles/r04_k2_lane_histogram.txt): the tail of a tile, when most of its pixels are finished or wait for the shading block.  Such a
// round runs with FOUR lanes per ray - ray k of the set on lanes 4k .. 4k+3, whoever owns them - each lane evaluating the SDFs j, j + 4, ...
// of the scene's list and the four partial minima meeting by DPP (eval_sdf_scene_group: first minimum wins, as in the reference); the ray's
// state travels by ds_bpermute once per round of VPT_K2_STEPS steps.  06_gridsdf_full 363 -> 388 Msamples/s, 07_sdfunction_synth (ten analytic
// SDFs) 141 -> 189; bit-identical (KAT tables, every whole-path case).  Re-measured with it: 4 waves per SIMD 356 / 180 (5 stay), SHADE_AT 16 /
// 28: 381 / 386, LIGHT_AT 2 / 8: 379 / 393 (but 169 on 07), STEPS 4 / 16: 372 / 395 (187 on 07): the round-3 settings stay.  Not kept: the
// radiance sum and the pixel coordinates parked in LDS as in K1 (scratch 260 -> 248 B per lane, 381 against 386 Msamples/s).
#pragma once
#include "vpt_mesh_kernel.hip.h"

#ifndef VPT_K2_WAVES
#define VPT_K2_WAVES 5       // waves per SIMD (round 3, with the settings below: 4: 324, 5: 341, 6: 284 Msamples/s on 06_gridsdf_full; round 2's kernel lost at 5)
#endif
#ifndef VPT_K2_SHADE_AT
#define VPT_K2_SHADE_AT 20   // lanes waiting for the shading block before the wave runs it (round 2: 8: -11 %, 16: -4 %, 24: best, 32: -5 %; round 3 with the light-march head inline: 16: -3 %, 20: +1 %, 32: -1 %)
#endif
#ifndef VPT_K2_WATCHDOG_TICKS
#define VPT_K2_WATCHDOG_TICKS 30000000000ull   // 300 s: two orders of magnitude above the longest wave of any test workload (the launch passes it as an argument; VPT_K2_WATCHDOG_MS overrides it for the tests of the error path)
#endif
#ifndef VPT_K2_LIGHT_INLINE
#define VPT_K2_LIGHT_INLINE 3   // (0: 299, 2: 313, 3: 333, 4: 315-320, 6: 334 Msamples/s on 06_gridsdf_full; 07_sdfunction_synth 121 -> 136) n > 0: the first n steps of an SDF light's pdf march run inside the shading block (55 % of the marches end within two steps: the ray recedes from the light), the rest as M_LIGHT trips; < 0: the whole march inline (round 2: 145 against 179 Msamples/s); 0: all of it as M_LIGHT trips
#endif
#ifndef VPT_K2_LIGHT_AT
#define VPT_K2_LIGHT_AT 4       // lanes in M_LIGHT before the wave spends a round on light-march steps while other lanes march the scene (round 2: 1: 178, 8: 204, 16: 204 Msamples/s; round 3, marches that survive their inline head: 06_gridsdf_full 1: 287, 2: 303, 4: 324, 8: 333; 07_sdfunction_synth 4: 150, 8: 136 - 4 is the better sum)
#endif
#ifndef VPT_K2_LIGHT_STEPS
#define VPT_K2_LIGHT_STEPS 16   // light-march steps per light round (32: 341, 16: 347 Msamples/s)
#endif
#ifndef VPT_K2_LIGHT_EXIT
#define VPT_K2_LIGHT_EXIT 0     // leave a light round as soon as none of its marches is alive
#endif
#ifndef VPT_K2_GROUP_MAX
#define VPT_K2_GROUP_MAX 16  // scene-march rounds with at most this many marching rays run four lanes per ray (0: never)
#endif
#ifndef VPT_K2_STEPS
#define VPT_K2_STEPS 12      // march steps between two looks at the wave's state (round 2: 2: -5 %, 4: -1.5 %, 8: best; end of round 4, 512 samples per launch: 8: 408 / 175.7, 12: 416.5 / 177.2, 16: 416.5 / 177.0, 24: 415.3 / 173.1 Msamples/s on 06_gridsdf_full / 07_sdfunction_synth)
#endif

// Diagnostic build (-DVPT_K2_STATS): where the lanes of a wave are, trip by trip (profiles/tools/k2_stats.py).  Never in the product build.
#ifdef VPT_K2_STATS
__device__ unsigned long long g_k2_stats[24];
enum { KS_TRIPS, KS_SCENE_ROUNDS, KS_SCENE_LANES, KS_LIGHT_ROUNDS, KS_LIGHT_LANES, KS_SHADE_ROUNDS, KS_SHADE_LANES, KS_DONE_LANES,
  KS_WAIT_LANES_AT_MARCH, KS_LIGHT_LANES_AT_SCENE, KS_SCENE_LANES_AT_SHADE, KS_CLK_SCENE, KS_CLK_LIGHT, KS_CLK_SHADE, KS_CLK_TOTAL,
  KS_SCENE_LE8, KS_SCENE_LE16, KS_SCENE_LE32, KS_SCENE_LANES_LE16, KS_SCENE_LANES_LE32, KS_COUNT };
#define K2_STAT(k, v) stats[k] += (unsigned long long)(v)
// shader-clock stamp that the scheduler cannot move (diagnostic build only)
#define K2_CLOCK(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) : : "memory")
#define K2_LAP(k, a, b) stats[k] += (b) - (a)
#else
#define K2_STAT(k, v)
#define K2_CLOCK(var)
#define K2_LAP(k, a, b)
#endif

enum { M_NEW = 0, M_SCENE = 1, M_HIT = 2, M_MISS = 3, M_LIGHT = 4, M_LIGHTS = 5, M_DONE = 6 };

// eval_sdf_scene (yocto_sdfs.cpp:7-26) by the four lanes of a group for ONE point: lane j evaluates the SDFs j, j + 4, ... of the list
// [voxel-grid instances][analytic SDFs] - the order the reference walks - and the four partial minima meet by DPP.  The reference keeps
// the FIRST minimum (`d < result`), so do the lanes (strict <, ascending) and the reduction (smaller value, on a tie the smaller index):
// the same (value, index) as the sequential loop, and every SDF value is computed by the same function from the same point.  All four
// lanes of a group must be active and hold the same p and t.
VPT_DEV sdf_hit eval_sdf_scene_group(const DScene& sc, const sdf_recs& recs, f3 p, float t, int j) {
  const int G = sc.num_vol_instances, S = G + sc.num_sdfs;
  float best = VPT_FLT_MAX;
  int   bc   = 0x7fffffff;
  for (int c = j; c < S; c += 4) {
    float d = c < G ? sdf_grid_world(sc, recs, c, p, t) : sdf_fn_world(recs, c - G, p);
    if (d < best) best = d, bc = c;
  }
  float ob = __int_as_float(quad_xor1(__float_as_int(best)));
  int   oc = quad_xor1(bc);
  bool  take = ob < best || (ob == best && oc < bc);
  best = take ? ob : best, bc = take ? oc : bc;
  ob = __int_as_float(quad_xor2(__float_as_int(best))), oc = quad_xor2(bc);
  take = ob < best || (ob == best && oc < bc);
  best = take ? ob : best, bc = take ? oc : bc;
  sdf_hit res = {best, bc < G ? bc : -1, (bc >= G && bc < S) ? bc - G : -1};
  return res;
}

// one step of spheretrace(scene, ray, maxiter) (yocto_pathtrace.cpp:289-307); returns the lane's next mode.  GROUP: the step of a ray that
// four lanes hold together (group_lane = the lane's place in its group): the scene SDF is evaluated by eval_sdf_scene_group, the rest by
// every lane alike
template <bool GROUP = false>
VPT_DEV int scene_march_step(const DScene& sc, const sdf_recs& recs, f3 ro, f3 rd, int maxiter, float& t, int& it, int& hit_instance, int& hit_sdf, int group_lane = 0) {
  if (!(it < maxiter && t < VPT_FLT_MAX)) return M_MISS;
  f3      p   = ro + rd * t;
  sdf_hit res = GROUP ? eval_sdf_scene_group(sc, recs, p, t, group_lane) : eval_sdf_scene(sc, recs, p, t);
  if (__builtin_fabsf(res.result) < (VPT_FLT_EPS * t)) {   // |x| < y and the ternary abs(x) < y agree for every x (they differ in the sign of a zero only)
    hit_instance = res.instance, hit_sdf = res.sdf;
    return M_HIT;
  }
  // further from every SDF than the radius of the ball around all of them: if the ray also recedes from that ball (and,
  // where the scene has unbounded SDFs - planes -, from each of those at a rate that outruns flt_eps * t), it misses
  if (res.result > sc.sdf_bound_r && sc.sdf_bound_r > 0 &&
      march_cannot_hit(mk3(sc.sdf_bound_cx, sc.sdf_bound_cy, sc.sdf_bound_cz), sc.sdf_bound_r, ro, p, rd)) {
    bool recede = true;
    if (sc.sdf_num_planes > 0) {   // flt_eps * t must stay below the planes' values (> r): t < 2e6 r as long as p is within 1e6 r of the ball
      f3 pc = p - mk3(sc.sdf_bound_cx, sc.sdf_bound_cy, sc.sdf_bound_cz);
      recede = dot(pc, pc) < 1e12f * sc.sdf_bound_r * sc.sdf_bound_r;
    }
    for (int idx = 0; sc.sdf_num_planes > 0 && idx < sc.num_sdfs; idx++) {
      const float4* rec = recs.fn + 6 * idx;
      if ((__float_as_int(rec[4].w) & 255) != VPT_SDF_PLANE) continue;
      // plane value y(s) = y0 + s * ny along the ray, y0 >= the scene minimum > r > 0: with ny >= 1e-3 it outgrows flt_eps * (t + s)
      float ny = transform_vector(unpack_frame(rec[0], rec[1], rec[2]), rd).y;
      if (!(ny >= 1e-3f)) recede = false;
    }
    if (recede) return M_MISS;
  }
  t += res.result, it++;
  return M_SCENE;
}
// one step of spheretrace(scene, ray, sdf, maxiter) (:267-286) inside sample_lights_pdf (:382-394) for SDF light `sdf`:
// on a hit adds the light's pdf term (normal at `position`, sic, :389) to `sum`; returns false when the march ended
VPT_DEV bool light_march_step(const sdf_recs& recs, int sdf, float area, f3 position, f3 direction, int maxiter, float& lt, int& lit, float& sum) {
  if (!(lit < maxiter && lt < VPT_FLT_MAX)) return false;
  f3    p   = position + direction * lt;
  float res = sdf_fn_world(recs, sdf, p);
  if (__builtin_fabsf(res) < (VPT_FLT_EPS * lt)) {
    f3 lnormal = eval_sdf_normal_function(recs, sdf, position, lt);
    sum += distance_squared(p, position) / (fabs_(dot(lnormal, direction)) * area);
    return false;
  }
  float4 bound = recs.fn[6 * sdf + 5];
  if (res > bound.w && bound.w > 0 && march_cannot_hit(xyz(bound), bound.w, position, p, direction)) return false;   // receding: a miss
  lt += res, lit++;
  return true;
}
// the lights of sample_lights_pdf that need no march (mesh lights, environments), one light
template <int FEAT = VPT_FEAT_ALL>
VPT_DEV float inline_light_pdf(const DScene& sc, int l, int kind, float4 r6, float4 r7, f3 position, f3 direction, const lane_stack& stk) {
  if ((FEAT & VPT_FEAT_SMALL_LIGHTS) && kind == VPT_LIGHT_SMALL_MESH) return small_light_pdf(sc, l, r6, r7, position, direction);
  if ((FEAT & VPT_FEAT_LARGE_LIGHTS) && kind == VPT_LIGHT_LARGE_MESH) return general_light_pdf(sc, sc.lights[l], position, direction, stk);
  return other_light_pdf<0>(sc, l, kind, r6, position, direction, 0);   // environments (an SDF light never gets here)
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
      while (light_march_step(scene_sdf_recs(sc), light.sdf, sc.light_cdf[light.cdf_offset + light.cdf_len - 1], position, direction, maxiter, lt, lit, sum)) {}
    } else sum += inline_light_pdf(sc, l, kind, r6, r7, position, direction, stk);
  }
  return sum * ((float)1 / (float)sc.num_lights);
}

template <int SH, int FEAT>
__device__ __forceinline__ void implicit_kernel_body(const DScene& sc, const DParams& pr, float4* __restrict__ image,
    int* __restrict__ hits, ulonglong2* __restrict__ rngs, int stack_cap, const sched_cfg& sched, unsigned* __restrict__ watchdog, unsigned long long watchdog_ticks) {
  extern __shared__ int lds_stack[];
  lane_stack stk;   // binary-node stack: only the pdf walk of an emissive mesh with a real BVH uses it
  stk.base = lds_stack + threadIdx.x;
  stk.cap  = stack_cap;
  // the SDF records, copied once into LDS behind the stack (they are read at every march step, the same address by all lanes)
  float4* lds_rec = (float4*)(lds_stack + (size_t)stack_cap * VPT_BLOCK);
  const int nfn4 = 6 * sc.num_sdfs, ngrid4 = 7 * sc.num_vol_instances;
  for (int i = threadIdx.x; i < nfn4; i += VPT_BLOCK) lds_rec[i] = sc.sdf_fn_rec[i];
  for (int i = threadIdx.x; i < ngrid4; i += VPT_BLOCK) lds_rec[nfn4 + i] = sc.sdf_grid_rec[i];
  __syncthreads();
  sdf_recs recs;
  recs.fn = lds_rec, recs.grid = lds_rec + nfn4;
  __shared__ unsigned long long s_wave_start;   // the start stamp, parked in LDS (clock_ticks, vpt_math.hip.h)
  if (threadIdx.x == 0) s_wave_start = clock_ticks(blockIdx.x);
  __syncthreads();
  const unsigned long long wave_start = s_wave_start;
  int trips = 0;
  const int wave = sched.order ? sched.order[blockIdx.x] : (int)blockIdx.x;

  int slot = wave * VPT_BLOCK + threadIdx.x;
  if (sched.lane_slot) slot = sched.lane_slot[slot];   // a split tile (vpt_capi.hip): this wave holds every 2^k-th pixel of it in its first lanes
  int px = 0, py = 0;
  const bool owner = slot >= 0 && slot < pr.nslots && slot_to_pixel(pr, slot, px, py);   // padding lanes own no pixel: they stay M_DONE
  if (__builtin_amdgcn_ballot_w64(owner) == 0) return;
  // the quorums of the state machine are sized for 64 pixels: a partly filled wave scales them with its pixels
  const int pixels = __popcll(__builtin_amdgcn_ballot_w64(owner));
  const int shade_at = max(1, (VPT_K2_SHADE_AT * pixels + 63) >> 6), light_at = max(1, (VPT_K2_LIGHT_AT * pixels + 63) >> 6);

  // ---- pixel state: one coalesced read, kept in registers for the whole launch ----------------
  f4    acc = mk4(0, 0, 0, 0);
  rng_t rng = {0, 0};
  if (owner) {
    float4     acc_in = image[slot];
    ulonglong2 r_in   = rngs[slot];
    acc = mk4(acc_in.x, acc_in.y, acc_in.z, acc_in.w), rng.state = r_in.x, rng.inc = r_in.y;
  }
  const vpt_camera& cam = sc.cameras[pr.camera];
  const int nb = pr.bounces, maxiter = pr.spheretrace_maxiter;
  const bool mis = !pr.noimplicit_mis;

  // ---- path state --------------------------------------------------------------------------------
  int   mode = owner ? M_NEW : M_DONE;
  f3    ro = mk3(0, 0, 0), rd = mk3(0, 0, 1);          // current ray (tmin = 1e-4, tmax = flt_max)
  float t = VPT_RAY_EPS;                               // distance / step count of the march the lane is in: the scene march or - a lane never runs both at once:
  int   it = 0, hit_id = -1;                            // what the scene march hit: a voxel-grid instance (>= 0) or an analytic SDF (~index)       // the scene march's t is consumed when its hit is shaded, before the light walk starts - an SDF light's pdf march
  f3    radiance = mk3(0, 0, 0), weight = mk3(1, 1, 1);
  float alpha  = 0;
  int   bounce = 0, sample = 0;
  // pending MIS evaluation: weight *= f / (0.5 pdf + 0.5 lights_pdf) is finished after the loop over the lights
  f3    mis_f = mk3(0, 0, 0);
  float mis_pdf = 0, lp_sum = 0;
  int   lp_light = 0;

#ifdef VPT_K2_STATS
  unsigned long long stats[KS_COUNT] = {};
#endif
  K2_CLOCK(cstart);
  bool gave_up = false;
  while (true) {
    // every wave reaches an exit: the state machine ends when all lanes are M_DONE; should a defect ever keep it from
    // getting there, the wave gives up after watchdog_ticks (VPT_K2_WATCHDOG_TICKS) of the 100 MHz clock and the launch reports it
    if (((trips++) & 255) == 0 && clock_ticks(trips) - wave_start > watchdog_ticks) {   // looked at on the first trip and on every 256th
      gave_up = true;
      break;
    }
    unsigned long long marching = __builtin_amdgcn_ballot_w64(mode == M_SCENE || mode == M_LIGHT);
    unsigned long long waiting  = __builtin_amdgcn_ballot_w64(mode == M_NEW || mode == M_HIT || mode == M_MISS || mode == M_LIGHTS);
    if ((marching | waiting) == 0) break;   // every lane M_DONE
    K2_STAT(KS_TRIPS, 1);
    K2_STAT(KS_DONE_LANES, 64 - __popcll(marching | waiting));

    if (marching != 0 && __popcll(waiting) < shade_at) {
      // ---- march steps ------------------------------------------------------------------------------
      K2_STAT(KS_WAIT_LANES_AT_MARCH, __popcll(waiting));
      K2_CLOCK(c0);
      const unsigned long long ms = __builtin_amdgcn_ballot_w64(mode == M_SCENE);
      if (VPT_K2_GROUP_MAX > 0 && sc.group_forms != 0 && ms != 0 && __popcll(ms) <= VPT_K2_GROUP_MAX) {
        // A small set of marching rays (a quarter of the scene rounds hold <= 16, a fifth <= 8: profiles/r04_k2_lane_histogram.txt): ray k of the
        // set on lanes 4k .. 4k+3 - whoever owns them: lanes whose pixel is finished, lanes that wait for the shading block - each lane
        // evaluating a quarter of the scene's SDFs per step (eval_sdf_scene_group); the ray's t sequence is the reference's, step for step
        K2_STAT(KS_SCENE_ROUNDS, 1);
        K2_STAT(KS_SCENE_LANES, __popcll(ms));
        K2_STAT(KS_SCENE_LE16, 1);
        const int  lane = threadIdx.x, gj = lane & 3;
        const bool mine = (ms >> lane) & 1;
        const int  rank = lanes_below(ms), nrays = __popcll(ms);
        int        gowner = quad_lane0(__builtin_amdgcn_ds_permute(mine ? rank << 4 : 4, lane));   // ray k's owner posts its lane number to lane 4k
        const bool gact = (lane >> 2) < nrays;
        if (!gact) gowner = lane;
        const f3 gro = pull(gowner, ro), grd = pull(gowner, rd);
        float    gt  = pull(gowner, t);
        int      git = pull(gowner, it), gmode = gact ? M_SCENE : M_DONE, ghi = -1, ghs = -1;
        for (int k = 0; k < VPT_K2_STEPS && __builtin_amdgcn_ballot_w64(gmode == M_SCENE) != 0; k++)
          if (gmode == M_SCENE) gmode = scene_march_step<true>(sc, recs, gro, grd, maxiter, gt, git, ghi, ghs, gj);
        const int   src = mine ? rank << 2 : 0;   // the owner of ray k reads lane 4k
        const int   nmode = pull(src, gmode), nit = pull(src, git), nhi = pull(src, ghi), nhs = pull(src, ghs);
        const float nt = pull(src, gt);
        if (mine) {
          mode = nmode, t = nt, it = nit;
          if (nmode == M_HIT) hit_id = nhi >= 0 ? nhi : ~nhs;
        }
      } else if (ms != 0) {
        K2_STAT(KS_SCENE_ROUNDS, 1);
        K2_STAT(KS_SCENE_LANES, __popcll(__builtin_amdgcn_ballot_w64(mode == M_SCENE)));
        K2_STAT(KS_SCENE_LE8, __popcll(__builtin_amdgcn_ballot_w64(mode == M_SCENE)) <= 8);
        K2_STAT(KS_SCENE_LE16, __popcll(__builtin_amdgcn_ballot_w64(mode == M_SCENE)) <= 16);
        K2_STAT(KS_SCENE_LE32, __popcll(__builtin_amdgcn_ballot_w64(mode == M_SCENE)) <= 32);
        K2_STAT(KS_SCENE_LANES_LE16, __popcll(__builtin_amdgcn_ballot_w64(mode == M_SCENE)) <= 16 ? __popcll(__builtin_amdgcn_ballot_w64(mode == M_SCENE)) : 0);
        K2_STAT(KS_SCENE_LANES_LE32, __popcll(__builtin_amdgcn_ballot_w64(mode == M_SCENE)) <= 32 ? __popcll(__builtin_amdgcn_ballot_w64(mode == M_SCENE)) : 0);
        K2_STAT(KS_LIGHT_LANES_AT_SCENE, __popcll(__builtin_amdgcn_ballot_w64(mode == M_LIGHT)));
        for (int k = 0; k < VPT_K2_STEPS; k++)
          if (mode == M_SCENE) {
            int hi = -1, hs = -1;
            mode = scene_march_step(sc, recs, ro, rd, maxiter, t, it, hi, hs);
            if (mode == M_HIT) hit_id = hi >= 0 ? hi : ~hs;
          }
      }
      K2_CLOCK(c1);
      K2_LAP(KS_CLK_SCENE, c0, c1);
      // SDF-light marches: cheap steps (one analytic SDF), several per trip; lanes of one light at a time so that the
      // light's record is wave-uniform (scalar loads)
      unsigned long long lm = __builtin_amdgcn_ballot_w64(mode == M_LIGHT);
      if (__popcll(lm) < light_at && __builtin_amdgcn_ballot_w64(mode == M_SCENE) != 0) lm = 0;   // too few: let them wait for company
      if (lm != 0) {
        K2_STAT(KS_LIGHT_ROUNDS, 1);
        K2_STAT(KS_LIGHT_LANES, __popcll(lm));
      }
      while (lm != 0) {
        int  l    = __builtin_amdgcn_readlane(lp_light, __ffsll((long long)lm) - 1);   // the light of the first lane still to serve
        bool mine = mode == M_LIGHT && lp_light == l;
        if (mine) {
          const vpt_light& light = sc.lights[l];
          float area = sc.light_cdf[light.cdf_offset + light.cdf_len - 1];
          for (int k = 0; k < VPT_K2_LIGHT_STEPS; k++) {
            if (mode == M_LIGHT && !light_march_step(recs, light.sdf, area, ro, rd, maxiter, t, it, lp_sum)) mode = M_LIGHTS, lp_light++;
            if (VPT_K2_LIGHT_EXIT && __builtin_amdgcn_ballot_w64(mode == M_LIGHT) == 0) break;   // every march of this light has ended
          }
        }
        lm &= ~__builtin_amdgcn_ballot_w64(mine);
      }
      K2_CLOCK(c2);
      K2_LAP(KS_CLK_LIGHT, c1, c2);
      continue;
    }
    K2_CLOCK(c3);

    // ---- shading block: the lanes that wait for it ---------------------------------------------------
    K2_STAT(KS_SHADE_ROUNDS, 1);
    K2_STAT(KS_SHADE_LANES, __popcll(waiting));
    K2_STAT(KS_SCENE_LANES_AT_SHADE, __popcll(marching));
    bool finish = false, next_vertex = false;   // next_vertex: a path vertex was completed, the new ray is in (ro, rd)
    if (mode == M_MISS) {   // cpp:444-447 / 545
      if constexpr (SH == K_IMPLICIT) radiance = radiance + weight * eval_environment(sc, rd);
      finish = true;
    } else if (mode == M_HIT) {
      f3 position = ro + rd * t;
      const int hit_instance = hit_id >= 0 ? hit_id : VPT_INVALID, hit_sdf = hit_id >= 0 ? VPT_INVALID : ~hit_id;
      f3 normal   = hit_instance != VPT_INVALID ? eval_sdf_normal_grid(sc, recs, hit_instance, position, t)
                                                : eval_sdf_normal_function(recs, hit_sdf, position, t);
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
              incoming  = sample_lights<FEAT>(sc, position, rl, rel, ruv);
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
          if (kind == VPT_LIGHT_SDF) {
            t = VPT_RAY_EPS, it = 0;
            if (VPT_K2_LIGHT_INLINE != 0) {   // the march's first steps (all of them if < 0) here, in the shading block
              const vpt_light& light = sc.lights[lp_light];
              float area = sc.light_cdf[light.cdf_offset + light.cdf_len - 1];
              bool  alive = true;
              for (int k = 0; alive && (VPT_K2_LIGHT_INLINE < 0 || k < VPT_K2_LIGHT_INLINE); k++) alive = light_march_step(recs, light.sdf, area, ro, rd, maxiter, t, it, lp_sum);
              if (!alive) {
                lp_light++;
                continue;
              }
            }
            mode = M_LIGHT;   // needs (the rest of) a march: hand over
            break;
          }
          lp_sum += inline_light_pdf<FEAT>(sc, lp_light, kind, r6, r7, ro, rd, stk);
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
      f4 rad = mk4(radiance.x, radiance.y, radiance.z, SH == K_IMPLICIT ? 1.0f : alpha);   // (shade_implicit's alpha is always 1: no register for it)
      if (!(isfinite(rad.x) && isfinite(rad.y) && isfinite(rad.z) && isfinite(rad.w))) rad = mk4(0, 0, 0, 0);
      acc = acc + rad;
      sample++;
      mode = M_NEW;
    }
    if (mode == M_NEW) {
      if (sample == pr.nsamples) mode = M_DONE;
      else {
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
    K2_CLOCK(c4);
    K2_LAP(KS_CLK_SHADE, c3, c4);
  }

  if (gave_up && threadIdx.x == 0 && watchdog) atomicAdd(watchdog, 1u);
  if (owner) {
    image[slot] = make_float4(acc.x, acc.y, acc.z, acc.w);
    hits[slot] += pr.nsamples;
    ulonglong2 r_out;
    r_out.x = rng.state, r_out.y = rng.inc;
    rngs[slot] = r_out;
  }
#ifdef VPT_K2_STATS
  K2_CLOCK(cend);
  K2_LAP(KS_CLK_TOTAL, cstart, cend);
  if (threadIdx.x == 0)
    for (int k = 0; k < KS_COUNT; k++) atomicAdd(&g_k2_stats[k], stats[k]);
#endif
  if (sched.cost && threadIdx.x == 0) {
    unsigned long long dt = clock_ticks(__float_as_int(acc.x)) - wave_start;   // after the last sample was accumulated
    sched.cost[wave] = dt < 0xffffffffull ? (unsigned)dt : 0xffffffffu;
#ifdef VPT_WAVE_TIMES
    if (wave < 65536) {   // diagnostic build: the launch's occupancy timeline (profiles/tools/wave_slots.py), as K1 records it
      g_vpt_wave_times[2 * wave] = wave_start, g_vpt_wave_times[2 * wave + 1] = wave_start + dt;
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_vpt_wave_hw[wave] = (xcc & 0xf) << 16 | (hw & 0xffff);
    }
#endif
  }
}

template <int SH, int FEAT>
__global__ void __launch_bounds__(VPT_BLOCK, VPT_K2_WAVES) vpt_render_kernel(DScene sc, DParams pr, float4* __restrict__ image,
    int* __restrict__ hits, ulonglong2* __restrict__ rngs, int stack_cap, sched_cfg sched, unsigned* __restrict__ watchdog, unsigned long long watchdog_ticks) {
  implicit_kernel_body<SH, FEAT>(sc, pr, image, hits, rngs, stack_cap, sched, watchdog, watchdog_ticks);
}
// The same kernel under another name: the short launch that measures per-wave costs when none are known yet (vpt_capi.hip),
// kept apart so that profiles of vpt_render_kernel only hold full launches (as vpt_mesh_pilot_kernel for K1).
template <int SH, int FEAT>
__global__ void __launch_bounds__(VPT_BLOCK, VPT_K2_WAVES) vpt_render_pilot_kernel(DScene sc, DParams pr, float4* __restrict__ image,
    int* __restrict__ hits, ulonglong2* __restrict__ rngs, int stack_cap, sched_cfg sched, unsigned* __restrict__ watchdog, unsigned long long watchdog_ticks) {
  implicit_kernel_body<SH, FEAT>(sc, pr, image, hits, rngs, stack_cap, sched, watchdog, watchdog_ticks);
}
