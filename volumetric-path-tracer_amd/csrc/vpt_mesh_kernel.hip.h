// vpt_mesh_kernel.hip.h — K1, the production kernel for the mesh shaders (volpathtrace,
// pathtrace, naive, eyelight, normal/texcoord/color).
//
// Design:
//  * one lane = one pixel for the whole launch, one wave64 = one 8x8 tile; the pixel's PCG32 stream,
//    radiance sum and hit count stay in registers, HBM state is touched once per launch;
//  * PER-LANE PATH REGENERATION: the outer loop's trip is "one path vertex"; a lane whose path ended
//    starts its pixel's next sample at once.  Every trip is [one BVH query for all lanes] -> [shading],
//    so the lanes of a wave stay phase-aligned although they sit at different bounces / samples;
//  * the BVH query code exists ONCE: the path's next vertex (two-level scene traversal) and, for
//    emissive meshes with a real BVH, the hops of the mesh-light pdf walk of sample_lights_pdf
//    (yocto_pathtrace.cpp:363-378) run through the same traversal as extra trips (state ST_LPDF).
//    Lights whose shape is a single leaf (<= 4 primitives, e.g. the area-light quads of the test
//    scenes) are walked inline in the shading phase instead (small_light_pdf): no trip, no stack;
//  * WHILE-WHILE TRAVERSAL over 128-byte QUAD NODES: one step tests the boxes of the four grandchildren of
//    a node of the reference's binary BVH (two levels per step); lanes walk internal nodes until each
//    holds a leaf (or is done), then all test leaf primitives together, then instance entries — unlike
//    work is never interleaved in one loop body;
//  * (ref, t0) stacks in LDS, entry-major, conflict-free (HBM overflow for BVHs whose worst case does not
//    fit); near / far planes chosen by the ray's sign and v_min3/v_max3 when no slab product can be NaN
//    (identical results), the reference's NaN-asymmetric ternary form otherwise;
//  * small scenes test the root boxes of all instances in lockstep at the start of a query (tmax = inf);
//  * GROUP FORMS (round 4): every lane of the wave goes through every query - a lane without a ray as a helper - and a node or leaf
//    phase that holds at most 16 rays runs on four lanes per ray (traverse(), below): the idle lanes of a query get work;
//  * waves are launched longest first from the durations the previous launch recorded (sched_cfg).
//
// Exactness of the quad-node traversal.  The reference pops a node, tests its box against the current
// ray.tmax and only then looks at the children (yocto_bvh.cpp:728-750).  Here a node's box is tested when
// its grandparent is visited: t0 = max(near planes, tmin), t1 = min(far planes, tmax) * 1.00000024f.
//  - tmax only shrinks, so a box failing now would also fail at its pop: not pushing it is exact;
//  - a box passing now is pushed with t0; at its pop the reference's test with the smaller tmax' equals
//    (t0 <= far*k) && (t0 <= tmax'*k) because x -> x*k is monotone; the first factor is known true, so the
//    pop test `t0 <= tmax'*k` is the reference's test bit for bit;
//  - the skipped middle level: a child's box lies inside its parent's and float -, *, min, max are monotone,
//    so t0(parent) <= t0(child): a child passing its own pop-time test implies the parent passed its test
//    (earlier, with a larger tmax); a parent that would have failed has no passing child;
//  - visit order: the reference pushes child 0 then child 1 when the ray is negative along the node's axis
//    (child 1 popped first) and repeats that when the child is popped, so the grandchildren are visited
//    [group by the node's axis][member by the child's axis]; primitive order inside leaves is unchanged (a
//    lane that reached a leaf does nothing until it has tested it).  Even exact ties in distance resolve
//    as in the reference.
#pragma once
#include "vpt_kernels.hip.h"

// (ref, t0) stack of one lane: the first `cap` entries live in LDS (entry-major: conflict-free), deeper
// ones in a per-launch HBM array (entry-major too: coalesced).  `cap` covers what traversals use in
// practice; the HBM part only makes the worst case (three pending siblings on every quad level) safe.
struct stack_cfg {
  int        cap;      // entries per lane in LDS
  int        spill;    // entries per lane in HBM
  int2*      mem;      // spill * lanes entries
  long long  lanes;    // lanes of the launch (= entry stride)
};
template <bool SPILL>
struct lane_stack2 {
  int*      base;   // &lds[threadIdx.x]; entry e: ref at base[(2e)*VPT_BLOCK], t0 at base[(2e+1)*VPT_BLOCK]
  int       cap;
  int2*     deep;   // &mem[global lane]; entry cap + e at deep[e * lanes]
  long long lanes;
  VPT_DEV void push(int& sp, int ref, float t0) const {
    if (!SPILL || sp < cap) base[(2 * sp) * VPT_BLOCK] = ref, base[(2 * sp + 1) * VPT_BLOCK] = __float_as_int(t0);
    else deep[(sp - cap) * lanes] = make_int2(ref, __float_as_int(t0));
    sp++;
  }
  VPT_DEV void put(int pos, int ref, float t0) const {   // LDS part only
    base[(2 * pos) * VPT_BLOCK] = ref, base[(2 * pos + 1) * VPT_BLOCK] = __float_as_int(t0);
  }
  static constexpr bool spills = SPILL;
  VPT_DEV void pop(int& sp, int& ref, float& t0) const {
    sp--;
    load(sp, ref, t0);
  }
  // entry `pos`, wherever it lives
  VPT_DEV void store(int pos, int ref, float t0) const {
    if (!SPILL || pos < cap) put(pos, ref, t0);
    else deep[(pos - cap) * lanes] = make_int2(ref, __float_as_int(t0));
  }
  VPT_DEV void load(int pos, int& ref, float& t0) const {
    if (!SPILL || pos < cap) ref = base[(2 * pos) * VPT_BLOCK], t0 = __int_as_float(base[(2 * pos + 1) * VPT_BLOCK]);
    else {
      int2 e = deep[(pos - cap) * lanes];
      ref = e.x, t0 = __int_as_float(e.y);
    }
  }
  // The HBM part is there for the worst case; a traversal rarely gets near it.  Wave-uniform shortcuts of the overflow variant: when no lane
  // that is here can reach the HBM part with its next pushes (pops), the step runs the unchecked LDS code of the plain variant
  VPT_DEV bool room_for(int sp, int entries) const { return !SPILL || __builtin_amdgcn_ballot_w64(sp + entries > cap) == 0; }
  VPT_DEV void load_lds(int pos, int& ref, float& t0) const { ref = base[(2 * pos) * VPT_BLOCK], t0 = __int_as_float(base[(2 * pos + 1) * VPT_BLOCK]); }
  // the stack of another lane of this wave (the helper lanes of a group form work on their ray owner's column)
  VPT_DEV lane_stack2 of_lane(int lane) const {
    lane_stack2 s = *this;
    s.base = base + (lane - (int)threadIdx.x), s.deep = deep + (lane - (int)threadIdx.x);
    return s;
  }
};

// ---- cross-lane moves (traverse(): the group forms).  DPP and ds_bpermute deliver 0 from a lane that is switched off, so
// these are only used where every lane they read is known to be active: inside group_nodes() / group_leaves(), which run only when the whole
// wave is in the call (exec = all ones), on groups of four lanes that are switched on and off together.
VPT_DEV int quad_lane0(int v) { return __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true); }   // quad_perm [0,0,0,0]
VPT_DEV int quad_xor1(int v) { return __builtin_amdgcn_mov_dpp(v, 0xb1, 0xf, 0xf, true); }    // quad_perm [1,0,3,2]
VPT_DEV int quad_xor2(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4e, 0xf, 0xf, true); }    // quad_perm [2,3,0,1]
VPT_DEV int quad_or(int v) {
  v |= quad_xor1(v);
  v |= quad_xor2(v);
  return v;
}
VPT_DEV int   pull(int lane, int v) { return __builtin_amdgcn_ds_bpermute(lane << 2, v); }
VPT_DEV float pull(int lane, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(lane << 2, __float_as_int(v))); }
VPT_DEV f3    pull(int lane, f3 v) { return mk3(pull(lane, v.x), pull(lane, v.y), pull(lane, v.z)); }
VPT_DEV int   lanes_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }
template <bool SPILL>
VPT_DEV lane_stack2<SPILL> make_lane_stack(int* lds, const stack_cfg& cfg) {
  lane_stack2<SPILL> stk;
  stk.base = lds + threadIdx.x, stk.cap = cfg.cap;
  stk.deep = cfg.mem + ((long long)blockIdx.x * VPT_BLOCK + threadIdx.x), stk.lanes = cfg.lanes;
  return stk;
}

#define VPT_BOX_K 1.00000024f

// intersect_bbox(ray, dinv, bbox) (yocto_geometry.h:858-868), also returning the entry distance
VPT_DEV bool slab_pass(f3 it_min, f3 it_max, float tmin, float tmax, float& t0);
VPT_DEV bool box_pass(f3 bmin, f3 bmax, f3 o, f3 dinv, float tmin, float tmax, float& t0) {
  return slab_pass((bmin - o) * dinv, (bmax - o) * dinv, tmin, tmax, t0);
}
VPT_DEV bool slab_pass(f3 it_min, f3 it_max, float tmin, float tmax, float& t0) {
  f3 lo = vmin3(it_min, it_max), hi = vmax3(it_min, it_max);
  t0       = fmax_(max3(lo), tmin);
  float t1 = fmin_(min3(hi), tmax);
  t1 *= VPT_BOX_K;
  return t0 <= t1;
}
VPT_DEV int sign_bits(f3 dinv) { return (dinv.x < 0 ? 1 : 0) | (dinv.y < 0 ? 2 : 0) | (dinv.z < 0 ? 4 : 0); }

// same test with 3-operand min/max; only valid when no product can be NaN (no zero in the direction):
// min/max of non-NaN values do not depend on how they are associated, so the result is the reference's
VPT_DEV bool box_pass_fast(f3 bmin, f3 bmax, f3 o, f3 dinv, float tmin, float tmax, float& t0) {
  f3 a = (bmin - o) * dinv, b = (bmax - o) * dinv;
  t0       = hw_max3(hw_max(hw_min(a.x, b.x), hw_min(a.y, b.y)), hw_min(a.z, b.z), tmin);
  float t1 = hw_min3(hw_min(hw_max(a.x, b.x), hw_max(a.y, b.y)), hw_max(a.z, b.z), tmax) * VPT_BOX_K;
  return t0 <= t1;
}
VPT_DEV bool box_test(bool slow, f3 bmin, f3 bmax, f3 o, f3 dinv, float tmin, float tmax, float& t0) {
  if (slow) return box_pass(bmin, bmax, o, dinv, tmin, tmax, t0);
  return box_pass_fast(bmin, bmax, o, dinv, tmin, tmax, t0);
}
VPT_DEV bool has_zero(f3 d) { return d.x == 0 || d.y == 0 || d.z == 0; }
// a ray whose slab products can be NaN (0 * inf): a zero direction component, or one so small that 1/d overflows
VPT_DEV bool nan_prone(f3 d, f3 dinv) {
  return has_zero(d) || !(__builtin_fabsf(dinv.x) < VPT_FLT_MAX && __builtin_fabsf(dinv.y) < VPT_FLT_MAX && __builtin_fabsf(dinv.z) < VPT_FLT_MAX);
}
// near-plane half of the fast test: with a finite non-zero 1/d, (lo - o)/d <= (hi - o)/d for d > 0 and >= for
// d < 0 (float - and * are monotone), so min/max of the two slab products per axis is a choice by the sign of d
VPT_DEV bool slab_pass_signed(float nx, float ny, float nz, float fx, float fy, float fz, float tmin, float tmax, float& t0) {
  t0       = hw_max3(hw_max(nx, ny), nz, tmin);
  float t1 = hw_min3(hw_min(fx, fy), fz, tmax) * VPT_BOX_K;
  return t0 <= t1;
}

// One BVH query.  only_instance < 0: intersect_bvh(bvh, scene, ray) (yocto_bvh.cpp:800-871);
// only_instance >= 0: intersect_bvh(bvh, scene, instance, ray) (:874-881).  Ray = {wo, wd, 1e-4, flt_max}.
//
// EVERY lane of the wave that is in the surrounding code calls this, with `active` = "this lane has a ray": a lane without one
// (its path ended, its pixel is finished, it owns no pixel) lends its registers and issue slots to the others' rays (the group
// forms, below).  The function is correct from any control flow; the group forms only run when the whole wave is in the call.
//
// Per-lane ("own") form.  The lane's next action is kept in `cur` (>= 0: quad node of the current level, VPT_NONE: nothing
// left at this level, other negatives: a leaf).  Three kinds of work, never interleaved in one loop body:
//   A  quad nodes: fetch, four box tests, descend into the first-visited passing child directly (it would be popped next
//      with the same tmax, so its pop test is a tautology) and push the others, last first;
//   B  shape leaf: primitive tests in order;
//   C  scene leaf / pending instances: transform the ray, test the instance's root box (the test the reference's shape-level
//      loop does first, yocto_bvh.cpp:728-733); instances that miss it are skipped without ever leaving world space.
//
// GROUP FORMS of A and B (round 4).  In the own form a node step runs with 9.8 of 64 lanes switched on and a primitive test with
// 12.2 (profiles/r01_v8_section_counters.txt): the lanes of a wave need different kinds of work at different times, and most node
// steps belong to the tail of a query, when a handful of long rays is left.  Whenever at most VPT_COOP_MAX (16) rays of the wave are
// in a phase, the phase runs on groups of FOUR lanes per ray - ray k of the set on lanes 4k .. 4k+3, whoever owns those lanes; the
// ray's state travels there by ds_bpermute and its results back - with the SAME loop shape as the own form (phase A until no ray of
// the set holds a node, then phase B once), so the wave takes the same number of steps, each a fraction of the work:
//   A  lane c of a group loads and tests grandchild c of the quad node (8 dwords and ~25 instructions instead of 29 dwords and
//      ~75); the reference's visit order is a 2-bit rank per lane computed from the three split axes and the ray's signs, the
//      four (pass, rank) pairs meet by DPP quad_perm, each lane stores its own candidate into the OWNER's LDS stack column
//      at the position its rank gives it, and the first-visited one becomes the group's next node;
//   B  the (<= 4) primitives of a leaf are tested one per lane against the tmax at leaf entry and reduced to the reference's
//      winner: the smallest t, on an exact tie the later primitive (intersect_triangle only rejects t > tmax,
//      yocto_geometry.h:786-819, and the leaf loop overwrites, yocto_bvh.cpp:770-789; a primitive's t does not depend on
//      tmax, only its acceptance does, so min-then-last is what the sequential loop leaves behind).  A NaN distance (a
//      degenerate triangle met by a ray with a zero component) makes every later comparison in the reference true: a group
//      that meets one, or holds a NaN tmax, reports nothing and its owner walks the leaf in the own form.
// Per-box and per-primitive arithmetic is the own form's, operand for operand: the bits are the same (tests: KAT intersect_*,
// test_intersect_is_bit_identical_on_edge_case_rays with dense and with sparse waves, every whole-path case).
// Measured (03_volume, 1280x533x256 spp): 254.0 -> 227.0 ms per launch.  A first version kept a ray in its group across phases until it
// needed C work ("sessions"): fewer transfers, but the rays outside a session waited for its longest member - 174 wave-level node
// steps per 64 samples against the own form's 132 - and it LOST (276 ms, profiles/r04_k1_group_forms.txt).
#define VPT_NONE (-2147483647 - 1)
#ifndef VPT_HOIST_MAX
#define VPT_HOIST_MAX 16   // scenes with at most this many instances test all root boxes at the start of a query
#endif
#ifndef VPT_COOP_MAX
#define VPT_COOP_MAX 16    // a phase with at most this many rays runs in its group form (four lanes per ray); 0: own forms only
#endif
#ifndef VPT_COOP_NODES
#define VPT_COOP_NODES VPT_COOP_MAX   // (experiments: the group form of one phase only)
#endif
#ifndef VPT_COOP_LEAVES
#define VPT_COOP_LEAVES VPT_COOP_MAX
#endif
#ifdef VPT_TRAVERSE_GUARD
__device__ unsigned g_vpt_guard_trips;   // diagnostic build: queries that were cut short after VPT_TRAVERSE_GUARD loop rounds (must stay 0)
#endif
template <bool COMPACT = false, class STK>
VPT_DEV hit_t traverse(const DScene& sc, bool active, f3 wo, f3 wd, int only_instance, const STK& stk) {
  constexpr int LS = COMPACT ? 3 : 4;   // float4 per leaf record
  hit_t r;
  r.instance = -1, r.element = -1, r.uv = mk2(0, 0), r.distance = 0, r.hit = false, r.prim = 0;   // (hit and distance are filled in at the end: instance >= 0, tmax)
  const float tmin = VPT_RAY_EPS;
  float tmax = VPT_FLT_MAX;
  const bool whole_wave = __builtin_amdgcn_ballot_w64(true) == ~0ull && sc.group_forms != 0;   // the group forms move data between lanes: all of them have to be here
  // (a lane without a ray holds whatever its last ray left in wo / wd: it must not steer the wave-wide choices of the reciprocal and slab forms)
  const f3   winv = rcp3_exact(active ? wd : mk3(1, 1, 1));
  bool       wslow = active && nan_prone(wd, winv);
  f3    co = wo, cd = wd, cinv = winv;
  int   csgn = sign_bits(winv);
  bool  slow = wslow;
  int   sp = 0, shape_base = -1, pend = 0, cur_inst = -1, cur = VPT_NONE;
  int   wnb = 0, leafb = 0;   // the current level's quad nodes start at sc.scene_wnodes[8 * wnb], its leaf records at sc.leaf_prims[4 * leafb]
  bool  done = !active;

  // pop entries of the current level until one passes the reference's pop-time box test
  auto pop_valid = [&]() {
    int base = shape_base >= 0 ? shape_base : 0;
    const bool lds_only = stk.room_for(sp, 0);   // sp only falls from here on
    while (sp > base) {
      int   ref;
      float t0;
      VPT_CNT(CNT_POP);
      sp--;
      if (lds_only) stk.load_lds(sp, ref, t0);
      else stk.load(sp, ref, t0);
      if (t0 <= tmax * VPT_BOX_K) return ref;
    }
    return VPT_NONE;
  };
  // Phase C.  Enter the pending instances of a scene leaf (pend = first slot << 4 | count) one after another,
  // from their 96-byte enter records (vpt_device.h), until one passes the root-box test the reference's
  // shape-level loop does first (yocto_bvh.cpp:728-733); instances that miss it never leave world space.
  // At scene level {cd, cinv, csgn, slow} always equal the world ray's, so an instance whose inverse frame
  // is {I, -o} only needs co: 1*d + 0*d' + 0*d'' == d bit for bit when no component is zero, and
  // ((1*o.x + 0*o.y) + 0*o.z) + t == o.x + t.  co itself is restored once, when the leaf is exhausted.
  // Returns the lane's next reference.
  unsigned reach = 0xffffffffu;   // bit s clear: the instance in slot s cannot pass its root-box test whatever tmax is (below)
  auto enter_pending = [&]() {
    while (pend & 15) {
      int slot = pend >> 4;
      pend += 15;   // first slot + 1, count - 1
      if (!((reach >> (slot & 31)) & 1)) continue;
      VPT_CNT(CNT_ENTER);
      const float4* e = sc.scene_enter + 6 * (long long)slot;
      float4 e2 = e[2], e3 = e[3], e4 = e[4], e5 = e[5];
      bool   general = !__float_as_int(e5.z) || wslow;
      if (general) {
        frame inv = unpack_frame(e[0], e[1], e2);
        co = transform_point(inv, wo), cd = transform_vector(inv, wd);
        cinv = rcp3_exact(cd), slow = nan_prone(cd, cinv) || tmax != tmax;   // (a NaN tmax: own_leaf)
      } else {
        co = mk3(wo.x + e2.y, wo.y + e2.z, wo.z + e2.w);
      }
      float t0;
      if (__float_as_int(e5.w) && box_test(slow, mk3(e3.x, e3.y, e3.z), mk3(e3.w, e4.x, e4.y), co, cinv, tmin, tmax, t0)) {
        if (general) csgn = sign_bits(cinv);
        cur_inst = __float_as_int(e5.y), shape_base = sp;
        wnb = __float_as_int(e4.w), leafb = __float_as_int(e5.x);
        return __float_as_int(e4.z);   // the instance's root: visited next with the same tmax
      }
      if (general) cd = wd, cinv = winv, slow = wslow;
    }
    co = wo;
    return pop_valid();
  };

  if (active) {
    if (only_instance < 0) {
      float t0;
      if (sc.num_scene_nodes && box_test(slow, mk3(sc.scene_root_lo_x, sc.scene_root_lo_y, sc.scene_root_lo_z),
                                    mk3(sc.scene_root_hi_x, sc.scene_root_hi_y, sc.scene_root_hi_z), co, cinv, tmin, tmax, t0))
        cur = sc.scene_root_ref;
      // Small scenes: every lane tests the root boxes of ALL instances now, in lockstep (the slot index is
      // wave-uniform: scalar record loads, no divergence), with tmax = inf.  A box missed with tmax = inf is missed
      // with any tmax (t1 = min(far, tmax) * k only shrinks), so phase C skips those instances without fetching
      // their records; the others are tested again there against the current tmax, as the reference does.
      // Entering instances one lane at a time is the most divergent part of a query (13 of 64 lanes active).
      if (sc.num_scene_prims <= VPT_HOIST_MAX && cur != VPT_NONE) {
        bool any_slow = __builtin_amdgcn_ballot_w64(wslow) != 0;
        reach = 0;
        for (int s = 0; s < sc.num_scene_prims; s++) {
          const float4* e = sc.scene_enter + 6 * s;
          float4 e2 = e[2], e3 = e[3], e4 = e[4], e5 = e[5];
          // the record is the same for every lane: branch on its flags as scalars, each arm complete in itself
          int  translation = __builtin_amdgcn_readfirstlane(__float_as_int(e5.z));
          int  has_root    = __builtin_amdgcn_readfirstlane(__float_as_int(e5.w));
          if (!has_root) continue;
          f3    blo = mk3(e3.x, e3.y, e3.z), bhi = mk3(e3.w, e4.x, e4.y);
          float t0;
          bool  pass;
          if (translation && !any_slow) {
            pass = box_pass_fast(blo, bhi, mk3(wo.x + e2.y, wo.y + e2.z, wo.z + e2.w), winv, tmin, VPT_FLT_MAX, t0);
          } else {
            frame inv  = unpack_frame(e[0], e[1], e2);
            f3    ld   = transform_vector(inv, wd);
            f3    linv = rcp3_exact(ld);
            pass = box_test(nan_prone(ld, linv), blo, bhi, transform_point(inv, wo), linv, tmin, VPT_FLT_MAX, t0);
          }
          if (pass) reach |= 1u << s;
        }
      }
      if (cur == VPT_NONE) done = true;   // the ray misses the scene's box
    } else {   // single-instance query (yocto_bvh.cpp:874-881)
      pend = (sc.slot_of_instance[only_instance] << 4) | 1;
      cur  = enter_pending();
      shape_base = 0;
      if (cur == VPT_NONE) done = true;
    }
  }

  // ---- own form of A: one quad-node step of this lane's ray ------------------------------------------------------------
  auto own_node_step = [&]() {
    VPT_CNT(CNT_NODE);
    const float4* q = sc.scene_wnodes + 8 * (long long)(wnb + cur);
    // rows of the node: lo.x lo.y lo.z hi.x hi.y hi.z (four children each).  The near plane of an axis is
    // the lo row for a positive direction, the hi row for a negative one: fetch them by the ray's signs
    int    ix = (csgn & 1) ? 3 : 0, iy = (csgn & 2) ? 4 : 1, iz = (csgn & 4) ? 5 : 2;
    float4 nx = q[ix], fx = q[3 - ix], ny = q[iy], fy = q[5 - iy], nz = q[iz], fz = q[7 - iz], qr = q[6];
    int    meta = __float_as_int(q[7].x);
    float4 anx = make_float4((nx.x - co.x) * cinv.x, (nx.y - co.x) * cinv.x, (nx.z - co.x) * cinv.x, (nx.w - co.x) * cinv.x);
    float4 afx = make_float4((fx.x - co.x) * cinv.x, (fx.y - co.x) * cinv.x, (fx.z - co.x) * cinv.x, (fx.w - co.x) * cinv.x);
    float4 any = make_float4((ny.x - co.y) * cinv.y, (ny.y - co.y) * cinv.y, (ny.z - co.y) * cinv.y, (ny.w - co.y) * cinv.y);
    float4 afy = make_float4((fy.x - co.y) * cinv.y, (fy.y - co.y) * cinv.y, (fy.z - co.y) * cinv.y, (fy.w - co.y) * cinv.y);
    float4 anz = make_float4((nz.x - co.z) * cinv.z, (nz.y - co.z) * cinv.z, (nz.z - co.z) * cinv.z, (nz.w - co.z) * cinv.z);
    float4 afz = make_float4((fz.x - co.z) * cinv.z, (fz.y - co.z) * cinv.z, (fz.z - co.z) * cinv.z, (fz.w - co.z) * cinv.z);
    float  t0, t1, t2, t3;
    bool   p0, p1, p2, p3;
    if (__builtin_amdgcn_ballot_w64(slow) != 0) {   // some lane may meet 0 * inf: the reference's NaN-asymmetric form for all
      // (lo - o) * inv and (hi - o) * inv are the same products, told apart again by the sign
      bool sx = csgn & 1, sy = csgn & 2, sz = csgn & 4;
      p0 = slab_pass(mk3(sx ? afx.x : anx.x, sy ? afy.x : any.x, sz ? afz.x : anz.x), mk3(sx ? anx.x : afx.x, sy ? any.x : afy.x, sz ? anz.x : afz.x), tmin, tmax, t0);
      p1 = slab_pass(mk3(sx ? afx.y : anx.y, sy ? afy.y : any.y, sz ? afz.y : anz.y), mk3(sx ? anx.y : afx.y, sy ? any.y : afy.y, sz ? anz.y : afz.y), tmin, tmax, t1);
      p2 = slab_pass(mk3(sx ? afx.z : anx.z, sy ? afy.z : any.z, sz ? afz.z : anz.z), mk3(sx ? anx.z : afx.z, sy ? any.z : afy.z, sz ? anz.z : afz.z), tmin, tmax, t2);
      p3 = slab_pass(mk3(sx ? afx.w : anx.w, sy ? afy.w : any.w, sz ? afz.w : anz.w), mk3(sx ? anx.w : afx.w, sy ? any.w : afy.w, sz ? anz.w : afz.w), tmin, tmax, t3);
    } else {
      p0 = slab_pass_signed(anx.x, any.x, anz.x, afx.x, afy.x, afz.x, tmin, tmax, t0);
      p1 = slab_pass_signed(anx.y, any.y, anz.y, afx.y, afy.y, afz.y, tmin, tmax, t1);
      p2 = slab_pass_signed(anx.z, any.z, anz.z, afx.z, afy.z, afz.z, tmin, tmax, t2);
      p3 = slab_pass_signed(anx.w, any.w, anz.w, afx.w, afy.w, afz.w, tmin, tmax, t3);
    }
    // slots 0,1 = children of child 0, slots 2,3 = children of child 1 of the binary node.  The reference
    // pushes child 0 then child 1 when the ray is negative along the node's axis (child 1 popped first),
    // and does the same one level down when that child is popped: visit order = [group][member].
    int  r0 = p0 ? __float_as_int(qr.x) : VPT_NONE, r1 = p1 ? __float_as_int(qr.y) : VPT_NONE;
    int  r2 = p2 ? __float_as_int(qr.z) : VPT_NONE, r3 = p3 ? __float_as_int(qr.w) : VPT_NONE;
    bool gn = (csgn >> (meta & 3)) & 1, g0 = (csgn >> ((meta >> 2) & 3)) & 1, g1 = (csgn >> ((meta >> 4) & 3)) & 1;
    int   a0 = g0 ? r1 : r0, a1 = g0 ? r0 : r1, b0 = g1 ? r3 : r2, b1 = g1 ? r2 : r3;
    float s0 = g0 ? t1 : t0, s1 = g0 ? t0 : t1, u0 = g1 ? t3 : t2, u1 = g1 ? t2 : t3;
    int   v0 = gn ? b0 : a0, v1 = gn ? b1 : a1, v2 = gn ? a0 : b0, v3 = gn ? a1 : b1;
    float w0 = gn ? u0 : s0, w1 = gn ? u1 : s1, w2 = gn ? s0 : u0, w3 = gn ? s1 : u1;
    // push the later-visited ones (last first); the first-visited one is taken directly: it would be
    // popped next with the same tmax, so its pop test is a tautology
    int next;
    if (stk.room_for(sp, 4)) {
      // branch-free: every candidate is stored; one that is not pushed lands on the free entry above the new
      // top (the host sizes the LDS part one entry larger than the worst case; the overflow variant takes this path while all lanes have room)
      bool f0 = v0 != VPT_NONE, f1 = v1 != VPT_NONE, f2 = v2 != VPT_NONE, f3 = v3 != VPT_NONE;
      bool q3 = f3 && (f2 || f1 || f0), q2 = f2 && (f1 || f0), q1 = f1 && f0;
      int  top = sp + (int)q3 + (int)q2 + (int)q1;
      stk.put(q3 ? sp : top, v3, w3);
      stk.put(q2 ? sp + (int)q3 : top, v2, w2);
      stk.put(q1 ? sp + (int)q3 + (int)q2 : top, v1, w1);
      sp   = top;
      next = f0 ? v0 : f1 ? v1 : f2 ? v2 : v3;
    } else {
      float nt = w3;
      next     = v3;
      if (v2 != VPT_NONE) {
        if (next != VPT_NONE) stk.push(sp, next, nt);
        next = v2, nt = w2;
      }
      if (v1 != VPT_NONE) {
        if (next != VPT_NONE) stk.push(sp, next, nt);
        next = v1, nt = w1;
      }
      if (v0 != VPT_NONE) {
        if (next != VPT_NONE) stk.push(sp, next, nt);
        next = v0;
      }
    }
    cur = next != VPT_NONE ? next : pop_valid();
  };

  // ---- own form of B: the primitives of the shape leaf `cur`, in order ---------------------------------------------------
  auto own_leaf = [&]() {
    int code = ~cur, start = code >> 4, num = code & 15;
    VPT_CNT(CNT_LEAF);
    // software-pipelined: the next primitive's record is in flight while this one is tested (a leaf's
    // records are contiguous; one past the last primitive of the pool is still inside the padded array)
    const float4* rec = leaf_rec<COMPACT>(sc, leafb + start);
    float4 n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[LS - 1];   // (a compact record ends after its third corner: p3 = p2, the reference's triangle)
    for (int k = 0; k < num; k++) {
      VPT_CNT(CNT_PRIM);
      float4 r0 = n0, r1 = n1, r2 = n2, r3 = n3;
      rec += LS;
      n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[LS - 1];
      if (intersect_quad(co, cd, tmin, tmax, xyz(r0), xyz(r1), xyz(r2), xyz(r3), r.uv, tmax))   // (an accepted hit's distance IS the new tmax)
        r.instance = cur_inst, r.prim = leafb + start + k;
    }
    // A hit with a NaN distance (a triangle of denormal size met exactly at a corner: 0 * inf) leaves tmax = NaN, and the reference's
    // fmin(far, tmax) = (far < tmax) ? far : tmax then fails every later box test.  The hardware min / max of the fast box forms drop a
    // NaN operand instead, so such a ray walks on in the reference's NaN-asymmetric form (tests/test_gpu_parity.py: degenerate triangles)
    if (tmax != tmax) slow = true, wslow = true;
    cur = pop_valid();
  };

#ifdef VPT_TRAVERSE_GUARD
  int guard = 0;
#endif
  // ---- phase A of a small set: the rays of the lanes in `m` (at most 16, each on a quad node) on four lanes each, until every one of
  // them holds a leaf or nothing (the loop the own form runs lane by lane: same steps, same order, a quarter of the work per lane) ----
  auto group_nodes = [&](unsigned long long m) {
    VPT_CNT_MASK(CNT_SESSION, m);   // "lanes" of this counter = rays handed over
    const int  lane = threadIdx.x, j = lane & 3;
    const bool mine = (m >> lane) & 1;
    const int  rank = lanes_below(m), n = __popcll(m);
    // ray k of the set (the k-th set bit of m) goes to lanes 4k .. 4k+3: its owner posts its lane number to lane 4k (lanes without a
    // ray post to lane 1, which nobody reads), the group's other lanes copy it
    int        owner = quad_lane0(__builtin_amdgcn_ds_permute(mine ? rank << 4 : 4, lane));
    const bool gact  = (lane >> 2) < n;
    if (!gact) owner = lane;
    const f3    gco = pull(owner, co), gcinv = pull(owner, cinv);
    const float gtmax = pull(owner, tmax);
    int         gcur = pull(owner, cur), gsp = pull(owner, sp);
    const int   gwnb = pull(owner, wnb);
    const int   gmisc = pull(owner, (shape_base >= 0 ? shape_base : 0) | csgn << 8 | (slow ? 2048 : 0));
    if (!gact) gcur = VPT_NONE;
    const int  gfloor = gmisc & 255, gcsgn = (gmisc >> 8) & 7;   // pop floor of the ray's level, sign bits of its direction
    const bool gslow = (gmisc & 2048) != 0;                      // NaN-prone direction
    const STK  gstk = stk.of_lane(owner);
    while (__builtin_amdgcn_ballot_w64(gcur >= 0) != 0) {
      VPT_CNT_MASK(CNT_GNODE, __builtin_amdgcn_ballot_w64(gcur >= 0));
      VPT_HIST(__popcll(__builtin_amdgcn_ballot_w64(gcur >= 0)) >> 2);
#ifdef VPT_TRAVERSE_GUARD
      if (++guard > VPT_TRAVERSE_GUARD) gcur = VPT_NONE;   // (the outer loop reports it)
#endif
      if (gcur >= 0) {
        const float* q = (const float*)(sc.scene_wnodes + 8 * (long long)(gwnb + gcur));
        float lox = q[j], loy = q[4 + j], loz = q[8 + j], hix = q[12 + j], hiy = q[16 + j], hiz = q[20 + j];
        int   ref = __float_as_int(q[24 + j]), meta = __float_as_int(q[28]);
        float ax = (lox - gco.x) * gcinv.x, ay = (loy - gco.y) * gcinv.y, az = (loz - gco.z) * gcinv.z;
        float bx = (hix - gco.x) * gcinv.x, by = (hiy - gco.y) * gcinv.y, bz = (hiz - gco.z) * gcinv.z;
        float t0;
        bool  pass;
        if (__builtin_amdgcn_ballot_w64(gslow) != 0) pass = slab_pass(mk3(ax, ay, az), mk3(bx, by, bz), tmin, gtmax, t0);   // the reference's form
        else {   // no product can be NaN: min / max of the two products per axis in any association (as slab_pass_signed)
          t0       = hw_max3(hw_max(hw_min(ax, bx), hw_min(ay, by)), hw_min(az, bz), tmin);
          float t1 = hw_min3(hw_min(hw_max(ax, bx), hw_max(ay, by)), hw_max(az, bz), gtmax) * VPT_BOX_K;
          pass     = t0 <= t1;
        }
        const int  rj = pass ? ref : VPT_NONE;   // an empty slot holds VPT_NONE itself
        const bool present = rj != VPT_NONE;
        // visit rank of grandchild j in the reference's order [group by the node's axis][member by the child's axis]
        const int gn = (gcsgn >> (meta & 3)) & 1, g0 = (gcsgn >> ((meta >> 2) & 3)) & 1, g1 = (gcsgn >> ((meta >> 4) & 3)) & 1;
        const int grp = j >> 1;
        const int k = ((grp ^ gn) << 1) | ((j & 1) ^ (grp ? g1 : g0));
        const int pv = quad_or(present ? 1 << k : 0);              // passing children by visit rank
        const int kfirst = pv ? __builtin_ctz(pv) : 4;
        // the first-visited one is the next node (popped next with the same tmax: its pop test is a tautology); the others are
        // pushed last-visited first: rank k lands above the passing ranks greater than k
        if (present && k > kfirst) {
          if (gstk.room_for(gsp, 4)) gstk.put(gsp + __builtin_popcount(pv >> (k + 1)), rj, t0);
          else gstk.store(gsp + __builtin_popcount(pv >> (k + 1)), rj, t0);
        }
        const int next = quad_or(present && k == kfirst ? rj : 0);
        gsp += pv ? __builtin_popcount(pv) - 1 : 0;
        gcur = next;
        if (!pv) {   // pop_valid on the owner's column; the four lanes of a group read the same entries
          gcur = VPT_NONE;
          const bool lds_only = gstk.room_for(gsp, 0);
          while (gsp > gfloor) {
            int   pref;
            float pt0;
            gsp--;
            if (lds_only) gstk.load_lds(gsp, pref, pt0);
            else gstk.load(gsp, pref, pt0);
            if (pt0 <= gtmax * VPT_BOX_K) {
              gcur = pref;
              break;
            }
          }
        }
      }
    }
    // hand back: the owner of ray k reads lane 4k
    const int src = mine ? rank << 2 : 0;
    const int ncur = pull(src, gcur), nsp = pull(src, gsp);
    if (mine) cur = ncur, sp = nsp;
  };

  // ---- phase B of a small set: the rays of the lanes in `m` (at most 16, each on a shape leaf): the (<= 4) primitives of a leaf one per
  // lane of the ray's group.  Returns true for a lane whose leaf is still to be walked in the own form (a NaN met: see the file header) ----
  auto group_leaves = [&](unsigned long long m) {
    const int  lane = threadIdx.x, j = lane & 3;
    const bool mine = (m >> lane) & 1;
    const int  rank = lanes_below(m), n = __popcll(m);
    int        owner = quad_lane0(__builtin_amdgcn_ds_permute(mine ? rank << 4 : 4, lane));
    const bool gact  = (lane >> 2) < n;
    if (!gact) owner = lane;
    VPT_CNT_MASK(CNT_GLEAF, m);
    const f3 gco = pull(owner, co), gcd = pull(owner, cd);
    float    gtmax = pull(owner, tmax);
    const int gcode = pull(owner, ~cur), gleafb = pull(owner, leafb);
    const int start = gcode >> 4, num = gact ? gcode & 15 : 0;
    bool  ghit = false, bad = gact && gtmax != gtmax;
    float gu = 0, gv = 0;
    int   gprim = 0;
    const int rounds = __builtin_amdgcn_ballot_w64(num > 4) != 0 ? 4 : 1;   // the reference's leaves hold <= 4 primitives; the format allows 15
    for (int b = 0; b < 4 * rounds; b += 4) {
      f2    uv = mk2(0, 0);
      float t = 0;
      bool  hit = false;
      if (b + j < num && !bad) {
        const float4* rec = leaf_rec<COMPACT>(sc, gleafb + start + b + j);
        float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[LS - 1];
        hit  = intersect_quad(gco, gcd, tmin, gtmax, xyz(r0), xyz(r1), xyz(r2), xyz(r3), uv, t);
      }
      bad = bad || quad_or(hit && t != t ? 1 : 0) != 0;
      // smallest t of the four, on a tie the later primitive (a hit's t is finite: t <= tmax <= flt_max)
      float key = hit && !bad ? t : __builtin_inff();
      int   idx = j;
      float ok = __int_as_float(quad_xor1(__float_as_int(key)));
      int   oi = quad_xor1(idx);
      bool  take = ok < key || (ok == key && oi > idx);
      key = take ? ok : key, idx = take ? oi : idx;
      ok = __int_as_float(quad_xor2(__float_as_int(key))), oi = quad_xor2(idx);
      take = ok < key || (ok == key && oi > idx);
      key = take ? ok : key, idx = take ? oi : idx;
      const bool win = j == idx && key < __builtin_inff();
      const int  wu = quad_or(win ? __float_as_int(uv.x) : 0), wv = quad_or(win ? __float_as_int(uv.y) : 0);
      if (key < __builtin_inff()) ghit = true, gtmax = key, gu = __int_as_float(wu), gv = __int_as_float(wv), gprim = gleafb + start + b + idx;
    }
    // hand back: the owner of ray k reads lane 4k.  A group that met a NaN reports nothing: its owner walks the whole leaf itself
    const int src = mine ? rank << 2 : 0;
    const int nflag = pull(src, (ghit ? 1 : 0) | (bad ? 2 : 0));
    const float nt = pull(src, gtmax), nu = pull(src, gu), nv = pull(src, gv);
    const int   np = pull(src, gprim);
    if (mine && nflag == 1) tmax = nt, r.uv = mk2(nu, nv), r.prim = np, r.instance = cur_inst;
    return mine && (nflag & 2) != 0;
  };

  while (true) {
    VPT_CNT(CNT_OUTER);
#ifdef VPT_TRAVERSE_GUARD
    if (++guard > VPT_TRAVERSE_GUARD) {
      if (!done) atomicAdd(&g_vpt_guard_trips, 1u);
      break;
    }
#endif
    // ---- A: quad nodes, until no lane holds one.  More than VPT_COOP_MAX rays: every lane steps its own; fewer: four lanes per ray ----------
    VPT_T0(TM_NODES);
    const unsigned long long ma = __builtin_amdgcn_ballot_w64(!done && cur >= 0);
    if (ma != 0) {
      if (VPT_COOP_NODES > 0 && whole_wave && __popcll(ma) <= VPT_COOP_NODES) group_nodes(ma);
      else if (!done && cur >= 0) own_node_step();
      VPT_T1(TM_NODES);
      continue;
    }
    VPT_T1(TM_NODES);
    // ---- B: shape leaves --------------------------------------------------------------------------------------------------------------
    VPT_T0(TM_PRIMS);
    {
      const bool at_leaf = !done && cur != VPT_NONE && shape_base >= 0;   // (cur < 0 for every lane here)
      const unsigned long long mb = __builtin_amdgcn_ballot_w64(at_leaf);
      bool own = at_leaf;
      if (VPT_COOP_LEAVES > 0 && whole_wave && mb != 0 && __popcll(mb) <= VPT_COOP_LEAVES) {
        own = group_leaves(mb);
        if (at_leaf && !own) cur = pop_valid();
      }
      if (own) own_leaf();
    }
    VPT_T1(TM_PRIMS);
    // ---- C: a level is exhausted or a scene leaf is reached ---------------------------------------------------------------------
    VPT_T0(TM_ENTER);
    if (!done && cur < 0 && (cur == VPT_NONE || shape_base < 0)) {
      bool go = true;
      if (cur == VPT_NONE) {
        if (shape_base < 0 || only_instance >= 0) done = true, go = false;   // nothing left: query finished
        else {   // leaving an instance: back to world space, then its leaf's next instance or the next scene entry
          shape_base = -1, wnb = 0;
          cd = wd, cinv = winv, csgn = sign_bits(winv), slow = wslow;   // co: enter_pending
        }
      } else pend = ~cur;   // a scene leaf: its instances are entered one after another, in order
      if (go) {
        cur = enter_pending();
        if (cur == VPT_NONE && shape_base < 0) done = true;
      }
    }
    VPT_T1(TM_ENTER);
    if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
  }
  r.hit = r.instance >= 0;
  r.distance = r.hit ? tmax : 0;
  if (r.hit) r.element = __float_as_int(leaf_rec<COMPACT>(sc, r.prim)[0].w);   // the element id sits in the hit record: read once here rather than carried
  return r;
}

// sample_lights_pdf's mesh-light walk (yocto_pathtrace.cpp:359-380) for a light whose shape BVH is a
// single leaf: each hop is transform_ray + the root box test + <= 4 primitive tests, done inline from the
// light's record (DScene::light_rec / light_prims): every load depends on the light id only.
VPT_DEV float small_light_pdf(const DScene& sc, int light_id, float4 r6, float4 r7, f3 position, f3 direction) {
  const float4* rec   = sc.light_rec + 8 * (long long)light_id;
  const float4* prims = sc.light_prims + 20 * (long long)light_id;
  frame  inv   = unpack_frame(rec[0], rec[1], rec[2]);
  f3     ld    = transform_vector(inv, direction);
  f3     linv  = rcp3_exact(ld);
  bool   lslow = nan_prone(ld, linv);
  int    count = (__float_as_int(r7.w) >> 8) & 15;
  float  area  = r6.w;
  float  lpdf  = 0.0f;
  f3     next_position = position;
  for (int hop = 0; hop < 100; hop++) {
    f3    lo   = transform_point(inv, next_position);
    float tmax = VPT_FLT_MAX, t0;
    if (!box_test(lslow, mk3(r6.x, r6.y, r6.z), mk3(r7.x, r7.y, r7.z), lo, linv, VPT_RAY_EPS, tmax, t0)) break;
    int   kh = -1;
    f2    uv = mk2(0, 0);
    float dist = 0;
    for (int k = 0; k < count; k++) {
      float4 c0 = prims[5 * k], c1 = prims[5 * k + 1], c2 = prims[5 * k + 2], c3 = prims[5 * k + 3];
      if (intersect_quad(lo, ld, VPT_RAY_EPS, tmax, xyz(c0), xyz(c1), xyz(c2), xyz(c3), uv, dist)) kh = k, tmax = dist;
    }
    if (kh < 0) break;
    // eval_position (yocto_scene.cpp:279-303) from the hit primitive's own corners: a triangle repeats its last corner
    float4 c0 = prims[5 * kh], c1 = prims[5 * kh + 1], c2 = prims[5 * kh + 2], c3 = prims[5 * kh + 3], cn = prims[5 * kh + 4];
    bool   tri = __float_as_int(cn.w) != 0;   // the shape holds triangles (interpolate_triangle), not quads
    f3     lp  = tri ? tri_lerp(xyz(c0), xyz(c1), xyz(c2), uv)
                     : (uv.x + uv.y <= 1 ? tri_lerp(xyz(c0), xyz(c1), xyz(c3), uv) : tri_lerp(xyz(c2), xyz(c3), xyz(c1), 1 - uv));
    f3 lposition = transform_point(unpack_frame(rec[3], rec[4], rec[5]), lp);
    f3 lnormal   = xyz(cn);
    lpdf += distance_squared(lposition, position) / (fabs_(dot(lnormal, direction)) * area);
    next_position = lposition + direction * 1e-3f;
  }
  return lpdf;
}

// pdf of the non-mesh lights (environment / sdf), one light: yocto_pathtrace.cpp:381-417
template <int FEAT = VPT_FEAT_ALL>
VPT_DEV float other_light_pdf(const DScene& sc, int light_id, int kind, float4 r6, f3 position, f3 direction, int maxiter) {
  if (kind == VPT_LIGHT_ENV_CONST) return 1 / (4 * VPT_PI);
  const vpt_light& light = sc.lights[light_id];
  const float*     cdf   = sc.light_cdf + light.cdf_offset;
  if ((FEAT & VPT_FEAT_SDF_LIGHTS) && kind == VPT_LIGHT_SDF) {
    st_hit h = spheretrace_one(sc, position, direction, light.sdf, maxiter);
    if (!h.hit) return 0;
    f3 lposition = position + direction * h.dist;
    f3 lnormal   = eval_sdf_normal_function(scene_sdf_recs(sc), h.sdf, position, h.dist);
    return distance_squared(lposition, position) / (fabs_(dot(lnormal, direction)) * cdf[light.cdf_len - 1]);
  }
  if (kind == VPT_LIGHT_ENV_TEX) {
    const float4* rec = sc.light_rec + 8 * (long long)light_id;
    int tw = __float_as_int(r6.x), th = __float_as_int(r6.y);
    f3 wl = transform_direction(unpack_frame(rec[0], rec[1], rec[2]), direction);
    f2 tc = mk2(atan2f(wl.z, wl.x) / (2 * VPT_PI), acosf(clampf(wl.y, -1.0f, 1.0f)) / VPT_PI);
    if (tc.x < 0) tc.x += 1;
    int i = clampi((int)(tc.x * tw), 0, tw - 1), j = clampi((int)(tc.y * th), 0, th - 1);
    int idx = j * tw + i;
    float prob  = (idx == 0 ? cdf[0] : cdf[idx] - cdf[idx - 1]) / r6.z;
    float angle = (2 * VPT_PI / tw) * (VPT_PI / th) * sinf(VPT_PI * (j + 0.5f) / th);
    return prob / angle;
  }
  return 0;
}

// sample_lights_pdf's mesh-light walk (yocto_pathtrace.cpp:359-380) for an emissive mesh with a real BVH;
// binary-node form with a refs-only LDS stack, for the kernel that does not carry the quad-node traversal (K2)
VPT_DEV float general_light_pdf(const DScene& sc, const vpt_light& light, f3 position, f3 direction, const lane_stack& stk) {
  const DInstance& inst = sc.instances[light.instance];
  float area = sc.light_cdf[light.cdf_offset + light.cdf_len - 1];
  float lpdf = 0.0f;
  f3    next_position = position;
  for (int hop = 0; hop < 100; hop++) {
    hit_t h = trace_instance(sc, light.instance, next_position, direction, stk);
    if (!h.hit) break;
    f3 lposition = eval_position(sc, inst, h.element, h.uv);
    f3 lnormal   = eval_element_normal(sc, inst, h.element);
    lpdf += distance_squared(lposition, position) / (fabs_(dot(lnormal, direction)) * area);
    next_position = lposition + direction * 1e-3f;
  }
  return lpdf;
}
// one hop of the same walk for an emissive mesh with a real BVH (yocto_pathtrace.cpp:363-378): `h` is the hit of the
// single-instance query from the walk's current position; returns the hop's pdf term and moves the walk on
VPT_DEV float large_light_hop(const DScene& sc, int light_id, const hit_t& h, f3 position, f3 direction, f3& next_position) {
  const vpt_light& light = sc.lights[light_id];
  const DInstance& inst  = sc.instances[light.instance];
  f3    lposition = eval_position(sc, inst, h.element, h.uv);
  f3    lnormal   = eval_element_normal(sc, inst, h.element);
  float area      = sc.light_cdf[light.cdf_offset + light.cdf_len - 1];
  next_position   = lposition + direction * 1e-3f;
  return distance_squared(lposition, position) / (fabs_(dot(lnormal, direction)) * area);
}

// The shading point of a surface hit: eval_shading_position, eval_shading_normal and eval_material
// (yocto_scene.cpp:460-579) in one go.  `prim` is the hit's slot in leaf_prims / leaf_attrs.
template <bool COMPACT = false>
VPT_DEV void eval_surface_point(const DScene& sc, const DInstance& inst, const vpt_material& mat, int prim, int element, f2 uv,
    f3 outgoing, f3& position, f3& normal, mpoint& m) {
  f2 texcoord;
  f4 color_shp = mk4(1, 1, 1, 1);
  if ((inst.shape_flags & (VPT_SHP_NORMALS | VPT_SHP_COLORS)) == VPT_SHP_NORMALS) {
    // the common case: everything about the shading point sits behind the hit's primitive slot
    eval_surface_slot<COMPACT>(sc, inst, prim, uv, position, normal, texcoord);
  } else {   // no vertex normals (element normal) or vertex colours: through the element's vertex indices
    position  = eval_position(sc, inst, element, uv);
    normal    = eval_normal(sc, inst, element, uv);
    texcoord  = eval_texcoord(sc, inst, element, uv);
    color_shp = eval_color(sc, inst, element, uv);
  }
  // eval_shading_normal, yocto_scene.cpp:476-503
  if (mat.normal_tex != VPT_INVALID) normal = eval_normalmap(sc, inst, element, uv, normal, mat.normal_tex);
  if (mat.type != VPT_MAT_REFRACTIVE && !(dot(normal, outgoing) >= 0)) normal = -normal;
  m = eval_material_at(sc, mat, texcoord, color_shp);
}

enum { ST_NEW = 0, ST_MAIN = 1, ST_LPDF = 2 };

#ifndef VPT_WAVES_PER_SIMD
#define VPT_WAVES_PER_SIMD 2
#endif

#ifdef VPT_WAVE_TIMES
// diagnostic build: start / end of every wave on the 100 MHz wall clock, to draw the launch's occupancy timeline
__device__ unsigned long long g_vpt_wave_times[2 * 65536];
__device__ unsigned g_vpt_wave_hw[65536];   // XCC_ID << 16 | HW_ID[15:0] (wave, simd, pipe, cu, sh, se)
#endif

template <int SH, bool SPILL, int FEAT>
VPT_DEV void mesh_kernel_body(const DScene& sc, const DParams& pr, float4* __restrict__ image, int* __restrict__ hits,
    ulonglong2* __restrict__ rngs, const stack_cfg& stack, const sched_cfg& sched) {
  extern __shared__ int lds_stack[];
  const lane_stack2<SPILL> stk = make_lane_stack<SPILL>(lds_stack, stack);
  __shared__ unsigned long long s_wave_start;   // the start stamp, parked in LDS (clock_ticks, vpt_math.hip.h)
  if (threadIdx.x == 0) s_wave_start = clock_ticks(blockIdx.x);
  const int wave = sched.order ? sched.order[blockIdx.x] : (int)blockIdx.x;
#ifdef VPT_COUNTERS
  if (threadIdx.x < 16) s_vpt_time[threadIdx.x] = 0;
#endif

  int slot = wave * VPT_BLOCK + threadIdx.x;
  if (sched.lane_slot) slot = sched.lane_slot[slot];   // a split tile: this wave holds every 2^k-th pixel of it in its first lanes
  // Cold per-pixel state lives in LDS behind the stacks instead of in registers for the whole launch: the radiance sum
  // (touched once per sample) and the pixel's coordinates (once per sample) - five words per lane.
  float* const park = (float*)(lds_stack + stack.cap * 2 * VPT_BLOCK) + threadIdx.x;
  // Lanes that own no pixel (padding slots of a ragged frame, the empty lanes of a split tile) stay in the kernel: the whole wave
  // goes through every BVH query together, and a lane without a ray works on the others' rays there (traverse(): group forms)
  // (no lane mask is kept for "owns a pixel" / "has samples left": both are read off slot, state and sample, which are there anyway -
  // the kernel has no scalar register to spare)
  {
    int px0 = 0, py0 = 0;
    if (!(slot >= 0 && slot < pr.nslots && slot_to_pixel(pr, slot, px0, py0))) slot = -1;
    park[4 * VPT_BLOCK] = __int_as_float(px0 | (py0 << 16));
  }

  float4     acc_in = slot >= 0 ? image[slot] : make_float4(0, 0, 0, 0);
  park[0] = acc_in.x, park[VPT_BLOCK] = acc_in.y, park[2 * VPT_BLOCK] = acc_in.z, park[3 * VPT_BLOCK] = acc_in.w;
  ulonglong2 r_in   = slot >= 0 ? rngs[slot] : make_ulonglong2(0, 1);
  rng_t      rng    = {r_in.x, r_in.y};
  const int  nb     = (SH == K_EYELIGHT) ? max(pr.bounces, 4) : pr.bounces;
  constexpr bool HAS_MIS = (SH == K_VOLPATH || SH == K_PATH);
  constexpr bool HAS_LARGE = HAS_MIS && (FEAT & VPT_FEAT_LARGE_LIGHTS) != 0;   // else ST_LPDF is never entered

  // path state
  ray_t ray      = make_ray(mk3(0, 0, 0), mk3(0, 0, 1));
  f3    radiance = mk3(0, 0, 0), weight = mk3(1, 1, 1);
  float alpha  = 0;
  int   bounce = 0, sample = 0, state = ST_NEW;
  bool  in_medium = false;
  f3    med_density = mk3(0, 0, 0), med_scattering = mk3(0, 0, 0), med_emission = mk3(0, 0, 0);
  float med_g = 0;
  // pending MIS evaluation: f / (0.5 pdf + 0.5 sum_lights) is finished after the light-pdf walk
  f3    mis_f = mk3(0, 0, 0), lp_pos = mk3(0, 0, 0);
  float mis_pdf = 0, lp_sum = 0, lp_cur = 0;
  int   lp_light = 0, lp_hop = 0;
  bool  mis_toggle = false;

  if (slot < 0) sample = pr.nsamples;   // a lane without a pixel has nothing to render: it starts where the others end
  VPT_T0(TM_KERNEL);
  while (true) {
    const bool alive = !(state == ST_NEW && sample == pr.nsamples);   // this lane's pixel still has samples to render
    if (__builtin_amdgcn_ballot_w64(alive) == 0) break;   // wave-uniform: every lane stays until the tile is finished
    if (alive && state == ST_NEW) {
      VPT_CNT(CNT_GENERATE);
      VPT_T0(TM_GENERATE);
      const vpt_camera& cam = sc.cameras[pr.camera];
      float u, v;
      if (pr.preview) {
        const int pxy = __float_as_int(park[4 * VPT_BLOCK]), px = pxy & 0xffff, py = pxy >> 16;
        u = (px + 0.5f) / pr.width, v = (py + 0.5f) / pr.height;
      } else {
        const int pxy = __float_as_int(park[4 * VPT_BLOCK]), px = pxy & 0xffff, py = pxy >> 16;
        u = (px + rand1f(rng)) / pr.width;
        v = (py + rand1f(rng)) / pr.height;
      }
      f2 lens;
      lens.x   = rand1f(rng);
      lens.y   = rand1f(rng);
      ray      = eval_camera(cam, mk2(u, v), lens);
      radiance = mk3(0, 0, 0), weight = mk3(1, 1, 1);
      alpha = 0, bounce = 0, in_medium = false, state = ST_MAIN;
      VPT_T1(TM_GENERATE);
    }

    // ---- the one BVH query of this trip: the whole wave is in the call, lanes without a ray as helpers ---------------
    const bool query = alive && !(state == ST_MAIN && SH != K_DEBUG && bounce >= nb);
    bool finish = alive && !query;   // the path ran out of bounces
    const bool lpdf_query = HAS_LARGE && query && state == ST_LPDF;
    const int  qinst      = lpdf_query ? sc.lights[lp_light].instance : -1;
    VPT_T0(TM_QUERY);
    hit_t h = traverse<(FEAT & VPT_FEAT_COMPACT_TRIS) != 0>(sc, query, lpdf_query ? lp_pos : ray.o, ray.d, qinst, stk);
    VPT_T1(TM_QUERY);
    if constexpr (!HAS_LARGE) {
      // without light walks that span trips (ST_LPDF) a pending MIS evaluation never outlives its trip: say so, or its seven words
      // stay allocated through every BVH query
      mis_f = mk3(0, 0, 0), mis_pdf = 0, lp_sum = 0, lp_light = 0, mis_toggle = false;
    }
    if (query) {
      VPT_CNT(CNT_TRIP);

      bool advance_lights = false;   // continue the light-pdf walk at lp_light
      if (lpdf_query) {
        // one hop of the mesh-light pdf loop, yocto_pathtrace.cpp:363-378 (position = ray.o, direction = ray.d)
        bool light_done = true;
        if (h.hit) {
          lp_cur += large_light_hop(sc, lp_light, h, ray.o, ray.d, lp_pos);
          lp_hop++;
          light_done = lp_hop >= 100;
        }
        if (light_done) lp_sum += lp_cur, lp_light++, advance_lights = true;
      } else if (!h.hit) {
        if constexpr (SH != K_DEBUG) radiance = radiance + weight * eval_environment(sc, ray.d);
        finish = true;
      } else if constexpr (SH == K_DEBUG) {   // shade_normal / texcoord / color, cpp:893-930
        const DInstance& inst = sc.instances[h.instance];
        if (pr.shader == VPT_SHADER_NORMAL) radiance = eval_shading_normal(sc, inst, h.element, h.uv, -ray.d);
        else if (pr.shader == VPT_SHADER_TEXCOORD) {
          f2 t     = eval_texcoord(sc, inst, h.element, h.uv);
          radiance = mk3(t.x, t.y, 0);
        } else radiance = eval_material(sc, inst, h.element, h.uv).color;
        alpha  = 1;
        finish = true;
      } else {
        bool in_volume = false;
        // a scattering event of pathtrace / volpathtrace is finished after the light sampling both event kinds
        // share (one sample_lights for the lanes of either kind): 1 = surface, 2 = medium
        int    scatter = 0;
        bool   want_lights = false, vol_boundary = false;
        f3     outgoing = -ray.d, position = mk3(0, 0, 0), normal = mk3(0, 0, 0), incoming = mk3(0, 0, 0);
        f2     l_ruv = mk2(0, 0);
        float  l_rel = 0, l_rl = 0;
        mpoint m;
        VPT_T0(TM_MEDIUM);
        if constexpr (SH == K_VOLPATH) {
          if (in_medium) {   // cpp:586-596 — rd is drawn before rl
            float rd       = rand1f(rng);
            float rl       = rand1f(rng);
            float distance = sample_transmittance(med_density, h.distance, rl, rd);
            weight = weight * (vexp3(-med_density * distance) / sample_transmittance_pdf(med_density, distance, h.distance));
            in_volume  = distance < h.distance;
            h.distance = distance;
          }
        }
        VPT_T1(TM_MEDIUM);
        VPT_T0(TM_SURFACE);
        if (!in_volume) {
          const DInstance& inst = sc.instances[h.instance];
          VPT_T0(TM_SURF_GEOM);
          eval_surface_point<(FEAT & VPT_FEAT_COMPACT_TRIS) != 0>(sc, inst, sc.materials[inst.material], h.prim, h.element, h.uv, outgoing, position, normal, m);
          VPT_T1(TM_SURF_GEOM);
          if (m.opacity < 1 && rand1f(rng) >= m.opacity) {
            ray = make_ray(position + ray.d * 1e-2f, ray.d);   // bounce -= 1; continue
          } else {
            if (bounce == 0) alpha = 1;
            radiance = radiance + weight * eval_emission(m.emission, normal, outgoing);
            if constexpr (SH == K_EYELIGHT) {   // cpp:869-886
              incoming = outgoing;
              radiance = radiance + weight * VPT_PI * eval_bsdfcos(m, normal, outgoing, incoming);
              if (!is_delta(m)) finish = true;
              else {
                incoming = sample_delta(m, normal, outgoing, rand1f(rng));
                if (is_zero3(incoming)) finish = true;
                else {
                  weight = weight * (eval_delta(m, normal, outgoing, incoming) / sample_delta_pdf(m, normal, outgoing, incoming));
                  if (is_zero3(weight) || !finite3(weight)) finish = true;
                  else ray = make_ray(position, incoming);
                }
              }
              bounce++;
            } else if constexpr (SH == K_NAIVE) {   // cpp:802-828
              if (m.roughness != 0) {
                f2 rn;
                rn.x      = rand1f(rng);
                rn.y      = rand1f(rng);
                float rnl = rand1f(rng);
                incoming  = sample_bsdfcos(m, normal, outgoing, rnl, rn);
                if (is_zero3(incoming)) finish = true;
                else weight = weight * (eval_bsdfcos(m, normal, outgoing, incoming) / sample_bsdfcos_pdf(m, normal, outgoing, incoming));
              } else {
                incoming = sample_delta(m, normal, outgoing, rand1f(rng));
                if (is_zero3(incoming)) finish = true;
                else weight = weight * (eval_delta(m, normal, outgoing, incoming) / sample_delta_pdf(m, normal, outgoing, incoming));
              }
              if (!finish) {
                if (!survive(weight, bounce, rng)) finish = true;
                else ray = make_ray(position, incoming);
              }
              bounce++;
            } else {   // pathtrace / volpathtrace, cpp:619-651
              vol_boundary = SH == K_VOLPATH && is_volumetric_type(sc.materials[inst.material].type);
              if (!is_delta(m)) {
                if (rand1f(rng) < 0.5f) {
                  f2 rn;
                  rn.x      = rand1f(rng);
                  rn.y      = rand1f(rng);
                  float rnl = rand1f(rng);
                  incoming  = sample_bsdfcos(m, normal, outgoing, rnl, rn);
                } else {
                  l_ruv.x = rand1f(rng);
                  l_ruv.y = rand1f(rng);
                  l_rel   = rand1f(rng);
                  l_rl    = rand1f(rng);
                  want_lights = true;
                }
                scatter = 1;
              } else {
                VPT_T0(TM_SURF_DELTA);
                float rnl = rand1f(rng);
                incoming  = sample_delta(m, normal, outgoing, rnl);
                weight    = weight * (eval_delta(m, normal, outgoing, incoming) / sample_delta_pdf(m, normal, outgoing, incoming));
                if (vol_boundary && dot(normal, outgoing) * dot(normal, incoming) < 0) {
                  if (!in_medium) {
                    in_medium   = true;
                    med_density = m.density, med_scattering = m.scattering, med_emission = m.emission, med_g = m.scanisotropy;
                  } else {
                    in_medium = false;
                  }
                }
                ray = make_ray(position, incoming);
                if (!survive(weight, bounce, rng)) finish = true;
                bounce++;
                VPT_T1(TM_SURF_DELTA);
              }
            }
          }
        }
        VPT_T1(TM_SURFACE);
        VPT_T0(TM_VOLUME);
        if (in_volume) if constexpr (SH == K_VOLPATH) {   // volume event, cpp:654-673
          position = ray_point(ray, h.distance);
          radiance = radiance + weight * eval_emission(med_emission, position, outgoing);   // (sic) cpp:660
          if (rand1f(rng) < 0.5f) {
            f2 rn;
            rn.x = rand1f(rng);
            rn.y = rand1f(rng);
            (void)rand1f(rng);   // rnl is drawn and ignored, cpp:665
            incoming = sample_phasefunction(med_g, outgoing, rn);
          } else {
            l_ruv.x = rand1f(rng);
            l_ruv.y = rand1f(rng);
            l_rel   = rand1f(rng);
            l_rl    = rand1f(rng);
            want_lights = true;
          }
          scatter = 2;
        }
        VPT_T1(TM_VOLUME);
        if constexpr (HAS_MIS) {
          VPT_T0(TM_SAMPLE_LIGHTS);
          if (want_lights) incoming = sample_lights<FEAT>(sc, position, l_rl, l_rel, l_ruv);
          VPT_T1(TM_SAMPLE_LIGHTS);
          VPT_T0(TM_SCATTER_EVAL);
          if (scatter == 1) {   // cpp:626-648
            if (is_zero3(incoming)) finish = true;
            else {
              mis_f      = eval_bsdfcos(m, normal, outgoing, incoming);
              mis_pdf    = sample_bsdfcos_pdf(m, normal, outgoing, incoming);
              mis_toggle = vol_boundary && dot(normal, outgoing) * dot(normal, incoming) < 0;
              if (mis_toggle && !in_medium)   // entering: the medium slot is free, fill it now
                med_density = m.density, med_scattering = m.scattering, med_emission = m.emission, med_g = m.scanisotropy;
              ray = make_ray(position, incoming);
              lp_sum = 0, lp_light = 0, advance_lights = true;
            }
          } else if (scatter == 2) {   // cpp:666-671
            mis_f      = med_density * med_scattering * eval_phasefunction(med_g, incoming, outgoing);
            mis_pdf    = eval_phasefunction(med_g, outgoing, incoming);
            mis_toggle = false;
            ray        = make_ray(position, incoming);
            lp_sum = 0, lp_light = 0, advance_lights = true;
          }
          VPT_T1(TM_SCATTER_EVAL);
        }
      }

      if constexpr (HAS_MIS) {
        if (advance_lights) {   // sample_lights_pdf's loop over lights, resumable (cpp:353-421)
          VPT_T0(TM_LIGHTS_PDF);
          state = ST_MAIN;
          while (lp_light < sc.num_lights) {
            float4 r6 = sc.light_rec[8 * lp_light + 6], r7 = sc.light_rec[8 * lp_light + 7];
            int    kind = __float_as_int(r7.w) & 255;
            if (kind == VPT_LIGHT_SMALL_MESH) {   // single-leaf shape: inline walk
              lp_sum += small_light_pdf(sc, lp_light, r6, r7, ray.o, ray.d);
            } else if (HAS_LARGE && kind == VPT_LIGHT_LARGE_MESH) {   // needs real BVH hops: hand over to the traversal (extra trips)
              lp_cur = 0, lp_hop = 0, lp_pos = ray.o, state = ST_LPDF;
              break;
            } else {
              lp_sum += other_light_pdf<FEAT>(sc, lp_light, kind, r6, ray.o, ray.d, pr.spheretrace_maxiter);
            }
            lp_light++;
          }
          if (state == ST_MAIN) {   // all lights visited: finish the MIS weight (cpp:630-634 / 668-671)
            float lights_pdf = lp_sum * ((float)1 / (float)sc.num_lights);
            weight = weight * (mis_f / (0.5f * mis_pdf + 0.5f * lights_pdf));
            if (mis_toggle) in_medium = !in_medium;   // cpp:642-648
            if (!survive(weight, bounce, rng)) finish = true;
            bounce++;
          }
          VPT_T1(TM_LIGHTS_PDF);
        }
      }
    }

    if (finish) {   // cpp:1087-1089
      f4 rad = mk4(radiance.x, radiance.y, radiance.z, alpha);
      if (!(isfinite(rad.x) && isfinite(rad.y) && isfinite(rad.z) && isfinite(rad.w))) rad = mk4(0, 0, 0, 0);
      park[0] = park[0] + rad.x, park[VPT_BLOCK] = park[VPT_BLOCK] + rad.y, park[2 * VPT_BLOCK] = park[2 * VPT_BLOCK] + rad.z, park[3 * VPT_BLOCK] = park[3 * VPT_BLOCK] + rad.w;
      sample++;
      state = ST_NEW;
    }
  }

  VPT_T1(TM_KERNEL);
#ifdef VPT_COUNTERS
  if ((threadIdx.x & 63) == 0)   // lane 0 owns a pixel whenever the wave does (padding lanes sit at the end)
    for (int k = 0; k < 16; k++) atomicAdd(&g_vpt_cnt[32 + k], s_vpt_time[k]);
#endif
  if (slot >= 0) {
    image[slot] = make_float4(park[0], park[VPT_BLOCK], park[2 * VPT_BLOCK], park[3 * VPT_BLOCK]);
    hits[slot] += pr.nsamples;
    ulonglong2 r_out;
    r_out.x = rng.state, r_out.y = rng.inc;
    rngs[slot] = r_out;
  }
  if (sched.cost && threadIdx.x == 0 && slot >= 0) {   // lane 0 (the tile's corner pixel) exists whenever the wave owns a pixel
    const unsigned long long wave_start = s_wave_start;
    unsigned long long dt = clock_ticks(slot) - wave_start;   // after the last sample was accumulated
    sched.cost[wave] = dt < 0xffffffffull ? (unsigned)dt : 0xffffffffu;
#ifdef VPT_WAVE_TIMES
    if (wave < 65536) {
      g_vpt_wave_times[2 * wave] = wave_start, g_vpt_wave_times[2 * wave + 1] = wave_start + dt;
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_vpt_wave_hw[wave] = (xcc & 0xf) << 16 | (hw & 0xffff);
    }
#endif
  }
}

template <int SH, bool SPILL, int FEAT>
__global__ void __launch_bounds__(VPT_BLOCK, VPT_WAVES_PER_SIMD) vpt_mesh_kernel(DScene sc, DParams pr,
    float4* __restrict__ image, int* __restrict__ hits, ulonglong2* __restrict__ rngs, stack_cfg stack, sched_cfg sched) {
  mesh_kernel_body<SH, SPILL, FEAT>(sc, pr, image, hits, rngs, stack, sched);
}
// The same kernel under another name: the one-sample launch that measures per-wave costs when none are
// known yet (vpt_capi.hip).  Kept apart so that profiles of vpt_mesh_kernel only hold full launches.
template <int SH, bool SPILL, int FEAT>
__global__ void __launch_bounds__(VPT_BLOCK, VPT_WAVES_PER_SIMD) vpt_mesh_pilot_kernel(DScene sc, DParams pr,
    float4* __restrict__ image, int* __restrict__ hits, ulonglong2* __restrict__ rngs, stack_cfg stack, sched_cfg sched) {
  mesh_kernel_body<SH, SPILL, FEAT>(sc, pr, image, hits, rngs, stack, sched);
}
