// vpt_kat_kernels.hip.h — the known-answer-test kernel behind vpt_kat() (include/vpt_kat.h): one lane per
// record, each op calls the device function(s) the render kernels call for that piece of the path — no
// copies of their arithmetic — so a table produced by the reference's function of the same name checks the
// production code itself.  Record layouts: include/vpt_kat.h.
#pragma once
#include "vpt_kat.h"
#include "vpt_implicit_kernel.hip.h"

VPT_DEV void kat_put3(float* o, f3 v) { o[0] = v.x, o[1] = v.y, o[2] = v.z; }

// sample_lights_pdf as K1 evaluates it (vpt_mesh_kernel.hip.h, `advance_lights`): light records, inline walks of
// single-leaf mesh lights, quad-node hops for emissive meshes with a real BVH
// (every lane of the wave calls this, `live` = the lane holds a record: the hops go through traverse() as a whole wave)
template <class STK>
VPT_DEV float kat_lights_pdf_k1(const DScene& sc, bool live, f3 position, f3 direction, int maxiter, const STK& stk) {
  float sum = 0;
  for (int l = 0; l < sc.num_lights; l++) {
    float4 r6 = sc.light_rec[8 * l + 6], r7 = sc.light_rec[8 * l + 7];
    int    kind = __builtin_amdgcn_readfirstlane(__float_as_int(r7.w) & 255);   // the same light for every lane
    if (kind == VPT_LIGHT_SMALL_MESH) {
      if (live) sum += small_light_pdf(sc, l, r6, r7, position, direction);
    } else if (kind == VPT_LIGHT_LARGE_MESH) {
      float cur = 0;
      f3    pos = position;
      bool  walking = live;
      for (int hop = 0; hop < 100 && __builtin_amdgcn_ballot_w64(walking) != 0; hop++) {
        hit_t h = traverse(sc, walking, pos, direction, sc.lights[l].instance, stk);
        if (walking && !h.hit) walking = false;
        if (walking) cur += large_light_hop(sc, l, h, position, direction, pos);
      }
      sum += cur;
    } else {
      if (live) sum += other_light_pdf(sc, l, kind, r6, position, direction, maxiter);
    }
  }
  return sum * ((float)1 / (float)sc.num_lights);
}

template <bool SPILL>
__global__ void __launch_bounds__(VPT_BLOCK, VPT_WAVES_PER_SIMD) vpt_kat_kernel(DScene sc, int op, int iparam, int n, int si, int so,
    const float* __restrict__ in, const int* __restrict__ aux, float* __restrict__ out, stack_cfg stack, int stack_cap) {
  extern __shared__ int lds_stack[];
  const lane_stack2<SPILL> stk4 = make_lane_stack<SPILL>(lds_stack, stack);   // quad-node traversal (K1)
  lane_stack stk2;                                                             // binary-node traversal (K2's light walk)
  stk2.base = lds_stack + threadIdx.x, stk2.cap = stack_cap;
  int i = blockIdx.x * VPT_BLOCK + threadIdx.x;
  const bool live = i < n;
  // the two ops that run BVH queries keep the whole wave together (traverse(): the group forms need every lane); `op` is the same for all lanes
  if (op == VPT_KAT_INTERSECT || op == VPT_KAT_LIGHTS_PDF) {
    const float* a = in + (long long)(live ? i : 0) * si;
    float*       o = out + (long long)(live ? i : 0) * so;
    if (op == VPT_KAT_INTERSECT) {
      hit_t h = traverse(sc, live, ld3(a), ld3(a + 3), (int)a[6], stk4);
      if (!live) return;
      o[0] = h.hit ? (float)h.instance : -1.0f, o[1] = h.hit ? (float)h.element : -1.0f;
      o[2] = h.hit ? h.uv.x : 0, o[3] = h.hit ? h.uv.y : 0, o[4] = h.hit ? h.distance : 0;
    } else {
      float pdf = kat_lights_pdf_k1(sc, live, ld3(a), ld3(a + 3), iparam, stk4);
      if (live) o[0] = pdf;
    }
    return;
  }
  if (!live) return;
  const float* a = in + (long long)i * si;
  float*       o = out + (long long)i * so;
  for (int k = 0; k < so; k++) o[k] = 0;
  switch (op) {
    case VPT_KAT_LOBES: {
      mpoint m;
      m.type = (int)a[0], m.emission = mk3(0, 0, 0), m.color = ld3(a + 1), m.opacity = 1;
      m.roughness = a[4], m.metallic = a[5], m.ior = a[6];
      m.density = mk3(0, 0, 0), m.scattering = mk3(0, 0, 0), m.scanisotropy = 0;
      f3    normal = ld3(a + 7), outgoing = ld3(a + 10), alt = ld3(a + 16);
      float rnl = a[13];
      f2    rn  = mk2(a[14], a[15]);
      f3    s_in = sample_bsdfcos(m, normal, outgoing, rnl, rn);
      kat_put3(o, s_in);
      if (!is_zero3(s_in)) kat_put3(o + 3, eval_bsdfcos(m, normal, outgoing, s_in)), o[6] = sample_bsdfcos_pdf(m, normal, outgoing, s_in);
      kat_put3(o + 7, eval_bsdfcos(m, normal, outgoing, alt));
      o[10]   = sample_bsdfcos_pdf(m, normal, outgoing, alt);
      f3 d_in = sample_delta(m, normal, outgoing, rnl);
      kat_put3(o + 11, d_in);
      if (!is_zero3(d_in)) kat_put3(o + 14, eval_delta(m, normal, outgoing, d_in)), o[17] = sample_delta_pdf(m, normal, outgoing, d_in);
      kat_put3(o + 18, eval_delta(m, normal, outgoing, alt));
      o[21] = sample_delta_pdf(m, normal, outgoing, alt);
    } break;
    case VPT_KAT_MEDIA: {
      f3    density = ld3(a), outgoing = ld3(a + 7), incoming = ld3(a + 12);
      float maxd = a[3], rl = a[4], rd = a[5], g = a[6];
      f2    rn = mk2(a[10], a[11]);
      float distance = sample_transmittance(density, maxd, rl, rd);
      o[0] = distance, o[1] = sample_transmittance_pdf(density, distance, maxd);
      kat_put3(o + 2, vexp3(-density * distance));   // eval_transmittance as the shaders inline it
      o[5]     = eval_phasefunction(g, outgoing, incoming);
      f3 s_dir = sample_phasefunction(g, outgoing, rn);
      kat_put3(o + 6, s_dir);
      o[9] = eval_phasefunction(g, outgoing, s_dir);
    } break;
    case VPT_KAT_TEXTURE: {
      f4 c = eval_texture(sc, (int)a[0], mk2(a[1], a[2]), a[3] != 0);
      o[0] = c.x, o[1] = c.y, o[2] = c.z, o[3] = c.w;
    } break;
    case VPT_KAT_CAMERA: {
      ray_t ray = eval_camera(sc.cameras[(int)a[0]], mk2(a[1], a[2]), mk2(a[3], a[4]));
      kat_put3(o, ray.o), kat_put3(o + 3, ray.d);
    } break;
    case VPT_KAT_SURFACE: {
      const DInstance& inst = sc.instances[(int)a[0]];
      f3     position, normal;
      mpoint m;
      eval_surface_point(sc, inst, sc.materials[inst.material], aux[i], (int)a[1], mk2(a[2], a[3]), ld3(a + 4), position, normal, m);
      kat_put3(o, position), kat_put3(o + 3, normal);
      o[6] = (float)m.type;
      kat_put3(o + 7, m.emission), kat_put3(o + 10, m.color);
      o[13] = m.opacity, o[14] = m.roughness, o[15] = m.metallic, o[16] = m.ior;
      kat_put3(o + 17, m.density), kat_put3(o + 20, m.scattering);
      o[23] = m.scanisotropy;
    } break;
    case VPT_KAT_ENVIRONMENT: kat_put3(o, eval_environment(sc, ld3(a))); break;
    case VPT_KAT_SAMPLE_LIGHTS: kat_put3(o, sample_lights(sc, ld3(a), a[3], a[4], mk2(a[5], a[6]))); break;
    case VPT_KAT_LIGHTS_PDF_K2: o[0] = lights_pdf_k2(sc, ld3(a), ld3(a + 3), iparam, stk2); break;
    case VPT_KAT_SDF_SCENE: {
      sdf_hit r = eval_sdf_scene(sc, scene_sdf_recs(sc), ld3(a), a[3]);
      o[0] = r.result, o[1] = (float)r.instance, o[2] = (float)r.sdf;
    } break;
    case VPT_KAT_SDF_NORMAL: {
      int idx = (int)a[1];
      f3  nrm = (int)a[0] == 0 ? eval_sdf_normal_grid(sc, scene_sdf_recs(sc), idx, ld3(a + 2), a[5]) : eval_sdf_normal_function(scene_sdf_recs(sc), idx, ld3(a + 2), a[5]);
      kat_put3(o, nrm);
    } break;
    case VPT_KAT_SPHERETRACE: {
      int sdf = (int)a[6];
      if (sdf < 0) {   // the whole-scene form as K2 runs it: scene_march_step until the march ends
        float t = VPT_RAY_EPS;
        int   it = 0, inst = -1, fn = -1, mode;
        do mode = scene_march_step(sc, scene_sdf_recs(sc), ld3(a), ld3(a + 3), iparam, t, it, inst, fn);
        while (mode == M_SCENE);
        o[0] = mode == M_HIT ? 1.0f : 0.0f, o[1] = mode == M_HIT ? t : VPT_FLT_MAX, o[2] = (float)inst, o[3] = (float)fn;
      } else {         // the single-SDF form (K1's SDF-light pdf; K2's is covered by VPT_KAT_LIGHTS_PDF_K2)
        st_hit h = spheretrace_one(sc, ld3(a), ld3(a + 3), sdf, iparam);
        o[0] = h.hit ? 1.0f : 0.0f, o[1] = h.dist, o[2] = (float)h.instance, o[3] = (float)h.sdf;
      }
    } break;
    case VPT_KAT_VOLUME: o[0] = eval_volume(sc, sc.volumes[(int)a[0]], ld3(a + 1)); break;
    case VPT_KAT_SDF_FUNCTION: o[0] = sdf_fn_local(sc.sdf_fn_rec + 6 * (int)a[0], ld3(a + 1)); break;
    default: break;
  }
}
