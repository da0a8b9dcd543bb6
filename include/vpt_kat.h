/* include/vpt_kat.h — known-answer-test entry points of libvpt_hip.so.
 *
 * The hot path is pinned end to end by float32 pathtrace_state equality with the reference (tests/),
 * but whole-path equality cannot reach code no test scene reaches, and cannot tell WHICH function
 * drifted when a pixel differs.  vpt_kat() therefore runs the kernels' own device functions — the ones
 * vpt_mesh_kernel / vpt_render_kernel call, not copies — on a batch of host-supplied arguments, one lane
 * per record, and returns their results, so that each can be compared with a table produced by the
 * reference's function of the same name (oracle/ref_tables.cpp, compiled from the reference's sources;
 * tables committed under tests/golden/kat_*.npz) — SURVEY.md §8(c)(4).
 *
 * Records are float32; integers (ids, type tags, flags) travel as floats (all of them are far below
 * 2^24).  `in` holds n * in_stride floats, `out` receives n * out_stride floats.  Rays carry the reference's
 * defaults tmin = 1e-4, tmax = flt_max (ray3f, yocto_geometry.h:118-123).
 *
 *  op                     record in                                              record out                                            reference function (file:line)
 *  VPT_KAT_LOBES          type color[3] roughness metallic ior normal[3]         s_in[3] s_f[3] s_pdf  a_f[3] a_pdf                     sample_bsdfcos / eval_bsdfcos / sample_bsdfcos_pdf,
 *                         outgoing[3] rnl rn[2] alt_incoming[3]          (19)    d_in[3] d_f[3] d_pdf  ad_f[3] ad_pdf            (22)   sample_delta / eval_delta / sample_delta_pdf
 *                                                                                                                                       (yocto_pathtrace.cpp:92-236 over yocto_shading.h:543-1039).
 *                         roughness is material_point::roughness (already squared / clamped).  s_* use the sampled direction, a_* and
 *                         ad_* the given one; as in the shaders (cpp:629) nothing is evaluated for a zero sampled direction (zeros).
 *  VPT_KAT_MEDIA          density[3] max_distance rl rd g outgoing[3] rn[2]      distance pdf transmittance[3] phase s_dir[3] s_phase   sample_transmittance(_pdf), eval_transmittance,
 *                         incoming[3]                                    (15)                                                    (10)   eval/sample_phasefunction (yocto_shading.h:1047-1096)
 *  VPT_KAT_TEXTURE        texture u v as_linear                           (4)    rgba[4]                                          (4)   eval_texture (yocto_scene.cpp:128-161)
 *  VPT_KAT_CAMERA         camera u v lens_u lens_v                        (5)    o[3] d[3]                                        (6)   eval_camera (yocto_scene.cpp:67-102)
 *  VPT_KAT_INTERSECT      o[3] d[3] instance (-1: whole scene)            (7)    instance element u v distance (-1 -1 0 0 0: miss)  (5)   intersect_bvh (yocto_bvh.cpp:1097-1113)
 *  VPT_KAT_SURFACE        instance element u v outgoing[3]                (7)    position[3] normal[3] type emission[3] color[3]        eval_shading_position / eval_shading_normal /
 *                                                                                opacity roughness metallic ior density[3]              eval_material (yocto_scene.cpp:460-579)
 *                                                                                scattering[3] scanisotropy                      (24)
 *  VPT_KAT_ENVIRONMENT    direction[3]                                    (3)    rgb[3]                                           (3)   eval_environment (yocto_scene.cpp:634-651)
 *  VPT_KAT_SAMPLE_LIGHTS  position[3] rl rel ruv[2]                       (7)    direction[3]                                     (3)   sample_lights (yocto_pathtrace.cpp:312-350)
 *  VPT_KAT_LIGHTS_PDF     position[3] direction[3]                        (6)    pdf                                              (1)   sample_lights_pdf (yocto_pathtrace.cpp:353-421); iparam =
 *                                                                                                                                       spheretrace_maxiter.  K1's code path (light records, inline
 *                                                                                                                                       single-leaf walks, quad-node hops)
 *  VPT_KAT_LIGHTS_PDF_K2  same                                                   same                                                   same function through K2's code path (vpt_scene.hip.h)
 *  VPT_KAT_SDF_SCENE      p[3] t                                          (4)    result instance sdf                              (3)   eval_sdf_scene (yocto_sdfs.cpp:7-26)
 *  VPT_KAT_SDF_NORMAL     kind (0: vol_instance, 1: sdf) index p[3] t     (6)    normal[3]                                        (3)   eval_sdf_normal (yocto_sdfs.cpp:67-89)
 *  VPT_KAT_SPHERETRACE    o[3] d[3] sdf (-1: whole scene)                 (7)    hit dist instance sdf (miss: 0 flt_max -1 -1)    (4)   spheretrace (yocto_pathtrace.cpp:267-307); iparam = maxiter
 *  VPT_KAT_VOLUME         volume uvw[3]                                   (4)    value                                            (1)   eval_volume (yocto_sdfs.cpp:92-127)
 *  VPT_KAT_SDF_FUNCTION   sdf p[3] (in the sdf's local frame)             (4)    distance                                         (1)   sdf_data::f, i.e. sd_* of yocto_sdfs.h:43-80 as bound by
 *                                                                                                                                       yocto_sceneio.cpp:3684-3730
 */
#ifndef VPT_KAT_H_
#define VPT_KAT_H_

#include "vpt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vpt_kat_op {
  VPT_KAT_LOBES = 0, VPT_KAT_MEDIA = 1, VPT_KAT_TEXTURE = 2, VPT_KAT_CAMERA = 3, VPT_KAT_INTERSECT = 4,
  VPT_KAT_SURFACE = 5, VPT_KAT_ENVIRONMENT = 6, VPT_KAT_SAMPLE_LIGHTS = 7, VPT_KAT_LIGHTS_PDF = 8,
  VPT_KAT_LIGHTS_PDF_K2 = 9, VPT_KAT_SDF_SCENE = 10, VPT_KAT_SDF_NORMAL = 11, VPT_KAT_SPHERETRACE = 12,
  VPT_KAT_VOLUME = 13, VPT_KAT_SDF_FUNCTION = 14, VPT_KAT_OP_COUNT = 15
} vpt_kat_op;

/* floats per input / output record of `op`; returns VPT_ERR_INVALID_ARG for an unknown op */
int vpt_kat_strides(int op, int* in_stride, int* out_stride);

/* Runs `op` on n records on the scene's device (synchronous).  Scene-free ops (LOBES, MEDIA) ignore the
 * scene's content but still need a handle for the device.  Ids inside records are range-checked on the
 * host first (VPT_ERR_INVALID_ARG): a KAT batch can never index outside the scene's tables. */
int vpt_kat(vpt_scene* scene, int op, int iparam, int n, const float* in, float* out);

/* Named forms of two of the ops (what a maintainer would reach for first). */
/* spheretrace(scene, ray, maxiter) (sdf < 0) / spheretrace(scene, ray, sdf, maxiter) for n rays {o, d}:
 * ids[3i..] = {hit, instance, sdf}, t[i] = distance (flt_max on a miss). */
int vpt_spheretrace(vpt_scene* scene, int n, const float* rays, int sdf, int maxiter, int32_t* ids, float* t);
/* VPT_KAT_LOBES with its record layout. */
int vpt_eval_lobes(vpt_scene* scene, int n, const float* in19, float* out22);

#ifdef __cplusplus
}
#endif
#endif /* VPT_KAT_H_ */
