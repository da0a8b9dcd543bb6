/* include/vpt.h — C-ABI of the MI355X volumetric path-tracing integrator (libvpt_hip.so).
 *
 * This is the drop-in boundary for ONE function of the reference:
 *
 *   void pathtrace_samples(pathtrace_state&, const scene_data&, const bvh_scene&,
 *                          const pathtrace_lights&, const pathtrace_params&)
 *                                      libs/yocto_pathtrace/yocto_pathtrace.h:133-135
 *                                      libs/yocto_pathtrace/yocto_pathtrace.cpp:1052-1092
 *
 * The reference has no FFI of its own: its seam is that C++ free
 * function, whose arguments are STL containers.  The C-ABI below is what a maintainer binds
 * behind it: plain pointers + counts to the *flattened* forms of the same four inputs
 * (INTEGRATION.md shows the ~80-line flattening stub for the reference tree).
 *
 * Conventions
 *  - every struct is little-endian POD with the reference's field order where one exists;
 *  - indices are int32, -1 (VPT_INVALID) == reference `invalidid`;
 *  - frames are 12 floats, column layout x,y,z,o (yocto_math.h:1099-1107);
 *  - nothing here owns caller memory: vpt_scene_create() copies everything to the device;
 *  - all entry points return 0 on success, a negative vpt_status otherwise, and never throw;
 *    the message for the last failure on the calling thread is vpt_last_error().
 */
#ifndef VPT_H_
#define VPT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPT_INVALID (-1)

typedef enum vpt_status {
  VPT_OK                = 0,
  VPT_ERR_INVALID_ARG   = -1, /* null pointer, bad size, index out of range                */
  VPT_ERR_NO_DEVICE     = -2, /* no gfx950 device / HIP runtime unavailable                */
  VPT_ERR_HIP           = -3, /* a HIP call failed (message has hipGetErrorString)         */
  VPT_ERR_UNKNOWN_SHADER = -4, /* reference: get_shader throws "sampler unknown" (cpp:947) */
  VPT_ERR_UNSUPPORTED   = -5  /* scene feature outside the hot-path scope (points/lines)   */
} vpt_status;

/* pathtrace_shader_type, yocto_pathtrace.h:74-84 (same order, same names) */
typedef enum vpt_shader {
  VPT_SHADER_VOLPATHTRACE    = 0,
  VPT_SHADER_PATHTRACE       = 1,
  VPT_SHADER_NAIVE           = 2,
  VPT_SHADER_EYELIGHT        = 3,
  VPT_SHADER_NORMAL          = 4,
  VPT_SHADER_TEXCOORD        = 5,
  VPT_SHADER_COLOR           = 6,
  VPT_SHADER_IMPLICIT        = 7,
  VPT_SHADER_IMPLICIT_NORMAL = 8
} vpt_shader;

/* material_type, yocto_scene.h:105-110 */
typedef enum vpt_material_type {
  VPT_MAT_MATTE = 0, VPT_MAT_GLOSSY = 1, VPT_MAT_REFLECTIVE = 2, VPT_MAT_TRANSPARENT = 3,
  VPT_MAT_REFRACTIVE = 4, VPT_MAT_SUBSURFACE = 5, VPT_MAT_VOLUMETRIC = 6, VPT_MAT_GLTFPBR = 7
} vpt_material_type;

/* sdf_type, yocto_sdfs.h:23 */
typedef enum vpt_sdf_type {
  VPT_SDF_BBOX = 0, VPT_SDF_BOX = 1, VPT_SDF_CAPPED_CONE = 2, VPT_SDF_PLANE = 3,
  VPT_SDF_SPHERE = 4, VPT_SDF_TORUS = 5
} vpt_sdf_type;

/* frame3f, yocto_math.h:1099-1107 */
typedef struct vpt_frame { float x[3], y[3], z[3], o[3]; } vpt_frame;

/* camera_data, yocto_scene.h:84-92 */
typedef struct vpt_camera {
  vpt_frame frame;
  int32_t   orthographic;
  float     lens, film, aspect, focus, aperture;
} vpt_camera;

/* bvh_node, yocto_bvh.h:73-79 — identical 32-byte layout */
typedef struct vpt_bvh_node {
  float   bbox_min[3], bbox_max[3];
  int32_t start;    /* first child (internal) or first slot in the primitive array (leaf) */
  int16_t num;      /* 2 (internal) or #primitives (leaf, <= 4)                            */
  int8_t  axis;     /* split axis                                                          */
  uint8_t internal; /* bool                                                                */
} vpt_bvh_node;

/* shape_data, yocto_shape.h:74-87 — offsets into the pooled vertex / element arrays.
 * Element indices stay shape-local (add *_offset when fetching).  Exactly one of
 * num_triangles / num_quads is non-zero on the hot path; points/lines are out of scope. */
typedef struct vpt_shape {
  int32_t num_vertices;
  int32_t position_offset;  /* into positions[] (float3 units)                */
  int32_t normal_offset;    /* into normals[]   (float3 units), -1 if absent  */
  int32_t texcoord_offset;  /* into texcoords[] (float2 units), -1 if absent  */
  int32_t color_offset;     /* into colors[]    (float4 units), -1 if absent  */
  int32_t num_triangles, triangle_offset; /* into triangles[] (int3 units)    */
  int32_t num_quads, quad_offset;         /* into quads[]     (int4 units)    */
  int32_t num_bvh_nodes, bvh_node_offset; /* into shape_bvh_nodes[]           */
  int32_t bvh_prim_offset;                /* into shape_bvh_prims[]           */
} vpt_shape;

/* instance_data, yocto_scene.h:143-149 */
typedef struct vpt_instance {
  vpt_frame frame;
  int32_t   shape;
  int32_t   material;
} vpt_instance;

/* material_data, yocto_scene.h:121-140 */
typedef struct vpt_material {
  int32_t type;
  float   emission[3];
  float   color[3];
  float   roughness, metallic, ior;
  float   scattering[3];
  float   scanisotropy, trdepth, opacity;
  int32_t emission_tex, color_tex, roughness_tex, scattering_tex, normal_tex;
} vpt_material;

/* texture_data, yocto_scene.h:96-102 — pixels live in one of two pools */
typedef struct vpt_texture {
  int32_t width, height;
  int32_t linear;   /* texture_data::linear                                     */
  int32_t is_float; /* 1: pixelsf (float4 pool), 0: pixelsb (uchar4 pool)       */
  int64_t offset;   /* first texel, in texels, inside its pool                  */
} vpt_texture;

/* environment_data, yocto_scene.h:152-157 */
typedef struct vpt_environment {
  vpt_frame frame;
  float     emission[3];
  int32_t   emission_tex;
} vpt_environment;

/* volume<float>, yocto_scene.h:203-212 */
typedef struct vpt_volume {
  int32_t whd[3];
  float   res;
  int64_t offset; /* first voxel inside voxels[]; index x + y*W + z*W*H */
} vpt_volume;

/* volume_instance, yocto_scene.h:214-219 */
typedef struct vpt_volume_instance {
  vpt_frame frame;
  int32_t   volume;
  int32_t   material;
  float     scalef;
} vpt_volume_instance;

/* sdf_data, yocto_scene.h:194-200.  The reference stores a std::function built in
 * yocto_sceneio.cpp:3684-3730; here it is a tagged union:
 *   BBOX        p = {thickness, w, h, d}   sd_bbox(p, {w,h,d}, thickness)
 *   BOX         uses whd                   sd_box(p - whd/2, whd/2)
 *   CAPPED_CONE p = {height, r1, r2}
 *   PLANE       —
 *   SPHERE      p = {radius}
 *   TORUS       p = {r1, r2}
 * `whd` is sdf_data::whd (only set for BOX; used by the SDF light sampling). */
typedef struct vpt_sdf {
  vpt_frame frame;
  int32_t   type;
  int32_t   material;
  float     whd[3];
  float     p[4];
} vpt_sdf;

/* pathtrace_light, yocto_pathtrace.h:106-111 */
typedef struct vpt_light {
  int32_t instance, environment, sdf;
  int32_t cdf_len;
  int64_t cdf_offset; /* into light_cdf[] */
} vpt_light;

/* The flattened (scene_data, bvh_scene, pathtrace_lights) triple. */
typedef struct vpt_scene_desc {
  int32_t num_cameras;       const vpt_camera*          cameras;
  int32_t num_instances;     const vpt_instance*        instances;
  int32_t num_shapes;        const vpt_shape*           shapes;
  int32_t num_materials;     const vpt_material*        materials;
  int32_t num_textures;      const vpt_texture*         textures;
  int32_t num_environments;  const vpt_environment*     environments;
  int32_t num_volumes;       const vpt_volume*          volumes;
  int32_t num_vol_instances; const vpt_volume_instance* vol_instances;
  int32_t num_sdfs;          const vpt_sdf*             sdfs;
  int32_t num_lights;        const vpt_light*           lights;

  /* pooled vertex / element data (counts in elements of the stated unit) */
  int64_t num_positions;  const float*   positions;  /* float3 */
  int64_t num_normals;    const float*   normals;    /* float3 */
  int64_t num_texcoords;  const float*   texcoords;  /* float2 */
  int64_t num_colors;     const float*   colors;     /* float4 */
  int64_t num_triangles;  const int32_t* triangles;  /* int3   */
  int64_t num_quads;      const int32_t* quads;      /* int4   */

  /* texture / voxel / cdf pools */
  int64_t num_texels_f;   const float*   texels_f;   /* float4 */
  int64_t num_texels_b;   const uint8_t* texels_b;   /* uchar4 */
  int64_t num_voxels;     const float*   voxels;
  int64_t num_light_cdf;  const float*   light_cdf;

  /* two-level BVH, bvh_data yocto_bvh.h:87-92 */
  int32_t num_scene_bvh_nodes;  const vpt_bvh_node* scene_bvh_nodes;
  int32_t num_scene_bvh_prims;  const int32_t*      scene_bvh_prims; /* instance ids */
  int64_t num_shape_bvh_nodes;  const vpt_bvh_node* shape_bvh_nodes; /* pooled */
  int64_t num_shape_bvh_prims;  const int32_t*      shape_bvh_prims; /* pooled, element ids */
} vpt_scene_desc;

/* pathtrace_params, yocto_pathtrace.h:87-99 (fields the path reads) */
typedef struct vpt_params {
  int32_t camera;
  int32_t resolution;
  int32_t shader;   /* vpt_shader */
  int32_t samples;  /* total samples requested: the call is a no-op once reached (cpp:1055);
                       ==1 selects the pixel-centre preview branch (cpp:1059-1068)          */
  int32_t bounces;
  int32_t noparallel;          /* accepted, ignored (one lane per pixel)                    */
  int32_t noimplicit_mis;
  int32_t spheretrace_maxiter;
} vpt_params;

/* Opaque device-side scene.  NOT thread-safe: a handle carries the launch schedule and staging buffers of its last
 * call, so calls on ONE handle must not overlap (the reference serialises its own calls the same way);
 * different handles may be used from different threads.  Launches on one handle may move between streams: the schedule
 * tables written on the previous launch's stream are waited for. */
typedef struct vpt_scene vpt_scene;

/* How pixels are laid out in device-resident state and shared between GPUs.
 * The image is cut into tile_w x tile_h pixel tiles (row-major tile order); tile t belongs
 * to rank (t % nranks) and is that rank's local tile (t / nranks).  A rank's state arrays are
 * TILE-MAJOR and compact: local index l = local_tile * (tile_w*tile_h) + (py*tile_w + px).
 * Slots whose pixel falls outside the image are padding and never touched. */
typedef struct vpt_layout {
  int32_t width, height;
  int32_t tile_w, tile_h; /* tile_w*tile_h must be a multiple of 64 (one wave64 per 8x8) */
  int32_t rank, nranks;
} vpt_layout;

/* ---- queries ----------------------------------------------------------------------- */
int         vpt_device_count(void);
const char* vpt_last_error(void);
const char* vpt_version(void);

/* ---- scene ------------------------------------------------------------------------- */
/* Validates every index/offset in `desc` against its pool, precomputes
 * inverse(frame, non_rigid=true) per instance with the reference's adjoint/determinant
 * formula (yocto_math.h:2802-2808, 2948-2956), uploads to `device`.                      */
int  vpt_scene_create(const vpt_scene_desc* desc, int device, vpt_scene** out);
void vpt_scene_destroy(vpt_scene* scene);

/* ---- the drop-in for pathtrace_samples() --------------------------------------------
 * Host, row-major (idx = j*width + i) caller-owned state, exactly pathtrace_state
 * (yocto_pathtrace.h:57-64): image float4[w*h], hits int32[w*h], rng {u64 state, u64 inc}[w*h].
 * Renders min(nsamples, params->samples - *samples_io) passes; the result equals that many
 * consecutive reference calls.  *samples_io is state.samples (in/out).                    */
int vpt_render(vpt_scene* scene, const vpt_params* params, int nsamples, int width, int height,
               float* image_rgba, int32_t* hits, uint64_t* rng, int* samples_io);

/* ---- the same over several GPUs of this process ("multi-GPU fan-out is internal", §8(e)) ----------
 * One host thread per GPU; the frame's 8x8-pixel tiles are dealt round-robin (tile t -> devices[t % ndev]); every
 * device holds the whole scene and the state of its own tiles, so rendering needs no communication and the result is
 * bit-identical to vpt_render's (pixels own their RNG streams).
 *
 * Residency: the tile state (radiance sums, hit counts, PCG32 streams) STAYS ON THE DEVICES between calls
 * (the reference's pathtrace_state is the same progressive accumulator, yocto_pathtrace.h:57-64).
 *   vpt_multi_set_state   host pathtrace_state -> devices (pinned staging, one thread per GPU)
 *   vpt_multi_render      with null image/hits/rng: `nsamples` passes on the resident state, nothing is transferred;
 *                         with host pointers: the contract of vpt_render - the arrays ARE the state: they are read on
 *                         every call and current after it (yocto_pathtrace.cpp:1081-1090).  RULE: a device's part of
 *                         the upload is skipped only if the device provably holds it - the three arrays are the very
 *                         ones (same addresses, size, *samples_io > 0) the previous call on this handle downloaded
 *                         into, and a 64-bit checksum over every word of that part (taken after the download, re-taken
 *                         from the arrays now) is unchanged.  An in-place edit of any pixel, another state object, a
 *                         fresh make_state: uploaded, nothing to announce.  VPT_MULTI_RESIDENT=0 makes every call
 *                         upload everything; vpt_multi_uploaded_parts() tells what the last call did.
 *   vpt_multi_get_state   devices -> host arrays, on demand
 *   vpt_multi_get_render  get_render of the resident state, assembled on devices[0]: the float4 tile buffers travel there
 *                         over xGMI by grouped RCCL send / receive (RCCL is bound on first use, a copy the process already
 *                         carries is reused; one device never touches it unless VPT_MULTI_FORCE_RCCL=1, which sends the
 *                         buffer to itself through a one-rank communicator) or, where RCCL cannot be had, by
 *                         hipMemcpyPeerAsync; then the kernel of vpt_resolve_device.
 * Like vpt_render, a render fails with VPT_ERR_HIP if a wave of the implicit kernel gave up on its watchdog. */
typedef struct vpt_multi vpt_multi;
int  vpt_multi_create(const vpt_scene_desc* desc, const int* devices, int ndev, vpt_multi** out);
void vpt_multi_destroy(vpt_multi* m);
int  vpt_multi_device_count(const vpt_multi* m);
/* how vpt_multi_get_render moves the parts: "rccl", "peer-copy" (several devices, no RCCL) or "local" (one device) */
const char* vpt_multi_transport(const vpt_multi* m);
int  vpt_multi_set_state(vpt_multi* m, int width, int height, const float* image_rgba, const int32_t* hits,
                         const uint64_t* rng, int samples);
int  vpt_multi_get_state(vpt_multi* m, float* image_rgba, int32_t* hits, uint64_t* rng, int* samples);
/* pathtrace_samples over all GPUs of `m` (see above for null host pointers) */
int  vpt_multi_render(vpt_multi* m, const vpt_params* params, int nsamples, int width, int height,
                      float* image_rgba, int32_t* hits, uint64_t* rng, int* samples_io);
/* get_render (yocto_pathtrace.cpp:1105-1116) of the resident state: row-major float4 image * (1 / samples) into the
 * caller's host buffer, gathered and resolved on devices[0] */
int  vpt_multi_get_render(vpt_multi* m, float* image_rgba);
/* how many devices' parts the last vpt_multi_render call with host pointers uploaded (0 ... device count) */
int  vpt_multi_uploaded_parts(const vpt_multi* m);

/* ---- device-resident state (bench / multi-GPU; buffers owned by the caller, e.g. torch) */
/* number of state slots a rank needs for `layout` (multiple of tile_w*tile_h) */
int64_t vpt_layout_slots(const vpt_layout* layout);
/* host row-major <-> device tile-major (only this rank's pixels are touched) */
int vpt_state_upload(const vpt_layout* layout, const float* image_rgba, const int32_t* hits,
                     const uint64_t* rng, void* d_image, void* d_hits, void* d_rng, void* stream);
int vpt_state_download(const vpt_layout* layout, const void* d_image, const void* d_hits,
                       const void* d_rng, float* image_rgba, int32_t* hits, uint64_t* rng,
                       void* stream);
/* nsamples passes over this rank's pixels; asynchronous on `stream` (hipStream_t) - with one exception per layout: the call
 * that takes the tile-splitting decision (below: the second full call on a layout that is short of waves) waits for the previous
 * launch and reads its per-tile costs back (a few tens of ms at 3840x1600) before it enqueues.  params->samples == 1
 * selects the pixel-centre preview branch (yocto_pathtrace.cpp:1059-1068).
 * Scheduling: a wave renders all samples of its 64 pixels, so the scene handle remembers how long every
 * wave of the last launch took and starts the next launch on the same layout / camera / shader longest wave
 * first.  Without such a record and with nsamples >= 16, the first nsamples/64 (1..16) samples are rendered by a separate pilot
 * launch that takes the measurement.  Neither changes the result  (pixels are independent, batching is exact). */
int vpt_render_device(vpt_scene* scene, const vpt_params* params, const vpt_layout* layout,
                      int nsamples, void* d_image, void* d_hits, void* d_rng, void* stream);
/* get_render (yocto_pathtrace.cpp:1105-1116) on device: gathered tile-major float4 sums of ALL
 * ranks ([nranks][slots]) -> row-major float4 image * (1/samples).                          */
int vpt_resolve_device(const vpt_layout* layout, const void* d_tiles_all_ranks, int samples,
                       void* d_image_rowmajor, void* stream);

/* the same followed by the 8-bit output stage of save_image (rgb_to_srgb + float_to_byte, yocto_color.h:207-231;
 * yocto_sceneio.cpp:509-571 then hands the bytes to the encoder): row-major RGBA8 on the device, 4 B per pixel
 * to download for a preview instead of 16.  Uses the device's powf: a byte may differ by one from the host
 * routine where the curve lands within an ulp of a quantisation step (parity checks use the host routine). */
int vpt_resolve_srgb8_device(const vpt_layout* layout, const void* d_tiles_all_ranks, int samples,
                             void* d_rgba8_rowmajor, void* stream);

/* per-launch profile of the last vpt_render_device on this scene (HIP events on `stream`); synchronises with that
 * launch.  Like vpt_render it returns VPT_ERR_HIP if a wave of the implicit kernel gave up on its watchdog (a wave
 * that has not finished after 300 s leaves the kernel instead of holding the GPU: a defect, never a workload). */
int vpt_last_kernel_ms(vpt_scene* scene, float* ms);
/* the watchdog check alone (synchronous; the caller has waited for its launches): VPT_ERR_HIP if any wave of the
 * implicit kernel on this scene handle has given up since the handle was created */
int vpt_check_watchdog(vpt_scene* scene);

/* How long every wave of the last vpt_render_device launch on this scene ran, in ticks of the 100 MHz wall clock:
 * what the next launch's longest-first order is made from, exposed for load-balance analysis (critical path = the
 * largest entry, slot time = their sum).  Wave w renders state slots 64 w .. 64 w + 63 - unless the launch split
 * costly tiles into several partly filled waves (below), in which case the entries follow the split launch's waves.
 * Synchronises with the launch.  *count = waves of the launch; at most `capacity` entries are written.
 *
 * Tile splitting.  A wave runs all samples of its pixels one after the other, so a launch is at least as long as its
 * costliest tile.  When a layout shares the frame among ranks (nranks > 1) the work per GPU falls with N and that
 * chain does not, and a small frame (fewer than three tiles per wave slot of the chip) is in the same position: from the
 * second full call on such a layout the mesh kernels run the costliest tiles as 2^k waves
 * that hold every 2^k-th pixel (fewer live lanes diverge less: a wave with 8 lanes takes ~0.45 of the time), chosen
 * once from the measured per-tile costs by simulating the launch's schedule.  Results do not depend on it (pixels own
 * their RNG streams).  VPT_SPLIT=0 disables it, VPT_SPLIT=1 considers it for every layout (it never pays on a full-size frame on one GPU). */
int vpt_last_wave_costs(vpt_scene* scene, unsigned* ticks, int capacity, int* count);

/* Bytes per primitive of the records the path tracers' BVH leaf tests / shading fetch on this scene: 64 / 96 in general (four corner
 * positions; four normals + four texcoords), 48 / 64 when every shape of the scene holds triangles (three corners; three normals with the
 * texcoords in their spare words).  Such a scene keeps both forms on the device: pathtrace and volpathtrace read the short ones on scenes
 * without emissive meshes that need a BVH walk and without SDF lights, everything else reads the general ones.  Same results either way
 * (VPT_NO_COMPACT_TRIANGLES=1 at scene creation keeps the general records only: the tests' A/B switch). */
int vpt_scene_record_bytes(const vpt_scene* scene, int* leaf_bytes, int* attribute_bytes);

/* intersect_bvh(bvh, scene, ray) (instance < 0) / intersect_bvh(bvh, scene, instance, ray) of yocto_bvh.h, for a
 * batch of `n` host rays {o.xyz, d.xyz} with the reference's default tmin = 1e-4, tmax = flt_max, through the
 * kernels' own traversal.  ids[2i..] = {instance, element} (-1, -1 on a miss), uvt[3i..] = {u, v, distance}.
 * Synchronous.  The parity tests use it to compare the traversal with the reference's bit for bit on rays path
 * tracing rarely produces (axis-aligned, grazing a box plane, denormal direction components). */
int vpt_intersect(vpt_scene* scene, int n, const float* rays, int instance, int32_t* ids, float* uvt);

/* build_bvh(bvh, bboxes, highquality = false) of the reference (libs/yocto/yocto_bvh.cpp:447-507, split_middle
 * :411-441) on the device: `n` boxes {min.xyz, max.xyz} in, the reference's node array and primitive order out - the
 * same nodes under the same ids, the same primitive permutation, the same float bits (the reference's build is depth
 * first over a stack with std::partition; csrc/vpt_bvh_build.hip says how a level-parallel build arrives at the same
 * arrays).  nodes: room for `capacity` >= max(1, 2 n - 1) entries; *num_nodes = entries written; primitives: n ints.
 * Synchronous.  SURVEY §8(f) row 4 (load-time callers of the hot path). */
int vpt_build_bvh(int device, const float* bboxes, int n, vpt_bvh_node* nodes, int capacity, int* num_nodes, int32_t* primitives);

/* The float32 half of one level of the reference's Catmull-Clark subdivision (tesselate_catmullclark,
 * libs/yocto_pathtrace/yocto_pathtrace.cpp:1119-1226) on the device.  The caller supplies the level's
 * topology (integers: host/vpt_tesselate.cpp builds it as the reference does): the cage's edges in the reference's numbering,
 * its faces (a quad with z == w is a triangle), the refined faces, and per refined vertex - old vertices first, then one per
 * edge, then one per face - its kind (`valence`: 0 locked boundary, 1 creased boundary, 2 smooth) and the items whose centroids
 * the reference's averaging pass adds to it, IN THE ORDER ITS LOOPS REACH THEM (items[offsets[v] .. offsets[v + 1]): valence 0:
 * the vertex id, once per contribution; valence 1: vertex pairs (a, b) of crease edges; valence 2: refined face ids).  The
 * device computes the refined points, the averages in that order and the correction: `new_vertices` (num_vertices + num_edges
 * + num_faces entries of `dim` floats) holds the same bits as the reference's `vert` after the level.  Synchronous; every
 * index is validated on the host before a kernel runs. */
typedef struct vpt_subdiv_level {
  int32_t        dim;                                   /* floats per vertex: 3 (positions) or 2 (texcoords) */
  int32_t        num_vertices, num_edges, num_faces;    /* of the cage                                        */
  int32_t        num_new_faces;
  const int32_t* edges;                                 /* int2 [num_edges]                                   */
  const int32_t* faces;                                 /* int4 [num_faces]                                   */
  const int32_t* new_faces;                             /* int4 [num_new_faces]                               */
  const int32_t* valence;                               /* [num_vertices + num_edges + num_faces]             */
  const int32_t* offsets;                               /* [that + 1]                                         */
  const int32_t* items;
  int64_t        num_items;
} vpt_subdiv_level;
int vpt_subdivide_vertices(int device, const vpt_subdiv_level* level, const float* vertices, float* new_vertices);
/* The rest of tesselate_surface's float work on the device (yocto_pathtrace.cpp:1239, 1256-1271), same ordered-list scheme, same bits:
 * vpt_vertex_normals   quads_normals (corners = 4; a quad with z == w is a triangle and adds to three vertices) / triangles_normals
 *                      (corners = 3) of yocto_shape.cpp:1478-1512: area-weighted face normals added per vertex IN FACE ORDER, normalised;
 *                      positions / normals are float3 arrays, faces int4 / int3.  The per-vertex face lists are built on the host here.
 * vpt_displace_vertices  new_positions = positions + normals * displacement * (mean(xyz(eval_texture(texture, uv, as_linear = true)))
 *                      [- 0.5 for an 8-bit texture]) (cpp:1259-1265) through the render kernels' own eval_texture; `texels` is the
 *                      texture's first texel (uchar4 or float4 by texture->is_float; texture->offset is ignored).
 * Synchronous; indices are validated on the host before a kernel runs. */
int vpt_vertex_normals(int device, int32_t num_vertices, const float* positions, int32_t num_faces, int32_t corners, const int32_t* faces, float* normals);
int vpt_displace_vertices(int device, const vpt_texture* texture, const void* texels, float displacement, int32_t num_vertices,
                          const float* positions, const float* normals, const float* texcoords, float* new_positions);

/* Device self-test of an arithmetic shortcut the kernels rely on for bit-exact parity: the reference divides
 * (1 / d per ray, yocto_bvh.cpp:806-808; 1 / det per triangle, yocto_geometry.h:690), the kernels use
 * v_rcp_f32 + one Newton step where every lane's operand has a biased exponent in 1..250.  Runs all 2^32 bit
 * patterns on `device` and returns in *mismatches how many in-range inputs differ from the IEEE quotient
 * (must be 0), in *fallbacks how many patterns are out of range (handled by a real division). */
int vpt_selftest_reciprocal(int device, unsigned long long* mismatches, unsigned long long* fallbacks);
/* Device self-test of the search structure that replaces std::upper_bound over a light's CDF in sample_discrete
 * (yocto_sampling.h:385-390; 21 dependent probes on a 2 M-texel environment map): `n` probe values - CDF
 * entries, their float neighbours, both ends, uniform values - are looked up through the guide table / 16-ary
 * levels and through the plain binary search; *mismatches must come back 0.  *indexed: 0 the light's CDF is
 * short and uses the binary search itself, 1 16-ary levels, 2 levels + guide table. */
int vpt_selftest_light_cdf(vpt_scene* scene, int light, int n, unsigned long long* mismatches, int* indexed);

#ifdef __cplusplus
}
#endif
#endif /* VPT_H_ */
