// vpt_device.h — device-resident scene layout shared by the C-ABI host code (vpt_capi.hip)
// and the kernels (vpt_kernels.hip.inc).  All arrays live in HBM for the life of a vpt_scene;
// layouts are chosen for 16-byte vector loads (global_load_dwordx4) per lane:
//
//   bvh node      32 B  = 2 x float4   {min.xyz, max.x} {max.yz, start, num|axis<<16|internal<<24}
//   leaf record   64 B  = 4 x float4   the 4 corner positions of one quad (a triangle repeats its
//                                      last corner) stored IN BVH LEAF ORDER, element id in p0.w:
//                                      a leaf's <=4 primitives are one contiguous <=256 B run, and
//                                      the reference's prims[] -> quads[] -> positions[] double
//                                      indirection (yocto_bvh.cpp:780-789) is gone from traversal
//   instance     128 B  = 8 x float4   inverse frame (3x4) first: it is what traversal reads
//   vertex attrs       float4 positions / float4 normals / float2 texcoords / float4 colors
//   textures           uchar4 or float4 texels; sRGB->linear through a 256-entry float LUT computed
//                      on the host with the same powf the reference calls per fetch
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vpt.h"

enum { VPT_SHP_TRIANGLES = 1, VPT_SHP_NORMALS = 2, VPT_SHP_TEXCOORDS = 4, VPT_SHP_COLORS = 8 };

struct DInstance {       // 128 B
  float4 inv[3];         // inverse(frame, non_rigid=true): rows packed as x,y,z columns + o: see pack
  float4 fwd[3];
  int    shape, material;
  int    translation_only;   // rotation part is exactly the identity: local direction == world direction
  int    shape_flags;        // VPT_SHP_* of its shape: shading then needs no DShape fetch
  int    pad[4];
};
// A frame {x,y,z,o} (4 columns of 3) packed in 3 float4: {x.x,x.y,x.z,y.x} {y.y,y.z,z.x,z.y} {z.z,o.x,o.y,o.z}

struct DShape {          // 80 B
  int num_nodes, node_offset;  // into shape_nodes (in nodes)
  int leaf_offset;             // into leaf_prims (in records), slot = leaf_offset + node.start + k
  int is_triangles;            // shape.triangles non-empty (reference tests triangles first)
  int elem_offset;             // into elems (int4 per element; triangles repeat z in w)
  int vertex_offset;           // into positions
  int normal_offset, texcoord_offset, color_offset;  // -1 if absent
  int num_elems;
  int stack_need;              // max traversal stack depth for this shape's BVH
  int root_ref;                // wide-node reference of the BVH root (see DScene::shape_wnodes)
  int wnode_offset;            // into shape_wnodes (in wide nodes)
  float root_box[6];           // bbox of the root: the reference tests it at the first pop
  int pad2;
};

// 16-ary index over one light's CDF.  Level 0 is a copy of the CDF, level k+1 holds the last element of
// every group of 16 of level k (its maximum: the CDF is non-decreasing); every level is padded to a multiple
// of 16 with +inf and the top level has <= 16 entries.  std::upper_bound (what sample_discrete does,
// yocto_sampling.h:385-390) is unique on sorted data, so a top-down search over the levels returns the
// reference's index with one 64-byte fetch per level instead of one dependent probe per bit (2 M-entry
// environment CDF: 6 vs 21).
enum { VPT_LIGHT_SMALL_MESH = 0, VPT_LIGHT_LARGE_MESH = 1, VPT_LIGHT_ENV_TEX = 2, VPT_LIGHT_ENV_CONST = 3, VPT_LIGHT_SDF = 4, VPT_LIGHT_NONE = 5 };

// On top of it a guide table (the "cutpoint" method): the clamped sample r falls into bucket
// b = min(int(r * guide_scale), guide_buckets - 1); light_guide[guide_offset + b] = {lo, hi} brackets
// upper_bound for every r of that bucket (bounds widened on the host by more than the rounding of
// r * guide_scale).  When hi - lo <= 16 the answer is lo + #{cdf[lo .. lo+15] <= r}: two dependent fetches
// instead of six; longer brackets (dark stretches of an environment map) take the 16-ary levels.
struct DCdfIndex {
  int   levels;        // 0: no index (short or non-monotone CDF: plain binary search); else number of levels incl. level 0
  int   top_count;     // valid entries of the top level
  int   offset[8];     // offset[k]: start of level k in light_index_pool
  int   guide_offset, guide_buckets;
  float guide_scale;
  int   pad;
};

struct DScene {
  // counts
  int num_cameras, num_instances, num_shapes, num_materials, num_textures, num_environments;
  int num_volumes, num_vol_instances, num_sdfs, num_lights, num_scene_nodes, num_scene_prims;
  // bvh
  const float4* scene_nodes;   // 2 per node
  const int*    scene_prims;
  const float4* shape_nodes;   // 2 per node, pooled
  const float4* leaf_prims;    // 4 per slot, pooled
  // "wide" nodes for the unified traversal: one 64-byte record per INTERNAL node holding BOTH child
  // boxes and child references, so a visit costs one fetch and culled children are never fetched.
  //   q0 = {L.min.xyz, L.max.x} q1 = {L.max.yz, R.min.xy} q2 = {R.min.z, R.max.xyz}
  //   q3 = {ref0, ref1, axis, -} (ints); ref >= 0: internal wide node; ref < 0: leaf, ~ref = start<<4 | num
  const float4* scene_wnodes;   // ONE array: the scene's quad nodes, then the shapes' (traverse() names a level by its first node's index)
  const float4* shape_wnodes;   // = scene_wnodes + 8 * (number of scene quad nodes)
  // "enter records": everything the traversal needs to enter the instance stored at a scene-BVH
  // primitive slot, in one 96-byte gather instead of the prims[] -> instances[] -> shapes[] chain:
  //   e0..e2 = inverse frame (packed), e3 = {root lo.xyz, root hi.x}, e4 = {root hi.y, root hi.z,
  //   root_ref, first quad node of the shape in scene_wnodes}, e5 = {leaf_offset, instance id, translation_only, num_nodes} (ints)
  const float4* scene_enter;      // 6 per scene-BVH primitive slot
  const int*    slot_of_instance; // instance id -> slot (single-instance queries)
  int   scene_root_ref;
  int   group_forms;   // 1: phases with few rays run on four lanes per ray (traverse(), K2's scene march); 0 (VPT_NO_GROUP_FORMS=1): own forms only - same bits (tests)
  float scene_root_lo_x, scene_root_lo_y, scene_root_lo_z, scene_root_hi_x, scene_root_hi_y, scene_root_hi_z;
  // geometry
  const DInstance* instances;
  const DShape*    shapes;
  const int4*      elems;
  const float4*    positions;
  const float4*    normals;
  const float2*    texcoords;
  const float4*    colors;
  // appearance
  const vpt_material*    materials;
  const vpt_texture*     textures;
  const float4*          texels_f;
  const uchar4*          texels_b;
  const float*           srgb_lut;      // 256 entries
  const vpt_environment* environments;
  const float4*          env_inv;       // 3 float4 per environment: inverse(frame) (rigid)
  // lights
  const vpt_light* lights;
  const float*     light_cdf;
  // Light records: what sample_lights_pdf needs per light, behind ONE index (light id) instead of the chain
  // lights[] -> instances[] -> shapes[] -> leaf_prims[] -> elems[] -> positions[].  light_rec: 8 float4 per light:
  // [0..2] inverse frame of the instance (mesh) / of the environment, [3..5] forward frame (mesh / environment),
  // [6] = {root box lo.xyz, area} (mesh) or {tex width, tex height (as int bits), cdf total, 0} (environment),
  // [7] = {root box hi.xyz, kind | count << 8} with kind = VPT_LIGHT_*.  light_prims: for single-leaf mesh lights,
  // 4 x 5 float4 per light: the leaf's primitives as corner positions (element id in p0.w) + the element's
  // world-space normal (eval_element_normal), computed on the device at scene creation.
  // Vertex attributes per primitive slot, parallel to leaf_prims: 6 float4 = the four corners' normals, then
  // their texcoords as 4 x float2.  A hit carries its slot, so position, normal and texcoord of the shading
  // point are one fetch level away instead of instances[] -> shapes[] -> elems[] -> positions/normals/texcoords[].
  const float4*    leaf_attrs;
  const float4*    light_rec;
  const float4*    light_prims;
  const DCdfIndex* light_index;      // per light: 16-ary search index over its CDF (levels == 0: plain binary search)
  const float*     light_index_pool;
  const int2*      light_guide;
  // implicit surfaces
  const vpt_volume*          volumes;
  const float*               voxels;
  const vpt_volume_instance* vol_instances;
  const vpt_sdf*             sdfs;
  const float4*              sdf_inv;   // 3 float4 per sdf: inverse(frame) (rigid), for light sampling
  // SDF evaluation records (layout: vpt_scene.hip.h, "SDF records"): 6 float4 per analytic SDF, 7 per voxel-grid instance
  const float4*              sdf_fn_rec;
  const float4*              sdf_grid_rec;
  // ball around everything bounded the SDF shaders can hit (grids' boxes, all analytic SDFs but planes), world space;
  // sdf_bound_r <= 0: no early-out for escaping rays.  sdf_num_planes: analytic SDFs of unbounded type
  float sdf_bound_cx, sdf_bound_cy, sdf_bound_cz, sdf_bound_r;
  int   sdf_num_planes, pad3;
  const vpt_camera*          cameras;
  // Compact records of a scene whose shapes all hold triangles (else null), beside the general ones above and in the same slot order:
  // tri_prims: 3 float4 per slot = the three corners, element id in p0.w (48 bytes against 64);
  // tri_attrs: 4 float4 per slot = {n0, t0.x} {n1, t0.y} {n2, t1.x} {t1.y, t2.x, t2.y, -} (64 bytes against 96).
  // Read by the kernel instances compiled for them (VPT_FEAT_COMPACT_TRIS, vpt_scene.hip.h); a layout known at compile time costs the
  // instances for quads nothing, the same choice taken at run time cost them 7 % (profiles/r04_record_layout.txt)
  const float4* tri_prims;
  const float4* tri_attrs;
};

struct DParams {
  int   camera, shader, bounces, noimplicit_mis, spheretrace_maxiter;
  int   preview;       // params.samples == 1 branch (yocto_pathtrace.cpp:1059-1068)
  int   nsamples;      // passes to render in this launch
  // layout
  int   width, height, tile_w, tile_h, tiles_x, tiles_y, rank, nranks;
  int   nslots;        // state slots of this rank (multiple of 64)
};
