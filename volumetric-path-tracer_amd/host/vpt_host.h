// vpt_host.h — host-side mirror of the reference's renderer interface
// (libs/yocto_pathtrace/yocto_pathtrace.h:57-139): same type and function names, same
// argument meaning, same error behaviour, so a caller of the reference can switch by
// changing the namespace.  Everything that is *hot* (pathtrace_samples) goes through the
// C-ABI in include/vpt.h to the HIP kernels; everything here is load-time host work:
// scene containers, flattening, BVH build (yocto_bvh.cpp:411-611), light CDFs
// (yocto_pathtrace.cpp:983-1049) and per-pixel PCG32 seeding (yocto_pathtrace.cpp:960-980).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "vpt.h"

namespace vpt {

using std::string;
using std::vector;

inline const int invalidid = -1;

// ---- small POD math (only what load-time code needs; no operator zoo) ----------------------
struct vec2f { float x = 0, y = 0; };
struct vec3f { float x = 0, y = 0, z = 0; };
struct vec4f { float x = 0, y = 0, z = 0, w = 0; };
struct vec3i { int x = 0, y = 0, z = 0; };
struct vec4i { int x = 0, y = 0, z = 0, w = 0; };
struct vec4b { uint8_t x = 0, y = 0, z = 0, w = 0; };
struct frame3f {
  vec3f x = {1, 0, 0}, y = {0, 1, 0}, z = {0, 0, 1}, o = {0, 0, 0};
};
static_assert(sizeof(frame3f) == sizeof(vpt_frame), "frame layout");

// ---- scene containers (yocto_scene.h:84-250, yocto_shape.h:74-87) ---------------------------
struct camera_data {
  frame3f frame        = {};
  bool    orthographic = false;
  float   lens = 0.050f, film = 0.036f, aspect = 1.500f, focus = 10000, aperture = 0;
};
struct texture_data {
  int           width = 0, height = 0;
  bool          linear  = false;
  vector<vec4f> pixelsf = {};
  vector<vec4b> pixelsb = {};
};
enum struct material_type { matte, glossy, reflective, transparent, refractive, subsurface, volumetric, gltfpbr };
inline const auto material_type_names = vector<string>{"matte", "glossy", "reflective",
    "transparent", "refractive", "subsurface", "volumetric", "gltfpbr"};
struct material_data {
  material_type type = material_type::matte;
  vec3f emission = {0, 0, 0}, color = {0, 0, 0};
  float roughness = 0, metallic = 0, ior = 1.5f;
  vec3f scattering   = {0, 0, 0};
  float scanisotropy = 0, trdepth = 0.01f, opacity = 1;
  int   emission_tex = invalidid, color_tex = invalidid, roughness_tex = invalidid,
      scattering_tex = invalidid, normal_tex = invalidid;
};
struct shape_data {
  vector<int>   points    = {};  // out of hot-path scope: load keeps them, flatten rejects them
  vector<vec3i> triangles = {};
  vector<vec4i> quads     = {};
  vector<vec3f> positions = {};
  vector<vec3f> normals   = {};
  vector<vec2f> texcoords = {};
  vector<vec4f> colors    = {};
};
struct instance_data {
  frame3f frame = {};
  int     shape = invalidid, material = invalidid;
};
struct environment_data {
  frame3f frame        = {};
  vec3f   emission     = {0, 0, 0};
  int     emission_tex = invalidid;
};
enum struct sdf_type { bbox, box, capped_cone, plane, sphere, torus };
inline const auto sdf_type_names = vector<string>{"bbox", "box", "capped_cone", "plane", "sphere", "torus"};
struct sdf_data {  // tagged union instead of std::function (yocto_scene.h:194-200)
  int      material = invalidid;
  frame3f  frame    = {};
  vec3f    whd      = {0, 0, 0};
  sdf_type type     = sdf_type::plane;
  float    p[4]     = {0, 0, 0, 0};
};
struct volume_data {  // volume<float>
  vec3i         whd = {0, 0, 0};
  vector<float> vol = {};
  float         res = 0;
};
struct volume_instance {
  int     volume = invalidid, material = invalidid;
  float   scalef = 1;
  frame3f frame  = {};
};
struct subdiv_data {  // yocto_scene.h:161-183: face-varying control cage + subdivision / displacement settings
  vector<vec4i> quadspos = {}, quadsnorm = {}, quadstexcoord = {};
  vector<vec3f> positions = {}, normals = {};
  vector<vec2f> texcoords = {};
  int   subdivisions = 0;
  bool  catmullclark = true, smooth = true;
  float displacement     = 0;
  int   displacement_tex = invalidid;
  int   shape            = invalidid;
};
struct scene_data {
  vector<camera_data>      cameras       = {};
  vector<instance_data>    instances     = {};
  vector<environment_data> environments  = {};
  vector<shape_data>       shapes        = {};
  vector<texture_data>     textures      = {};
  vector<material_data>    materials     = {};
  vector<volume_data>      volumes       = {};
  vector<volume_instance>  vol_instances = {};
  vector<sdf_data>         sdfs          = {};
  vector<subdiv_data>      subdivs       = {};
  string                   copyright     = "";
};

// ---- bvh (yocto_bvh.h:73-92) ----------------------------------------------------------------
using bvh_node = vpt_bvh_node;
struct bvh_data {
  vector<bvh_node> nodes      = {};
  vector<int>      primitives = {};
  vector<bvh_data> shapes     = {};
};
using bvh_scene = bvh_data;

// ---- renderer API (yocto_pathtrace.h:57-139) ------------------------------------------------
struct rng_state {
  uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
};
struct pathtrace_state {
  int               width = 0, height = 0, samples = 0;
  vector<vec4f>     image = {};
  vector<int>       hits  = {};
  vector<rng_state> rngs  = {};
};
enum struct pathtrace_shader_type { volpathtrace, pathtrace, naive, eyelight, normal, texcoord, color, implicit, implicit_normal };
inline const auto pathtrace_shader_names = vector<string>{"volpathtrace", "pathtrace", "naive",
    "eyelight", "normal", "texcoord", "color", "implicit", "implicit_normal"};
struct pathtrace_params {
  int                   camera              = 0;
  int                   resolution          = 720;
  pathtrace_shader_type shader              = pathtrace_shader_type::pathtrace;
  int                   samples             = 512;
  int                   bounces             = 4;
  bool                  noparallel          = false;
  int                   pratio              = 8;
  float                 exposure            = 0;
  bool                  filmic              = false;
  bool                  noimplicit_mis      = false;
  int                   spheretrace_maxiter = 450;
};
struct pathtrace_light {
  int           instance = invalidid, environment = invalidid, sdf = invalidid;
  vector<float> elements_cdf = {};
};
struct pathtrace_lights {
  vector<pathtrace_light> lights = {};
};
struct color_image {
  int           width = 0, height = 0;
  bool          linear = true;
  vector<vec4f> pixels = {};
};

pathtrace_state  make_state(const scene_data& scene, const pathtrace_params& params);
bvh_scene        make_bvh(const scene_data& scene, const pathtrace_params& params);
// Extension: the same trees (node for node, hash-equal) built on GPU `device` by vpt_build_bvh; throws if that fails.
bvh_scene        make_bvh_device(const scene_data& scene, const pathtrace_params& params, int device = 0);
// build_bvh over `n` boxes {min.xyz, max.xyz} on the host (what make_bvh runs per shape and for the instances)
bvh_data         build_bvh_host(const float* bboxes, int n);
pathtrace_lights make_lights(const scene_data& scene, const pathtrace_params& params);
// tesselate_surfaces (yocto_pathtrace.cpp:1119-1280): Catmull-Clark subdivision of every subdiv's cage (positions with
// creased boundaries, texcoords with locked boundaries), split_facevarying, quads_to_triangles, displacement along the
// vertex normals by the mean of the displacement texture (- 0.5 for 8-bit ones), smooth normals; replaces the subdiv's shape.
// Host arithmetic in the reference's order: the same float bits (tests/golden/*_stats.json hold the reference's hashes).
void             tesselate_surfaces(scene_data& scene);
// one level of the reference's tesselate_catmullclark (cpp:1119-1226) on a quad mesh whose vertices have `dim` (2 or 3)
// floats; a quad with z == w is a triangle.  quads / verts are replaced by the next level's.
void             tesselate_catmullclark(vector<vec4i>& quads, vector<float>& verts, int dim, bool lock_boundary);
// The two halves of a level.  Topology (integers): the refined faces and, per refined vertex, the ordered list of items the
// reference's averaging pass adds to it (layout: vpt_subdiv_level of include/vpt.h).  Vertices (float32): host form.
struct subdiv_level {
  int           nv = 0, ne = 0, nf = 0;
  vector<int>   edges;    // 2 per edge
  vector<vec4i> faces, tquads;
  vector<int>   valence, offsets, items;
};
void catmullclark_topology(const vector<vec4i>& quads, int num_vertices, bool lock_boundary, subdiv_level& level);
void subdivide_vertices(const subdiv_level& level, int dim, const vector<float>& verts, vector<float>& out);
// Extension: the same with every float stage on GPU `device` - the per-level vertex arithmetic (edge and face points, the
// averaging pass in face order, the correction pass: vpt_subdivide_vertices), the smooth normals before and after the
// displacement (vpt_vertex_normals) and the displacement itself (vpt_displace_vertices); the integer work - topology,
// split_facevarying, quads_to_triangles - stays on the host.  Same bits.
void             tesselate_surfaces_device(scene_data& scene, int device = 0);
// Single stages of tesselate_surface (device < 0: the host loops; else GPU `device`: vpt_vertex_normals / vpt_displace_vertices):
// quads_normals / triangles_normals (yocto_shape.cpp:1478-1512; corners = 4 / 3) and the displacement step (cpp:1259-1265).
vector<vec3f>    vertex_normals(const vector<vec3f>& positions, const int* faces, int num_faces, int corners, int device = -1);
vector<vec3f>    displace_vertices(const texture_data& texture, float displacement, const vector<vec3f>& positions, const vector<vec3f>& normals,
       const vector<vec2f>& texcoords, int device = -1);
// Progressively computes an image: ONE sample per pixel per call, on the GPU (vpt_render).
// Throws std::runtime_error("sampler unknown") for a bad shader (reference cpp:947-950) and
// std::runtime_error with vpt_last_error() if the HIP path is unavailable — there is no CPU
// fallback.
void pathtrace_samples(pathtrace_state& state, const scene_data& scene, const bvh_scene& bvh,
    const pathtrace_lights& lights, const pathtrace_params& params);
// Extension: `count` consecutive calls in one launch batch (same result).
void pathtrace_samples(pathtrace_state& state, const scene_data& scene, const bvh_scene& bvh,
    const pathtrace_lights& lights, const pathtrace_params& params, int count);
// Drop the cached device copy of `scene`.  pathtrace_samples notices by itself a scene rebuilt at the same address and every
// in-place edit of the small tables (cameras, instances, materials, environments, volume instances, SDFs, lights: hashed in full
// on each call); in-place edits of BULK data - vertex arrays, texels, voxels, BVH nodes, light CDFs - are only sampled at head
// and tail and have to be announced with this call.
void pathtrace_release(const scene_data& scene);
// Extension: the GPUs pathtrace_samples renders on (default {0}).  With more than one, the frame's 8x8 tiles are dealt
// round-robin to them (vpt_multi of include/vpt.h); the result does not depend on the list.  Drops the cached copies.
void pathtrace_set_devices(const vector<int>& devices);
color_image get_render(const pathtrace_state& state);
void        get_render(color_image& render, const pathtrace_state& state);

// ---- flattening to the C-ABI ----------------------------------------------------------------
struct flat_scene {
  vpt_scene_desc desc = {};
  // backing storage for desc
  vector<vpt_camera> cameras; vector<vpt_instance> instances; vector<vpt_shape> shapes;
  vector<vpt_material> materials; vector<vpt_texture> textures;
  vector<vpt_environment> environments; vector<vpt_volume> volumes;
  vector<vpt_volume_instance> vol_instances; vector<vpt_sdf> sdfs; vector<vpt_light> lights;
  vector<vec3f> positions, normals; vector<vec2f> texcoords; vector<vec4f> colors;
  vector<vec3i> triangles; vector<vec4i> quads;
  vector<vec4f> texels_f; vector<vec4b> texels_b; vector<float> voxels, light_cdf;
  vector<bvh_node> scene_nodes, shape_nodes; vector<int> scene_prims, shape_prims;
  flat_scene() = default;
  flat_scene(const flat_scene&) = delete;
  flat_scene& operator=(const flat_scene&) = delete;
};
// Throws std::invalid_argument for features outside the hot-path scope (points, lines).
void flatten_scene(flat_scene& flat, const scene_data& scene, const bvh_scene& bvh,
    const pathtrace_lights& lights);
vpt_params to_abi(const pathtrace_params& params);

// ---- scene / image IO (yocto_sceneio.h:89-211 subset: JSON 4.2, binary+ascii PLY, OBJ geometry (v / vn / vt / f / l / p),
//      PNG, HDR, .sdf text/binary) ----------------------------------------------------------------------
bool load_scene(const string& filename, scene_data& scene, string& error);
bool load_shape(const string& filename, shape_data& shape, string& error, bool flip_texcoord);
bool load_subdiv(const string& filename, subdiv_data& subdiv, string& error);
bool load_texture(const string& filename, texture_data& texture, string& error);
bool load_volume(const string& filename, volume_data& vol, bool binary, string& error);
bool save_image(const string& filename, const color_image& image, string& error);
// the members of a --config file (yocto_cli.cpp:912-945) as (option name, value text) pairs
bool load_cli_config(const string& filename, vector<std::pair<string, string>>& options, string& error);
// output quantisation (yocto_color.h:207-231, yocto_image.cpp:870-874)
vector<vec4b> linear_to_srgb8(const color_image& image);
// bit-faithful restatement of the reference's baseline JPEG writer at quality 75
// (libs/yocto/ext/stb_image_write.h:1250-1611); needed by the parity metric
vector<uint8_t> encode_jpeg_q75(int width, int height, const vector<vec4b>& rgba);
vector<uint8_t> encode_png(int width, int height, const vector<vec4b>& rgba);

}  // namespace vpt
