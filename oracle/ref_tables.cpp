// oracle/ref_tables.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Known-answer tables from the REFERENCE'S OWN functions.  The dispatch
// functions of the reference's integrator (eval_bsdfcos, sample_lights, sample_lights_pdf, ...) are
// file-local (`static`, yocto_pathtrace.cpp:86-421), so this translation unit includes that source file
// from where it lies under /root/reference (oracle/Makefile passes the include path; nothing is copied)
// and calls them directly; everything else is the reference's public API (yocto_scene.h, yocto_bvh.h,
// yocto_sdfs.h, yocto_shading.h).  Links against the same reference objects as ref_driver, minus
// yocto_pathtrace.o (compiled here).
//
//   ref_tables <scene.json | -> <op> <iparam> <in.bin> <out.bin>
//
// in.bin: n records of float32, out.bin: n records of float32; record layouts and op numbers are those of
// include/vpt_kat.h (the table in its header comment).  Scene-free ops take "-" as the scene.

#include <yocto_pathtrace/yocto_pathtrace.cpp>   // the reference source itself: gives access to its static functions

#include <yocto/yocto_sceneio.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "vpt_kat.h"

using namespace yocto;

static void die(const std::string& msg) {
  fprintf(stderr, "ref_tables: %s\n", msg.c_str());
  exit(1);
}

static const int k_strides[VPT_KAT_OP_COUNT][2] = {{19, 22}, {15, 10}, {4, 4}, {5, 6}, {7, 5}, {7, 24}, {3, 3}, {7, 3},
    {6, 1}, {6, 1}, {4, 3}, {6, 3}, {7, 4}, {4, 1}, {4, 1}};

static vec3f v3(const float* p) { return {p[0], p[1], p[2]}; }
static void  put3(float* p, const vec3f& v) { p[0] = v.x, p[1] = v.y, p[2] = v.z; }

int main(int argc, const char** argv) {
  if (argc != 6) die("usage: ref_tables <scene.json|-> <op> <iparam> <in.bin> <out.bin>");
  auto scene_name = std::string{argv[1]};
  auto op         = atoi(argv[2]);
  auto iparam     = atoi(argv[3]);
  if (op < 0 || op >= VPT_KAT_OP_COUNT) die("unknown op");
  auto si = k_strides[op][0], so = k_strides[op][1];

  auto f = fopen(argv[4], "rb");
  if (!f) die("cannot read input");
  fseek(f, 0, SEEK_END);
  auto bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (bytes % (4 * si) != 0) die("input size is not a multiple of the record size");
  auto n  = (int)(bytes / (4 * si));
  auto in = std::vector<float>((size_t)n * si);
  if (fread(in.data(), 4, in.size(), f) != in.size()) die("short read");
  fclose(f);
  auto out = std::vector<float>((size_t)n * so, 0.0f);

  auto scene  = scene_data{};
  auto bvh    = bvh_scene{};
  auto lights = pathtrace_lights{};
  auto params = pathtrace_params{};
  if (scene_name != "-") {
    auto error = std::string{};
    if (!load_scene(scene_name, scene, error)) die(error);
    tesselate_surfaces(scene);
    bvh    = make_bvh(scene, params);
    lights = make_lights(scene, params);
  }

  for (auto r = 0; r < n; r++) {
    auto a = in.data() + (size_t)r * si;
    auto o = out.data() + (size_t)r * so;
    switch (op) {
      case VPT_KAT_LOBES: {
        auto m      = material_point{};
        m.type      = (material_type)(int)a[0];
        m.color     = v3(a + 1);
        m.roughness = a[4], m.metallic = a[5], m.ior = a[6];
        auto normal = v3(a + 7), outgoing = v3(a + 10);
        auto rnl = a[13];
        auto rn  = vec2f{a[14], a[15]};
        auto alt = v3(a + 16);
        auto s_in = sample_bsdfcos(m, normal, outgoing, rnl, rn);
        put3(o, s_in);
        if (s_in != vec3f{0, 0, 0}) {
          put3(o + 3, eval_bsdfcos(m, normal, outgoing, s_in));
          o[6] = sample_bsdfcos_pdf(m, normal, outgoing, s_in);
        }
        put3(o + 7, eval_bsdfcos(m, normal, outgoing, alt));
        o[10]     = sample_bsdfcos_pdf(m, normal, outgoing, alt);
        auto d_in = sample_delta(m, normal, outgoing, rnl);
        put3(o + 11, d_in);
        if (d_in != vec3f{0, 0, 0}) {
          put3(o + 14, eval_delta(m, normal, outgoing, d_in));
          o[17] = sample_delta_pdf(m, normal, outgoing, d_in);
        }
        put3(o + 18, eval_delta(m, normal, outgoing, alt));
        o[21] = sample_delta_pdf(m, normal, outgoing, alt);
      } break;
      case VPT_KAT_MEDIA: {
        auto density = v3(a);
        auto maxd = a[3], rl = a[4], rd = a[5], g = a[6];
        auto outgoing = v3(a + 7);
        auto rn       = vec2f{a[10], a[11]};
        auto incoming = v3(a + 12);
        auto distance = sample_transmittance(density, maxd, rl, rd);
        o[0]          = distance;
        o[1]          = sample_transmittance_pdf(density, distance, maxd);
        put3(o + 2, eval_transmittance(density, distance));
        o[5]       = eval_phasefunction(g, outgoing, incoming);
        auto s_dir = sample_phasefunction(g, outgoing, rn);
        put3(o + 6, s_dir);
        o[9] = eval_phasefunction(g, outgoing, s_dir);
      } break;
      case VPT_KAT_TEXTURE: {
        auto t = (int)a[0];
        if (t < 0 || t >= (int)scene.textures.size()) die("texture out of range");
        auto c = eval_texture(scene.textures[t], vec2f{a[1], a[2]}, a[3] != 0);
        o[0] = c.x, o[1] = c.y, o[2] = c.z, o[3] = c.w;
      } break;
      case VPT_KAT_CAMERA: {
        auto c = (int)a[0];
        if (c < 0 || c >= (int)scene.cameras.size()) die("camera out of range");
        auto ray = eval_camera(scene.cameras[c], vec2f{a[1], a[2]}, vec2f{a[3], a[4]});
        put3(o, ray.o), put3(o + 3, ray.d);
      } break;
      case VPT_KAT_INTERSECT: {
        auto ray  = ray3f{v3(a), v3(a + 3)};
        auto inst = (int)a[6];
        if (inst >= (int)scene.instances.size()) die("instance out of range");
        auto isec = inst < 0 ? intersect_bvh(bvh, scene, ray) : intersect_bvh(bvh, scene, inst, ray);
        o[0] = isec.hit ? (float)isec.instance : -1.0f, o[1] = isec.hit ? (float)isec.element : -1.0f;
        o[2] = isec.hit ? isec.uv.x : 0, o[3] = isec.hit ? isec.uv.y : 0, o[4] = isec.hit ? isec.distance : 0;
      } break;
      case VPT_KAT_SURFACE: {
        auto inst = (int)a[0], element = (int)a[1];
        if (inst < 0 || inst >= (int)scene.instances.size()) die("instance out of range");
        auto& instance = scene.instances[inst];
        auto  uv       = vec2f{a[2], a[3]};
        auto  outgoing = v3(a + 4);
        put3(o, eval_shading_position(scene, instance, element, uv, outgoing));
        put3(o + 3, eval_shading_normal(scene, instance, element, uv, outgoing));
        auto m = eval_material(scene, instance, element, uv);
        o[6]   = (float)(int)m.type;
        put3(o + 7, m.emission), put3(o + 10, m.color);
        o[13] = m.opacity, o[14] = m.roughness, o[15] = m.metallic, o[16] = m.ior;
        put3(o + 17, m.density), put3(o + 20, m.scattering);
        o[23] = m.scanisotropy;
      } break;
      case VPT_KAT_ENVIRONMENT: put3(o, eval_environment(scene, v3(a))); break;
      case VPT_KAT_SAMPLE_LIGHTS: put3(o, sample_lights(scene, lights, v3(a), a[3], a[4], vec2f{a[5], a[6]})); break;
      case VPT_KAT_LIGHTS_PDF:
      case VPT_KAT_LIGHTS_PDF_K2: o[0] = sample_lights_pdf(scene, bvh, lights, v3(a), v3(a + 3), iparam); break;
      case VPT_KAT_SDF_SCENE: {
        auto res = eval_sdf_scene(scene, v3(a), a[3]);
        o[0] = res.result, o[1] = (float)res.instance, o[2] = (float)res.sdf;
      } break;
      case VPT_KAT_SDF_NORMAL: {
        auto kind = (int)a[0], idx = (int)a[1];
        if (kind == 0) {
          if (idx < 0 || idx >= (int)scene.vol_instances.size()) die("vol_instance out of range");
          auto& inst = scene.vol_instances[idx];
          put3(o, eval_sdf_normal(scene.volumes[inst.volume], inst, v3(a + 2), a[5]));
        } else {
          if (idx < 0 || idx >= (int)scene.sdfs.size()) die("sdf out of range");
          put3(o, eval_sdf_normal(scene.sdfs[idx], v3(a + 2), a[5]));
        }
      } break;
      case VPT_KAT_SPHERETRACE: {
        auto ray = ray3f{v3(a), v3(a + 3)};
        auto sdf = (int)a[6];
        if (sdf >= (int)scene.sdfs.size()) die("sdf out of range");
        auto res = sdf < 0 ? spheretrace(scene, ray, iparam) : spheretrace(scene, ray, sdf, iparam);
        o[0] = res.hit ? 1.0f : 0.0f, o[1] = res.dist, o[2] = (float)res.instance, o[3] = (float)res.sdf;
      } break;
      case VPT_KAT_VOLUME: {
        auto v = (int)a[0];
        if (v < 0 || v >= (int)scene.volumes.size()) die("volume out of range");
        o[0] = eval_volume(scene.volumes[v], v3(a + 1));
      } break;
      case VPT_KAT_SDF_FUNCTION: {
        auto s = (int)a[0];
        if (s < 0 || s >= (int)scene.sdfs.size()) die("sdf out of range");
        o[0] = scene.sdfs[s].f(v3(a + 1));
      } break;
    }
  }

  f = fopen(argv[5], "wb");
  if (!f) die("cannot write output");
  fwrite(out.data(), 4, out.size(), f);
  fclose(f);
  return 0;
}
