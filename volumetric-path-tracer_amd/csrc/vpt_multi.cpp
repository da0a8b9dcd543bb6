// vpt_multi.cpp — the multi-GPU fan-out behind the boundary (SURVEY.md §8(b) "multi-GPU fan-out is internal", §8(e)):
// one process, ndev GPUs, one host thread per GPU.  The frame is cut into 8x8-pixel tiles, tile t belongs to
// devices[t % ndev] (the vpt_layout partition of include/vpt.h); every device holds the whole scene (read-only, tens of
// MB) and the state of its own tiles only, so the render itself needs no communication at all.
//
// Residency ("each GPU keeps its tiles' image/hits/rng resident across sample batches").  The tile state
// lives on the devices between calls.  vpt_multi_set_state uploads a host pathtrace_state, vpt_multi_get_state downloads
// it, vpt_multi_render with null host pointers renders on what is resident and moves nothing.  vpt_multi_render with
// host pointers keeps the contract of vpt_render (the caller's arrays are valid after every call and are READ on every
// call, as the reference reads the live state, yocto_pathtrace.cpp:1081-1090).  A device's part of the upload is skipped
// only when it is certain that the device already holds it: same frame size and sample count, the SAME three base
// pointers as the call that last downloaded into them, and a 64-bit checksum over EVERY word of the part (radiance sum,
// hit count, RNG state of each of its pixels), taken by the device's thread when it stored the download and taken again
// from the caller's arrays before the skip - one edited pixel anywhere, or another state object at another address,
// is uploaded (round 3 compared 64 probe pixels only; tests/test_multi_gpu.py edits a pixel no probe looked at).
// VPT_MULTI_RESIDENT=0 switches the skip off altogether.  Host staging is pinned (hipHostMalloc), one tile-major buffer
// per device; the row-major <-> tile-major scatter runs in the per-device threads.
//
// The one exchange is the frame assembly of vpt_multi_get_render: the float4 tile buffers travel to devices[0] over
// xGMI with RCCL (grouped ncclSend / ncclRecv: ndev - 1 concurrent point-to-point transfers, no ring, no reduction) and
// are resolved there by the same kernel the single-GPU path uses.  RCCL is bound at run time, not linked: a copy that
// the process already carries (PyTorch ships its own) is reused (dlopen RTLD_NOLOAD), otherwise the library is opened
// RTLD_LOCAL; a process that renders on one GPU never loads it.  Where RCCL cannot be had (library missing,
// ncclCommInitAll fails, the same device listed twice as the one-GPU tests do) the parts travel by
// hipMemcpyPeerAsync instead and vpt_multi_transport() says so.  VPT_MULTI_FORCE_RCCL=1 builds a communicator even for
// one device and sends that device's buffer to itself through ncclSend / ncclRecv, so that every RCCL entry point used
// here runs on a one-GPU box (tests/test_multi_gpu.py).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "vpt.h"

int vpt_set_error(int code, const char* fmt, ...);   // vpt_capi.hip: records the message for vpt_last_error() on this thread

namespace {

// ---- the few RCCL entry points used, resolved at run time ------------------------------------------------------
typedef struct ncclComm* ncclComm_t;
enum { ncclSuccess = 0, ncclFloat = 7 };   // rccl.h: ncclResult_t / ncclDataType_t values (ncclFloat32 = 7)
struct rccl_api {
  void* lib = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*)                                 = nullptr;
  int (*CommDestroy)(ncclComm_t)                                                   = nullptr;
  int (*GroupStart)()                                                              = nullptr;
  int (*GroupEnd)()                                                                = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t)              = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t)                    = nullptr;
  const char* (*GetErrorString)(int)                                               = nullptr;
  const char* error_text(int r) const { return GetErrorString ? GetErrorString(r) : "failed"; }
};
bool load_rccl(rccl_api& api, std::string& why) {
  // VPT_MULTI_RCCL_LIB names the library (tests: a stub whose ncclCommInitAll fails); otherwise a copy the process
  // already holds is reused before a second one is opened, and nothing is added to the global symbol scope
  std::vector<std::string> names;
  if (const char* e = getenv("VPT_MULTI_RCCL_LIB")) names = {e};
  else names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (int pass = 0; pass < 2 && !api.lib; pass++)
    for (auto& name : names) {
      api.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (api.lib) break;
    }
  if (!api.lib) {
    const char* e = dlerror();
    why = std::string("cannot load RCCL: ") + (e ? e : "not found");
    return false;
  }
  auto sym = [&](const char* n) { return dlsym(api.lib, n); };
  api.CommInitAll    = (decltype(api.CommInitAll))sym("ncclCommInitAll");
  api.CommDestroy    = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.GroupStart     = (decltype(api.GroupStart))sym("ncclGroupStart");
  api.GroupEnd       = (decltype(api.GroupEnd))sym("ncclGroupEnd");
  api.Send           = (decltype(api.Send))sym("ncclSend");
  api.Recv           = (decltype(api.Recv))sym("ncclRecv");
  api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  if (!api.CommInitAll || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send || !api.Recv) {
    why = "the RCCL library lacks an expected entry point";
    return false;
  }
  return true;
}

// slot -> row-major pixel index (or -1) of `rank`'s part of the frame: host mirror of slot_to_pixel (vpt_kernels.hip.h)
void slot_map(int width, int height, int rank, int nranks, std::vector<int>& pixel_of_slot) {
  const int tile = 8, per_tile = 64;
  long long tiles_x = (width + tile - 1) / tile, tiles_y = (height + tile - 1) / tile, tiles = tiles_x * tiles_y;
  long long local = (tiles + nranks - 1) / nranks;
  pixel_of_slot.assign((size_t)(local * per_tile), -1);
  for (long long lt = 0; lt < local; lt++) {
    long long t = lt * nranks + rank;
    if (t >= tiles) continue;
    long long ty = t / tiles_x, tx = t - ty * tiles_x;
    for (int q = 0; q < per_tile; q++) {
      long long px = tx * tile + (q & 7), py = ty * tile + (q >> 3);
      if (px < width && py < height) pixel_of_slot[(size_t)(lt * per_tile + q)] = (int)(py * width + px);
    }
  }
}

struct device_part {
  int         device = 0;
  vpt_scene*  scene  = nullptr;
  hipStream_t stream = nullptr;
  void *d_image = nullptr, *d_hits = nullptr, *d_rng = nullptr;   // tile-major state of this device's tiles (resident)
  std::vector<int> pixel_of_slot;
  float*    h_image = nullptr;   // pinned staging, tile-major
  int32_t*  h_hits  = nullptr;
  uint64_t* h_rng   = nullptr;
  uint64_t  mirror_sum = 0;      // checksum of this part of the caller's arrays as the last download left it
  bool      mirrored   = false;  // mirror_sum describes what the device holds now
  bool      uploaded   = false;  // the last render call uploaded this part (vpt_multi_uploaded_parts, tests)
  int         rc = VPT_OK;
  std::string error;
};

}  // namespace

struct vpt_multi {
  std::vector<device_part> parts;
  int  width = 0, height = 0, samples = 0;   // the frame the device buffers are sized for / the samples they hold
  bool resident = false;                     // the device buffers hold a state (set_state or a render put it there)
  const void *mirror_image = nullptr, *mirror_hits = nullptr, *mirror_rng = nullptr;   // the caller's arrays the last download wrote (part of the residency key)
  bool distinct = true;                      // all devices different
  bool use_rccl = false;                     // frame assembly by ncclSend / ncclRecv (else hipMemcpyPeerAsync)
  rccl_api                rccl;
  std::vector<ncclComm_t> comms;
  void *d_gathered = nullptr, *d_frame = nullptr;   // on parts[0].device
};

namespace {
void forget_mirror(vpt_multi* m) {
  m->mirror_image = m->mirror_hits = m->mirror_rng = nullptr;
  for (auto& p : m->parts) p.mirrored = false;
}
void free_buffers(vpt_multi* m) {
  for (auto& p : m->parts) {
    if (!p.scene) continue;   // never came to life on its device (vpt_multi_create failed before it)
    (void)hipSetDevice(p.device);
    for (void** b : {&p.d_image, &p.d_hits, &p.d_rng})
      if (*b) (void)hipFree(*b), *b = nullptr;
    for (void** b : {(void**)&p.h_image, (void**)&p.h_hits, (void**)&p.h_rng})
      if (*b) (void)hipHostFree(*b), *b = nullptr;
  }
  if (!m->parts.empty() && m->parts[0].scene) (void)hipSetDevice(m->parts[0].device);
  for (void** b : {&m->d_gathered, &m->d_frame})
    if (*b) (void)hipFree(*b), *b = nullptr;
  m->width = m->height = m->samples = 0;
  m->resident = false;
  forget_mirror(m);
}
#define HIP_OK(expr, p)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (p).rc = VPT_ERR_HIP, (p).error = std::string(#expr ": ") + hipGetErrorString(e_);      \
      return;                                                                                 \
    }                                                                                         \
  } while (0)

// device and pinned buffers for a width x height frame (no-op when they already fit it)
int size_for(vpt_multi* m, int width, int height) {
  if (m->width == width && m->height == height) return VPT_OK;
  free_buffers(m);
  const int ndev = (int)m->parts.size();
  for (int i = 0; i < ndev; i++) {
    auto& p = m->parts[(size_t)i];
    slot_map(width, height, i, ndev, p.pixel_of_slot);
    size_t n = p.pixel_of_slot.size();
    if (hipSetDevice(p.device) != hipSuccess || hipMalloc(&p.d_image, n * 16) != hipSuccess || hipMalloc(&p.d_hits, n * 4) != hipSuccess ||
        hipMalloc(&p.d_rng, n * 16) != hipSuccess || hipHostMalloc((void**)&p.h_image, n * 16, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&p.h_hits, n * 4, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&p.h_rng, n * 16, hipHostMallocDefault) != hipSuccess) {
      free_buffers(m);
      return vpt_set_error(VPT_ERR_HIP, "cannot allocate the tile state for device %d", p.device);
    }
  }
  m->width = width, m->height = height;
  return VPT_OK;
}

// run work(i) for every part, one host thread per GPU; the first failure is the call's error
template <typename F>
int fan_out(vpt_multi* m, F work) {
  const int ndev = (int)m->parts.size();
  for (auto& p : m->parts) p.rc = VPT_OK, p.error.clear();
  std::vector<std::thread> threads;
  for (int i = 1; i < ndev; i++) threads.emplace_back(work, i);
  work(0);
  for (auto& t : threads) t.join();
  for (auto& p : m->parts)
    if (p.rc != VPT_OK) return vpt_set_error(p.rc, "device %d: %s", p.device, p.error.c_str());
  return VPT_OK;
}

// 64-bit checksum over every word of one device's part of the caller's arrays.  Per pixel the five words (two of the
// radiance sum, the hit count, two of the RNG state) enter a multilinear form with odd multipliers, offset by the pixel's
// index: a change of any single word changes the pixel's term for certain (multiplication by an odd number is a bijection
// mod 2^64, and so is x ^ x >> 29), several changes at once cancel with probability 2^-64.
uint64_t part_checksum(const device_part& p, const float* image_rgba, const int32_t* hits, const uint64_t* rng) {
  uint64_t sum = 0;
  const size_t n = p.pixel_of_slot.size();
  for (size_t s = 0; s < n; s++) {
    int px = p.pixel_of_slot[s];
    if (px < 0) continue;
    uint64_t w[2];
    memcpy(w, image_rgba + 4 * (size_t)px, 16);
    uint64_t h = ((uint64_t)px + 1) * 0x9e3779b97f4a7c15ull;
    h += w[0] * 0xff51afd7ed558ccdull + w[1] * 0xc4ceb9fe1a85ec53ull + (uint64_t)(uint32_t)hits[px] * 0xd6e8feb86659fd93ull;
    h += rng[2 * (size_t)px] * 0x2545f4914f6cdd1dull + rng[2 * (size_t)px + 1] * 0x94d049bb133111ebull;
    sum += h ^ (h >> 29);
  }
  return sum;
}

// part i: the caller's row-major arrays -> pinned tile-major staging -> device
void upload_part(vpt_multi* m, int i, const float* image_rgba, const int32_t* hits, const uint64_t* rng) {
  auto&  p = m->parts[(size_t)i];
  size_t n = p.pixel_of_slot.size();
  for (size_t s = 0; s < n; s++) {
    int px = p.pixel_of_slot[s];
    if (px < 0) {   // padding slots are never touched by the kernels; keep the staging defined
      memset(&p.h_image[4 * s], 0, 16), p.h_hits[s] = 0, p.h_rng[2 * s] = p.h_rng[2 * s + 1] = 0;
      continue;
    }
    memcpy(&p.h_image[4 * s], image_rgba + 4 * (size_t)px, 16);
    p.h_hits[s] = hits[px];
    p.h_rng[2 * s] = rng[2 * (size_t)px], p.h_rng[2 * s + 1] = rng[2 * (size_t)px + 1];
  }
  HIP_OK(hipSetDevice(p.device), p);
  HIP_OK(hipMemcpyAsync(p.d_image, p.h_image, n * 16, hipMemcpyHostToDevice, p.stream), p);
  HIP_OK(hipMemcpyAsync(p.d_hits, p.h_hits, n * 4, hipMemcpyHostToDevice, p.stream), p);
  HIP_OK(hipMemcpyAsync(p.d_rng, p.h_rng, n * 16, hipMemcpyHostToDevice, p.stream), p);
}
// part i: device -> pinned staging -> the caller's arrays (tiles are disjoint between devices: the threads write different pixels)
void download_part(vpt_multi* m, int i, float* image_rgba, int32_t* hits, uint64_t* rng) {
  auto&  p = m->parts[(size_t)i];
  size_t n = p.pixel_of_slot.size();
  HIP_OK(hipSetDevice(p.device), p);
  HIP_OK(hipMemcpyAsync(p.h_image, p.d_image, n * 16, hipMemcpyDeviceToHost, p.stream), p);
  HIP_OK(hipMemcpyAsync(p.h_hits, p.d_hits, n * 4, hipMemcpyDeviceToHost, p.stream), p);
  HIP_OK(hipMemcpyAsync(p.h_rng, p.d_rng, n * 16, hipMemcpyDeviceToHost, p.stream), p);
  HIP_OK(hipStreamSynchronize(p.stream), p);
  for (size_t s = 0; s < n; s++) {
    int px = p.pixel_of_slot[s];
    if (px < 0) continue;
    memcpy(image_rgba + 4 * (size_t)px, &p.h_image[4 * s], 16);
    hits[px] = p.h_hits[s];
    rng[2 * (size_t)px] = p.h_rng[2 * s], rng[2 * (size_t)px + 1] = p.h_rng[2 * s + 1];
  }
  p.mirror_sum = part_checksum(p, image_rgba, hits, rng), p.mirrored = true;
}

bool always_upload() {   // VPT_MULTI_RESIDENT=0: every host-pointer call uploads the caller's arrays (no checksum pass)
  const char* e = getenv("VPT_MULTI_RESIDENT");
  return e && !strcmp(e, "0");
}
}  // namespace

extern "C" {

int vpt_multi_create(const vpt_scene_desc* desc, const int* devices, int ndev, vpt_multi** out) {
  if (!desc || !devices || !out || ndev < 1 || ndev > 64) return vpt_set_error(VPT_ERR_INVALID_ARG, "bad argument");
  *out = nullptr;
  auto m = new vpt_multi{};
  m->parts.resize((size_t)ndev);
  std::set<int> seen;
  for (int i = 0; i < ndev; i++) {
    auto& p  = m->parts[(size_t)i];
    p.device = devices[i];
    if (!seen.insert(devices[i]).second) m->distinct = false;
    if (int rc = vpt_scene_create(desc, devices[i], &p.scene)) {   // validates, uploads; sets the error text
      vpt_multi_destroy(m);
      return rc;
    }
    if (hipSetDevice(p.device) != hipSuccess || hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking) != hipSuccess) {
      vpt_multi_destroy(m);
      return vpt_set_error(VPT_ERR_HIP, "cannot create a stream on device %d", devices[i]);
    }
  }
  const char* force = getenv("VPT_MULTI_FORCE_RCCL");
  const bool  want_rccl = m->distinct && (ndev > 1 || (force && !strcmp(force, "1")));
  if (want_rccl) {
    std::string why;
    if (load_rccl(m->rccl, why)) {
      m->comms.assign((size_t)ndev, nullptr);
      if (int r = m->rccl.CommInitAll(m->comms.data(), ndev, devices)) {
        why = std::string("ncclCommInitAll: ") + m->rccl.error_text(r);
        for (auto c : m->comms)   // whatever it had created before it failed
          if (c) (void)m->rccl.CommDestroy(c);
        m->comms.clear();
      } else m->use_rccl = true;
    }
    if (!m->use_rccl) fprintf(stderr, "[vpt] multi-GPU frame assembly falls back to hipMemcpyPeerAsync (%s)\n", why.c_str());
  }
  *out = m;
  return VPT_OK;
}

void vpt_multi_destroy(vpt_multi* m) {
  if (!m) return;
  for (auto c : m->comms)
    if (c) (void)m->rccl.CommDestroy(c);
  free_buffers(m);
  for (auto& p : m->parts) {
    if (!p.scene) continue;
    (void)hipSetDevice(p.device);
    if (p.stream) (void)hipStreamDestroy(p.stream);
    vpt_scene_destroy(p.scene);
  }
  (void)hipGetLastError();   // a failed call above must not surface in the next entry point's error check
  delete m;
}

int vpt_multi_device_count(const vpt_multi* m) { return m ? (int)m->parts.size() : 0; }

const char* vpt_multi_transport(const vpt_multi* m) { return !m ? "" : m->use_rccl ? "rccl" : m->parts.size() > 1 ? "peer-copy" : "local"; }

int vpt_multi_set_state(vpt_multi* m, int width, int height, const float* image_rgba, const int32_t* hits, const uint64_t* rng, int samples) {
  if (!m || !image_rgba || !hits || !rng) return vpt_set_error(VPT_ERR_INVALID_ARG, "null argument");
  if (width <= 0 || height <= 0 || samples < 0) return vpt_set_error(VPT_ERR_INVALID_ARG, "bad image size or sample count");
  if (int rc = size_for(m, width, height)) return rc;
  m->resident = false, forget_mirror(m);
  int rc = fan_out(m, [&](int i) {
    upload_part(m, i, image_rgba, hits, rng);
    auto& p = m->parts[(size_t)i];
    if (p.rc == VPT_OK) HIP_OK(hipStreamSynchronize(p.stream), p);   // the pinned staging is reused by the next call
  });
  if (rc != VPT_OK) return rc;
  m->resident = true, m->samples = samples;
  return VPT_OK;
}

int vpt_multi_get_state(vpt_multi* m, float* image_rgba, int32_t* hits, uint64_t* rng, int* samples) {
  if (!m || !image_rgba || !hits || !rng) return vpt_set_error(VPT_ERR_INVALID_ARG, "null argument");
  if (!m->resident) return vpt_set_error(VPT_ERR_INVALID_ARG, "no state on the devices");
  forget_mirror(m);
  if (int rc = fan_out(m, [&](int i) { download_part(m, i, image_rgba, hits, rng); })) return rc;
  m->mirror_image = image_rgba, m->mirror_hits = hits, m->mirror_rng = rng;
  if (samples) *samples = m->samples;
  return VPT_OK;
}

int vpt_multi_render(vpt_multi* m, const vpt_params* params, int nsamples, int width, int height, float* image_rgba,
    int32_t* hits, uint64_t* rng, int* samples_io) {
  if (!m || !params || !samples_io) return vpt_set_error(VPT_ERR_INVALID_ARG, "null argument");
  const bool on_device = !image_rgba && !hits && !rng;   // render on the resident state, move nothing
  if (!on_device && (!image_rgba || !hits || !rng)) return vpt_set_error(VPT_ERR_INVALID_ARG, "null argument");
  if (width <= 0 || height <= 0) return vpt_set_error(VPT_ERR_INVALID_ARG, "bad image size");
  if (params->shader < 0 || params->shader > VPT_SHADER_IMPLICIT_NORMAL) return vpt_set_error(VPT_ERR_UNKNOWN_SHADER, "sampler unknown");
  if (on_device && (!m->resident || m->width != width || m->height != height || m->samples != *samples_io))
    return vpt_set_error(VPT_ERR_INVALID_ARG, "no %dx%d state with %d samples on the devices (vpt_multi_set_state first)", width, height, *samples_io);
  int todo = params->samples - *samples_io;   // no-op once reached, yocto_pathtrace.cpp:1055
  if (nsamples < todo) todo = nsamples;
  if (todo <= 0) return VPT_OK;
  const int ndev = (int)m->parts.size();
  // The caller's arrays are the state, as in the reference: every part of them is uploaded unless the device provably holds it
  // already - the arrays are the very ones (same addresses, same size, same sample count) the last call downloaded into, and
  // the part's checksum over all of its words, re-taken now by the device's thread, is what that download stored.
  const bool may_skip = !on_device && m->resident && !always_upload() && m->width == width && m->height == height && m->samples == *samples_io &&
                        m->samples > 0 && m->mirror_image == image_rgba && m->mirror_hits == hits && m->mirror_rng == rng;
  if (!on_device && !may_skip) {
    if (int rc = size_for(m, width, height)) return rc;
    m->resident = false, forget_mirror(m);
  }
  int rc = fan_out(m, [&](int i) {
    auto& p = m->parts[(size_t)i];
    p.uploaded = false;
    if (!on_device && !(may_skip && p.mirrored && part_checksum(p, image_rgba, hits, rng) == p.mirror_sum)) {
      p.mirrored = false;
      upload_part(m, i, image_rgba, hits, rng);
      if (p.rc != VPT_OK) return;
      p.uploaded = true;
    }
    HIP_OK(hipSetDevice(p.device), p);
    vpt_layout lay = {width, height, 8, 8, i, ndev};
    if (int r = vpt_render_device(p.scene, params, &lay, todo, p.d_image, p.d_hits, p.d_rng, p.stream)) {
      p.rc = r, p.error = vpt_last_error();
      return;
    }
    if (!on_device) download_part(m, i, image_rgba, hits, rng);   // ends with a stream synchronisation
    else HIP_OK(hipStreamSynchronize(p.stream), p);
    if (p.rc != VPT_OK) return;
    if (int r = vpt_check_watchdog(p.scene)) p.rc = r, p.error = vpt_last_error();   // a wave of the implicit kernel gave up: incomplete image
  });
  if (rc != VPT_OK) {
    m->resident = false, forget_mirror(m);   // some devices may have advanced, others not
    return rc;
  }
  *samples_io += todo;
  m->samples = *samples_io, m->resident = true;
  if (!on_device) m->mirror_image = image_rgba, m->mirror_hits = hits, m->mirror_rng = rng;   // download_part took the checksums
  else forget_mirror(m);   // the devices moved on; no host array mirrors them
  return VPT_OK;
}

// parts of the last vpt_multi_render call with host pointers that were uploaded (0 ... device count): lets a caller - and
// the tests - see what the residency check decided
int vpt_multi_uploaded_parts(const vpt_multi* m) {
  int n = 0;
  if (m)
    for (auto& p : m->parts) n += p.uploaded;
  return n;
}

int vpt_multi_get_render(vpt_multi* m, float* image_rgba) {
  if (!m || !image_rgba) return vpt_set_error(VPT_ERR_INVALID_ARG, "null argument");
  if (!m->resident || m->width <= 0 || m->samples <= 0) return vpt_set_error(VPT_ERR_INVALID_ARG, "nothing rendered yet");
  const int ndev = (int)m->parts.size();
  auto&     p0   = m->parts[0];
  size_t    n    = p0.pixel_of_slot.size();   // slots per device (the same on every device)
  if (hipSetDevice(p0.device) != hipSuccess) return vpt_set_error(VPT_ERR_HIP, "hipSetDevice failed");
  if (!m->d_gathered) {
    if (hipMalloc(&m->d_gathered, (size_t)ndev * n * 16) != hipSuccess || hipMalloc(&m->d_frame, (size_t)m->width * m->height * 16) != hipSuccess)
      return vpt_set_error(VPT_ERR_HIP, "cannot allocate the gather buffers on device %d", p0.device);
  }
  // the tile buffers of all devices -> devices[0]
  if (m->use_rccl) {
    // grouped point-to-point: every other device sends its part to rank 0, which receives ndev - 1 parts; a one-device
    // communicator (VPT_MULTI_FORCE_RCCL=1) sends its part to itself, which is legal inside a group
    const int first = ndev > 1 ? 1 : 0;   // several devices: devices[0]'s own part is a local copy
    if (first == 1 && hipMemcpyAsync(m->d_gathered, p0.d_image, n * 16, hipMemcpyDeviceToDevice, p0.stream) != hipSuccess)
      return vpt_set_error(VPT_ERR_HIP, "local tile copy failed");
    int r = m->rccl.GroupStart();
    for (int i = first; i < ndev && r == ncclSuccess; i++) {
      r = m->rccl.Send(m->parts[(size_t)i].d_image, n * 4, ncclFloat, 0, m->comms[(size_t)i], m->parts[(size_t)i].stream);
      if (r == ncclSuccess) r = m->rccl.Recv((char*)m->d_gathered + (size_t)i * n * 16, n * 4, ncclFloat, i, m->comms[0], p0.stream);
    }
    int e = m->rccl.GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) return vpt_set_error(VPT_ERR_HIP, "RCCL tile gather: %s", m->rccl.error_text(r));
    for (int i = 1; i < ndev; i++) {   // the sends ran on the senders' streams
      (void)hipSetDevice(m->parts[(size_t)i].device);
      if (hipStreamSynchronize(m->parts[(size_t)i].stream) != hipSuccess) return vpt_set_error(VPT_ERR_HIP, "send stream failed");
    }
    (void)hipSetDevice(p0.device);
  } else {
    for (int i = 0; i < ndev; i++) {   // no RCCL (or the same physical device listed more than once): peer copies
      auto&      p = m->parts[(size_t)i];
      hipError_t e = p.device == p0.device ? hipMemcpyAsync((char*)m->d_gathered + (size_t)i * n * 16, p.d_image, n * 16, hipMemcpyDeviceToDevice, p0.stream)
                                           : hipMemcpyPeerAsync((char*)m->d_gathered + (size_t)i * n * 16, p0.device, p.d_image, p.device, n * 16, p0.stream);
      if (e != hipSuccess) return vpt_set_error(VPT_ERR_HIP, "tile copy from device %d: %s", p.device, hipGetErrorString(e));
    }
  }
  vpt_layout lay = {m->width, m->height, 8, 8, 0, ndev};
  if (int rc = vpt_resolve_device(&lay, m->d_gathered, m->samples, m->d_frame, p0.stream)) return rc;
  if (hipMemcpyAsync(image_rgba, m->d_frame, (size_t)m->width * m->height * 16, hipMemcpyDeviceToHost, p0.stream) != hipSuccess ||
      hipStreamSynchronize(p0.stream) != hipSuccess)
    return vpt_set_error(VPT_ERR_HIP, "frame download failed");
  return VPT_OK;
}

}  // extern "C"
