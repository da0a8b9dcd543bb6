// vpt_multi.cpp — the multi-GPU fan-out behind the boundary (SURVEY.md §8(b) "multi-GPU fan-out is internal", §8(e)):
// one process, ndev GPUs, one host thread per GPU.  The frame is cut into 8x8-pixel tiles, tile t belongs to
// devices[t % ndev] (the vpt_layout partition of include/vpt.h); every device holds the whole scene (read-only, tens of
// MB) and the state of its own tiles only, so the render itself needs no communication at all.  The one exchange is
// the frame assembly of vpt_multi_get_render: the float4 tile buffers travel to devices[0] over xGMI with RCCL
// (grouped ncclSend / ncclRecv: ndev - 1 concurrent point-to-point transfers, no ring, no reduction) and are resolved
// there by the same kernel the single-GPU path uses.
//
// RCCL is loaded with dlopen when a communicator is first needed (ndev > 1 distinct devices), not linked: a process
// that renders on one GPU never loads it, and a process that already carries an RCCL (PyTorch ships its own) is not
// handed a second copy at link time.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "vpt.h"

// from vpt_capi.hip: the calling thread's error text
extern "C" const char* vpt_last_error(void);
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
};
bool load_rccl(rccl_api& api, std::string& why) {
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (api.lib) break;
  }
  if (!api.lib) {
    why = std::string("cannot load RCCL: ") + dlerror();
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
  void *d_image = nullptr, *d_hits = nullptr, *d_rng = nullptr;   // tile-major state of this device's tiles
  std::vector<int>      pixel_of_slot;
  std::vector<float>    h_image;   // staging, tile-major
  std::vector<int32_t>  h_hits;
  std::vector<uint64_t> h_rng;
  int         rc = VPT_OK;
  std::string error;
};

}  // namespace

struct vpt_multi {
  std::vector<device_part> parts;
  int  width = 0, height = 0, samples = 0;   // the frame the device buffers currently hold
  bool distinct = true;                      // all devices different (else the gather falls back to copies: RCCL refuses duplicates)
  rccl_api                rccl;
  std::vector<ncclComm_t> comms;
  void *d_gathered = nullptr, *d_frame = nullptr;   // on parts[0].device
};

namespace {
void free_buffers(vpt_multi* m) {
  for (auto& p : m->parts) {
    if (!p.scene) continue;   // never came to life on its device (vpt_multi_create failed before it)
    (void)hipSetDevice(p.device);
    for (void** b : {&p.d_image, &p.d_hits, &p.d_rng})
      if (*b) (void)hipFree(*b), *b = nullptr;
  }
  if (!m->parts.empty() && m->parts[0].scene) (void)hipSetDevice(m->parts[0].device);
  for (void** b : {&m->d_gathered, &m->d_frame})
    if (*b) (void)hipFree(*b), *b = nullptr;
  m->width = m->height = 0;
}
#define HIP_OK(expr, p)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (p).rc = VPT_ERR_HIP, (p).error = std::string(#expr ": ") + hipGetErrorString(e_);      \
      return;                                                                                 \
    }                                                                                         \
  } while (0)
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
  if (ndev > 1 && m->distinct) {
    std::string why;
    if (!load_rccl(m->rccl, why)) {
      vpt_multi_destroy(m);
      return vpt_set_error(VPT_ERR_HIP, "%s", why.c_str());
    }
    m->comms.assign((size_t)ndev, nullptr);
    if (int r = m->rccl.CommInitAll(m->comms.data(), ndev, devices)) {
      m->comms.clear();
      vpt_multi_destroy(m);
      return vpt_set_error(VPT_ERR_HIP, "ncclCommInitAll: %s", m->rccl.GetErrorString ? m->rccl.GetErrorString(r) : "failed");
    }
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

int vpt_multi_render(vpt_multi* m, const vpt_params* params, int nsamples, int width, int height, float* image_rgba,
    int32_t* hits, uint64_t* rng, int* samples_io) {
  if (!m || !params || !image_rgba || !hits || !rng || !samples_io) return vpt_set_error(VPT_ERR_INVALID_ARG, "null argument");
  if (width <= 0 || height <= 0) return vpt_set_error(VPT_ERR_INVALID_ARG, "bad image size");
  if (params->shader < 0 || params->shader > VPT_SHADER_IMPLICIT_NORMAL) return vpt_set_error(VPT_ERR_UNKNOWN_SHADER, "sampler unknown");
  int todo = params->samples - *samples_io;   // no-op once reached, yocto_pathtrace.cpp:1055
  if (nsamples < todo) todo = nsamples;
  if (todo <= 0) return VPT_OK;
  const int ndev = (int)m->parts.size();
  if (m->width != width || m->height != height) {
    free_buffers(m);
    for (int i = 0; i < ndev; i++) {
      auto& p = m->parts[(size_t)i];
      slot_map(width, height, i, ndev, p.pixel_of_slot);
      size_t n = p.pixel_of_slot.size();
      p.h_image.assign(4 * n, 0.0f), p.h_hits.assign(n, 0), p.h_rng.assign(2 * n, 0);
      if (hipSetDevice(p.device) != hipSuccess || hipMalloc(&p.d_image, n * 16) != hipSuccess || hipMalloc(&p.d_hits, n * 4) != hipSuccess ||
          hipMalloc(&p.d_rng, n * 16) != hipSuccess) {
        free_buffers(m);
        return vpt_set_error(VPT_ERR_HIP, "cannot allocate the tile state on device %d", p.device);
      }
    }
    m->width = width, m->height = height;
  }
  // one host thread per GPU: gather its tiles' state from the caller's arrays, upload, render, download, scatter back
  auto work = [&](int i) {
    auto&  p = m->parts[(size_t)i];
    size_t n = p.pixel_of_slot.size();
    for (size_t s = 0; s < n; s++) {
      int px = p.pixel_of_slot[s];
      if (px < 0) continue;
      memcpy(&p.h_image[4 * s], image_rgba + 4 * (size_t)px, 16);
      p.h_hits[s] = hits[px];
      p.h_rng[2 * s] = rng[2 * (size_t)px], p.h_rng[2 * s + 1] = rng[2 * (size_t)px + 1];
    }
    HIP_OK(hipSetDevice(p.device), p);
    HIP_OK(hipMemcpyAsync(p.d_image, p.h_image.data(), n * 16, hipMemcpyHostToDevice, p.stream), p);
    HIP_OK(hipMemcpyAsync(p.d_hits, p.h_hits.data(), n * 4, hipMemcpyHostToDevice, p.stream), p);
    HIP_OK(hipMemcpyAsync(p.d_rng, p.h_rng.data(), n * 16, hipMemcpyHostToDevice, p.stream), p);
    vpt_layout lay = {width, height, 8, 8, i, ndev};
    if (int rc = vpt_render_device(p.scene, params, &lay, todo, p.d_image, p.d_hits, p.d_rng, p.stream)) {
      p.rc = rc, p.error = vpt_last_error();
      return;
    }
    HIP_OK(hipMemcpyAsync(p.h_image.data(), p.d_image, n * 16, hipMemcpyDeviceToHost, p.stream), p);
    HIP_OK(hipMemcpyAsync(p.h_hits.data(), p.d_hits, n * 4, hipMemcpyDeviceToHost, p.stream), p);
    HIP_OK(hipMemcpyAsync(p.h_rng.data(), p.d_rng, n * 16, hipMemcpyDeviceToHost, p.stream), p);
    HIP_OK(hipStreamSynchronize(p.stream), p);
    for (size_t s = 0; s < n; s++) {   // tiles are disjoint between devices: the threads write different pixels
      int px = p.pixel_of_slot[s];
      if (px < 0) continue;
      memcpy(image_rgba + 4 * (size_t)px, &p.h_image[4 * s], 16);
      hits[px] = p.h_hits[s];
      rng[2 * (size_t)px] = p.h_rng[2 * s], rng[2 * (size_t)px + 1] = p.h_rng[2 * s + 1];
    }
  };
  for (auto& p : m->parts) p.rc = VPT_OK, p.error.clear();
  std::vector<std::thread> threads;
  for (int i = 1; i < ndev; i++) threads.emplace_back(work, i);
  work(0);
  for (auto& t : threads) t.join();
  for (auto& p : m->parts)
    if (p.rc != VPT_OK) return vpt_set_error(p.rc, "device %d: %s", p.device, p.error.c_str());
  *samples_io += todo;
  m->samples = *samples_io;
  return VPT_OK;
}

int vpt_multi_get_render(vpt_multi* m, float* image_rgba) {
  if (!m || !image_rgba) return vpt_set_error(VPT_ERR_INVALID_ARG, "null argument");
  if (m->width <= 0 || m->samples <= 0) return vpt_set_error(VPT_ERR_INVALID_ARG, "nothing rendered yet");
  const int ndev = (int)m->parts.size();
  auto&     p0   = m->parts[0];
  size_t    n    = p0.pixel_of_slot.size();   // slots per device (the same on every device)
  if (hipSetDevice(p0.device) != hipSuccess) return vpt_set_error(VPT_ERR_HIP, "hipSetDevice failed");
  if (!m->d_gathered) {
    if (hipMalloc(&m->d_gathered, (size_t)ndev * n * 16) != hipSuccess || hipMalloc(&m->d_frame, (size_t)m->width * m->height * 16) != hipSuccess)
      return vpt_set_error(VPT_ERR_HIP, "cannot allocate the gather buffers on device %d", p0.device);
  }
  // the tile buffers of all devices -> devices[0]: its own by a local copy, the others over xGMI
  if (hipMemcpyAsync(m->d_gathered, p0.d_image, n * 16, hipMemcpyDeviceToDevice, p0.stream) != hipSuccess)
    return vpt_set_error(VPT_ERR_HIP, "local tile copy failed");
  if (ndev > 1 && m->distinct) {
    int r = m->rccl.GroupStart();
    for (int i = 1; i < ndev && r == ncclSuccess; i++) {
      r = m->rccl.Send(m->parts[(size_t)i].d_image, n * 4, ncclFloat, 0, m->comms[(size_t)i], m->parts[(size_t)i].stream);
      if (r == ncclSuccess) r = m->rccl.Recv((char*)m->d_gathered + (size_t)i * n * 16, n * 4, ncclFloat, i, m->comms[0], p0.stream);
    }
    int e = m->rccl.GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) return vpt_set_error(VPT_ERR_HIP, "RCCL tile gather: %s", m->rccl.GetErrorString ? m->rccl.GetErrorString(r) : "failed");
    for (int i = 1; i < ndev; i++) {   // the sends ran on the senders' streams
      (void)hipSetDevice(m->parts[(size_t)i].device);
      if (hipStreamSynchronize(m->parts[(size_t)i].stream) != hipSuccess) return vpt_set_error(VPT_ERR_HIP, "send stream failed");
    }
    (void)hipSetDevice(p0.device);
  } else {
    for (int i = 1; i < ndev; i++)   // the same physical device listed more than once (tests): plain copies
      if (hipMemcpyAsync((char*)m->d_gathered + (size_t)i * n * 16, m->parts[(size_t)i].d_image, n * 16, hipMemcpyDeviceToDevice, p0.stream) != hipSuccess)
        return vpt_set_error(VPT_ERR_HIP, "tile copy failed");
  }
  vpt_layout lay = {m->width, m->height, 8, 8, 0, ndev};
  if (int rc = vpt_resolve_device(&lay, m->d_gathered, m->samples, m->d_frame, p0.stream)) return rc;
  if (hipMemcpyAsync(image_rgba, m->d_frame, (size_t)m->width * m->height * 16, hipMemcpyDeviceToHost, p0.stream) != hipSuccess ||
      hipStreamSynchronize(p0.stream) != hipSuccess)
    return vpt_set_error(VPT_ERR_HIP, "frame download failed");
  return VPT_OK;
}

}  // extern "C"
