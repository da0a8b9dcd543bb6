// vpt_bvh_build.hip — the reference's BVH build on the device: build_bvh(bvh, bboxes, highquality = false)
// (libs/yocto/yocto_bvh.cpp:447-507) with split_middle (:411-441), reproduced NODE FOR NODE: same node array (ids in the
// reference's creation order), same primitive order, same float bits.  SURVEY §8(f) row 4; include/vpt.h: vpt_build_bvh.
//
// The reference builds depth first from an explicit stack; a node's ids and its primitives' order are by-products of that
// order and of libstdc++'s std::partition.  Here the tree grows one LEVEL per round over all of the level's nodes at
// once, and three observations make the result identical:
//
//  * bounds.  merge() folds min(a, b) = a < b ? a : b and max(a, b) = a > b ? a : b over the node's primitives in array
//    order, so among equal candidates the LAST one wins - visible only as the sign of a zero.  A parallel reduction over
//    the key (value with -0 == +0, position in the primitive array) with the later position preferred picks the same
//    element; the result's bits are then read from that element.  Keys are 64-bit: atomicMin / atomicMax.
//  * partition.  libstdc++'s std::partition (bidirectional form, stl_algo.h __partition) walks inwards from both ends and
//    swaps the k-th element from the left that fails the predicate with the k-th from the right that passes it, while
//    the former is left of the latter.  If the range holds T passing elements, these are exactly the failing elements at
//    positions < start + T and the passing ones at positions >= start + T, paired in that order; nothing else moves.
//    With an exclusive scan of the predicate over the whole primitive array every element finds its rank, hence its
//    partner, on its own.
//  * node ids.  The reference appends a node's two children when the node is popped, and pops the right child first: ids
//    follow the pre-order that visits right subtrees first.  With icount(X) = internal nodes in X's subtree and
//    pre(X) = internal nodes visited before X: pre(right) = pre(P) + 1, pre(left) = pre(P) + 1 + icount(right), and the
//    children of X sit at 1 + 2 pre(X), 2 + 2 pre(X).  Two sweeps over the levels (up for icount, down for pre).
//
// Work per level: O(n) threads, a handful of 64-bit atomics per live primitive, one rocPRIM scan.  The host reads back
// one counter per level (the number of nodes the level created).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "vpt.h"

int vpt_set_error(int code, const char* fmt, ...);   // vpt_capi.hip: records the message for vpt_last_error() on this thread

namespace {

#define BVH_TRY(expr)                                                                                    \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return vpt_set_error(VPT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));     \
  } while (0)

constexpr int BVH_MAX_PRIMS = 4;   // yocto_bvh.cpp:444

// a node under construction, in creation (level) order
struct tnode {
  int start, end;      // its range of the primitive array
  int parent, which;   // temp id of the parent (-1: root), 0 = left / 1 = right child
  int child;           // temp id of the left child (right = child + 1), -1: leaf
  int axis, mid, swap; // split_middle's result; swap: the partition moves elements
  float split;
  int icount, pre;     // internal nodes in the subtree / visited before this node in the reference's order
};
struct tkeys {   // reduction keys: [0..2] min of bbox.min, [3..5] max of bbox.max, [6..8] min of centres, [9..11] max of centres
  unsigned long long k[12];
};

__device__ __forceinline__ unsigned ordered(float f) {   // monotone float -> unsigned, -0 and +0 alike
  unsigned u = __float_as_uint(f == 0.0f ? 0.0f : f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long min_key(float f, int pos) { return (unsigned long long)ordered(f) << 32 | (0xffffffffu - (unsigned)pos); }
__device__ __forceinline__ unsigned long long max_key(float f, int pos) { return (unsigned long long)ordered(f) << 32 | (unsigned)pos; }
__device__ __forceinline__ int min_pos(unsigned long long k) { return (int)(0xffffffffu - (unsigned)(k & 0xffffffffull)); }
__device__ __forceinline__ int max_pos(unsigned long long k) { return (int)(k & 0xffffffffull); }

__global__ void k_centers(int n, const float* __restrict__ bb, float* __restrict__ ctr) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int c = 0; c < 3; c++) ctr[3 * i + c] = (bb[6 * i + c] + bb[6 * i + 3 + c]) / 2;   // center(bbox), yocto_geometry.h: (min + max) / 2
}
__global__ void k_init(int n, int* prims, int* node_of) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) prims[i] = i, node_of[i] = 0;
}
__global__ void k_reset_keys(int lb, int le, tkeys* keys) {
  int t = lb + blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= le) return;
  for (int c = 0; c < 3; c++) keys[t].k[c] = keys[t].k[6 + c] = ~0ull, keys[t].k[3 + c] = keys[t].k[9 + c] = 0ull;
}
// bounds of every live node: each primitive position offers its box and its centre to its node.  Positions of a node are
// contiguous, so near the root all 64 lanes of a wave belong to the same node: they fold their keys across the wave first and
// lane 0 makes the 12 atomic updates (without this the first levels serialise n atomics on 12 addresses).
__device__ __forceinline__ unsigned long long wave_min(unsigned long long v) {
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned long long o = __shfl_xor(v, m, 64);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ unsigned long long wave_max(unsigned long long v) {
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned long long o = __shfl_xor(v, m, 64);
    v = o > v ? o : v;
  }
  return v;
}
__global__ void k_bounds(int n, const int* __restrict__ prims, const int* __restrict__ node_of, const float* __restrict__ bb,
    const float* __restrict__ ctr, tkeys* keys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int t = i < n ? node_of[i] : -1;
  int t0 = __builtin_amdgcn_readfirstlane(t);
  bool uniform = __builtin_amdgcn_ballot_w64(t != t0) == 0;   // the whole wave (blockDim is a multiple of 64) in one node
  if (uniform && t0 < 0) return;
  unsigned long long k[12];
  if (t >= 0) {
    int p = prims[i];
    for (int c = 0; c < 3; c++) {
      k[c]     = min_key(bb[6 * p + c], i);
      k[3 + c] = max_key(bb[6 * p + 3 + c], i);
      k[6 + c] = min_key(ctr[3 * p + c], i);
      k[9 + c] = max_key(ctr[3 * p + c], i);
    }
  }
  if (uniform) {
    for (int c = 0; c < 3; c++) k[c] = wave_min(k[c]), k[3 + c] = wave_max(k[3 + c]), k[6 + c] = wave_min(k[6 + c]), k[9 + c] = wave_max(k[9 + c]);
    if ((threadIdx.x & 63) != 0) return;
  } else if (t < 0) return;
  for (int c = 0; c < 3; c++) {
    atomicMin(&keys[t].k[c], k[c]);
    atomicMax(&keys[t].k[3 + c], k[3 + c]);
    atomicMin(&keys[t].k[6 + c], k[6 + c]);
    atomicMax(&keys[t].k[9 + c], k[9 + c]);
  }
}
// the node's box (bits of the winning elements) and split_middle's choice of axis and plane
__global__ void k_decide(int lb, int le, tnode* nodes, const tkeys* keys, const int* __restrict__ prims, const float* __restrict__ bb,
    const float* __restrict__ ctr, float* boxes) {
  int t = lb + blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= le) return;
  tnode& nd = nodes[t];
  for (int c = 0; c < 3; c++) {
    boxes[6 * t + c]     = bb[6 * prims[min_pos(keys[t].k[c])] + c];
    boxes[6 * t + 3 + c] = bb[6 * prims[max_pos(keys[t].k[3 + c])] + 3 + c];
  }
  nd.child = -1, nd.axis = 0, nd.swap = 0, nd.mid = (nd.start + nd.end) / 2, nd.split = 0;
  if (nd.end - nd.start <= BVH_MAX_PRIMS) return;   // a leaf
  float cmin[3], cmax[3], csize[3];
  for (int c = 0; c < 3; c++) {
    cmin[c]  = ctr[3 * prims[min_pos(keys[t].k[6 + c])] + c];
    cmax[c]  = ctr[3 * prims[max_pos(keys[t].k[9 + c])] + c];
    csize[c] = cmax[c] - cmin[c];
  }
  nd.child = -2;   // internal; the children are allocated by k_children
  if (csize[0] == 0 && csize[1] == 0 && csize[2] == 0) return;   // :422: halves, axis 0
  int axis = 0;
  if (csize[0] >= csize[1] && csize[0] >= csize[2]) axis = 0;
  if (csize[1] >= csize[0] && csize[1] >= csize[2]) axis = 1;
  if (csize[2] >= csize[0] && csize[2] >= csize[1]) axis = 2;
  nd.axis = axis, nd.split = (cmin[axis] + cmax[axis]) / 2, nd.swap = 1;
}
__global__ void k_flags(int n, const int* __restrict__ prims, const int* __restrict__ node_of, const tnode* __restrict__ nodes,
    const float* __restrict__ ctr, int* flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  int f = 0;
  if (i < n) {
    int t = node_of[i];
    if (t >= 0 && nodes[t].swap) f = ctr[3 * prims[i] + nodes[t].axis] < nodes[t].split ? 1 : 0;
  }
  flag[i] = f;   // flag[n] = 0: the scan's last entry is the total
}
// split position and the two children of every internal node of the level
__global__ void k_children(int lb, int le, tnode* nodes, const int* __restrict__ tscan, int* counter) {
  int t = lb + blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= le) return;
  tnode& nd = nodes[t];
  if (nd.child == -1) return;
  if (nd.swap) {
    int mid = nd.start + (tscan[nd.end] - tscan[nd.start]);
    if (mid == nd.start || mid == nd.end) nd.swap = 0;   // :436: could not split: halves, nothing moved
    else nd.mid = mid;
  }
  int c = atomicAdd(counter, 2);
  nd.child = c;
  nodes[c]     = tnode{nd.start, nd.mid, t, 0, -1, 0, 0, 0, 0.0f, 0, 0};
  nodes[c + 1] = tnode{nd.mid, nd.end, t, 1, -1, 0, 0, 0, 0.0f, 0, 0};
}
// std::partition, step 1: every passing element right of the split position notes where it is, by its rank from the right
__global__ void k_partners(int n, const int* __restrict__ node_of, const tnode* __restrict__ nodes, const int* __restrict__ flag,
    const int* __restrict__ tscan, int* partner) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int t = node_of[i];
  if (t < 0 || !nodes[t].swap) return;
  const tnode& nd = nodes[t];
  if (i >= nd.mid && flag[i]) partner[nd.start + (tscan[nd.end] - tscan[i + 1])] = i;
}
// step 2: every failing element left of the split position swaps with the passing element of its rank; then every
// position learns which child it now belongs to
__global__ void k_swap(int n, int* prims, int* node_of, const tnode* __restrict__ nodes, const int* __restrict__ flag,
    const int* __restrict__ tscan, const int* __restrict__ partner) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int t = node_of[i];
  if (t < 0) return;
  const tnode& nd = nodes[t];
  if (nd.swap && i < nd.mid && !flag[i]) {
    int k = (i - nd.start) - (tscan[i] - tscan[nd.start]);
    int j = partner[nd.start + k];
    int a = prims[i];
    prims[i] = prims[j], prims[j] = a;
  }
  node_of[i] = nd.child < 0 ? -1 : (i < nd.mid ? nd.child : nd.child + 1);
}
__global__ void k_icount(int lb, int le, tnode* nodes) {
  int t = lb + blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= le) return;
  tnode& nd = nodes[t];
  nd.icount = nd.child < 0 ? 0 : 1 + nodes[nd.child].icount + nodes[nd.child + 1].icount;
}
__global__ void k_pre(int lb, int le, tnode* nodes) {
  int t = lb + blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= le) return;
  const tnode& nd = nodes[t];
  if (nd.child < 0) return;
  nodes[nd.child + 1].pre = nd.pre + 1;
  nodes[nd.child].pre     = nd.pre + 1 + nodes[nd.child + 1].icount;
}
__global__ void k_emit(int count, const tnode* __restrict__ nodes, const float* __restrict__ boxes, vpt_bvh_node* out) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const tnode& nd = nodes[t];
  int id = nd.parent < 0 ? 0 : 1 + 2 * nodes[nd.parent].pre + nd.which;
  vpt_bvh_node o;
  for (int c = 0; c < 3; c++) o.bbox_min[c] = boxes[6 * t + c], o.bbox_max[c] = boxes[6 * t + 3 + c];
  if (nd.child >= 0) o.start = 1 + 2 * nd.pre, o.num = 2, o.axis = (int8_t)nd.axis, o.internal = 1;
  else o.start = nd.start, o.num = (int16_t)(nd.end - nd.start), o.axis = 0, o.internal = 0;
  out[id] = o;
}

struct dev_buffers {   // freed on every exit path
  std::vector<void*> ptrs;
  ~dev_buffers() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <typename T>
  hipError_t alloc(T** p, size_t count) {
    hipError_t e = hipMalloc((void**)p, (count ? count : 1) * sizeof(T));
    if (e == hipSuccess) ptrs.push_back(*p);
    return e;
  }
};

}  // namespace

extern "C" int vpt_build_bvh(int device, const float* bboxes, int n, vpt_bvh_node* nodes_out, int capacity, int* num_nodes, int* primitives) {
  if (n < 0 || !nodes_out || !num_nodes || (n > 0 && (!bboxes || !primitives))) return vpt_set_error(VPT_ERR_INVALID_ARG, "bad argument");
  if (capacity < (n > 0 ? 2 * n - 1 : 1)) return vpt_set_error(VPT_ERR_INVALID_ARG, "node capacity %d < %d (2 n - 1)", capacity, n > 0 ? 2 * n - 1 : 1);
  if (n == 0) {   // the reference's root of an empty build: an invalid box, a leaf without primitives
    vpt_bvh_node o;
    for (int c = 0; c < 3; c++) o.bbox_min[c] = 3.402823466e+38f, o.bbox_max[c] = -3.402823466e+38f;
    o.start = 0, o.num = 0, o.axis = 0, o.internal = 0;
    nodes_out[0] = o, *num_nodes = 1;
    return VPT_OK;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return vpt_set_error(VPT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return vpt_set_error(VPT_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
  BVH_TRY(hipSetDevice(device));

  dev_buffers B;
  float *bb = nullptr, *ctr = nullptr, *boxes = nullptr;
  int *  prims = nullptr, *node_of = nullptr, *flag = nullptr, *tscan = nullptr, *partner = nullptr, *counter = nullptr;
  tnode* nodes = nullptr;
  tkeys* keys  = nullptr;
  vpt_bvh_node* out = nullptr;
  const size_t cap = 2 * (size_t)n;
  BVH_TRY(B.alloc(&bb, 6 * (size_t)n));
  BVH_TRY(B.alloc(&ctr, 3 * (size_t)n));
  BVH_TRY(B.alloc(&boxes, 6 * cap));
  BVH_TRY(B.alloc(&prims, (size_t)n));
  BVH_TRY(B.alloc(&node_of, (size_t)n));
  BVH_TRY(B.alloc(&flag, (size_t)n + 1));
  BVH_TRY(B.alloc(&tscan, (size_t)n + 1));
  BVH_TRY(B.alloc(&partner, (size_t)n));
  BVH_TRY(B.alloc(&counter, 1));
  BVH_TRY(B.alloc(&nodes, cap));
  BVH_TRY(B.alloc(&keys, cap));
  BVH_TRY(B.alloc(&out, cap));
  size_t scan_bytes = 0;
  BVH_TRY(rocprim::exclusive_scan((void*)nullptr, scan_bytes, flag, tscan, 0, (size_t)n + 1, rocprim::plus<int>()));
  char* scan_temp = nullptr;
  BVH_TRY(B.alloc(&scan_temp, scan_bytes));

  const int  TB = 256;
  const dim3 gn((n + TB - 1) / TB), gn1((n + 1 + TB - 1) / TB);
  auto blocks = [&](int count) { return dim3((count + TB - 1) / TB); };
  BVH_TRY(hipMemcpy(bb, bboxes, 6 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_centers, gn, dim3(TB), 0, 0, n, bb, ctr);
  hipLaunchKernelGGL(k_init, gn, dim3(TB), 0, 0, n, prims, node_of);
  tnode root = {0, n, -1, 0, -1, 0, 0, 0, 0.0f, 0, 0};
  BVH_TRY(hipMemcpy(nodes, &root, sizeof(root), hipMemcpyHostToDevice));
  int count = 1;
  BVH_TRY(hipMemcpy(counter, &count, 4, hipMemcpyHostToDevice));

  std::vector<int> level_begin = {0};
  int lb = 0, le = 1;
  while (lb < le) {
    if (level_begin.size() > 4096) return vpt_set_error(VPT_ERR_HIP, "BVH build did not terminate");
    int ln = le - lb;
    hipLaunchKernelGGL(k_reset_keys, blocks(ln), dim3(TB), 0, 0, lb, le, keys);
    hipLaunchKernelGGL(k_bounds, gn, dim3(TB), 0, 0, n, prims, node_of, bb, ctr, keys);
    hipLaunchKernelGGL(k_decide, blocks(ln), dim3(TB), 0, 0, lb, le, nodes, keys, prims, bb, ctr, boxes);
    hipLaunchKernelGGL(k_flags, gn1, dim3(TB), 0, 0, n, prims, node_of, nodes, ctr, flag);
    BVH_TRY(rocprim::exclusive_scan((void*)scan_temp, scan_bytes, flag, tscan, 0, (size_t)n + 1, rocprim::plus<int>()));
    hipLaunchKernelGGL(k_children, blocks(ln), dim3(TB), 0, 0, lb, le, nodes, tscan, counter);
    hipLaunchKernelGGL(k_partners, gn, dim3(TB), 0, 0, n, node_of, nodes, flag, tscan, partner);
    hipLaunchKernelGGL(k_swap, gn, dim3(TB), 0, 0, n, prims, node_of, nodes, flag, tscan, partner);
    BVH_TRY(hipGetLastError());
    BVH_TRY(hipMemcpy(&count, counter, 4, hipMemcpyDeviceToHost));   // also the level's barrier for the host
    if ((size_t)count > cap) return vpt_set_error(VPT_ERR_HIP, "BVH build produced %d nodes for %d primitives", count, n);
    lb = le, le = count;
    level_begin.push_back(lb);
  }
  // level_begin = starts of the levels, the last entry = count
  const int nlevels = (int)level_begin.size() - 1;
  for (int l = nlevels - 1; l >= 0; l--) {
    int a = level_begin[l], b = level_begin[l + 1];
    if (b > a) hipLaunchKernelGGL(k_icount, blocks(b - a), dim3(TB), 0, 0, a, b, nodes);
  }
  for (int l = 0; l < nlevels; l++) {
    int a = level_begin[l], b = level_begin[l + 1];
    if (b > a) hipLaunchKernelGGL(k_pre, blocks(b - a), dim3(TB), 0, 0, a, b, nodes);
  }
  hipLaunchKernelGGL(k_emit, blocks(count), dim3(TB), 0, 0, count, nodes, boxes, out);
  BVH_TRY(hipGetLastError());
  BVH_TRY(hipMemcpy(nodes_out, out, (size_t)count * sizeof(vpt_bvh_node), hipMemcpyDeviceToHost));
  BVH_TRY(hipMemcpy(primitives, prims, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  *num_nodes = count;
  return VPT_OK;
}
