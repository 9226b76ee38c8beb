// PROSAC ordering on the device: the first top_k positions of "indices sorted by weight, descending"
// (reference pose/Utility.hpp:107-118 sortIndexes, consumed through getSortedIdx, pose/AOOnlyPoseAdapter.hpp:233-254) for a
// dense frame's worth of weights.  PROSAC reads only a prefix of the order (pose/Utility.hpp ProsacSampler: n grows by at most
// one per draw), so this is a TOP-K SELECT + SORT, not a full sort:
//   1. keys: (weight, index) packed into one 64-bit integer whose ASCENDING order is "weight descending, index ascending" -- the
//      total order the host path uses (Utility.hpp rpe::prosac_key; ties to the lower index), so the result is THE host's prefix;
//   2. two-level radix select of the cut: histogram of the keys' top 11 bits (LDS histogram per workgroup, one global add per
//      non-empty bin), one thread finds the bin in which the running count reaches top_k, then the same on the next 11 bits inside
//      that bin -- the cut is known to 22 bits (sign, exponent, 13 mantissa bits of the weight);
//   3. every key up to the cut is compacted into a candidate list (order irrelevant): top_k plus the few keys sharing the cut's prefix;
//   4. ONE workgroup sorts the candidates in LDS (bitonic, <= 8192 keys of 8 bytes = 64 KiB) and writes the first top_k indices.
// Integer work only: bit-exact by construction, checked against the host order in tests/test_gpu_prosac_order.py.  If the crossing
// bin holds so many (near-)equal weights that the candidates exceed the LDS sort, the call reports it and the host path takes over.
#include "rpe_kernels.h"

namespace rpe {
namespace {

constexpr int kBins = 2048;
constexpr int kSortCap = 8192;   // 64 KiB of LDS

__device__ __forceinline__ unsigned long long prosac_key(float w, unsigned int index) {
  w += 0.0f;   // -0 -> +0: equal weights compare equal as integers too
  unsigned int u = __float_as_uint(w);
  u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;   // order-preserving map of IEEE floats to unsigned
  return ((unsigned long long)(~u) << 32) | index;
}

// level 0: histogram of the keys' bits 63..53 over all keys; level 1: of bits 52..42 over the keys whose bits 63..53 equal ctl[0]
__global__ __launch_bounds__(256) void prosac_hist_kernel(const float* __restrict__ w, int n, unsigned int* __restrict__ hist,
                             const unsigned int* __restrict__ ctl, int level) {
  __shared__ unsigned int h[kBins];
  for (int i = threadIdx.x; i < kBins; i += 256) h[i] = 0;
  __syncthreads();
  const unsigned int coarse = level ? ctl[0] : 0u;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const unsigned long long k = prosac_key(w[i], (unsigned int)i);
    if (level == 0) atomicAdd(&h[(unsigned int)(k >> 53)], 1u);
    else if ((unsigned int)(k >> 53) == coarse) atomicAdd(&h[(unsigned int)(k >> 42) & (kBins - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kBins; i += 256) if (h[i]) atomicAdd(&hist[i], h[i]);
}

// level 0: ctl[0] = coarse bin in which the running count reaches top_k, ctl[3] = keys in the bins before it.
// level 1: ctl[1] = fine bin (inside the coarse one) in which it does, ctl[4] = number of candidates = keys up to and including that
// fine bin;
//          ctl[2] = compaction cursor (zeroed).  The histogram is left zero for the next pass / call.
__global__ __launch_bounds__(256) void prosac_pick_kernel(unsigned int* __restrict__ hist, int top_k, unsigned int* __restrict__ ctl,
    int level) {
  __shared__ unsigned int h[kBins];
  __shared__ unsigned int part[256];
  unsigned int sum = 0;
  for (int i = 0; i < kBins / 256; i++) { const unsigned int v = hist[threadIdx.x * (kBins / 256) + i];
      h[threadIdx.x * (kBins / 256) + i] = v; sum += v; }
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int run = level ? ctl[3] : 0u, before = run;
    int bin = kBins - 1, grp = 255;
    for (int g = 0; g < 256; g++) { if (run + part[g] >= (unsigned int)top_k) { grp = g; break; } run += part[g]; }
    for (int b = grp * (kBins / 256); b < kBins; b++) {
      before = run;
      run += h[b];
      if (run >= (unsigned int)top_k) { bin = b; break; }
    }
    if (level == 0) { ctl[0] = (unsigned int)bin; ctl[3] = before; }
    else { ctl[1] = (unsigned int)bin; ctl[4] = run; ctl[2] = 0; }
  }
  for (int i = threadIdx.x; i < kBins; i += 256) hist[i] = 0;   // left zero for the next pass / call
}

__global__ __launch_bounds__(256) void prosac_compact_kernel(const float* __restrict__ w, int n, unsigned int* __restrict__ ctl,
                                                             unsigned long long* __restrict__ cand, int cap) {
  const unsigned int cut = (ctl[0] << 11) | ctl[1];   // 22-bit prefix of the last key kept
  if (ctl[4] > (unsigned int)cap) return;             // too many (near-)equal weights around the cut: the host path takes over
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const unsigned long long k = prosac_key(w[i], (unsigned int)i);
    if ((unsigned int)(k >> 42) <= cut) cand[atomicAdd(&ctl[2], 1u)] = k;
  }
}

__global__ __launch_bounds__(1024) void prosac_sort_kernel(const unsigned long long* __restrict__ cand,
    const unsigned int* __restrict__ ctl, int top_k,
                                                           int* __restrict__ order, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const unsigned int count = ctl[4];
  if (count > (unsigned int)kSortCap
      || count < (unsigned int)top_k) { if (threadIdx.x == 0) *status = count > (unsigned int)kSortCap ? 1 : 2; return; }
  unsigned int m = 1;
  while (m < count) m <<= 1;
  for (unsigned int i = threadIdx.x; i < m; i += 1024) keys[i] = i < count ? cand[i] : ~0ull;
  __syncthreads();
  for (unsigned int k = 2; k <= m; k <<= 1)
    for (unsigned int j = k >> 1; j > 0; j >>= 1) {
      for (unsigned int i = threadIdx.x; i < m; i += 1024) {
        const unsigned int l = i ^ j;
        if (l > i) {
          const unsigned long long a = keys[i], b = keys[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[l] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < top_k; i += 1024) order[i] = (int)(unsigned int)keys[i];
  if (threadIdx.x == 0) *status = 0;
}

}  // namespace

// d_w: n floats in HBM.  d_hist: kBins uints (zero on entry; left zero), d_ctl: 8 uints, d_cand: kSortCap keys, d_order: top_k ints,
// d_status: 0 ok, 1 too many candidates (ties), 2 internal count mismatch.
hipError_t launch_prosac_order(const float* d_w, int n, int top_k, unsigned int* d_hist, unsigned int* d_ctl, unsigned long long* d_cand,
                               int* d_order, int* d_status, hipStream_t s) {
  const int G = n >= 256 * 512 ? 512 : (n + 255) / 256;
  hipLaunchKernelGGL(prosac_hist_kernel, dim3(G), dim3(256), 0, s, d_w, n, d_hist, d_ctl, 0);
  hipLaunchKernelGGL(prosac_pick_kernel, dim3(1), dim3(256), 0, s, d_hist, top_k, d_ctl, 0);
  hipLaunchKernelGGL(prosac_hist_kernel, dim3(G), dim3(256), 0, s, d_w, n, d_hist, d_ctl, 1);
  hipLaunchKernelGGL(prosac_pick_kernel, dim3(1), dim3(256), 0, s, d_hist, top_k, d_ctl, 1);
  hipLaunchKernelGGL(prosac_compact_kernel, dim3(G), dim3(256), 0, s, d_w, n, d_ctl, d_cand, kSortCap);
  hipLaunchKernelGGL(prosac_sort_kernel, dim3(1), dim3(1024), (size_t)kSortCap * sizeof(unsigned long long), s, d_cand, d_ctl, top_k,
      d_order, d_status);
  return hipGetLastError();
}

void preload_prosac() {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, (const void*)prosac_pick_kernel) != hipSuccess) (void)hipGetLastError();
}

}  // namespace rpe
