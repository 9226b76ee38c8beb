// One-way hand-off latency between two workgroups of one launch on gfx950, by where they run (same XCD = same L2, or different XCDs) and by
// the cache-policy bits on the store and on the polling load.  Development aid for the resident kernels' hand-off (never infer these from
// a table for another part):  hipcc --offload-arch=gfx950 -O3 -o xcd_handoff xcd_handoff.hip && ./xcd_handoff
// Workgroup a stores k into its flag, workgroup b polls it, answers in its own flag, a polls that: `iters` round trips, timed with the
// 100 MHz s_memrealtime of workgroup a; one-way = total / (2 iters).  A poll that never sees the value (a stale line in a cache that nobody
// invalidates) gives up after `kSpins` loads and the combination is reported as "stale".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

static const int kSpins = 400000;

template <int ST> __device__ __forceinline__ void put(unsigned long long* p, unsigned long long v) {
  if (ST == 0) asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  if (ST == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  if (ST == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  if (ST == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  if (ST == 4) asm volatile("global_store_dwordx2 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
template <int LD> __device__ __forceinline__ unsigned long long get(const unsigned long long* p) {
  unsigned long long v;
  if (LD == 0) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  if (LD == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  if (LD == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  if (LD == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  if (LD == 4) asm volatile("buffer_inv sc0\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  if (LD == 5) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}

// out[0..grid): XCC id of each workgroup; out[64]: ticks of the exchange (100 MHz); out[65]: 1 if a poll gave up
template <int ST, int LD>
__global__ __launch_bounds__(64) void pingpong(unsigned long long* flags, unsigned long long* out, int a, int b, int iters, unsigned long long base) {
  unsigned int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = xcc & 15;
  if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
  if (threadIdx.x != 0) return;
  unsigned long long* mine = flags + ((int)blockIdx.x == a ? 0 : 64);
  const unsigned long long* theirs = flags + ((int)blockIdx.x == a ? 64 : 0);
  unsigned long long t0 = 0, t1 = 0;
  bool lost = false;
  if ((int)blockIdx.x == a) {
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
    for (int k = 1; k <= iters && !lost; k++) {
      put<ST>(mine, base + k);
      int spins = 0;
      while (get<LD>(theirs) != base + k) if (++spins > kSpins) { lost = true; break; }
    }
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory");
    out[64] = t1 - t0;
    out[65] = lost;
  } else {
    for (int k = 1; k <= iters && !lost; k++) {
      int spins = 0;
      while (get<LD>(theirs) != base + k) if (++spins > kSpins) { lost = true; break; }
      put<ST>(mine, base + k);
    }
  }
}

static const char* st_name[] = {"store", "store sc0", "store sc1", "store sc0 sc1", "store nt"};
static const char* ld_name[] = {"load", "load sc0", "load sc1", "load sc0 sc1", "buffer_inv sc0 + load", "load nt"};
static unsigned long long g_base = 0;

template <int ST, int LD> static void one(unsigned long long* d_flags, unsigned long long* d_out, int a, int b, const char* where, bool fine) {
  const int iters = 2000, grid = 32;
  std::vector<unsigned long long> h(66);
  double best = 1e30; bool lost = false;
  for (int rep = 0; rep < 3; rep++) {
    g_base += 1u << 20;
    hipLaunchKernelGGL((pingpong<ST, LD>), dim3(grid), dim3(64), 0, 0, d_flags, d_out, a, b, iters, g_base);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * 66, hipMemcpyDeviceToHost);
    lost |= h[65] != 0;
    const double ns = (double)h[64] * 10.0 / (2.0 * iters);
    if (ns < best) best = ns;
  }
  printf("{\"memory\": \"%s\", \"pair\": \"%s\", \"wg\": [%d, %d], \"xcc\": [%d, %d], \"store\": \"%s\", \"load\": \"%s\", \"one_way_ns\": %s%.1f}\n", fine ? "fine-grained" : "hipMalloc",
         where, a, b, (int)h[a], (int)h[b], st_name[ST], ld_name[LD], lost ? "\"stale\", \"ns_until_gave_up\": " : "", best);
  fflush(stdout);
}

template <int ST> static void loads(unsigned long long* f, unsigned long long* o, int a, int b, const char* w, bool fine) {
  one<ST, 0>(f, o, a, b, w, fine); one<ST, 1>(f, o, a, b, w, fine); one<ST, 2>(f, o, a, b, w, fine); one<ST, 3>(f, o, a, b, w, fine);
  one<ST, 4>(f, o, a, b, w, fine); one<ST, 5>(f, o, a, b, w, fine);
}

int main() {
  unsigned long long *d_flags, *d_fine, *d_out;
  hipMalloc(&d_flags, 4096); hipMemset(d_flags, 0, 4096);
  hipExtMallocWithFlags((void**)&d_fine, 4096, hipDeviceMallocFinegrained); hipMemset(d_fine, 0, 4096);
  hipMalloc(&d_out, 1024); hipMemset(d_out, 0, 1024);
  for (int mem = 0; mem < 2; mem++) {
    unsigned long long* f = mem ? d_fine : d_flags;
    for (int pair = 0; pair < 3; pair++) {
      // workgroups go to the XCDs round robin (checked in the output through the XCC ids): 0 and 8 share one, 0 and 1 do not, 0 and 4 sit
      // on XCDs of different IO dies
      const int a = 0, b = pair == 0 ? 8 : pair == 1 ? 1 : 4;
      const char* w = pair == 0 ? "same XCD" : pair == 1 ? "XCD 0 -> 1" : "XCD 0 -> 4";
      loads<0>(f, d_out, a, b, w, mem); loads<1>(f, d_out, a, b, w, mem); loads<2>(f, d_out, a, b, w, mem); loads<3>(f, d_out, a, b, w, mem);
    }
  }
  return 0;
}
