// How does the host learn that a kernel's result (in device or host-mapped memory) is complete, and what does each way cost after the
// kernel?  Development aid for the sharded Gauss-Newton step (rpe_dist.hip): the record of a step passes through a collective on the
// stream, so the kernel that wrote it cannot raise the flag itself.  Variants, each timed from the launch call to the host seeing the
// signal (median of `iters`):
//   self     the kernel stores the flag itself (what the single-GPU kernels do; the floor)
//   kernel   a second one-workgroup kernel behind it stores data + flag (launch_publish_f64)
//   value    hipStreamWriteValue64 behind it (a command-processor write; data already in host-mapped memory)
//   event    hipEventRecord behind it, the host spins on hipEventQuery
//   sync     hipStreamSynchronize
// hipcc --offload-arch=gfx950 -O3 -o stream_signal stream_signal.hip && ./stream_signal
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void work(double* dst, unsigned long long* flag, unsigned long long seq, int spin) {
  // ~`spin` x 40 ns of dependent work, then 32 doubles
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x < 32) __hip_atomic_store(dst + threadIdx.x, (double)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void publish(const double* src, double* dst, unsigned long long* flag, unsigned long long seq) {
  if (threadIdx.x < 32) __hip_atomic_store(dst + threadIdx.x, src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double *d_buf, *h_buf;
  unsigned long long* h_flag;
  CK(hipMalloc(&d_buf, 64 * 8));
  CK(hipHostMalloc(&h_buf, 64 * 8, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostMalloc(&h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
  *h_flag = 0;
  int can = 0;
  (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  std::printf("{\"can_use_stream_wait_value\": %d}\n", can);
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const int iters = 3000;
  unsigned long long seq = 0;
  for (int spin : {0, 500}) {   // kernel body ~0 / ~5 us (100 MHz ticks)
    for (const char* var : {"self", "kernel", "value", "event", "sync"}) {
      const std::string v = var;
      if (v == "value" && !can) continue;
      std::vector<double> ts;
      for (int i = 0; i < iters + 200; i++) {
        ++seq;
        const double t0 = now_us();
        if (v == "self") hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, h_buf, h_flag, seq, spin);
        else if (v == "kernel") { hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, d_buf, (unsigned long long*)nullptr, seq, spin);
                                  hipLaunchKernelGGL(publish, dim3(1), dim3(64), 0, s, d_buf, h_buf, h_flag, seq); }
        else { hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, h_buf, (unsigned long long*)nullptr, seq, spin);
               if (v == "value") CK(hipStreamWriteValue64(s, h_flag, seq, 0));
               else if (v == "event") CK(hipEventRecord(ev, s)); }
        if (v == "event") { while (hipEventQuery(ev) == hipErrorNotReady) {} }
        else if (v == "sync") CK(hipStreamSynchronize(s));
        else { while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq) {} }
        const double t1 = now_us();
        if (h_buf[0] != (double)seq && v != "kernel") { /* data must be there with the signal */ std::printf("{\"variant\": \"%s\", \"error\": \"data behind the signal\"}\n", var); break; }
        if (i >= 200) ts.push_back(t1 - t0);
      }
      CK(hipStreamSynchronize(s));
      if (ts.empty()) continue;
      std::sort(ts.begin(), ts.end());
      std::printf("{\"variant\": \"%s\", \"kernel_spin_ticks\": %d, \"median_us\": %.2f, \"p10_us\": %.2f, \"p90_us\": %.2f}\n", var, spin, ts[ts.size() / 2], ts[ts.size() / 10], ts[ts.size() * 9 / 10]);
      std::fflush(stdout);
    }
  }
  return 0;
}
