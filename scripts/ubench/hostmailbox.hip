// Host <-> resident-kernel ping-pong on MI355X: how long does one round trip take when a RESIDENT kernel waits for a word the host
// writes and answers into pinned host memory?  (Design input for the host-driven resident Gauss-Newton loop: the alternative is a
// kernel launch per iteration.)  Variants of the host -> GPU direction:
//   A  the word lives in pinned, coherent HOST memory; the GPU polls it across PCIe
//   B  the word lives in fine-grained DEVICE memory that the CPU writes through the PCIe BAR (if the runtime maps it for the host)
// GPU -> host is always a system-scope store into pinned host memory, which the host spins on.
// Every spin is bounded (the kernel gives up after ~2 s of the 100 MHz clock).
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/hostmailbox.hip -o scripts/ubench/hostmailbox && ./scripts/ubench/hostmailbox
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void pingpong(const unsigned long long* in, unsigned long long* out, int rounds, int nblocks_poll) {
  // every workgroup polls (as all workgroups of a resident reduction kernel would); workgroup 0 answers
  for (int r = 1; r <= rounds; r++) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)r) {
      if (wall_clock64() - t0 > 200000000ull) return;   // 2 s
      __builtin_amdgcn_s_sleep(1);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(out, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static sigjmp_buf g_jmp;
static void on_segv(int) { siglongjmp(g_jmp, 1); }

static double run(const char* name, unsigned long long* in_host_view, const unsigned long long* in_dev_view, unsigned long long* out, int rounds, int blocks) {
  *out = 0;
  *(volatile unsigned long long*)in_host_view = 0;
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipLaunchKernelGGL(pingpong, dim3(blocks), dim3(64), 0, s, in_dev_view, out, rounds, blocks);
  std::vector<double> rt;
  volatile unsigned long long* vout = out;
  volatile unsigned long long* vin = in_host_view;
  // give the kernel time to become resident
  std::this_thread::sleep_for(std::chrono::milliseconds(5));
  for (int r = 1; r <= rounds; r++) {
    const auto t0 = std::chrono::steady_clock::now();
    *vin = (unsigned long long)r;
    __sync_synchronize();
    long spins = 0;
    while (*vout < (unsigned long long)r) { if (++spins > 400000000L) { printf("%s: timeout at round %d\n", name, r); (void)hipStreamSynchronize(s); return -1; } }
    rt.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  (void)hipStreamSynchronize(s);
  (void)hipStreamDestroy(s);
  std::sort(rt.begin(), rt.end());
  printf("{\"variant\": \"%s\", \"blocks\": %d, \"rounds\": %d, \"rtt_us_median\": %.2f, \"rtt_us_p10\": %.2f, \"rtt_us_p90\": %.2f}\n", name, blocks, rounds,
         rt[rt.size() / 2], rt[rt.size() / 10], rt[rt.size() * 9 / 10]);
  return rt[rt.size() / 2];
}

int main() {
  CK(hipSetDevice(0));
  unsigned long long *out = nullptr, *in_host = nullptr;
  CK(hipHostMalloc((void**)&out, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostMalloc((void**)&in_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
  const int rounds = 2000;
  for (int blocks : {1, 150, 256}) run("A: GPU polls pinned host memory", in_host, in_host, out, rounds, blocks);
  // B: fine-grained device memory written by the CPU
  unsigned long long* in_dev = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&in_dev, 4096, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { printf("{\"variant\": \"B\", \"error\": \"hipExtMallocWithFlags: %s\"}\n", hipGetErrorString(e)); return 0; }
  CK(hipMemset(in_dev, 0, 4096));
  CK(hipDeviceSynchronize());
  signal(SIGSEGV, on_segv);
  signal(SIGBUS, on_segv);
  if (sigsetjmp(g_jmp, 1) == 0) {
    *(volatile unsigned long long*)in_dev = 0;   // faults if the runtime did not map the allocation for the CPU
    printf("{\"variant\": \"B\", \"cpu_write_to_device_memory\": \"ok\"}\n");
    for (int blocks : {1, 150, 256}) run("B: CPU writes fine-grained device memory (BAR)", in_dev, in_dev, out, rounds, blocks);
  } else {
    printf("{\"variant\": \"B\", \"cpu_write_to_device_memory\": \"fault: not host-accessible\"}\n");
  }
  return 0;
}
