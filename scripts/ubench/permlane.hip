// what v_permlane32_swap / v_permlane16_swap (gfx950) do to two registers: prints the lane contents (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  const unsigned a = threadIdx.x, b = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane32_swap r[0]", "permlane32_swap r[1]", "permlane16_swap r[0]", "permlane16_swap r[1]"};
  for (int t = 0; t < 4; t++) { printf("%s:", names[t]); for (int i = 0; i < 64; i++) printf(" %u", h[64 * t + i]); printf("\n"); }
  return 0;
}
