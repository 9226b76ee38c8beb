// Read-bandwidth microbenchmark on MI355X: what can a streaming reduction reach, and does the reference's xyz-interleaved
// (12-byte record) layout cost anything against a perfectly coalesced stream?  (development aid; results quoted in DESIGN.md)
//   A  coalesced float4: lane l reads element base + l            (1 KiB contiguous per wave instruction)
//   B  record-strided float4 x3: lane l reads 3 consecutive float4 at 48*l (what normal_eq_kernel does)
//   C  as B but two independent groups in flight per thread
//   D  float4 copy (read + write) for the 6.3 TB/s figure of MI355X_MICROARCH.md
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int BLK> __global__ __launch_bounds__(BLK) void kA(const float4* __restrict__ a, const float4* __restrict__ b, size_t n4, float* out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < n4; i += (size_t)gridDim.x * BLK) {
    float4 u = a[i], v = b[i];
    s += u.x + u.y + u.z + u.w + v.x + v.y + v.z + v.w;
  }
  if (s == 12345.678f) out[0] = s;
}
template <int BLK> __global__ __launch_bounds__(BLK) void kB(const float4* __restrict__ a, const float4* __restrict__ b, size_t groups, float* out) {
  float s = 0;
  for (size_t g = (size_t)blockIdx.x * BLK + threadIdx.x; g < groups; g += (size_t)gridDim.x * BLK) {
    const float4* p = a + 3 * g; const float4* q = b + 3 * g;
    float4 u0 = p[0], u1 = p[1], u2 = p[2], v0 = q[0], v1 = q[1], v2 = q[2];
    s += u0.x + u0.w + u1.y + u2.z + v0.x + v0.w + v1.y + v2.z + u0.y + u0.z + u1.x + u1.z + u1.w + u2.x + u2.y + u2.w + v0.y + v0.z + v1.x + v1.z + v1.w + v2.x + v2.y + v2.w;
  }
  if (s == 12345.678f) out[0] = s;
}
template <int BLK> __global__ __launch_bounds__(BLK) void kC(const float4* __restrict__ a, const float4* __restrict__ b, size_t groups, float* out) {
  float s = 0;
  const size_t stride = (size_t)gridDim.x * BLK;
  for (size_t g = (size_t)blockIdx.x * BLK + threadIdx.x; g < groups; g += 2 * stride) {
    const size_t g2 = g + stride < groups ? g + stride : g;
    const float4* p = a + 3 * g; const float4* q = b + 3 * g; const float4* p2 = a + 3 * g2; const float4* q2 = b + 3 * g2;
    float4 u0 = p[0], u1 = p[1], u2 = p[2], v0 = q[0], v1 = q[1], v2 = q[2];
    float4 w0 = p2[0], w1 = p2[1], w2 = p2[2], x0 = q2[0], x1 = q2[1], x2 = q2[2];
    s += u0.x + u0.w + u1.y + u2.z + v0.x + v0.w + v1.y + v2.z + w0.x + w1.y + w2.z + x0.x + x1.y + x2.z + u0.y + u1.x + u2.w + v0.z + v1.w + v2.x + w0.w + w1.z + w2.y + x0.y + x1.x + x2.w;
  }
  if (s == 12345.678f) out[0] = s;
}
template <int BLK> __global__ __launch_bounds__(BLK) void kD(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < n4; i += (size_t)gridDim.x * BLK) b[i] = a[i];
}

int main() {
  const size_t n = 20000000;  // correspondences: 2 arrays x 240 MB
  const size_t n4 = n * 3 / 4, groups = n / 4;
  float4 *a, *b; float* out;
  CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0x3c, n4 * 16)); CK(hipMemset(b, 0x3d, n4 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0); for (int i = 0; i < 20; i++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-34s %8.1f us  %7.0f GB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
  };
  const double rb = 2.0 * n4 * 16;
  for (int G : {512, 1024, 2048, 4096}) {
    char nm[64];
    snprintf(nm, 64, "A coalesced   256thr x%d", G); run(nm, [&] { hipLaunchKernelGGL(kA<256>, dim3(G), dim3(256), 0, 0, a, b, n4, out); }, rb);
    snprintf(nm, 64, "B rec-strided 256thr x%d", G); run(nm, [&] { hipLaunchKernelGGL(kB<256>, dim3(G), dim3(256), 0, 0, a, b, groups, out); }, rb);
    snprintf(nm, 64, "C rec-strided x2 256thr x%d", G); run(nm, [&] { hipLaunchKernelGGL(kC<256>, dim3(G), dim3(256), 0, 0, a, b, groups, out); }, rb);
  }
  for (int G : {256, 512, 1024}) {
    char nm[64];
    snprintf(nm, 64, "A coalesced   512thr x%d", G); run(nm, [&] { hipLaunchKernelGGL(kA<512>, dim3(G), dim3(512), 0, 0, a, b, n4, out); }, rb);
    snprintf(nm, 64, "B rec-strided 512thr x%d", G); run(nm, [&] { hipLaunchKernelGGL(kB<512>, dim3(G), dim3(512), 0, 0, a, b, groups, out); }, rb);
    snprintf(nm, 64, "C rec-strided x2 512thr x%d", G); run(nm, [&] { hipLaunchKernelGGL(kC<512>, dim3(G), dim3(512), 0, 0, a, b, groups, out); }, rb);
  }
  run("D float4 copy 256thr x4096", [&] { hipLaunchKernelGGL(kD<256>, dim3(4096), dim3(256), 0, 0, a, b, n4); }, rb);
  return 0;
}
