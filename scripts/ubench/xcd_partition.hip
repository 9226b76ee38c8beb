// Would a partition by XCD recover the start skew of a one-launch streaming reduction?  (scripts/ubench/dispatch_ramp.hip: the XCDs'
// dispatchers start their workgroups 0 ... 1.2 us apart; the launch ends when the LAST XCD's run is collected.)  A stand-in for
// normal_eq_kernel at 1 M point-to-plane: three arrays of 12 MB read with 16-byte loads, a few fused multiply-adds per value, one
// partial per workgroup.  Variants: the grid-stride sweep over all workgroups (what ships), and per-XCD contiguous shares swept by the
// XCD's own workgroups, share_x = (1 + (mean(s) - s_x) / B) / 8 for a table s of start offsets measured in the same process and
// B = the assumed body time -- swept.  Time = launch to last workgroup's partial stored (HIP events), median of `iters` launches, each
// launched on an idle GPU as the product's calls are.
// hipcc --offload-arch=gfx950 -O3 -o xcd_partition xcd_partition.hip && ./xcd_partition
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Shares { long long begin[9]; };   // groups [begin[x], begin[x + 1]) belong to XCD x

__global__ __launch_bounds__(512) void sweep(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                             long long groups, Shares sh, int by_xcd, double* __restrict__ out, unsigned long long* __restrict__ t0s) {
  if (threadIdx.x == 0 && blockIdx.x < 8 && t0s) t0s[blockIdx.x] = wall_clock64();
  long long g, end, stride;
  if (by_xcd) {
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3, per = (gridDim.x + 7 - x) / 8;   // workgroups go to the XCDs round robin
    g = sh.begin[x] + (long long)k * 512 + threadIdx.x; end = sh.begin[x + 1]; stride = (long long)per * 512;
  } else {
    g = (long long)blockIdx.x * 512 + threadIdx.x; end = groups; stride = (long long)gridDim.x * 512;
  }
  float acc0 = 0, acc1 = 0, acc2 = 0;
  for (; g < end; g += stride) {
    const float4 u0 = a[3 * g], u1 = a[3 * g + 1], u2 = a[3 * g + 2];
    const float4 v0 = b[3 * g], v1 = b[3 * g + 1], v2 = b[3 * g + 2];
    const float4 w0 = c[3 * g], w1 = c[3 * g + 1], w2 = c[3 * g + 2];
    acc0 = fmaf(u0.x, v0.x, fmaf(u0.y, v0.y, fmaf(u0.z, v0.z, fmaf(u0.w, v0.w, acc0)))) + w0.x * w1.y;
    acc1 = fmaf(u1.x, v1.x, fmaf(u1.y, v1.y, fmaf(u1.z, v1.z, fmaf(u1.w, v1.w, acc1)))) + w1.x * w2.y;
    acc2 = fmaf(u2.x, v2.x, fmaf(u2.y, v2.y, fmaf(u2.z, v2.z, fmaf(u2.w, v2.w, acc2)))) + w2.x * w0.y;
  }
  double s = (double)acc0 + (double)acc1 + (double)acc2;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  __shared__ double red[8];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < 8; w++) t += red[w]; out[blockIdx.x] = t; }
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  const long long n = 1000000, groups = n / 4;   // a group = 4 correspondences = three 16-byte loads per array
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float4 *a, *b, *c;
  double* out;
  unsigned long long *t0d, t0h[8];
  CK(hipMalloc(&a, groups * 48)); CK(hipMalloc(&b, groups * 48)); CK(hipMalloc(&c, groups * 48));
  CK(hipMemset(a, 0, groups * 48)); CK(hipMemset(b, 0, groups * 48)); CK(hipMemset(c, 0, groups * 48));
  CK(hipMalloc(&out, 4096 * 8)); CK(hipMalloc(&t0d, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 300;
  for (int grid : {245, 256, 490}) {
    // the start offsets of the eight XCDs' first workgroups on this part, median of 100 launches
    std::vector<std::vector<double>> off(8);
    Shares flat;
    for (int x = 0; x <= 8; x++) flat.begin[x] = groups * x / 8;
    for (int it = 0; it < 100; it++) {
      hipLaunchKernelGGL(sweep, dim3(grid), dim3(512), 0, s, a, b, c, groups, flat, 0, out, t0d);
      CK(hipMemcpyAsync(t0h, t0d, 64, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      const unsigned long long m = *std::min_element(t0h, t0h + 8);
      for (int x = 0; x < 8; x++) off[x].push_back((t0h[x] - m) * 0.01);
    }
    double sx[8], mean = 0;
    for (int x = 0; x < 8; x++) { sx[x] = med(off[x]); mean += sx[x] / 8; }
    auto run = [&](const Shares& sh, int by_xcd) -> double {
      std::vector<double> ts;
      for (int it = 0; it < iters + 20; it++) {
        (void)(hipStreamSynchronize(s));
        (void)(hipEventRecord(e0, s));
        hipLaunchKernelGGL(sweep, dim3(grid), dim3(512), 0, s, a, b, c, groups, sh, by_xcd, out, (unsigned long long*)nullptr);
        (void)(hipEventRecord(e1, s));
        (void)(hipEventSynchronize(e1));
        float ms = 0;
        (void)(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 20) ts.push_back(ms * 1e3);
      }
      return med(ts);
    };
    std::printf("{\"grid\": %d, \"xcd_start_us\": [%.2f, %.2f, %.2f, %.2f, %.2f, %.2f, %.2f, %.2f], ", grid, sx[0], sx[1], sx[2], sx[3], sx[4], sx[5], sx[6], sx[7]);
    const double t_flat = run(flat, 0), t_even = run(flat, 1);
    std::printf("\"grid_stride_us\": %.2f, \"by_xcd_even_us\": %.2f, \"by_xcd_weighted_us\": {", t_flat, t_even);
    bool first = true;
    for (double B : {2.0, 3.0, 4.0, 6.0, 10.0}) {
      Shares sh;
      double f[8], tot = 0;
      for (int x = 0; x < 8; x++) { f[x] = std::max(0.05, 1.0 + (mean - sx[x]) / B); tot += f[x]; }
      double accf = 0;
      sh.begin[0] = 0;
      for (int x = 0; x < 8; x++) { accf += f[x] / tot; sh.begin[x + 1] = x == 7 ? groups : (long long)(groups * accf); }
      std::printf("%s\"B=%.0f\": %.2f", first ? "" : ", ", B, run(sh, 1));
      first = false;
    }
    std::printf("}}\n");
  }
  return 0;
}
