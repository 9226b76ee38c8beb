// When does each workgroup of a one-launch kernel START, by XCD and by index?  Development aid for the one-launch reduction kernels
// (rpe_normal_eq.hip): their timeline says the last of 150-256 workgroups starts ~2 us after the first, which is a fifth of the launch
// at 1 M correspondences.  Is that a per-workgroup (or per-wave) dispatch cost - then fewer, fatter workgroups start sooner - or a skew
// between the XCDs' dispatchers, which no geometry changes?  Every workgroup stamps the 100 MHz clock and its XCC id at entry, spins
// `spin` ticks, and stamps again; the host prints, per (grid, block, LDS) geometry and as medians over the launches: the start of the
// first / median / last workgroup of every XCD after the launch's first start, and the same by index decile.
// hipcc --offload-arch=gfx950 -O3 -o dispatch_ramp dispatch_ramp.hip && ./dispatch_ramp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Stamp { unsigned long long t0, t1; unsigned xcc, pad; };

__global__ void probe(Stamp* out, int spin) {
  extern __shared__ char lds[];
  unsigned long long t0 = wall_clock64();
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (lds && threadIdx.x == 0xffff) lds[0] = 1;
  while (wall_clock64() - t0 < (unsigned long long)spin) {}
  __syncthreads();
  if (threadIdx.x == 0) {
    out[blockIdx.x].t0 = t0;
    out[blockIdx.x].t1 = wall_clock64();
    out[blockIdx.x].xcc = xcc & 0xf;
  }
}

static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Stamp *d, *h;
  const int maxg = 4096;
  CK(hipMalloc(&d, maxg * sizeof(Stamp)));
  CK(hipHostMalloc(&h, maxg * sizeof(Stamp)));
  CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int launches = 300;
  struct Geo { int grid, block, lds; };
  const Geo geos[] = {{150, 512, 0}, {150, 256, 0}, {300, 256, 0}, {150, 1024, 0}, {75, 1024, 0}, {256, 512, 0}, {256, 256, 0},
                      {512, 256, 0}, {128, 1024, 0}, {256, 1024, 0}, {512, 512, 0}, {1024, 256, 0}, {256, 512, 82 * 1024}, {256, 64, 0},
                      {2048, 64, 0}};
  for (int spin : {300}) {
    for (const Geo& g : geos) {
      std::vector<std::vector<double>> xf(8), xm(8), xl(8), dec(10);
      std::vector<double> last, endl;
      for (int it = 0; it < launches + 20; it++) {
        hipLaunchKernelGGL(probe, dim3(g.grid), dim3(g.block), g.lds, s, d, spin);
        CK(hipMemcpyAsync(h, d, g.grid * sizeof(Stamp), hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        if (it < 20) continue;
        unsigned long long t00 = ~0ull, t1m = 0;
        for (int b = 0; b < g.grid; b++) { t00 = std::min(t00, h[b].t0); t1m = std::max(t1m, h[b].t1); }
        std::vector<double> per[8];
        double mx = 0;
        for (int b = 0; b < g.grid; b++) {
          double t = (h[b].t0 - t00) * 0.01;
          per[h[b].xcc & 7].push_back(t);
          mx = std::max(mx, t);
        }
        last.push_back(mx);
        endl.push_back((t1m - t00) * 0.01);
        for (int x = 0; x < 8; x++) {
          if (per[x].empty()) continue;
          std::sort(per[x].begin(), per[x].end());
          xf[x].push_back(per[x].front()); xm[x].push_back(per[x][per[x].size() / 2]); xl[x].push_back(per[x].back());
        }
        for (int k = 0; k < 10; k++) {
          int b0 = g.grid * k / 10, b1 = std::max(b0 + 1, g.grid * (k + 1) / 10);
          double m = 0;
          for (int b = b0; b < b1; b++) m = std::max(m, (h[b].t0 - t00) * 0.01);
          dec[k].push_back(m);
        }
      }
      std::printf("{\"grid\": %d, \"block\": %d, \"lds\": %d, \"spin_us\": %.1f, \"start_last_us\": %.2f, \"all_done_us\": %.2f, \"xcd_first_med_last_us\": [",
                  g.grid, g.block, g.lds, spin * 0.01, med(last), med(endl));
      for (int x = 0; x < 8; x++)
        if (!xf[x].empty()) std::printf("%s[%.2f, %.2f, %.2f]", x ? ", " : "", med(xf[x]), med(xm[x]), med(xl[x]));
      std::printf("], \"latest_start_by_index_decile_us\": [");
      for (int k = 0; k < 10; k++) std::printf("%s%.2f", k ? ", " : "", med(dec[k]));
      std::printf("]}\n");
    }
  }
  return 0;
}
