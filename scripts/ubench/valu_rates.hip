// Issue cost of the VALU instructions the normal-equation kernels are made of, on gfx950: cycles per wave64 instruction on one SIMD, for
// one and two waves per SIMD, eight independent accumulators each (no dependency stalls).  Development aid (never infer an instruction's
// price from a table for another part):  hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

typedef float f2 __attribute__((ext_vector_type(2)));
enum { FMA32, PKFMA32, PKMUL32, FMA64, ADD64, MUL64, CVT_64_32, CVT_32_64, CNDMASK, RCP32, RSQ32, RCP64, CNDMASK64, BFI, AND32, CMP32, MOV32, NKINDS };
static const char* names[NKINDS] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cndmask_b32",
                                    "v_rcp_f32", "v_rsq_f32", "v_rcp_f64", "v_cndmask_b32_e64 (sgpr pair)", "v_bfi_b32", "v_and_b32", "v_cmp_lt_f32_e64", "v_mov_b32"};

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(int iters, unsigned long long* out, float seed) {
  float a[8], x = seed, y = seed + 1.f;
  f2 p[8], px = {seed, seed}, py = {seed + 1.f, seed};
  double d[8], dx = seed, dy = seed + 1.0;
#pragma unroll
  for (int k = 0; k < 8; k++) { a[k] = seed + k; p[k] = f2{seed + k, seed}; d[k] = seed + k; }
  unsigned long long msk = (unsigned long long)iters * 0x9E3779B97F4A7C15ull;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (KIND == FMA32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
      if (KIND == PKFMA32) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(px), "v"(py));
      if (KIND == PKMUL32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(px));
      if (KIND == FMA64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[k]) : "v"(dx), "v"(dy));
      if (KIND == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(dx));
      if (KIND == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(dx));
      if (KIND == CVT_64_32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(a[k]));
      if (KIND == CVT_32_64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[k]) : "v"(d[k]));
      if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(x));
      if (KIND == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
      if (KIND == RSQ32) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
      if (KIND == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[k]));
      if (KIND == CNDMASK64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "s"(msk));
      if (KIND == BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(x), "v"(y));
      if (KIND == AND32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
      if (KIND == CMP32) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(msk) : "v"(a[k]), "v"(x));
      if (KIND == MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(x));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory");
  float s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) s += a[k] + p[k].x + p[k].y + (float)d[k];
  if (s == 12345.678f || msk == 77) out[0] = 1;   // keep the chains alive
  if (threadIdx.x % 64 == 0) out[1 + (blockIdx.x * 4 + threadIdx.x / 64)] = t1 - t0;
}

template <int KIND> static void run(int waves_per_simd, unsigned long long* d_out, std::vector<unsigned long long>& h) {
  const int iters = 2000, grid = 256 * waves_per_simd;
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(256), 0, 0, iters, d_out, 1.0f);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(256), 0, 0, iters, d_out, 1.0f);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * (1 + grid * 4), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < grid * 4; i++) sum += (double)h[1 + i];
  const double per_wave = sum / (grid * 4) / (iters * 8.0);
  printf("{\"instruction\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instruction_per_wave\": %.2f, \"simd_cycles_per_instruction\": %.2f}\n", names[KIND], waves_per_simd,
         per_wave, per_wave / waves_per_simd);
}

int main() {
  unsigned long long* d_out;
  hipMalloc(&d_out, sizeof(unsigned long long) * (1 + 512 * 4));
  std::vector<unsigned long long> h(1 + 512 * 4);
  for (int w = 1; w <= 2; w++) {
    run<FMA32>(w, d_out, h); run<PKFMA32>(w, d_out, h); run<PKMUL32>(w, d_out, h); run<FMA64>(w, d_out, h); run<ADD64>(w, d_out, h); run<MUL64>(w, d_out, h);
    run<CVT_64_32>(w, d_out, h); run<CVT_32_64>(w, d_out, h); run<CNDMASK>(w, d_out, h); run<RCP32>(w, d_out, h); run<RSQ32>(w, d_out, h); run<RCP64>(w, d_out, h);
    run<CNDMASK64>(w, d_out, h); run<BFI>(w, d_out, h); run<AND32>(w, d_out, h); run<CMP32>(w, d_out, h); run<MOV32>(w, d_out, h);
  }
  return 0;
}
