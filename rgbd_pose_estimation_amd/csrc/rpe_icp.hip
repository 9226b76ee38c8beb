// F3 + K1/K2 fused: ICP rounds (projective association + normal equations), one launch per round and the RESIDENT form.
#include "rpe_residuals.hpp"
#include "rpe_assoc.h"

namespace rpe {

// ================================================================================================
// F3 + K1/K2 fused: one ICP round in ONE pass.  Each frame vertex is paired with the model by projective association
// (rpe_assoc.h, the same function the stand-alone association kernel runs) and its residual is accumulated at once: the
// pairs never exist in HBM (48 B/pixel read -- frame vertex + normal streamed, model vertex + normal gathered -- against
// 120 B/pixel for the association pass plus 36 B/pixel for the normal-equation pass).  The pairing function
// and the per-pixel arithmetic are those of the two-kernel path; only the summation order differs (1e-13 relative).
// ================================================================================================
template <int KIND, int BLK>
__global__ __launch_bounds__(BLK) void icp_fused_kernel(const float* __restrict__ vmap, const float* __restrict__ nmap, int64_t n,
                                                        const float* __restrict__ mv, const float* __restrict__ mn, AssocParams P,
                                                        PoseK<double> pose, Finish fin) {
  constexpr int NACC = KIND == KIND_P2P ? 17 : 29;
  if (fin.gn != nullptr) {
    if (fin.gn->done) return;
#pragma unroll
    for (int k = 0; k < 9; k++) pose.R[k] = fin.gn_pose[k];
#pragma unroll
    for (int k = 0; k < 3; k++) pose.t[k] = fin.gn_pose[9 + k];
  }
  PoseF T;
#pragma unroll
  for (int k = 0; k < 9; k++) T.R[k] = (float)pose.R[k];
#pragma unroll
  for (int k = 0; k < 3; k++) T.t[k] = (float)pose.t[k];
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) acc[k] = 0.0;
  const short m_none[4] = {1, 1, 1, 1};
  const float w_none[4] = {1.f, 1.f, 1.f, 1.f};
  const float nan = __int_as_float(0x7fc00000);
  const int64_t full = n / 4;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const float4* __restrict__ v4 = reinterpret_cast<const float4*>(vmap);
  const float4* __restrict__ n4 = reinterpret_cast<const float4*>(nmap);
  for (int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x; g < full; g += stride) {
    float V[12], N[12], vw[12], vb[12], vc[12];
    unpack3(v4[3 * g], v4[3 * g + 1], v4[3 * g + 2], V);
    unpack3(n4[3 * g], n4[3 * g + 1], n4[3 * g + 2], N);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float gx, gy, gz;
      const bool ok = associate_pixel(T, P, mv, mn, V[3 * i], V[3 * i + 1], V[3 * i + 2], N[3 * i], N[3 * i + 1], N[3 * i + 2],
          vw[3 * i],
                                      vw[3 * i + 1], vw[3 * i + 2], gx, gy, gz);
#pragma unroll
      for (int k = 0; k < 3; k++) { vb[3 * i + k] = ok ? V[3 * i + k] : nan; vc[3 * i + k] = ok ? N[3 * i + k] : nan; }
    }
    normal_eq_group<float, KIND, false, false, NACC>(pose, vw, vb, vc, m_none, w_none, 4, acc);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && full * 4 < n) {  // leftover pixels
    float vw[12], vb[12], vc[12];
    const int left = (int)(n - full * 4);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float x = nan, y = nan, z = nan, nx = nan, ny = nan, nz = nan, gx, gy, gz;
      if (i < left) {
        const int64_t q = 3 * (full * 4 + i);
        x = vmap[q]; y = vmap[q + 1]; z = vmap[q + 2]; nx = nmap[q]; ny = nmap[q + 1]; nz = nmap[q + 2];
      }
      const bool ok = associate_pixel(T, P, mv, mn, x, y, z, nx, ny, nz, vw[3 * i], vw[3 * i + 1], vw[3 * i + 2], gx, gy, gz);
      vb[3 * i] = ok ? x : nan; vb[3 * i + 1] = ok ? y : nan; vb[3 * i + 2] = ok ? z : nan;
      vc[3 * i] = ok ? nx : nan; vc[3 * i + 1] = ok ? ny : nan; vc[3 * i + 2] = ok ? nz : nan;
    }
    normal_eq_group<float, KIND, false, false, NACC>(pose, vw, vb, vc, m_none, w_none, left, acc);
  }
  reduce_and_finish<NACC, kNeLd, KIND == KIND_P2P ? 1 : 0, BLK>(acc, fin);
}

// RESIDENT form of the fused ICP round (host-driven ICP: rpe_icp with fused = 1, device_resident = 0): ONE launch for the whole loop.
// The frame's vertices and normals (one group of 4 pixels per thread at 640 x 480) are read once and stay in registers; every
// iteration the workgroups wait for the host's pose (resident_wait_pose), pair their pixels with the model under that pose
// (associate_pixel: the model vertex / normal gathers are the only memory traffic of an iteration) and accumulate the normal equations,
// and the sums reach the host through the collecting stage (resident_cross_stage).  Pairing function and per-pixel arithmetic are
// the fused kernel's; only the order of the cross-workgroup sums differs.
template <int KIND, int BLK, bool IN_REGS, bool AUTO>
__global__ __launch_bounds__(BLK) void icp_resident_kernel(const float* __restrict__ vmap, const float* __restrict__ nmap, int64_t n,
                                                           const float* __restrict__ mv, const float* __restrict__ mn, AssocParams P,
                                                           const unsigned long long* __restrict__ ctl, unsigned long long first_tag,
                                                           int max_iters, Finish fin) {
  constexpr int NACC = KIND == KIND_P2P ? 17 : 29;
  __shared__ double s_pose[12];
  __shared__ int s_go;
  const short m_none[4] = {1, 1, 1, 1};
  const float w_none[4] = {1.f, 1.f, 1.f, 1.f};
  const float nan = __int_as_float(0x7fc00000);
  const int64_t full = n / 4, groups = (n + 3) / 4;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const int64_t g0 = (int64_t)blockIdx.x * BLK + threadIdx.x;
  // one group of frame pixels: 16-byte loads for whole groups, bounds-checked scalars for the ragged last one
  auto load_pixels = [&](int64_t g, float (&V)[12], float (&N)[12]) {
    if (g < full) {
      const float4* v4 = reinterpret_cast<const float4*>(vmap);
      const float4* n4 = reinterpret_cast<const float4*>(nmap);
      unpack3(v4[3 * g], v4[3 * g + 1], v4[3 * g + 2], V);
      unpack3(n4[3 * g], n4[3 * g + 1], n4[3 * g + 2], N);
    } else {
#pragma unroll
      for (int i = 0; i < 12; i++) {
        const int64_t q = 12 * g + i;
        const bool in = q < 3 * n;
        V[i] = in ? vmap[q] : nan;
        N[i] = in ? nmap[q] : nan;
      }
    }
  };
  float rV[12], rN[12];
  const bool mine = IN_REGS && g0 < groups;
  if (mine) load_pixels(g0, rV, rN);
  // AUTO (rpe_icp with device_resident): no host in the loop -- first pose from HBM, every later one from the workgroup's own solve
  double tol = 0.0;
  if (AUTO) {
    if (threadIdx.x < 12) s_pose[threadIdx.x] = fin.gn_pose[threadIdx.x];
    tol = fin.gn->tol;
    __syncthreads();
  }
  for (int it = 1; it <= max_iters; it++) {
    if (!AUTO && resident_wait_pose<BLK>(ctl, first_tag + (unsigned long long)it, s_pose, &s_go, fin.pose_wait_ticks) != 1) return;
    PoseK<double> pose;
#pragma unroll
    for (int k = 0; k < 9; k++) pose.R[k] = s_pose[k];
#pragma unroll
    for (int k = 0; k < 3; k++) pose.t[k] = s_pose[9 + k];
    PoseF T;
#pragma unroll
    for (int k = 0; k < 9; k++) T.R[k] = (float)pose.R[k];
#pragma unroll
    for (int k = 0; k < 3; k++) T.t[k] = (float)pose.t[k];
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = 0.0;
    auto pair_and_add = [&](const float (&V)[12], const float (&N)[12], int present) {
      float vw[12], vb[12], vc[12];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float gx, gy, gz;
        const bool ok = associate_pixel(T, P, mv, mn, V[3 * i], V[3 * i + 1], V[3 * i + 2], N[3 * i], N[3 * i + 1], N[3 * i + 2],
            vw[3 * i],
                                        vw[3 * i + 1], vw[3 * i + 2], gx, gy, gz);
#pragma unroll
        for (int k = 0; k < 3; k++) { vb[3 * i + k] = ok ? V[3 * i + k] : nan; vc[3 * i + k] = ok ? N[3 * i + k] : nan; }
      }
      normal_eq_group<float, KIND, false, false, NACC>(pose, vw, vb, vc, m_none, w_none, present, acc);
    };
    if (IN_REGS) {
      if (mine) pair_and_add(rV, rN, g0 < full ? 4 : (int)(n - full * 4));
    } else {
      for (int64_t g = g0; g < groups; g += stride) {
        float V[12], N[12];
        load_pixels(g, V, N);
        pair_and_add(V, N, g < full ? 4 : (int)(n - full * 4));
      }
    }
    if (AUTO) {
      if (resident_auto_stage<NACC, BLK>(acc, fin, first_tag + (unsigned long long)it, it, max_iters, tol, s_pose) != 0) return;
      continue;
    }
    if (!resident_cross_stage<NACC, BLK>(acc, fin, first_tag + (unsigned long long)it, fin.seq + (unsigned long long)it, false)) return;
  }
}

hipError_t launch_icp_fused(const float* vmap, const float* nmap, int64_t n, const float* mv, const float* mn, const Camera& mcam,
                            const PoseF& M, float dist_sq, float cos_thr, int use_normals, int kind, const double* pose12, const ReduceTarget& rt,
                            hipStream_t s) {
  AssocParams P;
  P.mcam = mcam; P.M = M; P.dist_sq = dist_sq; P.cos_thr = cos_thr; P.use_normals = use_normals;
  const PoseK<double> pose = make_pose<double>(pose12);
  const Finish fin = make_finish(rt);
  // geometry: the body (dependent gathers + fp64 accumulation) is heavier than a streaming pass, so every thread gets ONE pixel
  // group and the grid covers the image (150 workgroups of 512 at 640 x 480: 17.4 us per round against 20.6 us with the 128
  // workgroups a streaming reduction of this size uses; measured, scripts/icp_sweep.sh); RPE_ICP_BLOCK / RPE_ICP_GRID override
  static const int env_blk = getenv("RPE_ICP_BLOCK") ? atoi(getenv("RPE_ICP_BLOCK")) : 0;
  static const int env_grid = getenv("RPE_ICP_GRID") ? atoi(getenv("RPE_ICP_GRID")) : 0;
  const int blk = env_blk == 256 || env_blk == 512 ? env_blk : pick_block(rt);
  const int64_t groups = (n + 3) / 4;
  int G = env_grid > 0 ? env_grid : rt.max_blocks;
  if ((int64_t)G > (groups + blk - 1) / blk) G = (int)((groups + blk - 1) / blk);
  if (G < 1) G = 1;
  if (blk == 512) {
    if (kind == KIND_P2P) hipLaunchKernelGGL((icp_fused_kernel<KIND_P2P, 512>), dim3(G), dim3(512), 0, s, vmap, nmap, n, mv, mn, P,
        pose, fin);
    else hipLaunchKernelGGL((icp_fused_kernel<KIND_P2PLANE, 512>), dim3(G), dim3(512), 0, s, vmap, nmap, n, mv, mn, P, pose, fin);
  } else {
    if (kind == KIND_P2P) hipLaunchKernelGGL((icp_fused_kernel<KIND_P2P, 256>), dim3(G), dim3(256), 0, s, vmap, nmap, n, mv, mn, P,
        pose, fin);
    else hipLaunchKernelGGL((icp_fused_kernel<KIND_P2PLANE, 256>), dim3(G), dim3(256), 0, s, vmap, nmap, n, mv, mn, P, pose, fin);
  }
  return hipGetLastError();
}

// resident ICP loop: grid / record geometry exactly as the resident normal-equation kernel's (pixels in groups of 4)
void icp_resident_geometry(int64_t n, int kind, int max_blocks, int* grid, int* nacc, int* max_rows, int* rows_auto) {
  DeviceArrays A{};
  A.n = n; A.dtype = 0;
  resident_geometry(A, kind, max_blocks, grid, nacc, max_rows, rows_auto);
}
hipError_t launch_icp_resident(const float* vmap, const float* nmap, int64_t n, const float* mv, const float* mn, const Camera& mcam,
                               const PoseF& M, float dist_sq, float cos_thr, int use_normals, int kind, const unsigned long long* ctl,
                               unsigned long long first_tag, int max_iters, const ReduceTarget& rt, hipStream_t s) {
  if (kind != KIND_P2P && kind != KIND_P2PLANE) return hipErrorInvalidValue;
  AssocParams P;
  P.mcam = mcam; P.M = M; P.dist_sq = dist_sq; P.cos_thr = cos_thr; P.use_normals = use_normals;
  constexpr int BLK = 512;
  const int cap = std::max(1, resident_cap_device());
  const int G = reduce_grid(n, 4, rt.max_blocks < cap ? rt.max_blocks : cap, BLK);
  const int64_t groups = (n + 3) / 4;
  const bool in_regs = (int64_t)G * BLK >= groups;
  Finish fin = make_finish(rt);
  const int max_rows = 4 * (BLK / (kind == KIND_P2P ? 17 : 29));
  if (fin.rows > max_rows) fin.rows = max_rows;
  if (fin.rows < 1) fin.rows = 1;
#define RPE_ICP_RES2(K, R, AU) hipLaunchKernelGGL((icp_resident_kernel<K, BLK, R, AU>), dim3(G), dim3(BLK), 0, s, vmap, nmap, n, mv, mn, P, ctl, first_tag, max_iters, fin)
#define RPE_ICP_RES(K, R) do { if (fin.gn != nullptr) RPE_ICP_RES2(K, R, true); else RPE_ICP_RES2(K, R, false); } while (0)
  if (kind == KIND_P2P) { if (in_regs) RPE_ICP_RES(KIND_P2P, true); else RPE_ICP_RES(KIND_P2P, false); }
  else { if (in_regs) RPE_ICP_RES(KIND_P2PLANE, true); else RPE_ICP_RES(KIND_P2PLANE, false); }
#undef RPE_ICP_RES
#undef RPE_ICP_RES2
  return hipGetLastError();
}

void preload_icp() {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, (const void*)icp_fused_kernel<KIND_P2P, 512>) != hipSuccess) (void)hipGetLastError();
}

}  // namespace rpe
