// K1+K2+K3 fused: joint Gauss-Newton normal equations of up to four residual kinds in one pass.
#include "rpe_residuals.hpp"

namespace rpe {

// ================================================================================================
// K1+K2+K3 fused: joint Gauss-Newton normal equations of up to four residual kinds in ONE pass over the arrays
// (3D-3D point-to-point or point-to-plane, 2D-3D bearing, normal-normal), each with its modality's inlier mask,
// per-correspondence weight, a scale and an optional robust (IRLS) weight.  This is the single-kernel form of the
// objective nl_shinji_kneip_ls alternates over (M33 + sigma (M23 + MNN), AbsoluteOrientationNormal.hpp:484-510).
// Record: H upper triangle (21) | g (6) | sum scale w r^2 | sum w.   Up to 60 B/corr + masks/weights.
// ================================================================================================
enum { TERM_P2P = 1, TERM_P2PLANE = 2, TERM_BEARING = 4, TERM_NORMAL = 8 };
struct JointParams { double scale[4]; int robust[4]; double robust_k[4]; };  // indexed by residual kind 0..3

// `robust` is a kernel argument (wave-uniform): the branch is a scalar one, and the common case -- no robust weight -- pays
// neither the square root its argument needs nor the two divisions
template <class C, class F> __device__ __forceinline__ C robust_weight(int robust, C k, F norm_of_residual) {
  if (robust == 0) return C(1);
  const C s = norm_of_residual();
  const C huber = s <= k ? C(1) : k / s;
  const C q = s / k;
  const C cauchy = C(1) / (C(1) + q * q);
  return robust == 1 ? huber : cauchy;
}
// point-to-point block written straight into the packed record (J = [I | -[p]x]: 35 flops instead of 3 generic rows)
template <class C> __device__ __forceinline__ void p2p_packed(C px, C py, C pz, C rx, C ry, C rz, C w, C w_unscaled, C (&s)[29]) {
  s[0] += w; s[6] += w; s[11] += w;
  s[4] = fma(w, pz, s[4]); s[5] = fma(-w, py, s[5]); s[8] = fma(-w, pz, s[8]); s[10] = fma(w, px, s[10]);
  s[12] = fma(w, py, s[12]); s[13] = fma(-w, px, s[13]);
  const C wx = w * px, wy = w * py, wz = w * pz;
  s[15] = fma(wy, py, fma(wz, pz, s[15])); s[16] = fma(-wx, py, s[16]); s[17] = fma(-wx, pz, s[17]);
  s[18] = fma(wx, px, fma(wz, pz, s[18])); s[19] = fma(-wy, pz, s[19]); s[20] = fma(wx, px, fma(wy, py, s[20]));
  const C wrx = w * rx, wry = w * ry, wrz = w * rz;
  s[21] += wrx; s[22] += wry; s[23] += wrz;
  s[24] += py * wrz - pz * wry; s[25] += pz * wrx - px * wrz; s[26] += px * wry - py * wrx;
  s[27] = fma(wrx, rx, fma(wry, ry, fma(wrz, rz, s[27])));
  s[28] += w_unscaled;
}

template <class T, int TERMS>
__device__ __forceinline__ void joint_group(const PoseK<double>& pose, const JointParams& prm, const T (&vw)[3 * Pk<T>::P],
                                            const T (&vc)[3 * Pk<T>::P], const T (&vb)[3 * Pk<T>::P], const T (&vnw)[3 * Pk<T>::P],
                                            const T (&vnc)[3 * Pk<T>::P], const short (&k23)[Pk<T>::P], const short (&k33)[Pk<T>::P],
                                            const short (&knn)[Pk<T>::P], const T (&u23)[Pk<T>::P], const T (&u33)[Pk<T>::P],
                                            const T (&unn)[Pk<T>::P], int npresent, double (&acc)[29]) {
  constexpr int P = Pk<T>::P;
  constexpr bool HAS33 = (TERMS & (TERM_P2P | TERM_P2PLANE)) != 0;
  T s[29];
#pragma unroll
  for (int k = 0; k < 29; k++) s[k] = T(0);
#pragma unroll
  for (int i = 0; i < P; i++) {
    const bool present = i < npresent;
    T x = vw[3 * i], y = vw[3 * i + 1], z = vw[3 * i + 2];
    double pxd, pyd, pzd;
    transform<T>(pose, x, y, z, pxd, pyd, pzd);
    const T px = (T)pxd, py = (T)pyd, pz = (T)pzd;
    if (HAS33) {
      const T cx = vc[3 * i], cy = vc[3 * i + 1], cz = vc[3 * i + 2];
      const bool on = present & (k33[i] == 1) & !all_nan(cx, cy, cz);
      const T rx = on ? (T)(pxd - (double)cx) : T(0), ry = on ? (T)(pyd - (double)cy) : T(0), rz = on ? (T)(pzd - (double)cz) : T(0);
      const T qx = on ? px : T(0), qy = on ? py : T(0), qz = on ? pz : T(0);
      if (TERMS & TERM_P2P) {
        const T w0 = on ? u33[i] : T(0);
        const T w = w0 * robust_weight<T>(prm.robust[0], (T)prm.robust_k[0], [&]() { return sqrt(rx * rx + ry * ry + rz * rz); });
        p2p_packed<T>(qx, qy, qz, rx, ry, rz, (T)prm.scale[0] * w, w, s);
      }
      if (TERMS & TERM_P2PLANE) {
        const T nx = on ? vnc[3 * i] : T(0), ny = on ? vnc[3 * i + 1] : T(0), nz = on ? vnc[3 * i + 2] : T(0);
        const T r = nx * rx + ny * ry + nz * rz;
        const T w0 = on ? u33[i] : T(0);
        const T w = w0 * robust_weight<T>(prm.robust[1], (T)prm.robust_k[1], [&]() { return fabs(r); });
        const T J[6] = {nx, ny, nz, qy * nz - qz * ny, qz * nx - qx * nz, qx * ny - qy * nx};
        add_row(J, r, (T)prm.scale[1] * w, s);
        s[28] += w;
      }
    }
    if (TERMS & TERM_BEARING) {
      const T bx0 = vb[3 * i], by0 = vb[3 * i + 1], bz0 = vb[3 * i + 2];
      const bool on = present & (k23[i] == 1) & !all_nan(bx0, by0, bz0);
      const T bx = on ? bx0 : T(0), by = on ? by0 : T(0), bz = on ? bz0 : T(1);
      const double sx = on ? pxd : 0.0, sy = on ? pyd : 0.0, sz = on ? pzd : 1.0;  // keeps the normalisation finite when off
      const double invd = rsqrt64(sx * sx + sy * sy + sz * sz);
      const double hxd = sx * invd, hyd = sy * invd, hzd = sz * invd;
      const T r[3] = {(T)(hyd * (double)bz - hzd * (double)by), (T)(hzd * (double)bx - hxd * (double)bz), (T)(hxd * (double)by - hyd * (double)bx)};
      const T w0 = on ? u23[i] : T(0);
      const T w = w0 * robust_weight<T>(prm.robust[2], (T)prm.robust_k[2], [&]() { return sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]); });
      const T ws = (T)prm.scale[2] * w;
      const T qx = (T)sx, qy = (T)sy, qz = (T)sz, inv = (T)invd;
      const T h[3] = {(T)hxd, (T)hyd, (T)hzd};
      const T Bx[3][3] = {{T(0), -bz, by}, {bz, T(0), -bx}, {-by, bx, T(0)}};
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const T bh = Bx[u][0] * h[0] + Bx[u][1] * h[1] + Bx[u][2] * h[2];
        const T a0 = -(Bx[u][0] - bh * h[0]) * inv, a1 = -(Bx[u][1] - bh * h[1]) * inv, a2 = -(Bx[u][2] - bh * h[2]) * inv;
        const T J[6] = {a0, a1, a2, qy * a2 - qz * a1, qz * a0 - qx * a2, qx * a1 - qy * a0};
        add_row(J, r[u], ws, s);
      }
      s[28] += w;
    }
    if (TERMS & TERM_NORMAL) {
      const T mx = vnw[3 * i], my = vnw[3 * i + 1], mz = vnw[3 * i + 2];
      const T cx0 = vnc[3 * i], cy0 = vnc[3 * i + 1], cz0 = vnc[3 * i + 2];
      const bool on = present & (knn[i] == 1) & !all_nan(cx0, cy0, cz0);
      // q = R Nw and r = q - Nc in fp64, like p and its residuals: rounding R to fp32 would leave a 6e-8 step floor
      const double mxd = mx, myd = my, mzd = mz;
      const double qxd = fma(pose.R[0], mxd, fma(pose.R[1], myd, pose.R[2] * mzd));
      const double qyd = fma(pose.R[3], mxd, fma(pose.R[4], myd, pose.R[5] * mzd));
      const double qzd = fma(pose.R[6], mxd, fma(pose.R[7], myd, pose.R[8] * mzd));
      const T qx = on ? (T)qxd : T(0), qy = on ? (T)qyd : T(0), qz = on ? (T)qzd : T(0);
      const T rx = on ? (T)(qxd - (double)cx0) : T(0), ry = on ? (T)(qyd - (double)cy0) : T(0), rz = on ? (T)(qzd - (double)cz0) : T(0);
      const T w0 = on ? unn[i] : T(0);
      const T w = w0 * robust_weight<T>(prm.robust[3], (T)prm.robust_k[3], [&]() { return sqrt(rx * rx + ry * ry + rz * rz); });
      const T ws = (T)prm.scale[3] * w;
      // J = [0 | -[q]x] : only the rotation block:  H_ww += |q|^2 I - q q^T ,  g_w += q x r
      const T wx = ws * qx, wy = ws * qy, wz = ws * qz;
      s[15] = fma(wy, qy, fma(wz, qz, s[15])); s[16] = fma(-wx, qy, s[16]); s[17] = fma(-wx, qz, s[17]);
      s[18] = fma(wx, qx, fma(wz, qz, s[18])); s[19] = fma(-wy, qz, s[19]); s[20] = fma(wx, qx, fma(wy, qy, s[20]));
      s[24] += ws * (qy * rz - qz * ry); s[25] += ws * (qz * rx - qx * rz); s[26] += ws * (qx * ry - qy * rx);
      s[27] = fma(ws * rx, rx, fma(ws * ry, ry, fma(ws * rz, rz, s[27])));
      s[28] += w;
    }
  }
#pragma unroll
  for (int k = 0; k < 29; k++) acc[k] += (double)s[k];
}

template <class T, int TERMS, int BLK>
__global__ __launch_bounds__(BLK) void normal_eq_joint_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const T* __restrict__ bv,
                                                              const T* __restrict__ nw, const T* __restrict__ nc,
                                                              const short* __restrict__ m23, const short* __restrict__ m33,
                                                              const short* __restrict__ mnn, const T* __restrict__ w23,
                                                              const T* __restrict__ w33, const T* __restrict__ wnn, int64_t n,
                                                              PoseK<double> pose, JointParams prm, Finish fin) {
  constexpr int P = Pk<T>::P;
  constexpr bool HAS33 = (TERMS & (TERM_P2P | TERM_P2PLANE)) != 0;
  constexpr bool NEED_NC = (TERMS & (TERM_P2PLANE | TERM_NORMAL)) != 0;
  if (fin.gn != nullptr) {
    if (fin.gn->done) return;
#pragma unroll
    for (int k = 0; k < 9; k++) pose.R[k] = fin.gn_pose[k];
#pragma unroll
    for (int k = 0; k < 3; k++) pose.t[k] = fin.gn_pose[9 + k];
  }
  double acc[29];
#pragma unroll
  for (int k = 0; k < 29; k++) acc[k] = 0.0;
  const int64_t groups = (n + P - 1) / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  for (int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x; g < groups; g += stride) {
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    short k23[P], k33[P], knn[P];
    T u23[P], u33[P], unn[P];
#pragma unroll
    for (int i = 0; i < P; i++) { k23[i] = k33[i] = knn[i] = 1; u23[i] = u33[i] = unn[i] = T(1); }
    load_group<T>(xw, g, n, vw);
    if (HAS33) { load_group<T>(xc, g, n, vc); if (m33) load_mask_group(m33, g, n, k33); if (w33) load_weight_group(w33, g, n, u33); }
    if (TERMS & TERM_BEARING) { load_group<T>(bv, g, n, vb); if (m23) load_mask_group(m23, g, n, k23); if (w23) load_weight_group(w23, g, n, u23); }
    if (TERMS & TERM_NORMAL) { load_group<T>(nw, g, n, vnw); if (mnn) load_mask_group(mnn, g, n, knn); if (wnn) load_weight_group(wnn, g, n, unn); }
    if (NEED_NC) load_group<T>(nc, g, n, vnc);
    const int64_t left = n - g * P;
    joint_group<T, TERMS>(pose, prm, vw, vc, vb, vnw, vnc, k23, k33, knn, u23, u33, unn, left < P ? (int)left : P, acc);
  }
  reduce_and_finish<29, kNeLd, 0, BLK>(acc, fin);
}

template <class T, int TERMS, int BLK>
static void joint_launch_b(const DeviceArrays& A, int flags, const PoseK<double>& pose, const JointParams& prm, const ReduceTarget& rt, hipStream_t s);
template <class T, int TERMS>
static void joint_launch(const DeviceArrays& A, int flags, const PoseK<double>& pose, const JointParams& prm, const ReduceTarget& rt, hipStream_t s) {
  static const int env_blk = getenv("RPE_JOINT_BLOCK") ? atoi(getenv("RPE_JOINT_BLOCK")) : 0;
  // register-heavy kernel (up to three residual kinds, 29 fp64 accumulators): frames of the 640x480 class run 20 % faster with
  // 256-thread workgroups (one wave per SIMD, more workgroups in flight: 27.9 us vs 35.6 us at 307200), streaming sizes slightly
  // faster with 512 (10 M: 205 us vs 217 us)
  const int blk = env_blk == 256 || env_blk == 512 ? env_blk : (rt.block == 256 || rt.block == 512 ? rt.block : (A.n <= 2000000 ? 256 : 512));
  if (blk == 256) joint_launch_b<T, TERMS, 256>(A, flags, pose, prm, rt, s);
  else joint_launch_b<T, TERMS, 512>(A, flags, pose, prm, rt, s);
}
template <class T, int TERMS, int BLK>
static void joint_launch_b(const DeviceArrays& A, int flags, const PoseK<double>& pose, const JointParams& prm, const ReduceTarget& rt, hipStream_t s) {
  const bool um = (flags & F_USE_MASK) != 0, uw = (flags & F_USE_WEIGHT) != 0;
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks, BLK);
  hipLaunchKernelGGL((normal_eq_joint_kernel<T, TERMS, BLK>), dim3(G), dim3(BLK), 0, s, (const T*)A.a[0], (const T*)A.a[1], (const T*)A.a[2],
                     (const T*)A.a[3], (const T*)A.a[4], um ? (const short*)A.mask[0] : nullptr, um ? (const short*)A.mask[1] : nullptr,
                     um ? (const short*)A.mask[2] : nullptr, uw ? (const T*)A.weight[0] : nullptr, uw ? (const T*)A.weight[1] : nullptr,
                     uw ? (const T*)A.weight[2] : nullptr, A.n, pose, prm, make_finish(rt));
}
template <class T>
static hipError_t joint_t(const DeviceArrays& A, int terms, int flags, const double* pose12, const double* scale4, const int* robust4,
                          const double* robust_k4, const ReduceTarget& rt, hipStream_t s) {
  const PoseK<double> pose = make_pose<double>(pose12);
  JointParams prm;
  for (int k = 0; k < 4; k++) { prm.scale[k] = scale4[k]; prm.robust[k] = robust4[k]; prm.robust_k[k] = robust_k4[k]; }
  switch (terms) {
#define RPE_JOINT_CASE(M) case M: joint_launch<T, M>(A, flags, pose, prm, rt, s); break;
    RPE_JOINT_CASE(1) RPE_JOINT_CASE(2) RPE_JOINT_CASE(4) RPE_JOINT_CASE(8) RPE_JOINT_CASE(5) RPE_JOINT_CASE(6) RPE_JOINT_CASE(9)
    RPE_JOINT_CASE(10) RPE_JOINT_CASE(12) RPE_JOINT_CASE(13) RPE_JOINT_CASE(14)
#undef RPE_JOINT_CASE
    default: return hipErrorInvalidValue;  // empty set, or point-to-point together with point-to-plane
  }
  return hipGetLastError();
}
hipError_t launch_normal_eq_joint(const DeviceArrays& A, int terms, int flags, const double* pose12, const double* scale4, const int* robust4,
                                  const double* robust_k4, const ReduceTarget& rt, hipStream_t s) {
  return A.dtype ? joint_t<double>(A, terms, flags, pose12, scale4, robust4, robust_k4, rt, s)
                 : joint_t<float>(A, terms, flags, pose12, scale4, robust4, robust_k4, rt, s);
}

}  // namespace rpe
