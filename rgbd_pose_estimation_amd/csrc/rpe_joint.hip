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
enum { TERM_P2P = 1, TERM_P2PLANE = 2, TERM_BEARING = 4, TERM_NORMAL = 8, TERM_REPROJ = 16 };   // 1 << residual kind
struct JointParams { double scale[5]; int robust[5]; double robust_k[5]; };  // indexed by residual kind 0..4

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
      // wave-uniform skip: configs[2] has bearings for 2 000 of 307 200 correspondences -- a wave none of whose lanes holds one pays a
      // ballot instead of the term (inside, everything stays predicated by `on`, so the sums do not depend on the branch)
      if (__builtin_amdgcn_ballot_w64(on) != 0) {
        const T bx = on ? bx0 : T(0), by = on ? by0 : T(0), bz = on ? bz0 : T(1);
        const double sx = on ? pxd : 0.0, sy = on ? pyd : 0.0, sz = on ? pzd : 1.0;  // keeps the normalisation finite when off
        const T w0 = on ? u23[i] : T(0);
        const T w = w0 * robust_weight<T>(prm.robust[2], (T)prm.robust_k[2],
            [&]() { return bearing_residual_norm<T>(sx, sy, sz, bx, by, bz); });
        bearing_point<T>(sx, sy, sz, bx, by, bz, (T)prm.scale[2] * w, w, s);
      }
    }
    if (TERMS & TERM_REPROJ) {   // the 2D-3D term as a pixel reprojection residual (alternative to the bearing form; same arrays)
      const T bx0 = vb[3 * i], by0 = vb[3 * i + 1], bz0 = vb[3 * i + 2];
      const bool on = present & (k23[i] == 1) & !all_nan(bx0, by0, bz0);
      if (__builtin_amdgcn_ballot_w64(on) != 0) {
        const T bx = on ? bx0 : T(0), by = on ? by0 : T(0), bz = on ? bz0 : T(1);
        const double sx = on ? pxd : 0.0, sy = on ? pyd : 0.0, sz = on ? pzd : 1.0;
        const T w0 = on ? u23[i] : T(0);
        const T w = w0 * robust_weight<T>(prm.robust[4], (T)prm.robust_k[4], [&]() { return reproj_residual_norm<T>(sx, sy, sz, bx, by, bz); });
        reproj_point<T>(sx, sy, sz, bx, by, bz, (T)prm.scale[4] * w, w, s);
      }
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

// the arrays, masks and weights of the joint kernels (null = absent) ...
template <class T> struct JointArrays {
  const T *xw, *xc, *bv, *nw, *nc;
  const short *m23, *m33, *mnn;
  const T *w23, *w33, *wnn;
};
// ... and one group of P correspondences of them in registers: only what the term set reads is loaded (absent masks / weights read as
// 1)
template <class T> struct JointRegs {
  enum { P = Pk<T>::P };
  T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
  short k23[P], k33[P], knn[P];
  T u23[P], u33[P], unn[P];
  template <int TERMS> __device__ __forceinline__ void load(const JointArrays<T>& A, int64_t g, int64_t n) {
    constexpr bool HAS33 = (TERMS & (TERM_P2P | TERM_P2PLANE)) != 0;
    constexpr bool NEED_NC = (TERMS & (TERM_P2PLANE | TERM_NORMAL)) != 0;
#pragma unroll
    for (int i = 0; i < P; i++) { k23[i] = k33[i] = knn[i] = 1; u23[i] = u33[i] = unn[i] = T(1); }
    load_group<T>(A.xw, g, n, vw);
    if (HAS33) {
      load_group<T>(A.xc, g, n, vc);
      if (A.m33) load_mask_group(A.m33, g, n, k33);
      if (A.w33) load_weight_group(A.w33, g, n, u33);
    }
    if (TERMS & (TERM_BEARING | TERM_REPROJ)) {
      load_group<T>(A.bv, g, n, vb);
      if (A.m23) load_mask_group(A.m23, g, n, k23);
      if (A.w23) load_weight_group(A.w23, g, n, u23);
    }
    if (TERMS & TERM_NORMAL) {
      load_group<T>(A.nw, g, n, vnw);
      if (A.mnn) load_mask_group(A.mnn, g, n, knn);
      if (A.wnn) load_weight_group(A.wnn, g, n, unn);
    }
    if (NEED_NC) load_group<T>(A.nc, g, n, vnc);
  }
};

template <class T, int TERMS, int BLK>
__global__ __launch_bounds__(BLK) void normal_eq_joint_kernel(JointArrays<T> A, int64_t n, PoseK<double> pose, JointParams prm,
    Finish fin) {
  constexpr int P = Pk<T>::P;
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
    JointRegs<T> q;
    q.template load<TERMS>(A, g, n);
    const int64_t left = n - g * P;
    joint_group<T, TERMS>(pose, prm, q.vw, q.vc, q.vb, q.vnw, q.vnc, q.k23, q.k33, q.knn, q.u23, q.u33, q.unn,
        left < P ? (int)left : P, acc);
  }
  reduce_and_finish<29, kNeLd, 0, BLK>(acc, fin);
}

// RESIDENT form (rpe_gn_refine_joint on one GPU): ONE launch for the whole refinement, as normal_eq_resident_kernel -- every iteration
// the workgroups wait for the host's pose in the control block (resident_wait_pose), evaluate their slice of the joint objective and
// hand
// the 29 sums to the collecting stage (resident_cross_stage); the host adds the run records, solves and updates.  Frame-sized problems
// (IN_REGS: one group per thread) read their arrays once per refinement.
template <class T, int TERMS, int BLK, bool IN_REGS, bool AUTO>
__global__ __launch_bounds__(BLK) void normal_eq_joint_resident_kernel(JointArrays<T> A, int64_t n, JointParams prm,
                                                                       const unsigned long long* __restrict__ ctl,
                                                                       unsigned long long first_tag, int max_iters, Finish fin) {
  constexpr int P = Pk<T>::P;
  __shared__ double s_pose[12];
  __shared__ int s_go;
  const int64_t groups = (n + P - 1) / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const int64_t g0 = (int64_t)blockIdx.x * BLK + threadIdx.x;
  JointRegs<T> mine;
  const bool have = IN_REGS && g0 < groups;
  if (have) mine.template load<TERMS>(A, g0, n);
  // AUTO (rpe_gn_refine_device with several residual kinds): no host in the loop -- the first pose from HBM, every later one from the
  // workgroup's own solve (resident_auto_stage), exactly as the single-kind resident kernel's autonomous form
  double tol = 0.0;
  if (AUTO) {
    if (threadIdx.x < 12) s_pose[threadIdx.x] = fin.gn_pose[threadIdx.x];
    tol = fin.gn->tol;
    __syncthreads();
  }
  for (int it = 1; it <= max_iters; it++) {
    // stop requested or no host
    if (!AUTO && resident_wait_pose<BLK>(ctl, first_tag + (unsigned long long)it, s_pose, &s_go, fin.pose_wait_ticks) != 1) return;
    PoseK<double> pose;
#pragma unroll
    for (int k = 0; k < 9; k++) pose.R[k] = s_pose[k];
#pragma unroll
    for (int k = 0; k < 3; k++) pose.t[k] = s_pose[9 + k];
    double acc[29];
#pragma unroll
    for (int k = 0; k < 29; k++) acc[k] = 0.0;
    if (IN_REGS) {
      const int64_t left = n - g0 * P;
      if (have)
        joint_group<T, TERMS>(pose, prm, mine.vw, mine.vc, mine.vb, mine.vnw, mine.vnc, mine.k23, mine.k33, mine.knn, mine.u23,
            mine.u33,
                              mine.unn, left < P ? (int)left : P, acc);
    } else {
      for (int64_t g = g0; g < groups; g += stride) {
        JointRegs<T> q;
        q.template load<TERMS>(A, g, n);
        const int64_t left = n - g * P;
        joint_group<T, TERMS>(pose, prm, q.vw, q.vc, q.vb, q.vnw, q.vnc, q.k23, q.k33, q.knn, q.u23, q.u33, q.unn,
            left < P ? (int)left : P, acc);
      }
    }
    if (AUTO) {
      if (resident_auto_stage<29, BLK>(acc, fin, first_tag + (unsigned long long)it, it, max_iters, tol, s_pose) != 0) return;
      continue;
    }
    if (!resident_cross_stage<29, BLK>(acc, fin, first_tag + (unsigned long long)it, fin.seq + (unsigned long long)it, false)) return;
  }
}

template <class T> static JointArrays<T> joint_arrays(const DeviceArrays& A, bool um, bool uw) {
  JointArrays<T> J;
  J.xw = (const T*)A.a[0]; J.xc = (const T*)A.a[1]; J.bv = (const T*)A.a[2]; J.nw = (const T*)A.a[3]; J.nc = (const T*)A.a[4];
  J.m23 = um ? (const short*)A.mask[0] : nullptr; J.m33 = um ? (const short*)A.mask[1] : nullptr; J.mnn = um ? (const short*)A.mask[2] : nullptr;
  J.w23 = uw ? (const T*)A.weight[0] : nullptr; J.w33 = uw ? (const T*)A.weight[1] : nullptr; J.wnn = uw ? (const T*)A.weight[2] : nullptr;
  return J;
}
template <class T, int TERMS, int BLK>
static void joint_launch_b(const DeviceArrays& A, int flags, const PoseK<double>& pose, const JointParams& prm, const ReduceTarget& rt,
    hipStream_t s);
template <class T, int TERMS>
static void joint_launch(const DeviceArrays& A, int flags, const PoseK<double>& pose, const JointParams& prm, const ReduceTarget& rt,
    hipStream_t s) {
  static const int env_blk = getenv("RPE_JOINT_BLOCK") ? atoi(getenv("RPE_JOINT_BLOCK")) : 0;
  // register-heavy kernel (up to three residual kinds, 29 fp64 accumulators): frames of the 640x480 class run 20 % faster with
  // 256-thread workgroups (one wave per SIMD, more workgroups in flight: 27.9 us vs 35.6 us at 307200), streaming sizes slightly
  // faster with 512 (10 M: 205 us vs 217 us)
  const int blk = env_blk == 256 || env_blk == 512 ? env_blk : (rt.block == 256
      || rt.block == 512 ? rt.block : (A.n <= 2000000 ? 256 : 512));
  if (blk == 256) joint_launch_b<T, TERMS, 256>(A, flags, pose, prm, rt, s);
  else joint_launch_b<T, TERMS, 512>(A, flags, pose, prm, rt, s);
}
template <class T, int TERMS, int BLK>
static void joint_launch_b(const DeviceArrays& A, int flags, const PoseK<double>& pose, const JointParams& prm, const ReduceTarget& rt,
    hipStream_t s) {
  const bool um = (flags & F_USE_MASK) != 0, uw = (flags & F_USE_WEIGHT) != 0;
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks, BLK);
  hipLaunchKernelGGL((normal_eq_joint_kernel<T, TERMS, BLK>), dim3(G), dim3(BLK), 0, s, joint_arrays<T>(A, um, uw), A.n, pose, prm,
      make_finish(rt));
}
template <class T>
static hipError_t joint_t(const DeviceArrays& A, int terms, int flags, const double* pose12, const double* scale4, const int* robust4,
                          const double* robust_k4, const ReduceTarget& rt, hipStream_t s) {
  const PoseK<double> pose = make_pose<double>(pose12);
  JointParams prm;
  for (int k = 0; k < 5; k++) { prm.scale[k] = scale4[k]; prm.robust[k] = robust4[k]; prm.robust_k[k] = robust_k4[k]; }   // (arrays of 5: kinds 0..4)
  switch (terms) {
#define RPE_JOINT_CASE(M) case M: joint_launch<T, M>(A, flags, pose, prm, rt, s); break;
    RPE_JOINT_CASE(1) RPE_JOINT_CASE(2) RPE_JOINT_CASE(4) RPE_JOINT_CASE(8) RPE_JOINT_CASE(5) RPE_JOINT_CASE(6) RPE_JOINT_CASE(9)
    RPE_JOINT_CASE(10) RPE_JOINT_CASE(12) RPE_JOINT_CASE(13) RPE_JOINT_CASE(14)
    RPE_JOINT_CASE(16) RPE_JOINT_CASE(17) RPE_JOINT_CASE(18) RPE_JOINT_CASE(24) RPE_JOINT_CASE(25) RPE_JOINT_CASE(26)   // ... with the reprojection form of the 2D-3D term
#undef RPE_JOINT_CASE
    default: return hipErrorInvalidValue;  // empty set, or point-to-point together with point-to-plane
  }
  return hipGetLastError();
}
hipError_t launch_normal_eq_joint(const DeviceArrays& A, int terms, int flags, const double* pose12, const double* scale4,
    const int* robust4,
                                  const double* robust_k4, const ReduceTarget& rt, hipStream_t s) {
  return A.dtype ? joint_t<double>(A, terms, flags, pose12, scale4, robust4, robust_k4, rt, s)
                 : joint_t<float>(A, terms, flags, pose12, scale4, robust4, robust_k4, rt, s);
}

// resident form: grid / record geometry as the resident normal-equation kernel's 29-sum kinds (resident_geometry with a non-p2p kind)
template <class T, int TERMS>
static void joint_resident_launch(const DeviceArrays& A, int flags, const JointParams& prm, const unsigned long long* ctl,
    unsigned long long first_tag,
                                  int max_iters, const ReduceTarget& rt, hipStream_t s) {
  constexpr int BLK = 512;
  const bool um = (flags & F_USE_MASK) != 0, uw = (flags & F_USE_WEIGHT) != 0;
  const int cap = std::max(1, resident_cap_device());
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks < cap ? rt.max_blocks : cap, BLK);
  const int64_t groups = (A.n + Pk<T>::P - 1) / Pk<T>::P;
  // one group per thread kept in registers across the iterations -- except the three-array fp32 term sets, whose resident groups would
  // spill (measured with scripts/kernel_resources.py: 180-316 bytes per lane); those re-read their cache-resident slice every iteration
  constexpr bool regs_fit = !(sizeof(T) == 4 && (TERMS == 12 || TERMS == 13 || TERMS == 14 || TERMS == 24 || TERMS == 25 || TERMS == 26));
  const bool in_regs = regs_fit && (int64_t)G * BLK >= groups;
  Finish fin = make_finish(rt);
  constexpr int kMaxRows = 4 * (BLK / 29);
  if (fin.rows > kMaxRows) fin.rows = kMaxRows;
  if (fin.rows < 1) fin.rows = 1;
  const JointArrays<T> J = joint_arrays<T>(A, um, uw);
#define RPE_JOINT_RES(R, AU) \
  hipLaunchKernelGGL((normal_eq_joint_resident_kernel<T, TERMS, BLK, R, AU>), dim3(G), dim3(BLK), 0, s, J, A.n, prm, ctl, first_tag, max_iters, fin)
  if (fin.gn != nullptr) { if (in_regs) RPE_JOINT_RES(true, true); else RPE_JOINT_RES(false, true); }
  else { if (in_regs) RPE_JOINT_RES(true, false); else RPE_JOINT_RES(false, false); }
#undef RPE_JOINT_RES
}
template <class T>
static hipError_t joint_resident_t(const DeviceArrays& A, int terms, int flags, const double* scale4, const int* robust4,
    const double* robust_k4,
                                   const unsigned long long* ctl, unsigned long long first_tag, int max_iters, const ReduceTarget& rt, hipStream_t s) {
  JointParams prm;
  for (int k = 0; k < 5; k++) { prm.scale[k] = scale4[k]; prm.robust[k] = robust4[k]; prm.robust_k[k] = robust_k4[k]; }   // (arrays of 5: kinds 0..4)
  switch (terms) {
#define RPE_JOINT_CASE(M) case M: joint_resident_launch<T, M>(A, flags, prm, ctl, first_tag, max_iters, rt, s); break;
    RPE_JOINT_CASE(1) RPE_JOINT_CASE(2) RPE_JOINT_CASE(4) RPE_JOINT_CASE(8) RPE_JOINT_CASE(5) RPE_JOINT_CASE(6) RPE_JOINT_CASE(9)
    RPE_JOINT_CASE(10) RPE_JOINT_CASE(12) RPE_JOINT_CASE(13) RPE_JOINT_CASE(14)
    RPE_JOINT_CASE(16) RPE_JOINT_CASE(17) RPE_JOINT_CASE(18) RPE_JOINT_CASE(24) RPE_JOINT_CASE(25) RPE_JOINT_CASE(26)   // ... with the reprojection form of the 2D-3D term
#undef RPE_JOINT_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_normal_eq_joint_resident(const DeviceArrays& A, int terms, int flags, const double* scale4, const int* robust4,
                                           const double* robust_k4, const unsigned long long* ctl, unsigned long long first_tag, int max_iters,
                                           const ReduceTarget& rt, hipStream_t s) {
  return A.dtype ? joint_resident_t<double>(A, terms, flags, scale4, robust4, robust_k4, ctl, first_tag, max_iters, rt, s)
                 : joint_resident_t<float>(A, terms, flags, scale4, robust4, robust_k4, ctl, first_tag, max_iters, rt, s);
}

void preload_joint() {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, (const void*)normal_eq_joint_kernel<float, TERM_P2P, 256>) != hipSuccess) (void)hipGetLastError();
}

}  // namespace rpe
