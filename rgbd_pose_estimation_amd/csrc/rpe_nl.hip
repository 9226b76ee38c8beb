// K1': closed-form moments; K5: one round of nl_shinji_kneip_ls + find_opt_cc; the small publish kernels.
#include "rpe_reduce.hpp"

namespace rpe {

// ================================================================================================
// K1' : closed-form moments (both passes of shinji() in one): w | w Xw | w Xc | w Xc Xw^T | w |Xc|^2 | count
// fp32 x fp32 products are exact in fp64, so only the fp64 summation rounds.
// ================================================================================================
// CLEAN: every value of the group is finite (group_dirty, rpe_reduce.hpp, said so for the whole wave): no NaN guard and no selects on
// the coordinates -- a masked-off correspondence is switched off by its weight alone (0 x finite = 0), the sums are the same bits
template <class T, bool MASK, bool WEIGHT, bool CLEAN = false>
__device__ __forceinline__ void moments_group(const T (&vw)[3 * Pk<T>::P], const T (&vc)[3 * Pk<T>::P], const short (&m)[Pk<T>::P],
                                              const T (&wv)[Pk<T>::P], int npresent, int skip_invalid, double (&acc)[18]) {
  constexpr int P = Pk<T>::P;
#pragma unroll
  for (int i = 0; i < P; i++) {
    double x = vw[3 * i], y = vw[3 * i + 1], z = vw[3 * i + 2];
    double cx = vc[3 * i], cy = vc[3 * i + 1], cz = vc[3 * i + 2];
    bool use = CLEAN ? true : (i < npresent && !(skip_invalid && all_nan(cx, cy, cz)));
    if (MASK) use = use && m[i] == 1;
    const double wi = use ? (WEIGHT ? (double)wv[i] : 1.0) : 0.0;
    if (!CLEAN) { x = use ? x : 0.0; y = use ? y : 0.0; z = use ? z : 0.0; cx = use ? cx : 0.0; cy = use ? cy : 0.0; cz = use ? cz : 0.0; }
    const double wcx = wi * cx, wcy = wi * cy, wcz = wi * cz;
    acc[0] += wi;
    acc[1] = fma(wi, x, acc[1]); acc[2] = fma(wi, y, acc[2]); acc[3] = fma(wi, z, acc[3]);
    acc[4] += wcx; acc[5] += wcy; acc[6] += wcz;
    acc[7] = fma(wcx, x, acc[7]); acc[8] = fma(wcx, y, acc[8]); acc[9] = fma(wcx, z, acc[9]);
    acc[10] = fma(wcy, x, acc[10]); acc[11] = fma(wcy, y, acc[11]); acc[12] = fma(wcy, z, acc[12]);
    acc[13] = fma(wcz, x, acc[13]); acc[14] = fma(wcz, y, acc[14]); acc[15] = fma(wcz, z, acc[15]);
    acc[16] = fma(wcx, cx, fma(wcy, cy, fma(wcz, cz, acc[16])));
    acc[17] += use ? 1.0 : 0.0;
  }
}

template <class T, int BLK, bool MASK, bool WEIGHT>
__global__ __launch_bounds__(BLK) void moments_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const short* __restrict__ mask,
                             const T* __restrict__ weight, int64_t n, int skip_invalid, Finish fin) {
  constexpr int P = Pk<T>::P;
  typedef typename Pk<T>::V V;
  double acc[18];
#pragma unroll
  for (int k = 0; k < 18; k++) acc[k] = 0.0;
  const int64_t full = n / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const V* __restrict__ xw4 = reinterpret_cast<const V*>(xw);
  const V* __restrict__ xc4 = reinterpret_cast<const V*>(xc);
  // same software pipeline as normal_eq_kernel: next group's loads in flight while this one is accumulated
  int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
  V a0, a1, a2, b0, b1, b2;
  short m[P];
  T wv[P];
  if (g < full) {
    a0 = xw4[3 * g]; a1 = xw4[3 * g + 1]; a2 = xw4[3 * g + 2];
    b0 = xc4[3 * g]; b1 = xc4[3 * g + 1]; b2 = xc4[3 * g + 2];
    if (MASK) load_mask_full(mask, g, m);
    if (WEIGHT) load_weight_full(weight, g, wv);
  }
  while (g < full) {
    const int64_t gn = g + stride;
    const int64_t gl = gn < full ? gn : g;
    const V na0 = xw4[3 * gl], na1 = xw4[3 * gl + 1], na2 = xw4[3 * gl + 2];
    const V nb0 = xc4[3 * gl], nb1 = xc4[3 * gl + 1], nb2 = xc4[3 * gl + 2];
    short nm[P];
    T nwv[P];
    if (MASK) load_mask_full(mask, gl, nm);
    if (WEIGHT) load_weight_full(weight, gl, nwv);
    T vw[3 * P], vc[3 * P];
    unpack3(a0, a1, a2, vw);
    unpack3(b0, b1, b2, vc);
    // wave-uniform: no NaN / infinity in what any lane loaded -> the form without guards and selects
    if (__builtin_amdgcn_ballot_w64(group_dirty<V>(a0, a1, a2, b0, b1, b2)) == 0) moments_group<T, MASK, WEIGHT, true>(vw, vc, m, wv, P,
        skip_invalid, acc);
    else moments_group<T, MASK, WEIGHT, false>(vw, vc, m, wv, P, skip_invalid, acc);
    a0 = na0; a1 = na1; a2 = na2; b0 = nb0; b1 = nb1; b2 = nb2;
#pragma unroll
    for (int i = 0; i < P; i++) { if (MASK) m[i] = nm[i]; if (WEIGHT) wv[i] = nwv[i]; }
    g = gn;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && full * P < n) {
    T vw[3 * P], vc[3 * P];
    load_group<T>(xw, full, n, vw);
    load_group<T>(xc, full, n, vc);
    if (MASK) load_scalars<T, short>(mask, full, n, m, (short)0);
    if (WEIGHT) load_scalars<T, T>(weight, full, n, wv, T(0));
    moments_group<T, MASK, WEIGHT>(vw, vc, m, wv, (int)(n - full * P), skip_invalid, acc);
  }
  reduce_and_finish<18, kNeLd, 0, BLK>(acc, fin);
}

// ================================================================================================
// R1 : lsq_pnp (P3P.hpp:472-502) -- sum over ALL correspondences of the sine of the angle between the predicted and the observed
// bearing, err_i = | normalize(R Xw_i + t) x bv_i |, every err_i evaluated in the array dtype by the reference's own operation
// sequence (SO3 * v = Eigen _transformVector, sophus/so3.hpp:238-240; normalize() = division by the square root of the squared
// norm; cross; norm), contraction off: the terms are the reference's values bit for bit.  The reference adds them one after the
// other in Tp; here they are added in fp64 (per-thread, then the fixed-order two-stage reduction), so the total differs from the
// reference's only by its own accumulated rounding.  record: sum err | count
// ================================================================================================
template <class T> struct SinePose { T qw, qx, qy, qz, t[3]; };
template <class T>
__device__ __forceinline__ T sine_error(const SinePose<T>& q, T x, T y, T z, T bx, T by, T bz) {
#pragma clang fp contract(off)
  T ux = q.qy * z - q.qz * y, uy = q.qz * x - q.qx * z, uz = q.qx * y - q.qy * x;
  ux = ux + ux; uy = uy + uy; uz = uz + uz;
  const T cx = q.qy * uz - q.qz * uy, cy = q.qz * ux - q.qx * uz, cz = q.qx * uy - q.qy * ux;
  T px = ((x + q.qw * ux) + cx) + q.t[0], py = ((y + q.qw * uy) + cy) + q.t[1], pz = ((z + q.qw * uz) + cz) + q.t[2];
  const T len = sqrt(px * px + py * py + pz * pz);
  px = px / len; py = py / len; pz = pz / len;
  const T ex = py * bz - pz * by, ey = pz * bx - px * bz, ez = px * by - py * bx;
  return sqrt(ex * ex + ey * ey + ez * ez);
}
template <class T, int BLK>
__global__ __launch_bounds__(BLK) void sine_error_kernel(const T* __restrict__ xw, const T* __restrict__ bv, int64_t n, SinePose<T> q, Finish fin) {
  constexpr int P = Pk<T>::P;
  typedef typename Pk<T>::V V;
  double acc[2] = {0.0, 0.0};
  const int64_t full = n / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const V* __restrict__ xw4 = reinterpret_cast<const V*>(xw);
  const V* __restrict__ bv4 = reinterpret_cast<const V*>(bv);
  int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
  V a0, a1, a2, b0, b1, b2;
  if (g < full) { a0 = xw4[3 * g]; a1 = xw4[3 * g + 1]; a2 = xw4[3 * g + 2]; b0 = bv4[3 * g]; b1 = bv4[3 * g + 1]; b2 = bv4[3 * g + 2]; }
  while (g < full) {   // the streaming pipeline of moments_kernel: the next group's loads are in flight while this one is evaluated
    const int64_t gn = g + stride;
    const int64_t gl = gn < full ? gn : g;
    const V na0 = xw4[3 * gl], na1 = xw4[3 * gl + 1], na2 = xw4[3 * gl + 2];
    const V nb0 = bv4[3 * gl], nb1 = bv4[3 * gl + 1], nb2 = bv4[3 * gl + 2];
    T vw[3 * P], vb[3 * P];
    unpack3(a0, a1, a2, vw);
    unpack3(b0, b1, b2, vb);
#pragma unroll
    for (int i = 0; i < P; i++) acc[0] += (double)sine_error<T>(q, vw[3 * i], vw[3 * i + 1], vw[3 * i + 2], vb[3 * i], vb[3 * i + 1], vb[3 * i + 2]);
    acc[1] += (double)P;
    a0 = na0; a1 = na1; a2 = na2; b0 = nb0; b1 = nb1; b2 = nb2;
    g = gn;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && full * P < n) {
    T vw[3 * P], vb[3 * P];
    load_group<T>(xw, full, n, vw);
    load_group<T>(bv, full, n, vb);
    for (int i = 0; i < (int)(n - full * P); i++) {
      acc[0] += (double)sine_error<T>(q, vw[3 * i], vw[3 * i + 1], vw[3 * i + 2], vb[3 * i], vb[3 * i + 1], vb[3 * i + 2]);
      acc[1] += 1.0;
    }
  }
  reduce_and_finish<2, kNeLd, 0, BLK>(acc, fin);
}

// ================================================================================================
// K5 : one round of nl_shinji_kneip_ls + find_opt_cc  (AbsoluteOrientationNormal.hpp:484-505, :24-39)
// record (44): M23 (9) TW K | M33 (9) sigma | MNN (9) TL M | AA xx xy xz yy yz zz | bb (3) | pad
// ================================================================================================
struct NlParams { double c_opt[3], Cw[3], Cc[3], Rwc[9]; };

// one correspondence of the round: the three modality blocks, each switched on by its inlier flag (predicated, no branches).
// Arithmetic in the array dtype T (the reference's Tp, AbsoluteOrientationNormal.hpp:484-505, accumulates in Tp as well); the
// caller adds each group's partial sums s[] into fp64 accumulators, so only the per-term rounding of T remains (as in K1-K3).
template <class T>
struct NlConst { T c_opt[3], Cw[3], Cc[3], Rwc[9]; };
template <class T> __device__ __forceinline__ NlConst<T> nl_const(const NlParams& p) {
  NlConst<T> k;
#pragma unroll
  for (int i = 0; i < 3; i++) { k.c_opt[i] = (T)p.c_opt[i]; k.Cw[i] = (T)p.Cw[i]; k.Cc[i] = (T)p.Cc[i]; }
#pragma unroll
  for (int i = 0; i < 9; i++) k.Rwc[i] = (T)p.Rwc[i];
  return k;
}
// The three modality blocks of one correspondence.  Every slot of the record belongs to exactly one block (2D-3D: 0-10 and 32-40,
// 3D-3D: 11-20, normals: 21-31), so the blocks of a group can run one after the other without changing any slot's summation order.
template <class T>
__device__ __forceinline__ void nl_term23(const NlConst<T>& prm, T x, T y, T z, bool on, T w23v, T bx_, T by_, T bz_, T (&acc)[44]) {
  // 2D-3D inliers: M23 and the find_opt_cc sums.  w = 0 switches the term off.
  const T w = on ? w23v : T(0);
  T ax = x - prm.c_opt[0], ay = y - prm.c_opt[1], az = z - prm.c_opt[2];
  const T n2 = ax * ax + ay * ay + az * az;
  const T inv = T(1) / sqrt(n2);
  ax = on ? ax * inv : T(0); ay = on ? ay * inv : T(0); az = on ? az * inv : T(0);  // selects: NaN * 0 must not reach the sums
  const T bx = on ? bx_ : T(0), by = on ? by_ : T(0), bz = on ? bz_ : T(0);
  acc[0] = fma(w * bx, ax, acc[0]); acc[1] = fma(w * bx, ay, acc[1]); acc[2] = fma(w * bx, az, acc[2]);
  acc[3] = fma(w * by, ax, acc[3]); acc[4] = fma(w * by, ay, acc[4]); acc[5] = fma(w * by, az, acc[5]);
  acc[6] = fma(w * bz, ax, acc[6]); acc[7] = fma(w * bz, ay, acc[7]); acc[8] = fma(w * bz, az, acc[8]);
  acc[9] += w; acc[10] += on ? T(1) : T(0);
  // find_opt_cc: v = Rwc * bv ; A = I - v v^T ; AA += A ; bb += A * Xw
  const T vx = prm.Rwc[0] * bx + prm.Rwc[1] * by + prm.Rwc[2] * bz;
  const T vy = prm.Rwc[3] * bx + prm.Rwc[4] * by + prm.Rwc[5] * bz;
  const T vz = prm.Rwc[6] * bx + prm.Rwc[7] * by + prm.Rwc[8] * bz;
  const T o = on ? T(1) : T(0);
  const T xo = on ? x : T(0), yo = on ? y : T(0), zo = on ? z : T(0);
  const T Axx = o - vx * vx, Axy = -vx * vy, Axz = -vx * vz, Ayy = o - vy * vy, Ayz = -vy * vz, Azz = o - vz * vz;
  acc[32] += Axx; acc[33] += Axy; acc[34] += Axz; acc[35] += Ayy; acc[36] += Ayz; acc[37] += Azz;
  acc[38] += Axx * xo + Axy * yo + Axz * zo;
  acc[39] += Axy * xo + Ayy * yo + Ayz * zo;
  acc[40] += Axz * xo + Ayz * yo + Azz * zo;
}
template <class T>
__device__ __forceinline__ void nl_term33(const NlConst<T>& prm, T x, T y, T z, bool on, T w33v, T cx_, T cy_, T cz_, T (&acc)[44]) {
  // 3D-3D inliers: centred covariance and sigma
  const T v = on ? w33v : T(0);
  const T ax = on ? x - prm.Cw[0] : T(0), ay = on ? y - prm.Cw[1] : T(0), az = on ? z - prm.Cw[2] : T(0);
  const T cx = on ? cx_ - prm.Cc[0] : T(0), cy = on ? cy_ - prm.Cc[1] : T(0), cz = on ? cz_ - prm.Cc[2] : T(0);
  acc[20] += v * (cx * cx + cy * cy + cz * cz);
  acc[11] = fma(v * cx, ax, acc[11]); acc[12] = fma(v * cx, ay, acc[12]); acc[13] = fma(v * cx, az, acc[13]);
  acc[14] = fma(v * cy, ax, acc[14]); acc[15] = fma(v * cy, ay, acc[15]); acc[16] = fma(v * cy, az, acc[16]);
  acc[17] = fma(v * cz, ax, acc[17]); acc[18] = fma(v * cz, ay, acc[18]); acc[19] = fma(v * cz, az, acc[19]);
}
template <class T>
__device__ __forceinline__ void nl_termnn(bool on, T wnnv, T nwx, T nwy, T nwz, T ncx, T ncy, T ncz, T (&acc)[44]) {
  // normal-normal inliers
  const T l = on ? wnnv : T(0);
  const T ax = on ? nwx : T(0), ay = on ? nwy : T(0), az = on ? nwz : T(0);
  const T cx = on ? ncx : T(0), cy = on ? ncy : T(0), cz = on ? ncz : T(0);
  acc[21] = fma(l * cx, ax, acc[21]); acc[22] = fma(l * cx, ay, acc[22]); acc[23] = fma(l * cx, az, acc[23]);
  acc[24] = fma(l * cy, ax, acc[24]); acc[25] = fma(l * cy, ay, acc[25]); acc[26] = fma(l * cy, az, acc[26]);
  acc[27] = fma(l * cz, ax, acc[27]); acc[28] = fma(l * cz, ay, acc[28]); acc[29] = fma(l * cz, az, acc[29]);
  acc[30] += l; acc[31] += on ? T(1) : T(0);
}
template <class T>
__device__ __forceinline__ void nl_point(const NlConst<T>& prm, T x, T y, T z, bool on23, T w23v, T bx_, T by_, T bz_, bool on33, T w33v,
                                         T cx_, T cy_, T cz_, bool onnn, T wnnv, T nwx, T nwy, T nwz, T ncx, T ncy, T ncz, T (&acc)[44]) {
  nl_term23<T>(prm, x, y, z, on23, w23v, bx_, by_, bz_, acc);
  nl_term33<T>(prm, x, y, z, on33, w33v, cx_, cy_, cz_, acc);
  nl_termnn<T>(onnn, wnnv, nwx, nwy, nwz, ncx, ncy, ncz, acc);
}
template <int A, int B, class T> __device__ __forceinline__ void nl_flush(double (&acc)[44], const T (&sg)[44]) {
#pragma unroll
  for (int k = A; k < B; k++) acc[k] += (double)sg[k];
}

// generic form: any subset of the arrays / masks / weights, bounds-checked loads
template <class T, int BLK>
__global__ __launch_bounds__(BLK) void nl_round_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const T* __restrict__ bv,
                                                       const T* __restrict__ nw, const T* __restrict__ nc,
                                                       const short* __restrict__ k23, const short* __restrict__ k33,
                                                       const short* __restrict__ knn, const T* __restrict__ w23,
                                                       const T* __restrict__ w33, const T* __restrict__ wnn, int64_t n, NlParams prm,
                                                       Finish fin) {
  constexpr int P = Pk<T>::P;
  const NlConst<T> kc = nl_const<T>(prm);
  double acc[44];
#pragma unroll
  for (int k = 0; k < 44; k++) acc[k] = 0.0;
  const int64_t groups = (n + P - 1) / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  for (int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x; g < groups; g += stride) {
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    short a23[P], a33[P], ann[P];
    T u23[P], u33[P], unn[P];
    load_group<T>(xw, g, n, vw);
    load_mask_group(k23, g, n, a23);
    load_mask_group(k33, g, n, a33);
    if (knn) load_mask_group(knn, g, n, ann);
    if (w23) load_weight_group(w23, g, n, u23);
    if (w33) load_weight_group(w33, g, n, u33);
    if (wnn) load_weight_group(wnn, g, n, unn);
    if (bv) load_group<T>(bv, g, n, vb);
    if (xc) load_group<T>(xc, g, n, vc);
    if (nw) { load_group<T>(nw, g, n, vnw); load_group<T>(nc, g, n, vnc); }
    T sg[44];
#pragma unroll
    for (int k = 0; k < 44; k++) sg[k] = T(0);
#pragma unroll
    for (int i = 0; i < P; i++) {
      nl_point<T>(kc, vw[3 * i], vw[3 * i + 1], vw[3 * i + 2], bv != nullptr && a23[i] == 1, w23 ? u23[i] : T(1), vb[3 * i],
          vb[3 * i + 1],
                  vb[3 * i + 2], xc != nullptr && a33[i] == 1, w33 ? u33[i] : T(1), vc[3 * i], vc[3 * i + 1], vc[3 * i + 2],
                  nw != nullptr && knn != nullptr && ann[i] == 1, wnn ? unn[i] : T(1), vnw[3 * i], vnw[3 * i + 1], vnw[3 * i + 2],
                  vnc[3 * i], vnc[3 * i + 1], vnc[3 * i + 2], sg);
    }
#pragma unroll
    for (int k = 0; k < 44; k++) acc[k] += (double)sg[k];
  }
  reduce_and_finish<44, kNlLd, 0, BLK>(acc, fin);
}

template <class T> __device__ __forceinline__ void pin_weights(T (&u)[Pk<T>::P]) {
  typedef typename Pk<T>::V V;
  V t;
  __builtin_memcpy(&t, u, 16);
  pin16_here(t);
  __builtin_memcpy(u, &t, 16);
}
template <class T, int BLK, bool WEIGHT>
__global__ __launch_bounds__(BLK, 512 / BLK) void nl_round_full_kernel(const T* __restrict__ xw, const T* __restrict__ xc,
                                                           const T* __restrict__ bv, const T* __restrict__ nw, const T* __restrict__ nc,
                                                           const short* __restrict__ k23, const short* __restrict__ k33,
                                                           const short* __restrict__ knn, const T* __restrict__ w23,
                                                           const T* __restrict__ w33, const T* __restrict__ wnn, int64_t n, NlParams prm,
                                                           Finish fin) {
  constexpr int P = Pk<T>::P;
  typedef typename Pk<T>::V V;
  const NlConst<T> kc = nl_const<T>(prm);
  double acc[44];
#pragma unroll
  for (int k = 0; k < 44; k++) acc[k] = 0.0;
  const int64_t full = n / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const V* __restrict__ pw = reinterpret_cast<const V*>(xw);
  const V* __restrict__ pc = reinterpret_cast<const V*>(xc);
  const V* __restrict__ pb = reinterpret_cast<const V*>(bv);
  const V* __restrict__ pnw = reinterpret_cast<const V*>(nw);
  const V* __restrict__ pnc = reinterpret_cast<const V*>(nc);
  int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
  V w0 = {}, w1 = {}, w2 = {}, b0 = {}, b1 = {}, b2 = {}, c0 = {}, c1 = {}, c2 = {}, p0 = {}, p1 = {}, p2 = {}, q0 = {}, q1 = {}, q2 = {};
  short m23[P], m33[P], mnn[P];
  T u23[P], u33[P], unn[P];
#pragma unroll
  for (int i = 0; i < P; i++) { m23[i] = m33[i] = mnn[i] = 0; u23[i] = u33[i] = unn[i] = T(1); }
  if (g < full) {
    w0 = pw[3 * g]; w1 = pw[3 * g + 1]; w2 = pw[3 * g + 2];
    b0 = pb[3 * g]; b1 = pb[3 * g + 1]; b2 = pb[3 * g + 2];
    load_mask_full(k23, g, m23);
    if (WEIGHT) load_weight_full(w23, g, u23);
    c0 = pc[3 * g]; c1 = pc[3 * g + 1]; c2 = pc[3 * g + 2];
    load_mask_full(k33, g, m33);
    if (WEIGHT) load_weight_full(w33, g, u33);
    p0 = pnw[3 * g]; p1 = pnw[3 * g + 1]; p2 = pnw[3 * g + 2];
    q0 = pnc[3 * g]; q1 = pnc[3 * g + 1]; q2 = pnc[3 * g + 2];
    load_mask_full(knn, g, mnn);
    if (WEIGHT) load_weight_full(wnn, g, unn);
  }
  while (g < full) {
    const int64_t gn = g + stride;
    const int64_t gl = gn < full ? gn : g;  // clamp: the last trip re-reads its own (cached) group instead of branching
    const V x0 = pw[3 * gl], x1 = pw[3 * gl + 1], x2 = pw[3 * gl + 2];   // the world points are read by two blocks: double-buffered
    __builtin_amdgcn_sched_barrier(0);
    T vw[3 * P], va[3 * P], vb[3 * P];
    T sg[44];
#pragma unroll
    for (int k = 0; k < 44; k++) sg[k] = T(0);
    // the vectors a block consumes become opaque 16-byte values HERE: any repacking the optimiser wants for its packed arithmetic
    // happens after this point, not right behind the loads (where it would wait for them with everything else still in flight)
    pin16_here(w0); pin16_here(w1); pin16_here(w2); pin16_here(b0); pin16_here(b1); pin16_here(b2);
    if (WEIGHT) pin_weights<T>(u23);
    unpack3(w0, w1, w2, vw);
    unpack3(b0, b1, b2, va);
#pragma unroll
    for (int i = 0; i < P; i++) {
      nl_term23<T>(kc, vw[3 * i], vw[3 * i + 1], vw[3 * i + 2], m23[i] == 1, WEIGHT ? u23[i] : T(1), va[3 * i], va[3 * i + 1], va[3 * i + 2], sg);
    }
    nl_flush<0, 11>(acc, sg); nl_flush<32, 41>(acc, sg);
    __builtin_amdgcn_sched_barrier(0);
    b0 = pb[3 * gl]; b1 = pb[3 * gl + 1]; b2 = pb[3 * gl + 2];
    load_mask_full(k23, gl, m23);
    if (WEIGHT) load_weight_full(w23, gl, u23);
    __builtin_amdgcn_sched_barrier(0);
    pin16_here(c0); pin16_here(c1); pin16_here(c2);
    if (WEIGHT) pin_weights<T>(u33);
    unpack3(c0, c1, c2, va);
#pragma unroll
    for (int i = 0; i < P; i++) {
      nl_term33<T>(kc, vw[3 * i], vw[3 * i + 1], vw[3 * i + 2], m33[i] == 1, WEIGHT ? u33[i] : T(1), va[3 * i], va[3 * i + 1], va[3 * i + 2], sg);
    }
    nl_flush<11, 21>(acc, sg);
    __builtin_amdgcn_sched_barrier(0);
    c0 = pc[3 * gl]; c1 = pc[3 * gl + 1]; c2 = pc[3 * gl + 2];
    load_mask_full(k33, gl, m33);
    if (WEIGHT) load_weight_full(w33, gl, u33);
    __builtin_amdgcn_sched_barrier(0);
    pin16_here(p0); pin16_here(p1); pin16_here(p2); pin16_here(q0); pin16_here(q1); pin16_here(q2);
    if (WEIGHT) pin_weights<T>(unn);
    unpack3(p0, p1, p2, va);
    unpack3(q0, q1, q2, vb);
#pragma unroll
    for (int i = 0; i < P; i++) {
      nl_termnn<T>(mnn[i] == 1, WEIGHT ? unn[i] : T(1), va[3 * i], va[3 * i + 1], va[3 * i + 2], vb[3 * i], vb[3 * i + 1], vb[3 * i + 2], sg);
    }
    nl_flush<21, 32>(acc, sg);
    __builtin_amdgcn_sched_barrier(0);
    p0 = pnw[3 * gl]; p1 = pnw[3 * gl + 1]; p2 = pnw[3 * gl + 2];
    q0 = pnc[3 * gl]; q1 = pnc[3 * gl + 1]; q2 = pnc[3 * gl + 2];
    load_mask_full(knn, gl, mnn);
    if (WEIGHT) load_weight_full(wnn, gl, unn);
    __builtin_amdgcn_sched_barrier(0);
    w0 = x0; w1 = x1; w2 = x2;
    g = gn;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && full * P < n) {  // leftover correspondences through the bounds-checked loaders
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    short a23[P], a33[P], ann[P];
    T t23[P], t33[P], tnn[P];
    load_group<T>(xw, full, n, vw); load_group<T>(xc, full, n, vc); load_group<T>(bv, full, n, vb);
    load_group<T>(nw, full, n, vnw); load_group<T>(nc, full, n, vnc);
    load_mask_group(k23, full, n, a23); load_mask_group(k33, full, n, a33); load_mask_group(knn, full, n, ann);
    if (WEIGHT) { load_weight_group(w23, full, n, t23); load_weight_group(w33, full, n, t33); load_weight_group(wnn, full, n, tnn); }
    T sg[44];
#pragma unroll
    for (int k = 0; k < 44; k++) sg[k] = T(0);
#pragma unroll
    for (int i = 0; i < P; i++) {
      nl_point<T>(kc, vw[3 * i], vw[3 * i + 1], vw[3 * i + 2], a23[i] == 1, WEIGHT ? t23[i] : T(1), vb[3 * i], vb[3 * i + 1],
                  vb[3 * i + 2], a33[i] == 1, WEIGHT ? t33[i] : T(1), vc[3 * i], vc[3 * i + 1], vc[3 * i + 2], ann[i] == 1,
                  WEIGHT ? tnn[i] : T(1), vnw[3 * i], vnw[3 * i + 1], vnw[3 * i + 2], vnc[3 * i], vnc[3 * i + 1], vnc[3 * i + 2], sg);
    }
    nl_flush<0, 44>(acc, sg);
  }
  reduce_and_finish<44, kNlLd, 0, BLK>(acc, fin);
}

// ---- after a collective: copy the reduced record from HBM to the pinned host slot and raise the sequence word
template <class E>
__global__ void publish_kernel(const E* __restrict__ src, int count, E* __restrict__ h_dst, unsigned long long* __restrict__ h_flag,
                               unsigned long long seq) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) __hip_atomic_store(h_dst + i, src[i], __ATOMIC_RELAXED,
      __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(h_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... as tagged 16-byte pairs {value, sequence} (store_tagged_pair: one system-scope store each, the tag travels with the value): no
// drain of the PCIe writes in front of a flag -- the host waits until every pair carries the sequence value (wait_host_partials)
__global__ void publish_pairs_kernel(const double* __restrict__ src, int count, double* __restrict__ h_pairs, unsigned long long seq) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) store_tagged_pair(h_pairs, i, src[i], seq);
}
hipError_t launch_publish_pairs(const double* d_src, int count, double* h_pairs, unsigned long long seq, hipStream_t s) {
  hipLaunchKernelGGL(publish_pairs_kernel, dim3(1), dim3(256), 0, s, d_src, count, h_pairs, seq);
  return hipGetLastError();
}
hipError_t launch_publish_i32(const int* d_src, int count, int* h_dst, unsigned long long* h_flag, unsigned long long seq,
    hipStream_t s) {
  hipLaunchKernelGGL((publish_kernel<int>), dim3(1), dim3(256), 0, s, d_src, count, h_dst, h_flag, seq);
  return hipGetLastError();
}
// vote counters: publish to the host AND clear them for the next scoring launch (the counters are accumulated with atomics, so
// they must start at zero; clearing here saves a memset per launch)
__global__ void publish_votes_kernel(int* __restrict__ votes, int count, int* __restrict__ h_dst, unsigned long long* __restrict__ h_flag,
                                     unsigned long long seq) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    __hip_atomic_store(h_dst + i, votes[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    votes[i] = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(h_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// sharded variant: all-reduce(sum) of the counters across the ranks through the peers' mailboxes before publishing (one 8-byte
// word {count | step tag} per hypothesis and source rank; rank-ordered integer sums; bounded wait like p2p_allreduce32)
__global__ __launch_bounds__(1024) void publish_votes_p2p_kernel(int* __restrict__ votes, int count, const P2PDesc* __restrict__ desc,
                                                                 unsigned long long step, int* __restrict__ h_dst, int* __restrict__ h_status,
                                                                 unsigned long long* __restrict__ h_flag, unsigned long long seq) {
  const P2PDesc& D = *desc;
  const unsigned int tag = (unsigned int)(step % 0xFFFFFFFFull) + 1u;
  const size_t parity = (size_t)(step & 1ull);
  const size_t base = kP2PRecordWords + parity * kP2PMaxWorld * kMaxScoreH;
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    const unsigned long long word = ((unsigned long long)tag << 32) | (unsigned int)votes[i];
    for (int r = 0; r < D.world; r++)
      __hip_atomic_store(D.peer[r] + base + (size_t)D.rank * kMaxScoreH + i, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const unsigned long long* box = D.peer[D.rank] + base;
  const unsigned long long t0 = wall_clock64();
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    int total = 0;
    for (int r = 0; r < D.world; r++) {
      unsigned long long w;
      for (;;) {
        w = __hip_atomic_load(box + (size_t)r * kMaxScoreH + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned int)(w >> 32) == tag) break;
        if (wall_clock64() - t0 > 1000000000ull) { s_bad = 1; break; }
      }
      total += (int)(unsigned int)w;
    }
    __hip_atomic_store(h_dst + i, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    votes[i] = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(h_status, s_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(h_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
hipError_t launch_publish_votes_p2p(int* d_votes, int count, const P2PDesc* p2p, unsigned long long step, int* h_dst, int* h_status,
                                    unsigned long long* h_flag, unsigned long long seq, hipStream_t s) {
  hipLaunchKernelGGL(publish_votes_p2p_kernel, dim3(1), dim3(count > 256 ? 1024 : 256), 0, s, d_votes, count, p2p, step, h_dst,
      h_status, h_flag, seq);
  return hipGetLastError();
}
hipError_t launch_publish_votes(int* d_votes, int count, int* h_dst, unsigned long long* h_flag, unsigned long long seq,
    hipStream_t s) {
  hipLaunchKernelGGL(publish_votes_kernel, dim3(1), dim3(count > 256 ? 1024 : 256), 0, s, d_votes, count, h_dst, h_flag, seq);
  return hipGetLastError();
}

template <class T, int BLK>
static void moments_launch(const DeviceArrays& A, int flags, const ReduceTarget& rt, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  const short* mask = (flags & F_USE_MASK) ? A.mask[1] : nullptr;
  const T* weight = (flags & F_USE_WEIGHT) ? (const T*)A.weight[1] : nullptr;
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks, BLK);
  const Finish fin = make_finish(rt);
  const int skip = (flags & F_SKIP_INVALID) ? 1 : 0;
  const T* xw = (const T*)A.a[0];
  const T* xc = (const T*)A.a[1];
  if (mask && weight) RPE_LAUNCH_EV((moments_kernel<T, BLK, true, true>), dim3(G), dim3(BLK), 0, s, e0, e1, xw, xc, mask, weight, A.n, skip, fin);
  else if (mask) RPE_LAUNCH_EV((moments_kernel<T, BLK, true, false>), dim3(G), dim3(BLK), 0, s, e0, e1, xw, xc, mask, weight, A.n, skip, fin);
  else if (weight) RPE_LAUNCH_EV((moments_kernel<T, BLK, false, true>), dim3(G), dim3(BLK), 0, s, e0, e1, xw, xc, mask, weight, A.n, skip, fin);
  else RPE_LAUNCH_EV((moments_kernel<T, BLK, false, false>), dim3(G), dim3(BLK), 0, s, e0, e1, xw, xc, mask, weight, A.n, skip, fin);
}
template <class T>
static hipError_t moments_t(const DeviceArrays& A, int flags, const ReduceTarget& rt, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  const int blk = pick_block(rt);
  if (blk == 512) moments_launch<T, 512>(A, flags, rt, s, e0, e1);
  else moments_launch<T, 256>(A, flags, rt, s, e0, e1);
  return hipGetLastError();
}
hipError_t launch_moments(const DeviceArrays& A, int flags, const ReduceTarget& rt, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  return A.dtype ? moments_t<double>(A, flags, rt, s, e0, e1) : moments_t<float>(A, flags, rt, s, e0, e1);
}

template <class T>
static hipError_t sine_error_t(const DeviceArrays& A, const double* pose7, const ReduceTarget& rt, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  SinePose<T> q;
  q.qw = (T)pose7[0]; q.qx = (T)pose7[1]; q.qy = (T)pose7[2]; q.qz = (T)pose7[3];
  for (int i = 0; i < 3; i++) q.t[i] = (T)pose7[4 + i];
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks, 256);
  RPE_LAUNCH_EV((sine_error_kernel<T, 256>), dim3(G), dim3(256), 0, s, e0, e1, (const T*)A.a[0], (const T*)A.a[2], A.n, q, make_finish(rt));
  return hipGetLastError();
}
hipError_t launch_sine_error(const DeviceArrays& A, const double* pose7, const ReduceTarget& rt, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  return A.dtype ? sine_error_t<double>(A, pose7, rt, s, e0, e1) : sine_error_t<float>(A, pose7, rt, s, e0, e1);
}

template <class T, int BLK>
static void nl_round_launch(const DeviceArrays& A, const NlParams& prm, const ReduceTarget& rt, hipStream_t s, hipEvent_t e0,
    hipEvent_t e1) {
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks, BLK);
  const bool all_arrays = A.a[0] && A.a[1] && A.a[2] && A.a[3] && A.a[4] && A.mask[0] && A.mask[1] && A.mask[2];
  const int nweights = (A.weight[0] != nullptr) + (A.weight[1] != nullptr) + (A.weight[2] != nullptr);
#define RPE_NL_ARGS (const T*)A.a[0], (const T*)A.a[1], (const T*)A.a[2], (const T*)A.a[3], (const T*)A.a[4], (const short*)A.mask[0],      \
                    (const short*)A.mask[1], (const short*)A.mask[2], (const T*)A.weight[0], (const T*)A.weight[1], (const T*)A.weight[2], A.n, prm, \
                    make_finish(rt)
  if (all_arrays && nweights == 0) RPE_LAUNCH_EV((nl_round_full_kernel<T, BLK, false>), dim3(G), dim3(BLK), 0, s, e0, e1, RPE_NL_ARGS);
  else if (all_arrays && nweights == 3) RPE_LAUNCH_EV((nl_round_full_kernel<T, BLK, true>), dim3(G), dim3(BLK), 0, s, e0, e1, RPE_NL_ARGS);
  else RPE_LAUNCH_EV((nl_round_kernel<T, BLK>), dim3(G), dim3(BLK), 0, s, e0, e1, RPE_NL_ARGS);
#undef RPE_NL_ARGS
}
template <class T>
static hipError_t nl_round_t(const DeviceArrays& A, const double* params24, const ReduceTarget& rt, hipStream_t s, hipEvent_t e0,
    hipEvent_t e1) {
  NlParams prm;
  for (int i = 0; i < 3; i++) { prm.c_opt[i] = params24[i]; prm.Cw[i] = params24[3 + i]; prm.Cc[i] = params24[6 + i]; }
  for (int i = 0; i < 9; i++) prm.Rwc[i] = params24[9 + i];
  // 44 fp64 accumulators + one group of streamed data need ~240 VGPRs (two waves per SIMD, no spills): 256-thread workgroups unless
  // the caller's reduce target asks for 512 (the same two waves per SIMD in half as many workgroups)
  // (fp64 arrays: 256 only -- the generic form's 512-thread instance is 3 registers short)
  if (rt.block == 512 && sizeof(T) == 4) { if constexpr (sizeof(T) == 4) nl_round_launch<T, 512>(A, prm, rt, s, e0, e1); }
  else nl_round_launch<T, 256>(A, prm, rt, s, e0, e1);
  return hipGetLastError();
}
hipError_t launch_nl_round(const DeviceArrays& A, const double* params24, const ReduceTarget& rt, hipStream_t s, hipEvent_t e0,
    hipEvent_t e1) {
  return A.dtype ? nl_round_t<double>(A, params24, rt, s, e0, e1) : nl_round_t<float>(A, params24, rt, s, e0, e1);
}

void preload_nl() {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, (const void*)publish_votes_kernel) != hipSuccess) (void)hipGetLastError();
}

}  // namespace rpe
