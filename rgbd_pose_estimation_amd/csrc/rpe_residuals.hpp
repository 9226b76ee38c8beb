// Residual / Jacobian arithmetic of the Gauss-Newton normal equations (K1 point-to-point, K2 point-to-plane, K3 bearing) and the
// resident-loop stages shared by the normal-equation and the ICP kernels.
#pragma once
#include "rpe_reduce.hpp"

namespace rpe {

// ================================================================================================
// K1 / K2 / K3 : Gauss-Newton normal equations
// ================================================================================================
// p2p keeps 17 structured sums: w | w p (3) | w p p^T (6) | w r (3) | w p x r (3) | w r^2
// The pose stays fp64 and p = R x + t, r = p - Xc are formed in fp64: the subtraction cancels ~3 digits
// (|p| ~ 10 m, |r| ~ 5 cm), so doing it in fp32 would dominate the error budget.  p and r are then rounded
// to the compute type C for the products (fp32 for fp32 arrays), and the sums are widened to fp64 per group.
// general packed record: H upper triangle (21) | g (6) | w r^2 | w
template <class C> __device__ __forceinline__ void add_row(const C (&J)[6], C r, C w, C (&s)[29]) {
  int k = 0;
#pragma unroll
  for (int a = 0; a < 6; a++) {
    const C wa = w * J[a];
#pragma unroll
    for (int b = a; b < 6; b++) { s[k] = fma(wa, J[b], s[k]); k++; }
    s[21 + a] = fma(wa, r, s[21 + a]);
  }
  s[27] = fma(w * r, r, s[27]);
}
// Two correspondences at once: the 35 products of a Jacobian row's outer product as 2-vectors (packed fp32 instructions for fp32
// arrays), each lane of the pair keeping its own partial sums; the pair's sums are added at the end of the group.
template <class V> __device__ __forceinline__ void add_row2(const V (&J)[6], V r, V w, V (&s)[29]) {
  int k = 0;
#pragma unroll
  for (int a = 0; a < 6; a++) {
    const V wa = w * J[a];
#pragma unroll
    for (int b = a; b < 6; b++) { s[k] = __builtin_elementwise_fma(wa, J[b], s[k]); k++; }
    s[21 + a] = __builtin_elementwise_fma(wa, r, s[21 + a]);
  }
  s[27] = __builtin_elementwise_fma(w * r, r, s[27]);
}
template <class C>
__device__ __forceinline__ void p2plane_pair(const PoseK<double>& T, const C (&x)[2], const C (&y)[2], const C (&z)[2], const C (&cx)[2],
                                             const C (&cy)[2], const C (&cz)[2], const C (&nx)[2], const C (&ny)[2], const C (&nz)[2],
                                             const C (&w)[2], C __attribute__((ext_vector_type(2))) (&s)[29]) {
  typedef C V __attribute__((ext_vector_type(2)));
  C px[2], py[2], pz[2], r[2];
#pragma unroll
  for (int e = 0; e < 2; e++) {   // the fp64 part stays per point: transform and the (cancelling) residual
    double pxd, pyd, pzd;
    transform<C>(T, x[e], y[e], z[e], pxd, pyd, pzd);
    px[e] = (C)pxd; py[e] = (C)pyd; pz[e] = (C)pzd;
    r[e] = (C)((double)nx[e] * (pxd - (double)cx[e]) + (double)ny[e] * (pyd - (double)cy[e]) + (double)nz[e] * (pzd - (double)cz[e]));
  }
  const V PX = {px[0], px[1]}, PY = {py[0], py[1]}, PZ = {pz[0], pz[1]}, NX = {nx[0], nx[1]}, NY = {ny[0], ny[1]}, NZ = {nz[0], nz[1]};
  const V J[6] = {NX, NY, NZ, PY * NZ - PZ * NY, PZ * NX - PX * NZ, PX * NY - PY * NX};  // [n ; p x n]
  const V R = {r[0], r[1]}, W = {w[0], w[1]};
  add_row2<V>(J, R, W, s);
  s[28] += W;
}
// ---- K3, bearing (sine) residual  r = p^ x bv  (P3P.hpp:482-485; p^ = p / |p|).
// Both r = -[bv]x p^ and its Jacobian -[bv]x (I - p^ p^T) / |p| [I | -[p]x] lie in the plane orthogonal to bv, so in an
// orthonormal basis (e1, e2) of that plane TWO rows carry all of J^T J, J^T r and |r|^2 (the third would be zero):
//     rho_i = e_i . p^ ,   J_i = a_i^T [I | -[p]x] ,   a_i = (e_i - rho_i p^) / |p| ,   weight  w |bv|^2
// (with p^ = alpha e1 + beta e2 + gamma bv^:  r = |bv| (beta e1 - alpha e2), i.e. the rows above up to order and sign, which
// neither J^T J nor J^T r sees) -- 70 accumulation FMAs per correspondence instead of 105, no fp64 cross product and no fp64
// normalisation.  What stays fp64 is what cancels: p = R x + t and the two dots e_i . p (near the optimum p^ ~ bv is orthogonal
// to e_i); 1 / |p| only scales rho_i and J_i relatively, so the fp32 hardware rsqrt serves.  The basis is the branch-free
// construction of Duff et al. ("Building an orthonormal basis, revisited", 2017) from bv itself: it depends on the
// correspondence only, never on the pose, so its rounding (e_i off the plane by <= 1e-7) acts like a fixed 1e-7 rad bearing
// offset per correspondence, three orders below the measurement noise.
// Lane-type helpers: V is a scalar (float, double) or a 2-vector of them (pairs of correspondences: packed fp32 instructions).
template <class V> struct LaneOps;
template <> struct LaneOps<float> {
  static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static __device__ __forceinline__ float sign1(float x) { return x < 0.f ? -1.f : 1.f; }
};
template <> struct LaneOps<double> {
  // fp32 estimate + two Newton steps: ~1 ulp of fp64 without the IEEE sqrt + divide sequences
  static __device__ __forceinline__ double rsqrt(double x) {
    double y = (double)__builtin_amdgcn_rsqf((float)x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    y = y * fma(-hx * y, y, 1.5);
    return y;
  }
  static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
  static __device__ __forceinline__ double sign1(double x) { return x < 0.0 ? -1.0 : 1.0; }
};
template <class C> struct LaneOps<C __attribute__((ext_vector_type(2)))> {
  typedef C V __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ V rsqrt(V x) { return V{LaneOps<C>::rsqrt(x.x), LaneOps<C>::rsqrt(x.y)}; }
  static __device__ __forceinline__ V rcp(V x) { return V{LaneOps<C>::rcp(x.x), LaneOps<C>::rcp(x.y)}; }
  static __device__ __forceinline__ V sign1(V x) { return V{LaneOps<C>::sign1(x.x), LaneOps<C>::sign1(x.y)}; }
};
// (e1, e2): orthonormal basis of the plane orthogonal to the unit vector (bx, by, bz)
template <class V>
__device__ __forceinline__ void tangent_basis(V bx, V by, V bz, V (&e1)[3], V (&e2)[3]) {
  const V sg = LaneOps<V>::sign1(bz);
  const V a = -LaneOps<V>::rcp(sg + bz);
  const V bxa = bx * a, c = bxa * by;
  e1[0] = __builtin_elementwise_fma(sg * bx, bxa, V(1)); e1[1] = sg * c; e1[2] = -sg * bx;
  e2[0] = c; e2[1] = __builtin_elementwise_fma(by * by, a, sg); e2[2] = -by;
}
// the two rows of one correspondence (or of a pair) into the packed record.  p: the transformed point rounded to the compute type;
// d1, d2: e1 . p and e2 . p formed in fp64 by the caller and rounded; w: weight of the rows (0 switches the correspondence off),
// w_count: what the weight sum s[28] receives (the joint kernel scales w per term, the weight sum stays unscaled).
template <class V>
__device__ __forceinline__ void bearing_rows(V px, V py, V pz, const V (&e1)[3], const V (&e2)[3], V d1, V d2, V bx, V by, V bz, V w,
                                             V w_count, V (&s)[29]) {
  // + 1e-30: a switched-off point (w = 0) sitting at p = 0 keeps 1 / |p| finite, so that 0 x finite = 0 reaches the sums
  const V inv = LaneOps<V>::rsqrt(__builtin_elementwise_fma(px, px, __builtin_elementwise_fma(py, py,
      __builtin_elementwise_fma(pz, pz, V(1e-30f)))));
  const V hx = px * inv, hy = py * inv, hz = pz * inv;
  const V wb = w * __builtin_elementwise_fma(bx, bx, __builtin_elementwise_fma(by, by, bz * bz));   // |bv|^2 = 1 to rounding
  const V rho[2] = {d1 * inv, d2 * inv};
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const V (&e)[3] = u == 0 ? e1 : e2;
    const V a0 = __builtin_elementwise_fma(-rho[u], hx, e[0]) * inv, a1 = __builtin_elementwise_fma(-rho[u], hy, e[1]) * inv,
            a2 = __builtin_elementwise_fma(-rho[u], hz, e[2]) * inv;
    // J = a^T [I | -[p]x] : translation part a, rotation part (p x a)
    const V J[6] = {a0, a1, a2, py * a2 - pz * a1, pz * a0 - px * a2, px * a1 - py * a0};
    add_row2<V>(J, rho[u], wb, s);
  }
  s[28] += w_count;
}
// e . p with p in fp64 (the cancelling part of the residual)
__device__ __forceinline__ double dot_e_p(double ex, double ey, double ez, double px, double py, double pz) {
  return fma(ex, px, fma(ey, py, ez * pz));
}
// one correspondence, scalar lanes (the joint kernel's bearing term)
template <class C>
__device__ __forceinline__ void bearing_point(double pxd, double pyd, double pzd, C bx, C by, C bz, C w, C w_count, C (&s)[29]) {
  C e1[3], e2[3];
  tangent_basis<C>(bx, by, bz, e1, e2);
  const C d1 = (C)dot_e_p(e1[0], e1[1], e1[2], pxd, pyd, pzd), d2 = (C)dot_e_p(e2[0], e2[1], e2[2], pxd, pyd, pzd);
  bearing_rows<C>((C)pxd, (C)pyd, (C)pzd, e1, e2, d1, d2, bx, by, bz, w, w_count, s);
}
// |r| = |p^ x bv| of one correspondence (the robust weights' argument), from the same two rows
template <class C>
__device__ __forceinline__ C bearing_residual_norm(double pxd, double pyd, double pzd, C bx, C by, C bz) {
  C e1[3], e2[3];
  tangent_basis<C>(bx, by, bz, e1, e2);
  const C d1 = (C)dot_e_p(e1[0], e1[1], e1[2], pxd, pyd, pzd), d2 = (C)dot_e_p(e2[0], e2[1], e2[2], pxd, pyd, pzd);
  const C px = (C)pxd, py = (C)pyd, pz = (C)pzd;
  const C inv = LaneOps<C>::rsqrt(fma(px, px, fma(py, py, pz * pz)));
  return sqrt((d1 * d1 + d2 * d2) * (bx * bx + by * by + bz * bz)) * inv;
}

// ---- K3', pixel reprojection residual (the pixel conversion of /root/reference/TestMain.cpp:35-36 with the
// principal point at the origin, PoseAdapterBase.hpp:44), in NORMALISED image coordinates (f = 1: the focal length multiplies r and J
// alike, so the step does not depend on it; callers that want pixel units scale the term by f^2):
//     r = (p_x / p_z - bv_x / bv_z ,  p_y / p_z - bv_y / bv_z) ,   J = 1 / p_z [[1, 0, -u], [0, 1, -v]] [I | -[p]x] ,  (u, v) = (p_x, p_y) / p_z
// i.e. two rows J_i = a_i^T [I | -[p]x] with a_1 = (1, 0, -u) / p_z, a_2 = (0, 1, -v) / p_z -- the shape of the bearing rows.  What
// cancels is the numerator of r_1 = (p_x bv_z - bv_x p_z) / (p_z bv_z): formed in fp64 from the fp64 point, the quotient's denominator
// and everything in J take the array dtype and the hardware reciprocal.  A correspondence whose point is not in front of the camera
// (p_z <= kReprojMinZ) or whose bearing has no forward component (bv_z <= kReprojMinZ) contributes nothing, and does not count.
constexpr double kReprojMinZ = 1e-6;
template <class V>
__device__ __forceinline__ void reproj_rows(V px, V py, V pz, V n1, V n2, V bz, V w, V w_count, V (&s)[29]) {
  const V ipz = LaneOps<V>::rcp(pz);
  const V inv = ipz * LaneOps<V>::rcp(bz);           // 1 / (p_z bv_z)
  const V r1 = n1 * inv, r2 = n2 * inv;
  const V g1 = -(px * ipz) * ipz, g2 = -(py * ipz) * ipz;    // third components of a_1, a_2
  // J = [a ; p x a] with a_1 = (ipz, 0, g1), a_2 = (0, ipz, g2)
  const V J1[6] = {ipz, V(0), g1, py * g1, __builtin_elementwise_fma(pz, ipz, -(px * g1)), -(py * ipz)};
  const V J2[6] = {V(0), ipz, g2, __builtin_elementwise_fma(py, g2, -(pz * ipz)), -(px * g2), px * ipz};
  add_row2<V>(J1, r1, w, s);
  add_row2<V>(J2, r2, w, s);
  s[28] += w_count;
}
// one correspondence: the fp64 part (numerators of the two quotients) and the validity test; a switched-off correspondence gets the
// harmless geometry p = (0, 0, 1), bv_z = 1 so that everything stays finite at weight 0
template <class C>
__device__ __forceinline__ void reproj_prepare(double pxd, double pyd, double pzd, C bx, C by, C bz, C& w, C& px, C& py, C& pz, C& n1,
                                               C& n2, C& bzs) {
  const bool ok = pzd > kReprojMinZ && (double)bz > kReprojMinZ;
  w = ok ? w : C(0);
  n1 = ok ? (C)fma(pxd, (double)bz, -(double)bx * pzd) : C(0);
  n2 = ok ? (C)fma(pyd, (double)bz, -(double)by * pzd) : C(0);
  px = ok ? (C)pxd : C(0); py = ok ? (C)pyd : C(0); pz = ok ? (C)pzd : C(1);
  bzs = ok ? bz : C(1);
}
template <class C>
__device__ __forceinline__ void reproj_pair(const PoseK<double>& T, const C (&x)[2], const C (&y)[2], const C (&z)[2], const C (&bx)[2],
                                            const C (&by)[2], const C (&bz)[2], const C (&w)[2], C __attribute__((ext_vector_type(2))) (&s)[29]) {
  typedef C V __attribute__((ext_vector_type(2)));
  C px[2], py[2], pz[2], n1[2], n2[2], bzs[2], wi[2];
#pragma unroll
  for (int e = 0; e < 2; e++) {
    double pxd, pyd, pzd;
    transform<C>(T, x[e], y[e], z[e], pxd, pyd, pzd);
    wi[e] = w[e];
    reproj_prepare<C>(pxd, pyd, pzd, bx[e], by[e], bz[e], wi[e], px[e], py[e], pz[e], n1[e], n2[e], bzs[e]);
  }
  reproj_rows<V>(V{px[0], px[1]}, V{py[0], py[1]}, V{pz[0], pz[1]}, V{n1[0], n1[1]}, V{n2[0], n2[1]}, V{bzs[0], bzs[1]},
                 V{wi[0], wi[1]}, V{wi[0], wi[1]}, s);
}
// one correspondence, scalar lanes (the joint kernel's reprojection term); w: scaled weight of the rows, w_count: what the weight sum gets
template <class C>
__device__ __forceinline__ void reproj_point(double pxd, double pyd, double pzd, C bx, C by, C bz, C w, C w_count, C (&s)[29]) {
  C px, py, pz, n1, n2, bzs, wc = w_count;
  reproj_prepare<C>(pxd, pyd, pzd, bx, by, bz, wc, px, py, pz, n1, n2, bzs);
  reproj_rows<C>(px, py, pz, n1, n2, bzs, wc == C(0) ? C(0) : w, wc, s);
}
// |r| of one correspondence (the robust weights' argument); 0 where the correspondence does not count
template <class C>
__device__ __forceinline__ C reproj_residual_norm(double pxd, double pyd, double pzd, C bx, C by, C bz) {
  if (!(pzd > kReprojMinZ && (double)bz > kReprojMinZ)) return C(0);
  const double r1 = pxd / pzd - (double)bx / (double)bz, r2 = pyd / pzd - (double)by / (double)bz;
  return (C)sqrt(r1 * r1 + r2 * r2);
}
// a pair of correspondences as 2-vectors
template <class C>
__device__ __forceinline__ void bearing_pair(const PoseK<double>& T, const C (&x)[2], const C (&y)[2], const C (&z)[2], const C (&bx)[2],
                                             const C (&by)[2], const C (&bz)[2], const C (&w)[2],
                                             C __attribute__((ext_vector_type(2))) (&s)[29]) {
  typedef C V __attribute__((ext_vector_type(2)));
  const V BX = {bx[0], bx[1]}, BY = {by[0], by[1]}, BZ = {bz[0], bz[1]};
  V e1[3], e2[3];
  tangent_basis<V>(BX, BY, BZ, e1, e2);
  C px[2], py[2], pz[2], d1[2], d2[2];
#pragma unroll
  for (int e = 0; e < 2; e++) {   // the fp64 part stays per point: transform and the two (cancelling) dots
    double pxd, pyd, pzd;
    transform<C>(T, x[e], y[e], z[e], pxd, pyd, pzd);
    px[e] = (C)pxd; py[e] = (C)pyd; pz[e] = (C)pzd;
    d1[e] = (C)dot_e_p(e1[0][e], e1[1][e], e1[2][e], pxd, pyd, pzd);
    d2[e] = (C)dot_e_p(e2[0][e], e2[1][e], e2[2][e], pxd, pyd, pzd);
  }
  bearing_rows<V>(V{px[0], px[1]}, V{py[0], py[1]}, V{pz[0], pz[1]}, e1, e2, V{d1[0], d1[1]}, V{d2[0], d2[1]}, BX, BY, BZ,
      V{w[0], w[1]},
                  V{w[0], w[1]}, s);
}


// ---- point-to-point on pairs of correspondences: the 17 structured sums as 2-vectors (packed fp32 instructions for fp32 arrays), each
// lane of the pair keeping its own partial sums.  p: the transformed point, r = p - Xc, both rounded to the compute type.
template <class V> __device__ __forceinline__ void p2p_accumulate(V px, V py, V pz, V rx, V ry, V rz, V w, V (&s)[17]) {
  const V wpx = w * px, wpy = w * py, wpz = w * pz;
  const V wrx = w * rx, wry = w * ry, wrz = w * rz;
  s[0] += w;
  s[1] += wpx; s[2] += wpy; s[3] += wpz;
  s[4] = __builtin_elementwise_fma(wpx, px, s[4]); s[5] = __builtin_elementwise_fma(wpx, py, s[5]);
  s[6] = __builtin_elementwise_fma(wpx, pz, s[6]); s[7] = __builtin_elementwise_fma(wpy, py, s[7]);
  s[8] = __builtin_elementwise_fma(wpy, pz, s[8]); s[9] = __builtin_elementwise_fma(wpz, pz, s[9]);
  s[10] += wrx; s[11] += wry; s[12] += wrz;
  s[13] += py * wrz - pz * wry;
  s[14] += pz * wrx - px * wrz;
  s[15] += px * wry - py * wrx;
  s[16] = __builtin_elementwise_fma(wrx, rx, __builtin_elementwise_fma(wry, ry, __builtin_elementwise_fma(wrz, rz, s[16])));
}
template <class C>
__device__ __forceinline__ void p2p_pair(const PoseK<double>& T, const C (&x)[2], const C (&y)[2], const C (&z)[2], const C (&cx)[2],
                                         const C (&cy)[2], const C (&cz)[2], const C (&w)[2], C __attribute__((ext_vector_type(2))) (&s)[17]) {
  typedef C V __attribute__((ext_vector_type(2)));
  C px[2], py[2], pz[2], rx[2], ry[2], rz[2];
#pragma unroll
  for (int e = 0; e < 2; e++) {   // the fp64 part stays per point: transform and the (cancelling) residual
    double pxd, pyd, pzd;
    transform<C>(T, x[e], y[e], z[e], pxd, pyd, pzd);
    px[e] = (C)pxd; py[e] = (C)pyd; pz[e] = (C)pzd;
    rx[e] = (C)(pxd - (double)cx[e]); ry[e] = (C)(pyd - (double)cy[e]); rz[e] = (C)(pzd - (double)cz[e]);
  }
  p2p_accumulate<V>(V{px[0], px[1]}, V{py[0], py[1]}, V{pz[0], pz[1]}, V{rx[0], rx[1]}, V{ry[0], ry[1]}, V{rz[0], rz[1]}, V{w[0], w[1]}, s);
}

// All three kinds on pairs: one group of P correspondences added into the 2-vector partial sums s2 (NOT widened:
// the
// caller decides how many groups share one widening into the fp64 accumulators -- flush_pairs).
// CLEAN: the caller vouches that every value of the arrays is finite (the C-ABI shim launches the CLEAN flavour of a kernel first and
// looks at the record it reads anyway: a NaN or an infinity anywhere in the arrays makes at least one sum non-finite -- 0 x NaN = NaN --
// and the launch is repeated in the guarded flavour, rpe_receive.hip clean-first protocol).  No NaN guards and no selects: a correspondence is
// switched off by its weight alone (0 x finite = 0: the same bits the guarded form adds); entries past the end of the arrays were
// loaded as zeros and get weight 0.  17 % of the guarded group's instructions.
template <class T, int KIND, bool MASK, bool WEIGHT, bool CLEAN, int NS>
__device__ __forceinline__ void pair_group(const PoseK<double>& pose, const T (&vw)[3 * Pk<T>::P], const T (&vb)[3 * Pk<T>::P],
                                           const T (&vc)[3 * Pk<T>::P], const short (&m)[Pk<T>::P], const T (&wv)[Pk<T>::P],
                                           int npresent, T __attribute__((ext_vector_type(2))) (&s2)[NS]) {
  constexpr int P = Pk<T>::P;
  static_assert(NS == (KIND == KIND_P2P ? 17 : 29), "sums per kind");
  static_assert(KIND == KIND_P2P || KIND == KIND_P2PLANE || KIND == KIND_BEARING || KIND == KIND_REPROJ, "residual kinds");
#pragma unroll
  for (int j = 0; j < P / 2; j++) {
    T x[2], y[2], z[2], bx[2], by[2], bz[2], nx[2], ny[2], nz[2], wi[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const int i = 2 * j + e;
      bx[e] = vb[3 * i]; by[e] = vb[3 * i + 1]; bz[e] = vb[3 * i + 2];
      T w = WEIGHT ? wv[i] : T(1);
      if (MASK) w = m[i] == 1 ? w : T(0);
      if (CLEAN) {
        w = i < npresent ? w : T(0);   // (folds away for full groups: npresent is the constant P there)
        x[e] = vw[3 * i]; y[e] = vw[3 * i + 1]; z[e] = vw[3 * i + 2];
        if (KIND == KIND_P2PLANE) { nx[e] = vc[3 * i]; ny[e] = vc[3 * i + 1]; nz[e] = vc[3 * i + 2]; }
      } else {
        w = (i < npresent && !all_nan(bx[e], by[e], bz[e])) ? w : T(0);
        const bool off = w == T(0);
        // keeps NaN / inf of skipped columns out of the sums (selects, not branches); bearing: p = t + R (0, 0, 1) != 0 keeps 1 / |p|
        // finite
        x[e] = off ? T(0) : vw[3 * i]; y[e] = off ? T(0) : vw[3 * i + 1];
        z[e] = off ? ((KIND == KIND_BEARING || KIND == KIND_REPROJ) ? T(1) : T(0)) : vw[3 * i + 2];
        bx[e] = off ? T(0) : bx[e]; by[e] = off ? T(0) : by[e]; bz[e] = off ? T(1) : bz[e];
        if (KIND == KIND_P2PLANE) { nx[e] = off ? T(0) : vc[3 * i]; ny[e] = off ? T(0) : vc[3 * i + 1];
            nz[e] = off ? T(0) : vc[3 * i + 2]; }
      }
      wi[e] = w;
    }
    if constexpr (KIND == KIND_P2P) p2p_pair<T>(pose, x, y, z, bx, by, bz, wi, s2);
    // 35 of the ~50 operations per point are the accumulation
    else if constexpr (KIND == KIND_P2PLANE) p2plane_pair<T>(pose, x, y, z, bx, by, bz, nx, ny, nz, wi, s2);
    // two rows per point: 70 of ~100
    else if constexpr (KIND == KIND_BEARING) bearing_pair<T>(pose, x, y, z, bx, by, bz, wi, s2);
    else {
      reproj_pair<T>(pose, x, y, z, bx, by, bz, wi, s2);
      // CLEAN promises "a NaN or an infinity anywhere in the arrays makes at least one sum non-finite" (rpe_receive.hip clean-first protocol).
      // reproj_prepare switches a correspondence with p_z or bv_z not above kReprojMinZ off -- which a NaN is -- and hands harmless
      // geometry on, so the inputs are multiplied into the cost slot by hand: 0 x finite = +0 (the sum of squares keeps its bits),
      // 0 x NaN = 0 x inf = NaN
      if (CLEAN) {
        typedef T V __attribute__((ext_vector_type(2)));
        const V all = ((V{x[0], x[1]} + V{y[0], y[1]}) + (V{z[0], z[1]} + V{bx[0], bx[1]})) + (V{by[0], by[1]} + V{bz[0], bz[1]});
        s2[27] = __builtin_elementwise_fma(V{T(0), T(0)}, all, s2[27]);
      }
    }
  }
}
// widen the pair sums into the fp64 accumulators and clear them.  A widening costs three instructions per sum (add the halves,
// convert, fp64 add: 87 per call), so the streaming loops let two groups (8 fp32 correspondences) share one.
template <class T, int NS>
__device__ __forceinline__ void flush_pairs(T __attribute__((ext_vector_type(2))) (&s2)[NS], double (&acc)[NS]) {
  typedef T V __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int k = 0; k < NS; k++) { acc[k] += (double)(s2[k].x + s2[k].y); s2[k] = V{T(0), T(0)}; }
}

// One group of P correspondences into the fp64 accumulators (the resident kernels, the fused ICP kernels, single groups): all three
// kinds on pairs of correspondences (pair_group), widened at the end of the group.  CLEAN: see pair_group.
template <class T, int KIND, bool MASK, bool WEIGHT, int NACC, bool CLEAN = false>
__device__ __forceinline__ void normal_eq_group(const PoseK<double>& pose, const T (&vw)[3 * Pk<T>::P], const T (&vb)[3 * Pk<T>::P],
                                                const T (&vc)[3 * Pk<T>::P], const short (&m)[Pk<T>::P], const T (&wv)[Pk<T>::P],
                                                int npresent, double (&acc)[NACC]) {
  typedef T V __attribute__((ext_vector_type(2)));
  V s2[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) s2[k] = V{T(0), T(0)};
  pair_group<T, KIND, MASK, WEIGHT, CLEAN, NACC>(pose, vw, vb, vc, m, wv, npresent, s2);
  flush_pairs<T, NACC>(s2, acc);
}
// ================================================================================================
// K1 / K2 / K3, RESIDENT form: the host-driven Gauss-Newton loop in ONE launch.
// The north-star loop keeps the 6x6 solve and the SE(3) exp-map on the host, so every iteration needs a host round trip; with one
// launch per iteration that round trip also pays a kernel launch, the dispatch ramp of the grid (1.3 - 2.3 us for 150 workgroups,
// profiles/r02_tail_timeline.jsonl) and a re-read of the arrays.  Here the grid stays resident between iterations: every workgroup
// waits for the next pose in a control block that lives in fine-grained DEVICE memory and that the host writes through the PCIe BAR
// (MI355X: 1.9 us host -> 256 polling workgroups -> host, scripts/ubench/hostmailbox.hip; polling pinned HOST memory from 150
// workgroups costs 13 us), evaluates its slice, and the last workgroup publishes the record exactly as normal_eq_kernel does.
// Frame-sized problems (one group per thread) keep their correspondences IN REGISTERS across the iterations -- the arrays are read
// from memory once per refinement, not once per iteration.
// Control block: 16 words of 8 bytes = two 64-byte halves, each carrying its own copy of the tag so that no ordering between the
// host's stores to the two halves is assumed:   [0] tag | [1..7] pose[0..6]   ||   [8..12] pose[7..11] | [13,14] - | [15] tag.
// The host writes the pose, then both tags (= first_tag + iteration; bit 63 set = stop).  Co-residency: the grid has at most one
// workgroup per CU (reduce_grid, max_blocks <= 256) and no workgroup waits for another one -- only for the host, and only for a
// bounded time (~2 s of the 100 MHz clock, then the kernel exits without publishing and the host reports an error).
// ================================================================================================
constexpr unsigned long long kResidentStop = 1ull << 63;
// one group of P correspondences through the bounds-checked loaders when it is the ragged last one (g == full), plain 16-byte loads
// otherwise
template <class T, int KIND, bool MASK, bool WEIGHT>
__device__ __forceinline__ void load_any_group(const T* __restrict__ xw, const T* __restrict__ b, const T* __restrict__ c,
                                               const short* __restrict__ mask, const T* __restrict__ weight, int64_t g, int64_t full,
                                               int64_t n, T (&vw)[3 * Pk<T>::P], T (&vb)[3 * Pk<T>::P], T (&vc)[3 * Pk<T>::P],
                                               short (&m)[Pk<T>::P], T (&wv)[Pk<T>::P]) {
  typedef typename Pk<T>::V V;
  if (g < full) {
    const V* xw4 = reinterpret_cast<const V*>(xw);
    const V* b4 = reinterpret_cast<const V*>(b);
    const V x0 = xw4[3 * g], x1 = xw4[3 * g + 1], x2 = xw4[3 * g + 2];
    const V y0 = b4[3 * g], y1 = b4[3 * g + 1], y2 = b4[3 * g + 2];
    unpack3(x0, x1, x2, vw);
    unpack3(y0, y1, y2, vb);
    if (KIND == KIND_P2PLANE) { const V* c4 = reinterpret_cast<const V*>(c);
        const V z0 = c4[3 * g], z1 = c4[3 * g + 1], z2 = c4[3 * g + 2]; unpack3(z0, z1, z2, vc); }
    if (MASK) load_mask_full(mask, g, m);
    if (WEIGHT) load_weight_full(weight, g, wv);
  } else {
    load_group<T>(xw, g, n, vw);
    load_group<T>(b, g, n, vb);
    if (KIND == KIND_P2PLANE) load_group<T>(c, g, n, vc);
    if (MASK) load_scalars<T, short>(mask, g, n, m, (short)0);
    if (WEIGHT) load_scalars<T, T>(weight, g, n, wv, T(0));
  }
}

// ---- the two halves of a RESIDENT iteration that do not depend on what is being summed (shared by the normal-equation and the ICP
// resident kernels).
// Wait for pose number `want` in the control block (16 words in fine-grained device memory that the host writes through the PCIe BAR:
// word 0 = tag, words 1..12 = pose, word 15 = tag again, so the two 64-byte halves may arrive in any order).  The first 16 lanes of
// wave 0 read one 8-byte word each until both tags match.  Returns 1 = go (pose in s_pose), 2 = stop requested, 3 = the host went
// away (2 s); the value is uniform over the workgroup.
constexpr int kAutoMaxRunSums = 1024;   // run records x sums an autonomous iteration reads per workgroup (resident_auto_stage)
// what a poll of the control block says: 0 = not yet, 1 = pose `want` is there, 2 = stop
__device__ __forceinline__ int resident_judge_pose(unsigned long long w, unsigned long long want) {
  const unsigned int lo = (unsigned int)w, hi = (unsigned int)(w >> 32);
  const unsigned long long ta = ((unsigned long long)__builtin_amdgcn_readlane(hi, 0) << 32) | __builtin_amdgcn_readlane(lo, 0);
  const unsigned long long tb = ((unsigned long long)__builtin_amdgcn_readlane(hi, 15) << 32) | __builtin_amdgcn_readlane(lo, 15);
  if (ta != tb) return 0;
  const unsigned long long num = ta & ~kResidentStop;
  if (num == want) return (ta & kResidentStop) ? 2 : 1;
  // Tags only grow within a context, and the host writes tag want + 1 only after it has this workgroup's sums of `want`: a larger
  // tag can only belong to a LATER call -- this launch was stopped and the stop tag has already been overwritten.  Leave.
  return num > want ? 2 : 0;
}
// One wave polls, one load at a time.  Tried and rejected (profiles/r03_poll_depth_ab.jsonl): 2 / 4 / 8 waves polling with staggered
// starts, the first to see the pose handing it to the others through LDS -- 5.87 vs 5.70 us per step (four alternations), no gain: the
// step waits for the LAST of 150 workgroups, and more pollers only add traffic on the control block.
template <int BLK>
__device__ __forceinline__ int resident_wait_pose(const unsigned long long* __restrict__ ctl, unsigned long long want,
                                                  double* __restrict__ s_pose, int* __restrict__ s_go,
                                                  unsigned long long wait_ticks = 200000000ull) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    int go = 0;
    unsigned long long w = 0;
    for (;;) {
      if (lane < 16) w = __hip_atomic_load(ctl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((go = resident_judge_pose(w, want)) != 0) break;
      if (wall_clock64() - t0 > wait_ticks) { go = 3; break; }   // the host went away: give up
      __builtin_amdgcn_s_sleep(2);
    }
    if (lane >= 1 && lane <= 12) s_pose[lane - 1] = __longlong_as_double((long long)w);
    if (lane == 0) *s_go = go;
  }
  __syncthreads();
  return *s_go;
}
// which run a workgroup of a resident grid belongs to, and which row of it: runs of R consecutive workgroups, or (stride S > 1) run r =
// workgroups r, r + S, r + 2 S ... Workgroups go to the XCDs round robin (scripts/ubench/xcd_handoff.hip reads HW_REG_XCC_ID: workgroup
// i runs on XCD i mod 8), so with S = 8 a run is the workgroups of ONE XCD and its granules cross no XCD boundary on their way to the
// collecting workgroup: 0.28 us in flight instead of 0.39 - 0.50 (same file).  Only the latency depends on that placement -- the stores
// and loads are the sc1 ones either way.
struct RunShape { int run, row, leader, rows, runs, step; };
__device__ __forceinline__ RunShape run_shape(const Finish& fin, int G) {
  RunShape s;
  const int b = (int)blockIdx.x;
  if (fin.stride > 1) {
    const int S = fin.stride < G ? fin.stride : G;
    s.run = b % S; s.row = b / S; s.leader = s.run; s.rows = (G - s.run + S - 1) / S; s.runs = S; s.step = S;
  } else {
    const int R = fin.rows;
    s.run = b / R; s.leader = s.run * R; s.row = b - s.leader; s.rows = min(R, G - s.leader); s.runs = (G + R - 1) / R; s.step = 1;
  }
  return s;
}
// Cross-workgroup stage of one resident iteration: COLLECTING workgroups + the host. Workgroups are taken in runs (run_shape above: one
// run per XCD, or runs of R = fin.rows consecutive workgroups); the first of a run collects: the others store their NACC sums as
// 16-byte granules {value, iteration tag} (one sc1 store per lane, no
// drain, no arrival counter) and go back to waiting for the next pose; every thread of the collecting workgroup polls its granule(s)
// (collect_rows: sc1 loads until the tag is this iteration's), the rows are added in a fixed order, and the run's NACC sums go to the
// host as tagged 16-byte pairs.  The host thread that owns the 6x6 solve adds the ceil(G / R) run records in run order.  So one
// hand-off hop on the GPU (about 1 us: a collecting wave reads a few hundred bytes, MI355X_MICROARCH.md "handoff-1to1"), a few hundred
// bytes over PCIe, and sums that are a fixed function of (G, R) whichever workgroup finishes first.  R = 1: every workgroup sends its
// own record (tiny problems).  A workgroup overwrites its granules only in the next iteration, which the host starts after it has
// received every run record, i.e. after the granules have been read.  Returns false if a granule never arrived (the kernel ends without
// publishing; the host reports that).
// ... from a value every thread < NACC already holds (the workgroup's sum number threadIdx.x): the resident scoring kernel's vote counts
template <int NACC, int BLK>
__device__ __forceinline__ bool resident_cross_own(double own, const Finish& fin, unsigned long long tag, unsigned long long seq, bool stamp_it) {
  constexpr int RGN = BLK / NACC;                       // rows a collecting workgroup takes with one granule per thread
  __shared__ double g_part[RGN][NACC];
  unsigned long long* gran = reinterpret_cast<unsigned long long*>(fin.partials);   // [workgroup][NACC] granules of 2 words
  const RunShape rs = run_shape(fin, (int)gridDim.x);
  const int run = rs.run, leader = rs.leader;
  if ((int)blockIdx.x != leader) {
    // test hook: a granule that never comes
    const bool withheld = fin.fault_tag != 0 && tag == fin.fault_tag && blockIdx.x + 1 == gridDim.x;
    if (threadIdx.x < NACC && !withheld) store_granule16(gran + 2 * ((size_t)blockIdx.x * NACC + threadIdx.x), own, tag);
#ifdef RPE_STAMPS
    if (stamp_it) RPE_STAMP(2);
#endif
    return true;
  }
  if (threadIdx.x < NACC) g_part[0][threadIdx.x] = own;
#ifdef RPE_STAMPS
  if (stamp_it) RPE_STAMP(2);
#endif
  const int rows = rs.rows;
  const bool lost = collect_rows<NACC, BLK>(gran, (int)gridDim.x, leader, rows, tag, g_part, rs.step);
#ifdef RPE_STAMPS
  if (stamp_it) RPE_STAMP(3);
#endif
  const bool ok = !__syncthreads_or(lost);   // every granule of the run has arrived (or one never will)
#ifdef RPE_STAMPS
  if (stamp_it) RPE_STAMP(4);
#endif
  // the run's sums by the threads that send them (sum_rows_lane: no second barrier); a run with a missing granule tells the host so (it
  // releases the grid and finishes with one launch per iteration).  g_part is rewritten only behind the next wait's workgroup barrier.
  if (threadIdx.x < NACC) store_tagged_pair(fin.out_host, run * NACC + threadIdx.x,
      ok ? sum_rows_lane<NACC, RGN>(g_part, rows < RGN ? rows : RGN, threadIdx.x) : __longlong_as_double((long long)kLostMarker), seq);
#ifdef RPE_STAMPS
  if (stamp_it) RPE_STAMP(5);
#endif
  return ok;
}
template <int NACC, int BLK>
__device__ __forceinline__ bool resident_cross_stage(const double (&acc)[NACC], const Finish& fin, unsigned long long tag,
                                                     unsigned long long seq, bool stamp_it) {
  constexpr int NW = BLK / 64;
  __shared__ double g_red[NW][NACC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wave_reduce_to<NACC>(acc, g_red[wave], lane);
  __syncthreads();
  double own = 0.0;
  if (threadIdx.x < NACC) {
#pragma unroll
    for (int w = 0; w < NW; w++) own += g_red[w][threadIdx.x];
  }
  return resident_cross_own<NACC, BLK>(own, fin, tag, seq, stamp_it);
}

// Cross-workgroup stage of one AUTONOMOUS resident iteration (rpe_gn_refine_device: solve and exp-map on the GPU, no host in the loop).
// First hop as above -- runs of R workgroups, the first of a run collects its rows' granules -- but the run's NACC sums go to a RUN
// RECORD in device memory (granules again: {value, iteration tag}, one sc1 store per lane) instead of to the host.  Second hop: EVERY
// workgroup reads all ceil(G / R) run records, adds them in run order (the order the host uses: bitwise the host-driven loop's
// record), expands the record, and its first lane solves the 6x6 system and applies the update to the workgroup's own copy of the
// pose in LDS.  All workgroups compute the same bits, so they agree on the next pose and on when to stop without another hop.
// Run records are double-buffered by iteration parity: a collecting workgroup can publish iteration i + 1 while a late workgroup of
// another run still reads iteration i; it cannot reach i + 2 before that workgroup has delivered its granules of i + 1, i.e. after
// it has finished reading i.  Granules need no second buffer: a workgroup writes iteration i + 1's after it has read run records
// that its collector published after reading iteration i's.  Returns 0 = next iteration, 1 = finished (workgroup 0 published pose |
// step | cost | iterations | status | weight sum to the host), 2 = a granule never arrived (2 s).
// Tried and rejected in round 4 (profiles/r04_auto_flat_ab.jsonl): ONE hop -- every workgroup reads all G x NACC granules itself (40 KB per
// workgroup and iteration at 150 x 17) instead of collecting per run: 6.74 against 6.0-6.06 us per iteration at 307 200 correspondences,
// 10.0 against 8.0 at 1 M (three alternations on one box): eighteen times the bytes cross the fabric, and that costs more than the hop.
template <int NACC, int BLK>
__device__ __forceinline__ int resident_auto_stage(const double (&acc)[NACC], const Finish& fin, unsigned long long tag, int it,
                                                   int max_iters, double tol, double* __restrict__ s_pose, bool stamp_it = false) {
  constexpr int NW = BLK / 64;
  constexpr int RGN = BLK / NACC;
  constexpr int MODE = NACC == 17 ? 1 : 0;
  __shared__ double a_red[NW][NACC];
  __shared__ double a_part[RGN][NACC];
  __shared__ double a_runs[kAutoMaxRunSums];
  __shared__ double a_tot[32];
  __shared__ double a_step;
  __shared__ int a_ok;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G = (int)gridDim.x;
  const RunShape rs = run_shape(fin, G);
  const int run = rs.run, leader = rs.leader, runs = rs.runs;
  unsigned long long* gran = reinterpret_cast<unsigned long long*>(fin.partials);          // [workgroup][NACC] granules of 2 words
  unsigned long long* rrec = gran + 2 * ((size_t)G * NACC + (size_t)(it & 1) * runs * NACC);   // [parity][run][NACC]
  wave_reduce_to<NACC>(acc, a_red[wave], lane);
  __syncthreads();
  if (threadIdx.x < NACC) {
    double own = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) own += a_red[w][threadIdx.x];
    if ((int)blockIdx.x != leader) store_granule16(gran + 2 * ((size_t)blockIdx.x * NACC + threadIdx.x), own, tag);
    else a_part[0][threadIdx.x] = own;
  }
#ifdef RPE_STAMPS
  if (stamp_it) RPE_STAMP(2);
#endif
  bool lost = runs * NACC > kAutoMaxRunSums;   // a geometry the launcher never chooses: reported like a lost granule, not overrun
  if (!lost && (int)blockIdx.x == leader) {
    const int rows = rs.rows;
    lost = __syncthreads_or(collect_rows<NACC, BLK>(gran, G, leader, rows, tag, a_part, rs.step));
    if (!lost) {   // uniform over the workgroup
      const double t = sum_rows<NACC, BLK>(a_part, rows < RGN ? rows : RGN);
      // (ONE workgroup: its sums are the record -- no run record through device memory, no poll: 1 us of a 4 us iteration)
      if (threadIdx.x < NACC) { if (G == 1) a_runs[threadIdx.x] = t; else store_granule16(rrec + 2 * ((size_t)run * NACC + threadIdx.x), t, tag); }
    }
  }
#ifdef RPE_STAMPS
  if (stamp_it) RPE_STAMP(3);
#endif
  if (!lost && G > 1) {
    const int total = runs * NACC;
    for (int i = threadIdx.x; i < total; i += BLK) {
      const unsigned long long t0 = wall_clock64();
      granule_t q;
      for (unsigned int spins = 1;; spins++) {
        q = load_granule16(rrec + 2 * (size_t)i);
        if ((((unsigned long long)q.w << 32) | q.z) == tag) break;
        if ((spins & 63u) == 0 && wall_clock64() - t0 > 200000000ull) { lost = true; break; }   // 2 s: a run record never came
      }
      a_runs[i] = __longlong_as_double((long long)(((unsigned long long)q.y << 32) | q.x));
    }
  }
  lost = __syncthreads_or(lost);
#ifdef RPE_STAMPS
  if (stamp_it) RPE_STAMP(4);
#endif
  // Wave 0 alone from here to the new pose (the other waves wait at the barrier below): lane j adds the runs' sums of value j in run
  // order -- the order the host uses -- the totals cross the wave through LDS (no workgroup barrier inside a wave), and lane 0
  // expands them in registers, solves and updates.  One workgroup barrier per iteration instead of four.
  if (threadIdx.x < 64) {
    if (!lost) {
      double t = 0.0;
      if (threadIdx.x < NACC) for (int r = 0; r < runs; r++) t += a_runs[r * NACC + threadIdx.x];
      if (threadIdx.x < 32) a_tot[threadIdx.x] = threadIdx.x < NACC ? t : 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef RPE_STAMPS
    if (stamp_it) RPE_STAMP(5);
#endif
    if (threadIdx.x == 0) {
      double step = 0.0;
      a_ok = !lost && gn_solve_update<MODE>(a_tot, s_pose, &step, fin.pivot_floor) ? 1 : 0;
      a_step = step;
    }
  }
  __syncthreads();
#ifdef RPE_STAMPS
  if (stamp_it) RPE_STAMP(6);
#endif
  const bool ok = a_ok != 0;
  const bool done = !ok || a_step < tol || it >= max_iters;
  if (done && blockIdx.x == 0 && threadIdx.x == 0 && fin.out_host) {
    for (int k = 0; k < 12; k++) __hip_atomic_store(fin.out_host + k, s_pose[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 12, a_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 13, lost ? 0.0 : record_entry<MODE>(a_tot, 27), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 14, (double)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 15, ok ? 0.0 : (lost ? 2.0 : 1.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 16, lost ? 0.0 : record_entry<MODE>(a_tot, 28), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(fin.out_host + 32), fin.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return lost ? 2 : (done ? 1 : 0);
}

// ---- AUTONOMOUS loop with a SOLVING WORKGROUP (round 5).  resident_auto_stage above makes every workgroup a reader of all run records
// and a solver: two cross-workgroup hops (granules -> per-XCD collectors -> run records -> everybody) in front of 150 identical solves,
// 6.0 us per iteration at 307 200 correspondences of which 2.8 us are the hops.  A frame-sized problem leaves a hundred compute units
// idle, so ONE extra workgroup plays the part the host plays in the host-driven loop: the workers store their sums as granules and wait
// for the next pose; the solving workgroup -- a ONE-WORKGROUP KERNEL OF ITS OWN on a second stream (auto_solver_kernel; inside the
// workers' kernel, inlined or as a call, it cost the main path 16-50 spilled registers or a 160-byte stack) -- polls ALL G x NACC
// granules itself (up to four
// in flight per thread: collect_rows), adds them in workgroup order, solves, applies the exp-map and publishes the pose as 14 granules
// {value, tag} that the first wave of every worker polls.  One hop in, one hop out, one solve.
// Pose record: granules 0..11 = pose, 12 = |delta|, 13 = code (0 go on, 1 finished, 2 failed / lost), all tagged first_tag + iteration
// of the iteration that is to USE the pose.  Single-buffered like the granules: the solver writes pose i + 1 after it has read every
// worker's sums of iteration i, i.e. after every worker has read pose i; a worker overwrites its granules of iteration i after it has
// read pose i + 1, which was written after they were read.
constexpr int kSolverPoseGranules = 14;
// The solving workgroup and its workers are two kernels that must run TOGETHER.  Should a platform run them one after the other (two
// streams that share a hardware queue), each waits for the other in vain: the first meetings -- the workers' wait for their first pose
// from the solver, the solver's waits for the sums of iterations 1 and 2 -- are bounded by 0.25 s instead of the 2 s of every other
// wait, the loop is reported lost, and the context finishes the refinement with one launch per iteration and never asks for a solving
// workgroup again.  (0.25 s: far beyond one iteration over any slice that fits a GPU, and beyond a host thread's hiccup between the
// two launches.)
constexpr unsigned long long kSolverMeetTicks = 25000000ull;
__device__ __forceinline__ unsigned long long* solver_pose_area(const Finish& fin, int workers, int nacc) {
  return reinterpret_cast<unsigned long long*>(fin.partials) + 2 * ((size_t)(workers + 8) * nacc);
}
// worker: wait for the pose record tagged `want`; returns the code (0 go on -- pose in s_pose --, 1 / 2 leave, 3 timed out)
template <int BLK>
__device__ __forceinline__ int solver_wait_pose(const unsigned long long* __restrict__ area, unsigned long long want, double* __restrict__ s_pose,
                                                int* __restrict__ s_go, unsigned long long wait_ticks = 200000000ull) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    granule_t q = {};
    int go = -1;
    for (unsigned int spins = 1;; spins++) {
      bool have = true;
      if (lane < kSolverPoseGranules) {
        q = load_granule16(area + 2 * (size_t)lane);
        have = (((unsigned long long)q.w << 32) | q.z) == want;
      }
      if (__builtin_amdgcn_ballot_w64(!have) == 0) { go = 0; break; }
      if ((spins & 63u) == 0 && wall_clock64() - t0 > wait_ticks) { go = 3; break; }   // (2 s by default): the solving workgroup went away
      __builtin_amdgcn_s_sleep(1);
    }
    const double v = __longlong_as_double((long long)(((unsigned long long)q.y << 32) | q.x));
    if (lane < 12) s_pose[lane] = v;
    if (lane == 13) *s_go = go == 3 ? 3 : (int)v;
  }
  __syncthreads();
  return *s_go;
}
// worker: this workgroup's NACC sums of the iteration as granules (every worker, no collecting workgroup)
template <int NACC, int BLK>
__device__ __forceinline__ void solver_send_sums(const double (&acc)[NACC], const Finish& fin, unsigned long long tag) {
  constexpr int NW = BLK / 64;
  __shared__ double w_red[NW][NACC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long* gran = reinterpret_cast<unsigned long long*>(fin.partials);
  wave_reduce_to<NACC>(acc, w_red[wave], lane);
  __syncthreads();
  if (threadIdx.x < NACC) {
    double own = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) own += w_red[w][threadIdx.x];
    store_granule16(gran + 2 * ((size_t)blockIdx.x * NACC + threadIdx.x), own, tag);
  }
  __syncthreads();   // w_red is reused by the next iteration
}
// the solving workgroup's whole life: iterations 1 .. max_iters
template <int NACC, int BLK>
__device__ __forceinline__ void solver_loop(const Finish& fin, int workers, unsigned long long first_tag, int max_iters) {
  constexpr int RGN = BLK / NACC;
  constexpr int MODE = NACC == 17 ? 1 : 0;
  __shared__ double v_part[RGN][NACC];
  __shared__ double v_tot[32];
  __shared__ double v_pose[12];
  __shared__ double v_step;
  __shared__ int v_code;
  unsigned long long* gran = reinterpret_cast<unsigned long long*>(fin.partials);
  unsigned long long* area = solver_pose_area(fin, workers, NACC);
#ifdef RPE_SOLVER_DEBUG
  const unsigned long long dbg_start = wall_clock64();
#endif
  if (threadIdx.x < 12) v_pose[threadIdx.x] = fin.gn_pose[threadIdx.x];
  const double tol = fin.gn->tol;
  __syncthreads();
  for (int it = 1; it <= max_iters; it++) {
    const unsigned long long tag = first_tag + (unsigned long long)it;
    if (threadIdx.x < NACC) v_part[0][threadIdx.x] = 0.0;   // "row 0" of collect_rows is the collector's own record: none here
    __syncthreads();
    // (the first two iterations wait kSolverMeetTicks only: a platform that runs the two kernels one after the other shows there)
    bool lost = collect_rows<NACC, BLK, 12>(gran, workers, -1, workers + 1, tag, v_part, 1, it <= 2 ? kSolverMeetTicks : 200000000ull);   // rows 1 .. workers = workgroups 0 .. workers - 1 (up to twelve granules in flight per thread: ONE sweep for a frame-sized grid of either record size)
    lost = __syncthreads_or(lost);
    if (lost && threadIdx.x == 0 && fin.out_host) {   // diagnostics for the host's error text: which workers' sums are missing
      int missing = 0, lo = workers, hi = -1;
      for (int w = 0; w < workers; w++) {
        const granule_t q = load_granule16(gran + 2 * ((size_t)w * NACC + NACC - 1));
        if ((((unsigned long long)q.w << 32) | q.z) != tag) { missing++; lo = w < lo ? w : lo; hi = w; }
      }
#ifdef RPE_SOLVER_DEBUG   // (diagnostic build: when did the workers start, relative to this workgroup?  profiles/r05_solver_room.txt)
      {
        unsigned long long mn = ~0ull, mx = 0; int started = 0;
        for (int w = 0; w < workers; w++) {
          const unsigned long long v = __hip_atomic_load(area + 32 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v > dbg_start - 100000000ull) { started++; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        }
        __hip_atomic_store(fin.out_host + 20, (double)started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin.out_host + 21, started ? ((double)mn - (double)dbg_start) * 0.01 : -1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin.out_host + 22, started ? ((double)mx - (double)dbg_start) * 0.01 : -1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin.out_host + 23, ((double)wall_clock64() - (double)dbg_start) * 0.01, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
#endif
      __hip_atomic_store(fin.out_host + 17, (double)missing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(fin.out_host + 18, (double)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(fin.out_host + 19, (double)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const double t = sum_rows<NACC, BLK>(v_part, workers + 1 < RGN ? workers + 1 : RGN);
    if (threadIdx.x < 64) {
      if (threadIdx.x < 32) v_tot[threadIdx.x] = threadIdx.x < NACC ? t : 0.0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (threadIdx.x == 0) {
        double step = 0.0;
        const bool ok = !lost && gn_solve_update<MODE>(v_tot, v_pose, &step, fin.pivot_floor);
        v_step = step;
        v_code = !ok ? 2 : ((step < tol || it >= max_iters) ? 1 : 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int lane = threadIdx.x;
      // finished: the result to the host FIRST (as resident_auto_stage publishes it), the pose record -- on whose code the workers leave
      // -- after it: the host takes "the workers' kernel has ended" for "the result is there"
      if (lane == 0 && v_code != 0 && fin.out_host) {
        const bool ok = v_code == 1;
        for (int k = 0; k < 12; k++) __hip_atomic_store(fin.out_host + k, v_pose[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin.out_host + 12, v_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin.out_host + 13, lost ? 0.0 : record_entry<MODE>(v_tot, 27), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin.out_host + 14, (double)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin.out_host + 15, ok ? 0.0 : (lost ? 2.0 : 1.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin.out_host + 16, lost ? 0.0 : record_entry<MODE>(v_tot, 28), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(fin.out_host + 32), fin.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < kSolverPoseGranules) {
        const double v = lane < 12 ? v_pose[lane] : (lane == 12 ? v_step : (double)v_code);
        store_granule16(area + 2 * (size_t)lane, v, tag + 1);
      }
    }
    __syncthreads();
    if (v_code != 0) return;
  }
}
// Grids for which the solving workgroup pays: enough workers that the two hops of resident_auto_stage hurt, and one compute unit to spare
// (the two kernels must be on the compute units together: the workers' grid is capped at one workgroup per compute unit)
static inline bool auto_solver_grid(int workers, int cap) {
  static const int env = getenv("RPE_AUTO_SOLVER") ? atoi(getenv("RPE_AUTO_SOLVER")) : 1;
  static const int env_min = getenv("RPE_AUTO_SOLVER_MIN") ? atoi(getenv("RPE_AUTO_SOLVER_MIN")) : 32;   // (measured: no gain at 10 workgroups, 5-9 % at 150, 25 % at 255)
  return env != 0 && workers >= env_min && workers + 1 <= cap;
}

// workgroup size of the resident kernels.  256-thread workgroups (two per CU) were measured and lose here -- 6.95-7.3 vs 6.0-6.7 us
// per step at 307 200 points, 46 vs 39 us at 10 M -- although they win for the one-launch kernels: twice the workgroups poll the
// control block and twice the granules cross the hop every iteration.  The largest grid: resident_cap_device() (rpe_kernels.h).
static inline int resident_block() { return 512; }
// ... of which the in-register instances of the normal-equation kernel run 512 / kResidentGroupsPerThread threads, each holding that
// many groups of its workgroup's 512-group slice (one wave per SIMD: rpe_normal_eq.hip normal_eq_resident_kernel)
constexpr int kResidentGroupsPerThread = 2;

}  // namespace rpe
