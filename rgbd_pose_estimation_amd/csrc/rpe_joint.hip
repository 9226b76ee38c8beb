// K1+K2+K3 fused: joint Gauss-Newton normal equations of up to three residual kinds in one pass -- one launch per call as a ROTATING
// one-register-set software pipeline, and the RESIDENT form for frame-sized problems (slice staged in LDS).
#include "rpe_residuals.hpp"

namespace rpe {

// ================================================================================================
// Joint Gauss-Newton normal equations of up to three residual kinds in ONE pass over the arrays: one 3D-3D kind (point-to-point or
// point-to-plane), one 2D-3D kind (bearing or pixel reprojection), normal-normal -- each with its modality's inlier mask,
// per-correspondence weight, a scale and an optional robust (IRLS) weight.  This is the single-kernel form of the objective
// nl_shinji_kneip_ls alternates over (M33 + sigma (M23 + MNN), AbsoluteOrientationNormal.hpp:484-510).
// Record: H upper triangle (21) | g (6) | sum scale w r^2 | sum w.   Up to 60 B/corr + masks/weights.
//
// Arithmetic: as the single-kind kernels (rpe_residuals.hpp) -- p = R Xw + t and everything that cancels in fp64, the products as
// packed fp32 instructions for fp32 arrays (slot-packed: RowSums / StructSums below), each term's sums widened into the fp64
// accumulators once per group.  p is formed once per correspondence and serves every term.
//
// Pipeline (the form of K5, rpe_nl.hip): ONE register set holds a group's arrays.  The terms of a group run one after the other, and
// the moment a term has consumed its arrays the next group's loads of exactly those arrays are issued into the same registers: the
// world points right after the transform, Xc after the 3D-3D term, the bearings after the 2D-3D term, the normals after the
// normal-normal term.  So 6-15 vector loads are in flight during every term's arithmetic, in the registers the group would occupy anyway
// -- a second register set does not fit beside the accumulators in the 256 registers a wave gets (the two-set form tried at the end of
// round 4 spilled 23-107 registers in every multi-term resident instance).
// ================================================================================================
enum { TERM_P2P = 1, TERM_P2PLANE = 2, TERM_BEARING = 4, TERM_NORMAL = 8, TERM_REPROJ = 16 };   // 1 << residual kind
struct JointParams { double scale[5]; int robust[5]; double robust_k[5]; };  // indexed by residual kind 0..4
// ... as the kernels take them: in the array dtype (wave-uniform: scalar registers, no conversions left to hoist into vector registers)
template <class T> struct JointK { T scale[5]; T robust_k[5]; int robust[5]; };
template <class T> static JointK<T> joint_k(const JointParams& p) {
  JointK<T> k;
  for (int i = 0; i < 5; i++) { k.scale[i] = (T)p.scale[i]; k.robust_k[i] = (T)p.robust_k[i]; k.robust[i] = p.robust[i]; }
  return k;
}

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
// ---- per-thread sums of a group in the array dtype, SLOT-PACKED: a 2-vector holds two neighbouring entries of the record for ONE
// correspondence (not one entry for two correspondences, as the single-kind kernels' pair sums do).  Same number of packed fp32
// instructions per product -- the multiplier w J_a is broadcast to both halves by the instruction's operand selectors, no move -- but
// 32 registers of sums instead of 58, which is what lets one group of up to five arrays, the fp64 points and the fp64 accumulators
// share the 256 registers of a wave without spilling.
// Columns are kept in the order and sign the Jacobian rows come out of the arithmetic in:   0 tx | 1 ty | 2 rx | 3 -ry | 4 tz | 5 rz
// (a row is [a ; p x a]; (p x a).x and -(p x a).y are ONE packed multiply-add pair, tz / rz sit together as the leftovers), the 6x6 upper
// triangle by ALIGNED column pairs:
//   h[0..2] = row 0 cols (0,1)(2,3)(4,5) | h[3..5] = row 1 (the (1,0) half is a duplicate of (0,1): never read) | h[6,7] = row 2 cols
//   (2,3)(4,5) | h[8,9] = row 3 (half of h[8] unused) | h[10] = row 4 (4,5) | h[11] = row 5 (half unused) | g[0..2] | cost | weight
// and flush() puts every entry where (and with the sign) the packed record has it.  One widening per group for all terms together.
template <class T> struct TermSums {
  typedef T V __attribute__((ext_vector_type(2)));
  V h[12], g[3];
  T cost, weight;
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int k = 0; k < 12; k++) h[k] = V{T(0), T(0)};
#pragma unroll
    for (int k = 0; k < 3; k++) g[k] = V{T(0), T(0)};
    cost = weight = T(0);
  }
  // one Jacobian row J = {(tx, ty), (rx, -ry), (tz, rz)}: 3 packed multiplies + 15 packed multiply-adds + 2 scalar instructions
  __device__ __forceinline__ void add_row(const V (&J)[3], T r, T w) {
    const V wJ[3] = {V(w) * J[0], V(w) * J[1], V(w) * J[2]};
    h[0] = __builtin_elementwise_fma(V(wJ[0].x), J[0], h[0]); h[1] = __builtin_elementwise_fma(V(wJ[0].x), J[1], h[1]);
    h[2] = __builtin_elementwise_fma(V(wJ[0].x), J[2], h[2]);
    h[3] = __builtin_elementwise_fma(V(wJ[0].y), J[0], h[3]); h[4] = __builtin_elementwise_fma(V(wJ[0].y), J[1], h[4]);
    h[5] = __builtin_elementwise_fma(V(wJ[0].y), J[2], h[5]);
    h[6] = __builtin_elementwise_fma(V(wJ[1].x), J[1], h[6]); h[7] = __builtin_elementwise_fma(V(wJ[1].x), J[2], h[7]);
    h[8] = __builtin_elementwise_fma(V(wJ[1].y), J[1], h[8]); h[9] = __builtin_elementwise_fma(V(wJ[1].y), J[2], h[9]);
    h[10] = __builtin_elementwise_fma(V(wJ[2].x), J[2], h[10]);
    h[11] = __builtin_elementwise_fma(V(wJ[2].y), J[2], h[11]);
    g[0] = __builtin_elementwise_fma(V(r), wJ[0], g[0]); g[1] = __builtin_elementwise_fma(V(r), wJ[1], g[1]);
    g[2] = __builtin_elementwise_fma(V(r), wJ[2], g[2]);
    cost = fma(w * r, r, cost);
  }
  // the row [a ; p x a] of a residual a . (p - ...):  (p x a).x = p_y a_z - p_z a_y ,  -(p x a).y = p_x a_z - p_z a_x
  static __device__ __forceinline__ void row_of(V axy, T az, V pxy, T pz, V (&J)[3]) {
    J[0] = axy;
    J[1] = __builtin_elementwise_fma(V{pxy.y, pxy.x}, V(az), -(V(pz) * V{axy.y, axy.x}));
    J[2] = V{az, fma(pxy.x, axy.y, -(pxy.y * axy.x))};
  }
  // widen into the packed fp64 record: H upper triangle (21) | g (6) | cost | weight
  __device__ __forceinline__ void flush(double (&acc)[29]) const {
    acc[0] += (double)h[0].x; acc[1] += (double)h[0].y; acc[3] += (double)h[1].x; acc[4] -= (double)h[1].y;
    acc[2] += (double)h[2].x; acc[5] += (double)h[2].y;
    acc[6] += (double)h[3].y; acc[8] += (double)h[4].x; acc[9] -= (double)h[4].y; acc[7] += (double)h[5].x; acc[10] += (double)h[5].y;
    acc[15] += (double)h[6].x; acc[16] -= (double)h[6].y; acc[12] += (double)h[7].x; acc[17] += (double)h[7].y;
    acc[18] += (double)h[8].y; acc[13] -= (double)h[9].x; acc[19] -= (double)h[9].y;
    acc[11] += (double)h[10].x; acc[14] += (double)h[10].y; acc[20] += (double)h[11].y;
    acc[21] += (double)g[0].x; acc[22] += (double)g[0].y; acc[24] += (double)g[1].x; acc[25] -= (double)g[1].y;
    acc[23] += (double)g[2].x; acc[26] += (double)g[2].y;
    acc[27] += (double)cost; acc[28] += (double)weight;
  }
};
// Structured sums of the terms whose Jacobian is [I | -[p]x] (point-to-point) or [0 | -[q]x] (normal-normal): weight, w p, the second
// moments w p p^T, w r, w p x r, w r^2 -- 20 packed / scalar instructions per correspondence instead of three generic rows -- expanded
// into the group's TermSums (in the array dtype) when the term is done.
template <class T> struct StructSums {
  typedef T V __attribute__((ext_vector_type(2)));
  V wp_xy, m_xx_xy, m_yx_yy, m_xz_yz, wr_xy, c_x_ny, cost_xy;   // c_x_ny = (sum (p x wr).x , - sum (p x wr).y)
  T wp_z, m_zz, wr_z, c_z, cost_z, n, weight;
  __device__ __forceinline__ void clear() {
    wp_xy = m_xx_xy = m_yx_yy = m_xz_yz = wr_xy = c_x_ny = cost_xy = V{T(0), T(0)};
    wp_z = m_zz = wr_z = c_z = cost_z = n = weight = T(0);
  }
  // TRANSLATION: the term has the identity block (point-to-point); without it (normal-normal) only the rotation block is summed
  template <bool TRANSLATION>
  __device__ __forceinline__ void add(V pxy, T pz, V rxy, T rz, T w, T w_unscaled) {
    const V wpxy = V(w) * pxy;
    const T wpz = w * pz;
    m_xx_xy = __builtin_elementwise_fma(V(wpxy.x), pxy, m_xx_xy);
    m_yx_yy = __builtin_elementwise_fma(V(wpxy.y), pxy, m_yx_yy);
    m_xz_yz = __builtin_elementwise_fma(wpxy, V(pz), m_xz_yz);
    m_zz = fma(wpz, pz, m_zz);
    const V wrxy = V(w) * rxy;
    const T wrz = w * rz;
    // (p x wr).x = p_y wr_z - p_z wr_y ,  -(p x wr).y = p_x wr_z - p_z wr_x :  (p_y, p_x) wr_z - p_z (wr_y, wr_x)
    c_x_ny = __builtin_elementwise_fma(V{pxy.y, pxy.x}, V(wrz), c_x_ny);
    c_x_ny = __builtin_elementwise_fma(V(-pz), V{wrxy.y, wrxy.x}, c_x_ny);
    c_z = fma(pxy.x, wrxy.y, c_z); c_z = fma(-pxy.y, wrxy.x, c_z);
    cost_xy = __builtin_elementwise_fma(wrxy, rxy, cost_xy);
    cost_z = fma(wrz, rz, cost_z);
    weight += w_unscaled;
    if (TRANSLATION) { wp_xy += wpxy; wp_z += wpz; wr_xy += wrxy; wr_z += wrz; n += w; }
  }
  // point-to-point: the group's TermSums START as the expansion of these sums (H_tt = n I, H_tr = -w [p]x, H_rr = w (|p|^2 I - p p^T),
  // g = (w r ; p x w r)), in TermSums' column order and signs
  __device__ __forceinline__ void expand_into(TermSums<T>& t) const {
    const T xx = m_xx_xy.x, xy = m_xx_xy.y, yy = m_yx_yy.y, xz = m_xz_yz.x, yz = m_xz_yz.y, zz = m_zz;
    const T sx = wp_xy.x, sy = wp_xy.y, sz = wp_z;
    t.h[0] = V{n, T(0)}; t.h[1] = V{T(0), -sz}; t.h[2] = V{T(0), -sy};
    t.h[3] = V{T(0), n}; t.h[4] = V{-sz, T(0)}; t.h[5] = V{T(0), sx};
    t.h[6] = V{yy + zz, xy}; t.h[7] = V{sy, -xz};
    t.h[8] = V{T(0), xx + zz}; t.h[9] = V{sx, yz};
    t.h[10] = V{n, T(0)}; t.h[11] = V{T(0), xx + yy};
    t.g[0] = wr_xy; t.g[1] = c_x_ny; t.g[2] = V{wr_z, c_z};
    t.cost = (cost_xy.x + cost_xy.y) + cost_z; t.weight = weight;
  }
  // normal-normal: the rotation block added to what the group's TermSums hold
  __device__ __forceinline__ void add_rotation_into(TermSums<T>& t) const {
    const T xx = m_xx_xy.x, xy = m_xx_xy.y, yy = m_yx_yy.y, xz = m_xz_yz.x, yz = m_xz_yz.y, zz = m_zz;
    t.h[6] += V{yy + zz, xy}; t.h[7].y -= xz; t.h[8].y += xx + zz; t.h[9].y += yz; t.h[11].y += xx + yy;
    t.g[1] += c_x_ny; t.g[2].y += c_z;
    t.cost += (cost_xy.x + cost_xy.y) + cost_z; t.weight += weight;
  }
};
// CLEAN flavour (rpe_receive.hip clean-first protocol): "a NaN or an infinity anywhere in the arrays makes at least one sum non-finite".
// Where a term's own logic would keep such a value out of the sums (the reprojection validity test), the values are multiplied into the
// cost by hand: 0 x finite = +0 (the sum of squares keeps its bits), 0 x NaN = 0 x inf = NaN.
template <class T> __device__ __forceinline__ void clean_poison(T a, T b, T c, T d, T e, T f, T& cost) {
  cost = fma(T(0), (a + b) + (c + d) + (e + f), cost);
}

// ---- the terms of ONE correspondence.  pd: the transformed point in fp64; use: present and passing the modality's inlier mask; u:
// per-correspondence weight (1 without weights).
template <class T, int TERMS, bool CLEAN>
__device__ __forceinline__ void joint_point33(const JointK<T>& prm, const double (&pd)[3], const T (&c)[3], const T (&nc)[3], bool use, T u,
                                              StructSums<T>& ss, TermSums<T>& rs) {
  typedef T V __attribute__((ext_vector_type(2)));
  const bool on = CLEAN ? use : (use & !all_nan(c[0], c[1], c[2]));
  const T w0 = on ? u : T(0);
  const double dx = pd[0] - (double)c[0], dy = pd[1] - (double)c[1], dz = pd[2] - (double)c[2];
  const bool keep = CLEAN || on;   // (guarded flavour: NaN / inf of a skipped column never reaches the sums)
  const T qx = keep ? (T)pd[0] : T(0), qy = keep ? (T)pd[1] : T(0), qz = keep ? (T)pd[2] : T(0);
  if (TERMS & TERM_P2P) {
    const T rx = keep ? (T)dx : T(0), ry = keep ? (T)dy : T(0), rz = keep ? (T)dz : T(0);
    const T w = w0 * robust_weight<T>(prm.robust[0], prm.robust_k[0], [&]() { return sqrt(rx * rx + ry * ry + rz * rz); });
    ss.template add<true>(V{qx, qy}, qz, V{rx, ry}, rz, prm.scale[0] * w, w);
  }
  if (TERMS & TERM_P2PLANE) {
    const T nx = keep ? nc[0] : T(0), ny = keep ? nc[1] : T(0), nz = keep ? nc[2] : T(0);
    const double rd = fma((double)nc[0], dx, fma((double)nc[1], dy, (double)nc[2] * dz));   // the cancelling part in fp64
    const T r = keep ? (T)rd : T(0);
    const T w = w0 * robust_weight<T>(prm.robust[1], prm.robust_k[1], [&]() { return fabs(r); });
    V J[3];
    TermSums<T>::row_of(V{nx, ny}, nz, V{qx, qy}, qz, J);   // [n ; p x n]
    rs.add_row(J, r, prm.scale[1] * w);
    rs.weight += w;
  }
}

template <class T, int TERMS, bool CLEAN>
__device__ __forceinline__ void joint_point23(const JointK<T>& prm, const double (&pd)[3], const T (&b)[3], bool on, T u, TermSums<T>& rs) {
  typedef T V __attribute__((ext_vector_type(2)));
  const bool keep = CLEAN || on;
  const T w0 = on ? u : T(0);
  // a switched-off correspondence gets the harmless geometry bv = p = (0, 0, 1): everything stays finite
  const T bx = keep ? b[0] : T(0), by = keep ? b[1] : T(0), bz = keep ? b[2] : T(1);
  const double sx = keep ? pd[0] : 0.0, sy = keep ? pd[1] : 0.0, sz = keep ? pd[2] : 1.0;
  if (TERMS & TERM_BEARING) {
    // r = p^ x bv in the tangent basis (e1, e2) of bv: two rows rho_i = e_i . p^, J_i = a_i^T [I | -[p]x], a_i = (e_i - rho_i p^) / |p|
    // (rpe_residuals.hpp, bearing_rows); the two dots in fp64 (they cancel), the rest in the array dtype
    T e1[3], e2[3];
    tangent_basis<T>(bx, by, bz, e1, e2);
    const T d1 = (T)dot_e_p(e1[0], e1[1], e1[2], sx, sy, sz), d2 = (T)dot_e_p(e2[0], e2[1], e2[2], sx, sy, sz);
    const T px = (T)sx, py = (T)sy, pz = (T)sz;
    const V pxy = {px, py};
    const T w = w0 * robust_weight<T>(prm.robust[2], prm.robust_k[2], [&]() { return bearing_residual_norm<T>(sx, sy, sz, bx, by, bz); });
    const T inv = LaneOps<T>::rsqrt(fma(px, px, fma(py, py, fma(pz, pz, T(1e-30f)))));
    const V hxy = pxy * V(inv);
    const T hz = pz * inv;
    const T wb = prm.scale[2] * w * fma(bx, bx, fma(by, by, bz * bz));   // |bv|^2 = 1 to rounding
    const T rho[2] = {d1 * inv, d2 * inv};
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const T (&e)[3] = k == 0 ? e1 : e2;
      const V axy = __builtin_elementwise_fma(V(-rho[k]), hxy, V{e[0], e[1]}) * V(inv);
      const T az = fma(-rho[k], hz, e[2]) * inv;
      V J[3];
      TermSums<T>::row_of(axy, az, pxy, pz, J);
      rs.add_row(J, rho[k], wb);
    }
    rs.weight += w;
  }
  if (TERMS & TERM_REPROJ) {   // the 2D-3D term as a pixel reprojection residual (alternative to the bearing form; same arrays)
    T px, py, pz, n1, n2, bzs;
    T wc = w0 * robust_weight<T>(prm.robust[4], prm.robust_k[4], [&]() { return reproj_residual_norm<T>(sx, sy, sz, bx, by, bz); });
    reproj_prepare<T>(sx, sy, sz, bx, by, bz, wc, px, py, pz, n1, n2, bzs);
    const T ipz = LaneOps<T>::rcp(pz);
    const T inv = ipz * LaneOps<T>::rcp(bzs);           // 1 / (p_z bv_z)
    const T g1 = -(px * ipz) * ipz, g2 = -(py * ipz) * ipz;    // rows a_1 = (ipz, 0, g1), a_2 = (0, ipz, g2) of [a ; p x a]
    const V pxy = {px, py};
    const T ws = prm.scale[4] * wc;
    V J[3];
    TermSums<T>::row_of(V{ipz, T(0)}, g1, pxy, pz, J);
    rs.add_row(J, n1 * inv, ws);
    TermSums<T>::row_of(V{T(0), ipz}, g2, pxy, pz, J);
    rs.add_row(J, n2 * inv, ws);
    rs.weight += wc;
    if (CLEAN) clean_poison<T>(bx, by, bz, (T)sx, (T)sy, (T)sz, rs.cost);
  }
}

template <class T, bool CLEAN>
__device__ __forceinline__ void joint_pointnn(const PoseK<double>& pose, const JointK<T>& prm, const T (&m)[3], const T (&c)[3], bool use,
                                              T u, StructSums<T>& ss) {
  typedef T V __attribute__((ext_vector_type(2)));
  const bool on = CLEAN ? use : (use & !all_nan(c[0], c[1], c[2]));
  const bool keep = CLEAN || on;
  const T w0 = on ? u : T(0);
  // q = R Nw and r = q - Nc in fp64, like p and its residuals: rounding R to fp32 would leave a 6e-8 step floor
  const double mx = m[0], my = m[1], mz = m[2];
  const double qxd = fma(pose.R[0], mx, fma(pose.R[1], my, pose.R[2] * mz));
  const double qyd = fma(pose.R[3], mx, fma(pose.R[4], my, pose.R[5] * mz));
  const double qzd = fma(pose.R[6], mx, fma(pose.R[7], my, pose.R[8] * mz));
  const T qx = keep ? (T)qxd : T(0), qy = keep ? (T)qyd : T(0), qz = keep ? (T)qzd : T(0);
  const T rx = keep ? (T)(qxd - (double)c[0]) : T(0), ry = keep ? (T)(qyd - (double)c[1]) : T(0), rz = keep ? (T)(qzd - (double)c[2]) : T(0);
  const T w = w0 * robust_weight<T>(prm.robust[3], prm.robust_k[3], [&]() { return sqrt(rx * rx + ry * ry + rz * rz); });
  // J = [0 | -[q]x] : only the rotation block:  H_ww += |q|^2 I - q q^T ,  g_w += q x r
  ss.template add<false>(V{qx, qy}, qz, V{rx, ry}, rz, prm.scale[3] * w, w);
}

// P inlier flags (short, 0 / 1) of a group as loaded: two per 32-bit register
template <int P> struct MaskP {
  unsigned int w[P / 2];
  __device__ __forceinline__ bool on(int i) const { return ((w[i >> 1] >> (16 * (i & 1))) & 0xffffu) == 1u; }
};
// ---- the terms of one GROUP of P correspondences from unpacked arrays.  At most two correspondences are in flight at a time
// (scheduling barrier after every second one): all four interleaved for instruction-level parallelism have their temporaries live four
// times over, and the register set has no room for that -- the second wave of the SIMD fills the issue slots instead.
// has_mask / has_weight: wave-uniform -- an absent mask or weight array was "loaded" from a dummy address (so that every trip issues
// the same number of loads and every wait counts exactly) and is ignored here.
#ifndef RPE_JOINT_SEQ_N
#define RPE_JOINT_SEQ_N 2
#endif
#define RPE_JOINT_SEQ(i) do { if (((i) + 1) % RPE_JOINT_SEQ_N == 0) __builtin_amdgcn_sched_barrier(0); } while (0)
template <class T> __device__ __forceinline__ void joint_transform(const PoseK<double>& pose, const T (&vw)[3 * Pk<T>::P], double (&pd)[Pk<T>::P][3]) {
#pragma unroll
  for (int i = 0; i < Pk<T>::P; i++) transform<T>(pose, vw[3 * i], vw[3 * i + 1], vw[3 * i + 2], pd[i][0], pd[i][1], pd[i][2]);
}
// 3D-3D term.  Point-to-point: structured sums, expanded into `rs` (which they START: no clearing); point-to-plane: rows into the
// cleared `rs`.
template <class T, int TERMS, bool CLEAN>
__device__ __forceinline__ void joint_group33(const JointK<T>& prm, const double (&pd)[Pk<T>::P][3], const T (&vc)[3 * Pk<T>::P],
                                              const T (&vnc)[3 * Pk<T>::P], const MaskP<Pk<T>::P>& k33, bool has_mask,
                                              const T (&u33)[Pk<T>::P], bool has_weight, TermSums<T>& rs) {
  StructSums<T> ss;
  if (TERMS & TERM_P2P) ss.clear(); else rs.clear();
#pragma unroll
  for (int i = 0; i < Pk<T>::P; i++) {
    const T c[3] = {vc[3 * i], vc[3 * i + 1], vc[3 * i + 2]};
    const T nc[3] = {vnc[3 * i], vnc[3 * i + 1], vnc[3 * i + 2]};
    joint_point33<T, TERMS, CLEAN>(prm, pd[i], c, nc, k33.on(i) | !has_mask, has_weight ? u33[i] : T(1), ss, rs);
    RPE_JOINT_SEQ(i);
  }
  if (TERMS & TERM_P2P) ss.expand_into(rs);
}
template <class T, int TERMS, bool CLEAN>
__device__ __forceinline__ void joint_group23(const JointK<T>& prm, const double (&pd)[Pk<T>::P][3], const T (&vb)[3 * Pk<T>::P],
                                              const MaskP<Pk<T>::P>& k23, bool has_mask, const T (&u23)[Pk<T>::P], bool has_weight,
                                              TermSums<T>& rs) {
  constexpr int P = Pk<T>::P;
  bool on[P], any = false;
  T chk = T(0);
#pragma unroll
  for (int i = 0; i < P; i++) {
    on[i] = k23.on(i) | !has_mask;
    if (!CLEAN) on[i] = on[i] & !all_nan(vb[3 * i], vb[3 * i + 1], vb[3 * i + 2]);
    any |= on[i];
    if (CLEAN) {
      chk += (vb[3 * i] + vb[3 * i + 1]) + vb[3 * i + 2];
      if ((TERMS & (TERM_P2P | TERM_P2PLANE)) == 0) chk += (T)((pd[i][0] + pd[i][1]) + pd[i][2]);   // (with a 3D-3D term the points have been through it)
    }
  }
  // wave-uniform skip: configs[2] has bearings for 2 000 of 307 200 correspondences -- a wave none of whose lanes holds one pays a
  // ballot instead of the term (inside, everything stays predicated, so the sums do not depend on the branch).  CLEAN flavour: a
  // non-finite value in what the term would read sends the wave THROUGH the term, where it reaches the sums (clean-first protocol).
  if (CLEAN) any |= !__builtin_isfinite(chk);
  if (__builtin_amdgcn_ballot_w64(any) == 0) return;
#pragma unroll
  for (int i = 0; i < P; i++) {
    const T b[3] = {vb[3 * i], vb[3 * i + 1], vb[3 * i + 2]};
    joint_point23<T, TERMS, CLEAN>(prm, pd[i], b, on[i], has_weight ? u23[i] : T(1), rs);
    RPE_JOINT_SEQ(i);
  }
}
template <class T, bool CLEAN>
__device__ __forceinline__ void joint_groupnn(const PoseK<double>& pose, const JointK<T>& prm, const T (&vnw)[3 * Pk<T>::P],
                                              const T (&vnc)[3 * Pk<T>::P], const MaskP<Pk<T>::P>& knn, bool has_mask,
                                              const T (&unn)[Pk<T>::P], bool has_weight, TermSums<T>& rs) {
  StructSums<T> ss;
  ss.clear();
#pragma unroll
  for (int i = 0; i < Pk<T>::P; i++) {
    const T m[3] = {vnw[3 * i], vnw[3 * i + 1], vnw[3 * i + 2]};
    const T c[3] = {vnc[3 * i], vnc[3 * i + 1], vnc[3 * i + 2]};
    joint_pointnn<T, CLEAN>(pose, prm, m, c, knn.on(i) | !has_mask, has_weight ? unn[i] : T(1), ss);
    RPE_JOINT_SEQ(i);
  }
  ss.add_rotation_into(rs);
}

// the arrays, masks and weights of the joint kernels (null = absent) ...
template <class T> struct JointArrays {
  const T *xw, *xc, *bv, *nw, *nc;
  const short *m23, *m33, *mnn;
  const T *w23, *w33, *wnn;
};
template <int TERMS> struct JointNeeds {
  static constexpr bool HAS33 = (TERMS & (TERM_P2P | TERM_P2PLANE)) != 0;
  static constexpr bool HAS23 = (TERMS & (TERM_BEARING | TERM_REPROJ)) != 0;
  static constexpr bool HASNN = (TERMS & TERM_NORMAL) != 0;
  static constexpr bool XW = HAS33 || HAS23;                       // the normal-normal term alone never reads the world points
  static constexpr bool NC_WITH_33 = (TERMS & TERM_P2PLANE) != 0 && !HASNN;   // camera normals: reloaded behind the last term that reads them
  static constexpr bool NC_SHARED = (TERMS & TERM_P2PLANE) != 0 && HASNN;     // ... read by the point-to-plane AND the normal-normal term
  static constexpr bool FIRST_IS_P2P = (TERMS & TERM_P2P) != 0;    // then the group's TermSums start as the expansion of its sums
  // sets with another term beside the normal-normal one: the normal arrays are NOT part of the rotating set -- they are asked for right before their term (of the current
  // group) and waited for there: with the accumulators, the fp64 points, the group's sums and three other arrays in flight the 256
  // registers of a wave have no room to hold them through the two heavier terms
  static constexpr bool LATE_NN = HASNN && (HAS33 || HAS23);
};
// a value every lane holds (read from LDS) moved into scalar registers: the pose of a resident iteration -- 24 vector registers otherwise
__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// Frame-sized RESIDENT problems (every thread owns one group for the whole refinement): the group is staged ONCE into the workgroup's
// LDS -- plane-major, one 16-byte vector per thread and plane, so every ds_read_b128 / ds_write_b128 is conflict free -- and every
// iteration reads it back term by term.  (Kept in registers instead, the group costs 36-78 registers for the whole iteration, and the
// multi-term instances spill; 36 B x 4 x 512 threads = 74 KB ... 123 KB of the compute unit's 160 KB.)  Byte offsets of the planes
// inside the dynamic LDS of a workgroup (an absent mask / weight plane: 0 -- read, and ignored).
struct LdsPlan { int xw, xc, bv, nw, nc, m23, m33, mnn, w23, w33, wnn, bytes; };
// ... and ONE group of them in registers, as the 16-byte vectors they were loaded as: the rotating register set.
template <class T> struct JointSet {
  enum { P = Pk<T>::P };
  typedef typename Pk<T>::V V;
  V w[3], c[3], b[3], nw[3], nc[3];
  MaskP<P> k23, k33, knn;
  T u23[P], u33[P], unn[P];
  // an absent mask / weight array is read at the base of the world points (one line for the whole wave: no traffic to speak of), so
  // that a trip issues the same loads whatever the call uses and the waits between them count exactly; the values are ignored
  static __device__ __forceinline__ void load3(const T* __restrict__ a, int64_t g, V (&v)[3]) {
    const V* __restrict__ q = reinterpret_cast<const V*>(a) + 3 * g;
    v[0] = q[0]; v[1] = q[1]; v[2] = q[2];
  }
  static __device__ __forceinline__ void load_mask(const short* __restrict__ m, const void* dummy, int64_t g, MaskP<P>& v) {
    const char* p = m ? reinterpret_cast<const char*>(m) + g * (2 * P) : reinterpret_cast<const char*>(dummy);
    if constexpr (P == 4) { const uint2 u = *reinterpret_cast<const uint2*>(p); v.w[0] = u.x; v.w[1] = u.y; }
    else v.w[0] = *reinterpret_cast<const unsigned int*>(p);
  }
  static __device__ __forceinline__ void load_weight(const T* __restrict__ w, const void* dummy, int64_t g, T (&v)[P]) {
    const char* p = w ? reinterpret_cast<const char*>(w) + g * 16 : reinterpret_cast<const char*>(dummy);
    const V u = *reinterpret_cast<const V*>(p);
    __builtin_memcpy(v, &u, 16);
  }
  __device__ __forceinline__ void load_xw(const JointArrays<T>& A, int64_t g) { load3(A.xw, g, w); }
  template <int TERMS> __device__ __forceinline__ void load_33(const JointArrays<T>& A, int64_t g) {
    load3(A.xc, g, c);
    if (JointNeeds<TERMS>::NC_WITH_33) load3(A.nc, g, nc);
    load_mask(A.m33, A.xc, g, k33);
    load_weight(A.w33, A.xc, g, u33);
  }
  __device__ __forceinline__ void load_23(const JointArrays<T>& A, int64_t g) {
    load3(A.bv, g, b);
    load_mask(A.m23, A.bv, g, k23);
    load_weight(A.w23, A.bv, g, u23);
  }
  // (with_nc = false: the camera normals are shared with the point-to-plane term and rotate by themselves -- load_nc)
  __device__ __forceinline__ void load_nn(const JointArrays<T>& A, int64_t g, bool with_nc = true) {
    load3(A.nw, g, nw);
    if (with_nc) load3(A.nc, g, nc);
    load_mask(A.mnn, A.nw, g, knn);
    load_weight(A.wnn, A.nw, g, unn);
  }
  __device__ __forceinline__ void load_nc(const JointArrays<T>& A, int64_t g) { load3(A.nc, g, nc); }
  template <int TERMS> __device__ __forceinline__ void load_all(const JointArrays<T>& A, int64_t g) {
    if (JointNeeds<TERMS>::XW) load_xw(A, g);
    if (JointNeeds<TERMS>::HAS33) load_33<TERMS>(A, g);
    if (JointNeeds<TERMS>::HAS23) load_23(A, g);
    if (JointNeeds<TERMS>::NC_SHARED) load_nc(A, g);
    if (JointNeeds<TERMS>::HASNN && !JointNeeds<TERMS>::LATE_NN) load_nn(A, g);
  }
  // ---- the same through the workgroup's LDS (LdsPlan): plane k of an xyz array = vector k of every thread's group
  template <int BLK> static __device__ __forceinline__ void lds_put3(unsigned char* lds, int off, const V (&v)[3]) {
    V* p = reinterpret_cast<V*>(lds + off) + threadIdx.x;
    p[0] = v[0]; p[BLK] = v[1]; p[2 * BLK] = v[2];
  }
  template <int BLK> static __device__ __forceinline__ void lds_get3(const unsigned char* lds, int off, V (&v)[3]) {
    const V* p = reinterpret_cast<const V*>(lds + off) + threadIdx.x;
    v[0] = p[0]; v[1] = p[BLK]; v[2] = p[2 * BLK];
  }
  static __device__ __forceinline__ void lds_put_mask(unsigned char* lds, int off, const MaskP<P>& m) {
    unsigned int* p = reinterpret_cast<unsigned int*>(lds + off) + (P / 2) * threadIdx.x;
#pragma unroll
    for (int k = 0; k < P / 2; k++) p[k] = m.w[k];
  }
  static __device__ __forceinline__ void lds_get_mask(const unsigned char* lds, int off, MaskP<P>& m) {
    const unsigned int* p = reinterpret_cast<const unsigned int*>(lds + off) + (P / 2) * threadIdx.x;
#pragma unroll
    for (int k = 0; k < P / 2; k++) m.w[k] = p[k];
  }
  static __device__ __forceinline__ void lds_put_weight(unsigned char* lds, int off, const T (&u)[P]) {
    V t; __builtin_memcpy(&t, u, 16);
    reinterpret_cast<V*>(lds + off)[threadIdx.x] = t;
  }
  static __device__ __forceinline__ void lds_get_weight(const unsigned char* lds, int off, T (&u)[P]) {
    const V t = reinterpret_cast<const V*>(lds + off)[threadIdx.x];
    __builtin_memcpy(u, &t, 16);
  }
  template <int TERMS, int BLK> __device__ __forceinline__ void stage_all(unsigned char* lds, const LdsPlan& pl, const JointArrays<T>& A) const {
    typedef JointNeeds<TERMS> N;
    if (N::XW) lds_put3<BLK>(lds, pl.xw, w);
    if (N::HAS33) { lds_put3<BLK>(lds, pl.xc, c); if (A.m33) lds_put_mask(lds, pl.m33, k33); if (A.w33) lds_put_weight(lds, pl.w33, u33); }
    if (N::HAS23) { lds_put3<BLK>(lds, pl.bv, b); if (A.m23) lds_put_mask(lds, pl.m23, k23); if (A.w23) lds_put_weight(lds, pl.w23, u23); }
    if (N::HASNN) { lds_put3<BLK>(lds, pl.nw, nw); if (A.mnn) lds_put_mask(lds, pl.mnn, knn); if (A.wnn) lds_put_weight(lds, pl.wnn, unn); }
    if (N::HASNN || (TERMS & TERM_P2PLANE)) lds_put3<BLK>(lds, pl.nc, nc);
  }
  template <int BLK> __device__ __forceinline__ void get_xw(const unsigned char* lds, const LdsPlan& pl) { lds_get3<BLK>(lds, pl.xw, w); }
  template <int TERMS, int BLK> __device__ __forceinline__ void get_33(const unsigned char* lds, const LdsPlan& pl) {
    lds_get3<BLK>(lds, pl.xc, c);
    if (TERMS & TERM_P2PLANE) lds_get3<BLK>(lds, pl.nc, nc);
    lds_get_mask(lds, pl.m33, k33);
    lds_get_weight(lds, pl.w33, u33);
  }
  template <int BLK> __device__ __forceinline__ void get_23(const unsigned char* lds, const LdsPlan& pl) {
    lds_get3<BLK>(lds, pl.bv, b);
    lds_get_mask(lds, pl.m23, k23);
    lds_get_weight(lds, pl.w23, u23);
  }
  template <int BLK> __device__ __forceinline__ void get_nn(const unsigned char* lds, const LdsPlan& pl) {
    lds_get3<BLK>(lds, pl.nw, nw);
    lds_get3<BLK>(lds, pl.nc, nc);
    lds_get_mask(lds, pl.mnn, knn);
    lds_get_weight(lds, pl.wnn, unn);
  }
  // what an iteration asks for before it has a pose: the world points and the first term's arrays
  template <int TERMS, int BLK> __device__ __forceinline__ void get_first(const unsigned char* lds, const LdsPlan& pl) {
    typedef JointNeeds<TERMS> N;
    if (N::XW) get_xw<BLK>(lds, pl);
    if (N::HAS33) get_33<TERMS, BLK>(lds, pl);
    else if (N::HAS23) get_23<BLK>(lds, pl);
    else get_nn<BLK>(lds, pl);
  }
};
template <class V> __device__ __forceinline__ void pin3_here(V (&v)[3]) { pin16_here(v[0]); pin16_here(v[1]); pin16_here(v[2]); }

// ONE TRIP of the rotating pipeline: the full group held in `q` is added into acc, and the loads of group `gl` are issued into the
// same registers, each array right behind the term that consumed it (the last trip of a thread asks for its own group again: a cached
// re-read instead of a branch, so that every trip issues the same loads and the waits count exactly).  The scheduling barriers and the
// volatile pins keep the optimiser from moving a reload above its term (it would need a second register set: spills) or sinking it to
// the next trip's use (load, wait, compute).
template <class T, int TERMS, bool CLEAN>
__device__ __forceinline__ void joint_trip(const PoseK<double>& pose, const JointK<T>& prm, const JointArrays<T>& A, JointSet<T>& q,
                                           int64_t g, int64_t gl, double (&acc)[29]) {
  constexpr int P = Pk<T>::P;
  typedef JointNeeds<TERMS> N;
  double pd[P][3];
  T va[3 * P], vn[3 * P];
  TermSums<T> rs;
  if (!N::HAS33) rs.clear();
  if (N::XW) {
    pin3_here(q.w);
    unpack3(q.w[0], q.w[1], q.w[2], va);
    joint_transform<T>(pose, va, pd);
    __builtin_amdgcn_sched_barrier(0);
    q.load_xw(A, gl);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (N::HAS33) {
    pin3_here(q.c);
    unpack3(q.c[0], q.c[1], q.c[2], va);
    if (TERMS & TERM_P2PLANE) { pin3_here(q.nc); unpack3(q.nc[0], q.nc[1], q.nc[2], vn); }
    joint_group33<T, TERMS, CLEAN>(prm, pd, va, vn, q.k33, A.m33 != nullptr, q.u33, A.w33 != nullptr, rs);
    __builtin_amdgcn_sched_barrier(0);
    q.template load_33<TERMS>(A, gl);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (N::HAS23) {
    pin3_here(q.b);
    unpack3(q.b[0], q.b[1], q.b[2], va);
    joint_group23<T, TERMS, CLEAN>(prm, pd, va, q.k23, A.m23 != nullptr, q.u23, A.w23 != nullptr, rs);
    __builtin_amdgcn_sched_barrier(0);
    q.load_23(A, gl);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (N::HASNN) {
    if (N::LATE_NN) { q.load_nn(A, g, !N::NC_SHARED); __builtin_amdgcn_sched_barrier(0); }
    pin3_here(q.nw); pin3_here(q.nc);
    unpack3(q.nw[0], q.nw[1], q.nw[2], va);
    unpack3(q.nc[0], q.nc[1], q.nc[2], vn);
    joint_groupnn<T, CLEAN>(pose, prm, va, vn, q.knn, A.mnn != nullptr, q.unn, A.wnn != nullptr, rs);
    __builtin_amdgcn_sched_barrier(0);
    if (!N::LATE_NN) q.load_nn(A, gl);
    if (N::NC_SHARED) q.load_nc(A, gl);   // (the next group's, behind the last term that reads this group's)
    __builtin_amdgcn_sched_barrier(0);
  }
  rs.flush(acc);
}

// ONE ITERATION over a group staged in LDS (LdsPlan): on entry the world points and the first term's arrays have been asked for
// (JointSet::get_first, issued before the wait for the pose); every term asks for the NEXT term's arrays before its own arithmetic, so
// an LDS round trip hides behind a term and no array occupies registers longer than one term before its use.
template <class T, int TERMS, int BLK, bool CLEAN>
__device__ __forceinline__ void joint_trip_lds(const PoseK<double>& pose, const JointK<T>& prm, const JointArrays<T>& A, JointSet<T>& q,
                                               const unsigned char* lds, const LdsPlan& pl, double (&acc)[29]) {
  constexpr int P = Pk<T>::P;
  typedef JointNeeds<TERMS> N;
  double pd[P][3];
  T va[3 * P], vn[3 * P];
  TermSums<T> rs;
  if (!N::HAS33) rs.clear();
  if (N::XW) {
    pin3_here(q.w);
    unpack3(q.w[0], q.w[1], q.w[2], va);
    joint_transform<T>(pose, va, pd);
  }
  // (three-term sets ask for a term's arrays right before the term: with 58 accumulator registers, the fp64 points and the group's
  // sums there is no room for a second term's arrays during the 2D-3D term -- two exposed LDS round trips per iteration instead of spills)
  constexpr bool AHEAD = !(N::HAS33 && N::HAS23 && N::HASNN);
  if (N::HAS33) {
    __builtin_amdgcn_sched_barrier(0);
    if (AHEAD) { if (N::HAS23) q.template get_23<BLK>(lds, pl); else if (N::HASNN) q.template get_nn<BLK>(lds, pl); }
    __builtin_amdgcn_sched_barrier(0);
    pin3_here(q.c);
    unpack3(q.c[0], q.c[1], q.c[2], va);
    if (TERMS & TERM_P2PLANE) { pin3_here(q.nc); unpack3(q.nc[0], q.nc[1], q.nc[2], vn); }
    joint_group33<T, TERMS, CLEAN>(prm, pd, va, vn, q.k33, A.m33 != nullptr, q.u33, A.w33 != nullptr, rs);
  }
  if (N::HAS23) {
    __builtin_amdgcn_sched_barrier(0);
    if (AHEAD) { if (N::HASNN) q.template get_nn<BLK>(lds, pl); } else q.template get_23<BLK>(lds, pl);
    __builtin_amdgcn_sched_barrier(0);
    pin3_here(q.b);
    unpack3(q.b[0], q.b[1], q.b[2], va);
    joint_group23<T, TERMS, CLEAN>(prm, pd, va, q.k23, A.m23 != nullptr, q.u23, A.w23 != nullptr, rs);
  }
  if (N::HASNN) {
    __builtin_amdgcn_sched_barrier(0);
    if (!AHEAD) q.template get_nn<BLK>(lds, pl);
    __builtin_amdgcn_sched_barrier(0);
    pin3_here(q.nw); pin3_here(q.nc);
    unpack3(q.nw[0], q.nw[1], q.nw[2], va);
    unpack3(q.nc[0], q.nc[1], q.nc[2], vn);
    joint_groupnn<T, CLEAN>(pose, prm, va, vn, q.knn, A.mnn != nullptr, q.unn, A.wnn != nullptr, rs);
  }
  rs.flush(acc);
}

// the ragged last group (n not a multiple of P): one thread, one correspondence at a time through plain element loads -- a
// correspondence's worth of registers, not a second group beside the rotating set
template <class T, int TERMS, bool CLEAN>
__device__ __forceinline__ void joint_leftover(const PoseK<double>& pose, const JointK<T>& prm, const JointArrays<T>& A, int64_t full,
                                               int64_t n, double (&acc)[29]) {
  typedef JointNeeds<TERMS> N;
  StructSums<T> ss33, ssnn;
  TermSums<T> rs;
  ss33.clear(); ssnn.clear(); rs.clear();
  for (int64_t i = full * Pk<T>::P; i < n; i++) {
    double pd[3] = {0.0, 0.0, 0.0};
    if (N::XW) transform<T>(pose, A.xw[3 * i], A.xw[3 * i + 1], A.xw[3 * i + 2], pd[0], pd[1], pd[2]);
    T nc[3] = {T(0), T(0), T(0)};
    if (N::HASNN || (TERMS & TERM_P2PLANE)) { nc[0] = A.nc[3 * i]; nc[1] = A.nc[3 * i + 1]; nc[2] = A.nc[3 * i + 2]; }
    if (N::HAS33) {
      const T c[3] = {A.xc[3 * i], A.xc[3 * i + 1], A.xc[3 * i + 2]};
      joint_point33<T, TERMS, CLEAN>(prm, pd, c, nc, !A.m33 || A.m33[i] == 1, A.w33 ? A.w33[i] : T(1), ss33, rs);
    }
    if (N::HAS23) {
      const T b[3] = {A.bv[3 * i], A.bv[3 * i + 1], A.bv[3 * i + 2]};
      bool on = !A.m23 || A.m23[i] == 1;
      if (!CLEAN) on = on & !all_nan(b[0], b[1], b[2]);
      joint_point23<T, TERMS, CLEAN>(prm, pd, b, on, A.w23 ? A.w23[i] : T(1), rs);
    }
    if (N::HASNN) {
      const T m[3] = {A.nw[3 * i], A.nw[3 * i + 1], A.nw[3 * i + 2]};
      joint_pointnn<T, CLEAN>(pose, prm, m, nc, !A.mnn || A.mnn[i] == 1, A.wnn ? A.wnn[i] : T(1), ssnn);
    }
  }
  rs.flush(acc);
  if (TERMS & TERM_P2P) { TermSums<T> t; ss33.expand_into(t); t.flush(acc); }
  if (N::HASNN) { TermSums<T> t; t.clear(); ssnn.add_rotation_into(t); t.flush(acc); }
}

// (fp32: two 256-thread workgroups per compute unit, 256 registers per wave; fp64 arrays -- twice the registers per term sum -- one, with
// the whole register file)
template <class T, int TERMS, int BLK, bool CLEAN>
__global__ __launch_bounds__(BLK, sizeof(T) == 4 ? 512 / BLK : 1) void normal_eq_joint_kernel(JointArrays<T> A, int64_t n, PoseK<double> pose, JointK<T> prm,
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
  const int64_t full = n / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
  JointSet<T> q;
  if (g < full) q.template load_all<TERMS>(A, g);
  while (g < full) {
    const int64_t gn = g + stride;
    joint_trip<T, TERMS, CLEAN>(pose, prm, A, q, g, gn < full ? gn : g, acc);
    g = gn;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && full * P < n) joint_leftover<T, TERMS, CLEAN>(pose, prm, A, full, n, acc);
  reduce_and_finish<29, kNeLd, 0, BLK>(acc, fin);
}

// RESIDENT form (rpe_gn_refine_joint on one GPU, rpe_gn_refine_device with several residual kinds): ONE launch for the whole
// refinement, as normal_eq_resident_kernel -- every iteration the workgroups wait for the host's pose in the control block
// (resident_wait_pose), evaluate their slice of the joint objective and hand the 29 sums to the collecting stage
// (resident_cross_stage); the host adds the run records, solves and updates.
// STAGED (frame-sized problems: one group per thread): the group is staged once into the workgroup's LDS (LdsPlan) and read back term
// by term every iteration (joint_trip_lds) -- every term set has this form.  Otherwise (more than one group per thread, or a slice
// beyond the LDS) the slice is streamed from memory every iteration with the rotating trips of the one-launch kernel, the last trip
// asking for the first group again; this form exists for the two-term fp32 sets (the three-term ones spill in it by 2-37 registers:
// those refinements run one launch per iteration -- joint_resident_fits tells the caller).
// AUTO (rpe_gn_refine_device): no host in the loop -- the first pose from HBM, every later one from the SOLVING WORKGROUP that the
// caller launches beside this grid (launch_auto_solver; rpe_residuals.hpp solver_loop).
template <class T, int TERMS, int BLK, bool AUTO, bool CLEAN, bool STAGED>
__global__ __launch_bounds__(BLK) void normal_eq_joint_resident_kernel(JointArrays<T> A, int64_t n, JointK<T> prm, LdsPlan pl,
                                                                       const unsigned long long* __restrict__ ctl,
                                                                       unsigned long long first_tag, int max_iters, Finish fin) {
  constexpr int P = Pk<T>::P;
  extern __shared__ __attribute__((aligned(16))) unsigned char j_lds[];
  __shared__ double s_pose[12];
  __shared__ int s_go;
  const int64_t full = n / P;
  // AUTO: the autonomous loop -- a solving workgroup (auto_solver_kernel, launched beside this grid) plays the host: the workers send
  // their sums as granules and wait for its pose record (rpe_residuals.hpp solver_loop)
  constexpr bool with_solver = AUTO;
  const int workers = (int)gridDim.x;
  const int64_t g0 = (int64_t)blockIdx.x * BLK + threadIdx.x;
  const bool mine = g0 < full;     // (STAGED: the launcher guarantees full <= workers * BLK: one group per thread)
  JointSet<T> q;
  if (mine) {
    q.template load_all<TERMS>(A, g0);
    if (STAGED && JointNeeds<TERMS>::LATE_NN) q.load_nn(A, g0, !JointNeeds<TERMS>::NC_SHARED);   // (not part of the rotating set: asked for here, for the staging)
    if (STAGED) q.template stage_all<TERMS, BLK>(j_lds, pl, A);   // every thread reads back only what it wrote itself: no barrier
  }
  if (AUTO) {
    if (threadIdx.x < 12) s_pose[threadIdx.x] = fin.gn_pose[threadIdx.x];
    __syncthreads();
  }
  for (int it = 1; it <= max_iters; it++) {
    if (STAGED) q.template get_first<TERMS, BLK>(j_lds, pl);   // in flight while the workgroup waits for its pose (a thread without a group reads its own unused slot)
    // stop requested or no host
    if (!AUTO && resident_wait_pose<BLK>(ctl, first_tag + (unsigned long long)it, s_pose, &s_go, fin.pose_wait_ticks) != 1) return;
    if (with_solver && it > 1 && solver_wait_pose<BLK>(solver_pose_area(fin, workers, 29), first_tag + (unsigned long long)it, s_pose, &s_go, it == 2 ? kSolverMeetTicks : 200000000ull) != 0) return;
    PoseK<double> pose;
#pragma unroll
    for (int k = 0; k < 9; k++) pose.R[k] = uniform_f64(s_pose[k]);
#pragma unroll
    for (int k = 0; k < 3; k++) pose.t[k] = uniform_f64(s_pose[9 + k]);
    double acc[29];
#pragma unroll
    for (int k = 0; k < 29; k++) acc[k] = 0.0;
    if (STAGED) { if (mine) joint_trip_lds<T, TERMS, BLK, CLEAN>(pose, prm, A, q, j_lds, pl, acc); }
    else {
      const int64_t stride = (int64_t)workers * BLK;
      for (int64_t g = g0; g < full;) {
        const int64_t gn = g + stride;
        joint_trip<T, TERMS, CLEAN>(pose, prm, A, q, g, gn < full ? gn : g0, acc);   // (the last trip asks for the FIRST group: the next iteration's)
        g = gn;
      }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && full * P < n) joint_leftover<T, TERMS, CLEAN>(pose, prm, A, full, n, acc);
    if (with_solver) { solver_send_sums<29, BLK>(acc, fin, first_tag + (unsigned long long)it); continue; }
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
// The CLEAN flavour exists for fp32 arrays (the dense-depth path) -- fp64 arrays always take the guarded one -- and not for the sets
// in which the point-to-plane and the normal-normal term share the camera normals (its instances are 2-14 registers short).
// joint_has_clean_flavour tells the C-ABI shim, whose clean-first protocol must know which flavour a launch really took.
template <class T, int TERMS> constexpr bool joint_has_clean() { return sizeof(T) == 4 && !JointNeeds<TERMS>::NC_SHARED; }
// The resident form exists for every fp32 term set and for the single-term fp64 sets: the multi-term fp64 instances would spill (term
// sums of 64 registers beside 58 accumulator registers in the 256 a wave of a 512-thread workgroup gets); those refinements run one
// launch per iteration.
// ... and its CLEAN flavour for all of those but the autonomous three-term instances (1-4 registers short: they run guarded)
template <class T, int TERMS, bool AUTO> constexpr bool joint_has_clean_resident() {
  return joint_has_clean<T, TERMS>() && !(AUTO && JointNeeds<TERMS>::HAS33 && JointNeeds<TERMS>::HAS23 && JointNeeds<TERMS>::HASNN);
}
template <class T, int TERMS> constexpr bool joint_has_resident() { return sizeof(T) == 4 || (TERMS & (TERMS - 1)) == 0; }

template <class T, int TERMS>
static void joint_launch(const DeviceArrays& A, int flags, const PoseK<double>& pose, const JointParams& prm64, const ReduceTarget& rt,
                         hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  const JointK<T> prm = joint_k<T>(prm64);
  // 256-thread workgroups, two per compute unit (the 256 registers a wave then gets hold the accumulators, a term's sums, the fp64
  // points and one group of up to five arrays)
  constexpr int BLK = 256;
  const bool um = (flags & F_USE_MASK) != 0, uw = (flags & F_USE_WEIGHT) != 0;
  // ... and the grid is capped at TWO per compute unit (2 x max_blocks): the joint loops are bound by instruction issue (SQ counters
  // at 10 M, point-to-point + bearing: one wave per SIMD spends 74 % of its life issuing vector instructions and the rest waiting for
  // its loads; the second wave fills those waits -- 83.9 -> 76.8 us, three terms 126 -> 109 us; a third changes nothing,
  // profiles/r05_joint_grid_ab.txt).  fp64 arrays keep one (their kernels hold the whole register file).
  const int G = reduce_grid(A.n, Pk<T>::P, sizeof(T) == 4 ? std::min(2 * rt.max_blocks, 4096) : rt.max_blocks, BLK);
  const JointArrays<T> J = joint_arrays<T>(A, um, uw);
  if constexpr (joint_has_clean<T, TERMS>()) {
    if (rt.clean) { RPE_LAUNCH_EV((normal_eq_joint_kernel<T, TERMS, BLK, true>), dim3(G), dim3(BLK), 0, s, e0, e1, J, A.n, pose, prm, make_finish(rt)); return; }
  }
  RPE_LAUNCH_EV((normal_eq_joint_kernel<T, TERMS, BLK, false>), dim3(G), dim3(BLK), 0, s, e0, e1, J, A.n, pose, prm, make_finish(rt));
}
#ifndef RPE_JOINT_TERM_SETS   // (experiments compile a subset: -D'RPE_JOINT_TERM_SETS(X)=X(5)')
#define RPE_JOINT_TERM_SETS(X) X(1) X(2) X(4) X(8) X(5) X(6) X(9) X(10) X(12) X(13) X(14) X(16) X(17) X(18) X(24) X(25) X(26)
#endif
template <class T>
static hipError_t joint_t(const DeviceArrays& A, int terms, int flags, const double* pose12, const double* scale4, const int* robust4,
                          const double* robust_k4, const ReduceTarget& rt, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  const PoseK<double> pose = make_pose<double>(pose12);
  JointParams prm;
  for (int k = 0; k < 5; k++) { prm.scale[k] = scale4[k]; prm.robust[k] = robust4[k]; prm.robust_k[k] = robust_k4[k]; }   // (arrays of 5: kinds 0..4)
  switch (terms) {
#define RPE_JOINT_CASE(M) case M: joint_launch<T, M>(A, flags, pose, prm, rt, s, e0, e1); break;
    RPE_JOINT_TERM_SETS(RPE_JOINT_CASE)
#undef RPE_JOINT_CASE
    default: return hipErrorInvalidValue;  // empty set, or point-to-point together with point-to-plane, or both 2D-3D forms
  }
  return hipGetLastError();
}
hipError_t launch_normal_eq_joint(const DeviceArrays& A, int terms, int flags, const double* pose12, const double* scale4,
                                  const int* robust4, const double* robust_k4, const ReduceTarget& rt, hipStream_t s, hipEvent_t e0,
                                  hipEvent_t e1) {
  return A.dtype ? joint_t<double>(A, terms, flags, pose12, scale4, robust4, robust_k4, rt, s, e0, e1)
                 : joint_t<float>(A, terms, flags, pose12, scale4, robust4, robust_k4, rt, s, e0, e1);
}

// resident form: grid / record geometry as the resident normal-equation kernel's 29-sum kinds (resident_geometry with a non-p2p kind)
template <class T, int TERMS, int BLK>
static LdsPlan joint_lds_plan(const JointArrays<T>& J) {
  typedef JointNeeds<TERMS> N;
  LdsPlan pl;
  int off = 0;
  // (an absent plane: offset 0 -- the kernel reads there and ignores what it reads)
  auto plane3 = [&](bool present) { if (!present) return 0; const int o = off; off += 3 * 16 * BLK; return o; };
  auto plane = [&](bool present, int bytes_per_thread) { if (!present) return 0; const int o = off; off += bytes_per_thread * BLK; return o; };
  pl.xw = plane3(N::XW); pl.xc = plane3(N::HAS33); pl.bv = plane3(N::HAS23); pl.nw = plane3(N::HASNN);
  pl.nc = plane3(N::HASNN || (TERMS & TERM_P2PLANE) != 0);
  pl.w23 = plane(N::HAS23 && J.w23, 16); pl.w33 = plane(N::HAS33 && J.w33, 16); pl.wnn = plane(N::HASNN && J.wnn, 16);
  pl.m23 = plane(N::HAS23 && J.m23, 2 * Pk<T>::P); pl.m33 = plane(N::HAS33 && J.m33, 2 * Pk<T>::P); pl.mnn = plane(N::HASNN && J.mnn, 2 * Pk<T>::P);
  pl.bytes = off;
  return pl;
}
// LDS a workgroup of the resident instance can use for its staged slice (per instance: the static LDS differs between the
// host-driven and the autonomous form); raises the instance's dynamic-LDS limit the first time
// the streaming (not staged) resident form: two-term fp32 sets (the autonomous CLEAN instance of point-to-plane + bearing is 2 registers
// short: it runs guarded)
template <class T, int TERMS> constexpr bool joint_has_stream_resident() {
  typedef JointNeeds<TERMS> N;
  return sizeof(T) == 4 && ((int)N::HAS33 + (int)N::HAS23 + (int)N::HASNN) == 2;
}
template <class T, int TERMS, bool AUTO> constexpr bool joint_has_clean_stream_resident() {
  return joint_has_stream_resident<T, TERMS>() && joint_has_clean<T, TERMS>() && !(AUTO && (TERMS & TERM_P2PLANE) != 0);
}
template <class T, int TERMS, bool AUTO, bool CLEAN> static int joint_resident_lds() {
  static int left[64];
  static bool known[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 0; }
  if (known[dev]) return left[dev];
  const void* kernel = (const void*)normal_eq_joint_resident_kernel<T, TERMS, 512, AUTO, CLEAN, true>;
  hipFuncAttributes fa;
  int total = 0, v = 0;
  if (hipFuncGetAttributes(&fa, kernel) == hipSuccess &&
      hipDeviceGetAttribute(&total, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess) {
    v = std::max(0, total - (int)fa.sharedSizeBytes - 256);
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, v) != hipSuccess) v = std::max(0, std::min(v, 65536 - (int)fa.sharedSizeBytes));
  }
  (void)hipGetLastError();
  left[dev] = v; known[dev] = true;
  return v;
}
// which resident form serves this call: 1 = staged (one group per thread and the workgroup's slice inside the LDS the instance has),
// 2 = streaming (two-term fp32 sets), 0 = none (one launch per iteration)
template <class T, int TERMS, bool AUTO>
static int joint_resident_form(const JointArrays<T>& J, int64_t n, int G, bool clean) {
  if constexpr (joint_has_resident<T, TERMS>()) {
    static const bool env_off = getenv("RPE_JOINT_LDS") && atoi(getenv("RPE_JOINT_LDS")) == 0;   // experiments: never stage
    const LdsPlan pl = joint_lds_plan<T, TERMS, 512>(J);
    bool fits = !env_off && (n / Pk<T>::P) <= (int64_t)G * 512;
    if (fits) {
      if constexpr (joint_has_clean_resident<T, TERMS, AUTO>()) fits = pl.bytes <= (clean ? joint_resident_lds<T, TERMS, AUTO, true>() : joint_resident_lds<T, TERMS, AUTO, false>());
      else fits = pl.bytes <= joint_resident_lds<T, TERMS, AUTO, false>();
    }
    if (fits) return 1;
    if constexpr (joint_has_stream_resident<T, TERMS>()) return 2;
  }
  return 0;
}
template <class T, int TERMS, bool AUTO, bool CLEAN, bool STAGED>
static void joint_resident_launch_k(const JointArrays<T>& J, int64_t n, int G, const JointK<T>& prm, const unsigned long long* ctl,
                                    unsigned long long first_tag, int max_iters, const Finish& fin, hipStream_t s) {
  constexpr int BLK = 512;
  LdsPlan pl = joint_lds_plan<T, TERMS, BLK>(J);
  if (STAGED) (void)joint_resident_lds<T, TERMS, AUTO, CLEAN>();   // (sets the instance's dynamic-LDS limit)
  else pl.bytes = 0;
  hipLaunchKernelGGL((normal_eq_joint_resident_kernel<T, TERMS, BLK, AUTO, CLEAN, STAGED>), dim3(G), dim3(BLK), (size_t)pl.bytes, s, J, n, prm, pl, ctl,
                     first_tag, max_iters, fin);
}
template <class T, int TERMS>
static void joint_resident_launch(const DeviceArrays& A, int flags, const JointParams& prm64, const unsigned long long* ctl,
                                  unsigned long long first_tag, int max_iters, const ReduceTarget& rt, hipStream_t s) {
  const JointK<T> prm = joint_k<T>(prm64);
  constexpr int BLK = 512;
  const bool um = (flags & F_USE_MASK) != 0, uw = (flags & F_USE_WEIGHT) != 0;
  const int cap = std::max(1, resident_cap_device());
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks < cap ? rt.max_blocks : cap, BLK);
  Finish fin = make_finish(rt);
  constexpr int kMaxRows = 4 * (BLK / 29);
  if (fin.rows > kMaxRows) fin.rows = kMaxRows;
  if (fin.rows < 1) fin.rows = 1;
  const JointArrays<T> J = joint_arrays<T>(A, um, uw);
  const bool au = fin.gn != nullptr;
  if constexpr (joint_has_resident<T, TERMS>()) {
    const int form = au ? joint_resident_form<T, TERMS, true>(J, A.n, G, rt.clean) : joint_resident_form<T, TERMS, false>(J, A.n, G, rt.clean);
    // (callers ask joint_resident_fits first; what has no resident form is never launched)
#define RPE_JOINT_RES(AU, C, ST) joint_resident_launch_k<T, TERMS, AU, C, ST>(J, A.n, G, prm, ctl, first_tag, max_iters, fin, s)
    if (form == 1) {
      if constexpr (joint_has_clean_resident<T, TERMS, true>()) { if (rt.clean && au) { RPE_JOINT_RES(true, true, true); return; } }
      if constexpr (joint_has_clean_resident<T, TERMS, false>()) { if (rt.clean && !au) { RPE_JOINT_RES(false, true, true); return; } }
      if (au) RPE_JOINT_RES(true, false, true); else RPE_JOINT_RES(false, false, true);
    } else if (form == 2) {
      if constexpr (joint_has_stream_resident<T, TERMS>()) {
        if constexpr (joint_has_clean_stream_resident<T, TERMS, true>()) { if (rt.clean && au) { RPE_JOINT_RES(true, true, false); return; } }
        if constexpr (joint_has_clean_stream_resident<T, TERMS, false>()) { if (rt.clean && !au) { RPE_JOINT_RES(false, true, false); return; } }
        if (au) RPE_JOINT_RES(true, false, false); else RPE_JOINT_RES(false, false, false);
      }
    }
#undef RPE_JOINT_RES
  }
}
template <class T>
static hipError_t joint_resident_t(const DeviceArrays& A, int terms, int flags, const double* scale4, const int* robust4,
                                   const double* robust_k4, const unsigned long long* ctl, unsigned long long first_tag, int max_iters,
                                   const ReduceTarget& rt, hipStream_t s) {
  JointParams prm;
  for (int k = 0; k < 5; k++) { prm.scale[k] = scale4[k]; prm.robust[k] = robust4[k]; prm.robust_k[k] = robust_k4[k]; }   // (arrays of 5: kinds 0..4)
  switch (terms) {
#define RPE_JOINT_CASE(M) case M: joint_resident_launch<T, M>(A, flags, prm, ctl, first_tag, max_iters, rt, s); break;
    RPE_JOINT_TERM_SETS(RPE_JOINT_CASE)
#undef RPE_JOINT_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
// May the refinement of this term set run on the resident kernel?  One group per thread within the co-residency cap, and the
// workgroup's slice (arrays of the terms, masks / weights in use) inside the LDS the instance has.
template <class T>
static bool joint_fits_t(const DeviceArrays& A, int terms, int flags, int max_blocks, bool autonomous, bool clean) {
  static const bool env_off = getenv("RPE_JOINT_RESIDENT") && atoi(getenv("RPE_JOINT_RESIDENT")) == 0;   // experiments
  if (env_off) return false;
  const int cap = std::max(1, resident_cap_device());
  const int G = reduce_grid(A.n, Pk<T>::P, max_blocks < cap ? max_blocks : cap, 512);
  const JointArrays<T> J = joint_arrays<T>(A, (flags & F_USE_MASK) != 0, (flags & F_USE_WEIGHT) != 0);
  switch (terms) {
#define RPE_JOINT_CASE(M) case M: return (autonomous ? joint_resident_form<T, M, true>(J, A.n, G, clean) : joint_resident_form<T, M, false>(J, A.n, G, clean)) != 0;
    RPE_JOINT_TERM_SETS(RPE_JOINT_CASE)
#undef RPE_JOINT_CASE
  }
  return false;
}
bool joint_has_clean_flavour(int dtype, int terms) {
  if (dtype) return false;
  switch (terms) {
#define RPE_JOINT_CASE(M) case M: return joint_has_clean<float, M>();
    RPE_JOINT_TERM_SETS(RPE_JOINT_CASE)
#undef RPE_JOINT_CASE
  }
  return false;
}
bool joint_resident_fits(const DeviceArrays& A, int terms, int flags, int max_blocks, bool autonomous, bool clean) {
  return A.dtype ? joint_fits_t<double>(A, terms, flags, max_blocks, autonomous, clean) : joint_fits_t<float>(A, terms, flags, max_blocks, autonomous, clean);
}
hipError_t launch_normal_eq_joint_resident(const DeviceArrays& A, int terms, int flags, const double* scale4, const int* robust4,
                                           const double* robust_k4, const unsigned long long* ctl, unsigned long long first_tag, int max_iters,
                                           const ReduceTarget& rt, hipStream_t s) {
  return A.dtype ? joint_resident_t<double>(A, terms, flags, scale4, robust4, robust_k4, ctl, first_tag, max_iters, rt, s)
                 : joint_resident_t<float>(A, terms, flags, scale4, robust4, robust_k4, ctl, first_tag, max_iters, rt, s);
}

void preload_joint() {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, (const void*)normal_eq_joint_kernel<float, TERM_P2P, 256, false>) != hipSuccess) (void)hipGetLastError();
}

}  // namespace rpe
