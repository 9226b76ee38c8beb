// K4: batched RANSAC hypothesis scoring (the eight vote loops) and K4b: the winner's inlier masks.
#include <type_traits>
#include "rpe_residuals.hpp"   // (the resident scoring kernel shares the resident loops' control block and collecting stage)

namespace rpe {

// ================================================================================================
// K4 : batched hypothesis scoring (vote loops V1..V8) and K4b : winner mask
// ================================================================================================
// ---- diagnostic build only (-DRPE_SCORE_STATS, rgbd_pose_estimation_amd/build.py build_score_stats; scripts/score_filter_stats.py):
// how often a wave of the EXACT 2D test falls through its band filter into the reference's square root + three divisions.
// [0] = wave evaluations of a pair, [1] = fall-throughs.  No counter exists in the product build.
#ifdef RPE_SCORE_STATS
static __device__ unsigned long long g_score_stats[4];
#define RPE_SCORE_STAT(k) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_score_stats[k], 1ull); } while (0)
#define RPE_SCORE_STAT_ADD(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_score_stats[k], (unsigned long long)(v)); } while (0)
#else
#define RPE_SCORE_STAT(k) do {} while (0)
#define RPE_SCORE_STAT_ADD(k, v) do {} while (0)
#endif
enum { VOTE_33 = 0, VOTE_23 = 1, VOTE_33_23 = 2, VOTE_NN_23 = 3, VOTE_NN_33 = 4, VOTE_NN_33_23 = 5, VOTE_23_MATRIX = 6 };
template <int KIND> struct VoteMods {
  static constexpr bool m33 = KIND == VOTE_33 || KIND == VOTE_33_23 || KIND == VOTE_NN_33 || KIND == VOTE_NN_33_23;
  static constexpr bool m23 = KIND == VOTE_23 || KIND == VOTE_33_23 || KIND == VOTE_NN_23 || KIND == VOTE_NN_33_23 || KIND == VOTE_23_MATRIX;
  static constexpr bool mnn = KIND == VOTE_NN_23 || KIND == VOTE_NN_33 || KIND == VOTE_NN_33_23;
  static constexpr bool need_xc = m33 || mnn;  // isValid() gates the N-N vote too
};

typedef unsigned long long wave_mask_t;   // one bit per lane of the wave
// one hypothesis in registers (wave-uniform -> SGPRs)
template <class T, bool EXACT> struct Hyp;
template <class T> struct Hyp<T, false> {
  T R[9], t[3];
  enum { STRIDE = 12 };
  __device__ __forceinline__ void load(const T* __restrict__ p, bool) {
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = p[k];
    t[0] = p[9]; t[1] = p[10]; t[2] = p[11];
  }
  __device__ __forceinline__ void rot(T x, T y, T z, T& ox, T& oy, T& oz) const {
    ox = fma(R[0], x, fma(R[1], y, R[2] * z));
    oy = fma(R[3], x, fma(R[4], y, R[5] * z));
    oz = fma(R[6], x, fma(R[7], y, R[8] * z));
  }
  // fast forms: squared distance / squared cosine compares (no sqrt, no divide)
  // (the transformed point is formed exactly as in23 forms it: a kind with both tests transforms once -- the common subexpression --
  // instead of twice; 9 of 33 multiply-adds per correspondence and hypothesis)
  __device__ __forceinline__ bool in33(T x, T y, T z, T cx, T cy, T cz, T thr_sq) const {
    const T px = fma(R[0], x, fma(R[1], y, fma(R[2], z, t[0])));
    const T py = fma(R[3], x, fma(R[4], y, fma(R[5], z, t[1])));
    const T pz = fma(R[6], x, fma(R[7], y, fma(R[8], z, t[2])));
    const T ex = px - cx, ey = py - cy, ez = pz - cz;
    return fma(ex, ex, fma(ey, ey, ez * ez)) < thr_sq;
  }
  __device__ __forceinline__ bool in23(T x, T y, T z, T bx, T by, T bz, T c, bool) const {
    const T px = fma(R[0], x, fma(R[1], y, fma(R[2], z, t[0])));
    const T py = fma(R[3], x, fma(R[4], y, fma(R[5], z, t[1])));
    const T pz = fma(R[6], x, fma(R[7], y, fma(R[8], z, t[2])));
    const T d = fma(px, bx, fma(py, by, pz * bz));
    const T n2 = fma(px, px, fma(py, py, pz * pz));
    // d > c |p|  <=>  d|d| > c|c| |p|^2  (u -> u|u| is strictly increasing): one branch-free form for either sign of c
    return d * fabs(d) > (c * fabs(c)) * n2;
  }
  __device__ __forceinline__ bool innn(T nwx, T nwy, T nwz, T ncx, T ncy, T ncz, T cnl) const {
    T rx, ry, rz;
    rot(nwx, nwy, nwz, rx, ry, rz);
    return fma(ncx, rx, fma(ncy, ry, ncz * rz)) > cnl;
  }
  // The same predicates on a PAIR of correspondences as 2-vectors -- element by element the operations above in the same order (the
  // same bits), written out so that the packed instructions do not depend on the auto-vectoriser's choice of pairs: left to it, the
  // hypothesis loop of the kinds with a 2D test carried 13-17 register moves per hypothesis to re-pair its operands.
  typedef T V2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ V2 fma2(T a, V2 b, V2 c) { return __builtin_elementwise_fma(V2(a), b, c); }
  __device__ __forceinline__ void xform2(V2 x, V2 y, V2 z, V2& px, V2& py, V2& pz) const {
    px = fma2(R[0], x, fma2(R[1], y, fma2(R[2], z, V2(t[0]))));
    py = fma2(R[3], x, fma2(R[4], y, fma2(R[5], z, V2(t[1]))));
    pz = fma2(R[6], x, fma2(R[7], y, fma2(R[8], z, V2(t[2]))));
  }
  static __device__ __forceinline__ void in33_p2(V2 px, V2 py, V2 pz, V2 cx, V2 cy, V2 cz, T thr_sq, bool& a, bool& b) {
    const V2 ex = px - cx, ey = py - cy, ez = pz - cz;
    const V2 s = __builtin_elementwise_fma(ex, ex, __builtin_elementwise_fma(ey, ey, ez * ez));
    a = s.x < thr_sq; b = s.y < thr_sq;
  }
  static __device__ __forceinline__ void in23_p2(V2 px, V2 py, V2 pz, V2 bx, V2 by, V2 bz, T c, bool& a, bool& b) {
    const V2 d = __builtin_elementwise_fma(px, bx, __builtin_elementwise_fma(py, by, pz * bz));
    const V2 n2 = __builtin_elementwise_fma(px, px, __builtin_elementwise_fma(py, py, pz * pz));
    const V2 rhs = V2(c * fabs(c)) * n2;
    a = d.x * fabs(d.x) > rhs.x; b = d.y * fabs(d.y) > rhs.y;
  }
  __device__ __forceinline__ void innn2(V2 nwx, V2 nwy, V2 nwz, V2 ncx, V2 ncy, V2 ncz, T cnl, bool& a, bool& b) const {
    const V2 rx = fma2(R[0], nwx, fma2(R[1], nwy, V2(R[2]) * nwz));
    const V2 ry = fma2(R[3], nwx, fma2(R[4], nwy, V2(R[5]) * nwz));
    const V2 rz = fma2(R[6], nwx, fma2(R[7], nwy, V2(R[8]) * nwz));
    const V2 s = __builtin_elementwise_fma(ncx, rx, __builtin_elementwise_fma(ncy, ry, ncz * rz));
    a = s.x > cnl; b = s.y > cnl;
  }
};
// exact form: the reference's own operation sequence in Tp, no FMA contraction.
//   R*x     = Eigen _transformVector (sophus/so3.hpp:238-240): uv = 2 (u x v); v + w uv + u x uv
// 3D test = |Xc - (R Xw + t)| < thre_3d with norm = sqrt(x^2 + y^2 + z^2) (AbsoluteOrientation.hpp:137-138), evaluated as x^2 + y^2 +
// z^2 < cut
//   2D test = normalize(R Xw + t) . bv > cos_thr, normalisation by division        (:413-418)
//   N-N     = Nc . (R Nw) > cos_nl                                                  (AbsoluteOrientationNormal.hpp:248-249)
template <class T> struct Hyp<T, true> {
  T qw, qx, qy, qz, t[3];
  T M[9];  // toRotationMatrix(), only for the kneip_ransac variant that multiplies by so3().matrix() (P3P.hpp:365)
  enum { STRIDE = 8 };
  __device__ __forceinline__ void load(const T* __restrict__ p, bool need_matrix) {
#pragma clang fp contract(off)
    qw = p[0]; qx = p[1]; qy = p[2]; qz = p[3]; t[0] = p[4]; t[1] = p[5]; t[2] = p[6];
    if (!need_matrix) return;
    const T tx = T(2) * qx, ty = T(2) * qy, tz = T(2) * qz;
    const T twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const T tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    M[0] = T(1) - (tyy + tzz); M[1] = txy - twz; M[2] = txz + twy;
    M[3] = txy + twz; M[4] = T(1) - (txx + tzz); M[5] = tyz - twx;
    M[6] = txz - twy; M[7] = tyz + twx; M[8] = T(1) - (txx + tyy);
  }
  __device__ __forceinline__ void rot(T x, T y, T z, T& ox, T& oy, T& oz) const {
#pragma clang fp contract(off)
    T ux = qy * z - qz * y, uy = qz * x - qx * z, uz = qx * y - qy * x;
    ux = ux + ux; uy = uy + uy; uz = uz + uz;
    const T cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
    ox = (x + qw * ux) + cx; oy = (y + qw * uy) + cy; oz = (z + qw * uz) + cz;
  }
  // `cut` = the smallest value whose correctly rounded square root reaches thre_3d (computed on the host, rpe_host.hpp sqrt_cut):
  // sqrt is monotonic, so  sqrt(s) < thre_3d  <=>  s < cut  for every s -- the reference's test, bit for bit, without the square root
  // (a third of the instructions of this predicate).  s is formed exactly as Eigen's squaredNorm() forms it.
  __device__ __forceinline__ bool in33(T x, T y, T z, T cx, T cy, T cz, T cut) const {
#pragma clang fp contract(off)
    T rx, ry, rz;
    rot(x, y, z, rx, ry, rz);
    const T ex = cx - (rx + t[0]), ey = cy - (ry + t[1]), ez = cz - (rz + t[2]);
    return (ex * ex + ey * ey + ez * ez) < cut;
  }
  // two correspondences at once as 2-vectors (element-wise IEEE operations in the same order as above: the same bits), so that the
  // fp32 forms issue as packed v_pk_mul_f32 / v_pk_add_f32 -- half the instructions of the scalar sequence
  typedef T V2 __attribute__((ext_vector_type(2)));
  __device__ __forceinline__ void rot2(V2 x, V2 y, V2 z, V2& ox, V2& oy, V2& oz) const {
#pragma clang fp contract(off)
    V2 ux = qy * z - qz * y, uy = qz * x - qx * z, uz = qx * y - qy * x;
    ux = ux + ux; uy = uy + uy; uz = uz + uz;
    const V2 cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
    ox = (x + qw * ux) + cx; oy = (y + qw * uy) + cy; oz = (z + qw * uz) + cz;
  }
  // so3().matrix() * x (kneip_ransac, P3P.hpp:365)
  __device__ __forceinline__ void rotm2(V2 x, V2 y, V2 z, V2& ox, V2& oy, V2& oz) const {
#pragma clang fp contract(off)
    ox = M[0] * x + M[1] * y + M[2] * z; oy = M[3] * x + M[4] * y + M[5] * z; oz = M[6] * x + M[7] * y + M[8] * z;
  }
  // the 3D and 2D tests of a pair, given the pair's rotated points (one rotation serves both tests, as in the scalar code after CSE)
  __device__ __forceinline__ void in33_rot_x2(V2 rx, V2 ry, V2 rz, V2 cx, V2 cy, V2 cz, T cut, bool& a, bool& b) const {
#pragma clang fp contract(off)
    const V2 ex = cx - (rx + t[0]), ey = cy - (ry + t[1]), ez = cz - (rz + t[2]);
    const V2 ss = ex * ex + ey * ey + ez * ez;
    a = ss.x < cut; b = ss.y < cut;
  }
  // The reference's 2D test -- normalise by three IEEE divisions behind a square root, dot, compare (AbsoluteOrientation.hpp:413-418)
  // -- costs five times the 3D test.  A cheap estimate DECIDES it outside a band around the threshold; inside the band (and for |p|^2
  // outside the normal range, NaN, infinity, non-unit bearings, thresholds <= 0) the reference's own operation sequence runs.  The band
  // (12 u, u = unit roundoff of Tp; 24 u until round 5) follows from what the two computations SHARE: both start from the same
  // N = fl(|p|^2) -- the same three products and two sums, the same bits -- so its roundings cancel; with s = sqrt(N), a_i = p_i bv_i / s,
  // c' = sum a_i and A = sum |a_i| <= (1 + 2u) |bv| <= 1.0006 (bearings within 1e-3 of unit length, checked per lane):
  //   reference  L = fl(s) (1 rounding), q_i = fl(p_i / L) (1), t_i = fl(q_i bv_i) (1), d = fl(fl(t_x + t_y) + t_z) (2 on x and y, 1 on z):
  //              every a_i carries at most 5 roundings  ->  |d - c'| <= 5.02 u A
  //   estimate   D = fma(p_x, bv_x, fma(p_y, bv_y, fl(p_z bv_z)))  (at most 3 roundings per term: |D - p.bv| <= 3.01 u |p| |bv|, i.e.
  //              |D / s - c'| <= 3.02 u A), compared in SQUARES -- no reciprocal square root (a quarter-rate instruction per
  //              element; in fp64 an fp32 estimate and two Newton steps), no product with it:
  //                  fl(D |D|) > fl(N hi2),  hi2 = fl(hi hi),  hi = fl(c + band)   =>   D / s > (c + band) (1 - 2.5 u)
  //              (one rounding each in D |D|, N hi2, hi2 and hi: the square root halves the first three), and with lo = fl(c - band)
  //              the mirror image; D |D| carries D's sign, so a negative D is "below" without a comparison of its own
  // so D / s > c + band - 2.5 u implies d > c + band - 10.6 u: outside a band of 12 u the estimate decides the reference's comparison.
  // The branch is wave-uniform (one ballot): a wave none of whose lanes is within the band never divides.  Votes stay the reference's,
  // bit for bit (tests/test_gpu_kernels.py test_score_exact_votes_bit_identical, the on-threshold cases of
  // tests/test_gpu_score_filter.py, the fuzz campaign).
  static constexpr T kBand23 = T(12) * (sizeof(T) == 4 ? T(5.9604644775390625e-08) : T(1.1102230246251565e-16));   // 12 u
  // the estimate of a pair: in = "above the band", sure = "outside the band" with |p|^2 in the range the bound covers (1e-30 .. 1e30
  // metres^2: no underflow in the squares, no overflow), per element
  struct Est23 { bool in0, in1, sure0, sure1; };
  static __device__ __forceinline__ Est23 estimate23(V2 px, V2 py, V2 pz, V2 n2, V2 bx, V2 by, V2 bz, T c) {
    const V2 dt = dot_fma2(px, py, pz, bx, by, bz);
    const T hi = c + kBand23, lo = c - kBand23;
    // (wave-uniform, hypothesis-independent: formed once.  A threshold within the band of zero: nothing is above +inf |p|^2 or below
    // -inf |p|^2 -- nothing is sure)
    const T hi2 = lo > T(0) ? hi * hi : __builtin_inff(), lo2 = lo > T(0) ? lo * lo : -__builtin_inff();
    const V2 d2 = {dt.x * fabs(dt.x), dt.y * fabs(dt.y)};
    const V2 h = n2 * hi2, l = n2 * lo2;
    const T tiny = T(1e-30), huge = T(1e30);
    Est23 e;
    e.in0 = d2.x > h.x; e.in1 = d2.y > h.y;
    e.sure0 = (e.in0 | (d2.x < l.x)) & (n2.x > tiny) & (n2.x < huge);
    e.sure1 = (e.in1 | (d2.y < l.y)) & (n2.y > tiny) & (n2.y < huge);
    return e;
  }
  static __device__ __forceinline__ V2 dot_fma2(V2 px, V2 py, V2 pz, V2 bx, V2 by, V2 bz) {
    return __builtin_elementwise_fma(px, bx, __builtin_elementwise_fma(py, by, pz * bz));
  }
  __device__ __forceinline__ void in23_rot_x2(V2 rx, V2 ry, V2 rz, V2 bx, V2 by, V2 bz, T c, bool& a, bool& b) const {
#pragma clang fp contract(off)
    V2 px = rx + t[0], py = ry + t[1], pz = rz + t[2];
    const V2 n2 = px * px + py * py + pz * pz;
    RPE_SCORE_STAT(0);
#ifndef RPE_NO_23_FILTER
    {
      const Est23 e = estimate23(px, py, pz, n2, bx, by, bz, c);
      // the bound above holds for UNIT bearings (A <= |bv| (1 + 2u)): both errors scale with |bv|, so a lane counts as decided only
      // if |bv|^2 is within 1e-3 of 1 (in which A <= 1.0006 is already counted); any other bearing -- the API does not normalise them --
      // takes the reference's own sequence.  Hypothesis-independent: hoisted out of the hypothesis loop.
      const V2 b2 = bx * bx + by * by + bz * bz;
      const bool unit0 = (b2.x > T(0.999)) & (b2.x < T(1.001)), unit1 = (b2.y > T(0.999)) & (b2.y < T(1.001));
      if (__builtin_amdgcn_ballot_w64(!((e.sure0 & unit0) & (e.sure1 & unit1))) == 0) { a = e.in0; b = e.in1; return; }
    }
#endif
    RPE_SCORE_STAT(1);
    const V2 len = {sqrt(n2.x), sqrt(n2.y)};
    px = px / len; py = py / len; pz = pz / len;
    const V2 d = px * bx + py * by + pz * bz;
    a = d.x > c; b = d.y > c;
  }
  // The estimate of a pair as LANE MASKS (the batched scoring kernel's form: the comparisons ARE the ballots, everything after them is
  // scalar -- as booleans the flags went through vector registers and back, two vector instructions per flag and pair): in = above the
  // band, sure = outside the band with |p|^2 in range; lanes with a non-unit bearing are the caller's to exclude (a mask of its own,
  // formed once per group).
  __device__ __forceinline__ void in23_masks(V2 rx, V2 ry, V2 rz, V2 bx, V2 by, V2 bz, T c, wave_mask_t& in_a, wave_mask_t& in_b,
                                             wave_mask_t& sure_a, wave_mask_t& sure_b) const {
#pragma clang fp contract(off)
    const V2 px = rx + t[0], py = ry + t[1], pz = rz + t[2];
    const V2 n2 = px * px + py * py + pz * pz;
    const V2 dt = dot_fma2(px, py, pz, bx, by, bz);
    // (a threshold within the band of zero: nothing is above +inf |p|^2, nothing below -inf |p|^2 -- no flag, no select)
    const T hi = c + kBand23, lo = c - kBand23;
    const T hi2 = lo > T(0) ? hi * hi : __builtin_inff(), lo2 = lo > T(0) ? lo * lo : -__builtin_inff();
    const V2 d2 = {dt.x * fabs(dt.x), dt.y * fabs(dt.y)};
    const V2 h = n2 * hi2, l = n2 * lo2;
    const T tiny = T(1e-30), huge = T(1e30);
    in_a = __builtin_amdgcn_ballot_w64(d2.x > h.x); in_b = __builtin_amdgcn_ballot_w64(d2.y > h.y);
    sure_a = (in_a | __builtin_amdgcn_ballot_w64(d2.x < l.x)) & __builtin_amdgcn_ballot_w64(n2.x > tiny) & __builtin_amdgcn_ballot_w64(n2.x < huge);
    sure_b = (in_b | __builtin_amdgcn_ballot_w64(d2.y < l.y)) & __builtin_amdgcn_ballot_w64(n2.y > tiny) & __builtin_amdgcn_ballot_w64(n2.y < huge);
  }
  // The same filter, DECIDING ONLY (booleans; callers without a queue): a = b = the decided votes, need_a / need_b = this element is inside the
  // band (or not a unit bearing, or |p|^2 out of range) and must be given the reference's own sequence -- by the caller, LATER: with
  // realistic bearing noise 0.5 % of the elements sit inside the band, i.e. one wave in two holds one, and a wave that runs the square
  // root and the six divisions for one lane pays them for all 128 elements (counted: profiles/r05_score_filter_stats.jsonl, 69 % of the
  // wave evaluations fell through at 24 u).  The caller queues those elements in LDS and evaluates them densely (score_kernel, DeferQ).
  __device__ __forceinline__ void in23_rot_x2_decide(V2 rx, V2 ry, V2 rz, V2 bx, V2 by, V2 bz, T c, bool& a, bool& b, bool& need_a, bool& need_b) const {
#pragma clang fp contract(off)
    const V2 px = rx + t[0], py = ry + t[1], pz = rz + t[2];
    const V2 n2 = px * px + py * py + pz * pz;
    const Est23 e = estimate23(px, py, pz, n2, bx, by, bz, c);
    const V2 b2 = bx * bx + by * by + bz * bz;
    const bool unit0 = (b2.x > T(0.999)) & (b2.x < T(1.001)), unit1 = (b2.y > T(0.999)) & (b2.y < T(1.001));
    need_a = !(e.sure0 & unit0);
    need_b = !(e.sure1 & unit1);
    a = e.in0 & !need_a; b = e.in1 & !need_b;
  }
  // ... and the reference's sequence on ONE rotated point (what in23_rot_x2 runs behind its filter, element by element: the same bits)
  static __device__ __forceinline__ bool in23_reference(T rx, T ry, T rz, T t0, T t1, T t2, T bx, T by, T bz, T c) {
#pragma clang fp contract(off)
    T px = rx + t0, py = ry + t1, pz = rz + t2;
    const T len = sqrt(px * px + py * py + pz * pz);
    px = px / len; py = py / len; pz = pz / len;
    return (px * bx + py * by + pz * bz) > c;
  }
  __device__ __forceinline__ void innnx2(V2 nwx, V2 nwy, V2 nwz, V2 ncx, V2 ncy, V2 ncz, T cnl, bool& a, bool& b) const {
#pragma clang fp contract(off)
    V2 rx, ry, rz;
    rot2(nwx, nwy, nwz, rx, ry, rz);
    const V2 d = ncx * rx + ncy * ry + ncz * rz;
    a = d.x > cnl; b = d.y > cnl;
  }
  __device__ __forceinline__ bool in23(T x, T y, T z, T bx, T by, T bz, T c, bool use_matrix) const {
#pragma clang fp contract(off)
    T rx, ry, rz;
    if (use_matrix) {
      rx = M[0] * x + M[1] * y + M[2] * z; ry = M[3] * x + M[4] * y + M[5] * z; rz = M[6] * x + M[7] * y + M[8] * z;
    } else {
      rot(x, y, z, rx, ry, rz);
    }
    T px = rx + t[0], py = ry + t[1], pz = rz + t[2];
    const T len = sqrt(px * px + py * py + pz * pz);
    px = px / len; py = py / len; pz = pz / len;
    return (px * bx + py * by + pz * bz) > c;
  }
  __device__ __forceinline__ bool innn(T nwx, T nwy, T nwz, T ncx, T ncy, T ncz, T cnl) const {
#pragma clang fp contract(off)
    T rx, ry, rz;
    rot(nwx, nwy, nwz, rx, ry, rz);
    return (ncx * rx + ncy * ry + ncz * rz) > cnl;
  }
};

// votes of a predicate over the wave, among the lanes of `mask`: the compare's lane mask IS the ballot (__builtin_amdgcn_ballot_w64 on
// the compare itself), the hypothesis-independent part of the predicate (present / valid) is a wave mask taken once per group and
// applied with one scalar AND, one s_bcnt1 counts.  __ballot(valid & pred) made the compiler materialise the combined predicate as
// a 0 / 1 VGPR and compare it with zero again: two vector instructions per predicate, 8 of the 46 per hypothesis and group in the 3D
// fast loop (profiles/r04_score_sq_counters.json).
__device__ __forceinline__ int votes_of(wave_mask_t mask, bool pred) {
  return __builtin_popcountll(__builtin_amdgcn_ballot_w64(pred) & mask);
}

// ---- deferred exact 2D votes (score_kernel, fp32 EXACT kinds with a 2D test): elements the band filter cannot decide are queued --
// rotated point, bearing, hypothesis slot: 8 values -- and given the reference's sequence DENSELY, 64 at a time, one element per lane.
// Every wave has a queue of its own in LDS (no atomics, no workgroup barriers: the count is a wave-uniform register) and drains it
// whenever 64 entries have gathered, and at the end of its tile.
constexpr int kDeferEntries = 192;   // per wave: fewer than 64 left by the last drain + at most 128 new ones from one pair
template <class T> struct DeferQ {
  T* entry;              // this wave's kDeferEntries x 8 values of LDS
  const T* poses;        // the chunk's hypotheses (exact layout: qw qx qy qz tx ty tz pad)
  int* votes;            // the workgroup's vote table (LDS)
  int n;                 // entries queued (wave-uniform)
  // the reference's sequence for entries [0, n): the votes go to the table by LDS atomics
  __device__ __forceinline__ void drain(T cthr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int lane = threadIdx.x & 63;
    for (int e = lane; e < n; e += 64) {
      const T* en = entry + 8 * e;
      const int slot = (int)en[6];
      const T* hq = poses + (size_t)slot * 8;
      if (Hyp<T, true>::in23_reference(en[0], en[1], en[2], hq[4], hq[5], hq[6], en[3], en[4], en[5], cthr)) atomicAdd(&votes[slot], 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    n = 0;
  }
  // the undecided elements of a PAIR (element a: *.x, element b: *.y)
  typedef T V2 __attribute__((ext_vector_type(2)));
  __device__ __forceinline__ void append(wave_mask_t ma, wave_mask_t mb, V2 rx, V2 ry, V2 rz, V2 bx, V2 by, V2 bz, int slot, T cthr) {
    RPE_SCORE_STAT(0);                 // (diagnostic build: wave-pair evaluations of the batched kernel ...
    if ((ma | mb) == 0 || slot < 0) return;   // (slot < 0: a padding hypothesis queues nothing)
    const bool need_a = (ma >> (threadIdx.x & 63)) & 1ull, need_b = (mb >> (threadIdx.x & 63)) & 1ull;
    RPE_SCORE_STAT(1);                 // ... of which at least one lane sits inside the band ...
    RPE_SCORE_STAT_ADD(2, __builtin_popcountll(ma) + __builtin_popcountll(mb));   // ... and how many elements that is)
    const int lane = threadIdx.x & 63;
    const wave_mask_t below = (1ull << lane) - 1ull;
    const int ca = __builtin_popcountll(ma);
    if (need_a) { T* e = entry + 8 * (n + __builtin_popcountll(ma & below)); e[0] = rx.x; e[1] = ry.x; e[2] = rz.x; e[3] = bx.x; e[4] = by.x; e[5] = bz.x; e[6] = (T)slot; }
    if (need_b) { T* e = entry + 8 * (n + ca + __builtin_popcountll(mb & below)); e[0] = rx.y; e[1] = ry.y; e[2] = rz.y; e[3] = bx.y; e[4] = by.y; e[5] = bz.y; e[6] = (T)slot; }
    n = __builtin_amdgcn_readfirstlane(n + ca + __builtin_popcountll(mb));   // (wave-uniform: kept in a scalar register)
    if (n >= 64) drain(cthr);
  }
};

// votes of ONE hypothesis over one group of P correspondences, summed over the wave (every lane gets the wave's count).  Predicates are
// evaluated unconditionally and masked with '&': no divergent branches; the compare IS the ballot.  EXACT: the 3D and normal tests run
// on pairs of correspondences as 2-vectors (packed fp32 instructions), the 2D test (a square root and three divisions) stays scalar.
// DEFER: queue the undecided 2D votes in `defer` (score_kernel's deferred-exact queue; the other callers pass a dummy and DEFER = false);
// slot = where the hypothesis' deferred votes go (< 0: a padding hypothesis whose count is dropped -- nothing is queued for it)
template <int P> __device__ __forceinline__ bool any_bearing(const wave_mask_t (&present)[P]) {
  wave_mask_t m = 0;
#pragma unroll
  for (int i = 0; i < P; i++) m |= present[i];
  return m != 0;
}
// W23 (fast mode; wave-uniform, decided ONCE per group by the caller -- any_bearing() -- not per correspondence: four scalar branches
// per hypothesis cut the fast loops' straight-line code into pieces and cost the all-bearing scenes 12-20 %): false = no lane of the
// wave has a bearing, the 2D test is left out altogether.  The exact kinds keep their per-pair test (always W23 = true).
template <class T, int KIND, bool EXACT, bool DEFER = false, bool W23 = true>
__device__ __forceinline__ int count_group_votes(const Hyp<T, EXACT>& hyp, const T (&vw)[3 * Pk<T>::P], const T (&vc)[3 * Pk<T>::P],
                                                 const T (&vb)[3 * Pk<T>::P], const T (&vnw)[3 * Pk<T>::P], const T (&vnc)[3 * Pk<T>::P],
                                                 const wave_mask_t (&present)[Pk<T>::P], const wave_mask_t (&valid)[Pk<T>::P], T thr33, T cthr, T cnl,
                                                 DeferQ<T>& defer, int slot = -1, const wave_mask_t* unit = nullptr) {
  // (unit[], DEFER only: the lanes of present[] whose bearing is of unit length to 1e-3 -- what the 2D estimate's bound covers)
  // (present[] for the 2D test: lanes whose correspondence exists AND whose bearing holds no NaN -- a NaN bearing makes the reference's
  // comparison false whatever the hypothesis, so such a lane never votes, and a wave without any bearing at all -- configs[2] has 2 000
  // of them among 307 200 correspondences -- skips the test: the callers build the masks that way, once per group)
  constexpr int P = Pk<T>::P;
  typedef VoteMods<KIND> MD;
  int cnt = 0;
  if constexpr (EXACT) {
    typedef T V2 __attribute__((ext_vector_type(2)));
    static_assert(P % 2 == 0, "pairs");
#pragma unroll
    for (int j = 0; j < P / 2; j++) {
      const int a = 2 * j, b = 2 * j + 1;
      const V2 x = {vw[3 * a], vw[3 * b]}, y = {vw[3 * a + 1], vw[3 * b + 1]}, z = {vw[3 * a + 2], vw[3 * b + 2]};
      if (MD::mnn) {
        const V2 nwx = {vnw[3 * a], vnw[3 * b]}, nwy = {vnw[3 * a + 1], vnw[3 * b + 1]}, nwz = {vnw[3 * a + 2], vnw[3 * b + 2]};
        const V2 ncx = {vnc[3 * a], vnc[3 * b]}, ncy = {vnc[3 * a + 1], vnc[3 * b + 1]}, ncz = {vnc[3 * a + 2], vnc[3 * b + 2]};
        bool va, vb2;
        hyp.innnx2(nwx, nwy, nwz, ncx, ncy, ncz, cnl, va, vb2);
        cnt += votes_of(valid[a], va) + votes_of(valid[b], vb2);
      }
      V2 rx, ry, rz;
      if (KIND == VOTE_23_MATRIX) hyp.rotm2(x, y, z, rx, ry, rz); else hyp.rot2(x, y, z, rx, ry, rz);
      if (MD::m33) {
        const V2 cx = {vc[3 * a], vc[3 * b]}, cy = {vc[3 * a + 1], vc[3 * b + 1]}, cz = {vc[3 * a + 2], vc[3 * b + 2]};
        bool va, vb2;
        hyp.in33_rot_x2(rx, ry, rz, cx, cy, cz, thr33, va, vb2);
        cnt += votes_of(valid[a], va) + votes_of(valid[b], vb2);
      }
      if (MD::m23 && (present[a] | present[b]) != 0) {
        const V2 bx = {vb[3 * a], vb[3 * b]}, by = {vb[3 * a + 1], vb[3 * b + 1]}, bz = {vb[3 * a + 2], vb[3 * b + 2]};
        bool va, vb2;
        if constexpr (DEFER) {
          wave_mask_t ia, ib, sa, sb;
          hyp.in23_masks(rx, ry, rz, bx, by, bz, cthr, ia, ib, sa, sb);
          sa &= unit[a]; sb &= unit[b];                                   // decided: outside the band, in range, a present unit bearing
          defer.append(present[a] & ~sa, present[b] & ~sb, rx, ry, rz, bx, by, bz, slot, cthr);
          cnt += __builtin_popcountll(ia & sa) + __builtin_popcountll(ib & sb);
        } else {
          hyp.in23_rot_x2(rx, ry, rz, bx, by, bz, cthr, va, vb2);
          cnt += votes_of(present[a], va) + votes_of(present[b], vb2);
        }
      }
    }
  } else {
    typedef T V2 __attribute__((ext_vector_type(2)));
    static_assert(P % 2 == 0, "pairs");
#pragma unroll
    for (int j = 0; j < P / 2; j++) {
      const int a = 2 * j, b = 2 * j + 1;
      bool va, vb2;
      if (MD::mnn) {
        const V2 nwx = {vnw[3 * a], vnw[3 * b]}, nwy = {vnw[3 * a + 1], vnw[3 * b + 1]}, nwz = {vnw[3 * a + 2], vnw[3 * b + 2]};
        const V2 ncx = {vnc[3 * a], vnc[3 * b]}, ncy = {vnc[3 * a + 1], vnc[3 * b + 1]}, ncz = {vnc[3 * a + 2], vnc[3 * b + 2]};
        hyp.innn2(nwx, nwy, nwz, ncx, ncy, ncz, cnl, va, vb2);
        cnt += votes_of(valid[a], va) + votes_of(valid[b], vb2);
      }
      if (MD::m33 || (MD::m23 && W23)) {
        const V2 x = {vw[3 * a], vw[3 * b]}, y = {vw[3 * a + 1], vw[3 * b + 1]}, z = {vw[3 * a + 2], vw[3 * b + 2]};
        V2 px, py, pz;
        hyp.xform2(x, y, z, px, py, pz);   // one transform for the 3D and the 2D test
        if (MD::m33) {
          const V2 cx = {vc[3 * a], vc[3 * b]}, cy = {vc[3 * a + 1], vc[3 * b + 1]}, cz = {vc[3 * a + 2], vc[3 * b + 2]};
          Hyp<T, false>::in33_p2(px, py, pz, cx, cy, cz, thr33, va, vb2);
          cnt += votes_of(valid[a], va) + votes_of(valid[b], vb2);
        }
        if (MD::m23 && W23) {
          const V2 bx = {vb[3 * a], vb[3 * b]}, by = {vb[3 * a + 1], vb[3 * b + 1]}, bz = {vb[3 * a + 2], vb[3 * b + 2]};
          Hyp<T, false>::in23_p2(px, py, pz, bx, by, bz, cthr, va, vb2);
          cnt += votes_of(present[a], va) + votes_of(present[b], vb2);
        }
      }
    }
  }
  return cnt;
}

// grid = (x: correspondence tiles, grid-stride) x (y: chunks of `hchunk` hypotheses).  Small problems (640x480 frames)
// cannot fill 256 CUs with one tile sweep, so the hypothesis list is split across blockIdx.y and the (L2-resident)
// arrays are swept once per chunk; large problems use one chunk so the arrays stream from HBM once per launch.
template <class T, int KIND, bool EXACT>
__global__ __launch_bounds__(kBlock) void score_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const T* __restrict__ bv,
                                                       const T* __restrict__ nw, const T* __restrict__ nc, int64_t n,
                                                       const T* __restrict__ poses, int H, int hchunk, T thr33, T cthr, T cnl,
                                                       int* __restrict__ votes) {
  constexpr int P = Pk<T>::P;
  typedef VoteMods<KIND> MD;
  extern __shared__ int lds_votes[];
  const int hbeg = blockIdx.y * hchunk;
  const int hcnt = min(hchunk, H - hbeg);
  for (int i = threadIdx.x; i < hcnt; i += kBlock) lds_votes[i] = 0;
  // deferred exact 2D votes (DeferQ): every wave's queue sits behind the vote table in the dynamic LDS (score_launch sizes it)
  constexpr bool DEFER = sizeof(T) == 4 && EXACT && MD::m23 && KIND != VOTE_23_MATRIX;
  DeferQ<T> dq;
  dq.entry = reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(lds_votes) + (((size_t)hchunk * sizeof(int) + 15) & ~(size_t)15)) + (size_t)(threadIdx.x >> 6) * kDeferEntries * 8;
  dq.poses = poses + (size_t)hbeg * Hyp<T, EXACT>::STRIDE;
  dq.votes = lds_votes;
  dq.n = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t groups = (n + P - 1) / P;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  // loop bound is workgroup-uniform so that every lane of a wave takes part in the per-hypothesis ballots
  for (int64_t gb = (int64_t)blockIdx.x * kBlock; gb < groups; gb += stride) {
    const int64_t g = gb + threadIdx.x;
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    wave_mask_t present[P], valid[P], unit[P];   // hypothesis-independent: lane masks, once per group
    load_group<T>(xw, g, n, vw);
    if (MD::need_xc) load_group<T>(xc, g, n, vc);
    if (MD::m23) load_group<T>(bv, g, n, vb);
    if (MD::mnn) { load_group<T>(nw, g, n, vnw); load_group<T>(nc, g, n, vnc); }
#pragma unroll
    for (int i = 0; i < P; i++) {
      const bool here = (g * P + i) < n;
      // (only the 2D test reads present[]: lanes with a NaN in their bearing are left out -- see count_group_votes)
      present[i] = __builtin_amdgcn_ballot_w64(here & (!MD::m23 || !(vb[3 * i] != vb[3 * i] || vb[3 * i + 1] != vb[3 * i + 1] || vb[3 * i + 2] != vb[3 * i + 2])));
      valid[i] = __builtin_amdgcn_ballot_w64(here & (!MD::need_xc || !all_nan(vc[3 * i], vc[3 * i + 1], vc[3 * i + 2])));
      unit[i] = 0;
      if (DEFER) {
        const T b2 = vb[3 * i] * vb[3 * i] + vb[3 * i + 1] * vb[3 * i + 1] + vb[3 * i + 2] * vb[3 * i + 2];
        unit[i] = present[i] & __builtin_amdgcn_ballot_w64((b2 > T(0.999)) & (b2 < T(1.001)));
      }
    }
    auto score_list = [&](auto w23) {
    constexpr bool W23 = decltype(w23)::value;
    for (int h0 = 0; h0 < hcnt; h0 += 64) {
      const int hmax = min(64, hcnt - h0);
      int mine = 0;
      // HU hypotheses per trip: their scalar loads are issued together and waited for once -- with one hypothesis per trip a wave of the
      // 3D fast kind spent 47 % of its life in that wait (SQ_WAIT_ANY / SQ_WAVE_CYCLES, profiles/r04_score_sq_counters.json; a prefetch
      // of the next hypothesis is sunk back to its use by the compiler, so the wait is shared instead of hidden).  Four where the
      // registers allow, two for the exact 2D test (profiles/r04_score_hypotheses_per_trip_ab.txt).
      constexpr int HU = (MD::m23 && EXACT) ? 2 : 4;
      const T* hp = poses + (size_t)(hbeg + h0) * Hyp<T, EXACT>::STRIDE;
      for (int hl = 0; hl < hmax; hl += HU) {
        Hyp<T, EXACT> hyp[HU];
#pragma unroll
        for (int u = 0; u < HU; u++)   // past the end of the list: the last hypothesis again (its count is dropped below)
          hyp[u].load(hp + (size_t)(hl + u < hmax ? hl + u : hmax - 1) * Hyp<T, EXACT>::STRIDE, KIND == VOTE_23_MATRIX);
#pragma unroll
        for (int u = 0; u < HU; u++) {
          const int cnt = count_group_votes<T, KIND, EXACT, DEFER, W23>(hyp[u], vw, vc, vb, vnw, vnc, present, valid, thr33, cthr, cnl, dq,
                                                                        hl + u < hmax ? h0 + hl + u : -1, unit);
          mine = (lane == hl + u) ? cnt : mine;   // every hypothesis once per block of 64, `mine` starts at 0: a select, not an add
        }
      }
      if (lane < hmax && mine != 0) atomicAdd(&lds_votes[h0 + lane], mine);
    }
    };
    // The whole list with or without the 2D test: a wave none of whose lanes holds a bearing -- configs[2] has 2 000 among 307 200
    // correspondences -- never enters it.  Fast mode only: the exact kinds, heavier in registers, keep their per-pair test (with the
    // loop duplicated, NN + 3D + 2D exact took 324 instead of 297 us per 512 x 307 200 pass).
    if (!EXACT && MD::m23 && !any_bearing<P>(present)) score_list(std::false_type{}); else score_list(std::true_type{});
    if (DEFER && dq.n > 0) dq.drain(cthr);   // what is left of the tile's undecided 2D votes
  }
  __syncthreads();
  for (int i = threadIdx.x; i < hcnt; i += kBlock) {
    const int v = lds_votes[i];
    if (v != 0) atomicAdd(&votes[hbeg + i], v);
  }
}

// Small batches (the first RANSAC batches: 8 .. 32 hypotheses): ONE launch and no device-side staging at all -- the hypotheses arrive
// as
// a kernel argument (no H2D copy), every wave counts as above (lane h holds hypothesis h's count), the per-wave counts go straight into
// the collecting stage (collect_and_send: integers < 2^53 as doubles, exact), and the host adds the run records.  Replaces copy +
// scoring kernel + read-out kernel + flag (42 us per batch of 16 at 640 x 480) for lists of up to HB hypotheses.
template <class T, int HB, int STRIDE> struct SmallPoses { T v[HB * STRIDE]; };
template <class T, int KIND, bool EXACT, int HB>
__global__ __launch_bounds__(kBlock) void score_small_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const T* __restrict__ bv,
                             const T* __restrict__ nw, const T* __restrict__ nc, int64_t n, SmallPoses<T, HB, Hyp<T, EXACT>::STRIDE> sp,
                             const T* __restrict__ dposes, int H, int hs, T thr33, T cthr, T cnl, Finish fin) {
  constexpr int P = Pk<T>::P;
  typedef VoteMods<KIND> MD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // hs (1, 2 or 4) waves share a tile of correspondences and split the list between them (wave copy c takes hypotheses c, c + hs, ...):
  // a frame-sized problem then runs as 4 x as many, 4 x shorter waves -- the pass is a long serial chain per wave (every predicate of
  // every hypothesis on the wave's points), so with one group per thread it is bound by that chain, not by memory or issue rate
  const int per_copy = (kBlock / 64) / hs, copy = wave / per_copy;
  const int tile = kBlock / hs;
  const int64_t groups = (n + P - 1) / P;
  const int64_t stride = (int64_t)gridDim.x * tile;
  int mine = 0;
  // loop bound is workgroup-uniform so that every lane of a wave takes part in the per-hypothesis ballots
  for (int64_t gb = (int64_t)blockIdx.x * tile; gb < groups; gb += stride) {
    const int64_t g = gb + (wave % per_copy) * 64 + lane;
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    wave_mask_t present[P], valid[P];   // hypothesis-independent: lane masks, once per group
    load_group<T>(xw, g, n, vw);
    if (MD::need_xc) load_group<T>(xc, g, n, vc);
    if (MD::m23) load_group<T>(bv, g, n, vb);
    if (MD::mnn) { load_group<T>(nw, g, n, vnw); load_group<T>(nc, g, n, vnc); }
#pragma unroll
    for (int i = 0; i < P; i++) {
      const bool here = (g * P + i) < n;
      // (only the 2D test reads present[]: lanes with a NaN in their bearing are left out -- see count_group_votes)
      present[i] = __builtin_amdgcn_ballot_w64(here & (!MD::m23 || !(vb[3 * i] != vb[3 * i] || vb[3 * i + 1] != vb[3 * i + 1] || vb[3 * i + 2] != vb[3 * i + 2])));
      valid[i] = __builtin_amdgcn_ballot_w64(here & (!MD::need_xc || !all_nan(vc[3 * i], vc[3 * i + 1], vc[3 * i + 2])));
    }
    const bool w23 = EXACT || !MD::m23 || any_bearing<P>(present);
    for (int hl = copy; hl < H; hl += hs) {
      Hyp<T, EXACT> hyp;
      if (dposes) hyp.load(dposes + (size_t)hl * Hyp<T, EXACT>::STRIDE, KIND == VOTE_23_MATRIX);   // a list generated on the device
      else hyp.load(sp.v + hl * Hyp<T, EXACT>::STRIDE, KIND == VOTE_23_MATRIX);
      DeferQ<T> none;   // (no queue here: the in-place filter)
      const int cnt = w23 ? count_group_votes<T, KIND, EXACT, false, true>(hyp, vw, vc, vb, vnw, vnc, present, valid, thr33, cthr, cnl, none)
                          : count_group_votes<T, KIND, EXACT, false, false>(hyp, vw, vc, vb, vnw, vnc, present, valid, thr33, cthr, cnl, none);
      mine += (lane == hl) ? cnt : 0;
    }
  }
  __shared__ double red[kBlock / 64][HB];
  if (lane < HB) red[wave][lane] = (double)mine;
  __syncthreads();
  collect_and_send<HB, 0, kBlock>(red, fin);
}

// P flags of one group as ONE store (8 bytes for fp32 / P = 4, 4 bytes for fp64 / P = 2): a thread owns P consecutive
// correspondences, so its shorts are contiguous; 2-byte scattered stores cost an order of magnitude more per byte.
// the masks are written once and not read by this kernel: nontemporal stores (2-3 % of the launch at 10 M correspondences)
__device__ __forceinline__ void store_mask_full(short* __restrict__ m, int64_t g, const bool (&v)[4]) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  u2 u;
  u.x = (unsigned int)v[0] | ((unsigned int)v[1] << 16);
  u.y = (unsigned int)v[2] | ((unsigned int)v[3] << 16);
  __builtin_nontemporal_store(u, reinterpret_cast<u2*>(m + 4 * g));
}
__device__ __forceinline__ void store_mask_full(short* __restrict__ m, int64_t g, const bool (&v)[2]) {
  __builtin_nontemporal_store((unsigned int)v[0] | ((unsigned int)v[1] << 16), reinterpret_cast<unsigned int*>(m + 2 * g));
}

template <class T> struct PoseArg { T v[12]; };  // one hypothesis by value (kernel argument): no H2D copy for a single pose

// the inlier flags of ONE hypothesis over one full group: the predicates of count_group_votes (EXACT: pairs of correspondences as
// 2-vectors, one rotation of the world point for the 3D and the 2D test, the 2D test behind its wave-uniform filter), kept as flags
template <class T, int KIND, bool EXACT>
__device__ __forceinline__ void group_flags(const Hyp<T, EXACT>& hyp, const T (&vw)[3 * Pk<T>::P], const T (&vc)[3 * Pk<T>::P],
                                            const T (&vb)[3 * Pk<T>::P], const T (&vnw)[3 * Pk<T>::P], const T (&vnc)[3 * Pk<T>::P],
                                            T thr33, T cthr, T cnl, bool (&f23)[Pk<T>::P], bool (&f33)[Pk<T>::P], bool (&fnn)[Pk<T>::P]) {
  constexpr int P = Pk<T>::P;
  typedef VoteMods<KIND> MD;
  bool valid[P];
#pragma unroll
  for (int i = 0; i < P; i++) {
    valid[i] = !MD::need_xc || !all_nan(vc[3 * i], vc[3 * i + 1], vc[3 * i + 2]);
    f23[i] = f33[i] = fnn[i] = false;
  }
  if constexpr (EXACT) {
    typedef T V2 __attribute__((ext_vector_type(2)));
    static_assert(P % 2 == 0, "pairs");
#pragma unroll
    for (int j = 0; j < P / 2; j++) {
      const int a = 2 * j, b = 2 * j + 1;
      const V2 x = {vw[3 * a], vw[3 * b]}, y = {vw[3 * a + 1], vw[3 * b + 1]}, z = {vw[3 * a + 2], vw[3 * b + 2]};
      if (MD::mnn) {
        const V2 nwx = {vnw[3 * a], vnw[3 * b]}, nwy = {vnw[3 * a + 1], vnw[3 * b + 1]}, nwz = {vnw[3 * a + 2], vnw[3 * b + 2]};
        const V2 ncx = {vnc[3 * a], vnc[3 * b]}, ncy = {vnc[3 * a + 1], vnc[3 * b + 1]}, ncz = {vnc[3 * a + 2], vnc[3 * b + 2]};
        bool va, vb2;
        hyp.innnx2(nwx, nwy, nwz, ncx, ncy, ncz, cnl, va, vb2);
        fnn[a] = valid[a] & va; fnn[b] = valid[b] & vb2;
      }
      V2 rx, ry, rz;
      if (KIND == VOTE_23_MATRIX) hyp.rotm2(x, y, z, rx, ry, rz); else hyp.rot2(x, y, z, rx, ry, rz);
      if (MD::m33) {
        const V2 cx = {vc[3 * a], vc[3 * b]}, cy = {vc[3 * a + 1], vc[3 * b + 1]}, cz = {vc[3 * a + 2], vc[3 * b + 2]};
        bool va, vb2;
        hyp.in33_rot_x2(rx, ry, rz, cx, cy, cz, thr33, va, vb2);
        f33[a] = valid[a] & va; f33[b] = valid[b] & vb2;
      }
      if (MD::m23) {
        const V2 bx = {vb[3 * a], vb[3 * b]}, by = {vb[3 * a + 1], vb[3 * b + 1]}, bz = {vb[3 * a + 2], vb[3 * b + 2]};
        hyp.in23_rot_x2(rx, ry, rz, bx, by, bz, cthr, f23[a], f23[b]);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < P; i++) {
      const T x = vw[3 * i], y = vw[3 * i + 1], z = vw[3 * i + 2];
      if (MD::mnn) fnn[i] = valid[i] & hyp.innn(vnw[3 * i], vnw[3 * i + 1], vnw[3 * i + 2], vnc[3 * i], vnc[3 * i + 1], vnc[3 * i + 2], cnl);
      if (MD::m33) f33[i] = valid[i] & hyp.in33(x, y, z, vc[3 * i], vc[3 * i + 1], vc[3 * i + 2], thr33);
      if (MD::m23) f23[i] = hyp.in23(x, y, z, vb[3 * i], vb[3 * i + 1], vb[3 * i + 2], cthr, KIND == VOTE_23_MATRIX);
    }
  }
}
// K4b: the winner's inlier masks.  The next group's vector loads are in flight while the current group's predicates are evaluated and
// its masks stored (the software pipeline of normal_eq_kernel, pinned down the same way).  What keeps the kinds with two or three
// masks at 0.66 of the HBM peak at 10 M correspondences is the STORES: with the stores compiled out the same loops run at 0.81-0.88
// (36.8 / 61.0 / 101.9 us against 41.2 / 76.0 / 125.4 for one / two / three masks, profiles/r04_k4b_stores_ab.jsonl) -- 20 MB of
// mask cost 4 / 15 / 23 us, far more than their share of the bytes; writes interleaved with a read stream are what the memory
// system likes least.
template <class T, int KIND, bool EXACT>
__global__ __launch_bounds__(kBlock) void mask_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const T* __restrict__ bv,
                                                           const T* __restrict__ nw, const T* __restrict__ nc, int64_t n,
                                                           PoseArg<T> pose, T thr33, T cthr, T cnl, short* __restrict__ m23,
                                                           short* __restrict__ m33, short* __restrict__ mnn, Finish fin) {
  constexpr int P = Pk<T>::P;
  typedef typename Pk<T>::V V;
  typedef VoteMods<KIND> MD;
  constexpr bool need[5] = {true, MD::need_xc, MD::m23, MD::mnn, MD::mnn};
  Hyp<T, EXACT> hyp;
  hyp.load(pose.v, KIND == VOTE_23_MATRIX);
  int cnt = 0;
  const int64_t full = n / P;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const V* __restrict__ a4[5] = {reinterpret_cast<const V*>(xw), reinterpret_cast<const V*>(xc), reinterpret_cast<const V*>(bv),
                                 reinterpret_cast<const V*>(nw), reinterpret_cast<const V*>(nc)};
  int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  V cur[5][3];
#pragma unroll
  for (int a = 0; a < 5; a++) { cur[a][0] = V{}; cur[a][1] = V{}; cur[a][2] = V{}; }
  if (g < full) {
#pragma unroll
    for (int a = 0; a < 5; a++)
      if (need[a]) { cur[a][0] = a4[a][3 * g]; cur[a][1] = a4[a][3 * g + 1]; cur[a][2] = a4[a][3 * g + 2]; }
  }
  while (g < full) {
    const int64_t gn = g + stride;
    const int64_t gl = gn < full ? gn : g;  // clamp: the last trip re-reads its own (cached) group instead of branching
    V nxt[5][3];
#pragma unroll
    for (int a = 0; a < 5; a++)
      if (need[a]) { nxt[a][0] = a4[a][3 * gl]; nxt[a][1] = a4[a][3 * gl + 1]; nxt[a][2] = a4[a][3 * gl + 2]; }
    __builtin_amdgcn_sched_barrier(0);
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    unpack3(cur[0][0], cur[0][1], cur[0][2], vw);
    if (MD::need_xc) unpack3(cur[1][0], cur[1][1], cur[1][2], vc);
    if (MD::m23) unpack3(cur[2][0], cur[2][1], cur[2][2], vb);
    if (MD::mnn) { unpack3(cur[3][0], cur[3][1], cur[3][2], vnw); unpack3(cur[4][0], cur[4][1], cur[4][2], vnc); }
    bool f23[P], f33[P], fnn[P];
    group_flags<T, KIND, EXACT>(hyp, vw, vc, vb, vnw, vnc, thr33, cthr, cnl, f23, f33, fnn);
#pragma unroll
    for (int i = 0; i < P; i++) cnt += (int)fnn[i] + (int)f33[i] + (int)f23[i];
    if (MD::mnn) store_mask_full(mnn, g, fnn);
    if (MD::m33) store_mask_full(m33, g, f33);
    if (MD::m23) store_mask_full(m23, g, f23);
#pragma unroll
    for (int a = 0; a < 5; a++)
      if (need[a]) { cur[a][0] = nxt[a][0]; cur[a][1] = nxt[a][1]; cur[a][2] = nxt[a][2]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < 5; a++)
      if (need[a]) { pin16(cur[a][0]); pin16(cur[a][1]); pin16(cur[a][2]); }
    g = gn;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && full * P < n) {  // leftover correspondences through the bounds-checked loaders
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    load_group<T>(xw, full, n, vw);
    if (MD::need_xc) load_group<T>(xc, full, n, vc);
    if (MD::m23) load_group<T>(bv, full, n, vb);
    if (MD::mnn) { load_group<T>(nw, full, n, vnw); load_group<T>(nc, full, n, vnc); }
#pragma unroll
    for (int i = 0; i < P; i++) {
      const int64_t idx = full * P + i;
      if (idx < n) {
        const bool valid = !MD::need_xc || !all_nan(vc[3 * i], vc[3 * i + 1], vc[3 * i + 2]);
        const T x = vw[3 * i], y = vw[3 * i + 1], z = vw[3 * i + 2];
        const bool fn = MD::mnn ? (valid & hyp.innn(vnw[3 * i], vnw[3 * i + 1], vnw[3 * i + 2], vnc[3 * i], vnc[3 * i + 1], vnc[3 * i + 2],
            cnl)) : false;
        const bool f3 = MD::m33 ? (valid & hyp.in33(x, y, z, vc[3 * i], vc[3 * i + 1], vc[3 * i + 2], thr33)) : false;
        const bool f2 = MD::m23 ? hyp.in23(x, y, z, vb[3 * i], vb[3 * i + 1], vb[3 * i + 2], cthr, KIND == VOTE_23_MATRIX) : false;
        cnt += (int)fn + (int)f3 + (int)f2;
        if (MD::mnn) mnn[idx] = fn;
        if (MD::m33) m33[idx] = f3;
        if (MD::m23) m23[idx] = f2;
      }
    }
  }
  double acc[1] = {(double)cnt};
  reduce_and_finish<1, kNeLd, 0, kBlock>(acc, fin);
}

// ================================================================================================
// K4r: RESIDENT scoring -- a whole RANSAC run on resident arrays in ONE launch (the reference's real entry points are whole runs:
// ao_ransac, Library.cpp:47-75; shinji_ransac2, AbsoluteOrientation.hpp:159-213).  Scored batch by batch with one launch each, a run
// at 640x480 is 2-3 launches of 8-32 hypotheses plus the mask launch: 37-75 us of which ~7.5 us per launch are fixed.  Here the grid
// stays resident for the run, as the Gauss-Newton loops' does: every thread keeps its group of correspondences in registers, the host
// hands every batch over through the control block in fine-grained device memory (written through the PCIe BAR), the vote counts come
// back as run records through the collecting stage (resident_cross_own), and the best-so-far / adaptive-Iter replay stays on the host
// exactly as before.  The winner's masks are written by the same grid (op 1).  Frame-sized problems only (one group per thread).
// Control block (512 words of 8 bytes): [0] tag | [1] count (low 32 bits), op (high: 0 score, 1 masks) | [2 ...] count hypotheses in
// the scoring layout of the mode, values of the array dtype | [511] tag again.  The host writes payload and header, a store fence, both
// tags (bit 63: stop -- with op 1 the masks are still written and their record sent before the grid leaves, any other op leaves at once).
// ================================================================================================
constexpr int kSessionHyps = kSessionHypsMax;          // hypotheses per batch
constexpr int kSessionCtlWords = kSessionCtlWordsMax;  // the context's control block: 4 KB
template <class T, int KIND, bool EXACT>
__global__ __launch_bounds__(512) void score_resident_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const T* __restrict__ bv,
                                                             const T* __restrict__ nw, const T* __restrict__ nc, int64_t n,
                                                             const unsigned long long* __restrict__ ctl, unsigned long long first_tag, T thr33,
                                                             T cthr, T cnl, short* __restrict__ m23, short* __restrict__ m33,
                                                             short* __restrict__ mnn, Finish fin) {
  constexpr int P = Pk<T>::P, BLK = 512, HB = kSessionHyps;
  constexpr int STRIDE = Hyp<T, EXACT>::STRIDE;
  constexpr int PAYLOAD_WORDS = (HB * STRIDE * (int)sizeof(T) + 7) / 8;
  static_assert(2 + PAYLOAD_WORDS < kSessionCtlWords, "the batch fits the control block");
  typedef VoteMods<KIND> MD;
  __shared__ unsigned long long s_words[PAYLOAD_WORDS];
  __shared__ unsigned long long s_hdr;
  __shared__ int s_go;
  __shared__ double red[BLK / 64][HB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
  T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
  wave_mask_t present[P], valid[P];
  load_group<T>(xw, g, n, vw);
  if (MD::need_xc) load_group<T>(xc, g, n, vc);
  if (MD::m23) load_group<T>(bv, g, n, vb);
  if (MD::mnn) { load_group<T>(nw, g, n, vnw); load_group<T>(nc, g, n, vnc); }
#pragma unroll
  for (int i = 0; i < P; i++) {
    const bool here = (g * P + i) < n;
    present[i] = __builtin_amdgcn_ballot_w64(here & (!MD::m23 || !(vb[3 * i] != vb[3 * i] || vb[3 * i + 1] != vb[3 * i + 1] || vb[3 * i + 2] != vb[3 * i + 2])));
    valid[i] = __builtin_amdgcn_ballot_w64(here & (!MD::need_xc || !all_nan(vc[3 * i], vc[3 * i + 1], vc[3 * i + 2])));
  }
  const bool w23 = EXACT || !MD::m23 || any_bearing<P>(present);
  for (unsigned long long b = 1;; b++) {
    const unsigned long long want = first_tag + b;
    // ---- wait for batch b (both tags), bounded like the Gauss-Newton loops' wait for a pose
    if (threadIdx.x < 64) {
      const unsigned long long t0 = wall_clock64();
      int go = 0;
      unsigned long long w = 0;
      for (;;) {
        if (lane < 3) w = __hip_atomic_load(ctl + (lane == 0 ? 0 : (lane == 1 ? kSessionCtlWords - 1 : 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned int lo = (unsigned int)w, hi = (unsigned int)(w >> 32);
        const unsigned long long ta = ((unsigned long long)__builtin_amdgcn_readlane(hi, 0) << 32) | __builtin_amdgcn_readlane(lo, 0);
        const unsigned long long tb = ((unsigned long long)__builtin_amdgcn_readlane(hi, 1) << 32) | __builtin_amdgcn_readlane(lo, 1);
        if (ta == tb) {
          const unsigned long long num = ta & ~kResidentStop;
          if (num == want) { go = (ta & kResidentStop) ? 4 : 1; break; }   // (4: the session's last message)
          if (num > want) { go = 2; break; }   // a later call's tag: this launch is over
        }
        if (wall_clock64() - t0 > fin.pose_wait_ticks) { go = 3; break; }   // the host went away
        __builtin_amdgcn_s_sleep(2);
      }
      if (lane == 2) s_hdr = w;   // (read in the same sweep as the matching tags or a later one: the header was written before them)
      if (lane == 0) s_go = go;
    }
    __syncthreads();
    const int go = s_go;
    if (go != 1 && go != 4) return;
    // header and payload, now that the tags are known good (the header copy above may predate them): ONE round of loads -- thread 0
    // the header, threads 1 .. PAYLOAD_WORDS the largest payload a batch can have (what lies beyond this batch's is stale and unused)
    if (threadIdx.x == 0) s_hdr = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if ((int)threadIdx.x <= PAYLOAD_WORDS) s_words[threadIdx.x - 1] = __hip_atomic_load(ctl + 1 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    const int count = (int)(unsigned int)s_hdr, op = (int)(s_hdr >> 32);
    if (go == 4 && op != 1) return;   // a plain stop; with op 1: "write these masks, send their record, and leave"
    // a header that is neither "score" nor "masks", or a count beyond a batch (a torn or raced message): leave as a lost grid would --
    // the host recovers through its bounded wait and the launch path -- rather than write masks from a stale payload
    if (op < 0 || op > 1 || count < 0 || count > HB) return;
    const T* batch = reinterpret_cast<const T*>(s_words);
    double own = 0.0;
    if (op == 0) {
      int mine = 0;
      for (int h = 0; h < count; h++) {
        Hyp<T, EXACT> hyp;
        hyp.load(batch + h * STRIDE, KIND == VOTE_23_MATRIX);
        DeferQ<T> none;   // (no queue here: the in-place filter)
        const int cnt = w23 ? count_group_votes<T, KIND, EXACT, false, true>(hyp, vw, vc, vb, vnw, vnc, present, valid, thr33, cthr, cnl, none)
                            : count_group_votes<T, KIND, EXACT, false, false>(hyp, vw, vc, vb, vnw, vnc, present, valid, thr33, cthr, cnl, none);
        mine = (lane == h) ? cnt : mine;
      }
      if (lane < HB) red[wave][lane] = (double)mine;
      __syncthreads();
      if (threadIdx.x < HB) {
#pragma unroll
        for (int w = 0; w < BLK / 64; w++) own += red[w][threadIdx.x];
      }
    } else {
      // the masks of ONE hypothesis (the run's winner; fast layout: R row-major, t -- exact: quaternion, t) for this thread's group, and
      // the total of the votes in sum 0
      Hyp<T, EXACT> hyp;
      hyp.load(batch, KIND == VOTE_23_MATRIX);
      bool f23[P], f33[P], fnn[P];
      group_flags<T, KIND, EXACT>(hyp, vw, vc, vb, vnw, vnc, thr33, cthr, cnl, f23, f33, fnn);
      int cnt = 0;
      const int64_t full = n / P;
#pragma unroll
      for (int i = 0; i < P; i++) {
        const bool here = (g * P + i) < n;
        f23[i] = f23[i] & here; f33[i] = f33[i] & here; fnn[i] = fnn[i] & here;
        cnt += (int)fnn[i] + (int)f33[i] + (int)f23[i];
      }
      if (g < full) {
        if (MD::mnn) store_mask_full(mnn, g, fnn);
        if (MD::m33) store_mask_full(m33, g, f33);
        if (MD::m23) store_mask_full(m23, g, f23);
      } else {
#pragma unroll
        for (int i = 0; i < P; i++) {
          const int64_t idx = g * P + i;
          if (idx < n) { if (MD::mnn) mnn[idx] = fnn[i]; if (MD::m33) m33[idx] = f33[i]; if (MD::m23) m23[idx] = f23[i]; }
        }
      }
      const double wsum = wave_sum_to_lane63((double)cnt);
      if (lane < HB) red[wave][lane] = 0.0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane == 63) red[wave][0] = wsum;
      __syncthreads();
      if (threadIdx.x < HB) {
#pragma unroll
        for (int w = 0; w < BLK / 64; w++) own += red[w][threadIdx.x];
      }
    }
    if (!resident_cross_own<HB, BLK>(own, fin, want, fin.seq + b, false)) return;
    if (go == 4) return;
  }
}
template <class T, int KIND, bool EXACT>
static void score_launch(const DeviceArrays& A, const void* d_poses, int H, const double* thr, int* d_votes, int G, hipStream_t s) {
  // Split the hypothesis list over blockIdx.y (chunks of a multiple of 16 hypotheses) until there are ~8192 workgroups: the loop is a
  // chain of dependent vector and scalar instructions per wave, so many short waves per SIMD hide more of it than few long ones, and
  // a problem that is small for the machine (few tiles, or few hypotheses) fills it only through the split.  Against round 3 (chunks of
  // 64, 2048 workgroups): 512 hypotheses x 307 200 correspondences -9 %, 128 x 307 200 -19 %, 512 x 50 000 -25 %, 2048 x 307 200 and
  // 512 x 1 M unchanged (profiles/r04_score_grid_ab.txt).
  int gy = 1, hchunk = H;
  constexpr int target = 8192, gran = 16;
  if (G < target && H > gran) {
    const int chunks = (H + gran - 1) / gran;
    gy = (target + G - 1) / G;
    if (gy > chunks) gy = chunks;
    hchunk = ((chunks + gy - 1) / gy) * gran;
    gy = (H + hchunk - 1) / hchunk;
  }
  // (vote table, and behind it the deferred-exact queues of the fp32 exact kinds with a 2D test: one per wave)
  const bool queues = sizeof(T) == 4 && EXACT && VoteMods<KIND>::m23 && KIND != VOTE_23_MATRIX;
  const size_t lds = (((size_t)hchunk * sizeof(int) + 15) & ~(size_t)15) + (queues ? (size_t)(kBlock / 64) * kDeferEntries * 8 * sizeof(T) : 0);
  hipLaunchKernelGGL((score_kernel<T, KIND, EXACT>), dim3(G, gy), dim3(kBlock), lds, s, (const T*)A.a[0],
      (const T*)A.a[1],
                     (const T*)A.a[2], (const T*)A.a[3], (const T*)A.a[4], A.n, (const T*)d_poses, H, hchunk, (T)thr[0], (T)thr[1], (T)thr[2], d_votes);
}
template <class T, int KIND, bool EXACT>
static void mask_launch(const DeviceArrays& A, const double* pose12, const double* thr, const ReduceTarget& rt, hipStream_t s,
    hipEvent_t e0, hipEvent_t e1) {
  PoseArg<T> pa;
  for (int i = 0; i < 12; i++) pa.v[i] = (T)pose12[i];
  // As many workgroups as are RESIDENT at once, no more: the kernel needs 136-147 VGPRs, so three 256-thread workgroups fit a CU, and
  // a fourth per CU (the round-3 grid of 1024) ran as a second, one-third-full pass: 81-84 against 75-76 us at 10 M correspondences
  // for the 3D + 2D masks (profiles/r04_grid_sweep.txt).
  // Asked of the runtime once per instantiation.
  static const int resident = [] {
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)mask_kernel<T, KIND, EXACT>, kBlock, 0) != hipSuccess || per_cu < 1) {
      (void)hipGetLastError();
      per_cu = 3;
    }
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
      (void)hipGetLastError();
      cus = 256;
    }
    return per_cu * cus;
  }();
  const int cap = resident > 4096 ? 4096 : resident;   // one partial record per workgroup: the scratch holds 4096
  const int G = grid_for(A.n, Pk<T>::P, cap);
  RPE_LAUNCH_EV((mask_kernel<T, KIND, EXACT>), dim3(G), dim3(kBlock), 0, s, e0, e1, (const T*)A.a[0], (const T*)A.a[1], (const T*)A.a[2],
                     (const T*)A.a[3], (const T*)A.a[4], A.n, pa, (T)thr[0], (T)thr[1], (T)thr[2], A.mask[0], A.mask[1], A.mask[2],
                     make_finish(rt));
}
#define RPE_KIND_SWITCH(FN, T, EX, ...)                                    \
  switch (kind) {                                                          \
    case VOTE_33: FN<T, VOTE_33, EX>(__VA_ARGS__); break;                  \
    case VOTE_23: FN<T, VOTE_23, EX>(__VA_ARGS__); break;                  \
    case VOTE_33_23: FN<T, VOTE_33_23, EX>(__VA_ARGS__); break;            \
    case VOTE_NN_23: FN<T, VOTE_NN_23, EX>(__VA_ARGS__); break;            \
    case VOTE_NN_33: FN<T, VOTE_NN_33, EX>(__VA_ARGS__); break;            \
    case VOTE_NN_33_23: FN<T, VOTE_NN_33_23, EX>(__VA_ARGS__); break;      \
    case VOTE_23_MATRIX: FN<T, VOTE_23_MATRIX, EX>(__VA_ARGS__); break;    \
    default: return hipErrorInvalidValue;                                  \
  }

template <class T, int KIND, bool EXACT>
static void score_resident_launch(const DeviceArrays& A, const unsigned long long* ctl, unsigned long long first_tag, const double* thr, int G,
                                  const ReduceTarget& rt, hipStream_t s) {
  Finish fin = make_finish(rt);
  constexpr int kMaxRows = 4 * (512 / kSessionHyps);
  if (fin.rows > kMaxRows) fin.rows = kMaxRows;
  if (fin.rows < 1) fin.rows = 1;
  hipLaunchKernelGGL((score_resident_kernel<T, KIND, EXACT>), dim3(G), dim3(512), 0, s, (const T*)A.a[0], (const T*)A.a[1], (const T*)A.a[2],
                     (const T*)A.a[3], (const T*)A.a[4], A.n, ctl, first_tag, (T)thr[0], (T)thr[1], (T)thr[2], A.mask[0], A.mask[1], A.mask[2], fin);
}
// grid of the resident scoring kernel (one group per thread), or 0 if the problem is not frame-sized for this device
int score_resident_grid(const DeviceArrays& A, int max_blocks) {
  const int P = A.dtype ? 2 : 4;
  const int cap = std::max(1, resident_cap_device());
  const int64_t groups = (A.n + P - 1) / P;
  const int64_t G = (groups + 511) / 512;
  return G >= 1 && G <= std::min(cap, max_blocks) ? (int)G : 0;
}
hipError_t launch_score_resident(const DeviceArrays& A, int kind, int exact, const unsigned long long* ctl, unsigned long long first_tag,
                                 const double* thr3, int grid, const ReduceTarget& rt, hipStream_t s) {
  if (grid < 1) return hipErrorInvalidValue;
  if (A.dtype) {
    if (exact) { RPE_KIND_SWITCH(score_resident_launch, double, true, A, ctl, first_tag, thr3, grid, rt, s) }
    else { RPE_KIND_SWITCH(score_resident_launch, double, false, A, ctl, first_tag, thr3, grid, rt, s) }
  } else {
    if (exact) { RPE_KIND_SWITCH(score_resident_launch, float, true, A, ctl, first_tag, thr3, grid, rt, s) }
    else { RPE_KIND_SWITCH(score_resident_launch, float, false, A, ctl, first_tag, thr3, grid, rt, s) }
  }
  return hipGetLastError();
}

hipError_t launch_score(const DeviceArrays& A, int kind, int exact, const void* d_poses, int H, const double* thr3, int* d_votes,
                        int max_blocks, hipStream_t s) {
  if (H < 1 || H > kMaxScoreH) return hipErrorInvalidValue;
  // d_votes must be zero on entry: allocated zeroed, and re-zeroed by launch_publish_votes after every read-out
  if (A.dtype) {
    const int G = grid_for(A.n, 2, max_blocks);
    if (exact) { RPE_KIND_SWITCH(score_launch, double, true, A, d_poses, H, thr3, d_votes, G, s) }
    else { RPE_KIND_SWITCH(score_launch, double, false, A, d_poses, H, thr3, d_votes, G, s) }
  } else {
    const int G = grid_for(A.n, 4, max_blocks);
    if (exact) { RPE_KIND_SWITCH(score_launch, float, true, A, d_poses, H, thr3, d_votes, G, s) }
    else { RPE_KIND_SWITCH(score_launch, float, false, A, d_poses, H, thr3, d_votes, G, s) }
  }
  return hipGetLastError();
}
template <class T, int KIND, bool EXACT>
static void score_small_launch(const DeviceArrays& A, const void* h_poses, const void* d_poses, int H, const double* thr,
                               const ReduceTarget& rt, int cap, hipStream_t s) {
  constexpr int STRIDE = Hyp<T, EXACT>::STRIDE;
  const Finish fin = make_finish(rt);
  // up to a million correspondences the list is split over the 4 waves of a workgroup (RPE_SCORE_SPLIT = 1 | 2 | 4 overrides)
  static const int env_hs = getenv("RPE_SCORE_SPLIT") ? atoi(getenv("RPE_SCORE_SPLIT")) : 0;
  const int hs = (env_hs == 1 || env_hs == 2 || env_hs == 4) ? env_hs : (A.n <= (int64_t)1 << 20 ? 4 : 1);
  const int64_t tiles = ((A.n + Pk<T>::P - 1) / Pk<T>::P + kBlock / hs - 1) / (kBlock / hs);
  const int G = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, cap));
#define RPE_SMALL(HB)                                                                                                                        \
  do {                                                                                                                                       \
    SmallPoses<T, HB, STRIDE> sp;                                                                                                            \
    std::memset(&sp, 0, sizeof(sp));                                                                                                         \
    if (h_poses) std::memcpy(sp.v, h_poses, (size_t)H * STRIDE * sizeof(T));                                                                 \
    hipLaunchKernelGGL((score_small_kernel<T, KIND, EXACT, HB>), dim3(G), dim3(kBlock), 0, s, (const T*)A.a[0], (const T*)A.a[1], (const T*)A.a[2], \
                       (const T*)A.a[3], (const T*)A.a[4], A.n, sp, (const T*)d_poses, H, hs, (T)thr[0], (T)thr[1], (T)thr[2], fin);                                 \
  } while (0)
  if (H <= 16) RPE_SMALL(16); else RPE_SMALL(32);
#undef RPE_SMALL
}
// largest list the single-launch form takes for this dtype / mode (the hypotheses travel as a kernel argument of at most 2 KB)
int score_small_cap(int dtype, int exact) {
  const int bytes = (exact ? 8 : 12) * (dtype ? 8 : 4);
  return 32 * bytes <= 2048 ? 32 : 16;
}
// h_poses: H hypotheses staged in HOST memory in the scoring layout of `exact`, values of the array dtype (they travel in the kernel
// argument) -- or null and d_poses: the same list in HBM (a device-generated batch).  The vote counts arrive through rt (a collecting
// target): record[h] = votes of hypothesis h.
hipError_t launch_score_small(const DeviceArrays& A, int kind, int exact, const void* h_poses, const void* d_poses, int H,
                              const double* thr3, const ReduceTarget& rt, hipStream_t s) {
  if (H < 1 || H > score_small_cap(A.dtype, exact) || rt.rows < 1 || (!h_poses == !d_poses)) return hipErrorInvalidValue;
  const int cap = 2048;   // workgroups (grid-stride beyond)
  if (A.dtype) {
    if (exact) { RPE_KIND_SWITCH(score_small_launch, double, true, A, h_poses, d_poses, H, thr3, rt, cap, s) }
    else { RPE_KIND_SWITCH(score_small_launch, double, false, A, h_poses, d_poses, H, thr3, rt, cap, s) }
  } else {
    if (exact) { RPE_KIND_SWITCH(score_small_launch, float, true, A, h_poses, d_poses, H, thr3, rt, cap, s) }
    else { RPE_KIND_SWITCH(score_small_launch, float, false, A, h_poses, d_poses, H, thr3, rt, cap, s) }
  }
  return hipGetLastError();
}
hipError_t launch_mask(const DeviceArrays& A, int kind, int exact, const double* pose12, const double* thr3, const ReduceTarget& rt,
                       hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  if (A.dtype) {
    if (exact) { RPE_KIND_SWITCH(mask_launch, double, true, A, pose12, thr3, rt, s, e0, e1) }
    else { RPE_KIND_SWITCH(mask_launch, double, false, A, pose12, thr3, rt, s, e0, e1) }
  } else {
    if (exact) { RPE_KIND_SWITCH(mask_launch, float, true, A, pose12, thr3, rt, s, e0, e1) }
    else { RPE_KIND_SWITCH(mask_launch, float, false, A, pose12, thr3, rt, s, e0, e1) }
  }
  return hipGetLastError();
}

#ifdef RPE_SCORE_STATS
}  // namespace rpe
extern "C" int rpe_debug_read_score_stats(unsigned long long* out4) {   // diagnostic build only: read and clear the counters
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(rpe::g_score_stats), 32) != hipSuccess) return -1;
  static unsigned long long zeros[4];
  return hipMemcpyToSymbol(HIP_SYMBOL(rpe::g_score_stats), zeros, 32) == hipSuccess ? 0 : -1;
}
namespace rpe {
#endif
void preload_score() {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, (const void*)mask_kernel<float, VOTE_33, true>) != hipSuccess) (void)hipGetLastError();
}

}  // namespace rpe
