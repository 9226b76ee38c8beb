// Batched hypothesis generation on the device for the 3D-3D RANSAC solvers (shinji_ransac,
// shinji_ransac2 -- the core of ao_ransac): one thread per RANSAC iteration draws its minimal sample from the SAME random stream
// the host sampler would have used (PCG32 skip-ahead to its position; RandomElements' partial Fisher-Yates over the identity
// table), gathers the three correspondences from the HBM-resident arrays, and runs the closed-form fit shinji() for K = 3
// (pose/AbsoluteOrientation.hpp; reference :47-99) -- by compiling the very function the host path runs (rpe::rigid_fit<T>,
// rpe/linalg.hpp, __host__ __device__: T arithmetic in the reference's operation order, Jacobi SVD) with FMA contraction off and
// IEEE division / square root, so that every hypothesis is BITWISE the one the host would have produced and the
// sequential replay in pose/RansacEngine.hpp reaches the same pose, votes, Iter and mask.  The poses land in HBM in the scoring
// kernel's layout (no staging, no H2D copy) and, as quaternion + translation, in pinned host memory for the replay.
#pragma clang fp contract(off)
#include "../include/rpe/linalg.hpp"
#include "rpe_kernels.h"

namespace rpe {
namespace {

struct Pcg32 {   // rpe::Rand31 (pose/Utility.hpp): PCG32 XSH-RR, output >> 1
  unsigned long long state, inc;
  __device__ unsigned int step() {
    const unsigned long long old = state;
    state = old * 6364136223846793005ULL + inc;
    const unsigned int xorshifted = (unsigned int)(((old >> 18u) ^ old) >> 27u), rot = (unsigned int)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
  }
  __device__ int next31() { return (int)(step() >> 1); }
  __device__ void advance(unsigned long long delta) {   // LCG skip-ahead (Brown, "Random number generation with arbitrary strides")
    unsigned long long cur_mult = 6364136223846793005ULL, cur_plus = inc, acc_mult = 1ULL, acc_plus = 0ULL;
    while (delta > 0) {
      if (delta & 1ULL) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
      cur_plus = (cur_mult + 1ULL) * cur_plus;
      cur_mult *= cur_mult;
      delta >>= 1;
    }
    state = acc_mult * state + acc_plus;
  }
};

template <class T> struct Eps;
template <> struct Eps<float> { __device__ static float value() { return 1e-5f; } };      // rpe::LieEps (rpe/types.hpp)
template <> struct Eps<double> { __device__ static double value() { return 1e-10; } };

// out_pose: scoring layout (exact: qw qx qy qz tx ty tz 0 ; fast: R row-major 9, t 3) in T.  h_q7: 8 T per iteration in pinned host
// memory (qw qx qy qz tx ty tz valid).
template <class T>
__global__ __launch_bounds__(64) void gen_shinji_kernel(const T* __restrict__ xw, const T* __restrict__ xc, int n, unsigned long long state,
                             unsigned long long inc, int iters, int exact, T* __restrict__ out_pose, T* __restrict__ h_q7) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= iters) return;
  constexpr int K = 3;
  Pcg32 rng{state, inc};
  rng.advance((unsigned long long)K * (unsigned long long)i);
  // RandomElements::run(K): position j swaps with rnd() % (j + 1), j = n-1 ... n-K, over a table that starts as the identity
  int pos[2 * K], val[2 * K], cnt = 0, sel[K];
  auto get = [&](int p) { for (int k = 0; k < cnt; k++) if (pos[k] == p) return val[k]; return p; };
  auto set = [&](int p, int v) { for (int k = 0; k < cnt; k++) if (pos[k] == p) { val[k] = v; return; } pos[cnt] = p; val[cnt] = v;
      cnt++; };
  for (int s = 0, top = n - 1; s < K; s++, top--) {
    const int pick = rng.next31() % (top + 1);
    const int vp = get(pick), vt = get(top);
    set(pick, vt);
    set(top, vp);
    sel[s] = vp;
  }
  T X_w[3 * K], X_c[3 * K];   // 3 x K column-major, as the host's MatrixX
  bool valid = true;
  for (int s = 0; s < K; s++) {
    const T cx = xc[3 * (size_t)sel[s]], cy = xc[3 * (size_t)sel[s] + 1], cz = xc[3 * (size_t)sel[s] + 2];
    valid = valid && (cx == cx || cy == cy || cz == cz);   // isValid: not all three NaN
    X_c[3 * s] = cx; X_c[3 * s + 1] = cy; X_c[3 * s + 2] = cz;
    X_w[3 * s] = xw[3 * (size_t)sel[s]]; X_w[3 * s + 1] = xw[3 * (size_t)sel[s] + 1]; X_w[3 * s + 2] = xw[3 * (size_t)sel[s] + 2];
  }
  T q[4] = {T(1), T(0), T(0), T(0)}, t[3] = {T(0), T(0), T(0)};
  // shinji<T>(X_w, X_c, 3): the host's own function (rpe/linalg.hpp rigid_fit, T arithmetic in the reference's order).  A fit whose
  // rotation fails the SO3 constructor's test is skipped by the host solvers, so it is reported as not valid here too.
  if (valid) valid = rigid_fit<T>(X_w, X_c, K, K, Eps<T>::value(), q, t);
  T* hq = h_q7 + 8 * (size_t)i;
  hq[0] = q[0]; hq[1] = q[1]; hq[2] = q[2]; hq[3] = q[3]; hq[4] = t[0]; hq[5] = t[1]; hq[6] = t[2]; hq[7] = valid ? T(1) : T(0);
  if (exact) {
    T* o = out_pose + 8 * (size_t)i;
    o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3]; o[4] = t[0]; o[5] = t[1]; o[6] = t[2]; o[7] = T(0);
  } else {   // the host stages fast-mode poses as quat_to_R<double>(q) rounded to T
    const Quat<double> qd{(double)q[0], (double)q[1], (double)q[2], (double)q[3]};
    double Rd[9];
    quat_to_R<double>(qd, Rd);
    T* o = out_pose + 12 * (size_t)i;
    for (int k = 0; k < 9; k++) o[k] = (T)Rd[k];
    o[9] = t[0]; o[10] = t[1]; o[11] = t[2];
  }
}

// ---- P3P on the device (FAST scoring mode only: tolerance parity, not bit parity -- the host's kneip goes through std::complex pow /
// sqrt / division, which no device math library reproduces bit for bit).  Kneip, Scaramuzza, Siegwart, "A novel parametrization of the
// P3P problem" (CVPR 2011) as the reference evaluates it (pose/P3P.hpp:63-232; quartic by Ferrari with complex intermediates, :11-60),
// in fp64 whatever the array dtype.
struct Cx { double re, im; };
__device__ inline Cx cx(double r, double i = 0.0) { return Cx{r, i}; }
__device__ inline Cx operator+(Cx a, Cx b) { return Cx{a.re + b.re, a.im + b.im}; }
__device__ inline Cx operator-(Cx a, Cx b) { return Cx{a.re - b.re, a.im - b.im}; }
__device__ inline Cx operator-(Cx a) { return Cx{-a.re, -a.im}; }
__device__ inline Cx operator*(Cx a, Cx b) { return Cx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ inline Cx operator*(double s, Cx a) { return Cx{s * a.re, s * a.im}; }
__device__ inline Cx operator/(Cx a, Cx b) { const double d = b.re * b.re + b.im * b.im;
    return Cx{(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d}; }
__device__ inline Cx cpow(Cx a, double p) {   // principal branch
  const double r = hypot(a.re, a.im);
  if (r == 0.0) return Cx{0.0, 0.0};
  const double th = atan2(a.im, a.re), rp = pow(r, p);
  return Cx{rp * cos(p * th), rp * sin(p * th)};
}
__device__ inline Cx csqrt(Cx a) { return cpow(a, 0.5); }

// real parts of the four roots of a[0] x^4 + ... + a[4]
__device__ inline void o4_roots_dev(const double a[5], double roots[4]) {
  const double A = a[0], B = a[1], C = a[2], D = a[3], E = a[4];
  const double A2 = A * A, B2 = B * B, A3 = A2 * A, B3 = B2 * B, A4 = A3 * A, B4 = B3 * B;
  const double alpha = -3 * B2 / (8 * A2) + C / A;
  const double beta = B3 / (8 * A3) - B * C / (2 * A2) + D / A;
  const double gamma = -3 * B4 / (256 * A4) + B2 * C / (16 * A3) - B * D / (4 * A2) + E / A;
  const double alpha2 = alpha * alpha, alpha3 = alpha2 * alpha;
  const Cx P = cx(-alpha2 / 12 - gamma), Q = cx(-alpha3 / 108 + alpha * gamma / 3 - beta * beta / 8);
  const Cx R = -(0.5 * Q) + csqrt(0.25 * (Q * Q) + (1.0 / 27.0) * (P * P * P));
  const Cx U = cpow(R, 1.0 / 3.0);
  Cx y;
  if (U.re == 0) y = cx(-5.0 * alpha / 6.0) - cpow(Q, 1.0 / 3.0);
  else y = cx(-5.0 * alpha / 6.0) - P / (3.0 * U) + U;
  const Cx w = csqrt(cx(alpha) + 2.0 * y);
  const Cx up = csqrt(-(cx(3.0 * alpha) + 2.0 * y + cx(2.0 * beta) / w));
  const Cx um = csqrt(-(cx(3.0 * alpha) + 2.0 * y - cx(2.0 * beta) / w));
  const double shift = -B / (4.0 * A);
  roots[0] = shift + 0.5 * (w.re + up.re);
  roots[1] = shift + 0.5 * (w.re - up.re);
  roots[2] = shift + 0.5 * (-w.re + um.re);
  roots[3] = shift + 0.5 * (-w.re - um.re);
}

struct V3d { double x, y, z; };
__device__ inline V3d operator-(V3d a, V3d b) { return V3d{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline V3d operator+(V3d a, V3d b) { return V3d{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline V3d operator*(double s, V3d a) { return V3d{s * a.x, s * a.y, s * a.z}; }
__device__ inline double dot(V3d a, V3d b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline V3d cross(V3d a, V3d b) { return V3d{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ inline double norm(V3d a) { return sqrt(dot(a, a)); }
// matrix given by its rows
__device__ inline V3d mulr(const V3d rows[3], V3d v) { return V3d{dot(rows[0], v), dot(rows[1], v), dot(rows[2], v)}; }

// up to four (R row-major, t) with Xc = R Xw + t from three world points and their unit bearings; returns how many
__device__ inline int kneip_dev(const V3d Pw[3], const V3d bv[3], double rot_eps, double Rs[4][9], double ts[4][3]) {
  V3d P1 = Pw[0], P2 = Pw[1], P3 = Pw[2];
  const V3d edge12 = P2 - P1;
  if (norm(cross(edge12, P3 - P1)) == 0) return 0;   // collinear world points
  V3d f1 = bv[0], f2 = bv[1], f3 = bv[2];
  V3d Tc[3];
  auto camera_frame = [&]() {
    V3d e3 = cross(f1, f2);
    e3 = (1.0 / norm(e3)) * e3;
    const V3d e2 = cross(e3, f1);
    Tc[0] = f1; Tc[1] = e2; Tc[2] = e3;
    f3 = mulr(Tc, f3);
  };
  camera_frame();
  if (f3.z > 0) {
    f1 = bv[1]; f2 = bv[0]; f3 = bv[2];
    camera_frame();
    P1 = Pw[1]; P2 = Pw[0]; P3 = Pw[2];
  }
  V3d n1 = P2 - P1;
  n1 = (1.0 / norm(n1)) * n1;
  V3d n3 = cross(n1, P3 - P1);
  n3 = (1.0 / norm(n3)) * n3;
  const V3d n2 = cross(n3, n1);
  const V3d Nw[3] = {n1, n2, n3};
  P3 = mulr(Nw, P3 - P1);
  const double d12 = norm(edge12);
  const double f_1 = f3.x / f3.z, f_2 = f3.y / f3.z, p_1 = P3.x, p_2 = P3.y;
  const double cos_beta = dot(f1, f2);
  double b = 1 / (1 - cos_beta * cos_beta) - 1;
  b = cos_beta < 0 ? -sqrt(b) : sqrt(b);
  const double f1s = f_1 * f_1, f2s = f_2 * f_2, p1s = p_1 * p_1, p1c = p1s * p_1, p1q = p1c * p_1, p2s = p_2 * p_2, p2c = p2s * p_2, p2q = p2c * p_2;
  const double ds = d12 * d12, bs = b * b;
  double q[5];
  q[0] = -f2s * p2q - p2q * f1s - p2q;
  q[1] = 2 * p2c * d12 * b + 2 * f2s * p2c * d12 * b - 2 * f_2 * p2c * f_1 * d12;
  q[2] = -f2s * p2s * p1s - f2s * p2s * ds * bs - f2s * p2s * ds + f2s * p2q + p2q * f1s + 2 * p_1 * p2s * d12 +
         2 * f_1 * f_2 * p_1 * p2s * d12 * b - p2s * p1s * f1s + 2 * p_1 * p2s * f2s * d12 - p2s * ds * bs - 2 * p1s * p2s;
  q[3] = 2 * p1s * p_2 * d12 * b + 2 * f_2 * p2c * f_1 * d12 - 2 * f2s * p2c * d12 * b - 2 * p_1 * p_2 * ds * b;
  q[4] = -2 * f_2 * p2s * f_1 * p_1 * d12 * b + f2s * p2s * ds + 2 * p1c * d12 - p1s * ds + f2s * p2s * p1s - p1q -
         2 * f2s * p2s * p_1 * d12 + p2s * f1s * p1s + f2s * p2s * ds * bs;
  double roots[4];
  o4_roots_dev(q, roots);
  int count = 0;
  for (int i = 0; i < 4; i++) {
    const double cos_theta = roots[i];
    if (cos_theta != cos_theta || cos_theta > 1.0 || cos_theta < -1.0) continue;
    const double cot_alpha = (-f_1 * p_1 / f_2 - cos_theta * p_2 + d12 * b) / (-f_1 * cos_theta * p_2 / f_2 + p_1 - d12);
    const double sin_theta = sqrt(1 - cos_theta * cos_theta);
    const double sin_alpha = sqrt(1 / (cot_alpha * cot_alpha + 1));
    double cos_alpha = sqrt(1 - sin_alpha * sin_alpha);
    if (cot_alpha < 0) cos_alpha = -cos_alpha;
    const double k = d12 * (sin_alpha * b + cos_alpha);
    const V3d Cl{k * cos_alpha, cos_theta * k * sin_alpha, sin_theta * k * sin_alpha};
    // camera centre in the world frame: P1 + Nw^T Cl
    const V3d Cw = P1 + (Cl.x * n1 + Cl.y * n2) + Cl.z * n3;
    const double Q[9] = {-cos_alpha, -sin_alpha * cos_theta, -sin_alpha * sin_theta, sin_alpha, -cos_alpha * cos_theta, -cos_alpha * sin_theta,
                         0.0, -sin_theta, cos_theta};
    // R = Tc^T Q Nw
    const double N9[9] = {n1.x, n1.y, n1.z, n2.x, n2.y, n2.z, n3.x, n3.y, n3.z};
    const double T9[9] = {Tc[0].x, Tc[1].x, Tc[2].x, Tc[0].y, Tc[1].y, Tc[2].y, Tc[0].z, Tc[1].z, Tc[2].z};   // Tc^T
    double QN[9], R[9];
    mat3_mul(Q, N9, QN);
    mat3_mul(T9, QN, R);
    // the SO3 constructor's test, with the array dtype's tolerance (bearings are unit to that precision)
    if (R[0] != R[0] || !is_rotation<double>(R, rot_eps)) continue;
    for (int e = 0; e < 9; e++) Rs[count][e] = R[e];
    ts[count][0] = -(R[0] * Cw.x + R[1] * Cw.y + R[2] * Cw.z);
    ts[count][1] = -(R[3] * Cw.x + R[4] * Cw.y + R[5] * Cw.z);
    ts[count][2] = -(R[6] * Cw.x + R[7] * Cw.y + R[8] * Cw.z);
    count++;
  }
  return count;
}

// rotation by `angle` about the unit `axis` (Rodrigues), row-major
__device__ inline void angle_axis(double angle, V3d k, double R[9]) {
  const double c = cos(angle), s = sin(angle), v = 1.0 - c;
  R[0] = c + k.x * k.x * v;       R[1] = k.x * k.y * v - k.z * s; R[2] = k.x * k.z * v + k.y * s;
  R[3] = k.y * k.x * v + k.z * s; R[4] = c + k.y * k.y * v;       R[5] = k.y * k.z * v - k.x * s;
  R[6] = k.z * k.x * v - k.y * s; R[7] = k.z * k.y * v + k.x * s; R[8] = c + k.z * k.z * v;
}
// the 2-point + normal solver (pose/AbsoluteOrientationNormal.hpp:77-142): align the normals with the x axis in both frames, then the
// in-plane directions of the second point (the unsigned angle the reference takes, acos of their dot product)
__device__ inline void nl_2p_dev(V3d pt1_c, V3d nl1_c, V3d pt2_c, V3d pt1_w, V3d nl1_w, V3d pt2_w, double R[9], double t[3]) {
  auto to_x_axis = [](V3d nv, double Rx[9]) {
    V3d axis{0.0, nv.z, -nv.y};
    axis = (1.0 / norm(axis)) * axis;
    angle_axis(acos(nv.x), axis, Rx);
  };
  double Rw[9], Rc[9], Rp[9];
  to_x_axis(nl1_w, Rw);
  to_x_axis(nl1_c, Rc);
  auto apply = [](const double M[9], V3d v) { return V3d{M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z}; };
  V3d a = apply(Rw, pt2_w - pt1_w); a.x = 0.0; a = (1.0 / norm(a)) * a;
  V3d b = apply(Rc, pt2_c - pt1_c); b.x = 0.0; b = (1.0 / norm(b)) * b;
  angle_axis(acos(dot(a, b)), V3d{1.0, 0.0, 0.0}, Rp);
  const double RcT[9] = {Rc[0], Rc[3], Rc[6], Rc[1], Rc[4], Rc[7], Rc[2], Rc[5], Rc[8]};
  double PW[9];
  mat3_mul(Rp, Rw, PW);
  mat3_mul(RcT, PW, R);
  const V3d rp = apply(R, pt1_w);
  t[0] = pt1_c.x - rp.x; t[1] = pt1_c.y - rp.y; t[2] = pt1_c.z - rp.z;
}

// One thread per RANSAC iteration of a plain-RANSAC solver with a 4-point sample.  Slots per iteration, in the order in which the
// reference scores an iteration's hypotheses:
//   solver 0 kneip_ransac: P3P | 1 shinji_kneip_ransac: 3-point fit, P3P | 2 nl_kneip_ransac: P3P | 3 nl_shinji_ransac: 3-point fit,
//   nl_2p | 4 nl_shinji_kneip_ransac: 3-point fit, P3P, nl_2p.
// The sample is the host sampler's (4 draws per iteration from the same PCG32 stream); the P3P branch is the one that best reprojects
// the 4th correspondence (P3P.hpp:250-294 / :341-360); nl_2p runs on whatever the first two sampled columns hold, valid or not, and
// always yields a hypothesis (as the reference, AbsoluteOrientationNormal.hpp:315,389).  Poses go to HBM in the FAST scoring layout
// and, as quaternion + translation + valid flag, to pinned host memory for the replay.
template <class T>
__global__ __launch_bounds__(64) void gen_p3p_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const T* __restrict__ bvp,
                                                     const T* __restrict__ nwp, const T* __restrict__ ncp, int n, int solver,
                                                     unsigned long long state, unsigned long long inc, int iters, T* __restrict__ out_pose,
                                                     T* __restrict__ h_q7) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= iters) return;
  constexpr int K = 4;
  Pcg32 rng{state, inc};
  rng.advance((unsigned long long)K * (unsigned long long)i);
  int pos[2 * K], val[2 * K], cnt = 0, sel[K];
  auto get = [&](int p) { for (int k = 0; k < cnt; k++) if (pos[k] == p) return val[k]; return p; };
  auto set = [&](int p, int v) { for (int k = 0; k < cnt; k++) if (pos[k] == p) { val[k] = v; return; } pos[cnt] = p; val[cnt] = v;
      cnt++; };
  for (int s = 0, top = n - 1; s < K; s++, top--) {
    const int pick = rng.next31() % (top + 1);
    const int vp = get(pick), vt = get(top);
    set(pick, vt);
    set(top, vp);
    sel[s] = vp;
  }
  const bool has_fit = solver == 1 || solver == 3 || solver == 4, has_p3p = solver != 3, has_nl = solver == 3 || solver == 4;
  const int slots = (has_fit ? 1 : 0) + (has_p3p ? 1 : 0) + (has_nl ? 1 : 0);
  int slot_next = 0;
  auto put = [&](int slot, const double R[9], const double t[3], bool valid) {
    const size_t o = (size_t)slots * i + slot;
    T* op = out_pose + 12 * o;
    for (int k = 0; k < 9; k++) op[k] = (T)R[k];
    op[9] = (T)t[0]; op[10] = (T)t[1]; op[11] = (T)t[2];
    T Rt[9];
    for (int k = 0; k < 9; k++) Rt[k] = (T)R[k];
    const Quat<T> q = quat_from_R<T>(Rt);
    T* hq = h_q7 + 8 * o;
    hq[0] = q.w; hq[1] = q.x; hq[2] = q.y; hq[3] = q.z; hq[4] = (T)t[0]; hq[5] = (T)t[1]; hq[6] = (T)t[2]; hq[7] = valid ? T(1) : T(0);
  };
  const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3[3] = {0, 0, 0};
  if (has_fit) {   // the 3-point closed form on the first three correspondences (needs valid camera points)
    T X_w[9], X_c[9];
    bool valid = true;
    for (int s = 0; s < 3; s++) {
      const T cx_ = xc[3 * (size_t)sel[s]], cy_ = xc[3 * (size_t)sel[s] + 1], cz_ = xc[3 * (size_t)sel[s] + 2];
      valid = valid && (cx_ == cx_ || cy_ == cy_ || cz_ == cz_);
      X_c[3 * s] = cx_; X_c[3 * s + 1] = cy_; X_c[3 * s + 2] = cz_;
      X_w[3 * s] = xw[3 * (size_t)sel[s]]; X_w[3 * s + 1] = xw[3 * (size_t)sel[s] + 1]; X_w[3 * s + 2] = xw[3 * (size_t)sel[s] + 2];
    }
    T q[4] = {T(1), T(0), T(0), T(0)}, t[3] = {T(0), T(0), T(0)};
    if (valid) valid = rigid_fit<T>(X_w, X_c, 3, 3, Eps<T>::value(), q, t);
    double Rd[9];
    const Quat<double> qd{(double)q[0], (double)q[1], (double)q[2], (double)q[3]};
    quat_to_R<double>(qd, Rd);
    const double td[3] = {(double)t[0], (double)t[1], (double)t[2]};
    put(slot_next++, Rd, td, valid);
  }
  V3d Pw[4];
  for (int s = 0; s < 4; s++) Pw[s] = V3d{(double)xw[3 * (size_t)sel[s]], (double)xw[3 * (size_t)sel[s] + 1], (double)xw[3 * (size_t)sel[s] + 2]};
  if (has_p3p) {
  V3d bv[4];
  for (int s = 0; s < 4; s++) bv[s] = V3d{(double)bvp[3 * (size_t)sel[s]], (double)bvp[3 * (size_t)sel[s] + 1], (double)bvp[3 * (size_t)sel[s] + 2]};
  double Rs[4][9], ts[4][3];
  const int found = kneip_dev(Pw, bv, (double)Eps<T>::value(), Rs, ts);
  double best = 1e300;
  int arg = -1;
  for (int k = 0; k < found; k++) {
    V3d pc{Rs[k][0] * Pw[3].x + Rs[k][1] * Pw[3].y + Rs[k][2] * Pw[3].z + ts[k][0], Rs[k][3] * Pw[3].x + Rs[k][4] * Pw[3].y + Rs[k][5] * Pw[3].z + ts[k][1],
           Rs[k][6] * Pw[3].x + Rs[k][7] * Pw[3].y + Rs[k][8] * Pw[3].z + ts[k][2]};
    pc = (1.0 / norm(pc)) * pc;
    const double score = 1.0 - dot(pc, bv[3]);
    if (score < best) { best = score; arg = k; }
  }
  if (arg >= 0) put(slot_next++, Rs[arg], ts[arg], true);
  else put(slot_next++, I9, z3, false);
  }
  if (has_nl) {
    auto col = [&](const T* a, int s) { return V3d{(double)a[3 * (size_t)sel[s]], (double)a[3 * (size_t)sel[s] + 1], (double)a[3 * (size_t)sel[s] + 2]}; };
    double R[9], t[3];
    nl_2p_dev(col(xc, 0), col(ncp, 0), col(xc, 1), Pw[0], col(nwp, 0), Pw[1], R, t);
    put(slot_next++, R, t, true);   // a degenerate sample gives a NaN pose, which scores no votes -- it still counts as a hypothesis
  }
}

}  // namespace

// slots per iteration of a solver (0 kneip, 1 shinji + kneip, 2 nl_kneip, 3 nl_shinji, 4 nl_shinji_kneip); 0 for an unknown solver
int gen_p3p_slots(int solver) {
  switch (solver) { case 0: case 2: return 1; case 1: case 3: return 2; case 4: return 3; default: return 0; }
}
hipError_t launch_gen_p3p(const DeviceArrays& A, int solver, unsigned long long state, unsigned long long inc, int iters, void* d_poses,
                          void* h_q7, hipStream_t s) {
  if (iters < 1) return hipSuccess;
  if (gen_p3p_slots(solver) == 0) return hipErrorInvalidValue;
  const int G = (iters + 63) / 64;
  if (A.dtype) hipLaunchKernelGGL(gen_p3p_kernel<double>, dim3(G), dim3(64), 0, s, (const double*)A.a[0], (const double*)A.a[1],
      (const double*)A.a[2], (const double*)A.a[3], (const double*)A.a[4], (int)A.n, solver, state, inc, iters, (double*)d_poses, (double*)h_q7);
  else hipLaunchKernelGGL(gen_p3p_kernel<float>, dim3(G), dim3(64), 0, s, (const float*)A.a[0], (const float*)A.a[1],
      (const float*)A.a[2], (const float*)A.a[3], (const float*)A.a[4], (int)A.n, solver, state, inc, iters, (float*)d_poses, (float*)h_q7);
  return hipGetLastError();
}

hipError_t launch_gen_shinji(const DeviceArrays& A, unsigned long long state, unsigned long long inc, int iters, int exact, void* d_poses,
                             void* h_q7, hipStream_t s) {
  if (iters < 1) return hipSuccess;
  const int G = (iters + 63) / 64;
  if (A.dtype) hipLaunchKernelGGL(gen_shinji_kernel<double>, dim3(G), dim3(64), 0, s, (const double*)A.a[0], (const double*)A.a[1],
      (int)A.n, state, inc, iters, exact, (double*)d_poses, (double*)h_q7);
  else hipLaunchKernelGGL(gen_shinji_kernel<float>, dim3(G), dim3(64), 0, s, (const float*)A.a[0], (const float*)A.a[1], (int)A.n,
      state, inc, iters, exact, (float*)d_poses, (float*)h_q7);
  return hipGetLastError();
}

void preload_hypotheses() {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, (const void*)gen_shinji_kernel<float>) != hipSuccess) (void)hipGetLastError();
}

}  // namespace rpe
