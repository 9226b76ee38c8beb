// rpe/linalg.hpp -- Eigen-free small-matrix and Lie-group code of the product's HOST side.
//
// The reference leans on Eigen3 (not vendored, absent from this image) and a vendored Sophus for its
// O(1) algebra: 3x3 SVD (pose/AbsoluteOrientation.hpp:79), quaternion <-> matrix and q*v
// (sophus/so3.hpp:204-240,561-585), SE3 exp (sophus/se3.hpp:321-342).  This header provides that slice on
// plain arrays.  Everything on the device-facing boundary is double; Tp-typed shims live in pose/*.hpp.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>
#include <limits>
#include <algorithm>

// the small solves below also run on the GPU (device-resident Gauss-Newton loop, last workgroup of the kernel)
#if defined(__HIPCC__)
#define RPE_HD __host__ __device__
#else
#define RPE_HD
#endif

namespace rpe {

struct Vec3d {
  double v[3];
  RPE_HD Vec3d() : v{0, 0, 0} {}
  RPE_HD Vec3d(double a, double b, double c) : v{a, b, c} {}
  RPE_HD double& operator[](int i) { return v[i]; }
  RPE_HD double operator[](int i) const { return v[i]; }
};
RPE_HD inline Vec3d operator+(const Vec3d& a, const Vec3d& b) { return Vec3d(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
RPE_HD inline Vec3d operator-(const Vec3d& a, const Vec3d& b) { return Vec3d(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
RPE_HD inline Vec3d operator*(double s, const Vec3d& a) { return Vec3d(s * a[0], s * a[1], s * a[2]); }
RPE_HD inline double dot3(const Vec3d& a, const Vec3d& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
RPE_HD inline Vec3d cross3(const Vec3d& a, const Vec3d& b) {
  return Vec3d(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
RPE_HD inline double norm3(const Vec3d& a) { return std::sqrt(dot3(a, a)); }

// row-major 3x3
struct Mat3d {
  double a[9];
  RPE_HD Mat3d() { for (int i = 0; i < 9; i++) a[i] = 0.0; }
  RPE_HD static Mat3d eye() { Mat3d m; m.a[0] = m.a[4] = m.a[8] = 1.0; return m; }
  RPE_HD double& operator()(int r, int c) { return a[3 * r + c]; }
  RPE_HD double operator()(int r, int c) const { return a[3 * r + c]; }
  RPE_HD Vec3d col(int c) const { return Vec3d(a[c], a[3 + c], a[6 + c]); }
  RPE_HD void setcol(int c, const Vec3d& v) { a[c] = v[0]; a[3 + c] = v[1]; a[6 + c] = v[2]; }
};
RPE_HD inline Mat3d mul(const Mat3d& x, const Mat3d& y) {
  Mat3d r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = x(i, 0) * y(0, j) + x(i, 1) * y(1, j) + x(i, 2) * y(2, j);
  return r;
}
RPE_HD inline Vec3d mul(const Mat3d& x, const Vec3d& v) {
  return Vec3d(x(0, 0) * v[0] + x(0, 1) * v[1] + x(0, 2) * v[2], x(1, 0) * v[0] + x(1, 1) * v[1] + x(1, 2) * v[2],
               x(2, 0) * v[0] + x(2, 1) * v[1] + x(2, 2) * v[2]);
}
RPE_HD inline Mat3d transposed(const Mat3d& x) { Mat3d r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = x(j, i);
    return r; }
RPE_HD inline double det3(const Mat3d& m) {
  return m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
         m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
}

// ---- 3x3 SVD in T = float / double, host and device: the two-sided Jacobi iteration that Eigen::JacobiSVD runs on a square
// matrix (the reference's only SVD: JacobiSVD<Matrix<Tp,-1,-1>>(M, ComputeFullU | ComputeFullV), pose/AbsoluteOrientation.hpp:79,
// pose/AbsoluteOrientationNormal.hpp:44,188,512; Eigen 3.3: SVD/JacobiSVD.h, Jacobi/Jacobi.h).  Following that published algorithm
// step for step -- and not a mathematically equivalent one -- is what makes a minimal-sample hypothesis here the SAME Tp values the
// reference's arithmetic yields, so that RANSAC consensus sets can be compared exactly and not "up to rounding at the threshold":
//   W = A / max|a_ij| ; U = V = I ; repeat sweeps over (p, q) = (1,0) (2,0) (2,1) while some |W_pq| or |W_qp| exceeds
//   max(min_normal, 2 eps max_k |W_kk|):  G1 symmetrises the 2x2 block, J diagonalises the symmetric block (Jacobi),
//   W <- (G1 J^T) W J ; U <- U (G1 J^T)^T ; V <- V J ;  finally s_k = |W_kk| scale, sign into U's column, descending by swaps.
// Matrices are ROW-major 9-arrays.  U and V are full orthogonal bases whatever the rank (they start from I).
template <class T> struct SvdJ { T U[9], V[9], s[3]; };
template <class T> struct Givens { T c, s; };   // [c s; -s c]
template <class T> RPE_HD inline T absv(T x) { return std::fabs(x); }
template <class T> RPE_HD inline T maxv(T a, T b) { return a < b ? b : a; }   // std::max(a, b)
// rows p, q:  x' = c x + s y ; y' = -s x + c y
template <class T> RPE_HD inline void givens_rows(T* M, int p, int q, const Givens<T>& g) {
  for (int k = 0; k < 3; k++) { const T x = M[3 * p + k], y = M[3 * q + k]; M[3 * p + k] = g.c * x + g.s * y;
      M[3 * q + k] = -g.s * x + g.c * y; }
}
// columns p, q:  x' = c x - s y ; y' = s x + c y
template <class T> RPE_HD inline void givens_cols(T* M, int p, int q, const Givens<T>& g) {
  for (int k = 0; k < 3; k++) { const T x = M[3 * k + p], y = M[3 * k + q]; M[3 * k + p] = g.c * x - g.s * y;
      M[3 * k + q] = g.s * x + g.c * y; }
}
template <class T> RPE_HD inline SvdJ<T> jacobi_svd3(const T* A) {
  const T tiny = std::numeric_limits<T>::min(), prec = T(2) * std::numeric_limits<T>::epsilon();
  T scale = T(0);
  for (int i = 0; i < 9; i++) scale = maxv(scale, absv(A[i]));
  if (scale == T(0)) scale = T(1);
  SvdJ<T> r;
  T W[9];
  for (int i = 0; i < 9; i++) { W[i] = A[i] / scale; r.U[i] = r.V[i] = (i % 4 == 0) ? T(1) : T(0); }
  T big = maxv(absv(W[0]), maxv(absv(W[4]), absv(W[8])));
  bool done = false;
  for (int sweep = 0; !done && sweep < 1000; sweep++) {
    done = true;
    for (int p = 1; p < 3; p++) for (int q = 0; q < p; q++) {
      const T thr = maxv(tiny, prec * big);
      if (!(absv(W[3 * p + q]) > thr || absv(W[3 * q + p]) > thr)) continue;
      done = false;
      // 2x2 block [a b; c d] = rows/cols (p, q)
      T a = W[4 * p], b = W[3 * p + q], c = W[3 * q + p], d = W[4 * q];
      Givens<T> g1;
      const T tr = a + d, df = c - b;
      if (absv(df) < tiny) { g1.s = T(0); g1.c = T(1); }
      else { const T u = tr / df, h = std::sqrt(T(1) + u * u); g1.s = T(1) / h; g1.c = u / h; }
      { const T a0 = a, b0 = b, c0 = c, d0 = d; a = g1.c * a0 + g1.s * c0; b = g1.c * b0 + g1.s * d0; c = -g1.s * a0 + g1.c * c0;
          d = -g1.s * b0 + g1.c * d0; }
      (void)c;
      Givens<T> jr;   // Jacobi rotation of the now symmetric block [a b; b d]
      const T two_b = T(2) * absv(b);
      if (two_b < tiny) { jr.c = T(1); jr.s = T(0); }
      else {
        const T tau = (a - d) / two_b, w = std::sqrt(tau * tau + T(1));
        const T t = tau > T(0) ? T(1) / (tau + w) : T(1) / (tau - w);
        const T sgn = t > T(0) ? T(1) : T(-1), n = T(1) / std::sqrt(t * t + T(1));
        jr.s = -sgn * (b / absv(b)) * absv(t) * n;
        jr.c = n;
      }
      const Givens<T> jl{g1.c * jr.c - g1.s * (-jr.s), g1.c * (-jr.s) + g1.s * jr.c};   // g1 * jr^T
      givens_rows(W, p, q, jl);
      givens_cols(r.U, p, q, Givens<T>{jl.c, -jl.s});
      givens_cols(W, p, q, jr);
      givens_cols(r.V, p, q, jr);
      big = maxv(big, maxv(absv(W[4 * p]), absv(W[4 * q])));
    }
  }
  for (int k = 0; k < 3; k++) {
    const T m = absv(W[4 * k]);
    r.s[k] = m;
    if (m != T(0)) { const T sg = W[4 * k] / m; for (int i = 0; i < 3; i++) r.U[3 * i + k] *= sg; }
  }
  for (int k = 0; k < 3; k++) r.s[k] *= scale;
  for (int k = 0; k < 3; k++) {
    int best = k;
    for (int j = k + 1; j < 3; j++) if (r.s[j] > r.s[best]) best = j;
    if (r.s[best] == T(0)) break;
    if (best != k) {
      const T ts = r.s[k]; r.s[k] = r.s[best]; r.s[best] = ts;
      for (int i = 0; i < 3; i++) {
        const T tu = r.U[3 * i + k]; r.U[3 * i + k] = r.U[3 * i + best]; r.U[3 * i + best] = tu;
        const T tv = r.V[3 * i + k]; r.V[3 * i + k] = r.V[3 * i + best]; r.V[3 * i + best] = tv;
      }
    }
  }
  return r;
}

// A = U diag(s) V^T, s descending, U and V orthogonal (double; the O(N) least-squares paths)
struct Svd3 { Mat3d U, V; double s[3]; };
RPE_HD inline Svd3 svd3(const Mat3d& A) {
  const SvdJ<double> d = jacobi_svd3<double>(A.a);
  Svd3 r;
  for (int i = 0; i < 9; i++) { r.U.a[i] = d.U[i]; r.V.a[i] = d.V[i]; }
  for (int k = 0; k < 3; k++) r.s[k] = d.s[k];
  return r;
}
RPE_HD inline Vec3d svd_solve3(const Mat3d& A, const Vec3d& b) {
  Svd3 d = svd3(A);
  Vec3d y = mul(transposed(d.U), b);
  const double thr = 3.0 * std::numeric_limits<double>::epsilon() * d.s[0];
  for (int k = 0; k < 3; k++) y[k] = d.s[k] > thr ? y[k] / d.s[k] : 0.0;
  return mul(d.V, y);
}

// Rotation closest to the cross-covariance M (Kabsch / Umeyama without scale): R = U diag(1,1,det(UV^T)) V^T
RPE_HD inline Mat3d rotation_from_covariance(const Mat3d& M) {
  Svd3 d = svd3(M);
  Mat3d UVt = mul(d.U, transposed(d.V));
  if (det3(UVt) < 0) {
    Mat3d Uf = d.U;
    for (int i = 0; i < 3; i++) Uf(i, 2) = -Uf(i, 2);
    return mul(Uf, transposed(d.V));
  }
  return UVt;
}

// ---- quaternions (w, x, y, z), templated so that Tp-typed adapters can hold a Sophus-like SO3<Tp>
template <class T> struct Quat { T w, x, y, z; };
template <class T> RPE_HD Quat<T> quat_from_R(const T* R /*row-major*/) {
  Quat<T> q;
  const T tr = R[0] + R[4] + R[8];
  if (tr > T(0)) {
    T s = std::sqrt(tr + T(1));
    q.w = T(0.5) * s; s = T(0.5) / s;
    q.x = (R[7] - R[5]) * s; q.y = (R[2] - R[6]) * s; q.z = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    T s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + T(1));
    T v[3];
    v[i] = T(0.5) * s; s = T(0.5) / s;
    q.w = (R[3 * k + j] - R[3 * j + k]) * s;
    v[j] = (R[3 * j + i] + R[3 * i + j]) * s;
    v[k] = (R[3 * k + i] + R[3 * i + k]) * s;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}
template <class T> RPE_HD void quat_to_R(const Quat<T>& q, T* R /*row-major*/) {
  const T tx = T(2) * q.x, ty = T(2) * q.y, tz = T(2) * q.z;
  const T twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const T tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = T(1) - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = T(1) - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = T(1) - (txx + tyy);
}
template <class T> RPE_HD void quat_rotate(const Quat<T>& q, const T* v, T* out) {
  T ux = q.y * v[2] - q.z * v[1], uy = q.z * v[0] - q.x * v[2], uz = q.x * v[1] - q.y * v[0];
  ux += ux; uy += uy; uz += uz;
  const T cx = q.y * uz - q.z * uy, cy = q.z * ux - q.x * uz, cz = q.x * uy - q.y * ux;
  out[0] = (v[0] + q.w * ux) + cx; out[1] = (v[1] + q.w * uy) + cy; out[2] = (v[2] + q.w * uz) + cz;
}
template <class T> RPE_HD Quat<T> quat_mul(const Quat<T>& a, const Quat<T>& b) {
  return Quat<T>{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                 a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}

// ---- closed-form rigid fit Xc ~ R Xw + t on K columns (Umeyama 1991 without scale: the reference's shinji(),
// pose/AbsoluteOrientation.hpp:47-99), every operation in T and in the reference's order: column sums / K, centred outer products
// summed column by column, the sum divided by `cols` (the reference divides by X_w_.cols(), :75 -- equal to K at every call site),
// Jacobi SVD, R = U V^T, or U diag(1,1,-1) V^T when det(U V^T) < 0, quaternion taken from R without renormalising (what
// Sophus::SO3(Matrix3) does, sophus/so3.hpp:561-566), t = Cc - q * Cw with the quaternion rotation.  Returns false where
// SOPHUS_ENSURE would have aborted the reference (R not orthogonal to LieEps, or det(R) <= 0); q and t are filled either way.
// Xw, Xc: 3 x K column-major.  Host and device (the batched 3-point generator of csrc/rpe_hypotheses.hip runs this very function).
template <class T> RPE_HD inline void mat3_mul(const T* X, const T* Y, T* Z) {   // row-major, terms added in column order
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Z[3 * i + j] = X[3 * i] * Y[j] + X[3 * i + 1] * Y[3 + j] + X[3 * i + 2] * Y[6 + j];
}
template <class T> RPE_HD inline T mat3_det(const T* a) {
  return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}
// |R R^T - I|_F < eps and det R > 0: the two SOPHUS_ENSUREs of SO3(Matrix3) (sophus/rotation_matrix.hpp:13-24)
template <class T> RPE_HD inline bool is_rotation(const T* R, T eps) {
  T Rt[9], E[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[3 * i + j] = R[3 * j + i];
  mat3_mul(R, Rt, E);
  T f = T(0);
  for (int i = 0; i < 9; i++) { const T d = E[i] - ((i % 4 == 0) ? T(1) : T(0)); f += d * d; }
  return (std::sqrt(f) < eps) && (mat3_det(R) > T(0));
}
template <class T> RPE_HD inline bool rigid_fit(const T* Xw, const T* Xc, int K, int cols, T eps, T q[4], T t[3]) {
  T Cw[3] = {T(0), T(0), T(0)}, Cc[3] = {T(0), T(0), T(0)};
  for (int n = 0; n < K; n++) for (int k = 0; k < 3; k++) { Cw[k] += Xw[3 * n + k]; Cc[k] += Xc[3 * n + k]; }
  for (int k = 0; k < 3; k++) { Cw[k] /= (T)K; Cc[k] /= (T)K; }
  T M[9];
  for (int i = 0; i < 9; i++) M[i] = T(0);
  for (int n = 0; n < K; n++) {
    T Aw[3], Ac[3];
    for (int k = 0; k < 3; k++) { Aw[k] = Xw[3 * n + k] - Cw[k]; Ac[k] = Xc[3 * n + k] - Cc[k]; }
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[3 * r + c] += Ac[r] * Aw[c];
  }
  for (int i = 0; i < 9; i++) M[i] /= (T)cols;
  const SvdJ<T> d = jacobi_svd3<T>(M);
  T Vt[9], R[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Vt[3 * i + j] = d.V[3 * j + i];
  mat3_mul(d.U, Vt, R);
  if (mat3_det(R) < T(0)) {
    const T F[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(-1)};
    T UF[9];
    mat3_mul(d.U, F, UF);
    mat3_mul(UF, Vt, R);
  }
  const Quat<T> qq = quat_from_R<T>(R);
  q[0] = qq.w; q[1] = qq.x; q[2] = qq.y; q[3] = qq.z;
  T rc[3];
  quat_rotate<T>(qq, Cw, rc);
  for (int k = 0; k < 3; k++) t[k] = Cc[k] - rc[k];
  return is_rotation(R, eps);
}

// ---- SE(3) exponential, tangent (upsilon, omega), as Sophus (se3.hpp:321-342 / so3.hpp:322-355)
RPE_HD inline void se3_exp(const double a[6], double R[9], double t[3]) {
  const double wx = a[3], wy = a[4], wz = a[5];
  const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
  double imag, real;
  if (th < 1e-10) { imag = 0.5 - th2 / 48.0 + th2 * th2 / 3840.0; real = 1.0 - th2 / 8.0 + th2 * th2 / 384.0; }
  else { imag = sin(0.5 * th) / th; real = cos(0.5 * th); }
  Quat<double> q{real, imag * wx, imag * wy, imag * wz};
  quat_to_R(q, R);
  const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double W2[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
  double V[9];
  if (th < 1e-10) { for (int k = 0; k < 9; k++) V[k] = R[k]; }
  else {
    const double c1 = (1.0 - cos(th)) / th2, c2 = (th - sin(th)) / (th2 * th);
    for (int k = 0; k < 9; k++) V[k] = (k % 4 == 0 ? 1.0 : 0.0) + c1 * W[k] + c2 * W2[k];
  }
  for (int i = 0; i < 3; i++) t[i] = V[3 * i] * a[0] + V[3 * i + 1] * a[1] + V[3 * i + 2] * a[2];
}
// pose12 (R row-major | t) <- exp(delta) * pose12
RPE_HD inline void se3_left_update(const double delta[6], double pose[12]) {
  double Rd[9], td[3], Rn[9], tn[3];
  se3_exp(delta, Rd, td);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) Rn[3 * i + j] = Rd[3 * i] * pose[j] + Rd[3 * i + 1] * pose[3 + j] + Rd[3 * i + 2] * pose[6 + j];
    tn[i] = Rd[3 * i] * pose[9] + Rd[3 * i + 1] * pose[10] + Rd[3 * i + 2] * pose[11] + td[i];
  }
  for (int k = 0; k < 9; k++) pose[k] = Rn[k];
  for (int k = 0; k < 3; k++) pose[9 + k] = tn[k];
}

// Relative pivot floor of the 6x6 solves: a pivot of a RANK-DEFICIENT system (one repeated point, points on a line, a single plane seen
// point-to-plane) cancels to the rounding noise of the sums it is made of, and whether that noise comes out positive depends on the
// order of the sums -- so the floor sits above the noise of the PRODUCT dtype: the entries of H are sums of products rounded to the
// array dtype (relative noise between eps / sqrt(N) and eps of a diagonal entry), hence 16 eps of that dtype (9.5e-7 for fp32 arrays,
// 1e-12 at least).  A system refused by this floor has a direction determined to less than six (fp32) digits of its diagonal: its
// update would be rounding noise.  The same floor on the host (solve_normal_eq6) and on the device (gn_solve_update, rpe_reduce.hpp).
RPE_HD inline double pivot_floor(bool f64_products) { return f64_products ? 1e-12 : 16.0 * 5.9604644775390625e-08; }

// Solve the 6x6 SPD system H d = -g given the packed record (H upper triangle row-major 21 | g 6).  false if not SPD, i.e. if a pivot
// is not above rel_floor x its diagonal entry.
RPE_HD inline bool solve_normal_eq6(const double* ne, double d[6], double rel_floor = 1e-12) {
  double A[6][6];
  int k = 0;
  for (int i = 0; i < 6; i++) for (int j = i; j < 6; j++) { A[i][j] = ne[k]; A[j][i] = ne[k]; k++; }
  const double* g = ne + 21;
  // LDL^T without pivoting
  double L[6][6] = {{0}}, D[6];
  for (int j = 0; j < 6; j++) {
    double dj = A[j][j];
    for (int m = 0; m < j; m++) dj -= L[j][m] * L[j][m] * D[m];
    // a pivot that cancelled to rounding noise (rank-deficient sets: one repeated point, points on a line, a single plane for
    // point-to-plane) is "not positive definite" too: dividing by it would hand back a finite but meaningless update
    if (!(dj > rel_floor * A[j][j]) || !(dj < 1e300)) return false;
    D[j] = dj;
    L[j][j] = 1.0;
    for (int i = j + 1; i < 6; i++) {
      double s = A[i][j];
      for (int m = 0; m < j; m++) s -= L[i][m] * L[j][m] * D[m];
      L[i][j] = s / dj;
    }
  }
  double y[6];
  for (int i = 0; i < 6; i++) { double s = -g[i]; for (int m = 0; m < i; m++) s -= L[i][m] * y[m]; y[i] = s; }
  for (int i = 0; i < 6; i++) y[i] /= D[i];
  for (int i = 5; i >= 0; i--) { double s = y[i]; for (int m = i + 1; m < 6; m++) s -= L[m][i] * d[m]; d[i] = s; }
  for (int i = 0; i < 6; i++) if (!(d[i] == d[i]) || !(d[i] < 1e300 && d[i] > -1e300)) return false;
  return true;
}

}  // namespace rpe
