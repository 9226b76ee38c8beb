// ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it.
//
// orc_linalg.hpp: the small slice of Eigen / Sophus arithmetic that the reference's hot path uses,
// restated in plain C++ because Eigen3 (un-vendored, version unpinned: /root/reference/CMakeLists.txt:17)
// is absent from this image, so the reference itself cannot be compiled here.
//
// PARITY UNPINNED: the reference ships no golden vectors and cannot be built, so this restatement is
// pinned against (a) an independent numpy/scipy implementation (tests/golden/make_golden.py) and
// (b) analytic known-answer cases, not against outputs of the reference binary.
//
// Arithmetic follows the Eigen 3.3 formulas the reference reaches through Sophus:
//   quaternion <- matrix        sophus/so3.hpp:561  (Eigen::Quaternion(Matrix3) trace branch form)
//   q * v  (rotate)             sophus/so3.hpp:238-240 (Eigen _transformVector: v + w*2(u x v) + u x 2(u x v))
//   q -> matrix                 sophus/so3.hpp:204-206 (Eigen toRotationMatrix)
//   SO3 product + renormalise   sophus/so3.hpp:218-222,258-275
//   SO3/SE3 exp, log            sophus/so3.hpp:313-355,450-499 ; sophus/se3.hpp:321-342,463-496
//   epsilon                     sophus/common.hpp:137-151 (1e-10 double, 1e-5 float)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#include <algorithm>
#include <complex>

namespace orc {

template <class T> struct Eps { static T v() { return T(1e-10); } };
template <> struct Eps<float> { static float v() { return 1e-5f; } };

template <class T> struct V3 {
  T x, y, z;
  V3() : x(0), y(0), z(0) {}
  V3(T a, T b, T c) : x(a), y(b), z(c) {}
  T& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  T operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <class T> inline V3<T> operator+(const V3<T>& a, const V3<T>& b) { return V3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> inline V3<T> operator-(const V3<T>& a, const V3<T>& b) { return V3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> inline V3<T> operator-(const V3<T>& a) { return V3<T>(-a.x, -a.y, -a.z); }
template <class T> inline V3<T> operator*(T s, const V3<T>& a) { return V3<T>(s * a.x, s * a.y, s * a.z); }
template <class T> inline V3<T> operator*(const V3<T>& a, T s) { return V3<T>(a.x * s, a.y * s, a.z * s); }
template <class T> inline V3<T> operator/(const V3<T>& a, T s) { return V3<T>(a.x / s, a.y / s, a.z / s); }
template <class T> inline T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> inline T sqnorm(const V3<T>& a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
template <class T> inline T norm(const V3<T>& a) { return std::sqrt(sqnorm(a)); }
template <class T> inline V3<T> cross(const V3<T>& a, const V3<T>& b) {
  return V3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <class T> inline V3<T> normalized(const V3<T>& a) { return a / norm(a); }
template <class T> inline bool isnan3(const V3<T>& a) { return a.x != a.x && a.y != a.y && a.z != a.z; }

// 3x3, element (r,c) at m[r][c]
template <class T> struct M3 {
  T m[3][3];
  M3() { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] = T(0); }
  static M3 identity() { M3 r; r.m[0][0] = r.m[1][1] = r.m[2][2] = T(1); return r; }
  T& operator()(int r, int c) { return m[r][c]; }
  T operator()(int r, int c) const { return m[r][c]; }
  V3<T> col(int c) const { return V3<T>(m[0][c], m[1][c], m[2][c]); }
  V3<T> row(int r) const { return V3<T>(m[r][0], m[r][1], m[r][2]); }
  void set_col(int c, const V3<T>& v) { m[0][c] = v.x; m[1][c] = v.y; m[2][c] = v.z; }
  void set_row(int r, const V3<T>& v) { m[r][0] = v.x; m[r][1] = v.y; m[r][2] = v.z; }
};
template <class T> inline M3<T> operator*(const M3<T>& a, const M3<T>& b) {
  M3<T> r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    T s = T(0);
    for (int k = 0; k < 3; k++) s += a.m[i][k] * b.m[k][j];
    r.m[i][j] = s;
  }
  return r;
}
template <class T> inline V3<T> operator*(const M3<T>& a, const V3<T>& v) {
  return V3<T>(a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
               a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
               a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z);
}
template <class T> inline M3<T> operator+(const M3<T>& a, const M3<T>& b) {
  M3<T> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j]; return r;
}
template <class T> inline M3<T> operator-(const M3<T>& a, const M3<T>& b) {
  M3<T> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] - b.m[i][j]; return r;
}
template <class T> inline M3<T> operator*(T s, const M3<T>& a) {
  M3<T> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = s * a.m[i][j]; return r;
}
template <class T> inline M3<T> transpose(const M3<T>& a) {
  M3<T> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i]; return r;
}
template <class T> inline T det(const M3<T>& a) {
  return a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1]) -
         a.m[0][1] * (a.m[1][0] * a.m[2][2] - a.m[1][2] * a.m[2][0]) +
         a.m[0][2] * (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]);
}
template <class T> inline M3<T> outer(const V3<T>& a, const V3<T>& b) {
  M3<T> r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a[i] * b[j];
  return r;
}
template <class T> inline T frob(const M3<T>& a) {
  T s = 0; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) s += a.m[i][j] * a.m[i][j]; return std::sqrt(s);
}
template <class T> inline M3<T> hat(const V3<T>& w) {
  M3<T> r;
  r.m[0][1] = -w.z; r.m[0][2] = w.y;
  r.m[1][0] = w.z;  r.m[1][2] = -w.x;
  r.m[2][0] = -w.y; r.m[2][1] = w.x;
  return r;
}

// ---- 3x3 SVD: the algorithm of Eigen::JacobiSVD (Eigen 3.3.x, Eigen/src/SVD/JacobiSVD.h + Eigen/src/Jacobi/Jacobi.h), restated.
// Eigen is an un-vendored, version-unpinned dependency of the reference (CMakeLists.txt:17); every SVD on the hot path is
// JacobiSVD<Matrix<Tp,...>>(M, ComputeFullU | ComputeFullV) on a square 3x3 (AbsoluteOrientation.hpp:79,
// AbsoluteOrientationNormal.hpp:44,188,512), i.e. no QR preconditioner runs.  The published algorithm:
//   scale = max |a_ij| (1 if zero);  W = A / scale;  U = V = I;  maxDiag = max_i |W_ii|
//   sweep until no 2x2 block needed work:  for p = 1..2, q = 0..p-1:
//     threshold = max(min_positive, 2 eps * maxDiag);  if |W_pq| > threshold or |W_qp| > threshold:
//       (j_left, j_right) = real_2x2_jacobi_svd(W, p, q);  W <- j_left applied on the left, U <- U j_left^T, W <- W j_right, V <- V j_right
//       maxDiag = max(maxDiag, |W_pp|, |W_qq|)
//   s_i = |W_ii| * scale, column i of U multiplied by sign(W_ii); then sorted descending by swapping (selection order).
// A = U diag(s) V^T with U, V orthogonal full bases also when A is rank deficient (they start from I).
template <class T> struct SVD3 { M3<T> U, V; T s[3]; };

template <class T> struct JacobiRot {   // Eigen::JacobiRotation<T>: the 2x2 matrix [c s; -s c]
  T c, s;
  JacobiRot transpose() const { return JacobiRot{c, -s}; }
  JacobiRot operator*(const JacobiRot& o) const { return JacobiRot{c * o.c - s * o.s, c * o.s + s * o.c}; }
};
// Jacobi.h makeJacobi(x, y, z): rotation J with J^T [x y; y z] J diagonal
template <class T> inline JacobiRot<T> make_jacobi(T x, T y, T z) {
  const T deno = T(2) * std::fabs(y);
  if (deno < std::numeric_limits<T>::min()) return JacobiRot<T>{T(1), T(0)};
  const T tau = (x - z) / deno;
  const T w = std::sqrt(tau * tau + T(1));
  const T t = tau > T(0) ? T(1) / (tau + w) : T(1) / (tau - w);
  const T sign_t = t > T(0) ? T(1) : T(-1);
  const T n = T(1) / std::sqrt(t * t + T(1));
  return JacobiRot<T>{n, -sign_t * (y / std::fabs(y)) * std::fabs(t) * n};
}
// rows p, q of m <- J^T-style left application (Eigen applyOnTheLeft(p, q, j)): x' = c x + s y ; y' = -s x + c y
template <class T> inline void rot_left(M3<T>& m, int p, int q, const JacobiRot<T>& j) {
  for (int k = 0; k < 3; k++) { const T x = m.m[p][k], y = m.m[q][k]; m.m[p][k] = j.c * x + j.s * y; m.m[q][k] = -j.s * x + j.c * y; }
}
// columns p, q of m <- m J (Eigen applyOnTheRight(p, q, j)): x' = c x - s y ; y' = s x + c y
template <class T> inline void rot_right(M3<T>& m, int p, int q, const JacobiRot<T>& j) {
  for (int k = 0; k < 3; k++) { const T x = m.m[k][p], y = m.m[k][q]; m.m[k][p] = j.c * x - j.s * y; m.m[k][q] = j.s * x + j.c * y; }
}

template <class T> SVD3<T> svd3(const M3<T>& A_in) {
  T scale = T(0);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) scale = std::max(scale, std::fabs(A_in.m[i][j]));
  if (scale == T(0)) scale = T(1);
  M3<T> W, U = M3<T>::identity(), V = M3<T>::identity();
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) W.m[i][j] = A_in.m[i][j] / scale;
  const T precision = T(2) * std::numeric_limits<T>::epsilon(), consider_as_zero = std::numeric_limits<T>::min();
  T max_diag = std::max(std::fabs(W.m[0][0]), std::max(std::fabs(W.m[1][1]), std::fabs(W.m[2][2])));
  bool finished = false;
  for (int sweep = 0; !finished && sweep < 1000; sweep++) {   // Eigen has no sweep cap; NaN input would spin, so a generous one
    finished = true;
    for (int p = 1; p < 3; p++) for (int q = 0; q < p; q++) {
      const T threshold = std::max(consider_as_zero, precision * max_diag);
      if (std::fabs(W.m[p][q]) > threshold || std::fabs(W.m[q][p]) > threshold) {
        finished = false;
        // real_2x2_jacobi_svd: first a rotation that makes the 2x2 block symmetric, then the symmetric Jacobi rotation
        T m00 = W.m[p][p], m01 = W.m[p][q], m10 = W.m[q][p], m11 = W.m[q][q];
        JacobiRot<T> rot1;
        const T t = m00 + m11, d = m10 - m01;
        if (std::fabs(d) < std::numeric_limits<T>::min()) { rot1.s = T(0); rot1.c = T(1); }
        else { const T u = t / d, tmp = std::sqrt(T(1) + u * u); rot1.s = T(1) / tmp; rot1.c = u / tmp; }
        { const T x0 = m00, x1 = m01, y0 = m10, y1 = m11;   // m.applyOnTheLeft(0, 1, rot1)
          m00 = rot1.c * x0 + rot1.s * y0; m01 = rot1.c * x1 + rot1.s * y1; m10 = -rot1.s * x0 + rot1.c * y0; m11 = -rot1.s * x1 + rot1.c * y1; }
        const JacobiRot<T> j_right = make_jacobi(m00, m01, m11);
        const JacobiRot<T> j_left = rot1 * j_right.transpose();
        rot_left(W, p, q, j_left);
        rot_right(U, p, q, j_left.transpose());
        rot_right(W, p, q, j_right);
        rot_right(V, p, q, j_right);
        max_diag = std::max(max_diag, std::max(std::fabs(W.m[p][p]), std::fabs(W.m[q][q])));
      }
    }
  }
  SVD3<T> r;
  for (int i = 0; i < 3; i++) {
    const T a = std::fabs(W.m[i][i]);
    r.s[i] = a;
    if (a != T(0)) { const T sg = W.m[i][i] / a; for (int k = 0; k < 3; k++) U.m[k][i] *= sg; }
  }
  for (int i = 0; i < 3; i++) r.s[i] *= scale;
  for (int i = 0; i < 3; i++) {   // descending, by swaps with the largest remaining value (first one on ties)
    int pos = i;
    for (int k = i + 1; k < 3; k++) if (r.s[k] > r.s[pos]) pos = k;
    if (r.s[pos] == T(0)) break;
    if (pos != i) {
      std::swap(r.s[i], r.s[pos]);
      for (int k = 0; k < 3; k++) { std::swap(U.m[k][i], U.m[k][pos]); std::swap(V.m[k][i], V.m[k][pos]); }
    }
  }
  r.U = U; r.V = V;
  return r;
}

// least-squares solve of A x = b through the SVD (Eigen jacobiSvd().solve, used by find_opt_cc)
template <class T> V3<T> svd_solve(const M3<T>& A, const V3<T>& b) {
  SVD3<T> d = svd3(A);
  V3<T> y = transpose(d.U) * b;
  const T thr = std::numeric_limits<T>::epsilon() * T(3) * d.s[0];
  for (int k = 0; k < 3; k++) y[k] = d.s[k] > thr ? y[k] / d.s[k] : T(0);
  return d.V * y;
}

// ---- unit quaternion (w, x, y, z)
template <class T> struct Quat {
  T w, x, y, z;
  Quat() : w(1), x(0), y(0), z(0) {}
  Quat(T w_, T x_, T y_, T z_) : w(w_), x(x_), y(y_), z(z_) {}
  V3<T> vec() const { return V3<T>(x, y, z); }
  T sqnorm() const { return w * w + x * x + y * y + z * z; }
};
template <class T> inline Quat<T> quat_mul(const Quat<T>& a, const Quat<T>& b) {
  return Quat<T>(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
                 a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                 a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                 a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x);
}
template <class T> Quat<T> quat_from_matrix(const M3<T>& m) {
  Quat<T> q;
  T t = m(0, 0) + m(1, 1) + m(2, 2);
  if (t > T(0)) {
    t = std::sqrt(t + T(1));
    q.w = T(0.5) * t;
    t = T(0.5) / t;
    q.x = (m(2, 1) - m(1, 2)) * t;
    q.y = (m(0, 2) - m(2, 0)) * t;
    q.z = (m(1, 0) - m(0, 1)) * t;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + T(1));
    T v[3];
    v[i] = T(0.5) * t;
    t = T(0.5) / t;
    q.w = (m(k, j) - m(j, k)) * t;
    v[j] = (m(j, i) + m(i, j)) * t;
    v[k] = (m(k, i) + m(i, k)) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}
template <class T> M3<T> quat_to_matrix(const Quat<T>& q) {
  M3<T> r;
  const T tx = T(2) * q.x, ty = T(2) * q.y, tz = T(2) * q.z;
  const T twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const T txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const T tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  r(0, 0) = T(1) - (tyy + tzz); r(0, 1) = txy - twz;          r(0, 2) = txz + twy;
  r(1, 0) = txy + twz;          r(1, 1) = T(1) - (txx + tzz); r(1, 2) = tyz - twx;
  r(2, 0) = txz - twy;          r(2, 1) = tyz + twx;          r(2, 2) = T(1) - (txx + tyy);
  return r;
}
template <class T> inline V3<T> quat_rotate(const Quat<T>& q, const V3<T>& v) {
  V3<T> u = q.vec();
  V3<T> uv = cross(u, v);
  uv = uv + uv;
  return (v + q.w * uv) + cross(u, uv);
}

// ---- SO3 / SE3 with the Sophus semantics the reference relies on
template <class T> struct SO3 {
  Quat<T> q;
  bool ok;  // false where SOPHUS_ENSURE would have aborted (sophus/common.hpp:114-133)
  SO3() : q(), ok(true) {}
  explicit SO3(const M3<T>& R) : q(quat_from_matrix(R)), ok(true) {   // so3.hpp:561-566
    M3<T> E = R * transpose(R) - M3<T>::identity();
    if (!(frob(E) < Eps<T>::v()) || !(det(R) > T(0))) ok = false;
  }
  explicit SO3(const Quat<T>& qq) : q(qq), ok(true) {                  // so3.hpp:578-585 (normalises)
    T len = std::sqrt(q.sqnorm());
    if (!(len >= Eps<T>::v())) { ok = false; return; }
    q.w /= len; q.x /= len; q.y /= len; q.z /= len;
  }
  static SO3 from_angle_axis(T angle, const V3<T>& axis) {             // Eigen Quaternion(AngleAxis)
    T ha = T(0.5) * angle, s = std::sin(ha);
    return SO3(Quat<T>(std::cos(ha), s * axis.x, s * axis.y, s * axis.z));
  }
  SO3 inverse() const { SO3 r; r.q = Quat<T>(q.w, -q.x, -q.y, -q.z); r.ok = ok; return r; }  // no renormalise in effect
  M3<T> matrix() const { return quat_to_matrix(q); }
  V3<T> operator*(const V3<T>& p) const { return quat_rotate(q, p); }
  SO3 operator*(const SO3& o) const {                                  // so3.hpp:258-275
    SO3 r; r.q = quat_mul(q, o.q); r.ok = ok && o.ok;
    T n2 = r.q.sqnorm();
    if (n2 != T(1)) { T f = T(2) / (T(1) + n2); r.q.w *= f; r.q.x *= f; r.q.y *= f; r.q.z *= f; }
    return r;
  }
  static SO3 exp(const V3<T>& omega, T* theta_out = nullptr) {         // so3.hpp:322-355
    T theta_sq = sqnorm(omega), theta = std::sqrt(theta_sq), half = T(0.5) * theta;
    T imag, real;
    if (theta < Eps<T>::v()) {
      T t4 = theta_sq * theta_sq;
      imag = T(0.5) - T(1.0 / 48.0) * theta_sq + T(1.0 / 3840.0) * t4;
      real = T(1) - T(1.0 / 8.0) * theta_sq + T(1.0 / 384.0) * t4;
    } else {
      imag = std::sin(half) / theta;
      real = std::cos(half);
    }
    if (theta_out) *theta_out = theta;
    SO3 r; r.q = Quat<T>(real, imag * omega.x, imag * omega.y, imag * omega.z);
    return r;
  }
  V3<T> log(T* theta_out = nullptr) const {                            // so3.hpp:466-499
    T sq_n = sqnorm(q.vec()), n = std::sqrt(sq_n), w = q.w, f;
    if (n < Eps<T>::v()) {
      f = T(2) / w - T(2) * sq_n / (w * w * w);
    } else if (std::fabs(w) < Eps<T>::v()) {
      f = (w > T(0) ? T(M_PI) : -T(M_PI)) / n;
    } else {
      f = T(2) * std::atan(n / w) / n;
    }
    if (theta_out) *theta_out = f * n;
    return f * q.vec();
  }
};

template <class T> struct SE3 {
  SO3<T> R;
  V3<T> t;
  SE3() {}
  SE3(const SO3<T>& r, const V3<T>& tt) : R(r), t(tt) {}
  V3<T> operator*(const V3<T>& p) const { return R * p + t; }
  SE3 operator*(const SE3& o) const { return SE3(R * o.R, t + R * o.t); }
  SE3 inverse() const { SO3<T> ri = R.inverse(); return SE3(ri, ri * (-t)); }
  static SE3 exp(const T a[6]) {                                       // se3.hpp:321-342, a = (upsilon, omega)
    V3<T> ups(a[0], a[1], a[2]), om(a[3], a[4], a[5]);
    T theta;
    SO3<T> so3 = SO3<T>::exp(om, &theta);
    M3<T> Om = hat(om), Om2 = Om * Om, V;
    if (theta < Eps<T>::v()) {
      V = so3.matrix();
    } else {
      T th2 = theta * theta;
      V = M3<T>::identity() + ((T(1) - std::cos(theta)) / th2) * Om + ((theta - std::sin(theta)) / (th2 * theta)) * Om2;
    }
    return SE3(so3, V * ups);
  }
  void log(T out[6]) const {                                           // se3.hpp:463-496
    T theta;
    V3<T> om = R.log(&theta);
    M3<T> Om = hat(om), Vinv;
    if (std::fabs(theta) < Eps<T>::v()) {
      Vinv = M3<T>::identity() - T(0.5) * Om + T(1. / 12.) * (Om * Om);
    } else {
      T half = T(0.5) * theta;
      Vinv = M3<T>::identity() - T(0.5) * Om +
             ((T(1) - theta * std::cos(half) / (T(2) * std::sin(half))) / (theta * theta)) * (Om * Om);
    }
    V3<T> u = Vinv * t;
    out[0] = u.x; out[1] = u.y; out[2] = u.z; out[3] = om.x; out[4] = om.y; out[5] = om.z;
  }
};

// angle of the rotation matrix as Eigen::AngleAxis(Matrix3) reports it (via the quaternion)
template <class T> T rotation_angle(const M3<T>& R) {
  Quat<T> q = quat_from_matrix(R);
  T n = norm(q.vec());
  if (n != T(0)) return T(2) * std::atan2(n, std::fabs(q.w));
  return T(0);
}

// column-major dynamic matrix, just enough of Eigen::Matrix<T,Dynamic,Dynamic>
template <class T> struct MatX {
  int r, c;
  std::vector<T> d;
  MatX() : r(0), c(0) {}
  MatX(int rows, int cols) : r(rows), c(cols), d((size_t)rows * cols, T(0)) {}
  void resize(int rows, int cols) { r = rows; c = cols; d.assign((size_t)rows * cols, T(0)); }
  int rows() const { return r; }
  int cols() const { return c; }
  T& operator()(int i, int j) { return d[(size_t)j * r + i]; }
  T operator()(int i, int j) const { return d[(size_t)j * r + i]; }
  T* data() { return d.data(); }
  const T* data() const { return d.data(); }
  V3<T> col3(int j) const { const T* p = &d[(size_t)j * r]; return V3<T>(p[0], p[1], p[2]); }
  void set_col3(int j, const V3<T>& v) { T* p = &d[(size_t)j * r]; p[0] = v.x; p[1] = v.y; p[2] = v.z; }
  void set_zero() { std::fill(d.begin(), d.end(), T(0)); }
};

// The reference draws sample indices with libc rand() (pose/Utility.hpp:148,212,229), unseeded.  The
// restatement makes the stream explicit and portable: PCG32 (O'Neill, pcg-random.org, XSH-RR 64/32),
// output >> 1 so that it has rand()'s 31-bit range.  Product and oracle both implement THIS stream so
// that sampled index sequences (integer work) are bit-identical.
struct Rand31 {
  uint64_t state, inc;
  explicit Rand31(uint64_t seed = 1, uint64_t seq = 54) { reseed(seed, seq); }
  void reseed(uint64_t seed, uint64_t seq = 54) {
    state = 0; inc = (seq << 1) | 1u;
    next32(); state += seed; next32();
  }
  uint32_t next32() {
    uint64_t old = state;
    state = old * 6364136223846793005ULL + inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((-rot) & 31));
  }
  int operator()() { return (int)(next32() >> 1); }
};

}  // namespace orc
