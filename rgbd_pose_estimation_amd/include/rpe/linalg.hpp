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
RPE_HD inline Mat3d transposed(const Mat3d& x) { Mat3d r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = x(j, i); return r; }
RPE_HD inline double det3(const Mat3d& m) {
  return m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
         m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
}

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations: S = V diag(l) V^T.
RPE_HD inline void sym_eig3(const Mat3d& S_in, Mat3d* V, double l[3]) {
  Mat3d S = S_in;
  *V = Mat3d::eye();
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = S(0, 1) * S(0, 1) + S(0, 2) * S(0, 2) + S(1, 2) * S(1, 2);
    double dia = S(0, 0) * S(0, 0) + S(1, 1) * S(1, 1) + S(2, 2) * S(2, 2);
    if (off <= 1e-34 * dia || off == 0.0) break;
    for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
      if (S(p, q) == 0.0) continue;
      double theta = (S(q, q) - S(p, p)) / (2.0 * S(p, q));
      double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
      double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
      for (int k = 0; k < 3; k++) {  // S <- S J
        double skp = S(k, p), skq = S(k, q);
        S(k, p) = c * skp - s * skq; S(k, q) = s * skp + c * skq;
      }
      for (int k = 0; k < 3; k++) {  // S <- J^T S
        double spk = S(p, k), sqk = S(q, k);
        S(p, k) = c * spk - s * sqk; S(q, k) = s * spk + c * sqk;
      }
      for (int k = 0; k < 3; k++) {
        double vkp = (*V)(k, p), vkq = (*V)(k, q);
        (*V)(k, p) = c * vkp - s * vkq; (*V)(k, q) = s * vkp + c * vkq;
      }
    }
  }
  l[0] = S(0, 0); l[1] = S(1, 1); l[2] = S(2, 2);
}

// A = U diag(s) V^T, s descending, U and V orthogonal (completed when A is rank deficient).
struct Svd3 { Mat3d U, V; double s[3]; };
RPE_HD inline Svd3 svd3(const Mat3d& A) {
  Mat3d W; double l[3];
  sym_eig3(mul(transposed(A), A), &W, l);
  int ord[3] = {0, 1, 2};
  for (int i = 1; i < 3; i++) {   // insertion sort, descending, stable: what std::sort does for 3 elements, usable on the device
    const int o = ord[i];
    int j = i;
    while (j > 0 && l[o] > l[ord[j - 1]]) { ord[j] = ord[j - 1]; j--; }
    ord[j] = o;
  }
  Svd3 r;
  Vec3d av[3];
  for (int k = 0; k < 3; k++) { r.V.setcol(k, W.col(ord[k])); av[k] = mul(A, r.V.col(k)); r.s[k] = norm3(av[k]); }
  // modified Gram-Schmidt on A V: exact orthogonality of U even when V is only accurate to rounding
  const double tiny = 4.0 * std::numeric_limits<double>::epsilon() * (r.s[0] > 0 ? r.s[0] : 1.0);
  Vec3d u[3];
  int have = 0;
  for (int k = 0; k < 3; k++) {
    Vec3d w = av[k];
    for (int j = 0; j < have; j++) w = w - dot3(u[j], w) * u[j];
    double nw = norm3(w);
    if (nw > tiny && have == k) { u[have++] = (1.0 / nw) * w; }
    else break;
  }
  if (have == 0) { u[0] = Vec3d(1, 0, 0); u[1] = Vec3d(0, 1, 0); u[2] = Vec3d(0, 0, 1); }
  else if (have == 1) {
    Vec3d e = std::fabs(u[0][0]) < 0.6 ? Vec3d(1, 0, 0) : Vec3d(0, 1, 0);
    Vec3d w = cross3(u[0], e); u[1] = (1.0 / norm3(w)) * w; u[2] = cross3(u[0], u[1]);
  } else if (have == 2) { u[2] = cross3(u[0], u[1]); }
  for (int k = 0; k < 3; k++) r.U.setcol(k, u[k]);
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
template <class T> void quat_rotate(const Quat<T>& q, const T* v, T* out) {
  T ux = q.y * v[2] - q.z * v[1], uy = q.z * v[0] - q.x * v[2], uz = q.x * v[1] - q.y * v[0];
  ux += ux; uy += uy; uz += uz;
  const T cx = q.y * uz - q.z * uy, cy = q.z * ux - q.x * uz, cz = q.x * uy - q.y * ux;
  out[0] = (v[0] + q.w * ux) + cx; out[1] = (v[1] + q.w * uy) + cy; out[2] = (v[2] + q.w * uz) + cz;
}
template <class T> Quat<T> quat_mul(const Quat<T>& a, const Quat<T>& b) {
  return Quat<T>{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                 a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
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

// Solve the 6x6 SPD system H d = -g given the packed record (H upper triangle row-major 21 | g 6).  false if not SPD.
RPE_HD inline bool solve_normal_eq6(const double* ne, double d[6]) {
  double A[6][6];
  int k = 0;
  for (int i = 0; i < 6; i++) for (int j = i; j < 6; j++) { A[i][j] = ne[k]; A[j][i] = ne[k]; k++; }
  const double* g = ne + 21;
  // LDL^T without pivoting
  double L[6][6] = {{0}}, D[6];
  for (int j = 0; j < 6; j++) {
    double dj = A[j][j];
    for (int m = 0; m < j; m++) dj -= L[j][m] * L[j][m] * D[m];
    if (!(dj > 0) || !(dj < 1e300)) return false;
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
