// rpe/types.hpp -- the value types the drop-in pose/*.hpp headers expose in place of Eigen / Sophus ones.
//
// The reference's adapters traffic in Eigen::Matrix<Tp,3,1>, Eigen::Matrix<Tp,Dynamic,Dynamic>,
// Sophus::SO3<Tp> and Sophus::SE3<Tp> (pose/PoseAdapterBase.hpp:31-37).  Eigen is not available to this build, so
// the same spellings are provided here with the subset of members the pose/ headers and their callers use:
//   Point3<Tp>        x y z + operator[] / (i), arithmetic, norm, normalize, dot, cross
//   MatrixX<Tp>       column-major dynamic matrix: rows cols data col(i) operator()(r,c) resize setZero Zero Ones
//   SO3<Tp>, SE3<Tp>  unit-quaternion rotation / rigid transform with Sophus' semantics (sophus/so3.hpp, se3.hpp)
// `namespace Sophus` aliases SO3/SE3 so that caller code written against the reference keeps compiling.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <limits>
#include "linalg.hpp"

namespace rpe {

template <class Tp> struct Point3 {
  Tp v[3];
  Point3() : v{Tp(0), Tp(0), Tp(0)} {}
  Point3(Tp a, Tp b, Tp c) : v{a, b, c} {}
  explicit Point3(const Tp* p) : v{p[0], p[1], p[2]} {}
  static Point3 Zero() { return Point3(); }
  Tp& operator[](int i) { return v[i]; }
  Tp operator[](int i) const { return v[i]; }
  Tp& operator()(int i) { return v[i]; }
  Tp operator()(int i) const { return v[i]; }
  Tp x() const { return v[0]; }
  Tp y() const { return v[1]; }
  Tp z() const { return v[2]; }
  Tp squaredNorm() const { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; }
  Tp norm() const { return std::sqrt(squaredNorm()); }
  void normalize() { Tp n = norm(); v[0] /= n; v[1] /= n; v[2] /= n; }
  Point3 normalized() const { Point3 r = *this; r.normalize(); return r; }
  Tp dot(const Point3& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
  Point3 cross(const Point3& o) const {
    return Point3(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]);
  }
  Point3& operator+=(const Point3& o) { v[0] += o.v[0]; v[1] += o.v[1]; v[2] += o.v[2]; return *this; }
  Point3& operator-=(const Point3& o) { v[0] -= o.v[0]; v[1] -= o.v[1]; v[2] -= o.v[2]; return *this; }
  Point3& operator/=(Tp s) { v[0] /= s; v[1] /= s; v[2] /= s; return *this; }
  Point3& operator*=(Tp s) { v[0] *= s; v[1] *= s; v[2] *= s; return *this; }
  const Tp* data() const { return v; }
  Tp* data() { return v; }
};
template <class Tp> inline Point3<Tp> operator+(Point3<Tp> a, const Point3<Tp>& b) { return a += b; }
template <class Tp> inline Point3<Tp> operator-(Point3<Tp> a, const Point3<Tp>& b) { return a -= b; }
template <class Tp> inline Point3<Tp> operator-(const Point3<Tp>& a) { return Point3<Tp>(-a.v[0], -a.v[1], -a.v[2]); }
template <class Tp> inline Point3<Tp> operator*(Tp s, Point3<Tp> a) { return a *= s; }
template <class Tp> inline Point3<Tp> operator*(Point3<Tp> a, Tp s) { return a *= s; }
template <class Tp> inline Point3<Tp> operator/(Point3<Tp> a, Tp s) { return a /= s; }

// 3x3 in Tp, row-major storage, (r,c) access
template <class Tp> struct Matrix3 {
  Tp a[9];
  Matrix3() { for (int i = 0; i < 9; i++) a[i] = Tp(0); }
  static Matrix3 Identity() { Matrix3 m; m.a[0] = m.a[4] = m.a[8] = Tp(1); return m; }
  static Matrix3 Zero() { return Matrix3(); }
  Tp& operator()(int r, int c) { return a[3 * r + c]; }
  Tp operator()(int r, int c) const { return a[3 * r + c]; }
  Matrix3 transpose() const { Matrix3 t; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t(i, j) = (*this)(j, i); return t; }
  Tp determinant() const {
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
  }
  Point3<Tp> operator*(const Point3<Tp>& p) const {
    return Point3<Tp>(a[0] * p[0] + a[1] * p[1] + a[2] * p[2], a[3] * p[0] + a[4] * p[1] + a[5] * p[2],
        a[6] * p[0] + a[7] * p[1] + a[8] * p[2]);
  }
  Matrix3 operator*(const Matrix3& o) const {
    Matrix3 r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = (*this)(i, 0) * o(0, j) + (*this)(i, 1) * o(1, j) + (*this)(i, 2) * o(2, j);
    return r;
  }
  void setRow(int r, const Point3<Tp>& p) { a[3 * r] = p[0]; a[3 * r + 1] = p[1]; a[3 * r + 2] = p[2]; }
  const Tp* data() const { return a; }  // ROW-major (unlike Eigen's default): use (r,c) when order matters
};

// column-major dynamic matrix: the layout Eigen::Matrix<Tp,Dynamic,Dynamic> hands to the reference
template <class Tp> class MatrixX {
 public:
  MatrixX() : _r(0), _c(0) {}
  MatrixX(int rows, int cols) : _r(rows), _c(cols), _d((size_t)rows * cols, Tp(0)) {}
  static MatrixX Zero(int r, int c) { return MatrixX(r, c); }
  static MatrixX Ones(int r, int c) { MatrixX m(r, c); std::fill(m._d.begin(), m._d.end(), Tp(1)); return m; }
  void resize(int rows, int cols) { _r = rows; _c = cols; _d.assign((size_t)rows * cols, Tp(0)); }
  void setZero() { std::fill(_d.begin(), _d.end(), Tp(0)); }
  void setOnes() { std::fill(_d.begin(), _d.end(), Tp(1)); }
  int rows() const { return _r; }
  int cols() const { return _c; }
  Tp* data() { return _d.data(); }
  const Tp* data() const { return _d.data(); }
  Tp& operator()(int r, int c) { return _d[(size_t)c * _r + r]; }
  Tp operator()(int r, int c) const { return _d[(size_t)c * _r + r]; }
  Point3<Tp> col(int c) const { return Point3<Tp>(&_d[(size_t)c * _r]); }  // 3-row matrices
  void setCol(int c, const Point3<Tp>& p) { Tp* q = &_d[(size_t)c * _r]; q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; }
  long sum() const { long s = 0; for (const Tp& v : _d) s += (long)v; return s; }
 private:
  int _r, _c;
  std::vector<Tp> _d;
};
typedef MatrixX<short> MatrixXs;

template <class Tp> struct LieEps { static Tp value() { return Tp(1e-10); } };
template <> struct LieEps<float> { static float value() { return 1e-5f; } };

// Rotation as a unit quaternion, Sophus semantics: ctor from a matrix does NOT renormalise, ctor from a quaternion does,
// group product applies the first-order renormalisation of sophus/so3.hpp:258-275.  valid() is false where
// SOPHUS_ENSURE would have aborted the reference (non-orthogonal input, sophus/so3.hpp:561-566).
template <class Tp> class SO3 {
 public:
  SO3() : _q{Tp(1), Tp(0), Tp(0), Tp(0)}, _ok(true) {}
  SO3(const Matrix3<Tp>& R) : _q(quat_from_R<Tp>(R.a)), _ok(true) {
    Matrix3<Tp> E = R * R.transpose();
    Tp f = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Tp d = E(i, j) - (i == j ? Tp(1) : Tp(0)); f += d * d; }
    if (!(std::sqrt(f) < LieEps<Tp>::value()) || !(R.determinant() > Tp(0))) _ok = false;
  }
  static SO3 fromQuaternion(Tp w, Tp x, Tp y, Tp z) {
    SO3 r; Tp n = std::sqrt(w * w + x * x + y * y + z * z);
    // the reference aborts here (SOPHUS_ENSURE); the values are kept
    if (!(n >= LieEps<Tp>::value())) { r._q = Quat<Tp>{w, x, y, z}; r._ok = false; return r; }
    r._q = Quat<Tp>{w / n, x / n, y / n, z / n};
    return r;
  }
  static SO3 fromQuaternionRaw(Tp w, Tp x, Tp y, Tp z) { SO3 r; r._q = Quat<Tp>{w, x, y, z}; return r; }
  static SO3 fromAngleAxis(Tp angle, const Point3<Tp>& axis) {
    const Tp h = Tp(0.5) * angle, s = std::sin(h);
    return fromQuaternion(std::cos(h), s * axis[0], s * axis[1], s * axis[2]);
  }
  static SO3 exp(const Point3<Tp>& omega) {
    const Tp th2 = omega.squaredNorm(), th = std::sqrt(th2);
    Tp imag, real;
    if (th < LieEps<Tp>::value()) { imag = Tp(0.5) - th2 / Tp(48) + th2 * th2 / Tp(3840);
        real = Tp(1) - th2 / Tp(8) + th2 * th2 / Tp(384); }
    else { imag = std::sin(Tp(0.5) * th) / th; real = std::cos(Tp(0.5) * th); }
    return fromQuaternionRaw(real, imag * omega[0], imag * omega[1], imag * omega[2]);
  }
  bool valid() const { return _ok; }
  void invalidate() { _ok = false; }
  const Quat<Tp>& unit_quaternion() const { return _q; }
  SO3 inverse() const { SO3 r = fromQuaternionRaw(_q.w, -_q.x, -_q.y, -_q.z); r._ok = _ok; return r; }
  Matrix3<Tp> matrix() const { Matrix3<Tp> m; quat_to_R<Tp>(_q, m.a); return m; }
  Point3<Tp> operator*(const Point3<Tp>& p) const { Point3<Tp> o; quat_rotate<Tp>(_q, p.v, o.v); return o; }
  SO3 operator*(const SO3& o) const {
    Quat<Tp> q = quat_mul(_q, o._q);
    const Tp n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    if (n2 != Tp(1)) { const Tp f = Tp(2) / (Tp(1) + n2); q.w *= f; q.x *= f; q.y *= f; q.z *= f; }
    SO3 r = fromQuaternionRaw(q.w, q.x, q.y, q.z); r._ok = _ok && o._ok; return r;
  }
 private:
  Quat<Tp> _q;
  bool _ok;
};

template <class Tp> class SE3 {
 public:
  SE3() {}
  SE3(const SO3<Tp>& R, const Point3<Tp>& t) : _R(R), _t(t) {}
  SO3<Tp>& so3() { return _R; }
  const SO3<Tp>& so3() const { return _R; }
  Point3<Tp>& translation() { return _t; }
  const Point3<Tp>& translation() const { return _t; }
  Point3<Tp> operator*(const Point3<Tp>& p) const { return _R * p + _t; }
  SE3 inverse() const { SO3<Tp> ri = _R.inverse(); return SE3(ri, ri * (-_t)); }
  // tangent a = (upsilon, omega), sophus/se3.hpp:321-342; evaluated in double and rounded to Tp
  static SE3 exp(const Tp a[6]) {
    double ad[6], R[9], t[3];
    for (int i = 0; i < 6; i++) ad[i] = a[i];
    se3_exp(ad, R, t);
    Matrix3<Tp> m; for (int i = 0; i < 9; i++) m.a[i] = (Tp)R[i];
    SE3 r; r._R = SO3<Tp>::fromQuaternion(quat_from_R<Tp>(m.a).w, quat_from_R<Tp>(m.a).x, quat_from_R<Tp>(m.a).y,
        quat_from_R<Tp>(m.a).z);
    r._t = Point3<Tp>((Tp)t[0], (Tp)t[1], (Tp)t[2]);
    return r;
  }
 private:
  SO3<Tp> _R;
  Point3<Tp> _t;
};

}  // namespace rpe

#ifndef RPE_NO_SOPHUS_ALIAS
namespace Sophus {
using rpe::SO3;
using rpe::SE3;
}  // namespace Sophus
#endif
