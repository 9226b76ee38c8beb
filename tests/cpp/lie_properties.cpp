// CPU: the properties Sophus' own test suite checks for its Lie groups (sophus/tests.hpp:43-198 -- exp against the matrix exponential
// of hat(x), the group action against the matrix form, the product against the matrix product, inverse) on the drop-in rpe::SO3 /
// rpe::SE3 (rpe/types.hpp), for Tp = double and float, on the kind of element / tangent / point sets that file uses (identity, tiny
// angles on both sides of the series switch, angles near pi, mixed translations).  No GPU call.
#include <cmath>
#include <cstdio>
#include <vector>
#include "rpe/types.hpp"

using rpe::Matrix3;
using rpe::Point3;

static int g_bad = 0;
#define CHECK(cond, ...) do { if (!(cond)) { g_bad++; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

// exp of a 4x4 matrix in long double: scaling and squaring around a 30-term Taylor series (the test's own yardstick)
struct M4 { long double a[16]; };
static M4 mul(const M4& x, const M4& y) {
  M4 r;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { long double s = 0; for (int k = 0; k < 4; k++) s += x.a[4 * i + k] * y.a[4 * k + j]; r.a[4 * i + j] = s; }
  return r;
}
static M4 expm(M4 h) {
  long double nrm = 0;
  for (long double v : h.a) nrm += std::fabs(v);
  int sq = 0;
  while (nrm > 0.25L) { nrm *= 0.5L; sq++; }
  for (long double& v : h.a) v = std::ldexp(v, -sq);
  M4 e{}, term{};
  for (int i = 0; i < 4; i++) e.a[5 * i] = term.a[5 * i] = 1.0L;
  for (int k = 1; k <= 30; k++) { term = mul(term, h); for (long double& v : term.a) v /= k; for (int i = 0; i < 16; i++) e.a[i] += term.a[i]; }
  for (int s = 0; s < sq; s++) e = mul(e, e);
  return e;
}
static M4 hat6(const long double a[6]) {   // (upsilon, omega) -> 4x4
  M4 h{};
  h.a[1] = -a[5]; h.a[2] = a[4]; h.a[4] = a[5]; h.a[6] = -a[3]; h.a[8] = -a[4]; h.a[9] = a[3];
  h.a[3] = a[0]; h.a[7] = a[1]; h.a[11] = a[2];
  return h;
}

template <class Tp> static void run(const char* name, double eps) {
  std::vector<Point3<Tp>> omegas = {Point3<Tp>(0, 0, 0), Point3<Tp>(Tp(1e-12), 0, 0), Point3<Tp>(0, Tp(3e-11), Tp(-2e-11)), Point3<Tp>(Tp(2e-10), Tp(1e-10), 0),
                                    Point3<Tp>(Tp(1e-6), Tp(-2e-6), Tp(3e-6)), Point3<Tp>(Tp(0.2), Tp(0.5), 0), Point3<Tp>(Tp(0.2), Tp(0.5), Tp(-1)),
                                    Point3<Tp>(0, 0, Tp(3.14159)), Point3<Tp>(Tp(3.1), Tp(0.3), Tp(-0.2)), Point3<Tp>(Tp(-1), Tp(1), Tp(2))};
  std::vector<Point3<Tp>> trans = {Point3<Tp>(0, 0, 0), Point3<Tp>(1, 10, 5), Point3<Tp>(Tp(-0.01), Tp(0.02), Tp(-5)), Point3<Tp>(Tp(100), Tp(-3), Tp(0.5))};
  std::vector<Point3<Tp>> points = {Point3<Tp>(1, 2, 4), Point3<Tp>(0, 0, 0), Point3<Tp>(Tp(-0.3), Tp(7), Tp(0.001))};
  // ---- SO3
  std::vector<rpe::SO3<Tp>> Rs;
  for (const auto& w : omegas) {
    const rpe::SO3<Tp> R = rpe::SO3<Tp>::exp(w);
    Rs.push_back(R);
    const long double a[6] = {0, 0, 0, (long double)w[0], (long double)w[1], (long double)w[2]};
    const M4 E = expm(hat6(a));
    const Matrix3<Tp> M = R.matrix();
    double d = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) d = std::fmax(d, std::fabs((double)(M(i, j) - (Tp)E.a[4 * i + j])));
    CHECK(d <= 10 * eps, "%s exp(x) vs expm(hat x): %g at omega (%g %g %g)", name, d, (double)w[0], (double)w[1], (double)w[2]);
    // unit quaternion, orthonormal matrix
    const auto& q = R.unit_quaternion();
    CHECK(std::fabs((double)(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z) - 1.0) <= 4 * eps, "%s |q| - 1", name);
  }
  for (const auto& A : Rs) {
    const Matrix3<Tp> MA = A.matrix();
    for (const auto& p : points) {   // group action against the matrix form
      const Point3<Tp> r1 = A * p, r2 = MA * p;
      CHECK(std::fabs((double)(r1 - r2).norm()) <= eps * (1 + (double)p.norm()), "%s R*p vs matrix*p: %g", name, (double)(r1 - r2).norm());
    }
    for (const auto& B : Rs) {       // product against the matrix product
      const Matrix3<Tp> P1 = (A * B).matrix(), P2 = MA * B.matrix();
      double d = 0;
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) d = std::fmax(d, std::fabs((double)(P1(i, j) - P2(i, j))));
      CHECK(d <= 4 * eps, "%s (A*B).matrix() vs A.matrix()*B.matrix(): %g", name, d);
    }
    const Matrix3<Tp> I = (A * A.inverse()).matrix();
    double d = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) d = std::fmax(d, std::fabs((double)(I(i, j) - (i == j ? Tp(1) : Tp(0)))));
    CHECK(d <= 4 * eps, "%s A*inverse(A): %g", name, d);
    // matrix -> SO3 -> matrix round trip (the constructor the solvers use on U V^T)
    const rpe::SO3<Tp> back(MA);
    CHECK(back.valid(), "%s SO3(matrix) rejects a rotation", name);
    const Matrix3<Tp> MB = back.matrix();
    d = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) d = std::fmax(d, std::fabs((double)(MB(i, j) - MA(i, j))));
    CHECK(d <= 4 * eps, "%s SO3(R).matrix() vs R: %g", name, d);
  }
  {  // what SOPHUS_ENSURE would have stopped: a scaled and a reflected matrix are flagged, not silently accepted
    Matrix3<Tp> S = Rs[5].matrix();
    for (int i = 0; i < 9; i++) S.a[i] *= Tp(1.01);
    CHECK(!rpe::SO3<Tp>(S).valid(), "%s scaled matrix accepted", name);
    Matrix3<Tp> F = Rs[5].matrix();
    for (int j = 0; j < 3; j++) F(2, j) = -F(2, j);
    CHECK(!rpe::SO3<Tp>(F).valid(), "%s reflection accepted", name);
  }
  // ---- SE3
  for (const auto& w : omegas) for (const auto& u : trans) {
    const Tp a[6] = {u[0], u[1], u[2], w[0], w[1], w[2]};
    const long double al[6] = {(long double)u[0], (long double)u[1], (long double)u[2], (long double)w[0], (long double)w[1], (long double)w[2]};
    const rpe::SE3<Tp> T = rpe::SE3<Tp>::exp(a);
    const M4 E = expm(hat6(al));
    const Matrix3<Tp> M = T.so3().matrix();
    double d = 0, scale = 1 + (double)u.norm();
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) d = std::fmax(d, std::fabs((double)(M(i, j) - (Tp)E.a[4 * i + j])));
      d = std::fmax(d, std::fabs((double)(T.translation()[i] - (Tp)E.a[4 * i + 3])) / scale); }
    CHECK(d <= 10 * eps, "%s SE3 exp(x) vs expm(hat x): %g", name, d);
    for (const auto& p : points) {
      const Point3<Tp> r1 = T * p, r2 = M * p + T.translation();
      CHECK(std::fabs((double)(r1 - r2).norm()) <= eps * (scale + (double)p.norm()), "%s T*p vs R p + t", name);
      const Point3<Tp> back = T.inverse() * r1;
      CHECK(std::fabs((double)(back - p).norm()) <= 8 * eps * (scale + (double)p.norm()), "%s inverse(T)*(T*p) vs p: %g", name, (double)(back - p).norm());
    }
  }
}

int main() {
  run<double>("double", 1e-10);   // SMALL_EPS of sophus/tests.hpp for double
  run<float>("float", 1e-5);
  if (g_bad) { std::printf("lie_properties: %d failures\n", g_bad); return 1; }
  std::printf("lie_properties: ok\n");
  return 0;
}
