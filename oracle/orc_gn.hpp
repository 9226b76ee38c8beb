// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header).
//
// orc_gn.hpp: fp64 CPU statement of the Gauss-Newton formulation the north star asks for.  The
// reference has NO Jacobian / Gauss-Newton code, so there is no reference text to
// follow line by line; what pins each objective is:
//   point-to-point   same objective as shinji()  /root/reference/pose/AbsoluteOrientation.hpp:47-99
//                    => the converged GN pose must equal the closed-form pose (checked in tests)
//   point-to-plane   no reference counterpart (F2): PARITY UNPINNED, pinned by this fp64 GN + numpy
//   bearing (sine)   residual definition |normalize(R Xw + t) x bv|  pose/P3P.hpp:482-485,
//                    pose/PnPPoseAdapter.hpp:204-210 ; the refinement itself is new (F3)
//   normal-normal    r = R Nw - Nc: the alignment the reference scores with Nc.(R Nw) (AbsoluteOrientationNormal.hpp:248)
//                    and fits through MNN in nl_shinji_kneip_ls (:498-503); as a GN term it is new
//   reprojection     r = (p_x/p_z - bv_x/bv_z, p_y/p_z - bv_y/bv_z): the pixel residual of SURVEY.md Appendix B row 4 in normalised
//                    image coordinates (f = 1; pixel conversion /root/reference/TestMain.cpp:35-36, principal point at the origin:
//                    pose/PoseAdapterBase.hpp:44); no reference refinement uses it: PARITY UNPINNED, pinned by numerical Jacobians
// Conventions: Xc = R Xw + t ; left perturbation T <- exp(delta) T, delta = (upsilon, omega) in the
// Sophus order (sophus/se3.hpp:314-316) ; p = R Xw + t ; dp/ddelta = [ I | -[p]x ].
// Output layout (29 doubles): H upper triangle row-major (21) | g = J^T r (6) | sum w r^2 | sum w.
#pragma once
#include "orc_linalg.hpp"

namespace orc {

enum GnKind { GN_P2P = 0, GN_P2PLANE = 1, GN_BEARING = 2, GN_NORMAL = 3, GN_REPROJ = 4 };
// pixel reprojection (GN_REPROJ): a correspondence whose point is not in front of the camera or whose bearing has no forward component
constexpr double kReprojMinZ = 1e-6;
enum GnRobust { ROBUST_NONE = 0, ROBUST_HUBER = 1, ROBUST_CAUCHY = 2 };
// IRLS weight of a residual block of norm s: Huber min(1, k/s), Cauchy 1/(1 + (s/k)^2)
inline double robust_weight(int robust, double k, double s) {
  if (robust == ROBUST_HUBER) return s <= k ? 1.0 : k / s;
  if (robust == ROBUST_CAUCHY) return 1.0 / (1.0 + (s / k) * (s / k));
  return 1.0;
}

struct NormalEq {
  double H[6][6];
  double g[6];
  double cost, wsum;
  NormalEq() { std::memset(this, 0, sizeof(*this)); }
  void add_row(const double J[6], double r, double w) {
    for (int a = 0; a < 6; a++) {
      g[a] += w * J[a] * r;
      for (int b = a; b < 6; b++) H[a][b] += w * J[a] * J[b];
    }
    cost += w * r * r;
  }
  void pack(double out[29]) const {
    int k = 0;
    for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) out[k++] = H[a][b];
    for (int a = 0; a < 6; a++) out[k++] = g[a];
    out[k++] = cost; out[k++] = wsum;
  }
};

// pose: R row-major 9 + t 3 (double).  a = Xw, b = Xc (p2p/p2plane) or bv (bearing), c = Nc (p2plane).
// mask (short, may be null): only entries == 1 contribute.  weight (may be null): per-correspondence w.
// NaN-invalid columns of b (all three NaN, the reference's isValid) are skipped.
template <class Tin>
void gn_normal_eq(int kind, const Tin* a, const Tin* b, const Tin* c, const short* mask, const Tin* weight, long n,
                  const double pose[12], NormalEq* ne, int robust = ROBUST_NONE, double robust_k = 1.0) {
  const double* R = pose; const double* t = pose + 9;
  for (long i = 0; i < n; i++) {
    if (mask && mask[i] != 1) continue;
    double bx = b[3 * i], by = b[3 * i + 1], bz = b[3 * i + 2];
    if (bx != bx && by != by && bz != bz) continue;
    double w = weight ? (double)weight[i] : 1.0;
    double x = a[3 * i], y = a[3 * i + 1], z = a[3 * i + 2];
    double p[3] = {R[0] * x + R[1] * y + R[2] * z + t[0], R[3] * x + R[4] * y + R[5] * z + t[1], R[6] * x + R[7] * y + R[8] * z + t[2]};
    // Jp = [I | -[p]x] : rows of dp/ddelta
    const double Jp[3][6] = {{1, 0, 0, 0, p[2], -p[1]}, {0, 1, 0, -p[2], 0, p[0]}, {0, 0, 1, p[1], -p[0], 0}};
    if (kind == GN_P2P) {
      double r[3] = {p[0] - bx, p[1] - by, p[2] - bz};
      w *= robust_weight(robust, robust_k, std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]));
      for (int k = 0; k < 3; k++) ne->add_row(Jp[k], r[k], w);
    } else if (kind == GN_P2PLANE) {
      double nx = c[3 * i], ny = c[3 * i + 1], nz = c[3 * i + 2];
      double r = nx * (p[0] - bx) + ny * (p[1] - by) + nz * (p[2] - bz);
      w *= robust_weight(robust, robust_k, std::fabs(r));
      double J[6];
      for (int k = 0; k < 6; k++) J[k] = nx * Jp[0][k] + ny * Jp[1][k] + nz * Jp[2][k];
      ne->add_row(J, r, w);
    } else if (kind == GN_REPROJ) {
      if (!(p[2] > kReprojMinZ) || !(bz > kReprojMinZ)) continue;   // not in front of the camera: contributes nothing, does not count
      const double u = p[0] / p[2], v = p[1] / p[2];
      const double r[2] = {u - bx / bz, v - by / bz};
      w *= robust_weight(robust, robust_k, std::sqrt(r[0] * r[0] + r[1] * r[1]));
      // d(u, v)/dp = 1/p_z [[1, 0, -u], [0, 1, -v]]
      const double A[2][3] = {{1.0 / p[2], 0.0, -u / p[2]}, {0.0, 1.0 / p[2], -v / p[2]}};
      for (int q = 0; q < 2; q++) {
        double J[6];
        for (int k = 0; k < 6; k++) J[k] = A[q][0] * Jp[0][k] + A[q][1] * Jp[1][k] + A[q][2] * Jp[2][k];
        ne->add_row(J, r[q], w);
      }
    } else if (kind == GN_NORMAL) {
      // a = Nw, b = Nc ; q = R Nw ; r = q - Nc ; dq/ddelta = [0 | -[q]x]
      double q[3] = {R[0] * x + R[1] * y + R[2] * z, R[3] * x + R[4] * y + R[5] * z, R[6] * x + R[7] * y + R[8] * z};
      double r[3] = {q[0] - bx, q[1] - by, q[2] - bz};
      w *= robust_weight(robust, robust_k, std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]));
      const double Jq[3][6] = {{0, 0, 0, 0, q[2], -q[1]}, {0, 0, 0, -q[2], 0, q[0]}, {0, 0, 0, q[1], -q[0], 0}};
      for (int k = 0; k < 3; k++) ne->add_row(Jq[k], r[k], w);
    } else {
      double len = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
      double ph[3] = {p[0] / len, p[1] / len, p[2] / len};
      double r[3] = {ph[1] * bz - ph[2] * by, ph[2] * bx - ph[0] * bz, ph[0] * by - ph[1] * bx};  // ph x bv
      w *= robust_weight(robust, robust_k, std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]));
      // A = -[bv]x (I - ph ph^T) / len   (3x3), J = A Jp
      double P[3][3], A[3][3];
      for (int u = 0; u < 3; u++) for (int v = 0; v < 3; v++) P[u][v] = ((u == v ? 1.0 : 0.0) - ph[u] * ph[v]) / len;
      const double Bx[3][3] = {{0, -bz, by}, {bz, 0, -bx}, {-by, bx, 0}};
      for (int u = 0; u < 3; u++) for (int v = 0; v < 3; v++) {
        double s = 0; for (int k = 0; k < 3; k++) s += Bx[u][k] * P[k][v];
        A[u][v] = -s;
      }
      for (int u = 0; u < 3; u++) {
        double J[6];
        for (int k = 0; k < 6; k++) J[k] = A[u][0] * Jp[0][k] + A[u][1] * Jp[1][k] + A[u][2] * Jp[2][k];
        ne->add_row(J, r[u], w);
      }
    }
    ne->wsum += w;
  }
}

// solve H delta = -g (H symmetric positive definite, upper triangle given) by Cholesky; returns false if not SPD
inline bool gn_solve6(const double Hup[6][6], const double g[6], double delta[6]) {
  double L[6][6] = {{0}};
  for (int i = 0; i < 6; i++) {
    for (int j = 0; j <= i; j++) {
      double s = Hup[j][i];
      for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
      if (i == j) { if (!(s > 0)) return false; L[i][i] = std::sqrt(s); }
      else L[i][j] = s / L[j][j];
    }
  }
  double y[6];
  for (int i = 0; i < 6; i++) { double s = -g[i]; for (int k = 0; k < i; k++) s -= L[i][k] * y[k]; y[i] = s / L[i][i]; }
  for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[k][i] * delta[k]; delta[i] = s / L[i][i]; }
  return true;
}

inline void gn_apply(const double delta[6], double pose[12]) {  // T <- exp(delta) T
  SE3<double> d = SE3<double>::exp(delta);
  M3<double> Rd = d.R.matrix(), R;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = pose[3 * i + j];
  V3<double> t(pose[9], pose[10], pose[11]);
  M3<double> Rn = Rd * R;
  V3<double> tn = Rd * t + d.t;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) pose[3 * i + j] = Rn(i, j);
  pose[9] = tn.x; pose[10] = tn.y; pose[11] = tn.z;
}

// terms: up to 3 residual blocks summed into one normal equation (joint 3D-3D + 2D-3D refinement etc.)
struct GnTerm { int kind; const void* a; const void* b; const void* c; const short* mask; const void* weight; double scale; int robust; double robust_k; };

template <class Tin>
int gn_refine(const GnTerm* terms, int nterms, long n, double pose[12], int max_iter, double tol, double* last_step, double* final_cost) {
  int it = 0;
  double step = 0, cost = 0;
  for (; it < max_iter; it++) {
    NormalEq tot;
    for (int k = 0; k < nterms; k++) {
      NormalEq ne;
      gn_normal_eq<Tin>(terms[k].kind, (const Tin*)terms[k].a, (const Tin*)terms[k].b, (const Tin*)terms[k].c, terms[k].mask,
                        (const Tin*)terms[k].weight, n, pose, &ne, terms[k].robust, terms[k].robust_k);
      double s = terms[k].scale;
      for (int a = 0; a < 6; a++) { tot.g[a] += s * ne.g[a]; for (int b = a; b < 6; b++) tot.H[a][b] += s * ne.H[a][b]; }
      tot.cost += s * ne.cost; tot.wsum += ne.wsum;
    }
    cost = tot.cost;
    double delta[6];
    if (!gn_solve6(tot.H, tot.g, delta)) { it = -1 - it; break; }
    gn_apply(delta, pose);
    step = 0; for (int a = 0; a < 6; a++) step += delta[a] * delta[a];
    step = std::sqrt(step);
    if (step < tol) { it++; break; }
  }
  if (last_step) *last_step = step;
  if (final_cost) *final_cost = cost;
  return it;
}

}  // namespace orc
