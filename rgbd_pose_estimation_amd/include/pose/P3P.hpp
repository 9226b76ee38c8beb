// pose/P3P.hpp -- drop-in for /root/reference/pose/P3P.hpp: the P3P minimal solver (host, O(1) per hypothesis),
// the adaptive RANSAC iteration bound, and kneip_ransac / kneip_prosac whose O(Iter x N) vote loop
// (reference :362-376, :439-453) runs on the GPU as batched hypothesis scoring (rpe_score, kernel K4).
//
// Functions kept, same names and argument meaning:
//   o4_roots, kneip_main, kneip (both overloads), RANSACUpdateNumIters, kneip_ransac, kneip_prosac, lsq_pnp
#ifndef RPE_P3P_POSE_HEADER
#define RPE_P3P_POSE_HEADER

#include <complex>
#include <limits>
#include <vector>
#include "Utility.hpp"
#include "PnPPoseAdapter.hpp"
#include "RansacEngine.hpp"

// Real parts of the four roots of a[0] x^4 + a[1] x^3 + a[2] x^2 + a[3] x + a[4] by Ferrari's method with complex
// intermediates (reference :11-60): depressed quartic y^4 + alpha y^2 + beta y + gamma, resolvent cubic via Cardano.
// The P3P hypotheses feed threshold tests, so the evaluation keeps the reference's operation order in Tp: powers by repeated
// multiplication, every quotient formed as there, and -- a C++ detail of the reference -- pow(beta, 2) / 8 in the resolvent is a
// double expression (pow(Tp, int) promotes), so for Tp = float that term is subtracted in double before rounding.
template <typename Tp>
std::vector<Tp> o4_roots(const Tp a[5]) {
  typedef std::complex<Tp> Cx;
  const Tp A = a[0], B = a[1], C = a[2], D = a[3], E = a[4];
  const Tp A2 = A * A, B2 = B * B;
  const Tp A3 = A2 * A, B3 = B2 * B;
  const Tp A4 = A3 * A, B4 = B3 * B;
  const Tp alpha = -3 * B2 / (8 * A2) + C / A;
  const Tp beta = B3 / (8 * A3) - B * C / (2 * A2) + D / A;
  const Tp gamma = -3 * B4 / (256 * A4) + B2 * C / (16 * A3) - B * D / (4 * A2) + E / A;
  const Tp alpha2 = alpha * alpha, alpha3 = alpha2 * alpha;
  const Cx P(-alpha2 / 12 - gamma, 0);
  const Cx Q((Tp)((double)(-alpha3 / 108 + alpha * gamma / 3) - (double)beta * (double)beta / 8), 0);
  const Cx R = -Q / Tp(2.0) + std::sqrt(std::pow(Q, Tp(2.)) / Tp(4.) + std::pow(P, Tp(3.)) / Tp(27.));
  const Cx U = std::pow(R, Tp(1.0 / 3.0));
  Cx y;
  if (U.real() == 0) y = -Tp(5.0) * alpha / Tp(6.) - std::pow(Q, Tp(1.0 / 3.0));
  else y = -Tp(5.0) * alpha / Tp(6.) - P / (Tp(3.) * U) + U;
  const Cx w = std::sqrt(alpha + Tp(2.) * y);
  const Cx up = std::sqrt(-(Tp(3.) * alpha + Tp(2.) * y + Tp(2.) * beta / w));
  const Cx um = std::sqrt(-(Tp(3.) * alpha + Tp(2.) * y - Tp(2.) * beta / w));
  const Tp shift = -B / (Tp(4.) * A);
  std::vector<Tp> roots(4);
  roots[0] = (shift + Tp(0.5) * (w + up)).real();
  roots[1] = (shift + Tp(0.5) * (w - up)).real();
  roots[2] = (shift + Tp(0.5) * (-w + um)).real();
  roots[3] = (shift + Tp(0.5) * (-w - um)).real();
  return roots;
}

// Kneip, Scaramuzza, Siegwart: "A novel parametrization of the P3P problem" (CVPR 2011); reference :63-232.
// X_w, bv: 3 x (>= 3) column-major; up to four (R_cw, t) with Xc = R Xw + t.  Operation order of the reference throughout
// (products and sums left to right as written there; `1 / (1 - cos_beta^2) - 1` in double because pow(Tp, int) promotes).
template <typename Tp>
void kneip_main(const rpe::MatrixX<Tp>& X_w, const rpe::MatrixX<Tp>& bv, std::vector<rpe::SE3<Tp> >* p_solutions_) {
  typedef rpe::Point3<Tp> V3;
  typedef rpe::Matrix3<Tp> M3;
  p_solutions_->clear();
  V3 P1 = X_w.col(0), P2 = X_w.col(1), P3 = X_w.col(2);
  const V3 edge12 = P2 - P1;                           // kept from BEFORE any swap (:129 reads its norm afterwards)
  if (edge12.cross(P3 - P1).norm() == 0) return;       // collinear world points
  V3 f1 = bv.col(0), f2 = bv.col(1), f3 = bv.col(2);

  // intermediate camera frame tau = (f1, (f1 x f2) x f1, f1 x f2), rows of Tcam
  M3 Tcam;
  auto camera_frame = [&]() {
    V3 e3 = f1.cross(f2);
    e3 = e3 / e3.norm();
    const V3 e2 = e3.cross(f1);
    Tcam.setRow(0, f1); Tcam.setRow(1, e2); Tcam.setRow(2, e3);
    f3 = Tcam * f3;
  };
  camera_frame();
  if (f3[2] > 0) {  // keep theta in [0, pi]: swap the roles of points 1 and 2
    f1 = bv.col(1); f2 = bv.col(0); f3 = bv.col(2);
    camera_frame();
    P1 = X_w.col(1); P2 = X_w.col(0); P3 = X_w.col(2);
  }
  // intermediate world frame eta, rows of Nw
  V3 n1 = P2 - P1;
  n1 = n1 / n1.norm();
  V3 n3 = n1.cross(P3 - P1);
  n3 = n3 / n3.norm();
  const V3 n2 = n3.cross(n1);
  M3 Nw; Nw.setRow(0, n1); Nw.setRow(1, n2); Nw.setRow(2, n3);
  P3 = Nw * (P3 - P1);

  const Tp d12 = edge12.norm();
  const Tp f_1 = f3[0] / f3[2], f_2 = f3[1] / f3[2], p_1 = P3[0], p_2 = P3[1];
  const Tp cos_beta = f1.dot(f2);
  Tp b = (Tp)(1 / (1 - (double)cos_beta * (double)cos_beta) - 1);
  b = cos_beta < 0 ? -std::sqrt(b) : std::sqrt(b);

  // quartic in cos(theta): the reference's 5 coefficient sums, term by term
  const Tp f1s = f_1 * f_1, f2s = f_2 * f_2;
  const Tp p1s = p_1 * p_1, p1c = p1s * p_1, p1q = p1c * p_1;
  const Tp p2s = p_2 * p_2, p2c = p2s * p_2, p2q = p2c * p_2;
  const Tp ds = d12 * d12, bs = b * b;
  Tp q[5];
  q[0] = -f2s * p2q - p2q * f1s - p2q;
  q[1] = 2 * p2c * d12 * b + 2 * f2s * p2c * d12 * b - 2 * f_2 * p2c * f_1 * d12;
  q[2] = -f2s * p2s * p1s - f2s * p2s * ds * bs - f2s * p2s * ds + f2s * p2q + p2q * f1s + 2 * p_1 * p2s * d12 +
         2 * f_1 * f_2 * p_1 * p2s * d12 * b - p2s * p1s * f1s + 2 * p_1 * p2s * f2s * d12 - p2s * ds * bs - 2 * p1s * p2s;
  q[3] = 2 * p1s * p_2 * d12 * b + 2 * f_2 * p2c * f_1 * d12 - 2 * f2s * p2c * d12 * b - 2 * p_1 * p_2 * ds * b;
  q[4] = -2 * f_2 * p2s * f_1 * p_1 * d12 * b + f2s * p2s * ds + 2 * p1c * d12 - p1s * ds + f2s * p2s * p1s - p1q -
         2 * f2s * p2s * p_1 * d12 + p2s * f1s * p1s + f2s * p2s * ds * bs;
  const std::vector<Tp> roots = o4_roots<Tp>(q);

  for (int i = 0; i < 4; i++) {
    const Tp cos_theta = roots[i];
    if (cos_theta != cos_theta) continue;
    const Tp cot_alpha = (-f_1 * p_1 / f_2 - cos_theta * p_2 + d12 * b) / (-f_1 * cos_theta * p_2 / f_2 + p_1 - d12);
    if (cos_theta > Tp(1) || cos_theta < Tp(-1)) continue;
    const Tp sin_theta = std::sqrt(1 - cos_theta * cos_theta);
    const Tp sin_alpha = std::sqrt(1 / (cot_alpha * cot_alpha + 1));
    Tp cos_alpha = std::sqrt(1 - sin_alpha * sin_alpha);
    if (cot_alpha < 0) cos_alpha = -cos_alpha;
    V3 C(d12 * cos_alpha * (sin_alpha * b + cos_alpha), cos_theta * d12 * sin_alpha * (sin_alpha * b + cos_alpha),
         sin_theta * d12 * sin_alpha * (sin_alpha * b + cos_alpha));
    C = P1 + Nw.transpose() * C;  // camera centre in the world frame
    M3 Q;
    Q(0, 0) = -cos_alpha; Q(0, 1) = -sin_alpha * cos_theta; Q(0, 2) = -sin_alpha * sin_theta;
    Q(1, 0) = sin_alpha;  Q(1, 1) = -cos_alpha * cos_theta; Q(1, 2) = -cos_alpha * sin_theta;
    Q(2, 0) = Tp(0);      Q(2, 1) = -sin_theta;             Q(2, 2) = cos_theta;
    const M3 R = Tcam.transpose() * Q * Nw;
    if (R(0, 0) != R(0, 0)) continue;
    rpe::SO3<Tp> so3(R);
    if (!so3.valid()) continue;  // the reference would abort here (SOPHUS_ENSURE)
    p_solutions_->push_back(rpe::SE3<Tp>(so3, -(R * C)));
  }
}

template <typename Tp>
std::vector<rpe::SE3<Tp> > kneip(PnPPoseAdapter<Tp>& adapter, int i0 = 0, int i1 = 1, int i2 = 2) {  // reference :234-248
  rpe::MatrixX<Tp> bv(3, 3), X_w(3, 3);
  const int idx[3] = {i0, i1, i2};
  for (int k = 0; k < 3; k++) { bv.setCol(k, adapter.getBearingVector(idx[k])); X_w.setCol(k, adapter.getPointGlob(idx[k])); }
  std::vector<rpe::SE3<Tp> > solutions;
  kneip_main<Tp>(X_w, bv, &solutions);
  return solutions;
}

// pick the P3P branch that best reprojects the 4th correspondence (reference :250-294)
template <typename Tp>
bool kneip(const rpe::MatrixX<Tp>& X_w_, const rpe::MatrixX<Tp>& bv_, rpe::SE3<Tp>* p_sol_) {
  std::vector<rpe::SE3<Tp> > cand;
  kneip_main<Tp>(X_w_, bv_, &cand);
  Tp best = std::numeric_limits<Tp>::max();
  int arg = -1;
  for (int i = 0; i < (int)cand.size(); i++) {
    rpe::Point3<Tp> pc = cand[i].so3().matrix() * X_w_.col(3) + cand[i].translation();
    pc = pc / pc.norm();
    const Tp score = Tp(1.0) - pc.dot(bv_.col(3));
    if (score < best) { best = score; arg = i; }
  }
  if (arg < 0) return false;
  *p_sol_ = cand[arg];
  return true;
}

// OpenCV-style adaptive iteration bound (reference :296-318)
template <typename T>
int RANSACUpdateNumIters(T p, T ep, const int modelPoints, const int maxIters) {
  p = std::min(std::max(p, T(0.)), T(1.));
  ep = std::min(std::max(ep, T(0.)), T(1.));
  T num = std::max(T(1. - p), std::numeric_limits<T>::epsilon());
  T denom = T(1.) - std::pow(T(1. - ep), modelPoints);
  if (denom < std::numeric_limits<T>::epsilon()) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : int(num / denom + 0.5f);
}

namespace rpe {
template <typename Tp>
void kneip_sac(PnPPoseAdapter<Tp>& adapter, const Tp thre_2d_, int& Iter, Tp confidence, bool prosac, const RunOptions& opt) {
  Rand31& rnd = opt.stream();
  const int N = adapter.getNumberCorrespondences();
  const int K = 4;
  RandomElements<int> re(N);
  if (prosac) { const double t0 = rpe::Settings::get().profile ? rpe::now_us() : 0; adapter.sortIdx(rpe::prosac_prefix(Iter, K));
      if (rpe::Settings::get().profile) rpe::Settings::get().prof.sort += rpe::now_us() - t0; }
  ProsacSampler<Tp> ps(K, N);
  VoteSpec<Tp> spec;
  // kneip_ransac multiplies by so3().matrix() (:365) and kneip_prosac by so3() (:442): two arithmetic variants
  spec.kind = prosac ? RPE_VOTE_23 : RPE_VOTE_23_MATRIX;
  spec.cos_thr = std::cos(std::atan(thre_2d_ / adapter.getFocal()));
  spec.modalities = 1; spec.model_points = K;
  adapter.device().template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  adapter.device().template ensure<Tp>(RPE_BV, adapter.bearingData(), N);
  auto gen = [&](std::vector<SE3<Tp> >& out) {
    std::vector<int> sel;
    if (prosac) { ps.sample(&sel, rnd); adapter.getSortedIdx(sel); } else re.run(K, &sel, rnd);
    std::vector<SE3<Tp> > sols = kneip<Tp>(adapter, sel[0], sel[1], sel[2]);
    Tp best = Tp(1000000.0);
    int arg = -1;
    for (int i = 0; i < (int)sols.size(); i++) {
      Point3<Tp> pc = sols[i].so3().matrix() * adapter.getPointGlob(sel[3]) + sols[i].translation();
      pc = pc / pc.norm();
      const Tp score = Tp(1.0) - pc.dot(adapter.getBearingVector(sel[3]));
      if (score < best) { best = score; arg = i; }
    }
    if (arg >= 0) out.push_back(sols[arg]);
  };
  auto commit = [&](int cols, unsigned device_cols) { adapter.forgetInlierIdx(); adapter.setInlierFromDevice(cols, device_cols); };
  // plain RANSAC in FAST scoring mode: batches beyond the first few are generated on the device too (4 draws per iteration: every
  // iteration's position in the random stream is known up front)
  const Settings& cfg = Settings::get();
  if (!prosac && opt.mode() == RPE_SCORE_FAST && cfg.device_hypotheses && N >= K && !cfg.capture && !cfg.replay)
    ransac_engine_device_p3p<Tp>(adapter, spec, /*solver=*/0, gen, commit, Iter, confidence, /*mask_cols=*/1, opt);
  else
    ransac_engine<Tp>(adapter, spec, gen, commit, Iter, confidence, /*mask_cols=*/1, opt);
  adapter.cvtInlier();
}
}  // namespace rpe

template <typename Tp>
void kneip_ransac(PnPPoseAdapter<Tp>& adapter, const Tp thre_2d_, int& Iter, Tp confidence = 0.99,  // reference :320-392
    const rpe::RunOptions& opt = rpe::RunOptions()) {
  rpe::kneip_sac<Tp>(adapter, thre_2d_, Iter, confidence, false, opt);
}
template <typename Tp>
void kneip_prosac(PnPPoseAdapter<Tp>& adapter, const Tp thre_2d_, int& Iter, Tp confidence = 0.99,  // reference :395-469
    const rpe::RunOptions& opt = rpe::RunOptions()) {
  rpe::kneip_sac<Tp>(adapter, thre_2d_, Iter, confidence, true, opt);
}

// Sum of the per-correspondence sine residuals at the adapter's pose (reference :472-502 prints it; returned here): ONE pass of the
// device kernel over the adapter's HBM-resident arrays (rpe_sine_error_sum, csrc/rpe_nl.hip sine_error_kernel).  Every term is
// getError(i) bit for bit -- the reference's operation sequence in Tp; the terms are added in fp64 on the device, so the total differs
// from the reference's sequential Tp sum only by that sum's own accumulated rounding.
template <typename Tp>
Tp lsq_pnp(PnPPoseAdapter<Tp>& adapter) {
  const int N = adapter.getNumberCorrespondences();
  if (N <= 0) return Tp(0);
  rpe::DeviceSet& dev = adapter.device();
  dev.template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  dev.template ensure<Tp>(RPE_BV, adapter.bearingData(), N);
  double q7[7], total = 0;
  rpe::pose7<Tp>(rpe::SE3<Tp>(adapter.getRcw(), adapter.gettw()), q7);
  rpe::check(rpe_sine_error_sum(dev.ctx(), q7, &total, nullptr), "rpe_sine_error_sum");
  return (Tp)total;
}

#endif
