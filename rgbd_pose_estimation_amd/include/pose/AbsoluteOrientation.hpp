// pose/AbsoluteOrientation.hpp -- drop-in for /root/reference/pose/AbsoluteOrientation.hpp.
//
// Same free functions, same argument meaning, results delivered by mutating the adapter:
//   calc_percentage_err, calc_err, shinji, shinji_ransac, shinji_ransac2, shinji_prosac, shinji_ls, shinji_ls1,
//   shinji_ls2, assign_sample, shinji_kneip_ransac, shinji_kneip_prosac
// What moved to the GPU (everything that is O(N)):
//   * the least-squares sums of shinji_ls / _ls1 / _ls2 (reference :279-290, :303-314, :327-336 gather + the two
//     passes of shinji :56-73) -> ONE streaming pass, kernel K1' (rpe_p2p_moments), fp64 accumulation
//   * the vote loops :133-143, :190-200, :248-258, :403-422, :480-499 -> batched scoring, kernel K4 (RansacEngine.hpp)
// What stays on the host (O(1)): sampling, the 3-point shinji(), P3P, the 3x3 SVD, RANSACUpdateNumIters.
#ifndef RPE_AO_POSE_HEADER
#define RPE_AO_POSE_HEADER

#include <vector>
#include "AOPoseAdapter.hpp"
#include "AOOnlyPoseAdapter.hpp"
#include "P3P.hpp"
#include "RansacEngine.hpp"

// percentage errors of the adapter's pose against (R_cw_, t_w_)  (reference :11-27; not quaternion-sign invariant)
template <typename Tp>
rpe::Point3<Tp> calc_percentage_err(const rpe::SO3<Tp>& R_cw_, const rpe::Point3<Tp>& t_w_, const PoseAdapterBase<Tp>* p_ad) {
  const rpe::Point3<Tp> te = R_cw_ * t_w_ - p_ad->getRcw() * p_ad->gettw();
  const Tp t_e = te.norm() / p_ad->gettw().norm() * 100;
  const rpe::Quat<Tp> a = R_cw_.unit_quaternion(), b = p_ad->getRcw().unit_quaternion();
  const Tp dw = a.w - b.w, dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  const Tp r_e = std::sqrt(dw * dw + dx * dx + dy * dy + dz * dz) / std::sqrt(b.w * b.w + b.x * b.x + b.y * b.y + b.z * b.z) * 100;
  return rpe::Point3<Tp>(t_e, r_e, Tp(0));  // (t_e, r_e); third entry unused
}

// (|t|, angle) of SE * GT^-1  (reference :29-43): the "rad / relative-t" parity metric
template <typename Tp>
rpe::Point3<Tp> calc_err(const rpe::SE3<Tp>& GT_cw_, const rpe::SE3<Tp>& SE_cw_) {
  const rpe::Matrix3<Tp> Rd = SE_cw_.so3().matrix() * GT_cw_.so3().matrix().transpose();
  const rpe::Point3<Tp> td = SE_cw_.translation() - Rd * GT_cw_.translation();
  const rpe::Quat<Tp> q = rpe::quat_from_R<Tp>(Rd.a);
  const Tp n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  const Tp angle = n != Tp(0) ? Tp(2) * std::atan2(n, std::fabs(q.w)) : Tp(0);
  return rpe::Point3<Tp>(td.norm(), angle, Tp(0));
}

// Closed-form rigid fit on the first K columns (Umeyama 1991 without scale; reference :47-99), evaluated in Tp in the reference's
// operation order (rpe::rigid_fit, rpe/linalg.hpp): inside RANSAC, K = 3, the hypothesis must be the reference's own Tp values or the
// consensus sets differ at the thresholds.  The O(N) uses go through rpe::pose_from_device_moments instead (one GPU pass, fp64 sums).
// so3().valid() is false where the reference's SO3(Matrix3) constructor would abort (SOPHUS_ENSURE): the solvers skip such a sample.
template <typename Tp>
rpe::SE3<Tp> shinji(const rpe::MatrixX<Tp>& X_w_, const rpe::MatrixX<Tp>& X_c_, int K) {
  Tp q[4], t[3];
  const bool ok = rpe::rigid_fit<Tp>(X_w_.data(), X_c_.data(), K, X_w_.cols(), rpe::LieEps<Tp>::value(), q, t);
  rpe::SO3<Tp> R = rpe::SO3<Tp>::fromQuaternionRaw(q[0], q[1], q[2], q[3]);
  if (!ok) R.invalidate();
  return rpe::SE3<Tp>(R, rpe::Point3<Tp>(t[0], t[1], t[2]));
}

namespace rpe {
// closed-form pose from one pass of kernel K1' over the adapter's HBM-resident arrays
template <typename Tp, class Adapter>
void pose_from_device_moments(Adapter& adapter, int flags) {
  const int N = adapter.getNumberCorrespondences();
  DeviceSet& dev = adapter.device();
  dev.template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  dev.template ensure<Tp>(RPE_XC, adapter.pointsCurrData(), N);
  if (flags & RPE_USE_MASK) adapter.pushMask33();
  double m[18], R[9], t[3];
  check(rpe_p2p_moments(dev.ctx(), flags, m), "rpe_p2p_moments");
  check(rpe_pose_from_moments(m, R, t), "rpe_pose_from_moments");
  Matrix3<Tp> Rt;
  for (int i = 0; i < 9; i++) Rt.a[i] = (Tp)R[i];
  const Quat<Tp> q = quat_from_R<Tp>(Rt.a);
  adapter.setRcw(SO3<Tp>::fromQuaternion(q.w, q.x, q.y, q.z));
  adapter.sett(Point3<Tp>((Tp)t[0], (Tp)t[1], (Tp)t[2]));
}

template <typename Tp, class Adapter>
void shinji_sac(Adapter& adapter, const Tp dist_thre_3d_, int& Iter, Tp confidence, bool prosac, const RunOptions& opt) {
  Rand31& rnd = opt.stream();
  const int N = adapter.getNumberCorrespondences();
  const int K = 3;
  RandomElements<int> re(N);
  if (prosac) { const double t0 = rpe::Settings::get().profile ? rpe::now_us() : 0; adapter.sortIdx(rpe::prosac_prefix(Iter, K));
      if (rpe::Settings::get().profile) rpe::Settings::get().prof.sort += rpe::now_us() - t0; }
  ProsacSampler<Tp> ps(K, N);
  VoteSpec<Tp> spec;
  spec.kind = RPE_VOTE_33; spec.thre_3d = dist_thre_3d_; spec.modalities = 1; spec.model_points = K;
  adapter.device().template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  adapter.device().template ensure<Tp>(RPE_XC, adapter.pointsCurrData(), N);
  MatrixX<Tp> Xw(3, K), Xc(3, K);
  auto gen = [&](std::vector<SE3<Tp> >& out) {
    std::vector<int> sel;
    if (prosac) { ps.sample(&sel, rnd); adapter.getSortedIdx(sel); } else re.run(K, &sel, rnd);
    for (int s = 0; s < K; s++) {
      if (!adapter.isValid(sel[s])) return;  // invalid sample: the reference 'continue's
      Xw.setCol(s, adapter.getPointGlob(sel[s]));
      Xc.setCol(s, adapter.getPointCurr(sel[s]));
    }
    const SE3<Tp> fit = shinji<Tp>(Xw, Xc, K);
    if (fit.so3().valid()) out.push_back(fit);
  };
  auto commit = [&](int cols, unsigned device_cols) { adapter.forgetInlierIdx(); adapter.setInlierFromDevice(cols, device_cols); };
  // plain RANSAC consumes exactly K draws per iteration, so every iteration's position in the random stream is known up front and
  // the whole iteration can run on the device; PROSAC's sampler rejects duplicates (a variable number of draws) and stays on the host
  if (!prosac && Settings::get().device_hypotheses && N >= K && !Settings::get().capture
      && !Settings::get().replay) ransac_engine_device33<Tp>(adapter, spec, gen, commit, Iter, confidence, /*mask_cols=*/2, opt);
  else ransac_engine<Tp>(adapter, spec, gen, commit, Iter, confidence, /*mask_cols=*/2, opt);
  adapter.cvtInlier();
}
}  // namespace rpe

template <typename Tp>
void shinji_ransac(AOPoseAdapter<Tp>& adapter, const Tp dist_thre_3d_, int& Iter, Tp confidence = 0.99,  // reference :101-156
    const rpe::RunOptions& opt = rpe::RunOptions()) {
  rpe::shinji_sac<Tp>(adapter, dist_thre_3d_, Iter, confidence, false, opt);
}
template <typename Tp>
void shinji_ransac2(AOOnlyPoseAdapter<Tp>& adapter, const Tp dist_thre_3d_, int& Iter, Tp confidence = 0.99,  // reference :158-213
    const rpe::RunOptions& opt = rpe::RunOptions()) {
  rpe::shinji_sac<Tp>(adapter, dist_thre_3d_, Iter, confidence, false, opt);
}
template <typename Tp>
void shinji_prosac(AOOnlyPoseAdapter<Tp>& adapter, const Tp dist_thre_3d_, int& Iter, Tp confidence = 0.99,  // reference :215-271
    const rpe::RunOptions& opt = rpe::RunOptions()) {
  rpe::shinji_sac<Tp>(adapter, dist_thre_3d_, Iter, confidence, true, opt);
}

// least squares over the 3D-3D inliers (reference :273-296 / :298-320)
template <typename Tp>
void shinji_ls(AOPoseAdapter<Tp>& adapter) { rpe::pose_from_device_moments<Tp>(adapter, RPE_USE_MASK); }
template <typename Tp>
void shinji_ls1(AOOnlyPoseAdapter<Tp>& adapter) { rpe::pose_from_device_moments<Tp>(adapter, RPE_USE_MASK); }
// least squares over ALL correspondences, no validity test (reference :322-342): what Library.cpp ao() runs
template <typename Tp>
void shinji_ls2(AOOnlyPoseAdapter<Tp>& adapter) { rpe::pose_from_device_moments<Tp>(adapter, 0); }

// fill the 4-sample matrices; true when the first three have valid camera points (reference :344-365)
template <typename Tp>
bool assign_sample(const AOPoseAdapter<Tp>& adapter, const std::vector<int>& selected_cols_, rpe::MatrixX<Tp>* p_X_w_,
                   rpe::MatrixX<Tp>* p_X_c_, rpe::MatrixX<Tp>* p_bv_) {
  const int K = (int)selected_cols_.size() - 1;
  int nValid = 0;
  for (int s = 0; s < K; s++) {
    p_X_w_->setCol(s, adapter.getPointGlob(selected_cols_[s]));
    p_bv_->setCol(s, adapter.getBearingVector(selected_cols_[s]));
    if (adapter.isValid(selected_cols_[s])) { p_X_c_->setCol(s, adapter.getPointCurr(selected_cols_[s])); nValid++; }
  }
  p_X_w_->setCol(3, adapter.getPointGlob(selected_cols_[3]));
  p_bv_->setCol(3, adapter.getBearingVector(selected_cols_[3]));
  return nValid == K;
}

namespace rpe {
template <typename Tp>
void shinji_kneip_sac(AOPoseAdapter<Tp>& adapter, const Tp dist_thre_3d_, const Tp thre_2d_, int& Iter, Tp confidence, bool prosac,
    const RunOptions& opt) {
  Rand31& rnd = opt.stream();
  const int N = adapter.getNumberCorrespondences();
  const int K = 3;
  RandomElements<int> re(N);
  if (prosac) { const double t0 = rpe::Settings::get().profile ? rpe::now_us() : 0; adapter.sortIdx(rpe::prosac_prefix(Iter, K + 1));
      if (rpe::Settings::get().profile) rpe::Settings::get().prof.sort += rpe::now_us() - t0; }
  ProsacSampler<Tp> ps(K + 1, N);
  VoteSpec<Tp> spec;
  spec.kind = RPE_VOTE_33_23; spec.thre_3d = dist_thre_3d_;
  spec.cos_thr = std::cos(std::atan(thre_2d_ / adapter.getFocal()));
  spec.modalities = 2; spec.model_points = K;
  DeviceSet& dev = adapter.device();
  dev.template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  dev.template ensure<Tp>(RPE_XC, adapter.pointsCurrData(), N);
  dev.template ensure<Tp>(RPE_BV, adapter.bearingData(), N);
  MatrixX<Tp> X_w(3, K + 1), X_c(3, K + 1), bv(3, K + 1);
  auto gen = [&](std::vector<SE3<Tp> >& out) {
    std::vector<int> sel;
    if (prosac) { ps.sample(&sel, rnd); adapter.getSortedIdx(sel); } else re.run(K + 1, &sel, rnd);
    if (assign_sample<Tp>(adapter, sel, &X_w, &X_c, &bv)) { const SE3<Tp> fit = shinji<Tp>(X_w, X_c, K);
        if (fit.so3().valid()) out.push_back(fit); }
    SE3<Tp> sk;
    if (kneip<Tp>(X_w, bv, &sk)) out.push_back(sk);
  };
  auto commit = [&](int cols, unsigned device_cols) {
    { PnPPoseAdapter<Tp>* p23 = &adapter; p23->forgetInlierIdx(); adapter.forgetInlierIdx(); }  // both are requested again below
    adapter.setInlierFromDevice(cols, device_cols);
  };
  const Settings& cfg = Settings::get();
  if (!prosac && opt.mode() == RPE_SCORE_FAST && cfg.device_hypotheses && N >= K + 1 && !cfg.capture && !cfg.replay)
    // FAST mode: later batches generated on the device
    ransac_engine_device_p3p<Tp>(adapter, spec, /*solver=*/1, gen, commit, Iter, confidence, /*mask_cols=*/2, opt);
  else
    ransac_engine<Tp>(adapter, spec, gen, commit, Iter, confidence, /*mask_cols=*/2, opt);
  PnPPoseAdapter<Tp>* pAdapter = &adapter;
  pAdapter->cvtInlier();
  adapter.cvtInlier();
}
}  // namespace rpe

template <typename Tp>
// :367-438
void shinji_kneip_ransac(AOPoseAdapter<Tp>& adapter, const Tp dist_thre_3d_, const Tp thre_2d_, int& Iter, Tp confidence = 0.99,
    const rpe::RunOptions& opt = rpe::RunOptions()) {
  rpe::shinji_kneip_sac<Tp>(adapter, dist_thre_3d_, thre_2d_, Iter, confidence, false, opt);
}
template <typename Tp>
// :440-515
void shinji_kneip_prosac(AOPoseAdapter<Tp>& adapter, const Tp dist_thre_3d_, const Tp thre_2d_, int& Iter, Tp confidence = 0.99,
    const rpe::RunOptions& opt = rpe::RunOptions()) {
  rpe::shinji_kneip_sac<Tp>(adapter, dist_thre_3d_, thre_2d_, Iter, confidence, true, opt);
}

#endif
