// pose/AbsoluteOrientationNormal.hpp -- drop-in for /root/reference/pose/AbsoluteOrientationNormal.hpp
// (normal-aware solvers).  Kept: find_opt_cc, assign_sample, nl_2p, nl_kneip_ransac, nl_shinji_ransac,
// nl_shinji_kneip_ransac, nl_shinji_kneip_ls.  Not kept: nl_shinji_ls (reference :145-213), which is dead code that
// does not compile when instantiated (assigns a 3x3 product to a 3x1 at :195).
//
// GPU work: the three vote loops (:245-264, :322-337, :397-423) -> kernel K4; every O(N) pass of
// nl_shinji_kneip_ls (:457-469 centroids, :484-505 the M23/M33/MNN sums, :24-39 find_opt_cc) -> kernels K1' and K5.
#ifndef RPE_AO_NORM_POSE_HEADER
#define RPE_AO_NORM_POSE_HEADER

#include <limits>
#include <vector>
#include "NormalAOPoseAdapter.hpp"
#include "AbsoluteOrientation.hpp"

namespace rpe {
template <typename Tp>
void ensure_all_arrays(NormalAOPoseAdapter<Tp>& adapter) {
  const int N = adapter.getNumberCorrespondences();
  DeviceSet& dev = adapter.device();
  dev.template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  dev.template ensure<Tp>(RPE_XC, adapter.pointsCurrData(), N);
  dev.template ensure<Tp>(RPE_BV, adapter.bearingData(), N);
  dev.template ensure<Tp>(RPE_NW, adapter.normalGlobData(), N);
  dev.template ensure<Tp>(RPE_NC, adapter.normalCurrData(), N);
}
template <typename Tp>
void sync_masks_and_weights(NormalAOPoseAdapter<Tp>& adapter) {
  DeviceSet& dev = adapter.device();
  adapter.pushMask23();
  adapter.pushMask33();
  adapter.pushMaskNN();
  dev.template upload_weight<Tp>(RPE_MOD_23, adapter.weights23(), Tp(1));
  dev.template upload_weight<Tp>(RPE_MOD_33, adapter.weights33(), adapter.weightScale33());
  dev.template upload_weight<Tp>(RPE_MOD_NN, adapter.weightsNN(), adapter.weightScaleNN());
}
// one fused pass of kernel K5; returns the 44-value record described in include/rgbd_pose_hip.h
template <typename Tp>
void nl_round_on_device(NormalAOPoseAdapter<Tp>& adapter, const Point3<Tp>& c_opt, const Point3<Tp>& Cw, const Point3<Tp>& Cc,
    double out44[44]) {
  ensure_all_arrays<Tp>(adapter);
  sync_masks_and_weights<Tp>(adapter);
  const Matrix3<Tp> Rwc = adapter.getRcw().inverse().matrix();
  double c3[3], cw3[3], cc3[3], R9[9];
  for (int i = 0; i < 3; i++) { c3[i] = c_opt[i]; cw3[i] = Cw[i]; cc3[i] = Cc[i]; }
  for (int i = 0; i < 9; i++) R9[i] = Rwc.a[i];
  check(rpe_nl_round(adapter.device().ctx(), c3, cw3, cc3, R9, out44), "rpe_nl_round");
}
template <typename Tp>
Point3<Tp> cc_from_record(const double* r) {
  Mat3d AA;
  AA(0, 0) = r[32]; AA(0, 1) = AA(1, 0) = r[33]; AA(0, 2) = AA(2, 0) = r[34]; AA(1, 1) = r[35]; AA(1, 2) = AA(2, 1) = r[36]; AA(2, 2) = r[37];
  const Vec3d bb(r[38], r[39], r[40]);
  if (std::fabs(det3(AA)) < 0.0001) {
    const Tp nan = std::numeric_limits<Tp>::quiet_NaN();
    return Point3<Tp>(nan, nan, nan);
  }
  const Vec3d c = svd_solve3(AA, bb);
  return Point3<Tp>((Tp)c[0], (Tp)c[1], (Tp)c[2]);
}
}  // namespace rpe

// Optimal camera centre for a fixed rotation from the 2D-3D inliers (Slabaugh et al. 2001; reference :13-46):
// AA = sum (I - v v^T), bb = sum (I - v v^T) Xw, v = R_wc bv; NaN vector when |det AA| < 1e-4.
template <typename Tp>
rpe::Point3<Tp> find_opt_cc(NormalAOPoseAdapter<Tp>& adapter) {
  double rec[44];
  rpe::nl_round_on_device<Tp>(adapter, rpe::Point3<Tp>(), rpe::Point3<Tp>(), rpe::Point3<Tp>(), rec);
  return rpe::cc_from_record<Tp>(rec);
}

template <typename Tp>
bool assign_sample(const NormalAOPoseAdapter<Tp>& adapter, const std::vector<int>& selected_cols_, rpe::MatrixX<Tp>* p_X_w_,
                   // reference :48-75
                   rpe::MatrixX<Tp>* p_N_w_, rpe::MatrixX<Tp>* p_X_c_, rpe::MatrixX<Tp>* p_N_c_, rpe::MatrixX<Tp>* p_bv_) {
  const int K = (int)selected_cols_.size() - 1;
  int nValid = 0;
  for (int s = 0; s < K; s++) {
    const int c = selected_cols_[s];
    p_X_w_->setCol(s, adapter.getPointGlob(c));
    p_N_w_->setCol(s, adapter.getNormalGlob(c));
    p_bv_->setCol(s, adapter.getBearingVector(c));
    if (adapter.isValid(c)) { p_X_c_->setCol(s, adapter.getPointCurr(c)); p_N_c_->setCol(s, adapter.getNormalCurr(c)); nValid++; }
  }
  p_X_w_->setCol(3, adapter.getPointGlob(selected_cols_[3]));
  p_N_w_->setCol(3, adapter.getNormalGlob(selected_cols_[3]));
  p_bv_->setCol(3, adapter.getBearingVector(selected_cols_[3]));
  return nValid == K;
}

// Pose from two points and the normal of the first (Drost et al. 2010; reference :77-142): rotate both normals onto
// the x axis, then one rotation about x aligns the second point.  As in the reference the in-plane angle comes from
// acos() and therefore has no sign: the solver is exact only when that rotation is counter-clockwise.
template <typename Tp>
void nl_2p(const rpe::Point3<Tp>& pt1_c, const rpe::Point3<Tp>& nl1_c, const rpe::Point3<Tp>& pt2_c, const rpe::Point3<Tp>& pt1_w,
           const rpe::Point3<Tp>& nl1_w, const rpe::Point3<Tp>& pt2_w, rpe::SE3<Tp>* p_solution) {
  typedef rpe::Point3<Tp> V3;
  typedef rpe::SO3<Tp> Rot;
  auto to_x_axis = [](const V3& n) {  // rotation taking unit n to (1,0,0): angle acos(n.x) about n x e_x
    V3 axis(Tp(0), n[2], -n[1]);
    axis.normalize();
    return Rot::fromAngleAxis(std::acos(n[0]), axis);
  };
  const Rot R_g_f_w = to_x_axis(nl1_w), R_gp_f_c = to_x_axis(nl1_c);
  V3 a = R_g_f_w * (pt2_w - pt1_w); a[0] = Tp(0); a.normalize();
  V3 b = R_gp_f_c * (pt2_c - pt1_c); b[0] = Tp(0); b.normalize();
  const Rot R_gp_f_g = Rot::fromAngleAxis(std::acos(a.dot(b)), V3(Tp(1), Tp(0), Tp(0)));
  p_solution->so3() = R_gp_f_c.inverse() * R_gp_f_g * R_g_f_w;
  p_solution->translation() = pt1_c - p_solution->so3() * pt1_w;
}

namespace rpe {
// which = 0: nl_kneip_ransac (:215-284), 1: nl_shinji_ransac (:286-354), 2: nl_shinji_kneip_ransac (:356-445)
template <typename Tp>
void nl_sac(NormalAOPoseAdapter<Tp>& adapter, int which, const Tp thre_3d_, const Tp thre_2d_, const Tp nl_thre, int& Iter,
    Tp confidence, const RunOptions& opt) {
  Rand31& rnd = opt.stream();
  const int N = adapter.getNumberCorrespondences();
  const int K = 3;
  RandomElements<int> re(N);
  VoteSpec<Tp> spec;
  spec.kind = which == 0 ? RPE_VOTE_NN_23 : (which == 1 ? RPE_VOTE_NN_33 : RPE_VOTE_NN_33_23);
  spec.thre_3d = thre_3d_;
  if (which != 1) spec.cos_thr = std::cos(std::atan(thre_2d_ / adapter.getFocal()));
  spec.cos_nl = std::cos(nl_thre);
  spec.modalities = which == 2 ? 3 : 2;
  spec.model_points = K;
  ensure_all_arrays<Tp>(adapter);
  MatrixX<Tp> Xw(3, K + 1), Xc(3, K + 1), bv(3, K + 1), Nw(3, K + 1), Nc(3, K + 1);
  auto gen = [&](std::vector<SE3<Tp> >& out) {
    std::vector<int> sel;
    re.run(K + 1, &sel, rnd);
    const bool all_valid = assign_sample<Tp>(adapter, sel, &Xw, &Nw, &Xc, &Nc, &bv);
    if (which == 0) {
      SE3<Tp> sk;
      if (kneip<Tp>(Xw, bv, &sk)) out.push_back(sk);
      return;
    }
    if (all_valid) { const SE3<Tp> fit = shinji<Tp>(Xw, Xc, K); if (fit.so3().valid()) out.push_back(fit); }
    if (which == 2) { SE3<Tp> sk; if (kneip<Tp>(Xw, bv, &sk)) out.push_back(sk); }
    SE3<Tp> sn;  // NB the reference runs nl_2p on whatever Xc/Nc hold, also when the sample was not all valid (:315, :389)
    nl_2p<Tp>(Xc.col(0), Nc.col(0), Xc.col(1), Xw.col(0), Nw.col(0), Xw.col(1), &sn);
    out.push_back(sn);
  };
  auto commit = [&](int cols, unsigned device_cols) {
    // the lists requested again after the engine (same levels as the cvtInlier calls below)
    if (which != 1) { PnPPoseAdapter<Tp>* p = &adapter; p->forgetInlierIdx(); }
    if (which != 0) { AOPoseAdapter<Tp>* p = &adapter; p->forgetInlierIdx(); }
    adapter.forgetInlierIdx();
    adapter.setInlierFromDevice(cols, device_cols);
  };
  const Settings& cfg = Settings::get();
  if (opt.mode() == RPE_SCORE_FAST && cfg.device_hypotheses && N >= K + 1 && !cfg.capture && !cfg.replay)
    // FAST mode: later batches generated on the device
    ransac_engine_device_p3p<Tp>(adapter, spec, /*solver=*/2 + which, gen, commit, Iter, confidence, /*mask_cols=*/3, opt);
  else
    ransac_engine<Tp>(adapter, spec, gen, commit, Iter, confidence, /*mask_cols=*/3, opt);
  if (which != 1) { PnPPoseAdapter<Tp>* p = &adapter; p->cvtInlier(); }
  if (which != 0) { AOPoseAdapter<Tp>* p = &adapter; p->cvtInlier(); }
  adapter.cvtInlier();
}
}  // namespace rpe

template <typename Tp>
void nl_kneip_ransac(NormalAOPoseAdapter<Tp>& adapter, const Tp thre_2d_, const Tp nl_thre, int& Iter, Tp confidence = 0.99,
    const rpe::RunOptions& opt = rpe::RunOptions()) {
  rpe::nl_sac<Tp>(adapter, 0, Tp(0), thre_2d_, nl_thre, Iter, confidence, opt);
}
template <typename Tp>
void nl_shinji_ransac(NormalAOPoseAdapter<Tp>& adapter, const Tp thre_3d_, const Tp nl_thre, int& Iter, Tp confidence = 0.99,
    const rpe::RunOptions& opt = rpe::RunOptions()) {
  rpe::nl_sac<Tp>(adapter, 1, thre_3d_, Tp(0), nl_thre, Iter, confidence, opt);
}
template <typename Tp>
void nl_shinji_kneip_ransac(NormalAOPoseAdapter<Tp>& adapter, const Tp thre_3d_, const Tp thre_2d_, const Tp nl_thre, int& Iter,
                            Tp confidence = 0.99, const rpe::RunOptions& opt = rpe::RunOptions()) {
  rpe::nl_sac<Tp>(adapter, 2, thre_3d_, thre_2d_, nl_thre, Iter, confidence, opt);
}

// Joint least squares over the three inlier sets (reference :447-552): weighted 3D centroids, then three rounds of
// { M23 (bearing x direction-to-point), M33 (centred covariance), MNN (normal covariance) -> M33 + sigma (M23 + MNN)
//   -> SVD -> R ; camera centre blended from the 3D fit and find_opt_cc }.
// bug_compatible = true reproduces the reference exactly: M33/M23/MNN, TW/TL and the counts K/M are declared outside
// the round loop (:473-477 vs :481), so rounds 2 and 3 accumulate on top of the already normalised matrices and K keeps
// growing.  false resets them every round (the evident intent).
template <typename Tp>
void nl_shinji_kneip_ls(NormalAOPoseAdapter<Tp>& adapter, bool bug_compatible = true) {
  typedef rpe::Point3<Tp> V3;
  using rpe::Mat3d;
  if (adapter.getMaxVotes() == 0) return;
  rpe::DeviceSet& dev = adapter.device();
  rpe::ensure_all_arrays<Tp>(adapter);
  rpe::sync_masks_and_weights<Tp>(adapter);
  // weighted centroids of the 3D-3D inliers: one pass of K1' (mask33, weight33)
  double m[18];
  rpe::check(rpe_p2p_moments(dev.ctx(), RPE_USE_MASK | (adapter.weights33().empty() ? 0 : RPE_USE_WEIGHT), m), "rpe_p2p_moments");
  const int N = (int)m[17];
  const double TV = m[0];
  V3 Cw((Tp)m[1], (Tp)m[2], (Tp)m[3]), Cc((Tp)m[4], (Tp)m[5], (Tp)m[6]);
  if (N > 2) { Cw /= (Tp)TV; Cc /= (Tp)TV; }

  Mat3d M33, MNN, M23;
  double TL = 0, TW = 0;
  long M = 0, K = 0;
  V3 c_opt = adapter.getRcw().inverse() * (-adapter.gettw());  // camera centre in the world frame
  rpe::SO3<Tp> R_opt;
  for (int round = 0; round < 3; round++) {
    if (!bug_compatible) { M33 = Mat3d(); MNN = Mat3d(); M23 = Mat3d(); TL = TW = 0; M = K = 0; }
    double rec[44];
    rpe::nl_round_on_device<Tp>(adapter, c_opt, Cw, Cc, rec);
    for (int i = 0; i < 9; i++) { M23.a[i] += rec[i]; M33.a[i] += rec[11 + i]; MNN.a[i] += rec[21 + i]; }
    TW += rec[9]; K += (long)rec[10];
    double sigma_w_sqr = rec[20];
    TL += rec[30]; M += (long)rec[31];
    if (N > 2) { for (int i = 0; i < 9; i++) M33.a[i] /= TV; sigma_w_sqr /= TV; } else { M33 = Mat3d(); sigma_w_sqr = 1.; }
    if (M > 0) { for (int i = 0; i < 9; i++) MNN.a[i] /= TL; } else { MNN = Mat3d(); }
    if (K > 0) { for (int i = 0; i < 9; i++) M23.a[i] /= TW; } else { M23 = Mat3d(); }
    for (int i = 0; i < 9; i++) M33.a[i] += sigma_w_sqr * (M23.a[i] + MNN.a[i]);
    const Mat3d Rd = rpe::rotation_from_covariance(M33);
    rpe::Matrix3<Tp> Rt;
    for (int i = 0; i < 9; i++) Rt.a[i] = (Tp)Rd.a[i];
    const rpe::Quat<Tp> q = rpe::quat_from_R<Tp>(Rt.a);
    R_opt = rpe::SO3<Tp>::fromQuaternion(q.w, q.x, q.y, q.z);
    const V3 c = Cw - R_opt.inverse() * Cc;
    const V3 cp = rpe::cc_from_record<Tp>(rec);  // find_opt_cc at the adapter's (RANSAC) rotation, as the reference calls it
    if (N > 2) {
      if (cp[0] == cp[0]) c_opt = (Tp(K) / (K + N)) * cp + (Tp(N) / (K + N)) * c;
      else c_opt = c;
    } else {
      if (cp[0] == cp[0]) c_opt = cp;
      else break;
    }
  }
  adapter.setRcw(R_opt);
  adapter.sett(R_opt * (-c_opt));
}

#endif
