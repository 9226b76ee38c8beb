// pose/GaussNewton.hpp -- the iterative least-squares path of the north star, at adapter level.  NEW: the reference
// has no Jacobian / Gauss-Newton code; these functions refine the pose an adapter carries by
// Gauss-Newton on SE(3) (left update T <- exp(delta) T, tangent order (upsilon, omega) as sophus/se3.hpp:314-342) with the
// per-correspondence residuals and 6-DoF Jacobians reduced to the 6x6 / 6x1 normal equations on the GPU:
//   gn_refine_p2p      r = R Xw + t - Xc               same objective as shinji() (AbsoluteOrientation.hpp:47-99)   K1
//   gn_refine_p2plane  r = Nc . (R Xw + t - Xc)        point-to-plane, no reference counterpart                     K2
//   gn_refine_bearing  r = normalize(R Xw + t) x bv    the residual of lsq_pnp / getError (P3P.hpp:482-485)         K3
//   gn_refine_reproj   r = f (p_x/p_z - bv_x/bv_z, p_y/p_z - bv_y/bv_z)   pixel reprojection (the conversion of
//                      TestMain.cpp:35-36 with the adapter's focal length, principal point at the origin)           K3'
//   gn_refine_joint    scale_33 * p2p + scale_23 * bearing over both inlier sets of an AOPoseAdapter      K1+K3 fused
//   gn_refine_full     3D-3D (point-to-point or point-to-plane) + 2D-3D + normal-normal terms of a NormalAOPoseAdapter in
//                      ONE fused pass per iteration, optional robust (Huber / Cauchy) weights and the adapter's weights:
//                      the Gauss-Newton counterpart of nl_shinji_kneip_ls (AbsoluteOrientationNormal.hpp:447-552)
// All use the adapter's inlier masks (what RANSAC left there) unless use_inliers = false.  Return = iterations run.
#ifndef RPE_GAUSS_NEWTON_HEADER
#define RPE_GAUSS_NEWTON_HEADER

#include "AOOnlyPoseAdapter.hpp"
#include "AOPoseAdapter.hpp"
#include "NormalAOPoseAdapter.hpp"
#include "AbsoluteOrientationNormal.hpp"

namespace rpe {
template <typename Tp, class Adapter>
int gn_run(Adapter& adapter, int nterms, const int* kinds, const double* scales, bool use_inliers, int max_iter, double tol) {
  double pose[12];
  const Matrix3<Tp> R = adapter.getRcw().matrix();
  for (int i = 0; i < 9; i++) pose[i] = R.a[i];
  for (int i = 0; i < 3; i++) pose[9 + i] = adapter.gettw()[i];
  int iters = 0;
  double step = 0, cost = 0;
  check(rpe_gn_refine(adapter.device().ctx(), nterms, kinds, scales, use_inliers ? RPE_USE_MASK : 0, pose, max_iter, tol, &iters,
      &step, &cost),
        "rpe_gn_refine");
  Matrix3<Tp> Rt;
  for (int i = 0; i < 9; i++) Rt.a[i] = (Tp)pose[i];
  const Quat<Tp> q = quat_from_R<Tp>(Rt.a);
  adapter.setRcw(SO3<Tp>::fromQuaternion(q.w, q.x, q.y, q.z));
  adapter.sett(Point3<Tp>((Tp)pose[9], (Tp)pose[10], (Tp)pose[11]));
  return iters;
}
}  // namespace rpe

template <typename Tp, class Adapter>  // AOOnlyPoseAdapter, AOPoseAdapter or NormalAOPoseAdapter
int gn_refine_p2p(Adapter& adapter, int max_iter = 20, double tol = 1e-9, bool use_inliers = true) {
  const int N = adapter.getNumberCorrespondences();
  adapter.device().template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  adapter.device().template ensure<Tp>(RPE_XC, adapter.pointsCurrData(), N);
  if (use_inliers) adapter.pushMask33();
  const int kind = RPE_RES_P2P;
  return rpe::gn_run<Tp>(adapter, 1, &kind, nullptr, use_inliers, max_iter, tol);
}
template <typename Tp>
int gn_refine_p2plane(NormalAOPoseAdapter<Tp>& adapter, int max_iter = 20, double tol = 1e-9, bool use_inliers = true) {
  const int N = adapter.getNumberCorrespondences();
  adapter.device().template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  adapter.device().template ensure<Tp>(RPE_XC, adapter.pointsCurrData(), N);
  adapter.device().template ensure<Tp>(RPE_NC, adapter.normalCurrData(), N);
  if (use_inliers) adapter.pushMask33();
  const int kind = RPE_RES_P2PLANE;
  return rpe::gn_run<Tp>(adapter, 1, &kind, nullptr, use_inliers, max_iter, tol);
}
template <typename Tp>
int gn_refine_bearing(PnPPoseAdapter<Tp>& adapter, int max_iter = 20, double tol = 1e-9, bool use_inliers = true) {
  const int N = adapter.getNumberCorrespondences();
  adapter.device().template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  adapter.device().template ensure<Tp>(RPE_BV, adapter.bearingData(), N);
  if (use_inliers) adapter.pushMask23();
  const int kind = RPE_RES_BEARING;
  return rpe::gn_run<Tp>(adapter, 1, &kind, nullptr, use_inliers, max_iter, tol);
}
// 2D-3D refinement on the PIXEL reprojection residual: the cost is in pixels^2 of the adapter's focal length (getFocal()); the step
// itself does not depend on the focal length
template <typename Tp>
int gn_refine_reproj(PnPPoseAdapter<Tp>& adapter, int max_iter = 20, double tol = 1e-9, bool use_inliers = true) {
  const int N = adapter.getNumberCorrespondences();
  adapter.device().template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  adapter.device().template ensure<Tp>(RPE_BV, adapter.bearingData(), N);
  if (use_inliers) adapter.pushMask23();
  const int kind = RPE_RES_REPROJ;
  const double f = (double)adapter.getFocal(), scale = f * f;
  return rpe::gn_run<Tp>(adapter, 1, &kind, &scale, use_inliers, max_iter, tol);
}
template <typename Tp>
int gn_refine_joint(AOPoseAdapter<Tp>& adapter, double scale_33 = 1.0, double scale_23 = 1.0, int max_iter = 20, double tol = 1e-9,
                    bool use_inliers = true) {
  const int N = adapter.getNumberCorrespondences();
  adapter.device().template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  adapter.device().template ensure<Tp>(RPE_XC, adapter.pointsCurrData(), N);
  adapter.device().template ensure<Tp>(RPE_BV, adapter.bearingData(), N);
  if (use_inliers) { adapter.pushMask33(); adapter.pushMask23(); }
  const int kinds[2] = {RPE_RES_P2P, RPE_RES_BEARING};
  const double scales[2] = {scale_33, scale_23};
  return rpe::gn_run<Tp>(adapter, 2, kinds, scales, use_inliers, max_iter, tol);
}

namespace rpe {
struct JointOptions {
  double scale_33 = 1.0, scale_23 = 1.0, scale_nn = 1.0;
  bool point_to_plane = false;        // 3D-3D term: r = Nc . (p - Xc) instead of p - Xc
  int robust = RPE_ROBUST_NONE;       // applied to every term with the k below
  double k_33 = 0.1, k_23 = 0.01, k_nn = 0.1;
  bool use_inliers = true, use_weights = false;
  int max_iter = 20;
  double tol = 1e-9;
};
}  // namespace rpe

template <typename Tp>
int gn_refine_full(NormalAOPoseAdapter<Tp>& adapter, const rpe::JointOptions& o = rpe::JointOptions()) {
  rpe::ensure_all_arrays<Tp>(adapter);
  if (o.use_inliers || o.use_weights) rpe::sync_masks_and_weights<Tp>(adapter);
  rpe_term terms[3] = {{o.point_to_plane ? RPE_RES_P2PLANE : RPE_RES_P2P, o.scale_33, o.robust, o.k_33},
                       {RPE_RES_BEARING, o.scale_23, o.robust, o.k_23},
                       {RPE_RES_NORMAL, o.scale_nn, o.robust, o.k_nn}};
  double pose[12];
  const rpe::Matrix3<Tp> R = adapter.getRcw().matrix();
  for (int i = 0; i < 9; i++) pose[i] = R.a[i];
  for (int i = 0; i < 3; i++) pose[9 + i] = adapter.gettw()[i];
  int iters = 0;
  double step = 0, cost = 0;
  const int flags = (o.use_inliers ? RPE_USE_MASK : 0) | ((o.use_weights && !adapter.weights33().empty()) ? RPE_USE_WEIGHT : 0);
  rpe::check(rpe_gn_refine_joint(adapter.device().ctx(), 3, terms, flags, pose, o.max_iter, o.tol, &iters, &step, &cost),
      "rpe_gn_refine_joint");
  rpe::Matrix3<Tp> Rt;
  for (int i = 0; i < 9; i++) Rt.a[i] = (Tp)pose[i];
  const rpe::Quat<Tp> q = rpe::quat_from_R<Tp>(Rt.a);
  adapter.setRcw(rpe::SO3<Tp>::fromQuaternion(q.w, q.x, q.y, q.z));
  adapter.sett(rpe::Point3<Tp>((Tp)pose[9], (Tp)pose[10], (Tp)pose[11]));
  return iters;
}

#endif
