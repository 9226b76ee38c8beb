// pose/GaussNewton.hpp -- the iterative least-squares path of the north star, at adapter level.  NEW: the reference
// has no Jacobian / Gauss-Newton code; these functions refine the pose an adapter carries by
// Gauss-Newton on SE(3) (left update T <- exp(delta) T, tangent order (upsilon, omega) as sophus/se3.hpp:314-342) with the
// per-correspondence residuals and 6-DoF Jacobians reduced to the 6x6 / 6x1 normal equations on the GPU:
//   gn_refine_p2p      r = R Xw + t - Xc               same objective as shinji() (AbsoluteOrientation.hpp:47-99)   K1
//   gn_refine_p2plane  r = Nc . (R Xw + t - Xc)        point-to-plane, no reference counterpart                     K2
//   gn_refine_bearing  r = normalize(R Xw + t) x bv    the residual of lsq_pnp / getError (P3P.hpp:482-485)         K3
//   gn_refine_joint    scale_33 * p2p + scale_23 * bearing over both inlier sets of an AOPoseAdapter
// All use the adapter's inlier masks (what RANSAC left there) unless use_inliers = false.  Return = iterations run.
#ifndef RPE_GAUSS_NEWTON_HEADER
#define RPE_GAUSS_NEWTON_HEADER

#include "AOOnlyPoseAdapter.hpp"
#include "AOPoseAdapter.hpp"
#include "NormalAOPoseAdapter.hpp"

namespace rpe {
template <typename Tp, class Adapter>
int gn_run(Adapter& adapter, int nterms, const int* kinds, const double* scales, bool use_inliers, int max_iter, double tol) {
  double pose[12];
  const Matrix3<Tp> R = adapter.getRcw().matrix();
  for (int i = 0; i < 9; i++) pose[i] = R.a[i];
  for (int i = 0; i < 3; i++) pose[9 + i] = adapter.gettw()[i];
  int iters = 0;
  double step = 0, cost = 0;
  check(rpe_gn_refine(adapter.device().ctx(), nterms, kinds, scales, use_inliers ? RPE_USE_MASK : 0, pose, max_iter, tol, &iters, &step, &cost),
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
  if (use_inliers) adapter.device().upload_mask(RPE_MOD_33, adapter.inlierMask33());
  const int kind = RPE_RES_P2P;
  return rpe::gn_run<Tp>(adapter, 1, &kind, nullptr, use_inliers, max_iter, tol);
}
template <typename Tp>
int gn_refine_p2plane(NormalAOPoseAdapter<Tp>& adapter, int max_iter = 20, double tol = 1e-9, bool use_inliers = true) {
  const int N = adapter.getNumberCorrespondences();
  adapter.device().template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  adapter.device().template ensure<Tp>(RPE_XC, adapter.pointsCurrData(), N);
  adapter.device().template ensure<Tp>(RPE_NC, adapter.normalCurrData(), N);
  if (use_inliers) adapter.device().upload_mask(RPE_MOD_33, adapter.inlierMask33());
  const int kind = RPE_RES_P2PLANE;
  return rpe::gn_run<Tp>(adapter, 1, &kind, nullptr, use_inliers, max_iter, tol);
}
template <typename Tp>
int gn_refine_bearing(PnPPoseAdapter<Tp>& adapter, int max_iter = 20, double tol = 1e-9, bool use_inliers = true) {
  const int N = adapter.getNumberCorrespondences();
  adapter.device().template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  adapter.device().template ensure<Tp>(RPE_BV, adapter.bearingData(), N);
  if (use_inliers) adapter.device().upload_mask(RPE_MOD_23, adapter.inlierMask23());
  const int kind = RPE_RES_BEARING;
  return rpe::gn_run<Tp>(adapter, 1, &kind, nullptr, use_inliers, max_iter, tol);
}
template <typename Tp>
int gn_refine_joint(AOPoseAdapter<Tp>& adapter, double scale_33 = 1.0, double scale_23 = 1.0, int max_iter = 20, double tol = 1e-9,
                    bool use_inliers = true) {
  const int N = adapter.getNumberCorrespondences();
  adapter.device().template ensure<Tp>(RPE_XW, adapter.pointsGlobData(), N);
  adapter.device().template ensure<Tp>(RPE_XC, adapter.pointsCurrData(), N);
  adapter.device().template ensure<Tp>(RPE_BV, adapter.bearingData(), N);
  if (use_inliers) { adapter.device().upload_mask(RPE_MOD_33, adapter.inlierMask33()); adapter.device().upload_mask(RPE_MOD_23, adapter.inlierMask23()); }
  const int kinds[2] = {RPE_RES_P2P, RPE_RES_BEARING};
  const double scales[2] = {scale_33, scale_23};
  return rpe::gn_run<Tp>(adapter, 2, kinds, scales, use_inliers, max_iter, tol);
}

#endif
