// pose/Simulator.hpp -- drop-in for /root/reference/pose/Simulator.hpp: synthetic RGB-D correspondence scenes.
// Same free functions, same argument order and meaning; host code (input generation sits outside the hot path).
// The reference draws from an unseeded global std::default_random_engine and Eigen's Random() (:13-14,19,33,140);
// here one seedable engine backs both (rpe::sim_seed), so scenes are reproducible.  Distributions, parameters and
// the outlier placement (incl. "normal outliers overwrite the FIRST columns", :114-119) follow the reference.
#ifndef RPE_SIMULATOR_HEADER
#define RPE_SIMULATOR_HEADER

#include <cmath>
#include <random>
#include <vector>
#include "../rpe/types.hpp"
#include "Utility.hpp"

namespace rpe {
inline std::mt19937_64& sim_engine() { static std::mt19937_64 e(5489u); return e; }
inline void sim_seed(uint64_t s) { sim_engine().seed(s); }
// Eigen Random(): U[-1,1]
template <typename T> inline T sim_uniform() { return std::uniform_real_distribution<T>(T(-1), T(1))(sim_engine()); }
template <typename T> inline T sim_normal() { return (T)std::normal_distribution<double>(0., 1.)(sim_engine()); }
template <typename T> inline Point3<T> sim_vec3(bool gaussian) {
  return gaussian ? Point3<T>(sim_normal<T>(), sim_normal<T>(), sim_normal<T>()) : Point3<T>(sim_uniform<T>(), sim_uniform<T>(),
      sim_uniform<T>());
}
// the reference's outlier index draw goes through RandomElements (rand()); use the library stream for it
inline std::vector<int> sim_pick(int number, int count) {
  RandomElements<int> re(number);
  std::vector<int> idx;
  re.run(count, &idx);
  return idx;
}
}  // namespace rpe

template <typename T>
rpe::Point3<T> generate_random_translation_uniform(T size) { return size * rpe::sim_vec3<T>(false); }  // reference :16-21

// R = Rz(rz) Ry(ry) Rx(rx), (rx, ry, rz) = a (v0, v1/2, v2) clamped to [-pi,pi] x [-pi/2,pi/2] x [-pi,pi]  (reference :23-83)
template <typename T>
rpe::SO3<T> generate_random_rotation(T max_angle_radian_, bool use_guassian_ = true) {
  const rpe::Point3<T> v = rpe::sim_vec3<T>(use_guassian_);
  auto clamp = [](T a, T lim) { return a > lim ? lim : (a < -lim ? -lim : a); };
  const T rx = clamp(max_angle_radian_ * v[0], T(M_PI)), ry = clamp(max_angle_radian_ * v[1] * T(.5), T(M_PI / 2.)),
          rz = clamp(max_angle_radian_ * v[2], T(M_PI));
  const T cx = std::cos(rx), sx = std::sin(rx), cy = std::cos(ry), sy = std::sin(ry), cz = std::cos(rz), sz = std::sin(rz);
  rpe::Matrix3<T> R;
  R(0, 0) = cz * cy; R(0, 1) = cz * sy * sx - sz * cx; R(0, 2) = cz * sy * cx + sz * sx;
  R(1, 0) = sz * cy; R(1, 1) = sz * sy * sx + cz * cx; R(1, 2) = sz * sy * cx - cz * sx;
  R(2, 0) = -sy;     R(2, 1) = cy * sx;                R(2, 2) = cy * cx;
  const rpe::Quat<T> q = rpe::quat_from_R<T>(R.a);
  return rpe::SO3<T>::fromQuaternion(q.w, q.x, q.y, q.z);
}

// normal-normal correspondences (reference :85-130)
template <typename T>
void simulate_nl_nl_correspondences(const rpe::SO3<T>& R_cw_, int number_, T noise_nl_, T outlier_ratio_nl_, bool use_guassian_,
                                    rpe::MatrixX<T>* pM_, rpe::MatrixX<T>* pN_, rpe::MatrixX<T>* pN_gt = NULL,
                                    rpe::MatrixX<T>* p_all_weights_ = NULL) {
  typedef rpe::Point3<T> V3;
  pM_->resize(3, number_); pN_->resize(3, number_);
  rpe::MatrixX<T> N_gt(3, number_);
  std::vector<T> w(number_);
  const V3 down(0, 0, -1);
  auto facing_camera = [](const V3& n) { return !(std::acos(n[2]) < T(M_PI / 2)); };
  for (int i = 0; i < number_; i++) {
    V3 g, noisy;
    do {
      g = generate_random_rotation<T>(T(M_PI / 2.), false) * down; g.normalize();
      noisy = generate_random_rotation<T>(noise_nl_, use_guassian_) * g; noisy.normalize();
    } while (!facing_camera(noisy));
    V3 m = R_cw_.inverse() * g; m.normalize();
    N_gt.setCol(i, g); pM_->setCol(i, m); pN_->setCol(i, noisy);
    w[i] = noisy.dot(g);
  }
  const int out = int(outlier_ratio_nl_ * number_ + T(.5));
  rpe::sim_pick(number_, out);  // drawn and ignored by the reference (:112-113)
  for (int i = 0; i < out; i++) {
    V3 g;
    do { g = generate_random_rotation<T>(T(M_PI / 2), false) * down; g.normalize(); } while (!facing_camera(g));
    pN_->setCol(i, g);
  }
  if (pN_gt) *pN_gt = N_gt;
  if (p_all_weights_) for (int i = 0; i < number_; i++) (*p_all_weights_)(i, 2) = w[i];
}

template <typename T>
rpe::Point3<T> generate_a_random_point(T min_depth_, T max_depth_, T tan_fov_x, T tan_fov_y) {  // reference :135-145
  const rpe::Point3<T> u = rpe::sim_vec3<T>(false);
  return rpe::Point3<T>(u[0] * tan_fov_x * max_depth_, u[1] * tan_fov_y * max_depth_,
      (u[2] + T(1.)) / T(2.) * (max_depth_ - min_depth_) + min_depth_);
}

template <typename T>
rpe::MatrixX<T> project_point_cloud(const rpe::MatrixX<T>& pt_c, T f_) {  // reference :148-156
  rpe::MatrixX<T> px(2, pt_c.cols());
  for (int i = 0; i < pt_c.cols(); i++) { px(0, i) = f_ * pt_c(0, i) / pt_c(2, i); px(1, i) = f_ * pt_c(1, i) / pt_c(2, i); }
  return px;
}

// rejection sampling inside the 640x480 frustum (reference :158-173)
template <typename T>
rpe::MatrixX<T> simulate_rand_point_cloud_in_frustum(int number_, T f_, T min_depth_, T max_depth_) {
  const T tx = T(320. / f_), ty = T(240. / f_);
  rpe::MatrixX<T> cloud(3, number_);
  for (int i = 0; i < number_; i++) {
    rpe::Point3<T> P;
    do { P = generate_a_random_point<T>(min_depth_, max_depth_, tx, ty);
        } while (!(std::fabs(P[0] / P[2]) < tx && std::fabs(P[1] / P[2]) < ty));
    cloud.setCol(i, P);
  }
  return cloud;
}

// 2D-3D (reference :175-233): pQ_ world points (clean), pU_ unit bearings (pixel noise + outliers), pP_gt camera points
template <typename T>
void simulate_2d_3d_correspondences(const rpe::SO3<T>& R_cw_, const rpe::Point3<T>& t_w_, int number_, T noise_, T outlier_ratio_,
                                    T min_depth_, T max_depth_, T f_, bool use_guassian_, rpe::MatrixX<T>* pQ_, rpe::MatrixX<T>* pU_,
                                    rpe::MatrixX<T>* pP_gt = NULL, rpe::MatrixX<T>* p_all_weights_ = NULL) {
  const rpe::MatrixX<T> P_gt = simulate_rand_point_cloud_in_frustum<T>(number_, f_, min_depth_, max_depth_);
  rpe::MatrixX<T> kp = project_point_cloud<T>(P_gt, f_);
  pQ_->resize(3, number_);
  std::vector<T> w(number_);
  for (int i = 0; i < number_; i++) {
    pQ_->setCol(i, R_cw_.inverse() * (P_gt.col(i) - t_w_));
    const T r0 = use_guassian_ ? rpe::sim_normal<T>() : rpe::sim_uniform<T>(), r1 = use_guassian_ ? rpe::sim_normal<T>() : rpe::sim_uniform<T>();
    w[i] = T(1.) / std::sqrt(r0 * r0 + r1 * r1);
    kp(0, i) += noise_ * r0; kp(1, i) += noise_ * r1;
  }
  const int out = int(outlier_ratio_ * number_ + .5);
  const rpe::MatrixX<T> out_px = project_point_cloud<T>(simulate_rand_point_cloud_in_frustum<T>(out, f_, min_depth_, max_depth_), f_);
  const std::vector<int> idx = rpe::sim_pick(number_, out);
  for (int i = 0; i < out; i++) { kp(0, idx[i]) = out_px(0, i); kp(1, idx[i]) = out_px(1, i); }
  pU_->resize(3, number_);
  for (int c = 0; c < number_; c++) { rpe::Point3<T> b(kp(0, c), kp(1, c), f_); b.normalize(); pU_->setCol(c, b); }
  if (pP_gt) *pP_gt = P_gt;
  if (p_all_weights_) for (int i = 0; i < number_; i++) (*p_all_weights_)(i, 0) = w[i];
}

// 2D-3D + noisy world points (reference :235-265); used as AOPoseAdapter(U, P, Q)
template <typename T>
void simulate_2d_3d_3d_correspondences(const rpe::SO3<T>& R_cw_, const rpe::Point3<T>& t_w_, int number_, T noise_2d_, T noise_3d_,
                                       T outlier_ratio_, T min_depth_, T max_depth_, T f_, bool use_guassian_, rpe::MatrixX<T>* pQ_,
                                       rpe::MatrixX<T>* pU_, rpe::MatrixX<T>* pP_gt = NULL, rpe::MatrixX<T>* p_all_weights_ = NULL) {
  simulate_2d_3d_correspondences<T>(R_cw_, t_w_, number_, noise_2d_, outlier_ratio_, min_depth_, max_depth_, f_, use_guassian_, pQ_,
      pU_, pP_gt,
                                    p_all_weights_);
  for (int i = 0; i < number_; i++) {
    const rpe::Point3<T> rv = rpe::sim_vec3<T>(use_guassian_);
    pQ_->setCol(i, pQ_->col(i) + noise_3d_ * rv);
    if (p_all_weights_) (*p_all_weights_)(i, 1) = T(1.) / rv.norm();
  }
}

// 3D-3D (reference :268-314): noisy world points with outliers that "remain in CRS"; camera points clean
template <typename T>
void simulate_3d_3d_correspondences(const rpe::SO3<T>& R_cw_, const rpe::Point3<T>& t_w_, int number_, T noise_, T outlier_ratio_,
                                    T min_depth_, T max_depth_, T f_, bool use_guassian_, rpe::MatrixX<T>* pQ_,
                                    rpe::MatrixX<T>* pP_gt = NULL, rpe::MatrixX<T>* p_all_weights_ = NULL) {
  const rpe::MatrixX<T> P_gt = simulate_rand_point_cloud_in_frustum<T>(number_, f_, min_depth_, max_depth_);
  pQ_->resize(3, number_);
  for (int i = 0; i < number_; i++) {
    const rpe::Point3<T> rv = rpe::sim_vec3<T>(use_guassian_);
    pQ_->setCol(i, R_cw_.inverse() * (P_gt.col(i) - t_w_) + noise_ * rv);
    if (p_all_weights_) (*p_all_weights_)(i, 1) = T(1.) / rv.norm();
  }
  const int out = int(outlier_ratio_ * number_ + .5);
  const rpe::MatrixX<T> junk = simulate_rand_point_cloud_in_frustum<T>(out, f_, min_depth_, max_depth_);
  const std::vector<int> idx = rpe::sim_pick(number_, out);
  for (int i = 0; i < out; i++) pQ_->setCol(idx[i], junk.col(i));
  if (pP_gt) *pP_gt = P_gt;
}

// all three modalities (reference :316-367); used as NormalAOPoseAdapter(U, P, N, Q, M)
template <typename T>
void simulate_2d_3d_nl_correspondences(const rpe::SO3<T>& R_cw_, const rpe::Point3<T>& t_w_, int number_, T n2D_, T or_2D_, T n3D_,
    T or_3D_,
                                       T nNl_, T or_Nl_, T min_depth_, T max_depth_, T f_, bool use_guassian_, rpe::MatrixX<T>* pQ_,
                                       rpe::MatrixX<T>* pM_, rpe::MatrixX<T>* pP_, rpe::MatrixX<T>* pN_, rpe::MatrixX<T>* pU_,
                                       rpe::MatrixX<T>* p_all_weights_ = NULL) {
  rpe::MatrixX<T> all_weights(number_, 3), P_gt, nl_c_gt;
  simulate_2d_3d_correspondences<T>(R_cw_, t_w_, number_, n2D_, or_2D_, min_depth_, max_depth_, f_, use_guassian_, pQ_, pU_, &P_gt,
      &all_weights);
  simulate_nl_nl_correspondences<T>(R_cw_, number_, nNl_, or_Nl_, true, pM_, pN_, &nl_c_gt, &all_weights);
  pP_->resize(3, number_);
  for (int i = 0; i < number_; i++) {
    const rpe::Point3<T> rv = rpe::sim_vec3<T>(use_guassian_);
    all_weights(i, 1) = T(1.) / rv.norm();
    pP_->setCol(i, P_gt.col(i) + n3D_ * rv);
  }
  const int out = int(or_3D_ * number_ + .5);
  const std::vector<int> idx = rpe::sim_pick(number_, out);
  const rpe::MatrixX<T> junk = simulate_rand_point_cloud_in_frustum<T>(out, f_, min_depth_, max_depth_);
  for (int i = 0; i < out; i++) pP_->setCol(idx[i], junk.col(i));
  if (p_all_weights_) *p_all_weights_ = all_weights;
}

// Kinect noise model of Nguyen, Izadi, Lovell (3DIMPVT 2012) (reference :369-387)
template <typename T>
T lateral_noise_kinect(T theta_, T z_, T f_) { return (T(.8) + T(.035) * theta_ / (T(M_PI / 2.) - theta_)) * z_ / f_; }
template <typename T>
T axial_noise_kinect(T theta_, T z_) {
  const T base = T(.0012) + T(.0019) * (z_ - T(0.4)) * (z_ - T(0.4));
  if (std::fabs(theta_) <= T(M_PI / 3.)) return base;
  return base + T(.0001) * theta_ * theta_ / std::sqrt(z_) / (T(M_PI / 2) - theta_) / (T(M_PI / 2) - theta_);
}

template <typename T>
void simulate_kinect_2d_3d_nl_correspondences(const rpe::SO3<T>& R_cw_, const rpe::Point3<T>& t_w_, int number_, T noise_2d_,
    T outlier_ratio_2d_,
                                              T outlier_ratio_3d_, T noise_nl_, T outlier_ratio_nl_, T min_depth_, T max_depth_, T f_,
                                              rpe::MatrixX<T>* p_pt_w_, rpe::MatrixX<T>* p_nl_w_, rpe::MatrixX<T>* p_pt_c_,
                                              // :389-436
                                              rpe::MatrixX<T>* p_nl_c_, rpe::MatrixX<T>* p_bv_, rpe::MatrixX<T>* p_weights_ = NULL) {
  rpe::MatrixX<T> all_weights(number_, 3), pt_c_gt, nl_c_gt;
  simulate_2d_3d_correspondences<T>(R_cw_, t_w_, number_, noise_2d_, outlier_ratio_2d_, min_depth_, max_depth_, f_, true, p_pt_w_,
      p_bv_, &pt_c_gt,
                                    &all_weights);
  simulate_nl_nl_correspondences<T>(R_cw_, number_, noise_nl_, outlier_ratio_nl_, true, p_nl_w_, p_nl_c_, &nl_c_gt, &all_weights);
  const T sigma_min = axial_noise_kinect<T>(T(.0), min_depth_);
  p_pt_c_->resize(3, number_);
  for (int i = 0; i < number_; i++) {
    const T theta = std::acos(nl_c_gt.col(i).dot(rpe::Point3<T>(0, 0, -1)));
    const T z = pt_c_gt(2, i);
    const T sl = lateral_noise_kinect<T>(theta, z, f_), sa = axial_noise_kinect<T>(theta, z);
    p_pt_c_->setCol(i, pt_c_gt.col(i) + rpe::Point3<T>(sl * rpe::sim_normal<T>(), sl * rpe::sim_normal<T>(),
        sa * rpe::sim_normal<T>()));
    all_weights(i, 1) = T(sigma_min / sa);
  }
  const int out = int(outlier_ratio_3d_ * number_ + .5);
  const std::vector<int> idx = rpe::sim_pick(number_, out);
  const rpe::MatrixX<T> junk = simulate_rand_point_cloud_in_frustum<T>(out, f_, min_depth_, max_depth_);
  for (int i = 0; i < out; i++) p_pt_c_->setCol(idx[i], junk.col(i));
  if (p_weights_) *p_weights_ = all_weights;
}

#endif
