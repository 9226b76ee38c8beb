// examples/icp_main.cpp -- ADDITIVE demo (no reference counterpart): frame-to-frame registration of two dense 640 x 480 depth
// frames of the simulator's camera (f = 585, Simulator.hpp:160-162) with the front end of pose/DepthFrontEnd.hpp:
//   1. dense projective ICP on the GPU (pairs never leave HBM), from the previous frame's pose;
//   2. the same pairs handed to a NormalAOPoseAdapter WITHOUT an upload: nl_shinji_ransac on them (the reference's solver,
//      AbsoluteOrientationNormal.hpp:270-341) followed by the fused Gauss-Newton refinement.
// The scene is an analytic room with spheres, ray-cast here so that the true motion is known.  Exit code 0 = both paths
// recover the motion.   Build: python -m rgbd_pose_estimation_amd.build --examples
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "AbsoluteOrientationNormal.hpp"
#include "DepthFrontEnd.hpp"
#include "GaussNewton.hpp"

typedef rpe::SE3<double> Pose;
typedef rpe::Point3<double> P3;

static std::vector<unsigned short> render(const Pose& T_cw, const rpe::PinholeCamera& cam) {
  const double lo[3] = {-2.5, -1.6, -1.0}, hi[3] = {2.7, 1.5, 5.0};
  const double sph[5][4] = {{-0.9, 0.3, 2.6, 0.55}, {0.8, -0.4, 3.4, 0.7}, {0.1, 0.9, 2.0, 0.35}, {1.6, 0.8, 4.2, 0.5}, {-1.7, -0.7, 3.9, 0.6}};
  const Pose T_wc = T_cw.inverse();
  const P3 C = T_wc.translation();
  std::vector<unsigned short> depth((size_t)cam.width * cam.height, 0);
  for (int v = 0; v < cam.height; v++)
    for (int u = 0; u < cam.width; u++) {
      const P3 d = T_wc.so3() * P3((u - cam.cx) / cam.fx, (v - cam.cy) / cam.fy, 1.0);   // z_cam = lambda
      double best = 1e30;
      for (int ax = 0; ax < 3; ax++)
        for (int side = 0; side < 2; side++) {
          const double l = ((side ? hi[ax] : lo[ax]) - C[ax]) / d[ax];
          if (!(l > 1e-9) || l >= best) continue;
          bool in = true;
          for (int o = 0; o < 3; o++) if (o != ax) { const double p = C[o] + l * d[o]; in = in && p >= lo[o] - 1e-9 && p <= hi[o] + 1e-9; }
          if (in) best = l;
        }
      for (const auto& s : sph) {
        const P3 oc = C - P3(s[0], s[1], s[2]);
        const double a = d.dot(d), b = 2 * d.dot(oc), c = oc.dot(oc) - s[3] * s[3], disc = b * b - 4 * a * c;
        if (disc > 0) { const double l = (-b - std::sqrt(disc)) / (2 * a); if (l > 1e-9 && l < best) best = l; }
      }
      if (best < 60.0) depth[(size_t)v * cam.width + u] = (unsigned short)std::lround(best * 1000.0);
    }
  return depth;
}

static void errors(const Pose& est, const Pose& truth, double* rot_rad, double* trans_m) {
  const rpe::Matrix3<double> D = est.so3().matrix() * truth.so3().matrix().transpose();
  const double c = std::max(-1.0, std::min(1.0, (D(0, 0) + D(1, 1) + D(2, 2) - 1) / 2));
  *rot_rad = std::acos(c);
  *trans_m = (est.translation() - truth.translation()).norm();
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const double scale = argc > 1 ? std::atof(argv[1]) : 1.0;   // size of the inter-frame motion (1 = 2.4 deg, 4.4 cm)
  rpe::PinholeCamera cam;
  const Pose TA(rpe::SO3<double>::exp(P3(0.05, -0.1, 0.02)), P3(0.1, -0.05, 0.2));
  const double m[6] = {0.03 * scale, -0.02 * scale, 0.025 * scale, 0.02 * scale, -0.015 * scale, 0.01 * scale};
  const Pose dT = Pose::exp(m);
  const Pose TB(dT.so3() * TA.so3(), dT.so3() * TA.translation() + dT.translation());
  const std::vector<unsigned short> dA = render(TA, cam), dB = render(TB, cam);
  double r0, t0;
  errors(TA, TB, &r0, &t0);
  std::printf("%%motion between the frames: %.5f rad, %.5f m\n", r0, t0);

  try {
    rpe::DepthFrontEnd fe;
    fe.setDepth(dA.data(), cam);
    fe.setModelFromFrame(TA);
    fe.setDepth(dB.data(), cam);

    // ---- 1. dense ICP, everything on the GPU
    rpe::IcpOptions o;
    o.max_iter = 20; o.tol = 1e-7; o.dist_thr = 0.15; o.cos_thr = 0.8;
    Pose T = TA;
    fe.icp(T, o);  // warm-up (first launches)
    T = TA;
    const double t_icp0 = now_ms();
    const rpe::IcpResult res = fe.icp(T, o);
    const double t_icp = now_ms() - t_icp0;
    double r1, t1;
    errors(T, TB, &r1, &t1);
    std::printf("icp      iterations %d  pairs %lld  rot_err %.3e rad  trans_err %.3e m  (%.3f ms, %.1f us / round)\n", res.iterations, res.pairs, r1,
                t1, t_icp, 1e3 * t_icp / res.iterations);

    // ---- 2. the reference's RANSAC on the same dense pairs, arrays already in HBM
    rpe::DepthFrontEnd::Pairs P = fe.pairs(TA, 0.3, 0.5);   // loose gates around the stale pose: these pairs contain outliers
    NormalAOPoseAdapter<float> adapter(P.bv, P.xc, P.nc, P.xw, P.nw);
    adapter.setFocal((float)cam.fx, (float)cam.fy);
    fe.attach(adapter, P);
    int iter = 200;
    const double t_r0 = now_ms();
    nl_shinji_ransac<float>(adapter, 0.05f, 0.15f, iter, 0.99f);
    rpe::JointOptions jo;
    jo.point_to_plane = true; jo.scale_23 = 0.0; jo.scale_nn = 0.0; jo.max_iter = 10; jo.tol = 1e-7;
    gn_refine_full<float>(adapter, jo);
    const double t_r = now_ms() - t_r0;
    const rpe::Matrix3<float> Rf = adapter.getRcw().matrix();
    double p12[12];
    for (int i = 0; i < 9; i++) p12[i] = Rf.a[i];
    for (int i = 0; i < 3; i++) p12[9 + i] = adapter.gettw()[i];
    double r2, t2;
    errors(rpe::DepthFrontEnd::pose_of(p12), TB, &r2, &t2);
    std::printf("ransac   pairs %lld  votes %d  iterations %d  rot_err %.3e rad  trans_err %.3e m  (%.3f ms)\n", P.count, adapter.getMaxVotes(), iter, r2, t2, t_r);
    const bool ok = r1 < 1e-3 && t1 < 3e-3 && r2 < 0.25 * r0 && t2 < 0.25 * t0;
    std::printf("%%summary icp_main -> %s\n", ok ? "ok" : "LARGE");
    return ok ? 0 : 1;
  } catch (const rpe::DeviceError& e) {
    std::fprintf(stderr, "device error %d: %s\n", e.code, e.what());
    return 2;
  }
}
