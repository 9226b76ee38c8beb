// pose/DepthFrontEnd.hpp -- ADDITIVE (no reference counterpart): the step before the adapters.  The reference's callers
// hand the adapters 3 x N host matrices (TestMain.cpp:145-183 fills them from Simulator.hpp); with a depth camera those
// matrices come from a depth frame.  This class keeps that production on the GPU (Part 3 of include/rgbd_pose_hip.h):
//
//   rpe::DepthFrontEnd fe;
//   fe.setDepth(depth_mm, cam);  fe.setModelFromFrame(T_cw_prev);      // previous frame becomes the model
//   fe.setDepth(next_depth_mm, cam);
//   rpe::IcpResult r = fe.icp(T_cw_guess);                              // dense projective ICP, pose refined in place
//   rpe::DepthFrontEnd::Pairs pairs = fe.pairs(T_cw_guess);             // ... or: the adapters' matrices, born in HBM
//   NormalAOPoseAdapter<float> adapter(pairs.bv, pairs.xc, pairs.nc, pairs.xw, pairs.nw);
//   fe.attach(adapter, pairs);                                          // the solvers find the arrays resident: no upload
//   nl_shinji_kneip_ransac<float>(adapter, ...);
//
// Camera: the simulator's pinhole (Simulator.hpp:150-162).  Poses cross this interface as Sophus::SE3<double>.
#ifndef RPE_DEPTH_FRONT_END_HEADER
#define RPE_DEPTH_FRONT_END_HEADER

#include "PoseAdapterBase.hpp"

namespace rpe {

struct PinholeCamera {
  double fx = 585., fy = 585., cx = 320., cy = 240.;   // Simulator.hpp:160-162, SimpleMain.cpp:42
  int width = 640, height = 480;
};
// metres = raw value * scale; vertices outside (dmin, dmax) are invalid; normals are dropped across jumps > max_jump
struct DepthRange {
  double scale, dmin, dmax, max_jump;
  static DepthRange millimetres() { return DepthRange{0.001, 0.3, 8.0, 0.1}; }   // the usual uint16 sensor frame
  static DepthRange metres() { return DepthRange{1.0, 0.3, 8.0, 0.1}; }
};
struct IcpOptions {
  int kind = RPE_RES_P2PLANE;
  int max_iter = 10;
  double tol = 1e-6, dist_thr = 0.1, cos_thr = 0.9;
  // fused + host update = ONE resident launch for the whole loop (the fastest form: 11 us per round at 640 x 480); device_resident =
  // true
  // keeps solve and update on the GPU instead (also one launch, the grid iterates by itself: 12 us per round; no busy host thread)
  bool use_normals = true, device_resident = false, fused = true;
};
struct IcpResult { int iterations = 0; double last_step = 0, cost = 0; long long pairs = 0; };

class DepthFrontEnd {
 public:
  typedef SE3<double> Pose;
  struct Pairs {            // the five adapter matrices, index = frame pixel; unpaired pixels are NaN columns of xc / bv / nc
    MatrixX<float> bv, xc, nc, xw, nw;
    long long count = 0;
  };

  explicit DepthFrontEnd(int device = Settings::get().device) : _ctx(nullptr), _device(device), _pixels(0) {
    check(rpe_create(&_ctx, device, nullptr), "rpe_create");
  }
  ~DepthFrontEnd() { if (_ctx) rpe_destroy(_ctx); }
  DepthFrontEnd(const DepthFrontEnd&) = delete;
  DepthFrontEnd& operator=(const DepthFrontEnd&) = delete;

  void setDepth(const unsigned short* depth, const PinholeCamera& cam,
      const DepthRange& r = DepthRange::millimetres()) { set(depth, RPE_DEPTH_U16, cam, r); }
  void setDepth(const float* depth, const PinholeCamera& cam,
      const DepthRange& r = DepthRange::metres()) { set(depth, RPE_DEPTH_F32, cam, r); }
  // the current frame, seen from T_cw, becomes the model the next frames are registered against
  void setModelFromFrame(const Pose& T_cw) {
    double p[12]; pose12(T_cw, p);
    check(rpe_model_from_frame(_ctx, p), "rpe_model_from_frame");
  }
  void setModel(const MatrixX<float>& vertex_w, const MatrixX<float>& normal_w, const PinholeCamera& cam, const Pose& T_cw) {
    double p[12]; pose12(T_cw, p);
    const rpe_camera k = cam_of(cam);
    if (vertex_w.cols() != cam.width * cam.height || normal_w.cols() != vertex_w.cols()) throw DeviceError(RPE_ERR_ARG,
        "setModel: maps must be 3 x width*height");
    check(rpe_model_upload(_ctx, vertex_w.data(), normal_w.data(), &k, p), "rpe_model_upload");
  }
  MatrixX<float> map(int which) const {
    MatrixX<float> m(3, _pixels);
    check(rpe_frame_download(_ctx, which, m.data()), "rpe_frame_download");
    return m;
  }
  long long associate(const Pose& guess, double dist_thr = 0.1, double cos_thr = 0.9, bool use_normals = true) {
    double p[12]; pose12(guess, p);
    int64_t m = 0;
    check(rpe_associate(_ctx, p, dist_thr, cos_thr, use_normals ? 1 : 0, &m), "rpe_associate");
    return (long long)m;
  }
  // dense ICP from `pose` (in/out)
  IcpResult icp(Pose& pose, const IcpOptions& o = IcpOptions()) {
    double p[12]; pose12(pose, p);
    rpe_icp_options opt;
    opt.kind = o.kind; opt.max_iter = o.max_iter; opt.tol = o.tol; opt.dist_thr = o.dist_thr; opt.cos_thr = o.cos_thr;
    opt.use_normals = o.use_normals; opt.device_resident = o.device_resident; opt.fused = o.fused;
    IcpResult r;
    int64_t m = 0;
    check(rpe_icp(_ctx, &opt, p, &r.iterations, &r.last_step, &r.cost, &m), "rpe_icp");
    r.pairs = (long long)m;
    pose = pose_of(p);
    return r;
  }
  // associate under `guess` and bring the five arrays to the host (the adapters' getters and the minimal solvers read them)
  Pairs pairs(const Pose& guess, double dist_thr = 0.1, double cos_thr = 0.9, bool use_normals = true) {
    Pairs P;
    P.count = associate(guess, dist_thr, cos_thr, use_normals);
    MatrixX<float>* dst[RPE_NUM_ARRAYS] = {&P.xw, &P.xc, &P.bv, &P.nw, &P.nc};
    for (int s = 0; s < RPE_NUM_ARRAYS; s++) { dst[s]->resize(3, _pixels);
        check(rpe_download(_ctx, s, dst[s]->data()), "rpe_download"); }
    return P;
  }
  // an adapter constructed over `P` runs its solvers on this front end's context: the arrays are already in HBM
  template <class Adapter> void attach(Adapter& adapter, const Pairs& P) {
    const void* host[RPE_NUM_ARRAYS] = {P.xw.data(), P.xc.data(), P.bv.data(), P.nw.data(), P.nc.data()};
    adapter.device().adopt(_ctx, _device, _pixels, RPE_F32, host);
  }
  rpe_context* context() { return _ctx; }
  int pixels() const { return _pixels; }

  static void pose12(const Pose& T, double p[12]) {
    const Matrix3<double> R = T.so3().matrix();
    for (int i = 0; i < 9; i++) p[i] = R.a[i];
    for (int i = 0; i < 3; i++) p[9 + i] = T.translation()[i];
  }
  static Pose pose_of(const double p[12]) {
    Matrix3<double> R; for (int i = 0; i < 9; i++) R.a[i] = p[i];
    const Quat<double> q = quat_from_R<double>(R.a);
    return Pose(SO3<double>::fromQuaternion(q.w, q.x, q.y, q.z), Point3<double>(p[9], p[10], p[11]));
  }

 private:
  static rpe_camera cam_of(const PinholeCamera& c) { rpe_camera k; k.fx = c.fx; k.fy = c.fy; k.cx = c.cx; k.cy = c.cy;
      k.width = c.width; k.height = c.height; return k; }
  void set(const void* depth, int type, const PinholeCamera& cam, const DepthRange& r) {
    const rpe_camera k = cam_of(cam);
    check(rpe_frame_set_depth(_ctx, depth, type, &k, r.scale, r.dmin, r.dmax, r.max_jump), "rpe_frame_set_depth");
    _pixels = cam.width * cam.height;
  }
  rpe_context* _ctx;
  int _device, _pixels;
};

}  // namespace rpe

#endif
