// Part 1 of include/rgbd_pose_hip.h -- the reference's own FFI (Library.cpp:15-82) -- and rpe_run, both written
// against the drop-in C++ headers (pose/*.hpp) exactly the way the reference's Library.cpp / TestMain.cpp use theirs.
#include "../../include/rgbd_pose_hip.h"
#include "../include/pose/AbsoluteOrientation.hpp"
#include "../include/pose/AbsoluteOrientationNormal.hpp"
#include "../include/pose/GaussNewton.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

namespace rpe { int set_error(int code, const char* msg); }

namespace {

bool quiet() { const char* q = getenv("RPE_QUIET"); return q && q[0] == '1'; }

// the reference aborts on failure too (SOPHUS_ENSURE -> abort, sophus/common.hpp:114-133): ao() has no error channel
[[noreturn]] void die(const char* where, const char* what) {
  std::fprintf(stderr, "librgbdpose_hip: %s failed: %s\n", where, what);
  std::abort();
}

// caller-owned xyz-interleaved buffer seen as the 3 x n column-major matrix the adapters take
template <class Tp> struct HostMat {
  const Tp* p; int n;
  const Tp* data() const { return p; }
  int rows() const { return 3; }
  int cols() const { return n; }
};
template <class Tp> struct HostWeights {  // n x wcols column-major
  const Tp* p; int n, c;
  int rows() const { return n; }
  int cols() const { return c; }
  Tp operator()(int r, int k) const { return p[(size_t)k * n + r]; }
};

template <class Tp, class Adapter> void write_pose(Adapter& ad, double* R9, double* t3) {
  const rpe::Matrix3<Tp> R = ad.getRcw().matrix();
  for (int i = 0; i < 9; i++) R9[i] = (double)R.a[i];
  for (int i = 0; i < 3; i++) t3[i] = (double)ad.gettw()[i];
}
template <class Tp, class Adapter> void read_pose(Adapter& ad, const double* R9, const double* t3) {
  rpe::Matrix3<Tp> R;
  for (int i = 0; i < 9; i++) R.a[i] = (Tp)R9[i];
  const rpe::Quat<Tp> q = rpe::quat_from_R<Tp>(R.a);
  ad.setRcw(rpe::SO3<Tp>::fromQuaternion(q.w, q.x, q.y, q.z));
  ad.sett(rpe::Point3<Tp>((Tp)t3[0], (Tp)t3[1], (Tp)t3[2]));
}

enum { M_SHINJI_RANSAC = 0, M_SHINJI_RANSAC2 = 1, M_SHINJI_PROSAC = 2, M_KNEIP_RANSAC = 3, M_KNEIP_PROSAC = 4, M_SK_RANSAC = 5,
       M_SK_PROSAC = 6, M_NL_KNEIP_RANSAC = 7, M_NL_SHINJI_RANSAC = 8, M_NL_SK_RANSAC = 9, M_NONE = 10 };
enum { LS_NONE = 0, LS_SHINJI_INLIERS = 1, LS_NL_BUGCOMPAT = 2, LS_NL_FIXED = 3, LS_SHINJI_ALL = 4, LS_GN_P2P = 5, LS_GN_JOINT = 6,
       LS_GN_P2PLANE = 7, LS_GN_BEARING = 8, LS_GN_REPROJ = 9 };

template <class Tp>
int run_t(int method, const rpe_problem* p, double thre_3d, double thre_2d, double thre_nl, int* iter_io, double confidence,
    uint64_t seed, int score_mode,
          int ls, const short* mask_in, double* R9, double* t3, int* max_votes, short* mask_out) {
  const int n = p->n;
  // everything this run may vary lives in the call: its own random stream (the `seed` argument) and its scoring mode.  Nothing
  // process-wide is written, so concurrent calls neither interleave one stream nor see each other's mode.
  rpe::Rand31 rng(seed);
  rpe::RunOptions opt;
  opt.rng = &rng; opt.score_mode = score_mode;
  HostMat<Tp> bv{(const Tp*)p->bv, n}, xc{(const Tp*)p->xc, n}, nc{(const Tp*)p->nc, n}, xw{(const Tp*)p->xw, n}, nw{(const Tp*)p->nw, n};
  HostWeights<Tp> w{(const Tp*)p->weights, n, p->wcols};
  int Iter = iter_io ? *iter_io : 0;
  // mask_out: rows 23 | 33 | NN of n shorts each; a modality the adapter does not have reads as zero.  Written straight from the device
  // copy where that is the current one (copyInlierMask*).
  auto zero_row = [&](int row) { if (mask_out) std::memset(mask_out + (size_t)row * n, 0, (size_t)n * sizeof(short)); };
  const bool has_n = p->nc && p->nw, has_bv = p->bv != nullptr, has_xc = p->xc != nullptr;
  const bool aoonly = method == M_SHINJI_RANSAC2 || method == M_SHINJI_PROSAC || (method == M_NONE && !has_bv);
  const bool pnp = method == M_KNEIP_RANSAC || method == M_KNEIP_PROSAC || (method == M_NONE && has_bv && !has_xc);
  const bool ao = method == M_SHINJI_RANSAC || method == M_SK_RANSAC || method == M_SK_PROSAC || (method == M_NONE && has_bv && has_xc
      && !has_n);
  if (aoonly) {
    AOOnlyPoseAdapter<Tp> ad(xc, xw);
    ad.setFocal((Tp)p->fx, (Tp)p->fy);
    if (p->weights) ad.setWeights(w);
    if (method == M_SHINJI_RANSAC2) shinji_ransac2<Tp>(ad, (Tp)thre_3d, Iter, (Tp)confidence, opt);
    if (method == M_SHINJI_PROSAC) shinji_prosac<Tp>(ad, (Tp)thre_3d, Iter, (Tp)confidence, opt);
    if (method == M_NONE) {
      read_pose<Tp>(ad, R9, t3);
      if (mask_in) { rpe::MatrixXs m(n, 2); for (int i = 0; i < n; i++) m(i, 1) = mask_in[n + i]; ad.setInlier(m); ad.cvtInlier(); }
    }
    if (ls == LS_SHINJI_INLIERS) shinji_ls1<Tp>(ad);
    if (ls == LS_SHINJI_ALL) shinji_ls2<Tp>(ad);
    if (ls == LS_GN_P2P) Iter = gn_refine_p2p<Tp>(ad);
    write_pose<Tp>(ad, R9, t3);
    if (max_votes) *max_votes = ad.getMaxVotes();
    // host copies are fetched only when asked for
    if (mask_out) { zero_row(0); ad.copyInlierMask33(mask_out + (size_t)n); zero_row(2); }
  } else if (pnp) {
    PnPPoseAdapter<Tp> ad(bv, xw);
    ad.setFocal((Tp)p->fx, (Tp)p->fy);
    if (p->weights) ad.setWeights(w);
    if (method == M_KNEIP_RANSAC) kneip_ransac<Tp>(ad, (Tp)thre_2d, Iter, (Tp)confidence, opt);
    if (method == M_KNEIP_PROSAC) kneip_prosac<Tp>(ad, (Tp)thre_2d, Iter, (Tp)confidence, opt);
    if (method == M_NONE) {
      read_pose<Tp>(ad, R9, t3);
      if (mask_in) { rpe::MatrixXs m(n, 1); for (int i = 0; i < n; i++) m(i, 0) = mask_in[i]; ad.setInlier(m); ad.cvtInlier(); }
    }
    if (ls == LS_GN_BEARING) Iter = gn_refine_bearing<Tp>(ad);
    if (ls == LS_GN_REPROJ) Iter = gn_refine_reproj<Tp>(ad);
    write_pose<Tp>(ad, R9, t3);
    if (max_votes) *max_votes = ad.getMaxVotes();
    if (mask_out) { ad.copyInlierMask23(mask_out); zero_row(1); zero_row(2); }
  } else if (ao) {
    AOPoseAdapter<Tp> ad(bv, xc, xw);
    ad.setFocal((Tp)p->fx, (Tp)p->fy);
    if (p->weights) ad.setWeights(w);
    if (method == M_SHINJI_RANSAC) shinji_ransac<Tp>(ad, (Tp)thre_3d, Iter, (Tp)confidence, opt);
    if (method == M_SK_RANSAC) shinji_kneip_ransac<Tp>(ad, (Tp)thre_3d, (Tp)thre_2d, Iter, (Tp)confidence, opt);
    if (method == M_SK_PROSAC) shinji_kneip_prosac<Tp>(ad, (Tp)thre_3d, (Tp)thre_2d, Iter, (Tp)confidence, opt);
    if (method == M_NONE) {
      read_pose<Tp>(ad, R9, t3);
      if (mask_in) {
        rpe::MatrixXs m(n, 2); for (int i = 0; i < n; i++) { m(i, 0) = mask_in[i]; m(i, 1) = mask_in[n + i]; }
        ad.setInlier(m); PnPPoseAdapter<Tp>* b = &ad; b->cvtInlier(); ad.cvtInlier();
      }
    }
    if (ls == LS_SHINJI_INLIERS) shinji_ls<Tp>(ad);
    if (ls == LS_GN_P2P) Iter = gn_refine_p2p<Tp>(ad);
    if (ls == LS_GN_JOINT) Iter = gn_refine_joint<Tp>(ad);
    if (ls == LS_GN_BEARING) Iter = gn_refine_bearing<Tp>(ad);
    if (ls == LS_GN_REPROJ) Iter = gn_refine_reproj<Tp>(ad);
    write_pose<Tp>(ad, R9, t3);
    if (max_votes) *max_votes = ad.getMaxVotes();
    if (mask_out) { ad.copyInlierMask23(mask_out); ad.copyInlierMask33(mask_out + (size_t)n); zero_row(2); }
  } else {
    NormalAOPoseAdapter<Tp> ad(bv, xc, nc, xw, nw);
    ad.setFocal((Tp)p->fx, (Tp)p->fy);
    if (method == M_NL_KNEIP_RANSAC) nl_kneip_ransac<Tp>(ad, (Tp)thre_2d, (Tp)thre_nl, Iter, (Tp)confidence, opt);
    if (method == M_NL_SHINJI_RANSAC) nl_shinji_ransac<Tp>(ad, (Tp)thre_3d, (Tp)thre_nl, Iter, (Tp)confidence, opt);
    if (method == M_NL_SK_RANSAC) nl_shinji_kneip_ransac<Tp>(ad, (Tp)thre_3d, (Tp)thre_2d, (Tp)thre_nl, Iter, (Tp)confidence, opt);
    if (method == M_NONE) {
      read_pose<Tp>(ad, R9, t3);
      ad.setMaxVotes(max_votes ? *max_votes : 1);
      if (mask_in) {
        rpe::MatrixXs m(n, 3); for (int i = 0; i < n; i++) { m(i, 0) = mask_in[i]; m(i, 1) = mask_in[n + i];
            m(i, 2) = mask_in[2 * n + i]; }
        ad.setInlier(m);
      }
    }
    if (p->weights) ad.setWeights(w);  // TestMain.cpp:215-221: 'opt' runs unweighted, 'dw' after setWeights(all_weights)
    if (ls == LS_NL_BUGCOMPAT) nl_shinji_kneip_ls<Tp>(ad, true);
    if (ls == LS_NL_FIXED) nl_shinji_kneip_ls<Tp>(ad, false);
    if (ls == LS_SHINJI_INLIERS) shinji_ls<Tp>(ad);
    if (ls == LS_GN_P2P) Iter = gn_refine_p2p<Tp>(ad);
    if (ls == LS_GN_P2PLANE) Iter = gn_refine_p2plane<Tp>(ad);
    if (ls == LS_GN_JOINT) Iter = gn_refine_joint<Tp>(ad);
    write_pose<Tp>(ad, R9, t3);
    if (max_votes) *max_votes = ad.getMaxVotes();
    if (mask_out) { ad.copyInlierMask23(mask_out); ad.copyInlierMask33(mask_out + (size_t)n); ad.copyInlierMaskNN(mask_out + 2 * (size_t)n); }
  }
  if (iter_io) *iter_io = Iter;
  return RPE_OK;
}

template <class Tp> rpe::MatrixX<Tp> mat_from(const double* p, int cols) {
  rpe::MatrixX<Tp> m(3, cols);
  for (int i = 0; i < 3 * cols; i++) m.data()[i] = (Tp)p[i];
  return m;
}
template <class Tp> void pose_out(const rpe::SE3<Tp>& s, double* R9, double* t3) {
  const rpe::Matrix3<Tp> R = s.so3().matrix();
  for (int i = 0; i < 9; i++) R9[i] = (double)R.a[i];
  for (int i = 0; i < 3; i++) t3[i] = (double)s.translation()[i];
}

void write_rowmajor(AOOnlyPoseAdapter<float>& adapter, float* R_cw_, float* t_) {
  // row-major storage == the reference's Rp = R^T column-major dump (Library.cpp:35-39)
  const rpe::Matrix3<float> R = adapter.getRcw().matrix();
  for (int i = 0; i < 9; i++) R_cw_[i] = R.a[i];
  for (int i = 0; i < 3; i++) t_[i] = adapter.gettw()[i];
}

}  // namespace

extern "C" {

void ao(float* x_w_, float* x_c_, int n_, float* R_cw_, float* t_) {
  if (!quiet()) std::cout << "ao()" << std::endl;
  try {
    HostMat<float> Xw{x_w_, n_}, Xc{x_c_, n_};
    AOOnlyPoseAdapter<float> adapter(Xc, Xw);
    const float f = 555.f;
    adapter.setFocal(f, f);
    shinji_ls2<float>(adapter);
    write_rowmajor(adapter, R_cw_, t_);
  } catch (const std::exception& e) { die("ao", e.what()); }
}

void ao_ransac(float* x_w_, float* x_c_, int n_, float* R_cw_, float* t_) {
  if (!quiet()) std::cout << "ao()" << std::endl;  // sic: the reference prints "ao()" here too (Library.cpp:49)
  try {
    HostMat<float> Xw{x_w_, n_}, Xc{x_c_, n_};
    AOOnlyPoseAdapter<float> adapter(Xc, Xw);
    const float f = 555.f;
    adapter.setFocal(f, f);
    int updated_iter = 1000;
    const float thre_3d = 0.1f, confidence = 0.99999f;
    // the reference draws from the process-global rand() here (not thread-safe, never seeded); this call owns its stream
    rpe::Rand31 rng(1);
    if (const char* s = getenv("RPE_SEED")) rng.reseed(strtoull(s, nullptr, 10));
    rpe::RunOptions opt;
    opt.rng = &rng;
    shinji_ransac2<float>(adapter, thre_3d, updated_iter, confidence, opt);
    if (!quiet()) {
      std::cout << "updated_iter = " << updated_iter << std::endl;
      std::cout << "inliers = " << adapter.getMaxVotes() << std::endl;
    }
    shinji_ls1<float>(adapter);
    write_rowmajor(adapter, R_cw_, t_);
  } catch (const std::exception& e) { die("ao_ransac", e.what()); }
}

void py2c(float* array, int N) {
  for (int i = 0; i < N; i++) std::cout << array[i] << std::endl;
}

// ---- host-side pieces of the solvers, callable without a GPU (CPU tests of the host logic; other-language hosts)
void rpe_host_random_elements(int n, int m, uint64_t seed, int draws, int* out) {
  rpe::Rand31 rnd(seed);
  RandomElements<int> re(n);
  std::vector<int> v;
  for (int d = 0; d < draws; d++) { re.run(m, &v, rnd); for (int k = 0; k < m; k++) out[d * m + k] = v[k]; }
}
void rpe_host_prosac_samples(int dtype, int m, int n, uint64_t seed, int draws, int* out) {
  rpe::Rand31 rnd(seed);
  std::vector<int> v;
  if (dtype == RPE_F64) { ProsacSampler<double> ps(m, n); for (int d = 0; d < draws; d++) { ps.sample(&v, rnd);
      for (int k = 0; k < m; k++) out[d * m + k] = v[k]; } }
  else { ProsacSampler<float> ps(m, n); for (int d = 0; d < draws; d++) { ps.sample(&v, rnd);
      for (int k = 0; k < m; k++) out[d * m + k] = v[k]; } }
}
int rpe_host_update_num_iters(int dtype, double p, double ep, int model_points, int max_iters) {
  return dtype == RPE_F64 ? RANSACUpdateNumIters<double>(p, ep, model_points, max_iters)
                          : RANSACUpdateNumIters<float>((float)p, (float)ep, model_points, max_iters);
}
void rpe_host_sort_indexes(const double* w, int n, int* out) {
  std::vector<double> v(w, w + n);
  std::vector<int> idx = sortIndexes<double>(v);
  for (int i = 0; i < n; i++) out[i] = idx[i];
}
// xw4, bv4: 3 x 4 column-major doubles (values are rounded to dtype).  sols12: up to 4 x (R9 | t3).  returns the count
int rpe_host_kneip_main(int dtype, const double* xw4, const double* bv4, double* sols12) {
  int cnt = 0;
  if (dtype == RPE_F64) { std::vector<rpe::SE3<double> > v; kneip_main<double>(mat_from<double>(xw4, 4), mat_from<double>(bv4, 4), &v);
    for (auto& s : v) { pose_out(s, sols12 + 12 * cnt, sols12 + 12 * cnt + 9); cnt++; } }
  else { std::vector<rpe::SE3<float> > v; kneip_main<float>(mat_from<float>(xw4, 4), mat_from<float>(bv4, 4), &v);
    for (auto& s : v) { pose_out(s, sols12 + 12 * cnt, sols12 + 12 * cnt + 9); cnt++; } }
  return cnt;
}
int rpe_host_kneip(int dtype, const double* xw4, const double* bv4, double* R9, double* t3) {
  if (dtype == RPE_F64) { rpe::SE3<double> s; if (!kneip<double>(mat_from<double>(xw4, 4), mat_from<double>(bv4, 4), &s)) return 0;
      pose_out(s, R9, t3); return 1; }
  rpe::SE3<float> s; if (!kneip<float>(mat_from<float>(xw4, 4), mat_from<float>(bv4, 4), &s)) return 0; pose_out(s, R9, t3); return 1;
}
// v18: pt1_c nl1_c pt2_c pt1_w nl1_w pt2_w
void rpe_host_nl_2p(int dtype, const double* v18, double* R9, double* t3) {
  if (dtype == RPE_F64) { rpe::Point3<double> a[6]; for (int i = 0; i < 6; i++) a[i] = rpe::Point3<double>(v18 + 3 * i);
    rpe::SE3<double> s; nl_2p<double>(a[0], a[1], a[2], a[3], a[4], a[5], &s); pose_out(s, R9, t3); }
  else { rpe::Point3<float> a[6]; for (int i = 0; i < 6; i++) a[i] = rpe::Point3<float>((float)v18[3 * i], (float)v18[3 * i + 1],
      (float)v18[3 * i + 2]);
    rpe::SE3<float> s; nl_2p<float>(a[0], a[1], a[2], a[3], a[4], a[5], &s); pose_out(s, R9, t3); }
}
// shinji() on K host columns (3 x K column-major doubles)
void rpe_host_shinji(int dtype, const double* xw, const double* xc, int K, double* R9, double* t3) {
  if (dtype == RPE_F64) pose_out(shinji<double>(mat_from<double>(xw, K), mat_from<double>(xc, K), K), R9, t3);
  else pose_out(shinji<float>(mat_from<float>(xw, K), mat_from<float>(xc, K), K), R9, t3);
}
void rpe_host_se3_exp(const double* a6, double* R9, double* t3) { rpe::se3_exp(a6, R9, t3); }
void rpe_host_svd3(const double* A9, double* U9, double* s3, double* V9) {
  rpe::Mat3d A; for (int i = 0; i < 9; i++) A.a[i] = A9[i];
  const rpe::Svd3 d = rpe::svd3(A);
  for (int i = 0; i < 9; i++) { U9[i] = d.U.a[i]; V9[i] = d.V.a[i]; }
  for (int i = 0; i < 3; i++) s3[i] = d.s[i];
}
// (t_e, r_e) of calc_err and calc_percentage_err (AbsoluteOrientation.hpp:11-43), double
void rpe_host_calc_err(const double* Rgt9, const double* tgt3, const double* Rse9, const double* tse3, double* err2, double* pct2) {
  rpe::Matrix3<double> A, B;
  for (int i = 0; i < 9; i++) { A.a[i] = Rgt9[i]; B.a[i] = Rse9[i]; }
  const rpe::SO3<double> Ra(A), Rb(B);
  const rpe::Point3<double> ta(tgt3), tb(tse3);
  const rpe::Point3<double> e = calc_err<double>(rpe::SE3<double>(Ra, ta), rpe::SE3<double>(Rb, tb));
  err2[0] = e[0]; err2[1] = e[1];
  rpe::MatrixX<double> none(3, 0);
  AOOnlyPoseAdapter<double> ad(none, none);
  ad.setRcw(Rb); ad.sett(tb);
  const rpe::Point3<double> p = calc_percentage_err<double>(Ra, ta, &ad);
  pct2[0] = p[0]; pct2[1] = p[1];
}

int rpe_run(int method, const rpe_problem* p, double thre_3d, double thre_2d, double thre_nl, int* iter_io, double confidence,
    uint64_t seed,
            int ls, int score_mode, const short* mask_in, double* R9, double* t3, int* max_votes, short* mask_out) {
  if (!p || !R9 || !t3 || p->n <= 0 || !p->xw) return rpe::set_error(RPE_ERR_ARG, "rpe_run: bad argument");
  int rc;
  try {
    rc = p->dtype == RPE_F64 ? run_t<double>(method, p, thre_3d, thre_2d, thre_nl, iter_io, confidence, seed, score_mode, ls, mask_in, R9,
        t3, max_votes, mask_out)
                             : run_t<float>(method, p, thre_3d, thre_2d, thre_nl, iter_io, confidence, seed, score_mode, ls, mask_in,
                                 R9, t3, max_votes, mask_out);
  } catch (const rpe::DeviceError& e) {
    rc = rpe::set_error(e.code, e.what());
  } catch (const std::exception& e) {
    rc = rpe::set_error(RPE_ERR_STATE, e.what());
  }
  return rc;
}


// ---- explicit hypothesis streams (rpe::Settings::capture / replay, rpe/device.hpp)
int rpe_host_hypotheses(int method, const rpe_problem* p, int iters, uint64_t seed, double* q7_out, int cap, int* first_out) {
  if (!p || p->n <= 0 || !p->xw || iters < 0 || !q7_out || !first_out || method < 0 || method > 9) return rpe::set_error(RPE_ERR_ARG,
      "rpe_host_hypotheses: bad argument");
  rpe::Settings::HypothesisList list;
  rpe::Settings::get().capture = &list;
  int it = iters, mv = 0, rc = RPE_OK;
  double R9[9], t3[3];
  try {
    rc = p->dtype == RPE_F64 ? run_t<double>(method, p, 1.0, 1.0, 1.0, &it, 0.99, seed, -1, 0, nullptr, R9, t3, &mv, nullptr)
                             : run_t<float>(method, p, 1.0, 1.0, 1.0, &it, 0.99, seed, -1, 0, nullptr, R9, t3, &mv, nullptr);
  } catch (const std::exception& e) { rc = rpe::set_error(RPE_ERR_STATE, e.what()); }
  rpe::Settings::get().capture = nullptr;
  if (rc != RPE_OK) return rc;
  if (list.first.empty()) list.first.assign((size_t)iters + 1, 0);
  const int H = (int)(list.q7.size() / 7);
  if (H > cap || (int)list.first.size() != iters + 1) return rpe::set_error(RPE_ERR_ARG,
      "rpe_host_hypotheses: output capacity too small");
  if (!list.q7.empty()) std::memcpy(q7_out, list.q7.data(), sizeof(double) * list.q7.size());
  std::memcpy(first_out, list.first.data(), sizeof(int) * list.first.size());
  return H;
}

int rpe_run_replay(int method, const rpe_problem* p, const double* poses7, const int* first, int list_iters, double thre_3d, double thre_2d,
                   double thre_nl, int* iter_io, double confidence, int ls, int score_mode, double* R9, double* t3, int* max_votes,
                   short* mask_out) {
  if (!poses7 || !first || list_iters < 0) return rpe::set_error(RPE_ERR_ARG, "rpe_run_replay: bad argument");
  if (first[0] != 0) return rpe::set_error(RPE_ERR_ARG, "rpe_run_replay: first[0] must be 0");
  // first[i] .. first[i + 1] index the hypothesis array: non-negative and non-decreasing, or the copy below runs wild
  for (int i = 0; i < list_iters; i++)
    if (first[i + 1] < first[i]) return rpe::set_error(RPE_ERR_ARG, "rpe_run_replay: first[] must be non-decreasing");
  rpe::Settings::HypothesisList list;
  list.first.assign(first, first + list_iters + 1);
  list.q7.assign(poses7, poses7 + 7 * (size_t)first[list_iters]);
  rpe::Settings::get().replay = &list;
  const int rc = rpe_run(method, p, thre_3d, thre_2d, thre_nl, iter_io, confidence, 1, ls, score_mode, nullptr, R9, t3, max_votes,
      mask_out);
  rpe::Settings::get().replay = nullptr;
  return rc;
}

}  // extern "C"
