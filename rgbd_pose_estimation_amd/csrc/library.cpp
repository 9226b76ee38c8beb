// Part 1 of include/rgbd_pose_hip.h -- the reference's own FFI (Library.cpp:15-82) -- and rpe_run, both written
// against the drop-in C++ headers (pose/*.hpp) exactly the way the reference's Library.cpp / TestMain.cpp use theirs.
#include "../../include/rgbd_pose_hip.h"
#include "../include/pose/AbsoluteOrientation.hpp"
#include "../include/pose/AbsoluteOrientationNormal.hpp"
#include "../include/pose/GaussNewton.hpp"
#include <cstdio>
#include <cstdlib>
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
       LS_GN_P2PLANE = 7, LS_GN_BEARING = 8 };

template <class Tp>
int run_t(int method, const rpe_problem* p, double thre_3d, double thre_2d, double thre_nl, int* iter_io, double confidence, uint64_t seed,
          int ls, const short* mask_in, double* R9, double* t3, int* max_votes, short* mask_out) {
  const int n = p->n;
  HostMat<Tp> bv{(const Tp*)p->bv, n}, xc{(const Tp*)p->xc, n}, nc{(const Tp*)p->nc, n}, xw{(const Tp*)p->xw, n}, nw{(const Tp*)p->nw, n};
  HostWeights<Tp> w{(const Tp*)p->weights, n, p->wcols};
  int Iter = iter_io ? *iter_io : 0;
  rpe::seed(seed);
  auto masks_out = [&](const std::vector<short>* m23, const std::vector<short>* m33, const std::vector<short>* mnn) {
    if (!mask_out) return;
    for (int i = 0; i < n; i++) {
      mask_out[i] = m23 ? (*m23)[i] : 0; mask_out[n + i] = m33 ? (*m33)[i] : 0; mask_out[2 * n + i] = mnn ? (*mnn)[i] : 0;
    }
  };
  const bool has_n = p->nc && p->nw, has_bv = p->bv != nullptr, has_xc = p->xc != nullptr;
  const bool aoonly = method == M_SHINJI_RANSAC2 || method == M_SHINJI_PROSAC || (method == M_NONE && !has_bv);
  const bool pnp = method == M_KNEIP_RANSAC || method == M_KNEIP_PROSAC || (method == M_NONE && has_bv && !has_xc);
  const bool ao = method == M_SHINJI_RANSAC || method == M_SK_RANSAC || method == M_SK_PROSAC || (method == M_NONE && has_bv && has_xc && !has_n);
  if (aoonly) {
    AOOnlyPoseAdapter<Tp> ad(xc, xw);
    ad.setFocal((Tp)p->fx, (Tp)p->fy);
    if (p->weights) ad.setWeights(w);
    if (method == M_SHINJI_RANSAC2) shinji_ransac2<Tp>(ad, (Tp)thre_3d, Iter, (Tp)confidence);
    if (method == M_SHINJI_PROSAC) shinji_prosac<Tp>(ad, (Tp)thre_3d, Iter, (Tp)confidence);
    if (method == M_NONE) {
      read_pose<Tp>(ad, R9, t3);
      if (mask_in) { rpe::MatrixXs m(n, 2); for (int i = 0; i < n; i++) m(i, 1) = mask_in[n + i]; ad.setInlier(m); ad.cvtInlier(); }
    }
    if (ls == LS_SHINJI_INLIERS) shinji_ls1<Tp>(ad);
    if (ls == LS_SHINJI_ALL) shinji_ls2<Tp>(ad);
    if (ls == LS_GN_P2P) Iter = gn_refine_p2p<Tp>(ad);
    write_pose<Tp>(ad, R9, t3);
    if (max_votes) *max_votes = ad.getMaxVotes();
    masks_out(nullptr, &ad.inlierMask33(), nullptr);
  } else if (pnp) {
    PnPPoseAdapter<Tp> ad(bv, xw);
    ad.setFocal((Tp)p->fx, (Tp)p->fy);
    if (p->weights) ad.setWeights(w);
    if (method == M_KNEIP_RANSAC) kneip_ransac<Tp>(ad, (Tp)thre_2d, Iter, (Tp)confidence);
    if (method == M_KNEIP_PROSAC) kneip_prosac<Tp>(ad, (Tp)thre_2d, Iter, (Tp)confidence);
    if (method == M_NONE) {
      read_pose<Tp>(ad, R9, t3);
      if (mask_in) { rpe::MatrixXs m(n, 1); for (int i = 0; i < n; i++) m(i, 0) = mask_in[i]; ad.setInlier(m); ad.cvtInlier(); }
    }
    if (ls == LS_GN_BEARING) Iter = gn_refine_bearing<Tp>(ad);
    write_pose<Tp>(ad, R9, t3);
    if (max_votes) *max_votes = ad.getMaxVotes();
    masks_out(&ad.inlierMask23(), nullptr, nullptr);
  } else if (ao) {
    AOPoseAdapter<Tp> ad(bv, xc, xw);
    ad.setFocal((Tp)p->fx, (Tp)p->fy);
    if (p->weights) ad.setWeights(w);
    if (method == M_SHINJI_RANSAC) shinji_ransac<Tp>(ad, (Tp)thre_3d, Iter, (Tp)confidence);
    if (method == M_SK_RANSAC) shinji_kneip_ransac<Tp>(ad, (Tp)thre_3d, (Tp)thre_2d, Iter, (Tp)confidence);
    if (method == M_SK_PROSAC) shinji_kneip_prosac<Tp>(ad, (Tp)thre_3d, (Tp)thre_2d, Iter, (Tp)confidence);
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
    write_pose<Tp>(ad, R9, t3);
    if (max_votes) *max_votes = ad.getMaxVotes();
    masks_out(&ad.inlierMask23(), &ad.inlierMask33(), nullptr);
  } else {
    NormalAOPoseAdapter<Tp> ad(bv, xc, nc, xw, nw);
    ad.setFocal((Tp)p->fx, (Tp)p->fy);
    if (method == M_NL_KNEIP_RANSAC) nl_kneip_ransac<Tp>(ad, (Tp)thre_2d, (Tp)thre_nl, Iter, (Tp)confidence);
    if (method == M_NL_SHINJI_RANSAC) nl_shinji_ransac<Tp>(ad, (Tp)thre_3d, (Tp)thre_nl, Iter, (Tp)confidence);
    if (method == M_NL_SK_RANSAC) nl_shinji_kneip_ransac<Tp>(ad, (Tp)thre_3d, (Tp)thre_2d, (Tp)thre_nl, Iter, (Tp)confidence);
    if (method == M_NONE) {
      read_pose<Tp>(ad, R9, t3);
      ad.setMaxVotes(max_votes ? *max_votes : 1);
      if (mask_in) {
        rpe::MatrixXs m(n, 3); for (int i = 0; i < n; i++) { m(i, 0) = mask_in[i]; m(i, 1) = mask_in[n + i]; m(i, 2) = mask_in[2 * n + i]; }
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
    masks_out(&ad.inlierMask23(), &ad.inlierMask33(), &ad.inlierMaskNN());
  }
  if (iter_io) *iter_io = Iter;
  return RPE_OK;
}

void write_rowmajor(AOOnlyPoseAdapter<float>& adapter, float* R_cw_, float* t_) {
  const rpe::Matrix3<float> R = adapter.getRcw().matrix();  // row-major storage == the reference's Rp = R^T column-major dump (Library.cpp:35-39)
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
    if (const char* s = getenv("RPE_SEED")) rpe::seed(strtoull(s, nullptr, 10)); else rpe::seed(1);
    shinji_ransac2<float>(adapter, thre_3d, updated_iter, confidence);
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

int rpe_run(int method, const rpe_problem* p, double thre_3d, double thre_2d, double thre_nl, int* iter_io, double confidence, uint64_t seed,
            int ls, int score_mode, const short* mask_in, double* R9, double* t3, int* max_votes, short* mask_out) {
  if (!p || !R9 || !t3 || p->n <= 0 || !p->xw) return rpe::set_error(RPE_ERR_ARG, "rpe_run: bad argument");
  const int saved = rpe::Settings::get().score_mode;
  rpe::Settings::get().score_mode = score_mode;
  int rc;
  try {
    rc = p->dtype == RPE_F64 ? run_t<double>(method, p, thre_3d, thre_2d, thre_nl, iter_io, confidence, seed, ls, mask_in, R9, t3, max_votes, mask_out)
                             : run_t<float>(method, p, thre_3d, thre_2d, thre_nl, iter_io, confidence, seed, ls, mask_in, R9, t3, max_votes, mask_out);
  } catch (const rpe::DeviceError& e) {
    rc = rpe::set_error(e.code, e.what());
  } catch (const std::exception& e) {
    rc = rpe::set_error(RPE_ERR_STATE, e.what());
  }
  rpe::Settings::get().score_mode = saved;
  return rc;
}

}  // extern "C"
