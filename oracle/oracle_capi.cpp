// ORACLE -- TEST INFRASTRUCTURE ONLY.  extern "C" surface of the CPU restatement for ctypes
// (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).  Never linked into or loaded by the product.
// PARITY UNPINNED (see orc_linalg.hpp): no reference golden vectors, reference not buildable (Eigen absent).
#include "orc_pose.hpp"
#include "orc_gn.hpp"
#include <cstdio>
#include <chrono>
#include <atomic>
#include <thread>

using namespace orc;

extern "C" {

// arrays are 3 x n column-major (xyz interleaved), dtype float (is_f64 = 0) or double (1); any may be NULL
typedef struct {
  int n;
  const void* bv;   // bearing vectors (camera frame, unit)
  const void* xc;   // points, camera frame ("points_c"; NaN column = no 3-D measurement)
  const void* nc;   // normals, camera frame
  const void* xw;   // points, world frame ("points_g")
  const void* nw;   // normals, world frame
  const void* weights;  // n x wcols column-major or NULL
  int wcols;
  double fx, fy;
} orc_problem;

}  // extern "C"

namespace {

template <class T> MatX<T> load3(const void* p, int n) {
  MatX<T> m(3, p ? n : 0);
  if (p) std::memcpy(m.data(), p, sizeof(T) * 3 * (size_t)n);
  return m;
}
template <class T> void put_pose(const SO3<T>& R, const V3<T>& t, double* R9, double* t3) {
  M3<T> m = R.matrix();
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R9[3 * i + j] = (double)m(i, j);
  t3[0] = (double)t.x; t3[1] = (double)t.y; t3[2] = (double)t.z;
}
template <class T> SE3<T> pose7(const double* q7) {  // (qw qx qy qz tx ty tz), already in T precision
  SE3<T> s;
  s.R.q = Quat<T>((T)q7[0], (T)q7[1], (T)q7[2], (T)q7[3]);
  s.t = V3<T>((T)q7[4], (T)q7[5], (T)q7[6]);
  return s;
}

template <class T> struct Problem {
  MatX<T> bv, xc, nc, xw, nw, w;
  int n;
  explicit Problem(const orc_problem* p) : n(p->n) {
    bv = load3<T>(p->bv, n); xc = load3<T>(p->xc, n); nc = load3<T>(p->nc, n); xw = load3<T>(p->xw, n); nw = load3<T>(p->nw, n);
    if (p->weights) { w.resize(n, p->wcols); std::memcpy(w.data(), p->weights, sizeof(T) * (size_t)n * p->wcols); }
  }
};

template <class T> void export_masks(const MaskCol* m23, const MaskCol* m33, const MaskCol* mnn, int n, short* out) {
  if (!out) return;
  for (int i = 0; i < n; i++) {
    out[i] = m23 ? (*m23)[i] : 0;
    out[n + i] = m33 ? (*m33)[i] : 0;
    out[2 * n + i] = mnn ? (*mnn)[i] : 0;
  }
}

// method ids shared with tests/ (see tests/oracle_lib.py)
enum { M_SHINJI_RANSAC = 0, M_SHINJI_RANSAC2 = 1, M_SHINJI_PROSAC = 2, M_KNEIP_RANSAC = 3, M_KNEIP_PROSAC = 4, M_SK_RANSAC = 5,
       M_SK_PROSAC = 6, M_NL_KNEIP_RANSAC = 7, M_NL_SHINJI_RANSAC = 8, M_NL_SK_RANSAC = 9, M_NONE = 10 };
enum { LS_NONE = 0, LS_SHINJI_INLIERS = 1, LS_NL_BUGCOMPAT = 2, LS_NL_FIXED = 3, LS_SHINJI_ALL = 4 };

template <class T>
int run_pipeline(int method, const orc_problem* op, double thre_3d, double thre_2d, double thre_nl, int* iter_io, double confidence,
                 uint64_t seed, int ls, int adapter_kind_for_none, const short* mask_in, double* R9, double* t3, int* max_votes,
                 short* mask_out) {
  Problem<T> P(op);
  Rand31 rnd(seed);
  int Iter = iter_io ? *iter_io : 0;
  const int n = P.n;
  if (method == M_SHINJI_RANSAC2 || method == M_SHINJI_PROSAC || (method == M_NONE && adapter_kind_for_none == 0)) {
    AOOnlyPoseAdapter<T> ad(P.xc, P.xw);
    ad.setFocal((T)op->fx, (T)op->fy);
    if (op->weights) ad.setWeights(P.w);
    if (method == M_SHINJI_RANSAC2) shinji_ransac2<T>(ad, (T)thre_3d, Iter, (T)confidence, rnd);
    if (method == M_SHINJI_PROSAC) shinji_prosac<T>(ad, (T)thre_3d, Iter, (T)confidence, rnd);
    if (method == M_NONE && mask_in) { MaskX m(n, 2); for (int i = 0; i < n; i++) m(i, 1) = mask_in[n + i]; ad.setInlier(m); ad.cvtInlier(); }
    if (ls == LS_SHINJI_INLIERS) shinji_ls1<T>(ad);
    if (ls == LS_SHINJI_ALL) shinji_ls2<T>(ad);
    put_pose(ad.getRcw(), ad.gettw(), R9, t3);
    if (max_votes) *max_votes = ad.getMaxVotes();
    export_masks<T>(nullptr, &ad.mask33(), nullptr, n, mask_out);
  } else if (method == M_KNEIP_RANSAC || method == M_KNEIP_PROSAC) {
    PnPPoseAdapter<T> ad(P.bv, P.xw);
    ad.setFocal((T)op->fx, (T)op->fy);
    if (op->weights) ad.setWeights(P.w);
    if (method == M_KNEIP_RANSAC) kneip_ransac<T>(ad, (T)thre_2d, Iter, (T)confidence, rnd);
    else kneip_prosac<T>(ad, (T)thre_2d, Iter, (T)confidence, rnd);
    put_pose(ad.getRcw(), ad.gettw(), R9, t3);
    if (max_votes) *max_votes = ad.getMaxVotes();
    export_masks<T>(&ad.mask23(), nullptr, nullptr, n, mask_out);
  } else if (method == M_SHINJI_RANSAC || method == M_SK_RANSAC || method == M_SK_PROSAC || (method == M_NONE && adapter_kind_for_none == 1)) {
    AOPoseAdapter<T> ad(P.bv, P.xc, P.xw);
    ad.setFocal((T)op->fx, (T)op->fy);
    if (op->weights) ad.setWeights(P.w);
    if (method == M_SHINJI_RANSAC) shinji_ransac<T>(ad, (T)thre_3d, Iter, (T)confidence, rnd);
    if (method == M_SK_RANSAC) shinji_kneip_ransac<T>(ad, (T)thre_3d, (T)thre_2d, Iter, (T)confidence, rnd);
    if (method == M_SK_PROSAC) shinji_kneip_prosac<T>(ad, (T)thre_3d, (T)thre_2d, Iter, (T)confidence, rnd);
    if (method == M_NONE && mask_in) {
      MaskX m(n, 2); for (int i = 0; i < n; i++) { m(i, 0) = mask_in[i]; m(i, 1) = mask_in[n + i]; }
      ad.setInlier(m); PnPPoseAdapter<T>* p = &ad; p->cvtInlier(); ad.cvtInlier();
    }
    if (ls == LS_SHINJI_INLIERS) shinji_ls<T>(ad);
    put_pose(ad.getRcw(), ad.gettw(), R9, t3);
    if (max_votes) *max_votes = ad.getMaxVotes();
    export_masks<T>(&ad.mask23(), &ad.mask33(), nullptr, n, mask_out);
  } else {
    NormalAOPoseAdapter<T> ad(P.bv, P.xc, P.nc, P.xw, P.nw);
    ad.setFocal((T)op->fx, (T)op->fy);
    if (method == M_NL_KNEIP_RANSAC) nl_kneip_ransac<T>(ad, (T)thre_2d, (T)thre_nl, Iter, (T)confidence, rnd);
    if (method == M_NL_SHINJI_RANSAC) nl_shinji_ransac<T>(ad, (T)thre_3d, (T)thre_nl, Iter, (T)confidence, rnd);
    if (method == M_NL_SK_RANSAC) nl_shinji_kneip_ransac<T>(ad, (T)thre_3d, (T)thre_2d, (T)thre_nl, Iter, (T)confidence, rnd);
    if (method == M_NONE) {
      // caller supplies pose (R9,t3 in) + masks + max_votes; used to test the LS stage in isolation
      M3<T> R; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = (T)R9[3 * i + j];
      SO3<T> so3(R); so3.ok = true;
      ad.setRcw(so3); ad.sett(V3<T>((T)t3[0], (T)t3[1], (T)t3[2]));
      ad.setMaxVotes(max_votes ? *max_votes : 1);
      if (mask_in) {
        MaskX m(n, 3); for (int i = 0; i < n; i++) { m(i, 0) = mask_in[i]; m(i, 1) = mask_in[n + i]; m(i, 2) = mask_in[2 * n + i]; }
        ad.setInlier(m);
      }
    }
    // TestMain.cpp:215-221: 'opt' runs with unit weights, 'dw' after setWeights(all_weights)
    if (op->weights) ad.setWeights(P.w);
    if (ls == LS_NL_BUGCOMPAT) nl_shinji_kneip_ls<T>(ad, true);
    if (ls == LS_NL_FIXED) nl_shinji_kneip_ls<T>(ad, false);
    if (ls == LS_SHINJI_INLIERS) { AOPoseAdapter<T>* p = &ad; p->cvtInlier(); shinji_ls<T>(ad); }
    put_pose(ad.getRcw(), ad.gettw(), R9, t3);
    if (max_votes) *max_votes = ad.getMaxVotes();
    export_masks<T>(&ad.mask23(), &ad.mask33(), &ad.maskNN(), n, mask_out);
  }
  if (iter_io) *iter_io = Iter;
  return 0;
}

enum { V_33 = 0, V_23 = 1, V_33_23 = 2, V_NN_23 = 3, V_NN_33 = 4, V_NN_33_23 = 5, V_23_MATRIX = 6 };

template <class T>
void run_votes(int kind, const orc_problem* op, const double* poses7, int H, double thre_3d, double cos_thr, double cos_nl, int* votes,
               short* mask_out, int mask_for) {
  Problem<T> P(op);
  const int n = P.n;
  for (int h = 0; h < H; h++) {
    SE3<T> s = pose7<T>(poses7 + 7 * h);
    MaskX m(n, 3);
    int v = 0;
    if (kind == V_33) { AOOnlyPoseAdapter<T> ad(P.xc, P.xw); MaskX m2(n, 2); v = vote_33<T>(ad, s, (T)thre_3d, m2); for (int i = 0; i < n; i++) m(i, 1) = m2(i, 1); }
    else if (kind == V_23 || kind == V_23_MATRIX) { PnPPoseAdapter<T> ad(P.bv, P.xw); MaskX m1(n, 1); v = vote_23<T>(ad, s, (T)cos_thr, m1, kind == V_23_MATRIX); for (int i = 0; i < n; i++) m(i, 0) = m1(i, 0); }
    else if (kind == V_33_23) { AOPoseAdapter<T> ad(P.bv, P.xc, P.xw); MaskX m2(n, 2); v = vote_33_23<T>(ad, s, (T)thre_3d, (T)cos_thr, m2); for (int i = 0; i < n; i++) { m(i, 0) = m2(i, 0); m(i, 1) = m2(i, 1); } }
    else {
      NormalAOPoseAdapter<T> ad(P.bv, P.xc, P.nc, P.xw, P.nw);
      if (kind == V_NN_23) v = vote_nn_23<T>(ad, s, (T)cos_thr, (T)cos_nl, m);
      if (kind == V_NN_33) v = vote_nn_33<T>(ad, s, (T)thre_3d, (T)cos_nl, m);
      if (kind == V_NN_33_23) v = vote_nn_33_23<T>(ad, s, (T)thre_3d, (T)cos_thr, (T)cos_nl, m);
    }
    votes[h] = v;
    if (mask_out && h == mask_for) std::memcpy(mask_out, m.d.data(), sizeof(short) * 3 * (size_t)n);
  }
}

}  // namespace

extern "C" {

int orc_abi_version() { return 1; }

// ---- closed form (A1..A3) --------------------------------------------------------------------
// shinji() on the first K columns. R9 row-major. returns 0 ok, 1 if SOPHUS_ENSURE would have aborted.
int orc_shinji(int is_f64, const void* xw, const void* xc, int n, int K, double* R9, double* t3) {
  if (is_f64) { MatX<double> a = load3<double>(xw, n), b = load3<double>(xc, n); SE3<double> s = shinji<double>(a, b, K); put_pose(s.R, s.t, R9, t3); return s.R.ok ? 0 : 1; }
  MatX<float> a = load3<float>(xw, n), b = load3<float>(xc, n); SE3<float> s = shinji<float>(a, b, K); put_pose(s.R, s.t, R9, t3); return s.R.ok ? 0 : 1;
}
// fp64 shinji evaluated on fp32 inputs: the accuracy oracle for the 307k config (SURVEY.md section 7 "Precision")
int orc_shinji_f32in_f64(const float* xw, const float* xc, int n, double* R9, double* t3) {
  MatX<double> a(3, n), b(3, n);
  for (size_t i = 0; i < (size_t)3 * n; i++) { a.data()[i] = xw[i]; b.data()[i] = xc[i]; }
  SE3<double> s = shinji<double>(a, b, n); put_pose(s.R, s.t, R9, t3); return s.R.ok ? 0 : 1;
}
// Library.cpp:17-45 ao(): float, AOOnlyPoseAdapter + shinji_ls2, R_cw_ row-major (no stdout print)
void orc_ao(float* x_w, float* x_c, int n, float* R_cw, float* t) {
  MatX<float> Xw = load3<float>(x_w, n), Xc = load3<float>(x_c, n);
  AOOnlyPoseAdapter<float> adapter(Xc, Xw);
  adapter.setFocal(555.f, 555.f);
  shinji_ls2<float>(adapter);
  M3<float> R = adapter.getRcw().matrix();
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R_cw[3 * i + j] = R(i, j);
  V3<float> tw = adapter.gettw();
  t[0] = tw.x; t[1] = tw.y; t[2] = tw.z;
}
// Library.cpp:47-75 ao_ransac(): Iter=1000, thre_3d=0.1, confidence=0.99999, then shinji_ls1
void orc_ao_ransac(float* x_w, float* x_c, int n, float* R_cw, float* t, uint64_t seed, int* iter_out, int* votes_out) {
  MatX<float> Xw = load3<float>(x_w, n), Xc = load3<float>(x_c, n);
  AOOnlyPoseAdapter<float> adapter(Xc, Xw);
  adapter.setFocal(555.f, 555.f);
  int it = 1000;
  Rand31 rnd(seed);
  shinji_ransac2<float>(adapter, 0.1f, it, 0.99999f, rnd);
  shinji_ls1<float>(adapter);
  M3<float> R = adapter.getRcw().matrix();
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R_cw[3 * i + j] = R(i, j);
  V3<float> tw = adapter.gettw();
  t[0] = tw.x; t[1] = tw.y; t[2] = tw.z;
  if (iter_out) *iter_out = it;
  if (votes_out) *votes_out = adapter.getMaxVotes();
}

// ---- full pipelines -----------------------------------------------------------------------------
int orc_run(int is_f64, int method, const orc_problem* p, double thre_3d, double thre_2d, double thre_nl, int* iter_io, double confidence,
            uint64_t seed, int ls, int adapter_kind_for_none, const short* mask_in, double* R9, double* t3, int* max_votes, short* mask_out) {
  return is_f64 ? run_pipeline<double>(method, p, thre_3d, thre_2d, thre_nl, iter_io, confidence, seed, ls, adapter_kind_for_none, mask_in, R9, t3, max_votes, mask_out)
                : run_pipeline<float>(method, p, thre_3d, thre_2d, thre_nl, iter_io, confidence, seed, ls, adapter_kind_for_none, mask_in, R9, t3, max_votes, mask_out);
}

// ---- explicit hypothesis streams (orc_pose.hpp "explicit hypothesis streams") --------------------------------------------------
// The hypotheses `method` generates in `iters` iterations from `seed` (its sampler + minimal solvers, nothing voted on):
// q7_out[cap x 7] = qw qx qy qz tx ty tz, first_out[iters + 1].  Returns the number of hypotheses, or -1 if cap is too small.
int orc_hypotheses(int is_f64, int method, const orc_problem* p, int iters, uint64_t seed, double* q7_out, int cap, int* first_out) {
  HypList list;
  capture_sink() = &list;
  int it = iters, mv = 0;
  double R9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t3[3] = {0, 0, 0};
  if (is_f64) run_pipeline<double>(method, p, 1.0, 1.0, 1.0, &it, 0.99, seed, 0, 0, nullptr, R9, t3, &mv, nullptr);
  else run_pipeline<float>(method, p, 1.0, 1.0, 1.0, &it, 0.99, seed, 0, 0, nullptr, R9, t3, &mv, nullptr);
  capture_sink() = nullptr;
  if (list.first.empty()) list.first.assign((size_t)iters + 1, 0);
  const int H = (int)(list.q7.size() / 7);
  if (H > cap || (int)list.first.size() != iters + 1) return -1;
  if (!list.q7.empty()) std::memcpy(q7_out, list.q7.data(), sizeof(double) * list.q7.size());
  std::memcpy(first_out, list.first.data(), sizeof(int) * list.first.size());
  return H;
}
// orc_run with the hypotheses of iteration i taken from poses7[first[i] .. first[i+1]) instead of the sampler + minimal solvers
int orc_run_replay(int is_f64, int method, const orc_problem* p, const double* poses7, const int* first, int list_iters, double thre_3d,
                   double thre_2d, double thre_nl, int* iter_io, double confidence, int ls, double* R9, double* t3, int* max_votes, short* mask_out) {
  HypList list;
  list.first.assign(first, first + list_iters + 1);
  list.q7.assign(poses7, poses7 + 7 * (size_t)first[list_iters]);
  replay_source() = &list;
  const int rc = orc_run(is_f64, method, p, thre_3d, thre_2d, thre_nl, iter_io, confidence, 1, ls, 0, nullptr, R9, t3, max_votes, mask_out);
  replay_source() = nullptr;
  return rc;
}

// ---- vote loops on an explicit hypothesis list (V1..V8); poses7 = H x (qw qx qy qz tx ty tz) --------
// thresholds are passed already converted the way the reference does it: cos_thr = cos(atan(thre_2d/f)), cos_nl = cos(nl_thre)
void orc_votes(int is_f64, int kind, const orc_problem* p, const double* poses7, int H, double thre_3d, double cos_thr, double cos_nl,
               int* votes, short* mask_out, int mask_for) {
  if (is_f64) run_votes<double>(kind, p, poses7, H, thre_3d, cos_thr, cos_nl, votes, mask_out, mask_for);
  else run_votes<float>(kind, p, poses7, H, thre_3d, cos_thr, cos_nl, votes, mask_out, mask_for);
}
// cos(atan(thre_2d / f)) and cos(nl_thre) evaluated in Tp exactly as the solvers do (AbsoluteOrientation.hpp:373)
double orc_cos_thr(int is_f64, double thre_2d, double fx, double fy) {
  if (is_f64) { double f = (fx + fy) / 2; return std::cos(std::atan(thre_2d / f)); }
  float f = ((float)fx + (float)fy) / 2; return (double)std::cos(std::atan((float)thre_2d / f));
}
double orc_cos_nl(int is_f64, double nl_thre) { return is_f64 ? std::cos(nl_thre) : (double)std::cos((float)nl_thre); }

// per-correspondence 3D residual norms |Xc - (q*Xw + t)| in Tp (lets tests reason about threshold-boundary cases)
void orc_residual_33(int is_f64, const void* xw, const void* xc, int n, const double* q7, double* out) {
  if (is_f64) { MatX<double> a = load3<double>(xw, n), b = load3<double>(xc, n); SE3<double> s = pose7<double>(q7);
    for (int i = 0; i < n; i++) out[i] = norm(b.col3(i) - (s.R * a.col3(i) + s.t)); }
  else { MatX<float> a = load3<float>(xw, n), b = load3<float>(xc, n); SE3<float> s = pose7<float>(q7);
    for (int i = 0; i < n; i++) out[i] = (double)norm(b.col3(i) - (s.R * a.col3(i) + s.t)); }
}
void orc_cos_23(int is_f64, const void* xw, const void* bv, int n, const double* q7, double* out) {
  if (is_f64) { MatX<double> a = load3<double>(xw, n), b = load3<double>(bv, n); SE3<double> s = pose7<double>(q7);
    for (int i = 0; i < n; i++) { V3<double> pc = s.R * a.col3(i) + s.t; pc = pc / norm(pc); out[i] = dot(pc, b.col3(i)); } }
  else { MatX<float> a = load3<float>(xw, n), b = load3<float>(bv, n); SE3<float> s = pose7<float>(q7);
    for (int i = 0; i < n; i++) { V3<float> pc = s.R * a.col3(i) + s.t; pc = pc / norm(pc); out[i] = (double)dot(pc, b.col3(i)); } }
}
void orc_cos_nn(int is_f64, const void* nw, const void* nc, int n, const double* q7, double* out) {
  if (is_f64) { MatX<double> a = load3<double>(nw, n), b = load3<double>(nc, n); SE3<double> s = pose7<double>(q7);
    for (int i = 0; i < n; i++) out[i] = dot(b.col3(i), s.R * a.col3(i)); }
  else { MatX<float> a = load3<float>(nw, n), b = load3<float>(nc, n); SE3<float> s = pose7<float>(q7);
    for (int i = 0; i < n; i++) out[i] = (double)dot(b.col3(i), s.R * a.col3(i)); }
}

// R1 lsq_pnp (P3P.hpp:472-502) through the PnPPoseAdapter at pose q7: out2[0] = the reference's total (the terms added one after the other
// in Tp), out2[1] = the SAME Tp terms added in double (what the device kernel adds); terms (optional, n doubles) = every getError(i)
void orc_lsq_pnp(int is_f64, const void* xw, const void* bv, int n, const double* q7, double* out2, double* terms) {
  if (is_f64) {
    MatX<double> a = load3<double>(xw, n), b = load3<double>(bv, n); SE3<double> s = pose7<double>(q7);
    PnPPoseAdapter<double> ad(b, a); ad.setRcw(s.R); ad.sett(s.t);
    out2[0] = lsq_pnp<double>(ad); out2[1] = 0;
    for (int i = 0; i < n; i++) { const double e = ad.getError(i); out2[1] += e; if (terms) terms[i] = e; }
  } else {
    MatX<float> a = load3<float>(xw, n), b = load3<float>(bv, n); SE3<float> s = pose7<float>(q7);
    PnPPoseAdapter<float> ad(b, a); ad.setRcw(s.R); ad.sett(s.t);
    out2[0] = (double)lsq_pnp<float>(ad); out2[1] = 0;
    for (int i = 0; i < n; i++) { const float e = ad.getError(i); out2[1] += (double)e; if (terms) terms[i] = (double)e; }
  }
}

// ---- U1, samplers -------------------------------------------------------------------------------
int orc_ransac_update_num_iters(int is_f64, double p, double ep, int modelPoints, int maxIters) {
  return is_f64 ? RANSACUpdateNumIters<double>(p, ep, modelPoints, maxIters) : RANSACUpdateNumIters<float>((float)p, (float)ep, modelPoints, maxIters);
}
void orc_rand31(uint64_t seed, int count, int* out) { Rand31 r(seed); for (int i = 0; i < count; i++) out[i] = r(); }
void orc_random_elements(int n, int m, uint64_t seed, int draws, int* out) {
  RandomElements re(n); Rand31 r(seed); std::vector<int> v;
  for (int d = 0; d < draws; d++) { re.run(m, &v, r); for (int k = 0; k < m; k++) out[d * m + k] = v[k]; }
}
void orc_prosac_samples(int is_f64, int m, int n, uint64_t seed, int draws, int* out) {
  Rand31 r(seed); std::vector<int> v;
  if (is_f64) { ProsacSampler<double> ps(m, n); for (int d = 0; d < draws; d++) { ps.sample(&v, r); for (int k = 0; k < m; k++) out[d * m + k] = v[k]; } }
  else { ProsacSampler<float> ps(m, n); for (int d = 0; d < draws; d++) { ps.sample(&v, r); for (int k = 0; k < m; k++) out[d * m + k] = v[k]; } }
}
void orc_sort_indexes(int is_f64, const void* w, int n, int* out) {
  std::vector<int> idx;
  if (is_f64) { std::vector<double> v((const double*)w, (const double*)w + n); idx = sortIndexes<double>(v); }
  else { std::vector<float> v((const float*)w, (const float*)w + n); idx = sortIndexes<float>(v); }
  for (int i = 0; i < n; i++) out[i] = idx[i];
}

// ---- minimal solvers (H1) -----------------------------------------------------------------------
// xw, bv: 3 x 4 column-major (4th column used by orc_kneip only). sols: up to 4 x (R9 row-major, t3). returns count
int orc_kneip_main(int is_f64, const void* xw, const void* bv, double* sols12) {
  int cnt = 0;
  if (is_f64) { MatX<double> a = load3<double>(xw, 4), b = load3<double>(bv, 4); std::vector<SE3<double> > v; kneip_main<double>(a, b, &v);
    for (auto& s : v) { put_pose(s.R, s.t, sols12 + 12 * cnt, sols12 + 12 * cnt + 9); cnt++; } }
  else { MatX<float> a = load3<float>(xw, 4), b = load3<float>(bv, 4); std::vector<SE3<float> > v; kneip_main<float>(a, b, &v);
    for (auto& s : v) { put_pose(s.R, s.t, sols12 + 12 * cnt, sols12 + 12 * cnt + 9); cnt++; } }
  return cnt;
}
int orc_kneip(int is_f64, const void* xw, const void* bv, double* R9, double* t3) {
  if (is_f64) { MatX<double> a = load3<double>(xw, 4), b = load3<double>(bv, 4); SE3<double> s; if (!kneip<double>(a, b, &s)) return 0; put_pose(s.R, s.t, R9, t3); return 1; }
  MatX<float> a = load3<float>(xw, 4), b = load3<float>(bv, 4); SE3<float> s; if (!kneip<float>(a, b, &s)) return 0; put_pose(s.R, s.t, R9, t3); return 1;
}
void orc_o4_roots(const double* p5, double* roots4) { o4_roots<double>(p5, roots4); }
// six 3-vectors in the order of nl_2p's arguments (pt1_c nl1_c pt2_c pt1_w nl1_w pt2_w), double in, T arithmetic
void orc_nl_2p(int is_f64, const double* v18, double* R9, double* t3) {
  if (is_f64) { V3<double> a[6]; for (int i = 0; i < 6; i++) a[i] = V3<double>(v18[3 * i], v18[3 * i + 1], v18[3 * i + 2]);
    SE3<double> s; nl_2p<double>(a[0], a[1], a[2], a[3], a[4], a[5], &s); put_pose(s.R, s.t, R9, t3); }
  else { V3<float> a[6]; for (int i = 0; i < 6; i++) a[i] = V3<float>((float)v18[3 * i], (float)v18[3 * i + 1], (float)v18[3 * i + 2]);
    SE3<float> s; nl_2p<float>(a[0], a[1], a[2], a[3], a[4], a[5], &s); put_pose(s.R, s.t, R9, t3); }
}
// find_opt_cc with pose + 2D mask supplied (L2)
void orc_find_opt_cc(int is_f64, const orc_problem* p, const double* R9, const short* mask23, double* c3) {
  if (is_f64) { Problem<double> P(p); NormalAOPoseAdapter<double> ad(P.bv, P.xc, P.nc, P.xw, P.nw);
    M3<double> R; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = R9[3 * i + j];
    ad.setRcw(SO3<double>(R)); MaskX m(P.n, 1); for (int i = 0; i < P.n; i++) m(i, 0) = mask23[i]; ad.setInlier(m);
    V3<double> c = find_opt_cc<double>(ad); c3[0] = c.x; c3[1] = c.y; c3[2] = c.z; }
  else { Problem<float> P(p); NormalAOPoseAdapter<float> ad(P.bv, P.xc, P.nc, P.xw, P.nw);
    M3<float> R; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = (float)R9[3 * i + j];
    ad.setRcw(SO3<float>(R)); MaskX m(P.n, 1); for (int i = 0; i < P.n; i++) m(i, 0) = mask23[i]; ad.setInlier(m);
    V3<float> c = find_opt_cc<float>(ad); c3[0] = c.x; c3[1] = c.y; c3[2] = c.z; }
}

// ---- E1 -------------------------------------------------------------------------------------------
void orc_calc_err(const double* Rgt9, const double* tgt3, const double* Rse9, const double* tse3, double* out2) {
  M3<double> A, B;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { A(i, j) = Rgt9[3 * i + j]; B(i, j) = Rse9[3 * i + j]; }
  calc_err<double>(A, V3<double>(tgt3[0], tgt3[1], tgt3[2]), B, V3<double>(tse3[0], tse3[1], tse3[2]), out2);
}
void orc_calc_percentage_err(const double* Rgt9, const double* tgt3, const double* Rse9, const double* tse3, double* out2) {
  M3<double> A, B;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { A(i, j) = Rgt9[3 * i + j]; B(i, j) = Rse9[3 * i + j]; }
  MatX<double> e(3, 0);
  AOOnlyPoseAdapter<double> ad(e, e);
  ad.setRcw(SO3<double>(B)); ad.sett(V3<double>(tse3[0], tse3[1], tse3[2]));
  calc_percentage_err<double>(SO3<double>(A), V3<double>(tgt3[0], tgt3[1], tgt3[2]), &ad, out2);
}

// ---- X1 / linear algebra probes -------------------------------------------------------------------
void orc_se3_exp(const double* a6, double* R9, double* t3) { SE3<double> s = SE3<double>::exp(a6); put_pose(s.R, s.t, R9, t3); }
void orc_se3_log(const double* R9, const double* t3, double* a6) {
  M3<double> R; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = R9[3 * i + j];
  SE3<double> s(SO3<double>(R), V3<double>(t3[0], t3[1], t3[2])); s.log(a6);
}
void orc_svd3(const double* A9, double* U9, double* s3, double* V9) {
  M3<double> A; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A(i, j) = A9[3 * i + j];
  SVD3<double> d = svd3(A);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { U9[3 * i + j] = d.U(i, j); V9[3 * i + j] = d.V(i, j); }
  for (int i = 0; i < 3; i++) s3[i] = d.s[i];
}
void orc_quat_from_R(int is_f64, const double* R9, double* q4) {
  if (is_f64) { M3<double> R; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = R9[3 * i + j]; Quat<double> q = quat_from_matrix(R); q4[0] = q.w; q4[1] = q.x; q4[2] = q.y; q4[3] = q.z; }
  else { M3<float> R; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = (float)R9[3 * i + j]; Quat<float> q = quat_from_matrix(R); q4[0] = q.w; q4[1] = q.x; q4[2] = q.y; q4[3] = q.z; }
}

// ---- Gauss-Newton (north-star formulation), fp64 arithmetic on float or double inputs ----------------
void orc_gn_normal_eq(int in_f64, int kind, const void* a, const void* b, const void* c, const short* mask, const void* weight, long n,
                      const double* pose12, double* out29) {
  NormalEq ne;
  if (in_f64) gn_normal_eq<double>(kind, (const double*)a, (const double*)b, (const double*)c, mask, (const double*)weight, n, pose12, &ne);
  else gn_normal_eq<float>(kind, (const float*)a, (const float*)b, (const float*)c, mask, (const float*)weight, n, pose12, &ne);
  ne.pack(out29);
}
void orc_gn_normal_eq_robust(int in_f64, int kind, const void* a, const void* b, const void* c, const short* mask, const void* weight, long n,
                             const double* pose12, int robust, double robust_k, double* out29) {
  NormalEq ne;
  if (in_f64) gn_normal_eq<double>(kind, (const double*)a, (const double*)b, (const double*)c, mask, (const double*)weight, n, pose12, &ne, robust, robust_k);
  else gn_normal_eq<float>(kind, (const float*)a, (const float*)b, (const float*)c, mask, (const float*)weight, n, pose12, &ne, robust, robust_k);
  ne.pack(out29);
}
int orc_gn_solve(const double* packed29, double* delta6) {
  double H[6][6] = {{0}}, g[6]; int k = 0;
  for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) H[a][b] = packed29[k++];
  for (int a = 0; a < 6; a++) g[a] = packed29[k++];
  return gn_solve6(H, g, delta6) ? 0 : 1;
}
void orc_gn_apply(const double* delta6, double* pose12) { gn_apply(delta6, pose12); }
// up to 3 terms; per term: kind, a, b, c, mask, weight, scale
int orc_gn_refine(int in_f64, int nterms, const int* kinds, const void** as, const void** bs, const void** cs, const short** masks,
                  const void** weights, const double* scales, const int* robusts, const double* robust_ks, long n, double* pose12, int max_iter,
                  double tol, double* last_step, double* final_cost) {
  GnTerm t[4];
  for (int k = 0; k < nterms && k < 4; k++) { t[k].kind = kinds[k]; t[k].a = as[k]; t[k].b = bs[k]; t[k].c = cs ? cs[k] : nullptr;
    t[k].mask = masks ? masks[k] : nullptr; t[k].weight = weights ? weights[k] : nullptr; t[k].scale = scales ? scales[k] : 1.0;
    t[k].robust = robusts ? robusts[k] : 0; t[k].robust_k = robust_ks ? robust_ks[k] : 1.0; }
  return in_f64 ? gn_refine<double>(t, nterms, n, pose12, max_iter, tol, last_step, final_cost)
                : gn_refine<float>(t, nterms, n, pose12, max_iter, tol, last_step, final_cost);
}

// ---- CPU baseline legs (bench.py cpu_baseline): wall time of `reps` repetitions, seconds ------------------
// shinji_ls2<float> through the adapter's virtual getters, as Library.cpp:ao() runs it
double orc_time_ao(float* x_w, float* x_c, int n, int reps, float* R_cw, float* t) {
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) orc_ao(x_w, x_c, n, R_cw, t);
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
// one fp64 GN normal-equation pass (the CPU counterpart of one bench "step")
double orc_time_gn_p2p(const float* x_w, const float* x_c, long n, int reps, const double* pose12, double* out29) {
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) { NormalEq ne; gn_normal_eq<float>(GN_P2P, x_w, x_c, nullptr, nullptr, nullptr, n, pose12, &ne); ne.pack(out29); }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
// the same pass spread over `threads` host threads (contiguous shards, per-thread sums added at the end): the all-core CPU figure
// SURVEY.md 8(d) asks for beside the single-thread one.  Every thread runs its shard `reps` times; the wall time of the whole is returned.
double orc_time_gn_p2p_threads(const float* x_w, const float* x_c, long n, int reps, const double* pose12, double* out29, int threads) {
  if (threads < 1) threads = 1;
  std::vector<NormalEq> part((size_t)threads);
  std::vector<std::thread> pool;
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < threads; k++)
    pool.emplace_back([&, k]() {
      const long lo = n * k / threads, hi = n * (k + 1) / threads;
      for (int r = 0; r < reps; r++) { NormalEq ne; gn_normal_eq<float>(GN_P2P, x_w + 3 * lo, x_c + 3 * lo, nullptr, nullptr, nullptr, hi - lo, pose12, &ne); part[(size_t)k] = ne; }
    });
  for (std::thread& th : pool) th.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  NormalEq tot;
  for (const NormalEq& ne : part) { for (int a = 0; a < 6; a++) { tot.g[a] += ne.g[a]; for (int b = a; b < 6; b++) tot.H[a][b] += ne.H[a][b]; } tot.cost += ne.cost; tot.wsum += ne.wsum; }
  tot.pack(out29);
  return dt;
}
// ---- all-core variants of the SAME restatement ("an OpenMP all-core variant of the same restatement for fairness").
// The reference is single-threaded; these spread its O(N) loops over `threads` host threads by contiguous index ranges, every thread
// running the reference's per-index body (virtual getters, by-value 3-vectors) on its range, per-thread partial sums added in thread
// order.  One pool per call, phases separated by a spinning barrier (a thread spawn per phase would cost more than the phase).
namespace {
struct SpinBarrier {
  std::atomic<int> count{0}, sense{0};
  int n;
  explicit SpinBarrier(int n_) : n(n_) {}
  void wait() {
    const int s = sense.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1) { count.store(0, std::memory_order_relaxed); sense.store(s ^ 1, std::memory_order_release); }
    else {   // spin briefly, then let the scheduler run whoever is late (a busy box would otherwise burn a time slice per barrier)
      for (unsigned spins = 0; sense.load(std::memory_order_acquire) == s; spins++) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if (spins > 2000) std::this_thread::yield();
      }
    }
  }
};
}  // namespace
// shinji_ls2<float> (= Library.cpp ao()): copy-in, gather through the getters, centroid pass, covariance pass -- each over index ranges;
// the O(1) tail (M / cols, SVD, det test, t) on thread 0.  Returns the wall time of `reps` repetitions.
double orc_time_ao_threads(float* x_w, float* x_c, int n, int reps, int threads, float* R_cw, float* t) {
  if (threads < 1) threads = 1;
  if (threads > n) threads = n > 0 ? n : 1;
  typedef float T;
  MatX<T> Xw0(3, n), Xc0(3, n), Xw(3, n), Xc(3, n);
  AOOnlyPoseAdapter<T> adapter(Xc0, Xw0);
  adapter.setFocal(555.f, 555.f);
  std::vector<V3<T>> pCw((size_t)threads), pCc((size_t)threads);
  std::vector<M3<T>> pM((size_t)threads);
  V3<T> Cw, Cc;
  SpinBarrier bar(threads);
  auto body = [&](int k) {
    const int lo = (int)((long)n * k / threads), hi = (int)((long)n * (k + 1) / threads);
    for (int r = 0; r < reps; r++) {
      for (int i = lo; i < hi; i++) {   // Library.cpp:20-22: the arrays copied into the matrices the adapter refers to
        Xw0.set_col3(i, V3<T>(x_w[3 * i], x_w[3 * i + 1], x_w[3 * i + 2]));
        Xc0.set_col3(i, V3<T>(x_c[3 * i], x_c[3 * i + 1], x_c[3 * i + 2]));
      }
      for (int i = lo; i < hi; i++) { Xw.set_col3(i, adapter.getPointGlob(i)); Xc.set_col3(i, adapter.getPointCurr(i)); }   // shinji_ls2 :331-336
      V3<T> cw, cc;
      for (int i = lo; i < hi; i++) { cw = cw + Xw.col3(i); cc = cc + Xc.col3(i); }   // shinji :57-62
      pCw[(size_t)k] = cw; pCc[(size_t)k] = cc;
      bar.wait();
      if (k == 0) {
        V3<T> a, b;
        for (int j = 0; j < threads; j++) { a = a + pCw[(size_t)j]; b = b + pCc[(size_t)j]; }
        Cw = a / (T)n; Cc = b / (T)n;
      }
      bar.wait();
      M3<T> M;
      T sigma_w = 0, sigma_c = 0;
      for (int i = lo; i < hi; i++) {   // :66-74
        V3<T> Aw = Xw.col3(i) - Cw; sigma_w += norm(Aw);
        V3<T> Ac = Xc.col3(i) - Cc; sigma_c += norm(Ac);
        M = M + outer(Ac, Aw);
      }
      (void)sigma_w; (void)sigma_c;
      pM[(size_t)k] = M;
      bar.wait();
      if (k == 0) {
        M3<T> Mt;
        for (int j = 0; j < threads; j++) Mt = Mt + pM[(size_t)j];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Mt(i, j) = Mt(i, j) / (T)n;
        SVD3<T> d = svd3(Mt);
        M3<T> Tmp = d.U * transpose(d.V);
        SO3<T> R;
        if (det(Tmp) < T(0)) { M3<T> I = M3<T>::identity(); I(2, 2) = -1; R = SO3<T>(d.U * I * transpose(d.V)); }
        else R = SO3<T>(Tmp);
        V3<T> tt = Cc - R * Cw;
        adapter.setRcw(R); adapter.sett(tt);
      }
      bar.wait();
    }
  };
  std::vector<std::thread> pool;
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 1; k < threads; k++) pool.emplace_back(body, k);
  body(0);
  for (std::thread& th : pool) th.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  M3<T> R = adapter.getRcw().matrix();
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R_cw[3 * i + j] = R(i, j);
  V3<T> tw = adapter.gettw();
  t[0] = tw.x; t[1] = tw.y; t[2] = tw.z;
  return dt;
}
// the 3D-3D vote loop (V1/V2, vote_33's body) over index ranges: every thread writes its range of the mask and counts its votes
double orc_time_votes33_threads(const float* x_w, const float* x_c, int n, const double* poses7, int H, float thre_3d, int* votes, int threads) {
  if (threads < 1) threads = 1;
  if (threads > n) threads = n > 0 ? n : 1;
  MatX<float> Xw = load3<float>(x_w, n), Xc = load3<float>(x_c, n);
  AOOnlyPoseAdapter<float> ad(Xc, Xw);
  MaskX m(n, 2);
  std::vector<int> part((size_t)threads);
  SpinBarrier bar(threads);
  auto body = [&](int k) {
    const int lo = (int)((long)n * k / threads), hi = (int)((long)n * (k + 1) / threads);
    for (int h = 0; h < H; h++) {
      const SE3<float> s = pose7<float>(poses7 + 7 * h);
      int v = 0;
      for (int c = lo; c < hi; c++) {   // AbsoluteOrientation.hpp:190-200
        m(c, 1) = 0;
        if (ad.isValid(c)) {
          V3<float> e = ad.getPointCurr(c) - (s.R * ad.getPointGlob(c) + s.t);
          if (norm(e) < thre_3d) { m(c, 1) = 1; v++; }
        }
      }
      part[(size_t)k] = v;
      bar.wait();
      if (k == 0) { int tot = 0; for (int j = 0; j < threads; j++) tot += part[(size_t)j]; votes[h] = tot; }
      bar.wait();
    }
  };
  std::vector<std::thread> pool;
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 1; k < threads; k++) pool.emplace_back(body, k);
  body(0);
  for (std::thread& th : pool) th.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
// vote loop of shinji_ransac2 (V2) for H hypotheses, float
double orc_time_votes33(const float* x_w, const float* x_c, int n, const double* poses7, int H, float thre_3d, int* votes) {
  MatX<float> Xw = load3<float>(x_w, n), Xc = load3<float>(x_c, n);
  AOOnlyPoseAdapter<float> ad(Xc, Xw);
  MaskX m(n, 2);
  auto t0 = std::chrono::steady_clock::now();
  for (int h = 0; h < H; h++) { SE3<float> s = pose7<float>(poses7 + 7 * h); votes[h] = vote_33<float>(ad, s, thre_3d, m); }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
