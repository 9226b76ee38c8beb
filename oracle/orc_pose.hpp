// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header).  PARITY UNPINNED (no reference
// golden vectors exist; the reference cannot be compiled here because Eigen3 is absent).
//
// orc_pose.hpp: CPU restatement of the reference's hot path, function by function, keeping its
// structure (per-index virtual getters returning a 3-vector by value, scalar loops in Tp, masks as
// short, strict inequalities, best-so-far on strict '>').  Every function cites the file:line of
// /root/reference it follows.  Deviations from the reference are listed here and nowhere hidden:
//   D1  inlier index lists are int, not short (reference overflows for N > 32767:
//       pose/AOOnlyPoseAdapter.hpp:222-231, AbsoluteOrientation.hpp:279-288).
//   D2  rand() is an explicit, seedable PCG32-based 31-bit stream (orc::Rand31).
//   D3  PROSAC's "n-th point" index is clamped to N-1 (reference reads out of bounds when n == N,
//       pose/Utility.hpp:238 with AOOnlyPoseAdapter.hpp:246-254).
//   D4  where SOPHUS_ENSURE would abort the process, SO3::ok is cleared and the hypothesis is skipped.
#pragma once
#include "orc_linalg.hpp"
#include <numeric>
#include <climits>

namespace orc {

typedef std::vector<short> MaskCol;
struct MaskX {  // Matrix<short,Dynamic,Dynamic>, column-major
  int r, c;
  std::vector<short> d;
  MaskX() : r(0), c(0) {}
  MaskX(int rows, int cols) : r(rows), c(cols), d((size_t)rows * cols, 0) {}
  short& operator()(int i, int j) { return d[(size_t)j * r + i]; }
  short operator()(int i, int j) const { return d[(size_t)j * r + i]; }
  void set_zero() { std::fill(d.begin(), d.end(), (short)0); }
  long sum() const { long s = 0; for (short v : d) s += v; return s; }
};

// ------------------------------------------------------------------ adapters (T1)
// pose/PoseAdapterBase.hpp:28-146
template <class T> class PoseAdapterBase {
 public:
  typedef V3<T> Point3;
  PoseAdapterBase() : _t_w(), _R_cw(), _fx(0), _fy(0), _cx(0), _cy(0) {}
  virtual ~PoseAdapterBase() {}
  virtual Point3 getBearingVector(int index) const = 0;
  virtual T getWeight(int index) const = 0;
  virtual Point3 getPointGlob(int index) const = 0;
  virtual int getNumberCorrespondences() const = 0;
  Point3 gettw() const { return _t_w; }
  void sett(const Point3& t) { _t_w = t; }
  SO3<T> getRcw() const { return _R_cw; }
  void setRcw(const SO3<T>& R) { _R_cw = R; }
  void setFocal(T fx, T fy) { _fx = fx; _fy = fy; }
  T getFocal() const { return (_fx + _fy) / 2; }
  SE3<T> getTcw() const { return SE3<T>(_R_cw, _t_w); }
 protected:
  Point3 _t_w;
  SO3<T> _R_cw;
  T _fx, _fy, _cx, _cy;
};

// pose/Utility.hpp:107-118 (descending).  The reference's comparator (v[a] > v[b]) leaves the order of EQUAL weights to
// std::sort, i.e. unspecified; ties go to the lower index here so that the order is unique and comparable across builds.
template <class T> std::vector<int> sortIndexes(const std::vector<T>& v) {
  std::vector<int> idx(v.size());
  std::iota(idx.begin(), idx.end(), 0);
  std::sort(idx.begin(), idx.end(), [&v](int a, int b) { return v[a] > v[b] || (v[a] == v[b] && a < b); });
  return idx;
}

// pose/AOOnlyPoseAdapter.hpp:26-255
template <class T> class AOOnlyPoseAdapter : public PoseAdapterBase<T> {
 public:
  typedef V3<T> Point3;
  AOOnlyPoseAdapter(const MatX<T>& points_c, const MatX<T>& points_g)
      : _points_c(points_c), _points_g(points_g), _inliers_3d(points_c.cols(), 1), _max_votes(0) {}
  virtual Point3 getBearingVector(int) const { return Point3(); }
  virtual Point3 getPointCurr(int i) const { return _points_c.col3(i); }
  virtual Point3 getPointGlob(int i) const { return _points_g.col3(i); }
  virtual T getWeight(int) const { return T(1.); }
  virtual int getNumberCorrespondences() const { return _points_g.cols(); }
  virtual bool isValid(int i) const {  // :147-152  (|| : invalid only if ALL THREE are NaN)
    Point3 p = _points_c.col3(i);
    return p.x == p.x || p.y == p.y || p.z == p.z;
  }
  bool isInlier33(int i) const { return _inliers_3d[i] == 1; }
  T weight33(int i) const { return _weights_3d.empty() ? T(1.0) : _weights_3d[i]; }  // :175-183 (no /SHRT_MAX)
  void setMaxVotes(int v) { _max_votes = v; }
  int getMaxVotes() const { return _max_votes; }
  virtual void setInlier(const MaskX& m) {  // :185-198 : only a >=2-column mask is consumed, column 1
    if (m.c != 1) for (int i = 0; i < m.r; i++) _inliers_3d[i] = m(i, 1);
  }
  virtual void setWeights(const MatX<T>& w) {  // :200-212 (tests rows()==1, then takes column 1)
    if (w.rows() != 1) { _weights_3d.resize(w.rows()); for (int i = 0; i < w.rows(); i++) _weights_3d[i] = w(i, 1); }
  }
  const std::vector<int>& getInlierIdx() const { return _vInliersAO; }
  void cvtInlier() {  // :219-231 (D1)
    _vInliersAO.clear();
    for (int r = 0; r < (int)_inliers_3d.size(); r++) if (1 == _inliers_3d[r]) _vInliersAO.push_back(r);
  }
  void sortIdx() { _idx = sortIndexes<T>(_weights_3d); }  // :233-243
  void getSortedIdx(std::vector<int>& sel) const {        // :245-254
    for (size_t i = 0; i < sel.size(); i++) { int j = sel[i]; if (j < (int)_idx.size()) sel[i] = _idx[j]; }
  }
  const MaskCol& mask33() const { return _inliers_3d; }
 protected:
  const MatX<T>& _points_c;
  const MatX<T>& _points_g;
  MaskCol _inliers_3d;
  std::vector<T> _weights_3d;
  std::vector<int> _idx;
  std::vector<int> _vInliersAO;
  int _max_votes;
};

// pose/PnPPoseAdapter.hpp:27-255
template <class T> class PnPPoseAdapter : public PoseAdapterBase<T> {
 public:
  typedef V3<T> Point3;
  PnPPoseAdapter(const MatX<T>& bearingVectors, const MatX<T>& points)
      : _bearingVectors(bearingVectors), _points_g(points), _inliers(bearingVectors.cols(), 1), _max_votes(0) {}
  virtual Point3 getBearingVector(int i) const { return _bearingVectors.col3(i); }
  virtual T getWeight(int) const { return T(1.); }
  virtual Point3 getPointGlob(int i) const { return _points_g.col3(i); }
  virtual int getNumberCorrespondences() const { return _bearingVectors.cols(); }
  virtual void setInlier(const MaskX& m) { for (int i = 0; i < m.r; i++) _inliers[i] = m(i, 0); }  // :196-202 memcpy of column 0
  virtual void setWeights(const MatX<T>& w) { _weights.resize(w.rows()); for (int i = 0; i < w.rows(); i++) _weights[i] = w(i, 0); }
  const std::vector<int>& getInlierIdx() const { return _vInliersPnP; }
  void cvtInlier() {  // :224-237 (D1)
    _vInliersPnP.clear();
    for (int r = 0; r < (int)_inliers.size(); r++) if (1 == _inliers[r]) _vInliersPnP.push_back(r);
  }
  T getError(int i) const {  // :204-210
    Point3 Xc = this->_R_cw * getPointGlob(i) + this->_t_w;
    Xc = normalized(Xc);
    return norm(cross(Xc, getBearingVector(i)));
  }
  void setMaxVotes(int v) { _max_votes = v; }
  int getMaxVotes() const { return _max_votes; }
  bool isInlier23(int i) const { return _inliers[i] == 1; }
  T weight23(int i) const { return _weights.empty() ? T(1.0) : _weights[i]; }  // :180-188
  void sortIdx() { _idx = sortIndexes<T>(_weights); }
  void getSortedIdx(std::vector<int>& sel) const {
    for (size_t i = 0; i < sel.size(); i++) { int j = sel[i]; if (j < (int)_idx.size()) sel[i] = _idx[j]; }
  }
  const MaskCol& mask23() const { return _inliers; }
 protected:
  const MatX<T>& _bearingVectors;
  const MatX<T>& _points_g;
  MaskCol _inliers;
  std::vector<T> _weights;
  std::vector<int> _idx;
  std::vector<int> _vInliersPnP;
  int _max_votes;
};

// pose/AOPoseAdapter.hpp:26-217
template <class T> class AOPoseAdapter : public PnPPoseAdapter<T> {
 public:
  typedef V3<T> Point3;
  AOPoseAdapter(const MatX<T>& bearingVectors, const MatX<T>& points_c, const MatX<T>& points_g)
      : PnPPoseAdapter<T>(bearingVectors, points_g), _points_c(points_c), _inliers_3d(bearingVectors.cols(), 1) {}
  virtual Point3 getPointCurr(int i) const { return _points_c.col3(i); }
  virtual bool isValid(int i) const { Point3 p = _points_c.col3(i); return p.x == p.x || p.y == p.y || p.z == p.z; }
  bool isInlier33(int i) const { return _inliers_3d[i] == 1; }
  T weight33(int i) const {  // :161-169 : divides by SHRT_MAX
    return _weights_3d.empty() ? T(1.0) : T(_weights_3d[i]) / std::numeric_limits<short>::max();
  }
  virtual void setInlier(const MaskX& m) {  // :171-184
    PnPPoseAdapter<T>::setInlier(m);
    if (m.c != 1) for (int i = 0; i < m.r; i++) _inliers_3d[i] = m(i, 1);
  }
  virtual void setWeights(const MatX<T>& w) {  // :186-199
    PnPPoseAdapter<T>::setWeights(w);
    if (w.rows() != 1) { _weights_3d.resize(w.rows()); for (int i = 0; i < w.rows(); i++) _weights_3d[i] = w(i, 1); }
  }
  const std::vector<int>& getInlierIdx() const { return _vInliersAO; }
  void cvtInlier() {
    _vInliersAO.clear();
    for (int r = 0; r < (int)_inliers_3d.size(); r++) if (1 == _inliers_3d[r]) _vInliersAO.push_back(r);
  }
  const MaskCol& mask33() const { return _inliers_3d; }
 protected:
  const MatX<T>& _points_c;
  MaskCol _inliers_3d;
  std::vector<T> _weights_3d;
  std::vector<int> _vInliersAO;
};

// pose/NormalAOPoseAdapter.hpp:16-231
template <class T> class NormalAOPoseAdapter : public AOPoseAdapter<T> {
 public:
  typedef V3<T> Point3;
  NormalAOPoseAdapter(const MatX<T>& bearingVectors, const MatX<T>& points_c, const MatX<T>& normal_c,
                      const MatX<T>& points_g, const MatX<T>& normal_g)
      : AOPoseAdapter<T>(bearingVectors, points_c, points_g), _normal_c(normal_c), _normal_g(normal_g),
        _inliers_nl(bearingVectors.cols(), 1) {}
  bool isInlierNN(int i) const { return _inliers_nl[i] == 1; }
  T weightNN(int i) const {  // :153-161
    return _weights_nl.empty() ? T(1.0) : T(_weights_nl[i]) / std::numeric_limits<short>::max();
  }
  virtual Point3 getNormalCurr(int i) const { return _normal_c.col3(i); }
  virtual Point3 getNormalGlob(int i) const { return _normal_g.col3(i); }
  virtual void setInlier(const MaskX& m) {  // :179-195
    if (m.c == 1) PnPPoseAdapter<T>::setInlier(m);
    if (m.c == 2) AOPoseAdapter<T>::setInlier(m);
    if (m.c == 3) { AOPoseAdapter<T>::setInlier(m); for (int i = 0; i < m.r; i++) _inliers_nl[i] = m(i, 2); }
  }
  virtual void setWeights(const MatX<T>& w) {  // :197-212 (tests cols(), unlike the AO adapters)
    if (w.cols() == 1) PnPPoseAdapter<T>::setWeights(w);
    if (w.cols() == 2) AOPoseAdapter<T>::setWeights(w);
    if (w.cols() == 3) {
      AOPoseAdapter<T>::setWeights(w);
      _weights_nl.resize(w.rows());
      for (int i = 0; i < w.rows(); i++) _weights_nl[i] = w(i, 2);
    }
  }
  const std::vector<int>& getInlierIdx() const { return _vInliersNN; }
  void cvtInlier() {
    _vInliersNN.clear();
    for (int r = 0; r < (int)_inliers_nl.size(); r++) if (1 == _inliers_nl[r]) _vInliersNN.push_back(r);
  }
  const MaskCol& maskNN() const { return _inliers_nl; }
 protected:
  const MatX<T>& _normal_c;
  const MatX<T>& _normal_g;
  MaskCol _inliers_nl;
  std::vector<T> _weights_nl;
  std::vector<int> _vInliersNN;
};

// ------------------------------------------------------------------ samplers (pose/Utility.hpp:120-250)
class RandomElements {  // :124-156
 public:
  explicit RandomElements(int n) : _idx(n), _n(n) {}
  void run(int m, std::vector<int>* out, Rand31& rnd) {
    out->clear();
    for (int i = 0; i < _n; i++) _idx[i] = i;
    for (int j = _n - 1; j > _n - m - 1; j--) {
      int ridx = rnd() % (j + 1);
      int temp = _idx[ridx];
      _idx[ridx] = _idx[j];
      _idx[j] = temp;
      out->push_back(temp);
    }
  }
 private:
  std::vector<int> _idx;
  int _n;
};

template <class T> class ProsacSampler {  // :161-250 (Chum & Matas growth function, T_N = 20000)
 public:
  ProsacSampler(int min_num_samples, int num_datapoints) : _N(num_datapoints), _T_N(20000), _t(1), _m(min_num_samples) {}
  void setSampleNumber(int k) { _t = k; }
  bool sample(std::vector<int>* subset, Rand31& rnd) {
    T t_n = (T)_T_N;
    int n = _m;
    for (int i = 0; i < _m; i++) t_n *= static_cast<T>(n - i) / (_N - i);
    T t_n_prime = 1.0;
    for (int t = 1; t <= _t; t++) {
      if (t > t_n_prime && n < _N) {
        T t_n_plus1 = (t_n * (n + 1.0)) / (n + 1.0 - _m);
        t_n_prime += std::ceil(t_n_plus1 - t_n);
        t_n = t_n_plus1;
        n++;
      }
    }
    subset->clear();
    std::vector<int> used;
    if (t_n_prime < _t) {
      for (int i = 0; i < _m; i++) {
        int r;
        while (std::find(used.begin(), used.end(), (r = rnd() % n)) != used.end()) {}
        used.push_back(r);
        subset->push_back(r);
      }
    } else {
      for (int i = 0; i < _m - 1; i++) {
        int r;
        while (std::find(used.begin(), used.end(), (r = rnd() % (n - 1))) != used.end()) {}
        used.push_back(r);
        subset->push_back(r);
      }
      subset->push_back(n < _N ? n : _N - 1);  // D3
    }
    _t++;
    return true;
  }
 private:
  int _N, _T_N, _t, _m;
};

// ------------------------------------------------------------------ U1  pose/P3P.hpp:296-318
template <class T> int RANSACUpdateNumIters(T p, T ep, const int modelPoints, const int maxIters) {
  p = std::max(p, T(0.)); p = std::min(p, T(1.));
  ep = std::max(ep, T(0.)); ep = std::min(ep, T(1.));
  T num = std::max(T(1. - p), std::numeric_limits<T>::epsilon());
  T denom = T(1.) - std::pow(T(1. - ep), modelPoints);
  if (denom < std::numeric_limits<T>::epsilon()) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : int(num / denom + 0.5f);
}

// ------------------------------------------------------------------ E1  pose/AbsoluteOrientation.hpp:11-43
template <class T> void calc_percentage_err(const SO3<T>& R_cw, const V3<T>& t_w, const PoseAdapterBase<T>* ad, T out[2]) {
  V3<T> te = R_cw * t_w - ad->getRcw() * ad->gettw();
  out[0] = norm(te) / norm(ad->gettw()) * 100;
  Quat<T> a = R_cw.q, b = ad->getRcw().q;
  T dw = a.w - b.w, dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  out[1] = std::sqrt(dw * dw + dx * dx + dy * dy + dz * dz) / std::sqrt(b.sqnorm()) * 100;
}
template <class T> void calc_err(const M3<T>& Rgt, const V3<T>& tgt, const M3<T>& Rse, const V3<T>& tse, T out[2]) {
  // Diff = SE * GT^-1 ; t_e = |Diff.t| ; r_e = angle(Diff.R)   (:29-43)
  M3<T> Rd = Rse * transpose(Rgt);
  V3<T> td = tse - Rd * tgt;
  out[0] = norm(td);
  out[1] = rotation_angle(Rd);
}

// ------------------------------------------------------------------ A1  shinji  pose/AbsoluteOrientation.hpp:47-99
template <class T> SE3<T> shinji(const MatX<T>& X_w, const MatX<T>& X_c, int K) {
  V3<T> Cw, Cc;
  for (int n = 0; n < K; n++) { Cw = Cw + X_w.col3(n); Cc = Cc + X_c.col3(n); }
  Cw = Cw / (T)K; Cc = Cc / (T)K;
  M3<T> M;
  T sigma_w = 0, sigma_c = 0;
  for (int n = 0; n < K; n++) {
    V3<T> Aw = X_w.col3(n) - Cw; sigma_w += norm(Aw);
    V3<T> Ac = X_c.col3(n) - Cc; sigma_c += norm(Ac);
    M = M + outer(Ac, Aw);
  }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M(i, j) = M(i, j) / (T)X_w.cols();  // :75 'M /= (Tp) X_w_.cols()': cols(), not K; Eigen 3.3 divides
  (void)sigma_w; (void)sigma_c;
  SVD3<T> d = svd3(M);
  M3<T> Tmp = d.U * transpose(d.V);
  SO3<T> R;
  if (det(Tmp) < T(0)) {
    M3<T> I = M3<T>::identity(); I(2, 2) = -1;
    R = SO3<T>(d.U * I * transpose(d.V));
  } else {
    R = SO3<T>(Tmp);
  }
  V3<T> t = Cc - R * Cw;
  return SE3<T>(R, t);
}

// A2/A3  :273-342
template <class T, class Ad> void shinji_ls_inliers(Ad& adapter) {  // shinji_ls (AOPoseAdapter) / shinji_ls1 (AOOnly)
  const std::vector<int>& v = adapter.getInlierIdx();
  int K = (int)v.size();
  MatX<T> Xw(3, K), Xc(3, K);
  for (int i = 0; i < K; i++) { int idx = v[i]; Xw.set_col3(i, adapter.getPointGlob(idx)); Xc.set_col3(i, adapter.getPointCurr(idx)); }
  SE3<T> s = shinji<T>(Xw, Xc, K);
  adapter.setRcw(s.R); adapter.sett(s.t);
}
template <class T> void shinji_ls(AOPoseAdapter<T>& a) { shinji_ls_inliers<T>(a); }
template <class T> void shinji_ls1(AOOnlyPoseAdapter<T>& a) { shinji_ls_inliers<T>(a); }
template <class T> void shinji_ls2(AOOnlyPoseAdapter<T>& adapter) {
  int K = adapter.getNumberCorrespondences();
  MatX<T> Xw(3, K), Xc(3, K);
  for (int i = 0; i < K; i++) { Xw.set_col3(i, adapter.getPointGlob(i)); Xc.set_col3(i, adapter.getPointCurr(i)); }
  SE3<T> s = shinji<T>(Xw, Xc, K);
  adapter.setRcw(s.R); adapter.sett(s.t);
}

// ------------------------------------------------------------------ vote loops V1..V8
// One function per modality combination; bodies are the reference loop bodies verbatim in meaning.
// 3D-3D only: V1/V2/V3  AbsoluteOrientation.hpp:133-143,190-200,248-258   (mask N x 2, column 1)
template <class T, class Ad> int vote_33(const Ad& ad, const SE3<T>& s, T thre_3d, MaskX& inl) {
  int votes = 0;
  inl.set_zero();
  const int N = ad.getNumberCorrespondences();
  for (int c = 0; c < N; c++) {
    if (ad.isValid(c)) {
      V3<T> e = ad.getPointCurr(c) - (s.R * ad.getPointGlob(c) + s.t);
      if (norm(e) < thre_3d) { inl(c, 1) = 1; votes++; }
    }
  }
  return votes;
}
// 2D-3D only: V5  P3P.hpp:362-376 (ransac: matrix product) / :439-453 (prosac: quaternion product)
template <class T> int vote_23(const PnPPoseAdapter<T>& ad, const SE3<T>& s, T cos_thr, MaskX& inl, bool use_matrix) {
  int votes = 0;
  inl.set_zero();
  const int N = ad.getNumberCorrespondences();
  M3<T> Rm = s.R.matrix();
  for (int i = 0; i < N; i++) {
    V3<T> Xw = ad.getPointGlob(i);
    V3<T> Xc = (use_matrix ? Rm * Xw : s.R * Xw) + s.t;
    Xc = Xc / norm(Xc);
    T cos_a = dot(Xc, ad.getBearingVector(i));
    if (cos_a > cos_thr) { inl(i, 0) = 1; votes++; }
  }
  return votes;
}
// 3D-3D + 2D-3D: V4  AbsoluteOrientation.hpp:403-422,480-499  (mask N x 2)
template <class T> int vote_33_23(const AOPoseAdapter<T>& ad, const SE3<T>& s, T thre_3d, T cos_thr, MaskX& inl) {
  int votes = 0;
  inl.set_zero();
  const int N = ad.getNumberCorrespondences();
  for (int c = 0; c < N; c++) {
    if (ad.isValid(c)) {
      V3<T> e = ad.getPointCurr(c) - (s.R * ad.getPointGlob(c) + s.t);
      if (norm(e) < thre_3d) { inl(c, 1) = 1; votes++; }
    }
    V3<T> pc = s.R * ad.getPointGlob(c) + s.t;
    pc = pc / norm(pc);
    T cos_a = dot(pc, ad.getBearingVector(c));
    if (cos_a > cos_thr) { inl(c, 0) = 1; votes++; }
  }
  return votes;
}
// N-N + 2D-3D: V6  AbsoluteOrientationNormal.hpp:245-264  (mask N x 3)
template <class T> int vote_nn_23(const NormalAOPoseAdapter<T>& ad, const SE3<T>& s, T cos_thr, T cos_nl, MaskX& inl) {
  int votes = 0;
  inl.set_zero();
  const int N = ad.getNumberCorrespondences();
  for (int c = 0; c < N; c++) {
    if (ad.isValid(c)) {
      T ca = dot(ad.getNormalCurr(c), s.R * ad.getNormalGlob(c));
      if (ca > cos_nl) { inl(c, 2) = 1; votes++; }
    }
    V3<T> pc = s.R * ad.getPointGlob(c) + s.t;
    pc = pc / norm(pc);
    T cos_a = dot(pc, ad.getBearingVector(c));
    if (cos_a > cos_thr) { inl(c, 0) = 1; votes++; }
  }
  return votes;
}
// N-N + 3D-3D: V7  AbsoluteOrientationNormal.hpp:322-337
template <class T> int vote_nn_33(const NormalAOPoseAdapter<T>& ad, const SE3<T>& s, T thre_3d, T cos_nl, MaskX& inl) {
  int votes = 0;
  inl.set_zero();
  const int N = ad.getNumberCorrespondences();
  for (int c = 0; c < N; c++) {
    if (ad.isValid(c)) {
      T ca = dot(ad.getNormalCurr(c), s.R * ad.getNormalGlob(c));
      if (ca > cos_nl) { inl(c, 2) = 1; votes++; }
      V3<T> e = ad.getPointCurr(c) - (s.R * ad.getPointGlob(c) + s.t);
      if (norm(e) < thre_3d) { inl(c, 1) = 1; votes++; }
    }
  }
  return votes;
}
// all three: V8  AbsoluteOrientationNormal.hpp:397-423
template <class T> int vote_nn_33_23(const NormalAOPoseAdapter<T>& ad, const SE3<T>& s, T thre_3d, T cos_thr, T cos_nl, MaskX& inl) {
  int votes = 0;
  inl.set_zero();
  const int N = ad.getNumberCorrespondences();
  for (int c = 0; c < N; c++) {
    if (ad.isValid(c)) {
      T ca = dot(ad.getNormalCurr(c), s.R * ad.getNormalGlob(c));
      if (ca > cos_nl) { inl(c, 2) = 1; votes++; }
      V3<T> e = ad.getPointCurr(c) - (s.R * ad.getPointGlob(c) + s.t);
      if (norm(e) < thre_3d) { inl(c, 1) = 1; votes++; }
    }
    V3<T> pc = s.R * ad.getPointGlob(c) + s.t;
    pc = pc / norm(pc);
    T cos_a = dot(pc, ad.getBearingVector(c));
    if (cos_a > cos_thr) { inl(c, 0) = 1; votes++; }
  }
  return votes;
}

// ------------------------------------------------------------------ P3P  pose/P3P.hpp:11-294
// Ferrari's closed form with complex intermediates (:11-60). p = (A,B,C,D,E), A x^4 + ... + E.
template <class T> void o4_roots(const T p[5], T roots[4]) {
  typedef std::complex<T> C;
  const T A = p[0], B = p[1], Cc = p[2], D = p[3], E = p[4];
  const T A2 = A * A, B2 = B * B, A3 = A2 * A, B3 = B2 * B, A4 = A3 * A, B4 = B3 * B;
  const T alpha = -3 * B2 / (8 * A2) + Cc / A;
  const T beta = B3 / (8 * A3) - B * Cc / (2 * A2) + D / A;
  const T gamma = -3 * B4 / (256 * A4) + B2 * Cc / (16 * A3) - B * D / (4 * A2) + E / A;
  const T al2 = alpha * alpha, al3 = al2 * alpha;
  C P(-al2 / 12 - gamma, 0);
  C Q(-al3 / 108 + alpha * gamma / 3 - std::pow(beta, 2) / 8, 0);
  C R = -Q / T(2.0) + std::sqrt(std::pow(Q, T(2.)) / T(4.) + std::pow(P, T(3.)) / T(27.));
  C U = std::pow(R, T(1.0 / 3.0));
  C y;
  if (U.real() == 0) y = -T(5.0) * alpha / T(6.) - std::pow(Q, T(1.0 / 3.0));
  else y = -T(5.0) * alpha / T(6.) - P / (T(3.) * U) + U;
  C w = std::sqrt(alpha + T(2.) * y);
  C s1 = std::sqrt(-(T(3.) * alpha + T(2.) * y + T(2.) * beta / w));
  C s2 = std::sqrt(-(T(3.) * alpha + T(2.) * y - T(2.) * beta / w));
  const T sh = -B / (T(4.) * A);
  roots[0] = (sh + T(0.5) * (w + s1)).real();
  roots[1] = (sh + T(0.5) * (w - s1)).real();
  roots[2] = (sh + T(0.5) * (-w + s2)).real();
  roots[3] = (sh + T(0.5) * (-w - s2)).real();
}

// Kneip, Scaramuzza, Siegwart (CVPR 2011) P3P.  X_w, bv: 3 x >=3.  (:63-232)
template <class T> void kneip_main(const MatX<T>& X_w, const MatX<T>& bv, std::vector<SE3<T> >* sols) {
  sols->clear();
  V3<T> P1 = X_w.col3(0), P2 = X_w.col3(1), P3 = X_w.col3(2);
  V3<T> temp1 = P2 - P1, temp2 = P3 - P1;
  if (norm(cross(temp1, temp2)) == 0) return;
  V3<T> f1 = bv.col3(0), f2 = bv.col3(1), f3 = bv.col3(2);
  V3<T> e1 = f1, e3 = cross(f1, f2);
  e3 = e3 / norm(e3);
  V3<T> e2 = cross(e3, e1);
  M3<T> RR;
  RR.set_row(0, e1); RR.set_row(1, e2); RR.set_row(2, e3);
  f3 = RR * f3;
  if (f3.z > 0) {
    f1 = bv.col3(1); f2 = bv.col3(0); f3 = bv.col3(2);
    e1 = f1; e3 = cross(f1, f2); e3 = e3 / norm(e3); e2 = cross(e3, e1);
    RR.set_row(0, e1); RR.set_row(1, e2); RR.set_row(2, e3);
    f3 = RR * f3;
    P1 = X_w.col3(1); P2 = X_w.col3(0); P3 = X_w.col3(2);
  }
  V3<T> n1 = P2 - P1; n1 = n1 / norm(n1);
  V3<T> n3 = cross(n1, P3 - P1); n3 = n3 / norm(n3);
  V3<T> n2 = cross(n3, n1);
  M3<T> N;
  N.set_row(0, n1); N.set_row(1, n2); N.set_row(2, n3);
  P3 = N * (P3 - P1);
  // NOTE :129 uses temp1 = X_w.col(1) - X_w.col(0) computed BEFORE the swap; its norm is swap-invariant.
  const T d12 = norm(temp1);
  const T f_1 = f3.x / f3.z, f_2 = f3.y / f3.z, p_1 = P3.x, p_2 = P3.y;
  const T cos_beta = dot(f1, f2);
  T b = 1 / (1 - std::pow(cos_beta, 2)) - 1;
  b = cos_beta < 0 ? -std::sqrt(b) : std::sqrt(b);
  const T f1s = f_1 * f_1, f2s = f_2 * f_2, p1s = p_1 * p_1, p1c = p1s * p_1, p1q = p1c * p_1;
  const T p2s = p_2 * p_2, p2c = p2s * p_2, p2q = p2c * p_2, d2 = d12 * d12, b2 = b * b;
  T fac[5];
  fac[0] = -f2s * p2q - p2q * f1s - p2q;
  fac[1] = 2 * p2c * d12 * b + 2 * f2s * p2c * d12 * b - 2 * f_2 * p2c * f_1 * d12;
  fac[2] = -f2s * p2s * p1s - f2s * p2s * d2 * b2 - f2s * p2s * d2 + f2s * p2q + p2q * f1s + 2 * p_1 * p2s * d12 +
           2 * f_1 * f_2 * p_1 * p2s * d12 * b - p2s * p1s * f1s + 2 * p_1 * p2s * f2s * d12 - p2s * d2 * b2 - 2 * p1s * p2s;
  fac[3] = 2 * p1s * p_2 * d12 * b + 2 * f_2 * p2c * f_1 * d12 - 2 * f2s * p2c * d12 * b - 2 * p_1 * p_2 * d2 * b;
  fac[4] = -2 * f_2 * p2s * f_1 * p_1 * d12 * b + f2s * p2s * d2 + 2 * p1c * d12 - p1s * d2 + f2s * p2s * p1s - p1q -
           2 * f2s * p2s * p_1 * d12 + p2s * f1s * p1s + f2s * p2s * d2 * b2;
  T roots[4];
  o4_roots<T>(fac, roots);
  for (int i = 0; i < 4; i++) {
    if (roots[i] != roots[i]) continue;
    T cot_alpha = (-f_1 * p_1 / f_2 - roots[i] * p_2 + d12 * b) / (-f_1 * roots[i] * p_2 / f_2 + p_1 - d12);
    T cos_theta = roots[i];
    if (cos_theta > T(1) || cos_theta < T(-1)) continue;
    T sin_theta = std::sqrt(1 - roots[i] * roots[i]);
    T sin_alpha = std::sqrt(1 / (cot_alpha * cot_alpha + 1));
    T cos_alpha = std::sqrt(1 - sin_alpha * sin_alpha);
    if (cot_alpha < 0) cos_alpha = -cos_alpha;
    V3<T> C(d12 * cos_alpha * (sin_alpha * b + cos_alpha), cos_theta * d12 * sin_alpha * (sin_alpha * b + cos_alpha),
            sin_theta * d12 * sin_alpha * (sin_alpha * b + cos_alpha));
    C = P1 + transpose(N) * C;
    M3<T> R;
    R(0, 0) = -cos_alpha; R(0, 1) = -sin_alpha * cos_theta; R(0, 2) = -sin_alpha * sin_theta;
    R(1, 0) = sin_alpha;  R(1, 1) = -cos_alpha * cos_theta; R(1, 2) = -cos_alpha * sin_theta;
    R(2, 0) = 0.0;        R(2, 1) = -sin_theta;             R(2, 2) = cos_theta;
    R = transpose(RR) * R * N;
    if (R(0, 0) != R(0, 0)) continue;
    SO3<T> so3(R);
    if (!so3.ok) continue;  // D4
    sols->push_back(SE3<T>(so3, -(R * C)));
  }
}

// disambiguate with the 4th point (:250-294): returns false if no solution
template <class T> bool kneip(const MatX<T>& X_w, const MatX<T>& bv, SE3<T>* sol) {
  std::vector<SE3<T> > v;
  kneip_main<T>(X_w, bv, &v);
  T minScore = std::numeric_limits<T>::max();
  int minIndex = -1;
  for (int i = 0; i < (int)v.size(); i++) {
    V3<T> pc = v[i].R.matrix() * X_w.col3(3) + v[i].t;
    pc = pc / norm(pc);
    T score = T(1.0) - dot(pc, bv.col3(3));
    if (score < minScore) { minScore = score; minIndex = i; }
  }
  if (minIndex != -1) { *sol = v[minIndex]; return true; }
  return false;
}

// ------------------------------------------------------------------ explicit hypothesis streams (test plumbing, not in the reference)
// "both CPU restatement and GPU consume the same sample list".  With capture set, the drivers below only
// generate: the hypotheses of each of the `Iter` iterations are appended (qw qx qy qz tx ty tz) and nothing is voted on.  With
// replay set, an iteration's hypotheses come from the given list instead of the sampler + minimal solvers; everything after that
// (vote loop, strict '>', adaptive Iter, masks) is the reference's.
struct HypList { std::vector<double> q7; std::vector<int> first; };   // first[i] .. first[i+1]: hypotheses of iteration i
inline HypList*& capture_sink() { static HypList* p = nullptr; return p; }
inline const HypList*& replay_source() { static const HypList* p = nullptr; return p; }
template <class T> bool hyp_replay(int ii, std::vector<SE3<T> >* sols) {
  const HypList* in = replay_source();
  if (!in) return false;
  sols->clear();
  if (ii + 1 < (int)in->first.size())
    for (int h = in->first[ii]; h < in->first[ii + 1]; h++) {
      const double* q = &in->q7[7 * (size_t)h];
      SE3<T> s; s.R.q = Quat<T>((T)q[0], (T)q[1], (T)q[2], (T)q[3]); s.t = V3<T>((T)q[4], (T)q[5], (T)q[6]);
      sols->push_back(s);
    }
  return true;
}
template <class T> bool hyp_capture(const std::vector<SE3<T> >& sols) {
  HypList* out = capture_sink();
  if (!out) return false;
  if (out->first.empty()) out->first.push_back(0);
  for (const SE3<T>& s : sols) {
    const double q[7] = {(double)s.R.q.w, (double)s.R.q.x, (double)s.R.q.y, (double)s.R.q.z, (double)s.t.x, (double)s.t.y, (double)s.t.z};
    out->q7.insert(out->q7.end(), q, q + 7);
  }
  out->first.push_back((int)(out->q7.size() / 7));
  return true;
}

// ------------------------------------------------------------------ RANSAC drivers
// shinji_ransac (AOPoseAdapter) :101-156, shinji_ransac2 (AOOnly) :158-213
template <class T, class Ad> void shinji_ransac_impl(Ad& adapter, const T thre_3d, int& Iter, T confidence, Rand31& rnd) {
  const int N = adapter.getNumberCorrespondences();
  RandomElements re(N);
  const int K = 3;
  adapter.setMaxVotes(-1);
  MaskX inliers(N, 2);
  for (int ii = 0; ii < Iter; ii++) {
    std::vector<SE3<T> > sols;
    if (!hyp_replay<T>(ii, &sols)) {
      std::vector<int> sel;
      re.run(K, &sel, rnd);
      MatX<T> Xw(3, K), Xc(3, K);
      bool invalid = false;
      for (int s = 0; s < K; s++) {
        Xw.set_col3(s, adapter.getPointGlob(sel[s]));
        if (adapter.isValid(sel[s])) Xc.set_col3(s, adapter.getPointCurr(sel[s]));
        else invalid = true;
      }
      if (!invalid) { SE3<T> fit = shinji<T>(Xw, Xc, K); if (fit.R.ok) sols.push_back(fit); }  // D4
    }
    if (hyp_capture<T>(sols) || sols.empty()) continue;
    const SE3<T>& sol = sols[0];
    int votes = vote_33<T>(adapter, sol, thre_3d, inliers);
    if (votes > adapter.getMaxVotes()) {
      adapter.setMaxVotes(votes);
      adapter.setRcw(sol.R); adapter.sett(sol.t);
      adapter.setInlier(inliers);
      Iter = RANSACUpdateNumIters(confidence, (T)(N - votes) / N, K, Iter);
    }
  }
  adapter.cvtInlier();
}
template <class T> void shinji_ransac(AOPoseAdapter<T>& a, const T thr, int& Iter, T conf, Rand31& rnd) { shinji_ransac_impl<T>(a, thr, Iter, conf, rnd); }
template <class T> void shinji_ransac2(AOOnlyPoseAdapter<T>& a, const T thr, int& Iter, T conf, Rand31& rnd) { shinji_ransac_impl<T>(a, thr, Iter, conf, rnd); }

// shinji_prosac :215-271
template <class T> void shinji_prosac(AOOnlyPoseAdapter<T>& adapter, const T thre_3d, int& Iter, T confidence, Rand31& rnd) {
  const int N = adapter.getNumberCorrespondences();
  adapter.sortIdx();
  const int K = 3;
  ProsacSampler<T> ps(K, N);
  adapter.setMaxVotes(-1);
  MaskX inliers(N, 2);
  for (int ii = 0; ii < Iter; ii++) {
    std::vector<SE3<T> > sols;
    if (!hyp_replay<T>(ii, &sols)) {
      std::vector<int> sel;
      ps.sample(&sel, rnd);
      adapter.getSortedIdx(sel);
      MatX<T> Xw(3, K), Xc(3, K);
      bool invalid = false;
      for (int s = 0; s < K; s++) {
        Xw.set_col3(s, adapter.getPointGlob(sel[s]));
        if (adapter.isValid(sel[s])) Xc.set_col3(s, adapter.getPointCurr(sel[s]));
        else invalid = true;
      }
      if (!invalid) { SE3<T> fit = shinji<T>(Xw, Xc, K); if (fit.R.ok) sols.push_back(fit); }
    }
    if (hyp_capture<T>(sols) || sols.empty()) continue;
    const SE3<T>& sol = sols[0];
    int votes = vote_33<T>(adapter, sol, thre_3d, inliers);
    if (votes > adapter.getMaxVotes()) {
      adapter.setMaxVotes(votes);
      adapter.setRcw(sol.R); adapter.sett(sol.t);
      adapter.setInlier(inliers);
      Iter = RANSACUpdateNumIters(confidence, (T)(N - votes) / N, K, Iter);
    }
  }
  adapter.cvtInlier();
}

// kneip_ransac / kneip_prosac  pose/P3P.hpp:320-469   (K = 4 goes to RANSACUpdateNumIters)
template <class T> void kneip_ransac_impl(PnPPoseAdapter<T>& adapter, const T thre_2d, int& Iter, T confidence, Rand31& rnd, bool prosac) {
  const T cos_thr = std::cos(std::atan(thre_2d / adapter.getFocal()));
  const int N = adapter.getNumberCorrespondences();
  if (prosac) adapter.sortIdx();
  RandomElements re(N);
  const int K = 4;
  ProsacSampler<T> ps(K, N);
  adapter.setMaxVotes(-1);
  MaskX inliers(N, 1);
  for (int it = 0; it < Iter; it++) {
    std::vector<SE3<T> > picked;   // the iteration's hypothesis: the P3P branch that best reprojects the 4th point
    if (!hyp_replay<T>(it, &picked)) {
      std::vector<int> sel;
      if (prosac) { ps.sample(&sel, rnd); adapter.getSortedIdx(sel); }
      else re.run(K, &sel, rnd);
      MatX<T> bv(3, 3), Xw(3, 3);
      for (int k = 0; k < 3; k++) { bv.set_col3(k, adapter.getBearingVector(sel[k])); Xw.set_col3(k, adapter.getPointGlob(sel[k])); }
      std::vector<SE3<T> > sols;
      kneip_main<T>(Xw, bv, &sols);
      T minScore = 1000000.0;
      int minIndex = -1;
      for (int i = 0; i < (int)sols.size(); i++) {
        V3<T> pw = adapter.getPointGlob(sel[3]);
        V3<T> pc = sols[i].R.matrix() * pw + sols[i].t;
        pc = pc / norm(pc);
        T score = T(1.0) - dot(pc, adapter.getBearingVector(sel[3]));
        if (score < minScore) { minScore = score; minIndex = i; }
      }
      if (minIndex != -1) picked.push_back(sols[minIndex]);
    }
    if (hyp_capture<T>(picked)) continue;
    if (!picked.empty()) {
      const SE3<T>& out = picked[0];
      int votes = vote_23<T>(adapter, out, cos_thr, inliers, /*use_matrix=*/!prosac);
      if (votes > adapter.getMaxVotes()) {
        adapter.setMaxVotes(votes);
        adapter.setRcw(out.R); adapter.sett(out.t);
        adapter.setInlier(inliers);
        Iter = RANSACUpdateNumIters(confidence, (T)(N - votes) / N, K, Iter);
      }
    }
  }
  adapter.cvtInlier();
}
template <class T> void kneip_ransac(PnPPoseAdapter<T>& a, const T thr, int& Iter, T conf, Rand31& rnd) { kneip_ransac_impl<T>(a, thr, Iter, conf, rnd, false); }
template <class T> void kneip_prosac(PnPPoseAdapter<T>& a, const T thr, int& Iter, T conf, Rand31& rnd) { kneip_ransac_impl<T>(a, thr, Iter, conf, rnd, true); }

// assign_sample  AbsoluteOrientation.hpp:344-365
template <class T> bool assign_sample(const AOPoseAdapter<T>& ad, const std::vector<int>& sel, MatX<T>* Xw, MatX<T>* Xc, MatX<T>* bv) {
  int K = (int)sel.size() - 1, nValid = 0;
  for (int s = 0; s < K; s++) {
    Xw->set_col3(s, ad.getPointGlob(sel[s]));
    bv->set_col3(s, ad.getBearingVector(sel[s]));
    if (ad.isValid(sel[s])) { Xc->set_col3(s, ad.getPointCurr(sel[s])); nValid++; }
  }
  Xw->set_col3(3, ad.getPointGlob(sel[3]));
  bv->set_col3(3, ad.getBearingVector(sel[3]));
  return nValid == K;
}

// shinji_kneip_ransac :367-438 / shinji_kneip_prosac :440-515
template <class T> void shinji_kneip_impl(AOPoseAdapter<T>& adapter, const T thre_3d, const T thre_2d, int& Iter, T confidence, Rand31& rnd, bool prosac) {
  const T cos_thr = std::cos(std::atan(thre_2d / adapter.getFocal()));
  const int N = adapter.getNumberCorrespondences();
  RandomElements re(N);
  const int K = 3;
  MatX<T> X_w(3, K + 1), X_c(3, K + 1), bv(3, K + 1);
  MaskX inliers(N, 2);
  if (prosac) adapter.sortIdx();
  ProsacSampler<T> ps(K + 1, N);
  adapter.setMaxVotes(-1);
  for (int ii = 0; ii < Iter; ii++) {
    SE3<T> sk, ss;
    std::vector<SE3<T> > sols;
    if (!hyp_replay<T>(ii, &sols)) {
      std::vector<int> sel;
      if (prosac) { ps.sample(&sel, rnd); adapter.getSortedIdx(sel); }
      else re.run(K + 1, &sel, rnd);
      if (assign_sample<T>(adapter, sel, &X_w, &X_c, &bv)) { ss = shinji<T>(X_w, X_c, K); if (ss.R.ok) sols.push_back(ss); }
      if (kneip<T>(X_w, bv, &sk)) sols.push_back(sk);
    }
    if (hyp_capture<T>(sols)) continue;
    for (size_t h = 0; h < sols.size(); h++) {
      int votes = vote_33_23<T>(adapter, sols[h], thre_3d, cos_thr, inliers);
      if (votes > adapter.getMaxVotes()) {
        adapter.setMaxVotes(votes);
        adapter.setRcw(sols[h].R); adapter.sett(sols[h].t);
        adapter.setInlier(inliers);
        Iter = RANSACUpdateNumIters(confidence, (T)(N * 2 - votes) / N / 2, K, Iter);
      }
    }
  }
  PnPPoseAdapter<T>* p = &adapter;
  p->cvtInlier();
  adapter.cvtInlier();
}
template <class T> void shinji_kneip_ransac(AOPoseAdapter<T>& a, T t3, T t2, int& Iter, T conf, Rand31& rnd) { shinji_kneip_impl<T>(a, t3, t2, Iter, conf, rnd, false); }
template <class T> void shinji_kneip_prosac(AOPoseAdapter<T>& a, T t3, T t2, int& Iter, T conf, Rand31& rnd) { shinji_kneip_impl<T>(a, t3, t2, Iter, conf, rnd, true); }

// ------------------------------------------------------------------ normal-aware  pose/AbsoluteOrientationNormal.hpp
// L2 find_opt_cc :13-46   (uses the adapter's CURRENT pose, i.e. the RANSAC pose while L1 iterates)
template <class T> V3<T> find_opt_cc(NormalAOPoseAdapter<T>& adapter) {
  M3<T> Rwc = adapter.getRcw().inverse().matrix();
  M3<T> AA; V3<T> bb;
  for (int i = 0; i < adapter.getNumberCorrespondences(); i++) {
    if (adapter.isInlier23(i)) {
      V3<T> v = Rwc * adapter.getBearingVector(i);
      M3<T> A;
      A(0, 0) = 1 - v.x * v.x;
      A(1, 0) = A(0, 1) = -v.x * v.y;
      A(2, 0) = A(0, 2) = -v.x * v.z;
      A(1, 1) = 1 - v.y * v.y;
      A(2, 1) = A(1, 2) = -v.y * v.z;
      A(2, 2) = 1 - v.z * v.z;
      V3<T> b = A * adapter.getPointGlob(i);
      AA = AA + A;
      bb = bb + b;
    }
  }
  if (std::fabs(det(AA)) < T(0.0001)) {
    T nan = std::numeric_limits<T>::quiet_NaN();
    return V3<T>(nan, nan, nan);
  }
  return svd_solve(AA, bb);
}

// assign_sample (normals) :48-75
template <class T> bool assign_sample_nl(const NormalAOPoseAdapter<T>& ad, const std::vector<int>& sel, MatX<T>* Xw, MatX<T>* Nw,
                                         MatX<T>* Xc, MatX<T>* Nc, MatX<T>* bv) {
  int K = (int)sel.size() - 1, nValid = 0;
  for (int s = 0; s < K; s++) {
    Xw->set_col3(s, ad.getPointGlob(sel[s]));
    Nw->set_col3(s, ad.getNormalGlob(sel[s]));
    bv->set_col3(s, ad.getBearingVector(sel[s]));
    if (ad.isValid(sel[s])) { Xc->set_col3(s, ad.getPointCurr(sel[s])); Nc->set_col3(s, ad.getNormalCurr(sel[s])); nValid++; }
  }
  Xw->set_col3(3, ad.getPointGlob(sel[3]));
  Nw->set_col3(3, ad.getNormalGlob(sel[3]));
  bv->set_col3(3, ad.getBearingVector(sel[3]));
  return nValid == K;
}

// nl_2p :77-142 (Drost et al. 2010 two points + one normal)
template <class T> void nl_2p(const V3<T>& pt1_c, const V3<T>& nl1_c, const V3<T>& pt2_c, const V3<T>& pt1_w, const V3<T>& nl1_w,
                              const V3<T>& pt2_w, SE3<T>* sol) {
  V3<T> c_w = pt1_w;
  T alpha = std::acos(nl1_w.x);
  V3<T> axis(0, nl1_w.z, -nl1_w.y);
  axis = normalized(axis);
  SO3<T> R_g_f_w = SO3<T>::from_angle_axis(alpha, axis);
  V3<T> c_c = pt1_c;
  T beta = std::acos(nl1_c.x);
  V3<T> axis2(0, nl1_c.z, -nl1_c.y);
  axis2 = normalized(axis2);
  SO3<T> R_gp_f_c = SO3<T>::from_angle_axis(beta, axis2);
  V3<T> pt2_g = R_g_f_w * (pt2_w - c_w); pt2_g.x = T(0); pt2_g = normalized(pt2_g);
  V3<T> pt2_gp = R_gp_f_c * (pt2_c - c_c); pt2_gp.x = T(0); pt2_gp = normalized(pt2_gp);
  T gamma = std::acos(dot(pt2_g, pt2_gp));
  SO3<T> R_gp_f_g = SO3<T>::from_angle_axis(gamma, V3<T>(1, 0, 0));
  SO3<T> R_c_f_gp = R_gp_f_c.inverse();
  sol->R = R_c_f_gp * R_gp_f_g * R_g_f_w;
  sol->t = c_c - sol->R * c_w;
}

// nl_kneip_ransac :215-284
template <class T> void nl_kneip_ransac(NormalAOPoseAdapter<T>& adapter, const T thre_2d, const T nl_thre, int& Iter, T confidence, Rand31& rnd) {
  const T cos_thr = std::cos(std::atan(thre_2d / adapter.getFocal()));
  const T cos_nl = std::cos(nl_thre);
  const int N = adapter.getNumberCorrespondences();
  RandomElements re(N);
  const int K = 3;
  MatX<T> Xw(3, K + 1), Xc(3, K + 1), bv(3, K + 1), Nw(3, K + 1), Nc(3, K + 1);
  MaskX inl(N, 3);
  adapter.setMaxVotes(-1);
  for (int ii = 0; ii < Iter; ii++) {
    std::vector<SE3<T> > sols;
    if (!hyp_replay<T>(ii, &sols)) {
      SE3<T> one;
      std::vector<int> sel;
      re.run(K + 1, &sel, rnd);
      assign_sample_nl<T>(adapter, sel, &Xw, &Nw, &Xc, &Nc, &bv);
      if (kneip<T>(Xw, bv, &one)) sols.push_back(one);
    }
    if (hyp_capture<T>(sols) || sols.empty()) continue;
    const SE3<T>& sk = sols[0];
    int votes = vote_nn_23<T>(adapter, sk, cos_thr, cos_nl, inl);
    if (votes > adapter.getMaxVotes()) {
      adapter.setMaxVotes(votes);
      adapter.setRcw(sk.R); adapter.sett(sk.t);
      adapter.setInlier(inl);
      Iter = RANSACUpdateNumIters(confidence, (T)(N * 2 - votes) / N / 2, K, Iter);
    }
  }
  PnPPoseAdapter<T>* p = &adapter;
  p->cvtInlier();
  adapter.cvtInlier();
}

// nl_shinji_ransac :286-354
template <class T> void nl_shinji_ransac(NormalAOPoseAdapter<T>& adapter, const T thre_3d, const T nl_thre, int& Iter, T confidence, Rand31& rnd) {
  const T cos_nl = std::cos(nl_thre);
  const int N = adapter.getNumberCorrespondences();
  RandomElements re(N);
  const int K = 3;
  MatX<T> Xw(3, K + 1), Xc(3, K + 1), bv(3, K + 1), Nw(3, K + 1), Nc(3, K + 1);
  MaskX inl(N, 3);
  adapter.setMaxVotes(-1);
  for (int ii = 0; ii < Iter; ii++) {
    SE3<T> ss, sn;
    std::vector<SE3<T> > sols;
    if (!hyp_replay<T>(ii, &sols)) {
      std::vector<int> sel;
      re.run(K + 1, &sel, rnd);
      if (assign_sample_nl<T>(adapter, sel, &Xw, &Nw, &Xc, &Nc, &bv)) { ss = shinji<T>(Xw, Xc, K); if (ss.R.ok) sols.push_back(ss); }
      nl_2p<T>(Xc.col3(0), Nc.col3(0), Xc.col3(1), Xw.col3(0), Nw.col3(0), Xw.col3(1), &sn);
      sols.push_back(sn);
    }
    if (hyp_capture<T>(sols)) continue;
    for (size_t h = 0; h < sols.size(); h++) {
      int votes = vote_nn_33<T>(adapter, sols[h], thre_3d, cos_nl, inl);
      if (votes > adapter.getMaxVotes()) {
        adapter.setMaxVotes(votes);
        adapter.setRcw(sols[h].R); adapter.sett(sols[h].t);
        adapter.setInlier(inl);
        Iter = RANSACUpdateNumIters(confidence, (T)(N * 2 - votes) / N / 2, K, Iter);
      }
    }
  }
  AOPoseAdapter<T>* p = &adapter;
  p->cvtInlier();
  adapter.cvtInlier();
}

// nl_shinji_kneip_ransac :356-445
template <class T> void nl_shinji_kneip_ransac(NormalAOPoseAdapter<T>& adapter, const T thre_3d, const T thre_2d, const T nl_thre, int& Iter,
                                               T confidence, Rand31& rnd) {
  const T cos_thr = std::cos(std::atan(thre_2d / adapter.getFocal()));
  const T cos_nl = std::cos(nl_thre);
  const int N = adapter.getNumberCorrespondences();
  RandomElements re(N);
  const int K = 3;
  MatX<T> Xw(3, K + 1), Xc(3, K + 1), bv(3, K + 1), Nw(3, K + 1), Nc(3, K + 1);
  MaskX inl(N, 3);
  adapter.setMaxVotes(-1);
  for (int ii = 0; ii < Iter; ii++) {
    SE3<T> sk, ss, sn;
    std::vector<SE3<T> > sols;
    if (!hyp_replay<T>(ii, &sols)) {
      std::vector<int> sel;
      re.run(K + 1, &sel, rnd);
      if (assign_sample_nl<T>(adapter, sel, &Xw, &Nw, &Xc, &Nc, &bv)) { ss = shinji<T>(Xw, Xc, K); if (ss.R.ok) sols.push_back(ss); }
      if (kneip<T>(Xw, bv, &sk)) sols.push_back(sk);
      nl_2p<T>(Xc.col3(0), Nc.col3(0), Xc.col3(1), Xw.col3(0), Nw.col3(0), Xw.col3(1), &sn);
      sols.push_back(sn);
    }
    if (hyp_capture<T>(sols)) continue;
    for (size_t h = 0; h < sols.size(); h++) {
      int votes = vote_nn_33_23<T>(adapter, sols[h], thre_3d, cos_thr, cos_nl, inl);
      if (votes > adapter.getMaxVotes()) {
        adapter.setMaxVotes(votes);
        adapter.setRcw(sols[h].R); adapter.sett(sols[h].t);
        adapter.setInlier(inl);
        Iter = RANSACUpdateNumIters(confidence, (T)(N * 3 - votes) / N / 3, K, Iter);
      }
    }
  }
  PnPPoseAdapter<T>* p1 = &adapter; p1->cvtInlier();
  AOPoseAdapter<T>* p2 = &adapter; p2->cvtInlier();
  adapter.cvtInlier();
}

// L1 nl_shinji_kneip_ls :447-552.  bug_compatible=true reproduces the reference exactly: M33/M23/MNN,
// TW/TL and the counts K/M are NOT reset between the three rounds (:473-477 sit outside the loop :481).
template <class T> void nl_shinji_kneip_ls(NormalAOPoseAdapter<T>& adapter, bool bug_compatible = true) {
  if (adapter.getMaxVotes() == 0) return;
  const int NC = adapter.getNumberCorrespondences();
  V3<T> Cw, Cc; int N = 0; T TV = 0;
  for (int n = 0; n < NC; n++) {
    if (adapter.isInlier33(n)) {
      T v = adapter.weight33(n);
      Cw = Cw + v * adapter.getPointGlob(n);
      Cc = Cc + v * adapter.getPointCurr(n);
      TV += v; N++;
    }
  }
  if (N > 2) { Cw = Cw / TV; Cc = Cc / TV; }
  M3<T> M33, MNN, M23;
  int M = 0; T TL = 0;
  int K = 0; T TW = 0;
  V3<T> c_opt = adapter.getRcw().inverse() * (-adapter.gettw());
  SO3<T> R_opt;
  for (int ii = 0; ii < 3; ii++) {
    if (!bug_compatible) { M33 = M3<T>(); MNN = M3<T>(); M23 = M3<T>(); M = 0; TL = 0; K = 0; TW = 0; }
    T sigma_w_sqr = 0.;
    for (int nC = 0; nC < NC; nC++) {
      if (adapter.isInlier23(nC)) {
        T w = adapter.weight23(nC);
        V3<T> Aw = adapter.getPointGlob(nC) - c_opt; Aw = normalized(Aw);
        V3<T> Ac = adapter.getBearingVector(nC);
        M23 = M23 + w * outer(Ac, Aw);
        TW += w; K++;
      }
      if (adapter.isInlier33(nC)) {
        T v = adapter.weight33(nC);
        V3<T> Aw = adapter.getPointGlob(nC) - Cw;
        V3<T> Ac = adapter.getPointCurr(nC) - Cc; sigma_w_sqr += v * sqnorm(Ac);
        M33 = M33 + v * outer(Ac, Aw);
      }
      if (adapter.isInlierNN(nC)) {
        T lambda = adapter.weightNN(nC);
        M3<T> o = outer(adapter.getNormalCurr(nC), adapter.getNormalGlob(nC));
        MNN = MNN + lambda * o;
        TL += lambda; M++;
      }
    }
    if (N > 2) { M33 = (T(1) / TV) * M33; sigma_w_sqr /= TV; } else { M33 = M3<T>(); sigma_w_sqr = 1.; }
    if (M > 0) { MNN = (T(1) / TL) * MNN; } else { MNN = M3<T>(); }
    if (K > 0) { M23 = (T(1) / TW) * M23; } else { M23 = M3<T>(); }
    M33 = M33 + sigma_w_sqr * (M23 + MNN);
    SVD3<T> d = svd3(M33);
    M3<T> TMP = d.U * transpose(d.V);
    if (det(TMP) < 0) { M3<T> I = M3<T>::identity(); I(2, 2) = -1; R_opt = SO3<T>(d.U * I * transpose(d.V)); }
    else R_opt = SO3<T>(TMP);
    V3<T> c = Cw - R_opt.inverse() * Cc;
    V3<T> cp = find_opt_cc<T>(adapter);
    if (N > 2) {
      if (cp.x == cp.x) c_opt = (T(K) / (K + N)) * cp + (T(N) / (K + N)) * c;
      else c_opt = c;
    } else {
      if (cp.x == cp.x) c_opt = cp;
      else break;
    }
  }
  adapter.setRcw(R_opt);
  adapter.sett(R_opt * (-c_opt));
}

// R1  lsq_pnp :472-502 : sum of sine residuals (the reference prints it; we return it)
template <class T> T lsq_pnp(PnPPoseAdapter<T>& adapter) {
  T total = 0;
  for (int i = 0; i < adapter.getNumberCorrespondences(); i++) {
    V3<T> Xc = adapter.getRcw() * adapter.getPointGlob(i) + adapter.gettw();
    Xc = normalized(Xc);
    total += norm(cross(Xc, adapter.getBearingVector(i)));
  }
  return total;
}

}  // namespace orc
