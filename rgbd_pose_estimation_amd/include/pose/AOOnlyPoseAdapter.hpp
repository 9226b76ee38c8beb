// pose/AOOnlyPoseAdapter.hpp -- drop-in for /root/reference/pose/AOOnlyPoseAdapter.hpp:26-255 (3D-3D only).
// Differences, all additive or fixes of undefined behaviour:
//   * inlier index lists are int, not short (the reference's `for (short r...)` loop never runs for N > 32767,
//     :222-231, so RANSAC -> LS breaks at 640x480 = 307200 points)
//   * the referenced matrices may be any type with data()/rows()/cols() (column-major 3 x N)
//   * pointsCurrData()/pointsGlobData() expose the contiguous arrays to the device backend
#ifndef RPE_AO_ONLY_POSE_ADAPTER_HEADER
#define RPE_AO_ONLY_POSE_ADAPTER_HEADER

#include <algorithm>
#include <iostream>
#include <numeric>
#include "PoseAdapterBase.hpp"
#include "Utility.hpp"

template <typename Tp>
class AOOnlyPoseAdapter : public PoseAdapterBase<Tp> {
 protected:
  using PoseAdapterBase<Tp>::_t_w;
  using PoseAdapterBase<Tp>::_R_cw;

 public:
  typedef typename PoseAdapterBase<Tp>::Vector3 Vector3;
  typedef typename PoseAdapterBase<Tp>::SO3_T SO3_T;
  typedef typename PoseAdapterBase<Tp>::Point3 Point3;
  typedef rpe::MatrixX<Tp> MatrixX;

  template <class M> AOOnlyPoseAdapter(const M& points_c, const M& points_g)
      : PoseAdapterBase<Tp>(), _points_c(points_c), _points_g(points_g) { init(); }
  template <class M> AOOnlyPoseAdapter(const M& points_c, const M& points_g, const SO3_T& R)
      : PoseAdapterBase<Tp>(R), _points_c(points_c), _points_g(points_g) { init(); }
  template <class M> AOOnlyPoseAdapter(const M& points_c, const M& points_g, const Vector3& t, const SO3_T& R)
      : PoseAdapterBase<Tp>(t, R), _points_c(points_c), _points_g(points_g) { init(); }
  virtual ~AOOnlyPoseAdapter() {}

  bool isInlier33(int index) const { return mask33()[index] == 1; }
  Tp weight33(int index) const { return _weights_3d.empty() ? Tp(1.0) : _weights_3d[index]; }  // raw weight (reference :175-183)
  virtual Point3 getBearingVector(int) const { return Point3(); }
  virtual Point3 getPointCurr(int index) const { return _points_c.col(index); }
  virtual Point3 getPointGlob(int index) const { return _points_g.col(index); }
  virtual Tp getWeight(int) const { return Tp(1.); }
  virtual int getNumberCorrespondences() const { return _points_g.cols(); }

  void setMaxVotes(int votes) { _max_votes = votes; }
  int getMaxVotes() { return _max_votes; }

  // a column is invalid only when ALL THREE coordinates are NaN (reference :147-152 uses ||)
  virtual bool isValid(int index) const { Point3 p = _points_c.col(index); return p[0] == p[0] || p[1] == p[1] || p[2] == p[2]; }
  // N x 2 mask: column 1 is the 3D-3D inlier flag; an N x 1 mask is ignored (reference :185-198)
  virtual void setInlier(const rpe::MatrixXs& inliers) {
    if (inliers.cols() != 1) {
      flushInlierIdx();
      std::vector<short>& m = _inliers_3d.replace(this->device(), RPE_MOD_33);
      for (int i = 0; i < inliers.rows(); i++) m[i] = inliers(i, 1);
    }
  }
  // additive, for the GPU solvers: setInlier() of an N x cols matrix whose columns in `device_cols` (bit c = column c) are the
  // masks a kernel just wrote on the device -- adopted without a download -- and whose other columns are zero, as the matrix the
  // reference's solvers build would be (AbsoluteOrientation.hpp:134-143 allocates N x 2 and fills column 1 only)
  virtual void setInlierFromDevice(int cols, unsigned device_cols) {
    if (cols != 1) {
      flushInlierIdx();
      if (device_cols & 2u) _inliers_3d.device_is_newer(this->device(), RPE_MOD_33);
      else _inliers_3d.set_all(this->device(), RPE_MOD_33, (short)0);
    }
  }
  // N x 3 weights: column 1 (reference :200-212 tests rows() == 1)
  template <class M> void setWeights(const M& weights) {
    if (weights.rows() != 1) {
      _weights_3d.resize(weights.rows());
      for (int i = 0; i < (int)weights.rows(); i++) _weights_3d[i] = weights(i, 1);
      _idx.clear(); _idx_top = 0;   // the cached PROSAC order belongs to the old weights
      this->device().weight_changed_on_host(RPE_MOD_33);
    }
  }
  virtual void printInlier() const { for (short v : mask33()) std::cout << v << " "; std::cout << std::endl; }
  const std::vector<int>& getInlierIdx() const { flushInlierIdx(); return _vInliersAO.get(mask33()); }
  void cvtInlier() { _vInliersAO.request(); }  // built on first read (rpe::InlierIndex)
  void forgetInlierIdx() { _vInliersAO.drop(); }  // additive, for solvers: see rpe::InlierIndex::drop
  // top_k >= 0: only the first top_k positions of the order are needed now (the rest is sorted on demand)
  // (the order is a pure function of the weights: a prefix at least as long as the one asked for, computed since the last
  // setWeights, is reused -- several PROSAC solvers on one adapter, TestMain.cpp:186-221, sort once)
  void sortIdx(int top_k = -1) {
    const int want = top_k < 0 || top_k > (int)_weights_3d.size() ? (int)_weights_3d.size() : top_k;
    if (_idx_top >= want && (int)_idx.size() >= want && want > 0) return;
    // top-k select + sort on the GPU for a dense frame's weights ...
    _idx = rpe::device_prosac_order<Tp>(this->device(), _weights_3d, top_k);
    if (_idx.empty()) _idx = sortIndexes<Tp>(_weights_3d, top_k);                  // ... the same prefix on the host otherwise
    _idx_top = (int)_idx.size();
  }
  void getSortedIdx(std::vector<int>& select_) const { mapSortedIdx<Tp>(_weights_3d, _idx, select_); }

  // ---- additive accessors for the device backend
  const Tp* pointsCurrData() const { return _points_c.p; }
  const Tp* pointsGlobData() const { return _points_g.p; }
  // host copy, for modification
  std::vector<short>& inlierMask33() { flushInlierIdx(); return _inliers_3d.edit(this->device(), RPE_MOD_33); }
  // host copy, read only
  const std::vector<short>& inlierMask33() const { return mask33(); }
  void copyInlierMask33(short* dst) const { _inliers_3d.copy_to(this->device(), RPE_MOD_33, dst); }   // additive: n shorts, no host copy kept
  // device copy current
  void pushMask33() const { _inliers_3d.push(this->device(), RPE_MOD_33); }
  virtual void syncHostMasks() const { (void)mask33(); }
  const std::vector<Tp>& weights33() const { return _weights_3d; }
  Tp weightScale33() const { return Tp(1); }

 protected:
  void init() { _inliers_3d.assign((size_t)_points_c.cols(), (short)1); _max_votes = 0; }
  const std::vector<short>& mask33() const { return _inliers_3d.read(this->device(), RPE_MOD_33); }
  void flushInlierIdx() const { if (_vInliersAO.pending()) _vInliersAO.flush(mask33()); }
  rpe::ColumnView<Tp> _points_c, _points_g;
  rpe::HostMask _inliers_3d;
  std::vector<Tp> _weights_3d;
  mutable std::vector<int> _idx;
  mutable int _idx_top = 0;   // how many leading positions of _idx are valid for the current weights
  rpe::InlierIndex _vInliersAO;
  int _max_votes;
};

#endif
