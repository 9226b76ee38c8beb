// pose/PnPPoseAdapter.hpp -- drop-in for /root/reference/pose/PnPPoseAdapter.hpp:27-255 (2D-3D: bearings + world points).
#ifndef RPE_PNP_POSE_ADAPTER_HEADER
#define RPE_PNP_POSE_ADAPTER_HEADER

#include <iostream>
#include "PoseAdapterBase.hpp"
#include "Utility.hpp"

template <typename Tp>
class PnPPoseAdapter : public PoseAdapterBase<Tp> {
 protected:
  using PoseAdapterBase<Tp>::_t_w;
  using PoseAdapterBase<Tp>::_R_cw;

 public:
  typedef typename PoseAdapterBase<Tp>::Vector3 Vector3;
  typedef typename PoseAdapterBase<Tp>::SO3_T SO3_T;
  typedef typename PoseAdapterBase<Tp>::Point3 Point3;
  typedef rpe::MatrixX<Tp> MatrixX;

  template <class M> PnPPoseAdapter(const M& bearingVectors, const M& points)
      : PoseAdapterBase<Tp>(), _bearingVectors(bearingVectors), _points_g(points) { init(); }
  template <class M> PnPPoseAdapter(const M& bearingVectors, const M& points, const SO3_T& R)
      : PoseAdapterBase<Tp>(R), _bearingVectors(bearingVectors), _points_g(points) { init(); }
  template <class M> PnPPoseAdapter(const M& bearingVectors, const M& points, const Vector3& t, const SO3_T& R)
      : PoseAdapterBase<Tp>(t, R), _bearingVectors(bearingVectors), _points_g(points) { init(); }
  virtual ~PnPPoseAdapter() {}

  virtual Point3 getBearingVector(int index) const { return _bearingVectors.col(index); }
  virtual Tp getWeight(int) const { return Tp(1.); }
  virtual Point3 getPointGlob(int index) const { return _points_g.col(index); }
  virtual int getNumberCorrespondences() const { return _bearingVectors.cols(); }

  // column 0 of the mask (reference :196-202 memcpy's rows()*2 bytes = column 0 of a column-major short matrix)
  virtual void setInlier(const rpe::MatrixXs& inliers) {
    flushInlierIdx23();
    std::vector<short>& m = _inliers.replace(this->device(), RPE_MOD_23);
    for (int i = 0; i < inliers.rows(); i++) m[i] = inliers(i, 0);
  }
  // additive, for the GPU solvers (see AOOnlyPoseAdapter::setInlierFromDevice): column 0 from the device, or zero
  virtual void setInlierFromDevice(int cols, unsigned device_cols) {
    (void)cols;
    flushInlierIdx23();
    if (device_cols & 1u) _inliers.device_is_newer(this->device(), RPE_MOD_23);
    else _inliers.set_all(this->device(), RPE_MOD_23, (short)0);
  }
  template <class M> void setWeights(const M& weights) { setWeights23(weights); }
  virtual void printInlier() const { for (short v : mask23()) std::cout << v << " "; std::cout << std::endl; }
  const std::vector<int>& getInlierIdx() const { flushInlierIdx23(); return _vInliersPnP.get(mask23()); }
  void cvtInlier() { _vInliersPnP.request(); }  // built on first read (rpe::InlierIndex)
  void forgetInlierIdx() { _vInliersPnP.drop(); }  // additive, for solvers: see rpe::InlierIndex::drop
  // sine of the angle between predicted and observed bearing (reference :204-210)
  Tp getError(int index) const {
    Point3 Xc = _R_cw * getPointGlob(index) + _t_w;
    Xc.normalize();
    return Xc.cross(getBearingVector(index)).norm();
  }
  void setMaxVotes(int votes) { _max_votes = votes; }
  int getMaxVotes() { return _max_votes; }
  bool isInlier23(int index) const { return mask23()[index] == 1; }
  Tp weight23(int index) const { return _weights.empty() ? Tp(1.0) : _weights[index]; }
  // top_k >= 0: only the first top_k positions of the order are needed now (the rest is sorted on demand)
  void sortIdx(int top_k = -1) {   // cached per weight set, see AOOnlyPoseAdapter::sortIdx
    const int want = top_k < 0 || top_k > (int)_weights.size() ? (int)_weights.size() : top_k;
    if (_idx_top >= want && (int)_idx.size() >= want && want > 0) return;
    // top-k select + sort on the GPU for a dense frame's weights ...
    _idx = rpe::device_prosac_order<Tp>(this->device(), _weights, top_k);
    if (_idx.empty()) _idx = sortIndexes<Tp>(_weights, top_k);                  // ... the same prefix on the host otherwise
    _idx_top = (int)_idx.size();
  }
  void getSortedIdx(std::vector<int>& select_) const { mapSortedIdx<Tp>(_weights, _idx, select_); }

  // ---- additive accessors for the device backend
  const Tp* bearingData() const { return _bearingVectors.p; }
  const Tp* pointsGlobData() const { return _points_g.p; }
  std::vector<short>& inlierMask23() { flushInlierIdx23(); return _inliers.edit(this->device(), RPE_MOD_23); }
  const std::vector<short>& inlierMask23() const { return mask23(); }
  void copyInlierMask23(short* dst) const { _inliers.copy_to(this->device(), RPE_MOD_23, dst); }   // additive: n shorts, no host copy kept
  void pushMask23() const { _inliers.push(this->device(), RPE_MOD_23); }
  virtual void syncHostMasks() const { (void)mask23(); }
  const std::vector<Tp>& weights23() const { return _weights; }

 protected:
  void init() { _inliers.assign((size_t)_bearingVectors.cols(), (short)1); _max_votes = 0; }
  const std::vector<short>& mask23() const { return _inliers.read(this->device(), RPE_MOD_23); }
  void flushInlierIdx23() const { if (_vInliersPnP.pending()) _vInliersPnP.flush(mask23()); }
  template <class M> void setWeights23(const M& weights) {
    _weights.resize(weights.rows());
    _idx.clear(); _idx_top = 0;   // the cached PROSAC order belongs to the old weights
    for (int i = 0; i < (int)weights.rows(); i++) _weights[i] = weights(i, 0);
    this->device().weight_changed_on_host(RPE_MOD_23);
  }
  rpe::ColumnView<Tp> _bearingVectors, _points_g;
  rpe::HostMask _inliers;
  std::vector<Tp> _weights;
  mutable std::vector<int> _idx;
  mutable int _idx_top = 0;
  rpe::InlierIndex _vInliersPnP;
  int _max_votes;
};

#endif
