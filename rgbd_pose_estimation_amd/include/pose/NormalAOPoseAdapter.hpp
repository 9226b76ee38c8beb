// pose/NormalAOPoseAdapter.hpp -- drop-in for /root/reference/pose/NormalAOPoseAdapter.hpp:16-231
// (2D-3D + 3D-3D + normal-normal correspondences).
#ifndef RPE_NORMAL_AO_POSE_ADAPTER_HEADER
#define RPE_NORMAL_AO_POSE_ADAPTER_HEADER

#include "AOPoseAdapter.hpp"

template <typename Tp>
class NormalAOPoseAdapter : public AOPoseAdapter<Tp> {
 protected:
  using PoseAdapterBase<Tp>::_t_w;
  using PoseAdapterBase<Tp>::_R_cw;
  using PnPPoseAdapter<Tp>::_bearingVectors;
  using PnPPoseAdapter<Tp>::_points_g;
  using AOPoseAdapter<Tp>::_points_c;
  typedef typename PoseAdapterBase<Tp>::Point3 Point3;

 public:
  typedef typename PoseAdapterBase<Tp>::Vector3 Vector3;
  typedef typename PoseAdapterBase<Tp>::SO3_T SO3_T;
  typedef typename PnPPoseAdapter<Tp>::MatrixX MatrixX;

  template <class M>
  NormalAOPoseAdapter(const M& bearingVectors, const M& points_c, const M& normal_c, const M& points_g, const M& normal_g)
      : AOPoseAdapter<Tp>(bearingVectors, points_c, points_g), _normal_c(normal_c), _normal_g(normal_g) { initn(); }
  template <class M>
  NormalAOPoseAdapter(const M& bearingVectors, const M& points_c, const M& normal_c, const M& points_g, const M& normal_g,
      const SO3_T& R)
      : AOPoseAdapter<Tp>(bearingVectors, points_c, points_g, R), _normal_c(normal_c), _normal_g(normal_g) { initn(); }
  template <class M>
  NormalAOPoseAdapter(const M& bearingVectors, const M& points_c, const M& normal_c, const M& points_g, const M& normal_g,
                      const Vector3& t, const SO3_T& R)
      : AOPoseAdapter<Tp>(bearingVectors, points_c, points_g, t, R), _normal_c(normal_c), _normal_g(normal_g) { initn(); }
  virtual ~NormalAOPoseAdapter() {}

  bool isInlierNN(int index) const { return maskNN()[index] == 1; }
  // :153-161
  Tp weightNN(int index) const { return _weights_nl.empty() ? Tp(1.0) : Tp(_weights_nl[index]) / std::numeric_limits<short>::max(); }
  virtual Point3 getNormalCurr(int index) const { return _normal_c.col(index); }
  virtual Point3 getNormalGlob(int index) const { return _normal_g.col(index); }
  virtual void setInlier(const rpe::MatrixXs& inliers) {  // reference :179-195
    if (inliers.cols() == 1) PnPPoseAdapter<Tp>::setInlier(inliers);
    if (inliers.cols() == 2) AOPoseAdapter<Tp>::setInlier(inliers);
    if (inliers.cols() == 3) {
      AOPoseAdapter<Tp>::setInlier(inliers);
      flushInlierIdxNN();
      std::vector<short>& m = _inliers_nl.replace(this->device(), RPE_MOD_NN);
      for (int i = 0; i < inliers.rows(); i++) m[i] = inliers(i, 2);
    }
  }
  virtual void setInlierFromDevice(int cols, unsigned device_cols) {   // additive, see AOOnlyPoseAdapter::setInlierFromDevice
    if (cols == 1) PnPPoseAdapter<Tp>::setInlierFromDevice(cols, device_cols);
    if (cols == 2) AOPoseAdapter<Tp>::setInlierFromDevice(cols, device_cols);
    if (cols == 3) {
      AOPoseAdapter<Tp>::setInlierFromDevice(cols, device_cols);
      flushInlierIdxNN();
      if (device_cols & 4u) _inliers_nl.device_is_newer(this->device(), RPE_MOD_NN);
      else _inliers_nl.set_all(this->device(), RPE_MOD_NN, (short)0);
    }
  }
  template <class M> void setWeights(const M& weights) {  // reference :197-212 dispatches on cols()
    if (weights.cols() == 1) this->setWeights23(weights);
    if (weights.cols() == 2) AOPoseAdapter<Tp>::setWeights(weights);
    if (weights.cols() == 3) {
      AOPoseAdapter<Tp>::setWeights(weights);
      _weights_nl.resize(weights.rows());
      for (int i = 0; i < (int)weights.rows(); i++) _weights_nl[i] = weights(i, 2);
      this->device().weight_changed_on_host(RPE_MOD_NN);
    }
  }
  virtual void printInlier() const {
    AOPoseAdapter<Tp>::printInlier();
    for (short v : maskNN()) std::cout << v << " ";
    std::cout << std::endl;
  }
  const std::vector<int>& getInlierIdx() const { flushInlierIdxNN(); return _vInliersNN.get(maskNN()); }
  void cvtInlier() { _vInliersNN.request(); }  // built on first read (rpe::InlierIndex)
  void forgetInlierIdx() { _vInliersNN.drop(); }  // additive, for solvers: see rpe::InlierIndex::drop

  // ---- additive accessors for the device backend
  const Tp* normalCurrData() const { return _normal_c.p; }
  const Tp* normalGlobData() const { return _normal_g.p; }
  std::vector<short>& inlierMaskNN() { flushInlierIdxNN(); return _inliers_nl.edit(this->device(), RPE_MOD_NN); }
  const std::vector<short>& inlierMaskNN() const { return maskNN(); }
  void copyInlierMaskNN(short* dst) const { _inliers_nl.copy_to(this->device(), RPE_MOD_NN, dst); }   // additive: n shorts, no host copy kept
  void pushMaskNN() const { _inliers_nl.push(this->device(), RPE_MOD_NN); }
  virtual void syncHostMasks() const { AOPoseAdapter<Tp>::syncHostMasks(); (void)maskNN(); }
  const std::vector<Tp>& weightsNN() const { return _weights_nl; }
  Tp weightScaleNN() const { return (Tp)std::numeric_limits<short>::max(); }

 protected:
  void initn() { _inliers_nl.assign((size_t)_bearingVectors.cols(), (short)1); }
  const std::vector<short>& maskNN() const { return _inliers_nl.read(this->device(), RPE_MOD_NN); }
  void flushInlierIdxNN() const { if (_vInliersNN.pending()) _vInliersNN.flush(maskNN()); }
  rpe::ColumnView<Tp> _normal_c, _normal_g;
  rpe::HostMask _inliers_nl;
  std::vector<Tp> _weights_nl;
  rpe::InlierIndex _vInliersNN;
};

#endif
