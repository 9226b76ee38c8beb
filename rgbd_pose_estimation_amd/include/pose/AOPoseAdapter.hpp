// pose/AOPoseAdapter.hpp -- drop-in for /root/reference/pose/AOPoseAdapter.hpp:26-217 (2D-3D + 3D-3D; a NaN
// points_c column means "no 3-D measurement", :147-152).
#ifndef RPE_AO_POSE_ADAPTER_HEADER
#define RPE_AO_POSE_ADAPTER_HEADER

#include <limits>
#include "PnPPoseAdapter.hpp"

template <typename Tp>
class AOPoseAdapter : public PnPPoseAdapter<Tp> {
 protected:
  using PoseAdapterBase<Tp>::_t_w;
  using PoseAdapterBase<Tp>::_R_cw;
  using PnPPoseAdapter<Tp>::_bearingVectors;
  using PnPPoseAdapter<Tp>::_points_g;

 public:
  typedef typename PoseAdapterBase<Tp>::Vector3 Vector3;
  typedef typename PoseAdapterBase<Tp>::SO3_T SO3_T;
  typedef typename PoseAdapterBase<Tp>::Point3 Point3;
  typedef typename PnPPoseAdapter<Tp>::MatrixX MatrixX;

  template <class M> AOPoseAdapter(const M& bearingVectors, const M& points_c, const M& points_g)
      : PnPPoseAdapter<Tp>(bearingVectors, points_g), _points_c(points_c) { init3(); }
  template <class M> AOPoseAdapter(const M& bearingVectors, const M& points_c, const M& points_g, const SO3_T& R)
      : PnPPoseAdapter<Tp>(bearingVectors, points_g, R), _points_c(points_c) { init3(); }
  template <class M> AOPoseAdapter(const M& bearingVectors, const M& points_c, const M& points_g, const Vector3& t, const SO3_T& R)
      : PnPPoseAdapter<Tp>(bearingVectors, points_g, t, R), _points_c(points_c) { init3(); }
  virtual ~AOPoseAdapter() {}

  bool isInlier33(int index) const { return mask33()[index] == 1; }
  // NB divides by SHRT_MAX, unlike AOOnlyPoseAdapter (reference :161-169)
  Tp weight33(int index) const { return _weights_3d.empty() ? Tp(1.0) : Tp(_weights_3d[index]) / std::numeric_limits<short>::max(); }
  virtual Point3 getPointCurr(int index) const { return _points_c.col(index); }
  virtual bool isValid(int index) const { Point3 p = _points_c.col(index); return p[0] == p[0] || p[1] == p[1] || p[2] == p[2]; }
  virtual void setInlier(const rpe::MatrixXs& inliers) {  // reference :171-184
    PnPPoseAdapter<Tp>::setInlier(inliers);
    if (inliers.cols() != 1) {
      flushInlierIdx33();
      std::vector<short>& m = _inliers_3d.replace(this->device(), RPE_MOD_33);
      for (int i = 0; i < inliers.rows(); i++) m[i] = inliers(i, 1);
    }
  }
  virtual void setInlierFromDevice(int cols, unsigned device_cols) {   // additive, see AOOnlyPoseAdapter::setInlierFromDevice
    PnPPoseAdapter<Tp>::setInlierFromDevice(cols, device_cols);
    if (cols != 1) {
      flushInlierIdx33();
      if (device_cols & 2u) _inliers_3d.device_is_newer(this->device(), RPE_MOD_33);
      else _inliers_3d.set_all(this->device(), RPE_MOD_33, (short)0);
    }
  }
  template <class M> void setWeights(const M& weights) {  // reference :186-199
    this->setWeights23(weights);
    if (weights.rows() != 1) setWeights33(weights);
  }
  virtual void printInlier() const {
    PnPPoseAdapter<Tp>::printInlier();
    for (short v : mask33()) std::cout << v << " ";
    std::cout << std::endl;
  }
  const std::vector<int>& getInlierIdx() const { flushInlierIdx33(); return _vInliersAO.get(mask33()); }
  void cvtInlier() { _vInliersAO.request(); }  // built on first read (rpe::InlierIndex)
  void forgetInlierIdx() { _vInliersAO.drop(); }  // additive, for solvers: see rpe::InlierIndex::drop

  // ---- additive accessors for the device backend
  const Tp* pointsCurrData() const { return _points_c.p; }
  std::vector<short>& inlierMask33() { flushInlierIdx33(); return _inliers_3d.edit(this->device(), RPE_MOD_33); }
  const std::vector<short>& inlierMask33() const { return mask33(); }
  void copyInlierMask33(short* dst) const { _inliers_3d.copy_to(this->device(), RPE_MOD_33, dst); }   // additive: n shorts, no host copy kept
  void pushMask33() const { _inliers_3d.push(this->device(), RPE_MOD_33); }
  virtual void syncHostMasks() const { PnPPoseAdapter<Tp>::syncHostMasks(); (void)mask33(); }
  const std::vector<Tp>& weights33() const { return _weights_3d; }
  Tp weightScale33() const { return (Tp)std::numeric_limits<short>::max(); }

 protected:
  void init3() { _inliers_3d.assign((size_t)_bearingVectors.cols(), (short)1); }
  const std::vector<short>& mask33() const { return _inliers_3d.read(this->device(), RPE_MOD_33); }
  void flushInlierIdx33() const { if (_vInliersAO.pending()) _vInliersAO.flush(mask33()); }
  template <class M> void setWeights33(const M& weights) {
    _weights_3d.resize(weights.rows());
    for (int i = 0; i < (int)weights.rows(); i++) _weights_3d[i] = weights(i, 1);
    this->device().weight_changed_on_host(RPE_MOD_33);
  }
  rpe::ColumnView<Tp> _points_c;
  rpe::HostMask _inliers_3d;
  std::vector<Tp> _weights_3d;
  rpe::InlierIndex _vInliersAO;
};

#endif
