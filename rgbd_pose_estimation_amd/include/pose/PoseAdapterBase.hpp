// pose/PoseAdapterBase.hpp -- drop-in for the reference's header of the same name
// (/root/reference/pose/PoseAdapterBase.hpp:28-146): the abstract correspondence accessor + pose carrier.
// Same class name, same virtual interface, same pose/focal accessors; Eigen/Sophus types are replaced by the
// Eigen-free value types of rpe/types.hpp (same spellings inside the class: Vector3, Point3, SO3_T, SE3_T).
// Additive: device() -- the adapter's correspondence arrays resident in HBM, used by the HIP-backed solvers.
#ifndef RPE_POSE_ADAPTERBASE_HEADER
#define RPE_POSE_ADAPTERBASE_HEADER

#include <memory>
#include <cstring>
#include <vector>
#include "../rpe/types.hpp"
#include "../rpe/device.hpp"

template <typename Tp>
class PoseAdapterBase {
 public:
  typedef rpe::Point3<Tp> Vector3;
  typedef rpe::Point3<Tp> Point3;
  typedef rpe::SO3<Tp> SO3_T;
  typedef rpe::SE3<Tp> SE3_T;

  PoseAdapterBase() : _fx(0), _fy(0), _cx(0), _cy(0) {}
  explicit PoseAdapterBase(const SO3_T& R) : _R_cw(R), _fx(0), _fy(0), _cx(0), _cy(0) {}
  PoseAdapterBase(const Vector3& t, const SO3_T& R) : _t_w(t), _R_cw(R), _fx(0), _fy(0), _cx(0), _cy(0) {}
  virtual ~PoseAdapterBase() {}

  // per-correspondence access (reference :81-101)
  virtual Point3 getBearingVector(int index) const = 0;
  virtual Tp getWeight(int index) const = 0;
  virtual Point3 getPointGlob(int index) const = 0;
  virtual int getNumberCorrespondences() const = 0;

  // pose: Xc = R_cw * Xw + t_w (reference :109-130)
  Vector3 gettw() const { return _t_w; }
  void sett(const Vector3& t) { _t_w = t; }
  SO3_T getRcw() const { return _R_cw; }
  void setRcw(const SO3_T& R) { _R_cw = R; }
  void setFocal(const Tp fx, const Tp fy) { _fx = fx; _fy = fy; }
  Tp getFocal() const { return (_fx + _fy) / 2; }
  SE3_T getTcw() { return SE3_T(_R_cw, _t_w); }

  // ---- additive: HBM residency of this adapter's arrays
  rpe::DeviceSet& device() const {
    if (!_dev) _dev.reset(new rpe::DeviceSet());
    return *_dev;
  }
  // call after changing the CONTENTS of a matrix the adapter references (the adapters hold references, reference
  // :93-95 of AOOnlyPoseAdapter.hpp, and cache uploads by address)
  void invalidateDevice() { syncHostMasks(); _dev.reset(); }
  // bring every host-side inlier mask up to date (they may live on the device only, see rpe::HostMask)
  virtual void syncHostMasks() const {}

 protected:
  Vector3 _t_w;
  SO3_T _R_cw;
  Tp _fx, _fy, _cx, _cy;
  mutable std::shared_ptr<rpe::DeviceSet> _dev;
};

namespace rpe {
// Non-owning view of a caller-owned 3 x N column-major matrix: anything with data() / rows() / cols()
// (rpe::MatrixX<Tp>, Eigen::Matrix<Tp,Dynamic,Dynamic>, Eigen::Map<...>) binds to it.
template <class Tp> struct ColumnView {
  const Tp* p;
  int n;
  ColumnView() : p(nullptr), n(0) {}
  template <class M> explicit ColumnView(const M& m) : p(m.data()), n((int)m.cols()) {}
  Point3<Tp> col(int i) const { return Point3<Tp>(p + 3 * (size_t)i); }
  int cols() const { return n; }
};
// rows of a 0/1 mask that are 1, in order (the adapters' cvtInlier, e.g. reference AOOnlyPoseAdapter.hpp:214-222).  Branch-free
// compaction: the inlier pattern is unpredictable, a conditional push_back mispredicts on every outlier (4x slower at 307200 rows)
inline void indices_of_ones(const std::vector<short>& mask, std::vector<int>& out) {
  out.resize(mask.size() + 1);
  int* o = out.data();
  size_t k = 0;
  for (size_t r = 0; r < mask.size(); r++) { o[k] = (int)r; k += (mask[r] == 1); }
  out.resize(k);
}
// One modality's inlier mask as the adapters keep it on the host (N shorts, 0/1) -- except that after a RANSAC run on the GPU
// the TRUTH is the device copy (rpe_inlier_mask wrote it there), and most callers never look at the host copy: the least-squares
// stages read the device masks.  So the host copy is fetched on first access instead of after every solver run (0.15 ms per
// modality at 307200 rows, half of a whole RANSAC run).  read(): host access; edit(): host access that will modify;
// replace(): the whole content is about to be overwritten; device_is_newer(): a kernel just wrote the device copy.
// assign() is LAZY too: an adapter is constructed over every frame, and filling N shorts per modality up front (a fresh 600 KB
// allocation at 640 x 480: page faults and all, ~50 us each) is wasted on the calls that never look at a mask on the host -- ao()
// spends a fifth of its wall time there.  The vector is materialised by the first access that needs it.
class HostMask {
 public:
  void assign(size_t n, short v) { _n = n; _fill = v; _lazy = true; _stale = false; }
  size_t size() const { return _lazy ? _n : _v.size(); }
  const std::vector<short>& read(DeviceSet& dev, int mod) const {
    materialise();
    if (_stale) { dev.download_mask(mod, _v.data()); _stale = false; }
    return _v;
  }
  std::vector<short>& edit(DeviceSet& dev, int mod) { read(dev, mod); dev.mask_changed_on_host(mod); return _v; }
  std::vector<short>& replace(DeviceSet& dev, int mod) { materialise(); _stale = false; dev.mask_changed_on_host(mod); return _v; }
  void device_is_newer(DeviceSet& dev, int mod) { _stale = true; dev.mask_written_on_device(mod); }
  // every row becomes v: recorded, not written (a solver zeroes the modalities it does not vote on after every run)
  void set_all(DeviceSet& dev, int mod, short v) { _n = size(); _fill = v; _lazy = true; _stale = false; dev.mask_changed_on_host(mod); }
  // make the device copy current (no-op when it already is, in particular when it is the newer one)
  void push(DeviceSet& dev, int mod) const { if (!dev.mask_fresh(mod)) { materialise(); dev.upload_mask(mod, _v); } }
  // the current content into the CALLER's buffer (n shorts): straight from the device when the device copy is the newer one -- no host
  // vector is materialised, filled and copied on the way (rpe_run's mask_out: 0.24-0.44 ms of a 0.55-1.0 ms call went there)
  void copy_to(DeviceSet& dev, int mod, short* dst) const {
    if (_stale) dev.download_mask(mod, dst);
    else if (_lazy) std::fill(dst, dst + _n, _fill);
    else std::memcpy(dst, _v.data(), _v.size() * sizeof(short));
  }
 private:
  void materialise() const { if (_lazy) { _v.assign(_n, _fill); _lazy = false; } }
  mutable std::vector<short> _v;
  mutable size_t _n = 0;
  mutable short _fill = 0;
  mutable bool _lazy = false;
  mutable bool _stale = false;
};

// The index list behind cvtInlier() / getInlierIdx().  cvtInlier() only RECORDS the request; the list is built when it is
// first read, or just before the mask it was requested for changes -- so a reader always sees the list of the mask as it was
// when cvtInlier() ran (the reference's behaviour), and solvers that never read it (every pipeline here keeps its masks on
// the device) do not pay 0.16 ms per modality at 307200 rows.
class InlierIndex {
 public:
  void request() { _pending = true; }
  // a solver that is about to replace the mask AND request the list again may drop an unread request: between those two steps
  // nothing can read the list, so the skipped snapshot is unobservable
  void drop() { _pending = false; }
  bool pending() const { return _pending; }
  void flush(const std::vector<short>& mask) const { if (_pending) { indices_of_ones(mask, _idx); _pending = false; } }
  const std::vector<int>& get(const std::vector<short>& mask) const { flush(mask); return _idx; }
 private:
  mutable std::vector<int> _idx;
  mutable bool _pending = false;
};
}  // namespace rpe

#endif
