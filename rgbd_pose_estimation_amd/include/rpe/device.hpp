// rpe/device.hpp -- how the header-only solvers of pose/*.hpp reach the GPU: a small RAII layer over the C ABI
// (include/rgbd_pose_hip.h).  The host side stays C++; every device call goes through the extern "C" shim.
// There is no CPU path behind these calls: a failing status throws rpe::DeviceError (no fallback, no silent retry).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <mutex>
#include <stdexcept>
#include <utility>
#include <string>
#include <vector>
#include <type_traits>
#include "../../../include/rgbd_pose_hip.h"
#include "random.hpp"

namespace rpe {

struct DeviceError : std::runtime_error {
  int code;
  DeviceError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc, const char* where) {
  if (rc != RPE_OK) throw DeviceError(rc, std::string(where) + ": " + rpe_last_error());
}
template <class Tp> struct DType;
template <> struct DType<float> { enum { value = RPE_F32 }; };
template <> struct DType<double> { enum { value = RPE_F64 }; };

// process-wide defaults for the solver templates (the reference's free functions have no room for extra arguments); the two
// hypothesis-stream hooks at the end are per thread
// where a RANSAC / PROSAC run spent its wall time (microseconds), accumulated while Settings::profile is set
struct EngineProfile { double generate = 0, score = 0, replay = 0, mask = 0, upload = 0, sort = 0; int hypotheses = 0, batches = 0; };
struct Settings {
  int device = 0;
  bool profile = false;
  EngineProfile prof;
  // RPE_SCORE_EXACT (default): the vote kernels replay the reference's operation sequence in Tp, so consensus sets, adapted Iter and
  // inlier masks are bit-exact to the repository's CPU restatement of the reference, whose one non-trivial third-party algorithm --
  // Eigen 3.3's JacobiSVD on a 3 x 3 -- is itself a restatement of the published algorithm, unverified against a real Eigen build
  // (none exists in this image: DESIGN.md section 3, "parity unpinned"). RPE_SCORE_FAST (opt-in, RPE_SCORE_FAST=1 in the environment
  // sets the default):
  // rotation-matrix FMA form, 2.7x the scoring rate; votes can differ for correspondences within rounding of a threshold.
  int score_mode = std::getenv("RPE_SCORE_FAST") && std::getenv("RPE_SCORE_FAST")[0] == '1' ? RPE_SCORE_FAST : RPE_SCORE_EXACT;
  // RANSAC iterations generated + scored per round trip: starts at first_batch and doubles up to max_batch.  The adaptive bound
  // usually drops below a few dozen iterations as soon as one all-inlier sample has been scored, so a small first batch ends most
  // runs after ONE round (307 200 points, RPE_SCORE_EXACT: shinji_kneip_ransac 68 us with 8 against 291 us with 64; every run of
  // profiles/r02_engine_profile.txt ended in its first batch).  RPE_FIRST_BATCH overrides.
  int first_batch = std::getenv("RPE_FIRST_BATCH") ? std::atoi(std::getenv("RPE_FIRST_BATCH")) : 8, max_batch = 2048;
  bool score_session = true;   // serve a run's short batches and masks by one resident launch where the context can (RPE_SCORE_SESSION=0: never)
  // 3D-3D RANSAC (shinji_ransac / shinji_ransac2): sample + 3-point fit on the device too (rpe_ransac33_batch), bitwise the host's
  // hypotheses; false = host generation (RPE_HOST_HYPOTHESES=1 sets that default)
  bool device_hypotheses = std::getenv("RPE_HOST_HYPOTHESES") == nullptr;
  // How a solver run decides whether the HBM copy of a caller's matrix is still that matrix (the adapters hold REFERENCES and the
  // reference re-reads them on every call: pose/AOOnlyPoseAdapter.hpp:93-95, :147-152):
  //   FP_SAMPLED (default)  length + 32 cache lines at even spacing: about a microsecond per array; a refilled buffer is always caught,
  //                         an edit confined to lines that are not sampled -- NaN-marking a few columns in place -- is NOT
  //   FP_FULL               every byte hashed (RPE_FINGERPRINT=full): any in-place edit is caught; one pass over the host array per
  //                         array and solver run (3.7 MB at 640 x 480: ~0.15 ms per array on one host core, several times a solver run
  //                         on resident arrays, which is why it is opt-in)
  //   FP_OFF                address only (RPE_FINGERPRINT=off): the caller calls invalidateDevice() after every change
  enum { FP_OFF = 0, FP_SAMPLED = 1, FP_FULL = 2 };
  int fingerprint = fingerprint_from_env();
  static int fingerprint_from_env() {
    const char* e = std::getenv("RPE_FINGERPRINT");
    if (!e) return FP_SAMPLED;
    return std::strcmp(e, "full") == 0 ? FP_FULL : (std::strcmp(e, "off") == 0 ? FP_OFF : FP_SAMPLED);
  }
  // Hypothesis streams made explicit (parity tests, SURVEY.md section 8d "both sides consume the same sample list"):
  //   capture != null: the RANSAC / PROSAC engines only GENERATE -- the hypotheses of `Iter` iterations are appended to *capture,
  //                    nothing is scored and no device is touched (rpe_host_hypotheses);
  //   replay  != null: the engines take their hypotheses from *replay instead of sampling (rpe_run_replay); scoring, the
  //                    best-so-far / adaptive-Iter replay and the winner's masks run as usual.
  // first[i] .. first[i+1]: hypotheses of iteration i; 7 doubles each
  struct HypothesisList { std::vector<double> q7; std::vector<int> first; };
  // Per THREAD (the entry points that set them -- rpe_host_hypotheses, rpe_run_replay -- run the solver on the calling thread): a
  // solver running concurrently on another thread keeps sampling and scoring its own hypotheses.
  static inline thread_local HypothesisList* capture = nullptr;
  static inline thread_local const HypothesisList* replay = nullptr;
  static Settings& get() { static Settings s; return s; }
};

// What ONE solver run may set for itself without touching process-wide state: the last, defaulted argument of every *_ransac /
// *_prosac free function (the reference's signatures stay callable as they are).  The defaults are the reference's semantics -- one
// process-global random stream (rand(), /root/reference/pose/Utility.hpp:148) and the process-wide scoring mode; a caller that runs
// solvers on several threads gives each run its own stream, which is what rpe_run / rpe_host_hypotheses / ao_ransac do with their seed.
struct RunOptions {
  Rand31* rng = nullptr;   // null: rpe::global_rng()
  int score_mode = -1;     // < 0: Settings::get().score_mode
  Rand31& stream() const { return rng ? *rng : global_rng(); }
  int mode() const { return score_mode >= 0 ? score_mode : Settings::get().score_mode; }
};

inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

// One adapter's correspondence arrays resident in HBM.  Uploaded once, reused by every solver run on that adapter
// (TestMain.cpp runs seven solvers on one adapter, :186-221).
class ContextPool {
 public:
  // idle contexts are deliberately NOT destroyed at process exit: by then the HIP runtime may already be unloading
  ~ContextPool() {}
  rpe_context* acquire(int device) {
    std::lock_guard<std::mutex> g(_m);
    for (size_t i = 0; i < _idle.size(); i++)
      if (_idle[i].first == device) { rpe_context* c = _idle[i].second; _idle.erase(_idle.begin() + i); return c; }
    return nullptr;
  }
  void release(rpe_context* c, int device) {
    std::lock_guard<std::mutex> g(_m);
    if (_idle.size() < 4) _idle.emplace_back(device, c); else rpe_destroy(c);
  }
 private:
  std::mutex _m;
  std::vector<std::pair<int, rpe_context*> > _idle;
};
inline ContextPool& pool() { static ContextPool p; return p; }

// Fingerprint of a host array: length + 32 cache lines sampled at even spacing (first and last among them), hashed word by word.
// The adapters hold REFERENCES to the caller's matrices and the reference library re-reads them on every call
// (pose/AOOnlyPoseAdapter.hpp:93-95,147-152); the HBM copy is keyed by the host address, so a caller that refills the same buffer
// with the next frame must not be served the previous frame's upload.  About a microsecond per array and solver run.  A refill
// changes (nearly) every line and is always caught; an edit confined to lines that are not sampled -- NaN-marking a few columns in
// place, the reference's "invalid measurement" idiom -- is not: Settings::fingerprint = FP_FULL (RPE_FINGERPRINT=full) hashes every
// byte instead, and invalidateDevice() remains the explicit, free way to say "the matrices changed".
// full = true: EVERY byte enters (four independent multiply-rotate lanes over 8-byte words, folded at the end): about 25 GB/s on one
// host core, so that a sparse in-place edit is seen too (Settings::FP_FULL).
inline unsigned long long host_fingerprint(const void* p, size_t bytes, bool full = false) {
  unsigned long long h = 0x9E3779B97F4A7C15ull ^ (unsigned long long)bytes;
  if (!p || bytes == 0) return h;
  const unsigned char* b = static_cast<const unsigned char*>(p);
  if (full) {
    unsigned long long a[4] = {h, h ^ 0xC2B2AE3D27D4EB4Full, h ^ 0x165667B19E3779F9ull, h ^ 0x27D4EB2F165667C5ull};
    const size_t words = bytes / 8, blocks = words / 4;
    for (size_t k = 0; k < blocks; k++) {
      unsigned long long w[4];
      std::memcpy(w, b + 32 * k, 32);
      for (int i = 0; i < 4; i++) { a[i] = (a[i] ^ w[i]) * 0x9E3779B97F4A7C15ull; a[i] = (a[i] << 31) | (a[i] >> 33); }
    }
    unsigned long long tail[4] = {0, 0, 0, 0};
    std::memcpy(tail, b + 32 * blocks, bytes - 32 * blocks);
    for (int i = 0; i < 4; i++) { a[i] = (a[i] ^ tail[i]) * 0x9E3779B97F4A7C15ull; a[i] = (a[i] << 31) | (a[i] >> 33); }
    unsigned long long r = a[0];
    for (int i = 1; i < 4; i++) { r = (r ^ a[i]) * 0xFF51AFD7ED558CCDull; r ^= r >> 29; }
    return r ^ 0x1ull;   // (never the sampled value of the same content: the modes are not mixed up when the setting changes)
  }
  const size_t lines = (bytes + 63) / 64, samples = lines < 32 ? lines : 32;
  for (size_t k = 0; k < samples; k++) {
    const size_t line = samples > 1 ? k * (lines - 1) / (samples - 1) : 0;
    const size_t off = line * 64, len = bytes - off < 64 ? bytes - off : 64;
    unsigned long long w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::memcpy(w, b + off, len);
    for (int i = 0; i < 8; i++) { h ^= w[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); h *= 0xFF51AFD7ED558CCDull; }
  }
  return h;
}

class DeviceSet {
 public:
  DeviceSet() : _ctx(nullptr), _device(0), _n(0), _dtype(-1), _borrowed(false) {
    for (int i = 0; i < RPE_NUM_ARRAYS; i++) { _src[i] = nullptr; _fp[i] = 0; _fp_known[i] = false; }
    for (int i = 0; i < 3; i++) { _mask_fresh[i] = false; _weight_fresh[i] = false; }
  }
  ~DeviceSet() { if (_ctx && !_borrowed) pool().release(_ctx, _device); }
  DeviceSet(const DeviceSet&) = delete;
  DeviceSet& operator=(const DeviceSet&) = delete;

  // Contexts (stream + workspace + array storage, ~1 ms to create) are recycled through a small process-wide pool: a
  // caller that builds one adapter per frame, as Library.cpp:ao() does, pays for context creation once, not per call.
  rpe_context* ctx() {
    if (!_ctx) {
      _device = Settings::get().device;
      _ctx = pool().acquire(_device);
      if (!_ctx) check(rpe_create(&_ctx, _device, nullptr), "rpe_create");
    }
    return _ctx;
  }
  // Use a context whose arrays were PRODUCED on the device (the depth front end, pose/DepthFrontEnd.hpp): host[slot] is the
  // address of the host copy of each array (what the adapter was constructed over), so ensure() finds them resident and
  // uploads nothing.  The context stays owned by the caller and must outlive this set.
  void adopt(rpe_context* ctx, int device, int64_t n, int dtype, const void* const host[RPE_NUM_ARRAYS]) {
    if (_ctx && !_borrowed) pool().release(_ctx, _device);
    _ctx = ctx; _device = device; _n = n; _dtype = dtype; _borrowed = true;
    // produced on the device: the host side is an address only
    for (int i = 0; i < RPE_NUM_ARRAYS; i++) { _src[i] = host[i]; _fp_known[i] = false; }
    for (int i = 0; i < 3; i++) { _mask_fresh[i] = false; _weight_fresh[i] = false; }
  }
  // make sure array `slot` in HBM is the host array at `host` (3 x n of Tp)
  template <class Tp> void ensure(int slot, const Tp* host, int64_t n) {
    if (Settings::get().capture) return;   // generation only: nothing runs on the device
    if (_n != n || _dtype != (int)DType<Tp>::value) {  // first use of this set (a recycled context still holds its last frame)
      check(rpe_set_problem(ctx(), n, DType<Tp>::value), "rpe_set_problem");
      _n = n; _dtype = DType<Tp>::value;
      for (int i = 0; i < RPE_NUM_ARRAYS; i++) { _src[i] = nullptr; _fp_known[i] = false; }
      for (int i = 0; i < 3; i++) { _mask_fresh[i] = false; _weight_fresh[i] = false; }
    }
    // same address as the resident copy's source: still the same CONTENT?  (arrays adopted from the device have no host content)
    const int fpmode = Settings::get().fingerprint;
    const bool fp_full = fpmode == Settings::FP_FULL;
    const bool check_content = fpmode != Settings::FP_OFF && (!_borrowed || _fp_known[slot]);
    const unsigned long long fp = check_content ? host_fingerprint(host, (size_t)n * 3 * sizeof(Tp), fp_full) : 0;
    if (_src[slot] != (const void*)host || (check_content && _fp_known[slot] && fp != _fp[slot])) {
      const double t0 = Settings::get().profile ? now_us() : 0;
      check(rpe_upload(ctx(), slot, host), "rpe_upload");
      _src[slot] = host;
      _fp[slot] = check_content ? fp : host_fingerprint(host, (size_t)n * 3 * sizeof(Tp), fp_full);
      _fp_known[slot] = fpmode != Settings::FP_OFF;
      if (Settings::get().profile) { check(rpe_synchronize(ctx()), "rpe_synchronize"); Settings::get().prof.upload += now_us() - t0; }
    }
  }
  void upload_mask(int mod, const std::vector<short>& m) {
    if (_mask_fresh[mod]) return;
    check(rpe_upload_mask(ctx(), mod, m.data()), "rpe_upload_mask");
    _mask_fresh[mod] = true;
  }
  template <class Tp> void upload_weight(int mod, const std::vector<Tp>& w, Tp scale) {
    if (_weight_fresh[mod]) return;
    if (w.empty()) { check(rpe_upload_weight(ctx(), mod, nullptr), "rpe_upload_weight"); }
    else {
      std::vector<Tp> s(w.size());
      for (size_t i = 0; i < w.size(); i++) s[i] = w[i] / scale;
      check(rpe_upload_weight(ctx(), mod, s.data()), "rpe_upload_weight");
    }
    _weight_fresh[mod] = true;
  }
  bool mask_fresh(int mod) const { return _mask_fresh[mod]; }
  void mask_changed_on_host(int mod) { _mask_fresh[mod] = false; }
  void mask_written_on_device(int mod) { _mask_fresh[mod] = true; }
  void weight_changed_on_host(int mod) { _weight_fresh[mod] = false; }
  void download_mask(int mod, short* dst) { check(rpe_download_mask(ctx(), mod, dst), "rpe_download_mask"); }
  void download_mask(int mod, std::vector<short>& m) {
    m.resize((size_t)_n);
    check(rpe_download_mask(ctx(), mod, m.data()), "rpe_download_mask");
  }
 private:
  rpe_context* _ctx;
  int _device;
  int64_t _n;
  int _dtype;
  const void* _src[RPE_NUM_ARRAYS];
  unsigned long long _fp[RPE_NUM_ARRAYS];   // content fingerprint of the host array each resident copy was uploaded from
  bool _fp_known[RPE_NUM_ARRAYS];
  bool _mask_fresh[3], _weight_fresh[3];
  bool _borrowed;
};

// PROSAC order on the device (rpe_prosac_order): the first top_k positions of "indices by weight, descending, ties to the lower index"
// for a dense frame's weights -- the very prefix sortIndexes<float>(w, top_k) (pose/Utility.hpp) returns.  Empty result = not done
// here (other Tp, short arrays, long prefixes, capture mode, heavy ties around the cut, RPE_HOST_PROSAC=1): the caller sorts on the
// host.
template <class Tp> inline std::vector<int> device_prosac_order(DeviceSet&, const std::vector<Tp>&, int) { return std::vector<int>(); }
template <> inline std::vector<int> device_prosac_order<float>(DeviceSet& dev, const std::vector<float>& w, int top_k) {
  static const bool host_only = std::getenv("RPE_HOST_PROSAC") != nullptr;
  if (host_only || Settings::get().capture || w.size() < 65536 || top_k < 1 || top_k > 4096) return std::vector<int>();
  std::vector<int> order((size_t)std::min<size_t>((size_t)top_k, w.size()));
  const int rc = rpe_prosac_order(dev.ctx(), w.data(), (int)w.size(), top_k, order.data());
  if (rc == RPE_ERR_STATE) return std::vector<int>();   // heavy ties: host order
  check(rc, "rpe_prosac_order");
  return order;
}

}  // namespace rpe
