// extern "C" shim of librgbdpose_hip.so, Part 2 of include/rgbd_pose_hip.h: context, HBM-resident
// correspondence arrays, kernel launches, the small host-side solves.  There is NO CPU fallback: every entry
// point that computes fails with RPE_ERR_NO_DEVICE when no HIP device is usable.
#include "../../include/rgbd_pose_hip.h"
#include "rpe_kernels.h"
#include "../include/rpe/linalg.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: the RCCL entry points are resolved with dlopen/dlsym (no DT_NEEDED on librccl)

#include <cstdarg>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <limits>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>
#include <algorithm>
#include <cctype>
#include <sched.h>
#include <unistd.h>

namespace {

thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_TRY(expr)                                                                         \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(RPE_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

size_t elem_size(int dtype) { return dtype == RPE_F64 ? 8 : 4; }
inline double clock_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }

// orders the host's stores into BAR-mapped device memory (possibly write-combining): data before tags, tags out at once
inline void store_fence() {
#if defined(__x86_64__)
  __asm__ __volatile__("sfence" ::: "memory");
#else
  __sync_synchronize();
#endif
}

}  // namespace

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl r;
  if (!r.h) {
    // same soname as the copy PyTorch-ROCm bundles: if torch is in the process its librccl is reused
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.h) break; }
    if (r.h) {
      r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
      r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
      r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
      r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
      r.CommCount = (decltype(r.CommCount))dlsym(r.h, "ncclCommCount");
      r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
    }
  }
  return r;
}
#define NCCL_TRY(expr)                                                                                                  \
  do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return fail(RPE_ERR_HIP, "%s: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(r_) : "rccl error"); } while (0)
}  // namespace

namespace rpe {
// lets library.cpp (adapter-level pipelines) report through the same rpe_last_error() channel
int set_error(int code, const char* msg) { g_err = msg ? msg : ""; return code; }
}  // namespace rpe

struct rpe_context {
  int device = 0;
  hipStream_t stream = nullptr;
  // resident scoring session (rpe_score_session_begin ... _end): the grid of score_resident_kernel waits for batches in c->ctl
  struct { bool active = false; int kind = 0, mode = 0, grid = 0, runs = 0, batches = 0; double thre_3d = 0, cos_thr = 0, cos_nl = 0;
           unsigned long long base = 0, id = 0;
           double last_us = 0, wait_us = 2e6;   // host clock of the last message / the grid's bounded wait: a message that comes later
           bool pend_late = false;              // than that finds no grid -- the caller's pause, not a lost grid (nothing is counted)
           // every hypothesis the session has scored (pose as the caller gave it -> votes): the winner's total is known without
           // waiting for the masks' own record
           std::vector<double> seen_pose; std::vector<int> seen_votes;
           // the session's LAST message was "write these masks and leave" and its record has not been looked at yet (session_verify)
           bool pending = false; unsigned long long pend_tag = 0; int pend_votes = 0; double pend_pose[7] = {0, 0, 0, 0, 0, 0, 0}; } sess;
  hipStream_t stream2 = nullptr;   // the solving workgroup of the autonomous resident loops runs beside its workers (created on first use)
  hipEvent_t ev_stream2 = nullptr; // ... behind the uploads of the start pose / loop state on `stream`
  bool auto_solver = true;         // ... until the two kernels once failed to meet (a platform that serialises them)
  bool own_stream = false;
  int64_t n = 0;
  int dtype = RPE_F32;
  // ACTIVE pointers of the current problem (null = not uploaded / bound) ...
  void* arr[RPE_NUM_ARRAYS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  short* mask[3] = {nullptr, nullptr, nullptr};
  void* weight[3] = {nullptr, nullptr, nullptr};
  // ... and the storage this context owns; it survives rpe_set_problem so that a pooled context (rpe/device.hpp) serving
  // one frame after another does not pay hipMalloc/hipFree per call
  void* store[RPE_NUM_ARRAYS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap[RPE_NUM_ARRAYS] = {0, 0, 0, 0, 0};
  short* mask_store[3] = {nullptr, nullptr, nullptr};
  size_t mask_cap[3] = {0, 0, 0};
  void* weight_store[3] = {nullptr, nullptr, nullptr};
  size_t weight_cap[3] = {0, 0, 0};
  int max_blocks = 256;          // reduction kernels: cap on workgroups = one per CU (multiples of 256 only: 320 or 384 lose 20-30 %)
  int score_blocks = 2048;       // scoring / mask kernels (256-thread workgroups)
  int block = 0;                 // reduction workgroup size override (RPE_BLOCK), 0 = default
  // What is known about the CONTENT of each array, for the choice between the CLEAN flavour of the normal-equation kernels (no NaN
  // guards) and the guarded one (clean_first below): 0 unknown, 1 verified finite, 2 holds a NaN or an infinity (the reference's
  // NaN-marked "invalid measurement" columns, AOPoseAdapter.hpp:147-152).  Reset by every upload / bind / device-side producer.
  unsigned char arr_state[RPE_NUM_ARRAYS] = {0, 0, 0, 0, 0};
  bool arr_bound[RPE_NUM_ARRAYS] = {false, false, false, false, false};   // caller-owned device memory: may change between calls
  bool guard_always = false;     // RPE_GUARD_ALWAYS=1: never launch the CLEAN flavour (experiments, A/B)
  int host_cpu_request = -2;     // RPE_HOST_CPU at rpe_create: -2 none, -1 auto (rpe_tune_host_thread at the first resident refinement), >= 0 that CPU
  bool host_cpu_done = false;
  double* d_partials = nullptr;  // max_blocks * kNlLd doubles
  double* d_out = nullptr;       // 64 doubles
  double* h_out = nullptr;       // pinned + device-mapped, 64 doubles + sequence word: kernels publish straight into it
  unsigned int* d_ticket = nullptr;
  unsigned long long seq = 0;
  double* d_gn_pose = nullptr;          // device-resident Gauss-Newton: pose (12 doubles) ...
  rpe::GnState* d_gn_state = nullptr;   // ... and loop state, both in HBM
  void* d_poses = nullptr;       // kMaxScoreH * 12 doubles
  void* h_poses = nullptr;       // pinned staging
  int* d_votes = nullptr;        // kMaxScoreH ints
  int* h_votes = nullptr;        // pinned
  // optional HIP-event timing of the stage-1 normal-equation kernel (bench.py roofline leg)
  std::vector<hipEvent_t> ev0, ev1;
  size_t ev_used = 0;
  ncclComm_t comm = nullptr;      // this rank's communicator for the per-iteration all-reduce (rpe_comm_init)
  int comm_world = 1;
  unsigned long long* h_flag2 = nullptr;  // pinned sequence word of the vote publish
  unsigned long long vote_seq = 0;
  bool timing = false;
  int timing_stride = 1;
  unsigned long long timing_calls = 0;
  // peer-to-peer all-reduce over xGMI (rpe_p2p_*): own mailbox (fine-grained HBM, IPC-exported), the peers' mailboxes as
  // mapped here, the descriptor the kernels read, and the collective step counter (identical on every rank)
  unsigned long long* p2p_box = nullptr;
  void* p2p_peer[rpe::kP2PMaxWorld] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  rpe::P2PDesc* d_p2p = nullptr;
  int p2p_world = 0, p2p_rank = 0, p2p_world_saved = 0;
  unsigned long long p2p_step = 0;
  unsigned long long p2p_vote_step = 0;   // the same for the vote counters of sharded scoring
  // resident Gauss-Newton loop (rpe_gn_refine on one GPU): control block in fine-grained device memory that the HOST writes through
  // the PCIe BAR and every workgroup of the resident kernel polls (layout: rpe_residuals.hpp).  Null when the device memory is not
  // host-accessible (no large BAR): the loop then launches one kernel per iteration.
  volatile unsigned long long* ctl = nullptr;
  bool resident = false;
  int resident_lost = 0;          // resident loops that lost a granule / ended early and were finished with one launch per iteration
  int resident_cap = 0;           // workgroups of a resident kernel this device holds at once (rpe::resident_cap_device)
  bool host_resident = false;     // the HOST-driven resident loops can run here: large BAR + control block (c->ctl)
  // fault injection of the tests, set through rpe_debug_inject_resident_fault only (never from the environment)
  int test_fault_iter = 0;        // > 0: the last workgroup withholds its sums of this iteration of the next resident loops
  double test_pose_wait_s = 0;    // > 0: length of the workgroups' bounded wait for the next pose
  // pinned + mapped: tagged 16-byte pairs {value, sequence} -- the run records of collecting launches, added here on the host
  double* h_big = nullptr;
  size_t h_big_pairs = 0;
  bool collecting = false;        // the launch in flight publishes run records into h_big (collect_target)
  rpe_host_exchange* hostex = nullptr;   // host-side all-reduce between the node's rank processes (rpe_hostex_init)
  int hostex_world = 1;
  // two ranks on one GPU: no resident kernels (they would wait for each other's hosts without both being resident)
  bool hostex_shared_gpu = false;
  // PROSAC order on the device (rpe_prosac_order): scratch
  float* ps_w = nullptr; size_t ps_w_cap = 0;
  unsigned int* ps_hist = nullptr;        // 2048 + 8 uints (histogram | control words)
  unsigned long long* ps_cand = nullptr;  // kProsacSortCap keys
  int* ps_order = nullptr;                // kProsacMaxTopK + 1 ints (order | status)
  // optional host-clock profile of the resident loop (rpe_debug_loop_profile): time spent waiting for records vs the host's own turn
  bool loop_prof = false;
  double prof_wait_us = 0, prof_host_us = 0;
  long long prof_steps = 0;
  void* h_stage = nullptr;        // pinned staging for device -> host copies into caller (pageable) memory
  size_t h_stage_cap = 0;
  // front end (Part 3): the current depth frame's maps and the model it is registered against, all in HBM
  struct Frontend {
    rpe::Camera cam{}, mcam{};
    bool have_frame = false, have_model = false;
    void* d_depth = nullptr; size_t depth_cap = 0;
    float* fmap[3] = {nullptr, nullptr, nullptr};   // vertex, normal, bearing (camera frame)
    size_t fcap = 0;
    float* mmap[2] = {nullptr, nullptr};            // model vertex, normal (world frame)
    size_t mcap = 0;
    double mpose[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    int* d_count = nullptr;
  } fe;

  rpe::DeviceArrays arrays() const {
    rpe::DeviceArrays A;
    for (int i = 0; i < RPE_NUM_ARRAYS; i++) A.a[i] = arr[i];
    for (int i = 0; i < 3; i++) { A.mask[i] = mask[i]; A.weight[i] = weight[i]; }
    A.n = n; A.dtype = dtype;
    return A;
  }
};

namespace {

int ensure_mask(rpe_context* c, int mod, bool fill_ones) {
  if (c->mask[mod]) return RPE_OK;
  const size_t need = (size_t)c->n * sizeof(short);
  if (!c->mask_store[mod] || c->mask_cap[mod] < need) {
    if (c->mask_store[mod]) { HIP_TRY(hipFree(c->mask_store[mod])); c->mask_store[mod] = nullptr; c->mask_cap[mod] = 0; }
    HIP_TRY(hipMalloc((void**)&c->mask_store[mod], need ? need : 2));
    c->mask_cap[mod] = need;
  }
  c->mask[mod] = c->mask_store[mod];
  if (fill_ones && c->n)   // adapters start with all-ones masks (e.g. AOPoseAdapter.hpp:103-106): filled on the device, in stream order
    HIP_TRY(hipMemsetD16Async((hipDeviceptr_t)c->mask[mod], (unsigned short)1, (size_t)c->n, c->stream));
  return RPE_OK;
}

// Device -> caller memory.  A D2H copy into pageable memory is staged by the runtime in small pinned chunks (measured ~6 GB/s
// for a 614 KB mask); one DMA into the context's own pinned buffer followed by a host memcpy is about twice as fast.
int copy_to_host(rpe_context* c, void* dst, const void* d_src, size_t bytes) {
  if (bytes == 0) return RPE_OK;
  if (bytes > ((size_t)64 << 20)) {  // very large arrays: not worth pinning that much memory
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RPE_OK;
  }
  if (c->h_stage_cap < bytes) {
    if (c->h_stage) { HIP_TRY(hipHostFree(c->h_stage)); c->h_stage = nullptr; c->h_stage_cap = 0; }
    const size_t cap = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    HIP_TRY(hipHostMalloc(&c->h_stage, cap, hipHostMallocDefault));
    c->h_stage_cap = cap;
  }
  HIP_TRY(hipMemcpyAsync(c->h_stage, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  std::memcpy(dst, c->h_stage, bytes);
  return RPE_OK;
}

int need_arrays(rpe_context* c, std::initializer_list<int> slots) {
  if (!c) return fail(RPE_ERR_ARG, "null context");
  if (c->n <= 0) return fail(RPE_ERR_STATE, "rpe_set_problem was not called (n = %lld)", (long long)c->n);
  static const char* names[] = {"XW (points_g)", "XC (points_c)", "BV (bearingVectors)", "NW (normal_g)", "NC (normal_c)"};
  for (int s : slots) if (!c->arr[s]) return fail(RPE_ERR_STATE, "array %s was never uploaded or bound", names[s]);
  return RPE_OK;
}

// RPE_RESIDENT_STRIDE (experiments): runs of every stride-th workgroup; 0 / 1 = runs of consecutive workgroups.  Clamped to 2 .. 16: a
// stride is a number of RUNS, every run sends up to 44 sums to the host as tagged pairs, and the pinned pair buffer (h_big) and the
// autonomous loop's run records (kAutoMaxRunSums) are sized for at most ~16 runs of the widest record.
int run_stride_from_env() {
  const char* e = getenv("RPE_RESIDENT_STRIDE");
  if (!e) return 8;
  const int v = atoi(e);
  if (v <= 1) return 0;
  return v > 16 ? 16 : v;
}

rpe::ReduceTarget host_target(rpe_context* c) {
  rpe::ReduceTarget rt;
  rt.d_partials = c->d_partials; rt.d_ticket = c->d_ticket; rt.max_blocks = c->max_blocks; rt.block = c->block;
  rt.pivot_floor = rpe::pivot_floor(c->dtype == RPE_F64);
  rt.d_out = nullptr; rt.h_out = c->h_out; rt.seq = ++c->seq;
  c->collecting = false;
  return rt;
}
// host-consumed result of ONE launch on a single GPU: collecting workgroups + host-side final sum (rpe_reduce.hpp collect_and_send);
// wait_host then assembles the record in c->h_out.  RPE_COLLECT=0: the arrival-counter tail (as the device / collective targets use)
rpe::ReduceTarget collect_target(rpe_context* c) {
  rpe::ReduceTarget rt = host_target(c);
  static const bool on = !(getenv("RPE_COLLECT") && atoi(getenv("RPE_COLLECT")) == 0);
  static const int stride = run_stride_from_env();
  if (on) { rt.h_out = c->h_big; rt.rows = 1 << 20; rt.stride = stride; c->collecting = true; }
  return rt;
}
rpe::ReduceTarget device_target(rpe_context* c, double* d_out) {
  rpe::ReduceTarget rt;
  rt.d_partials = c->d_partials; rt.d_ticket = c->d_ticket; rt.max_blocks = c->max_blocks; rt.block = c->block;
  rt.pivot_floor = rpe::pivot_floor(c->dtype == RPE_F64);
  rt.d_out = d_out; rt.h_out = nullptr; rt.seq = 0;
  c->collecting = false;
  return rt;
}
// Spin on the sequence word the kernel's last workgroup stores after the record (pinned, coherent host memory).
int wait_collect(rpe_context* c, int ld);
int wait_host(rpe_context* c, int ld) {
  if (c->collecting) { c->collecting = false; return wait_collect(c, ld); }
  volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(c->h_out + ld);
  const unsigned long long want = c->seq;
  for (unsigned long long spins = 0;; spins++) {
    if (__atomic_load_n(const_cast<unsigned long long*>(flag), __ATOMIC_ACQUIRE) == want) return RPE_OK;
    if ((spins & 0xFFFFF) == 0xFFFFF) {  // every ~1M polls: has the stream died?
      hipError_t q = hipStreamQuery(c->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail(RPE_ERR_HIP, "stream error while waiting for a kernel result: %s",
          hipGetErrorString(q));
      // (an autonomous loop's result comes from its solving workgroup on the second stream: the workers' kernel ends before it does)
      if (q == hipSuccess && c->stream2) { const hipError_t q2 = hipStreamQuery(c->stream2); if (q2 == hipErrorNotReady) continue; if (q2 != hipSuccess) (void)hipGetLastError(); }
      if (q == hipSuccess && __atomic_load_n(const_cast<unsigned long long*>(flag), __ATOMIC_ACQUIRE) != want)
        return fail(RPE_ERR_HIP, "kernel finished without publishing its result (sequence %llu)", want);
    }
  }
}

// Host-side final sum (resident loop): `grid` collecting workgroups each sent `nacc` pairs {value, seq}; add them in run order as they
// arrive (a fixed order).  Records that are not there yet are waited for one by one, so the summation overlaps the arrival of the
// later ones.
constexpr int kResidentLost = -1000;   // internal (never returned through the C ABI): the resident grid lost a granule or ended early
constexpr int kResidentDirty = -1001;  // internal: the CLEAN flavour's first record was not finite -- the arrays need the guarded flavour
int wait_host_partials(rpe_context* c, int grid, int nacc, double* totals, int first_slot = 0, bool resident = false) {
  unsigned long long* pairs = reinterpret_cast<unsigned long long*>(c->h_big) + 2 * (size_t)first_slot;
  const unsigned long long want = c->seq;
  for (int k = 0; k < nacc; k++) totals[k] = 0.0;
  unsigned long long spins = 0;
  bool lost = false;
  // All tags first, in branch-free sweeps (independent loads: the cache misses on lines the device has just written overlap), then the
  // sums in run order -- 0.1 us per resident step faster than waiting pair by pair (four A/B alternations,
  // scripts/env_ab_r03.py);
  // RPE_HOST_SWEEP=0 selects the pair-by-pair wait.
  static const int sweep = getenv("RPE_HOST_SWEEP") ? atoi(getenv("RPE_HOST_SWEEP")) : 1;
  if (sweep) {
    const int total = grid * nacc;
    for (;;) {
      unsigned long long missing = 0;
      // independent loads: the misses overlap
      for (int i = 0; i < total; i++) missing |= __atomic_load_n(pairs + 2 * (size_t)i + 1, __ATOMIC_RELAXED) ^ want;
      if (!missing) break;
      if ((++spins & 0x3FFFF) == 0) {
        hipError_t q = hipStreamQuery(c->stream);
        if (q != hipSuccess && q != hipErrorNotReady) return fail(RPE_ERR_HIP, "stream error while waiting for a kernel result: %s",
            hipGetErrorString(q));
        if (q == hipSuccess) {
          missing = 0;
          for (int i = 0; i < total; i++) missing |= __atomic_load_n(pairs + 2 * (size_t)i + 1, __ATOMIC_RELAXED) ^ want;
          if (missing) { (void)fail(RPE_ERR_HIP, "the kernel ended without publishing record %llu", want);
              return resident ? kResidentLost : RPE_ERR_HIP; }
        }
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  for (int g = 0; g < grid; g++) {
    unsigned long long* rec = pairs + 2 * (size_t)g * nacc;
    for (int k = nacc - 1; k >= 0; k--) {
      while (__atomic_load_n(rec + 2 * k + 1, __ATOMIC_ACQUIRE) != want) {
        if ((++spins & 0xFFFFF) == 0) {
          hipError_t q = hipStreamQuery(c->stream);
          if (q != hipSuccess && q != hipErrorNotReady) return fail(RPE_ERR_HIP, "stream error while waiting for a kernel result: %s",
              hipGetErrorString(q));
          if (q == hipSuccess && __atomic_load_n(rec + 2 * k + 1, __ATOMIC_ACQUIRE) != want) {
            (void)fail(RPE_ERR_HIP, "the kernel ended without publishing record %llu (run %d)", want, g);
            return resident ? kResidentLost : RPE_ERR_HIP;
          }
        }
      }
    }
    for (int k = 0; k < nacc; k++) {
      double v;
      const unsigned long long w = __atomic_load_n(rec + 2 * k, __ATOMIC_RELAXED);
      if (w == rpe::kResidentLostMarker) lost = true;   // this run's collecting workgroup never got one of its granules
      std::memcpy(&v, &w, 8);
      totals[k] += v;
    }
  }
  if (lost) {
    (void)fail(RPE_ERR_HIP, "a workgroup's sums never reached its collecting workgroup (record %llu)", want);
    return resident ? kResidentLost : RPE_ERR_HIP;
  }
  return RPE_OK;
}
// the 17 structured point-to-point sums -> the packed record (same map as record_entry<1> in rpe_reduce.hpp)
void expand_p2p17(const double* t, double* ne) {
  for (int i = 0; i < 32; i++) ne[i] = 0.0;
  const double nn = t[0], Sx = t[1], Sy = t[2], Sz = t[3], xx = t[4], xy = t[5], xz = t[6], yy = t[7], yz = t[8], zz = t[9];
  ne[0] = ne[6] = ne[11] = ne[28] = nn;
  ne[4] = Sz; ne[5] = -Sy; ne[8] = -Sz; ne[10] = Sx; ne[12] = Sy; ne[13] = -Sx;
  ne[15] = yy + zz; ne[16] = -xy; ne[17] = -xz; ne[18] = xx + zz; ne[19] = -yz; ne[20] = xx + yy;
  for (int i = 21; i <= 26; i++) ne[i] = t[i - 11];
  ne[27] = t[16];
}

// Result of a collecting launch (collect_target): the header pair says how many run records of how many sums to expect; add them in
// run order and lay the record out in c->h_out as the flag path would have left it.
int wait_collect(rpe_context* c, int ld) {
  unsigned long long* pairs = reinterpret_cast<unsigned long long*>(c->h_big);
  const unsigned long long want = c->seq;
  for (unsigned long long spins = 1; __atomic_load_n(pairs + 1, __ATOMIC_ACQUIRE) != want; spins++) {
    if ((spins & 0xFFFFF) == 0) {
      hipError_t q = hipStreamQuery(c->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail(RPE_ERR_HIP, "stream error while waiting for a kernel result: %s",
          hipGetErrorString(q));
      if (q == hipSuccess && __atomic_load_n(pairs + 1, __ATOMIC_ACQUIRE) != want)
        return fail(RPE_ERR_HIP,
            "kernel finished without publishing its result (sequence %llu; header %llx %llu, first pair %llx %llu)", want,
                    pairs[0], pairs[1], pairs[2], pairs[3]);
    }
  }
  const unsigned long long hdr = __atomic_load_n(pairs, __ATOMIC_RELAXED);
  const int runs = (int)(hdr & 0xFFFF), nacc = (int)((hdr >> 16) & 0xFF), mode = (int)((hdr >> 24) & 0xFF);
  if (runs < 1 || nacc < 1 || nacc > 64 || nacc > ld || (size_t)(1 + runs * nacc) > c->h_big_pairs) return fail(RPE_ERR_HIP,
      "malformed result header (%d runs of %d sums)", runs, nacc);
  double tot[64];
  int rc = wait_host_partials(c, runs, nacc, tot, 1);
  if (rc) return rc;
  if (mode == 1) expand_p2p17(tot, c->h_out);
  else { for (int i = 0; i < ld; i++) c->h_out[i] = i < nacc ? tot[i] : 0.0; }
  return RPE_OK;
}

// same spin on an arbitrary pinned sequence word
int wait_flag(rpe_context* c, unsigned long long* flag, unsigned long long want) {
  for (unsigned long long spins = 0;; spins++) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == want) return RPE_OK;
    if ((spins & 0xFFFFF) == 0xFFFFF) {
      hipError_t q = hipStreamQuery(c->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail(RPE_ERR_HIP, "stream error while waiting for a kernel result: %s",
          hipGetErrorString(q));
      if (q == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) return fail(RPE_ERR_HIP,
          "kernel finished without publishing its result (sequence %llu)", want);
    }
  }
}

// ---- CLEAN-first protocol.  The CLEAN flavour of a normal-equation kernel carries no NaN guards (17 % fewer instructions per group,
// rpe_residuals.hpp pair_group); it is exact for arrays whose values are all finite, and for any other content at least one sum of its
// record is non-finite (a NaN or an infinity anywhere multiplies into the sums even at weight 0).  So a launch whose record the host
// reads anyway takes the CLEAN flavour first, looks at the record, and repeats the launch in the guarded flavour if it is not finite --
// one wasted launch per upload of NaN-marked arrays, after which the arrays are known to need the guards.  A launch whose record is
// consumed on the device (collectives, the autonomous loops) takes the CLEAN flavour only over arrays already verified.
enum { kArrUnknown = 0, kArrClean = 1, kArrDirty = 2 };
unsigned kind_slot_bits(int kind) {
  switch (kind) {
    case RPE_RES_P2P: return (1u << RPE_XW) | (1u << RPE_XC);
    case RPE_RES_P2PLANE: return (1u << RPE_XW) | (1u << RPE_XC) | (1u << RPE_NC);
    case RPE_RES_BEARING: case RPE_RES_REPROJ: return (1u << RPE_XW) | (1u << RPE_BV);
    case RPE_RES_NORMAL: return (1u << RPE_NW) | (1u << RPE_NC);
  }
  return 0;
}
bool take_clean(const rpe_context* c, int kind, bool host_verifies) {
  const unsigned bits = kind_slot_bits(kind);
  if (c->guard_always || bits == 0 || c->dtype == RPE_F64) return false;   // (the CLEAN flavours exist for fp32 arrays)
  bool all_verified = true;
  for (int s = 0; s < RPE_NUM_ARRAYS; s++) if (bits & (1u << s)) {
    if (c->arr_state[s] == kArrDirty) return false;
    if (c->arr_state[s] != kArrClean) all_verified = false;
  }
  return host_verifies || all_verified;
}
bool record_finite(const double* rec, int count) {
  double s = 0.0;
  for (int i = 0; i < count; i++) s += rec[i];
  return std::isfinite(s);
}
// what a CLEAN launch's record said about the arrays of `kind`.  Caller-owned (bound) arrays are never promoted: they may change
// between calls without the context hearing of it.
void note_clean_launch(rpe_context* c, int kind, bool finite) {
  const unsigned bits = kind_slot_bits(kind);
  for (int s = 0; s < RPE_NUM_ARRAYS; s++) if (bits & (1u << s)) {
    if (!finite) c->arr_state[s] = kArrDirty;
    else if (!c->arr_bound[s]) c->arr_state[s] = kArrClean;
  }
}
// ... for a SET of residual kinds (the joint kernels; bits = 1 << kind): CLEAN only if every kind of the set may take it
bool take_clean_terms(const rpe_context* c, int bits, bool host_verifies) {
  if (!rpe::joint_has_clean_flavour(c->dtype == RPE_F64 ? 1 : 0, bits)) return false;   // (the launch would run guarded: its finite record says nothing about the arrays)
  for (int k = 0; k <= 4; k++) if ((bits & (1 << k)) && !take_clean(c, k, host_verifies)) return false;
  return bits != 0;
}
void note_clean_terms(rpe_context* c, int bits, bool finite) {
  for (int k = 0; k <= 4; k++) if (bits & (1 << k)) note_clean_launch(c, k, finite);
}
void arrays_changed(rpe_context* c, int slot, bool bound) { c->arr_state[slot] = kArrUnknown; c->arr_bound[slot] = bound; }

int kind_arrays(rpe_context* c, int kind) {
  switch (kind) {
    case RPE_RES_P2P: return need_arrays(c, {RPE_XW, RPE_XC});
    case RPE_RES_P2PLANE: return need_arrays(c, {RPE_XW, RPE_XC, RPE_NC});
    case RPE_RES_BEARING: case RPE_RES_REPROJ: return need_arrays(c, {RPE_XW, RPE_BV});
    case RPE_RES_NORMAL: return need_arrays(c, {RPE_XW, RPE_NW, RPE_NC});
  }
  return fail(RPE_ERR_ARG, "unknown residual kind %d", kind);
}

int check_flags(rpe_context* c, int kind, int flags) {
  const int mod = (kind == RPE_RES_BEARING || kind == RPE_RES_REPROJ) ? RPE_MOD_23 : (kind == RPE_RES_NORMAL ? RPE_MOD_NN : RPE_MOD_33);
  if ((flags & RPE_USE_MASK) && !c->mask[mod]) return fail(RPE_ERR_STATE, "RPE_USE_MASK but no mask for modality %d", mod);
  if ((flags & RPE_USE_WEIGHT) && !c->weight[mod]) return fail(RPE_ERR_STATE, "RPE_USE_WEIGHT but no weight for modality %d", mod);
  return RPE_OK;
}

}  // namespace

// The exact 3D test is  sqrt(s) < thre_3d  in the array dtype (Eigen norm(), AbsoluteOrientation.hpp:137-138).  The correctly rounded
// square root is monotonic, so the set of s that pass is { s < cut } with cut = the smallest value whose square root reaches the
// threshold; the kernels compare s with `cut` and never take the root.  Found by stepping from thr^2 with the host's own sqrt.
template <class T> static T sqrt_cut(T thr) {
  if (thr != thr) return thr;                                   // NaN: nothing passes, either way
  if (!(thr > T(0))) return T(0);                               // sqrt(s) < thr <= 0 never holds; s < 0 never holds
  if (std::isinf(thr)) return thr;                              // every finite s passes
  T x = thr * thr;
  if (std::isinf(x)) x = std::numeric_limits<T>::max();
  while (x > T(0) && std::sqrt(x) >= thr) x = std::nextafter(x, T(0));
  while (std::sqrt(x) < thr) x = std::nextafter(x, std::numeric_limits<T>::infinity());
  return x;
}

// Host side of a RESIDENT loop (rpe_gn_refine, rpe_icp): ONE launch (`launch(rt, base)`) whose grid stays resident; the host hands
// every
// pose to it through the control block in device memory (two stores' worth of PCIe latency instead of a kernel launch per iteration),
// receives the run records of every iteration, adds them, solves the 6x6 system and applies the SE(3) update, as the one-launch-per-
// iteration loop does.  Pose i carries tag base + i, the records of iteration i carry sequence base + i.
// Cross-workgroup stage: runs of `rows` workgroups are added by the first workgroup of the run (granule hand-off, one hop), the run
// records come to the host, which adds them in run order.  A handful of small records (grid x sums <= 1024 pairs, i.e. a few thousand
// correspondences): rows = 1, every workgroup sends its own record and nothing is handed over on the GPU at all; otherwise one run
// per XCD (eight run records: 136 pairs for point-to-point at 640 x 480), or -- small grids, RPE_RESIDENT_STRIDE=0 -- runs of
// consecutive workgroups, one granule per collecting thread and up to four when that keeps the number of runs at <= 8
// (resident_run_shape).
// One resident loop per GPU at a time within this process: two resident grids launched together (two contexts, two threads) could each
// get only part of their workgroups onto the CUs and then wait for workgroups that cannot start (the bounded waits would end both with
// an error).  Other PROCESSES on the same GPU are the caller's to serialise.
// (a lock that may be given back by another thread than the one that took it: a scoring session holds the device's resident slot
// from rpe_score_session_begin to whatever call ends it, and a context may be handed from one thread to the next in between --
// std::mutex forbids that)
struct ResidentSlot {
  std::mutex m; std::condition_variable cv; bool busy = false;
  void lock() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return !busy; }); busy = true; }
  bool try_lock() { std::lock_guard<std::mutex> lk(m); if (busy) return false; busy = true; return true; }
  void unlock() { { std::lock_guard<std::mutex> lk(m); busy = false; } cv.notify_one(); }
};
static ResidentSlot& resident_mutex(int device) {
  static ResidentSlot m[64];
  return m[device >= 0 && device < 64 ? device : 0];
}

// Run shape of a resident grid (resident_host_loop and the autonomous launches use the same one, so their run records are the same):
// tiny problems send every workgroup's record (rows 1); grids of 32 workgroups and more are collected per XCD -- run r = workgroups r,
// r + 8, ... (rpe_residuals.hpp run_shape; RPE_RESIDENT_STRIDE=0 keeps runs of consecutive workgroups); RPE_RESIDENT_ROWS forces a
// run length of consecutive workgroups.  Returns the number of runs.
static int resident_run_shape(int grid, int nacc, int max_rows, int rows_auto, rpe::ReduceTarget* rt) {
  static const int env_rows = getenv("RPE_RESIDENT_ROWS") ? atoi(getenv("RPE_RESIDENT_ROWS")) : 0;
  static const int env_stride = run_stride_from_env();
  rt->stride = 0;
  if (env_rows >= 1) rt->rows = std::min(env_rows, max_rows);
  else if (grid * nacc <= 1024) rt->rows = 1;
  else if (env_stride > 1 && grid >= 4 * env_stride && (grid + env_stride - 1) / env_stride <= max_rows) {
    rt->stride = env_stride; rt->rows = (grid + env_stride - 1) / env_stride;
    return env_stride;
  } else rt->rows = rows_auto;
  return (grid + rt->rows - 1) / rt->rows;
}

// A resident grid of this context was lost (not all of it on the compute units at once, or a workgroup held up beyond its bounded
// wait) and the refinement was finished with one launch per iteration.  The second loss switches resident loops off for the context,
// host-driven and autonomous alike (rpe_debug_resident_state reports enabled = 0 from then on).
static void note_lost_grid(rpe_context* c) {
  if (++c->resident_lost >= 2) c->resident = false;
}

template <class Launch>
static int resident_host_loop(rpe_context* c, Launch launch, int grid, int nacc, int max_rows, int rows_auto, double cost_scale,
                              double* pose12, int max_iter, double tol, int* it_out, double* step_out, double* cost_out, double* weight_out,
                              const char* what, bool clean = false, bool* first_record_finite = nullptr) {
  const unsigned long long base = c->seq;
  auto hand_over = [&](const double* p, unsigned long long tag) {
    if (p) for (int k = 0; k < 12; k++) { unsigned long long w; std::memcpy(&w, &p[k], 8); c->ctl[1 + k] = w; }   // words 1..7 | 8..12
    store_fence();
    c->ctl[0] = tag; c->ctl[15] = tag;
    store_fence();
  };
  hand_over(pose12, base + 1);
  rpe::ReduceTarget rt = host_target(c);
  rt.seq = base;
  // a rank that waits for a slow peer inside the host-side exchange (up to its 10 s) must not lose its own grid meanwhile
  if (c->hostex) rt.pose_wait_ticks = 1200000000ull;
  // tests (rpe_debug_inject_resident_fault): a long pose wait, to see that a lost grid is RELEASED rather than timed out
  if (c->test_pose_wait_s > 0) rt.pose_wait_ticks = (unsigned long long)(c->test_pose_wait_s * 1e8);
  if (c->test_fault_iter >= 1 && c->test_fault_iter <= max_iter) rt.fault_tag = base + (unsigned long long)c->test_fault_iter;
  const int runs = resident_run_shape(grid, nacc, max_rows, rows_auto, &rt);
  rt.h_out = c->h_big;
  rt.clean = clean;   // normal-equation kernels: the flavour without NaN guards; its FIRST record is checked below
  c->seq = base;
  {
    const hipError_t e = launch(rt, base);
    if (e != hipSuccess) return fail(RPE_ERR_HIP, "resident launch: %s", hipGetErrorString(e));
  }
  int status = RPE_OK, received = 0, it = 0, rc;   // records received so far = poses the grid has consumed
  double step = 0, cost = 0, weight = 0;
  double tp = c->loop_prof ? clock_us() : 0;
  for (;;) {
    c->seq = base + (unsigned long long)received + 1;
    double ne[32], d[6];
    double tot[32];
    if ((rc = wait_host_partials(c, runs, nacc, tot, 0, true))) { status = rc; break; }
    if (nacc == 17) expand_p2p17(tot, ne); else { for (int i = 0; i < 32; i++) ne[i] = i < nacc ? tot[i] : 0.0; }
    // CLEAN flavour: a NaN or an infinity anywhere in the arrays shows in the very first record (before any pose update could
    // produce one): stop the grid; the caller repeats the refinement with the guarded flavour, from the same start pose
    if (clean && received == 0 && !record_finite(ne, 29)) { status = kResidentDirty; received++; break; }
    if (clean && received == 0 && first_record_finite) *first_record_finite = true;   // only THIS vouches for the arrays' content
    if (c->hostex && (rc = rpe_host_exchange_allreduce_f64(c->hostex, ne, 32))) { status = rc; received++; break; }
    received++;
    if (c->loop_prof) { const double t = clock_us(); if (received > 1) { c->prof_wait_us += t - tp; c->prof_steps++; } tp = t; }
    cost = cost_scale * ne[27]; weight = ne[28];
    if (!rpe::solve_normal_eq6(ne, d, rpe::pivot_floor(c->dtype == RPE_F64))) { status = fail(RPE_ERR_DEGENERATE,
        "%s are not positive definite at iteration %d (weight sum %g)", what, it, ne[28]); break; }
    rpe::se3_left_update(d, pose12);
    step = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    it = received;
    if (step < tol || received == max_iter) break;
    hand_over(pose12, base + (unsigned long long)received + 1);
    if (c->loop_prof) { const double t = clock_us(); c->prof_host_us += t - tp; tp = t; }
  }
  // the grid is still waiting: release it
  // Release a grid that is still waiting.  After an early stop every workgroup waits for pose received + 1.  After a LOST grid the
  // workgroups that delivered their sums of the unfinished iteration already wait for pose received + 2: stop with that number -- a
  // workgroup still waiting for received + 1 leaves on it too (a larger tag means "this launch is over", resident_wait_pose).
  if (status == kResidentLost) hand_over(nullptr, (base + (unsigned long long)received + 2) | rpe::kResidentStopBit);
  else if (received < max_iter) hand_over(nullptr, (base + (unsigned long long)received + 1) | rpe::kResidentStopBit);
  c->seq = base + (unsigned long long)max_iter + 1;   // stays ahead of every tag / sequence value this launch could use
  *it_out = it; *step_out = step; *cost_out = cost; *weight_out = weight;
  if (status == kResidentLost) {
    // Not all of the grid was on the compute units at once (another process on the GPU, a smaller partition than the occupancy query
    // promised) or a workgroup was held up for more than its bounded wait.  pose12 holds the pose after `it` whole iterations: the
    // caller finishes with one launch per iteration.  A context that sees this twice stops using resident loops.
    note_lost_grid(c);
  }
  return status;
}

// ---- which CPUs, and pinning the calling thread (rpe_tune_host_thread, RPE_HOST_CPU)
namespace {
std::vector<int> parse_cpulist(const char* path) {
  std::vector<int> out;
  FILE* f = std::fopen(path, "r");
  if (!f) return out;
  char buf[4096];
  if (std::fgets(buf, sizeof buf, f)) {
    for (char* p = buf; *p;) {
      while (*p && !std::isdigit((unsigned char)*p)) p++;
      if (!*p) break;
      const long lo = std::strtol(p, &p, 10);
      long hi = lo;
      if (*p == '-') hi = std::strtol(p + 1, &p, 10);
      for (long v = lo; v <= hi && v < 4096; v++) out.push_back((int)v);
    }
  }
  std::fclose(f);
  return out;
}
bool pin_calling_thread(int cpu) {
  cpu_set_t set;
  CPU_ZERO(&set);
  CPU_SET(cpu, &set);
  return sched_setaffinity(0, sizeof set, &set) == 0;   // pid 0: the calling thread
}
}  // namespace

// ---- resident scoring session (K4r, rpe_score.hip): ONE launch serves the batches of a RANSAC run and the winner's masks.
// Which session did THIS thread open?  By number, not by pointer: a context may be handed to another thread, which may end the session
// (or destroy the context) without this thread hearing of it.  The open session of a device -- there is at most one: it holds the
// resident slot -- is registered with its number; a thread that finds its own number still registered knows the context is alive.
struct OpenSession { std::mutex m; rpe_context* ctx = nullptr; unsigned long long id = 0; };
static OpenSession& open_session(int device) {
  static OpenSession o[64];
  return o[device >= 0 && device < 64 ? device : 0];
}
static std::atomic<unsigned long long> g_session_ids{0};
static thread_local unsigned long long t_session_id = 0;
static thread_local int t_session_dev = -1;
static void session_close(rpe_context* c);
// the session this thread holds open on `device`, if it still is one (and forgets it otherwise)
static rpe_context* my_open_session(int device) {
  if (t_session_id == 0 || t_session_dev != device) return nullptr;
  OpenSession& o = open_session(device);
  rpe_context* ctx = nullptr;
  { std::lock_guard<std::mutex> lk(o.m); if (o.id == t_session_id) ctx = o.ctx; }
  if (!ctx) t_session_id = 0;
  return ctx;
}
static void session_registered(rpe_context* c, unsigned long long id) {
  OpenSession& o = open_session(c->device);
  { std::lock_guard<std::mutex> lk(o.m); o.ctx = c; o.id = id; }
  t_session_id = id; t_session_dev = c->device;
}
static void session_unregistered(rpe_context* c, unsigned long long id) {
  OpenSession& o = open_session(c->device);
  { std::lock_guard<std::mutex> lk(o.m); if (o.id == id) { o.ctx = nullptr; o.id = 0; } }
  if (t_session_id == id) t_session_id = 0;
}
static int mask_by_launch(rpe_context* c, int kind, int mode, const double* pose7, double thre_3d, double cos_thr, double cos_nl, int* votes_out);
static void session_message(rpe_context* c, int op, const void* staged, int count, size_t bytes, unsigned long long tag) {
  const size_t words = (bytes + 7) / 8;
  unsigned long long buf[rpe::kSessionCtlWordsMax];
  if (words) { buf[words - 1] = 0; std::memcpy(buf, staged, bytes); }
  for (size_t k = 0; k < words; k++) c->ctl[2 + k] = buf[k];
  c->ctl[1] = (unsigned long long)(unsigned int)count | ((unsigned long long)op << 32);
  store_fence();
  c->ctl[0] = tag; c->ctl[rpe::kSessionCtlWordsMax - 1] = tag;
  store_fence();
}
// Closes a session that is open (idempotent): the stop message releases the grid, the per-device resident slot is given back.
static void session_close(rpe_context* c) {
  if (!c || !c->sess.active) return;
  session_unregistered(c, c->sess.id);
  c->sess.active = false;
  session_message(c, 2, nullptr, 0, 0, (c->sess.base + (unsigned long long)c->sess.batches + 1) | rpe::kResidentStopBit);
  c->seq = c->sess.base + (unsigned long long)c->sess.batches + 2;   // stays ahead of every tag / sequence value the launch could use
  resident_mutex(c->device).unlock();
}
// The masks of a session's last message ("write them and leave", session_final_masks) were not waited for.  Look at their record now
// -- it has long arrived -- and, should the grid have gone away before it consumed the message (a host stalled beyond the grid's
// bounded wait), write the masks with the one-launch kernel: either way they are in place, in stream order, for whoever reads them.
static void session_verify(rpe_context* c) {
  if (!c || !c->sess.pending) return;
  c->sess.pending = false;
  const unsigned long long keep = c->seq;
  c->seq = c->sess.pend_tag;
  double tot[rpe::kSessionHypsMax];
  const int rc = wait_host_partials(c, c->sess.runs, rpe::kSessionHypsMax, tot, 0, true);
  c->seq = keep;
  if (rc == RPE_OK && (int)tot[0] == c->sess.pend_votes) return;
  if (rc == kResidentLost && !c->sess.pend_late) note_lost_grid(c);
  int votes = 0;
  (void)hipSetDevice(c->device);   // (the callers set the device after their session_end)
  (void)mask_by_launch(c, c->sess.kind, c->sess.mode, c->sess.pend_pose, c->sess.thre_3d, c->sess.cos_thr, c->sess.cos_nl, &votes);
}
// Every entry point that queues work behind the context's stream, reads the masks or reuses the host-side record area calls this first.
static void session_end(rpe_context* c) {
  // (a session of ANOTHER context of this thread on the same GPU holds the device's resident slot: a resident loop of `c` would wait
  // for it forever)
  if (c) { rpe_context* mine = my_open_session(c->device); if (mine && mine != c) session_close(mine); }
  session_close(c);
  session_verify(c);
}
// one batch through the open session: op 0 = score `count` hypotheses (staged: the kernel's layout, values of the array dtype), op 1 =
// the masks of one; totals = the 32 sums of the batch.  On a failure the session is closed and the caller takes the launch path.
static int session_batch(rpe_context* c, int op, const void* staged, int count, size_t bytes, double* totals) {
  const unsigned long long tag = c->sess.base + (unsigned long long)(c->sess.batches + 1);
  const double now = clock_us();
  const bool late = now - c->sess.last_us > 0.8 * c->sess.wait_us;
  c->sess.last_us = now;
  session_message(c, op, staged, count, bytes, tag);
  c->sess.batches++;
  c->seq = tag;
  const int rc = wait_host_partials(c, c->sess.runs, rpe::kSessionHypsMax, totals, 0, true);
  if (rc != RPE_OK) { if (rc == kResidentLost && !late) note_lost_grid(c); session_close(c); return rc == kResidentLost ? RPE_ERR_HIP : rc; }
  return RPE_OK;
}
// The session's LAST message: the masks of a hypothesis whose vote total is already known (it was scored in this session), together
// with the stop.  Nothing is waited for: the grid writes the masks, sends their record and leaves on its own; the context's stream
// orders every later reader behind it, and session_verify looks at the record at the next call.
static void session_final_masks(rpe_context* c, const void* staged, size_t bytes, const double* pose7, int votes) {
  const unsigned long long tag = c->sess.base + (unsigned long long)(c->sess.batches + 1);
  c->sess.pend_late = clock_us() - c->sess.last_us > 0.8 * c->sess.wait_us;
  session_message(c, 1, staged, 1, bytes, tag | rpe::kResidentStopBit);
  c->sess.batches++;
  session_unregistered(c, c->sess.id);
  c->sess.active = false;
  c->sess.pending = true; c->sess.pend_tag = tag; c->sess.pend_votes = votes;
  std::memcpy(c->sess.pend_pose, pose7, sizeof c->sess.pend_pose);
  c->seq = tag + 1;
  resident_mutex(c->device).unlock();   // (the grid waits for nobody any more: another resident grid may start beside it)
}
static bool session_seen(const rpe_context* c, const double* pose7, int* votes) {
  const size_t count = c->sess.seen_votes.size();
  for (size_t i = count; i-- > 0;)   // (the winner is usually among the latest)
    if (std::memcmp(&c->sess.seen_pose[7 * i], pose7, 7 * sizeof(double)) == 0) { *votes = c->sess.seen_votes[i]; return true; }
  return false;
}
static bool session_matches(const rpe_context* c, int kind, int mode, double thre_3d, double cos_thr, double cos_nl) {
  return c->sess.active && c->sess.kind == kind && c->sess.mode == mode && c->sess.thre_3d == thre_3d && c->sess.cos_thr == cos_thr &&
         c->sess.cos_nl == cos_nl;
}

extern "C" {

int rpe_abi_version(void) { return 1; }
const char* rpe_last_error(void) { return g_err.c_str(); }

int rpe_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

int rpe_create(rpe_context** out, int device, void* stream) {
  if (!out) return fail(RPE_ERR_ARG, "null out");
  *out = nullptr;
  const int nd = rpe_device_count();
  if (nd <= 0) return fail(RPE_ERR_NO_DEVICE, "no HIP device is visible; librgbdpose_hip has no CPU fallback");
  if (device < 0 || device >= nd) return fail(RPE_ERR_ARG, "device %d out of range (have %d)", device, nd);
  HIP_TRY(hipSetDevice(device));
  rpe_context* c = new rpe_context();
  c->device = device;
  if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
  else { hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); if (e != hipSuccess) { delete c;
      return fail(RPE_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); } c->own_stream = true; }
  if (const char* mb = getenv("RPE_MAX_BLOCKS")) { int v = atoi(mb); if (v >= 1 && v <= 4096) c->max_blocks = v; }
  if (const char* mb = getenv("RPE_SCORE_BLOCKS")) { int v = atoi(mb); if (v >= 1 && v <= 65535) c->score_blocks = v; }
  if (const char* mb = getenv("RPE_BLOCK")) { int v = atoi(mb); if (v == 256 || v == 512) c->block = v; }
  if (const char* f = getenv("RPE_GUARD_ALWAYS")) c->guard_always = atoi(f) != 0;
  if (const char* f = getenv("RPE_HOST_CPU")) c->host_cpu_request = std::strcmp(f, "auto") == 0 ? -1 : (std::isdigit((unsigned char)f[0]) ? atoi(f) : -2);
  hipError_t e = hipSuccess;
  // scratch of the cross-workgroup stages, whichever layout a launch uses: (4096 + 8 shard) records of kNlLd doubles, or 16-byte
  // granules [workgroup <= 4096][sums <= 44] followed by the autonomous loop's run records [2 parities][<= kAutoMaxRunSums = 1024]
  const size_t partial_doubles = std::max<size_t>((size_t)(4096 + 8) * rpe::kNlLd, (size_t)2 * 4096 * 44 + (size_t)2 * 2 * 1024);
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_partials, partial_doubles * sizeof(double));
  // granule tags start below every sequence value
  if (e == hipSuccess) e = hipMemset(c->d_partials, 0, partial_doubles * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_out, 64 * sizeof(double));
  c->h_big_pairs = 8192 + 64;
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_big, c->h_big_pairs * 16, hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) std::memset(c->h_big, 0, c->h_big_pairs * 16);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_out, 80 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) { std::memset(c->h_out, 0, 80 * sizeof(double)); e = hipMalloc((void**)&c->d_ticket, 9 * 128); }
  if (e == hipSuccess) e = hipMemset(c->d_ticket, 0, 9 * 128);
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_gn_pose, 16 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_gn_state, sizeof(rpe::GnState));
  if (e == hipSuccess) e = hipMalloc(&c->d_poses, (size_t)rpe::kMaxScoreH * 12 * sizeof(double));
  // staging for pose uploads; also written directly by the hypothesis generator
  if (e == hipSuccess) e = hipHostMalloc(&c->h_poses, (size_t)rpe::kMaxScoreH * 12 * sizeof(double),
      hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_votes, (size_t)rpe::kMaxScoreH * sizeof(int));
  // the scoring kernels accumulate into zeroed counters
  if (e == hipSuccess) e = hipMemset(c->d_votes, 0, (size_t)rpe::kMaxScoreH * sizeof(int));
  // pinned + device-mapped: the vote read-out kernel stores straight into it; the sequence word sits behind the counters
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_votes, ((size_t)rpe::kMaxScoreH + 4) * sizeof(int),
      hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) { std::memset(c->h_votes, 0, ((size_t)rpe::kMaxScoreH + 4) * sizeof(int));
      c->h_flag2 = reinterpret_cast<unsigned long long*>(c->h_votes + rpe::kMaxScoreH); }
  // PROSAC order scratch (rpe_prosac_order): histogram + control words (zero between calls), candidate keys, order + status
  if (e == hipSuccess) e = hipMalloc((void**)&c->ps_hist, (2048 + 8) * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMemset(c->ps_hist, 0, (2048 + 8) * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMalloc((void**)&c->ps_cand, (size_t)rpe::kProsacSortCap * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc((void**)&c->ps_order, ((size_t)rpe::kProsacMaxTopK + 1) * sizeof(int));
  if (e != hipSuccess) { rpe_destroy(c); return fail(RPE_ERR_HIP, "workspace allocation: %s", hipGetErrorString(e)); }
  {  // Resident loops.  The co-residency cap is a property of the device (0: not even one workgroup of the resident kernels per
     // compute unit) and gates both forms; the AUTONOMOUS form (rpe_gn_refine_device, device_resident ICP) needs nothing else.  The
     // HOST-driven form also needs device memory the CPU can store into (large BAR: the control block); RPE_RESIDENT=0 switches that
     // form off and leaves the autonomous one alone (RPE_DEVICE_LOOP_RESIDENT=0 is its switch).
    c->resident_cap = rpe::resident_cap_device();
    c->resident = c->resident_cap >= 1;
    int large_bar = 0;
    const char* env = getenv("RPE_RESIDENT");
    if (c->resident && !(env && env[0] == '0') && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) == hipSuccess
        && large_bar) {
      void* p = nullptr;
      if (hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained) == hipSuccess && hipMemset(p, 0, 4096) == hipSuccess &&
          hipDeviceSynchronize() == hipSuccess) {
        c->ctl = (volatile unsigned long long*)p;
        c->host_resident = true;
      } else { (void)hipGetLastError(); if (p) (void)hipFree(p); }
    } else (void)hipGetLastError();
  }
  {  // first context on this device: load every kernel unit's code object now, not at the first launch out of each
    static std::mutex m;
    static bool loaded[64];
    std::lock_guard<std::mutex> g(m);
    if (device < 64 && !loaded[device]) {
      rpe::preload_normal_eq(); rpe::preload_icp(); rpe::preload_joint(); rpe::preload_score(); rpe::preload_nl();
      rpe::preload_frontend(); rpe::preload_hypotheses(); rpe::preload_prosac();
      loaded[device] = true;
    }
  }
  *out = c;
  return RPE_OK;
}

void rpe_destroy(rpe_context* c) {
  session_end(c);
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (int i = 0; i < RPE_NUM_ARRAYS; i++) if (c->store[i]) (void)hipFree(c->store[i]);
  for (int i = 0; i < 3; i++) { if (c->mask_store[i]) (void)hipFree(c->mask_store[i]);
      if (c->weight_store[i]) (void)hipFree(c->weight_store[i]); }
  if (c->d_partials) (void)hipFree(c->d_partials);
  if (c->d_out) (void)hipFree(c->d_out);
  if (c->d_ticket) (void)hipFree(c->d_ticket);
  if (c->d_gn_pose) (void)hipFree(c->d_gn_pose);
  if (c->d_gn_state) (void)hipFree(c->d_gn_state);
  if (c->h_out) (void)hipHostFree(c->h_out);
  if (c->d_poses) (void)hipFree(c->d_poses);
  if (c->h_poses) (void)hipHostFree(c->h_poses);
  if (c->d_votes) (void)hipFree(c->d_votes);
  if (c->h_votes) (void)hipHostFree(c->h_votes);
  (void)rpe_p2p_destroy(c);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->ctl) (void)hipFree((void*)c->ctl);
  if (c->h_big) (void)hipHostFree(c->h_big);
  if (c->hostex) rpe_host_exchange_close(c->hostex);
  if (c->ps_w) (void)hipFree(c->ps_w);
  if (c->ps_hist) (void)hipFree(c->ps_hist);
  if (c->ps_cand) (void)hipFree(c->ps_cand);
  if (c->ps_order) (void)hipFree(c->ps_order);
  if (c->fe.d_depth) (void)hipFree(c->fe.d_depth);
  for (float* m : c->fe.fmap) if (m) (void)hipFree(m);
  for (float* m : c->fe.mmap) if (m) (void)hipFree(m);
  if (c->fe.d_count) (void)hipFree(c->fe.d_count);
  if (c->comm && rccl().ok) { (void)rccl().CommDestroy(c->comm); c->comm = nullptr; }
  for (hipEvent_t e : c->ev0) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev1) (void)hipEventDestroy(e);
  if (c->ev_stream2) (void)hipEventDestroy(c->ev_stream2);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int rpe_synchronize(rpe_context* c) {
  session_end(c);
  if (!c) return fail(RPE_ERR_ARG, "null context");
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RPE_OK;
}

int rpe_set_problem(rpe_context* c, int64_t n, int dtype) {
  session_end(c);
  if (!c) return fail(RPE_ERR_ARG, "null context");
  if (n < 0 || (dtype != RPE_F32 && dtype != RPE_F64)) return fail(RPE_ERR_ARG, "bad n (%lld) or dtype (%d)", (long long)n, dtype);
  HIP_TRY(hipSetDevice(c->device));
  // a new problem (also one of the same size: new frame) invalidates every array, mask and weight; storage is kept
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int i = 0; i < RPE_NUM_ARRAYS; i++) { c->arr[i] = nullptr; arrays_changed(c, i, false); }
  for (int i = 0; i < 3; i++) { c->mask[i] = nullptr; c->weight[i] = nullptr; }
  c->n = n; c->dtype = dtype;
  return RPE_OK;
}

int rpe_upload(rpe_context* c, int slot, const void* host) {
  session_end(c);
  if (!c || slot < 0 || slot >= RPE_NUM_ARRAYS || !host) return fail(RPE_ERR_ARG, "rpe_upload: bad argument");
  if (c->n <= 0) return fail(RPE_ERR_STATE, "rpe_set_problem first");
  HIP_TRY(hipSetDevice(c->device));
  const size_t bytes = (size_t)c->n * 3 * elem_size(c->dtype);
  if (!c->store[slot] || c->cap[slot] < bytes) {
    if (c->store[slot]) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->store[slot])); c->store[slot] = nullptr;
        c->cap[slot] = 0; }
    HIP_TRY(hipMalloc(&c->store[slot], bytes));
    c->cap[slot] = bytes;
  }
  c->arr[slot] = c->store[slot];
  arrays_changed(c, slot, false);
  HIP_TRY(hipMemcpyAsync(c->arr[slot], host, bytes, hipMemcpyHostToDevice, c->stream));
  return RPE_OK;
}

int rpe_download(rpe_context* c, int slot, void* host) {
  session_end(c);
  if (!c || slot < 0 || slot >= RPE_NUM_ARRAYS || !host) return fail(RPE_ERR_ARG, "rpe_download: bad argument");
  if (!c->arr[slot]) return fail(RPE_ERR_STATE, "array slot %d was never uploaded, bound or produced", slot);
  HIP_TRY(hipSetDevice(c->device));
  return copy_to_host(c, host, c->arr[slot], (size_t)c->n * 3 * elem_size(c->dtype));
}

int rpe_bind(rpe_context* c, int slot, const void* device_ptr) {
  session_end(c);
  if (!c || slot < 0 || slot >= RPE_NUM_ARRAYS) return fail(RPE_ERR_ARG, "rpe_bind: bad argument");
  if (device_ptr && ((uintptr_t)device_ptr & 15u)) return fail(RPE_ERR_ALIGN, "device pointer %p is not 16-byte aligned", device_ptr);
  c->arr[slot] = const_cast<void*>(device_ptr);  // not owned; the context's own storage for this slot stays allocated but idle
  arrays_changed(c, slot, true);
  return RPE_OK;
}

int rpe_upload_mask(rpe_context* c, int mod, const short* host_mask) {
  session_end(c);
  if (!c || mod < 0 || mod > 2) return fail(RPE_ERR_ARG, "rpe_upload_mask: bad argument");
  HIP_TRY(hipSetDevice(c->device));
  if (!host_mask) { c->mask[mod] = nullptr; return RPE_OK; }
  int rc = ensure_mask(c, mod, false);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->mask[mod], host_mask, (size_t)c->n * sizeof(short), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RPE_OK;
}

int rpe_upload_weight(rpe_context* c, int mod, const void* host_weight) {
  session_end(c);
  if (!c || mod < 0 || mod > 2) return fail(RPE_ERR_ARG, "rpe_upload_weight: bad argument");
  HIP_TRY(hipSetDevice(c->device));
  if (!host_weight) { c->weight[mod] = nullptr; return RPE_OK; }
  const size_t need = (size_t)c->n * elem_size(c->dtype);
  if (!c->weight_store[mod] || c->weight_cap[mod] < need) {
    if (c->weight_store[mod]) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->weight_store[mod]));
        c->weight_store[mod] = nullptr; }
    HIP_TRY(hipMalloc(&c->weight_store[mod], need ? need : 8));
    c->weight_cap[mod] = need;
  }
  c->weight[mod] = c->weight_store[mod];
  HIP_TRY(hipMemcpyAsync(c->weight[mod], host_weight, need, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RPE_OK;
}

int rpe_download_mask(rpe_context* c, int mod, short* host_mask) {
  session_end(c);
  if (!c || mod < 0 || mod > 2 || !host_mask) return fail(RPE_ERR_ARG, "rpe_download_mask: bad argument");
  if (!c->mask[mod]) return fail(RPE_ERR_STATE, "no mask for modality %d", mod);
  HIP_TRY(hipSetDevice(c->device));
  return copy_to_host(c, host_mask, c->mask[mod], (size_t)c->n * sizeof(short));
}

// ---------------------------------------------------------------------------------------------- K1'
// the event pair of the next timed launch (rpe_timing_enable), or nulls
static void timing_pair(rpe_context* c, hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = nullptr; *e1 = nullptr;
  if (c->timing && c->ev_used < c->ev0.size() && (c->timing_calls++ % c->timing_stride) == 0) { *e0 = c->ev0[c->ev_used];
      *e1 = c->ev1[c->ev_used]; c->ev_used++; }
}

int rpe_p2p_moments(rpe_context* c, int flags, double* out18) {
  session_end(c);
  int rc = need_arrays(c, {RPE_XW, RPE_XC});
  if (rc) return rc;
  if (!out18) return fail(RPE_ERR_ARG, "null out18");
  if ((rc = check_flags(c, RPE_RES_P2P, flags))) return rc;
  HIP_TRY(hipSetDevice(c->device));
  hipEvent_t e0, e1;
  timing_pair(c, &e0, &e1);
  HIP_TRY(rpe::launch_moments(c->arrays(), flags, collect_target(c), c->stream, e0, e1));
  if ((rc = wait_host(c, rpe::kNeLd))) return rc;
  for (int i = 0; i < 18; i++) out18[i] = c->h_out[i];
  return RPE_OK;
}

int rpe_pose_from_moments(const double* m, double* R9, double* t3) {
  if (!m || !R9 || !t3) return fail(RPE_ERR_ARG, "null argument");
  const double n = m[0];
  if (!(n > 0)) return fail(RPE_ERR_DEGENERATE, "moment record has total weight %g", n);
  rpe::Vec3d Cw(m[1] / n, m[2] / n, m[3] / n), Cc(m[4] / n, m[5] / n, m[6] / n);
  rpe::Mat3d M;  // sum w (Xc - Cc)(Xw - Cw)^T / n  ==  S/n - Cc Cw^T
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M(i, j) = m[7 + 3 * i + j] / n - Cc[i] * Cw[j];
  rpe::Mat3d R = rpe::rotation_from_covariance(M);
  rpe::Vec3d t = Cc - rpe::mul(R, Cw);
  for (int i = 0; i < 9; i++) { if (!std::isfinite(R.a[i])) return fail(RPE_ERR_DEGENERATE, "non-finite rotation"); R9[i] = R.a[i]; }
  for (int i = 0; i < 3; i++) t3[i] = t[i];
  return RPE_OK;
}

// ---------------------------------------------------------------------------------------------- K1/K2/K3
static int joint_launch_checked(rpe_context* c, int nterms, const rpe_term* terms, int flags, const double* pose12, bool clean, int* bits_out);
static int normal_eq_launch(rpe_context* c, int kind, int flags, const double* pose12, double* d_out32, bool clean) {
  if (kind == RPE_RES_NORMAL && !d_out32) {
    const rpe_term t = {RPE_RES_NORMAL, 1.0, RPE_ROBUST_NONE, 1.0};
    return joint_launch_checked(c, 1, &t, flags, pose12, false, nullptr);   // (guarded: this caller does not look at the record's finiteness)
  }
  if (kind == RPE_RES_NORMAL) return fail(RPE_ERR_ARG, "RPE_RES_NORMAL is served by rpe_normal_eq / rpe_normal_eq_joint (host record)");
  int rc = kind_arrays(c, kind);
  if (rc) return rc;
  if (!pose12) return fail(RPE_ERR_ARG, "null argument");
  if ((rc = check_flags(c, kind, flags))) return rc;
  HIP_TRY(hipSetDevice(c->device));
  hipEvent_t e0, e1;
  timing_pair(c, &e0, &e1);
  rpe::ReduceTarget rt = d_out32 ? device_target(c, d_out32) : collect_target(c);
  rt.clean = clean;
  HIP_TRY(rpe::launch_normal_eq(c->arrays(), kind, flags, pose12, rt, c->stream, e0, e1));
  return RPE_OK;
}

int rpe_normal_eq_device(rpe_context* c, int kind, int flags, const double* pose12, double* d_out32) {
  session_end(c);
  if (!d_out32) return fail(RPE_ERR_ARG, "null d_out32");
  return normal_eq_launch(c, kind, flags, pose12, d_out32, c && take_clean(c, kind, false));   // nobody on the host sees this record
}

int rpe_timing_enable(rpe_context* c, int max_records, int stride) {
  if (!c || max_records < 0 || stride < 1) return fail(RPE_ERR_ARG, "rpe_timing_enable: bad argument");
  c->timing_stride = stride; c->timing_calls = 0;
  HIP_TRY(hipSetDevice(c->device));
  while ((int)c->ev0.size() < max_records) {
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    c->ev0.push_back(a); c->ev1.push_back(b);
  }
  c->ev_used = 0;
  c->timing = max_records > 0;
  return RPE_OK;
}

int rpe_timing_collect(rpe_context* c, int* count, double* total_ms, double* min_ms) {
  if (!c) return fail(RPE_ERR_ARG, "null context");
  HIP_TRY(hipStreamSynchronize(c->stream));
  double tot = 0, mn = 1e30;
  for (size_t i = 0; i < c->ev_used; i++) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]));
    tot += ms; if (ms < mn) mn = ms;
  }
  if (count) *count = (int)c->ev_used;
  if (total_ms) *total_ms = tot;
  if (min_ms) *min_ms = c->ev_used ? mn : 0.0;
  c->ev_used = 0;
  return RPE_OK;
}

int rpe_timing_calibrate(rpe_context* c, int pairs, double* avg_ms, double* min_ms) {
  session_end(c);
  if (!c || pairs < 1 || pairs > 4096) return fail(RPE_ERR_ARG, "rpe_timing_calibrate: bad argument");
  HIP_TRY(hipSetDevice(c->device));
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a));
  HIP_TRY(hipEventCreate(&b));
  double tot = 0, mn = 1e30;
  for (int i = 0; i < pairs; i++) {  // one pair at a time, stream idle in between: the way the timed launches see their pair
    HIP_TRY(hipEventRecord(a, c->stream));
    HIP_TRY(hipEventRecord(b, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    tot += ms; if (ms < mn) mn = ms;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  if (avg_ms) *avg_ms = tot / pairs;
  if (min_ms) *min_ms = mn;
  return RPE_OK;
}

// One Gauss-Newton step on one GPU: normal equations (device) -> solve -> pose <- exp(delta) * pose (host).
int rpe_gn_step(rpe_context* c, int kind, int flags, double* pose12, double* ne32_out, double* step_norm) {
  session_end(c);
  double ne[32], d[6];
  int rc = rpe_normal_eq(c, kind, flags, pose12, ne);
  if (rc) return rc;
  if (!rpe::solve_normal_eq6(ne, d, rpe::pivot_floor(c->dtype == RPE_F64))) return fail(RPE_ERR_DEGENERATE, "normal equations are not positive definite (weight sum %g)",
      ne[28]);
  rpe::se3_left_update(d, pose12);
  if (ne32_out) std::memcpy(ne32_out, ne, sizeof(ne));
  if (step_norm) *step_norm = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
  return RPE_OK;
}

int rpe_normal_eq(rpe_context* c, int kind, int flags, const double* pose12, double* out32) {
  session_end(c);
  if (!out32) return fail(RPE_ERR_ARG, "null out32");
  const bool clean = c && take_clean(c, kind, true);   // CLEAN flavour first: this record is looked at right here
  int rc = normal_eq_launch(c, kind, flags, pose12, nullptr, clean);  // null device target = publish to pinned host memory
  if (rc) return rc;
  if ((rc = wait_host(c, rpe::kNeLd))) return rc;
  if (clean) {
    const bool finite = record_finite(c->h_out, 29);
    note_clean_launch(c, kind, finite);
    if (!finite) {   // a NaN or an infinity in the arrays (or in the weights): once more with the guards
      if ((rc = normal_eq_launch(c, kind, flags, pose12, nullptr, false))) return rc;
      if ((rc = wait_host(c, rpe::kNeLd))) return rc;
    }
  }
  for (int i = 0; i < 32; i++) out32[i] = c->h_out[i];
  out32[29] = rpe::pivot_floor(c->dtype == RPE_F64);   // for rpe_gn_solve: the floor that goes with this record's product dtype
  return RPE_OK;
}

struct JointSpec { int bits = 0, robust[5] = {0, 0, 0, 0, 0}; double scale[5] = {0, 0, 0, 0, 0}, rk[5] = {1, 1, 1, 1, 1}; };   // by kind 0..4
static int joint_spec(rpe_context* c, int nterms, const rpe_term* terms, int flags, const double* pose12, JointSpec* out) {
  if (!c || !terms || nterms < 1 || nterms > 4 || !pose12) return fail(RPE_ERR_ARG, "rpe_normal_eq_joint: bad argument");
  int& bits = out->bits;
  int (&robust)[5] = out->robust;
  double (&scale)[5] = out->scale, (&rk)[5] = out->rk;
  for (int t = 0; t < nterms; t++) {
    const int k = terms[t].kind;
    if (k < 0 || k > 4) return fail(RPE_ERR_ARG, "unknown residual kind %d", k);
    if (bits & (1 << k)) return fail(RPE_ERR_ARG, "residual kind %d listed twice", k);
    int rc = kind_arrays(c, k);
    if (rc) return rc;
    if ((rc = check_flags(c, k, flags))) return rc;
    if (terms[t].robust < 0 || terms[t].robust > 2 || (terms[t].robust
        && !(terms[t].robust_k > 0))) return fail(RPE_ERR_ARG, "bad robust setting");
    bits |= 1 << k; scale[k] = terms[t].scale; robust[k] = terms[t].robust; rk[k] = terms[t].robust_k > 0 ? terms[t].robust_k : 1.0;
  }
  if ((bits & 1) && (bits & 2)) return fail(RPE_ERR_ARG, "point-to-point and point-to-plane are alternatives for the 3D-3D term");
  if ((bits & 4) && (bits & 16)) return fail(RPE_ERR_ARG, "bearing and reprojection are alternatives for the 2D-3D term");
  return RPE_OK;
}
static int joint_launch_checked(rpe_context* c, int nterms, const rpe_term* terms, int flags, const double* pose12, bool clean, int* bits_out) {
  JointSpec sp;
  int rc = joint_spec(c, nterms, terms, flags, pose12, &sp);
  if (rc) return rc;
  if (bits_out) *bits_out = sp.bits;
  HIP_TRY(hipSetDevice(c->device));
  rpe::ReduceTarget rt = collect_target(c);
  rt.clean = clean && take_clean_terms(c, sp.bits, true);
  hipEvent_t e0, e1;
  timing_pair(c, &e0, &e1);
  HIP_TRY(rpe::launch_normal_eq_joint(c->arrays(), sp.bits, flags, pose12, sp.scale, sp.robust, sp.rk, rt, c->stream, e0, e1));
  return rt.clean ? 1 : RPE_OK;   // 1 = launched in the CLEAN flavour: the caller looks at the record
}

int rpe_normal_eq_joint(rpe_context* c, int nterms, const rpe_term* terms, int flags, const double* pose12, double* out32) {
  session_end(c);
  if (!out32) return fail(RPE_ERR_ARG, "null out32");
  // CLEAN flavour first (fp32 arrays): this record is looked at right here -- a NaN or an infinity in the arrays shows in it, the
  // launch is repeated guarded and the arrays are remembered as needing the guards (clean-first protocol, as rpe_normal_eq)
  int bits = 0;
  int rc = joint_launch_checked(c, nterms, terms, flags, pose12, true, &bits);
  if (rc != RPE_OK && rc != 1) return rc;
  const bool clean = rc == 1;
  if ((rc = wait_host(c, rpe::kNeLd))) return rc;
  if (clean) {
    const bool finite = record_finite(c->h_out, 29);
    note_clean_terms(c, bits, finite);
    if (!finite) {
      if ((rc = joint_launch_checked(c, nterms, terms, flags, pose12, false, nullptr))) return rc;
      if ((rc = wait_host(c, rpe::kNeLd))) return rc;
    }
  }
  for (int i = 0; i < 32; i++) out32[i] = c->h_out[i];
  out32[29] = rpe::pivot_floor(c->dtype == RPE_F64);
  return RPE_OK;
}

int rpe_gn_refine_joint(rpe_context* c, int nterms, const rpe_term* terms, int flags, double* pose12, int max_iter, double tol,
                        int* iters_out, double* last_step, double* final_cost) {
  session_end(c);
  int it = 0;
  double step = 0, cost = 0;
  if (c && c->resident && c->host_resident && max_iter >= 2 && !c->hostex && !c->comm && c->p2p_world_saved < 1) {
    // ONE launch for the whole refinement, as rpe_gn_refine: the grid of the joint kernel stays resident, the host hands every pose
    // over through the control block, adds the run records, solves and updates.  Frame-sized problems only (one group per thread,
    // staged in LDS: rpe_joint.hip joint_resident_fits); larger ones take the loop below, one launch per iteration.
    JointSpec sp;
    int rc = joint_spec(c, nterms, terms, flags, pose12, &sp);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    double start[12];
    std::memcpy(start, pose12, sizeof(start));
    for (int attempt = 0; attempt < 2; attempt++) {
      const bool clean = take_clean_terms(c, sp.bits, true);   // CLEAN flavour first; its first record is checked
      if (!rpe::joint_resident_fits(c->arrays(), sp.bits, flags, c->max_blocks, false, clean)) { rc = kResidentLost; it = 0; break; }
      int grid = 0, nacc = 0, max_rows = 1, rows_auto = 1;
      rpe::resident_geometry(c->arrays(), RPE_RES_P2PLANE, c->max_blocks, &grid, &nacc, &max_rows, &rows_auto);   // the 29-sum geometry
      double weight = 0;
      bool verified = false;
      auto launch = [&](const rpe::ReduceTarget& rt, unsigned long long base) -> hipError_t {
        return rpe::launch_normal_eq_joint_resident(c->arrays(), sp.bits, flags, sp.scale, sp.robust, sp.rk,
                                                    (const unsigned long long*)c->ctl, base, max_iter, rt, c->stream);
      };
      { std::lock_guard<ResidentSlot> one_resident_grid(resident_mutex(c->device));
        rc = resident_host_loop(c, launch, grid, nacc, max_rows, rows_auto, 1.0, pose12, max_iter, tol, &it, &step, &cost, &weight,
            "normal equations", clean, &verified); }
      if (clean && rc == kResidentDirty) note_clean_terms(c, sp.bits, false);
      else if (clean && verified) note_clean_terms(c, sp.bits, true);
      if (rc != kResidentDirty) break;   // else: NaN-marked arrays -- once more, guarded, from the untouched start pose
      std::memcpy(pose12, start, sizeof(start));
      it = 0;
    }
    if (rc != kResidentLost) {
      if (iters_out) *iters_out = it;
      if (rc != RPE_OK) return rc;
      if (last_step) *last_step = step;
      if (final_cost) *final_cost = cost;
      return RPE_OK;
    }
    // the resident grid was lost after `it` whole iterations: carry on below, one launch per iteration
  }
  for (; it < max_iter; it++) {
    double ne[32], d[6];
    int rc = rpe_normal_eq_joint(c, nterms, terms, flags, pose12, ne);
    if (rc) return rc;
    cost = ne[27];
    if (!rpe::solve_normal_eq6(ne, d, rpe::pivot_floor(c->dtype == RPE_F64))) {
      if (iters_out) *iters_out = it;
      return fail(RPE_ERR_DEGENERATE, "normal equations are not positive definite at iteration %d (weight sum %g)", it, ne[28]);
    }
    rpe::se3_left_update(d, pose12);
    step = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    if (step < tol) { it++; break; }
  }
  if (iters_out) *iters_out = it;
  if (last_step) *last_step = step;
  if (final_cost) *final_cost = cost;
  return RPE_OK;
}

// Device-resident Gauss-Newton: the pose and the loop state live in HBM; every iteration is ONE launch whose last workgroup
// solves the 6x6 system and applies the exp-map update; the host only enqueues the launches and waits for the final record.
int rpe_gn_refine_device(rpe_context* c, int nterms, const rpe_term* terms, int flags, double* pose12, int max_iter, double tol,
                         int* iters_out, double* last_step, double* final_cost) {
  session_end(c);
  if (!c || !terms || nterms < 1 || nterms > 4 || !pose12 || max_iter < 1) return fail(RPE_ERR_ARG,
      "rpe_gn_refine_device: bad argument");
  int bits = 0, robust[5] = {0, 0, 0, 0, 0};
  double scale[5] = {0, 0, 0, 0, 0}, rk[5] = {1, 1, 1, 1, 1};
  for (int t = 0; t < nterms; t++) {
    const int k = terms[t].kind;
    if (k < 0 || k > 4 || (bits & (1 << k))) return fail(RPE_ERR_ARG, "bad residual kind list");
    int rc = kind_arrays(c, k);
    if (rc) return rc;
    if ((rc = check_flags(c, k, flags))) return rc;
    bits |= 1 << k; scale[k] = terms[t].scale; robust[k] = terms[t].robust; rk[k] = terms[t].robust_k > 0 ? terms[t].robust_k : 1.0;
  }
  if ((bits & 1) && (bits & 2)) return fail(RPE_ERR_ARG, "point-to-point and point-to-plane are alternatives for the 3D-3D term");
  if ((bits & 4) && (bits & 16)) return fail(RPE_ERR_ARG, "bearing and reprojection are alternatives for the 2D-3D term");
  HIP_TRY(hipSetDevice(c->device));
  rpe::GnState st;
  st.tol = tol; st.step = 0; st.cost = 0; st.max_iters = max_iter; st.iters = 0; st.done = 0; st.status = 0;
  HIP_TRY(hipMemcpyAsync(c->d_gn_pose, pose12, 12 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_gn_state, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
  rpe::ReduceTarget rt = host_target(c);   // ONE sequence value for the whole batch: only the finishing launch publishes
  rt.gn_pose = c->d_gn_pose; rt.gn = c->d_gn_state;
  // a single plain kind other than N-N uses the dedicated kernel (structured sums for p2p), several kinds the fused one
  const bool single = nterms == 1 && terms[0].kind != RPE_RES_NORMAL && terms[0].robust == 0 && terms[0].scale == 1.0;
  // sharded (rpe_p2p_init): every launch's last workgroup first exchanges the record with the peers, then solves -- identical
  // records on every rank give identical poses and identical stop decisions, so the loop stays one launch per iteration at any
  // number of GPUs.  Launches after convergence skip the exchange on every rank alike; the step counter advances per launch.
  const bool sharded = c->p2p_world >= 1;
  // One GPU, one of the two 3D-3D kinds: ONE launch for the whole loop.  The grid stays resident and iterates by itself -- granule
  // hand-off to the collecting workgroups, run records read back by every workgroup, solve + exp-map in every workgroup alike
  // (rpe_residuals.hpp resident_auto_stage); the host hears from it once, when the loop has finished.  RPE_DEVICE_LOOP_RESIDENT=0: one
  // launch per iteration, as the other residual kinds and the sharded loop keep.
  static const bool auto_on = !(getenv("RPE_DEVICE_LOOP_RESIDENT") && atoi(getenv("RPE_DEVICE_LOOP_RESIDENT")) == 0);
  double pose_in[12];
  std::memcpy(pose_in, pose12, sizeof(pose_in));
  const bool joint_clean = !single && take_clean_terms(c, bits, false);   // no host in these loops: CLEAN only over verified arrays
  // A SOLVING WORKGROUP beside the grid (rpe_residuals.hpp solver_loop): a one-workgroup kernel on a second stream that sums the
  // workers' granules, solves, and hands the poses out -- one hop in and one out instead of two hops in front of `grid` identical
  // solves.  It needs a compute unit of its own, so the workers' grid is capped one below the co-residency cap.  The two kernels must
  // run together; a platform that serialises them ends in the bounded waits (a lost grid, below) and the context never tries again.
  // Single kinds on fewer than 8 workgroups keep the form in which every workgroup solves (one workgroup: no hop at all); the joint
  // kernels have only the solving-workgroup form.
  bool use_solver = false;
  int auto_blocks = c->max_blocks;
  if (auto_on && c->resident && !sharded && !c->comm && !c->hostex && max_iter >= 2 && c->auto_solver) {
    const int capped = std::min(c->max_blocks, rpe::auto_solver_cap());
    int g = 0, na = 0, mr = 1, ra = 1;
    if (capped >= 1) rpe::resident_geometry(c->arrays(), single ? terms[0].kind : RPE_RES_P2PLANE, capped, &g, &na, &mr, &ra);
    if (g >= 1 && (rpe::auto_solver_workers(g) > 0 || !single)) {
      if (!c->stream2) { hipStream_t s2 = nullptr; if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess) c->stream2 = s2; else (void)hipGetLastError(); }
      if (c->stream2) { use_solver = true; auto_blocks = capped; }
    }
  }
  if (auto_on && c->resident && !sharded && !c->comm && !c->hostex && max_iter >= 2 &&
      (single ? rpe::normal_eq_resident_fits(c->arrays(), terms[0].kind, auto_blocks, !use_solver)
              : (use_solver && rpe::joint_resident_fits(c->arrays(), bits, flags, auto_blocks, true, joint_clean)))) {
    // a single plain kind: the dedicated kernel (17 structured sums for point-to-point); anything else: the joint kernel (29 sums)
    int grid = 0, nacc = 0, max_rows = 1, rows_auto = 1;
    rpe::resident_geometry(c->arrays(), single ? terms[0].kind : RPE_RES_P2PLANE, auto_blocks, &grid, &nacc, &max_rows, &rows_auto);
    rt.max_blocks = auto_blocks;
    const unsigned long long base = c->seq;          // granule / run-record tags base + 1 ... base + max_iter
    // as the host-driven loop: the run records are the ones its host would add
    (void)resident_run_shape(grid, nacc, max_rows, rows_auto, &rt);
    c->seq = base + (unsigned long long)max_iter + 1;
    rt.seq = c->seq;                                  // published with the result
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing && c->ev_used < c->ev0.size() && (c->timing_calls++ % c->timing_stride) == 0) { e0 = c->ev0[c->ev_used];
        e1 = c->ev1[c->ev_used]; c->ev_used++; }
    std::lock_guard<ResidentSlot> one_resident_grid(resident_mutex(c->device));   // until the result has arrived
    rt.clean = single ? take_clean(c, terms[0].kind, false) : joint_clean;   // no host in this loop: CLEAN only over verified arrays
    if (use_solver) {
      rt.solver = 1;
      // the solving workgroup reads the start pose and the loop state too: its stream waits for their upload on `stream`
      if (!c->ev_stream2) HIP_TRY(hipEventCreateWithFlags(&c->ev_stream2, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(c->ev_stream2, c->stream));
      HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_stream2, 0));
      HIP_TRY(rpe::launch_auto_solver(nacc, grid, base, max_iter, rt, c->stream2));
    }
    if (single) HIP_TRY(rpe::launch_normal_eq_resident(c->arrays(), terms[0].kind, flags, nullptr, base, max_iter, rt, c->stream, e0,
        e1));
    else HIP_TRY(rpe::launch_normal_eq_joint_resident(c->arrays(), bits, flags, scale, robust, rk, nullptr, base, max_iter, rt,
        c->stream));
    int rc = wait_host(c, rpe::kNeLd);
    if (rc) return rc;
    if (c->h_out[15] != 2.0) {
      for (int i = 0; i < 12; i++) pose12[i] = c->h_out[i];
      if (last_step) *last_step = c->h_out[12];
      if (final_cost) *final_cost = c->h_out[13];
      if (iters_out) *iters_out = (int)c->h_out[14];
      if (c->h_out[15] != 0.0) return fail(RPE_ERR_DEGENERATE, "normal equations are not positive definite at iteration %d",
          (int)c->h_out[14] - 1);
      return RPE_OK;
    }
    // a workgroup's sums never arrived (the grid was not all resident at once): once more from the start pose, one launch per iteration
    if (rt.solver) {   // (the solving workgroup and its workers did not meet)
      c->auto_solver = false;
      (void)hipStreamSynchronize(c->stream2);
      (void)fail(RPE_ERR_HIP, "autonomous loop: the solving workgroup missed the sums of %d of %d workers (workgroups %d .. %d) at iteration %d; finished with one launch per iteration",
                 (int)c->h_out[17], grid, (int)c->h_out[18], (int)c->h_out[19], (int)c->h_out[14]);
#ifdef RPE_SOLVER_DEBUG
      (void)fail(RPE_ERR_HIP, "DBG missing %d of %d (wg %d..%d) it %d | workers started %d, first %+.1f us, last %+.1f us after the solver; scan at %+.1f us", (int)c->h_out[17], grid,
                 (int)c->h_out[18], (int)c->h_out[19], (int)c->h_out[14], (int)c->h_out[20], c->h_out[21], c->h_out[22], c->h_out[23]);
#endif
    }
    else note_lost_grid(c);
    rt.solver = 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_gn_pose, pose_in, 12 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_gn_state, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
    rt = host_target(c);
    rt.gn_pose = c->d_gn_pose; rt.gn = c->d_gn_state;
  }
  for (int it = 0; it < max_iter; it++) {
    if (sharded) { rt.p2p = c->d_p2p; rt.p2p_step = c->p2p_step++; }
    rt.clean = single ? take_clean(c, terms[0].kind, false) : joint_clean;
    if (single) HIP_TRY(rpe::launch_normal_eq(c->arrays(), terms[0].kind, flags, pose_in, rt, c->stream));
    else HIP_TRY(rpe::launch_normal_eq_joint(c->arrays(), bits, flags, pose_in, scale, robust, rk, rt, c->stream));
  }
  int rc = wait_host(c, rpe::kNeLd);
  if (rc) return rc;
  for (int i = 0; i < 12; i++) pose12[i] = c->h_out[i];
  if (last_step) *last_step = c->h_out[12];
  if (final_cost) *final_cost = c->h_out[13];
  if (iters_out) *iters_out = (int)c->h_out[14];
  if (c->h_out[15] == 2.0) return fail(RPE_ERR_HIP,
      "peer-to-peer exchange timed out at iteration %d (a peer did not deliver its record)", (int)c->h_out[14] - 1);
  if (c->h_out[15] != 0.0) return fail(RPE_ERR_DEGENERATE, "normal equations are not positive definite at iteration %d",
      (int)c->h_out[14] - 1);
  return RPE_OK;
}

// Test hook: what ONE iteration of the device-resident loop does with a record -- solve H delta = -g by the kernel's register LDL^T and
// apply pose <- exp(delta) pose by the kernel's own exponential map (sophus/se3.hpp:321-342) -- on the GPU, for a record and pose of
// the caller's.  Returns RPE_ERR_DEGENERATE where the device solve refuses the system.
int rpe_debug_device_gn_update(rpe_context* c, const double* ne32, double* pose12, double* step_norm) {
  session_end(c);
  if (!c || !ne32 || !pose12) return fail(RPE_ERR_ARG, "rpe_debug_device_gn_update: bad argument");
  HIP_TRY(hipSetDevice(c->device));
  double buf[48];
  for (int i = 0; i < 32; i++) buf[i] = ne32[i];
  for (int i = 0; i < 12; i++) buf[32 + i] = pose12[i];
  buf[44] = buf[45] = 0;
  HIP_TRY(hipMemcpyAsync(c->d_out, buf, sizeof(buf), hipMemcpyHostToDevice, c->stream));   // d_out holds 64 doubles
  HIP_TRY(rpe::launch_gn_update_probe(c->d_out, c->d_out + 32, c->d_out + 44, ne32[29] > 1e-12 && ne32[29] < 1e-3 ? ne32[29] : 1e-12, c->stream));
  HIP_TRY(hipMemcpyAsync(buf, c->d_out, sizeof(buf), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (buf[45] == 0.0) return fail(RPE_ERR_DEGENERATE, "device solve: normal equations are not positive definite");
  for (int i = 0; i < 12; i++) pose12[i] = buf[32 + i];
  if (step_norm) *step_norm = buf[44];
  return RPE_OK;
}

// State of the resident loops of a context: enabled (at least one workgroup of the resident kernels per compute unit and fewer than two
// lost grids; bit 1 of *enabled: the host-driven form is available too -- large BAR, RPE_RESIDENT != 0), how many refinements were
// finished with one launch per iteration after their grid was lost, and the co-residency cap of the device.
int rpe_debug_resident_state(rpe_context* c, int* enabled, int* lost, int* cap) {
  if (!c) return fail(RPE_ERR_ARG, "null context");
  if (enabled) *enabled = (c->resident ? (c->host_resident ? 3 : 1) : 0) | (c->resident && c->auto_solver ? 4 : 0);
  if (lost) *lost = c->resident_lost;
  if (cap) *cap = c->resident_cap;
  return RPE_OK;
}

// Fault injection for the tests, per context (the production path reads no environment variable for this): iteration > 0 = the last
// workgroup of the next HOST-driven resident loops withholds its sums of that iteration (its collecting workgroup gives up after its
// bounded wait, the host finishes with one launch per iteration); pose_wait_s > 0 = length of the workgroups' bounded wait for the next
// pose (0.5 .. 60 s).  (0, 0) switches both off.
int rpe_debug_inject_resident_fault(rpe_context* c, int iteration, double pose_wait_s) {
  if (!c || iteration < 0 || pose_wait_s < 0 || (pose_wait_s > 0 && (pose_wait_s < 0.5 || pose_wait_s > 60.0)))
    return fail(RPE_ERR_ARG, "rpe_debug_inject_resident_fault: bad argument");
  c->test_fault_iter = iteration; c->test_pose_wait_s = pose_wait_s;
  return RPE_OK;
}

int rpe_gn_solve(const double* ne32, double* delta6) {
  if (!ne32 || !delta6) return fail(RPE_ERR_ARG, "null argument");
  // ne32[29]: the relative pivot floor of the arithmetic that produced the record (rpe_normal_eq* fill it in; 0 = 1e-12)
  if (!rpe::solve_normal_eq6(ne32, delta6, ne32[29] > 1e-12 && ne32[29] < 1e-3 ? ne32[29] : 1e-12)) return fail(RPE_ERR_DEGENERATE,
      "normal equations are not positive definite");
  return RPE_OK;
}

int rpe_gn_apply(const double* delta6, double* pose12) {
  if (!delta6 || !pose12) return fail(RPE_ERR_ARG, "null argument");
  rpe::se3_left_update(delta6, pose12);
  return RPE_OK;
}

// Host-clock profile of the resident loop: enable = 1 clears and starts, enable = 0 stops and reports the per-loop sums (microseconds)
// of (a) waiting for a record = hand-over in flight + one kernel iteration + record in flight, (b) the host's turn = solve + update +
// hand-over stores, over `steps` steady-state iterations (the first one of every call, which contains the launch, is left out).
int rpe_debug_loop_profile(rpe_context* c, int enable, double* wait_us, double* host_us, long long* steps) {
  if (!c) return fail(RPE_ERR_ARG, "null context");
  if (enable) { c->loop_prof = true; c->prof_wait_us = c->prof_host_us = 0; c->prof_steps = 0; return RPE_OK; }
  c->loop_prof = false;
  if (wait_us) *wait_us = c->prof_wait_us;
  if (host_us) *host_us = c->prof_host_us;
  if (steps) *steps = c->prof_steps;
  return RPE_OK;
}

int rpe_gn_refine(rpe_context* c, int nterms, const int* kinds, const double* scales, int flags, double* pose12, int max_iter, double tol,
                  int* iters_out, double* last_step, double* final_cost) {
  session_end(c);
  if (!c || nterms < 1 || nterms > 4 || !kinds || !pose12) return fail(RPE_ERR_ARG, "rpe_gn_refine: bad argument");
  if (nterms > 1 || kinds[0] == RPE_RES_NORMAL) {  // several residual kinds: ONE fused pass per iteration
    rpe_term terms[4];
    for (int t = 0; t < nterms; t++) { terms[t].kind = kinds[t]; terms[t].scale = scales ? scales[t] : 1.0; terms[t].robust = 0;
        terms[t].robust_k = 1.0; }
    return rpe_gn_refine_joint(c, nterms, terms, flags, pose12, max_iter, tol, iters_out, last_step, final_cost);
  }
  int it = 0;
  double step = 0, cost = 0;
  const double sc = scales ? scales[0] : 1.0;
  // sharded contexts: only with the host-side exchange (every rank's host thread adds the peers' records to its own each iteration);
  // RCCL / in-kernel peer-to-peer contexts take rpe_gn_steps_dist
  const bool sharded_ok = c->hostex ? !c->hostex_shared_gpu : (!c->comm && c->p2p_world_saved < 1);
  if (c->resident && c->host_resident && max_iter >= 2 && sharded_ok && rpe::normal_eq_resident_fits(c->arrays(), kinds[0], c->max_blocks)) {
    // ONE launch for the whole loop: the grid stays resident, the host hands every new pose to it through the control block in
    // device memory (two stores' worth of PCIe latency instead of a kernel launch per iteration) and solves / updates as before.
    int rc = kind_arrays(c, kinds[0]);
    if (rc) return rc;
    if ((rc = check_flags(c, kinds[0], flags))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (c->host_cpu_request != -2 && !c->host_cpu_done) {   // RPE_HOST_CPU: pin / tune the thread that spins here, once per context
      c->host_cpu_done = true;
      if (c->host_cpu_request >= 0) (void)pin_calling_thread(c->host_cpu_request);
      // (auto-tuning runs trial refinements; on a sharded context each of them would take part in the ranks' exchange, and the number
      // of trials is a per-rank matter -- cpusets, local_cpulist -- so the ranks would fall out of step: single-GPU contexts only)
      else if (!c->hostex && !c->comm && c->p2p_world < 1 && c->p2p_world_saved < 1)
        (void)rpe_tune_host_thread(c, kinds[0], flags, pose12, 200, 5, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
    }
    int grid = 0, nacc = 0, max_rows = 1, rows_auto = 1;
    rpe::resident_geometry(c->arrays(), kinds[0], c->max_blocks, &grid, &nacc, &max_rows, &rows_auto);
    const int kind = kinds[0];
    double weight = 0;
    auto launch = [&](const rpe::ReduceTarget& rt, unsigned long long base) -> hipError_t {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (c->timing && c->ev_used < c->ev0.size() && (c->timing_calls++ % c->timing_stride) == 0) { e0 = c->ev0[c->ev_used];
          e1 = c->ev1[c->ev_used]; c->ev_used++; }
      return rpe::launch_normal_eq_resident(c->arrays(), kind, flags, (const unsigned long long*)c->ctl, base, max_iter, rt, c->stream,
          e0, e1);
    };
    for (int attempt = 0; attempt < 2; attempt++) {
      const bool clean = take_clean(c, kind, true);   // CLEAN flavour first; its first record is checked
      bool verified = false;
      { std::lock_guard<ResidentSlot> one_resident_grid(resident_mutex(c->device));
        rc = resident_host_loop(c, launch, grid, nacc, max_rows, rows_auto, sc, pose12, max_iter, tol, &it, &step, &cost, &weight,
            "normal equations", clean, &verified); }
      // promoted to "verified finite" only by a first record that was received and finite: a launch error, a wait that timed out or a
      // grid lost before the first record say nothing about the arrays (their state stays as it was)
      if (clean && rc == kResidentDirty) note_clean_launch(c, kind, false);
      else if (clean && verified) note_clean_launch(c, kind, true);
      if (rc != kResidentDirty) break;   // else: NaN-marked arrays -- once more, guarded, from the untouched start pose
      it = 0;
    }
    if (rc != kResidentLost) {
      if (iters_out) *iters_out = it;
      if (rc != RPE_OK) return rc;
      if (last_step) *last_step = step;
      if (final_cost) *final_cost = cost;
      return RPE_OK;
    }
    // the resident grid was lost after `it` whole iterations: carry on from pose12 below, one launch per iteration
  }
  for (; it < max_iter; it++) {
    double ne[32], d[6];
    int rc = rpe_normal_eq(c, kinds[0], flags, pose12, ne);
    if (rc) return rc;
    if (c->hostex && (rc = rpe_host_exchange_allreduce_f64(c->hostex, ne, 32))) return rc;
    cost = sc * ne[27];
    if (!rpe::solve_normal_eq6(ne, d, rpe::pivot_floor(c->dtype == RPE_F64))) {
      if (iters_out) *iters_out = it;
      return fail(RPE_ERR_DEGENERATE, "normal equations are not positive definite at iteration %d (weight sum %g)", it, ne[28]);
    }
    rpe::se3_left_update(d, pose12);
    step = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    if (step < tol) { it++; break; }
  }
  if (iters_out) *iters_out = it;
  if (last_step) *last_step = step;
  if (final_cost) *final_cost = cost;
  return RPE_OK;
}

// ---------------------------------------------------------------------------------------------- host thread of the resident loops
// The thread that calls rpe_gn_refine spins on the records of every iteration and writes every pose through the PCIe BAR: which CPU
// it sits on is worth 5-10 % of a step (the cores of one socket are alike to about 1 %, the sockets differ by up to 10 % -- in either
// direction, whatever sysfs calls GPU-local; the first cores of a socket take the interrupts and are 3-4 % slower).  This call
// MEASURES it: a handful of candidate CPUs -- the current one, three spread over the GPU-local CPUs, two over the others and one SMT
// sibling -- each pinned in turn and timed with `reps` refinements of `steps` iterations over the context's own arrays (tol = 0, from
// pose12, which is left unchanged); the calling thread then stays pinned to the fastest (sched_setaffinity on the calling thread
// only).  Opt-in: nothing pins a thread unless this is called, or RPE_HOST_CPU=auto | <cpu> is in the environment (then the first
// host-driven resident refinement of a context does it with its own arguments).  Costs candidates x (reps + 1) x steps iterations.

int rpe_tune_host_thread(rpe_context* c, int kind, int flags, const double* pose12, int steps, int reps, int* best_cpu, double* best_us,
                         int* trial_cpus, double* trial_us, int cap, int* ntrials) {
  session_end(c);
  if (!c || !pose12 || steps < 2 || reps < 1 || cap < 0 || (cap > 0 && (!trial_cpus || !trial_us)))
    return fail(RPE_ERR_ARG, "rpe_tune_host_thread: bad argument");
  if (!(c->resident && c->host_resident)) return fail(RPE_ERR_STATE, "rpe_tune_host_thread: this context runs no host-driven resident loop");
  cpu_set_t original;
  CPU_ZERO(&original);
  if (sched_getaffinity(0, sizeof original, &original) != 0) return fail(RPE_ERR_STATE, "sched_getaffinity failed");
  // GPU-local CPUs from sysfs (by PCI bus id), the rest of the online CPUs as "far"
  char bus[64] = {0};
  std::vector<int> local, online = parse_cpulist("/sys/devices/system/cpu/online");
  if (hipDeviceGetPCIBusId(bus, sizeof bus, c->device) == hipSuccess) {
    for (char* p = bus; *p; p++) *p = (char)std::tolower((unsigned char)*p);
    local = parse_cpulist((std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist").c_str());
  } else (void)hipGetLastError();
  const int here = sched_getcpu();
  const int half = (int)online.size() / 2;   // SMT siblings are numbered in the upper half on the hosts this was measured on
  auto is_local = [&](int v) { return std::find(local.begin(), local.end(), v) != local.end(); };
  std::vector<int> near_phys, far_phys, far_all;
  for (int v : online) {
    if (v == here) continue;
    if (is_local(v)) { if (v < half || half == 0) near_phys.push_back(v); }
    else { far_all.push_back(v); if (v < half || half == 0) far_phys.push_back(v); }
  }
  std::vector<int> cand;
  auto add = [&](int v) { if (v >= 0 && std::find(cand.begin(), cand.end(), v) == cand.end()) cand.push_back(v); };
  auto spread = [&](const std::vector<int>& v, double f) { return v.empty() ? -1 : v[std::min(v.size() - 1, (size_t)(v.size() * f))]; };
  add(here);
  add(spread(near_phys, 0.5)); add(spread(near_phys, 0.75)); add(spread(near_phys, 0.9));
  add(spread(far_phys, 0.02)); add(spread(far_phys, 0.5));
  add(spread(far_all, 0.5));
  int tried = 0, pick = -1;
  double pick_us = 1e300;
  std::vector<double> ts((size_t)reps);
  int rc = RPE_OK;
  for (int cpu : cand) {
    if (!pin_calling_thread(cpu)) continue;   // outside the process's cpuset: not a candidate
    double p[12];
    int its = 0;
    double st = 0, co = 0;
    std::memcpy(p, pose12, sizeof p);
    if ((rc = rpe_gn_refine(c, 1, &kind, nullptr, flags, p, steps, 0.0, &its, &st, &co))) break;
    for (int r = 0; r < reps; r++) {
      std::memcpy(p, pose12, sizeof p);
      if ((rc = rpe_synchronize(c))) break;
      const double t0 = clock_us();
      if ((rc = rpe_gn_refine(c, 1, &kind, nullptr, flags, p, steps, 0.0, &its, &st, &co))) break;
      if ((rc = rpe_synchronize(c))) break;
      ts[(size_t)r] = (clock_us() - t0) / steps;
    }
    if (rc) break;
    std::sort(ts.begin(), ts.end());
    const double med = ts[(size_t)reps / 2];
    if (tried < cap) { trial_cpus[tried] = cpu; trial_us[tried] = med; }
    tried++;
    if (med < pick_us) { pick_us = med; pick = cpu; }
  }
  if (rc || pick < 0) { (void)sched_setaffinity(0, sizeof original, &original); return rc ? rc : fail(RPE_ERR_STATE, "no candidate CPU could be pinned"); }
  pin_calling_thread(pick);
  if (best_cpu) *best_cpu = pick;
  if (best_us) *best_us = pick_us;
  if (ntrials) *ntrials = tried < cap ? tried : cap;
  return RPE_OK;
}

// ---------------------------------------------------------------------------------------------- RCCL (multi-GPU)

int rpe_comm_unique_id(void* id128) {
  if (!id128) return fail(RPE_ERR_ARG, "null id");
  if (!rccl().ok) return fail(RPE_ERR_STATE, "librccl.so.1 could not be loaded: %s", dlerror());
  ncclUniqueId id;
  NCCL_TRY(rccl().GetUniqueId(&id));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, 128);
  return RPE_OK;
}

int rpe_comm_init(rpe_context* c, int world, int rank, const void* id128) {
  session_end(c);
  if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return fail(RPE_ERR_ARG, "rpe_comm_init: bad argument");
  if (!rccl().ok) return fail(RPE_ERR_STATE, "librccl.so.1 could not be loaded");
  HIP_TRY(hipSetDevice(c->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  NCCL_TRY(rccl().CommInitRank(&c->comm, world, id, rank));
  c->comm_world = world;
  return RPE_OK;
}

// ranks of the context's RCCL communicator as the communicator itself reports them (ncclCommCount); 0 = no communicator
int rpe_comm_count(rpe_context* c, int* ranks) {
  if (!c || !ranks) return fail(RPE_ERR_ARG, "rpe_comm_count: bad argument");
  *ranks = 0;
  if (!c->comm) return RPE_OK;
  if (!rccl().CommCount) return fail(RPE_ERR_STATE, "ncclCommCount is not exported by the loaded librccl");
  NCCL_TRY(rccl().CommCount(c->comm, ranks));
  return RPE_OK;
}

// PCI bus id of the context's GPU ("0000:05:00.0"): one process per GPU means every rank of a node reports a different one
int rpe_device_bus_id(rpe_context* c, char* buf, int len) {
  if (!c || !buf || len < 16) return fail(RPE_ERR_ARG, "rpe_device_bus_id: bad argument (need a buffer of >= 16 bytes)");
  HIP_TRY(hipDeviceGetPCIBusId(buf, len, c->device));
  return RPE_OK;
}

int rpe_comm_destroy(rpe_context* c) {
  session_end(c);
  if (!c) return fail(RPE_ERR_ARG, "null context");
  if (c->comm) { (void)hipStreamSynchronize(c->stream); NCCL_TRY(rccl().CommDestroy(c->comm)); c->comm = nullptr; c->comm_world = 1; }
  return RPE_OK;
}

// ---- peer-to-peer exchange over xGMI (one process per GPU, one node, <= 8 ranks)
int rpe_p2p_export(rpe_context* c, void* handle64) {
  session_end(c);
  if (!c || !handle64) return fail(RPE_ERR_ARG, "rpe_p2p_export: bad argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  HIP_TRY(hipSetDevice(c->device));
  if (!c->p2p_box) {
    void* p = nullptr;
    // fine-grained (uncached across the fabric) device memory, as collective libraries use for their flag buffers
    hipError_t e = hipExtMallocWithFlags(&p, rpe::kP2PMailboxBytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { (void)hipGetLastError(); HIP_TRY(hipMalloc(&p, rpe::kP2PMailboxBytes)); }
    c->p2p_box = (unsigned long long*)p;
  }
  HIP_TRY(hipMemset(c->p2p_box, 0, rpe::kP2PMailboxBytes));
  HIP_TRY(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  HIP_TRY(hipIpcGetMemHandle(&h, c->p2p_box));
  std::memcpy(handle64, &h, 64);
  return RPE_OK;
}

int rpe_p2p_init(rpe_context* c, int world, int rank, const void* handles) {
  session_end(c);
  if (!c || !handles || world < 1 || world > rpe::kP2PMaxWorld || rank < 0 || rank >= world) return fail(RPE_ERR_ARG,
      "rpe_p2p_init: bad argument (1 <= world <= 8)");
  if (!c->p2p_box) return fail(RPE_ERR_STATE, "rpe_p2p_export first");
  HIP_TRY(hipSetDevice(c->device));
  for (int r = 0; r < rpe::kP2PMaxWorld; r++)   // a second init: drop the mappings of the first
    if (c->p2p_peer[r]) { (void)hipIpcCloseMemHandle(c->p2p_peer[r]); c->p2p_peer[r] = nullptr; }
  c->p2p_world = 0; c->p2p_world_saved = 0;
  // A new session restarts the step counters at 0, so the mailbox must not hold the tags of an earlier one (tag 1 left in the
  // parity-0 slots would make the new step 0 accept stale records).  Peers write here only inside an exchange, and ranks enter their
  // first exchange together (a barrier after init, see the header), i.e. after every rank has passed this point.
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemset(c->p2p_box, 0, rpe::kP2PMailboxBytes));
  HIP_TRY(hipDeviceSynchronize());
  rpe::P2PDesc d;
  d.world = world; d.rank = rank;
  for (int r = 0; r < rpe::kP2PMaxWorld; r++) d.peer[r] = nullptr;
  for (int r = 0; r < world; r++) {
    if (r == rank) { d.peer[r] = c->p2p_box; continue; }
    hipIpcMemHandle_t h;
    std::memcpy(&h, (const char*)handles + 64 * (size_t)r, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      for (int k = 0; k < r; k++) if (c->p2p_peer[k]) { (void)hipIpcCloseMemHandle(c->p2p_peer[k]); c->p2p_peer[k] = nullptr; }
      return fail(RPE_ERR_HIP, "hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
    }
    c->p2p_peer[r] = p;
    d.peer[r] = (unsigned long long*)p;
  }
  if (!c->d_p2p) HIP_TRY(hipMalloc((void**)&c->d_p2p, sizeof(rpe::P2PDesc)));
  HIP_TRY(hipMemcpy(c->d_p2p, &d, sizeof(d), hipMemcpyHostToDevice));
  c->p2p_world = world; c->p2p_world_saved = world; c->p2p_rank = rank; c->p2p_step = 0; c->p2p_vote_step = 0;
  return RPE_OK;
}

// pause = 1: keep the mailboxes mapped but let rpe_gn_step_dist / rpe_score use the RCCL communicator (or nothing); 0 resumes.  Every
// rank must switch at the same point of its call sequence.
int rpe_p2p_pause(rpe_context* c, int pause) {
  session_end(c);
  if (!c) return fail(RPE_ERR_ARG, "null context");
  if (!c->d_p2p || c->p2p_world_saved < 1) return fail(RPE_ERR_STATE, "rpe_p2p_init was not called");
  c->p2p_world = pause ? 0 : c->p2p_world_saved;
  return RPE_OK;
}

int rpe_p2p_destroy(rpe_context* c) {
  session_end(c);
  if (!c) return fail(RPE_ERR_ARG, "null context");
  if (!c->p2p_box && !c->d_p2p) return RPE_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (int r = 0; r < rpe::kP2PMaxWorld; r++) if (c->p2p_peer[r]) { (void)hipIpcCloseMemHandle(c->p2p_peer[r]);
      c->p2p_peer[r] = nullptr; }
  if (c->d_p2p) { (void)hipFree(c->d_p2p); c->d_p2p = nullptr; }
  if (c->p2p_box) { (void)hipFree(c->p2p_box); c->p2p_box = nullptr; }
  c->p2p_world = 0; c->p2p_world_saved = 0; c->p2p_step = 0;
  return RPE_OK;
}

// Sharded Gauss-Newton step: local normal equations -> in-place all-reduce(sum) of the 32-double record over RCCL on the
// context's stream -> publish to pinned host memory -> (every rank, identically) solve + exp-map update.
// ---- host-side exchange between the rank processes of one node (csrc/rpe_hostex.cpp)
int rpe_hostex_init(rpe_context* c, int world, int rank, const char* name, int create) {
  session_end(c);
  if (!c || !name) return fail(RPE_ERR_ARG, "rpe_hostex_init: bad argument");
  if (c->hostex) return fail(RPE_ERR_STATE, "rpe_hostex_init: an exchange is already set (rpe_hostex_destroy first)");
  rpe_host_exchange* h = nullptr;
  int rc = rpe_host_exchange_open(name, world, rank, create, 10.0, &h);
  if (rc) return rc;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, c->device) != hipSuccess) { (void)hipGetLastError();
      std::snprintf(bus, sizeof bus, "device%d", c->device); }
  (void)rpe_host_exchange_set_label(h, bus);
  double probe[1] = {1.0};   // first exchange: every rank is here, and every rank's GPU label is in place
  rc = rpe_host_exchange_allreduce_f64(h, probe, 1);
  if (rc == RPE_OK && probe[0] != (double)world) rc = fail(RPE_ERR_STATE, "host exchange: %g of %d ranks answered", probe[0], world);
  if (rc) { rpe_host_exchange_close(h); return rc; }
  if (create) (void)rpe_host_exchange_unlink(h);   // everyone has it mapped: the name can go (nothing is left behind in /dev/shm)
  static const bool allow_shared = getenv("RPE_HOSTEX_ALLOW_SHARED") && atoi(getenv("RPE_HOSTEX_ALLOW_SHARED")) != 0;
  c->hostex_shared_gpu = rpe_host_exchange_labels_collide(h) != 0 && !allow_shared;
  c->hostex = h; c->hostex_world = world;
  return RPE_OK;
}
int rpe_hostex_destroy(rpe_context* c) {
  session_end(c);
  if (!c) return fail(RPE_ERR_ARG, "null context");
  if (c->hostex) { rpe_host_exchange_close(c->hostex); c->hostex = nullptr; c->hostex_world = 1; c->hostex_shared_gpu = false; }
  return RPE_OK;
}

int rpe_gn_step_dist(rpe_context* c, int kind, int flags, double* pose12, double* ne32_out, double* step_norm) {
  session_end(c);
  if (c && c->hostex) {   // ONE launch with the single-GPU collecting stage; the shards' records meet on the hosts
    double ne[32], d[6];
    int rc = rpe_normal_eq(c, kind, flags, pose12, ne);
    if (rc) return rc;
    if ((rc = rpe_host_exchange_allreduce_f64(c->hostex, ne, 32))) return rc;
    ne[29] = rpe::pivot_floor(c->dtype == RPE_F64);   // slot 29 is not a sum: after the exchange it held world x floor
    if (!rpe::solve_normal_eq6(ne, d, rpe::pivot_floor(c->dtype == RPE_F64))) return fail(RPE_ERR_DEGENERATE, "normal equations are not positive definite (weight sum %g)",
        ne[28]);
    rpe::se3_left_update(d, pose12);
    if (ne32_out) std::memcpy(ne32_out, ne, sizeof(ne));
    if (step_norm) *step_norm = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    return RPE_OK;
  }
  if (!c || (!c->comm && c->p2p_world < 1)) return fail(RPE_ERR_STATE, "neither rpe_p2p_init nor rpe_comm_init was called");
  int rc;
  if (c->p2p_world >= 1) {
    // ONE launch: the kernel's last workgroup exchanges the record with the peers over xGMI, sums in rank order, publishes
    if (kind == RPE_RES_NORMAL) return fail(RPE_ERR_ARG, "RPE_RES_NORMAL is not served by the sharded step");
    if ((rc = kind_arrays(c, kind))) return rc;
    if (!pose12) return fail(RPE_ERR_ARG, "null argument");
    if ((rc = check_flags(c, kind, flags))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing && c->ev_used < c->ev0.size() && (c->timing_calls++ % c->timing_stride) == 0) { e0 = c->ev0[c->ev_used];
        e1 = c->ev1[c->ev_used]; c->ev_used++; }
    rpe::ReduceTarget rt = host_target(c);
    rt.p2p = c->d_p2p; rt.p2p_step = c->p2p_step++;
    rt.clean = take_clean(c, kind, false);   // the record is summed with the peers' inside the kernel
    HIP_TRY(rpe::launch_normal_eq(c->arrays(), kind, flags, pose12, rt, c->stream, e0, e1));
  } else {
    if ((rc = normal_eq_launch(c, kind, flags, pose12, c->d_out, take_clean(c, kind, false)))) return rc;
    NCCL_TRY(rccl().AllReduce(c->d_out, c->d_out, 32, ncclFloat64, ncclSum, c->comm, c->stream));
    const unsigned long long seq = ++c->seq;
    HIP_TRY(rpe::launch_publish_f64(c->d_out, 32, c->h_out, reinterpret_cast<unsigned long long*>(c->h_out + rpe::kNeLd), seq,
        c->stream));
  }
  if ((rc = wait_host(c, rpe::kNeLd))) return rc;
  double ne[32], d[6];
  for (int i = 0; i < 32; i++) ne[i] = c->h_out[i];
  if (c->p2p_world >= 1 && ne[31] != 0.0) return fail(RPE_ERR_HIP,
      "peer-to-peer exchange timed out at step %llu (a peer did not deliver its record)", c->p2p_step - 1);
  ne[29] = rpe::pivot_floor(c->dtype == RPE_F64);   // the record handed out carries the floor rpe_gn_solve reads, as rpe_normal_eq's does
  if (!rpe::solve_normal_eq6(ne, d, rpe::pivot_floor(c->dtype == RPE_F64))) return fail(RPE_ERR_DEGENERATE, "normal equations are not positive definite (weight sum %g)",
      ne[28]);
  rpe::se3_left_update(d, pose12);
  if (ne32_out) std::memcpy(ne32_out, ne, sizeof(ne));
  if (step_norm) *step_norm = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
  return RPE_OK;
}

// `steps` sharded steps in one call (the host loop stays inside the library, as rpe_gn_refine keeps it for one GPU)
int rpe_gn_steps_dist(rpe_context* c, int kind, int flags, double* pose12, int steps, double* last_step_norm) {
  session_end(c);
  if (steps < 0) return fail(RPE_ERR_ARG, "rpe_gn_steps_dist: negative step count");
  double sn = 0;
  for (int k = 0; k < steps; k++) {
    const int rc = rpe_gn_step_dist(c, kind, flags, pose12, nullptr, &sn);
    if (rc) return rc;
  }
  if (last_step_norm) *last_step_norm = sn;
  return RPE_OK;
}

// ---------------------------------------------------------------------------------------------- K4
static int vote_arrays(rpe_context* c, int kind) {
  switch (kind) {
    case RPE_VOTE_33: return need_arrays(c, {RPE_XW, RPE_XC});
    case RPE_VOTE_23: case RPE_VOTE_23_MATRIX: return need_arrays(c, {RPE_XW, RPE_BV});
    case RPE_VOTE_33_23: return need_arrays(c, {RPE_XW, RPE_XC, RPE_BV});
    case RPE_VOTE_NN_23: return need_arrays(c, {RPE_XW, RPE_XC, RPE_BV, RPE_NW, RPE_NC});
    case RPE_VOTE_NN_33: return need_arrays(c, {RPE_XW, RPE_XC, RPE_NW, RPE_NC});
    case RPE_VOTE_NN_33_23: return need_arrays(c, {RPE_XW, RPE_XC, RPE_BV, RPE_NW, RPE_NC});
  }
  return fail(RPE_ERR_ARG, "unknown vote kind %d", kind);
}

// host -> staging in the kernel's layout.  fast: R(9) t(3) ; exact: q(4) t(3) pad
static void stage_poses(int dtype, int exact, const double* poses7, int H, void* dst) {
  for (int h = 0; h < H; h++) {
    const double* p = poses7 + 7 * h;
    double v[12];
    int cnt;
    if (exact) { for (int k = 0; k < 7; k++) v[k] = p[k]; v[7] = 0; cnt = 8; }
    else {
      rpe::Quat<double> q{p[0], p[1], p[2], p[3]};
      rpe::quat_to_R(q, v);
      v[9] = p[4]; v[10] = p[5]; v[11] = p[6]; cnt = 12;
    }
    if (dtype == RPE_F64) std::memcpy((double*)dst + (size_t)h * cnt, v, cnt * sizeof(double));
    else { float* f = (float*)dst + (size_t)h * cnt; for (int k = 0; k < cnt; k++) f[k] = (float)v[k]; }
  }
}
static void stage_thresholds(int dtype, int exact, double thre_3d, double cos_thr, double cos_nl, double thr[3]) {
  if (exact) thr[0] = dtype == RPE_F64 ? sqrt_cut<double>(thre_3d) : (double)sqrt_cut<float>((float)thre_3d);
  else thr[0] = dtype == RPE_F64 ? thre_3d * thre_3d : (double)((float)thre_3d * (float)thre_3d);
  thr[1] = cos_thr; thr[2] = cos_nl;
}

// The scoring kernels ACCUMULATE into c->d_votes and rely on the read-out kernel to leave the counters zero.  If anything between
// launch_score and the read-out fails (a collective, a launch), the counters would stay dirty and every later scoring call would be
// silently wrong: clear them on the way out.
static int votes_or_clear(rpe_context* c, hipError_t e, int count) {
  if (e == hipSuccess) return RPE_OK;
  (void)hipMemsetAsync(c->d_votes, 0, (size_t)count * sizeof(int), c->stream);
  return fail(RPE_ERR_HIP, "vote read-out: %s", hipGetErrorString(e));
}
static int nccl_votes_or_clear(rpe_context* c, ncclResult_t r, int count) {
  if (r == ncclSuccess) return RPE_OK;
  (void)hipMemsetAsync(c->d_votes, 0, (size_t)count * sizeof(int), c->stream);
  return fail(RPE_ERR_HIP, "all-reduce of the vote counters: %s", rccl().GetErrorString ? rccl().GetErrorString(r) : "rccl error");
}

// test hook: the value the exact kernels compare the squared 3D residual with (dtype 0: evaluated in float, 1: in double)
double rpe_host_sqrt_cut(int dtype, double thre_3d) { return dtype == RPE_F64 ? sqrt_cut<double>(thre_3d) : (double)sqrt_cut<float>((float)thre_3d); }

int rpe_score(rpe_context* c, int kind, int mode, const double* poses7, int H, double thre_3d, double cos_thr, double cos_nl,
    int* votes_out) {
  int rc = vote_arrays(c, kind);
  if (rc) return rc;
  if (!poses7 || !votes_out || H < 0) return fail(RPE_ERR_ARG, "rpe_score: bad argument");
  HIP_TRY(hipSetDevice(c->device));
  const int exact = mode == RPE_SCORE_EXACT;
  if (c->sess.active) {   // a resident scoring session is open: short lists with its parameters go through its grid
    if (H >= 1 && H <= 4 * rpe::kSessionHypsMax && session_matches(c, kind, mode, thre_3d, cos_thr, cos_nl)) {
      const size_t per = (size_t)(exact ? 8 : 12) * elem_size(c->dtype);
      stage_poses(c->dtype, exact, poses7, H, c->h_poses);
      int h0 = 0;
      for (; h0 < H; h0 += rpe::kSessionHypsMax) {
        const int hn = std::min(rpe::kSessionHypsMax, H - h0);
        double tot[rpe::kSessionHypsMax];
        if (session_batch(c, 0, (const char*)c->h_poses + (size_t)h0 * per, hn, (size_t)hn * per, tot) != RPE_OK) break;   // (closed: launch)
        for (int i = 0; i < hn; i++) votes_out[h0 + i] = (int)tot[i];
      }
      if (h0 >= H) {
        if (c->sess.seen_votes.size() + (size_t)H <= 1024) {
          c->sess.seen_pose.insert(c->sess.seen_pose.end(), poses7, poses7 + (size_t)7 * H);
          c->sess.seen_votes.insert(c->sess.seen_votes.end(), votes_out, votes_out + H);
        }
        return RPE_OK;
      }
    } else session_end(c);
  } else session_end(c);   // (a closed session's masks may still be unverified)
  double thr[3];
  stage_thresholds(c->dtype, exact, thre_3d, cos_thr, cos_nl, thr);
  const size_t per = (exact ? 8 : 12) * elem_size(c->dtype);
  for (int h0 = 0; h0 < H; h0 += rpe::kMaxScoreH) {
    const int hb = std::min(rpe::kMaxScoreH, H - h0);
    stage_poses(c->dtype, exact, poses7 + (size_t)7 * h0, hb, c->h_poses);
    if ((c->hostex || (!c->comm && c->p2p_world < 1)) && hb <= rpe::score_small_cap(c->dtype, exact)) {
      // short list on one GPU: ONE launch -- the hypotheses ride in the kernel argument, the counts come back as run records
      rpe::ReduceTarget rt = collect_target(c);
      if (rt.rows > 0) {
        HIP_TRY(rpe::launch_score_small(c->arrays(), kind, exact, c->h_poses, nullptr, hb, thr, rt, c->stream));
        if ((rc = wait_host(c, rpe::kNeLd))) return rc;
        for (int i = 0; i < hb; i++) votes_out[h0 + i] = (int)c->h_out[i];
        // sharded: the shards' counts meet on the hosts
        if (c->hostex && (rc = rpe_host_exchange_allreduce_i32(c->hostex, votes_out + h0, hb))) return rc;
        continue;
      }
    }
    HIP_TRY(hipMemcpyAsync(c->d_poses, c->h_poses, per * hb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(rpe::launch_score(c->arrays(), kind, exact, c->d_poses, hb, thr, c->d_votes, c->score_blocks, c->stream));
    // sharded correspondences
    if (!c->hostex && c->comm && c->p2p_world < 1 && (rc = nccl_votes_or_clear(c, rccl().AllReduce(c->d_votes, c->d_votes, (size_t)hb,
        ncclInt32, ncclSum, c->comm, c->stream), hb))) return rc;
    // read-out without a D2H copy or a stream synchronisation: a tiny kernel stores the counters into pinned host memory, raises
    // a sequence word the host spins on, and clears the counters for the next launch
    const unsigned long long seq = ++c->vote_seq;
    // sharded correspondences, one node: the read-out kernel also exchanges and sums the counters
    if (!c->hostex && c->p2p_world >= 1) {
      if ((rc = votes_or_clear(c, rpe::launch_publish_votes_p2p(c->d_votes, hb, c->d_p2p, c->p2p_vote_step++, c->h_votes,
          c->h_votes + rpe::kMaxScoreH + 2, c->h_flag2, seq, c->stream), hb))) return rc;
      if ((rc = wait_flag(c, c->h_flag2, seq))) return rc;
      if (c->h_votes[rpe::kMaxScoreH + 2] != 0) return fail(RPE_ERR_HIP,
          "peer-to-peer exchange of the vote counters timed out (a peer did not deliver)");
    } else {
      if ((rc = votes_or_clear(c, rpe::launch_publish_votes(c->d_votes, hb, c->h_votes, c->h_flag2, seq, c->stream), hb))) return rc;
      if ((rc = wait_flag(c, c->h_flag2, seq))) return rc;
    }
    std::memcpy(votes_out + h0, c->h_votes, (size_t)hb * sizeof(int));
    if (c->hostex && (rc = rpe_host_exchange_allreduce_i32(c->hostex, votes_out + h0, hb))) return rc;
  }
  return RPE_OK;
}

// Device-side generation + scoring of one batch of 3D-3D RANSAC iterations (the vote loop V1/V2 with its hypothesis generator H1)
int rpe_ransac33_batch(rpe_context* c, uint64_t rng_state, uint64_t rng_inc, int iters, int mode, double thre_3d, int* votes_out,
                       double* q7_out, unsigned char* valid_out) {
  session_end(c);
  int rc = need_arrays(c, {RPE_XW, RPE_XC});
  if (rc) return rc;
  if (!votes_out || !q7_out || !valid_out || iters < 1 || iters > rpe::kMaxScoreH) return fail(RPE_ERR_ARG,
      "rpe_ransac33_batch: bad argument (1 <= iters <= %d)", rpe::kMaxScoreH);
  if (c->n < 3) return fail(RPE_ERR_ARG, "rpe_ransac33_batch: fewer than 3 correspondences");
  // The generator samples THIS context's arrays: on a sharded context (rpe_comm_init / rpe_p2p_init) iteration i would be a different
  // pose on every rank and the summed votes would mix unrelated hypotheses.  Sharded RANSAC = host hypotheses (every rank the same
  // list) + rpe_score, which all-reduces the votes of IDENTICAL poses.
  if (c->comm || c->p2p_world >= 1 || c->p2p_world_saved >= 1 || c->hostex)
    return fail(RPE_ERR_STATE,
        "rpe_ransac33_batch samples the local arrays and is not defined on a sharded context; generate hypotheses once and use rpe_score");
  HIP_TRY(hipSetDevice(c->device));
  const int exact = mode == RPE_SCORE_EXACT;
  double thr[3];
  stage_thresholds(c->dtype, exact, thre_3d, 2.0, 2.0, thr);
  HIP_TRY(rpe::launch_gen_shinji(c->arrays(), rng_state, rng_inc, iters, exact, c->d_poses, c->h_poses, c->stream));
  rpe::ReduceTarget rt = iters <= rpe::score_small_cap(c->dtype, exact) ? collect_target(c) : host_target(c);
  if (rt.rows > 0) {   // short batch: the scoring kernel itself hands the counts to the host (no read-out kernel)
    HIP_TRY(rpe::launch_score_small(c->arrays(), RPE_VOTE_33, exact, nullptr, c->d_poses, iters, thr, rt, c->stream));
    if ((rc = wait_host(c, rpe::kNeLd))) return rc;
    for (int i = 0; i < iters; i++) votes_out[i] = (int)c->h_out[i];
  } else {
    HIP_TRY(rpe::launch_score(c->arrays(), RPE_VOTE_33, exact, c->d_poses, iters, thr, c->d_votes, c->score_blocks, c->stream));
    const unsigned long long seq = ++c->vote_seq;
    if ((rc = votes_or_clear(c, rpe::launch_publish_votes(c->d_votes, iters, c->h_votes, c->h_flag2, seq, c->stream),
        iters))) return rc;
    if ((rc = wait_flag(c, c->h_flag2, seq))) return rc;
    std::memcpy(votes_out, c->h_votes, (size_t)iters * sizeof(int));
  }
  // the generator stored the hypotheses into pinned host memory before the scoring kernel ran (same stream): they are complete
  for (int i = 0; i < iters; i++) {
    if (c->dtype == RPE_F64) { const double* h = (const double*)c->h_poses + 8 * (size_t)i;
        for (int k = 0; k < 7; k++) q7_out[7 * (size_t)i + k] = h[k]; valid_out[i] = h[7] != 0.0; }
    else { const float* h = (const float*)c->h_poses + 8 * (size_t)i; for (int k = 0; k < 7; k++) q7_out[7 * (size_t)i + k] = h[k];
        valid_out[i] = h[7] != 0.0f; }
  }
  return RPE_OK;
}

// Device-side generation + scoring of one batch of iterations of a plain-RANSAC solver with a 4-point sample, FAST scoring mode
// (tolerance parity with the host's hypotheses, not bit parity): solver 0 = kneip_ransac, 1 = shinji_kneip_ransac, 2 = nl_kneip_ransac,
// 3 = nl_shinji_ransac, 4 = nl_shinji_kneip_ransac (slots per iteration: 1, 2, 1, 2, 3)
int rpe_ransac_p3p_batch(rpe_context* c, int solver, uint64_t rng_state, uint64_t rng_inc, int iters, double thre_3d, double cos_thr,
                         double cos_nl, int* votes_out, double* q7_out, unsigned char* valid_out) {
  session_end(c);
  const int per = rpe::gen_p3p_slots(solver);
  if (per == 0) return fail(RPE_ERR_ARG, "rpe_ransac_p3p_batch: solver must be 0 .. 4");
  static const int kinds[5] = {RPE_VOTE_23, RPE_VOTE_33_23, RPE_VOTE_NN_23, RPE_VOTE_NN_33, RPE_VOTE_NN_33_23};
  const int kind = kinds[solver];
  int rc = vote_arrays(c, kind);
  if (rc) return rc;
  if (solver == 3 && (rc = need_arrays(c, {RPE_XW, RPE_XC, RPE_NW, RPE_NC}))) return rc;
  if (!votes_out || !q7_out || !valid_out || iters < 1 || (int64_t)iters * per > rpe::kMaxScoreH)
    return fail(RPE_ERR_ARG, "rpe_ransac_p3p_batch: bad argument (1 <= iters x slots <= %d)", rpe::kMaxScoreH);
  if (c->n < 4) return fail(RPE_ERR_ARG, "rpe_ransac_p3p_batch: fewer than 4 correspondences");
  // as rpe_ransac33_batch: the generator samples the local shard
  if (c->comm || c->p2p_world >= 1 || c->p2p_world_saved >= 1 || c->hostex)
    return fail(RPE_ERR_STATE,
        "rpe_ransac_p3p_batch samples the local arrays and is not defined on a sharded context; generate hypotheses once and use rpe_score");
  HIP_TRY(hipSetDevice(c->device));
  const int slots = iters * per;
  double thr[3];
  stage_thresholds(c->dtype, /*exact=*/0, thre_3d, cos_thr, cos_nl, thr);
  HIP_TRY(rpe::launch_gen_p3p(c->arrays(), solver, rng_state, rng_inc, iters, c->d_poses, c->h_poses, c->stream));
  HIP_TRY(rpe::launch_score(c->arrays(), kind, 0, c->d_poses, slots, thr, c->d_votes, c->score_blocks, c->stream));
  const unsigned long long seq = ++c->vote_seq;
  if ((rc = votes_or_clear(c, rpe::launch_publish_votes(c->d_votes, slots, c->h_votes, c->h_flag2, seq, c->stream), slots))) return rc;
  if ((rc = wait_flag(c, c->h_flag2, seq))) return rc;
  std::memcpy(votes_out, c->h_votes, (size_t)slots * sizeof(int));
  for (int i = 0; i < slots; i++) {
    if (c->dtype == RPE_F64) { const double* h = (const double*)c->h_poses + 8 * (size_t)i;
        for (int k = 0; k < 7; k++) q7_out[7 * (size_t)i + k] = h[k]; valid_out[i] = h[7] != 0.0; }
    else { const float* h = (const float*)c->h_poses + 8 * (size_t)i; for (int k = 0; k < 7; k++) q7_out[7 * (size_t)i + k] = h[k];
        valid_out[i] = h[7] != 0.0f; }
  }
  return RPE_OK;
}

static int mask_by_launch(rpe_context* c, int kind, int mode, const double* pose7, double thre_3d, double cos_thr, double cos_nl, int* votes_out) {
  const int exact = mode == RPE_SCORE_EXACT;
  double thr[3];
  stage_thresholds(c->dtype, exact, thre_3d, cos_thr, cos_nl, thr);
  double staged[12];
  stage_poses(RPE_F64, exact, pose7, 1, staged);  // layout only; the launcher rounds to the array dtype
  hipEvent_t e0, e1;
  timing_pair(c, &e0, &e1);
  HIP_TRY(rpe::launch_mask(c->arrays(), kind, exact, staged, thr, collect_target(c), c->stream, e0, e1));
  int rc;
  if ((rc = wait_host(c, rpe::kNeLd))) return rc;
  c->h_votes[0] = (int)c->h_out[0];
  if (votes_out) *votes_out = c->h_votes[0];
  return RPE_OK;
}
int rpe_inlier_mask(rpe_context* c, int kind, int mode, const double* pose7, double thre_3d, double cos_thr, double cos_nl,
    int* votes_out) {
  int rc = vote_arrays(c, kind);
  if (rc) return rc;
  if (!pose7) return fail(RPE_ERR_ARG, "null pose");
  HIP_TRY(hipSetDevice(c->device));
  if (session_matches(c, kind, mode, thre_3d, cos_thr, cos_nl)) {   // by the session's grid (launched with the mask arrays of this kind)
    const int exact = mode == RPE_SCORE_EXACT;
    double one[12];
    stage_poses(c->dtype, exact, pose7, 1, one);
    const size_t bytes = (size_t)(exact ? 8 : 12) * elem_size(c->dtype);
    int known = 0;
    static const bool lazy = !(getenv("RPE_SESSION_LAZY_MASK") && atoi(getenv("RPE_SESSION_LAZY_MASK")) == 0);
    if (lazy && session_seen(c, pose7, &known)) {   // the run's winner: its total is known, the masks end the session
      session_final_masks(c, one, bytes, pose7, known);
      c->h_votes[0] = known;
      if (votes_out) *votes_out = known;
      return RPE_OK;
    }
    double tot[rpe::kSessionHypsMax];
    if (session_batch(c, 1, one, 1, bytes, tot) == RPE_OK) {
      c->h_votes[0] = (int)tot[0];
      if (votes_out) *votes_out = c->h_votes[0];
      return RPE_OK;
    }
  }
  session_end(c);
  const bool m33 = kind == RPE_VOTE_33 || kind == RPE_VOTE_33_23 || kind == RPE_VOTE_NN_33 || kind == RPE_VOTE_NN_33_23;
  const bool m23 = kind == RPE_VOTE_23 || kind == RPE_VOTE_23_MATRIX || kind == RPE_VOTE_33_23 || kind == RPE_VOTE_NN_23 || kind == RPE_VOTE_NN_33_23;
  const bool mnn = kind == RPE_VOTE_NN_23 || kind == RPE_VOTE_NN_33 || kind == RPE_VOTE_NN_33_23;
  if (m23 && (rc = ensure_mask(c, RPE_MOD_23, true))) return rc;
  if (m33 && (rc = ensure_mask(c, RPE_MOD_33, true))) return rc;
  if (mnn && (rc = ensure_mask(c, RPE_MOD_NN, true))) return rc;
  return mask_by_launch(c, kind, mode, pose7, thre_3d, cos_thr, cos_nl, votes_out);
}

// ---------------------------------------------------------------------------------------------- resident scoring session
// Resident scoring session: the batches of ONE RANSAC run (rpe_score with at most 32 hypotheses and exactly these parameters) and the
// winner's masks (rpe_inlier_mask) are served by one resident launch instead of a launch each.  RPE_ERR_STATE if the context cannot
// run one (no large-BAR control block, a sharded context, a problem beyond one group per thread of the co-resident grid): the caller
// simply goes on -- rpe_score / rpe_inlier_mask then launch as always.  Any other call on the context closes the session.
int rpe_score_session_begin(rpe_context* c, int kind, int mode, double thre_3d, double cos_thr, double cos_nl) {
  if (!c) return fail(RPE_ERR_ARG, "null context");
  session_end(c);
  if (t_session_id) { rpe_context* mine = my_open_session(t_session_dev); if (mine) session_close(mine); }   // one session per thread
  int rc = vote_arrays(c, kind);
  if (rc) return rc;
  static const bool off = getenv("RPE_SCORE_SESSION") && atoi(getenv("RPE_SCORE_SESSION")) == 0;
  if (off || !c->resident || !c->host_resident || c->hostex || c->comm || c->p2p_world >= 1 || c->p2p_world_saved >= 1)
    return fail(RPE_ERR_STATE, "no resident scoring session on this context");
  HIP_TRY(hipSetDevice(c->device));
  const int grid = rpe::score_resident_grid(c->arrays(), c->max_blocks);
  if (grid < 1) return fail(RPE_ERR_STATE, "the problem is not frame-sized: no resident scoring session");
  const bool m33 = kind == RPE_VOTE_33 || kind == RPE_VOTE_33_23 || kind == RPE_VOTE_NN_33 || kind == RPE_VOTE_NN_33_23;
  const bool m23 = kind == RPE_VOTE_23 || kind == RPE_VOTE_23_MATRIX || kind == RPE_VOTE_33_23 || kind == RPE_VOTE_NN_23 || kind == RPE_VOTE_NN_33_23;
  const bool mnn = kind == RPE_VOTE_NN_23 || kind == RPE_VOTE_NN_33 || kind == RPE_VOTE_NN_33_23;
  if (m23 && (rc = ensure_mask(c, RPE_MOD_23, true))) return rc;
  if (m33 && (rc = ensure_mask(c, RPE_MOD_33, true))) return rc;
  if (mnn && (rc = ensure_mask(c, RPE_MOD_NN, true))) return rc;
  const int exact = mode == RPE_SCORE_EXACT;
  double thr[3];
  stage_thresholds(c->dtype, exact, thre_3d, cos_thr, cos_nl, thr);
  resident_mutex(c->device).lock();
  const unsigned long long base = c->seq;
  rpe::ReduceTarget rt = host_target(c);
  rt.seq = base;
  rt.h_out = c->h_big;
  if (c->test_pose_wait_s > 0) rt.pose_wait_ticks = (unsigned long long)(c->test_pose_wait_s * 1e8);   // tests: a grid that gives up soon
  c->sess.wait_us = (double)rt.pose_wait_ticks * 0.01;   // (100 MHz clock)
  c->sess.last_us = clock_us();
  const int nacc = rpe::kSessionHypsMax, rgn = 512 / nacc;
  int mult = (grid + rgn * 8 - 1) / (rgn * 8);
  mult = mult < 1 ? 1 : (mult > 4 ? 4 : mult);
  const int runs = resident_run_shape(grid, nacc, 4 * rgn, rgn * mult, &rt);
  c->seq = base;
  const hipError_t e = rpe::launch_score_resident(c->arrays(), kind, exact, (const unsigned long long*)c->ctl, base, thr, grid, rt, c->stream);
  if (e != hipSuccess) { resident_mutex(c->device).unlock(); return fail(RPE_ERR_HIP, "resident scoring launch: %s", hipGetErrorString(e)); }
  c->sess.active = true; c->sess.kind = kind; c->sess.mode = mode; c->sess.grid = grid; c->sess.runs = runs; c->sess.batches = 0;
  c->sess.thre_3d = thre_3d; c->sess.cos_thr = cos_thr; c->sess.cos_nl = cos_nl; c->sess.base = base;
  c->sess.seen_pose.clear(); c->sess.seen_votes.clear();
  c->sess.id = ++g_session_ids;
  session_registered(c, c->sess.id);
  return RPE_OK;
}
int rpe_score_session_end(rpe_context* c) {
  if (!c) return fail(RPE_ERR_ARG, "null context");
  session_close(c);   // (masks of the session are complete in stream order; their record is looked at by the next call that needs to)
  return RPE_OK;
}

// ---------------------------------------------------------------------------------------------- PROSAC order
int rpe_prosac_order(rpe_context* c, const float* weights, int n, int top_k, int* order_out) {
  session_end(c);
  if (!c || !weights || !order_out || n < 1 || top_k < 1) return fail(RPE_ERR_ARG, "rpe_prosac_order: bad argument");
  if (top_k > n) top_k = n;
  if (top_k > rpe::kProsacMaxTopK) return fail(RPE_ERR_ARG,
      "rpe_prosac_order: top_k %d exceeds %d (sort the longer prefix on the host)", top_k, rpe::kProsacMaxTopK);
  HIP_TRY(hipSetDevice(c->device));
  if (c->ps_w_cap < (size_t)n) {
    if (c->ps_w) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->ps_w)); c->ps_w = nullptr; c->ps_w_cap = 0; }
    HIP_TRY(hipMalloc((void**)&c->ps_w, (size_t)n * sizeof(float)));
    c->ps_w_cap = (size_t)n;
  }
  HIP_TRY(hipMemcpyAsync(c->ps_w, weights, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(rpe::launch_prosac_order(c->ps_w, n, top_k, c->ps_hist, c->ps_hist + 2048, c->ps_cand, c->ps_order,
      c->ps_order + rpe::kProsacMaxTopK, c->stream));
  std::vector<int> host((size_t)rpe::kProsacMaxTopK + 1);
  int rc = copy_to_host(c, host.data(), c->ps_order, host.size() * sizeof(int));
  if (rc) return rc;
  const int status = host[(size_t)rpe::kProsacMaxTopK];
  if (status != 0) return fail(RPE_ERR_STATE,
      status == 1 ? "rpe_prosac_order: too many (near-)equal weights around the cut for the device sort; use the host order"
                                                          : "rpe_prosac_order: fewer candidates than top_k");
  std::memcpy(order_out, host.data(), (size_t)top_k * sizeof(int));
  return RPE_OK;
}

// ---------------------------------------------------------------------------------------------- K5
int rpe_nl_round(rpe_context* c, const double* c_opt3, const double* Cw3, const double* Cc3, const double* Rwc9, double* out44) {
  session_end(c);
  int rc = need_arrays(c, {RPE_XW});
  if (rc) return rc;
  // the kernel reads the normal arrays as a PAIR (one without the other would dereference a null pointer on the device)
  if ((c->arr[RPE_NW] != nullptr) != (c->arr[RPE_NC] != nullptr))
    return fail(RPE_ERR_STATE, "rpe_nl_round: NW (normal_g) and NC (normal_c) must be uploaded together (have %s only)",
        c->arr[RPE_NW] ? "NW" : "NC");
  if (!c_opt3 || !Cw3 || !Cc3 || !Rwc9 || !out44) return fail(RPE_ERR_ARG, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  // masks default to all ones, exactly as a freshly constructed adapter
  if ((rc = ensure_mask(c, RPE_MOD_23, true))) return rc;
  if ((rc = ensure_mask(c, RPE_MOD_33, true))) return rc;
  if (c->arr[RPE_NW] && (rc = ensure_mask(c, RPE_MOD_NN, true))) return rc;
  double prm[24];
  for (int i = 0; i < 3; i++) { prm[i] = c_opt3[i]; prm[3 + i] = Cw3[i]; prm[6 + i] = Cc3[i]; }
  for (int i = 0; i < 9; i++) prm[9 + i] = Rwc9[i];
  for (int i = 18; i < 24; i++) prm[i] = 0;
  hipEvent_t e0, e1;
  timing_pair(c, &e0, &e1);
  HIP_TRY(rpe::launch_nl_round(c->arrays(), prm, collect_target(c), c->stream, e0, e1));
  if ((rc = wait_host(c, rpe::kNlLd))) return rc;
  for (int i = 0; i < 44; i++) out44[i] = c->h_out[i];
  return RPE_OK;
}

// ---------------------------------------------------------------------------------------------- Part 3: front end
namespace {
int camera_of(const rpe_camera* cam, rpe::Camera* out) {
  if (!cam || cam->width < 1 || cam->height < 1 || !(cam->fx > 0) || !(cam->fy > 0)
      || (int64_t)cam->width * cam->height > (int64_t)1 << 28)
    return fail(RPE_ERR_ARG, "bad camera (need width, height >= 1 and fx, fy > 0)");
  out->fx = (float)cam->fx; out->fy = (float)cam->fy; out->cx = (float)cam->cx; out->cy = (float)cam->cy;
  out->width = cam->width; out->height = cam->height;
  return RPE_OK;
}
rpe::PoseF pose_f(const double* p12) {
  rpe::PoseF T;
  for (int i = 0; i < 9; i++) T.R[i] = (float)p12[i];
  for (int i = 0; i < 3; i++) T.t[i] = (float)p12[9 + i];
  return T;
}
// (re)allocate `count` float maps of n pixels each
int ensure_maps(rpe_context* c, float** maps, int count, size_t* cap, int64_t n) {
  const size_t bytes = (size_t)n * 3 * sizeof(float);
  if (maps[0] && *cap >= bytes) return RPE_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int i = 0; i < count; i++) { if (maps[i]) { HIP_TRY(hipFree(maps[i])); maps[i] = nullptr; } }
  *cap = 0;
  for (int i = 0; i < count; i++) HIP_TRY(hipMalloc((void**)&maps[i], bytes));
  *cap = bytes;
  return RPE_OK;
}
// the solver slots the association writes: the context's own storage, n = pixels, fp32
int claim_slots(rpe_context* c, int64_t n) {
  const size_t bytes = (size_t)n * 3 * sizeof(float);
  for (int s = 0; s < RPE_NUM_ARRAYS; s++) {
    if (!c->store[s] || c->cap[s] < bytes) {
      if (c->store[s]) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->store[s])); c->store[s] = nullptr; c->cap[s] = 0;
          }
      HIP_TRY(hipMalloc(&c->store[s], bytes));
      c->cap[s] = bytes;
    }
  }
  if (c->n != n || c->dtype != RPE_F32) {  // a different problem was loaded before: its masks / weights do not apply
    for (int i = 0; i < 3; i++) { c->mask[i] = nullptr; c->weight[i] = nullptr; }
  }
  c->n = n; c->dtype = RPE_F32;
  // (the association kernel rewrites them every round, NaN-marking the pixels without a partner: never promoted to "verified")
  for (int s = 0; s < RPE_NUM_ARRAYS; s++) { c->arr[s] = c->store[s]; arrays_changed(c, s, true); }
  return RPE_OK;
}
int associate_launch(rpe_context* c, const double* pose12, double dist_thr, double cos_thr, int use_normals, bool pose_on_device,
    bool count) {
  auto& F = c->fe;
  const int64_t n = (int64_t)F.cam.width * F.cam.height;
  const float d = (float)dist_thr;
  if (count) HIP_TRY(hipMemsetAsync(F.d_count, 0, sizeof(int), c->stream));
  HIP_TRY(rpe::launch_associate(F.fmap[0], F.fmap[1], F.fmap[2], n, F.mmap[0], F.mmap[1], F.mcam, pose_f(pose12), pose_f(F.mpose),
      d * d,
                                (float)cos_thr, use_normals, pose_on_device ? c->d_gn_pose : nullptr,
                                pose_on_device ? &c->d_gn_state->done : nullptr, (float*)c->arr[RPE_XW], (float*)c->arr[RPE_XC],
                                (float*)c->arr[RPE_BV], (float*)c->arr[RPE_NW], (float*)c->arr[RPE_NC], count ? F.d_count : nullptr, c->stream));
  return RPE_OK;
}
int associate_ready(rpe_context* c) {
  if (!c) return fail(RPE_ERR_ARG, "null context");
  if (!c->fe.have_frame) return fail(RPE_ERR_STATE, "no frame: call rpe_frame_set_depth first");
  if (!c->fe.have_model) return fail(RPE_ERR_STATE, "no model: call rpe_model_from_frame or rpe_model_upload first");
  return RPE_OK;
}
}  // namespace

int rpe_frame_set_depth(rpe_context* c, const void* depth, int depth_type, const rpe_camera* cam, double depth_scale, double dmin,
                        double dmax, double max_jump) {
  session_end(c);
  if (!c || !depth || (depth_type != RPE_DEPTH_U16 && depth_type != RPE_DEPTH_F32)) return fail(RPE_ERR_ARG,
      "rpe_frame_set_depth: bad argument");
  rpe::Camera k;
  int rc = camera_of(cam, &k);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  auto& F = c->fe;
  const int64_t n = (int64_t)k.width * k.height;
  const size_t bytes = (size_t)n * (depth_type == RPE_DEPTH_U16 ? 2 : 4);
  if (!F.d_depth || F.depth_cap < bytes) {
    if (F.d_depth) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(F.d_depth)); F.d_depth = nullptr; F.depth_cap = 0; }
    HIP_TRY(hipMalloc(&F.d_depth, bytes));
    F.depth_cap = bytes;
  }
  if (!F.d_count) HIP_TRY(hipMalloc((void**)&F.d_count, 64));
  if ((rc = ensure_maps(c, F.fmap, 3, &F.fcap, n))) return rc;
  F.have_frame = false;
  HIP_TRY(hipMemcpyAsync(F.d_depth, depth, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(rpe::launch_frame_maps(F.d_depth, depth_type, k, (float)depth_scale, (float)dmin, (float)dmax, (float)max_jump, F.fmap[0],
      F.fmap[1],
                                 F.fmap[2], c->stream));
  F.cam = k; F.have_frame = true;
  return RPE_OK;
}

int rpe_frame_download(rpe_context* c, int which, float* out) {
  session_end(c);
  if (!c || !out || which < 0 || which > RPE_MAP_MODEL_NORMAL) return fail(RPE_ERR_ARG, "rpe_frame_download: bad argument");
  auto& F = c->fe;
  const bool model = which >= RPE_MAP_MODEL_VERTEX;
  if (model ? !F.have_model : !F.have_frame) return fail(RPE_ERR_STATE, model ? "no model" : "no frame");
  const rpe::Camera& k = model ? F.mcam : F.cam;
  const float* src = model ? F.mmap[which - RPE_MAP_MODEL_VERTEX] : F.fmap[which];
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(out, src, (size_t)k.width * k.height * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RPE_OK;
}

int rpe_model_from_frame(rpe_context* c, const double* pose12) {
  session_end(c);
  if (!c || !pose12) return fail(RPE_ERR_ARG, "rpe_model_from_frame: bad argument");
  auto& F = c->fe;
  if (!F.have_frame) return fail(RPE_ERR_STATE, "no frame: call rpe_frame_set_depth first");
  HIP_TRY(hipSetDevice(c->device));
  const int64_t n = (int64_t)F.cam.width * F.cam.height;
  int rc = ensure_maps(c, F.mmap, 2, &F.mcap, n);
  if (rc) return rc;
  HIP_TRY(rpe::launch_to_world(F.fmap[0], F.fmap[1], n, pose_f(pose12), F.mmap[0], F.mmap[1], c->stream));
  F.mcam = F.cam;
  std::memcpy(F.mpose, pose12, sizeof(F.mpose));
  F.have_model = true;
  return RPE_OK;
}

int rpe_model_upload(rpe_context* c, const float* vertex_w, const float* normal_w, const rpe_camera* cam, const double* pose12) {
  session_end(c);
  if (!c || !vertex_w || !normal_w || !pose12) return fail(RPE_ERR_ARG, "rpe_model_upload: bad argument");
  rpe::Camera k;
  int rc = camera_of(cam, &k);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  auto& F = c->fe;
  const int64_t n = (int64_t)k.width * k.height;
  if ((rc = ensure_maps(c, F.mmap, 2, &F.mcap, n))) return rc;
  HIP_TRY(hipMemcpyAsync(F.mmap[0], vertex_w, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(F.mmap[1], normal_w, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // the caller may free its buffers on return
  F.mcam = k;
  std::memcpy(F.mpose, pose12, sizeof(F.mpose));
  F.have_model = true;
  return RPE_OK;
}

int rpe_associate(rpe_context* c, const double* pose12, double dist_thr, double cos_thr, int use_normals, int64_t* matched) {
  session_end(c);
  int rc = associate_ready(c);
  if (rc) return rc;
  if (!pose12 || !(dist_thr >= 0)) return fail(RPE_ERR_ARG, "rpe_associate: bad argument");
  HIP_TRY(hipSetDevice(c->device));
  if ((rc = claim_slots(c, (int64_t)c->fe.cam.width * c->fe.cam.height))) return rc;
  if ((rc = associate_launch(c, pose12, dist_thr, cos_thr, use_normals, false, matched != nullptr))) return rc;
  // read-out without a D2H copy or a stream synchronisation: a tiny kernel stores the counter into pinned host memory and raises a
  // sequence word
  if (matched) {
    const unsigned long long seq = ++c->vote_seq;
    HIP_TRY(rpe::launch_publish_i32(c->fe.d_count, 1, c->h_votes, c->h_flag2, seq, c->stream));
    if ((rc = wait_flag(c, c->h_flag2, seq))) return rc;
    *matched = c->h_votes[0];
  }
  return RPE_OK;
}

int rpe_icp(rpe_context* c, const rpe_icp_options* o, double* pose12, int* iters_out, double* last_step, double* final_cost,
    int64_t* matched) {
  session_end(c);
  int rc = associate_ready(c);
  if (rc) return rc;
  if (!o || !pose12 || o->max_iter < 1 || (o->kind != RPE_RES_P2P && o->kind != RPE_RES_P2PLANE) || !(o->dist_thr >= 0))
    return fail(RPE_ERR_ARG, "rpe_icp: bad options (kind must be RPE_RES_P2P or RPE_RES_P2PLANE, max_iter >= 1)");
  if (o->kind == RPE_RES_P2PLANE && !o->use_normals)
    return fail(RPE_ERR_ARG, "rpe_icp: point-to-plane needs use_normals = 1 (pairs without a frame normal would poison the sums)");
  HIP_TRY(hipSetDevice(c->device));
  if ((rc = claim_slots(c, (int64_t)c->fe.cam.width * c->fe.cam.height))) return rc;
  int it = 0;
  double step = 0, cost = 0, pairs = 0;
  bool host_rounds = false;
  auto& F = c->fe;
  const int64_t n = (int64_t)F.cam.width * F.cam.height;
  const float dgate = (float)o->dist_thr;
  // one round's kernels, enqueued on the context's stream
  auto round = [&](const double* pose, const rpe::ReduceTarget& rt, bool pose_on_device) -> int {
    if (o->fused) {
      HIP_TRY(rpe::launch_icp_fused(F.fmap[0], F.fmap[1], n, F.mmap[0], F.mmap[1], F.mcam, pose_f(F.mpose), dgate * dgate,
          (float)o->cos_thr,
                                    o->use_normals, o->kind, pose, rt, c->stream));
      return RPE_OK;
    }
    int r = associate_launch(c, pose, o->dist_thr, o->cos_thr, o->use_normals, pose_on_device, false);
    if (r) return r;
    HIP_TRY(rpe::launch_normal_eq(c->arrays(), o->kind, 0, pose, rt, c->stream));
    return RPE_OK;
  };
  if (o->device_resident) {
    rpe::GnState st;
    st.tol = o->tol; st.step = 0; st.cost = 0; st.max_iters = o->max_iter; st.iters = 0; st.done = 0; st.status = 0;
    HIP_TRY(hipMemcpyAsync(c->d_gn_pose, pose12, 12 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_gn_state, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
    rpe::ReduceTarget rt = host_target(c);
    rt.gn_pose = c->d_gn_pose; rt.gn = c->d_gn_state;
    static const bool auto_on = !(getenv("RPE_DEVICE_LOOP_RESIDENT") && atoi(getenv("RPE_DEVICE_LOOP_RESIDENT")) == 0);
    std::unique_lock<ResidentSlot> one_resident_grid(resident_mutex(c->device), std::defer_lock);
    bool one_launch = false;
    if (auto_on && c->resident && o->fused && o->max_iter >= 2 && !c->hostex && !c->comm
        && c->p2p_world < 1) {
      // ONE launch: the resident grid pairs, sums, solves and updates by itself (icp_resident_kernel with resident_auto_stage)
      one_launch = true;
      one_resident_grid.lock();   // until the result has arrived (end of this block's scope)
      int grid = 0, nacc = 0, max_rows = 1, rows_auto = 1;
      rpe::icp_resident_geometry(n, o->kind, c->max_blocks, &grid, &nacc, &max_rows, &rows_auto);
      const unsigned long long base = c->seq;
      (void)resident_run_shape(grid, nacc, max_rows, rows_auto, &rt);
      c->seq = base + (unsigned long long)o->max_iter + 1;
      rt.seq = c->seq;
      HIP_TRY(rpe::launch_icp_resident(F.fmap[0], F.fmap[1], n, F.mmap[0], F.mmap[1], F.mcam, pose_f(F.mpose), dgate * dgate,
          (float)o->cos_thr, o->use_normals,
                                       o->kind, nullptr, base, o->max_iter, rt, c->stream));
    } else {
      for (int k = 0; k < o->max_iter; k++) if ((rc = round(pose12, rt, true))) return rc;
    }
    if ((rc = wait_host(c, rpe::kNeLd))) return rc;
    if (one_launch && c->h_out[15] == 2.0) {
      // a workgroup's sums never arrived (the grid was not all resident at once): once more from the start pose, one launch per round
      note_lost_grid(c);
      HIP_TRY(hipStreamSynchronize(c->stream));
      HIP_TRY(hipMemcpyAsync(c->d_gn_pose, pose12, 12 * sizeof(double), hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipMemcpyAsync(c->d_gn_state, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
      rt = host_target(c);
      rt.gn_pose = c->d_gn_pose; rt.gn = c->d_gn_state;
      for (int k = 0; k < o->max_iter; k++) if ((rc = round(pose12, rt, true))) return rc;
      if ((rc = wait_host(c, rpe::kNeLd))) return rc;
    }
    for (int i = 0; i < 12; i++) pose12[i] = c->h_out[i];
    step = c->h_out[12]; cost = c->h_out[13]; it = (int)c->h_out[14]; pairs = c->h_out[16];
    if (c->h_out[15] == 2.0) { if (iters_out) *iters_out = it; return fail(RPE_ERR_HIP,
        "ICP device loop: a workgroup's sums never arrived at iteration %d", it); }
    if (c->h_out[15] != 0.0) { if (iters_out) *iters_out = it; return fail(RPE_ERR_DEGENERATE,
        "ICP: normal equations are not positive definite at iteration %d", it - 1); }
  } else if (o->fused && c->resident && c->host_resident && o->max_iter >= 2 && !c->hostex && !c->comm && c->p2p_world_saved < 1) {
    // host-driven ICP in ONE launch: the frame's pixels stay in registers, every iteration the host hands the pose over, the grid pairs
    // its pixels with the model under that pose and sends the run records back (rpe_icp.hip icp_resident_kernel)
    int grid = 0, nacc = 0, max_rows = 1, rows_auto = 1;
    rpe::icp_resident_geometry(n, o->kind, c->max_blocks, &grid, &nacc, &max_rows, &rows_auto);
    auto launch = [&](const rpe::ReduceTarget& rt, unsigned long long base) -> hipError_t {
      return rpe::launch_icp_resident(F.fmap[0], F.fmap[1], n, F.mmap[0], F.mmap[1], F.mcam, pose_f(F.mpose), dgate * dgate,
          (float)o->cos_thr, o->use_normals,
                                      o->kind, (const unsigned long long*)c->ctl, base, o->max_iter, rt, c->stream);
    };
    { std::lock_guard<ResidentSlot> one_resident_grid(resident_mutex(c->device));
      rc = resident_host_loop(c, launch, grid, nacc, max_rows, rows_auto, 1.0, pose12, o->max_iter, o->tol, &it, &step, &cost, &pairs,
          "ICP: normal equations"); }
    if (rc != RPE_OK && rc != kResidentLost) { if (iters_out) *iters_out = it; return rc; }
    host_rounds = rc == kResidentLost;   // the grid was lost after `it` whole rounds: the rest one launch per round
  } else host_rounds = true;
  if (host_rounds) {
    for (; it < o->max_iter; it++) {
      if ((rc = round(pose12, collect_target(c), false))) return rc;
      if ((rc = wait_host(c, rpe::kNeLd))) return rc;
      double ne[32], d[6];
      for (int i = 0; i < 32; i++) ne[i] = c->h_out[i];
      cost = ne[27]; pairs = ne[28];
      if (!rpe::solve_normal_eq6(ne, d, rpe::pivot_floor(c->dtype == RPE_F64))) {
        if (iters_out) *iters_out = it;
        return fail(RPE_ERR_DEGENERATE, "ICP: normal equations are not positive definite at iteration %d (%g pairs)", it, pairs);
      }
      rpe::se3_left_update(d, pose12);
      step = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
      if (step < o->tol) { it++; break; }
    }
  }
  // leave the pairs in the slots
  if (o->fused && (rc = associate_launch(c, pose12, o->dist_thr, o->cos_thr, o->use_normals, false, false))) return rc;
  if (iters_out) *iters_out = it;
  if (last_step) *last_step = step;
  if (final_cost) *final_cost = cost;
  if (matched) *matched = (int64_t)pairs;   // pairs of the last round (the record's weight sum)
  return RPE_OK;
}

}  // extern "C"
