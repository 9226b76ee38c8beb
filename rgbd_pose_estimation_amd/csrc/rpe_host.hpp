// Host side of librgbdpose_hip.so, shared by the units that implement Part 2 / Part 3 of include/rgbd_pose_hip.h:
//   rpe_context.hip       context life cycle, HBM-resident arrays, masks, weights, launch timing
//   rpe_receive.hip       where a launch leaves its result and how the host receives it (run records, flags), clean-first protocol
//   rpe_capi.hip          the thin extern "C" shim: one entry point per kernel (K1', K1-K3, joint, K4, K4b, K5, PROSAC order)
//   rpe_refine.hip        the Gauss-Newton loops (host-driven resident, autonomous, one launch per iteration), host-thread tuning
//   rpe_session.hip       resident scoring sessions (K4r)
//   rpe_dist.hip          sharded contexts: RCCL communicator, in-kernel peer-to-peer, host-side exchange, sharded steps
//   rpe_frontend_api.hip  Part 3: depth-frame front end and ICP
// Everything in namespace rpeh is internal to the library (hidden visibility).  There is NO CPU fallback anywhere behind this header.
#pragma once
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
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>
#include <algorithm>
#include <cctype>
#include <sched.h>
#include <unistd.h>
#include <cstring>

struct rpe_context {
  int device = 0;
  hipStream_t stream = nullptr;
  // resident scoring session (rpe_score_session_begin ... _end): the grid of score_resident_kernel waits for batches in c->ctl
  struct { bool active = false; int kind = 0, mode = 0, grid = 0, runs = 0, batches = 0; double thre_3d = 0, cos_thr = 0, cos_nl = 0;
           unsigned long long base = 0, id = 0, slot_token = 0;   // slot_token: this session's hold on the device's resident slot
           double last_us = 0, wait_us = 2e6;   // host clock of the last message / the grid's bounded wait: a message that comes later
           bool pend_late = false;              // than that finds no grid -- the caller's pause, not a lost grid (nothing is counted)
           // every hypothesis the session has scored (pose as the caller gave it -> votes): the winner's total is known without
           // waiting for the masks' own record
           std::vector<double> seen_pose; std::vector<int> seen_votes;
           // the session's LAST message was "write these masks and leave" and its record has not been looked at yet (session_verify)
           bool pending = false; unsigned long long pend_tag = 0; int pend_votes = 0; double pend_pose[7] = {0, 0, 0, 0, 0, 0, 0}; } sess;
  // guards sess and the control block against the one other party that may touch them: the thread that opened a session and handed
  // the context on, when it closes "its" session from a call on another context (rpe_session.hip close_session_of_this_thread)
  std::recursive_mutex sess_m;
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
  double* d_out = nullptr;       // 64 doubles + kRunSlots x kRunLd doubles of run records (rpe_dist.hip)
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

namespace rpeh __attribute__((visibility("hidden"))) {

// ---- errors: status code + thread-local message (rpe_last_error)
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#define HIP_TRY(expr)                                                                         \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess) return rpeh::fail(RPE_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

inline size_t elem_size(int dtype) { return dtype == RPE_F64 ? 8 : 4; }
inline double clock_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }

// orders the host's stores into BAR-mapped device memory (possibly write-combining): data before tags, tags out at once
inline void store_fence() {
#if defined(__x86_64__)
  __asm__ __volatile__("sfence" ::: "memory");
#else
  __sync_synchronize();
#endif
}

// ---- RCCL, resolved with dlopen / dlsym (no DT_NEEDED on librccl): rpe_dist.hip
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
Rccl& rccl();
#define NCCL_TRY(expr)                                                                                                  \
  do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return rpeh::fail(RPE_ERR_HIP, "%s: %s", #expr, rpeh::rccl().GetErrorString ? rpeh::rccl().GetErrorString(r_) : "rccl error"); } while (0)

// ---- rpe_context.hip
int ensure_mask(rpe_context* c, int mod, bool fill_ones);
int copy_to_host(rpe_context* c, void* dst, const void* d_src, size_t bytes);
int need_arrays(rpe_context* c, std::initializer_list<int> slots);
void timing_pair(rpe_context* c, hipEvent_t* e0, hipEvent_t* e1);   // the event pair of the next timed launch (rpe_timing_enable), or nulls

// ---- rpe_receive.hip: launch targets, the host's side of the result hand-off, clean-first protocol
constexpr int kResidentLost = -1000;   // internal (never returned through the C ABI): the resident grid lost a granule or ended early
constexpr int kResidentDirty = -1001;  // internal: the CLEAN flavour's first record was not finite -- the arrays need the guarded flavour
constexpr int kResidentBusy = -1002;   // internal: a scoring session of another context holds the device's resident slot -- no resident grid now
enum { kArrUnknown = 0, kArrClean = 1, kArrDirty = 2 };
int run_stride_from_env();
rpe::ReduceTarget host_target(rpe_context* c);
rpe::ReduceTarget collect_target(rpe_context* c);
rpe::ReduceTarget device_target(rpe_context* c, double* d_out);
rpe::ReduceTarget device_runs_target(rpe_context* c);   // collecting launch whose run records stay on the device (c->d_out + 64)
int wait_host(rpe_context* c, int ld);
int wait_collect(rpe_context* c, int ld);
int wait_host_partials(rpe_context* c, int grid, int nacc, double* totals, int first_slot = 0, bool resident = false);
int wait_flag(rpe_context* c, unsigned long long* flag, unsigned long long want);
void expand_p2p17(const double* t, double* ne);
unsigned kind_slot_bits(int kind);
bool take_clean(const rpe_context* c, int kind, bool host_verifies);
bool record_finite(const double* rec, int count);
void note_clean_launch(rpe_context* c, int kind, bool finite);
bool take_clean_terms(const rpe_context* c, int bits, bool host_verifies);
void note_clean_terms(rpe_context* c, int bits, bool finite);
void arrays_changed(rpe_context* c, int slot, bool bound);
int kind_arrays(rpe_context* c, int kind);
int check_flags(rpe_context* c, int kind, int flags);

// The exact 3D test is  sqrt(s) < thre_3d  in the array dtype (Eigen norm(), AbsoluteOrientation.hpp:137-138).  The correctly rounded
// square root is monotonic, so the set of s that pass is { s < cut } with cut = the smallest value whose square root reaches the
// threshold; the kernels compare s with `cut` and never take the root.  Found by stepping from thr^2 with the host's own sqrt.
template <class T> inline T sqrt_cut(T thr) {
  if (thr != thr) return thr;                                   // NaN: nothing passes, either way
  if (!(thr > T(0))) return T(0);                               // sqrt(s) < thr <= 0 never holds; s < 0 never holds
  if (std::isinf(thr)) return thr;                              // every finite s passes
  T x = thr * thr;
  if (std::isinf(x)) x = std::numeric_limits<T>::max();
  while (x > T(0) && std::sqrt(x) >= thr) x = std::nextafter(x, T(0));
  while (std::sqrt(x) < thr) x = std::nextafter(x, std::numeric_limits<T>::infinity());
  return x;
}

// ---- rpe_capi.hip: launch helpers other units use
int normal_eq_launch(rpe_context* c, int kind, int flags, const double* pose12, double* d_out32, bool clean);
struct JointSpec { int bits = 0, robust[5] = {0, 0, 0, 0, 0}; double scale[5] = {0, 0, 0, 0, 0}, rk[5] = {1, 1, 1, 1, 1}; };   // by kind 0..4
int joint_spec(rpe_context* c, int nterms, const rpe_term* terms, int flags, const double* pose12, JointSpec* out);
int joint_launch_checked(rpe_context* c, int nterms, const rpe_term* terms, int flags, const double* pose12, bool clean, int* bits_out);
int vote_arrays(rpe_context* c, int kind);
void stage_poses(int dtype, int exact, const double* poses7, int H, void* dst);
void stage_thresholds(int dtype, int exact, double thre_3d, double cos_thr, double cos_nl, double thr[3]);
int mask_by_launch(rpe_context* c, int kind, int mode, const double* pose7, double thre_3d, double cos_thr, double cos_nl, int* votes_out);

// ---- rpe_refine.hip: the per-device resident slot and the host side of a resident loop
// One resident loop per GPU at a time within this process: two resident grids launched together (two contexts, two threads) could each
// get only part of their workgroups onto the CUs and then wait for workgroups that cannot start (the bounded waits would end both with
// an error).  Other PROCESSES on the same GPU are the caller's to serialise.
// The slot is held by a TOKEN, not by a thread: a scoring session holds it from rpe_score_session_begin to whatever call ends it, and
// a context may be handed from one thread to the next in between.  Nobody ever blocks on a session: a resident LOOP holds the slot
// within one call and all its waits are bounded, so a second loop waits for it; a SESSION holds it across calls, so whoever finds one
// in the slot either takes the slot over -- if the session's grid has provably left: no message for longer than the grid's bounded
// wait; the session's owner then finds its token revoked at its next message and goes on with ordinary launches -- or gets no slot
// at all (acquire returns 0: a loop) and runs without a resident grid, one launch per iteration -- or waits its turn (another
// session), at most until the holder's grid would have left.  (Round 5 blocked there without a time-out: a session left open by an
// exception, or an owner waiting for the blocked thread, hung the process.)
struct ResidentSlot {
  std::mutex m; std::condition_variable cv;
  unsigned long long owner = 0, tokens = 0;   // token of the holder (0 = free)
  bool session = false;                       // the holder is a scoring session ...
  double sess_last_us = 0, sess_wait_us = 0;  // ... whose grid leaves by itself sess_wait_us after its last message (host clock)
  unsigned long long acquire(bool as_session, double wait_us = 0) {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      if (owner == 0) break;
      if (session) {
        const double left_us = sess_last_us + sess_wait_us + 5e4 - clock_us();
        if (left_us <= 0) break;      // its grid has left: the slot changes hands
        if (!as_session) return 0;    // a loop does not wait for a session: it runs without a resident grid
        cv.wait_for(lk, std::chrono::microseconds((long long)left_us + 1));   // a session waits its turn -- at most until then
        continue;
      }
      cv.wait(lk);
    }
    owner = ++tokens; session = as_session; sess_last_us = clock_us(); sess_wait_us = wait_us;
    return owner;
  }
  // session holder, before every message: still mine?  (and the grid's wait starts anew)
  bool touch(unsigned long long token) { std::lock_guard<std::mutex> lk(m); if (owner != token) return false; sess_last_us = clock_us(); return true; }
  void release(unsigned long long token) {
    { std::lock_guard<std::mutex> lk(m); if (owner != token || token == 0) return; owner = 0; session = false; }
    cv.notify_all();
  }
};
// a resident loop's hold on the slot for the length of a scope; false = a session is in the slot: run without a resident grid
struct SlotHold {
  ResidentSlot& slot; unsigned long long token;
  explicit SlotHold(ResidentSlot& s) : slot(s), token(s.acquire(false)) {}
  ~SlotHold() { slot.release(token); }
  SlotHold(const SlotHold&) = delete; SlotHold& operator=(const SlotHold&) = delete;
  explicit operator bool() const { return token != 0; }
};
ResidentSlot& resident_mutex(int device);
int resident_run_shape(int grid, int nacc, int max_rows, int rows_auto, rpe::ReduceTarget* rt);
void note_lost_grid(rpe_context* c);

// ---- rpe_session.hip
void session_end(rpe_context* c);      // every entry point that queues work behind the context's stream, reads the masks or reuses the host-side record area calls this first
void session_close(rpe_context* c);
int session_batch(rpe_context* c, int op, const void* staged, int count, size_t bytes, double* totals);
void session_final_masks(rpe_context* c, const void* staged, size_t bytes, const double* pose7, int votes);
bool session_seen(const rpe_context* c, const double* pose7, int* votes);
bool session_matches(const rpe_context* c, int kind, int mode, double thre_3d, double cos_thr, double cos_nl);

// Host side of a RESIDENT loop (rpe_gn_refine, rpe_icp): ONE launch (`launch(rt, base)`) whose grid stays resident; the host hands
// every pose to it through the control block in device memory (two stores' worth of PCIe latency instead of a kernel launch per
// iteration), receives the run records of every iteration, adds them, solves the 6x6 system and applies the SE(3) update, as the
// one-launch-per-iteration loop does.  Pose i carries tag base + i, the records of iteration i carry sequence base + i.
// Cross-workgroup stage: runs of `rows` workgroups are added by the first workgroup of the run (granule hand-off, one hop), the run
// records come to the host, which adds them in run order.  A handful of small records (grid x sums <= 1024 pairs, i.e. a few thousand
// correspondences): rows = 1, every workgroup sends its own record and nothing is handed over on the GPU at all; otherwise one run
// per XCD (eight run records: 136 pairs for point-to-point at 640 x 480), or -- small grids, RPE_RESIDENT_STRIDE=0 -- runs of
// consecutive workgroups, one granule per collecting thread and up to four when that keeps the number of runs at <= 8
// (resident_run_shape).
template <class Launch>
inline int resident_host_loop(rpe_context* c, Launch launch, int grid, int nacc, int max_rows, int rows_auto, double cost_scale,
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

}  // namespace rpeh
