// Internal launcher interface between the host units (rpe_host.hpp: rpe_capi.hip, rpe_refine.hip ...) and the gfx950 kernel units (rpe_normal_eq.hip,
// rpe_icp.hip, rpe_joint.hip, rpe_score.hip, rpe_nl.hip; shared device code: rpe_reduce.hpp, rpe_residuals.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rpe {

constexpr int kBlock = 256;          // 4 wave64 per workgroup
constexpr int kNeLd = 32;            // doubles per normal-equation / moment partial record
constexpr int kNlLd = 64;            // doubles per nl_round partial record
constexpr int kMaxScoreH = 8192;     // hypotheses per scoring launch (LDS vote table = 32 KiB)
// run records of a collecting launch kept on the device for a collective (ReduceTarget: rows > 0 with d_out set): kRunSlots slots of
// kRunLd doubles; a launch of <= 256 workgroups has at most 8 runs (rpe_reduce.hpp collect_and_send), absent ones are zero
constexpr int kRunSlots = 8;
constexpr int kRunLd = 32;

// Correspondence arrays resident in HBM.  3 x n column-major (xyz interleaved), dtype 0 = f32, 1 = f64.
struct DeviceArrays {
  const void* a[5];        // RPE_XW, RPE_XC, RPE_BV, RPE_NW, RPE_NC
  short* mask[3];          // RPE_MOD_23 / 33 / NN (n shorts each) or null
  const void* weight[3];   // n Tp each or null
  int64_t n;
  int dtype;
};

// State of a device-resident Gauss-Newton loop (lives in HBM next to the pose; one launch per iteration, no host round trip).
struct GnState {
  double tol;        // stop when |delta| < tol
  double step, cost; // of the last iteration
  int max_iters;     // stop after this many iterations
  int iters;         // iterations done
  int done;          // 1: converged / failed / max_iters reached -> later launches of the same batch return at once
  int status;        // 0 ok, 1 normal equations not positive definite
};

// Peer-to-peer all-reduce of the 32-double record over xGMI, inside the reduction kernel (one process per GPU, <= 8 ranks).
// Every rank owns a mailbox in fine-grained HBM that all peers map through HIP IPC:
//   mailbox[parity 2][source rank kP2PMaxWorld][64 words], word = { lo 32 bits: one half of a double, hi 32 bits: step tag }.
// The tag travels WITH the data in one 8-byte store (the "LL" flag-in-data protocol), so no ordering between stores is assumed.
constexpr int kP2PMaxWorld = 8;
constexpr int kP2PWords = 64;                                  // 32 doubles = 64 halves
constexpr size_t kP2PRecordWords = (size_t)2 * kP2PMaxWorld * kP2PWords;             // [parity][source rank][64]
constexpr size_t kP2PVoteWords = (size_t)2 * kP2PMaxWorld * kMaxScoreH;              // [parity][source rank][hypothesis]
constexpr size_t kP2PMailboxBytes = (kP2PRecordWords + kP2PVoteWords) * 8;           // records first, vote counters behind
struct P2PDesc {
  int world, rank;
  unsigned long long* peer[kP2PMaxWorld];   // every rank's mailbox as mapped in THIS process; peer[rank] is the own one
};

// Where a reduction kernel leaves its result (both stages run inside one launch, see reduce_and_finish).
struct ReduceTarget {
  double* d_partials;          // max_blocks * kNlLd doubles of scratch
  unsigned int* d_ticket;      // 9 arrival counters 128 B apart (8 shards + top), zero between launches
  int max_blocks;              // cap on workgroups (= partial records)
  int block;                   // workgroup size 256 / 512 / 1024, 0 = default
  double* d_out;               // record in HBM (for a collective), or null
  double* h_out;               // record in pinned host memory + sequence word at [LD], or null
  unsigned long long seq;      // sequence value published after the record
  double* gn_pose = nullptr;   // device-resident GN: 12 doubles in HBM, read at kernel start, updated by the last workgroup
  GnState* gn = nullptr;       // its state (null = ordinary launch: pose from the kernel argument, record published)
  const P2PDesc* p2p = nullptr;   // multi-GPU: exchange + sum the record with the peers before publishing (h_out path only)
  unsigned long long p2p_step = 0;   // collective step counter, identical on every rank (tag + mailbox parity)
  int tail = -1;               // cross-workgroup stage of the ordinary kernels: -1 = default / RPE_TAIL
  // resident kernels: wait for the host's next pose at most this long (100 MHz ticks; 2 s)
  unsigned long long pose_wait_ticks = 200000000ull;
  unsigned long long fault_tag = 0;                    // test hook: see Finish
  double pivot_floor = 1e-12;  // device-side 6x6 solves: relative pivot floor (rpe::pivot_floor of the arrays' dtype, rpe/linalg.hpp)
  // > 0: collecting workgroups + host-side final sum -- runs of up to `rows` workgroups are added by the first workgroup of
  int rows = 0;
  // normal-equation kernels (one launch and resident): the flavour WITHOUT NaN guards.  Only for arrays known or about to be verified
  // to hold finite values: the shim launches it first and repeats the launch in the guarded flavour if the record comes back
  // non-finite (rpe_receive.hip clean-first protocol); results nobody on the host inspects use it only for arrays already verified
  bool clean = false;
  int solver = 0;              // autonomous resident loops: 1 = launch_auto_solver's workgroup sums, solves and hands the poses out
  const double* chain_runs = nullptr;   // chained sharded steps: see Finish (rpe_reduce.hpp)
  double* chain_pose_out = nullptr;
  int stride = 0;              // resident kernels: > 1 = strided runs (see Finish)
                               // the run, the run records go to h_out as tagged pairs (ordinary kernels: behind a header pair; h_out
                               // must hold
                               // 1 + ceil(grid / run length) x sums pairs), and the host adds them in order. Host-consumed, single-GPU
                               // results only
};
// ev_begin / ev_end (optional): the launch goes through hipExtLaunchKernelGGL and the two events receive the dispatch's own begin /
// end timestamps -- what rocprofv3 reports for the kernel (bench roofline timing)
#define RPE_LAUNCH_EV(KERNEL, GRID, BLOCK, SHMEM, STREAM, EV0, EV1, ...)                                              \
  do {                                                                                                                \
    if ((EV0) && (EV1)) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, STREAM, EV0, EV1, 0, __VA_ARGS__);            \
    else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, STREAM, __VA_ARGS__);                                         \
  } while (0)
hipError_t launch_normal_eq(const DeviceArrays& A, int kind, int flags, const double* pose12, const ReduceTarget& rt, hipStream_t s,
                            hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);
// RESIDENT form of the same kernels: one launch serves up to max_iters Gauss-Newton iterations; between iterations every workgroup
// waits for the next pose in `ctl` (16 words in fine-grained device memory written by the host: layout in rpe_residuals.hpp), tagged
// first_tag + i; the run records of iteration i (rt.rows) are published with sequence value rt.seq + i.  Needs host-writable device
// memory (large BAR).
constexpr unsigned long long kResidentStopBit = 1ull << 63;
// = kLostMarker (rpe_reduce.hpp): a run whose granules never arrived
constexpr unsigned long long kResidentLostMarker = 0x7ff8dead00c0ffeeull;
// How many 512-thread workgroups of the resident kernels the CURRENT device holds at once (occupancy of the heaviest instances x
// compute
// units, at most 256; 0 = none fits: no resident loops). Every resident launcher caps its grid with it: a collecting workgroup waits
// for
// workgroups of its own launch, so all of them must be on the compute units together.
int resident_cap_device();
// Each kernel unit is a code object of its own that the runtime loads on the first launch out of it (a few milliseconds, once per
// process and device).  rpe_create touches one kernel of every unit so that the first frame does not pay for it in the middle of a run.
void preload_normal_eq(); void preload_icp(); void preload_joint(); void preload_score(); void preload_nl();
void preload_frontend(); void preload_hypotheses(); void preload_prosac();
void resident_geometry(const DeviceArrays& A, int kind, int max_blocks, int* grid, int* nacc, int* max_rows, int* rows_auto);
// The solving workgroup of an autonomous resident loop (rpe_residuals.hpp solver_loop): ONE workgroup, launched on a stream of its own
// BEFORE the workers' kernel (launch_normal_eq_resident / launch_normal_eq_joint_resident with rt.solver = 1, same rt otherwise); nacc =
// 17 (point-to-point) or 29; workers = the grid of the workers' kernel (resident_geometry).  auto_solver_workers: that grid if the
// solving workgroup applies to it (enough workers, one compute unit to spare, RPE_AUTO_SOLVER != 0), else 0.
int auto_solver_workers(int grid);
int auto_solver_cap();   // the largest workers' grid that leaves the solving workgroup its compute unit
hipError_t launch_auto_solver(int nacc, int workers, unsigned long long first_tag, int max_iters, const ReduceTarget& rt, hipStream_t s);
// false: no resident instance serves this problem (fp64 arrays, 2D-3D kinds, more than one group per thread): one launch per iteration
bool normal_eq_resident_fits(const DeviceArrays& A, int kind, int max_blocks, bool autonomous_no_solver = false);
hipError_t launch_normal_eq_resident(const DeviceArrays& A, int kind, int flags, const unsigned long long* ctl,
    unsigned long long first_tag,
                                     int max_iters, const ReduceTarget& rt, hipStream_t s, hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);
// test hook: one application of the device-resident loop's 6x6 LDL^T solve + SE(3) exp-map update (d_step_ok: |delta|, ok flag)
hipError_t launch_gn_update_probe(const double* d_rec32, double* d_pose12, double* d_step_ok, double pivot_floor, hipStream_t s);
hipError_t launch_moments(const DeviceArrays& A, int flags, const ReduceTarget& rt, hipStream_t s, hipEvent_t ev_begin = nullptr,
                          hipEvent_t ev_end = nullptr);
// chained sharded steps: the LAST solve + update (all-reduced run records of the final step, its pose) and the result to the host as
// tagged pairs: pose (12) | |delta| | cost | iterations | status
hipError_t launch_chain_finish(int kind, const double* d_runs, const double* d_pose, GnState* d_state, double pivot_floor, double* h_pairs,
                               unsigned long long seq, hipStream_t s);
// R1 lsq_pnp: sum of the sine residuals at pose7 (quaternion | t), arrays XW and BV; record = sum | count
hipError_t launch_sine_error(const DeviceArrays& A, const double* pose7, const ReduceTarget& rt, hipStream_t s, hipEvent_t ev_begin = nullptr,
                             hipEvent_t ev_end = nullptr);
// fused joint normal equations: terms = bit set over residual kinds (1 << kind); scale / robust / robust_k indexed by kind
hipError_t launch_normal_eq_joint(const DeviceArrays& A, int terms, int flags, const double* pose12, const double* scale4,
                                  const int* robust4, const double* robust_k4, const ReduceTarget& rt, hipStream_t s,
                                  hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);
// RESIDENT form of the joint kernel (one launch per refinement; control block, tags and run records as launch_normal_eq_resident;
// geometry = resident_geometry of a 29-sum kind)
// the resident form serves frame-sized problems only (one group per thread, the workgroup's slice staged in its LDS): true if this
// term set / size / mask-and-weight use fits; otherwise the refinement runs one launch of the joint kernel per iteration
bool joint_resident_fits(const DeviceArrays& A, int terms, int flags, int max_blocks, bool autonomous, bool clean);
// does the joint kernel of this term set have a CLEAN flavour (no NaN guards)?  rt.clean is ignored where it has none
bool joint_has_clean_flavour(int dtype, int terms);
hipError_t launch_normal_eq_joint_resident(const DeviceArrays& A, int terms, int flags, const double* scale4, const int* robust4,
                                           const double* robust_k4, const unsigned long long* ctl, unsigned long long first_tag, int max_iters,
                                           const ReduceTarget& rt, hipStream_t s);
// d_poses: H x 12 (fast: R row-major, t) or H x 8 (exact: qw qx qy qz tx ty tz pad) values of the array dtype.
// thr: {thre_3d (fast: squared), cos_thr, cos_nl} as doubles holding values of the array dtype.
// d_votes[0..H) must be zero on entry (launch_publish_votes leaves them so)
hipError_t launch_score(const DeviceArrays& A, int kind, int exact, const void* d_poses, int H, const double* thr3, int* d_votes,
                        int max_blocks, hipStream_t s);
// single-launch form for short lists (H <= score_small_cap): the hypotheses, staged in HOST memory in the layout above, travel as a
// kernel
// argument (h_poses), or are read from HBM (d_poses: a device-generated batch; exactly one of the two is non-null); rt must be a
// collecting target (rt.rows > 0); record[h] of the result = votes of hypothesis h
int score_small_cap(int dtype, int exact);
hipError_t launch_score_small(const DeviceArrays& A, int kind, int exact, const void* h_poses, const void* d_poses, int H,
    const double* thr3,
                              const ReduceTarget& rt, hipStream_t s);
// RESIDENT scoring (K4r): one launch serves a whole RANSAC run on resident arrays -- batches of up to kSessionHyps hypotheses (op 0) and
// the winner's masks (op 1) handed over through the control block (layout in rpe_score.hip), the counts back as run records of
// kSessionHyps sums each, published with sequence value rt.seq + batch number.  grid = score_resident_grid (0: not frame-sized).
constexpr int kSessionHypsMax = 32;
constexpr int kSessionCtlWordsMax = 512;   // = the context's 4-KB control block
int score_resident_grid(const DeviceArrays& A, int max_blocks);
hipError_t launch_score_resident(const DeviceArrays& A, int kind, int exact, const unsigned long long* ctl, unsigned long long first_tag,
                                 const double* thr3, int grid, const ReduceTarget& rt, hipStream_t s);
// pose12: fast = R row-major (9) t (3); exact = qw qx qy qz tx ty tz (rest ignored).  The vote total is record[0] of rt.
hipError_t launch_mask(const DeviceArrays& A, int kind, int exact, const double* pose12, const double* thr3, const ReduceTarget& rt,
                       hipStream_t s, hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);
// params24 = c_opt(3) Cw(3) Cc(3) Rwc(9) pad; record = 64 doubles
hipError_t launch_nl_round(const DeviceArrays& A, const double* params24, const ReduceTarget& rt, hipStream_t s,
                           hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);

// copy `count` reduced values from HBM to pinned host memory and then store `seq` to *h_flag (the host spins on it)
hipError_t launch_publish_pairs(const double* d_src, int count, double* h_pairs, unsigned long long seq, hipStream_t s);   // {value, seq} pairs, no flag
// sharded scoring: exchange the `count` vote counters with the peers (same mailbox protocol as the records, one word per
// hypothesis), add them in rank order, publish the totals, zero the counters.  *h_status is set to 1 if a peer timed out.
hipError_t launch_publish_votes_p2p(int* d_votes, int count, const P2PDesc* p2p, unsigned long long step, int* h_dst, int* h_status,
                                    unsigned long long* h_flag, unsigned long long seq, hipStream_t s);
// publish the vote counters to pinned host memory, raise the sequence word, and zero the counters for the next launch_score
hipError_t launch_publish_votes(int* d_votes, int count, int* h_dst, unsigned long long* h_flag, unsigned long long seq, hipStream_t s);
hipError_t launch_publish_i32(const int* d_src, int count, int* h_dst, unsigned long long* h_flag, unsigned long long seq,
    hipStream_t s);

// ---- batched hypothesis generation (rpe_hypotheses.hip): `iters` RANSAC iterations of the 3-point closed form, sampled from the
// PCG32 stream (state, inc) exactly as the host sampler would; poses to d_poses in the scoring layout of `exact`, and to
// h_q7 (pinned, 8 values of the array dtype per iteration: qw qx qy qz tx ty tz valid) for the host's replay
hipError_t launch_gen_shinji(const DeviceArrays& A, unsigned long long state, unsigned long long inc, int iters, int exact,
    void* d_poses,
                             void* h_q7, hipStream_t s);

// FAST-mode generator of the plain-RANSAC solvers with a 4-point sample (tolerance parity): solver 0 = kneip_ransac (P3P), 1 =
// shinji_kneip_ransac (3-point fit, P3P), 2 = nl_kneip_ransac (P3P), 3 = nl_shinji_ransac (3-point fit, nl_2p), 4 =
// nl_shinji_kneip_ransac (3-point fit, P3P, nl_2p): gen_p3p_slots(solver) slots per iteration; same sample stream as the host (4 draws
// per iteration); d_poses in the FAST scoring layout (12 values per slot), h_q7 pinned, 8 values per slot (qw qx qy qz tx ty tz valid)
int gen_p3p_slots(int solver);
hipError_t launch_gen_p3p(const DeviceArrays& A, int solver, unsigned long long state, unsigned long long inc, int iters,
    void* d_poses, void* h_q7,
                          hipStream_t s);

// ---- PROSAC order (rpe_prosac.hip): the first top_k (<= kProsacMaxTopK) positions of "indices by weight descending, ties to the lower
// index" for n float weights in HBM.  d_hist: 2048 uints, zero on entry and on exit; d_ctl: 8 uints; d_cand: kProsacSortCap keys;
// d_status: 0 ok, 1 = more candidates than the LDS sort holds (heavy ties around the cut): use the host order.
constexpr int kProsacSortCap = 8192;
constexpr int kProsacMaxTopK = 4096;
hipError_t launch_prosac_order(const float* d_w, int n, int top_k, unsigned int* d_hist, unsigned int* d_ctl,
    unsigned long long* d_cand, int* d_order,
                               int* d_status, hipStream_t s);

// ---- front end (rpe_frontend.hip): depth frame -> maps -> projective association; fp32 throughout
struct Camera { float fx, fy, cx, cy; int width, height; };
struct PoseF { float R[9]; float t[3]; };   // Xc = R Xw + t, R row-major
// depth_type 0 = uint16 (metres = value * scale), 1 = float32 (metres = value * scale); maps are 3 x (width*height) floats
hipError_t launch_frame_maps(const void* d_depth, int depth_type, const Camera& cam, float scale, float dmin, float dmax,
    float max_jump,
                             float* vmap, float* nmap, float* bmap, hipStream_t s);
hipError_t launch_to_world(const float* vmap, const float* nmap, int64_t n, const PoseF& T, float* vw, float* nw, hipStream_t s);
// d_count (may be null): incremented by the number of associated pixels.  pose_dev (may be null): T read from HBM (12 doubles).
// done (may be null): device flag; when set the launch does nothing.
hipError_t launch_associate(const float* vmap, const float* nmap, const float* bmap, int64_t n, const float* mv, const float* mn,
                            const Camera& mcam, const PoseF& T, const PoseF& M, float dist_sq, float cos_thr, int use_normals,
                            const double* pose_dev, const int* done, float* xw, float* xc, float* bv, float* nw, float* nc, int* d_count,
                            hipStream_t s);
// one ICP round in one kernel: association + normal equations of kind 0 (p2p) / 1 (p2plane, frame normals); record as launch_normal_eq
hipError_t launch_icp_fused(const float* vmap, const float* nmap, int64_t n, const float* mv, const float* mn, const Camera& mcam,
                            const PoseF& M, float dist_sq, float cos_thr, int use_normals, int kind, const double* pose12, const ReduceTarget& rt,
                            hipStream_t s);

// resident ICP loop (one launch; poses through the control block, run records to the host -- as launch_normal_eq_resident)
void icp_resident_geometry(int64_t n, int kind, int max_blocks, int* grid, int* nacc, int* max_rows, int* rows_auto);
hipError_t launch_icp_resident(const float* vmap, const float* nmap, int64_t n, const float* mv, const float* mn, const Camera& mcam,
    const PoseF& M,
                               float dist_sq, float cos_thr, int use_normals, int kind, const unsigned long long* ctl, unsigned long long first_tag,
                               int max_iters, const ReduceTarget& rt, hipStream_t s);

}  // namespace rpe
