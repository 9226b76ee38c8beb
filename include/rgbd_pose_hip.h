/* rgbd_pose_hip.h -- C ABI of librgbdpose_hip.so, the MI355X (gfx950) backend of the RGB-D absolute-pose
 * hot path of ShudaLi/rgbd_pose_estimation.  Plain pointers and sizes only; no C++ / torch types.
 *
 * Part 1 is the reference's own FFI, byte for byte (reference Library.cpp:15-82 -> libabsolute.so).
 * Part 2 is additive: a handle-based API over correspondence arrays that stay resident in HBM, one entry
 * point per hot loop of the reference plus the Gauss-Newton formulation the north
 * star asks for.  Every function returns 0 on success or a negative rpe_status; rpe_last_error() explains.
 *
 * Conventions (same as the reference): Xc = R_cw * Xw + t (pose/AbsoluteOrientation.hpp:51); 3 x N arrays
 * are column-major = N packed xyz triples (Eigen Map<MatrixXf>(p,3,n), Library.cpp:20-22); rotation
 * matrices cross this boundary ROW-major (Library.cpp:35-39); masks are short 0/1 (N x cols column-major,
 * column 0 = 2D-3D, 1 = 3D-3D, 2 = normal-normal: pose/NormalAOPoseAdapter.hpp:179-195).
 */
#ifndef RGBD_POSE_HIP_H
#define RGBD_POSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Part 1 -- drop-in replacements for libabsolute.so
 * ---------------------------------------------------------------------------------------------- */

/* Replaces ao() (Library.cpp:17-45): closed-form 3D-3D absolute orientation over ALL n correspondences
 * (AOOnlyPoseAdapter + shinji_ls2, pose/AbsoluteOrientation.hpp:322-342).  x_w_, x_c_: n xyz float triples
 * (host).  R_cw_[9] row-major, t_[3].  The moment sums run on the GPU with fp64 accumulation; the 3x3 SVD
 * on the host.  Prints "ao()" like the reference unless RPE_QUIET=1.  Aborts like the reference
 * (SOPHUS_ENSURE) only if no HIP device is usable: then it prints the reason and calls abort(). */
void ao(float* x_w_, float* x_c_, int n_, float* R_cw_, float* t_);

/* Replaces ao_ransac() (Library.cpp:47-75): shinji_ransac2 (Iter=1000, thre_3d=0.1, confidence=0.99999,
 * pose/AbsoluteOrientation.hpp:158-213) with the vote loop scored on the GPU in hypothesis batches, then
 * shinji_ls1 over the inliers (:298-320).  Sampling uses the documented Rand31 stream (seed RPE_SEED, default 1)
 * instead of the reference's unseeded rand(). */
void ao_ransac(float* x_w_, float* x_c_, int n_, float* R_cw_, float* t_);

/* Replaces py2c() (Library.cpp:77-81): prints N floats, one per line. */
void py2c(float* array, int N);

/* ------------------------------------------------------------------------------------------------
 * Part 2 -- additive handle-based API
 * ---------------------------------------------------------------------------------------------- */

typedef enum {
  RPE_OK = 0,
  RPE_ERR_NO_DEVICE = -1,     /* no HIP device / HIP runtime error: the product has NO CPU fallback */
  RPE_ERR_HIP = -2,
  RPE_ERR_ARG = -3,
  RPE_ERR_STATE = -4,         /* a required array was never uploaded / bound */
  RPE_ERR_DEGENERATE = -5,    /* normal equations not positive definite, or NaN result.  A pivot at or below 16 eps of the ARRAYS' dtype x its
                               * diagonal entry (9.5e-7 for fp32 arrays, 1e-12 for fp64) counts: the cancelled pivots of rank-deficient sets -- one
                               * repeated point, a line, a single plane seen point-to-plane -- are rounding noise of the products, of either sign */
  RPE_ERR_ALIGN = -6          /* bound device pointer not 16-byte aligned */
} rpe_status;

typedef struct rpe_context rpe_context;

int rpe_abi_version(void);
const char* rpe_last_error(void);
int rpe_device_count(void);            /* number of usable HIP devices (0 on a CPU-only host) */

/* stream: a hipStream_t the caller owns (e.g. torch's current stream), or NULL for a private stream. */
int rpe_create(rpe_context** out, int device, void* stream);
void rpe_destroy(rpe_context* ctx);
int rpe_synchronize(rpe_context* ctx);

/* dtype of the correspondence arrays: Tp of the reference's templates */
enum { RPE_F32 = 0, RPE_F64 = 1 };
/* array slots.  Names follow the adapters' members. */
enum {
  RPE_XW = 0,   /* points_g       world points         PnPPoseAdapter.hpp:100        */
  RPE_XC = 1,   /* points_c       camera points        AOPoseAdapter.hpp:95 (NaN column = invalid, :147-152) */
  RPE_BV = 2,   /* bearingVectors unit bearings        PnPPoseAdapter.hpp:98         */
  RPE_NW = 3,   /* normal_g       world normals        NormalAOPoseAdapter.hpp:95    */
  RPE_NC = 4,   /* normal_c       camera normals       NormalAOPoseAdapter.hpp:94    */
  RPE_NUM_ARRAYS = 5
};
/* per-modality inlier masks (short) and weights (Tp), index = mask column */
enum { RPE_MOD_23 = 0, RPE_MOD_33 = 1, RPE_MOD_NN = 2 };

/* Declare the correspondence count and dtype; (re)allocates nothing until an upload. */
int rpe_set_problem(rpe_context* ctx, int64_t n, int dtype);
/* Copy a host array (3 x n, dtype of the problem) into HBM.  Asynchronous on the context stream. */
int rpe_upload(rpe_context* ctx, int slot, const void* host);
/* Copy array `slot` back to the host (3 x n of the problem's dtype); synchronises.  For arrays produced on the device
 * (rpe_associate) and for tests. */
int rpe_download(rpe_context* ctx, int slot, void* host);
/* Use a buffer that already lives in HBM (must be 16-byte aligned, 3*n elements); no copy, not owned. */
int rpe_bind(rpe_context* ctx, int slot, const void* device_ptr);
/* n shorts (0/1) / n weights for one modality; host pointers, NULL clears. */
int rpe_upload_mask(rpe_context* ctx, int modality, const short* host_mask);
int rpe_upload_weight(rpe_context* ctx, int modality, const void* host_weight);
/* Copy the device mask written by rpe_inlier_mask back to the host (n shorts). */
int rpe_download_mask(rpe_context* ctx, int modality, short* host_mask);

/* ---- K1' closed-form moments: the two passes of shinji() (AbsoluteOrientation.hpp:56-73) fused into ONE
 * pass.  out[18] = { sum w, sum w*Xw (3), sum w*Xc (3), sum w*Xc*Xw^T (9, row-major), sum w*|Xc|^2, count of
 * contributing correspondences }.
 * flags: RPE_USE_MASK -> only mask33 == 1 (shinji_ls/shinji_ls1 inlier set, :279-288); RPE_USE_WEIGHT ->
 * w = weight33 (nl_shinji_kneip_ls centroid pass, AbsoluteOrientationNormal.hpp:457-469);
 * RPE_SKIP_INVALID -> skip NaN columns (isValid).  fp32 inputs are widened, all arithmetic is fp64. */
enum { RPE_USE_MASK = 1, RPE_USE_WEIGHT = 2, RPE_SKIP_INVALID = 4 };
int rpe_p2p_moments(rpe_context* ctx, int flags, double* out18);
/* Closed-form pose from the moments (host: 3x3 SVD, det fix, t = Cc - R*Cw; AbsoluteOrientation.hpp:75-95). */
int rpe_pose_from_moments(const double* m18, double* R9, double* t3);

/* ---- R1 lsq_pnp (P3P.hpp:472-502): the sum over ALL correspondences of the sine of the angle between predicted and observed
 * bearing, sum_i | normalize(R*Xw_i + t) x bv_i | (what PnPPoseAdapter::getError(i) returns, PnPPoseAdapter.hpp:204-210).
 * pose7 = unit quaternion (w, x, y, z) | t, rounded to the array dtype; every term is evaluated in the array dtype by the
 * reference's own operation sequence (the reference's bits), the terms are added in fp64 -- the reference adds them one after
 * the other in Tp, so its printed total differs from *sum_out by its own accumulated rounding only.  Arrays XW, BV (24 B/corr fp32).
 * count_out (optional): the number of terms. */
int rpe_sine_error_sum(rpe_context* ctx, const double* pose7, double* sum_out, int64_t* count_out);

/* ---- K1/K2/K3 Gauss-Newton normal equations (new formulation; objective of K1 == shinji()).
 * kind: residual.  pose12 = R row-major (9) | t (3).  out32: H upper triangle row-major (21) | g (6) |
 * sum w r^2 | sum w | 3 pad.  Tangent order (upsilon, omega), update T <- exp(delta)*T (sophus/se3.hpp:314-342).
 * The pose enters the kernel in fp64; p = R*Xw + t and the residual are formed in fp64 (the subtraction
 * cancels ~3 digits), the products in the array dtype, the sums in fp64. */
enum {
  RPE_RES_P2P = 0,      /* r = R*Xw + t - Xc                         (3)  arrays XW, XC      24 B/corr fp32 */
  RPE_RES_P2PLANE = 1,  /* r = Nc . (R*Xw + t - Xc)                  (1)  arrays XW, XC, NC  36 B/corr      */
  RPE_RES_BEARING = 2,  /* r = normalize(R*Xw + t) x bv  (P3P.hpp:482-485) (3)  arrays XW, BV  24 B/corr      */
  RPE_RES_NORMAL = 3,   /* r = R*Nw - Nc   (alignment scored at AbsoluteOrientationNormal.hpp:248) (3)  arrays NW, NC  24 B/corr;
                           rotation only.  Served by the joint kernel (rpe_normal_eq_joint); mask / weight of modality NN */
  RPE_RES_REPROJ = 4    /* 2D-3D pixel reprojection, r = (p_x/p_z - bv_x/bv_z, p_y/p_z - bv_y/bv_z), p = R*Xw + t   (2)  arrays XW, BV
                           24 B/corr; mask / weight of modality 23.  The pixel conversion of TestMain.cpp:35-36 with the principal point at
                           the origin (PoseAdapterBase.hpp:44), in NORMALISED image coordinates: the focal length multiplies r and J alike,
                           so the step does not depend on it -- scale the term by f^2 (rpe_term.scale, `scales` of rpe_gn_refine) for H, g
                           and the cost in pixels.  Correspondences with p_z <= 1e-6 or bv_z <= 1e-6 (not in front of the camera) contribute
                           nothing and do not count.  Alternative to RPE_RES_BEARING for the 2D-3D term of the joint kernel. */
};
int rpe_normal_eq(rpe_context* ctx, int kind, int flags, const double* pose12, double* out32);
/* Same, result left in HBM at d_out32 (32 doubles, must not be NULL) for a caller-side collective (RCCL
 * all-reduce); asynchronous on the context's stream. */
int rpe_normal_eq_device(rpe_context* ctx, int kind, int flags, const double* pose12, double* d_out32);
/* Host: solve H*delta = -g (LDL^T); RPE_ERR_DEGENERATE if H is not positive definite.  ne32[29], as rpe_normal_eq* fill it in, is the
 * relative pivot floor that goes with the record's product dtype (0 = 1e-12). */
int rpe_gn_solve(const double* ne32, double* delta6);
/* Host: pose <- exp(delta) * pose  (Sophus SE3::exp, sophus/se3.hpp:321-342). */
int rpe_gn_apply(const double* delta6, double* pose12);
/* ---- fused joint normal equations: up to four residual kinds (at most one of P2P / P2PLANE) in ONE pass over the arrays,
 * each term with its modality's inlier mask (RPE_USE_MASK) and weights (RPE_USE_WEIGHT), a scale, and an optional robust
 * IRLS weight on the residual-block norm s: Huber min(1, k/s) or Cauchy 1/(1 + (s/k)^2).  out32 as rpe_normal_eq, with
 * cost = sum scale*w*r^2 and the last entry = number-weighted count over all terms.  This is the single-kernel form of the
 * joint 2D-3D + 3D-3D + N-N objective the reference's nl_shinji_kneip_ls alternates over (:484-510). */
enum { RPE_ROBUST_NONE = 0, RPE_ROBUST_HUBER = 1, RPE_ROBUST_CAUCHY = 2 };
typedef struct { int kind; double scale; int robust; double robust_k; } rpe_term;
int rpe_normal_eq_joint(rpe_context* ctx, int nterms, const rpe_term* terms, int flags, const double* pose12, double* out32);
int rpe_gn_refine_joint(rpe_context* ctx, int nterms, const rpe_term* terms, int flags, double* pose12, int max_iter, double tol,
                        int* iters_out, double* last_step, double* final_cost);

/* Device-resident variant of rpe_gn_refine_joint: pose and loop state stay on the GPU, which solves the 6x6 system (LDL^T) and
 * applies the SE(3) exp-map update itself; the host waits once.  One GPU and a single plain point-to-point or point-to-plane term
 * (scale 1, no robust weight): ONE launch whose resident grid iterates by itself until |delta| < tol or max_iter.  Otherwise one
 * launch per iteration whose last workgroup solves (the host enqueues max_iter launches; those after convergence return
 * immediately).  Same arithmetic as the host loop.  RPE_DEVICE_LOOP_RESIDENT=0 selects the per-iteration form everywhere.
 * After rpe_p2p_init the loop is SHARDED: every launch's last workgroup first exchanges and sums the record with its peers, so
 * the refinement stays one launch per iteration on any number of GPUs of a node (collective: all ranks call it alike). */
int rpe_gn_refine_device(rpe_context* ctx, int nterms, const rpe_term* terms, int flags, double* pose12, int max_iter, double tol,
                         int* iters_out, double* last_step, double* final_cost);

/* Test hook for the device-resident loop: ONE application, ON THE GPU, of what its last workgroup does with a record -- the register
 * LDL^T solve of H delta = -g and pose <- exp(delta) * pose with the kernel's own SE(3) exponential (sophus/se3.hpp:321-342) -- to a
 * record and pose of the caller's.  RPE_ERR_DEGENERATE where the device solve refuses the system. */
int rpe_debug_device_gn_update(rpe_context* ctx, const double* ne32, double* pose12, double* step_norm);

/* One Gauss-Newton step on one GPU (kernel -> D2H of the 32-double record -> solve -> exp-map update of pose12).
 * ne32_out / step_norm may be NULL. */
int rpe_gn_step(rpe_context* ctx, int kind, int flags, double* pose12, double* ne32_out, double* step_norm);
/* ---- multi-GPU: one process per GPU, correspondences sharded by contiguous index ranges, pose replicated.
 * rank 0 obtains a 128-byte id (rpe_comm_unique_id) and hands it to every rank by any means (MPI, torch.distributed,
 * a file); each rank then calls rpe_comm_init on its context (RCCL communicator on that context's device; librccl is
 * resolved at run time).  rpe_gn_step_dist = rpe_gn_step with ONE in-place all-reduce(sum) of the 32-double record over
 * RCCL/xGMI between the kernel and the host solve; every rank ends the step with the same pose.  With a communicator set,
 * rpe_score all-reduces the H int32 vote counters the same way. */
int rpe_comm_unique_id(void* id128);
int rpe_comm_init(rpe_context* ctx, int world, int rank, const void* id128);
int rpe_comm_destroy(rpe_context* ctx);
/* ranks of the context's RCCL communicator as the communicator reports them (ncclCommCount; 0 = none), and the PCI bus id of the
 * context's GPU (one process per GPU: every rank of a node reports a different one) */
int rpe_comm_count(rpe_context* ctx, int* ranks);
int rpe_device_bus_id(rpe_context* ctx, char* buf, int len);
int rpe_gn_step_dist(rpe_context* ctx, int kind, int flags, double* pose12, double* ne32_out, double* step_norm);
/* `steps` such steps in one call (every rank passes the same count). */
int rpe_gn_steps_dist(rpe_context* ctx, int kind, int flags, double* pose12, int steps, double* last_step_norm);
/* The same `steps` steps over the RCCL communicator with the HOST OUT OF THE LOOP: every launch takes its pose from the launch before
 * it (each workgroup adds that step's all-reduced run records, solves the 6x6 system and applies the exp-map itself), so the calling
 * thread enqueues steps x {kernel, ncclAllReduce} plus one finishing kernel and waits once -- a launch's latency overlaps the kernels
 * in front of it instead of adding to every step.  The SE(3) update runs on the device (the device-resident loops' solve, equal to the
 * host's to 1e-13); rpe_gn_steps_dist is the form with the exp-map on the host.  Needs rpe_comm_init; every rank passes the same
 * arguments and ends with the same pose.  RPE_ERR_DEGENERATE if a step's normal equations are not positive definite. */
int rpe_gn_steps_dist_device(rpe_context* ctx, int kind, int flags, double* pose12, int steps, double* last_step_norm);
/* Peer-to-peer variant of the collective for ONE node (<= 8 ranks): instead of RCCL, the normal-equation kernel's last
 * workgroup writes the 32-double record straight into a mailbox of every peer over xGMI (HIP IPC mappings, flag-in-data
 * words), waits for the peers' records in its own mailbox, adds them in rank order and publishes the sum -- the whole sharded
 * step is ONE kernel launch, and every rank gets bitwise the same record.  Each rank calls rpe_p2p_export (64-byte IPC handle),
 * all handles are gathered by any means (world x 64 bytes, rank order), each rank calls rpe_p2p_init; rpe_gn_step_dist then
 * uses this path (it takes precedence over an RCCL communicator), and rpe_score exchanges and sums its vote counters the same
 * way.  A rank that waits more than 10 s for a peer fails the step with RPE_ERR_HIP instead of hanging; ranks should therefore
 * enter their first exchange together (one local launch + a barrier).  Callers barrier before rpe_p2p_destroy. */
int rpe_p2p_export(rpe_context* ctx, void* handle64);
int rpe_p2p_init(rpe_context* ctx, int world, int rank, const void* handles);
/* pause = 1 keeps the mailboxes but routes rpe_gn_step_dist / rpe_score through the RCCL communicator (stand-by); 0 resumes. */
int rpe_p2p_pause(rpe_context* ctx, int pause);
int rpe_p2p_destroy(rpe_context* ctx);
/* What the exact scoring kernels compare the SQUARED 3D residual with (dtype RPE_F32: evaluated in float): the smallest value whose
 * correctly rounded square root reaches thre_3d, so that  sqrt(s) < thre_3d  <=>  s < cut  for every s (test hook; no GPU involved). */
double rpe_host_sqrt_cut(int dtype, double thre_3d);
/* Host-side exchange for ONE node (the third way to all-reduce; csrc/rpe_hostex.cpp): on one GPU a reduction's final sum already
 * happens on the host (a few run records per launch, added by the calling thread), so with sharded correspondences every rank's host
 * thread holds its shard's record microseconds after its kernel -- and the rank processes share the node's memory.  The records are
 * exchanged between the host threads through a POSIX shared-memory segment and added in RANK ORDER (bitwise the same sums on every
 * rank): no collective kernel, no GPU-side wait for a peer.  rpe_hostex_init(ctx, world, rank, name, create): `name` ("/...") is
 * agreed by any means; exactly one rank passes create = 1 and must do so before the others open (they wait up to the time-out for the
 * segment to appear).  With an exchange set,
 * rpe_gn_step_dist / rpe_gn_steps_dist / rpe_gn_refine are sharded steps (the exchange takes precedence over rpe_p2p_* and
 * rpe_comm_*), rpe_gn_refine keeps its RESIDENT kernel per rank (one GPU per rank; ranks that share a GPU fall back to one launch per
 * iteration, because two resident grids that wait for each other's hosts cannot both be resident), and rpe_score adds the vote counters
 * the same way.  Every wait is bounded (10 s): a missing peer fails the call with RPE_ERR_STATE. */
int rpe_hostex_init(rpe_context* ctx, int world, int rank, const char* name, int create);
int rpe_hostex_destroy(rpe_context* ctx);
/* The exchange by itself (no GPU involved; what the two calls above wrap): */
typedef struct rpe_host_exchange rpe_host_exchange;
int rpe_host_exchange_open(const char* name, int world, int rank, int create, double timeout_s, rpe_host_exchange** out);
int rpe_host_exchange_allreduce_f64(rpe_host_exchange* h, double* v, int n);   /* in place, 1 <= n <= 64, sums in rank order */
int rpe_host_exchange_allreduce_i32(rpe_host_exchange* h, int* v, int n);      /* in place, 1 <= n <= 8192 */
int rpe_host_exchange_set_label(rpe_host_exchange* h, const char* label);      /* e.g. the rank's GPU (PCI bus id) */
int rpe_host_exchange_labels_collide(rpe_host_exchange* h);                    /* after an exchange: do two ranks carry the same label? */
int rpe_host_exchange_unlink(rpe_host_exchange* h);                            /* drop the name once every rank has opened it */
void rpe_host_exchange_close(rpe_host_exchange* h);
/* Whole refinement loop on one GPU: up to 3 residual kinds summed with scales; stops when |delta| < tol.
 * iters_out = iterations run; returns RPE_ERR_DEGENERATE if a solve failed. */
int rpe_gn_refine(rpe_context* ctx, int nterms, const int* kinds, const double* scales, int flags, double* pose12, int max_iter,
                  double tol, int* iters_out, double* last_step, double* final_cost);

/* Host-clock profile of rpe_gn_refine's resident loop (one GPU): enable = 1 clears and starts; enable = 0 stops and returns the sums, in
 * microseconds over `steps` steady-state iterations, of the host's WAIT for a record (pose hand-over in flight + one iteration of the
 * resident kernel + record in flight) and of the host's own turn (6x6 solve + SE(3) update + hand-over stores). */
int rpe_debug_loop_profile(rpe_context* ctx, int enable, double* wait_us, double* host_us, long long* steps);

/* State of a context's RESIDENT loops (rpe_gn_refine, rpe_gn_refine_joint, rpe_icp, rpe_gn_refine_device run as ONE launch whose grid
 * must be on the compute units all at once).  enabled: large-BAR device, at least one workgroup of the resident kernels per compute
 * unit, and fewer than two lost grids so far; lost: refinements whose grid lost a workgroup's sums (another process on the GPU, a
 * partition smaller than the occupancy query promised) and that were FINISHED with one launch per iteration -- such a call still
 * succeeds, a context that sees it twice stops using resident loops; cap: workgroups of a resident kernel the device holds at once
 * (occupancy x compute units, at most 256; RPE_RESIDENT_CAP lowers it).  enabled is a bit set: 1 resident loops, 2 host-driven ones
 * (large BAR), 4 the autonomous loops still ask for their solving workgroup (cleared once the two kernels did not meet). */
int rpe_debug_resident_state(rpe_context* ctx, int* enabled, int* lost, int* cap);
/* Test hook, per context: the last workgroup of the next host-driven resident loops withholds its sums of `iteration` (> 0), and the
 * workgroups wait `pose_wait_s` seconds (0.5 .. 60; 0 = default 2 s) for the next pose.  (0, 0) = off.  Nothing in the library reads
 * a fault from the environment. */
int rpe_debug_inject_resident_fault(rpe_context* ctx, int iteration, double pose_wait_s);

/* Which CPU should the thread that drives the resident Gauss-Newton loop sit on?  (It spins on every iteration's records and writes
 * every pose through the PCIe BAR: the choice is worth 5-10 % of a step.)  Measures a handful of candidates -- the current CPU, three
 * spread over the GPU-local CPUs (sysfs local_cpulist of the device), two over the others, one SMT sibling -- with `reps` refinements of
 * `steps` iterations each over the context's OWN arrays (kind / flags as rpe_gn_refine, tol = 0, from pose12, which is not modified),
 * then leaves the CALLING THREAD pinned to the fastest (sched_setaffinity, this thread only) and reports it: best_cpu / best_us (us per
 * iteration), and up to `cap` (cpu, us) trials.  Opt-in; the environment RPE_HOST_CPU=auto makes the first host-driven resident
 * refinement of every context do this by itself (200 iterations x 5 per candidate), RPE_HOST_CPU=<cpu> pins without measuring.
 * HSA_ENABLE_INTERRUPT=0 in the process environment (a ROCm runtime knob: completion signals polled instead of interrupt-driven,
 * -0.25 us per step of a short refinement) is the caller's choice: it must be set before the runtime initialises. */
int rpe_tune_host_thread(rpe_context* ctx, int kind, int flags, const double* pose12, int steps, int reps, int* best_cpu, double* best_us,
                         int* trial_cpus, double* trial_us, int cap, int* ntrials);

/* HIP-event timing of the one-launch reduction kernels (rpe_normal_eq*, rpe_p2p_moments, rpe_nl_round, rpe_inlier_mask), on the
 * context's stream: after enable(max_records, stride) every stride-th such call launches its kernel with an event pair that receives the dispatch's own begin / end
 * timestamps (hipExtLaunchKernelGGL: what rocprofv3 reports for the kernel, no marker packets); collect() synchronises,
 * returns the number of pairs and their total / minimum elapsed milliseconds, and rearms.  enable(0, 1) = off. */
int rpe_timing_enable(rpe_context* ctx, int max_records, int stride);
int rpe_timing_collect(rpe_context* ctx, int* count, double* total_ms, double* min_ms);
/* Elapsed time an EMPTY event pair reports on this context's stream (average and minimum over `pairs` pairs): what the pair
 * itself adds to every interval rpe_timing_collect returns (2-5 us on MI355X), so that event-based kernel times can be
 * compared with rocprofv3's dispatch timestamps. */
int rpe_timing_calibrate(rpe_context* ctx, int pairs, double* avg_ms, double* min_ms);

/* ---- K4 batched hypothesis scoring: the vote loops V1..V8.
 * kind selects the modality set exactly as the reference's loops combine them. */
enum {
  RPE_VOTE_33 = 0,        /* shinji_ransac / ransac2 / prosac      AbsoluteOrientation.hpp:133-143,190-200,248-258 */
  RPE_VOTE_23 = 1,        /* kneip_ransac / prosac                 P3P.hpp:362-376,439-453                         */
  RPE_VOTE_33_23 = 2,     /* shinji_kneip_ransac / prosac          AbsoluteOrientation.hpp:403-422,480-499         */
  RPE_VOTE_NN_23 = 3,     /* nl_kneip_ransac                       AbsoluteOrientationNormal.hpp:245-264           */
  RPE_VOTE_NN_33 = 4,     /* nl_shinji_ransac                      :322-337                                        */
  RPE_VOTE_NN_33_23 = 5,  /* nl_shinji_kneip_ransac                :397-423                                        */
  RPE_VOTE_23_MATRIX = 6  /* kneip_ransac multiplies by so3().matrix() (P3P.hpp:365) where kneip_prosac uses so3()*x (:442):
                             differs from RPE_VOTE_23 only in RPE_SCORE_EXACT arithmetic                     */
};
/* arithmetic: RPE_SCORE_FAST = rotation-matrix FMA form, squared-distance compare (results can differ from
 * the reference only for correspondences within rounding of a threshold); RPE_SCORE_EXACT = the reference's
 * own operation sequence in Tp (quaternion rotate, sqrt, divide, no FMA contraction): votes bit-identical
 * to the CPU path. */
enum { RPE_SCORE_FAST = 0, RPE_SCORE_EXACT = 1 };
/* poses7: H x (qw qx qy qz tx ty tz) doubles holding Tp-representable values (a Sophus::SE3<Tp>).
 * thre_3d in metres; cos_thr = cos(atan(thre_2d/f)); cos_nl = cos(nl_thre), already evaluated in Tp.
 * votes_out[H]: total votes per hypothesis (sum over the modalities of `kind`). */
int rpe_score(rpe_context* ctx, int kind, int mode, const double* poses7, int H, double thre_3d, double cos_thr, double cos_nl,
              int* votes_out);
/* One batch of `iters` 3D-3D RANSAC iterations entirely on the device: per iteration a thread draws the 3-point sample from the
 * PCG32 stream (rng_state, rng_inc: rpe::Rand31's state, Utility.hpp) at its own position (3 draws per iteration, skip-ahead), runs
 * the closed-form fit shinji() (AbsoluteOrientation.hpp:47-99) on the resident arrays, and the batch is scored by K4 without the
 * poses ever leaving HBM.  Bitwise the hypotheses, in the order, the host sampler + solver would have produced (same functions,
 * no FMA contraction).  votes_out[iters]; q7_out[iters x 7] = qw qx qy qz tx ty tz holding Tp values; valid_out[i] = 0 where the
 * sample hit an invalid (all-NaN) camera point and the reference skips the iteration.  The caller advances its stream by 3*iters. */
int rpe_ransac33_batch(rpe_context* ctx, uint64_t rng_state, uint64_t rng_inc, int iters, int mode, double thre_3d, int* votes_out,
                       double* q7_out, unsigned char* valid_out);
/* The same for the plain-RANSAC solvers with a 4-point sample, FAST scoring mode only (the device solvers agree with
 * the host's to rounding, not bit for bit -- the vote-exact default keeps the host generators).  solver: 0 = kneip_ransac (one slot
 * per iteration: the P3P branch that best reprojects the 4th sample), 1 = shinji_kneip_ransac (3-point fit, P3P), 2 = nl_kneip_ransac
 * (P3P), 3 = nl_shinji_ransac (3-point fit, nl_2p), 4 = nl_shinji_kneip_ransac (3-point fit, P3P, nl_2p).  The sample of iteration i is
 * the host sampler's (4 draws per iteration from (rng_state, rng_inc)).  votes_out / valid_out: iters x slots, q7_out: x 7.
 * cos_thr = cos(atan(thre_2d / f)), cos_nl = cos(nl_thre). */
int rpe_ransac_p3p_batch(rpe_context* ctx, int solver, uint64_t rng_state, uint64_t rng_inc, int iters, double thre_3d, double cos_thr,
                         double cos_nl, int* votes_out, double* q7_out, unsigned char* valid_out);
/* K4b: write the winner's inlier masks into the context's device masks (all modalities of `kind`; others
 * untouched) -- what setInlier() stores; returns the vote total. */
int rpe_inlier_mask(rpe_context* ctx, int kind, int mode, const double* pose7, double thre_3d, double cos_thr, double cos_nl,
                    int* votes_out);
/* K4r -- resident scoring session: between _begin and _end, rpe_score calls of at most 128 hypotheses and rpe_inlier_mask calls with
 * exactly these parameters are served by ONE resident launch (the hypotheses travel through the context's control block, the vote
 * counts return as run records): a whole RANSAC run of the reference (pose/AbsoluteOrientation.hpp:169-209: sample, score, keep the
 * best, shrink Iter, finally the winner's mask) costs one kernel launch instead of one per batch and one for the masks.  Results are
 * those of the calls outside a session, bit for bit.  RPE_ERR_STATE when the context cannot hold one (resident kernels unavailable,
 * a sharded context, more correspondences than one group per thread of a co-resident grid): the caller goes on without.  Any other
 * call on the context -- and a longer hypothesis list, or other parameters -- ends the session implicitly; _end is idempotent.  One
 * session per host thread; while it is open, resident loops of other contexts on the same GPU wait for it. */
int rpe_score_session_begin(rpe_context* ctx, int kind, int mode, double thre_3d, double cos_thr, double cos_nl);
int rpe_score_session_end(rpe_context* ctx);

/* ---- PROSAC order: the first top_k (<= 4096) entries of "indices sorted by weight, descending" (pose/Utility.hpp:107-118 sortIndexes as
 * PROSAC consumes it through getSortedIdx, pose/AOOnlyPoseAdapter.hpp:233-254) for n float weights (host pointer), computed on the GPU:
 * two-level radix select of the cut + LDS bitonic sort of the candidates.  Equal weights are ordered by index (the reference's
 * comparator leaves their order to std::sort), which makes the order unique and the prefix well defined.  RPE_ERR_STATE when more than
 * 8192 (near-)equal weights surround the cut: the caller then sorts on the host (the C++ adapters do). */
int rpe_prosac_order(rpe_context* ctx, const float* weights, int n, int top_k, int* order_out);

/* ---- K5 one round of nl_shinji_kneip_ls (AbsoluteOrientationNormal.hpp:484-505) + find_opt_cc (:24-39),
 * fused into one pass over up to 60 B/corr.  in: c_opt[3], Cw[3], Cc[3], Rwc9 (row-major, rotation used by
 * find_opt_cc).  out44: M23 (9) TW K | M33 (9) sigma | MNN (9) TL M | AA (6: xx xy xz yy yz zz) bb (3) | pad.
 * Sums are the FRESH contributions of this round; the host applies the reference's accumulate-across-rounds
 * recurrences.  Masks/weights of all three modalities are honoured (weights optional, scaled as the adapters do). */
int rpe_nl_round(rpe_context* ctx, const double* c_opt3, const double* Cw3, const double* Cc3, const double* Rwc9, double* out44);

/* ---- adapter-level pipelines, for hosts that cannot include the C++ headers (and for the parity tests): builds the
 * adapter the reference's demos would build for the arrays given (AOOnly / PnP / AO / NormalAO), runs one solver of
 * pose/ *.hpp on it, optionally a least-squares stage, and returns pose, votes, adapted Iter and the inlier masks.
 * method: 0 shinji_ransac  1 shinji_ransac2  2 shinji_prosac  3 kneip_ransac  4 kneip_prosac  5 shinji_kneip_ransac
 *         6 shinji_kneip_prosac  7 nl_kneip_ransac  8 nl_shinji_ransac  9 nl_shinji_kneip_ransac
 *         10 none (pose R9/t3 and mask_in are INPUTS: least-squares stage only)
 * ls:     0 none  1 shinji_ls / shinji_ls1 (inliers)  2 nl_shinji_kneip_ls (bug-compatible)  3 nl_shinji_kneip_ls (fixed)
 *         4 shinji_ls2 (all)  5 gn_refine_p2p  6 gn_refine_joint  7 gn_refine_p2plane  8 gn_refine_bearing  9 gn_refine_reproj
 * mask_in / mask_out: 3 x n shorts, row 0 = 2D-3D, 1 = 3D-3D, 2 = normal-normal.  seed: sampler stream.
 * RE-ENTRANT: every call builds its own random stream from `seed` and carries `score_mode` as a per-call option (rpe::RunOptions,
 * rpe/device.hpp); nothing process-wide is written, so concurrent calls from several threads -- different seeds, different modes --
 * each produce exactly what the same call produces alone (tests/cpp/reentrancy_host.cpp under ThreadSanitizer,
 * tests/test_gpu_pipelines.py::test_concurrent_runs_with_different_seeds_and_modes).  The reference's samplers share the process-global
 * rand() (pose/Utility.hpp:148,212,229; Library.cpp ao_ransac) and are not thread-safe; ao_ransac() here owns its stream too.  The
 * C++ free functions (shinji_ransac2<Tp>(adapter, thr, Iter, conf) ...) keep the reference's semantics when called as the reference
 * calls them -- one process-global stream, rpe::seed() in place of srand() -- and take the same options as a last, defaulted argument. */
typedef struct {
  int n;
  int dtype;              /* RPE_F32 / RPE_F64 */
  const void* bv;         /* host pointers, 3 x n, or NULL */
  const void* xc;
  const void* nc;
  const void* xw;
  const void* nw;
  const void* weights;    /* n x wcols column-major or NULL */
  int wcols;
  double fx, fy;
} rpe_problem;
int rpe_run(int method, const rpe_problem* p, double thre_3d, double thre_2d, double thre_nl, int* iter_io, double confidence,
            uint64_t seed, int ls, int score_mode, const short* mask_in, double* R9, double* t3, int* max_votes, short* mask_out);

/* The hypothesis stream of a solver made explicit ("both CPU restatement and GPU consume the same sample
 * list").  rpe_host_hypotheses runs `method`'s sampler (seeded as rpe_run does) and minimal solvers -- shinji K = 3
 * (AbsoluteOrientation.hpp:47-99), kneip (P3P.hpp:63-294), nl_2p (AbsoluteOrientationNormal.hpp:77-142), in the order the solver's
 * loop tries them -- for `iters` iterations WITHOUT scoring anything: no GPU is needed.  q7_out[cap x 7] = qw qx qy qz tx ty tz
 * (Tp values), first_out[iters + 1]: the hypotheses of iteration i are first_out[i] .. first_out[i+1].  Returns their number.
 * rpe_run_replay is rpe_run with the hypotheses of iteration i TAKEN from poses7[first[i] .. first[i+1]) instead of being sampled:
 * scoring on the GPU, best-so-far on strict '>', adaptive Iter, winner's masks and the optional least-squares stage as in rpe_run. */
int rpe_host_hypotheses(int method, const rpe_problem* p, int iters, uint64_t seed, double* q7_out, int cap, int* first_out);
int rpe_run_replay(int method, const rpe_problem* p, const double* poses7, const int* first, int list_iters, double thre_3d,
    double thre_2d,
                   double thre_nl, int* iter_io, double confidence, int ls, int score_mode, double* R9, double* t3, int* max_votes,
                   short* mask_out);

/* ------------------------------------------------------------------------------------------------
 * Part 3 -- front end (additive): the step BEFORE the hot path.  A depth frame becomes
 * the adapters' arrays directly in HBM: points_c / normal_c / bearingVectors of the frame, points_g / normal_g of the
 * model it is registered against.  Camera model = the reference simulator's pinhole (u - cx = fx * X / Z,
 * pose/Simulator.hpp:150-162; defaults f = 585, 640 x 480, principal point at the centre).  All fp32.
 * The reference has no counterpart of this stage; its contract is the numpy statement the parity tests hold.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { double fx, fy, cx, cy; int width, height; } rpe_camera;
enum { RPE_DEPTH_U16 = 0, RPE_DEPTH_F32 = 1 };
/* F1: upload one depth image (host, row-major, width*height values; metres = value * depth_scale) and build the frame's
 * maps: vertex map (NaN where depth is 0 / NaN / outside (dmin, dmax)), unit bearing vectors (every pixel), normal map
 * (central differences, towards the camera; NaN on the border, next to invalid depth, or where a neighbour's depth
 * differs by more than max_jump metres). */
int rpe_frame_set_depth(rpe_context* ctx, const void* depth, int depth_type, const rpe_camera* cam, double depth_scale, double dmin,
                        double dmax, double max_jump);
enum { RPE_MAP_VERTEX = 0, RPE_MAP_NORMAL = 1, RPE_MAP_BEARING = 2, RPE_MAP_MODEL_VERTEX = 3, RPE_MAP_MODEL_NORMAL = 4 };
/* copy one map (3 x width*height floats) to the host */
int rpe_frame_download(rpe_context* ctx, int which, float* out);
/* F2: the model := this frame's maps moved to the world frame under pose12 (Xw = R^T (Xc - t)); the model view's pose
 * and camera := pose12 and the frame's camera.  Device to device. */
int rpe_model_from_frame(rpe_context* ctx, const double* pose12);
/* ... or a model rendered elsewhere: world-frame vertex / normal maps (host, 3 x width*height floats, NaN = empty)
 * as seen from the view `pose12` (world -> model camera) with intrinsics `cam`. */
int rpe_model_upload(rpe_context* ctx, const float* vertex_w, const float* normal_w, const rpe_camera* cam, const double* pose12);
/* F3: projective data association of the frame against the model under the pose guess pose12 (frame: Xc = R Xw + t).
 * Each frame vertex is moved to the world, projected into the model view (nearest pixel), and paired with the model
 * vertex there if it lies within dist_thr metres and (use_normals) the normals agree to cos_thr.  Declares the problem
 * (n = width*height, RPE_F32) and fills XW, XC, BV, NW, NC in place, index = frame pixel; pixels without a partner get a
 * NaN column in XC / BV / NC (the reference's isValid convention, AOPoseAdapter.hpp:147-152) which every kernel skips.
 * matched (may be NULL: no host synchronisation) = number of pairs. */
int rpe_associate(rpe_context* ctx, const double* pose12, double dist_thr, double cos_thr, int use_normals, int64_t* matched);
/* ICP: max_iter rounds of { rpe_associate under the current pose ; one Gauss-Newton step of residual `kind`
 * (RPE_RES_P2PLANE uses the FRAME's normals, RPE_RES_P2P none) }.
 * device_resident = 1 keeps pose, solve and exp-map on the GPU (one host wait at the end; with fused = 1 ONE launch whose resident grid
 * iterates by itself), 0 solves on the host each round
 * (with fused = 1 that is ONE resident launch for the whole loop: the frame's pixels stay in registers and the host hands a pose to
 * the waiting grid every round, as rpe_gn_refine does -- the fastest form, needs a large-BAR device).
 * fused = 1 pairs and accumulates in ONE kernel per round (the pairs never exist in HBM: 48 B/pixel instead of 156); same
 * pairing function and per-pixel arithmetic as the two-kernel path, the sums differ in rounding only.  On return XW XC BV NW NC hold the pairs under the
 * RETURNED pose when fused, under the pose of the last round otherwise. */
typedef struct { int kind; int max_iter; double tol; double dist_thr; double cos_thr; int use_normals; int device_resident; int fused;
    } rpe_icp_options;
int rpe_icp(rpe_context* ctx, const rpe_icp_options* opt, double* pose12, int* iters_out, double* last_step, double* final_cost,
            int64_t* matched);

/* ---- host-side pieces of the solvers (no GPU needed): sampling, minimal solvers, small algebra.  They exist so that
 * hosts in other languages do not have to re-implement them, and so that the host logic can be tested on a CPU box.
 * 3 x K inputs are column-major doubles whose values are rounded to dtype before use. */
void rpe_host_random_elements(int n, int m, uint64_t seed, int draws, int* out);                 /* Utility.hpp:124-156 */
void rpe_host_prosac_samples(int dtype, int m, int n, uint64_t seed, int draws, int* out);       /* Utility.hpp:161-250 */
int rpe_host_update_num_iters(int dtype, double p, double ep, int model_points, int max_iters);  /* P3P.hpp:296-318    */
void rpe_host_sort_indexes(const double* w, int n, int* out);                                     /* Utility.hpp:107-118 */
int rpe_host_kneip_main(int dtype, const double* xw4, const double* bv4, double* sols12);         /* P3P.hpp:63-232     */
int rpe_host_kneip(int dtype, const double* xw4, const double* bv4, double* R9, double* t3);      /* P3P.hpp:250-294    */
void rpe_host_nl_2p(int dtype, const double* v18, double* R9,
    double* t3);                        /* AbsoluteOrientationNormal.hpp:77-142 */
void rpe_host_shinji(int dtype, const double* xw, const double* xc, int K, double* R9, double* t3); /* AbsoluteOrientation.hpp:47-99 */
void rpe_host_se3_exp(const double* a6, double* R9, double* t3);                                  /* sophus/se3.hpp:321-342 */
void rpe_host_svd3(const double* A9, double* U9, double* s3, double* V9);
void rpe_host_calc_err(const double* Rgt9, const double* tgt3, const double* Rse9, const double* tse3, double* err2, double* pct2);

#ifdef __cplusplus
}
#endif
#endif /* RGBD_POSE_HIP_H */
