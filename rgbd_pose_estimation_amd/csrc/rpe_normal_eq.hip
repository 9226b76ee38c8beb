// K1 / K2 / K3: Gauss-Newton normal-equation kernels (one launch per call, and the RESIDENT form), their launchers,
// and the test hook of the device-side 6x6 solve.
#include "rpe_residuals.hpp"

namespace rpe {

// test hook (rpe_debug_device_gn_update): the device-resident loop's solve + SE(3) update on a record and pose of the caller's, so that
// the LDL^T solve and the exponential map the last workgroup runs can be checked against the oracle / golden values in isolation
__global__ void gn_update_probe_kernel(const double* __restrict__ rec, double* __restrict__ pose, double* __restrict__ step_ok,
                                       double rel_floor) {
  __shared__ double s_rec[32];
  __shared__ double s_pose[12];
  if (threadIdx.x < 32) s_rec[threadIdx.x] = rec[threadIdx.x];
  if (threadIdx.x < 12) s_pose[threadIdx.x] = pose[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    double step = 0.0;
    const bool ok = gn_solve_update<0>(s_rec, s_pose, &step, rel_floor);
    step_ok[0] = step; step_ok[1] = ok ? 1.0 : 0.0;
    if (ok) for (int k = 0; k < 12; k++) pose[k] = s_pose[k];
  }
}
hipError_t launch_gn_update_probe(const double* d_rec, double* d_pose, double* d_step_ok, double pivot_floor, hipStream_t s) {
  hipLaunchKernelGGL(gn_update_probe_kernel, dim3(1), dim3(64), 0, s, d_rec, d_pose, d_step_ok, pivot_floor);
  return hipGetLastError();
}

// The streaming loop is bound by the bytes the memory system delivers, not by instruction issue: at 1 M correspondences the CLEAN
// flavour executes 24 % fewer vector instructions per wave than the guarded one and takes the same time, and loading two groups
// ahead instead of one changes nothing either (profiles/r04_streaming_ab.jsonl) -- only the bearing kind, with two
// Jacobian rows per correspondence, still feels its arithmetic.
template <class T, int KIND, int BLK, bool MASK, bool WEIGHT, bool CLEAN>
__global__ __launch_bounds__(BLK) void normal_eq_kernel(const T* __restrict__ xw, const T* __restrict__ b, const T* __restrict__ c,
                                                        const short* __restrict__ mask, const T* __restrict__ weight, int64_t n,
                                                        PoseK<double> pose, Finish fin) {
  constexpr int P = Pk<T>::P;
  constexpr int NACC = KIND == KIND_P2P ? 17 : 29;
  typedef typename Pk<T>::V V;
  if (fin.gn != nullptr) {  // device-resident Gauss-Newton: finished loops cost an empty launch; the pose lives in HBM
    if (fin.gn->done) return;
    if (fin.chain_runs != nullptr) {   // chained sharded step: this launch's pose from the step before (chained_pose, rpe_reduce.hpp)
      if (!chained_pose<KIND == KIND_P2P ? 1 : 0>(fin, pose)) return;
    } else {
#pragma unroll
      for (int k = 0; k < 9; k++) pose.R[k] = fin.gn_pose[k];
#pragma unroll
      for (int k = 0; k < 3; k++) pose.t[k] = fin.gn_pose[9 + k];
    }
  }
  RPE_STAMP(0);
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) acc[k] = 0.0;
  const int64_t full = n / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const V* __restrict__ xw4 = reinterpret_cast<const V*>(xw);
  const V* __restrict__ b4 = reinterpret_cast<const V*>(b);
  const V* __restrict__ c4 = reinterpret_cast<const V*>(c);
  // All three kinds run on PAIRS of correspondences as 2-vectors (packed fp32 instructions for fp32 arrays).  The pair sums of kShare
  // consecutive groups share one widening into the fp64 accumulators (flush_pairs: three instructions per sum): four for the bearing
  // kind, whose loop feels its instruction count; two for point-to-plane; point-to-point widens every group, so that its record is the
  // same fp64 sum of per-group fp32 sums whatever the grid -- the sums of shards add up to the sum of the whole to fp64 rounding
  // (tests/test_gpu_fullsize.py::test_config5_shards_add_up_10M).
  typedef T V2 __attribute__((ext_vector_type(2)));
  constexpr int kShare = KIND == KIND_BEARING ? 4 : (KIND == KIND_P2PLANE ? 2 : 1);
  V2 carry[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) carry[k] = V2{T(0), T(0)};
  int carried = 0;
  // software pipeline: the loads of the NEXT group are in flight while the current one is reduced, so a CU's waves do not all
  // alternate between "everyone waits on memory" and "everyone computes"
  int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
  V a0, a1, a2, b0, b1, b2, c0, c1, c2;
  short m[P];
  T wv[P];
  if (g < full) {
    a0 = xw4[3 * g]; a1 = xw4[3 * g + 1]; a2 = xw4[3 * g + 2];
    b0 = b4[3 * g]; b1 = b4[3 * g + 1]; b2 = b4[3 * g + 2];
    if (KIND == KIND_P2PLANE) { c0 = c4[3 * g]; c1 = c4[3 * g + 1]; c2 = c4[3 * g + 2]; }
    if (MASK) load_mask_full(mask, g, m);
    if (WEIGHT) load_weight_full(weight, g, wv);
    pin16(a0); pin16(a1); pin16(a2); pin16(b0); pin16(b1); pin16(b2);
    if (KIND == KIND_P2PLANE) { pin16(c0); pin16(c1); pin16(c2); }
  }
#if defined(RPE_STAMPS) && RPE_STAMPS >= 2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // diagnostic build 2: when have the first loads landed?
  RPE_STAMP(11);
#endif
  while (g < full) {
    const int64_t gn = g + stride;
    const int64_t gl = gn < full ? gn : g;  // clamp: the last trip re-reads its own (cached) group instead of branching
    const V na0 = xw4[3 * gl], na1 = xw4[3 * gl + 1], na2 = xw4[3 * gl + 2];
    const V nb0 = b4[3 * gl], nb1 = b4[3 * gl + 1], nb2 = b4[3 * gl + 2];
    V nc0, nc1, nc2;
    if (KIND == KIND_P2PLANE) { nc0 = c4[3 * gl]; nc1 = c4[3 * gl + 1]; nc2 = c4[3 * gl + 2]; }
    short nm[P];
    T nwv[P];
    if (MASK) load_mask_full(mask, gl, nm);
    if (WEIGHT) load_weight_full(weight, gl, nwv);
    // The pipeline is pinned down at both ends: nothing below may move above this point and the loads above may not sink below it
    // (scheduling barrier), and the next group's vectors become opaque 16-byte values only AFTER the current group's arithmetic
    // (pin16 below) -- without the first the optimiser rotates the loop of the point-to-point flavour into "load, wait, compute" (no
    // load in flight during the arithmetic: 92 instead of 80 us at 20 M), without the second it narrows and re-splits the loads of the
    // CLEAN bearing flavour into 12- and 8-byte pieces at odd offsets (three times the launch time).
    __builtin_amdgcn_sched_barrier(0);
    T vw[3 * P], vb[3 * P], vc[3 * P];
    unpack3(a0, a1, a2, vw);
    unpack3(b0, b1, b2, vb);
    if (KIND == KIND_P2PLANE) unpack3(c0, c1, c2, vc);
    pair_group<T, KIND, MASK, WEIGHT, CLEAN, NACC>(pose, vw, vb, vc, m, wv, P, carry);
    if (++carried == kShare) { flush_pairs<T, NACC>(carry, acc); carried = 0; }
    a0 = na0; a1 = na1; a2 = na2; b0 = nb0; b1 = nb1; b2 = nb2;
    if (KIND == KIND_P2PLANE) { c0 = nc0; c1 = nc1; c2 = nc2; }
    __builtin_amdgcn_sched_barrier(0);
    pin16(a0); pin16(a1); pin16(a2); pin16(b0); pin16(b1); pin16(b2);
    if (KIND == KIND_P2PLANE) { pin16(c0); pin16(c1); pin16(c2); }
#pragma unroll
    for (int i = 0; i < P; i++) { if (MASK) m[i] = nm[i]; if (WEIGHT) wv[i] = nwv[i]; }
    g = gn;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && full * P < n) {  // leftover correspondences: bounds-checked loads (zeros past the end)
    T vw[3 * P], vb[3 * P], vc[3 * P];
    short lm[P];
    T lwv[P];
    load_group<T>(xw, full, n, vw);
    load_group<T>(b, full, n, vb);
    if (KIND == KIND_P2PLANE) load_group<T>(c, full, n, vc);
    if (MASK) load_scalars<T, short>(mask, full, n, lm, (short)0);
    if (WEIGHT) load_scalars<T, T>(weight, full, n, lwv, T(0));
    pair_group<T, KIND, MASK, WEIGHT, CLEAN, NACC>(pose, vw, vb, vc, lm, lwv, (int)(n - full * P), carry);
    carried = 1;
  }
  if (carried) flush_pairs<T, NACC>(carry, acc);
  RPE_STAMP(1);
  reduce_and_finish<NACC, kNeLd, KIND == KIND_P2P ? 1 : 0, BLK>(acc, fin);
}

// NREG: > 0 = the grid covers all groups with NREG groups per thread (frame-sized problems): each thread loads its groups ONCE, before
// the loop, and keeps them in registers for the whole refinement; 0 = the slice is re-read every iteration (it stays cache resident).
// A workgroup's slice is BLK x NREG consecutive groups, thread t holding groups t, t + BLK, ... of it.  The in-register instances run
// 256 threads x 2 groups: an iteration of a frame-sized problem is bound by VECTOR INSTRUCTION ISSUE, not by memory (about 230
// instructions of arithmetic per group and 180 for the wave's reduce-scatter of the sums, four cycles each), and the 512-thread x 1
// form put two waves on every SIMD of the 150 compute units it used -- 2 x (230 + 180) instructions per SIMD and iteration; one wave
// per SIMD with two groups issues 2 x 230 + 180.
// AUTO: 0 = host-driven; 1 = autonomous, every workgroup adds the run records and solves (resident_auto_stage: grids too small for
// 2, always in registers); 2 = autonomous with a solving workgroup beside the grid (auto_solver_kernel, solver_loop)
template <class T, int KIND, int BLK, bool MASK, bool WEIGHT, int NREG, int AUTO, bool CLEAN>
__global__ __launch_bounds__(BLK) void normal_eq_resident_kernel(const T* __restrict__ xw, const T* __restrict__ b, const T* __restrict__ c,
                             const short* __restrict__ mask, const T* __restrict__ weight, int64_t n,
                             const unsigned long long* __restrict__ ctl, unsigned long long first_tag, int max_iters, Finish fin) {
  constexpr int P = Pk<T>::P;
  constexpr int NACC = KIND == KIND_P2P ? 17 : 29;
  constexpr bool IN_REGS = NREG > 0;
  constexpr int GPT = IN_REGS ? NREG : 1;
  __shared__ double s_pose[12];
  __shared__ int s_go;
  const int64_t full = n / P, groups = (n + P - 1) / P;
  // autonomous loop with a solving workgroup (auto_solver_kernel below, launched beside this grid): the workers send sums and wait
  constexpr bool with_solver = AUTO == 2;
  const int workers = (int)gridDim.x;
  const int64_t stride = (int64_t)workers * BLK;
  const int64_t g0 = (int64_t)blockIdx.x * (BLK * GPT) + threadIdx.x;
  T rw[GPT][3 * P], rb[GPT][3 * P], rc[GPT][3 * P];
  short rm[GPT][P];
  T rwv[GPT][P];
  int rpresent[GPT];
#pragma unroll
  for (int r = 0; r < GPT; r++) {
    rpresent[r] = 0;
    const int64_t g = g0 + (int64_t)r * BLK;
    if (IN_REGS && g < groups) {
      load_any_group<T, KIND, MASK, WEIGHT>(xw, b, c, mask, weight, g, full, n, rw[r], rb[r], rc[r], rm[r], rwv[r]);
      rpresent[r] = g < full ? P : (int)(n - full * P);
    }
  }
  // autonomous form (fin.gn set): the first pose comes from HBM, every later one from this workgroup's own solve (resident_auto_stage)
#ifdef RPE_SOLVER_DEBUG
  if (with_solver && threadIdx.x == 0) __hip_atomic_store(solver_pose_area(fin, workers, NACC) + 32 + blockIdx.x, wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  constexpr bool autonomous = AUTO != 0;   // a template parameter: the host-driven instances carry no call to the solve (registers, scratch)
  double tol = 0.0;
  if (autonomous) {
    if (threadIdx.x < 12) s_pose[threadIdx.x] = fin.gn_pose[threadIdx.x];
    tol = fin.gn->tol;
    __syncthreads();
  }
  for (int it = 1; it <= max_iters; it++) {
    // stop requested or no host: uniform for the workgroup
    if (!autonomous && resident_wait_pose<BLK>(ctl, first_tag + (unsigned long long)it, s_pose, &s_go,
        fin.pose_wait_ticks) != 1) return;
    if (with_solver && it > 1 && solver_wait_pose<BLK>(solver_pose_area(fin, workers, NACC), first_tag + (unsigned long long)it, s_pose, &s_go, it == 2 ? kSolverMeetTicks : 200000000ull) != 0) return;
#ifdef RPE_STAMPS
    const bool stamp_it = it == 1000;
    if (stamp_it) RPE_STAMP(0);
#endif
    PoseK<double> pose;
#pragma unroll
    for (int k = 0; k < 9; k++) pose.R[k] = s_pose[k];
#pragma unroll
    for (int k = 0; k < 3; k++) pose.t[k] = s_pose[9 + k];
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = 0.0;
    if (IN_REGS) {
#pragma unroll
      for (int r = 0; r < GPT; r++)
        if (rpresent[r] > 0) normal_eq_group<T, KIND, MASK, WEIGHT, NACC, CLEAN>(pose, rw[r], rb[r], rc[r], rm[r], rwv[r], rpresent[r], acc);
    } else {
      for (int64_t g = g0; g < groups; g += stride) {
        T vw[3 * P], vb[3 * P], vc[3 * P];
        short mm[P];
        T ww[P];
        load_any_group<T, KIND, MASK, WEIGHT>(xw, b, c, mask, weight, g, full, n, vw, vb, vc, mm, ww);
        normal_eq_group<T, KIND, MASK, WEIGHT, NACC, CLEAN>(pose, vw, vb, vc, mm, ww, g < full ? P : (int)(n - full * P), acc);
      }
    }
#ifdef RPE_STAMPS
    if (stamp_it) RPE_STAMP(1);
#endif
#ifndef RPE_STAMPS
    const bool stamp_it = false;
#endif
    if (with_solver) { solver_send_sums<NACC, BLK>(acc, fin, first_tag + (unsigned long long)it); continue; }
    if (autonomous) {
      if (resident_auto_stage<NACC, BLK>(acc, fin, first_tag + (unsigned long long)it, it, max_iters, tol, s_pose,
          stamp_it) != 0) return;
      continue;
    }
    if (!resident_cross_stage<NACC, BLK>(acc, fin, first_tag + (unsigned long long)it, fin.seq + (unsigned long long)it,
        stamp_it)) return;
  }
}

#ifdef RPE_STAMPS
}  // namespace rpe
// diagnostic build only: copy the stamp buffer to the host (after a stream synchronise) and clear it
extern "C" int rpe_debug_read_stamps(unsigned long long* out, int nwords) {
  if (nwords > 4096 * 16) nwords = 4096 * 16;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rpe::g_stamps), (size_t)nwords * 8) != hipSuccess) return -1;
  static unsigned long long zeros[4096 * 16];
  return hipMemcpyToSymbol(HIP_SYMBOL(rpe::g_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : -1;
}
namespace rpe {
#endif

// the end of a chain of sharded steps: one workgroup adds the last step's all-reduced run records, solves, updates and sends the result
template <int MODE>
__global__ __launch_bounds__(64) void chain_finish_kernel(const double* __restrict__ runs, const double* __restrict__ pose_in, GnState* st,
                                                          double rel_floor, double* __restrict__ h_pairs, unsigned long long seq) {
  __shared__ double f_tot[32];
  __shared__ double f_pose[12];
  __shared__ double f_out[16];
  if (threadIdx.x < 12) f_pose[threadIdx.x] = pose_in[threadIdx.x];
  if (threadIdx.x < 32) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < kRunSlots; r++) t += runs[r * kRunLd + threadIdx.x];
    f_tot[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double step = st->step;
    bool ok = st->status == 0;
    if (!st->done) ok = gn_solve_update<MODE>(f_tot, f_pose, &step, rel_floor);
    for (int k = 0; k < 12; k++) f_out[k] = f_pose[k];
    f_out[12] = step; f_out[13] = st->done ? st->cost : record_entry<MODE>(f_tot, 27); f_out[14] = (double)(st->iters + (st->done ? 0 : 1));
    f_out[15] = ok ? 0.0 : 1.0;
  }
  __syncthreads();
  if (threadIdx.x < 16) store_tagged_pair(h_pairs, threadIdx.x, f_out[threadIdx.x], seq);
}
hipError_t launch_chain_finish(int kind, const double* d_runs, const double* d_pose, GnState* d_state, double pivot_floor, double* h_pairs,
                               unsigned long long seq, hipStream_t s) {
  if (kind == KIND_P2P) hipLaunchKernelGGL((chain_finish_kernel<1>), dim3(1), dim3(64), 0, s, d_runs, d_pose, d_state, pivot_floor, h_pairs, seq);
  else hipLaunchKernelGGL((chain_finish_kernel<0>), dim3(1), dim3(64), 0, s, d_runs, d_pose, d_state, pivot_floor, h_pairs, seq);
  return hipGetLastError();
}

template <class T, int KIND, int BLK>
static void normal_eq_launch(const DeviceArrays& A, int flags, const PoseK<double>& pose, const ReduceTarget& rt, hipStream_t s,
                             hipEvent_t ev0, hipEvent_t ev1) {
  const T* xw = (const T*)A.a[0];
  const T* b = (const T*)((KIND == KIND_BEARING || KIND == KIND_REPROJ) ? A.a[2] : A.a[1]);
  const T* c = (const T*)A.a[4];
  const int mod = (KIND == KIND_BEARING || KIND == KIND_REPROJ) ? 0 : 1;
  const short* mask = (flags & F_USE_MASK) ? A.mask[mod] : nullptr;
  const T* weight = (flags & F_USE_WEIGHT) ? (const T*)A.weight[mod] : nullptr;
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks, BLK);
  const Finish fin = make_finish(rt);
  // timed launches (bench.py's roofline leg) go through hipExtLaunchKernelGGL: the two events then carry the dispatch's own begin / end
  // timestamps -- what rocprofv3 reports for the kernel -- instead of bracketing it with two marker packets (which adds their latency)
#define RPE_NE_LAUNCH2(M, W, C) RPE_LAUNCH_EV((normal_eq_kernel<T, KIND, BLK, M, W, C>), dim3(G), dim3(BLK), 0, s, ev0, ev1, xw, b, c, mask, weight, A.n, pose, fin)
  // (the CLEAN flavour exists for fp32 arrays -- the dense-depth path; fp64 arrays always take the guarded one)
#define RPE_NE_LAUNCH(M, W) do { if constexpr (sizeof(T) == 4) { if (rt.clean) { RPE_NE_LAUNCH2(M, W, true); break; } } RPE_NE_LAUNCH2(M, W, false); } while (0)
  if (mask && weight) RPE_NE_LAUNCH(true, true);
  else if (mask) RPE_NE_LAUNCH(true, false);
  else if (weight) RPE_NE_LAUNCH(false, true);
  else RPE_NE_LAUNCH(false, false);
#undef RPE_NE_LAUNCH
#undef RPE_NE_LAUNCH2
}
template <class T>
static hipError_t normal_eq_t(const DeviceArrays& A, int kind, int flags, const double* pose12, const ReduceTarget& rt, hipStream_t s,
                              hipEvent_t ev0, hipEvent_t ev1) {
  const PoseK<double> pose = make_pose<double>(pose12);
  // fp64 arrays, the 29-sum kinds: 256-thread workgroups only (one wave per SIMD with the whole register file; their 512-thread
  // instances spilled 2-183 registers)
  const int blk = (sizeof(T) == 8 && kind != KIND_P2P) ? 256 : pick_block(rt);
  if (kind == KIND_P2P) {
    if (blk == 512) normal_eq_launch<T, KIND_P2P, 512>(A, flags, pose, rt, s, ev0, ev1);
    else normal_eq_launch<T, KIND_P2P, 256>(A, flags, pose, rt, s, ev0, ev1);
  } else if (kind == KIND_P2PLANE) {
    if (blk == 512) { if constexpr (sizeof(T) == 4) normal_eq_launch<T, KIND_P2PLANE, 512>(A, flags, pose, rt, s, ev0, ev1); }
    else normal_eq_launch<T, KIND_P2PLANE, 256>(A, flags, pose, rt, s, ev0, ev1);
  } else if (kind == KIND_BEARING) {
    if (blk == 512) { if constexpr (sizeof(T) == 4) normal_eq_launch<T, KIND_BEARING, 512>(A, flags, pose, rt, s, ev0, ev1); }
    else normal_eq_launch<T, KIND_BEARING, 256>(A, flags, pose, rt, s, ev0, ev1);
  } else if (kind == KIND_REPROJ) {
    if (blk == 512) { if constexpr (sizeof(T) == 4) normal_eq_launch<T, KIND_REPROJ, 512>(A, flags, pose, rt, s, ev0, ev1); }
    else normal_eq_launch<T, KIND_REPROJ, 256>(A, flags, pose, rt, s, ev0, ev1);
  } else return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t launch_normal_eq(const DeviceArrays& A, int kind, int flags, const double* pose12, const ReduceTarget& rt, hipStream_t s,
                            hipEvent_t ev0, hipEvent_t ev1) {
  return A.dtype ? normal_eq_t<double>(A, kind, flags, pose12, rt, s, ev0, ev1) : normal_eq_t<float>(A, kind, flags, pose12, rt, s,
      ev0, ev1);
}

template <class T, int KIND> constexpr bool resident_streams() { return !(sizeof(T) == 8 && (KIND == KIND_BEARING || KIND == KIND_REPROJ)); }
// resident form: ONE launch for up to max_iters iterations; ctl = the control block in fine-grained device memory, first_tag + i =
// tag of pose i (i = 1 ...), rt.seq + i = sequence value published with record i
template <class T, int KIND, int BLK>
static void resident_launch(const DeviceArrays& A, int flags, const unsigned long long* ctl, unsigned long long first_tag, int max_iters,
                            const ReduceTarget& rt, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
  const T* xw = (const T*)A.a[0];
  const T* b = (const T*)((KIND == KIND_BEARING || KIND == KIND_REPROJ) ? A.a[2] : A.a[1]);
  const T* c = (const T*)A.a[4];
  const int mod = (KIND == KIND_BEARING || KIND == KIND_REPROJ) ? 0 : 1;
  const short* mask = (flags & F_USE_MASK) ? A.mask[mod] : nullptr;
  const T* weight = (flags & F_USE_WEIGHT) ? (const T*)A.weight[mod] : nullptr;
  const int cap = std::max(1, resident_cap_device());
  // every workgroup resident at once: 8 waves per CU
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks < cap ? rt.max_blocks : cap, BLK);
  const int64_t groups = (A.n + Pk<T>::P - 1) / Pk<T>::P;
  const bool in_regs = (int64_t)G * BLK >= groups;
  Finish fin = make_finish(rt);
  constexpr int kMaxRows = 4 * (BLK / (KIND == KIND_P2P ? 17 : 29));   // up to 4 granules per collecting thread
  if (fin.rows > kMaxRows) fin.rows = kMaxRows;
  if (fin.rows < 1) fin.rows = 1;
  if (fin.gn == nullptr) fin.solver = 0;
  // in registers (R): 256 threads x 2 groups per workgroup -- the same BLK-group slice per workgroup, one wave per SIMD; streaming: BLK
  // threads, one group per thread and trip
#define RPE_RES_LAUNCH4(M, W, R, AU, C) do { if constexpr (R) RPE_LAUNCH_EV((normal_eq_resident_kernel<T, KIND, BLK / kResidentGroupsPerThread, M, W, kResidentGroupsPerThread, AU, C>), dim3(G), dim3(BLK / kResidentGroupsPerThread), 0, s, ev0, ev1, xw, b, c, mask, weight, A.n, ctl, first_tag, max_iters, fin); \
                                             else RPE_LAUNCH_EV((normal_eq_resident_kernel<T, KIND, BLK, M, W, 0, AU, C>), dim3(G), dim3(BLK), 0, s, ev0, ev1, xw, b, c, mask, weight, A.n, ctl, first_tag, max_iters, fin); } while (0)
#define RPE_RES_LAUNCH3(M, W, R, AU) do { if constexpr (sizeof(T) == 4) { if (rt.clean) { RPE_RES_LAUNCH4(M, W, R, AU, true); break; } } RPE_RES_LAUNCH4(M, W, R, AU, false); } while (0)
  // (autonomous without a solving workgroup: only grids of fewer than 8 workgroups -- always in registers)
#define RPE_RES_LAUNCH2(M, W, R) do { if (fin.gn != nullptr) { if (fin.solver) RPE_RES_LAUNCH3(M, W, R, 2); else if constexpr (R) RPE_RES_LAUNCH3(M, W, R, 1); } \
                                      else RPE_RES_LAUNCH3(M, W, R, 0); } while (0)
  // (fp64 arrays, the two-row 2D-3D kinds, more than one group per thread: no instance -- it spilled 7-45 registers; the callers ask
  // normal_eq_resident_fits first and run those refinements one launch per iteration)
#define RPE_RES_LAUNCH(M, W) do { if (in_regs) RPE_RES_LAUNCH2(M, W, true); else if constexpr (resident_streams<T, KIND>()) RPE_RES_LAUNCH2(M, W, false); } while (0)
  if (mask && weight) RPE_RES_LAUNCH(true, true);
  else if (mask) RPE_RES_LAUNCH(true, false);
  else if (weight) RPE_RES_LAUNCH(false, true);
  else RPE_RES_LAUNCH(false, false);
#undef RPE_RES_LAUNCH
#undef RPE_RES_LAUNCH2
#undef RPE_RES_LAUNCH3
#undef RPE_RES_LAUNCH4
}
template <class T>
static hipError_t resident_t(const DeviceArrays& A, int kind, int flags, const unsigned long long* ctl, unsigned long long first_tag,
                             int max_iters, const ReduceTarget& rt, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
  if (kind == KIND_P2P) resident_launch<T, KIND_P2P, 512>(A, flags, ctl, first_tag, max_iters, rt, s, ev0, ev1);
  else if (kind == KIND_P2PLANE) resident_launch<T, KIND_P2PLANE, 512>(A, flags, ctl, first_tag, max_iters, rt, s, ev0, ev1);
  else if (kind == KIND_BEARING) resident_launch<T, KIND_BEARING, 512>(A, flags, ctl, first_tag, max_iters, rt, s, ev0, ev1);
  else if (kind == KIND_REPROJ) resident_launch<T, KIND_REPROJ, 512>(A, flags, ctl, first_tag, max_iters, rt, s, ev0, ev1);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
// Co-residency of the resident kernels, per device: the occupancy the runtime reports for the heaviest instances (512-thread
// workgroups, up to 256 VGPRs: one workgroup per compute unit) times the compute units of THIS device -- a CPX / DPX partition or a
// smaller part has fewer than 256 -- and never more than one workgroup per compute unit, whatever a light instance would allow (the
// occupancy query is known to come out one block too high at some SGPR counts: MI355X_MICROARCH.md, correctness boundaries).  What
// cannot be known here is another process on the same GPU; that case is caught at run time (a run whose granules never arrive tells
// the host, which finishes the refinement with one launch per iteration: rpe_host.hpp resident_host_loop).
int resident_cap_device() {
  static int cap[64];
  static bool known[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 0; }
  if (known[dev]) return cap[dev];
  int cus = 0, c = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); cus = 0; }
  int per_cu = 1;
  constexpr int kRegBlk = 512 / kResidentGroupsPerThread;
  const void* heavy[] = {(const void*)normal_eq_resident_kernel<float, KIND_P2PLANE, 512, true, true, 0, 0, false>,
                         (const void*)normal_eq_resident_kernel<float, KIND_P2P, kRegBlk, true, false, kResidentGroupsPerThread, 1, false>,
                         (const void*)normal_eq_resident_kernel<double, KIND_BEARING, kRegBlk, true, true, kResidentGroupsPerThread, 0, false>};
  const int heavy_blk[] = {512, kRegBlk, kRegBlk};
  for (int i = 0; i < 3; i++) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, heavy[i], heavy_blk[i], 0) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
    if (nb < per_cu) per_cu = nb;
  }
  c = per_cu >= 1 ? cus : 0;
  if (c > 256) c = 256;
  // experiments / tests: a smaller device
  if (const char* e = getenv("RPE_RESIDENT_CAP")) { const int v = atoi(e); if (v >= 0 && v < c) c = v; }
  cap[dev] = c; known[dev] = true;
  return c;
}
// grid the resident kernel runs with, the number of sums per record, the longest run of workgroups one collecting workgroup can take,
// and the run length used unless the caller forces one: BLK / sums rows (one granule per collecting thread) times 1..4, aiming at <= 8
// runs
void resident_geometry(const DeviceArrays& A, int kind, int max_blocks, int* grid, int* nacc, int* max_rows, int* rows_auto) {
  const int P = A.dtype ? 2 : 4;
  const int blk = resident_block(), cap = std::max(1, resident_cap_device());
  *grid = reduce_grid(A.n, P, max_blocks < cap ? max_blocks : cap, blk);
  *nacc = kind == KIND_P2P ? 17 : 29;
  const int rgn = blk / *nacc;
  *max_rows = 4 * rgn;
  int mult = (*grid + rgn * 8 - 1) / (rgn * 8);
  if (mult > 4) mult = 4;
  if (mult < 1) mult = 1;
  *rows_auto = rgn * mult;
}
// ---- the solving workgroup of the autonomous loops, as a kernel of its own (rpe_residuals.hpp solver_loop)
template <int NACC>
__global__ __launch_bounds__(512) void auto_solver_kernel(int workers, unsigned long long first_tag, int max_iters, Finish fin) {
  solver_loop<NACC, 512>(fin, workers, first_tag, max_iters);
}
int auto_solver_workers(int grid) { return auto_solver_grid(grid, std::max(1, resident_cap_device())) ? grid : 0; }
// Workers beside a solving workgroup: ONE COMPUTE UNIT PER SHADER ENGINE stays free (7 of 8: 224 workers on a whole MI355X).  The
// dispatcher assigns a workgroup to a shader engine when it takes it off the queue, not when a compute unit is free: with more than 7
// heavy workers per engine (one workgroup fills a compute unit, and only the lightest point-to-point instance can share one with the
// solver) the engine that also holds the solving workgroup is one compute unit short, its last worker starts only when another
// workgroup of that engine leaves -- and they all wait for its sums.  Measured: 225-255 workers lose a loop within ten refinements
// of >= 1 M point-to-plane correspondences (the late worker's start stamp coincides with the others' bounded wait running out), 224
// and fewer never in 360; profiles/r05_solver_room.txt.  RPE_AUTO_SOLVER_ROOM overrides the number of compute units left free.
int auto_solver_cap() {
  const int cap = std::max(1, resident_cap_device());
  static const int env_room = getenv("RPE_AUTO_SOLVER_ROOM") ? atoi(getenv("RPE_AUTO_SOLVER_ROOM")) : 0;
  const int room = env_room >= 1 ? env_room : std::max(1, cap / 8);
  return std::max(1, cap - room);
}
hipError_t launch_auto_solver(int nacc, int workers, unsigned long long first_tag, int max_iters, const ReduceTarget& rt, hipStream_t s) {
  const Finish fin = make_finish(rt);
  if (nacc == 17) hipLaunchKernelGGL((auto_solver_kernel<17>), dim3(1), dim3(512), 0, s, workers, first_tag, max_iters, fin);
  else hipLaunchKernelGGL((auto_solver_kernel<29>), dim3(1), dim3(512), 0, s, workers, first_tag, max_iters, fin);
  return hipGetLastError();
}
// has the resident kernel an instance for this problem?  Everything but (a) fp64 arrays of the 2D-3D kinds beyond one group per thread,
// (b) autonomous loops WITHOUT a solving workgroup beyond one group per thread (that form exists for small grids only).
bool normal_eq_resident_fits(const DeviceArrays& A, int kind, int max_blocks, bool autonomous_no_solver) {
  const bool narrow = A.dtype && (kind == KIND_BEARING || kind == KIND_REPROJ);
  if (!narrow && !autonomous_no_solver) return true;
  const int P = A.dtype ? 2 : 4;
  const int blk = resident_block(), cap = std::max(1, resident_cap_device());
  const int G = reduce_grid(A.n, P, max_blocks < cap ? max_blocks : cap, blk);
  return (int64_t)G * blk >= (A.n + P - 1) / P;
}
hipError_t launch_normal_eq_resident(const DeviceArrays& A, int kind, int flags, const unsigned long long* ctl,
                                     unsigned long long first_tag, int max_iters, const ReduceTarget& rt, hipStream_t s, hipEvent_t ev0,
                                     hipEvent_t ev1) {
  return A.dtype ? resident_t<double>(A, kind, flags, ctl, first_tag, max_iters, rt, s, ev0, ev1)
                 : resident_t<float>(A, kind, flags, ctl, first_tag, max_iters, rt, s, ev0, ev1);
}

void preload_normal_eq() {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, (const void*)gn_update_probe_kernel) != hipSuccess) (void)hipGetLastError();
}

}  // namespace rpe
