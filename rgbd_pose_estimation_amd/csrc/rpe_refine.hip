// The Gauss-Newton loops of include/rgbd_pose_hip.h Part 2 (north-star formulation): rpe_gn_refine (host-driven
// resident loop: ONE launch, the host solves the 6x6 system and applies the SE(3) exp-map -- sophus/se3.hpp:321-342 -- every
// iteration), rpe_gn_refine_joint (several residual kinds fused), rpe_gn_refine_device (autonomous: solve + update on the device),
// each with its one-launch-per-iteration form for what does not fit a resident grid; the per-device resident slot; tuning of the host
// thread that spins in the loop.
#include "rpe_host.hpp"
#include <memory>
using namespace rpeh;

namespace rpeh {
ResidentSlot& resident_mutex(int device) {
  static ResidentSlot m[64];
  return m[device >= 0 && device < 64 ? device : 0];
}

// Run shape of a resident grid (resident_host_loop and the autonomous launches use the same one, so their run records are the same):
// tiny problems send every workgroup's record (rows 1); grids of 32 workgroups and more are collected per XCD -- run r = workgroups r,
// r + 8, ... (rpe_residuals.hpp run_shape; RPE_RESIDENT_STRIDE=0 keeps runs of consecutive workgroups); RPE_RESIDENT_ROWS forces a
// run length of consecutive workgroups.  Returns the number of runs.
int resident_run_shape(int grid, int nacc, int max_rows, int rows_auto, rpe::ReduceTarget* rt) {
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
void note_lost_grid(rpe_context* c) {
  if (++c->resident_lost >= 2) c->resident = false;
}
}  // namespace rpeh

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
extern "C" {
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
      { SlotHold one_resident_grid(resident_mutex(c->device));
        if (!one_resident_grid) { rc = kResidentBusy; it = 0; break; }   // (a session holds the slot: one launch per iteration below)
        rc = resident_host_loop(c, launch, grid, nacc, max_rows, rows_auto, 1.0, pose12, max_iter, tol, &it, &step, &cost, &weight,
            "normal equations", clean, &verified); }
      if (clean && rc == kResidentDirty) note_clean_terms(c, sp.bits, false);
      else if (clean && verified) note_clean_terms(c, sp.bits, true);
      if (rc != kResidentDirty) break;   // else: NaN-marked arrays -- once more, guarded, from the untouched start pose
      std::memcpy(pose12, start, sizeof(start));
      it = 0;
    }
    if (rc != kResidentLost && rc != kResidentBusy) {
      if (iters_out) *iters_out = it;
      if (rc != RPE_OK) return rc;
      if (last_step) *last_step = step;
      if (final_cost) *final_cost = cost;
      return RPE_OK;
    }
    // the resident grid was lost after `it` whole iterations (or never launched): carry on below, one launch per iteration
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
  const bool want_auto = auto_on && c->resident && !sharded && !c->comm && !c->hostex && max_iter >= 2 &&
      (single ? rpe::normal_eq_resident_fits(c->arrays(), terms[0].kind, auto_blocks, !use_solver)
              : (use_solver && rpe::joint_resident_fits(c->arrays(), bits, flags, auto_blocks, true, joint_clean)));
  // the device's resident slot, until the result has arrived (a session in the slot: no resident grid, one launch per iteration below)
  std::unique_ptr<SlotHold> one_resident_grid(want_auto ? new SlotHold(resident_mutex(c->device)) : nullptr);
  if (want_auto && *one_resident_grid) {
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
  one_resident_grid.reset();
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
      { SlotHold one_resident_grid(resident_mutex(c->device));
        if (!one_resident_grid) { rc = kResidentBusy; it = 0; break; }   // (a session holds the slot: one launch per iteration below)
        rc = resident_host_loop(c, launch, grid, nacc, max_rows, rows_auto, sc, pose12, max_iter, tol, &it, &step, &cost, &weight,
            "normal equations", clean, &verified); }
      // promoted to "verified finite" only by a first record that was received and finite: a launch error, a wait that timed out or a
      // grid lost before the first record say nothing about the arrays (their state stays as it was)
      if (clean && rc == kResidentDirty) note_clean_launch(c, kind, false);
      else if (clean && verified) note_clean_launch(c, kind, true);
      if (rc != kResidentDirty) break;   // else: NaN-marked arrays -- once more, guarded, from the untouched start pose
      it = 0;
    }
    if (rc != kResidentLost && rc != kResidentBusy) {
      if (iters_out) *iters_out = it;
      if (rc != RPE_OK) return rc;
      if (last_step) *last_step = step;
      if (final_cost) *final_cost = cost;
      return RPE_OK;
    }
    // the resident grid was lost after `it` whole iterations (or never launched): carry on from pose12 below, one launch per iteration
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

}  // extern "C"
