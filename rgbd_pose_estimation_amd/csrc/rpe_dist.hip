// Sharded contexts (one process per GPU, contiguous index ranges of every per-correspondence array): the RCCL
// communicator the library owns (rpe_comm_*; librccl resolved with dlopen), the in-kernel peer-to-peer exchange over xGMI
// (rpe_p2p_*), the host-side exchange through POSIX shared memory (rpe_hostex_*; csrc/rpe_hostex.cpp), and the sharded
// Gauss-Newton step that all-reduces the 32-double record once per iteration (rpe_gn_step_dist / rpe_gn_steps_dist).
#include "rpe_host.hpp"
using namespace rpeh;

namespace rpeh {
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
}  // namespace rpeh

extern "C" {
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
    // RCCL: ONE launch whose collecting stage leaves its run records in device memory (<= 8 per launch, as the single-GPU launch sends
    // to the host: one hop on the device, no arrival counters), ONE in-place all-reduce of the kRunSlots x kRunLd doubles -- the
    // ranks' run records added slot by slot --, and a one-workgroup kernel that sends them to the host as tagged pairs; every rank's
    // host adds the all-reduced run records in run order (identical sums on every rank) and expands the record.
    if (kind == RPE_RES_NORMAL) return fail(RPE_ERR_ARG, "RPE_RES_NORMAL is not served by the sharded step");
    if ((rc = kind_arrays(c, kind))) return rc;
    if (!pose12) return fail(RPE_ERR_ARG, "null argument");
    if ((rc = check_flags(c, kind, flags))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipEvent_t e0, e1;
    timing_pair(c, &e0, &e1);
    rpe::ReduceTarget rt = device_runs_target(c);
    rt.clean = take_clean(c, kind, false);   // nobody on the host sees this rank's own record
    HIP_TRY(rpe::launch_normal_eq(c->arrays(), kind, flags, pose12, rt, c->stream, e0, e1));
    double* runs = c->d_out + 64;
    constexpr int kRunDoubles = rpe::kRunSlots * rpe::kRunLd;
    NCCL_TRY(rccl().AllReduce(runs, runs, kRunDoubles, ncclFloat64, ncclSum, c->comm, c->stream));
    const unsigned long long seq = ++c->seq;
    HIP_TRY(rpe::launch_publish_pairs(runs, kRunDoubles, c->h_big, seq, c->stream));
    double tot[rpe::kRunLd];
    if ((rc = wait_host_partials(c, rpe::kRunSlots, rpe::kRunLd, tot))) return rc;
    if (kind == RPE_RES_P2P) expand_p2p17(tot, c->h_out);
    else { for (int i = 0; i < 32; i++) c->h_out[i] = i < 29 ? tot[i] : 0.0; }
  }
  if (c->p2p_world >= 1 && (rc = wait_host(c, rpe::kNeLd))) return rc;
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

// `steps` sharded steps CHAINED ON THE DEVICE: every launch takes its pose from the launch before it -- each workgroup adds that step's
// all-reduced run records, solves and applies the exp-map itself (chained_pose, rpe_reduce.hpp) -- so the host is not in the loop: it
// enqueues steps x {kernel, ncclAllReduce} and one finishing kernel, and waits once.  A launch's own latency (6.7 us on this stack for
// an empty kernel, scripts/ubench/stream_signal.hip) then overlaps the kernels in front of it instead of adding to every step, and no
// result has to reach the host between steps: what is left per step is the kernel, the collective and two kernel boundaries.  The SE(3)
// update runs on the device here (the device-resident loops' solve: gn_solve_update, checked against the host's to 1e-13); the
// host-side form is rpe_gn_steps_dist.  Identical poses on every rank (identical all-reduced records, identical arithmetic).
int rpe_gn_steps_dist_device(rpe_context* c, int kind, int flags, double* pose12, int steps, double* last_step_norm) {
  session_end(c);
  if (!c || !pose12 || steps < 1) return fail(RPE_ERR_ARG, "rpe_gn_steps_dist_device: bad argument");
  if (!c->comm) return fail(RPE_ERR_STATE, "rpe_gn_steps_dist_device: rpe_comm_init was not called");
  if (kind == RPE_RES_NORMAL) return fail(RPE_ERR_ARG, "RPE_RES_NORMAL is not served by the sharded step");
  int rc = kind_arrays(c, kind);
  if (rc) return rc;
  if ((rc = check_flags(c, kind, flags))) return rc;
  HIP_TRY(hipSetDevice(c->device));
  rpe::GnState st;
  st.tol = 0; st.step = 0; st.cost = 0; st.max_iters = steps; st.iters = 0; st.done = 0; st.status = 0;
  HIP_TRY(hipMemcpyAsync(c->d_gn_pose, pose12, 12 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_gn_state, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
  constexpr int kRunDoubles = rpe::kRunSlots * rpe::kRunLd;
  const bool clean = take_clean(c, kind, false);   // nobody on the host sees a record: CLEAN only over verified arrays
  for (int i = 0; i < steps; i++) {
    rpe::ReduceTarget rt = device_runs_target(c);
    rt.d_out = c->d_out + 64 + (size_t)(i & 1) * kRunDoubles;              // this step's run records ...
    rt.gn = c->d_gn_state;
    rt.gn_pose = c->d_gn_pose + 16 * (size_t)((i + 1) & 1);                // ... its pose: of the step before, updated by ...
    if (i == 0) rt.gn_pose = c->d_gn_pose;                                 // (the first launch: the uploaded pose as it stands)
    else { rt.chain_runs = c->d_out + 64 + (size_t)((i - 1) & 1) * kRunDoubles;   // ... that step's all-reduced run records,
           rt.chain_pose_out = c->d_gn_pose + 16 * (size_t)(i & 1); }             // and kept for the next launch
    rt.clean = clean;
    hipEvent_t e0, e1;
    timing_pair(c, &e0, &e1);
    HIP_TRY(rpe::launch_normal_eq(c->arrays(), kind, flags, pose12, rt, c->stream, e0, e1));
    NCCL_TRY(rccl().AllReduce(rt.d_out, rt.d_out, kRunDoubles, ncclFloat64, ncclSum, c->comm, c->stream));
  }
  const unsigned long long seq = ++c->seq;
  HIP_TRY(rpe::launch_chain_finish(kind, c->d_out + 64 + (size_t)((steps - 1) & 1) * kRunDoubles, c->d_gn_pose + 16 * (size_t)((steps - 1) & 1),
      c->d_gn_state, rpe::pivot_floor(c->dtype == RPE_F64), c->h_big, seq, c->stream));
  double out[16];
  if ((rc = wait_host_partials(c, 1, 16, out))) return rc;
  if (out[15] != 0.0) return fail(RPE_ERR_DEGENERATE, "normal equations are not positive definite at step %d", (int)out[14] - 1);
  for (int i = 0; i < 12; i++) pose12[i] = out[i];
  if (last_step_norm) *last_step_norm = out[12];
  return RPE_OK;
}

}  // extern "C"
