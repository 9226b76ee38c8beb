// Part 3 of include/rgbd_pose_hip.h: the depth-frame front end (back-projection, normals, projective association: kernels in
// rpe_frontend.hip) and ICP over it (fused rounds / resident grids: rpe_icp.hip).  No reference counterpart.
#include "rpe_host.hpp"
#include <memory>
using namespace rpeh;

extern "C" {
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
    const bool want_auto = auto_on && c->resident && o->fused && o->max_iter >= 2 && !c->hostex && !c->comm && c->p2p_world < 1;
    // the device's resident slot until the result has arrived (end of this block's scope); a session in the slot: one launch per round
    std::unique_ptr<SlotHold> one_resident_grid(want_auto ? new SlotHold(resident_mutex(c->device)) : nullptr);
    bool one_launch = false;
    if (want_auto && *one_resident_grid) {
      // ONE launch: the resident grid pairs, sums, solves and updates by itself (icp_resident_kernel with resident_auto_stage)
      one_launch = true;
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
    { SlotHold one_resident_grid(resident_mutex(c->device));
      if (!one_resident_grid) rc = kResidentBusy;   // (a session holds the slot: every round a launch)
      else rc = resident_host_loop(c, launch, grid, nacc, max_rows, rows_auto, 1.0, pose12, o->max_iter, o->tol, &it, &step, &cost, &pairs,
          "ICP: normal equations"); }
    if (rc != RPE_OK && rc != kResidentLost && rc != kResidentBusy) { if (iters_out) *iters_out = it; return rc; }
    host_rounds = rc == kResidentLost || rc == kResidentBusy;   // the grid was lost after `it` whole rounds (or never launched): the rest one launch per round
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
