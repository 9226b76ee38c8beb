// The thin extern "C" shim of librgbdpose_hip.so, Part 2 of include/rgbd_pose_hip.h: one entry point per kernel -- K1' moments, K1 / K2 /
// K3 normal equations (single kind and joint), K4 scoring and the batched RANSAC iterations, K4b inlier masks, K5, the PROSAC order --
// and the small host-side solves.  Argument checks, the launch through rpe_kernels.h, the wait for the record: nothing else lives
// here (context: rpe_context.hip; result hand-off: rpe_receive.hip; loops: rpe_refine.hip; sessions: rpe_session.hip; sharding:
// rpe_dist.hip; front end: rpe_frontend_api.hip).  There is NO CPU fallback.
#include "rpe_host.hpp"
using namespace rpeh;

namespace rpeh {

int normal_eq_launch(rpe_context* c, int kind, int flags, const double* pose12, double* d_out32, bool clean) {
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

int joint_spec(rpe_context* c, int nterms, const rpe_term* terms, int flags, const double* pose12, JointSpec* out) {
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

int joint_launch_checked(rpe_context* c, int nterms, const rpe_term* terms, int flags, const double* pose12, bool clean, int* bits_out) {
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

int vote_arrays(rpe_context* c, int kind) {
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
void stage_poses(int dtype, int exact, const double* poses7, int H, void* dst) {
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

void stage_thresholds(int dtype, int exact, double thre_3d, double cos_thr, double cos_nl, double thr[3]) {
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

int mask_by_launch(rpe_context* c, int kind, int mode, const double* pose7, double thre_3d, double cos_thr, double cos_nl, int* votes_out) {
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

}  // namespace rpeh

extern "C" {

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

// ---------------------------------------------------------------------------------------------- R1 lsq_pnp
int rpe_sine_error_sum(rpe_context* c, const double* pose7, double* sum_out, int64_t* count_out) {
  session_end(c);
  int rc = need_arrays(c, {RPE_XW, RPE_BV});
  if (rc) return rc;
  if (!pose7 || !sum_out) return fail(RPE_ERR_ARG, "rpe_sine_error_sum: null argument");
  HIP_TRY(hipSetDevice(c->device));
  hipEvent_t e0, e1;
  timing_pair(c, &e0, &e1);
  HIP_TRY(rpe::launch_sine_error(c->arrays(), pose7, collect_target(c), c->stream, e0, e1));
  if ((rc = wait_host(c, rpe::kNeLd))) return rc;
  *sum_out = c->h_out[0];
  if (count_out) *count_out = (int64_t)c->h_out[1];
  return RPE_OK;
}

// ---------------------------------------------------------------------------------------------- K1/K2/K3

int rpe_normal_eq_device(rpe_context* c, int kind, int flags, const double* pose12, double* d_out32) {
  session_end(c);
  if (!d_out32) return fail(RPE_ERR_ARG, "null d_out32");
  return normal_eq_launch(c, kind, flags, pose12, d_out32, c && take_clean(c, kind, false));   // nobody on the host sees this record
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

// ---------------------------------------------------------------------------------------------- K4



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

}  // extern "C"
