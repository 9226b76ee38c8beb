// Resident scoring sessions (K4r, rpe_score.hip score_resident_kernel): ONE launch serves the batches of a RANSAC run -- the vote
// loops of AbsoluteOrientation.hpp:133-143,190-200,248-258 ... -- and the winner's masks.
#include "rpe_host.hpp"
using namespace rpeh;

namespace rpeh {
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
static void session_close_locked(rpe_context* c, bool registry_held);
// The session THIS thread opened on `device` -- if it still is that thread's: a context may have been handed to another thread, which
// takes the session over with its first batch (session_batch) -- is closed when the thread turns to another context: one session per
// thread, and a resident loop of the other context gets the slot.  Under the registry's lock the context cannot be destroyed
// (rpe_destroy's session_close passes through that lock); its session lock is only TRIED: a batch in flight on another thread means
// the session is no longer this thread's to close.
static void close_session_of_this_thread(int device, rpe_context* except) {
  if (t_session_id == 0 || t_session_dev != device) return;
  OpenSession& o = open_session(device);
  std::lock_guard<std::mutex> lk(o.m);
  rpe_context* ctx = o.id == t_session_id ? o.ctx : nullptr;
  if (ctx == except && ctx) return;   // (the caller closes its own)
  t_session_id = 0;
  if (!ctx || !ctx->sess_m.try_lock()) return;
  if (ctx->sess.active && o.id == ctx->sess.id) session_close_locked(ctx, true);
  ctx->sess_m.unlock();
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
void session_close(rpe_context* c) {
  if (!c) return;
  std::lock_guard<std::recursive_mutex> sk(c->sess_m);
  session_close_locked(c, false);
}
static void session_close_locked(rpe_context* c, bool registry_held) {
  if (!c->sess.active) return;
  if (registry_held) { OpenSession& o = open_session(c->device); if (o.id == c->sess.id) { o.ctx = nullptr; o.id = 0; } }
  else session_unregistered(c, c->sess.id);
  c->sess.active = false;
  session_message(c, 2, nullptr, 0, 0, (c->sess.base + (unsigned long long)c->sess.batches + 1) | rpe::kResidentStopBit);
  c->seq = c->sess.base + (unsigned long long)c->sess.batches + 2;   // stays ahead of every tag / sequence value the launch could use
  resident_mutex(c->device).release(c->sess.slot_token);   // (no-op if the slot was taken over meanwhile: the token is no longer the owner's)
  c->sess.slot_token = 0;
}
// The masks of a session's last message ("write them and leave", session_final_masks) were not waited for.  Look at their record now
// -- it has long arrived -- and, should the grid have gone away before it consumed the message (a host stalled beyond the grid's
// bounded wait), write the masks with the one-launch kernel: either way they are in place, in stream order, for whoever reads them.
static void session_verify(rpe_context* c) {
  if (!c) return;
  std::lock_guard<std::recursive_mutex> sk(c->sess_m);
  if (!c->sess.pending) return;
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
void session_end(rpe_context* c) {
  if (c) close_session_of_this_thread(c->device, c);
  session_close(c);
  session_verify(c);
}
// one batch through the open session: op 0 = score `count` hypotheses (staged: the kernel's layout, values of the array dtype), op 1 =
// the masks of one; totals = the 32 sums of the batch.  On a failure the session is closed and the caller takes the launch path.
int session_batch(rpe_context* c, int op, const void* staged, int count, size_t bytes, double* totals) {
  std::lock_guard<std::recursive_mutex> sk(c->sess_m);
  if (!c->sess.active) return RPE_ERR_STATE;   // (closed meanwhile by the thread that had opened it: the caller takes the launch path)
  // the slot may have changed hands (no message for longer than the grid's bounded wait: the grid has left): an ordinary launch then
  if (!resident_mutex(c->device).touch(c->sess.slot_token)) { session_close_locked(c, false); return RPE_ERR_STATE; }
  // a context handed to this thread with its session open: the session is this thread's from now on (a new number: the opener's no
  // longer matches, so the opener's calls on other contexts leave it alone)
  if (t_session_id != c->sess.id) { session_unregistered(c, c->sess.id); c->sess.id = ++g_session_ids; session_registered(c, c->sess.id); }
  const unsigned long long tag = c->sess.base + (unsigned long long)(c->sess.batches + 1);
  const double now = clock_us();
  const bool late = now - c->sess.last_us > 0.8 * c->sess.wait_us;
  c->sess.last_us = now;
  session_message(c, op, staged, count, bytes, tag);
  c->sess.batches++;
  c->seq = tag;
  const int rc = wait_host_partials(c, c->sess.runs, rpe::kSessionHypsMax, totals, 0, true);
  if (rc != RPE_OK) { if (rc == kResidentLost && !late) note_lost_grid(c); session_close_locked(c, false); return rc == kResidentLost ? RPE_ERR_HIP : rc; }
  return RPE_OK;
}
// The session's LAST message: the masks of a hypothesis whose vote total is already known (it was scored in this session), together
// with the stop.  Nothing is waited for: the grid writes the masks, sends their record and leaves on its own; the context's stream
// orders every later reader behind it, and session_verify looks at the record at the next call.
void session_final_masks(rpe_context* c, const void* staged, size_t bytes, const double* pose7, int votes) {
  std::lock_guard<std::recursive_mutex> sk(c->sess_m);
  const unsigned long long tag = c->sess.base + (unsigned long long)(c->sess.batches + 1);
  c->sess.pend_late = clock_us() - c->sess.last_us > 0.8 * c->sess.wait_us;
  session_message(c, 1, staged, 1, bytes, tag | rpe::kResidentStopBit);
  c->sess.batches++;
  session_unregistered(c, c->sess.id);
  c->sess.active = false;
  c->sess.pending = true; c->sess.pend_tag = tag; c->sess.pend_votes = votes;
  std::memcpy(c->sess.pend_pose, pose7, sizeof c->sess.pend_pose);
  c->seq = tag + 1;
  resident_mutex(c->device).release(c->sess.slot_token);   // (the grid waits for nobody any more: another resident grid may start beside it)
  c->sess.slot_token = 0;
}
bool session_seen(const rpe_context* c, const double* pose7, int* votes) {
  const size_t count = c->sess.seen_votes.size();
  for (size_t i = count; i-- > 0;)   // (the winner is usually among the latest)
    if (std::memcmp(&c->sess.seen_pose[7 * i], pose7, 7 * sizeof(double)) == 0) { *votes = c->sess.seen_votes[i]; return true; }
  return false;
}
bool session_matches(const rpe_context* c, int kind, int mode, double thre_3d, double cos_thr, double cos_nl) {
  return c->sess.active && c->sess.kind == kind && c->sess.mode == mode && c->sess.thre_3d == thre_3d && c->sess.cos_thr == cos_thr &&
         c->sess.cos_nl == cos_nl;
}
}  // namespace rpeh

extern "C" {
// ---------------------------------------------------------------------------------------------- resident scoring session
// Resident scoring session: the batches of ONE RANSAC run (rpe_score with at most 32 hypotheses and exactly these parameters) and the
// winner's masks (rpe_inlier_mask) are served by one resident launch instead of a launch each.  RPE_ERR_STATE if the context cannot
// run one (no large-BAR control block, a sharded context, a problem beyond one group per thread of the co-resident grid): the caller
// simply goes on -- rpe_score / rpe_inlier_mask then launch as always.  Any other call on the context closes the session.
int rpe_score_session_begin(rpe_context* c, int kind, int mode, double thre_3d, double cos_thr, double cos_nl) {
  if (!c) return fail(RPE_ERR_ARG, "null context");
  session_end(c);   // (also closes the session this thread may hold open on another context of this GPU: one session per thread)
  if (t_session_id && t_session_dev != c->device) close_session_of_this_thread(t_session_dev, nullptr);
  std::lock_guard<std::recursive_mutex> sk(c->sess_m);
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
  const unsigned long long token = resident_mutex(c->device).acquire(true, (c->test_pose_wait_s > 0 ? c->test_pose_wait_s : 2.0) * 1e6);
  if (!token) return fail(RPE_ERR_STATE, "the device's resident slot is held by another session: no resident scoring session now");
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
  if (e != hipSuccess) { resident_mutex(c->device).release(token); return fail(RPE_ERR_HIP, "resident scoring launch: %s", hipGetErrorString(e)); }
  c->sess.slot_token = token;
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

}  // extern "C"
