// pose/RansacEngine.hpp -- the one RANSAC/PROSAC driver behind every *_ransac / *_prosac of pose/*.hpp.
//
// The reference runs, per iteration: sample -> minimal solver(s) -> an O(N) vote loop over all correspondences ->
// "if (votes > best) { keep pose + mask; Iter = RANSACUpdateNumIters(...) }" (e.g. AbsoluteOrientation.hpp:169-209).
// The vote loops are the hot path.  Here hypotheses are generated on the host in batches, each batch is scored in
// ONE pass over the HBM-resident arrays (rpe_score, kernel K4), and the batch results are then REPLAYED IN ORDER with
// the reference's exact sequential semantics: strict '>' (ties keep the earlier hypothesis), Iter shrinks the moment
// a better hypothesis is seen, and iterations at or beyond the new Iter are discarded.  The outcome (pose, votes,
// Iter, mask) is therefore what the sequential loop would produce from the same hypothesis stream; only the sampler
// may have been advanced further than the reference would have advanced rand().
// The winner's inlier mask is written once at the end (rpe_inlier_mask, kernel K4b) instead of at every improvement, and stays on
// the device until somebody reads the adapter's host copy.
// On a frame-sized problem the whole run -- every short batch and the masks -- is served by ONE resident launch (a scoring session,
// rpe_score_session_begin / _end, kernel K4r): the batches travel through the context's control block, the votes come back as run
// records, and the replay below is unchanged.
#ifndef RPE_RANSAC_ENGINE_HEADER
#define RPE_RANSAC_ENGINE_HEADER

#include <algorithm>
#include <vector>
#include "../rpe/types.hpp"
#include "../rpe/device.hpp"

template <typename T> int RANSACUpdateNumIters(T p, T ep, const int modelPoints, const int maxIters);

namespace rpe {

template <class Tp> struct VoteSpec {
  int kind = RPE_VOTE_33;
  Tp thre_3d = Tp(0);
  Tp cos_thr = Tp(2);   // cos(atan(thre_2d / f))
  Tp cos_nl = Tp(2);    // cos(nl_thre)
  int modalities = 1;   // m in the outlier ratio (m N - votes) / (m N)
  int model_points = 3; // K handed to RANSACUpdateNumIters
};

// how many leading positions of the PROSAC order a run of at most `iters` iterations can read: the sampler's n starts at K and
// grows by at most one per draw, position n itself is read, and the engine never draws more than the current iteration bound
// (a batch is clipped to Iter - it, and Iter only shrinks)
inline int prosac_prefix(int iters, int K) { return iters + K + 2; }

template <class Tp> inline void pose7(const SE3<Tp>& s, double* q7) {
  const Quat<Tp>& q = s.so3().unit_quaternion();
  q7[0] = q.w; q7[1] = q.x; q7[2] = q.y; q7[3] = q.z;
  q7[4] = s.translation()[0]; q7[5] = s.translation()[1]; q7[6] = s.translation()[2];
}

// Adapter: any pose adapter (setMaxVotes/getMaxVotes/setRcw/sett/device()).
// produce(iters, hyps, first, votes): advance the sampler by `iters` reference iterations; hyps = their 0..3 hypotheses each, in the
//   reference's order; first[i] .. first[i+1] = the hypotheses of iteration i (first.size() == iters + 1); votes[h] = score of hyps[h].
// commit(cols, device_cols): the winner's masks are on the device; the adapter adopts them (setInlierFromDevice).
template <class Tp, class Adapter, class Produce, class Commit>
void ransac_engine_batched(Adapter& adapter, const VoteSpec<Tp>& spec, Produce produce, Commit commit, int& Iter, Tp confidence,
    int mask_cols, const RunOptions& opt) {
  const int N = adapter.getNumberCorrespondences();
  Settings& cfg = Settings::get();
  const bool prof = cfg.profile;
  double tp = prof ? now_us() : 0;
  auto lap = [&](double& acc) { if (prof) { const double t = now_us(); acc += t - tp; tp = t; } };
  rpe_context* ctx = adapter.device().ctx();
  adapter.setMaxVotes(-1);
  // ONE resident launch serves the short batches and the winner's masks of this run (rpe_score_session_begin; refused -- no large-BAR
  // control block, a sharded context, more than a frame's worth of correspondences -- the calls below launch one kernel each as before)
  struct Session {
    rpe_context* ctx; bool open;
    ~Session() { if (open) rpe_score_session_end(ctx); }
  } session = {ctx, cfg.score_session && Iter > 0 &&
      rpe_score_session_begin(ctx, spec.kind, opt.mode(), (double)spec.thre_3d, (double)spec.cos_thr, (double)spec.cos_nl) == RPE_OK};
  bool have_best = false;
  SE3<Tp> best;
  int it = 0;
  int batch = std::max(1, cfg.first_batch);
  std::vector<SE3<Tp> > hyps;
  std::vector<int> first;   // first[i] = index into hyps of iteration (it + i)'s first hypothesis
  std::vector<int> votes;
  while (it < Iter) {
    const int iters = std::min(batch, Iter - it);
    hyps.clear(); first.clear(); votes.clear();
    if (prof) tp = now_us();
    produce(iters, hyps, first, votes);
    lap(cfg.prof.score);
    if (prof) { cfg.prof.hypotheses += (int)hyps.size(); cfg.prof.batches++; }
    // sequential replay
    for (int i = 0; i < iters && it + i < Iter; i++) {
      for (int h = first[i]; h < first[i + 1]; h++) {
        if (votes[h] > adapter.getMaxVotes()) {
          adapter.setMaxVotes(votes[h]);
          adapter.setRcw(hyps[h].so3());
          adapter.sett(hyps[h].translation());
          best = hyps[h]; have_best = true;
          const int mN = spec.modalities * N;
          // (Tp)(m N - votes) / N / m, evaluated like the reference (e.g. AbsoluteOrientation.hpp:429)
          const Tp ep = spec.modalities == 1 ? (Tp)(N - votes[h]) / N : (Tp)(mN - votes[h]) / N / spec.modalities;
          Iter = RANSACUpdateNumIters(confidence, ep, spec.model_points, Iter);
        }
      }
    }
    lap(cfg.prof.replay);
    it += iters;
    batch = std::min(batch * 2, std::max(1, cfg.max_batch));
  }
  if (have_best) {
    if (prof) tp = now_us();
    double b7[7];
    pose7<Tp>(best, b7);
    int total = 0;
    check(rpe_inlier_mask(ctx, spec.kind, opt.mode(), b7, (double)spec.thre_3d, (double)spec.cos_thr, (double)spec.cos_nl, &total),
          "rpe_inlier_mask");
    if (session.open) { session.open = false; check(rpe_score_session_end(ctx), "rpe_score_session_end"); }
    const bool has23 = spec.kind == RPE_VOTE_23 || spec.kind == RPE_VOTE_23_MATRIX || spec.kind == RPE_VOTE_33_23 ||
                       spec.kind == RPE_VOTE_NN_23 || spec.kind == RPE_VOTE_NN_33_23;
    const bool has33 = spec.kind == RPE_VOTE_33 || spec.kind == RPE_VOTE_33_23 || spec.kind == RPE_VOTE_NN_33 || spec.kind == RPE_VOTE_NN_33_23;
    const bool hasnn = spec.kind == RPE_VOTE_NN_23 || spec.kind == RPE_VOTE_NN_33 || spec.kind == RPE_VOTE_NN_33_23;
    // the device masks ARE the result: the adapter adopts them without a download (its host copy is fetched on first access,
    // rpe::HostMask); columns of modalities this solver does not vote on become zero, as in the matrix the reference builds
    const unsigned device_cols = (has23 && mask_cols >= 1 ? 1u : 0u) | (has33 && mask_cols >= 2 ? 2u : 0u) | (hasnn
        && mask_cols >= 3 ? 4u : 0u);
    commit(mask_cols, device_cols);

    lap(cfg.prof.mask);
  }
}

// gen(out): advance the sampler by ONE reference iteration and append its 0..3 hypotheses, in the reference's order (host-side
// minimal solvers); the batch is scored by rpe_score (kernel K4).
template <class Tp, class Adapter, class Gen, class Commit>
void ransac_engine(Adapter& adapter, const VoteSpec<Tp>& spec, Gen gen, Commit commit, int& Iter, Tp confidence, int mask_cols,
    const RunOptions& opt) {
  Settings& cfg = Settings::get();
  if (cfg.capture) {   // generation only (rpe_host_hypotheses): the stream of `Iter` iterations, no device
    std::vector<SE3<Tp> > hyps;
    Settings::HypothesisList& out = *cfg.capture;
    if (out.first.empty()) out.first.push_back(0);
    for (int i = 0; i < Iter; i++) {
      hyps.clear();
      gen(hyps);
      for (const SE3<Tp>& h : hyps) { double q[7]; pose7<Tp>(h, q); out.q7.insert(out.q7.end(), q, q + 7); }
      out.first.push_back((int)(out.q7.size() / 7));
    }
    return;
  }
  rpe_context* ctx = adapter.device().ctx();
  std::vector<double> q7;
  int replay_pos = 0;
  auto produce = [&](int iters, std::vector<SE3<Tp> >& hyps, std::vector<int>& first, std::vector<int>& votes) {
    const double t0 = cfg.profile ? now_us() : 0;
    first.assign(1, 0);
    if (cfg.replay) {   // a GIVEN hypothesis list (rpe_run_replay); iterations past its end have no hypotheses
      const Settings::HypothesisList& in = *cfg.replay;
      const int have = (int)in.first.size() - 1;
      for (int i = 0; i < iters; i++, replay_pos++) {
        if (replay_pos < have)
          for (int h = in.first[replay_pos]; h < in.first[replay_pos + 1]; h++) {
            const double* q = &in.q7[7 * (size_t)h];
            hyps.push_back(SE3<Tp>(SO3<Tp>::fromQuaternionRaw((Tp)q[0], (Tp)q[1], (Tp)q[2], (Tp)q[3]),
                Point3<Tp>((Tp)q[4], (Tp)q[5], (Tp)q[6])));
          }
        first.push_back((int)hyps.size());
      }
    } else
    for (int i = 0; i < iters; i++) { gen(hyps); first.push_back((int)hyps.size()); }
    // the caller books the whole call as "score"
    if (cfg.profile) { const double t1 = now_us(); cfg.prof.generate += t1 - t0; cfg.prof.score -= t1 - t0; }
    if (hyps.empty()) return;
    q7.resize(hyps.size() * 7);
    for (size_t h = 0; h < hyps.size(); h++) pose7<Tp>(hyps[h], &q7[7 * h]);
    votes.resize(hyps.size());
    check(rpe_score(ctx, spec.kind, opt.mode(), q7.data(), (int)hyps.size(), (double)spec.thre_3d, (double)spec.cos_thr,
                    (double)spec.cos_nl, votes.data()), "rpe_score");
  };
  ransac_engine_batched<Tp>(adapter, spec, produce, commit, Iter, confidence, mask_cols, opt);
}

// The 3D-3D solvers (shinji_ransac / shinji_ransac2): sampling, the 3-point fit and the scoring all run on the device
// (rpe_ransac33_batch); the hypotheses are bitwise the ones the host generator above would produce from the same random stream,
// which is advanced here by the K draws per iteration the host sampler would have consumed.
// Short batches (the first ones: at most kHostBatch iterations) are generated by the host's `gen` instead -- eight 3-point fits take
// the
// CPU 4 us, while a device batch of eight is one thread per fit running a 3x3 Jacobi SVD (25 us of latency) -- and scored by
// rpe_score's
// single-launch form; the random stream is the same either way, so the two can alternate batch by batch.
template <class Tp, class Adapter, class Gen, class Commit>
void ransac_engine_device33(Adapter& adapter, const VoteSpec<Tp>& spec, Gen gen, Commit commit, int& Iter, Tp confidence,
    int mask_cols, const RunOptions& opt) {
  constexpr int kHostBatch = 32;
  Settings& cfg = Settings::get();
  rpe_context* ctx = adapter.device().ctx();
  std::vector<double> q7;
  std::vector<unsigned char> valid;
  std::vector<int> all_votes;
  auto produce = [&](int iters, std::vector<SE3<Tp> >& hyps, std::vector<int>& first, std::vector<int>& votes) {
    first.assign(1, 0);
    if (iters <= kHostBatch) {
      const double t0 = cfg.profile ? now_us() : 0;
      for (int i = 0; i < iters; i++) { gen(hyps); first.push_back((int)hyps.size()); }
      if (cfg.profile) { const double t1 = now_us(); cfg.prof.generate += t1 - t0; cfg.prof.score -= t1 - t0; }
      if (hyps.empty()) return;
      q7.resize(hyps.size() * 7);
      for (size_t h = 0; h < hyps.size(); h++) pose7<Tp>(hyps[h], &q7[7 * h]);
      votes.resize(hyps.size());
      check(rpe_score(ctx, spec.kind, opt.mode(), q7.data(), (int)hyps.size(), (double)spec.thre_3d, (double)spec.cos_thr,
                      (double)spec.cos_nl, votes.data()), "rpe_score");
      return;
    }
    Rand31& g = opt.stream();
    q7.resize((size_t)iters * 7); valid.resize((size_t)iters); all_votes.resize((size_t)iters);
    for (int done = 0; done < iters;) {   // the device call takes at most kMaxScoreH iterations at a time
      const int chunk = std::min(iters - done, 8192);
      check(rpe_ransac33_batch(ctx, g.state(), g.inc(), chunk, opt.mode(), (double)spec.thre_3d, all_votes.data() + done,
                               q7.data() + 7 * (size_t)done, valid.data() + done), "rpe_ransac33_batch");
      g.advance((uint64_t)spec.model_points * (uint64_t)chunk);
      done += chunk;
    }
    for (int i = 0; i < iters; i++) {
      if (valid[(size_t)i]) {
        const double* q = &q7[7 * (size_t)i];
        hyps.push_back(SE3<Tp>(SO3<Tp>::fromQuaternionRaw((Tp)q[0], (Tp)q[1], (Tp)q[2], (Tp)q[3]),
            Point3<Tp>((Tp)q[4], (Tp)q[5], (Tp)q[6])));
        votes.push_back(all_votes[(size_t)i]);
      }
      first.push_back((int)hyps.size());
    }
  };
  ransac_engine_batched<Tp>(adapter, spec, produce, commit, Iter, confidence, mask_cols, opt);
}

// The plain-RANSAC solvers with a 4-point sample in FAST scoring mode (solver 0 kneip_ransac, 1 shinji_kneip_ransac, 2 nl_kneip_ransac,
// 3 nl_shinji_ransac, 4 nl_shinji_kneip_ransac): batches beyond the first few are generated AND scored on the device
// (rpe_ransac_p3p_batch; one slot per hypothesis an iteration can yield, invalid slots skipped in the replay). The device P3P agrees
// with the host's to rounding only, which is why the vote-exact default never takes this path.
template <class Tp, class Adapter, class Gen, class Commit>
void ransac_engine_device_p3p(Adapter& adapter, const VoteSpec<Tp>& spec, int solver, Gen gen, Commit commit, int& Iter, Tp confidence,
    int mask_cols, const RunOptions& opt) {
  constexpr int kHostBatch = 32;
  Settings& cfg = Settings::get();
  rpe_context* ctx = adapter.device().ctx();
  static const int kSlots[5] = {1, 2, 1, 2, 3};
  const int per = kSlots[solver];
  std::vector<double> q7;
  std::vector<unsigned char> valid;
  std::vector<int> all_votes;
  auto produce = [&](int iters, std::vector<SE3<Tp> >& hyps, std::vector<int>& first, std::vector<int>& votes) {
    first.assign(1, 0);
    if (iters <= kHostBatch) {
      const double t0 = cfg.profile ? now_us() : 0;
      for (int i = 0; i < iters; i++) { gen(hyps); first.push_back((int)hyps.size()); }
      if (cfg.profile) { const double t1 = now_us(); cfg.prof.generate += t1 - t0; cfg.prof.score -= t1 - t0; }
      if (hyps.empty()) return;
      q7.resize(hyps.size() * 7);
      for (size_t h = 0; h < hyps.size(); h++) pose7<Tp>(hyps[h], &q7[7 * h]);
      votes.resize(hyps.size());
      check(rpe_score(ctx, spec.kind, opt.mode(), q7.data(), (int)hyps.size(), (double)spec.thre_3d, (double)spec.cos_thr,
                      (double)spec.cos_nl, votes.data()), "rpe_score");
      return;
    }
    Rand31& g = opt.stream();
    const size_t slots = (size_t)iters * per;
    q7.resize(slots * 7); valid.resize(slots); all_votes.resize(slots);
    for (int done = 0; done < iters;) {
      const int chunk = std::min(iters - done, 8192 / per);
      check(rpe_ransac_p3p_batch(ctx, solver, g.state(), g.inc(), chunk, (double)spec.thre_3d, (double)spec.cos_thr,
          (double)spec.cos_nl, all_votes.data() + (size_t)done * per,
                                 q7.data() + 7 * (size_t)done * per, valid.data() + (size_t)done * per), "rpe_ransac_p3p_batch");
      g.advance(4ull * (uint64_t)chunk);   // the host sampler draws 4 per iteration
      done += chunk;
    }
    for (int i = 0; i < iters; i++) {
      for (int k = 0; k < per; k++) {
        const size_t sl = (size_t)i * per + k;
        if (valid[sl]) {
          const double* q = &q7[7 * sl];
          hyps.push_back(SE3<Tp>(SO3<Tp>::fromQuaternionRaw((Tp)q[0], (Tp)q[1], (Tp)q[2], (Tp)q[3]),
              Point3<Tp>((Tp)q[4], (Tp)q[5], (Tp)q[6])));
          votes.push_back(all_votes[sl]);
        }
      }
      first.push_back((int)hyps.size());
    }
  };
  ransac_engine_batched<Tp>(adapter, spec, produce, commit, Iter, confidence, mask_cols, opt);
}

}  // namespace rpe

#endif
