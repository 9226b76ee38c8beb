// examples/engine_profile.cpp -- where the adapter-level solvers spend their wall time on a dense frame (development aid
// behind DESIGN.md's pipeline analysis): the reference's solvers on N = 307200 simulated correspondences, float, run twice
// per solver (second run: arrays already resident), with the RANSAC engine's phase timers on.
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include "AbsoluteOrientationNormal.hpp"
#include "GaussNewton.hpp"
#include "P3P.hpp"
#include "Simulator.hpp"

typedef float T;

int main(int argc, char** argv) {
  const int N = argc > 1 ? std::atoi(argv[1]) : 307200;
  const bool fresh = argc > 2 && std::string(argv[argc - 1]) == "fresh";   // last argument "fresh": a FRESH adapter per run (rpe_run, the demos)
  const bool totals_only = argc > 2 && std::string(argv[2]) != "fresh";   // any other second argument: phase timers off (they also switch off the generate/score overlap)
  rpe::sim_seed(7);
  const rpe::Point3<T> t = generate_random_translation_uniform<T>(5.0);
  const rpe::SO3<T> R = generate_random_rotation<T>(M_PI / 2, false);
  rpe::MatrixX<T> Q, M, P, Nn, U, W(N, 3);
  simulate_2d_3d_nl_correspondences<T>(R, t, N, 3.0f, 0.1f, 0.05f, 0.1f, 0.035f, 0.1f, 0.4f, 8.0f, 585.0f, true, &Q, &M, &P, &Nn, &U, &W);
  const T thre_3d = 0.2f, thre_2d = 8.0f, thre_nl = 0.1f, conf = 0.99f;
  rpe::Settings& cfg = rpe::Settings::get();
  cfg.profile = !totals_only;
  struct Row { const char* name; std::function<void(NormalAOPoseAdapter<T>&, int&)> run; };
  const Row rows[] = {
      {"shinji_ransac", [&](NormalAOPoseAdapter<T>& a, int& it) { shinji_ransac<T>(a, thre_3d, it, conf); }},
      {"kneip_ransac", [&](NormalAOPoseAdapter<T>& a, int& it) { kneip_ransac<T>(a, thre_2d, it, conf); }},
      {"kneip_prosac", [&](NormalAOPoseAdapter<T>& a, int& it) { kneip_prosac<T>(a, thre_2d, it, conf); }},
      {"shinji_kneip_ransac", [&](NormalAOPoseAdapter<T>& a, int& it) { shinji_kneip_ransac<T>(a, thre_3d, thre_2d, it, conf); }},
      {"shinji_kneip_prosac", [&](NormalAOPoseAdapter<T>& a, int& it) { shinji_kneip_prosac<T>(a, thre_3d, thre_2d, it, conf); }},
      {"nl_shinji_kneip_ransac", [&](NormalAOPoseAdapter<T>& a, int& it) { nl_shinji_kneip_ransac<T>(a, thre_3d, thre_2d, thre_nl, it, conf); }},
  };
  if (fresh) {
    try {
      std::printf("%-24s %5s %9s | %8s %8s %8s %8s %8s %8s | %5s %4s %8s   (fresh adapter per run: every array uploaded again)\n", "solver (N)", "run", "total_us", "upload", "sort",
                  "generate", "score", "replay", "mask", "hyps", "bat", "votes");
      for (const Row& r : rows)
        for (int rep = 0; rep < 3; rep++) {
          cfg.prof = rpe::EngineProfile();
          int it = 300;
          const double t0 = rpe::now_us();
          NormalAOPoseAdapter<T> adapter(U, P, Nn, Q, M);
          adapter.setFocal(585.0f, 585.0f);
          const double t1 = rpe::now_us();
          r.run(adapter, it);
          const double dt = rpe::now_us() - t0;
          const rpe::EngineProfile& p = cfg.prof;
          std::printf("%-24s %5d %9.0f | %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f | %5d %4d %8d   ctor %.0f us\n", r.name, rep, dt, p.upload, p.sort, p.generate, p.score, p.replay,
                      p.mask, p.hypotheses, p.batches, adapter.getMaxVotes(), t1 - t0);
        }
      return 0;
    } catch (const rpe::DeviceError& e) {
      std::fprintf(stderr, "device error %d: %s\n", e.code, e.what());
      return 2;
    }
  }
  try {
    NormalAOPoseAdapter<T> adapter(U, P, Nn, Q, M);
    adapter.setFocal(585.0f, 585.0f);
    adapter.setWeights(W);
    std::printf("%-24s %5s %9s | %8s %8s %8s %8s %8s %8s | %5s %4s %8s\n", "solver (N=307200)", "run", "total_us", "upload", "sort", "generate", "score", "replay",
                "mask", "hyps", "bat", "votes");
    for (const Row& r : rows)
      for (int rep = 0; rep < 2; rep++) {
        cfg.prof = rpe::EngineProfile();
        int it = 1000;
        const double t0 = rpe::now_us();
        r.run(adapter, it);
        const double dt = rpe::now_us() - t0;
        const rpe::EngineProfile& p = cfg.prof;
        std::printf("%-24s %5d %9.0f | %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f | %5d %4d %8d\n", r.name, rep, dt, p.upload, p.sort, p.generate, p.score, p.replay, p.mask,
                    p.hypotheses, p.batches, adapter.getMaxVotes());
      }
    {  // the reference's 3D-3D entry point on the same frame (shinji_ransac2, AbsoluteOrientation.hpp:158-213): arrays resident, 8 runs
      AOOnlyPoseAdapter<T> ao(P, Q);
      ao.setFocal(585.0f, 585.0f);
      for (int rep = 0; rep < 8; rep++) {
        cfg.prof = rpe::EngineProfile();
        int it = 1000;
        const double t0 = rpe::now_us();
        shinji_ransac2<T>(ao, thre_3d, it, conf);
        const double dt = rpe::now_us() - t0;
        const rpe::EngineProfile& p = cfg.prof;
        std::printf("%-24s %5d %9.1f | %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f | %5d %4d %8d   Iter %d\n", "shinji_ransac2", rep, dt, p.upload, p.sort, p.generate, p.score,
                    p.replay, p.mask, p.hypotheses, p.batches, ao.getMaxVotes(), it);
      }
    }
    return 0;
  } catch (const rpe::DeviceError& e) {
    std::fprintf(stderr, "device error %d: %s\n", e.code, e.what());
    return 2;
  }
}
