// examples/gn_refine_main.cpp -- the headline loop of bench.py from plain C++ over the C ABI (include/rgbd_pose_hip.h): a 640 x 480
// frame of 3D-3D correspondences, one scoring pass for the inlier mask, then K Gauss-Newton iterations in ONE resident launch
// (rpe_gn_refine, host update through the control block), timed (1) as the process happens to be placed and (2) after
// rpe_tune_host_thread has measured a few CPUs and pinned this thread to the fastest -- the tuning bench.py's headline uses, one library
// call away for any caller (or RPE_HOST_CPU=auto in the environment).   usage: gn_refine_main [steps = 20] [repetitions = 50] [n = 307200]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../include/rgbd_pose_hip.h"

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHECK(call) do { const int rc_ = (call); if (rc_ != RPE_OK) { std::fprintf(stderr, "%s: %s\n", #call, rpe_last_error()); return 2; } } while (0)

int main(int argc, char** argv) {
  const int steps = argc > 1 ? std::atoi(argv[1]) : 20, reps = argc > 2 ? std::atoi(argv[2]) : 50, n = argc > 3 ? std::atoi(argv[3]) : 307200;
  // the frame: camera points in a frustum, world points through a known pose + 5 cm noise, 10 % gross outliers (Parameters.yml values)
  std::mt19937_64 g(7);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::normal_distribution<float> N01(0.f, 1.f);
  const double w[3] = {0.3, -0.2, 0.25}, th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double R[9], t[3] = {0.8, -0.4, 1.2};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double k2 = 0; for (int k = 0; k < 3; k++) k2 += K[3 * i + k] * K[3 * k + j];
    R[3 * i + j] = (i == j) + std::sin(th) / th * K[3 * i + j] + (1 - std::cos(th)) / (th * th) * k2;
  }
  std::vector<float> xw(3 * (size_t)n), xc(3 * (size_t)n);
  for (int i = 0; i < n; i++) {
    const float z = 0.4f + 3.8f * (U(g) + 1.f), x = U(g) * 0.55f * z, y = U(g) * 0.41f * z, pc[3] = {x, y, z};
    for (int k = 0; k < 3; k++) xc[3 * (size_t)i + k] = pc[k];
    const bool out = (i % 10) == 3;
    for (int k = 0; k < 3; k++) {   // Xw = R^T (Xc - t) + noise
      double v = 0; for (int j = 0; j < 3; j++) v += R[3 * j + k] * (pc[j] - t[j]);
      xw[3 * (size_t)i + k] = (float)v + 0.05f * N01(g) + (out ? 2.f * U(g) : 0.f);
    }
  }
  rpe_context* ctx = nullptr;
  CHECK(rpe_create(&ctx, 0, nullptr));
  CHECK(rpe_set_problem(ctx, n, RPE_F32));
  CHECK(rpe_upload(ctx, RPE_XW, xw.data()));
  CHECK(rpe_upload(ctx, RPE_XC, xc.data()));
  // start pose: the truth shifted by 3 cm (what a RANSAC winner looks like); its inliers at 0.2 m
  double pose0[12];
  for (int i = 0; i < 9; i++) pose0[i] = R[i];
  for (int i = 0; i < 3; i++) pose0[9 + i] = t[i] + 0.03;
  {
    // quaternion of R for the scoring entry point
    const double tr = R[0] + R[4] + R[8], qw = 0.5 * std::sqrt(1 + tr), q7[7] = {qw, (R[7] - R[5]) / (4 * qw), (R[2] - R[6]) / (4 * qw), (R[3] - R[1]) / (4 * qw),
                                                                                  pose0[9], pose0[10], pose0[11]};
    int votes = 0;
    CHECK(rpe_inlier_mask(ctx, RPE_VOTE_33, RPE_SCORE_EXACT, q7, 0.2, 2.0, 2.0, &votes));
    std::printf("%d correspondences, %d inliers at 0.2 m\n", n, votes);
  }
  const int kind = RPE_RES_P2P;
  auto region = [&](double* us_per_step) -> int {   // `reps` regions of exactly `steps` iterations, median
    std::vector<double> ts;
    for (int r = 0; r < reps + 3; r++) {
      double p[12];
      std::copy(pose0, pose0 + 12, p);
      int its = 0;
      double st = 0, co = 0;
      if (rpe_synchronize(ctx)) return 1;
      const double t0 = now_us();
      if (rpe_gn_refine(ctx, 1, &kind, nullptr, RPE_USE_MASK, p, steps, 0.0, &its, &st, &co)) return 1;
      if (rpe_synchronize(ctx)) return 1;
      if (r >= 3) ts.push_back((now_us() - t0) / steps);
    }
    std::sort(ts.begin(), ts.end());
    *us_per_step = ts[ts.size() / 2];
    return 0;
  };
  { double p[12]; std::copy(pose0, pose0 + 12, p); int its; double st, co;   // warm the GPU for a second
    const double t0 = now_us(); while (now_us() - t0 < 1e6) CHECK(rpe_gn_refine(ctx, 1, &kind, nullptr, RPE_USE_MASK, p, steps, 0.0, &its, &st, &co)); }
  double untuned = 0, tuned = 0, best_us = 0;
  if (region(&untuned)) { std::fprintf(stderr, "%s\n", rpe_last_error()); return 2; }
  int cpu = -1, cpus[16], nt = 0;
  double us[16];
  CHECK(rpe_tune_host_thread(ctx, kind, RPE_USE_MASK, pose0, steps, std::max(3, std::min(12, 6000 / steps)), &cpu, &best_us, cpus, us, 16, &nt));
  if (region(&tuned)) { std::fprintf(stderr, "%s\n", rpe_last_error()); return 2; }
  std::printf("{\"steps\": %d, \"repetitions\": %d, \"us_per_step_untuned\": %.3f, \"us_per_step_tuned\": %.3f, \"cpu\": %d, \"trials\": {", steps, reps, untuned, tuned, cpu);
  for (int i = 0; i < nt; i++) std::printf("%s\"%d\": %.3f", i ? ", " : "", cpus[i], us[i]);
  std::printf("}}\n");
  rpe_destroy(ctx);
  return 0;
}
