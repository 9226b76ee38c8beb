// CPU ThreadSanitizer build (tests/test_sanitizers_cpu.py): the C ABI is re-entrant.  Compiled TOGETHER with csrc/library.cpp (so
// the drop-in headers, samplers, minimal solvers and the RANSAC engine are instrumented) and oracle/oracle_capi.cpp under
// -fsanitize=thread.  Eight threads call rpe_host_hypotheses at once -- every solver, both dtypes, different seeds -- and each
// must get, bit for bit, the stream the same call yields alone AND the stream the oracle (orc_hypotheses) yields for that seed.
// The reference's samplers draw from the process-global rand() (/root/reference/pose/Utility.hpp:148,212,229) and are not
// thread-safe; here every call owns its stream (rpe::RunOptions::rng, built from the `seed` argument).  No GPU call is made.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "../../include/rgbd_pose_hip.h"
#include "AbsoluteOrientation.hpp"

extern "C" {
typedef struct { int n; const void* bv; const void* xc; const void* nc; const void* xw; const void* nw; const void* weights; int wcols; double fx, fy; } orc_problem;
int orc_hypotheses(int is_f64, int method, const orc_problem* p, int iters, uint64_t seed, double* q7_out, int cap, int* first_out);
}

static std::atomic<int> fails{0};
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

struct Lcg { uint64_t s; double u() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(s >> 11) / 9007199254740992.0; } double n() { return std::sqrt(-2 * std::log(u() + 1e-300)) * std::cos(6.283185307179586 * u()); } };

template <class T> struct Scene { std::vector<T> xw, xc, bv, nw, nc, w; };
template <class T> Scene<T> make_scene(int n, uint64_t seed) {
  Lcg g{seed};
  const double R[9] = {0.36, 0.48, -0.8, -0.8, 0.6, 0.0, 0.48, 0.64, 0.6}, t[3] = {0.3, -0.2, 0.5};
  Scene<T> s;
  s.xw.resize(3 * n); s.xc.resize(3 * n); s.bv.resize(3 * n); s.nw.resize(3 * n); s.nc.resize(3 * n); s.w.resize(3 * n);
  for (int i = 0; i < n; i++) {
    double pc[3] = {2 * g.u() - 1, 2 * g.u() - 1, 1.5 + 3 * g.u()}, pw[3], nrm[3] = {g.n(), g.n(), g.n() - 2}, nwv[3];
    const double d = std::sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]), nn = std::sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    for (int k = 0; k < 3; k++) { pw[k] = 0; nwv[k] = 0; for (int j = 0; j < 3; j++) { pw[k] += R[3 * j + k] * (pc[j] - t[j]); nwv[k] += R[3 * j + k] * nrm[j] / nn; } }
    for (int k = 0; k < 3; k++) {
      s.xw[3 * i + k] = (T)pw[k];
      s.xc[3 * i + k] = (T)(pc[k] + 0.02 * g.n());
      s.bv[3 * i + k] = (T)(pc[k] / d);
      s.nw[3 * i + k] = (T)nwv[k];
      s.nc[3 * i + k] = (T)(nrm[k] / nn);
      s.w[(size_t)k * n + i] = (T)(0.1 + g.u());
    }
    if (i % 17 == 3) for (int k = 0; k < 3; k++) s.xc[3 * i + k] = (T)NAN;
  }
  return s;
}

struct Stream { std::vector<double> q; std::vector<int> first; int h = -1; };

template <class T> Stream product_stream(const Scene<T>& s, int n, int is_f64, int method, int iters, uint64_t seed) {
  const bool need_bv = method != 1 && method != 2, need_xc = method != 3 && method != 4, need_n = method >= 7;
  rpe_problem p{n, is_f64 ? RPE_F64 : RPE_F32, need_bv ? s.bv.data() : nullptr, need_xc ? s.xc.data() : nullptr, need_n ? s.nc.data() : nullptr, s.xw.data(),
                need_n ? s.nw.data() : nullptr, s.w.data(), 3, 585.0, 585.0};
  Stream out;
  out.q.assign(7 * (size_t)(3 * iters + 1), 0.0); out.first.assign(iters + 1, 0);
  out.h = rpe_host_hypotheses(method, &p, iters, seed, out.q.data(), 3 * iters + 1, out.first.data());
  return out;
}
template <class T> Stream oracle_stream(const Scene<T>& s, int n, int is_f64, int method, int iters, uint64_t seed) {
  const bool need_bv = method != 1 && method != 2, need_xc = method != 3 && method != 4, need_n = method >= 7;
  orc_problem o{n, need_bv ? s.bv.data() : nullptr, need_xc ? s.xc.data() : nullptr, need_n ? s.nc.data() : nullptr, s.xw.data(),
                need_n ? s.nw.data() : nullptr, s.w.data(), 3, 585.0, 585.0};
  Stream out;
  out.q.assign(7 * (size_t)(3 * iters + 1), 0.0); out.first.assign(iters + 1, 0);
  out.h = orc_hypotheses(is_f64, method, &o, iters, seed, out.q.data(), 3 * iters + 1, out.first.data());
  return out;
}
static bool same(const Stream& a, const Stream& b) {
  return a.h == b.h && a.h > 0 && a.first == b.first && std::memcmp(a.q.data(), b.q.data(), sizeof(double) * 7 * (size_t)a.h) == 0;
}

int main() {
  const int n = 500, iters = 30, threads = 8, reps = 6;
  const Scene<float> sf = make_scene<float>(n, 5);
  const Scene<double> sd = make_scene<double>(n, 6);
  // what every (thread, method) yields alone, and what the oracle yields for the same seed
  std::vector<std::vector<Stream> > alone(threads, std::vector<Stream>(10));
  for (int t = 0; t < threads; t++)
    for (int m = 0; m < 10; m++) {
      const uint64_t seed = 100 + 7 * (uint64_t)t;
      alone[t][m] = (t & 1) ? product_stream(sd, n, 1, m, iters, seed) : product_stream(sf, n, 0, m, iters, seed);
      const Stream o = (t & 1) ? oracle_stream(sd, n, 1, m, iters, seed) : oracle_stream(sf, n, 0, m, iters, seed);
      CHECK(same(alone[t][m], o));
    }
  // the same calls, all threads at once
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++)
    th.emplace_back([&, t] {
      const uint64_t seed = 100 + 7 * (uint64_t)t;
      for (int r = 0; r < reps; r++)
        for (int m = 0; m < 10; m++) {
          const int method = (m + t) % 10;   // neighbours run different solvers at the same moment
          const Stream got = (t & 1) ? product_stream(sd, n, 1, method, iters, seed) : product_stream(sf, n, 0, method, iters, seed);
          CHECK(same(got, alone[t][method]));
        }
    });
  for (std::thread& x : th) x.join();
  // the drop-in free functions keep the reference's semantics when no stream is passed: ONE process-global stream, advanced by
  // every draw (rpe::seed replaces srand); with a stream of their own they leave it alone
  {
    rpe::seed(9);
    const uint64_t before = rpe::global_rng().state();
    RandomElements<int> re(100);
    std::vector<int> a, b;
    re.run(4, &a);
    CHECK(rpe::global_rng().state() != before);
    const uint64_t mid = rpe::global_rng().state();
    rpe::Rand31 own(9);
    re.run(4, &b, own);
    CHECK(rpe::global_rng().state() == mid && a == b);
    rpe::RunOptions opt;
    CHECK(&opt.stream() == &rpe::global_rng());
    opt.rng = &own;
    CHECK(&opt.stream() == &own);
    CHECK(opt.mode() == rpe::Settings::get().score_mode);
    opt.score_mode = RPE_SCORE_FAST;
    CHECK(opt.mode() == RPE_SCORE_FAST);
  }
  std::printf(fails.load() ? "reentrancy_host: %d FAILED\n" : "reentrancy_host: ok\n", fails.load());
  return fails.load() ? 1 : 0;
}
