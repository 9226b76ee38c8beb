// CPU sanitizer build ("-fsanitize=address,undefined for host code"; GPU ASan is not available on this pool).
// This driver is compiled TOGETHER with csrc/library.cpp (the host side of the product: drop-in headers, samplers, minimal solvers,
// RANSAC engine in capture mode, rpe_host_*) and oracle/oracle_capi.cpp (the CPU restatement) with
// -fsanitize=address,undefined -fno-sanitize-recover=all, and run by tests/test_sanitizers_cpu.py.  No GPU call is made: the
// GPU-facing C ABI is only linked (librgbdpose_hip.so), never entered.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <unistd.h>
#include <vector>
#include "../../include/rgbd_pose_hip.h"

extern "C" {
typedef struct { int n; const void* bv; const void* xc; const void* nc; const void* xw; const void* nw; const void* weights; int wcols; double fx, fy; } orc_problem;
int orc_hypotheses(int is_f64, int method, const orc_problem* p, int iters, uint64_t seed, double* q7_out, int cap, int* first_out);
int orc_run(int is_f64, int method, const orc_problem* p, double thre_3d, double thre_2d, double thre_nl, int* iter_io, double confidence, uint64_t seed, int ls,
            int adapter_kind_for_none, const short* mask_in, double* R9, double* t3, int* max_votes, short* mask_out);
int orc_run_replay(int is_f64, int method, const orc_problem* p, const double* poses7, const int* first, int list_iters, double thre_3d, double thre_2d,
                   double thre_nl, int* iter_io, double confidence, int ls, double* R9, double* t3, int* max_votes, short* mask_out);
int orc_shinji(int is_f64, const void* xw, const void* xc, int n, int K, double* R9, double* t3);
void orc_ao(float* x_w, float* x_c, int n, float* R_cw, float* t);
void orc_svd3(const double* A9, double* U9, double* s3, double* V9);
void orc_se3_exp(const double* a6, double* R9, double* t3);
}

static int fails = 0;
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
    const bool outlier = g.u() < 0.15;
    for (int k = 0; k < 3; k++) {
      s.xw[3 * i + k] = (T)pw[k];
      s.xc[3 * i + k] = (T)(pc[k] + 0.02 * g.n() + (outlier ? g.n() : 0));
      s.bv[3 * i + k] = (T)(pc[k] / d);   // unit bearing (P3P builds an orthonormal frame from it; the reference aborts on a non-unit one)
      s.nw[3 * i + k] = (T)nwv[k];
      s.nc[3 * i + k] = (T)(nrm[k] / nn);
      s.w[(size_t)k * n + i] = (T)(0.1 + g.u());
    }
    if (i % 17 == 3) for (int k = 0; k < 3; k++) s.xc[3 * i + k] = (T)NAN;   // isValid() == false
    {  // a little angular noise on the bearing, renormalised
      double b[3] = {pc[0] / d + 0.002 * g.n(), pc[1] / d + 0.002 * g.n(), pc[2] / d + 0.002 * g.n()};
      const double bn = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
      for (int k = 0; k < 3; k++) s.bv[3 * i + k] = (T)(b[k] / bn);
    }
  }
  return s;
}

template <class T> void streams(int is_f64) {
  const int n = 600, iters = 60;
  Scene<T> s = make_scene<T>(n, 42 + is_f64);
  for (int method = 0; method < 10; method++) {
    const bool need_bv = method != 1 && method != 2, need_xc = method != 3 && method != 4, need_n = method >= 7;
    rpe_problem p{n, is_f64 ? RPE_F64 : RPE_F32, need_bv ? s.bv.data() : nullptr, need_xc ? s.xc.data() : nullptr, need_n ? s.nc.data() : nullptr, s.xw.data(),
                  need_n ? s.nw.data() : nullptr, s.w.data(), 3, 585.0, 585.0};
    orc_problem o{n, p.bv, p.xc, p.nc, p.xw, p.nw, p.weights, 3, 585.0, 585.0};
    std::vector<double> q1(7 * (3 * iters + 1)), q2(q1.size());
    std::vector<int> f1(iters + 1), f2(iters + 1);
    const int h1 = rpe_host_hypotheses(method, &p, iters, 7, q1.data(), 3 * iters + 1, f1.data());
    const int h2 = orc_hypotheses(is_f64, method, &o, iters, 7, q2.data(), 3 * iters + 1, f2.data());
    CHECK(h1 == h2 && h1 > 0);
    CHECK(f1 == f2);
    CHECK(h1 != h2 || std::memcmp(q1.data(), q2.data(), sizeof(double) * 7 * (size_t)h1) == 0);
    // the oracle's whole pipeline on the CPU (vote loops, adaptive Iter, masks) and its replay of the product's list
    int it = 40, mv = 0;
    double R9[9], t3[3];
    std::vector<short> mask(3 * n);
    orc_run(is_f64, method, &o, 0.1, 4.0, 0.15, &it, 0.99, 5, 0, 0, nullptr, R9, t3, &mv, mask.data());
    CHECK(mv >= 0 && it >= 0 && it <= 40);
    int it2 = iters, mv2 = 0;
    orc_run_replay(is_f64, method, &o, q1.data(), f1.data(), iters, 0.1, 4.0, 0.15, &it2, 0.99, 0, R9, t3, &mv2, mask.data());
    CHECK(mv2 > 0 && it2 <= iters);
  }
}

int main() {
  streams<float>(0);
  streams<double>(1);
  // samplers / small algebra of the product's host side
  std::vector<int> idx(4 * 50);
  rpe_host_random_elements(1000, 4, 3, 50, idx.data());
  for (int v : idx) CHECK(v >= 0 && v < 1000);
  rpe_host_prosac_samples(RPE_F32, 4, 1000, 3, 50, idx.data());
  rpe_host_prosac_samples(RPE_F64, 4, 5, 3, 50, idx.data());       // n close to m: the clamped n-th point
  for (int v : idx) CHECK(v >= 0 && v < 1000);
  CHECK(rpe_host_update_num_iters(RPE_F32, 0.99, 0.5, 3, 1000) > 0);
  std::vector<double> w(300);
  Lcg g{9};
  for (double& x : w) x = std::floor(10 * g.u());
  std::vector<int> order(300);
  rpe_host_sort_indexes(w.data(), 300, order.data());
  for (int i = 1; i < 300; i++) CHECK(w[order[i - 1]] > w[order[i]] || (w[order[i - 1]] == w[order[i]] && order[i - 1] < order[i]));
  double A[9], U[9], sv[3], V[9], Uo[9], so[3], Vo[9];
  for (int rep = 0; rep < 200; rep++) {
    for (double& a : A) a = g.n();
    if (rep % 5 == 0) for (int k = 0; k < 3; k++) A[6 + k] = A[k];       // rank deficient
    rpe_host_svd3(A, U, sv, V);
    orc_svd3(A, Uo, so, Vo);
    CHECK(std::memcmp(U, Uo, sizeof(U)) == 0 && std::memcmp(V, Vo, sizeof(V)) == 0 && std::memcmp(sv, so, sizeof(sv)) == 0);
    CHECK(sv[0] >= sv[1] && sv[1] >= sv[2] && sv[2] >= 0);
  }
  const double a6[6] = {0.1, -0.2, 0.3, 0.02, -0.01, 0.03};
  double R9[9], t3[3], R9o[9], t3o[3];
  rpe_host_se3_exp(a6, R9, t3);
  orc_se3_exp(a6, R9o, t3o);
  for (int k = 0; k < 9; k++) CHECK(std::fabs(R9[k] - R9o[k]) < 1e-14);
  double err[2], pct[2];
  rpe_host_calc_err(R9, t3, R9o, t3o, err, pct);
  CHECK(err[0] < 1e-12 && err[1] < 1e-6);
  // degenerate inputs must not trip the sanitizers either: collinear P3P sample, zero matrices, NaN sample
  double xw4[12] = {0, 0, 1, 0, 0, 2, 0, 0, 3, 1, 1, 1}, bv4[12] = {0, 0, 1, 0, 0, 1, 0, 0, 1, 0.1, 0, 0.99}, sols[48];
  CHECK(rpe_host_kneip_main(RPE_F32, xw4, bv4, sols) == 0);
  xw4[0] = NAN;
  (void)rpe_host_kneip_main(RPE_F64, xw4, bv4, sols);
  std::memset(A, 0, sizeof(A));
  rpe_host_svd3(A, U, sv, V);
  CHECK(sv[0] == 0 && U[0] == 1 && V[4] == 1);
  // the host-side exchange between rank processes (csrc/rpe_hostex.cpp), three "ranks" as threads of this process, each with its own mapping
  {
    char name[64];
    std::snprintf(name, sizeof name, "/rpe_hx_san_%d", (int)getpid());
    const int world = 3, steps = 300;
    int bad[3] = {0, 0, 0};
    auto rank_main = [&](int rank) {
      rpe_host_exchange* hx = nullptr;
      if (rpe_host_exchange_open(name, world, rank, rank == 0, 20.0, &hx) != RPE_OK) { bad[rank] = 1000; return; }
      for (int s = 1; s <= steps; s++) {
        const int n = 1 + (s * 5) % 64;
        double v[64];
        for (int i = 0; i < n; i++) v[i] = (rank + 1) * 0.5 + i + s;
        if (rpe_host_exchange_allreduce_f64(hx, v, n) != RPE_OK) { bad[rank]++; continue; }
        for (int i = 0; i < n; i++) bad[rank] += v[i] != ((0.5 + i + s) + (1.0 + i + s)) + (1.5 + i + s);
        if (s % 4 == 0) {
          std::vector<int> c(1 + (s * 37) % 8192, rank + 1);
          if (rpe_host_exchange_allreduce_i32(hx, c.data(), (int)c.size()) != RPE_OK) { bad[rank]++; continue; }
          for (int x : c) bad[rank] += x != 6;
        }
      }
      if (rank == 0) rpe_host_exchange_unlink(hx);
      rpe_host_exchange_close(hx);
    };
    std::thread t1(rank_main, 1), t2(rank_main, 2);
    rank_main(0);
    t1.join(); t2.join();
    CHECK(bad[0] == 0 && bad[1] == 0 && bad[2] == 0);
    double one[1] = {1.0};
    CHECK(rpe_host_exchange_allreduce_f64(nullptr, one, 1) != RPE_OK);
  }
  std::printf(fails ? "sanitize_host: %d FAILED\n" : "sanitize_host: ok\n", fails);
  return fails ? 1 : 0;
}
