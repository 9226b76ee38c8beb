// GPU-side semantics of the drop-in adapters that the oracle comparison does not see (compiled and run by
// tests/test_gpu_examples.py): masks adopted on the device and fetched lazily, the reference's column semantics of setInlier,
// index lists, invalidateDevice(), host edits pushed back before a least-squares stage.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include "AbsoluteOrientation.hpp"
#include "AbsoluteOrientationNormal.hpp"
#include "GaussNewton.hpp"
#include "P3P.hpp"
#include "Simulator.hpp"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)
typedef float T;

template <class A> static int count23(const A& a, int N) { int k = 0; for (int i = 0; i < N; i++) k += a.isInlier23(i); return k; }
template <class A> static int count33(const A& a, int N) { int k = 0; for (int i = 0; i < N; i++) k += a.isInlier33(i); return k; }

// host generation vs device generation of the 3D-3D RANSAC hypotheses: same random stream -> bitwise the same run
template <class Tp>
static void check_device_hypotheses(int N, double nan_frac, int score_mode, unsigned long long seed) {
  rpe::sim_seed(seed);
  const rpe::Point3<Tp> t = generate_random_translation_uniform<Tp>(5.0);
  const rpe::SO3<Tp> R = generate_random_rotation<Tp>(M_PI / 2, false);
  rpe::MatrixX<Tp> Q, P, U, W(N, 3);
  simulate_2d_3d_3d_correspondences<Tp>(R, t, N, (Tp)2.0, (Tp)0.03, (Tp)0.2, (Tp)0.4, (Tp)8.0, (Tp)585.0, true, &Q, &U, &P, &W);
  for (int i = 0; i < N; i++) if ((i * 2654435761u % 1000) < nan_frac * 1000) { P(0, i) = P(1, i) = P(2, i) = std::numeric_limits<Tp>::quiet_NaN(); }
  rpe::Settings& cfg = rpe::Settings::get();
  cfg.score_mode = score_mode;
  struct Out { int votes, iter; rpe::Quat<Tp> q; rpe::Point3<Tp> t; std::vector<short> m33; unsigned long long rng; };
  Out out[2];
  for (int dev = 0; dev < 2; dev++) {
    cfg.device_hypotheses = dev == 1;
    AOPoseAdapter<Tp> a(U, P, Q);
    a.setFocal((Tp)585, (Tp)585);
    rpe::seed(seed + 17);
    int it = 700;
    shinji_ransac<Tp>(a, (Tp)0.1, it, (Tp)0.999);
    out[dev].votes = a.getMaxVotes(); out[dev].iter = it; out[dev].q = a.getRcw().unit_quaternion(); out[dev].t = a.gettw();
    const AOPoseAdapter<Tp>& ca = a;
    out[dev].m33 = ca.inlierMask33();
    out[dev].rng = rpe::global_rng().state();
  }
  cfg.device_hypotheses = true; cfg.score_mode = RPE_SCORE_FAST;
  CHECK(out[0].votes == out[1].votes && out[0].iter == out[1].iter && out[0].votes > 0);
  CHECK(std::memcmp(&out[0].q, &out[1].q, sizeof(out[0].q)) == 0 && std::memcmp(&out[0].t, &out[1].t, sizeof(out[0].t)) == 0);
  CHECK(out[0].m33 == out[1].m33);
  CHECK(out[0].rng == out[1].rng);   // both advanced the random stream by the same number of draws
}

int main() {
  for (unsigned long long seed = 1; seed <= 4; seed++) {
    check_device_hypotheses<float>(5000, 0.0, RPE_SCORE_EXACT, seed);
    check_device_hypotheses<float>(20011, 0.15, RPE_SCORE_FAST, seed);
    check_device_hypotheses<double>(3001, 0.1, RPE_SCORE_EXACT, seed);
    check_device_hypotheses<double>(9, 0.0, RPE_SCORE_FAST, seed);
  }
  const int N = 20000;
  rpe::sim_seed(3);
  const rpe::Point3<T> t = generate_random_translation_uniform<T>(5.0);
  const rpe::SO3<T> R = generate_random_rotation<T>(M_PI / 2, false);
  rpe::MatrixX<T> Q, M, P, Nn, U, W(N, 3);
  simulate_2d_3d_nl_correspondences<T>(R, t, N, 2.0f, 0.1f, 0.03f, 0.1f, 0.035f, 0.1f, 0.4f, 8.0f, 585.0f, true, &Q, &M, &P, &Nn, &U, &W);
  try {
    NormalAOPoseAdapter<T> a(U, P, Nn, Q, M);
    a.setFocal(585.f, 585.f);
    AOPoseAdapter<T>* a33 = &a;
    PnPPoseAdapter<T>* a23 = &a;
    // ---- shinji_ransac votes on the 3-D modality only: its N x 2 matrix has column 0 all zero (AbsoluteOrientation.hpp:134-143)
    int it = 500;
    shinji_ransac<T>(a, 0.15f, it, 0.99f);
    CHECK(a.getMaxVotes() > N / 2);
    CHECK(count33(a, N) == a.getMaxVotes());            // host copy fetched on first read == what the kernel counted
    CHECK((int)a33->getInlierIdx().size() == a.getMaxVotes());
    CHECK(count23(a, N) == 0);
    const rpe::SO3<T> R1 = a.getRcw();
    // ---- kneip_ransac votes on 2-D only (N x 1): the 3-D mask stays as it was
    it = 500;
    kneip_ransac<T>(a, 8.0f, it, 0.99f);
    const int v23 = a.getMaxVotes();
    CHECK(v23 > N / 2 && count23(a, N) == v23 && (int)a23->getInlierIdx().size() == v23);
    CHECK(count33(a, N) == (int)a33->getInlierIdx().size());
    // ---- lsq_pnp (reference P3P.hpp:472-502) at the pose kneip_ransac left: ONE device pass; its terms are getError(i) bit for bit and
    // are added in double, so the total equals the host's double sum of the same float terms to the last few bits
    {
      double host = 0;
      for (int i = 0; i < N; i++) host += (double)a23->getError(i);
      const T dev = lsq_pnp<T>(*a23);
      CHECK(std::fabs((double)dev - host) <= 2e-7 * host && host > 0);
    }
    // ---- all three modalities
    it = 500;
    nl_shinji_kneip_ransac<T>(a, 0.15f, 8.0f, 0.1f, it, 0.99f);
    int nn = 0;
    for (int i = 0; i < N; i++) nn += a.isInlierNN(i);
    CHECK(count23(a, N) + count33(a, N) + nn == a.getMaxVotes());
    CHECK((int)a.getInlierIdx().size() == nn);
    // ---- invalidateDevice(): the device copies go away, the host copies must have been brought up to date first
    const int c23 = count23(a, N), c33 = count33(a, N);
    it = 500;
    shinji_kneip_ransac<T>(a, 0.15f, 8.0f, it, 0.99f);   // new masks, on the device only
    const int votes = a.getMaxVotes();
    a.invalidateDevice();
    CHECK(count23(a, N) + count33(a, N) == votes);
    (void)c23; (void)c33;
    // ---- a host edit reaches the device before the next least-squares stage: keep 3 inliers only -> the fit goes through them
    rpe::MatrixXs m(N, 2);
    for (int i = 0; i < 3; i++) m(7 * i + 1, 1) = 1;
    a.setInlier(m);
    shinji_ls<T>(a);
    for (int i = 0; i < 3; i++) {
      const int c = 7 * i + 1;
      const rpe::Point3<T> e = a.getRcw() * a.getPointGlob(c) + a.gettw() - a.getPointCurr(c);
      CHECK(e.norm() < 0.12f);
    }
    std::vector<short>& edit = a33->inlierMask33();      // writable host copy: the device copy is refreshed on the next use
    for (int i = 0; i < N; i++) edit[i] = 1;
    shinji_ls<T>(a);
    const rpe::Matrix3<T> D = a.getRcw().matrix() * R.matrix().transpose();
    CHECK(D(0, 0) + D(1, 1) + D(2, 2) > 2.99f);          // all points again: close to the true rotation despite 10 % outliers
    (void)R1;
    // ---- the caller refills the SAME matrices with another frame (the adapters hold references, the reference re-reads them on every
    // call: AOOnlyPoseAdapter.hpp:93-95): the next solver run must see the new content, not the HBM copy keyed by the old address
    {
      rpe::sim_seed(11);
      const rpe::Point3<T> tA = generate_random_translation_uniform<T>(5.0);
      const rpe::SO3<T> RA = generate_random_rotation<T>(M_PI / 2, false);
      rpe::MatrixX<T> QA, PA;
      simulate_3d_3d_correspondences<T>(RA, tA, N, 0.01f, 0.0f, 0.4f, 8.0f, 585.0f, true, &QA, &PA, nullptr);
      AOOnlyPoseAdapter<T> b(PA, QA);
      shinji_ls2<T>(b);
      const rpe::Matrix3<T> DA = b.getRcw().matrix() * RA.matrix().transpose();
      CHECK(DA(0, 0) + DA(1, 1) + DA(2, 2) > 2.9999f);
      rpe::sim_seed(12);
      const rpe::Point3<T> tB = generate_random_translation_uniform<T>(5.0);
      const rpe::SO3<T> RB = generate_random_rotation<T>(M_PI / 2, false);
      rpe::MatrixX<T> QB, PB;
      simulate_3d_3d_correspondences<T>(RB, tB, N, 0.01f, 0.0f, 0.4f, 8.0f, 585.0f, true, &QB, &PB, nullptr);
      std::memcpy(QA.data(), QB.data(), sizeof(T) * 3 * N);   // in place: same addresses, new frame
      std::memcpy(PA.data(), PB.data(), sizeof(T) * 3 * N);
      shinji_ls2<T>(b);
      const rpe::Matrix3<T> DB = b.getRcw().matrix() * RB.matrix().transpose();
      CHECK(DB(0, 0) + DB(1, 1) + DB(2, 2) > 2.9999f);        // the second frame's pose, not the first's
      const rpe::Point3<T> dt = b.gettw() - tB;
      CHECK(dt.norm() < 0.01f);
    }
    // ---- a SPARSE in-place edit on a full frame: three camera points are gross outliers; the caller then NaN-marks exactly those
    // columns in its own matrix (AOPoseAdapter.hpp:147-152 "invalid measurement") and refines again.  The sampled fingerprint (32 of
    // 57 600 cache lines) does not see three columns change; Settings::FP_FULL (RPE_FINGERPRINT=full) does, and invalidateDevice() is
    // the explicit way.
    {
      const int NB = 307200;
      rpe::sim_seed(21);
      const rpe::Point3<T> tC = generate_random_translation_uniform<T>(5.0);
      const rpe::SO3<T> RC = generate_random_rotation<T>(M_PI / 2, false);
      rpe::MatrixX<T> QC, PC;
      simulate_3d_3d_correspondences<T>(RC, tC, NB, 0.01f, 0.0f, 0.4f, 8.0f, 585.0f, true, &QC, &PC, nullptr);
      const int bad[3] = {12345, 150001, 290000};
      for (int c : bad) PC.setCol(c, PC.col(c) + rpe::Point3<T>(T(300), T(-200), T(250)));
      auto refine_err = [&](AOOnlyPoseAdapter<T>& ad) {
        ad.setRcw(RC); ad.sett(tC + rpe::Point3<T>(T(0.01), T(0.01), T(-0.01)));
        gn_refine_p2p<T>(ad, 20, 1e-9, /*use_inliers=*/false);
        return (double)(ad.gettw() - tC).norm();
      };
      const int saved = rpe::Settings::get().fingerprint;
      for (int mode : {(int)rpe::Settings::FP_SAMPLED, (int)rpe::Settings::FP_FULL}) {
        rpe::Settings::get().fingerprint = mode;
        rpe::MatrixX<T> P2 = PC;                              // a fresh matrix per mode (its own address)
        AOOnlyPoseAdapter<T> ad(P2, QC);
        const double e_out = refine_err(ad);
        CHECK(e_out > 1e-3);                                  // three points 440 m off pull the translation by millimetres
        for (int c : bad) P2.setCol(c, rpe::Point3<T>(T(NAN), T(NAN), T(NAN)));   // in place, three columns of 307 200
        const double e_marked = refine_err(ad);
        if (mode == rpe::Settings::FP_FULL) CHECK(e_marked < 2e-4);             // seen: the frame was uploaded again, the columns are skipped
        else {
          CHECK(e_marked > 1e-3);                             // NOT seen (documented limit of the sampled fingerprint) ...
          ad.invalidateDevice();
          CHECK(refine_err(ad) < 2e-4);                       // ... until the caller says so
        }
      }
      rpe::Settings::get().fingerprint = saved;
    }
    std::printf(fails ? "device_logic: %d FAILURES\n" : "device_logic: ok\n", fails);
    return fails ? 1 : 0;
  } catch (const rpe::DeviceError& e) {
    std::fprintf(stderr, "device error %d: %s\n", e.code, e.what());
    return 2;
  }
}
