// GPU-side semantics of the drop-in adapters that the oracle comparison does not see (compiled and run by
// tests/test_gpu_examples.py): masks adopted on the device and fetched lazily, the reference's column semantics of setInlier,
// index lists, invalidateDevice(), host edits pushed back before a least-squares stage.
#include <cstdio>
#include <cstdlib>
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

int main() {
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
    std::printf(fails ? "device_logic: %d FAILURES\n" : "device_logic: ok\n", fails);
    return fails ? 1 : 0;
  } catch (const rpe::DeviceError& e) {
    std::fprintf(stderr, "device error %d: %s\n", e.code, e.what());
    return 2;
  }
}
