// Batched hypothesis generation on the device for the 3D-3D RANSAC solvers (shinji_ransac,
// shinji_ransac2 -- the core of ao_ransac): one thread per RANSAC iteration draws its minimal sample from the SAME random stream
// the host sampler would have used (PCG32 skip-ahead to its position; RandomElements' partial Fisher-Yates over the identity
// table), gathers the three correspondences from the HBM-resident arrays, and runs the closed-form fit shinji() for K = 3
// (pose/AbsoluteOrientation.hpp; reference :47-99) -- by compiling the very function the host path runs (rpe::rigid_fit<T>,
// rpe/linalg.hpp, __host__ __device__: T arithmetic in the reference's operation order, Jacobi SVD) with FMA contraction off and
// IEEE division / square root, so that every hypothesis is BITWISE the one the host would have produced and the
// sequential replay in pose/RansacEngine.hpp reaches the same pose, votes, Iter and mask.  The poses land in HBM in the scoring
// kernel's layout (no staging, no H2D copy) and, as quaternion + translation, in pinned host memory for the replay.
#pragma clang fp contract(off)
#include "../include/rpe/linalg.hpp"
#include "rpe_kernels.h"

namespace rpe {
namespace {

struct Pcg32 {   // rpe::Rand31 (pose/Utility.hpp): PCG32 XSH-RR, output >> 1
  unsigned long long state, inc;
  __device__ unsigned int step() {
    const unsigned long long old = state;
    state = old * 6364136223846793005ULL + inc;
    const unsigned int xorshifted = (unsigned int)(((old >> 18u) ^ old) >> 27u), rot = (unsigned int)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
  }
  __device__ int next31() { return (int)(step() >> 1); }
  __device__ void advance(unsigned long long delta) {   // LCG skip-ahead (Brown, "Random number generation with arbitrary strides")
    unsigned long long cur_mult = 6364136223846793005ULL, cur_plus = inc, acc_mult = 1ULL, acc_plus = 0ULL;
    while (delta > 0) {
      if (delta & 1ULL) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
      cur_plus = (cur_mult + 1ULL) * cur_plus;
      cur_mult *= cur_mult;
      delta >>= 1;
    }
    state = acc_mult * state + acc_plus;
  }
};

template <class T> struct Eps;
template <> struct Eps<float> { __device__ static float value() { return 1e-5f; } };      // rpe::LieEps (rpe/types.hpp)
template <> struct Eps<double> { __device__ static double value() { return 1e-10; } };

// out_pose: scoring layout (exact: qw qx qy qz tx ty tz 0 ; fast: R row-major 9, t 3) in T.  h_q7: 8 T per iteration in pinned host
// memory (qw qx qy qz tx ty tz valid).
template <class T>
__global__ __launch_bounds__(64) void gen_shinji_kernel(const T* __restrict__ xw, const T* __restrict__ xc, int n, unsigned long long state,
                                                        unsigned long long inc, int iters, int exact, T* __restrict__ out_pose,
                                                        T* __restrict__ h_q7) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= iters) return;
  constexpr int K = 3;
  Pcg32 rng{state, inc};
  rng.advance((unsigned long long)K * (unsigned long long)i);
  // RandomElements::run(K): position j swaps with rnd() % (j + 1), j = n-1 ... n-K, over a table that starts as the identity
  int pos[2 * K], val[2 * K], cnt = 0, sel[K];
  auto get = [&](int p) { for (int k = 0; k < cnt; k++) if (pos[k] == p) return val[k]; return p; };
  auto set = [&](int p, int v) { for (int k = 0; k < cnt; k++) if (pos[k] == p) { val[k] = v; return; } pos[cnt] = p; val[cnt] = v; cnt++; };
  for (int s = 0, top = n - 1; s < K; s++, top--) {
    const int pick = rng.next31() % (top + 1);
    const int vp = get(pick), vt = get(top);
    set(pick, vt);
    set(top, vp);
    sel[s] = vp;
  }
  T X_w[3 * K], X_c[3 * K];   // 3 x K column-major, as the host's MatrixX
  bool valid = true;
  for (int s = 0; s < K; s++) {
    const T cx = xc[3 * (size_t)sel[s]], cy = xc[3 * (size_t)sel[s] + 1], cz = xc[3 * (size_t)sel[s] + 2];
    valid = valid && (cx == cx || cy == cy || cz == cz);   // isValid: not all three NaN
    X_c[3 * s] = cx; X_c[3 * s + 1] = cy; X_c[3 * s + 2] = cz;
    X_w[3 * s] = xw[3 * (size_t)sel[s]]; X_w[3 * s + 1] = xw[3 * (size_t)sel[s] + 1]; X_w[3 * s + 2] = xw[3 * (size_t)sel[s] + 2];
  }
  T q[4] = {T(1), T(0), T(0), T(0)}, t[3] = {T(0), T(0), T(0)};
  // shinji<T>(X_w, X_c, 3): the host's own function (rpe/linalg.hpp rigid_fit, T arithmetic in the reference's order).  A fit whose
  // rotation fails the SO3 constructor's test is skipped by the host solvers, so it is reported as not valid here too.
  if (valid) valid = rigid_fit<T>(X_w, X_c, K, K, Eps<T>::value(), q, t);
  T* hq = h_q7 + 8 * (size_t)i;
  hq[0] = q[0]; hq[1] = q[1]; hq[2] = q[2]; hq[3] = q[3]; hq[4] = t[0]; hq[5] = t[1]; hq[6] = t[2]; hq[7] = valid ? T(1) : T(0);
  if (exact) {
    T* o = out_pose + 8 * (size_t)i;
    o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3]; o[4] = t[0]; o[5] = t[1]; o[6] = t[2]; o[7] = T(0);
  } else {   // the host stages fast-mode poses as quat_to_R<double>(q) rounded to T
    const Quat<double> qd{(double)q[0], (double)q[1], (double)q[2], (double)q[3]};
    double Rd[9];
    quat_to_R<double>(qd, Rd);
    T* o = out_pose + 12 * (size_t)i;
    for (int k = 0; k < 9; k++) o[k] = (T)Rd[k];
    o[9] = t[0]; o[10] = t[1]; o[11] = t[2];
  }
}

}  // namespace

hipError_t launch_gen_shinji(const DeviceArrays& A, unsigned long long state, unsigned long long inc, int iters, int exact, void* d_poses,
                             void* h_q7, hipStream_t s) {
  if (iters < 1) return hipSuccess;
  const int G = (iters + 63) / 64;
  if (A.dtype) hipLaunchKernelGGL(gen_shinji_kernel<double>, dim3(G), dim3(64), 0, s, (const double*)A.a[0], (const double*)A.a[1], (int)A.n, state, inc, iters, exact, (double*)d_poses, (double*)h_q7);
  else hipLaunchKernelGGL(gen_shinji_kernel<float>, dim3(G), dim3(64), 0, s, (const float*)A.a[0], (const float*)A.a[1], (int)A.n, state, inc, iters, exact, (float*)d_poses, (float*)h_q7);
  return hipGetLastError();
}

}  // namespace rpe
