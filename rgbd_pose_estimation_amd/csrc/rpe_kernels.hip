// gfx950 (MI355X, CDNA4) kernels of the RGB-D absolute-pose hot path.  Hand-written HIP, wave64.
//
// All kernels are streaming reductions / counts over the correspondence index c of the reference's O(N)
// loops.  They are HBM-bandwidth bound (about 2.5 flop/B), so there is no MFMA here:
// the design rules are (1) 16-byte vector loads of the reference's native xyz-interleaved 3 x N arrays --
// a thread owns P consecutive correspondences (P = 4 for fp32 = three float4, P = 2 for fp64 = three
// double2), so every byte of every 128-B line is consumed by one lane within three back-to-back loads;
// (2) per-thread fp64 accumulators fed by per-group sums in the array dtype, a reduce-scatter across the 64 lanes
// (v_permlane32/16_swap + DPP, no LDS traffic), one LDS hop across the waves of a workgroup, one 256-B partial record per workgroup;
// (3) the second stage inside the SAME launch: write-through records, a two-level arrival count, and the last
// workgroup sums the records in a fixed order (deterministic, no float atomics), expands them to the 6x6 / 6x1
// normal equations and publishes them -- to HBM, to pinned host memory, to the peers' mailboxes over xGMI, or
// straight into an in-kernel 6x6 solve + SE(3) update (reduce_and_finish);
// (4) grids of at most a few workgroups per CU with a grid-stride, software-pipelined loop, so a launch covers
// all 8 XCDs and a workgroup re-reads the same slice every Gauss-Newton iteration (it stays cache resident).
#include "rpe_kernels.h"
#include <cstring>
#include <algorithm>
#include <hip/hip_ext.h>
#include "rpe_assoc.h"

namespace rpe {

// ---- diagnostic build only (-DRPE_STAMPS, scripts/tail_timeline.py): thread 0 of every workgroup stamps the 100 MHz constant clock at the
// phase boundaries of the reduction kernels into a buffer of its own (16 words per workgroup); no stamp exists in the product build.
#ifdef RPE_STAMPS
__device__ unsigned long long g_stamps[4096 * 16];
#define RPE_STAMP(k)                                                                                             \
  do {                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    if (threadIdx.x == 0) {                                                                                      \
      unsigned long long t_;                                                                                     \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory");                          \
      g_stamps[(size_t)blockIdx.x * 16 + (k)] = t_;                                                              \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  } while (0)
#else
#define RPE_STAMP(k) do {} while (0)
#endif

enum { KIND_P2P = 0, KIND_P2PLANE = 1, KIND_BEARING = 2 };
enum { F_USE_MASK = 1, F_USE_WEIGHT = 2, F_SKIP_INVALID = 4 };

template <class T> struct Pk;
template <> struct Pk<float> { enum { P = 4 }; typedef float4 V; };
template <> struct Pk<double> { enum { P = 2 }; typedef double2 V; };

template <class T> struct PoseK { T R[9]; T t[3]; };

__device__ __forceinline__ void unpack3(const float4& a, const float4& b, const float4& c, float (&v)[12]) {
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
}
__device__ __forceinline__ void unpack3(const double2& a, const double2& b, const double2& c, double (&v)[6]) {
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
}

// group g = correspondences [P*g, P*g + P).  Entries past n read as zero.
template <class T>
__device__ __forceinline__ void load_group(const T* __restrict__ a, int64_t g, int64_t n, T (&v)[3 * Pk<T>::P]) {
  constexpr int P = Pk<T>::P;
  if ((g + 1) * P <= n) {
    const typename Pk<T>::V* q = reinterpret_cast<const typename Pk<T>::V*>(a) + 3 * g;
    typename Pk<T>::V v0 = q[0], v1 = q[1], v2 = q[2];
    unpack3(v0, v1, v2, v);
  } else {
#pragma unroll
    for (int i = 0; i < 3 * P; i++) { int64_t idx = g * (3 * P) + i; v[i] = idx < 3 * n ? a[idx] : T(0); }
  }
}
template <class T, class S>
__device__ __forceinline__ void load_scalars(const S* __restrict__ a, int64_t g, int64_t n, S (&v)[Pk<T>::P], S fill) {
  constexpr int P = Pk<T>::P;
#pragma unroll
  for (int i = 0; i < P; i++) { int64_t idx = g * P + i; v[i] = idx < n ? a[idx] : fill; }
}
// P inlier flags (short) of group g with one 8-byte (fp32, P = 4) or 4-byte (fp64, P = 2) load
__device__ __forceinline__ void load_mask_group(const short* __restrict__ m, int64_t g, int64_t n, short (&v)[4]) {
  if ((g + 1) * 4 <= n) {
    const uint2 u = *reinterpret_cast<const uint2*>(m + 4 * g);
    v[0] = (short)(u.x & 0xffffu); v[1] = (short)(u.x >> 16); v[2] = (short)(u.y & 0xffffu); v[3] = (short)(u.y >> 16);
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) { int64_t idx = g * 4 + i; v[i] = idx < n ? m[idx] : (short)0; }
  }
}
__device__ __forceinline__ void load_mask_group(const short* __restrict__ m, int64_t g, int64_t n, short (&v)[2]) {
  if ((g + 1) * 2 <= n) {
    const unsigned int u = *reinterpret_cast<const unsigned int*>(m + 2 * g);
    v[0] = (short)(u & 0xffffu); v[1] = (short)(u >> 16);
  } else {
#pragma unroll
    for (int i = 0; i < 2; i++) { int64_t idx = g * 2 + i; v[i] = idx < n ? m[idx] : (short)0; }
  }
}
__device__ __forceinline__ void load_weight_group(const float* __restrict__ w, int64_t g, int64_t n, float (&v)[4]) {
  if ((g + 1) * 4 <= n) { const float4 u = *reinterpret_cast<const float4*>(w + 4 * g); v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w; }
  else {
#pragma unroll
    for (int i = 0; i < 4; i++) { int64_t idx = g * 4 + i; v[i] = idx < n ? w[idx] : 0.f; }
  }
}
__device__ __forceinline__ void load_weight_group(const double* __restrict__ w, int64_t g, int64_t n, double (&v)[2]) {
  if ((g + 1) * 2 <= n) { const double2 u = *reinterpret_cast<const double2*>(w + 2 * g); v[0] = u.x; v[1] = u.y; }
  else {
#pragma unroll
    for (int i = 0; i < 2; i++) { int64_t idx = g * 2 + i; v[i] = idx < n ? w[idx] : 0.0; }
  }
}
__device__ __forceinline__ void load_mask_full(const short* __restrict__ m, int64_t g, short (&v)[4]) {
  const uint2 u = *reinterpret_cast<const uint2*>(m + 4 * g);
  v[0] = (short)(u.x & 0xffffu); v[1] = (short)(u.x >> 16); v[2] = (short)(u.y & 0xffffu); v[3] = (short)(u.y >> 16);
}
__device__ __forceinline__ void load_mask_full(const short* __restrict__ m, int64_t g, short (&v)[2]) {
  const unsigned int u = *reinterpret_cast<const unsigned int*>(m + 2 * g);
  v[0] = (short)(u & 0xffffu); v[1] = (short)(u >> 16);
}
__device__ __forceinline__ void load_weight_full(const float* __restrict__ w, int64_t g, float (&v)[4]) {
  const float4 u = *reinterpret_cast<const float4*>(w + 4 * g); v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
}
__device__ __forceinline__ void load_weight_full(const double* __restrict__ w, int64_t g, double (&v)[2]) {
  const double2 u = *reinterpret_cast<const double2*>(w + 2 * g); v[0] = u.x; v[1] = u.y;
}
template <class T> __device__ __forceinline__ bool all_nan(T x, T y, T z) { return x != x && y != y && z != z; }

// ---- two-stage reduction inside ONE launch.
// Stage 1 (every workgroup): wave64 reduce-scatter (below), one LDS hop across the waves, one LD-double partial record
// in HBM.  Stage 2 (the workgroup whose ticket is last): sums the G records IN ROW ORDER -- the result does not
// depend on which workgroup happens to be last, so it is bitwise reproducible -- expands it to the packed
// normal-equation record and publishes it to HBM (for a collective) and/or to pinned host memory followed by a
// sequence word the host spins on (no D2H copy kernel, no stream synchronise on the critical path).
// Hand-off protocol = the FENCE-FREE form of cdna_hip_programming.md Guideline 16 ("sc1 loads in place of the acquire", the valid-forms
// table of MI355X_MICROARCH.md, first row): every byte of a partial record is stored write-through (relaxed agent-scope atomic store =
// global_store ... sc1) by ONE wave, that wave drains vmcnt(0), the workgroup barriers, ONE lane adds to the arrival counter (relaxed,
// agent scope), and the workgroup whose add came last reads the records -- after a workgroup barrier -- with relaxed agent-scope
// atomic loads (= global_load ... sc1, L1-bypassing) and nothing else.  There is NO release / acquire fence: under the HIP / LLVM
// memory model alone this would be a data race; what makes it a hand-off is the gfx942 / gfx950 lowering of those accesses (sc1 write-
// through to the memory side, sc1 loads served past the per-CU L1, vmcnt covering write-through completion), measured in the guide.
// It saves the ~1.7 us a release fence and the ~1.7 us an acquire fence cost per launch (the whole kernel takes ~8 us).  The guard
// below keeps the file from being compiled for an architecture where that lowering has not been established.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "rpe_kernels.hip: the fence-free cross-workgroup hand-off (reduce_and_finish) is only established for gfx942 / gfx950"
#endif
// wave64 sum by DPP cross-lane moves (no LDS traffic): butterfly inside each row of 16 lanes (quad_perm, row_ror),
// then row_bcast:15 / row_bcast:31 fold the four rows; the total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v += dpp_move<0xb1, 0xf>(v);    // quad_perm:[1,0,3,2]
  v += dpp_move<0x4e, 0xf>(v);    // quad_perm:[2,3,0,1]
  v += dpp_move<0x124, 0xf>(v);   // row_ror:4
  v += dpp_move<0x128, 0xf>(v);   // row_ror:8
  v += dpp_move<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_move<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
  return v;
}


// ---- many values at once: a REDUCE-SCATTER across the wave instead of NACC independent butterflies.  At every step a lane
// keeps one half of its values and hands the other half to its partner, so the number of cross-lane operations halves each
// time: 32 values cost 31 exchange-and-add steps instead of 32 x 6 (124 VALU instructions instead of 576 for fp64).  The
// first two steps use gfx950's v_permlane32_swap / v_permlane16_swap, which exchange the halves (rows) of TWO registers in
// one instruction: after swap(a, b) the sum of the two results holds a's pair sums in the lower half (even rows) and b's in
// the upper half (odd rows).  The remaining steps pair lanes with DPP moves (row_ror:8, row_half_mirror, quad_perm) and a
// select on the lane bit that tells the partners apart.  The order of the additions is fixed, so results stay reproducible.
__device__ __forceinline__ double swap_add32(double x, double y) {   // lower 32 lanes end with x(l) + x(l+32), upper with y(l-32) + y(l)
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap_add16(double x, double y) {   // even rows end with x's row-pair sums, odd rows with y's
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
template <int CTRL> __device__ __forceinline__ double pair_add(double x, double y, bool upper) {   // lanes with upper = 0 keep x, the others y
  const double keep = upper ? y : x, give = upper ? x : y;
  return keep + dpp_move<CTRL, 0xf>(give);
}
// 32 values -> lane l holds the wave total of value (l >> 1)
__device__ __forceinline__ double wave_reduce_scatter32(const double (&v)[32], int lane) {
  double a[16], b[8], c[4], d[2];
#pragma unroll
  for (int j = 0; j < 16; j++) a[j] = swap_add32(v[j], v[j + 16]);
#pragma unroll
  for (int j = 0; j < 8; j++) b[j] = swap_add16(a[j], a[j + 8]);
#pragma unroll
  for (int j = 0; j < 4; j++) c[j] = pair_add<0x128>(b[j], b[j + 4], (lane & 8) != 0);    // row_ror:8         partner l ^ 8
#pragma unroll
  for (int j = 0; j < 2; j++) d[j] = pair_add<0x141>(c[j], c[j + 2], (lane & 4) != 0);    // row_half_mirror  partner l ^ 7
  double e = pair_add<0x1b>(d[0], d[1], (lane & 2) != 0);                                  // quad_perm:[3,2,1,0] partner l ^ 3
  e += dpp_move<0xb1, 0xf>(e);                                                             // quad_perm:[1,0,3,2] partner l ^ 1
  return e;
}
// 16 values -> lane l holds the wave total of value (l >> 2) & 15
__device__ __forceinline__ double wave_reduce_scatter16(const double (&v)[16], int lane) {
  double a[8], b[4], c[2];
#pragma unroll
  for (int j = 0; j < 8; j++) a[j] = swap_add32(v[j], v[j + 8]);
#pragma unroll
  for (int j = 0; j < 4; j++) b[j] = swap_add16(a[j], a[j + 4]);
#pragma unroll
  for (int j = 0; j < 2; j++) c[j] = pair_add<0x128>(b[j], b[j + 2], (lane & 8) != 0);
  double d = pair_add<0x141>(c[0], c[1], (lane & 4) != 0);
  d += dpp_move<0x1b, 0xf>(d);
  d += dpp_move<0xb1, 0xf>(d);
  return d;
}
// wave totals of acc[0 .. NACC) into out[0 .. NACC) (LDS row of this wave)
template <int NACC>
__device__ __forceinline__ void wave_reduce_to(const double (&acc)[NACC], double* __restrict__ out, int lane) {
  int done = 0;
  if constexpr (NACC >= 24) {          // a block of 32 (padded with zeros) -- 29 and 44 accumulators
    double v[32];
#pragma unroll
    for (int k = 0; k < 32; k++) v[k] = k < NACC ? acc[k] : 0.0;
    const double r = wave_reduce_scatter32(v, lane);
    if ((lane & 1) == 0 && (lane >> 1) < NACC) out[lane >> 1] = r;
    done = 32;
  } else if constexpr (NACC >= 12) {   // a block of 16 -- 17 accumulators
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = acc[k];
    const double r = wave_reduce_scatter16(v, lane);
    if ((lane & 3) == 0) out[(lane >> 2) & 15] = r;
    done = 16;
  }
  if constexpr (NACC > 32 && NACC - 32 > 4) {   // second block for the 44-value record: 12 more in a block of 16
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = 32 + k < NACC ? acc[32 + k < NACC ? 32 + k : 0] : 0.0;
    const double r = wave_reduce_scatter16(v, lane);
    if ((lane & 3) == 0 && 32 + ((lane >> 2) & 15) < NACC) out[32 + ((lane >> 2) & 15)] = r;
    done = 48;
  }
#pragma unroll
  for (int k = 0; k < NACC; k++) {     // the stragglers (1 of 17; everything for tiny records) one butterfly each
    if (k >= done) {
      const double t = wave_sum_to_lane63(acc[k]);
      if (lane == 63) out[k] = t;
    }
  }
}

struct Finish {
  double* partials;            // gridDim.x * LD doubles
  unsigned int* ticket;        // 9 counters, 32 uints apart, zero before the launch; rearmed by the last arrivers
  double* out_dev;             // LD doubles in HBM, or null
  double* out_host;            // LD doubles + 1 sequence word in pinned host memory, or null
  unsigned long long seq;      // value published after the record
  double* gn_pose;             // device-resident Gauss-Newton: pose in HBM (null = pose comes as a kernel argument)
  GnState* gn;                 // and its state
  const P2PDesc* p2p;          // multi-GPU peer-to-peer all-reduce of the record (null = single GPU / collective done elsewhere)
  unsigned long long p2p_step;
  int tail;                    // cross-workgroup tail: 0 = all records summed by the last workgroup, 1 = per-shard sums first, 2 = 0 with one load batch
  int rows;                    // > 0: collecting workgroups + host-side final sum (collect_and_send / the resident kernel): cap on the run length
};

// ---- all-reduce(sum) of the 32-double record across <= 8 GPUs, by the first wave of the LAST workgroup, without leaving the
// kernel: lane l owns half l of the record (two 32-bit halves per double); it stores {half, tag} as ONE 8-byte word into slot
// [parity][my rank][l] of every rank's mailbox (remote stores travel over xGMI), then polls slot [parity][r][l] of its OWN
// mailbox for every r until the tag shows up, and adds the records in rank order -- the same order on every rank, so all
// ranks publish bitwise the same sums.  Parity alternates per step: a fast peer's next record cannot overwrite one that is
// still being read.  Bounded wait (10 s of the 100 MHz clock): on a timeout *failed is set and the caller publishes an
// error marker instead of hanging the GPU.  `val`: lanes 0..31 hold the local record.  Returns the global record in lanes 0..31.
__device__ __forceinline__ double p2p_allreduce32(double val, const Finish& fin, int* failed) {
  const P2PDesc& D = *fin.p2p;
  const int lane = threadIdx.x & 63;
  const unsigned int tag = (unsigned int)(fin.p2p_step % 0xFFFFFFFFull) + 1u;   // never 0 (= an empty mailbox)
  const size_t parity = (size_t)(fin.p2p_step & 1ull);
  const double mine = __shfl(val, lane >> 1, 64);
  const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
  const unsigned int half = (lane & 1) ? (unsigned int)(bits >> 32) : (unsigned int)bits;
  const unsigned long long word = ((unsigned long long)tag << 32) | half;
  const size_t slot = (parity * kP2PMaxWorld + (size_t)D.rank) * kP2PWords + lane;
  for (int r = 0; r < D.world; r++) __hip_atomic_store(D.peer[r] + slot, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long* box = D.peer[D.rank] + parity * kP2PMaxWorld * kP2PWords;
  const unsigned long long t0 = wall_clock64();
  double sum = 0.0;
  int bad = 0;
  for (int r = 0; r < D.world; r++) {
    unsigned long long w;
    for (;;) {
      w = __hip_atomic_load(box + (size_t)r * kP2PWords + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((unsigned int)(w >> 32) == tag) break;
      if (wall_clock64() - t0 > 1000000000ull) { bad = 1; break; }   // 10 s of the 100 MHz constant clock
    }
    if (__any(bad)) { bad = 1; break; }
    const unsigned int lo = __shfl((unsigned int)w, (lane << 1) & 63, 64), hi = __shfl((unsigned int)w, ((lane << 1) + 1) & 63, 64);
    sum += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));   // meaningful in lanes 0..31
  }
  *failed = bad;
  return sum;
}

// ---- device-resident Gauss-Newton: solve H d = -g (LDL^T) and T <- exp(d) T by ONE lane of the last workgroup.
// Fully unrolled so that every matrix entry is a register (a rolled version over LDS arrays took ~10 us per call: one
// lane, ~500 dependent LDS round trips); the streaming body's occupancy is unaffected as long as the kernel stays within
// 256 VGPRs (one 512-thread workgroup per CU = 2 waves per SIMD).  Arithmetic follows rpe/linalg.hpp except for two latency savers
// (one reciprocal per pivot, one sincos of the half angle): last-bit differences, checked against the golden (1e-13).
__device__ __noinline__ bool gn_solve_update(const double* __restrict__ rec /* LDS */, double* __restrict__ pose /* LDS, 12, in/out */,
                                             double* step_out) {
  double A[6][6], Lm[6][6], D[6], Dinv[6], y[6], d[6];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
      for (int j = i; j < 6; j++) { A[i][j] = rec[k]; A[j][i] = rec[k]; k++; }
    }
  }
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double dj = A[j][j];
#pragma unroll
    for (int m = 0; m < j; m++) dj -= Lm[j][m] * Lm[j][m] * D[m];
    ok = ok && (dj > 1e-12 * A[j][j]) && (dj < 1e300);   // relative pivot floor, as rpe/linalg.hpp solve_normal_eq6
    D[j] = dj;
    const double inv = 1.0 / dj;   // ONE division per column (the host divides every entry; the results differ in the last bit at most)
    Dinv[j] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double sacc = A[i][j];
#pragma unroll
      for (int m = 0; m < j; m++) sacc -= Lm[i][m] * Lm[j][m] * D[m];
      Lm[i][j] = sacc * inv;
    }
  }
  if (!ok) return false;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double sacc = -rec[21 + i];
#pragma unroll
    for (int m = 0; m < i; m++) sacc -= Lm[i][m] * y[m];
    y[i] = sacc;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] *= Dinv[i];
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double sacc = y[i];
#pragma unroll
    for (int m = i + 1; m < 6; m++) sacc -= Lm[m][i] * d[m];
    d[i] = sacc;
  }
  double n2 = 0.0;
#pragma unroll
  for (int i = 0; i < 6; i++) { ok = ok && (d[i] == d[i]) && (d[i] < 1e300 && d[i] > -1e300); n2 += d[i] * d[i]; }
  if (!ok) return false;
  *step_out = sqrt(n2);
  // exp(d): rotation from the quaternion (cos(th/2), sin(th/2) w / th), V = I + c1 W + c2 W^2  (sophus/se3.hpp:321-342)
  const double wx = d[3], wy = d[4], wz = d[5];
  const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
  // ONE sincos of the half angle serves the quaternion and, through sin th = 2 s c and 1 - cos th = 2 s^2, the V matrix (a single lane
  // runs this: four separate fp64 sin / cos calls were a quarter of the solve's time)
  double imag, real, sh = 0.0, ch = 1.0;
  if (th < 1e-10) { imag = 0.5 - th2 / 48.0 + th2 * th2 / 3840.0; real = 1.0 - th2 / 8.0 + th2 * th2 / 384.0; }
  else { sincos(0.5 * th, &sh, &ch); imag = sh / th; real = ch; }
  double Rd[9], V[9];
  {
    const double qw = real, qx = imag * wx, qy = imag * wy, qz = imag * wz;
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    Rd[0] = 1 - (tyy + tzz); Rd[1] = txy - twz; Rd[2] = txz + twy;
    Rd[3] = txy + twz; Rd[4] = 1 - (txx + tzz); Rd[5] = tyz - twx;
    Rd[6] = txz - twy; Rd[7] = tyz + twx; Rd[8] = 1 - (txx + tyy);
  }
  const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  if (th < 1e-10) {
#pragma unroll
    for (int k = 0; k < 9; k++) V[k] = Rd[k];
  } else {
    const double c1 = (2.0 * sh * sh) / th2, c2 = (th - 2.0 * sh * ch) / (th2 * th);
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
      for (int jj = 0; jj < 3; jj++) {
        const double w2 = W[3 * i] * W[jj] + W[3 * i + 1] * W[3 + jj] + W[3 * i + 2] * W[6 + jj];
        V[3 * i + jj] = (i == jj ? 1.0 : 0.0) + c1 * W[3 * i + jj] + c2 * w2;
      }
    }
  }
  double P0[12], Pn[12];
#pragma unroll
  for (int k = 0; k < 12; k++) P0[k] = pose[k];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double td = V[3 * i] * d[0] + V[3 * i + 1] * d[1] + V[3 * i + 2] * d[2];
#pragma unroll
    for (int jj = 0; jj < 3; jj++) Pn[3 * i + jj] = Rd[3 * i] * P0[jj] + Rd[3 * i + 1] * P0[3 + jj] + Rd[3 * i + 2] * P0[6 + jj];
    Pn[9 + i] = Rd[3 * i] * P0[9] + Rd[3 * i + 1] * P0[10] + Rd[3 * i + 2] * P0[11] + td;
  }
#pragma unroll
  for (int k = 0; k < 12; k++) pose[k] = Pn[k];
  return true;
}

// fixed-order column sums of `count` partial records, rows first, first + step, ...: thread (j, rg) takes every RG-th of them,
// U independent sc1 loads in flight, then the RG row-group sums are added in row-group order -> tot[j] (valid for threadIdx.x < LD
// after the caller's barrier).  The order depends on (first, step, count) only, never on which workgroup runs it.
// test hook (rpe_debug_device_gn_update): the device-resident loop's solve + SE(3) update on a record and pose of the caller's, so that
// the LDL^T solve and the exponential map the last workgroup runs can be checked against the oracle / golden values in isolation
__global__ void gn_update_probe_kernel(const double* __restrict__ rec, double* __restrict__ pose, double* __restrict__ step_ok) {
  __shared__ double s_rec[32];
  __shared__ double s_pose[12];
  if (threadIdx.x < 32) s_rec[threadIdx.x] = rec[threadIdx.x];
  if (threadIdx.x < 12) s_pose[threadIdx.x] = pose[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    double step = 0.0;
    const bool ok = gn_solve_update(s_rec, s_pose, &step);
    step_ok[0] = step; step_ok[1] = ok ? 1.0 : 0.0;
    if (ok) for (int k = 0; k < 12; k++) pose[k] = s_pose[k];
  }
}
hipError_t launch_gn_update_probe(const double* d_rec, double* d_pose, double* d_step_ok, hipStream_t s) {
  hipLaunchKernelGGL(gn_update_probe_kernel, dim3(1), dim3(64), 0, s, d_rec, d_pose, d_step_ok);
  return hipGetLastError();
}

// the packed 32-entry record entry `i` from the totals (LDS): MODE 0 = the totals are the record, MODE 1 = the 17 structured
// point-to-point sums expanded to H upper triangle (21) | g (6) | cost | weight
template <int MODE> __device__ __forceinline__ double record_entry(const double* __restrict__ tot, int i) {
  if (MODE == 0) return tot[i];
  const double nn = tot[0], Sx = tot[1], Sy = tot[2], Sz = tot[3];
  const double xx = tot[4], xy = tot[5], xz = tot[6], yy = tot[7], yz = tot[8], zz = tot[9];
  switch (i) {
    case 0: case 6: case 11: case 28: return nn;     // (0,0) (1,1) (2,2) ; weight sum
    case 4: return Sz;    case 5: return -Sy;          // (0,4) (0,5)
    case 8: return -Sz;   case 10: return Sx;          // (1,3) (1,5)
    case 12: return Sy;   case 13: return -Sx;         // (2,3) (2,4)
    case 15: return yy + zz; case 16: return -xy; case 17: return -xz;   // row 3
    case 18: return xx + zz; case 19: return -yz;                         // row 4
    case 20: return xx + yy;                                              // row 5
    case 21: case 22: case 23: case 24: case 25: case 26: return tot[i - 11];   // g = (sum r, sum p x r)
    case 27: return tot[16];
    default: return 0.0;
  }
}
// one value to the host WITH the sequence number in ONE 16-byte SYSTEM-scope store (sc0 sc1: straight out over PCIe); the host waits
// until every pair carries the sequence value, so no ordering between the stores, no drain and no separate flag are needed.
// (A plain or nt 16-byte store to this memory was observed never to reach the host while the kernel stays resident.  There is no
// 16-byte atomic builtin, hence the instruction itself; s_nop 1: the data registers must not be reused before the store reads them.)
__device__ __forceinline__ void store_tagged_pair(double* __restrict__ out_host, int slot, double val, unsigned long long seq) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long bits = (unsigned long long)__double_as_longlong(val);
  u32x4 pr;
  pr.x = (unsigned int)bits; pr.y = (unsigned int)(bits >> 32); pr.z = (unsigned int)seq; pr.w = (unsigned int)(seq >> 32);
  const unsigned long long* dst = reinterpret_cast<const unsigned long long*>(out_host) + 2 * slot;
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(dst), "v"(pr) : "memory");
}
// 16-byte granules {value, tag} between workgroups of one launch (agent scope): written by ONE sc1 (write-through) store, read by ONE
// sc1 load -- the tag travels with the value, so neither a drain nor an arrival counter is needed (cdna_hip_programming.md Guideline 16,
// recipe R2, with 16-byte granules: observed untorn on gfx950).
typedef unsigned int granule_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_granule16(unsigned long long* __restrict__ g, double val, unsigned long long tag) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(val);
  granule_t pr;
  pr.x = (unsigned int)bits; pr.y = (unsigned int)(bits >> 32); pr.z = (unsigned int)tag; pr.w = (unsigned int)(tag >> 32);
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(g), "v"(pr) : "memory");
}
__device__ __forceinline__ granule_t load_granule16(const unsigned long long* __restrict__ g) {
  granule_t pr;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(pr) : "v"(g) : "memory");
  return pr;
}

template <int NACC, int LD, int BLK, int U>
__device__ __forceinline__ void sum_records(const double* __restrict__ partials, int first, int step, int count, double (*part)[LD],
                                            double* __restrict__ tot) {
  constexpr int RG = BLK / LD;
  const int j = threadIdx.x % LD, rg = threadIdx.x / LD;
  double s = 0.0;
  if (j < NACC) {
    for (int r0 = rg; r0 < count; r0 += RG * U) {
      double v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int r = r0 + u * RG;
        v[u] = r < count ? __hip_atomic_load(partials + (size_t)(first + r * step) * LD + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; u++) s += v[u];
    }
  }
  part[rg][j] = s;
  __syncthreads();
  if (threadIdx.x < LD) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < RG; k++) t += part[k][threadIdx.x];
    tot[threadIdx.x] = threadIdx.x < NACC ? t : 0.0;
  }
}

// ---- cross-workgroup stage for results the HOST consumes (single GPU): collecting workgroups + a host-side final sum.
// Workgroups are taken in runs of R = min(fin.rows, BLK / NACC); the first of a run collects: the others store their NACC sums as
// 16-byte granules {value, launch sequence number} (one sc1 store per lane; no drain, no arrival counter) and are done; every thread of
// the collecting workgroup polls ONE granule (sc1 load until the tag is this launch's), the rows are added in row order, and the run's
// NACC sums go to pinned host memory as tagged 16-byte pairs (slot 1 + run * NACC + j; slot 0 = a header pair from workgroup 0 that
// tells the host how many runs and sums to expect).  The host adds the runs in run order and expands the record (rpe_capi.hip
// wait_collect).  One hand-off hop of ~1 us replaces the arrival counters + the last workgroup's re-read of all G records + the
// drain before the flag (profiles/r02_tail_timeline.jsonl); the sums are a fixed function of (G, R) whichever workgroup finishes first.
// Placement-independent: only the ceil(G / R) collecting workgroups ever wait, and only for workgroups that never wait themselves.
// the collecting workgroup's read: thread (r, j) = (tid / NACC, tid % NACC) takes rows r, r + RGN, r + 2 RGN ... of the run (row 0 is the
// workgroup's own record, already in part[0]), up to CH granules in flight at once (buffer loads with the sc1 bit, aux 16, re-issued
// until every tag is this launch's), added in increasing row order into part[r][j].  Returns true if a granule never arrived (2 s).
template <int NACC, int BLK, int CH = 4>
__device__ __forceinline__ bool collect_rows(unsigned long long* __restrict__ gran, int G, int leader, int rows, unsigned long long tag,
                                             double (*part)[NACC]) {
  constexpr int RGN = BLK / NACC;
  const int j = threadIdx.x % NACC, r = threadIdx.x / NACC;
  bool lost = false;
  if (CH == 1) {   // runs of at most RGN rows (the resident kernel): one granule per thread, polled by itself
    if (r >= 1 && r < rows) {
      const unsigned long long* src = gran + 2 * ((size_t)(leader + r) * NACC + j);
      const unsigned long long t0 = wall_clock64();
      granule_t q;
      for (unsigned int spins = 1;; spins++) {
        q = load_granule16(src);
        if ((((unsigned long long)q.w << 32) | q.z) == tag) break;
        if ((spins & 63u) == 0 && wall_clock64() - t0 > 200000000ull) { lost = true; break; }   // 2 s: a workgroup never delivered
      }
      part[r][j] = __longlong_as_double((long long)(((unsigned long long)q.y << 32) | q.x));
    }
  } else if (r < RGN && r < rows) {
    double sum = r == 0 ? part[0][j] : 0.0;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(gran), 0, G * NACC * 16, 0x00020000);
    const unsigned long long t0 = wall_clock64();
    for (int k0 = r == 0 ? RGN : r; k0 < rows && !lost; k0 += CH * RGN) {
      granule_t q[CH];
      for (unsigned int spins = 1;; spins++) {
        bool pending = false;
        asm volatile("" ::: "memory");   // the loads below are re-issued every sweep (to the compiler they read memory nobody writes)
#pragma unroll
        for (int u = 0; u < CH; u++) {
          const int row = k0 + u * RGN;
          if (row < rows) q[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((leader + row) * NACC + j) * 16, 0, 16);
        }
#pragma unroll
        for (int u = 0; u < CH; u++) {
          const int row = k0 + u * RGN;
          if (row < rows && (((unsigned long long)q[u].w << 32) | q[u].z) != tag) pending = true;
        }
        if (!pending) break;
        if ((spins & 63u) == 0 && wall_clock64() - t0 > 200000000ull) { lost = true; break; }   // 2 s: a workgroup never delivered
      }
#pragma unroll
      for (int u = 0; u < CH; u++) {
        const int row = k0 + u * RGN;
        if (row < rows && !lost) sum += __longlong_as_double((long long)(((unsigned long long)q[u].y << 32) | q[u].x));
      }
    }
    part[r][j] = sum;
  }
  return lost;
}

// ---- cross-workgroup stage for results the HOST consumes (single GPU): collecting workgroups + a host-side final sum.
// Workgroups are taken in runs of R; the first of a run collects: the others store their NACC sums as 16-byte granules {value, launch
// sequence number} (one sc1 store per lane; no drain, no arrival counter) and are done; the collecting workgroup reads its run's
// granules (collect_rows), adds the rows in a fixed order and sends the run's NACC sums to pinned host memory as tagged 16-byte pairs
// (slot 1 + run * NACC + j; slot 0 = a header pair from workgroup 0 that tells the host how many runs of how many sums to expect).
// The host adds the runs in run order and expands the record (rpe_capi.hip wait_collect).  R = BLK / NACC rows (one granule per
// collecting thread) times 1..4, aiming at <= 8 runs; longer still if the runs would not fit in ~512 pairs.  One hand-off hop of ~1 us replaces the arrival counters + the last
// workgroup's re-read of all G records + the drain before the flag (profiles/r02_tail_timeline.jsonl); the sums are a fixed function
// of G whichever workgroup finishes first.  Placement-independent: only the collecting workgroups ever wait, and only for workgroups
// that never wait themselves.
template <int NACC, int MODE, int BLK>
__device__ __forceinline__ void collect_and_send(const double (*red)[NACC], const Finish& fin) {
  constexpr int NW = BLK / 64;
  constexpr int RGN = BLK / NACC;
  constexpr int kMaxRuns = 512 / NACC > 0 ? 512 / NACC : 1;
  __shared__ double c_part[RGN][NACC];
  unsigned long long* gran = reinterpret_cast<unsigned long long*>(fin.partials);   // [workgroup][NACC] granules of 2 words
  const int G = gridDim.x;
  // run length: aim at <= 8 runs with up to 4 granules per collecting thread (one batch of loads in flight: the hop costs the same as
  // with one), and never more pairs than ~512 whatever the grid
  int mult = (G + RGN * 8 - 1) / (RGN * 8);
  if (mult > 4) mult = 4;
  const int mult_cap = (G + RGN * kMaxRuns - 1) / (RGN * kMaxRuns);
  if (mult < mult_cap) mult = mult_cap;
  int R = RGN * mult;
  if (R > fin.rows) R = fin.rows;
  const int run = blockIdx.x / R, leader = run * R;
  if (threadIdx.x < NACC) {
    double own = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < NW; w++) own += red[w][threadIdx.x];
    if ((int)blockIdx.x != leader) store_granule16(gran + 2 * ((size_t)blockIdx.x * NACC + threadIdx.x), own, fin.seq);
    else c_part[0][threadIdx.x] = own;
  }
  RPE_STAMP(4);
  if ((int)blockIdx.x != leader) return;
  const int rows = min(R, G - leader);
  const bool lost = collect_rows<NACC, BLK>(gran, G, leader, rows, fin.seq, c_part);
  RPE_STAMP(7);
  if (__syncthreads_or(lost)) return;   // nothing published: the host reports the kernel as having finished without its result
  RPE_STAMP(8);
  if (threadIdx.x < NACC) {
    double t = 0.0;
    const int nr = rows < RGN ? rows : RGN;
    for (int k = 0; k < nr; k++) t += c_part[k][threadIdx.x];
    store_tagged_pair(fin.out_host, 1 + run * NACC + threadIdx.x, t, fin.seq);
  }
  if (blockIdx.x == 0 && threadIdx.x == 64) {   // header: runs | sums per run << 16 | record layout << 24
    const unsigned long long hdr = (unsigned long long)((G + R - 1) / R) | ((unsigned long long)NACC << 16) | ((unsigned long long)MODE << 24);
    store_tagged_pair(fin.out_host, 0, __longlong_as_double((long long)hdr), fin.seq);
  }
  RPE_STAMP(9);
}

template <int NACC, int LD, int MODE, int BLK>
__device__ __forceinline__ void reduce_and_finish(double (&acc)[NACC], const Finish& fin) {
  constexpr int NW = BLK / 64;
  constexpr int RG = BLK / LD;
  __shared__ double red[NW][NACC];
  __shared__ double part[RG][LD];
  __shared__ double tot[LD];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wave_reduce_to<NACC>(acc, red[wave], lane);
  RPE_STAMP(2);
  __syncthreads();
  RPE_STAMP(3);
  if (fin.rows > 0) { collect_and_send<NACC, MODE, BLK>(red, fin); return; }
  const int G = gridDim.x;
  if (G > 1) {
    // Hand-off without fences (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "Valid forms"): EVERY byte of
    // the partial records is stored write-through (relaxed agent-scope atomic store = global_store sc1) and loaded
    // L1-bypassing (relaxed agent-scope atomic load = global_load sc1); the storing wave drains vmcnt before the
    // workgroup barrier, one lane then adds to the ticket, and the workgroup whose add returned G-1 reads after a barrier.
    if (threadIdx.x < NACC) {
      double s = red[0][threadIdx.x];
#pragma unroll
      for (int w = 1; w < NW; w++) s += red[w][threadIdx.x];
      __hip_atomic_store(fin.partials + (size_t)blockIdx.x * LD + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RPE_STAMP(4);
    __syncthreads();
    RPE_STAMP(5);
    const int shard = blockIdx.x & 7;
    const int in_shard = (G - shard + 7) >> 3, shards = G < 8 ? G : 8;
    const int tailv = fin.tail & 3;
    if (tailv != 1) {
      if (threadIdx.x == 0) {
        // two-level arrival count: 8 shard counters (one 128-B line each) + a top counter.  A single counter costs
        // ~12 ns per arrival at the memory side (MI355X_MICROARCH.md "fanin"), i.e. 3+ us for a few hundred workgroups.
        int last = 0;
        unsigned int* sc = fin.ticket + 32 * shard;
        if (__hip_atomic_fetch_add(sc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)in_shard - 1) {
          __hip_atomic_store(sc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // rearm for the next launch
          unsigned int* top = fin.ticket + 32 * 8;
          if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)shards - 1) {
            __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = 1;
          }
        }
        s_last = last;
      }
      RPE_STAMP(6);
      __syncthreads();
      if (!s_last) return;
      RPE_STAMP(7);
      if (tailv == 2) sum_records<NACC, LD, BLK, 16>(fin.partials, 0, 1, G, part, tot);
      else sum_records<NACC, LD, BLK, 8>(fin.partials, 0, 1, G, part, tot);
      RPE_STAMP(8);
    } else {
      // hierarchical tail: the last arriver of each shard sums ITS shard's records (rows shard, shard + 8, ...) into one shard
      // record behind the G workgroup records, then arrives at the top counter; the last shard to arrive sums the <= 8 shard
      // records in shard order.  Same hand-off rules at both levels; the result is a fixed function of (G, records).
      if (threadIdx.x == 0) {
        unsigned int* sc = fin.ticket + 32 * shard;
        const int last = __hip_atomic_fetch_add(sc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)in_shard - 1;
        if (last) __hip_atomic_store(sc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
      }
      RPE_STAMP(6);
      __syncthreads();
      if (!s_last) return;
      RPE_STAMP(7);
      sum_records<NACC, LD, BLK, 4>(fin.partials, shard, 8, in_shard, part, tot);
      __syncthreads();
      if (threadIdx.x < LD) __hip_atomic_store(fin.partials + (size_t)(G + shard) * LD + threadIdx.x, tot[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      RPE_STAMP(8);
      if (threadIdx.x == 0) {
        unsigned int* top = fin.ticket + 32 * 8;
        const int last = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)shards - 1;
        if (last) __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
      }
      __syncthreads();
      if (!s_last) return;
      RPE_STAMP(12);
      sum_records<NACC, LD, BLK, 1>(fin.partials, G, 1, shards, part, tot);
      RPE_STAMP(13);
    }
  } else {
    if (threadIdx.x < LD) {
      double t = 0.0;
      if (threadIdx.x < NACC) {
#pragma unroll
        for (int w = 0; w < NW; w++) t += red[w][threadIdx.x];
      }
      tot[threadIdx.x] = t;
    }
  }
  __syncthreads();
  // publish
  double val = 0.0;
  if (threadIdx.x < LD) val = record_entry<MODE>(tot, threadIdx.x);
  if (LD == 32 && fin.p2p != nullptr && threadIdx.x < 64) {   // wave 0 (uniform branch): all 64 lanes take part in the exchange
    int failed = 0;
    val = p2p_allreduce32(val, fin, &failed);
    if (failed && threadIdx.x == 31) val = 1e300;   // error marker in the last (padding) entry of the record: the host checks it
  }
  if (threadIdx.x < LD) {
    if (fin.gn == nullptr) {
      if (fin.out_dev) fin.out_dev[threadIdx.x] = val;
      // pinned, coherent host memory: system-scope stores go straight out over PCIe (posted, ordered)
      if (fin.out_host) __hip_atomic_store(fin.out_host + threadIdx.x, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (fin.gn != nullptr) {
    // device-resident Gauss-Newton: this workgroup solves the 6x6 system, updates the pose in HBM and decides whether
    // the loop is finished; only a finished loop is published to the host (pose 12 | step | cost | iters | status | weight sum)
    __shared__ double gn_rec[LD];
    __shared__ double gn_pose_s[12];
    if (threadIdx.x < LD) gn_rec[threadIdx.x] = val;
    if (threadIdx.x < 12) gn_pose_s[threadIdx.x] = fin.gn_pose[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
      GnState* st = fin.gn;
      double step = 0.0;
      const bool delivered = !(LD == 32 && fin.p2p != nullptr && gn_rec[LD - 1] != 0.0);   // sharded loop: did every peer's record arrive?
      const bool ok = delivered && gn_solve_update(gn_rec, gn_pose_s, &step);
      const int iters = st->iters + 1;
      const int done = (!ok) || step < st->tol || iters >= st->max_iters;
      st->iters = iters; st->step = step; st->cost = gn_rec[27]; st->status = ok ? 0 : (delivered ? 1 : 2); st->done = done;
      if (ok) { for (int k = 0; k < 12; k++) fin.gn_pose[k] = gn_pose_s[k]; }
      if (done && fin.out_host) {
        for (int k = 0; k < 12; k++) __hip_atomic_store(fin.out_host + k, gn_pose_s[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin.out_host + 12, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin.out_host + 13, gn_rec[27], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin.out_host + 14, (double)iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin.out_host + 15, ok ? 0.0 : (delivered ? 1.0 : 2.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fin.out_host + 16, gn_rec[28], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // weight sum of the last round
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(fin.out_host + LD), fin.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  if (fin.out_host) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RPE_STAMP(9);
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(fin.out_host + LD), fin.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    RPE_STAMP(10);
  }
}

// ================================================================================================
// K1 / K2 / K3 : Gauss-Newton normal equations
// ================================================================================================
// p2p keeps 17 structured sums: w | w p (3) | w p p^T (6) | w r (3) | w p x r (3) | w r^2
// The pose stays fp64 and p = R x + t, r = p - Xc are formed in fp64: the subtraction cancels ~3 digits
// (|p| ~ 10 m, |r| ~ 5 cm), so doing it in fp32 would dominate the error budget.  p and r are then rounded
// to the compute type C for the products (fp32 for fp32 arrays), and the sums are widened to fp64 per group.
template <class C>
__device__ __forceinline__ void transform(const PoseK<double>& T, C x, C y, C z, double& px, double& py, double& pz) {
  const double xd = x, yd = y, zd = z;
  px = fma(T.R[0], xd, fma(T.R[1], yd, fma(T.R[2], zd, T.t[0])));
  py = fma(T.R[3], xd, fma(T.R[4], yd, fma(T.R[5], zd, T.t[1])));
  pz = fma(T.R[6], xd, fma(T.R[7], yd, fma(T.R[8], zd, T.t[2])));
}
template <class C>
__device__ __forceinline__ void p2p_point(const PoseK<double>& T, C x, C y, C z, C cx, C cy, C cz, C w, C (&s)[17]) {
  double pxd, pyd, pzd;
  transform<C>(T, x, y, z, pxd, pyd, pzd);
  const C px = (C)pxd, py = (C)pyd, pz = (C)pzd;
  const C rx = (C)(pxd - (double)cx), ry = (C)(pyd - (double)cy), rz = (C)(pzd - (double)cz);
  const C wpx = w * px, wpy = w * py, wpz = w * pz;
  const C wrx = w * rx, wry = w * ry, wrz = w * rz;
  s[0] += w;
  s[1] += wpx; s[2] += wpy; s[3] += wpz;
  s[4] = fma(wpx, px, s[4]); s[5] = fma(wpx, py, s[5]); s[6] = fma(wpx, pz, s[6]);
  s[7] = fma(wpy, py, s[7]); s[8] = fma(wpy, pz, s[8]); s[9] = fma(wpz, pz, s[9]);
  s[10] += wrx; s[11] += wry; s[12] += wrz;
  s[13] += py * wrz - pz * wry;
  s[14] += pz * wrx - px * wrz;
  s[15] += px * wry - py * wrx;
  s[16] = fma(wrx, rx, fma(wry, ry, fma(wrz, rz, s[16])));
}
// general packed record: H upper triangle (21) | g (6) | w r^2 | w
template <class C> __device__ __forceinline__ void add_row(const C (&J)[6], C r, C w, C (&s)[29]) {
  int k = 0;
#pragma unroll
  for (int a = 0; a < 6; a++) {
    const C wa = w * J[a];
#pragma unroll
    for (int b = a; b < 6; b++) { s[k] = fma(wa, J[b], s[k]); k++; }
    s[21 + a] = fma(wa, r, s[21 + a]);
  }
  s[27] = fma(w * r, r, s[27]);
}
template <class C>
__device__ __forceinline__ void p2plane_point(const PoseK<double>& T, C x, C y, C z, C cx, C cy, C cz, C nx, C ny, C nz, C w, C (&s)[29]) {
  double pxd, pyd, pzd;
  transform<C>(T, x, y, z, pxd, pyd, pzd);
  const C px = (C)pxd, py = (C)pyd, pz = (C)pzd;
  const C r = (C)((double)nx * (pxd - (double)cx) + (double)ny * (pyd - (double)cy) + (double)nz * (pzd - (double)cz));
  const C J[6] = {nx, ny, nz, py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx};  // [n ; p x n]
  add_row(J, r, w, s);
  s[28] += w;
}
// Two correspondences at once: the 35 products of a Jacobian row's outer product as 2-vectors (packed fp32 instructions for fp32
// arrays), each lane of the pair keeping its own partial sums; the pair's sums are added at the end of the group.
template <class V> __device__ __forceinline__ void add_row2(const V (&J)[6], V r, V w, V (&s)[29]) {
  int k = 0;
#pragma unroll
  for (int a = 0; a < 6; a++) {
    const V wa = w * J[a];
#pragma unroll
    for (int b = a; b < 6; b++) { s[k] = __builtin_elementwise_fma(wa, J[b], s[k]); k++; }
    s[21 + a] = __builtin_elementwise_fma(wa, r, s[21 + a]);
  }
  s[27] = __builtin_elementwise_fma(w * r, r, s[27]);
}
template <class C>
__device__ __forceinline__ void p2plane_pair(const PoseK<double>& T, const C (&x)[2], const C (&y)[2], const C (&z)[2], const C (&cx)[2], const C (&cy)[2],
                                             const C (&cz)[2], const C (&nx)[2], const C (&ny)[2], const C (&nz)[2], const C (&w)[2],
                                             C __attribute__((ext_vector_type(2))) (&s)[29]) {
  typedef C V __attribute__((ext_vector_type(2)));
  C px[2], py[2], pz[2], r[2];
#pragma unroll
  for (int e = 0; e < 2; e++) {   // the fp64 part stays per point: transform and the (cancelling) residual
    double pxd, pyd, pzd;
    transform<C>(T, x[e], y[e], z[e], pxd, pyd, pzd);
    px[e] = (C)pxd; py[e] = (C)pyd; pz[e] = (C)pzd;
    r[e] = (C)((double)nx[e] * (pxd - (double)cx[e]) + (double)ny[e] * (pyd - (double)cy[e]) + (double)nz[e] * (pzd - (double)cz[e]));
  }
  const V PX = {px[0], px[1]}, PY = {py[0], py[1]}, PZ = {pz[0], pz[1]}, NX = {nx[0], nx[1]}, NY = {ny[0], ny[1]}, NZ = {nz[0], nz[1]};
  const V J[6] = {NX, NY, NZ, PY * NZ - PZ * NY, PZ * NX - PX * NZ, PX * NY - PY * NX};  // [n ; p x n]
  const V R = {r[0], r[1]}, W = {w[0], w[1]};
  add_row2<V>(J, R, W, s);
  s[28] += W;
}
// 1 / sqrt(x) in fp64 without the ~45-instruction IEEE sqrt + divide sequences: the fp32 hardware estimate (v_rsq_f32, 1e-7)
// refined by two Newton steps y <- y (3/2 - x/2 y^2), each squaring the error: ~1 ulp of fp64 in ~10 instructions.  x is a
// squared point norm in metres^2 (fits fp32 comfortably).
__device__ __forceinline__ double rsqrt64(double x) {
  double y = (double)rsqrtf((float)x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
}

template <class C>
__device__ __forceinline__ void bearing_point(const PoseK<double>& T, C x, C y, C z, C bx, C by, C bz, C w, C (&s)[29]) {
  double pxd, pyd, pzd;
  transform<C>(T, x, y, z, pxd, pyd, pzd);
  const double invd = rsqrt64(pxd * pxd + pyd * pyd + pzd * pzd);
  const double hxd = pxd * invd, hyd = pyd * invd, hzd = pzd * invd;
  // sine residual p^ x bv: near the optimum p^ ~ bv, so this too is a cancelling difference -> fp64
  const C r[3] = {(C)(hyd * (double)bz - hzd * (double)by), (C)(hzd * (double)bx - hxd * (double)bz), (C)(hxd * (double)by - hyd * (double)bx)};
  const C px = (C)pxd, py = (C)pyd, pz = (C)pzd, inv = (C)invd;
  const C hx = (C)hxd, hy = (C)hyd, hz = (C)hzd;
  // A = -[bv]x (I - h h^T) * inv ; row u of A = -(e_u^T [bv]x) (I - h h^T) inv
  const C Bx[3][3] = {{C(0), -bz, by}, {bz, C(0), -bx}, {-by, bx, C(0)}};
  const C h[3] = {hx, hy, hz};
#pragma unroll
  for (int u = 0; u < 3; u++) {
    const C bh = Bx[u][0] * h[0] + Bx[u][1] * h[1] + Bx[u][2] * h[2];
    const C a0 = -(Bx[u][0] - bh * h[0]) * inv, a1 = -(Bx[u][1] - bh * h[1]) * inv, a2 = -(Bx[u][2] - bh * h[2]) * inv;
    // J = a^T [I | -[p]x] : translation part a, rotation part (p x a)
    const C J[6] = {a0, a1, a2, py * a2 - pz * a1, pz * a0 - px * a2, px * a1 - py * a0};
    add_row(J, r[u], w, s);
  }
  s[28] += w;
}

// The main loop runs over FULL groups only and is branch-free (mask / weight presence are template flags), so the
// compiler issues all 16-byte loads of an iteration up front behind one wait; the <= P-1 leftover correspondences
// are handled once, by thread 0 of workgroup 0, through the bounds-checked loaders.
template <class T, int KIND, bool MASK, bool WEIGHT, int NACC>
__device__ __forceinline__ void normal_eq_group(const PoseK<double>& pose, const T (&vw)[3 * Pk<T>::P], const T (&vb)[3 * Pk<T>::P],
                                                const T (&vc)[3 * Pk<T>::P], const short (&m)[Pk<T>::P], const T (&wv)[Pk<T>::P],
                                                int npresent, double (&acc)[NACC]) {
  constexpr int P = Pk<T>::P;
  if constexpr (KIND == KIND_P2PLANE) {   // pairs of correspondences (the accumulation is 35 of the ~50 operations per point)
    typedef T V __attribute__((ext_vector_type(2)));
    V s2[29];
#pragma unroll
    for (int k = 0; k < 29; k++) s2[k] = V{T(0), T(0)};
#pragma unroll
    for (int j = 0; j < P / 2; j++) {
      T x[2], y[2], z[2], bx[2], by[2], bz[2], nx[2], ny[2], nz[2], wi[2];
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int i = 2 * j + e;
        x[e] = vw[3 * i]; y[e] = vw[3 * i + 1]; z[e] = vw[3 * i + 2];
        bx[e] = vb[3 * i]; by[e] = vb[3 * i + 1]; bz[e] = vb[3 * i + 2];
        T w = WEIGHT ? wv[i] : T(1);
        if (MASK) w = m[i] == 1 ? w : T(0);
        w = (i < npresent && !all_nan(bx[e], by[e], bz[e])) ? w : T(0);
        const bool off = w == T(0);
        // keeps NaN / inf of skipped columns out of the sums (selects, not branches)
        x[e] = off ? T(0) : x[e]; y[e] = off ? T(0) : y[e]; z[e] = off ? T(0) : z[e];
        bx[e] = off ? T(0) : bx[e]; by[e] = off ? T(0) : by[e]; bz[e] = off ? T(1) : bz[e];
        nx[e] = off ? T(0) : vc[3 * i]; ny[e] = off ? T(0) : vc[3 * i + 1]; nz[e] = off ? T(0) : vc[3 * i + 2];
        wi[e] = w;
      }
      p2plane_pair<T>(pose, x, y, z, bx, by, bz, nx, ny, nz, wi, s2);
    }
#pragma unroll
    for (int k = 0; k < 29; k++) acc[k] += (double)(s2[k].x + s2[k].y);
    return;
  }
  T s[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) s[k] = T(0);
#pragma unroll
  for (int i = 0; i < P; i++) {
    T x = vw[3 * i], y = vw[3 * i + 1], z = vw[3 * i + 2];
    T bx = vb[3 * i], by = vb[3 * i + 1], bz = vb[3 * i + 2];
    T wi = WEIGHT ? wv[i] : T(1);
    if (MASK) wi = m[i] == 1 ? wi : T(0);
    wi = (i < npresent && !all_nan(bx, by, bz)) ? wi : T(0);
    const bool off = wi == T(0);
    // keeps NaN / inf of skipped columns out of the sums (selects, not branches)
    x = off ? T(0) : x; y = off ? T(0) : y; bx = off ? T(0) : bx; by = off ? T(0) : by; bz = off ? T(1) : bz;
    if (KIND == KIND_P2P) {
      z = off ? T(0) : z;
      p2p_point<T>(pose, x, y, z, bx, by, bz, wi, reinterpret_cast<T(&)[17]>(s));
    } else if (KIND == KIND_P2PLANE) {
      z = off ? T(0) : z;
      const T nx = off ? T(0) : vc[3 * i], ny = off ? T(0) : vc[3 * i + 1], nz = off ? T(0) : vc[3 * i + 2];
      p2plane_point<T>(pose, x, y, z, bx, by, bz, nx, ny, nz, wi, reinterpret_cast<T(&)[29]>(s));
    } else {
      z = off ? T(1) : z;  // p != 0 so that the normalisation stays finite
      bearing_point<T>(pose, x, y, z, bx, by, bz, wi, reinterpret_cast<T(&)[29]>(s));
    }
  }
#pragma unroll
  for (int k = 0; k < NACC; k++) acc[k] += (double)s[k];
}

template <class T, int KIND, int BLK, bool MASK, bool WEIGHT>
__global__ __launch_bounds__(BLK) void normal_eq_kernel(const T* __restrict__ xw, const T* __restrict__ b, const T* __restrict__ c,
                                                        const short* __restrict__ mask, const T* __restrict__ weight, int64_t n,
                                                        PoseK<double> pose, Finish fin) {
  constexpr int P = Pk<T>::P;
  constexpr int NACC = KIND == KIND_P2P ? 17 : 29;
  typedef typename Pk<T>::V V;
  if (fin.gn != nullptr) {  // device-resident Gauss-Newton: finished loops cost an empty launch; the pose lives in HBM
    if (fin.gn->done) return;
#pragma unroll
    for (int k = 0; k < 9; k++) pose.R[k] = fin.gn_pose[k];
#pragma unroll
    for (int k = 0; k < 3; k++) pose.t[k] = fin.gn_pose[9 + k];
  }
  RPE_STAMP(0);
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) acc[k] = 0.0;
  const int64_t full = n / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const V* __restrict__ xw4 = reinterpret_cast<const V*>(xw);
  const V* __restrict__ b4 = reinterpret_cast<const V*>(b);
  const V* __restrict__ c4 = reinterpret_cast<const V*>(c);
  // software pipeline: the loads of the NEXT group are in flight while the current one is reduced, so a CU's waves do not
  // all alternate between "everyone waits on HBM" and "everyone computes" (measured +x% at 20 M correspondences)
  int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
  V a0, a1, a2, b0, b1, b2, c0, c1, c2;
  short m[P];
  T wv[P];
  if (g < full) {
    a0 = xw4[3 * g]; a1 = xw4[3 * g + 1]; a2 = xw4[3 * g + 2];
    b0 = b4[3 * g]; b1 = b4[3 * g + 1]; b2 = b4[3 * g + 2];
    if (KIND == KIND_P2PLANE) { c0 = c4[3 * g]; c1 = c4[3 * g + 1]; c2 = c4[3 * g + 2]; }
    if (MASK) load_mask_full(mask, g, m);
    if (WEIGHT) load_weight_full(weight, g, wv);
  }
#if defined(RPE_STAMPS) && RPE_STAMPS >= 2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // diagnostic build 2: when have the first loads landed?
  RPE_STAMP(11);
#endif
  if (stride >= full && !(fin.tail & 8)) {   // frame-sized problems: one group per thread (reduce_grid), nothing to pipeline -- straight-line body
    if (g < full) {
      T vw[3 * P], vb[3 * P], vc[3 * P];
      unpack3(a0, a1, a2, vw);
      unpack3(b0, b1, b2, vb);
      if (KIND == KIND_P2PLANE) unpack3(c0, c1, c2, vc);
      normal_eq_group<T, KIND, MASK, WEIGHT, NACC>(pose, vw, vb, vc, m, wv, P, acc);
    }
    g = full;
  }
  while (g < full) {
    const int64_t gn = g + stride;
    const int64_t gl = gn < full ? gn : g;  // clamp: the last iteration re-reads its own (cached) group instead of branching
    const V na0 = xw4[3 * gl], na1 = xw4[3 * gl + 1], na2 = xw4[3 * gl + 2];
    const V nb0 = b4[3 * gl], nb1 = b4[3 * gl + 1], nb2 = b4[3 * gl + 2];
    V nc0, nc1, nc2;
    if (KIND == KIND_P2PLANE) { nc0 = c4[3 * gl]; nc1 = c4[3 * gl + 1]; nc2 = c4[3 * gl + 2]; }
    short nm[P];
    T nwv[P];
    if (MASK) load_mask_full(mask, gl, nm);
    if (WEIGHT) load_weight_full(weight, gl, nwv);
    T vw[3 * P], vb[3 * P], vc[3 * P];
    unpack3(a0, a1, a2, vw);
    unpack3(b0, b1, b2, vb);
    if (KIND == KIND_P2PLANE) unpack3(c0, c1, c2, vc);
    normal_eq_group<T, KIND, MASK, WEIGHT, NACC>(pose, vw, vb, vc, m, wv, P, acc);
    a0 = na0; a1 = na1; a2 = na2; b0 = nb0; b1 = nb1; b2 = nb2;
    if (KIND == KIND_P2PLANE) { c0 = nc0; c1 = nc1; c2 = nc2; }
#pragma unroll
    for (int i = 0; i < P; i++) { if (MASK) m[i] = nm[i]; if (WEIGHT) wv[i] = nwv[i]; }
    g = gn;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && full * P < n) {  // leftover correspondences
    T vw[3 * P], vb[3 * P], vc[3 * P];
    short m[P];
    T wv[P];
    load_group<T>(xw, full, n, vw);
    load_group<T>(b, full, n, vb);
    if (KIND == KIND_P2PLANE) load_group<T>(c, full, n, vc);
    if (MASK) load_scalars<T, short>(mask, full, n, m, (short)0);
    if (WEIGHT) load_scalars<T, T>(weight, full, n, wv, T(0));
    normal_eq_group<T, KIND, MASK, WEIGHT, NACC>(pose, vw, vb, vc, m, wv, (int)(n - full * P), acc);
  }
  RPE_STAMP(1);
  reduce_and_finish<NACC, kNeLd, KIND == KIND_P2P ? 1 : 0, BLK>(acc, fin);
}

// ================================================================================================
// K1 / K2 / K3, RESIDENT form: the host-driven Gauss-Newton loop in ONE launch.
// The north-star loop keeps the 6x6 solve and the SE(3) exp-map on the host, so every iteration needs a host round trip; with one
// launch per iteration that round trip also pays a kernel launch, the dispatch ramp of the grid (1.3 - 2.3 us for 150 workgroups,
// profiles/r02_tail_timeline.jsonl) and a re-read of the arrays.  Here the grid stays resident between iterations: every workgroup
// waits for the next pose in a control block that lives in fine-grained DEVICE memory and that the host writes through the PCIe BAR
// (MI355X: 1.9 us host -> 256 polling workgroups -> host, scripts/ubench/hostmailbox.hip; polling pinned HOST memory from 150
// workgroups costs 13 us), evaluates its slice, and the last workgroup publishes the record exactly as normal_eq_kernel does.
// Frame-sized problems (one group per thread) keep their correspondences IN REGISTERS across the iterations -- the arrays are read
// from memory once per refinement, not once per iteration.
// Control block: 16 words of 8 bytes = two 64-byte halves, each carrying its own copy of the tag so that no ordering between the
// host's stores to the two halves is assumed:   [0] tag | [1..7] pose[0..6]   ||   [8..12] pose[7..11] | [13,14] - | [15] tag.
// The host writes the pose, then both tags (= first_tag + iteration; bit 63 set = stop).  Co-residency: the grid has at most one
// workgroup per CU (reduce_grid, max_blocks <= 256) and no workgroup waits for another one -- only for the host, and only for a
// bounded time (~2 s of the 100 MHz clock, then the kernel exits without publishing and the host reports an error).
// ================================================================================================
constexpr unsigned long long kResidentStop = 1ull << 63;
// one group of P correspondences through the bounds-checked loaders when it is the ragged last one (g == full), plain 16-byte loads otherwise
template <class T, int KIND, bool MASK, bool WEIGHT>
__device__ __forceinline__ void load_any_group(const T* __restrict__ xw, const T* __restrict__ b, const T* __restrict__ c, const short* __restrict__ mask,
                                               const T* __restrict__ weight, int64_t g, int64_t full, int64_t n, T (&vw)[3 * Pk<T>::P],
                                               T (&vb)[3 * Pk<T>::P], T (&vc)[3 * Pk<T>::P], short (&m)[Pk<T>::P], T (&wv)[Pk<T>::P]) {
  typedef typename Pk<T>::V V;
  if (g < full) {
    const V* xw4 = reinterpret_cast<const V*>(xw);
    const V* b4 = reinterpret_cast<const V*>(b);
    const V x0 = xw4[3 * g], x1 = xw4[3 * g + 1], x2 = xw4[3 * g + 2];
    const V y0 = b4[3 * g], y1 = b4[3 * g + 1], y2 = b4[3 * g + 2];
    unpack3(x0, x1, x2, vw);
    unpack3(y0, y1, y2, vb);
    if (KIND == KIND_P2PLANE) { const V* c4 = reinterpret_cast<const V*>(c); const V z0 = c4[3 * g], z1 = c4[3 * g + 1], z2 = c4[3 * g + 2]; unpack3(z0, z1, z2, vc); }
    if (MASK) load_mask_full(mask, g, m);
    if (WEIGHT) load_weight_full(weight, g, wv);
  } else {
    load_group<T>(xw, g, n, vw);
    load_group<T>(b, g, n, vb);
    if (KIND == KIND_P2PLANE) load_group<T>(c, g, n, vc);
    if (MASK) load_scalars<T, short>(mask, g, n, m, (short)0);
    if (WEIGHT) load_scalars<T, T>(weight, g, n, wv, T(0));
  }
}

// ---- the two halves of a RESIDENT iteration that do not depend on what is being summed (shared by the normal-equation and the ICP
// resident kernels).
// Wait for pose number `want` in the control block (16 words in fine-grained device memory that the host writes through the PCIe BAR:
// word 0 = tag, words 1..12 = pose, word 15 = tag again, so the two 64-byte halves may arrive in any order).  The first 16 lanes of
// wave 0 read one 8-byte word each until both tags match.  Returns 1 = go (pose in s_pose), 2 = stop requested, 3 = the host went
// away (2 s); the value is uniform over the workgroup.
constexpr int kAutoMaxRunSums = 1024;   // run records x sums an autonomous iteration reads per workgroup (resident_auto_stage)
template <int BLK>
__device__ __forceinline__ int resident_wait_pose(const unsigned long long* __restrict__ ctl, unsigned long long want, double* __restrict__ s_pose,
                                                  int* __restrict__ s_go) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    int go = 0;
    unsigned long long w = 0;
    for (;;) {
      if (lane < 16) w = __hip_atomic_load(ctl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned long long ta = __shfl(w, 0, 64), tb = __shfl(w, 15, 64);
      if (ta == tb && (ta & ~kResidentStop) == want) { go = (ta & kResidentStop) ? 2 : 1; break; }
      if (wall_clock64() - t0 > 200000000ull) { go = 3; break; }   // the host went away: give up (2 s)
      __builtin_amdgcn_s_sleep(2);
    }
    if (lane >= 1 && lane <= 12) s_pose[lane - 1] = __longlong_as_double((long long)w);
    if (lane == 0) *s_go = go;
  }
  __syncthreads();
  return *s_go;
}
// Cross-workgroup stage of one resident iteration: COLLECTING workgroups + the host.  Workgroups are taken in runs of R = fin.rows; the
// first of a run collects: the others store their NACC sums as 16-byte granules {value, iteration tag} (one sc1 store per lane, no
// drain, no arrival counter) and go back to waiting for the next pose; every thread of the collecting workgroup polls its granule(s)
// (collect_rows: sc1 loads until the tag is this iteration's), the rows are added in a fixed order, and the run's NACC sums go to the
// host as tagged 16-byte pairs.  The host thread that owns the 6x6 solve adds the ceil(G / R) run records in run order.  So one
// hand-off hop on the GPU (about 1 us: a collecting wave reads a few hundred bytes, MI355X_MICROARCH.md "handoff-1to1"), a few hundred
// bytes over PCIe, and sums that are a fixed function of (G, R) whichever workgroup finishes first.  R = 1: every workgroup sends its
// own record (tiny problems).  A workgroup overwrites its granules only in the next iteration, which the host starts after it has
// received every run record, i.e. after the granules have been read.  Returns false if a granule never arrived (the kernel ends without
// publishing; the host reports that).
template <int NACC, int BLK>
__device__ __forceinline__ bool resident_cross_stage(const double (&acc)[NACC], const Finish& fin, unsigned long long tag, unsigned long long seq,
                                                     bool stamp_it) {
  constexpr int NW = BLK / 64;
  constexpr int RGN = BLK / NACC;                       // rows a collecting workgroup takes with one granule per thread
  __shared__ double g_red[NW][NACC];
  __shared__ double g_part[RGN][NACC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long* gran = reinterpret_cast<unsigned long long*>(fin.partials);   // [workgroup][NACC] granules of 2 words
  const int R = fin.rows, run = blockIdx.x / R, leader = run * R;
  wave_reduce_to<NACC>(acc, g_red[wave], lane);
  __syncthreads();
  if (threadIdx.x < NACC) {
    double own = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) own += g_red[w][threadIdx.x];
    if ((int)blockIdx.x != leader) store_granule16(gran + 2 * ((size_t)blockIdx.x * NACC + threadIdx.x), own, tag);
    else g_part[0][threadIdx.x] = own;
  }
#ifdef RPE_STAMPS
  if (stamp_it) RPE_STAMP(2);
#endif
  bool ok = true;
  if ((int)blockIdx.x == leader) {
    const int rows = min(R, (int)gridDim.x - leader);
    const bool lost = collect_rows<NACC, BLK>(gran, (int)gridDim.x, leader, rows, tag, g_part);
#ifdef RPE_STAMPS
    if (stamp_it) RPE_STAMP(3);
#endif
    if (__syncthreads_or(lost)) ok = false;
#ifdef RPE_STAMPS
    if (stamp_it) RPE_STAMP(4);
#endif
    if (ok && threadIdx.x < NACC) {
      double t = 0.0;
      const int nr = rows < RGN ? rows : RGN;
      for (int k = 0; k < nr; k++) t += g_part[k][threadIdx.x];
      store_tagged_pair(fin.out_host, run * NACC + threadIdx.x, t, seq);
    }
#ifdef RPE_STAMPS
    if (stamp_it) RPE_STAMP(5);
#endif
  }
  __syncthreads();   // g_red / g_part are reused by the next iteration
  return ok;
}

// Cross-workgroup stage of one AUTONOMOUS resident iteration (rpe_gn_refine_device: solve and exp-map on the GPU, no host in the loop).
// First hop as above -- runs of R workgroups, the first of a run collects its rows' granules -- but the run's NACC sums go to a RUN
// RECORD in device memory (granules again: {value, iteration tag}, one sc1 store per lane) instead of to the host.  Second hop: EVERY
// workgroup reads all ceil(G / R) run records, adds them in run order (the order the host uses: bitwise the host-driven loop's
// record), expands the record, and its first lane solves the 6x6 system and applies the update to the workgroup's own copy of the
// pose in LDS.  All workgroups compute the same bits, so they agree on the next pose and on when to stop without another hop.
// Run records are double-buffered by iteration parity: a collecting workgroup can publish iteration i + 1 while a late workgroup of
// another run still reads iteration i; it cannot reach i + 2 before that workgroup has delivered its granules of i + 1, i.e. after
// it has finished reading i.  Granules need no second buffer: a workgroup writes iteration i + 1's after it has read run records
// that its collector published after reading iteration i's.  Returns 0 = next iteration, 1 = finished (workgroup 0 published pose |
// step | cost | iterations | status | weight sum to the host), 2 = a granule never arrived (2 s).
template <int NACC, int BLK>
__device__ __forceinline__ int resident_auto_stage(const double (&acc)[NACC], const Finish& fin, unsigned long long tag, int it, int max_iters,
                                                   double tol, double* __restrict__ s_pose) {
  constexpr int NW = BLK / 64;
  constexpr int RGN = BLK / NACC;
  constexpr int MODE = NACC == 17 ? 1 : 0;
  __shared__ double a_red[NW][NACC];
  __shared__ double a_part[RGN][NACC];
  __shared__ double a_runs[kAutoMaxRunSums];
  __shared__ double a_tot[32], a_rec[32];
  __shared__ double a_step;
  __shared__ int a_ok;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G = (int)gridDim.x, R = fin.rows, run = blockIdx.x / R, leader = run * R, runs = (G + R - 1) / R;
  unsigned long long* gran = reinterpret_cast<unsigned long long*>(fin.partials);          // [workgroup][NACC] granules of 2 words
  unsigned long long* rrec = gran + 2 * ((size_t)G * NACC + (size_t)(it & 1) * runs * NACC);   // [parity][run][NACC]
  wave_reduce_to<NACC>(acc, a_red[wave], lane);
  __syncthreads();
  if (threadIdx.x < NACC) {
    double own = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) own += a_red[w][threadIdx.x];
    if ((int)blockIdx.x != leader) store_granule16(gran + 2 * ((size_t)blockIdx.x * NACC + threadIdx.x), own, tag);
    else a_part[0][threadIdx.x] = own;
  }
  bool lost = runs * NACC > kAutoMaxRunSums;   // a geometry the launcher never chooses: reported like a lost granule, not overrun
  if (!lost && (int)blockIdx.x == leader) {
    const int rows = min(R, G - leader);
    lost = __syncthreads_or(collect_rows<NACC, BLK>(gran, G, leader, rows, tag, a_part));
    if (!lost && threadIdx.x < NACC) {
      double t = 0.0;
      const int nr = rows < RGN ? rows : RGN;
      for (int k = 0; k < nr; k++) t += a_part[k][threadIdx.x];
      store_granule16(rrec + 2 * ((size_t)run * NACC + threadIdx.x), t, tag);
    }
  }
  if (!lost) {
    const int total = runs * NACC;
    for (int i = threadIdx.x; i < total; i += BLK) {
      const unsigned long long t0 = wall_clock64();
      granule_t q;
      for (unsigned int spins = 1;; spins++) {
        q = load_granule16(rrec + 2 * (size_t)i);
        if ((((unsigned long long)q.w << 32) | q.z) == tag) break;
        if ((spins & 63u) == 0 && wall_clock64() - t0 > 200000000ull) { lost = true; break; }   // 2 s: a run record never came
      }
      a_runs[i] = __longlong_as_double((long long)(((unsigned long long)q.y << 32) | q.x));
    }
  }
  lost = __syncthreads_or(lost);
  if (!lost && threadIdx.x < 32) {
    double t = 0.0;
    if (threadIdx.x < NACC) for (int r = 0; r < runs; r++) t += a_runs[r * NACC + threadIdx.x];
    a_tot[threadIdx.x] = t;
  }
  __syncthreads();
  if (!lost && threadIdx.x < 32) a_rec[threadIdx.x] = record_entry<MODE>(a_tot, threadIdx.x);
  __syncthreads();
  if (threadIdx.x == 0) {
    double step = 0.0;
    a_ok = !lost && gn_solve_update(a_rec, s_pose, &step) ? 1 : 0;
    a_step = step;
  }
  __syncthreads();
  const bool ok = a_ok != 0;
  const bool done = !ok || a_step < tol || it >= max_iters;
  if (done && blockIdx.x == 0 && threadIdx.x == 0 && fin.out_host) {
    for (int k = 0; k < 12; k++) __hip_atomic_store(fin.out_host + k, s_pose[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 12, a_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 13, lost ? 0.0 : a_rec[27], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 14, (double)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 15, ok ? 0.0 : (lost ? 2.0 : 1.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.out_host + 16, lost ? 0.0 : a_rec[28], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(fin.out_host + 32), fin.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return lost ? 2 : (done ? 1 : 0);
}

// IN_REGS: the grid covers all groups with one group per thread (frame-sized problems): each thread loads its group ONCE, before the
// loop, and keeps it in registers for the whole refinement.  Otherwise the slice is re-read every iteration (it stays cache resident).
template <class T, int KIND, int BLK, bool MASK, bool WEIGHT, bool IN_REGS, bool AUTO>
__global__ __launch_bounds__(BLK) void normal_eq_resident_kernel(const T* __restrict__ xw, const T* __restrict__ b, const T* __restrict__ c,
                                                                 const short* __restrict__ mask, const T* __restrict__ weight, int64_t n,
                                                                 const unsigned long long* __restrict__ ctl, unsigned long long first_tag,
                                                                 int max_iters, Finish fin) {
  constexpr int P = Pk<T>::P;
  constexpr int NACC = KIND == KIND_P2P ? 17 : 29;
  __shared__ double s_pose[12];
  __shared__ int s_go;
  const int64_t full = n / P, groups = (n + P - 1) / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const int64_t g0 = (int64_t)blockIdx.x * BLK + threadIdx.x;
  T rw[3 * P], rb[3 * P], rc[3 * P];
  short rm[P];
  T rwv[P];
  int rpresent = 0;
  if (IN_REGS && g0 < groups) {
    load_any_group<T, KIND, MASK, WEIGHT>(xw, b, c, mask, weight, g0, full, n, rw, rb, rc, rm, rwv);
    rpresent = g0 < full ? P : (int)(n - full * P);
  }
  // autonomous form (fin.gn set): the first pose comes from HBM, every later one from this workgroup's own solve (resident_auto_stage)
  constexpr bool autonomous = AUTO;   // a template parameter: the host-driven instances carry no call to the solve (registers, scratch)
  double tol = 0.0;
  if (autonomous) {
    if (threadIdx.x < 12) s_pose[threadIdx.x] = fin.gn_pose[threadIdx.x];
    tol = fin.gn->tol;
    __syncthreads();
  }
  for (int it = 1; it <= max_iters; it++) {
    if (!autonomous && resident_wait_pose<BLK>(ctl, first_tag + (unsigned long long)it, s_pose, &s_go) != 1) return;   // stop requested or no host: uniform for the workgroup
#ifdef RPE_STAMPS
    const bool stamp_it = it == 1000;
    if (stamp_it) RPE_STAMP(0);
#endif
    PoseK<double> pose;
#pragma unroll
    for (int k = 0; k < 9; k++) pose.R[k] = s_pose[k];
#pragma unroll
    for (int k = 0; k < 3; k++) pose.t[k] = s_pose[9 + k];
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = 0.0;
    if (IN_REGS) {
      if (rpresent > 0) normal_eq_group<T, KIND, MASK, WEIGHT, NACC>(pose, rw, rb, rc, rm, rwv, rpresent, acc);
    } else {
      for (int64_t g = g0; g < groups; g += stride) {
        T vw[3 * P], vb[3 * P], vc[3 * P];
        short mm[P];
        T ww[P];
        load_any_group<T, KIND, MASK, WEIGHT>(xw, b, c, mask, weight, g, full, n, vw, vb, vc, mm, ww);
        normal_eq_group<T, KIND, MASK, WEIGHT, NACC>(pose, vw, vb, vc, mm, ww, g < full ? P : (int)(n - full * P), acc);
      }
    }
#ifdef RPE_STAMPS
    if (stamp_it) RPE_STAMP(1);
#endif
#ifndef RPE_STAMPS
    const bool stamp_it = false;
#endif
    if (autonomous) {
      if (resident_auto_stage<NACC, BLK>(acc, fin, first_tag + (unsigned long long)it, it, max_iters, tol, s_pose) != 0) return;
      continue;
    }
    if (!resident_cross_stage<NACC, BLK>(acc, fin, first_tag + (unsigned long long)it, fin.seq + (unsigned long long)it, stamp_it)) return;
  }
}

// ================================================================================================
// F3 + K1/K2 fused: one ICP round in ONE pass.  Each frame vertex is paired with the model by projective association
// (rpe_assoc.h, the same function the stand-alone association kernel runs) and its residual is accumulated at once: the
// pairs never exist in HBM (48 B/pixel read -- frame vertex + normal streamed, model vertex + normal gathered -- against
// 120 B/pixel for the association pass plus 36 B/pixel for the normal-equation pass).  The pairing function
// and the per-pixel arithmetic are those of the two-kernel path; only the summation order differs (1e-13 relative).
// ================================================================================================
template <int KIND, int BLK>
__global__ __launch_bounds__(BLK) void icp_fused_kernel(const float* __restrict__ vmap, const float* __restrict__ nmap, int64_t n,
                                                        const float* __restrict__ mv, const float* __restrict__ mn, AssocParams P,
                                                        PoseK<double> pose, Finish fin) {
  constexpr int NACC = KIND == KIND_P2P ? 17 : 29;
  if (fin.gn != nullptr) {
    if (fin.gn->done) return;
#pragma unroll
    for (int k = 0; k < 9; k++) pose.R[k] = fin.gn_pose[k];
#pragma unroll
    for (int k = 0; k < 3; k++) pose.t[k] = fin.gn_pose[9 + k];
  }
  PoseF T;
#pragma unroll
  for (int k = 0; k < 9; k++) T.R[k] = (float)pose.R[k];
#pragma unroll
  for (int k = 0; k < 3; k++) T.t[k] = (float)pose.t[k];
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) acc[k] = 0.0;
  const short m_none[4] = {1, 1, 1, 1};
  const float w_none[4] = {1.f, 1.f, 1.f, 1.f};
  const float nan = __int_as_float(0x7fc00000);
  const int64_t full = n / 4;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const float4* __restrict__ v4 = reinterpret_cast<const float4*>(vmap);
  const float4* __restrict__ n4 = reinterpret_cast<const float4*>(nmap);
  for (int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x; g < full; g += stride) {
    float V[12], N[12], vw[12], vb[12], vc[12];
    unpack3(v4[3 * g], v4[3 * g + 1], v4[3 * g + 2], V);
    unpack3(n4[3 * g], n4[3 * g + 1], n4[3 * g + 2], N);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float gx, gy, gz;
      const bool ok = associate_pixel(T, P, mv, mn, V[3 * i], V[3 * i + 1], V[3 * i + 2], N[3 * i], N[3 * i + 1], N[3 * i + 2], vw[3 * i],
                                      vw[3 * i + 1], vw[3 * i + 2], gx, gy, gz);
#pragma unroll
      for (int k = 0; k < 3; k++) { vb[3 * i + k] = ok ? V[3 * i + k] : nan; vc[3 * i + k] = ok ? N[3 * i + k] : nan; }
    }
    normal_eq_group<float, KIND, false, false, NACC>(pose, vw, vb, vc, m_none, w_none, 4, acc);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && full * 4 < n) {  // leftover pixels
    float vw[12], vb[12], vc[12];
    const int left = (int)(n - full * 4);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float x = nan, y = nan, z = nan, nx = nan, ny = nan, nz = nan, gx, gy, gz;
      if (i < left) {
        const int64_t q = 3 * (full * 4 + i);
        x = vmap[q]; y = vmap[q + 1]; z = vmap[q + 2]; nx = nmap[q]; ny = nmap[q + 1]; nz = nmap[q + 2];
      }
      const bool ok = associate_pixel(T, P, mv, mn, x, y, z, nx, ny, nz, vw[3 * i], vw[3 * i + 1], vw[3 * i + 2], gx, gy, gz);
      vb[3 * i] = ok ? x : nan; vb[3 * i + 1] = ok ? y : nan; vb[3 * i + 2] = ok ? z : nan;
      vc[3 * i] = ok ? nx : nan; vc[3 * i + 1] = ok ? ny : nan; vc[3 * i + 2] = ok ? nz : nan;
    }
    normal_eq_group<float, KIND, false, false, NACC>(pose, vw, vb, vc, m_none, w_none, left, acc);
  }
  reduce_and_finish<NACC, kNeLd, KIND == KIND_P2P ? 1 : 0, BLK>(acc, fin);
}

// RESIDENT form of the fused ICP round (host-driven ICP: rpe_icp with fused = 1, device_resident = 0): ONE launch for the whole loop.
// The frame's vertices and normals (one group of 4 pixels per thread at 640 x 480) are read once and stay in registers; every
// iteration the workgroups wait for the host's pose (resident_wait_pose), pair their pixels with the model under that pose
// (associate_pixel: the model vertex / normal gathers are the only memory traffic of an iteration) and accumulate the normal equations,
// and the sums reach the host through the collecting stage (resident_cross_stage).  Pairing function and per-pixel arithmetic are
// the fused kernel's; only the order of the cross-workgroup sums differs.
template <int KIND, int BLK, bool IN_REGS, bool AUTO>
__global__ __launch_bounds__(BLK) void icp_resident_kernel(const float* __restrict__ vmap, const float* __restrict__ nmap, int64_t n,
                                                           const float* __restrict__ mv, const float* __restrict__ mn, AssocParams P,
                                                           const unsigned long long* __restrict__ ctl, unsigned long long first_tag,
                                                           int max_iters, Finish fin) {
  constexpr int NACC = KIND == KIND_P2P ? 17 : 29;
  __shared__ double s_pose[12];
  __shared__ int s_go;
  const short m_none[4] = {1, 1, 1, 1};
  const float w_none[4] = {1.f, 1.f, 1.f, 1.f};
  const float nan = __int_as_float(0x7fc00000);
  const int64_t full = n / 4, groups = (n + 3) / 4;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const int64_t g0 = (int64_t)blockIdx.x * BLK + threadIdx.x;
  // one group of frame pixels: 16-byte loads for whole groups, bounds-checked scalars for the ragged last one
  auto load_pixels = [&](int64_t g, float (&V)[12], float (&N)[12]) {
    if (g < full) {
      const float4* v4 = reinterpret_cast<const float4*>(vmap);
      const float4* n4 = reinterpret_cast<const float4*>(nmap);
      unpack3(v4[3 * g], v4[3 * g + 1], v4[3 * g + 2], V);
      unpack3(n4[3 * g], n4[3 * g + 1], n4[3 * g + 2], N);
    } else {
#pragma unroll
      for (int i = 0; i < 12; i++) {
        const int64_t q = 12 * g + i;
        const bool in = q < 3 * n;
        V[i] = in ? vmap[q] : nan;
        N[i] = in ? nmap[q] : nan;
      }
    }
  };
  float rV[12], rN[12];
  const bool mine = IN_REGS && g0 < groups;
  if (mine) load_pixels(g0, rV, rN);
  // AUTO (rpe_icp with device_resident): no host in the loop -- first pose from HBM, every later one from the workgroup's own solve
  double tol = 0.0;
  if (AUTO) {
    if (threadIdx.x < 12) s_pose[threadIdx.x] = fin.gn_pose[threadIdx.x];
    tol = fin.gn->tol;
    __syncthreads();
  }
  for (int it = 1; it <= max_iters; it++) {
    if (!AUTO && resident_wait_pose<BLK>(ctl, first_tag + (unsigned long long)it, s_pose, &s_go) != 1) return;
    PoseK<double> pose;
#pragma unroll
    for (int k = 0; k < 9; k++) pose.R[k] = s_pose[k];
#pragma unroll
    for (int k = 0; k < 3; k++) pose.t[k] = s_pose[9 + k];
    PoseF T;
#pragma unroll
    for (int k = 0; k < 9; k++) T.R[k] = (float)pose.R[k];
#pragma unroll
    for (int k = 0; k < 3; k++) T.t[k] = (float)pose.t[k];
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = 0.0;
    auto pair_and_add = [&](const float (&V)[12], const float (&N)[12], int present) {
      float vw[12], vb[12], vc[12];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float gx, gy, gz;
        const bool ok = associate_pixel(T, P, mv, mn, V[3 * i], V[3 * i + 1], V[3 * i + 2], N[3 * i], N[3 * i + 1], N[3 * i + 2], vw[3 * i],
                                        vw[3 * i + 1], vw[3 * i + 2], gx, gy, gz);
#pragma unroll
        for (int k = 0; k < 3; k++) { vb[3 * i + k] = ok ? V[3 * i + k] : nan; vc[3 * i + k] = ok ? N[3 * i + k] : nan; }
      }
      normal_eq_group<float, KIND, false, false, NACC>(pose, vw, vb, vc, m_none, w_none, present, acc);
    };
    if (IN_REGS) {
      if (mine) pair_and_add(rV, rN, g0 < full ? 4 : (int)(n - full * 4));
    } else {
      for (int64_t g = g0; g < groups; g += stride) {
        float V[12], N[12];
        load_pixels(g, V, N);
        pair_and_add(V, N, g < full ? 4 : (int)(n - full * 4));
      }
    }
    if (AUTO) {
      if (resident_auto_stage<NACC, BLK>(acc, fin, first_tag + (unsigned long long)it, it, max_iters, tol, s_pose) != 0) return;
      continue;
    }
    if (!resident_cross_stage<NACC, BLK>(acc, fin, first_tag + (unsigned long long)it, fin.seq + (unsigned long long)it, false)) return;
  }
}

// ================================================================================================
// K1+K2+K3 fused: joint Gauss-Newton normal equations of up to four residual kinds in ONE pass over the arrays
// (3D-3D point-to-point or point-to-plane, 2D-3D bearing, normal-normal), each with its modality's inlier mask,
// per-correspondence weight, a scale and an optional robust (IRLS) weight.  This is the single-kernel form of the
// objective nl_shinji_kneip_ls alternates over (M33 + sigma (M23 + MNN), AbsoluteOrientationNormal.hpp:484-510).
// Record: H upper triangle (21) | g (6) | sum scale w r^2 | sum w.   Up to 60 B/corr + masks/weights.
// ================================================================================================
enum { TERM_P2P = 1, TERM_P2PLANE = 2, TERM_BEARING = 4, TERM_NORMAL = 8 };
struct JointParams { double scale[4]; int robust[4]; double robust_k[4]; };  // indexed by residual kind 0..3

// `robust` is a kernel argument (wave-uniform): the branch is a scalar one, and the common case -- no robust weight -- pays
// neither the square root its argument needs nor the two divisions
template <class C, class F> __device__ __forceinline__ C robust_weight(int robust, C k, F norm_of_residual) {
  if (robust == 0) return C(1);
  const C s = norm_of_residual();
  const C huber = s <= k ? C(1) : k / s;
  const C q = s / k;
  const C cauchy = C(1) / (C(1) + q * q);
  return robust == 1 ? huber : cauchy;
}
// point-to-point block written straight into the packed record (J = [I | -[p]x]: 35 flops instead of 3 generic rows)
template <class C> __device__ __forceinline__ void p2p_packed(C px, C py, C pz, C rx, C ry, C rz, C w, C w_unscaled, C (&s)[29]) {
  s[0] += w; s[6] += w; s[11] += w;
  s[4] = fma(w, pz, s[4]); s[5] = fma(-w, py, s[5]); s[8] = fma(-w, pz, s[8]); s[10] = fma(w, px, s[10]);
  s[12] = fma(w, py, s[12]); s[13] = fma(-w, px, s[13]);
  const C wx = w * px, wy = w * py, wz = w * pz;
  s[15] = fma(wy, py, fma(wz, pz, s[15])); s[16] = fma(-wx, py, s[16]); s[17] = fma(-wx, pz, s[17]);
  s[18] = fma(wx, px, fma(wz, pz, s[18])); s[19] = fma(-wy, pz, s[19]); s[20] = fma(wx, px, fma(wy, py, s[20]));
  const C wrx = w * rx, wry = w * ry, wrz = w * rz;
  s[21] += wrx; s[22] += wry; s[23] += wrz;
  s[24] += py * wrz - pz * wry; s[25] += pz * wrx - px * wrz; s[26] += px * wry - py * wrx;
  s[27] = fma(wrx, rx, fma(wry, ry, fma(wrz, rz, s[27])));
  s[28] += w_unscaled;
}

template <class T, int TERMS>
__device__ __forceinline__ void joint_group(const PoseK<double>& pose, const JointParams& prm, const T (&vw)[3 * Pk<T>::P],
                                            const T (&vc)[3 * Pk<T>::P], const T (&vb)[3 * Pk<T>::P], const T (&vnw)[3 * Pk<T>::P],
                                            const T (&vnc)[3 * Pk<T>::P], const short (&k23)[Pk<T>::P], const short (&k33)[Pk<T>::P],
                                            const short (&knn)[Pk<T>::P], const T (&u23)[Pk<T>::P], const T (&u33)[Pk<T>::P],
                                            const T (&unn)[Pk<T>::P], int npresent, double (&acc)[29]) {
  constexpr int P = Pk<T>::P;
  constexpr bool HAS33 = (TERMS & (TERM_P2P | TERM_P2PLANE)) != 0;
  T s[29];
#pragma unroll
  for (int k = 0; k < 29; k++) s[k] = T(0);
#pragma unroll
  for (int i = 0; i < P; i++) {
    const bool present = i < npresent;
    T x = vw[3 * i], y = vw[3 * i + 1], z = vw[3 * i + 2];
    double pxd, pyd, pzd;
    transform<T>(pose, x, y, z, pxd, pyd, pzd);
    const T px = (T)pxd, py = (T)pyd, pz = (T)pzd;
    if (HAS33) {
      const T cx = vc[3 * i], cy = vc[3 * i + 1], cz = vc[3 * i + 2];
      const bool on = present & (k33[i] == 1) & !all_nan(cx, cy, cz);
      const T rx = on ? (T)(pxd - (double)cx) : T(0), ry = on ? (T)(pyd - (double)cy) : T(0), rz = on ? (T)(pzd - (double)cz) : T(0);
      const T qx = on ? px : T(0), qy = on ? py : T(0), qz = on ? pz : T(0);
      if (TERMS & TERM_P2P) {
        const T w0 = on ? u33[i] : T(0);
        const T w = w0 * robust_weight<T>(prm.robust[0], (T)prm.robust_k[0], [&]() { return sqrt(rx * rx + ry * ry + rz * rz); });
        p2p_packed<T>(qx, qy, qz, rx, ry, rz, (T)prm.scale[0] * w, w, s);
      }
      if (TERMS & TERM_P2PLANE) {
        const T nx = on ? vnc[3 * i] : T(0), ny = on ? vnc[3 * i + 1] : T(0), nz = on ? vnc[3 * i + 2] : T(0);
        const T r = nx * rx + ny * ry + nz * rz;
        const T w0 = on ? u33[i] : T(0);
        const T w = w0 * robust_weight<T>(prm.robust[1], (T)prm.robust_k[1], [&]() { return fabs(r); });
        const T J[6] = {nx, ny, nz, qy * nz - qz * ny, qz * nx - qx * nz, qx * ny - qy * nx};
        add_row(J, r, (T)prm.scale[1] * w, s);
        s[28] += w;
      }
    }
    if (TERMS & TERM_BEARING) {
      const T bx0 = vb[3 * i], by0 = vb[3 * i + 1], bz0 = vb[3 * i + 2];
      const bool on = present & (k23[i] == 1) & !all_nan(bx0, by0, bz0);
      const T bx = on ? bx0 : T(0), by = on ? by0 : T(0), bz = on ? bz0 : T(1);
      const double sx = on ? pxd : 0.0, sy = on ? pyd : 0.0, sz = on ? pzd : 1.0;  // keeps the normalisation finite when off
      const double invd = rsqrt64(sx * sx + sy * sy + sz * sz);
      const double hxd = sx * invd, hyd = sy * invd, hzd = sz * invd;
      const T r[3] = {(T)(hyd * (double)bz - hzd * (double)by), (T)(hzd * (double)bx - hxd * (double)bz), (T)(hxd * (double)by - hyd * (double)bx)};
      const T w0 = on ? u23[i] : T(0);
      const T w = w0 * robust_weight<T>(prm.robust[2], (T)prm.robust_k[2], [&]() { return sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]); });
      const T ws = (T)prm.scale[2] * w;
      const T qx = (T)sx, qy = (T)sy, qz = (T)sz, inv = (T)invd;
      const T h[3] = {(T)hxd, (T)hyd, (T)hzd};
      const T Bx[3][3] = {{T(0), -bz, by}, {bz, T(0), -bx}, {-by, bx, T(0)}};
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const T bh = Bx[u][0] * h[0] + Bx[u][1] * h[1] + Bx[u][2] * h[2];
        const T a0 = -(Bx[u][0] - bh * h[0]) * inv, a1 = -(Bx[u][1] - bh * h[1]) * inv, a2 = -(Bx[u][2] - bh * h[2]) * inv;
        const T J[6] = {a0, a1, a2, qy * a2 - qz * a1, qz * a0 - qx * a2, qx * a1 - qy * a0};
        add_row(J, r[u], ws, s);
      }
      s[28] += w;
    }
    if (TERMS & TERM_NORMAL) {
      const T mx = vnw[3 * i], my = vnw[3 * i + 1], mz = vnw[3 * i + 2];
      const T cx0 = vnc[3 * i], cy0 = vnc[3 * i + 1], cz0 = vnc[3 * i + 2];
      const bool on = present & (knn[i] == 1) & !all_nan(cx0, cy0, cz0);
      // q = R Nw and r = q - Nc in fp64, like p and its residuals: rounding R to fp32 would leave a 6e-8 step floor
      const double mxd = mx, myd = my, mzd = mz;
      const double qxd = fma(pose.R[0], mxd, fma(pose.R[1], myd, pose.R[2] * mzd));
      const double qyd = fma(pose.R[3], mxd, fma(pose.R[4], myd, pose.R[5] * mzd));
      const double qzd = fma(pose.R[6], mxd, fma(pose.R[7], myd, pose.R[8] * mzd));
      const T qx = on ? (T)qxd : T(0), qy = on ? (T)qyd : T(0), qz = on ? (T)qzd : T(0);
      const T rx = on ? (T)(qxd - (double)cx0) : T(0), ry = on ? (T)(qyd - (double)cy0) : T(0), rz = on ? (T)(qzd - (double)cz0) : T(0);
      const T w0 = on ? unn[i] : T(0);
      const T w = w0 * robust_weight<T>(prm.robust[3], (T)prm.robust_k[3], [&]() { return sqrt(rx * rx + ry * ry + rz * rz); });
      const T ws = (T)prm.scale[3] * w;
      // J = [0 | -[q]x] : only the rotation block:  H_ww += |q|^2 I - q q^T ,  g_w += q x r
      const T wx = ws * qx, wy = ws * qy, wz = ws * qz;
      s[15] = fma(wy, qy, fma(wz, qz, s[15])); s[16] = fma(-wx, qy, s[16]); s[17] = fma(-wx, qz, s[17]);
      s[18] = fma(wx, qx, fma(wz, qz, s[18])); s[19] = fma(-wy, qz, s[19]); s[20] = fma(wx, qx, fma(wy, qy, s[20]));
      s[24] += ws * (qy * rz - qz * ry); s[25] += ws * (qz * rx - qx * rz); s[26] += ws * (qx * ry - qy * rx);
      s[27] = fma(ws * rx, rx, fma(ws * ry, ry, fma(ws * rz, rz, s[27])));
      s[28] += w;
    }
  }
#pragma unroll
  for (int k = 0; k < 29; k++) acc[k] += (double)s[k];
}

template <class T, int TERMS, int BLK>
__global__ __launch_bounds__(BLK) void normal_eq_joint_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const T* __restrict__ bv,
                                                              const T* __restrict__ nw, const T* __restrict__ nc,
                                                              const short* __restrict__ m23, const short* __restrict__ m33,
                                                              const short* __restrict__ mnn, const T* __restrict__ w23,
                                                              const T* __restrict__ w33, const T* __restrict__ wnn, int64_t n,
                                                              PoseK<double> pose, JointParams prm, Finish fin) {
  constexpr int P = Pk<T>::P;
  constexpr bool HAS33 = (TERMS & (TERM_P2P | TERM_P2PLANE)) != 0;
  constexpr bool NEED_NC = (TERMS & (TERM_P2PLANE | TERM_NORMAL)) != 0;
  if (fin.gn != nullptr) {
    if (fin.gn->done) return;
#pragma unroll
    for (int k = 0; k < 9; k++) pose.R[k] = fin.gn_pose[k];
#pragma unroll
    for (int k = 0; k < 3; k++) pose.t[k] = fin.gn_pose[9 + k];
  }
  double acc[29];
#pragma unroll
  for (int k = 0; k < 29; k++) acc[k] = 0.0;
  const int64_t groups = (n + P - 1) / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  for (int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x; g < groups; g += stride) {
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    short k23[P], k33[P], knn[P];
    T u23[P], u33[P], unn[P];
#pragma unroll
    for (int i = 0; i < P; i++) { k23[i] = k33[i] = knn[i] = 1; u23[i] = u33[i] = unn[i] = T(1); }
    load_group<T>(xw, g, n, vw);
    if (HAS33) { load_group<T>(xc, g, n, vc); if (m33) load_mask_group(m33, g, n, k33); if (w33) load_weight_group(w33, g, n, u33); }
    if (TERMS & TERM_BEARING) { load_group<T>(bv, g, n, vb); if (m23) load_mask_group(m23, g, n, k23); if (w23) load_weight_group(w23, g, n, u23); }
    if (TERMS & TERM_NORMAL) { load_group<T>(nw, g, n, vnw); if (mnn) load_mask_group(mnn, g, n, knn); if (wnn) load_weight_group(wnn, g, n, unn); }
    if (NEED_NC) load_group<T>(nc, g, n, vnc);
    const int64_t left = n - g * P;
    joint_group<T, TERMS>(pose, prm, vw, vc, vb, vnw, vnc, k23, k33, knn, u23, u33, unn, left < P ? (int)left : P, acc);
  }
  reduce_and_finish<29, kNeLd, 0, BLK>(acc, fin);
}

// ================================================================================================
// K1' : closed-form moments (both passes of shinji() in one): w | w Xw | w Xc | w Xc Xw^T | w |Xc|^2 | count
// fp32 x fp32 products are exact in fp64, so only the fp64 summation rounds.
// ================================================================================================
template <class T, bool MASK, bool WEIGHT>
__device__ __forceinline__ void moments_group(const T (&vw)[3 * Pk<T>::P], const T (&vc)[3 * Pk<T>::P], const short (&m)[Pk<T>::P],
                                              const T (&wv)[Pk<T>::P], int npresent, int skip_invalid, double (&acc)[18]) {
  constexpr int P = Pk<T>::P;
#pragma unroll
  for (int i = 0; i < P; i++) {
    double x = vw[3 * i], y = vw[3 * i + 1], z = vw[3 * i + 2];
    double cx = vc[3 * i], cy = vc[3 * i + 1], cz = vc[3 * i + 2];
    bool use = i < npresent && !(skip_invalid && all_nan(cx, cy, cz));
    if (MASK) use = use && m[i] == 1;
    const double wi = use ? (WEIGHT ? (double)wv[i] : 1.0) : 0.0;
    x = use ? x : 0.0; y = use ? y : 0.0; z = use ? z : 0.0; cx = use ? cx : 0.0; cy = use ? cy : 0.0; cz = use ? cz : 0.0;
    const double wcx = wi * cx, wcy = wi * cy, wcz = wi * cz;
    acc[0] += wi;
    acc[1] = fma(wi, x, acc[1]); acc[2] = fma(wi, y, acc[2]); acc[3] = fma(wi, z, acc[3]);
    acc[4] += wcx; acc[5] += wcy; acc[6] += wcz;
    acc[7] = fma(wcx, x, acc[7]); acc[8] = fma(wcx, y, acc[8]); acc[9] = fma(wcx, z, acc[9]);
    acc[10] = fma(wcy, x, acc[10]); acc[11] = fma(wcy, y, acc[11]); acc[12] = fma(wcy, z, acc[12]);
    acc[13] = fma(wcz, x, acc[13]); acc[14] = fma(wcz, y, acc[14]); acc[15] = fma(wcz, z, acc[15]);
    acc[16] = fma(wcx, cx, fma(wcy, cy, fma(wcz, cz, acc[16])));
    acc[17] += use ? 1.0 : 0.0;
  }
}

template <class T, int BLK, bool MASK, bool WEIGHT>
__global__ __launch_bounds__(BLK) void moments_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const short* __restrict__ mask,
                                                      const T* __restrict__ weight, int64_t n, int skip_invalid, Finish fin) {
  constexpr int P = Pk<T>::P;
  typedef typename Pk<T>::V V;
  double acc[18];
#pragma unroll
  for (int k = 0; k < 18; k++) acc[k] = 0.0;
  const int64_t full = n / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const V* __restrict__ xw4 = reinterpret_cast<const V*>(xw);
  const V* __restrict__ xc4 = reinterpret_cast<const V*>(xc);
  // same software pipeline as normal_eq_kernel: next group's loads in flight while this one is accumulated
  int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
  V a0, a1, a2, b0, b1, b2;
  short m[P];
  T wv[P];
  if (g < full) {
    a0 = xw4[3 * g]; a1 = xw4[3 * g + 1]; a2 = xw4[3 * g + 2];
    b0 = xc4[3 * g]; b1 = xc4[3 * g + 1]; b2 = xc4[3 * g + 2];
    if (MASK) load_mask_full(mask, g, m);
    if (WEIGHT) load_weight_full(weight, g, wv);
  }
  while (g < full) {
    const int64_t gn = g + stride;
    const int64_t gl = gn < full ? gn : g;
    const V na0 = xw4[3 * gl], na1 = xw4[3 * gl + 1], na2 = xw4[3 * gl + 2];
    const V nb0 = xc4[3 * gl], nb1 = xc4[3 * gl + 1], nb2 = xc4[3 * gl + 2];
    short nm[P];
    T nwv[P];
    if (MASK) load_mask_full(mask, gl, nm);
    if (WEIGHT) load_weight_full(weight, gl, nwv);
    T vw[3 * P], vc[3 * P];
    unpack3(a0, a1, a2, vw);
    unpack3(b0, b1, b2, vc);
    moments_group<T, MASK, WEIGHT>(vw, vc, m, wv, P, skip_invalid, acc);
    a0 = na0; a1 = na1; a2 = na2; b0 = nb0; b1 = nb1; b2 = nb2;
#pragma unroll
    for (int i = 0; i < P; i++) { if (MASK) m[i] = nm[i]; if (WEIGHT) wv[i] = nwv[i]; }
    g = gn;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && full * P < n) {
    T vw[3 * P], vc[3 * P];
    load_group<T>(xw, full, n, vw);
    load_group<T>(xc, full, n, vc);
    if (MASK) load_scalars<T, short>(mask, full, n, m, (short)0);
    if (WEIGHT) load_scalars<T, T>(weight, full, n, wv, T(0));
    moments_group<T, MASK, WEIGHT>(vw, vc, m, wv, (int)(n - full * P), skip_invalid, acc);
  }
  reduce_and_finish<18, kNeLd, 0, BLK>(acc, fin);
}

// ================================================================================================
// K4 : batched hypothesis scoring (vote loops V1..V8) and K4b : winner mask
// ================================================================================================
enum { VOTE_33 = 0, VOTE_23 = 1, VOTE_33_23 = 2, VOTE_NN_23 = 3, VOTE_NN_33 = 4, VOTE_NN_33_23 = 5, VOTE_23_MATRIX = 6 };
template <int KIND> struct VoteMods {
  static constexpr bool m33 = KIND == VOTE_33 || KIND == VOTE_33_23 || KIND == VOTE_NN_33 || KIND == VOTE_NN_33_23;
  static constexpr bool m23 = KIND == VOTE_23 || KIND == VOTE_33_23 || KIND == VOTE_NN_23 || KIND == VOTE_NN_33_23 || KIND == VOTE_23_MATRIX;
  static constexpr bool mnn = KIND == VOTE_NN_23 || KIND == VOTE_NN_33 || KIND == VOTE_NN_33_23;
  static constexpr bool need_xc = m33 || mnn;  // isValid() gates the N-N vote too
};

// one hypothesis in registers (wave-uniform -> SGPRs)
template <class T, bool EXACT> struct Hyp;
template <class T> struct Hyp<T, false> {
  T R[9], t[3];
  enum { STRIDE = 12 };
  __device__ __forceinline__ void load(const T* __restrict__ p, bool) {
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = p[k];
    t[0] = p[9]; t[1] = p[10]; t[2] = p[11];
  }
  __device__ __forceinline__ void rot(T x, T y, T z, T& ox, T& oy, T& oz) const {
    ox = fma(R[0], x, fma(R[1], y, R[2] * z));
    oy = fma(R[3], x, fma(R[4], y, R[5] * z));
    oz = fma(R[6], x, fma(R[7], y, R[8] * z));
  }
  // fast forms: squared distance / squared cosine compares (no sqrt, no divide)
  __device__ __forceinline__ bool in33(T x, T y, T z, T cx, T cy, T cz, T thr_sq) const {
    const T ex = fma(R[0], x, fma(R[1], y, fma(R[2], z, t[0] - cx)));
    const T ey = fma(R[3], x, fma(R[4], y, fma(R[5], z, t[1] - cy)));
    const T ez = fma(R[6], x, fma(R[7], y, fma(R[8], z, t[2] - cz)));
    return fma(ex, ex, fma(ey, ey, ez * ez)) < thr_sq;
  }
  __device__ __forceinline__ bool in23(T x, T y, T z, T bx, T by, T bz, T c, bool) const {
    const T px = fma(R[0], x, fma(R[1], y, fma(R[2], z, t[0])));
    const T py = fma(R[3], x, fma(R[4], y, fma(R[5], z, t[1])));
    const T pz = fma(R[6], x, fma(R[7], y, fma(R[8], z, t[2])));
    const T d = fma(px, bx, fma(py, by, pz * bz));
    const T n2 = fma(px, px, fma(py, py, pz * pz));
    // d > c |p|  <=>  d|d| > c|c| |p|^2  (u -> u|u| is strictly increasing): one branch-free form for either sign of c
    return d * fabs(d) > (c * fabs(c)) * n2;
  }
  __device__ __forceinline__ bool innn(T nwx, T nwy, T nwz, T ncx, T ncy, T ncz, T cnl) const {
    T rx, ry, rz;
    rot(nwx, nwy, nwz, rx, ry, rz);
    return fma(ncx, rx, fma(ncy, ry, ncz * rz)) > cnl;
  }
};
// exact form: the reference's own operation sequence in Tp, no FMA contraction.
//   R*x     = Eigen _transformVector (sophus/so3.hpp:238-240): uv = 2 (u x v); v + w uv + u x uv
//   3D test = |Xc - (R Xw + t)| < thre_3d with norm = sqrt(x^2 + y^2 + z^2)      (AbsoluteOrientation.hpp:137-138), evaluated as x^2 + y^2 + z^2 < cut
//   2D test = normalize(R Xw + t) . bv > cos_thr, normalisation by division        (:413-418)
//   N-N     = Nc . (R Nw) > cos_nl                                                  (AbsoluteOrientationNormal.hpp:248-249)
template <class T> struct Hyp<T, true> {
  T qw, qx, qy, qz, t[3];
  T M[9];  // toRotationMatrix(), only for the kneip_ransac variant that multiplies by so3().matrix() (P3P.hpp:365)
  enum { STRIDE = 8 };
  __device__ __forceinline__ void load(const T* __restrict__ p, bool need_matrix) {
#pragma clang fp contract(off)
    qw = p[0]; qx = p[1]; qy = p[2]; qz = p[3]; t[0] = p[4]; t[1] = p[5]; t[2] = p[6];
    if (!need_matrix) return;
    const T tx = T(2) * qx, ty = T(2) * qy, tz = T(2) * qz;
    const T twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const T tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    M[0] = T(1) - (tyy + tzz); M[1] = txy - twz; M[2] = txz + twy;
    M[3] = txy + twz; M[4] = T(1) - (txx + tzz); M[5] = tyz - twx;
    M[6] = txz - twy; M[7] = tyz + twx; M[8] = T(1) - (txx + tyy);
  }
  __device__ __forceinline__ void rot(T x, T y, T z, T& ox, T& oy, T& oz) const {
#pragma clang fp contract(off)
    T ux = qy * z - qz * y, uy = qz * x - qx * z, uz = qx * y - qy * x;
    ux = ux + ux; uy = uy + uy; uz = uz + uz;
    const T cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
    ox = (x + qw * ux) + cx; oy = (y + qw * uy) + cy; oz = (z + qw * uz) + cz;
  }
  // `cut` = the smallest value whose correctly rounded square root reaches thre_3d (computed on the host, rpe_capi.hip sqrt_cut):
  // sqrt is monotonic, so  sqrt(s) < thre_3d  <=>  s < cut  for every s -- the reference's test, bit for bit, without the square root
  // (a third of the instructions of this predicate).  s is formed exactly as Eigen's squaredNorm() forms it.
  __device__ __forceinline__ bool in33(T x, T y, T z, T cx, T cy, T cz, T cut) const {
#pragma clang fp contract(off)
    T rx, ry, rz;
    rot(x, y, z, rx, ry, rz);
    const T ex = cx - (rx + t[0]), ey = cy - (ry + t[1]), ez = cz - (rz + t[2]);
    return (ex * ex + ey * ey + ez * ez) < cut;
  }
  // two correspondences at once as 2-vectors (element-wise IEEE operations in the same order as above: the same bits), so that the
  // fp32 forms issue as packed v_pk_mul_f32 / v_pk_add_f32 -- half the instructions of the scalar sequence
  typedef T V2 __attribute__((ext_vector_type(2)));
  __device__ __forceinline__ void rot2(V2 x, V2 y, V2 z, V2& ox, V2& oy, V2& oz) const {
#pragma clang fp contract(off)
    V2 ux = qy * z - qz * y, uy = qz * x - qx * z, uz = qx * y - qy * x;
    ux = ux + ux; uy = uy + uy; uz = uz + uz;
    const V2 cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
    ox = (x + qw * ux) + cx; oy = (y + qw * uy) + cy; oz = (z + qw * uz) + cz;
  }
  __device__ __forceinline__ void rotm2(V2 x, V2 y, V2 z, V2& ox, V2& oy, V2& oz) const {   // so3().matrix() * x (kneip_ransac, P3P.hpp:365)
#pragma clang fp contract(off)
    ox = M[0] * x + M[1] * y + M[2] * z; oy = M[3] * x + M[4] * y + M[5] * z; oz = M[6] * x + M[7] * y + M[8] * z;
  }
  // the 3D and 2D tests of a pair, given the pair's rotated points (one rotation serves both tests, as in the scalar code after CSE)
  __device__ __forceinline__ void in33_rot_x2(V2 rx, V2 ry, V2 rz, V2 cx, V2 cy, V2 cz, T cut, bool& a, bool& b) const {
#pragma clang fp contract(off)
    const V2 ex = cx - (rx + t[0]), ey = cy - (ry + t[1]), ez = cz - (rz + t[2]);
    const V2 ss = ex * ex + ey * ey + ez * ez;
    a = ss.x < cut; b = ss.y < cut;
  }
  __device__ __forceinline__ void in23_rot_x2(V2 rx, V2 ry, V2 rz, V2 bx, V2 by, V2 bz, T c, bool& a, bool& b) const {
#pragma clang fp contract(off)
    V2 px = rx + t[0], py = ry + t[1], pz = rz + t[2];
    const V2 n2 = px * px + py * py + pz * pz;
    const V2 len = {sqrt(n2.x), sqrt(n2.y)};
    px = px / len; py = py / len; pz = pz / len;
    const V2 d = px * bx + py * by + pz * bz;
    a = d.x > c; b = d.y > c;
  }
  __device__ __forceinline__ void innnx2(V2 nwx, V2 nwy, V2 nwz, V2 ncx, V2 ncy, V2 ncz, T cnl, bool& a, bool& b) const {
#pragma clang fp contract(off)
    V2 rx, ry, rz;
    rot2(nwx, nwy, nwz, rx, ry, rz);
    const V2 d = ncx * rx + ncy * ry + ncz * rz;
    a = d.x > cnl; b = d.y > cnl;
  }
  __device__ __forceinline__ bool in23(T x, T y, T z, T bx, T by, T bz, T c, bool use_matrix) const {
#pragma clang fp contract(off)
    T rx, ry, rz;
    if (use_matrix) {
      rx = M[0] * x + M[1] * y + M[2] * z; ry = M[3] * x + M[4] * y + M[5] * z; rz = M[6] * x + M[7] * y + M[8] * z;
    } else {
      rot(x, y, z, rx, ry, rz);
    }
    T px = rx + t[0], py = ry + t[1], pz = rz + t[2];
    const T len = sqrt(px * px + py * py + pz * pz);
    px = px / len; py = py / len; pz = pz / len;
    return (px * bx + py * by + pz * bz) > c;
  }
  __device__ __forceinline__ bool innn(T nwx, T nwy, T nwz, T ncx, T ncy, T ncz, T cnl) const {
#pragma clang fp contract(off)
    T rx, ry, rz;
    rot(nwx, nwy, nwz, rx, ry, rz);
    return (ncx * rx + ncy * ry + ncz * rz) > cnl;
  }
};

// votes of ONE hypothesis over one group of P correspondences, summed over the wave (every lane gets the wave's count).  Predicates are
// evaluated unconditionally and masked with '&': no divergent branches; the compare IS the ballot.  EXACT: the 3D and normal tests run
// on pairs of correspondences as 2-vectors (packed fp32 instructions), the 2D test (a square root and three divisions) stays scalar.
template <class T, int KIND, bool EXACT>
__device__ __forceinline__ int count_group_votes(const Hyp<T, EXACT>& hyp, const T (&vw)[3 * Pk<T>::P], const T (&vc)[3 * Pk<T>::P],
                                                 const T (&vb)[3 * Pk<T>::P], const T (&vnw)[3 * Pk<T>::P], const T (&vnc)[3 * Pk<T>::P],
                                                 const bool (&present)[Pk<T>::P], const bool (&valid)[Pk<T>::P], T thr33, T cthr, T cnl) {
  constexpr int P = Pk<T>::P;
  typedef VoteMods<KIND> MD;
  int cnt = 0;
  if constexpr (EXACT) {
    typedef T V2 __attribute__((ext_vector_type(2)));
    static_assert(P % 2 == 0, "pairs");
#pragma unroll
    for (int j = 0; j < P / 2; j++) {
      const int a = 2 * j, b = 2 * j + 1;
      const V2 x = {vw[3 * a], vw[3 * b]}, y = {vw[3 * a + 1], vw[3 * b + 1]}, z = {vw[3 * a + 2], vw[3 * b + 2]};
      if (MD::mnn) {
        const V2 nwx = {vnw[3 * a], vnw[3 * b]}, nwy = {vnw[3 * a + 1], vnw[3 * b + 1]}, nwz = {vnw[3 * a + 2], vnw[3 * b + 2]};
        const V2 ncx = {vnc[3 * a], vnc[3 * b]}, ncy = {vnc[3 * a + 1], vnc[3 * b + 1]}, ncz = {vnc[3 * a + 2], vnc[3 * b + 2]};
        bool va, vb2;
        hyp.innnx2(nwx, nwy, nwz, ncx, ncy, ncz, cnl, va, vb2);
        cnt += __popcll(__ballot(valid[a] & va)) + __popcll(__ballot(valid[b] & vb2));
      }
      V2 rx, ry, rz;
      if (KIND == VOTE_23_MATRIX) hyp.rotm2(x, y, z, rx, ry, rz); else hyp.rot2(x, y, z, rx, ry, rz);
      if (MD::m33) {
        const V2 cx = {vc[3 * a], vc[3 * b]}, cy = {vc[3 * a + 1], vc[3 * b + 1]}, cz = {vc[3 * a + 2], vc[3 * b + 2]};
        bool va, vb2;
        hyp.in33_rot_x2(rx, ry, rz, cx, cy, cz, thr33, va, vb2);
        cnt += __popcll(__ballot(valid[a] & va)) + __popcll(__ballot(valid[b] & vb2));
      }
      if (MD::m23) {
        const V2 bx = {vb[3 * a], vb[3 * b]}, by = {vb[3 * a + 1], vb[3 * b + 1]}, bz = {vb[3 * a + 2], vb[3 * b + 2]};
        bool va, vb2;
        hyp.in23_rot_x2(rx, ry, rz, bx, by, bz, cthr, va, vb2);
        cnt += __popcll(__ballot(present[a] & va)) + __popcll(__ballot(present[b] & vb2));
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < P; i++) {
      const T x = vw[3 * i], y = vw[3 * i + 1], z = vw[3 * i + 2];
      if (MD::mnn) {
        const bool v = valid[i] & hyp.innn(vnw[3 * i], vnw[3 * i + 1], vnw[3 * i + 2], vnc[3 * i], vnc[3 * i + 1], vnc[3 * i + 2], cnl);
        cnt += __popcll(__ballot(v));
      }
      if (MD::m33) {
        const bool v = valid[i] & hyp.in33(x, y, z, vc[3 * i], vc[3 * i + 1], vc[3 * i + 2], thr33);
        cnt += __popcll(__ballot(v));
      }
      if (MD::m23) {
        const bool v = present[i] & hyp.in23(x, y, z, vb[3 * i], vb[3 * i + 1], vb[3 * i + 2], cthr, KIND == VOTE_23_MATRIX);
        cnt += __popcll(__ballot(v));
      }
    }
  }
  return cnt;
}

// grid = (x: correspondence tiles, grid-stride) x (y: chunks of `hchunk` hypotheses).  Small problems (640x480 frames)
// cannot fill 256 CUs with one tile sweep, so the hypothesis list is split across blockIdx.y and the (L2-resident)
// arrays are swept once per chunk; large problems use one chunk so the arrays stream from HBM once per launch.
template <class T, int KIND, bool EXACT>
__global__ __launch_bounds__(kBlock) void score_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const T* __restrict__ bv,
                                                       const T* __restrict__ nw, const T* __restrict__ nc, int64_t n,
                                                       const T* __restrict__ poses, int H, int hchunk, T thr33, T cthr, T cnl,
                                                       int* __restrict__ votes) {
  constexpr int P = Pk<T>::P;
  typedef VoteMods<KIND> MD;
  extern __shared__ int lds_votes[];
  const int hbeg = blockIdx.y * hchunk;
  const int hcnt = min(hchunk, H - hbeg);
  for (int i = threadIdx.x; i < hcnt; i += kBlock) lds_votes[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t groups = (n + P - 1) / P;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  // loop bound is workgroup-uniform so that every lane of a wave takes part in the per-hypothesis ballots
  for (int64_t gb = (int64_t)blockIdx.x * kBlock; gb < groups; gb += stride) {
    const int64_t g = gb + threadIdx.x;
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    bool present[P], valid[P];
    load_group<T>(xw, g, n, vw);
    if (MD::need_xc) load_group<T>(xc, g, n, vc);
    if (MD::m23) load_group<T>(bv, g, n, vb);
    if (MD::mnn) { load_group<T>(nw, g, n, vnw); load_group<T>(nc, g, n, vnc); }
#pragma unroll
    for (int i = 0; i < P; i++) {
      present[i] = (g * P + i) < n;
      valid[i] = present[i] & (!MD::need_xc || !all_nan(vc[3 * i], vc[3 * i + 1], vc[3 * i + 2]));
    }
    for (int h0 = 0; h0 < hcnt; h0 += 64) {
      const int hmax = min(64, hcnt - h0);
      int mine = 0;
      for (int hl = 0; hl < hmax; hl++) {
        Hyp<T, EXACT> hyp;
        hyp.load(poses + (size_t)(hbeg + h0 + hl) * Hyp<T, EXACT>::STRIDE, KIND == VOTE_23_MATRIX);
        const int cnt = count_group_votes<T, KIND, EXACT>(hyp, vw, vc, vb, vnw, vnc, present, valid, thr33, cthr, cnl);
        mine += (lane == hl) ? cnt : 0;
      }
      if (lane < hmax && mine != 0) atomicAdd(&lds_votes[h0 + lane], mine);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < hcnt; i += kBlock) {
    const int v = lds_votes[i];
    if (v != 0) atomicAdd(&votes[hbeg + i], v);
  }
}

// Small batches (the first RANSAC batches: 8 .. 32 hypotheses): ONE launch and no device-side staging at all -- the hypotheses arrive as
// a kernel argument (no H2D copy), every wave counts as above (lane h holds hypothesis h's count), the per-wave counts go straight into
// the collecting stage (collect_and_send: integers < 2^53 as doubles, exact), and the host adds the run records.  Replaces copy +
// scoring kernel + read-out kernel + flag (42 us per batch of 16 at 640 x 480) for lists of up to HB hypotheses.
template <class T, int HB, int STRIDE> struct SmallPoses { T v[HB * STRIDE]; };
template <class T, int KIND, bool EXACT, int HB>
__global__ __launch_bounds__(kBlock) void score_small_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const T* __restrict__ bv,
                                                             const T* __restrict__ nw, const T* __restrict__ nc, int64_t n,
                                                             SmallPoses<T, HB, Hyp<T, EXACT>::STRIDE> sp, const T* __restrict__ dposes, int H, int hs,
                                                             T thr33, T cthr, T cnl, Finish fin) {
  constexpr int P = Pk<T>::P;
  typedef VoteMods<KIND> MD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // hs (1, 2 or 4) waves share a tile of correspondences and split the list between them (wave copy c takes hypotheses c, c + hs, ...):
  // a frame-sized problem then runs as 4 x as many, 4 x shorter waves -- the pass is a long serial chain per wave (every predicate of
  // every hypothesis on the wave's points), so with one group per thread it is bound by that chain, not by memory or issue rate
  const int per_copy = (kBlock / 64) / hs, copy = wave / per_copy;
  const int tile = kBlock / hs;
  const int64_t groups = (n + P - 1) / P;
  const int64_t stride = (int64_t)gridDim.x * tile;
  int mine = 0;
  // loop bound is workgroup-uniform so that every lane of a wave takes part in the per-hypothesis ballots
  for (int64_t gb = (int64_t)blockIdx.x * tile; gb < groups; gb += stride) {
    const int64_t g = gb + (wave % per_copy) * 64 + lane;
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    bool present[P], valid[P];
    load_group<T>(xw, g, n, vw);
    if (MD::need_xc) load_group<T>(xc, g, n, vc);
    if (MD::m23) load_group<T>(bv, g, n, vb);
    if (MD::mnn) { load_group<T>(nw, g, n, vnw); load_group<T>(nc, g, n, vnc); }
#pragma unroll
    for (int i = 0; i < P; i++) {
      present[i] = (g * P + i) < n;
      valid[i] = present[i] & (!MD::need_xc || !all_nan(vc[3 * i], vc[3 * i + 1], vc[3 * i + 2]));
    }
    for (int hl = copy; hl < H; hl += hs) {
      Hyp<T, EXACT> hyp;
      if (dposes) hyp.load(dposes + (size_t)hl * Hyp<T, EXACT>::STRIDE, KIND == VOTE_23_MATRIX);   // a list generated on the device
      else hyp.load(sp.v + hl * Hyp<T, EXACT>::STRIDE, KIND == VOTE_23_MATRIX);
      const int cnt = count_group_votes<T, KIND, EXACT>(hyp, vw, vc, vb, vnw, vnc, present, valid, thr33, cthr, cnl);
      mine += (lane == hl) ? cnt : 0;
    }
  }
  __shared__ double red[kBlock / 64][HB];
  if (lane < HB) red[wave][lane] = (double)mine;
  __syncthreads();
  collect_and_send<HB, 0, kBlock>(red, fin);
}

// P flags of one group as ONE store (8 bytes for fp32 / P = 4, 4 bytes for fp64 / P = 2): a thread owns P consecutive
// correspondences, so its shorts are contiguous; 2-byte scattered stores cost an order of magnitude more per byte.
__device__ __forceinline__ void store_mask_full(short* __restrict__ m, int64_t g, const bool (&v)[4]) {
  uint2 u;
  u.x = (unsigned int)v[0] | ((unsigned int)v[1] << 16);
  u.y = (unsigned int)v[2] | ((unsigned int)v[3] << 16);
  *reinterpret_cast<uint2*>(m + 4 * g) = u;
}
__device__ __forceinline__ void store_mask_full(short* __restrict__ m, int64_t g, const bool (&v)[2]) {
  *reinterpret_cast<unsigned int*>(m + 2 * g) = (unsigned int)v[0] | ((unsigned int)v[1] << 16);
}

template <class T> struct PoseArg { T v[12]; };  // one hypothesis by value (kernel argument): no H2D copy for a single pose

template <class T, int KIND, bool EXACT>
__global__ __launch_bounds__(kBlock) void mask_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const T* __restrict__ bv,
                                                      const T* __restrict__ nw, const T* __restrict__ nc, int64_t n,
                                                      PoseArg<T> pose, T thr33, T cthr, T cnl, short* __restrict__ m23,
                                                      short* __restrict__ m33, short* __restrict__ mnn, Finish fin) {
  constexpr int P = Pk<T>::P;
  typedef VoteMods<KIND> MD;
  Hyp<T, EXACT> hyp;
  hyp.load(pose.v, KIND == VOTE_23_MATRIX);
  int cnt = 0;
  const int64_t groups = (n + P - 1) / P, full = n / P;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += stride) {
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    load_group<T>(xw, g, n, vw);
    if (MD::need_xc) load_group<T>(xc, g, n, vc);
    if (MD::m23) load_group<T>(bv, g, n, vb);
    if (MD::mnn) { load_group<T>(nw, g, n, vnw); load_group<T>(nc, g, n, vnc); }
    bool f23[P], f33[P], fnn[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
      const bool present = (g * P + i) < n;
      const bool valid = present & (!MD::need_xc || !all_nan(vc[3 * i], vc[3 * i + 1], vc[3 * i + 2]));
      const T x = vw[3 * i], y = vw[3 * i + 1], z = vw[3 * i + 2];
      fnn[i] = MD::mnn ? (valid & hyp.innn(vnw[3 * i], vnw[3 * i + 1], vnw[3 * i + 2], vnc[3 * i], vnc[3 * i + 1], vnc[3 * i + 2], cnl)) : false;
      f33[i] = MD::m33 ? (valid & hyp.in33(x, y, z, vc[3 * i], vc[3 * i + 1], vc[3 * i + 2], thr33)) : false;
      f23[i] = MD::m23 ? (present & hyp.in23(x, y, z, vb[3 * i], vb[3 * i + 1], vb[3 * i + 2], cthr, KIND == VOTE_23_MATRIX)) : false;
      cnt += (int)fnn[i] + (int)f33[i] + (int)f23[i];
    }
    if (g < full) {
      if (MD::mnn) store_mask_full(mnn, g, fnn);
      if (MD::m33) store_mask_full(m33, g, f33);
      if (MD::m23) store_mask_full(m23, g, f23);
    } else {
#pragma unroll
      for (int i = 0; i < P; i++) {
        const int64_t idx = g * P + i;
        if (idx < n) {
          if (MD::mnn) mnn[idx] = fnn[i];
          if (MD::m33) m33[idx] = f33[i];
          if (MD::m23) m23[idx] = f23[i];
        }
      }
    }
  }
  // the vote total rides the same in-launch reduction + pinned-host publish as the normal equations (exact: integers < 2^53)
  double acc[1] = {(double)cnt};
  reduce_and_finish<1, kNeLd, 0, kBlock>(acc, fin);
}

// ================================================================================================
// K5 : one round of nl_shinji_kneip_ls + find_opt_cc  (AbsoluteOrientationNormal.hpp:484-505, :24-39)
// record (44): M23 (9) TW K | M33 (9) sigma | MNN (9) TL M | AA xx xy xz yy yz zz | bb (3) | pad
// ================================================================================================
struct NlParams { double c_opt[3], Cw[3], Cc[3], Rwc[9]; };

// one correspondence of the round: the three modality blocks, each switched on by its inlier flag (predicated, no branches).
// Arithmetic in the array dtype T (the reference's Tp, AbsoluteOrientationNormal.hpp:484-505, accumulates in Tp as well); the
// caller adds each group's partial sums s[] into fp64 accumulators, so only the per-term rounding of T remains (as in K1-K3).
template <class T>
struct NlConst { T c_opt[3], Cw[3], Cc[3], Rwc[9]; };
template <class T> __device__ __forceinline__ NlConst<T> nl_const(const NlParams& p) {
  NlConst<T> k;
#pragma unroll
  for (int i = 0; i < 3; i++) { k.c_opt[i] = (T)p.c_opt[i]; k.Cw[i] = (T)p.Cw[i]; k.Cc[i] = (T)p.Cc[i]; }
#pragma unroll
  for (int i = 0; i < 9; i++) k.Rwc[i] = (T)p.Rwc[i];
  return k;
}
template <class T>
__device__ __forceinline__ void nl_point(const NlConst<T>& prm, T x, T y, T z, bool on23, T w23v, T bx_, T by_, T bz_, bool on33, T w33v, T cx_, T cy_,
                                         T cz_, bool onnn, T wnnv, T nwx, T nwy, T nwz, T ncx, T ncy, T ncz, T (&acc)[44]) {
  {  // 2D-3D inliers: M23 and the find_opt_cc sums.  w = 0 switches the term off.
    const bool on = on23;
    const T w = on ? w23v : T(0);
    T ax = x - prm.c_opt[0], ay = y - prm.c_opt[1], az = z - prm.c_opt[2];
    const T n2 = ax * ax + ay * ay + az * az;
    const T inv = T(1) / sqrt(n2);
    ax = on ? ax * inv : T(0); ay = on ? ay * inv : T(0); az = on ? az * inv : T(0);  // selects: NaN * 0 must not reach the sums
    const T bx = on ? bx_ : T(0), by = on ? by_ : T(0), bz = on ? bz_ : T(0);
    acc[0] = fma(w * bx, ax, acc[0]); acc[1] = fma(w * bx, ay, acc[1]); acc[2] = fma(w * bx, az, acc[2]);
    acc[3] = fma(w * by, ax, acc[3]); acc[4] = fma(w * by, ay, acc[4]); acc[5] = fma(w * by, az, acc[5]);
    acc[6] = fma(w * bz, ax, acc[6]); acc[7] = fma(w * bz, ay, acc[7]); acc[8] = fma(w * bz, az, acc[8]);
    acc[9] += w; acc[10] += on ? T(1) : T(0);
    // find_opt_cc: v = Rwc * bv ; A = I - v v^T ; AA += A ; bb += A * Xw
    const T vx = prm.Rwc[0] * bx + prm.Rwc[1] * by + prm.Rwc[2] * bz;
    const T vy = prm.Rwc[3] * bx + prm.Rwc[4] * by + prm.Rwc[5] * bz;
    const T vz = prm.Rwc[6] * bx + prm.Rwc[7] * by + prm.Rwc[8] * bz;
    const T o = on ? T(1) : T(0);
    const T xo = on ? x : T(0), yo = on ? y : T(0), zo = on ? z : T(0);
    const T Axx = o - vx * vx, Axy = -vx * vy, Axz = -vx * vz, Ayy = o - vy * vy, Ayz = -vy * vz, Azz = o - vz * vz;
    acc[32] += Axx; acc[33] += Axy; acc[34] += Axz; acc[35] += Ayy; acc[36] += Ayz; acc[37] += Azz;
    acc[38] += Axx * xo + Axy * yo + Axz * zo;
    acc[39] += Axy * xo + Ayy * yo + Ayz * zo;
    acc[40] += Axz * xo + Ayz * yo + Azz * zo;
  }
  {  // 3D-3D inliers: centred covariance and sigma
    const bool on = on33;
    const T v = on ? w33v : T(0);
    const T ax = on ? x - prm.Cw[0] : T(0), ay = on ? y - prm.Cw[1] : T(0), az = on ? z - prm.Cw[2] : T(0);
    const T cx = on ? cx_ - prm.Cc[0] : T(0), cy = on ? cy_ - prm.Cc[1] : T(0), cz = on ? cz_ - prm.Cc[2] : T(0);
    acc[20] += v * (cx * cx + cy * cy + cz * cz);
    acc[11] = fma(v * cx, ax, acc[11]); acc[12] = fma(v * cx, ay, acc[12]); acc[13] = fma(v * cx, az, acc[13]);
    acc[14] = fma(v * cy, ax, acc[14]); acc[15] = fma(v * cy, ay, acc[15]); acc[16] = fma(v * cy, az, acc[16]);
    acc[17] = fma(v * cz, ax, acc[17]); acc[18] = fma(v * cz, ay, acc[18]); acc[19] = fma(v * cz, az, acc[19]);
  }
  {  // normal-normal inliers
    const bool on = onnn;
    const T l = on ? wnnv : T(0);
    const T ax = on ? nwx : T(0), ay = on ? nwy : T(0), az = on ? nwz : T(0);
    const T cx = on ? ncx : T(0), cy = on ? ncy : T(0), cz = on ? ncz : T(0);
    acc[21] = fma(l * cx, ax, acc[21]); acc[22] = fma(l * cx, ay, acc[22]); acc[23] = fma(l * cx, az, acc[23]);
    acc[24] = fma(l * cy, ax, acc[24]); acc[25] = fma(l * cy, ay, acc[25]); acc[26] = fma(l * cy, az, acc[26]);
    acc[27] = fma(l * cz, ax, acc[27]); acc[28] = fma(l * cz, ay, acc[28]); acc[29] = fma(l * cz, az, acc[29]);
    acc[30] += l; acc[31] += on ? T(1) : T(0);
  }
}

// generic form: any subset of the arrays / masks / weights, bounds-checked loads
template <class T, int BLK>
__global__ __launch_bounds__(BLK) void nl_round_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const T* __restrict__ bv,
                                                       const T* __restrict__ nw, const T* __restrict__ nc,
                                                       const short* __restrict__ k23, const short* __restrict__ k33,
                                                       const short* __restrict__ knn, const T* __restrict__ w23,
                                                       const T* __restrict__ w33, const T* __restrict__ wnn, int64_t n, NlParams prm,
                                                       Finish fin) {
  constexpr int P = Pk<T>::P;
  const NlConst<T> kc = nl_const<T>(prm);
  double acc[44];
#pragma unroll
  for (int k = 0; k < 44; k++) acc[k] = 0.0;
  const int64_t groups = (n + P - 1) / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  for (int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x; g < groups; g += stride) {
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    short a23[P], a33[P], ann[P];
    T u23[P], u33[P], unn[P];
    load_group<T>(xw, g, n, vw);
    load_mask_group(k23, g, n, a23);
    load_mask_group(k33, g, n, a33);
    if (knn) load_mask_group(knn, g, n, ann);
    if (w23) load_weight_group(w23, g, n, u23);
    if (w33) load_weight_group(w33, g, n, u33);
    if (wnn) load_weight_group(wnn, g, n, unn);
    if (bv) load_group<T>(bv, g, n, vb);
    if (xc) load_group<T>(xc, g, n, vc);
    if (nw) { load_group<T>(nw, g, n, vnw); load_group<T>(nc, g, n, vnc); }
    T sg[44];
#pragma unroll
    for (int k = 0; k < 44; k++) sg[k] = T(0);
#pragma unroll
    for (int i = 0; i < P; i++) {
      nl_point<T>(kc, vw[3 * i], vw[3 * i + 1], vw[3 * i + 2], bv != nullptr && a23[i] == 1, w23 ? u23[i] : T(1), vb[3 * i], vb[3 * i + 1],
                  vb[3 * i + 2], xc != nullptr && a33[i] == 1, w33 ? u33[i] : T(1), vc[3 * i], vc[3 * i + 1], vc[3 * i + 2],
                  nw != nullptr && knn != nullptr && ann[i] == 1, wnn ? unn[i] : T(1), vnw[3 * i], vnw[3 * i + 1], vnw[3 * i + 2],
                  vnc[3 * i], vnc[3 * i + 1], vnc[3 * i + 2], sg);
    }
#pragma unroll
    for (int k = 0; k < 44; k++) acc[k] += (double)sg[k];
  }
  reduce_and_finish<44, kNlLd, 0, BLK>(acc, fin);
}

// the common case -- all five arrays and all three masks present, weights all or none -- without bounds checks or pointer tests in
// the loop, and with the next group's 15 vector loads in flight while the current group is reduced (as normal_eq_kernel does)
template <class T, int BLK, bool WEIGHT>
__global__ __launch_bounds__(BLK) void nl_round_full_kernel(const T* __restrict__ xw, const T* __restrict__ xc, const T* __restrict__ bv,
                                                            const T* __restrict__ nw, const T* __restrict__ nc,
                                                            const short* __restrict__ k23, const short* __restrict__ k33,
                                                            const short* __restrict__ knn, const T* __restrict__ w23,
                                                            const T* __restrict__ w33, const T* __restrict__ wnn, int64_t n, NlParams prm,
                                                            Finish fin) {
  constexpr int P = Pk<T>::P;
  typedef typename Pk<T>::V V;
  // software pipeline depth is bounded by registers (88 for the fp64 accumulators alone): without weights all five arrays of the
  // NEXT group are in flight during the arithmetic; with weights only the three arrays consumed first are, the normals (consumed
  // last) and the weights are loaded at the top of the iteration that uses them
  constexpr int NPRE = WEIGHT ? 3 : 5;
  const NlConst<T> kc = nl_const<T>(prm);
  double acc[44];
#pragma unroll
  for (int k = 0; k < 44; k++) acc[k] = 0.0;
  const int64_t full = n / P;
  const int64_t stride = (int64_t)gridDim.x * BLK;
  const V* __restrict__ a4[5] = {reinterpret_cast<const V*>(xw), reinterpret_cast<const V*>(xc), reinterpret_cast<const V*>(bv),
                                 reinterpret_cast<const V*>(nw), reinterpret_cast<const V*>(nc)};
  int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
  V cur[5][3];
  short m[3][P];
  if (g < full) {
#pragma unroll
    for (int a = 0; a < NPRE; a++) { cur[a][0] = a4[a][3 * g]; cur[a][1] = a4[a][3 * g + 1]; cur[a][2] = a4[a][3 * g + 2]; }
    load_mask_full(k23, g, m[0]); load_mask_full(k33, g, m[1]); load_mask_full(knn, g, m[2]);
  }
  while (g < full) {
    const int64_t gn = g + stride;
    const int64_t gl = gn < full ? gn : g;  // clamp: the last iteration re-reads its own (cached) group instead of branching
    V nxt[NPRE][3];
    short nm[3][P];
    T wv[3][P];
#pragma unroll
    for (int a = NPRE; a < 5; a++) { cur[a][0] = a4[a][3 * g]; cur[a][1] = a4[a][3 * g + 1]; cur[a][2] = a4[a][3 * g + 2]; }
    if (WEIGHT) { load_weight_full(w23, g, wv[0]); load_weight_full(w33, g, wv[1]); load_weight_full(wnn, g, wv[2]); }
#pragma unroll
    for (int a = 0; a < NPRE; a++) { nxt[a][0] = a4[a][3 * gl]; nxt[a][1] = a4[a][3 * gl + 1]; nxt[a][2] = a4[a][3 * gl + 2]; }
    load_mask_full(k23, gl, nm[0]); load_mask_full(k33, gl, nm[1]); load_mask_full(knn, gl, nm[2]);
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    unpack3(cur[0][0], cur[0][1], cur[0][2], vw);
    unpack3(cur[1][0], cur[1][1], cur[1][2], vc);
    unpack3(cur[2][0], cur[2][1], cur[2][2], vb);
    unpack3(cur[3][0], cur[3][1], cur[3][2], vnw);
    unpack3(cur[4][0], cur[4][1], cur[4][2], vnc);
    T sg[44];
#pragma unroll
    for (int k = 0; k < 44; k++) sg[k] = T(0);
#pragma unroll
    for (int i = 0; i < P; i++) {
      nl_point<T>(kc, vw[3 * i], vw[3 * i + 1], vw[3 * i + 2], m[0][i] == 1, WEIGHT ? wv[0][i] : T(1), vb[3 * i], vb[3 * i + 1], vb[3 * i + 2],
                  m[1][i] == 1, WEIGHT ? wv[1][i] : T(1), vc[3 * i], vc[3 * i + 1], vc[3 * i + 2], m[2][i] == 1, WEIGHT ? wv[2][i] : T(1),
                  vnw[3 * i], vnw[3 * i + 1], vnw[3 * i + 2], vnc[3 * i], vnc[3 * i + 1], vnc[3 * i + 2], sg);
    }
#pragma unroll
    for (int k = 0; k < 44; k++) acc[k] += (double)sg[k];
#pragma unroll
    for (int a = 0; a < NPRE; a++) { cur[a][0] = nxt[a][0]; cur[a][1] = nxt[a][1]; cur[a][2] = nxt[a][2]; }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
      for (int i = 0; i < P; i++) m[k][i] = nm[k][i];
    }
    g = gn;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && full * P < n) {  // leftover correspondences through the bounds-checked loaders
    T vw[3 * P], vc[3 * P], vb[3 * P], vnw[3 * P], vnc[3 * P];
    short a23[P], a33[P], ann[P];
    T u23[P], u33[P], unn[P];
    load_group<T>(xw, full, n, vw); load_group<T>(xc, full, n, vc); load_group<T>(bv, full, n, vb);
    load_group<T>(nw, full, n, vnw); load_group<T>(nc, full, n, vnc);
    load_mask_group(k23, full, n, a23); load_mask_group(k33, full, n, a33); load_mask_group(knn, full, n, ann);
    if (WEIGHT) { load_weight_group(w23, full, n, u23); load_weight_group(w33, full, n, u33); load_weight_group(wnn, full, n, unn); }
    T sg[44];
#pragma unroll
    for (int k = 0; k < 44; k++) sg[k] = T(0);
#pragma unroll
    for (int i = 0; i < P; i++) {
      nl_point<T>(kc, vw[3 * i], vw[3 * i + 1], vw[3 * i + 2], a23[i] == 1, WEIGHT ? u23[i] : T(1), vb[3 * i], vb[3 * i + 1], vb[3 * i + 2],
                  a33[i] == 1, WEIGHT ? u33[i] : T(1), vc[3 * i], vc[3 * i + 1], vc[3 * i + 2], ann[i] == 1, WEIGHT ? unn[i] : T(1),
                  vnw[3 * i], vnw[3 * i + 1], vnw[3 * i + 2], vnc[3 * i], vnc[3 * i + 1], vnc[3 * i + 2], sg);
    }
#pragma unroll
    for (int k = 0; k < 44; k++) acc[k] += (double)sg[k];
  }
  reduce_and_finish<44, kNlLd, 0, BLK>(acc, fin);
}

// ---- after a collective: copy the reduced record from HBM to the pinned host slot and raise the sequence word
template <class E>
__global__ void publish_kernel(const E* __restrict__ src, int count, E* __restrict__ h_dst, unsigned long long* __restrict__ h_flag,
                               unsigned long long seq) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) __hip_atomic_store(h_dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(h_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_publish_f64(const double* d_src, int count, double* h_dst, unsigned long long* h_flag, unsigned long long seq, hipStream_t s) {
  hipLaunchKernelGGL((publish_kernel<double>), dim3(1), dim3(64), 0, s, d_src, count, h_dst, h_flag, seq);
  return hipGetLastError();
}
hipError_t launch_publish_i32(const int* d_src, int count, int* h_dst, unsigned long long* h_flag, unsigned long long seq, hipStream_t s) {
  hipLaunchKernelGGL((publish_kernel<int>), dim3(1), dim3(256), 0, s, d_src, count, h_dst, h_flag, seq);
  return hipGetLastError();
}
// vote counters: publish to the host AND clear them for the next scoring launch (the counters are accumulated with atomics, so
// they must start at zero; clearing here saves a memset per launch)
__global__ void publish_votes_kernel(int* __restrict__ votes, int count, int* __restrict__ h_dst, unsigned long long* __restrict__ h_flag,
                                     unsigned long long seq) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    __hip_atomic_store(h_dst + i, votes[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    votes[i] = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(h_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// sharded variant: all-reduce(sum) of the counters across the ranks through the peers' mailboxes before publishing (one 8-byte
// word {count | step tag} per hypothesis and source rank; rank-ordered integer sums; bounded wait like p2p_allreduce32)
__global__ __launch_bounds__(1024) void publish_votes_p2p_kernel(int* __restrict__ votes, int count, const P2PDesc* __restrict__ desc,
                                                                 unsigned long long step, int* __restrict__ h_dst, int* __restrict__ h_status,
                                                                 unsigned long long* __restrict__ h_flag, unsigned long long seq) {
  const P2PDesc& D = *desc;
  const unsigned int tag = (unsigned int)(step % 0xFFFFFFFFull) + 1u;
  const size_t parity = (size_t)(step & 1ull);
  const size_t base = kP2PRecordWords + parity * kP2PMaxWorld * kMaxScoreH;
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    const unsigned long long word = ((unsigned long long)tag << 32) | (unsigned int)votes[i];
    for (int r = 0; r < D.world; r++)
      __hip_atomic_store(D.peer[r] + base + (size_t)D.rank * kMaxScoreH + i, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const unsigned long long* box = D.peer[D.rank] + base;
  const unsigned long long t0 = wall_clock64();
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    int total = 0;
    for (int r = 0; r < D.world; r++) {
      unsigned long long w;
      for (;;) {
        w = __hip_atomic_load(box + (size_t)r * kMaxScoreH + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned int)(w >> 32) == tag) break;
        if (wall_clock64() - t0 > 1000000000ull) { s_bad = 1; break; }
      }
      total += (int)(unsigned int)w;
    }
    __hip_atomic_store(h_dst + i, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    votes[i] = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(h_status, s_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(h_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
hipError_t launch_publish_votes_p2p(int* d_votes, int count, const P2PDesc* p2p, unsigned long long step, int* h_dst, int* h_status,
                                    unsigned long long* h_flag, unsigned long long seq, hipStream_t s) {
  hipLaunchKernelGGL(publish_votes_p2p_kernel, dim3(1), dim3(count > 256 ? 1024 : 256), 0, s, d_votes, count, p2p, step, h_dst, h_status, h_flag, seq);
  return hipGetLastError();
}
hipError_t launch_publish_votes(int* d_votes, int count, int* h_dst, unsigned long long* h_flag, unsigned long long seq, hipStream_t s) {
  hipLaunchKernelGGL(publish_votes_kernel, dim3(1), dim3(count > 256 ? 1024 : 256), 0, s, d_votes, count, h_dst, h_flag, seq);
  return hipGetLastError();
}

#ifdef RPE_STAMPS
}  // namespace rpe
// diagnostic build only: copy the stamp buffer to the host (after a stream synchronise) and clear it
extern "C" int rpe_debug_read_stamps(unsigned long long* out, int nwords) {
  if (nwords > 4096 * 16) nwords = 4096 * 16;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rpe::g_stamps), (size_t)nwords * 8) != hipSuccess) return -1;
  static unsigned long long zeros[4096 * 16];
  return hipMemcpyToSymbol(HIP_SYMBOL(rpe::g_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : -1;
}
namespace rpe {
#endif

// ================================================================================================
// launchers
// ================================================================================================
static inline int grid_for(int64_t n, int P, int max_blocks, int block = kBlock) {
  int64_t groups = (n + P - 1) / P;
  int64_t g = (groups + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}
// Reduction kernels: one partial record per workgroup, so fewer workgroups = a shorter tail; but a thread that owns two groups
// also issues twice the arithmetic, and since the wave stage became a reduce-scatter the arithmetic is what is left of the body.
// Measured on MI355X: 307 200 correspondences run 6 % faster with one group per thread (150 workgroups of 512: 8.5 us) than with
// 128 workgroups (9.2 us); from about half a million correspondences every thread gets two groups (>= 128 workgroups), and from
// 10 M up the cap of 2 workgroups per CU wins.
static inline int reduce_grid(int64_t n, int P, int max_blocks, int block) {
  const int64_t groups = (n + P - 1) / P;
  const int64_t one = (groups + block - 1) / block, two = (groups + 2 * (int64_t)block - 1) / (2 * (int64_t)block);
  int64_t g = two < 128 ? one : two;
  if (block <= 256 && one <= max_blocks) g = one;   // 256-thread workgroups (collecting stage): one group per thread while that is at most 2 workgroups per CU
  static const int force = getenv("RPE_REDUCE_GROUPS") ? atoi(getenv("RPE_REDUCE_GROUPS")) : 0;   // experiments: 1 / 2 groups per thread
  if (force == 1) g = one; else if (force == 2) g = two;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}
template <class T> static PoseK<T> make_pose(const double* p12) {
  PoseK<T> k;
  for (int i = 0; i < 9; i++) k.R[i] = (T)p12[i];
  for (int i = 0; i < 3; i++) k.t[i] = (T)p12[9 + i];
  return k;
}
static Finish make_finish(const ReduceTarget& rt) {
  Finish f;
  f.partials = rt.d_partials; f.ticket = rt.d_ticket; f.out_dev = rt.d_out; f.out_host = rt.h_out; f.seq = rt.seq;
  f.gn_pose = rt.gn_pose; f.gn = rt.gn;
  f.p2p = rt.p2p; f.p2p_step = rt.p2p_step;
  // default 2: the last workgroup reads all records in ONE batch of loads (measured against 0 = two batches and 1 = per-shard sums
  // first, profiles/r02_tail_timeline.jsonl: 7.9 / 8.1 / 8.7 us per launch at 307 200 points); RPE_TAIL overrides for experiments
  static const int env_tail = getenv("RPE_TAIL") ? atoi(getenv("RPE_TAIL")) : 2;
  f.tail = rt.tail >= 0 ? rt.tail : env_tail;
  f.rows = rt.rows > 0 ? rt.rows : 0;
  return f;
}
// Launch geometry of the reduction kernels.  The tail (arrival count + fixed-order sum of one record per workgroup)
// costs latency proportional to the number of workgroups, the body wants every CU busy: 512-thread workgroups, at
// most 2 per CU (512 records), is the measured sweet spot on MI355X from 307 200 correspondences up; rt.block /
// rt.max_blocks (RPE_BLOCK / RPE_MAX_BLOCKS) override it for experiments.
static inline int pick_block(const ReduceTarget& rt, bool allow_1024) {
  // with the collecting stage the cross-workgroup cost no longer grows with the number of workgroups, and 256-thread workgroups (one
  // wave per SIMD on a frame, two workgroups per CU beyond) win: 5.7 vs 6.3 us at 307 200 points, configs[3] cold 11.8 vs 13.0 us
  int b = rt.block > 0 ? rt.block : (rt.rows > 0 ? 256 : 512);
  if (b >= 1024 && allow_1024) return 1024;
  if (b >= 512) return 512;
  return 256;
}

template <class T, int KIND, int BLK>
static void normal_eq_launch(const DeviceArrays& A, int flags, const PoseK<double>& pose, const ReduceTarget& rt, hipStream_t s, hipEvent_t ev0,
                             hipEvent_t ev1) {
  const T* xw = (const T*)A.a[0];
  const T* b = (const T*)(KIND == KIND_BEARING ? A.a[2] : A.a[1]);
  const T* c = (const T*)A.a[4];
  const int mod = KIND == KIND_BEARING ? 0 : 1;
  const short* mask = (flags & F_USE_MASK) ? A.mask[mod] : nullptr;
  const T* weight = (flags & F_USE_WEIGHT) ? (const T*)A.weight[mod] : nullptr;
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks, BLK);
  const Finish fin = make_finish(rt);
  // timed launches (bench.py's roofline leg) go through hipExtLaunchKernelGGL: the two events then carry the dispatch's own begin / end
  // timestamps -- what rocprofv3 reports for the kernel -- instead of bracketing it with two marker packets (which adds their latency)
#define RPE_NE_LAUNCH(M, W)                                                                                                            \
  do {                                                                                                                                 \
    if (ev0 && ev1) hipExtLaunchKernelGGL((normal_eq_kernel<T, KIND, BLK, M, W>), dim3(G), dim3(BLK), 0, s, ev0, ev1, 0, xw, b, c, mask, weight, A.n, pose, fin); \
    else hipLaunchKernelGGL((normal_eq_kernel<T, KIND, BLK, M, W>), dim3(G), dim3(BLK), 0, s, xw, b, c, mask, weight, A.n, pose, fin);     \
  } while (0)
  if (mask && weight) RPE_NE_LAUNCH(true, true);
  else if (mask) RPE_NE_LAUNCH(true, false);
  else if (weight) RPE_NE_LAUNCH(false, true);
  else RPE_NE_LAUNCH(false, false);
#undef RPE_NE_LAUNCH
}
template <class T>
static hipError_t normal_eq_t(const DeviceArrays& A, int kind, int flags, const double* pose12, const ReduceTarget& rt, hipStream_t s,
                              hipEvent_t ev0, hipEvent_t ev1) {
  const PoseK<double> pose = make_pose<double>(pose12);
  const int blk = pick_block(rt, kind == KIND_P2P);
  if (kind == KIND_P2P) {
    if (blk == 1024) normal_eq_launch<T, KIND_P2P, 1024>(A, flags, pose, rt, s, ev0, ev1);
    else if (blk == 512) normal_eq_launch<T, KIND_P2P, 512>(A, flags, pose, rt, s, ev0, ev1);
    else normal_eq_launch<T, KIND_P2P, 256>(A, flags, pose, rt, s, ev0, ev1);
  } else if (kind == KIND_P2PLANE) {
    if (blk == 512) normal_eq_launch<T, KIND_P2PLANE, 512>(A, flags, pose, rt, s, ev0, ev1);
    else normal_eq_launch<T, KIND_P2PLANE, 256>(A, flags, pose, rt, s, ev0, ev1);
  } else {
    if (blk == 512) normal_eq_launch<T, KIND_BEARING, 512>(A, flags, pose, rt, s, ev0, ev1);
    else normal_eq_launch<T, KIND_BEARING, 256>(A, flags, pose, rt, s, ev0, ev1);
  }
  return hipGetLastError();
}
hipError_t launch_normal_eq(const DeviceArrays& A, int kind, int flags, const double* pose12, const ReduceTarget& rt, hipStream_t s,
                            hipEvent_t ev0, hipEvent_t ev1) {
  return A.dtype ? normal_eq_t<double>(A, kind, flags, pose12, rt, s, ev0, ev1) : normal_eq_t<float>(A, kind, flags, pose12, rt, s, ev0, ev1);
}

// workgroup size of the resident kernel and the largest grid that is resident at once (one workgroup per CU).  256-thread workgroups
// (two per CU) were measured and lose here -- 6.95-7.3 vs 6.0-6.7 us per step at 307 200 points, 46 vs 39 us at 10 M -- although
// they win for the one-launch kernels: twice the workgroups poll the control block and twice the granules cross the hop every iteration
static inline int resident_block() { return 512; }
static inline int resident_cap(int) { return 256; }
// resident form: ONE launch for up to max_iters iterations; ctl = the control block in fine-grained device memory, first_tag + i =
// tag of pose i (i = 1 ...), rt.seq + i = sequence value published with record i
template <class T, int KIND, int BLK>
static void resident_launch(const DeviceArrays& A, int flags, const unsigned long long* ctl, unsigned long long first_tag, int max_iters,
                            const ReduceTarget& rt, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
  const T* xw = (const T*)A.a[0];
  const T* b = (const T*)(KIND == KIND_BEARING ? A.a[2] : A.a[1]);
  const T* c = (const T*)A.a[4];
  const int mod = KIND == KIND_BEARING ? 0 : 1;
  const short* mask = (flags & F_USE_MASK) ? A.mask[mod] : nullptr;
  const T* weight = (flags & F_USE_WEIGHT) ? (const T*)A.weight[mod] : nullptr;
  const int cap = resident_cap(BLK);
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks < cap ? rt.max_blocks : cap, BLK);   // every workgroup resident at once: 8 waves per CU
  const int64_t groups = (A.n + Pk<T>::P - 1) / Pk<T>::P;
  const bool in_regs = (int64_t)G * BLK >= groups;
  Finish fin = make_finish(rt);
  constexpr int kMaxRows = 4 * (BLK / (KIND == KIND_P2P ? 17 : 29));   // up to 4 granules per collecting thread
  if (fin.rows > kMaxRows) fin.rows = kMaxRows;
  if (fin.rows < 1) fin.rows = 1;
#define RPE_RES_LAUNCH3(M, W, R, AU)                                                                                                         \
  do {                                                                                                                                       \
    if (ev0 && ev1) hipExtLaunchKernelGGL((normal_eq_resident_kernel<T, KIND, BLK, M, W, R, AU>), dim3(G), dim3(BLK), 0, s, ev0, ev1, 0, xw, b, c, mask, weight, A.n, ctl, first_tag, max_iters, fin); \
    else hipLaunchKernelGGL((normal_eq_resident_kernel<T, KIND, BLK, M, W, R, AU>), dim3(G), dim3(BLK), 0, s, xw, b, c, mask, weight, A.n, ctl, first_tag, max_iters, fin);     \
  } while (0)
#define RPE_RES_LAUNCH2(M, W, R) do { if (fin.gn != nullptr) RPE_RES_LAUNCH3(M, W, R, true); else RPE_RES_LAUNCH3(M, W, R, false); } while (0)
#define RPE_RES_LAUNCH(M, W) do { if (in_regs) RPE_RES_LAUNCH2(M, W, true); else RPE_RES_LAUNCH2(M, W, false); } while (0)
  if (mask && weight) RPE_RES_LAUNCH(true, true);
  else if (mask) RPE_RES_LAUNCH(true, false);
  else if (weight) RPE_RES_LAUNCH(false, true);
  else RPE_RES_LAUNCH(false, false);
#undef RPE_RES_LAUNCH
#undef RPE_RES_LAUNCH2
#undef RPE_RES_LAUNCH3
}
template <class T>
static hipError_t resident_t(const DeviceArrays& A, int kind, int flags, const unsigned long long* ctl, unsigned long long first_tag, int max_iters,
                             const ReduceTarget& rt, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
  // the two 3D-3D kinds only: the bearing residual's register footprint (fp64 normalisation, three Jacobian rows) leaves no room
  // for a resident group without spilling; it keeps one launch per iteration
  if (kind == KIND_P2P) resident_launch<T, KIND_P2P, 512>(A, flags, ctl, first_tag, max_iters, rt, s, ev0, ev1);
  else if (kind == KIND_P2PLANE) resident_launch<T, KIND_P2PLANE, 512>(A, flags, ctl, first_tag, max_iters, rt, s, ev0, ev1);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
// grid the resident kernel runs with, the number of sums per record, the longest run of workgroups one collecting workgroup can take,
// and the run length used unless the caller forces one: BLK / sums rows (one granule per collecting thread) times 1..4, aiming at <= 8 runs
void resident_geometry(const DeviceArrays& A, int kind, int max_blocks, int* grid, int* nacc, int* max_rows, int* rows_auto) {
  const int P = A.dtype ? 2 : 4;
  const int blk = resident_block(), cap = resident_cap(blk);
  *grid = reduce_grid(A.n, P, max_blocks < cap ? max_blocks : cap, blk);
  *nacc = kind == KIND_P2P ? 17 : 29;
  const int rgn = blk / *nacc;
  *max_rows = 4 * rgn;
  int mult = (*grid + rgn * 8 - 1) / (rgn * 8);
  if (mult > 4) mult = 4;
  if (mult < 1) mult = 1;
  *rows_auto = rgn * mult;
}
hipError_t launch_normal_eq_resident(const DeviceArrays& A, int kind, int flags, const unsigned long long* ctl, unsigned long long first_tag,
                                     int max_iters, const ReduceTarget& rt, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
  return A.dtype ? resident_t<double>(A, kind, flags, ctl, first_tag, max_iters, rt, s, ev0, ev1)
                 : resident_t<float>(A, kind, flags, ctl, first_tag, max_iters, rt, s, ev0, ev1);
}

hipError_t launch_icp_fused(const float* vmap, const float* nmap, int64_t n, const float* mv, const float* mn, const Camera& mcam,
                            const PoseF& M, float dist_sq, float cos_thr, int use_normals, int kind, const double* pose12, const ReduceTarget& rt,
                            hipStream_t s) {
  AssocParams P;
  P.mcam = mcam; P.M = M; P.dist_sq = dist_sq; P.cos_thr = cos_thr; P.use_normals = use_normals;
  const PoseK<double> pose = make_pose<double>(pose12);
  const Finish fin = make_finish(rt);
  // geometry: the body (dependent gathers + fp64 accumulation) is heavier than a streaming pass, so every thread gets ONE pixel
  // group and the grid covers the image (150 workgroups of 512 at 640 x 480: 17.4 us per round against 20.6 us with the 128
  // workgroups a streaming reduction of this size uses; measured, scripts/icp_sweep.sh); RPE_ICP_BLOCK / RPE_ICP_GRID override
  static const int env_blk = getenv("RPE_ICP_BLOCK") ? atoi(getenv("RPE_ICP_BLOCK")) : 0;
  static const int env_grid = getenv("RPE_ICP_GRID") ? atoi(getenv("RPE_ICP_GRID")) : 0;
  const int blk = env_blk == 256 || env_blk == 512 ? env_blk : pick_block(rt, false);
  const int64_t groups = (n + 3) / 4;
  int G = env_grid > 0 ? env_grid : rt.max_blocks;
  if ((int64_t)G > (groups + blk - 1) / blk) G = (int)((groups + blk - 1) / blk);
  if (G < 1) G = 1;
  if (blk == 512) {
    if (kind == KIND_P2P) hipLaunchKernelGGL((icp_fused_kernel<KIND_P2P, 512>), dim3(G), dim3(512), 0, s, vmap, nmap, n, mv, mn, P, pose, fin);
    else hipLaunchKernelGGL((icp_fused_kernel<KIND_P2PLANE, 512>), dim3(G), dim3(512), 0, s, vmap, nmap, n, mv, mn, P, pose, fin);
  } else {
    if (kind == KIND_P2P) hipLaunchKernelGGL((icp_fused_kernel<KIND_P2P, 256>), dim3(G), dim3(256), 0, s, vmap, nmap, n, mv, mn, P, pose, fin);
    else hipLaunchKernelGGL((icp_fused_kernel<KIND_P2PLANE, 256>), dim3(G), dim3(256), 0, s, vmap, nmap, n, mv, mn, P, pose, fin);
  }
  return hipGetLastError();
}

// resident ICP loop: grid / record geometry exactly as the resident normal-equation kernel's (pixels in groups of 4)
void icp_resident_geometry(int64_t n, int kind, int max_blocks, int* grid, int* nacc, int* max_rows, int* rows_auto) {
  DeviceArrays A{};
  A.n = n; A.dtype = 0;
  resident_geometry(A, kind, max_blocks, grid, nacc, max_rows, rows_auto);
}
hipError_t launch_icp_resident(const float* vmap, const float* nmap, int64_t n, const float* mv, const float* mn, const Camera& mcam, const PoseF& M,
                               float dist_sq, float cos_thr, int use_normals, int kind, const unsigned long long* ctl, unsigned long long first_tag,
                               int max_iters, const ReduceTarget& rt, hipStream_t s) {
  if (kind != KIND_P2P && kind != KIND_P2PLANE) return hipErrorInvalidValue;
  AssocParams P;
  P.mcam = mcam; P.M = M; P.dist_sq = dist_sq; P.cos_thr = cos_thr; P.use_normals = use_normals;
  constexpr int BLK = 512;
  const int cap = resident_cap(BLK);
  const int G = reduce_grid(n, 4, rt.max_blocks < cap ? rt.max_blocks : cap, BLK);
  const int64_t groups = (n + 3) / 4;
  const bool in_regs = (int64_t)G * BLK >= groups;
  Finish fin = make_finish(rt);
  const int max_rows = 4 * (BLK / (kind == KIND_P2P ? 17 : 29));
  if (fin.rows > max_rows) fin.rows = max_rows;
  if (fin.rows < 1) fin.rows = 1;
#define RPE_ICP_RES2(K, R, AU) hipLaunchKernelGGL((icp_resident_kernel<K, BLK, R, AU>), dim3(G), dim3(BLK), 0, s, vmap, nmap, n, mv, mn, P, ctl, first_tag, max_iters, fin)
#define RPE_ICP_RES(K, R) do { if (fin.gn != nullptr) RPE_ICP_RES2(K, R, true); else RPE_ICP_RES2(K, R, false); } while (0)
  if (kind == KIND_P2P) { if (in_regs) RPE_ICP_RES(KIND_P2P, true); else RPE_ICP_RES(KIND_P2P, false); }
  else { if (in_regs) RPE_ICP_RES(KIND_P2PLANE, true); else RPE_ICP_RES(KIND_P2PLANE, false); }
#undef RPE_ICP_RES
#undef RPE_ICP_RES2
  return hipGetLastError();
}

template <class T, int TERMS, int BLK>
static void joint_launch_b(const DeviceArrays& A, int flags, const PoseK<double>& pose, const JointParams& prm, const ReduceTarget& rt, hipStream_t s);
template <class T, int TERMS>
static void joint_launch(const DeviceArrays& A, int flags, const PoseK<double>& pose, const JointParams& prm, const ReduceTarget& rt, hipStream_t s) {
  static const int env_blk = getenv("RPE_JOINT_BLOCK") ? atoi(getenv("RPE_JOINT_BLOCK")) : 0;
  // register-heavy kernel (up to three residual kinds, 29 fp64 accumulators): frames of the 640x480 class run 20 % faster with
  // 256-thread workgroups (one wave per SIMD, more workgroups in flight: 27.9 us vs 35.6 us at 307200), streaming sizes slightly
  // faster with 512 (10 M: 205 us vs 217 us)
  const int blk = env_blk == 256 || env_blk == 512 ? env_blk : (rt.block == 256 || rt.block == 512 ? rt.block : (A.n <= 2000000 ? 256 : 512));
  if (blk == 256) joint_launch_b<T, TERMS, 256>(A, flags, pose, prm, rt, s);
  else joint_launch_b<T, TERMS, 512>(A, flags, pose, prm, rt, s);
}
template <class T, int TERMS, int BLK>
static void joint_launch_b(const DeviceArrays& A, int flags, const PoseK<double>& pose, const JointParams& prm, const ReduceTarget& rt, hipStream_t s) {
  const bool um = (flags & F_USE_MASK) != 0, uw = (flags & F_USE_WEIGHT) != 0;
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks, BLK);
  hipLaunchKernelGGL((normal_eq_joint_kernel<T, TERMS, BLK>), dim3(G), dim3(BLK), 0, s, (const T*)A.a[0], (const T*)A.a[1], (const T*)A.a[2],
                     (const T*)A.a[3], (const T*)A.a[4], um ? (const short*)A.mask[0] : nullptr, um ? (const short*)A.mask[1] : nullptr,
                     um ? (const short*)A.mask[2] : nullptr, uw ? (const T*)A.weight[0] : nullptr, uw ? (const T*)A.weight[1] : nullptr,
                     uw ? (const T*)A.weight[2] : nullptr, A.n, pose, prm, make_finish(rt));
}
template <class T>
static hipError_t joint_t(const DeviceArrays& A, int terms, int flags, const double* pose12, const double* scale4, const int* robust4,
                          const double* robust_k4, const ReduceTarget& rt, hipStream_t s) {
  const PoseK<double> pose = make_pose<double>(pose12);
  JointParams prm;
  for (int k = 0; k < 4; k++) { prm.scale[k] = scale4[k]; prm.robust[k] = robust4[k]; prm.robust_k[k] = robust_k4[k]; }
  switch (terms) {
#define RPE_JOINT_CASE(M) case M: joint_launch<T, M>(A, flags, pose, prm, rt, s); break;
    RPE_JOINT_CASE(1) RPE_JOINT_CASE(2) RPE_JOINT_CASE(4) RPE_JOINT_CASE(8) RPE_JOINT_CASE(5) RPE_JOINT_CASE(6) RPE_JOINT_CASE(9)
    RPE_JOINT_CASE(10) RPE_JOINT_CASE(12) RPE_JOINT_CASE(13) RPE_JOINT_CASE(14)
#undef RPE_JOINT_CASE
    default: return hipErrorInvalidValue;  // empty set, or point-to-point together with point-to-plane
  }
  return hipGetLastError();
}
hipError_t launch_normal_eq_joint(const DeviceArrays& A, int terms, int flags, const double* pose12, const double* scale4, const int* robust4,
                                  const double* robust_k4, const ReduceTarget& rt, hipStream_t s) {
  return A.dtype ? joint_t<double>(A, terms, flags, pose12, scale4, robust4, robust_k4, rt, s)
                 : joint_t<float>(A, terms, flags, pose12, scale4, robust4, robust_k4, rt, s);
}

template <class T, int BLK>
static void moments_launch(const DeviceArrays& A, int flags, const ReduceTarget& rt, hipStream_t s) {
  const short* mask = (flags & F_USE_MASK) ? A.mask[1] : nullptr;
  const T* weight = (flags & F_USE_WEIGHT) ? (const T*)A.weight[1] : nullptr;
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks, BLK);
  const Finish fin = make_finish(rt);
  const int skip = (flags & F_SKIP_INVALID) ? 1 : 0;
  const T* xw = (const T*)A.a[0];
  const T* xc = (const T*)A.a[1];
  if (mask && weight) hipLaunchKernelGGL((moments_kernel<T, BLK, true, true>), dim3(G), dim3(BLK), 0, s, xw, xc, mask, weight, A.n, skip, fin);
  else if (mask) hipLaunchKernelGGL((moments_kernel<T, BLK, true, false>), dim3(G), dim3(BLK), 0, s, xw, xc, mask, weight, A.n, skip, fin);
  else if (weight) hipLaunchKernelGGL((moments_kernel<T, BLK, false, true>), dim3(G), dim3(BLK), 0, s, xw, xc, mask, weight, A.n, skip, fin);
  else hipLaunchKernelGGL((moments_kernel<T, BLK, false, false>), dim3(G), dim3(BLK), 0, s, xw, xc, mask, weight, A.n, skip, fin);
}
template <class T>
static hipError_t moments_t(const DeviceArrays& A, int flags, const ReduceTarget& rt, hipStream_t s) {
  const int blk = pick_block(rt, true);
  if (blk == 1024) moments_launch<T, 1024>(A, flags, rt, s);
  else if (blk == 512) moments_launch<T, 512>(A, flags, rt, s);
  else moments_launch<T, 256>(A, flags, rt, s);
  return hipGetLastError();
}
hipError_t launch_moments(const DeviceArrays& A, int flags, const ReduceTarget& rt, hipStream_t s) {
  return A.dtype ? moments_t<double>(A, flags, rt, s) : moments_t<float>(A, flags, rt, s);
}

template <class T, int KIND, bool EXACT>
static void score_launch(const DeviceArrays& A, const void* d_poses, int H, const double* thr, int* d_votes, int G, hipStream_t s) {
  // enough workgroups for ~8 per CU: split the hypothesis list (in multiples of 64) over blockIdx.y when one sweep is too few
  int gy = 1, hchunk = H;
  if (G < 2048 && H > 64) {
    const int chunks64 = (H + 63) / 64;
    gy = (2048 + G - 1) / G;
    if (gy > chunks64) gy = chunks64;
    hchunk = ((chunks64 + gy - 1) / gy) * 64;
    gy = (H + hchunk - 1) / hchunk;
  }
  hipLaunchKernelGGL((score_kernel<T, KIND, EXACT>), dim3(G, gy), dim3(kBlock), (size_t)hchunk * sizeof(int), s, (const T*)A.a[0], (const T*)A.a[1],
                     (const T*)A.a[2], (const T*)A.a[3], (const T*)A.a[4], A.n, (const T*)d_poses, H, hchunk, (T)thr[0], (T)thr[1], (T)thr[2], d_votes);
}
template <class T, int KIND, bool EXACT>
static void mask_launch(const DeviceArrays& A, const double* pose12, const double* thr, const ReduceTarget& rt, int G, hipStream_t s) {
  PoseArg<T> pa;
  for (int i = 0; i < 12; i++) pa.v[i] = (T)pose12[i];
  hipLaunchKernelGGL((mask_kernel<T, KIND, EXACT>), dim3(G), dim3(kBlock), 0, s, (const T*)A.a[0], (const T*)A.a[1], (const T*)A.a[2],
                     (const T*)A.a[3], (const T*)A.a[4], A.n, pa, (T)thr[0], (T)thr[1], (T)thr[2], A.mask[0], A.mask[1], A.mask[2],
                     make_finish(rt));
}
#define RPE_KIND_SWITCH(FN, T, EX, ...)                                    \
  switch (kind) {                                                          \
    case VOTE_33: FN<T, VOTE_33, EX>(__VA_ARGS__); break;                  \
    case VOTE_23: FN<T, VOTE_23, EX>(__VA_ARGS__); break;                  \
    case VOTE_33_23: FN<T, VOTE_33_23, EX>(__VA_ARGS__); break;            \
    case VOTE_NN_23: FN<T, VOTE_NN_23, EX>(__VA_ARGS__); break;            \
    case VOTE_NN_33: FN<T, VOTE_NN_33, EX>(__VA_ARGS__); break;            \
    case VOTE_NN_33_23: FN<T, VOTE_NN_33_23, EX>(__VA_ARGS__); break;      \
    case VOTE_23_MATRIX: FN<T, VOTE_23_MATRIX, EX>(__VA_ARGS__); break;    \
    default: return hipErrorInvalidValue;                                  \
  }

hipError_t launch_score(const DeviceArrays& A, int kind, int exact, const void* d_poses, int H, const double* thr3, int* d_votes,
                        int max_blocks, hipStream_t s) {
  if (H < 1 || H > kMaxScoreH) return hipErrorInvalidValue;
  // d_votes must be zero on entry: allocated zeroed, and re-zeroed by launch_publish_votes after every read-out
  if (A.dtype) {
    const int G = grid_for(A.n, 2, max_blocks);
    if (exact) { RPE_KIND_SWITCH(score_launch, double, true, A, d_poses, H, thr3, d_votes, G, s) }
    else { RPE_KIND_SWITCH(score_launch, double, false, A, d_poses, H, thr3, d_votes, G, s) }
  } else {
    const int G = grid_for(A.n, 4, max_blocks);
    if (exact) { RPE_KIND_SWITCH(score_launch, float, true, A, d_poses, H, thr3, d_votes, G, s) }
    else { RPE_KIND_SWITCH(score_launch, float, false, A, d_poses, H, thr3, d_votes, G, s) }
  }
  return hipGetLastError();
}
template <class T, int KIND, bool EXACT>
static void score_small_launch(const DeviceArrays& A, const void* h_poses, const void* d_poses, int H, const double* thr, const ReduceTarget& rt, int cap,
                               hipStream_t s) {
  constexpr int STRIDE = Hyp<T, EXACT>::STRIDE;
  const Finish fin = make_finish(rt);
  // up to a million correspondences the list is split over the 4 waves of a workgroup (RPE_SCORE_SPLIT = 1 | 2 | 4 overrides)
  static const int env_hs = getenv("RPE_SCORE_SPLIT") ? atoi(getenv("RPE_SCORE_SPLIT")) : 0;
  const int hs = (env_hs == 1 || env_hs == 2 || env_hs == 4) ? env_hs : (A.n <= (int64_t)1 << 20 ? 4 : 1);
  const int64_t tiles = ((A.n + Pk<T>::P - 1) / Pk<T>::P + kBlock / hs - 1) / (kBlock / hs);
  const int G = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, cap));
#define RPE_SMALL(HB)                                                                                                                        \
  do {                                                                                                                                       \
    SmallPoses<T, HB, STRIDE> sp;                                                                                                            \
    std::memset(&sp, 0, sizeof(sp));                                                                                                         \
    if (h_poses) std::memcpy(sp.v, h_poses, (size_t)H * STRIDE * sizeof(T));                                                                 \
    hipLaunchKernelGGL((score_small_kernel<T, KIND, EXACT, HB>), dim3(G), dim3(kBlock), 0, s, (const T*)A.a[0], (const T*)A.a[1], (const T*)A.a[2], \
                       (const T*)A.a[3], (const T*)A.a[4], A.n, sp, (const T*)d_poses, H, hs, (T)thr[0], (T)thr[1], (T)thr[2], fin);                                 \
  } while (0)
  if (H <= 16) RPE_SMALL(16); else RPE_SMALL(32);
#undef RPE_SMALL
}
// largest list the single-launch form takes for this dtype / mode (the hypotheses travel as a kernel argument of at most 2 KB)
int score_small_cap(int dtype, int exact) {
  const int bytes = (exact ? 8 : 12) * (dtype ? 8 : 4);
  return 32 * bytes <= 2048 ? 32 : 16;
}
// h_poses: H hypotheses staged in HOST memory in the scoring layout of `exact`, values of the array dtype (they travel in the kernel
// argument) -- or null and d_poses: the same list in HBM (a device-generated batch).  The vote counts arrive through rt (a collecting
// target): record[h] = votes of hypothesis h.
hipError_t launch_score_small(const DeviceArrays& A, int kind, int exact, const void* h_poses, const void* d_poses, int H, const double* thr3,
                              const ReduceTarget& rt, hipStream_t s) {
  if (H < 1 || H > score_small_cap(A.dtype, exact) || rt.rows < 1 || (!h_poses == !d_poses)) return hipErrorInvalidValue;
  const int cap = 2048;   // workgroups (grid-stride beyond)
  if (A.dtype) {
    if (exact) { RPE_KIND_SWITCH(score_small_launch, double, true, A, h_poses, d_poses, H, thr3, rt, cap, s) }
    else { RPE_KIND_SWITCH(score_small_launch, double, false, A, h_poses, d_poses, H, thr3, rt, cap, s) }
  } else {
    if (exact) { RPE_KIND_SWITCH(score_small_launch, float, true, A, h_poses, d_poses, H, thr3, rt, cap, s) }
    else { RPE_KIND_SWITCH(score_small_launch, float, false, A, h_poses, d_poses, H, thr3, rt, cap, s) }
  }
  return hipGetLastError();
}
hipError_t launch_mask(const DeviceArrays& A, int kind, int exact, const double* pose12, const double* thr3, const ReduceTarget& rt,
                       hipStream_t s) {
  const int cap = rt.max_blocks < 1024 ? 1024 : rt.max_blocks;  // streaming + stores: 4 workgroups of 256 per CU
  if (A.dtype) {
    const int G = grid_for(A.n, 2, cap);
    if (exact) { RPE_KIND_SWITCH(mask_launch, double, true, A, pose12, thr3, rt, G, s) }
    else { RPE_KIND_SWITCH(mask_launch, double, false, A, pose12, thr3, rt, G, s) }
  } else {
    const int G = grid_for(A.n, 4, cap);
    if (exact) { RPE_KIND_SWITCH(mask_launch, float, true, A, pose12, thr3, rt, G, s) }
    else { RPE_KIND_SWITCH(mask_launch, float, false, A, pose12, thr3, rt, G, s) }
  }
  return hipGetLastError();
}

template <class T, int BLK>
static void nl_round_launch(const DeviceArrays& A, const NlParams& prm, const ReduceTarget& rt, hipStream_t s) {
  const int G = reduce_grid(A.n, Pk<T>::P, rt.max_blocks, BLK);
  const bool all_arrays = A.a[0] && A.a[1] && A.a[2] && A.a[3] && A.a[4] && A.mask[0] && A.mask[1] && A.mask[2];
  const int nweights = (A.weight[0] != nullptr) + (A.weight[1] != nullptr) + (A.weight[2] != nullptr);
#define RPE_NL_ARGS (const T*)A.a[0], (const T*)A.a[1], (const T*)A.a[2], (const T*)A.a[3], (const T*)A.a[4], (const short*)A.mask[0],      \
                    (const short*)A.mask[1], (const short*)A.mask[2], (const T*)A.weight[0], (const T*)A.weight[1], (const T*)A.weight[2], A.n, prm, \
                    make_finish(rt)
  if (all_arrays && nweights == 0) hipLaunchKernelGGL((nl_round_full_kernel<T, BLK, false>), dim3(G), dim3(BLK), 0, s, RPE_NL_ARGS);
  else if (all_arrays && nweights == 3) hipLaunchKernelGGL((nl_round_full_kernel<T, BLK, true>), dim3(G), dim3(BLK), 0, s, RPE_NL_ARGS);
  else hipLaunchKernelGGL((nl_round_kernel<T, BLK>), dim3(G), dim3(BLK), 0, s, RPE_NL_ARGS);
#undef RPE_NL_ARGS
}
template <class T>
static hipError_t nl_round_t(const DeviceArrays& A, const double* params24, const ReduceTarget& rt, hipStream_t s) {
  NlParams prm;
  for (int i = 0; i < 3; i++) { prm.c_opt[i] = params24[i]; prm.Cw[i] = params24[3 + i]; prm.Cc[i] = params24[6 + i]; }
  for (int i = 0; i < 9; i++) prm.Rwc[i] = params24[9 + i];
  // 44 fp64 accumulators + a software-pipelined group need ~250 VGPRs: 256-thread workgroups (one wave per SIMD, no spills) beat
  // 512-thread ones here (10 M correspondences: 135 us vs 137 us unweighted, 158 us vs 172 us weighted; 307 200: 29 us vs 31 us)
  if (rt.block == 512) nl_round_launch<T, 512>(A, prm, rt, s);
  else nl_round_launch<T, 256>(A, prm, rt, s);
  return hipGetLastError();
}
hipError_t launch_nl_round(const DeviceArrays& A, const double* params24, const ReduceTarget& rt, hipStream_t s) {
  return A.dtype ? nl_round_t<double>(A, params24, rt, s) : nl_round_t<float>(A, params24, rt, s);
}

}  // namespace rpe
