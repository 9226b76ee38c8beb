// Shared device code of the gfx950 (MI355X, CDNA4) kernels of the RGB-D absolute-pose hot path: the 16-byte group loaders of
// the reference's native xyz-interleaved 3 x N arrays, the wave64 reduce-scatter, and the cross-workgroup stages of the
// in-launch two-stage reduction (collecting workgroups + host-side final sum; arrival counters + last workgroup).
// Hand-written HIP, wave64.  Included by every kernel unit (rpe_normal_eq.hip, rpe_icp.hip, rpe_joint.hip, rpe_score.hip,
// rpe_nl.hip); everything here is inline / template / static, one copy per unit.
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
#pragma once
#include "rpe_kernels.h"
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <hip/hip_ext.h>

namespace rpe {

// ---- diagnostic build only (-DRPE_STAMPS, scripts/tail_timeline.py): thread 0 of every workgroup stamps the 100 MHz constant clock at
// the
// phase boundaries of the reduction kernels into a buffer of its own (16 words per workgroup); no stamp exists in the product build.
#ifdef RPE_STAMPS
static __device__ unsigned long long g_stamps[4096 * 16];
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

enum { KIND_P2P = 0, KIND_P2PLANE = 1, KIND_BEARING = 2, KIND_REPROJ = 4 };   // = RPE_RES_* (3 is the normal-normal term of the joint kernel)
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

// A 16-byte vector in registers made opaque to the optimiser (no instruction is emitted): what is loaded as one 16-byte vector stays one
// -- no narrowing or re-splitting of the load, no repacking behind it -- up to this point.  pin16 is a data dependence only (it may
// move between the definition and the first use; the streaming loops put a scheduling barrier in front of it); pin16_here is
// volatile and keeps its place in program order (between the scheduling barriers of the rotating K5 pipeline).
template <class V> __device__ __forceinline__ void pin16(V& v) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  static_assert(sizeof(V) == 16, "16-byte vectors");
  u4 t = __builtin_bit_cast(u4, v);
  asm("" : "+v"(t));
  v = __builtin_bit_cast(V, t);
}
template <class V> __device__ __forceinline__ void pin16_here(V& v) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  static_assert(sizeof(V) == 16, "16-byte vectors");
  u4 t = __builtin_bit_cast(u4, v);
  asm volatile("" : "+v"(t));
  v = __builtin_bit_cast(V, t);
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
  if ((g + 1) * 4 <= n) { const float4 u = *reinterpret_cast<const float4*>(w + 4 * g); v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
      }
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

// Is any value of this group's loaded vectors NaN or infinite?  (One sum over everything the group loaded and one class test: a NaN or
// an infinity anywhere makes the sum NaN or infinite.)  A wave none of whose lanes says yes takes the CLEAN form of the group below --
// no NaN guards, no selects -- which is 17 % of the streaming loop's instructions; a wave with a NaN-marked column (the reference's
// "invalid measurement" idiom, AOPoseAdapter.hpp:147-152) takes the guarded form.
__device__ __forceinline__ bool vec_dirty(const float4& a) { const float s = (a.x + a.y) + (a.z + a.w); return !__builtin_isfinite(s); }
__device__ __forceinline__ bool vec_dirty(const double2& a) { const double s = a.x + a.y; return !__builtin_isfinite(s); }
__device__ __forceinline__ float4 vadd(const float4& a, const float4& b) { return float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ double2 vadd(const double2& a, const double2& b) { return double2{a.x + b.x, a.y + b.y}; }
template <class V> __device__ __forceinline__ bool group_dirty(const V& a0, const V& a1, const V& a2, const V& b0, const V& b1, const V& b2) {
  return vec_dirty(vadd(vadd(vadd(a0, a1), vadd(a2, b0)), vadd(b1, b2)));
}
template <class V> __device__ __forceinline__ bool group_dirty(const V& a0, const V& a1, const V& a2, const V& b0, const V& b1, const V& b2,
                                                               const V& c0, const V& c1, const V& c2) {
  return vec_dirty(vadd(vadd(vadd(a0, a1), vadd(a2, b0)), vadd(vadd(b1, b2), vadd(vadd(c0, c1), c2))));
}

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
#error "rpe_reduce.hpp: the fence-free cross-workgroup hand-off (reduce_and_finish) is only established for gfx942 / gfx950"
#endif
// wave64 sum by DPP cross-lane moves (no LDS traffic): butterfly inside each row of 16 lanes (quad_perm, row_ror),
// then row_bcast:15 / row_bcast:31 fold the four rows; the total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
  // The lanes a DPP move does not write keep the destination's OLD value.  With all rows enabled and a control whose source lane always
  // exists (quad_perm, row_ror, row_half_mirror) every lane is written, so the old value is never seen: it is left undefined (an
  // output-only empty asm: a register, no instruction) instead of the "v_mov_b32 v, 0" per half that a zero would cost -- 2 of the 5
  // instructions of an fp64 step of the wave reductions, which are bound by instruction issue.  The row_bcast
  // steps write two rows only and add the rest as zeros: they keep the zero.
  constexpr bool all_written = ROW_MASK == 0xf && (CTRL < 0x100 || (CTRL >= 0x121 && CTRL <= 0x12f) || CTRL == 0x140 || CTRL == 0x141);
  int old_lo = 0, old_hi = 0;
  if (all_written) { asm("" : "=v"(old_lo)); asm("" : "=v"(old_hi)); }
  const int lo = __builtin_amdgcn_update_dpp(old_lo, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(old_hi, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
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
// lower 32 lanes end with x(l) + x(l+32), upper with y(l-32) + y(l)
__device__ __forceinline__ double swap_add32(double x, double y) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap_add16(double x, double y) {   // even rows end with x's row-pair sums, odd rows with y's
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// lanes with upper = 0 keep x, the others y
template <int CTRL> __device__ __forceinline__ double pair_add(double x, double y, bool upper) {
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
  // cross-workgroup tail: 0 = all records summed by the last workgroup, 1 = per-shard sums first, 2 = 0 with one load batch
  int tail;
  // > 0: collecting workgroups + host-side final sum (collect_and_send / the resident kernel): cap on the run length
  int rows;
  // resident kernels: > 1 = run r is the workgroups r, r + stride, r + 2 stride ... (the workgroups of ONE XCD when stride = 8: they
  // are
  // dispatched to the XCDs round robin), collected by workgroup r; 0 / 1 = runs of `rows` consecutive workgroups
  int stride;
  // resident kernels: how long a workgroup waits for the host's next pose (100 MHz ticks) before it gives up
  unsigned long long pose_wait_ticks;
  unsigned long long fault_tag;         // test hook (0 = off): the LAST workgroup withholds its granules of the iteration with this tag
  double pivot_floor;                   // device-side 6x6 solves: relative pivot floor (rpe::pivot_floor, rpe/linalg.hpp)
  int solver;                           // autonomous resident loops: 1 = a solving workgroup (auto_solver_kernel, its own launch) plays the host
  // chained sharded steps (rpe_gn_steps_dist_device): the all-reduced run records of the step before (kRunSlots x kRunLd doubles; null
  // = first step: the pose is gn_pose as it stands) and where workgroup 0 leaves the pose this launch works with (for the next one)
  const double* chain_runs;
  double* chain_pose_out;
};
// what a collecting workgroup sends to the host in place of its run's sums when a granule of the run never arrived: a quiet NaN with a
// payload no arithmetic produces; the host then releases the grid and finishes the refinement with one launch per iteration
constexpr unsigned long long kLostMarker = 0x7ff8dead00c0ffeeull;

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
// ---- device-resident Gauss-Newton: solve H d = -g and T <- exp(d) T by ONE lane.
// One lane runs this between two iterations of a loop that otherwise takes ~4 us, so what counts is the length of the DEPENDENT chain,
// not the operation count (an instruction of one lane costs what an instruction of 64 costs, and the next dependent one waits ~8
// cycles):
// * right-looking elimination of the upper triangle (the Schur complement of an SPD matrix stays symmetric) with the right-hand side as
//    a seventh column: per pivot one reciprocal, then 5 - k independent factors and their independent updates -- six short steps
//    instead of the left-looking column recurrences' chains of dependent multiply-subtracts; same pivots, hence the same pivot test as
//    rpe/linalg.hpp solve_normal_eq6 (d_k > 1e-12 H_kk);
//  * the reciprocals from v_rcp_f64 + two Newton steps (4 dependent FMAs) instead of the ~12-instruction IEEE division sequence;
//  * back substitution column-wise (x_i = b_i / d_i, then every remaining b_r -= U_ri x_i independently);
// * exp(d) without any division, square root or sincos for |w| < 0.5 rad (every Gauss-Newton step in practice): sin(h)/h, cos(h) of the
//    half angle and (theta - sin theta)/theta^3 are even power series in the angle, evaluated by Horner in w.w; larger steps take the
//    closed forms.
// Fully unrolled so that every matrix entry is a register.  Results agree with the host's LDL^T + rpe::se3_exp to rounding (checked
// against the golden to 1e-13: tests/test_gpu_joint.py, rpe_debug_device_gn_update).
// Tried and rejected in round 4: the same solve spread over ONE WAVE (lane 8 i + j holds U[i][j] or b[i]; pivot by v_readlane, row k
// gathered by ds_bpermute, one fma per elimination step; pose update one entry per lane).  An instruction costs a wave its four
// cycles whether one lane is active or all, and of the ~600 instructions of this function only ~120 are the elimination: the wave
// form executes about as many (selects, gathers and their waits replace the multiply-adds).  Per iteration of the autonomous loop
// (scripts/device_loop_time.py, profiles/r04_device_solve_wave_ab.jsonl; wave form against this one): point-to-point at 307 200
// correspondences 6.14-6.18 against 5.78 us, point-to-plane 6.32-6.35 against 6.54, joint 9.11-9.15 against 9.19; one workgroup
// (1 000 correspondences) 4.31-4.38 / 4.41-4.49 / 5.83-5.89 against 4.02 / 4.60 / 6.01 -- no gain beyond the noise between boxes.
static __device__ __forceinline__ double rcp_newton(double d) {
  double y = __builtin_amdgcn_rcp(d);
  double e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  e = fma(-d, y, 1.0);
  return fma(y, e, y);
}
// MODE: how `tot` (LDS) holds the normal equations -- 0 = the packed record itself, 1 = the 17 structured point-to-point sums (expanded
// here, in registers: record_entry with compile-time indices costs a few adds, a separate expansion pass costs two barriers).
template <int MODE>
static __device__ __noinline__ bool gn_solve_update(const double* __restrict__ tot /* LDS */, double* __restrict__ pose /* LDS, 12,
                                                    in/out */, double* step_out, double rel_floor) {
  double U[6][6], b[6], inv[6], d[6];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
      for (int j = i; j < 6; j++) { U[i][j] = record_entry<MODE>(tot, k); k++; }
      b[i] = -record_entry<MODE>(tot, 21 + i);
    }
  }
  double diag0[6];
#pragma unroll
  for (int i = 0; i < 6; i++) diag0[i] = U[i][i];
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double piv = U[k][k];
    ok = ok && (piv > rel_floor * diag0[k]) && (piv < 1e300);   // relative pivot floor, as rpe/linalg.hpp solve_normal_eq6
    inv[k] = rcp_newton(piv);
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      const double f = U[k][i] * inv[k];   // = L_ik (symmetry of the Schur complement)
#pragma unroll
      for (int j = i; j < 6; j++) U[i][j] = fma(-f, U[k][j], U[i][j]);
      b[i] = fma(-f, b[k], b[i]);
    }
  }
  if (!ok) return false;
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    d[i] = b[i] * inv[i];
#pragma unroll
    for (int r = 0; r < i; r++) b[r] = fma(-U[r][i], d[i], b[r]);
  }
  double n2 = 0.0;
#pragma unroll
  for (int i = 0; i < 6; i++) { ok = ok && (d[i] == d[i]) && (d[i] < 1e300 && d[i] > -1e300); n2 += d[i] * d[i]; }
  if (!ok) return false;
  *step_out = sqrt(n2);
  // exp(d): rotation from the quaternion (cos(th/2), sin(th/2) w / th), V = I + c1 W + c2 W^2  (sophus/se3.hpp:321-342)
  const double wx = d[3], wy = d[4], wz = d[5];
  const double th2 = wx * wx + wy * wy + wz * wz;
  double imag, real, c1, c2;   // sin(th/2)/th, cos(th/2), (1 - cos th)/th^2, (th - sin th)/th^3
  if (th2 < 0.25) {
    const double x = 0.25 * th2;   // (th/2)^2 <= 1/16: the series below are at rounding level after 8 terms
    double S = -1.0 / 1307674368000.0, Cc = 1.0 / 87178291200.0, K = -1.0 / 355687428096000.0;
    S = fma(S, x, 1.0 / 6227020800.0); S = fma(S, x, -1.0 / 39916800.0); S = fma(S, x, 1.0 / 362880.0); S = fma(S, x, -1.0 / 5040.0);
    S = fma(S, x, 1.0 / 120.0); S = fma(S, x, -1.0 / 6.0); S = fma(S, x, 1.0);                                   // sin(h)/h
    Cc = fma(Cc, x, -1.0 / 479001600.0); Cc = fma(Cc, x, 1.0 / 3628800.0); Cc = fma(Cc, x, -1.0 / 40320.0); Cc = fma(Cc, x, 1.0 / 720.0);
    Cc = fma(Cc, x, -1.0 / 24.0); Cc = fma(Cc, x, 0.5); Cc = fma(Cc, -x, 1.0);                                  // cos(h)
    K = fma(K, th2, 1.0 / 1307674368000.0); K = fma(K, th2, -1.0 / 6227020800.0); K = fma(K, th2, 1.0 / 39916800.0);
    // (th - sin th)/th^3
    K = fma(K, th2, -1.0 / 362880.0); K = fma(K, th2, 1.0 / 5040.0); K = fma(K, th2, -1.0 / 120.0); K = fma(K, th2, 1.0 / 6.0);
    imag = 0.5 * S; real = Cc; c1 = 0.5 * S * S; c2 = K;
  } else {
    const double th = sqrt(th2);
    double sh, ch;
    sincos(0.5 * th, &sh, &ch);
    imag = sh / th; real = ch;
    c1 = (2.0 * sh * sh) / th2; c2 = (th - 2.0 * sh * ch) / (th2 * th);
  }
  double Rd[9], V[9];
  {
    const double qw = real, qx = imag * wx, qy = imag * wy, qz = imag * wz;
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    Rd[0] = 1 - (tyy + tzz); Rd[1] = txy - twz; Rd[2] = txz + twy;
    Rd[3] = txy + twz; Rd[4] = 1 - (txx + tzz); Rd[5] = tyz - twx;
    Rd[6] = txz - twy; Rd[7] = tyz + twx; Rd[8] = 1 - (txx + tyy);
  }
  {
    // V = I + c1 W + c2 W^2,  W^2 = w w^T - |w|^2 I
    const double dxx = c2 * wx * wx, dyy = c2 * wy * wy, dzz = c2 * wz * wz, dxy = c2 * wx * wy, dxz = c2 * wx * wz, dyz = c2 * wy * wz;
    const double diag = 1.0 - c2 * th2;
    V[0] = diag + dxx; V[1] = fma(-c1, wz, dxy); V[2] = fma(c1, wy, dxz);
    V[3] = fma(c1, wz, dxy); V[4] = diag + dyy; V[5] = fma(-c1, wx, dyz);
    V[6] = fma(-c1, wy, dxz); V[7] = fma(c1, wx, dyz); V[8] = diag + dzz;
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

// ---- chained sharded steps (rpe_dist.hip rpe_gn_steps_dist_device): the pose a launch works with comes from the launch BEFORE it --
// every workgroup adds the all-reduced run records of that step in run order (the order the host uses: identical sums in every
// workgroup and on every rank), solves and applies the exp-map to the pose of that step, all redundantly: no hop, no host.  Workgroup
// 0 leaves the pose (for the next launch) and the loop state.  Returns false -- uniformly -- where the solve refused the system
// (state.done is then set: the later launches of the chain return at once).
template <int MODE>
__device__ __forceinline__ bool chained_pose(const Finish& fin, PoseK<double>& pose) {
  __shared__ double c_tot[32];
  __shared__ double c_pose[12];
  __shared__ int c_ok;
  if (threadIdx.x < 12) c_pose[threadIdx.x] = fin.gn_pose[threadIdx.x];
  if (threadIdx.x < 32) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < kRunSlots; r++) t += fin.chain_runs[r * kRunLd + threadIdx.x];
    c_tot[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double step = 0.0;
    const bool ok = gn_solve_update<MODE>(c_tot, c_pose, &step, fin.pivot_floor);
    c_ok = ok ? 1 : 0;
    if (blockIdx.x == 0) {
      GnState* st = fin.gn;
      st->iters = st->iters + 1; st->step = step; st->cost = record_entry<MODE>(c_tot, 27);
      if (!ok) { st->status = 1; st->done = 1; }
      if (ok) { for (int k = 0; k < 12; k++) fin.chain_pose_out[k] = c_pose[k]; }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 9; k++) pose.R[k] = c_pose[k];
#pragma unroll
  for (int k = 0; k < 3; k++) pose.t[k] = c_pose[9 + k];
  return c_ok != 0;
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
// sc1 load -- the tag travels with the value, so neither a drain nor an arrival counter is needed (cdna_hip_programming.md Guideline
// 16,
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

// fixed-order column sums of `count` partial records, rows first, first + step, ...: thread (j, rg) takes every RG-th of them,
// U independent sc1 loads in flight, then the RG row-group sums are added in row-group order -> tot[j] (valid for threadIdx.x < LD
// after the caller's barrier).  The order depends on (first, step, count) only, never on which workgroup runs it.
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
        v[u] = r < count ? __hip_atomic_load(partials + (size_t)(first + r * step) * LD + j, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT) : 0.0;
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
// the collecting workgroup's read: thread (r, j) = (tid / NACC, tid % NACC) takes rows r, r + RGN, r + 2 RGN ... of the run (row 0 is
// the
// workgroup's own record, already in part[0]), up to CH granules in flight at once (buffer loads with the sc1 bit, aux 16, re-issued
// until every tag is this launch's), added in increasing row order into part[r][j].  Returns true if a granule never arrived (2 s).
template <int NACC, int BLK, int CH = 4>
__device__ __forceinline__ bool collect_rows(unsigned long long* __restrict__ gran, int G, int leader, int rows, unsigned long long tag,
                                             double (*part)[NACC], int step = 1,   // row k of the run = workgroup leader + k * step
                                             unsigned long long wait_ticks = 200000000ull) {   // 2 s of the 100 MHz clock
  constexpr int RGN = BLK / NACC;
  const int j = threadIdx.x % NACC, r = threadIdx.x / NACC;
  bool lost = false;
  if (CH == 1) {   // runs of at most RGN rows (the resident kernel): one granule per thread, polled by itself
    if (r >= 1 && r < rows) {
      const unsigned long long* src = gran + 2 * ((size_t)(leader + r * step) * NACC + j);
      const unsigned long long t0 = wall_clock64();
      granule_t q;
      for (unsigned int spins = 1;; spins++) {
        q = load_granule16(src);
        if ((((unsigned long long)q.w << 32) | q.z) == tag) break;
        if ((spins & 63u) == 0 && wall_clock64() - t0 > wait_ticks) { lost = true; break; }   // (2 s by default): a workgroup never delivered
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
          if (row < rows) q[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((leader + row * step) * NACC + j) * 16, 0, 16);
        }
#pragma unroll
        for (int u = 0; u < CH; u++) {
          const int row = k0 + u * RGN;
          if (row < rows && (((unsigned long long)q[u].w << 32) | q[u].z) != tag) pending = true;
        }
        if (!pending) break;
        if ((spins & 63u) == 0 && wall_clock64() - t0 > wait_ticks) { lost = true; break; }   // (2 s by default): a workgroup never delivered
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

// Fixed-order sum over the nr (<= RGN) rows of part[..][NACC] (LDS) by the threads j < NACC alone, behind the ONE workgroup barrier that
// says every granule has arrived: two interleaved chains (even rows, odd rows: the LDS reads are all in flight at once, the dependent
// fp64 additions are half as deep), added at the end.  Replaces sum_rows -- one thread per (block of six rows, value), a second
// workgroup barrier, the block sums -- on the critical path of every resident iteration and of every one-launch reduction: the second
// barrier and the second LDS round cost more than the 15 - 30 additions they spread (0.24 us from "all granules read" to "run record
// stored" at 150 workgroups, profiles/r05_resident_timeline.jsonl).  The order is a fixed function of nr.
template <int NACC, int RGN>
__device__ __forceinline__ double sum_rows_lane(const double (*part)[NACC], int nr, int j) {
  // the rows are read unconditionally and sixteen at once (a read behind a test of nr would wait for the one before it; rows past nr
  // hold older values: selected away), chunk after chunk while rows are left -- records of a few sums have hundreds of rows in the
  // array (RGN = BLK / NACC) of which a launch fills a few dozen
  constexpr int CH = RGN < 16 ? RGN : 16;
  double a = 0.0, b = 0.0;
  for (int k0 = 0; k0 < nr; k0 += CH) {
    double v[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) v[k] = part[k0 + k < RGN ? k0 + k : RGN - 1][j];
#pragma unroll
    for (int k = 0; k < CH; k += 2) {
      a += k0 + k < nr ? v[k] : 0.0;
      if (k + 1 < CH) b += k0 + k + 1 < nr ? v[k + 1] : 0.0;
    }
  }
  return a + b;
}

// Fixed-order sum over the nr (<= RGN) rows of part[..][NACC] (LDS), for every value j, by the whole collecting workgroup: blocks of
// six
// rows first -- one thread per (block, value), six LDS reads in flight -- a workgroup barrier, then the block sums in block order.  A
// collecting workgroup sits on the critical path of every launch and of every resident iteration; one thread per value walking up to 30
// rows one dependent LDS read after the other cost 0.5 us there (stamps, profiles/r03_resident_timeline.jsonl).  The order is a fixed
// function of (nr), so results stay bitwise reproducible.  Every thread of the workgroup must call it; threads < NACC get the totals.
template <int NACC, int BLK>
__device__ __forceinline__ double sum_rows(const double (*part)[NACC], int nr) {
  constexpr int RGN = BLK / NACC, RB = 6, NB = (RGN + RB - 1) / RB;
  static_assert(NB * NACC <= BLK, "one thread per (block of rows, value)");
  __shared__ double s_blk[NB][NACC];
  const int j = threadIdx.x % NACC, q = threadIdx.x / NACC;
  if (q < NB) {
    double v[RB];
#pragma unroll
    for (int u = 0; u < RB; u++) { const int k = q * RB + u; v[u] = k < nr ? part[k < RGN ? k : 0][j] : 0.0; }
    double t = v[0];
#pragma unroll
    for (int u = 1; u < RB; u++) t += v[u];
    s_blk[q][j] = t;
  }
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x < NACC) {
    t = s_blk[0][threadIdx.x];
#pragma unroll
    for (int b = 1; b < NB; b++) t += s_blk[b][threadIdx.x];
  }
  return t;
}

// ---- cross-workgroup stage for results the HOST consumes (single GPU): collecting workgroups + a host-side final sum.
// Workgroups are taken in runs of R; the first of a run collects: the others store their NACC sums as 16-byte granules {value, launch
// sequence number} (one sc1 store per lane; no drain, no arrival counter) and are done; the collecting workgroup reads its run's
// granules (collect_rows), adds the rows in a fixed order and sends the run's NACC sums to pinned host memory as tagged 16-byte pairs
// (slot 1 + run * NACC + j; slot 0 = a header pair from workgroup 0 that tells the host how many runs of how many sums to expect).
// The host adds the runs in run order and expands the record (rpe_receive.hip wait_collect).  R = BLK / NACC rows (one granule per
// collecting thread) times 1..4, aiming at <= 8 runs; longer still if the runs would not fit in ~512 pairs. One hand-off hop of ~1 us
// replaces the arrival counters + the last
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
  // grids of 32 workgroups and more: one run per XCD (run r = workgroups r, r + 8, ...: workgroups go to the XCDs round robin, so no
  // granule crosses an XCD boundary on its way to its collecting workgroup -- rpe_residuals.hpp run_shape), when a run then fits the
  // four granules per collecting thread
  const int S = fin.stride;
  // (S runs of NACC sums must also keep to the ~512 pairs of the consecutive-run shape: the pair buffer on the host is sized for that)
  const bool per_xcd = S > 1 && S * NACC <= 512 + NACC && G >= 4 * S && (G + S - 1) / S <= 4 * RGN && (G + S - 1) / S <= fin.rows;
  const int run = per_xcd ? (int)blockIdx.x % S : (int)blockIdx.x / R, leader = per_xcd ? run : run * R, step = per_xcd ? S : 1;
  const int nruns = per_xcd ? S : (G + R - 1) / R;
  if (threadIdx.x < NACC) {
    double own = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < NW; w++) own += red[w][threadIdx.x];
    if ((int)blockIdx.x != leader) store_granule16(gran + 2 * ((size_t)blockIdx.x * NACC + threadIdx.x), own, fin.seq);
    else c_part[0][threadIdx.x] = own;
  }
  RPE_STAMP(4);
  if ((int)blockIdx.x != leader) return;
  const int rows = per_xcd ? (G - run + S - 1) / S : min(R, G - leader);
  const bool lost = collect_rows<NACC, BLK>(gran, G, leader, rows, fin.seq, c_part, step);
  RPE_STAMP(7);
  if (__syncthreads_or(lost)) return;   // nothing published: the host reports the kernel as having finished without its result
  RPE_STAMP(8);
  if (fin.out_dev != nullptr) {
    // sharded step behind a collective (rpe_dist.hip): the run records stay on the DEVICE, kRunSlots slots of kRunLd doubles that the
    // collective then adds element by element across the ranks (every rank's host adds the all-reduced run records in run order
    // afterwards: no second hop on the device, no arrival counters).  Plain stores: the collective is a later kernel on the stream.
    // Slots of runs this grid does not have are cleared by workgroup 0 (the buffer is all-reduced in place: they hold the peers' sums
    // of the step before).
    if (threadIdx.x < NACC) fin.out_dev[run * kRunLd + threadIdx.x] = sum_rows_lane<NACC, RGN>(c_part, rows < RGN ? rows : RGN, threadIdx.x);
    if (blockIdx.x == 0) for (int i = nruns * kRunLd + (int)threadIdx.x; i < kRunSlots * kRunLd; i += BLK) fin.out_dev[i] = 0.0;
    return;
  }
  if (threadIdx.x < NACC) store_tagged_pair(fin.out_host, 1 + run * NACC + threadIdx.x,
      sum_rows_lane<NACC, RGN>(c_part, rows < RGN ? rows : RGN, threadIdx.x), fin.seq);
  if (blockIdx.x == 0 && threadIdx.x == 64) {   // header: runs | sums per run << 16 | record layout << 24
    const unsigned long long hdr = (unsigned long long)nruns | ((unsigned long long)NACC << 16) | ((unsigned long long)MODE << 24);
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
      if (threadIdx.x < LD) __hip_atomic_store(fin.partials + (size_t)(G + shard) * LD + threadIdx.x, tot[threadIdx.x],
          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
      // sharded loop: did every peer's record arrive?
      const bool delivered = !(LD == 32 && fin.p2p != nullptr && gn_rec[LD - 1] != 0.0);
      const bool ok = delivered && gn_solve_update<0>(gn_rec, gn_pose_s, &step, fin.pivot_floor);
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
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(fin.out_host + LD), fin.seq, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  if (fin.out_host) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RPE_STAMP(9);
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(fin.out_host + LD), fin.seq, __ATOMIC_RELAXED,
          __HIP_MEMORY_SCOPE_SYSTEM);
    }
    RPE_STAMP(10);
  }
}

// p = R x + t in fp64 (the pose stays fp64 everywhere: the residuals are cancelling differences)
template <class C>
__device__ __forceinline__ void transform(const PoseK<double>& T, C x, C y, C z, double& px, double& py, double& pz) {
  const double xd = x, yd = y, zd = z;
  px = fma(T.R[0], xd, fma(T.R[1], yd, fma(T.R[2], zd, T.t[0])));
  py = fma(T.R[3], xd, fma(T.R[4], yd, fma(T.R[5], zd, T.t[1])));
  pz = fma(T.R[6], xd, fma(T.R[7], yd, fma(T.R[8], zd, T.t[2])));
}

// ---- launch geometry shared by the units
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
  // 256-thread workgroups (collecting stage): one group per thread while that is at most 2 workgroups per CU
  if (block <= 256 && one <= max_blocks) g = one;
  // experiments: 1 / 2 groups per thread
  static const int force = getenv("RPE_REDUCE_GROUPS") ? atoi(getenv("RPE_REDUCE_GROUPS")) : 0;
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
  f.stride = rt.stride > 1 ? rt.stride : 0;
  f.pose_wait_ticks = rt.pose_wait_ticks; f.fault_tag = rt.fault_tag; f.pivot_floor = rt.pivot_floor;
  f.solver = rt.solver;
  f.chain_runs = rt.chain_runs; f.chain_pose_out = rt.chain_pose_out;
  return f;
}
// Launch geometry of the reduction kernels.  The tail (arrival count + fixed-order sum of one record per workgroup)
// costs latency proportional to the number of workgroups, the body wants every CU busy: 512-thread workgroups, at
// most 2 per CU (512 records), is the measured sweet spot on MI355X from 307 200 correspondences up; rt.block /
// rt.max_blocks (RPE_BLOCK / RPE_MAX_BLOCKS) override it for experiments.
static inline int pick_block(const ReduceTarget& rt) {
  // with the collecting stage the cross-workgroup cost no longer grows with the number of workgroups, and 256-thread workgroups (one
  // wave per SIMD on a frame, two workgroups per CU beyond) win: 5.7 vs 6.3 us at 307 200 points, configs[3] cold 11.8 vs 13.0 us.
  // (1024-thread workgroups existed until round 4: 128 registers per wave, every instance spilled 7-87 of them -- removed.)
  const int b = rt.block > 0 ? rt.block : (rt.rows > 0 ? 256 : 512);
  return b >= 512 ? 512 : 256;
}

}  // namespace rpe
