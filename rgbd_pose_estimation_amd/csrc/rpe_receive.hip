// Where a launch leaves its result and how the host receives it: the launch targets (host record + sequence word, collecting
// workgroups + host-side final sum, device record for a collective), the spins on pinned host memory (tagged pairs / flags; every
// wait watches the stream and ends with an error if the kernel ended without publishing), and the clean-first protocol that chooses
// between the CLEAN (no NaN guards) and the guarded flavour of the normal-equation kernels.
#include "rpe_host.hpp"
namespace rpeh {

// RPE_RESIDENT_STRIDE (experiments): runs of every stride-th workgroup; 0 / 1 = runs of consecutive workgroups.  Clamped to 2 .. 16: a
// stride is a number of RUNS, every run sends up to 44 sums to the host as tagged pairs, and the pinned pair buffer (h_big) and the
// autonomous loop's run records (kAutoMaxRunSums) are sized for at most ~16 runs of the widest record.
int run_stride_from_env() {
  const char* e = getenv("RPE_RESIDENT_STRIDE");
  if (!e) return 8;
  const int v = atoi(e);
  if (v <= 1) return 0;
  return v > 16 ? 16 : v;
}

rpe::ReduceTarget host_target(rpe_context* c) {
  rpe::ReduceTarget rt;
  rt.d_partials = c->d_partials; rt.d_ticket = c->d_ticket; rt.max_blocks = c->max_blocks; rt.block = c->block;
  rt.pivot_floor = rpe::pivot_floor(c->dtype == RPE_F64);
  rt.d_out = nullptr; rt.h_out = c->h_out; rt.seq = ++c->seq;
  c->collecting = false;
  return rt;
}
// host-consumed result of ONE launch on a single GPU: collecting workgroups + host-side final sum (rpe_reduce.hpp collect_and_send);
// wait_host then assembles the record in c->h_out.  RPE_COLLECT=0: the arrival-counter tail (as the device / collective targets use)
rpe::ReduceTarget collect_target(rpe_context* c) {
  rpe::ReduceTarget rt = host_target(c);
  static const bool on = !(getenv("RPE_COLLECT") && atoi(getenv("RPE_COLLECT")) == 0);
  static const int stride = run_stride_from_env();
  if (on) { rt.h_out = c->h_big; rt.rows = 1 << 20; rt.stride = stride; c->collecting = true; }
  return rt;
}
rpe::ReduceTarget device_target(rpe_context* c, double* d_out) {
  rpe::ReduceTarget rt;
  rt.d_partials = c->d_partials; rt.d_ticket = c->d_ticket; rt.max_blocks = c->max_blocks; rt.block = c->block;
  rt.pivot_floor = rpe::pivot_floor(c->dtype == RPE_F64);
  rt.d_out = d_out; rt.h_out = nullptr; rt.seq = 0;
  c->collecting = false;
  return rt;
}
// sharded step behind a collective: the collecting stage of the host-consumed launches, its run records left in device memory
// (kRunSlots x kRunLd doubles at c->d_out + 64) for the collective to add across the ranks -- one hop on the device, as on one GPU
rpe::ReduceTarget device_runs_target(rpe_context* c) {
  rpe::ReduceTarget rt = device_target(c, c->d_out + 64);
  rt.rows = 1 << 20; rt.stride = 8;
  if (rt.max_blocks > 256) rt.max_blocks = 256;   // at most kRunSlots = 8 runs (collect_and_send: one per XCD, or <= 8 of consecutive workgroups): RPE_MAX_BLOCKS cannot push a launch beyond the slots
  rt.seq = ++c->seq;   // the granules of the collecting stage carry the launch's sequence number as their tag
  return rt;
}
// Spin on the sequence word the kernel's last workgroup stores after the record (pinned, coherent host memory).
int wait_host(rpe_context* c, int ld) {
  if (c->collecting) { c->collecting = false; return wait_collect(c, ld); }
  volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(c->h_out + ld);
  const unsigned long long want = c->seq;
  for (unsigned long long spins = 0;; spins++) {
    if (__atomic_load_n(const_cast<unsigned long long*>(flag), __ATOMIC_ACQUIRE) == want) return RPE_OK;
    if ((spins & 0xFFFFF) == 0xFFFFF) {  // every ~1M polls: has the stream died?
      hipError_t q = hipStreamQuery(c->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail(RPE_ERR_HIP, "stream error while waiting for a kernel result: %s",
          hipGetErrorString(q));
      // (an autonomous loop's result comes from its solving workgroup on the second stream: the workers' kernel ends before it does)
      if (q == hipSuccess && c->stream2) { const hipError_t q2 = hipStreamQuery(c->stream2); if (q2 == hipErrorNotReady) continue; if (q2 != hipSuccess) (void)hipGetLastError(); }
      if (q == hipSuccess && __atomic_load_n(const_cast<unsigned long long*>(flag), __ATOMIC_ACQUIRE) != want)
        return fail(RPE_ERR_HIP, "kernel finished without publishing its result (sequence %llu)", want);
    }
  }
}

// Host-side final sum (resident loop): `grid` collecting workgroups each sent `nacc` pairs {value, seq}; add them in run order as they
// arrive (a fixed order).  Records that are not there yet are waited for one by one, so the summation overlaps the arrival of the
// later ones.
int wait_host_partials(rpe_context* c, int grid, int nacc, double* totals, int first_slot, bool resident) {
  unsigned long long* pairs = reinterpret_cast<unsigned long long*>(c->h_big) + 2 * (size_t)first_slot;
  const unsigned long long want = c->seq;
  for (int k = 0; k < nacc; k++) totals[k] = 0.0;
  unsigned long long spins = 0;
  bool lost = false;
  // All tags first, in branch-free sweeps (independent loads: the cache misses on lines the device has just written overlap), then the
  // sums in run order -- 0.1 us per resident step faster than waiting pair by pair (four A/B alternations,
  // scripts/env_ab_r03.py);
  // RPE_HOST_SWEEP=0 selects the pair-by-pair wait.
  static const int sweep = getenv("RPE_HOST_SWEEP") ? atoi(getenv("RPE_HOST_SWEEP")) : 1;
  if (sweep) {
    const int total = grid * nacc;
    for (;;) {
      unsigned long long missing = 0;
      // independent loads: the misses overlap
      for (int i = 0; i < total; i++) missing |= __atomic_load_n(pairs + 2 * (size_t)i + 1, __ATOMIC_RELAXED) ^ want;
      if (!missing) break;
      if ((++spins & 0x3FFFF) == 0) {
        hipError_t q = hipStreamQuery(c->stream);
        if (q != hipSuccess && q != hipErrorNotReady) return fail(RPE_ERR_HIP, "stream error while waiting for a kernel result: %s",
            hipGetErrorString(q));
        if (q == hipSuccess) {
          missing = 0;
          for (int i = 0; i < total; i++) missing |= __atomic_load_n(pairs + 2 * (size_t)i + 1, __ATOMIC_RELAXED) ^ want;
          if (missing) { (void)fail(RPE_ERR_HIP, "the kernel ended without publishing record %llu", want);
              return resident ? kResidentLost : RPE_ERR_HIP; }
        }
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  for (int g = 0; g < grid; g++) {
    unsigned long long* rec = pairs + 2 * (size_t)g * nacc;
    for (int k = nacc - 1; k >= 0; k--) {
      while (__atomic_load_n(rec + 2 * k + 1, __ATOMIC_ACQUIRE) != want) {
        if ((++spins & 0xFFFFF) == 0) {
          hipError_t q = hipStreamQuery(c->stream);
          if (q != hipSuccess && q != hipErrorNotReady) return fail(RPE_ERR_HIP, "stream error while waiting for a kernel result: %s",
              hipGetErrorString(q));
          if (q == hipSuccess && __atomic_load_n(rec + 2 * k + 1, __ATOMIC_ACQUIRE) != want) {
            (void)fail(RPE_ERR_HIP, "the kernel ended without publishing record %llu (run %d)", want, g);
            return resident ? kResidentLost : RPE_ERR_HIP;
          }
        }
      }
    }
    for (int k = 0; k < nacc; k++) {
      double v;
      const unsigned long long w = __atomic_load_n(rec + 2 * k, __ATOMIC_RELAXED);
      if (w == rpe::kResidentLostMarker) lost = true;   // this run's collecting workgroup never got one of its granules
      std::memcpy(&v, &w, 8);
      totals[k] += v;
    }
  }
  if (lost) {
    (void)fail(RPE_ERR_HIP, "a workgroup's sums never reached its collecting workgroup (record %llu)", want);
    return resident ? kResidentLost : RPE_ERR_HIP;
  }
  return RPE_OK;
}
// the 17 structured point-to-point sums -> the packed record (same map as record_entry<1> in rpe_reduce.hpp)
void expand_p2p17(const double* t, double* ne) {
  for (int i = 0; i < 32; i++) ne[i] = 0.0;
  const double nn = t[0], Sx = t[1], Sy = t[2], Sz = t[3], xx = t[4], xy = t[5], xz = t[6], yy = t[7], yz = t[8], zz = t[9];
  ne[0] = ne[6] = ne[11] = ne[28] = nn;
  ne[4] = Sz; ne[5] = -Sy; ne[8] = -Sz; ne[10] = Sx; ne[12] = Sy; ne[13] = -Sx;
  ne[15] = yy + zz; ne[16] = -xy; ne[17] = -xz; ne[18] = xx + zz; ne[19] = -yz; ne[20] = xx + yy;
  for (int i = 21; i <= 26; i++) ne[i] = t[i - 11];
  ne[27] = t[16];
}

// Result of a collecting launch (collect_target): the header pair says how many run records of how many sums to expect; add them in
// run order and lay the record out in c->h_out as the flag path would have left it.
int wait_collect(rpe_context* c, int ld) {
  unsigned long long* pairs = reinterpret_cast<unsigned long long*>(c->h_big);
  const unsigned long long want = c->seq;
  for (unsigned long long spins = 1; __atomic_load_n(pairs + 1, __ATOMIC_ACQUIRE) != want; spins++) {
    if ((spins & 0xFFFFF) == 0) {
      hipError_t q = hipStreamQuery(c->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail(RPE_ERR_HIP, "stream error while waiting for a kernel result: %s",
          hipGetErrorString(q));
      if (q == hipSuccess && __atomic_load_n(pairs + 1, __ATOMIC_ACQUIRE) != want)
        return fail(RPE_ERR_HIP,
            "kernel finished without publishing its result (sequence %llu; header %llx %llu, first pair %llx %llu)", want,
                    pairs[0], pairs[1], pairs[2], pairs[3]);
    }
  }
  const unsigned long long hdr = __atomic_load_n(pairs, __ATOMIC_RELAXED);
  const int runs = (int)(hdr & 0xFFFF), nacc = (int)((hdr >> 16) & 0xFF), mode = (int)((hdr >> 24) & 0xFF);
  if (runs < 1 || nacc < 1 || nacc > 64 || nacc > ld || (size_t)(1 + runs * nacc) > c->h_big_pairs) return fail(RPE_ERR_HIP,
      "malformed result header (%d runs of %d sums)", runs, nacc);
  double tot[64];
  int rc = wait_host_partials(c, runs, nacc, tot, 1);
  if (rc) return rc;
  if (mode == 1) expand_p2p17(tot, c->h_out);
  else { for (int i = 0; i < ld; i++) c->h_out[i] = i < nacc ? tot[i] : 0.0; }
  return RPE_OK;
}

// same spin on an arbitrary pinned sequence word
int wait_flag(rpe_context* c, unsigned long long* flag, unsigned long long want) {
  for (unsigned long long spins = 0;; spins++) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == want) return RPE_OK;
    if ((spins & 0xFFFFF) == 0xFFFFF) {
      hipError_t q = hipStreamQuery(c->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail(RPE_ERR_HIP, "stream error while waiting for a kernel result: %s",
          hipGetErrorString(q));
      if (q == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) return fail(RPE_ERR_HIP,
          "kernel finished without publishing its result (sequence %llu)", want);
    }
  }
}

// ---- CLEAN-first protocol.  The CLEAN flavour of a normal-equation kernel carries no NaN guards (17 % fewer instructions per group,
// rpe_residuals.hpp pair_group); it is exact for arrays whose values are all finite, and for any other content at least one sum of its
// record is non-finite (a NaN or an infinity anywhere multiplies into the sums even at weight 0).  So a launch whose record the host
// reads anyway takes the CLEAN flavour first, looks at the record, and repeats the launch in the guarded flavour if it is not finite --
// one wasted launch per upload of NaN-marked arrays, after which the arrays are known to need the guards.  A launch whose record is
// consumed on the device (collectives, the autonomous loops) takes the CLEAN flavour only over arrays already verified.
unsigned kind_slot_bits(int kind) {
  switch (kind) {
    case RPE_RES_P2P: return (1u << RPE_XW) | (1u << RPE_XC);
    case RPE_RES_P2PLANE: return (1u << RPE_XW) | (1u << RPE_XC) | (1u << RPE_NC);
    case RPE_RES_BEARING: case RPE_RES_REPROJ: return (1u << RPE_XW) | (1u << RPE_BV);
    case RPE_RES_NORMAL: return (1u << RPE_NW) | (1u << RPE_NC);
  }
  return 0;
}
bool take_clean(const rpe_context* c, int kind, bool host_verifies) {
  const unsigned bits = kind_slot_bits(kind);
  if (c->guard_always || bits == 0 || c->dtype == RPE_F64) return false;   // (the CLEAN flavours exist for fp32 arrays)
  bool all_verified = true;
  for (int s = 0; s < RPE_NUM_ARRAYS; s++) if (bits & (1u << s)) {
    if (c->arr_state[s] == kArrDirty) return false;
    if (c->arr_state[s] != kArrClean) all_verified = false;
  }
  return host_verifies || all_verified;
}
bool record_finite(const double* rec, int count) {
  double s = 0.0;
  for (int i = 0; i < count; i++) s += rec[i];
  return std::isfinite(s);
}
// what a CLEAN launch's record said about the arrays of `kind`.  Caller-owned (bound) arrays are never promoted: they may change
// between calls without the context hearing of it.
void note_clean_launch(rpe_context* c, int kind, bool finite) {
  const unsigned bits = kind_slot_bits(kind);
  for (int s = 0; s < RPE_NUM_ARRAYS; s++) if (bits & (1u << s)) {
    if (!finite) c->arr_state[s] = kArrDirty;
    else if (!c->arr_bound[s]) c->arr_state[s] = kArrClean;
  }
}
// ... for a SET of residual kinds (the joint kernels; bits = 1 << kind): CLEAN only if every kind of the set may take it
bool take_clean_terms(const rpe_context* c, int bits, bool host_verifies) {
  if (!rpe::joint_has_clean_flavour(c->dtype == RPE_F64 ? 1 : 0, bits)) return false;   // (the launch would run guarded: its finite record says nothing about the arrays)
  for (int k = 0; k <= 4; k++) if ((bits & (1 << k)) && !take_clean(c, k, host_verifies)) return false;
  return bits != 0;
}
void note_clean_terms(rpe_context* c, int bits, bool finite) {
  for (int k = 0; k <= 4; k++) if (bits & (1 << k)) note_clean_launch(c, k, finite);
}
void arrays_changed(rpe_context* c, int slot, bool bound) { c->arr_state[slot] = kArrUnknown; c->arr_bound[slot] = bound; }

int kind_arrays(rpe_context* c, int kind) {
  switch (kind) {
    case RPE_RES_P2P: return need_arrays(c, {RPE_XW, RPE_XC});
    case RPE_RES_P2PLANE: return need_arrays(c, {RPE_XW, RPE_XC, RPE_NC});
    case RPE_RES_BEARING: case RPE_RES_REPROJ: return need_arrays(c, {RPE_XW, RPE_BV});
    case RPE_RES_NORMAL: return need_arrays(c, {RPE_XW, RPE_NW, RPE_NC});
  }
  return fail(RPE_ERR_ARG, "unknown residual kind %d", kind);
}

int check_flags(rpe_context* c, int kind, int flags) {
  const int mod = (kind == RPE_RES_BEARING || kind == RPE_RES_REPROJ) ? RPE_MOD_23 : (kind == RPE_RES_NORMAL ? RPE_MOD_NN : RPE_MOD_33);
  if ((flags & RPE_USE_MASK) && !c->mask[mod]) return fail(RPE_ERR_STATE, "RPE_USE_MASK but no mask for modality %d", mod);
  if ((flags & RPE_USE_WEIGHT) && !c->weight[mod]) return fail(RPE_ERR_STATE, "RPE_USE_WEIGHT but no weight for modality %d", mod);
  return RPE_OK;
}
}  // namespace rpeh
