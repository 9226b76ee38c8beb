// Context life cycle of librgbdpose_hip.so (include/rgbd_pose_hip.h Part 2): rpe_create / rpe_destroy, the HBM-resident correspondence
// arrays, masks and weights (rpe_upload / rpe_bind / rpe_download ...), the HIP-event timing of the reduction launches, and the error
// channel (status code + thread-local message).  There is NO CPU fallback: every entry point that computes fails with
// RPE_ERR_NO_DEVICE when no HIP device is usable.
#include "rpe_host.hpp"
using namespace rpeh;
namespace { thread_local std::string g_err; }
namespace rpeh {
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}
}  // namespace rpeh

namespace rpe {
// lets library.cpp (adapter-level pipelines) report through the same rpe_last_error() channel
int set_error(int code, const char* msg) { g_err = msg ? msg : ""; return code; }
}  // namespace rpe
namespace rpeh {
int ensure_mask(rpe_context* c, int mod, bool fill_ones) {
  if (c->mask[mod]) return RPE_OK;
  const size_t need = (size_t)c->n * sizeof(short);
  if (!c->mask_store[mod] || c->mask_cap[mod] < need) {
    if (c->mask_store[mod]) { HIP_TRY(hipFree(c->mask_store[mod])); c->mask_store[mod] = nullptr; c->mask_cap[mod] = 0; }
    HIP_TRY(hipMalloc((void**)&c->mask_store[mod], need ? need : 2));
    c->mask_cap[mod] = need;
  }
  c->mask[mod] = c->mask_store[mod];
  if (fill_ones && c->n)   // adapters start with all-ones masks (e.g. AOPoseAdapter.hpp:103-106): filled on the device, in stream order
    HIP_TRY(hipMemsetD16Async((hipDeviceptr_t)c->mask[mod], (unsigned short)1, (size_t)c->n, c->stream));
  return RPE_OK;
}

// Device -> caller memory.  A D2H copy into pageable memory is staged by the runtime in small pinned chunks (measured ~6 GB/s
// for a 614 KB mask); one DMA into the context's own pinned buffer followed by a host memcpy is about twice as fast.
int copy_to_host(rpe_context* c, void* dst, const void* d_src, size_t bytes) {
  if (bytes == 0) return RPE_OK;
  if (bytes > ((size_t)64 << 20)) {  // very large arrays: not worth pinning that much memory
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RPE_OK;
  }
  if (c->h_stage_cap < bytes) {
    if (c->h_stage) { HIP_TRY(hipHostFree(c->h_stage)); c->h_stage = nullptr; c->h_stage_cap = 0; }
    const size_t cap = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    HIP_TRY(hipHostMalloc(&c->h_stage, cap, hipHostMallocDefault));
    c->h_stage_cap = cap;
  }
  HIP_TRY(hipMemcpyAsync(c->h_stage, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  std::memcpy(dst, c->h_stage, bytes);
  return RPE_OK;
}

int need_arrays(rpe_context* c, std::initializer_list<int> slots) {
  if (!c) return fail(RPE_ERR_ARG, "null context");
  if (c->n <= 0) return fail(RPE_ERR_STATE, "rpe_set_problem was not called (n = %lld)", (long long)c->n);
  static const char* names[] = {"XW (points_g)", "XC (points_c)", "BV (bearingVectors)", "NW (normal_g)", "NC (normal_c)"};
  for (int s : slots) if (!c->arr[s]) return fail(RPE_ERR_STATE, "array %s was never uploaded or bound", names[s]);
  return RPE_OK;
}

// the event pair of the next timed launch (rpe_timing_enable), or nulls
void timing_pair(rpe_context* c, hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = nullptr; *e1 = nullptr;
  if (c->timing && c->ev_used < c->ev0.size() && (c->timing_calls++ % c->timing_stride) == 0) { *e0 = c->ev0[c->ev_used];
      *e1 = c->ev1[c->ev_used]; c->ev_used++; }
}
}  // namespace rpeh

extern "C" {
int rpe_abi_version(void) { return 1; }
const char* rpe_last_error(void) { return g_err.c_str(); }

int rpe_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

int rpe_create(rpe_context** out, int device, void* stream) {
  if (!out) return fail(RPE_ERR_ARG, "null out");
  *out = nullptr;
  const int nd = rpe_device_count();
  if (nd <= 0) return fail(RPE_ERR_NO_DEVICE, "no HIP device is visible; librgbdpose_hip has no CPU fallback");
  if (device < 0 || device >= nd) return fail(RPE_ERR_ARG, "device %d out of range (have %d)", device, nd);
  HIP_TRY(hipSetDevice(device));
  rpe_context* c = new rpe_context();
  c->device = device;
  if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
  else { hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); if (e != hipSuccess) { delete c;
      return fail(RPE_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); } c->own_stream = true; }
  if (const char* mb = getenv("RPE_MAX_BLOCKS")) { int v = atoi(mb); if (v >= 1 && v <= 4096) c->max_blocks = v; }
  if (const char* mb = getenv("RPE_SCORE_BLOCKS")) { int v = atoi(mb); if (v >= 1 && v <= 65535) c->score_blocks = v; }
  if (const char* mb = getenv("RPE_BLOCK")) { int v = atoi(mb); if (v == 256 || v == 512) c->block = v; }
  if (const char* f = getenv("RPE_GUARD_ALWAYS")) c->guard_always = atoi(f) != 0;
  if (const char* f = getenv("RPE_HOST_CPU")) c->host_cpu_request = std::strcmp(f, "auto") == 0 ? -1 : (std::isdigit((unsigned char)f[0]) ? atoi(f) : -2);
  hipError_t e = hipSuccess;
  // scratch of the cross-workgroup stages, whichever layout a launch uses: (4096 + 8 shard) records of kNlLd doubles, or 16-byte
  // granules [workgroup <= 4096][sums <= 44] followed by the autonomous loop's run records [2 parities][<= kAutoMaxRunSums = 1024]
  const size_t partial_doubles = std::max<size_t>((size_t)(4096 + 8) * rpe::kNlLd, (size_t)2 * 4096 * 44 + (size_t)2 * 2 * 1024);
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_partials, partial_doubles * sizeof(double));
  // granule tags start below every sequence value
  if (e == hipSuccess) e = hipMemset(c->d_partials, 0, partial_doubles * sizeof(double));
  // 64 doubles (a record for a collective, the solve probe) + the run records of a sharded step (rpe_dist.hip), zero between steps
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_out, (64 + 2 * rpe::kRunSlots * rpe::kRunLd) * sizeof(double));   // (two sets: chained steps alternate)
  if (e == hipSuccess) e = hipMemset(c->d_out, 0, (64 + 2 * rpe::kRunSlots * rpe::kRunLd) * sizeof(double));
  c->h_big_pairs = 8192 + 64;
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_big, c->h_big_pairs * 16, hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) std::memset(c->h_big, 0, c->h_big_pairs * 16);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_out, 80 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) { std::memset(c->h_out, 0, 80 * sizeof(double)); e = hipMalloc((void**)&c->d_ticket, 9 * 128); }
  if (e == hipSuccess) e = hipMemset(c->d_ticket, 0, 9 * 128);
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_gn_pose, 32 * sizeof(double));   // 12 (+ a second 12 at 16: chained steps alternate)
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_gn_state, sizeof(rpe::GnState));
  if (e == hipSuccess) e = hipMalloc(&c->d_poses, (size_t)rpe::kMaxScoreH * 12 * sizeof(double));
  // staging for pose uploads; also written directly by the hypothesis generator
  if (e == hipSuccess) e = hipHostMalloc(&c->h_poses, (size_t)rpe::kMaxScoreH * 12 * sizeof(double),
      hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_votes, (size_t)rpe::kMaxScoreH * sizeof(int));
  // the scoring kernels accumulate into zeroed counters
  if (e == hipSuccess) e = hipMemset(c->d_votes, 0, (size_t)rpe::kMaxScoreH * sizeof(int));
  // pinned + device-mapped: the vote read-out kernel stores straight into it; the sequence word sits behind the counters
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_votes, ((size_t)rpe::kMaxScoreH + 4) * sizeof(int),
      hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) { std::memset(c->h_votes, 0, ((size_t)rpe::kMaxScoreH + 4) * sizeof(int));
      c->h_flag2 = reinterpret_cast<unsigned long long*>(c->h_votes + rpe::kMaxScoreH); }
  // PROSAC order scratch (rpe_prosac_order): histogram + control words (zero between calls), candidate keys, order + status
  if (e == hipSuccess) e = hipMalloc((void**)&c->ps_hist, (2048 + 8) * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMemset(c->ps_hist, 0, (2048 + 8) * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMalloc((void**)&c->ps_cand, (size_t)rpe::kProsacSortCap * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc((void**)&c->ps_order, ((size_t)rpe::kProsacMaxTopK + 1) * sizeof(int));
  if (e != hipSuccess) { rpe_destroy(c); return fail(RPE_ERR_HIP, "workspace allocation: %s", hipGetErrorString(e)); }
  {  // Resident loops.  The co-residency cap is a property of the device (0: not even one workgroup of the resident kernels per
     // compute unit) and gates both forms; the AUTONOMOUS form (rpe_gn_refine_device, device_resident ICP) needs nothing else.  The
     // HOST-driven form also needs device memory the CPU can store into (large BAR: the control block); RPE_RESIDENT=0 switches that
     // form off and leaves the autonomous one alone (RPE_DEVICE_LOOP_RESIDENT=0 is its switch).
    c->resident_cap = rpe::resident_cap_device();
    c->resident = c->resident_cap >= 1;
    int large_bar = 0;
    const char* env = getenv("RPE_RESIDENT");
    if (c->resident && !(env && env[0] == '0') && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) == hipSuccess
        && large_bar) {
      void* p = nullptr;
      if (hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained) == hipSuccess && hipMemset(p, 0, 4096) == hipSuccess &&
          hipDeviceSynchronize() == hipSuccess) {
        c->ctl = (volatile unsigned long long*)p;
        c->host_resident = true;
      } else { (void)hipGetLastError(); if (p) (void)hipFree(p); }
    } else (void)hipGetLastError();
  }
  {  // first context on this device: load every kernel unit's code object now, not at the first launch out of each
    static std::mutex m;
    static bool loaded[64];
    std::lock_guard<std::mutex> g(m);
    if (device < 64 && !loaded[device]) {
      rpe::preload_normal_eq(); rpe::preload_icp(); rpe::preload_joint(); rpe::preload_score(); rpe::preload_nl();
      rpe::preload_frontend(); rpe::preload_hypotheses(); rpe::preload_prosac();
      loaded[device] = true;
    }
  }
  *out = c;
  return RPE_OK;
}

void rpe_destroy(rpe_context* c) {
  session_end(c);
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (int i = 0; i < RPE_NUM_ARRAYS; i++) if (c->store[i]) (void)hipFree(c->store[i]);
  for (int i = 0; i < 3; i++) { if (c->mask_store[i]) (void)hipFree(c->mask_store[i]);
      if (c->weight_store[i]) (void)hipFree(c->weight_store[i]); }
  if (c->d_partials) (void)hipFree(c->d_partials);
  if (c->d_out) (void)hipFree(c->d_out);
  if (c->d_ticket) (void)hipFree(c->d_ticket);
  if (c->d_gn_pose) (void)hipFree(c->d_gn_pose);
  if (c->d_gn_state) (void)hipFree(c->d_gn_state);
  if (c->h_out) (void)hipHostFree(c->h_out);
  if (c->d_poses) (void)hipFree(c->d_poses);
  if (c->h_poses) (void)hipHostFree(c->h_poses);
  if (c->d_votes) (void)hipFree(c->d_votes);
  if (c->h_votes) (void)hipHostFree(c->h_votes);
  (void)rpe_p2p_destroy(c);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->ctl) (void)hipFree((void*)c->ctl);
  if (c->h_big) (void)hipHostFree(c->h_big);
  if (c->hostex) rpe_host_exchange_close(c->hostex);
  if (c->ps_w) (void)hipFree(c->ps_w);
  if (c->ps_hist) (void)hipFree(c->ps_hist);
  if (c->ps_cand) (void)hipFree(c->ps_cand);
  if (c->ps_order) (void)hipFree(c->ps_order);
  if (c->fe.d_depth) (void)hipFree(c->fe.d_depth);
  for (float* m : c->fe.fmap) if (m) (void)hipFree(m);
  for (float* m : c->fe.mmap) if (m) (void)hipFree(m);
  if (c->fe.d_count) (void)hipFree(c->fe.d_count);
  if (c->comm && rccl().ok) { (void)rccl().CommDestroy(c->comm); c->comm = nullptr; }
  for (hipEvent_t e : c->ev0) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev1) (void)hipEventDestroy(e);
  if (c->ev_stream2) (void)hipEventDestroy(c->ev_stream2);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int rpe_synchronize(rpe_context* c) {
  session_end(c);
  if (!c) return fail(RPE_ERR_ARG, "null context");
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RPE_OK;
}

int rpe_set_problem(rpe_context* c, int64_t n, int dtype) {
  session_end(c);
  if (!c) return fail(RPE_ERR_ARG, "null context");
  if (n < 0 || (dtype != RPE_F32 && dtype != RPE_F64)) return fail(RPE_ERR_ARG, "bad n (%lld) or dtype (%d)", (long long)n, dtype);
  HIP_TRY(hipSetDevice(c->device));
  // a new problem (also one of the same size: new frame) invalidates every array, mask and weight; storage is kept
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int i = 0; i < RPE_NUM_ARRAYS; i++) { c->arr[i] = nullptr; arrays_changed(c, i, false); }
  for (int i = 0; i < 3; i++) { c->mask[i] = nullptr; c->weight[i] = nullptr; }
  c->n = n; c->dtype = dtype;
  return RPE_OK;
}

int rpe_upload(rpe_context* c, int slot, const void* host) {
  session_end(c);
  if (!c || slot < 0 || slot >= RPE_NUM_ARRAYS || !host) return fail(RPE_ERR_ARG, "rpe_upload: bad argument");
  if (c->n <= 0) return fail(RPE_ERR_STATE, "rpe_set_problem first");
  HIP_TRY(hipSetDevice(c->device));
  const size_t bytes = (size_t)c->n * 3 * elem_size(c->dtype);
  if (!c->store[slot] || c->cap[slot] < bytes) {
    if (c->store[slot]) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->store[slot])); c->store[slot] = nullptr;
        c->cap[slot] = 0; }
    HIP_TRY(hipMalloc(&c->store[slot], bytes));
    c->cap[slot] = bytes;
  }
  c->arr[slot] = c->store[slot];
  arrays_changed(c, slot, false);
  HIP_TRY(hipMemcpyAsync(c->arr[slot], host, bytes, hipMemcpyHostToDevice, c->stream));
  return RPE_OK;
}

int rpe_download(rpe_context* c, int slot, void* host) {
  session_end(c);
  if (!c || slot < 0 || slot >= RPE_NUM_ARRAYS || !host) return fail(RPE_ERR_ARG, "rpe_download: bad argument");
  if (!c->arr[slot]) return fail(RPE_ERR_STATE, "array slot %d was never uploaded, bound or produced", slot);
  HIP_TRY(hipSetDevice(c->device));
  return copy_to_host(c, host, c->arr[slot], (size_t)c->n * 3 * elem_size(c->dtype));
}

int rpe_bind(rpe_context* c, int slot, const void* device_ptr) {
  session_end(c);
  if (!c || slot < 0 || slot >= RPE_NUM_ARRAYS) return fail(RPE_ERR_ARG, "rpe_bind: bad argument");
  if (device_ptr && ((uintptr_t)device_ptr & 15u)) return fail(RPE_ERR_ALIGN, "device pointer %p is not 16-byte aligned", device_ptr);
  c->arr[slot] = const_cast<void*>(device_ptr);  // not owned; the context's own storage for this slot stays allocated but idle
  arrays_changed(c, slot, true);
  return RPE_OK;
}

int rpe_upload_mask(rpe_context* c, int mod, const short* host_mask) {
  session_end(c);
  if (!c || mod < 0 || mod > 2) return fail(RPE_ERR_ARG, "rpe_upload_mask: bad argument");
  HIP_TRY(hipSetDevice(c->device));
  if (!host_mask) { c->mask[mod] = nullptr; return RPE_OK; }
  int rc = ensure_mask(c, mod, false);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->mask[mod], host_mask, (size_t)c->n * sizeof(short), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RPE_OK;
}

int rpe_upload_weight(rpe_context* c, int mod, const void* host_weight) {
  session_end(c);
  if (!c || mod < 0 || mod > 2) return fail(RPE_ERR_ARG, "rpe_upload_weight: bad argument");
  HIP_TRY(hipSetDevice(c->device));
  if (!host_weight) { c->weight[mod] = nullptr; return RPE_OK; }
  const size_t need = (size_t)c->n * elem_size(c->dtype);
  if (!c->weight_store[mod] || c->weight_cap[mod] < need) {
    if (c->weight_store[mod]) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->weight_store[mod]));
        c->weight_store[mod] = nullptr; }
    HIP_TRY(hipMalloc(&c->weight_store[mod], need ? need : 8));
    c->weight_cap[mod] = need;
  }
  c->weight[mod] = c->weight_store[mod];
  HIP_TRY(hipMemcpyAsync(c->weight[mod], host_weight, need, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RPE_OK;
}

int rpe_download_mask(rpe_context* c, int mod, short* host_mask) {
  session_end(c);
  if (!c || mod < 0 || mod > 2 || !host_mask) return fail(RPE_ERR_ARG, "rpe_download_mask: bad argument");
  if (!c->mask[mod]) return fail(RPE_ERR_STATE, "no mask for modality %d", mod);
  HIP_TRY(hipSetDevice(c->device));
  return copy_to_host(c, host_mask, c->mask[mod], (size_t)c->n * sizeof(short));
}
// the event pair of the next timed launch (rpe_timing_enable), or nulls
int rpe_timing_enable(rpe_context* c, int max_records, int stride) {
  if (!c || max_records < 0 || stride < 1) return fail(RPE_ERR_ARG, "rpe_timing_enable: bad argument");
  c->timing_stride = stride; c->timing_calls = 0;
  HIP_TRY(hipSetDevice(c->device));
  while ((int)c->ev0.size() < max_records) {
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    c->ev0.push_back(a); c->ev1.push_back(b);
  }
  c->ev_used = 0;
  c->timing = max_records > 0;
  return RPE_OK;
}

int rpe_timing_collect(rpe_context* c, int* count, double* total_ms, double* min_ms) {
  session_end(c);   // (the synchronise below would otherwise sit behind an open session's grid for its whole bounded wait)
  if (!c) return fail(RPE_ERR_ARG, "null context");
  HIP_TRY(hipStreamSynchronize(c->stream));
  double tot = 0, mn = 1e30;
  for (size_t i = 0; i < c->ev_used; i++) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]));
    tot += ms; if (ms < mn) mn = ms;
  }
  if (count) *count = (int)c->ev_used;
  if (total_ms) *total_ms = tot;
  if (min_ms) *min_ms = c->ev_used ? mn : 0.0;
  c->ev_used = 0;
  return RPE_OK;
}

int rpe_timing_calibrate(rpe_context* c, int pairs, double* avg_ms, double* min_ms) {
  session_end(c);
  if (!c || pairs < 1 || pairs > 4096) return fail(RPE_ERR_ARG, "rpe_timing_calibrate: bad argument");
  HIP_TRY(hipSetDevice(c->device));
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a));
  HIP_TRY(hipEventCreate(&b));
  double tot = 0, mn = 1e30;
  for (int i = 0; i < pairs; i++) {  // one pair at a time, stream idle in between: the way the timed launches see their pair
    HIP_TRY(hipEventRecord(a, c->stream));
    HIP_TRY(hipEventRecord(b, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    tot += ms; if (ms < mn) mn = ms;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  if (avg_ms) *avg_ms = tot / pairs;
  if (min_ms) *min_ms = mn;
  return RPE_OK;
}
}  // extern "C"
