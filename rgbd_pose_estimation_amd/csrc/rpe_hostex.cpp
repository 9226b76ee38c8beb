// Host-side all-reduce between the rank processes of ONE node (include/rgbd_pose_hip.h, rpe_host_exchange_* / rpe_hostex_*).
//
// Why it exists: on one GPU the cross-workgroup sums of a reduction already end on the HOST (collecting workgroups send a few run
// records as tagged pairs; the thread that owns the 6x6 solve adds them, rpe_receive.hip wait_collect).  With the correspondences sharded
// over the GPUs of a node every rank's host thread therefore holds its shard's 32-double record a few microseconds after its kernel's
// last workgroup -- and the rank processes share the node's memory.  Exchanging 256 bytes between host threads through a POSIX
// shared-memory segment costs a cache-line transfer per peer (well under a microsecond), where a collective on the GPUs costs two more
// kernel launches and ~20 us of small-message latency per Gauss-Newton iteration.  The RCCL path (rpe_comm_init) stays; bench.py
// times both.
//
// Protocol: one slot per (parity, rank) and message class; a rank writes its payload, then its step number with release order; every
// rank reads all slots of the step's parity in RANK ORDER once their step numbers match (acquire) and adds them in that order, so all
// ranks compute bitwise the same sums.  Parity alternates per step: nobody can be two steps ahead of a rank that is still reading
// (completing step s + 1 needs that rank's step-(s + 1) slot, which it writes after it has finished reading step s).  Every wait is
// bounded (timeout_s): a missing peer fails the call instead of hanging it.
#include "../../include/rgbd_pose_hip.h"
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <vector>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace rpe { int set_error(int code, const char* msg); }

namespace {

constexpr int kMaxWorld = 8;
constexpr int kMaxF64 = 64;
constexpr int kMaxI32 = 8192;
constexpr unsigned long long kMagic = 0x7270655f68783031ull;   // "rpe_hx01"

struct alignas(64) RecSlot { double v[kMaxF64]; unsigned long long step; int n; char pad[64 - 12]; };
struct alignas(64) VoteSlot { int v[kMaxI32]; unsigned long long step; int n; char pad[64 - 12]; };
struct alignas(64) Header { unsigned long long magic; int world; char pad0[52]; char busid[kMaxWorld][64]; };
struct Segment { Header h; RecSlot rec[2][kMaxWorld]; VoteSlot votes[2][kMaxWorld]; };

inline void cpu_relax() {
#if defined(__x86_64__)
  _mm_pause();
#endif
}
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int err(int code, const std::string& msg) { return rpe::set_error(code, msg.c_str()); }

}  // namespace

struct rpe_host_exchange {
  Segment* seg = nullptr;
  std::string name;
  int world = 1, rank = 0;
  double timeout_s = 10.0;
  unsigned long long rec_step = 0, vote_step = 0;
  bool owner = false, unlinked = false;
  bool broken = false;   // a peer missed a step: this rank's step counters have moved on without it, the handle cannot be used again
};

extern "C" {

int rpe_host_exchange_open(const char* name, int world, int rank, int create, double timeout_s, rpe_host_exchange** out) {
  if (!name || name[0] != '/' || !out || world < 1 || world > kMaxWorld || rank < 0 || rank >= world)
    return err(RPE_ERR_ARG, "rpe_host_exchange_open: bad argument (name must start with '/', 1 <= world <= 8)");
  if (!(timeout_s > 0)) timeout_s = 10.0;
  int fd = -1;
  if (create) {
    (void)shm_unlink(name);   // a stale segment of the same name (a crashed run)
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return err(RPE_ERR_STATE, std::string("shm_open(create ") + name + "): " + std::strerror(errno));
    if (ftruncate(fd, (off_t)sizeof(Segment)) != 0) { const int e = errno; close(fd); (void)shm_unlink(name);
        return err(RPE_ERR_STATE, std::string("ftruncate: ") + std::strerror(e)); }
  } else {
    const double t0 = now_s();
    for (;;) {   // the creating rank may be a moment behind
      fd = shm_open(name, O_RDWR, 0600);
      if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(Segment)) break;
        close(fd); fd = -1;
      }
      if (now_s() - t0 > timeout_s) return err(RPE_ERR_STATE, std::string("host exchange segment ") + name + " did not appear");
      usleep(200);
    }
  }
  void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  const int e = errno;
  close(fd);
  if (p == MAP_FAILED) { if (create) (void)shm_unlink(name); return err(RPE_ERR_STATE,
      std::string("mmap of the host exchange segment: ") + std::strerror(e)); }
  rpe_host_exchange* h = new rpe_host_exchange;
  h->seg = static_cast<Segment*>(p); h->name = name; h->world = world; h->rank = rank; h->timeout_s = timeout_s; h->owner = create != 0;
  if (create) {   // a fresh segment is zero-filled: step numbers start below every step
    h->seg->h.world = world;
    __atomic_store_n(&h->seg->h.magic, kMagic, __ATOMIC_RELEASE);
  } else {
    const double t0 = now_s();
    while (__atomic_load_n(&h->seg->h.magic, __ATOMIC_ACQUIRE) != kMagic) {
      if (now_s() - t0 > timeout_s) { munmap(p, sizeof(Segment)); delete h;
          return err(RPE_ERR_STATE, "host exchange segment was never initialised by its creator"); }
      cpu_relax();
    }
    if (h->seg->h.world != world) { munmap(p, sizeof(Segment)); delete h;
        return err(RPE_ERR_ARG, "host exchange segment belongs to a different world size"); }
  }
  *out = h;
  return RPE_OK;
}

void rpe_host_exchange_close(rpe_host_exchange* h) {
  if (!h) return;
  if (h->owner && !h->unlinked) (void)shm_unlink(h->name.c_str());
  if (h->seg) munmap(h->seg, sizeof(Segment));
  delete h;
}

// remove the NAME (the memory stays until the last rank unmaps it): call once every rank has opened the segment, e.g. after the first
// exchange
int rpe_host_exchange_unlink(rpe_host_exchange* h) {
  if (!h) return err(RPE_ERR_ARG, "null exchange");
  if (!h->unlinked) { (void)shm_unlink(h->name.c_str()); h->unlinked = true; }
  return RPE_OK;
}

int rpe_host_exchange_set_label(rpe_host_exchange* h, const char* label) {
  if (!h || !label) return err(RPE_ERR_ARG, "rpe_host_exchange_set_label: bad argument");
  std::strncpy(h->seg->h.busid[h->rank], label, 63);
  h->seg->h.busid[h->rank][63] = 0;
  return RPE_OK;
}
// 1 if two ranks carry the same non-empty label (labels are complete after the first exchange that follows every rank's set_label)
int rpe_host_exchange_labels_collide(rpe_host_exchange* h) {
  if (!h) return 0;
  for (int a = 0; a < h->world; a++)
    for (int b = a + 1; b < h->world; b++)
      if (h->seg->h.busid[a][0] && std::strncmp(h->seg->h.busid[a], h->seg->h.busid[b], 64) == 0) return 1;
  return 0;
}

int rpe_host_exchange_allreduce_f64(rpe_host_exchange* h, double* v, int n) {
  if (!h || !v || n < 1 || n > kMaxF64) return err(RPE_ERR_ARG, "rpe_host_exchange_allreduce_f64: bad argument (1 <= n <= 64)");
  if (h->broken) return err(RPE_ERR_STATE,
      "host exchange: an earlier step timed out; close this exchange and open a new one on every rank");
  const unsigned long long step = ++h->rec_step;
  RecSlot* row = h->seg->rec[step & 1];
  RecSlot& mine = row[h->rank];
  std::memcpy(mine.v, v, (size_t)n * sizeof(double));
  mine.n = n;
  __atomic_store_n(&mine.step, step, __ATOMIC_RELEASE);
  double sum[kMaxF64];
  for (int i = 0; i < n; i++) sum[i] = 0.0;
  double t0 = 0;
  for (int r = 0; r < h->world; r++) {   // rank order: every rank forms bitwise the same sums
    RecSlot& s = row[r];
    for (unsigned long long spins = 1; __atomic_load_n(&s.step, __ATOMIC_ACQUIRE) != step; spins++) {
      cpu_relax();
      if ((spins & 0xFFF) == 0) {
        if (t0 == 0) t0 = now_s();
        else if (now_s() - t0 > h->timeout_s) {
          char msg[160];
          std::snprintf(msg, sizeof msg, "host exchange: rank %d did not deliver its record of step %llu within %.1f s", r, step,
              h->timeout_s);
          h->broken = true;   // the caller's v is untouched (the sums live in a local buffer until every rank has delivered)
          return err(RPE_ERR_STATE, msg);
        }
      }
    }
    if (s.n != n) { h->broken = true; return err(RPE_ERR_STATE, "host exchange: ranks disagree on the record length of a step"); }
    for (int i = 0; i < n; i++) sum[i] += s.v[i];
  }
  std::memcpy(v, sum, (size_t)n * sizeof(double));
  return RPE_OK;
}

int rpe_host_exchange_allreduce_i32(rpe_host_exchange* h, int* v, int n) {
  if (!h || !v || n < 1 || n > kMaxI32) return err(RPE_ERR_ARG, "rpe_host_exchange_allreduce_i32: bad argument (1 <= n <= 8192)");
  if (h->broken) return err(RPE_ERR_STATE,
      "host exchange: an earlier step timed out; close this exchange and open a new one on every rank");
  const unsigned long long step = ++h->vote_step;
  VoteSlot* row = h->seg->votes[step & 1];
  VoteSlot& mine = row[h->rank];
  std::memcpy(mine.v, v, (size_t)n * sizeof(int));
  mine.n = n;
  __atomic_store_n(&mine.step, step, __ATOMIC_RELEASE);
  std::vector<int> sum((size_t)n, 0);   // the caller's v changes only once every rank has delivered
  double t0 = 0;
  for (int r = 0; r < h->world; r++) {
    VoteSlot& s = row[r];
    for (unsigned long long spins = 1; __atomic_load_n(&s.step, __ATOMIC_ACQUIRE) != step; spins++) {
      cpu_relax();
      if ((spins & 0xFFF) == 0) {
        if (t0 == 0) t0 = now_s();
        else if (now_s() - t0 > h->timeout_s) {
          char msg[160];
          std::snprintf(msg, sizeof msg, "host exchange: rank %d did not deliver its counters of step %llu within %.1f s", r, step,
              h->timeout_s);
          h->broken = true;
          return err(RPE_ERR_STATE, msg);
        }
      }
    }
    if (s.n != n) { h->broken = true; return err(RPE_ERR_STATE, "host exchange: ranks disagree on the counter count of a step"); }
    for (int i = 0; i < n; i++) sum[(size_t)i] += s.v[i];
  }
  std::memcpy(v, sum.data(), (size_t)n * sizeof(int));
  return RPE_OK;
}

}  // extern "C"
