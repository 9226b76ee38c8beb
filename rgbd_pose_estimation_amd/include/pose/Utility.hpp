// pose/Utility.hpp -- drop-in for the sampling utilities of /root/reference/pose/Utility.hpp:107-250
// (sortIndexes, RandomElements, ProsacSampler).  The unused helpers of that file (getNeighbourIdxCylinder,
// matNormL1, stream operators) are out of scope.
//
// The reference draws indices with libc rand(), never seeded (:148,212,229).  Here the stream is explicit and
// portable -- rpe::Rand31, PCG32 (XSH-RR 64/32, pcg-random.org) shifted right once to rand()'s 31-bit range -- so
// that runs are reproducible and the sampled index sequences can be compared bit for bit with the CPU oracle.
// rpe::global_rng() plays the role of the process-global rand() state; rpe::seed(s) replaces srand(s).
#ifndef RPE_UTILITY_HEADER
#define RPE_UTILITY_HEADER

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <utility>
#include <vector>
#include "../rpe/random.hpp"


// indices that sort v in DESCENDING order (reference :107-118).  The reference's comparator leaves the order of equal
// weights to std::sort; here ties go to the lower index, which makes the order unique -- so the first top_k entries can be
// produced alone (std::partial_sort, O(N log top_k)) and are exactly the prefix of the full order.  PROSAC reads position
// j of this order only for j <= m + (number of draws so far) (ProsacSampler below: n grows by at most one per draw), so a
// run bounded by Iter iterations needs a prefix of about Iter entries, not all N: 20 ms -> 0.4 ms at N = 307200.
namespace rpe {
// (weight, index) as one integer whose ASCENDING order is "weight descending, then index ascending"; float only
inline uint64_t prosac_key(float w, int index) {
  w += 0.0f;  // -0 -> +0, so that equal weights compare equal as integers too
  uint32_t u;
  std::memcpy(&u, &w, 4);
  u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;   // order-preserving map of IEEE floats to unsigned
  return ((uint64_t)(~u) << 32) | (uint32_t)index;
}
template <class T> inline void sort_candidates(std::vector<std::pair<T, int> >& cand, int top_k, std::vector<int>& order) {
  std::sort(cand.begin(), cand.end(), [](const std::pair<T, int>& a, const std::pair<T,
      int>& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
  order.resize((size_t)top_k);
  for (int i = 0; i < top_k; i++) order[i] = cand[i].second;
}
inline void sort_candidates(std::vector<std::pair<float, int> >& cand, int top_k, std::vector<int>& order) {
  std::vector<uint64_t> keys(cand.size());
  for (size_t i = 0; i < cand.size(); i++) keys[i] = prosac_key(cand[i].first, cand[i].second);
  // plain integer compares; select the prefix first (linear), then order only the prefix
  if ((size_t)top_k < keys.size()) { std::nth_element(keys.begin(), keys.begin() + top_k, keys.end()); keys.resize((size_t)top_k); }
  std::sort(keys.begin(), keys.end());
  order.resize((size_t)top_k);
  for (int i = 0; i < top_k; i++) order[i] = (int)(uint32_t)keys[i];
}
}  // namespace rpe

template <typename T>
std::vector<int> sortIndexes(const std::vector<T>& v, int top_k = -1) {
  const size_t N = v.size();
  auto before = [&v](int a, int b) { return v[a] > v[b] || (v[a] == v[b] && a < b); };
  std::vector<int> order;
  if (top_k >= 0 && (size_t)top_k < N && N >= 65536 && (size_t)top_k * 16 < N) {
    // a short prefix of a long array: a strided subsample gives a cut value that keeps about 2 top_k candidates; every element
    // at or above the cut is collected in one sequential scan.  If at least top_k were found, the true prefix is among them
    // (anything below the cut ranks after all of them), so sorting the candidates alone is exact; otherwise fall through.
    const size_t S = 1024, stride = N / S;
    std::vector<T> sample(S);
    for (size_t i = 0; i < S; i++) sample[i] = v[i * stride];
    std::sort(sample.begin(), sample.end(), [](T a, T b) { return a > b; });
    const size_t rank = std::min(S - 1, (size_t)(2.0 * top_k * S / N) + 12);
    const T cut = sample[rank];
    std::vector<std::pair<T, int> > cand;
    cand.reserve((size_t)top_k * 3);
    const T* p = v.data();
    for (size_t i = 0; i < N; i++) if (p[i] >= cut) cand.emplace_back(p[i], (int)i);
    if (cand.size() >= (size_t)top_k) { rpe::sort_candidates(cand, top_k, order); return order; }
  }
  order.resize(N);
  std::iota(order.begin(), order.end(), 0);
  if (top_k < 0 || (size_t)top_k >= N) { std::sort(order.begin(), order.end(), before); return order; }
  std::partial_sort(order.begin(), order.begin() + top_k, order.end(), before);
  order.resize((size_t)top_k);
  return order;
}
// positions -> correspondence indices through a (possibly partial) order; a position beyond the prefix extends it to the full order
template <typename T>
void mapSortedIdx(const std::vector<T>& weights, std::vector<int>& order, std::vector<int>& select_) {
  for (size_t i = 0; i < select_.size(); ++i) {
    const int j = select_[i];
    if (j >= (int)order.size() && order.size() < weights.size()) order = sortIndexes<T>(weights);
    if (j < (int)order.size()) select_[i] = order[j];
  }
}

// m distinct indices out of n by a partial Fisher-Yates pass from the top (reference :124-156): position j swaps with
// rand() % (j + 1), j = n-1 ... n-m, on an index table that is RE-INITIALISED on every draw.  The reference rewrites
// all n entries per draw (:141-143) -- O(N) host work per RANSAC iteration, as much as the vote loop this backend moved to
// the GPU.  Because the table always starts as the identity, only the <= 2m entries a draw touches can differ from it:
// they are kept in a tiny side list, every other position p still holds p.  Same index stream, O(m^2) per draw.
template <class T>
class RandomElements {
 public:
  explicit RandomElements(int n) : _n(n) {}
  void run(int m, std::vector<T>* p_v_idx_) { run(m, p_v_idx_, rpe::global_rng()); }
  void run(int m, std::vector<T>* out, rpe::Rand31& rnd) {
    out->clear();
    _pos.clear(); _val.clear();
    for (int top = _n - 1; top > _n - m - 1; top--) {
      const int pick = rnd() % (top + 1);
      const T vp = get(pick), vt = get(top);
      set(pick, vt);
      set(top, vp);
      out->push_back(vp);
    }
  }
 private:
  T get(int p) const { for (size_t i = 0; i < _pos.size(); i++) if (_pos[i] == p) return _val[i]; return T(p); }
  void set(int p, T v) { for (size_t i = 0; i < _pos.size(); i++) if (_pos[i] == p) { _val[i] = v; return; } _pos.push_back(p);
      _val.push_back(v); }
  std::vector<int> _pos;
  std::vector<T> _val;
  int _n;
};

// PROSAC progressive sampler (Chum & Matas 2005) exactly as the reference drives it (:161-250): T_N = 20000, the
// growth function is re-evaluated from t = 1 on every call, and the "n-th point" of the else-branch is index n
// (not n-1) -- kept, but clamped to N-1 where the reference would read out of bounds (n == N).
template <class T>
class ProsacSampler {
 public:
  ProsacSampler(const int min_num_samples, const int num_datapoints) : _N(num_datapoints), _T_N(20000), _t(1), _m(min_num_samples) {}
  void setSampleNumber(int k) { _t = k; }
  bool sample(std::vector<int>* subset_indices) { return sample(subset_indices, rpe::global_rng()); }
  bool sample(std::vector<int>* subset, rpe::Rand31& rnd) {
    T t_n = (T)_T_N;
    int n = _m;
    for (int i = 0; i < _m; i++) t_n *= static_cast<T>(n - i) / (_N - i);
    T t_n_prime = 1.0;
    for (int t = 1; t <= _t; t++) {
      if (t > t_n_prime && n < _N) {
        const T next = (t_n * (n + 1.0)) / (n + 1.0 - _m);
        t_n_prime += std::ceil(next - t_n);
        t_n = next;
        n++;
      }
    }
    subset->clear();
    const bool from_top_n = t_n_prime < _t;
    const int draws = from_top_n ? _m : _m - 1, range = from_top_n ? n : n - 1;
    for (int i = 0; i < draws; i++) {
      int r;
      do { r = rnd() % range; } while (std::find(subset->begin(), subset->end(), r) != subset->end());
      subset->push_back(r);
    }
    if (!from_top_n) subset->push_back(n < _N ? n : _N - 1);
    _t++;
    return true;
  }
 private:
  int _N, _T_N, _t, _m;
};

#endif
