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
#include <numeric>
#include <vector>

namespace rpe {
class Rand31 {
 public:
  explicit Rand31(uint64_t seed = 1, uint64_t stream = 54) { reseed(seed, stream); }
  void reseed(uint64_t seed, uint64_t stream = 54) {
    _state = 0; _inc = (stream << 1) | 1u;
    step(); _state += seed; step();
  }
  int operator()() { return (int)(step() >> 1); }  // uniform in [0, 2^31)
 private:
  uint32_t step() {
    const uint64_t old = _state;
    _state = old * 6364136223846793005ULL + _inc;
    const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
  }
  uint64_t _state, _inc;
};
inline Rand31& global_rng() { static Rand31 g(1); return g; }
inline void seed(uint64_t s) { global_rng().reseed(s); }
}  // namespace rpe

// indices that sort v in DESCENDING order (reference :107-118)
template <typename T>
std::vector<int> sortIndexes(const std::vector<T>& v) {
  std::vector<int> order(v.size());
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&v](int a, int b) { return v[a] > v[b]; });
  return order;
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
  void set(int p, T v) { for (size_t i = 0; i < _pos.size(); i++) if (_pos[i] == p) { _val[i] = v; return; } _pos.push_back(p); _val.push_back(v); }
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
