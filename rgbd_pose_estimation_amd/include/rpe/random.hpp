// rpe/random.hpp -- the explicit, portable random stream that stands in for libc rand() (pose/Utility.hpp draws from it).
// rpe::Rand31 is PCG32 (XSH-RR 64/32, pcg-random.org) shifted right once to rand()'s 31-bit range.  rpe::global_rng() plays the
// role of the process-global rand() state of the reference (/root/reference/pose/Utility.hpp:148,212,229): it is what the drop-in
// free functions draw from when the caller passes no stream of its own, and like rand() it is NOT safe to share between threads.
// Callers that run solvers concurrently hand every run its own stream (rpe::RunOptions::rng, rpe/device.hpp); the C ABI
// (rpe_run, rpe_host_hypotheses, ao_ransac) always does.
#pragma once
#include <cstdint>

namespace rpe {
class Rand31 {
 public:
  explicit Rand31(uint64_t seed = 1, uint64_t stream = 54) { reseed(seed, stream); }
  void reseed(uint64_t seed, uint64_t stream = 54) {
    _state = 0; _inc = (stream << 1) | 1u;
    step(); _state += seed; step();
  }
  int operator()() { return (int)(step() >> 1); }  // uniform in [0, 2^31)
  // for samplers that run on the device from this stream's current position (one draw = one LCG step)
  uint64_t state() const { return _state; }
  uint64_t inc() const { return _inc; }
  void advance(uint64_t draws) {   // skip `draws` draws in O(log draws)
    uint64_t cur_mult = 6364136223846793005ULL, cur_plus = _inc, acc_mult = 1, acc_plus = 0;
    while (draws > 0) {
      if (draws & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
      cur_plus = (cur_mult + 1) * cur_plus;
      cur_mult *= cur_mult;
      draws >>= 1;
    }
    _state = acc_mult * _state + acc_plus;
  }
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
