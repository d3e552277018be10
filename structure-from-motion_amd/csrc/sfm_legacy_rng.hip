// Host side of the drop-in path: the sample-index draws of the reference loops.
//
// The reference draws its minimal samples with np.random.randint(0, N, size=k) from NumPy's global legacy
// generator (src/ransac_algorithm.py:94,116; src/estimate_translation.py:64), and a seeded run must leave that
// generator exactly where the reference leaves it.  NumPy spends 5-25 ns per integer there; the draws of one
// 100 000-iteration translation RANSAC or of one large block of the E/H loop then cost more than scoring them on
// the GPU.  This file restates the two published algorithms involved so that the package can draw the SAME integers
// from the SAME state in place at about 1 ns each:
//   - MT19937 (Matsumoto & Nishimura 1998): state of 624 words + position, regenerated 624 words at a time;
//   - NumPy's legacy bounded integers for ranges below 2^32: a 32-bit draw is masked with the smallest 2^k - 1 that
//     covers N - 1 and redrawn while it exceeds N - 1 (one fresh draw per attempt).
// No device code here; the functions are plain C ABI like the rest of the library.
#include <stdint.h>

#include "../../include/sfm_hip.h"

namespace {
constexpr int kN = 624, kM = 397;
constexpr uint32_t kMatrixA = 0x9908b0dfu, kUpper = 0x80000000u, kLower = 0x7fffffffu;

inline void mt_regenerate(uint32_t* key) {
  int i = 0;
  for (; i < kN - kM; ++i) {
    const uint32_t y = (key[i] & kUpper) | (key[i + 1] & kLower);
    key[i] = key[i + kM] ^ (y >> 1) ^ ((y & 1u) ? kMatrixA : 0u);
  }
  for (; i < kN - 1; ++i) {
    const uint32_t y = (key[i] & kUpper) | (key[i + 1] & kLower);
    key[i] = key[i + (kM - kN)] ^ (y >> 1) ^ ((y & 1u) ? kMatrixA : 0u);
  }
  const uint32_t y = (key[kN - 1] & kUpper) | (key[0] & kLower);
  key[kN - 1] = key[kM - 1] ^ (y >> 1) ^ ((y & 1u) ? kMatrixA : 0u);
}

inline uint32_t temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
}  // namespace

extern "C" int sfm_legacy_randint_fill(uint32_t* key, int32_t* pos, int64_t n, int64_t count, int32_t* out) {
  if (!key || !pos || (!out && count > 0) || n <= 0 || n > 0x7FFFFFFFLL || count < 0 || *pos < 0 || *pos > kN) return SFM_EINVAL;
  const uint32_t rng = (uint32_t)(n - 1);
  if (rng == 0) {  // a range of one value consumes nothing from the stream
    for (int64_t i = 0; i < count; ++i) out[i] = 0;
    return 0;
  }
  uint32_t mask = rng;
  mask |= mask >> 1;
  mask |= mask >> 2;
  mask |= mask >> 4;
  mask |= mask >> 8;
  mask |= mask >> 16;
  // Branch-free compaction: every word is written to the next output slot and the slot only advances when the word is
  // accepted (a rejection is a data-dependent event with probability up to one half: as a branch it would mispredict
  // every few words).  A run of m words yields at most m outputs, so runs are sized by what is still missing.
  int p = *pos;
  int64_t k = 0;
  while (k < count) {
    if (p == kN) {
      mt_regenerate(key);
      p = 0;
    }
    const int64_t left = count - k;
    const int m = (int)((kN - p) < left ? (kN - p) : left);
    for (int j = 0; j < m; ++j) {
      const uint32_t v = temper(key[p + j]) & mask;
      out[k] = (int32_t)v;
      k += (v <= rng) ? 1 : 0;
    }
    p += m;
  }
  *pos = p;
  return 0;
}
