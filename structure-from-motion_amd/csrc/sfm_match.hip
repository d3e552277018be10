// Brute-force 2-nearest-neighbour descriptor matching with the ratio test (gfx950 / MI355X only).
//
// SIFT descriptors as OpenCV emits them are 128 integer-valued components in [0, 255], so the
// squared L2 distance between two descriptors is an integer below 2^24: it is computed exactly
// with int8 matrix cores (v_mfma_i32_32x32x32_i8) as |a|^2 + |b|^2 - 2 a.b on the descriptors
// shifted by -128 (the shift cancels in a - b), and the float32 distance OpenCV reports is the
// correctly rounded square root of that integer.  Nothing here depends on a summation order.
//
// Layout: a wave owns 32 query descriptors (A fragments live in registers for the whole kernel)
// and walks the train descriptors in tiles of 32 (B fragments from LDS, where the four waves of a
// workgroup share each staged tile).
// In the 32x32 accumulator layout a lane holds one train column and 16 query rows, so the running
// best / second-best per (lane, row slot) needs no cross-lane traffic inside the loop; the 32
// column classes are merged once at the end with wave shuffles.  Both MFMA operands are loaded
// with the same byte -> k assignment, so the result does not depend on the instruction's k map.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sfm_hip.h"

namespace {

constexpr int kDim = 128;          // descriptor length
constexpr int kBlock = 256;        // 4 waves, 32 queries each
constexpr int kInvalidNorm = 1 << 29;
constexpr int kFloor = -(1 << 30);

using v4i = __attribute__((ext_vector_type(4))) int;
using v16i = __attribute__((ext_vector_type(16))) int;

inline void clear_stale_error() { (void)hipGetLastError(); }
inline int check_launch() { return (int)hipGetLastError(); }

// float32 / uint8 descriptors [m][128] -> int8 (value - 128) + squared norm of the shifted vector.
// `bad` is set when a component is not an integer in [0, 255].
template <typename TSrc>
__global__ __launch_bounds__(kBlock) void descriptors_to_i8_kernel(const TSrc* __restrict__ src, long long m, int8_t* __restrict__ dst,
                                                                   int32_t* __restrict__ norm, int32_t* __restrict__ bad) {
  const long long d = (long long)blockIdx.x * (kBlock / 32) + (threadIdx.x >> 5);
  const int part = threadIdx.x & 31;  // 4 components per lane
  int s = 0;
  unsigned packed = 0;
  bool invalid = false;
  if (d < m) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = (float)src[d * kDim + 4 * part + j];
      const int iv = (int)v;
      if (!(v >= 0.0f && v <= 255.0f) || (float)iv != v) invalid = true;
      const int sh = (iv & 255) - 128;
      s += sh * sh;
      packed |= (unsigned)(sh & 255) << (8 * j);
    }
    reinterpret_cast<unsigned*>(dst)[d * (kDim / 4) + part] = packed;
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
  if (d < m && part == 0) norm[d] = s;
  if (invalid) atomicOr(bad, 1);
}

__device__ __forceinline__ int med3i(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

// partial[split][query] = (e1, train index of e1, e2, 0) with e = 2 a.b - |b|^2 (larger = closer);
// squared distance = |a|^2 - e.
__global__ __launch_bounds__(kBlock) void knn2_kernel(const int8_t* __restrict__ q8, int nq, const int8_t* __restrict__ t8,
                                                      const int32_t* __restrict__ tnorm, int nt, int tiles_per_split,
                                                      int4* __restrict__ partial) {
  // Train tiles (32 descriptors x 128 B each) are shared by the four waves of the workgroup through LDS, kGroup tiles
  // per barrier (the kernel would otherwise be bound by L2 bandwidth: every wave streams the whole train set); rows
  // are padded to 144 B against bank conflicts.
  constexpr int kGroup = 4;
  __shared__ v4i tile[2][kGroup][32 * 9];
  __shared__ int tile_norm[2][kGroup][32];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int qbase = (blockIdx.x * (kBlock / 64) + wave) * 32;
  const bool active = qbase < nq;  // wave-uniform; inactive waves still stage tiles and meet the barriers
  const int ntiles = (nt + 31) / 32;
  const int tile_begin = blockIdx.y * tiles_per_split;
  const int tile_end = min(ntiles, tile_begin + tiles_per_split);

  v4i a[4];
  {
    const int row = min(qbase + r, nq - 1);
    const v4i* p = reinterpret_cast<const v4i*>(q8 + (long long)row * kDim + 64 * h);
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = p[s];
  }
  int b1[16], b2[16], t1[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { b1[i] = kFloor; b2[i] = kFloor; t1[i] = 0; }

  // staging: thread t fetches the 16 bytes `part` of train column sc of each tile of the group
  const int sc = threadIdx.x >> 3, part = threadIdx.x & 7;
  v4i stage[kGroup];
  int stage_norm[kGroup];
  auto fetch = [&](int tt0) {
#pragma unroll
    for (int g = 0; g < kGroup; ++g) {
      const int col = (tt0 + g) * 32 + sc;
      const int colc = min(col, nt - 1);  // tiles past the end repeat the last descriptor with an invalid norm
      stage[g] = *reinterpret_cast<const v4i*>(t8 + (long long)colc * kDim + 16 * part);
      stage_norm[g] = col < nt ? tnorm[colc] : kInvalidNorm;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int g = 0; g < kGroup; ++g) {
      tile[buf][g][sc * 9 + part] = stage[g];
      if (part == 0) tile_norm[buf][g][sc] = stage_norm[g];
    }
  };
  if (tile_begin < tile_end) {
    fetch(tile_begin);
    commit(0);
  }
  __syncthreads();
  int buf = 0;
  for (int tt0 = tile_begin; tt0 < tile_end; tt0 += kGroup) {
    const bool more = tt0 + kGroup < tile_end;
    if (more) fetch(tt0 + kGroup);  // in flight while this group is scored
    if (active) {
#pragma unroll
      for (int g = 0; g < kGroup; ++g) {
        const int tt = tt0 + g;
        if (tt < tile_end) {  // uniform
          v4i b[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) b[s] = tile[buf][g][r * 9 + 4 * h + s];
          const int ntn = -tile_norm[buf][g][r];
          v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[s], acc, 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int e = (acc[i] << 1) + ntn;
            t1[i] = e > b1[i] ? tt : t1[i];
            b2[i] = med3i(b1[i], b2[i], e);
            b1[i] = max(b1[i], e);
          }
        }
      }
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (!active) return;

  // merge the 32 column classes of each half-wave: key = (e + 2^30, ~train index), the largest key is the
  // closest descriptor with the lowest index on ties; only the distance of the runner-up is kept.
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    unsigned long long k1 = ((unsigned long long)(unsigned)(b1[i] - kFloor) << 32) | (0xFFFFFFFFu - (unsigned)(t1[i] * 32 + r));
    unsigned long long k2 = (unsigned long long)(unsigned)(b2[i] - kFloor) << 32;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const unsigned long long y1 = __shfl_xor(k1, o, 64);
      const unsigned long long y2 = __shfl_xor(k2, o, 64);
      const unsigned long long lo = k1 < y1 ? k1 : y1;
      const unsigned long long hi2 = k2 > y2 ? k2 : y2;
      k1 = k1 > y1 ? k1 : y1;
      k2 = lo > hi2 ? lo : hi2;
    }
    const int q = qbase + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (r == 0 && q < nq) {
      const int e1 = (int)(unsigned)(k1 >> 32) + kFloor;
      const int e2 = (int)(unsigned)(k2 >> 32) + kFloor;
      const int id = (int)(0xFFFFFFFFu - (unsigned)(k1 & 0xFFFFFFFFu));
      partial[(long long)blockIdx.y * nq + q] = make_int4(e1, id, e2, 0);
    }
  }
}

// Merges the splits, turns e into distances and applies the ratio test in double, as the
// reference's `m.distance < ratio * n.distance` does on the two float32 distances.
__global__ __launch_bounds__(kBlock) void knn2_finalize_kernel(const int4* __restrict__ partial, int splits, int nq,
                                                               const int32_t* __restrict__ qnorm, double ratio,
                                                               int32_t* __restrict__ train_idx, float* __restrict__ dist1,
                                                               float* __restrict__ dist2, uint8_t* __restrict__ good,
                                                               int32_t* __restrict__ good_count) {
  const int q = blockIdx.x * kBlock + threadIdx.x;
  bool ok = false;
  if (q < nq) {
    int e1 = kFloor, e2 = kFloor, id = 0;
    for (int s = 0; s < splits; ++s) {
      const int4 p = partial[(long long)s * nq + q];
      if (p.x > e1 || (p.x == e1 && p.y < id)) {
        e2 = max(e2, e1);
        e1 = p.x;
        id = p.y;
      } else {
        e2 = max(e2, p.x);
      }
      e2 = max(e2, p.z);
    }
    const int qn = qnorm[q];
    const float d1 = sqrtf((float)(qn - e1));
    const float d2 = sqrtf((float)(qn - e2));
    ok = (double)d1 < ratio * (double)d2;
    train_idx[q] = id;
    dist1[q] = d1;
    dist2[q] = d2;
    good[q] = ok ? 1 : 0;
  }
  const unsigned long long ballot = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && ballot) atomicAdd(good_count, __popcll(ballot));
}

int choose_match_splits(int nq, int nt) {
  const long long waves = (nq + 31) / 32;
  const int ntiles = (nt + 31) / 32;
  long long s = (4096 + waves - 1) / waves;
  const int max_s = (ntiles + 7) / 8;  // at least 8 tiles per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  return (int)s;
}

}  // namespace

extern "C" {

int sfm_descriptors_to_i8(const void* desc, int32_t src_is_u8, int64_t m, int8_t* desc8, int32_t* norm, int32_t* bad, void* stream) {
  clear_stale_error();
  if (m < 0) return SFM_EINVAL;
  if (m == 0) return 0;
  if (!desc || !desc8 || !norm || !bad || ((uintptr_t)desc8 & 15u) != 0) return SFM_EINVAL;
  const long long blocks = (m + kBlock / 32 - 1) / (kBlock / 32);
  if (blocks > 0x7FFFFFFFLL) return SFM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(bad, 0, sizeof(int32_t), st);
  if (e != hipSuccess) return (int)e;
  if (src_is_u8)
    descriptors_to_i8_kernel<uint8_t><<<(unsigned)blocks, kBlock, 0, st>>>((const uint8_t*)desc, m, desc8, norm, bad);
  else
    descriptors_to_i8_kernel<float><<<(unsigned)blocks, kBlock, 0, st>>>((const float*)desc, m, desc8, norm, bad);
  return check_launch();
}

int64_t sfm_knn2_workspace_bytes(int32_t nq, int32_t nt) {
  if (nq <= 0 || nt <= 0) return 0;
  return (int64_t)choose_match_splits(nq, nt) * nq * (int64_t)sizeof(int4);
}

int sfm_knn2_ratio_i8(const int8_t* q8, const int32_t* qnorm, int32_t nq, const int8_t* t8, const int32_t* tnorm, int32_t nt, double ratio,
                      void* workspace, int64_t workspace_bytes, int32_t* train_idx, float* dist1, float* dist2, uint8_t* good,
                      int32_t* good_count, void* stream) {
  clear_stale_error();
  if (nq < 0 || nt < 2 || !(ratio >= 0.0)) return SFM_EINVAL;
  if (!good_count) return SFM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(good_count, 0, sizeof(int32_t), st);
  if (e != hipSuccess) return (int)e;
  if (nq == 0) return 0;
  if (!q8 || !qnorm || !t8 || !tnorm || !workspace || !train_idx || !dist1 || !dist2 || !good) return SFM_EINVAL;
  if ((((uintptr_t)q8 | (uintptr_t)t8 | (uintptr_t)workspace) & 15u) != 0) return SFM_EINVAL;
  if (workspace_bytes < sfm_knn2_workspace_bytes(nq, nt)) return SFM_EINVAL;
  const int splits = choose_match_splits(nq, nt);
  const int ntiles = (nt + 31) / 32;
  const int tiles_per_split = (ntiles + splits - 1) / splits;
  const int qblocks = (nq + 127) / 128;
  knn2_kernel<<<dim3(qblocks, splits), kBlock, 0, st>>>(q8, nq, t8, tnorm, nt, tiles_per_split, (int4*)workspace);
  int rc = check_launch();
  if (rc) return rc;
  knn2_finalize_kernel<<<(nq + kBlock - 1) / kBlock, kBlock, 0, st>>>((const int4*)workspace, splits, nq, qnorm, ratio, train_idx, dist1,
                                                                      dist2, good, good_count);
  return check_launch();
}

}  // extern "C"
