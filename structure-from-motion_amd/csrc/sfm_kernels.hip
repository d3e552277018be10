// HIP kernels and C ABI of libsfm_hip.so (gfx950 / MI355X only).
//
// Layout of the hot kernels: one hypothesis per lane.  A lane gathers its minimal sample,
// solves the model in registers (sfm_math.hpp) and then walks the correspondence array.
// The correspondence index is wave-uniform, so the compiler fetches each correspondence
// with scalar loads into SGPRs; the VALU only executes the residual arithmetic, with the
// model in VGPRs and the point as the scalar operand of each instruction.  Counts are
// accumulated per lane (compare + add-with-carry), so no cross-lane reduction is needed.
// That is the all-float64 form (eh_score_kernel, t_score_kernel).  The forms in use take everything that is a contraction
// of hypothesis-side coefficients with match-side monomials -- the bilinear form x2' E x1, the squared line norms, the rows
// H x1 and the translation residuals -- from the bf16 matrix cores (three-piece splits of the float64 values), decide a
// pair in float32 and re-evaluate it in float64 only inside a rigorous error band: eh_score_mfma2_kernel (exact E/H counts),
// t_score_mfma_kernel (exact translation counts), eh_upper_mfma2_kernel (upper bounds for the bulk arg-max).  The
// descriptor matcher in sfm_match.hip uses the int8 matrix cores.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "../../include/sfm_hip.h"
#include "sfm_math.hpp"

using namespace sfm;

namespace {

constexpr int kBlock = 256;
constexpr int kMaxCams = 64;

// Diagnostic build only (-DSFM_CLOCK_STAMPS, never the shipped library): wave 0..3 of every workgroup of the three
// matrix-core kernels stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its tile loop into a buffer
// of its own that no kernel reads; sfm_debug_clock_stamps copies it out.  held clock = cycles / ticks x 100 MHz.
// A record is 8 x uint64: cycles, ticks, tiles walked, start tick, end tick, (XCC_ID << 32) | HW_ID of the wave, the tick at
// which the wave entered the kernel, a count kept with SFM_STAMP_COUNT (lanes for which a condition held).
#ifdef SFM_CLOCK_STAMPS
constexpr int kStampSlots = 16384;
__device__ unsigned long long g_stamps[kStampSlots * 8];
#define SFM_STAMP_ENTRY()                                                        \
  const unsigned long long stamp_entry_ = __builtin_amdgcn_s_memrealtime(); \
  unsigned long long stamp_count_ = 0
#define SFM_STAMP_COUNT(cond) stamp_count_ += __popcll(__ballot(cond))
#define SFM_STAMP_BEGIN()                                                   \
  const unsigned long long stamp_r0_ = __builtin_amdgcn_s_memrealtime(); \
  const unsigned long long stamp_t0_ = __builtin_amdgcn_s_memtime()
#define SFM_STAMP_END(units)                                                                                                    \
  do {                                                                                                                          \
    const unsigned long long stamp_t1_ = __builtin_amdgcn_s_memtime();                                                          \
    const unsigned long long stamp_r1_ = __builtin_amdgcn_s_memrealtime();                                                      \
    if ((threadIdx.x & 63) == 0) {                                                                                              \
      const unsigned long long wg_ = blockIdx.x + (unsigned long long)gridDim.x * (blockIdx.y + (unsigned long long)gridDim.y * blockIdx.z); \
      unsigned long long* dst_ = g_stamps + 8 * ((wg_ * (blockDim.x >> 6) + (threadIdx.x >> 6)) % kStampSlots);                 \
      dst_[0] = stamp_t1_ - stamp_t0_;                                                                                          \
      dst_[1] = stamp_r1_ - stamp_r0_;                                                                                          \
      dst_[2] = (unsigned long long)(units);                                                                                    \
      dst_[3] = stamp_r0_;                                                                                                      \
      dst_[4] = stamp_r1_;                                                                                                      \
      dst_[5] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4); \
      dst_[6] = stamp_entry_;                                                                                                   \
      dst_[7] = stamp_count_;                                                                                                   \
    }                                                                                                                           \
  } while (0)
#else
#define SFM_STAMP_ENTRY() do { } while (0)
#define SFM_STAMP_COUNT(cond) do { } while (0)
#define SFM_STAMP_BEGIN() do { } while (0)
#define SFM_STAMP_END(units) do { } while (0)
#endif

struct CamOffsets {
  int64_t off[kMaxCams + 1];
};

// Sample indices come from the caller: they are clamped into [0, n) so that a bad index can never become an
// out-of-bounds read (the model of such a hypothesis is meaningless, as it would be in the reference's
// `x1[:, inds]` with a wrapped negative index).
__device__ __forceinline__ int clamp_index(int i, int n) { return min(max(i, 0), n - 1); }

template <typename T>
__device__ __forceinline__ void gather_samples(const double4* __restrict__ corr, int n, const int32_t* __restrict__ row, T (&p8)[8][4],
                                               T (&p4)[4][4]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const double4 c = corr[clamp_index(row[i], n)];
    p8[i][0] = T(c.x); p8[i][1] = T(c.y); p8[i][2] = T(c.z); p8[i][3] = T(c.w);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double4 c = corr[clamp_index(row[8 + i], n)];
    p4[i][0] = T(c.x); p4[i][1] = T(c.y); p4[i][2] = T(c.z); p4[i][3] = T(c.w);
  }
}

// Index sets generated on the device from (seed, global hypothesis id): width 12 for the E/H
// stage (8 + 4 matches), width 2 for the translation stage.
template <int kWidth>
__global__ __launch_bounds__(kBlock) void index_sets_kernel(unsigned long long seed, unsigned long long first_id, long long nb, unsigned n,
                                                            int32_t* __restrict__ idx) {
  const long long b = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (b >= nb) return;
  int32_t row[kWidth];
  philox_index_row<kWidth>(seed, first_id + (unsigned long long)b, n, row);
  if constexpr (kWidth % 4 == 0) {
    int4* dst = reinterpret_cast<int4*>(idx + b * kWidth);
#pragma unroll
    for (int j = 0; j < kWidth / 4; ++j) dst[j] = make_int4(row[4 * j], row[4 * j + 1], row[4 * j + 2], row[4 * j + 3]);
  } else {
    int2* dst = reinterpret_cast<int2*>(idx + b * kWidth);
#pragma unroll
    for (int j = 0; j < kWidth / 2; ++j) dst[j] = make_int2(row[2 * j], row[2 * j + 1]);
  }
}

// ---------------------------------------------------------------------------------------
// E/H hypothesis scoring.  grid = (hypothesis tiles, point splits).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void eh_score_kernel(const double4* __restrict__ corr, int n, const int32_t* __restrict__ idx,
                                                          long long nb, double two_thr_e2, double thr_h2, int* __restrict__ count_e,
                                                          int* __restrict__ count_h, int points_per_split, int use_atomics) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long h = gid < nb ? gid : nb - 1;  // tail lanes repeat the last hypothesis: every lane stays active
  double E[9], H[9];
  {
    double p8[8][4], p4[4][4];
    gather_samples<double>(corr, n, idx + 12 * h, p8, p4);
    solve_E8<double>(p8, E);
    solve_H4<double>(p4, H);
  }
  const int j0 = blockIdx.y * points_per_split;
  const int j1 = min(n, j0 + points_per_split);
  int ce = 0, ch = 0;
#pragma unroll 4
  for (int j = j0; j < j1; ++j) {
    const double4 p = corr[j];
    ce += epipolar_inlier<double>(E, p.x, p.y, p.z, p.w, two_thr_e2) ? 1 : 0;
    ch += transfer_inlier<double>(H, p.x, p.y, p.z, p.w, thr_h2) ? 1 : 0;
  }
  if (gid < nb) {
    if (use_atomics) {
      atomicAdd(count_e + gid, ce);
      atomicAdd(count_h + gid, ch);
    } else {
      count_e[gid] = ce;
      count_h[gid] = ch;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Float32 pre-filter: UPPER BOUNDS of the inlier counts.
// The models are solved in float64 exactly as above, scaled to max-abs 1 (both tests are
// homogeneous in the model, so the scale is free and float32 cannot overflow) and rounded to
// float32; the matches are read from a float32 copy.  Each test is replaced by its
// upper-bound form (sfm_math.hpp) whose slack constants are derived per hypothesis from a
// first-order float32 error bound (input rounding of matches and model + the roundings of
// the fma chains, doubled), so that every float64 inlier is also counted here: up >= exact.
// Callers re-score the few hypotheses whose upper bound reaches the best exact count with
// eh_score_kernel and so obtain the exact winner at about half the VALU cost per residual.
// ---------------------------------------------------------------------------------------
constexpr double kEps32 = 5.9604644775390625e-08;  // 2^-24

__device__ __forceinline__ double to_unit_f32(const double (&M)[9], float (&Mf)[9], double (&Ma)[9]) {
  double mx = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) mx = fmax(mx, fabs(M[i]));
  const double inv = (mx > 0.0 && mx < 1.0e300) ? 1.0 / mx : 1.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    Mf[i] = (float)(M[i] * inv);
    Ma[i] = fabs(M[i] * inv);
  }
  return inv;
}

__device__ __forceinline__ float round_up_f32(double v) {
  const float f = (float)v;
  return ((double)f < v) ? nextafterf(f, INFINITY) : f;
}

__device__ __forceinline__ unsigned f32_up_bits(double v) { return __float_as_uint(round_up_f32(v)); }

// max |coordinate| over the matches, measured on the device: each lane folds its match into a float32 bit pattern
// (rounded up; non-negative floats order like their bits and a NaN orders above infinity), one unsigned atomicMax per
// wave.  Every lane of the wave must call this (the reduction shuffles across all 64 lanes); *dst starts at zero.
__device__ __forceinline__ void publish_absmax(const double4& c, bool live, unsigned* __restrict__ dst) {
  unsigned m = 0u;
  if (live) m = max(max(f32_up_bits(fabs(c.x)), f32_up_bits(fabs(c.y))), max(f32_up_bits(fabs(c.z)), f32_up_bits(fabs(c.w))));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(dst, m);
}

// The coordinate bound the error analysis uses: the larger of the caller's hint and the measured maximum, so an
// understated hint cannot void a bound.  Infinity = unusable (a NaN or absurdly large coordinate): callers then report
// trivial bounds or evaluate in float64.
__device__ __forceinline__ double usable_xmax(double hint, unsigned measured_bits) {
  const double m = (double)__uint_as_float(measured_bits);
  double x = (hint > m) ? hint : m;  // a NaN measurement stays NaN
  if (x < 0x1.0p-20) x = 0x1.0p-20;
  return (x < 1.0e6) ? x : (double)INFINITY;
}

struct UpperModel {
  float E[9], H[9];
  float kp_e, c2_e, T_h, c1_h, c0_h;
  float k1_e, c_e;  // one-sided (loose) epipolar form
  float rt_h, g_h;  // box (loose) transfer form
};

// solves hypothesis h in float64 and derives the float32 models and slack constants
__device__ __forceinline__ void solve_upper_model(const double4* __restrict__ corr, int n, const int32_t* __restrict__ row, double two_thr_e2,
                                                  double thr_h2, double xmax, double safety, UpperModel& m) {
  double Ed[9], Hd[9], A[9];
  double p8[8][4], p4[4][4];
  gather_samples<double>(corr, n, row, p8, p4);
  solve_E8<double>(p8, Ed);
  solve_H4<double>(p4, Hd);
  to_unit_f32(Ed, m.E, A);
  {
    // row sums (for l = E x1) and column sums (for m = E^T x2) of |E| with |x|,|y| <= xmax
    const double R0 = xmax * (A[0] + A[1]) + A[2], R1 = xmax * (A[3] + A[4]) + A[5], R2 = xmax * (A[6] + A[7]) + A[8];
    const double C0 = xmax * (A[0] + A[3]) + A[6], C1 = xmax * (A[1] + A[4]) + A[7];
    const double Es = 7.0 * kEps32 * (xmax * (R0 + R1) + R2);
    const double eL = 6.0 * kEps32 * (R0 + R1 + C0 + C1);
    const double rk = sqrt(two_thr_e2);
    const double G = safety * (Es + rk * eL);
    const double lam = rk;
    m.kp_e = round_up_f32(two_thr_e2 * (1.0 + G / lam) * (1.0 + 16.0 * kEps32));
    m.c2_e = round_up_f32((G * lam + G * G) * (1.0 + 16.0 * kEps32));
    const double mu = 1.0e-3;
    m.k1_e = round_up_f32(two_thr_e2 * (1.0 + mu) * (1.0 + 16.0 * kEps32));
    m.c_e = round_up_f32(G * G * (1.0 + 1.0 / mu) * (1.0 + 16.0 * kEps32));
  }
  to_unit_f32(Hd, m.H, A);
  {
    const double R0 = xmax * (A[0] + A[1]) + A[2], R1 = xmax * (A[3] + A[4]) + A[5], R2 = xmax * (A[6] + A[7]) + A[8];
    const double ez = 4.0 * kEps32 * R2;
    const double Dx = 6.0 * kEps32 * (R0 + xmax * R2), Dy = 6.0 * kEps32 * (R1 + xmax * R2);
    const double rt = sqrt(thr_h2);
    const double G = safety * (rt * ez + Dx + Dy);
    m.T_h = round_up_f32(thr_h2 * (1.0 + 16.0 * kEps32));
    m.c1_h = round_up_f32(2.0 * rt * G * (1.0 + 16.0 * kEps32));
    m.c0_h = round_up_f32(G * G * (1.0 + 16.0 * kEps32));
    m.rt_h = round_up_f32(rt * (1.0 + 16.0 * kEps32));
    m.g_h = round_up_f32(G * (1.0 + 16.0 * kEps32));
  }
}

// LDS-staged form of the pre-filter.  On gfx950 a float32 VALU instruction that reads an SGPR
// operand issues at half rate (4 cycles per wave instead of 2; profiles/microbench/valu_rates),
// which taxes 20 of the 38 instructions of a residual pair when the match comes from scalar
// loads.  Here the workgroup stages tiles of matches in LDS (coalesced global loads, double
// buffered, one barrier per tile) and every lane reads the current match with a broadcast
// ds_read_b128 into VGPRs, so the whole residual runs on VGPR operands at full rate.
template <int kTilePoints>
__global__ __launch_bounds__(kBlock) void eh_upper_lds_kernel(const double4* __restrict__ corr, const float4* __restrict__ corr32, int n,
                                                              const int32_t* __restrict__ idx, long long nb, double two_thr_e2, double thr_h2,
                                                              double xmax, double safety, int* __restrict__ up_e, int* __restrict__ up_h,
                                                              int points_per_split, int use_atomics) {
  __shared__ float4 tile[2][kTilePoints];
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long h = gid < nb ? gid : nb - 1;
  const double xm = usable_xmax(xmax, __float_as_uint(corr32[n].x));
  UpperModel um;
  solve_upper_model(corr, n, idx + 12 * h, two_thr_e2, thr_h2, xm, safety, um);
  const float(&E)[9] = um.E;
  const float(&H)[9] = um.H;
  const float kp_e = um.kp_e, c2_e = um.c2_e, T_h = um.T_h, c1_h = um.c1_h, c0_h = um.c0_h;
  const int j0 = blockIdx.y * points_per_split;
  const int j1 = min(n, j0 + points_per_split);
  const float qnan = __builtin_nanf("");
  const float4 pad = make_float4(qnan, qnan, qnan, qnan);  // a NaN match fails both tests
  constexpr int kPerThread = kTilePoints / kBlock;
  float4 stage[kPerThread];
  auto fetch = [&](int base) {
#pragma unroll
    for (int r = 0; r < kPerThread; ++r) {
      const int j = base + r * kBlock + (int)threadIdx.x;
      stage[r] = (j < j1) ? corr32[j] : pad;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int r = 0; r < kPerThread; ++r) tile[buf][r * kBlock + threadIdx.x] = stage[r];
  };
  int ce = 0, ch = 0;
  fetch(j0);
  commit(0);
  __syncthreads();
  int buf = 0;
  for (int base = j0; base < j1; base += kTilePoints) {
    const bool more = base + kTilePoints < j1;
    if (more) fetch(base + kTilePoints);  // global loads in flight while this tile is scored
    const float4* __restrict__ t = tile[buf];
#pragma clang loop vectorize(disable) interleave(disable)
#pragma unroll 8
    for (int p = 0; p < kTilePoints; ++p) {
      const float4 q = t[p];
      ce += epipolar_upper_f32(E, q.x, q.y, q.z, q.w, kp_e, c2_e) ? 1 : 0;
      ch += transfer_upper_f32(H, q.x, q.y, q.z, q.w, T_h, c1_h, c0_h) ? 1 : 0;
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (!(xm < 1.0e6)) ce = ch = max(0, j1 - j0);  // no usable coordinate bound: the trivial bound
  if (gid < nb) {
    if (use_atomics) {
      atomicAdd(up_e + gid, ce);
      atomicAdd(up_h + gid, ch);
    } else {
      up_e[gid] = ce;
      up_h[gid] = ch;
    }
  }
}

template <bool kOneSided>
__global__ __launch_bounds__(kBlock) void eh_upper_kernel(const double4* __restrict__ corr, const float4* __restrict__ corr32, int n,
                                                          const int32_t* __restrict__ idx, long long nb, double two_thr_e2, double thr_h2,
                                                          double xmax, double safety, int* __restrict__ up_e, int* __restrict__ up_h,
                                                          int points_per_split, int use_atomics) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long h = gid < nb ? gid : nb - 1;
  const double xm = usable_xmax(xmax, __float_as_uint(corr32[n].x));
  UpperModel um;
  solve_upper_model(corr, n, idx + 12 * h, two_thr_e2, thr_h2, xm, safety, um);
  const float(&E)[9] = um.E;
  const float(&H)[9] = um.H;
  const float kp_e = um.kp_e, c2_e = um.c2_e, T_h = um.T_h, c1_h = um.c1_h, c0_h = um.c0_h;
  const int j0 = blockIdx.y * points_per_split;
  const int j1 = min(n, j0 + points_per_split);
  int ce = 0, ch = 0;
  // the compiler packs two matches per v_pk_fma_f32 here (22 VALU instructions per pair instead of 38)
  const float k1_e = um.k1_e, c_e = um.c_e, rt_h = um.rt_h, g_h = um.g_h;
#pragma unroll 8
  for (int j = j0; j < j1; ++j) {
    const float4 p = corr32[j];
    if constexpr (kOneSided) {
      ce += epipolar_upper_onesided_f32(E, p.x, p.y, p.z, p.w, k1_e, c_e) ? 1 : 0;
      ch += transfer_upper_box_f32(H, p.x, p.y, p.z, p.w, rt_h, g_h) ? 1 : 0;
    } else {
      ce += epipolar_upper_f32(E, p.x, p.y, p.z, p.w, kp_e, c2_e) ? 1 : 0;
      ch += transfer_upper_f32(H, p.x, p.y, p.z, p.w, T_h, c1_h, c0_h) ? 1 : 0;
    }
  }
  if (!(xm < 1.0e6)) ce = ch = max(0, j1 - j0);  // no usable coordinate bound: the trivial bound
  if (gid < nb) {
    if (use_atomics) {
      atomicAdd(up_e + gid, ce);
      atomicAdd(up_h + gid, ch);
    } else {
      up_e[gid] = ce;
      up_h[gid] = ch;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Matrix-core form of the loose pre-filter.  The products l = E x1 and (hx, hy, hz) = H x1 over
// all (hypothesis, match) pairs are two [3B x 3] x [3 x N] GEMMs, so they run on the bf16 matrix
// cores at float32-like accuracy: every model entry a and every coordinate x is split into three
// bf16 pieces (a = a0 + a1 + a2 + r, |r| <= 2^-26 |a|) and a.x is expanded into the six piece
// products of weight >= 2^-18 (a0x0, a0x1, a1x0, a0x2, a2x0, a1x1); with the constant third
// coordinate one row of the product is a K = 6 + 6 + 3 = 15 contraction = ONE
// v_mfma_f32_32x32x16_bf16 per 32 matches x 32 hypotheses, a quarter of the VALU time of the
// two fmas it replaces.  The VALU keeps only what is not GEMM shaped (s = l.x2, |l|^2, the
// box test): 12 instead of 26 instructions per pair, all on VGPR operands.
//
// A wave solves 64 hypotheses (one per lane, float64, as everywhere else), publishes their
// piece vectors through LDS and then owns two column blocks of 32 hypotheses; in the 32x32
// accumulator layout a lane holds ONE hypothesis column and 16 matches, so the per-hypothesis
// constants stay per-lane values and the counts need one cross-half add at the end.  Test
// outcomes are collected as sign bits: one v_alignbit per test shifts the sign of (lhs - rhs)
// into a history register and one popcount per 16 tests adds them up.
//
// Measured alternatives (same results, slower): hypotheses as rows with scalar popcounts of the
// compare masks (the one scalar ALU per CU becomes the bottleneck: 4.6 ms against 3.5 ms on the
// 10^4 x 10^6 workload) and a software pipeline that issues the next stage's MFMAs before the
// current stage's vector work (4.0 ms at 251 VGPRs).  In-kernel clock stamps show 1.95-2.0 GHz
// under this kernel; what counts is the number of ISSUE SLOTS: v_alignbit, v_min3, v_cmp issue at
// half rate on gfx950, which is why the default form below counts with clamped floats instead.
//
// Error bound used for the slack (upper-bound property): piece expansion (dropped products and
// residuals) <= 2^-24 R, accumulation inside the matrix core <= 32 * 2^-24 R for any summation
// order with rounding or truncation to float32 at every step (profiles/microbench/mfma_bf16_accum
// measures <= 4.7 * 2^-24 on this chip), R = sum of |a||x| over the row.
// ---------------------------------------------------------------------------------------
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr double kEpsMfma = 34.0 * kEps32;

__device__ __forceinline__ unsigned bf16_rn_bits(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}

// v = p[0] + p[1] + p[2] + r with bf16 pieces p and |r| <= 2^-26 |v|
__device__ __forceinline__ void split3_bf16(double v, unsigned (&p)[3]) {
  double r = v;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    p[i] = bf16_rn_bits((float)r);
    r -= (double)__uint_as_float(p[i] << 16);
  }
}

// the 16 bf16 of one operand vector as two uint4 (k = 0..7, 8..15); `model` selects the
// hypothesis-side order (a0 a0 a1 a0 a2 a1 | b.. | c0 c1 c2 0), else the match-side order
// (x0 x1 x0 x2 x0 x1 | y.. | 1 1 1 0), so that element k of one times element k of the other
// enumerates the kept piece products.
__device__ __forceinline__ void pack_pieces(double a, double b, double c, bool model, uint4& lo, uint4& hi) {
  unsigned pa[3], pb[3], pc[3];
  split3_bf16(a, pa);
  split3_bf16(b, pb);
  unsigned e[16];
  if (model) {
    split3_bf16(c, pc);
    e[0] = pa[0]; e[1] = pa[0]; e[2] = pa[1]; e[3] = pa[0]; e[4] = pa[2]; e[5] = pa[1];
    e[6] = pb[0]; e[7] = pb[0]; e[8] = pb[1]; e[9] = pb[0]; e[10] = pb[2]; e[11] = pb[1];
    e[12] = pc[0]; e[13] = pc[1]; e[14] = pc[2]; e[15] = 0;
  } else {
    e[0] = pa[0]; e[1] = pa[1]; e[2] = pa[0]; e[3] = pa[2]; e[4] = pa[0]; e[5] = pa[1];
    e[6] = pb[0]; e[7] = pb[1]; e[8] = pb[0]; e[9] = pb[2]; e[10] = pb[0]; e[11] = pb[1];
    e[12] = 0x3F80u; e[13] = 0x3F80u; e[14] = 0x3F80u; e[15] = 0;
  }
  lo = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
  hi = make_uint4(e[8] | (e[9] << 16), e[10] | (e[11] << 16), e[12] | (e[13] << 16), e[14] | (e[15] << 16));
}

// Uniform constants of the matrix-core pre-filter.  Both tests are homogeneous in the model, so every model is scaled
// such that ITS error bound G equals the fixed kGe / kGh; with a fixed mu the slack constants are then the same for all
// hypotheses.  Dividing the one-sided epipolar test s^2 - c < k1 n1 by k1 = k (1 + mu) moves k1 into the match side:
//     (l0 x2' + l1 y2' + l2')^2 - c' < l0^2 + l1^2,   x2' = x2 / sqrt(k1), y2' = y2 / sqrt(k1), l2' = l2 / sqrt(k1), c' = c / k1
// (three fmas after s instead of four instructions); the box test is divided by sqrt(k1) as well, so that one scaled
// copy of (x2, y2) serves both:  |x2' hz - hx'| < rt' |hz| + g'  with hx' = hx / sqrt(k1), rt' = rt / sqrt(k1), g' = g / sqrt(k1).
constexpr double kGe = 0x1.0p-18, kGh = 0x1.0p-18, kMuE = 0.025;
__host__ __device__ inline double uniform_k1(double two_thr_e2) { return two_thr_e2 * (1.0 + kMuE) * (1.0 + 32.0 * kEps32); }

// Match-side operands for the matrix-core pre-filter: pieces[npad][2] (the 16 bf16 of (x1, y1, 1)) and
// xy[npad] = float32 (x2, y2) / sqrt(k1); npad = n rounded up to 32 (the scoring kernel masks the padding); then a
// 16-byte trailer: the threshold tag (double) and the measured max |coordinate| (float32 bits).
__global__ __launch_bounds__(kBlock) void corr_to_mfma_kernel(const double4* __restrict__ corr, int n, int npad, double inv_sqrt_k1,
                                                              uint4* __restrict__ pieces, float2* __restrict__ xy) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) *reinterpret_cast<double*>(xy + npad) = inv_sqrt_k1;  // tag: which threshold these operands were prepared for
  double4 c = make_double4(0.0, 0.0, 0.0, 0.0);
  if (j < n) {
    c = corr[j];
    uint4 lo, hi;
    pack_pieces(c.x, c.y, 1.0, false, lo, hi);
    pieces[2 * j] = lo;
    pieces[2 * j + 1] = hi;
    xy[j] = make_float2((float)(c.z * inv_sqrt_k1), (float)(c.w * inv_sqrt_k1));
  } else if (j < npad) {
    pieces[2 * j] = make_uint4(0, 0, 0, 0);
    pieces[2 * j + 1] = make_uint4(0, 0, 0, 0);
    xy[j] = make_float2(0.0f, 0.0f);
  }
  publish_absmax(c, j < n, reinterpret_cast<unsigned*>(xy + npad) + 2);  // behind the tag; zeroed by the launcher
}

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// Sign histories of the MFMA kernels hold 16 matches per lane half (bit 15-i = match 16*half + i of a tile of 32); the
// only tile with padding is tile n/32 when n is not a multiple of 32: this is its mask of real matches.
__device__ __forceinline__ unsigned tile_tail_mask(int n, int half) {
  const int nv = min(16, max(0, (n & 31) - 16 * half));
  return (0xFFFFu << (16 - nv)) & 0xFFFFu;
}

// min(m, |a|, |b|) in one instruction (m >= 0); a NaN operand is ignored, as by fminf
// the lanes whose condition holds (HIP's __ballot takes an int: the bool would first become 0 / 1 through a v_cndmask)
__device__ __forceinline__ unsigned long long ballot64(bool c) { return __builtin_amdgcn_ballot_w64(c); }

__device__ __forceinline__ float min3_abs(float m, float a, float b) {
  float r;
  asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
}

// kFloatCount: test outcomes are produced as 0.0f / 1.0f by the output clamp of the test's last instruction and summed
// with float adds (both full-rate instructions; v_alignbit_b32, like v_min/v_max/v_add3/v_cmp, issues at half rate on
// gfx950: profiles/microbench/valu_issue_probe).  For that the models are scaled by 2^40 (E) and 2^44 (H) on top of the
// uniform-band scaling, which shifts exponents only: a test value is then zero, negative, or at least 1 in magnitude
// (the constants c' 2^80 >= 2^50 and g' 2^44 >= 2^26 put every float that can cancel against them on a grid coarser
// than 1), so clamp(value) is exactly 1 for a pass and 0 otherwise.  The launcher selects this form only for thresholds
// with 2^-45 < k1 < 0.6, where neither the squares (<= 0.7 / k1 and (0.84 / xmax)^2, times 2^80) can overflow nor c'
// fall below 2^-30; the sign-history form serves every other threshold.
constexpr double kLambdaE = 0x1.0p40, kLambdaH = 0x1.0p44;

// clamp(c - a a) and clamp(b - |a|) to [0, 1], each ONE full-rate instruction: the clamp is the output modifier of the
// fma / subtraction itself (a NaN becomes 0: kernels run with DX10_CLAMP set)
__device__ __forceinline__ float clamp01_c_minus_sq(float a, float c) {
  float r;
  asm("v_fma_f32 %0, -%1, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(c));
  return r;
}
__device__ __forceinline__ float clamp01_b_minus_abs(float b, float a) {
  float r;
  asm("v_sub_f32_e64 %0, %1, |%2| clamp" : "=v"(r) : "v"(b), "v"(a));
  return r;
}

template <bool kDoE, bool kDoH, bool kFloatCount>
__global__ __launch_bounds__(kBlock, 3) void eh_upper_mfma_kernel(const double4* __restrict__ corr, const uint4* __restrict__ pieces,
                                                                  const float2* __restrict__ xy, int n, const int32_t* __restrict__ idx,
                                                                  long long nb, double two_thr_e2, double thr_h2, double xmax, double safety,
                                                                  int* __restrict__ up_e, int* __restrict__ up_h, int tiles_per_split,
                                                                  int use_atomics) {
  __shared__ unsigned char fin[kBlock / 64][64];
  __shared__ uint4 bstage[kBlock / 64][64][6][2];
  SFM_STAMP_ENTRY();
  // the solve is a long serial chain per wave: it runs ahead of the other waves' tile loops (which saturate the vector
  // issue and would starve it otherwise), so that all resident waves are inside their tile loops most of the time
  __builtin_amdgcn_s_setprio(3);
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long hyp = gid < nb ? gid : nb - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const double inv_sqrt_k1 = 1.0 / sqrt(uniform_k1(two_thr_e2));
  const int npad_all = ((n + 31) / 32) * 32;
  const double xm = usable_xmax(xmax, reinterpret_cast<const unsigned*>(xy + npad_all)[2]);
  {
    double Ed[9], Hd[9], A[9];
    float unused[9];
    {
      double p8[8][4], p4[4][4];
      gather_samples<double>(corr, n, idx + 12 * hyp, p8, p4);
      solve_E8<double>(p8, Ed);
      solve_H4<double>(p4, Hd);
    }
    // A model with a non-finite entry has the float64 count 0 (every comparison with NaN is false) but an undefined
    // sign pattern here: its count is replaced by 0 when it is written.
    {
      double chk_e = 0.0, chk_h = 0.0;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        chk_e = fma(Ed[i], 0.0, chk_e);  // NaN as soon as one entry is NaN or infinite
        chk_h = fma(Hd[i], 0.0, chk_h);
      }
      fin[wave][lane] = (unsigned char)((chk_e == 0.0 ? 1 : 0) | (chk_h == 0.0 ? 2 : 0));  // parked in LDS: no register lives across the loop
    }
    double inv = to_unit_f32(Ed, unused, A);
    {
      const double R0 = xm * (A[0] + A[1]) + A[2], R1 = xm * (A[3] + A[4]) + A[5], R2 = xm * (A[6] + A[7]) + A[8];
      // |s - s_true|: errors of l0, l1, l2 carried through s = l0 x2 + l1 y2 + l2, float32 rounding of x2, y2 and of the two fmas
      const double Es = (kEpsMfma + 4.0 * kEps32) * (xm * (R0 + R1) + R2);
      // |L - L_true|, L = |(l0, l1)|: errors of l0 and l1 plus the float32 evaluation of l0^2 + l1^2
      const double eL = (kEpsMfma + 2.0 * kEps32) * (R0 + R1);
      const double G = safety * (Es + sqrt(two_thr_e2) * eL);
      inv *= kGe / G;  // now this model's G is kGe (G > 0: the unit-scaled model has an entry of magnitude 1)
      if constexpr (kFloatCount) inv *= kLambdaE;
    }
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const double sc = v == 2 ? inv * inv_sqrt_k1 : inv;
      pack_pieces(Ed[3 * v] * sc, Ed[3 * v + 1] * sc, Ed[3 * v + 2] * sc, true, bstage[wave][lane][v][0], bstage[wave][lane][v][1]);
    }
    inv = to_unit_f32(Hd, unused, A);
    {
      const double R0 = xm * (A[0] + A[1]) + A[2], R1 = xm * (A[3] + A[4]) + A[5], R2 = xm * (A[6] + A[7]) + A[8];
      const double ez = kEpsMfma * R2;
      const double Dx = (kEpsMfma + 3.0 * kEps32) * (R0 + xm * R2), Dy = (kEpsMfma + 3.0 * kEps32) * (R1 + xm * R2);
      const double G = safety * (sqrt(thr_h2) * ez + Dx + Dy);
      inv *= kGh / G;
      if constexpr (kFloatCount) inv *= kLambdaH;
    }
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const double sc = v == 2 ? inv : inv * inv_sqrt_k1;
      pack_pieces(Hd[3 * v] * sc, Hd[3 * v + 1] * sc, Hd[3 * v + 2] * sc, true, bstage[wave][lane][3 + v][0], bstage[wave][lane][3 + v][1]);
    }
  }
  __syncthreads();
  bf16x8 B[2][6];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int v = 0; v < 6; ++v) B[k][v] = as_bf16x8(bstage[wave][32 * k + col][v][half]);
  // c' = G^2 (1 + 1/mu) / k1, rt' = sqrt(T) / sqrt(k1), g' = G / sqrt(k1), each rounded up with room for the float32 roundings
  const float cprime = round_up_f32(kGe * kGe * (1.0 + 1.0 / kMuE) * inv_sqrt_k1 * inv_sqrt_k1 * (1.0 + 32.0 * kEps32)) *
                       (kFloatCount ? (float)(kLambdaE * kLambdaE) : 1.0f);  // the scale factors are powers of two: exact
  const float rtp = round_up_f32(sqrt(thr_h2) * inv_sqrt_k1 * (1.0 + 16.0 * kEps32));
  const float gp = round_up_f32(kGh * inv_sqrt_k1 * (1.0 + 16.0 * kEps32)) * (kFloatCount ? (float)kLambdaH : 1.0f);
  // accumulator register i of a lane in half h is matrix row (i&3) + 8(i>>2) + 4h; the A row that
  // lane `col` supplies is loaded from the match whose position in the tile is 16h' + i for that row,
  // so that half h scores the 16 consecutive matches 16h .. 16h+15 of the tile
  const int apos = 16 * ((col >> 2) & 1) + (col & 3) + 4 * (col >> 3);
  const int ntiles = (n + 31) / 32;
  const int t0 = blockIdx.y * tiles_per_split;
  const int t1 = min(ntiles, t0 + tiles_per_split);
  int ce[2] = {0, 0}, ch[2] = {0, 0};
  unsigned he[2] = {0, 0}, hh[2] = {0, 0};  // sign-bit histories (kFloatCount = false): one v_alignbit per test
  float fe[2] = {0.0f, 0.0f}, fh[2] = {0.0f, 0.0f};  // float counts (kFloatCount = true): exact integers below 2^24
  const f32x16 zero = {0};
  // bit 15-i of a history is match 16*half + i of a tile; only tile n/32 (when n is not a multiple of 32) has padding to mask out
  const unsigned tail_mask = tile_tail_mask(n, half);
  const uint4* asrc = pieces + 2 * (t0 * 32 + apos) + half;
  const float4* xsrc = reinterpret_cast<const float4*>(xy + t0 * 32 + 16 * half);
  // Match-side operands of the current tile: loaded one tile ahead.  The loads of tile t + 1 are issued inside the last
  // of the four test blocks of tile t, each into the register its last reader has just released (A after the block's
  // MFMAs, (x2, y2) pair by pair), so a wave meets no load latency at the top of a tile: with three waves per SIMD, whose
  // combined issue rate barely exceeds what the SIMD can take, every stall of a wave is lost issue time.
  bf16x8 A = as_bf16x8(*asrc);
  float x2[16], y2[16];
  auto load_xy = [&](const float4* q, int pair) {
    const float4 v = q[pair];
    x2[2 * pair] = v.x; y2[2 * pair] = v.y; x2[2 * pair + 1] = v.z; y2[2 * pair + 1] = v.w;
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) load_xy(xsrc, i);
  // one tile of 32 matches against the wave's 64 hypotheses; kMasked: the tile with padding (float counts are weighted
  // with the 0/1 mask of its real matches, at the same instruction count); anext / xnext: where the next tile's operands are
  auto score_tile = [&](auto masked_tag, const uint4* anext, const float4* xnext) {
    constexpr bool kMasked = decltype(masked_tag)::value;
    constexpr bool kLastE = kDoE && !kDoH;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if constexpr (kDoE) {
        const f32x16 l0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[k][0], zero, 0, 0, 0);
        const f32x16 l1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[k][1], zero, 0, 0, 0);
        const f32x16 l2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[k][2], zero, 0, 0, 0);
        if (kLastE && k == 1) A = as_bf16x8(*anext);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float s = fma(l0[i], x2[i], fma(l1[i], y2[i], l2[i]));  // s / sqrt(k1)
          if (kLastE && k == 1 && (i & 1)) load_xy(xnext, i >> 1);
          if constexpr (kFloatCount) {
            const float t = clamp01_c_minus_sq(s, fma(l1[i], l1[i], fma(l0[i], l0[i], cprime)));  // 1 <=> inlier (or within the slack)
            if constexpr (kMasked)
              fe[k] = fma(t, ((tail_mask >> (15 - i)) & 1u) ? 1.0f : 0.0f, fe[k]);
            else
              fe[k] += t;
          } else {
            const float d = fma(s, s, fma(-l1[i], l1[i], fma(-l0[i], l0[i], -cprime)));  // inlier (or within the slack) <=> d < 0
            he[k] = __builtin_amdgcn_alignbit(he[k], __float_as_uint(d), 31);  // shift the sign bit into the history
          }
        }
        if constexpr (!kFloatCount) ce[k] += __popc(he[k] & (kMasked ? tail_mask : 0xFFFFu));
      }
      if constexpr (kDoH) {
        const f32x16 hx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[k][3], zero, 0, 0, 0);
        const f32x16 hy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[k][4], zero, 0, 0, 0);
        const f32x16 hz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[k][5], zero, 0, 0, 0);
        if (k == 1) A = as_bf16x8(*anext);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float dx = fma(x2[i], hz[i], -hx[i]);
          const float dy = fma(y2[i], hz[i], -hy[i]);
          if (k == 1 && (i & 1)) load_xy(xnext, i >> 1);
          const float lim = fma(rtp, fabsf(hz[i]), gp);
          if constexpr (kFloatCount) {
            const float tx = clamp01_b_minus_abs(lim, dx), ty = clamp01_b_minus_abs(lim, dy);  // both 1 <=> inside the box
            if constexpr (kMasked)
              fh[k] = fma(tx * ty, ((tail_mask >> (15 - i)) & 1u) ? 1.0f : 0.0f, fh[k]);
            else
              fh[k] = fma(tx, ty, fh[k]);
          } else {
            const float d = fmaxf(fabsf(dx), fabsf(dy)) - lim;
            hh[k] = __builtin_amdgcn_alignbit(hh[k], __float_as_uint(d), 31);
          }
        }
        if constexpr (!kFloatCount) ch[k] += __popc(hh[k] & (kMasked ? tail_mask : 0xFFFFu));
      }
    }
  };
  __builtin_amdgcn_s_setprio(0);
  SFM_STAMP_BEGIN();
  const int t_partial = (n & 31) ? (n >> 5) : ntiles;  // the tile with padding, if any
  const int t_full = min(t1, t_partial);
  for (int t = t0; t < t_full; ++t) {
    const bool more = t + 1 < t1;  // the last tile of the range loads itself again (in bounds, unused)
    const uint4* anext = more ? asrc + 64 : asrc;
    const float4* xnext = more ? xsrc + 16 : xsrc;
    score_tile(std::false_type{}, anext, xnext);
    asrc = anext;
    xsrc = xnext;
  }
  if (t_partial >= t0 && t_partial < t1) score_tile(std::true_type{}, asrc, xsrc);
  if constexpr (kFloatCount) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      ce[k] = (int)fe[k];
      ch[k] = (int)fh[k];
    }
  }
  SFM_STAMP_END(t1 - t0);
  // the two halves of the wave scored different matches for the same 32 hypotheses
  int se[2], sh[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    se[k] = ce[k] + __shfl_xor(ce[k], 32);
    sh[k] = ch[k] + __shfl_xor(ch[k], 32);
  }
  // lane L reports hypothesis L = 32*half + col, the one it solved itself
  const unsigned ok = fin[wave][lane];
  int oute = (ok & 1u) ? (half ? se[1] : se[0]) : 0, outh = (ok & 2u) ? (half ? sh[1] : sh[0]) : 0;
  // operands prepared for another threshold, or matches without a usable coordinate bound, cannot give valid bounds:
  // report the trivial bound (all matches of this split)
  if (*reinterpret_cast<const double*>(xy + npad_all) != inv_sqrt_k1 || !(xm < 1.0e6)) oute = outh = max(0, min(n, t1 * 32) - t0 * 32);
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // the hypothesis id is formed again here rather than held in two registers across the tile loop
  const long long out_id = (long long)blockIdx.x * blockDim.x + tid;
  if (out_id < nb) {
    if (use_atomics) {
      atomicAdd(up_e + out_id, oute);
      atomicAdd(up_h + out_id, outh);
    } else {
      up_e[out_id] = oute;
      up_h[out_id] = outh;
    }
  }
}

// ---------------------------------------------------------------------------------------
// EXACT E/H counts (those of eh_score_kernel) at matrix-core + float32 cost, FIRST FORM (launched only with
// SFM_HIP_EHX_FORM=0 or for more than 2^25 matches; eh_score_mfma2_kernel below is the form in use): the scheme of
// t_score_mfma_kernel applied to both tests.  Rows l = E x1, (m0, m1) = (E^T x2)_{0,1} and h = H x1 come
// from bf16 MFMAs of three-piece splits; the float32 values
//     fE = |s| - sqrt(k) g,  g = sqrt(n1 n2 / (n1 + n2))        and        dH = dx^2 + dy^2 - (sqrt(T) hz)^2
// decide a pair by their sign unless they lie inside a rigorous band around zero, and each band is ONE constant per
// hypothesis, so the inner loops only keep the smallest |fE| / |dH| of a register row (half a v_min3 per pair) and
// look for the untrusted pairs of a row only when that minimum trips the band:
//   epipolar: the reference's test mean(d1^2, d2^2) < thr^2 is fE < 0 (k = 2 thr^2); g = L1 L2 / sqrt(L1^2 + L2^2) is
//   1-Lipschitz in the line norms L1, L2 and at most min(L1, L2), so the computed fE_c is within
//   Es + sqrt(k)(eL1 + eL2) + (float32 evaluation of g and of the final fma) of fE;
//   transfer: as in t_score_mfma_kernel with r = |(dx, dy)| and q = sqrt(T) hz, |q| taken at its bound over all matches.
// Pairs inside a band are re-evaluated with the float64 tests of eh_score_kernel (models parked in LDS).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void corr_to_mfma2_kernel(const double4* __restrict__ corr, int n, int npad, uint4* __restrict__ pieces,
                                                               float2* __restrict__ xy) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  uint4* out = pieces + 4 * (long long)j;
  double4 c = make_double4(0.0, 0.0, 0.0, 0.0);
  if (j < n) {
    c = corr[j];
    pack_pieces(c.x, c.y, 1.0, false, out[0], out[1]);
    pack_pieces(c.z, c.w, 1.0, false, out[2], out[3]);
    xy[j] = make_float2((float)c.z, (float)c.w);
  } else if (j < npad) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = make_uint4(0, 0, 0, 0);
    xy[j] = make_float2(0.0f, 0.0f);
  }
  publish_absmax(c, j < n, reinterpret_cast<unsigned*>(xy + npad));  // 16-byte trailer, zeroed by the launcher
}

// kDoE / kDoH: which of the two counts this instantiation produces (the other model is neither solved nor scored, its
// count array is not touched); the split forms need fewer registers and run at a higher occupancy (kMinBlocks).
template <bool kDoE, bool kDoH, int kMinBlocks>
__global__ __launch_bounds__(kBlock, kMinBlocks) void eh_score_mfma_kernel(const double4* __restrict__ corr, const uint4* __restrict__ pieces,
                                                                           const float2* __restrict__ xy, int n,
                                                                           const int32_t* __restrict__ idx, long long nb, double two_thr_e2,
                                                                           double thr_h2, double xmax, int* __restrict__ count_e,
                                                                           int* __restrict__ count_h, int tiles_per_split, int use_atomics) {
  constexpr int kVecE = kDoE ? 5 : 0, kVec = kVecE + (kDoH ? 3 : 0);  // piece vectors per hypothesis: E rows 0-2, E columns 0-1, H rows 0-2
  constexpr int kModel = 9 * ((kDoE ? 1 : 0) + (kDoH ? 1 : 0));       // doubles per hypothesis parked for the float64 re-evaluation
  static_assert(kModel * 8 <= kVec * 32, "the parked models reuse the staging area of the piece vectors");
  __shared__ unsigned char fin[kBlock / 64][64];
  __shared__ uint4 bstage[kBlock / 64][64][kVec][2];
  SFM_STAMP_ENTRY();
  __builtin_amdgcn_s_setprio(3);  // the solve runs ahead of the other waves' tile loops (see eh_upper_mfma_kernel)
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long hyp = gid < nb ? gid : nb - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const double safety = 2.0;
  const double xm = usable_xmax(xmax, *reinterpret_cast<const unsigned*>(xy + ((n + 31) / 32) * 32));
  const bool all_f64 = !(xm < 1.0e5);  // no usable coordinate bound (or float32 products that could overflow): every pair takes the float64 test
  float ge_own = 0.0f, gh_own = 0.0f;
  double Ed[9], Hd[9];
  {
    double A[9];
    float unused[9];
    {
      double p8[8][4], p4[4][4];
      gather_samples<double>(corr, n, idx + 12 * hyp, p8, p4);
      if constexpr (kDoE) solve_E8<double>(p8, Ed);
      if constexpr (kDoH) solve_H4<double>(p4, Hd);
    }
    {
      double chk_e = 0.0, chk_h = 0.0;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        if constexpr (kDoE) chk_e = fma(Ed[i], 0.0, chk_e);
        if constexpr (kDoH) chk_h = fma(Hd[i], 0.0, chk_h);
      }
      fin[wave][lane] = (unsigned char)((chk_e == 0.0 ? 1 : 0) | (chk_h == 0.0 ? 2 : 0));
    }
    if constexpr (kDoE) {
      const double inv = to_unit_f32(Ed, unused, A);
      const double R0 = xm * (A[0] + A[1]) + A[2], R1 = xm * (A[3] + A[4]) + A[5], R2 = xm * (A[6] + A[7]) + A[8];
      const double C0 = xm * (A[0] + A[3]) + A[6], C1 = xm * (A[1] + A[4]) + A[7];
      const double Es = (kEpsMfma + 4.0 * kEps32) * (xm * (R0 + R1) + R2);
      const double eL = (kEpsMfma + 2.0 * kEps32) * (R0 + R1 + C0 + C1);
      const double rk = sqrt(two_thr_e2);
      const double G = safety * (Es + rk * eL);
      // float32 evaluation of g = P rsq(P S + tiny): relative error below 8 eps where P S is far above the floor, at most
      // g <= 1.5e-5 in absolute terms where it is not (then min(n1, n2) < 2e-10); g <= min(L1, L2) <= Lb; the final fma
      // and the float32 value of sqrt(k) add 2 eps (|s| + sqrt(k) g)
      const double Lb = fmin(R0 + R1, C0 + C1), Sb = xm * (R0 + R1) + R2;
      ge_own = all_f64 ? __builtin_inff() : round_up_f32((G + rk * (8.0 * kEps32 * Lb + 1.5e-5) + 2.0 * kEps32 * (Sb + rk * Lb)) * 1.000001);
#pragma unroll
      for (int v = 0; v < 3; ++v)
        pack_pieces(Ed[3 * v] * inv, Ed[3 * v + 1] * inv, Ed[3 * v + 2] * inv, true, bstage[wave][lane][v][0], bstage[wave][lane][v][1]);
#pragma unroll
      for (int v = 0; v < 2; ++v)
        pack_pieces(Ed[v] * inv, Ed[3 + v] * inv, Ed[6 + v] * inv, true, bstage[wave][lane][3 + v][0], bstage[wave][lane][3 + v][1]);
    }
    if constexpr (kDoH) {
      const double inv = to_unit_f32(Hd, unused, A);
      const double R0 = xm * (A[0] + A[1]) + A[2], R1 = xm * (A[3] + A[4]) + A[5], R2 = xm * (A[6] + A[7]) + A[8];
      const double rt = sqrt(thr_h2);
      const double Dx = (kEpsMfma + 3.0 * kEps32) * (R0 + xm * R2), Dy = (kEpsMfma + 3.0 * kEps32) * (R1 + xm * R2);
      const double eq = rt * (kEpsMfma + 2.0 * kEps32) * R2;
      const double e = safety * (Dx + Dy + eq);
      // |dH_c| of a pair that the float32 sign may get wrong is below (4 e + 13 eps Q) |q_c| + 4.1 e^2 with |q_c| <= Q + e, Q = sqrt(T) R2
      const double Q = rt * R2, c1 = 4.0 * e + 13.0 * kEps32 * Q * 1.001;
      gh_own = all_f64 ? __builtin_inff() : round_up_f32((c1 * (Q * 1.000001 + e) + 4.1 * e * e) * 1.000001);
#pragma unroll
      for (int v = 0; v < 3; ++v)
        pack_pieces(Hd[3 * v] * inv, Hd[3 * v + 1] * inv, Hd[3 * v + 2] * inv, true, bstage[wave][lane][kVecE + v][0],
                    bstage[wave][lane][kVecE + v][1]);
    }
  }
  __syncthreads();
  bf16x8 B[2][kVec];
  float ge[2], gh[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int v = 0; v < kVec; ++v) B[k][v] = as_bf16x8(bstage[wave][32 * k + col][v][half]);
    ge[k] = __shfl(ge_own, 32 * k + col);
    gh[k] = __shfl(gh_own, 32 * k + col);
  }
  // the piece vectors now live in registers: the wave's part of the staging area is reused for the float64 models (in the
  // solver's own scale) that the re-evaluation of a pair inside a band needs
  __syncthreads();
  double* mstage = reinterpret_cast<double*>(&bstage[wave][0][0][0]);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    if constexpr (kDoE) mstage[kModel * lane + i] = Ed[i];
    if constexpr (kDoH) mstage[kModel * lane + (kDoE ? 9 : 0) + i] = Hd[i];
  }
  // E alone at three waves per SIMD is a dozen registers short: three of the piece vectors of the second hypothesis group
  // stay in LDS behind the models (three 16-byte reads per tile) instead of being spilled to scratch memory by the compiler
  constexpr bool kParkColumns = kDoE && !kDoH;
  static_assert(!kParkColumns || kModel * 8 * 64 + 3 * 64 * 16 <= kVec * 32 * 64, "the parked vectors fit behind the models");
  uint4* parked = reinterpret_cast<uint4*>(mstage + kModel * 64);
  if constexpr (kParkColumns) {
#pragma unroll
    for (int v = 0; v < 3; ++v) parked[64 * v + lane] = __builtin_bit_cast(uint4, B[1][2 + v]);
  }
  __syncthreads();
  const float rkf = (float)sqrt(two_thr_e2), rtf = (float)sqrt(thr_h2);
  const int apos = 16 * ((col >> 2) & 1) + (col & 3) + 4 * (col >> 3);  // see eh_upper_mfma_kernel
  const int ntiles = (n + 31) / 32;
  const int t0 = blockIdx.y * tiles_per_split;
  const int t1 = min(ntiles, t0 + tiles_per_split);
  int ce[2] = {0, 0}, ch[2] = {0, 0};
  const f32x16 zero = {0};
  const unsigned tail_mask = tile_tail_mask(n, half);
  const uint4* src = pieces + 4 * (long long)(t0 * 32 + apos) + half;
  const float4* xsrc = reinterpret_cast<const float4*>(xy + t0 * 32 + 16 * half);
  __builtin_amdgcn_s_setprio(0);
  SFM_STAMP_BEGIN();
  for (int t = t0; t < t1; ++t, src += 128, xsrc += 16) {
    const bf16x8 A1 = as_bf16x8(src[0]);
    const unsigned vmask = t == (n >> 5) ? tail_mask : 0xFFFFu;
    float x2[16], y2[16];
    {
      const float4* q = xsrc;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 v = q[i];
        x2[2 * i] = v.x; y2[2 * i] = v.y; x2[2 * i + 1] = v.z; y2[2 * i + 1] = v.w;
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      unsigned amb_e = 0, amb_h = 0;
      if constexpr (kDoE) {
        const bf16x8 A2 = as_bf16x8(src[2]);
        bf16x8 b2 = B[k][2], b3 = B[k][3], b4 = B[k][4];
        if constexpr (kParkColumns) {
          if (k == 1) {
            int at = lane;
            asm volatile("" : "+v"(at));  // keeps the reads inside the loop (hoisted, they would be spilled again)
            b2 = as_bf16x8(parked[at]);
            b3 = as_bf16x8(parked[64 + at]);
            b4 = as_bf16x8(parked[128 + at]);
          }
        }
        const f32x16 l0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B[k][0], zero, 0, 0, 0);
        const f32x16 l1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B[k][1], zero, 0, 0, 0);
        const f32x16 l2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, b2, zero, 0, 0, 0);
        const f32x16 m0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, b3, zero, 0, 0, 0);
        const f32x16 m1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, b4, zero, 0, 0, 0);
        auto fE = [&](int i) {  // < 0 <=> inlier
          const float s = fma(l0[i], x2[i], fma(l1[i], y2[i], l2[i]));
          const float n1 = fma(l0[i], l0[i], l1[i] * l1[i]);
          const float n2 = fma(m0[i], m0[i], m1[i] * m1[i]);
          const float S = n1 + n2, P = n1 * n2;
          const float g = P * __builtin_amdgcn_rsqf(fma(P, S, 1.0e-36f));  // raw v_rsq_f32: 1 ulp; the floor keeps a degenerate line finite
          return fma(-rkf, g, fabsf(s));
        };
        unsigned hneg = 0;
        float closest = __builtin_inff();
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const float f0 = fE(i), f1 = fE(i + 1);
          hneg = __builtin_amdgcn_alignbit(hneg, __float_as_uint(f0), 31);
          hneg = __builtin_amdgcn_alignbit(hneg, __float_as_uint(f1), 31);
          closest = min3_abs(closest, f0, f1);
        }
        ce[k] += __popc(hneg & vmask);
        if (closest <= ge[k]) {  // rare: a pair of this row is too close to the threshold for its float32 sign (or is padding)
          unsigned hamb = 0;
#pragma unroll
          for (int i = 0; i < 16; ++i) hamb = __builtin_amdgcn_alignbit(hamb, __float_as_uint(fabsf(fE(i)) - ge[k]), 31);
          amb_e = all_f64 ? vmask : (hamb & vmask);
          ce[k] -= __popc(hneg & amb_e);  // take the untrusted signs back out
        }
      }
      if constexpr (kDoH) {
        const f32x16 hx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B[k][kVecE], zero, 0, 0, 0);
        const f32x16 hy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B[k][kVecE + 1], zero, 0, 0, 0);
        const f32x16 hz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B[k][kVecE + 2], zero, 0, 0, 0);
        auto dH = [&](int i) {  // < 0 <=> inlier
          const float dx = fma(x2[i], hz[i], -hx[i]);
          const float dy = fma(y2[i], hz[i], -hy[i]);
          const float q = rtf * hz[i];
          return fma(-q, q, fma(dy, dy, dx * dx));
        };
        unsigned hneg = 0;
        float closest = __builtin_inff();
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const float d0 = dH(i), d1 = dH(i + 1);
          hneg = __builtin_amdgcn_alignbit(hneg, __float_as_uint(d0), 31);
          hneg = __builtin_amdgcn_alignbit(hneg, __float_as_uint(d1), 31);
          closest = min3_abs(closest, d0, d1);
        }
        ch[k] += __popc(hneg & vmask);
        if (closest <= gh[k]) {
          unsigned hamb = 0;
#pragma unroll
          for (int i = 0; i < 16; ++i) hamb = __builtin_amdgcn_alignbit(hamb, __float_as_uint(fabsf(dH(i)) - gh[k]), 31);
          amb_h = all_f64 ? vmask : (hamb & vmask);
          ch[k] -= __popc(hneg & amb_h);
        }
      }
      if (amb_e | amb_h) {
        const double* M = mstage + kModel * (32 * k + col);
        if (amb_e) {
          double E[9];
#pragma unroll
          for (int i = 0; i < 9; ++i) E[i] = M[i];
          while (amb_e) {
            const int b = __ffs((int)amb_e) - 1;  // bit 15-i of a history is match 16*half + i of the tile
            amb_e &= amb_e - 1;
            const double4 p = corr[t * 32 + 16 * half + 15 - b];
            ce[k] += epipolar_inlier<double>(E, p.x, p.y, p.z, p.w, two_thr_e2) ? 1 : 0;
          }
        }
        if (amb_h) {
          double H[9];
#pragma unroll
          for (int i = 0; i < 9; ++i) H[i] = M[(kDoE ? 9 : 0) + i];
          while (amb_h) {
            const int b = __ffs((int)amb_h) - 1;
            amb_h &= amb_h - 1;
            const double4 p = corr[t * 32 + 16 * half + 15 - b];
            ch[k] += transfer_inlier<double>(H, p.x, p.y, p.z, p.w, thr_h2) ? 1 : 0;
          }
        }
      }
    }
  }
  SFM_STAMP_END(t1 - t0);
  int se[2], sh[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    se[k] = ce[k] + __shfl_xor(ce[k], 32);
    sh[k] = ch[k] + __shfl_xor(ch[k], 32);
  }
  const unsigned ok = fin[wave][lane];
  const int oute = (ok & 1u) ? (half ? se[1] : se[0]) : 0, outh = (ok & 2u) ? (half ? sh[1] : sh[0]) : 0;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // the hypothesis id is formed again here: held across the tile loop it costs two registers (a spill at three waves)
  const long long out_id = (long long)blockIdx.x * blockDim.x + tid;
  if (out_id < nb) {
    if (use_atomics) {
      if constexpr (kDoE) atomicAdd(count_e + out_id, oute);
      if constexpr (kDoH) atomicAdd(count_h + out_id, outh);
    } else {
      if constexpr (kDoE) count_e[out_id] = oute;
      if constexpr (kDoH) count_h[out_id] = outh;
    }
  }
}

// ---------------------------------------------------------------------------------------
// The same exact counts with the WHOLE epipolar test on the matrix cores' outputs.  The float64 test of eh_score_kernel is
//     s^2 (n1 + n2) < k n1 n2,   s = x2^T E x1,  n1 = |(E x1)_{0,1}|^2,  n2 = |(E^T x2)_{0,1}|^2,  k = 2 thr^2,
// and s is a bilinear form in (x1, x2), n1 and n2 are quadratic forms in x1 and in x2: all three are contractions of
// match-side monomials with hypothesis-side coefficients (both formed in float64 and split into bf16 pieces), so they come
// out of bf16 MFMAs, five per 32 x 32 pairs (every MFMA costs the SIMD about as much issue time as ten vector instructions):
//     s   = [x2 x1, x2 y1, y2 x1] . [E00, E01, E10] + [y2 y1, x2, y2] . [E11, E02, E12] + [x1, y1, 1] . [E20, E21, E22]
//           three chained MFMAs, five of the nine piece products per monomial (pack_quad: relative error 1.1 * 2^-18)
//     n1' = k n1 = [x1^2, y1^2, x1, y1, x1 y1, 1] . k [Qaa, Qbb, Qa, Qb, Qab, Q1]       (Q = coefficients of |(E x1)_{0,1}|^2)
//     n2  =        [x2^2, y2^2, x2, y2, x2 y2, 1] . [...]
//           one MFMA each (pack_form: three piece products per monomial, two for x y and for the constant); the norms enter
//           the test scaled by k, so 2^-16 of their value and 2^-9 of the x y term are ample.
// The vector ALU is left with
//     S = n1'/k + n2,  P = n1' n2,  D = s^2 S - P      (4 instructions; D < 0 <=> inlier)
//     r = |D| - S (A |s| + B)                          (2 instructions; r < 0 <=> the sign of D is not trusted)
// and one v_alignbit for each of the two signs: 10 issue slots per pair.
// Error bound (everything relative to the float64 value of D on the unrounded inputs; u = 2^-24):
//   |s_c - s| <= es: piece products + 32 u per MFMA of the chain on (its own terms + what it is handed), Sb >= sum |E||x2||x1|;
//   |n_c - n| <= en, eS = en1 + en2; n2's constant carries an offset 2 eS, so that S <= S_c - eS and S_c >= eS;
//   |s_c^2 S_c - s^2 S| <= 3 eS s_c^2 + (2 |s_c| es + es^2) S_c + rounding,   |P_c - k n1 n2| <= 3 k eS S_c (1 + ...) + rounding,
//   float32 evaluation of D: 3.1 u |s_c| (Sb + es) S_c + u k (Lb2 + 4 eS) S_c,  Lb2 >= min(n1, n2),
// collected as |D_c - D| <= S_c (A |s_c| + B) + C s_c^2 with A = 2 es + 3.1 u (Sb + es), B = es^2 + 6 k eS + u k (Lb2 + 4 eS),
// C = 3 eS; the C term is bounded by a multiple of S_c for every pair whose sign can be wrong and rides in B (see below);
// A and B are doubled (safety) and rounded up.
// ---------------------------------------------------------------------------------------
constexpr int kEhxVecA = 8;  // match-side vectors per match: S1, S2, S3 (the nine monomials of s, five piece products each), N1, N2 (the forms n1, n2), A1 = (x1, y1, 1) for the H test, T1, T2 (the nine monomials of s with three products each: the early test)

// 5-term piece products of three monomials: hypothesis side (q0 q0 q1 q0 q2), match side (m0 m1 m0 m2 m0), 15 of 16 slots
__device__ __forceinline__ void pack_quad(double a, double b, double c, bool model, uint4& lo, uint4& hi) {
  unsigned p[3][3];
  split3_bf16(a, p[0]);
  split3_bf16(b, p[1]);
  split3_bf16(c, p[2]);
  unsigned e[16];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    if (model) {
      e[5 * m] = p[m][0]; e[5 * m + 1] = p[m][0]; e[5 * m + 2] = p[m][1]; e[5 * m + 3] = p[m][0]; e[5 * m + 4] = p[m][2];
    } else {
      e[5 * m] = p[m][0]; e[5 * m + 1] = p[m][1]; e[5 * m + 2] = p[m][0]; e[5 * m + 3] = p[m][2]; e[5 * m + 4] = p[m][0];
    }
  }
  e[15] = 0;
  lo = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
  hi = make_uint4(e[8] | (e[9] << 16), e[10] | (e[11] << 16), e[12] | (e[13] << 16), e[14] | (e[15] << 16));
}

// A whole quadratic form a x^2 + b x y + c y^2 + d x + e y + f in ONE operand vector: three piece products (m0 c0, m1 c0, m0 c1)
// for x^2, y^2, x and y, two (m0 c0, m0 c1) for x y and two for the constant (match side 1, 1).  Relative error of a kept
// triple 3.1 * 2^-18, of the x y term 2^-9 * 1.01 (its second match-side piece is dropped), of the constant 2^-17.
__device__ __forceinline__ void pack_form(double xx, double xy, double yy, double x, double y, double one, bool model, uint4& lo, uint4& hi) {
  unsigned p[3], e[16];
  const double three[4] = {xx, yy, x, y};
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    split3_bf16(three[m], p);
    if (model) {
      e[3 * m] = p[0]; e[3 * m + 1] = p[0]; e[3 * m + 2] = p[1];
    } else {
      e[3 * m] = p[0]; e[3 * m + 1] = p[1]; e[3 * m + 2] = p[0];
    }
  }
  split3_bf16(xy, p);
  e[12] = p[0]; e[13] = model ? p[1] : p[0];
  split3_bf16(one, p);
  e[14] = p[0]; e[15] = model ? p[1] : p[0];
  lo = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
  hi = make_uint4(e[8] | (e[9] << 16), e[10] | (e[11] << 16), e[12] | (e[13] << 16), e[14] | (e[15] << 16));
}

// Three piece products (m0 c0, m1 c0, m0 c1) of up to five monomials in one operand vector (relative error of a kept triple
// 3.1 * 2^-18); unused slots are zero.
template <int kCount>
__device__ __forceinline__ void pack_tri(const double (&v)[kCount], bool model, uint4& lo, uint4& hi) {
  static_assert(kCount <= 5, "16 slots");
  unsigned p[3], e[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) e[i] = 0;
#pragma unroll
  for (int m = 0; m < kCount; ++m) {
    split3_bf16(v[m], p);
    if (model) {
      e[3 * m] = p[0]; e[3 * m + 1] = p[0]; e[3 * m + 2] = p[1];
    } else {
      e[3 * m] = p[0]; e[3 * m + 1] = p[1]; e[3 * m + 2] = p[0];
    }
  }
  lo = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
  hi = make_uint4(e[8] | (e[9] << 16), e[10] | (e[11] << 16), e[12] | (e[13] << 16), e[14] | (e[15] << 16));
}

// pieces[tile][vector][match of the tile][half]: a wave's load of one vector of one tile is 1 KB of consecutive bytes
__global__ __launch_bounds__(kBlock) void corr_to_mfma3_kernel(const double4* __restrict__ corr, int n, int npad, uint4* __restrict__ pieces,
                                                               float2* __restrict__ xy) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  double4 c = make_double4(0.0, 0.0, 0.0, 0.0);
  if (j < npad) {
    uint4* out = pieces + ((long long)(j >> 5) * (kEhxVecA * 32) + (j & 31)) * 2;  // vector v at out[64 v], out[64 v + 1]
    if (j < n) {
      c = corr[j];
      pack_quad(c.z * c.x, c.z * c.y, c.w * c.x, false, out[0], out[1]);
      pack_quad(c.w * c.y, c.z, c.w, false, out[64], out[65]);
      pack_form(c.x * c.x, c.x * c.y, c.y * c.y, c.x, c.y, 1.0, false, out[128], out[129]);
      pack_form(c.z * c.z, c.z * c.w, c.w * c.w, c.z, c.w, 1.0, false, out[192], out[193]);
      pack_pieces(c.x, c.y, 1.0, false, out[256], out[257]);
      pack_quad(c.x, c.y, 1.0, false, out[320], out[321]);
      const double t1[5] = {c.z * c.x, c.z * c.y, c.w * c.x, c.w * c.y, c.z}, t2[4] = {c.w, c.x, c.y, 1.0};
      pack_tri<5>(t1, false, out[384], out[385]);
      pack_tri<4>(t2, false, out[448], out[449]);
      xy[j] = make_float2((float)c.z, (float)c.w);
    } else {
#pragma unroll
      for (int v = 0; v < kEhxVecA; ++v) out[64 * v] = out[64 * v + 1] = make_uint4(0, 0, 0, 0);
      xy[j] = make_float2(0.0f, 0.0f);
    }
  }
  publish_absmax(c, j < n, reinterpret_cast<unsigned*>(xy + npad));  // 16-byte trailer, zeroed by the launcher
}

// ---------------------------------------------------------------------------------------
// Matrix-core pre-filter, SECOND FORM (the one in use; eh_upper_mfma_kernel stays reachable with SFM_HIP_UPPER_FORM=0):
// the one-sided epipolar test whole from the matrix cores.  s = x2^T E x1 is a bilinear form and k1 n1 = k1 |(E x1)_{0,1}|^2
// a quadratic form in x1, i.e. contractions of match-side monomials with hypothesis-side coefficients, exactly as in
// eh_score_mfma2_kernel: two chained MFMAs give s (vectors T1, T2: three of the nine piece products per monomial) and
// one gives N = k1 n1 + c (vector N1: the whole quadratic form, the slack constant c riding in its constant term), so the
// test s^2 < N costs ONE vector instruction per pair (plus the count) instead of five after the rows E x1.
// Upper-bound property.  A float64 inlier has s^2 (n1 + n2) < k n1 n2, hence s^2 < k n1.  With |s_c - s| <= es
// (bound of eh_score_mfma2_kernel, same vectors, same chain order) and AM-GM, s_c^2 <= (1 + mu) s^2 + (1 + 1/mu) es^2
// < k1 n1 + (1 + 1/mu) es^2 with k1 >= k (1 + mu), for any mu > 0 (chosen per hypothesis); the computed N_c errs by at most k1 en1 + kn c (piece products of the
// form, the x y monomial at 2^-9, accumulation: en1 and kn as in eh_score_mfma2_kernel), so with
//     c = ((1 + 1/mu) (f es)^2 + f k1 en1) / (1 - f kn)        (f = the caller's safety factor)
// every float64 inlier has s_c^2 < N_c, and the fma N_c - s_c^2 has the sign of the exact difference.
// kFloatCount (as in eh_upper_mfma_kernel): the outcome is clamp(N_c - s_c^2), exactly 0 or 1 when the hypothesis is
// scaled by a power of two lambda with lambda^2 (1 + 1/mu) (f es)^2 >= 2^50: N_c' >= 2^50 is then a multiple of 2^27, an
// s_c' whose square can cancel against it is at least 2^24.9 and so a multiple of 2, and the exact difference is an
// integer: zero, negative, or at least 1.  lambda^2 beyond 2^116 could overflow a coefficient: such a hypothesis (all
// coordinates below about 2^-12) reports the trivial bound.
// The transfer half is the box test of eh_upper_mfma_kernel, unchanged.
// Match-side operands: those of eh_upper_mfma_kernel (piece vector of (x1, y1, 1), (x2, y2) / sqrt(k1), trailer) followed by
// evec[tile][T1, T2, N1][match of the tile][half] (T1, T2: the nine monomials of s with three piece products each): 136 bytes per match.
// ---------------------------------------------------------------------------------------
constexpr int kUp2Vec = 3;

__global__ __launch_bounds__(kBlock) void corr_to_upper2_kernel(const double4* __restrict__ corr, int n, int npad, double inv_sqrt_k1,
                                                                uint4* __restrict__ pieces, float2* __restrict__ xy, uint4* __restrict__ evec) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) *reinterpret_cast<double*>(xy + npad) = inv_sqrt_k1;  // tag: which threshold these operands were prepared for
  double4 c = make_double4(0.0, 0.0, 0.0, 0.0);
  if (j < npad) {
    uint4* out = evec + ((long long)(j >> 5) * (kUp2Vec * 32) + (j & 31)) * 2;  // vector v at out[64 v], out[64 v + 1]
    if (j < n) {
      c = corr[j];
      pack_pieces(c.x, c.y, 1.0, false, pieces[2 * j], pieces[2 * j + 1]);
      xy[j] = make_float2((float)(c.z * inv_sqrt_k1), (float)(c.w * inv_sqrt_k1));
      const double t1[5] = {c.z * c.x, c.z * c.y, c.w * c.x, c.w * c.y, c.z}, t2[4] = {c.w, c.x, c.y, 1.0};
      pack_tri<5>(t1, false, out[0], out[1]);
      pack_tri<4>(t2, false, out[64], out[65]);
      pack_form(c.x * c.x, c.x * c.y, c.y * c.y, c.x, c.y, 1.0, false, out[128], out[129]);
    } else {
      pieces[2 * j] = pieces[2 * j + 1] = make_uint4(0, 0, 0, 0);
      xy[j] = make_float2(0.0f, 0.0f);
#pragma unroll
      for (int v = 0; v < kUp2Vec; ++v) out[64 * v] = out[64 * v + 1] = make_uint4(0, 0, 0, 0);
    }
  }
  publish_absmax(c, j < n, reinterpret_cast<unsigned*>(xy + npad) + 2);  // behind the tag; zeroed by the launcher
}

template <bool kDoE, bool kDoH, bool kFloatCount>
__global__ __launch_bounds__(kBlock, 3) void eh_upper_mfma2_kernel(const double4* __restrict__ corr, const uint4* __restrict__ pieces,
                                                                   const float2* __restrict__ xy, const uint4* __restrict__ evec, int n,
                                                                   const int32_t* __restrict__ idx, long long nb, double two_thr_e2,
                                                                   double thr_h2, double xmax, double safety, int* __restrict__ up_e,
                                                                   int* __restrict__ up_h, int tiles_per_split, int use_atomics) {
  __shared__ unsigned char fin[kBlock / 64][64];
  __shared__ uint4 bstage[kBlock / 64][64][4][2];  // four piece vectors per round
  SFM_STAMP_ENTRY();
  __builtin_amdgcn_s_setprio(3);  // the solve runs ahead of the other waves' tile loops (see eh_upper_mfma_kernel)
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long hyp = gid < nb ? gid : nb - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const double inv_sqrt_k1 = 1.0 / sqrt(uniform_k1(two_thr_e2));  // the scale of (x2, y2) in the operands of the box test
  const int npad_all = ((n + 31) / 32) * 32;
  const double xm = usable_xmax(xmax, reinterpret_cast<const unsigned*>(xy + npad_all)[2]);
  constexpr bool kPark = kDoE && kDoH;
  bf16x8 BE[2][kDoE ? 3 : 1], BH[2][kDoH ? 3 : 1];
  {
    double Ed[9], Hd[9], A[9];
    float unused[9];
    {
      double p8[8][4], p4[4][4];
      gather_samples<double>(corr, n, idx + 12 * hyp, p8, p4);
      if constexpr (kDoE) solve_E8<double>(p8, Ed);
      if constexpr (kDoH) solve_H4<double>(p4, Hd);
    }
    // bit 0 / 1: the E / H model is finite (a model with a non-finite entry has the float64 count 0); bit 2: the E bound of
    // this hypothesis is the trivial one
    unsigned flags = 0;
    {
      double chk_e = 0.0, chk_h = 0.0;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        if constexpr (kDoE) chk_e = fma(Ed[i], 0.0, chk_e);  // NaN as soon as one entry is NaN or infinite
        if constexpr (kDoH) chk_h = fma(Hd[i], 0.0, chk_h);
      }
      flags = (chk_e == 0.0 ? 1u : 0u) | (chk_h == 0.0 ? 2u : 0u);
    }
    if constexpr (kDoE) {
      double e[9];
      const double inv = to_unit_f32(Ed, unused, A);
#pragma unroll
      for (int i = 0; i < 9; ++i) e[i] = Ed[i] * inv;
      const double R0 = xm * (A[0] + A[1]) + A[2], R1 = xm * (A[3] + A[4]) + A[5], R2 = xm * (A[6] + A[7]) + A[8];
      const double Sb = xm * (R0 + R1) + R2, N1b = R0 * R0 + R1 * R1;
      const double u = kEps32;
      // |s_c - s|: three of the nine piece products of every monomial are kept (3.1 * 2^-18 Sb); the two MFMAs of the chain hold
      // the terms of (x2 x1, x2 y1, y2 x1, y2 y1, x2) and of (y2, x1, y1, 1) in this order, and each adds at most 32 u (its own
      // terms + what it is handed), so the first group's magnitude is counted twice
      const double es = (3.1 * 0x1.0p-18 / u * Sb + 32.0 * (1.0 + 0x1.0p-5) * (2.0 * (xm * xm * (A[0] + A[1] + A[3] + A[4]) + xm * A[2]) +
                                                                              (xm * (A[5] + A[6] + A[7]) + A[8]))) * u;
      const double q1ab = 2.0 * fabs(fma(e[0], e[1], e[3] * e[4]));
      const double kn = 3.1 * 0x1.0p-18 + 34.0 * u;
      const double en1 = (kn * N1b + 0x1.02p-9 * xm * xm * q1ab) * 1.0001;
      // Any mu > 0 is valid; the widening of the squared threshold, mu + (1 + 1/mu) (f es)^2 / (k n1), is smallest at
      // mu = f es / sqrt(k n1): taken at the mean of n1 over coordinates spread evenly in [-xm, xm]
      const double n1_ref = A[2] * A[2] + A[5] * A[5] + xm * xm / 3.0 * (A[0] * A[0] + A[1] * A[1] + A[3] * A[3] + A[4] * A[4]);
      const double mu = fmin(0.5, fmax(0x1.0p-7, safety * es / sqrt(two_thr_e2 * n1_ref)));  // NaN (n1_ref = 0) -> 2^-7
      const double k1 = two_thr_e2 * (1.0 + mu) * (1.0 + 32.0 * kEps32);
      const double floor_c = (1.0 + 1.0 / mu) * (safety * es) * (safety * es);
      const double cst = (floor_c + safety * k1 * en1) / (1.0 - safety * kn) * 1.0001;
      double lam = 1.0;
      if constexpr (kFloatCount) {
        int ex = 0;
        (void)frexp(floor_c, &ex);  // floor_c >= 2^(ex - 1)
        int lg = (52 - ex) / 2;     // ceil((51 - ex) / 2) for every sign of 51 - ex that can occur with the guard below
        if (2 * lg < 51 - ex) ++lg;
        if (!(floor_c > 0.0) || lg > 58 || lg < -30) {
          flags |= 4u;
          lg = 0;
        }
        lam = ldexp(1.0, lg);
      }
      const double l2 = lam * lam * k1;
      const double c1[5] = {e[0] * lam, e[1] * lam, e[3] * lam, e[4] * lam, e[2] * lam}, c2[4] = {e[5] * lam, e[6] * lam, e[7] * lam, e[8] * lam};
      pack_tri<5>(c1, true, bstage[wave][lane][0][0], bstage[wave][lane][0][1]);
      pack_tri<4>(c2, true, bstage[wave][lane][1][0], bstage[wave][lane][1][1]);
      pack_form(l2 * fma(e[0], e[0], e[3] * e[3]), 2.0 * l2 * fma(e[0], e[1], e[3] * e[4]), l2 * fma(e[1], e[1], e[4] * e[4]),
                2.0 * l2 * fma(e[0], e[2], e[3] * e[5]), 2.0 * l2 * fma(e[1], e[2], e[4] * e[5]),
                l2 * fma(e[2], e[2], e[5] * e[5]) + lam * lam * cst, true, bstage[wave][lane][2][0], bstage[wave][lane][2][1]);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int v = 0; v < 3; ++v) BE[k][v] = as_bf16x8(bstage[wave][32 * k + col][v][half]);
      __syncthreads();
    }
    if constexpr (kDoH) {
      double inv = to_unit_f32(Hd, unused, A);
      const double R0 = xm * (A[0] + A[1]) + A[2], R1 = xm * (A[3] + A[4]) + A[5], R2 = xm * (A[6] + A[7]) + A[8];
      const double ez = kEpsMfma * R2;
      const double Dx = (kEpsMfma + 3.0 * kEps32) * (R0 + xm * R2), Dy = (kEpsMfma + 3.0 * kEps32) * (R1 + xm * R2);
      const double G = safety * (sqrt(thr_h2) * ez + Dx + Dy);
      inv *= kGh / G;
      if constexpr (kFloatCount) inv *= kLambdaH;
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const double sc = v == 2 ? inv : inv * inv_sqrt_k1;
        pack_pieces(Hd[3 * v] * sc, Hd[3 * v + 1] * sc, Hd[3 * v + 2] * sc, true, bstage[wave][lane][v][0], bstage[wave][lane][v][1]);
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int v = 0; v < 3; ++v) BH[k][v] = as_bf16x8(bstage[wave][32 * k + col][v][half]);
      if constexpr (kPark) {
        // the fused instantiation is a dozen registers short at three waves per SIMD: four of the six transfer-side piece
        // vectors wait in LDS (one 16-byte read each per tile, consecutive lanes at consecutive addresses) instead of
        // being spilled to scratch memory by the compiler
        __syncthreads();
        uint4* parked = &bstage[wave][0][0][0];
#pragma unroll
        for (int v = 0; v < 3; ++v) parked[64 * v + lane] = __builtin_bit_cast(uint4, BH[1][v]);
        parked[192 + lane] = __builtin_bit_cast(uint4, BH[0][2]);
        __syncthreads();
      }
    }
    fin[wave][lane] = (unsigned char)flags;  // parked in LDS: no register lives across the loop
  }
  const uint4* parked = &bstage[wave][0][0][0];
  const float rtp = round_up_f32(sqrt(thr_h2) * inv_sqrt_k1 * (1.0 + 16.0 * kEps32));
  const float gp = round_up_f32(kGh * inv_sqrt_k1 * (1.0 + 16.0 * kEps32)) * (kFloatCount ? (float)kLambdaH : 1.0f);
  const int apos = 16 * ((col >> 2) & 1) + (col & 3) + 4 * (col >> 3);  // see eh_upper_mfma_kernel
  const int ntiles = (n + 31) / 32;
  const int t0 = blockIdx.y * tiles_per_split;
  const int t1 = min(ntiles, t0 + tiles_per_split);
  int ce[2] = {0, 0}, ch[2] = {0, 0};
  unsigned he[2] = {0, 0}, hh[2] = {0, 0};
  float fe[2] = {0.0f, 0.0f}, fh[2] = {0.0f, 0.0f};
  const f32x16 zero = {0};
  const unsigned tail_mask = tile_tail_mask(n, half);
  const uint4* asrc = pieces + 2 * (t0 * 32 + apos) + half;
  const float4* xsrc = reinterpret_cast<const float4*>(xy + t0 * 32 + 16 * half);
  const uint4* esrc = evec + ((long long)t0 * (kUp2Vec * 32) + apos) * 2 + half;
  // every match-side operand is loaded one tile ahead, into the registers its last reader has just released
  bf16x8 A, AT1, AT2, AN1;
  float x2[kDoH ? 16 : 1], y2[kDoH ? 16 : 1];
  auto load_e = [&](const uint4* q) {
    AT1 = as_bf16x8(q[0]);
    AT2 = as_bf16x8(q[64]);
    AN1 = as_bf16x8(q[128]);
  };
  auto load_xy = [&](const float4* q, int pair) {
    const float4 v = q[pair];
    x2[2 * pair] = v.x; y2[2 * pair] = v.y; x2[2 * pair + 1] = v.z; y2[2 * pair + 1] = v.w;
  };
  if constexpr (kDoE) load_e(esrc);
  if constexpr (kDoH) {
    A = as_bf16x8(*asrc);
#pragma unroll
    for (int i = 0; i < 8; ++i) load_xy(xsrc, i);
  }
  auto score_tile = [&](auto masked_tag, const uint4* anext, const float4* xnext, const uint4* enext) {
    constexpr bool kMasked = decltype(masked_tag)::value;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if constexpr (kDoE) {
        f32x16 sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT1, BE[k][0], zero, 0, 0, 0);
        const f32x16 nv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AN1, BE[k][2], zero, 0, 0, 0);
        sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT2, BE[k][1], sv, 0, 0, 0);
        if (k == 1) load_e(enext);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if constexpr (kFloatCount) {
            // (the clamp is left to the compiler here: it must see the first reader of the matrix-core results to place
            // the wait states that those need; behind an asm statement it does not)
            const float t = __builtin_amdgcn_fmed3f(fma(-sv[i], sv[i], nv[i]), 0.0f, 1.0f);  // 1 <=> inlier (or within the slack)
            if constexpr (kMasked)
              fe[k] = fma(t, ((tail_mask >> (15 - i)) & 1u) ? 1.0f : 0.0f, fe[k]);
            else
              fe[k] += t;
          } else {
            const float d = fma(sv[i], sv[i], -nv[i]);  // inlier (or within the slack) <=> d < 0
            he[k] = __builtin_amdgcn_alignbit(he[k], __float_as_uint(d), 31);
          }
        }
        if constexpr (!kFloatCount) ce[k] += __popc(he[k] & (kMasked ? tail_mask : 0xFFFFu));
      }
      if constexpr (kDoH) {
        bf16x8 h0 = BH[k][0], h1 = BH[k][1], h2 = BH[k][2];
        if constexpr (kPark) {
          int at = lane;
          asm volatile("" : "+v"(at));  // keeps the reads inside the loop (hoisted, they would be spilled again)
          if (k == 1) {
            h0 = as_bf16x8(parked[at]);
            h1 = as_bf16x8(parked[64 + at]);
            h2 = as_bf16x8(parked[128 + at]);
          } else {
            h2 = as_bf16x8(parked[192 + at]);
          }
        }
        const f32x16 hx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, h0, zero, 0, 0, 0);
        const f32x16 hy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, h1, zero, 0, 0, 0);
        const f32x16 hz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, h2, zero, 0, 0, 0);
        if (k == 1) A = as_bf16x8(*anext);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float dx = fma(x2[i], hz[i], -hx[i]);
          const float dy = fma(y2[i], hz[i], -hy[i]);
          if (k == 1 && (i & 1)) load_xy(xnext, i >> 1);
          const float lim = fma(rtp, fabsf(hz[i]), gp);
          if constexpr (kFloatCount) {
            const float tx = clamp01_b_minus_abs(lim, dx), ty = clamp01_b_minus_abs(lim, dy);  // both 1 <=> inside the box
            if constexpr (kMasked)
              fh[k] = fma(tx * ty, ((tail_mask >> (15 - i)) & 1u) ? 1.0f : 0.0f, fh[k]);
            else
              fh[k] = fma(tx, ty, fh[k]);
          } else {
            const float d = fmaxf(fabsf(dx), fabsf(dy)) - lim;
            hh[k] = __builtin_amdgcn_alignbit(hh[k], __float_as_uint(d), 31);
          }
        }
        if constexpr (!kFloatCount) ch[k] += __popc(hh[k] & (kMasked ? tail_mask : 0xFFFFu));
      }
    }
  };
  __builtin_amdgcn_s_setprio(0);
  SFM_STAMP_BEGIN();
  const int t_partial = (n & 31) ? (n >> 5) : ntiles;  // the tile with padding, if any
  const int t_full = min(t1, t_partial);
  for (int t = t0; t < t_full; ++t) {
    const bool more = t + 1 < t1;  // the last tile of the range loads itself again (in bounds, unused)
    const uint4* anext = more ? asrc + 64 : asrc;
    const float4* xnext = more ? xsrc + 16 : xsrc;
    const uint4* enext = more ? esrc + kUp2Vec * 64 : esrc;
    score_tile(std::false_type{}, anext, xnext, enext);
    asrc = anext;
    xsrc = xnext;
    esrc = enext;
  }
  if (t_partial >= t0 && t_partial < t1) score_tile(std::true_type{}, asrc, xsrc, esrc);
  if constexpr (kFloatCount) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      ce[k] = (int)fe[k];
      ch[k] = (int)fh[k];
    }
  }
  SFM_STAMP_END(t1 - t0);
  int se[2], sh[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    se[k] = ce[k] + __shfl_xor(ce[k], 32);
    sh[k] = ch[k] + __shfl_xor(ch[k], 32);
  }
  const unsigned ok = fin[wave][lane];
  const int trivial = max(0, min(n, t1 * 32) - t0 * 32);  // all matches of this split
  int oute = (ok & 1u) ? ((ok & 4u) ? trivial : (half ? se[1] : se[0])) : 0, outh = (ok & 2u) ? (half ? sh[1] : sh[0]) : 0;
  // operands prepared for another threshold, or matches without a usable coordinate bound, cannot give valid bounds
  if (*reinterpret_cast<const double*>(xy + npad_all) != inv_sqrt_k1 || !(xm < 1.0e6)) oute = outh = trivial;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // the hypothesis id is formed again here rather than held in two registers across the tile loop
  const long long out_id = (long long)blockIdx.x * blockDim.x + tid;
  if (out_id < nb) {
    if (use_atomics) {
      if constexpr (kDoE) atomicAdd(up_e + out_id, oute);
      if constexpr (kDoH) atomicAdd(up_h + out_id, outh);
    } else {
      if constexpr (kDoE) up_e[out_id] = oute;
      if constexpr (kDoH) up_h[out_id] = outh;
    }
  }
}

template <bool kDoE, bool kDoH, int kMinBlocks, bool kEarly>
__global__ __launch_bounds__(kBlock, kMinBlocks) void eh_score_mfma2_kernel(const double4* __restrict__ corr, const uint4* __restrict__ pieces,
                                                                            const float2* __restrict__ xy, int n,
                                                                            const int32_t* __restrict__ idx, long long nb, double two_thr_e2,
                                                                            double thr_h2, double xmax, int* __restrict__ count_e,
                                                                            int* __restrict__ count_h, int tiles_per_split, int use_atomics) {
  constexpr bool early = kEarly;  // an instantiation of its own: the form without early loops is the code it was before they existed
  constexpr int kModel = 9 * ((kDoE ? 1 : 0) + (kDoH ? 1 : 0));  // doubles per hypothesis parked for the float64 re-evaluation
  constexpr bool kPark = kDoE != kDoH;          // a one-model instantiation keeps some piece vectors in LDS, see below
  constexpr int kParkedE = 0;  // piece vectors of the second hypothesis group that the E-only instantiation keeps in LDS
  constexpr int kParked = kPark ? 4 : 0;  // H: four vectors re-read per tile; E: the four vectors of k n1 and n2 wait there while the early loop runs
  constexpr int kAfter = kModel * 8 * 64 + kParked * 16 * 64;  // the staging area later holds the float64 models and the parked vectors
  constexpr int kStageBytes = kAfter > 4 * 32 * 64 ? kAfter : 4 * 32 * 64;  // four piece vectors per round first
  constexpr int kQueue = 256;  // pairs whose float32 sign is not trusted wait here for a dense float64 pass (per wave)
  __shared__ unsigned char fin[kBlock / 64][64];
  __shared__ __align__(16) unsigned char stage_raw[kBlock / 64][kStageBytes];
  __shared__ unsigned qbuf[kBlock / 64][kQueue];
  __shared__ int qcnt[kBlock / 64][(kDoE && kDoH) ? 2 : 1][64];
  SFM_STAMP_ENTRY();
  __builtin_amdgcn_s_setprio(3);  // the solve runs ahead of the other waves' tile loops (see eh_upper_mfma_kernel)
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long hyp = gid < nb ? gid : nb - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const double safety = 2.0;
  const double xm = usable_xmax(xmax, *reinterpret_cast<const unsigned*>(xy + ((n + 31) / 32) * 32));
  const bool all_f64 = !(xm < 1.0e4);  // no usable coordinate bound (or float32 products that could overflow): every pair takes the float64 test
  uint4(*bstage)[4][2] = reinterpret_cast<uint4(*)[4][2]>(&stage_raw[wave][0]);  // [lane][vector of the round][half]
  double Ed[9], Hd[9];
  float ea_own = 0.0f, eb_own = 0.0f, gh_own = 0.0f;
  unsigned early_all_own = 0u;  // 0xFFFF: every pair of this hypothesis passes the early test (its outcome cannot be produced exactly)
  bf16x8 BE[2][kDoE ? 6 : 1], BH[2][kDoH ? 3 : 1];  // BE[.][5]: the form of the early test
  bf16x8 BT[2][(kDoE && kEarly) ? 2 : 1];           // the early test's s: three piece products per monomial in two vectors
  bf16x8 BHe[2][(kDoH && kEarly) ? 3 : 1];          // the early test of the H launch: dx in two vectors, the form of (rt hz)^2 + c
  {
    double p8[8][4], p4[4][4];
    gather_samples<double>(corr, n, idx + 12 * hyp, p8, p4);
    if constexpr (kDoE) solve_E8<double>(p8, Ed);
    if constexpr (kDoH) solve_H4<double>(p4, Hd);
  }
  {
    double chk_e = 0.0, chk_h = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      if constexpr (kDoE) chk_e = fma(Ed[i], 0.0, chk_e);
      if constexpr (kDoH) chk_h = fma(Hd[i], 0.0, chk_h);
    }
    fin[wave][lane] = (unsigned char)((chk_e == 0.0 ? 1 : 0) | (chk_h == 0.0 ? 2 : 0));
  }
  auto fetch_round = [&](auto& B, int first, int count) {  // piece vectors first .. first + count - 1 of both hypothesis groups
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int v = 0; v < count; ++v) B[k][first + v] = as_bf16x8(bstage[32 * k + col][v][half]);
    __syncthreads();
  };
  if constexpr (kDoE) {
    double A[9], e[9], n2_offset;
    float unused[9];
    const double inv = to_unit_f32(Ed, unused, A);
#pragma unroll
    for (int i = 0; i < 9; ++i) e[i] = Ed[i] * inv;
    const double k = two_thr_e2;
    const double R0 = xm * (A[0] + A[1]) + A[2], R1 = xm * (A[3] + A[4]) + A[5], R2 = xm * (A[6] + A[7]) + A[8];
    const double C0 = xm * (A[0] + A[3]) + A[6], C1 = xm * (A[1] + A[4]) + A[7];
    const double Sb = xm * (R0 + R1) + R2, N1b = R0 * R0 + R1 * R1, N2b = C0 * C0 + C1 * C1, Lb2 = fmin(N1b, N2b);
    const double u = kEps32;
    // |s_c - s|: five of the nine piece products of every monomial are kept (1.1 * 2^-18 Sb), and every MFMA of the chain adds at
    // most 32 u (its own terms + what it is handed): the three MFMAs hold the terms of (x2 x1, x2 y1, y2 x1), (y2 y1, x2, y2) and
    // (x1, y1, 1) in this order, so the first group's magnitude is counted three times and the last one's once (2^-5: the
    // lower-order piece products and the rounding of what is handed on)
    const double es = (1.1 * 0x1.0p-18 / u * Sb + 32.0 * (1.0 + 0x1.0p-5) * (3.0 * xm * xm * (A[0] + A[1] + A[3]) + 2.0 * (xm * xm * A[4] + xm * (A[2] + A[5])) +
                                                                            (xm * (A[6] + A[7]) + A[8]))) * u;
    // n1 and n2 are one MFMA each: kept piece products 3.1 * 2^-18, accumulation 34 u, and the x y monomial with two of
    // three products (2^-9 of its term, |x y| <= xm^2)
    const double q1ab = 2.0 * fabs(fma(e[0], e[1], e[3] * e[4])), q2ab = 2.0 * fabs(fma(e[0], e[3], e[1] * e[4]));
    const double kn = 3.1 * 0x1.0p-18 + 34.0 * u;
    const double eS = (kn * (N1b + N2b) + 0x1.02p-9 * xm * xm * (q1ab + q2ab)) * 1.0001;  // en1 + en2, the offset 2 eS of n2 (below) included in its bound
    const double cA = 2.0 * es + 3.1 * u * (Sb + es);
    const double cB = es * es + 6.0 * k * eS + u * k * (Lb2 + 4.0 * eS);
    const double cC = 3.0 * eS;
    const double infl = safety * (1.0 + 0x1.0p-18);
    // A pair whose float32 sign is wrong has s_c^2 <= k S_c / 2 + 2 es^2 (computed inlier: s_c^2 S_c < k n1_c n2_c <= k S_c^2 / 4;
    // true inlier: s_c^2 <= 2 s^2 + 2 es^2 and s^2 < k n1 n2 / (n1 + n2) <= k S / 4), and S_c >= eS: the s^2 term cC s_c^2 of
    // the band is therefore below S_c (cC k / 2 + 6 es^2) and rides in the constant term.
    ea_own = round_up_f32(cA * infl);
    eb_own = round_up_f32((cB + 1.001 * cC * k / 2.0 + 6.01 * infl * es * es) * infl);
    n2_offset = 2.0 * eS;
    // round 1: the three vectors of s
    pack_quad(e[0], e[1], e[3], true, bstage[lane][0][0], bstage[lane][0][1]);
    pack_quad(e[4], e[2], e[5], true, bstage[lane][1][0], bstage[lane][1][1]);
    pack_quad(e[6], e[7], e[8], true, bstage[lane][2][0], bstage[lane][2][1]);
    fetch_round(BE, 0, 3);
    // round 2: the quadratic forms k n1 and n2, one vector each
    pack_form(k * fma(e[0], e[0], e[3] * e[3]), 2.0 * k * fma(e[0], e[1], e[3] * e[4]), k * fma(e[1], e[1], e[4] * e[4]),
              2.0 * k * fma(e[0], e[2], e[3] * e[5]), 2.0 * k * fma(e[1], e[2], e[4] * e[5]), k * fma(e[2], e[2], e[5] * e[5]), true, bstage[lane][0][0],
              bstage[lane][0][1]);
    pack_form(fma(e[0], e[0], e[1] * e[1]), 2.0 * fma(e[0], e[3], e[1] * e[4]), fma(e[3], e[3], e[4] * e[4]), 2.0 * fma(e[0], e[6], e[1] * e[7]),
              2.0 * fma(e[3], e[6], e[4] * e[7]), fma(e[6], e[6], e[7] * e[7]) + n2_offset, true, bstage[lane][1][0], bstage[lane][1][1]);
    double elam = 1.0;
    if constexpr (kEarly) {
      // the early test s^2 < k1 n1 + c of eh_upper_mfma2_kernel (safety factor 2): a pair that fails it is no float64 inlier
      // its s comes from two chained MFMAs (terms of (x2 x1, x2 y1, y2 x1, y2 y1, x2), then of (y2, x1, y1, 1)), three piece products each
      const double est = (3.1 * 0x1.0p-18 / u * Sb + 32.0 * (1.0 + 0x1.0p-5) * (2.0 * (xm * xm * (A[0] + A[1] + A[3] + A[4]) + xm * A[2]) +
                                                                               (xm * (A[5] + A[6] + A[7]) + A[8]))) * u;
      const double en1 = (kn * N1b + 0x1.02p-9 * xm * xm * q1ab) * 1.0001;
      const double n1_ref = A[2] * A[2] + A[5] * A[5] + xm * xm / 3.0 * (A[0] * A[0] + A[1] * A[1] + A[3] * A[3] + A[4] * A[4]);
      const double mu = fmin(0.5, fmax(0x1.0p-7, safety * est / sqrt(k * n1_ref)));
      const double k1 = k * (1.0 + mu) * (1.0 + 32.0 * kEps32);
      const double floor_c = (1.0 + 1.0 / mu) * (safety * est) * (safety * est);
      const double cst = (floor_c + safety * k1 * en1) / (1.0 - safety * kn) * 1.0001;
      // The outcome is produced as clamp(N - s^2), exactly 0 or 1 (see eh_upper_mfma2_kernel: the hypothesis is scaled by a power of
      // two lambda with lambda^2 floor_c >= 2^50), and the outcomes of a row are gathered with acc = 2 acc + outcome, both
      // full-rate instructions (a v_alignbit per pair costs two issue slots).  Where that scale would leave float32's range,
      // or the threshold lies outside the range the argument covers, every pair of the hypothesis counts as passing.
      int ex = 0;
      (void)frexp(floor_c, &ex);  // floor_c >= 2^(ex - 1)
      int lg = (52 - ex) / 2;
      if (2 * lg < 51 - ex) ++lg;
      if (!(floor_c > 0.0) || lg > 58 || lg < -30 || !(k > 0x1.0p-45 && k < 0.55)) {
        early_all_own = 0xFFFFu;
        lg = 0;
      }
      elam = ldexp(1.0, lg);
      const double l2 = elam * elam * k1;
      pack_form(l2 * fma(e[0], e[0], e[3] * e[3]), 2.0 * l2 * fma(e[0], e[1], e[3] * e[4]), l2 * fma(e[1], e[1], e[4] * e[4]),
                2.0 * l2 * fma(e[0], e[2], e[3] * e[5]), 2.0 * l2 * fma(e[1], e[2], e[4] * e[5]), l2 * fma(e[2], e[2], e[5] * e[5]) + elam * elam * cst, true,
                bstage[lane][2][0], bstage[lane][2][1]);
    }
    fetch_round(BE, 3, kEarly ? 3 : 2);
    if constexpr (kEarly) {  // third staging pass: the two vectors of the early test's s
      const double c1[5] = {e[0] * elam, e[1] * elam, e[3] * elam, e[4] * elam, e[2] * elam}, c2[4] = {e[5] * elam, e[6] * elam, e[7] * elam, e[8] * elam};
      pack_tri<5>(c1, true, bstage[lane][0][0], bstage[lane][0][1]);
      pack_tri<4>(c2, true, bstage[lane][1][0], bstage[lane][1][1]);
      fetch_round(BT, 0, 2);
    }
  }
  if constexpr (kDoH) {
    double A[9];
    float unused[9];
    const double inv = to_unit_f32(Hd, unused, A);
    const double R0 = xm * (A[0] + A[1]) + A[2], R1 = xm * (A[3] + A[4]) + A[5], R2 = xm * (A[6] + A[7]) + A[8];
    const double rt = sqrt(thr_h2);
    // The test value is d = dx^2 + dy^2 - q^2 = P M + dy^2 with P = dx - q = x2 hz - (hx + rt hz), M = P + 2 q: the first
    // vector carries hx + rt hz.  e bounds |P_c - P|, |M_c - M| and |dy_c - dy| (matrix-core error of the rows, float32
    // rounding of x2, y2, 2 rt and of the three fused multiply-adds); the band follows as in t_score_mfma_kernel with
    // Q = rt |hz| at its bound over all matches.
    const double e = safety * (kEpsMfma + 4.0 * kEps32) * (fmax(R0, R1) + (xm + 3.0 * rt) * R2) * 1.001;
    const double Q = rt * R2;
    gh_own = !(xm < 1.0e5) ? __builtin_inff() : round_up_f32(1.001 * (4.0 * Q * e + 2.01 * kEps32 * Q * Q + 30.1 * e * e));
    pack_pieces((Hd[0] + rt * Hd[6]) * inv, (Hd[1] + rt * Hd[7]) * inv, (Hd[2] + rt * Hd[8]) * inv, true, bstage[lane][0][0], bstage[lane][0][1]);
#pragma unroll
    for (int v = 1; v < 3; ++v) pack_pieces(Hd[3 * v] * inv, Hd[3 * v + 1] * inv, Hd[3 * v + 2] * inv, true, bstage[lane][v][0], bstage[lane][v][1]);
    fetch_round(BH, 0, 3);
    if constexpr (kEarly) {
      // The early test: dx^2 + dy^2 < T hz^2 implies dx^2 < T hz^2.  dx = x2 hz - hx is a bilinear form in the monomials of the
      // vectors T1, T2 (two chained MFMAs, three piece products each) and T hz^2 a quadratic form in x1 (vector N1): the same
      // shape as the early test of the E launch, with the same slack algebra (|dx_c - dx| <= ed, AM-GM with mu, the form's own
      // error eq) and the same exact 0 / 1 outcome under a power-of-two scale of the hypothesis.
      double h[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) h[i] = Hd[i] * inv;
      const double u = kEps32, kn = 3.1 * 0x1.0p-18 + 34.0 * u;
      const double TA = xm * xm * (A[6] + A[7]) + xm * A[8], TB = xm * (A[0] + A[1]) + A[2];
      const double ed = (3.1 * 0x1.0p-18 / u * (TA + TB) + 32.0 * (1.0 + 0x1.0p-5) * (2.0 * TA + TB)) * u;
      const double hz_ref = sqrt(A[8] * A[8] + xm * xm / 3.0 * (A[6] * A[6] + A[7] * A[7]));
      const double mu = fmin(0.5, fmax(0x1.0p-7, safety * ed / (rt * hz_ref)));
      const double T1 = thr_h2 * (1.0 + mu) * (1.0 + 32.0 * kEps32);
      const double eq = (kn * T1 * R2 * R2 + 0x1.02p-9 * xm * xm * fabs(2.0 * T1 * h[6] * h[7])) * 1.0001;
      const double floor_c = (1.0 + 1.0 / mu) * (safety * ed) * (safety * ed);
      const double cst = (floor_c + safety * eq) / (1.0 - safety * kn) * 1.0001;
      int ex = 0;
      (void)frexp(floor_c, &ex);  // floor_c >= 2^(ex - 1)
      int lg = (52 - ex) / 2;
      if (2 * lg < 51 - ex) ++lg;
      if (!(floor_c > 0.0) || lg > 58 || lg < -30 || !(thr_h2 > 0x1.0p-45 && thr_h2 < 0.5) || !(xm < 1.0e5)) {
        early_all_own = 0xFFFFu;
        lg = 0;
      }
      const double lam = ldexp(1.0, lg), l2 = lam * lam * T1;
      const double c1[5] = {h[6] * lam, h[7] * lam, 0.0, 0.0, h[8] * lam}, c2[4] = {0.0, -h[0] * lam, -h[1] * lam, -h[2] * lam};
      pack_tri<5>(c1, true, bstage[lane][0][0], bstage[lane][0][1]);
      pack_tri<4>(c2, true, bstage[lane][1][0], bstage[lane][1][1]);
      pack_form(l2 * h[6] * h[6], 2.0 * l2 * h[6] * h[7], l2 * h[7] * h[7], 2.0 * l2 * h[6] * h[8], 2.0 * l2 * h[7] * h[8],
                l2 * h[8] * h[8] + lam * lam * cst, true, bstage[lane][2][0], bstage[lane][2][1]);
      fetch_round(BHe, 0, 3);
    }
  }
  float ea[2], eb[2], gh[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    ea[k] = __shfl(ea_own, 32 * k + col);
    eb[k] = __shfl(eb_own, 32 * k + col);
    gh[k] = __shfl(gh_own, 32 * k + col);
  }
  unsigned early_all[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) early_all[k] = (unsigned)__shfl((int)early_all_own, 32 * k + col);
  qcnt[wave][0][lane] = 0;
  if constexpr (kDoE && kDoH) qcnt[wave][1][lane] = 0;
  // The one-model instantiations (E at three, H at four waves per SIMD) are a dozen registers short: three (E) or four (H) piece vectors
  // stay in LDS behind the models (16-byte reads per tile) instead of being spilled to
  // scratch memory by the compiler

  uint4* parked = reinterpret_cast<uint4*>(&stage_raw[wave][kModel * 8 * 64]);
  // the staging area now holds the float64 models (in the solver's own scale) for the re-evaluation of pairs inside a band
  double* mstage = reinterpret_cast<double*>(&stage_raw[wave][0]);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    if constexpr (kDoE) mstage[kModel * lane + i] = Ed[i];
    if constexpr (kDoH) mstage[kModel * lane + (kDoE ? 9 : 0) + i] = Hd[i];
  }
  if constexpr (kPark) {
    if constexpr (kDoE) {
      if (early) {
#pragma unroll
        for (int v = 0; v < 4; ++v) parked[64 * v + lane] = __builtin_bit_cast(uint4, BE[v >> 1][kDoE ? 3 + (v & 1) : 0]);
      }
    } else {
#pragma unroll
      for (int v = 0; v < 3; ++v) parked[64 * v + lane] = __builtin_bit_cast(uint4, BH[1][kDoH ? v : 0]);
      parked[64 * 3 + lane] = __builtin_bit_cast(uint4, BH[0][kDoH ? 2 : 0]);
    }
  }
  __syncthreads();
  const float kinv = (float)(1.0 / two_thr_e2), two_rt = (float)(2.0 * sqrt(thr_h2));
  const int apos = 16 * ((col >> 2) & 1) + (col & 3) + 4 * (col >> 3);  // see eh_upper_mfma_kernel
  const int ntiles = (n + 31) / 32;
  const int t0 = blockIdx.y * tiles_per_split;
  const int t1 = min(ntiles, t0 + tiles_per_split);
  int ce[2] = {0, 0}, ch[2] = {0, 0};
  const f32x16 zero = {0};
  const unsigned tail_mask = tile_tail_mask(n, half);
  const uint4* src = pieces + ((long long)t0 * (kEhxVecA * 32) + apos) * 2 + half;
  const float4* xsrc = reinterpret_cast<const float4*>(xy + t0 * 32 + 16 * half);
  // Pairs inside a band are not evaluated where they are found (a lane or two of a wave at a time, with the other lanes idle,
  // on about half of all rows): their (match, hypothesis) ids are appended to a queue of the wave, and the queue is
  // worked off 64 pairs at a time with the float64 tests of eh_score_kernel; the verdicts are added to per-hypothesis
  // counters in LDS.  entry = match << 7 | test << 6 | hypothesis of the wave.
  const unsigned force_f64 = all_f64 ? 0xFFFFu : 0u, force_h = !(xm < 1.0e5) ? 0xFFFFu : 0u;  // wave-uniform
  int qlen = 0, qtot = 0;  // qtot: pairs queued so far (wave-uniform)
  auto drain = [&]() {
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < qlen; base += 64) {
      if (base + lane < qlen) {
        const unsigned e = qbuf[wave][base + lane];
        const int h = (int)(e & 63u), test_h = kDoE ? (kDoH ? (int)((e >> 6) & 1u) : 0) : 1;
        const double* M = mstage + kModel * h + ((kDoE && test_h) ? 9 : 0);
        double Mm[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Mm[i] = M[i];
        const double4 p = corr[e >> 7];
        bool in;
        if (test_h)
          in = transfer_inlier<double>(Mm, p.x, p.y, p.z, p.w, thr_h2);
        else
          in = epipolar_inlier<double>(Mm, p.x, p.y, p.z, p.w, two_thr_e2);
        if (in) atomicAdd(&qcnt[wave][(kDoE && kDoH) ? test_h : 0][h], 1);
      }
    }
    qlen = 0;
    __builtin_amdgcn_wave_barrier();
  };
  // Called by the whole wave with any = the lanes that have a pair to queue (not empty).  Both hypothesis groups of a tile go in
  // one pass (a pass costs as much for one group as for two): bits 0-15 of `both` = group 0, bits 16-31 = group 1, bit 15 - i of
  // either half = match first_match + i.  One entry per lane and round, so a pass takes as many rounds as the fullest lane has pairs.
  auto enqueue2 = [&](unsigned both, unsigned long long any, int test_h, int first_match) {
    do {
      const bool act = both != 0u;
      const int bit = __ffs((int)both) - 1;
      both &= both - 1u;
      const unsigned entry = ((unsigned)(first_match + 15 - (bit & 15)) << 7) | ((unsigned)test_h << 6) | (unsigned)(2 * (bit & 16) + col);
      const int pos = qlen + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(any >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)any, 0u));
      if (act) qbuf[wave][pos] = entry;
      qlen += __popcll(any);
      qtot += __popcll(any);
      if (qlen > kQueue - 64) drain();
      any = ballot64(both != 0u);
    } while (any);
  };
  __builtin_amdgcn_s_setprio(0);
  SFM_STAMP_BEGIN();
  // The match-side vectors of a tile are loaded one tile ahead: after the last MFMA of a tile has been issued their
  // registers are free, and the loads of the next tile then run under the vector work of the second hypothesis group.
  bf16x8 A1, AS1, AS2, AS3, AN1, AN2;
  auto load_tile = [&](const uint4* q) {
    if constexpr (kDoH) A1 = as_bf16x8(q[4 * 64]);
    if constexpr (kDoE) {
      AS1 = as_bf16x8(q[0]);
      AS2 = as_bf16x8(q[64]);
      AN1 = as_bf16x8(q[2 * 64]);
      AN2 = as_bf16x8(q[3 * 64]);
      AS3 = as_bf16x8(q[5 * 64]);
    }
  };
  bf16x8 AT1, AT2;
  auto load_early = [&](const uint4* q) {  // what the early loop of the E launch reads of a tile
    AT1 = as_bf16x8(q[6 * 64]);
    AT2 = as_bf16x8(q[7 * 64]);
    AN1 = as_bf16x8(q[2 * 64]);
  };
  if (t0 < t1) {
    if constexpr (kPark && kEarly)
      load_early(src);
    else
      load_tile(src);
  }
  int t = t0;
  // EARLY LOOP (one-model instantiations).  Each test is implied by a cheaper one whose failure proves a pair to be no float64
  // inlier: the symmetric epipolar test by the one-sided s^2 < k1 n1 + c of eh_upper_mfma2_kernel, the disc test
  // dx^2 + dy^2 < q^2 by its x strip dx^2 < (rt hz)^2 + c.  Either runs as three MFMAs (the left side's root from two chained
  // ones, the whole right side as one quadratic form with the slack in its constant term) and two issue slots per pair (the
  // outcome clamp(right - left^2), exactly 0 or 1 under the power-of-two scale of the hypothesis, and the gather
  // acc = 2 acc + outcome) instead of five MFMAs and ten slots (E) / three and eight (H).  The pairs that pass (a few per
  // thousand for most hypotheses) go straight to the float64 queue;
  // nothing else is counted here.  The queue costs about eight issue cycles per pair, the early loop saves 500 (E) / 350 (H)
  // per tile of 2048 pairs: a wave with several hypotheses from all-inlier samples is better off with the full float32
  // tests below.  It goes over to them for good when a tile has queued more than kDensePairs pairs, and at once (reloading
  // the tile it is on) when more than kDenseRows of the tile's 128 sixteen-pair rows hold a passing pair.
  if (kPark && early) {
    constexpr int kDenseRows = 40, kDensePairs = kDoE ? 64 : 32;
    for (; t < t1; ++t, xsrc += 16) {
      unsigned fill = t == (n >> 5) ? 0u : 0xFFFFu;
      asm volatile("" : "+s"(fill));
      const unsigned vmask = tail_mask | fill;
      const bool more = t + 1 < t1;
      if (more) src += kEhxVecA * 64;
      unsigned pass[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        unsigned hp = 0;
        if constexpr (kDoE) {
          f32x16 sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT1, BT[k][0], zero, 0, 0, 0);
          const f32x16 nv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AN1, BE[k][kDoE ? 5 : 0], zero, 0, 0, 0);
          sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT2, BT[k][(kDoE && kEarly) ? 1 : 0], sv, 0, 0, 0);
          if (k == 1) load_early(src);
          float acc = 0.0f;  // the outcomes of the row as a binary number: bit 15 - i = pair i (exact: below 2^16)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc = fma(acc, 2.0f, __builtin_amdgcn_fmed3f(fma(-sv[i], sv[i], nv[i]), 0.0f, 1.0f));  // 1 <=> passes
          hp = (unsigned)acc;
          pass[k] = (hp | force_f64 | early_all[k]) & vmask;
        } else {
          f32x16 sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT1, BHe[k][0], zero, 0, 0, 0);  // dx
          const f32x16 nv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AN1, BHe[k][(kDoH && kEarly) ? 2 : 0], zero, 0, 0, 0);  // (rt hz)^2 + c
          sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT2, BHe[k][(kDoH && kEarly) ? 1 : 0], sv, 0, 0, 0);
          if (k == 1) load_early(src);
          float acc = 0.0f;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc = fma(acc, 2.0f, __builtin_amdgcn_fmed3f(fma(-sv[i], sv[i], nv[i]), 0.0f, 1.0f));  // 1 <=> passes
          hp = (unsigned)acc;
          pass[k] = (hp | force_h | early_all[k]) & vmask;
        }
      }
      const unsigned long long any0 = ballot64(pass[0] != 0u), any1 = ballot64(pass[1] != 0u);
      if (__popcll(any0) + __popcll(any1) > kDenseRows) {
        if (more) src -= kEhxVecA * 64;  // this tile again (the next one was on its way)
        break;
      }
      const int before = qtot;
      if (any0 | any1) enqueue2(pass[0] | (pass[1] << 16), any0 | any1, kDoE ? 0 : 1, t * 32 + 16 * half);
      if (qtot - before > kDensePairs) {
        ++t;
        xsrc += 16;
        break;
      }
    }
    if (t < t1) load_tile(src);  // the early loop only ever loaded its own three vectors of a tile
    if constexpr (kDoE) {  // the vectors of k n1 and n2 come back for the full tests (unconditionally: they are dead inside the early loop)
      int at = lane;
      asm volatile("" : "+v"(at));
#pragma unroll
      for (int v = 0; v < 4; ++v) BE[v >> 1][kDoE ? 3 + (v & 1) : 0] = as_bf16x8(parked[64 * v + at]);
    }
  }
  for (; t < t1; ++t, xsrc += 16) {
    // (a v_cndmask costs 18 issue cycles of the SIMD on this chip: the per-tile masks are formed with scalar selects and an OR)
    unsigned fill = t == (n >> 5) ? 0u : 0xFFFFu;
    asm volatile("" : "+s"(fill));  // keeps the select scalar
    const unsigned vmask = tail_mask | fill;
    if (t + 1 < t1) src += kEhxVecA * 64;  // the last tile loads itself again
    float x2[kDoH ? 16 : 1], y2[kDoH ? 16 : 1];
    if constexpr (kDoH) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 v = xsrc[i];
        x2[2 * i] = v.x; y2[2 * i] = v.y; x2[2 * i + 1] = v.z; y2[2 * i + 1] = v.w;
      }
    }
    unsigned both_e = 0, both_h = 0;  // bits 0-15: group 0, bits 16-31: group 1
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      unsigned amb_e = 0, amb_h = 0;
      if constexpr (kDoE) {
        bf16x8 bn[2] = {BE[k][3], BE[k][4]};
        if constexpr (kPark && kParkedE > 0) {
          if (k == 1) {
            int at = lane;
            asm volatile("" : "+v"(at));  // keeps the reads inside the loop (hoisted, they would be spilled again)
#pragma unroll
            for (int v = 0; v < kParkedE; ++v) bn[2 - kParkedE + v] = as_bf16x8(parked[64 * v + at]);
          }
        }
        f32x16 sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AS1, BE[k][0], zero, 0, 0, 0);
        const f32x16 n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AN1, bn[0], zero, 0, 0, 0);
        sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AS2, BE[k][1], sv, 0, 0, 0);
        const f32x16 n2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AN2, bn[1], zero, 0, 0, 0);
        sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AS3, BE[k][2], sv, 0, 0, 0);
        if constexpr (!kDoH) {
          if (k == 1) load_tile(src);
        }
        unsigned hneg = 0, hamb = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float s2 = sv[i] * sv[i];
          const float S = fma(n1[i], kinv, n2[i]);  // n1 comes as k n1, n2 with its offset: S >= the true n1 + n2 + eS
          const float P = n1[i] * n2[i];
          const float dv = fma(s2, S, -P);  // < 0 <=> inlier
          const float w = fma(fabsf(sv[i]), ea[k], eb[k]);
          const float rv = fma(-S, w, fabsf(dv));  // < 0 <=> the float32 sign of dv is not trusted
          hneg = __builtin_amdgcn_alignbit(hneg, __float_as_uint(dv), 31);
          hamb = __builtin_amdgcn_alignbit(hamb, __float_as_uint(rv), 31);
        }
        amb_e = (hamb | force_f64) & vmask;  // padding has dv = rv = -0 or +0: masked either way
        ce[k] += __popc(hneg & vmask & ~amb_e);
      }
      if constexpr (kDoH) {
        bf16x8 h0 = BH[k][0], h1 = BH[k][1], h2 = BH[k][2];
        if constexpr (kPark) {
          int at = lane;
          asm volatile("" : "+v"(at));
          if (k == 1) {
            h0 = as_bf16x8(parked[at]);
            h1 = as_bf16x8(parked[64 + at]);
            h2 = as_bf16x8(parked[128 + at]);
          } else {
            h2 = as_bf16x8(parked[192 + at]);
          }
        }
        const f32x16 hx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, h0, zero, 0, 0, 0);
        const f32x16 hy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, h1, zero, 0, 0, 0);
        const f32x16 hz = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, h2, zero, 0, 0, 0);
        if (k == 1) load_tile(src);
        auto dH = [&](int i) {  // < 0 <=> inlier (hx holds hx + rt hz)
          const float P = fma(x2[i], hz[i], -hx[i]);
          const float M = fma(two_rt, hz[i], P);
          const float dy = fma(y2[i], hz[i], -hy[i]);
          return fma(P, M, dy * dy);
        };
        // the band is one constant per hypothesis: the loop keeps the smallest |dH| of the row (half a v_min3 per pair) and
        // only a wave in which some row trips the band looks for the pairs (all its lanes walk that block: no lane waits)
        unsigned hneg = 0;
        float closest = __builtin_inff();
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const float d0 = dH(i), d1 = dH(i + 1);
          hneg = __builtin_amdgcn_alignbit(hneg, __float_as_uint(d0), 31);
          hneg = __builtin_amdgcn_alignbit(hneg, __float_as_uint(d1), 31);
          closest = min3_abs(closest, d0, d1);
        }
        ch[k] += __popc(hneg & vmask);
        if (ballot64(closest <= gh[k])) {
          unsigned hamb = 0;
#pragma unroll
          for (int i = 0; i < 16; ++i) hamb = __builtin_amdgcn_alignbit(hamb, __float_as_uint(fabsf(dH(i)) - gh[k]), 31);  // the same values again
          amb_h = (hamb | force_h) & vmask;
          ch[k] -= __popc(hneg & amb_h);  // take the untrusted signs back out
        }
      }
      SFM_STAMP_COUNT((amb_e | amb_h) != 0u);
      both_e |= amb_e << (16 * k);
      both_h |= amb_h << (16 * k);
    }
    // the pairs inside a band of both hypothesis groups are queued in one pass per tile and test
    if constexpr (kDoE) {
      if (const unsigned long long any = ballot64(both_e != 0u)) enqueue2(both_e, any, 0, t * 32 + 16 * half);
    }
    if constexpr (kDoH) {
      if (const unsigned long long any = ballot64(both_h != 0u)) enqueue2(both_h, any, 1, t * 32 + 16 * half);
    }
  }
  drain();
  SFM_STAMP_END(t1 - t0);
  int se[2], sh[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    se[k] = ce[k] + __shfl_xor(ce[k], 32);
    sh[k] = ch[k] + __shfl_xor(ch[k], 32);
  }
  // thread ids are formed again here: values derived from them and held across the tile loop only for these last lines cost
  // registers there (a spill at three waves per SIMD)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane_o = tid & 63, wave_o = tid >> 6, half_o = lane_o >> 5;
  const unsigned ok = fin[wave_o][lane_o];
  const int oute = (ok & 1u) ? (half_o ? se[1] : se[0]) + (kDoE ? qcnt[wave_o][0][lane_o] : 0) : 0;
  const int outh = (ok & 2u) ? (half_o ? sh[1] : sh[0]) + (kDoH ? qcnt[wave_o][(kDoE && kDoH) ? 1 : 0][lane_o] : 0) : 0;
  const long long out_id = (long long)blockIdx.x * blockDim.x + tid;
  if (out_id < nb) {
    if (use_atomics) {
      if constexpr (kDoE) atomicAdd(count_e + out_id, oute);
      if constexpr (kDoH) atomicAdd(count_h + out_id, outh);
    } else {
      if constexpr (kDoE) count_e[out_id] = oute;
      if constexpr (kDoH) count_h[out_id] = outh;
    }
  }
}

__global__ __launch_bounds__(kBlock) void corr_to_f32_kernel(const double4* __restrict__ corr, int n, float4* __restrict__ corr32) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  double4 c = make_double4(0.0, 0.0, 0.0, 0.0);
  if (j < n) {
    c = corr[j];
    corr32[j] = make_float4((float)c.x, (float)c.y, (float)c.z, (float)c.w);
  }
  publish_absmax(c, j < n, reinterpret_cast<unsigned*>(corr32 + n));  // trailer row n, zeroed by the launcher
}

__global__ __launch_bounds__(kBlock) void eh_models_kernel(const double4* __restrict__ corr, int n, const int32_t* __restrict__ idx, long long nb,
                                                           double* __restrict__ F_out, double* __restrict__ E_out,
                                                           double* __restrict__ H_out) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long h = gid < nb ? gid : nb - 1;
  double F[9], E[9], H[9];
  double p8[8][4], p4[4][4];
  gather_samples<double>(corr, n, idx + 12 * h, p8, p4);
  solve_E8<double>(p8, E, F);
  solve_H4<double>(p4, H);
  if (gid < nb) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      if (F_out) F_out[9 * gid + i] = F[i];
      if (E_out) E_out[9 * gid + i] = E[i];
      if (H_out) H_out[9 * gid + i] = H[i];
    }
  }
}

// ---------------------------------------------------------------------------------------
// DLT over ANY number of correspondences (src/ransac_algorithm.py:216-263 and :266-293 loop over all columns they are
// given; the pipeline itself only ever passes minimal samples, which the in-register solvers above serve).
// One wave: every lane folds its share of the rows of the n x 9 (F) or 2n x 9 (H) system into a 9 x 9 upper triangle
// with Givens rotations (a QR factorisation row by row: the right singular vectors of the triangle are those of the whole
// system, and nothing is squared), the 64 triangles are merged by a butterfly of the same rotations, and the right
// singular vector of the smallest singular value of the merged triangle comes from the one-sided Jacobi of sfm_math.hpp.
// ---------------------------------------------------------------------------------------
template <int kFrom>
__device__ __forceinline__ void givens_fold_row(double (&R)[9][9], double (&a)[9]) {
#pragma unroll
  for (int k = kFrom; k < 9; ++k) {
    const double r = sqrt(fma(R[k][k], R[k][k], a[k] * a[k]));
    const bool act = r > 0.0;
    const double c = act ? R[k][k] / r : 1.0, sn = act ? a[k] / r : 0.0;
#pragma unroll
    for (int j = k; j < 9; ++j) {
      const double t = fma(c, R[k][j], sn * a[j]);
      a[j] = fma(-sn, R[k][j], c * a[j]);
      R[k][j] = t;
    }
  }
}

template <int kRow>
__device__ __forceinline__ void merge_triangle_rows(double (&R)[9][9], const double (&O)[9][9]) {
  if constexpr (kRow < 9) {
    double a[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) a[j] = j >= kRow ? O[kRow][j] : 0.0;
    givens_fold_row<kRow>(R, a);
    merge_triangle_rows<kRow + 1>(R, O);
  }
}

__global__ __launch_bounds__(64) void dlt_general_kernel(const double4* __restrict__ corr, int n, int homography, double* __restrict__ out) {
  const int lane = threadIdx.x;
  double R[9][9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) R[i][j] = 0.0;
  for (int p = lane; p < n; p += 64) {
    const double4 c = corr[p];
    const double x1 = c.x, y1 = c.y, x2 = c.z, y2 = c.w;
    if (homography) {  // :281-282
      double a[9] = {-x1, -y1, -1.0, 0.0, 0.0, 0.0, x2 * x1, x2 * y1, x2};
      givens_fold_row<0>(R, a);
      double b[9] = {0.0, 0.0, 0.0, -x1, -y1, -1.0, y2 * x1, y2 * y1, y2};
      givens_fold_row<0>(R, b);
    } else {  // :243-253
      double a[9] = {x1 * x2, x1 * y2, x1, y1 * x2, y1 * y2, y1, x2, y2, 1.0};
      givens_fold_row<0>(R, a);
    }
  }
#pragma unroll 1
  for (int off = 32; off > 0; off >>= 1) {
    double O[9][9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int j = 0; j < 9; ++j) O[i][j] = j >= i ? __shfl_xor(R[i][j], off) : 0.0;
    merge_triangle_rows<0>(R, O);
  }
  // every lane now holds a triangle of the whole system (equal up to rounding and row signs); lane 0's is used
  double g[9][9], v[9][9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) g[i][j] = __shfl(R[i][j], 0);
  jacobi_svd<double, 9>(g, v);
  double best = INFINITY;
  int kmin = 0;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    double nj = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) nj = fma(g[i][j], g[i][j], nj);
    if (nj < best) {
      best = nj;
      kmin = j;
    }
  }
  if (lane == 0) {
    double q[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < 9; ++j) t = (j == kmin) ? v[i][j] : t;  // static indices: the column is selected, not addressed
      q[i] = t;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      if (homography)
        out[i] = q[i] / q[8];  // reshape(3, 3) / H[2, 2]  (:290-293)
      else
        out[3 * (i % 3) + i / 3] = q[i];  // F = v.reshape(3, 3).T  (:261)
    }
  }
}

// ---------------------------------------------------------------------------------------
// Record-event kernels: one point per lane.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int block_count(bool flag) {
  // per-wave ballot/popcount, then one LDS add per wave
  __shared__ int wave_sum[kBlock / 64];
  const unsigned long long b = __ballot(flag);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_sum[wave] = __popcll(b);
  __syncthreads();
  int total = 0;
  if (threadIdx.x == 0) {
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += wave_sum[w];
  }
  __syncthreads();
  return total;  // valid in thread 0
}

// models of ONE hypothesis whose index row arrives as a kernel argument (record events of the replay)
struct RowArg {
  int32_t v[12];
};

template <bool kEpipolar>
__global__ __launch_bounds__(kBlock) void mask_kernel(const double* __restrict__ M, const double4* __restrict__ corr, int n, double thr_arg,
                                                      uint8_t* __restrict__ mask, int* __restrict__ count) {
  double m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = M[i];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  bool in = false;
  if (j < n) {
    const double4 p = corr[j];
    in = kEpipolar ? epipolar_inlier<double>(m, p.x, p.y, p.z, p.w, thr_arg) : transfer_inlier<double>(m, p.x, p.y, p.z, p.w, thr_arg);
    mask[j] = in ? 1 : 0;
  }
  const int c = block_count(in);
  if (count && threadIdx.x == 0 && c) atomicAdd(count, c);
}

// |l . x2| / sqrt(l0^2 + l1^2) with l = F x1, for callers that want the distances themselves
// (src/ransac_algorithm.py:193-213).  x1, x2 are the 3xN arrays (row stride ld).
__global__ __launch_bounds__(kBlock) void epipolar_errors_kernel(const double* __restrict__ F, const double* __restrict__ x1,
                                                                 const double* __restrict__ x2, int n, long long ld, double* __restrict__ d) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const double a = x1[j], b = x1[ld + j], c = x1[2 * ld + j];
  double l[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) l[i] = (F[3 * i] * a + F[3 * i + 1] * b) + F[3 * i + 2] * c;
  const double nrm = sqrt(l[0] * l[0] + l[1] * l[1]);
  d[j] = fabs(((l[0] / nrm) * x2[j] + (l[1] / nrm) * x2[ld + j]) + (l[2] / nrm) * x2[2 * ld + j]);
}

__global__ void enforce_essential_kernel(const double* __restrict__ F, double* __restrict__ E) {
  double A[3][3], U[3][3], S[3], V[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) A[i][j] = F[3 * i + j];
  svd3<double>(A, U, S, V);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) E[3 * i + j] = fma(U[i][0], V[j][0], U[i][1] * V[j][1]);
  }
}

__global__ void essential_candidates_kernel(const double* __restrict__ E, double* __restrict__ P2s) {
  double e[9], P[4][12];
#pragma unroll
  for (int i = 0; i < 9; ++i) e[i] = E[i];
  essential_candidates<double>(e, P);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 12; ++i) P2s[12 * c + i] = P[c][i];
  }
}

__global__ __launch_bounds__(kBlock) void chirality4_kernel(const double* __restrict__ P2s, const double4* __restrict__ corr,
                                                            const uint8_t* __restrict__ mask, int n, int* __restrict__ counts) {
  const double P1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = gid < n ? gid : n - 1;
  const bool live = gid < n && (mask == nullptr || mask[j] != 0);
  const double4 p = corr[j];
  for (int c = 0; c < 4; ++c) {
    double P2[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) P2[i] = P2s[12 * c + i];
    double X[4];
    triangulate_point<double>(P1, P2, p.x, p.y, p.z, p.w, X);
    // (P2 X)_z and (P1 X)_z with X = v / v_3, as in src/chirality_check.py:56
    const double z2 = ((P2[8] * X[0] + P2[9] * X[1]) + P2[10] * X[2]) + P2[11] * X[3];
    const double z1 = ((P1[8] * X[0] + P1[9] * X[1]) + P1[10] * X[2]) + P1[11] * X[3];
    const int cnt = block_count(live && z2 > 0.0 && z1 > 0.0);
    if (threadIdx.x == 0 && cnt) atomicAdd(counts + c, cnt);
  }
}

// one block: sign vote over all points, then thread 0 decomposes
__global__ __launch_bounds__(1024) void homography_to_rt_kernel(const double* __restrict__ H, const double4* __restrict__ corr, int n,
                                                                double* __restrict__ out) {
  __shared__ int votes;
  if (threadIdx.x == 0) votes = 0;
  __syncthreads();
  double h[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) h[i] = H[i];
  int mine = 0;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const double4 p = corr[j];
    // sum(x2 * (H @ x1)) > 0, src/ransac_algorithm.py:359
    const double hx = (h[0] * p.x + h[1] * p.y) + h[2];
    const double hy = (h[3] * p.x + h[4] * p.y) + h[5];
    const double hz = (h[6] * p.x + h[7] * p.y) + h[8];
    const double s = (p.z * hx + p.w * hy) + hz;
    mine += s > 0.0 ? 1 : 0;
  }
  atomicAdd(&votes, mine);
  __syncthreads();
  const bool flip = (double)votes < (double)n / 2.0;
  double R1[3][3], R2[3][3], t1[3], t2[3], Ea[9], Eb[9];
  homography_decompose<double>(h, flip, R1, t1, R2, t2);
  skew_times<double>(t1, R1, Ea);
  skew_times<double>(t2, R2, Eb);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      out[i] = R1[i / 3][i % 3];
      out[12 + i] = R2[i / 3][i % 3];
      out[24 + i] = Ea[i];
      out[33 + i] = Eb[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      out[9 + i] = t1[i];
      out[21 + i] = t2[i];
    }
    out[42] = flip ? 1.0 : 0.0;
  }
}

__global__ __launch_bounds__(kBlock) void triangulate_kernel(const double* __restrict__ P1g, const double* __restrict__ P2g,
                                                             const double* __restrict__ x1, const double* __restrict__ x2, int m,
                                                             long long ld, double* __restrict__ X) {
  double P1[12], P2[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    P1[i] = P1g[i];
    P2[i] = P2g[i];
  }
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = gid < m ? gid : m - 1;
  double out[4];
  triangulate_point<double>(P1, P2, x1[j], x1[ld + j], x2[j], x2[ld + j], out);
  if (gid < m) {
#pragma unroll
    for (int i = 0; i < 4; ++i) X[(long long)i * m + j] = out[i];
  }
}

// ---------------------------------------------------------------------------------------
// A record event of the E/H loop as ONE launch.  Every workgroup repeats the short serial part (model of the sample,
// sign vote over all matches, decomposition, motion candidates: wave 0, the results handed to the other waves
// through LDS) and then handles its own 256 matches (inlier masks, four triangulations per motion set, chirality
// counts), so the event needs neither a second launch nor a grid-wide synchronisation.  Same device functions, same
// operation order as the single-purpose kernels above: identical masks, counts and candidates.
// scratch layout (doubles): E 0..8, H 9..17, P2s 18..65, rt 70..112, P2s of the second solution 113..160;
// (int32) chirality counts at double slots 66 and 161, mask counts at 68 and 163; the launcher zeroes the counters.
// ---------------------------------------------------------------------------------------
constexpr int kRecE = 0, kRecH = 9, kRecP = 18, kRecCnt = 66, kRecMaskCnt = 68, kRecRt = 70, kRecP2 = 113, kRecCnt2 = 161, kRecMaskCnt2 = 163;
constexpr int kRecDoubles = 256;

// chirality counts of the four cameras P[0..3] (LDS) for this thread's match; adds the workgroup's counts to counts[0..3]
__device__ __forceinline__ void chirality4_block(const double* __restrict__ P, const double4& p, bool live, int* __restrict__ counts) {
  const double P1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  for (int c = 0; c < 4; ++c) {
    double P2[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) P2[i] = P[12 * c + i];
    double X[4];
    triangulate_point<double>(P1, P2, p.x, p.y, p.z, p.w, X);
    const double z2 = ((P2[8] * X[0] + P2[9] * X[1]) + P2[10] * X[2]) + P2[11] * X[3];
    const double z1 = ((P1[8] * X[0] + P1[9] * X[1]) + P1[10] * X[2]) + P1[11] * X[3];
    const int cnt = block_count(live && z2 > 0.0 && z1 > 0.0);
    if (threadIdx.x == 0 && cnt) atomicAdd(counts + c, cnt);
  }
}

__global__ __launch_bounds__(kBlock) void e_record_kernel(const double4* __restrict__ corr, int n, RowArg row, double two_thr_e2,
                                                          uint8_t* __restrict__ mask, double* __restrict__ scratch) {
  __shared__ double sE[9], sP[48];
  if (threadIdx.x < 64) {  // wave 0, all lanes on the same sample
    double E[9], H[9], P[4][12];
    {
      double p8[8][4], p4[4][4];
      gather_samples<double>(corr, n, row.v, p8, p4);
      solve_E8<double>(p8, E);
      solve_H4<double>(p4, H);
    }
    essential_candidates<double>(E, P);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sE[i] = E[i];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 12; ++i) sP[12 * c + i] = P[c][i];
      if (blockIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          scratch[kRecE + i] = E[i];
          scratch[kRecH + i] = H[i];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int i = 0; i < 12; ++i) scratch[kRecP + 12 * c + i] = P[c][i];
      }
    }
  }
  __syncthreads();
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const double4 p = corr[gid < n ? gid : n - 1];
  double E[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) E[i] = sE[i];
  const bool in = gid < n && epipolar_inlier<double>(E, p.x, p.y, p.z, p.w, two_thr_e2);
  if (gid < n) mask[gid] = in ? 1 : 0;
  const int c = block_count(in);
  if (threadIdx.x == 0 && c) atomicAdd(reinterpret_cast<int*>(scratch + kRecMaskCnt), c);
  chirality4_block(sP, p, gid < n, reinterpret_cast<int*>(scratch + kRecCnt));  // essential_to_RT over ALL matches (:110)
}

// grid = (match tiles, events).  Event e takes its index set from rows[e] (device memory) when rows is given, else
// from `row`; its masks are mask_a/mask_b + e * n and its scratch block is scratch + e * kRecDoubles.
__global__ __launch_bounds__(kBlock) void h_record_kernel(const double4* __restrict__ corr, int n, RowArg row, const int32_t* __restrict__ rows,
                                                          double two_thr_e2, uint8_t* __restrict__ mask_a, uint8_t* __restrict__ mask_b,
                                                          double* __restrict__ scratch) {
  __shared__ double sH[9], sRt[43], sP[2][48];
  __shared__ int votes;
  if (rows) {
#pragma unroll
    for (int i = 0; i < 12; ++i) row.v[i] = rows[12 * blockIdx.y + i];
  }
  mask_a += (size_t)blockIdx.y * n;
  mask_b += (size_t)blockIdx.y * n;
  scratch += (size_t)blockIdx.y * kRecDoubles;
  if (threadIdx.x == 0) votes = 0;
  if (threadIdx.x < 64) {
    double H[9];
    {
      double p8[8][4], p4[4][4];
      gather_samples<double>(corr, n, row.v, p8, p4);
      solve_H4<double>(p4, H);
    }
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sH[i] = H[i];
    }
  }
  __syncthreads();
  {
    // sign vote over ALL matches, repeated by every workgroup: sum(x2 * (H @ x1)) > 0, src/ransac_algorithm.py:359
    double h[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = sH[i];
    int mine = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      const double4 q = corr[j];
      const double hx = (h[0] * q.x + h[1] * q.y) + h[2];
      const double hy = (h[3] * q.x + h[4] * q.y) + h[5];
      const double hz = (h[6] * q.x + h[7] * q.y) + h[8];
      const double sgn = (q.z * hx + q.w * hy) + hz;
      mine += sgn > 0.0 ? 1 : 0;
    }
    if (mine) atomicAdd(&votes, mine);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    double h[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = sH[i];
    const bool flip = (double)votes < (double)n / 2.0;
    double R1[3][3], R2[3][3], t1[3], t2[3], Ea[9], Eb[9];
    homography_decompose<double>(h, flip, R1, t1, R2, t2);
    skew_times<double>(t1, R1, Ea);
    skew_times<double>(t2, R2, Eb);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        sRt[i] = R1[i / 3][i % 3];
        sRt[12 + i] = R2[i / 3][i % 3];
        sRt[24 + i] = Ea[i];
        sRt[33 + i] = Eb[i];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        sRt[9 + i] = t1[i];
        sRt[21 + i] = t2[i];
      }
      sRt[42] = flip ? 1.0 : 0.0;
    }
    double P[4][12];
    essential_candidates<double>(Ea, P);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 12; ++i) sP[0][12 * c + i] = P[c][i];
    }
    essential_candidates<double>(Eb, P);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 12; ++i) sP[1][12 * c + i] = P[c][i];
    }
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (threadIdx.x < 9) scratch[kRecH + threadIdx.x] = sH[threadIdx.x];
    if (threadIdx.x < 43) scratch[kRecRt + threadIdx.x] = sRt[threadIdx.x];
    if (threadIdx.x < 48) {
      scratch[kRecP + threadIdx.x] = sP[0][threadIdx.x];
      scratch[kRecP2 + threadIdx.x] = sP[1][threadIdx.x];
    }
  }
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const double4 p = corr[gid < n ? gid : n - 1];
  for (int k = 0; k < 2; ++k) {  // the two solutions: epipolar mask of E = [t]x R, essential_to_RT over the masked matches (:133-147)
    double E[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) E[i] = sRt[(k ? 33 : 24) + i];
    const bool in = gid < n && epipolar_inlier<double>(E, p.x, p.y, p.z, p.w, two_thr_e2);
    if (gid < n) (k ? mask_b : mask_a)[gid] = in ? 1 : 0;
    const int c = block_count(in);
    if (threadIdx.x == 0 && c) atomicAdd(reinterpret_cast<int*>(scratch + (k ? kRecMaskCnt2 : kRecMaskCnt)), c);
    chirality4_block(sP[k], p, in, reinterpret_cast<int*>(scratch + (k ? kRecCnt2 : kRecCnt)));
  }
}

// ---------------------------------------------------------------------------------------
// Translation stages
// ---------------------------------------------------------------------------------------
template <typename TX>
__global__ __launch_bounds__(kBlock) void pack_points_kernel(const double* __restrict__ R, const TX* __restrict__ X,
                                                             const double* __restrict__ x, int m, double* __restrict__ pts) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const double X0 = (double)X[j], X1 = (double)X[m + j], X2 = (double)X[2LL * m + j];
#pragma unroll
  for (int i = 0; i < 3; ++i) pts[6LL * j + i] = (R[3 * i] * X0 + R[3 * i + 1] * X1) + R[3 * i + 2] * X2;
  pts[6LL * j + 3] = x[j];
  pts[6LL * j + 4] = x[m + j];
  pts[6LL * j + 5] = x[2LL * m + j];
}

__device__ __forceinline__ void load_t_sample(const double* __restrict__ pts, int m, const int32_t* __restrict__ pair, double (&x)[2][3],
                                              double (&r)[2][3]) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double* p = pts + 6LL * clamp_index(pair[k], m);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      r[k][i] = p[i];
      x[k][i] = p[3 + i];
    }
  }
}

// Two DIFFERENT indices can name the same correspondence twice (a keypoint with two orientations matched to one 3-D
// point): the 6 x 3 system then has rank 2 exactly as for a repeated index, and gelsd's minimum-norm solution is what
// the reference computes (src/estimate_translation.py:104), so such a sample takes solve_T2's closed form as well.
__device__ __forceinline__ bool same_correspondence(const double (&x)[2][3], const double (&r)[2][3]) {
  bool same = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) same = same && x[0][i] == x[1][i] && r[0][i] == r[1][i];
  return same;
}

// grid = (hypothesis tiles, cameras, point splits)
__global__ __launch_bounds__(kBlock) void t_score_kernel(const double* __restrict__ pts_all, CamOffsets offs, const int32_t* __restrict__ idx,
                                                         long long nb, double thr2, int* __restrict__ counts, int points_per_split,
                                                         int use_atomics) {
  const int cam = blockIdx.y;
  const double* __restrict__ pts = pts_all + 6 * offs.off[cam];
  const int m = (int)(offs.off[cam + 1] - offs.off[cam]);
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long h = gid < nb ? gid : nb - 1;
  const int32_t* pair = idx + 2 * ((long long)cam * nb + h);
  double T[3];
  {
    double x[2][3], r[2][3];
    load_t_sample(pts, m, pair, x, r);
    solve_T2<double>(x, r, pair[0] == pair[1] || same_correspondence(x, r), T);
  }
  const int j0 = blockIdx.z * points_per_split;
  const int j1 = min(m, j0 + points_per_split);
  int c = 0;
#pragma unroll 4
  for (int j = j0; j < j1; ++j) {
    const double* p = pts + 6LL * j;
    c += reprojection_inlier<double>(T[0], T[1], T[2], p[0], p[1], p[2], p[3], p[4], thr2) ? 1 : 0;
  }
  if (gid < nb) {
    int* dst = counts + (long long)cam * nb + gid;
    if (use_atomics)
      atomicAdd(dst, c);
    else
      *dst = c;
  }
}

__global__ __launch_bounds__(kBlock) void t_models_kernel(const double* __restrict__ pts, int m, const int32_t* __restrict__ idx, long long nb,
                                                          double* __restrict__ T_out) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nb) return;
  double x[2][3], r[2][3], T[3];
  load_t_sample(pts, m, idx + 2 * gid, x, r);
  solve_T2<double>(x, r, idx[2 * gid] == idx[2 * gid + 1] || same_correspondence(x, r), T);
#pragma unroll
  for (int i = 0; i < 3; ++i) T_out[3 * gid + i] = T[i];
}

// float32 pre-filter of the translation RANSAC: upper bounds of the counts t_score_kernel returns.
// T is solved in float64 and rounded; points come from a float32 copy [m][8] = (RX, x, y, pad);
// the slack covers input rounding and the float32 evaluation (see reprojection_upper_f32), with
// |RX| <= rmax[cam] and |x|,|y| <= xmax.
struct CamBounds {
  double rmax[kMaxCams];
};

__global__ __launch_bounds__(kBlock) void t_upper_kernel(const double* __restrict__ pts_all, const float* __restrict__ pts32_all, CamOffsets offs,
                                                         CamBounds bounds, const int32_t* __restrict__ idx, long long nb, double thr2,
                                                         double xmax, double safety, int* __restrict__ up, int points_per_split,
                                                         int use_atomics) {
  const int cam = blockIdx.y;
  const double* __restrict__ pts = pts_all + 6 * offs.off[cam];
  const float* __restrict__ pts32 = pts32_all + 8 * offs.off[cam];
  const int m = (int)(offs.off[cam + 1] - offs.off[cam]);
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long h = gid < nb ? gid : nb - 1;
  const int32_t* pair = idx + 2 * ((long long)cam * nb + h);
  float Tf[3], Tthr, c1, c0;
  {
    double T[3], x[2][3], r[2][3];
    load_t_sample(pts, m, pair, x, r);
    solve_T2<double>(x, r, pair[0] == pair[1] || same_correspondence(x, r), T);
#pragma unroll
    for (int i = 0; i < 3; ++i) Tf[i] = (float)T[i];
    const double P = bounds.rmax[cam] + fmax(fabs(T[0]), fmax(fabs(T[1]), fabs(T[2])));
    const double ez = 2.0 * kEps32 * P;
    const double D1 = 8.0 * kEps32 * P * (1.0 + xmax);
    const double thr = sqrt(thr2);
    const double G = safety * (thr * ez + 1.4142135623730951 * D1);
    Tthr = round_up_f32(thr2 * (1.0 + 16.0 * kEps32));
    c1 = round_up_f32(2.0 * thr * G * (1.0 + 16.0 * kEps32));
    c0 = round_up_f32(G * G * (1.0 + 16.0 * kEps32));
  }
  const int j0 = blockIdx.z * points_per_split;
  const int j1 = min(m, j0 + points_per_split);
  int c = 0;
#pragma unroll 8
  for (int j = j0; j < j1; ++j) {
    const float* p = pts32 + 8LL * j;
    c += reprojection_upper_f32(Tf[0], Tf[1], Tf[2], p[0], p[1], p[2], p[3], p[4], Tthr, c1, c0) ? 1 : 0;
  }
  if (gid < nb) {
    int* dst = up + (long long)cam * nb + gid;
    if (use_atomics)
      atomicAdd(dst, c);
    else
      *dst = c;
  }
}

// ---------------------------------------------------------------------------------------
// Translation RANSAC with EXACT float64 counts at matrix-core + float32 cost.
//   dx = (rx - u rz) + tx - u tz,  dy = (ry - v rz) + ty - v tz,  q = thr (rz + tz)
// are bilinear in (hypothesis, correspondence), and so are P = dx - q and M = dx + q: P, M and dy are one bf16 MFMA
// each per 32 x 32 pairs from three-piece splits (same expansion and error bound as in eh_upper_mfma_kernel), and the
// test value d = dx^2 + dy^2 - q^2 = P M + dy^2 costs two float32 instructions per pair.  Its sign decides the pair
// unless |d_c| <= delta, a rigorous bound of |d_c - d| for any pair whose sign could be wrong:
//   with e >= |P_c - P|, |M_c - M|, |dy_c - dy|, r = |(dx, dy)|, Q = |q| and u = 2^-24,
//   |P_c M_c + dy_c^2 - d| <= (|P| + |M| + 2 |dy| + 4 e) e + 2 e^2 <= (2 max(r, Q) + 2 r) e + 6 e^2  =: m,
//   |d_c - d| <= (m (1 + u) + 1.001 u (r + e)^2) / (1 - u);
//   a wrong sign needs |d| <= |d_c - d|; for r = Q + x, x >= 6 e + 2 u Q, d = x (2 Q + x) exceeds that bound, so
//   r < Q (1 + 2 u) + 6 e, and then |d_c| <= |d_c - d| <= 4 Q e + 2.01 u Q^2 + 30.1 e^2 (times 1.001 for the rounding of the
//   float64 test itself, which is 2^-26 of these terms).
// delta takes Q at its bound over the camera's correspondences and is therefore ONE constant per hypothesis: the inner
// loop only keeps the smallest |d_c| of a register row (half a v_min3 per pair) and compares it with delta once per
// 16 pairs.  Pairs inside the band (about one in 10^6) are queued and re-evaluated with the float64 test of
// t_score_kernel, so the counts are identical to t_score_kernel's for every hypothesis.
// ---------------------------------------------------------------------------------------
struct CamPad {
  int64_t poff[kMaxCams + 1];  // offsets of the cameras in the padded (multiple of 32) point numbering
};

__device__ __forceinline__ void pack16(const unsigned (&e)[16], uint4& lo, uint4& hi) {
  lo = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
  hi = make_uint4(e[8] | (e[9] << 16), e[10] | (e[11] << 16), e[12] | (e[13] << 16), e[14] | (e[15] << 16));
}

// operand vector (lead, mid, bil): k 0..2 = pieces of `lead` (or ones), k 3..5 = pieces of `mid` (or ones),
// k 6..11 = the six-term pattern of `bil` in model or match order; unused entries are zero.
__device__ __forceinline__ void pack_t_vector(bool lead_ones, double lead, bool mid_ones, double mid, bool has_bil, double bil, bool model_order,
                                              uint4& lo, uint4& hi) {
  unsigned e[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) e[i] = 0;
  unsigned p[3];
  if (lead_ones) {
    e[0] = e[1] = e[2] = 0x3F80u;
  } else {
    split3_bf16(lead, p);
    e[0] = p[0]; e[1] = p[1]; e[2] = p[2];
  }
  if (mid_ones) {
    e[3] = e[4] = e[5] = 0x3F80u;
  } else {
    split3_bf16(mid, p);
    e[3] = p[0]; e[4] = p[1]; e[5] = p[2];
  }
  if (has_bil) {
    split3_bf16(bil, p);
    if (model_order) {
      e[6] = p[0]; e[7] = p[0]; e[8] = p[1]; e[9] = p[0]; e[10] = p[2]; e[11] = p[1];
    } else {
      e[6] = p[0]; e[7] = p[1]; e[8] = p[0]; e[9] = p[2]; e[10] = p[0]; e[11] = p[1];
    }
  }
  pack16(e, lo, hi);
}

// grid = (padded point blocks, cameras).  pieces[P][3][2]: the match-side vectors of P = dx - q, dy and M = dx + q;
// maxima[cam][4] = max |rx - u rz|, |ry - v rz|, |rz|, max(|u|, |v|) as float32 bit patterns (rounded up).
__global__ __launch_bounds__(kBlock) void t_prep_mfma_kernel(const double* __restrict__ pts_all, CamOffsets offs, CamPad pads, double thr,
                                                             uint4* __restrict__ pieces, unsigned* __restrict__ maxima) {
  const int cam = blockIdx.y;
  const int m = (int)(offs.off[cam + 1] - offs.off[cam]);
  const int mp = (int)(pads.poff[cam + 1] - pads.poff[cam]);
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned mx[4] = {0u, 0u, 0u, 0u};
  if (j < mp) {
    uint4* out = pieces + 6 * (pads.poff[cam] + j);
    if (j < m) {
      const double* p = pts_all + 6 * (offs.off[cam] + j);
      const double rx = p[0], ry = p[1], rz = p[2], u = p[3], v = p[4];
      const double a = fma(-u, rz, rx), b = fma(-v, rz, ry);
      pack_t_vector(false, a - thr * rz, true, 0.0, true, -u, false, out[0], out[1]);
      pack_t_vector(false, b, true, 0.0, true, -v, false, out[2], out[3]);
      pack_t_vector(false, a + thr * rz, true, 0.0, true, -u, false, out[4], out[5]);
      mx[0] = f32_up_bits(fabs(a));
      mx[1] = f32_up_bits(fabs(b));
      mx[2] = f32_up_bits(fabs(rz));
      mx[3] = f32_up_bits(fmax(fabs(u), fabs(v)));
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) out[i] = make_uint4(0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx[c] = max(mx[c], (unsigned)__shfl_xor((int)mx[c], o));  // non-negative floats order like their bits
    if ((threadIdx.x & 63) == 0 && mx[c]) atomicMax(maxima + 4 * cam + c, mx[c]);
  }
}

// grid = (hypothesis tiles, cameras, splits over the camera's tiles of 32 correspondences)
__global__ __launch_bounds__(kBlock, 2) void t_score_mfma_kernel(const double* __restrict__ pts_all, const uint4* __restrict__ pieces,
                                                                 const unsigned* __restrict__ maxima, CamOffsets offs, CamPad pads,
                                                                 const int32_t* __restrict__ idx, long long nb, double thr2,
                                                                 int* __restrict__ counts, int tiles_per_split, int use_atomics,
                                                                 int grid_tiles, int grid_cams, int grid_splits) {
  constexpr int kQueue = 256;  // pairs whose float32 sign is not trusted wait here for a dense float64 pass (per wave)
  __shared__ uint4 bstage[kBlock / 64][64][3][2];
  __shared__ double tstage[kBlock / 64][64][3];
  __shared__ unsigned qbuf[kBlock / 64][kQueue];
  __shared__ int qcnt[kBlock / 64][64];
  SFM_STAMP_ENTRY();
  __builtin_amdgcn_s_setprio(3);  // the solve runs ahead of the other waves' tile loops (see eh_upper_mfma_kernel)
  // grid_tiles == 0: a 3-D grid (hypothesis tile, camera, split).  Otherwise a 1-D grid of grid_tiles * grid_cams *
  // grid_splits workgroups (grid_splits a multiple of 8) laid out so that linear ids that are equal modulo 8, which the
  // dispatcher deals to the same XCD, work on the same eighth of the splits: every XCD's L2 then only ever holds its own
  // correspondence ranges (a speed matter only: any placement gives the same counts).
  int bx = blockIdx.x, cam = blockIdx.y, bz = blockIdx.z;
  if (grid_tiles) {
    const long long L = blockIdx.x;
    const int xcd = (int)(L & 7);
    const long long q = L >> 3;
    const int per = grid_splits >> 3;
    bz = (int)(q % per) * 8 + xcd;
    const long long r = q / per;
    bx = (int)(r % grid_tiles);
    cam = (int)(r / grid_tiles);
  }
  const double* __restrict__ pts = pts_all + 6 * offs.off[cam];
  const uint4* __restrict__ pc = pieces + 6 * pads.poff[cam];
  const int m = (int)(offs.off[cam + 1] - offs.off[cam]);
  const long long gid = (long long)bx * blockDim.x + threadIdx.x;
  const long long hyp = gid < nb ? gid : nb - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  float band_own;
  {
    const int32_t* pair = idx + 2 * ((long long)cam * nb + hyp);
    double T[3];
    {
      double x[2][3], r[2][3];
      load_t_sample(pts, m, pair, x, r);
      solve_T2<double>(x, r, pair[0] == pair[1] || same_correspondence(x, r), T);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) tstage[wave][lane][i] = T[i];
    const double tmax = fmax(fabs(T[0]), fmax(fabs(T[1]), fabs(T[2])));
    bool usable = tmax < 1.0e6;  // false for NaN / infinite / absurd translations: those are scored entirely in float64
    const double thr = sqrt(thr2);
    const double amax = __uint_as_float(maxima[4 * cam]), bmax = __uint_as_float(maxima[4 * cam + 1]);
    const double zmax = __uint_as_float(maxima[4 * cam + 2]), uvmax = __uint_as_float(maxima[4 * cam + 3]);
    const double Rx = amax + fabs(T[0]) + uvmax * fabs(T[2]), Ry = bmax + fabs(T[1]) + uvmax * fabs(T[2]);
    const double Rq = thr * (zmax + fabs(T[2]));
    // beyond the first limit the float32 products could overflow (false for NaN as well), below the second they could underflow
    usable = usable && Rx + Ry + Rq < 1.0e15 && fmax(Rx, Ry) + Rq > 1.0e-9;
    // every row sum of |a||x| of the three forms is below max(Rx, Ry) + Rq (the last factor: float64 rounding of the operands)
    const double e = kEpsMfma * (fmax(Rx, Ry) + Rq) * 1.000001;
    const double u = kEps32;
    band_own = usable ? round_up_f32(1.001 * (4.0 * Rq * e + 2.01 * u * Rq * Rq + 30.1 * e * e)) : __builtin_inff();
    const double tx = usable ? T[0] : 0.0, ty = usable ? T[1] : 0.0, tz = usable ? T[2] : 0.0;
    pack_t_vector(true, 0.0, false, tx - thr * tz, true, tz, true, bstage[wave][lane][0][0], bstage[wave][lane][0][1]);
    pack_t_vector(true, 0.0, false, ty, true, tz, true, bstage[wave][lane][1][0], bstage[wave][lane][1][1]);
    pack_t_vector(true, 0.0, false, tx + thr * tz, true, tz, true, bstage[wave][lane][2][0], bstage[wave][lane][2][1]);
  }
  __syncthreads();
  bf16x8 B[2][3];
  float band[2];
  unsigned force[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int v = 0; v < 3; ++v) B[k][v] = as_bf16x8(bstage[wave][32 * k + col][v][half]);
    band[k] = __shfl(band_own, 32 * k + col);
    force[k] = band[k] == __builtin_inff() ? 0xFFFFu : 0u;
  }
  const int apos = 16 * ((col >> 2) & 1) + (col & 3) + 4 * (col >> 3);  // see eh_upper_mfma_kernel
  const int ntiles = (m + 31) / 32;
  const int t0 = bz * tiles_per_split;
  const int t1 = min(ntiles, t0 + tiles_per_split);
  int cnt[2] = {0, 0};
  const f32x16 zero = {0};
  const unsigned tail_mask = tile_tail_mask(m, half);
  const uint4* src = pc + 6 * (t0 * 32 + apos) + half;
  // Pairs inside a band are not evaluated where they are found (a lane or two at a time with the rest of the wave idle, on a
  // quarter of all rows): their (correspondence, hypothesis) ids are appended to a queue of the wave, which is worked off 64
  // pairs at a time with the float64 test of t_score_kernel; the verdicts are added to per-hypothesis counters in LDS.
  qcnt[wave][lane] = 0;
  int qlen = 0;
  auto drain = [&]() {
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < qlen; base += 64) {
      if (base + lane < qlen) {
        const unsigned e = qbuf[wave][base + lane];
        const int h = (int)(e & 63u);
        const double* T = tstage[wave][h];
        const double* p = pts + 6LL * (e >> 6);
        if (reprojection_inlier<double>(T[0], T[1], T[2], p[0], p[1], p[2], p[3], p[4], thr2)) atomicAdd(&qcnt[wave][h], 1);
      }
    }
    qlen = 0;
    __builtin_amdgcn_wave_barrier();
  };
  // called by the whole wave with any = the lanes that have a pair to queue (not empty); bit 15-i of amb = correspondence first + i
  auto enqueue = [&](unsigned amb, unsigned long long any, int hyp_of_wave, int first) {
    do {
      const bool act = amb != 0u;
      const int bit = __ffs((int)amb) - 1;
      amb &= amb - 1u;
      const unsigned entry = ((unsigned)(first + 15 - bit) << 6) | (unsigned)hyp_of_wave;
      const int pos = qlen + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(any >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)any, 0u));
      if (act) qbuf[wave][pos] = entry;
      qlen += __popcll(any);
      if (qlen > kQueue - 64) drain();
      any = ballot64(amb != 0u);
    } while (any);
  };
  __builtin_amdgcn_s_setprio(0);
  SFM_STAMP_BEGIN();
  // The correspondence-side vectors are loaded one tile ahead: once the MFMAs of the second hypothesis group have been
  // issued their registers are free, and the next tile's loads run under that group's vector work.
  bf16x8 Ap, Ay, Am;
  auto load_tile = [&](const uint4* p) {
    Ap = as_bf16x8(p[0]);
    Ay = as_bf16x8(p[2]);
    Am = as_bf16x8(p[4]);
  };
  if (t0 < t1) load_tile(src);
  for (int t = t0; t < t1; ++t) {
    unsigned fill = t == (m >> 5) ? 0u : 0xFFFFu;  // scalar select + OR (a v_cndmask costs 18 issue cycles of the SIMD)
    asm volatile("" : "+s"(fill));
    const unsigned vmask = tail_mask | fill;
    if (t + 1 < t1) src += 192;  // the last tile loads itself again
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ap, B[k][0], zero, 0, 0, 0);
      const f32x16 dy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ay, B[k][1], zero, 0, 0, 0);
      const f32x16 dm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, B[k][2], zero, 0, 0, 0);
      if (k == 1) load_tile(src);
      unsigned hneg = 0;
      float closest = __builtin_inff();  // the smallest |d| of this lane's 16 pairs
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const float d0 = fma(dp[i], dm[i], dy[i] * dy[i]);  // < 0 <=> inlier
        const float d1 = fma(dp[i + 1], dm[i + 1], dy[i + 1] * dy[i + 1]);
        hneg = __builtin_amdgcn_alignbit(hneg, __float_as_uint(d0), 31);
        hneg = __builtin_amdgcn_alignbit(hneg, __float_as_uint(d1), 31);
        closest = min3_abs(closest, d0, d1);
      }
      cnt[k] += __popc(hneg & vmask);
      SFM_STAMP_COUNT(closest <= band[k]);
      if (ballot64(closest <= band[k])) {  // rare per row, not per wave: some pair of some lane's row is too close to the threshold (or is padding)
        unsigned hamb = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float d = fma(dp[i], dm[i], dy[i] * dy[i]);  // the same instructions give the same value again
          hamb = __builtin_amdgcn_alignbit(hamb, __float_as_uint(fabsf(d) - band[k]), 31);  // <= 0 would be exact; < 0 after the inflation of band
        }
        const unsigned amb = (hamb | force[k]) & vmask;  // an unusable hypothesis takes the float64 test everywhere
        cnt[k] -= __popc(hneg & amb);  // take the untrusted signs back out
        if (const unsigned long long any = ballot64(amb != 0u)) enqueue(amb, any, 32 * k + col, t * 32 + 16 * half);
      }
    }
  }
  drain();
  SFM_STAMP_END(t1 - t0);
  const int s0 = cnt[0] + __shfl_xor(cnt[0], 32), s1 = cnt[1] + __shfl_xor(cnt[1], 32);
  const int out = (half ? s1 : s0) + qcnt[wave][lane];  // lane L reports hypothesis L = 32*half + col
  if (gid < nb) {
    int* dst = counts + (long long)cam * nb + gid;
    if (use_atomics)
      atomicAdd(dst, out);
    else
      *dst = out;
  }
}

__global__ __launch_bounds__(kBlock) void pts_to_f32_kernel(const double* __restrict__ pts, long long m, float* __restrict__ pts32) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
#pragma unroll
  for (int i = 0; i < 5; ++i) pts32[8 * j + i] = (float)pts[6 * j + i];
#pragma unroll
  for (int i = 5; i < 8; ++i) pts32[8 * j + i] = 0.0f;
}

// inlier counts of given translations over one camera's points (one translation per lane)
__global__ __launch_bounds__(kBlock) void t_count_kernel(const double* __restrict__ pts, int m, const double* __restrict__ Ts, long long nb,
                                                         double thr2, int* __restrict__ counts) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long h = gid < nb ? gid : nb - 1;
  const double T0 = Ts[3 * h], T1 = Ts[3 * h + 1], T2 = Ts[3 * h + 2];
  int c = 0;
#pragma unroll 4
  for (int j = 0; j < m; ++j) {
    const double* p = pts + 6LL * j;
    c += reprojection_inlier<double>(T0, T1, T2, p[0], p[1], p[2], p[3], p[4], thr2) ? 1 : 0;
  }
  if (gid < nb) counts[gid] = c;
}

// first maximum of counts[cam][0..nb): key = (count << 32) | (0xFFFFFFFF - index), max over keys
__global__ __launch_bounds__(kBlock) void argmax_kernel(const int* __restrict__ counts, long long nb, unsigned long long index_base,
                                                        unsigned long long* __restrict__ best) {
  const int cam = blockIdx.y;
  unsigned long long key = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += (long long)gridDim.x * blockDim.x) {
    const unsigned long long k =
        ((unsigned long long)(unsigned)counts[(long long)cam * nb + i] << 32) | (0xFFFFFFFFull - (index_base + (unsigned long long)i));
    key = k > key ? k : key;
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const unsigned long long o = __shfl_xor(key, s, 64);
    key = o > key ? o : key;
  }
  __shared__ unsigned long long wave_key[kBlock / 64];
  if ((threadIdx.x & 63) == 0) wave_key[threadIdx.x >> 6] = key;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) key = wave_key[w] > key ? wave_key[w] : key;
    if (key) atomicMax(best + cam, key);  // one atomic per workgroup
  }
}

// ---------------------------------------------------------------------------------------
// Helpers of the bound cascade (sfm_eh_best): all tiny, all device-resident bookkeeping.
// ---------------------------------------------------------------------------------------
struct BestInfo {  // mirrored on the host, 8 x int64
  long long lead_e, lead_h, floor_e, floor_h, ncand_e, ncand_h, pad0, pad1;
};

// copies the index rows of the two hypotheses with the largest bounds into rows[2][12]
__global__ void gather_leaders_kernel(const unsigned long long* __restrict__ keys, const int32_t* __restrict__ idx, long long nb,
                                      int32_t* __restrict__ rows, BestInfo* __restrict__ info) {
  const int t = threadIdx.x;
  if (t < 24) {
    const int which = t / 12;
    long long id = (long long)(0xFFFFFFFFull - (keys[which] & 0xFFFFFFFFull));
    if (keys[which] == 0 || id >= nb) id = 0;
    rows[t] = idx[12 * id + (t % 12)];
    if (t % 12 == 0) {
      if (which == 0) info->lead_e = id; else info->lead_h = id;
    }
  }
}

// floors = exact counts of the leaders; counts the hypotheses whose bound reaches the floor and
// appends their ids (unordered) to cand_e / cand_h while they fit into `cap` entries each
__global__ __launch_bounds__(kBlock) void select_candidates_kernel(const int* __restrict__ up_e, const int* __restrict__ up_h, long long nb,
                                                                   const int* __restrict__ lead_counts /* ce[2] ch[2] */,
                                                                   BestInfo* __restrict__ info, int32_t* __restrict__ cand_e,
                                                                   int32_t* __restrict__ cand_h, int cap) {
  const int fe = lead_counts[0], fh = lead_counts[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    info->floor_e = fe;
    info->floor_h = fh;
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += (long long)gridDim.x * blockDim.x) {
    if (up_e[i] >= fe) {
      const unsigned long long slot = atomicAdd((unsigned long long*)&info->ncand_e, 1ull);
      if (slot < (unsigned long long)cap) cand_e[slot] = (int32_t)i;
    }
    if (up_h[i] >= fh) {
      const unsigned long long slot = atomicAdd((unsigned long long*)&info->ncand_h, 1ull);
      if (slot < (unsigned long long)cap) cand_h[slot] = (int32_t)i;
    }
  }
}

// rows_out[k][12] = idx[ids[k]][12]
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const int32_t* __restrict__ ids, int count, const int32_t* __restrict__ idx,
                                                             int32_t* __restrict__ rows_out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 12LL * count) rows_out[t] = idx[12LL * ids[t / 12] + (t % 12)];
}

// Speculative last level: while the host has not looked at the candidate counts yet, the first kSpecCap
// candidates of each class are already gathered (slots beyond the count repeat the class leader) ...
constexpr int kSpecCap = 256;

struct BestPending {  // what the host reads back in one copy: 8 x int64 + ids and exact counts of the speculative slots
  BestInfo info;
  int32_t ids[2 * kSpecCap];
  int32_t counts[2 * kSpecCap];
};

__global__ __launch_bounds__(kBlock) void gather_spec_rows_kernel(const BestInfo* __restrict__ info, const int32_t* __restrict__ cand_e,
                                                                  const int32_t* __restrict__ cand_h, const int32_t* __restrict__ idx,
                                                                  int32_t* __restrict__ rows, int32_t* __restrict__ ids_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * kSpecCap * 12) return;
  const int slot = t / 12, cls = slot / kSpecCap, j = slot % kSpecCap;
  const long long ncand = cls ? info->ncand_h : info->ncand_e;
  const bool live = j < ncand;
  const long long id = live ? (long long)(cls ? cand_h : cand_e)[j] : (cls ? info->lead_h : info->lead_e);
  rows[t] = idx[12 * id + (t % 12)];
  if (t % 12 == 0) ids_out[slot] = live ? (int32_t)id : -1;
}

// ... and scored exactly; this packs the outcome for the host
__global__ __launch_bounds__(kBlock) void pack_pending_kernel(const BestInfo* __restrict__ info, const int32_t* __restrict__ ids,
                                                              const int* __restrict__ sc_e, const int* __restrict__ sc_h,
                                                              BestPending* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) out->info = *info;
  if (t < 2 * kSpecCap) {
    out->ids[t] = ids[t];
    out->counts[t] = t < kSpecCap ? sc_e[t] : sc_h[t];
  }
}

// keeps the candidates whose tight bound still reaches the floor (in place, unordered)
__global__ __launch_bounds__(kBlock) void refilter_kernel(const int32_t* __restrict__ ids_in, const int* __restrict__ bound, int count, int floor,
                                                          int32_t* __restrict__ ids_out, unsigned long long* __restrict__ n_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count && bound[t] >= floor) ids_out[atomicAdd(n_out, 1ull)] = ids_in[t];
}

__global__ void t_winner_kernel(const double* __restrict__ pts_all, CamOffsets offs, const int32_t* __restrict__ idx, long long nb,
                                const unsigned long long* __restrict__ best, double* __restrict__ T_best) {
  const int cam = blockIdx.x;
  const unsigned long long key = best[cam];
  const long long i = (long long)(0xFFFFFFFFull - (key & 0xFFFFFFFFull));
  const long long h = (key == 0 || i >= nb) ? 0 : i;
  const double* pts = pts_all + 6 * offs.off[cam];
  const int m = (int)(offs.off[cam + 1] - offs.off[cam]);
  const int32_t* pair = idx + 2 * ((long long)cam * nb + h);
  double x[2][3], r[2][3], T[3];
  load_t_sample(pts, m, pair, x, r);
  solve_T2<double>(x, r, pair[0] == pair[1] || same_correspondence(x, r), T);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) T_best[3 * cam + k] = T[k];
  }
}

// ---------------------------------------------------------------------------------------
// Levenberg-Marquardt on T, one workgroup per camera, all iterations in one launch.
// ---------------------------------------------------------------------------------------
constexpr int kLmBlock = 1024;

template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* sh /* [NV][16] */) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v[k] += __shfl_xor(v[k], s, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) sh[k * 16 + wave] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[k * 16 + w];
    v[k] = t;  // every thread holds the same total, summed in the same order
  }
}

__global__ __launch_bounds__(kLmBlock) void lm_kernel(const double* __restrict__ Kg, const double* __restrict__ pts_all, CamOffsets offs,
                                                      const double* __restrict__ T0, int iters, double mu, double* __restrict__ work,
                                                      double* __restrict__ T_out, double* __restrict__ history) {
  __shared__ double sh[10 * 16];
  const int cam = blockIdx.x;
  const double* __restrict__ pts = pts_all + 6 * offs.off[cam];
  const int m = (int)(offs.off[cam + 1] - offs.off[cam]);
  double* __restrict__ r = work + 2 * offs.off[cam];  // [dx_0..dx_{m-1} | dy_0..dy_{m-1}]
  double K[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) K[i] = Kg[i];
  double T[3] = {T0[3 * cam], T0[3 * cam + 1], T0[3 * cam + 2]};
  for (int it = 0; it < iters; ++it) {
    // residuals of the current T (src/refine_translation.py:125-132)
    double err[1] = {0.0};
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
      const double* p = pts + 6LL * j;
      const double pz = p[2] + T[2];
      const double dx = (p[0] + T[0]) / pz - p[3];
      const double dy = (p[1] + T[1]) / pz - p[4];
      r[j] = dx;
      r[m + j] = dy;
      err[0] += dx * dx + dy * dy;
    }
    __syncthreads();  // r is read across threads below (same workgroup, global memory)
    __threadfence_block();
    // normal equations: C = J^T J (6 unique), c = J^T r with Jacobian row k paired with r[k]
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
      const double* p = pts + 6LL * j;
      const double Xc[3] = {p[0] + T[0], p[1] + T[1], p[2] + T[2]};
      double xp[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) xp[a] = (K[3 * a] * Xc[0] + K[3 * a + 1] * Xc[1]) + K[3 * a + 2] * Xc[2];
      const double inv = 1.0 / (xp[2] * xp[2]);
      double Ju[3], Jv[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        Ju[i] = (xp[2] * K[i] - xp[0] * K[6 + i]) * inv;
        Jv[i] = (xp[2] * K[3 + i] - xp[1] * K[6 + i]) * inv;
      }
      const double ru = r[2 * j], rv = r[2 * j + 1];
      acc[0] += Ju[0] * Ju[0] + Jv[0] * Jv[0];
      acc[1] += Ju[0] * Ju[1] + Jv[0] * Jv[1];
      acc[2] += Ju[0] * Ju[2] + Jv[0] * Jv[2];
      acc[3] += Ju[1] * Ju[1] + Jv[1] * Jv[1];
      acc[4] += Ju[1] * Ju[2] + Jv[1] * Jv[2];
      acc[5] += Ju[2] * Ju[2] + Jv[2] * Jv[2];
      acc[6] += Ju[0] * ru + Jv[0] * rv;
      acc[7] += Ju[1] * ru + Jv[1] * rv;
      acc[8] += Ju[2] * ru + Jv[2] * rv;
    }
    double tot[10];
#pragma unroll
    for (int i = 0; i < 9; ++i) tot[i] = acc[i];
    tot[9] = err[0];
    block_sum<10>(tot, sh);
    // delta = solve(-C, c) with C = J^T J + mu I: Gaussian elimination with partial pivoting
    double A[3][4] = {{-(tot[0] + mu), -tot[1], -tot[2], tot[6]},
                      {-tot[1], -(tot[3] + mu), -tot[4], tot[7]},
                      {-tot[2], -tot[4], -(tot[5] + mu), tot[8]}};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
      for (int i = k + 1; i < 3; ++i) {
        if (fabs(A[i][k]) > fabs(A[k][k])) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const double t = A[k][c];
            A[k][c] = A[i][c];
            A[i][c] = t;
          }
        }
      }
#pragma unroll
      for (int i = k + 1; i < 3; ++i) {
        const double f = A[i][k] / A[k][k];
#pragma unroll
        for (int c = k; c < 4; ++c) A[i][c] -= f * A[k][c];
      }
    }
    double d[3];
    d[2] = A[2][3] / A[2][2];
    d[1] = (A[1][3] - A[1][2] * d[2]) / A[1][1];
    d[0] = (A[0][3] - A[0][1] * d[1] - A[0][2] * d[2]) / A[0][0];
    const double Tn[3] = {T[0] + d[0], T[1] + d[1], T[2] + d[2]};
    double en[1] = {0.0};
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
      const double* p = pts + 6LL * j;
      const double pz = p[2] + Tn[2];
      const double dx = (p[0] + Tn[0]) / pz - p[3];
      const double dy = (p[1] + Tn[1]) / pz - p[4];
      en[0] += dx * dx + dy * dy;
    }
    block_sum<1>(en, sh);
    if (en[0] < tot[9]) {
      T[0] = Tn[0];
      T[1] = Tn[1];
      T[2] = Tn[2];
    }
    if (history && threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) history[((long long)cam * iters + it) * 3 + k] = T[k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) T_out[3 * cam + k] = T[k];
  }
}

// hipGetLastError() reports the last error of ANY earlier HIP call on this thread (other
// libraries in the process included), so every entry point clears it before it launches.
inline void clear_stale_error() { (void)hipGetLastError(); }
inline int check_launch() { return (int)hipGetLastError(); }

inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// Splits the point range when there are too few hypothesis tiles to fill 256 CUs.
inline int choose_splits(long long tiles, int npoints) {
  const long long want = 2048;
  if (tiles >= want) return 1;
  long long s = (want + tiles - 1) / tiles;
  const long long max_s = (npoints + 255) / 256;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 65535) s = 65535;
  return (int)s;
}

// workgroups of `kernel` that the current device holds at once (0 when the query fails)
template <typename K>
long long chip_slots(K kernel) {
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess) return 0;
  return (long long)cus * per_cu;
}

}  // namespace

#define SFM_TRY(expr)            \
  do {                           \
    int rc_ = (int)(expr);       \
    if (rc_ != 0) return rc_;    \
  } while (0)

extern "C" {

const char* sfm_hip_error_string(int code) {
  if (code == SFM_EINVAL) return "sfm_hip: invalid argument";
  if (code == SFM_ETOOLARGE) return "sfm_hip: more matches than the fixed scratch of the no-allocation entry point holds (SFM_NOALLOC_MAX_MATCHES)";
  return hipGetErrorString((hipError_t)code);
}

int sfm_hip_abi_version(void) { return 4; }

#ifdef SFM_CLOCK_STAMPS
// diagnostic build only: copies the first `slots` stamp records (8 x uint64 each)
int sfm_debug_clock_stamps(unsigned long long* host_out, int32_t slots, int32_t clear) {
  if (!host_out || slots <= 0 || slots > kStampSlots) return SFM_EINVAL;
  SFM_TRY(hipDeviceSynchronize());
  SFM_TRY(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8 * (size_t)slots));
  if (clear) {
    void* p = nullptr;
    SFM_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps)));
    SFM_TRY(hipMemset(p, 0, sizeof(unsigned long long) * 8 * (size_t)kStampSlots));
  }
  return 0;
}
#endif

static int launch_eh_score(const double* corr, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h, int32_t* count_e,
                           int32_t* count_h, hipStream_t st) {
  if (!corr || !idx || !count_e || !count_h || n <= 0 || nb < 0) return SFM_EINVAL;
  if (nb == 0) return 0;
  const long long tiles = (nb + kBlock - 1) / kBlock;
  if (tiles > 0x7FFFFFFFLL) return SFM_EINVAL;
  const int splits = choose_splits(tiles, n);
  const int pps = div_up(n, splits);
  if (splits > 1) {
    hipError_t e = hipMemsetAsync(count_e, 0, sizeof(int32_t) * nb, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(count_h, 0, sizeof(int32_t) * nb, st);
    if (e != hipSuccess) return (int)e;
  }
  // the reference squares the epipolar threshold (src/ransac_algorithm.py:102) but compares the transfer distance with
  // 3 * threshold itself (:327): a non-positive thr_h admits nothing
  eh_score_kernel<<<dim3((unsigned)tiles, splits), kBlock, 0, st>>>((const double4*)corr, n, idx, nb, 2.0 * thr_e * thr_e,
                                                                    thr_h > 0.0 ? thr_h * thr_h : -1.0, count_e, count_h, pps, splits > 1);
  return check_launch();
}

int sfm_eh_score_f64(const double* corr, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h, int32_t* count_e,
                     int32_t* count_h, void* stream) {
  clear_stale_error();
  return launch_eh_score(corr, n, idx, nb, thr_e, thr_h, count_e, count_h, (hipStream_t)stream);
}

int64_t sfm_corr_f32_bytes(int32_t n) { return n < 0 ? -1 : ((int64_t)n + 1) * 16; }

int sfm_corr_to_f32(const double* corr, int32_t n, float* corr32, void* stream) {
  clear_stale_error();
  if (!corr || !corr32 || n < 0 || ((uintptr_t)corr32 & 15u) != 0) return SFM_EINVAL;
  SFM_TRY(hipMemsetAsync(corr32 + 4 * (size_t)n, 0, 16, (hipStream_t)stream));  // trailer: measured max |coordinate|
  if (n == 0) return 0;
  corr_to_f32_kernel<<<div_up(n, kBlock), kBlock, 0, (hipStream_t)stream>>>((const double4*)corr, n, (float4*)corr32);
  return check_launch();
}

static int launch_eh_upper(const double* corr, const float* corr32, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h,
                           double xmax, double safety, int32_t one_sided, int32_t* up_e, int32_t* up_h, hipStream_t st) {
  if (!corr || !corr32 || !idx || !up_e || !up_h || n <= 0 || nb < 0) return SFM_EINVAL;
  if (!(xmax >= 0.0 && xmax < 1.0e300 && safety >= 1.0 && safety <= 1.0e3)) return SFM_EINVAL;
  if (nb == 0) return 0;
  const long long tiles = (nb + kBlock - 1) / kBlock;
  if (tiles > 0x7FFFFFFFLL) return SFM_EINVAL;
  const int splits = choose_splits(tiles, n);
  const int pps = div_up(n, splits);
  if (splits > 1) {
    hipError_t e = hipMemsetAsync(up_e, 0, sizeof(int32_t) * nb, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(up_h, 0, sizeof(int32_t) * nb, st);
    if (e != hipSuccess) return (int)e;
  }
  const double ke = 2.0 * thr_e * thr_e, kh = thr_h * thr_h;
  // Both forms give identical counts.  The scalar-load form is the default: on MI355X this kernel runs
  // into the power limit (fewer cycles are answered by a lower clock), and the form with fewer
  // instruction issues and SGPR-broadcast operands is the one that finishes first.
  static const bool lds_form = getenv("SFM_HIP_UPPER_LDS") && atoi(getenv("SFM_HIP_UPPER_LDS")) != 0;
  if (one_sided)
    eh_upper_kernel<true><<<dim3((unsigned)tiles, splits), kBlock, 0, st>>>((const double4*)corr, (const float4*)corr32, n, idx, nb, ke, kh, xmax,
                                                                            safety, up_e, up_h, pps, splits > 1);
  else if (lds_form)
    eh_upper_lds_kernel<1024><<<dim3((unsigned)tiles, splits), kBlock, 0, st>>>((const double4*)corr, (const float4*)corr32, n, idx, nb, ke, kh,
                                                                                xmax, safety, up_e, up_h, pps, splits > 1);
  else
    eh_upper_kernel<false><<<dim3((unsigned)tiles, splits), kBlock, 0, st>>>((const double4*)corr, (const float4*)corr32, n, idx, nb, ke, kh, xmax,
                                                                             safety, up_e, up_h, pps, splits > 1);
  return check_launch();
}

int sfm_eh_upper_f32(const double* corr, const float* corr32, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h,
                     double xmax, double safety, int32_t one_sided, int32_t* up_e, int32_t* up_h, void* stream) {
  clear_stale_error();
  return launch_eh_upper(corr, corr32, n, idx, nb, thr_e, thr_h, xmax, safety, one_sided, up_e, up_h, (hipStream_t)stream);
}

// match-side operands of the matrix-core pre-filter: [npad][32 B] piece vectors, [npad] float2, a 16-byte trailer, then
// the four 32-byte vectors per match of the epipolar forms (tile-major, see corr_to_upper2_kernel)
static inline int padded32(int n) { return (n + 31) & ~31; }

int64_t sfm_corr_mfma_bytes(int32_t n) { return n <= 0 ? 0 : (int64_t)padded32(n) * (40 + kUp2Vec * 32) + 16; }

static int launch_corr_to_mfma(const double* corr, int32_t n, double thr_e, void* corrm, hipStream_t st) {
  if (!corr || !corrm || n <= 0 || !(thr_e > 0.0) || ((uintptr_t)corrm & 15u) != 0) return SFM_EINVAL;
  const int npad = padded32(n);
  SFM_TRY(hipMemsetAsync((char*)corrm + (size_t)npad * 40, 0, 16, st));  // trailer: threshold tag + measured max |coordinate|
  corr_to_upper2_kernel<<<div_up(npad, kBlock), kBlock, 0, st>>>((const double4*)corr, n, npad, 1.0 / sqrt(uniform_k1(2.0 * thr_e * thr_e)),
                                                                 (uint4*)corrm, (float2*)((char*)corrm + (size_t)npad * 32),
                                                                 (uint4*)((char*)corrm + (size_t)npad * 40 + 16));
  return check_launch();
}

int sfm_corr_to_mfma(const double* corr, int32_t n, double thr_e, void* corrm, void* stream) {
  clear_stale_error();
  return launch_corr_to_mfma(corr, n, thr_e, corrm, (hipStream_t)stream);
}

static int launch_eh_upper_mfma(const double* corr, const void* corrm, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h,
                                double xmax, double safety, int32_t* up_e, int32_t* up_h, hipStream_t st, int part = 0,
                                bool cut_last_round = true) {
  if (!corr || !corrm || !idx || !up_e || !up_h || n <= 0 || nb < 0 || ((uintptr_t)corrm & 15u) != 0) return SFM_EINVAL;
  if (!(xmax >= 0.0 && xmax < 1.0e300 && safety >= 1.0 && safety <= 1.0e3)) return SFM_EINVAL;
  if (nb == 0) return 0;
  const long long tiles = (nb + kBlock - 1) / kBlock;
  if (tiles > 0x7FFFFFFFLL) return SFM_EINVAL;
  const int npad = padded32(n);
  const int ntiles = npad / 32;
  int splits = choose_splits(tiles, n);
  // Large match sets are also split so that the workgroups in flight stream the same 65536 matches (2.6 MB of operands):
  // they then stay in the 4 MB L2 of every XCD (3 % at 200 000 matches).
  static const int span = getenv("SFM_HIP_EH_SPAN") ? atoi(getenv("SFM_HIP_EH_SPAN")) : 65536;
  if (span > 0 && tiles > 1 && splits < div_up(n, span)) splits = div_up(n, span);
  if (splits > ntiles) splits = ntiles;
  const uint4* pieces = (const uint4*)corrm;
  const float2* xy = (const float2*)((const char*)corrm + (size_t)npad * 32);
  const uint4* evec = (const uint4*)((const char*)corrm + (size_t)npad * 40 + 16);
  // SFM_HIP_UPPER_FORM=0: the epipolar test from the rows E x1 on the vector ALU (eh_upper_mfma_kernel) instead of s and
  // k1 n1 + c whole from the matrix cores (eh_upper_mfma2_kernel)
  static const bool form2 = !(getenv("SFM_HIP_UPPER_FORM") && atoi(getenv("SFM_HIP_UPPER_FORM")) == 0);
  const double ke = 2.0 * thr_e * thr_e, kh = thr_h * thr_h;
  // clamped float counts (see eh_upper_mfma_kernel) for thresholds inside the range their exactness argument covers
  static const bool float_env = !(getenv("SFM_HIP_EH_FLOATCOUNT") && atoi(getenv("SFM_HIP_EH_FLOATCOUNT")) == 0);
  const double k1 = uniform_k1(ke);
  const bool float_count = float_env && k1 > 0x1.0p-45 && k1 < 0.6 && kh > 0x1.0p-45 && kh < 1.0e6;
  // one launch over the hypothesis tiles [t_lo, t_hi) with the match range cut into `sp` pieces; which = 0: both tests in
  // one launch, 1 / 2: the epipolar / the transfer test alone (second form: that launch solves and writes its own model only)
  auto launch = [&](long long t_lo, long long t_hi, int sp, int which) -> int {
    const long long off = t_lo * kBlock;
    const long long cnt = (t_hi * kBlock < nb ? t_hi * kBlock : nb) - off;
    const int tps = div_up(ntiles, sp);
    const bool own_only = form2 && which != 0;
    if (sp > 1) {
      if (!(own_only && which == 2)) SFM_TRY(hipMemsetAsync(up_e + off, 0, sizeof(int32_t) * cnt, st));
      if (!(own_only && which == 1)) SFM_TRY(hipMemsetAsync(up_h + off, 0, sizeof(int32_t) * cnt, st));
    }
    const dim3 grid((unsigned)(t_hi - t_lo), sp);
#define SFM_LAUNCH_UPPER(E_, H_, F_)                                                                                                          \
  do {                                                                                                                                        \
    if (form2)                                                                                                                                \
      eh_upper_mfma2_kernel<E_, H_, F_><<<grid, kBlock, 0, st>>>((const double4*)corr, pieces, xy, evec, n, idx + 12 * off, cnt, ke, kh, xmax,  \
                                                                 safety, up_e + off, up_h + off, tps, sp > 1);                                \
    else                                                                                                                                      \
      eh_upper_mfma_kernel<E_, H_, F_><<<grid, kBlock, 0, st>>>((const double4*)corr, pieces, xy, n, idx + 12 * off, cnt, ke, kh, xmax, safety, \
                                                                up_e + off, up_h + off, tps, sp > 1);                                         \
  } while (0)
    if (which == 1) {
      if (float_count) SFM_LAUNCH_UPPER(true, false, true); else SFM_LAUNCH_UPPER(true, false, false);
    } else if (which == 2) {
      if (float_count) SFM_LAUNCH_UPPER(false, true, true); else SFM_LAUNCH_UPPER(false, true, false);
    } else {
      if (float_count) SFM_LAUNCH_UPPER(true, true, true); else SFM_LAUNCH_UPPER(true, true, false);
    }
#undef SFM_LAUNCH_UPPER
    return check_launch();
  };
  // All workgroups do the same work, so a grid runs in rounds of as many workgroups as the chip holds at once, and a last
  // round that is only a few percent full still takes a round.  When that happens the tiles of the last round are
  // launched apart with their match range cut into pieces: as many workgroups as a full round, each a fraction of the work
  // (2-3 % for one call at a time; a caller that keeps several blocks in flight on several streams fills the last round
  // with the next block's work anyway and is better off without the extra launch: cut_last_round = false).
  static const long long slots_both = form2 ? chip_slots(eh_upper_mfma2_kernel<true, true, true>) : chip_slots(eh_upper_mfma_kernel<true, true, true>);
  static const long long slots_e = form2 ? chip_slots(eh_upper_mfma2_kernel<true, false, true>) : chip_slots(eh_upper_mfma_kernel<true, false, true>);
  static const long long slots_h = form2 ? chip_slots(eh_upper_mfma2_kernel<false, true, true>) : chip_slots(eh_upper_mfma_kernel<false, true, true>);
  static const bool cut_tail = !(getenv("SFM_HIP_EH_TAIL") && atoi(getenv("SFM_HIP_EH_TAIL")) == 0);
  auto launch_all = [&](int which) -> int {
    const long long slots = which == 1 ? slots_e : which == 2 ? slots_h : slots_both;
    const long long rem = slots > 0 ? tiles % slots : 0;
    if (cut_tail && cut_last_round && splits == 1 && tiles > slots && rem > 0 && rem * 4 <= slots) {
      int sp = (int)(slots / rem);
      if (sp > 16) sp = 16;  // every piece solves its hypotheses again: keep that below the work it saves
      if (sp > ntiles) sp = ntiles;
      SFM_TRY(launch(0, tiles - rem, 1, which));
      return launch(tiles - rem, tiles, sp, which);
    }
    return launch(0, tiles, splits, which);
  };
  if (part != 0) {  // measurement only: one test alone; the other array receives zeros
    SFM_TRY(hipMemsetAsync(part == 1 ? up_h : up_e, 0, sizeof(int32_t) * nb, st));
    return launch_all(part);
  }
  // Second form: one launch per test from 2^28 pairs on.  Each then fits four waves per SIMD (the fused instantiation: three,
  // with four piece vectors waiting in LDS), solves only its own model and reads the index rows once more; below that size
  // the second launch costs more than it gains.
  static const int split_env = getenv("SFM_HIP_UPPER_SPLIT") ? atoi(getenv("SFM_HIP_UPPER_SPLIT")) : -1;
  const bool split_tests = form2 && (split_env >= 0 ? split_env != 0 : (long long)nb * n >= (1LL << 28));
  if (split_tests) {
    SFM_TRY(launch_all(1));
    return launch_all(2);
  }
  return launch_all(0);
}

int sfm_eh_upper_mfma_f32(const double* corr, const void* corrm, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h,
                          double xmax, double safety, int32_t* up_e, int32_t* up_h, void* stream) {
  clear_stale_error();
  return launch_eh_upper_mfma(corr, corrm, n, idx, nb, thr_e, thr_h, xmax, safety, up_e, up_h, (hipStream_t)stream);
}

int sfm_eh_upper_mfma_part_f32(const double* corr, const void* corrm, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h,
                               double xmax, double safety, int32_t part, int32_t* up_e, int32_t* up_h, void* stream) {
  clear_stale_error();
  if (part != 1 && part != 2) return SFM_EINVAL;
  return launch_eh_upper_mfma(corr, corrm, n, idx, nb, thr_e, thr_h, xmax, safety, up_e, up_h, (hipStream_t)stream, part);
}

// bytes of match-side operands per match: six 32-byte piece vectors + (x2, y2) as float32
constexpr int kEhxBytesPerMatch = kEhxVecA * 32 + 8;

int64_t sfm_eh_score_mixed_workspace_bytes(int64_t nb, int32_t n) {
  if (nb < 0 || n <= 0) return -1;
  return ((int64_t)padded32(n) * kEhxBytesPerMatch + 16 + 255) / 256 * 256;
}

// workspace: sfm_eh_score_mixed_workspace_bytes(nb, n) bytes of device memory, 256-byte aligned (checked by the callers)
static int run_eh_score_mixed(const double* corr, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h, double xmax,
                              void* workspace, int32_t* count_e, int32_t* count_h, hipStream_t st, int32_t options = 0) {
  if (!corr || !idx || !count_e || !count_h || !workspace || n <= 0 || nb < 0) return SFM_EINVAL;
  if (!(xmax >= 0.0 && xmax < 1.0e300)) return SFM_EINVAL;
  if (nb == 0) return 0;
  // without positive thresholds there is no band to speak of: the float64 kernel answers (all counts zero for thr_h <= 0)
  if (!(thr_e > 0.0 && thr_h > 0.0)) return launch_eh_score(corr, n, idx, nb, thr_e, thr_h, count_e, count_h, st);
  const long long tiles = (nb + kBlock - 1) / kBlock;
  if (tiles > 0x7FFFFFFFLL) return SFM_EINVAL;
  const int npad = padded32(n);
  const int ntiles = npad / 32;
  // SFM_HIP_EHX_FORM=0: the epipolar test from the rows E x1 and E^T x2 on the vector ALU (eh_score_mfma_kernel, 72 bytes of
  // operands per match) instead of s, n1, n2 whole from the matrix cores (eh_score_mfma2_kernel, 264 bytes per match)
  static const bool form2_env = !(getenv("SFM_HIP_EHX_FORM") && atoi(getenv("SFM_HIP_EHX_FORM")) == 0);
  const bool form2 = form2_env && n < (1 << 25);  // its queue entries keep the match index in 25 bits
  uint4* pieces = (uint4*)workspace;
  float2* xy = (float2*)((char*)workspace + (size_t)npad * (form2 ? kEhxVecA * 32 : 64));
  int splits = choose_splits(tiles, n);
  // the span of matches that the workgroups in flight share in L2 (as in launch_eh_upper_mfma)
  static const int span_env = getenv("SFM_HIP_EHX_SPAN") ? atoi(getenv("SFM_HIP_EHX_SPAN")) : -1;
  const int span = span_env >= 0 ? span_env : (form2 ? 12288 : 32768);
  if (span > 0 && tiles > 1 && splits < div_up(n, span)) splits = div_up(n, span);
  if (splits > ntiles) splits = ntiles;
  const int tps = div_up(ntiles, splits);
  if (splits > 1) {
    SFM_TRY(hipMemsetAsync(count_e, 0, sizeof(int32_t) * nb, st));
    SFM_TRY(hipMemsetAsync(count_h, 0, sizeof(int32_t) * nb, st));
  }
  SFM_TRY(hipMemsetAsync(xy + npad, 0, 16, st));  // trailer: measured max |coordinate|
  if (form2)
    corr_to_mfma3_kernel<<<div_up(npad, kBlock), kBlock, 0, st>>>((const double4*)corr, n, npad, pieces, xy);
  else
    corr_to_mfma2_kernel<<<div_up(npad, kBlock), kBlock, 0, st>>>((const double4*)corr, n, npad, pieces, xy);
  SFM_TRY(check_launch());
  // two launches (one model each, at a higher occupancy) pay from about 2^28 pairs on; below that the second launch costs more
  static const int split_env = getenv("SFM_HIP_EHX_SPLIT") ? atoi(getenv("SFM_HIP_EHX_SPLIT")) : -1;
  const bool split_models = split_env >= 0 ? split_env != 0 : (long long)nb * n >= (1LL << 28);
  const dim3 grid((unsigned)tiles, splits);
  // SFM_HIP_EHX_PART=1 / 2: a timing aid, only the E / only the H launch of the two-launch form (the other count array is not written)
  static const int part = getenv("SFM_HIP_EHX_PART") ? atoi(getenv("SFM_HIP_EHX_PART")) : 0;
  if (form2) {
    // always one launch per model (at 3 and 4 waves per SIMD; a fused instantiation would need 256 registers and spill)
    // As in launch_eh_upper_mfma: a last round of the grid that is less than a quarter full is launched apart with its
    // match range cut into pieces (the E launch holds 768 workgroups at a time: 3907 of them are 5.09 rounds).
    // SFM_HIP_EHX_EARLY=0: every pair through the full float32 tests (no early loop)
    static const int early_default = !(getenv("SFM_HIP_EHX_EARLY") && atoi(getenv("SFM_HIP_EHX_EARLY")) == 0);
    const int early_env = (early_default && !(options & SFM_EHX_EVERY_PAIR_FULL_TEST)) ? 1 : 0;
    static const bool cut_tail = !(getenv("SFM_HIP_EHX_TAIL") && atoi(getenv("SFM_HIP_EHX_TAIL")) == 0);
    static const long long slots_e = chip_slots(eh_score_mfma2_kernel<true, false, 3, false>), slots_h = chip_slots(eh_score_mfma2_kernel<false, true, 4, false>);
    auto launch_model = [&](bool model_e, long long slots) -> int {
      auto one = [&](long long t_lo, long long t_hi, int sp, bool zeroed) -> int {
        const long long off = t_lo * kBlock;
        const long long cnt = (t_hi * kBlock < nb ? t_hi * kBlock : nb) - off;
        int32_t* dst = (model_e ? count_e : count_h) + off;
        if (sp > 1 && !zeroed) SFM_TRY(hipMemsetAsync(dst, 0, sizeof(int32_t) * cnt, st));
        const dim3 g((unsigned)(t_hi - t_lo), sp);
        const int tp = div_up(ntiles, sp);
#define SFM_LAUNCH_EHX(E_, H_, W_, EARLY_)                                                                                                  \
  eh_score_mfma2_kernel<E_, H_, W_, EARLY_><<<g, kBlock, 0, st>>>((const double4*)corr, pieces, xy, n, idx + 12 * off, cnt, 2.0 * thr_e * thr_e, \
                                                                  thr_h * thr_h, xmax, count_e + off, count_h + off, tp, sp > 1)
        if (model_e) {
          if (early_env) SFM_LAUNCH_EHX(true, false, 3, true); else SFM_LAUNCH_EHX(true, false, 3, false);
        } else {
          if (early_env) SFM_LAUNCH_EHX(false, true, 4, true); else SFM_LAUNCH_EHX(false, true, 4, false);
        }
#undef SFM_LAUNCH_EHX
        return check_launch();
      };
      const long long rem = slots > 0 ? tiles % slots : 0;
      if (cut_tail && splits == 1 && tiles > slots && rem > 0 && rem * 4 <= slots) {
        int sp = (int)(slots / rem);
        if (sp > 16) sp = 16;  // every piece solves its hypotheses again: keep that below the work it saves
        if (sp > ntiles) sp = ntiles;
        SFM_TRY(one(0, tiles - rem, 1, false));
        return one(tiles - rem, tiles, sp, false);
      }
      return one(0, tiles, splits, true);  // with splits > 1 both count arrays were zeroed above
    };
    if (part != 2) SFM_TRY(launch_model(true, slots_e));
    if (part != 1) SFM_TRY(launch_model(false, slots_h));
    return 0;
  }
  if (split_models) {
    if (part != 2)
      eh_score_mfma_kernel<true, false, 3><<<grid, kBlock, 0, st>>>((const double4*)corr, pieces, xy, n, idx, nb, 2.0 * thr_e * thr_e, thr_h * thr_h,
                                                                   xmax, count_e, count_h, tps, splits > 1);
    SFM_TRY(check_launch());
    if (part != 1)
      eh_score_mfma_kernel<false, true, 4><<<grid, kBlock, 0, st>>>((const double4*)corr, pieces, xy, n, idx, nb, 2.0 * thr_e * thr_e, thr_h * thr_h,
                                                                 xmax, count_e, count_h, tps, splits > 1);
  } else {
    eh_score_mfma_kernel<true, true, 2><<<grid, kBlock, 0, st>>>((const double4*)corr, pieces, xy, n, idx, nb, 2.0 * thr_e * thr_e, thr_h * thr_h,
                                                                xmax, count_e, count_h, tps, splits > 1);
  }
  return check_launch();
}

int sfm_eh_score_mixed_opt(const double* corr, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h, double xmax,
                           int32_t options, void* workspace, int64_t workspace_bytes, int32_t* count_e, int32_t* count_h, void* stream) {
  clear_stale_error();
  if (!workspace || n <= 0 || nb < 0 || ((uintptr_t)workspace & 255u) != 0 || (options & ~SFM_EHX_EVERY_PAIR_FULL_TEST) != 0) return SFM_EINVAL;
  if (workspace_bytes < sfm_eh_score_mixed_workspace_bytes(nb, n)) return SFM_EINVAL;
  return run_eh_score_mixed(corr, n, idx, nb, thr_e, thr_h, xmax, workspace, count_e, count_h, (hipStream_t)stream, options);
}

int sfm_eh_score_mixed(const double* corr, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h, double xmax, void* workspace,
                       int64_t workspace_bytes, int32_t* count_e, int32_t* count_h, void* stream) {
  return sfm_eh_score_mixed_opt(corr, n, idx, nb, thr_e, thr_h, xmax, 0, workspace, workspace_bytes, count_e, count_h, stream);
}

// ---- the same scoring for callers that cannot allocate ---------------------------------------
// The match-side operands live in a file-scope device array instead of a caller workspace.  Each GPU holds its own
// copy of the array at its own address: the address is looked up once per device (with that device current) and kept;
// host code only ever uses that looked-up pointer, never the array's name.
static constexpr size_t kNoallocBytes = ((size_t)SFM_NOALLOC_MAX_MATCHES * kEhxBytesPerMatch + 16 + 255) / 256 * 256;
static __device__ __align__(256) unsigned char scratch[kNoallocBytes];
namespace {
constexpr int kMaxDevices = 64;
void* g_noalloc_ptr[kMaxDevices] = {};
std::mutex g_noalloc_mutex;

// The array of the device that owns `stream` (a NULL stream belongs to the current device).  The first call per device
// looks the address up with that device current (and restores the caller's device); later calls only read the table.
int noalloc_scratch(hipStream_t stream, void** out) {
  int cur = -1, dev = -1;
  SFM_TRY(hipGetDevice(&cur));
  dev = cur;
  if (stream) SFM_TRY(hipStreamGetDevice(stream, &dev));
  if (dev < 0 || dev >= kMaxDevices) return SFM_EINVAL;
  std::lock_guard<std::mutex> lock(g_noalloc_mutex);
  if (!g_noalloc_ptr[dev]) {
    void* p = nullptr;
    if (dev != cur) SFM_TRY(hipSetDevice(dev));
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (dev != cur) SFM_TRY(hipSetDevice(cur));
    if (e != hipSuccess) return (int)e;
    if (!p) return SFM_EINVAL;
    g_noalloc_ptr[dev] = p;
  }
  *out = g_noalloc_ptr[dev];
  return 0;
}
}  // namespace

int sfm_eh_score_noalloc(const double* corr, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h, int32_t* count_e,
                         int32_t* count_h, void* stream) {
  clear_stale_error();
  if (n <= 0 || nb < 0) return SFM_EINVAL;
  if (n > SFM_NOALLOC_MAX_MATCHES) return SFM_ETOOLARGE;
  void* ws = nullptr;
  SFM_TRY(noalloc_scratch((hipStream_t)stream, &ws));
  // run_eh_score_mixed rewrites every operand of the n (padded to 32) matches and zeroes the 16-byte trailer that the
  // measurement of max |coordinate| accumulates into, so nothing of an earlier call is read
  return run_eh_score_mixed(corr, n, idx, nb, thr_e, thr_h, 0.0, ws, count_e, count_h, (hipStream_t)stream);
}

// ---- the whole bound cascade as one call -------------------------------------------------
namespace {
constexpr int kCandCap = 65536;      // candidates kept per class; more than that -> score everything in float64
constexpr int kTightenAbove = 4096;  // more candidates than this are first re-bounded with the tight float32 tests

struct BestWs {
  void* corrm;
  int32_t* spec_ids;
  BestPending* pending;
  int *up_e, *up_h, *lead_counts, *sc_e, *sc_h;
  unsigned long long *keys, *n2;
  BestInfo* info;
  int32_t *lead_rows, *cand_e, *cand_h, *cand2, *rows;
};

inline size_t carve(char* base, int64_t nb, int32_t n, BestWs* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += (bytes + 255) & ~(size_t)255;
    return p;
  };
  w->up_e = (int*)take(sizeof(int) * nb);
  w->up_h = (int*)take(sizeof(int) * nb);
  w->keys = (unsigned long long*)take(sizeof(unsigned long long) * 2);
  w->n2 = (unsigned long long*)take(sizeof(unsigned long long) * 2);
  w->info = (BestInfo*)take(sizeof(BestInfo));
  w->lead_rows = (int32_t*)take(sizeof(int32_t) * 24);
  w->lead_counts = (int*)take(sizeof(int) * 4);
  w->cand_e = (int32_t*)take(sizeof(int32_t) * kCandCap);
  w->cand_h = (int32_t*)take(sizeof(int32_t) * kCandCap);
  w->cand2 = (int32_t*)take(sizeof(int32_t) * 2 * kCandCap);
  w->rows = (int32_t*)take(sizeof(int32_t) * 12 * 2 * kCandCap);
  w->sc_e = (int*)take(sizeof(int) * 2 * kCandCap);
  w->sc_h = (int*)take(sizeof(int) * 2 * kCandCap);
  w->corrm = take((size_t)sfm_corr_mfma_bytes(n));
  w->spec_ids = (int32_t*)take(sizeof(int32_t) * 2 * kSpecCap);
  w->pending = (BestPending*)take(sizeof(BestPending));
  return off;
}
}  // namespace

int64_t sfm_eh_best_workspace_bytes(int64_t nb, int32_t n) {
  BestWs w;
  return (nb < 0 || n < 0) ? -1 : (int64_t)carve(nullptr, nb, n, &w);
}


int64_t sfm_eh_best_pending_bytes(void) { return (int64_t)sizeof(BestPending); }

static int eh_best_submit(const double* corr, const float* corr32, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h,
                          double xmax, double safety, int32_t loose, void* workspace, int64_t workspace_bytes, void* host_pending,
                          void* stream, bool blocks_in_flight) {
  clear_stale_error();
  if (!corr || !corr32 || !idx || !workspace || !host_pending || n <= 0 || nb <= 0 || (uint64_t)nb > 0x7FFFFFFFull) return SFM_EINVAL;
  BestWs w;
  if (((uintptr_t)workspace & 255u) != 0 || (int64_t)carve((char*)workspace, nb, n, &w) > workspace_bytes) return SFM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // level 1: float32 bounds for every hypothesis, leaders, their exact counts, candidate lists
  if (loose == 2) {
    SFM_TRY(launch_corr_to_mfma(corr, n, thr_e, w.corrm, st));
    SFM_TRY(launch_eh_upper_mfma(corr, w.corrm, n, idx, nb, thr_e, thr_h, xmax, safety, w.up_e, w.up_h, st, 0, !blocks_in_flight));
  } else {
    SFM_TRY(launch_eh_upper(corr, corr32, n, idx, nb, thr_e, thr_h, xmax, safety, loose, w.up_e, w.up_h, st));
  }
  SFM_TRY(hipMemsetAsync(w.keys, 0, 256 * 3, st));  // keys, n2, info (three consecutive 256-byte slots)
  const long long tiles = (nb + kBlock - 1) / kBlock;
  const int ablocks = (int)(tiles < 256 ? tiles : 256);
  argmax_kernel<<<dim3(ablocks, 1), kBlock, 0, st>>>(w.up_e, nb, 0ull, w.keys);
  argmax_kernel<<<dim3(ablocks, 1), kBlock, 0, st>>>(w.up_h, nb, 0ull, w.keys + 1);
  gather_leaders_kernel<<<1, 64, 0, st>>>(w.keys, idx, nb, w.lead_rows, w.info);
  SFM_TRY(check_launch());
  SFM_TRY(launch_eh_score(corr, n, w.lead_rows, 2, thr_e, thr_h, w.lead_counts, w.lead_counts + 2, st));
  select_candidates_kernel<<<ablocks, kBlock, 0, st>>>(w.up_e, w.up_h, nb, w.lead_counts, w.info, w.cand_e, w.cand_h, kCandCap);
  // speculative last level: the first kSpecCap candidates of each class are scored exactly right away, so that in
  // the usual case (a handful of candidates) the host needs one look at the outcome and no further launch
  gather_spec_rows_kernel<<<div_up(2 * kSpecCap * 12, kBlock), kBlock, 0, st>>>(w.info, w.cand_e, w.cand_h, idx, w.rows, w.spec_ids);
  SFM_TRY(check_launch());
  SFM_TRY(launch_eh_score(corr, n, w.rows, 2 * kSpecCap, thr_e, thr_h, w.sc_e, w.sc_h, st));
  pack_pending_kernel<<<div_up(2 * kSpecCap, kBlock), kBlock, 0, st>>>(w.info, w.spec_ids, w.sc_e, w.sc_h, w.pending);
  SFM_TRY(check_launch());
  SFM_TRY(hipMemcpyAsync(host_pending, w.pending, sizeof(BestPending), hipMemcpyDeviceToHost, st));
  return 0;
}

int sfm_eh_best_submit(const double* corr, const float* corr32, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h,
                       double xmax, double safety, int32_t loose, void* workspace, int64_t workspace_bytes, void* host_pending,
                       void* stream) {
  return eh_best_submit(corr, corr32, n, idx, nb, thr_e, thr_h, xmax, safety, loose, workspace, workspace_bytes, host_pending, stream, true);
}

int sfm_eh_best_collect(const double* corr, const float* corr32, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h,
                        double xmax, double safety, int32_t loose, void* workspace, int64_t workspace_bytes, const void* host_pending,
                        int64_t* result, void* stream) {
  clear_stale_error();
  if (!corr || !corr32 || !idx || !workspace || !host_pending || !result || n <= 0 || nb <= 0 || (uint64_t)nb > 0x7FFFFFFFull)
    return SFM_EINVAL;
  BestWs w;
  if (((uintptr_t)workspace & 255u) != 0 || (int64_t)carve((char*)workspace, nb, n, &w) > workspace_bytes) return SFM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < 8; ++i) result[i] = 0;
  SFM_TRY(hipStreamSynchronize(st));
  const BestPending* pend = (const BestPending*)host_pending;
  BestInfo info = pend->info;
  result[4] = info.ncand_e;
  result[5] = info.ncand_h;
  if (info.ncand_e <= kSpecCap && info.ncand_h <= kSpecCap) {
    // every candidate was already scored exactly: first maximum = largest count, lowest id
    for (int cls = 0; cls < 2; ++cls) {
      const int cnt_n = (int)(cls ? info.ncand_h : info.ncand_e);
      long long bc = -1, bi = -1;
      for (int k = 0; k < cnt_n; ++k) {
        const long long c = pend->counts[cls * kSpecCap + k], id = pend->ids[cls * kSpecCap + k];
        if (c > bc || (c == bc && id < bi)) {
          bc = c;
          bi = id;
        }
      }
      if (bi < 0) {  // cannot happen (the leader always qualifies); keep the leader's exact count
        bc = cls ? info.floor_h : info.floor_e;
        bi = cls ? info.lead_h : info.lead_e;
      }
      result[2 * cls] = bc;
      result[2 * cls + 1] = bi;
    }
    result[7] = info.ncand_e + info.ncand_h;
    return 0;
  }
  result[0] = info.floor_e;
  result[1] = info.lead_e;
  result[2] = info.floor_h;
  result[3] = info.lead_h;
  if (loose && (info.ncand_e > kCandCap || info.ncand_h > kCandCap)) {
    // too many survivors of the loose bounds for the candidate lists (e.g. a scene whose best count is
    // tiny): apply the tight bounds to every hypothesis and select again with the same floors
    SFM_TRY(launch_eh_upper(corr, corr32, n, idx, nb, thr_e, thr_h, xmax, safety, 0, w.up_e, w.up_h, st));
    const long long tiles = (nb + kBlock - 1) / kBlock;
    const int ablocks = (int)(tiles < 256 ? tiles : 256);
    SFM_TRY(hipMemsetAsync(&w.info->ncand_e, 0, 2 * sizeof(long long), st));
    select_candidates_kernel<<<ablocks, kBlock, 0, st>>>(w.up_e, w.up_h, nb, w.lead_counts, w.info, w.cand_e, w.cand_h, kCandCap);
    SFM_TRY(check_launch());
    SFM_TRY(hipMemcpyAsync(&info, w.info, sizeof(info), hipMemcpyDeviceToHost, st));
    SFM_TRY(hipStreamSynchronize(st));
    result[6] = nb;
    loose = 0;  // the lists now hold survivors of the tight bounds
  }
  if (info.ncand_e > kCandCap || info.ncand_h > kCandCap) {
    // (near-)universal ties: no bound can help, score everything exactly
    SFM_TRY(launch_eh_score(corr, n, idx, nb, thr_e, thr_h, w.up_e, w.up_h, st));
    SFM_TRY(hipMemsetAsync(w.keys, 0, 256, st));
    const long long tiles = (nb + kBlock - 1) / kBlock;
    const int ablocks = (int)(tiles < 256 ? tiles : 256);
    argmax_kernel<<<dim3(ablocks, 1), kBlock, 0, st>>>(w.up_e, nb, 0ull, w.keys);
    argmax_kernel<<<dim3(ablocks, 1), kBlock, 0, st>>>(w.up_h, nb, 0ull, w.keys + 1);
    SFM_TRY(check_launch());
    unsigned long long keys[2];
    SFM_TRY(hipMemcpyAsync(keys, w.keys, sizeof(keys), hipMemcpyDeviceToHost, st));
    SFM_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < 2; ++k) {
      result[2 * k] = (int64_t)(keys[k] >> 32);
      result[2 * k + 1] = (int64_t)(0xFFFFFFFFull - (keys[k] & 0xFFFFFFFFull));
    }
    result[7] = nb;
    return 0;
  }
  int ne = (int)info.ncand_e, nh = (int)info.ncand_h;
  int32_t* list_e = w.cand_e;
  int32_t* list_h = w.cand_h;
  if (loose && ne + nh > kTightenAbove) {
    // level 2: tight float32 bounds on the candidates only
    gather_rows_kernel<<<div_up(12LL * ne, kBlock), kBlock, 0, st>>>(w.cand_e, ne, idx, w.rows);
    gather_rows_kernel<<<div_up(12LL * nh, kBlock), kBlock, 0, st>>>(w.cand_h, nh, idx, w.rows + 12LL * ne);
    SFM_TRY(check_launch());
    SFM_TRY(launch_eh_upper(corr, corr32, n, w.rows, ne + nh, thr_e, thr_h, xmax, safety, 0, w.sc_e, w.sc_h, st));
    SFM_TRY(hipMemsetAsync(w.n2, 0, 2 * sizeof(unsigned long long), st));
    refilter_kernel<<<div_up(ne, kBlock), kBlock, 0, st>>>(w.cand_e, w.sc_e, ne, (int)info.floor_e, w.cand2, w.n2);
    refilter_kernel<<<div_up(nh, kBlock), kBlock, 0, st>>>(w.cand_h, w.sc_h + ne, nh, (int)info.floor_h, w.cand2 + kCandCap, w.n2 + 1);
    SFM_TRY(check_launch());
    unsigned long long n2[2];
    SFM_TRY(hipMemcpyAsync(n2, w.n2, sizeof(n2), hipMemcpyDeviceToHost, st));
    SFM_TRY(hipStreamSynchronize(st));
    result[6] = ne + nh;
    ne = (int)n2[0];
    nh = (int)n2[1];
    list_e = w.cand2;
    list_h = w.cand2 + kCandCap;
  }
  // level 3: exact float64 counts of what is left; first maximum = largest count, lowest id
  gather_rows_kernel<<<div_up(12LL * ne, kBlock), kBlock, 0, st>>>(list_e, ne, idx, w.rows);
  gather_rows_kernel<<<div_up(12LL * nh, kBlock), kBlock, 0, st>>>(list_h, nh, idx, w.rows + 12LL * ne);
  SFM_TRY(check_launch());
  SFM_TRY(launch_eh_score(corr, n, w.rows, ne + nh, thr_e, thr_h, w.sc_e, w.sc_h, st));
  static thread_local int32_t* host_buf = nullptr;
  static thread_local size_t host_cap = 0;
  const size_t need = (size_t)(ne + nh) * 2;
  if (host_cap < need) {
    free(host_buf);
    host_buf = (int32_t*)malloc(sizeof(int32_t) * need);
    host_cap = host_buf ? need : 0;
    if (!host_buf) return SFM_EINVAL;
  }
  int32_t* ids = host_buf;
  int32_t* cnt = host_buf + (ne + nh);
  SFM_TRY(hipMemcpyAsync(ids, list_e, sizeof(int32_t) * ne, hipMemcpyDeviceToHost, st));
  SFM_TRY(hipMemcpyAsync(ids + ne, list_h, sizeof(int32_t) * nh, hipMemcpyDeviceToHost, st));
  SFM_TRY(hipMemcpyAsync(cnt, w.sc_e, sizeof(int32_t) * ne, hipMemcpyDeviceToHost, st));
  SFM_TRY(hipMemcpyAsync(cnt + ne, w.sc_h + ne, sizeof(int32_t) * nh, hipMemcpyDeviceToHost, st));
  SFM_TRY(hipStreamSynchronize(st));
  for (int cls = 0; cls < 2; ++cls) {
    const int lo = cls ? ne : 0, hi = cls ? ne + nh : ne;
    long long bc = -1, bi = -1;
    for (int k = lo; k < hi; ++k)
      if (cnt[k] > bc || (cnt[k] == bc && ids[k] < bi)) {
        bc = cnt[k];
        bi = ids[k];
      }
    if (bi >= 0) {
      result[2 * cls] = bc;
      result[2 * cls + 1] = bi;
    }
  }
  result[7] = ne + nh;
  return 0;
}

int sfm_eh_best(const double* corr, const float* corr32, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h, double xmax,
                double safety, int32_t loose, void* workspace, int64_t workspace_bytes, int64_t* result, void* stream) {
  static thread_local BestPending pending;  // pageable: the copy into it completes before hipMemcpyAsync returns or at the sync below
  SFM_TRY(eh_best_submit(corr, corr32, n, idx, nb, thr_e, thr_h, xmax, safety, loose, workspace, workspace_bytes, &pending, stream, false));
  return sfm_eh_best_collect(corr, corr32, n, idx, nb, thr_e, thr_h, xmax, safety, loose, workspace, workspace_bytes, &pending, result,
                             stream);
}

int sfm_eh_models_f64(const double* corr, int32_t n, const int32_t* idx, int64_t nb, double* F_out, double* E_out, double* H_out,
                      void* stream) {
  clear_stale_error();
  if (!corr || !idx || n <= 0 || nb < 0) return SFM_EINVAL;
  if (nb == 0) return 0;
  eh_models_kernel<<<div_up(nb, kBlock), kBlock, 0, (hipStream_t)stream>>>((const double4*)corr, n, idx, nb, F_out, E_out, H_out);
  return check_launch();
}

int sfm_dlt_general_f64(const double* corr, int32_t n, int32_t homography, double* out, void* stream) {
  clear_stale_error();
  if (!corr || !out || n < (homography ? 4 : 8)) return SFM_EINVAL;
  dlt_general_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const double4*)corr, n, homography ? 1 : 0, out);
  return check_launch();
}

static int mask_common(bool epi, const double* M, const double* corr, int32_t n, double thr, uint8_t* mask, int32_t* count, void* stream) {
  clear_stale_error();
  if (!M || !corr || !mask || n < 0) return SFM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (count) {
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
  }
  if (n == 0) return 0;
  if (epi)
    mask_kernel<true><<<div_up(n, kBlock), kBlock, 0, st>>>(M, (const double4*)corr, n, 2.0 * thr * thr, mask, count);
  else
    mask_kernel<false><<<div_up(n, kBlock), kBlock, 0, st>>>(M, (const double4*)corr, n, thr > 0.0 ? thr * thr : -1.0, mask, count);
  return check_launch();
}

int sfm_epipolar_mask_f64(const double* E, const double* corr, int32_t n, double thr, uint8_t* mask, int32_t* count, void* stream) {
  clear_stale_error();
  return mask_common(true, E, corr, n, thr, mask, count, stream);
}

int sfm_transfer_mask_f64(const double* H, const double* corr, int32_t n, double thr, uint8_t* mask, int32_t* count, void* stream) {
  clear_stale_error();
  return mask_common(false, H, corr, n, thr, mask, count, stream);
}

int sfm_epipolar_errors_f64(const double* F, const double* x1, const double* x2, int32_t n, int64_t ld, double* d, void* stream) {
  clear_stale_error();
  if (!F || !x1 || !x2 || !d || n < 0 || ld < n) return SFM_EINVAL;
  if (n == 0) return 0;
  epipolar_errors_kernel<<<div_up(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(F, x1, x2, n, ld, d);
  return check_launch();
}

int sfm_enforce_essential_f64(const double* F, double* E, void* stream) {
  clear_stale_error();
  if (!F || !E) return SFM_EINVAL;
  enforce_essential_kernel<<<1, 64, 0, (hipStream_t)stream>>>(F, E);
  return check_launch();
}

int sfm_essential_candidates_f64(const double* E, double* P2s, void* stream) {
  clear_stale_error();
  if (!E || !P2s) return SFM_EINVAL;
  essential_candidates_kernel<<<1, 64, 0, (hipStream_t)stream>>>(E, P2s);
  return check_launch();
}

int sfm_chirality4_f64(const double* P2s, const double* corr, const uint8_t* mask, int32_t n, int32_t* counts, void* stream) {
  clear_stale_error();
  if (!P2s || !corr || !counts || n < 0) return SFM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(counts, 0, 4 * sizeof(int32_t), st);
  if (e != hipSuccess) return (int)e;
  if (n == 0) return 0;
  chirality4_kernel<<<div_up(n, kBlock), kBlock, 0, st>>>(P2s, (const double4*)corr, mask, n, counts);
  return check_launch();
}

int sfm_homography_to_rt_f64(const double* H, const double* corr, int32_t n, double* out, void* stream) {
  clear_stale_error();
  if (!H || !corr || !out || n < 0) return SFM_EINVAL;
  homography_to_rt_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(H, (const double4*)corr, n, out);
  return check_launch();
}

// ---- record events of the E/H replay: one launch and one read-back each ------------------------
int64_t sfm_record_scratch_doubles(void) { return kRecDoubles; }

static int record_row(const int32_t* row, int32_t n, RowArg* r) {
  for (int i = 0; i < 12; ++i) {
    if (row[i] < 0 || row[i] >= n) return SFM_EINVAL;
    r->v[i] = row[i];
  }
  return 0;
}

int sfm_e_record_f64(const double* corr, int32_t n, const int32_t* row, double thr, uint8_t* mask, double* scratch, double* host_out,
                     void* stream) {
  clear_stale_error();
  if (!corr || !row || !mask || !scratch || !host_out || n <= 0) return SFM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  RowArg r;
  SFM_TRY(record_row(row, n, &r));
  SFM_TRY(hipMemsetAsync(scratch + kRecCnt, 0, 4 * sizeof(double), st));
  e_record_kernel<<<div_up(n, kBlock), kBlock, 0, st>>>((const double4*)corr, n, r, 2.0 * thr * thr, mask, scratch);
  SFM_TRY(check_launch());
  SFM_TRY(hipMemcpyAsync(host_out, scratch, sizeof(double) * 70, hipMemcpyDeviceToHost, st));
  return (int)hipStreamSynchronize(st);
}

int sfm_h_record_f64(const double* corr, int32_t n, const int32_t* row, double thr, uint8_t* mask_a, uint8_t* mask_b, double* scratch,
                     double* host_out, void* stream) {
  clear_stale_error();
  if (!corr || !row || !mask_a || !mask_b || !scratch || !host_out || n <= 0) return SFM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  RowArg r;
  SFM_TRY(record_row(row, n, &r));
  SFM_TRY(hipMemsetAsync(scratch + kRecCnt, 0, 4 * sizeof(double), st));
  SFM_TRY(hipMemsetAsync(scratch + kRecCnt2, 0, 4 * sizeof(double), st));
  h_record_kernel<<<div_up(n, kBlock), kBlock, 0, st>>>((const double4*)corr, n, r, nullptr, 2.0 * thr * thr, mask_a, mask_b, scratch);
  SFM_TRY(check_launch());
  SFM_TRY(hipMemcpyAsync(host_out, scratch, sizeof(double) * 165, hipMemcpyDeviceToHost, st));
  return (int)hipStreamSynchronize(st);
}

int sfm_h_record_batch_f64(const double* corr, int32_t n, const int32_t* rows, int32_t nev, double thr, uint8_t* mask_a, uint8_t* mask_b,
                           double* scratch, double* host_out, void* stream) {
  clear_stale_error();
  if (!corr || !rows || !mask_a || !mask_b || !scratch || !host_out || n <= 0 || nev < 0 || nev > 65535) return SFM_EINVAL;
  if (nev == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  SFM_TRY(hipMemsetAsync(scratch, 0, sizeof(double) * kRecDoubles * nev, st));
  h_record_kernel<<<dim3(div_up(n, kBlock), nev), kBlock, 0, st>>>((const double4*)corr, n, RowArg{}, rows, 2.0 * thr * thr, mask_a, mask_b,
                                                                  scratch);
  SFM_TRY(check_launch());
  SFM_TRY(hipMemcpyAsync(host_out, scratch, sizeof(double) * kRecDoubles * nev, hipMemcpyDeviceToHost, st));
  return (int)hipStreamSynchronize(st);
}

int sfm_triangulate_f64(const double* P1, const double* P2, const double* x1, const double* x2, int32_t m, int64_t ld, double* X,
                        void* stream) {
  clear_stale_error();
  if (!P1 || !P2 || !x1 || !x2 || !X || m < 0 || ld < m) return SFM_EINVAL;
  if (m == 0) return 0;
  triangulate_kernel<<<div_up(m, kBlock), kBlock, 0, (hipStream_t)stream>>>(P1, P2, x1, x2, m, ld, X);
  return check_launch();
}

int sfm_pack_points_f64(const double* R, const void* X, int32_t x_is_f64, const double* x, int32_t m, double* pts, void* stream) {
  clear_stale_error();
  if (!R || !X || !x || !pts || m < 0) return SFM_EINVAL;
  if (m == 0) return 0;
  if (x_is_f64)
    pack_points_kernel<double><<<div_up(m, kBlock), kBlock, 0, (hipStream_t)stream>>>(R, (const double*)X, x, m, pts);
  else
    pack_points_kernel<float><<<div_up(m, kBlock), kBlock, 0, (hipStream_t)stream>>>(R, (const float*)X, x, m, pts);
  return check_launch();
}

static int fill_offsets(const int64_t* cam_off, int32_t ncam, CamOffsets* o) {
  if (!cam_off || ncam <= 0 || ncam > kMaxCams) return SFM_EINVAL;
  for (int i = 0; i <= ncam; ++i) {
    o->off[i] = cam_off[i];
    if (i && cam_off[i] < cam_off[i - 1]) return SFM_EINVAL;
    if (i && cam_off[i] - cam_off[i - 1] > 0x7FFFFFFFLL) return SFM_EINVAL;
  }
  return 0;
}

int sfm_t_ransac_f64(const double* pts, const int64_t* cam_off, int32_t ncam, const int32_t* idx, int64_t nb, double thr, int32_t* counts,
                     uint64_t* best, double* T_best, void* stream) {
  clear_stale_error();
  if (!pts || !idx || !counts || !best || !T_best || nb <= 0) return SFM_EINVAL;
  CamOffsets offs;
  int rc = fill_offsets(cam_off, ncam, &offs);
  if (rc) return rc;
  int mmax = 0;
  for (int c = 0; c < ncam; ++c) {
    const int m = (int)(offs.off[c + 1] - offs.off[c]);
    if (m <= 0) return SFM_EINVAL;  // an empty camera has no sample to draw
    mmax = m > mmax ? m : mmax;
  }
  hipStream_t st = (hipStream_t)stream;
  const long long tiles = (nb + kBlock - 1) / kBlock;
  if (tiles > 0x7FFFFFFFLL) return SFM_EINVAL;
  const int splits = choose_splits(tiles * ncam, mmax);
  const int pps = div_up(mmax, splits);
  hipError_t e;
  if (splits > 1) {
    e = hipMemsetAsync(counts, 0, sizeof(int32_t) * nb * ncam, st);
    if (e != hipSuccess) return (int)e;
  }
  e = hipMemsetAsync(best, 0, sizeof(uint64_t) * ncam, st);
  if (e != hipSuccess) return (int)e;
  // count_inliers compares a norm with the threshold itself (src/estimate_translation.py:127): a non-positive one admits nothing
  t_score_kernel<<<dim3((unsigned)tiles, ncam, splits), kBlock, 0, st>>>(pts, offs, idx, nb, thr > 0.0 ? thr * thr : -1.0, counts, pps,
                                                                         splits > 1);
  rc = check_launch();
  if (rc) return rc;
  const int ablocks = (int)(tiles < 256 ? tiles : 256);
  argmax_kernel<<<dim3(ablocks, ncam), kBlock, 0, st>>>(counts, nb, 0ull, (unsigned long long*)best);
  rc = check_launch();
  if (rc) return rc;
  t_winner_kernel<<<ncam, 64, 0, st>>>(pts, offs, idx, nb, (const unsigned long long*)best, T_best);
  return check_launch();
}

static int fill_pads(const CamOffsets& offs, int32_t ncam, CamPad* pads) {
  pads->poff[0] = 0;
  for (int c = 0; c < ncam; ++c) {
    const int64_t m = offs.off[c + 1] - offs.off[c];
    pads->poff[c + 1] = pads->poff[c] + ((m + 31) & ~(int64_t)31);
  }
  return 0;
}

int64_t sfm_t_ransac_mixed_workspace_bytes(const int64_t* cam_off, int32_t ncam) {
  CamOffsets offs;
  if (fill_offsets(cam_off, ncam, &offs)) return -1;
  CamPad pads;
  fill_pads(offs, ncam, &pads);
  return 1024 + pads.poff[ncam] * 96;
}

int sfm_t_ransac_mixed(const double* pts, const int64_t* cam_off, int32_t ncam, const int32_t* idx, int64_t nb, double thr, void* workspace,
                       int64_t workspace_bytes, int32_t* counts, uint64_t* best, double* T_best, void* stream) {
  clear_stale_error();
  if (!pts || !idx || !counts || !best || !T_best || !workspace || nb <= 0) return SFM_EINVAL;
  if (((uintptr_t)workspace & 255u) != 0) return SFM_EINVAL;
  if (!(thr > 0.0)) return sfm_t_ransac_f64(pts, cam_off, ncam, idx, nb, thr, counts, best, T_best, stream);  // nothing to band: all counts zero
  CamOffsets offs;
  int rc = fill_offsets(cam_off, ncam, &offs);
  if (rc) return rc;
  CamPad pads;
  fill_pads(offs, ncam, &pads);
  if (workspace_bytes < 1024 + pads.poff[ncam] * 96) return SFM_EINVAL;  // 1024-byte header: 4 maxima per camera
  int mmax = 0;
  for (int c = 0; c < ncam; ++c) {
    const int m = (int)(offs.off[c + 1] - offs.off[c]);
    if (m <= 0) return SFM_EINVAL;
    mmax = m > mmax ? m : mmax;
  }
  hipStream_t st = (hipStream_t)stream;
  unsigned* maxima = (unsigned*)workspace;
  uint4* pieces = (uint4*)((char*)workspace + 1024);
  const long long tiles = (nb + kBlock - 1) / kBlock;
  if (tiles > 0x7FFFFFFFLL) return SFM_EINVAL;
  const int ptiles = (mmax + 31) / 32;
  int splits = choose_splits(tiles * ncam, mmax);
  // The correspondence range is also split so that the workgroups in flight (grid order: hypothesis tile, camera, split)
  // stream the same 8192 correspondences (768 KB of operands) of a few cameras: the working set then stays in the 4 MB L2
  // of every XCD instead of cycling a camera's 96 bytes per correspondence through it once per hypothesis tile.
  static const int span = getenv("SFM_HIP_T_SPAN") ? atoi(getenv("SFM_HIP_T_SPAN")) : 8192;
  if (span > 0 && tiles > 1 && splits < div_up(mmax, span)) splits = div_up(mmax, span);
  if (splits > ptiles) splits = ptiles;
  // with 8 or more splits: a multiple of 8 of them and the XCD-aware 1-D grid (see the kernel)
  static const bool xcd_grid = !(getenv("SFM_HIP_T_XCD") && atoi(getenv("SFM_HIP_T_XCD")) == 0);
  const bool xcd = xcd_grid && splits >= 5 && ptiles >= 8 && tiles * ncam * ((splits + 7) / 8 * 8) <= 0x7FFFFFFFLL;
  if (xcd) splits = (splits + 7) / 8 * 8 <= ptiles ? (splits + 7) / 8 * 8 : 8;
  const int tps = div_up(ptiles, splits);
  SFM_TRY(hipMemsetAsync(maxima, 0, 1024, st));
  if (splits > 1) SFM_TRY(hipMemsetAsync(counts, 0, sizeof(int32_t) * nb * ncam, st));
  SFM_TRY(hipMemsetAsync(best, 0, sizeof(uint64_t) * ncam, st));
  t_prep_mfma_kernel<<<dim3(div_up(ptiles * 32, kBlock), ncam), kBlock, 0, st>>>(pts, offs, pads, thr, pieces, maxima);
  SFM_TRY(check_launch());
  if (xcd)
    t_score_mfma_kernel<<<dim3((unsigned)(tiles * ncam * splits)), kBlock, 0, st>>>(pts, pieces, maxima, offs, pads, idx, nb, thr * thr, counts,
                                                                                    tps, 1, (int)tiles, ncam, splits);
  else
    t_score_mfma_kernel<<<dim3((unsigned)tiles, ncam, splits), kBlock, 0, st>>>(pts, pieces, maxima, offs, pads, idx, nb, thr * thr, counts, tps,
                                                                                splits > 1, 0, 0, 0);
  SFM_TRY(check_launch());
  const int ablocks = (int)(tiles < 256 ? tiles : 256);
  argmax_kernel<<<dim3(ablocks, ncam), kBlock, 0, st>>>(counts, nb, 0ull, (unsigned long long*)best);
  SFM_TRY(check_launch());
  t_winner_kernel<<<ncam, 64, 0, st>>>(pts, offs, idx, nb, (const unsigned long long*)best, T_best);
  return check_launch();
}

int sfm_t_count_f64(const double* pts, int32_t m, const double* T, int64_t nb, double thr, int32_t* counts, void* stream) {
  clear_stale_error();
  if (!pts || !T || !counts || m < 0 || nb < 0) return SFM_EINVAL;
  if (nb == 0) return 0;
  t_count_kernel<<<div_up(nb, kBlock), kBlock, 0, (hipStream_t)stream>>>(pts, m, T, nb, thr > 0.0 ? thr * thr : -1.0, counts);
  return check_launch();
}

int sfm_argmax_count_i32(const int32_t* counts, int64_t nb, uint64_t index_base, uint64_t* best, void* stream) {
  clear_stale_error();
  if (!counts || !best || nb < 0 || index_base + (uint64_t)nb > 0xFFFFFFFFull) return SFM_EINVAL;
  if (nb == 0) return 0;
  const long long tiles = (nb + kBlock - 1) / kBlock;
  const int ablocks = (int)(tiles < 1024 ? tiles : 1024);
  argmax_kernel<<<dim3(ablocks, 1), kBlock, 0, (hipStream_t)stream>>>(counts, nb, index_base, (unsigned long long*)best);
  return check_launch();
}

int sfm_pts_to_f32(const double* pts, int64_t m, float* pts32, void* stream) {
  clear_stale_error();
  if (!pts || !pts32 || m < 0) return SFM_EINVAL;
  if (m == 0) return 0;
  pts_to_f32_kernel<<<div_up(m, kBlock), kBlock, 0, (hipStream_t)stream>>>(pts, m, pts32);
  return check_launch();
}

int sfm_t_upper_f32(const double* pts, const float* pts32, const int64_t* cam_off, int32_t ncam, const int32_t* idx, int64_t nb, double thr,
                    const double* rmax, double xmax, double safety, int32_t* up, void* stream) {
  clear_stale_error();
  if (!pts || !pts32 || !idx || !up || !rmax || nb <= 0) return SFM_EINVAL;
  if (!(xmax >= 0.0 && xmax < 1.0e300 && safety >= 1.0 && safety <= 1.0e3)) return SFM_EINVAL;
  CamOffsets offs;
  int rc = fill_offsets(cam_off, ncam, &offs);
  if (rc) return rc;
  CamBounds bounds;
  int mmax = 0;
  for (int c = 0; c < ncam; ++c) {
    const int m = (int)(offs.off[c + 1] - offs.off[c]);
    if (m <= 0 || !(rmax[c] >= 0.0 && rmax[c] < 1.0e12)) return SFM_EINVAL;
    bounds.rmax[c] = rmax[c];
    mmax = m > mmax ? m : mmax;
  }
  hipStream_t st = (hipStream_t)stream;
  const long long tiles = (nb + kBlock - 1) / kBlock;
  if (tiles > 0x7FFFFFFFLL) return SFM_EINVAL;
  const int splits = choose_splits(tiles * ncam, mmax);
  const int pps = div_up(mmax, splits);
  if (splits > 1) {
    hipError_t e = hipMemsetAsync(up, 0, sizeof(int32_t) * nb * ncam, st);
    if (e != hipSuccess) return (int)e;
  }
  t_upper_kernel<<<dim3((unsigned)tiles, ncam, splits), kBlock, 0, st>>>(pts, pts32, offs, bounds, idx, nb, thr * thr, xmax, safety, up, pps,
                                                                         splits > 1);
  return check_launch();
}

int sfm_argmax_count_batched_i32(const int32_t* counts, int64_t nb, int32_t nrows, uint64_t* best, void* stream) {
  clear_stale_error();
  if (!counts || !best || nb <= 0 || nrows <= 0 || nrows > 65535 || (uint64_t)nb > 0xFFFFFFFFull) return SFM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(best, 0, sizeof(uint64_t) * nrows, st);
  if (e != hipSuccess) return (int)e;
  const long long tiles = (nb + kBlock - 1) / kBlock;
  const int ablocks = (int)(tiles < 256 ? tiles : 256);
  argmax_kernel<<<dim3(ablocks, nrows), kBlock, 0, st>>>(counts, nb, 0ull, (unsigned long long*)best);
  return check_launch();
}

int sfm_t_models_f64(const double* pts, int32_t m, const int32_t* idx, int64_t nb, double* T_out, void* stream) {
  clear_stale_error();
  if (!pts || !idx || !T_out || m <= 0 || nb < 0) return SFM_EINVAL;
  if (nb == 0) return 0;
  t_models_kernel<<<div_up(nb, kBlock), kBlock, 0, (hipStream_t)stream>>>(pts, m, idx, nb, T_out);
  return check_launch();
}

int sfm_lm_translation_f64(const double* K, const double* pts, const int64_t* cam_off, int32_t ncam, const double* T0, int32_t iters,
                           double mu, double* work, double* T_out, double* history, void* stream) {
  clear_stale_error();
  if (!K || !pts || !T0 || !work || !T_out || iters < 0) return SFM_EINVAL;
  CamOffsets offs;
  int rc = fill_offsets(cam_off, ncam, &offs);
  if (rc) return rc;
  lm_kernel<<<ncam, kLmBlock, 0, (hipStream_t)stream>>>(K, pts, offs, T0, iters, mu, work, T_out, history);
  return check_launch();
}

int sfm_index_sets_i32(uint64_t seed, uint64_t first_id, int64_t nb, int32_t n, int32_t width, int32_t* idx, void* stream) {
  clear_stale_error();
  if (nb < 0 || n <= 0 || (width != 12 && width != 2)) return SFM_EINVAL;
  if (nb == 0) return 0;
  if (!idx || ((uintptr_t)idx & (width == 12 ? 15u : 7u)) != 0) return SFM_EINVAL;
  const long long tiles = (nb + kBlock - 1) / kBlock;
  if (tiles > 0x7FFFFFFFLL) return SFM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (width == 12)
    index_sets_kernel<12><<<(unsigned)tiles, kBlock, 0, st>>>(seed, first_id, nb, (unsigned)n, idx);
  else
    index_sets_kernel<2><<<(unsigned)tiles, kBlock, 0, st>>>(seed, first_id, nb, (unsigned)n, idx);
  return check_launch();
}

}  // extern "C"
