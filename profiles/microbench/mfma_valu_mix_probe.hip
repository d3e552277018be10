// The loop shape of the scoring kernels -- M x v_mfma_f32_32x32x16_bf16, then N float32 vector instructions that read the
// accumulators those MFMAs wrote -- at W waves per SIMD, measured without a tail: every wave runs until a common
// deadline (s_memrealtime) and counts its iterations; one workgroup of 4 W waves per CU keeps the W waves of a SIMD
// resident together.  ENC 2: the vector instructions are v_fmac_f32 (VOP2, 4 bytes); ENC 3: v_fma_f32 with a separate
// destination (VOP3, 8 bytes).  Printed: shader cycles per iteration per SIMD beside the issue bound 2 N + 8 M, and per wave.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_mix_probe mfma_valu_mix_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
struct Rec { unsigned long long iters, cycles, ticks; };

template <int M, int N, int ENC>
__global__ __launch_bounds__(1024) void k(Rec* out, const uint4* __restrict__ in, unsigned long long ticks) {
  const uint4 ua = in[threadIdx.x & 255], ub = in[256 + (threadIdx.x & 255)];
  const bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  float x[16], y[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { x[i] = __uint_as_float(ua.x) * (i + 1); y[i] = x[i] * 0.5f; }
  const float c0 = __uint_as_float(ub.y);
  constexpr int MA = M > 0 ? M : 1;
  f32x16 acc[MA];
  const f32x16 zero = {0};
#pragma unroll
  for (int m = 0; m < MA; ++m) acc[m] = zero;
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long deadline = r0 + ticks;
  unsigned long long iters = 0, r1 = r0;
  while (r1 < deadline) {
#pragma unroll 1
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, zero, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float av = M > 0 ? acc[j % MA][(j / MA) % 16] : y[j % 16];
        if (ENC == 2)
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[j % 16]) : "v"(av), "v"(c0));
        else
          asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[j % 16]) : "v"(av), "v"(c0), "v"(x[(j + 8) % 16]));
      }
    }
    iters += 4;
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += x[i];
  if (r == 12345.678f) out[0].iters = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = Rec{iters, t1 - t0, r1 - r0};
}

// The same measurement with the MFMAs in three accumulation chains of C0, C1 and C2 instructions (each chain adds into one
// accumulator: a K = 16 C contraction), followed by N v_fma_f32 that read the three accumulators: the shape of a test whose
// bilinear and quadratic forms come whole from the matrix cores.
template <int C0, int C1, int C2, int N>
__global__ __launch_bounds__(1024) void kchain(Rec* out, const uint4* __restrict__ in, unsigned long long ticks) {
  const uint4 ua = in[threadIdx.x & 255], ub = in[256 + (threadIdx.x & 255)];
  // every MFMA gets operand registers of its own (equal values, but the compiler cannot know: equal operands would let it
  // merge the chains)
  bf16x8 a[3][4], b[3][4];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      a[c][m] = __builtin_bit_cast(bf16x8, ua);
      b[c][m] = __builtin_bit_cast(bf16x8, ub);
      asm volatile("" : "+v"(a[c][m]), "+v"(b[c][m]));
    }
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = __uint_as_float(ua.x) * (i + 1);
  const float c0 = __uint_as_float(ub.y);
  const f32x16 zero = {0};
  f32x16 acc[3] = {zero, zero, zero};
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long deadline = r0 + ticks;
  unsigned long long iters = 0, r1 = r0;
  constexpr int CH[3] = {C0, C1, C2};
  while (r1 < deadline) {
#pragma unroll 1
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int m = 0; m < CH[c]; ++m) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c][m], b[c][m], m == 0 ? zero : acc[c], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float av = acc[j % 3][(j / 3) % 16];
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[j % 16]) : "v"(av), "v"(c0), "v"(x[(j + 8) % 16]));
      }
    }
    iters += 4;
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += x[i];
  if (r == 12345.678f) out[0].iters = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = Rec{iters, t1 - t0, r1 - r0};
}

// The chained shape software-pipelined inside a wave: the MFMAs of iteration i + 1 are issued before the vector
// instructions of iteration i, into a second set of accumulators (48 more registers: two waves per SIMD at most).
template <int C0, int C1, int C2, int N, int LB>
__global__ __launch_bounds__(LB) void kchain_pipelined(Rec* out, const uint4* __restrict__ in, unsigned long long ticks) {
  const uint4 ua = in[threadIdx.x & 255], ub = in[256 + (threadIdx.x & 255)];
  bf16x8 a[3][4], b[3][4];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      a[c][m] = __builtin_bit_cast(bf16x8, ua);
      b[c][m] = __builtin_bit_cast(bf16x8, ub);
      asm volatile("" : "+v"(a[c][m]), "+v"(b[c][m]));
    }
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = __uint_as_float(ua.x) * (i + 1);
  const float c0 = __uint_as_float(ub.y);
  const f32x16 zero = {0};
  f32x16 cur[3] = {zero, zero, zero}, nxt[3];
  constexpr int CH[3] = {C0, C1, C2};
  auto issue = [&](f32x16 (&acc)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int m = 0; m < CH[c]; ++m) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c][m], b[c][m], m == 0 ? zero : acc[c], 0, 0, 0);
    }
  };
  auto consume = [&](const f32x16 (&acc)[3]) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float av = acc[j % 3][(j / 3) % 16];
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[j % 16]) : "v"(av), "v"(c0), "v"(x[(j + 8) % 16]));
    }
  };
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long deadline = r0 + ticks;
  unsigned long long iters = 0, r1 = r0;
  issue(cur);
  while (r1 < deadline) {
#pragma unroll 1
    for (int rep = 0; rep < 2; ++rep) {
      issue(nxt);
      consume(cur);
      issue(cur);
      consume(nxt);
    }
    iters += 4;
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += x[i] + cur[0][i];
  if (r == 12345.678f) out[0].iters = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = Rec{iters, t1 - t0, r1 - r0};
}

template <int C0, int C1, int C2, int N, int LB = 512>
int run_chain_pipelined(int w, const uint4* in, Rec* out) {
  const int blocks = 256, threads = 256 * w, waves = blocks * threads / 64;
  if (threads > LB) return 0;  // the register budget of this instantiation does not hold that many waves
  kchain_pipelined<C0, C1, C2, N, LB><<<blocks, threads>>>(out, in, 100000ull * 300);
  CHECK(hipDeviceSynchronize());
  std::vector<Rec> h(waves);
  CHECK(hipMemcpy(h.data(), out, sizeof(Rec) * waves, hipMemcpyDeviceToHost));
  double it = 0, cyc = 0;
  std::vector<double> clk;
  for (auto& r : h) { it += (double)r.iters; cyc += (double)r.cycles; clk.push_back((double)r.cycles / (double)r.ticks * 0.1); }
  std::sort(clk.begin(), clk.end());
  const double per_simd_it = it / 1024.0, mean_cyc = cyc / waves;
  constexpr int M = C0 + C1 + C2;
  printf("pipelined chains %d+%d+%d MFMA + N=%3d v_fma (VOP3)  waves/SIMD %d: %7.1f cycles per iteration per SIMD (issue bound %4d = %.2f of it; matrix pipe %4d = %.2f of it), held clock %.3f GHz\n",
         C0, C1, C2, N, w, mean_cyc / per_simd_it, 2 * N + 8 * M, (2 * N + 8 * M) / (mean_cyc / per_simd_it), 32 * M, 32 * M / (mean_cyc / per_simd_it), clk[clk.size() / 2]);
  fflush(stdout);
  return 0;
}

template <int C0, int C1, int C2, int N>
int run_chain(int w, const uint4* in, Rec* out) {
  const int blocks = 256, threads = 256 * w, waves = blocks * threads / 64;
  kchain<C0, C1, C2, N><<<blocks, threads>>>(out, in, 100000ull * 300);
  CHECK(hipDeviceSynchronize());
  std::vector<Rec> h(waves);
  CHECK(hipMemcpy(h.data(), out, sizeof(Rec) * waves, hipMemcpyDeviceToHost));
  double it = 0, cyc = 0;
  std::vector<double> clk;
  for (auto& r : h) { it += (double)r.iters; cyc += (double)r.cycles; clk.push_back((double)r.cycles / (double)r.ticks * 0.1); }
  std::sort(clk.begin(), clk.end());
  const double per_simd_it = it / 1024.0, mean_cyc = cyc / waves;
  constexpr int M = C0 + C1 + C2;
  printf("chains %d+%d+%d MFMA + N=%3d v_fma (VOP3)  waves/SIMD %d: %7.1f cycles per iteration per SIMD (issue bound %4d = %.2f of it; matrix pipe %4d = %.2f of it), held clock %.3f GHz\n",
         C0, C1, C2, N, w, mean_cyc / per_simd_it, 2 * N + 8 * M, (2 * N + 8 * M) / (mean_cyc / per_simd_it), 32 * M, 32 * M / (mean_cyc / per_simd_it), clk[clk.size() / 2]);
  fflush(stdout);
  return 0;
}

template <int M, int N, int ENC>
int run(int w, const uint4* in, Rec* out) {
  const int blocks = 256, threads = 256 * w, waves = blocks * threads / 64;
  k<M, N, ENC><<<blocks, threads>>>(out, in, 100000ull * 300);
  CHECK(hipDeviceSynchronize());
  std::vector<Rec> h(waves);
  CHECK(hipMemcpy(h.data(), out, sizeof(Rec) * waves, hipMemcpyDeviceToHost));
  double it = 0, cyc = 0;
  std::vector<double> clk;
  for (auto& r : h) { it += (double)r.iters; cyc += (double)r.cycles; clk.push_back((double)r.cycles / (double)r.ticks * 0.1); }
  std::sort(clk.begin(), clk.end());
  const double per_simd_it = it / 1024.0, mean_cyc = cyc / waves;
  printf("M=%d MFMA + N=%3d %s  waves/SIMD %d: %7.1f cycles per iteration per SIMD (issue bound %4d = %.2f of it), %7.1f per wave, held clock %.3f GHz\n", M, N,
         ENC == 2 ? "v_fmac (VOP2)" : "v_fma  (VOP3)", w, mean_cyc / per_simd_it, 2 * N + 8 * M, (2 * N + 8 * M) / (mean_cyc / per_simd_it), mean_cyc / (per_simd_it / w),
         clk[clk.size() / 2]);
  fflush(stdout);
  return 0;
}

int main(int argc, char**) {
  uint4* in; Rec* out;
  CHECK(hipMalloc(&in, sizeof(uint4) * 512));
  CHECK(hipMalloc(&out, sizeof(Rec) * 256 * 16));
  std::vector<unsigned> h(2048);
  unsigned s = 12345u;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    const unsigned lo = 0x3F00u | ((s >> 8) & 0xFFu), hi = 0x3F00u | ((s >> 16) & 0xFFu) | ((s >> 31) << 15);
    v = lo | (hi << 16);
  }
  CHECK(hipMemcpy(in, h.data(), 8192, hipMemcpyHostToDevice));
  if (argc > 2) {  // the software-pipelined chained shapes only
    for (int w : {1, 2}) {
      if (run_chain_pipelined<4, 2, 2, 160>(w, in, out)) return 1;
      if (run_chain_pipelined<4, 1, 1, 160>(w, in, out)) return 1;
      if (run_chain_pipelined<1, 1, 1, 144>(w, in, out)) return 1;
      if (run_chain_pipelined<1, 1, 1, 288>(w, in, out)) return 1;
    }
    if (run_chain_pipelined<4, 1, 1, 160, 768>(3, in, out)) return 1;  // 156 registers: three waves fit
    for (int w : {3, 4}) {                                             // 118 registers: four waves fit
      if (run_chain_pipelined<1, 1, 1, 144, 1024>(w, in, out)) return 1;
      if (run_chain_pipelined<1, 1, 1, 288, 1024>(w, in, out)) return 1;
    }
    for (int w : {1, 2, 3}) {
      if (run_chain<4, 2, 2, 160>(w, in, out)) return 1;
      if (run_chain<1, 1, 1, 144>(w, in, out)) return 1;
    }
    return 0;
  }
  if (argc > 1) {  // the chained shapes only
    for (int w : {2, 3, 4}) {
      if (run_chain<4, 1, 1, 160>(w, in, out)) return 1;
      if (run_chain<4, 2, 2, 160>(w, in, out)) return 1;
      if (run_chain<4, 3, 3, 160>(w, in, out)) return 1;
      if (run_chain<1, 1, 1, 288>(w, in, out)) return 1;
      if (run_chain<1, 1, 2, 128>(w, in, out)) return 1;
      if (run_chain<3, 3, 1, 96>(w, in, out)) return 1;
      if (run_chain<1, 1, 1, 144>(w, in, out)) return 1;
    }
    return 0;
  }
  for (int w : {2, 3, 4}) {
    if (run<3, 96, 2>(w, in, out)) return 1;
    if (run<3, 96, 3>(w, in, out)) return 1;
    if (run<0, 96, 2>(w, in, out)) return 1;
    if (run<0, 96, 3>(w, in, out)) return 1;
    if (run<5, 240, 2>(w, in, out)) return 1;
    if (run<5, 240, 3>(w, in, out)) return 1;
    if (run<3, 48, 2>(w, in, out)) return 1;
    if (run<1, 32, 2>(w, in, out)) return 1;
    if (run<1, 0, 2>(w, in, out)) return 1;
  }
  return 0;
}
