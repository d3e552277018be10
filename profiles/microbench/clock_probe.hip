// In-kernel clock of an MI355X under vector load, measured the way the chip itself counts:
//   clock = delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz reference) x 100 MHz,
// stamped once around the loop of every wave, after about two seconds of back-to-back launches of the same
// kernel on random data (the clock the chip HOLDS, not the boost of a cold start).
// Loops:  M x v_mfma_f32_32x32x16_bf16 then N x v_fma_f32 per iteration (MODE 1: the FMAs read the accumulators the
// MFMAs of the same iteration wrote, the shape of the scoring kernels; MODE 0: independent registers), W waves per SIMD
// on all 256 CUs: one workgroup of 4 W waves per CU, so the W waves of a SIMD start together.  Every wave runs a FIXED
// number of iterations, and the waves of a SIMD are not served equally, so they finish at different times ("waves
// resident" below 1): cycles per iteration per SIMD are then an underestimate and the figures per vector instruction an
// overestimate; valu_issue_probe.hip and mfma_valu_mix_probe.hip measure rates without that tail.  Printed per
// configuration: wall time per launch, held clock (median over waves), shader cycles per iteration and SIMD, nanoseconds
// per vector instruction and SIMD, and how much of the launch a wave was resident for.
// Build: hipcc -O3 --offload-arch=gfx950 -o clock_probe clock_probe.hip ; run: ./clock_probe [iterations]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
static int ITER = 40000;

struct Stamp {
  unsigned long long cycles, ticks, t_begin, t_end;
};

template <int M, int N, int MODE>
__global__ __launch_bounds__(1024) void k(float* out, const uint4* __restrict__ in, Stamp* st, int iters) {
  const uint4 ua = in[threadIdx.x & 255], ub = in[256 + (threadIdx.x & 255)];
  const bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = __uint_as_float(ua.x) * (i + 1);
  const float c0 = __uint_as_float(ub.y), c1 = __uint_as_float(ub.z);
  constexpr int MA = M > 0 ? M : 1;
  f32x16 acc[MA];
  const f32x16 zero = {0};
#pragma unroll
  for (int m = 0; m < MA; ++m) acc[m] = zero;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, MODE == 1 ? zero : acc[m], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (MODE == 1)
        x[j % 16] = fmaf(acc[j % MA][(j / MA) % 16], c0, x[j % 16]);
      else
        x[j % 16] = fmaf(x[j % 16], c0, c1);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(x[i]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    r += x[i];
#pragma unroll
    for (int m = 0; m < MA; ++m) r += acc[m][i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = Stamp{t1 - t0, r1 - r0, r0, r1};
}

template <int M, int N, int MODE>
int run(int waves_per_simd) {
  // one workgroup of 4 * W waves per CU: the W waves of a SIMD start together and stay together
  const int blocks = 256, threads = 256 * waves_per_simd, waves = blocks * threads / 64;
  float* out;
  uint4* in;
  Stamp* st;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * threads));
  CHECK(hipMalloc(&in, sizeof(uint4) * 512));
  CHECK(hipMalloc(&st, sizeof(Stamp) * waves));
  std::vector<unsigned> h(2048);
  unsigned s = 12345u;
  for (auto& v : h) {  // random bf16 pairs of moderate magnitude, both signs
    s = s * 1664525u + 1013904223u;
    const unsigned lo = 0x3F00u | ((s >> 8) & 0xFFu), hi = 0x3F00u | ((s >> 16) & 0xFFu) | ((s >> 31) << 15);
    v = lo | (hi << 16);
  }
  CHECK(hipMemcpy(in, h.data(), 8192, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  // about two seconds of back-to-back launches before anything is measured
  k<M, N, MODE><<<blocks, threads>>>(out, in, st, ITER);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<M, N, MODE><<<blocks, threads>>>(out, in, st, ITER);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float one_ms;
  CHECK(hipEventElapsedTime(&one_ms, e0, e1));
  const int warm = std::max(3, (int)(2000.0f / one_ms));
  for (int w = 0; w < warm; ++w) k<M, N, MODE><<<blocks, threads>>>(out, in, st, ITER);
  CHECK(hipEventRecord(e0));
  const int reps = 5;
  for (int w = 0; w < reps; ++w) k<M, N, MODE><<<blocks, threads>>>(out, in, st, ITER);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::vector<Stamp> hs(waves);
  CHECK(hipMemcpy(hs.data(), st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
  std::vector<double> clk, cyc;
  unsigned long long first = ~0ull, last = 0;
  double resident = 0;
  for (auto& v : hs) {
    clk.push_back((double)v.cycles / (double)v.ticks * 0.1);  // GHz
    cyc.push_back((double)v.cycles);
    first = std::min(first, v.t_begin);
    last = std::max(last, v.t_end);
  }
  for (auto& v : hs) resident += (double)(v.t_end - v.t_begin) / (double)(last - first);
  resident /= hs.size();
  std::sort(clk.begin(), clk.end());
  std::sort(cyc.begin(), cyc.end());
  const double clock = clk[clk.size() / 2], cycles = cyc[cyc.size() / 2];
  const double per_iter = cycles / ITER / waves_per_simd;
  const double ns_per_valu = N > 0 ? ms * 1e6 / ((double)ITER * waves_per_simd * N) : 0.0;
  printf("mode %d  M=%d MFMA  N=%3d VALU  waves/SIMD %d: %8.3f ms  held clock %.3f GHz (min %.3f max %.3f)  %7.1f cycles per iteration per SIMD "
         "(issue-bound %4d, serialised %4d)  %.3f ns per VALU instruction per SIMD  waves resident %.3f of the launch\n",
         MODE, M, N, waves_per_simd, ms, clock, clk.front(), clk.back(), per_iter, 2 * N + 8 * M, 2 * N + 32 * M, ns_per_valu, resident);
  fflush(stdout);
  CHECK(hipFree(out));
  CHECK(hipFree(in));
  CHECK(hipFree(st));
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1) ITER = atoi(argv[1]);
  for (int w : {1, 2, 3, 4}) {
    if (run<0, 96, 0>(w)) return 1;
    if (run<3, 96, 1>(w)) return 1;
    if (run<1, 0, 0>(w)) return 1;
  }
  return 0;
}
