// Microbenchmark: how far do bf16 MFMAs and float32 VALU instructions of the waves of ONE SIMD overlap on gfx950?
// Every wave runs   loop { M x v_mfma_f32_32x32x16_bf16 ; N x v_fma_f32 }   with W waves per SIMD on all 256 CUs.
//   mode 0: the VALU instructions are independent of the MFMA results (separate registers);
//   mode 1: they read the accumulator the MFMAs of the same iteration wrote (the shape of the scoring kernels:
//           the matrix core produces a tile, the vector ALU tests it);
//   mode 2: as mode 1 with N/2 v_pk_fma_f32 on register pairs of the accumulator (the same flops in half the
//           instructions).
// Prints shader cycles per iteration and SIMD (s_memtime) beside the two ideal figures: issue-bound
// (2 cycles per VALU instruction + 8 per MFMA) and serialised (2 per VALU instruction + 32 per MFMA).
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_overlap mfma_valu_overlap.hip ; run: ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
static int ITER = 4000;  // iterations per wave (argv[1]); long runs show the clock the chip holds under sustained load

template <int M, int N, int MODE>
__global__ __launch_bounds__(256) void k(float* out, const uint4* __restrict__ in, unsigned long long* cyc, int iters) {
  const uint4 ua = in[threadIdx.x], ub = in[256 + threadIdx.x];
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
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, MODE == 1 ? zero : acc[m], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (MODE == 1)
        x[j % 16] = fmaf(acc[j % MA][(j / MA) % 16], c0, x[j % 16]);
      else if (MODE == 0)
        x[j % 16] = fmaf(x[j % 16], c0, c1);
    }
    if (MODE == 2) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      const v2f cc = {c0, c0};
#pragma unroll
      for (int j = 0; j < N / 2; ++j) {
        const int r = 2 * ((j / MA) % 8);
        v2f a2 = {acc[j % MA][r], acc[j % MA][r + 1]};
        v2f x2 = {x[2 * (j % 8)], x[2 * (j % 8) + 1]};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x2) : "v"(a2), "v"(cc));
        x[2 * (j % 8)] = x2.x;
        x[2 * (j % 8) + 1] = x2.y;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(x[i]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    r += x[i];
#pragma unroll
    for (int m = 0; m < MA; ++m) r += acc[m][i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int M, int N, int MODE>
int run(int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;
  float* out;
  uint4* in;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
  CHECK(hipMalloc(&in, sizeof(uint4) * 512));
  CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4));
  std::vector<unsigned> h(2048);
  unsigned s = 12345u;
  for (auto& v : h) {  // random bf16 pairs of moderate magnitude (random data: the clock the chip holds depends on it)
    s = s * 1664525u + 1013904223u;
    const unsigned lo = 0x3F00u | ((s >> 8) & 0xFFu), hi = 0x3F00u | ((s >> 16) & 0xFFu) | ((s >> 31) << 15);
    v = lo | (hi << 16);
  }
  CHECK(hipMemcpy(in, h.data(), 8192, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) k<M, N, MODE><<<blocks, 256>>>(out, in, cyc, ITER);
  CHECK(hipEventRecord(e0));
  const int reps = 5;
  for (int w = 0; w < reps; ++w) k<M, N, MODE><<<blocks, 256>>>(out, in, cyc, ITER);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::vector<unsigned long long> hc(blocks * 4);
  CHECK(hipMemcpy(hc.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto v : hc) mean += (double)v;
  mean /= hc.size();
  // every wave of a SIMD runs ITER iterations concurrently: per SIMD, W * ITER iterations complete in `mean` cycles
  const double per_iter = mean / ITER / waves_per_simd;
  printf("mode %d  M=%d MFMA  N=%3d VALU  waves/SIMD %d: %7.3f ms  %7.1f cycles per iteration per SIMD   issue-bound %4d  serialised %4d   (clock %.2f GHz)\n",
         MODE, M, N, waves_per_simd, ms, per_iter, 2 * N + 8 * M, 2 * N + 32 * M, mean / (ms * 1e-3) / 1e9);
  CHECK(hipFree(out));
  CHECK(hipFree(in));
  CHECK(hipFree(cyc));
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1) ITER = atoi(argv[1]);
  for (int w : {2, 3, 4}) {
    if (run<1, 0, 0>(w)) return 1;
    if (run<0, 32, 0>(w)) return 1;
    if (run<0, 96, 0>(w)) return 1;
    if (run<1, 8, 0>(w)) return 1;
    if (run<1, 16, 0>(w)) return 1;
    if (run<1, 32, 0>(w)) return 1;
    if (run<1, 64, 0>(w)) return 1;
    if (run<3, 96, 0>(w)) return 1;
    if (run<1, 16, 1>(w)) return 1;
    if (run<1, 32, 1>(w)) return 1;
    if (run<1, 64, 1>(w)) return 1;
    if (run<3, 96, 1>(w)) return 1;
    if (run<3, 96, 2>(w)) return 1;
  }
  return 0;
}
