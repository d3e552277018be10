// Microbenchmark: sustained VALU issue rates on gfx950 for the instruction kinds the scoring
// kernels use.  Each kernel runs ITER iterations of 32 independent accumulator chains per lane.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 4096;
constexpr int NACC = 16;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ sc, float a, float b) {
  float acc[NACC];
  double dacc[NACC / 2];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) dacc[i] = threadIdx.x * 1e-3 + i;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f pacc[NACC / 2];
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) pacc[i] = v2f{acc[2 * i], acc[2 * i + 1]};
  const float s0 = sc[0], s1 = sc[1];  // uniform -> SGPR
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) {  // v_fma_f32, all VGPR operands
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
    } else if (KIND == 1) {  // v_fma_f32 with an SGPR operand
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fmaf(acc[i], s0, s1 > 0 ? acc[(i + 1) % NACC] : b);
    } else if (KIND == 2) {  // v_pk_fma_f32
#pragma unroll
      for (int i = 0; i < NACC / 2; ++i) pacc[i] = __builtin_elementwise_fma(pacc[i], v2f{a, a}, v2f{b, b});
    } else if (KIND == 3) {  // v_fma_f64
#pragma unroll
      for (int i = 0; i < NACC / 2; ++i) dacc[i] = __builtin_fma(dacc[i], (double)a, (double)b);
    } else if (KIND == 4) {  // v_mul_f32
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = acc[i] * a;
    } else if (KIND == 5) {  // v_cmp + v_addc (count idiom)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] += (acc[(i + 1) % NACC] < a) ? 1.0f : 0.0f;
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) r += acc[i];
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) r += (float)dacc[i] + pacc[i].x + pacc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
int run(const char* name, int ops_per_iter, int waves_per_simd, float flops_per_op) {
  float *out, *sc;
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd) blocks of 4 waves = waves_per_simd per SIMD
  CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
  CHECK(hipMalloc(&sc, 64));
  float h[2] = {1.0001f, 0.5f};
  CHECK(hipMemcpy(sc, h, 8, hipMemcpyHostToDevice));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int w = 0; w < 2; ++w) k<KIND><<<blocks, 256>>>(out, sc, 1.0001f, 0.5f);
  CHECK(hipEventRecord(a));
  const int reps = 5;
  for (int w = 0; w < reps; ++w) k<KIND><<<blocks, 256>>>(out, sc, 1.0001f, 0.5f);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  const double wave_instr = (double)blocks * 4 * ITER * ops_per_iter;   // wave-instructions
  const double per_simd = wave_instr / 1024.0;
  printf("%-28s waves/SIMD %d: %8.3f ms  %6.2f ns per wave-instr per SIMD  (%.2f cycles @2.4GHz)  %7.1f TFLOP/s\n", name, waves_per_simd, ms,
         ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, wave_instr * 64 * flops_per_op / (ms * 1e-3) / 1e12);
  CHECK(hipFree(out));
  CHECK(hipFree(sc));
  return 0;
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    if (run<0>("v_fma_f32 vgpr", NACC, w, 2)) return 1;
    if (run<1>("v_fma_f32 sgpr operand", NACC, w, 2)) return 1;
    if (run<2>("v_pk_fma_f32", NACC / 2, w, 4)) return 1;
    if (run<3>("v_fma_f64", NACC / 2, w, 2)) return 1;
    if (run<4>("v_mul_f32", NACC, w, 1)) return 1;
    if (run<5>("v_cmp+v_cndmask+v_add", NACC, w, 1)) return 1;
  }
  return 0;
}
