// Microbenchmark: sustained VALU issue rates on gfx950 for the instruction forms the scoring
// kernels use.  Every measured instruction is written as inline asm (the compiler cannot
// rewrite it); each loop iteration issues 16 independent instructions per pattern slot.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 2048;
constexpr int NACC = 16;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ sc) {
  float acc[NACC];
  double dacc[NACC];
  int cnt[NACC];
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f pacc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    acc[i] = threadIdx.x * 1e-3f + i;
    dacc[i] = threadIdx.x * 1e-3 + i;
    cnt[i] = 0;
    pacc[i] = v2f{acc[i], acc[i] + 0.5f};
  }
  const float s0 = sc[0];          // uniform -> SGPR
  float vb = sc[threadIdx.x & 1];  // per-lane -> VGPR
  float va = sc[(threadIdx.x & 1) ^ 1];
  const double ds0 = (double)s0;
  double dvb = (double)vb;
  v2f pa = v2f{va, va}, pb = v2f{vb, vb};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(acc[i]) : "v"(va), "v"(vb));
      if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(acc[i]) : "s"(s0), "v"(vb));
      if (KIND == 2) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[i]) : "s"(s0), "v"(vb));
      if (KIND == 3) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(acc[i]) : "s"(s0));
      if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(pacc[i]) : "v"(pa), "v"(pb));
      if (KIND == 5) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(dacc[i]) : "s"(ds0), "v"(dvb));
      if (KIND == 6)  // count idiom A: compare to VCC, add-with-carry
        asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(cnt[i]) : "v"(acc[i]), "v"(vb) : "vcc");
      if (KIND == 7)  // count idiom B: compare to an SGPR pair, select 0/1, add
        asm volatile("v_cmp_lt_f32_e64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 v255, 0, 1, s[20:21]\n\tv_add_u32_e32 %0, %0, v255"
                     : "+v"(cnt[i]) : "v"(acc[i]), "v"(vb) : "s20", "s21", "v255");
      if (KIND == 8) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(acc[i]) : "v"(vb));
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) r += acc[i] + (float)dacc[i] + pacc[i].x + pacc[i].y + cnt[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
int run(const char* name, int instr_per_slot, int waves_per_simd, float flops_per_instr) {
  float *out, *sc;
  const int blocks = 256 * waves_per_simd;  // 4 waves per block -> waves_per_simd per SIMD on 256 CUs
  CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
  CHECK(hipMalloc(&sc, 64));
  float h[2] = {1.0001f, 0.5f};
  CHECK(hipMemcpy(sc, h, 8, hipMemcpyHostToDevice));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int w = 0; w < 3; ++w) k<KIND><<<blocks, 256>>>(out, sc);
  CHECK(hipEventRecord(a));
  const int reps = 10;
  for (int w = 0; w < reps; ++w) k<KIND><<<blocks, 256>>>(out, sc);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  const double wave_instr = (double)blocks * 4 * ITER * NACC * instr_per_slot;
  const double per_simd = wave_instr / 1024.0;
  printf("%-34s waves/SIMD %d: %7.3f ms  %5.2f ns per wave-instruction per SIMD  %7.1f TFLOP/s\n", name, waves_per_simd, ms, ms * 1e6 / per_simd,
         wave_instr * 64 * flops_per_instr / (ms * 1e-3) / 1e12);
  CHECK(hipFree(out));
  CHECK(hipFree(sc));
  return 0;
}

int main() {
  for (int w : {1, 2, 3, 4, 8}) {
    if (run<0>("v_fma_f32 v,v,v", 1, w, 2)) return 1;
    if (run<1>("v_fma_f32 s,v,v", 1, w, 2)) return 1;
    if (run<2>("v_fmac_f32_e32 s,v", 1, w, 2)) return 1;
    if (run<3>("v_mul_f32_e32 s,v", 1, w, 1)) return 1;
    if (run<8>("v_add_f32_e32 v,v", 1, w, 1)) return 1;
    if (run<4>("v_pk_fma_f32 v,v,v", 1, w, 4)) return 1;
    if (run<5>("v_fma_f64 s,v,v", 1, w, 2)) return 1;
    if (run<6>("v_cmp(vcc)+v_addc", 2, w, 0)) return 1;
    if (run<7>("v_cmp(sgpr)+v_cndmask+v_add", 3, w, 0)) return 1;
  }
  return 0;
}
