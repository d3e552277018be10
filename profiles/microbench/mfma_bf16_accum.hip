// Accumulation numerics of v_mfma_f32_32x32x16_bf16 on gfx950: how far is D from the exactly
// rounded sum of the 16 exact products?  Inputs mimic a 3-piece bf16 split of two operands
// (terms of magnitude 1, 2^-9, 2^-18) and also uniformly scaled random terms.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_bf16_accum mfma_bf16_accum.hip && ./mfma_bf16_accum
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void k(const uint16_t* A, const uint16_t* B, float* D, int tiles) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint16_t* a = A + ((size_t)t * 32 + r) * 16 + 8 * h;  // A[row r][k = 8h + j]
    const uint16_t* b = B + ((size_t)t * 32 + r) * 16 + 8 * h;  // B[k = 8h + j][col r]
    bf16x8 av, bv;
    for (int j = 0; j < 8; ++j) {
      av[j] = __builtin_bit_cast(__bf16, a[j]);
      bv[j] = __builtin_bit_cast(__bf16, b[j]);
    }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      D[((size_t)t * 32 + row) * 32 + r] = acc[i];
    }
  }
}

static uint16_t to_bf16(double v) {
  float f = (float)v;
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static double from_bf16(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main() {
  const int tiles = 2048;
  std::vector<uint16_t> A((size_t)tiles * 32 * 16), B(A.size());
  srand(1);
  auto rnd = []() { return (rand() / (double)RAND_MAX) * 2.0 - 1.0; };
  for (int mode = 0; mode < 3; ++mode) {
    for (size_t v = 0; v < (size_t)tiles * 32; ++v) {
      for (int kk = 0; kk < 16; ++kk) {
        double sa, sb;
        if (mode == 0) { sa = 1.0; sb = 1.0; }
        else if (mode == 1) { sa = ldexp(1.0, -9 * (kk % 3)); sb = ldexp(1.0, -9 * ((kk / 3) % 3)); }
        else { sa = ldexp(1.0, -(rand() % 20)); sb = ldexp(1.0, -(rand() % 20)); }
        A[v * 16 + kk] = to_bf16(rnd() * sa);
        B[v * 16 + kk] = to_bf16(rnd() * sb);
      }
    }
    uint16_t *dA, *dB; float* dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, (size_t)tiles * 1024 * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    k<<<256, 64>>>(dA, dB, dD, tiles);
    std::vector<float> D((size_t)tiles * 1024);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double worst_abs = 0, worst_ulp = 0, bias = 0; long n = 0, exact_rn = 0, toward_zero = 0, inexact = 0;
    for (int t = 0; t < tiles; ++t)
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double s = 0, sabs = 0;
          for (int kk = 0; kk < 16; ++kk) {
            const double p = from_bf16(A[((size_t)t * 32 + i) * 16 + kk]) * from_bf16(B[((size_t)t * 32 + j) * 16 + kk]);
            s += p;  // 16 products of 16-bit mantissas: exact in double for these exponent ranges
            sabs += fabs(p);
          }
          const double d = D[((size_t)t * 32 + i) * 32 + j];
          const double e = fabs(d - s) / (sabs * ldexp(1.0, -24));
          if (e > worst_abs) worst_abs = e;
          const double ulp = ldexp(1.0, ilogb(fabs(s) > 0 ? fabs(s) : 1e-300) - 23);
          if (fabs(d - s) / ulp > worst_ulp) worst_ulp = fabs(d - s) / ulp;
          if ((float)s == (float)d) ++exact_rn;
          if (d != s) { ++inexact; if (fabs(d) < fabs(s)) ++toward_zero; }
          bias += (d - s) / (sabs * ldexp(1.0, -24));
          ++n;
        }
    printf("mode %d: outputs %ld  max|err| = %.3f x 2^-24 x sum|terms|  max %.3f ulp(of the sum)  equal to RN(exact sum): %.4f  "
           "inexact results smaller in magnitude: %.4f  mean signed err %.4f\n",
           mode, n, worst_abs, worst_ulp, exact_rn / (double)n, toward_zero / (double)(inexact ? inexact : 1), bias / n);
    hipFree(dA); hipFree(dB); hipFree(dD);
  }
  return 0;
}
