// The C ABI used from plain C++ (no Python, no torch): hipMalloc'd buffers, device-generated index sets, exact scoring,
// the bulk arg-max step and the exact translation RANSAC; every result is cross-checked inside this program.
//   hipcc -O2 --offload-arch=gfx950 -Iinclude tests/c_abi/c_abi_demo.cpp -Lstructure-from-motion_amd/lib -lsfm_hip -o tests/c_abi/c_abi_demo
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "sfm_hip.h"

#define CHECK(x)                                                                          \
  do {                                                                                    \
    int rc_ = (int)(x);                                                                   \
    if (rc_ != 0) {                                                                       \
      fprintf(stderr, "%s failed: %d (%s)\n", #x, rc_, sfm_hip_error_string(rc_));        \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

static double urand() { return rand() / (double)RAND_MAX; }

int main() {
  srand(3);
  const int n = 3000;
  const int64_t nb = 50000;
  // a synthetic two-view scene: half of the matches follow a rigid motion, the rest are uniform
  std::vector<double> corr(4 * (size_t)n);
  const double c = cos(0.1), s = sin(0.1), t[3] = {1.0, 0.1, 0.05};
  double xmax = 0;
  for (int j = 0; j < n; ++j) {
    const double X = 4 * urand() - 2, Y = 4 * urand() - 2, Z = 4 + 4 * urand();
    const double x2 = c * X + s * Z + t[0], y2 = Y + t[1], z2 = -s * X + c * Z + t[2];
    corr[4 * j] = X / Z;
    corr[4 * j + 1] = Y / Z;
    if (j % 2) {
      corr[4 * j + 2] = x2 / z2 + 1e-4 * (urand() - 0.5);
      corr[4 * j + 3] = y2 / z2 + 1e-4 * (urand() - 0.5);
    } else {
      corr[4 * j + 2] = 0.8 * urand() - 0.4;
      corr[4 * j + 3] = 0.5 * urand() - 0.25;
    }
    for (int k = 0; k < 4; ++k) xmax = fmax(xmax, fabs(corr[4 * j + k]));
  }
  const double thr = 1.0 / 2400.0;
  double* d_corr;
  float* d_corr32;
  int32_t *d_idx, *d_ce, *d_ch;
  CHECK(hipMalloc(&d_corr, corr.size() * sizeof(double)));
  CHECK(hipMalloc(&d_corr32, (size_t)sfm_corr_f32_bytes(n)));
  CHECK(hipMalloc(&d_idx, 12 * nb * sizeof(int32_t)));
  CHECK(hipMalloc(&d_ce, nb * sizeof(int32_t)));
  CHECK(hipMalloc(&d_ch, nb * sizeof(int32_t)));
  CHECK(hipMemcpy(d_corr, corr.data(), corr.size() * sizeof(double), hipMemcpyHostToDevice));
  CHECK(sfm_corr_to_f32(d_corr, n, d_corr32, nullptr));
  CHECK(sfm_index_sets_i32(42, 0, nb, n, 12, d_idx, nullptr));
  CHECK(sfm_eh_score_f64(d_corr, n, d_idx, nb, thr, 3 * thr, d_ce, d_ch, nullptr));
  std::vector<int32_t> ce(nb), ch(nb);
  CHECK(hipMemcpy(ce.data(), d_ce, nb * sizeof(int32_t), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(ch.data(), d_ch, nb * sizeof(int32_t), hipMemcpyDeviceToHost));
  int64_t be = 0, bh = 0;
  for (int64_t i = 1; i < nb; ++i) {
    if (ce[i] > ce[be]) be = i;
    if (ch[i] > ch[bh]) bh = i;
  }
  const int64_t ws_bytes = sfm_eh_best_workspace_bytes(nb, n);
  void* d_ws;
  CHECK(hipMalloc(&d_ws, ws_bytes));
  for (int loose = 0; loose <= 2; ++loose) {
    int64_t res[8];
    CHECK(sfm_eh_best(d_corr, d_corr32, n, d_idx, nb, thr, 3 * thr, xmax, 2.0, loose, d_ws, ws_bytes, res, nullptr));
    if (res[0] != ce[be] || res[1] != be || res[2] != ch[bh] || res[3] != bh) {
      fprintf(stderr, "sfm_eh_best(loose=%d): (%lld,%lld,%lld,%lld) but the first maxima are (%d,%lld,%d,%lld)\n", loose, (long long)res[0],
              (long long)res[1], (long long)res[2], (long long)res[3], ce[be], (long long)be, ch[bh], (long long)bh);
      return 2;
    }
  }
  // exact counts from the matrix cores + float32: with the early loops (options 0) and with every pair through the full tests
  {
    const int64_t xs_bytes = sfm_eh_score_mixed_workspace_bytes(nb, n);
    void* d_xs;
    int32_t *d_me, *d_mh;
    CHECK(hipMalloc(&d_xs, xs_bytes));
    CHECK(hipMalloc(&d_me, nb * sizeof(int32_t)));
    CHECK(hipMalloc(&d_mh, nb * sizeof(int32_t)));
    std::vector<int32_t> me(nb), mh(nb);
    for (int options = 0; options <= SFM_EHX_EVERY_PAIR_FULL_TEST; ++options) {
      CHECK(sfm_eh_score_mixed_opt(d_corr, n, d_idx, nb, thr, 3 * thr, xmax, options, d_xs, xs_bytes, d_me, d_mh, nullptr));
      CHECK(hipMemcpy(me.data(), d_me, nb * sizeof(int32_t), hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(mh.data(), d_mh, nb * sizeof(int32_t), hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < nb; ++i) {
        if (me[i] != ce[i] || mh[i] != ch[i]) {
          fprintf(stderr, "sfm_eh_score_mixed_opt(options=%d): hypothesis %lld has (%d, %d), float64 (%d, %d)\n", options, (long long)i, me[i], mh[i],
                  ce[i], ch[i]);
          return 4;
        }
      }
    }
    CHECK(hipFree(d_xs));
    CHECK(hipFree(d_me));
    CHECK(hipFree(d_mh));
  }
  // translation RANSAC: all-float64 against matrix cores + float32 with the float64 band
  const int m = 5000;
  const int64_t nbt = 20000;
  std::vector<double> pts(6 * (size_t)m);
  for (int j = 0; j < m; ++j) {
    const double X = 4 * urand() - 2, Y = 4 * urand() - 2, Z = 4 + 4 * urand();
    pts[6 * j] = X; pts[6 * j + 1] = Y; pts[6 * j + 2] = Z;
    const bool in = j % 2;
    pts[6 * j + 3] = in ? (X + 0.3) / (Z + 0.2) + 2e-4 * (urand() - 0.5) : 0.8 * urand() - 0.4;
    pts[6 * j + 4] = in ? (Y - 0.1) / (Z + 0.2) + 2e-4 * (urand() - 0.5) : 0.5 * urand() - 0.25;
    pts[6 * j + 5] = 1.0;
  }
  double *d_pts, *d_T;
  int32_t *d_idx2, *d_c1, *d_c2;
  uint64_t* d_best;
  CHECK(hipMalloc(&d_pts, pts.size() * sizeof(double)));
  CHECK(hipMalloc(&d_T, 3 * sizeof(double)));
  CHECK(hipMalloc(&d_idx2, 2 * nbt * sizeof(int32_t)));
  CHECK(hipMalloc(&d_c1, nbt * sizeof(int32_t)));
  CHECK(hipMalloc(&d_c2, nbt * sizeof(int32_t)));
  CHECK(hipMalloc(&d_best, sizeof(uint64_t)));
  CHECK(hipMemcpy(d_pts, pts.data(), pts.size() * sizeof(double), hipMemcpyHostToDevice));
  CHECK(sfm_index_sets_i32(7, 0, nbt, m, 2, d_idx2, nullptr));
  const int64_t cam_off[2] = {0, m};
  const double thr_t = 4.0 / 2400.0;
  uint64_t best1 = 0, best2 = 0;
  CHECK(sfm_t_ransac_f64(d_pts, cam_off, 1, d_idx2, nbt, thr_t, d_c1, d_best, d_T, nullptr));
  CHECK(hipMemcpy(&best1, d_best, sizeof(best1), hipMemcpyDeviceToHost));
  const int64_t tws_bytes = sfm_t_ransac_mixed_workspace_bytes(cam_off, 1);
  void* d_tws;
  CHECK(hipMalloc(&d_tws, tws_bytes));
  CHECK(sfm_t_ransac_mixed(d_pts, cam_off, 1, d_idx2, nbt, thr_t, d_tws, tws_bytes, d_c2, d_best, d_T, nullptr));
  CHECK(hipMemcpy(&best2, d_best, sizeof(best2), hipMemcpyDeviceToHost));
  std::vector<int32_t> c1(nbt), c2(nbt);
  CHECK(hipMemcpy(c1.data(), d_c1, nbt * sizeof(int32_t), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(c2.data(), d_c2, nbt * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < nbt; ++i)
    if (c1[i] != c2[i]) {
      fprintf(stderr, "translation counts differ at %lld: %d vs %d\n", (long long)i, c1[i], c2[i]);
      return 3;
    }
  if (best1 != best2 || (best1 >> 32) < (uint64_t)(m / 2 - 50)) {
    fprintf(stderr, "translation winners: %llx vs %llx\n", (unsigned long long)best1, (unsigned long long)best2);
    return 4;
  }
  double T[3];
  CHECK(hipMemcpy(T, d_T, sizeof(T), hipMemcpyDeviceToHost));
  printf("c_abi_demo ok: E winner (%d, %lld), H winner (%d, %lld), T winner count %llu, T = (%.4f, %.4f, %.4f)\n", ce[be], (long long)be,
         ch[bh], (long long)bh, (unsigned long long)(best1 >> 32), T[0], T[1], T[2]);
  return 0;
}
