// Host-side harness for the unit tests: compiles the library's small-matrix numerics
// (csrc/sfm_math.hpp, host branch) with g++ so that the solvers can be checked against the
// oracle on a machine without a GPU.  Not part of the shipped library.
#include <cstdint>
#include "sfm_math.hpp"

using namespace sfm;

extern "C" {

void h_solve_eh(const double* corr, const int32_t* idx, int64_t nb, double* E, double* H) {
  for (int64_t b = 0; b < nb; ++b) {
    double p8[8][4], p4[4][4];
    for (int i = 0; i < 8; ++i)
      for (int c = 0; c < 4; ++c) p8[i][c] = corr[4 * (int64_t)idx[12 * b + i] + c];
    for (int i = 0; i < 4; ++i)
      for (int c = 0; c < 4; ++c) p4[i][c] = corr[4 * (int64_t)idx[12 * b + 8 + i] + c];
    double e[9], h[9];
    solve_E8<double>(p8, e);
    solve_H4<double>(p4, h);
    for (int i = 0; i < 9; ++i) { E[9 * b + i] = e[i]; H[9 * b + i] = h[i]; }
  }
}

void h_score_eh(const double* corr, int n, const double* E, const double* H, int64_t nb, double thr_e, double thr_h,
                int32_t* ce, int32_t* ch) {
  for (int64_t b = 0; b < nb; ++b) {
    double e[9], h[9];
    for (int i = 0; i < 9; ++i) { e[i] = E[9 * b + i]; h[i] = H[9 * b + i]; }
    int a = 0, c = 0;
    for (int j = 0; j < n; ++j) {
      const double* p = corr + 4 * (int64_t)j;
      a += epipolar_inlier<double>(e, p[0], p[1], p[2], p[3], 2 * thr_e * thr_e);
      c += transfer_inlier<double>(h, p[0], p[1], p[2], p[3], thr_h * thr_h);
    }
    ce[b] = a; ch[b] = c;
  }
}

void h_triangulate(const double* P1, const double* P2, const double* x1, const double* x2, int m, int ld, double* X) {
  double p1[12], p2[12];
  for (int i = 0; i < 12; ++i) { p1[i] = P1[i]; p2[i] = P2[i]; }
  for (int j = 0; j < m; ++j) {
    double out[4];
    triangulate_point<double>(p1, p2, x1[j], x1[ld + j], x2[j], x2[ld + j], out);
    for (int i = 0; i < 4; ++i) X[(int64_t)i * m + j] = out[i];
  }
}

void h_essential_candidates(const double* E, double* P) {
  double e[9], p[4][12];
  for (int i = 0; i < 9; ++i) e[i] = E[i];
  essential_candidates<double>(e, p);
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < 12; ++i) P[12 * c + i] = p[c][i];
}

void h_homography_decompose(const double* H, int flip, double* out /* R1 t1 R2 t2 Ea Eb = 42 */) {
  double h[9], R1[3][3], R2[3][3], t1[3], t2[3], Ea[9], Eb[9];
  for (int i = 0; i < 9; ++i) h[i] = H[i];
  homography_decompose<double>(h, flip != 0, R1, t1, R2, t2);
  skew_times<double>(t1, R1, Ea);
  skew_times<double>(t2, R2, Eb);
  for (int i = 0; i < 9; ++i) { out[i] = R1[i / 3][i % 3]; out[12 + i] = R2[i / 3][i % 3]; out[24 + i] = Ea[i]; out[33 + i] = Eb[i]; }
  for (int i = 0; i < 3; ++i) { out[9 + i] = t1[i]; out[21 + i] = t2[i]; }
}

void h_solve_t2(const double* pts /* [m][6]: rx ry rz u v w */, const int32_t* idx, int64_t nb, double* T) {
  for (int64_t b = 0; b < nb; ++b) {
    double x[2][3], r[2][3], t[3];
    for (int j = 0; j < 2; ++j) {
      const double* p = pts + 6 * (int64_t)idx[2 * b + j];
      r[j][0] = p[0]; r[j][1] = p[1]; r[j][2] = p[2];
      x[j][0] = p[3]; x[j][1] = p[4]; x[j][2] = p[5];
    }
    solve_T2<double>(x, r, idx[2 * b] == idx[2 * b + 1], t);
    for (int i = 0; i < 3; ++i) T[3 * b + i] = t[i];
  }
}

void h_score_t(const double* pts, int m, const double* T, int64_t nb, double thr, int32_t* cnt) {
  for (int64_t b = 0; b < nb; ++b) {
    int c = 0;
    for (int j = 0; j < m; ++j) {
      const double* p = pts + 6 * (int64_t)j;
      c += reprojection_inlier<double>(T[3 * b], T[3 * b + 1], T[3 * b + 2], p[0], p[1], p[2], p[3], p[4], thr * thr);
    }
    cnt[b] = c;
  }
}
void h_index_sets(uint64_t seed, uint64_t first_id, int64_t nb, uint32_t n, int width, int32_t* idx) {
  for (int64_t b = 0; b < nb; ++b) {
    if (width == 12) {
      int32_t row[12];
      philox_index_row<12>(seed, first_id + (uint64_t)b, n, row);
      for (int j = 0; j < 12; ++j) idx[12 * b + j] = row[j];
    } else {
      int32_t row[2];
      philox_index_row<2>(seed, first_id + (uint64_t)b, n, row);
      for (int j = 0; j < 2; ++j) idx[2 * b + j] = row[j];
    }
  }
}

void h_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
  uint32_t o[4];
  philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], o);
  for (int j = 0; j < 4; ++j) out[j] = o[j];
}

}  // extern "C"
