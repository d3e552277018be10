// Device-side small-matrix numerics for the robust-estimation kernels (gfx950).
//
// Everything here is written for one model per lane with all matrix indices known at
// compile time, so that every matrix lives in VGPRs (a runtime-indexed array would be
// demoted to scratch memory).  Arithmetic uses explicit fma() and the translation unit is
// compiled with -ffp-contract=off: the bulk scoring kernel and the record-event kernels
// then evaluate one residual with exactly the same instruction sequence, so a mask
// recomputed at a record event always sums to the count the bulk kernel reported.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SFM_FN __device__ __forceinline__
#define SFM_WAVE_ALL(x) __all(x)
#else
// Host build of the same numerics.  Only the unit-test harness under tests/ compiles this
// branch (to debug the small-matrix code without a GPU); the shipped library is device code.
#include <cmath>
#include <cstdint>
#define SFM_FN inline
#define SFM_WAVE_ALL(x) (x)
using std::copysign;
using std::fabs;
using std::fma;
using std::fmax;
using std::sqrt;
#endif

namespace sfm {

template <typename T> struct num_traits;
template <> struct num_traits<double> {
  static constexpr double eps = 2.220446049250313e-16;
};
template <> struct num_traits<float> {
  static constexpr float eps = 1.1920929e-7f;
};

template <typename T> SFM_FN T tsqrt(T x) { return sqrt(x); }
template <typename T> SFM_FN T tabs(T x) { return fabs(x); }

// ---------------------------------------------------------------------------------------
// Null vector of an 8x9 system.  a holds the TRANSPOSE (9 rows, 8 columns); Householder QR
// of it leaves the null vector as the last column of Q:  q = H0 H1 ... H7 e8.
// Backward stable (error ~ eps * cond), no pivoting, no data-dependent control flow.
// Restates the role of `Vh[-1]` in the reference's DLT solvers
// (src/ransac_algorithm.py:256-258 and :287-290); the sign of q is arbitrary there too.
// ---------------------------------------------------------------------------------------
template <typename T>
SFM_FN void null_vector_9x8(T (&a)[9][8], T (&q)[9]) {
  T beta[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    T tail2 = T(0);
#pragma unroll
    for (int i = k + 1; i < 9; ++i) tail2 = fma(a[i][k], a[i][k], tail2);
    const T x0 = a[k][k];
    const T nrm = tsqrt(fma(x0, x0, tail2));
    const T v0 = (x0 >= T(0)) ? x0 + nrm : x0 - nrm;
    const T vtv = fma(v0, v0, tail2);
    const T b = (vtv > T(0)) ? T(2) / vtv : T(0);
    a[k][k] = v0;
    beta[k] = b;
#pragma unroll
    for (int j = k + 1; j < 8; ++j) {
      T d = T(0);
#pragma unroll
      for (int i = k; i < 9; ++i) d = fma(a[i][k], a[i][j], d);
      d *= b;
#pragma unroll
      for (int i = k; i < 9; ++i) a[i][j] = fma(-d, a[i][k], a[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) q[i] = (i == 8) ? T(1) : T(0);
#pragma unroll
  for (int k = 7; k >= 0; --k) {
    T d = T(0);
#pragma unroll
    for (int i = k; i < 9; ++i) d = fma(a[i][k], q[i], d);
    d *= beta[k];
#pragma unroll
    for (int i = k; i < 9; ++i) q[i] = fma(-d, a[i][k], q[i]);
  }
}

// ---------------------------------------------------------------------------------------
// One-sided (Hestenes) Jacobi SVD of an NxN matrix held as g[row][col].
// On return the columns of g are sigma_i * u_i (mutually orthogonal) and v holds the right
// singular vectors as columns, A v_i = g_i.  Relative accuracy ~eps even for the smallest
// singular vector, which is what DLT triangulation needs (src/utils.py:30-31).
// The sweep loop exits when every lane of the wave has converged (callers keep all 64 lanes
// active), but a lane stops rotating with its OWN first converged sweep: the result depends on
// the lane's matrix only, never on which other models share the wave.
// ---------------------------------------------------------------------------------------
template <typename T, int N>
SFM_FN void jacobi_svd(T (&g)[N][N], T (&v)[N][N]) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) v[i][j] = (i == j) ? T(1) : T(0);
  const T tol = T(4) * num_traits<T>::eps;
  bool done = false;
  for (int sweep = 0; sweep < 16; ++sweep) {
    T off = T(0);
#pragma unroll
    for (int p = 0; p < N - 1; ++p) {
#pragma unroll
      for (int q = p + 1; q < N; ++q) {
        T al = T(0), be = T(0), ga = T(0);
#pragma unroll
        for (int i = 0; i < N; ++i) {
          al = fma(g[i][p], g[i][p], al);
          be = fma(g[i][q], g[i][q], be);
          ga = fma(g[i][p], g[i][q], ga);
        }
        const T lim = tsqrt(al * be);
        const T aga = tabs(ga);
        const bool rot = !done && aga > num_traits<T>::eps * lim;
        if (rot) off = fmax(off, aga / lim);
        const T zeta = (be - al) / (T(2) * ga);
        T t = copysign(T(1), zeta) / (tabs(zeta) + tsqrt(fma(zeta, zeta, T(1))));
        if (!rot || !(t == t)) t = T(0);
        const T c = T(1) / tsqrt(fma(t, t, T(1)));
        const T s = c * t;
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const T gp = g[i][p], gq = g[i][q];
          g[i][p] = fma(c, gp, -(s * gq));
          g[i][q] = fma(s, gp, c * gq);
          const T vp = v[i][p], vq = v[i][q];
          v[i][p] = fma(c, vp, -(s * vq));
          v[i][q] = fma(s, vp, c * vq);
        }
      }
    }
    done = done || off <= tol;
    if (SFM_WAVE_ALL(done)) break;
  }
}

// Full 3x3 SVD, singular values descending, A = U diag(S) V^T.  U's third column is
// +-(u0 x u1), signed so that A v2 = +S2 u2: normalising the third column of A V instead would
// return rounding noise when S2 is (nearly) zero, as it is for every essential matrix.
template <typename T>
SFM_FN void svd3(const T (&A)[3][3], T (&U)[3][3], T (&S)[3], T (&V)[3][3]) {
  T g[3][3], v[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) g[i][j] = A[i][j];
  jacobi_svd<T, 3>(g, v);
  T n[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) n[j] = tsqrt(fma(g[0][j], g[0][j], fma(g[1][j], g[1][j], g[2][j] * g[2][j])));
  // sorting network on (n, column) with conditional swaps: all indices stay static
#define SFM_CSWAP(a, b)                                   \
  if (n[a] < n[b]) {                                      \
    T tn = n[a]; n[a] = n[b]; n[b] = tn;                  \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {       \
      T tg = g[i][a]; g[i][a] = g[i][b]; g[i][b] = tg;    \
      T tv = v[i][a]; v[i][a] = v[i][b]; v[i][b] = tv;    \
    }                                                     \
  }
  SFM_CSWAP(0, 1)
  SFM_CSWAP(1, 2)
  SFM_CSWAP(0, 1)
#undef SFM_CSWAP
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    S[j] = n[j];
    const T inv = T(1) / n[j];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      U[i][j] = g[i][j] * inv;
      V[i][j] = v[i][j];
    }
  }
  const T c0 = fma(U[1][0], U[2][1], -(U[2][0] * U[1][1]));
  const T c1 = fma(U[2][0], U[0][1], -(U[0][0] * U[2][1]));
  const T c2 = fma(U[0][0], U[1][1], -(U[1][0] * U[0][1]));
  const T dir = fma(c0, g[0][2], fma(c1, g[1][2], c2 * g[2][2]));
  const T sg = (dir < T(0)) ? T(-1) : T(1);
  U[0][2] = sg * c0;
  U[1][2] = sg * c1;
  U[2][2] = sg * c2;
}

template <typename T>
SFM_FN T det3(const T (&M)[3][3]) {
  return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) -
         M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
         M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}

// ---------------------------------------------------------------------------------------
// Minimal solvers.  Correspondences are (x1, y1, x2, y2) with both third coordinates 1.
// ---------------------------------------------------------------------------------------

// 8-point essential matrix: null vector of the rows kron(x1_i, x2_i)
// (src/ransac_algorithm.py:240-253), F[r][c] = v[3c + r] (:261), then the two largest
// singular values are set to one and the third to zero (:187-190).  E is row-major.
template <typename T>
SFM_FN void solve_E8(const T (&p)[8][4], T (&E)[9], T (&F)[9]) {
  T a[9][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const T x1 = p[i][0], y1 = p[i][1], x2 = p[i][2], y2 = p[i][3];
    a[0][i] = x1 * x2; a[1][i] = x1 * y2; a[2][i] = x1;
    a[3][i] = y1 * x2; a[4][i] = y1 * y2; a[5][i] = y1;
    a[6][i] = x2;      a[7][i] = y2;      a[8][i] = T(1);
  }
  T q[9];
  null_vector_9x8<T>(a, q);
  T g[3][3], v[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      g[r][c] = q[3 * c + r];
      F[3 * r + c] = q[3 * c + r];
    }
  jacobi_svd<T, 3>(g, v);
  T n[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) n[j] = fma(g[0][j], g[0][j], fma(g[1][j], g[1][j], g[2][j] * g[2][j]));
  // E = sum over the two columns with the largest norms of u_j v_j^T; the weight of the
  // smallest one is zero
  const int kmin = (n[0] <= n[1]) ? ((n[0] <= n[2]) ? 0 : 2) : ((n[1] <= n[2]) ? 1 : 2);
  T w[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) w[j] = (j == kmin) ? T(0) : T(1) / tsqrt(n[j]);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      T e = T(0);
#pragma unroll
      for (int j = 0; j < 3; ++j) e = fma(g[r][j] * w[j], v[c][j], e);
      E[3 * r + c] = e;
    }
}

template <typename T>
SFM_FN void solve_E8(const T (&p)[8][4], T (&E)[9]) {
  T F[9];
  solve_E8<T>(p, E, F);
}

// 4-point homography: null vector of the two rows per correspondence of
// src/ransac_algorithm.py:281-282, reshaped row-major and divided by H[2][2] (:290-293).
template <typename T>
SFM_FN void solve_H4(const T (&p)[4][4], T (&H)[9]) {
  T a[9][8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const T x1 = p[i][0], y1 = p[i][1], x2 = p[i][2], y2 = p[i][3];
    a[0][2 * i] = -x1;   a[1][2 * i] = -y1;   a[2][2 * i] = T(-1);
    a[3][2 * i] = T(0);  a[4][2 * i] = T(0);  a[5][2 * i] = T(0);
    a[6][2 * i] = x2 * x1; a[7][2 * i] = x2 * y1; a[8][2 * i] = x2;
    a[0][2 * i + 1] = T(0);  a[1][2 * i + 1] = T(0);  a[2][2 * i + 1] = T(0);
    a[3][2 * i + 1] = -x1;   a[4][2 * i + 1] = -y1;   a[5][2 * i + 1] = T(-1);
    a[6][2 * i + 1] = y2 * x1; a[7][2 * i + 1] = y2 * y1; a[8][2 * i + 1] = y2;
  }
  T q[9];
  null_vector_9x8<T>(a, q);
#pragma unroll
  for (int i = 0; i < 9; ++i) H[i] = q[i] / q[8];
}

// ---------------------------------------------------------------------------------------
// Residual tests (division free; equivalent to the reference's formulas for finite inputs
// and false whenever the reference's comparison involves inf/nan).
// ---------------------------------------------------------------------------------------

// Symmetric epipolar test of src/ransac_algorithm.py:98-101 with epipolar_errors :206-212:
//   (s^2/n1 + s^2/n2)/2 < thr^2,  s = x2^T E x1, n1 = |(E x1)_{0,1}|^2, n2 = |(E^T x2)_{0,1}|^2
// evaluated as  s^2 (n1+n2) < (2 thr^2) n1 n2.
template <typename T>
SFM_FN bool epipolar_inlier(const T (&E)[9], T x1, T y1, T x2, T y2, T two_thr2) {
  const T l0 = fma(E[0], x1, fma(E[1], y1, E[2]));
  const T l1 = fma(E[3], x1, fma(E[4], y1, E[5]));
  const T l2 = fma(E[6], x1, fma(E[7], y1, E[8]));
  const T s = fma(l0, x2, fma(l1, y2, l2));
  const T m0 = fma(E[0], x2, fma(E[3], y2, E[6]));
  const T m1 = fma(E[1], x2, fma(E[4], y2, E[7]));
  const T n1 = fma(l0, l0, l1 * l1);
  const T n2 = fma(m0, m0, m1 * m1);
  const T ss = s * s;
  return ss * (n1 + n2) < two_thr2 * (n1 * n2);
}

// One-way transfer error of src/ransac_algorithm.py:320-327: |x2 - pi(H x1)| < thr,
// evaluated as |x2 hz - (hx,hy)|^2 < thr^2 hz^2.
template <typename T>
SFM_FN bool transfer_inlier(const T (&H)[9], T x1, T y1, T x2, T y2, T thr2) {
  const T hx = fma(H[0], x1, fma(H[1], y1, H[2]));
  const T hy = fma(H[3], x1, fma(H[4], y1, H[5]));
  const T hz = fma(H[6], x1, fma(H[7], y1, H[8]));
  const T dx = fma(x2, hz, -hx);
  const T dy = fma(y2, hz, -hy);
  const T d2 = fma(dx, dx, dy * dy);
  return d2 < thr2 * (hz * hz);
}

// ---------------------------------------------------------------------------------------
// Float32 UPPER-BOUND forms of the two tests: true whenever the float64 test above is true
// for the unrounded inputs, given absolute error bounds of the float32 evaluation.
//
// Epipolar: the exact test is |s| < sqrt(k) g(L1, L2) with L = sqrt(n), g = L1 L2 / sqrt(L1^2+L2^2)
// (1-Lipschitz in L1 and in L2).  With |s - s_true| <= Es and |L - L_true| <= eL this implies
// |s| < sqrt(k) g + G, G = Es + sqrt(k)(eL1 + eL2); multiplying by sqrt(n1+n2), squaring and
// bounding the cross term by AM-GM with weight lambda gives
//      (s^2 - c2)(n1 + n2) < k' n1 n2,   k' = k (1 + G/lambda),  c2 = G lambda + G^2.
// Transfer: |d| < sqrt(T)|hz| with |d - d_true| <= D, |hz - hz_true| <= ez implies
//      d^2 < T hz^2 + c1 |hz| + c0,  c1 = 2 sqrt(T) G, c0 = G^2, G = sqrt(T) ez + D.
// The constants are per hypothesis (computed once from the model's row/column abs sums).
// ---------------------------------------------------------------------------------------
SFM_FN bool epipolar_upper_f32(const float (&E)[9], float x1, float y1, float x2, float y2, float kp, float c2) {
  const float l0 = fma(E[0], x1, fma(E[1], y1, E[2]));
  const float l1 = fma(E[3], x1, fma(E[4], y1, E[5]));
  const float l2 = fma(E[6], x1, fma(E[7], y1, E[8]));
  const float s = fma(l0, x2, fma(l1, y2, l2));
  const float m0 = fma(E[0], x2, fma(E[3], y2, E[6]));
  const float m1 = fma(E[1], x2, fma(E[4], y2, E[7]));
  const float n1 = fma(l0, l0, l1 * l1);
  const float n2 = fma(m0, m0, m1 * m1);
  const float ss = fma(s, s, -c2);
  return ss * (n1 + n2) < kp * (n1 * n2);
}

// Cheaper, looser upper-bound form of the epipolar test: since n2/(n1+n2) < 1 the symmetric test
// implies the one-sided  s^2 < k n1  (distance to the epipolar line in view 2 below sqrt(2) thr),
// which needs neither E^T x2 nor n2.  With the float32 error bounds (|s - s_true| <= Es,
// |L1 - L1_true| <= eL, G = Es + sqrt(k) eL) and 2ab <= mu a^2 + b^2/mu:
//      s^2 - c < k1 n1,   k1 = k (1 + mu),  c = G^2 (1 + 1/mu).
SFM_FN bool epipolar_upper_onesided_f32(const float (&E)[9], float x1, float y1, float x2, float y2, float k1, float c) {
  const float l0 = fma(E[0], x1, fma(E[1], y1, E[2]));
  const float l1 = fma(E[3], x1, fma(E[4], y1, E[5]));
  const float l2 = fma(E[6], x1, fma(E[7], y1, E[8]));
  const float s = fma(l0, x2, fma(l1, y2, l2));
  const float n1 = fma(l0, l0, l1 * l1);
  return fma(s, s, -c) < k1 * n1;
}

SFM_FN bool transfer_upper_f32(const float (&H)[9], float x1, float y1, float x2, float y2, float T, float c1, float c0) {
  const float hx = fma(H[0], x1, fma(H[1], y1, H[2]));
  const float hy = fma(H[3], x1, fma(H[4], y1, H[5]));
  const float hz = fma(H[6], x1, fma(H[7], y1, H[8]));
  const float dx = fma(x2, hz, -hx);
  const float dy = fma(y2, hz, -hy);
  const float d2 = fma(dx, dx, dy * dy);
  return d2 < fma(T, hz * hz, fma(c1, fabs(hz), c0));
}

// Upper-bound form of the reprojection test below, float32:  |d| < thr |pz| with |d - d_true| <= D and
// |pz - pz_true| <= ez implies d^2 < T pz^2 + c1 |pz| + c0 (c1 = 2 thr G, c0 = G^2, G = thr ez + D).
SFM_FN bool reprojection_upper_f32(float tx, float ty, float tz, float rx, float ry, float rz, float u, float v, float T, float c1, float c0) {
  const float px = rx + tx, py = ry + ty, pz = rz + tz;
  const float dx = fma(-u, pz, px);
  const float dy = fma(-v, pz, py);
  const float d2 = fma(dx, dx, dy * dy);
  return d2 < fma(T, pz * pz, fma(c1, fabs(pz), c0));
}

// Cheaper, slightly looser upper-bound form of the transfer test: |x2 - pi(H x1)| < thr implies that
// each component alone differs by less than thr (a box instead of a disc: 4/pi of its area), i.e.
// |x2 hz - hx| < thr |hz| and |y2 hz - hy| < thr |hz| (+ the float32 slack G): no squares.
// rt = sqrt(T) rounded up.
SFM_FN bool transfer_upper_box_f32(const float (&H)[9], float x1, float y1, float x2, float y2, float rt, float g) {
  const float hx = fma(H[0], x1, fma(H[1], y1, H[2]));
  const float hy = fma(H[3], x1, fma(H[4], y1, H[5]));
  const float hz = fma(H[6], x1, fma(H[7], y1, H[8]));
  const float dx = fma(x2, hz, -hx);
  const float dy = fma(y2, hz, -hy);
  const float lim = fma(rt, fabs(hz), g);
  return (fabs(dx) < lim) & (fabs(dy) < lim);
}

// Reprojection test of src/estimate_translation.py:123-127 with RX precomputed:
// |pi(RX + T) - x| < thr  as  |(RX+T)_{xy} - x (RX+T)_z|^2 < thr^2 (RX+T)_z^2.
template <typename T>
SFM_FN bool reprojection_inlier(T tx, T ty, T tz, T rx, T ry, T rz, T u, T v, T thr2) {
  const T px = rx + tx, py = ry + ty, pz = rz + tz;
  const T dx = fma(-u, pz, px);
  const T dy = fma(-v, pz, py);
  const T d2 = fma(dx, dx, dy * dy);
  return d2 < thr2 * (pz * pz);
}

// ---------------------------------------------------------------------------------------
// DLT triangulation of one point (src/utils.py:22-32): rows x*P[2]-P[0], y*P[2]-P[1] for both
// cameras, smallest right singular vector, divided by its fourth component.
// ---------------------------------------------------------------------------------------
template <typename T>
SFM_FN void triangulate_point(const T (&P1)[12], const T (&P2)[12], T x1, T y1, T x2, T y2, T (&X)[4]) {
  T g[4][4], v[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    g[0][c] = fma(x1, P1[8 + c], -P1[c]);
    g[1][c] = fma(y1, P1[8 + c], -P1[4 + c]);
    g[2][c] = fma(x2, P2[8 + c], -P2[c]);
    g[3][c] = fma(y2, P2[8 + c], -P2[4 + c]);
  }
  jacobi_svd<T, 4>(g, v);
  T n[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) n[j] = fma(g[0][j], g[0][j], fma(g[1][j], g[1][j], fma(g[2][j], g[2][j], g[3][j] * g[3][j])));
  T best = n[0];
  T x[4] = {v[0][0], v[1][0], v[2][0], v[3][0]};
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    const bool lt = n[j] < best;
    best = lt ? n[j] : best;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = lt ? v[i][j] : x[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) X[i] = x[i] / x[3];
}

// ---------------------------------------------------------------------------------------
// 2-point translation: least squares of [x_j]x T = -[x_j]x (R X_j), j = 0,1
// (src/estimate_translation.py:93-104) by Householder QR of the stacked 6x3 system; with a
// duplicated index (rank 2) the minimum-norm solution gelsd returns has the closed form
// T = lambda x - RX, lambda = (x . RX)/(x . x).
// ---------------------------------------------------------------------------------------
template <typename T>
SFM_FN void solve_T2(const T (&x)[2][3], const T (&r)[2][3], bool duplicate, T (&Tout)[3]) {
  if (duplicate) {
    const T xx = fma(x[0][0], x[0][0], fma(x[0][1], x[0][1], x[0][2] * x[0][2]));
    const T xr = fma(x[0][0], r[0][0], fma(x[0][1], r[0][1], x[0][2] * r[0][2]));
    const T lam = xr / xx;
#pragma unroll
    for (int i = 0; i < 3; ++i) Tout[i] = fma(lam, x[0][i], -r[0][i]);
    return;
  }
  T a[6][3], b[6];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const T X = x[j][0], Y = x[j][1], Z = x[j][2];
    a[3 * j + 0][0] = T(0); a[3 * j + 0][1] = -Z;   a[3 * j + 0][2] = Y;
    a[3 * j + 1][0] = Z;    a[3 * j + 1][1] = T(0); a[3 * j + 1][2] = -X;
    a[3 * j + 2][0] = -Y;   a[3 * j + 2][1] = X;    a[3 * j + 2][2] = T(0);
    // b = -[x]x r
    b[3 * j + 0] = -fma(-Z, r[j][1], Y * r[j][2]);
    b[3 * j + 1] = -fma(Z, r[j][0], -(X * r[j][2]));
    b[3 * j + 2] = -fma(-Y, r[j][0], X * r[j][1]);
  }
  T diag[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    T tail2 = T(0);
#pragma unroll
    for (int i = k + 1; i < 6; ++i) tail2 = fma(a[i][k], a[i][k], tail2);
    const T x0 = a[k][k];
    const T nrm = tsqrt(fma(x0, x0, tail2));
    const T v0 = (x0 >= T(0)) ? x0 + nrm : x0 - nrm;
    const T vtv = fma(v0, v0, tail2);
    const T bb = (vtv > T(0)) ? T(2) / vtv : T(0);
    diag[k] = (x0 >= T(0)) ? -nrm : nrm;
    a[k][k] = v0;
#pragma unroll
    for (int j = k + 1; j < 3; ++j) {
      T d = T(0);
#pragma unroll
      for (int i = k; i < 6; ++i) d = fma(a[i][k], a[i][j], d);
      d *= bb;
#pragma unroll
      for (int i = k; i < 6; ++i) a[i][j] = fma(-d, a[i][k], a[i][j]);
    }
    T d = T(0);
#pragma unroll
    for (int i = k; i < 6; ++i) d = fma(a[i][k], b[i], d);
    d *= bb;
#pragma unroll
    for (int i = k; i < 6; ++i) b[i] = fma(-d, a[i][k], b[i]);
  }
  // back substitution with R = [diag0 a01 a02; 0 diag1 a12; 0 0 diag2]
  Tout[2] = b[2] / diag[2];
  Tout[1] = fma(-a[1][2], Tout[2], b[1]) / diag[1];
  Tout[0] = fma(-a[0][2], Tout[2], fma(-a[0][1], Tout[1], b[0])) / diag[0];
}

// ---------------------------------------------------------------------------------------
// Motion candidates of an essential matrix (src/ransac_algorithm.py:422-435):
// SVD, Vt negated when det(U) det(Vt) < 0, then [U W Vt | +-u3], [U W^T Vt | +-u3].
// P[c] is a row-major 3x4 camera.
// ---------------------------------------------------------------------------------------
template <typename T>
SFM_FN void essential_candidates(const T (&E)[9], T (&P)[4][12]) {
  T A[3][3], U[3][3], S[3], V[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) A[i][j] = E[3 * i + j];
  svd3<T>(A, U, S, V);
  const T sgn = (det3<T>(U) * det3<T>(V) < T(0)) ? T(-1) : T(1);
  // U W has columns (u1, -u0, u2); U W^T has columns (-u1, u0, u2)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const T a = fma(U[i][1], V[j][0], fma(-U[i][0], V[j][1], U[i][2] * V[j][2])) * sgn;
      const T b = fma(-U[i][1], V[j][0], fma(U[i][0], V[j][1], U[i][2] * V[j][2])) * sgn;
      P[0][4 * i + j] = a;
      P[1][4 * i + j] = a;
      P[2][4 * i + j] = b;
      P[3][4 * i + j] = b;
    }
    P[0][4 * i + 3] = U[i][2];
    P[1][4 * i + 3] = -U[i][2];
    P[2][4 * i + 3] = U[i][2];
    P[3][4 * i + 3] = -U[i][2];
  }
}

// ---------------------------------------------------------------------------------------
// Two (R, t) solutions of a homography (src/ransac_algorithm.py:349-403).  `flip` is the
// outcome of the reference's sign vote (:359-361): fewer than half of the points have
// x2 . (H x1) > 0.  Also returns E = [t]x R for both (src/ransac_algorithm.py:129-130).
// ---------------------------------------------------------------------------------------
template <typename T>
SFM_FN void unitize(T a, T b, T& ua, T& ub) {
  const T den = T(1) / tsqrt(fma(a, a, b * b));
  ua = a * den;
  ub = b * den;
}

template <typename T>
SFM_FN void skew_times(const T (&t)[3], const T (&R)[3][3], T (&E)[9]) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    E[0 + j] = fma(-t[2], R[1][j], t[1] * R[2][j]);
    E[3 + j] = fma(t[2], R[0][j], -(t[0] * R[2][j]));
    E[6 + j] = fma(-t[1], R[0][j], t[0] * R[1][j]);
  }
}

template <typename T>
SFM_FN void homography_decompose(const T (&Hin)[9], bool flip, T (&R1)[3][3], T (&t1)[3], T (&R2)[3][3], T (&t2)[3]) {
  T A[3][3], U[3][3], S[3], V[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) A[i][j] = flip ? -Hin[3 * i + j] : Hin[3 * i + j];
  svd3<T>(A, U, S, V);
  const T s1 = S[0] / S[1];
  const T s3 = S[2] / S[1];
  const T a1 = tsqrt(T(1) - s3 * s3);
  const T b1 = tsqrt(s1 * s1 - T(1));
  T a, b, c, d, e, f;
  unitize<T>(a1, b1, a, b);
  unitize<T>(T(1) + s1 * s3, a1 * b1, c, d);
  unitize<T>(-b / s1, -a / s3, e, f);
  T n1z = fma(b, V[2][0], -(a * V[2][2]));
  T n2z = fma(b, V[2][0], a * V[2][2]);
  T ta[3], tb[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ta[i] = fma(e, V[i][0], f * V[i][2]);
    tb[i] = fma(e, V[i][0], -(f * V[i][2]));
    if (n1z < T(0)) ta[i] = -ta[i];
    if (n2z < T(0)) tb[i] = -tb[i];
  }
  // R1 = U [[c,0,d],[0,1,0],[-d,0,c]] Vt ; R2 = U [[c,0,-d],[0,1,0],[d,0,c]] Vt
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const T m10 = fma(U[i][0], c, -(U[i][2] * d)), m12 = fma(U[i][0], d, U[i][2] * c);
    const T m20 = fma(U[i][0], c, U[i][2] * d), m22 = fma(-U[i][0], d, U[i][2] * c);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      R1[i][j] = fma(m10, V[j][0], fma(U[i][1], V[j][1], m12 * V[j][2]));
      R2[i][j] = fma(m20, V[j][0], fma(U[i][1], V[j][1], m22 * V[j][2]));
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    t1[i] = fma(R1[i][0], ta[0], fma(R1[i][1], ta[1], R1[i][2] * ta[2]));
    t2[i] = fma(R2[i][0], tb[0], fma(R2[i][1], tb[1], R2[i][2] * tb[2]));
  }
}

// ---------------------------------------------------------------------------------------
// Counter-based index sets: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy
// as 1, 2, 3", SC'11).  Word w of hypothesis `id` under `seed` is
//   philox4x32_10(counter = (id_lo, id_hi, w / 4, 0), key = (seed_lo, seed_hi))[w % 4]
// and the index is the high half of word * n, so the sets are a pure function of (seed, id, n)
// and every sharding of an id range produces the same sets.
// ---------------------------------------------------------------------------------------
SFM_FN void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Fills row[0..width) with indices in [0, n) for hypothesis `id`.
template <int kWidth>
SFM_FN void philox_index_row(uint64_t seed, uint64_t id, uint32_t n, int32_t (&row)[kWidth]) {
#pragma unroll
  for (int blk = 0; blk < (kWidth + 3) / 4; ++blk) {
    uint32_t w[4];
    philox4x32_10((uint32_t)id, (uint32_t)(id >> 32), (uint32_t)blk, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * blk + j < kWidth) row[4 * blk + j] = (int32_t)(((uint64_t)w[j] * n) >> 32);
  }
}

}  // namespace sfm
