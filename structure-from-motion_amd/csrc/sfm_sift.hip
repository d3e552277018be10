// SIFT detection and description on the GPU (gfx950 / MI355X only): what the reference obtains from
// cv2.SIFT_create().detectAndCompute(gray, None) (src/extract_sift.py:104-114, src/utils.py:78-80).
//
// The algorithm is Lowe's SIFT with the default parameters of OpenCV's features2d module (base image doubled by
// cv::resize's INTER_LINEAR rule, 3 layers per octave, sigma 1.6, contrast threshold 0.04, edge threshold 10, float32
// pyramid, 36-bin orientation histograms, 4 x 4 x 8 descriptors scaled by 512 and saturated to [0, 255]).
// Everything is float32 with separate multiplies and adds (the library is built with -ffp-contract=off) in one
// fixed, documented order (ascending taps, window samples in row-major order), so that the detection stage (pyramid,
// extrema, sub-pixel refinement) is reproducible bit for bit by any IEEE float32 restatement and only expf / powf /
// cosf / sinf can differ in the last place.  That order is this file's own: OpenCV sums the taps of its symmetric
// column filter in pairs and computes the first blur's sigma in float32, so its pyramid can differ from this one in the
// last place; nothing here was compared with OpenCV's output (it is not installed where this was built).
//
// Stages and their shapes:
//   pyramid      separable Gaussian (rows and 64 x 64 column tiles staged in LDS with their reflected margins; the
//                column pass also writes the difference to the previous layer), nearest-neighbour decimation;
//   extrema      one thread per interior pixel and layer: 26-neighbour test, candidates appended through a counter;
//   refinement   one thread per candidate: quadratic fit (<= 5 steps), contrast and edge tests, orientation
//                histogram kept in LDS as [bin][lane] (no bank conflicts, sequential float32 sums per keypoint);
//   description  16 lanes per keypoint, each with its own 4 x 4 x 10 histogram in LDS as [bin][lane] (40 KB per wave)
//                over every 16th window row; the partial histograms are added in a fixed order, so descriptors do
//                not depend on how waves are scheduled.
// The order in which candidates and keypoints are appended is not deterministic; the host sorts them with OpenCV's
// keypoint order before duplicates are removed, which makes the final list deterministic.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/sfm_hip.h"

namespace {

constexpr int kLayers = 3;            // nOctaveLayers
constexpr int kGauss = kLayers + 3;   // Gaussian images per octave
constexpr int kDog = kLayers + 2;     // difference images per octave
constexpr int kMaxOctaves = 16;
constexpr int kBorder = 5;            // SIFT_IMG_BORDER
constexpr int kMaxSteps = 5;          // SIFT_MAX_INTERP_STEPS
constexpr int kOriBins = 36;
constexpr int kMaxTaps = 64;
constexpr int kBlock = 256;
constexpr float kSigma = 1.6f;
constexpr double kSigmaD = 1.6;  // the pyramid's sigmas are computed in double
constexpr float kContrastThr = 0.04f;
constexpr float kEdgeThr = 10.0f;

inline void clear_stale_error() { (void)hipGetLastError(); }
inline int check_launch() { return (int)hipGetLastError(); }

#define SIFT_TRY(expr)           \
  do {                           \
    hipError_t e_ = (expr);      \
    if (e_ != hipSuccess) return (int)e_; \
  } while (0)

struct Taps {
  float k[kMaxTaps];
  int n;
};

// Placement of everything in the caller's workspace (offsets in floats); the same function of (w, h) for
// sfm_sift_detect, which fills it, and sfm_sift_describe, which reads the Gaussian images back.
struct Plan {
  int noct;
  int W[kMaxOctaves], H[kMaxOctaves];
  long long gauss[kMaxOctaves];  // kGauss planes of W x H each
  long long dog[kMaxOctaves];    // kDog planes
  long long tmp_a, tmp_b;        // two planes of the base size
  long long cand;                // int4[cand_cap]
  long long counters;            // int[4]: candidates, keypoints
  int cand_cap;
  long long total;               // floats
};

Plan make_plan(int w, int h) {
  Plan p{};
  const int W0 = 2 * w, H0 = 2 * h;
  int noct = (int)lrint(log((double)(W0 < H0 ? W0 : H0)) / log(2.0) - 2.0) + 1;  // firstOctave = -1
  if (noct < 1) noct = 1;
  if (noct > kMaxOctaves) noct = kMaxOctaves;
  p.noct = noct;
  long long off = 0;
  int W = W0, H = H0;
  for (int o = 0; o < noct; ++o) {
    p.W[o] = W;
    p.H[o] = H;
    p.gauss[o] = off;
    off += (long long)kGauss * W * H;
    p.dog[o] = off;
    off += (long long)kDog * W * H;
    W = W / 2 > 1 ? W / 2 : 1;
    H = H / 2 > 1 ? H / 2 : 1;
  }
  p.tmp_a = off;
  off += (long long)W0 * H0;
  p.tmp_b = off;
  off += (long long)W0 * H0;
  // A pixel of a layer is a 26-neighbour extremum with probability 2 / 27 at most for any continuous noise (7.4 % of the
  // pixels): a buffer of 1/8 of the base image always holds the candidates of ONE layer, and those of a whole image of any
  // ordinary content; sfm_sift_detect falls back to one layer at a time when a whole image does not fit.
  long long cap = (long long)W0 * H0 / 8;
  if (cap < 65536) cap = 65536;
  // test hook: a smaller buffer, to exercise the layer-by-layer path on small images (read once per process)
  static const long long cap_env = getenv("SFM_HIP_SIFT_CAND_CAP") ? atoll(getenv("SFM_HIP_SIFT_CAND_CAP")) : 0;
  if (cap_env > 0) cap = cap_env;
  p.cand_cap = (int)cap;
  p.cand = off;
  off += 4 * cap;
  p.counters = off;
  off += 4;
  p.total = off;
  return p;
}

// getGaussianKernel(cvRound(sigma * 8 + 1) | 1, sigma, CV_32F)
int gaussian_taps(double sigma, Taps* t) {
  int n = ((int)lrint(sigma * 8.0 + 1.0)) | 1;
  if (n > kMaxTaps) return SFM_EINVAL;
  double sum = 0.0;
  for (int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    t->k[i] = (float)exp(-(x * x) / (2.0 * sigma * sigma));
    sum += (double)t->k[i];
  }
  const double s = 1.0 / sum;
  for (int i = 0; i < n; ++i) t->k[i] = (float)((double)t->k[i] * s);
  t->n = n;
  return 0;
}

// BORDER_REFLECT_101 for any offset (the last octaves are smaller than the kernels)
__device__ __forceinline__ int reflect101(int i, int n) {
  if ((unsigned)i < (unsigned)n) return i;  // interior: no division
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i %= p;
  if (i < 0) i += p;
  return i >= n ? p - i : i;
}

// The doubled base image as cv::resize(..., INTER_LINEAR) produces it (the branch SIFT_create() takes by default,
// enable_precise_upscale = false): pixel centres are aligned, so dst(X) samples src at X / 2 - 0.25 with the border
// replicated: even X = 2m -> 0.25 src[m-1] + 0.75 src[m], odd X = 2m+1 -> 0.75 src[m] + 0.25 src[m+1], rows first, then
// columns.  All products and sums are exact in float32 (8-bit pixels times 1/4, 3/4, 1/16 ... 9/16).
__global__ __launch_bounds__(kBlock) void upsample2x_kernel(const uint8_t* __restrict__ src, int w, int h, float* __restrict__ dst) {
  const int X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y;
  if (X >= 2 * w) return;
  const int mx = X >> 1, my = Y >> 1;
  const int x0 = (X & 1) ? mx : max(mx - 1, 0), x1 = (X & 1) ? min(mx + 1, w - 1) : mx;
  const int y0 = (Y & 1) ? my : max(my - 1, 0), y1 = (Y & 1) ? min(my + 1, h - 1) : my;
  const float ax = (X & 1) ? 0.75f : 0.25f, bx = 1.0f - ax;  // weights of x0 and x1
  const float ay = (Y & 1) ? 0.75f : 0.25f, by = 1.0f - ay;
  const float a = (float)src[(long long)y0 * w + x0] * ax + (float)src[(long long)y0 * w + x1] * bx;
  const float b = (float)src[(long long)y1 * w + x0] * ax + (float)src[(long long)y1 * w + x1] * bx;
  dst[(long long)Y * 2 * w + X] = a * ay + b * by;
}

// one workgroup = 256 consecutive pixels of one row, staged in LDS with the reflected margins
__global__ __launch_bounds__(kBlock) void blur_rows_kernel(const float* __restrict__ src, int w, int h, Taps taps, float* __restrict__ dst) {
  __shared__ float line[kBlock + kMaxTaps];
  const int y = blockIdx.y, x0 = blockIdx.x * kBlock, r = (taps.n - 1) / 2;
  const float* row = src + (long long)y * w;
  for (int i = threadIdx.x; i < kBlock + 2 * r; i += kBlock) line[i] = row[reflect101(x0 + i - r, w)];
  __syncthreads();
  const int x = x0 + threadIdx.x;
  if (x >= w) return;
  float acc = line[threadIdx.x] * taps.k[0];
  for (int t = 1; t < taps.n; ++t) acc = acc + line[threadIdx.x + t] * taps.k[t];
  dst[(long long)y * w + x] = acc;
}

// one workgroup = a tile of 64 columns x 64 rows staged in LDS with the reflected margins above and below; a thread
// owns one column and 16 consecutive rows.  When `prev` is given (the Gaussian image this blur started from) the
// difference image new - prev is written as well.
// Two forms: the windowed one (kernels of up to 27 taps, every kernel of the default scale space) pads the taps with
// zeros to 27 around the centre and keeps a thread's 16 + 26 input values in registers, so that each staged value is
// read from LDS once instead of once per tap (a zero tap adds +-0: the sums are those of the unpadded kernel bit for
// bit); the generic one reads LDS once per tap and takes any kernel size.
constexpr int kColTile = 64;
constexpr int kWinR = 13, kWinTaps = 2 * kWinR + 1;
__global__ __launch_bounds__(kBlock) void blur_cols_window_kernel(const float* __restrict__ src, int w, int h, Taps taps, float* __restrict__ dst,
                                                                  const float* __restrict__ prev, float* __restrict__ diff) {
  __shared__ float tile[kColTile + 2 * kWinR][kColTile];
  const int x0 = blockIdx.x * kColTile, y0 = blockIdx.y * kColTile;
  const int cx = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int x = x0 + cx;
  const int xs = min(x, w - 1);
  for (int i = part; i < kColTile + 2 * kWinR; i += kBlock / 64) tile[i][cx] = src[(long long)reflect101(y0 + i - kWinR, h) * w + xs];
  __syncthreads();
  if (x >= w) return;
  const int pad = kWinR - (taps.n - 1) / 2;  // zero taps on either side
  float win[16 + 2 * kWinR];
#pragma unroll
  for (int j = 0; j < 16 + 2 * kWinR; ++j) win[j] = tile[part * 16 + j][cx];
  float k[kWinTaps];
#pragma unroll
  for (int t = 0; t < kWinTaps; ++t) k[t] = (t >= pad && t < pad + taps.n) ? taps.k[t - pad] : 0.0f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int y = y0 + part * 16 + j;
    float acc = win[j] * k[0];
#pragma unroll
    for (int t = 1; t < kWinTaps; ++t) acc = acc + win[j + t] * k[t];
    if (y < h) {
      const long long p = (long long)y * w + x;
      dst[p] = acc;
      if (diff) diff[p] = acc - prev[p];
    }
  }
}

__global__ __launch_bounds__(kBlock) void blur_cols_kernel(const float* __restrict__ src, int w, int h, Taps taps, float* __restrict__ dst,
                                                           const float* __restrict__ prev, float* __restrict__ diff) {
  __shared__ float tile[kColTile + kMaxTaps][kColTile];
  const int r = (taps.n - 1) / 2;
  const int x0 = blockIdx.x * kColTile, y0 = blockIdx.y * kColTile;
  const int cx = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int x = x0 + cx;
  const int xs = min(x, w - 1);
  for (int i = part; i < kColTile + 2 * r; i += kBlock / 64) tile[i][cx] = src[(long long)reflect101(y0 + i - r, h) * w + xs];
  __syncthreads();
  if (x >= w) return;
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const int ly = part * 16 + j, y = y0 + ly;
    if (y >= h) break;
    float acc = tile[ly][cx] * taps.k[0];
    for (int t = 1; t < taps.n; ++t) acc = acc + tile[ly + t][cx] * taps.k[t];
    const long long p = (long long)y * w + x;
    dst[p] = acc;
    if (diff) diff[p] = acc - prev[p];
  }
}

// resize(src, Size(W / 2, H / 2), INTER_NEAREST)
__global__ __launch_bounds__(kBlock) void downsample2_kernel(const float* __restrict__ src, int W, int H, int w, int h, float* __restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const int sx = min((int)floor((double)x * ((double)W / (double)w)), W - 1);
  const int sy = min((int)floor((double)y * ((double)H / (double)h)), H - 1);
  dst[(long long)y * w + x] = src[(long long)sy * W + sx];
}

// grid = (x tiles, interior rows, layers 1..kLayers); candidate = (octave, layer, r, c)
__global__ __launch_bounds__(kBlock) void extrema_kernel(const float* __restrict__ dog, int w, int h, int octave, int first_layer, float thr,
                                                         int4* __restrict__ cand, int cand_cap, int* __restrict__ counters) {
  const int c = kBorder + blockIdx.x * blockDim.x + threadIdx.x;
  const int r = kBorder + blockIdx.y;
  const int layer = first_layer + blockIdx.z;
  if (c >= w - kBorder) return;
  const long long plane = (long long)w * h;
  const float* cur = dog + layer * plane;
  const float val = cur[(long long)r * w + c];
  if (!(fabsf(val) > thr)) return;
  bool is_max = val > 0.0f, is_min = val < 0.0f;
#pragma unroll
  for (int dl = -1; dl <= 1; ++dl) {
    const float* img = cur + dl * plane;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const float nb = img[(long long)(r + dy) * w + (c + dx)];
        is_max = is_max && val >= nb;
        is_min = is_min && val <= nb;
      }
  }
  if (is_max || is_min) {
    const int slot = atomicAdd(counters, 1);
    if (slot < cand_cap) cand[slot] = make_int4(octave, layer, r, c);
  }
}

// Matx33f::solve(b, DECOMP_LU) of OpenCV: Cramer's rule, zero vector for a singular matrix
__device__ __forceinline__ void solve3(const float (&a)[3][3], const float (&b)[3], float (&x)[3]) {
  const float m00 = a[1][1] * a[2][2] - a[1][2] * a[2][1];
  const float m01 = a[1][0] * a[2][2] - a[1][2] * a[2][0];
  const float m02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
  float d = (a[0][0] * m00 - a[0][1] * m01) + a[0][2] * m02;
  if (d == 0.0f) {
    x[0] = x[1] = x[2] = 0.0f;
    return;
  }
  d = 1.0f / d;
  const float t0 = b[1] * a[2][2] - a[1][2] * b[2];
  const float t1 = b[1] * a[2][1] - a[1][1] * b[2];
  const float t2 = a[1][0] * b[2] - b[1] * a[2][0];
  x[0] = d * ((b[0] * m00 - a[0][1] * t0) + a[0][2] * t1);
  x[1] = d * ((a[0][0] * t0 - b[0] * m01) + a[0][2] * t2);
  x[2] = d * ((a[0][0] * (-t1) - a[0][1] * t2) + b[0] * m02);
}

struct Derivs {
  float dD[3], dxx, dyy, dss, dxy, dxs, dys;
};

__device__ __forceinline__ Derivs derivatives(const float* __restrict__ dog, long long plane, int w, int layer, int r, int c) {
  const float* img = dog + layer * plane;
  const float* prev = img - plane;
  const float* next = img + plane;
  const long long p = (long long)r * w + c;
  const float img_scale = (float)(1.0 / 255.0);
  const float ds = img_scale * 0.5f, sds = img_scale, cds = img_scale * 0.25f;
  Derivs d;
  d.dD[0] = (img[p + 1] - img[p - 1]) * ds;
  d.dD[1] = (img[p + w] - img[p - w]) * ds;
  d.dD[2] = (next[p] - prev[p]) * ds;
  const float v2 = img[p] * 2.0f;
  d.dxx = ((img[p + 1] + img[p - 1]) - v2) * sds;
  d.dyy = ((img[p + w] + img[p - w]) - v2) * sds;
  d.dss = ((next[p] + prev[p]) - v2) * sds;
  d.dxy = (((img[p + w + 1] - img[p + w - 1]) - img[p - w + 1]) + img[p - w - 1]) * cds;
  d.dxs = (((next[p + 1] - next[p - 1]) - prev[p + 1]) + prev[p - 1]) * cds;
  d.dys = (((next[p + w] - next[p - w]) - prev[p + w]) + prev[p - w]) * cds;
  return d;
}

// OpenCV's fastAtan2: degrees in [0, 360), odd polynomial of degree 7
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = (float)(0.9997878412794807 * (180.0 / M_PI)), p3 = (float)(-0.3258083974640975 * (180.0 / M_PI));
  const float p5 = (float)(0.1555786518463281 * (180.0 / M_PI)), p7 = (float)(-0.04432655554792128 * (180.0 / M_PI));
  const float ax = fabsf(x), ay = fabsf(y);
  const float big = fmaxf(ax, ay), small = fminf(ax, ay);
  const float c = small / (big + 2.220446049250313e-16f);
  const float c2 = c * c;
  float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  if (!(ax >= ay)) a = 90.0f - a;
  if (x < 0.0f) a = 180.0f - a;
  if (y < 0.0f) a = 360.0f - a;
  return a;
}

struct PlanDev {
  int W[kMaxOctaves], H[kMaxOctaves];
  long long gauss[kMaxOctaves], dog[kMaxOctaves];
};

// keypoint record: x, y, size, angle, response, packed octave (integer bits)
__device__ __forceinline__ void emit_keypoint(float* __restrict__ kps, int cap, int* __restrict__ counters, float x, float y, float size,
                                              float angle, float response, int packed) {
  const int slot = atomicAdd(counters + 1, 1);
  if (slot >= cap) return;
  float* k = kps + 6LL * slot;
  k[0] = x;
  k[1] = y;
  k[2] = size;
  k[3] = angle;
  k[4] = response;
  k[5] = __int_as_float(packed);
}

// one thread per candidate; 64 threads per workgroup, orientation histograms in LDS as [bin][lane]
__global__ __launch_bounds__(64) void refine_orient_kernel(const float* __restrict__ ws, PlanDev plan, const int4* __restrict__ cand, int cand_cap,
                                                           int* __restrict__ counters, float* __restrict__ kps, int kp_cap) {
  __shared__ float temphist[kOriBins][64];
  __shared__ float hist[kOriBins][64];
  const int lane = threadIdx.x;
  const int i = blockIdx.x * 64 + lane;
  const int ncand = min(counters[0], cand_cap);
  if (i >= ncand) return;
  const int4 cd = cand[i];
  const int octv = cd.x;
  int layer = cd.y, r = cd.z, c = cd.w;
  const int w = plan.W[octv], h = plan.H[octv];
  const long long plane = (long long)w * h;
  const float* dog = ws + plan.dog[octv];
  float xi = 0.0f, xr = 0.0f, xc = 0.0f;
  int it = 0;
  for (; it < kMaxSteps; ++it) {
    const Derivs d = derivatives(dog, plane, w, layer, r, c);
    const float Hm[3][3] = {{d.dxx, d.dxy, d.dxs}, {d.dxy, d.dyy, d.dys}, {d.dxs, d.dys, d.dss}};
    float X[3];
    solve3(Hm, d.dD, X);
    xi = -X[2];
    xr = -X[1];
    xc = -X[0];
    if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
    const float big = (float)(2147483647 / 3);
    if (fabsf(xi) > big || fabsf(xr) > big || fabsf(xc) > big) return;
    c += (int)rintf(xc);
    r += (int)rintf(xr);
    layer += (int)rintf(xi);
    if (layer < 1 || layer > kLayers || c < kBorder || c >= w - kBorder || r < kBorder || r >= h - kBorder) return;
  }
  if (it >= kMaxSteps) return;
  float response;
  {
    const Derivs d = derivatives(dog, plane, w, layer, r, c);
    const float t = (d.dD[0] * xc + d.dD[1] * xr) + d.dD[2] * xi;
    const float contr = dog[layer * plane + (long long)r * w + c] * (float)(1.0 / 255.0) + t * 0.5f;
    if (fabsf(contr) * (float)kLayers < kContrastThr) return;
    const float tr = d.dxx + d.dyy;
    const float det = d.dxx * d.dyy - d.dxy * d.dxy;
    if (det <= 0.0f || (tr * tr) * kEdgeThr >= ((kEdgeThr + 1.0f) * (kEdgeThr + 1.0f)) * det) return;
    response = fabsf(contr);
  }
  const float oscale = (float)(1 << octv);
  const float kx = ((float)c + xc) * oscale, ky = ((float)r + xr) * oscale;
  const int packed = octv + (layer << 8) + ((int)rintf((xi + 0.5f) * 255.0f) << 16);
  const float size = ((kSigma * powf(2.0f, ((float)layer + xi) / (float)kLayers)) * oscale) * 2.0f;
  // orientation histogram on the Gaussian image of the keypoint's layer (calcOrientationHist)
  const float scl_octv = (size * 0.5f) / oscale;
  const int radius = (int)rintf(4.5f * scl_octv);
  const float sigma = 1.5f * scl_octv;
  const float expf_scale = -1.0f / ((2.0f * sigma) * sigma);
  const float* img = ws + plan.gauss[octv] + layer * plane;
  for (int b = 0; b < kOriBins; ++b) temphist[b][lane] = 0.0f;
  for (int di = -radius; di <= radius; ++di) {
    const int y = r + di;
    if (y <= 0 || y >= h - 1) continue;
    for (int dj = -radius; dj <= radius; ++dj) {
      const int x = c + dj;
      if (x <= 0 || x >= w - 1) continue;
      const long long p = (long long)y * w + x;
      const float dx = img[p + 1] - img[p - 1];
      const float dy = img[p - w] - img[p + w];
      const float wgt = expf((float)(di * di + dj * dj) * expf_scale);
      const float ori = fast_atan2_deg(dy, dx);
      const float mag = sqrtf(dx * dx + dy * dy);
      int bin = (int)rintf(ori * (float)(kOriBins / 360.0));
      if (bin >= kOriBins) bin -= kOriBins;
      if (bin < 0) bin += kOriBins;
      temphist[bin][lane] = temphist[bin][lane] + wgt * mag;
    }
  }
  float omax = 0.0f;
  for (int b = 0; b < kOriBins; ++b) {
    const int m2 = (b + kOriBins - 2) % kOriBins, m1 = (b + kOriBins - 1) % kOriBins, p1 = (b + 1) % kOriBins, p2 = (b + 2) % kOriBins;
    const float v = ((temphist[m2][lane] + temphist[p2][lane]) * (1.0f / 16.0f) + (temphist[m1][lane] + temphist[p1][lane]) * (4.0f / 16.0f)) +
                    temphist[b][lane] * (6.0f / 16.0f);
    hist[b][lane] = v;
    omax = b == 0 ? v : fmaxf(omax, v);
  }
  const float mag_thr = omax * 0.8f;
  for (int j = 0; j < kOriBins; ++j) {
    const float left = hist[(j + kOriBins - 1) % kOriBins][lane], right = hist[(j + 1) % kOriBins][lane], mid = hist[j][lane];
    if (mid > left && mid > right && mid >= mag_thr) {
      float bin = (float)j + (0.5f * (left - right)) / ((left - 2.0f * mid) + right);
      bin = bin < 0.0f ? (float)kOriBins + bin : (bin >= (float)kOriBins ? bin - (float)kOriBins : bin);
      float angle = 360.0f - (float)(360.0 / kOriBins) * bin;
      if (fabsf(angle - 360.0f) < 1.1920929e-07f) angle = 0.0f;
      emit_keypoint(kps, kp_cap, counters, kx, ky, size, angle, response, packed);
    }
  }
}

// 16 lanes per keypoint (final list, image coordinates), 4 keypoints per wave.  Lane `sub` of a keypoint accumulates
// the window rows di = -radius + sub, + 16, ... into its own 4 x 4 x 10 histogram (LDS, [bin][lane]); the 16 partial
// histograms are then added in lane order, lane `sub` finishing cell `sub` of the 4 x 4 grid.  Every sum has a fixed
// order, so a descriptor does not depend on how waves are scheduled.
__global__ __launch_bounds__(64) void describe_kernel(const float* __restrict__ ws, PlanDev plan, const float* __restrict__ kps, int n,
                                                      uint8_t* __restrict__ desc) {
  constexpr int d = 4, nb = 8, kCell = nb + 2, kSub = 16;
  __shared__ float hist[d * d * kCell][64];
  __shared__ float part[64];
  const int lane = threadIdx.x, sub = lane & (kSub - 1), group = lane - sub;
  const int i = blockIdx.x * (64 / kSub) + lane / kSub;
  const bool valid = i < n;
  for (int b = 0; b < d * d * kCell; ++b) hist[b][lane] = 0.0f;
  if (valid) {
    const float* k = kps + 6LL * i;
    const int packed = __float_as_int(k[5]);
    int octave = packed & 255;
    const int layer = (packed >> 8) & 255;
    octave = octave < 128 ? octave : (-128 | octave);
    const float scale = octave >= 0 ? 1.0f / (float)(1 << octave) : (float)(1 << -octave);
    const int o = octave + 1;  // firstOctave = -1
    const int w = plan.W[o], h = plan.H[o];
    const float* img = ws + plan.gauss[o] + (long long)layer * w * h;
    const float ptx = k[0] * scale, pty = k[1] * scale;
    float ori = 360.0f - k[3];
    if (fabsf(ori - 360.0f) < 1.1920929e-07f) ori = 0.0f;
    const float scl = (k[2] * scale) * 0.5f;
    const int px = (int)rintf(ptx), py = (int)rintf(pty);
    float cos_t = (float)cos((double)ori * (M_PI / 180.0)), sin_t = (float)sin((double)ori * (M_PI / 180.0));
    const float bins_per_rad = (float)(nb / 360.0);
    const float exp_scale = -1.0f / (float)(d * d * 0.5);
    const float hist_width = 3.0f * scl;
    int radius = (int)lrint((double)hist_width * 1.4142135623730951 * (d + 1) * 0.5);
    radius = min(radius, (int)sqrt((double)w * w + (double)h * h));
    cos_t = cos_t / hist_width;
    sin_t = sin_t / hist_width;
    for (int di = -radius + sub; di <= radius; di += kSub)
      for (int dj = -radius; dj <= radius; ++dj) {
        const float c_rot = (float)dj * cos_t - (float)di * sin_t;
        const float r_rot = (float)dj * sin_t + (float)di * cos_t;
        float rbin = (r_rot + (float)(d / 2)) - 0.5f;
        float cbin = (c_rot + (float)(d / 2)) - 0.5f;
        const int r = py + di, c = px + dj;
        if (!(rbin > -1.0f && rbin < (float)d && cbin > -1.0f && cbin < (float)d && r > 0 && r < h - 1 && c > 0 && c < w - 1)) continue;
        const long long p = (long long)r * w + c;
        const float dx = img[p + 1] - img[p - 1];
        const float dy = img[p - w] - img[p + w];
        const float wgt = expf((c_rot * c_rot + r_rot * r_rot) * exp_scale);
        const float ang = fast_atan2_deg(dy, dx);
        const float mag = sqrtf(dx * dx + dy * dy) * wgt;
        float obin = (ang - ori) * bins_per_rad;
        const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
        int o0 = (int)floorf(obin);
        rbin -= (float)r0;
        cbin -= (float)c0;
        obin -= (float)o0;
        if (o0 < 0) o0 += nb;
        if (o0 >= nb) o0 -= nb;
        const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
        const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
        const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
        const float v111 = v_rc11 * obin, v110 = v_rc11 - v111;
        const float v101 = v_rc10 * obin, v100 = v_rc10 - v101;
        const float v011 = v_rc01 * obin, v010 = v_rc01 - v011;
        const float v001 = v_rc00 * obin, v000 = v_rc00 - v001;
        // cells (r0, c0) .. (r0 + 1, c0 + 1) with r0, c0 in [-1, d - 1]; the cells outside [0, d) never reach the descriptor
        const bool ra = r0 >= 0, rb = r0 + 1 < d, ca = c0 >= 0, cb = c0 + 1 < d;
        const int base = (r0 * d + c0) * kCell + o0;
        if (ra && ca) {
          hist[base][lane] = hist[base][lane] + v000;
          hist[base + 1][lane] = hist[base + 1][lane] + v001;
        }
        if (ra && cb) {
          hist[base + kCell][lane] = hist[base + kCell][lane] + v010;
          hist[base + kCell + 1][lane] = hist[base + kCell + 1][lane] + v011;
        }
        if (rb && ca) {
          hist[base + d * kCell][lane] = hist[base + d * kCell][lane] + v100;
          hist[base + d * kCell + 1][lane] = hist[base + d * kCell + 1][lane] + v101;
        }
        if (rb && cb) {
          hist[base + (d + 1) * kCell][lane] = hist[base + (d + 1) * kCell][lane] + v110;
          hist[base + (d + 1) * kCell + 1][lane] = hist[base + (d + 1) * kCell + 1][lane] + v111;
        }
      }
  }
  __syncthreads();
  // lane `sub` finishes cell `sub`: the 16 partial histograms in lane order, then the orientation bins nb and nb + 1
  // wrap onto 0 and 1
  float cell[kCell];
#pragma unroll
  for (int b = 0; b < kCell; ++b) {
    float v = hist[sub * kCell + b][group];
    for (int l = 1; l < kSub; ++l) v = v + hist[sub * kCell + b][group + l];
    cell[b] = v;
  }
  cell[0] = cell[0] + cell[nb];
  cell[1] = cell[1] + cell[nb + 1];
  // normalise, clip at 0.2, renormalise, scale by 512, saturate; squared norms: per cell, then over the cells in order
  auto group_sum = [&](float mine) {
    __syncthreads();
    part[lane] = mine;
    __syncthreads();
    float t = part[group];
    for (int l = 1; l < kSub; ++l) t = t + part[group + l];
    return t;
  };
  float sq = 0.0f;
#pragma unroll
  for (int b = 0; b < nb; ++b) sq = sq + cell[b] * cell[b];
  const float thr = sqrtf(group_sum(sq)) * 0.2f;
  sq = 0.0f;
#pragma unroll
  for (int b = 0; b < nb; ++b) {
    cell[b] = fminf(cell[b], thr);
    sq = sq + cell[b] * cell[b];
  }
  const float s = 512.0f / fmaxf(sqrtf(group_sum(sq)), 1.1920929e-07f);
  if (!valid) return;
  uint8_t* out = desc + 128LL * i + sub * nb;
#pragma unroll
  for (int b = 0; b < nb; ++b) out[b] = (uint8_t)fminf(fmaxf(rintf(cell[b] * s), 0.0f), 255.0f);
}

PlanDev to_dev(const Plan& p) {
  PlanDev d{};
  for (int o = 0; o < p.noct; ++o) {
    d.W[o] = p.W[o];
    d.H[o] = p.H[o];
    d.gauss[o] = p.gauss[o];
    d.dog[o] = p.dog[o];
  }
  return d;
}

inline unsigned div_up(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

int launch_blur(const float* src, float* tmp, float* dst, int w, int h, double sigma, float* diff, hipStream_t st) {
  Taps taps;
  const int rc = gaussian_taps(sigma, &taps);
  if (rc) return rc;
  blur_rows_kernel<<<dim3(div_up(w, kBlock), h), kBlock, 0, st>>>(src, w, h, taps, tmp);
  const dim3 grid(div_up(w, kColTile), div_up(h, kColTile));
  if (taps.n <= kWinTaps)
    blur_cols_window_kernel<<<grid, kBlock, 0, st>>>(tmp, w, h, taps, dst, diff ? src : nullptr, diff);
  else
    blur_cols_kernel<<<grid, kBlock, 0, st>>>(tmp, w, h, taps, dst, diff ? src : nullptr, diff);
  return check_launch();
}

}  // namespace

extern "C" {

size_t sfm_sift_workspace_bytes(int32_t w, int32_t h) {
  if (w < 1 || h < 1 || (long long)w * h > (1LL << 28)) return 0;
  return (size_t)make_plan(w, h).total * sizeof(float);
}

int sfm_sift_gaussian_taps(double sigma, float* taps, int32_t* ksize) {
  if (!taps || !ksize || !(sigma > 0.0)) return SFM_EINVAL;
  Taps t;
  const int rc = gaussian_taps(sigma, &t);
  if (rc) return rc;
  for (int i = 0; i < t.n; ++i) taps[i] = t.k[i];
  *ksize = t.n;
  return 0;
}

int sfm_sift_octaves(int32_t w, int32_t h, int32_t* noct, int32_t* widths, int32_t* heights, int64_t* gauss_offsets) {
  if (w < 1 || h < 1 || !noct) return SFM_EINVAL;
  const Plan p = make_plan(w, h);
  *noct = p.noct;
  for (int o = 0; o < p.noct; ++o) {
    if (widths) widths[o] = p.W[o];
    if (heights) heights[o] = p.H[o];
    if (gauss_offsets) gauss_offsets[o] = p.gauss[o];
  }
  return 0;
}

int sfm_sift_detect(const uint8_t* gray, int32_t w, int32_t h, void* workspace, size_t workspace_bytes, float* keypoints, int32_t capacity,
                    int32_t* count, void* stream) {
  clear_stale_error();
  if (!gray || !workspace || !keypoints || !count || w < 1 || h < 1 || capacity < 1 || (long long)w * h > (1LL << 28)) return SFM_EINVAL;
  const Plan p = make_plan(w, h);
  if (workspace_bytes < (size_t)p.total * sizeof(float)) return SFM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  int* counters = (int*)(ws + p.counters);
  int4* cand = (int4*)(ws + p.cand);
  SIFT_TRY(hipMemsetAsync(counters, 0, 16, st));
  const int W0 = p.W[0], H0 = p.H[0];
  upsample2x_kernel<<<dim3(div_up(W0, kBlock), H0), kBlock, 0, st>>>(gray, w, h, ws + p.tmp_a);
  SIFT_TRY((hipError_t)check_launch());
  // the input is taken to carry a blur of 0.5 (1.0 after doubling): bring the base image to sigma
  double sig_diff2 = kSigmaD * kSigmaD - 0.5 * 0.5 * 4.0;
  if (sig_diff2 < 0.01) sig_diff2 = 0.01;
  int rc = launch_blur(ws + p.tmp_a, ws + p.tmp_b, ws + p.gauss[0], W0, H0, sqrt(sig_diff2), nullptr, st);
  if (rc) return rc;
  double sig[kGauss];
  sig[0] = kSigmaD;
  const double kf = pow(2.0, 1.0 / kLayers);
  for (int i = 1; i < kGauss; ++i) {
    const double prev = pow(kf, (double)(i - 1)) * kSigmaD, total = prev * kf;
    sig[i] = sqrt(total * total - prev * prev);
  }
  const float thr = (float)floor(0.5 * kContrastThr / kLayers * 255.0);
  for (int o = 0; o < p.noct; ++o) {
    const int W = p.W[o], H = p.H[o];
    const long long plane = (long long)W * H;
    float* g = ws + p.gauss[o];
    float* dg = ws + p.dog[o];
    if (o > 0) {
      const float* src = ws + p.gauss[o - 1] + (long long)kLayers * p.W[o - 1] * p.H[o - 1];
      downsample2_kernel<<<dim3(div_up(W, kBlock), H), kBlock, 0, st>>>(src, p.W[o - 1], p.H[o - 1], W, H, g);
      SIFT_TRY((hipError_t)check_launch());
    }
    for (int i = 1; i < kGauss; ++i) {  // Gaussian image i and the difference image i - 1 = G[i] - G[i - 1] in one pass
      rc = launch_blur(g + (i - 1) * plane, ws + p.tmp_b, g + i * plane, W, H, sig[i], dg + (i - 1) * plane, st);
      if (rc) return rc;
    }
    if (W > 2 * kBorder && H > 2 * kBorder) {
      extrema_kernel<<<dim3(div_up(W - 2 * kBorder, kBlock), H - 2 * kBorder, kLayers), kBlock, 0, st>>>(dg, W, H, o, 1, thr, cand, p.cand_cap, counters);
      SIFT_TRY((hipError_t)check_launch());
    }
  }
  int host_counts[4] = {0, 0, 0, 0};
  SIFT_TRY(hipMemcpyAsync(host_counts, counters, 16, hipMemcpyDeviceToHost, st));
  SIFT_TRY(hipStreamSynchronize(st));
  if (host_counts[0] <= p.cand_cap) {
    if (host_counts[0] > 0) {
      refine_orient_kernel<<<div_up(host_counts[0], 64), 64, 0, st>>>(ws, to_dev(p), cand, p.cand_cap, counters, keypoints, capacity);
      SIFT_TRY((hipError_t)check_launch());
    }
  } else {
    // more extrema than the candidate buffer holds (a noise image): the pyramid is complete, so the extrema are found again
    // and refined one layer at a time; the keypoint counter keeps running across the layers
    for (int o = 0; o < p.noct; ++o) {
      const int W = p.W[o], H = p.H[o];
      if (!(W > 2 * kBorder && H > 2 * kBorder)) continue;
      for (int layer = 1; layer <= kLayers; ++layer) {
        SIFT_TRY(hipMemsetAsync(counters, 0, 4, st));
        extrema_kernel<<<dim3(div_up(W - 2 * kBorder, kBlock), H - 2 * kBorder, 1), kBlock, 0, st>>>(ws + p.dog[o], W, H, o, layer, thr, cand, p.cand_cap,
                                                                                                   counters);
        SIFT_TRY((hipError_t)check_launch());
        SIFT_TRY(hipMemcpyAsync(host_counts, counters, 16, hipMemcpyDeviceToHost, st));
        SIFT_TRY(hipStreamSynchronize(st));
        if (host_counts[0] > p.cand_cap) return SFM_ETOOLARGE;  // cannot happen for a buffer of 1/8 of the base image (see make_plan)
        if (host_counts[0] > 0) {
          refine_orient_kernel<<<div_up(host_counts[0], 64), 64, 0, st>>>(ws, to_dev(p), cand, p.cand_cap, counters, keypoints, capacity);
          SIFT_TRY((hipError_t)check_launch());
        }
      }
    }
  }
  SIFT_TRY(hipMemcpyAsync(host_counts, counters, 16, hipMemcpyDeviceToHost, st));
  SIFT_TRY(hipStreamSynchronize(st));
  *count = host_counts[1];
  return host_counts[1] > capacity ? SFM_ETOOLARGE : 0;
}

int sfm_sift_describe(const void* workspace, int32_t w, int32_t h, const float* keypoints, int32_t n, uint8_t* descriptors, void* stream) {
  clear_stale_error();
  if (!workspace || w < 1 || h < 1 || n < 0 || (n > 0 && (!keypoints || !descriptors))) return SFM_EINVAL;
  if (n == 0) return 0;
  const Plan p = make_plan(w, h);
  describe_kernel<<<div_up(n, 4), 64, 0, (hipStream_t)stream>>>((const float*)workspace, to_dev(p), keypoints, n, descriptors);
  return check_launch();
}

}  // extern "C"
