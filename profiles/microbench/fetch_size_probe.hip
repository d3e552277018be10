// What rocprofv3's FETCH_SIZE reports on an MI355X for the read patterns of the scoring kernels, against byte counts that
// are known exactly: every kernel below reads a 192 MiB buffer (larger than L2; the pass is preceded by a 512 MiB sweep of
// another buffer so that nothing is left on the die) once, in a different shape.  Run under
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_size_probe        and        rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- ./fetch_size_probe
// and compare the per-kernel counter with the bytes each kernel prints.
// Build: hipcc -O3 --offload-arch=gfx950 -o fetch_size_probe fetch_size_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// 16 bytes per lane, consecutive lanes consecutive: the calibrated case (128-byte requests)
__global__ void read_16B_coalesced(const uint4* __restrict__ p, long long n16, unsigned* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned a = 0;
  if (i < n16) { const uint4 v = p[i]; a = v.x ^ v.y ^ v.z ^ v.w; }
  if (a == 0x12345678u) out[0] = a;
}
// rows of 48 bytes, one row per lane, ALL three 16-byte parts (the fused kernels' index rows)
__global__ void read_48B_rows_all(const uint4* __restrict__ p, long long rows, unsigned* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned a = 0;
  if (i < rows) { const uint4 u = p[3 * i], v = p[3 * i + 1], w = p[3 * i + 2]; a = u.x ^ v.y ^ w.z; }
  if (a == 0x12345678u) out[0] = a;
}
// the same rows, first 32 bytes only (an E-only launch)
__global__ void read_48B_rows_first32(const uint4* __restrict__ p, long long rows, unsigned* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned a = 0;
  if (i < rows) { const uint4 u = p[3 * i], v = p[3 * i + 1]; a = u.x ^ v.y; }
  if (a == 0x12345678u) out[0] = a;
}
// the same rows, last 16 bytes only (an H-only launch)
__global__ void read_48B_rows_last16(const uint4* __restrict__ p, long long rows, unsigned* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned a = 0;
  if (i < rows) { const uint4 w = p[3 * i + 2]; a = w.z; }
  if (a == 0x12345678u) out[0] = a;
}
// 8 bytes per lane, consecutive (the translation kernels' index pairs)
__global__ void read_8B_coalesced(const uint2* __restrict__ p, long long n8, unsigned* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned a = 0;
  if (i < n8) { const uint2 v = p[i]; a = v.x ^ v.y; }
  if (a == 0x12345678u) out[0] = a;
}
// 4 bytes per lane, consecutive
__global__ void read_4B_coalesced(const unsigned* __restrict__ p, long long n4, unsigned* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned a = 0;
  if (i < n4) a = p[i];
  if (a == 0x12345678u) out[0] = a;
}
// 32 bytes per lane at a pseudo-random 32-byte-aligned place of the buffer (a gathered correspondence)
__global__ void read_32B_gather(const uint4* __restrict__ p, long long n32, long long count, unsigned* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned a = 0;
  if (i < count) {
    const unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
    const long long at = (long long)((h >> 20) % (unsigned long long)n32);
    const uint4 u = p[2 * at], v = p[2 * at + 1];
    a = u.x ^ v.w;
  }
  if (a == 0x12345678u) out[0] = a;
}
__global__ void sweep(const uint4* __restrict__ p, long long n16, unsigned* out) {
  unsigned a = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) a ^= p[i].x;
  if (a == 0x12345678u) out[0] = a;
}

int main() {
  const long long bytes = 192ll << 20, flush_bytes = 512ll << 20;
  uint4 *buf, *flush;
  unsigned* out;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&flush, flush_bytes));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipMemset(flush, 2, flush_bytes));
  const int B = 256;
  auto grid = [&](long long n) { return (unsigned)((n + B - 1) / B); };
#define FLUSH() sweep<<<4096, 256>>>(flush, flush_bytes / 16, out)
  for (int rep = 0; rep < 2; ++rep) {
    const long long n16 = bytes / 16, rows = bytes / 48, n8 = bytes / 8, n4 = bytes / 4, n32 = bytes / 32, gathers = 1 << 20;
    FLUSH(); read_16B_coalesced<<<grid(n16), B>>>(buf, n16, out);
    FLUSH(); read_48B_rows_all<<<grid(rows), B>>>(buf, rows, out);
    FLUSH(); read_48B_rows_first32<<<grid(rows), B>>>(buf, rows, out);
    FLUSH(); read_48B_rows_last16<<<grid(rows), B>>>(buf, rows, out);
    FLUSH(); read_8B_coalesced<<<grid(n8), B>>>((const uint2*)buf, n8, out);
    FLUSH(); read_4B_coalesced<<<grid(n4), B>>>((const unsigned*)buf, n4, out);
    FLUSH(); read_32B_gather<<<grid(gathers), B>>>(buf, n32, gathers, out);
    CHECK(hipDeviceSynchronize());
    if (rep == 0)
      printf("bytes a kernel must fetch: read_16B_coalesced %lld, read_48B_rows_all %lld, read_48B_rows_first32 %lld useful of %lld touched lines, "
             "read_48B_rows_last16 %lld useful of %lld, read_8B_coalesced %lld, read_4B_coalesced %lld, read_32B_gather %lld (x 32 B; 64-byte sectors %lld, 128-byte lines %lld)\n",
             bytes, rows * 48, rows * 32, rows * 48, rows * 16, rows * 48, bytes, bytes, gathers * 32, gathers * 64, gathers * 128);
  }
  return 0;
}
