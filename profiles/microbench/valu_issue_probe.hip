// What one SIMD of an MI355X sustains in float32 vector instructions, by operand pattern and by waves per SIMD.
// Every wave runs its body until a common deadline (s_memrealtime) and counts iterations, so there is no tail: all W
// waves of a SIMD are resident and issuing for the whole measurement (one workgroup of 4 W waves per CU).
// Bodies are written with explicit registers, 64 instructions each, to control which VGPR banks (register number mod 4)
// the three source operands of an instruction come from:
//   0  v_fma_f32   d, a, b, d     a, b, d in three different banks
//   1  v_fma_f32   d, a, b, d     a, b and d in the SAME bank
//   2  v_fma_f32   d, a, b, d     a and d in the same bank, b in another
//   3  v_fma_f32   d, a, a, d     one register twice, d in another bank
//   4  v_fmac_f32  d, a, b        (VOP2 encoding, 4 bytes) three different banks
//   5  v_alignbit_b32 d, d, a, 31 a in another bank
//   6  v_min3_f32  d, d, |a|, |b| three different banks
//   7  v_fma_f32   d, -a, a, -c   (modifiers) a, c in different banks
//   8  v_mul_f32   d, a, b        two sources, different banks
//   9  v_mul_f32   d, a, b        two sources, same bank
//  10  v_cmp_lt_f32 vcc, a, b ; v_addc_co_u32 d, vcc, 0, d, vcc     (compare + add-with-carry: two instructions per test)
//  11  v_lshrrev_b32 d, 31, a     12  v_add3_u32 d, d, a, b      13  v_max_f32 d, |a|, |b|     14  v_sub_f32 d, |a|, b
//  15  v_rsq_f32 d, a             16  v_bcnt_u32_b32 d, a, d     17  v_perm_b32 d, a, b, c     18  v_min_f32 d, d, |a|
//  19  v_pk_fma_f32 d[0:1], a[0:1], b[0:1], d[0:1] (counted as ONE instruction)   20  v_add_f32 d, a, d (VOP2)
//  21  v_cndmask_b32 d, a, b, vcc   22  v_and_or_b32 d, a, b, d   23  v_mad_u32_u24 d, a, b, d   24  v_cvt_pk_bf16_f32 d, a, b
// Prints instructions per cycle per SIMD (cycles from s_memtime) and the held clock.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_issue_probe valu_issue_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Rec {
  unsigned long long iters, cycles, ticks;
};

// 16 destination registers v[32..47] so that dependent reuse is 16 instructions apart; sources from v[48..63], v[64..79]
#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define XSTR(x) #x
#define STR(x) XSTR(x)

// register numbers: D(i) = 32 + i; A(i), B(i) chosen per pattern
#define I_P0(i) "v_fma_f32 v[32+" #i "], v[49+" #i "], v[66+" #i "], v[32+" #i "]\n"
#define I_P1(i) "v_fma_f32 v[32+" #i "], v[48+" #i "], v[64+" #i "], v[32+" #i "]\n"
#define I_P2(i) "v_fma_f32 v[32+" #i "], v[48+" #i "], v[65+" #i "], v[32+" #i "]\n"
#define I_P3(i) "v_fma_f32 v[32+" #i "], v[49+" #i "], v[49+" #i "], v[32+" #i "]\n"
#define I_P4(i) "v_fmac_f32 v[32+" #i "], v[49+" #i "], v[66+" #i "]\n"
#define I_P5(i) "v_alignbit_b32 v[32+" #i "], v[32+" #i "], v[49+" #i "], 31\n"
#define I_P6(i) "v_min3_f32 v[32+" #i "], v[32+" #i "], |v[49+" #i "]|, |v[66+" #i "]|\n"
#define I_P7(i) "v_fma_f32 v[32+" #i "], -v[49+" #i "], v[49+" #i "], -v[66+" #i "]\n"
#define I_P8(i) "v_mul_f32 v[32+" #i "], v[49+" #i "], v[66+" #i "]\n"
#define I_P9(i) "v_mul_f32 v[32+" #i "], v[48+" #i "], v[64+" #i "]\n"
#define I_P10(i) "v_cmp_lt_f32 vcc, v[49+" #i "], v[66+" #i "]\n v_addc_co_u32 v[32+" #i "], vcc, 0, v[32+" #i "], vcc\n"
#define I_P11(i) "v_lshrrev_b32 v[32+" #i "], 31, v[49+" #i "]\n"
#define I_P12(i) "v_add3_u32 v[32+" #i "], v[32+" #i "], v[49+" #i "], v[66+" #i "]\n"
#define I_P13(i) "v_max_f32 v[32+" #i "], |v[49+" #i "]|, |v[66+" #i "]|\n"
#define I_P14(i) "v_sub_f32 v[32+" #i "], |v[49+" #i "]|, v[66+" #i "]\n"
#define I_P15(i) "v_rsq_f32 v[32+" #i "], v[49+" #i "]\n"
#define I_P16(i) "v_bcnt_u32_b32 v[32+" #i "], v[49+" #i "], v[32+" #i "]\n"
#define I_P17(i) "v_perm_b32 v[32+" #i "], v[49+" #i "], v[66+" #i "], v[48]\n"
#define I_P18(i) "v_min_f32 v[32+" #i "], v[32+" #i "], |v[49+" #i "]|\n"
#define I_P19(i) "v_pk_fma_f32 v[32+2*(" #i "/2):33+2*(" #i "/2)], v[48+2*(" #i "/2):49+2*(" #i "/2)], v[64+2*(" #i "/2):65+2*(" #i "/2)], v[32+2*(" #i "/2):33+2*(" #i "/2)]\n"
#define I_P20(i) "v_add_f32 v[32+" #i "], v[49+" #i "], v[32+" #i "]\n"
#define I_P21(i) "v_cndmask_b32 v[32+" #i "], v[49+" #i "], v[66+" #i "], vcc\n"
#define I_P22(i) "v_and_or_b32 v[32+" #i "], v[49+" #i "], v[66+" #i "], v[32+" #i "]\n"
#define I_P23(i) "v_mad_u32_u24 v[32+" #i "], v[49+" #i "], v[66+" #i "], v[32+" #i "]\n"
#define I_P24(i) "v_cvt_pk_bf16_f32 v[32+" #i "], v[49+" #i "], v[66+" #i "]\n"

#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
             "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
             "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82"

template <int P>
__device__ __forceinline__ void body() {
  if (P == 0) asm volatile(REP16(I_P0) REP16(I_P0) REP16(I_P0) REP16(I_P0) ::: CLOB);
  if (P == 1) asm volatile(REP16(I_P1) REP16(I_P1) REP16(I_P1) REP16(I_P1) ::: CLOB);
  if (P == 2) asm volatile(REP16(I_P2) REP16(I_P2) REP16(I_P2) REP16(I_P2) ::: CLOB);
  if (P == 3) asm volatile(REP16(I_P3) REP16(I_P3) REP16(I_P3) REP16(I_P3) ::: CLOB);
  if (P == 4) asm volatile(REP16(I_P4) REP16(I_P4) REP16(I_P4) REP16(I_P4) ::: CLOB);
  if (P == 5) asm volatile(REP16(I_P5) REP16(I_P5) REP16(I_P5) REP16(I_P5) ::: CLOB);
  if (P == 6) asm volatile(REP16(I_P6) REP16(I_P6) REP16(I_P6) REP16(I_P6) ::: CLOB);
  if (P == 7) asm volatile(REP16(I_P7) REP16(I_P7) REP16(I_P7) REP16(I_P7) ::: CLOB);
  if (P == 8) asm volatile(REP16(I_P8) REP16(I_P8) REP16(I_P8) REP16(I_P8) ::: CLOB);
  if (P == 9) asm volatile(REP16(I_P9) REP16(I_P9) REP16(I_P9) REP16(I_P9) ::: CLOB);
  if (P == 10) asm volatile(REP16(I_P10) REP16(I_P10) REP16(I_P10) REP16(I_P10) ::: CLOB, "vcc");
  if (P == 11) asm volatile(REP16(I_P11) REP16(I_P11) REP16(I_P11) REP16(I_P11) ::: CLOB, "vcc");
  if (P == 12) asm volatile(REP16(I_P12) REP16(I_P12) REP16(I_P12) REP16(I_P12) ::: CLOB, "vcc");
  if (P == 13) asm volatile(REP16(I_P13) REP16(I_P13) REP16(I_P13) REP16(I_P13) ::: CLOB, "vcc");
  if (P == 14) asm volatile(REP16(I_P14) REP16(I_P14) REP16(I_P14) REP16(I_P14) ::: CLOB, "vcc");
  if (P == 15) asm volatile(REP16(I_P15) REP16(I_P15) REP16(I_P15) REP16(I_P15) ::: CLOB, "vcc");
  if (P == 16) asm volatile(REP16(I_P16) REP16(I_P16) REP16(I_P16) REP16(I_P16) ::: CLOB, "vcc");
  if (P == 17) asm volatile(REP16(I_P17) REP16(I_P17) REP16(I_P17) REP16(I_P17) ::: CLOB, "vcc");
  if (P == 18) asm volatile(REP16(I_P18) REP16(I_P18) REP16(I_P18) REP16(I_P18) ::: CLOB, "vcc");
  if (P == 19) asm volatile(REP16(I_P19) REP16(I_P19) REP16(I_P19) REP16(I_P19) ::: CLOB, "vcc");
  if (P == 20) asm volatile(REP16(I_P20) REP16(I_P20) REP16(I_P20) REP16(I_P20) ::: CLOB, "vcc");
  if (P == 21) asm volatile(REP16(I_P21) REP16(I_P21) REP16(I_P21) REP16(I_P21) ::: CLOB, "vcc");
  if (P == 22) asm volatile(REP16(I_P22) REP16(I_P22) REP16(I_P22) REP16(I_P22) ::: CLOB, "vcc");
  if (P == 23) asm volatile(REP16(I_P23) REP16(I_P23) REP16(I_P23) REP16(I_P23) ::: CLOB, "vcc");
  if (P == 24) asm volatile(REP16(I_P24) REP16(I_P24) REP16(I_P24) REP16(I_P24) ::: CLOB, "vcc");
}

template <int P>
__global__ __launch_bounds__(1024) void k(Rec* out, const float* __restrict__ in, unsigned long long ticks) {
  // small finite operands: products shrink towards zero slowly, nothing overflows
  const float a = in[threadIdx.x & 255], b = in[256 + (threadIdx.x & 255)];
  asm volatile(
      "v_mov_b32 v32, %0\n v_mov_b32 v33, %1\n v_mov_b32 v34, %0\n v_mov_b32 v35, %1\n v_mov_b32 v36, %0\n v_mov_b32 v37, %1\n v_mov_b32 v38, %0\n v_mov_b32 v39, %1\n"
      "v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n v_mov_b32 v42, %0\n v_mov_b32 v43, %1\n v_mov_b32 v44, %0\n v_mov_b32 v45, %1\n v_mov_b32 v46, %0\n v_mov_b32 v47, %1\n"
      "v_mov_b32 v48, %1\n v_mov_b32 v49, %0\n v_mov_b32 v50, %1\n v_mov_b32 v51, %0\n v_mov_b32 v52, %1\n v_mov_b32 v53, %0\n v_mov_b32 v54, %1\n v_mov_b32 v55, %0\n"
      "v_mov_b32 v56, %1\n v_mov_b32 v57, %0\n v_mov_b32 v58, %1\n v_mov_b32 v59, %0\n v_mov_b32 v60, %1\n v_mov_b32 v61, %0\n v_mov_b32 v62, %1\n v_mov_b32 v63, %0\n"
      "v_mov_b32 v64, %0\n v_mov_b32 v65, %1\n v_mov_b32 v66, %0\n v_mov_b32 v67, %1\n v_mov_b32 v68, %0\n v_mov_b32 v69, %1\n v_mov_b32 v70, %0\n v_mov_b32 v71, %1\n"
      "v_mov_b32 v72, %0\n v_mov_b32 v73, %1\n v_mov_b32 v74, %0\n v_mov_b32 v75, %1\n v_mov_b32 v76, %0\n v_mov_b32 v77, %1\n v_mov_b32 v78, %0\n v_mov_b32 v79, %1\n"
      "v_mov_b32 v80, %0\n v_mov_b32 v81, %1\n v_mov_b32 v82, %0\n" ::"v"(a), "v"(b)
      : CLOB);
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long deadline = r0 + ticks;
  unsigned long long iters = 0, r1 = r0;
  while (r1 < deadline) {
#pragma unroll 1
    for (int j = 0; j < 8; ++j) body<P>();
    iters += 8;
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sink;
  asm volatile("v_add_f32 %0, v32, v47" : "=v"(sink)::CLOB);
  if (sink == 12345.678f) out[0].iters = 1;  // keeps the registers alive
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = Rec{iters, t1 - t0, r1 - r0};
}

template <int P>
int run(int w, const float* in, Rec* out) {
  const int blocks = 256, threads = 256 * w, waves = blocks * threads / 64;
  const unsigned long long ticks = 100000ull * 300;  // 0.3 s per configuration (100 MHz ticks)
  k<P><<<blocks, threads>>>(out, in, ticks);
  CHECK(hipDeviceSynchronize());
  std::vector<Rec> h(waves);
  CHECK(hipMemcpy(h.data(), out, sizeof(Rec) * waves, hipMemcpyDeviceToHost));
  double inst = 0, cyc = 0;
  std::vector<double> clk;
  for (auto& r : h) {
    inst += (double)r.iters * 64.0 * (P == 10 ? 2.0 : 1.0);
    cyc += (double)r.cycles;
    clk.push_back((double)r.cycles / (double)r.ticks * 0.1);
  }
  std::sort(clk.begin(), clk.end());
  const double per_simd_inst = inst / 1024.0, mean_cyc = cyc / waves;
  printf("pattern %d  waves/SIMD %d: %.3f instructions per cycle per SIMD (%.2f cycles per instruction), per wave one instruction every %.2f cycles, held clock %.3f GHz\n",
         P, w, per_simd_inst / mean_cyc, mean_cyc / per_simd_inst, mean_cyc / (per_simd_inst / w), clk[clk.size() / 2]);
  fflush(stdout);
  return 0;
}

int main(int argc, char**) {
  float* in;
  Rec* out;
  CHECK(hipMalloc(&in, sizeof(float) * 512));
  CHECK(hipMalloc(&out, sizeof(Rec) * 256 * 16));
  std::vector<float> h(512);
  unsigned s = 777u;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = (0.5f + (float)((s >> 8) & 0xFFFF) / 131072.0f) * ((s >> 31) ? -1.0f : 1.0f);  // 0.5 .. 1.0, both signs
  }
  CHECK(hipMemcpy(in, h.data(), sizeof(float) * 512, hipMemcpyHostToDevice));
  const bool all = argc > 1;  // any argument: also the operand-bank patterns 0-9 at every occupancy
  for (int w : {1, 2, 4}) {
    if (all || w == 4) {
      if (run<0>(w, in, out)) return 1;
      if (run<1>(w, in, out)) return 1;
      if (run<2>(w, in, out)) return 1;
      if (run<3>(w, in, out)) return 1;
      if (run<4>(w, in, out)) return 1;
      if (run<5>(w, in, out)) return 1;
      if (run<6>(w, in, out)) return 1;
      if (run<7>(w, in, out)) return 1;
      if (run<8>(w, in, out)) return 1;
      if (run<9>(w, in, out)) return 1;
    }
    if (run<10>(w, in, out)) return 1;
    if (run<11>(w, in, out)) return 1;
    if (run<12>(w, in, out)) return 1;
    if (run<13>(w, in, out)) return 1;
    if (run<14>(w, in, out)) return 1;
    if (run<15>(w, in, out)) return 1;
    if (run<16>(w, in, out)) return 1;
    if (run<17>(w, in, out)) return 1;
    if (run<18>(w, in, out)) return 1;
    if (run<19>(w, in, out)) return 1;
    if (run<20>(w, in, out)) return 1;
    if (run<21>(w, in, out)) return 1;
    if (run<22>(w, in, out)) return 1;
    if (run<23>(w, in, out)) return 1;
    if (run<24>(w, in, out)) return 1;
  }
  return 0;
}
