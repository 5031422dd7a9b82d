// VALU issue-rate microbenchmark for gfx950: which f32 vector forms reach which rate.
//   hipcc --offload-arch=gfx950 -O3 -o build/valu_peak scripts/microbench/valu_peak.hip && build/valu_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
  const f32x2 ps = {s, s};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // v_add_f32, VGPR operands, 8 independent chains
      REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                        "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));)
    } else if (MODE == 1) {  // v_fma_f32
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                        "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));)
    } else if (MODE == 2) {  // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                        "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(ps));)
    } else if (MODE == 3) {  // v_pk_add_f32
      REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                        "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(ps));)
    } else if (MODE == 4) {  // v_sub_f32 with an SGPR operand
      REP8(asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n"
                        "v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));)
    } else if (MODE == 5) {  // one dependent chain: sub, mul, add (the kNN inner step)
      REP8(asm volatile("v_sub_f32 %1, %8, %2\n v_mul_f32 %1, %1, %1\n v_add_f32 %0, %0, %1\n"
                        "v_sub_f32 %3, %8, %4\n v_mul_f32 %3, %3, %3\n v_add_f32 %0, %0, %3\n"
                        "v_sub_f32 %5, %8, %6\n v_mul_f32 %5, %5, %5\n v_add_f32 %0, %0, %5\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));)
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[0] + p6[0] + p7[0];
}

template <int MODE>
void run(const char* name, int ops_per_inst, int inst_per_iter, int waves_per_simd) {
  float* out;
  const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100, 1.0001f);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double inst = (double)blocks * 4 * iters * inst_per_iter;  // wave-instructions
  printf("%-34s waves/SIMD=%d  %.2f ms  %.2f T lane-ops/s  %.2f cyc/inst/SIMD @2.4GHz\n", name, waves_per_simd, ms,
         inst * 64 * ops_per_inst / ms / 1e9, ms * 1e-3 * 2.4e9 / (inst / 1024));
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_add_f32 (vgpr)", 1, 64, w);
    run<1>("v_fma_f32", 1, 64, w);
    run<2>("v_pk_fma_f32", 2, 64, w);
    run<3>("v_pk_add_f32", 2, 64, w);
    run<4>("v_sub_f32 (sgpr operand)", 1, 64, w);
    run<5>("sub/mul/add dependent chain", 1, 72, w);
  }
  return 0;
}
