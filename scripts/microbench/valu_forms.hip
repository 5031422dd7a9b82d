// Per-form VALU issue cost on gfx950 (cycles per wave-instruction per SIMD at 8 waves/SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o build/valu_forms scripts/microbench/valu_forms.hip && build/valu_forms
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
#define FORM8(TXT, ...) REP8(asm volatile(TXT TXT TXT TXT TXT TXT TXT TXT : __VA_ARGS__);)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = p0 + 1.f;
  const f32x2 ps = {s, s};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { FORM8("v_sub_f32 %0, %4, %1\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s)) }          // sgpr, vgpr
    if (MODE == 1) { FORM8("v_sub_f32 %0, %1, %2\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s)) }          // 2 distinct vgprs
    if (MODE == 2) { FORM8("v_mul_f32 %0, %1, %1\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s)) }          // same vgpr twice
    if (MODE == 3) { FORM8("v_add_f32 %0, %0, %1\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s)) }          // dependent accumulate
    if (MODE == 4) { FORM8("v_mov_b32 %0, %4\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s)) }              // sgpr -> vgpr
    if (MODE == 5) { FORM8("v_pk_add_f32 %0, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n", "+v"(p0), "+v"(p1), "+v"(p2) : "s"(ps)) }  // pk, sgpr pair
    if (MODE == 6) { FORM8("v_pk_add_f32 %0, %1, %2\n", "+v"(p0), "+v"(p1), "+v"(p2) : "s"(ps)) }                // pk, vgpr pairs
    if (MODE == 7) { FORM8("v_pk_mul_f32 %0, %1, %1\n", "+v"(p0), "+v"(p1), "+v"(p2) : "s"(ps)) }                // pk, same pair
    if (MODE == 8) { FORM8("v_sub_f32 %1, %4, %2\n v_mul_f32 %1, %1, %1\n v_add_f32 %0, %0, %1\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s)) }  // kNN step, sgpr
    if (MODE == 9) { FORM8("v_sub_f32 %1, %3, %2\n v_mul_f32 %1, %1, %1\n v_add_f32 %0, %0, %1\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s)) }  // kNN step, vgpr
    if (MODE == 10) { FORM8("v_pk_add_f32 %1, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %1, %1, %1\n v_pk_add_f32 %0, %0, %1\n", "+v"(p0), "+v"(p1), "+v"(p2) : "s"(ps)) }  // packed step, sgpr pair
    if (MODE == 11) { FORM8("v_pk_add_f32 %1, %2, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %1, %1, %1\n v_pk_add_f32 %0, %0, %1\n", "+v"(p0), "+v"(p1), "+v"(p2) : "s"(ps)) }  // packed step, vgpr
    if (MODE == 12) { FORM8("v_fma_f32 %0, %1, %1, %0\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s)) }     // fma 2 distinct
    if (MODE == 13) { FORM8("v_fma_f32 %0, %1, %2, %0\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s)) }     // fma 3 distinct
    if (MODE == 14) { FORM8("v_sub_f32_dpp %0, %1, %2 row_newbcast:3\n", "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s)) }  // dpp broadcast operand
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0[0] + p1[1] + p2[0];
}

template <int MODE>
void run(const char* name, int inst_per_form) {
  float* out;
  const int blocks = 256 * 8;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 10000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100, 1.0001f);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double inst = (double)blocks * 4 * iters * 64 * inst_per_form;
  printf("%-46s %.2f cyc/inst/SIMD @2.4GHz\n", name, ms * 1e-3 * 2.4e9 / (inst / 1024));
  hipFree(out);
}

int main() {
  run<0>("v_sub_f32 v, s, v", 1);
  run<1>("v_sub_f32 v, v1, v2", 1);
  run<2>("v_mul_f32 v, v1, v1", 1);
  run<3>("v_add_f32 v0, v0, v1 (dependent)", 1);
  run<4>("v_mov_b32 v, s", 1);
  run<5>("v_pk_add_f32 v2, v2, s2 (neg)", 1);
  run<6>("v_pk_add_f32 v2, v2, v2", 1);
  run<7>("v_pk_mul_f32 v2, v2, v2 (same)", 1);
  run<8>("step sub(s)/mul/add            [per inst]", 3);
  run<9>("step sub(v)/mul/add            [per inst]", 3);
  run<10>("packed step pk_add(s2)/pk_mul/pk_add [per inst]", 3);
  run<11>("packed step, vgpr op_sel          [per inst]", 3);
  run<12>("v_fma_f32 v0, v1, v1, v0", 1);
  run<13>("v_fma_f32 v0, v1, v2, v0", 1);
  run<14>("v_sub_f32_dpp row_newbcast", 1);
  return 0;
}
