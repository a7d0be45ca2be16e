// Cost of individual VALU instructions in the shadow of bf16 MFMAs (one wave per SIMD, register-only): per MFMA, N copies of
// one instruction kind on independent registers.  ns per MFMA per SIMD; 13.5-15.6 = bare.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_cost.hip -o /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND, int N>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) loop(float* out, int iters, float a0, unsigned mask) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + threadIdx.x * 1e-3f); b[e] = (__bf16)1.0f; }
  float x[8]; unsigned u[8];
  for (int e = 0; e < 8; ++e) { x[e] = a0 * (1.37f + e) + threadIdx.x * 1e-4f; u[e] = threadIdx.x * 77u + e; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const int e = k & 7;
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[e]) : "v"(x[(e + 1) & 7]));
        if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[e]) : "v"(x[e]), "v"(x[(e + 1) & 7]));
        if (KIND == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[e]) : "v"(u[(e + 1) & 7]));
        if (KIND == 3) asm volatile("v_and_b32 %0, %2, %1" : "=v"(u[e]) : "v"(u[(e + 1) & 7]), "s"(mask));
        if (KIND == 4) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[e]) : "v"(u[(e + 1) & 7]));
        if (KIND == 5) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x[e]) : "v"(x[(e + 1) & 7]), "v"(x[(e + 2) & 7]));
        if (KIND == 6) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(double*)&x[2 * (k & 3)]) : "v"(*(double*)&x[2 * ((k + 1) & 3)]), "v"(*(double*)&x[2 * ((k + 2) & 3)]));
        if (KIND == 7) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[e]) : "v"(u[(e + 1) & 7]), "v"(u[(e + 2) & 7]), "s"(mask));
        if (KIND == 8) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[e]) : "v"(x[(e + 1) & 7]), "v"(x[(e + 2) & 7]), "v"(x[(e + 3) & 7]));
        if (KIND == 9) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x[e]) : "s"(mask), "v"(u[(e + 1) & 7]));   // D += A . B (packed bf16 pairs)
        if (KIND == 10) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x[e]) : "v"(u[(e + 2) & 7]), "v"(u[(e + 1) & 7]));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int e = 0; e < 8; ++e) s += x[e] + (float)u[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int N> void run(const char* what) {
  const int blocks = 256, iters = 2000;
  float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  loop<KIND, N><<<blocks, 256>>>(out, 10, 1.f, 0xffff0000u);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    loop<KIND, N><<<blocks, 256>>>(out, iters, 1.f, 0xffff0000u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-34s x%d per MFMA: %.1f ns per MFMA\n", what, N, best * 1e6 / ((double)iters * 32));
  (void)hipFree(out);
}
int main() {
  run<0, 0>("nothing");
  run<0, 6>("v_add_f32"); run<0, 10>("v_add_f32");
  run<1, 6>("v_cvt_pk_bf16_f32"); run<1, 10>("v_cvt_pk_bf16_f32");
  run<2, 6>("v_and_b32 literal"); run<2, 10>("v_and_b32 literal");
  run<3, 6>("v_and_b32 sgpr"); run<3, 10>("v_and_b32 sgpr");
  run<4, 6>("v_lshlrev_b32"); run<4, 10>("v_lshlrev_b32");
  run<5, 6>("v_sub_f32"); run<5, 10>("v_sub_f32");
  run<6, 3>("v_pk_add_f32"); run<6, 5>("v_pk_add_f32"); run<6, 6>("v_pk_add_f32");
  run<7, 6>("v_perm_b32"); run<7, 10>("v_perm_b32");
  run<8, 6>("v_fma_f32"); run<8, 10>("v_fma_f32");
  run<9, 6>("v_dot2c_f32_bf16 sgpr, vgpr"); run<9, 10>("v_dot2c_f32_bf16 sgpr, vgpr");
  run<10, 6>("v_dot2c_f32_bf16 vgpr, vgpr"); run<10, 10>("v_dot2c_f32_bf16 vgpr, vgpr");
  return 0;
}
