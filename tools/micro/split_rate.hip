// What does the exact 3-way bf16 split cost next to bf16 MFMAs?  One wave per SIMD (or two), a register-only loop:
// per MFMA, SPLITS pair-splits (11 VALU instructions each: 3 v_cvt_pk_bf16_f32, 2 shifts, 2 ands, 4 subtractions) of
// independent values.  Prints ns per MFMA per SIMD; 13.5 ns = the bare MFMA rate.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/micro/split_rate.hip -o /tmp/split_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk(float a, float b) { bf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b; return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

template <int WAVES, int MODE, int PER4>      // PER4 = pair-splits per 4 MFMAs
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES / 4, WAVES / 4))) loop(float* out, int iters, float a0) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + threadIdx.x * 1e-3f); b[e] = (__bf16)1.0f; }
  float x[16];
  for (int e = 0; e < 16; ++e) x[e] = a0 * (1.37f + e) + threadIdx.x * 1e-4f;
  unsigned sink = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < PER4; ++s) {
        float x0 = x[(2 * s) & 15], x1 = x[(2 * s + 1) & 15];
        if (MODE == 0) {                 // the real split
          const unsigned q0 = pk(x0, x1);
          const float r0 = x0 - lo(q0), r1 = x1 - hi(q0);
          const unsigned q1 = pk(r0, r1);
          const float s0 = r0 - lo(q1), s1 = r1 - hi(q1);
          const unsigned q2 = pk(s0, s1);
          sink ^= q0 + q1 + q2;
          x[(2 * s) & 15] = x0 + 0.5f; x[(2 * s + 1) & 15] = x1 + 0.25f;
        } else {                          // the same number of plain adds (13 VALU)
          float u = x0, v = x1;
#pragma unroll
          for (int k = 0; k < 6; ++k) { u = u + 0.5f; v = v + 0.25f; }
          x[(2 * s) & 15] = u; x[(2 * s + 1) & 15] = v + u;
        }
      }
      asm volatile("" : "+v"(sink));
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, (PER4 * 13 + 3) / 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, (PER4 * 13 + 3) / 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, (PER4 * 13 + 3) / 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, (PER4 * 13 + 3) / 4, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int e = 0; e < 16; ++e) s += x[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)sink;
}

template <int WAVES, int MODE, int PER4>
void run(const char* what, int iters) {
  const int blocks = 256;
  float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 64 * WAVES);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  loop<WAVES, MODE, PER4><<<blocks, 64 * WAVES>>>(out, 10, 1.f);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    loop<WAVES, MODE, PER4><<<blocks, 64 * WAVES>>>(out, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double n_mfma_simd = (double)iters * 32 * (WAVES / 4);
  printf("%d wave(s)/SIMD, %s, %4.1f VALU per MFMA: %.1f ns per MFMA per SIMD\n", WAVES / 4, what, PER4 * 13 / 4.0, best * 1e6 / n_mfma_simd);
  (void)hipFree(out);
}

int main() {
  run<4, 0, 0>("no VALU", 4000);
  run<4, 1, 1>("plain adds", 4000); run<4, 1, 2>("plain adds", 4000);
  run<4, 0, 1>("bf16 split", 4000); run<4, 0, 2>("bf16 split", 4000); run<4, 0, 3>("bf16 split", 4000);
  run<8, 0, 0>("no VALU", 4000);
  run<8, 1, 2>("plain adds", 4000);
  run<8, 0, 1>("bf16 split", 4000); run<8, 0, 2>("bf16 split", 4000);
  return 0;
}
