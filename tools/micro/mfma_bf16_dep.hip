// bf16 MFMA rate against the accumulator dependency distance: a register-only loop of v_mfma_f32_32x32x16_bf16 over NACC
// independent accumulators per wave (each accumulator is reused every NACC-th MFMA), one wave per SIMD, every CU busy.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_bf16_dep.hip -o /tmp/mfma_bf16_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int FILL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) loop(float* out, int iters, float a0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + threadIdx.x * 1e-3f); b[e] = (__bf16)1.0f; }
  float f[8];
  for (int e = 0; e < 8; ++e) f[e] = a0 + e;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < FILL; ++e) f[e] = f[e] * 1.0001f + 0.5f;     // FILL independent VALU instructions per MFMA (2 each: mul, add)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2 * FILL, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int e = 0; e < 8; ++e) s += f[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int FILL>
void run(int blocks, int iters) {
  float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  loop<NACC, FILL><<<blocks, 256>>>(out, 10, 1.f);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    loop<NACC, FILL><<<blocks, 256>>>(out, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double n_mfma = (double)iters * 8 * NACC;
  printf("accumulators %2d, %d VALU per MFMA: %.1f ns per MFMA per SIMD, %.0f TFLOP/s\n", NACC, 2 * FILL,
         best * 1e6 / n_mfma, (double)blocks * 4 * n_mfma * 32768.0 / best * 1e-9);
  (void)hipFree(out);
}

int main() {
  run<1, 0>(256, 20000); run<2, 0>(256, 20000); run<4, 0>(256, 10000); run<8, 0>(256, 5000); run<16, 0>(256, 2500);
  run<4, 1>(256, 10000); run<4, 2>(256, 10000); run<4, 3>(256, 10000); run<4, 4>(256, 10000);
  run<16, 2>(256, 2500); run<16, 3>(256, 2500); run<16, 4>(256, 2500);
  return 0;
}
