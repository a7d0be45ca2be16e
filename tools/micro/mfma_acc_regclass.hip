// Does v_mfma_f32_32x32x16_bf16 run at the same rate with its accumulator (srcC / vDst) in ArchVGPRs as in AccVGPRs?
// hipcc selects the VGPR form of an MFMA whenever the kernel's register budget fits 256 registers (two waves per SIMD and
// up: amdgpu_waves_per_eu(2, ..)), the AGPR form when a wave may own more (one wave per SIMD).  Same loop, NACC rotating
// accumulators, one wave per SIMD resident in both cases (256 workgroups of 4 waves on 256 CUs).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_acc_regclass.hip -o /tmp/mfma_acc_regclass
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__device__ __forceinline__ void body(float* out, int iters, float a0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + threadIdx.x * 1e-3f); b[e] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) loop_agpr(float* out, int iters, float a0) { body<NACC>(out, iters, a0); }
template <int NACC> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) loop_vgpr(float* out, int iters, float a0) { body<NACC>(out, iters, a0); }
template <int NACC> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) loop_12(float* out, int iters, float a0) { body<NACC>(out, iters, a0); }

template <class K>
void run(const char* name, K kernel, int nacc, int blocks, int iters) {
  float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  kernel<<<blocks, 256>>>(out, 10, 1.f);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    kernel<<<blocks, 256>>>(out, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double n_mfma = (double)iters * 8 * nacc * (blocks / 256.0);
  printf("%-28s %2d accumulators, %d workgroups: %.1f ns per MFMA per SIMD\n", name, nacc, blocks, best * 1e6 / n_mfma);
  (void)hipFree(out);
}

int main() {
  run("AGPR form (1 wave/SIMD)", loop_agpr<2>, 2, 256, 20000);
  run("AGPR form (1 wave/SIMD)", loop_agpr<8>, 8, 256, 5000);
  run("VGPR form (budget 256)", loop_vgpr<2>, 2, 256, 20000);
  run("VGPR form (budget 256)", loop_vgpr<8>, 8, 256, 5000);
  run("VGPR form, 2 waves/SIMD", loop_vgpr<8>, 8, 512, 5000);
  run("waves_per_eu(1,2)", loop_12<8>, 8, 256, 5000);
  run("waves_per_eu(1,2), 2 waves/SIMD", loop_12<8>, 8, 512, 5000);
  return 0;
}
