// Attainable fp32-MFMA rate on this chip: a register-only loop of v_mfma_f32_32x32x2_f32 with 4 independent
// accumulators per wave (the dependency structure of conv3x3_mfma_kernel<4,...>), no memory traffic.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(const char* name, int blocks, int iters) {
  float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, 256>>>(out, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
    printf("%s blocks=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, iters, ms, flops / ms * 1e-9);
  }
  hipFree(out);
}

int main() {
  run<4>("4 acc/wave, 1 wave/SIMD", 256, 20000);        // one workgroup per CU
  run<4>("4 acc/wave, 2 waves/SIMD", 512, 20000);
  run<4>("4 acc/wave, 2 waves/SIMD, long", 512, 200000); // ~0.7 s: sustained clocks
  run<1>("1 acc/wave (dependent chain), 2 waves/SIMD", 512, 20000);
  run<1>("1 acc/wave (dependent chain), 1 wave/SIMD", 256, 20000);
  run<2>("2 acc/wave, 1 wave/SIMD", 256, 20000);
  return 0;
}
