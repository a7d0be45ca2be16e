// What does the sustained v_mfma_f32_32x32x16_bf16 rate depend on?  tools/micro/mfma_bf16_dep.hip feeds every MFMA the same two
// operand registers holding trivial values (1.0, 1 + lane / 1000) and reaches the nominal 32-cycle rate (13.5 ns per MFMA and
// SIMD = 2.48 PFLOP/s); real kernels feed a different A / B fragment of real data to (nearly) every MFMA.  This probe varies
// both: NA x NB distinct operand register quads (which registers: no effect) and the operand VALUES (all ones against
// pseudo-random: the whole effect -- with non-trivial data the chip sustains ~19.5 ns per MFMA, 0.69 of the nominal rate, every
// CU busy).  8 rotating accumulators, one or two waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_operand_variety.hip -o /tmp/mfma_operand_variety
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NA, int NB, int MODE>
__global__ void __launch_bounds__(256, 2) loop(float* out, const float* in, int iters) {
  constexpr int NACC = 8;
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[NA], b[NB];
  for (int q = 0; q < NA; ++q) for (int e = 0; e < 8; ++e) a[q][e] = (__bf16)in[(threadIdx.x * 7 + q * 13 + e) & 1023];
  for (int q = 0; q < NB; ++q) for (int e = 0; e < 8; ++e) b[q][e] = (__bf16)in[(threadIdx.x * 3 + q * 29 + e) & 1023];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        // MODE 0: operands change with every MFMA; MODE 1: A is held for 4 consecutive MFMAs, B changes; MODE 2: both held for 8
        const int ia = MODE == 0 ? (i + k) % NA : MODE == 1 ? ((i >> 2) + k) % NA : k % NA;
        const int ib = MODE == 0 ? (i * 3 + k) % NB : MODE == 1 ? (i + k) % NB : k % NB;
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia], b[ib], acc[i], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
void run(const char* name, K kernel, int blocks, int iters, const float* in) {
  float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  kernel<<<blocks, 256>>>(out, in, 10);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    kernel<<<blocks, 256>>>(out, in, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double n_mfma = (double)iters * 48 * (blocks / 256.0);
  printf("%-64s %3d workgroups: %.1f ns per MFMA per SIMD\n", name, blocks, best * 1e6 / n_mfma);
  (void)hipFree(out);
}

int main() {
  float* in; (void)hipMalloc(&in, 4096);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  (void)hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  float* ones; (void)hipMalloc(&ones, 4096);
  float h1[1024];
  for (int i = 0; i < 1024; ++i) h1[i] = 1.0f;
  (void)hipMemcpy(ones, h1, 4096, hipMemcpyHostToDevice);
  printf("-- operand VALUES: all 1.0\n");
  run("1 A x 1 B register quad (every MFMA the same operands)", loop<1, 1, 0>, 256, 4000, ones);
  run("6 A x 12 B, operands change with every MFMA", loop<6, 12, 0>, 256, 4000, ones);
  run("6 A x 12 B, operands change with every MFMA, 2 waves / SIMD", loop<6, 12, 0>, 512, 4000, ones);
  printf("-- operand VALUES: pseudo-random in [-0.5, 0.5)\n");
  run("1 A x 1 B register quad (every MFMA the same operands)", loop<1, 1, 0>, 256, 4000, in);
  run("6 A x 12 B, operands change with every MFMA", loop<6, 12, 0>, 256, 4000, in);
  run("6 A x 12 B, operands change with every MFMA, 2 waves / SIMD", loop<6, 12, 0>, 512, 4000, in);
  run("6 A x 12 B, A held for 4 MFMAs", loop<6, 12, 1>, 256, 4000, in);
  run("6 A x 12 B, A and B held for 8 MFMAs", loop<6, 12, 2>, 256, 4000, in);
  run("2 A x 2 B, operands change with every MFMA", loop<2, 2, 0>, 256, 4000, in);
  return 0;
}
