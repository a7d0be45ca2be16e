// How fast can one CU pull an L2-resident stream (the weight fragments of the Winograd kernels) -- by LDS-DMA
// (global_load_lds_dwordx4, 1 KB per wave instruction) or by plain global_load_dwordx4 into registers -- with every CU of the
// chip doing the same, alone and beside a stream of bf16 MFMAs?  8 waves per CU (two per SIMD), one workgroup per CU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/l2_stream_probe.hip -o /tmp/l2_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: LDS-DMA into a private 8 KB ring per wave; MODE 1: global_load_dwordx4 into registers; MFMA = bf16 MFMAs per piece
template <int MODE, int MFMA, int DEPTH>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
probe(const unsigned char* __restrict__ src, unsigned window, int iters, float* out) {
  extern __shared__ __align__(16) float lds[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.0f + lane * 1e-3f); b[e] = (__bf16)1.0f; }
  u32x4 sink = {0u, 0u, 0u, 0u};
  u32x4 buf[DEPTH];
  const unsigned vo = (unsigned)lane * 16u;
  unsigned off = (unsigned)(wv * 1024 + blockIdx.x * 8192) % window;
  // prime DEPTH pieces
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    if (MODE == 0) {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(vo), "s"(lds_base + (unsigned)(wv * 8192 + (d & 7) * 1024)), "s"(src + off) : "memory", "m0");
    } else {
      buf[d] = *reinterpret_cast<const u32x4*>(src + off + vo);
    }
    off += 8192; if (off >= window) off -= window;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (MODE == 0) {
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DEPTH - 1) : "memory");
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(vo), "s"(lds_base + (unsigned)(wv * 8192 + (d & 7) * 1024)), "s"(src + off) : "memory", "m0");
      } else {
        sink ^= buf[d];
        buf[d] = *reinterpret_cast<const u32x4*>(src + off + vo);
      }
      off += 8192; if (off >= window) off -= window;
#pragma unroll
      for (int k = 0; k < MFMA; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k & 3], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (MODE == 1) for (int d = 0; d < DEPTH; ++d) sink ^= buf[d];
  s += (float)(sink[0] ^ sink[1] ^ sink[2] ^ sink[3]) * 1e-30f + lds[threadIdx.x];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int MFMA, int DEPTH>
void run(unsigned window, int iters, const unsigned char* src, float* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, MFMA, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t lds = 100 * 1024;                     // one workgroup per CU
  probe<MODE, MFMA, DEPTH><<<256, 512, lds>>>(src, window, 10, out);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    probe<MODE, MFMA, DEPTH><<<256, 512, lds>>>(src, window, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double pieces = (double)iters * DEPTH * 8;                 // per CU
  const double ns_piece = best * 1e6 / pieces;
  printf("%s window %5u KB, %d in flight per wave, %d MFMA per piece: %.1f ns per 1-KB piece per CU = %.1f GB/s per CU (%.1f B/clk at 2.4 GHz), chip %.2f TB/s",
         MODE == 0 ? "LDS-DMA  " : "registers", window >> 10, DEPTH, MFMA, ns_piece, 1024.0 / ns_piece, 1024.0 / ns_piece / 2.4, 256 * 1.024 / ns_piece);
  if (MFMA) printf(";  %.1f ns per MFMA per SIMD", best * 1e6 / ((double)iters * DEPTH * MFMA * 2));
  printf("\n");
}

int main() {
  unsigned char* src; float* out;
  const unsigned big = 64u << 20;
  (void)hipMalloc(&src, big); (void)hipMemset(src, 1, big);
  (void)hipMalloc(&out, sizeof(float) * 256 * 512);
  const unsigned w_small = 1536u << 10;               // the 128 -> 128 layer's split weights: L2-resident
  run<0, 0, 4>(w_small, 2000, src, out);
  run<0, 0, 8>(w_small, 1000, src, out);
  run<1, 0, 4>(w_small, 2000, src, out);
  run<1, 0, 8>(w_small, 1000, src, out);
  run<0, 0, 8>(big, 1000, src, out);
  run<1, 0, 8>(big, 1000, src, out);
  run<0, 3, 8>(w_small, 1000, src, out);
  run<1, 3, 8>(w_small, 1000, src, out);
  run<0, 6, 8>(w_small, 500, src, out);
  run<1, 6, 8>(w_small, 500, src, out);
  run<0, 12, 8>(w_small, 300, src, out);
  run<1, 12, 8>(w_small, 300, src, out);
  return 0;
}
