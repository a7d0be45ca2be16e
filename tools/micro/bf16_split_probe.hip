// Can the bf16 matrix cores reproduce a float32 GEMM?  Every float32 value is EXACTLY a1 + a2 + a3 with three bf16
// numbers (8 + 8 + 8 significand bits, round-to-nearest splits), so a*b = sum of 9 bf16 x bf16 products, each exact in
// float32; the three smallest (a2*b3, a3*b2, a3*b3 <= 2^-26 |a*b|) are below float32's own product rounding.  This probe
// measures, for one 32x32 output tile and K = 128 / 1152 on Gaussian operands, the error against a float64 host result of
//   (f) the float32 MFMA chain (v_mfma_f32_32x32x2_f32, what the Winograd kernels use today),
//   (6) six bf16 MFMAs per 16 k (v_mfma_f32_32x32x16_bf16; small terms first),
//   (3) three bf16 MFMAs (a1b1 + a1b2 + a2b1), for scale,
// in units of 2^-24 * sum_k |a_k b_k| (the float32 accumulation scale), plus the mean SIGNED error (a truncating
// accumulator inside the bf16 MFMA would show as a bias).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/bf16_split_probe.hip -o /tmp/bf16_split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __bf16 to_bf16_rn(float v) { return (__bf16)v; }

// A: [32][K] row-major, B: [K][32] row-major; out[mode][32][32]
template <int MODE>
__global__ void __launch_bounds__(64) gemm_tile(const float* __restrict__ A, const float* __restrict__ B, int K, float* __restrict__ out) {
  const int l = threadIdx.x, i = l & 31, hi = l >> 5;
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (MODE == 0) {
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + hi], B[(k + hi) * 32 + i], acc, 0, 0, 0);
  } else {
    for (int k0 = 0; k0 < K; k0 += 16) {
      bf16x8 a[3], b[3];
      for (int e = 0; e < 8; ++e) {
        float va = A[i * K + k0 + hi * 8 + e], vb = B[(k0 + hi * 8 + e) * 32 + i];
        __bf16 a1 = to_bf16_rn(va); float ra = va - (float)a1; __bf16 a2 = to_bf16_rn(ra); __bf16 a3 = to_bf16_rn(ra - (float)a2);
        __bf16 b1 = to_bf16_rn(vb); float rb = vb - (float)b1; __bf16 b2 = to_bf16_rn(rb); __bf16 b3 = to_bf16_rn(rb - (float)b2);
        a[0][e] = a1; a[1][e] = a2; a[2][e] = a3; b[0][e] = b1; b[1][e] = b2; b[2][e] = b3;
      }
      if (MODE == 6 || MODE == 9) {
        if (MODE == 9) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
      }
      if (MODE == 7) {        // 6 terms, the small ones in their own accumulator, added once at the end
        // (handled below through a second pass: MODE 7 = large terms only here)
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + i] = acc[r];
}

static double gauss() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }

template <int MODE> void run(const char* name, int K, bool positive) {
  std::vector<float> A(32 * K), B(K * 32);
  for (auto& v : A) v = (float)(positive ? fabs(gauss()) : gauss());
  for (auto& v : B) v = (float)(positive ? fabs(gauss()) : gauss());
  float *dA, *dB, *dO; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, 1024 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  gemm_tile<MODE><<<1, 64>>>(dA, dB, K, dO);
  std::vector<float> O(1024); hipMemcpy(O.data(), dO, 4096, hipMemcpyDeviceToHost);
  double sum_abs = 0, sum_sq = 0, sum_signed = 0, worst = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double ref = 0, scale = 0;
    for (int k = 0; k < K; ++k) { double p = (double)A[i * K + k] * (double)B[k * 32 + j]; ref += p; scale += fabs(p); }
    double e = ((double)O[i * 32 + j] - ref) / (scale * ldexp(1.0, -24));
    sum_abs += fabs(e); sum_sq += e * e; sum_signed += e; if (fabs(e) > worst) worst = fabs(e);
  }
  printf("%-28s K=%4d %s  mean|e| %.3f  rms %.3f  max %.3f  mean signed %+.3f   [units of 2^-24 sum|a b|]\n", name, K,
         positive ? "positive" : "gaussian", sum_abs / 1024, sqrt(sum_sq / 1024), worst, sum_signed / 1024);
  hipFree(dA); hipFree(dB); hipFree(dO);
}

int main() {
  for (int pos = 0; pos < 2; ++pos)
    for (int K : {16, 128, 1152}) {
      srand(1234 + K); run<0>("float32 MFMA chain", K, pos);
      srand(1234 + K); run<3>("bf16 x3 (a1b1+a1b2+a2b1)", K, pos);
      srand(1234 + K); run<6>("bf16 x6", K, pos);
      srand(1234 + K); run<9>("bf16 x9", K, pos);
    }
  return 0;
}
