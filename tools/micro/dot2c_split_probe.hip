// Is v_dot2c_f32_bf16 (D += A.lo * B.lo + A.hi * B.hi, packed bf16 pairs, gfx950) an EXACT "subtract the bf16 in this half-word"?
// The truncating 3-way split of m4d_common.h per pair of float32 values: q0 = perm(hi16(x1), hi16(x0)); r = x - hi16(x) [and + sub
// per element]; q1 = perm(hi16(r1), hi16(r0)); s = r - hi16(r); q2 = perm(hi16(s1), hi16(s0)): 11 VALU instructions.  With the dot
// product the two (and, sub) pairs of a step become two v_dot2c with the constant operands {-1, 0} / {0, -1}: 7 instructions.
// This probe compares the two over adversarial values (all exponents, denormal residues, negative zero, Inf / NaN excluded)
// bit for bit.  Build: hipcc -O3 --offload-arch=gfx950 tools/micro/dot2c_split_probe.hip -o /tmp/dot2c_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned perm_hi(float x1, float x0) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}
__device__ __forceinline__ float hi_f(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u); }
__global__ void probe(const float* x, unsigned* ref, unsigned* got, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float x0 = x[2 * i], x1 = x[2 * i + 1];
  {  // reference: and + sub
    const unsigned q0 = perm_hi(x1, x0);
    const float r0 = x0 - hi_f(x0), r1 = x1 - hi_f(x1);
    const unsigned q1 = perm_hi(r1, r0);
    const float s0 = r0 - hi_f(r0), s1 = r1 - hi_f(r1);
    ref[3 * i] = q0; ref[3 * i + 1] = q1; ref[3 * i + 2] = perm_hi(s1, s0);
  }
  {  // v_dot2c
    const bf16x2 m_lo = {(__bf16)-1.0f, (__bf16)0.0f}, m_hi = {(__bf16)0.0f, (__bf16)-1.0f};
    const unsigned q0 = perm_hi(x1, x0);
    const float r0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, q0), m_lo, x0, false);
    const float r1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, q0), m_hi, x1, false);
    const unsigned q1 = perm_hi(r1, r0);
    const float s0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, q1), m_lo, r0, false);
    const float s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, q1), m_hi, r1, false);
    got[3 * i] = q0; got[3 * i + 1] = q1; got[3 * i + 2] = perm_hi(s1, s0);
  }
}
int main() {
  std::vector<float> h;
  uint32_t st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st; };
  // every exponent with random mantissas (finite), both signs; special mantissa patterns
  for (int e = 0; e < 255; ++e)
    for (int k = 0; k < 4096; ++k) {
      uint32_t m = rnd() & 0x7fffffu;
      if (k < 8) m = (k & 1) ? 0x7fffffu >> (k * 2) : (1u << (k * 2));
      if (k == 8) m = 0;
      if (k == 9) m = 0x00ffffu;
      if (k == 10) m = 0x0000ffu;
      if (k == 11) m = 0x7f0000u;
      const uint32_t bits = ((rnd() & 1u) << 31) | ((uint32_t)e << 23) | m;
      float f; memcpy(&f, &bits, 4); h.push_back(f);
    }
  const int n = (int)h.size();
  float* dx; unsigned *dr, *dg;
  (void)hipMalloc(&dx, 4 * n); (void)hipMalloc(&dr, 4 * 3 * (n / 2)); (void)hipMalloc(&dg, 4 * 3 * (n / 2));
  (void)hipMemcpy(dx, h.data(), 4 * n, hipMemcpyHostToDevice);
  probe<<<(n / 2 + 255) / 256, 256>>>(dx, dr, dg, n);
  std::vector<unsigned> r(3 * (n / 2)), g(3 * (n / 2));
  (void)hipMemcpy(r.data(), dr, 4 * r.size(), hipMemcpyDeviceToHost); (void)hipMemcpy(g.data(), dg, 4 * g.size(), hipMemcpyDeviceToHost);
  long long bad = 0, bad_normal = 0; int shown = 0;
  for (int i = 0; i < n / 2; ++i)
    for (int p = 0; p < 3; ++p)
      if (r[3 * i + p] != g[3 * i + p]) {
        ++bad;
        uint32_t b0, b1; memcpy(&b0, &h[2 * i], 4); memcpy(&b1, &h[2 * i + 1], 4);
        const int e0 = (b0 >> 23) & 255, e1 = (b1 >> 23) & 255;
        if (e0 >= 40 && e1 >= 40) ++bad_normal;
        if (shown < 12) { printf("pair %d part %d: x = %08x %08x (exponents %d %d)  and+sub %08x  dot2c %08x\n", i, p, b0, b1, e0, e1, r[3 * i + p], g[3 * i + p]); ++shown; }
      }
  printf("%d pairs, %lld of %d packed words differ; %lld of them with both exponents >= 40 (|x| >= 2^-87)\n", n / 2, bad, 3 * (n / 2), bad_normal);
  return 0;
}
