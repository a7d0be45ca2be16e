// How fast does one CU issue global_store_dwordx4, as a function of how the 64 lanes' 16-byte pieces are laid out?
// 256 workgroups x 512 threads (8 waves, one workgroup per CU via 150 KB of LDS), every thread issues 8 float4 stores per
// "unit" (64 KB per workgroup per unit, the output of one wino6 unit), NU units; s_memtime around the store phase.
//   pattern 0: 8 lanes = 128 contiguous bytes, lane groups 1024 B apart (one 32-cout N-tile of a 128-channel NHWC pixel row: today)
//   pattern 1: 16 lanes = 256 contiguous bytes, groups 1024 B apart (both N-tiles of a 64-cout group)
//   pattern 2: 64 lanes = 1024 contiguous bytes
// hipcc -O3 --offload-arch=gfx950 tools/micro/store_issue_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int PATTERN>
__global__ void __launch_bounds__(512) probe(float* out, long long per_wg_floats, int nu, unsigned long long* cyc) {
  extern __shared__ float lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float* base = out + (long long)blockIdx.x * per_wg_floats;
  lds[t] = (float)t;
  __syncthreads();
  const float4 v = make_float4(lds[t], lds[(t + 1) & 511], lds[(t + 2) & 511], lds[(t + 3) & 511]);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int u = 0; u < nu; ++u) {
    float* ub = base + (long long)u * 16384;                  // 64 KB per unit
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      long long off;                                           // in floats
      const int inst = wave * 8 + s;                           // 64 store instructions per unit
      if (PATTERN == 0) off = (long long)(lane >> 3) * 256 + (lane & 7) * 4 + (inst & 7) * 32 + (inst >> 3) * 2048;
      else if (PATTERN == 1) off = (long long)(lane >> 4) * 256 + (lane & 15) * 4 + (inst & 3) * 64 + (inst >> 2) * 1024;
      else off = (long long)inst * 256 + lane * 4;
      *reinterpret_cast<float4*>(ub + off) = v;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 256, nu = 32;      // fewer workgroups than CUs: the per-CU store path without the HBM write limit
  const long long per_wg = 16384ll * nu;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * per_wg * wgs);
  hipMalloc(&cyc, sizeof(unsigned long long) * wgs);
  std::vector<unsigned long long> h(wgs);
  auto run = [&](auto kern, const char* name) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 150 * 1024, 0, out, per_wg, nu, cyc);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost);
      double s = 0; for (auto c : h) s += (double)c;
      if (rep == 2) printf("%s: %.1f us per launch, %.0f cycles per unit and workgroup (64 stores of 1 KB), %.1f B/clk/CU, %.2f TB/s\n", name,
                           ms * 1e3, s / wgs / nu, 65536.0 / (s / wgs / nu), 65536.0 * nu * wgs / (ms * 1e-3) / 1e12);
    }
  };
  run(probe<0>, "8 x 128 B per instruction ");
  run(probe<1>, "4 x 256 B per instruction ");
  run(probe<2>, "1 x 1024 B per instruction");
  return 0;
}
