// Shared between m4d_sncv.hip and m4d_dscv.hip: the argument block of the SNCV kernels and the small-map SNCV body, so that
// on the coarse levels the two cost volumes of a level can share one launch (m4d_dscv_sncv_fwd).
#pragma once
#include <hip/hip_runtime.h>

namespace m4d_sncv {

struct SncvArgs {
  const float* c1; const float* c2; int h, w, C, r, d, k, nc;
  float* out; int out_stride; int th, tw; int tiles_x;
};

// Small maps (the three coarsest levels: a few hundred to a few thousand pixels, C >= 96): one lane per OUTPUT element,
// channel runs read as float4 straight from global memory (the whole map is L2 / vector-L1 resident), consecutive lanes
// = consecutive output channels of a pixel (coalesced stores).  Same arithmetic, same order as every other SNCV kernel:
// products rounded individually, summed in channel order, / NC, leaky_relu.  ``block`` / ``nblocks`` = this workgroup's
// index and the number of workgroups working on the volume (grid-stride loop).
template <int NC>
__device__ __forceinline__ void sncv_small_body(const SncvArgs& a, int total_px, long long block, long long nblocks) {
  const int mo = 2 * a.r + 1;
  const int och = mo * mo * a.k;
  const long long total = (long long)total_px * och;
  for (long long idx = block * blockDim.x + threadIdx.x; idx < total; idx += nblocks * blockDim.x) {
    const int ch = (int)(idx % och);
    const int gp = (int)(idx / och);
    const int kk = ch % a.k;
    const int dsp = ch / a.k;
    const int y = dsp / mo, x = dsp - y * mo;
    const int gx = gp % a.w;
    const int gyb = gp / a.w;                      // bi * h + gy
    const int gy = gyb % a.h;
    const int sy = gy + (y - a.r) * a.d, sx = gx + (x - a.r) * a.d;
    const bool in = sy >= 0 && sy < a.h && sx >= 0 && sx < a.w;
    const float* p1 = a.c1 + (long long)gp * a.C + kk * NC;
    const float* p2 = a.c2 + ((long long)(gyb - gy + (in ? sy : gy)) * a.w + (in ? sx : gx)) * a.C + kk * NC;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < NC; c += 4) {
      const float4 u = *reinterpret_cast<const float4*>(p1 + c);
      float4 v = *reinterpret_cast<const float4*>(p2 + c);
      if (!in) v = make_float4(0.f, 0.f, 0.f, 0.f);                     // zero padding (depth_operations.py:293)
      if (c == 0) acc = u.x * v.x; else acc = acc + u.x * v.x;
      acc = acc + u.y * v.y;
      acc = acc + u.z * v.z;
      acc = acc + u.w * v.w;
    }
    const float mean = acc / (float)NC;
    a.out[(long long)gp * a.out_stride + ch] = mean > 0.f ? mean : mean * 0.1f;
  }
}

inline long long sncv_small_blocks(const SncvArgs& a, int b) {
  const long long total = (long long)b * a.h * a.w * (2 * a.r + 1) * (2 * a.r + 1) * a.k;
  long long g = (total + 255) / 256;
  return g > 256 * 16 ? 256 * 16 : g;
}

}  // namespace m4d_sncv
