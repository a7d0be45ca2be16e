// SNCV -- the spatial-neighbourhood cost volume (cost_volume,
// utils/depth_operations.py:284-313).
//
// The reference pads c2, transposes both maps to NCHW and then issues
// (2r+1)^2 * k separate slice / multiply / reduce_mean ops.  Here a workgroup
// stages one (TH+2R) x (TW+2R) halo tile of c2 (all channels, zero-filled outside
// the image, pixel stride padded to C+4 floats so that 16-byte LDS reads of
// neighbouring pixels fall on distinct bank slots) ONCE in LDS; each lane owns one
// (pixel, cut), keeps its c1 channel run in registers and sweeps the (2r+1)^2
// window out of LDS with ds_read_b128.  HBM traffic is the algorithmic minimum
// plus the halo: read C floats (x halo factor) and write (2r+1)^2 * k floats per
// pixel.
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

struct SncvArgs {
  const float* c1; const float* c2; int h, w, C, r, d, k, nc;
  float* out; int out_stride; int th, tw; int tiles_x;
};

template <int NC>
__global__ void __launch_bounds__(256)
sncv_lds_kernel(const SncvArgs a) {
  extern __shared__ __align__(16) float tile[];
  const int bi = blockIdx.y;
  const int tile_y = (blockIdx.x / a.tiles_x) * a.th;
  const int tile_x = (blockIdx.x % a.tiles_x) * a.tw;
  const int R = a.r * a.d;
  const int hw_t = a.tw + 2 * R;              // halo tile width (pixels)
  const int hh_t = a.th + 2 * R;
  const int C = a.C, CP = C + 4;
  const int c4n = C >> 2;
  const float* c2b = a.c2 + (long long)bi * a.h * a.w * C;

  // ---- stage the halo tile: coalesced float4 rows, zero outside the image (:293)
  for (int idx = threadIdx.x; idx < hh_t * hw_t * c4n; idx += blockDim.x) {
    const int c4 = idx % c4n;
    const int hp = idx / c4n;
    const int py = hp / hw_t, pxx = hp % hw_t;
    const int gy = tile_y - R + py, gx = tile_x - R + pxx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w)
      v = *reinterpret_cast<const float4*>(c2b + ((long long)gy * a.w + gx) * C + c4 * 4);
    *reinterpret_cast<float4*>(tile + hp * CP + c4 * 4) = v;
  }
  __syncthreads();

  const int k = a.k;
  const int mo = 2 * a.r + 1;
  const bool same = (a.c1 == a.c2);
  const float n_c = (float)NC;
  for (int item = threadIdx.x; item < a.th * a.tw * k; item += blockDim.x) {
    const int kk = item % k;
    const int lp = item / k;
    const int ty = lp / a.tw, tx = lp % a.tw;
    const int gy = tile_y + ty, gx = tile_x + tx;
    if (gy >= a.h || gx >= a.w) continue;
    const long long gp = ((long long)bi * a.h + gy) * a.w + gx;
    float c1r[NC];
    if (same) {
      const float* p = tile + ((ty + R) * hw_t + tx + R) * CP + kk * NC;
#pragma unroll
      for (int c = 0; c < NC; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + c);
        c1r[c] = v.x; c1r[c + 1] = v.y; c1r[c + 2] = v.z; c1r[c + 3] = v.w;
      }
    } else {
      const float* p = a.c1 + gp * C + kk * NC;
#pragma unroll
      for (int c = 0; c < NC; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + c);
        c1r[c] = v.x; c1r[c + 1] = v.y; c1r[c + 2] = v.z; c1r[c + 3] = v.w;
      }
    }
    float* o = a.out + gp * a.out_stride + kk;
    for (int y = 0; y < mo; ++y) {
      const float* row = tile + ((ty + y * a.d) * hw_t + tx) * CP + kk * NC;
      for (int x = 0; x < mo; ++x) {
        const float* p = row + x * a.d * CP;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NC; c += 4) {
          const float4 v = *reinterpret_cast<const float4*>(p + c);
          if (c == 0) acc = c1r[0] * v.x; else acc = acc + c1r[c] * v.x;
          acc = acc + c1r[c + 1] * v.y;
          acc = acc + c1r[c + 2] * v.z;
          acc = acc + c1r[c + 3] * v.w;
        }
        const float mean = acc / n_c;                                   // :308
        o[(y * mo + x) * k] = mean > 0.f ? mean : mean * 0.1f;          // :311
      }
    }
  }
}

// Any C / k / alignment / window: one lane per output element, global reads.
__global__ void __launch_bounds__(256)
sncv_generic_kernel(const SncvArgs a, long long total) {
  const int mo = 2 * a.r + 1;
  const int och = mo * mo * a.k;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % och);
    const long long gp = idx / och;
    const int kk = ch % a.k;
    const int dsp = ch / a.k;
    const int y = dsp / mo, x = dsp % mo;
    const int gx = (int)(gp % a.w);
    const int gy = (int)((gp / a.w) % a.h);
    const long long bi = gp / ((long long)a.w * a.h);
    const int sy = gy + (y - a.r) * a.d, sx = gx + (x - a.r) * a.d;
    const bool in = sy >= 0 && sy < a.h && sx >= 0 && sx < a.w;
    const float* p1 = a.c1 + gp * a.C + kk * a.nc;
    const float* p2 = a.c2 + ((bi * a.h + (in ? sy : 0)) * a.w + (in ? sx : 0)) * (long long)a.C + kk * a.nc;
    float acc = 0.f;
    for (int c = 0; c < a.nc; ++c) {
      const float pr = p1[c] * (in ? p2[c] : 0.0f);
      if (c == 0) acc = pr; else acc = acc + pr;
    }
    const float mean = acc / (float)a.nc;
    a.out[gp * a.out_stride + ch] = mean > 0.f ? mean : mean * 0.1f;
  }
}

template <int NC>
void launch_lds(const SncvArgs& a, int b, size_t lds, hipStream_t s) {
  const int tiles = a.tiles_x * ((a.h + a.th - 1) / a.th);
  static bool attr_set = false;     // raising the dynamic-LDS cap is idempotent; set once per instantiation
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sncv_lds_kernel<NC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(sncv_lds_kernel<NC>, dim3(tiles, b), dim3(256), lds, s, a);
}

}  // namespace

extern "C" int m4d_sncv_fwd(const float* c1, const float* c2, int b, int h, int w, int C, int search_range,
                            int dilation_rate, int nbre_cuts, float* out, int out_stride, void* stream) {
  M4D_CHECK_ARG(c1 && c2 && out);
  M4D_CHECK_ARG(b > 0 && h > 0 && w > 0 && C > 0 && search_range >= 0 && dilation_rate >= 1 && nbre_cuts > 0);
  M4D_CHECK_ARG(C % nbre_cuts == 0);
  const int mo = 2 * search_range + 1;
  M4D_CHECK_ARG(out_stride >= mo * mo * nbre_cuts);
  SncvArgs a;
  a.c1 = c1; a.c2 = c2; a.h = h; a.w = w; a.C = C; a.r = search_range; a.d = dilation_rate;
  a.k = nbre_cuts; a.nc = C / nbre_cuts; a.out = out; a.out_stride = out_stride;
  hipStream_t s = (hipStream_t)stream;
  const int R = search_range * dilation_rate;
  const bool aligned = (((uintptr_t)c1 | (uintptr_t)c2) & 15u) == 0;
  const bool nc_ok = a.nc == 4 || a.nc == 8 || a.nc == 16 || a.nc == 24 || a.nc == 32;
  // Tile choice: 256 lanes = TH*TW*k items when possible, shrink until the halo fits in LDS.
  int tw = 32, th = 8;
  while (tw * th * nbre_cuts > 256 && tw > 4) tw >>= 1;
  while (tw * th * nbre_cuts > 256 && th > 2) th >>= 1;
  const size_t budget = 64 * 1024;           // two workgroups per CU
  auto lds_bytes = [&](int th_, int tw_) { return (size_t)(th_ + 2 * R) * (tw_ + 2 * R) * (C + 4) * sizeof(float); };
  while (lds_bytes(th, tw) > budget && (tw > 8 || th > 4)) { if (tw > th * 2 || th <= 4) tw >>= 1; else th >>= 1; }
  const size_t lds = lds_bytes(th, tw);
  if (aligned && nc_ok && lds <= 160 * 1024) {
    a.th = th; a.tw = tw; a.tiles_x = (w + tw - 1) / tw;
    switch (a.nc) {
      case 4: launch_lds<4>(a, b, lds, s); break;
      case 8: launch_lds<8>(a, b, lds, s); break;
      case 16: launch_lds<16>(a, b, lds, s); break;
      case 24: launch_lds<24>(a, b, lds, s); break;
      default: launch_lds<32>(a, b, lds, s); break;
    }
  } else {
    a.th = a.tw = a.tiles_x = 0;
    const long long total = (long long)b * h * w * mo * mo * nbre_cuts;
    long long g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    hipLaunchKernelGGL(sncv_generic_kernel, dim3((int)g), dim3(256), 0, s, a, total);
  }
  return M4D_LAUNCH_RESULT();
}
