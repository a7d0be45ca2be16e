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

// MO = 2r+1 known at compile time (MO > 0): the (2r+1)^2 results of a lane are kept in
// registers, then -- after a barrier, once every lane has finished reading the halo tile --
// written into the SAME LDS bytes as rows of (2r+1)^2*k floats per pixel and streamed out
// by whole waves: each store instruction covers one pixel's contiguous channel run
// (196*k bytes at r = 3) instead of 64 scattered 4-byte pieces, which is what bounded the
// first version of this kernel (L2 request rate, not bytes).  MO == 0: runtime window,
// direct per-lane stores (large windows whose results do not fit the register file).
template <int NC, int MO>
__global__ void __launch_bounds__(256)
sncv_lds_kernel(const SncvArgs a) {
  extern __shared__ __align__(16) float tile[];
  const int bi = blockIdx.y;
  int blk = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) blk = (blk & 7) * (nb >> 3) + (blk >> 3);   // one contiguous band of tiles per XCD (halo reuse in its L2)
  const int tile_y = (blk / a.tiles_x) * a.th;
  const int tile_x = (blk % a.tiles_x) * a.tw;
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
  const int mo = MO > 0 ? MO : 2 * a.r + 1;
  const bool same = (a.c1 == a.c2);
  const float n_c = (float)NC;
  const int items = a.th * a.tw * k;           // <= 256 by construction when MO > 0
  float res[MO > 0 ? MO * MO : 1];
  for (int item = threadIdx.x; item < (MO > 0 ? 256 : items); item += blockDim.x) {
    const int kk = item % k;
    const int lp = item / k;
    const int ty = lp / a.tw, tx = lp % a.tw;
    const int gy = tile_y + ty, gx = tile_x + tx;
    const bool valid = item < items && gy < a.h && gx < a.w;
    if (MO == 0 && !valid) continue;
    if (valid) {
      const long long gp = ((long long)bi * a.h + gy) * a.w + gx;
      float c1r[NC];
      if (same) {                               // c1 == c2: the pixel's own vector is already in LDS
        const float* p1 = tile + ((ty + R) * hw_t + tx + R) * CP + kk * NC;
#pragma unroll
        for (int c = 0; c < NC; c += 4) {
          const float4 v = *reinterpret_cast<const float4*>(p1 + c);
          c1r[c] = v.x; c1r[c + 1] = v.y; c1r[c + 2] = v.z; c1r[c + 3] = v.w;
        }
      } else {
        const float* p1 = a.c1 + gp * C + kk * NC;
#pragma unroll
        for (int c = 0; c < NC; c += 4) {
          const float4 v = *reinterpret_cast<const float4*>(p1 + c);
          c1r[c] = v.x; c1r[c + 1] = v.y; c1r[c + 2] = v.z; c1r[c + 3] = v.w;
        }
      }
      float* o = a.out + gp * a.out_stride + kk;
#pragma unroll
      for (int y = 0; y < mo; ++y) {
        const float* row = tile + ((ty + y * a.d) * hw_t + tx) * CP + kk * NC;
#pragma unroll
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
          const float r = mean > 0.f ? mean : mean * 0.1f;                // :311
          if (MO > 0) res[y * MO + x] = r; else o[(y * mo + x) * k] = r;
        }
      }
    }
    if (MO > 0) {
      __syncthreads();                        // every lane is done with the halo tile
      const int och = MO * MO * k;
      if (valid) {
#pragma unroll
        for (int d = 0; d < MO * MO; ++d) tile[lp * och + d * k + kk] = res[d];
      }
      __syncthreads();
      const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
      for (int p = wave; p < a.th * a.tw; p += 4) {
        const int py = p / a.tw, pxx = p % a.tw;
        const int oy = tile_y + py, ox = tile_x + pxx;
        if (oy >= a.h || ox >= a.w) continue;                             // wave-uniform
        float* o = a.out + (((long long)bi * a.h + oy) * a.w + ox) * a.out_stride;
        for (int ch = lane; ch < och; ch += 64) o[ch] = tile[p * och + ch];
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

template <int NC, int MO>
void launch_lds_mo(const SncvArgs& a, int b, size_t lds, hipStream_t s) {
  const int tiles = a.tiles_x * ((a.h + a.th - 1) / a.th);
  static bool attr_set = false;     // raising the dynamic-LDS cap is idempotent; set once per instantiation
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sncv_lds_kernel<NC, MO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((sncv_lds_kernel<NC, MO>), dim3(tiles, b), dim3(256), lds, s, a);
}

template <int NC>
void launch_lds(const SncvArgs& a, int b, size_t lds, bool staged, hipStream_t s) {
  if (staged && a.r == 3) launch_lds_mo<NC, 7>(a, b, lds, s);
  else if (staged && a.r == 2) launch_lds_mo<NC, 5>(a, b, lds, s);
  else launch_lds_mo<NC, 0>(a, b, lds, s);
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
  size_t lds = lds_bytes(th, tw);
  if (aligned && nc_ok && lds <= 160 * 1024) {
    a.th = th; a.tw = tw; a.tiles_x = (w + tw - 1) / tw;
    // register-staged, coalesced-store variant: needs one item per lane and the output rows of the
    // tile to fit in LDS (they reuse the halo tile's bytes).
    const size_t stage = (size_t)th * tw * mo * mo * nbre_cuts * sizeof(float);
    const bool staged = (search_range == 2 || search_range == 3) && th * tw * nbre_cuts <= 256 && stage <= 160 * 1024;
    if (staged && stage > lds) lds = stage;
    switch (a.nc) {
      case 4: launch_lds<4>(a, b, lds, staged, s); break;
      case 8: launch_lds<8>(a, b, lds, staged, s); break;
      case 16: launch_lds<16>(a, b, lds, staged, s); break;
      case 24: launch_lds<24>(a, b, lds, staged, s); break;
      default: launch_lds<32>(a, b, lds, staged, s); break;
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
