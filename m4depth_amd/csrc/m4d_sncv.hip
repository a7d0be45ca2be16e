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
#include <cstdlib>
#include "m4d_common.h"
#include "m4d_sncv_small.h"
#include "../../include/m4depth_hip.h"

namespace {

using m4d_sncv::SncvArgs;

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
  for (int idx = threadIdx.x; idx < hh_t * hw_t * c4n; idx += blockDim.x) {      // unconditional (clamped) loads, see sncv7_kernel
    const int c4 = idx % c4n;
    const int hp = idx / c4n;
    const int py = hp / hw_t, pxx = hp % hw_t;
    const int gy = tile_y - R + py, gx = tile_x - R + pxx;
    const bool ok = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
    const float4 v = *reinterpret_cast<const float4*>(c2b + ((long long)cy * a.w + cx) * C + c4 * 4);
    *reinterpret_cast<float4*>(tile + hp * CP + c4 * 4) = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
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

// ---------------------------------------------------------------------------------
// r = 3, dilation 1, c1 == c2 specialisation with everything known at compile time
// (channels per cut NC, cuts K, tile TW x TH with TW*TH*K = 256 lanes): every LDS access
// is ds_read/write with an immediate offset, there is no index arithmetic on runtime tile
// geometry, the products are formed on register pairs (v_pk_mul_f32), and the workgroup is
// PERSISTENT: it walks a band of tiles and prefetches the next tile's halo into registers
// while the current one is being correlated and written out, so the global-load latency
// (which bounded the one-tile-per-workgroup version at three workgroups per CU) is hidden.
// Arithmetic: products rounded individually, summed in channel order, / NC, leaky_relu --
// identical to the oracle, bit for bit.
typedef float sncv_f2 __attribute__((ext_vector_type(2)));

// YS > 1 splits the 7 window rows of an item over YS lane groups (256 * YS threads per workgroup): at batch 1 a level has
// fewer (pixel, cut) items than the chip has lanes (levels 1-3: 1-2 waves per SIMD), and a lane's 49 sequential channel
// sums are one long latency chain; with YS = 4 a lane owns 2 rows (14 results) and four times as many waves are in flight.
template <int NC, int K, int TW, int TH, int YS>
__global__ void __launch_bounds__(256 * YS)
sncv7_kernel(const float* __restrict__ c, int b, int h, int w, float* __restrict__ out, int out_stride,
             int tiles_x, int tiles_y, int tiles_per_wg_band) {
  constexpr int R = 3, MO = 7;
  constexpr int NT = 256 * YS;                        // threads per workgroup
  constexpr int RPL = (MO + YS - 1) / YS;             // window rows per lane
  constexpr int C = NC * K, CP = C + 4, C4 = C / 4;
  constexpr int HWT = TW + 2 * R, HHT = TH + 2 * R;
  constexpr int P = TW * TH;
  constexpr int HALO_F4 = HHT * HWT * C4;
  constexpr int U = (HALO_F4 + NT - 1) / NT;          // float4 loads per lane to stage one halo
  constexpr int OCH = MO * MO * K;
  constexpr int ROUNDS = (P * OCH + NT - 1) / NT;     // store rounds
  static_assert(P * K == 256, "one (pixel, cut) item per lane group");
  extern __shared__ __align__(16) float tile[];       // halo tile, later re-used as the output stage
  const int t = threadIdx.x;
  const int item = t & 255;
  const int ys = YS > 1 ? __builtin_amdgcn_readfirstlane(t >> 8) : 0;        // wave-uniform -> scalar
  const int tiles_img = tiles_x * tiles_y;
  const long long total_tiles = (long long)tiles_img * b;
  // XCD banding: workgroup g runs on XCD g % 8 and owns every (gridDim/8)-th tile of band g % 8
  const int nwg = gridDim.x;
  const int band = blockIdx.x & 7, lane_in_band = blockIdx.x >> 3, wg_per_band = nwg >> 3;
  const long long band_len = (total_tiles + 7) / 8;
  const long long band_lo = band * band_len;
  const long long band_hi = band_lo + band_len < total_tiles ? band_lo + band_len : total_tiles;

  const int kk = item % K, lp = item / K;
  const int ty = lp / TW, tx = lp % TW;

  // The halo loads are UNCONDITIONAL (clamped coordinates, validity applied at the commit): a load under a branch makes
  // hipcc wait for it at the merge.
  float4 pre[U];
  unsigned pre_ok = 0;
  auto issue_loads = [&](long long tile_id) {
    const int bi = (int)(tile_id / tiles_img);
    const int tl = (int)(tile_id - (long long)bi * tiles_img);
    const int y0 = (tl / tiles_x) * TH - R, x0 = (tl % tiles_x) * TW - R;
    const float* img = c + (long long)bi * h * w * C;
    pre_ok = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = min(u * NT + t, HALO_F4 - 1);
      const int hp = idx / C4, c4 = idx % C4;
      const int py = hp / HWT, pxx = hp % HWT;
      const int gy = y0 + py, gx = x0 + pxx;
      const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
      pre_ok |= ok ? (1u << u) : 0u;
      const int cy = min(max(gy, 0), h - 1), cx = min(max(gx, 0), w - 1);
      pre[u] = *reinterpret_cast<const float4*>(img + ((long long)cy * w + cx) * C + c4 * 4);
    }
  };
  auto commit_loads = [&]() {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = u * NT + t;
      const int hp = idx / C4, c4 = idx % C4;
      const bool ok = (pre_ok >> u) & 1u;
      const float4 v = pre[u];
      if (idx < HALO_F4)                                                            // zero padding (:293)
        *reinterpret_cast<float4*>(tile + hp * CP + c4 * 4) = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
  };

  long long cur = band_lo + lane_in_band;
  if (cur >= band_hi) return;
  issue_loads(cur);
  commit_loads();
  __syncthreads();
  while (true) {
    const long long nxt = cur + wg_per_band;
    const bool has_next = nxt < band_hi;
    if (has_next) issue_loads(nxt);                    // in flight during the whole body below

    const int bi = (int)(cur / tiles_img);
    const int tl = (int)(cur - (long long)bi * tiles_img);
    const int tile_y = (tl / tiles_x) * TH, tile_x = (tl % tiles_x) * TW;

    // ---- correlate: lane = (pixel, cut, row group); c1 = the pixel's own vector (tile centre)
    const float* base = tile + (ty * HWT + tx) * CP + kk * NC;
    sncv_f2 c1p[NC / 2];
#pragma unroll
    for (int cc = 0; cc < NC; cc += 4) {
      const float4 v = *reinterpret_cast<const float4*>(base + (R * HWT + R) * CP + cc);
      c1p[cc / 2] = sncv_f2{v.x, v.y};
      c1p[cc / 2 + 1] = sncv_f2{v.z, v.w};
    }
    float res[RPL * MO];
    const int y_lo = ys * RPL;
#pragma unroll
    for (int yy = 0; yy < RPL; ++yy) {
      const int y = y_lo + yy;
      if (YS > 1 && MO % YS != 0 && y >= MO) break;    // uneven split: the last group has fewer rows (wave-uniform)
#pragma unroll
      for (int x = 0; x < MO; ++x) {
        float acc = 0.f;
#pragma unroll
        for (int cc = 0; cc < NC; cc += 4) {
          const float4 v = *reinterpret_cast<const float4*>(base + (y * HWT + x) * CP + cc);
          const sncv_f2 pa = c1p[cc / 2] * sncv_f2{v.x, v.y};
          const sncv_f2 pb = c1p[cc / 2 + 1] * sncv_f2{v.z, v.w};
          if (cc == 0) acc = pa.x; else acc = acc + pa.x;
          acc = acc + pa.y; acc = acc + pb.x; acc = acc + pb.y;
        }
        const float mean = acc / (float)NC;                             // :308
        res[yy * MO + x] = fmaxf(mean, mean * 0.1f);                    // leaky_relu(0.1) == max(x, 0.1 x) (:311)
      }
    }
    __syncthreads();                                   // every lane is done reading the halo
    // ---- output rows of the tile, [pixel][(y*7+x)*K + kk], in the same LDS bytes
#pragma unroll
    for (int yy = 0; yy < RPL; ++yy) {
      if (y_lo + yy < MO) {
#pragma unroll
        for (int x = 0; x < MO; ++x) tile[lp * OCH + ((y_lo + yy) * MO + x) * K + kk] = res[yy * MO + x];
      }
    }
    __syncthreads();
#pragma unroll 7
    for (int it = 0; it < ROUNDS; ++it) {
      const int e = it * NT + t;
      const int pxl = e / OCH, ch = e % OCH;
      const int oy = tile_y + pxl / TW, ox = tile_x + pxl % TW;
      if (e < P * OCH && oy < h && ox < w)
        out[(((long long)bi * h + oy) * w + ox) * out_stride + ch] = tile[e];
    }
    if (!has_next) break;
    __syncthreads();                                   // stage fully read before the halo overwrites it
    commit_loads();
    __syncthreads();
    cur = nxt;
  }
}

template <int NC, int K, int TW, int TH, int YS>
bool launch_sncv7_ys(const float* c, int b, int h, int w, float* out, int out_stride, hipStream_t s) {
  constexpr int C = NC * K, CP = C + 4, HWT = TW + 6, HHT = TH + 6, OCH = 49 * K;
  constexpr size_t halo = (size_t)HHT * HWT * CP * sizeof(float);
  constexpr size_t stage = (size_t)TW * TH * OCH * sizeof(float);
  constexpr size_t lds = halo > stage ? halo : stage;
  static_assert(lds <= 160 * 1024, "tile does not fit LDS");
  const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH;
  const long long total = (long long)tiles_x * tiles_y * b;
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 3) per_cu = 3;
  if (per_cu * YS > 8) per_cu = 8 / YS;                             // 2048 threads per CU
  long long nwg = 256LL * (per_cu > 0 ? per_cu : 1);                // persistent: one resident wave of workgroups
  if (nwg > total) nwg = (total + 7) / 8 * 8;
  if (nwg < 8) nwg = 8;
  M4D_LDS_OPT_IN(&sncv7_kernel<NC, K, TW, TH, YS>);
  m4d_launch((sncv7_kernel<NC, K, TW, TH, YS>), dim3((int)nwg), dim3(256 * YS), lds, s, c, b, h, w, out, out_stride,
                     tiles_x, tiles_y, 0);
  return true;
}

// Row split by how many (pixel, cut) items the launch has: enough lanes for ~8 waves per SIMD without a split -> YS = 1.
template <int NC, int K, int TW, int TH>
bool launch_sncv7(const float* c, int b, int h, int w, float* out, int out_stride, hipStream_t s) {
  static int ys_env = -1;                             // M4D_SNCV_YS = 1 | 2 | 4 forces the split (profiling)
  if (ys_env < 0) { const char* e = getenv("M4D_SNCV_YS"); ys_env = e ? atoi(e) : 0; }
  const long long items = (long long)b * h * w * K;
  int ys = ys_env ? ys_env : (items >= 1000000 ? 1 : (items >= 400000 ? 2 : 4));
  if (NC > 16 && ys > 2) ys = 2;                       // 32 channels per cut do not fit the 128 VGPRs of a 1024-thread workgroup
  if (ys == 4) return launch_sncv7_ys<NC, K, TW, TH, (NC > 16 ? 2 : 4)>(c, b, h, w, out, out_stride, s);
  if (ys == 2) return launch_sncv7_ys<NC, K, TW, TH, 2>(c, b, h, w, out, out_stride, s);
  return launch_sncv7_ys<NC, K, TW, TH, 1>(c, b, h, w, out, out_stride, s);
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

// Small maps (the three coarsest levels): the halo-tile kernel above runs a handful of workgroups that each spend most of
// their time staging a halo several times the size of their tile (19-28 us for 120-1920 pixels); m4d_sncv_small.h has the
// one-lane-per-output body used instead (3-6 us).
template <int NC>
__global__ void __launch_bounds__(256)
sncv_small_kernel(const SncvArgs a, int total_px) { m4d_sncv::sncv_small_body<NC>(a, total_px, blockIdx.x, gridDim.x); }

template <int NC>
void launch_small(const SncvArgs& a, int b, hipStream_t s) {
  m4d_launch((sncv_small_kernel<NC>), dim3((int)m4d_sncv::sncv_small_blocks(a, b)), dim3(256), 0, s, a, b * a.h * a.w);
}

template <int NC, int MO>
void launch_lds_mo(const SncvArgs& a, int b, size_t lds, hipStream_t s) {
  const int tiles = a.tiles_x * ((a.h + a.th - 1) / a.th);
  M4D_LDS_OPT_IN(&sncv_lds_kernel<NC, MO>);
  m4d_launch((sncv_lds_kernel<NC, MO>), dim3(tiles, b), dim3(256), lds, s, a);
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
  static int variant = -1;                  // M4D_SNCV_VARIANT=0 disables the specialised kernels (debugging)
  if (variant < 0) { const char* e = getenv("M4D_SNCV_VARIANT"); variant = e ? atoi(e) : 1; }
  static int small_px = -1;                 // M4D_SNCV_SMALL_PX: maps up to this many pixels (batch included) take the small-map kernel
  if (small_px < 0) { const char* e = getenv("M4D_SNCV_SMALL_PX"); small_px = e ? atoi(e) : 6000; }
  if (variant == 1 && aligned && nc_ok && (long long)b * h * w <= small_px) {
    a.th = a.tw = a.tiles_x = 0;
    switch (a.nc) {
      case 4: launch_small<4>(a, b, s); break;
      case 8: launch_small<8>(a, b, s); break;
      case 16: launch_small<16>(a, b, s); break;
      case 24: launch_small<24>(a, b, s); break;
      default: launch_small<32>(a, b, s); break;
    }
    return M4D_LAUNCH_RESULT();
  }
  if (variant == 1 && aligned && c1 == c2 && search_range == 3 && dilation_rate == 1) {
    bool done = false;
    if (a.nc == 16 && nbre_cuts == 1) done = launch_sncv7<16, 1, 32, 8>(c1, b, h, w, out, out_stride, s);
    else if (a.nc == 16 && nbre_cuts == 2) done = launch_sncv7<16, 2, 16, 8>(c1, b, h, w, out, out_stride, s);
    else if (a.nc == 32 && nbre_cuts == 2) done = launch_sncv7<32, 2, 16, 8>(c1, b, h, w, out, out_stride, s);
    // coarser levels (C >= 96: a few hundred pixels per image) stay on the generic LDS kernel below:
    // the persistent kernel's 250+ VGPRs buy nothing there (measured slower).
    if (done) return M4D_LAUNCH_RESULT();
  }
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
    m4d_launch(sncv_generic_kernel, dim3((int)g), dim3(256), 0, s, a, total);
  }
  return M4D_LAUNCH_RESULT();
}
