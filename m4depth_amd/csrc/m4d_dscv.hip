// DSCV -- the parallax-sweeping cost volume (get_parallax_sweeping_cv,
// utils/depth_operations.py:224-281), fused.
//
// The reference materialises (2r+1) tiled copies of c1, of concat[c2, disp_prev_t]
// and of the flow field, warps them with 4 tf.gather passes and then multiplies /
// reduces in float16: ~17x the algorithmic HBM traffic.  Here
// nothing but c1, c2, the two parallax maps and the outputs touches HBM.
//
// Wave layout (dscv_wave_kernel): the C channels of a pixel are spread over LP = C/4
// adjacent lanes, 16 bytes each, so a wave covers 64/LP neighbouring pixels and every
// corner fetch of a pixel is ONE contiguous 4*C-byte run (one or two cache lines)
// instead of C/4 strided 16-byte pieces.  The 2r+1 query points of a pixel are
// computed once, hypothesis t by lane t mod LP of the pixel, and broadcast with wave
// shuffles; the per-cut float16 products are summed in channel order by a shuffle
// chain over the G = LP/k lanes of the cut (same order as the sequential oracle, so
// the result is bit-identical).  Workgroups are remapped so that each XCD sweeps a
// contiguous band of the image: the gathers of neighbouring pixels then hit the
// XCD's own 4 MiB L2.
#include "m4d_common.h"
#include "m4d_sncv_small.h"
#include "m4d_level_pre.h"
#include "../../include/m4depth_hip.h"

namespace {

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

struct DscvArgs {
  const float* c1; const float* c2; const float* disp_prev_t; const float* disp;
  const float* rot; int rot_c; const float* trans; const float* cam_f; const float* cam_c;
  int h, w, C, r, k, nc, cv_accum;
  float* cv; int cv_stride; float* prev_disp; float* log_center; int log_stride; float log_scale;
  int32_t* index_out;
  int ablate;            // profiling only (m4d_dscv_set_ablation): 1 = no cv stores, 2 = no corner loads, 4 = no centre feature
  // ONFLY instantiations only (m4d_level_front_small: the level's opening glue is not a launch of its own, so its two maps do
  // not exist): disp = 2 x the x2 upsampling of the coarser level's parallax (pl_para [b,ph,pw], NULL at the coarsest level: 1),
  // disp_prev_t at a corner = prev_d2para of the depth memory there -- the expressions of level_pre_body, evaluated in place
  const float* pl_para = nullptr; int ph = 0, pw = 0; const float* depth_prev_t = nullptr;
};

__device__ __forceinline__ void dscv_query(const M4dPixel& px, float start_x, float start_y, float disp,
                                           int i, int j, int t, int r, float& qy, float& qx) {
  const float n = (float)(t - r);
  const float p = fminf(fmaxf(disp + n, 1e-6f), 1e6f);      // :235-236
  const float divider = px.s / p;                           // :262
  const float dxx = px.delta_x / divider;                   // :263
  const float dyy = px.delta_y / divider;
  const float flow_x = (px.proj_x + dxx) - start_x;         // :264
  const float flow_y = (px.proj_y + dyy) - start_y;
  qy = (float)j + flow_y;                                   // dense_image_warp.py:244
  qx = (float)i + flow_x;
}

// Corner sources for dscv_pixel: the NHWC previous-frame features in HBM/L2, or the
// window staged in LDS by dscv_tile_kernel (pixel stride CP = C + 4 floats).
struct DscvGlobalFetch {
  const float* base;       // image base + 4*q
  int w, C; long long rs;
  __device__ __forceinline__ void operator()(int y0, int x0, float4& tl, float4& tr, float4& bl, float4& br) const {
    const float* p = base + ((long long)y0 * w + x0) * C;
    tl = *reinterpret_cast<const float4*>(p);
    tr = *reinterpret_cast<const float4*>(p + C);
    bl = *reinterpret_cast<const float4*>(p + rs);
    br = *reinterpret_cast<const float4*>(p + rs + C);
  }
};
struct DscvLdsFetch {
  const float* base;       // window base + 4*q
  int ymin, xmin, WW, CP;
  __device__ __forceinline__ void operator()(int y0, int x0, float4& tl, float4& tr, float4& bl, float4& br) const {
    const float* p = base + ((y0 - ymin) * WW + (x0 - xmin)) * CP;
    tl = *reinterpret_cast<const float4*>(p);
    tr = *reinterpret_cast<const float4*>(p + CP);
    bl = *reinterpret_cast<const float4*>(p + WW * CP);
    br = *reinterpret_cast<const float4*>(p + WW * CP + CP);
  }
};

// One pixel slot of a wave: LP lanes per pixel (C = 4*LP channels, 16 bytes per lane), G
// lanes per cut (nc = 4*G), NCP = 2r+1 hypotheses (compile time).  The kernels built on
// this are latency-bound rather than byte-bound, so the body is written for memory-level
// parallelism: the query points are computed once (hypothesis t by lane t mod LP) and
// exchanged with shuffles, the 4 corner loads of HB hypotheses are issued together, and
// the channel-order cross-lane sums of those HB hypotheses run as G-1 rounds of HB
// independent shuffles.  A compiler memory barrier closes each batch so that the loads of
// later batches are not hoisted (which costs >190 VGPRs and the occupancy with it).
template <int LP, int G, int NCP, class Fetch, bool ONFLY = false>
__device__ __forceinline__ void dscv_pixel(const DscvArgs& a, const M4dMotion& m, int bi, int i, int j, bool active,
                                           int q, int g, int kk, int base_lane, const Fetch& fetch) {
  constexpr int J = (NCP + LP - 1) / LP;       // hypotheses computed by each lane
  constexpr int NC = 4 * G;
  constexpr int HB = 3;                        // hypotheses in flight per batch
  constexpr int C = 4 * LP;
  const int r = (NCP - 1) / 2;
  const int hw = a.h * a.w;
  const long long gp = (long long)bi * hw + (long long)j * a.w + i;
  const M4dPixel px = m4d_pixel_factors(m, i, j);
  const float start_x = px.x * m.fx;           // :256
  const float start_y = px.y * m.fy;
  float disp;
  if (ONFLY) {                                  // para_prev_l of level_pre_body (m4depth_network.py:198, 203), same expressions
    disp = 1.0f;
    if (a.pl_para != nullptr) {
      const ResizeAxis ya = resize_axis(j, (float)a.ph / (float)a.h, a.ph), xa = resize_axis(i, (float)a.pw / (float)a.w, a.pw);
      disp = resize_sample(a.pl_para + (long long)bi * a.ph * a.pw, a.pw, 1, 0, ya, xa) * 2.0f;
    }
  } else {
    disp = a.disp[gp];
  }
  float oqy[J], oqx[J];
#pragma unroll
  for (int jj = 0; jj < J; ++jj) {
    const int t = jj * LP + q;
    dscv_query(px, start_x, start_y, disp, i, j, t < NCP ? t : 0, r, oqy[jj], oqx[jj]);
  }
  // this lane's 4 channels of c1, pre-rounded to half (:276)
  const float4 c1v = *reinterpret_cast<const float4*>(a.c1 + gp * C + 4 * q);
  const float c1a = m4d_round_half(c1v.x), c1b = m4d_round_half(c1v.y);
  const float c1c = m4d_round_half(c1v.z), c1d = m4d_round_half(c1v.w);
  const float* dpt = (ONFLY ? a.depth_prev_t : a.disp_prev_t) + (long long)bi * hw;
  const bool seq16 = a.cv_accum != 0;
  float* o = a.cv + gp * a.cv_stride + kk * NCP;

#pragma unroll
  for (int tb = 0; tb < NCP; tb += HB) {
    float4 vtl[HB], vtr[HB], vbl[HB], vbr[HB];
    float ay[HB], ax[HB];
    int y0[HB], x0[HB];
#pragma unroll
    for (int u = 0; u < HB; ++u) {
      if (tb + u < NCP) {
        const int t = tb + u;
        const float qy = __shfl(oqy[t / LP], base_lane + (t % LP));
        const float qx = __shfl(oqx[t / LP], base_lane + (t % LP));
        m4d_bilinear_axis(qy, a.h, y0[u], ay[u]);
        m4d_bilinear_axis(qx, a.w, x0[u], ax[u]);
        fetch(y0[u], x0[u], vtl[u], vtr[u], vbl[u], vbr[u]);
      }
    }
    float part[HB][4], acc[HB];
#pragma unroll
    for (int u = 0; u < HB; ++u) {
      if (tb + u < NCP) {
        const int t = tb + u;
        part[u][0] = m4d_round_half(c1a * m4d_round_half(m4d_lerp2(vtl[u].x, vtr[u].x, vbl[u].x, vbr[u].x, ax[u], ay[u])));
        part[u][1] = m4d_round_half(c1b * m4d_round_half(m4d_lerp2(vtl[u].y, vtr[u].y, vbl[u].y, vbr[u].y, ax[u], ay[u])));
        part[u][2] = m4d_round_half(c1c * m4d_round_half(m4d_lerp2(vtl[u].z, vtr[u].z, vbl[u].z, vbr[u].z, ax[u], ay[u])));
        part[u][3] = m4d_round_half(c1d * m4d_round_half(m4d_lerp2(vtl[u].w, vtr[u].w, vbl[u].w, vbr[u].w, ax[u], ay[u])));
        acc[u] = !seq16 ? ((part[u][0] + part[u][1]) + part[u][2]) + part[u][3]
                        : m4d_round_half(m4d_round_half(m4d_round_half(part[u][0] + part[u][1]) + part[u][2]) + part[u][3]);
        if (active && q == 0) {
          if (a.index_out) {
            a.index_out[(gp * NCP + t) * 2] = y0[u];
            a.index_out[(gp * NCP + t) * 2 + 1] = x0[u];
          }
          const bool centre = (t == r) && a.log_center != nullptr;
          if (a.prev_disp != nullptr || centre) {
            const float* d0 = dpt + (long long)y0[u] * a.w + x0[u];                 // the extra channel of :268
            float c00 = d0[0], c01 = d0[1], c10 = d0[a.w], c11 = d0[a.w + 1];
            if (ONFLY) {                                                            // prev_d2para (:218) at the four corners
              c00 = m4d_prev_d2para_px(m, c00, x0[u], y0[u]);     c01 = m4d_prev_d2para_px(m, c01, x0[u] + 1, y0[u]);
              c10 = m4d_prev_d2para_px(m, c10, x0[u], y0[u] + 1); c11 = m4d_prev_d2para_px(m, c11, x0[u] + 1, y0[u] + 1);
            }
            const float wd = m4d_lerp2(c00, c01, c10, c11, ax[u], ay[u]);
            if (a.prev_disp) a.prev_disp[gp * NCP + t] = wd;
            if (centre) a.log_center[gp * a.log_stride] = logf(wd * a.log_scale);   // m4depth_network.py:238
          }
        }
      }
    }
    // sequential (channel-order) sums across the G lanes of the cut
#pragma unroll
    for (int s = 1; s < G; ++s) {
      float prev[HB];
#pragma unroll
      for (int u = 0; u < HB; ++u) if (tb + u < NCP) prev[u] = __shfl_up(acc[u], 1);
      if (g == s) {
#pragma unroll
        for (int u = 0; u < HB; ++u)
          if (tb + u < NCP)
            acc[u] = !seq16 ? (((prev[u] + part[u][0]) + part[u][1]) + part[u][2]) + part[u][3]
                            : m4d_round_half(m4d_round_half(m4d_round_half(m4d_round_half(prev[u] + part[u][0]) + part[u][1]) + part[u][2]) + part[u][3]);
      }
    }
    if (active && g == G - 1) {
#pragma unroll
      for (int u = 0; u < HB; ++u)
        if (tb + u < NCP) o[tb + u] = m4d_round_half(acc[u] / (float)NC);            // :277-278
    }
    asm volatile("" ::: "memory");              // keep the next batch's loads below this point
  }
}

// Wave kernel: pixels in raster order, 64/LP per wave, corners gathered through L2/L1.
template <int LP, int G, int NCP, bool ONFLY = false>
__device__ __forceinline__ void dscv_wave_body(const DscvArgs& a, int blk, int nb, int bi) {
  constexpr int PPW = 64 / LP;                 // pixels per wave
  constexpr int C = 4 * LP;
  const int hw = a.h * a.w;
  // XCD-aware remap: workgroup b runs on XCD b % 8; give each XCD a contiguous band.
  if ((nb & 7) == 0) blk = (blk & 7) * (nb >> 3) + (blk >> 3);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slot = lane / LP;                  // pixel slot in the wave
  const int q = lane - slot * LP;              // float4 index inside the pixel's feature vector
  int pix = (blk * 4 + wave) * PPW + slot;
  const bool active = slot < PPW && pix < hw;
  if (!active) pix = hw - 1;                   // keep addresses valid; results are discarded
  const M4dMotion m = m4d_load_motion(a.rot, a.rot_c, a.trans, a.cam_f, a.cam_c, bi);
  DscvGlobalFetch fetch;
  fetch.base = a.c2 + (long long)bi * hw * C + 4 * q;
  fetch.w = a.w; fetch.C = C; fetch.rs = (long long)a.w * C;
  dscv_pixel<LP, G, NCP, DscvGlobalFetch, ONFLY>(a, m, bi, pix % a.w, pix / a.w, active, q, q % G, q / G, slot * LP, fetch);
}

template <int LP, int G, int NCP>
__global__ void __launch_bounds__(256)
dscv_wave_kernel(const DscvArgs a) { dscv_wave_body<LP, G, NCP>(a, blockIdx.x, gridDim.x, blockIdx.y); }

// Coarse levels: the two cost volumes of a level are independent and both tiny -- one launch for both (one kernel boundary
// less on the latency chain): the first nb * b workgroups run the DSCV wave kernel, the rest the small-map SNCV body
// (channels per cut = 4 * G).
template <int LP, int G, int NCP>
__global__ void __launch_bounds__(256)
dscv_sncv_small_kernel(const DscvArgs a, int nb, int b, const m4d_sncv::SncvArgs sa, int total_px, int nb_sncv) {
  const int blk = blockIdx.x;
  if (blk < nb * b) dscv_wave_body<LP, G, NCP>(a, blk % nb, nb, blk / nb);
  else m4d_sncv::sncv_small_body<4 * G>(sa, total_px, blk - nb * b, nb_sncv);
}

// ... and with the level's opening glue as well (m4d_level_front_small): [DSCV | SNCV | level_pre] workgroups in ONE launch.  The
// features arrive per-cut normalised (m4d_normalize_levels, off the level's latency chain), the DSCV evaluates the two maps
// level_pre would have written (para_prev_l, para_prev_t) where it needs them, level_pre's workgroups write the log / memory
// features of the refiner input.  Nothing depends on anything else inside the launch.
template <int LP, int G, int NCP>
__global__ void __launch_bounds__(256)
level_front_small_kernel(const DscvArgs a, int nb, int b, const m4d_sncv::SncvArgs sa, int total_px, int nb_sncv,
                         const m4d_level::LevelPreArgs pa, int pre_gx) {
  const int blk = blockIdx.x;
  if (blk < nb * b) dscv_wave_body<LP, G, NCP, true>(a, blk % nb, nb, blk / nb);
  else if (blk < nb * b + nb_sncv) m4d_sncv::sncv_small_body<4 * G>(sa, total_px, blk - nb * b, nb_sncv);
  else m4d_level::level_pre_body(pa, (blk - nb * b - nb_sncv) % pre_gx, (blk - nb * b - nb_sncv) / pre_gx, pre_gx);
}

// Stage a WH x WW window of an NHWC image into LDS (pixel stride C+4 floats).  A window
// row is one contiguous run of WW*C floats in memory; 8 independent 16-byte loads are
// issued per lane before the first LDS store, so the copy runs at memory-level
// parallelism instead of one round trip per iteration.
template <int C>
__device__ __forceinline__ void dscv_stage_window(const float* __restrict__ img, float* __restrict__ win, int w,
                                                  int ymin, int xmin, int WH, int WW, int t) {
  constexpr int c4n = C / 4, CP = C + 4, U = 8;
  const int row_f4 = WW * c4n;
  const int total = WH * row_f4;
  for (int base = 0; base < total; base += 256 * U) {
    float4 v[U];
    int dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * 256 + t;
      dst[u] = -1;
      if (idx < total) {
        const int wy = idx / row_f4, e = idx - wy * row_f4;
        const int wx = e / c4n, c4 = e - wx * c4n;
        v[u] = *reinterpret_cast<const float4*>(img + ((long long)(ymin + wy) * w + xmin) * C + e * 4);
        dst[u] = (wy * WW + wx) * CP + c4 * 4;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (dst[u] >= 0) *reinterpret_cast<float4*>(win + dst[u]) = v[u];
  }
}

// ---------------------------------------------------------------------------------
// LDS search-window variant.  Every gather of dscv_wave_kernel goes through L2 -> L1
// (36 x 4C bytes per pixel, half of every 128-byte line unused: measured L2-throughput
// bound).  Here a workgroup owns a TW x TH tile:
//   1. lane = pixel: the two END hypotheses of each pixel (the 2r+1 query points are
//      collinear and monotone in the parallax, so their footprints are bounded by the
//      end points) give the bounding box of every gather of the tile -- one workgroup
//      min/max reduction;
//   2. if the box fits the LDS window, that window of the previous-frame features is
//      staged with coalesced 16-byte loads (each byte once; pixel stride C+4 floats so
//      that ds_read_b128 of neighbouring pixels fall on distinct bank slots);
//   3. the wave layout of dscv_wave_kernel then takes the 4 corners with ds_read_b128
//      from LDS (block-uniform fallback: the global gathers).
// Arithmetic and summation order are identical to the other variants (bit-exact).
struct DscvTileArgs {
  DscvArgs d;
  int tw, th, tiles_x, tiles, win_cap_px;
  unsigned int* fallback_counter;               // optional: counts workgroups that took the global path
  unsigned long long* stamps;                   // optional (profiling): 4 s_memtime stamps + window size per workgroup
};

template <int LP, int G, int NCP>
__global__ void __launch_bounds__(256, 3)
dscv_tile_kernel(const DscvTileArgs ta) {
  extern __shared__ __align__(16) float smem[];
  const DscvArgs& a = ta.d;
  constexpr int PPW = 64 / LP;
  constexpr int C = 4 * LP, CP = C + 4;
  const int r = (NCP - 1) / 2;
  const int bi = blockIdx.y;
  const int hw = a.h * a.w;
  int blk = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) blk = (blk & 7) * (nb >> 3) + (blk >> 3);     // XCD band
  const int tile_y = (blk / ta.tiles_x) * ta.th;
  const int tile_x = (blk % ta.tiles_x) * ta.tw;
  const int P = ta.tw * ta.th;
  int* red = reinterpret_cast<int*>(smem);             // 16 ints of reduction scratch
  float* win = smem + 16;                              // staged window (16-byte aligned)
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;

  const M4dMotion m = m4d_load_motion(a.rot, a.rot_c, a.trans, a.cam_f, a.cam_c, bi);

  // ---- phase 1: footprint bounding box from the two end hypotheses of pixel t
  int ymin = 0x7fffffff, ymax = -1, xmin = 0x7fffffff, xmax = -1;
  if (t < P) {
    const int ty = t / ta.tw, tx = t - ty * ta.tw;
    const int j = tile_y + ty, i = tile_x + tx;
    if (j < a.h && i < a.w) {
      const M4dPixel px = m4d_pixel_factors(m, i, j);
      const float start_x = px.x * m.fx, start_y = px.y * m.fy;
      const float disp = a.disp[(long long)bi * hw + (long long)j * a.w + i];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float qy, qx;
        dscv_query(px, start_x, start_y, disp, i, j, e == 0 ? 0 : NCP - 1, r, qy, qx);
        int y0, x0; float ay, ax;
        m4d_bilinear_axis(qy, a.h, y0, ay);
        m4d_bilinear_axis(qx, a.w, x0, ax);
        ymin = min(ymin, y0); ymax = max(ymax, y0);
        xmin = min(xmin, x0); xmax = max(xmax, x0);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ymin = min(ymin, __shfl_xor(ymin, o)); ymax = max(ymax, __shfl_xor(ymax, o));
    xmin = min(xmin, __shfl_xor(xmin, o)); xmax = max(xmax, __shfl_xor(xmax, o));
  }
  if (lane == 0) { red[wave * 4] = ymin; red[wave * 4 + 1] = ymax; red[wave * 4 + 2] = xmin; red[wave * 4 + 3] = xmax; }
  __syncthreads();
  ymin = min(min(red[0], red[4]), min(red[8], red[12]));
  ymax = max(max(red[1], red[5]), max(red[9], red[13]));
  xmin = min(min(red[2], red[6]), min(red[10], red[14]));
  xmax = max(max(red[3], red[7]), max(red[11], red[15]));
  const int WH = ymax - ymin + 2, WW = xmax - xmin + 2;            // footprints are 2x2
  const bool use_lds = ymax >= 0 && WH * WW <= ta.win_cap_px;      // block-uniform

  // ---- phase 2: stage the window (each byte once, coalesced rows)
  const float* c2img = a.c2 + (long long)bi * hw * C;
  if (use_lds) {
    dscv_stage_window<C>(c2img, win, a.w, ymin, xmin, WH, WW, t);
  } else if (t == 0 && ta.fallback_counter) {
    atomicAdd(ta.fallback_counter, 1u);
  }
  __syncthreads();

  // ---- phase 3: wave layout of dscv_wave_kernel, corners from LDS
  const int slot = lane / LP;
  const int q = lane - slot * LP;
  DscvLdsFetch lf;
  lf.base = win + 4 * q; lf.ymin = ymin; lf.xmin = xmin; lf.WW = WW; lf.CP = CP;
  DscvGlobalFetch gf;
  gf.base = c2img + 4 * q; gf.w = a.w; gf.C = C; gf.rs = (long long)a.w * C;
  for (int base = 0; base < P; base += 4 * PPW) {
    const int lp = base + wave * PPW + slot;
    bool active = slot < PPW && lp < P;
    int ty = 0, tx = 0;
    if (active) { ty = lp / ta.tw; tx = lp - ty * ta.tw; }
    int j = tile_y + ty, i = tile_x + tx;
    active = active && j < a.h && i < a.w;
    if (!active) { j = tile_y; i = tile_x; }                       // pixel 0 of the tile is always valid
    if (use_lds) dscv_pixel<LP, G, NCP>(a, m, bi, i, j, active, q, q % G, q / G, slot * LP, lf);
    else dscv_pixel<LP, G, NCP>(a, m, bi, i, j, active, q, q % G, q / G, slot * LP, gf);
  }
}

// ---------------------------------------------------------------------------------
// Hypothesis-per-lane variant on the LDS window.  Once the window is in LDS, coalescing
// no longer constrains the lane layout, so the cross-lane machinery of dscv_pixel (query
// exchange, shuffle-chain sums, geometry repeated by the C/4 lanes of a pixel) can go:
// a lane owns (pixel, cut, every LH-th hypothesis), reads the 4 corners of its cut's
// whole channel run with ds_read_b128 and sums the float16 products in channel order in
// its own registers -- the oracle's order by construction.  ~40 % fewer VALU instructions
// per pixel than the channels-across-lanes layout at C = 16.
template <int NC, int K, int NCP, int HPL>
__global__ void __launch_bounds__(256)
dscv_hyp_kernel(const DscvTileArgs ta) {
  extern __shared__ __align__(16) float smem[];
  const DscvArgs& a = ta.d;                     // HPL = hypotheses per lane
  constexpr int LH = (NCP + HPL - 1) / HPL;     // lanes sharing one (pixel, cut)
  constexpr int LPP = LH * K;                   // lanes per pixel
  constexpr int PPW = 64 / LPP;                 // pixels per wave
  constexpr int C = NC * K, CP = C + 4;
  constexpr int CH = NC / 4;                    // float4 chunks per cut
  const int r = (NCP - 1) / 2;
  const int bi = blockIdx.y;
  const int hw = a.h * a.w;
  int blk = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) blk = (blk & 7) * (nb >> 3) + (blk >> 3);     // XCD band
  const int tile_y = (blk / ta.tiles_x) * ta.th;
  const int tile_x = (blk % ta.tiles_x) * ta.tw;
  const int P = ta.tw * ta.th;
  int* red = reinterpret_cast<int*>(smem);
  float* win = smem + 16;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const M4dMotion m = m4d_load_motion(a.rot, a.rot_c, a.trans, a.cam_f, a.cam_c, bi);
  unsigned long long* st = ta.stamps ? ta.stamps + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 6 : nullptr;
  if (st && t == 0) st[0] = __builtin_readcyclecounter();

  // ---- phase 1: footprint bounding box from the two end hypotheses of pixel t
  int ymin = 0x7fffffff, ymax = -1, xmin = 0x7fffffff, xmax = -1;
  if (t < P) {
    const int ty = t / ta.tw, tx = t - ty * ta.tw;
    const int j = tile_y + ty, i = tile_x + tx;
    if (j < a.h && i < a.w) {
      const M4dPixel px = m4d_pixel_factors(m, i, j);
      const float start_x = px.x * m.fx, start_y = px.y * m.fy;
      const float disp = a.disp[(long long)bi * hw + (long long)j * a.w + i];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float qy, qx;
        dscv_query(px, start_x, start_y, disp, i, j, e == 0 ? 0 : NCP - 1, r, qy, qx);
        int y0, x0; float ay, ax;
        m4d_bilinear_axis(qy, a.h, y0, ay);
        m4d_bilinear_axis(qx, a.w, x0, ax);
        ymin = min(ymin, y0); ymax = max(ymax, y0);
        xmin = min(xmin, x0); xmax = max(xmax, x0);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ymin = min(ymin, __shfl_xor(ymin, o)); ymax = max(ymax, __shfl_xor(ymax, o));
    xmin = min(xmin, __shfl_xor(xmin, o)); xmax = max(xmax, __shfl_xor(xmax, o));
  }
  if (lane == 0) { red[wave * 4] = ymin; red[wave * 4 + 1] = ymax; red[wave * 4 + 2] = xmin; red[wave * 4 + 3] = xmax; }
  __syncthreads();
  ymin = min(min(red[0], red[4]), min(red[8], red[12]));
  ymax = max(max(red[1], red[5]), max(red[9], red[13]));
  xmin = min(min(red[2], red[6]), min(red[10], red[14]));
  xmax = max(max(red[3], red[7]), max(red[11], red[15]));
  const int WH = ymax - ymin + 2, WW = xmax - xmin + 2;
  const bool use_lds = ymax >= 0 && WH * WW <= ta.win_cap_px;      // block-uniform
  if (st && t == 0) { st[1] = __builtin_readcyclecounter(); st[4] = (unsigned long long)(WH * WW); st[5] = (unsigned long long)WW; }

  // ---- phase 2: stage the window
  const float* c2img = a.c2 + (long long)bi * hw * C;
  if (use_lds) {
    if (!(a.ablate & 16)) dscv_stage_window<C>(c2img, win, a.w, ymin, xmin, WH, WW, t);
  } else if (t == 0 && ta.fallback_counter) {
    atomicAdd(ta.fallback_counter, 1u);
  }
  __syncthreads();
  if (st && t == 0) st[2] = __builtin_readcyclecounter();
  if (a.ablate & 8) { if (st && t == 0) st[3] = st[2]; return; }

  // ---- phase 3: lane = (pixel slot, cut, hypothesis group)
  const int slot = lane / LPP;
  const int rem = lane - slot * LPP;
  const int kk = rem / LH;
  const int hg = rem - kk * LH;
  const float* dpt = a.disp_prev_t + (long long)bi * hw;
  const bool seq16 = a.cv_accum != 0;
  const long long rs = (long long)a.w * C;
  for (int base = 0; base < P; base += 4 * PPW) {
    const int lp = base + wave * PPW + slot;
    bool active = slot < PPW && lp < P;
    int ty = 0, tx = 0;
    if (active) { ty = lp / ta.tw; tx = lp - ty * ta.tw; }
    int j = tile_y + ty, i = tile_x + tx;
    active = active && j < a.h && i < a.w;
    if (!active) { j = tile_y; i = tile_x; }
    const long long gp = (long long)bi * hw + (long long)j * a.w + i;
    const M4dPixel px = m4d_pixel_factors(m, i, j);
    const float start_x = px.x * m.fx, start_y = px.y * m.fy;
    const float disp = a.disp[gp];
    half2_t c1p[NC / 2];                                  // this cut's c1 run as packed halves (:276)
#pragma unroll
    for (int c = 0; c < NC; c += 4) {
      const float4 v = *reinterpret_cast<const float4*>(a.c1 + gp * C + kk * NC + c);
      const float2_t lo = {v.x, v.y}, hi = {v.z, v.w};
      c1p[c / 2] = __builtin_convertvector(lo, half2_t);
      c1p[c / 2 + 1] = __builtin_convertvector(hi, half2_t);
    }
#pragma unroll
    for (int mi = 0; mi < HPL; ++mi) {
      const int hyp = hg + LH * mi;
      const bool hyp_on = hyp < NCP;                      // only relevant when NCP % HPL != 0
      float qy, qx;
      dscv_query(px, start_x, start_y, disp, i, j, hyp_on ? hyp : 0, r, qy, qx);
      int y0, x0;
      float ay, ax;
      m4d_bilinear_axis(qy, a.h, y0, ay);
      m4d_bilinear_axis(qx, a.w, x0, ax);
      float4 vtl[CH], vtr[CH], vbl[CH], vbr[CH];
      if (a.ablate & 2) {
#pragma unroll
        for (int c = 0; c < CH; ++c) { vtl[c] = vtr[c] = vbl[c] = vbr[c] = make_float4(ax, ay, ax, ay); }
      } else if (use_lds) {
        const float* p = win + ((y0 - ymin) * WW + (x0 - xmin)) * CP + kk * NC;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          vtl[c] = *reinterpret_cast<const float4*>(p + 4 * c);
          vtr[c] = *reinterpret_cast<const float4*>(p + CP + 4 * c);
          vbl[c] = *reinterpret_cast<const float4*>(p + WW * CP + 4 * c);
          vbr[c] = *reinterpret_cast<const float4*>(p + WW * CP + CP + 4 * c);
        }
      } else {
        const float* p = c2img + ((long long)y0 * a.w + x0) * C + kk * NC;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          vtl[c] = *reinterpret_cast<const float4*>(p + 4 * c);
          vtr[c] = *reinterpret_cast<const float4*>(p + C + 4 * c);
          vbl[c] = *reinterpret_cast<const float4*>(p + rs + 4 * c);
          vbr[c] = *reinterpret_cast<const float4*>(p + rs + C + 4 * c);
        }
      }
      // channel math on register pairs (v_pk_mul/add_f32, v_cvt_pk_f16_f32, v_pk_mul_f16): same
      // IEEE operations and the same sequential sum as the scalar form, half the instructions
      float acc = 0.f;
      const float2_t ax2 = {ax, ax}, ay2 = {ay, ay};
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float2_t tlA = {vtl[c].x, vtl[c].y}, trA = {vtr[c].x, vtr[c].y}, blA = {vbl[c].x, vbl[c].y}, brA = {vbr[c].x, vbr[c].y};
        const float2_t tlB = {vtl[c].z, vtl[c].w}, trB = {vtr[c].z, vtr[c].w}, blB = {vbl[c].z, vbl[c].w}, brB = {vbr[c].z, vbr[c].w};
        const float2_t topA = ax2 * (trA - tlA) + tlA, botA = ax2 * (brA - blA) + blA;
        const float2_t topB = ax2 * (trB - tlB) + tlB, botB = ax2 * (brB - blB) + blB;
        const half2_t wA = __builtin_convertvector(ay2 * (botA - topA) + topA, half2_t);
        const half2_t wB = __builtin_convertvector(ay2 * (botB - topB) + topB, half2_t);
        const half2_t pA = c1p[2 * c] * wA, pB = c1p[2 * c + 1] * wB;
        if (!seq16) {
          if (c == 0) acc = (float)pA.x; else acc = acc + (float)pA.x;
          acc = acc + (float)pA.y; acc = acc + (float)pB.x; acc = acc + (float)pB.y;
        } else {
          if (c == 0) acc = (float)pA.x; else acc = m4d_round_half(acc + (float)pA.x);
          acc = m4d_round_half(acc + (float)pA.y); acc = m4d_round_half(acc + (float)pB.x); acc = m4d_round_half(acc + (float)pB.y);
        }
      }
      if (active && hyp_on) {
        if (!(a.ablate & 1) || acc == 12345.f)
          a.cv[gp * a.cv_stride + kk * NCP + hyp] = m4d_round_half(acc / (float)NC);     // :277-278
        if (kk == 0 && !(a.ablate & 4)) {
          if (a.index_out) {
            a.index_out[(gp * NCP + hyp) * 2] = y0;
            a.index_out[(gp * NCP + hyp) * 2 + 1] = x0;
          }
          const bool centre = (hyp == r) && a.log_center != nullptr;
          if (a.prev_disp != nullptr || centre) {
            const float* d0 = dpt + (long long)y0 * a.w + x0;
            const float wd = m4d_lerp2(d0[0], d0[1], d0[a.w], d0[a.w + 1], ax, ay);
            if (a.prev_disp) a.prev_disp[gp * NCP + hyp] = wd;
            if (centre) a.log_center[gp * a.log_stride] = logf(wd * a.log_scale);
          }
        }
      }
      asm volatile("" ::: "memory");
    }
  }
  if (st && t == 0) st[3] = __builtin_readcyclecounter();
}

// Any C / cuts / alignment / search range: one lane per (pixel, cut), scalar loads.
__global__ void __launch_bounds__(256)
dscv_generic_kernel(const DscvArgs a) {
  const int bi = blockIdx.y;
  const int hw = a.h * a.w;
  const int k = a.k;
  const int t_id = blockIdx.x * blockDim.x + threadIdx.x;
  if (t_id >= hw * k) return;
  const int kk = t_id % k;
  const int pix = t_id / k;
  const int i = pix % a.w, j = pix / a.w;
  const long long gp = (long long)bi * hw + pix;
  const int nc = a.nc;
  const int C = a.C;
  const int ncp = 2 * a.r + 1;
  const M4dMotion m = m4d_load_motion(a.rot, a.rot_c, a.trans, a.cam_f, a.cam_c, bi);
  const M4dPixel px = m4d_pixel_factors(m, i, j);
  const float start_x = px.x * m.fx;
  const float start_y = px.y * m.fy;
  const float disp = a.disp[gp];
  const float* c1p = a.c1 + gp * C + kk * nc;
  const float* c2b = a.c2 + (long long)bi * hw * C + kk * nc;
  const float* dpt = a.disp_prev_t + (long long)bi * hw;
  const long long rs = (long long)a.w * C;
  for (int t = 0; t < ncp; ++t) {
    float qy, qx;
    dscv_query(px, start_x, start_y, disp, i, j, t, a.r, qy, qx);
    int y0, x0;
    float ay, ax;
    m4d_bilinear_axis(qy, a.h, y0, ay);
    m4d_bilinear_axis(qx, a.w, x0, ax);
    const float* tl = c2b + ((long long)y0 * a.w + x0) * C;
    float acc = 0.0f;
    for (int c = 0; c < nc; ++c) {
      const float wv = m4d_round_half(m4d_lerp2(tl[c], tl[C + c], tl[rs + c], tl[rs + C + c], ax, ay));
      const float pr = m4d_round_half(m4d_round_half(c1p[c]) * wv);
      if (c == 0) acc = pr;
      else acc = (a.cv_accum == 0) ? (acc + pr) : m4d_round_half(acc + pr);
    }
    a.cv[gp * a.cv_stride + kk * ncp + t] = m4d_round_half(acc / (float)nc);
    if (kk == 0) {
      if (a.index_out) {
        a.index_out[(gp * ncp + t) * 2] = y0;
        a.index_out[(gp * ncp + t) * 2 + 1] = x0;
      }
      const bool centre = (t == a.r) && a.log_center != nullptr;
      if (a.prev_disp != nullptr || centre) {
        const float* d0 = dpt + (long long)y0 * a.w + x0;
        const float wd = m4d_lerp2(d0[0], d0[1], d0[a.w], d0[a.w + 1], ax, ay);
        if (a.prev_disp) a.prev_disp[gp * ncp + t] = wd;
        if (centre) a.log_center[gp * a.log_stride] = logf(wd * a.log_scale);
      }
    }
  }
}

template <int LP, int G>
bool launch_wave(const DscvArgs& a, int b, hipStream_t s) {
  constexpr int PPW = 64 / LP;
  const int hw = a.h * a.w;
  int nb = (hw + 4 * PPW - 1) / (4 * PPW);
  if (a.r == 4) m4d_launch((dscv_wave_kernel<LP, G, 9>), dim3(nb, b), dim3(256), 0, s, a);
  else if (a.r == 2) m4d_launch((dscv_wave_kernel<LP, G, 5>), dim3(nb, b), dim3(256), 0, s, a);
  else if (a.r == 6) m4d_launch((dscv_wave_kernel<LP, G, 13>), dim3(nb, b), dim3(256), 0, s, a);
  else return false;
  return true;
}

// Tile geometry / LDS budget of the window variant.  52 KiB per workgroup = three
// workgroups per CU (160 KiB): one stages while the others gather.
constexpr size_t kDscvLdsBudget = 52 * 1024;
unsigned int* g_dscv_fallback_counter = nullptr;     // debug hook (m4d_dscv_set_fallback_counter)
unsigned long long* g_dscv_stamps = nullptr;         // debug hook (m4d_dscv_set_stamps)

template <int LP, int G, int NCP>
void launch_tile_ncp(const DscvTileArgs& ta, int b, hipStream_t s) {
  M4D_LDS_OPT_IN(&dscv_tile_kernel<LP, G, NCP>);
  m4d_launch((dscv_tile_kernel<LP, G, NCP>), dim3(ta.tiles, b), dim3(256), kDscvLdsBudget, s, ta);
}

template <int NC, int K, int NCP, int HPL>
void launch_hyp_ncp(const DscvTileArgs& ta, int b, size_t lds, hipStream_t s) {
  M4D_LDS_OPT_IN(&dscv_hyp_kernel<NC, K, NCP, HPL>);
  m4d_launch((dscv_hyp_kernel<NC, K, NCP, HPL>), dim3(ta.tiles, b), dim3(256), lds, s, ta);
}

// Tile / LDS choice of the hypothesis-per-lane kernel: largest tile whose minimal window (tile +
// sweep + margin) fits 52 KiB (3 workgroups per CU), else 76 KiB (2 per CU).
template <int NC, int K>
bool launch_hyp(const DscvArgs& a, int b, hipStream_t s, bool all_in_lane) {
  constexpr int C = NC * K, CP = C + 4;
  const int ncp = 2 * a.r + 1;
  if (ncp != 9) return false;
  const size_t fixed = 16 * sizeof(float);
  const int cand[3][2] = {{32, 8}, {16, 8}, {16, 4}};
  for (size_t budget : {(size_t)52 * 1024, (size_t)76 * 1024}) {
    for (int ci = 0; ci < 3; ++ci) {
      const int tw = cand[ci][0], th = cand[ci][1];
      if (tw > a.w * 2 || th > a.h * 2) continue;
      // optimistic: tile + about half the sweep in each direction; larger boxes take the
      // (block-uniform) global-gather fallback inside the kernel
      const size_t min_win = (size_t)(tw + ncp) * (th + ncp / 2 + 2) * CP * sizeof(float);
      if (fixed + min_win > budget) continue;
      DscvTileArgs ta;
      ta.d = a; ta.tw = tw; ta.th = th;
      ta.tiles_x = (a.w + tw - 1) / tw;
      ta.tiles = ta.tiles_x * ((a.h + th - 1) / th);
      ta.win_cap_px = (int)((budget - fixed) / (CP * sizeof(float)));
      ta.fallback_counter = g_dscv_fallback_counter;
      ta.stamps = g_dscv_stamps;
      if (all_in_lane) launch_hyp_ncp<NC, K, 9, 9>(ta, b, budget, s);
      else launch_hyp_ncp<NC, K, 9, 3>(ta, b, budget, s);
      return true;
    }
  }
  return false;
}

template <int LP, int G>
bool launch_tile(const DscvArgs& a, int b, hipStream_t s) {
  constexpr int C = 4 * LP, CP = C + 4;
  const int ncp = 2 * a.r + 1;
  if (ncp != 9 && ncp != 5) return false;
  int tw = 32, th = 8;
  if (a.w < 32) tw = 16;
  if (a.h < 8) th = 4;
  const size_t fixed = 16 * sizeof(float);
  while (true) {
    // the window must at least hold the tile plus the sweep of the hypotheses and a margin
    const size_t min_win = (size_t)(tw + ncp + 2) * (th + ncp + 2) * CP * sizeof(float);
    if (fixed + min_win <= kDscvLdsBudget) break;
    if (tw > 16) tw >>= 1; else if (th > 4) th >>= 1; else return false;   // < 64 pixels per tile: caller uses the wave kernel
  }
  DscvTileArgs ta;
  ta.d = a; ta.tw = tw; ta.th = th;
  ta.tiles_x = (a.w + tw - 1) / tw;
  ta.tiles = ta.tiles_x * ((a.h + th - 1) / th);
  ta.win_cap_px = (int)((kDscvLdsBudget - fixed) / (CP * sizeof(float)));
  ta.fallback_counter = g_dscv_fallback_counter;
  ta.stamps = nullptr;
  if (ncp == 9) launch_tile_ncp<LP, G, 9>(ta, b, s);
  else launch_tile_ncp<LP, G, 5>(ta, b, s);
  return true;
}

}  // namespace

extern "C" void m4d_dscv_set_fallback_counter(unsigned int* device_counter) {
  g_dscv_fallback_counter = device_counter;
}

extern "C" void m4d_dscv_set_stamps(unsigned long long* device_buffer) { g_dscv_stamps = device_buffer; }
static int g_dscv_ablate = 0;
extern "C" void m4d_dscv_set_ablation(int mask) { g_dscv_ablate = mask; }
// 0: generic, 1: wave kernel (global gathers; DEFAULT -- fastest measured, profiles/),
// 2: LDS window + channels-across-lanes, 3 / 4: LDS window + 3 / all hypotheses per lane.
static int g_dscv_variant = 1;
extern "C" void m4d_dscv_set_variant(int v) { g_dscv_variant = v; }

extern "C" int m4d_dscv_fwd(const float* c1, const float* c2, const float* disp_prev_t, const float* disp,
                            const float* rot, int rot_c, const float* trans, const float* cam_f,
                            const float* cam_c, int b, int h, int w, int C, int search_range, int nbre_cuts,
                            int cv_accum, float* cv, int cv_stride, float* prev_disp,
                            float* log_center, int log_stride, float log_scale,
                            int32_t* index_out, void* stream) {
  M4D_CHECK_ARG(c1 && c2 && disp_prev_t && disp && rot && trans && cam_f && cam_c && cv);
  M4D_CHECK_ARG(b > 0 && h >= 2 && w >= 2 && C > 0 && search_range >= 0 && nbre_cuts > 0);
  M4D_CHECK_ARG(rot_c == 3 || rot_c == 4);
  M4D_CHECK_ARG(C % nbre_cuts == 0);
  M4D_CHECK_ARG(cv_accum == 0 || cv_accum == 1);
  M4D_CHECK_ARG(cv_stride >= nbre_cuts * (2 * search_range + 1));
  M4D_CHECK_ARG(log_center == nullptr || log_stride >= 1);
  DscvArgs a;
  a.c1 = c1; a.c2 = c2; a.disp_prev_t = disp_prev_t; a.disp = disp;
  a.rot = rot; a.rot_c = rot_c; a.trans = trans; a.cam_f = cam_f; a.cam_c = cam_c;
  a.h = h; a.w = w; a.C = C; a.r = search_range; a.k = nbre_cuts; a.nc = C / nbre_cuts; a.cv_accum = cv_accum;
  a.cv = cv; a.cv_stride = cv_stride; a.prev_disp = prev_disp; a.log_center = log_center;
  a.log_stride = log_stride; a.log_scale = log_scale; a.index_out = index_out; a.ablate = g_dscv_ablate;
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = (((uintptr_t)c1 | (uintptr_t)c2) & 15u) == 0 && (a.nc % 4 == 0);
  const int lp = C / 4, g = a.nc / 4;
  const bool fits = aligned && (2 * search_range + 1) <= 16;
  // (LP, G) pairs of the 6-level pyramid (C = 16..192, cuts 1,2,2,4,4,8) plus the small
  // shapes the unit tests use; everything else takes the generic kernel.
  if (fits && (g_dscv_variant == 3 || g_dscv_variant == 4)) {     // hypotheses in-lane on the LDS window
    const bool ail = g_dscv_variant == 4;
    if (a.nc == 16 && nbre_cuts == 1 && launch_hyp<16, 1>(a, b, s, ail)) return M4D_LAUNCH_RESULT();
    if (a.nc == 16 && nbre_cuts == 2 && launch_hyp<16, 2>(a, b, s, ail)) return M4D_LAUNCH_RESULT();
    if (a.nc == 32 && nbre_cuts == 2 && launch_hyp<32, 2>(a, b, s, ail)) return M4D_LAUNCH_RESULT();
  }
  const bool tile = fits && g_dscv_variant == 2;
  if (g_dscv_variant == 0) {
    const long long threads = (long long)h * w * nbre_cuts;
    m4d_launch(dscv_generic_kernel, dim3(m4d_blocks(threads, 256), b), dim3(256), 0, s, a);
    return M4D_LAUNCH_RESULT();
  }
  if (tile && lp == 4 && g == 4 && launch_tile<4, 4>(a, b, s)) return M4D_LAUNCH_RESULT();
  if (tile && lp == 8 && g == 4 && launch_tile<8, 4>(a, b, s)) return M4D_LAUNCH_RESULT();
  if (tile && lp == 16 && g == 8 && launch_tile<16, 8>(a, b, s)) return M4D_LAUNCH_RESULT();
  if (tile && lp == 24 && g == 6 && launch_tile<24, 6>(a, b, s)) return M4D_LAUNCH_RESULT();
  if (tile && lp == 32 && g == 8 && launch_tile<32, 8>(a, b, s)) return M4D_LAUNCH_RESULT();
  if (tile && lp == 48 && g == 6 && launch_tile<48, 6>(a, b, s)) return M4D_LAUNCH_RESULT();
  bool done = false;
  if (fits && lp == 4 && g == 4) done = launch_wave<4, 4>(a, b, s);          // C=16  k=1
  else if (fits && lp == 8 && g == 4) done = launch_wave<8, 4>(a, b, s);     // C=32  k=2
  else if (fits && lp == 16 && g == 8) done = launch_wave<16, 8>(a, b, s);   // C=64  k=2
  else if (fits && lp == 24 && g == 6) done = launch_wave<24, 6>(a, b, s);   // C=96  k=4
  else if (fits && lp == 32 && g == 8) done = launch_wave<32, 8>(a, b, s);   // C=128 k=4
  else if (fits && lp == 48 && g == 6) done = launch_wave<48, 6>(a, b, s);   // C=192 k=8
  else if (fits && lp == 8 && g == 8) done = launch_wave<8, 8>(a, b, s);     // C=32  k=1
  else if (fits && lp == 4 && g == 2) done = launch_wave<4, 2>(a, b, s);     // C=16  k=2
  if (!done) {                                  // any other shape / search range
    const long long threads = (long long)h * w * nbre_cuts;
    m4d_launch(dscv_generic_kernel, dim3(m4d_blocks(threads, 256), b), dim3(256), 0, s, a);
  }
  return M4D_LAUNCH_RESULT();
}

template <int LP, int G>
bool launch_wave_sncv(const DscvArgs& a, int b, const m4d_sncv::SncvArgs& sa, hipStream_t s) {
  constexpr int PPW = 64 / LP;
  const int hw = a.h * a.w;
  const int nb = (hw + 4 * PPW - 1) / (4 * PPW);
  const int nb_sncv = (int)m4d_sncv::sncv_small_blocks(sa, b);
  const dim3 grid((unsigned)(nb * b + nb_sncv));
  const int total_px = b * hw;
  if (a.r == 4) m4d_launch((dscv_sncv_small_kernel<LP, G, 9>), grid, dim3(256), 0, s, a, nb, b, sa, total_px, nb_sncv);
  else if (a.r == 2) m4d_launch((dscv_sncv_small_kernel<LP, G, 5>), grid, dim3(256), 0, s, a, nb, b, sa, total_px, nb_sncv);
  else return false;
  return true;
}

template <int LP, int G>
bool launch_front_small(const DscvArgs& a, int b, const m4d_sncv::SncvArgs& sa, const m4d_level::LevelPreArgs& pa, hipStream_t s) {
  constexpr int PPW = 64 / LP;
  const int hw = a.h * a.w;
  const int nb = (hw + 4 * PPW - 1) / (4 * PPW);
  const int nb_sncv = (int)m4d_sncv::sncv_small_blocks(sa, b);
  int pre_gx = m4d_blocks((long long)hw, 256);
  if (pre_gx > 4096) pre_gx = 4096;
  const dim3 grid((unsigned)(nb * b + nb_sncv + pre_gx * b));
  const int total_px = b * hw;
  if (a.r == 4) m4d_launch((level_front_small_kernel<LP, G, 9>), grid, dim3(256), 0, s, a, nb, b, sa, total_px, nb_sncv, pa, pre_gx);
  else if (a.r == 2) m4d_launch((level_front_small_kernel<LP, G, 5>), grid, dim3(256), 0, s, a, nb, b, sa, total_px, nb_sncv, pa, pre_gx);
  else return false;
  return true;
}

static bool front_small_shape(int C, int nbre_cuts, int dscv_range, int sncv_range) {
  if (C <= 0 || nbre_cuts <= 0 || C % nbre_cuts != 0 || (C / nbre_cuts) % 4 != 0) return false;
  if (!(dscv_range == 4 || dscv_range == 2) || sncv_range < 0) return false;
  const int lp = C / 4, g = C / nbre_cuts / 4;
  return (lp == 24 && g == 6) || (lp == 32 && g == 8) || (lp == 48 && g == 6) || (lp == 16 && g == 8) || (lp == 8 && g == 4) ||
         (lp == 4 && g == 4);
}

extern "C" int m4d_level_front_small_supported(int C, int nbre_cuts, int dscv_range, int sncv_range) {
  return (front_small_shape(C, nbre_cuts, dscv_range, sncv_range) && g_dscv_variant == 1) ? 1 : 0;
}

extern "C" int m4d_level_front_small(const float* norm_f, const float* prev_f, const float* depth_prev_t,
                                     const float* prev_l_parallax, const float* prev_l_other, int ph, int pw,
                                     const float* rot, int rot_c, const float* trans, const float* cam_f, const float* cam_c,
                                     int b, int h, int w, int C, int nbre_cuts, int dscv_range, int sncv_range, int cv_accum,
                                     float* f_input, int f_stride, float log_scale, void* stream) {
  M4D_CHECK_ARG(norm_f && prev_f && depth_prev_t && rot && trans && cam_f && cam_c && f_input);
  M4D_CHECK_ARG(b > 0 && h >= 2 && w >= 2 && (rot_c == 3 || rot_c == 4) && (cv_accum == 0 || cv_accum == 1));
  M4D_CHECK_ARG(front_small_shape(C, nbre_cuts, dscv_range, sncv_range) && g_dscv_variant == 1);
  M4D_CHECK_ARG((prev_l_parallax == nullptr) == (prev_l_other == nullptr));
  if (prev_l_parallax) M4D_CHECK_ARG(ph > 0 && pw > 0);
  M4D_CHECK_ARG(((((uintptr_t)norm_f | (uintptr_t)prev_f)) & 15u) == 0);
  const int k = nbre_cuts, ncp = 2 * dscv_range + 1, mo = 2 * sncv_range + 1;
  const int f_in = ncp * k + 1 + 4 + mo * mo * k + 1;            // cv | log para_l | other(4) | sncv | log para_t
  M4D_CHECK_ARG(f_stride >= f_in);
  const int log_off = ncp * k, other_off = log_off + 1, sncv_off = log_off + 5;
  DscvArgs a;
  a.c1 = norm_f; a.c2 = prev_f; a.disp_prev_t = nullptr; a.disp = nullptr;
  a.rot = rot; a.rot_c = rot_c; a.trans = trans; a.cam_f = cam_f; a.cam_c = cam_c;
  a.h = h; a.w = w; a.C = C; a.r = dscv_range; a.k = k; a.nc = C / k; a.cv_accum = cv_accum;
  a.cv = f_input; a.cv_stride = f_stride; a.prev_disp = nullptr; a.log_center = f_input + (f_in - 1);
  a.log_stride = f_stride; a.log_scale = log_scale; a.index_out = nullptr; a.ablate = g_dscv_ablate;
  a.pl_para = prev_l_parallax; a.ph = ph; a.pw = pw; a.depth_prev_t = depth_prev_t;
  m4d_sncv::SncvArgs sa;
  sa.c1 = norm_f; sa.c2 = norm_f; sa.h = h; sa.w = w; sa.C = C; sa.r = sncv_range; sa.d = 1; sa.k = k; sa.nc = C / k;
  sa.out = f_input + sncv_off; sa.out_stride = f_stride; sa.th = sa.tw = sa.tiles_x = 0;
  m4d_level::LevelPreArgs pa;
  pa.pl_depth = nullptr; pa.pl_para = prev_l_parallax; pa.pl_other = prev_l_other; pa.ph = ph; pa.pw = pw;
  pa.depth_prev_t = nullptr; pa.trans = trans; pa.cam_f = cam_f; pa.cam_c = cam_c; pa.h = h; pa.w = w;
  pa.para_prev_l = nullptr; pa.depth_prev_l = nullptr; pa.other_prev_l = nullptr; pa.para_prev_t = nullptr;
  pa.f_input = f_input; pa.f_stride = f_stride; pa.log_off = log_off; pa.other_off = other_off; pa.log_scale = log_scale;
  pa.depth_state_reset = nullptr;
  hipStream_t s = (hipStream_t)stream;
  const int lp = C / 4, g = C / k / 4;
  bool done = false;
  if (lp == 24 && g == 6) done = launch_front_small<24, 6>(a, b, sa, pa, s);        // C=96  k=4
  else if (lp == 32 && g == 8) done = launch_front_small<32, 8>(a, b, sa, pa, s);   // C=128 k=4
  else if (lp == 48 && g == 6) done = launch_front_small<48, 6>(a, b, sa, pa, s);   // C=192 k=8
  else if (lp == 16 && g == 8) done = launch_front_small<16, 8>(a, b, sa, pa, s);   // C=64  k=2
  else if (lp == 8 && g == 4) done = launch_front_small<8, 4>(a, b, sa, pa, s);     // C=32  k=2
  else if (lp == 4 && g == 4) done = launch_front_small<4, 4>(a, b, sa, pa, s);     // C=16  k=1
  if (!done) return (int)hipErrorInvalidValue;
  return M4D_LAUNCH_RESULT();
}

// m4d_dscv_fwd followed by m4d_sncv_fwd(c1, c1, ...) of the same level; on small maps (<= 6000 pixels, the pyramid's
// channel / cut pairs) both run in ONE launch, otherwise one after the other.  Same results bit for bit.
extern "C" int m4d_dscv_sncv_fwd(const float* c1, const float* c2, const float* disp_prev_t, const float* disp,
                                 const float* rot, int rot_c, const float* trans, const float* cam_f,
                                 const float* cam_c, int b, int h, int w, int C, int search_range, int nbre_cuts,
                                 int cv_accum, float* cv, int cv_stride, float* prev_disp,
                                 float* log_center, int log_stride, float log_scale,
                                 int sncv_search_range, float* sncv_out, int sncv_out_stride, void* stream) {
  M4D_CHECK_ARG(c1 && c2 && disp_prev_t && disp && rot && trans && cam_f && cam_c && cv && sncv_out);
  M4D_CHECK_ARG(b > 0 && h >= 2 && w >= 2 && C > 0 && search_range >= 0 && nbre_cuts > 0 && C % nbre_cuts == 0);
  const int nc = C / nbre_cuts, mo = 2 * sncv_search_range + 1;
  const bool aligned = (((uintptr_t)c1 | (uintptr_t)c2) & 15u) == 0 && (nc % 4 == 0);
  const int lp = C / 4, g = nc / 4;
  static int small_px = -1;                            // the same threshold as m4d_sncv_fwd's small-map kernel
  if (small_px < 0) { const char* e = getenv("M4D_SNCV_SMALL_PX"); small_px = e ? atoi(e) : 6000; }
  const bool merged_ok = aligned && g_dscv_variant == 1 && (rot_c == 3 || rot_c == 4) && (cv_accum == 0 || cv_accum == 1) &&
                         cv_stride >= nbre_cuts * (2 * search_range + 1) && sncv_search_range >= 0 &&
                         sncv_out_stride >= mo * mo * nbre_cuts && (long long)b * h * w <= small_px &&
                         (log_center == nullptr || log_stride >= 1);
  if (merged_ok) {
    DscvArgs a;
    a.c1 = c1; a.c2 = c2; a.disp_prev_t = disp_prev_t; a.disp = disp;
    a.rot = rot; a.rot_c = rot_c; a.trans = trans; a.cam_f = cam_f; a.cam_c = cam_c;
    a.h = h; a.w = w; a.C = C; a.r = search_range; a.k = nbre_cuts; a.nc = nc; a.cv_accum = cv_accum;
    a.cv = cv; a.cv_stride = cv_stride; a.prev_disp = prev_disp; a.log_center = log_center;
    a.log_stride = log_stride; a.log_scale = log_scale; a.index_out = nullptr; a.ablate = g_dscv_ablate;
    m4d_sncv::SncvArgs sa;
    sa.c1 = c1; sa.c2 = c1; sa.h = h; sa.w = w; sa.C = C; sa.r = sncv_search_range; sa.d = 1; sa.k = nbre_cuts; sa.nc = nc;
    sa.out = sncv_out; sa.out_stride = sncv_out_stride; sa.th = sa.tw = sa.tiles_x = 0;
    hipStream_t s = (hipStream_t)stream;
    bool done = false;
    if (lp == 24 && g == 6) done = launch_wave_sncv<24, 6>(a, b, sa, s);        // C=96  k=4
    else if (lp == 32 && g == 8) done = launch_wave_sncv<32, 8>(a, b, sa, s);   // C=128 k=4
    else if (lp == 48 && g == 6) done = launch_wave_sncv<48, 6>(a, b, sa, s);   // C=192 k=8
    else if (lp == 16 && g == 8) done = launch_wave_sncv<16, 8>(a, b, sa, s);   // C=64  k=2
    else if (lp == 8 && g == 4) done = launch_wave_sncv<8, 4>(a, b, sa, s);     // C=32  k=2
    else if (lp == 4 && g == 4) done = launch_wave_sncv<4, 4>(a, b, sa, s);     // C=16  k=1
    if (done) return M4D_LAUNCH_RESULT();
  }
  const int rc = m4d_dscv_fwd(c1, c2, disp_prev_t, disp, rot, rot_c, trans, cam_f, cam_c, b, h, w, C, search_range, nbre_cuts,
                              cv_accum, cv, cv_stride, prev_disp, log_center, log_stride, log_scale, nullptr, stream);
  if (rc != 0) return rc;
  return m4d_sncv_fwd(c1, c1, b, h, w, C, sncv_search_range, 1, nbre_cuts, sncv_out, sncv_out_stride, stream);
}
