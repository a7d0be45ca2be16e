// Training-side convolution kernels that complete SURVEY section 8 row f-1 without MIOpen (train_step,
// m4depth_network.py:371-399: tf.GradientTape differentiates every Conv2D of the encoder and the refiners):
//
//   * m4d_conv3x3_wgrad: the weight gradient of a 3x3 TF-'SAME' convolution (stride 1 or 2),
//         dW[o][ky][kx][i] = sum over (b, oy, ox) of g[b,oy,ox,o] * xpad[b, oy*s + ky - pt, ox*s + kx - pl, i]
//     as 9 GEMMs (one per tap) with M = Cout, N = Cin and K = the output pixels, on v_mfma_f32_32x32x2_f32 (exact float32
//     fused multiply-adds, deterministic).  A workgroup owns a 32 x 32 (Cout, Cin) block for all 9 taps (9 accumulator
//     tiles per wave) and a slice of the pixels: it walks 8 x 16-pixel tiles, staging the gradient tile and the input's
//     halo tile once in LDS for the 9 taps; its four waves split the tile's rows.  The pixel slices (split-K) and the four
//     waves leave their partial sums in a workspace; a second kernel adds them in a fixed order and writes the gradient in
//     the parameter's own memory layout (OIHW tensor, channels-last strides = [O][ky][kx][I]).
//   * m4d_conv3x3_wgrad_cin3: the same gradient for the 3-channel image layer (K = 27 is too thin for the matrix cores):
//     per-workgroup partial sums on the vector ALUs, same two-stage fixed-order reduction.
//   * m4d_dilate2: the gradient of a stride-2 layer's output, spread onto the input grid (zeros in between), so that the
//     data gradient of the stride-2 layer is the stride-1 MFMA convolution with the rotated / transposed weights that
//     training.py already uses for the stride-1 layers.
#include <type_traits>
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTH = 8, kTW = 16;                  // output-pixel tile of one staging round
constexpr int kGS = 32;                           // LDS pixel stride of the gradient tile (floats)
constexpr int kXS = 36;                           // LDS pixel stride of the input tile: 16-byte aligned rows, and the two k-lanes
                                                  // of a fragment read (pixels ox, ox + 1: 1 or 2 input pixels apart) overlap on 4-8 banks only

struct WgradArgs {
  const float* x; const float* g; int b, h, w, cin, oh, ow, cout, stride, pt, pl;
  float* partial;          // [slices][9][mblk * 32][nblk * 32]
  int mblk, nblk, slices, tiles_x, tiles_y;
};

// VG / VX: floats per staging load of the gradient / input tile (4 when the channel count is a multiple of 4, 2 for the even
// refiner-input widths 122 / 238 / 470, else 1): the vector ALU shares its issue slots with the fp32 matrix cores, so staging
// instructions are not free.  With 16-byte loads on both sides and stride 1 the next tile's loads are issued before the MFMA
// loop of the current one and committed to LDS after it.
template <int V> struct WgVec { typedef float type; };
template <> struct WgVec<2> { typedef float2 type; };
template <> struct WgVec<4> { typedef float4 type; };
template <int V>
__device__ __forceinline__ void wg_copy(const float* src, float* dst, bool ok) {
  if constexpr (V == 4) {
    const float4 v = *reinterpret_cast<const float4*>(src);
    *reinterpret_cast<float4*>(dst) = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  } else if constexpr (V == 2) {
    const float2 v = *reinterpret_cast<const float2*>(src);
    *reinterpret_cast<float2*>(dst) = make_float2(ok ? v.x : 0.f, ok ? v.y : 0.f);
  } else {
    *dst = ok ? *src : 0.f;
  }
}

template <int STRIDE, int VG, int VX>
__global__ void __launch_bounds__(256, 2)
conv3x3_wgrad_kernel(const WgradArgs a) {
  constexpr int XH = (kTH - 1) * STRIDE + 3, XW = (kTW - 1) * STRIDE + 3;
  constexpr bool VEC = VG == 4 && VX == 4;
  constexpr int GE = 32 / VG, XE = 32 / VX;                          // staging elements per pixel
  constexpr int GV = kTH * kTW * GE;                                 // staging elements of the gradient tile
  constexpr int XV = XH * XW * XE;
  constexpr bool PRE = VEC && STRIDE == 1;        // next tile's loads held in registers across the MFMA loop (10 float4 per lane)
  constexpr int GU = PRE ? (GV + 255) / 256 : 1, XU = PRE ? (XV + 255) / 256 : 1;
  extern __shared__ __align__(16) float smem[];
  float* gs = smem;                               // [kTH][kTW][32 couts]
  float* xs = smem + kTH * kTW * kGS;             // [XH][XW][kXS]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int mb = blockIdx.x % a.mblk, nb = blockIdx.x / a.mblk;
  const int slice = blockIdx.y;
  const int o0 = mb * 32, i0 = nb * 32;
  const long long tiles_img = (long long)a.tiles_x * a.tiles_y;
  const long long total_tiles = tiles_img * a.b;
  f32x16 acc[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;
  const int ml = lane & 31, kh = lane >> 5;

  typedef typename std::conditional<VEC, float4, float>::type vec_t;
  vec_t gpre[GU], xpre[XU];
  unsigned g_ok = 0;
  unsigned long long x_ok = 0;
  auto stage = [&](long long tile) {               // the other shapes: load and store element by element
    const int bi = (int)(tile / tiles_img);
    const int tl = (int)(tile - (long long)bi * tiles_img);
    const int oy0 = (tl / a.tiles_x) * kTH, ox0 = (tl % a.tiles_x) * kTW;
    const float* gimg = a.g + (long long)bi * a.oh * a.ow * a.cout;
    const float* ximg = a.x + (long long)bi * a.h * a.w * a.cin;
    const int y_base = oy0 * STRIDE - a.pt, x_base = ox0 * STRIDE - a.pl;
    for (int e = t; e < GV; e += 256) {
      const int c = (e % GE) * VG, p = e / GE;
      const int oy = oy0 + p / kTW, ox = ox0 + p % kTW;
      const bool ok = oy < a.oh && ox < a.ow && o0 + c < a.cout;
      wg_copy<VG>(gimg + ((long long)min(oy, a.oh - 1) * a.ow + min(ox, a.ow - 1)) * a.cout + min(o0 + c, a.cout - VG),
                  gs + p * kGS + c, ok);
    }
    for (int e = t; e < XV; e += 256) {
      const int c = (e % XE) * VX, p = e / XE;
      const int yy = y_base + p / XW, xx = x_base + p % XW;
      const bool ok = yy >= 0 && yy < a.h && xx >= 0 && xx < a.w && i0 + c < a.cin;
      wg_copy<VX>(ximg + ((long long)min(max(yy, 0), a.h - 1) * a.w + min(max(xx, 0), a.w - 1)) * a.cin + min(i0 + c, a.cin - VX),
                  xs + p * kXS + c, ok);
    }
  };
  auto issue = [&](long long tile) {              // unconditional clamped loads; validity applied at the commit
    const int bi = (int)(tile / tiles_img);
    const int tl = (int)(tile - (long long)bi * tiles_img);
    const int oy0 = (tl / a.tiles_x) * kTH, ox0 = (tl % a.tiles_x) * kTW;
    const float* gimg = a.g + (long long)bi * a.oh * a.ow * a.cout;
    const float* ximg = a.x + (long long)bi * a.h * a.w * a.cin;
    const int y_base = oy0 * STRIDE - a.pt, x_base = ox0 * STRIDE - a.pl;
    g_ok = 0; x_ok = 0;
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int e = min(u * 256 + t, GV - 1);
      const int c = VEC ? (e & 7) * 4 : (e & 31), p = VEC ? (e >> 3) : (e >> 5);
      const int oy = oy0 + p / kTW, ox = ox0 + p % kTW;
      const bool ok = oy < a.oh && ox < a.ow && o0 + c < a.cout;
      g_ok |= ok ? (1u << u) : 0u;
      const float* src = gimg + ((long long)min(oy, a.oh - 1) * a.ow + min(ox, a.ow - 1)) * a.cout + min(o0 + c, a.cout - (VEC ? 4 : 1));
      gpre[u] = *reinterpret_cast<const vec_t*>(src);
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int e = min(u * 256 + t, XV - 1);
      const int c = VEC ? (e & 7) * 4 : (e & 31), p = VEC ? (e >> 3) : (e >> 5);
      const int yy = y_base + p / XW, xx = x_base + p % XW;
      const bool ok = yy >= 0 && yy < a.h && xx >= 0 && xx < a.w && i0 + c < a.cin;
      x_ok |= ok ? (1ull << u) : 0ull;
      const float* src = ximg + ((long long)min(max(yy, 0), a.h - 1) * a.w + min(max(xx, 0), a.w - 1)) * a.cin + min(i0 + c, a.cin - (VEC ? 4 : 1));
      xpre[u] = *reinterpret_cast<const vec_t*>(src);
    }
  };
  auto commit = [&]() {                            // zero outside the map / beyond the channel count ('SAME' zero padding)
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int e = u * 256 + t;
      if (e < GV) {
        const bool ok = (g_ok >> u) & 1u;
        if constexpr (VEC) {
          const float4 v = gpre[u];
          *reinterpret_cast<float4*>(gs + (e >> 3) * kGS + (e & 7) * 4) = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        } else {
          gs[(e >> 5) * kGS + (e & 31)] = ok ? gpre[u] : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int e = u * 256 + t;
      if (e < XV) {
        const bool ok = (x_ok >> u) & 1ull;
        if constexpr (VEC) {
          const float4 v = xpre[u];
          *reinterpret_cast<float4*>(xs + (e >> 3) * kXS + (e & 7) * 4) = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        } else {
          xs[(e >> 5) * kXS + (e & 31)] = ok ? xpre[u] : 0.f;
        }
      }
    }
  };

  long long tile = slice;
  if (tile < total_tiles) {
    if constexpr (PRE) { issue(tile); commit(); } else { stage(tile); }
  }
  __syncthreads();
  for (; tile < total_tiles; tile += a.slices) {
    const long long nxt = tile + a.slices;
    const bool has_next = nxt < total_tiles;
    if constexpr (PRE) { if (has_next) issue(nxt); }      // in flight during the MFMA loop below
    // ---- wave w: tile rows 2w, 2w+1; one k-step = output pixels (ox, ox + 1) of a row
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * wave + rr;
#pragma unroll 2
      for (int ks = 0; ks < kTW / 2; ++ks) {
        const int px = 2 * ks + kh;                                       // this lane's k index = pixel of the pair
        const float av = gs[(row * kTW + px) * kGS + ml];                 // A[m = cout][k]
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float bv = xs[((row * STRIDE + ky) * XW + px * STRIDE + kx) * kXS + ml];   // B[k][n = cin]
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ky * 3 + kx], 0, 0, 0);
          }
      }
    }
    __syncthreads();
    if (has_next) {
      if constexpr (PRE) commit(); else stage(nxt);
    }
    __syncthreads();
  }
  // ---- the four waves' sums meet in LDS, tap by tap, and are added in wave order (fixed order: deterministic); the
  //      workgroup leaves ONE partial [9][M][N] per pixel slice (M = mblk * 32, N = nblk * 32)
  const int M = a.mblk * 32, N = a.nblk * 32;
  float* out = a.partial + ((long long)slice * 9) * M * N;
  float* red = smem;                                                      // [4 waves][32][32]: 16 KB, the staging area is free now
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mr = (r & 3) + 8 * (r >> 2) + 4 * kh;                     // C/D map of the 32x32 MFMA: col = lane & 31
      red[(wave * 32 + mr) * 32 + ml] = acc[tp][r];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = u * 256 + t;                                          // (row, col) of the 32 x 32 block
      const float sum = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
      out[((long long)tp * M + o0 + (e >> 5)) * N + i0 + (e & 31)] = sum;
    }
    __syncthreads();
  }
}

// dW[o][tap][i] = the partials added in a fixed order (deterministic): 4 lane groups sum a quarter of the partials each
// (loads unrolled 8 deep: the sum is sequential, the loads are not), then the four quarters are added in order through LDS.
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ partial, int n_part, int M, int N, int cout, int cin, float* __restrict__ dw) {
  __shared__ float q4[4][64];
  const long long total = (long long)cout * 9 * cin;
  const int lo = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long long stride = 9LL * M * N;
  const int per = (n_part + 3) / 4;
  const int k0 = grp * per, k1 = min(k0 + per, n_part);
  for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
    const long long idx = base + lo;
    float s = 0.f;
    if (idx < total) {
      const int i = (int)(idx % cin);
      const int tp = (int)((idx / cin) % 9);
      const int o = (int)(idx / ((long long)cin * 9));
      const float* p = partial + ((long long)tp * M + o) * N + i;
#pragma unroll 8
      for (int k = k0; k < k1; ++k) s = (k == k0) ? p[k * stride] : s + p[k * stride];
    }
    q4[grp][lo] = s;
    __syncthreads();
    if (grp == 0 && idx < total) {
      float r = q4[0][lo];
      for (int gq = 1; gq < 4; ++gq) if (gq * per < n_part) r = r + q4[gq][lo];
      dw[idx] = r;
    }
    __syncthreads();
  }
}

// ---- thin layers (Cin = 3: the image layer; Cin = 16: encoder level 0 / 1), Cout <= 32 ------------------------------------
// The taps are folded into N: M = Cout (<= 32: one MFMA M-tile), N = (tap, cin) = 9 * Cin columns = NT N-tiles (27 -> 1 tile,
// 144 -> 5 tiles instead of the 9 quarter-full tiles of the kernel above), K = output pixels.  No LDS staging: a wave walks
// items of 8 output pixels of a row (4 k-steps); lane (m, kh) loads g[pixel + kh][m] and, as column n of every N-tile,
// x[(pixel + kh) * stride + tap(n)][cin(n)] straight from global memory (each tap of a pixel is one short contiguous run;
// L1 / L2 resident), the next item's loads in flight during the MFMAs.
template <int CIN, int STRIDE>
__global__ void __launch_bounds__(256)
wgrad_thin_kernel(const float* __restrict__ x, const float* __restrict__ g, int b, int h, int w, int oh, int ow, int cout,
                  int pt, int pl, int n_waves, float* __restrict__ partial) {
  constexpr int NCOL = 9 * CIN, NT = (NCOL + 31) / 32;
  __shared__ float red[4 * 32 * 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_id = blockIdx.x * 4 + wave;
  const int ml = lane & 31, kh = lane >> 5;
  const bool m_ok = ml < cout;
  int dy[NT], dxc[NT];                                                    // per N-tile: this lane's tap row offset, and (tap col offset, cin) folded
  bool n_ok[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = j * 32 + ml;
    const int tap = min(n, NCOL - 1) / CIN, ci = min(n, NCOL - 1) % CIN;
    n_ok[j] = n < NCOL;
    dy[j] = tap / 3 - pt;
    dxc[j] = (tap % 3 - pl) * CIN + ci;                                   // float offset inside an input row, relative to pixel ox * stride
  }
  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int oct_per_row = (ow + 7) / 8;
  const long long items = (long long)b * oh * oct_per_row;
  float av0[4], bv0[NT][4], av1[4], bv1[NT][4];               // two operand sets, statically indexed (ping-pong)
  auto load = [&](long long item, float* a4, float (*b4)[4]) {
    const int row = (int)(item / oct_per_row);                            // bi * oh + oy
    const int p0 = (int)(item - (long long)row * oct_per_row) * 8;
    const int oy = row % oh;
    const int bi = row / oh;
    const float* grow = g + (long long)row * ow * cout + (m_ok ? ml : 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int px = p0 + 2 * u + kh;
      const float gv = grow[(long long)min(px, ow - 1) * cout];
      a4[u] = (px < ow && m_ok) ? gv : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int sy = oy * STRIDE + dy[j];
      const bool y_ok = n_ok[j] && sy >= 0 && sy < h;
      const float* xrow = x + ((long long)bi * h + min(max(sy, 0), h - 1)) * w * CIN;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int px = p0 + 2 * u + kh;
        const int off = px * STRIDE * CIN + dxc[j];                       // float offset in the row: (sx * CIN + ci)
        const bool ok = px < ow && y_ok && off >= 0 && off < w * CIN;
        const float xv = xrow[min(max(off, 0), w * CIN - 1)];
        b4[j][u] = ok ? xv : 0.f;
      }
    }
  };
  auto mma = [&](const float* a4, const float (*b4)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u], b4[j][u], acc[j], 0, 0, 0);
  };
  long long item = wave_id;
  if (item < items) load(item, av0, bv0);
  while (item < items) {                                                  // next item's loads in flight during the MFMAs
    long long nxt = item + n_waves;
    if (nxt < items) load(nxt, av1, bv1);
    mma(av0, bv0);
    item = nxt;
    if (item >= items) break;
    nxt = item + n_waves;
    if (nxt < items) load(nxt, av0, bv0);
    mma(av1, bv1);
    item = nxt;
  }
  // the four waves of the workgroup add up in wave order through LDS, N-tile by N-tile; one partial [cout][9 * CIN] per workgroup
  float* out = partial + (long long)blockIdx.x * NCOL * cout;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + ml] = acc[j][r];   // C/D map: row = Cout index
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += 256) {
      const int o = e >> 5, n = j * 32 + (e & 31);
      if (o < cout && n < NCOL) out[o * NCOL + n] = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];   // [o][ky][kx][i]
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256)
wgrad_cin3_reduce_kernel(const float* __restrict__ partial, int n_part, int n_out, float* __restrict__ dw) {
  __shared__ float q4[4][64];
  const int lo = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lo;
  const int per = (n_part + 3) / 4;
  const int k0 = grp * per, k1 = min(k0 + per, n_part);
  float s = 0.f;
  if (idx < n_out) {
#pragma unroll 8
    for (int k = k0; k < k1; ++k) s = (k == k0) ? partial[(long long)k * n_out + idx] : s + partial[(long long)k * n_out + idx];
  }
  q4[grp][lo] = s;
  __syncthreads();
  if (grp == 0 && idx < n_out) {
    float r = q4[0][lo];
    for (int gq = 1; gq < 4; ++gq) if (gq * per < n_part) r = r + q4[gq][lo];
    dw[idx] = r;
  }
}

// ---- g [b,oh,ow,C] -> out [b,h,w,C]: out[y][x] = g[(y - dy) / 2][(x - dx) / 2] where both are even, 0 elsewhere ---------
__global__ void __launch_bounds__(256)
dilate2_kernel(const float* __restrict__ g, int oh, int ow, int C, int h, int w, int dy, int dx, long long total4,
               float* __restrict__ out) {
  const int c4n = C >> 2;
  for (long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i4 % c4n);
    long long p = i4 / c4n;
    const int xx = (int)(p % w); p /= w;
    const int yy = (int)(p % h);
    const long long bi = p / h;
    const int sy = yy - dy, sx = xx - dx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sy >= 0 && sx >= 0 && (sy & 1) == 0 && (sx & 1) == 0 && (sy >> 1) < oh && (sx >> 1) < ow)
      v = *reinterpret_cast<const float4*>(g + ((bi * oh + (sy >> 1)) * ow + (sx >> 1)) * C + c4 * 4);
    *reinterpret_cast<float4*>(out + i4 * 4) = v;
  }
}

// pixel slices (split-K) of a launch with ``blocks`` 32 x 32 (Cout, Cin) blocks: about 768 workgroups in total
inline long long wgrad_max_slices(int blocks) {
  long long sl = (768 + blocks - 1) / blocks;
  if (sl > 256) sl = 256;                     // narrow layers: the partial sums' traffic (slices x 9 x M x N floats) outweighs a fuller grid
  return sl < 1 ? 1 : sl;
}

inline int same_pad_before(int in, int stride) {
  const int out = (in + stride - 1) / stride;
  const int total = (out - 1) * stride + 3 - in;
  return (total > 0 ? total : 0) / 2;
}

}  // namespace

extern "C" long long m4d_conv3x3_wgrad_workspace_floats(int b, int h, int w, int cin, int cout, int stride) {
  if (b <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || (stride != 1 && stride != 2)) return 0;
  if (cin < 8 || (cin == 16 && cout <= 32)) return 512LL * 9 * cin * cout;
  const int mblk = (cout + 31) / 32, nblk = (cin + 31) / 32;
  return wgrad_max_slices(mblk * nblk) * 9 * mblk * 32 * nblk * 32;
}

extern "C" int m4d_conv3x3_wgrad(const float* x, const float* g, int b, int h, int w, int cin, int cout, int stride,
                                 float* workspace, long long workspace_floats, float* dw, void* stream) {
  M4D_CHECK_ARG(x && g && workspace && dw && b > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && (stride == 1 || stride == 2));
  M4D_CHECK_ARG(workspace_floats >= m4d_conv3x3_wgrad_workspace_floats(b, h, w, cin, cout, stride));
  hipStream_t s = (hipStream_t)stream;
  const int oh = (h + stride - 1) / stride, ow = (w + stride - 1) / stride;
  if (cin < 8 || (cin == 16 && cout <= 32)) {
    M4D_CHECK_ARG((cin == 3 && stride == 1 && cout <= 32) || cin == 16);
    const long long items = (long long)b * oh * ((ow + 7) / 8);
    int n_wg = (int)(items < 2048 ? (items + 3) / 4 : 512);                // <= 512 workgroups = 2048 waves
    if (n_wg < 1) n_wg = 1;
    const int pt = same_pad_before(h, stride), pl = same_pad_before(w, stride);
    if (cin == 3) m4d_launch((wgrad_thin_kernel<3, 1>), dim3(n_wg), dim3(256), 0, s, x, g, b, h, w, oh, ow, cout, pt, pl, n_wg * 4, workspace);
    else if (stride == 1) m4d_launch((wgrad_thin_kernel<16, 1>), dim3(n_wg), dim3(256), 0, s, x, g, b, h, w, oh, ow, cout, pt, pl, n_wg * 4, workspace);
    else m4d_launch((wgrad_thin_kernel<16, 2>), dim3(n_wg), dim3(256), 0, s, x, g, b, h, w, oh, ow, cout, pt, pl, n_wg * 4, workspace);
    m4d_launch(wgrad_cin3_reduce_kernel, dim3((9 * cin * cout + 63) / 64), dim3(256), 0, s,
                       (const float*)workspace, n_wg, 9 * cin * cout, dw);
    return M4D_LAUNCH_RESULT();
  }
  WgradArgs a;
  a.x = x; a.g = g; a.b = b; a.h = h; a.w = w; a.cin = cin; a.oh = oh; a.ow = ow; a.cout = cout; a.stride = stride;
  a.pt = same_pad_before(h, stride); a.pl = same_pad_before(w, stride);
  a.mblk = (cout + 31) / 32; a.nblk = (cin + 31) / 32;
  a.tiles_x = (ow + kTW - 1) / kTW; a.tiles_y = (oh + kTH - 1) / kTH;
  const long long total_tiles = (long long)a.tiles_x * a.tiles_y * b;
  long long slices = wgrad_max_slices(a.mblk * a.nblk);                     // ~3 workgroups per CU over the whole launch
  if (slices > total_tiles) slices = total_tiles;
  a.slices = (int)slices;
  a.partial = workspace;
  const int XH = (kTH - 1) * stride + 3, XW = (kTW - 1) * stride + 3;
  const size_t lds = (size_t)(kTH * kTW * kGS + XH * XW * kXS) * sizeof(float);
  const bool al = ((((uintptr_t)x | (uintptr_t)g)) & 15u) == 0;
  const int vg = (al && cout % 4 == 0) ? 4 : 1;
  const int vx = (al && cin % 4 == 0) ? 4 : ((al && cin % 2 == 0) ? 2 : 1);
  const dim3 grid(a.mblk * a.nblk, a.slices);
#define M4D_WGRAD_LAUNCH(S, G, X)                                                                                          \
  do {                                                                                                                     \
    M4D_LDS_OPT_IN(&conv3x3_wgrad_kernel<S, G, X>);                                                                        \
    m4d_launch((conv3x3_wgrad_kernel<S, G, X>), grid, dim3(256), lds, s, a);                                      \
  } while (0)
  if (stride == 1) {
    if (vg == 4 && vx == 4) M4D_WGRAD_LAUNCH(1, 4, 4);
    else if (vg == 4 && vx == 2) M4D_WGRAD_LAUNCH(1, 4, 2);
    else M4D_WGRAD_LAUNCH(1, 1, 1);
  } else {
    if (vg == 4 && vx == 4) M4D_WGRAD_LAUNCH(2, 4, 4);
    else M4D_WGRAD_LAUNCH(2, 1, 1);
  }
#undef M4D_WGRAD_LAUNCH
  const long long total = (long long)cout * 9 * cin;
  long long gsz = (total + 63) / 64;
  if (gsz > 8192) gsz = 8192;
  m4d_launch(wgrad_reduce_kernel, dim3((unsigned)gsz), dim3(256), 0, s, (const float*)workspace, a.slices,
                     a.mblk * 32, a.nblk * 32, cout, cin, dw);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_dilate2(const float* g, int b, int oh, int ow, int C, int h, int w, float* out, void* stream) {
  M4D_CHECK_ARG(g && out && b > 0 && oh > 0 && ow > 0 && C > 0 && C % 4 == 0 && h > 0 && w > 0);
  M4D_CHECK_ARG(oh == (h + 1) / 2 && ow == (w + 1) / 2);
  M4D_CHECK_ARG(((((uintptr_t)g | (uintptr_t)out)) & 15u) == 0);
  // forward: out_s2[oy] reads x[2 oy + ky - pt]; its adjoint as a stride-1 'SAME' correlation (pad 1) with the rotated
  // kernel needs the gradient at input row 2 oy + (1 - pt)
  const int dy = 1 - same_pad_before(h, 2), dx = 1 - same_pad_before(w, 2);
  const long long total4 = (long long)b * h * w * (C / 4);
  long long gsz = (total4 + 255) / 256;
  if (gsz > 8192) gsz = 8192;
  m4d_launch(dilate2_kernel, dim3((unsigned)gsz), dim3(256), 0, (hipStream_t)stream, g, oh, ow, C, h, w, dy, dx, total4, out);
  return M4D_LAUNCH_RESULT();
}
