// Latency-first 3x3 'SAME' convolution for SMALL maps (the DispRefiner layers of the coarse pyramid levels at batch 1,
// m4depth_network.py:104-135 at 6x20 ... 48x160 pixels) -- round 5.
//
// Why: at batch 1 the step is  encoder + 3 x (coarse-to-fine chain of one frame) + level 1  (ROCm's hipGraph executor runs the
// frames' chains one after the other), and a chain is ~40 launches of 8-24 us each for layers worth < 1 us
// of chip time.  What a launch costs there is its LATENCY CHAIN, not its work: conv3x3_small6_kernel walks 2-8 K chunks per wave
// one after the other, each chunk = global loads (weights 27 KB + halo) -> registers -> LDS -> MFMA, ~3 us per chunk whatever
// the arithmetic, on 6-60 of the 256 CUs.  This kernel cuts the chain to ONE memory round trip per launch:
//   * K is split over waves AND workgroups until a wave owns at most 2 chunks; everything a wave needs is requested in the
//     first microsecond: its weight fragments for both chunks straight into the registers the MFMAs read (fragment-major
//     packing, 27 x 16 B per lane and chunk, 1-KB coalesced wave loads: no LDS hop for weights), then the workgroup's halo;
//   * partial sums over the K slices of different workgroups are NOT reduced here: slice z writes its raw partial tile into slab
//     z, and the CONSUMING layer adds the slabs in slice order (+ the producer's bias, leaky_relu) while it stages its halo --
//     deterministic, no inter-workgroup synchronisation, no extra launch;
//   * K slices of the waves of one workgroup are added in wave order through LDS (as conv3x3_small6_kernel does);
//   * the waves of a workgroup are  cgw cout groups x kw K sub-slices  (cgw * kw = 4): wide maps share one staged halo between
//     four 32-cout groups, tiny maps spend all four waves on K;
//   * a wave multiplies MT = 1 / 2 / 4 M-tiles (8x4 pixels each) by the same weight fragments (larger maps: fewer weight bytes
//     per flop, fewer workgroups).
// Arithmetic = conv3x3_small6_kernel's: float32 operands as exact sums of three bf16 terms, 6 of the 9 term products on
// v_mfma_f32_32x32x16_bf16, float32 accumulation; the summation ORDER over K differs (chunk -> (slice, wave, round) here), so the
// two kernels agree to float32 rounding, not bitwise; every variant of THIS kernel with the same (kw, s_out) gives the same bits.
#include <cstdlib>
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

typedef float lat_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 lat_bf16x8 __attribute__((ext_vector_type(8)));

struct LatArgs {
  const float* x; long long x_slab; int s_in;       // input: s_in partial slabs (>= 1), x_slab floats apart
  const float* x_bias; float x_slope;               // the PRODUCER's bias / leaky-relu slope, applied while staging; nullptr = x is final
  const unsigned char* wp;                          // pack_conv_weights_lat: [group][chunk][tap][part][lane][8] bf16
  const float* bias; float slope;                   // this layer's epilogue, applied only when s_out == 1
  float* out; long long out_slab; int s_out;        // s_out K slices = gridDim.z; slice z -> out + z * out_slab (raw partial sums)
  int b, h, w, Cin, Cout, n_chunks, n_groups, tiles_x, tiles_y;
  int cgw_log2, chunks_per_slice;                   // waves = (1 << cgw_log2) cout groups x (4 >> cgw_log2) K sub-slices
  int oh, ow, pad_y, pad_x;                         // output size; TF 'SAME' pad before (1 at stride 1; 0 or 1 at stride 2)
};

constexpr int kLatRow = 24;                         // floats per staged halo pixel and chunk: 3 parts x 16 bf16 = 96 bytes

// Stage `n_st` chunks (first chunk c0) of the halo: add the SIN partial slabs in slab order, the producer's bias + leaky_relu,
// zero outside the map, exact 3-way bf16 split, [chunk][pixel][part][16 channels] in LDS.
template <int SIN, int HW, int HP>
__device__ __forceinline__ void lat_stage(const LatArgs& a, float* lds_a, int t, int bi, int gy0, int gx0, int c0, int n_st) {
  constexpr int IB = SIN >= 3 ? 2 : 4;              // items per thread in flight: IB * SIN 16-byte loads
  const int n_items = n_st * HP * 4;
  for (int base = 0; base < n_items; base += 256 * IB) {
    float4 v[IB][SIN];
    int dst[IB], ch[IB];
    bool ok[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int idx = base + i * 256 + t;
      const int idc = min(idx, n_items - 1);
      const int j = idc / (HP * 4), rem = idc - j * (HP * 4);
      const int hp = rem >> 2, q = rem & 3;
      const int gy = gy0 + hp / HW, gx = gx0 + hp % HW;             // (gy0, gx0) = the halo's first input pixel
      ch[i] = (c0 + j) * 16 + 4 * q;
      ok[i] = idx < n_items && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w && ch[i] < a.Cin;
      dst[i] = idx < n_items ? (j * HP + hp) * kLatRow + 2 * q : -1;
      const long long off = (((long long)bi * a.h + min(max(gy, 0), a.h - 1)) * a.w + min(max(gx, 0), a.w - 1)) * a.Cin
                            + min(ch[i], a.Cin - 4);
#pragma unroll
      for (int s = 0; s < SIN; ++s) v[i][s] = *reinterpret_cast<const float4*>(a.x + s * a.x_slab + off);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      float e[4] = {v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w};
#pragma unroll
      for (int s = 1; s < SIN; ++s) { e[0] += v[i][s].x; e[1] += v[i][s].y; e[2] += v[i][s].z; e[3] += v[i][s].w; }
      if (a.x_bias != nullptr) {
        const float4 bb = *reinterpret_cast<const float4*>(a.x_bias + min(ch[i], a.Cin - 4));
        const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float u = e[k] + bv[k]; e[k] = u > 0.f ? u : u * a.x_slope; }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] = ok[i] ? e[k] : 0.f;
      unsigned h0, m0, l0, h1, m1, l1;
      m4d_split3_pair(e[0], e[1], h0, m0, l0);
      m4d_split3_pair(e[2], e[3], h1, m1, l1);
      if (dst[i] >= 0) {
        *reinterpret_cast<uint2*>(lds_a + dst[i]) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(lds_a + dst[i] + 8) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(lds_a + dst[i] + 16) = make_uint2(l0, l1);
      }
    }
  }
}

// MW ("M over waves", MTX * MTY = 4): the four waves of a workgroup are the four M-tiles of its 16x8-pixel tile, every wave all of
// the K slice for the workgroup's ONE cout group (kw = cgw = 1) -- for narrow short-K layers on LARGER maps (the encoder's
// 32 -> 32 and 64 -> 64 stride-2 layers: one or two cout groups, 2-4 chunks), where the other split leaves waves idle.
template <int MTX, int MTY, int STRIDE, bool MW>
__device__ __forceinline__ void lat_body(const LatArgs& a, const int bx, const int by, const int bz, float* lds_dyn) {
  constexpr int MT = MTX * MTY, MA = MW ? 1 : MT;     // M-tiles of the workgroup's tile / accumulators per wave
  static_assert(!MW || MT == 4, "one M-tile per wave");
  constexpr int TW = 8 * MTX, TH = 4 * MTY, HW = (TW - 1) * STRIDE + 3, HP = HW * ((TH - 1) * STRIDE + 3);   // output tile, input halo
  constexpr int kChunkF = HP * kLatRow;               // floats of one staged chunk
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int cgw = MW ? 1 : 1 << a.cgw_log2, kw = MW ? 1 : 4 >> a.cgw_log2;
  const int cg = MW ? 0 : wave & (cgw - 1), ks = MW ? 0 : wave >> a.cgw_log2;
  const int tiles = a.tiles_x * a.tiles_y;
  const int bi = bx / tiles, tile = bx - bi * tiles;
  const int ty0 = (tile / a.tiles_x) * TH, tx0 = (tile % a.tiles_x) * TW;
  const int g = by * cgw + cg;
  const bool active = g < a.n_groups;                 // Cout = 96: the fourth wave of a 4-group workgroup idles
  const int c_begin = bz * a.chunks_per_slice;
  const int c_end = min(a.n_chunks, c_begin + a.chunks_per_slice);
  const int n_slice = max(c_end - c_begin, 0);
  const int rounds = (n_slice + kw - 1) / kw;
  const int n = lane & 31, kh = lane >> 5;
  const int co = g * 32 + n;
  // epilogue operand first: nothing at the end of the launch waits for memory
  const float my_bias = (a.s_out == 1 && active && co < a.Cout) ? a.bias[co] : 0.f;

  lat_f32x16 acc[MA];
#pragma unroll
  for (int m = 0; m < MA; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  const unsigned char* wg = a.wp + ((long long)min(g, a.n_groups - 1) * a.n_chunks) * (27 * 1024) + lane * 16;

  for (int r0 = 0; r0 < rounds; r0 += 2) {
    // ---- this wave's weight fragments of up to two chunks, straight into MFMA operand registers
    float4 bq[2][3][9];
    bool valid[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = c_begin + (r0 + q) * kw + ks;
      valid[q] = active && (r0 + q) < rounds && c < c_end;
      const unsigned char* wc = wg + (long long)min(c, a.n_chunks - 1) * (27 * 1024);
      if (valid[q]) {                                 // (wave-uniform: a wave without a second chunk fetches nothing for it -- these
#pragma unroll                                        //  launches are bound by the bytes ONE CU pulls, 27 KB per wave and chunk)
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) bq[q][p][tp] = *reinterpret_cast<const float4*>(wc + (tp * 3 + p) * 1024);
      }
    }
    // ---- the workgroup's halo of the same chunks (the weights are on their way)
    const int c0 = c_begin + r0 * kw;
    const int n_st = min(2 * kw, c_end - c0);
    switch (a.s_in) {
      case 1: lat_stage<1, HW, HP>(a, lds_dyn, t, bi, ty0 * STRIDE - a.pad_y, tx0 * STRIDE - a.pad_x, c0, n_st); break;
      case 2: lat_stage<2, HW, HP>(a, lds_dyn, t, bi, ty0 * STRIDE - a.pad_y, tx0 * STRIDE - a.pad_x, c0, n_st); break;
      case 3: lat_stage<3, HW, HP>(a, lds_dyn, t, bi, ty0 * STRIDE - a.pad_y, tx0 * STRIDE - a.pad_x, c0, n_st); break;
      default: lat_stage<4, HW, HP>(a, lds_dyn, t, bi, ty0 * STRIDE - a.pad_y, tx0 * STRIDE - a.pad_x, c0, n_st); break;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (valid[q]) {
        const float* ac = lds_dyn + (q * kw + ks) * kChunkF + kh * 4;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const lat_bf16x8 b0 = __builtin_bit_cast(lat_bf16x8, bq[q][0][tp]);
          const lat_bf16x8 b1 = __builtin_bit_cast(lat_bf16x8, bq[q][1][tp]);
          const lat_bf16x8 b2 = __builtin_bit_cast(lat_bf16x8, bq[q][2][tp]);
#pragma unroll
          for (int m = 0; m < MA; ++m) {
            const int mi = MW ? wave : m, my = mi / MTX, mx = mi % MTX;
            const float* ap = ac + (((my * 4 + (n >> 3)) * STRIDE + tp / 3) * HW + (mx * 8 + (n & 7)) * STRIDE + tp % 3) * kLatRow;
            const lat_bf16x8 a0 = *reinterpret_cast<const lat_bf16x8*>(ap);
            const lat_bf16x8 a1 = *reinterpret_cast<const lat_bf16x8*>(ap + 8);
            const lat_bf16x8 a2 = *reinterpret_cast<const lat_bf16x8*>(ap + 16);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[m], 0, 0, 0);       // small terms first
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[m], 0, 0, 0);
          }
        }
      }
    }
    if (r0 + 2 < rounds) __syncthreads();             // the next stage overwrites the halo
  }
  // ---- K sub-slices of the workgroup's waves, added in wave order through LDS
  if (kw > 1) {
    __syncthreads();                                  // the reduction buffer aliases the halo
    float* red = lds_dyn;                             // [kw - 1][cgw][MT][16][64]
    if (ks > 0) {
#pragma unroll
      for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((((ks - 1) * cgw + cg) * MT + m) * 16 + r) * 64 + lane] = acc[m][r];
    }
    __syncthreads();
    if (ks == 0) {
      for (int k = 1; k < kw; ++k)
#pragma unroll
        for (int m = 0; m < MA; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][r] += red[((((k - 1) * cgw + cg) * MT + m) * 16 + r) * 64 + lane];
    }
  }
  if (ks == 0 && active && co < a.Cout) {
    float* op = a.out + bz * a.out_slab + (long long)bi * a.oh * a.ow * a.Cout + co;
    const bool final_out = a.s_out == 1;
#pragma unroll
    for (int m = 0; m < MA; ++m) {
      const int mi = MW ? wave : m, my = mi / MTX, mx = mi % MTX;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mr = (r & 3) + 8 * (r >> 2) + 4 * kh;  // C/D map of the 32x32 MFMA: col = lane & 31, row = mr
        const int oy = ty0 + my * 4 + (mr >> 3), ox = tx0 + mx * 8 + (mr & 7);
        if (oy < a.oh && ox < a.ow) {
          float v = acc[m][r];
          if (final_out) { v += my_bias; v = v > 0.f ? v : v * a.slope; }
          op[((long long)oy * a.ow + ox) * a.Cout] = v;
        }
      }
    }
  }
}

template <int MTX, int MTY, int STRIDE, bool MW = false>
__global__ void __launch_bounds__(256, 1)
conv3x3_lat_kernel(const LatArgs a) {
  extern __shared__ __align__(16) float lds_dyn[];
  lat_body<MTX, MTY, STRIDE, MW>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, lds_dyn);
}

// (Round 5's one-launch CHAIN of these layers -- resident workgroups drawing the separate launches' work items from a ticket
// counter, m4d_conv3x3_lat_chain -- was bit-identical and 2-6x slower than the launches it replaced: ~12.5 us of hand-over per
// item, profiles/r05_lat_chain_vs_separate.txt, DESIGN_HISTORY.md.  Never dispatched; deleted in round 6, ABI 6.)

// out = leaky_relu(bias + slab_0 + slab_1 + ...), slabs added in slab order: the dense form of a partial-sum activation
// (what every consumer of m4d_conv3x3_lat's slabs computes while staging) for consumers that cannot, and for inspection
__global__ void __launch_bounds__(256)
partial_finish_kernel(const float* __restrict__ x, long long slab, int s_in, const float* __restrict__ bias, float slope,
                      long long n4, int C4, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    for (int s = 1; s < s_in; ++s) {
      const float4 u = reinterpret_cast<const float4*>(x + s * slab)[i];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const float4 bb = reinterpret_cast<const float4*>(bias)[i % C4];
    v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
    v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
    v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

template <int MTX, int MTY, int STRIDE, bool MW = false>
int lat_launch(const LatArgs& a, int kw_in, hipStream_t s) {
  constexpr int HP = ((8 * MTX - 1) * STRIDE + 3) * ((4 * MTY - 1) * STRIDE + 3), MT = MTX * MTY;
  const int kw = MW ? 1 : kw_in;
  const int cgw = MW ? 1 : 4 / kw;
  const size_t lds_a = (size_t)2 * kw * HP * kLatRow * 4;
  const size_t lds_r = (size_t)(kw - 1) * cgw * MT * 16 * 64 * 4;
  const size_t lds = lds_a > lds_r ? lds_a : lds_r;
  if (lds > 64 * 1024) M4D_LDS_OPT_IN(&conv3x3_lat_kernel<MTX, MTY, STRIDE, MW>);    // per function and per device (m4d_common.h)
  const dim3 grid((unsigned)(a.b * a.tiles_x * a.tiles_y), (unsigned)((a.n_groups + cgw - 1) / cgw), (unsigned)a.s_out);
  m4d_launch(conv3x3_lat_kernel<MTX, MTY, STRIDE, MW>, grid, dim3(256), lds, s, a);
  return 0;
}

}  // namespace

// argument checks + LatArgs of one layer (kw: 1 for the M-over-waves form)
static int lat_make_args(LatArgs& a, int& kw, const float* x, int s_in, long long x_slab_floats, const float* x_bias, float x_slope,
                         const void* wp, const float* bias, int b, int h, int w, int Cin, int Cout, int stride, float slope,
                         int mt, int s_out, float* out, long long out_slab_floats) {
  M4D_CHECK_ARG(x && wp && bias && out && b > 0 && h > 0 && w > 0 && (stride == 1 || stride == 2));
  M4D_CHECK_ARG(Cin >= 16 && Cin % 4 == 0 && Cout >= 1);
  const int oh = (h + stride - 1) / stride, ow = (w + stride - 1) / stride;
  M4D_CHECK_ARG(s_in >= 1 && s_in <= 4 && (s_in == 1 || x_slab_floats >= (long long)b * h * w * Cin));
  M4D_CHECK_ARG(s_out >= 1 && (s_out == 1 || out_slab_floats >= (long long)b * oh * ow * Cout));
  M4D_CHECK_ARG((mt == 1 || mt == 2 || mt == 4 || mt == 8) && (kw == 1 || kw == 2 || kw == 4));   // mt 8 = "M over waves" (kw unused)
  if (mt == 8) kw = 1;
  a.x = x; a.x_slab = x_slab_floats; a.s_in = s_in; a.x_bias = x_bias; a.x_slope = x_slope;
  a.wp = reinterpret_cast<const unsigned char*>(wp); a.bias = bias; a.slope = slope;
  a.out = out; a.out_slab = out_slab_floats; a.s_out = s_out;
  a.b = b; a.h = h; a.w = w; a.Cin = Cin; a.Cout = Cout; a.oh = oh; a.ow = ow;
  // TF 'SAME': total pad = max((out - 1) * stride + 3 - in, 0), before = total / 2
  a.pad_y = ((oh - 1) * stride + 3 - h > 0 ? (oh - 1) * stride + 3 - h : 0) / 2;
  a.pad_x = ((ow - 1) * stride + 3 - w > 0 ? (ow - 1) * stride + 3 - w : 0) / 2;
  a.n_chunks = (Cin + 15) / 16; a.n_groups = (Cout + 31) / 32;
  M4D_CHECK_ARG(s_out <= a.n_chunks);
  a.cgw_log2 = kw == 1 ? 2 : (kw == 2 ? 1 : 0);
  a.chunks_per_slice = (a.n_chunks + s_out - 1) / s_out;
  M4D_CHECK_ARG((long long)(s_out - 1) * a.chunks_per_slice < a.n_chunks);     // no empty K slice
  const int mtx = mt >= 4 ? 2 : 1, mty = mt >= 2 ? 2 : 1;
  a.tiles_x = (ow + 8 * mtx - 1) / (8 * mtx); a.tiles_y = (oh + 4 * mty - 1) / (4 * mty);
  // LDS: two staged rounds of kw chunks each
  const size_t lds_a = (size_t)2 * kw * ((8 * mtx - 1) * stride + 3) * ((4 * mty - 1) * stride + 3) * kLatRow * 4;
  M4D_CHECK_ARG(lds_a <= 160 * 1024);
  return 0;
}

static int lat_launch_any(const float* x, int s_in, long long x_slab_floats, const float* x_bias, float x_slope,
                          const void* wp, const float* bias, int b, int h, int w, int Cin, int Cout, int stride, float slope,
                          int mt, int kw, int s_out, float* out, long long out_slab_floats, void* stream) {
  LatArgs a;
  const int rc = lat_make_args(a, kw, x, s_in, x_slab_floats, x_bias, x_slope, wp, bias, b, h, w, Cin, Cout, stride, slope, mt, s_out,
                               out, out_slab_floats);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  int rl;
  if (stride == 1) {
    if (mt == 1) rl = lat_launch<1, 1, 1>(a, kw, s);
    else if (mt == 2) rl = lat_launch<1, 2, 1>(a, kw, s);
    else if (mt == 4) rl = lat_launch<2, 2, 1>(a, kw, s);
    else rl = lat_launch<2, 2, 1, true>(a, kw, s);
  } else {
    if (mt == 1) rl = lat_launch<1, 1, 2>(a, kw, s);
    else if (mt == 2) rl = lat_launch<1, 2, 2>(a, kw, s);
    else if (mt == 4) rl = lat_launch<2, 2, 2>(a, kw, s);
    else rl = lat_launch<2, 2, 2, true>(a, kw, s);
  }
  if (rl) return rl;
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_conv3x3_lat(const float* x, int s_in, long long x_slab_floats, const float* x_bias, float x_slope,
                               const void* wp, const float* bias, int b, int h, int w, int Cin, int Cout, float slope,
                               int mt, int kw, int s_out, float* out, long long out_slab_floats, void* stream) {
  return lat_launch_any(x, s_in, x_slab_floats, x_bias, x_slope, wp, bias, b, h, w, Cin, Cout, 1, slope, mt, kw, s_out, out,
                        out_slab_floats, stream);
}

extern "C" int m4d_conv3x3s_lat(const float* x, int s_in, long long x_slab_floats, const float* x_bias, float x_slope,
                                const void* wp, const float* bias, int b, int h, int w, int Cin, int Cout, int stride, float slope,
                                int mt, int kw, int s_out, float* out, long long out_slab_floats, void* stream) {
  return lat_launch_any(x, s_in, x_slab_floats, x_bias, x_slope, wp, bias, b, h, w, Cin, Cout, stride, slope, mt, kw, s_out, out,
                        out_slab_floats, stream);
}

extern "C" int m4d_partial_finish(const float* x, int s_in, long long x_slab_floats, const float* bias, float slope,
                                  long long pixels, int C, float* out, void* stream) {
  M4D_CHECK_ARG(x && bias && out && s_in >= 1 && pixels > 0 && C >= 4 && C % 4 == 0);
  M4D_CHECK_ARG(s_in == 1 || x_slab_floats >= pixels * C);
  const long long n4 = pixels * C / 4;
  const long long blocks = (n4 + 255) / 256;
  m4d_launch(partial_finish_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream,
             x, x_slab_floats, s_in, bias, slope, n4, C / 4, out);
  return M4D_LAUNCH_RESULT();
}
