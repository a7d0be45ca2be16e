// 3x3 'SAME' convolution (stride 1 or 2, TensorFlow padding rule) on NHWC float32 with the Keras
// bias add and the leaky_relu that follows it fused into the epilogue (m4depth_network.py:104-135:
// the seven DispRefiner convolutions; :63-72: the stride-1 and stride-2 encoder convolutions) --
// the "next" row f-2 of SURVEY section 8: all 73.6 GMAC of a frame.
//
// Implicit GEMM on the f32-input matrix cores: v_mfma_f32_32x32x2_f32 is exact float32 (bitwise
// an fmaf chain in k order: one rounding per product-accumulate, no wider accumulator), runs at
// the fp32 vector peak (157 TFLOP/s) and is DETERMINISTIC -- unlike several of the MIOpen
// solvers this replaces (atomic split-K; tools/debug_determinism3.py).
//
//   M = pixels, N = output channels, K = 9 taps x Cin.
//   workgroup = 4 waves = one 16x8 pixel tile x BN = 32*NT output channels; wave w owns tile
//   rows 2w, 2w+1 (32 pixels = one MFMA M-tile) and NT accumulators (32 x 32 each).
//   K is walked in chunks of 16 input channels; per chunk the (16+2)x(8+2) halo of the input is
//   staged once in LDS and used by all 9 taps (the 3x3 window is an LDS address offset), the
//   weights of 3 taps at a time.  Channels are stored even-first/odd-last inside a chunk so
//   that the two k-lanes of the MFMA (lane>>5) read the 8 k-steps of a chunk as two
//   ds_read_b128; rows are padded to 20 floats (conflict-free 16-byte reads).
//   The global loads of the NEXT stage are issued into registers before the MFMAs of the
//   current one (software pipeline), so one workgroup per SIMD already hides memory latency.
#include <cstdlib>
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  const float* x; const float* wp; const float* bias; float* out;
  int b, h, w, Cin, Cout, CoutPad, n_chunks, tiles_x, tiles_y;
  int oh, ow, pad_y, pad_x;           // output size and TF 'SAME' pad before (1 at stride 1; 0 or 1 at stride 2)
  float slope;
  int ksplit, chunks_per_split;       // split-K (coarse pyramid levels): partial sums -> ws, reduced in order
  float* ws;
  // DINL variant (encoder level 0): x is the RAW first convolution; DomainNormalization + leaky_relu(dn_slope) are applied
  // while the halo is committed to LDS (m4depth_network.py:44-48,82-84)
  const float* dn_mean; const float* dn_var; const float* dn_scale; const float* dn_bias; float dn_slope;
  int ablate;                         // profiling only (M4D_CONV_ABLATE): 1 = no re-staging after the first stage, 2 = no MFMAs, 4 = no stores
};

constexpr int kTW = 16, kTH = 8;                    // output tile
constexpr int kKC = 16, kRS = 20;                   // chunk size, LDS row stride (floats)

// TS = taps staged per pipeline stage (3 or 9).  Narrow workgroups (NT <= 2: small layers, coarse
// levels, K-splits) do little MFMA work per stage, so they stage all 9 taps of a chunk at once:
// a third of the barriers and global round trips on what is a latency-bound launch.
// DB = double-buffered weight stages (wide stride-1 variants): the next stage's weights are written
// to the other LDS buffer BEFORE the MFMA burst of the current stage and its global loads are issued
// two stages ahead, so a stage costs one barrier instead of two and the LDS writes hide under MFMAs.
// MINB = workgroups per CU the register allocation must allow.  The widest variant needs 190 registers
// (2 waves per SIMD) when left alone; bounded to 168 (accumulators in VGPRs, no scratch) a third workgroup per
// CU fits and fills the barrier / staging bubbles of the other two: +8 % at batch 32 (9.84 -> 9.08 ms for the
// level-1 128->128 layer), but at batch 1 the 960 workgroups of that layer then run as 768 + 192 (a 25 %-full
// second round) and the launch gets slower -- so the host picks MINB = 3 only for grids of >= 4 full rounds.
template <int NT, int TS, int STRIDE, bool DB, int MINB, bool DINL = false>
__global__ void __launch_bounds__(256, MINB)
conv3x3_mfma_kernel(const ConvArgs a) {
  constexpr int kHWT = (kTW - 1) * STRIDE + 3, kHHT = (kTH - 1) * STRIDE + 3, kHP = kHWT * kHHT;   // input halo: 18x10 / 33x17
  constexpr int BN = 32 * NT;
  constexpr int SPC = 9 / TS;                         // stages per chunk
  constexpr int A_F2 = kHP * (kKC / 2);               // float2 loads to stage one halo chunk (1440)
  constexpr int A_PER = (A_F2 + 255) / 256;           // 6
  constexpr int B_F4 = TS * BN * (kKC / 4);           // float4 loads to stage TS taps of weights
  constexpr int B_PER = (B_F4 + 255) / 256;           // 6 at NT = 4, TS = 3
  extern __shared__ __align__(16) float lds_dyn[];
  float* lds_a = lds_dyn;                             // [kHP][kRS]
  float* lds_b = lds_dyn + kHP * kRS;                 // [TS * BN][kRS]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int tile = blockIdx.x;
  const int tile_y = (tile / a.tiles_x) * kTH, tile_x = (tile % a.tiles_x) * kTW;
  const int n0 = blockIdx.y * BN;
  const int bi = blockIdx.z / a.ksplit, ks = blockIdx.z - bi * a.ksplit;
  const int chunk_lo = ks * a.chunks_per_split;
  const int chunk_hi = min(chunk_lo + a.chunks_per_split, a.n_chunks);
  const float* ximg = a.x + (long long)bi * a.h * a.w * a.Cin;

  float2 ra[A_PER];
  float4 rb[B_PER];

  // ---- stage loaders (global -> registers) and committers (registers -> LDS)
  auto load_a = [&](int chunk) {
    const int c0 = chunk * kKC;
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const int idx = u * 256 + t;
      const int hp = idx >> 3, k2 = (idx & 7) * 2;     // halo pixel, channel pair inside the chunk
      const int gy = tile_y * STRIDE - a.pad_y + hp / kHWT, gx = tile_x * STRIDE - a.pad_x + hp % kHWT;
      ra[u] = make_float2(0.f, 0.f);
      if (idx < A_F2 && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w && c0 + k2 < a.Cin) {
        const float* px = ximg + ((long long)gy * a.w + gx) * a.Cin + c0 + k2;
        if (a.Cin & 1) { ra[u].x = px[0]; if (c0 + k2 + 1 < a.Cin) ra[u].y = px[1]; }     // 3-channel input image
        else ra[u] = *reinterpret_cast<const float2*>(px);
      }
    }
  };
  // DINL: this lane always holds the same channel pair (2 * (t & 7)) of some halo pixel; the 8 lanes of a pixel
  // are adjacent, so the per-pixel l2 norm over the 16 channels is three xor-shuffles.
  float dn_mu[2] = {0.f, 0.f}, dn_dv[2] = {1.f, 1.f}, dn_sc[2] = {1.f, 1.f}, dn_bs[2] = {0.f, 0.f};
  if (DINL) {
    const int c = 2 * (t & 7);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      dn_mu[e] = a.dn_mean[bi * 16 + c + e];
      dn_dv[e] = 1.0f / (a.dn_var[bi * 16 + c + e] + 1e-12f);   // (x - mean) / (var + 1e-12), variance not std (:47), as a multiply
      dn_sc[e] = a.dn_scale[c + e];
      dn_bs[e] = a.dn_bias[c + e];
    }
  }
  auto commit_a = [&]() {
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const int idx = u * 256 + t;
      const int hp = idx >> 3, kp = idx & 7;
      float2 v = ra[u];
      if (DINL) {
        const float nx = (v.x - dn_mu[0]) * dn_dv[0], ny = (v.y - dn_mu[1]) * dn_dv[1];
        float ss = nx * nx + ny * ny;
        ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));                          // tf.math.l2_normalize
        float ox = dn_sc[0] * (nx * inv) + dn_bs[0], oy = dn_sc[1] * (ny * inv) + dn_bs[1];
        ox = ox > 0.f ? ox : ox * a.dn_slope; oy = oy > 0.f ? oy : oy * a.dn_slope;
        const int hpc = hp < kHP ? hp : kHP - 1;
        const int gy = tile_y * STRIDE - a.pad_y + hpc / kHWT, gx = tile_x * STRIDE - a.pad_x + hpc % kHWT;
        const bool inside = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;                  // the padding is zeros of the NORMALISED map
        v = make_float2(inside ? ox : 0.f, inside ? oy : 0.f);
      }
      if (idx < A_F2) {                                // even channel -> slot kp, odd channel -> slot 8 + kp
        lds_a[hp * kRS + kp] = v.x;
        lds_a[hp * kRS + 8 + kp] = v.y;
      }
    }
  };
  auto load_b = [&](int chunk, int s) {
    // wp layout: [chunk][tap][CoutPad][16 (even-first)] -> rows n0..n0+BN of taps TS*s..TS*s+TS-1
#pragma unroll
    for (int u = 0; u < B_PER; ++u) {
      const int idx = u * 256 + t;
      const int row = idx >> 2, c4 = idx & 3;          // row = tap_local * BN + n
      const int tl = row / BN, n = row % BN;
      rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < B_F4)
        rb[u] = *reinterpret_cast<const float4*>(
            a.wp + ((((long long)chunk * 9 + TS * s + tl) * a.CoutPad + n0 + n) * kKC + c4 * 4));
    }
  };
  auto commit_b = [&](int buf) {
#pragma unroll
    for (int u = 0; u < B_PER; ++u) {
      const int idx = u * 256 + t;
      const int row = idx >> 2, c4 = idx & 3;
      if (idx < B_F4) *reinterpret_cast<float4*>(lds_b + buf * (TS * BN * kRS) + row * kRS + c4 * 4) = rb[u];
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

  const int m = lane & 31, kh = lane >> 5;
  const int prow = 2 * wave + (m >> 4), pcol = m & 15;        // pixel of this lane inside the tile
  const float* a_lane = lds_a + (prow * STRIDE * kHWT + pcol * STRIDE) * kRS + kh * 8;
  const float* b_lane = lds_b + m * kRS + kh * 8;

  // Operand fragments of one tap: 8 k-steps of A (this lane's pixel) and of B (this lane's output channel, NT tiles).
  struct Frag { float av[8]; float bv[NT][8]; };
  auto read_frag = [&](int s, int buf, int tl, Frag& f) {
    const int ky = TS == 9 ? tl / 3 : s, kx = TS == 9 ? tl % 3 : tl;
    const float* ap = a_lane + (ky * kHWT + kx) * kRS;
    const float4 a0 = *reinterpret_cast<const float4*>(ap);
    const float4 a1 = *reinterpret_cast<const float4*>(ap + 4);
    f.av[0] = a0.x; f.av[1] = a0.y; f.av[2] = a0.z; f.av[3] = a0.w;
    f.av[4] = a1.x; f.av[5] = a1.y; f.av[6] = a1.z; f.av[7] = a1.w;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float* bp = b_lane + buf * (TS * BN * kRS) + (tl * BN + nt * 32) * kRS;
      const float4 b0 = *reinterpret_cast<const float4*>(bp);
      const float4 b1 = *reinterpret_cast<const float4*>(bp + 4);
      f.bv[nt][0] = b0.x; f.bv[nt][1] = b0.y; f.bv[nt][2] = b0.z; f.bv[nt][3] = b0.w;
      f.bv[nt][4] = b1.x; f.bv[nt][5] = b1.y; f.bv[nt][6] = b1.z; f.bv[nt][7] = b1.w;
    }
  };
  auto mfma_frag = [&](const Frag& f) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.av[ks], f.bv[nt][ks], acc[nt], 0, 0, 0);
  };
  // PIPE: the LDS reads of tap t+1 are issued before the MFMA burst of tap t (two fragment sets live), so the
  // ~150-cycle LDS round trip of a tap hides under 8*NT MFMAs instead of stalling the wave between bursts.
  constexpr bool PIPE = (MINB == 2) && (NT == 4) && (STRIDE == 1);
  auto mfma_stage = [&](int s, int buf) {           // TS taps x 8 k-steps x NT tiles
    if constexpr (PIPE) {
      Frag f[2];
      read_frag(s, buf, 0, f[0]);
#pragma unroll
      for (int tl = 0; tl < TS; ++tl) {
        if (tl + 1 < TS) read_frag(s, buf, tl + 1, f[(tl + 1) & 1]);
        mfma_frag(f[tl & 1]);
      }
    } else {
#pragma unroll
      for (int tl = 0; tl < TS; ++tl) {
        const int ky = TS == 9 ? tl / 3 : s, kx = TS == 9 ? tl % 3 : tl;
        const float* ap = a_lane + (ky * kHWT + kx) * kRS;
        const float4 a0 = *reinterpret_cast<const float4*>(ap);
        const float4 a1 = *reinterpret_cast<const float4*>(ap + 4);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        float bv[NT][8];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float* bp = b_lane + buf * (TS * BN * kRS) + (tl * BN + nt * 32) * kRS;
          const float4 b0 = *reinterpret_cast<const float4*>(bp);
          const float4 b1 = *reinterpret_cast<const float4*>(bp + 4);
          bv[nt][0] = b0.x; bv[nt][1] = b0.y; bv[nt][2] = b0.z; bv[nt][3] = b0.w;
          bv[nt][4] = b1.x; bv[nt][5] = b1.y; bv[nt][6] = b1.z; bv[nt][7] = b1.w;
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], bv[nt][ks], acc[nt], 0, 0, 0);
      }
    }
  };

  const int n_stages = (chunk_hi - chunk_lo) * SPC;
  load_a(chunk_lo);
  load_b(chunk_lo, 0);
  commit_a();
  commit_b(0);
  __syncthreads();
  if (DB) {
    if (n_stages > 1) load_b(chunk_lo + 1 / SPC, 1 % SPC);          // stage 1 (registers), committed in stage 0
    for (int st = 0; st < n_stages; ++st) {
      const int s = st % SPC, cur = st & 1;
      const bool has_next = st + 1 < n_stages;
      const int nchunk = chunk_lo + (st + 1) / SPC, ns = (st + 1) % SPC;
      if (has_next) commit_b(cur ^ 1);               // buffer last read in stage st-1: free since that barrier
      if (st + 2 < n_stages) load_b(chunk_lo + (st + 2) / SPC, (st + 2) % SPC);
      if (has_next && ns == 0) load_a(nchunk);       // next stage opens a new chunk: its halo, committed below
      mfma_stage(s, cur);
      if (has_next && ns == 0) {
        __syncthreads();                             // every wave is done with this chunk's halo
        commit_a();
      }
      __syncthreads();
    }
  } else {
    for (int st = 0; st < n_stages; ++st) {
      const int s = st % SPC;
      const bool has_next = st + 1 < n_stages;
      const int nchunk = chunk_lo + (st + 1) / SPC, ns = (st + 1) % SPC;
      const bool restage = has_next && !(a.ablate & 1);
      if (restage) {
        load_b(nchunk, ns);
        if (ns == 0) load_a(nchunk);
      }
      if (!(a.ablate & 2)) mfma_stage(s, 0);
      __syncthreads();                               // all waves done with lds_b (and lds_a when ns == 0)
      if (restage) {
        commit_b(0);
        if (ns == 0) commit_a();
      }
      __syncthreads();
    }
  }

  // ---- epilogue: + bias, leaky_relu, NHWC store.  C/D map of the 32x32 MFMA:
  //      col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  if (a.ksplit > 1) {                                  // raw partial sums; conv_splitk_reduce_kernel finishes
    float* wimg = a.ws + ((long long)ks * a.b + bi) * a.oh * a.ow * a.CoutPad;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = n0 + nt * 32 + m;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mr = (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int oy = tile_y + 2 * wave + (mr >> 4), ox = tile_x + (mr & 15);
        if (oy < a.oh && ox < a.ow) wimg[((long long)oy * a.ow + ox) * a.CoutPad + co] = acc[nt][r];
      }
    }
    return;
  }
  if (a.ablate & 4) { if (acc[0][0] != 1.2345e-30f) return; }   // profiling only: no output stores (the compare keeps the MFMAs alive)
  float* oimg = a.out + (long long)bi * a.oh * a.ow * a.Cout;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = n0 + nt * 32 + m;
    const float bias = co < a.Cout ? a.bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mr = (r & 3) + 8 * (r >> 2) + 4 * kh;  // pixel index inside the wave's 32
      const int oy = tile_y + 2 * wave + (mr >> 4), ox = tile_x + (mr & 15);
      if (co < a.Cout && oy < a.oh && ox < a.ow) {
        float v = acc[nt][r] + bias;
        v = v > 0.f ? v : v * a.slope;
        oimg[((long long)oy * a.ow + ox) * a.Cout + co] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Small maps (the coarsest pyramid levels: 120 - 2000 pixels).  The kernel above needs split-K there to produce enough
// workgroups, i.e. TWO launches per layer (partial sums + ordered reduce) on what is a pure latency chain -- five such
// layers per level, and the first frame's coarse-level chain runs with the chip otherwise idle (tools/step_profile.py).
// Here the K split happens INSIDE the workgroup: a workgroup is one 8x4-pixel tile (one MFMA M-tile) x 32 output
// channels, and its four waves walk interleaved K chunks (chunk c -> wave c % 4) completely independently -- each wave
// stages its own halo + weight chunk in its own LDS region, no workgroup barrier inside the K loop -- then the four
// 32x32 partial tiles are added in wave order through LDS, + bias, leaky_relu, store.  One launch per layer,
// deterministic.  Same packed weights as the kernel above.
constexpr int kSW = 8, kSH = 4;                      // output tile
constexpr int kSHW = kSW + 2, kSHP = (kSW + 2) * (kSH + 2);   // halo 10 x 6 = 60 pixels
constexpr int kSA = kSHP * kRS;                      // floats of one wave's halo chunk   (4.8 KB)
constexpr int kSB = 9 * 32 * kRS;                    // floats of one wave's weight chunk (23 KB)

// STRIDE 2 (the coarse stride-2 layers of the encoder): the halo of the 8x4 output tile is 17 x 9 pixels, the A fragment of
// output pixel (y, x) and tap (ky, kx) sits at halo pixel (2 y + ky, 2 x + kx); TF 'SAME' pad before = a.pad_y / a.pad_x.
template <int STRIDE>
__global__ void __launch_bounds__(256, 1)            // one workgroup per CU (111 / 141 KB of LDS): the whole register file is available
conv3x3_small_kernel(const ConvArgs a) {
  constexpr int kSHW = (kSW - 1) * STRIDE + 3, kSHH = (kSH - 1) * STRIDE + 3, kSHP = kSHW * kSHH;   // halo 10 x 6 / 17 x 9 pixels
  constexpr int kSA = kSHP * kRS;                     // floats of one wave's halo chunk (4.8 / 12.2 KB)
  constexpr int A_PER = (kSHP * 4 + 63) / 64;         // 4 / 10 float4 per lane
  constexpr int B_PER = (9 * 32 * 4) / 64;            // 18 float4 per lane
  extern __shared__ __align__(16) float lds_dyn[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  float* lds_a = lds_dyn + wave * (kSA + kSB);
  float* lds_b = lds_a + kSA;
  const int tile = blockIdx.x;
  const int tile_y = (tile / a.tiles_x) * kSH, tile_x = (tile % a.tiles_x) * kSW;
  const int n0 = blockIdx.y * 32;
  const int bi = blockIdx.z;
  const float* ximg = a.x + (long long)bi * a.h * a.w * a.Cin;

  // hoisted loader geometry: halo pixel (clamped) and validity per A slot, weight row offset per B slot
  int a_off[A_PER], a_dst[A_PER];
  unsigned a_ok = 0;
  const int aq = lane & 3;                            // channels 4aq .. 4aq+3 of the chunk
#pragma unroll
  for (int u = 0; u < A_PER; ++u) {
    const int idx = u * 64 + lane;
    const int hp = min(idx >> 2, kSHP - 1);
    const int gy = tile_y * STRIDE - a.pad_y + hp / kSHW, gx = tile_x * STRIDE - a.pad_x + hp % kSHW;
    const bool ok = (idx >> 2) < kSHP && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    a_ok |= ok ? (1u << u) : 0u;
    a_off[u] = (min(max(gy, 0), a.h - 1) * a.w + min(max(gx, 0), a.w - 1)) * a.Cin + 4 * aq;
    a_dst[u] = (idx >> 2) < kSHP ? hp * kRS + 2 * aq : -1;
  }
  int b_off[B_PER];
#pragma unroll
  for (int u = 0; u < B_PER; ++u) {
    const int idx = u * 64 + lane;
    const int row = idx >> 2, c4 = idx & 3;           // row = tap * 32 + n
    b_off[u] = (((row >> 5) * a.CoutPad + n0 + (row & 31)) * kKC + c4 * 4);
  }
  const long long b_chunk = 9LL * a.CoutPad * kKC;

  float4 ra[A_PER], rb0[B_PER / 2], rb1[B_PER / 2];   // two halves: one 72-dword array is not kept in registers by hipcc
  bool ra_ch_ok = true;
  auto load_chunk = [&](int chunk) __attribute__((always_inline)) {   // unconditional loads (clamped), validity applied at the commit
    const int c0 = chunk * kKC;
    ra_ch_ok = c0 + 4 * aq < a.Cin;
    const int cc = min(c0, a.Cin - 4 - 4 * aq);       // keeps the 16-byte read inside the pixel's channel run
#pragma unroll
    for (int u = 0; u < A_PER; ++u) ra[u] = *reinterpret_cast<const float4*>(ximg + a_off[u] + cc);
    const float* wb = a.wp + chunk * b_chunk;
#pragma unroll
    for (int u = 0; u < B_PER / 2; ++u) {
      rb0[u] = *reinterpret_cast<const float4*>(wb + b_off[u]);
      rb1[u] = *reinterpret_cast<const float4*>(wb + b_off[B_PER / 2 + u]);
    }
  };
  auto commit_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const bool ok = ((a_ok >> u) & 1u) && ra_ch_ok;
      const float4 v = ra[u];
      if (a_dst[u] >= 0) {                            // even channels -> slots 0..7, odd channels -> slots 8..15
        *reinterpret_cast<float2*>(lds_a + a_dst[u]) = make_float2(ok ? v.x : 0.f, ok ? v.z : 0.f);
        *reinterpret_cast<float2*>(lds_a + a_dst[u] + 8) = make_float2(ok ? v.y : 0.f, ok ? v.w : 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < B_PER / 2; ++u) {
      const int idx = u * 64 + lane, idx1 = (B_PER / 2 + u) * 64 + lane;
      const float4 v0 = rb0[u], v1 = rb1[u];         // component-wise: a whole-struct copy out of the array keeps it in scratch
      *reinterpret_cast<float4*>(lds_b + (idx >> 2) * kRS + (idx & 3) * 4) = make_float4(v0.x, v0.y, v0.z, v0.w);
      *reinterpret_cast<float4*>(lds_b + (idx1 >> 2) * kRS + (idx1 & 3) * 4) = make_float4(v1.x, v1.y, v1.z, v1.w);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int m = lane & 31, kh = lane >> 5;
  const float* a_lane = lds_a + ((m >> 3) * STRIDE * kSHW + (m & 7) * STRIDE) * kRS + kh * 8;
  const float* b_lane = lds_b + m * kRS + kh * 8;
  auto compute_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* ap = a_lane + ((tap / 3) * kSHW + (tap % 3)) * kRS;
      const float* bp = b_lane + tap * 32 * kRS;
      const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
      const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], bv[ks], acc, 0, 0, 0);
    }
  };

  // this wave's chunks: wave, wave + 4, ...  The next chunk's loads are in flight while the current one is multiplied;
  // the LDS region is private to the wave, whose DS operations execute in order: no workgroup barrier here.
  // The last round is peeled: it loads nothing (rounds 1-4 kept the loop uniform with a surplus, never committed load -- which
  // the __syncthreads() below then waited for: a memory round trip at the end of every launch of a latency-bound kernel).
  if (wave < a.n_chunks) {
    load_chunk(wave);
    int c = wave;
    for (; c + 4 < a.n_chunks; c += 4) {
      commit_chunk();
      __builtin_amdgcn_wave_barrier();
      load_chunk(c + 4);                               // unconditional inside the loop (a load under a branch is waited for at the merge)
      compute_chunk();
      __builtin_amdgcn_wave_barrier();
    }
    commit_chunk();
    __builtin_amdgcn_wave_barrier();
    compute_chunk();
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();                                     // the reduction buffer aliases the staging regions

  // ---- the four partial tiles, added in wave order, + bias, leaky_relu, NHWC store
  constexpr int kRP = 36;                              // floats per pixel row of a partial tile: 32 couts + 4 pad
  float* red = lds_dyn;                                // [4 waves][32 pixels][kRP]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mr = (r & 3) + 8 * (r >> 2) + 4 * kh;    // C/D map of the 32x32 MFMA: col = lane & 31, row = mr
    red[(wave * 32 + mr) * kRP + m] = acc[r];
  }
  __syncthreads();
  {
    const int p = t >> 3, cq = t & 7;                  // pixel of the tile, output-channel quad
    const int oy = tile_y + (p >> 3), ox = tile_x + (p & 7);
    const int co = n0 + 4 * cq;
    if (oy < a.oh && ox < a.ow && co < a.Cout) {
      const float4 s0 = *reinterpret_cast<const float4*>(red + (0 * 32 + p) * kRP + 4 * cq);
      const float4 s1 = *reinterpret_cast<const float4*>(red + (1 * 32 + p) * kRP + 4 * cq);
      const float4 s2 = *reinterpret_cast<const float4*>(red + (2 * 32 + p) * kRP + 4 * cq);
      const float4 s3 = *reinterpret_cast<const float4*>(red + (3 * 32 + p) * kRP + 4 * cq);
      const float sum[4] = {((s0.x + s1.x) + s2.x) + s3.x, ((s0.y + s1.y) + s2.y) + s3.y,
                            ((s0.z + s1.z) + s2.z) + s3.z, ((s0.w + s1.w) + s2.w) + s3.w};
      float* op = a.out + (((long long)bi * a.oh + oy) * a.ow + ox) * a.Cout + co;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (co + e < a.Cout) {
          float v = sum[e] + a.bias[co + e];
          op[e] = v > 0.f ? v : v * a.slope;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same small-map kernel with float32 operands on the BF16 matrix cores (see m4d_wino6.hip for the arithmetic: every
// float32 value = the exact sum of three bf16 terms, 6 of the 9 term products kept, float32 accumulation).  In a DIRECT
// convolution a staged input value meets 9 taps x 32 output channels, so the split (13 VALU instructions per pair of
// values, once per value while the halo chunk is committed to LDS) costs next to nothing and the MFMA work of a 16-channel
// chunk drops from 72 x 64 cycles (v_mfma_f32_32x32x2_f32) to 54 x 32 (v_mfma_f32_32x32x16_bf16) -- these launches are a
// pure per-wave latency chain (one wave per SIMD, most of the chip idle), so that IS their run time.  Weights split on the
// host (network_ops.pack_conv_weights_small6: [chunk][tap][CoutPad][part][16] bf16, 96 bytes per (tap, cout) row, the
// layout the LDS image keeps); LDS per wave: halo 60 x 96 B + weights 288 x 96 B = 33 KB.
typedef __bf16 s6_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s6_bf16x2 __attribute__((ext_vector_type(2)));
constexpr int kS6Row = 24;                           // floats per LDS row: 3 parts x 16 bf16 = 96 bytes
constexpr int kS6A = kSHP * kS6Row;                  // one wave's halo chunk   (5.8 KB)
constexpr int kS6B = 9 * 32 * kS6Row;                // one wave's weight chunk (27.6 KB)

__device__ __forceinline__ unsigned s6_pk(float a, float b) {
  s6_bf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}

__global__ void __launch_bounds__(256, 1)
conv3x3_small6_kernel(const ConvArgs a) {
  constexpr int A_PER = (kSHP * 4 + 63) / 64;         // 4 float4 per lane (halo pixel x channel quad)
  constexpr int B_PER = (9 * 32 * 6) / 64;            // 27 float4 per lane (row x 16-byte piece)
  extern __shared__ __align__(16) float lds_dyn[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  float* lds_a = lds_dyn + wave * (kS6A + kS6B);
  float* lds_b = lds_a + kS6A;
  const int tile = blockIdx.x;
  const int tile_y = (tile / a.tiles_x) * kSH, tile_x = (tile % a.tiles_x) * kSW;
  const int n0 = blockIdx.y * 32;
  const int bi = blockIdx.z;
  const float* ximg = a.x + (long long)bi * a.h * a.w * a.Cin;
  const unsigned char* wsplit = reinterpret_cast<const unsigned char*>(a.wp);

  int a_off[A_PER], a_dst[A_PER];
  unsigned a_ok = 0;
  const int aq = lane & 3;                            // channels 4aq .. 4aq+3 of the chunk
#pragma unroll
  for (int u = 0; u < A_PER; ++u) {
    const int idx = u * 64 + lane;
    const int hp = min(idx >> 2, kSHP - 1);
    const int gy = tile_y - 1 + hp / kSHW, gx = tile_x - 1 + hp % kSHW;
    const bool ok = (idx >> 2) < kSHP && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    a_ok |= ok ? (1u << u) : 0u;
    a_off[u] = (min(max(gy, 0), a.h - 1) * a.w + min(max(gx, 0), a.w - 1)) * a.Cin + 4 * aq;
    a_dst[u] = (idx >> 2) < kSHP ? hp * kS6Row + 2 * aq : -1;        // float index of this quad's 8 bytes inside part 0
  }
  int b_off[B_PER];                                   // byte offsets inside a chunk's weight block
#pragma unroll
  for (int u = 0; u < B_PER; ++u) {
    const int idx = u * 64 + lane;
    const int row = idx / 6, c6 = idx - row * 6;      // row = tap * 32 + n
    b_off[u] = (((row >> 5) * a.CoutPad + n0 + (row & 31)) * 96 + c6 * 16);
  }
  const long long b_chunk = 9LL * a.CoutPad * 96;     // bytes

  float4 ra[1][A_PER], rb[1][3][B_PER / 3];           // thirds: one 108-dword array is not kept in registers by hipcc
  bool ra_ch_ok[1] = {true};
  auto load_chunk = [&](int chunk, int set) __attribute__((always_inline)) {
    const int c0 = chunk * kKC;
    ra_ch_ok[set] = c0 + 4 * aq < a.Cin;
    const int cc = min(c0, a.Cin - 4 - 4 * aq);
#pragma unroll
    for (int u = 0; u < A_PER; ++u) ra[set][u] = *reinterpret_cast<const float4*>(ximg + a_off[u] + cc);
    const unsigned char* wb = wsplit + chunk * b_chunk;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int u = 0; u < B_PER / 3; ++u) rb[set][g][u] = *reinterpret_cast<const float4*>(wb + b_off[g * (B_PER / 3) + u]);
  };
  auto commit_chunk = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const bool ok = ((a_ok >> u) & 1u) && ra_ch_ok[set];
      const float v[4] = {ok ? ra[set][u].x : 0.f, ok ? ra[set][u].y : 0.f, ok ? ra[set][u].z : 0.f, ok ? ra[set][u].w : 0.f};
      unsigned p0[2], p1[2], p2[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {                   // exact 3-way split of a pair: hi, mid (round to nearest), lo (exact)
        const float x0 = v[2 * e], x1 = v[2 * e + 1];
        const unsigned q0 = s6_pk(x0, x1);
        const float r0 = x0 - __builtin_bit_cast(float, q0 << 16), r1 = x1 - __builtin_bit_cast(float, q0 & 0xffff0000u);
        const unsigned q1 = s6_pk(r0, r1);
        const float s0 = r0 - __builtin_bit_cast(float, q1 << 16), s1 = r1 - __builtin_bit_cast(float, q1 & 0xffff0000u);
        p0[e] = q0; p1[e] = q1;
        p2[e] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
      }
      if (a_dst[u] >= 0) {
        *reinterpret_cast<uint2*>(lds_a + a_dst[u]) = make_uint2(p0[0], p0[1]);
        *reinterpret_cast<uint2*>(lds_a + a_dst[u] + 8) = make_uint2(p1[0], p1[1]);
        *reinterpret_cast<uint2*>(lds_a + a_dst[u] + 16) = make_uint2(p2[0], p2[1]);
      }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int u = 0; u < B_PER / 3; ++u) {
        const int idx = (g * (B_PER / 3) + u) * 64 + lane;
        const float4 w4 = rb[set][g][u];              // component-wise: a whole-struct copy out of the array keeps it in scratch
        *reinterpret_cast<float4*>(lds_b + idx * 4) = make_float4(w4.x, w4.y, w4.z, w4.w);     // rows of 6 pieces, contiguous
      }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int m = lane & 31, kh = lane >> 5;
  const float* a_lane = lds_a + ((m >> 3) * kSHW + (m & 7)) * kS6Row + kh * 4;       // + part * 8 floats
  const float* b_lane = lds_b + m * kS6Row + kh * 4;
  auto compute_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* ap = a_lane + ((tap / 3) * kSHW + (tap % 3)) * kS6Row;
      const float* bp = b_lane + tap * 32 * kS6Row;
      s6_bf16x8 av[3], bv[3];
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        av[part] = *reinterpret_cast<const s6_bf16x8*>(ap + part * 8);
        bv[part] = *reinterpret_cast<const s6_bf16x8*>(bp + part * 8);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[2], acc, 0, 0, 0);       // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[2], bv[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[0], acc, 0, 0, 0);
    }
  };

  // (Measured: a second chunk of operands in flight -- 2 x 108 staging registers -- is slower: 24.2 vs 22.5 us on the
  // 472-channel layer; the chunk time is set by how many bytes ONE CU keeps in flight, not by this wave's prefetch depth.)
  // (last round peeled: no surplus load for the __syncthreads() below to wait for -- see conv3x3_small_kernel)
  if (wave < a.n_chunks) {
    load_chunk(wave, 0);
    int c = wave;
    for (; c + 4 < a.n_chunks; c += 4) {
      commit_chunk(0);
      __builtin_amdgcn_wave_barrier();
      load_chunk(c + 4, 0);
      compute_chunk();
      __builtin_amdgcn_wave_barrier();
    }
    commit_chunk(0);
    __builtin_amdgcn_wave_barrier();
    compute_chunk();
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();                                     // the reduction buffer aliases the staging regions

  constexpr int kRP = 36;
  float* red = lds_dyn;                                // [4 waves][32 pixels][kRP]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mr = (r & 3) + 8 * (r >> 2) + 4 * kh;
    red[(wave * 32 + mr) * kRP + m] = acc[r];
  }
  __syncthreads();
  {
    const int p = t >> 3, cq = t & 7;
    const int oy = tile_y + (p >> 3), ox = tile_x + (p & 7);
    const int co = n0 + 4 * cq;
    if (oy < a.oh && ox < a.ow && co < a.Cout) {
      const float4 s0 = *reinterpret_cast<const float4*>(red + (0 * 32 + p) * kRP + 4 * cq);
      const float4 s1 = *reinterpret_cast<const float4*>(red + (1 * 32 + p) * kRP + 4 * cq);
      const float4 s2 = *reinterpret_cast<const float4*>(red + (2 * 32 + p) * kRP + 4 * cq);
      const float4 s3 = *reinterpret_cast<const float4*>(red + (3 * 32 + p) * kRP + 4 * cq);
      const float sum[4] = {((s0.x + s1.x) + s2.x) + s3.x, ((s0.y + s1.y) + s2.y) + s3.y,
                            ((s0.z + s1.z) + s2.z) + s3.z, ((s0.w + s1.w) + s2.w) + s3.w};
      float* op = a.out + (((long long)bi * a.oh + oy) * a.ow + ox) * a.Cout + co;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (co + e < a.Cout) {
          float v = sum[e] + a.bias[co + e];
          op[e] = v > 0.f ? v : v * a.slope;
        }
      }
    }
  }
}

// out = leaky_relu(bias + sum_ks ws[ks]) with the partial sums added in split order (deterministic)
__global__ void __launch_bounds__(256)
conv_splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias, long long pixels, int Cout,
                          int CoutPad, int ksplit, float slope, float* __restrict__ out) {
  const long long total = pixels * Cout;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / Cout;
    const int co = (int)(i - p * Cout);
    float v = ws[p * CoutPad + co];
    for (int k = 1; k < ksplit; ++k) v = v + ws[((long long)k * pixels + p) * CoutPad + co];
    v = v + bias[co];
    out[i] = v > 0.f ? v : v * slope;
  }
}

template <int NT, int TS, int STRIDE, int MINB>
void launch_conv(const ConvArgs& a, dim3 grid, hipStream_t s) {
  constexpr bool DB = false;   // double-buffered weight stages: implemented, bit-identical, measured 2-4 % SLOWER (b=1 and b=32) -> off
  constexpr int HP = ((kTW - 1) * STRIDE + 3) * ((kTH - 1) * STRIDE + 3);
  constexpr size_t lds = (size_t)(HP * kRS + (DB ? 2 : 1) * TS * 32 * NT * kRS) * sizeof(float);
  if (lds > 64 * 1024) M4D_LDS_OPT_IN(&conv3x3_mfma_kernel<NT, TS, STRIDE, DB, MINB>);
  m4d_launch((conv3x3_mfma_kernel<NT, TS, STRIDE, DB, MINB>), grid, dim3(256), lds, s, a);
}

template <int STRIDE>
void dispatch_conv(const ConvArgs& a, int nt, dim3 grid, hipStream_t s) {
  const long long blocks = (long long)grid.x * grid.y * grid.z;
  static long long occ3_min_blocks = -1;       // M4D_CONV_OCC3_MIN_BLOCKS: A/B switch for the 3-workgroups-per-CU variant
  if (occ3_min_blocks < 0) { const char* e = getenv("M4D_CONV_OCC3_MIN_BLOCKS"); occ3_min_blocks = e ? atoll(e) : 4 * 768; }
  // MINB pins each variant to the occupancy its natural register allocation had (2-3-3-4 workgroups per CU at
  // stride 1, 1-2-2-2 at stride 2) -- an explicit bound of 1 would let the scheduler hoist loads until only one fits.
  switch (nt) {
    case 4:
      if (STRIDE == 1 && blocks >= occ3_min_blocks) launch_conv<4, 3, STRIDE, STRIDE == 1 ? 3 : 1>(a, grid, s);
      else launch_conv<4, 3, STRIDE, STRIDE == 1 ? 2 : 1>(a, grid, s);
      break;
    // (2 workgroups per CU for small grids of the 96/64-wide variants was tried too: no measurable difference)
    case 3: launch_conv<3, 3, STRIDE, STRIDE == 1 ? 3 : 2>(a, grid, s); break;
    case 2: launch_conv<2, STRIDE == 1 ? 9 : 3, STRIDE, STRIDE == 1 ? 3 : 2>(a, grid, s); break;
    default: launch_conv<1, 9, STRIDE, STRIDE == 1 ? 4 : 2>(a, grid, s); break;
  }
}

}  // namespace

extern "C" int m4d_conv3x3_bias_act(const float* x, const float* wp, const float* bias, int b, int h, int w,
                                    int Cin, int Cout, int CoutPad, float slope, float* out, void* stream) {
  return m4d_conv3x3s_bias_act_ws(x, wp, bias, b, h, w, Cin, Cout, CoutPad, 1, slope, out, nullptr, 0, stream);
}

// Small maps: one launch with the K split inside the workgroup instead of split-K + reduce (conv3x3_small_kernel).
static int launch_conv3x3_small(const float* x, const float* wp, const float* bias, int b, int h, int w, int Cin, int Cout,
                                int CoutPad, int stride, float slope, float* out, void* stream) {
  M4D_CHECK_ARG(x && wp && bias && out && b > 0 && h > 0 && w > 0 && Cout > 0 && (stride == 1 || stride == 2));
  M4D_CHECK_ARG(Cin >= 16 && Cin % 4 == 0 && CoutPad % 32 == 0 && CoutPad >= Cout);
  M4D_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)wp)) & 15u) == 0);
  ConvArgs a;
  a.x = x; a.wp = wp; a.bias = bias; a.out = out; a.b = b; a.h = h; a.w = w; a.Cin = Cin; a.Cout = Cout;
  a.CoutPad = CoutPad; a.n_chunks = (Cin + kKC - 1) / kKC;
  a.dn_mean = a.dn_var = a.dn_scale = a.dn_bias = nullptr; a.dn_slope = 1.0f;
  a.ablate = 0; a.slope = slope;
  a.oh = (h + stride - 1) / stride; a.ow = (w + stride - 1) / stride;
  const int tot_y = (a.oh - 1) * stride + 3 - h, tot_x = (a.ow - 1) * stride + 3 - w;   // TF 'SAME'
  a.pad_y = (tot_y > 0 ? tot_y : 0) / 2; a.pad_x = (tot_x > 0 ? tot_x : 0) / 2;
  a.tiles_x = (a.ow + kSW - 1) / kSW; a.tiles_y = (a.oh + kSH - 1) / kSH;
  a.ksplit = 1; a.chunks_per_split = a.n_chunks; a.ws = nullptr;
  const int halo = ((kSW - 1) * stride + 3) * ((kSH - 1) * stride + 3);
  const size_t lds = (size_t)4 * (halo * kRS + kSB) * sizeof(float);                    // 111 / 141 KB
  if (stride == 1) M4D_LDS_OPT_IN(&conv3x3_small_kernel<1>);
  else M4D_LDS_OPT_IN(&conv3x3_small_kernel<2>);
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y), (unsigned)(CoutPad / 32), (unsigned)b);
  if (stride == 1) m4d_launch(conv3x3_small_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, a);
  else m4d_launch(conv3x3_small_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, a);
  return M4D_LAUNCH_RESULT();
}
extern "C" int m4d_conv3x3_small_bias_act(const float* x, const float* wp, const float* bias, int b, int h, int w,
                                          int Cin, int Cout, int CoutPad, float slope, float* out, void* stream) {
  return launch_conv3x3_small(x, wp, bias, b, h, w, Cin, Cout, CoutPad, 1, slope, out, stream);
}
// ... with stride 1 or 2 (TF 'SAME'); out [b, ceil(h/stride), ceil(w/stride), Cout]
extern "C" int m4d_conv3x3s_small_bias_act(const float* x, const float* wp, const float* bias, int b, int h, int w,
                                           int Cin, int Cout, int CoutPad, int stride, float slope, float* out, void* stream) {
  return launch_conv3x3_small(x, wp, bias, b, h, w, Cin, Cout, CoutPad, stride, slope, out, stream);
}

// The same layer with float32 operands on the bf16 matrix cores (conv3x3_small6_kernel); wp6 from pack_conv_weights_small6.
extern "C" int m4d_conv3x3_small6_bias_act(const float* x, const void* wp6, const float* bias, int b, int h, int w,
                                           int Cin, int Cout, int CoutPad, float slope, float* out, void* stream) {
  M4D_CHECK_ARG(x && wp6 && bias && out && b > 0 && h > 0 && w > 0 && Cout > 0);
  M4D_CHECK_ARG(Cin >= 16 && Cin % 4 == 0 && CoutPad % 32 == 0 && CoutPad >= Cout);
  M4D_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)wp6)) & 15u) == 0);
  ConvArgs a;
  a.x = x; a.wp = reinterpret_cast<const float*>(wp6); a.bias = bias; a.out = out; a.b = b; a.h = h; a.w = w; a.Cin = Cin; a.Cout = Cout;
  a.CoutPad = CoutPad; a.n_chunks = (Cin + kKC - 1) / kKC;
  a.dn_mean = a.dn_var = a.dn_scale = a.dn_bias = nullptr; a.dn_slope = 1.0f;
  a.ablate = 0; a.oh = h; a.ow = w; a.pad_y = a.pad_x = 1; a.slope = slope;
  a.tiles_x = (w + kSW - 1) / kSW; a.tiles_y = (h + kSH - 1) / kSH;
  a.ksplit = 1; a.chunks_per_split = a.n_chunks; a.ws = nullptr;
  constexpr size_t lds = (size_t)4 * (kS6A + kS6B) * sizeof(float);                     // 134 KB
  M4D_LDS_OPT_IN(&conv3x3_small6_kernel);
  m4d_launch(conv3x3_small6_kernel, dim3((unsigned)(a.tiles_x * a.tiles_y), (unsigned)(CoutPad / 32), (unsigned)b), dim3(256),
                     lds, (hipStream_t)stream, a);
  return M4D_LAUNCH_RESULT();
}

extern "C" long long m4d_conv3x3_workspace_floats(int b, int h, int w, int CoutPad) {
  return 32LL * b * h * w * CoutPad;                   // up to 32 K-splits
}

extern "C" int m4d_conv3x3_bias_act_ws(const float* x, const float* wp, const float* bias, int b, int h, int w,
                                       int Cin, int Cout, int CoutPad, float slope, float* out, float* workspace,
                                       long long workspace_floats, void* stream) {
  return m4d_conv3x3s_bias_act_ws(x, wp, bias, b, h, w, Cin, Cout, CoutPad, 1, slope, out, workspace,
                                  workspace_floats, stream);
}

// Encoder level 0, second convolution: stride-2 3x3 on the DomainNormalization of x_raw (16 channels), the
// normalisation + leaky_relu(dn_slope) fused into the halo staging.
extern "C" int m4d_conv3x3s2_dinl_bias_act(const float* x_raw, const float* mean, const float* var, const float* dn_scale,
                                           const float* dn_bias, float dn_slope, const float* wp, const float* bias,
                                           int b, int h, int w, int Cout, int CoutPad, float slope, float* out, void* stream) {
  M4D_CHECK_ARG(x_raw && mean && var && dn_scale && dn_bias && wp && bias && out && b > 0 && h > 0 && w > 0 && Cout > 0);
  M4D_CHECK_ARG(CoutPad == 32 && Cout <= 32);                 // one N-tile: the reference's encoder level 0 has 16 filters
  M4D_CHECK_ARG(((((uintptr_t)x_raw) & 7u) == 0) && ((((uintptr_t)wp) & 15u) == 0));
  ConvArgs a;
  a.x = x_raw; a.wp = wp; a.bias = bias; a.out = out; a.b = b; a.h = h; a.w = w; a.Cin = 16; a.Cout = Cout;
  a.CoutPad = CoutPad; a.n_chunks = 1; a.ablate = 0;
  a.oh = (h + 1) / 2; a.ow = (w + 1) / 2;
  const int tph = (a.oh - 1) * 2 + 3 - h, tpw = (a.ow - 1) * 2 + 3 - w;
  a.pad_y = (tph > 0 ? tph : 0) / 2; a.pad_x = (tpw > 0 ? tpw : 0) / 2;
  a.tiles_x = (a.ow + kTW - 1) / kTW; a.tiles_y = (a.oh + kTH - 1) / kTH; a.slope = slope;
  a.ksplit = 1; a.chunks_per_split = 1; a.ws = nullptr;
  a.dn_mean = mean; a.dn_var = var; a.dn_scale = dn_scale; a.dn_bias = dn_bias; a.dn_slope = dn_slope;
  constexpr int HP = ((kTW - 1) * 2 + 3) * ((kTH - 1) * 2 + 3);
  constexpr size_t lds = (size_t)(HP * kRS + 9 * 32 * kRS) * sizeof(float);
  if (lds > 64 * 1024) M4D_LDS_OPT_IN(&conv3x3_mfma_kernel<1, 9, 2, false, 2, true>);
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y), 1, (unsigned)b);
  m4d_launch((conv3x3_mfma_kernel<1, 9, 2, false, 2, true>), grid, dim3(256), lds, (hipStream_t)stream, a);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_conv3x3s_bias_act_ws(const float* x, const float* wp, const float* bias, int b, int h, int w,
                                        int Cin, int Cout, int CoutPad, int stride, float slope, float* out,
                                        float* workspace, long long workspace_floats, void* stream) {
  M4D_CHECK_ARG(x && wp && bias && out && b > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0);
  M4D_CHECK_ARG((stride == 1 || stride == 2) && CoutPad % 32 == 0 && CoutPad >= Cout);
  M4D_CHECK_ARG(((((uintptr_t)x) & (Cin % 2 == 0 ? 7u : 3u)) == 0) && ((((uintptr_t)wp) & 15u) == 0));
  ConvArgs a;
  a.x = x; a.wp = wp; a.bias = bias; a.out = out; a.b = b; a.h = h; a.w = w; a.Cin = Cin; a.Cout = Cout;
  a.CoutPad = CoutPad; a.n_chunks = (Cin + kKC - 1) / kKC;
  a.dn_mean = a.dn_var = a.dn_scale = a.dn_bias = nullptr; a.dn_slope = 1.0f;
  static int ablate = -1;
  if (ablate < 0) { const char* e = getenv("M4D_CONV_ABLATE"); ablate = e ? atoi(e) : 0; }
  a.ablate = ablate;
  // TensorFlow 'SAME': out = ceil(in / stride), total pad = max((out-1)*stride + 3 - in, 0), before = total / 2
  a.oh = (h + stride - 1) / stride; a.ow = (w + stride - 1) / stride;
  const int tph = (a.oh - 1) * stride + 3 - h, tpw = (a.ow - 1) * stride + 3 - w;
  a.pad_y = (tph > 0 ? tph : 0) / 2; a.pad_x = (tpw > 0 ? tpw : 0) / 2;
  a.tiles_x = (a.ow + kTW - 1) / kTW; a.tiles_y = (a.oh + kTH - 1) / kTH; a.slope = slope;
  const long long tiles = (long long)a.tiles_x * a.tiles_y;
  const int n32 = CoutPad / 32;
  // N-tiles per workgroup: as wide as possible (A reuse) while the launch still fills the chip
  int nt = n32 >= 4 && n32 % 4 == 0 ? 4 : (n32 % 3 == 0 ? 3 : (n32 % 2 == 0 ? 2 : 1));
  while (nt > 1 && tiles * b * (n32 / nt) < 512) nt = (nt == 4 || nt == 2) ? nt / 2 : 1;
  // split-K when even the narrowest N split leaves most of the chip idle (coarse pyramid levels):
  // every split keeps >= 2 chunks, partial sums go through the caller's workspace.  (One chunk per split / up to 32 splits
  // measured +0.1-0.5 % end to end, tools/ab_bench.sh -- not worth moving the rounding of every coarse-level layer.)
  const long long blocks = tiles * b * (n32 / nt);
  int ksplit = 1;
  static int sk_target = -1, sk_min_chunks = -1, sk_max = -1;     // tuning knobs (profiling): M4D_CONV_SPLITK_TARGET / _MIN_CHUNKS / _MAX
  if (sk_target < 0) { const char* e = getenv("M4D_CONV_SPLITK_TARGET"); sk_target = e ? atoi(e) : 256; }
  if (sk_min_chunks < 0) { const char* e = getenv("M4D_CONV_SPLITK_MIN_CHUNKS"); sk_min_chunks = e ? atoi(e) : 2; }
  if (sk_max < 0) { const char* e = getenv("M4D_CONV_SPLITK_MAX"); sk_max = e ? atoi(e) : 16; }
  if (workspace != nullptr && blocks < 128 && a.n_chunks >= 4) {
    ksplit = (int)((sk_target + blocks - 1) / blocks);
    if (ksplit > a.n_chunks / sk_min_chunks) ksplit = a.n_chunks / sk_min_chunks;
    if (ksplit > sk_max) ksplit = sk_max;
    while (ksplit > 1 && (long long)ksplit * b * a.oh * a.ow * CoutPad > workspace_floats) --ksplit;
  }
  a.chunks_per_split = (a.n_chunks + ksplit - 1) / ksplit;
  ksplit = (a.n_chunks + a.chunks_per_split - 1) / a.chunks_per_split;     // no empty split
  a.ksplit = ksplit; a.ws = workspace;
  const dim3 grid((unsigned)tiles, (unsigned)(n32 / nt), (unsigned)(b * ksplit));
  hipStream_t s = (hipStream_t)stream;
  if (stride == 1) dispatch_conv<1>(a, nt, grid, s);
  else dispatch_conv<2>(a, nt, grid, s);
  if (ksplit > 1) {
    const long long pixels = (long long)b * a.oh * a.ow;
    long long g = (pixels * Cout + 255) / 256;
    if (g > 2048) g = 2048;
    m4d_launch(conv_splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, workspace, bias, pixels, Cout,
                       CoutPad, ksplit, slope, out);
  }
  return M4D_LAUNCH_RESULT();
}
