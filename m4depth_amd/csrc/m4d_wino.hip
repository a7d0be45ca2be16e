// 3x3 stride-1 'SAME' convolution + bias + leaky_relu, Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
// The wide refiner / encoder layers are MFMA-bound (m4d_conv.hip runs them at 0.6-0.8 of the 157 TFLOP/s
// fp32 peak), so the remaining lever is arithmetic: F(2x2,3x3) computes a 2x2 output tile from a 4x4 input
// tile with 16 multiplies per (input channel, output channel) instead of 36 -- 2.25x fewer MFMA flops for the
// same result up to float32 rounding (measured error 1.5-1.8x that of direct float32 summation; the
// transforms only add and halve: B^T, A^T have entries in {0, +-1}, G in {0, +-1/2, 1}).
//
//   V = B^T d B        (4x4 input tile d, per input channel)          -- computed in the kernel, through LDS
//   U = G g G^T        (3x3 filter g, per (cin, cout))                -- computed once on the host (pack)
//   M[p] = sum_c V[p][c] * U[p][c]          for the 16 positions p    -- 16 independent GEMMs on MFMA
//   Y = A^T M A        (2x2 outputs) + bias, leaky_relu               -- epilogue, through LDS
//
// Workgroup = 4 waves = one 16x8 output tile (32 Winograd tiles = one MFMA M-tile) x BN = 32*NT output
// channels.  Wave w owns positions 4w..4w+3: it needs every tile (A operand: V, shared through LDS) but its
// own slice of U (B operand), which therefore goes global -> registers directly (L2-resident, prefetched
// one position ahead) and never occupies LDS.  K is walked in chunks of 16 input channels: raw halo
// (18x10 pixels) -> LDS, transform (one (tile, channel pair) per lane) -> V in LDS, 4 positions x 8 k-steps
// x NT MFMAs per wave.  After the K loop the 16 positions of a (tile, cout) live in four waves; they meet in
// LDS (two passes of 16 tiles per N-tile) for the output transform.
// Deterministic: fixed summation order, no atomics.
#include <cstdlib>
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoArgs {
  const float* x; const float* wu; const float* bias; float* out;
  int b, h, w, Cin, Cout, CoutPad, n_chunks, tiles_x, tiles_y;
  float slope;
  unsigned long long* stamps;   // profiling only (m4d_wino_set_stamps): per workgroup, per chunk, 5 cycle counters of wave 0
  int ablate;      // profiling only (M4D_WINO_ABLATE): 1 = weights loaded once, 2 = no MFMAs, 4 = no input transform, 8 = no epilogue
};

constexpr int kTW = 16, kTH = 8;                 // output tile (pixels)
constexpr int kHW = kTW + 2, kHH = kTH + 2;      // input halo 18 x 10
constexpr int kHP = kHW * kHH;                   // 180 halo pixels
constexpr int kKC = 16;                          // input channels per chunk
constexpr int kRS = 20;                          // LDS row stride (floats): 16 + 4 pad, conflict-free 16-byte reads
constexpr int kNT32 = 32;                        // Winograd tiles per workgroup (8 x 4) = one MFMA M-tile

template <int NT>
__global__ void __launch_bounds__(256, 2)
conv3x3_wino_kernel(const WinoArgs a) {
  constexpr int BN = 32 * NT;
  constexpr int A_F2 = kHP * (kKC / 2);          // float2 loads per halo chunk (1440)
  constexpr int A_PER = (A_F2 + 255) / 256;      // 6
  extern __shared__ __align__(16) float lds[];
  float* raw = lds;                              // [kHP][kRS]           14.4 KB
  float* V = lds + kHP * kRS;                    // [16][32][kRS]        40 KB   (epilogue: M[16][16][32+1] reuses lds)

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // XCD-aware order: workgroup L runs on XCD L % 8; give every XCD a contiguous band of tiles and put the N-groups of
  // a tile next to each other inside it, so that a halo (and its neighbours') is fetched into that XCD's L2 once.
  const int n_tiles = a.tiles_x * a.tiles_y, n_groups = a.CoutPad / BN;
  int tile, ng;
  {
    const int L = blockIdx.x;
    if ((n_tiles & 7) == 0) {
      const int xcd = L & 7, idx = L >> 3;
      const int tpx = n_tiles >> 3;
      if (a.ablate & 64) { tile = xcd * tpx + idx / n_groups; ng = idx % n_groups; }
      else { ng = idx / tpx; tile = xcd * tpx + idx % tpx; }          // an XCD walks its band once per N-group
    } else {
      tile = L / n_groups;
      ng = L % n_groups;
    }
  }
  // (tried: s_setprio on every second workgroup, and a start-up delay to put the two workgroups of a CU in anti-phase --
  // neither changes the kernel time)
  const int tile_y = (tile / a.tiles_x) * kTH, tile_x = (tile % a.tiles_x) * kTW;
  const int n0 = ng * BN;
  const int bi = blockIdx.y;
  const float* ximg = a.x + (long long)bi * a.h * a.w * a.Cin;

  // Raw halo of the next chunk: one register set, issued right after the commit of the current one so that the loads fly
  // during the transform AND the MFMA phase.  (Tried: two register sets / two chunks ahead -- spills at 256 VGPRs, 2x slower;
  // 32 channels per round trip -- 256 VGPRs, 8 % slower.)
  // The loads are UNCONDITIONAL (coordinates clamped into the image, validity kept as a bit mask and applied at the
  // commit): with `if (inside) r = load` hipcc branches around every load and waits vmcnt(0) at each merge -- six
  // dependent round trips, measured 4.9 k cycles per chunk for this block of code, a third of the chunk period.
  float2 ra[A_PER];
  unsigned ra_valid = 0;
  auto load_raw = [&](int chunk) {
    const int c0 = chunk * kKC;
    ra_valid = 0;
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const int idx = u * 256 + t;
      const int hp = min(idx >> 3, kHP - 1), k2 = (idx & 7) * 2;
      const int gy = tile_y - 1 + hp / kHW, gx = tile_x - 1 + hp % kHW;
      const bool ok = idx < A_F2 && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w && c0 + k2 < a.Cin;
      ra_valid |= ok ? (1u << u) : 0u;
      const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1), cc = min(c0 + k2, a.Cin - 2);
      ra[u] = *reinterpret_cast<const float2*>(ximg + ((long long)cy * a.w + cx) * a.Cin + cc);
    }
  };
  auto commit_raw = [&]() {
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const int idx = u * 256 + t;
      const bool ok = (ra_valid >> u) & 1u;
      const float2 v = make_float2(ok ? ra[u].x : 0.f, ok ? ra[u].y : 0.f);
      if (idx < A_F2) *reinterpret_cast<float2*>(raw + (idx >> 3) * kRS + (idx & 7) * 2) = v;
    }
  };
  // input transform: V[p][tile][c] = (B^T d B)[p]; 32 tiles x 16 channels = 512 items, two per lane (item = (tile, channel),
  // one after the other: 24 live values instead of 48 with channel pairs -- the register file is the scarce resource here).
  // Column transform first, row by row as the inputs arrive.
  auto transform = [&]() {
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
      const int item = it * 256 + t;
      const int tt = item >> 4, ch = item & 15;
      const int tty = tt >> 3, ttx = tt & 7;
      float c[4][4];                               // c = d B
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* rp = raw + ((2 * tty + i) * kHW + 2 * ttx) * kRS + ch;
        const float d0 = rp[0], d1 = rp[kRS], d2 = rp[2 * kRS], d3 = rp[3 * kRS];
        c[i][0] = d0 - d2; c[i][1] = d1 + d2; c[i][2] = d2 - d1; c[i][3] = d1 - d3;
      }
      float* vp = V + tt * kRS + ch;
#pragma unroll
      for (int j = 0; j < 4; ++j) {                // rows: B^T (d B); position p = i * 4 + j
        vp[(0 + j) * kNT32 * kRS] = c[0][j] - c[2][j];
        vp[(4 + j) * kNT32 * kRS] = c[1][j] + c[2][j];
        vp[(8 + j) * kNT32 * kRS] = c[2][j] - c[1][j];
        vp[(12 + j) * kNT32 * kRS] = c[1][j] - c[3][j];
      }
    }
  };

  f32x16 acc[4][NT];
#pragma unroll
  for (int pi = 0; pi < 4; ++pi)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pi][nt][r] = 0.f;

  const int m = lane & 31, kh = lane >> 5;
  // B operand: wu[chunk][pos][CoutPad][16] (natural channel order); this lane reads channels 8kh..8kh+7 of cout n0+nt*32+m
  const float* wlane = a.wu + ((long long)(4 * wave) * a.CoutPad + n0 + m) * kKC + kh * 8;
  const long long w_pos = (long long)a.CoutPad * kKC;            // stride between positions
  const long long w_chunk = 16 * w_pos;
  const float* vlane = V + ((4 * wave) * kNT32 + m) * kRS + kh * 8;

  // weights: ring of 4 buffers of one (position, N-tile) fragment each (2 float4), walked in the order the MFMAs use
  // them and loaded 3 fragments ahead (an L2 round trip is longer than one fragment's 8 MFMAs = 512 cycles).  The ring
  // index (NT * (4 * chunk + pi) + nt) % 4 does not depend on the chunk: every register index below is a constant.
  float4 bq[4][2];
  const int n_frag = a.n_chunks * 4 * NT;
  auto load_b = [&](int q, int buf) {              // q = NT * (4 * chunk + pi) + nt
    const int step = q / NT, nt = q % NT;
    const float* wp = wlane + (step >> 2) * w_chunk + (step & 3) * w_pos + nt * 32 * kKC;
    bq[buf][0] = *reinterpret_cast<const float4*>(wp);
    bq[buf][1] = *reinterpret_cast<const float4*>(wp + 4);
  };

  load_raw(0);
  load_b(0, 0);
  if (n_frag > 1) load_b(1, 1);
  if (n_frag > 2) load_b(2, 2);
  unsigned long long* st = (a.stamps != nullptr && t == 0 && blockIdx.y == 0 && blockIdx.x < 512)
                               ? a.stamps + (long long)blockIdx.x * (5 * 40 + 2) : nullptr;
  if (st) st[0] = __builtin_readcyclecounter();
  for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
    if (st && chunk < 40) st[2 + chunk * 5 + 0] = __builtin_readcyclecounter();
    if (!(a.ablate & 32) || chunk == 0) commit_raw();
    if (st && chunk < 40 && (a.ablate & 128)) st[2 + chunk * 5 + 1] = __builtin_readcyclecounter();     // variant: stamp 1 = after the commit
    if (chunk + 1 < a.n_chunks && !(a.ablate & 16)) load_raw(chunk + 1);
    if (st && chunk < 40 && (a.ablate & 256)) st[2 + chunk * 5 + 1] = __builtin_readcyclecounter();     // variant: stamp 1 = after the load issue
    __syncthreads();                               // raw visible; every wave is done with V
    if (st && chunk < 40 && !(a.ablate & 384)) st[2 + chunk * 5 + 1] = __builtin_readcyclecounter();
    if (!(a.ablate & 4)) transform();
    if (st && chunk < 40) st[2 + chunk * 5 + 2] = __builtin_readcyclecounter();
    __syncthreads();
    if (st && chunk < 40) st[2 + chunk * 5 + 3] = __builtin_readcyclecounter();
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) {
      const float* ap = vlane + pi * kNT32 * kRS;
      const float4 a0 = *reinterpret_cast<const float4*>(ap);
      const float4 a1 = *reinterpret_cast<const float4*>(ap + 4);
      const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ring = (NT * pi + nt) % 4;
        const int q = NT * (4 * chunk + pi) + nt;
        if (q + 3 < n_frag && !(a.ablate & 1)) load_b(q + 3, (ring + 3) % 4);
        const float4 b0 = bq[ring][0], b1 = bq[ring][1];
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        if (!(a.ablate & 2)) {
#pragma unroll
          for (int ks = 0; ks < 8; ++ks)
            acc[pi][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], bv[ks], acc[pi][nt], 0, 0, 0);
        } else {
          acc[pi][nt][0] += av[0] * bv[0];
        }
      }
    }
    if (st && chunk < 40) st[2 + chunk * 5 + 4] = __builtin_readcyclecounter();
  }
  if (st) st[2 + (a.n_chunks < 40 ? a.n_chunks : 40) * 5 - 1] = __builtin_readcyclecounter();
  __syncthreads();                                 // the epilogue's M buffer aliases raw / V
  if ((a.ablate & 8) && acc[0][0][0] != 1.2345e-30f) return;

  // ---- output transform Y = A^T M A.  Wave w holds row i = w of the 4x4 position grid (positions 4w..4w+3 = columns
  // j = 0..3), so the column half, R[i][k] = (M A)[i][k], is done in registers; only the two R values per (tile, cout)
  // and wave meet in LDS, where Y[0][k] = R[0][k] + R[1][k] + R[2][k], Y[1][k] = R[1][k] - R[2][k] - R[3][k].
  // C/D map of the 32x32 MFMA: col (cout) = lane & 31, row (tile) = (reg & 3) + 8 * (reg >> 2) + 4 * kh.
  constexpr int kMS = 36;                          // row stride (floats): 32 couts + 4 pad (16-byte aligned rows)
  float* Rb = lds;                                 // [4 rows i][2 k][32 tiles][kMS] = 36.9 KB
  float* oimg = a.out + (long long)bi * a.h * a.w * a.Cout;
  const bool vec_ok = (a.Cout & 3) == 0;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r];
      const int trow = (r & 3) + 8 * (r >> 2) + 4 * kh;
      Rb[((wave * 2 + 0) * kNT32 + trow) * kMS + m] = (m0 + m1) + m2;
      Rb[((wave * 2 + 1) * kNT32 + trow) * kMS + m] = (m1 - m2) - m3;
    }
    __syncthreads();
    {                                              // 32 tiles x 8 cout quads = 256 items, one per lane, 16-byte loads / stores
      const int cq = t & 7, tg = t >> 3;           // Winograd tile 0..31 of the workgroup
      const int ty2 = tg >> 3, tx2 = tg & 7;
      const int co = n0 + nt * 32 + 4 * cq;
      float4 rv[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) rv[i][k] = *reinterpret_cast<const float4*>(Rb + ((i * 2 + k) * kNT32 + tg) * kMS + 4 * cq);
      if (co < a.Cout) {
        float bs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bs[e] = co + e < a.Cout ? a.bias[co + e] : 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float* r0 = reinterpret_cast<const float*>(&rv[0][k]); const float* r1 = reinterpret_cast<const float*>(&rv[1][k]);
          const float* r2 = reinterpret_cast<const float*>(&rv[2][k]); const float* r3 = reinterpret_cast<const float*>(&rv[3][k]);
          float y0[4], y1[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v0 = ((r0[e] + r1[e]) + r2[e]) + bs[e];
            float v1 = ((r1[e] - r2[e]) - r3[e]) + bs[e];
            y0[e] = v0 > 0.f ? v0 : v0 * a.slope;
            y1[e] = v1 > 0.f ? v1 : v1 * a.slope;
          }
          const int ox = tile_x + 2 * tx2 + k, oy = tile_y + 2 * ty2;
          if (ox < a.w) {
#pragma unroll
            for (int l = 0; l < 2; ++l) {
              if (oy + l < a.h) {
                float* op = oimg + ((long long)(oy + l) * a.w + ox) * a.Cout + co;
                const float* y = l ? y1 : y0;
                if (vec_ok) m4d_store16(op, y[0], y[1], y[2], y[3]);   // (m4d_common.h: one instruction, not guarded pieces)
                else { for (int e = 0; e < 4; ++e) if (co + e < a.Cout) op[e] = y[e]; }
              }
            }
          }
        }
      }
    }
    if (nt + 1 < NT) __syncthreads();
  }
  if (st) st[1] = __builtin_readcyclecounter();
}

template <int NT>
void launch_wino(const WinoArgs& a, hipStream_t s) {
  constexpr size_t lds = (size_t)(kHP * kRS + 16 * kNT32 * kRS) * sizeof(float);       // 14.4 + 40 KB
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * (a.CoutPad / (32 * NT))), (unsigned)a.b);
  m4d_launch((conv3x3_wino_kernel<NT>), grid, dim3(256), lds, s, a);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Variant 2: the per-phase stamps of the kernel above show its MFMA loop waiting on the weight fragments -- every
// fragment (2 KB per wave) feeds only 8 MFMAs, ~20 TB/s of L2 traffic at the full MFMA rate, more than L2 delivers.
// Here a wave keeps 4 positions x TWO M-tiles (a 16x16-pixel workgroup tile = 64 Winograd tiles) x one N-tile, so a
// fragment feeds 16 MFMAs: half the weight traffic per flop.  To keep V (16 positions x 64 tiles) within 48 KB of LDS
// the K chunk is 8 channels; the weight ring is 8 fragments deep (4 VGPRs each, 6 fragments of lookahead).
constexpr int kT2 = 16;                          // output tile 16 x 16
constexpr int kH2 = kT2 + 2;                     // halo 18 x 18
constexpr int kHP2 = kH2 * kH2;                  // 324 halo pixels
constexpr int kC2 = 8;                           // channels per chunk
constexpr int kRS2 = 12;                         // LDS row stride (floats): 8 + 4 pad (48 B: conflict-free 16-byte reads)
constexpr int kNT64 = 64;                        // Winograd tiles per workgroup (8 x 8) = two MFMA M-tiles
constexpr int kWSkew = 64;                       // floats of padding after each position block of the packed weights: the
                                                 // 16 position blocks of a chunk then start on different L2 channels

__global__ void __launch_bounds__(256, 2)
conv3x3_wino2_kernel(const WinoArgs a) {
  constexpr int A_F4 = kHP2 * 2;                 // float4 loads per halo chunk (648)
  constexpr int A_PER = (A_F4 + 255) / 256;      // 3
  extern __shared__ __align__(16) float lds[];
  float* raw = lds;                              // [kHP2][kRS2]         15.2 KB
  float* V = lds + kHP2 * kRS2;                  // [16][64][kRS2]       48 KB

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n_tiles = a.tiles_x * a.tiles_y, n_groups = a.CoutPad / 32;
  int tile, ng;
  {
    const int L = blockIdx.x;
    if ((n_tiles & 7) == 0) {
      const int xcd = L & 7, idx = L >> 3;
      tile = xcd * (n_tiles >> 3) + idx / n_groups;
      ng = idx % n_groups;
    } else {
      tile = L / n_groups;
      ng = L % n_groups;
    }
  }
  const int tile_y = (tile / a.tiles_x) * kT2, tile_x = (tile % a.tiles_x) * kT2;
  const int n0 = ng * 32;
  const int bi = blockIdx.y;
  const float* ximg = a.x + (long long)bi * a.h * a.w * a.Cin;

  float4 ra[A_PER];
  unsigned ra_valid = 0;
  auto load_raw = [&](int chunk) {               // unconditional loads, validity applied at the commit (see above)
    const int c0 = chunk * kC2;
    ra_valid = 0;
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const int idx = u * 256 + t;
      const int hp = min(idx >> 1, kHP2 - 1), k4 = (idx & 1) * 4;
      const int gy = tile_y - 1 + hp / kH2, gx = tile_x - 1 + hp % kH2;
      const bool ok = idx < A_F4 && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w && c0 + k4 < a.Cin;
      ra_valid |= ok ? (1u << u) : 0u;
      const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1), cc = min(c0 + k4, a.Cin - 4);
      ra[u] = *reinterpret_cast<const float4*>(ximg + ((long long)cy * a.w + cx) * a.Cin + cc);
    }
  };
  auto commit_raw = [&]() {
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const int idx = u * 256 + t;
      const bool ok = (ra_valid >> u) & 1u;
      const float4 v = ok ? ra[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < A_F4) *reinterpret_cast<float4*>(raw + (idx >> 1) * kRS2 + (idx & 1) * 4) = v;
    }
  };
  // input transform: lane = (tile, channel pair): 64 tiles x 4 pairs = 256 items, float2 arithmetic
  const int tt = t >> 2, cp = t & 3;
  const int tty = tt >> 3, ttx = tt & 7;
  auto transform = [&]() {
    float2 c[4][4];                                // c = d B, row by row as the inputs arrive
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* rp = raw + ((2 * tty + i) * kH2 + 2 * ttx) * kRS2 + 2 * cp;
      const float2 d0 = *reinterpret_cast<const float2*>(rp), d1 = *reinterpret_cast<const float2*>(rp + kRS2);
      const float2 d2 = *reinterpret_cast<const float2*>(rp + 2 * kRS2), d3 = *reinterpret_cast<const float2*>(rp + 3 * kRS2);
      c[i][0] = make_float2(d0.x - d2.x, d0.y - d2.y);
      c[i][1] = make_float2(d1.x + d2.x, d1.y + d2.y);
      c[i][2] = make_float2(d2.x - d1.x, d2.y - d1.y);
      c[i][3] = make_float2(d1.x - d3.x, d1.y - d3.y);
    }
    float* vp = V + tt * kRS2 + 2 * cp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                  // rows: B^T (d B); position p = i * 4 + j
      *reinterpret_cast<float2*>(vp + (0 + j) * kNT64 * kRS2) = make_float2(c[0][j].x - c[2][j].x, c[0][j].y - c[2][j].y);
      *reinterpret_cast<float2*>(vp + (4 + j) * kNT64 * kRS2) = make_float2(c[1][j].x + c[2][j].x, c[1][j].y + c[2][j].y);
      *reinterpret_cast<float2*>(vp + (8 + j) * kNT64 * kRS2) = make_float2(c[2][j].x - c[1][j].x, c[2][j].y - c[1][j].y);
      *reinterpret_cast<float2*>(vp + (12 + j) * kNT64 * kRS2) = make_float2(c[1][j].x - c[3][j].x, c[1][j].y - c[3][j].y);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int pi = 0; pi < 4; ++pi)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pi][mt][r] = 0.f;

  const int m = lane & 31, kh = lane >> 5;
  // weights: wu[chunk][pos][CoutPad][8]; this lane reads channels 4kh..4kh+3 of cout n0 + m
  const long long w_pos = (long long)a.CoutPad * kC2 + kWSkew;
  const long long w_chunk = 16 * w_pos;
  const float* wlane = a.wu + (4 * wave) * w_pos + (n0 + m) * kC2 + kh * 4;
  const float* vlane = V + ((4 * wave) * kNT64 + m) * kRS2 + kh * 4;

  float4 bq[8];
  const int n_frag = a.n_chunks * 4;
  auto load_b = [&](int q, int buf) { bq[buf] = *reinterpret_cast<const float4*>(wlane + (q >> 2) * w_chunk + (q & 3) * w_pos); };

  unsigned long long* st = (a.stamps != nullptr && t == 0 && blockIdx.y == 0 && blockIdx.x < 512)
                               ? a.stamps + (long long)blockIdx.x * (5 * 40 + 2) : nullptr;
  if (st) st[0] = __builtin_readcyclecounter();
  // (Tried: raw halo double buffered in LDS with the commit of chunk c+1 and the loads of chunk c+2 issued inside the
  // MFMA phase of chunk c -- one phase less per chunk on paper, 5 % slower measured: hipcc places the commit's wait and
  // the address arithmetic in front of the MFMAs instead of between them.)
  load_raw(0);
#pragma unroll
  for (int q = 0; q < 6; ++q)
    if (q < n_frag) load_b(q, q);
  for (int chunk0 = 0; chunk0 < a.n_chunks; chunk0 += 2) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {               // two chunks per iteration: the ring index (4 * chunk + pi) % 8 is static
      const int chunk = chunk0 + cc;
      if (chunk < a.n_chunks) {
        if (st && chunk < 40) st[2 + chunk * 5 + 0] = __builtin_readcyclecounter();
        commit_raw();
        if (chunk + 1 < a.n_chunks) load_raw(chunk + 1);
        __syncthreads();                           // raw visible; every wave is done with V
        if (st && chunk < 40) st[2 + chunk * 5 + 1] = __builtin_readcyclecounter();
        transform();
        if (st && chunk < 40) st[2 + chunk * 5 + 2] = __builtin_readcyclecounter();
        __syncthreads();
        if (st && chunk < 40) st[2 + chunk * 5 + 3] = __builtin_readcyclecounter();
        // A fragments are read one position ahead of the MFMAs that use them (the LDS round trip otherwise sits in front
        // of every 8-MFMA group), and the weight prefetch is unconditional (index clamped): no branch inside the loop.
        float4 af[2][2];
        af[0][0] = *reinterpret_cast<const float4*>(vlane);
        af[0][1] = *reinterpret_cast<const float4*>(vlane + 32 * kRS2);
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
          const int ring = cc * 4 + pi;
          const int q = chunk * 4 + pi;
          load_b(min(q + 6, n_frag - 1), (ring + 6) % 8);
          if (pi + 1 < 4) {
            af[(pi + 1) & 1][0] = *reinterpret_cast<const float4*>(vlane + ((pi + 1) * kNT64) * kRS2);
            af[(pi + 1) & 1][1] = *reinterpret_cast<const float4*>(vlane + ((pi + 1) * kNT64 + 32) * kRS2);
          }
          const float4 b0 = bq[ring];
          const float bv[4] = {b0.x, b0.y, b0.z, b0.w};
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const float4 a0 = af[pi & 1][mt];
            const float av[4] = {a0.x, a0.y, a0.z, a0.w};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
              acc[pi][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], bv[ks], acc[pi][mt], 0, 0, 0);
          }
        }
        if (st && chunk < 40) st[2 + chunk * 5 + 4] = __builtin_readcyclecounter();
      }
    }
  }
  __syncthreads();                                 // the epilogue buffer aliases raw / V

  // ---- output transform, as above; one pass per M-tile
  constexpr int kMS = 36;                          // row stride (floats): 32 couts + 4 pad (16-byte aligned rows)
  float* Rb = lds;                                 // [4 rows i][2 k][32 tiles][kMS] = 36.9 KB
  float* oimg = a.out + (long long)bi * a.h * a.w * a.Cout;
  const bool vec_ok = (a.Cout & 3) == 0;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][mt][r], m1 = acc[1][mt][r], m2 = acc[2][mt][r], m3 = acc[3][mt][r];
      const int trow = (r & 3) + 8 * (r >> 2) + 4 * kh;
      Rb[((wave * 2 + 0) * 32 + trow) * kMS + m] = (m0 + m1) + m2;
      Rb[((wave * 2 + 1) * 32 + trow) * kMS + m] = (m1 - m2) - m3;
    }
    __syncthreads();
    {                                              // 32 tiles x 8 cout quads = 256 items, one per lane, 16-byte loads / stores
      const int cq = t & 7, tl = t >> 3;           // tile inside this M-tile
      const int tg = mt * 32 + tl;                 // Winograd tile 0..63 of the workgroup (8 x 8)
      const int ty2 = tg >> 3, tx2 = tg & 7;
      const int co = n0 + 4 * cq;
      float4 rv[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) rv[i][k] = *reinterpret_cast<const float4*>(Rb + ((i * 2 + k) * 32 + tl) * kMS + 4 * cq);
      if (co < a.Cout) {
        float bs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bs[e] = co + e < a.Cout ? a.bias[co + e] : 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float* r0 = reinterpret_cast<const float*>(&rv[0][k]); const float* r1 = reinterpret_cast<const float*>(&rv[1][k]);
          const float* r2 = reinterpret_cast<const float*>(&rv[2][k]); const float* r3 = reinterpret_cast<const float*>(&rv[3][k]);
          float y0[4], y1[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v0 = ((r0[e] + r1[e]) + r2[e]) + bs[e];
            float v1 = ((r1[e] - r2[e]) - r3[e]) + bs[e];
            y0[e] = v0 > 0.f ? v0 : v0 * a.slope;
            y1[e] = v1 > 0.f ? v1 : v1 * a.slope;
          }
          const int ox = tile_x + 2 * tx2 + k, oy = tile_y + 2 * ty2;
          if (ox < a.w) {
#pragma unroll
            for (int l = 0; l < 2; ++l) {
              if (oy + l < a.h) {
                float* op = oimg + ((long long)(oy + l) * a.w + ox) * a.Cout + co;
                const float* y = l ? y1 : y0;
                if (vec_ok) m4d_store16(op, y[0], y[1], y[2], y[3]);   // (m4d_common.h: one instruction, not guarded pieces)
                else { for (int e = 0; e < 4; ++e) if (co + e < a.Cout) op[e] = y[e]; }
              }
            }
          }
        }
      }
    }
    if (mt == 0) __syncthreads();
  }
  if (st) st[1] = __builtin_readcyclecounter();
}

// ---------------------------------------------------------------------------------------------------------------
// (Tried, measured, removed -- git history has it: "variant 3", the same arithmetic with the waves of a 512-thread
// workgroup SPECIALISED, 4 waves issuing only MFMAs and 4 waves only staging + transforming one chunk ahead into a
// double-buffered V.  Bit-identical, 15 % slower.  The phase stamps show why: while a wave streams fp32 MFMAs, a second
// wave on the same SIMD gets one or two VALU instructions issued per MFMA (64 cycles) -- fp32 MFMA runs at the VALU
// fp32 rate and evidently on the same issue port -- so the ~100-instruction producer stretched to the length of the MFMA
// phase and the barrier made the phases add up instead of overlapping; wave priorities and padding the MFMA stream
// with s_nop changed nothing.  Loads issued four phases ahead made no difference either: it is issue, not latency.)

// ---------------------------------------------------------------------------------------------------------------
// Variant 4 = variant 2's arithmetic (bit-identical results), restructured around what the stamps and counters say
// limits variant 2 (matrix cores busy 48 %): not the amount of VALU work but the phases in which NO wave of a SIMD has
// an MFMA to issue -- two barriers per chunk with the LDS-read -> transform -> LDS-write chain between them.  Here
//   * V is double buffered and the transform of chunk c+1 is issued by every wave BETWEEN ITS OWN MFMA groups of chunk
//     c (pinned with sched_barrier): one barrier per chunk, and the transform's LDS latencies sit under MFMAs;
//   * a workgroup is 8 waves = 4 position rows x TWO N-tiles (64 couts): the transformed tile is used twice, each
//     wave transforms half an item (8 of the 16 positions), and both waves of a SIMD are always in the same phase;
//   * raw halo double buffered as well: commit of chunk c+2 and loads of chunk c+3 ride in the same phase.
// (Measured and dropped: the refiner's 96-wide layer on this kernel with CoutPad = 128, the wave of the all-padding N-tile
// taking part in the staging but issuing no MFMAs: -1.8 % end to end against kernel 2 for that layer -- the half-idle
// workgroup still holds its CU for most of a full one's time.)
// One workgroup per CU (126 KB of LDS in the K loop, 147 KB for the epilogue staging).  Needs CoutPad % 64 == 0.
constexpr int kRawF = kHP2 * kRS2;               // floats per raw buffer   (15.2 KB)
constexpr int kVF = 16 * kNT64 * kRS2;           // floats per V buffer     (48 KB)

template <bool STAMPS>
__global__ void __launch_bounds__(512)
conv3x3_wino4_kernel(const WinoArgs a) {
  constexpr int A_F4 = kHP2 * 2;                 // float4 loads per halo chunk (648)
  extern __shared__ __align__(16) float lds[];
  float* raw = lds;                              // [2][kHP2][kRS2]
  float* V = lds + 2 * kRawF;                    // [2][16][64][kRS2]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);           // scalar: everything derived from it stays off the VALU
  const int pr = wave & 3, nt = wave >> 2;       // position row, N-tile of this wave
  const int n_tiles = a.tiles_x * a.tiles_y, n_groups = a.CoutPad / 64;
  int tile, ng;
  {
    const int L = blockIdx.x;
    if ((n_tiles & 7) == 0) {
      const int xcd = L & 7, idx = L >> 3;
      tile = xcd * (n_tiles >> 3) + idx / n_groups;
      ng = idx % n_groups;
    } else {
      tile = L / n_groups;
      ng = L % n_groups;
    }
  }
  const int tile_y = (tile / a.tiles_x) * kT2, tile_x = (tile % a.tiles_x) * kT2;
  const int n0 = ng * 64 + nt * 32;
  const int bi = blockIdx.y;
  const int n = a.n_chunks, last = n - 1;
  const float* ximg = a.x + (long long)bi * a.h * a.w * a.Cin;

  // ---- raw staging: 648 float4 per chunk over 512 threads (slot 1 only for t < 136); pixel part hoisted
  const int k4 = (t & 1) * 4;
  int pix_off0, pix_off1;
  bool pix_ok0, pix_ok1;
  {
    auto pixel = [&](int idx, int& off, bool& ok) {
      const int hp = min(idx >> 1, kHP2 - 1);
      const int gy = tile_y - 1 + hp / kH2, gx = tile_x - 1 + hp % kH2;
      ok = idx < A_F4 && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      off = (min(max(gy, 0), a.h - 1) * a.w + min(max(gx, 0), a.w - 1)) * a.Cin;
    };
    pixel(t, pix_off0, pix_ok0);
    pixel(512 + t, pix_off1, pix_ok1);
  }
  const int raw_dst0 = (t >> 1) * kRS2 + k4;                           // slot 0: hp = t >> 1
  const int raw_dst1 = min((512 + t) >> 1, kHP2 - 1) * kRS2 + k4;      // slot 1: hp = 256 + (t >> 1)
  const bool slot1 = t < A_F4 - 512;
  // VALU instructions are not free next to fp32 MFMAs (same issue port): the common case -- the whole halo inside the
  // image, a full 8-channel chunk -- is a uniform base + a hoisted 32-bit lane offset, and a plain store at the commit
  const bool interior = tile_y >= 1 && tile_x >= 1 && tile_y + kT2 + 1 <= a.h && tile_x + kT2 + 1 <= a.w;
  const unsigned voff0 = (unsigned)(pix_off0 + k4) * 4u, voff1 = (unsigned)(pix_off1 + k4) * 4u;     // bytes
  float4 ra0, ra1;
  auto load_raw = [&](int chunk) {               // Cin % 8 == 0 (launcher): every chunk is full; loads unconditional (clamped pixel)
    const char* base = reinterpret_cast<const char*>(ximg) + (size_t)chunk * (kC2 * 4);
    ra0 = *reinterpret_cast<const float4*>(base + voff0);
    ra1 = *reinterpret_cast<const float4*>(base + voff1);
  };
  auto commit_raw = [&](float* rb) {
    if (interior) {
      *reinterpret_cast<float4*>(rb + raw_dst0) = ra0;
      if (slot1) *reinterpret_cast<float4*>(rb + raw_dst1) = ra1;
    } else {
      *reinterpret_cast<float4*>(rb + raw_dst0) =
          make_float4(pix_ok0 ? ra0.x : 0.f, pix_ok0 ? ra0.y : 0.f, pix_ok0 ? ra0.z : 0.f, pix_ok0 ? ra0.w : 0.f);
      if (slot1)
        *reinterpret_cast<float4*>(rb + raw_dst1) =
            make_float4(pix_ok1 ? ra1.x : 0.f, pix_ok1 ? ra1.y : 0.f, pix_ok1 ? ra1.z : 0.f, pix_ok1 ? ra1.w : 0.f);
    }
  };

  // ---- input transform: item = (Winograd tile, channel pair) on t & 255; waves 0-3 (hf = 0) produce positions 0-7
  // (rows 0, 1 of B^T (d B)) from input rows 0-2, waves 4-7 positions 8-15 (rows 2, 3) from input rows 1-3
  const int ti = t & 255;
  const int tt = ti >> 2, cp = ti & 3;
  const int tty = tt >> 3, ttx = tt & 7;
  const int hf = nt;                             // scalar
  const int raw_src = ((2 * tty + hf) * kH2 + 2 * ttx) * kRS2 + 2 * cp;
  const int v_dst = tt * kRS2 + 2 * cp + hf * 8 * kNT64 * kRS2;
  float2 d[3][4];
  auto read_raw = [&](const float* rb) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const float2*>(rb + raw_src + (i * kH2 + j) * kRS2);
  };
  auto sub2 = [](float2 x, float2 y) { return make_float2(x.x - y.x, x.y - y.y); };
  auto add2 = [](float2 x, float2 y) { return make_float2(x.x + y.x, x.y + y.y); };
  auto transform_write = [&](float* vb) {        // c = d B per input row, then two rows of B^T c
    float2 c[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      c[i][0] = sub2(d[i][0], d[i][2]);
      c[i][1] = add2(d[i][1], d[i][2]);
      c[i][2] = sub2(d[i][2], d[i][1]);
      c[i][3] = sub2(d[i][1], d[i][3]);
    }
    float* vp = vb + v_dst;
    if (hf == 0) {                               // c[0..2] = rows 0, 1, 2
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<float2*>(vp + (0 + j) * kNT64 * kRS2) = sub2(c[0][j], c[2][j]);
        *reinterpret_cast<float2*>(vp + (4 + j) * kNT64 * kRS2) = add2(c[1][j], c[2][j]);
      }
    } else {                                     // c[0..2] = rows 1, 2, 3
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<float2*>(vp + (0 + j) * kNT64 * kRS2) = sub2(c[1][j], c[0][j]);     // positions 8 + j:  c2 - c1
        *reinterpret_cast<float2*>(vp + (4 + j) * kNT64 * kRS2) = sub2(c[0][j], c[2][j]);     // positions 12 + j: c1 - c3
      }
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int pi = 0; pi < 4; ++pi)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pi][mt][r] = 0.f;

  const int m = lane & 31, kh = lane >> 5;
  // weights: wu[chunk][pos][CoutPad][8]; this lane reads channels 4kh..4kh+3 of cout n0 + m
  const long long w_pos = (long long)a.CoutPad * kC2 + kWSkew;
  const long long w_chunk = 16 * w_pos;
  const char* wbase = reinterpret_cast<const char*>(a.wu + (4 * pr) * w_pos + (long long)n0 * kC2);    // uniform
  const unsigned wlane_off = (unsigned)((m * kC2 + kh * 4) * 4);                                        // bytes, per lane
  const float* vlane0 = V + ((4 * pr) * kNT64 + m) * kRS2 + kh * 4;
  float4 bq[8];
  const int n_frag = n * 4;
  auto load_b = [&](int q, int buf) {
    bq[buf] = *reinterpret_cast<const float4*>(wbase + ((q >> 2) * w_chunk + (q & 3) * w_pos) * 4 + wlane_off);
  };

  // phase stamps (m4d_wino_set_stamps): thread 0 of the first 512 workgroups
  unsigned long long* st = (STAMPS && a.stamps != nullptr && t == 0 && blockIdx.y == 0 && blockIdx.x < 512)
                               ? a.stamps + (long long)blockIdx.x * (5 * 40 + 2) : nullptr;
  if (STAMPS && st) st[0] = __builtin_readcyclecounter();
  // ---- prologue: raw(0) -> V(0), raw(1) committed, raw(2) in registers
  load_raw(0);
#pragma unroll
  for (int q = 0; q < 6; ++q) load_b(min(q, n_frag - 1), q);
  commit_raw(raw);
  load_raw(min(1, last));
  __syncthreads();
  read_raw(raw);
  transform_write(V);
  commit_raw(raw + kRawF);
  load_raw(min(2, last));
  __syncthreads();

  for (int chunk0 = 0; chunk0 < n; chunk0 += 2) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {               // two chunks per iteration: ring index and buffer parity are static
      const int chunk = chunk0 + cc;
      if (chunk < n) {
        const float* vlane = vlane0 + cc * kVF;
        const float* raw_next = raw + (cc ^ 1) * kRawF;    // raw(chunk + 1), committed during the previous phase
        float* raw_free = raw + cc * kRawF;                // raw(chunk) was transformed during the previous phase
        float* v_next = V + (cc ^ 1) * kVF;
        if (STAMPS && st && chunk < 40) st[2 + chunk * 5 + 0] = __builtin_readcyclecounter();
        float4 af[2][2];
        af[0][0] = *reinterpret_cast<const float4*>(vlane);
        af[0][1] = *reinterpret_cast<const float4*>(vlane + 32 * kRS2);
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
          const int ring = cc * 4 + pi;
          const int q = chunk * 4 + pi;
          load_b(min(q + 6, n_frag - 1), (ring + 6) % 8);
          if (pi + 1 < 4) {
            af[(pi + 1) & 1][0] = *reinterpret_cast<const float4*>(vlane + ((pi + 1) * kNT64) * kRS2);
            af[(pi + 1) & 1][1] = *reinterpret_cast<const float4*>(vlane + ((pi + 1) * kNT64 + 32) * kRS2);
          }
          // the side work of this phase, one piece in front of each MFMA group (surplus work past the last chunk is
          // harmless: it lands in buffers nobody reads any more).  (Tried: different slots for the two waves of a SIMD,
          // which otherwise run this stream in lockstep -- slower.)
          if (pi == 0) read_raw(raw_next);
          if (pi == 2) {
            if (STAMPS && st && chunk < 40) st[2 + chunk * 5 + 1] = __builtin_readcyclecounter();
            transform_write(v_next);
            if (STAMPS && st && chunk < 40) st[2 + chunk * 5 + 2] = __builtin_readcyclecounter();
          }
          if (pi == 3) {
            if (STAMPS && st && chunk < 40) st[2 + chunk * 5 + 3] = __builtin_readcyclecounter();
            commit_raw(raw_free);
            load_raw(min(chunk + 3, last));
          }
          __builtin_amdgcn_sched_barrier(0);
          const float4 b0 = bq[ring];
          const float bv[4] = {b0.x, b0.y, b0.z, b0.w};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              const float4 a0 = af[pi & 1][mt];
              const float av[4] = {a0.x, a0.y, a0.z, a0.w};
              acc[pi][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], bv[ks], acc[pi][mt], 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (STAMPS && st && chunk < 40) st[2 + chunk * 5 + 4] = __builtin_readcyclecounter();
        __syncthreads();
      }
    }
  }
  if (STAMPS && st) st[1] = __builtin_readcyclecounter();
  // every wave is past the barrier that ends the last phase: raw / V are free, the epilogue buffer aliases them

  // ---- output transform: rows of A^T (M A) through LDS per (N-tile, M-tile), then one 2x2-output item x 4 couts per thread
  constexpr int kMS = 36;                          // row stride (floats): 32 couts + 4 pad (16-byte aligned rows)
  constexpr int kRbMT = 4 * 2 * 32 * kMS;          // floats per (N-tile, M-tile): [4 rows i][2 k][32 tiles][kMS] = 36.9 KB
  float* Rb = lds;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    float* rb = Rb + (nt * 2 + mt) * kRbMT;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][mt][r], m1 = acc[1][mt][r], m2 = acc[2][mt][r], m3 = acc[3][mt][r];
      const int trow = (r & 3) + 8 * (r >> 2) + 4 * kh;
      rb[((pr * 2 + 0) * 32 + trow) * kMS + m] = (m0 + m1) + m2;
      rb[((pr * 2 + 1) * 32 + trow) * kMS + m] = (m1 - m2) - m3;
    }
  }
  __syncthreads();
  float* oimg = a.out + (long long)bi * a.h * a.w * a.Cout;
  const bool vec_ok = (a.Cout & 3) == 0;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = it * 512 + t;                 // (N-tile, M-tile, tile in M-tile, cout quad)
    const int cq = item & 7, tl = (item >> 3) & 31, mt = (item >> 8) & 1, ont = item >> 9;
    const float* rb = Rb + (ont * 2 + mt) * kRbMT;
    const int tg = mt * 32 + tl;                   // Winograd tile 0..63 of the workgroup (8 x 8)
    const int ty2 = tg >> 3, tx2 = tg & 7;
    const int co = ng * 64 + ont * 32 + 4 * cq;
    float4 rv[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) rv[i][k] = *reinterpret_cast<const float4*>(rb + ((i * 2 + k) * 32 + tl) * kMS + 4 * cq);
    if (co < a.Cout) {
      float bs[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) bs[e] = co + e < a.Cout ? a.bias[co + e] : 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float* r0 = reinterpret_cast<const float*>(&rv[0][k]); const float* r1 = reinterpret_cast<const float*>(&rv[1][k]);
        const float* r2 = reinterpret_cast<const float*>(&rv[2][k]); const float* r3 = reinterpret_cast<const float*>(&rv[3][k]);
        float y0[4], y1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v0 = ((r0[e] + r1[e]) + r2[e]) + bs[e];
          float v1 = ((r1[e] - r2[e]) - r3[e]) + bs[e];
          y0[e] = v0 > 0.f ? v0 : v0 * a.slope;
          y1[e] = v1 > 0.f ? v1 : v1 * a.slope;
        }
        const int ox = tile_x + 2 * tx2 + k, oy = tile_y + 2 * ty2;
        if (ox < a.w) {
#pragma unroll
          for (int l = 0; l < 2; ++l) {
            if (oy + l < a.h) {
              float* op = oimg + ((long long)(oy + l) * a.w + ox) * a.Cout + co;
              const float* y = l ? y1 : y0;
              if (vec_ok) m4d_store16(op, y[0], y[1], y[2], y[3]);   // (m4d_common.h: one instruction, not guarded pieces)
              else { for (int e = 0; e < 4; ++e) if (co + e < a.Cout) op[e] = y[e]; }
            }
          }
        }
      }
    }
  }
  if (STAMPS && st) st[2 + 5 * 40 - 1] = __builtin_readcyclecounter();      // end of the epilogue (slot of chunk 39, never a real chunk here)
}

static unsigned long long* g_wino_stamps = nullptr;
extern "C" void m4d_wino_set_stamps(unsigned long long* device_buffer) { g_wino_stamps = device_buffer; }

extern "C" int m4d_conv3x3_wino_bias_act(const float* x, const float* wu, const float* bias, int b, int h, int w,
                                         int Cin, int Cout, int CoutPad, float slope, float* out, void* stream) {
  M4D_CHECK_ARG(x && wu && bias && out && b > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0);
  M4D_CHECK_ARG(CoutPad % 32 == 0 && CoutPad >= Cout);
  M4D_CHECK_ARG(Cin % 2 == 0 && ((((uintptr_t)x) & 7u) == 0) && ((((uintptr_t)wu) & 15u) == 0));
  WinoArgs a;
  a.x = x; a.wu = wu; a.bias = bias; a.out = out; a.b = b; a.h = h; a.w = w; a.Cin = Cin; a.Cout = Cout;
  a.CoutPad = CoutPad; a.n_chunks = (Cin + kKC - 1) / kKC; a.slope = slope;
  a.tiles_x = (w + kTW - 1) / kTW; a.tiles_y = (h + kTH - 1) / kTH;
  static int ablate = -1;
  if (ablate < 0) { const char* e = getenv("M4D_WINO_ABLATE"); ablate = e ? atoi(e) : 0; }
  a.ablate = ablate;
  a.stamps = g_wino_stamps;

  hipStream_t s = (hipStream_t)stream;
  if ((CoutPad / 32) % 2 == 0) launch_wino<2>(a, s);
  else launch_wino<1>(a, s);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_conv3x3_wino2_bias_act(const float* x, const float* wu8, const float* bias, int b, int h, int w,
                                          int Cin, int Cout, int CoutPad, float slope, float* out, void* stream) {
  M4D_CHECK_ARG(x && wu8 && bias && out && b > 0 && h > 0 && w > 0 && Cin >= 4 && Cout > 0);
  M4D_CHECK_ARG(CoutPad % 32 == 0 && CoutPad >= Cout);
  M4D_CHECK_ARG(Cin % 4 == 0 && ((((uintptr_t)x) & 15u) == 0) && ((((uintptr_t)wu8) & 15u) == 0));
  WinoArgs a;
  a.x = x; a.wu = wu8; a.bias = bias; a.out = out; a.b = b; a.h = h; a.w = w; a.Cin = Cin; a.Cout = Cout;
  a.CoutPad = CoutPad; a.n_chunks = (Cin + kC2 - 1) / kC2; a.slope = slope;
  a.tiles_x = (w + kT2 - 1) / kT2; a.tiles_y = (h + kT2 - 1) / kT2;
  a.ablate = 0; a.stamps = g_wino_stamps;
  constexpr size_t lds = (size_t)(kHP2 * kRS2 + 16 * kNT64 * kRS2) * sizeof(float);     // 15.2 + 48 KB
  M4D_LDS_OPT_IN(&conv3x3_wino2_kernel);           // more than 64 KB of dynamic LDS needs the opt-in
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * (CoutPad / 32)), (unsigned)b);
  // Kernel 4 (512 threads, 64 couts, one workgroup per CU) where its grid still fills the chip and the K loop is long
  // enough to pay for its prologue (tools/bench_wino_variants.py: 3-13 % faster from 240 workgroups up; in isolation slower at 120 and
  // for 2-chunk layers; inside the pipelined step, where other frames' kernels share the chip, it pays from ~100 workgroups:
  // +0.5 % frames/s at batch 1, round 2); otherwise kernel 2 (twice as many, smaller workgroups).  Same results bit for bit.
  // M4D_WINO_VARIANT=2 forces kernel 2, M4D_WINO4_MIN_WG moves the threshold.
  static int variant = -1, min_wg4 = -1;
  if (variant < 0) { const char* e = getenv("M4D_WINO_VARIANT"); variant = e ? atoi(e) : 4; }
  if (min_wg4 < 0) { const char* e = getenv("M4D_WINO4_MIN_WG"); min_wg4 = e ? atoi(e) : 100; }
  if (variant == 4 && CoutPad % 64 == 0 && Cin >= 32 && Cin % 8 == 0 && (long long)a.tiles_x * a.tiles_y * (CoutPad / 64) * b >= min_wg4) {
    constexpr size_t lds4 = (size_t)(4 * 4 * 2 * 32 * 36) * sizeof(float);              // epilogue staging 147 KB (K loop: 126 KB)
    static_assert(lds4 >= (size_t)(2 * kRawF + 2 * kVF) * sizeof(float), "epilogue staging must cover the K-loop buffers");
    if (a.stamps) M4D_LDS_OPT_IN(&conv3x3_wino4_kernel<true>);
    else M4D_LDS_OPT_IN(&conv3x3_wino4_kernel<false>);
    const dim3 grid4((unsigned)(a.tiles_x * a.tiles_y * (CoutPad / 64)), (unsigned)b);
    if (a.stamps) m4d_launch(conv3x3_wino4_kernel<true>, grid4, dim3(512), lds4, (hipStream_t)stream, a);
    else m4d_launch(conv3x3_wino4_kernel<false>, grid4, dim3(512), lds4, (hipStream_t)stream, a);
    return M4D_LAUNCH_RESULT();
  }
  m4d_launch(conv3x3_wino2_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
  return M4D_LAUNCH_RESULT();
}
