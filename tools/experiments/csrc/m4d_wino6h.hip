// The HALF-TILE variant of the bf16-split Winograd convolution (m4d_wino6.hip) for grids that cannot fill the chip: one
// workgroup = a 16x8-pixel tile (32 Winograd tiles = ONE MFMA M-tile) x 64 output channels, all 16 positions in one pass.
//
// Why: at batch 1 level 3 of the 384x1280 pyramid is 30 tiles of 16x16 pixels -- 60 workgroups of m4d_wino6.hip on 256 CUs,
// each a serial chain of 32 positions x ~1550 cycles: 30 us per layer whatever the chip could do (0.08 of the bf16 peak), and
// level 3 sits on the exposed coarse-to-fine chain of a frame.  Half the pixels per workgroup = twice the workgroups, each
// with half the matrix work.
//
// Structure (that of m4d_wino6w.hip, one pass): wave (pr, ch) owns positions (pr, 2 ch) and (pr, 2 ch + 1) for the M-tile
// and both N-tiles: 4 accumulators.  Nothing but the raw halo is shared between waves: every wave streams the B fragments
// of its two positions (12 KB per 16-channel chunk, m4d_wino6's packed layout) by LDS-DMA into a private ring, refilled
// slot by slot for the next chunk as soon as a fragment is in registers (hand-counted vmcnt, no barrier for B); ONE raw
// s_barrier per chunk publishes the raw halo (18x10 pixels, three buffers: a chunk's transformed inputs are produced half a
// chunk ahead of its matrix products).  The price is twice the fragment bytes per MFMA (a fragment meets one M-tile): fine
// where most CUs are idle anyway, which is the only place this kernel is dispatched.
// Bit-identical to conv3x3_wino6_kernel: same products, same accumulation order, same association in the output transform.
#include <type_traits>
#include "m4d_common.h"
#include "../../../include/m4depth_hip.h"
#include "../../../include/m4depth_hip_experiments.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct W6wArgs {
  const float* x; const unsigned char* wu; const float* bias; float* out;
  int b, h, w, Cin, Cout, CoutPad, n_chunks, tiles_x, tiles_y;
  float slope;
  unsigned long long* stamps;   // profiling only (M4D_W6W_ABLATIONS builds): workgroup x wave x 16 s_memtime values
};

constexpr int kTW = 16, kTH = 8;                // output tile (pixels): 8 x 4 Winograd tiles
constexpr int kHH = kTH + 2;                     // halo rows (18 columns as in m4d_wino6.hip)
constexpr int kJ = 10;
constexpr int kRow = 2 * kJ;
constexpr int kQuad = kHH * kRow;                // slots per channel quad (200)
constexpr int kRawUsed = 4 * kQuad;              // 800 slots per 16-channel chunk
constexpr int kRawDma = 13;                      // LDS-DMA instructions per chunk (64 slots each)
constexpr int kRawSlots = kRawDma * 64;          // 832 slots = 13312 B per buffer
constexpr int kRingOff = 3 * kRawSlots * 16;     // three raw buffers, then 8 rings of 12 KB
constexpr int kMS = 36;
constexpr int kStageFloats = 2 * 16 * 32 * kMS;  // [N-tile][position][tile][kMS] = 147456 B

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  bf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lo_f32(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f32(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }


__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv3x3_wino6h_kernel(const W6wArgs a) {
  extern __shared__ __align__(16) float lds[];
  float4* raw = reinterpret_cast<float4*>(lds);                      // [3][kRawSlots]
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;

  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int pr = wv & 3, ch = wv >> 2;                               // position row; column pair (positions 2 ch, 2 ch + 1)
  const int m = lane & 31, kh = lane >> 5;
  const int n_tiles = a.tiles_x * a.tiles_y, n_groups = a.CoutPad / 64;
  int tile, ng;
  {
    const int L = blockIdx.x;
    if ((n_tiles & 7) == 0) {                                        // the N-groups of one pixel tile on one XCD (its L2 holds the halo)
      const int xcd = L & 7, idx = L >> 3;
      tile = xcd * (n_tiles >> 3) + idx / n_groups;
      ng = idx % n_groups;
    } else {
      tile = L / n_groups;
      ng = L % n_groups;
    }
  }
  const int tile_y = (tile / a.tiles_x) * kTH, tile_x = (tile % a.tiles_x) * kTW;
  const int bi = blockIdx.y;
  const int n = a.n_chunks, last = n - 1;
  const float* ximg = a.x + (long long)bi * a.h * a.w * a.Cin;

  // ---- raw halo by LDS-DMA: instruction i fills slots 64 i .. 64 i + 63; wave wv issues i = wv and wv + 8 (clamped to 12:
  // waves 5-7 repeat the last piece); pixels outside the image / pad slots read past num_records -> zeros
  i32x4 rsrc;
  {
    const unsigned long long xa = (unsigned long long)ximg;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(xa & 0xffffffffull));
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)((xa >> 32) & 0xffffull));
    rsrc[2] = a.h * a.w * a.Cin * 4;
    rsrc[3] = 0x00020000;
  }
  unsigned rvoff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = min(wv + 8 * k, kRawDma - 1);
    const int s = i * 64 + lane;
    const int q = s / kQuad, rem = s - q * kQuad;
    const int hy = rem / kRow, r2 = rem - hy * kRow;
    const int e = r2 / kJ, j = r2 - e * kJ;
    const int hx = 2 * j + e;
    const int gy = tile_y - 1 + hy, gx = tile_x - 1 + hx;
    const bool ok = s < kRawUsed && j < 9 && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    rvoff[k] = ok ? (unsigned)(((gy * a.w + gx) * a.Cin + q * 4) * 4) : 0x80000000u;
  }
  auto raw_dma = [&](int chunk, int buf, int k) {
    const int i = min(wv + 8 * k, kRawDma - 1);
    const unsigned lds_dst = lds_base + (unsigned)((buf * kRawSlots + i * 64) * 16);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                 : : "v"(rvoff[k]), "s"(lds_dst), "s"(rsrc), "s"(chunk * 64) : "memory", "m0");
  };

  // ---- position row pr of B^T d uses raw rows (ra, rb) of the 4x4 input tile: d0 - d2, d1 + d2, d2 - d1, d1 - d3
  const int ra = pr == 0 ? 0 : (pr == 2 ? 2 : 1);
  const int rb = pr == 0 ? 2 : (pr == 1 ? 2 : (pr == 2 ? 1 : 3));
  const float sgn = pr == 1 ? 1.f : -1.f;
  const int offA = ra * kRow, offB = rb * kRow;
  const int ty0 = m >> 3, tx = m & 7;
  const int src0 = (2 * kh) * kQuad + (2 * ty0) * kRow + tx;         // slot of (raw row 0 of the lane's tile, column 0, quad 2 kh)

  // t columns (ca, cb) of position column c: V = t[ca] + csgn t[cb]  (t0 - t2, t1 + t2, t2 - t1, t1 - t3)
  auto col_a = [](int c) { return c == 0 ? 0 : (c == 2 ? 2 : 1); };
  auto col_b = [](int c) { return c == 0 ? 2 : (c == 1 ? 2 : (c == 2 ? 1 : 3)); };
  auto read_t = [&](const float4* rbuf, int c, float (&tv)[2][8]) {
#pragma unroll
    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int col = cc == 0 ? col_a(c) : col_b(c);
        const int s = src0 + qq * kQuad + (col & 1) * kJ + (col >> 1);
        const float4 da = rbuf[s + offA], db = rbuf[s + offB];
        tv[cc][4 * qq + 0] = __builtin_fmaf(sgn, db.x, da.x);          // exact product: one rounding, = da +- db
        tv[cc][4 * qq + 1] = __builtin_fmaf(sgn, db.y, da.y);
        tv[cc][4 * qq + 2] = __builtin_fmaf(sgn, db.z, da.z);
        tv[cc][4 * qq + 3] = __builtin_fmaf(sgn, db.w, da.w);
      }
  };
  auto gen_pair = [&](float csgn, const float (&tv)[2][8], int e, u32x4 (&A)[3]) {
    const float v0 = __builtin_fmaf(csgn, tv[1][2 * e], tv[0][2 * e]);
    const float v1 = __builtin_fmaf(csgn, tv[1][2 * e + 1], tv[0][2 * e + 1]);
    unsigned p_hi, p_mid, p_lo;
    m4d_split3_pair(v0, v1, p_hi, p_mid, p_lo);                            // m4d_common.h
    A[0][e] = p_hi; A[1][e] = p_mid; A[2][e] = p_lo;
  };
  const int c0 = 2 * ch, c1 = 2 * ch + 1;
  const float csgn0 = c0 == 1 ? 1.f : -1.f, csgn1 = c1 == 1 ? 1.f : -1.f;

  // ---- B fragments: wu[chunk][CoutPad / 64][16 positions][2 N-tiles][3 parts][64 lanes][8 bf16]: the 6 KB of a position are
  // contiguous; ring slot s = position-of-the-wave * 6 + N-tile * 3 + part
  const long long w_pos = 6 * 1024;
  const long long w_chunk = (long long)n_groups * 16 * w_pos;
  const unsigned char* wbase = a.wu + ((long long)ng * 16 + 4 * pr + c0) * w_pos;   // chunk 0, first position of this wave
  const unsigned ring = lds_base + (unsigned)(kRingOff + wv * 12288);
  const unsigned char* ring_p = reinterpret_cast<const unsigned char*>(lds) + kRingOff + wv * 12288;
  const unsigned bl = (unsigned)lane * 16u;
  // all 6 fragments of position pl (0 / 1) of this wave for chunk `chunk` into slots 6 pl .. 6 pl + 5; their previous
  // fragments must be in registers (lgkmcnt(0): their ds_reads were issued one step earlier)
  auto b_dma = [&](int chunk, int pl) {
    const unsigned char* g = wbase + chunk * w_chunk + pl * w_pos;
    const unsigned dst = ring + (unsigned)(pl * 6144);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %2\n\tglobal_load_lds_dwordx4 %0, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %2 offset:2048"
                 : : "v"(bl), "s"(dst), "s"(g) : "memory", "m0");
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %2\n\tglobal_load_lds_dwordx4 %0, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %2 offset:2048"
                 : : "v"(bl), "s"(dst + 3072u), "s"(g + 3072) : "memory", "m0");
  };
  auto frag = [&](int pl, int nt, int part) {
    return *reinterpret_cast<const bf16x8*>(ring_p + (pl * 6 + nt * 3 + part) * 1024 + bl);
  };
#define M4D_W6H_WAIT(nn) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(nn) : "memory")

  f32x16 acc[2][2];                                // [position of the wave][N-tile]
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pl][nt][r] = 0.f;

  // ---- prologue: raw(0), raw(1), B(0) -- everything waited for; V(0) of the first position; its fragments
#pragma unroll
  for (int k = 0; k < 2; ++k) raw_dma(0, 0, k);
#pragma unroll
  for (int k = 0; k < 2; ++k) raw_dma(min(1, last), 1, k);
  b_dma(0, 0);
  b_dma(0, 1);
  M4D_W6H_WAIT(0);
  __builtin_amdgcn_s_barrier();
  u32x4 A[2][3];                                   // [position of the wave][part]: packed bf16 pairs
  bf16x8 B[2][2][3];                               // [set = position of the wave][N-tile][part]
  float tv0[2][8], tv1[2][8];
  read_t(raw, c0, tv0);
#pragma unroll
  for (int e = 0; e < 4; ++e) gen_pair(csgn0, tv0, e, A[0]);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int part = 0; part < 3; ++part) B[0][nt][part] = frag(0, nt, part);

  auto pin_a = [&](u32x4 (&X)[3]) {
#pragma unroll
    for (int part = 0; part < 3; ++part) asm volatile("" : "+v"(X[part]));
  };
  // 6 of the 9 term products, the small ones first, the two N-tiles interleaved (independent accumulators)
#define M4D_W6H_MFMA(pl, ap, bp)                                                                                       \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                                     \
    acc[pl][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[pl][ap]), B[pl][nt][bp], acc[pl][nt], 0, 0, 0);
#define M4D_W6H_MFMAS(pl) M4D_W6H_MFMA(pl, 0, 2) M4D_W6H_MFMA(pl, 2, 0) M4D_W6H_MFMA(pl, 1, 1) M4D_W6H_MFMA(pl, 0, 1) M4D_W6H_MFMA(pl, 1, 0) M4D_W6H_MFMA(pl, 0, 0)
#define M4D_W6H_PIPE(n_ds, n_valu)                                                                                     \
  __builtin_amdgcn_sched_group_barrier(0x100, n_ds, 0);                                                                \
  _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x002, n_valu, 0);                                                            \
  }                                                                                                                    \
  __builtin_amdgcn_sched_barrier(0);

  // One chunk: barrier (raw(k + 1) of every wave has landed; raw(k) is still valid: three buffers); step 0 = the 12 MFMAs of
  // the first position, beside them V(k) of the second position; step 1 = the second position, beside it V(k + 1) of the
  // first.  Each step: DMA of the next chunk's fragments into the slots whose fragments are in registers (+ in step 0 this
  // wave's two pieces of raw(k + 2)), then the other position's fragments from LDS.  14 DMAs per chunk:
  // [B x 6, raw x 2][B x 6] -- every fragment wait leaves 8 in flight, the barrier wait 6.
  for (int k = 0; k < n; ++k) {
    M4D_W6H_WAIT(6);
    __builtin_amdgcn_s_barrier();
    const int kn = min(k + 1, last);
    const float4* rcur = raw + (k % 3) * kRawSlots;
    const float4* rnext = raw + ((k + 1) % 3) * kRawSlots;
    // ---- step 0
    b_dma(kn, 0);
    raw_dma(min(k + 2, last), (k + 2) % 3, 0);
    raw_dma(min(k + 2, last), (k + 2) % 3, 1);
    M4D_W6H_WAIT(8);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int part = 0; part < 3; ++part) B[1][nt][part] = frag(1, nt, part);
    read_t(rcur, c1, tv1);
#pragma unroll
    for (int e = 0; e < 4; ++e) gen_pair(csgn1, tv1, e, A[1]);
    M4D_W6H_MFMAS(0)
    pin_a(A[1]);
    M4D_W6H_PIPE(14, 5)
    // ---- step 1
    b_dma(kn, 1);
    M4D_W6H_WAIT(8);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int part = 0; part < 3; ++part) B[0][nt][part] = frag(0, nt, part);
    read_t(rnext, c0, tv0);
#pragma unroll
    for (int e = 0; e < 4; ++e) gen_pair(csgn0, tv0, e, A[0]);
    M4D_W6H_MFMAS(1)
    pin_a(A[0]);
    M4D_W6H_PIPE(14, 5)
  }
#undef M4D_W6H_MFMA
#undef M4D_W6H_MFMAS
#undef M4D_W6H_PIPE
  M4D_W6H_WAIT(0);                                 // no DMA may land in LDS once the staging buffer reuses it
#undef M4D_W6H_WAIT
  __syncthreads();

  // ---- output transform: every wave stages its accumulators M[position][tile][cout]; one item = (N-tile, 2x2-output tile,
  // cout quad) per thread reads all 16 positions: rows of (M A) then A^T (M A), bias + leaky_relu, 16-byte stores
  float* St = lds;
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float* sb = St + ((nt * 16 + 4 * pr + 2 * ch + pl) * 32) * kMS;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int trow = (r & 3) + 8 * (r >> 2) + 4 * kh;
        sb[trow * kMS + m] = acc[pl][nt][r];
      }
    }
  __syncthreads();
  {
    const int cq = t & 7, tl = (t >> 3) & 31, ont = t >> 8;
    const int co = ng * 64 + ont * 32 + 4 * cq;
    float y[2][2][4];                              // [column k][row l][cout]
    float bs[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bs[e] = a.bias[min(co + e, a.Cout - 1)];
    float rr[4][2][4];                             // rows of (M A): [position row i][k][cout]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 mv[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) mv[cc] = *reinterpret_cast<const float4*>(St + ((ont * 16 + 4 * i + cc) * 32 + tl) * kMS + 4 * cq);
      const float* m0 = reinterpret_cast<const float*>(&mv[0]); const float* m1 = reinterpret_cast<const float*>(&mv[1]);
      const float* m2 = reinterpret_cast<const float*>(&mv[2]); const float* m3 = reinterpret_cast<const float*>(&mv[3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        rr[i][0][e] = (m0[e] + m1[e]) + m2[e];
        rr[i][1][e] = (m1[e] - m2[e]) - m3[e];
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v0 = ((rr[0][k][e] + rr[1][k][e]) + rr[2][k][e]) + bs[e];
        const float v1 = ((rr[1][k][e] - rr[2][k][e]) - rr[3][k][e]) + bs[e];
        y[k][0][e] = v0 > 0.f ? v0 : v0 * a.slope;
        y[k][1][e] = v1 > 0.f ? v1 : v1 * a.slope;
      }
    const int ox = tile_x + 2 * (tl & 7), oy = tile_y + 2 * (tl >> 3);
    float* op = a.out + (long long)bi * a.h * a.w * a.Cout + ((long long)oy * a.w + ox) * a.Cout + co;
    const bool vec_ok = (a.Cout & 3) == 0;
    if (co < a.Cout) {
#pragma unroll
      for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (ox + k < a.w && oy + l < a.h) {
            float* o2 = op + ((long long)l * a.w + k) * a.Cout;
            if (vec_ok && co + 3 < a.Cout) *reinterpret_cast<float4*>(o2) = make_float4(y[k][l][0], y[k][l][1], y[k][l][2], y[k][l][3]);
            else { for (int e = 0; e < 4; ++e) if (co + e < a.Cout) o2[e] = y[k][l][e]; }
          }
    }
  }
}

}  // namespace

// Launch for m4d_conv3x3_wino6_bias_act (m4d_wino6.hip decides when): any shape that kernel takes.
int m4d_wino6h_launch(const float* x, const void* wu6, const float* bias, int b, int h, int w, int Cin, int Cout, int CoutPad,
                      float slope, float* out, void* stream) {
  M4D_CHECK_ARG(CoutPad % 64 == 0 && CoutPad >= Cout && Cin % 16 == 0 && Cin >= 16);
  W6wArgs a;
  a.x = x; a.wu = reinterpret_cast<const unsigned char*>(wu6); a.bias = bias; a.out = out;
  a.b = b; a.h = h; a.w = w; a.Cin = Cin; a.Cout = Cout; a.CoutPad = CoutPad; a.n_chunks = Cin / 16; a.slope = slope;
  a.tiles_x = (w + kTW - 1) / kTW; a.tiles_y = (h + kTH - 1) / kTH;
  a.stamps = nullptr;
  constexpr size_t lds = (size_t)kStageFloats * sizeof(float) > (size_t)kRingOff + 8 * 12288 ? (size_t)kStageFloats * sizeof(float)
                                                                                         : (size_t)kRingOff + 8 * 12288;
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino6h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr_set;
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * (CoutPad / 64)), (unsigned)b);
  m4d_launch(conv3x3_wino6h_kernel, grid, dim3(512), lds, (hipStream_t)stream, a);
  return M4D_LAUNCH_RESULT();
}
