// The WIDE variant of the bf16-split Winograd convolution (m4d_wino6.hip): one workgroup = a 16x16-pixel tile x ALL output
// channels (96 or 128 = 3 / 4 MFMA N-tiles), in two passes over the position rows of the 4x4 Winograd transform.
//
// Why: m4d_wino6.hip's workgroup (16x16 pixels x 64 couts, all 16 positions) is VALU-issue bound -- every transformed input
// element V is split exactly into three bf16 terms (13 VALU instructions per element pair) and then meets only 2 N-tiles:
// 8.7 VALU instructions per MFMA (PMC, profiles/r02_wino6_pmc.txt), matrix cores 31 % busy.  The split work is per (tile,
// position, channel); the MFMA work is that x N-tiles.  All 4 N-tiles x 16 positions x 2 M-tiles of accumulators would be
// the whole register file of a CU (512 KB), so this kernel keeps HALF the positions at a time: pass p covers position rows
// 2p, 2p + 1 (8 positions x 2 M-tiles x NT N-tiles = 64 accumulator tiles = 8 per wave, as before), every V is used for NT = 3
// or 4 N-tiles, and the K loop runs twice (the raw halo is re-read by the second pass: L2 hits).  Per MFMA: ~3 VALU
// instructions instead of 8.7.
//
// Wave (prl, c) owns ONE position -- row 2p + prl, column c -- for both M-tiles and all N-tiles: 2 x NT accumulators.  Nothing
// but the raw halo is shared between waves: every wave streams its own B fragments (the host-split U, m4d_wino6's packed
// layout, 1 KB per (position, N-tile, part)) by LDS-DMA into a private ring of 3 NT slots, refilled slot by slot for the next
// chunk as soon as the fragment is in registers -- hand-counted s_waitcnt vmcnt(N), no barrier for B at all; ONE raw
// s_barrier per 16-channel chunk publishes the next raw halo (double buffered, LDS-DMA through a buffer descriptor as in
// m4d_wino6.hip).  V of the next chunk (read_t + split, both M-tiles) is produced between the MFMAs of the current one.
//
// The result is BIT-IDENTICAL to conv3x3_wino6_kernel: same products, same accumulation order per accumulator (chunk-major,
// a0 b2, a2 b0, a1 b1, a0 b1, a1 b0, a0 b0), same association in the output transform -- the column transform needs all four
// positions of a row (four waves: through the LDS staging), the row transform needs all four rows: pass 0 leaves
// s01 = R0 + R1 and R1 in the workgroup's own output pixels (scratch: nobody else touches them), pass 1 reads them back and
// finishes ((s01 + R2) + bias, ((R1 - R2) - R3) + bias, leaky_relu).  tests/test_gpu_ops.py compares the two kernels bit for bit.
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

// raw halo layout: identical to m4d_wino6.hip
constexpr int kT = 16, kH = kT + 2;
constexpr int kJ = 10;
constexpr int kRow = 2 * kJ;
constexpr int kQuad = kH * kRow;
constexpr int kRawUsed = 4 * kQuad;
constexpr int kRawDma = 23;
constexpr int kRawSlots = kRawDma * 64;
constexpr int kRingOff = 2 * kRawSlots * 16;     // byte offset of the per-wave B rings (47104)
constexpr int kMS = 36;                          // staging row stride (floats): 32 couts + 4 pad
constexpr int kStageFloats = 2 * 2 * 4 * 64 * kMS;   // [N-tile of the half][prl][c][tile][kMS] = 147456 B

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  bf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lo_f32(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f32(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

template <int N> using IC = std::integral_constant<int, N>;

// ABL: timing ablations (wrong results; tools/bench_wino6w.py --ablate): 1 = no V side work, 2 = no B DMA, 4 = no raw DMA,
// 8 = no barrier in the K loop, 16 = no B fragment reads, 32 = no MFMAs
template <int NT, int ABL = 0>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv3x3_wino6w_kernel(const W6wArgs a) {
  extern __shared__ __align__(16) float lds[];
  float4* raw = reinterpret_cast<float4*>(lds);                      // [2][kRawSlots]
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;

  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int prl = wv & 1, c = wv >> 1;                               // position row within the pass, position column
  const int m = lane & 31, kh = lane >> 5;
  const int n_tiles = a.tiles_x * a.tiles_y;
  int tile;
  {
    const int L = blockIdx.x;
    if ((n_tiles & 7) == 0) {                                        // consecutive workgroups go to different XCDs: give each XCD
      const int xcd = L & 7, idx = L >> 3;                           // a band of neighbouring tiles (its L2 serves the shared halos)
      tile = xcd * (n_tiles >> 3) + idx;
    } else {
      tile = L;
    }
  }
  const int tile_y = (tile / a.tiles_x) * kT, tile_x = (tile % a.tiles_x) * kT;
  const int bi = blockIdx.y;
  const int n = a.n_chunks, last = n - 1;
  const float* ximg = a.x + (long long)bi * a.h * a.w * a.Cin;

  // ---- raw halo by LDS-DMA (as in m4d_wino6.hip): instruction i fills slots 64 i .. 64 i + 63; wave wv issues i = wv, wv + 8,
  // wv + 16 (the 24th repeats the 23rd); pixels outside the image / pad slots read past num_records -> zeros
  i32x4 rsrc;
  {
    const unsigned long long xa = (unsigned long long)ximg;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(xa & 0xffffffffull));
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)((xa >> 32) & 0xffffull));
    rsrc[2] = a.h * a.w * a.Cin * 4;
    rsrc[3] = 0x00020000;
  }
  unsigned rvoff[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
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

  // ---- this wave's position column c: V = (t B)_c = t[ca] + csgn * t[cb]  (t0 - t2, t1 + t2, t2 - t1, t1 - t3)
  const int ca = c == 0 ? 0 : (c == 2 ? 2 : 1);
  const int cb = c == 0 ? 2 : (c == 1 ? 2 : (c == 2 ? 1 : 3));
  const float csgn = c == 1 ? 1.f : -1.f;
  const int ty0 = m >> 3, tx = m & 7;
  // slot of (raw row 0 of the lane's tile, tile column 0, quad 2 kh) for M-tile 0 / 1; + row * kRow, + column, + quad
  const int src_m0 = (2 * kh) * kQuad + (2 * ty0) * kRow + tx;
  const int colA = (ca & 1) * kJ + (ca >> 1), colB = (cb & 1) * kJ + (cb >> 1);

  float tv[2][8];                                  // t of the chunk being transformed: [ca / cb][channel]
  auto read_t = [&](const float4* rbuf, int mt2, int offA, int offB, float sgn) {
    const int s0 = src_m0 + mt2 * 8 * kRow;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int s = s0 + qq * kQuad + (cc == 0 ? colA : colB);
        const float4 da = rbuf[s + offA], db = rbuf[s + offB];
        tv[cc][4 * qq + 0] = __builtin_fmaf(sgn, db.x, da.x);          // exact product: one rounding, = da +- db
        tv[cc][4 * qq + 1] = __builtin_fmaf(sgn, db.y, da.y);
        tv[cc][4 * qq + 2] = __builtin_fmaf(sgn, db.z, da.z);
        tv[cc][4 * qq + 3] = __builtin_fmaf(sgn, db.w, da.w);
      }
  };
  // element pair e (channels 2e, 2e + 1) of V -> one packed word of each of the three bf16x8 operands (hi, mid, lo)
  auto gen_pair = [&](int e, u32x4 (&A)[3]) {
    const float v0 = __builtin_fmaf(csgn, tv[1][2 * e], tv[0][2 * e]);
    const float v1 = __builtin_fmaf(csgn, tv[1][2 * e + 1], tv[0][2 * e + 1]);
    unsigned p_hi, p_mid, p_lo;
    m4d_split3_pair(v0, v1, p_hi, p_mid, p_lo);                            // m4d_common.h
    A[0][e] = p_hi; A[1][e] = p_mid; A[2][e] = p_lo;
  };

  // ---- B fragments: wu[chunk][CoutPad / 64][16 positions][2 N-tiles][3 parts][64 lanes][8 bf16] (pack_conv_weights_wino6)
  const long long w_pos = 6 * 1024;
  const long long w_chunk = (long long)(a.CoutPad / 64) * 16 * w_pos;
  const unsigned ring = lds_base + (unsigned)(kRingOff + wv * (NT * 3072));
  const unsigned char* ring_p = reinterpret_cast<const unsigned char*>(lds) + kRingOff + wv * (NT * 3072);
  const unsigned bl = (unsigned)lane * 16u;
  // the three parts of N-tile nt (3 KB, contiguous on both sides) into slots 3 nt .. 3 nt + 2 of this wave's ring; the slots'
  // previous fragments must be in registers (lgkmcnt(0): their ds_reads were issued one step earlier)
  auto b_dma = [&](const unsigned char* gsrc, int nt) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %2\n\tglobal_load_lds_dwordx4 %0, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %2 offset:2048"
                 : : "v"(bl), "s"(ring + (unsigned)(nt * 3072)), "s"(gsrc) : "memory", "m0");
  };
  auto frag = [&](int nt, int part) {
    return *reinterpret_cast<const bf16x8*>(ring_p + (nt * 3 + part) * 1024 + bl);
  };
#define M4D_W6W_WAIT(nn) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(nn) : "memory")

  float* oimg = a.out + (long long)bi * a.h * a.w * a.Cout;
#ifdef M4D_W6W_ABLATIONS
  unsigned long long* st = (a.stamps != nullptr && lane == 0 && blockIdx.y == 0 && blockIdx.x < 64) ? a.stamps + ((long long)blockIdx.x * 8 + wv) * 16 : nullptr;
#define M4D_W6W_STAMP(i) if (st) st[i] = __builtin_readcyclecounter();
#else
#define M4D_W6W_STAMP(i)
#endif
  M4D_W6W_STAMP(0)

  for (int p = 0; p < 2; ++p) {
    // position row of this pass: B^T d uses raw rows (ra, rb) of the 4x4 input tile: d0 - d2, d1 + d2, d2 - d1, d1 - d3
    const int row = 2 * p + prl;
    const int ra = row == 0 ? 0 : (row == 2 ? 2 : 1);
    const int rb = row == 0 ? 2 : (row == 1 ? 2 : (row == 2 ? 1 : 3));
    const float sgn = row == 1 ? 1.f : -1.f;
    const int offA = ra * kRow, offB = rb * kRow;
    const unsigned char* wrow = a.wu + (long long)(4 * row + c) * w_pos;              // chunk 0, N-tile 0 of this position
    auto wfrag = [&](int chunk, int nt) { return wrow + chunk * w_chunk + (long long)(nt >> 1) * 16 * w_pos + (nt & 1) * 3072; };

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // ---- prologue: only raw(0) and the fragments of step 0 are waited for; raw(1) and the rest of B(0) land while V(0) is
    // produced.  Issue order raw(0), B(0)[0] | raw(1), B(0)[1..] -- the order of the steady state, so the K loop's counts hold
    // from its first iteration on (its barrier wait leaves 3 DMAs in flight: the last fragments of B(0)).
#pragma unroll
    for (int k = 0; k < 3; ++k) raw_dma(0, 0, k);
    b_dma(wfrag(0, 0), 0);
    M4D_W6W_STAMP(1 + 6 * p)
    M4D_W6W_WAIT(0);                               // (also: the scratch stores of pass 0 have completed)
    __builtin_amdgcn_s_barrier();
    M4D_W6W_STAMP(2 + 6 * p)
#pragma unroll
    for (int k = 0; k < 3; ++k) raw_dma(min(1, last), 1, k);
#pragma unroll
    for (int nt = 1; nt < NT; ++nt) b_dma(wfrag(0, nt), nt);
    u32x4 A[2][2][3];                              // [buffer][M-tile][part]: packed bf16 pairs
    bf16x8 B[2][3];                                // [set][part]
#pragma unroll
    for (int mt2 = 0; mt2 < 2; ++mt2) {
      read_t(raw, mt2, offA, offB, sgn);
#pragma unroll
      for (int e = 0; e < 4; ++e) gen_pair(e, A[0][mt2]);
    }
#pragma unroll
    for (int part = 0; part < 3; ++part) B[0][part] = frag(0, part);

    auto pin_a = [&](u32x4 (&X)[3], int e) {
#pragma unroll
      for (int part = 0; part < 3; ++part) asm volatile("" : "+v"(X[part][e]));
    };
    // One chunk: barrier (raw(k + 1) of every wave has landed); raw(k + 2) DMAs; then NT steps of 12 MFMAs (2 M-tiles x 6 term
    // products of N-tile j), each step: DMA of the NEXT chunk's fragments into the slots whose fragments are in registers,
    // the fragments of the next step from LDS, and a share of V(k + 1).
    // one step = the 12 MFMAs of N-tile j (2 M-tiles x 6 term products) + its share of the side work
    auto step = [&](auto PAR, auto J, int k, int kn, const float4* rnext) {
      constexpr int par = decltype(PAR)::value, j = decltype(J)::value;
      constexpr int cur = (par * NT + j) & 1, nxt = cur ^ 1;
      constexpr int jn = j == NT - 1 ? 0 : j + 1;
      u32x4 (&Ac)[2][3] = A[par];
      u32x4 (&An)[2][3] = A[par ^ 1];
      if constexpr (!(ABL & 2)) b_dma(wfrag(kn, j), j);
      // (one piece of raw(k + 2) per step: a burst of 24 right after the barrier would block every wave at once)
      // NT == 4: pieces in steps 0, 1, 2; NT == 3: two in step 0, one in step 1.  Either way 3 NT + 3 DMAs per chunk with the
      // last raw piece followed by 3 fragment DMAs, so: every fragment wait leaves 3 NT DMAs in flight, the barrier wait 3.
      if constexpr (!(ABL & 4)) {
        if constexpr (NT == 4) { if constexpr (j < 3) raw_dma(min(k + 2, last), k & 1, j); }
        else if constexpr (j == 0) { raw_dma(min(k + 2, last), k & 1, 0); raw_dma(min(k + 2, last), k & 1, 1); }
        else if constexpr (j == 1) raw_dma(min(k + 2, last), k & 1, 2);
      }
      if constexpr (!(ABL & 6)) M4D_W6W_WAIT(3 * NT);
      if constexpr (!(ABL & 16)) {
#pragma unroll
        for (int part = 0; part < 3; ++part) B[nxt][part] = frag(jn, part);
      }
      // V(k + 1): NT == 4: steps 0 / 2 read t of M-tile 0 / 1 and split pairs 0, 1; steps 1 / 3 split pairs 2, 3;
      // NT == 3: step 0: t(0), pairs 0-2; step 1: pair 3, t(1), pair 0; step 2: pairs 1-3
      constexpr int n_valu = NT == 4 ? ((j & 1) == 0 ? 4 : 3) : (j == 0 ? 5 : 4);
      if constexpr (ABL & 1) {
      } else if constexpr (NT == 4) {
        constexpr int mt2 = j >> 1;
        if constexpr ((j & 1) == 0) {
          read_t(rnext, mt2, offA, offB, sgn);
          gen_pair(0, An[mt2]); gen_pair(1, An[mt2]); pin_a(An[mt2], 0); pin_a(An[mt2], 1);
        } else {
          gen_pair(2, An[mt2]); gen_pair(3, An[mt2]); pin_a(An[mt2], 2); pin_a(An[mt2], 3);
        }
      } else {
        if constexpr (j == 0) {
          read_t(rnext, 0, offA, offB, sgn);
          gen_pair(0, An[0]); gen_pair(1, An[0]); gen_pair(2, An[0]); pin_a(An[0], 0); pin_a(An[0], 1); pin_a(An[0], 2);
        } else if constexpr (j == 1) {
          gen_pair(3, An[0]); pin_a(An[0], 3);
          read_t(rnext, 1, offA, offB, sgn);
          gen_pair(0, An[1]); pin_a(An[1], 0);
        } else {
          gen_pair(1, An[1]); gen_pair(2, An[1]); gen_pair(3, An[1]); pin_a(An[1], 1); pin_a(An[1], 2); pin_a(An[1], 3);
        }
      }
      // 6 of the 9 term products, the small ones first, the two M-tiles interleaved (independent accumulators)
#define M4D_W6W_MFMA(ap, bp)                                                                                           \
  _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                                     \
    acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ac[mt][ap]), B[cur][bp], acc[mt][j], 0, 0, 0);
      if constexpr (!(ABL & 32)) {
        M4D_W6W_MFMA(0, 2) M4D_W6W_MFMA(2, 0) M4D_W6W_MFMA(1, 1) M4D_W6W_MFMA(0, 1) M4D_W6W_MFMA(1, 0) M4D_W6W_MFMA(0, 0)
      } else {
#pragma unroll
        for (int part = 0; part < 3; ++part) asm volatile("" : : "v"(B[cur][part]), "v"(Ac[0][part]), "v"(Ac[1][part]));
      }
#undef M4D_W6W_MFMA
      // issue order inside the step: every LDS read first (the next step's fragments, t of the next chunk), then one MFMA
      // and n_valu vector instructions alternately
      if constexpr (ABL == 0) {
        constexpr bool reads_t = NT == 4 ? (j & 1) == 0 : j < 2;
        __builtin_amdgcn_sched_group_barrier(0x100, reads_t ? 11 : 3, 0);
#pragma unroll
        for (int i_ = 0; i_ < 12; ++i_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, n_valu, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // One chunk: barrier (raw(k + 1) of every wave has landed); raw(k + 2) DMAs; then NT steps, each: DMA of the NEXT chunk's
    // fragments into the slots whose fragments are in registers, the fragments of the next step from LDS, a share of V(k + 1).
    auto body = [&](auto PAR, int k) {
      if constexpr (!(ABL & 6)) M4D_W6W_WAIT(3);
      if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
      const int kn = min(k + 1, last);
      const float4* rnext = raw + ((k + 1) & 1) * kRawSlots;
      step(PAR, IC<0>{}, k, kn, rnext);
      step(PAR, IC<1>{}, k, kn, rnext);
      step(PAR, IC<2>{}, k, kn, rnext);
      if constexpr (NT == 4) step(PAR, IC<3>{}, k, kn, rnext);
    };
    M4D_W6W_STAMP(3 + 6 * p)
    for (int chunk = 0; chunk < n; chunk += 2) {
      body(IC<0>{}, chunk);
      if (chunk + 1 < n) body(IC<1>{}, chunk + 1);
    }
    M4D_W6W_STAMP(4 + 6 * p)
    M4D_W6W_WAIT(0);                               // no DMA may land in LDS once the staging buffer reuses it
    __syncthreads();
    M4D_W6W_STAMP(5 + 6 * p)

    // ---- output transform.  Per half (two N-tiles): every wave stages its accumulators M[row][c] (tile, cout); then one item
    // = (N-tile, 2x2-output tile, cout quad) per thread: column transform R[prl][k] from the four positions of a row, then
    // pass 0: s01 = R0 + R1 and R1 -> the item's own output pixels (scratch); pass 1: reads them back and finishes.
    float* St = lds;
    // (the item geometry is re-derived here from a laundered thread index: hoisted out of the pass loop it would stay live
    // across the K loop, whose register budget is full, and be spilled)
    int t_ep = t, tile_ep = tile;
    asm volatile("" : "+v"(t_ep), "+s"(tile_ep));
    const int cq = t_ep & 7, tl = t_ep >> 3;                          // this thread's items: cout quad, 2x2-output tile 0..63 (8 x 8)
    const int ox = (tile_ep % a.tiles_x) * kT + 2 * (tl & 7), oy = (tile_ep / a.tiles_x) * kT + 2 * (tl >> 3);
#pragma unroll
    for (int hf = 0; hf < (NT + 1) / 2; ++hf) {
      // pass 1: the scratch of this half's items early -- N-tile 0's loads return while the accumulators are staged, N-tile 1's
      // (issued after the barrier) while N-tile 0 is finished.  (All 16 loads up front would need 64 registers next to the
      // 128 accumulators: spills, measured slower.)
      float4 sc[2][2][2];                                             // [N-tile of the half][k][s01 / r1]
      auto load_scratch = [&](int ntl) {
        const int nt = 2 * hf + ntl, co = nt * 32 + 4 * cq;
        const float* op = oimg + ((long long)oy * a.w + ox) * a.Cout + co;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          // (row 1 of an odd-height image's last tile row: its scratch was never written; nor is its result)
          sc[ntl][k][0] = sc[ntl][k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (nt < NT && co < a.Cout && ox + k < a.w) {
            if (oy < a.h) sc[ntl][k][0] = *reinterpret_cast<const float4*>(op + (long long)k * a.Cout);
            if (oy + 1 < a.h) sc[ntl][k][1] = *reinterpret_cast<const float4*>(op + ((long long)a.w + k) * a.Cout);
          }
        }
      };
      if (p == 1) load_scratch(0);
#pragma unroll
      for (int ntl = 0; ntl < 2; ++ntl) {
        const int nt = 2 * hf + ntl;
        if (nt < NT) {
          float* sb = St + ((ntl * 2 + prl) * 4 + c) * (64 * kMS);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int trow = (r & 3) + 8 * (r >> 2) + 4 * kh;
              sb[(mt * 32 + trow) * kMS + m] = acc[mt][nt < NT ? nt : 0][r];
            }
        }
      }
      __syncthreads();
      if (p == 1) load_scratch(1);
#pragma unroll
      for (int ntl = 0; ntl < 2; ++ntl) {
        const int nt = 2 * hf + ntl;
        const int co = nt * 32 + 4 * cq;
        if (nt < NT && co < a.Cout) {
          float R[2][2][4];                                           // [row of the pass][k][cout]
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            float4 mv[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
              mv[cc] = *reinterpret_cast<const float4*>(St + (((ntl * 2 + rr) * 4 + cc) * 64 + tl) * kMS + 4 * cq);
            const float* m0 = reinterpret_cast<const float*>(&mv[0]); const float* m1 = reinterpret_cast<const float*>(&mv[1]);
            const float* m2 = reinterpret_cast<const float*>(&mv[2]); const float* m3 = reinterpret_cast<const float*>(&mv[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              R[rr][0][e] = (m0[e] + m1[e]) + m2[e];
              R[rr][1][e] = (m1[e] - m2[e]) - m3[e];
            }
          }
          float* op = oimg + ((long long)oy * a.w + ox) * a.Cout + co;
          if (p == 0) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
              if (ox + k < a.w) {
                if (oy < a.h)
                  *reinterpret_cast<float4*>(op + (long long)k * a.Cout) =
                      make_float4(R[0][k][0] + R[1][k][0], R[0][k][1] + R[1][k][1], R[0][k][2] + R[1][k][2], R[0][k][3] + R[1][k][3]);
                if (oy + 1 < a.h)
                  *reinterpret_cast<float4*>(op + ((long long)a.w + k) * a.Cout) = make_float4(R[1][k][0], R[1][k][1], R[1][k][2], R[1][k][3]);
              }
          } else {
            const float4 bs = *reinterpret_cast<const float4*>(a.bias + co);
            const float* bsp = reinterpret_cast<const float*>(&bs);
#pragma unroll
            for (int k = 0; k < 2; ++k)
              if (ox + k < a.w) {
                const float* s01p = reinterpret_cast<const float*>(&sc[ntl][k][0]); const float* r1p = reinterpret_cast<const float*>(&sc[ntl][k][1]);
                float y0[4], y1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float v0 = (s01p[e] + R[0][k][e]) + bsp[e];
                  const float v1 = ((r1p[e] - R[0][k][e]) - R[1][k][e]) + bsp[e];
                  y0[e] = v0 > 0.f ? v0 : v0 * a.slope;
                  y1[e] = v1 > 0.f ? v1 : v1 * a.slope;
                }
                if (oy < a.h) *reinterpret_cast<float4*>(op + (long long)k * a.Cout) = make_float4(y0[0], y0[1], y0[2], y0[3]);
                if (oy + 1 < a.h) *reinterpret_cast<float4*>(op + ((long long)a.w + k) * a.Cout) = make_float4(y1[0], y1[1], y1[2], y1[3]);
              }
          }
        }
      }
      __syncthreads();                             // the staging buffer is rewritten by the next half / the next pass's DMAs
    }
    M4D_W6W_STAMP(6 + 6 * p)
  }
#undef M4D_W6W_WAIT
}

}  // namespace

// Launch for m4d_conv3x3_wino6_bias_act (m4d_wino6.hip decides when): CoutPad == 128, 64 < Cout <= 128, Cout % 4 == 0.
// profiling builds only (make W6FLAGS=-DM4D_W6W_ABLATIONS; tools/w6w_ablate.py): timing ablations and phase stamps
static unsigned long long* g_wino6w_stamps = nullptr;
#ifdef M4D_W6W_ABLATIONS
static int g_wino6w_ablate = 0;
extern "C" void m4d_wino6w_set_ablation(int mask) { g_wino6w_ablate = mask; }
extern "C" void m4d_wino6w_set_stamps(unsigned long long* device_buffer) { g_wino6w_stamps = device_buffer; }
#endif

int m4d_wino6w_launch(const float* x, const void* wu6, const float* bias, int b, int h, int w, int Cin, int Cout, int CoutPad,
                      float slope, float* out, void* stream) {
  M4D_CHECK_ARG(CoutPad == 128 && Cout > 64 && Cout <= 128 && (Cout & 3) == 0 && Cin % 16 == 0 && Cin >= 16);
  M4D_CHECK_ARG(((((uintptr_t)bias) & 15u) == 0) && ((((uintptr_t)out) & 15u) == 0));
  W6wArgs a;
  a.x = x; a.wu = reinterpret_cast<const unsigned char*>(wu6); a.bias = bias; a.out = out;
  a.b = b; a.h = h; a.w = w; a.Cin = Cin; a.Cout = Cout; a.CoutPad = CoutPad; a.n_chunks = Cin / 16; a.slope = slope;
  a.tiles_x = (w + kT - 1) / kT; a.tiles_y = (h + kT - 1) / kT;
  a.stamps = g_wino6w_stamps;
  const int NT = (Cout + 31) / 32;
  constexpr size_t lds4 = (size_t)kStageFloats * sizeof(float) > (size_t)kRingOff + 8 * 4 * 3072 ? (size_t)kStageFloats * sizeof(float)
                                                                                              : (size_t)kRingOff + 8 * 4 * 3072;
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino6w_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino6w_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr_set;
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y), (unsigned)b);
#ifdef M4D_W6W_ABLATIONS
  if (NT == 4 && g_wino6w_ablate) {
#define M4D_ABL(mask) case mask: { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino6w_kernel<4, mask>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      m4d_launch((conv3x3_wino6w_kernel<4, mask>), grid, dim3(512), lds4, (hipStream_t)stream, a); return M4D_LAUNCH_RESULT(); }
    switch (g_wino6w_ablate) { M4D_ABL(1) M4D_ABL(2) M4D_ABL(4) M4D_ABL(6) M4D_ABL(8) M4D_ABL(16) M4D_ABL(32) M4D_ABL(17) M4D_ABL(23) M4D_ABL(31) M4D_ABL(33) M4D_ABL(38) default: break; }
#undef M4D_ABL
  }
#endif
  if (NT == 3) m4d_launch(conv3x3_wino6w_kernel<3>, grid, dim3(512), lds4, (hipStream_t)stream, a);
  else m4d_launch(conv3x3_wino6w_kernel<4>, grid, dim3(512), lds4, (hipStream_t)stream, a);
  return M4D_LAUNCH_RESULT();
}
