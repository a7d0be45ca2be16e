// The bf16-split Winograd convolution of m4d_wino6.hip (same arithmetic, same bits) with PERSISTENT workgroups.
//
// m4d_wino6.hip launches one workgroup per (16x16-pixel tile, 64 output channels): every workgroup pays a prologue (per-lane
// halo addressing with integer divisions, ~430 VALU instructions; a DMA round trip from an idle memory pipe before the first
// product) and an epilogue that nothing overlaps, and the two 64-cout groups of a tile fetch the tile's halo twice, from
// wherever the dispatcher put them -- its ablations (DESIGN_HISTORY.md, round 3) put prologue + epilogue at ~13 % of a launch.
// Here a workgroup (one per CU: 154 KB of LDS) walks a contiguous range of (tile, cout group) units, tile-major, so the groups
// of a tile run back to back on one CU, and the K loop's DMA stream simply CONTINUES across the unit boundary:
//
//   * the raw halo of the next unit's chunks 0 and 1 is fetched by the last two chunks of the current unit (the raw DMA of
//     chunk c fetches chunk c + 2 of the unified stream), the fragments of its positions 0 and 1 by positions 0 and 1 of the
//     last chunk -- into the same buffers / slots the steady state would use; positions 2 and 3 of the last chunk issue no
//     fragment DMA: ring slots 2 and 3 are then free, adjacent in LDS ([slot 0][slot 1][raw A][slot 2][slot 3][raw B]), and
//     take the epilogue's staging buffer, one (output column, M-tile) = 34 KB at a time, four passes;
//   * after the epilogue the fragments of positions 2 and 3 are issued and the next unit starts with everything else landed:
//     no DMA latency, no address set-up beyond a handful of adds (the per-lane halo slot decode is done once per workgroup).
//
// hand-counted waits (every DMA is inline asm, 1 VM op per raw piece, 3 per fragment fetch; loads retire in order; vmcnt counts
// the epilogue's global stores too, and a store / load in between can only make a vmcnt(N) stricter):
//   steady state, per position: 1 raw piece + 3 fragment pieces; position 0 ends with vmcnt(7), positions 1-3 with vmcnt(8) =
//   everything but this position's and (all but the oldest piece of) the previous position's DMAs has landed.
//   entry of a unit (chunk 0, position 0), in flight oldest first:
//     first unit (prologue, closed by vmcnt(3)):   B(3)[3]
//     later units: ... stores of passes 0-2 | B(2)[3] B(3)[3] | the last pass's 2 stores
//       end of position 0 (+4, vmcnt 7): with every store acknowledged the 7 newest loads are its own 4 + B(3)[3]: B(2) has
//                                        landed -> position 1 reads slot 2 (stores still in flight only make it stricter)
//       end of position 1 (+4, vmcnt 8): leaves positions 0 / 1's DMAs                        -> position 2 reads slot 3
//     (the raw halo of chunks 0 / 1 and slots 0 / 1 landed during the previous unit's last chunk and were published)
//   last chunk: position 0 also fetches the unit's 64 biases (1 op), positions 2 / 3 issue their raw piece only; vmcnt(5) /
//   vmcnt(2) close them (slot 0 / slot 1 of the next unit landed), then the epilogue's barriers publish them.
// Deterministic: a unit's arithmetic does not depend on which workgroup runs it; bit-identical to conv3x3_wino6_kernel
// (tests/test_gpu_ops.py::test_persistent_winograd_is_bitwise_the_one_tile_kernel).
#include <cstdlib>
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

#ifndef M4D_W6P_ABL
#define M4D_W6P_ABL 0      // timing ablations (wrong results; tools/w6p_ablate.sh): 1 no epilogue passes at all, 2 no global stores,
#endif                     // 8 no DMA waits in the first two positions of a unit (what the epilogue's stores cost through vmcnt)

namespace {

typedef float p6_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 p6_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned p6_u32x4 __attribute__((ext_vector_type(4)));
typedef int p6_i32x4 __attribute__((ext_vector_type(4)));

struct Wino6PArgs {
  const float* x; const unsigned char* wu; const float* bias; float* out;
  int b, h, w, Cin, Cout, CoutPad, n_chunks, tiles_x, tiles_y, units, team;
  float slope;
  int stagger, phases;          // > 0: the first 256 workgroups start up to `stagger` 100-MHz ticks apart, in `phases` groups (m4d_wino6.hip)
};

constexpr int pT = 16, pH = pT + 2;              // output tile, halo (pixels)
// raw halo in LDS exactly as in m4d_wino6.hip: [row][column parity][column / 2][4 channel quads + 1 pad slot] of 16-byte slots
constexpr int pJ = 10, pPix = 5, pRow = 2 * pJ * pPix;          // 100 slots per halo row
constexpr int pRawUsed = pH * pRow;              // 1800 slots per 16-channel chunk
constexpr int pRawDma = 29, pRawK = 4;           // 64-slot DMA pieces per chunk; up to 4 per wave
constexpr int pRawBytes = pRawDma * 64 * 16;     // 29696
constexpr int pSlotBytes = 4 * 6144;             // one ring slot = one position of every position row: [row][6 fragments of 1 KB]
// LDS map: [slot 0][slot 1][raw A][slot 2][slot 3][raw B]
constexpr int pOffSlot01 = 0;
constexpr int pOffRawA = 2 * pSlotBytes;                         // 49152
constexpr int pOffSlot23 = pOffRawA + pRawBytes;                 // 78848
constexpr int pOffRawB = pOffSlot23 + 2 * pSlotBytes;            // 128000
constexpr int pOffBias = pOffRawB + pRawBytes;                   // 157696: the unit's 64 biases (LDS-DMA'd by the last chunk)
constexpr int pLds = pOffBias + 256;
constexpr int pMS2 = 68;                         // epilogue row stride (floats): 64 couts + 4 pad (a tile's 16 b128 lanes: 16 distinct slots)
constexpr int pEpiFloats = 4 * 32 * pMS2;        // one (output column, M-tile): [4 rows i][32 tiles][pMS2] = 34816 B
static_assert(pEpiFloats * 4 <= 2 * pSlotBytes, "the epilogue pass buffer lives in ring slots 2-3");
static_assert(pLds <= 160 * 1024, "LDS budget of one CU");

__host__ __device__ constexpr int p_slot_off(int slot) { return slot < 2 ? pOffSlot01 + slot * pSlotBytes : pOffSlot23 + (slot - 2) * pSlotBytes; }

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv3x3_wino6p_kernel(const Wino6PArgs a) {
  extern __shared__ __align__(16) float lds[];
  unsigned char* const ldsb = reinterpret_cast<unsigned char*>(lds);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;   // LDS byte address (for M0)
  if (a.stagger > 0 && blockIdx.x < 256) {
    // the staggered first round of m4d_wino6.hip (short ranges at batch 1-4: the workgroups of a round would otherwise free their
    // CUs in lock step): a bounded delay in front of the kernel body, same bits
    const unsigned phase = (blockIdx.x >> 3) & (unsigned)(a.phases - 1);
    const unsigned long long t_end = wall_clock64() + (unsigned long long)(phase * (unsigned)a.stagger) / (unsigned)a.phases;
    for (int spin = 0; spin < 512 && wall_clock64() < t_end; ++spin) __builtin_amdgcn_s_sleep(16);
  }

  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int pr = wv & 3, mt = wv >> 2;                               // position row, M-tile (waves r and r + 4 share a SIMD)
  const int m = lane & 31, kh = lane >> 5;
  const int n_tiles = a.tiles_x * a.tiles_y, n_groups = a.CoutPad / 64;
  const int n = a.n_chunks;

  // ---- this workgroup's units.  TEAM mode (large grids): the n_groups workgroups of a team sit on ONE XCD (workgroup b runs
  // on XCD b % 8) and walk the SAME run of pixel tiles, each for its own 64-cout group, in step -- a tile's halo chunk is
  // fetched into that XCD's L2 once and read by the team's members at about the same time, and a workgroup's consecutive units
  // reuse one group's weights; the XCD's teams share a contiguous band of the tile list.  (Round 4, first form: one workgroup
  // ran a tile's groups back to back -- by then the 4-MB L2 had seen 7 MB of the other workgroups' halos: FETCH_SIZE 1.58x
  // the one-unit kernel's at batch 32.)  Otherwise: a contiguous range of the tile-major (tile, group) list.
  struct Unit { int bi, tile_y, tile_x, ng; };
  // HALF unit: only N-tile 0 of its cout group holds real output channels (Cout = 96: the second group).  Its K loop issues N-tile
  // 0's MFMAs only, its epilogue handles N-tile 0 only; the DMA protocol (pieces per wave and position, hence every vmcnt count)
  // is the same, but the waves that fetch N-tile 1's fragments (mt = 1) point all 64 lanes at one 16-byte word: one cache line per
  // piece instead of sixteen, into ring bytes a half unit never reads.
  const int last_group_couts = a.Cout - (n_groups - 1) * 64;
  auto is_half = [&](int ng) { return __builtin_amdgcn_readfirstlane((ng == n_groups - 1 && last_group_couts <= 32) ? 1 : 0) != 0; };
  const bool alternate = last_group_couts <= 32 && n_groups > 1;
  int i0, i1, member = -1;                         // unit indices [i0, i1): tiles of the team (team mode) or (tile, group) units
  if (a.team) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int teams_per_xcd = per_xcd / n_groups;
    member = k % n_groups;
    const int team_in_xcd = k / n_groups;
    if (team_in_xcd >= teams_per_xcd) return;      // (per_xcd not a multiple of n_groups: the leftover workgroups idle)
    const int n_teams = 8 * teams_per_xcd, team = xcd * teams_per_xcd + team_in_xcd;
    const int n_bt = a.units / n_groups;           // pixel tiles of the whole batch
    i0 = (int)((long long)team * n_bt / n_teams);
    i1 = (int)((long long)(team + 1) * n_bt / n_teams);
  } else {
    i0 = (int)((long long)blockIdx.x * a.units / gridDim.x);
    i1 = (int)((long long)(blockIdx.x + 1) * a.units / gridDim.x);
  }
  if (i0 >= i1) return;
  const int u0 = i0, u1 = i1;
  auto decode = [&](int u) {
    Unit r;
    const int bt = member >= 0 ? u : u / n_groups;
    // (team mode with a half last group -- Cout = 96 --: the members swap groups tile by tile, or the member with the half
    // units would run ahead of its team and finish a quarter earlier)
    r.ng = member >= 0 ? (alternate ? (member + bt) % n_groups : member) : u - bt * n_groups;
    r.bi = bt / n_tiles;
    const int tile = bt - r.bi * n_tiles;
    r.tile_y = (tile / a.tiles_x) * pT;
    r.tile_x = (tile % a.tiles_x) * pT;
    return r;
  };

  // ---- raw halo by LDS-DMA: piece i fills slots 64 i .. 64 i + 63 (lane = slot); wave wv issues pieces wv, wv + 8, wv + 16,
  // wv + 24 (29 pieces: waves 5-7 repeat the 29th).  The source of the DMAs: buffer descriptor of the unit's image + per-lane
  // byte offsets (past num_records for pixels outside the image and pad slots: the range check returns zeros).  The slot
  // decode (three divisions by constants) is redone per unit from an opaque copy of the lane index: kept live across the K loop
  // it would cost four registers the loop does not have (spilled, and a scratch reload drains every DMA in flight).
  p6_i32x4 rsrc;
  unsigned rvoff[pRawK];
  auto set_raw_source = [&](const Unit& un) {
    const unsigned long long xa = (unsigned long long)(a.x + (long long)un.bi * a.h * a.w * a.Cin);
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(xa & 0xffffffffull));
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)((xa >> 32) & 0xffffull));             // stride 0: raw buffer
    rsrc[2] = a.h * a.w * a.Cin * 4;                                                      // num_records (bytes)
    rsrc[3] = 0x00020000;
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int k = 0; k < pRawK; ++k) {
      const int i = min(wv + 8 * k, pRawDma - 1);
      const int s = i * 64 + ln;
      const int pix = s / pPix, q = s - pix * pPix;                  // q = 4: the pad slot
      const int hy = pix / (2 * pJ), r2 = pix - hy * (2 * pJ);
      const int e = r2 / pJ, j = r2 - e * pJ;
      const int hx = 2 * j + e;
      const int gy = un.tile_y - 1 + hy, gx = un.tile_x - 1 + hx;
      const bool ok = s < pRawUsed && q < 4 && j < 9 && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      rvoff[k] = ok ? (unsigned)(((gy * a.w + gx) * a.Cin + q * 4) * 4) : 0x80000000u;
    }
  };
  // piece k of a 16-channel chunk (byte offset chunk_off within a pixel's channel run) into raw buffer `buf`
  auto raw_dma = [&](int chunk_off, int buf, int k) {
    const int i = min(wv + 8 * k, pRawDma - 1);
    const unsigned lds_dst = lds_base + (unsigned)(pOffRawA + buf * (pOffRawB - pOffRawA) + i * 1024);
    // (every DMA statement sets M0 itself and declares it clobbered: nothing else in this kernel uses M0)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                 : : "v"(rvoff[k]), "s"(lds_dst), "s"(rsrc), "s"(chunk_off) : "memory", "m0");
  };
  auto raw_ptr = [&](int buf) { return reinterpret_cast<const float4*>(ldsb + pOffRawA + buf * (pOffRawB - pOffRawA)); };

  // ---- this lane's tile: M-tile mt holds tile rows 4 mt .. 4 mt + 3; lane m = (row m >> 3, column m & 7)
  // position row pr of B^T d uses raw rows (ra, rb) of the 4x4 input tile: d0 - d2, d1 + d2, d2 - d1, d1 - d3
  const int ra = pr == 0 ? 0 : (pr == 2 ? 2 : 1);
  const int rb_ = pr == 0 ? 2 : (pr == 1 ? 2 : (pr == 2 ? 1 : 3));
  const float sgn = pr == 1 ? 1.f : -1.f;
  const int ty0 = m >> 3, tx = m & 7;
  const int src0 = (8 * mt + 2 * ty0) * pRow + tx * pPix + 2 * kh;

  float tv[4][8];                                  // t_c of the current chunk: [column c][channel]
  auto read_t = [&](const float4* rbuf, int c0) {
#pragma unroll
    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int c = c0 + 2 * cc;
        const int s = src0 + qq + ((c & 1) * pJ + (c >> 1)) * pPix;
        const float4 da = rbuf[s + ra * pRow], db = rbuf[s + rb_ * pRow];
        tv[c][4 * qq + 0] = __builtin_fmaf(sgn, db.x, da.x);           // exact product: one rounding, = da +- db
        tv[c][4 * qq + 1] = __builtin_fmaf(sgn, db.y, da.y);
        tv[c][4 * qq + 2] = __builtin_fmaf(sgn, db.z, da.z);
        tv[c][4 * qq + 3] = __builtin_fmaf(sgn, db.w, da.w);
      }
  };
  auto gen_pair = [&](int c, int e, p6_u32x4 (&A)[3]) {
    float v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ch = 2 * e + h;
      v[h] = c == 0 ? tv[0][ch] - tv[2][ch] : c == 1 ? tv[1][ch] + tv[2][ch] : c == 2 ? tv[2][ch] - tv[1][ch] : tv[1][ch] - tv[3][ch];
    }
    unsigned p_hi, p_mid, p_lo;
    m4d_split3_pair(v[0], v[1], p_hi, p_mid, p_lo);
    A[0][e] = p_hi; A[1][e] = p_mid; A[2][e] = p_lo;
  };

  // ---- B operands: wu[chunk][N-group][16 positions][2 N-tiles][3 parts][64 lanes][8 bf16]; position p of a row -> ring slot p,
  // fetched one chunk ahead -- fragments 3 mt .. 3 mt + 2 by wave (pr, mt)
  const long long w_pos = 6 * 1024;
  const long long w_chunk = (long long)n_groups * 16 * w_pos;
  const unsigned bl = (unsigned)lane * 16u;
  // per-lane GLOBAL offset of a fragment DMA (the LDS side is M0 + 16 lane whatever this is): the lane's 16 bytes of the wave's
  // three fragments -- or, for N-tile 1's fragments of a half unit, one word for everybody (see above)
  auto b_lane_off = [&](bool half_unit) { return (half_unit && mt == 1) ? 0u : bl + (unsigned)mt * 3072u; };
  auto b_dma = [&](const unsigned char* gsrc, int slot, unsigned voff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %2\n\tglobal_load_lds_dwordx4 %0, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %2 offset:2048"
                 : : "v"(voff), "s"(lds_base + (unsigned)(p_slot_off(slot) + pr * 6144) + (unsigned)mt * 3072u), "s"(gsrc) : "memory", "m0");
  };
  // The unit's 64 biases by LDS-DMA too (lane = output channel of the group, clamped to the last real one): a compiler-issued
  // global load in the unit loop would be waited for with a vmcnt the compiler counts WITHOUT the asm DMAs, i.e. a full drain
  // of everything in flight.  Every wave issues it (the same 256 bytes to the same place): the waits count alike in all waves.
  auto bias_dma = [&](int ng) {
    const unsigned voff = (unsigned)min(ng * 64 + lane, a.Cout - 1) * 4u;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, %2"
                 : : "v"(voff), "s"(lds_base + (unsigned)pOffBias), "s"(a.bias) : "memory", "m0");
  };
  p6_bf16x8 B0[2][2], B1[2], B2[2];                // parts (hi, mid, lo) of the B operand per N-tile; the hi part double buffered
  auto frag = [&](int slot, int nt, int part) {
    return *reinterpret_cast<const p6_bf16x8*>(ldsb + p_slot_off(slot) + pr * 6144 + (nt * 3 + part) * 1024 + bl);
  };
#define P6_WAIT(nn) asm volatile("s_waitcnt vmcnt(" #nn ") lgkmcnt(0)" ::: "memory")
  // workgroup barrier for LDS traffic only: no vmcnt(0) (a __syncthreads() would also drain the DMAs and stores in flight)
#define P6_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

  p6_f32x16 acc[4][2];

  p6_u32x4 A[2][3];                                // [ring][part]: packed bf16 pairs
  auto pin_a = [&](p6_u32x4 (&X)[3]) {
#pragma unroll
    for (int part = 0; part < 3; ++part) asm volatile("" : "+v"(X[part]));
  };
  auto pin_t = [&](int c0) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(tv[c0 + 2 * cc][e]));
  };
  // (NTL = the unit's N-tiles, a compile-time constant of the two instantiations of a unit's K loop: 2, or 1 for a half unit)
#define P6_MFMA(c, ap, bv)                                                                                             \
  _Pragma("unroll") for (int nt = 0; nt < NTL; ++nt)                                                                   \
    acc[c][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(p6_bf16x8, A[(c) & 1][ap]), bv[nt], acc[c][nt], 0, 0, 0);
  // the first product into an accumulator: in the unit's first chunk (FIRST) on top of the constant 0 instead of a zeroed register
#define P6_MFMA0(c, ap, bv)                                                                                            \
  _Pragma("unroll") for (int nt = 0; nt < NTL; ++nt)                                                                   \
    acc[c][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(p6_bf16x8, A[(c) & 1][ap]), bv[nt],         \
                                                         FIRST ? p6_f32x16{} : acc[c][nt], 0, 0, 0);
  // a block has 2 NTL MFMAs and the same vector work either way
#define P6_PIPE(valu)                                                                                                  \
  _Pragma("unroll") for (int i_ = 0; i_ < 2 * NTL; ++i_) {                                                             \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x002, (valu) * (2 / NTL), 0);                                                \
  }
  // One position = three blocks of 2 NTL MFMAs (6 of the 9 term products, the small ones first); NEXT = prefetch the B registers
  // (from ring slot next_slot) and generate the A operands (position cn) of the next position between the MFMAs
#define P6_BLOCK0(c, cn, next_slot, valu, NEXT)                                                                        \
  if (NEXT) { B0[((c) & 1) ^ 1][0] = frag(next_slot, 0, 0); if (NTL == 2) B0[((c) & 1) ^ 1][1] = frag(next_slot, 1, 0); \
              gen_pair(cn, 0, A[((c) & 1) ^ 1]); gen_pair(cn, 1, A[((c) & 1) ^ 1]); }                                  \
  P6_MFMA0(c, 0, B2) P6_MFMA(c, 2, B0[(c) & 1])                                                                        \
  if (NEXT) { asm volatile("" : "+v"(A[((c) & 1) ^ 1][0][0]), "+v"(A[((c) & 1) ^ 1][0][1])); P6_PIPE(valu) }            \
  __builtin_amdgcn_sched_barrier(0);
#define P6_BLOCK1(c, cn, next_slot, valu, NEXT)                                                                        \
  if (NEXT) { B2[0] = frag(next_slot, 0, 2); if (NTL == 2) B2[1] = frag(next_slot, 1, 2); gen_pair(cn, 2, A[((c) & 1) ^ 1]); } \
  P6_MFMA(c, 1, B1) P6_MFMA(c, 0, B1)                                                                                  \
  if (NEXT) { asm volatile("" : "+v"(A[((c) & 1) ^ 1][0][2])); P6_PIPE(valu) }                                         \
  __builtin_amdgcn_sched_barrier(0);
#define P6_BLOCK2(c, cn, next_slot, valu, NEXT)                                                                        \
  if (NEXT) { B1[0] = frag(next_slot, 0, 1); if (NTL == 2) B1[1] = frag(next_slot, 1, 1); gen_pair(cn, 3, A[((c) & 1) ^ 1]); } \
  P6_MFMA(c, 1, B0[(c) & 1]) P6_MFMA(c, 0, B0[(c) & 1])                                                                \
  if (NEXT) { pin_a(A[((c) & 1) ^ 1]); P6_PIPE(valu) }                                                                 \
  __builtin_amdgcn_sched_barrier(0);

  if (M4D_W6P_ABL & 16) {                           // experiment: de-phase the CUs (their epilogue store bursts coincide otherwise)
    for (int i = 0; i < (int)(blockIdx.x & 7); ++i) __builtin_amdgcn_s_sleep(127);
  }
  // ---- true prologue (once per workgroup): raw(0), B(0), B(1), raw(1) | B(2), B(3) -- the state every later unit starts from
  Unit cur = decode(u0);
  set_raw_source(cur);
  const unsigned char* wc = a.wu + ((long long)cur.ng * 16 + 4 * pr) * w_pos;     // uniform: this row, chunk 0 of the unit
  int par = 0;                                     // raw buffer of the unit's chunk 0
#pragma unroll
  for (int k = 0; k < pRawK; ++k) raw_dma(0, 0, k);
  {
    const unsigned voff = b_lane_off(is_half(cur.ng));
    b_dma(wc, 0, voff);
    b_dma(wc + w_pos, 1, voff);
#pragma unroll
    for (int k = 0; k < pRawK; ++k) raw_dma(64, 1, k);                            // (n >= 2: the host checks)
    b_dma(wc + 2 * w_pos, 2, voff);
    b_dma(wc + 3 * w_pos, 3, voff);
  }
  P6_WAIT(3);                                      // only B(3) still flies (see the wait table above)
  __builtin_amdgcn_s_barrier();

  for (int u = u0; u < u1; ++u) {
    const bool has_next = u + 1 < u1;
    const Unit nxt = has_next ? decode(u + 1) : cur;                 // (the last unit's surplus DMAs re-fetch its own first chunks)
    const unsigned char* wnext = a.wu + ((long long)nxt.ng * 16 + 4 * pr) * w_pos;
    const bool half = is_half(cur.ng);
    const unsigned voff_cur = b_lane_off(half), voff_nxt = b_lane_off(is_half(nxt.ng));

    // The unit's K loop and epilogue, two instantiations.  hipcc lays them out one after the other behind scalar guards: what
    // one of them needs must not be produced before the branch, or it stays live across the other's loop (hence the empty asm),
    // and N-tile 1's accumulators must not be mentioned in the half unit's epilogue under a run-time condition (they would stay
    // live -- 64 registers -- through its K loop).
    auto unit_body = [&](auto ntl_tag) __attribute__((always_inline)) {
    constexpr int NTL = decltype(ntl_tag)::value;
    asm volatile("" ::: "memory");
    // ---- t(0), A(0, 0), B(0) in registers: raw chunk 0 and ring slot 0 have landed and were published by a barrier
    read_t(raw_ptr(par), 0);
    read_t(raw_ptr(par), 1);
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) { B0[0][nt] = frag(0, nt, 0); B1[nt] = frag(0, nt, 1); B2[nt] = frag(0, nt, 2); }
    // (the accumulators are not zeroed: the unit's first chunk, peeled below, multiplies its first products onto the constant 0.
    // Zeros carried over from the previous unit's epilogue -- round 4's first form -- are 128 registers live through BOTH
    // instantiations as hipcc lays them out, one after the other behind scalar guards: spilled)
#pragma unroll
    for (int e = 0; e < 4; ++e) gen_pair(0, e, A[0]);
    // every LDS read above has returned before this wave passes position 0's barrier, behind which the other waves' DMAs
    // start refilling that raw buffer and slot 0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- K loop.  chunk < n - 1: the uniform body of m4d_wino6.hip; the raw DMA of chunk c fetches chunk c + 2 of the unified
    // stream (from chunk n - 2 on: the next unit's chunks 0, 1), the fragment DMA of (c, position p) chunk c + 1's position p
    auto chunk_body = [&](int chunk, auto first_tag) __attribute__((always_inline)) {
      constexpr bool FIRST = decltype(first_tag)::value != 0;
      if (chunk == n - 2) set_raw_source(nxt);
      const unsigned char* wn = wc + w_chunk;
      const int roff = (chunk + 2 < n ? chunk + 2 : chunk + 2 - n) * 64;
      const int rbuf_w = (par + chunk) & 1;
      const float4* rnext = raw_ptr(rbuf_w ^ 1);
      // position 0: A(1) from t1, t2
      __builtin_amdgcn_s_barrier();
      P6_BLOCK0(0, 1, 1, 7, true)
      raw_dma(roff, rbuf_w, 0);
      P6_BLOCK1(0, 1, 1, 4, true)
      b_dma(wn, 0, voff_cur);
      P6_BLOCK2(0, 1, 1, 4, true)
      if (!(M4D_W6P_ABL & 8) || chunk > 0) { P6_WAIT(7); }
      // position 1: A(2) from t2, t1
      __builtin_amdgcn_s_barrier();
      P6_BLOCK0(1, 2, 2, 7, true)
      raw_dma(roff, rbuf_w, 1);
      P6_BLOCK1(1, 2, 2, 4, true)
      b_dma(wn + w_pos, 1, voff_cur);
      P6_BLOCK2(1, 2, 2, 4, true)
      if (!(M4D_W6P_ABL & 8) || chunk > 0) { P6_WAIT(8); }
      // position 2: A(3) from t1, t3; columns 0, 2 of t(chunk + 1)
      __builtin_amdgcn_s_barrier();
      P6_BLOCK0(2, 3, 3, 7, true)
      raw_dma(roff, rbuf_w, 2);
      read_t(rnext, 0);
      P6_BLOCK1(2, 3, 3, 6, true)
      b_dma(wn + 2 * w_pos, 2, voff_cur);
      pin_t(0);
      P6_BLOCK2(2, 3, 3, 6, true)
      P6_WAIT(8);
      // position 3: columns 1, 3 of t(chunk + 1) first (A(chunk + 1, 0) = t0 - t2 needs columns 0 and 2 only)
      __builtin_amdgcn_s_barrier();
      P6_BLOCK0(3, 0, 0, 7, true)
      raw_dma(roff, rbuf_w, 3);
      read_t(rnext, 1);
      P6_BLOCK1(3, 0, 0, 6, true)
      b_dma(wn + 3 * w_pos, 3, voff_cur);
      pin_t(1);
      P6_BLOCK2(3, 0, 0, 6, true)
      P6_WAIT(8);
      wc = wn;
    };
    chunk_body(0, m4d_int<1>{});                   // (n >= 2: the host checks)
    for (int chunk = 1; chunk < n - 1; ++chunk) chunk_body(chunk, m4d_int<0>{});
    // ---- last chunk: the next unit's raw chunk 1, its fragments of positions 0 and 1; ring slots 2 and 3 are left alone
    {
      constexpr bool FIRST = false;
      const int rbuf_w = (par + n - 1) & 1;
      __builtin_amdgcn_s_barrier();
      bias_dma(cur.ng);                            // (+1 VM op in this position: the waits below only get stricter)
      P6_BLOCK0(0, 1, 1, 7, true)
      raw_dma(64, rbuf_w, 0);
      P6_BLOCK1(0, 1, 1, 4, true)
      b_dma(wnext, 0, voff_nxt);
      P6_BLOCK2(0, 1, 1, 4, true)
      P6_WAIT(7);
      __builtin_amdgcn_s_barrier();
      P6_BLOCK0(1, 2, 2, 7, true)
      raw_dma(64, rbuf_w, 1);
      P6_BLOCK1(1, 2, 2, 4, true)
      b_dma(wnext + w_pos, 1, voff_nxt);
      P6_BLOCK2(1, 2, 2, 4, true)
      P6_WAIT(8);
      __builtin_amdgcn_s_barrier();
      P6_BLOCK0(2, 3, 3, 7, true)
      raw_dma(64, rbuf_w, 2);
      P6_BLOCK1(2, 3, 3, 4, true)
      P6_BLOCK2(2, 3, 3, 4, true)
      P6_WAIT(5);                                  // everything older than position 1's DMAs: slot 0 of the next unit
      __builtin_amdgcn_s_barrier();                // every wave is done with ring slots 2 and 3: they are the epilogue's from here on
      P6_BLOCK0(3, 0, 0, 0, false)
      raw_dma(64, rbuf_w, 3);
      P6_BLOCK1(3, 0, 0, 0, false)
      P6_BLOCK2(3, 0, 0, 0, false)
      P6_WAIT(2);                                  // ... older than positions 2 / 3's raw pieces: slot 1 of the next unit
    }

    // ---- output transform, bias, leaky_relu, stores: rows of A^T (M A) through LDS, one (output column k, M-tile) per pass with
    // BOTH N-tiles: the four waves of the M-tile write their 2 x 16 row-transformed values of that column, every thread then
    // finishes one (tile, cout quad of 64) -- 16 lanes = 256 contiguous bytes of a pixel per store instruction: a CU issues
    // such stores at >= 21 B/clk against 12.8 B/clk for 128-byte runs (tools/micro/store_issue_probe.hip; the store path, not
    // HBM, bounds an epilogue).  Four passes in ring slots 2-3.  The stores of pass p are issued during pass p + 1 (they fill
    // its barrier waits), the last pass's AFTER the next unit's fragment DMAs of positions 2 and 3.
    {
      float* rbuf = reinterpret_cast<float*>(ldsb + pOffSlot23);
      float* oimg = a.out + (long long)cur.bi * a.h * a.w * a.Cout;
      const bool vec_ok = (a.Cout & 3) == 0;
      const bool whole = cur.tile_x + pT <= a.w && cur.tile_y + pT <= a.h;      // uniform: no per-store bounds tests on interior tiles
      const bool fast = whole && vec_ok && cur.ng * 64 + 64 <= a.Cout;
      int te = t;
      asm volatile("" : "+v"(te));                                              // the epilogue's addressing is computed here, not kept live across the K loop
      const int cq = te & 15, tl = te >> 4;                                     // this thread's item: tile 0..31 of the pass's M-tile, cout quad 0..15
      const int me = te & 31, khe = (te >> 5) & 1;                              // (= m, kh: from the opaque copy, or the row addresses below are
                                                                                //  hoisted out of the unit loop, 30 registers that get spilled)
      const int co = cur.ng * 64 + 4 * cq;
      float4 rv[4];
      auto finish = [&](int pass) {                                            // bias, leaky_relu, stores of the item read in `pass`
        const int kcol = pass >> 1, omt = pass & 1;
        const float4 bs4 = *reinterpret_cast<const float4*>(ldsb + pOffBias + 16 * cq);
        const float bs[4] = {bs4.x, bs4.y, bs4.z, bs4.w};
        const float* r0 = reinterpret_cast<const float*>(&rv[0]); const float* r1 = reinterpret_cast<const float*>(&rv[1]);
        const float* r2 = reinterpret_cast<const float*>(&rv[2]); const float* r3 = reinterpret_cast<const float*>(&rv[3]);
        float y[2][4];                             // [row l][cout]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v0 = ((r0[e] + r1[e]) + r2[e]) + bs[e];
          const float v1 = ((r1[e] - r2[e]) - r3[e]) + bs[e];
          y[0][e] = v0 > 0.f ? v0 : v0 * a.slope;
          y[1][e] = v1 > 0.f ? v1 : v1 * a.slope;
        }
        const int tile = omt * 32 + tl;            // Winograd tile 0..63 of the workgroup (8 x 8)
        const int ox = cur.tile_x + 2 * (tile & 7) + kcol, oy = cur.tile_y + 2 * (tile >> 3);
        float* op = oimg + ((long long)oy * a.w + ox) * a.Cout + co;
        if (M4D_W6P_ABL & 2) {
          if (y[0][0] + y[1][1] + y[0][2] + y[1][3] == 123.456f) op[0] = 1.f;      // keeps the arithmetic alive
        } else if (fast) {
          // UNIFORM condition (whole tile, every cout quad of the group real, 16-byte rows) and nothing but the two 16-byte
          // stores in this arm: under a per-lane condition hipcc merged this arm with the guarded one below and stored the
          // second row as four separate dwords (the 12.8-B/clk pattern of tools/micro/store_issue_probe.hip, 4x slower)
          m4d_store16(op, y[0][0], y[0][1], y[0][2], y[0][3]);
          m4d_store16(op + (long long)a.w * a.Cout, y[1][0], y[1][1], y[1][2], y[1][3]);
        } else if (co < a.Cout && ox < a.w) {
#pragma unroll
          for (int l = 0; l < 2; ++l)
            if (oy + l < a.h) {
              float* o2 = op + (long long)l * a.w * a.Cout;
              if (vec_ok && co + 3 < a.Cout) m4d_store16(o2, y[l][0], y[l][1], y[l][2], y[l][3]);
              else { for (int e = 0; e < 4; ++e) if (co + e < a.Cout) o2[e] = y[l][e]; }
            }
        }
      };
      if (M4D_W6P_ABL & 1) {                                                    // keep the K loop's products alive
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) asm volatile("" : : "v"(acc[c][nt]));
      }
#pragma unroll
      for (int pass = 0; pass < ((M4D_W6P_ABL & 1) ? 0 : 4); ++pass) {
        const int kcol = pass >> 1, omt = pass & 1;
        if (mt == omt) {
#pragma unroll
          for (int ont = 0; ont < 2; ++ont) {
            if (ont >= NTL) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float m0 = acc[0][ont][r], m1 = acc[1][ont][r], m2 = acc[2][ont][r], m3 = acc[3][ont][r];
              const int trow = (r & 3) + 8 * (r >> 2) + 4 * khe;
              rbuf[(pr * 32 + trow) * pMS2 + ont * 32 + me] = kcol == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3;
            }
          }
        }
        if (pass > 0) finish(pass - 1);
        P6_LDS_BARRIER();
#pragma unroll
        for (int i = 0; i < 4; ++i) rv[i] = *reinterpret_cast<const float4*>(rbuf + (i * 32 + tl) * pMS2 + 4 * cq);
        P6_LDS_BARRIER();                            // the pass buffer is rewritten by the next pass / the fragment DMAs below
      }
      // ---- the next unit's fragments of positions 2 and 3 (slots 2, 3 are free again); everything else of it has landed
      if (has_next) {
        b_dma(wnext + 2 * w_pos, 2, voff_nxt);
        b_dma(wnext + 3 * w_pos, 3, voff_nxt);
      }
      if (!(M4D_W6P_ABL & 1)) finish(3);
    }
    };
    if (half) unit_body(m4d_int<1>{}); else unit_body(m4d_int<2>{});
    if (!has_next) break;
    cur = nxt;
    wc = wnext;
    par = (par + n) & 1;
  }
  P6_WAIT(0);                                      // the last unit's surplus DMAs: nothing may land in LDS after the workgroup is gone
#undef P6_MFMA
#undef P6_BLOCK0
#undef P6_BLOCK1
#undef P6_BLOCK2
#undef P6_PIPE
#undef P6_WAIT
#undef P6_LDS_BARRIER
}

}  // namespace

// Launch for m4d_conv3x3_wino6_bias_act (m4d_wino6.hip decides when): Cin >= 32.
// units_per_wg > 0: workgroups of that many consecutive units each (grid = ceil(units / units_per_wg), as many workgroups as
// that takes: the dispatcher places them as CUs free, one per CU at a time) instead of one static range per CU.
int m4d_wino6p_launch(const float* x, const void* wu6, const float* bias, int b, int h, int w, int Cin, int Cout, int CoutPad,
                      float slope, float* out, int units_per_wg, int stagger_us, int stagger_phases, void* stream) {
  M4D_CHECK_ARG(Cin % 16 == 0 && Cin >= 32 && CoutPad % 64 == 0 && CoutPad >= Cout);
  Wino6PArgs a;
  a.x = x; a.wu = reinterpret_cast<const unsigned char*>(wu6); a.bias = bias; a.out = out;
  a.b = b; a.h = h; a.w = w; a.Cin = Cin; a.Cout = Cout; a.CoutPad = CoutPad; a.n_chunks = Cin / 16; a.slope = slope;
  a.tiles_x = (w + pT - 1) / pT; a.tiles_y = (h + pT - 1) / pT;
  const long long units = (long long)b * a.tiles_x * a.tiles_y * (CoutPad / 64);
  M4D_CHECK_ARG(units < (1ll << 31));
  a.units = (int)units;
  // CU count and the > 64 KB dynamic-LDS opt-in per DEVICE (a process may drive several GPUs)
  M4D_LDS_OPT_IN(&conv3x3_wino6p_kernel);
  const int n_cu = m4d_device_cus();
  // One workgroup per CU at most (154 KB of LDS each).  Large grids: team mode on every CU (see the kernel).  Smaller ones: no
  // more workgroups than the longest range needs -- 960 units on 256 CUs are 4 per workgroup whichever way, so 240 workgroups
  // do it and 16 CUs stay free for the other frames' small kernels.
  const int n_groups = CoutPad / 64;
  a.team = (units_per_wg <= 0 && n_cu % 8 == 0 && n_cu / 8 >= n_groups && units >= 4ll * n_cu) ? 1 : 0;
  a.stagger = stagger_us * 100; a.phases = stagger_us > 0 ? stagger_phases : 1;
  // M4D_WINO6P_MAX_WG (measurement knob): at most this many persistent workgroups -- the other CUs stay free for the kernels of
  // the other frames' coarse levels, whose workgroups otherwise each wait for one of this kernel's ~27-us units to end
  static const int max_wg = [] { const char* e = getenv("M4D_WINO6P_MAX_WG"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1 << 30; }();
  const int cu_used = n_cu < max_wg ? n_cu : max_wg;
  const long long per_wg = (units + cu_used - 1) / cu_used;
  const unsigned grid = units_per_wg > 0 ? (unsigned)((units + units_per_wg - 1) / units_per_wg)
                                         : a.team ? (unsigned)n_cu : (unsigned)((units + per_wg - 1) / per_wg);
  m4d_launch(conv3x3_wino6p_kernel, dim3(grid), dim3(512), (size_t)pLds, (hipStream_t)stream, a);
  return M4D_LAUNCH_RESULT();
}
