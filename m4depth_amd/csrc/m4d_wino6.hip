// 3x3 stride-1 'SAME' convolution + bias + leaky_relu: Winograd F(2x2,3x3) with FLOAT32 operands on the BF16 matrix cores.
//
// fp32 MFMA (m4d_wino.hip) runs at the float32 vector rate, 1/16 of the bf16 MFMA rate.  Every float32 number is EXACTLY the
// sum of three bf16 numbers (8 + 8 + 8 significand bits: p1 = bf16(v), p2 = bf16(v - p1), p3 = (v - p1) - p2, round to
// nearest), and a bf16 x bf16 product is exact in the matrix core's float32 accumulator, so
//     a * b = sum_{i,j} a_i b_j     (9 terms);   dropping a2 b3, a3 b2, a3 b3 (each <= 2^-26 |a b|, below float32's own
//                                                product rounding) leaves 6 bf16 MFMAs per 16 k
// -- 6 x 32 cycles against 8 x 64 for v_mfma_f32_32x32x2_f32: 2.67x less matrix-core time at float32 accuracy (measured:
// error against float64 0.8x that of the fp32 MFMA chain, no bias: tools/micro/bf16_split_probe.hip,
// profiles/r02_bf16_split_probe.txt).  The filter transform U = G g G^T is split on the host (pack_conv_weights_wino6);
// the input transform V = B^T d B is computed and split in registers by the wave that multiplies it.
//
// Workgroup = 8 waves, TWO PER SIMD (256 registers each), = a 16x16-pixel output tile (64 Winograd tiles = two MFMA M-tiles)
// x 64 output channels (two N-tiles).  Wave (r, mt) owns row r of the 4x4 Winograd positions for M-tile mt: 4 positions x 2 N
// = 8 accumulators (128 AGPRs).  Its A operands need no exchange: V[r][c] of a tile depends on two raw rows of that tile only,
// so lane (tile m, k-half) reads those 2 x 4 pixels x 8 channels from the raw halo in LDS, forms t_c = d[ra][c] +- d[rb][c]
// once per 16-channel chunk, and per position V = t_c +- t_c', the 3-way split and the packing: 7.5 VALU instructions per
// element, ~5 per MFMA -- the kernel is balanced between the VALU port (which the MFMAs share: 8 + 4 n cycles for an MFMA and
// n other instructions of one wave, tools/micro/mfma_bf16_dep.hip) and the matrix core, not MFMA-bound any more; the second
// wave of the SIMD covers the LDS / DMA / scalar instructions and every latency.  B operands (the split U, fragment order:
// 1 KB per (position, N-tile, part)) and the raw halo both reach LDS by LDS-DMA (global_load_lds / buffer_load ... lds: no
// staging registers, no ds_write pass): the two waves of a position row share a 4-position ring of B fragments (filled four
// positions ahead, half by each), all waves share the double-buffered raw halo (29 KB each, [row][column parity][column / 2]
// [channel quad + 1 pad slot]: conflict-free 16-byte reads AND four consecutive DMA lanes = 64 contiguous bytes of one pixel --
// round 3: the quad-major layout before it made every DMA lane fetch 16 bytes of a different pixel, 64 lines per instruction,
// and cost 10 % of the kernel; pixels outside the image come back as zeros from the buffer descriptor's range check).  The DMAs are inline asm with hand-counted s_waitcnt vmcnt(N) -- the compiler would drain every DMA in
// flight (vmcnt(0)) before any LDS read it cannot prove independent; one raw s_barrier per position makes the partner's
// fragments visible.  The B registers are refilled part by part as the MFMAs of the current position retire them.
// Epilogue as in kernel 4 of m4d_wino.hip: rows of A^T (M A) through LDS, bias + leaky_relu, 16-byte stores (16 lanes = the 256
// contiguous bytes of a pixel's 64 couts).
// Round 4: a unit is prologue 2.6 + K loop 24 + epilogue 2.5 us on the level-1 128 -> 128 layer, and at batch 1 every layer is
// 1-4 such units per CU, so the unit's BOUNDARY counts: the first chunk is peeled and multiplies onto the constant 0 (no
// accumulator zeroing), the last chunk is peeled and issues no DMA (nothing for the epilogue to drain).  HALF units: a cout group
// of <= 32 channels (Cout = 96's second group, Cout = 32) runs N-tile 0 only, a second instantiation of the K loop.
// Deterministic: fixed summation order, no atomics.
#include <cstdlib>
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

#ifndef M4D_W6_ABL
#define M4D_W6_ABL 0       // timing ablations of the K loop (wrong results; tools/w6_ablate.sh): 1 no barrier, 2 no fragment DMA,
#endif                     // 4 no raw-halo DMA, 8 no fragment LDS reads, 16 no A-operand generation, 32 no DMA waits

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Wino6Args {
  const float* x; const unsigned char* wu; const float* bias; float* out;
  int b, h, w, Cin, Cout, CoutPad, n_chunks, tiles_x, tiles_y;
  float slope;
  unsigned long long* stamps;   // profiling only (m4d_wino6_set_stamps, tools/wino6_phases.py): 64 workgroups x 1280 words
  int phases;                   // power of two <= 32
  int stagger;                  // > 0: the first-round workgroups start up to this many 100-MHz ticks apart (see the launcher)
};

constexpr int kT = 16, kH = kT + 2;              // output tile, halo (pixels)
// Raw halo in LDS, [row][column parity][column / 2][channel quad + 1 pad slot]: the four 16-byte channel quads of a pixel are
// CONSECUTIVE slots, so four consecutive lanes of an LDS-DMA instruction fetch 64 contiguous bytes of one pixel (13 pixels =
// 13 lines per instruction; the layout of rounds 2-3, quad-major, made every lane fetch 16 bytes of a different pixel: 64 lines
// per instruction through a texture path that was busy 52-59 % of the kernel).  Pixel stride 5 slots, two rows = 200 slots =
// 8 mod 16: the 16-lane groups of a ds_read_b128 (tiles (ty0, tx) -> 8 ty0 + 5 tx mod 16) still cover 16 distinct slots.
constexpr int kJ = 10;                           // pixels per (row, column parity): 9 used
constexpr int kPix = 5;                          // slots per pixel: 4 channel quads + 1 pad
constexpr int kRow = 2 * kJ * kPix;              // slots per halo row (100)
constexpr int kRawUsed = kH * kRow;              // per chunk of 16 channels: 1800 slots
constexpr int kRawDma = 29;                      // LDS-DMA instructions per chunk: 64 slots each
constexpr int kRawK = 4;                         // ... = up to 4 per wave
constexpr int kRawSlots = kRawDma * 64;          // slots per buffer
constexpr int kBRingBytes = 4 * 6 * 1024;        // per position row: 4 positions x 6 fragments of 1 KB
constexpr int kBRingOff = 2 * kRawSlots * 16;    // byte offset of the B rings in LDS (47104)
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  bf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lo_f32(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f32(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// v[0..7] -> three bf16x8 operands (hi, mid, lo): v = hi + mid + lo exactly
__device__ __forceinline__ void split8(const float* v, bf16x8& a0, bf16x8& a1, bf16x8& a2) {
  u32x4 p0, p1, p2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x0 = v[2 * e], x1 = v[2 * e + 1];
    const unsigned q0 = pk_bf16(x0, x1);
    const float r0 = x0 - lo_f32(q0), r1 = x1 - hi_f32(q0);
    const unsigned q1 = pk_bf16(r0, r1);
    const float s0 = r0 - lo_f32(q1), s1 = r1 - hi_f32(q1);
    p0[e] = q0; p1[e] = q1; p2[e] = pk_bf16(s0, s1);
  }
  a0 = __builtin_bit_cast(bf16x8, p0); a1 = __builtin_bit_cast(bf16x8, p1); a2 = __builtin_bit_cast(bf16x8, p2);
}

template <bool STAMPS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv3x3_wino6_kernel(const Wino6Args a) {
  extern __shared__ __align__(16) float lds[];
  // every kernel argument fetched by the FIRST scalar loads (hipcc issued the load of the pointers ~100 instructions later, behind
  // the tile decode's divisions: a scalar-memory round trip right in front of the first DMA of every unit)
  asm volatile("" : : "s"(a.x), "s"(a.wu), "s"(a.out), "s"(a.bias));
  if (a.stagger > 0 && blockIdx.y == 0 && blockIdx.x < 256) {
    // One-per-CU workgroups of identical duration run in lock step: every CU frees at the same instant, once per unit time, and a
    // small kernel of another queue that arrives in between waits for that instant.  The first round's workgroups (the CUs of
    // an XCD take consecutive ids / 8) start in eight phases instead.
    const unsigned phase = (blockIdx.x >> 3) & (unsigned)(a.phases - 1);
    const unsigned long long t_end = wall_clock64() + (unsigned long long)(phase * (unsigned)a.stagger) / (unsigned)a.phases;
    for (int spin = 0; spin < 512 && wall_clock64() < t_end; ++spin) __builtin_amdgcn_s_sleep(16);   // (bounded: a delay, never a hang)
  }
  float4* raw = reinterpret_cast<float4*>(lds);                      // [2][kRawSlots]
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;   // LDS byte address (for M0)

  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);             // scalar
  const int pr = wv & 3, mt = wv >> 2;                               // position row, M-tile (waves r and r + 4 share a SIMD)
  const int m = lane & 31, kh = lane >> 5;
  const int n_tiles = a.tiles_x * a.tiles_y, n_groups = a.CoutPad / 64;
  int tile, ng;
  {
    const int L = blockIdx.x;
    if ((n_tiles & 7) == 0) {                                        // consecutive workgroups go to different XCDs: keep the
      const int xcd = L & 7, idx = L >> 3;                           // N-groups of one pixel tile on one XCD (its L2 holds the halo)
      tile = xcd * (n_tiles >> 3) + idx / n_groups;
      ng = idx % n_groups;
    } else {
      tile = L / n_groups;
      ng = L % n_groups;
    }
  }
  const int tile_y = (tile / a.tiles_x) * kT, tile_x = (tile % a.tiles_x) * kT;
  const int bi = blockIdx.y;
  const int n = a.n_chunks, last = n - 1;
  const float* ximg = a.x + (long long)bi * a.h * a.w * a.Cin;

  // ---- raw halo by LDS-DMA: instruction i fills slots 64 i .. 64 i + 63 (lane = slot); wave wv issues i = wv, wv + 8,
  // wv + 16, wv + 24 (29 pieces: waves 5-7 repeat the 29th, same bytes to the same slots).  Per lane: the byte offset of its slot's (pixel, channel quad) in the image, or an
  // offset past the buffer's num_records for pixels outside the image and pad slots: the range check returns zeros.
  i32x4 rsrc;
  {
    const unsigned long long xa = (unsigned long long)ximg;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(xa & 0xffffffffull));
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)((xa >> 32) & 0xffffull));            // stride 0: raw buffer
    rsrc[2] = a.h * a.w * a.Cin * 4;                                                     // num_records (bytes)
    rsrc[3] = 0x00020000;
  }
  unsigned rvoff[kRawK];
#pragma unroll
  for (int k = 0; k < kRawK; ++k) {
    const int i = min(wv + 8 * k, kRawDma - 1);
    const int s = i * 64 + lane;
    const int pix = s / kPix, q = s - pix * kPix;                    // q = 4: the pad slot
    const int hy = pix / (2 * kJ), r2 = pix - hy * (2 * kJ);
    const int e = r2 / kJ, j = r2 - e * kJ;
    const int hx = 2 * j + e;
    const int gy = tile_y - 1 + hy, gx = tile_x - 1 + hx;
    const bool ok = s < kRawUsed && q < 4 && j < 9 && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    rvoff[k] = ok ? (unsigned)(((gy * a.w + gx) * a.Cin + q * 4) * 4) : 0x80000000u;
  }
  // DMA k of raw(chunk) into buffer `buf`: LDS destination (wave-uniform byte address) in M0, + lane * 16
  auto raw_dma = [&](int chunk, int buf, int k) {
    const int i = min(wv + 8 * k, kRawDma - 1);
    const unsigned lds_dst = lds_base + (unsigned)((buf * kRawSlots + i * 64) * 16);
    // (every DMA statement sets M0 itself and declares it clobbered: nothing else in this kernel uses M0)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                 : : "v"(rvoff[k]), "s"(lds_dst), "s"(rsrc), "s"(chunk * 64) : "memory", "m0");
  };

  // ---- this lane's tile: M-tile mt holds tile rows 4 mt .. 4 mt + 3; lane m = (row m >> 3, column m & 7)
  // position row pr of B^T d uses raw rows (ra, rb) of the 4x4 input tile: d0 - d2, d1 + d2, d2 - d1, d1 - d3
  const int ra = pr == 0 ? 0 : (pr == 2 ? 2 : 1);
  const int rb_ = pr == 0 ? 2 : (pr == 1 ? 2 : (pr == 2 ? 1 : 3));
  const float sgn = pr == 1 ? 1.f : -1.f;
  const int ty0 = m >> 3, tx = m & 7;
  // slot of (raw row 0 of the tile, column 0, quad 2 kh); + row * kRow, + ((c & 1) * kJ + (c >> 1)) * kPix, + quad
  const int src0 = (8 * mt + 2 * ty0) * kRow + tx * kPix + 2 * kh;

  float tv[4][8];                                  // t_c of the current chunk: [column c][channel]
  // columns c0, c0 + 2 of t for the chunk in rbuf (two calls per chunk: the registers of columns 0, 2 are free one
  // position earlier than those of columns 1, 3)
  auto read_t = [&](const float4* rbuf, int c0) {
#pragma unroll
    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int c = c0 + 2 * cc;
        const int s = src0 + qq + ((c & 1) * kJ + (c >> 1)) * kPix;
        const float4 da = rbuf[s + ra * kRow], db = rbuf[s + rb_ * kRow];
        tv[c][4 * qq + 0] = __builtin_fmaf(sgn, db.x, da.x);           // exact product: one rounding, = da +- db
        tv[c][4 * qq + 1] = __builtin_fmaf(sgn, db.y, da.y);
        tv[c][4 * qq + 2] = __builtin_fmaf(sgn, db.z, da.z);
        tv[c][4 * qq + 3] = __builtin_fmaf(sgn, db.w, da.w);
      }
  };
  // A operands of position (pr, c): V = (t B)_c = t0 - t2, t1 + t2, t2 - t1, t1 - t3; element pair e (channels 2e, 2e + 1)
  // -> one packed word of each of the three bf16x8 operands (hi, mid, lo)
  auto gen_pair = [&](int c, int e, u32x4 (&A)[3]) {
    float v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ch = 2 * e + h;
      v[h] = c == 0 ? tv[0][ch] - tv[2][ch] : c == 1 ? tv[1][ch] + tv[2][ch] : c == 2 ? tv[2][ch] - tv[1][ch] : tv[1][ch] - tv[3][ch];
    }
    unsigned p_hi, p_mid, p_lo;
    m4d_split3_pair(v[0], v[1], p_hi, p_mid, p_lo);                            // m4d_common.h
    A[0][e] = p_hi; A[1][e] = p_mid; A[2][e] = p_lo;
  };

  // ---- B operands: wu[chunk][N-group][16 positions][2 N-tiles][3 parts][64 lanes][8 bf16]; a position row walks its 4
  // positions chunk after chunk (6 KB per position, contiguous) into its LDS ring: position q -> slot q & 3 = its column
  // (static), DMA'd four positions ahead -- fragments 3 mt .. 3 mt + 2 by wave (pr, mt).
  const long long w_pos = 6 * 1024;
  const long long w_chunk = (long long)n_groups * 16 * w_pos;
  const unsigned char* wc = a.wu + ((long long)ng * 16 + 4 * pr) * w_pos;        // uniform: this row, chunk 0
  const unsigned bring = lds_base + (unsigned)(kBRingOff + pr * kBRingBytes);    // LDS byte address of this row's ring
  const unsigned char* bring_p = reinterpret_cast<const unsigned char*>(lds) + kBRingOff + pr * kBRingBytes;
  const unsigned bl = (unsigned)lane * 16u;
  // HALF unit (uniform): only N-tile 0 of this cout group holds real output channels (Cout = 96: the second group) -- its K loop
  // issues N-tile 0's MFMAs only and the epilogue handles N-tile 0 only.  The DMA protocol stays what it is (same pieces per
  // wave and position, same vmcnt counts), but the waves that fetch N-tile 1's fragments (mt = 1) point all 64 lanes at one
  // 16-byte word: one cache line per piece instead of sixteen, into ring bytes nobody reads.
  const bool half = __builtin_amdgcn_readfirstlane(ng * 64 + 32 >= a.Cout ? 1 : 0) != 0;    // (scalar: the K loop branches on it)
  const unsigned bl_dma = (half && mt == 1) ? 0u : bl + (unsigned)mt * 3072u;
  // (the instruction offset of an LDS-DMA load moves BOTH the global address and the LDS address; the per-lane offset moves the
  // global address only: lane l always lands at M0 + 16 l)
  auto b_dma = [&](const unsigned char* gsrc, int slot) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %2\n\tglobal_load_lds_dwordx4 %0, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %2 offset:2048"
                 : : "v"(bl_dma), "s"(bring + (unsigned)(slot * 6144) + (unsigned)mt * 3072u), "s"(gsrc) : "memory", "m0");
  };
  bf16x8 B0[2][2], B1[2], B2[2];                   // parts (hi, mid, lo) of the B operand per N-tile; the hi part double buffered
  auto frag = [&](int slot, int nt, int part) {
    return *reinterpret_cast<const bf16x8*>(bring_p + slot * 6144 + (nt * 3 + part) * 1024 + bl);
  };
  auto frag_l = [&](bf16x8& dst, int slot, int nt, int part) { if (!(M4D_W6_ABL & 8)) dst = frag(slot, nt, part); };
  auto gen_l = [&](int c, int e, u32x4 (&A)[3]) { if (!(M4D_W6_ABL & 16)) gen_pair(c, e, A); };
#define M4D_W6_WAIT(nn) asm volatile("s_waitcnt vmcnt(" #nn ") lgkmcnt(0)" ::: "memory")

  f32x16 acc[4][2];                                // (never zeroed: the first chunk multiplies onto the constant 0)

  // stamps (profiling build only): lane 0 of every wave of the first 64 workgroups; per workgroup 1280 words: per wave 32
  // positions x (after barrier, -, before wait, after wait), then at 1024 the header of wave 0 (start, end of K loop, end)
  unsigned long long* st = (STAMPS && a.stamps != nullptr && lane == 0 && blockIdx.y == 0 && blockIdx.x < 64)
                               ? a.stamps + (long long)blockIdx.x * 1280 + wv * 128 : nullptr;
  if (STAMPS && st && wv == 0) st[1024 + 0] = __builtin_readcyclecounter();

  // ---- prologue: raw(0), B(0), B(1) | B(2), raw(1), B(3) by DMA in the order the K loop's vmcnt counts
  // assume: every end-of-position wait of the K loop is vmcnt(4 + 4) = "this position's and the previous position's DMAs may
  // still fly".  First chunk, DMAs in flight oldest first (kRawK = 4, pieces in brackets):
  //   after the prologue wait (10):  B(2)[3] raw(1)[4] B(3)[3]
  //   end of position 0 (+4, wait 8): retires B(2)[3] raw(1)[3]      -> position 1 reads slot 2: landed
  //   end of position 1 (+4, wait 8): retires raw(1)[1] B(3)[3]      -> position 2 reads slot 3 and raw(1): landed
  // then t(0), A(0, 0), B(0) in registers
#pragma unroll
  for (int k = 0; k < kRawK; ++k) raw_dma(0, 0, k);
  b_dma(wc, 0);
  b_dma(wc + w_pos, 1);
  // B(2) BEFORE raw(1): position 1 reads slot 2, position 2 reads raw(1).  The end-of-position waits of the first chunk retire
  // the prologue's DMAs oldest first, six per position: with raw(1) issued first (rounds 2-3) the wait that closes position 0
  // retired raw(1) and only TWO of B(2)'s three pieces, and position 1 read the third (the lo part of N-tile 0 / 1) with
  // nothing but elapsed time between the DMA and the read -- stale LDS under a loaded memory system, i.e. the
  // non-deterministic replicas of round 3's driver run.
  b_dma(wc + 2 * w_pos, 2);
#pragma unroll
  for (int k = 0; k < kRawK; ++k) raw_dma(min(1, last), 1, k);
  b_dma(wc + 3 * w_pos, 3);
  static_assert(kRawK == 4, "the vmcnt counts below assume 4 raw-halo pieces + 3 fragment pieces per position");
  M4D_W6_WAIT(10);                                 // raw(0), B(0), B(1) landed; B(2), raw(1), B(3) still in flight
  __builtin_amdgcn_s_barrier();
  u32x4 A[2][3];                                   // [ring][part]: packed bf16 pairs
  // (t(0), A(0, 0), B(0) into registers: first thing in the K loop's instantiation -- hipcc lays the two instantiations out one
  // after the other behind scalar guards, and anything produced HERE for both would stay live across the first one's loop)

  // The compiler may sink pure arithmetic past a sched_barrier (only the machine scheduler honours it): an empty asm that
  // "modifies" the freshly produced operands pins their producers inside the region they are meant to overlap with.
  auto pin_a = [&](u32x4 (&X)[3]) {
#pragma unroll
    for (int part = 0; part < 3; ++part) asm volatile("" : "+v"(X[part]));
  };
  auto pin_t = [&](int c0) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(tv[c0 + 2 * cc][e]));
  };
  // (NTL = the unit's N-tiles, a compile-time constant of the K loop's two instantiations below: 2, or 1 for a half unit)
#define M4D_W6_MFMA(c, ap, bv)                                                                                         \
  _Pragma("unroll") for (int nt = 0; nt < NTL; ++nt)                                                                   \
    acc[c][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[(c) & 1][ap]), bv[nt], acc[c][nt], 0, 0, 0);
  // the first product into an accumulator: in the first chunk (FIRST, peeled below) on top of the constant 0 -- the 128
  // accumulator registers are never zeroed (that was 128 v_mov per wave behind the prologue's DMA wait, on every unit's critical path)
#define M4D_W6_MFMA0(c, ap, bv)                                                                                        \
  _Pragma("unroll") for (int nt = 0; nt < NTL; ++nt)                                                                   \
    acc[c][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[(c) & 1][ap]), bv[nt],            \
                                                         FIRST ? f32x16{} : acc[c][nt], 0, 0, 0);
  // one MFMA, then `valu` vector instructions (8 + 4 n cycles of the VALU port for one MFMA and n others; the MFMA runs 32);
  // a block has 2 NTL MFMAs and the same vector work either way
#define M4D_W6_PIPE(valu)                                                                                              \
  _Pragma("unroll") for (int i_ = 0; i_ < 2 * NTL; ++i_) {                                                             \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x002, (valu) * (2 / NTL), 0);                                                \
  }
  // One position = three blocks of 4 MFMAs (6 of the 9 term products, the small ones first), each followed by one DMA: the
  // texture path (16 cycles per 1-KB piece, 32 pieces per position and CU) then works beside the MFMAs instead of in a burst
  // after the barrier that every wave would sit through.  The B registers of a part are refilled for the next position as
  // soon as the part's last MFMA is issued (lo after block 0, mid after block 1); the hi part, needed until the end, is
  // double buffered.
#define M4D_W6_BLOCK0(c, cn, next_slot, valu, NEXT)                                                                    \
  if (NEXT) { frag_l(B0[((c) & 1) ^ 1][0], next_slot, 0, 0); if (NTL == 2) frag_l(B0[((c) & 1) ^ 1][1], next_slot, 1, 0); \
              gen_l(cn, 0, A[((c) & 1) ^ 1]); gen_l(cn, 1, A[((c) & 1) ^ 1]); }                                          \
  M4D_W6_MFMA0(c, 0, B2) M4D_W6_MFMA(c, 2, B0[(c) & 1])                                                                \
  if (NEXT) { asm volatile("" : "+v"(A[((c) & 1) ^ 1][0][0]), "+v"(A[((c) & 1) ^ 1][0][1])); M4D_W6_PIPE(valu) }         \
  __builtin_amdgcn_sched_barrier(0);
#define M4D_W6_BLOCK1(c, cn, next_slot, valu, NEXT)                                                                    \
  if (NEXT) { frag_l(B2[0], next_slot, 0, 2); if (NTL == 2) frag_l(B2[1], next_slot, 1, 2);                              \
              gen_l(cn, 2, A[((c) & 1) ^ 1]); }                                                                          \
  M4D_W6_MFMA(c, 1, B1) M4D_W6_MFMA(c, 0, B1)                                                                          \
  if (NEXT) { asm volatile("" : "+v"(A[((c) & 1) ^ 1][0][2])); M4D_W6_PIPE(valu) }                                     \
  __builtin_amdgcn_sched_barrier(0);
#define M4D_W6_BLOCK2(c, cn, next_slot, valu, NEXT)                                                                    \
  if (NEXT) { frag_l(B1[0], next_slot, 0, 1); if (NTL == 2) frag_l(B1[1], next_slot, 1, 1);                              \
              gen_l(cn, 3, A[((c) & 1) ^ 1]); }                                                                          \
  M4D_W6_MFMA(c, 1, B0[(c) & 1]) M4D_W6_MFMA(c, 0, B0[(c) & 1])                                                        \
  if (NEXT) { pin_a(A[((c) & 1) ^ 1]); M4D_W6_PIPE(valu) }                                                             \
  __builtin_amdgcn_sched_barrier(0);

  // Uniform loop body (no branches: one scheduling region per position); the first and the last chunk are peeled.
  // Per position: barrier (the fragments of the NEXT position, DMA'd by both waves of the row, are visible from here on);
  // DMAs (one raw piece at columns 0-2, this wave's half of B four positions ahead); the MFMAs of this position interleaved
  // with the A operands of the next; then vmcnt(N) leaves exactly the DMAs of this and the previous position in flight
  // (N = 6 + their raw pieces), i.e. everything the barrier of the next position publishes has landed.
  int stq = 0;
#define M4D_W6_STAMP(k) if (STAMPS && st && stq < 32) st[stq * 4 + (k)] = __builtin_readcyclecounter();
#define W6L_BARRIER() do { if (!(M4D_W6_ABL & 1)) __builtin_amdgcn_s_barrier(); } while (0)
#define W6L_RAW(...) do { if (!(M4D_W6_ABL & 4)) raw_dma(__VA_ARGS__); } while (0)
#define W6L_BDMA(...) do { if (!(M4D_W6_ABL & 2)) b_dma(__VA_ARGS__); } while (0)
#define W6L_WAIT(nn) do { if (!(M4D_W6_ABL & 32)) { M4D_W6_WAIT(nn); } } while (0)
  auto k_loop = [&](auto ntl_tag) __attribute__((always_inline)) {
  constexpr int NTL = decltype(ntl_tag)::value;
  asm volatile("" ::: "memory");                   // (nothing below is hoisted above the branch that picks the instantiation)
  read_t(raw, 0);
  read_t(raw, 1);
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt) { B0[0][nt] = frag(0, nt, 0); B1[nt] = frag(0, nt, 1); B2[nt] = frag(0, nt, 2); }
#pragma unroll
  for (int e = 0; e < 4; ++e) gen_pair(0, e, A[0]);
  // every LDS read above (raw buffer 0, ring slot 0) has returned before this wave passes position 0's barrier, behind which
  // the other waves' DMAs start refilling that buffer and that slot (the K loop's waits include lgkmcnt(0) too)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // LAST chunk (peeled): no DMA -- there is nothing left to fetch, and the surplus fetches of rounds 2-4 (harmless re-fetches of
  // the last chunk) were 8 pieces in flight at the end of the K loop that the epilogue had to drain before reusing the LDS.  Its
  // waits, DMAs in flight at its entry oldest first:
  //   after a full chunk:           (n-2, 2)[4] (n-2, 3)[4]  -> position 0 ends with vmcnt(4) (slot 2, read by position 1, landed),
  //                                                             position 1 with vmcnt(0) (slot 3), positions 2 / 3 need none
  // (never the first chunk as well: a single-chunk layer runs the first chunk's body, with the surplus fetches of old)
  // Position 3 of it prepares nothing for a next chunk (NEXT = false), position 2 reads no t(chunk + 1).
  auto chunk_body = [&](int chunk, auto first_tag, auto last_tag) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value != 0, LAST = decltype(last_tag)::value != 0;
    const unsigned char* wn = chunk < last ? wc + w_chunk : wc;                   // scalar select: B of the next chunk
    const int rnext_c = min(chunk + 2, last);
    const float4* rnext = raw + ((chunk + 1) & 1) * kRawSlots;
    // position 0: A(1) from t1, t2
    W6L_BARRIER();
    M4D_W6_STAMP(0)
    M4D_W6_BLOCK0(0, 1, 1, 7, true)
    if (!LAST) W6L_RAW(rnext_c, chunk & 1, 0);
    M4D_W6_BLOCK1(0, 1, 1, 4, true)
    if (!LAST) W6L_BDMA(wn, 0);
    M4D_W6_BLOCK2(0, 1, 1, 4, true)
    M4D_W6_STAMP(2)
    if (!LAST) W6L_WAIT(8);                                                       // this position's DMAs + the previous position's
    else W6L_WAIT(4);
    M4D_W6_STAMP(3)
    if (STAMPS) ++stq;
    // position 1: A(2) from t2, t1
    W6L_BARRIER();
    M4D_W6_STAMP(0)
    M4D_W6_BLOCK0(1, 2, 2, 7, true)
    if (!LAST) W6L_RAW(rnext_c, chunk & 1, 1);
    M4D_W6_BLOCK1(1, 2, 2, 4, true)
    if (!LAST) W6L_BDMA(wn + w_pos, 1);
    M4D_W6_BLOCK2(1, 2, 2, 4, true)
    M4D_W6_STAMP(2)
    if (!LAST) W6L_WAIT(8);
    else W6L_WAIT(0);
    M4D_W6_STAMP(3)
    if (STAMPS) ++stq;
    // position 2: A(3) from t1, t3; columns 0, 2 of t(chunk + 1)
    W6L_BARRIER();
    M4D_W6_STAMP(0)
    M4D_W6_BLOCK0(2, 3, 3, 7, true)
    if (!LAST) { W6L_RAW(rnext_c, chunk & 1, 2); read_t(rnext, 0); }
    M4D_W6_BLOCK1(2, 3, 3, 6, true)
    if (!LAST) { W6L_BDMA(wn + 2 * w_pos, 2); pin_t(0); }
    M4D_W6_BLOCK2(2, 3, 3, 6, true)
    M4D_W6_STAMP(2)
    if (!LAST) W6L_WAIT(8);
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // (the fragment reads for position 3)
    M4D_W6_STAMP(3)
    if (STAMPS) ++stq;
    // position 3: columns 1, 3 of t(chunk + 1) first (A(chunk + 1, 0) = t0 - t2 needs column ... 0 and 2 only)
    W6L_BARRIER();
    M4D_W6_STAMP(0)
    M4D_W6_BLOCK0(3, 0, 0, 7, !LAST)
    if (!LAST) { W6L_RAW(rnext_c, chunk & 1, 3); read_t(rnext, 1); }
    M4D_W6_BLOCK1(3, 0, 0, 6, !LAST)
    if (!LAST) { W6L_BDMA(wn + 3 * w_pos, 3); pin_t(1); }
    M4D_W6_BLOCK2(3, 0, 0, 6, !LAST)
    M4D_W6_STAMP(2)
    if (!LAST) W6L_WAIT(8);
    M4D_W6_STAMP(3)
    if (STAMPS) ++stq;
    if (!LAST) wc = wn;
  };
  chunk_body(0, m4d_int<1>{}, m4d_int<0>{});     // (a single-chunk layer, Cin = 16, ends here: with its surplus fetches, as before)
  if (n > 1) {
    for (int chunk = 1; chunk < last; ++chunk) chunk_body(chunk, m4d_int<0>{}, m4d_int<0>{});
    chunk_body(last, m4d_int<0>{}, m4d_int<1>{});
  }
  };
  if (half) k_loop(m4d_int<1>{}); else k_loop(m4d_int<2>{});
#undef M4D_W6_MFMA
#undef M4D_W6_MFMA0
#undef M4D_W6_BLOCK0
#undef M4D_W6_BLOCK1
#undef M4D_W6_BLOCK2
#undef M4D_W6_PIPE
#undef M4D_W6_STAMP
  M4D_W6_WAIT(0);                                  // no DMA may land in LDS once the epilogue reuses it
#undef M4D_W6_WAIT
  if (STAMPS && st && wv == 0) st[1024 + 1] = __builtin_readcyclecounter();
  __syncthreads();                                 // every wave is done with raw / the rings: the epilogue buffer aliases them

  // ---- output transform: rows of A^T (M A) through LDS per (N-tile, M-tile), then one 2x2-output item x 4 couts per thread.
  // Items of a full unit: 16 consecutive lanes = the 16 cout quads of ONE pixel (both N-tiles: 256 contiguous bytes per store
  // instruction and pixel, 4 pixels per instruction).  Rounds 2-4 walked the N-tiles one after the other: 8 lanes = 128 bytes
  // per pixel, 8 pixels per instruction -- the pattern a CU issues at 16 B/clk instead of ~100 (tools/micro/
  // store_issue_probe.hip: 4004 against 645 cycles for a unit's 64 KB).  N-tile 1's staging blocks are skewed by 32 floats:
  // the two cout halves of a pixel then sit in different LDS banks (one 16-lane read group covers 64 distinct banks).
  const int cq16 = half ? (t & 7) : (t & 15);      // this thread's cout quad of the unit (both of its items)
  const int co = ng * 64 + 4 * cq16;
  float bs[4];                                     // its bias quad: loaded here (not held across the K loop), the latency hides
#pragma unroll                                     // behind the LDS pass below
  for (int e = 0; e < 4; ++e) bs[e] = a.bias[min(co + e, a.Cout - 1)];
  constexpr int kMS = 36;                          // row stride (floats): 32 couts + 4 pad (16-byte aligned rows)
  constexpr int kRbMT = 4 * 2 * 32 * kMS;          // floats per (N-tile, M-tile): [4 rows i][2 k][32 tiles][kMS] = 36.9 KB
  constexpr int kRbSkew = 32;                      // floats: N-tile 1's blocks start half a bank row later
  float* Rb = lds;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    if (nt == 1 && half) break;                    // (a half unit never touched N-tile 1's accumulators)
    float* rbuf = Rb + (nt * 2 + mt) * kRbMT + nt * kRbSkew;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r];
      const int trow = (r & 3) + 8 * (r >> 2) + 4 * kh;
      rbuf[((pr * 2 + 0) * 32 + trow) * kMS + m] = (m0 + m1) + m2;
      rbuf[((pr * 2 + 1) * 32 + trow) * kMS + m] = (m1 - m2) - m3;
    }
  }
  __syncthreads();
  float* oimg = a.out + (long long)bi * a.h * a.w * a.Cout;
  const bool vec_ok = (a.Cout & 3) == 0;
  const bool whole = tile_x + kT <= a.w && tile_y + kT <= a.h;      // uniform: no per-store bounds tests on interior tiles
  const bool fast = whole && vec_ok && ng * 64 + (half ? 32 : 64) <= a.Cout;   // ... and every cout quad of the unit is real
  const int ont = cq16 >> 3, cq = cq16 & 7;        // N-tile and quad inside it
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    if (it == 1 && half) break;
    // full unit: item = (M-tile it, tile, cout quad of 16); half unit: (M-tile, tile, cout quad of 8), one item per thread
    const int tl = half ? (t >> 3) & 31 : (t >> 4) & 31, omt = half ? (t >> 8) : it;
    const float* rbuf = Rb + (ont * 2 + omt) * kRbMT + ont * kRbSkew;
    const int tg = omt * 32 + tl;                  // Winograd tile 0..63 of the workgroup (8 x 8)
    const int ty2 = tg >> 3, tx2 = tg & 7;
    float4 rv[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) rv[i][k] = *reinterpret_cast<const float4*>(rbuf + ((i * 2 + k) * 32 + tl) * kMS + 4 * cq);
    float y[2][2][4];                              // [column k][row l][cout]
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float* r0 = reinterpret_cast<const float*>(&rv[0][k]); const float* r1 = reinterpret_cast<const float*>(&rv[1][k]);
      const float* r2 = reinterpret_cast<const float*>(&rv[2][k]); const float* r3 = reinterpret_cast<const float*>(&rv[3][k]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v0 = ((r0[e] + r1[e]) + r2[e]) + bs[e];
        const float v1 = ((r1[e] - r2[e]) - r3[e]) + bs[e];
        y[k][0][e] = v0 > 0.f ? v0 : v0 * a.slope;
        y[k][1][e] = v1 > 0.f ? v1 : v1 * a.slope;
      }
    }
    const int ox = tile_x + 2 * tx2, oy = tile_y + 2 * ty2;
    float* op = oimg + ((long long)oy * a.w + ox) * a.Cout + co;
    if (fast) {
      // UNIFORM condition and nothing but 16-byte stores in this arm (inline asm): under the per-lane condition of rounds 2-3
      // hipcc merged this arm with the guarded one below and stored three of the four pixels as separate dwords -- 128-byte
      // runs in 4-byte pieces, the slowest pattern of tools/micro/store_issue_probe.hip
#pragma unroll
      for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int k = 0; k < 2; ++k) m4d_store16(op + ((long long)l * a.w + k) * a.Cout, y[k][l][0], y[k][l][1], y[k][l][2], y[k][l][3]);
    } else if (co < a.Cout) {
#pragma unroll
      for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (ox + k < a.w && oy + l < a.h) {
            float* o2 = op + ((long long)l * a.w + k) * a.Cout;
            if (vec_ok && co + 3 < a.Cout) m4d_store16(o2, y[k][l][0], y[k][l][1], y[k][l][2], y[k][l][3]);
            else { for (int e = 0; e < 4; ++e) if (co + e < a.Cout) o2[e] = y[k][l][e]; }
          }
    }
  }
  if (STAMPS && st && wv == 0) st[1024 + 2] = __builtin_readcyclecounter();
}

// Grids of at least this many (tile, 64-cout) units run on the persistent kernel (m4d_wino6p.hip): 8 per CU of the 256.  Measured
// (round 4, profiles/r04_wino6_persistent.txt): alone it is 1.05-1.07x on the level-1 layers at batch 1 (960 units), 1.09-1.21x
// at batch 8 (7680), 0.93-0.97x where a workgroup gets a single unit (its four-pass epilogue in two ring slots is slower than
// the one-shot epilogue); inside the frame pipeline the batch-1 step does not get shorter with it (1396 vs 1401 frames/s, 4
// interleaved runs: static unit ranges against the dispatcher's dynamic placement beside other frames' kernels), the batch-32
// step does (+3.6 %, 1988 -> 2060 frames/s).
constexpr long long kPersistentMinUnits = 2048;
unsigned long long* g_wino6_stamps = nullptr;       // profiling hook (m4d_wino6_set_stamps): process-wide, NULL = off
#if M4D_EXPERIMENTS
int g_wino6_half_max_wg = 0;                       // grids of at most this many workgroups take the half-tile kernel under variant 0 (default: none)
int g_wino6_variant = 0;                           // 0 / 1 = this file's kernel, 2 = the wide kernel (m4d_wino6w.hip), 3 = the half-tile kernel
#endif

}  // namespace

extern "C" void m4d_wino6_set_stamps(unsigned long long* device_buffer) { g_wino6_stamps = device_buffer; }

// ---- staggered first round ---------------------------------------------------------------------------------------------------
// A launch of this kernel puts ONE workgroup on every CU (154 KB of LDS) and all of them take the same time: the CUs free in
// lock step, once per unit time (17-27 us), and a small kernel of another hipGraph branch -- a coarse-level kernel of the NEXT
// frame, whose latency chain is on the critical path of a batch-1 step -- that becomes ready in between waits for that instant
// (measured: ~25 us per dependent launch).  With stagger_us > 0 (an ARGUMENT of
// m4d_conv3x3_wino6_bias_act_ks since ABI 6 -- round 5's process-wide setter is gone) the first 256 workgroups start in
// `stagger_phases` groups spread over `stagger_us`: the lock step is broken for the whole launch, CUs free every range / phases
// us.  Costs the launch ~range / 2 at most (level-1 layers +0..3 us each, tools/bench_wino6.py); results unchanged (a delay).
// Which launches carry it is the caller's policy (network.py: grids of >= 200 workgroups at batch <= 4).

#if M4D_EXPERIMENTS
// include/m4depth_hip_experiments.h: bit-identical alternatives, measured not faster end to end (DESIGN_HISTORY.md)
// m4d_wino6w.hip: 16x16 pixels x all 96 / 128 output channels per workgroup, two passes over the position rows
int m4d_wino6w_launch(const float* x, const void* wu6, const float* bias, int b, int h, int w, int Cin, int Cout, int CoutPad,
                      float slope, float* out, void* stream);
// m4d_wino6h.hip: 16x8 pixels x 64 couts per workgroup (half the tile: twice the workgroups) for small grids
int m4d_wino6h_launch(const float* x, const void* wu6, const float* bias, int b, int h, int w, int Cin, int Cout, int CoutPad,
                      float slope, float* out, void* stream);
extern "C" void m4d_wino6_set_variant(int variant) { g_wino6_variant = variant; }
extern "C" void m4d_wino6_set_half_tile_max_workgroups(int max_wg) { g_wino6_half_max_wg = max_wg; }
#endif

// m4d_wino6p.hip: the same arithmetic with persistent workgroups (one per CU) walking (tile, cout group) units; bit-identical
int m4d_wino6p_launch(const float* x, const void* wu6, const float* bias, int b, int h, int w, int Cin, int Cout, int CoutPad,
                      float slope, float* out, int units_per_wg, int stagger_us, int stagger_phases, void* stream);

extern "C" long long m4d_wino6_persistent_min_units(void) {
  static const long long v = [] { const char* e = getenv("M4D_WINO6P_MIN_UNITS"); const long long u = e ? atoll(e) : 0;
                                  return u > 0 ? u : kPersistentMinUnits; }();             // (env: measurement knob)
  return v;
}

extern "C" int m4d_conv3x3_wino6_bias_act(const float* x, const void* wu6, const float* bias, int b, int h, int w,
                                          int Cin, int Cout, int CoutPad, float slope, float* out, void* stream) {
  return m4d_conv3x3_wino6_bias_act_k(x, wu6, bias, b, h, w, Cin, Cout, CoutPad, slope, out, 0, stream);
}

extern "C" int m4d_conv3x3_wino6_bias_act_k(const float* x, const void* wu6, const float* bias, int b, int h, int w,
                                            int Cin, int Cout, int CoutPad, float slope, float* out, int kernel, void* stream) {
  return m4d_conv3x3_wino6_bias_act_ks(x, wu6, bias, b, h, w, Cin, Cout, CoutPad, slope, out, kernel, 0, 0, stream);
}

extern "C" int m4d_conv3x3_wino6_bias_act_ks(const float* x, const void* wu6, const float* bias, int b, int h, int w,
                                             int Cin, int Cout, int CoutPad, float slope, float* out, int kernel,
                                             int stagger_us, int stagger_phases, void* stream) {
  M4D_CHECK_ARG(x && wu6 && bias && out && b > 0 && h > 0 && w > 0 && Cin >= 16 && Cout > 0);
  M4D_CHECK_ARG(CoutPad % 64 == 0 && CoutPad >= Cout && Cin % 16 == 0);
  M4D_CHECK_ARG(((((uintptr_t)x) & 15u) == 0) && ((((uintptr_t)wu6) & 15u) == 0));
  M4D_CHECK_ARG((long long)h * w * Cin * 4 < (1ll << 31));                              // one image = one buffer descriptor
  // kernel = 16 + n (1 <= n <= 15): persistent workgroups of n consecutive units each, placed by the dispatcher
  const int units_per_wg = kernel >= 17 && kernel <= 31 ? kernel - 16 : 0;
  M4D_CHECK_ARG(((kernel >= 0 && kernel <= 2) || units_per_wg > 0) && ((kernel != 2 && units_per_wg == 0) || Cin >= 32));
  M4D_CHECK_ARG(stagger_us >= 0 && stagger_us <= 1000);
  M4D_CHECK_ARG(stagger_us == 0 || (stagger_phases >= 1 && stagger_phases <= 32 && !(stagger_phases & (stagger_phases - 1))));
  {
    // kernel 0 = by grid size: the persistent kernel (m4d_wino6p.hip) on grids of many units per CU -- there it saves the
    // per-unit prologue, fetches a tile's halo once for its cout groups and stores 256-byte runs; smaller grids keep this
    // file's kernel (a cheaper epilogue, and the dispatcher places its workgroups dynamically beside other frames' kernels)
    const long long units = (long long)b * ((w + kT - 1) / kT) * ((h + kT - 1) / kT) * (CoutPad / 64);
    const bool persistent = kernel == 2 || units_per_wg > 0 || (kernel == 0 && Cin >= 32 && units >= m4d_wino6_persistent_min_units());
    if (persistent && g_wino6_stamps == nullptr)
      return m4d_wino6p_launch(x, wu6, bias, b, h, w, Cin, Cout, CoutPad, slope, out, units_per_wg,
                               units_per_wg > 0 ? stagger_us : 0, stagger_phases, stream);
  }
#if M4D_EXPERIMENTS
  {
    const bool wide_ok = CoutPad == 128 && Cout > 64 && (Cout & 3) == 0 && ((((uintptr_t)bias) | ((uintptr_t)out)) & 15u) == 0 &&
                         g_wino6_stamps == nullptr;
    if (wide_ok && g_wino6_variant == 2)
      return m4d_wino6w_launch(x, wu6, bias, b, h, w, Cin, Cout, CoutPad, slope, out, stream);
    const long long wgs = (long long)b * ((w + kT - 1) / kT) * ((h + kT - 1) / kT) * (CoutPad / 64);
    if (g_wino6_stamps == nullptr && (g_wino6_variant == 3 || (g_wino6_variant == 0 && wgs <= g_wino6_half_max_wg)))
      return m4d_wino6h_launch(x, wu6, bias, b, h, w, Cin, Cout, CoutPad, slope, out, stream);
  }
#endif
  Wino6Args a;
  a.x = x; a.wu = reinterpret_cast<const unsigned char*>(wu6); a.bias = bias; a.out = out;
  a.b = b; a.h = h; a.w = w; a.Cin = Cin; a.Cout = Cout; a.CoutPad = CoutPad; a.n_chunks = Cin / 16; a.slope = slope;
  a.tiles_x = (w + kT - 1) / kT; a.tiles_y = (h + kT - 1) / kT;
  a.stamps = g_wino6_stamps;
  a.stagger = stagger_us * 100;                    // 100-MHz ticks (wall_clock64); the one-workgroup-per-unit kernel only
  a.phases = stagger_us > 0 ? stagger_phases : 1;
  constexpr size_t lds_epi = (size_t)(4 * 4 * 2 * 32 * 36 + 32) * sizeof(float);        // epilogue staging 147 KB (+ N-tile 1's skew)
  constexpr size_t lds_loop = (size_t)kBRingOff + 4 * kBRingBytes;                      // K loop: raw halo x 2 + fragment rings (154 KB)
  constexpr size_t lds = lds_epi > lds_loop ? lds_epi : lds_loop;
  static_assert(lds <= 160 * 1024, "LDS budget of one CU");
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * (CoutPad / 64)), (unsigned)b);
  // more than 64 KB of dynamic LDS needs the opt-in, per kernel
  if (a.stamps) M4D_LDS_OPT_IN(&conv3x3_wino6_kernel<true>);
  else M4D_LDS_OPT_IN(&conv3x3_wino6_kernel<false>);
  if (a.stamps) m4d_launch(conv3x3_wino6_kernel<true>, grid, dim3(512), lds, (hipStream_t)stream, a);
  else m4d_launch(conv3x3_wino6_kernel<false>, grid, dim3(512), lds, (hipStream_t)stream, a);
  return M4D_LAUNCH_RESULT();
}
