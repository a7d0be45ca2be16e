// Tail of a DepthEstimatorLevel in ONE kernel, in the bf16-split arithmetic (m4depth_network.py:109-135, 247-260): the last two
// DispRefiner convolutions (32 -> 16 + leaky_relu, 16 -> 5) on the BF16 matrix cores with float32 operands -- every float32
// value is the exact sum of three bf16 terms, six of the nine term products kept, float32 accumulation (m4d_wino6.hip) --
// followed by exp/clip, parallax2depth and the state assign.  m4d_tail.hip is the same tail on the fp32 matrix cores (exact
// fp32 products; what conv_arith = "f32" runs); there the two MFMA chains were 60 % of the kernel at level 1 (ablations,
// DESIGN.md section 6): 72 x v_mfma_f32_16x16x4_f32 (32 cycles each) per 16 positions x 16 channels of conv6 become 54 x
// v_mfma_f32_16x16x32_bf16 (16 cycles each).
//
// PERSISTENT workgroups (at most two per CU, 4 waves each) walk the 14x10-pixel output tiles of the whole batch:
//  * conv6's weights live in REGISTERS as B fragments (9 taps x 3 parts x 4 VGPRs, fetched once per workgroup, not per tile),
//    conv7's (8 of the 16 output rows, 5 used) in 7.5 KB of LDS;
//  * stage A: the 18x14 input halo is split into its three bf16 planes while it is committed to LDS (once per value; it then
//    meets 9 taps x 16 channels), layout [part][channel half][pixel][2 x 16 B]: pixel stride 32 B, so the 16-lane groups a
//    ds_read_b128 is serviced in (8 lanes at k-quarter a, 8 at a + 1: MI355X_MICROARCH.md, LDS) cover 16 distinct 16-byte
//    slots -- even slots for a, odd for a + 1 -- and a 16-pixel ring row IS one MFMA M-tile (that is why the tile is 14 wide);
//  * stage B: conv6 on the 16x12 ring of positions conv7 needs (wave w: ring rows 3w .. 3w+2, three independent accumulators
//    per tap), + bias, leaky_relu, ZERO outside the image (conv7's zero padding), split again, bf16 planes to LDS;
//  * stage C: conv7 with K = 32 = two taps x 16 channels per MFMA (5 K-steps, the tenth tap has zero weights);
//  * stage D: one lane per pixel finishes the level (identical code to m4d_tail.hip).
// Deterministic; LDS 75.3 KB.  (Measured: issuing the NEXT tile's halo loads before stage B -- 32 more live registers beside
// the 108 of conv6's weights -- spills 28 registers and is 1.5x slower, 696 against 457 us at level 1, batch 32.)
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

typedef float t6_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 t6_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned t6_u32x4 __attribute__((ext_vector_type(4)));

struct Tail6Args {
  const float* x;                        // [b,h,w,32]
  const unsigned char* w6; const float* b6;   // B fragments [9 taps][3 parts][64 lanes][8 bf16], [16]
  const unsigned char* w7; const float* b7;   // [5 K-steps][3 parts][4 k-quarters][8 rows][8 bf16], [5]
  const float* rot; int rot_c; const float* trans; const float* cam_f; const float* cam_c;
  int h, w, tiles_x, tiles_per_image, items; float scale;
  float* parallax; float* depth; float* other; float* depth_state;
};

constexpr int kTW = 14, kTH = 10;                                // output tile
constexpr int kXW = kTW + 4, kXH = kTH + 4, kXP = kXW * kXH;     // input halo 18 x 14 = 252 pixels
constexpr int kMW = kTW + 2;                                     // conv6 ring: 16 wide (one M-tile per ring row) x 12 rows
constexpr int kXPlane = 254 * 32;                                // bytes per (part, channel half) plane: 2032 dwords = 16 mod 32 (ds_write_b64 groups)
constexpr int kXPart = 2 * kXPlane;
constexpr int kMPart = 196 * 32;                                 // conv6 output: [part][position (192 + conv7's over-read)][2 x 16 B]
constexpr int kOffMid = 3 * kXPart;                              // 48768
constexpr int kOffW7 = kOffMid + 3 * kMPart;                     // 67584
constexpr int kLds = kOffW7 + 5 * 3 * 512;                       // 75264 B

__global__ void __launch_bounds__(256, 2)
refiner_tail6_kernel(const Tail6Args a) {
  extern __shared__ __align__(16) unsigned char lds[];
  unsigned char* xin = lds;
  unsigned char* mid = lds + kOffMid;
  unsigned char* w7 = lds + kOffW7;
  float* out5 = reinterpret_cast<float*>(lds);                   // [10 rows][16][8]: aliases the halo (free after stage B)

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, kq = lane >> 4;

  // ---- once per workgroup: conv6's B fragments -> registers, conv7's -> LDS
  t6_bf16x8 B6[9][3];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int p = 0; p < 3; ++p)
      B6[tap][p] = *reinterpret_cast<const t6_bf16x8*>(a.w6 + ((tap * 3 + p) * 64 + lane) * 16);
  for (int idx = t; idx < 5 * 3 * 32; idx += 256)
    *reinterpret_cast<float4*>(w7 + idx * 16) = *reinterpret_cast<const float4*>(a.w7 + idx * 16);
  const float bias6 = a.b6[li];
  // conv7: lane (li, kq) multiplies tap 2 j + (kq >> 1) (the tenth tap does not exist: zero weights, any valid address)
  int c7_off[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int tap = min(2 * j + (kq >> 1), 8);
    c7_off[j] = ((tap / 3) * kMW + (tap % 3) + li) * 32 + (kq & 1) * 16;
  }
  const int b7_off = (kq * 8 + (li & 7)) * 16;                   // output columns 8..15 are never stored: they alias 0..7

  for (int item = blockIdx.x; item < a.items; item += gridDim.x) {
    const int bi = item / a.tiles_per_image, tile = item - bi * a.tiles_per_image;
    const int tile_y = (tile / a.tiles_x) * kTH, tile_x = (tile % a.tiles_x) * kTW;
    const float* ximg = a.x + (long long)bi * a.h * a.w * 32;

    // ---- stage A: input halo (zero outside the image), split into three bf16 planes
    {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = u * 256 + t;
        const int hp = min(idx >> 3, kXP - 1), c4 = idx & 7;
        const int gy = tile_y - 2 + hp / kXW, gx = tile_x - 2 + hp % kXW;
        const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
        v[u] = *reinterpret_cast<const float4*>(ximg + ((long long)cy * a.w + cx) * 32 + c4 * 4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = u * 256 + t;
        const int hp = min(idx >> 3, kXP - 1), c4 = idx & 7;
        const int gy = tile_y - 2 + hp / kXW, gx = tile_x - 2 + hp % kXW;
        const bool ok = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
        const float x0 = ok ? v[u].x : 0.f, x1 = ok ? v[u].y : 0.f, x2 = ok ? v[u].z : 0.f, x3 = ok ? v[u].w : 0.f;
        unsigned h0, m0, l0, h1, m1, l1;
        m4d_split3_pair(x0, x1, h0, m0, l0);
        m4d_split3_pair(x2, x3, h1, m1, l1);
        if (idx < kXP * 8) {
          unsigned char* dst = xin + (c4 >> 2) * kXPlane + hp * 32 + (c4 & 3) * 8;
          *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(dst + kXPart) = make_uint2(m0, m1);
          *reinterpret_cast<uint2*>(dst + 2 * kXPart) = make_uint2(l0, l1);
        }
      }
    }
    __syncthreads();

    // ---- stage B: conv6 (32 -> 16) on ring rows 3 wave .. 3 wave + 2
    {
      t6_f32x4 acc[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = t6_f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned char* ap = xin + (kq >> 1) * kXPlane + ((3 * wave) * kXW + li) * 32 + (kq & 1) * 16;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        t6_bf16x8 A[3][3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            A[q][p] = *reinterpret_cast<const t6_bf16x8*>(ap + ((q + tap / 3) * kXW + (tap % 3)) * 32 + p * kXPart);
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q][0], B6[tap][2], acc[q], 0, 0, 0);   // small terms first
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q][2], B6[tap][0], acc[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q][1], B6[tap][1], acc[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q][0], B6[tap][1], acc[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q][1], B6[tap][0], acc[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q][0], B6[tap][0], acc[q], 0, 0, 0);
      }
      // D: column = cout (li), rows = ring columns 4 kq + r of ring row 3 wave + q
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int rr = 3 * wave + q;
        const int gy = tile_y - 1 + rr;
        const bool row_ok = gy >= 0 && gy < a.h;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gx = tile_x - 1 + 4 * kq + r;
          float s = acc[q][r] + bias6;
          s = s > 0.f ? s : s * 0.1f;
          v[r] = (row_ok && gx >= 0 && gx < a.w) ? s : 0.f;      // conv7's zero padding
        }
        unsigned short* mp = reinterpret_cast<unsigned short*>(mid + (rr * kMW + 4 * kq) * 32 + (li >> 3) * 16 + (li & 7) * 2);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          unsigned hh, mm, ll;
          m4d_split3_pair(v[2 * e], v[2 * e + 1], hh, mm, ll);
          mp[(2 * e) * 16] = (unsigned short)(hh & 0xffffu);               mp[(2 * e + 1) * 16] = (unsigned short)(hh >> 16);
          mp[(2 * e) * 16 + kMPart / 2] = (unsigned short)(mm & 0xffffu);   mp[(2 * e + 1) * 16 + kMPart / 2] = (unsigned short)(mm >> 16);
          mp[(2 * e) * 16 + kMPart] = (unsigned short)(ll & 0xffffu);       mp[(2 * e + 1) * 16 + kMPart] = (unsigned short)(ll >> 16);
        }
      }
    }
    __syncthreads();

    // ---- stage C: conv7 (16 -> 5 of 16) on output rows wave, wave + 4, wave + 8
    {
      auto rows = [&](auto NR) __attribute__((always_inline)) {
        constexpr int nr = decltype(NR)::value;
        t6_f32x4 acc[nr];
#pragma unroll
        for (int q = 0; q < nr; ++q) acc[q] = t6_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          t6_bf16x8 Bv[3], A[nr][3];
#pragma unroll
          for (int p = 0; p < 3; ++p) Bv[p] = *reinterpret_cast<const t6_bf16x8*>(w7 + (j * 3 + p) * 512 + b7_off);
#pragma unroll
          for (int q = 0; q < nr; ++q)
#pragma unroll
            for (int p = 0; p < 3; ++p)
              A[q][p] = *reinterpret_cast<const t6_bf16x8*>(mid + p * kMPart + (wave + 4 * q) * kMW * 32 + c7_off[j]);
#pragma unroll
          for (int q = 0; q < nr; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q][0], Bv[2], acc[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < nr; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q][2], Bv[0], acc[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < nr; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q][1], Bv[1], acc[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < nr; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q][0], Bv[1], acc[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < nr; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q][1], Bv[0], acc[q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < nr; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q][0], Bv[0], acc[q], 0, 0, 0);
        }
        if (li < 5) {
#pragma unroll
          for (int q = 0; q < nr; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) out5[((wave + 4 * q) * 16 + 4 * kq + r) * 8 + li] = acc[q][r];
        }
      };
      if (wave < 2) rows(std::integral_constant<int, 3>{});
      else rows(std::integral_constant<int, 2>{});
    }
    __syncthreads();

    // ---- stage D: one lane per pixel: + bias, exp/clip/scale (:250), parallax2depth (:251), outputs (+ temporal state)
    if (t < kTW * kTH) {
      const int py = t / kTW, pxl = t - py * kTW;
      const int oy = tile_y + py, ox = tile_x + pxl;
      if (oy < a.h && ox < a.w) {
        const float* o = out5 + (py * 16 + pxl) * 8;
        const float r0 = o[0] + a.b7[0];
        const M4dMotion m = m4d_load_motion(a.rot, a.rot_c, a.trans, a.cam_f, a.cam_c, bi);
        const float para = expf(fminf(fmaxf(r0, -7.0f), 7.0f)) / a.scale;
        const M4dPixel px = m4d_pixel_factors(m, ox, oy);
        const float d = (px.s / para - m.tz) / px.alpha;
        const long long gp = ((long long)bi * a.h + oy) * a.w + ox;
        a.parallax[gp] = para;
        a.depth[gp] = d;
        if (a.depth_state) a.depth_state[gp] = d;
        *reinterpret_cast<float4*>(a.other + gp * 4) = make_float4(o[1] + a.b7[1], o[2] + a.b7[2], o[3] + a.b7[3], o[4] + a.b7[4]);
      }
    }
    __syncthreads();                                             // out5 aliases the next tile's halo
  }
}

}  // namespace

// include/m4depth_hip.h: m4d_refiner_tail6
extern "C" int m4d_refiner_tail6(const float* x32, const void* w6f, const float* b6, const void* w7f, const float* b7,
                                 const float* rot, int rot_c, const float* trans, const float* cam_f, const float* cam_c,
                                 int b, int h, int w, float scale, float* parallax, float* depth, float* other,
                                 float* depth_state, void* stream) {
  M4D_CHECK_ARG(x32 && w6f && b6 && w7f && b7 && trans && cam_f && cam_c && parallax && depth && other);
  M4D_CHECK_ARG(b > 0 && h > 0 && w > 0 && (rot == nullptr || rot_c == 3 || rot_c == 4));
  M4D_CHECK_ARG(((((uintptr_t)x32 | (uintptr_t)w6f | (uintptr_t)w7f | (uintptr_t)other)) & 15u) == 0);
  Tail6Args a;
  a.x = x32; a.w6 = reinterpret_cast<const unsigned char*>(w6f); a.b6 = b6; a.w7 = reinterpret_cast<const unsigned char*>(w7f); a.b7 = b7;
  a.rot = rot; a.rot_c = rot_c; a.trans = trans; a.cam_f = cam_f; a.cam_c = cam_c; a.h = h; a.w = w; a.scale = scale;
  a.parallax = parallax; a.depth = depth; a.other = other; a.depth_state = depth_state;
  a.tiles_x = (w + kTW - 1) / kTW;
  a.tiles_per_image = a.tiles_x * ((h + kTH - 1) / kTH);
  const long long items = (long long)a.tiles_per_image * b;
  M4D_CHECK_ARG(items < (1ll << 31));
  a.items = (int)items;
  M4D_LDS_OPT_IN_BYTES(kLds, &refiner_tail6_kernel);
  const int max_wg = 2 * m4d_device_cus();
  m4d_launch(refiner_tail6_kernel, dim3((unsigned)(items < max_wg ? items : max_wg)), dim3(256), (size_t)kLds, (hipStream_t)stream, a);
  return M4D_LAUNCH_RESULT();
}
