// Tail of a DepthEstimatorLevel in ONE kernel: the last two DispRefiner convolutions (32 -> 16 + leaky_relu, 16 -> 5;
// m4depth_network.py:109-135) and the "depth_estimator" glue (exp/clip, parallax2depth, state assign; :247-260).
//
// As separate launches these are three of the least efficient kernels of a level: 1.13 + 0.18 GFLOP at level 1 but 31 + 17 +
// 5 us (0.8 / 0.5 TB/s on 16- and 5-channel maps), and at the coarse levels three to five launches at the 4.6-10 us floor.
// Fused, the 32-channel input is read once, the 16- and 5-channel maps never touch HBM, and the level ends one launch after
// its last wide convolution.
//
// Workgroup = 4 waves = one 16x8 output tile.  conv6 is evaluated on the tile plus a one-pixel ring (18x10 = 180
// positions, 12 M-tiles of v_mfma_f32_16x16x4_f32, N = 16 exactly) from a 20x12 input halo in LDS; its output (+bias,
// leaky_relu, ZERO outside the image: it is conv7's zero padding) stays in LDS; conv7 (N = 5 of 16) runs on the 8 tile
// rows; the five outputs of a pixel meet in LDS and one lane per pixel finishes the level.
// K order inside the MFMAs: lane group kq = lane >> 4 owns 8 (conv6) / 4 (conv7) consecutive channels, so both operands
// are 16-byte LDS reads.  Deterministic.
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TailArgs {
  const float* x;                 // [b,h,w,32]
  const float* w6; const float* b6;   // [9][16][32], [16]
  const float* w7; const float* b7;   // [9][16][16] (rows 5..15 zero), [5]
  const float* rot; int rot_c; const float* trans; const float* cam_f; const float* cam_c;
  int h, w, tiles_x; float scale;
  float* parallax; float* depth; float* other; float* depth_state;
};

// (Measured: an 8 x 4-tile instantiation for the small maps of the coarse levels -- a quarter of the work per workgroup --
// changes nothing end to end: there the launch is bound by the staging round trip and the three barriers, not by the tile.)
// (Measured as well: the weights as register-resident B fragments read straight from global memory instead of staged in
// LDS -- 49 KB, three workgroups per CU, no weight staging in the prologue: 17.9 us average instead of 17.5, i.e. nothing.
// At level 1 the kernel is matrix-core bound at ~60 % (2.8 GFLOP with the ring recompute and conv7's 5-of-16 columns).)
constexpr int kTW = 16, kTH = 8;
constexpr int kXW = kTW + 4, kXH = kTH + 4, kXP = kXW * kXH;     // input halo 20 x 12 = 240
constexpr int kMW = kTW + 2, kMH = kTH + 2, kMP = kMW * kMH;     // conv6 positions 18 x 10 = 180
constexpr int kXS = 36, kMS = 20;                                // LDS row strides (floats): 32 + 4, 16 + 4
constexpr int kW6 = 9 * 16 * kXS, kW7 = 9 * 8 * kMS;               // w7: only 8 of the 16 packed cout rows are kept (5 used)

__global__ void __launch_bounds__(256, 2)
refiner_tail_kernel(const TailArgs a) {
  extern __shared__ __align__(16) float lds[];
  float* xin = lds;                       // [kXP][kXS]   34.6 KB   (later: out5 [128][8])
  float* mid = xin + kXP * kXS;           // [kMP][kMS]   14.4 KB
  float* w6 = mid + kMP * kMS;            // [9][16][kXS] 20.7 KB
  float* w7 = w6 + kW6;                   // [9][8][kMS]   5.8 KB  (75.5 KB in all: two workgroups per CU)

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bi = blockIdx.y;
  int blk = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) blk = (blk & 7) * (nb >> 3) + (blk >> 3);       // one band of tiles per XCD
  const int tile_y = (blk / a.tiles_x) * kTH, tile_x = (blk % a.tiles_x) * kTW;
  const float* ximg = a.x + (long long)bi * a.h * a.w * 32;

  // ---- stage A: input halo (zero outside the image) and both weight sets -> LDS.  Unconditional loads (clamped
  // coordinates + select): a branch around a load makes hipcc wait for it at the merge.
#pragma unroll
  for (int u = 0; u < (kXP * 8 + 255) / 256; ++u) {
    const int idx = u * 256 + t;
    const int hp = min(idx >> 3, kXP - 1), c4 = (idx & 7) * 4;
    const int gy = tile_y - 2 + hp / kXW, gx = tile_x - 2 + hp % kXW;
    const bool ok = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    const int cy = min(max(gy, 0), a.h - 1), cx = min(max(gx, 0), a.w - 1);
    float4 v = *reinterpret_cast<const float4*>(ximg + ((long long)cy * a.w + cx) * 32 + c4);
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < kXP * 8) *reinterpret_cast<float4*>(xin + hp * kXS + c4) = v;
  }
#pragma unroll
  for (int u = 0; u < (9 * 16 * 8 + 255) / 256; ++u) {               // w6: 1152 float4
    const int idx = u * 256 + t;
    if (idx < 9 * 16 * 8)
      *reinterpret_cast<float4*>(w6 + (idx >> 3) * kXS + (idx & 7) * 4) = *reinterpret_cast<const float4*>(a.w6 + idx * 4);
  }
#pragma unroll
  for (int u = 0; u < (9 * 8 * 4 + 255) / 256; ++u) {                // w7: rows 0..7 of every tap, 288 float4
    const int idx = u * 256 + t;
    if (idx < 9 * 8 * 4) {
      const int tap = idx >> 5, row = (idx >> 2) & 7, q = idx & 3;
      *reinterpret_cast<float4*>(w7 + (tap * 8 + row) * kMS + q * 4) = *reinterpret_cast<const float4*>(a.w7 + ((tap * 16 + row) * 4 + q) * 4);
    }
  }
  __syncthreads();

  const int li = lane & 15, kq = lane >> 4;
  // ---- stage B: conv6 (32 -> 16) + bias + leaky_relu on the 180 ring positions; wave w takes M-tiles 3w..3w+2
  {
    const float bias6 = a.b6[li];
#pragma unroll 1
    for (int mt = 0; mt < 3; ++mt) {
      const int mtile = wave * 3 + mt;
      const int mp = min(mtile * 16 + li, kMP - 1);                  // ring position of this lane's A row
      const int my = mp / kMW, mx = mp % kMW;
      const float* ap = xin + (my * kXW + mx) * kXS + kq * 8;
      const float* bp = w6 + li * kXS + kq * 8;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float* at = ap + ((tap / 3) * kXW + (tap % 3)) * kXS;
        const float4 a0 = *reinterpret_cast<const float4*>(at), a1 = *reinterpret_cast<const float4*>(at + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(bp + tap * 16 * kXS), b1 = *reinterpret_cast<const float4*>(bp + tap * 16 * kXS + 4);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], acc, 0, 0, 0);
      }
      // D: col = cout (li), row = position 4 * kq + r of the M-tile
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = mtile * 16 + 4 * kq + r;
        if (p < kMP) {
          const int gy = tile_y - 1 + p / kMW, gx = tile_x - 1 + p % kMW;
          float v = acc[r] + bias6;
          v = v > 0.f ? v : v * 0.1f;
          mid[p * kMS + li] = (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w) ? v : 0.f;      // conv7's zero padding
        }
      }
    }
  }
  __syncthreads();

  // ---- stage C: conv7 (16 -> 5 of 16) on the 8 tile rows; wave w takes rows 2w, 2w+1
  float* out5 = xin;                                                 // [128][8], the input halo is no longer needed
  {
#pragma unroll 1
    for (int mt = 0; mt < 2; ++mt) {
      const int row = wave * 2 + mt;
      const float* ap = mid + (row * kMW + li) * kMS + kq * 4;
      const float* bp = w7 + (li & 7) * kMS + kq * 4;                  // output columns 8..15 are never stored: they alias 0..7
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float4 a0 = *reinterpret_cast<const float4*>(ap + ((tap / 3) * kMW + (tap % 3)) * kMS);
        const float4 b0 = *reinterpret_cast<const float4*>(bp + tap * 8 * kMS);
        const float av[4] = {a0.x, a0.y, a0.z, a0.w};
        const float bv[4] = {b0.x, b0.y, b0.z, b0.w};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], acc, 0, 0, 0);
      }
      if (li < 5) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out5[(row * 16 + 4 * kq + r) * 8 + li] = acc[r];
      }
    }
  }
  __syncthreads();

  // ---- stage D: one lane per pixel: + bias, exp/clip/scale (:250), parallax2depth (:251), outputs (+ temporal state)
  if (t < kTW * kTH) {
    const int oy = tile_y + (t >> 4), ox = tile_x + (t & 15);
    if (oy < a.h && ox < a.w) {
      const float* o = out5 + t * 8;
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
}

}  // namespace

extern "C" int m4d_refiner_tail(const float* x32, const float* w6p, const float* b6, const float* w7p, const float* b7,
                                const float* rot, int rot_c, const float* trans, const float* cam_f, const float* cam_c,
                                int b, int h, int w, float scale, float* parallax, float* depth, float* other,
                                float* depth_state, void* stream) {
  M4D_CHECK_ARG(x32 && w6p && b6 && w7p && b7 && trans && cam_f && cam_c && parallax && depth && other);
  M4D_CHECK_ARG(b > 0 && h > 0 && w > 0 && (rot == nullptr || rot_c == 3 || rot_c == 4));
  M4D_CHECK_ARG(((((uintptr_t)x32 | (uintptr_t)w6p | (uintptr_t)w7p | (uintptr_t)other)) & 15u) == 0);
  TailArgs a;
  a.x = x32; a.w6 = w6p; a.b6 = b6; a.w7 = w7p; a.b7 = b7; a.rot = rot; a.rot_c = rot_c; a.trans = trans;
  a.cam_f = cam_f; a.cam_c = cam_c; a.h = h; a.w = w; a.scale = scale;
  a.parallax = parallax; a.depth = depth; a.other = other; a.depth_state = depth_state;
  a.tiles_x = (w + kTW - 1) / kTW;
  const int tiles = a.tiles_x * ((h + kTH - 1) / kTH);
  constexpr size_t lds = (size_t)(kXP * kXS + kMP * kMS + kW6 + kW7) * sizeof(float);          // 75.5 KB
  M4D_LDS_OPT_IN(&refiner_tail_kernel);
  m4d_launch(refiner_tail_kernel, dim3(tiles, b), dim3(256), lds, (hipStream_t)stream, a);
  return M4D_LAUNCH_RESULT();
}
