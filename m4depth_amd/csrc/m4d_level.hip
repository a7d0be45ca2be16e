// DepthEstimatorLevel glue (m4depth_network.py:179-204, 218, 224-227, 247-260):
// per-cut L2 normalisation, the TF-v1 / nearest resizes, and the two fused
// elementwise stages on either side of the refiner convolutions.  All of these are
// one-pass, one-lane-per-pixel (or per element) streaming kernels: HBM-bound by
// construction, no re-reads.
#include <cstdlib>
#include "m4d_common.h"
#include "m4d_level_pre.h"
#include "../../include/m4depth_hip.h"

namespace {

using m4d_level::LevelPreArgs;
using m4d_level::level_pre_body;
using m4d_level::normalize_cuts_group_body;

// ---- per-cut normalisation (:179-189) -------------------------------------------
template <bool VEC4>
__global__ void __launch_bounds__(256)
normalize_cuts_kernel(const float* __restrict__ x, long long items, int C, int k, int nc,
                      float* __restrict__ out) {
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items;
       it += (long long)gridDim.x * blockDim.x) {
    const float* p = x + it * nc;             // (pixel, cut) runs are contiguous: it = pix*k + kk
    float* o = out + it * nc;
    float acc = 0.f;
    if (VEC4) {
      for (int c = 0; c < nc; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + c);
        if (c == 0) acc = v.x * v.x; else acc = acc + v.x * v.x;
        acc = acc + v.y * v.y; acc = acc + v.z * v.z; acc = acc + v.w * v.w;
      }
      const float nrm = sqrtf(acc);
      for (int c = 0; c < nc; c += 4) {
        float4 v = *reinterpret_cast<const float4*>(p + c);
        v.x = v.x / nrm; v.y = v.y / nrm; v.z = v.z / nrm; v.w = v.w / nrm;
        *reinterpret_cast<float4*>(o + c) = v;
      }
    } else {
      for (int c = 0; c < nc; ++c) { const float v = p[c]; if (c == 0) acc = v * v; else acc = acc + v * v; }
      const float nrm = sqrtf(acc);
      for (int c = 0; c < nc; ++c) o[c] = p[c] / nrm;
    }
  }
}

template <int LPG>
__global__ void __launch_bounds__(256)
normalize_cuts_group_kernel(const float* __restrict__ x, long long items, float* __restrict__ out) {
  normalize_cuts_group_body<LPG>(x, items, out, blockIdx.x);
}

template <int LPG>
void launch_normalize_group(const float* x, long long items, float* out, hipStream_t s) {
  constexpr int GPW = 64 / LPG;
  const long long waves = (items + GPW - 1) / GPW;
  const long long blocks = (waves + 3) / 4;
  m4d_launch(normalize_cuts_group_kernel<LPG>, dim3((unsigned)blocks), dim3(256), 0, s, x, items, out);
}

// ---- tf.compat.v1.image.resize_bilinear, legacy coordinates (:202-204): ResizeAxis / resize_axis / resize_sample live in
// m4d_common.h (shared with the fused level front, m4d_front.hip)
__global__ void __launch_bounds__(256)
resize_bilinear_v1_kernel(const float* __restrict__ x, int ih, int iw, int c, int oh, int ow,
                          float mul, long long total, float* __restrict__ out) {
  const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(idx % c);
    long long r = idx / c;
    const int ox = (int)(r % ow); r /= ow;
    const int oy = (int)(r % oh);
    const long long bi = r / oh;
    const ResizeAxis ya = resize_axis(oy, sy, ih), xa = resize_axis(ox, sx, iw);
    out[idx] = resize_sample(x + bi * ih * iw * c, iw, c, cc, ya, xa) * mul;
  }
}

__global__ void __launch_bounds__(256)
resize_nearest_kernel(const float* __restrict__ x, int ih, int iw, int c, int oh, int ow,
                      long long total, float* __restrict__ out) {
  const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(idx % c);
    long long r = idx / c;
    const int ox = (int)(r % ow); r /= ow;
    const int oy = (int)(r % oh);
    const long long bi = r / oh;
    const int iy = min((int)floorf(((float)oy + 0.5f) * sy), ih - 1);
    const int ix = min((int)floorf(((float)ox + 0.5f) * sx), iw - 1);
    out[idx] = x[((bi * ih + iy) * iw + ix) * c + cc];
  }
}

__global__ void __launch_bounds__(256)
level_pre_kernel(const LevelPreArgs a) { level_pre_body(a, blockIdx.x, blockIdx.y, gridDim.x); }

// The level's two independent opening kernels in ONE launch (one boundary less on the coarse-level latency chain): the
// first pre_gx * b workgroups run level_pre, the rest the per-cut normalisation of the current features.
template <int LPG>
__global__ void __launch_bounds__(256)
level_pre_normalize_kernel(const LevelPreArgs a, int pre_gx, int pre_blocks, const float* __restrict__ nx, long long items,
                           float* __restrict__ nout) {
  const int blk = blockIdx.x;
  if (blk < pre_blocks) level_pre_body(a, blk % pre_gx, blk / pre_gx, pre_gx);
  else normalize_cuts_group_body<LPG>(nx, items, nout, blk - pre_blocks);
}

// ---- the reset frame of a whole pyramid in ONE launch ------------------------------
// On a new trajectory every level only seeds its memories (m4depth_network.py:207-214): prev_f_maps := the per-cut normalised
// features, depth_prev_t := 1000, and the level's estimate is the x2 upsampling of the coarser level's -- which starts from the
// constants of :198-200 at the coarsest level, and resize_sample of a constant map is that constant exactly (tl + (tr - tl) * a),
// so level l's estimate is the constant map (parallax 2^(levels below it), depth 1000, other 0) whatever the sizes.  Six dependent
// launches of ~5 us each (the upsampling chain) on the critical path of a batch-1 step become one.
constexpr int kResetMaxLevels = 8;
struct PyramidResetArgs {
  const float* features[kResetMaxLevels]; float* state_f[kResetMaxLevels]; float* depth_state[kResetMaxLevels];
  float* parallax[kResetMaxLevels]; float* depth[kResetMaxLevels]; float* other[kResetMaxLevels];
  long long pixels[kResetMaxLevels];            // b * h * w
  long long items[kResetMaxLevels];             // b * h * w * cuts
  int fill_blocks[kResetMaxLevels], first_block[kResetMaxLevels + 1], lpg[kResetMaxLevels];
  float parallax_value[kResetMaxLevels];
  int n_levels;
};

__global__ void __launch_bounds__(256)
pyramid_reset_kernel(const PyramidResetArgs a) {
  int l = 0;
  while (l + 1 < a.n_levels && (int)blockIdx.x >= a.first_block[l + 1]) ++l;      // block-uniform
  const int blk = (int)blockIdx.x - a.first_block[l];
  if (blk < a.fill_blocks[l]) {
    const float pv = a.parallax_value[l];
    float* __restrict__ para = a.parallax[l]; float* __restrict__ depth = a.depth[l]; float* __restrict__ other = a.other[l];
    float* __restrict__ dstate = a.depth_state[l];
    for (long long p = (long long)blk * 256 + threadIdx.x; p < a.pixels[l]; p += (long long)a.fill_blocks[l] * 256) {
      para[p] = pv; depth[p] = 1000.0f;
      *reinterpret_cast<float4*>(other + p * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      dstate[p] = 1000.0f;
    }
    return;
  }
  const long long nb = blk - a.fill_blocks[l];
  switch (a.lpg[l]) {
    case 2: normalize_cuts_group_body<2>(a.features[l], a.items[l], a.state_f[l], nb); break;
    case 4: normalize_cuts_group_body<4>(a.features[l], a.items[l], a.state_f[l], nb); break;
    case 6: normalize_cuts_group_body<6>(a.features[l], a.items[l], a.state_f[l], nb); break;
    default: normalize_cuts_group_body<8>(a.features[l], a.items[l], a.state_f[l], nb); break;
  }
}

// ---- the per-cut normalisation of SEVERAL feature maps in one launch ----------------------------------------------------------
// The normalised features of a frame depend on the encoder alone, not on the decoder's state: the coarse levels' maps (all frames
// of an encoder batch) are normalised right behind the encoder, off the levels' coarse-to-fine latency chains, and a coarse level
// then opens with ONE launch (m4d_level_front_small).  Same lane layout and summation chain as m4d_normalize_cuts: the same bits.
struct NormLevelsArgs {
  const float* x[kResetMaxLevels]; float* out[kResetMaxLevels]; long long items[kResetMaxLevels];
  int first_block[kResetMaxLevels + 1], lpg[kResetMaxLevels];
  int n;
};

__global__ void __launch_bounds__(256)
normalize_levels_kernel(const NormLevelsArgs a) {
  int l = 0;
  while (l + 1 < a.n && (int)blockIdx.x >= a.first_block[l + 1]) ++l;             // block-uniform
  const long long nb = (int)blockIdx.x - a.first_block[l];
  switch (a.lpg[l]) {
    case 2: normalize_cuts_group_body<2>(a.x[l], a.items[l], a.out[l], nb); break;
    case 4: normalize_cuts_group_body<4>(a.x[l], a.items[l], a.out[l], nb); break;
    case 6: normalize_cuts_group_body<6>(a.x[l], a.items[l], a.out[l], nb); break;
    default: normalize_cuts_group_body<8>(a.x[l], a.items[l], a.out[l], nb); break;
  }
}

extern "C" int m4d_normalize_levels(const m4d_norm_level* levels, int n_levels, void* stream) {
  M4D_CHECK_ARG(levels && n_levels > 0 && n_levels <= kResetMaxLevels);
  NormLevelsArgs a;
  long long blocks = 0;
  for (int l = 0; l < n_levels; ++l) {
    const m4d_norm_level& v = levels[l];
    M4D_CHECK_ARG(v.x && v.out && v.x != v.out && v.pixels > 0 && m4d_pyramid_reset_supported(v.C, v.nbre_cuts));
    M4D_CHECK_ARG(((((uintptr_t)v.x | (uintptr_t)v.out)) & 15u) == 0);
    a.x[l] = v.x; a.out[l] = v.out; a.items[l] = v.pixels * v.nbre_cuts; a.lpg[l] = v.C / v.nbre_cuts / 4;
    const long long gpw = 64 / a.lpg[l], waves = (a.items[l] + gpw - 1) / gpw, nblocks = (waves + 3) / 4;
    M4D_CHECK_ARG(blocks + nblocks < (1ll << 30));
    a.first_block[l] = (int)blocks;
    blocks += nblocks;
  }
  for (int l = n_levels; l <= kResetMaxLevels; ++l) a.first_block[l] = (int)blocks;
  for (int l = n_levels; l < kResetMaxLevels; ++l) { a.x[l] = nullptr; a.out[l] = nullptr; a.items[l] = 0; a.lpg[l] = 8; }
  a.n = n_levels;
  m4d_launch(normalize_levels_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_pyramid_reset_supported(int C, int nbre_cuts) {
  if (C <= 0 || nbre_cuts <= 0 || C % nbre_cuts != 0) return 0;
  const int nc = C / nbre_cuts;
  return (nc == 8 || nc == 16 || nc == 24 || nc == 32) ? 1 : 0;
}

extern "C" int m4d_pyramid_reset(const m4d_reset_level* levels, int n_levels, int b, void* stream) {
  M4D_CHECK_ARG(levels && n_levels > 0 && n_levels <= kResetMaxLevels && b > 0);
  PyramidResetArgs a;
  int blocks = 0;
  for (int l = 0; l < n_levels; ++l) {
    const m4d_reset_level& v = levels[l];
    M4D_CHECK_ARG(v.features && v.state_features && v.depth_state && v.parallax && v.depth && v.other && v.h > 0 && v.w > 0);
    M4D_CHECK_ARG(m4d_pyramid_reset_supported(v.C, v.nbre_cuts));
    M4D_CHECK_ARG(((((uintptr_t)v.features | (uintptr_t)v.state_features | (uintptr_t)v.other)) & 15u) == 0);
    M4D_CHECK_ARG(v.features != v.state_features);
    a.features[l] = v.features; a.state_f[l] = v.state_features; a.depth_state[l] = v.depth_state;
    a.parallax[l] = v.parallax; a.depth[l] = v.depth; a.other[l] = v.other; a.parallax_value[l] = v.parallax_value;
    a.pixels[l] = (long long)b * v.h * v.w;
    a.items[l] = a.pixels[l] * v.nbre_cuts;
    a.lpg[l] = v.C / v.nbre_cuts / 4;
    long long fb = (a.pixels[l] + 1023) / 1024;                                   // four pixels per thread
    if (fb > 1024) fb = 1024;
    a.fill_blocks[l] = (int)fb;
    const long long gpw = 64 / a.lpg[l], waves = (a.items[l] + gpw - 1) / gpw, nblocks = (waves + 3) / 4;
    M4D_CHECK_ARG(blocks + fb + nblocks < (1ll << 30));
    a.first_block[l] = blocks;
    blocks += (int)(fb + nblocks);
  }
  for (int l = n_levels; l <= kResetMaxLevels; ++l) a.first_block[l] = blocks;
  for (int l = n_levels; l < kResetMaxLevels; ++l) { a.fill_blocks[l] = 0; a.lpg[l] = 8; a.items[l] = 0; a.pixels[l] = 0; }
  a.n_levels = n_levels;
  m4d_launch(pyramid_reset_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return M4D_LAUNCH_RESULT();
}

// ---- fused "depth_estimator" tail (:247-260) --------------------------------------
__global__ void __launch_bounds__(256)
level_post_kernel(const float* __restrict__ ro, const float* __restrict__ rot, int rot_c,
                  const float* __restrict__ trans, const float* __restrict__ cam_f,
                  const float* __restrict__ cam_c, int h, int w, float scale,
                  float* __restrict__ parallax, float* __restrict__ depth, float* __restrict__ other,
                  float* __restrict__ depth_state) {
  const int bi = blockIdx.y;
  const M4dMotion m = m4d_load_motion(rot, rot_c, trans, cam_f, cam_c, bi);
  const int hw = h * w;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
    const int i = p % w, j = p / w;
    const long long gp = (long long)bi * hw + p;
    const float* r = ro + gp * 5;
    const float para = expf(fminf(fmaxf(r[0], -7.0f), 7.0f)) / scale;                 // :250
    const M4dPixel px = m4d_pixel_factors(m, i, j);
    const float d = (px.s / para - m.tz) / px.alpha;                                  // :251
    parallax[gp] = para;
    depth[gp] = d;
    if (depth_state) depth_state[gp] = d;
    *reinterpret_cast<float4*>(other + gp * 4) = make_float4(r[1], r[2], r[3], r[4]);
  }
}

// ---- convolution epilogue: + bias, leaky_relu (m4depth_network.py:84,87,123,133) -----
// PyTorch-ROCm runs a MIOpen convolution, a broadcast bias add and the activation as three
// kernels (three passes over the activation); this is one in-place pass.
template <bool VEC4>
__global__ void __launch_bounds__(256)
bias_act_kernel(const float* __restrict__ x, const float* __restrict__ bias, long long total, int C,
                float slope, float* __restrict__ out) {
  if (VEC4) {
    const int c4n = C >> 2;
    for (long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i4 < (total >> 2);
         i4 += (long long)gridDim.x * blockDim.x) {
      const int c = (int)(i4 % c4n) << 2;
      float4 v = *reinterpret_cast<const float4*>(x + (i4 << 2));
      const float4 b = *reinterpret_cast<const float4*>(bias + c);
      v.x = v.x + b.x; v.y = v.y + b.y; v.z = v.z + b.z; v.w = v.w + b.w;
      v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
      v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
      *reinterpret_cast<float4*>(out + (i4 << 2)) = v;
    }
  } else {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
      float v = x[i] + bias[i % C];
      out[i] = v > 0.f ? v : v * slope;
    }
  }
}

// Same epilogue, written into a zero-bordered [b,out_h,out_w,C] buffer at (off_y, off_x): the
// TF 'SAME' padding of the stride-2 convolution that follows (bottom/right only for even sizes)
// then costs nothing -- no separate pad/copy pass over the activation.
__global__ void __launch_bounds__(256)
bias_act_padded_kernel(const float* __restrict__ x, const float* __restrict__ bias, int h, int w, int C, float slope,
                       float* __restrict__ out, int out_h, int out_w, int off_y, int off_x, long long total4) {
  const int c4n = C >> 2;
  for (long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4;
       i4 += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i4 % c4n) << 2;
    long long p = i4 / c4n;
    const int i = (int)(p % w); p /= w;
    const int j = (int)(p % h);
    const long long bi = p / h;
    float4 v = *reinterpret_cast<const float4*>(x + (i4 << 2));
    const float4 bb = *reinterpret_cast<const float4*>(bias + c);
    v.x = v.x + bb.x; v.y = v.y + bb.y; v.z = v.z + bb.z; v.w = v.w + bb.w;
    v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
    v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
    *reinterpret_cast<float4*>(out + ((bi * out_h + j + off_y) * out_w + i + off_x) * C + c) = v;
  }
}

// level-local intrinsics camera / 2^(l+1) for every level in one launch (m4depth_network.py:300-302; x * 2^-k is exactly
// x / 2^k in float32)
__global__ void camera_pyramid_kernel(const float* __restrict__ f, const float* __restrict__ c, int n, int levels,
                                      float* __restrict__ f_out, float* __restrict__ c_out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * levels) return;
  const int l = idx / n, e = idx - l * n;
  const float sc = 1.0f / (float)(1u << (l + 1));
  f_out[idx] = f[e] * sc;
  c_out[idx] = c[e] * sc;
}

inline int grid1d(long long total) {
  long long g = (total + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  return (int)(g > 0 ? g : 1);
}

}  // namespace

extern "C" int m4d_normalize_cuts(const float* x, int b, int h, int w, int C, int nbre_cuts,
                                  float* out, void* stream) {
  M4D_CHECK_ARG(x && out && b > 0 && h > 0 && w > 0 && C > 0 && nbre_cuts > 0 && C % nbre_cuts == 0);
  const int nc = C / nbre_cuts;
  const long long items = (long long)b * h * w * nbre_cuts;
  const bool vec = (nc % 4 == 0) && ((((uintptr_t)x | (uintptr_t)out) & 15u) == 0);
  static int grouped = -1;                     // M4D_NORMALIZE_GROUPED=0: the one-lane-per-run kernel (same bits)
  if (grouped < 0) { const char* e = getenv("M4D_NORMALIZE_GROUPED"); grouped = e ? atoi(e) : 1; }
  if (vec && grouped && items < (1LL << 40)) {
    hipStream_t s = (hipStream_t)stream;
    switch (nc) {
      case 8: launch_normalize_group<2>(x, items, out, s); return M4D_LAUNCH_RESULT();
      case 16: launch_normalize_group<4>(x, items, out, s); return M4D_LAUNCH_RESULT();
      case 24: launch_normalize_group<6>(x, items, out, s); return M4D_LAUNCH_RESULT();
      case 32: launch_normalize_group<8>(x, items, out, s); return M4D_LAUNCH_RESULT();
      default: break;
    }
  }
  if (vec) m4d_launch(normalize_cuts_kernel<true>, dim3(grid1d(items)), dim3(256), 0, (hipStream_t)stream, x, items, C, nbre_cuts, nc, out);
  else m4d_launch(normalize_cuts_kernel<false>, dim3(grid1d(items)), dim3(256), 0, (hipStream_t)stream, x, items, C, nbre_cuts, nc, out);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_resize_bilinear_v1(const float* x, int b, int ih, int iw, int c, int oh, int ow,
                                      float mul, float* out, void* stream) {
  M4D_CHECK_ARG(x && out && b > 0 && ih > 0 && iw > 0 && c > 0 && oh > 0 && ow > 0);
  const long long total = (long long)b * oh * ow * c;
  m4d_launch(resize_bilinear_v1_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream,
                     x, ih, iw, c, oh, ow, mul, total, out);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_resize_nearest(const float* x, int b, int ih, int iw, int c, int oh, int ow,
                                  float* out, void* stream) {
  M4D_CHECK_ARG(x && out && b > 0 && ih > 0 && iw > 0 && c > 0 && oh > 0 && ow > 0);
  const long long total = (long long)b * oh * ow * c;
  m4d_launch(resize_nearest_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream,
                     x, ih, iw, c, oh, ow, total, out);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_level_pre_normalize(const float* prev_l_depth, const float* prev_l_parallax, const float* prev_l_other,
                                       int ph, int pw, const float* depth_prev_t, const float* trans,
                                       const float* cam_f, const float* cam_c, int b, int h, int w,
                                       float* para_prev_l, float* depth_prev_l, float* other_prev_l, float* para_prev_t,
                                       float* f_input, int f_stride, int log_off, int other_off, float log_scale,
                                       float* depth_state_reset,
                                       const float* norm_x, int C, int nbre_cuts, float* norm_out, void* stream) {
  M4D_CHECK_ARG(b > 0 && h > 0 && w > 0);
  M4D_CHECK_ARG(norm_x && norm_out && C > 0 && nbre_cuts > 0 && C % nbre_cuts == 0);
  const bool any_prev = prev_l_depth || prev_l_parallax || prev_l_other;
  if (any_prev) M4D_CHECK_ARG(prev_l_depth && prev_l_parallax && prev_l_other && ph > 0 && pw > 0);
  if (depth_prev_t) M4D_CHECK_ARG(trans && cam_f && cam_c && para_prev_t);
  M4D_CHECK_ARG(!(depth_state_reset && depth_state_reset == depth_prev_t));
  if (f_input) M4D_CHECK_ARG(f_stride > 0 && log_off >= 0 && log_off < f_stride && other_off + 4 <= f_stride);
  if (other_prev_l) M4D_CHECK_ARG((((uintptr_t)other_prev_l) & 15u) == 0);
  LevelPreArgs a;
  a.pl_depth = prev_l_depth; a.pl_para = prev_l_parallax; a.pl_other = prev_l_other; a.ph = ph; a.pw = pw;
  a.depth_prev_t = depth_prev_t; a.trans = trans; a.cam_f = cam_f; a.cam_c = cam_c; a.h = h; a.w = w;
  a.para_prev_l = para_prev_l; a.depth_prev_l = depth_prev_l; a.other_prev_l = other_prev_l; a.para_prev_t = para_prev_t;
  a.f_input = f_input; a.f_stride = f_stride; a.log_off = log_off; a.other_off = other_off; a.log_scale = log_scale;
  a.depth_state_reset = depth_state_reset;
  int gx = m4d_blocks((long long)h * w, 256);
  if (gx > 4096) gx = 4096;
  const int nc = C / nbre_cuts;
  const long long items = (long long)b * h * w * nbre_cuts;
  const bool vec = ((((uintptr_t)norm_x | (uintptr_t)norm_out) & 15u) == 0);
  const int lpg = nc / 4;
  if (vec && (nc == 8 || nc == 16 || nc == 24 || nc == 32)) {
    const long long gpw = 64 / lpg, waves = (items + gpw - 1) / gpw, nblocks = (waves + 3) / 4;
    const int pre_blocks = gx * b;
    const dim3 grid((unsigned)(pre_blocks + nblocks));
    hipStream_t s = (hipStream_t)stream;
    switch (lpg) {
      case 2: m4d_launch(level_pre_normalize_kernel<2>, grid, dim3(256), 0, s, a, gx, pre_blocks, norm_x, items, norm_out); break;
      case 4: m4d_launch(level_pre_normalize_kernel<4>, grid, dim3(256), 0, s, a, gx, pre_blocks, norm_x, items, norm_out); break;
      case 6: m4d_launch(level_pre_normalize_kernel<6>, grid, dim3(256), 0, s, a, gx, pre_blocks, norm_x, items, norm_out); break;
      default: m4d_launch(level_pre_normalize_kernel<8>, grid, dim3(256), 0, s, a, gx, pre_blocks, norm_x, items, norm_out); break;
    }
    return M4D_LAUNCH_RESULT();
  }
  m4d_launch(level_pre_kernel, dim3(gx, b), dim3(256), 0, (hipStream_t)stream, a);      // other channel counts: two launches
  const int rc = M4D_LAUNCH_RESULT();
  if (rc != 0) return rc;
  return m4d_normalize_cuts(norm_x, b, h, w, C, nbre_cuts, norm_out, stream);
}

extern "C" int m4d_level_pre(const float* prev_l_depth, const float* prev_l_parallax, const float* prev_l_other,
                             int ph, int pw, const float* depth_prev_t, const float* trans,
                             const float* cam_f, const float* cam_c, int b, int h, int w,
                             float* para_prev_l, float* depth_prev_l, float* other_prev_l, float* para_prev_t,
                             float* f_input, int f_stride, int log_off, int other_off, float log_scale,
                             float* depth_state_reset, void* stream) {
  M4D_CHECK_ARG(b > 0 && h > 0 && w > 0);
  const bool any_prev = prev_l_depth || prev_l_parallax || prev_l_other;
  if (any_prev) M4D_CHECK_ARG(prev_l_depth && prev_l_parallax && prev_l_other && ph > 0 && pw > 0);
  if (depth_prev_t) M4D_CHECK_ARG(trans && cam_f && cam_c && para_prev_t);
  M4D_CHECK_ARG(!(depth_state_reset && depth_state_reset == depth_prev_t));
  if (f_input) M4D_CHECK_ARG(f_stride > 0 && log_off >= 0 && log_off < f_stride && other_off + 4 <= f_stride);
  if (other_prev_l) M4D_CHECK_ARG((((uintptr_t)other_prev_l) & 15u) == 0);
  LevelPreArgs a;
  a.pl_depth = prev_l_depth; a.pl_para = prev_l_parallax; a.pl_other = prev_l_other; a.ph = ph; a.pw = pw;
  a.depth_prev_t = depth_prev_t; a.trans = trans; a.cam_f = cam_f; a.cam_c = cam_c; a.h = h; a.w = w;
  a.para_prev_l = para_prev_l; a.depth_prev_l = depth_prev_l; a.other_prev_l = other_prev_l; a.para_prev_t = para_prev_t;
  a.f_input = f_input; a.f_stride = f_stride; a.log_off = log_off; a.other_off = other_off; a.log_scale = log_scale;
  a.depth_state_reset = depth_state_reset;
  int gx = m4d_blocks((long long)h * w, 256);
  if (gx > 4096) gx = 4096;
  m4d_launch(level_pre_kernel, dim3(gx, b), dim3(256), 0, (hipStream_t)stream, a);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_camera_pyramid(const float* cam_f, const float* cam_c, int b, int levels, float* f_out, float* c_out,
                                  void* stream) {
  M4D_CHECK_ARG(cam_f && cam_c && f_out && c_out && b > 0 && levels > 0 && levels < 31);
  const int n = 2 * b;
  m4d_launch(camera_pyramid_kernel, dim3((n * levels + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     cam_f, cam_c, n, levels, f_out, c_out);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_level_post(const float* refiner_out, const float* rot, int rot_c, const float* trans,
                              const float* cam_f, const float* cam_c, int b, int h, int w, float scale,
                              float* parallax, float* depth, float* other, float* depth_state, void* stream) {
  M4D_CHECK_ARG(refiner_out && rot && trans && cam_f && cam_c && parallax && depth && other);
  M4D_CHECK_ARG(b > 0 && h > 0 && w > 0 && (rot_c == 3 || rot_c == 4));
  M4D_CHECK_ARG((((uintptr_t)other) & 15u) == 0);
  int gx = m4d_blocks((long long)h * w, 256);
  if (gx > 4096) gx = 4096;
  m4d_launch(level_post_kernel, dim3(gx, b), dim3(256), 0, (hipStream_t)stream,
                     refiner_out, rot, rot_c, trans, cam_f, cam_c, h, w, scale, parallax, depth, other, depth_state);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_bias_act(const float* x, const float* bias, long long rows, int C, float slope,
                            float* out, void* stream) {
  M4D_CHECK_ARG(x && bias && out && rows > 0 && C > 0);
  const long long total = rows * C;
  const bool vec = (C % 4 == 0) && ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)bias) & 15u) == 0);
  if (vec) m4d_launch(bias_act_kernel<true>, dim3(grid1d(total >> 2)), dim3(256), 0, (hipStream_t)stream, x, bias, total, C, slope, out);
  else m4d_launch(bias_act_kernel<false>, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, x, bias, total, C, slope, out);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_bias_act_padded(const float* x, const float* bias, int b, int h, int w, int C, float slope,
                                   float* out, int out_h, int out_w, int off_y, int off_x, void* stream) {
  M4D_CHECK_ARG(x && bias && out && b > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0);
  M4D_CHECK_ARG(off_y >= 0 && off_x >= 0 && off_y + h <= out_h && off_x + w <= out_w);
  M4D_CHECK_ARG(((((uintptr_t)x | (uintptr_t)out | (uintptr_t)bias)) & 15u) == 0);
  const long long total4 = (long long)b * h * w * (C / 4);
  m4d_launch(bias_act_padded_kernel, dim3(grid1d(total4)), dim3(256), 0, (hipStream_t)stream,
                     x, bias, h, w, C, slope, out, out_h, out_w, off_y, off_x, total4);
  return M4D_LAUNCH_RESULT();
}
