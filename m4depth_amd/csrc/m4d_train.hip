// Per-pixel glue of the TRAINING graph (train_step, m4depth_network.py:371-399), one kernel per
// autograd node instead of the 5-40 elementwise / indexing kernels a framework emits for it:
//   * adjoint of the legacy x2 bilinear upsample (m4depth_network.py:202-204)      -- gather, deterministic
//   * backward of the level tail exp(clip)/2^m -> parallax2depth (:247-251)
//   * backward of the per-cut normalisation (:179-189)
//   * backward of the convolution epilogue (leaky_relu mask + bias gradient)        -- 2-stage, deterministic
//   * re-packing of the live OIHW weights for the MFMA convolution (forward layout, or the
//     transposed / 180-degree-rotated layout of its data gradient)
//   * the log-depth L1 term of m4depth_loss (:491-536) for one pyramid level, 'map' and 'velodyne'
//     ground truth, forward (2-stage deterministic mean) and backward.
// All of it is HBM-bound streaming over small maps; what matters is the launch count.
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

inline int grid1d(long long n) { long long g = (n + 255) / 256; return (int)(g > 65535 ? 65535 : (g < 1 ? 1 : g)); }

// ---- adjoint of tf.compat.v1.image.resize_bilinear (ResizeAxis / resize_axis: m4d_common.h) ----------
// weight of input index `in` in output index `o` along one axis
__device__ __forceinline__ float axis_weight(int o, float scale, int in_n, int in) {
  const ResizeAxis a = resize_axis(o, scale, in_n);
  float wgt = 0.f;
  if (a.lo == in) wgt += 1.0f - a.lerp;
  if (a.hi == in) wgt += a.lerp;
  return wgt;
}

__global__ void __launch_bounds__(256)
resize_bilinear_v1_bwd_kernel(const float* __restrict__ g_out, int ih, int iw, int c, int oh, int ow,
                              float mul, long long total, float* __restrict__ g_in) {
  const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(idx % c);
    long long r = idx / c;
    const int x = (int)(r % iw); r /= iw;
    const int y = (int)(r % ih);
    const long long bi = r / ih;
    const int oy0 = max(0, (int)floorf((float)(y - 1) / sy)), oy1 = min(oh - 1, (int)ceilf((float)(y + 1) / sy));
    const int ox0 = max(0, (int)floorf((float)(x - 1) / sx)), ox1 = min(ow - 1, (int)ceilf((float)(x + 1) / sx));
    float acc = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
      const float wy = axis_weight(oy, sy, ih, y);
      if (wy == 0.f) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        const float wx = axis_weight(ox, sx, iw, x);
        if (wx == 0.f) continue;
        acc += g_out[((bi * oh + oy) * ow + ox) * c + cc] * (wy * wx);
      }
    }
    g_in[idx] = acc * mul;
  }
}

// ---- backward of the level tail -----------------------------------------------------
__global__ void __launch_bounds__(256)
level_post_bwd_kernel(const float* __restrict__ ro, const float* __restrict__ g_para,
                      const float* __restrict__ g_depth, const float* __restrict__ g_other,
                      const float* __restrict__ rot, int rot_c, const float* __restrict__ trans,
                      const float* __restrict__ cam_f, const float* __restrict__ cam_c, int h, int w,
                      float scale, float* __restrict__ g_ro) {
  const int bi = blockIdx.y;
  const M4dMotion m = m4d_load_motion(rot, rot_c, trans, cam_f, cam_c, bi);
  const int hw = h * w;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
    const int i = p % w, j = p / w;
    const long long gp = (long long)bi * hw + p;
    const float x = ro[gp * 5];
    const float para = expf(fminf(fmaxf(x, -7.0f), 7.0f)) / scale;                     // :250
    const M4dPixel px = m4d_pixel_factors(m, i, j);
    float gpar = g_para ? g_para[gp] : 0.f;
    if (g_depth) gpar += g_depth[gp] * (-(px.s / (para * para)) / px.alpha);            // d/dpara (s/para - tz)/alpha
    float* o = g_ro + gp * 5;
    o[0] = (x >= -7.0f && x <= 7.0f) ? gpar * para : 0.f;                               // clip passes inside its bounds
    if (g_other) {
      const float4 v = *reinterpret_cast<const float4*>(g_other + gp * 4);
      o[1] = v.x; o[2] = v.y; o[3] = v.z; o[4] = v.w;
    } else {
      o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f;
    }
  }
}

// ---- backward of x / ||x|| per cut ---------------------------------------------------
__global__ void __launch_bounds__(256)
normalize_cuts_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, long long groups, int nc,
                          float* __restrict__ gx) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < groups;
       i += (long long)gridDim.x * blockDim.x) {
    const float* px = x + i * nc;
    const float* pg = g + i * nc;
    float ss = 0.f, dot = 0.f;
    for (int c = 0; c < nc; ++c) { ss += px[c] * px[c]; dot += pg[c] * px[c]; }
    const float n = sqrtf(ss);
    const float k = dot / (ss * n);                      // sum(g*x) / n^3
    for (int c = 0; c < nc; ++c) gx[i * nc + c] = pg[c] / n - px[c] * k;
  }
}

// ---- backward of the convolution epilogue ------------------------------------------------
// gp = g * (out > 0 ? 1 : slope); partial[blk][c] = sum over the block's rows of gp[:, c].
__global__ void __launch_bounds__(256)
bias_act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out, long long rows, int C, float slope,
                    int rows_per_block, float* __restrict__ gp, float* __restrict__ partial) {
  __shared__ float red[256];
  const int T = (256 / C) * C;                           // threads in use: a whole number of rows
  const int RP = T / C;
  const int t = threadIdx.x;
  float acc = 0.f;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  if (t < T) {
    const int c = t % C, ro = t / C;
    for (long long r = r0 + ro; r < r1; r += RP) {
      const long long e = r * C + c;
      const float v = g[e] * (out[e] > 0.f ? 1.0f : slope);
      gp[e] = v;
      acc += v;
    }
  }
  red[t] = acc;
  __syncthreads();
  if (t < C) {
    float s = red[t];
    for (int k = 1; k < RP; ++k) s += red[t + k * C];
    partial[(long long)blockIdx.x * C + t] = s;
  }
}
// C > 256: one thread per channel per row chunk is not possible in one block row; generic fallback
__global__ void __launch_bounds__(256)
bias_act_bwd_wide_kernel(const float* __restrict__ g, const float* __restrict__ out, long long rows, int C, float slope,
                         int rows_per_block, float* __restrict__ gp, float* __restrict__ partial) {
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (long long r = r0; r < r1; ++r) {
      const long long e = r * C + c;
      const float v = g[e] * (out[e] > 0.f ? 1.0f : slope);
      gp[e] = v;
      acc += v;
    }
    partial[(long long)blockIdx.x * C + c] = acc;
  }
}
// one wave per channel: lane t sums the partials of blocks t, t+64, ... in block order, then a fixed
// butterfly combines the 64 lanes -- deterministic, and 16x shorter than one serial chain per channel
__global__ void __launch_bounds__(64)
bias_grad_finalize_kernel(const float* __restrict__ partial, int blocks, int C, float* __restrict__ g_bias) {
  const int c = blockIdx.x;
  float s = 0.f;
  for (int b = threadIdx.x; b < blocks; b += 64) s += partial[(long long)b * C + c];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (threadIdx.x == 0) g_bias[c] = s;
}

// ---- weight re-packing for conv3x3_mfma_kernel -----------------------------------------
// w: [O][3][3][I] (the channels-last memory of the OIHW parameter).  Output [chunk][tap][n][16],
// the 16 input channels of a chunk even-first.  transpose = 0: forward (K = I, N = O);
// 1: data gradient (K = O, N = I, taps rotated by 180 degrees).
__global__ void __launch_bounds__(256)
pack_conv_weights_kernel(const float* __restrict__ w, int O, int I, int transpose, int n_pad, long long total,
                         float* __restrict__ wp) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int p = (int)(idx & 15);
  long long r = idx >> 4;
  const int n = (int)(r % n_pad); r /= n_pad;
  const int t = (int)(r % 9);
  const int ch = (int)(r / 9);
  const int kc = ch * 16 + (p < 8 ? 2 * p : 2 * (p - 8) + 1);
  const int K = transpose ? O : I, N = transpose ? I : O;
  float v = 0.f;
  if (kc < K && n < N) {
    const int tap = transpose ? 8 - t : t;
    const int o = transpose ? kc : n, i = transpose ? n : kc;
    v = w[((long long)o * 9 + tap) * I + i];
  }
  wp[idx] = v;
}

// The same parameter in the layout of m4d_conv3x3_lat (network_ops.pack_conv_weights_lat, built on the host for inference):
// [N/32 groups][K/16 chunks][9 taps][3 parts][64 lanes][8] bf16, lane = k_half * 32 + n % 32, element e = channel 8 k_half + e,
// every float32 weight split exactly into three bf16 terms (m4d_split3_pair: the host splitter's roundings).  One thread per
// (group, chunk, tap, lane): 8 weights -> 3 x 16 bytes.
__global__ void __launch_bounds__(256)
pack_conv_weights_lat_kernel(const float* __restrict__ w, int O, int I, int transpose, int n_chunks, long long total,
                             uint4* __restrict__ wp) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  long long r = idx >> 6;
  const int t = (int)(r % 9); r /= 9;
  const int c = (int)(r % n_chunks);
  const int g = (int)(r / n_chunks);
  const int K = transpose ? O : I, N = transpose ? I : O;
  const int n = g * 32 + (lane & 31), k0 = c * 16 + (lane >> 5) * 8;
  const int tap = transpose ? 8 - t : t;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kc = k0 + e;
    float x = 0.f;
    if (kc < K && n < N) {
      const int o = transpose ? kc : n, i = transpose ? n : kc;
      x = w[((long long)o * 9 + tap) * I + i];
    }
    v[e] = x;
  }
  unsigned hi[4], mid[4], lo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) m4d_split3_pair(v[2 * e], v[2 * e + 1], hi[e], mid[e], lo[e]);
  uint4* dst = wp + (((long long)(g * n_chunks + c) * 9 + t) * 3) * 64 + lane;
  dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  dst[64] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
  dst[128] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

// ---- m4depth_loss, one pyramid level -------------------------------------------------------
struct LossAxis { int lo, hi; float lerp; };
__device__ __forceinline__ LossAxis half_pixel_axis(int o, float scale, int in_n) {
  const float src = ((float)o + 0.5f) * scale - 0.5f;      // tf.image.resize, half-pixel centres, no antialias
  const float fl = floorf(src);
  LossAxis a;
  a.lo = min(max((int)fl, 0), in_n - 1);
  a.hi = min(max((int)ceilf(src), 0), in_n - 1);
  a.lerp = src - fl;
  return a;
}
__device__ __forceinline__ float log_clip(float v) { return logf(fminf(fmaxf(v, 0.01f), 200.0f)); }

// gt_resized and the mask (1 for 'map') of one prediction pixel
__device__ __forceinline__ void loss_target(const float* __restrict__ gt, int H, int W, int h, int w, int y, int x,
                                            int velodyne, float& target, float& mask) {
  if (!velodyne) {
    const LossAxis ya = half_pixel_axis(y, (float)H / (float)h, H), xa = half_pixel_axis(x, (float)W / (float)w, W);
    const float tl = log_clip(gt[(long long)ya.lo * W + xa.lo]), tr = log_clip(gt[(long long)ya.lo * W + xa.hi]);
    const float bl = log_clip(gt[(long long)ya.hi * W + xa.lo]), br = log_clip(gt[(long long)ya.hi * W + xa.hi]);
    const float top = tl + (tr - tl) * xa.lerp;
    const float bot = bl + (br - bl) * xa.lerp;
    target = top + (bot - top) * ya.lerp;
    mask = 1.0f;
  } else {                                                 // :514-524: masked mean over the (H/h) x (W/w) block
    const int fy = H / h, fx = W / w;
    float s = 0.f, n = 0.f;
    for (int dy = 0; dy < fy; ++dy)
      for (int dx = 0; dx < fx; ++dx) {
        const float v = gt[(long long)(y * fy + dy) * W + (x * fx + dx)];
        if (v > 0.f) { s += log_clip(v); n += 1.0f; }
      }
    target = s / (n + 1e-12f);
    mask = n > 0.f ? 1.0f : 0.f;
  }
}

__global__ void __launch_bounds__(256)
loss_level_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int b, int h, int w, int H, int W,
                      int velodyne, float* __restrict__ partial) {
  __shared__ float rs[256], rn[256];
  const long long total = (long long)b * h * w;
  float s = 0.f, n = 0.f;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % w);
    const int y = (int)((idx / w) % h);
    const long long bi = idx / ((long long)h * w);
    float target, mask;
    loss_target(gt + bi * H * W, H, W, h, w, y, x, velodyne, target, mask);
    s += fabsf(target - log_clip(pred[idx])) * mask;
    n += mask;
  }
  rs[threadIdx.x] = s; rn[threadIdx.x] = n;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { rs[threadIdx.x] += rs[threadIdx.x + off]; rn[threadIdx.x] += rn[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = rs[0]; partial[2 * blockIdx.x + 1] = rn[0]; }
}
__global__ void loss_level_finalize_kernel(const float* __restrict__ partial, int blocks, int velodyne,
                                           float* __restrict__ out) {
  float s = 0.f, n = 0.f;
  for (int i = 0; i < blocks; ++i) { s += partial[2 * i]; n += partial[2 * i + 1]; }
  out[0] = s / (velodyne ? n + 1e-12f : n);                // reduce_mean / masked_reduce_mean (:506-507)
  out[1] = n;
}
__global__ void __launch_bounds__(256)
loss_level_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ stats,
                      const float* __restrict__ g_out, int b, int h, int w, int H, int W, int velodyne,
                      float* __restrict__ g_pred) {
  const long long total = (long long)b * h * w;
  const float n = stats[1];
  const float gscale = g_out[0] / (velodyne ? n + 1e-12f : n);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % w);
    const int y = (int)((idx / w) % h);
    const long long bi = idx / ((long long)h * w);
    float target, mask;
    loss_target(gt + bi * H * W, H, W, h, w, y, x, velodyne, target, mask);
    const float p = pred[idx];
    const float d = target - log_clip(p);
    const float sgn = d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.f);
    const float inside = (p >= 0.01f && p <= 200.0f) ? 1.0f / p : 0.f;     // clip passes inside its bounds; d log
    g_pred[idx] = -(gscale * mask * sgn) * inside;
  }
}

}  // namespace

extern "C" int m4d_resize_bilinear_v1_bwd(const float* g_out, int b, int ih, int iw, int c, int oh, int ow,
                                          float mul, float* g_in, void* stream) {
  M4D_CHECK_ARG(g_out && g_in && b > 0 && ih > 0 && iw > 0 && c > 0 && oh > 0 && ow > 0);
  const long long total = (long long)b * ih * iw * c;
  m4d_launch(resize_bilinear_v1_bwd_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream,
                     g_out, ih, iw, c, oh, ow, mul, total, g_in);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_level_post_bwd(const float* refiner_out, const float* g_parallax, const float* g_depth,
                                  const float* g_other, const float* rot, int rot_c, const float* trans,
                                  const float* cam_f, const float* cam_c, int b, int h, int w, float scale,
                                  float* g_refiner_out, void* stream) {
  M4D_CHECK_ARG(refiner_out && g_refiner_out && trans && cam_f && cam_c && b > 0 && h > 0 && w > 0);
  M4D_CHECK_ARG(rot == nullptr || rot_c == 3 || rot_c == 4);
  if (g_other) M4D_CHECK_ARG((((uintptr_t)g_other) & 15u) == 0);
  int gx = m4d_blocks((long long)h * w, 256);
  if (gx > 4096) gx = 4096;
  m4d_launch(level_post_bwd_kernel, dim3(gx, b), dim3(256), 0, (hipStream_t)stream, refiner_out, g_parallax,
                     g_depth, g_other, rot, rot_c, trans, cam_f, cam_c, h, w, scale, g_refiner_out);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_normalize_cuts_bwd(const float* x, const float* g, int b, int h, int w, int C, int nbre_cuts,
                                      float* g_x, void* stream) {
  M4D_CHECK_ARG(x && g && g_x && b > 0 && h > 0 && w > 0 && C > 0 && nbre_cuts > 0 && C % nbre_cuts == 0);
  const long long groups = (long long)b * h * w * nbre_cuts;
  m4d_launch(normalize_cuts_bwd_kernel, dim3(grid1d(groups)), dim3(256), 0, (hipStream_t)stream,
                     x, g, groups, C / nbre_cuts, g_x);
  return M4D_LAUNCH_RESULT();
}

static inline void bias_bwd_plan(long long rows, int C, int& blocks, int& rpb) {
  const int RP = C <= 256 ? 256 / C : 1;
  long long want = (rows + (long long)RP * 16 - 1) / ((long long)RP * 16);   // >= 16 iterations per thread
  if (want > 1024) want = 1024;
  if (want < 1) want = 1;
  rpb = (int)((rows + want - 1) / want);
  blocks = (int)((rows + rpb - 1) / rpb);
}

extern "C" long long m4d_bias_act_bwd_workspace_floats(long long rows, int C) {
  int blocks, rpb;
  bias_bwd_plan(rows, C, blocks, rpb);
  return (long long)blocks * C;
}

extern "C" int m4d_bias_act_bwd(const float* g, const float* out, long long rows, int C, float slope,
                                float* g_pre, float* g_bias, float* workspace, void* stream) {
  M4D_CHECK_ARG(g && out && g_pre && g_bias && workspace && rows > 0 && C > 0);
  int blocks, rpb;
  bias_bwd_plan(rows, C, blocks, rpb);
  hipStream_t s = (hipStream_t)stream;
  if (C <= 256)
    m4d_launch(bias_act_bwd_kernel, dim3(blocks), dim3(256), 0, s, g, out, rows, C, slope, rpb, g_pre, workspace);
  else
    m4d_launch(bias_act_bwd_wide_kernel, dim3(blocks), dim3(256), 0, s, g, out, rows, C, slope, rpb, g_pre, workspace);
  m4d_launch(bias_grad_finalize_kernel, dim3(C), dim3(64), 0, s, workspace, blocks, C, g_bias);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_pack_conv_weights(const float* w_ohwi, int O, int I, int transpose, float* wp, void* stream) {
  M4D_CHECK_ARG(w_ohwi && wp && O > 0 && I > 0);
  const int K = transpose ? O : I, N = transpose ? I : O;
  const int n_pad = (N + 31) / 32 * 32;
  const long long total = (long long)((K + 15) / 16) * 9 * n_pad * 16;
  m4d_launch(pack_conv_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     w_ohwi, O, I, transpose, n_pad, total, wp);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_pack_conv_weights_lat(const float* w_ohwi, int O, int I, int transpose, void* wp, void* stream) {
  M4D_CHECK_ARG(w_ohwi && wp && O > 0 && I > 0);
  const int K = transpose ? O : I, N = transpose ? I : O;
  const int n_chunks = (K + 15) / 16, n_groups = (N + 31) / 32;
  const long long total = (long long)n_groups * n_chunks * 9 * 64;
  m4d_launch(pack_conv_weights_lat_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
             w_ohwi, O, I, transpose, n_chunks, total, reinterpret_cast<uint4*>(wp));
  return M4D_LAUNCH_RESULT();
}

extern "C" long long m4d_loss_workspace_floats(void) { return 2 * 512; }

extern "C" int m4d_loss_level_fwd(const float* pred_depth, const float* gt_depth, int b, int h, int w, int H, int W,
                                  int velodyne, float* workspace, float* out2, void* stream) {
  M4D_CHECK_ARG(pred_depth && gt_depth && workspace && out2 && b > 0 && h > 0 && w > 0 && H >= h && W >= w);
  if (velodyne) M4D_CHECK_ARG(H % h == 0 && W % w == 0);
  long long blocks = ((long long)b * h * w + 255) / 256;
  if (blocks > 512) blocks = 512;
  hipStream_t s = (hipStream_t)stream;
  m4d_launch(loss_level_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pred_depth, gt_depth, b, h, w, H, W,
                     velodyne, workspace);
  m4d_launch(loss_level_finalize_kernel, dim3(1), dim3(1), 0, s, workspace, (int)blocks, velodyne, out2);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_loss_level_bwd(const float* pred_depth, const float* gt_depth, const float* stats2,
                                  const float* g_out, int b, int h, int w, int H, int W, int velodyne,
                                  float* g_pred, void* stream) {
  M4D_CHECK_ARG(pred_depth && gt_depth && stats2 && g_out && g_pred && b > 0 && h > 0 && w > 0);
  const long long total = (long long)b * h * w;
  m4d_launch(loss_level_bwd_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, pred_depth, gt_depth,
                     stats2, g_out, b, h, w, H, W, velodyne, g_pred);
  return M4D_LAUNCH_RESULT();
}
