// Two small fused reductions that the reference runs as a dozen (DINL) / ~50 (metrics)
// separate TF ops, and that PyTorch would run as as many tiny kernels:
//   * DomainNormalization (m4depth_network.py:44-48) fused with the leaky_relu that follows it
//     at encoder level 0 (:82-84): two deterministic two-stage reductions (mean, then the
//     two-pass variance TF computes) and ONE apply pass, instead of ~12 passes over the
//     largest activation of the network ([b,H,W,16]);
//   * the 7 depth metrics of metrics.py in one pass over (gt, est).
// Partial sums are combined in a fixed order (no atomics): results are run-to-run identical.
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

constexpr int kDinlMaxBlocks = 256;

// pass 0: sum x ; pass 1: sum (x - mean)^2, per (batch, channel); C % 4 == 0, C/4 divides 256
__global__ void __launch_bounds__(256)
dinl_partial_kernel(const float* __restrict__ x, const float* __restrict__ mean, int hw, int C, int pass,
                    float* __restrict__ partial) {
  __shared__ float4 sh[256];
  const int bi = blockIdx.y;
  const int c4n = C >> 2;
  const int ppi = 256 / c4n;                       // pixels per block iteration
  const int c4 = threadIdx.x % c4n, po = threadIdx.x / c4n;
  const float* xb = x + (long long)bi * hw * C;
  float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pass == 1) m = *reinterpret_cast<const float4*>(mean + bi * C + c4 * 4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = blockIdx.x * ppi + po; p < hw; p += gridDim.x * ppi) {
    float4 v = *reinterpret_cast<const float4*>(xb + (long long)p * C + c4 * 4);
    if (pass == 1) {
      v.x = v.x - m.x; v.y = v.y - m.y; v.z = v.z - m.z; v.w = v.w - m.w;
      v.x = v.x * v.x; v.y = v.y * v.y; v.z = v.z * v.z; v.w = v.w * v.w;
    }
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (po == 0) {                                   // fixed-order combine of the ppi partials of this channel group
    float4 s = sh[c4];
    for (int r = 1; r < ppi; ++r) {
      const float4 v = sh[r * c4n + c4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(partial + ((long long)bi * gridDim.x + blockIdx.x) * C + c4 * 4) = s;
  }
}

// 256 lanes: lane (j, c) sums every (256/C)-th partial of channel c, then the 256/C sub-sums are
// combined in index order -- a fixed summation tree, so the result is run-to-run identical.
__global__ void __launch_bounds__(256)
dinl_finalize_kernel(const float* __restrict__ partial, int nblk, int C, int hw, float* __restrict__ out) {
  __shared__ double sh[256];
  const int bi = blockIdx.x;
  const int c = threadIdx.x % C, j = threadIdx.x / C, nj = 256 / C;
  double s = 0.0;
  for (int k = j; k < nblk; k += nj) s += (double)partial[((long long)bi * nblk + k) * C + c];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (j == 0) {
    for (int r = 1; r < nj; ++r) s += sh[r * C + c];
    out[bi * C + c] = (float)(s / (double)hw);
  }
}

template <int C>
__global__ void __launch_bounds__(256)
dinl_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ var,
                  const float* __restrict__ scale, const float* __restrict__ bias, int hw, int w, float slope,
                  float* __restrict__ out, int out_h, int out_w, int off_y, int off_x) {
  const int bi = blockIdx.y;
  float mu[C], dv[C], sc[C], bs[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    mu[c] = mean[bi * C + c];
    dv[c] = var[bi * C + c] + 1e-12f;              // (x - mean) / (var + 1e-12): variance, not std (:47)
    sc[c] = scale[c];
    bs[c] = bias[c];
  }
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
    const float* px = x + ((long long)bi * hw + p) * C;
    float n[C];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < C; c += 4) {
      const float4 v = *reinterpret_cast<const float4*>(px + c);
      n[c] = (v.x - mu[c]) / dv[c]; n[c + 1] = (v.y - mu[c + 1]) / dv[c + 1];
      n[c + 2] = (v.z - mu[c + 2]) / dv[c + 2]; n[c + 3] = (v.w - mu[c + 3]) / dv[c + 3];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) ss = ss + n[c] * n[c];
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));                    // tf.math.l2_normalize
    float* po = out + (((long long)bi * out_h + (p / w + off_y)) * out_w + (p % w + off_x)) * C;   // dense or padded target
#pragma unroll
    for (int c = 0; c < C; c += 4) {
      float4 o;
      o.x = sc[c] * (n[c] * inv) + bs[c]; o.y = sc[c + 1] * (n[c + 1] * inv) + bs[c + 1];
      o.z = sc[c + 2] * (n[c + 2] * inv) + bs[c + 2]; o.w = sc[c + 3] * (n[c + 3] * inv) + bs[c + 3];
      o.x = o.x > 0.f ? o.x : o.x * slope; o.y = o.y > 0.f ? o.y : o.y * slope;
      o.z = o.z > 0.f ? o.z : o.z * slope; o.w = o.w > 0.f ? o.w : o.w * slope;
      *reinterpret_cast<float4*>(po + c) = o;
    }
  }
}

// ---- encoder level 0, first convolution: 3 -> 16 channels, stride 1, + bias, + the DINL mean partial sums ------
// K = 27 is far too small for the matrix cores and the layer writes the largest activation of the network
// ([b,H,W,16]) once; a direct convolution with the 27 x 16 weights read through the scalar cache.  The
// workgroup also leaves the per-channel sums of its outputs for the DINL mean (same partial layout as
// dinl_partial_kernel pass 0), which saves one full read of that activation.
template <int C>
__global__ void __launch_bounds__(256)
enc_head_conv_kernel(const float* __restrict__ img, const float* __restrict__ w27, const float* __restrict__ bias,
                     int h, int w, int bsz, long long stride_b, long long stride_t,
                     float* __restrict__ out, float* __restrict__ partial) {
  __shared__ float sh[4][C];
  const int bi = blockIdx.y;
  const int hw = h * w;
  // image bi = frame (bi / bsz) of sequence (bi % bsz): the frames of a [b,T,H,W,3] batch are encoded in one launch,
  // frame-major, straight from the sequence tensor (no stacking copy)
  const float* ib = img + (long long)(bi % bsz) * stride_b + (long long)(bi / bsz) * stride_t;
  float* ob = out + (long long)bi * hw * C;
  float csum[C];
#pragma unroll
  for (int c = 0; c < C; ++c) csum[c] = 0.f;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < hw; p += gridDim.x * 256) {
    const int y = p / w, x = p - y * w;
    typedef float f2 __attribute__((ext_vector_type(2)));         // packed v_pk_mul_f32 / v_pk_add_f32: two channels per op
    f2 acc2[C / 2];
#pragma unroll
    for (int c = 0; c < C / 2; ++c) acc2[c] = (f2){0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = x + kx - 1;
        const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;
        const float* px = ib + ((long long)(in ? yy : y) * w + (in ? xx : x)) * 3;
        const float v0 = in ? px[0] : 0.f, v1 = in ? px[1] : 0.f, v2 = in ? px[2] : 0.f;
        const f2* wt = reinterpret_cast<const f2*>(w27 + (ky * 3 + kx) * 3 * C);   // HWIO: [tap][cin][cout], uniform -> scalar loads
#pragma unroll
        for (int c = 0; c < C / 2; ++c)
          acc2[c] = ((acc2[c] + (f2){v0, v0} * wt[c]) + (f2){v1, v1} * wt[C / 2 + c]) + (f2){v2, v2} * wt[C + c];
      }
    }
    float acc[C];
#pragma unroll
    for (int c = 0; c < C / 2; ++c) { acc[2 * c] = acc2[c].x; acc[2 * c + 1] = acc2[c].y; }
#pragma unroll
    for (int c = 0; c < C; c += 4) {
      float4 o = make_float4(acc[c] + bias[c], acc[c + 1] + bias[c + 1], acc[c + 2] + bias[c + 2], acc[c + 3] + bias[c + 3]);
      *reinterpret_cast<float4*>(ob + (long long)p * C + c) = o;
      csum[c] += o.x; csum[c + 1] += o.y; csum[c + 2] += o.z; csum[c + 3] += o.w;
    }
  }
  // fixed-order reduction: butterfly inside the wave, then the 4 waves in index order
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float v = csum[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < C)
    partial[((long long)bi * gridDim.x + blockIdx.x) * C + threadIdx.x] =
        ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

// ---- metrics.py in one pass ---------------------------------------------------------
constexpr int kMetricSums = 9;
constexpr int kMetricBlocks = 512;

__global__ void __launch_bounds__(256)
metrics_partial_kernel(const float* __restrict__ gt_raw, const float* __restrict__ est_raw, long long n, float max_d,
                       double* __restrict__ partial) {
  float s[kMetricSums];
#pragma unroll
  for (int k = 0; k < kMetricSums; ++k) s[k] = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gt = fminf(fmaxf(gt_raw[i], 0.0f), max_d);               // m4depth_network.py:465-467
    const float est = fminf(fmaxf(est_raw[i], 0.001f), max_d);
    const bool m = gt > 1e-6f;                                            // metrics.py:3-5
    const float diff = gt - est;
    const float lg = logf(gt + 1e-6f), le = logf(est + 1e-6f);
    const bool m2 = lg > 1e-6f;                                           // RMSE_log masks on the LOG (metrics.py:24-28)
    const float th = fmaxf(gt / est, est / gt);
    if (m) {
      s[0] += 1.0f;
      s[1] += fabsf(diff) / (gt + 1e-6f);
      s[2] += diff * diff / (gt + 1e-6f);
      s[3] += diff * diff;
      s[6] += th < 1.25f ? 1.0f : 0.0f;
      s[7] += th < 1.5625f ? 1.0f : 0.0f;
      s[8] += th < 1.953125f ? 1.0f : 0.0f;
    }
    if (m2) { s[4] += 1.0f; s[5] += (lg - le) * (lg - le); }
  }
  __shared__ double sh[4][kMetricSums];
#pragma unroll
  for (int k = 0; k < kMetricSums; ++k) {
    double v = (double)s[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);             // fixed tree: deterministic
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kMetricSums)
    partial[(long long)blockIdx.x * kMetricSums + threadIdx.x] =
        ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

__global__ void __launch_bounds__(kMetricSums * 32)
metrics_finalize_kernel(const double* __restrict__ partial, int nblk, float* __restrict__ out7,
                        float* __restrict__ total7, float count, float* __restrict__ mean7) {
  __shared__ double sub[kMetricSums * 32];
  __shared__ double tot[kMetricSums];
  const int q = threadIdx.x % kMetricSums, j = threadIdx.x / kMetricSums;    // 32 sub-sums per quantity
  double s = 0.0;
  for (int k = j; k < nblk; k += 32) s += partial[(long long)k * kMetricSums + q];
  sub[threadIdx.x] = s;
  __syncthreads();
  if (j == 0) {
    for (int r = 1; r < 32; ++r) s += sub[r * kMetricSums + q];               // fixed order
    tot[q] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double cnt = tot[0] > 1.0 ? tot[0] : 1.0, cnt2 = tot[4] > 1.0 ? tot[4] : 1.0;
    out7[0] = (float)(tot[1] / cnt);                 // AbsRel
    out7[1] = (float)(tot[2] / cnt);                 // SqRel
    out7[2] = sqrtf((float)(tot[3] / cnt));          // RMSE
    out7[3] = sqrtf((float)(tot[5] / cnt2));         // RMSE_log
    out7[4] = (float)(tot[6] / cnt);                 // Delta1..3
    out7[5] = (float)(tot[7] / cnt);
    out7[6] = (float)(tot[8] / cnt);
    if (total7) {                                    // Keras Mean (metrics.py): total += per-batch value; result = total / count
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const float t = total7[i] + out7[i];
        total7[i] = t;
        if (mean7) mean7[i] = t / count;
      }
    }
  }
}

}  // namespace

extern "C" long long m4d_dinl_workspace_floats(int b, int C) {
  return (long long)b * (kDinlMaxBlocks + 2) * C;
}

extern "C" int m4d_dinl_fwd(const float* x, const float* scale, const float* bias, int b, int h, int w, int C,
                            float slope, float* workspace, float* out, void* stream) {
  return m4d_dinl_fwd_padded(x, scale, bias, b, h, w, C, slope, workspace, out, h, w, 0, 0, stream);
}

extern "C" int m4d_dinl_fwd_padded(const float* x, const float* scale, const float* bias, int b, int h, int w, int C,
                                   float slope, float* workspace, float* out, int out_h, int out_w, int off_y,
                                   int off_x, void* stream) {
  M4D_CHECK_ARG(x && scale && bias && workspace && out && b > 0 && h > 0 && w > 0);
  M4D_CHECK_ARG(off_y >= 0 && off_x >= 0 && off_y + h <= out_h && off_x + w <= out_w);
  M4D_CHECK_ARG(C == 16 || C == 32);               // the reference applies DINL to encoder level 0 only (16 channels)
  M4D_CHECK_ARG(((((uintptr_t)x | (uintptr_t)out | (uintptr_t)workspace)) & 15u) == 0);
  hipStream_t s = (hipStream_t)stream;
  const int hw = h * w;
  const int ppi = 256 / (C / 4);
  int nblk = (hw + ppi * 8 - 1) / (ppi * 8);
  if (nblk > kDinlMaxBlocks) nblk = kDinlMaxBlocks;
  float* partial = workspace;
  float* mean = workspace + (long long)b * kDinlMaxBlocks * C;
  float* var = mean + (long long)b * C;
  hipLaunchKernelGGL(dinl_partial_kernel, dim3(nblk, b), dim3(256), 0, s, x, (const float*)nullptr, hw, C, 0, partial);
  hipLaunchKernelGGL(dinl_finalize_kernel, dim3(b), dim3(256), 0, s, partial, nblk, C, hw, mean);
  hipLaunchKernelGGL(dinl_partial_kernel, dim3(nblk, b), dim3(256), 0, s, x, (const float*)mean, hw, C, 1, partial);
  hipLaunchKernelGGL(dinl_finalize_kernel, dim3(b), dim3(256), 0, s, partial, nblk, C, hw, var);
  int gx = m4d_blocks(hw, 256);
  if (gx > 2048) gx = 2048;
  if (C == 16) hipLaunchKernelGGL(dinl_apply_kernel<16>, dim3(gx, b), dim3(256), 0, s, x, mean, var, scale, bias, hw, w, slope, out, out_h, out_w, off_y, off_x);
  else hipLaunchKernelGGL(dinl_apply_kernel<32>, dim3(gx, b), dim3(256), 0, s, x, mean, var, scale, bias, hw, w, slope, out, out_h, out_w, off_y, off_x);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_enc_head_fwd(const float* images, int bsz, long long stride_b, long long stride_t,
                                const float* w_hwio, const float* bias, int b, int h, int w, int C,
                                float* workspace, float* raw_out, void* stream) {
  M4D_CHECK_ARG(images && w_hwio && bias && workspace && raw_out && b > 0 && h > 0 && w > 0);
  M4D_CHECK_ARG(C == 16);                            // encoder level 0 of the reference (m4depth_network.py:59)
  M4D_CHECK_ARG(bsz > 0 && b % bsz == 0 && stride_b >= 0 && stride_t >= 0);
  M4D_CHECK_ARG(((((uintptr_t)raw_out | (uintptr_t)workspace)) & 15u) == 0);
  hipStream_t s = (hipStream_t)stream;
  const int hw = h * w;
  int nblk = (hw + 256 * 8 - 1) / (256 * 8);       // (4x more, smaller workgroups: same kernel time, slower finalize)
  if (nblk > kDinlMaxBlocks) nblk = kDinlMaxBlocks;
  if (nblk < 1) nblk = 1;
  float* partial = workspace;
  float* mean = workspace + (long long)b * kDinlMaxBlocks * C;
  float* var = mean + (long long)b * C;
  hipLaunchKernelGGL(enc_head_conv_kernel<16>, dim3(nblk, b), dim3(256), 0, s, images, w_hwio, bias, h, w, bsz, stride_b, stride_t,
                     raw_out, partial);
  hipLaunchKernelGGL(dinl_finalize_kernel, dim3(b), dim3(256), 0, s, partial, nblk, C, hw, mean);
  const int ppi = 256 / (C / 4);
  int nblk2 = (hw + ppi * 8 - 1) / (ppi * 8);
  if (nblk2 > kDinlMaxBlocks) nblk2 = kDinlMaxBlocks;
  hipLaunchKernelGGL(dinl_partial_kernel, dim3(nblk2, b), dim3(256), 0, s, (const float*)raw_out, (const float*)mean, hw, C, 1, partial);
  hipLaunchKernelGGL(dinl_finalize_kernel, dim3(b), dim3(256), 0, s, partial, nblk2, C, hw, var);
  return M4D_LAUNCH_RESULT();
}

extern "C" long long m4d_metrics_workspace_bytes(void) { return (long long)kMetricBlocks * kMetricSums * sizeof(double); }

extern "C" int m4d_depth_metrics(const float* gt, const float* est, long long n, float max_d, void* workspace,
                                 float* out7, float* total7, float count, float* mean7, void* stream) {
  M4D_CHECK_ARG(gt && est && workspace && out7 && n > 0);
  M4D_CHECK_ARG(!mean7 || (total7 && count > 0.f));
  hipStream_t s = (hipStream_t)stream;
  long long g = (n + 255) / 256;
  const int nblk = (int)(g < kMetricBlocks ? g : kMetricBlocks);
  hipLaunchKernelGGL(metrics_partial_kernel, dim3(nblk), dim3(256), 0, s, gt, est, n, max_d, (double*)workspace);
  hipLaunchKernelGGL(metrics_finalize_kernel, dim3(1), dim3(kMetricSums * 32), 0, s, (const double*)workspace, nblk, out7, total7, count, mean7);
  return M4D_LAUNCH_RESULT();
}
