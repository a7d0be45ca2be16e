// Bilinear gathers: the BackProject op pair and dense_image_warp.
//
// Replaces cuda_backproject/backproject_op_gpu.cu.cc (thread-per-pixel with a
// serial channel loop: lanes 4*C bytes apart) with channel-across-lanes kernels:
// consecutive lanes read consecutive floats of the NHWC feature vector, so every
// corner fetch and every store is a contiguous run per pixel.
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

// ---- BackProject forward (backproject_op_gpu.cu.cc:19-79) -----------------------
__global__ void __launch_bounds__(256)
backproject_fwd_kernel(const float* __restrict__ input, const float* __restrict__ coords,
                       int B, int H, int W, int S, int F, int C, long long total,
                       float* __restrict__ out) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const long long item = idx / C;           // (n,h,w,s,f) flattened == coords row
    long long n = item;
    const int f = (int)(n % F); n /= F;
    n /= S;
    n /= W;
    n /= H;                                   // n = batch index
    const float x = coords[2 * item];
    const float y = coords[2 * item + 1];
    float v = 0.0f;
    if (x >= 0 && y >= 0 && x <= (float)(W - 1) && y <= (float)(H - 1)) {
      const int x0 = (int)floorf(x), x1 = (int)ceilf(x);
      const int y0 = (int)floorf(y), y1 = (int)ceilf(y);
      const float dx = x - (float)x0, dy = y - (float)y0;
      const float w00 = (1.0f - dy) * (1.0f - dx), w01 = (1.0f - dy) * dx;
      const float w10 = dy * (1.0f - dx), w11 = dy * dx;
      const long long base = (n * H * W * F + f) * (long long)C + c;
      const long long ps = (long long)F * C;
      const float im00 = input[base + ps * ((long long)y0 * W + x0)];
      const float im01 = input[base + ps * ((long long)y0 * W + x1)];
      const float im10 = input[base + ps * ((long long)y1 * W + x0)];
      const float im11 = input[base + ps * ((long long)y1 * W + x1)];
      v = ((im00 * w00 + im01 * w01) + im10 * w10) + im11 * w11;
    }
    out[idx] = v;
  }
}

// ---- BackProject backward (backproject_op_gpu.cu.cc:108-197) --------------------
// Scatter half: one lane per (item, channel), hardware fp32 atomic adds.
__global__ void __launch_bounds__(256)
backproject_bwd_scatter_kernel(const float* __restrict__ grad, const float* __restrict__ coords,
                               int B, int H, int W, int S, int F, int C, long long total,
                               float* __restrict__ input_grad) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const long long item = idx / C;
    long long n = item;
    const int f = (int)(n % F); n /= F;
    n /= S; n /= W; n /= H;
    const float x = coords[2 * item];
    const float y = coords[2 * item + 1];
    if (x >= 0 && y >= 0 && x <= (float)(W - 1) && y <= (float)(H - 1)) {
      const int x0 = (int)floorf(x), x1 = (int)ceilf(x);
      const int y0 = (int)floorf(y), y1 = (int)ceilf(y);
      const float dx = x - (float)x0, dy = y - (float)y0;
      const float w00 = (1.0f - dy) * (1.0f - dx), w01 = (1.0f - dy) * dx;
      const float w10 = dy * (1.0f - dx), w11 = dy * dx;
      const long long base = (n * H * W * F + f) * (long long)C + c;
      const long long ps = (long long)F * C;
      const float g = grad[idx];
      unsafeAtomicAdd(input_grad + base + ps * ((long long)y0 * W + x0), g * w00);
      unsafeAtomicAdd(input_grad + base + ps * ((long long)y0 * W + x1), g * w01);
      unsafeAtomicAdd(input_grad + base + ps * ((long long)y1 * W + x0), g * w10);
      unsafeAtomicAdd(input_grad + base + ps * ((long long)y1 * W + x1), g * w11);
    }
  }
}

// Coordinate-gradient half: one lane per item, sequential channel loop (same
// order as the reference kernel's loop, :171-186).
__global__ void __launch_bounds__(256)
backproject_bwd_coords_kernel(const float* __restrict__ grad, const float* __restrict__ input,
                              const float* __restrict__ coords, int B, int H, int W, int S, int F,
                              int C, long long items, float* __restrict__ coords_grad) {
  for (long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x; item < items;
       item += (long long)gridDim.x * blockDim.x) {
    long long n = item;
    const int f = (int)(n % F); n /= F;
    n /= S; n /= W; n /= H;
    const float x = coords[2 * item];
    const float y = coords[2 * item + 1];
    float gx = 0.0f, gy = 0.0f;
    if (x >= 0 && y >= 0 && x <= (float)(W - 1) && y <= (float)(H - 1)) {
      const int x0 = (int)floorf(x), x1 = (int)ceilf(x);
      const int y0 = (int)floorf(y), y1 = (int)ceilf(y);
      const float dx = x - (float)x0, dy = y - (float)y0;
      const float wx0 = 1.0f - dx, wx1 = dx, wy0 = 1.0f - dy, wy1 = dy;
      const long long base = (n * H * W * F + f) * (long long)C;
      const long long ps = (long long)F * C;
      const float* im00 = input + base + ps * ((long long)y0 * W + x0);
      const float* im01 = input + base + ps * ((long long)y0 * W + x1);
      const float* im10 = input + base + ps * ((long long)y1 * W + x0);
      const float* im11 = input + base + ps * ((long long)y1 * W + x1);
      const float* g = grad + item * C;
      for (int c = 0; c < C; ++c) {
        const float gc = g[c];
        gx = gx + gc * (wy0 * (im01[c] - im00[c]) + wy1 * (im11[c] - im10[c]));
        gy = gy + gc * (wx0 * (im10[c] - im00[c]) + wx1 * (im11[c] - im01[c]));
      }
    }
    coords_grad[2 * item] = gx;
    coords_grad[2 * item + 1] = gy;
  }
}

// ---- dense_image_warp / _interpolate_bilinear, TF-CPU branch
// (utils/dense_image_warp.py:61-192, 238-259).  pts is either a flow field on the
// image grid (add_grid: query = (j,i) + flow, :244) or N free query points (row, col).
__global__ void __launch_bounds__(256)
interp_bilinear_kernel(const float* __restrict__ image, const float* __restrict__ pts,
                       int B, int H, int W, int C, int N, int add_grid, long long total,
                       float* __restrict__ out, int32_t* __restrict__ index_out) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const long long q = idx / C;              // (b, n) flattened
    const long long bi = q / N;
    float qy = pts[2 * q];
    float qx = pts[2 * q + 1];
    if (add_grid) {
      const int n = (int)(q % N);
      qy = (float)(n / W) + qy;
      qx = (float)(n % W) + qx;
    }
    int y0, x0;
    float ay, ax;
    m4d_bilinear_axis(qy, H, y0, ay);
    m4d_bilinear_axis(qx, W, x0, ax);
    const float* p = image + ((bi * H + y0) * W + x0) * (long long)C + c;
    const long long rs = (long long)W * C;
    out[idx] = m4d_lerp2(p[0], p[C], p[rs], p[rs + C], ax, ay);
    if (index_out != nullptr && c == 0) {
      index_out[2 * q] = y0;
      index_out[2 * q + 1] = x0;
    }
  }
}

inline int grid_for(long long total, int threads) {
  long long g = (total + threads - 1) / threads;
  const long long cap = 256LL * 32;          // 256 CUs x 8 blocks of 4 waves, grid-stride beyond
  return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

}  // namespace

extern "C" int m4d_backproject_fwd(const float* input, const float* coords, const int dims[6],
                                   float* out, void* stream) {
  M4D_CHECK_ARG(input && coords && dims && out);
  for (int k = 0; k < 6; ++k) M4D_CHECK_ARG(dims[k] > 0);
  M4D_CHECK_ARG(dims[1] >= 1 && dims[2] >= 1);
  const long long total = (long long)dims[0] * dims[1] * dims[2] * dims[3] * dims[4] * dims[5];
  m4d_launch(backproject_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     input, coords, dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], total, out);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_backproject_bwd(const float* grad, const float* input, const float* coords,
                                   const int dims[6], float* input_grad, float* coords_grad, void* stream) {
  M4D_CHECK_ARG(grad && input && coords && dims && input_grad && coords_grad);
  for (int k = 0; k < 6; ++k) M4D_CHECK_ARG(dims[k] > 0);
  const long long items = (long long)dims[0] * dims[1] * dims[2] * dims[3] * dims[4];
  const long long total = items * dims[5];
  const size_t in_bytes = sizeof(float) * (size_t)dims[0] * dims[1] * dims[2] * dims[4] * dims[5];
#if M4D_EXPERIMENTS
  if (m4d_tape_recording()) return (int)hipErrorNotSupported;      // a memset is not a kernel launch: it would be lost from the replay
#endif
  hipError_t e = hipMemsetAsync(input_grad, 0, in_bytes, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  m4d_launch(backproject_bwd_scatter_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     grad, coords, dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], total, input_grad);
  m4d_launch(backproject_bwd_coords_kernel, dim3(grid_for(items, 256)), dim3(256), 0, (hipStream_t)stream,
                     grad, input, coords, dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], items, coords_grad);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_dense_image_warp(const float* image, const float* flow, int B, int H, int W, int C,
                                    float* out, int32_t* index_out, void* stream) {
  M4D_CHECK_ARG(image && flow && out);
  M4D_CHECK_ARG(B > 0 && C > 0 && H >= 2 && W >= 2);   // dense_image_warp.py:116-119
  const long long total = (long long)B * H * W * C;
  m4d_launch(interp_bilinear_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     image, flow, B, H, W, C, H * W, 1, total, out, index_out);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_interpolate_bilinear(const float* grid, const float* query, int B, int H, int W, int C,
                                        int N, float* out, int32_t* index_out, void* stream) {
  M4D_CHECK_ARG(grid && query && out);
  M4D_CHECK_ARG(B > 0 && C > 0 && N > 0 && H >= 2 && W >= 2);
  const long long total = (long long)B * N * C;
  m4d_launch(interp_bilinear_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     grid, query, B, H, W, C, N, 0, total, out, index_out);
  return M4D_LAUNCH_RESULT();
}
