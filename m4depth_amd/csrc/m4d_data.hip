// Data front end: what dataloaders/{midair,kitti,tartanair}.py::_decode_samples do after the file
// is decompressed -- cast, scale, resize -- fused into one pass on the device.
//
// The reference decodes on the host, converts the full-resolution image to float32 (12 bytes per
// pixel), resizes it with tf.image.resize and only then ships the network-sized frame to the GPU.
// Here the host uploads the decoder's raw output (3 bytes per RGB pixel, 2 per 16-bit depth
// pixel) and one kernel per frame batch writes the NHWC float32 network input:
//   RGB   : u8 / 255 -> tf.image.resize bilinear (half-pixel centres, no antialias)   midair.py:35-45
//   depth : Mid-Air  uint16 bits = float16 disparity, depth = 512 / x, bilinear        midair.py:49-55
//           KITTI    uint16 / 256, nearest, optional Garg/Eigen evaluation crop        kitti.py:43-50
//           TartanAir float32, nearest, zeroed where the resized RGB pixel is black    tartanair.py:37-45
// One IEEE rounding per operation in the order of oracle/m4depth_oracle_data.py (bit-exact).
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

inline int grid1d(long long n) { long long g = (n + 255) / 256; return (int)(g > 65535 ? 65535 : (g < 1 ? 1 : g)); }

struct Axis { int lo, hi; float lerp; };
__device__ __forceinline__ Axis half_pixel_axis(int o, float scale, int in_n) {
  const float src = ((float)o + 0.5f) * scale - 0.5f;
  const float fl = floorf(src);
  Axis a;
  a.lo = min(max((int)fl, 0), in_n - 1);
  a.hi = min(max((int)ceilf(src), 0), in_n - 1);
  a.lerp = src - fl;
  return a;
}
__device__ __forceinline__ int nearest_index(int o, float scale, int in_n) {      // half_pixel_centers nearest
  return min((int)floorf(((float)o + 0.5f) * scale), in_n - 1);
}
__device__ __forceinline__ float lerp2(float tl, float tr, float bl, float br, float xl, float yl) {
  const float top = tl + (tr - tl) * xl;
  const float bot = bl + (br - bl) * xl;
  return top + (bot - top) * yl;
}

__global__ void __launch_bounds__(256)
rgb8_resize_kernel(const uint8_t* __restrict__ img, int ih, int iw, int oh, int ow, long long total,
                   float* __restrict__ out) {
  const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % 3);
    long long r = idx / 3;
    const int x = (int)(r % ow); r /= ow;
    const int y = (int)(r % oh);
    const long long n = r / oh;
    const uint8_t* p = img + n * ih * iw * 3;
    const Axis ya = half_pixel_axis(y, sy, ih), xa = half_pixel_axis(x, sx, iw);
    const float tl = (float)p[((long long)ya.lo * iw + xa.lo) * 3 + c] / 255.0f;
    const float tr = (float)p[((long long)ya.lo * iw + xa.hi) * 3 + c] / 255.0f;
    const float bl = (float)p[((long long)ya.hi * iw + xa.lo) * 3 + c] / 255.0f;
    const float br = (float)p[((long long)ya.hi * iw + xa.hi) * 3 + c] / 255.0f;
    out[idx] = lerp2(tl, tr, bl, br, xa.lerp, ya.lerp);
  }
}

__device__ __forceinline__ float midair_depth(uint16_t bits) {
  __half_raw hr; hr.x = bits;
  return 512.0f / __half2float(__half(hr));                  // midair.py:52-53
}

__global__ void __launch_bounds__(256)
depth_resize_kernel(const void* __restrict__ raw, int kind, int ih, int iw, int oh, int ow,
                    const float* __restrict__ rgb, int crop_y0, int crop_y1, int crop_x0, int crop_x1,
                    long long total, float* __restrict__ out) {
  const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % ow);
    const int y = (int)((idx / ow) % oh);
    const long long n = idx / ((long long)oh * ow);
    float v;
    if (kind == 0) {                                          // Mid-Air: bilinear over 512 / float16
      const uint16_t* p = reinterpret_cast<const uint16_t*>(raw) + n * ih * iw;
      const Axis ya = half_pixel_axis(y, sy, ih), xa = half_pixel_axis(x, sx, iw);
      v = lerp2(midair_depth(p[(long long)ya.lo * iw + xa.lo]), midair_depth(p[(long long)ya.lo * iw + xa.hi]),
                midair_depth(p[(long long)ya.hi * iw + xa.lo]), midair_depth(p[(long long)ya.hi * iw + xa.hi]),
                xa.lerp, ya.lerp);
    } else if (kind == 1) {                                   // KITTI: uint16 / 256, nearest, evaluation crop
      const uint16_t* p = reinterpret_cast<const uint16_t*>(raw) + n * ih * iw;
      v = (float)p[(long long)nearest_index(y, sy, ih) * iw + nearest_index(x, sx, iw)] / 256.0f;
      if (crop_y1 > crop_y0) v = v * ((y >= crop_y0 && y < crop_y1 && x >= crop_x0 && x < crop_x1) ? 1.0f : 0.0f);
    } else {                                                  // TartanAir: float32, nearest, black-pixel mask
      const float* p = reinterpret_cast<const float*>(raw) + n * ih * iw;
      v = p[(long long)nearest_index(y, sy, ih) * iw + nearest_index(x, sx, iw)];
      if (rgb != nullptr) {
        const float* q = rgb + idx * 3;
        const float nrm = sqrtf((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);   // reduce_euclidean_norm > 0
        v = v * (nrm > 0.f ? 1.0f : 0.0f);
      }
    }
    out[idx] = v;
  }
}

}  // namespace

extern "C" int m4d_decode_rgb8_resize(const uint8_t* images, int n, int ih, int iw, int oh, int ow, float* out,
                                      void* stream) {
  M4D_CHECK_ARG(images && out && n > 0 && ih > 0 && iw > 0 && oh > 0 && ow > 0);
  const long long total = (long long)n * oh * ow * 3;
  m4d_launch(rgb8_resize_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, images, ih, iw, oh, ow,
                     total, out);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_decode_depth_resize(const void* raw, int kind, int n, int ih, int iw, int oh, int ow,
                                       const float* rgb_resized, const int crop[4], float* out, void* stream) {
  M4D_CHECK_ARG(raw && out && n > 0 && ih > 0 && iw > 0 && oh > 0 && ow > 0 && kind >= 0 && kind <= 2);
  M4D_CHECK_ARG((((uintptr_t)raw) & (kind == 2 ? 3u : 1u)) == 0);
  const long long total = (long long)n * oh * ow;
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  if (crop) { c0 = crop[0]; c1 = crop[1]; c2 = crop[2]; c3 = crop[3]; }
  m4d_launch(depth_resize_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, raw, kind, ih, iw, oh,
                     ow, rgb_resized, c0, c1, c2, c3, total, out);
  return M4D_LAUNCH_RESULT();
}
