// Parallax <-> depth converters and the reprojection flow field
// (utils/depth_operations.py:72-215).  One lane per pixel; every map is
// [b,h,w,1] so consecutive lanes touch consecutive floats.  The reference runs
// each converter as ~25 separate TF ops over [b,h*w,3,1] tensors; here each is
// one pass: read 4 B, write 4 B per pixel.
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

enum { OP_P2D = 0, OP_D2P = 1, OP_PREV_D2PARA = 2, OP_RECOMPUTE = 3 };

template <int OP>
__global__ void __launch_bounds__(256)
converter_kernel(const float* __restrict__ in, const float* __restrict__ rot, int rot_c,
                 const float* __restrict__ trans, const float* __restrict__ cam_f,
                 const float* __restrict__ cam_c, int h, int w, float* __restrict__ out) {
  const int bi = blockIdx.y;
  const M4dMotion m = m4d_load_motion(OP == OP_PREV_D2PARA ? nullptr : rot, rot_c, trans, cam_f, cam_c, bi);
  const int hw = h * w;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
    const int i = p % w, j = p / w;
    const long long o = (long long)bi * hw + p;
    const float v = in[o];
    float r;
    if (OP == OP_P2D) {
      const M4dPixel px = m4d_pixel_factors(m, i, j);
      r = (px.s / v - m.tz) / px.alpha;                       // :164
    } else if (OP == OP_D2P) {
      const M4dPixel px = m4d_pixel_factors(m, i, j);
      r = px.s / (v * px.alpha + m.tz);                       // :192
    } else if (OP == OP_PREV_D2PARA) {
      const float mx = ((float)i + 0.5f) - m.cx;
      const float my = ((float)j + 0.5f) - m.cy;
      const float cx = (mx / m.fx) * m.fx;                    // :207 keeps the /f*f round trip
      const float cy = (my / m.fy) * m.fy;
      const float den = v - m.tz;
      const float dx = (m.stx - m.tz * cx) / den;             // :210
      const float dy = (m.sty - m.tz * cy) / den;
      r = sqrtf(dx * dx + dy * dy);                           // :213
    } else {
      const float mx = ((float)i + 0.5f) - m.cx;
      const float my = ((float)j + 0.5f) - m.cy;
      const float x = mx / m.fx, y = my / m.fy;
      const float t0 = -m.tx, t1 = -m.ty, t2 = -m.tz;         // :119
      const float tv = (m.r20 * t0 + m.r21 * t1) + m.r22 * t2;
      const float pr = (m.r20 * x + m.r21 * y) + m.r22;
      r = fminf(fmaxf(pr * v + tv, 0.1f), 2000.0f);           // :136-137
    }
    out[o] = r;
  }
}

// reproject's flow (utils/depth_operations.py:84-103).
__global__ void __launch_bounds__(256)
reproject_flow_kernel(const float* __restrict__ depth, const float* __restrict__ rot, int rot_c,
                      const float* __restrict__ trans, const float* __restrict__ cam_f,
                      const float* __restrict__ cam_c, int h, int w, float* __restrict__ flow,
                      float* __restrict__ pmr, float* __restrict__ rotc) {
  const int bi = blockIdx.y;
  const M4dMotion m = m4d_load_motion(rot, rot_c, trans, cam_f, cam_c, bi);
  // combined_mat = diag(fx,fy,1) @ [R|t], left-to-right sequential dot products.
  const float T[3][4] = {{m.r00, m.r01, m.r02, m.tx}, {m.r10, m.r11, m.r12, m.ty}, {m.r20, m.r21, m.r22, m.tz}};
  const float P[3][3] = {{m.fx, 0.f, 0.f}, {0.f, m.fy, 0.f}, {0.f, 0.f, 1.f}};
  float M[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float acc = P[r][0] * T[0][c];
      acc = acc + P[r][1] * T[1][c];
      acc = acc + P[r][2] * T[2][c];
      M[r][c] = acc;
    }
  const int hw = h * w;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
    const int i = p % w, j = p / w;
    const long long o = (long long)bi * hw + p;
    const float d = depth[o];
    const float mx = ((float)i + 0.5f) - m.cx;
    const float my = ((float)j + 0.5f) - m.cy;
    const float pos0 = (mx / m.fx) * d, pos1 = (my / m.fy) * d, pos2 = 1.0f * d;
    float pr[3], rr[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float acc = M[r][0] * pos0;
      acc = acc + M[r][1] * pos1;
      acc = acc + M[r][2] * pos2;
      rr[r] = acc;
      pr[r] = acc + M[r][3] * 1.0f;
    }
    const float px = pr[0] / pr[2], py = pr[1] / pr[2];
    const float rx = rr[0] / rr[2], ry = rr[1] / rr[2];
    flow[2 * o] = py - my;                                     // reversed to (row, col), :103
    flow[2 * o + 1] = px - mx;
    if (pmr) { pmr[2 * o] = px - rx; pmr[2 * o + 1] = py - ry; }
    if (rotc) { rotc[2 * o] = rx; rotc[2 * o + 1] = ry; }
  }
}

template <int OP>
int launch_converter(const float* in, const float* rot, int rot_c, const float* trans, const float* cam_f,
                     const float* cam_c, int b, int h, int w, float* out, void* stream) {
  M4D_CHECK_ARG(in && trans && cam_f && cam_c && out);
  M4D_CHECK_ARG(b > 0 && h > 0 && w > 0);
  if (OP != OP_PREV_D2PARA) M4D_CHECK_ARG(rot && (rot_c == 3 || rot_c == 4));
  int gx = m4d_blocks((long long)h * w, 256);
  if (gx > 4096) gx = 4096;
  m4d_launch(converter_kernel<OP>, dim3(gx, b), dim3(256), 0, (hipStream_t)stream,
                     in, rot, rot_c, trans, cam_f, cam_c, h, w, out);
  return M4D_LAUNCH_RESULT();
}

}  // namespace

extern "C" int m4d_parallax2depth(const float* disp, const float* rot, int rot_c, const float* trans,
                                  const float* cam_f, const float* cam_c, int b, int h, int w,
                                  float* out, void* stream) {
  return launch_converter<OP_P2D>(disp, rot, rot_c, trans, cam_f, cam_c, b, h, w, out, stream);
}

extern "C" int m4d_depth2parallax(const float* depth, const float* rot, int rot_c, const float* trans,
                                  const float* cam_f, const float* cam_c, int b, int h, int w,
                                  float* out, void* stream) {
  return launch_converter<OP_D2P>(depth, rot, rot_c, trans, cam_f, cam_c, b, h, w, out, stream);
}

extern "C" int m4d_prev_d2para(const float* prev_d, const float* rot, int rot_c, const float* trans,
                               const float* cam_f, const float* cam_c, int b, int h, int w,
                               float* out, void* stream) {
  (void)rot; (void)rot_c;
  return launch_converter<OP_PREV_D2PARA>(prev_d, nullptr, 0, trans, cam_f, cam_c, b, h, w, out, stream);
}

extern "C" int m4d_recompute_depth(const float* depth, const float* rot, int rot_c, const float* trans,
                                   const float* cam_f, const float* cam_c, int b, int h, int w,
                                   float* out, void* stream) {
  return launch_converter<OP_RECOMPUTE>(depth, rot, rot_c, trans, cam_f, cam_c, b, h, w, out, stream);
}

extern "C" int m4d_reproject_flow(const float* depth, const float* rot, int rot_c, const float* trans,
                                  const float* cam_f, const float* cam_c, int b, int h, int w,
                                  float* flow, float* proj_minus_rot, float* rot_coord, void* stream) {
  M4D_CHECK_ARG(depth && rot && trans && cam_f && cam_c && flow);
  M4D_CHECK_ARG(b > 0 && h > 0 && w > 0 && (rot_c == 3 || rot_c == 4));
  int gx = m4d_blocks((long long)h * w, 256);
  if (gx > 4096) gx = 4096;
  m4d_launch(reproject_flow_kernel, dim3(gx, b), dim3(256), 0, (hipStream_t)stream,
                     depth, rot, rot_c, trans, cam_f, cam_c, h, w, flow, proj_minus_rot, rot_coord);
  return M4D_LAUNCH_RESULT();
}
