// DSCV -- the parallax-sweeping cost volume (get_parallax_sweeping_cv,
// utils/depth_operations.py:224-281), fused.
//
// The reference materialises (2r+1) tiled copies of c1, of concat[c2, disp_prev_t]
// and of the flow field, warps them with 4 tf.gather passes and then multiplies /
// reduces in float16: ~17x the algorithmic HBM traffic.  Here
// one lane owns one (pixel, cut): it derives the 2r+1 query points in registers
// from the per-sample motion, gathers the 4 corners of its cut's channel run
// (NC contiguous floats, 16-byte loads) straight from the NHWC previous-frame
// features, lerps in float32, forms the float16 products and reduces them
// sequentially -- nothing but c1, c2, the two parallax maps and the outputs ever
// touches HBM.  Lanes of one pixel are adjacent (cut-minor), so a wave reads
// 64/k whole feature vectors per corner.
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

struct DscvArgs {
  const float* c1; const float* c2; const float* disp_prev_t; const float* disp;
  const float* rot; int rot_c; const float* trans; const float* cam_f; const float* cam_c;
  int h, w, C, r, k, nc, cv_accum;
  float* cv; int cv_stride; float* prev_disp; float* log_center; int log_stride; float log_scale;
  int32_t* index_out;
};

// NC > 0: channels per cut known at compile time (multiple of 4, float4 path).
// NC == 0: runtime channel count, scalar loads (any C, any alignment).
template <int NC>
__global__ void __launch_bounds__(256)
dscv_kernel(const DscvArgs a) {
  const int bi = blockIdx.y;
  const int hw = a.h * a.w;
  const int k = a.k;
  const int t_id = blockIdx.x * blockDim.x + threadIdx.x;
  if (t_id >= hw * k) return;
  const int kk = t_id % k;
  const int pix = t_id / k;
  const int i = pix % a.w, j = pix / a.w;
  const long long gp = (long long)bi * hw + pix;            // global pixel index
  const int nc = NC > 0 ? NC : a.nc;
  const int C = a.C;
  const int ncp = 2 * a.r + 1;

  const M4dMotion m = m4d_load_motion(a.rot, a.rot_c, a.trans, a.cam_f, a.cam_c, bi);
  const M4dPixel px = m4d_pixel_factors(m, i, j);
  const float start_x = px.x * m.fx;                          // :256
  const float start_y = px.y * m.fy;
  const float disp = a.disp[gp];

  // c1 of this cut, pre-rounded to half (:276).
  float c1h[NC > 0 ? NC : 1];
  const float* c1p = a.c1 + gp * C + kk * nc;
  if (NC > 0) {
#pragma unroll
    for (int c = 0; c < NC; c += 4) {
      const float4 v = *reinterpret_cast<const float4*>(c1p + c);
      c1h[c] = m4d_round_half(v.x); c1h[c + 1] = m4d_round_half(v.y);
      c1h[c + 2] = m4d_round_half(v.z); c1h[c + 3] = m4d_round_half(v.w);
    }
  }
  const float* c2b = a.c2 + (long long)bi * hw * C + kk * nc;
  const float* dpt = a.disp_prev_t + (long long)bi * hw;
  const long long rs = (long long)a.w * C;
  const float inv_n = (float)nc;

  for (int t = 0; t < ncp; ++t) {
    const float n = (float)(t - a.r);
    const float p = fminf(fmaxf(disp + n, 1e-6f), 1e6f);      // :235-236
    const float divider = px.s / p;                           // :262
    const float dxx = px.delta_x / divider;                   // :263
    const float dyy = px.delta_y / divider;
    const float flow_x = (px.proj_x + dxx) - start_x;         // :264
    const float flow_y = (px.proj_y + dyy) - start_y;
    const float qy = (float)j + flow_y;                       // dense_image_warp.py:244
    const float qx = (float)i + flow_x;
    int y0, x0;
    float ay, ax;
    m4d_bilinear_axis(qy, a.h, y0, ay);
    m4d_bilinear_axis(qx, a.w, x0, ax);
    const float* tl = c2b + ((long long)y0 * a.w + x0) * C;
    float acc = 0.0f;
    if (NC > 0) {
#pragma unroll
      for (int c = 0; c < NC; c += 4) {
        const float4 vtl = *reinterpret_cast<const float4*>(tl + c);
        const float4 vtr = *reinterpret_cast<const float4*>(tl + C + c);
        const float4 vbl = *reinterpret_cast<const float4*>(tl + rs + c);
        const float4 vbr = *reinterpret_cast<const float4*>(tl + rs + C + c);
        const float w0 = m4d_round_half(m4d_lerp2(vtl.x, vtr.x, vbl.x, vbr.x, ax, ay));
        const float w1 = m4d_round_half(m4d_lerp2(vtl.y, vtr.y, vbl.y, vbr.y, ax, ay));
        const float w2 = m4d_round_half(m4d_lerp2(vtl.z, vtr.z, vbl.z, vbr.z, ax, ay));
        const float w3 = m4d_round_half(m4d_lerp2(vtl.w, vtr.w, vbl.w, vbr.w, ax, ay));
        const float p0 = m4d_round_half(c1h[c] * w0), p1 = m4d_round_half(c1h[c + 1] * w1);
        const float p2 = m4d_round_half(c1h[c + 2] * w2), p3 = m4d_round_half(c1h[c + 3] * w3);
        if (a.cv_accum == 0) {
          if (c == 0) acc = p0; else acc = acc + p0;
          acc = acc + p1; acc = acc + p2; acc = acc + p3;
        } else {
          if (c == 0) acc = p0; else acc = m4d_round_half(acc + p0);
          acc = m4d_round_half(acc + p1); acc = m4d_round_half(acc + p2); acc = m4d_round_half(acc + p3);
        }
      }
    } else {
      for (int c = 0; c < nc; ++c) {
        const float wv = m4d_round_half(m4d_lerp2(tl[c], tl[C + c], tl[rs + c], tl[rs + C + c], ax, ay));
        const float pr = m4d_round_half(m4d_round_half(c1p[c]) * wv);
        if (c == 0) acc = pr;
        else acc = (a.cv_accum == 0) ? (acc + pr) : m4d_round_half(acc + pr);
      }
    }
    a.cv[gp * a.cv_stride + kk * ncp + t] = m4d_round_half(acc / inv_n);   // :277-278

    if (kk == 0) {
      if (a.index_out) {
        a.index_out[(gp * ncp + t) * 2] = y0;
        a.index_out[(gp * ncp + t) * 2 + 1] = x0;
      }
      const bool centre = (t == a.r) && a.log_center != nullptr;
      if (a.prev_disp != nullptr || centre) {
        const float* d0 = dpt + (long long)y0 * a.w + x0;     // the extra channel of :268
        const float wd = m4d_lerp2(d0[0], d0[1], d0[a.w], d0[a.w + 1], ax, ay);
        if (a.prev_disp) a.prev_disp[gp * ncp + t] = wd;
        if (centre) a.log_center[gp * a.log_stride] = logf(wd * a.log_scale);   // m4depth_network.py:238
      }
    }
  }
}

}  // namespace

extern "C" int m4d_dscv_fwd(const float* c1, const float* c2, const float* disp_prev_t, const float* disp,
                            const float* rot, int rot_c, const float* trans, const float* cam_f,
                            const float* cam_c, int b, int h, int w, int C, int search_range, int nbre_cuts,
                            int cv_accum, float* cv, int cv_stride, float* prev_disp,
                            float* log_center, int log_stride, float log_scale,
                            int32_t* index_out, void* stream) {
  M4D_CHECK_ARG(c1 && c2 && disp_prev_t && disp && rot && trans && cam_f && cam_c && cv);
  M4D_CHECK_ARG(b > 0 && h >= 2 && w >= 2 && C > 0 && search_range >= 0 && nbre_cuts > 0);
  M4D_CHECK_ARG(rot_c == 3 || rot_c == 4);
  M4D_CHECK_ARG(C % nbre_cuts == 0);
  M4D_CHECK_ARG(cv_accum == 0 || cv_accum == 1);
  M4D_CHECK_ARG(cv_stride >= nbre_cuts * (2 * search_range + 1));
  M4D_CHECK_ARG(log_center == nullptr || log_stride >= 1);
  DscvArgs a;
  a.c1 = c1; a.c2 = c2; a.disp_prev_t = disp_prev_t; a.disp = disp;
  a.rot = rot; a.rot_c = rot_c; a.trans = trans; a.cam_f = cam_f; a.cam_c = cam_c;
  a.h = h; a.w = w; a.C = C; a.r = search_range; a.k = nbre_cuts; a.nc = C / nbre_cuts; a.cv_accum = cv_accum;
  a.cv = cv; a.cv_stride = cv_stride; a.prev_disp = prev_disp; a.log_center = log_center;
  a.log_stride = log_stride; a.log_scale = log_scale; a.index_out = index_out;
  const long long threads = (long long)h * w * nbre_cuts;
  const dim3 grid(m4d_blocks(threads, 256), b), block(256);
  const bool aligned = (((uintptr_t)c1 | (uintptr_t)c2) & 15u) == 0 && (a.nc % 4 == 0);
  hipStream_t s = (hipStream_t)stream;
  if (aligned && a.nc == 16) hipLaunchKernelGGL(dscv_kernel<16>, grid, block, 0, s, a);
  else if (aligned && a.nc == 24) hipLaunchKernelGGL(dscv_kernel<24>, grid, block, 0, s, a);
  else if (aligned && a.nc == 32) hipLaunchKernelGGL(dscv_kernel<32>, grid, block, 0, s, a);
  else if (aligned && a.nc == 8) hipLaunchKernelGGL(dscv_kernel<8>, grid, block, 0, s, a);
  else if (aligned && a.nc == 4) hipLaunchKernelGGL(dscv_kernel<4>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(dscv_kernel<0>, grid, block, 0, s, a);
  return M4D_LAUNCH_RESULT();
}
