// DSCV -- the parallax-sweeping cost volume (get_parallax_sweeping_cv,
// utils/depth_operations.py:224-281), fused.
//
// The reference materialises (2r+1) tiled copies of c1, of concat[c2, disp_prev_t]
// and of the flow field, warps them with 4 tf.gather passes and then multiplies /
// reduces in float16: ~17x the algorithmic HBM traffic.  Here
// nothing but c1, c2, the two parallax maps and the outputs touches HBM.
//
// Wave layout (dscv_wave_kernel): the C channels of a pixel are spread over LP = C/4
// adjacent lanes, 16 bytes each, so a wave covers 64/LP neighbouring pixels and every
// corner fetch of a pixel is ONE contiguous 4*C-byte run (one or two cache lines)
// instead of C/4 strided 16-byte pieces.  The 2r+1 query points of a pixel are
// computed once, hypothesis t by lane t mod LP of the pixel, and broadcast with wave
// shuffles; the per-cut float16 products are summed in channel order by a shuffle
// chain over the G = LP/k lanes of the cut (same order as the sequential oracle, so
// the result is bit-identical).  Workgroups are remapped so that each XCD sweeps a
// contiguous band of the image: the gathers of neighbouring pixels then hit the
// XCD's own 4 MiB L2.
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

__device__ __forceinline__ void dscv_query(const M4dPixel& px, float start_x, float start_y, float disp,
                                           int i, int j, int t, int r, float& qy, float& qx) {
  const float n = (float)(t - r);
  const float p = fminf(fmaxf(disp + n, 1e-6f), 1e6f);      // :235-236
  const float divider = px.s / p;                           // :262
  const float dxx = px.delta_x / divider;                   // :263
  const float dyy = px.delta_y / divider;
  const float flow_x = (px.proj_x + dxx) - start_x;         // :264
  const float flow_y = (px.proj_y + dyy) - start_y;
  qy = (float)j + flow_y;                                   // dense_image_warp.py:244
  qx = (float)i + flow_x;
}

// LP lanes per pixel (C = 4*LP channels), G lanes per cut (nc = 4*G, k = LP/G cuts).
template <int LP, int G>
__global__ void __launch_bounds__(256)
dscv_wave_kernel(const DscvArgs a) {
  constexpr int PPW = 64 / LP;                 // pixels per wave
  constexpr int J = (16 + LP - 1) / LP;        // owned hypotheses per lane (supports 2r+1 <= 16)
  constexpr int NC = 4 * G;
  const int bi = blockIdx.y;
  const int hw = a.h * a.w;
  const int C = 4 * LP;
  const int ncp = 2 * a.r + 1;
  // XCD-aware remap: workgroup b runs on XCD b % 8; give each XCD a contiguous band.
  int blk = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) blk = (blk & 7) * (nb >> 3) + (blk >> 3);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slot = lane / LP;                  // pixel slot in the wave
  const int q = lane - slot * LP;              // float4 index inside the pixel's feature vector
  const int g = q % G;                         // position inside the cut's lane group
  const int kk = q / G;
  const bool lane_on = slot < PPW;
  int pix = (blk * 4 + wave) * PPW + slot;
  const bool active = lane_on && pix < hw;
  if (!active) pix = hw - 1;                   // keep addresses valid; results are discarded
  const int i = pix % a.w, j = pix / a.w;
  const long long gp = (long long)bi * hw + pix;
  const int base_lane = slot * LP;

  const M4dMotion m = m4d_load_motion(a.rot, a.rot_c, a.trans, a.cam_f, a.cam_c, bi);
  const M4dPixel px = m4d_pixel_factors(m, i, j);
  const float start_x = px.x * m.fx;           // :256
  const float start_y = px.y * m.fy;
  const float disp = a.disp[gp];

  // hypotheses owned by this lane: t = jj*LP + q
  float oqy[J], oqx[J];
#pragma unroll
  for (int jj = 0; jj < J; ++jj) {
    const int t = jj * LP + q;
    oqy[jj] = 0.f; oqx[jj] = 0.f;
    if (jj * LP < ncp)                          // wave-uniform: skip rounds no hypothesis falls in
      dscv_query(px, start_x, start_y, disp, i, j, t < ncp ? t : 0, a.r, oqy[jj], oqx[jj]);
  }

  // this lane's 4 channels of c1, pre-rounded to half (:276)
  const float4 c1v = *reinterpret_cast<const float4*>(a.c1 + gp * C + 4 * q);
  const float c1a = m4d_round_half(c1v.x), c1b = m4d_round_half(c1v.y);
  const float c1c = m4d_round_half(c1v.z), c1d = m4d_round_half(c1v.w);
  const float* c2b = a.c2 + (long long)bi * hw * C + 4 * q;
  const float* dpt = a.disp_prev_t + (long long)bi * hw;
  const long long rs = (long long)a.w * C;
  const float n_c = (float)NC;
  const bool seq16 = a.cv_accum != 0;

#pragma unroll
  for (int jj = 0; jj < J; ++jj) {
#pragma unroll
    for (int qo = 0; qo < LP; ++qo) {
      const int t = jj * LP + qo;
      if (t >= ncp) break;                      // wave-uniform
      const float qy = __shfl(oqy[jj], base_lane + qo);
      const float qx = __shfl(oqx[jj], base_lane + qo);
      int y0, x0;
      float ay, ax;
      m4d_bilinear_axis(qy, a.h, y0, ay);
      m4d_bilinear_axis(qx, a.w, x0, ax);
      const float* tl = c2b + ((long long)y0 * a.w + x0) * C;
      const float4 vtl = *reinterpret_cast<const float4*>(tl);
      const float4 vtr = *reinterpret_cast<const float4*>(tl + C);
      const float4 vbl = *reinterpret_cast<const float4*>(tl + rs);
      const float4 vbr = *reinterpret_cast<const float4*>(tl + rs + C);
      const float p0 = m4d_round_half(c1a * m4d_round_half(m4d_lerp2(vtl.x, vtr.x, vbl.x, vbr.x, ax, ay)));
      const float p1 = m4d_round_half(c1b * m4d_round_half(m4d_lerp2(vtl.y, vtr.y, vbl.y, vbr.y, ax, ay)));
      const float p2 = m4d_round_half(c1c * m4d_round_half(m4d_lerp2(vtl.z, vtr.z, vbl.z, vbr.z, ax, ay)));
      const float p3 = m4d_round_half(c1d * m4d_round_half(m4d_lerp2(vtl.w, vtr.w, vbl.w, vbr.w, ax, ay)));
      // sequential (channel-order) sum across the G lanes of the cut
      float acc;
      if (!seq16) {
        acc = ((p0 + p1) + p2) + p3;
#pragma unroll
        for (int s = 1; s < G; ++s) {
          const float prev = __shfl_up(acc, 1);
          if (g == s) acc = (((prev + p0) + p1) + p2) + p3;
        }
      } else {
        acc = m4d_round_half(m4d_round_half(m4d_round_half(p0 + p1) + p2) + p3);
#pragma unroll
        for (int s = 1; s < G; ++s) {
          const float prev = __shfl_up(acc, 1);
          if (g == s)
            acc = m4d_round_half(m4d_round_half(m4d_round_half(m4d_round_half(prev + p0) + p1) + p2) + p3);
        }
      }
      if (active && g == G - 1)
        a.cv[gp * a.cv_stride + kk * ncp + t] = m4d_round_half(acc / n_c);      // :277-278
      if (active && q == 0) {
        if (a.index_out) {
          a.index_out[(gp * ncp + t) * 2] = y0;
          a.index_out[(gp * ncp + t) * 2 + 1] = x0;
        }
        const bool centre = (t == a.r) && a.log_center != nullptr;
        if (a.prev_disp != nullptr || centre) {
          const float* d0 = dpt + (long long)y0 * a.w + x0;                      // the extra channel of :268
          const float wd = m4d_lerp2(d0[0], d0[1], d0[a.w], d0[a.w + 1], ax, ay);
          if (a.prev_disp) a.prev_disp[gp * ncp + t] = wd;
          if (centre) a.log_center[gp * a.log_stride] = logf(wd * a.log_scale);  // m4depth_network.py:238
        }
      }
    }
  }
}

// Any C / cuts / alignment / search range: one lane per (pixel, cut), scalar loads.
__global__ void __launch_bounds__(256)
dscv_generic_kernel(const DscvArgs a) {
  const int bi = blockIdx.y;
  const int hw = a.h * a.w;
  const int k = a.k;
  const int t_id = blockIdx.x * blockDim.x + threadIdx.x;
  if (t_id >= hw * k) return;
  const int kk = t_id % k;
  const int pix = t_id / k;
  const int i = pix % a.w, j = pix / a.w;
  const long long gp = (long long)bi * hw + pix;
  const int nc = a.nc;
  const int C = a.C;
  const int ncp = 2 * a.r + 1;
  const M4dMotion m = m4d_load_motion(a.rot, a.rot_c, a.trans, a.cam_f, a.cam_c, bi);
  const M4dPixel px = m4d_pixel_factors(m, i, j);
  const float start_x = px.x * m.fx;
  const float start_y = px.y * m.fy;
  const float disp = a.disp[gp];
  const float* c1p = a.c1 + gp * C + kk * nc;
  const float* c2b = a.c2 + (long long)bi * hw * C + kk * nc;
  const float* dpt = a.disp_prev_t + (long long)bi * hw;
  const long long rs = (long long)a.w * C;
  for (int t = 0; t < ncp; ++t) {
    float qy, qx;
    dscv_query(px, start_x, start_y, disp, i, j, t, a.r, qy, qx);
    int y0, x0;
    float ay, ax;
    m4d_bilinear_axis(qy, a.h, y0, ay);
    m4d_bilinear_axis(qx, a.w, x0, ax);
    const float* tl = c2b + ((long long)y0 * a.w + x0) * C;
    float acc = 0.0f;
    for (int c = 0; c < nc; ++c) {
      const float wv = m4d_round_half(m4d_lerp2(tl[c], tl[C + c], tl[rs + c], tl[rs + C + c], ax, ay));
      const float pr = m4d_round_half(m4d_round_half(c1p[c]) * wv);
      if (c == 0) acc = pr;
      else acc = (a.cv_accum == 0) ? (acc + pr) : m4d_round_half(acc + pr);
    }
    a.cv[gp * a.cv_stride + kk * ncp + t] = m4d_round_half(acc / (float)nc);
    if (kk == 0) {
      if (a.index_out) {
        a.index_out[(gp * ncp + t) * 2] = y0;
        a.index_out[(gp * ncp + t) * 2 + 1] = x0;
      }
      const bool centre = (t == a.r) && a.log_center != nullptr;
      if (a.prev_disp != nullptr || centre) {
        const float* d0 = dpt + (long long)y0 * a.w + x0;
        const float wd = m4d_lerp2(d0[0], d0[1], d0[a.w], d0[a.w + 1], ax, ay);
        if (a.prev_disp) a.prev_disp[gp * ncp + t] = wd;
        if (centre) a.log_center[gp * a.log_stride] = logf(wd * a.log_scale);
      }
    }
  }
}

template <int LP, int G>
void launch_wave(const DscvArgs& a, int b, hipStream_t s) {
  constexpr int PPW = 64 / LP;
  const int hw = a.h * a.w;
  int nb = (hw + 4 * PPW - 1) / (4 * PPW);
  hipLaunchKernelGGL((dscv_wave_kernel<LP, G>), dim3(nb, b), dim3(256), 0, s, a);
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
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = (((uintptr_t)c1 | (uintptr_t)c2) & 15u) == 0 && (a.nc % 4 == 0);
  const int lp = C / 4, g = a.nc / 4;
  const bool fits = aligned && (2 * search_range + 1) <= 16;
  // (LP, G) pairs of the 6-level pyramid (C = 16..192, cuts 1,2,2,4,4,8) plus the small
  // shapes the unit tests use; everything else takes the generic kernel.
  if (fits && lp == 4 && g == 4) launch_wave<4, 4>(a, b, s);          // C=16  k=1
  else if (fits && lp == 8 && g == 4) launch_wave<8, 4>(a, b, s);     // C=32  k=2
  else if (fits && lp == 16 && g == 8) launch_wave<16, 8>(a, b, s);   // C=64  k=2
  else if (fits && lp == 24 && g == 6) launch_wave<24, 6>(a, b, s);   // C=96  k=4
  else if (fits && lp == 32 && g == 8) launch_wave<32, 8>(a, b, s);   // C=128 k=4
  else if (fits && lp == 48 && g == 6) launch_wave<48, 6>(a, b, s);   // C=192 k=8
  else if (fits && lp == 8 && g == 8) launch_wave<8, 8>(a, b, s);     // C=32  k=1
  else if (fits && lp == 4 && g == 2) launch_wave<4, 2>(a, b, s);     // C=16  k=2
  else {
    const long long threads = (long long)h * w * nbre_cuts;
    hipLaunchKernelGGL(dscv_generic_kernel, dim3(m4d_blocks(threads, 256), b), dim3(256), 0, s, a);
  }
  return M4D_LAUNCH_RESULT();
}
