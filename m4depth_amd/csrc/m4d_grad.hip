// Backward passes of the two cost volumes -- the gradients TensorFlow's autodiff derives
// from the graphs of get_parallax_sweeping_cv (utils/depth_operations.py:224-281) and
// cost_volume (:284-313), which train_step differentiates (m4depth_network.py:371-399).
//
// DSCV: for pixel p, hypothesis t, cut kk the forward is
//     cv = f32( mean_c f16(c1[p,c]) * f16(warp(c2)[p,t,c]) ),  prev_disp = warp(disp_prev_t)[p,t]
// with warp = the border-replicate bilinear of dense_image_warp.py:61-192 at the query point
// q(p,t) that depends on disp[p].  TF differentiates the half-precision part in half precision
// (cast / mean / multiply gradients run in float16), the warp into four scatter-adds
// (gather gradients) plus the alpha terms (floor has no gradient; clip passes inside [0,1]),
// and the query point back into disp through `delta / (s / disp)` (:262-264).  prev_d2para's
// result is wrapped in stop_gradient (:215) but the op's signature does not know that, so
// g_disp_prev_t is an optional output.
//
// Same wave layout as the forward kernel: LP = C/4 adjacent lanes own the 16-byte pieces of
// one pixel's feature vector, so every corner access (load of c2, atomic add into g_c2) of a
// pixel is one contiguous 4*C-byte run.  g_c1 / g_disp are owned by the pixel (plain stores,
// deterministic); g_c2 is a scatter: hardware float32 atomic adds (return-less, executed in
// L2), like the reference's BackProjectBackward (backproject_op_gpu.cu.cc:108-197).
//
// SNCV: out = leaky_relu(mean_c c1[p,c] * c2pad[p+d,c]); both gradients are gathers over
// the (2r+1)^2 window, one lane per (pixel, 4 channels), deterministic.
#include <cstdlib>
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

struct DscvBwdArgs {
  const float* c1; const float* c2; const float* disp_prev_t; const float* disp;
  const float* rot; int rot_c; const float* trans; const float* cam_f; const float* cam_c;
  int h, w, C, r, k;
  const float* g_cv; int g_cv_stride; const float* g_prev_disp;
  float* g_c1; float* g_c2; float* g_disp; float* g_disp_prev_t;
  int bw_log2, bh_log2, blocks_x, blocks_y;          // a workgroup's pixels: a 2^bw x 2^bh block (= 4 waves x 64 / LP pixels)
  int win_floats;                                    // LDS floats of the private g_c2 window (0 = always scatter straight to HBM)
};

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  __builtin_amdgcn_global_atomic_fadd_f32(p, v);
}
__device__ __forceinline__ void lds_add_f32(float* p, float v) {     // ds_add_f32 (no return value)
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// sum of v over the LP lanes [base, base+LP) of this wave (every lane gets the result)
__device__ __forceinline__ float group_sum(float v, int LP, int base, int q) {
  if ((LP & (LP - 1)) == 0) {
    for (int off = 1; off < LP; off <<= 1) v += __shfl_xor(v, off);
    return v;
  }
  float s = 0.f;
  for (int o = 0; o < LP; ++o) s += __shfl(v, base + o);    // same order on every lane
  return s;
}

// The g_c2 scatter, privatised (round 5).  The 9 queries of a pixel step ~1 pixel along its epipolar line and neighbouring pixels'
// queries are ~1 pixel apart, so the 9 x 4 corner cells of a workgroup's pixel block overlap heavily: the workgroup first finds the
// bounding box of its cells; if that window (x C floats) fits its LDS budget the 16 adds per lane and hypothesis go to LDS
// (ds_add_f32) and the window is flushed ONCE with global atomics (non-zero cells only) -- ~10x fewer L2 atomics on typical motion;
// a block whose cells spread wider (large rotation, epipole inside the block) scatters straight to HBM as before.  Either way the
// adds commute only up to float32 rounding (as in the reference's BackProjectBackward): same tolerance as before.
__global__ void __launch_bounds__(256)
dscv_bwd_kernel(const DscvBwdArgs a) {
  extern __shared__ __align__(16) float win[];
  __shared__ int box[4][4];
  const int C = a.C, LP = C >> 2;
  const int PPW = 64 / LP;
  const int bi = blockIdx.y;
  const int hw = a.h * a.w;
  int blk = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) blk = (blk & 7) * (nb >> 3) + (blk >> 3);     // one image band per XCD
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slot = lane / LP;
  const int q = lane - slot * LP;
  const int base = slot * LP;
  const int pb = wave * PPW + slot;                                // pixel of the block
  const int bx = blk % a.blocks_x, by = blk / a.blocks_x;
  int i = (bx << a.bw_log2) + (pb & ((1 << a.bw_log2) - 1)), j = (by << a.bh_log2) + (pb >> a.bw_log2);
  const bool active = slot < PPW && pb < (1 << (a.bw_log2 + a.bh_log2)) && i < a.w && j < a.h;
  if (!active) { i = min(i, a.w - 1); j = min(j, a.h - 1); }
  const int pix = j * a.w + i;
  const int ncp = 2 * a.r + 1;
  const int nc = C / a.k;
  const int kk = (4 * q) / nc;                                    // nc is a multiple of 4
  const long long gp = (long long)bi * hw + pix;

  const M4dMotion m = m4d_load_motion(a.rot, a.rot_c, a.trans, a.cam_f, a.cam_c, bi);
  const M4dPixel px = m4d_pixel_factors(m, i, j);
  const float start_x = px.x * m.fx, start_y = px.y * m.fy;
  const float disp = a.disp[gp];
  const float4 c1v = *reinterpret_cast<const float4*>(a.c1 + gp * C + 4 * q);
  const float c1h[4] = {m4d_round_half(c1v.x), m4d_round_half(c1v.y), m4d_round_half(c1v.z), m4d_round_half(c1v.w)};
  const float* c2b = a.c2 + (long long)bi * hw * C + 4 * q;
  float* g2b = a.g_c2 + (long long)bi * hw * C + 4 * q;
  const float* dpt = a.disp_prev_t + (long long)bi * hw;
  float* gdpt = a.g_disp_prev_t ? a.g_disp_prev_t + (long long)bi * hw : nullptr;
  const long long rs = (long long)a.w * C;
  const float n_half = (float)nc;                                  // exact in half for nc <= 2048

  // ---- the bounding box of the block's corner cells (the same query arithmetic as the loop below)
  bool priv = false;
  int wy0 = 0, wx0 = 0, ww = 0, wcells = 0;
  if (a.win_floats > 0) {
    int ymin = 1 << 30, ymax = -1, xmin = 1 << 30, xmax = -1;
    if (active) {
      for (int t = 0; t < ncp; ++t) {
        const float p = fminf(fmaxf(disp + (float)(t - a.r), 1e-6f), 1e6f);
        const float divider = px.s / p;
        const float qx = (float)i + ((px.proj_x + px.delta_x / divider) - start_x);
        const float qy = (float)j + ((px.proj_y + px.delta_y / divider) - start_y);
        int y0, x0; float ay, ax;
        m4d_bilinear_axis(qy, a.h, y0, ay);
        m4d_bilinear_axis(qx, a.w, x0, ax);
        ymin = min(ymin, y0); ymax = max(ymax, y0); xmin = min(xmin, x0); xmax = max(xmax, x0);
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      ymin = min(ymin, __shfl_xor(ymin, off)); ymax = max(ymax, __shfl_xor(ymax, off));
      xmin = min(xmin, __shfl_xor(xmin, off)); xmax = max(xmax, __shfl_xor(xmax, off));
    }
    if (lane == 0) { box[wave][0] = ymin; box[wave][1] = ymax; box[wave][2] = xmin; box[wave][3] = xmax; }
    __syncthreads();
    ymin = min(min(box[0][0], box[1][0]), min(box[2][0], box[3][0])); ymax = max(max(box[0][1], box[1][1]), max(box[2][1], box[3][1]));
    xmin = min(min(box[0][2], box[1][2]), min(box[2][2], box[3][2])); xmax = max(max(box[0][3], box[1][3]), max(box[2][3], box[3][3]));
    const int wh = ymax - ymin + 2;                                // + the bottom / right corners
    ww = xmax - xmin + 2;
    priv = ymax >= 0 && (long long)wh * ww * C <= a.win_floats;    // (uniform)
    if (priv) {
      wy0 = ymin; wx0 = xmin; wcells = wh * ww;
      const int n4 = (wh * ww * C) >> 2;
      for (int e = threadIdx.x; e < n4; e += 256) reinterpret_cast<float4*>(win)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      __syncthreads();
    }
  }

  float gc1[4] = {0.f, 0.f, 0.f, 0.f};
  float gdisp = 0.f;
  for (int t = 0; t < ncp; ++t) {
    const float nn = (float)(t - a.r);
    const float praw = disp + nn;
    const float p = fminf(fmaxf(praw, 1e-6f), 1e6f);
    const float divider = px.s / p;
    const float dxx = px.delta_x / divider, dyy = px.delta_y / divider;
    const float qx = (float)i + ((px.proj_x + dxx) - start_x);
    const float qy = (float)j + ((px.proj_y + dyy) - start_y);
    int y0, x0; float ay, ax;
    m4d_bilinear_axis(qy, a.h, y0, ay);
    m4d_bilinear_axis(qx, a.w, x0, ax);
    const float ry = qy - (float)y0, rx = qx - (float)x0;          // clip_by_value passes inside [0,1]
    const bool pass_y = ry >= 0.f && ry <= 1.f, pass_x = rx >= 0.f && rx <= 1.f;

    const long long off = ((long long)y0 * a.w + x0) * C;
    const float4 vtl = *reinterpret_cast<const float4*>(c2b + off);
    const float4 vtr = *reinterpret_cast<const float4*>(c2b + off + C);
    const float4 vbl = *reinterpret_cast<const float4*>(c2b + off + rs);
    const float4 vbr = *reinterpret_cast<const float4*>(c2b + off + rs + C);
    const float tl[4] = {vtl.x, vtl.y, vtl.z, vtl.w}, tr[4] = {vtr.x, vtr.y, vtr.z, vtr.w};
    const float bl[4] = {vbl.x, vbl.y, vbl.z, vbl.w}, br[4] = {vbr.x, vbr.y, vbr.z, vbr.w};

    // float16 part of the graph, differentiated in float16 (:276-278)
    const float g16 = m4d_round_half(a.g_cv[gp * a.g_cv_stride + kk * ncp + t]);
    const float gm = m4d_round_half(g16 / n_half);
    float day = 0.f, dax = 0.f;
    float dtl[4], dtr[4], dbl[4], dbr[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float top = ax * (tr[c] - tl[c]) + tl[c];
      const float bot = ax * (br[c] - bl[c]) + bl[c];
      const float wv = m4d_round_half(ay * (bot - top) + top);
      gc1[c] += m4d_round_half(gm * wv);
      const float gw = m4d_round_half(gm * c1h[c]);                // gradient w.r.t. the warped c2 value
      const float gbot = ay * gw, gtop = gw - gbot;
      day += gw * (bot - top);
      dax += gtop * (tr[c] - tl[c]) + gbot * (br[c] - bl[c]);
      dtr[c] = ax * gtop; dtl[c] = gtop - dtr[c];
      dbr[c] = ax * gbot; dbl[c] = gbot - dbr[c];
    }
    if (active) {
      if (priv) {
        float* wp = win + ((y0 - wy0) * ww + (x0 - wx0)) * C + 4 * q;
        const int wrs = ww * C;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          lds_add_f32(wp + c, dtl[c]);
          lds_add_f32(wp + C + c, dtr[c]);
          lds_add_f32(wp + wrs + c, dbl[c]);
          lds_add_f32(wp + wrs + C + c, dbr[c]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          atomic_add_f32(g2b + off + c, dtl[c]);
          atomic_add_f32(g2b + off + C + c, dtr[c]);
          atomic_add_f32(g2b + off + rs + c, dbl[c]);
          atomic_add_f32(g2b + off + rs + C + c, dbr[c]);
        }
      }
    }
    // the extra channel of :268 (disp_prev_t), handled by the pixel's first lane
    if (q == 0 && a.g_prev_disp != nullptr) {
      const float gw = a.g_prev_disp[gp * ncp + t];
      const float* d0 = dpt + (long long)y0 * a.w + x0;
      const float e_tl = d0[0], e_tr = d0[1], e_bl = d0[a.w], e_br = d0[a.w + 1];
      const float top = ax * (e_tr - e_tl) + e_tl;
      const float bot = ax * (e_br - e_bl) + e_bl;
      const float gbot = ay * gw, gtop = gw - gbot;
      day += gw * (bot - top);
      dax += gtop * (e_tr - e_tl) + gbot * (e_br - e_bl);
      if (gdpt != nullptr && active) {
        float* g0 = gdpt + (long long)y0 * a.w + x0;
        const float etr = ax * gtop, ebr = ax * gbot;
        atomic_add_f32(g0, gtop - etr);
        atomic_add_f32(g0 + 1, etr);
        atomic_add_f32(g0 + a.w, gbot - ebr);
        atomic_add_f32(g0 + a.w + 1, ebr);
      }
    }
    day = group_sum(day, LP, base, q);
    dax = group_sum(dax, LP, base, q);
    const float dqy = pass_y ? day : 0.f, dqx = pass_x ? dax : 0.f;
    // q = grid + proj + delta/divider - start ; divider = s / p  (:262-264)
    const float ddiv = -((dqx * dxx + dqy * dyy) / divider);
    const float dp = -(ddiv * divider) / p;
    if (praw >= 1e-6f && praw <= 1e6f) gdisp += dp;                // :236 clip_by_value
  }
  if (active) {
    *reinterpret_cast<float4*>(a.g_c1 + gp * C + 4 * q) = make_float4(gc1[0], gc1[1], gc1[2], gc1[3]);
    if (q == 0) a.g_disp[gp] = gdisp;
  }
  if (priv) {                                                      // ---- the window goes out once: non-zero values only
    __syncthreads();
    float* g2i = a.g_c2 + (long long)bi * hw * C;
    const int n = wcells * C;
    for (int e = threadIdx.x; e < n; e += 256) {
      const float v = win[e];
      if (v != 0.f) {
        const int cell = e / C, ch = e - cell * C;
        const int cy = wy0 + cell / ww, cx = wx0 + cell % ww;
        atomic_add_f32(g2i + ((long long)cy * a.w + cx) * C + ch, v);
      }
    }
  }
}

struct SncvBwdArgs {
  const float* c1; const float* c2; const float* out; int out_stride; const float* g; int g_stride;
  int h, w, C, r, d, k; float slope;
  float* g_c1; float* g_c2;
};

// One lane per (pixel, 4 channels).  g' = g * leaky_relu'(out) / nc.
//   g_c1[p,c] = sum_d g'[p, d]            * c2[p + d, c]        (zero outside the image: tf.pad)
//   g_c2[p,c] = sum_d g'[p - d, d]        * c1[p - d, c]
__global__ void __launch_bounds__(256)
sncv_bwd_kernel(const SncvBwdArgs a, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c4n = a.C >> 2;
  const int c4 = (int)(idx % c4n);
  const long long gp = idx / c4n;
  const int hw = a.h * a.w;
  const int bi = (int)(gp / hw);
  const int pix = (int)(gp - (long long)bi * hw);
  const int jy = pix / a.w, ix = pix % a.w;
  const int nc = a.C / a.k;
  const int kk = (4 * c4) / nc;
  const int mo = 2 * a.r + 1;
  const float n_c = (float)nc;
  const long long ib = (long long)bi * hw;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int y = 0; y < mo; ++y) {
    for (int x = 0; x < mo; ++x) {
      const int ch = (y * mo + x) * a.k + kk;
      const int dy = (y - a.r) * a.d, dx = (x - a.r) * a.d;
      if (a.g_c1) {
        const int ny = jy + dy, nx = ix + dx;
        if (ny >= 0 && ny < a.h && nx >= 0 && nx < a.w) {
          const float o = a.out[gp * a.out_stride + ch];
          const float gg = (a.g[gp * a.g_stride + ch] * (o > 0.f ? 1.0f : a.slope)) / n_c;
          const float4 v = *reinterpret_cast<const float4*>(a.c2 + (ib + (long long)ny * a.w + nx) * a.C + 4 * c4);
          s1[0] += gg * v.x; s1[1] += gg * v.y; s1[2] += gg * v.z; s1[3] += gg * v.w;
        }
      }
      if (a.g_c2) {
        const int sy = jy - dy, sx = ix - dx;                      // the pixel whose window reaches (jy,ix) at (y,x)
        if (sy >= 0 && sy < a.h && sx >= 0 && sx < a.w) {
          const long long sp = ib + (long long)sy * a.w + sx;
          const float o = a.out[sp * a.out_stride + ch];
          const float gg = (a.g[sp * a.g_stride + ch] * (o > 0.f ? 1.0f : a.slope)) / n_c;
          const float4 v = *reinterpret_cast<const float4*>(a.c1 + sp * a.C + 4 * c4);
          s2[0] += gg * v.x; s2[1] += gg * v.y; s2[2] += gg * v.z; s2[3] += gg * v.w;
        }
      }
    }
  }
  if (a.g_c1) *reinterpret_cast<float4*>(a.g_c1 + gp * a.C + 4 * c4) = make_float4(s1[0], s1[1], s1[2], s1[3]);
  if (a.g_c2) *reinterpret_cast<float4*>(a.g_c2 + gp * a.C + 4 * c4) = make_float4(s2[0], s2[1], s2[2], s2[3]);
}

}  // namespace

extern "C" int m4d_dscv_bwd(const float* c1, const float* c2, const float* disp_prev_t, const float* disp,
                            const float* rot, int rot_c, const float* trans, const float* cam_f,
                            const float* cam_c, int b, int h, int w, int C, int search_range, int nbre_cuts,
                            const float* g_cv, int g_cv_stride, const float* g_prev_disp,
                            float* g_c1, float* g_c2, float* g_disp, float* g_disp_prev_t, void* stream) {
  M4D_CHECK_ARG(c1 && c2 && disp_prev_t && disp && trans && cam_f && cam_c && g_cv && g_c1 && g_c2 && g_disp);
  M4D_CHECK_ARG(b > 0 && h >= 2 && w >= 2 && C > 0 && search_range >= 0 && nbre_cuts > 0);
  M4D_CHECK_ARG(C % 4 == 0 && C <= 256 && C % nbre_cuts == 0 && (C / nbre_cuts) % 4 == 0);
  M4D_CHECK_ARG(rot == nullptr || rot_c == 3 || rot_c == 4);
  M4D_CHECK_ARG(g_cv_stride >= nbre_cuts * (2 * search_range + 1));
  M4D_CHECK_ARG(((((uintptr_t)c1 | (uintptr_t)c2 | (uintptr_t)g_c1 | (uintptr_t)g_c2)) & 15u) == 0);
  hipStream_t s = (hipStream_t)stream;
  const size_t fbytes = (size_t)b * h * w * C * sizeof(float);
#if M4D_EXPERIMENTS
  if (m4d_tape_recording()) return (int)hipErrorNotSupported;      // a memset is not a kernel launch: it would be lost from the replay
#endif
  hipError_t e = hipMemsetAsync(g_c2, 0, fbytes, s);
  if (e != hipSuccess) return (int)e;
  if (g_disp_prev_t) {
    e = hipMemsetAsync(g_disp_prev_t, 0, (size_t)b * h * w * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
  }
  DscvBwdArgs a;
  a.c1 = c1; a.c2 = c2; a.disp_prev_t = disp_prev_t; a.disp = disp; a.rot = rot; a.rot_c = rot_c;
  a.trans = trans; a.cam_f = cam_f; a.cam_c = cam_c; a.h = h; a.w = w; a.C = C; a.r = search_range;
  a.k = nbre_cuts; a.g_cv = g_cv; a.g_cv_stride = g_cv_stride; a.g_prev_disp = g_prev_disp;
  a.g_c1 = g_c1; a.g_c2 = g_c2; a.g_disp = g_disp; a.g_disp_prev_t = g_disp_prev_t;
  const int LP = C / 4, PPW = 64 / LP;
  M4D_CHECK_ARG(PPW >= 1);
  // a workgroup's 4 * PPW pixels as a block (squarer blocks = smaller windows whatever the epipolar direction); PPW is a power
  // of two only for LP = 4, 8, 16, 32, 64 -- 24 / 48 lanes per pixel (C = 96 / 192) leave idle lanes and 8 / 4 pixels per block
  int ppb_log2 = 0;
  while ((2 << ppb_log2) <= 4 * PPW) ++ppb_log2;
  a.bh_log2 = ppb_log2 / 2; a.bw_log2 = ppb_log2 - a.bh_log2;
  // (4 * PPW not a power of two -- e.g. 12 lanes per pixel -- : the block holds the power of two below, the other slots idle)
  a.blocks_x = (w + (1 << a.bw_log2) - 1) >> a.bw_log2; a.blocks_y = (h + (1 << a.bh_log2) - 1) >> a.bh_log2;
  static const int win_kb = [] { const char* e = getenv("M4D_DSCV_BWD_WINDOW_KB"); const int v = e ? atoi(e) : 48; return v < 0 ? 0 : (v > 64 ? 64 : v); }();
  a.win_floats = win_kb * 256;                                     // 0 = the round-2 kernel's direct scatter (A/B)
  dim3 grid((unsigned)(a.blocks_x * a.blocks_y), b);
  m4d_launch(dscv_bwd_kernel, grid, dim3(256), (size_t)a.win_floats * sizeof(float), s, a);
  return M4D_LAUNCH_RESULT();
}

extern "C" int m4d_sncv_bwd(const float* c1, const float* c2, const float* out, int out_stride,
                            const float* g, int g_stride, int b, int h, int w, int C, int search_range,
                            int dilation_rate, int nbre_cuts, float slope, float* g_c1, float* g_c2,
                            void* stream) {
  M4D_CHECK_ARG(c1 && c2 && out && g && (g_c1 || g_c2));
  M4D_CHECK_ARG(b > 0 && h > 0 && w > 0 && C > 0 && search_range >= 0 && dilation_rate >= 1 && nbre_cuts > 0);
  M4D_CHECK_ARG(C % 4 == 0 && C % nbre_cuts == 0 && (C / nbre_cuts) % 4 == 0);
  const int mo = 2 * search_range + 1;
  M4D_CHECK_ARG(out_stride >= mo * mo * nbre_cuts && g_stride >= mo * mo * nbre_cuts);
  M4D_CHECK_ARG(((((uintptr_t)c1 | (uintptr_t)c2 | (uintptr_t)g_c1 | (uintptr_t)g_c2)) & 15u) == 0);
  SncvBwdArgs a;
  a.c1 = c1; a.c2 = c2; a.out = out; a.out_stride = out_stride; a.g = g; a.g_stride = g_stride;
  a.h = h; a.w = w; a.C = C; a.r = search_range; a.d = dilation_rate; a.k = nbre_cuts; a.slope = slope;
  a.g_c1 = g_c1; a.g_c2 = g_c2;
  const long long total = (long long)b * h * w * (C / 4);
  m4d_launch(sncv_bwd_kernel, dim3(m4d_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, a, total);
  return M4D_LAUNCH_RESULT();
}
