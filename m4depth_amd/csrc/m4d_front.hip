// The fused level front: everything DepthEstimatorLevel.call does between the encoder features and the refiner
// convolutions (m4depth_network.py:179-242) in ONE kernel --
//   per-cut normalisation of the current features (:179-189) and its store into the temporal memory (:211, :259),
//   x2 legacy-bilinear upsampling of the coarser level's estimate (:202-204), log-parallax feature (:224), memory
//   features (:227), prev_d2para of the depth memory (:218; evaluated only where the DSCV samples it),
//   DSCV get_parallax_sweeping_cv (:220-221, utils/depth_operations.py:224-281, incl. the bilinear warp of
//   utils/dense_image_warp.py), SNCV cost_volume (:232, depth_operations.py:284-313), time-recurrence feature (:238)
// -- assembling every pixel's whole refiner-input row [cv | log para_l | other(4) | sncv | log para_t] in LDS and
// writing it as contiguous rows (full cache lines).  It replaces three launches (level_pre + normalise, DSCV, SNCV) and
// their intermediates (para_prev_l / depth_prev_l / other_prev_l / para_prev_t maps written and re-read, the 4- and
// 36-byte pieces the separate kernels scatter into the 256-byte f_input rows: 2.8x write amplification measured).
//
// A workgroup owns a TW x TH pixel tile (TW*TH*K = 256 (pixel, cut) items):
//   A  stage the (TH+6) x (TW+6) halo of the RAW features in LDS (zero outside the image), normalise every (pixel, cut)
//      run in place -- the halo pixels are normalised redundantly by the neighbouring tiles instead of being exchanged;
//      meanwhile the first TW*TH lanes upsample the coarser estimate (parallax = the DSCV's centre hypothesis, log, other);
//   B  store the tile's normalised features to the state buffer; SNCV: lane = (pixel, cut, row group), its own vector
//      and the 7x7 window out of LDS, 49 / YS results in registers;
//   C  (halo dead, its LDS bytes become the output stage) DSCV in the wave layout of m4d_dscv.hip: C/4 lanes per pixel,
//      corners of the previous frame's features gathered through L2, float16 products summed in channel order by a
//      shuffle chain; the centre hypothesis also samples prev_d2para of the depth memory at its 4 corners;
//   D  the stage goes out as whole rows.
// Arithmetic, operand order and summation order are those of the separate kernels and of the oracle: bit-exact.
#include <cstdlib>
#include "m4d_common.h"
#include "../../include/m4depth_hip.h"

namespace {

typedef float front_f2 __attribute__((ext_vector_type(2)));
typedef _Float16 front_h2 __attribute__((ext_vector_type(2)));

// Two channels at a time on the packed float32 / float16 pipes (v_pk_add_f32 / v_pk_mul_f32, v_cvt_pk_f16_f32,
// v_pk_mul_f16): a*(r-l)+l twice, then across rows (utils/dense_image_warp.py:188-190) -- the same three roundings per
// step as m4d_lerp2, lane by lane.
__device__ __forceinline__ front_f2 front_lerp2(front_f2 tl, front_f2 tr, front_f2 bl, front_f2 br, float ax, float ay) {
  const front_f2 ax2 = {ax, ax}, ay2 = {ay, ay};
  const front_f2 top = ax2 * (tr - tl) + tl;
  const front_f2 bot = ax2 * (br - bl) + bl;
  return ay2 * (bot - top) + top;
}

struct FrontArgs {
  const float* raw;            // current features, raw [b,h,w,C]
  float* norm_out;             // their per-cut normalisation -> the buffer that becomes prev_f_maps
  const float* c2;             // previous frame's normalised features (temporal memory)
  const float* depth_prev_t;   // depth memory [b,h,w,1]
  const float* pl_para; const float* pl_other; int ph, pw;      // coarser level's estimate (nullptr at the coarsest level)
  const float* rot; int rot_c; const float* trans; const float* cam_f; const float* cam_c;
  int b, h, w;
  float* f_input; float log_scale; int cv_accum;
  int tiles_x, tiles_y;
  unsigned long long* stamps;  // profiling only (m4d_front_set_stamps): 7 cycle-counter stamps per workgroup
};

// RD / RS = the DSCV / SNCV search ranges: 4 / 3 are the reference's (m4depth_network.py:221,232); 6 / 6 is BASELINE configs[4]
// ("large-window LDS tiling stress": 13 hypotheses, a 13x13 window -- the halo is 6 pixels wide and a refiner-input row 188 ...
// 734 floats, so the tiles are 16x8 ... 4x4 pixels)
template <int NC, int K, int TW, int TH, int YS, int RD = 4, int RS = 3>
struct FrontGeom {
  static constexpr int R = RS, MO = 2 * RS + 1, NCP = 2 * RD + 1;
  static constexpr int C = NC * K, CP = C + 4, C4 = C / 4;
  static constexpr int HWT = TW + 2 * R, HHT = TH + 2 * R;
  static constexpr int P = TW * TH;
  static constexpr int IT = P * K;                              // (pixel, cut) items of the tile
  static constexpr int NT = IT * YS, NW = NT / 64;
  static constexpr int LP = C / 4, G = NC / 4, PPW = 64 / LP;
  static constexpr int F_IN = NCP * K + 1 + 4 + MO * MO * K + 1;
  static constexpr int F_ST = (F_IN + 7) / 8 * 8;               // row stride of the refiner input in HBM
  static constexpr int F_LDS = F_ST + 1;                        // odd row stride of the stage: conflict-free column writes
  static constexpr int LOG_OFF = NCP * K, OTHER_OFF = LOG_OFF + 1, SNCV_OFF = LOG_OFF + 5, LOGT_OFF = F_IN - 1;
  static constexpr int HALO = HHT * HWT * CP, STAGE = P * F_LDS;
  static constexpr int AREA = ((HALO > STAGE ? HALO : STAGE) + 3) / 4 * 4;
  static constexpr int LDS_FLOATS = AREA + P * 6;
  static constexpr int HALO_F4 = HHT * HWT * C4;
  static constexpr int U = (HALO_F4 + NT - 1) / NT;
  static constexpr int RPL = (MO + YS - 1) / YS;
  static constexpr int NPASS = (P + NW * PPW - 1) / (NW * PPW);
  static_assert(IT % 64 == 0 && NT <= 1024 && P <= NT, "whole waves per row group; one lane per pixel in the upsampling step");
  static_assert(LP >= 4 && LP <= 64, "the centre hypothesis spreads its 4 corners over 4 lanes of the pixel");
};

template <int NC, int K, int TW, int TH, int YS, bool SEQ16, int RD = 4, int RS = 3>
__global__ void __launch_bounds__(TW * TH * K * YS)
level_front_kernel(const FrontArgs a) {
  using Gm = FrontGeom<NC, K, TW, TH, YS, RD, RS>;
  constexpr int R = Gm::R, MO = Gm::MO, NCP = Gm::NCP, C = Gm::C, CP = Gm::CP, C4 = Gm::C4, HWT = Gm::HWT, HHT = Gm::HHT;
  constexpr int P = Gm::P, NT = Gm::NT, NW = Gm::NW, LP = Gm::LP, G = Gm::G, PPW = Gm::PPW;
  constexpr int F_IN = Gm::F_IN, F_ST = Gm::F_ST, F_LDS = Gm::F_LDS, U = Gm::U, RPL = Gm::RPL, NPASS = Gm::NPASS;
  extern __shared__ __align__(16) float smem[];
  float* tile = smem;                     // halo of the normalised features, later the output stage
  float* preS = smem + Gm::AREA;          // per pixel: parallax (x2 upsampled), log, other[4]
  const int t = threadIdx.x;
  const int tiles_img = a.tiles_x * a.tiles_y;
  int blk = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) blk = (blk & 7) * (nb >> 3) + (blk >> 3);        // one contiguous band of tiles per XCD (halo / gather reuse in its L2)
  const int bi = blk / tiles_img;
  const int tl = blk - bi * tiles_img;
  const int tile_y = (tl / a.tiles_x) * TH, tile_x = (tl % a.tiles_x) * TW;
  const int h = a.h, w = a.w, hw = h * w;
  const M4dMotion m = m4d_load_motion(a.rot, a.rot_c, a.trans, a.cam_f, a.cam_c, bi);
  unsigned long long* st = a.stamps ? a.stamps + (long long)blockIdx.x * 8 : nullptr;
  if (st && t == 0) st[0] = __builtin_readcyclecounter();

  // ---- A1: raw halo -> registers (unconditional clamped loads: a load under a branch is waited for at the merge)
  {
    const float* img = a.raw + (long long)bi * hw * C;
    float4 pre[U];
    unsigned pre_ok = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = min(u * NT + t, Gm::HALO_F4 - 1);
      const int hp = idx / C4, c4 = idx % C4;
      const int py = hp / HWT, pxx = hp % HWT;
      const int gy = tile_y - R + py, gx = tile_x - R + pxx;
      const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
      pre_ok |= ok ? (1u << u) : 0u;
      const int cy = min(max(gy, 0), h - 1), cx = min(max(gx, 0), w - 1);
      pre[u] = *reinterpret_cast<const float4*>(img + ((long long)cy * w + cx) * C + c4 * 4);
    }
    // ---- meanwhile: x2 upsample of the coarser estimate for the tile's pixels (:196-204, :224, :227)
    if (t < P) {
      const int ty = t / TW, tx = t % TW;
      const int j = min(tile_y + ty, h - 1), i = min(tile_x + tx, w - 1);
      float para = 1.0f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;                      // :198-200
      if (a.pl_para != nullptr) {
        const float sy = (float)a.ph / (float)h, sx = (float)a.pw / (float)w;
        const ResizeAxis ya = resize_axis(j, sy, a.ph), xa = resize_axis(i, sx, a.pw);
        const long long pb = (long long)bi * a.ph * a.pw;
        para = resize_sample(a.pl_para + pb, a.pw, 1, 0, ya, xa) * 2.0f;              // :203
        const float* ob = a.pl_other + pb * 4;
        o0 = resize_sample(ob, a.pw, 4, 0, ya, xa); o1 = resize_sample(ob, a.pw, 4, 1, ya, xa);
        o2 = resize_sample(ob, a.pw, 4, 2, ya, xa); o3 = resize_sample(ob, a.pw, 4, 3, ya, xa);
      }
      float* ps = preS + t * 6;
      ps[0] = para; ps[1] = logf(para * a.log_scale);                                  // :224
      ps[2] = o0; ps[3] = o1; ps[4] = o2; ps[5] = o3;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = u * NT + t;
      const int hp = idx / C4, c4 = idx % C4;
      const bool ok = (pre_ok >> u) & 1u;
      const float4 v = pre[u];
      if (idx < Gm::HALO_F4)                                                          // zero padding (depth_operations.py:293)
        *reinterpret_cast<float4*>(tile + hp * CP + c4 * 4) = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
  }
  __syncthreads();
  if (st && t == 0) st[1] = __builtin_readcyclecounter();
  // ---- A2: normalise every (pixel, cut) run of the halo in place (tf.linalg.normalize: x / sqrt(sum x^2), sequential sum)
  for (int it = t; it < HHT * HWT * K; it += NT) {
    const int hp = it / K, kk = it % K;
    const int py = hp / HWT, pxx = hp % HWT;
    const int gy = tile_y - R + py, gx = tile_x - R + pxx;
    if (gy < 0 || gy >= h || gx < 0 || gx >= w) continue;                             // padding stays zero
    float* p = tile + hp * CP + kk * NC;
    float4 v[NC / 4];
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < NC / 4; ++c) {
      v[c] = *reinterpret_cast<const float4*>(p + 4 * c);
      if (c == 0) acc = v[c].x * v[c].x; else acc = acc + v[c].x * v[c].x;
      acc = acc + v[c].y * v[c].y; acc = acc + v[c].z * v[c].z; acc = acc + v[c].w * v[c].w;
    }
    const float nrm = sqrtf(acc);
#pragma unroll
    for (int c = 0; c < NC / 4; ++c)
      *reinterpret_cast<float4*>(p + 4 * c) = make_float4(v[c].x / nrm, v[c].y / nrm, v[c].z / nrm, v[c].w / nrm);
  }
  __syncthreads();
  if (st && t == 0) st[2] = __builtin_readcyclecounter();
  // ---- B1: the tile's normalised features -> temporal memory (coalesced 16-byte stores)
  {
    float* nout = a.norm_out + (long long)bi * hw * C;
    for (int e = t; e < P * C4; e += NT) {
      const int pxl = e / C4, c4 = e % C4;
      const int ty = pxl / TW, tx = pxl % TW;
      const int gy = tile_y + ty, gx = tile_x + tx;
      if (gy < h && gx < w)
        *reinterpret_cast<float4*>(nout + ((long long)gy * w + gx) * C + c4 * 4) =
            *reinterpret_cast<const float4*>(tile + ((ty + R) * HWT + tx + R) * CP + c4 * 4);
    }
  }
  // ---- B2: SNCV (cost_volume with c1 == c2): lane = (pixel, cut, row group)
  const int item = t % Gm::IT;
  const int ys = YS > 1 ? __builtin_amdgcn_readfirstlane(t / Gm::IT) : 0;
  const int s_kk = item % K, s_lp = item / K;
  float res[RPL * MO];
  {
    const int ty = s_lp / TW, tx = s_lp % TW;
    const float* base = tile + (ty * HWT + tx) * CP + s_kk * NC;
    front_f2 c1p[NC / 2];
#pragma unroll
    for (int cc = 0; cc < NC; cc += 4) {
      const float4 v = *reinterpret_cast<const float4*>(base + (R * HWT + R) * CP + cc);
      c1p[cc / 2] = front_f2{v.x, v.y};
      c1p[cc / 2 + 1] = front_f2{v.z, v.w};
    }
    const int y_lo = ys * RPL;
#pragma unroll
    for (int yy = 0; yy < RPL; ++yy) {
      const int y = y_lo + yy;
      if (YS > 1 && MO % YS != 0 && y >= MO) break;          // uneven split: the last group has fewer rows (wave-uniform)
#pragma unroll
      for (int x = 0; x < MO; ++x) {
        float acc = 0.f;
#pragma unroll
        for (int cc = 0; cc < NC; cc += 4) {
          const float4 v = *reinterpret_cast<const float4*>(base + (y * HWT + x) * CP + cc);
          const front_f2 pa = c1p[cc / 2] * front_f2{v.x, v.y};
          const front_f2 pb = c1p[cc / 2 + 1] * front_f2{v.z, v.w};
          if (cc == 0) acc = pa.x; else acc = acc + pa.x;
          acc = acc + pa.y; acc = acc + pb.x; acc = acc + pb.y;
        }
        const float mean = acc / (float)NC;                                  // depth_operations.py:308
        res[yy * MO + x] = fmaxf(mean, mean * 0.1f);                         // leaky_relu(0.1) == max(x, 0.1 x) (:311)
      }
    }
  }
  if (st && t == 0) st[3] = __builtin_readcyclecounter();
  // ---- C0: the DSCV lanes take their 4 channels of c1 (pre-rounded to half, :276) out of the halo before it dies
  const int lane = t & 63, wave = t >> 6;
  const int slot = lane / LP;                                  // pixel slot in the wave
  const int q = lane - slot * LP;                              // float4 index inside the pixel's feature vector
  const int g = q % G, kk = q / G, base_lane = slot * LP;
  front_h2 c1h[NPASS][2];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    int p = (ps * NW + wave) * PPW + slot;
    if (slot >= PPW || p >= P) p = P - 1;
    const int ty = p / TW, tx = p % TW;
    const float4 v = *reinterpret_cast<const float4*>(tile + ((ty + R) * HWT + tx + R) * CP + 4 * q);
    c1h[ps][0] = __builtin_convertvector((front_f2){v.x, v.y}, front_h2);   // round-to-nearest-even casts (:276)
    c1h[ps][1] = __builtin_convertvector((front_f2){v.z, v.w}, front_h2);
  }
  __syncthreads();                                             // the halo is dead: its bytes are the output stage from here on
  // ---- C1: SNCV results, log / memory features and the padding channels into the stage
  {
    const int y_lo = ys * RPL;
#pragma unroll
    for (int yy = 0; yy < RPL; ++yy) {
      if (y_lo + yy < MO) {
#pragma unroll
        for (int x = 0; x < MO; ++x)
          tile[s_lp * F_LDS + Gm::SNCV_OFF + ((y_lo + yy) * MO + x) * K + s_kk] = res[yy * MO + x];
      }
    }
    if (t < P) {
      const float* ps = preS + t * 6;
      float* row = tile + t * F_LDS;
      row[Gm::LOG_OFF] = ps[1];
      row[Gm::OTHER_OFF] = ps[2]; row[Gm::OTHER_OFF + 1] = ps[3]; row[Gm::OTHER_OFF + 2] = ps[4]; row[Gm::OTHER_OFF + 3] = ps[5];
#pragma unroll
      for (int c = F_IN; c < F_ST; ++c) row[c] = 0.f;                        // channel padding of the refiner input stays zero
    }
  }
  if (st && t == 0) st[4] = __builtin_readcyclecounter();
  // ---- C2: DSCV (get_parallax_sweeping_cv), wave layout: LP lanes per pixel, PPW pixels per wave
  {
    const float* c2img = a.c2 + (long long)bi * hw * C + 4 * q;
    const float* dpt = a.depth_prev_t + (long long)bi * hw;
    const long long rs = (long long)w * C;
    constexpr bool seq16 = SEQ16;                              // cv_accum "fp16_seq": sequential float16 adds (oracle [UNPINNED] note)
    constexpr int J = (NCP + LP - 1) / LP;                     // hypotheses computed by each lane
    constexpr int HB = 3;                                      // hypotheses in flight per batch
    constexpr int r = (NCP - 1) / 2;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      int p = (ps * NW + wave) * PPW + slot;
      bool active = slot < PPW && p < P;
      if (!active) p = P - 1;
      const int ty = p / TW, tx = p % TW;
      int j = tile_y + ty, i = tile_x + tx;
      if (j >= h || i >= w) { active = false; j = min(j, h - 1); i = min(i, w - 1); }
      const M4dPixel px = m4d_pixel_factors(m, i, j);
      const float start_x = px.x * m.fx;                       // depth_operations.py:256
      const float start_y = px.y * m.fy;
      const float disp = preS[p * 6];
      // hypothesis tt = jj*LP + q of the pixel is prepared by lane q: query point, then the bilinear cell (floor clamped to
      // [0, size-2], packed as y0 << 16 | x0) and the two blend weights -- broadcast to the pixel's lanes below
      int oyx[J];
      float oay[J], oax[J];
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        const int tt = jj * LP + q;
        const float n = (float)((tt < NCP ? tt : 0) - r);
        const float pp = fminf(fmaxf(disp + n, 1e-6f), 1e6f);  // :235-236
        const float divider = px.s / pp;                       // :262
        const float dxx = px.delta_x / divider;                // :263
        const float dyy = px.delta_y / divider;
        const float flow_x = (px.proj_x + dxx) - start_x;      // :264
        const float flow_y = (px.proj_y + dyy) - start_y;
        const float qy = (float)j + flow_y;                    // dense_image_warp.py:244
        const float qx = (float)i + flow_x;
        int y0q, x0q;
        m4d_bilinear_axis(qy, h, y0q, oay[jj]);
        m4d_bilinear_axis(qx, w, x0q, oax[jj]);
        oyx[jj] = (y0q << 16) | x0q;
      }
      const front_h2 c1lo = c1h[ps][0], c1hi = c1h[ps][1];
      float* o = tile + p * F_LDS + kk * NCP;
#pragma unroll
      for (int tb = 0; tb < NCP; tb += HB) {
        float4 vtl[HB], vtr[HB], vbl[HB], vbr[HB];
        float ay[HB], ax[HB];
        int y0[HB], x0[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) {
          if (tb + u < NCP) {
            const int tt = tb + u;
            const int yx = __shfl(oyx[tt / LP], base_lane + (tt % LP));
            ay[u] = __shfl(oay[tt / LP], base_lane + (tt % LP));
            ax[u] = __shfl(oax[tt / LP], base_lane + (tt % LP));
            y0[u] = yx >> 16; x0[u] = yx & 0xffff;
            const float* cp = c2img + ((long long)y0[u] * w + x0[u]) * C;
            vtl[u] = *reinterpret_cast<const float4*>(cp);
            vtr[u] = *reinterpret_cast<const float4*>(cp + C);
            vbl[u] = *reinterpret_cast<const float4*>(cp + rs);
            vbr[u] = *reinterpret_cast<const float4*>(cp + rs + C);
          }
        }
        float part[HB][4], acc[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) {
          if (tb + u < NCP) {
            const int tt = tb + u;
            const front_f2 wlo = front_lerp2((front_f2){vtl[u].x, vtl[u].y}, (front_f2){vtr[u].x, vtr[u].y},
                                             (front_f2){vbl[u].x, vbl[u].y}, (front_f2){vbr[u].x, vbr[u].y}, ax[u], ay[u]);
            const front_f2 whi = front_lerp2((front_f2){vtl[u].z, vtl[u].w}, (front_f2){vtr[u].z, vtr[u].w},
                                             (front_f2){vbl[u].z, vbl[u].w}, (front_f2){vbr[u].z, vbr[u].w}, ax[u], ay[u]);
            // half(c1) * half(c2_warped) in float16 (:276): the float16 product of two float16 values is their exact product
            // rounded once, i.e. round_half(c1h * c2h)
            const front_f2 plo = __builtin_convertvector(c1lo * __builtin_convertvector(wlo, front_h2), front_f2);
            const front_f2 phi = __builtin_convertvector(c1hi * __builtin_convertvector(whi, front_h2), front_f2);
            part[u][0] = plo.x; part[u][1] = plo.y; part[u][2] = phi.x; part[u][3] = phi.y;
            acc[u] = !seq16 ? ((part[u][0] + part[u][1]) + part[u][2]) + part[u][3]
                            : m4d_round_half(m4d_round_half(m4d_round_half(part[u][0] + part[u][1]) + part[u][2]) + part[u][3]);
            if (tt == r) {
              // the extra channel of :268 for the centre hypothesis: para_prev_t = prev_d2para(depth memory) (:218) at the
              // 4 corners, one corner per lane q = 0..3 of the pixel, then the same bilinear blend (:238)
              const int cq = q & 3;
              const int ci = x0[u] + (cq & 1), cj = y0[u] + (cq >> 1);
              const float pc = m4d_prev_d2para_px(m, dpt[(long long)cj * w + ci], ci, cj);
              const float p01 = __shfl(pc, base_lane + 1), p10 = __shfl(pc, base_lane + 2), p11 = __shfl(pc, base_lane + 3);
              if (active && q == 0) {
                const float wd = m4d_lerp2(pc, p01, p10, p11, ax[u], ay[u]);
                tile[p * F_LDS + Gm::LOGT_OFF] = logf(wd * a.log_scale);     // m4depth_network.py:238
              }
            }
          }
        }
        // sequential (channel-order) sums across the G lanes of the cut
#pragma unroll
        for (int s = 1; s < G; ++s) {
          float prev[HB];
#pragma unroll
          for (int u = 0; u < HB; ++u) if (tb + u < NCP) prev[u] = __shfl_up(acc[u], 1);
          if (g == s) {
#pragma unroll
            for (int u = 0; u < HB; ++u)
              if (tb + u < NCP)
                acc[u] = !seq16 ? (((prev[u] + part[u][0]) + part[u][1]) + part[u][2]) + part[u][3]
                                : m4d_round_half(m4d_round_half(m4d_round_half(m4d_round_half(prev[u] + part[u][0]) + part[u][1]) + part[u][2]) + part[u][3]);
          }
        }
        if (active && q < LP && g == G - 1) {
#pragma unroll
          for (int u = 0; u < HB; ++u)
            if (tb + u < NCP) o[tb + u] = m4d_round_half(acc[u] / (float)NC);  // :277-278
        }
        asm volatile("" ::: "memory");            // keep the next batch's loads below this point
      }
    }
  }
  __syncthreads();
  if (st && t == 0) st[5] = __builtin_readcyclecounter();
  // ---- D: the tile's refiner-input rows, contiguous in HBM per tile row
  // 16 bytes per lane (round 4): the stage's rows have an odd stride (conflict-free column writes above), so a lane gathers its
  // four floats with four ds_read_b32 -- 64 lanes of one read still hit 64 distinct banks (bank = 4 quad + pixel + i) -- and
  // stores them as ONE global_store_dwordx4: a quarter of the store instructions, and a CU issues 16-byte pieces of a
  // 256-byte run several times faster than 4-byte pieces (tools/micro/store_issue_probe.hip; this phase was 12 % of the kernel)
  {
    float* fin = a.f_input + (long long)bi * hw * F_ST;
    constexpr int Q = F_ST / 4;                                // 16-byte quads per refiner-input row (F_ST is a multiple of 8)
    for (int e = t; e < P * Q; e += NT) {
      const int pxl = e / Q, q4 = e % Q;
      const int oy = tile_y + pxl / TW, ox = tile_x + pxl % TW;
      const float* src = tile + pxl * F_LDS + 4 * q4;
      const float v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
      if (oy < h && ox < w) m4d_store16(fin + ((long long)oy * w + ox) * F_ST + 4 * q4, v0, v1, v2, v3);
    }
  }
  if (st && t == 0) st[6] = __builtin_readcyclecounter();
}

template <int NC, int K, int TW, int TH, int YS, bool SEQ16, int RD = 4, int RS = 3>
int launch_front_acc(const FrontArgs& a0, hipStream_t s) {
  using Gm = FrontGeom<NC, K, TW, TH, YS, RD, RS>;
  constexpr size_t lds = (size_t)Gm::LDS_FLOATS * sizeof(float);
  static_assert(lds <= 160 * 1024, "tile does not fit LDS");
  FrontArgs a = a0;
  a.tiles_x = (a.w + TW - 1) / TW;
  a.tiles_y = (a.h + TH - 1) / TH;
  const long long total = (long long)a.tiles_x * a.tiles_y * a.b;
  if (total > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  M4D_LDS_OPT_IN(&level_front_kernel<NC, K, TW, TH, YS, SEQ16, RD, RS>);
  m4d_launch((level_front_kernel<NC, K, TW, TH, YS, SEQ16, RD, RS>), dim3((unsigned)total), dim3(Gm::NT), lds, s, a);
  return M4D_LAUNCH_RESULT();
}

template <int NC, int K, int TW, int TH, int YS, int RD = 4, int RS = 3>
int launch_front(const FrontArgs& a, hipStream_t s) {
  return a.cv_accum ? launch_front_acc<NC, K, TW, TH, YS, true, RD, RS>(a, s) : launch_front_acc<NC, K, TW, TH, YS, false, RD, RS>(a, s);
}

unsigned long long* g_front_stamps = nullptr;        // debug hook (m4d_front_set_stamps)

}  // namespace

extern "C" void m4d_front_set_stamps(unsigned long long* device_buffer) { g_front_stamps = device_buffer; }

extern "C" int m4d_level_front_supported(int C, int nbre_cuts, int dscv_range, int sncv_range, int f_stride) {
  if (nbre_cuts <= 0 || C % nbre_cuts != 0) return 0;
  const int nc = C / nbre_cuts;
  const int ncp = 2 * dscv_range + 1, mo = 2 * sncv_range + 1;
  const int f_in = (ncp + mo * mo) * nbre_cuts + 6;
  if (f_stride != (f_in + 7) / 8 * 8) return 0;
  if (dscv_range == 4 && sncv_range == 3)
    return (nc == 16 && nbre_cuts == 1) || (nc == 16 && nbre_cuts == 2) || (nc == 32 && nbre_cuts == 2) ||
           (nc == 24 && nbre_cuts == 4) || (nc == 32 && nbre_cuts == 4) || (nc == 24 && nbre_cuts == 8);     // levels 4, 5, 6
  if (dscv_range == 6 && sncv_range == 6)          // BASELINE configs[4]: levels 1-5 (level 6's 13x13 halo of 192 channels is 200 KB)
    return (nc == 16 && nbre_cuts == 1) || (nc == 16 && nbre_cuts == 2) || (nc == 32 && nbre_cuts == 2) ||
           (nc == 24 && nbre_cuts == 4) || (nc == 32 && nbre_cuts == 4);
  return 0;
}

extern "C" int m4d_level_front_r(const float* raw_f, float* norm_out, const float* prev_f, const float* depth_prev_t,
                                 const float* prev_l_parallax, const float* prev_l_other, int ph, int pw,
                                 const float* rot, int rot_c, const float* trans, const float* cam_f, const float* cam_c,
                                 int b, int h, int w, int C, int nbre_cuts, int dscv_range, int sncv_range, int cv_accum,
                                 float* f_input, int f_stride, float log_scale, void* stream) {
  M4D_CHECK_ARG(raw_f && norm_out && prev_f && depth_prev_t && rot && trans && cam_f && cam_c && f_input);
  M4D_CHECK_ARG(b > 0 && h >= 2 && w >= 2 && (rot_c == 3 || rot_c == 4) && (cv_accum == 0 || cv_accum == 1));
  M4D_CHECK_ARG((prev_l_parallax == nullptr) == (prev_l_other == nullptr));
  if (prev_l_parallax) M4D_CHECK_ARG(ph > 0 && pw > 0);
  M4D_CHECK_ARG(m4d_level_front_supported(C, nbre_cuts, dscv_range, sncv_range, f_stride));
  M4D_CHECK_ARG(((((uintptr_t)raw_f | (uintptr_t)norm_out | (uintptr_t)prev_f | (uintptr_t)f_input)) & 15u) == 0);
  M4D_CHECK_ARG(raw_f != norm_out && prev_f != norm_out);
  FrontArgs a;
  a.raw = raw_f; a.norm_out = norm_out; a.c2 = prev_f; a.depth_prev_t = depth_prev_t;
  a.pl_para = prev_l_parallax; a.pl_other = prev_l_other; a.ph = ph; a.pw = pw;
  a.rot = rot; a.rot_c = rot_c; a.trans = trans; a.cam_f = cam_f; a.cam_c = cam_c;
  a.b = b; a.h = h; a.w = w; a.f_input = f_input; a.log_scale = log_scale; a.cv_accum = cv_accum;
  a.tiles_x = a.tiles_y = 0;
  a.stamps = g_front_stamps;
  hipStream_t s = (hipStream_t)stream;
  const int nc = C / nbre_cuts;
  if (dscv_range == 6) {
    // 13 hypotheses, 13x13 window (round 5): the tile is what the refiner-input stage (188 / 370 / 734 floats per pixel) and the
    // 6-pixel halo leave of the LDS; the window rows are split four ways (13 = 4 + 4 + 4 + 1)
    if (nc == 16 && nbre_cuts == 1) return launch_front<16, 1, 16, 8, 4, 6, 6>(a, s);      // stage 99 KB, halo 45 KB
    if (nc == 16 && nbre_cuts == 2) return launch_front<16, 2, 8, 8, 4, 6, 6>(a, s);       // stage 97 KB, halo 58 KB
    if (nc == 32 && nbre_cuts == 2) return launch_front<32, 2, 8, 8, 4, 6, 6>(a, s);       // halo 109 KB
    if (nc == 24 && nbre_cuts == 4) return launch_front<24, 4, 8, 4, 4, 6, 6>(a, s);       // halo 128 KB
    return launch_front<32, 4, 4, 4, 4, 6, 6>(a, s);                                       // halo 135 KB
  }
  // Tile / workgroup shape by geometry and size (profiles/r02_front_tile_sweep.txt): one workgroup's pipeline is a serial
  // chain (stage -> normalise -> SNCV -> DSCV -> store), so a launch needs several workgroups per CU in flight; small maps
  // (levels 2-3 at batch 1: 30720 / 7680 pixels) take half-height tiles with the SNCV window rows split four ways.
  // M4D_FRONT_L1 / _L2 / _L3 force a shape (profiling).
  static int v1 = -1, v2 = -1, v3 = -1;
  if (v1 < 0) {
    const char* e = getenv("M4D_FRONT_L1"); v1 = e ? atoi(e) : 0;
    e = getenv("M4D_FRONT_L2"); v2 = e ? atoi(e) : 0;
    e = getenv("M4D_FRONT_L3"); v3 = e ? atoi(e) : 0;
  }
  const bool small_map = (long long)b * h * w < 100000;
  if (nc == 16 && nbre_cuts == 1) {
    switch (v1) {
      case 1: return launch_front<16, 1, 16, 8, 2>(a, s);
      case 2: return launch_front<16, 1, 16, 8, 4>(a, s);
      case 3: return launch_front<16, 1, 32, 8, 4>(a, s);
      default: return launch_front<16, 1, 32, 8, 2>(a, s);
    }
  }
  if (nc == 16 && nbre_cuts == 2) {
    switch (v2 ? v2 : (small_map ? 4 : 5)) {
      case 1: return launch_front<16, 2, 8, 8, 2>(a, s);
      case 2: return launch_front<16, 2, 8, 8, 4>(a, s);
      case 3: return launch_front<16, 2, 16, 4, 2>(a, s);
      case 4: return launch_front<16, 2, 16, 4, 4>(a, s);
      default: return launch_front<16, 2, 16, 8, 2>(a, s);
    }
  }
  // levels 4-6 of the 6-level pyramid (96 / 128 / 192 channels in 4 / 4 / 8 cuts: 24 / 32 / 48 lanes per pixel in the DSCV
  // phase, 6 or 8 lanes per cut): 8x8 tiles (8x4 for the 192-channel halo, 110 KB), half-size tiles with the window rows
  // split four ways on small maps
  if (nc == 24 && nbre_cuts == 4) return small_map ? launch_front<24, 4, 8, 4, 4>(a, s) : launch_front<24, 4, 8, 8, 2>(a, s);
  if (nc == 32 && nbre_cuts == 4) return small_map ? launch_front<32, 4, 8, 4, 4>(a, s) : launch_front<32, 4, 8, 8, 2>(a, s);
  if (nc == 24 && nbre_cuts == 8) return small_map ? launch_front<24, 8, 4, 4, 4>(a, s) : launch_front<24, 8, 8, 4, 2>(a, s);
  switch (v3 ? v3 : (small_map ? 5 : 6)) {
    case 1: return launch_front<32, 2, 8, 8, 2>(a, s);
    case 2: return launch_front<32, 2, 8, 4, 4>(a, s);
    case 3: return launch_front<32, 2, 16, 4, 2>(a, s);
    case 4: return launch_front<32, 2, 8, 8, 4>(a, s);
    case 5: return launch_front<32, 2, 16, 4, 4>(a, s);
    default: return launch_front<32, 2, 16, 8, 2>(a, s);
  }
}

extern "C" int m4d_level_front(const float* raw_f, float* norm_out, const float* prev_f, const float* depth_prev_t,
                               const float* prev_l_parallax, const float* prev_l_other, int ph, int pw,
                               const float* rot, int rot_c, const float* trans, const float* cam_f, const float* cam_c,
                               int b, int h, int w, int C, int nbre_cuts, int cv_accum,
                               float* f_input, int f_stride, float log_scale, void* stream) {
  return m4d_level_front_r(raw_f, norm_out, prev_f, depth_prev_t, prev_l_parallax, prev_l_other, ph, pw, rot, rot_c, trans, cam_f,
                           cam_c, b, h, w, C, nbre_cuts, 4, 3, cv_accum, f_input, f_stride, log_scale, stream);
}
