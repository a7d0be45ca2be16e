// Shared between m4d_level.hip and m4d_dscv.hip: the "preprocessor" glue of a level (m4depth_network.py:196-204, 218, 224-227) and
// the coalesced per-cut normalisation body, so that on the coarse levels the level's opening work can share ONE launch with its
// two cost volumes (m4d_level_front_small) and several levels' normalisations one launch (m4d_normalize_levels).
#pragma once
#include "m4d_common.h"

namespace m4d_level {

struct LevelPreArgs {
  const float* pl_depth; const float* pl_para; const float* pl_other; int ph, pw;
  const float* depth_prev_t; const float* trans; const float* cam_f; const float* cam_c;
  int h, w;
  float* para_prev_l; float* depth_prev_l; float* other_prev_l; float* para_prev_t;
  float* f_input; int f_stride, log_off, other_off; float log_scale;
  float* depth_state_reset;      // reset branch (:209): the level's depth memory := 1000 in the same pass
};

__device__ __forceinline__ void level_pre_body(const LevelPreArgs& a, int bx, int bi, int gdx) {
  const int hw = a.h * a.w;
  const bool has_prev = a.pl_para != nullptr;      // (the coarser level's maps come together; its depth map may be left out when
                                                   //  depth_prev_l is not wanted: m4d_level_front_small)
  const float sy = has_prev ? (float)a.ph / (float)a.h : 0.f;
  const float sx = has_prev ? (float)a.pw / (float)a.w : 0.f;
  float fx = 0.f, fy = 0.f, cx = 0.f, cy = 0.f, tx = 0.f, ty = 0.f, tz = 0.f;
  if (a.depth_prev_t) {
    fx = a.cam_f[bi * 2]; fy = a.cam_f[bi * 2 + 1];
    cx = a.cam_c[bi * 2]; cy = a.cam_c[bi * 2 + 1];
    tx = a.trans[bi * 3]; ty = a.trans[bi * 3 + 1]; tz = a.trans[bi * 3 + 2];
  }
  const float stx = tx * fx, sty = ty * fy;
  for (int p = bx * blockDim.x + threadIdx.x; p < hw; p += gdx * blockDim.x) {
    const int i = p % a.w, j = p / a.w;
    const long long gp = (long long)bi * hw + p;
    float para = 1.0f, depth = 1000.0f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;   // :198-200
    if (has_prev) {
      const ResizeAxis ya = resize_axis(j, sy, a.ph), xa = resize_axis(i, sx, a.pw);
      const long long pb = (long long)bi * a.ph * a.pw;
      para = resize_sample(a.pl_para + pb, a.pw, 1, 0, ya, xa) * 2.0f;             // :203
      if (a.pl_depth) depth = resize_sample(a.pl_depth + pb, a.pw, 1, 0, ya, xa);  // :204
      const float* ob = a.pl_other + pb * 4;
      o0 = resize_sample(ob, a.pw, 4, 0, ya, xa); o1 = resize_sample(ob, a.pw, 4, 1, ya, xa);
      o2 = resize_sample(ob, a.pw, 4, 2, ya, xa); o3 = resize_sample(ob, a.pw, 4, 3, ya, xa);
    }
    if (a.para_prev_l) a.para_prev_l[gp] = para;
    if (a.depth_prev_l) a.depth_prev_l[gp] = depth;
    if (a.other_prev_l) *reinterpret_cast<float4*>(a.other_prev_l + gp * 4) = make_float4(o0, o1, o2, o3);
    if (a.depth_state_reset) a.depth_state_reset[gp] = 1000.0f;
    if (a.depth_prev_t && a.para_prev_t) {                                           // prev_d2para, :218
      const float mx = ((float)i + 0.5f) - cx, my = ((float)j + 0.5f) - cy;
      const float ccx = (mx / fx) * fx, ccy = (my / fy) * fy;
      const float den = a.depth_prev_t[gp] - tz;
      const float dx = (stx - tz * ccx) / den, dy = (sty - tz * ccy) / den;
      a.para_prev_t[gp] = sqrtf(dx * dx + dy * dy);
    }
    if (a.f_input) {
      float* f = a.f_input + gp * a.f_stride;
      f[a.log_off] = logf(para * a.log_scale);                                       // :224
      if (a.other_off >= 0) { f[a.other_off] = o0; f[a.other_off + 1] = o1; f[a.other_off + 2] = o2; f[a.other_off + 3] = o3; }
    }
  }
}

// Coalesced variant for nc = 4 * LPG channels per cut: LPG consecutive lanes hold one (pixel, cut) run, 16 bytes each
// (the kernel above gives every lane a whole run: 16-byte loads 64-128 bytes apart, each run read twice).  The sum of
// squares keeps the reference's channel order: lane j continues the chain of lane j - 1 (LPG dependent steps through a
// lane shift), so the result is bit-identical to the sequential kernel.
template <int LPG>
__device__ __forceinline__ void normalize_cuts_group_body(const float* __restrict__ x, long long items, float* __restrict__ out,
                                                          long long block) {
  constexpr int GPW = 64 / LPG;                 // runs per wave (LPG = 6: 10 runs, 4 idle lanes)
  const int lane = threadIdx.x & 63;
  const long long wave = (block * blockDim.x + threadIdx.x) >> 6;
  const int g = lane / LPG, j = lane - g * LPG;
  const long long it = wave * GPW + g;
  const bool active = g < GPW && it < items;
  const long long itc = it < items ? it : items - 1;                   // unconditional load (clamped), masked store
  const float4 v = *reinterpret_cast<const float4*>(x + (itc * LPG + (g < GPW ? j : 0)) * 4);
  float acc = 0.f;
#pragma unroll
  for (int s = 0; s < LPG; ++s) {
    const float prev = __shfl_up(acc, 1);                             // the running sum of the lane before (same run for j >= 1)
    if (j == s) {
      acc = (s == 0) ? v.x * v.x : prev + v.x * v.x;
      acc = acc + v.y * v.y; acc = acc + v.z * v.z; acc = acc + v.w * v.w;
    }
  }
  const float tot = __shfl(acc, g < GPW ? g * LPG + LPG - 1 : lane);   // the run's last lane holds the full sum
  const float nrm = sqrtf(tot);
  if (active)
    *reinterpret_cast<float4*>(out + (it * LPG + j) * 4) = make_float4(v.x / nrm, v.y / nrm, v.z / nrm, v.w / nrm);
}

}  // namespace m4d_level
