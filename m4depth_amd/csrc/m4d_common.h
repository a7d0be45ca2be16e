// Shared device helpers for libm4depth_hip.so (gfx950 / CDNA4 only).
//
// Numerics contract: built with -ffp-contract=off; every expression below has
// one IEEE float32 rounding per operation in the operand order of
// oracle/m4depth_oracle.py (which cites the reference lines), so query points
// and therefore the int32 bilinear index grid are bit-identical to the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <atomic>

#define M4D_CHECK_ARG(cond) do { if (!(cond)) return (int)hipErrorInvalidValue; } while (0)
#define M4D_LAUNCH_RESULT() ((int)hipGetLastError())

static inline int m4d_blocks(long long n, int threads) { return (int)((n + threads - 1) / threads); }

// ---- m4d_launch(): every kernel of the library is launched through it ----------------------------------------------------
// In the product library it IS hipLaunchKernelGGL.  `make EXPERIMENTS=1` adds the launch tape (m4d_tape.hip,
// include/m4depth_hip_experiments.h): while the calling thread records (m4d_tape_begin .. m4d_tape_end) the launch is not
// executed but appended -- function, grid, block, dynamic LDS and a private copy of every argument -- to the tape, which
// m4d_tape_replay() later issues as plain stream launches from one host loop (DESIGN_HISTORY.md: not faster than hipGraph).
#ifndef M4D_EXPERIMENTS
#define M4D_EXPERIMENTS 0
#endif
#if M4D_EXPERIMENTS
#include <tuple>
#include <utility>
bool m4d_tape_recording();
void m4d_tape_push(const void* fn, dim3 grid, dim3 block, unsigned lds, void* const* params, const size_t* sizes, int n);

template <class Tuple, size_t... I>
inline void m4d_tape_push_tuple(const void* fn, dim3 grid, dim3 block, unsigned lds, Tuple& vals, std::index_sequence<I...>) {
  void* params[sizeof...(I) + 1] = {static_cast<void*>(&std::get<I>(vals))...};
  const size_t sizes[sizeof...(I) + 1] = {sizeof(std::tuple_element_t<I, Tuple>)...};
  m4d_tape_push(fn, grid, block, lds, params, sizes, (int)sizeof...(I));
}
#endif

// kernel launches issued by this library since it was loaded (m4d_launch_count(): how many launches one step is made of --
// the difference across a hipGraph capture of the step; a host-side counter, relaxed atomic, no effect on the launches)
extern "C" void m4d_count_launch(void);

template <class... KArgs, class... Args>
inline void m4d_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args&&... args) {
  m4d_count_launch();
#if M4D_EXPERIMENTS
  if (m4d_tape_recording()) {
    std::tuple<std::remove_cv_t<std::remove_reference_t<KArgs>>...> vals(static_cast<KArgs>(args)...);
    m4d_tape_push_tuple(reinterpret_cast<const void*>(kernel), grid, block, (unsigned)lds, vals, std::index_sequence_for<KArgs...>{});
    return;
  }
#endif
  hipLaunchKernelGGL(kernel, grid, block, lds, stream, static_cast<KArgs>(args)...);
}

// ---- the > 64 KB dynamic-LDS opt-in of a kernel: per FUNCTION and per DEVICE, from any host thread --------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the current device only, and a process may drive several GPUs from
// several threads (round 5 kept `static bool attr_set` flags: one device, unsynchronised -- ADVICE r5).  `done` = one bit per
// device ordinal; a lost race sets the attribute twice, which is harmless.  A failure is not swallowed: the launch that follows
// fails with the same condition and the entry point returns it (M4D_LAUNCH_RESULT).
inline void m4d_lds_opt_in_once(std::atomic<unsigned long long>& done, const void* fn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
  if ((done.load(std::memory_order_acquire) >> dev) & 1ull) return;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess)
    done.fetch_or(1ull << dev, std::memory_order_release);
}
#define M4D_LDS_OPT_IN_BYTES(bytes, ...)                                                                     \
  do {                                                                                                       \
    static std::atomic<unsigned long long> m4d_lds_done_{0};                                                 \
    m4d_lds_opt_in_once(m4d_lds_done_, reinterpret_cast<const void*>(__VA_ARGS__), (int)(bytes));            \
  } while (0)
#define M4D_LDS_OPT_IN(...) M4D_LDS_OPT_IN_BYTES(160 * 1024, __VA_ARGS__)

// compute units of the CURRENT device (cached per device ordinal; atomics: any host thread)
inline int m4d_device_cus() {
  static std::atomic<int> cus_of[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int c = cus_of[dev].load(std::memory_order_relaxed);
  if (c == 0) {
    c = 256;
    (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
    if (c <= 0) c = 256;
    cus_of[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}

// a compile-time integer as a value: picks the instantiation of a generic lambda (`body(m4d_int<2>{})`)
template <int N> struct m4d_int { static constexpr int value = N; };

// ---- one 16-byte global store that stays ONE instruction ------------------------------------------------------------------
// hipcc if-converts `if (vector_ok) *(float4*)p = v; else for (e) if (c + e < C) p[e] = v[e];` into guarded ELEMENT stores
// for both arms (dword / dwordx2 / dwordx3 pieces): a CU issues 128-byte runs of 4-byte pieces at 16 B/clk against ~100 B/clk
// for 16-byte pieces (tools/micro/store_issue_probe.hip, profiles/r04_wino6_persistent.txt).  As inline asm the store can
// neither be split nor merged with the element arm; the s_nop covers the store-data hazard hipcc handles for its own stores
// (the data VGPRs of a > 8-byte store must not be overwritten in the next wait states).
__device__ __forceinline__ void m4d_store16(float* p, float v0, float v1, float v2, float v3) {
  typedef float m4d_f32x4 __attribute__((ext_vector_type(4)));
  const m4d_f32x4 d = {v0, v1, v2, v3};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
}

// ---- exact 3-way bf16 split of a pair of float32 values (m4d_wino6*.hip) ---------------------------------------------------
// M4D_SPLIT_RN = 1 (this header's default: the latency-first / tail kernels): round to nearest -- hi = bf16(v), mid = bf16(v - hi),
// lo = (v - hi) - mid (exact, 8 significant bits left: its upper half IS the conversion); 13 VALU instructions per pair, two of
// them v_cvt_pk_bf16_f32 (8 issue cycles each beside MFMAs, tools/micro/valu_cost.hip); dropped term products <= 2^-26 |a b|.
// M4D_SPLIT_RN = 0 (the Winograd kernels m4d_wino6*.hip since round 6: csrc/Makefile W6FLAGS): the split by TRUNCATION -- hi = the
// upper 16 bits of v, r = v - hi exact, mid = the upper 16 bits of r, lo = r - mid exact: 11 plain VALU instructions per pair
// (-23 % of the operand-generation issue slots); mid / lo are up to 2x / 4x larger, dropped products <= 2^-25 + 2^-26 of |a b|
// (the weights stay split round-to-nearest on the host, so the dropped terms stay unbiased) -- still below float32's own product
// rounding (2^-24).  History: round 3 measured the level-1 layers 1-2 % faster alone and nothing end to end (1354 vs 1355
// frames/s) and kept round-to-nearest; round 6, on the kernel as it is now (operand generation = 14.6 % of the launch by
// ablation, profiles/r06_wino6_ablations.txt): **+1.0 % at batch 1 (1663 -> 1678 frames/s, three interleaved pairs), +1.5 % at
// batch 32 (2175 -> 2207)**, all 283 GPU tests green, error to float64 of the kernel still BELOW the fp32-MFMA kernel's (0.097-0.100
// against 0.118-0.124 of a float32 ulp-scale, tools/bench_wino6.py), the end-to-end float64 comparison unchanged to three digits
// (GPU median 7.87e-7, 99.997 % within 1e-4; profiles/r06_split_truncation_ab.txt).  Packed words: element 0 in
// the low half.  A non-finite value has no split (Inf - Inf): the products it enters come out NaN, where float32 arithmetic
// would have kept an Inf -- the host-side weight splitter refuses non-finite weights (network_ops.split_bf16x3).
#ifndef M4D_SPLIT_RN
#define M4D_SPLIT_RN 1
#endif
__device__ __forceinline__ void m4d_split3_pair(float v0, float v1, unsigned& hi, unsigned& mid, unsigned& lo) {
#if M4D_SPLIT_RN
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  bf16x2_t c0; c0[0] = (__bf16)v0; c0[1] = (__bf16)v1;
  const unsigned q0 = __builtin_bit_cast(unsigned, c0);
  const float r0 = v0 - __builtin_bit_cast(float, q0 << 16), r1 = v1 - __builtin_bit_cast(float, q0 & 0xffff0000u);
  bf16x2_t c1; c1[0] = (__bf16)r0; c1[1] = (__bf16)r1;
  const unsigned q1 = __builtin_bit_cast(unsigned, c1);
  const float s0 = r0 - __builtin_bit_cast(float, q1 << 16), s1 = r1 - __builtin_bit_cast(float, q1 & 0xffff0000u);
  hi = q0; mid = q1;
#else
  const unsigned u0 = __builtin_bit_cast(unsigned, v0), u1 = __builtin_bit_cast(unsigned, v1);
  hi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  const float r0 = v0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = v1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
  const unsigned w0 = __builtin_bit_cast(unsigned, r0), w1 = __builtin_bit_cast(unsigned, r1);
  mid = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
  const float s0 = r0 - __builtin_bit_cast(float, w0 & 0xffff0000u), s1 = r1 - __builtin_bit_cast(float, w1 & 0xffff0000u);
#endif
  lo = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
}

// Per-sample camera motion: rotation matrix (get_rot_mat, utils/depth_operations.py:18-53),
// translation scaled by the focal lengths, level-local intrinsics.
struct M4dMotion {
  float r00, r01, r02, r10, r11, r12, r20, r21, r22;
  float fx, fy, cx, cy;
  float tx, ty, tz;      // raw translation
  float stx, sty;        // t * f  (scaled_t, :157)
};

__device__ __forceinline__ M4dMotion m4d_load_motion(const float* __restrict__ rot, int rot_c,
                                                     const float* __restrict__ trans,
                                                     const float* __restrict__ cam_f,
                                                     const float* __restrict__ cam_c, int bi) {
  M4dMotion m;
  if (rot == nullptr) {
    m.r00 = 1.f; m.r01 = 0.f; m.r02 = 0.f; m.r10 = 0.f; m.r11 = 1.f; m.r12 = 0.f; m.r20 = 0.f; m.r21 = 0.f; m.r22 = 1.f;
  } else if (rot_c == 3) {
    const float r0 = rot[bi * 3 + 0], r1 = rot[bi * 3 + 1], r2 = rot[bi * 3 + 2];
    m.r00 = 1.f; m.r01 = -r2; m.r02 = r1;
    m.r10 = r2;  m.r11 = 1.f; m.r12 = -r0;
    m.r20 = -r1; m.r21 = r0;  m.r22 = 1.f;
  } else {
    const float w = rot[bi * 4 + 0], x = rot[bi * 4 + 1], y = rot[bi * 4 + 2], z = rot[bi * 4 + 3];
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    m.r00 = 1.0f - (tyy + tzz); m.r01 = txy - twz;          m.r02 = txz + twy;
    m.r10 = txy + twz;          m.r11 = 1.0f - (txx + tzz); m.r12 = tyz - twx;
    m.r20 = txz - twy;          m.r21 = tyz + twx;          m.r22 = 1.0f - (txx + tyy);
  }
  m.fx = cam_f[bi * 2 + 0]; m.fy = cam_f[bi * 2 + 1];
  m.cx = cam_c[bi * 2 + 0]; m.cy = cam_c[bi * 2 + 1];
  m.tx = trans[bi * 3 + 0]; m.ty = trans[bi * 3 + 1]; m.tz = trans[bi * 3 + 2];
  m.stx = m.tx * m.fx; m.sty = m.ty * m.fy;
  return m;
}

// The per-pixel factors shared by parallax2depth / depth2parallax / DSCV
// (utils/depth_operations.py:146-162, 174-190, 239-261).
struct M4dPixel {
  float x, y;            // coords2d = mesh / f
  float alpha, proj_x, proj_y, delta_x, delta_y, s;
};

__device__ __forceinline__ M4dPixel m4d_pixel_factors(const M4dMotion& m, int i, int j) {
  M4dPixel p;
  const float mx = ((float)i + 0.5f) - m.cx;
  const float my = ((float)j + 0.5f) - m.cy;
  p.x = mx / m.fx;
  p.y = my / m.fy;
  const float rcx = (m.r00 * p.x + m.r01 * p.y) + m.r02;
  const float rcy = (m.r10 * p.x + m.r11 * p.y) + m.r12;
  const float rcz = (m.r20 * p.x + m.r21 * p.y) + m.r22;
  p.alpha = rcz;
  p.proj_x = (rcx * m.fx) / p.alpha;
  p.proj_y = (rcy * m.fy) / p.alpha;
  p.delta_x = m.stx - m.tz * p.proj_x;
  p.delta_y = m.sty - m.tz * p.proj_y;
  p.s = sqrtf(p.delta_x * p.delta_x + p.delta_y * p.delta_y);
  return p;
}

// One dimension of _interpolate_bilinear (utils/dense_image_warp.py:127-154):
// floor clamped to [0,size-2], alpha clamped to [0,1].  fmaxf/fminf return the
// non-NaN operand, so a NaN query (t = 0 is 0/0 in the reference) yields index 0
// instead of an out-of-bounds read.
__device__ __forceinline__ void m4d_bilinear_axis(float q, int size, int& i0, float& a) {
  const float fl = fminf(fmaxf(0.0f, floorf(q)), (float)(size - 2));
  i0 = (int)fl;
  a = fminf(fmaxf(0.0f, q - fl), 1.0f);
}

// a*(r-l)+l twice, then across rows (utils/dense_image_warp.py:188-190).
__device__ __forceinline__ float m4d_lerp2(float tl, float tr, float bl, float br, float ax, float ay) {
  const float top = ax * (tr - tl) + tl;
  const float bot = ax * (br - bl) + bl;
  return ay * (bot - top) + top;
}

// float -> half -> float (round-to-nearest-even), the cast at depth_operations.py:276.
__device__ __forceinline__ float m4d_round_half(float v) { return __half2float(__float2half_rn(v)); }

// tf.compat.v1.image.resize_bilinear with legacy coordinates (m4depth_network.py:202-204): src = dst * (in / out),
// lower = floor, upper = min(ceil, in - 1), top + (bottom - top) * ylerp with top = tl + (tr - tl) * xlerp.
struct ResizeAxis { int lo, hi; float lerp; };
__device__ __forceinline__ ResizeAxis resize_axis(int o, float scale, int in_n) {
  const float src = (float)o * scale;
  const float fl = floorf(src);
  ResizeAxis a;
  a.lo = max((int)fl, 0);
  a.hi = min((int)ceilf(src), in_n - 1);
  a.lerp = src - fl;
  return a;
}
__device__ __forceinline__ float resize_sample(const float* __restrict__ img, int iw, int c, int cc,
                                               const ResizeAxis& ya, const ResizeAxis& xa) {
  const float tl = img[((long long)ya.lo * iw + xa.lo) * c + cc];
  const float tr = img[((long long)ya.lo * iw + xa.hi) * c + cc];
  const float bl = img[((long long)ya.hi * iw + xa.lo) * c + cc];
  const float br = img[((long long)ya.hi * iw + xa.hi) * c + cc];
  const float top = tl + (tr - tl) * xa.lerp;
  const float bot = bl + (br - bl) * xa.lerp;
  return top + (bot - top) * ya.lerp;
}


// prev_d2para (utils/depth_operations.py:197-215) of one pixel: delta = (t*f - tz*((mesh/f)*f)) / (depth - tz), |delta|.
__device__ __forceinline__ float m4d_prev_d2para_px(const M4dMotion& m, float depth, int i, int j) {
  const float mx = ((float)i + 0.5f) - m.cx, my = ((float)j + 0.5f) - m.cy;
  const float ccx = (mx / m.fx) * m.fx, ccy = (my / m.fy) * m.fy;
  const float den = depth - m.tz;
  const float dx = (m.stx - m.tz * ccx) / den, dy = (m.sty - m.tz * ccy) / den;
  return sqrtf(dx * dx + dy * dy);
}
