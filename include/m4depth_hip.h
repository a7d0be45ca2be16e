/*
 * m4depth_hip.h -- C ABI of libm4depth_hip.so: the MI355X (gfx950) native
 * implementation of M4Depth's per-frame parallax-cost-volume inference path.
 *
 * This is the drop-in boundary.  The reference's only native interface is the
 * TensorFlow custom op pair BackProject / BackProjectGrad
 * (cuda_backproject/backproject_op.cc:32-42, launchers declared at
 * cuda_backproject/backproject_op_gpu.h:17-22); m4d_backproject_fwd/_bwd
 * replace those launchers one for one.  Every other entry point replaces a
 * piece of TF graph code of utils/depth_operations.py, utils/dense_image_warp.py
 * or m4depth_network.py:167-262 that the reference runs as dozens of unfused TF
 * ops; each declaration cites the lines it replaces.
 *
 * Conventions
 *   - all tensors are dense row-major NHWC float32 in device (HBM) memory unless
 *     a stride argument says otherwise; i = column (x), j = row (y);
 *   - the caller allocates every buffer; kernels never allocate, never retain
 *     pointers, write every output element (no memset dependency) and are
 *     enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL =
 *     the default stream).  No host synchronisation; the calls are re-entrant,
 *     safe from several host threads on different streams, and hipGraph-capturable.
 *     The ONLY process-wide state is the profiling / debugging hooks named
 *     m4d_*_set_* (cycle-stamp buffers, the DSCV kernel selector, ablation masks):
 *     all off by default, not meant to be flipped while another thread launches,
 *     and never touched by the product's dispatch (m4depth_amd/network.py) -- every
 *     tuning choice of the dispatch (which Winograd kernel, its staggered first
 *     round) is an ARGUMENT of the call (ABI 6 removed round 5's
 *     m4d_wino6_set_stagger, the one setter the dispatch did touch).  The
 *     measured-and-not-dispatched alternative kernels and the launch tape are not
 *     in this library: `make EXPERIMENTS=1`, include/m4depth_hip_experiments.h;
 *   - return value: 0 on success, otherwise a hipError_t code (1 =
 *     hipErrorInvalidValue for bad arguments).  Never exits the process (the
 *     reference launcher does `exit(-1)`, backproject_op_gpu.cu.cc:95-100);
 *   - `rot` is [b, rot_c] with rot_c = 4 (quaternion w,x,y,z, not renormalised)
 *     or 3 (small-angle x,y,z) as in get_rot_mat (utils/depth_operations.py:18-53);
 *     `trans` is [b,3]; `cam_f`, `cam_c` are the level-local intrinsics [b,2]
 *     (fx,fy) / (cx,cy) (m4depth_network.py:300-302);
 *   - arithmetic is IEEE float32 with one rounding per operation (the library is
 *     built with -ffp-contract=off) in the operand order of oracle/m4depth_oracle.py,
 *     so integer index grids are bit-exact against the oracle.
 */
#ifndef M4DEPTH_HIP_H_
#define M4DEPTH_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M4D_ABI_VERSION 6   /* 6 (round 6): + m4d_normalize_levels, m4d_level_front_small(_supported) (a coarse level opens with ONE launch), m4d_conv3x3_wino6_bias_act_ks (the staggered first round as a per-launch argument), - m4d_wino6_set_stagger (process-wide state on the launch path), - m4d_conv3x3_lat_chain (measured 12.5 us per hand-over, never dispatched: deleted); 5 (round 5): + m4d_depth_metrics_strided, m4d_launch_count, m4d_conv3x3_lat, m4d_conv3x3s_lat, m4d_partial_finish, m4d_level_front_r, m4d_wino6_persistent_min_units, m4d_pack_conv_weights_lat, m4d_conv3x3_lat_chain, m4d_pyramid_reset(_supported), m4d_enc_level0_stats / _apply, m4d_wino6_set_stagger; 4 (round 4): + m4d_conv3x3_wino6_bias_act_k; the wino6 kernel selectors and the launch tape moved to m4depth_hip_experiments.h */

/* Library / device introspection (no GPU work). */
int m4d_abi_version(void);
/* Kernel launches this library has issued since it was loaded (host-side counter): the difference across one step -- or across
 * the hipGraph capture of one -- is the number of launches the step is made of (bench.py: roofline.launches_per_step). */
long long m4d_launch_count(void);
const char* m4d_build_info(void);

/* ---- the reference's native op ------------------------------------------------ */

/* BackProjectForwardLauncher (backproject_op_gpu.cu.cc:83-103; kernel :19-79).
 * dims = {B,H,W,S,F,C} as built by BackProjectOp::Compute (backproject_op.cc:67-80).
 * input [B,H,W,F,C], coords [B,H,W,S,F,2] as (x,y), out [B,H,W,S,F,C].
 * Coordinates outside [0,W-1]x[0,H-1] produce zeros. */
int m4d_backproject_fwd(const float* input, const float* coords, const int dims[6],
                        float* out, void* stream);

/* BackProjectBackwardLauncher (backproject_op_gpu.cu.cc:201-223; kernel :108-197).
 * grad [B,H,W,S,F,C] -> input_grad [B,H,W,F,C] (fp32 atomic scatter, zero-filled by
 * the call itself), coords_grad [B,H,W,S,F,2]. */
int m4d_backproject_bwd(const float* grad, const float* input, const float* coords,
                        const int dims[6], float* input_grad, float* coords_grad, void* stream);

/* ---- utils/dense_image_warp.py -------------------------------------------------- */

/* dense_image_warp (:195-268) on its TF-CPU branch (_interpolate_bilinear, :61-192):
 * query = (j,i) + flow[(row,col)], floor clamped to [0,size-2], alpha to [0,1],
 * lerp form a*(r-l)+l.  image [B,H,W,C], flow [B,H,W,2] -> out [B,H,W,C].
 * index_out (may be NULL): int32 [B,H,W,2] = (y0,x0) -- the bit-exact index grid. */
int m4d_dense_image_warp(const float* image, const float* flow, int B, int H, int W, int C,
                         float* out, int32_t* index_out, void* stream);

/* _interpolate_bilinear (:61-192) with indexing='ij': grid [B,H,W,C], query [B,N,2] as
 * (row, col) -> out [B,N,C]; index_out (may be NULL) int32 [B,N,2] = (y0,x0). */
int m4d_interpolate_bilinear(const float* grid, const float* query, int B, int H, int W, int C,
                             int N, float* out, int32_t* index_out, void* stream);

/* ---- utils/depth_operations.py: converters -------------------------------------- */

/* parallax2depth (:141-166): depth = (s/disp - tz)/alpha.  disp,out [b,h,w,1]. */
int m4d_parallax2depth(const float* disp, const float* rot, int rot_c, const float* trans,
                       const float* cam_f, const float* cam_c, int b, int h, int w,
                       float* out, void* stream);
/* depth2parallax (:169-194): disp = s/(depth*alpha + tz). */
int m4d_depth2parallax(const float* depth, const float* rot, int rot_c, const float* trans,
                       const float* cam_f, const float* cam_c, int b, int h, int w,
                       float* out, void* stream);
/* prev_d2para (:197-215); rot is accepted for signature parity and ignored. */
int m4d_prev_d2para(const float* prev_d, const float* rot, int rot_c, const float* trans,
                    const float* cam_f, const float* cam_c, int b, int h, int w,
                    float* out, void* stream);
/* The flow field reproject (:72-105) hands to dense_image_warp plus its two auxiliary
 * outputs.  depth [b,h,w,1] -> flow [b,h,w,2] (row,col), proj_minus_rot [b,h,w,2],
 * rot_coord [b,h,w,2] (either aux pointer may be NULL). */
int m4d_reproject_flow(const float* depth, const float* rot, int rot_c, const float* trans,
                       const float* cam_f, const float* cam_c, int b, int h, int w,
                       float* flow, float* proj_minus_rot, float* rot_coord, void* stream);
/* recompute_depth (:109-137). */
int m4d_recompute_depth(const float* depth, const float* rot, int rot_c, const float* trans,
                        const float* cam_f, const float* cam_c, int b, int h, int w,
                        float* out, void* stream);

/* ---- utils/depth_operations.py: cost volumes ------------------------------------ */

/* get_parallax_sweeping_cv, the DSCV (:224-281).
 * c1, c2 [b,h,w,C]; disp_prev_t, disp [b,h,w,1]; nbre_cuts divides C.
 * cv: channel kk*(2r+1)+(n+r) of pixel p is written at cv[p*cv_stride + kk*(2r+1)+(n+r)]
 *     (cv_stride >= k*(2r+1); pass k*(2r+1) for a dense [b,h,w,k*(2r+1)] tensor, or the
 *     refiner-input width to write straight into f_input).  Products are formed in
 *     float16 and averaged per cut (cv_accum: 0 = float32 sum, one rounding to half;
 *     1 = sequential half adds) as at :276-277.
 * prev_disp (may be NULL) [b,h,w,2r+1]: the warped disp_prev_t channel (:273,:280).
 * log_center (may be NULL): writes log(prev_disp[...,r] * log_scale) at
 *     log_center[p*log_stride] -- the time-recurrence feature of m4depth_network.py:238.
 * index_out (may be NULL): int32 [b,h,w,2r+1,2] = (y0,x0) of every hypothesis. */
int m4d_dscv_fwd(const float* c1, const float* c2, const float* disp_prev_t, const float* disp,
                 const float* rot, int rot_c, const float* trans, const float* cam_f,
                 const float* cam_c, int b, int h, int w, int C, int search_range, int nbre_cuts,
                 int cv_accum, float* cv, int cv_stride, float* prev_disp,
                 float* log_center, int log_stride, float log_scale,
                 int32_t* index_out, void* stream);

/* Tuning / debugging hooks of the DSCV (process-wide, not part of the reference surface):
 * variant 0 = generic kernel, 1 = wave kernel with global gathers (default: fastest measured),
 * 2 = LDS search-window kernel, 3 / 4 = LDS window with 3 / all hypotheses per lane.  The fallback counter (device uint32, may be NULL) is incremented by
 * every workgroup of the window kernel whose footprint box did not fit the LDS window. */
void m4d_dscv_set_variant(int variant);
void m4d_dscv_set_fallback_counter(unsigned int* device_counter);
/* Profiling only: ablation mask for the hypothesis-per-lane kernel (1 = skip cv stores, 2 = skip
 * corner loads, 4 = skip the centre / warped-parallax outputs).  Results are then meaningless. */
void m4d_dscv_set_ablation(int mask);
/* Profiling only: 6 uint64 per workgroup of the hypothesis-per-lane kernel = cycle counter at
 * entry / after the box reduction / after window staging / at exit, window pixels, window width. */
void m4d_dscv_set_stamps(unsigned long long* device_buffer);

/* cost_volume, the SNCV (:284-313): out channel ((y*(2r+1)+x)*k + kk) =
 * leaky_relu(mean_c c1[j,i,c] * c2pad[j+y*d, i+x*d, c], 0.1), written at
 * out[p*out_stride + channel]. */
int m4d_sncv_fwd(const float* c1, const float* c2, int b, int h, int w, int C, int search_range,
                 int dilation_rate, int nbre_cuts, float* out, int out_stride, void* stream);
/* m4d_dscv_fwd and m4d_sncv_fwd(c1, c1, ...) of one level (depth_operations.py:224-281 and :284-313 as called from
 * m4depth_network.py:220-233).  On small maps (the coarsest pyramid levels) the two independent volumes share ONE launch --
 * one kernel boundary less on the coarse-level latency chain; otherwise the two entries are called one after the other.
 * Same results, bit for bit. */
int m4d_dscv_sncv_fwd(const float* c1, const float* c2, const float* disp_prev_t, const float* disp,
                      const float* rot, int rot_c, const float* trans, const float* cam_f,
                      const float* cam_c, int b, int h, int w, int C, int search_range, int nbre_cuts,
                      int cv_accum, float* cv, int cv_stride, float* prev_disp,
                      float* log_center, int log_stride, float log_scale,
                      int sncv_search_range, float* sncv_out, int sncv_out_stride, void* stream);

/* The same stride-1 layer by Winograd F(2x2,3x3) on the fp32 matrix cores: 2.25x fewer multiply-adds, result
 * equal to the direct convolution up to float32 rounding (the transforms only add and halve).  wu = the filter
 * transformed on the host, U = G g G^T, packed [ceil(Cin/16)][16 positions][CoutPad][16 channels]
 * (network_ops.pack_conv_weights_winograd); CoutPad a multiple of 32.  Deterministic. */
int m4d_conv3x3_wino_bias_act(const float* x, const float* wu, const float* bias, int b, int h, int w,
                              int Cin, int Cout, int CoutPad, float slope, float* out, void* stream);
/* Variant with half the weight traffic per flop (16x16-pixel workgroup tile, 8-channel K chunks): wu8 packed
 * [ceil(Cin/8)][16 positions][CoutPad][8 channels]; Cin % 4 == 0, x 16-byte aligned. */
int m4d_conv3x3_wino2_bias_act(const float* x, const float* wu8, const float* bias, int b, int h, int w,
                               int Cin, int Cout, int CoutPad, float slope, float* out, void* stream);
/* Profiling only: 202 uint64 per workgroup (first 512 of batch item 0) = cycle counter of wave 0 at kernel entry / exit and,
 * per K chunk (first 40), at chunk start / after the first barrier / after the input transform / after the second barrier /
 * after the MFMA loop.  NULL switches it off. */
void m4d_wino_set_stamps(unsigned long long* device_buffer);
/* The same convolution with float32 operands on the BF16 matrix cores: every float32 operand is the exact sum of three
 * bf16 terms, six of the nine term products are accumulated in float32 (the dropped ones are below float32's product
 * rounding): float32 accuracy at 2.67x less matrix-core time.  wu6 = the transformed filter split and packed on the host
 * (network_ops.pack_conv_weights_wino6: [Cin/16][CoutPad/64][16][2][3][64][8] bf16); Cin % 16 == 0, CoutPad % 64 == 0.
 * Replaces the same tf.keras Conv2D + leaky_relu pairs as m4d_conv3x3_wino2_bias_act (m4depth_network.py:101-131). */
int m4d_conv3x3_wino6_bias_act(const float* x, const void* wu6, const float* bias, int b, int h, int w,
                               int Cin, int Cout, int CoutPad, float slope, float* out, void* stream);
/* The same call with the kernel named: 0 = chosen from the grid (what m4d_conv3x3_wino6_bias_act does), 1 = one workgroup
 * per (16x16-pixel tile, 64 output channels) (csrc/m4d_wino6.hip), 2 = persistent workgroups, one per CU, each walking a
 * contiguous range of (tile, cout group) units with the DMA stream continuing across unit boundaries (csrc/m4d_wino6p.hip;
 * Cin >= 32); 16 + n (1 <= n <= 15, round 6) = persistent workgroups of n CONSECUTIVE units each (n = 2 on a 128-cout layer: a
 * tile's two cout groups, one halo fetch, one unit boundary with the DMA stream running through it), as many workgroups as that
 * takes -- the dispatcher places them as CUs free instead of one static range per CU.  Bit-identical results; an argument, not
 * process state: tests and A/B timing use it. */
int m4d_conv3x3_wino6_bias_act_k(const float* x, const void* wu6, const float* bias, int b, int h, int w,
                                 int Cin, int Cout, int CoutPad, float slope, float* out, int kernel, void* stream);
/* Grids of at least this many (16x16-pixel tile, 64-cout) units run on the persistent-workgroup form of the kernel above when
 * `kernel` is 0 (what bench.py needs to name the kernel its roofline layer ran on). */
long long m4d_wino6_persistent_min_units(void);
/* Profiling only (tools/wino6_phases.py): per-position cycle stamps of the first 64 workgroups; NULL switches it off. */
void m4d_wino6_set_stamps(unsigned long long* device_buffer);
/* The same call with a STAGGERED FIRST ROUND as well (ABI 6; replaces round 5's process-wide m4d_wino6_set_stagger): with
 * stagger_us > 0 the one-workgroup-per-unit kernel starts its first 256 workgroups (one per CU) in stagger_phases groups (a
 * power of two <= 32) spread over stagger_us microseconds (<= 1000): identical workgroups would otherwise free every CU at the
 * same instant once per unit time, and the small kernels of another frame's coarse levels wait for that instant
 * (csrc/m4d_wino6.hip).  A per-launch argument like `kernel`: nothing is remembered between calls, two callers with different
 * settings never see each other's (tests/test_gpu_ops.py::test_winograd_stagger_is_per_call).  Ignored by the persistent
 * kernel; results are the same bits either way.  Which launches should carry it is the caller's policy (network.py: grids of
 * >= 200 workgroups at batch <= 4; whether it pays depends on the box). */
int m4d_conv3x3_wino6_bias_act_ks(const float* x, const void* wu6, const float* bias, int b, int h, int w,
                                  int Cin, int Cout, int CoutPad, float slope, float* out, int kernel,
                                  int stagger_us, int stagger_phases, void* stream);

/* The tail of a level in one kernel: the last two DispRefiner convolutions (32 -> 16 + leaky_relu(0.1), 16 -> 5;
 * m4depth_network.py:109-135) and m4d_level_post (:247-260).  x32 [b,h,w,32]; w6p [9][16][32] = kernel[ky][kx][k][n] as
 * [tap][n][k]; w7p [9][16][16] likewise with rows n >= 5 zero; outputs as m4d_level_post. */
int m4d_refiner_tail(const float* x32, const float* w6p, const float* b6, const float* w7p, const float* b7,
                     const float* rot, int rot_c, const float* trans, const float* cam_f, const float* cam_c,
                     int b, int h, int w, float scale, float* parallax, float* depth, float* other,
                     float* depth_state, void* stream);
/* The same tail with float32 operands split exactly into three bf16 terms on the bf16 matrix cores (csrc/m4d_tail6.hip; the
 * arithmetic of m4d_conv3x3_wino6_bias_act), persistent workgroups over 14x10-pixel tiles.  w6f / w7f = the B fragments of
 * network_ops.pack_refiner_tail_weights6: [9 taps][3 parts][64 lanes][8 bf16] with lane = (k-quarter, cout) holding input
 * channels 8 kq .. 8 kq + 7, and [5 K-steps][3 parts][4 k-quarters][8 couts][8 bf16] with k-quarter kq holding channels
 * 8 (kq & 1) .. + 7 of tap 2 j + (kq >> 1) (zeros for the tenth tap and couts 5..7). */
int m4d_refiner_tail6(const float* x32, const void* w6f, const float* b6, const void* w7f, const float* b7,
                      const float* rot, int rot_c, const float* trans, const float* cam_f, const float* cam_c,
                      int b, int h, int w, float scale, float* parallax, float* depth, float* other,
                      float* depth_state, void* stream);

/* ---- gradients of the cost volumes (train_step, m4depth_network.py:371-399) ------ */

/* Backward of m4d_dscv_fwd: what tf.GradientTape derives from depth_operations.py:224-281.
 * g_cv: gradient of cv, read at g_cv[p*g_cv_stride + kk*(2r+1)+t]; g_prev_disp (may be NULL)
 * [b,h,w,2r+1]: gradient of the warped-parallax output.  The float16 product / mean (:276-278)
 * are differentiated in float16 as TF does; floor has no gradient and the two clip_by_value
 * pass inside their bounds (dense_image_warp.py:147-154, depth_operations.py:236).
 * Outputs: g_c1 [b,h,w,C], g_disp [b,h,w,1] (plain stores); g_c2 [b,h,w,C] and the optional
 * g_disp_prev_t [b,h,w,1] are scatter-adds (zero-filled by the call, float32 atomics: the
 * summation order varies run to run, as in BackProjectBackward, backproject_op_gpu.cu.cc:108-197).
 * C % 4 == 0, (C / nbre_cuts) % 4 == 0, C <= 256, 16-byte aligned feature pointers. */
int m4d_dscv_bwd(const float* c1, const float* c2, const float* disp_prev_t, const float* disp,
                 const float* rot, int rot_c, const float* trans, const float* cam_f,
                 const float* cam_c, int b, int h, int w, int C, int search_range, int nbre_cuts,
                 const float* g_cv, int g_cv_stride, const float* g_prev_disp,
                 float* g_c1, float* g_c2, float* g_disp, float* g_disp_prev_t, void* stream);

/* Backward of m4d_sncv_fwd (depth_operations.py:284-313).  out = the forward result (decides
 * the leaky_relu branch), g = its gradient, both read with their own pixel strides.
 * g_c1 / g_c2 [b,h,w,C] (either may be NULL); when c1 and c2 are the same map (the model's
 * auto-correlation, m4depth_network.py:232) the caller adds the two.  Deterministic. */
int m4d_sncv_bwd(const float* c1, const float* c2, const float* out, int out_stride,
                 const float* g, int g_stride, int b, int h, int w, int C, int search_range,
                 int dilation_rate, int nbre_cuts, float slope, float* g_c1, float* g_c2,
                 void* stream);

/* ---- per-pixel glue of the training graph (one kernel per autograd node) ----------- */

/* Adjoint of m4d_resize_bilinear_v1: g_out [b,oh,ow,c] -> g_in [b,ih,iw,c] (gather: deterministic). */
int m4d_resize_bilinear_v1_bwd(const float* g_out, int b, int ih, int iw, int c, int oh, int ow,
                               float mul, float* g_in, void* stream);
/* Backward of m4d_level_post (m4depth_network.py:247-251): gradients of parallax / depth / other
 * (any may be NULL = zero) -> gradient of the refiner output [b,h,w,5]. */
int m4d_level_post_bwd(const float* refiner_out, const float* g_parallax, const float* g_depth,
                       const float* g_other, const float* rot, int rot_c, const float* trans,
                       const float* cam_f, const float* cam_c, int b, int h, int w, float scale,
                       float* g_refiner_out, void* stream);
/* Backward of m4d_normalize_cuts: g_x = g/n - x * sum(g*x)/n^3 per cut. */
int m4d_normalize_cuts_bwd(const float* x, const float* g, int b, int h, int w, int C, int nbre_cuts,
                           float* g_x, void* stream);
/* Backward of the convolution epilogue leaky_relu(conv + bias): g_pre = g * (out > 0 ? 1 : slope)
 * (tf.nn.leaky_relu's gradient), g_bias[c] = sum over rows of g_pre (two-stage, fixed order).
 * workspace: m4d_bias_act_bwd_workspace_floats(rows, C) floats. */
long long m4d_bias_act_bwd_workspace_floats(long long rows, int C);
int m4d_bias_act_bwd(const float* g, const float* out, long long rows, int C, float slope,
                     float* g_pre, float* g_bias, float* workspace, void* stream);
/* Device-side weight packing for m4d_conv3x3*_bias_act*: w_ohwi = [O][3][3][I] (the channels-last
 * memory of an OIHW parameter) -> [ceil(K/16)][9][Npad][16]; transpose = 0: K = I, N = O (forward),
 * 1: K = O, N = I with the taps rotated by 180 degrees (data gradient of the stride-1 layer). */
int m4d_pack_conv_weights(const float* w_ohwi, int O, int I, int transpose, float* wp, void* stream);
/* ... in the layout of m4d_conv3x3_lat ([N/32][K/16][9][3 parts][64 lanes][8] bf16, every weight split exactly into three bf16
 * terms: what network_ops.pack_conv_weights_lat builds on the host), for the forward (transpose = 0: K = I, N = O) or the data
 * gradient (transpose = 1: the rotated / transposed kernel, K = O, N = I); wp: ceil(N/32) * ceil(K/16) * 27 KB. */
int m4d_pack_conv_weights_lat(const float* w_ohwi, int O, int I, int transpose, void* wp, void* stream);
/* One level's term of m4depth_loss (m4depth_network.py:491-536), unweighted:
 * mean |resize(log clip(gt)) - log clip(pred)| ('map': tf.image.resize bilinear, :532) or the
 * hole-aware block mean of velodyne ground truth (:514-527).  out2 = (term, point count);
 * workspace: m4d_loss_workspace_floats() floats.  Backward: g_pred [b,h,w,1] from the scalar
 * gradient g_out (device) and out2 of the forward. */
long long m4d_loss_workspace_floats(void);
int m4d_loss_level_fwd(const float* pred_depth, const float* gt_depth, int b, int h, int w, int H, int W,
                       int velodyne, float* workspace, float* out2, void* stream);
int m4d_loss_level_bwd(const float* pred_depth, const float* gt_depth, const float* stats2,
                       const float* g_out, int b, int h, int w, int H, int W, int velodyne,
                       float* g_pred, void* stream);

/* ---- m4depth_network.py: DepthEstimatorLevel glue -------------------------------- */

/* Per-cut L2 normalisation (:179-189, tf.linalg.normalize, no epsilon). */
int m4d_normalize_cuts(const float* x, int b, int h, int w, int C, int nbre_cuts,
                       float* out, void* stream);

/* tf.compat.v1.image.resize_bilinear defaults (:202-204): x [b,ih,iw,c] -> out
 * [b,oh,ow,c], multiplied by `mul` (2.0 for the parallax map, :203). */
int m4d_resize_bilinear_v1(const float* x, int b, int ih, int iw, int c, int oh, int ow,
                           float mul, float* out, void* stream);
/* tf.image.resize(NEAREST_NEIGHBOR) (:368). */
int m4d_resize_nearest(const float* x, int b, int ih, int iw, int c, int oh, int ow,
                       float* out, void* stream);

/* Fused "preprocessor" glue of one level (:196-204, :218, :224-227):
 *   prev_l_* (coarser level estimate at [b,ph,pw,.]; all NULL at the coarsest level ->
 *   parallax 1, depth 1000, other 0) are x2-upsampled to para_prev_l [b,h,w,1] (x2),
 *   depth_prev_l [b,h,w,1], other_prev_l [b,h,w,4];
 *   para_prev_t [b,h,w,1] = prev_d2para(depth_prev_t) (skipped when depth_prev_t NULL);
 *   if f_input != NULL: f_input[p*f_stride + log_off] = log(para_prev_l * log_scale) and,
 *   when other_off >= 0, f_input[p*f_stride + other_off + 0..3] = other_prev_l;
 *   if depth_state_reset != NULL (the new_traj branch, :208-214): depth_state_reset [b,h,w,1] := 1000 (:209). */
int m4d_level_pre(const float* prev_l_depth, const float* prev_l_parallax, const float* prev_l_other,
                  int ph, int pw, const float* depth_prev_t, const float* trans,
                  const float* cam_f, const float* cam_c, int b, int h, int w,
                  float* para_prev_l, float* depth_prev_l, float* other_prev_l, float* para_prev_t,
                  float* f_input, int f_stride, int log_off, int other_off, float log_scale,
                  float* depth_state_reset, void* stream);
/* m4d_level_pre and m4d_normalize_cuts(norm_x -> norm_out) -- the two independent kernels that open a level
 * (m4depth_network.py:179-189 and :196-227) -- in ONE launch: one kernel boundary less on the coarse-level latency chain.
 * Same results, bit for bit. */
int m4d_level_pre_normalize(const float* prev_l_depth, const float* prev_l_parallax, const float* prev_l_other,
                            int ph, int pw, const float* depth_prev_t, const float* trans,
                            const float* cam_f, const float* cam_c, int b, int h, int w,
                            float* para_prev_l, float* depth_prev_l, float* other_prev_l, float* para_prev_t,
                            float* f_input, int f_stride, int log_off, int other_off, float log_scale,
                            float* depth_state_reset,
                            const float* norm_x, int C, int nbre_cuts, float* norm_out, void* stream);

/* The reset frame of a whole pyramid in ONE launch (m4depth_network.py:207-214 for every level of DepthEstimatorPyramid.call's
 * loop, :305-321, on a new trajectory): per level, prev_f_maps := the per-cut normalised features, depth_prev_t := 1000, and the
 * level's estimate -- the x2 upsampling chain of :202-204 started from the constants of :198-200, i.e. the constant maps
 * parallax = parallax_value (the caller passes 2^(number of coarser levels)), depth = 1000, other = 0.  Bit-identical to
 * m4d_level_pre_normalize called level by level, coarse to fine.  levels[] is read on the host during the call. */
typedef struct m4d_reset_level {
  const float* features;   /* [b,h,w,C] raw encoder features */
  float* state_features;   /* [b,h,w,C] <- normalised (becomes prev_f_maps) */
  float* depth_state;      /* [b,h,w,1] <- 1000 */
  float* parallax;         /* [b,h,w,1] <- parallax_value */
  float* depth;            /* [b,h,w,1] <- 1000 */
  float* other;            /* [b,h,w,4] <- 0 */
  int h, w, C, nbre_cuts;
  float parallax_value;
} m4d_reset_level;
int m4d_pyramid_reset_supported(int C, int nbre_cuts);      /* C / nbre_cuts in {8, 16, 24, 32} */
int m4d_pyramid_reset(const m4d_reset_level* levels, int n_levels, int b, void* stream);

/* The per-cut normalisation (tf.linalg.normalize per cut, m4depth_network.py:179-189) of up to 8 feature maps in ONE launch: the
 * normalised features of a frame depend on the encoder alone, so the coarse levels' maps of all frames of an encoder batch are
 * normalised right behind it, off the levels' latency chains (round 6).  x / out [pixels, C], pixels = frames * b * h * w; C /
 * nbre_cuts in {8, 16, 24, 32}.  The same bits as m4d_normalize_cuts. */
typedef struct { const float* x; float* out; long long pixels; int C, nbre_cuts; } m4d_norm_level;
int m4d_normalize_levels(const m4d_norm_level* levels, int n_levels, void* stream);
/* The opening of a COARSE level (maps <= 6000 pixels: levels 4-6 of the 384x1280 pyramid at batch 1) in ONE launch (round 6):
 * level_pre's glue + DSCV + SNCV of m4depth_network.py:196-242 on features that arrive per-cut normalised (norm_f [b,h,w,C]:
 * m4d_normalize_levels; it becomes prev_f_maps after the level, :211/:259).  [DSCV | SNCV | glue] workgroups of one grid, nothing
 * inside the launch depends on anything else in it: the DSCV evaluates the two maps the glue kernel would have written where it
 * needs them (para_prev_l = 2 x the x2 upsampling of the coarser level's parallax at its own pixel, prev_d2para of depth_prev_t
 * at the centre hypothesis' four corners), the glue workgroups write the log / memory features.  f_input [b,h,w,f_stride]:
 * channel order of m4d_level_front, padding channels untouched (the caller keeps them zero).  prev_l_parallax / prev_l_other
 * [b,ph,pw,1] / [b,ph,pw,4]: both NULL at the coarsest level.  Bit-identical to m4d_level_pre_normalize + m4d_dscv_sncv_fwd
 * (tests/test_gpu_ops.py::test_level_front_small_is_bitwise_the_separate_launches).  _supported: the pyramid's (C, cuts)
 * pairs, DSCV range 4 or 2. */
int m4d_level_front_small_supported(int C, int nbre_cuts, int dscv_range, int sncv_range);
int m4d_level_front_small(const float* norm_f, const float* prev_f, const float* depth_prev_t,
                          const float* prev_l_parallax, const float* prev_l_other, int ph, int pw,
                          const float* rot, int rot_c, const float* trans, const float* cam_f, const float* cam_c,
                          int b, int h, int w, int C, int nbre_cuts, int dscv_range, int sncv_range, int cv_accum,
                          float* f_input, int f_stride, float log_scale, void* stream);

/* The fused level front (m4depth_network.py:179-242 in one launch, default settings: all ablation blocks on, DSCV range 4,
 * SNCV range 3): per-cut normalisation of raw_f [b,h,w,C] -> norm_out (the buffer that becomes prev_f_maps, :211/:259);
 * x2 upsampling of the coarser level's parallax / other maps ([b,ph,pw,1] / [b,ph,pw,4]; both NULL at the coarsest level),
 * prev_d2para of depth_prev_t, DSCV against prev_f, SNCV of the normalised features, both log features -- assembled into
 * whole rows of f_input [b,h,w,f_stride] (channel order cv | log para_l | other(4) | sncv | log para_t, padding channels
 * zeroed), written as contiguous runs.  Bit-identical to m4d_level_pre_normalize + m4d_dscv_fwd + m4d_sncv_fwd.
 * m4d_level_front_supported: 1 if the (C, cuts, ranges, row stride) combination has a kernel (f_stride must be 58*cuts+6
 * rounded up to a multiple of 8), else 0 -- the caller then uses the three separate entry points. */
int m4d_level_front_supported(int C, int nbre_cuts, int dscv_range, int sncv_range, int f_stride);
void m4d_front_set_stamps(unsigned long long* device_buffer);   /* profiling: 8 x u64 per workgroup, phase cycle stamps */
int m4d_level_front(const float* raw_f, float* norm_out, const float* prev_f, const float* depth_prev_t,
                    const float* prev_l_parallax, const float* prev_l_other, int ph, int pw,
                    const float* rot, int rot_c, const float* trans, const float* cam_f, const float* cam_c,
                    int b, int h, int w, int C, int nbre_cuts, int cv_accum,
                    float* f_input, int f_stride, float log_scale, void* stream);
/* ... with the search ranges as arguments: (dscv_range, sncv_range) = (4, 3), the reference's hard-coded windows
 * (m4depth_network.py:221,232), or (6, 6), BASELINE configs[4] (13 hypotheses, 13x13 window: levels 1-5 of the 6-level pyramid;
 * f_stride = (2 dscv_range + 1 + (2 sncv_range + 1)^2) * cuts + 6 rounded up to a multiple of 8).  Bit-identical to the three
 * separate entry points with the same ranges. */
int m4d_level_front_r(const float* raw_f, float* norm_out, const float* prev_f, const float* depth_prev_t,
                      const float* prev_l_parallax, const float* prev_l_other, int ph, int pw,
                      const float* rot, int rot_c, const float* trans, const float* cam_f, const float* cam_c,
                      int b, int h, int w, int C, int nbre_cuts, int dscv_range, int sncv_range, int cv_accum,
                      float* f_input, int f_stride, float log_scale, void* stream);

/* Level-local intrinsics of DepthEstimatorPyramid.call (m4depth_network.py:300-302) for all levels in one launch:
 * f_out / c_out [levels,b,2], level l (0 = finest) = cam / 2^(l+1). */
int m4d_camera_pyramid(const float* cam_f, const float* cam_c, int b, int levels, float* f_out, float* c_out,
                       void* stream);

/* Fused "depth_estimator" tail of one level (:247-260): refiner_out [b,h,w,5] ->
 * parallax = exp(clip(out0,-7,7)) / scale, other = out[1:5], depth =
 * parallax2depth(parallax); depth is also stored into depth_state (may be NULL). */
int m4d_level_post(const float* refiner_out, const float* rot, int rot_c, const float* trans,
                   const float* cam_f, const float* cam_c, int b, int h, int w, float scale,
                   float* parallax, float* depth, float* other, float* depth_state, void* stream);

/* Convolution epilogue on an NHWC activation [rows, C] (rows = b*h*w): out = leaky_relu(x +
 * bias[c], slope) -- the bias add of Keras Conv2D and the tf.nn.leaky_relu(., 0.1) that
 * follows it (m4depth_network.py:84,87,123,133); slope = 1 gives the bias add alone (last
 * refiner convolution, :132).  In place when out == x. */
int m4d_bias_act(const float* x, const float* bias, long long rows, int C, float slope,
                 float* out, void* stream);

/* m4d_bias_act whose result lands inside a larger, zero-bordered [b,out_h,out_w,C] buffer at
 * (off_y,off_x): the TF 'SAME' padding of the stride-2 encoder convolution that consumes it
 * (m4depth_network.py:68-72; bottom/right only for even sizes) without a separate pad pass.
 * The caller keeps the border zero (the kernel never writes it).  C % 4 == 0. */
int m4d_bias_act_padded(const float* x, const float* bias, int b, int h, int w, int C, float slope,
                        float* out, int out_h, int out_w, int off_y, int off_x, void* stream);

/* Keras Conv2D(Cout, 3, strides 1, padding 'same') + bias + leaky_relu(slope) (slope = 1: no
 * activation) in one launch on the f32-input matrix cores (m4depth_network.py:104-135, :63-67).
 * x [b,h,w,Cin] NHWC (Cin even), out [b,h,w,Cout].  wp: the HWIO kernel re-packed by the host as
 * [ceil(Cin/16)][9 taps (ky*3+kx)][CoutPad][16] with CoutPad = Cout rounded up to 32, input
 * channels of a chunk stored even-first (0,2,..,14,1,3,..,15), zero padded
 * (m4depth_amd.network_ops.pack_conv_weights).  Deterministic; float32 exact (fmaf chain over
 * chunk, ky, kx, channel). */
int m4d_conv3x3_bias_act(const float* x, const float* wp, const float* bias, int b, int h, int w,
                         int Cin, int Cout, int CoutPad, float slope, float* out, void* stream);
/* Same, with a caller-owned workspace (m4d_conv3x3_workspace_floats floats) that lets small
 * problems (coarse pyramid levels: a handful of pixel tiles, K up to 9*470) split K over
 * workgroups; partial sums are reduced in split order by a second kernel (still deterministic). */
long long m4d_conv3x3_workspace_floats(int b, int h, int w, int CoutPad);
/* General form: stride 1 or 2 with TensorFlow 'SAME' padding computed inside (stride 2 on an even
 * size pads bottom/right only -- no padded copy of the input is needed); out is
 * [b, ceil(h/stride), ceil(w/stride), Cout].  Cin may be odd (the 3-channel input image). */
int m4d_conv3x3s_bias_act_ws(const float* x, const float* wp, const float* bias, int b, int h, int w,
                             int Cin, int Cout, int CoutPad, int stride, float slope, float* out,
                             float* workspace, long long workspace_floats, void* stream);
int m4d_conv3x3_bias_act_ws(const float* x, const float* wp, const float* bias, int b, int h, int w,
                            int Cin, int Cout, int CoutPad, float slope, float* out, float* workspace,
                            long long workspace_floats, void* stream);
/* Latency-first form of the same stride-1 layer for SMALL maps at batch 1 (round 5, csrc/m4d_convlat.hip; the DispRefiner
 * convolutions of the coarse pyramid levels, m4depth_network.py:116-135): every wave requests everything it needs -- its weight
 * fragments of at most two 16-channel chunks straight into MFMA operand registers, then the workgroup's halo -- in ONE memory
 * round trip; K is split over the waves of a workgroup (kw sub-slices, added in wave order) and over s_out workgroups, whose
 * raw partial tiles go to s_out slabs (out + z * out_slab_floats) that the CONSUMING call adds in slab order, + the producer's
 * bias and leaky_relu, while it stages its input (s_in / x_slab_floats / x_bias / x_slope; x_bias NULL = x is a finished
 * activation).  s_out == 1 writes the finished activation leaky_relu(sum + bias, slope).  float32 operands as exact 3 x bf16
 * splits (arithmetic of m4d_conv3x3_small6_bias_act; another summation order over K).  wp = network_ops.pack_conv_weights_lat
 * ([Cout/32][Cin/16][9 taps][3 parts][64 lanes][8] bf16: MFMA B-fragment order).  mt = 1 / 2 / 4 M-tiles (8x4 / 8x8 / 16x8
 * pixels per workgroup), kw = 1 / 2 / 4, 1 <= s_in <= 4, Cin >= 16, Cin % 4 == 0.  Deterministic. */
int m4d_conv3x3_lat(const float* x, int s_in, long long x_slab_floats, const float* x_bias, float x_slope,
                    const void* wp, const float* bias, int b, int h, int w, int Cin, int Cout, float slope,
                    int mt, int kw, int s_out, float* out, long long out_slab_floats, void* stream);
/* ... with stride 1 or 2 (TF 'SAME'; the coarse stride-2 layers of the encoder, m4depth_network.py:66-72): out / every slab
 * [b, ceil(h/stride), ceil(w/stride), Cout]. */
int m4d_conv3x3s_lat(const float* x, int s_in, long long x_slab_floats, const float* x_bias, float x_slope,
                     const void* wp, const float* bias, int b, int h, int w, int Cin, int Cout, int stride, float slope,
                     int mt, int kw, int s_out, float* out, long long out_slab_floats, void* stream);
/* out = leaky_relu(bias + slab_0 + ... + slab_{s_in-1}, slope), slabs added in slab order: the finished form of a partial-sum
 * activation, for consumers that do not add the slabs themselves and for inspection.  C % 4 == 0. */
int m4d_partial_finish(const float* x, int s_in, long long x_slab_floats, const float* bias, float slope,
                       long long pixels, int C, float* out, void* stream);

/* The same stride-1 layer for SMALL maps (the DispRefiner convolutions of the coarsest pyramid levels, m4depth_network.py:
 * 116-135 at 6x20 ... 24x80 pixels): ONE launch -- a workgroup is an 8x4-pixel tile x 32 output channels whose four waves
 * walk interleaved 16-channel K chunks independently and add their partial tiles in wave order -- instead of the split-K
 * pair (partial sums + ordered reduce) the general entry needs there.  Cin >= 16, Cin % 4 == 0; same packed weights;
 * deterministic.  The caller picks it by map size (m4depth_amd.network: refiner layers with b*h*w <= 2048 and Cin <= 256). */
int m4d_conv3x3_small_bias_act(const float* x, const float* wp, const float* bias, int b, int h, int w,
                               int Cin, int Cout, int CoutPad, float slope, float* out, void* stream);
/* ... with stride 1 or 2 (the coarse stride-2 layers of the encoder): out [b, ceil(h/stride), ceil(w/stride), Cout]. */
int m4d_conv3x3s_small_bias_act(const float* x, const float* wp, const float* bias, int b, int h, int w,
                                int Cin, int Cout, int CoutPad, int stride, float slope, float* out, void* stream);
/* ... with float32 operands split exactly into three bf16 terms on the bf16 matrix cores (float32 accuracy, see
 * m4d_conv3x3_wino6_bias_act); wp6 = network_ops.pack_conv_weights_small6 ([Cin/16][9][CoutPad][3][16] bf16). */
int m4d_conv3x3_small6_bias_act(const float* x, const void* wp6, const float* bias, int b, int h, int w,
                                int Cin, int Cout, int CoutPad, float slope, float* out, void* stream);

/* DomainNormalization (m4depth_network.py:44-48) fused with the leaky_relu(slope) that follows it
 * at encoder level 0 (:82-84; slope = 1 for the normalisation alone).  x, out [b,h,w,C] (C = 16 or
 * 32); mean and the two-pass variance over (h,w) per (b,c); (x-mean)/(var+1e-12); l2-normalise
 * over channels; scale*n + bias.  workspace: m4d_dinl_workspace_floats(b, C) floats. */
long long m4d_dinl_workspace_floats(int b, int C);
/* ... written into a [b,out_h,out_w,C] buffer at (off_y,off_x) (see m4d_bias_act_padded). */
int m4d_dinl_fwd_padded(const float* x, const float* scale, const float* bias, int b, int h, int w, int C,
                        float slope, float* workspace, float* out, int out_h, int out_w, int off_y,
                        int off_x, void* stream);
int m4d_dinl_fwd(const float* x, const float* scale, const float* bias, int b, int h, int w, int C,
                 float slope, float* workspace, float* out, void* stream);

/* Encoder level 0 (FeaturePyramid.call, m4depth_network.py:79-87 with DINL): the head of the network as two calls.
 * m4d_enc_head_fwd: conv3x3(images, w_hwio [3,3,3,16]) + bias -> raw_out [b,h,w,16], and the DINL statistics of
 *   raw_out into workspace (n = m4d_dinl_workspace_floats(b,16) floats): mean = the b*16 floats at n - 2*b*16, var the last b*16.
 *   Image n of the b = frames*bsz images starts at images + (n % bsz)*stride_b + (n / bsz)*stride_t floats: the frames of
 *   a [bsz,T,H,W,3] sequence batch (M4Depth.call encodes every frame, :358-360) are read in place, frame-major; a dense
 *   [b,h,w,3] batch is bsz = b, stride_b = h*w*3, stride_t = 0.
 * m4d_conv3x3s2_dinl_bias_act: the stride-2 convolution on leaky_relu(DomainNormalization(raw), dn_slope), the
 *   normalisation fused into its input staging; wp packed as for m4d_conv3x3s_bias_act_ws (CoutPad = 32). */
/* The same level in ONE call that never materialises the [b,h,w,16] map: conv3x3(3 -> 16) is recomputed from the images on
 * 16x16x4 MFMAs in each of three passes (channel sums, squared deviations, normalise + leaky_relu(dn_slope) + conv3x3
 * stride 2 (w2_hwio [3,3,16,16], TF 'SAME') + bias2 + leaky_relu(slope) -> out [b,(h+1)/2,(w+1)/2,16]).  Images addressed as
 * in m4d_enc_head_fwd; workspace m4d_dinl_workspace_floats(b,16) floats (mean / var left there as by m4d_enc_head_fwd). */
int m4d_enc_level0_fwd(const float* images, int bsz, long long stride_b, long long stride_t,
                       const float* w1_hwio, const float* bias1, const float* dn_scale, const float* dn_bias,
                       float dn_slope, const float* w2_hwio, const float* bias2, float slope,
                       int b, int h, int w, float* workspace, float* out, void* stream);
/* m4d_enc_level0_fwd as its two halves: the statistics passes (mean / var [b,16] of conv1's output per image and channel;
 * workspace >= b * kDinlMaxBlocks * 16 floats = m4d_dinl_workspace_floats(b,16) minus the mean / var tail) and the fused
 * normalise + stride-2 convolution pass reading them.  Per-image arithmetic: the statistics of all frames of a sequence may be
 * taken in one call and the frames applied in several (same bits as m4d_enc_level0_fwd on any grouping). */
int m4d_enc_level0_stats(const float* images, int bsz, long long stride_b, long long stride_t,
                         const float* w1_hwio, const float* bias1, int b, int h, int w, float* workspace,
                         float* mean, float* var, void* stream);
int m4d_enc_level0_apply(const float* images, int bsz, long long stride_b, long long stride_t,
                         const float* w1_hwio, const float* bias1, const float* mean, const float* var,
                         const float* dn_scale, const float* dn_bias, float dn_slope, const float* w2_hwio,
                         const float* bias2, float slope, int b, int h, int w, float* out, void* stream);
int m4d_enc_head_fwd(const float* images, int bsz, long long stride_b, long long stride_t,
                     const float* w_hwio, const float* bias, int b, int h, int w, int C,
                     float* workspace, float* raw_out, void* stream);
int m4d_conv3x3s2_dinl_bias_act(const float* x_raw, const float* mean, const float* var, const float* dn_scale,
                                const float* dn_bias, float dn_slope, const float* wp, const float* bias,
                                int b, int h, int w, int Cout, int CoutPad, float slope, float* out, void* stream);

/* The 7 metrics of metrics.py (AbsRel, SqRel, RMSE, RMSE_log, Delta1..3) of one batch in one
 * pass, including test_step's clipping gt in [0,max_d], est in [0.001,max_d]
 * (m4depth_network.py:465-467).  gt, est: n floats; out7: 7 floats (main.py:127-130 order);
 * workspace: m4d_metrics_workspace_bytes() bytes.  total7 (may be NULL): the 7 Keras-Mean totals, += out7 in the same
 * launch (compiled_metrics.update_state, :470); mean7 (may be NULL) = total7 / count, what test_step returns (:473-474). */
long long m4d_metrics_workspace_bytes(void);
int m4d_depth_metrics(const float* gt, const float* est, long long n, float max_d, void* workspace,
                      float* out7, float* total7, float count, float* mean7, void* stream);
/* The same with the ground truth read in place out of a larger tensor: image j (per_image floats) starts at
 * gt + j * gt_image_stride -- the last frame of a [b,T,H,W,1] sequence batch (test_step, m4depth_network.py:455-456) without
 * the dense copy a framework slice costs at batch > 1.  Same per-thread element order as the dense form: the same bits. */
int m4d_depth_metrics_strided(const float* gt, long long per_image, long long gt_image_stride, const float* est,
                              long long n, float max_d, void* workspace, float* out7, float* total7, float count,
                              float* mean7, void* stream);

/* ---- training without MIOpen (train_step, m4depth_network.py:371-399: tf.GradientTape through every Conv2D) ---------- */

/* Weight gradient of the 3x3 TF-'SAME' convolution y = conv(x [b,h,w,cin], W), stride 1 or 2, given g = dL/dy
 * [b,ceil(h/s),ceil(w/s),cout]: dw [cout][3][3][cin] floats (the memory layout of an OIHW parameter with channels-last
 * strides).  fp32 MFMA, two-stage fixed-order reduction over the pixels (deterministic).  cin < 8 is the 3-channel image
 * layer (stride 1, cout <= 32).  workspace: m4d_conv3x3_wgrad_workspace_floats(...) floats. */
long long m4d_conv3x3_wgrad_workspace_floats(int b, int h, int w, int cin, int cout, int stride);
int m4d_conv3x3_wgrad(const float* x, const float* g, int b, int h, int w, int cin, int cout, int stride,
                      float* workspace, long long workspace_floats, float* dw, void* stream);
/* g [b,ceil(h/2),ceil(w/2),C] spread onto the [b,h,w,C] grid of a stride-2 layer's input (zeros elsewhere), placed so that
 * the layer's data gradient is the stride-1 'SAME' convolution of the result with the rotated, transposed kernel
 * (m4d_pack_conv_weights(transpose = 1) + m4d_conv3x3s_bias_act_ws). */
int m4d_dilate2(const float* g, int b, int oh, int ow, int C, int h, int w, float* out, void* stream);

/* ---- dataloaders (midair / kitti / tartanair .py): _decode_samples after decompression ----------------------- */

/* RGB frames as the JPEG decoder leaves them, [n,ih,iw,3] uint8 -> [n,oh,ow,3] float32 =
 * tf.image.resize(cast(image)/255, [oh,ow]) (bilinear, half-pixel centres; midair.py:35-45,
 * kitti.py:25-36, tartanair.py:21-33). */
int m4d_decode_rgb8_resize(const uint8_t* images, int n, int ih, int iw, int oh, int ow, float* out,
                           void* stream);
/* Ground-truth maps [n,ih,iw] -> [n,oh,ow,1] float32.
 * kind 0  Mid-Air: uint16 bit patterns of float16 disparity, depth = 512/x, bilinear (midair.py:49-55)
 * kind 1  KITTI: uint16/256, nearest; crop (may be NULL) = {y0,y1,x0,x1} keeps [y0,y1)x[x0,x1) and
 *         zeroes the rest -- the Garg/Eigen evaluation mask (kitti.py:14-20,43-50)
 * kind 2  TartanAir: float32, nearest; rgb_resized (may be NULL) [n,oh,ow,3]: pixels whose colour
 *         is exactly black are zeroed (tartanair.py:37-45). */
int m4d_decode_depth_resize(const void* raw, int kind, int n, int ih, int iw, int oh, int ow,
                            const float* rgb_resized, const int crop[4], float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* M4DEPTH_HIP_H_ */
