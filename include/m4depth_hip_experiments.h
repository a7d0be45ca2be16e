/*
 * m4depth_hip_experiments.h -- entry points that exist ONLY in a `make -C m4depth_amd/csrc EXPERIMENTS=1` build of
 * libm4depth_hip.so (m4d_build_info() then says "+experiments").  Everything here was measured, found not to be faster
 * end to end, and is NOT dispatched by the product (DESIGN_HISTORY.md); it stays buildable for the bit-identity tests and
 * the profiling tools that compare against it.  The selectors are PROCESS-WIDE state: set them from one thread, while
 * nothing is launching.
 */
#ifndef M4DEPTH_HIP_EXPERIMENTS_H_
#define M4DEPTH_HIP_EXPERIMENTS_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Which kernel serves m4d_conv3x3_wino6_bias_act (results are bit-identical): 0 (default) and 1 = the 16x16-pixel x 64-cout
 * workgroups of m4d_wino6.hip, 2 = the wide kernel of m4d_wino6w.hip (16x16 pixels x all 96 / 128 couts, two passes over the
 * Winograd position rows) wherever it applies (64 < Cout <= 128, Cout % 4 == 0), 3 = the half-tile kernel of m4d_wino6h.hip
 * (16x8 pixels x 64 couts per workgroup) everywhere. */
void m4d_wino6_set_variant(int variant);
/* Under variant 0 the half-tile kernel serves the launches whose m4d_wino6.hip grid would have at most `max_wg` workgroups
 * (default 0 = none: faster alone on small grids, no gain inside the frame pipeline). */
void m4d_wino6_set_half_tile_max_workgroups(int max_wg);

/* ---- launch tape (csrc/m4d_tape.hip): a recorded sequence of this library's kernel launches, replayed as PLAIN STREAM
 * LAUNCHES from one host loop -- the host-side cost of a hipGraph replay without its effect on concurrent small kernels
 * (DESIGN_HISTORY.md).  Between m4d_tape_begin()
 * and m4d_tape_end() every entry point called BY THE SAME THREAD records its launches (function, grid, block, LDS, a copy of
 * every argument) instead of executing them; the recording pass must run on the buffers the replays will use.
 * m4d_tape_begin returns the tape id (-1: already recording), m4d_tape_end the number of launches recorded,
 * m4d_tape_replay issues them on `stream` in order (0 or a hipError_t). */
int m4d_tape_begin(void);
int m4d_tape_end(void);
int m4d_tape_length(int tape);
int m4d_tape_replay(int tape, void* stream);
int m4d_tape_free(int tape);
/* Entry points that enqueue a hipMemsetAsync beside their kernels (m4d_backproject_bwd, m4d_dscv_bwd, ...) return
 * hipErrorNotSupported while the calling thread records: a memset is not a kernel launch and would be lost from the replay. */

#ifdef __cplusplus
}
#endif
#endif /* M4DEPTH_HIP_EXPERIMENTS_H_ */
