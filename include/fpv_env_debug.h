/*
 * fpv_env_debug.h — diagnostic hooks of libfpv_env.so: phase stamps, launch-shape switches, calibration launches and the two
 * probes behind tests/test_graph_memset_gpu.py.  Not part of the product interface (include/fpv_env.h): nothing in the
 * reference stands above these, no facade calls them outside tools/, bench.py's phase report and the tests.
 */
#ifndef FPV_ENV_DEBUG_H
#define FPV_ENV_DEBUG_H

#include "fpv_env.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Process-wide tuning knob: hover handles with at most `max_drones` drones (default 32 768) run the env step as a
 * physics wave + helper wave pair per 64-drone tile (the helper draws the motor noise one sub-step pair ahead and
 * stores the observation tile); that pays while the helper wave finds a SIMD of its own.  Larger handles use the
 * single-wave kernel.  0 disables the pair kernel.  Results are bit-identical either way. */
int fpv_set_pair_threshold(int64_t max_drones);
/* The same for the four-tiles-per-workgroup kernel with a physics and a helper wave per tile (step_quad_kernel): handles with more
 * than min_drones and at most max_drones drones (fp32, motor noise, a multiple of 256) step through it; it takes precedence over
 * the wave-pair kernel.  (0, 0) switches it off. */
int fpv_set_quad_range(int64_t min_drones, int64_t max_drones);

/* Launches an empty kernel on `stream`: the floor of this library's launch path, for calibrating event timings. */
int fpv_noop(void* stream);

/* When device_buffer is not NULL, later fpv_rollout_step launches OF THIS HANDLE write s_memrealtime (100 MHz) phase stamps
 * [workgroup][16] into it — 0 kernel begins, 1 staging + statistics merge done, 2 policy done, 3 kernel ends; 4..7 tile 0's
 * physics wave: sub-step pairs done, observation + reward done, barrier 6, barrier 7; 8..15 pixel detector (fpv_rollstep.hip,
 * TailOut) — and fpv_rollout_steps launches [workgroup][T][8] (tools/rollout_phases.py).  NULL switches it off again.  The
 * pointer is a kernel argument: a captured graph keeps the value it was captured with. */
int fpv_debug_rollout_stamps(fpv_env* env, uint64_t* device_buffer);
/* Bound of every grid-wide wait inside this handle's fpv_rollout_steps launches in ticks of the 100 MHz counter (0 = the
 * default, 2 s).  A bound of one tick makes the first wait give up: the way the tests exercise the abandon path (every
 * workgroup leaves its loop, the status word is set, nothing hangs). */
int fpv_debug_rollout_timeout(fpv_env* env, uint64_t ticks);

/* Probes for "is a hipMemsetAsync node inside a captured HIP graph executed, in order, at every replay" (DESIGN §5):
 * fpv_debug_memset is hipMemsetAsync(dst, value, bytes, stream) as this library would issue it; fpv_debug_atomic_inc adds 1 to
 * each of n 32-bit words with an atomic — the consumer pattern of fpv_ledger_scan's counters. */
int fpv_debug_memset(void* dst, int32_t value, int64_t bytes, void* stream);
int fpv_debug_atomic_inc(uint32_t* words, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
