/*
 * jpdse_dev.h -- the extra entry points of the developer build libjpdse_hip_dev.so (compiled with -DJPDSE_DEV from
 * the same sources as libjpdse_hip.so).  Not part of the drop-in boundary: the shipped library neither exports this
 * symbols nor contains the run-time switches behind them (they are compile-time constants there) nor the timing-only
 * ablation kernels.  Used by scripts/ (same-process A/B measurements) and by the tests that compare two kernels of one
 * layer with each other; no reference counterpart.
 */
#ifndef JPDSE_DEV_H_
#define JPDSE_DEV_H_

#include "jpdse.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Developer A/B switch (kernel SELECTION only, results stay correct except for the timing-only ablation codes >= 100):
 * 0 = every convolution on the generic register-staged kernels; 1 (default) = all specialised bf16 kernels; 3 = no halo kernel;
 * 4 = no all-nine-taps weight gradient; 5 = heads without the Toeplitz GEMM; 6 = no split-K, no head kernel, no tap programs, no
 * persistent kernel (same summation order as the generic kernels: bit-comparable); 7 = reflect data gradient on the padded domain;
 * 8 = halo kernel always double buffered; 9 / 10 = long-K-only phase merging / no 128-row tiles; 12 = no all-taps weight gradient;
 * 13 = no tap-sum forward; 14 = no head kernel; 15 / 16 = XCD-aware halo tile orders; 18 = no thin-input forward kernel;
 * 19 = halo kernel on 16x16x32 MFMA fragments; 26 = merged stride-phase data gradient only from 384 tiles on; 27 = InstanceNorm
 * always as three kernels; 28 = InstanceNorm as one kernel with the in-launch exchange; 29 = the 64-channel / thin-input layers on
 * the halo / fast kernels instead of the row-streaming family; 32 = no InstanceNorm moments in conv epilogues; 35 = stride-2 data
 * gradients on the merged-phase fast kernel instead of the tap-program halo kernel (gemm_taps.h); 36 = 4x4 stride-1 layers on the
 * fast kernel alone; 38 = 3x3 layers with 32-pixel-wide grids on the split-K fast kernel instead of the nine-tap program;
 * 40 = reflect data gradient as halo kernel + four ring-strip GEMMs + ring_fold_kernel instead of the folded frame (gemm_halo.h
 * VIRT); 41 = the one-output-channel layers backward on the GEMM paths instead of thin_out1.h; 42 = 4x4 stride-2 data gradients on
 * the merged-phase fast kernel only; 43 = the same layers on the tap program + fringe (measured slower in the step);
 * 48 = fp32 generic kernel without split-K; 50 = the short-K layers on gemm_fast_kernel instead of the persistent form
 * (gemm_pers.h); 51 = every fast-kernel layer without split-K on the persistent form; 52 = the persistent form from one tile on
 * and for any K (tests); 53 / 56 = ResnetBlock / VGG forward on the four-wave / sixteen-wave halo kernels (gemm_halo4.h, gemm_halo16.h:
 * the wave-tile A/B of profiles/r04_halo_wavetile_ab.txt; results identical); 54 = dgrad2_rows with the timing-only conflict-free
 * addressing of round 4's first experiment (WRONG results); 55 = 32-pixel-wide 1024-channel weight gradients on the per-tap kernel
 * instead of the row-pair nine-tap form; 57 = the 32 -> 3 head forward on head_fwd_kernel instead of head_rows_kernel<7, 32>;
 * 58 = all-taps weight gradient (wgrad_taps.h) with the tiles of a pixel range co-located on one XCD (measured slower);
 * 61 = the short-K whole-round layers on the persistent form (gemm_pers.h), as shipped during round 4 before the epilogue fix (50 is the default again);
 * 60 = halo kernel without the XCD-aware tile order on its one-round grids (15 / 16 / 17 force an order on every grid);
 * 59 = the few-tile medium-K layers of the fast kernel on 256-row tiles as before round 4;
 * 201 / 202 / 203 = timing-only ablations of that kernel's loop (no DMA after the prologue / no fragment reads and MFMAs / neither);
 * 210 = timing-only ablation of gemm_fast_kernel: activation tiles staged for one tap in four (WRONG results; persistent form off).
 * 100 + bits = timing-only ablations of the halo loop.
 * Retired in round 4 with their negative results on record (profiles/r0*_ab.txt; the code paths are gone):
 * 21 / 22 / 24 (unpipelined loop forms of the nine-tap weight gradient), 23 (halo kernel, staggered DMA issue), 25 (halo kernel,
 * hand-pipelined fragment reads), 30 (fast kernel, XCD-aware tile order), 31 (ring strips on 128-row tiles), 44 / 45 (3- / 4-stage
 * rings for the 128-row short-K configurations), 46 (no 64-row tiles for the short loops).
 * Each call resets the others to their defaults. */
int jpdse_debug_set_fast_path(int32_t enable);

/* CU occupier for the one-GPU rehearsal of "compute kernels share the chip with a collective" (scripts/cu_contention.py,
 * profiles/r03_cu_contention.txt): launches `blocks` (1..128) workgroups on `stream`, each holding a whole CU's 160 KiB of LDS,
 * that sleep until *release_flag (host-visible memory, e.g. pinned) becomes non-zero or max_ms (<= 20000) of wall clock have
 * passed -- every wave exits by itself, the grid always drains.  No reference counterpart. */
int jpdse_debug_occupy_cus(int32_t blocks, const int32_t* release_flag, int32_t max_ms, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JPDSE_DEV_H_ */
