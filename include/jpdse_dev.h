/*
 * jpdse_dev.h -- the ONE extra entry point of the developer build libjpdse_hip_dev.so (compiled with -DJPDSE_DEV from
 * the same sources as libjpdse_hip.so).  Not part of the drop-in boundary: the shipped library neither exports this
 * symbol nor contains the run-time switches behind it (they are compile-time constants there) nor the timing-only
 * ablation kernels.  Used by scripts/ (same-process A/B measurements) and by the tests that compare two kernels of one
 * layer with each other; no reference counterpart.
 */
#ifndef JPDSE_DEV_H_
#define JPDSE_DEV_H_

#include "jpdse.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Developer A/B switch (kernel SELECTION only, results stay correct except for
 * the timing-only ablation codes >= 100): 0 = every convolution on the generic register-staged
 * kernels; 1 (default) = all specialised bf16 kernels; 3 = no halo kernel; 4 = no per-filter-row
 * weight gradient; 5 = heads without the Toeplitz GEMM; 6 = no split-K and no head kernel (same
 * summation order as the generic kernels: bit-comparable); 7 = reflect data gradient on the padded
 * domain; 8 = halo kernel always double buffered; 9 / 10 = long-K-only phase merging / no 128-row
 * tiles; 12 = no all-taps weight gradient; 13 = no tap-sum forward; 14 = no head kernel;
 * 15 / 16 = XCD-aware halo tile orders; 18 = no thin-input forward kernel; 19 = halo kernel on
 * 16x16x32 MFMA fragments.  Each call resets the others to their defaults. */
int jpdse_debug_set_fast_path(int32_t enable);

#ifdef __cplusplus
}
#endif
#endif /* JPDSE_DEV_H_ */
