// Implicit-GEMM convolution family for gfx950 (MI355X): forward, data-gradient and
// weight-gradient of every nn.Conv2d / nn.ConvTranspose2d on the JPD-SE hot path
// (reference call sites: include/jpdse.h "convolution family").
//
// Formulation ("row-run"): the input is materially padded to NHWC [N,Hp,Wp,Cs]; for an
// output pixel m=(n,oh,ow) and filter row r the S*Cs input values under that filter row are
// ONE contiguous run in memory (consecutive s are consecutive pixels).  So
//     A[m][(r,j)] = X[rowbase(m) + r*in_sr + j],   j in [0, Lk),  Lk = roundup(S*Cs, BKE)
// and the filter is packed as B[k][(r,j)] with zeros where j >= S*Cs or c >= C.  One kernel
// then serves every (R,S,stride,pad mode) and, through the output addressing
// (out_base/out_sh/out_sw), the stride-2 sub-pixel phases of dgrad / ConvTranspose2d.
//
// MFMA: bf16 -> v_mfma_f32_32x32x16_bf16, fp32 -> v_mfma_f32_32x32x2_f32 (exact fp32 fma
// chain).  LDS tiles are [rows][64 B] with the 16-byte slot index XOR-swizzled by
// (row>>3)&3 so that both the 16-B staging writes and the ds_read_b128 fragment reads are
// bank-conflict free (MI355X_MICROARCH.md, LDS: b128 reads are served in 16-lane groups
// over 64 banks).
#include "common.h"

#include <vector>

// Kernel-selection switches (which of two correct kernels runs) and timing-only ablations are DEVELOPER tools: in the
// shipped library (libjpdse_hip.so) they are compile-time constants -- no mutable global state behind the ABI besides the
// init-once launch attributes and the opt-in kernel timer -- and the ablation kernels are not even instantiated.  The
// developer build (libjpdse_hip_dev.so, -DJPDSE_DEV, include/jpdse_dev.h) makes them run-time variables behind
// jpdse_debug_set_fast_path for same-process A/B measurements and for the tests that compare two kernels of one layer.
// (JPDSE_SWITCH itself lives in common.h.)

namespace jpdse {

// =========================================================================================
// kernel arguments
// =========================================================================================
struct GemmFwdArgs {
  const void* A;
  const void* B;
  const float* bias;
  void* Y;
  int M, OH, OW;
  int Kout, Ks;
  int R, cpr;  // filter rows, 64-byte chunks per filter row
  int b_rows;
  long long b_row_stride;  // elements
  long long in_sn, in_sh, in_sw, in_sr, in_base;
  long long out_sn, out_sh, out_sw, out_base;
  int act;
  float slope;
  int col_mod, k_real;   // col_mod > 0: GEMM column -> channel (col % col_mod), live when < k_real (Toeplitz head)
  // split-K (round 4; small M: fewer tiles than CUs -- the fp32 layers of BASELINE config 2 at 512x256 batch 1): block
  // (tile, blockIdx.y = split) reduces K-chunks [T*split/splits, T*(split+1)/splits) and stores its fp32 partial tile into slab
  // `split` of `partial` ([splits][M][Ks]); gemm_splitk_finish_kernel adds the slabs in index order (deterministic) and applies
  // bias + activation.  splits is chosen by the launcher (generic_splitk_for); partial_cap = bytes available behind `partial`.
  int splits;
  float* partial;
  size_t partial_cap;
};

struct GemmWgradArgs {
  const void* X;
  const void* DY;
  float* DW;
  int M, OH, OW;
  int K, Ks;  // dy logical / storage channels
  int C, Cs;  // x  logical / storage channels
  int R, S;
  int run;    // S*Cs
  int col_tiles_per_r;
  long long in_sn, in_sh, in_sw, in_sr, in_base;
  long long dy_sn, dy_sh, dy_sw, dy_base;
  int chunks_total;
  int chunks_per_split;
  float* partial;            // splits > 1: split y stores its partial gradient into slab y (fp32, DW's layout);
  long long slab_stride;     // slab_reduce_kernel adds the slabs in a fixed order (no atomics: reproducible)
};

__device__ __forceinline__ int swz(int row, int slot) { return (row << 6) + (((slot ^ (row >> 3)) & 3) << 4); }

// activation as a compile-time parameter: epilogues dispatch the run-time `act` ONCE around their value loops (gemm_fast.h, conv_generic.h)
template <int ACT> __device__ __forceinline__ float act_ct(float v, float slope) {
  if constexpr (ACT == JPDSE_ACT_RELU) return v > 0.f ? v : 0.f;
  else if constexpr (ACT == JPDSE_ACT_LRELU) return v > 0.f ? v : v * slope;
  else if constexpr (ACT == JPDSE_ACT_TANH) return tanhf(v);
  else return v;
}
__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == JPDSE_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == JPDSE_ACT_LRELU) return v > 0.f ? v : v * slope;
  if (act == JPDSE_ACT_TANH) return tanhf(v);
  return v;
}

}  // namespace jpdse
#include "gemm_fast.h"
#include "gemm_halo.h"
#ifdef JPDSE_DEV
#include "gemm_halo4.h"
#include "gemm_halo16.h"
#endif
#include "gemm_taps.h"
#include "wgrad_fast.h"
#include "wgrad_thin.h"
#include "wgrad_nine.h"
#include "wgrad_taps.h"
#include "head_fwd.h"
#include "thin_fwd.h"
#include "conv_rows.h"
#include "gemm_pers.h"
#include "dgrad2_rows.h"
#include "head_rows.h"
#include "thin_dgrad2_rows.h"
#include "thin_rows.h"
#include "thin_in_rows.h"
#include "conv_generic.h"
#include "conv_plan.h"
#include "conv_launch.h"
#include "thin_out1.h"
#include "conv_dispatch_fwd.h"
#include "conv_dispatch_dgrad.h"
#include "conv_dispatch_wgrad.h"


using namespace jpdse;

extern "C" {

int jpdse_conv_out_shape(const jpdse_conv_desc* d, int32_t* OH, int32_t* OW) {
  if (int rc = validate(d)) return rc;
  JPDSE_REQUIRE(OH && OW, "conv_out_shape: null output");
  ConvPlan p;
  make_plan(d, &p);
  *OH = p.OH;
  *OW = p.OW;
  return JPDSE_OK;
}

#ifdef JPDSE_DEV
// CU occupier (developer build only; scripts/cu_contention.py): `blocks` workgroups that each hold a whole CU's LDS (no
// LDS-using workgroup can share the CU) and sleep until *release != 0 or `max_ms` of wall clock have passed -- a stand-in for
// the compute units a concurrent RCCL collective holds.  Every wave reaches its exit: the poll is bounded by s_memrealtime
// (100 MHz), so the grid drains even if the host never sets the flag.
__global__ __launch_bounds__(256) void occupy_cus_kernel(const int* __restrict__ release, unsigned long long max_ticks) {
  extern __shared__ __attribute__((aligned(16))) char occ_smem[];
  if (threadIdx.x == 0) reinterpret_cast<volatile int*>(occ_smem)[0] = (int)blockIdx.x;   // the allocation is live
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (true) {
    const int r = __hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (r != 0) break;
    if (__builtin_amdgcn_s_memrealtime() - t0 > max_ticks) break;
    __builtin_amdgcn_s_sleep(127);
  }
}

int jpdse_debug_occupy_cus(int32_t blocks, const int32_t* release_flag, int32_t max_ms, void* stream) {
  JPDSE_REQUIRE(blocks > 0 && blocks <= 128 && release_flag != nullptr && max_ms > 0 && max_ms <= 20000,
                "debug_occupy_cus: blocks in 1..128, a flag, max_ms in 1..20000");
  constexpr int lds = 160 * 1024;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_cus_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "occupy_cus: hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL(occupy_cus_kernel, dim3(blocks), dim3(256), lds, as_stream(stream), release_flag,
                     (unsigned long long)max_ms * 100000ULL);
  return check_launch("occupy_cus_kernel");
}

int jpdse_debug_set_fast_path(int32_t enable) {
  // 0: generic kernels only; 1: fast kernels (default schedule 0); 2: fast kernels, alternative schedule 1
  g_fast_enabled = enable != 0;
  g_halo_enabled = enable != 3;      // 3: fast kernels but no halo kernel (A/B)
  g_moments_fused = enable != 32 && enable != 6;   // 32: InstanceNorm moments always in their own pass (A/B); 6 keeps the generic kernels' rounding points
  g_ring_virt = enable != 40 && enable != 6;   // 40: reflect ring as four split-K strip GEMMs + ring_fold_kernel instead of the folded frame (A/B); 6 keeps every product in fp32 until the fold
  g_thin1_enabled = enable != 41 && enable != 6;   // 41: the one-output-channel layers (PatchGAN 512 -> 1) backward on the GEMM paths (A/B)
  g_taps_dgrad4_enabled = enable != 42 && enable != 6;   // 42: 4x4 stride-2 data gradients (PatchGAN layers 1-2) on the merged-phase fast kernel (A/B)
  g_taps_dgrad4_min_tiles = enable == 43 ? 1 : (1 << 30);     // 43: 4x4 stride-2 data gradients on the tap program + fringe (developer build only: slower in the step)
  g_taps9_enabled = enable != 38 && enable != 6;   // 38: 3x3 layers with 32-pixel-wide grids on the split-K fast kernel (A/B)
  g_taps4_enabled = enable != 36 && enable != 6;   // 36: 4x4 stride-1 layers on the fast kernel alone (A/B)
  g_taps_enabled = enable != 35 && enable != 6;   // 35: stride-2 data gradients on the merged-phase fast kernel instead of the tap-program halo kernel (A/B); 6 keeps the generic kernels' summation order (tap outer, slab inner)
  g_rows_enabled = enable != 29 && enable != 3;   // 29: 64-channel 3x3 layers on the halo / fast kernels instead of conv_rows (A/B)
  g_halo_abl = (enable >= 100 && enable < 200) ? enable - 100 : 0;
  g_wgrad_nine_enabled = enable != 4;    // 4: wide 3x3 layers on the per-tap fast weight gradient instead of the all-nine-taps one (A/B)
  g_wgrad_taps_enabled = enable != 12;   // 12: fast kernels without the all-taps weight gradient (A/B)
  g_ring_enabled = enable != 7 && enable != 3;   // 7: reflect data gradient on the padded domain + fold (A/B)
  g_merge_min_kt = enable == 9 ? 16 : 4;
  g_norm_fused = enable == 27 ? 0 : (enable == 28 ? 2 : 1);   // 27: InstanceNorm always as three kernels; 28: one-kernel form with the in-launch exchange (A/B)
  g_merge_min_tiles = enable == 26 ? 384 : 64;    // 9: merged stride-phase data gradient only for long K loops (A/B)
  g_fast_small = enable == 10 ? 0 : 20;     // 10: no 128-row / 2-stage configs for short K loops (A/B)   // 9: merged stride-phase data gradient also for short K loops (A/B)
  g_halo_xcd = enable == 15 ? 1 : (enable == 16 ? 2 : (enable == 17 ? 3 : 0));   // 17: 4 N-tiles x 8 patches per XCD (one-round grids)   // 15 / 16: XCD-aware tile orders of the halo kernel (A/B)
  g_halo_mf16 = enable == 19;         // 19: halo kernel on 16x16x32 MFMAs (A/B)
  g_halo_single = enable != 8;        // 8: halo kernel always with two patch buffers (A/B)
  g_head_fwd_enabled = enable != 14 && enable != 6;   // 14: heads on the Toeplitz GEMM (A/B); 6 keeps the generic order
  g_thin_fwd_enabled = enable != 18;  // 18: thin-input forward on the generic kernel (A/B)
  g_tapsum_enabled = enable != 13;    // 13: narrow-output layers without the tap-sum forward (A/B)
  g_thin_out_fast = enable != 13;     // 13: narrow-output long-K layers on the generic kernel (A/B)
  g_toep_enabled = enable != 5;       // 5: fast kernels, plain head forward
  g_pers_enabled = enable == 51 || enable == 52 || enable == 61;   // 61: the persistent form on its whole-round grids (the default of round 4 until the epilogue fix made gemm_fast_kernel faster); 50 = 1 now
  g_pers_max_kt = (enable == 51 || enable == 52) ? (1 << 20) : 24;
  g_pers_min_tiles = enable == 52 ? 1 : 256;      // 52: the persistent form from one tile on and for any K (tests)  // 51: every fast-kernel layer without split-K on the persistent form (A/B)
  g_halo4 = enable == 53 ? 1 : (enable == 56 ? 16 : 0);   // 56: ... on the sixteen-wave / 64 x 32 wave-tile form (A/B, gemm_halo16.h)
  //             // 53: plain halo forward on the four-wave / 128 x 64 wave-tile form (A/B, gemm_halo4.h)
  g_wgrad_nine32_enabled = enable != 55;     // 55: the 1024-channel trunk at 16 x 32 on the per-tap weight-gradient kernel (A/B)
  g_head_rows32 = enable != 57;       // 57: the 32 -> 3 head forward on head_fwd_kernel (A/B)
  g_wgrad_taps_abl = (enable >= 200 && enable < 204) ? enable - 200 : 0;   // 201 / 202 / 203: all-taps weight gradient without DMA / without MFMAs / neither (timing only)
  g_fast_abl = (enable >= 210 && enable < 220) ? (enable == 210 ? 16 : enable - 210) : (enable == 220 ? 32 : (enable == 221 ? 64 : (enable == 222 ? 96 : (enable == 223 ? 39 : (enable == 224 ? 192 : (enable == 225 ? 128 : 0))))));   // 224: no K loop, epilogue without its global stores; 225: whole kernel without the global stores of the epilogue   // 220 no epilogue, 221 no K loop, 222 neither (launch + set-up only), 223 MFMA-only loop without epilogue;   // 210-219: timing-only ablations of the 256 x 128 fast configuration (wrong results): 210 = activation tile staged for one tap in four; 210 + bits: 1 no DMA, 2 no barrier, 4 fragments of k-step 0 only, 8 no MFMAs
  g_halo_xcd_auto = enable != 60;     // 60: halo kernel, block b -> tile b on every grid (no XCD-aware order on the one-round grids, A/B)
  g_fast_fill = enable != 59;         // 59: few-tile medium-K layers on the 256-row tiles as before round 4 (A/B)
  g_wgrad_taps_xcd = enable == 58;    // 58: all-taps weight gradient with the tiles of a pixel range co-located on one XCD (A/B: slower)
  g_dgrad2_noconf = enable == 54;     // 54: dgrad2_rows_kernel with conflict-free LDS addresses (timing only, wrong results)
  g_generic_splitk = enable != 48;    // 48: fp32 generic kernel without split-K (A/B; BASELINE config 2)
  g_splitk_enabled = enable != 6;     // 6: fast kernels, no split-K  // 4: fast kernels but the per-tap weight-gradient kernel (A/B)
  return JPDSE_OK;
}
#endif

int jpdse_prof_select(int32_t enable, int32_t Ks, int64_t kdim, int32_t max_launches) {
  g_prof.on = false;
  g_prof.used = 0;
  if (!enable) return JPDSE_OK;
  JPDSE_REQUIRE(Ks > 0 && kdim > 0 && max_launches > 0, "prof_select: bad selection");
  while ((int)g_prof.ev.size() < 2 * max_launches) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return set_error(JPDSE_ELAUNCH, "prof_select: hipEventCreate failed");
    g_prof.ev.push_back(e);
  }
  g_prof.flops.assign(max_launches, 0.0);
  g_prof.cls.assign(max_launches, 0);
  g_prof.Ks = Ks;
  g_prof.kdim = kdim;
  g_prof.on = true;
  return JPDSE_OK;
}

static int prof_sum(int cls, double* total_ms, double* total_flops, int64_t* launches) {
  double ms = 0.0, fl = 0.0;
  int64_t n = 0;
  for (int i = 0; i < g_prof.used; ++i) {
    if (g_prof.cls[i] != cls) continue;
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess)
      return set_error(JPDSE_ELAUNCH, "prof_collect: hipEventSynchronize failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess)
      return set_error(JPDSE_ELAUNCH, "prof_collect: hipEventElapsedTime failed");
    ms += t;
    fl += g_prof.flops[i];
    ++n;
  }
  *total_ms = ms;
  *total_flops = fl;
  *launches = n;
  return JPDSE_OK;
}

int jpdse_prof_collect_class(int32_t cls, double* total_ms, double* total_flops, int64_t* launches) {
  JPDSE_REQUIRE(total_ms && total_flops && launches && cls >= 0 && cls <= 2, "prof_collect_class: bad argument");
  return prof_sum(cls, total_ms, total_flops, launches);
}

int jpdse_prof_collect(double* total_ms, double* total_flops, int64_t* launches) {
  JPDSE_REQUIRE(total_ms && total_flops && launches, "prof_collect: null output");
  const int rc = prof_sum(0, total_ms, total_flops, launches);
  g_prof.used = 0;
  return rc;
}

int jpdse_conv_plan_query(const jpdse_conv_desc* d, int32_t* out, int32_t n) {
  if (int rc = validate(d)) return rc;
  JPDSE_REQUIRE(out != nullptr && n >= 14 + 4 * 10, "conv_plan_query: need room for 54 int32");
  ConvPlan p;
  make_plan(d, &p);
  int32_t head[14] = {p.Cs, p.Ks, p.Hp, p.Wp, p.OH, p.OW, p.Lk_fwd, p.nph, p.PT, p.PB, p.PL, p.PR, p.DH, p.DW};
  for (int i = 0; i < 14; ++i) out[i] = head[i];
  for (int i = 0; i < 4; ++i) {
    int32_t* o = out + 14 + i * 10;
    if (i < p.nph) {
      const Phase& f = p.ph[i];
      int32_t v[10] = {f.qh, f.qw, f.Uh, f.Uw, f.i0h, f.cnth, f.i0w, f.cntw, f.Lk, (int32_t)f.pack_off};
      for (int j = 0; j < 10; ++j) o[j] = v[j];
    } else {
      for (int j = 0; j < 10; ++j) o[j] = 0;
    }
  }
  return JPDSE_OK;
}

size_t jpdse_conv_fwd_pack_size(const jpdse_conv_desc* d) {
  if (validate(d)) return 0;
  ConvPlan p;
  make_plan(d, &p);
  return p.fwd_pack_bytes;
}

size_t jpdse_conv_dgrad_pack_size(const jpdse_conv_desc* d) {
  if (validate(d)) return 0;
  ConvPlan p;
  make_plan(d, &p);
  return p.dgrad_pack_bytes;
}

size_t jpdse_conv_workspace_size(const jpdse_conv_desc* d) {
  if (validate(d)) return 0;
  ConvPlan p;
  make_plan(d, &p);
  // wgrad: padded x (+ tap-expanded dy for few-output-channel layers) + the slabs of the split reduction
  size_t slab_bytes = 0;
  if (d->dtype == JPDSE_BF16) (void)conv_wgrad_t<bf16_t>(d, p, nullptr, nullptr, nullptr, nullptr, nullptr, &slab_bytes);
  else (void)conv_wgrad_t<float>(d, p, nullptr, nullptr, nullptr, nullptr, nullptr, &slab_bytes);
  const size_t fwd = wgrad_front_bytes(d, p) + slab_bytes;
  size_t dgrad = p.dypad_bytes + p.dxp_bytes;
  if (thin1_shape_ok(d, p.Cs, p.Ks) && dgrad < (size_t)16 * p.Cs * sizeof(float)) dgrad = (size_t)16 * p.Cs * sizeof(float);   // tap table of thin1_dgrad_kernel
  size_t sk = p.splitk_off + p.splitk_bytes;
  if (p.ES == 2 && d->pad_mode == JPDSE_PAD_REFLECT && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1) {
    const size_t ring = (size_t)8 * d->N * (2 * (d->W + 2) + 2 * d->H) * p.Cs * 4;   // ring-strip slabs, <= 8 splits
    const size_t frame = (size_t)d->N * (2 * (d->W + 2) + 2 * d->H) * p.Ks * 2;       // folded frame of dy (bf16)
    sk = sk > ring ? sk : ring;
    sk = sk > frame ? sk : frame;
  }
  if (p.ES == 2 && d->R == 4 && d->S == 4 && d->stride == 1) {          // fringe slabs of the 4x4 tap-program path (forward / data gradient)
    const size_t f4 = taps4_fringe_bytes(d->N, p.OH, p.OW, p.Ks, 16 * p.Cs / 64);
    const size_t d4 = taps4_fringe_bytes(d->N, d->H, d->W, p.Cs, 16 * p.Ks / 64);
    sk = sk > f4 ? sk : f4;
    sk = sk > d4 ? sk : d4;
  }
  if (p.ES == 2 && d->R == 3 && d->S == 3 && d->stride == 1) {          // split-K slabs of the nine-tap program (8 x 32 grids), forward / data gradient
    const size_t f9 = (size_t)taps9_splits(d->N, p.OH, p.OW, p.Cs, p.Ks) * d->N * p.OH * p.OW * p.Ks * 4;
    const size_t d9 = (size_t)taps9_splits(d->N, d->H, d->W, p.Ks, p.Cs) * d->N * d->H * d->W * p.Cs * 4;
    const size_t t9 = p.splitk_off + (f9 > d9 ? f9 : d9);
    sk = sk > t9 ? sk : t9;
  }
  size_t m = fwd > dgrad ? fwd : dgrad;
  if (p.ES == 2 && d->stride == 1 && d->K * d->R * d->S <= 32) {      // Z of the tap-sum forward (tapsum_kernel)
    const size_t zb = (size_t)d->N * d->H * d->W * 32 * sizeof(float);
    m = m > zb ? m : zb;
  }
  const size_t taps = wgrad_taps_ws_bytes(d, p);
  m = m > taps ? m : taps;
  const size_t nine = wgrad_nine_ws_bytes(d, p);
  m = m > nine ? m : nine;
  return m > sk ? m : sk;
}

int jpdse_conv_pack_entries(const jpdse_conv_desc* d, const float* w, void* dgrad_pack, jpdse_pack_entry* out,
                            int32_t max_entries) {
  if (validate(d)) return -1;
  if (w == nullptr || dgrad_pack == nullptr || out == nullptr) return -1;
  ConvPlan p;
  make_plan(d, &p);
  int n = 0;
  for (int i = 0; i < p.nph; ++i) {
    const Phase& f = p.ph[i];
    if ((long long)p.Cs * f.Uh * f.Lk == 0) continue;
    if (!(f.Lk == f.Uw * p.Ks && p.Cs * p.Ks >= 64 * 64)) return -1;   // a phase that needs the padded-K kernel
    if (n >= max_entries) return -1;
    jpdse_pack_entry e = {};
    e.w = w;
    e.out = reinterpret_cast<char*>(dgrad_pack) + f.pack_off;
    e.K = d->K; e.Ks = p.Ks; e.C = d->C; e.Cs = p.Cs; e.R = d->R; e.S = d->S; e.st = d->stride;
    e.qh = f.qh; e.qw = f.qw; e.Uh = f.Uh; e.Uw = f.Uw; e.Lk = f.Lk;
    e.gx = (p.Cs + 63) / 64; e.gy = (p.Ks + 63) / 64;
    e.out_f32 = d->dtype == JPDSE_F32 ? 1 : 0;
    e.blocks = e.gx * e.gy * f.Uh * f.Uw;
    e.block0 = 0;            // filled by the caller (prefix sum over its whole table)
    out[n++] = e;
  }
  return n;
}

int jpdse_conv_pack_run(const jpdse_pack_entry* table_dev, int32_t n_entries, int64_t total_blocks, void* stream) {
  JPDSE_REQUIRE(table_dev != nullptr && n_entries > 0 && total_blocks > 0 && total_blocks < (1LL << 31),
                "conv_pack_run: bad table");
  hipLaunchKernelGGL(pack_dgrad_tile_many_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), table_dev,
                     n_entries);
  return check_launch("pack_dgrad_tile_many_kernel");
}

int jpdse_conv_pack_weights(const jpdse_conv_desc* d, const float* w, void* fwd_pack, void* dgrad_pack,
                            void* stream) {
  if (int rc = validate(d)) return rc;
  JPDSE_REQUIRE(w != nullptr, "conv_pack_weights: null master weights");
  ConvPlan p;
  make_plan(d, &p);
  hipStream_t s = as_stream(stream);
  const bool plain = (d->C == p.Cs) && (d->K == p.Ks) && (p.Lk_fwd == d->S * p.Cs);
  if (fwd_pack && plain) {
    const long long total8 = (long long)p.Ks * d->R * p.Lk_fwd / 8;
    if (d->dtype == JPDSE_BF16)
      hipLaunchKernelGGL((pack_fwd_cast_kernel<bf16_t>), dim3(ew_blocks(total8)), dim3(256), 0, s, w,
                         reinterpret_cast<bf16_t*>(fwd_pack), total8);
    else
      hipLaunchKernelGGL((pack_fwd_cast_kernel<float>), dim3(ew_blocks(total8)), dim3(256), 0, s, w,
                         reinterpret_cast<float*>(fwd_pack), total8);
    if (int rc = check_launch("pack_fwd_cast_kernel")) return rc;
  } else if (fwd_pack) {
    const long long total = (long long)p.Ks * d->R * p.Lk_fwd;
    if (d->dtype == JPDSE_BF16)
      hipLaunchKernelGGL((pack_fwd_kernel<bf16_t>), dim3(ew_blocks(total)), dim3(256), 0, s, w,
                         reinterpret_cast<bf16_t*>(fwd_pack), d->K, p.Ks, d->C, p.Cs, d->R, d->S, p.Lk_fwd, total);
    else
      hipLaunchKernelGGL((pack_fwd_kernel<float>), dim3(ew_blocks(total)), dim3(256), 0, s, w,
                         reinterpret_cast<float*>(fwd_pack), d->K, p.Ks, d->C, p.Cs, d->R, d->S, p.Lk_fwd, total);
    if (int rc = check_launch("pack_fwd_kernel")) return rc;
  }
  if (fwd_pack && p.thinf) {
    const long long total = (long long)d->R * p.Ks * p.KP_thin;
    hipLaunchKernelGGL(pack_thin_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, w,
                       reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(fwd_pack) + p.thin_pack_off), d->K, p.Ks, d->C,
                       p.Cs, d->R, d->S, p.KP_thin, total);
    if (int rc = check_launch("pack_thin_fwd_kernel")) return rc;
  }
  if (fwd_pack && p.toep) {
    const long long total = (long long)32 * d->R * p.Lk_toep;
    hipLaunchKernelGGL((pack_fwd_toep_kernel<bf16_t>), dim3(ew_blocks(total)), dim3(256), 0, s, w,
                       reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(fwd_pack) + p.fwd_pack_plain_bytes), d->K,
                       d->C, p.Cs, d->R, d->S, p.Lk_toep, total);
    if (int rc = check_launch("pack_fwd_toep_kernel")) return rc;
  }
  if (dgrad_pack) {
    PackPhaseTable tab = {};
    const long long esz = (long long)esize(d->dtype);
    for (int i = 0; i < p.nph; ++i) {
      const Phase& f = p.ph[i];
      const long long total = (long long)p.Cs * f.Uh * f.Lk;
      if (total == 0) continue;
      char* out = reinterpret_cast<char*>(dgrad_pack) + f.pack_off;
      if (f.Lk == f.Uw * p.Ks && p.Cs * p.Ks >= 64 * 64) {     // no K padding inside the panel: tiled transpose
        const dim3 grid((p.Cs + 63) / 64, (p.Ks + 63) / 64, f.Uh * f.Uw);
        if (d->dtype == JPDSE_BF16)
          hipLaunchKernelGGL((pack_dgrad_tile_kernel<bf16_t>), grid, dim3(256), 0, s, w, reinterpret_cast<bf16_t*>(out),
                             d->K, p.Ks, d->C, p.Cs, d->R, d->S, d->stride, f.qh, f.qw, f.Uh, f.Uw, f.Lk);
        else
          hipLaunchKernelGGL((pack_dgrad_tile_kernel<float>), grid, dim3(256), 0, s, w, reinterpret_cast<float*>(out),
                             d->K, p.Ks, d->C, p.Cs, d->R, d->S, d->stride, f.qh, f.qw, f.Uh, f.Uw, f.Lk);
        if (int rc = check_launch("pack_dgrad_tile_kernel")) return rc;
        continue;
      }
      // the other phases of the layer go into one launch below
      const int q = tab.n++;
      tab.first[q + 1] = tab.first[q] + total;
      tab.out_off[q] = (long long)f.pack_off / esz;
      tab.qh[q] = f.qh; tab.qw[q] = f.qw; tab.Uh[q] = f.Uh; tab.Uw[q] = f.Uw; tab.Lk[q] = f.Lk;
    }
    if (tab.n > 0) {
      const long long total = tab.first[tab.n];
      if (d->dtype == JPDSE_BF16)
        hipLaunchKernelGGL((pack_dgrad_phases_kernel<bf16_t>), dim3(ew_blocks(total)), dim3(256), 0, s, w,
                           reinterpret_cast<bf16_t*>(dgrad_pack), d->K, p.Ks, d->C, p.Cs, d->R, d->S, d->stride, tab);
      else
        hipLaunchKernelGGL((pack_dgrad_phases_kernel<float>), dim3(ew_blocks(total)), dim3(256), 0, s, w,
                           reinterpret_cast<float*>(dgrad_pack), d->K, p.Ks, d->C, p.Cs, d->R, d->S, d->stride, tab);
      if (int rc = check_launch("pack_dgrad_phases_kernel")) return rc;
    }
  }
  return JPDSE_OK;
}

int jpdse_conv_fwd(const jpdse_conv_desc* d, const void* x, const void* fwd_pack, const float* bias, void* y,
                   void* ws, size_t ws_bytes, void* stream) {
  if (int rc = validate(d)) return rc;
  JPDSE_REQUIRE(x && fwd_pack && y, "conv_fwd: null pointer");
  ConvPlan p;
  make_plan(d, &p);
  const size_t need = jpdse_conv_workspace_size(d);
  if (ws == nullptr || ws_bytes < need)
    return set_error(JPDSE_EWORKSPACE, "conv_fwd: workspace %zu < %zu", ws_bytes, need);
  return d->dtype == JPDSE_BF16 ? conv_fwd_t<bf16_t>(d, p, x, fwd_pack, bias, y, ws, as_stream(stream))
                                : conv_fwd_t<float>(d, p, x, fwd_pack, bias, y, ws, as_stream(stream));
}

int jpdse_conv_fwd_pool(const jpdse_conv_desc* d, const void* x, const void* fwd_pack, const float* bias, void* y,
                        void* y_pool, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = validate(d)) return rc;
  JPDSE_REQUIRE(x && fwd_pack && y && y_pool, "conv_fwd_pool: null pointer");
  ConvPlan p;
  make_plan(d, &p);
  JPDSE_REQUIRE(p.OH >= 2 && p.OW >= 2, "conv_fwd_pool: output of %d x %d pixels", p.OH, p.OW);
  const size_t need = jpdse_conv_workspace_size(d);
  if (ws == nullptr || ws_bytes < need)
    return set_error(JPDSE_EWORKSPACE, "conv_fwd_pool: workspace %zu < %zu", ws_bytes, need);
  bool pooled = false;
  const int rc = d->dtype == JPDSE_BF16
                     ? conv_fwd_t<bf16_t>(d, p, x, fwd_pack, bias, y, ws, as_stream(stream), nullptr, y_pool, &pooled)
                     : conv_fwd_t<float>(d, p, x, fwd_pack, bias, y, ws, as_stream(stream), nullptr, y_pool, &pooled);
  if (rc != JPDSE_OK || pooled) return rc;
  return jpdse_maxpool2_fwd(d->dtype, d->N, p.OH, p.OW, d->K, y, y_pool, stream);     // kernels without the pooled epilogue
}

int32_t jpdse_conv_moment_slots(const jpdse_conv_desc* d) {
  if (validate(d)) return 0;
  ConvPlan p;
  make_plan(d, &p);
  return conv_fwd_moment_slots(d, p);
}

int32_t jpdse_convT_moment_slots(const jpdse_conv_desc* d) {
  if (validate(d)) return 0;
  ConvPlan p;
  make_plan(d, &p);
  return convT_fwd_moment_slots(d, p);
}

int jpdse_conv_fwd_moments(const jpdse_conv_desc* d, const void* x, const void* fwd_pack, void* y, float* moments, void* ws,
                           size_t ws_bytes, void* stream) {
  if (int rc = validate(d)) return rc;
  JPDSE_REQUIRE(x && fwd_pack && y && moments, "conv_fwd_moments: null pointer");
  ConvPlan p;
  make_plan(d, &p);
  JPDSE_REQUIRE(conv_fwd_moment_slots(d, p) > 0, "conv_fwd_moments: this layer has no moment epilogue (jpdse_conv_moment_slots == 0)");
  const size_t need = jpdse_conv_workspace_size(d);
  if (ws == nullptr || ws_bytes < need)
    return set_error(JPDSE_EWORKSPACE, "conv_fwd_moments: workspace %zu < %zu", ws_bytes, need);
  return conv_fwd_t<bf16_t>(d, p, x, fwd_pack, nullptr, y, ws, as_stream(stream), moments);
}

int jpdse_convT_fwd_moments(const jpdse_conv_desc* d, const void* x, const void* dgrad_pack, void* y, float* moments, void* ws,
                            size_t ws_bytes, void* stream) {
  if (int rc = validate(d)) return rc;
  JPDSE_REQUIRE(x && dgrad_pack && y && moments, "convT_fwd_moments: null pointer");
  ConvPlan p;
  make_plan(d, &p);
  JPDSE_REQUIRE(convT_fwd_moment_slots(d, p) > 0, "convT_fwd_moments: this layer has no moment epilogue (jpdse_convT_moment_slots == 0)");
  const size_t need = jpdse_conv_workspace_size(d);
  if (ws == nullptr || ws_bytes < need)
    return set_error(JPDSE_EWORKSPACE, "convT_fwd_moments: workspace %zu < %zu", ws_bytes, need);
  return conv_dgrad_t<bf16_t>(d, p, x, dgrad_pack, y, ws, as_stream(stream), nullptr, nullptr, moments);
}

int jpdse_conv_dgrad(const jpdse_conv_desc* d, const void* dy, const void* dgrad_pack, void* dx, void* ws,
                     size_t ws_bytes, void* stream) {
  if (int rc = validate(d)) return rc;
  JPDSE_REQUIRE(dy && dgrad_pack && dx, "conv_dgrad: null pointer");
  ConvPlan p;
  make_plan(d, &p);
  const size_t need = jpdse_conv_workspace_size(d);
  if (ws == nullptr || ws_bytes < need) return set_error(JPDSE_EWORKSPACE, "conv_dgrad: workspace %zu < %zu", ws_bytes, need);
  return d->dtype == JPDSE_BF16 ? conv_dgrad_t<bf16_t>(d, p, dy, dgrad_pack, dx, ws, as_stream(stream))
                                : conv_dgrad_t<float>(d, p, dy, dgrad_pack, dx, ws, as_stream(stream));
}

int jpdse_conv_dgrad_relu(const jpdse_conv_desc* d, const void* dy, const void* dgrad_pack, const void* x, void* dx,
                          void* ws, size_t ws_bytes, void* stream) {
  JPDSE_REQUIRE(x != nullptr, "conv_dgrad_relu: null mask tensor");
  return jpdse_conv_dgrad_fused(d, dy, dgrad_pack, x, nullptr, dx, ws, ws_bytes, stream);
}

int jpdse_conv_dgrad_fused(const jpdse_conv_desc* d, const void* dy, const void* dgrad_pack, const void* x,
                           const void* addend, void* dx, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = validate(d)) return rc;
  JPDSE_REQUIRE(dy && dgrad_pack && dx, "conv_dgrad_fused: null pointer");
  ConvPlan p;
  make_plan(d, &p);
  const size_t need = jpdse_conv_workspace_size(d);
  if (ws == nullptr || ws_bytes < need)
    return set_error(JPDSE_EWORKSPACE, "conv_dgrad_fused: workspace %zu < %zu", ws_bytes, need);
  return d->dtype == JPDSE_BF16 ? conv_dgrad_t<bf16_t>(d, p, dy, dgrad_pack, dx, ws, as_stream(stream), x, addend)
                                : conv_dgrad_t<float>(d, p, dy, dgrad_pack, dx, ws, as_stream(stream), x, addend);
}

int jpdse_conv_dgrad_fused_lrelu(const jpdse_conv_desc* d, const void* dy, const void* dgrad_pack, const void* x,
                                 float slope, const void* addend, void* dx, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = validate(d)) return rc;
  JPDSE_REQUIRE(dy && dgrad_pack && dx && x, "conv_dgrad_fused_lrelu: null pointer");
  JPDSE_REQUIRE(slope >= 0.f && slope < 1.f, "conv_dgrad_fused_lrelu: slope %g outside [0, 1)", (double)slope);
  ConvPlan p;
  make_plan(d, &p);
  const size_t need = jpdse_conv_workspace_size(d);
  if (ws == nullptr || ws_bytes < need)
    return set_error(JPDSE_EWORKSPACE, "conv_dgrad_fused_lrelu: workspace %zu < %zu", ws_bytes, need);
  return d->dtype == JPDSE_BF16
             ? conv_dgrad_t<bf16_t>(d, p, dy, dgrad_pack, dx, ws, as_stream(stream), x, addend, nullptr, slope)
             : conv_dgrad_t<float>(d, p, dy, dgrad_pack, dx, ws, as_stream(stream), x, addend, nullptr, slope);
}

int32_t jpdse_conv_dgrad_nsum_slots(const jpdse_conv_desc* d) {
  if (validate(d)) return 0;
  ConvPlan p;
  make_plan(d, &p);
  return dgrad_nsum_slots(d, p);
}

int jpdse_conv_dgrad_fused_nsums(const jpdse_conv_desc* d, const void* dy, const void* dgrad_pack, const void* x,
                                 const void* addend, void* dx, const void* norm_x, const float* norm_stats, int32_t norm_act,
                                 float norm_slope, float* sums, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = validate(d)) return rc;
  JPDSE_REQUIRE(dy && dgrad_pack && dx && norm_x && norm_stats && sums, "conv_dgrad_fused_nsums: null pointer");
  JPDSE_REQUIRE(norm_act == JPDSE_ACT_NONE || norm_act == JPDSE_ACT_RELU || norm_act == JPDSE_ACT_LRELU,
                "conv_dgrad_fused_nsums: activation %d", norm_act);
  ConvPlan p;
  make_plan(d, &p);
  JPDSE_REQUIRE(d->dtype == JPDSE_BF16 && dgrad_nsum_slots(d, p) > 0,
                "conv_dgrad_fused_nsums: this layer has no norm-backward-sum epilogue (jpdse_conv_dgrad_nsum_slots == 0)");
  const size_t need = jpdse_conv_workspace_size(d);
  if (ws == nullptr || ws_bytes < need)
    return set_error(JPDSE_EWORKSPACE, "conv_dgrad_fused_nsums: workspace %zu < %zu", ws_bytes, need);
  const NormSink sink = {norm_x, norm_stats, sums, norm_act, norm_slope};
  return conv_dgrad_t<bf16_t>(d, p, dy, dgrad_pack, dx, ws, as_stream(stream), x, addend, nullptr, 0.f, &sink);
}

int jpdse_conv_wgrad(const jpdse_conv_desc* d, const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes,
                     void* stream) {
  if (int rc = validate(d)) return rc;
  JPDSE_REQUIRE(x && dy && dw, "conv_wgrad: null pointer");
  ConvPlan p;
  make_plan(d, &p);
  if (ws == nullptr || ws_bytes < jpdse_conv_workspace_size(d))
    return set_error(JPDSE_EWORKSPACE, "conv_wgrad: workspace %zu < %zu", ws_bytes, p.xpad_bytes);
  return d->dtype == JPDSE_BF16 ? conv_wgrad_t<bf16_t>(d, p, x, dy, dw, ws, as_stream(stream))
                                : conv_wgrad_t<float>(d, p, x, dy, dw, ws, as_stream(stream));
}

// nn.ConvTranspose2d == the data-gradient of the Conv2d described by `d` (see jpdse.h)
int jpdse_convT_fwd(const jpdse_conv_desc* d, const void* x, const void* dgrad_pack, void* y, void* ws,
                    size_t ws_bytes, void* stream) {
  return jpdse_conv_dgrad(d, x, dgrad_pack, y, ws, ws_bytes, stream);
}
int jpdse_convT_dgrad(const jpdse_conv_desc* d, const void* dy, const void* fwd_pack, void* dx, void* ws,
                      size_t ws_bytes, void* stream) {
  return jpdse_conv_fwd(d, dy, fwd_pack, nullptr, dx, ws, ws_bytes, stream);
}
int jpdse_convT_wgrad(const jpdse_conv_desc* d, const void* x, const void* dy, float* dw, void* ws,
                      size_t ws_bytes, void* stream) {
  return jpdse_conv_wgrad(d, dy, x, dw, ws, ws_bytes, stream);
}

}  // extern "C"
