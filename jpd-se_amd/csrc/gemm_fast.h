// Fast bf16 implicit-GEMM convolution kernel (forward / data gradient) for CDNA4.
//
// Used whenever the input channel storage is a multiple of 64 (every layer of the hot path except
// the 40-channel network inputs and the 8-channel images).  Differences from gemm_fwd_kernel:
//   * no materialised padding: the A loader resolves reflect / zero padding per K-tile (one K-tile
//     = 64 channels of ONE filter tap), zero taps read a zero page;
//   * staging is LDS-DMA: `global_load_lds_dwordx4` writes 1 KiB per wave instruction straight
//     into LDS (16 rows x 64 B of a half-tile).  The DMA destination is lane-linear, so the
//     conflict-free XOR swizzle of the 16-byte slots is applied to the per-lane SOURCE address and
//     again on the fragment reads (cdna_hip_programming.md rule 21);
//   * a 3-stage LDS ring (3 x 48 KiB for the 256x128 tile), one raw s_barrier per K-tile and a
//     COUNTED s_waitcnt vmcnt(N): the DMA of tile t+1 stays in flight across the barrier while
//     tile t is consumed and tile t+2 is issued (never vmcnt(0) inside the loop);
//   * 8 waves (2 per SIMD), BK = 64, s_setprio around the MFMA cluster.
#pragma once
#include "common.h"

namespace jpdse {

__device__ __attribute__((aligned(256))) uint32_t g_zero_page[64];   // 256 B of zeros (static init)

struct FastArgs {
  const bf16_t* X;     // NHWC input [N][IH][IW][Cs], NOT padded
  const bf16_t* B;     // panel [b_rows][R*S*Cs]
  const float* bias;
  bf16_t* Y;
  int M, OH, OW;
  int IH, IW, Cs;
  int R, S;
  int sy, sx;          // ih = oh*sy + r - py
  int py, px;
  int reflect;
  int Kout, Ks, b_rows;
  long long out_sn, out_sh, out_sw, out_base;
  int act;
  float slope;
  // split-K (small M: too few tiles for 256 CUs): block (tile, split) reduces K-tiles
  // [T*split/splits, T*(split+1)/splits) and stores its fp32 partial tile into slab `split` of
  // `partial` ([splits][M][Ks]); splitk_finish_kernel sums the slabs in a fixed order (deterministic).
  int splits;
  float* partial;
  // optional fused ReLU backward: Y is zeroed where mask <= 0 (mask has Y's addressing: the conv's input)
  const bf16_t* mask;
  const bf16_t* addend;   // optional: Y = result + addend (same addressing as Y), before the mask
  float mask_slope;       // LeakyReLU backward: Y is scaled by this where mask <= 0 (0 = ReLU backward)
  // optional generalisations (0 = the dense defaults): batch stride of X (a sub-image of a larger tensor),
  // B row stride and B offsets per filter-row / filter-column step (a sub-panel of a larger filter panel)
  long long x_sn, x_sh;   // batch / row strides of X in elements
  long long b_stride;
  int b_tap_r, b_tap_s;
  int no_finish;       // split-K: leave the slabs to the caller (no splitk_finish_kernel)
  long long x_extent;  // elements readable from X (0 = not given): the launcher refuses a problem whose last pixel lies beyond
  int xcd_map;         // N-tiles of an M-tile on consecutive slots of one XCD (set by the launcher)
  int abl;             // unused by the kernel (the ablations are compile-time VAR bits); kept so that the struct layout does not depend on the build
};

// Up to 4 independent problems in one launch (the stride-2 sub-pixel phases of a data gradient /
// ConvTranspose2d forward): block b belongs to problem q with first_tile[q] <= b < first_tile[q+1].
struct FastBatch {
  FastArgs p[4];
  int first_tile[5];
  int n;
  int small_m;         // host only: every problem has few rows (the ring strips): 128-row tiles, two blocks per CU
};

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {      // 32-bit LDS address of a generic pointer into LDS
  return (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)p);
}

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LDS tile [rows][128 B] (one row = 64 bf16 of K): the eight 16-byte slots of a row are XOR-swizzled
// with (row >> 1) & 7, which makes ds_read_b128 conflict free (rows of equal parity in a 16-lane read
// group get distinct slots) while every DMA instruction moves 8 FULL 128-byte lines (fragment-shaped
// 64-byte pieces cost 12-28 %: cdna_hip_programming.md, "x through LDS in full 128-B lines").
__device__ __forceinline__ int swz128(int row, int slot) { return (row << 7) + (((slot ^ (row >> 1)) & 7) << 4); }

// Epilogue staging: a wave's TM x TN accumulator tiles -> bias + activation -> bf16 into an LDS
// tile [rows][pitch = BN*2 + 64 bytes] so that the global stores are 16-byte vectors along the
// channel axis (the accumulator layout would give 2-byte stores, 64 B contiguous per 32 lanes).
// Lane pairs exchange one value per two accumulator rows so that every LDS store is a packed
// (col, col+1) dword; with the 64-byte pitch skew the even / odd lanes hit disjoint banks.
// The activation is a COMPILE-TIME parameter of the body and the run-time `act` is dispatched once around it (round 4): with
// apply_act(v, act, slope) called per value, hipcc kept the four-way choice -- incl. the tanh expansion -- as scalar branches
// around every one of the 64 values of a wave: ~7,900 instructions and 557 branches behind the last MFMA of the halo kernel,
// 6-8 us per 256 x 128 tile (timing-only ablations, profiles/r04_fast_astage_ablation.txt).  The lane-pair exchange is a DPP
// quad permute (no LDS round trip), as in the row-streaming kernels.
__device__ __forceinline__ float pair_swap(float v) {      // value of lane ^ 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}

template <int TM, int TN, int ACT>
__device__ __forceinline__ void acc_tile_to_lds_ct(char* tile, int pitch, int wrow0, int wcol0, int n0, int lane,
                                                   const f32x16 (&acc)[TM][TN], const float* bias, int Kout, float slope) {
  const int odd = lane & 1;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int lcol = wcol0 + j * 32 + (lane & 31);
    const int col = n0 + lcol;
    const bool live = col < Kout;
    const float bv = (bias != nullptr && live) ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int ep = 0; ep < 8; ++ep) {
        const int e = 2 * ep;
        const float v0 = live ? act_ct<ACT>(acc[i][j][e] + bv, slope) : 0.f;
        const float v1 = live ? act_ct<ACT>(acc[i][j][e + 1] + bv, slope) : 0.f;
        const float recv = pair_swap(odd ? v0 : v1);
        const float lo = odd ? recv : v0, hi = odd ? v1 : recv;
        const uint32_t word = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
        const int row = wrow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) + odd;
        *reinterpret_cast<uint32_t*>(tile + row * pitch + (lcol - odd) * 2) = word;
      }
    }
  }
}
template <int TM, int TN>
__device__ __forceinline__ void acc_tile_to_lds(char* tile, int pitch, int wrow0, int wcol0, int n0, int lane,
                                                const f32x16 (&acc)[TM][TN], const float* bias, int Kout, int act,
                                                float slope) {
  if (act == JPDSE_ACT_RELU) acc_tile_to_lds_ct<TM, TN, JPDSE_ACT_RELU>(tile, pitch, wrow0, wcol0, n0, lane, acc, bias, Kout, slope);
  else if (act == JPDSE_ACT_LRELU) acc_tile_to_lds_ct<TM, TN, JPDSE_ACT_LRELU>(tile, pitch, wrow0, wcol0, n0, lane, acc, bias, Kout, slope);
  else if (act == JPDSE_ACT_TANH) acc_tile_to_lds_ct<TM, TN, JPDSE_ACT_TANH>(tile, pitch, wrow0, wcol0, n0, lane, acc, bias, Kout, slope);
  else acc_tile_to_lds_ct<TM, TN, JPDSE_ACT_NONE>(tile, pitch, wrow0, wcol0, n0, lane, acc, bias, Kout, slope);
}

// Same for 16x16 accumulator blocks (v_mfma_f32_16x16x32_bf16: lane -> column lane & 15, rows 4 * (lane >> 4) + e)
template <int TM, int TN, int ACT>
__device__ __forceinline__ void acc16_tile_to_lds_ct(char* tile, int pitch, int wrow0, int wcol0, int n0, int lane,
                                                     const f32x4 (&acc)[TM][TN], const float* bias, int Kout, float slope) {
  const int odd = lane & 1;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int lcol = wcol0 + j * 16 + (lane & 15);
    const int col = n0 + lcol;
    const bool live = col < Kout;
    const float bv = (bias != nullptr && live) ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int ep = 0; ep < 2; ++ep) {
        const int e = 2 * ep;
        const float v0 = live ? act_ct<ACT>(acc[i][j][e] + bv, slope) : 0.f;
        const float v1 = live ? act_ct<ACT>(acc[i][j][e + 1] + bv, slope) : 0.f;
        const float recv = pair_swap(odd ? v0 : v1);
        const float lo = odd ? recv : v0, hi = odd ? v1 : recv;
        const uint32_t word = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
        const int row = wrow0 + i * 16 + 4 * (lane >> 4) + e + odd;
        *reinterpret_cast<uint32_t*>(tile + row * pitch + (lcol - odd) * 2) = word;
      }
    }
  }
}
template <int TM, int TN>
__device__ __forceinline__ void acc16_tile_to_lds(char* tile, int pitch, int wrow0, int wcol0, int n0, int lane,
                                                  const f32x4 (&acc)[TM][TN], const float* bias, int Kout, int act,
                                                  float slope) {
  if (act == JPDSE_ACT_RELU) acc16_tile_to_lds_ct<TM, TN, JPDSE_ACT_RELU>(tile, pitch, wrow0, wcol0, n0, lane, acc, bias, Kout, slope);
  else if (act == JPDSE_ACT_LRELU) acc16_tile_to_lds_ct<TM, TN, JPDSE_ACT_LRELU>(tile, pitch, wrow0, wcol0, n0, lane, acc, bias, Kout, slope);
  else if (act == JPDSE_ACT_TANH) acc16_tile_to_lds_ct<TM, TN, JPDSE_ACT_TANH>(tile, pitch, wrow0, wcol0, n0, lane, acc, bias, Kout, slope);
  else acc16_tile_to_lds_ct<TM, TN, JPDSE_ACT_NONE>(tile, pitch, wrow0, wcol0, n0, lane, acc, bias, Kout, slope);
}

// 8 bf16 values: v + w in fp32, rounded once
__device__ __forceinline__ u32x4 add_bf16x8(u32x4 v, u32x4 w) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = __uint_as_float(v[i] << 16) + __uint_as_float(w[i] << 16);
    const float hi = __uint_as_float(v[i] & 0xffff0000u) + __uint_as_float(w[i] & 0xffff0000u);
    v[i] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
  }
  return v;
}

// element-wise maximum of 8 bf16 values (exact: the result is one of the inputs)
__device__ __forceinline__ u32x4 max_bf16x8(u32x4 a, u32x4 b) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float alo = __uint_as_float(a[i] << 16), blo = __uint_as_float(b[i] << 16);
    const float ahi = __uint_as_float(a[i] & 0xffff0000u), bhi = __uint_as_float(b[i] & 0xffff0000u);
    a[i] = (__float_as_uint(fmaxf(alo, blo)) >> 16) | (__float_as_uint(fmaxf(ahi, bhi)) & 0xffff0000u);
  }
  return a;
}

// 8 bf16 values of `v` zeroed where the matching value of `m` is <= 0 (or NaN)
__device__ __forceinline__ u32x4 relu_mask8(u32x4 v, u32x4 m) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t lo = m[i] & 0xffffu, hi = m[i] >> 16;
    const uint32_t klo = (lo - 1u) < 0x7f80u ? 0xffffu : 0u;          // 0 < bits <= +inf
    const uint32_t khi = (hi - 1u) < 0x7f80u ? 0xffff0000u : 0u;
    v[i] &= (klo | khi);
  }
  return v;
}

// LeakyReLU form: values of `v` scaled by `slope` (and rounded to bf16 again) where the matching value of `m` is <= 0
__device__ __forceinline__ u32x4 lrelu_mask8(u32x4 v, u32x4 m, float slope) {
  if (slope == 0.f) return relu_mask8(v, m);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t lo = m[i] & 0xffffu, hi = m[i] >> 16;
    const uint32_t slo = f2bf(__uint_as_float(v[i] << 16) * slope);
    const uint32_t shi = f2bf(__uint_as_float(v[i] & 0xffff0000u) * slope);
    const uint32_t rlo = (lo - 1u) < 0x7f80u ? (v[i] & 0xffffu) : slo;
    const uint32_t rhi = (hi - 1u) < 0x7f80u ? (v[i] >> 16) : shi;
    v[i] = rlo | (rhi << 16);
  }
  return v;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// VAR: 0 in every shipped instantiation; the developer build instantiates timing-only ablations of two configurations
// (bits documented at the loader; modes 210-219 of jpdse_debug_set_fast_path).  (A software-pipelined fragment schedule was
// measured 2-3 % SLOWER than the plain one in the same process and removed.)
// STAGES = 3: ring with one tile in flight across the barrier (counted vmcnt).  STAGES = 2 (used by
// the 320-row tile, whose 3-stage ring would not fit 160 KiB): next tile issued right after the barrier,
// vmcnt(0) at the following one.
template <int WM, int WN, int TM, int TN, int VAR, int STAGES>
__global__ __launch_bounds__(64 * WM * WN) void gemm_fast_kernel(const FastBatch batch) {
  int q = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < batch.n && (int)blockIdx.x >= batch.first_tile[i]) q = i;
  const FastArgs& a = batch.p[q];
  const int block_id = (int)blockIdx.x - batch.first_tile[q];
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_TILE = BM * 128, B_TILE = BN * 128;
  constexpr int STAGE_BYTES = A_TILE + B_TILE;
  constexpr int A_UNITS = BM / 8, B_UNITS = BN / 8;   // 1 KiB DMA units: 8 rows x 128 B
  constexpr int AU = (A_UNITS + NW - 1) / NW, BU = (B_UNITS + NW - 1) / NW;
  static_assert(A_UNITS % NW == 0, "every wave stages the same number of A units");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_m = (a.M + BM - 1) / BM;
  const int ntiles = tiles_m * ((a.Ks + BN - 1) / BN);
  const int split = a.splits > 1 ? block_id / ntiles : 0;
  const int tile_id = a.splits > 1 ? block_id - split * ntiles : block_id;
  int tile_m = tile_id % tiles_m, tile_n = tile_id / tiles_m;
  if (a.xcd_map) {
    // The N-tiles of one M-tile read the same activation rows.  In the plain order they run tiles_m blocks apart, and the
    // activation tensors of these layers (34-134 MB) do not live that long in a 4 MB L2: every N-tile re-fetches its rows
    // from beyond L2 (PatchGAN layer 3: 1.49 GB per launch for a 34 MB input).  Blocks b, b + 8, ... share an XCD (speed only:
    // MI355X_MICROARCH.md "XCD placement"), so deal each XCD whole M-tiles: consecutive slots of one XCD = the N-tiles of one
    // M-tile.  The last tiles_m % 8 M-tiles keep the plain order.
    const int tiles_n = ntiles / tiles_m;
    const int m8 = tiles_m & ~7, head = m8 * tiles_n;
    if (tile_id < head) {
      const int x = tile_id & 7, j = tile_id >> 3;
      tile_n = j % tiles_n;
      tile_m = (j / tiles_n) * 8 + x;
    } else {
      const int rem = tiles_m - m8, t = tile_id - head;
      tile_m = m8 + t % rem;
      tile_n = t / rem;
    }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int lrow = lane >> 3, lslot = lane & 7;

  // ---- per-unit staging state -------------------------------------------------------------
  const bf16_t* a_xb[AU];             // image base + the lane's channel slot; in-image offsets are 32-bit (host-checked)
  int a_oh[AU], a_ow[AU], a_lds[AU];
#pragma unroll
  for (int i = 0; i < AU; ++i) {
    const int u = wid + i * NW;
    const int row = u * 8 + lrow;
    int m = m0 + row;
    m = m < a.M ? m : a.M - 1;
    const int ow = m % a.OW, t = m / a.OW, oh = t % a.OH, n = t / a.OH;
    a_xb[i] = a.X + (long long)n * (a.x_sn ? a.x_sn : (long long)a.IH * a.IW * a.Cs) + ((lslot ^ (row >> 1)) & 7) * 8;
    a_oh[i] = oh * a.sy - a.py;
    a_ow[i] = ow * a.sx - a.px;
    a_lds[i] = u * 1024;
  }
  const bf16_t* b_ptr[BU];
  int b_lds[BU];
  bool b_on[BU];
  const long long ktot = a.b_stride ? a.b_stride : (long long)a.R * a.S * a.Cs;
  const long long x_sh = a.x_sh ? a.x_sh : (long long)a.IW * a.Cs;
  const int b_tap_r = a.b_tap_r ? a.b_tap_r : a.S * a.Cs, b_tap_s = a.b_tap_s ? a.b_tap_s : a.Cs;
#pragma unroll
  for (int j = 0; j < BU; ++j) {
    const int u = wid + j * NW;
    b_on[j] = (B_UNITS % NW == 0) || u < B_UNITS;      // compile-time true for every shipped configuration
    const int uu = b_on[j] ? u : 0;
    const int row = uu * 8 + lrow;
    int br = n0 + row;
    br = br < a.b_rows ? br : a.b_rows - 1;
    b_ptr[j] = a.B + (long long)br * ktot + ((lslot ^ (row >> 1)) & 7) * 8;
    b_lds[j] = A_TILE + uu * 1024;
  }
  int n_b = 0;
#pragma unroll
  for (int j = 0; j < BU; ++j) n_b += b_on[j] ? 1 : 0;   // wave-uniform
  const int LW = (B_UNITS % NW == 0) ? AU + BU : AU + n_b;   // DMA instructions per wave per K-tile

  int a_rd[TM][4], b_rd[TN][4];       // byte offset of the lane's fragment for each of the 4 k16-steps
  {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * TM * 32 + i * 32 + r;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a_rd[i][ks] = swz128(row, 2 * ks + h);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * TN * 32 + j * 32 + r;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) b_rd[j][ks] = A_TILE + swz128(row, 2 * ks + h);
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int CC = a.Cs >> 6;
  const int T_all = a.R * a.S * CC;
  const int t_begin = a.splits > 1 ? (int)((long long)T_all * split / a.splits) : 0;
  const int t_end = a.splits > 1 ? (int)((long long)T_all * (split + 1) / a.splits) : T_all;
  const int T_total = t_end - t_begin;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);

  // K-tile cursor of the NEXT tile to issue.  The padded source address of a tap is resolved only
  // when the tap (r,s) changes; within a tap consecutive K-tiles advance the pointer by 64 channels.
  int istage = 0;
  int ic = t_begin % CC, is = (t_begin / CC) % a.S, ir = (t_begin / CC) / a.S;
  const bf16_t* a_src[AU];
  int a_step[AU];
  // Branch-free: the reflected coordinates are computed for both padding modes (they are in range for any input),
  // zero padding only adds the in-range predicate to the final select.  (The if / else form compiled to four exec
  // branches per unit, executed behind the barrier whenever the tap changes -- every K-tile for 64-channel inputs.)
  const int IHm1 = a.IH - 1, IWm1 = a.IW - 1;
  const int x_sh32 = (int)x_sh;
  const bool refl = a.reflect != 0;
  auto retap = [&]() {
#pragma unroll
    for (int i = 0; i < AU; ++i) {
      const int ih = a_oh[i] + ir, iw = a_ow[i] + is;
      int rh = ih < 0 ? -ih : ih, rw = iw < 0 ? -iw : iw;
      rh = rh > IHm1 ? 2 * IHm1 - rh : rh;
      rw = rw > IWm1 ? 2 * IWm1 - rw : rw;
      rh = rh < 0 ? 0 : rh;                             // zero padding wider than the image: any in-range pixel
      rw = rw < 0 ? 0 : rw;
      const bool ok = refl | (((unsigned)ih < (unsigned)a.IH) & ((unsigned)iw < (unsigned)a.IW));
      const bf16_t* const src = a_xb[i] + (unsigned)(rh * x_sh32 + rw * a.Cs);
      a_src[i] = ok ? src : zero;
      a_step[i] = ok ? 64 : 0;
    }
  };
  bool primed = false;                   // VAR & 1: set after the prologue issues
  auto issue = [&]() {
    char* const st = smem + istage * STAGE_BYTES;
    if (ic == 0) retap();
    // VAR (developer build, timing only, wrong results): 1 = no DMA after the prologue tiles, 2 = no barrier, 4 = fragments read for k-step 0 only,
    // 8 = no MFMAs (fragment reads kept alive), 16 = activation tile staged for one tap in four (the waits then pass early: optimistic),
    // 32 = no epilogue, 64 = no K loop
    const bool skip_a = (VAR & 16) != 0 && ((ir * a.S + is) & 3) != 0;
#pragma unroll
    for (int i = 0; i < AU; ++i) {
      if (!skip_a && !((VAR & 1) && primed)) glds16(a_src[i], st + a_lds[i]);
      a_src[i] += a_step[i];
    }
    const long long koff = (long long)ir * b_tap_r + is * b_tap_s + ic * 64;
#pragma unroll
    for (int j = 0; j < BU; ++j)
      if (b_on[j] && !((VAR & 1) && primed)) glds16(b_ptr[j] + koff, st + b_lds[j]);
    if (++ic == CC) {
      ic = 0;
      if (++is == a.S) { is = 0; ++ir; }
    }
    istage = istage == STAGES - 1 ? 0 : istage + 1;
  };

  if (ic != 0) {                          // split-K range starting inside a tap
    retap();
#pragma unroll
    for (int i = 0; i < AU; ++i) a_src[i] += ic * a_step[i];
  }
  constexpr int AHEAD = STAGES - 1;       // tiles issued before tile t is consumed
  static_assert(STAGES >= 2 && STAGES <= 4, "the counted waits below are written for up to three tiles ahead");
  static_assert(STAGES < 4 || B_UNITS % NW == 0, "4 stages: every wave issues the same number of pieces per tile");
  issue();
  if (AHEAD > 1 && T_total > 1) issue();
  if (AHEAD > 2 && T_total > 2) issue();
  int cstage = 0;
  primed = true;
  for (int t = 0; t < ((VAR & 64) ? 0 : T_total); ++t) {      // VAR & 64 (timing only): no loop at all -- prologue (+ epilogue) alone
    if (AHEAD > 2 && t + 2 < T_total) {
      wait_vmcnt<2 * (AU + BU)>();            // tiles t+1 and t+2 stay in flight
    } else if (AHEAD > 1 && t + 1 < T_total) {
      if (LW == AU + BU) wait_vmcnt<AU + BU>();
      else wait_vmcnt<AU + (BU > 0 ? BU - 1 : 0)>();
    } else {
      wait_vmcnt<0>();
    }
    if (!(VAR & 2)) __builtin_amdgcn_s_barrier();
    const char* const st = smem + cstage * STAGE_BYTES;
    if (t + AHEAD < T_total) issue();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s16x8 af[TM], bf[TN];
      const int ks_r = (VAR & 4) ? 0 : ks;
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const s16x8*>(st + a_rd[i][ks_r]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const s16x8*>(st + b_rd[j][ks_r]);
      if constexpr ((VAR & 8) != 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" :: "v"(af[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(bf[j]));
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    cstage = cstage == STAGES - 1 ? 0 : cstage + 1;
  }

  // ---- epilogue --------------------------------------------------------------------------
  if constexpr ((VAR & 32) != 0) {        // timing only: no epilogue (one store per thread keeps the accumulators alive)
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) keep += acc[i][j][e];
    if (keep == 12345.678f) a.Y[0] = 0;
    return;
  }
  if (a.splits > 1 || a.no_finish) {
    float* const slab = a.partial + (long long)split * a.M * a.Ks;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + (lane & 31);
      if (col >= a.Ks) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          if (m < a.M) slab[(long long)m * a.Ks + col] = acc[i][j][e];
        }
      }
    }
    return;
  }
  __syncthreads();
  long long* const row_off = reinterpret_cast<long long*>(smem);
  char* const tile = smem + BM * 8;
  constexpr int PITCH = BN * 2 + 64;
  static_assert(BM * 8 + BM * PITCH <= STAGES * STAGE_BYTES, "epilogue tile fits the pipeline LDS");
  for (int row = tid; row < BM; row += 64 * NW) {
    const int m = m0 + row;
    long long off = -1;
    if (m < a.M) {
      const int ow = m % a.OW, t = m / a.OW, oh = t % a.OH, n = t / a.OH;
      off = a.out_base + n * a.out_sn + oh * a.out_sh + ow * a.out_sw;
    }
    row_off[row] = off;
  }
  acc_tile_to_lds<TM, TN>(tile, PITCH, wm * TM * 32, wn * TN * 32, n0, lane, acc, a.bias, a.Kout, a.act, a.slope);
  __syncthreads();
  constexpr int VPR = BN / 8;
  constexpr int NV = BM * VPR / (64 * NW);
  static_assert(BM * VPR % (64 * NW) == 0, "every thread owns the same number of output vectors");
  if (a.addend != nullptr || a.mask != nullptr) {
    // fused operands loaded for all of the thread's vectors before the first store (see gemm_halo.h): groups of 8 vectors
    constexpr int G8 = NV < 8 ? NV : 8;
    for (int it0 = 0; it0 < NV; it0 += G8) {
      u32x4 addv[G8], mskv[G8];
#pragma unroll
      for (int g = 0; g < G8; ++g) {
        const int idx = tid + 64 * NW * (it0 + g);
        const int row = idx / VPR, v = idx - row * VPR;
        const long long off = row_off[row];
        const bool on = it0 + g < NV && off >= 0 && n0 + v * 8 < a.Ks;
        if (on && a.addend != nullptr) addv[g] = *reinterpret_cast<const u32x4*>(a.addend + off + n0 + v * 8);
        if (on && a.mask != nullptr) mskv[g] = *reinterpret_cast<const u32x4*>(a.mask + off + n0 + v * 8);
      }
#pragma unroll
      for (int g = 0; g < G8; ++g) {
        const int idx = tid + 64 * NW * (it0 + g);
        const int row = idx / VPR, v = idx - row * VPR;
        const long long off = row_off[row];
        if (it0 + g >= NV || off < 0 || n0 + v * 8 >= a.Ks) continue;
        u32x4 val = *reinterpret_cast<const u32x4*>(tile + row * PITCH + v * 16);
        if (a.addend != nullptr) val = add_bf16x8(val, addv[g]);
        if (a.mask != nullptr) val = lrelu_mask8(val, mskv[g], a.mask_slope);
        *reinterpret_cast<u32x4*>(a.Y + off + n0 + v * 8) = val;
      }
    }
    return;
  }
  for (int idx = tid; idx < BM * VPR; idx += 64 * NW) {
    const int row = idx / VPR, v = idx - row * VPR;
    const long long off = row_off[row];
    if (off < 0 || n0 + v * 8 >= a.Ks) continue;
    if constexpr ((VAR & 128) != 0) {      // timing only: the staging without the global stores (one conditional store keeps the reads alive)
      const u32x4 val = *reinterpret_cast<const u32x4*>(tile + row * PITCH + v * 16);
      if (val[0] == 0x12345678u && val[3] == 0x9abcdef0u) *reinterpret_cast<u32x4*>(a.Y + off + n0 + v * 8) = val;
      continue;
    }
    *reinterpret_cast<u32x4*>(a.Y + off + n0 + v * 8) = *reinterpret_cast<const u32x4*>(tile + row * PITCH + v * 16);
  }
}

// Sum of the split-K slabs + bias + activation -> bf16 output (8 channels per thread).
__global__ __launch_bounds__(256) void splitk_finish_kernel(const FastArgs a, long long total_vec) {
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= total_vec) return;
  const int vpr = a.Ks >> 3;
  const int m = (int)(v / vpr), c0 = (int)(v % vpr) * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  // slabs are read four at a time before they are added (same order): one L2 / HBM round trip per split otherwise
  const float* const base = a.partial + (long long)m * a.Ks + c0;
  const long long slab_elems = (long long)a.M * a.Ks;
  int sidx = 0;
  for (; sidx + 4 <= a.splits; sidx += 4) {
    float4 lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4* src = reinterpret_cast<const float4*>(base + (sidx + u) * slab_elems);
      lo[u] = src[0];
      hi[u] = src[1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] += lo[u].x; acc[1] += lo[u].y; acc[2] += lo[u].z; acc[3] += lo[u].w;
      acc[4] += hi[u].x; acc[5] += hi[u].y; acc[6] += hi[u].z; acc[7] += hi[u].w;
    }
  }
  for (; sidx < a.splits; ++sidx) {
    const float4* src = reinterpret_cast<const float4*>(base + sidx * slab_elems);
    const float4 lo = src[0], hi = src[1];
    acc[0] += lo.x; acc[1] += lo.y; acc[2] += lo.z; acc[3] += lo.w;
    acc[4] += hi.x; acc[5] += hi.y; acc[6] += hi.z; acc[7] += hi.w;
  }
  const int ow = m % a.OW, t = m / a.OW, oh = t % a.OH, n = t / a.OH;
  const long long off = a.out_base + n * a.out_sn + oh * a.out_sh + ow * a.out_sw;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int col = c0 + e;
    const bool live = col < a.Kout;
    const float bv = (a.bias != nullptr && live) ? a.bias[col] : 0.f;
    o[e] = live ? apply_act(acc[e] + bv, a.act, a.slope) : 0.f;
  }
  if (a.addend != nullptr) {
    float ad[8];
    Vec16<bf16_t>::load(a.addend + off + c0, ad);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += ad[e];
  }
  if (a.mask != nullptr) {
    float mk[8];
    Vec16<bf16_t>::load(a.mask + off + c0, mk);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = mk[e] > 0.f ? o[e] : (a.mask_slope == 0.f ? 0.f : bf16_round(o[e]) * a.mask_slope);
  }
  Vec16<bf16_t>::store(a.Y + off + c0, o);
}

}  // namespace jpdse
