// bf16 weight gradient for the NARROW layers at high resolution (64..256 output channels, 64..128 input
// channels: the stride-2 down / up-sampling convs next to the network ends, PatchGAN layer 1, the 64-channel
// ResnetBlocks of the LocalEnhancer).  There the per-tap GEMM is tiny (M x N = K x C <= 256 x 128) and
// wgrad_fast_kernel, which stages a dy tile and an x tile for every single tap, spends its time filling LDS:
// 9 (16) taps re-read the same dy pixels and nearly the same input pixels.
//
// Here a block owns ALL taps of NROW filter rows (T = NROW * S taps) of a (BM k) x (BN c) tile: per chunk
// (64 consecutive output pixels of one output row) it stages the dy tile once and the input patch
// (NROW rows x (63*ST + S) pixels x BN channels; reflect / zero padding resolved per pixel by the loader)
// once; the B fragment of tap (rr, s) is a transposed read (ds_read_b64_tr_b16) of the patch at pixel
// rr*PW + pix*ST + s.  A fragments feed T MFMAs each.  Stride 2 only changes the pixel step of those reads.
// The K x C tile count is 1..3, so the pixel range is split over all CUs and every block writes its
// partial tile into its own fp32 slab; wgrad_taps_reduce_kernel sums the slabs in a fixed order
// (deterministic, no atomics).
#pragma once
#include "common.h"
#include "gemm_fast.h"
#include "wgrad_fast.h"

namespace jpdse {

struct TapsWgArgs {
  const bf16_t* X;    // [N][IH][IW][Cs] unpadded
  const bf16_t* DY;   // [N][OH][OW][Ks]
  float* partial;     // [blocks][T][BM][BN] fp32 slabs
  float* DW;          // fp32 KRSC (reduce kernel)
  int N, IH, IW, OH, OW, Cs, C, Ks, K;
  int R, S, pad, reflect;
  int chunks_per_row, chunks_total;
  int k_tiles, r_groups, c_tiles;      // tile = (kt * r_groups + rg) * c_tiles + ct
  int blocks_per_tile, chunks_per_block;
  int tiles;                           // k_tiles * r_groups * c_tiles
  int xg_gs, xg_gpx;                   // XCD co-location (conv_dispatch_wgrad.h::taps_partition): group size, groups per XCD; 0 = linear order
  int abl;                             // developer build, timing only (modes 200 + bits): 1 = no DMA after the prologue, 2 = no fragment reads / MFMAs (the finer ablations of round 4 -- no B reads, reads without MFMAs, no padding arithmetic -- sat inside the loop, cost it 10 % themselves and were removed)
};

// LDS row of patch pixel (filter row rr, patch column col).  Stride 2: the columns of a patch row are stored DE-INTERLEAVED,
// [even columns | odd columns], so that the 16 pixels of a k-step (col = 2 pix + s) are CONSECUTIVE LDS rows and the stride-1
// swizzle applies.  (Round 4: with the columns in natural order the four rows of one transposed read were 2 * ROWB = 256 B apart,
// i.e. on the same 64-byte quarter of the 64 banks; a 128-byte row offers only two quarters to swizzle into, so every B read of the
// stride-2 instantiations was a two-way bank conflict -- 37-47 % of their LDS cycles, profiles/r04_pmc_kernels.txt -- in a loop whose
// LDS traffic (T + TMW fragment reads per T * TMW MFMAs) is its bound.)
template <int ST, int PW> __device__ __forceinline__ constexpr int taps_lrow(int rr, int col) {
  return ST == 2 ? rr * PW + (col >> 1) + (col & 1) * ((PW + 1) / 2) : rr * PW + col;
}

// BKP = output pixels per chunk, NSTG = ring depth.  Round 4: the stride-2 instantiations run 32-pixel chunks through a FOUR-stage
// ring (two chunks in flight behind the one being consumed, counted vmcnt) instead of 64-pixel chunks through two stages (nothing in
// flight at the wait): the loop was bound by the fill LATENCY of one 48-65 KB stage per 0.8-1.2 us of MFMA work, not by LDS reads
// (removing the bank conflicts above halved the LDS cycles and changed no time).
template <int TMW, int WM, int WN, int S, int NROW, int ST, int BKP = 64, int NSTG = 2>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_taps_kernel(const TapsWgArgs a) {
  constexpr int NW = WM * WN, T = NROW * S;
  constexpr int BM = WM * TMW * 32, BN = WN * 32;
  constexpr int A_ROWB = BM * 2, B_ROWB = BN * 2;
  constexpr int PW = (BKP - 1) * ST + S, NP = NROW * PW;
  constexpr int A_PPU = 1024 / A_ROWB, B_PPU = 1024 / B_ROWB;
  constexpr int A_UNITS = BKP / A_PPU, B_UNITS = (NP + B_PPU - 1) / B_PPU;
  constexpr int SCRATCH = A_UNITS + B_UNITS;          // one more 1 KiB unit per stage: target of the waves that have no unit left
  constexpr int A_STAGE = BKP * A_ROWB, STAGE = (SCRATCH + 1) * 1024;   // (every wave issues the same number of DMAs per chunk: counted vmcnt)
  // Loader waves (tried in round 4): with eight waves (two per SIMD) only waves 0..3 -- one per SIMD -- issue the chunk's LDS-DMA and
  // waves 4..7 go straight from the barrier to their MFMAs, because the loader is ~200 scalar / vector instructions per wave and chunk
  // and all waves leave the barrier together (ablations, profiles/r04_wgrad_taps_ab.txt: the DMA costs 0.5 us per 1.0 us chunk of
  // MFMA work although the transfer itself is hidden).
  // (Measured 8-17 % SLOWER, kept as the NI parameter: the loader waves become the critical path -- one wave per SIMD cannot hide its own
  // LDS round trips -- so NI = NW: every wave loads.)
  constexpr int NI = NW;
  constexpr int AU = (A_UNITS + NI - 1) / NI, BU = (B_UNITS + NI - 1) / NI;
  constexpr int G = AU + BU;                          // LDS-DMA instructions per wave and chunk
  static_assert(BM >= 64 && BN >= 64 && TMW * T <= 12 && BKP % 16 == 0 && NSTG >= 2 && NSTG <= 4, "tile shape");
  static_assert(A_STAGE == A_UNITS * 1024, "dy tile = whole units");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  const uint32_t lds0 = lds_addr_of(smem);

  int tile, sub;
  if (a.xg_gs) {
    // block b -> XCD b % 8; that XCD's blocks are xg_gpx whole groups; group = (pixel range, part), tile = part * group size + member
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int gl = j / a.xg_gs, mem = j - gl * a.xg_gs;
    const int grp = xcd * a.xg_gpx + gl, parts = a.tiles / a.xg_gs;
    sub = grp / parts;
    tile = (grp - sub * parts) * a.xg_gs + mem;
  } else {
    tile = blockIdx.x / a.blocks_per_tile;
    sub = blockIdx.x - tile * a.blocks_per_tile;
  }
  const int ct = tile % a.c_tiles, t1 = tile / a.c_tiles;
  const int rg = t1 % a.r_groups, kt = t1 / a.r_groups;
  const int k0 = kt * BM, c0 = ct * BN, r0 = rg * NROW;
  const int ch_begin = sub * a.chunks_per_block;
  int ch_end = ch_begin + a.chunks_per_block;
  ch_end = ch_end < a.chunks_total ? ch_end : a.chunks_total;

  // transposed fragment offsets: A per m-tile, B per tap
  int a_tr[TMW], b_tr[T];
  {
    const int g = lane >> 4, li = lane & 15, h = g >> 1, cb = g & 1, q = li >> 2, p = li & 3;
    const int pix = 8 * h + q;
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
      const int ch = (wm * TMW + i) * 32 + cb * 16 + 4 * p;
      a_tr[i] = pix * A_ROWB + ((((ch >> 3) ^ trswz<A_ROWB>(pix)) << 4) | ((ch & 7) << 1));
    }
    const int chb = wn * 32 + cb * 16 + 4 * p;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int row = taps_lrow<ST, PW>(t / S, pix * ST + (t % S));
      b_tr[t] = A_STAGE + row * B_ROWB + ((((chb >> 3) ^ trswz<B_ROWB>(row)) << 4) | ((chb & 7) << 1));
    }
  }

  // ---- loader.  Chunks are walked DOWN the image: chunk index = (n * chunks_per_row + column chunk) * OH + oh.  Everything that
  // depends on the column chunk alone -- the reflected / clipped patch column of every lane, the ragged end of the dy tile -- is resolved
  // ONCE per column (`set_column`), so that the per-chunk loader is the row arithmetic, one add and the zero-page select per unit:
  // ~80 instead of ~210 instructions per wave and chunk.  It matters because all eight waves leave the barrier together and run the
  // loader before their first MFMA: in round 4's ablations the DMA cost 0.5 us per 1.0 us chunk of MFMA work although the transfer
  // itself is hidden, and three scalar branches more per unit cost the launch 10 % (profiles/r04_wgrad_taps_ab.txt).  Consecutive
  // chunks also share NROW - ST input rows now (L2 hits).
  const int a_pl = lane / (A_ROWB / 16);              // pixel of the lane within a dy unit
  int a_loff[AU], a_cur[AU];                          // element offset inside the chunk's dy tile; this column's copy (-1: beyond the row)
#pragma unroll
  for (int i = 0; i < AU; ++i) {
    const int pix = (wid + NI * i) * A_PPU + a_pl;
    const bool have = A_UNITS % NI == 0 || wid + NI * i < A_UNITS;
    a_loff[i] = have ? pix * a.Ks + k0 + (((lane % (A_ROWB / 16)) ^ trswz<A_ROWB>(pix)) << 3) : -1;
  }
  int b_rr[BU], b_col[BU], b_sw[BU], b_cur[BU];       // patch row, patch column (-1: no pixel), swizzled channel offset; this column's
#pragma unroll                                        // reflected column offset rw * Cs + b_sw (-1: zero)
  for (int i = 0; i < BU; ++i) {
    const int u = wid + NI * i;
    const int q = u * B_PPU + lane / (B_ROWB / 16);
    const int j = q % PW;                             // position inside the LDS patch row -> patch column (de-interleaved for stride 2)
    const int col = ST == 2 ? (j < (PW + 1) / 2 ? 2 * j : 2 * (j - (PW + 1) / 2) + 1) : j;
    const bool have = (B_UNITS % NI == 0 || u < B_UNITS) && q < NP && r0 + q / PW < a.R;
    b_rr[i] = have ? q / PW : 0;
    b_col[i] = have ? col : -1;
    b_sw[i] = c0 + (((lane % (B_ROWB / 16)) ^ trswz<B_ROWB>(q)) << 3);
  }
  const int row_el = a.IW * a.Cs;

  f32x16 acc[T][TMW];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][i][e] = 0.f;

  int i_oh = __builtin_amdgcn_readfirstlane(ch_begin % a.OH);
  int i_cw = __builtin_amdgcn_readfirstlane((ch_begin / a.OH) % a.chunks_per_row);
  int i_n = __builtin_amdgcn_readfirstlane((ch_begin / a.OH) / a.chunks_per_row);
  const int IHm1 = a.IH - 1, IWm1 = a.IW - 1;
  const bool refl = a.reflect != 0;
  auto set_column = [&]() {
    const int ow0 = i_cw * BKP, iw0 = ow0 * ST - a.pad;
    const int px_left = a.OW - ow0;
#pragma unroll
    for (int i = 0; i < AU; ++i)
      a_cur[i] = ((wid + NI * i) * A_PPU + a_pl < px_left) ? a_loff[i] : -1;
#pragma unroll
    for (int i = 0; i < BU; ++i) {
      const int iw = iw0 + b_col[i];
      int rw = iw < 0 ? -iw : iw;
      rw = rw > IWm1 ? 2 * IWm1 - rw : rw;
      // in range after one reflection?  (false for ragged chunks overhanging by more than the image; zero padding: the raw coordinate)
      const bool okw = (b_col[i] >= 0) & ((unsigned)(refl ? rw : iw) <= (unsigned)IWm1);
      b_cur[i] = okw ? __mul24(rw, a.Cs) + b_sw[i] : -1;
    }
  };
  set_column();
  auto issue = [&](int /*chunk*/, int stage) {
    char* const st = smem + stage * STAGE;
    const bf16_t* const dy_base = a.DY + (((long long)i_n * a.OH + i_oh) * a.OW + i_cw * BKP) * a.Ks;      // uniform
    const bf16_t* const x_img = a.X + (long long)i_n * a.IH * row_el;                                          // uniform
    const int ih0 = i_oh * ST + r0 - a.pad;
#pragma unroll
    for (int i = 0; i < AU; ++i) {
      const bool have = A_UNITS % NI == 0 || wid + NI * i < A_UNITS;      // wave-uniform
      const bool ok = a_cur[i] >= 0;
      const bf16_t* const src = dy_base + (ok ? a_cur[i] : 0);
      glds16(ok ? src : zero, st + (have ? wid + NI * i : SCRATCH) * 1024);
    }
#pragma unroll
    for (int i = 0; i < BU; ++i) {
      const bool have = B_UNITS % NI == 0 || wid + NI * i < B_UNITS;      // wave-uniform
      const int ih = ih0 + b_rr[i];
      int rh = ih < 0 ? -ih : ih;
      rh = rh > IHm1 ? 2 * IHm1 - rh : rh;
      const bool ok = (b_cur[i] >= 0) & ((unsigned)(refl ? rh : ih) <= (unsigned)IHm1);
      const bf16_t* const src = x_img + (ok ? __mul24(rh, row_el) + b_cur[i] : 0);
      glds16(ok ? src : zero, st + (have ? A_UNITS + wid + NI * i : SCRATCH) * 1024);
    }
    if (++i_oh == a.OH) {                             // next column of chunks (wave-uniform, once per OH chunks)
      i_oh = 0;
      if (++i_cw == a.chunks_per_row) { i_cw = 0; ++i_n; }
      set_column();
    }
  };

  auto mma_step = [&]<int KS>(uint32_t sbase) {
    s16x8 af[TMW], bf[T];
#pragma unroll
    for (int i = 0; i < TMW; ++i) af[i] = tr_frag_asm<KS * 16 * A_ROWB, KS * 16 * A_ROWB + 4 * A_ROWB>(sbase + a_tr[i]);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      bf[t] = tr_frag_asm<KS * 16 * B_ROWB, KS * 16 * B_ROWB + 4 * B_ROWB>(sbase + b_tr[t]);      // 16 pixels = 16 LDS rows for either stride
    }
    tr_wait(af);
    tr_wait(bf);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int i = 0; i < TMW; ++i)
        acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[t], acc[t][i], 0, 0, 0);
  };

  // NSTG-stage ring: chunks c+1 .. c+NSTG-2 stay in flight while chunk c is consumed, chunk c+NSTG-1 is issued into the stage that
  // chunk c-1 left (all waves are past this iteration's barrier, i.e. done reading it).  The asm reads keep hipcc from draining the DMA.
#pragma unroll
  for (int j = 0; j < NSTG - 1; ++j)
    if (wid < NI && ch_begin + j < ch_end) issue(ch_begin + j, j);
  int stage = 0;
  for (int c = ch_begin; c < ch_end; ++c) {
    const int rem = ch_end - 1 - c;                   // chunks issued behind c (wave-uniform)
    if (NSTG >= 4 && rem >= 2) wait_vmcnt<2 * G>();
    else if (NSTG >= 3 && rem >= 1) wait_vmcnt<G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#ifdef JPDSE_DEV
    if (!(a.abl & 1))
#endif
    if (wid < NI && c + NSTG - 1 < ch_end) issue(c + NSTG - 1, stage == 0 ? NSTG - 1 : stage - 1);
    const uint32_t sbase = lds0 + stage * STAGE;
#ifdef JPDSE_DEV
    if (!(a.abl & 2))
#endif
    {
      __builtin_amdgcn_s_setprio(1);
      mma_step.template operator()<0>(sbase);
      if constexpr (BKP > 16) mma_step.template operator()<1>(sbase);
      if constexpr (BKP > 32) mma_step.template operator()<2>(sbase);
      if constexpr (BKP > 48) mma_step.template operator()<3>(sbase);
      __builtin_amdgcn_s_setprio(0);
    }
    stage = stage == NSTG - 1 ? 0 : stage + 1;
  }

  // partial tile -> this block's slab [T][BM][BN]
  float* const slab = a.partial + ((long long)tile * a.blocks_per_tile + sub) * (T * BM * BN);
  const int cc = wn * 32 + (lane & 31);
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = (wm * TMW + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        slab[(t * BM + k) * BN + cc] = acc[t][i][e];
      }
}

// dW[k][r][s][c] = sum over the blocks of the owning tile, fixed order.  256 threads = 64 consecutive
// elements (VEC = 1) or 16-byte vectors along c (VEC = 4: C % 4 == 0; the slabs are ~50 MB per launch and 4-byte loads
// read them at 1.9 TB/s) x 4 slab groups; each thread keeps 8 independent loads in flight (a serial loop over 256-512
// slabs per element was latency-bound: ~100 us), the groups are combined through LDS.
template <int BM, int BN, int S, int NROW, int VEC>
__global__ __launch_bounds__(256) void wgrad_taps_reduce_kernel(const TapsWgArgs a, long long total_vec) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  constexpr int T = NROW * S;
  constexpr long long SLAB = (long long)T * BM * BN;
  __shared__ vec_t red[4][64];
  const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long long vidx = (long long)blockIdx.x * 64 + el;
  vec_t sum = {};
  if (vidx < total_vec) {
    const long long idx = vidx * VEC;
    const int c = (int)(idx % a.C);
    long long t1 = idx / a.C;
    const int s = (int)(t1 % a.S);
    t1 /= a.S;
    const int r = (int)(t1 % a.R), k = (int)(t1 / a.R);
    const int kt = k / BM, ct = c / BN, rg = r / NROW;
    const int tile = (kt * a.r_groups + rg) * a.c_tiles + ct;
    const int t = (r - rg * NROW) * S + s;
    const float* src = a.partial + ((long long)tile * a.blocks_per_tile) * SLAB +
                       ((long long)t * BM + (k - kt * BM)) * BN + (c - ct * BN);
    const int per = (a.blocks_per_tile + 3) / 4;
    const int b0 = grp * per;
    int b1 = b0 + per;
    b1 = b1 < a.blocks_per_tile ? b1 : a.blocks_per_tile;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      vec_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const vec_t*>(src + (long long)(b + u) * SLAB);
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; b < b1; ++b) sum += *reinterpret_cast<const vec_t*>(src + (long long)b * SLAB);
  }
  red[grp][el] = sum;
  __syncthreads();
  if (grp == 0 && vidx < total_vec)
    *reinterpret_cast<vec_t*>(a.DW + vidx * VEC) = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
}

}  // namespace jpdse
