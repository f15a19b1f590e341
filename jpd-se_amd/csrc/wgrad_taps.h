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
};

// LDS row swizzle of the patch: a function of (row >> (ST-1)) so that the 4 rows of one transposed read
// (ST apart) land on distinct bank groups, and adding multiples of 4*ST rows leaves it unchanged
template <int ROWB, int ST> __device__ __forceinline__ int taps_swz(int row) { return trswz<ROWB>(row >> (ST - 1)); }

template <int TMW, int WM, int WN, int S, int NROW, int ST>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_taps_kernel(const TapsWgArgs a) {
  constexpr int NW = WM * WN, T = NROW * S;
  constexpr int BM = WM * TMW * 32, BN = WN * 32, BKP = 64;
  constexpr int A_ROWB = BM * 2, B_ROWB = BN * 2;
  constexpr int PW = (BKP - 1) * ST + S, NP = NROW * PW;
  constexpr int A_PPU = 1024 / A_ROWB, B_PPU = 1024 / B_ROWB;
  constexpr int A_UNITS = BKP / A_PPU, B_UNITS = (NP + B_PPU - 1) / B_PPU, UNITS = A_UNITS + B_UNITS;
  constexpr int A_STAGE = BKP * A_ROWB, STAGE = A_STAGE + B_UNITS * 1024;
  constexpr int AU = (A_UNITS + NW - 1) / NW, BU = (B_UNITS + NW - 1) / NW;
  static_assert(BM >= 64 && BN >= 64 && TMW * T <= 12, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  const uint32_t lds0 = lds_addr_of(smem);

  const int tile = blockIdx.x / a.blocks_per_tile, sub = blockIdx.x - tile * a.blocks_per_tile;
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
      const int row = (t / S) * PW + pix * ST + (t % S);
      b_tr[t] = A_STAGE + row * B_ROWB + ((((chb >> 3) ^ taps_swz<B_ROWB, ST>(row)) << 4) | ((chb & 7) << 1));
    }
  }

  // loader lane constants (32-bit element offsets; checked on the host)
  const int a_pl = lane / (A_ROWB / 16);              // pixel of the lane within a dy unit
  int a_loff[AU];
#pragma unroll
  for (int i = 0; i < AU; ++i) {
    const int pix = (wid + NW * i) * A_PPU + a_pl;
    a_loff[i] = pix * a.Ks + k0 + (((lane % (A_ROWB / 16)) ^ trswz<A_ROWB>(pix)) << 3);
  }
  int b_rj[BU], b_sw[BU];                             // (patch row << 16) | patch column; patch row 0x7fff: none
#pragma unroll
  for (int i = 0; i < BU; ++i) {
    const int u = wid + NW * i;
    const int q = u * B_PPU + lane / (B_ROWB / 16);
    b_rj[i] = ((q < NP ? q / PW : 0x7fff) << 16) | (q % PW);
    b_sw[i] = c0 + (((lane % (B_ROWB / 16)) ^ taps_swz<B_ROWB, ST>(q)) << 3);
  }
  const int row_el = a.IW * a.Cs;

  f32x16 acc[T][TMW];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][i][e] = 0.f;

  // chunks are issued in order: coordinates advance incrementally (wave-uniform, no division per chunk); padding is
  // resolved with selects -- reflected coordinates are the identity in range, so they are computed for both modes
  int i_cw = __builtin_amdgcn_readfirstlane(ch_begin % a.chunks_per_row);
  int i_row = __builtin_amdgcn_readfirstlane(ch_begin / a.chunks_per_row);       // n*OH + oh
  int i_oh = __builtin_amdgcn_readfirstlane(i_row % a.OH);
  int i_n = __builtin_amdgcn_readfirstlane(i_row / a.OH);
  const int IHm1 = a.IH - 1, IWm1 = a.IW - 1;
  const bool refl = a.reflect != 0;
  auto issue = [&](int /*chunk*/, int stage) {
    char* const st = smem + stage * STAGE;
    const int row = i_row, oh = i_oh, n = i_n;
    const int ow0 = i_cw * BKP;
    if (++i_cw == a.chunks_per_row) {
      i_cw = 0;
      ++i_row;
      if (++i_oh == a.OH) { i_oh = 0; ++i_n; }
    }
    const int px_left = a.OW - ow0;
    const bf16_t* const dy_base = a.DY + ((long long)row * a.OW + ow0) * a.Ks;      // uniform
    const bf16_t* const x_img = a.X + (long long)n * a.IH * row_el;                  // uniform
    const int ih0 = oh * ST + r0 - a.pad, iw0 = ow0 * ST - a.pad;
#pragma unroll
    for (int i = 0; i < AU; ++i) {
      if (A_UNITS % NW == 0 || wid + NW * i < A_UNITS) {
        const bool ok = (wid + NW * i) * A_PPU + a_pl < px_left;
        const bf16_t* const src = dy_base + (ok ? a_loff[i] : 0);
        glds16(ok ? src : zero, st + (wid + NW * i) * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < BU; ++i) {
      if (B_UNITS % NW == 0 || wid + NW * i < B_UNITS) {
        const int rr = b_rj[i] >> 16;
        const int ih = ih0 + rr, iw = iw0 + (b_rj[i] & 0xffff);
        int rh = ih < 0 ? -ih : ih, rw = iw < 0 ? -iw : iw;
        rh = rh > IHm1 ? 2 * IHm1 - rh : rh;
        rw = rw > IWm1 ? 2 * IWm1 - rw : rw;
        // in range after reflection?  (false for ragged chunks overhanging by more than the image, and for zero padding
        // whenever the raw coordinate is outside); r0 + rr >= R also covers the lanes beyond the patch
        const bool inr = refl ? (((unsigned)rh <= (unsigned)IHm1) & ((unsigned)rw <= (unsigned)IWm1))
                              : (((unsigned)ih < (unsigned)a.IH) & ((unsigned)iw < (unsigned)a.IW));
        const bool ok = (r0 + rr < a.R) & inr;
        const bf16_t* const src = x_img + (ok ? __mul24(rh, row_el) + __mul24(rw, a.Cs) + b_sw[i] : 0);
        glds16(ok ? src : zero, st + (A_UNITS + wid + NW * i) * 1024);
      }
    }
  };

  auto mma_step = [&]<int KS>(uint32_t sbase) {
    s16x8 af[TMW], bf[T];
#pragma unroll
    for (int i = 0; i < TMW; ++i) af[i] = tr_frag_asm<KS * 16 * A_ROWB, KS * 16 * A_ROWB + 4 * A_ROWB>(sbase + a_tr[i]);
#pragma unroll
    for (int t = 0; t < T; ++t)
      bf[t] = tr_frag_asm<KS * 16 * ST * B_ROWB, KS * 16 * ST * B_ROWB + 4 * ST * B_ROWB>(sbase + b_tr[t]);
    tr_wait(af);
    tr_wait(bf);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int i = 0; i < TMW; ++i)
        acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[t], acc[t][i], 0, 0, 0);
  };

  // 2-stage ring: the DMA of chunk c+1 flies while chunk c is consumed (the asm reads keep hipcc from draining it)
  int stage = 0;
  if (ch_begin < ch_end) issue(ch_begin, 0);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  for (int c = ch_begin; c < ch_end; ++c) {
    if (c + 1 < ch_end) issue(c + 1, stage ^ 1);
    const uint32_t sbase = lds0 + stage * STAGE;
    __builtin_amdgcn_s_setprio(1);
    mma_step.template operator()<0>(sbase);
    mma_step.template operator()<1>(sbase);
    mma_step.template operator()<2>(sbase);
    mma_step.template operator()<3>(sbase);
    __builtin_amdgcn_s_setprio(0);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    stage ^= 1;
  }

  // partial tile -> this block's slab [T][BM][BN]
  float* const slab = a.partial + (long long)blockIdx.x * (T * BM * BN);
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
