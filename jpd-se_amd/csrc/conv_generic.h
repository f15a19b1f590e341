// Generic register-staged implicit-GEMM kernels (fp32 and the bf16 layers no specialised kernel takes), the padding / fold /
// filter-packing kernels and the fixed-order slab reduction.  Part of conv_gemm.hip (one translation unit): included there, in
// order, after the kernel-argument structs.  The "row-run" formulation is described at the top of conv_gemm.hip.
#pragma once
#include <type_traits>

namespace jpdse {

// ---- MFMA over one 64-byte K chunk --------------------------------------------------------
template <typename T, int TM, int TN> struct MmaChunk;

template <int TM, int TN> struct MmaChunk<bf16_t, TM, TN> {
  __device__ static __forceinline__ void run(const char* As, const char* Bs, const int (&a_rd)[TM][2],
                                             const int (&b_rd)[TN][2], f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      s16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const s16x8*>(As + a_rd[i][u]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const s16x8*>(Bs + b_rd[j][u]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
};

template <int TM, int TN> struct MmaChunk<float, TM, TN> {
  __device__ static __forceinline__ void run(const char* As, const char* Bs, const int (&a_rd)[TM][2],
                                             const int (&b_rd)[TN][2], f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + a_rd[i][u]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bs + b_rd[j][u]);
      // lanes 0-31 carry k = 4*(2u)+q, lanes 32-63 k = 4*(2u+1)+q: any k permutation is
      // legal as long as A and B agree.
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q], bf[j][q], acc[i][j], 0, 0, 0);
    }
  }
};

// fp32 only: every kFlushChunks K-chunks the running MFMA accumulator is folded into a second
// fp32 accumulator and cleared.  A v_mfma_f32_32x32x2_f32 chain is a plain sequential fma chain
// (rounding error ~ sqrt(chain length)); two-level summation brings a K = 9216 reduction from
// ~7x torch-CPU's rounding error down to its level, which matters for fp32 parity of gradients
// through the sign()-discontinuous L1 losses.  bf16 keeps a single accumulator.
static constexpr int kFlushChunks = 16;
template <typename T, int TM, int TN> struct TwoLevel {
  __device__ static __forceinline__ void init(f32x16 (&)[TM][TN]) {}
  __device__ static __forceinline__ void flush(int, f32x16 (&)[TM][TN], f32x16 (&)[TM][TN]) {}
  __device__ static __forceinline__ void finish(f32x16 (&)[TM][TN], f32x16 (&)[TM][TN]) {}
  static constexpr int kMasters = 1;   // dummy storage
};
template <int TM, int TN> struct TwoLevel<float, TM, TN> {
  __device__ static __forceinline__ void init(f32x16 (&m)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) m[i][j][e] = 0.f;
  }
  __device__ static __forceinline__ void flush(int t, f32x16 (&acc)[TM][TN], f32x16 (&m)[TM][TN]) {
    if ((t % kFlushChunks) != kFlushChunks - 1) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          m[i][j][e] += acc[i][j][e];
          acc[i][j][e] = 0.f;
        }
  }
  __device__ static __forceinline__ void finish(f32x16 (&acc)[TM][TN], f32x16 (&m)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] += m[i][j][e];
  }
  static constexpr int kMasters = TM * TN;
};

template <int TM, int TN, int BMW, int BNW>
__device__ __forceinline__ void frag_offsets(int lane, int wm, int wn, int (&a_rd)[TM][2], int (&b_rd)[TN][2]) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * BMW + i * 32 + r;
    a_rd[i][0] = swz(row, h);
    a_rd[i][1] = swz(row, 2 + h);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * BNW + j * 32 + r;
    b_rd[j][0] = swz(row, h);
    b_rd[j][1] = swz(row, 2 + h);
  }
}

// =========================================================================================
// forward / dgrad GEMM:  Y[m][k] = act( sum_{r,j} A[m][(r,j)] * B[k][(r,j)] + bias[k] )
// =========================================================================================
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_fwd_kernel(const GemmFwdArgs a) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AV = (BM * 4 + NT - 1) / NT, BV = (BN * 4 + NT - 1) / NT;
  constexpr int ES = sizeof(T);
  static_assert((BM * 4) % NT == 0 || BM * 4 < NT, "A tile / threads");
  static_assert((BN * 4) % NT == 0 || BN * 4 < NT, "B tile / threads");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const As = smem;                // [2][BM*64]
  char* const Bs = smem + 2 * BM * 64;  // [2][BN*64]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_m = (a.M + BM - 1) / BM;
  const int tile_m = blockIdx.x % tiles_m, tile_n = blockIdx.x / tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const char* a_ptr[AV];
  int a_lds[AV];
#pragma unroll
  for (int i = 0; i < AV; ++i) {
    int v = tid + i * NT;
    v = v < BM * 4 ? v : BM * 4 - 1;  // surplus threads duplicate the last vector
    const int row = v >> 2, slot = v & 3;
    int m = m0 + row;
    m = m < a.M ? m : a.M - 1;
    const int ow = m % a.OW, t = m / a.OW, oh = t % a.OH, n = t / a.OH;
    const long long off = a.in_base + n * a.in_sn + oh * a.in_sh + ow * a.in_sw;
    a_ptr[i] = reinterpret_cast<const char*>(a.A) + off * ES + slot * 16;
    a_lds[i] = swz(row, slot);
  }
  const char* b_ptr[BV];
  int b_lds[BV];
#pragma unroll
  for (int i = 0; i < BV; ++i) {
    int v = tid + i * NT;
    v = v < BN * 4 ? v : BN * 4 - 1;
    const int row = v >> 2, slot = v & 3;
    int br = n0 + row;
    br = br < a.b_rows ? br : a.b_rows - 1;
    b_ptr[i] = reinterpret_cast<const char*>(a.B) + (long long)br * a.b_row_stride * ES + slot * 16;
    b_lds[i] = swz(row, slot);
  }

  int a_rd[TM][2], b_rd[TN][2];
  frag_offsets<TM, TN, BM / WM, BN / WN>(lane, wm, wn, a_rd, b_rd);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x16 master[(sizeof(T) == 4) ? TM : 1][(sizeof(T) == 4) ? TN : 1];
  using TL = TwoLevel<T, (sizeof(T) == 4) ? TM : 1, (sizeof(T) == 4) ? TN : 1>;
  if constexpr (sizeof(T) == 4) TL::init(master);
  const int T_all = a.R * a.cpr;
  const int split = a.splits > 1 ? (int)blockIdx.y : 0;
  const int t_begin = a.splits > 1 ? (int)((long long)T_all * split / a.splits) : 0;
  const int T_total = a.splits > 1 ? (int)((long long)T_all * (split + 1) / a.splits) : T_all;   // end of this block's chunk range
  const long long a_row_bytes = a.in_sr * ES;
  u32x4 areg[AV], breg[BV];
  int r = t_begin / a.cpr, jc = t_begin - r * a.cpr;
  // first chunk of the range
  {
    const long long a_off = (long long)r * a_row_bytes + (long long)jc * 64;
    const long long b_off = (long long)t_begin * 64;
#pragma unroll
    for (int i = 0; i < AV; ++i) areg[i] = *reinterpret_cast<const u32x4*>(a_ptr[i] + a_off);
#pragma unroll
    for (int i = 0; i < BV; ++i) breg[i] = *reinterpret_cast<const u32x4*>(b_ptr[i] + b_off);
#pragma unroll
    for (int i = 0; i < AV; ++i) *reinterpret_cast<u32x4*>(As + a_lds[i]) = areg[i];
#pragma unroll
    for (int i = 0; i < BV; ++i) *reinterpret_cast<u32x4*>(Bs + b_lds[i]) = breg[i];
  }
  __syncthreads();
  for (int t = t_begin; t < T_total; ++t) {
    const int cur = (t - t_begin) & 1;
    const bool more = (t + 1) < T_total;
    if (more) {
      if (++jc == a.cpr) { jc = 0; ++r; }
      const long long a_off = (long long)r * a_row_bytes + (long long)jc * 64;
      const long long b_off = (long long)(t + 1) * 64;
#pragma unroll
      for (int i = 0; i < AV; ++i) areg[i] = *reinterpret_cast<const u32x4*>(a_ptr[i] + a_off);
#pragma unroll
      for (int i = 0; i < BV; ++i) breg[i] = *reinterpret_cast<const u32x4*>(b_ptr[i] + b_off);
    }
    MmaChunk<T, TM, TN>::run(As + cur * BM * 64, Bs + cur * BN * 64, a_rd, b_rd, acc);
    if constexpr (sizeof(T) == 4) TL::flush(t - t_begin, acc, master);
    if (more) {
      char* const An = As + (cur ^ 1) * BM * 64;
      char* const Bn = Bs + (cur ^ 1) * BN * 64;
#pragma unroll
      for (int i = 0; i < AV; ++i) *reinterpret_cast<u32x4*>(An + a_lds[i]) = areg[i];
#pragma unroll
      for (int i = 0; i < BV; ++i) *reinterpret_cast<u32x4*>(Bn + b_lds[i]) = breg[i];
    }
    __syncthreads();
  }

  if constexpr (sizeof(T) == 4) TL::finish(acc, master);
  if (a.splits > 1) {
    // split-K: the raw fp32 partial tile into this split's slab (plain stores; gemm_splitk_finish_kernel sums the slabs)
    float* const slab = a.partial + (long long)split * a.M * a.Ks;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
      if (col >= a.Ks) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          if (m < a.M) slab[(long long)m * a.Ks + col] = acc[i][j][e];
        }
      }
    }
    return;
  }
  // ---- epilogue: bias + activation, NHWC store through the output addressing -------------
  long long* const row_off = reinterpret_cast<long long*>(smem);
  for (int row = tid; row < BM; row += NT) {
    const int m = m0 + row;
    long long off = -1;
    if (m < a.M) {
      const int ow = m % a.OW, t = m / a.OW, oh = t % a.OH, n = t / a.OH;
      off = a.out_base + n * a.out_sn + oh * a.out_sh + ow * a.out_sw;
    }
    row_off[row] = off;
  }
  __syncthreads();
  T* const Y = reinterpret_cast<T*>(a.Y);
  // the run-time activation is dispatched once around the value loops (round 4: per value it compiled to a four-way scalar branch
  // incl. the tanh expansion around each of the 64 stores, gemm_fast.h)
  auto store_tile = [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
      if (col >= a.Ks) continue;
      const int kk = a.col_mod > 0 ? col % a.col_mod : col;
      const bool live = a.col_mod > 0 ? kk < a.k_real : col < a.Kout;
      const float bv = (a.bias != nullptr && live) ? a.bias[kk] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          const long long off = row_off[row];
          if (off < 0) continue;
          const float v = live ? act_ct<ACT>(acc[i][j][e] + bv, a.slope) : 0.f;
          ElemOps<T>::st(Y + off + col, v);
        }
      }
    }
  };
  if (a.act == JPDSE_ACT_RELU) store_tile(std::integral_constant<int, JPDSE_ACT_RELU>{});
  else if (a.act == JPDSE_ACT_LRELU) store_tile(std::integral_constant<int, JPDSE_ACT_LRELU>{});
  else if (a.act == JPDSE_ACT_TANH) store_tile(std::integral_constant<int, JPDSE_ACT_TANH>{});
  else store_tile(std::integral_constant<int, JPDSE_ACT_NONE>{});
}

// Sum of the split-K slabs of gemm_fwd_kernel (index order: deterministic) + bias + activation -> output, 4 columns per
// thread (Ks is a multiple of 8).  Slabs are read four at a time before they are added (one round trip per group).
template <typename T>
__global__ __launch_bounds__(256) void gemm_splitk_finish_kernel(const GemmFwdArgs a, long long total_vec) {
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= total_vec) return;
  const int vpr = a.Ks >> 2;
  const int m = (int)(v / vpr), c0 = (int)(v % vpr) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* const base = a.partial + (long long)m * a.Ks + c0;
  const long long slab_elems = (long long)a.M * a.Ks;
  int sidx = 0;
  for (; sidx + 4 <= a.splits; sidx += 4) {
    f32x4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const f32x4*>(base + (sidx + u) * slab_elems);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += t[u];
  }
  for (; sidx < a.splits; ++sidx) acc += *reinterpret_cast<const f32x4*>(base + sidx * slab_elems);
  const int ow = m % a.OW, t = m / a.OW, oh = t % a.OH, n = t / a.OH;
  const long long off = a.out_base + n * a.out_sn + oh * a.out_sh + ow * a.out_sw;
  T* const Y = reinterpret_cast<T*>(a.Y);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int col = c0 + e;
    const bool live = col < a.Kout;
    const float bv = (a.bias != nullptr && live) ? a.bias[col] : 0.f;
    ElemOps<T>::st(Y + off + col, live ? apply_act(acc[e] + bv, a.act, a.slope) : 0.f);
  }
}

// =========================================================================================
// weight-gradient GEMM: DW[k][(r,j)] = sum_m DY[m][k] * X[rowbase(m) + r*in_sr + j]
// Both operands are transposed while being staged (pixels become the 64-byte K rows).
// =========================================================================================
template <typename T> struct WgStage;

// bf16: one item = 2 adjacent pixels x 8 channels -> 8 packed (p, p+1) dwords
template <> struct WgStage<bf16_t> {
  static constexpr int PIX = 32;
  template <int ROWS> static constexpr int items() { return (ROWS / 8) * 16; }
  struct Regs { u32x4 lo, hi; };
  template <int ROWS>
  __device__ static __forceinline__ void decode(int id, int& cg, int& pp) { cg = id % (ROWS / 8); pp = id / (ROWS / 8); }
  __device__ static __forceinline__ void write(char* lds, int cg, int pp, const Regs& rg) {
    const int slot = pp >> 2, within = (pp & 3) << 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t lo = rg.lo[e], hi = rg.hi[e];
      const uint32_t w0 = (lo & 0xffffu) | (hi << 16);
      const uint32_t w1 = (lo >> 16) | (hi & 0xffff0000u);
      const int row0 = cg * 8 + 2 * e, row1 = row0 + 1;
      *reinterpret_cast<uint32_t*>(lds + swz(row0, slot) + within) = w0;
      *reinterpret_cast<uint32_t*>(lds + swz(row1, slot) + within) = w1;
    }
  }
};

// fp32: one item = 1 pixel x 4 channels -> 4 dwords
template <> struct WgStage<float> {
  static constexpr int PIX = 16;
  template <int ROWS> static constexpr int items() { return (ROWS / 4) * 16; }
  struct Regs { u32x4 lo; };
  template <int ROWS>
  __device__ static __forceinline__ void decode(int id, int& cg, int& pp) { cg = id % (ROWS / 4); pp = id / (ROWS / 4); }
  __device__ static __forceinline__ void write(char* lds, int cg, int pp, const Regs& rg) {
    const int slot = pp >> 2, within = (pp & 3) << 2;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      *reinterpret_cast<uint32_t*>(lds + swz(cg * 4 + e, slot) + within) = rg.lo[e];
  }
};

struct PixCursor {
  int n, oh, ow;
  __device__ __forceinline__ void init(int m, int OH, int OW) {
    ow = m % OW;
    const int t = m / OW;
    oh = t % OH;
    n = t / OH;
  }
  __device__ __forceinline__ void advance(int d, int OH, int OW) {
    ow += d;
    while (ow >= OW) {
      ow -= OW;
      if (++oh == OH) { oh = 0; ++n; }
    }
  }
};

// (fp32, 128 x 128: without the second launch-bounds argument hipcc takes 208 VGPRs + 64 AGPRs = 272 registers -- ONE wave per
// SIMD, one block per CU; the ResnetBlock weight gradient of BASELINE config 2 (576 tiles of 32 chunks) then runs three rounds of
// single, latency-exposed waves: 300 us for 9.7 GFLOP.  Two waves per SIMD asked for: <= 256 registers.)
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (sizeof(T) == 4 && BM * BN >= 128 * 128) ? 2 : 1) void gemm_wgrad_kernel(const GemmWgradArgs a) {
  using ST = WgStage<T>;
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int ES = sizeof(T);
  constexpr int PIX = ST::PIX;
  constexpr int PSTEP = (ES == 2) ? 2 : 1;  // pixels per item
  constexpr int CW = 16 / ES;               // channels per item
  constexpr int AI = (ST::template items<BM>() + NT - 1) / NT;
  constexpr int BI = (ST::template items<BN>() + NT - 1) / NT;
  static_assert(ST::template items<BM>() % NT == 0 || ST::template items<BM>() < NT, "A items");
  static_assert(ST::template items<BN>() % NT == 0 || ST::template items<BN>() < NT, "B items");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const As = smem;
  char* const Bs = smem + 2 * BM * 64;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  // blockIdx.x -> (k tile, r, column tile)
  const int ct = blockIdx.x % a.col_tiles_per_r;
  const int t1 = blockIdx.x / a.col_tiles_per_r;
  const int r = t1 % a.R;
  const int kt = t1 / a.R;
  const int k0 = kt * BM, j0 = ct * BN;
  const int c_begin = blockIdx.y * a.chunks_per_split;
  int c_end = c_begin + a.chunks_per_split;
  c_end = c_end < a.chunks_total ? c_end : a.chunks_total;

  // per-item state
  bool a_on[AI], b_on[BI];
  int a_cg[AI], a_pp[AI], b_cg[BI], b_pp[BI];
  PixCursor a_cur[AI][PSTEP], b_cur[BI][PSTEP];
  int a_m[AI], b_m[BI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int id = tid + i * NT;
    a_on[i] = id < ST::template items<BM>();
    ST::template decode<BM>(a_on[i] ? id : 0, a_cg[i], a_pp[i]);
    a_m[i] = c_begin * PIX + a_pp[i] * PSTEP;
#pragma unroll
    for (int p = 0; p < PSTEP; ++p) {
      int m = a_m[i] + p;
      a_cur[i][p].init(m < a.M ? m : a.M - 1, a.OH, a.OW);
    }
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int id = tid + i * NT;
    b_on[i] = id < ST::template items<BN>();
    ST::template decode<BN>(b_on[i] ? id : 0, b_cg[i], b_pp[i]);
    b_m[i] = c_begin * PIX + b_pp[i] * PSTEP;
#pragma unroll
    for (int p = 0; p < PSTEP; ++p) {
      int m = b_m[i] + p;
      b_cur[i][p].init(m < a.M ? m : a.M - 1, a.OH, a.OW);
    }
  }
  // channel offsets (clamped so that every 16-byte load stays inside its pixel / run slack)
  int a_ch[AI], b_col[BI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int ch = k0 + a_cg[i] * CW;
    a_ch[i] = ch < a.Ks ? ch : a.Ks - CW;  // rows >= Ks are never stored
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) b_col[i] = j0 + b_cg[i] * CW;

  int a_rd[TM][2], b_rd[TN][2];
  frag_offsets<TM, TN, BM / WM, BN / WN>(lane, wm, wn, a_rd, b_rd);
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x16 master[(sizeof(T) == 4) ? TM : 1][(sizeof(T) == 4) ? TN : 1];
  using TL = TwoLevel<T, (sizeof(T) == 4) ? TM : 1, (sizeof(T) == 4) ? TN : 1>;
  if constexpr (sizeof(T) == 4) TL::init(master);
  const T* const DY = reinterpret_cast<const T*>(a.DY);
  const T* const X = reinterpret_cast<const T*>(a.X);
  typename ST::Regs areg[AI], breg[BI];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto load_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      if (!a_on[i]) continue;
#pragma unroll
      for (int p = 0; p < PSTEP; ++p) {
        const PixCursor& c = a_cur[i][p];
        const long long off = a.dy_base + c.n * a.dy_sn + c.oh * a.dy_sh + c.ow * a.dy_sw + a_ch[i];
        u32x4 v = *reinterpret_cast<const u32x4*>(DY + off);
        if (a_m[i] + p >= a.M) v = zero4;  // pixels past the end contribute nothing
        if (p == 0) areg[i].lo = v;
        else reinterpret_cast<u32x4*>(&areg[i])[PSTEP - 1] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      if (!b_on[i]) continue;
#pragma unroll
      for (int p = 0; p < PSTEP; ++p) {
        const PixCursor& c = b_cur[i][p];
        const long long off = a.in_base + c.n * a.in_sn + c.oh * a.in_sh + c.ow * a.in_sw +
                              (long long)r * a.in_sr + b_col[i];
        u32x4 v = *reinterpret_cast<const u32x4*>(X + off);
        if (p == 0) breg[i].lo = v;
        else reinterpret_cast<u32x4*>(&breg[i])[PSTEP - 1] = v;
      }
    }
  };
  auto advance = [&]() {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      a_m[i] += PIX;
#pragma unroll
      for (int p = 0; p < PSTEP; ++p)
        if (a_m[i] + p < a.M) a_cur[i][p].advance(PIX, a.OH, a.OW);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      b_m[i] += PIX;
#pragma unroll
      for (int p = 0; p < PSTEP; ++p)
        if (b_m[i] + p < a.M) b_cur[i][p].advance(PIX, a.OH, a.OW);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AI; ++i)
      if (a_on[i]) ST::write(As + buf * BM * 64, a_cg[i], a_pp[i], areg[i]);
#pragma unroll
    for (int i = 0; i < BI; ++i)
      if (b_on[i]) ST::write(Bs + buf * BN * 64, b_cg[i], b_pp[i], breg[i]);
  };

  if (c_begin < c_end) {
    load_chunk();
    store_chunk(0);
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int cur = (c - c_begin) & 1;
    const bool more = (c + 1) < c_end;
    if (more) {
      advance();
      load_chunk();
    }
    MmaChunk<T, TM, TN>::run(As + cur * BM * 64, Bs + cur * BN * 64, a_rd, b_rd, acc);
    if constexpr (sizeof(T) == 4) TL::flush(c - c_begin, acc, master);
    if (more) store_chunk(cur ^ 1);
    __syncthreads();
  }
  if constexpr (sizeof(T) == 4) TL::finish(acc, master);

  // ---- epilogue: scatter the (k, r, j) tile into the fp32 KRSC master-layout gradient (or this split's slab) -----
  float* const out = a.partial != nullptr ? a.partial + (long long)blockIdx.y * a.slab_stride : a.DW;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = j0 + wn * (BN / WN) + j * 32 + (lane & 31);
    if (col >= a.run) continue;
    const int s = col / a.Cs, cc = col - s * a.Cs;
    if (cc >= a.C) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = k0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (k >= a.K) continue;
        out[(((long long)k * a.R + r) * a.S + s) * a.C + cc] = acc[i][j][e];
      }
    }
  }
}

// =========================================================================================
// padding (materialises the padded NHWC input + zeroed slack) and its reflect adjoint
// =========================================================================================
// One block per padded row (n, hp): the source row is resolved once, lanes own a channel-vector
// column and walk the row's pixels -- no integer division in the copy loop.
template <typename T>
__global__ __launch_bounds__(256) void pad_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int H, int W,
                                                 int Cs, int pt, int pl, int Hp, int Wp, int mode, int tx_shift,
                                                 long long total_vec, long long slack_vec) {
  constexpr int VE = 16 / sizeof(T);
  const int cv = Cs / VE;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const int TX = 1 << tx_shift, TY = 256 >> tx_shift;
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x >> tx_shift;
  const int rows = N * Hp;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / Hp, hp = row - n * Hp;
    int h = hp - pt;
    bool row_ok = true;
    if (mode == JPDSE_PAD_REFLECT) h = h < 0 ? -h : (h >= H ? 2 * (H - 1) - h : h);
    else row_ok = (h >= 0) & (h < H);
    const T* srow = src + ((long long)n * H + (row_ok ? h : 0)) * W * Cs;
    T* drow = dst + (long long)row * Wp * Cs;
    // gridDim.y column segments per row (small tensors: N * Hp rows alone leave most CUs idle -- 18 blocks for a 16 x 32 map)
    const int segw = (Wp + (int)gridDim.y - 1) / (int)gridDim.y;
    const int w_lo = (int)blockIdx.y * segw, w_hi = w_lo + segw < Wp ? w_lo + segw : Wp;
    for (int c = tx; c < cv; c += TX) {
      for (int wp = w_lo + ty; wp < w_hi; wp += TY) {
        int w = wp - pl;
        bool ok = row_ok;
        if (mode == JPDSE_PAD_REFLECT) w = w < 0 ? -w : (w >= W ? 2 * (W - 1) - w : w);
        else ok = ok & (w >= 0) & (w < W);
        u32x4 v = zero4;
        if (ok) v = *reinterpret_cast<const u32x4*>(srow + (long long)w * Cs + c * VE);
        *reinterpret_cast<u32x4*>(drow + (long long)wp * Cs + c * VE) = v;
      }
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0)
    for (long long i = threadIdx.x; i < slack_vec; i += 256) *reinterpret_cast<u32x4*>(dst + (total_vec + i) * VE) = zero4;
}

// dx[h][w] = sum over the padded-domain aliases of (h,w) of dxp (adjoint of ReflectionPad2d(p))
template <typename T>
__global__ void reflect_fold_kernel(const T* __restrict__ dxp, T* __restrict__ dx, int N, int H, int W,
                                    int Cs, int p, long long total_vec) {
  constexpr int VE = 16 / sizeof(T);
  const int cv = Cs / VE;
  const int Hp = H + 2 * p, Wp = W + 2 * p;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total_vec;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv);
    long long t = idx / cv;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    int hs[3], ws[3], nh = 0, nw = 0;
    hs[nh++] = h + p;
    if (h >= 1 && h <= p) hs[nh++] = p - h;
    if (h <= H - 2 && h >= H - 1 - p) hs[nh++] = p + 2 * (H - 1) - h;
    ws[nw++] = w + p;
    if (w >= 1 && w <= p) ws[nw++] = p - w;
    if (w <= W - 2 && w >= W - 1 - p) ws[nw++] = p + 2 * (W - 1) - w;
    float accv[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) accv[e] = 0.f;
    for (int a = 0; a < nh; ++a)
      for (int b = 0; b < nw; ++b) {
        float v[VE];
        Vec16<T>::load(dxp + (((long long)n * Hp + hs[a]) * Wp + ws[b]) * Cs + c * VE, v);
#pragma unroll
        for (int e = 0; e < VE; ++e) accv[e] += v[e];
      }
    Vec16<T>::store(dx + idx * VE, accv);
  }
}

// =========================================================================================
// filter packing: fp32 KRSC master -> compute-dtype GEMM panels
// =========================================================================================
template <typename T>
__global__ void pack_fwd_kernel(const float* __restrict__ w, T* __restrict__ out, int K, int Ks, int C, int Cs,
                                int R, int S, int Lk, long long total) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % Lk);
    long long t = idx / Lk;
    const int r = (int)(t % R);
    const int k = (int)(t / R);
    const int s = j / Cs, c = j - s * Cs;
    float v = 0.f;
    if (k < K && s < S && c < C) v = w[(((long long)k * R + r) * S + s) * C + c];
    ElemOps<T>::st(out + idx, v);
  }
}

// Toeplitz panel of a stride-1 conv with <= 8 output channels (the 64->3 / 32->3 heads): GEMM column
// (dl, k) = output pixel ow4*4 + dl, channel k, so that a 32-wide MFMA tile carries 4 pixels x 8 channels
// instead of 8 channels + 24 dead columns; K-dim = (r, s', c) over the S+3 input pixels the 4 outputs share:
//   out[(dl*8 + k)][r][s'*Cs + c] = w[k][r][s' - dl][c]   (0 outside the filter)
template <typename T>
__global__ void pack_fwd_toep_kernel(const float* __restrict__ w, T* __restrict__ out, int K, int C, int Cs,
                                     int R, int S, int Lk, long long total) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % Lk);
    long long t = idx / Lk;
    const int r = (int)(t % R);
    const int row = (int)(t / R);
    const int dl = row >> 3, k = row & 7;
    const int sp = j / Cs, c = j - sp * Cs, s = sp - dl;
    float v = 0.f;
    if (k < K && s >= 0 && s < S && c < C) v = w[(((long long)k * R + r) * S + s) * C + c];
    ElemOps<T>::st(out + idx, v);
  }
}

// data-gradient panel of a stride phase: rows = input channels c, K-dim = (u', w', k)
// the stride phases of one layer in ONE launch (a stride-2 layer has four: PatchGAN layer 0 launched 16 of these 5-us kernels per step)
struct PackPhaseTable {
  long long first[5];          // element range [first[i], first[i + 1]) of the launch belongs to phase i
  long long out_off[4];        // element offset of the phase's panel in `out`
  int qh[4], qw[4], Uh[4], Uw[4], Lk[4];
  int n;
};
template <typename T>
__global__ void pack_dgrad_phases_kernel(const float* __restrict__ w, T* __restrict__ out, int K, int Ks, int C, int Cs, int R,
                                         int S, int st, const PackPhaseTable tab) {
  const long long total = tab.first[tab.n];
  for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    int ph = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (i < tab.n && g >= tab.first[i]) ph = i;
    const long long idx = g - tab.first[ph];
    const int Lk = tab.Lk[ph], Uh = tab.Uh[ph], Uw = tab.Uw[ph];
    const int j = (int)(idx % Lk);
    long long t = idx / Lk;
    const int up = (int)(t % Uh);
    const int c = (int)(t / Uh);
    const int wp = j / Ks, k = j - wp * Ks;
    float v = 0.f;
    if (c < C && k < K && wp < Uw) {
      const int r = tab.qh[ph] + st * (Uh - 1 - up), s = tab.qw[ph] + st * (Uw - 1 - wp);
      v = w[(((long long)k * R + r) * S + s) * C + c];
    }
    ElemOps<T>::st(out + tab.out_off[ph] + idx, v);
  }
}

// Fast paths (channel counts that are multiples of 8, i.e. every layer but the network inputs):
// the forward panel is then the plain compute-dtype cast of the KRSC master (8 elements per lane), and
// a data-gradient panel is a [k][c] -> [c][k] transpose per filter tap, done through an LDS tile so
// that both the fp32 reads (along c) and the 16-bit writes (along k) are coalesced.
template <typename T>
__global__ void pack_fwd_cast_kernel(const float* __restrict__ w, T* __restrict__ out, long long total8) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total8;
       idx += (long long)gridDim.x * blockDim.x) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(w + idx * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(w + idx * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ElemOps<T>::st(out + idx * 8 + e, a[e]);
      ElemOps<T>::st(out + idx * 8 + 4 + e, b[e]);
    }
  }
}

// out[c][up][wp*Ks + k] = w[k][r(up)][s(wp)][c];  grid = (c tiles of 64, k tiles of 64, Uh*Uw taps)
template <typename T>
__global__ __launch_bounds__(256) void pack_dgrad_tile_kernel(const float* __restrict__ w, T* __restrict__ out, int K,
                                                             int Ks, int C, int Cs, int R, int S, int st, int qh,
                                                             int qw, int Uh, int Uw, int Lk) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int up = blockIdx.z / Uw, wp = blockIdx.z % Uw;
  const int r = qh + st * (Uh - 1 - up), s = qw + st * (Uw - 1 - wp);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  for (int kk = ty; kk < 64; kk += 4) {
    const int k = k0 + kk, c = c0 + tx;
    tile[kk][tx] = (k < K && c < C) ? w[(((long long)k * R + r) * S + s) * C + c] : 0.f;
  }
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc, k = k0 + tx;
    if (c < Cs && k < Ks) ElemOps<T>::st(out + ((long long)c * Uh + up) * Lk + wp * Ks + k, tile[tx][cc]);
  }
}

// The same transpose for MANY layers / stride phases in one launch (per-layer launches of 8-18 us each added up
// to 1 ms per optimizer step): block -> table entry by binary search on block0, as in adam_kernel.
__global__ __launch_bounds__(256) void pack_dgrad_tile_many_kernel(const jpdse_pack_entry* __restrict__ table, int n) {
  __shared__ float tile[64][65];
  const long long blk = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block0 <= blk) lo = mid;
    else hi = mid - 1;
  }
  const jpdse_pack_entry e = table[lo];
  const int id = (int)(blk - e.block0);
  const int bx = id % e.gx, by = (id / e.gx) % e.gy, bz = id / (e.gx * e.gy);
  const int c0 = bx * 64, k0 = by * 64;
  const int up = bz / e.Uw, wp = bz % e.Uw;
  const int r = e.qh + e.st * (e.Uh - 1 - up), s = e.qw + e.st * (e.Uw - 1 - wp);
  const int t = threadIdx.x;
  if ((e.C & 3) == 0) {
    // 16-byte loads along c (4 per thread), 16-byte stores along k (2 per thread): a quarter of the memory instructions
    // of the scalar form below (0.34 -> 0.2 ms for the generator's 182 M weights)
    const int cq = (t & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = (t >> 4) + 16 * i;
      const int k = k0 + kk, c = c0 + cq;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (k < e.K && c < e.C) v = *reinterpret_cast<const f32x4*>(e.w + (((long long)k * e.R + r) * e.S + s) * e.C + c);
      tile[kk][cq] = v[0];
      tile[kk][cq + 1] = v[1];
      tile[kk][cq + 2] = v[2];
      tile[kk][cq + 3] = v[3];
    }
    __syncthreads();
    const int kq = (t & 7) * 8;
    if (e.out_f32) {                                    // fp32 panels (JPDSE_F32 layers): the same transpose, two 16-byte stores
      float* const outf = reinterpret_cast<float*>(e.out);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int cc = (t >> 3) + 32 * i;
        const int c = c0 + cc, k = k0 + kq;
        if (c < e.Cs && k < e.Ks) {
          float* const dst = outf + ((long long)c * e.Uh + up) * e.Lk + wp * e.Ks + k;
          const f32x4 lo = {tile[kq][cc], tile[kq + 1][cc], tile[kq + 2][cc], tile[kq + 3][cc]};
          const f32x4 hi = {tile[kq + 4][cc], tile[kq + 5][cc], tile[kq + 6][cc], tile[kq + 7][cc]};
          *reinterpret_cast<f32x4*>(dst) = lo;
          *reinterpret_cast<f32x4*>(dst + 4) = hi;
        }
      }
      return;
    }
    bf16_t* const out = reinterpret_cast<bf16_t*>(e.out);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int cc = (t >> 3) + 32 * i;
      const int c = c0 + cc, k = k0 + kq;
      if (c < e.Cs && k < e.Ks) {                       // Ks % 8 == 0: the 8 channels are in range together
        u32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          pk[j] = (uint32_t)f2bf(tile[kq + 2 * j][cc]) | ((uint32_t)f2bf(tile[kq + 2 * j + 1][cc]) << 16);
        // next read in the NEXT step's backward: nontemporal
        __builtin_nontemporal_store(pk, reinterpret_cast<u32x4*>(out + ((long long)c * e.Uh + up) * e.Lk + wp * e.Ks + k));
      }
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int kk = ty; kk < 64; kk += 4) {
    const int k = k0 + kk, c = c0 + tx;
    tile[kk][tx] = (k < e.K && c < e.C) ? e.w[(((long long)k * e.R + r) * e.S + s) * e.C + c] : 0.f;
  }
  __syncthreads();
  bf16_t* const out = reinterpret_cast<bf16_t*>(e.out);
  float* const outf = reinterpret_cast<float*>(e.out);
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc, k = k0 + tx;
    if (c < e.Cs && k < e.Ks) {
      const long long o = ((long long)c * e.Uh + up) * e.Lk + wp * e.Ks + k;
      if (e.out_f32) outf[o] = tile[tx][cc];
      else out[o] = f2bf(tile[tx][cc]);
    }
  }
}

// dw[i] = sum over b < nslabs of partial[b * stride + i], slabs added in index order (deterministic).  256 threads =
// 64 consecutive vectors x 4 slab groups, 8 loads in flight per thread; the groups are combined through LDS in a fixed
// order.  VEC = 4 when the element count and the slab stride are multiples of 4, else 1.
template <int VEC>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                         long long nvec, long long stride_vec, int nslabs) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  __shared__ vec_t red[4][64];
  const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long long idx = (long long)blockIdx.x * 64 + el;
  vec_t sum = {};
  if (idx < nvec) {
    const vec_t* src = reinterpret_cast<const vec_t*>(partial) + idx;
    const int per = (nslabs + 3) / 4;
    const int b0 = grp * per;
    int b1 = b0 + per;
    b1 = b1 < nslabs ? b1 : nslabs;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      vec_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(long long)(b + u) * stride_vec];
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; b < b1; ++b) sum += src[(long long)b * stride_vec];
  }
  red[grp][el] = sum;
  __syncthreads();
  if (grp == 0 && idx < nvec) reinterpret_cast<vec_t*>(dw)[idx] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
}

static int launch_slab_reduce(const float* partial, float* dw, long long n, long long stride, int nslabs, hipStream_t s) {
  if (n % 4 == 0 && stride % 4 == 0) {
    const long long nv = n / 4;
    hipLaunchKernelGGL((slab_reduce_kernel<4>), dim3((unsigned)((nv + 63) / 64)), dim3(256), 0, s, partial, dw, nv, stride / 4, nslabs);
  } else {
    hipLaunchKernelGGL((slab_reduce_kernel<1>), dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, partial, dw, n, stride, nslabs);
  }
  return jpdse::check_launch("slab_reduce_kernel");
}

}  // namespace jpdse
