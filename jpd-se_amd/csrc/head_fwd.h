// Forward of the image heads (ReflectionPad2d(3) + Conv2d(64|32 -> 3, 7x7) + Tanh, networks.py:148-152,243-246):
// very few output channels K, many taps.  As a direct implicit GEMM 29 of 32 MFMA columns are dead and every
// input pixel is re-read once per tap (the generic kernel ran it at 40 TFLOP/s; a Toeplitz arrangement got
// it to L2-bound).  Here the conv is factored through the taps:
//     Z[q][(r,k,s)] = sum_c x[q][c] * w[k][r][s][c]           (a GEMM with N = R*K*S = 147 live columns, K-dim = C)
//     y[oh][ow][k]  = bias[k] + sum_{r,s} Z[(oh + r, ow + s)][(r,k,s)]
// A block owns TH output rows x 64 output pixels.  It walks over the TH + R - 1 input rows of its band; per
// input row the 64 + S - 1 input pixels (reflect / zero padding resolved per pixel by the DMA loader) are staged
// once, one MFMA pass produces the row's Z tile (each wave = one 32-column tile of Z, 3 m-tiles), the tile goes to
// LDS column-major, and the thread that owns output (px, k) adds its R partial sums into the fp32 output
// tile in LDS (owner-computes: no atomics, deterministic).  Every input pixel is read ~(TH+R-1)/TH times.
#pragma once
#include "common.h"
#include "gemm_fast.h"

namespace jpdse {

// LDS stores as inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of every ordinary LDS STORE issued while
// an LDS-DMA is in flight (it cannot prove that the two do not overlap), which drained the input-row prefetch
// at the first Z store of every row.  The asm form is ordered by hand (lgkmcnt(0) + barrier before any reader).
__device__ __forceinline__ void lds_store128(uint32_t addr, f32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_store32(uint32_t addr, float v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr32(const void* p) {
  return (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)p);
}

struct HeadFwdArgs {
  const bf16_t* X;     // [N][H][W][CIN] unpadded
  const bf16_t* Wp;    // plain forward panel [Ks][R][S*CIN] (row k: (r, s, c) contiguous)
  const float* bias;
  bf16_t* Y;           // [N][OH][OW][Ks_out]
  int N, H, W, OH, OW, K, Ks_out, R, S, pad, reflect, act;
  float slope;
  int tiles_w, tiles_h;
};

constexpr int kHeadTH = 16;       // output rows per block
constexpr int kHeadMR = 96;       // staged input pixels per row (3 m-tiles >= 64 + S - 1)
constexpr int kHeadZP = 100;      // Z column pitch in floats (96 + 4: conflict-free 16-byte column stores)

// FR, FS: filter size as compile-time constants (the owner loop must be fully unrolled: with runtime bounds every
// one of its 49 LDS reads exposed its latency)
template <int CIN, int NT, int FR, int FS>        // NT = 32-column tiles of Z = waves per block
__global__ __launch_bounds__(64 * NT) void head_fwd_kernel(const HeadFwdArgs a) {
  constexpr int ROWB = CIN * 2;                         // 128 or 64 bytes per pixel / weight row
  constexpr int SLOTS = ROWB / 16;                      // 16-byte slots per row
  constexpr int SH = CIN == 64 ? 1 : 2;                 // swizzle: slot ^= (row >> SH) & (SLOTS - 1)
  constexpr int RPU = 1024 / ROWB;                      // rows per 1 KiB DMA unit
  constexpr int NCOLS = NT * 32;
  constexpr int B_BYTES = NCOLS * ROWB, A_BYTES = kHeadMR * ROWB;
  constexpr int Z_BYTES = NCOLS * kHeadZP * 4, O_BYTES = kHeadTH * 64 * 4 * 4;
  constexpr int KSTEPS = CIN / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Bs = smem;                                // weights [NCOLS][ROWB], swizzled
  char* const As = smem + B_BYTES;                      // 3 x [96][ROWB], swizzled (ring)
  float* const Zs = reinterpret_cast<float*>(smem + B_BYTES + 3 * A_BYTES);           // [NCOLS][96]
  float* const Os = reinterpret_cast<float*>(smem + B_BYTES + 3 * A_BYTES + Z_BYTES);  // [TH][64][4]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  const uint32_t zs0 = lds_addr32(Zs), os0 = lds_addr32(Os);
  const int tw = blockIdx.x % a.tiles_w, t1 = blockIdx.x / a.tiles_w;
  const int th = t1 % a.tiles_h, n = t1 / a.tiles_h;
  const int oh0 = th * kHeadTH, ow0 = tw * 64;
  const int ncols = a.R * a.K * a.S;                    // live columns, ordered (r, k, s)
  const int KS = a.K * a.S;

  // A zero the compiler cannot fold: the accumulators pass through a VALU add before the asm LDS stores, so that
  // hipcc inserts the MFMA -> VALU wait states itself (it does not know that the asm reads MFMA results as LDS
  // store data: without this the last m-tile was stored before its MFMAs had retired).  Loaded and pinned
  // before any DMA is issued.
  float fz = reinterpret_cast<const float*>(g_zero_page)[0];
  asm volatile("" : "+v"(fz));
  for (int i = tid; i < kHeadTH * 64 * 4; i += 64 * NT) Os[i] = 0.f;     // before any DMA is in flight (see lds_store)
  // ---- weights: LDS row col = (r, k, s) <- panel row k, offset (r*S + s)*CIN -----------------
  for (int u = wid; u < NCOLS / RPU; u += NT) {
    const int row = u * RPU + lane / SLOTS, slot = lane % SLOTS;
    const bf16_t* src = zero;
    if (row < ncols) {
      const int r = row / KS, rem = row - r * KS, k = rem / a.S, s = rem - k * a.S;
      src = a.Wp + ((long long)(k * a.R + r) * a.S + s) * CIN + ((slot ^ (row >> SH)) & (SLOTS - 1)) * 8;
    }
    glds16(src, Bs + u * 1024);
  }
  // fragment read offsets: A rows = pixels (3 m-tiles), B rows = this wave's 32 columns
  int a_rd[3][KSTEPS], b_rd[KSTEPS];
  {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int brow = wid * 32 + r;
      b_rd[ks] = brow * ROWB + (((2 * ks + h) ^ (brow >> SH)) & (SLOTS - 1)) * 16;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int arow = i * 32 + r;
        a_rd[i][ks] = arow * ROWB + (((2 * ks + h) ^ (arow >> SH)) & (SLOTS - 1)) * 16;
      }
    }
  }
  // owner of output (px, k): threads 0..191 (px = t % 64, k = t / 64 < 3 .. K)
  const int o_px = tid & 63, o_k = tid >> 6;
  const bool owner = o_k < a.K;

  const int n_rows = kHeadTH + a.R - 1;                 // input rows of the band (padded coordinates oh0 + j)
  auto issue = [&](int j, int stage) {
    int ih = oh0 + j - a.pad;
    bool row_ok = true;
    if (a.reflect) ih = ih < 0 ? -ih : (ih >= a.H ? 2 * (a.H - 1) - ih : ih);
    else row_ok = (unsigned)ih < (unsigned)a.H;
    row_ok = row_ok && (unsigned)ih < (unsigned)a.H;    // bands hanging over the image bottom
    const bf16_t* const xrow = a.X + ((long long)n * a.H + (row_ok ? ih : 0)) * a.W * CIN;
    for (int u = wid; u < kHeadMR / RPU; u += NT) {
      const int q = u * RPU + lane / SLOTS, slot = lane % SLOTS;
      int iw = ow0 - a.pad + q;
      bool ok = row_ok && q < 64 + a.S - 1;
      if (a.reflect) iw = iw < 0 ? -iw : (iw >= a.W ? 2 * (a.W - 1) - iw : iw);
      ok = ok && (unsigned)iw < (unsigned)a.W;
      const bf16_t* src = ok ? xrow + (long long)iw * CIN + ((slot ^ (q >> SH)) & (SLOTS - 1)) * 8 : zero;
      glds16(src, As + stage * A_BYTES + u * 1024);
    }
  };

  // 3-stage ring over the input rows with counted vmcnt (a 2-stage ring was bound by the DMA latency: one block
  // per CU, ~2 us per row).  DMA instructions per row and wave: U1 for the first EXTRA waves, U0 for the others.
  constexpr int UNITS = kHeadMR / RPU, U0 = UNITS / NT, U1 = (UNITS + NT - 1) / NT, EXTRA = UNITS % NT;
  issue(0, 0);
  if (n_rows > 1) issue(1, 1);
  for (int j = 0; j < n_rows; ++j) {
    const int stage = j % 3;
    if (j + 1 < n_rows) {
      if (wid < EXTRA) wait_vmcnt<U1>(); else wait_vmcnt<U0>();
    } else {
      wait_vmcnt<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // row j staged by every wave; Z / O updates of row j-1 finished
    asm volatile("" ::: "memory");                      // the raw barrier is no compiler fence: keep the LDS reads below it
    if (j + 2 < n_rows) issue(j + 2, (j + 2) % 3);
    // ---- Z tile of this input row: [96 pixels] x [this wave's 32 columns]
    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const char* const at = As + stage * A_BYTES;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const s16x8 bf = *reinterpret_cast<const s16x8*>(Bs + b_rd[ks]);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const s16x8 af = *reinterpret_cast<const s16x8*>(at + a_rd[i][ks]);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[i], 0, 0, 0);
      }
    }
    // column-major Z: Zs[col][q]; a lane's 4 consecutive accumulator rows are 4 consecutive q
    {
      const int col = wid * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int q = i * 32 + 8 * g + 4 * (lane >> 5);
          f32x4 v = {acc[i][4 * g] + fz, acc[i][4 * g + 1] + fz, acc[i][4 * g + 2] + fz, acc[i][4 * g + 3] + fz};
          lds_store128(zs0 + (col * kHeadZP + q) * 4, v);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // raw: the next row's DMA stays in flight
    asm volatile("" ::: "memory");
    // ---- owner-computes: output row jr = j - r gets sum_s Z[(r, k, s)][px + s]
    if (owner) {
      float part[FR];
#pragma unroll
      for (int r = 0; r < FR; ++r) {
        const float* zc = Zs + ((r * a.K + o_k) * FS) * kHeadZP + o_px;
        float v[FS];
#pragma unroll
        for (int s = 0; s < FS; ++s) v[s] = zc[s * kHeadZP + s];
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < FS; ++s) sum += v[s];
        part[r] = sum;
      }
      // read-modify-write of the owned outputs: all loads first (an asm store is a compiler memory barrier)
      float cur[FR];
#pragma unroll
      for (int r = 0; r < FR; ++r) {
        const int jr = j - r;
        const bool ok = jr >= 0 && jr < kHeadTH;
        cur[r] = Os[((ok ? jr : 0) * 64 + o_px) * 4 + o_k];
      }
#pragma unroll
      for (int r = 0; r < FR; ++r) {
        const int jr = j - r;
        if (jr >= 0 && jr < kHeadTH) lds_store32(os0 + (((jr * 64 + o_px) * 4 + o_k) * 4), cur[r] + part[r]);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- epilogue: bias + activation, 8 (Ks_out) channels per pixel, padding lanes zero
  for (int i = tid; i < kHeadTH * 64; i += 64 * NT) {
    const int jr = i >> 6, px = i & 63;
    const int oh = oh0 + jr, ow = ow0 + px;
    if (oh >= a.OH || ow >= a.OW) continue;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = 0.f;
      if (k < a.K && k < 4) t = apply_act(Os[i * 4 + k] + (a.bias != nullptr ? a.bias[k] : 0.f), a.act, a.slope);
      v[k] = t;
    }
    Vec16<bf16_t>::store(a.Y + (((long long)n * a.OH + oh) * a.OW + ow) * a.Ks_out, v);
  }
}

}  // namespace jpdse
