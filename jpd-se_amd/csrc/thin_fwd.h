// Forward of the stride-1 convs fed by a THIN input (the 40-channel network inputs of G, the 8-channel image of
// VGG19): y[oh][ow][k] = act(bias[k] + sum_r sum_j strip_{oh+r}[ow*Cs + j] * w_r[k][j]),  j over the S*Cs run.
//
// The generic kernel materialises a padded copy and re-reads every input pixel once per tap from L2.  Here a
// block owns 8 output rows x 64 output pixels x all K (32 / 64) output channels: the 8 + R - 1 input-row strips
// ((64 + S - 1) pixels x Cs channels, dense, reflect / zero padding resolved per pixel by the DMA loader) are
// staged ONCE and stay in LDS; the filter rows w_r ([K][S*Cs] padded to a 16-multiple + 8) stream through a
// 2-stage ring.  Wave w computes one output row (x a column group of K): the A fragment of k-step ks is a plain 16-byte LDS read of the
// strip at (pix*Cs + 16*ks) -- rows overlap, pitch 80 B is conflict-free -- the B fragment a read of w_r.
#pragma once
#include "common.h"
#include "gemm_fast.h"

namespace jpdse {

struct ThinFwdArgs {
  const bf16_t* X;     // [N][H][W][Cs] unpadded
  const bf16_t* Wt;    // thin panel [R][K][KP] (KP = round_up(S*Cs, 16) + 8), zero beyond the run / logical channels
  const float* bias;
  bf16_t* Y;           // [N][OH][OW][Ks]
  int N, H, W, OH, OW, Cs, K, Ks, R, S, pad, reflect, act;
  float slope;
  int KP, ksteps;      // panel row pitch (elements), 16-wide k-steps per filter row
  int strip_units;     // 1 KiB units per strip ((64 + S - 1) * Cs * 2 bytes rounded up)
  int w_units;         // 1 KiB units per filter row panel (K * KP * 2 bytes rounded up)
  int tiles_w, tiles_h;
  float* mom;          // optional moments of y for the InstanceNorm that follows: [N][Ks][mom_slots][2], slot = tile (full tiles only,
  int mom_slots;       //   no bias / activation: the launcher checks)
};

// TH output rows per block, 8 waves = TH rows x (8 / TH) column groups of TN 32-wide tiles (K = 32 * TN * 8 / TH);
// ST = stride (the run of output pixel ow starts at input pixel ow * ST); TW = output pixels per block row (64 / 32):
// strips hold (TW - 1) * ST + S pixels
template <int TN, int TH, int ST, int TW>
__global__ __launch_bounds__(512) void thin_fwd_kernel(const ThinFwdArgs a) {
  constexpr int WN = 8 / TH, MI = TW / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  const int tw = blockIdx.x % a.tiles_w, t1 = blockIdx.x / a.tiles_w;
  const int th = t1 % a.tiles_h, n = t1 / a.tiles_h;
  const int oh0 = th * TH, ow0 = tw * TW;
  const int wrow = wid % TH, wcol = wid / TH;          // this wave's output row / column group
  const int n_strips = (TH - 1) * ST + a.R;
  const int strip_bytes = a.strip_units * 1024, w_bytes = a.w_units * 1024;
  char* const strips = smem;
  char* const wring = smem + n_strips * strip_bytes;
  const int run_px = (TW - 1) * ST + a.S;

  // ---- stage all strips: unit u of strip j; lane -> 16 bytes at byte offset b of the dense strip ----------------
  for (int u = wid; u < n_strips * a.strip_units; u += 8) {
    const int j = u / a.strip_units, uu = u - j * a.strip_units;
    const int el = uu * 512 + lane * 8;                 // element offset inside the strip
    const int px = el / a.Cs, ch = el - px * a.Cs;
    int ih = oh0 * ST + j - a.pad, iw = ow0 * ST + px - a.pad;
    bool ok = px < run_px;
    if (a.reflect) {
      ih = ih < 0 ? -ih : (ih >= a.H ? 2 * (a.H - 1) - ih : ih);
      iw = iw < 0 ? -iw : (iw >= a.W ? 2 * (a.W - 1) - iw : iw);
    }
    ok = ok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    const bf16_t* src = ok ? a.X + (((long long)n * a.H + ih) * a.W + iw) * a.Cs + ch : zero;
    glds16(src, strips + u * 1024);
  }
  auto issue_w = [&](int r, int stage) {
    const bf16_t* const wr = a.Wt + (long long)r * a.K * a.KP;
    const long long lim = (long long)a.K * a.KP;
    for (int u = wid; u < a.w_units; u += 8) {
      const long long el = (long long)u * 512 + lane * 8;
      glds16(el + 8 <= lim ? wr + el : zero, wring + stage * w_bytes + u * 1024);
    }
  };
  issue_w(0, 0);

  f32x16 acc[MI][TN];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment addressing: A row = output pixel (lane & 31) + 32*i of this wave's output row; k-half = lane >> 5
  const int a_lane = ((lane & 31) * ST * a.Cs + (lane >> 5) * 8) * 2;
  const int a_i1 = 32 * ST * a.Cs * 2;
  const int b_lane = ((wcol * TN * 32 + (lane & 31)) * a.KP + (lane >> 5) * 8) * 2;
  const int b_j1 = 32 * a.KP * 2;

  for (int r = 0; r < a.R; ++r) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                      // w_r (and, for r = 0, the strips) staged; ring slot (r+1)&1 free
    asm volatile("" ::: "memory");
    if (r + 1 < a.R) issue_w(r + 1, (r + 1) & 1);
    const char* const sa = strips + (wrow * ST + r) * strip_bytes + a_lane;
    const char* const sb = wring + (r & 1) * w_bytes + b_lane;
    __builtin_amdgcn_s_setprio(1);
    for (int ks = 0; ks < a.ksteps; ++ks) {
      s16x8 af[MI], bf[TN];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const s16x8*>(sa + i * a_i1 + ks * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const s16x8*>(sb + j * b_j1 + ks * 32);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  }

  // ---- optional moments (common.h: (mean, M2) of the bf16-rounded values about a pilot): per lane over its column's pixels,
  // lane halves, then the TH row-waves of a column group merged through LDS with Chan's formula (equal counts, fixed order)
  if (a.mom != nullptr) {
    __syncthreads();                                   // the strips are dead
    float* const red = reinterpret_cast<float*>(smem);  // [TH][WN * TN * 32][2]
    constexpr int BNC = WN * TN * 32;
    constexpr float kWaveCount = (float)(MI * 32);     // pixels per (row-wave, column)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float pilot = bf16_round(acc[0][j][0]);      // this lane's own pilot
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = bf16_round(acc[i][j][e]) - pilot;
          s1 += v;
          s2 += v * v;
        }
      float mean, m2;                                    // lanes l and l + 32 share a column: half of its pixels each
      shifted_to_mean_m2(s1, s2, pilot, 0.5f * kWaveCount, mean, m2);
      const float mean_o = __shfl_xor(mean, 32, 64), m2_o = __shfl_xor(m2, 32, 64);
      chan_merge_equal(mean, m2, mean_o, m2_o, 0.5f * kWaveCount);
      if (lane < 32) {
        const int col = wcol * TN * 32 + j * 32 + lane;
        red[(wrow * BNC + col) * 2] = mean;
        red[(wrow * BNC + col) * 2 + 1] = m2;
      }
    }
    __syncthreads();
    if (tid < BNC) {
      float msum = 0.f;
      for (int r = 0; r < TH; ++r) msum += red[(r * BNC + tid) * 2];
      const float mean = msum * (1.f / TH);
      float m2 = 0.f;
      for (int r = 0; r < TH; ++r) {
        const float dm = red[(r * BNC + tid) * 2] - mean;
        m2 += red[(r * BNC + tid) * 2 + 1] + kWaveCount * dm * dm;
      }
      if (tid < a.Ks) {
        float* const o = a.mom + (((long long)n * a.Ks + tid) * a.mom_slots + (th * a.tiles_w + tw)) * 2;
        o[0] = mean;
        o[1] = m2;
      }
    }
  }
  // ---- epilogue: wave w = output row oh0 + w; bias + activation, 16-byte stores through LDS -----------------------
  __syncthreads();
  constexpr int BN = WN * TN * 32, PITCH = BN * 2 + 64;
  acc_tile_to_lds<MI, TN>(smem, PITCH, wrow * TW, wcol * TN * 32, 0, lane, acc, a.bias, a.K, a.act, a.slope);
  __syncthreads();
  constexpr int VPR = BN / 8;
  for (int idx = tid; idx < TH * TW * VPR; idx += 512) {
    const int row = idx / VPR, v = idx - row * VPR;
    const int oh = oh0 + row / TW, ow = ow0 + row % TW;
    if (oh >= a.OH || ow >= a.OW || v * 8 >= a.Ks) continue;
    *reinterpret_cast<u32x4*>(a.Y + (((long long)n * a.OH + oh) * a.OW + ow) * a.Ks + v * 8) =
        *reinterpret_cast<const u32x4*>(smem + row * PITCH + v * 16);
  }
}

// thin panel: out[r][k][j] = w[k][r][j / Cs][j % Cs]  (0 beyond the run, for padded channels and rows k >= K)
__global__ void pack_thin_fwd_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int K, int Kp, int C, int Cs,
                                     int R, int S, int KP, long long total) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % KP);
    long long t = idx / KP;
    const int k = (int)(t % Kp);
    const int r = (int)(t / Kp);
    const int s = j / Cs, c = j - s * Cs;
    float v = 0.f;
    if (k < K && s < S && c < C) v = w[(((long long)k * R + r) * S + s) * C + c];
    out[idx] = f2bf(v);
  }
}

}  // namespace jpdse
