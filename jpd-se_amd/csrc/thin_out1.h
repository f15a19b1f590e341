// Convolutions with ONE output channel at stride 1 -- the 4x4 512 -> 1 map that ends each PatchGAN scale
// (networks.py:447-449) -- backward.  Their data gradient writes a wide tensor from a 1-channel one and their weight
// gradient reduces a wide tensor against a 1-channel one: 16 MACs per element of the wide tensor, i.e. bound by that
// tensor's bytes, not by arithmetic.  Round 1-3 ran them as GEMMs with 7 of 8 (padded) output channels dead: a padded copy of
// dy + the generic kernel for the data gradient (60 us for a 70 MB result), a padded copy of x + a tap-expanded dy + a 1x1
// weight-gradient GEMM for the weight gradient (104 us for a 70 MB operand).  Here both are plain fp32-FMA kernels that touch
// the wide tensor once:
//   thin1_dgrad_kernel   a thread owns 8 channels (its 16 x 8 filter taps live in registers) and walks over pixels:
//                        dx[p][c] = sum_t w[t][c] * dy[p + off_t], fan-in addend in the same pass
//   thin1_wgrad_kernel   a thread owns (tap, 8 channels); a block = 16 taps x 128 channels walks over row segments of x staged
//                        once in LDS as fp32 (next segment prefetched into registers), dy rows zero-extended in LDS;
//                        one partial [16][Cs] per block strip -> fp32 slabs -> slab_reduce_kernel (fixed order)
#pragma once
#include "common.h"

namespace jpdse {

struct Thin1DgradArgs {
  const bf16_t* DY;      // [N][OH][OW][8], channel 0 live
  const bf16_t* B;       // single-phase data-gradient panel: B[c][u][v][k] at c * b_stride + u * Lk + v * 8 + k
  float* taps;           // workspace [R*S][Cs] fp32: the live column (k = 0) of the panel, tap-major (thin1_taps_kernel)
  bf16_t* DX;            // [N][H][W][Cs]
  const bf16_t* addend;  // optional, DX's addressing
  int N, H, W, Cs, OH, OW;
  int py, px;            // dx[ih][iw] = sum_{u,v} B[c][u][v][0] * dy[ih - py + u][iw - px + v]
  long long b_stride;
  int Lk;
  int nseg, segw;        // a row of dx = nseg runs of segw pixels; one run = one work item of a pixel lane
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kThin1Run = 64;     // longest run of a row one work item covers

// taps[t][c] = B[c][u][v][0], t = u * S + v: gathered once per call so that the blocks of thin1_dgrad_kernel load their filter
// registers as coalesced 32-byte pieces (gathered per block straight from the panel -- 128 two-byte loads per thread, every lane
// on its own cache line -- the load took 30 us of a 60 us launch)
__global__ __launch_bounds__(256) void thin1_taps_kernel(const bf16_t* __restrict__ B, float* __restrict__ taps, int Cs, long long b_stride,
                                                         int Lk, int S, int ntaps) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ntaps * Cs) return;
  const int t = i / Cs, c = i - t * Cs;
  taps[i] = bf2f(B[(long long)c * b_stride + (t / S) * Lk + (t % S) * 8]);
}

// 256 threads = PL pixel lanes x (Cs / 8) channel groups.  A pixel lane walks along a run of one image row: the R x S window of
// dy slides by one column per pixel (R new values per pixel instead of R * S), the 8-channel results are stored as one 16-byte
// vector -- the channel groups of a lane write one contiguous pixel.  The run's dy window (R rows x run + S - 1 columns) is
// staged in LDS first: with the loads inside the pixel loop every pixel paid a global round trip behind the previous store
// (two waves per SIMD: the 128 filter registers), 100 us for a 70 MB result; fan-in operands are loaded four pixels ahead.
template <int R, int S>
__global__ __launch_bounds__(256) void thin1_dgrad_kernel(const Thin1DgradArgs a) {
  __shared__ float dyw[32][R][kThin1Run + S];
  const int cv = a.Cs >> 3;                       // 8-channel groups: a power of two in [8, 256] (checked by the launcher)
  const int cg = threadIdx.x & (cv - 1);
  const int pl = threadIdx.x / cv, PL = 256 / cv;
  f32x2 w[R * S][4];
#pragma unroll
  for (int t = 0; t < R * S; ++t) {
    const float4 lo = *reinterpret_cast<const float4*>(a.taps + t * a.Cs + cg * 8);
    const float4 hi = *reinterpret_cast<const float4*>(a.taps + t * a.Cs + cg * 8 + 4);
    w[t][0] = (f32x2){lo.x, lo.y};
    w[t][1] = (f32x2){lo.z, lo.w};
    w[t][2] = (f32x2){hi.x, hi.y};
    w[t][3] = (f32x2){hi.z, hi.w};
  }
  const int items = a.N * a.H * a.nseg;
  const int rounds = (items + gridDim.x * PL - 1) / (gridDim.x * PL);      // the same trip count for every lane: barriers inside
  for (int k = 0; k < rounds; ++k) {
    const int it = (k * gridDim.x + blockIdx.x) * PL + pl;
    const bool live = it < items;
    const int itc = live ? it : 0;
    const int seg = itc % a.nseg, row = itc / a.nseg;
    const int ih = row % a.H, n = row / a.H;
    const int w0 = seg * a.segw;
    int wn = a.W - w0;
    wn = wn < a.segw ? wn : a.segw;
    wn = live ? wn : 0;
    __syncthreads();                              // the previous run's window reads are done
    for (int j = cg; j < R * (kThin1Run + S); j += cv) {
      const int u = j / (kThin1Run + S), c = j - u * (kThin1Run + S);
      const int oh = ih - a.py + u, ow = w0 - a.px + c;
      const bool ok = live && c < wn + S - 1 && ((unsigned)oh < (unsigned)a.OH) & ((unsigned)ow < (unsigned)a.OW);
      dyw[pl][u][c] = ok ? bf2f(a.DY[(((long long)n * a.OH + oh) * a.OW + ow) * 8]) : 0.f;
    }
    __syncthreads();
    float win[S][R];                              // win[v][u] = dy[ih - py + u][iw - px + v]
#pragma unroll
    for (int v = 1; v < S; ++v)
#pragma unroll
      for (int u = 0; u < R; ++u) win[v][u] = dyw[pl][u][v - 1];
    bf16_t* __restrict__ out = a.DX + (((long long)n * a.H + ih) * a.W + w0) * a.Cs + cg * 8;
    const bf16_t* __restrict__ add = a.addend != nullptr ? a.addend + (out - a.DX) : nullptr;
    for (int i0 = 0; i0 < wn; i0 += 4) {
      u32x4 adv[4];
      if (add != nullptr) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (i0 + q < wn) adv[q] = *reinterpret_cast<const u32x4*>(add + (long long)(i0 + q) * a.Cs);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + q;
        if (i >= wn) break;
#pragma unroll
        for (int v = 0; v + 1 < S; ++v)
#pragma unroll
          for (int u = 0; u < R; ++u) win[v][u] = win[v + 1][u];
#pragma unroll
        for (int u = 0; u < R; ++u) win[S - 1][u] = dyw[pl][u][i + S - 1];
        f32x2 acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = (f32x2){0.f, 0.f};
#pragma unroll
        for (int u = 0; u < R; ++u)
#pragma unroll
          for (int v = 0; v < S; ++v) {
            const f32x2 g = {win[v][u], win[v][u]};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += g * w[u * S + v][e];
          }
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[2 * e] = acc[e].x; o[2 * e + 1] = acc[e].y; }
        if (add != nullptr) {                     // as the GEMM epilogues do: the conv result is rounded, then the sum
          float ad[8];
          Vec16<bf16_t>::unpack(adv[q], ad);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = bf16_round(o[e]) + ad[e];
        }
        Vec16<bf16_t>::store(out + (long long)i * a.Cs, o);
      }
    }
  }
}

struct Thin1WgradArgs {
  const bf16_t* X;       // [N][H][W][Cs]
  const bf16_t* DY;      // [N][OH][OW][8], channel 0 live
  float* partial;        // [strips][R*S][Cs]
  int N, H, W, Cs, OH, OW, pad;
  int nseg, segw;        // a row of x = nseg segments of segw pixels (segw <= kThin1SegW)
  int items_per_strip;   // (image row, segment) items per block strip
};

constexpr int kThin1SegW = 32;

// dw[r][s][c] = sum_{n, ih, iw} x[n][ih][iw][c] * dy[n][ih + pad - r][iw + pad - s]
// A wave = 4 filter rows x 16 channel groups (128 channels); a block = 1..4 waves = 128..512 channels.  A thread owns the S taps
// of its filter row for 8 channels (S x 8 accumulators): per pixel one 16-byte LDS read of x and one new dy value (the S values
// of the row slide) feed S x 8 FMAs.  (First version: a thread per (tap, 8 channels), x as fp32 in LDS -- 36 bytes of LDS reads
// per 8 FMAs: LDS-bound at 2.3 x the time.)
template <int R, int S>
__global__ __launch_bounds__(256) void thin1_wgrad_kernel(const Thin1WgradArgs a) {
  static_assert(R == 4, "a wave = 4 filter rows x 16 channel groups");
  extern __shared__ __attribute__((aligned(16))) char thin1_smem[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int CB = 2 * nthr;                                                 // channels per block: 128 per wave
  bf16_t* const xs = reinterpret_cast<bf16_t*>(thin1_smem);                // [pixel][CB]
  float (*const dyz)[kThin1SegW + S] =                                     // dy row ih + pad - r, zero-extended, shifted by S - 1
      reinterpret_cast<float (*)[kThin1SegW + S]>(thin1_smem + (size_t)kThin1SegW * CB * sizeof(bf16_t));
  const int wave = tid >> 6, lane = tid & 63, r = lane >> 4, cg = lane & 15;
  const int nq = a.Cs / CB;
  const int cq = blockIdx.x % nq, strip = blockIdx.x / nq;
  const int items = a.N * a.H * a.nseg;
  const int it0 = strip * a.items_per_strip;
  int it1 = it0 + a.items_per_strip;
  it1 = it1 < items ? it1 : items;
  constexpr int XV = kThin1SegW * 16 / 64;         // 16-byte vectors of x per thread and item: segw x (CB / 8) over nthr threads
  constexpr int DN = R * (kThin1SegW + S);
  const int vpp = CB >> 3;                         // vectors per pixel
  u32x4 xr[XV];
  float dr[(DN + 63) / 64];
  auto fetch = [&](int it) {                       // loads of item `it` into registers
    const int seg = it % a.nseg, row = it / a.nseg;
    const int ih = row % a.H, n = row / a.H;
    const int w0 = seg * a.segw;
    int wn = a.W - w0;
    wn = wn < a.segw ? wn : a.segw;
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = tid + nthr * i, px = idx / vpp, v = idx - px * vpp;
      u32x4 z = {0u, 0u, 0u, 0u};
      if (px < wn) z = *reinterpret_cast<const u32x4*>(a.X + (((long long)n * a.H + ih) * a.W + w0 + px) * a.Cs + cq * CB + v * 8);
      xr[i] = z;
    }
#pragma unroll
    for (int i = 0; i < (DN + 63) / 64; ++i) {
      const int idx = tid + nthr * i;
      const int rr = idx / (kThin1SegW + S), k = idx - rr * (kThin1SegW + S);
      const int oh = ih + a.pad - rr;
      const int ow = w0 + k - (S - 1) + a.pad;          // entry k of the row: read by pixel px at tap column s as k = px + S - 1 - s
      const bool ok = idx < DN && ((unsigned)oh < (unsigned)a.OH) & ((unsigned)ow < (unsigned)a.OW);
      dr[i] = ok ? bf2f(a.DY[(((long long)n * a.OH + oh) * a.OW + ow) * 8]) : 0.f;
    }
  };
  float acc[S][8];
#pragma unroll
  for (int q = 0; q < S; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
  if (it0 < it1) fetch(it0);
  for (int it = it0; it < it1; ++it) {
    __syncthreads();                               // the previous item's reads are done
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = tid + nthr * i;
      *reinterpret_cast<u32x4*>(xs + (long long)idx * 8) = xr[i];          // idx = px * vpp + v: the tile is [pixel][CB]
    }
#pragma unroll
    for (int i = 0; i < (DN + 63) / 64; ++i) {
      const int idx = tid + nthr * i;
      if (idx < DN) (&dyz[0][0])[idx] = dr[i];
    }
    __syncthreads();
    if (it + 1 < it1) fetch(it + 1);               // in flight under the FMAs below
    const int seg = it % a.nseg;
    int wn = a.W - seg * a.segw;
    wn = wn < a.segw ? wn : a.segw;
    const bf16_t* xp = xs + wave * 128 + cg * 8;
    const float* dp = dyz[r];
    float win[S];                                  // win[j] = dyz[r][px + j]: tap column s reads win[S - 1 - s]
#pragma unroll
    for (int j = 1; j < S; ++j) win[j] = dp[j - 1];
    for (int px = 0; px < wn; ++px) {
#pragma unroll
      for (int j = 0; j + 1 < S; ++j) win[j] = win[j + 1];
      win[S - 1] = dp[px + S - 1];
      float xv[8];
      Vec16<bf16_t>::unpack(*reinterpret_cast<const u32x4*>(xp + px * CB), xv);
#pragma unroll
      for (int q = 0; q < S; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[q][e] += win[S - 1 - q] * xv[e];
    }
  }
  float* out = a.partial + ((long long)strip * (R * S) + r * S) * a.Cs + cq * CB + wave * 128 + cg * 8;
#pragma unroll
  for (int q = 0; q < S; ++q) {
    *reinterpret_cast<float4*>(out + (long long)q * a.Cs) = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
    *reinterpret_cast<float4*>(out + (long long)q * a.Cs + 4) = make_float4(acc[q][4], acc[q][5], acc[q][6], acc[q][7]);
  }
}

// dw[i] = sum over the strips' partials in index order; 16 vectors x 16 slab groups per block (slab_reduce_kernel's 64 x 4 gives
// this 8192-element gradient 32 blocks)
__global__ __launch_bounds__(256) void thin1_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nvec,
                                                          int nslabs) {
  __shared__ float4 red[16][16];
  const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + el;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < nvec) {
    const int per = (nslabs + 15) / 16;
    const int b0 = grp * per;
    int b1 = b0 + per;
    b1 = b1 < nslabs ? b1 : nslabs;
    const float4* src = reinterpret_cast<const float4*>(partial) + idx;
    for (int b = b0; b < b1; ++b) {
      const float4 v = src[(long long)b * nvec];
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
  }
  red[grp][el] = sum;
  __syncthreads();
  if (grp == 0 && idx < nvec) {
    float4 t = red[0][el];
#pragma unroll
    for (int g = 1; g < 16; ++g) { t.x += red[g][el].x; t.y += red[g][el].y; t.z += red[g][el].z; t.w += red[g][el].w; }
    reinterpret_cast<float4*>(dw)[idx] = t;
  }
}

JPDSE_SWITCH(int, g_thin1_enabled, 1);      // 41: the one-output-channel layers on the GEMM paths (A/B)

static bool thin1_shape_ok(const jpdse_conv_desc* d, int Cs, int Ks) {
  const int cv = Cs >> 3;
  return g_fast_enabled && g_thin1_enabled && d->dtype == JPDSE_BF16 && d->K == 1 && Ks == 8 && d->stride == 1 &&
         d->pad_mode != JPDSE_PAD_REFLECT && d->R == 4 && d->S == 4 && d->pad <= 3 && Cs % 128 == 0 && d->C == Cs &&
         cv <= 256 && (cv & (cv - 1)) == 0;
}

static int launch_thin1_dgrad(Thin1DgradArgs a, hipStream_t s) {
  const int cv = a.Cs >> 3;
  if (cv < 8 || cv > 256 || (cv & (cv - 1)) != 0 || a.N <= 0 || a.H <= 0 || a.W <= 0 || (long long)a.N * a.H * a.W >= (1LL << 28))
    return set_error(JPDSE_EINVAL, "thin1_dgrad: %d channels do not split over a 256-thread block (or the image is too large)", a.Cs);
  const int PL = 256 / cv;
  // runs of a row: enough work items for 512 blocks x PL lanes, at least 16 pixels each
  int nseg = (512 * PL + a.N * a.H - 1) / (a.N * a.H);
  const int max_seg = (a.W + 15) / 16, min_seg = (a.W + kThin1Run - 1) / kThin1Run;
  nseg = nseg > max_seg ? max_seg : nseg;
  nseg = nseg < min_seg ? min_seg : nseg;
  a.segw = (a.W + nseg - 1) / nseg;
  a.nseg = (a.W + a.segw - 1) / a.segw;
  const long long items = (long long)a.N * a.H * a.nseg;
  long long blocks = (items + PL - 1) / PL;
  blocks = blocks < 512 ? blocks : 512;            // one resident round: every block loads its filter taps once
  if (a.taps == nullptr) return set_error(JPDSE_EWORKSPACE, "thin1_dgrad: no workspace for the tap table");
  hipLaunchKernelGGL(thin1_taps_kernel, dim3((16 * a.Cs + 255) / 256), dim3(256), 0, s, a.B, a.taps, a.Cs, a.b_stride, a.Lk, 4, 16);
  if (int rc = check_launch("thin1_taps_kernel")) return rc;
  hipLaunchKernelGGL((thin1_dgrad_kernel<4, 4>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  return check_launch("thin1_dgrad_kernel");
}

// strips (= fp32 slabs) of the weight gradient for a problem of `items` (image row, segment) items
static int thin1_wgrad_strips(int items) { return items < 256 ? items : 256; }

static size_t thin1_wgrad_slab_bytes(int N, int H, int W, int Cs) {
  const int nseg = (W + kThin1SegW - 1) / kThin1SegW;
  return (size_t)thin1_wgrad_strips(N * H * nseg) * 16 * Cs * sizeof(float);
}

static int launch_thin1_wgrad(Thin1WgradArgs a, float* dw, hipStream_t s) {
  if (a.Cs % 128 != 0 || a.N <= 0 || a.H <= 0 || a.W <= 0 || a.partial == nullptr)
    return set_error(JPDSE_EINVAL, "thin1_wgrad: bad problem (%d channels)", a.Cs);
  const int wq = a.Cs % 512 == 0 ? 4 : (a.Cs % 256 == 0 ? 2 : 1);      // waves per block = 128-channel quarters it covers
  a.nseg = (a.W + kThin1SegW - 1) / kThin1SegW;
  a.segw = (a.W + a.nseg - 1) / a.nseg;
  a.nseg = (a.W + a.segw - 1) / a.segw;
  const int items = a.N * a.H * a.nseg;
  int strips = thin1_wgrad_strips(items);
  a.items_per_strip = (items + strips - 1) / strips;
  strips = (items + a.items_per_strip - 1) / a.items_per_strip;
  const int nq = a.Cs / (128 * wq);
  const int lds = kThin1SegW * 128 * wq * (int)sizeof(bf16_t) + 4 * (kThin1SegW + 4) * (int)sizeof(float);   // <= 33 KiB
  hipLaunchKernelGGL((thin1_wgrad_kernel<4, 4>), dim3((unsigned)(strips * nq)), dim3(64 * wq), lds, s, a);
  if (int rc = check_launch("thin1_wgrad_kernel")) return rc;
  const int nvec = 16 * a.Cs / 4;
  hipLaunchKernelGGL(thin1_reduce_kernel, dim3((nvec + 15) / 16), dim3(256), 0, s, a.partial, dw, nvec, strips);
  return check_launch("thin1_reduce_kernel");
}

}  // namespace jpdse
