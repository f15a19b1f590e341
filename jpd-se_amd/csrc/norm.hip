// InstanceNorm2d(affine=False) fused with ReLU / LeakyReLU / residual add, forward and
// backward, NHWC (reference: networks.py:31 and the activation sites listed in jpdse.h).
//
// HBM-bound.  Per-(n,c) reductions run as: (1) a moment kernel in which each thread owns one
// 16-byte channel vector column and walks pixels (coalesced: a pixel's channels are
// contiguous), partial sums per pixel-split written to the workspace; (2) a tiny finalize
// kernel (deterministic order, no atomics); (3) a fully parallel apply kernel.
//
// Tensors with few pixel splits per image (everything up to 67 MB in the bench step) take the register-held two-kernel
// form further down: no finalize launch, its work folded into the apply kernel's prologue.
#include "common.h"
#include <map>
#include <mutex>
#include <utility>

namespace jpdse {

struct MomentGeom {
  int N, HW, Cs, cv;      // cv = Cs / VE vector columns
  int TX, TY;             // threads across columns / pixels (TX*TY = 256), TX a power of two
  int tx_shift;           // log2(TX)
  int splits, pix_per_split;
};

static MomentGeom moment_geom(int N, int HW, int Cs, int VE) {
  MomentGeom g;
  g.N = N; g.HW = HW; g.Cs = Cs; g.cv = Cs / VE;
  int tx = 1;
  while (tx < g.cv && tx < 256) tx <<= 1;
  g.TX = tx; g.TY = 256 / tx;
  g.tx_shift = 0;
  while ((1 << g.tx_shift) < tx) ++g.tx_shift;
  // aim for ~2048 blocks in total, at least 8 pixels per thread row
  const int col_blocks = (g.cv + g.TX - 1) / g.TX;
  long long want = 2048 / ((long long)N * col_blocks);
  if (want < 1) want = 1;
  long long max_splits = HW / ((long long)g.TY * 8);
  if (max_splits < 1) max_splits = 1;
  if (want > max_splits) want = max_splits;
  if (want > 256) want = 256;
  g.splits = (int)want;
  g.pix_per_split = (HW + g.splits - 1) / g.splits;
  g.splits = (HW + g.pix_per_split - 1) / g.pix_per_split;
  return g;
}

#ifndef JPDSE_NORM_ILP
#define JPDSE_NORM_ILP 2      // pixels per thread and loop trip of the streaming norm kernels (bytes in flight; A/B with -DJPDSE_NORM_ILP=1)
#endif

// ---- per-element functors -----------------------------------------------------------------
// forward moments: (x - shift), (x - shift)^2 with shift = x[n, pixel 0, c]
template <typename T> struct FwdMoments {
  const T* x;
  __device__ __forceinline__ void prep(int n, int HW, int Cs, int c0, float (&aux)[8]) const {
    float v[Vec16<T>::N];
    Vec16<T>::load(x + (long long)n * HW * Cs + c0, v);
#pragma unroll
    for (int e = 0; e < Vec16<T>::N; ++e) aux[e] = v[e];
  }
  __device__ __forceinline__ void acc(const float (&v)[Vec16<T>::N], const float (&aux)[8], float (&s1)[8], float (&s2)[8]) const {
#pragma unroll
    for (int e = 0; e < Vec16<T>::N; ++e) {
      const float d = v[e] - aux[e];
      s1[e] += d;
      s2[e] += d * d;
    }
  }
  __device__ __forceinline__ void at(long long off, int, const float (&aux)[8], float (&s1)[8], float (&s2)[8]) const {
    float v[Vec16<T>::N];
    Vec16<T>::load(x + off, v);
    acc(v, aux, s1, s2);
  }
};

__device__ __forceinline__ float act_grad(float yhat, int act, float slope) {
  if (act == JPDSE_ACT_RELU) return yhat > 0.f ? 1.f : 0.f;
  if (act == JPDSE_ACT_LRELU) return yhat > 0.f ? 1.f : slope;
  return 1.f;
}

// backward moments: dz, dz*yhat with dz = dy*act'(yhat), yhat = (x-mean)*rstd
template <typename T> struct BwdMoments {
  const T* x;
  const T* dy;
  const float* stats;  // [N][Cs][2]
  int act;
  float slope;
  __device__ __forceinline__ void acc2(const float (&v)[Vec16<T>::N], const float (&g)[Vec16<T>::N], const float (&mean)[Vec16<T>::N],
                                       const float (&rstd)[Vec16<T>::N], float (&s1)[8], float (&s2)[8]) const {
#pragma unroll
    for (int e = 0; e < Vec16<T>::N; ++e) {
      const float yh = (v[e] - mean[e]) * rstd[e];
      const float dz = g[e] * act_grad(yh, act, slope);
      s1[e] += dz;
      s2[e] += dz * yh;
    }
  }
  // aux packs mean[e] in [0..VE) -- rstd kept separately
  __device__ __forceinline__ void at2(long long off, int n, int Cs, int c0, float (&s1)[8], float (&s2)[8]) const {
    float v[Vec16<T>::N], g[Vec16<T>::N];
    Vec16<T>::load(x + off, v);
    Vec16<T>::load(dy + off, g);
    const float* st = stats + ((long long)n * Cs + c0) * 2;
#pragma unroll
    for (int e = 0; e < Vec16<T>::N; ++e) {
      const float yh = (v[e] - st[2 * e]) * st[2 * e + 1];
      const float dz = g[e] * act_grad(yh, act, slope);
      s1[e] += dz;
      s2[e] += dz * yh;
    }
  }
};

// partial[n][split][c][2]
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void moment_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                    const float* __restrict__ stats, int act, float slope,
                                                    float* __restrict__ partial, MomentGeom g) {
  constexpr int VE = Vec16<T>::N;
  __shared__ float red[256 * 2 * VE];
  const int tx = threadIdx.x & (g.TX - 1), ty = threadIdx.x >> g.tx_shift;
  const int col_blocks = (g.cv + g.TX - 1) / g.TX;
  const int cb = blockIdx.x % col_blocks;
  const int split = (blockIdx.x / col_blocks) % g.splits;
  const int n = blockIdx.x / (col_blocks * g.splits);
  const int col = cb * g.TX + tx;
  const bool on = col < g.cv;
  const int c0 = col * VE;
  float s1[8], s2[8], aux[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; aux[e] = 0.f; }
  if (on) {
    const int p0 = split * g.pix_per_split;
    int p1 = p0 + g.pix_per_split;
    p1 = p1 < g.HW ? p1 : g.HW;
    const long long base = (long long)n * g.HW * g.Cs + c0;
    if (!BWD) {
      FwdMoments<T> f{x};
      f.prep(n, g.HW, g.Cs, c0, aux);
      // JPDSE_NORM_ILP pixels per trip, all loads issued before the first use (same summation order: p ascending)
      int p = p0 + ty;
      for (; p + (JPDSE_NORM_ILP - 1) * g.TY < p1; p += JPDSE_NORM_ILP * g.TY) {
        float v[JPDSE_NORM_ILP][VE];
#pragma unroll
        for (int u = 0; u < JPDSE_NORM_ILP; ++u) Vec16<T>::load(x + base + (long long)(p + u * g.TY) * g.Cs, v[u]);
#pragma unroll
        for (int u = 0; u < JPDSE_NORM_ILP; ++u) f.acc(v[u], aux, s1, s2);
      }
      for (; p < p1; p += g.TY) f.at(base + (long long)p * g.Cs, 0, aux, s1, s2);
    } else {
      BwdMoments<T> f{x, dy, stats, act, slope};
      float mean[VE], rstd[VE];
      {
        const float* st = stats + ((long long)n * g.Cs + c0) * 2;
#pragma unroll
        for (int e = 0; e < VE; ++e) { mean[e] = st[2 * e]; rstd[e] = st[2 * e + 1]; }
      }
      int p = p0 + ty;
      for (; p + (JPDSE_NORM_ILP - 1) * g.TY < p1; p += JPDSE_NORM_ILP * g.TY) {
        float v[JPDSE_NORM_ILP][VE], gr[JPDSE_NORM_ILP][VE];
#pragma unroll
        for (int u = 0; u < JPDSE_NORM_ILP; ++u) {
          const long long off = base + (long long)(p + u * g.TY) * g.Cs;
          Vec16<T>::load(x + off, v[u]);
          Vec16<T>::load(dy + off, gr[u]);
        }
#pragma unroll
        for (int u = 0; u < JPDSE_NORM_ILP; ++u) f.acc2(v[u], gr[u], mean, rstd, s1, s2);
      }
      for (; p < p1; p += g.TY) {
        float v[VE], gr[VE];
        Vec16<T>::load(x + base + (long long)p * g.Cs, v);
        Vec16<T>::load(dy + base + (long long)p * g.Cs, gr);
        f.acc2(v, gr, mean, rstd, s1, s2);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    red[(threadIdx.x * VE + e) * 2] = s1[e];
    red[(threadIdx.x * VE + e) * 2 + 1] = s2[e];
  }
  __syncthreads();
  if (ty == 0 && on) {
    float* out = partial + (((long long)n * g.splits + split) * g.Cs + c0) * 2;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      float a = 0.f, b = 0.f;
      for (int y = 0; y < g.TY; ++y) {
        a += red[((y * g.TX + tx) * VE + e) * 2];
        b += red[((y * g.TX + tx) * VE + e) * 2 + 1];
      }
      out[2 * e] = a;
      out[2 * e + 1] = b;
    }
  }
}

// Finalize kernels: 8 lanes cooperate on one (n,c) (strided over the pixel splits, then a
// shuffle reduction) -- a serial loop per thread was latency-bound (~40 us for 256 splits).
__device__ __forceinline__ void reduce_splits(const float* __restrict__ partial, int n, int c, int Cs, int splits,
                                              int sub, float& a, float& b) {
  a = 0.f;
  b = 0.f;
  // four partials per trip, loaded before any is added: the serial form (one L2 round trip per split) made these
  // tiny kernels ~8 us each, 72 of them per step
  const long long stride = (long long)Cs * 2;
  const float* p = partial + (((long long)n * splits + sub) * Cs + c) * 2;
  int s = sub;
  for (; s + 24 < splits; s += 32, p += 32 * stride) {
    const float2 v0 = *reinterpret_cast<const float2*>(p);
    const float2 v1 = *reinterpret_cast<const float2*>(p + 8 * stride);
    const float2 v2 = *reinterpret_cast<const float2*>(p + 16 * stride);
    const float2 v3 = *reinterpret_cast<const float2*>(p + 24 * stride);
    a += v0.x; b += v0.y;
    a += v1.x; b += v1.y;
    a += v2.x; b += v2.y;
    a += v3.x; b += v3.y;
  }
  for (; s < splits; s += 8, p += 8 * stride) {
    const float2 v = *reinterpret_cast<const float2*>(p);
    a += v.x;
    b += v.y;
  }
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
}

// forward finalize: partial sums of shifted values -> (mean, rstd)
template <typename T>
__global__ void finalize_fwd_kernel(const T* __restrict__ x, const float* __restrict__ partial,
                                    float* __restrict__ stats, int N, int HW, int Cs, int splits, float eps) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int idx = gid >> 3, sub = gid & 7;
  const bool on = idx < N * Cs;
  const int n = on ? idx / Cs : 0, c = on ? idx % Cs : 0;
  float a, b;
  reduce_splits(partial, n, c, Cs, on ? splits : 0, sub, a, b);
  if (!on || sub != 0) return;
  const float shift = ElemOps<T>::ld(x + (long long)n * HW * Cs + c);
  const float inv = 1.f / (float)HW;
  const float dm = a * inv;
  float var = b * inv - dm * dm;
  var = var > 0.f ? var : 0.f;
  stats[2 * idx] = shift + dm;
  stats[2 * idx + 1] = rsqrtf(var + eps);
}

// backward finalize: (mean(dz), mean(dz*yhat))
__global__ void finalize_bwd_kernel(const float* __restrict__ partial, float* __restrict__ sums, int N, int HW,
                                    int Cs, int splits) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int idx = gid >> 3, sub = gid & 7;
  const bool on = idx < N * Cs;
  const int n = on ? idx / Cs : 0, c = on ? idx % Cs : 0;
  float a, b;
  reduce_splits(partial, n, c, Cs, on ? splits : 0, sub, a, b);
  if (!on || sub != 0) return;
  const float inv = 1.f / (float)HW;
  sums[2 * idx] = a * inv;
  sums[2 * idx + 1] = b * inv;
}

// Apply kernels: same block decomposition as the moment kernel (one 16-byte channel-vector column per
// lane, pixels walked with stride TY), so the per-channel constants are loaded once per thread and the
// inner loop has no integer division; coalescing is along the contiguous channel axis.
template <typename T>
__global__ __launch_bounds__(256) void inorm_apply_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                            T* __restrict__ y, const float* __restrict__ stats,
                                                            int act, float slope, MomentGeom g) {
  constexpr int VE = Vec16<T>::N;
  const int tx = threadIdx.x & (g.TX - 1), ty = threadIdx.x >> g.tx_shift;
  const int col_blocks = (g.cv + g.TX - 1) / g.TX;
  const int cb = blockIdx.x % col_blocks;
  const int split = (blockIdx.x / col_blocks) % g.splits;
  const int n = blockIdx.x / (col_blocks * g.splits);
  const int col = cb * g.TX + tx;
  if (col >= g.cv) return;
  const int c0 = col * VE;
  float mean[VE], rstd[VE];
  const float* st = stats + ((long long)n * g.Cs + c0) * 2;
#pragma unroll
  for (int e = 0; e < VE; ++e) { mean[e] = st[2 * e]; rstd[e] = st[2 * e + 1]; }
  const int p0 = split * g.pix_per_split;
  int p1 = p0 + g.pix_per_split;
  p1 = p1 < g.HW ? p1 : g.HW;
  const long long base = (long long)n * g.HW * g.Cs + c0;
  // JPDSE_NORM_ILP pixels per trip: their loads are issued before the first store
  for (int p = p0 + ty; p < p1; p += JPDSE_NORM_ILP * g.TY) {
    float v[JPDSE_NORM_ILP][VE], r[JPDSE_NORM_ILP][VE];
#pragma unroll
    for (int u = 0; u < JPDSE_NORM_ILP; ++u) {
      const int pu = p + u * g.TY;
      if (pu < p1) {
        const long long off = base + (long long)pu * g.Cs;
        JPDSE_LOAD_LAST(T, x + off, v[u]);
        if (res != nullptr) Vec16<T>::load(res + off, r[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < JPDSE_NORM_ILP; ++u) {
      const int pu = p + u * g.TY;
      if (pu >= p1) break;
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        float t = (v[u][e] - mean[e]) * rstd[e];
        if (act == JPDSE_ACT_RELU) t = t > 0.f ? t : 0.f;
        else if (act == JPDSE_ACT_LRELU) t = t > 0.f ? t : t * slope;
        if (res != nullptr) t += r[u][e];
        v[u][e] = t;
      }
      Vec16<T>::store(y + base + (long long)pu * g.Cs, v[u]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void inorm_apply_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            T* __restrict__ dx, const float* __restrict__ stats,
                                                            const float* __restrict__ sums, int act, float slope,
                                                            MomentGeom g) {
  constexpr int VE = Vec16<T>::N;
  const int tx = threadIdx.x & (g.TX - 1), ty = threadIdx.x >> g.tx_shift;
  const int col_blocks = (g.cv + g.TX - 1) / g.TX;
  const int cb = blockIdx.x % col_blocks;
  const int split = (blockIdx.x / col_blocks) % g.splits;
  const int n = blockIdx.x / (col_blocks * g.splits);
  const int col = cb * g.TX + tx;
  if (col >= g.cv) return;
  const int c0 = col * VE;
  float mean[VE], rstd[VE], s1[VE], s2[VE];
  const long long sidx = ((long long)n * g.Cs + c0) * 2;
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    mean[e] = stats[sidx + 2 * e];
    rstd[e] = stats[sidx + 2 * e + 1];
    s1[e] = sums[sidx + 2 * e];
    s2[e] = sums[sidx + 2 * e + 1];
  }
  const int p0 = split * g.pix_per_split;
  int p1 = p0 + g.pix_per_split;
  p1 = p1 < g.HW ? p1 : g.HW;
  const long long base = (long long)n * g.HW * g.Cs + c0;
  for (int p = p0 + ty; p < p1; p += JPDSE_NORM_ILP * g.TY) {
    float v[JPDSE_NORM_ILP][VE], gr[JPDSE_NORM_ILP][VE];
#pragma unroll
    for (int u = 0; u < JPDSE_NORM_ILP; ++u) {
      const int pu = p + u * g.TY;
      if (pu < p1) {
        const long long off = base + (long long)pu * g.Cs;
        JPDSE_LOAD_LAST(T, x + off, v[u]);
        JPDSE_LOAD_LAST(T, dy + off, gr[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < JPDSE_NORM_ILP; ++u) {
      const int pu = p + u * g.TY;
      if (pu >= p1) break;
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const float yh = (v[u][e] - mean[e]) * rstd[e];
        const float dz = gr[u][e] * act_grad(yh, act, slope);
        v[u][e] = rstd[e] * (dz - s1[e] - yh * s2[e]);
      }
      Vec16<T>::store(dx + base + (long long)pu * g.Cs, v[u]);
    }
  }
}


// ---- register-held forms ------------------------------------------------------------------
// Geometry shared by the two forms below: a block owns `span` = TY * P pixels of TX 16-byte channel columns of one image and
// holds them in registers; the `splits` blocks of one (image, channel block) = one GROUP are consecutive in the grid.
//   TWO kernels (default, any tensor with <= 64 splits): PHASE 1 writes one row of partial sums per block, PHASE 2 loads
//     its pixels, sums the group's rows itself (split order: deterministic; the rows are L2 hits and load under the
//     pixel loads), and applies.  Against moment -> finalize -> apply this drops a launch and the finalize round trip.
//   ONE kernel (PHASE 0; developer A/B mode 28, tensors that fit a resident grid): the blocks of a group exchange their rows
//     inside the launch -- `sc1` stores and loads for every handed-off byte, one agent-scope add per storing workgroup
//     behind that workgroup's vmcnt(0) + barrier (MI355X_MICROARCH.md, "inter-workgroup visibility") -- and x / dy are
//     read once.  Measured on the ResnetBlock tensor (16.8 MB): each cross-XCD hop of the exchange costs ~2 us, which
//     eats what the saved launch and re-read give (profiles/r02_norm_forms.txt), so it is not the shipped form.
static inline int norm_form() {
#ifdef JPDSE_DEV
  return g_norm_fused;     // 1 = two kernels (shipped default), 0 = moment -> finalize -> apply, 2 = one kernel
#else
  return 1;
#endif
}
#ifdef JPDSE_DEV
int g_norm_fused = 1;
#endif

struct FusedGeom {
  int N, HW, Cs, cv;
  int TX, TY, tx_shift;          // threads across 16-byte channel columns / pixels (TX * TY = 256)
  int col_blocks, splits, span;  // blocks per image across channels / pixels; pixels per block (TY * P)
  int nv;                        // floats per partial row: TX * VE * 2
};

static FusedGeom fused_geom(int N, int HW, int Cs, int VE, int P) {
  FusedGeom g;
  g.N = N; g.HW = HW; g.Cs = Cs; g.cv = Cs / VE;
  int tx = 1;
  while (tx < g.cv && tx < 16) tx <<= 1;       // <= 256 contiguous bytes per pixel and block
  g.TX = tx; g.TY = 256 / tx;
  g.tx_shift = 0;
  while ((1 << g.tx_shift) < tx) ++g.tx_shift;
  g.col_blocks = (g.cv + g.TX - 1) / g.TX;
  g.span = g.TY * P;
  g.splits = (HW + g.span - 1) / g.span;
  g.nv = g.TX * VE * 2;
  return g;
}

__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Sum over the block's TY pixel rows: value j (< nv) of the block's partial row, returned in thread j.
template <int VE>
__device__ __forceinline__ float block_row(float* red, const float (&s1)[8], const float (&s2)[8], const FusedGeom& g) {
  const int tid = threadIdx.x;
  const int tx = tid & (g.TX - 1), ty = tid >> g.tx_shift;
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    red[ty * g.nv + (tx * VE + e) * 2] = s1[e];
    red[ty * g.nv + (tx * VE + e) * 2 + 1] = s2[e];
  }
  __syncthreads();
  float tsum = 0.f;
  if (tid < g.nv)
    for (int yy = 0; yy < g.TY; ++yy) tsum += red[yy * g.nv + tid];
  __syncthreads();                                    // red[] is free again
  return tsum;
}

// Sum of the group's rows in split order (sixteen loads in flight, added in order); COHERENT: rows written inside this launch.
template <bool COHERENT>
__device__ __forceinline__ float sum_rows(const float* rows, int splits, int nv) {
  float total = 0.f;
  for (int s0 = 0; s0 < splits; s0 += 16) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float* q = rows + (long long)(s0 + k) * nv;
      v[k] = 0.f;
      if (s0 + k < splits) v[k] = COHERENT ? ld_sc1(q) : *q;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) total += v[k];       // + 0.f for the absent rows changes nothing
  }
  return total;
}

// PHASE 0: on return red[j] holds the group total of value j.  `count`: one zeroed word per group, used once.
template <int VE>
__device__ __forceinline__ void exchange_totals(float* red, int* s_ctl, float tsum, float* partial, unsigned* count, int grp,
                                                int split, const FusedGeom& g) {
  const int tid = threadIdx.x;
  float total = tsum;
  if (g.splits > 1) {
    float* const rows = partial + (long long)grp * g.splits * g.nv;
    if (tid < g.nv) st_sc1(rows + (long long)split * g.nv + tid, tsum);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(count + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned it = 0;
      int gave_up = 0;
      while (ld_sc1(count + grp) < (unsigned)g.splits) {
        __builtin_amdgcn_s_sleep(2);
        if (++it > (1u << 22)) { gave_up = 1; break; }   // every wave reaches an exit; the result is then NaN (loud)
      }
      s_ctl[0] = gave_up;
    }
    __syncthreads();
    if (tid < g.nv) total = s_ctl[0] ? __builtin_nanf("") : sum_rows<true>(rows + tid, g.splits, g.nv);
  }
  if (tid < g.nv) red[tid] = total;
  __syncthreads();
}

template <typename T, int P, int PHASE>
__global__ __launch_bounds__(256, 4) void inorm_reg_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                             T* __restrict__ y, float* __restrict__ stats, float* partial,
                                                             unsigned* count, int act, float slope, float eps, FusedGeom g,
                                                             int mslots = 0) {
  // PHASE 3: statistics from the per-block (mean, M2) slots a conv epilogue wrote (`partial` = moments[n][c][mslots][2],
  // common.h): one kernel per norm -- each block merges the slots of its channels itself (Chan's formula, slot order) and applies
  constexpr int VE = Vec16<T>::N;
  __shared__ float red[256 * VE * 2];
  __shared__ int s_ctl[2];
  const int tid = threadIdx.x;
  const int tx = tid & (g.TX - 1), ty = tid >> g.tx_shift;
  const int split = blockIdx.x % g.splits, grp = blockIdx.x / g.splits;
  const int cb = grp % g.col_blocks, n = grp / g.col_blocks;
  const int col = cb * g.TX + tx;
  const bool on = col < g.cv;
  const int c0 = (on ? col : 0) * VE;
  const long long base = (long long)n * g.HW * g.Cs + c0;
  const int p0 = split * g.span + ty;
  u32x4 xv[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int p = p0 + i * g.TY;
    if (on && p < g.HW) xv[i] = *reinterpret_cast<const u32x4*>(x + base + (long long)p * g.Cs);
  }
  float aux[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { aux[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; }
  if (on && PHASE != 3) {
    float v[VE];
    Vec16<T>::load(x + base, v);                      // shift = the image's first pixel, as in the three-kernel form
#pragma unroll
    for (int e = 0; e < VE; ++e) aux[e] = v[e];
  }
  if constexpr (PHASE == 3) {
    const int nch = g.TX * VE;
    if (tid < nch) {
      const int ch = cb * nch + tid;
      float mean = 0.f, var = 0.f;
      if (ch < g.Cs) {
        const float2* const src = reinterpret_cast<const float2*>(partial) + ((long long)n * g.Cs + ch) * mslots;
        float a = 0.f;
        for (int q = 0; q < mslots; ++q) a += src[q].x;
        mean = a / (float)mslots;
        const float ns = (float)g.HW / (float)mslots;
        float b = 0.f;
        for (int q = 0; q < mslots; ++q) {
          const float2 v = src[q];
          const float dm = v.x - mean;
          b += v.y + ns * dm * dm;
        }
        var = b / (float)g.HW;
        var = var > 0.f ? var : 0.f;
      }
      red[tid * 2] = mean;
      red[tid * 2 + 1] = var;
    }
    __syncthreads();
  } else if constexpr (PHASE == 2) {
    const float* rows = partial + (long long)grp * g.splits * g.nv;
    const float total = tid < g.nv ? sum_rows<false>(rows + tid, g.splits, g.nv) : 0.f;
    if (tid < g.nv) red[tid] = total;
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int p = p0 + i * g.TY;
      if (on && p < g.HW) {
        float v[VE];
        Vec16<T>::unpack(xv[i], v);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const float d = v[e] - aux[e];
          s1[e] += d;
          s2[e] += d * d;
        }
      }
    }
    const float tsum = block_row<VE>(red, s1, s2, g);
    if constexpr (PHASE == 1) {
      if (tid < g.nv) partial[((long long)grp * g.splits + split) * g.nv + tid] = tsum;
      return;
    } else {
      exchange_totals<VE>(red, s_ctl, tsum, partial, count, grp, split, g);
    }
  }
  float mean[VE], rstd[VE];
  const float inv = 1.f / (float)g.HW;
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    if constexpr (PHASE == 3) {
      mean[e] = red[(tx * VE + e) * 2];
      rstd[e] = rsqrtf(red[(tx * VE + e) * 2 + 1] + eps);
    } else {
      const float dm = red[(tx * VE + e) * 2] * inv;
      float var = red[(tx * VE + e) * 2 + 1] * inv - dm * dm;
      var = var > 0.f ? var : 0.f;
      mean[e] = aux[e] + dm;
      rstd[e] = rsqrtf(var + eps);
    }
  }
  if (on && split == 0 && ty == 0) {
    float* st = stats + ((long long)n * g.Cs + c0) * 2;
#pragma unroll
    for (int e = 0; e < VE; ++e) { st[2 * e] = mean[e]; st[2 * e + 1] = rstd[e]; }
  }
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int p = p0 + i * g.TY;
    if (on && p < g.HW) {
      const long long off = base + (long long)p * g.Cs;
      float v[VE], r[VE];
      Vec16<T>::unpack(xv[i], v);
      if (res != nullptr) Vec16<T>::load(res + off, r);
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        float t = (v[e] - mean[e]) * rstd[e];
        if (act == JPDSE_ACT_RELU) t = t > 0.f ? t : 0.f;
        else if (act == JPDSE_ACT_LRELU) t = t > 0.f ? t : t * slope;
        if (res != nullptr) t += r[e];
        v[e] = t;
      }
      Vec16<T>::store(y + off, v);
    }
  }
}

template <typename T, int P, int PHASE>
__global__ __launch_bounds__(256, (P == 8 ? 3 : 2)) void inorm_reg_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                             T* __restrict__ dx,
                                                                             const float* __restrict__ stats, float* partial,
                                                                             unsigned* count, int act, float slope,
                                                                             FusedGeom g, int mslots = 0) {
  // PHASE 3 (round 4): the two sums come from the per-block slots that the epilogue of the data-gradient kernel producing dy
  // wrote (`partial` = sums[n][c][mslots][2], gemm_halo.h NSUM): ONE kernel per norm backward -- each block adds the slots of
  // its channels itself (slot order: deterministic) and applies
  constexpr int VE = Vec16<T>::N;
  __shared__ float red[256 * VE * 2];
  __shared__ int s_ctl[2];
  const int tid = threadIdx.x;
  const int tx = tid & (g.TX - 1), ty = tid >> g.tx_shift;
  const int split = blockIdx.x % g.splits, grp = blockIdx.x / g.splits;
  const int cb = grp % g.col_blocks, n = grp / g.col_blocks;
  const int col = cb * g.TX + tx;
  const bool on = col < g.cv;
  const int c0 = (on ? col : 0) * VE;
  const long long base = (long long)n * g.HW * g.Cs + c0;
  const int p0 = split * g.span + ty;
  u32x4 xv[P], gv[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int p = p0 + i * g.TY;
    if (on && p < g.HW) {
      xv[i] = *reinterpret_cast<const u32x4*>(x + base + (long long)p * g.Cs);
      gv[i] = *reinterpret_cast<const u32x4*>(dy + base + (long long)p * g.Cs);
    }
  }
  float mean[VE], rstd[VE], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  {
    const float* st = stats + ((long long)n * g.Cs + c0) * 2;
#pragma unroll
    for (int e = 0; e < VE; ++e) { mean[e] = st[2 * e]; rstd[e] = st[2 * e + 1]; }
  }
  if constexpr (PHASE == 3) {
    const int nch = g.TX * VE;
    if (tid < nch) {
      const int ch = cb * nch + tid;
      float a = 0.f, b = 0.f;
      if (ch < g.Cs) {
        const float2* const src = reinterpret_cast<const float2*>(partial) + ((long long)n * g.Cs + ch) * mslots;
        for (int q = 0; q < mslots; ++q) {
          const float2 v = src[q];
          a += v.x;
          b += v.y;
        }
      }
      red[tid * 2] = a;
      red[tid * 2 + 1] = b;
    }
    __syncthreads();
  } else if constexpr (PHASE == 2) {
    const float* rows = partial + (long long)grp * g.splits * g.nv;
    const float total = tid < g.nv ? sum_rows<false>(rows + tid, g.splits, g.nv) : 0.f;
    if (tid < g.nv) red[tid] = total;
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int p = p0 + i * g.TY;
      if (on && p < g.HW) {
        float v[VE], gr[VE];
        Vec16<T>::unpack(xv[i], v);
        Vec16<T>::unpack(gv[i], gr);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const float yh = (v[e] - mean[e]) * rstd[e];
          const float dz = gr[e] * act_grad(yh, act, slope);
          s1[e] += dz;
          s2[e] += dz * yh;
        }
      }
    }
    const float tsum = block_row<VE>(red, s1, s2, g);
    if constexpr (PHASE == 1) {
      if (tid < g.nv) partial[((long long)grp * g.splits + split) * g.nv + tid] = tsum;
      return;
    } else {
      exchange_totals<VE>(red, s_ctl, tsum, partial, count, grp, split, g);
    }
  }
  float m1[VE], m2[VE];
  const float inv = 1.f / (float)g.HW;
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    m1[e] = red[(tx * VE + e) * 2] * inv;
    m2[e] = red[(tx * VE + e) * 2 + 1] * inv;
  }
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int p = p0 + i * g.TY;
    if (on && p < g.HW) {
      float v[VE], gr[VE];
      Vec16<T>::unpack(xv[i], v);
      Vec16<T>::unpack(gv[i], gr);
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const float yh = (v[e] - mean[e]) * rstd[e];
        const float dz = gr[e] * act_grad(yh, act, slope);
        v[e] = rstd[e] * (dz - m1[e] - yh * m2[e]);
      }
      Vec16<T>::store(dx + base + (long long)p * g.Cs, v);
    }
  }
}

// One-kernel form only: zeroed count words, each used by one group of one launch; a (device, stream) ring that a
// stream-ordered memset refills when it runs out (every ~2000 launches).
constexpr int kCountSlots = 1 << 16;
struct SyncState { unsigned* dev = nullptr; int cursor = 0; };
static unsigned* count_slots(hipStream_t s, int groups) {
#ifndef JPDSE_DEV
  (void)s; (void)groups;
  return nullptr;            // the shipped library never allocates: the one-kernel form exists in the developer build only
#else
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, SyncState> table;
  int dev = 0;
  if (groups > kCountSlots || hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  SyncState& st = table[std::make_pair(dev, s)];
  if (st.dev == nullptr) {
    if (hipMalloc(reinterpret_cast<void**>(&st.dev), kCountSlots * sizeof(unsigned)) != hipSuccess) return nullptr;
    st.cursor = kCountSlots;     // forces the first fill
  }
  if (st.cursor + groups > kCountSlots) {
    if (hipMemsetAsync(st.dev, 0, kCountSlots * sizeof(unsigned), s) != hipSuccess) return nullptr;
    st.cursor = 0;
  }
  unsigned* p = st.dev + st.cursor;
  st.cursor += groups;
  return p;
#endif
}

// blocks of `kernel` that are resident at once (the exchange must never wait for a block that cannot start)
template <typename K> static int resident_blocks(K kernel) {
  int occ = 0, cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, 0) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return occ * cus;
}

// Pixels held per thread for the register-held forms (0: moment -> finalize -> apply).  `one`: the one-kernel form is wanted.
template <typename T, bool BWD> static int reg_pick(const jpdse_inorm_desc* d, bool one, FusedGeom* out) {
  constexpr int VE = Vec16<T>::N;
  static int cap8 = -1, cap16 = -1;
#ifdef JPDSE_DEV
  if (one && cap8 < 0) {
    if (BWD) {
      cap8 = resident_blocks(inorm_reg_bwd_kernel<T, 8, 0>);
      cap16 = resident_blocks(inorm_reg_bwd_kernel<T, 16, 0>);
    } else {
      cap8 = resident_blocks(inorm_reg_fwd_kernel<T, 8, 0>);
      cap16 = resident_blocks(inorm_reg_fwd_kernel<T, 16, 0>);
    }
  }
#endif
  // two-kernel form: 8 pixels per thread only -- with 16 (tensors of 17-34 MB) it measured equal or slower than the
  // three-kernel form (profiles/r02_norm_forms.txt)
  for (int P = 8; P <= (one ? 16 : 8); P *= 2) {
    const FusedGeom g = fused_geom(d->N, d->H * d->W, cpad(d->C), VE, P);
    const long long groups = (long long)d->N * g.col_blocks, blocks = groups * g.splits;
    if (g.splits > 64 || blocks > (1ll << 30)) continue;
    if (one && (groups > kCountSlots || blocks > (P == 8 ? cap8 : cap16))) continue;
    *out = g;
    return P;
  }
  return 0;
}

static size_t reg_ws_bytes(const jpdse_inorm_desc* d) {
  // sized for either P (P = 8 has the larger split count); 0 when the register-held forms do not apply
  const int VE = d->dtype == JPDSE_BF16 ? 8 : 4;
  const FusedGeom g = fused_geom(d->N, d->H * d->W, cpad(d->C), VE, 8);
  if (g.splits > 128) return 0;
  return align_up((size_t)d->N * g.col_blocks * g.splits * g.nv * sizeof(float), 256);
}

template <typename T, int P>
static int launch_reg_fwd(int form, const FusedGeom& fg, const jpdse_inorm_desc* d, const void* x, const void* res, void* y,
                          float* stats, float* part, unsigned* count, hipStream_t s) {
  const dim3 grid((unsigned)((size_t)d->N * fg.col_blocks * fg.splits));
  const T* xp = reinterpret_cast<const T*>(x);
  const T* rp = reinterpret_cast<const T*>(res);
  T* yp = reinterpret_cast<T*>(y);
#ifdef JPDSE_DEV
  if (form == 2) {
    hipLaunchKernelGGL((inorm_reg_fwd_kernel<T, P, 0>), grid, dim3(256), 0, s, xp, rp, yp, stats, part, count, d->act, d->slope,
                       d->eps, fg);
    return check_launch("inorm one-kernel fwd");
  }
#endif
  hipLaunchKernelGGL((inorm_reg_fwd_kernel<T, P, 1>), grid, dim3(256), 0, s, xp, rp, yp, stats, part, count, d->act, d->slope,
                     d->eps, fg);
  if (int rc = check_launch("inorm rows fwd")) return rc;
  hipLaunchKernelGGL((inorm_reg_fwd_kernel<T, P, 2>), grid, dim3(256), 0, s, xp, rp, yp, stats, part, count, d->act, d->slope,
                     d->eps, fg);
  return check_launch("inorm apply-from-rows fwd");
}

template <typename T, int P>
static int launch_reg_bwd(int form, const FusedGeom& fg, const jpdse_inorm_desc* d, const void* x, const float* stats,
                          const void* dy, void* dx, float* part, unsigned* count, hipStream_t s) {
  const dim3 grid((unsigned)((size_t)d->N * fg.col_blocks * fg.splits));
  const T* xp = reinterpret_cast<const T*>(x);
  const T* gp = reinterpret_cast<const T*>(dy);
  T* dp = reinterpret_cast<T*>(dx);
#ifdef JPDSE_DEV
  if (form == 2) {
    hipLaunchKernelGGL((inorm_reg_bwd_kernel<T, P, 0>), grid, dim3(256), 0, s, xp, gp, dp, stats, part, count, d->act, d->slope,
                       fg);
    return check_launch("inorm one-kernel bwd");
  }
#endif
  hipLaunchKernelGGL((inorm_reg_bwd_kernel<T, P, 1>), grid, dim3(256), 0, s, xp, gp, dp, stats, part, count, d->act, d->slope, fg);
  if (int rc = check_launch("inorm rows bwd")) return rc;
  hipLaunchKernelGGL((inorm_reg_bwd_kernel<T, P, 2>), grid, dim3(256), 0, s, xp, gp, dp, stats, part, count, d->act, d->slope, fg);
  return check_launch("inorm apply-from-rows bwd");
}

// 1 = handled by a register-held form, 0 = not applicable, < 0 = error
template <typename T>
static int try_reg_fwd(const jpdse_inorm_desc* d, const void* x, const void* res, void* y, float* stats, void* ws,
                       hipStream_t s) {
  int form = norm_form();
  if (form == 0) return 0;
  FusedGeom fg;
  int P = reg_pick<T, false>(d, form == 2, &fg);
  if (P == 0 && form == 2) { form = 1; P = reg_pick<T, false>(d, false, &fg); }
  if (P == 0) return 0;
  unsigned* count = nullptr;
  if (form == 2 && fg.splits > 1) {
    count = count_slots(s, d->N * fg.col_blocks);
    if (count == nullptr) form = 1;
  }
  float* part = reinterpret_cast<float*>(ws);
  const int rc = P == 8 ? launch_reg_fwd<T, 8>(form, fg, d, x, res, y, stats, part, count, s)
                        : launch_reg_fwd<T, 16>(form, fg, d, x, res, y, stats, part, count, s);
  return rc == JPDSE_OK ? 1 : -1;
}

template <typename T>
static int try_reg_bwd(const jpdse_inorm_desc* d, const void* x, const float* stats, const void* dy, void* dx, void* ws,
                       hipStream_t s) {
  int form = norm_form();
  if (form == 0) return 0;
  FusedGeom fg;
  int P = reg_pick<T, true>(d, form == 2, &fg);
  if (P == 0 && form == 2) { form = 1; P = reg_pick<T, true>(d, false, &fg); }
  if (P == 0) return 0;
  unsigned* count = nullptr;
  if (form == 2 && fg.splits > 1) {
    count = count_slots(s, d->N * fg.col_blocks);
    if (count == nullptr) form = 1;
  }
  float* part = reinterpret_cast<float*>(ws);
  const int rc = P == 8 ? launch_reg_bwd<T, 8>(form, fg, d, x, stats, dy, dx, part, count, s)
                        : launch_reg_bwd<T, 16>(form, fg, d, x, stats, dy, dx, part, count, s);
  return rc == JPDSE_OK ? 1 : -1;
}

static int validate(const jpdse_inorm_desc* d) {
  JPDSE_REQUIRE(d != nullptr, "inorm: null descriptor");
  JPDSE_REQUIRE(d->dtype == JPDSE_F32 || d->dtype == JPDSE_BF16, "inorm: bad dtype");
  JPDSE_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0, "inorm: non-positive shape");
  JPDSE_REQUIRE(d->act == JPDSE_ACT_NONE || d->act == JPDSE_ACT_RELU || d->act == JPDSE_ACT_LRELU,
                "inorm: unsupported activation %d", d->act);
  return JPDSE_OK;
}

static size_t ws_bytes_for(const jpdse_inorm_desc* d) {
  const int VE = d->dtype == JPDSE_BF16 ? 8 : 4;
  MomentGeom g = moment_geom(d->N, d->H * d->W, cpad(d->C), VE);
  const size_t partial = (size_t)d->N * g.splits * g.Cs * 2 * sizeof(float);
  const size_t sums = (size_t)d->N * g.Cs * 2 * sizeof(float);
  const size_t three = align_up(partial, 256) + align_up(sums, 256), one = reg_ws_bytes(d);
  return three > one ? three : one;
}

template <typename T>
static int inorm_fwd_t(const jpdse_inorm_desc* d, const void* x, const void* res, void* y, float* stats, void* ws,
                       hipStream_t s) {
  constexpr int VE = Vec16<T>::N;
  const int HW = d->H * d->W, Cs = cpad(d->C);
  {
    const int r = try_reg_fwd<T>(d, x, res, y, stats, ws, s);
    if (r != 0) return r > 0 ? JPDSE_OK : JPDSE_ELAUNCH;
  }
  MomentGeom g = moment_geom(d->N, HW, Cs, VE);
  float* partial = reinterpret_cast<float*>(ws);
  const int col_blocks = (g.cv + g.TX - 1) / g.TX;
  hipLaunchKernelGGL((moment_kernel<T, false>), dim3(d->N * g.splits * col_blocks), dim3(256), 0, s,
                     reinterpret_cast<const T*>(x), (const T*)nullptr, (const float*)nullptr, 0, 0.f, partial, g);
  if (int rc = check_launch("inorm moment fwd")) return rc;
  hipLaunchKernelGGL((finalize_fwd_kernel<T>), dim3((d->N * Cs * 8 + 255) / 256), dim3(256), 0, s,
                     reinterpret_cast<const T*>(x), partial, stats, d->N, HW, Cs, g.splits, d->eps);
  if (int rc = check_launch("inorm finalize fwd")) return rc;
  hipLaunchKernelGGL((inorm_apply_fwd_kernel<T>), dim3(d->N * g.splits * col_blocks), dim3(256), 0, s,
                     reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(res), reinterpret_cast<T*>(y), stats,
                     d->act, d->slope, g);
  return check_launch("inorm apply fwd");
}

template <typename T>
static int inorm_bwd_t(const jpdse_inorm_desc* d, const void* x, const float* stats, const void* dy, void* dx,
                       void* ws, hipStream_t s) {
  constexpr int VE = Vec16<T>::N;
  const int HW = d->H * d->W, Cs = cpad(d->C);
  {
    const int r = try_reg_bwd<T>(d, x, stats, dy, dx, ws, s);
    if (r != 0) return r > 0 ? JPDSE_OK : JPDSE_ELAUNCH;
  }
  MomentGeom g = moment_geom(d->N, HW, Cs, VE);
  float* partial = reinterpret_cast<float*>(ws);
  float* sums = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) +
                                         align_up((size_t)d->N * g.splits * g.Cs * 2 * sizeof(float), 256));
  const int col_blocks = (g.cv + g.TX - 1) / g.TX;
  hipLaunchKernelGGL((moment_kernel<T, true>), dim3(d->N * g.splits * col_blocks), dim3(256), 0, s,
                     reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(dy), stats, d->act, d->slope,
                     partial, g);
  if (int rc = check_launch("inorm moment bwd")) return rc;
  hipLaunchKernelGGL(finalize_bwd_kernel, dim3((d->N * Cs * 8 + 255) / 256), dim3(256), 0, s, partial, sums, d->N, HW,
                     Cs, g.splits);
  if (int rc = check_launch("inorm finalize bwd")) return rc;
  hipLaunchKernelGGL((inorm_apply_bwd_kernel<T>), dim3(d->N * g.splits * col_blocks), dim3(256), 0, s,
                     reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(dy), reinterpret_cast<T*>(dx), stats,
                     sums, d->act, d->slope, g);
  return check_launch("inorm apply bwd");
}


// Forward statistics from per-block moments written by a conv epilogue: moments[n][c][slot] = (mean, M2) of the block's
// (bf16-rounded) values, every slot over the same number of pixels HW / slots (common.h).  One wave per (n, c) merges them with
// Chan's parallel-variance formula in two passes over the slots (they are L2 hits): mean = average of the slot means, then
// M2 = sum_slots [M2_s + n_s (mean_s - mean)^2] -- no E[y^2] - mean^2 of un-shifted sums anywhere (the stand-alone moment
// kernels above shift by x[n,0,c] for the same reason).  Lane l takes slots l, l + 64, ... in order, then the fixed xor tree:
// deterministic.
__global__ __launch_bounds__(256) void finalize_slots_kernel(const float* __restrict__ mom, float* __restrict__ stats, int NC,
                                                            int slots, int HW, float eps) {
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pair >= NC) return;
  const float2* const src = reinterpret_cast<const float2*>(mom) + (long long)pair * slots;
  float a = 0.f;
  int s = lane;
  for (; s + 192 < slots; s += 256) {
    const float2 v0 = src[s], v1 = src[s + 64], v2 = src[s + 128], v3 = src[s + 192];
    a += v0.x;
    a += v1.x;
    a += v2.x;
    a += v3.x;
  }
  for (; s < slots; s += 64) a += src[s].x;
  const float mean = wave_sum(a) / (float)slots;
  const float ns = (float)HW / (float)slots;
  float b = 0.f;
  for (s = lane; s < slots; s += 64) {
    const float2 v = src[s];
    const float dm = v.x - mean;
    b += v.y + ns * dm * dm;
  }
  b = wave_sum(b);
  if (lane == 0) {
    float var = b / (float)HW;
    var = var > 0.f ? var : 0.f;
    stats[2 * pair] = mean;
    stats[2 * pair + 1] = rsqrtf(var + eps);
  }
}

// Backward sums from per-block slots written by the epilogue of the data-gradient kernel that produced dy:
// sums[n][c][slot] = (sum dz, sum dz * yhat) over the block's pixels.  (n, c) pairs are summed in slot order by 8 lanes each.
__global__ __launch_bounds__(256) void finalize_bwd_slots_kernel(const float* __restrict__ slots_in, float* __restrict__ sums, int NC,
                                                                int slots, int HW) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int idx = gid >> 3, sub = gid & 7;
  float a = 0.f, b = 0.f;
  if (idx < NC) {
    const float2* const src = reinterpret_cast<const float2*>(slots_in) + (long long)idx * slots;
    for (int q = sub; q < slots; q += 8) {
      const float2 v = src[q];
      a += v.x;
      b += v.y;
    }
  }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
  if (idx >= NC || sub != 0) return;
  const float inv = 1.f / (float)HW;
  sums[2 * idx] = a * inv;
  sums[2 * idx + 1] = b * inv;
}

template <typename T>
static int inorm_bwd_from_sums_t(const jpdse_inorm_desc* d, const void* x, const float* stats, const void* dy, const float* slot_sums,
                                 int slots, void* dx, void* ws, hipStream_t s) {
  constexpr int VE = Vec16<T>::N;
  const int HW = d->H * d->W, Cs = cpad(d->C);
  if (norm_form() != 0 && slots <= 64) {
    FusedGeom fg;
    const int P = reg_pick<T, true>(d, false, &fg);
    if (P == 8) {
      const dim3 grid((unsigned)((size_t)d->N * fg.col_blocks * fg.splits));
      hipLaunchKernelGGL((inorm_reg_bwd_kernel<T, 8, 3>), grid, dim3(256), 0, s, reinterpret_cast<const T*>(x),
                         reinterpret_cast<const T*>(dy), reinterpret_cast<T*>(dx), stats, const_cast<float*>(slot_sums), nullptr,
                         d->act, d->slope, fg, slots);
      return check_launch("inorm apply-from-slots bwd");
    }
  }
  float* const sums = reinterpret_cast<float*>(ws);
  const int NC = d->N * Cs;
  hipLaunchKernelGGL(finalize_bwd_slots_kernel, dim3((NC * 8 + 255) / 256), dim3(256), 0, s, slot_sums, sums, NC, slots, HW);
  if (int rc = check_launch("inorm finalize bwd from slots")) return rc;
  MomentGeom g = moment_geom(d->N, HW, Cs, VE);
  const int col_blocks = (g.cv + g.TX - 1) / g.TX;
  hipLaunchKernelGGL((inorm_apply_bwd_kernel<T>), dim3(d->N * g.splits * col_blocks), dim3(256), 0, s,
                     reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(dy), reinterpret_cast<T*>(dx), stats,
                     sums, d->act, d->slope, g);
  return check_launch("inorm apply bwd");
}

template <typename T>
static int inorm_from_moments_t(const jpdse_inorm_desc* d, const void* x, const float* mom, int slots, const void* res, void* y,
                                float* stats, hipStream_t s) {
  constexpr int VE = Vec16<T>::N;
  const int HW = d->H * d->W, Cs = cpad(d->C);
  const int NC = d->N * Cs;
  // tensors the register-held form covers (<= 64 pixel splits, e.g. the ResnetBlock norms) and few slots: ONE kernel, every
  // block merges its channels' slots itself
  if (norm_form() != 0 && slots <= 64) {
    FusedGeom fg;
    const int P = reg_pick<T, false>(d, false, &fg);
    if (P == 8) {
      const dim3 grid((unsigned)((size_t)d->N * fg.col_blocks * fg.splits));
      hipLaunchKernelGGL((inorm_reg_fwd_kernel<T, 8, 3>), grid, dim3(256), 0, s, reinterpret_cast<const T*>(x),
                         reinterpret_cast<const T*>(res), reinterpret_cast<T*>(y), stats, const_cast<float*>(mom), nullptr,
                         d->act, d->slope, d->eps, fg, slots);
      return check_launch("inorm apply-from-slots fwd");
    }
  }
  hipLaunchKernelGGL(finalize_slots_kernel, dim3((NC + 3) / 4), dim3(256), 0, s, mom, stats, NC, slots, HW, d->eps);
  if (int rc = check_launch("inorm finalize from moments")) return rc;
  MomentGeom g = moment_geom(d->N, HW, Cs, VE);
  const int col_blocks = (g.cv + g.TX - 1) / g.TX;
  hipLaunchKernelGGL((inorm_apply_fwd_kernel<T>), dim3(d->N * g.splits * col_blocks), dim3(256), 0, s,
                     reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(res), reinterpret_cast<T*>(y), stats,
                     d->act, d->slope, g);
  return check_launch("inorm apply fwd");
}

}  // namespace jpdse

using namespace jpdse;

// bytes of one [N][H][W][CPAD(C)] tensor of the descriptor's dtype (the unit of the algorithmic byte counts of jpdse_prof_hbm_*)
static double tensor_bytes(const jpdse_inorm_desc* d) {
  return (double)d->N * d->H * d->W * cpad(d->C) * (double)esize(d->dtype);
}

extern "C" {

size_t jpdse_inorm_workspace_size(const jpdse_inorm_desc* d) {
  if (jpdse::validate(d)) return 0;
  return ws_bytes_for(d);
}

int jpdse_inorm_fwd(const jpdse_inorm_desc* d, const void* x, const void* residual, void* y, float* stats, void* ws,
                    size_t ws_bytes, void* stream) {
  if (int rc = jpdse::validate(d)) return rc;
  JPDSE_REQUIRE(x && y && stats, "inorm_fwd: null pointer");
  JPDSE_REQUIRE(!d->has_residual || residual, "inorm_fwd: has_residual set but residual is null");
  if (ws == nullptr || ws_bytes < ws_bytes_for(d))
    return set_error(JPDSE_EWORKSPACE, "inorm_fwd: workspace %zu < %zu", ws_bytes, ws_bytes_for(d));
  const void* res = d->has_residual ? residual : nullptr;
  const int pslot = hbm_prof_begin(as_stream(stream));
  const int rc = d->dtype == JPDSE_BF16 ? inorm_fwd_t<bf16_t>(d, x, res, y, stats, ws, as_stream(stream))
                                        : inorm_fwd_t<float>(d, x, res, y, stats, ws, as_stream(stream));
  hbm_prof_end(pslot, JPDSE_HBM_INORM_FWD, tensor_bytes(d) * (d->has_residual ? 4.0 : 3.0), as_stream(stream));
  return rc;
}

int jpdse_inorm_bwd(const jpdse_inorm_desc* d, const void* x, const float* stats, const void* dy, void* dx, void* ws,
                    size_t ws_bytes, void* stream) {
  if (int rc = jpdse::validate(d)) return rc;
  JPDSE_REQUIRE(x && stats && dy && dx, "inorm_bwd: null pointer");
  if (ws == nullptr || ws_bytes < ws_bytes_for(d))
    return set_error(JPDSE_EWORKSPACE, "inorm_bwd: workspace %zu < %zu", ws_bytes, ws_bytes_for(d));
  const int pslot = hbm_prof_begin(as_stream(stream));
  const int rc = d->dtype == JPDSE_BF16 ? inorm_bwd_t<bf16_t>(d, x, stats, dy, dx, ws, as_stream(stream))
                                        : inorm_bwd_t<float>(d, x, stats, dy, dx, ws, as_stream(stream));
  hbm_prof_end(pslot, JPDSE_HBM_INORM_BWD, tensor_bytes(d) * 5.0, as_stream(stream));
  return rc;
}

int jpdse_inorm_bwd_from_sums(const jpdse_inorm_desc* d, const void* x, const float* stats, const void* dy, const float* sums,
                              int32_t slots, void* dx, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = jpdse::validate(d)) return rc;
  JPDSE_REQUIRE(x && stats && dy && dx && sums && slots > 0, "inorm_bwd_from_sums: null pointer / no slots");
  const size_t need = align_up((size_t)d->N * cpad(d->C) * 2 * sizeof(float), 256);
  if (ws == nullptr || ws_bytes < need) return set_error(JPDSE_EWORKSPACE, "inorm_bwd_from_sums: workspace %zu < %zu", ws_bytes, need);
  const int pslot = hbm_prof_begin(as_stream(stream));
  const int rc = d->dtype == JPDSE_BF16 ? inorm_bwd_from_sums_t<bf16_t>(d, x, stats, dy, sums, slots, dx, ws, as_stream(stream))
                                        : inorm_bwd_from_sums_t<float>(d, x, stats, dy, sums, slots, dx, ws, as_stream(stream));
  hbm_prof_end(pslot, JPDSE_HBM_INORM_BWD, tensor_bytes(d) * 3.0, as_stream(stream));      // x, dy read once, dx written
  return rc;
}

int jpdse_inorm_fwd_from_moments(const jpdse_inorm_desc* d, const void* x, const float* moments, int32_t slots,
                                 const void* residual, void* y, float* stats, void* stream) {
  if (int rc = jpdse::validate(d)) return rc;
  JPDSE_REQUIRE(x && y && stats && moments && slots > 0, "inorm_fwd_from_moments: null pointer / no slots");
  JPDSE_REQUIRE(!d->has_residual || residual, "inorm_fwd_from_moments: has_residual set but residual is null");
  const void* res = d->has_residual ? residual : nullptr;
  const int pslot = hbm_prof_begin(as_stream(stream));
  const int rc = d->dtype == JPDSE_BF16 ? inorm_from_moments_t<bf16_t>(d, x, moments, slots, res, y, stats, as_stream(stream))
                                        : inorm_from_moments_t<float>(d, x, moments, slots, res, y, stats, as_stream(stream));
  hbm_prof_end(pslot, JPDSE_HBM_INORM_FWD, tensor_bytes(d) * (d->has_residual ? 3.0 : 2.0), as_stream(stream));
  return rc;
}

}  // extern "C"
