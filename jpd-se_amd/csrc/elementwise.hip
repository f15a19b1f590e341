// HBM-bound kernels of the JPD-SE train step: pooling, activation backward, gradient fan-in,
// channel concat/slice, bias-gradient sums, API-boundary layout conversion, the
// one-hot/edge input builder, loss reductions and fused multi-tensor Adam.
// Reference call sites are listed per entry point in include/jpdse.h.
// All kernels move 16-byte vectors per lane along the contiguous NHWC channel axis.
#include "common.h"

namespace jpdse {

#define GRID_STRIDE(idx, total)                                                         \
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < (total); \
       idx += (long long)gridDim.x * blockDim.x)

// ---- pooling ------------------------------------------------------------------------------
template <typename T>
__global__ void avgpool3s2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int OH, int OW,
                                      int Cs, long long total_vec) {
  constexpr int VE = Vec16<T>::N;
  const int cv = Cs / VE;
  GRID_STRIDE(idx, total_vec) {
    const int c = (int)(idx % cv);
    long long t = idx / cv;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    int cnt = 0;
    for (int r = 0; r < 3; ++r) {
      const int ih = 2 * oh - 1 + r;
      if (ih < 0 || ih >= H) continue;
      for (int s = 0; s < 3; ++s) {
        const int iw = 2 * ow - 1 + s;
        if (iw < 0 || iw >= W) continue;
        float v[VE];
        Vec16<T>::load(x + (((long long)n * H + ih) * W + iw) * Cs + c * VE, v);
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] += v[e];
        ++cnt;
      }
    }
    const float inv = 1.f / (float)cnt;
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] *= inv;
    Vec16<T>::store(y + idx * VE, acc);
  }
}

__device__ __forceinline__ int win_count(int o, int L) {  // valid taps of window o along a length-L axis
  const int lo = 2 * o - 1, hi = 2 * o + 1;
  return (hi < L ? hi : L - 1) - (lo > 0 ? lo : 0) + 1;
}

template <typename T>
__global__ void avgpool3s2_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int OH, int OW,
                                      int Cs, long long total_vec) {
  constexpr int VE = Vec16<T>::N;
  const int cv = Cs / VE;
  GRID_STRIDE(idx, total_vec) {
    const int c = (int)(idx % cv);
    long long t = idx / cv;
    const int iw = (int)(t % W);
    t /= W;
    const int ih = (int)(t % H);
    const int n = (int)(t / H);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    const int oh0 = ih / 2, oh1 = (ih + 1) / 2;  // ceil((ih-1)/2) .. floor((ih+1)/2)
    const int ow0 = iw / 2, ow1 = (iw + 1) / 2;
    for (int oh = oh0; oh <= oh1; ++oh) {
      if (oh >= OH) continue;
      for (int ow = ow0; ow <= ow1; ++ow) {
        if (ow >= OW) continue;
        const float inv = 1.f / (float)(win_count(oh, H) * win_count(ow, W));
        float v[VE];
        Vec16<T>::load(dy + (((long long)n * OH + oh) * OW + ow) * Cs + c * VE, v);
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] += v[e] * inv;
      }
    }
    Vec16<T>::store(dx + idx * VE, acc);
  }
}

template <typename T>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int OH, int OW, int Cs,
                                    long long total_vec) {
  constexpr int VE = Vec16<T>::N;
  const int cv = Cs / VE;
  GRID_STRIDE(idx, total_vec) {
    const int c = (int)(idx % cv);
    long long t = idx / cv;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    const T* p = x + (((long long)n * H + 2 * oh) * W + 2 * ow) * Cs + c * VE;
    float a[VE], b[VE], cc[VE], d[VE];
    JPDSE_LOAD_LAST(T, p, a);
    JPDSE_LOAD_LAST(T, p + Cs, b);
    JPDSE_LOAD_LAST(T, p + (long long)W * Cs, cc);
    JPDSE_LOAD_LAST(T, p + (long long)W * Cs + Cs, d);
#pragma unroll
    for (int e = 0; e < VE; ++e) a[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(cc[e], d[e]));
    Vec16<T>::store(y + idx * VE, a);
  }
}

// gradient goes to the FIRST maximal element of each window in row-major order (torch semantics)
template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int H,
                                    int W, int OH, int OW, int Cs, long long total_vec) {
  constexpr int VE = Vec16<T>::N;
  const int cv = Cs / VE;
  GRID_STRIDE(idx, total_vec) {
    const int c = (int)(idx % cv);
    long long t = idx / cv;
    const int iw = (int)(t % W);
    t /= W;
    const int ih = (int)(t % H);
    const int n = (int)(t / H);
    float out[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) out[e] = 0.f;
    const int oh = ih >> 1, ow = iw >> 1;
    if (oh < OH && ow < OW) {
      const T* p = x + (((long long)n * H + 2 * oh) * W + 2 * ow) * Cs + c * VE;
      float w[4][VE], g[VE];
      Vec16<T>::load(p, w[0]);
      Vec16<T>::load(p + Cs, w[1]);
      Vec16<T>::load(p + (long long)W * Cs, w[2]);
      Vec16<T>::load(p + (long long)W * Cs + Cs, w[3]);
      Vec16<T>::load(dy + (((long long)n * OH + oh) * OW + ow) * Cs + c * VE, g);
      const int me = (ih & 1) * 2 + (iw & 1);
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        int best = 0;
        float bv = w[0][e];
#pragma unroll
        for (int k = 1; k < 4; ++k)
          if (w[k][e] > bv) { bv = w[k][e]; best = k; }
        out[e] = (best == me) ? g[e] : 0.f;
      }
    }
    Vec16<T>::store(dx + idx * VE, out);
  }
}

// even H and W: one thread per WINDOW and channel vector -- the four x vectors and the dy vector are loaded once and the four dx
// vectors written (the per-pixel form above loads the window once per pixel: 5 loads per store instead of 5 per 4 stores)
template <typename T>
__global__ void maxpool2_bwd_win_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int H, int W,
                                        int OH, int OW, int Cs, long long total_vec) {
  constexpr int VE = Vec16<T>::N;
  const int cv = Cs / VE;
  GRID_STRIDE(idx, total_vec) {
    const int c = (int)(idx % cv);
    long long t = idx / cv;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    const long long base = (((long long)n * H + 2 * oh) * W + 2 * ow) * Cs + c * VE;
    const long long offs[4] = {0, (long long)Cs, (long long)W * Cs, (long long)W * Cs + Cs};
    float w[4][VE], g[VE], out[4][VE];
#pragma unroll
    for (int k = 0; k < 4; ++k) JPDSE_LOAD_LAST(T, x + base + offs[k], w[k]);
    JPDSE_LOAD_LAST(T, dy + idx * VE, g);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      int best = 0;
      float bv = w[0][e];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (w[k][e] > bv) { bv = w[k][e]; best = k; }
#pragma unroll
      for (int k = 0; k < 4; ++k) out[k][e] = (best == k) ? g[e] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) Vec16<T>::store(dx + base + offs[k], out[k]);
  }
}

// ---- activation backward / add / zero ----------------------------------------------------------
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dz, int act,
                               float slope, long long total_vec) {
  constexpr int VE = Vec16<T>::N;
  GRID_STRIDE(idx, total_vec) {
    float a[VE], g[VE];
    Vec16<T>::load(y + idx * VE, a);
    Vec16<T>::load(dy + idx * VE, g);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      float d = 1.f;
      if (act == JPDSE_ACT_RELU) d = a[e] > 0.f ? 1.f : 0.f;
      else if (act == JPDSE_ACT_LRELU) d = a[e] > 0.f ? 1.f : slope;
      else if (act == JPDSE_ACT_TANH) d = 1.f - a[e] * a[e];
      g[e] *= d;
    }
    Vec16<T>::store(dz + idx * VE, g);
  }
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long long total_vec) {
  constexpr int VE = Vec16<T>::N;
  GRID_STRIDE(idx, total_vec) {
    float x[VE], y[VE];
    Vec16<T>::load(a + idx * VE, x);
    Vec16<T>::load(b + idx * VE, y);
#pragma unroll
    for (int e = 0; e < VE; ++e) x[e] += y[e];
    Vec16<T>::store(out + idx * VE, x);
  }
}

// ---- channel sum / copy ---------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ dy, float* __restrict__ partial,
                                                         long long npix, int Cs, int TX, int TY, long long pix_per_blk) {
  constexpr int VE = Vec16<T>::N;
  __shared__ float red[256 * VE];
  const int cv = Cs / VE;
  const int col_blocks = (cv + TX - 1) / TX;
  const int cb = blockIdx.x % col_blocks;
  const long long pb = blockIdx.x / col_blocks;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int col = cb * TX + tx;
  float s[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) s[e] = 0.f;
  if (col < cv) {
    const long long p0 = pb * pix_per_blk;
    long long p1 = p0 + pix_per_blk;
    p1 = p1 < npix ? p1 : npix;
    long long p = p0 + ty;
    // four loads in flight per thread (one per iteration ran at 1.6 TB/s); the sum order stays p ascending
    for (; p + 3LL * TY < p1; p += 4LL * TY) {
      float v[4][VE];
#pragma unroll
      for (int u = 0; u < 4; ++u) Vec16<T>::load(dy + (p + (long long)u * TY) * Cs + col * VE, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < VE; ++e) s[e] += v[u][e];
    }
    for (; p < p1; p += TY) {
      float v[VE];
      Vec16<T>::load(dy + p * Cs + col * VE, v);
#pragma unroll
      for (int e = 0; e < VE; ++e) s[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < VE; ++e) red[threadIdx.x * VE + e] = s[e];
  __syncthreads();
  if (ty == 0 && col < cv) {
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      float a = 0.f;
      for (int y = 0; y < TY; ++y) a += red[(y * TX + tx) * VE + e];
      partial[pb * Cs + col * VE + e] = a;
    }
  }
}

// 256 threads = 64 channels x 4 groups of partial blocks, 8 independent loads in flight per thread (a serial loop
// over up to 2048 partials per channel was latency-bound: 60 us); fixed summation order
__global__ __launch_bounds__(256) void channel_sum_final_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                               int Cs, int nblk) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a = 0.f;
  if (c < Cs) {
    const int per = (nblk + 3) / 4;
    const int b0 = grp * per;
    int b1 = b0 + per;
    b1 = b1 < nblk ? b1 : nblk;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(long long)(b + u) * Cs + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; b < b1; ++b) a += partial[(long long)b * Cs + c];
  }
  red[grp][cl] = a;
  __syncthreads();
  if (grp == 0 && c < Cs) out[c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

template <typename T>
__global__ void channel_copy_kernel(const T* __restrict__ src, int src_cs, int src_c0, T* __restrict__ dst, int dst_cs,
                                    int dst_c0, int nch, long long total) {
  GRID_STRIDE(idx, total) {
    const int i = (int)(idx % nch);
    const long long p = idx / nch;
    dst[p * dst_cs + dst_c0 + i] = src[p * src_cs + src_c0 + i];
  }
}

// out[p][:] = base[p][:] except channels [c0, c0+nch) = img[p][0..nch): torch.cat((labels, image), dim=1) when the
// label part of `base` is already in place (one pass instead of a copy of base + a channel copy)
template <typename T>
__global__ void concat_channels_kernel(const T* __restrict__ base, const T* __restrict__ img, T* __restrict__ out,
                                       int cs, int img_cs, int c0, int nch, long long total_vec) {
  constexpr int VE = Vec16<T>::N;
  const int cv = cs / VE;
  GRID_STRIDE(idx, total_vec) {
    const int v = (int)(idx % cv);
    const long long p = idx / cv;
    const int cb = v * VE;
    u32x4 val = *reinterpret_cast<const u32x4*>(base + p * cs + cb);
    if (cb + VE > c0 && cb < c0 + nch) {
      T tmp[VE];
      *reinterpret_cast<u32x4*>(tmp) = val;
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const int c = cb + e;
        if (c >= c0 && c < c0 + nch) tmp[e] = img[p * img_cs + (c - c0)];
      }
      val = *reinterpret_cast<const u32x4*>(tmp);
    }
    *reinterpret_cast<u32x4*>(out + p * cs + cb) = val;
  }
}

// ---- layout conversion ---------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, int Cs, long long HW,
                                    long long total) {
  // idx enumerates (n, c_storage, pixel) with the pixel fastest: coalesced reads of one plane
  GRID_STRIDE(idx, total) {
    const long long p = idx % HW;
    long long t = idx / HW;
    const int c = (int)(t % Cs);
    const long long n = t / Cs;
    const float v = c < C ? src[(n * C + c) * HW + p] : 0.f;
    ElemOps<T>::st(dst + (n * HW + p) * Cs + c, v);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int C, int Cs, long long HW,
                                    long long total) {
  GRID_STRIDE(idx, total) {
    const long long p = idx % HW;
    long long t = idx / HW;
    const int c = (int)(t % C);
    const long long n = t / C;
    dst[idx] = ElemOps<T>::ld(src + (n * HW + p) * Cs + c);
  }
}

template <typename T>
__global__ void onehot_edge_kernel(const float* __restrict__ label, const long long* __restrict__ inst,
                                   T* __restrict__ dst, int H, int W, int nlab, int cs, long long total_vec) {
  constexpr int VE = Vec16<T>::N;
  const int cv = cs / VE;
  GRID_STRIDE(idx, total_vec) {
    const int c = (int)(idx % cv);
    const long long pix = idx / cv;  // n*H*W + h*W + w
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    const int lab = (int)(long long)label[pix];
    float v[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = (c * VE + e == lab && lab < nlab) ? 1.f : 0.f;
    if (nlab >= c * VE && nlab < c * VE + VE) {
      const long long me = inst[pix];
      bool edge = false;
      if (w > 0) edge |= inst[pix - 1] != me;
      if (w < W - 1) edge |= inst[pix + 1] != me;
      if (h > 0) edge |= inst[pix - W] != me;
      if (h < H - 1) edge |= inst[pix + W] != me;
      v[nlab - c * VE] = edge ? 1.f : 0.f;
    }
    Vec16<T>::store(dst + idx * VE, v);
  }
}

// The whole input builder in one pass: one-hot labels + instance edge + the image channels, written to up to three
// destinations that differ only in the image (generator input with the decoded frame, discriminator input halves with the
// real / the generated image) -- model.py:375-394 (one-hot, edges), :595 and :456 (the two torch.cat).  img[i] == nullptr
// leaves the image lanes of destination i zero (the generated image does not exist yet: insert_channels_kernel fills it in).
struct BuilderArgs {
  const float* label;
  const long long* inst;
  void* dst[3];
  const void* img[3];
  int n_dst;
  int H, W, nlab, cs, img_cs, c0, nch;
};
// One thread per 16-byte vector, consecutive lanes = consecutive vectors (every store instruction of a wave writes 1 KiB of
// contiguous bytes; a thread-per-pixel form, whose lanes store 80 bytes apart, ran at 2.1 TB/s), 32-bit index arithmetic with
// the vector count per pixel a compile-time constant (the first version did four 64-bit divisions per vector: 1.9 TB/s).
template <typename T, int CV>
__global__ __launch_bounds__(256) void input_builder_kernel(const BuilderArgs a, unsigned total_vec) {
  constexpr int VE = Vec16<T>::N;
  // one vector per thread, no grid-stride loop: behind its three stores (which may alias the label planes as far as the
  // compiler knows) the next iteration's label load could not start, and the loop ran at memory latency per iteration
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx < total_vec) {
    const unsigned pix = idx / (unsigned)CV;
    const int cb = (int)(idx - pix * (unsigned)CV) * VE;
    const int lab = (int)(long long)a.label[pix];
    float v[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = (cb + e == lab && lab < a.nlab) ? 1.f : 0.f;
    if (a.nlab >= cb && a.nlab < cb + VE) {
      const unsigned w = pix % (unsigned)a.W;
      const unsigned h = (pix / (unsigned)a.W) % (unsigned)a.H;
      const long long me = a.inst[pix];
      bool edge = false;
      if (w > 0) edge |= a.inst[pix - 1] != me;
      if (w < (unsigned)a.W - 1) edge |= a.inst[pix + 1] != me;
      if (h > 0) edge |= a.inst[pix - a.W] != me;
      if (h < (unsigned)a.H - 1) edge |= a.inst[pix + a.W] != me;
#pragma unroll
      for (int e = 0; e < VE; ++e)
        if (cb + e == a.nlab) v[e] = edge ? 1.f : 0.f;
    }
    const bool has_img = cb + VE > a.c0 && cb < a.c0 + a.nch;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i >= a.n_dst) break;
      float o[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) o[e] = v[e];
      if (has_img) {
        const T* const img = reinterpret_cast<const T*>(a.img[i]);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const int ch = cb + e;
          if (ch >= a.c0 && ch < a.c0 + a.nch)
            o[e] = img != nullptr ? ElemOps<T>::ld(img + (size_t)pix * a.img_cs + (ch - a.c0)) : 0.f;
        }
      }
      Vec16<T>::store(reinterpret_cast<T*>(a.dst[i]) + (size_t)idx * VE, o);
    }
  }
}

// dst[..., c0 : c0 + nch] = img[..., 0 : nch] in place: only the 16-byte vectors of dst that hold those channels are touched
template <typename T>
__global__ void insert_channels_kernel(T* __restrict__ dst, const T* __restrict__ img, int cs, int img_cs, int c0, int nch,
                                       int v0, int nv, long long total) {
  constexpr int VE = Vec16<T>::N;
  GRID_STRIDE(idx, total) {
    const int v = v0 + (int)(idx % nv);
    const long long p = idx / nv;
    const int cb = v * VE;
    T tmp[VE];
    *reinterpret_cast<u32x4*>(tmp) = *reinterpret_cast<const u32x4*>(dst + p * cs + cb);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const int c = cb + e;
      if (c >= c0 && c < c0 + nch) tmp[e] = img[p * img_cs + (c - c0)];
    }
    *reinterpret_cast<u32x4*>(dst + p * cs + cb) = *reinterpret_cast<const u32x4*>(tmp);
  }
}

// plain device-to-device copy in 16-byte vectors (assembling a batched tensor from two producers)
__global__ __launch_bounds__(256) void copy16_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long long nvec) {
  GRID_STRIDE(i, nvec) dst[i] = src[i];
}

// ---- loss reductions (deterministic two-stage) --------------------------------------------------
enum { RED_L1 = 0, RED_MSE = 1, RED_MSE_CONST = 2 };
static constexpr int kRedBlocks = 1024;

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) t = red[0] + red[1] + red[2] + red[3];
  return t;
}

// da (optional, L1 only): the gradient sign(a-b) * gscale (* [a > 0] when relu_a), written in the same pass
template <typename T, int MODE>
__global__ __launch_bounds__(256) void loss_partial_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                          float target, int cs, float* __restrict__ partial,
                                                          long long total, T* __restrict__ da = nullptr,
                                                          float gscale = 0.f, int relu_a = 0) {
  constexpr int VE = Vec16<T>::N;
  __shared__ float red[4];
  float acc = 0.f;
  if (MODE == RED_MSE_CONST) {
    GRID_STRIDE(idx, total) {  // one logical channel (0) per pixel
      const float d = ElemOps<T>::ld(a + idx * cs) - target;
      acc += d * d;
    }
  } else {
    GRID_STRIDE(idx, total) {
      float x[VE], y[VE];
      Vec16<T>::load(a + idx * VE, x);
      JPDSE_LOAD_LAST(T, b + idx * VE, y);
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const float d = x[e] - y[e];
        acc += (MODE == RED_L1) ? fabsf(d) : d * d;
        if (MODE == RED_L1) {
          const bool dead = relu_a && !(x[e] > 0.f);
          x[e] = dead ? 0.f : (d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f));
        }
      }
      if (MODE == RED_L1 && da != nullptr) Vec16<T>::store(da + idx * VE, x);
    }
  }
  const float t = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ partial, int n, float inv_count,
                                                        float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  const float t = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = t * inv_count;
}

// deferred second stage of up to 32 loss terms in one launch: block t = term t (table by value in the kernel arguments)
struct LossTermTable { jpdse_loss_term t[32]; };
__global__ __launch_bounds__(256) void loss_final_many_kernel(const LossTermTable tab) {
  __shared__ float red[4];
  const jpdse_loss_term e = tab.t[blockIdx.x];
  float acc = 0.f;
  for (int i = threadIdx.x; i < e.n; i += 256) acc += e.partial[i];
  const float t = block_sum_256(acc, red);
  if (threadIdx.x == 0) e.out[0] = t * e.inv_count;
}

// relu_a: `a` is a ReLU output and the gradient is wanted w.r.t. the PRE-activation (masked where a <= 0)
template <typename T, int MODE>
__global__ void loss_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b, float target, int cs,
                                const float* __restrict__ gout, float scale, T* __restrict__ da, long long total,
                                int relu_a) {
  constexpr int VE = Vec16<T>::N;
  const float g = gout[0] * scale;
  if (MODE == RED_MSE_CONST) {
    GRID_STRIDE(idx, total) {  // idx over pixel*cs elements; only channel 0 carries gradient
      const long long p = idx / cs;
      const int c = (int)(idx - p * cs);
      float v = 0.f;
      if (c == 0) v = 2.f * (ElemOps<T>::ld(a + idx) - target) * g;
      ElemOps<T>::st(da + idx, v);
    }
  } else {
    GRID_STRIDE(idx, total) {
      float x[VE], y[VE];
      Vec16<T>::load(a + idx * VE, x);
      Vec16<T>::load(b + idx * VE, y);
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const float d = x[e] - y[e];
        const bool dead = relu_a && !(x[e] > 0.f);
        if (MODE == RED_L1) x[e] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
        else x[e] = 2.f * d * g;
        if (dead) x[e] = 0.f;
      }
      Vec16<T>::store(da + idx * VE, x);
    }
  }
}

// ---- Adam ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(const jpdse_adam_entry* __restrict__ table, int n_entries,
                                                  float lr_over_bc1, float beta1, float beta2, float eps,
                                                  float inv_sqrt_bc2, float grad_scale) {
  // locate the tensor owning this 1024-element block
  const long long blk = blockIdx.x;
  int lo = 0, hi = n_entries - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block0 <= blk) lo = mid;
    else hi = mid - 1;
  }
  const jpdse_adam_entry e = table[lo];
  const long long base = (blk - e.block0) * 1024 + threadIdx.x * 4;
  if (base >= e.n) return;
  if (base + 4 <= e.n && (((uintptr_t)(e.p + base) | (uintptr_t)(e.g + base) | (uintptr_t)(e.m + base) |
                           (uintptr_t)(e.v + base)) & 15) == 0) {
#ifndef JPDSE_NO_NT   // 2.2 GB of optimizer state streamed once per step: nontemporal (5.26 -> 5.67 TB/s, profiles/r03_nontemporal_ab.txt)
    f32x4 p = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(e.p + base));
    f32x4 g = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(e.g + base));
    f32x4 m = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(e.m + base));
    f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(e.v + base));
#else
    f32x4 p = *reinterpret_cast<const f32x4*>(e.p + base);
    f32x4 g = *reinterpret_cast<const f32x4*>(e.g + base);
    f32x4 m = *reinterpret_cast<const f32x4*>(e.m + base);
    f32x4 v = *reinterpret_cast<const f32x4*>(e.v + base);
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float gi = g[i] * grad_scale;
      m[i] = beta1 * m[i] + (1.f - beta1) * gi;
      v[i] = beta2 * v[i] + (1.f - beta2) * gi * gi;
      p[i] -= lr_over_bc1 * m[i] / (sqrtf(v[i]) * inv_sqrt_bc2 + eps);
    }
#ifndef JPDSE_NO_NT
    __builtin_nontemporal_store(p, reinterpret_cast<f32x4*>(e.p + base));
    __builtin_nontemporal_store(m, reinterpret_cast<f32x4*>(e.m + base));
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(e.v + base));
#else
    *reinterpret_cast<f32x4*>(e.p + base) = p;
    *reinterpret_cast<f32x4*>(e.m + base) = m;
    *reinterpret_cast<f32x4*>(e.v + base) = v;
#endif
    if (e.cast_bf16 != nullptr) {
      uint32_t w0 = (uint32_t)f2bf(p[0]) | ((uint32_t)f2bf(p[1]) << 16);
      uint32_t w1 = (uint32_t)f2bf(p[2]) | ((uint32_t)f2bf(p[3]) << 16);
      uint32_t* out = reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(e.cast_bf16) + base);
      __builtin_nontemporal_store(w0, out);         // the forward panel: next read in the next step
      __builtin_nontemporal_store(w1, out + 1);
    }
  } else {
    for (long long i = base; i < base + 4 && i < e.n; ++i) {
      const float gi = e.g[i] * grad_scale;
      const float m = beta1 * e.m[i] + (1.f - beta1) * gi;
      const float v = beta2 * e.v[i] + (1.f - beta2) * gi * gi;
      e.m[i] = m;
      e.v[i] = v;
      const float pn = e.p[i] - lr_over_bc1 * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
      e.p[i] = pn;
      if (e.cast_bf16 != nullptr) reinterpret_cast<bf16_t*>(e.cast_bf16)[i] = f2bf(pn);
    }
  }
}

// ---- fp32 <-> bf16 streams (gradient buckets of the bf16 all-reduce option) ---------------------
// 8 elements per lane: 32 B of fp32 <-> 16 B of bf16
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long total8) {
  GRID_STRIDE(idx, total8) {
    float v[8];
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + idx * 8), b = *reinterpret_cast<const f32x4*>(src + idx * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    Vec16<bf16_t>::store(dst + idx * 8, v);
  }
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long long total8) {
  GRID_STRIDE(idx, total8) {
    float v[8];
    Vec16<bf16_t>::load(src + idx * 8, v);
    const f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(dst + idx * 8) = a;
    *reinterpret_cast<f32x4*>(dst + idx * 8 + 4) = b;
  }
}

// ---- evaluation distortion on uint8-quantised images (ctu/utils/misc.py:64-95, pix2pixHD_model.py:636-641) ------
// q(x) = uint8( clip( (double(x) * std[c] + mean[c]) * 255.0, 0, 255 ) )  -- numpy evaluates this in float64 (the
// std / mean lists become float64 arrays) and astype(uint8) truncates; the same IEEE double operations, unfused, are
// done here, so the quantised images are bit-identical.  The |qa - qb| / (qa - qb)^2 sums are integers, kept exactly
// in double.
struct QuantParams { double mean[8]; double std[8]; };
__device__ __forceinline__ int quant_u8(float x, double sd, double mu) {
  double v = __dmul_rn(__dadd_rn(__dmul_rn((double)x, sd), mu), 255.0);
  v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
  return (int)v;           // truncation, as ndarray.astype(np.uint8) on a value in [0, 255]
}
template <typename TA, typename TB, int MODE>
__global__ __launch_bounds__(256) void quant_loss_partial_kernel(const TA* __restrict__ a, const TB* __restrict__ b,
                                                                int C, int cs, long long npix, QuantParams qp,
                                                                double* __restrict__ partial) {
  __shared__ double red[4];
  double acc = 0.0;
  GRID_STRIDE(p, npix) {
    for (int c = 0; c < C; ++c) {
      const int qa = quant_u8(ElemOps<TA>::ld(a + p * cs + c), qp.std[c], qp.mean[c]);
      const int qb = quant_u8(ElemOps<TB>::ld(b + p * cs + c), qp.std[c], qp.mean[c]);
      const int d = qa - qb;
      acc += (MODE == RED_L1) ? (double)(d < 0 ? -d : d) : (double)(d * d);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void quant_loss_final_kernel(const double* __restrict__ partial, int n, double inv_count,
                                                              float* __restrict__ out) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1] + red[2] + red[3]) * inv_count);
}

static int bad_dtype(int dtype) { return !(dtype == JPDSE_F32 || dtype == JPDSE_BF16); }

#define DISPATCH(dtype, KERNEL, grid, s, ...)                                                    \
  do {                                                                                           \
    if ((dtype) == JPDSE_BF16)                                                                   \
      hipLaunchKernelGGL((KERNEL<bf16_t>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);            \
    else                                                                                         \
      hipLaunchKernelGGL((KERNEL<float>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);             \
  } while (0)

template <typename T> static const T* cptr(const void* p) { return reinterpret_cast<const T*>(p); }
template <typename T> static T* mptr(void* p) { return reinterpret_cast<T*>(p); }

}  // namespace jpdse

using namespace jpdse;

extern "C" {

int jpdse_avgpool3s2_fwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, const void* x, void* y,
                         void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && x && y && N > 0 && H > 0 && W > 0 && C > 0, "avgpool3s2_fwd: bad argument");
  const int Cs = cpad(C), OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long tv = (long long)N * OH * OW * (Cs / (16 / (int)esize(dtype)));
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((avgpool3s2_fwd_kernel<bf16_t>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream),
                       cptr<bf16_t>(x), mptr<bf16_t>(y), H, W, OH, OW, Cs, tv);
  else
    hipLaunchKernelGGL((avgpool3s2_fwd_kernel<float>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream),
                       cptr<float>(x), mptr<float>(y), H, W, OH, OW, Cs, tv);
  return check_launch("avgpool3s2_fwd");
}

int jpdse_avgpool3s2_bwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, const void* dy, void* dx,
                         void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && dy && dx && N > 0 && H > 0 && W > 0 && C > 0, "avgpool3s2_bwd: bad argument");
  const int Cs = cpad(C), OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long tv = (long long)N * H * W * (Cs / (16 / (int)esize(dtype)));
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((avgpool3s2_bwd_kernel<bf16_t>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream),
                       cptr<bf16_t>(dy), mptr<bf16_t>(dx), H, W, OH, OW, Cs, tv);
  else
    hipLaunchKernelGGL((avgpool3s2_bwd_kernel<float>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream),
                       cptr<float>(dy), mptr<float>(dx), H, W, OH, OW, Cs, tv);
  return check_launch("avgpool3s2_bwd");
}

int jpdse_maxpool2_fwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, const void* x, void* y,
                       void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && x && y && N > 0 && H > 1 && W > 1 && C > 0, "maxpool2_fwd: bad argument");
  const int Cs = cpad(C), OH = H / 2, OW = W / 2;
  const long long tv = (long long)N * OH * OW * (Cs / (16 / (int)esize(dtype)));
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((maxpool2_fwd_kernel<bf16_t>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream),
                       cptr<bf16_t>(x), mptr<bf16_t>(y), H, W, OH, OW, Cs, tv);
  else
    hipLaunchKernelGGL((maxpool2_fwd_kernel<float>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream),
                       cptr<float>(x), mptr<float>(y), H, W, OH, OW, Cs, tv);
  return check_launch("maxpool2_fwd");
}

int jpdse_maxpool2_bwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, const void* x, const void* dy,
                       void* dx, void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && x && dy && dx && N > 0 && H > 1 && W > 1 && C > 0, "maxpool2_bwd: bad argument");
  const int Cs = cpad(C), OH = H / 2, OW = W / 2;
  const long long tv = (long long)N * H * W * (Cs / (16 / (int)esize(dtype)));
  if (H % 2 == 0 && W % 2 == 0) {
    const long long wv = tv / 4;
    if (dtype == JPDSE_BF16)
      hipLaunchKernelGGL((maxpool2_bwd_win_kernel<bf16_t>), dim3(ew_blocks(wv)), dim3(256), 0, as_stream(stream),
                         cptr<bf16_t>(x), cptr<bf16_t>(dy), mptr<bf16_t>(dx), H, W, OH, OW, Cs, wv);
    else
      hipLaunchKernelGGL((maxpool2_bwd_win_kernel<float>), dim3(ew_blocks(wv)), dim3(256), 0, as_stream(stream),
                         cptr<float>(x), cptr<float>(dy), mptr<float>(dx), H, W, OH, OW, Cs, wv);
    return check_launch("maxpool2_bwd");
  }
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((maxpool2_bwd_kernel<bf16_t>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream),
                       cptr<bf16_t>(x), cptr<bf16_t>(dy), mptr<bf16_t>(dx), H, W, OH, OW, Cs, tv);
  else
    hipLaunchKernelGGL((maxpool2_bwd_kernel<float>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream),
                       cptr<float>(x), cptr<float>(dy), mptr<float>(dx), H, W, OH, OW, Cs, tv);
  return check_launch("maxpool2_bwd");
}

int jpdse_act_bwd(int32_t dtype, int64_t n, int32_t act, float slope, const void* y, const void* dy, void* dz,
                  void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && y && dy && dz && n > 0, "act_bwd: bad argument");
  const int VE = 16 / (int)esize(dtype);
  JPDSE_REQUIRE(n % VE == 0, "act_bwd: n=%lld not a multiple of %d", (long long)n, VE);
  const long long tv = n / VE;
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((act_bwd_kernel<bf16_t>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream),
                       cptr<bf16_t>(y), cptr<bf16_t>(dy), mptr<bf16_t>(dz), act, slope, tv);
  else
    hipLaunchKernelGGL((act_bwd_kernel<float>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream), cptr<float>(y),
                       cptr<float>(dy), mptr<float>(dz), act, slope, tv);
  return check_launch("act_bwd");
}

int jpdse_add(int32_t dtype, int64_t n, const void* a, const void* b, void* out, void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && a && b && out && n > 0, "add: bad argument");
  const int VE = 16 / (int)esize(dtype);
  JPDSE_REQUIRE(n % VE == 0, "add: n=%lld not a multiple of %d", (long long)n, VE);
  const long long tv = n / VE;
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((add_kernel<bf16_t>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream), cptr<bf16_t>(a),
                       cptr<bf16_t>(b), mptr<bf16_t>(out), tv);
  else
    hipLaunchKernelGGL((add_kernel<float>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream), cptr<float>(a),
                       cptr<float>(b), mptr<float>(out), tv);
  return check_launch("add");
}

int jpdse_zero(int32_t dtype, int64_t n, void* p, void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && p && n >= 0, "zero: bad argument");
  if (n == 0) return JPDSE_OK;
  hipError_t e = hipMemsetAsync(p, 0, (size_t)n * esize(dtype), as_stream(stream));
  if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "zero: %s", hipGetErrorString(e));
  return JPDSE_OK;
}

static void csum_geom(int dtype, int64_t npix, int Cs, int& TX, int& TY, long long& ppb, int& nblk) {
  const int VE = 16 / (int)esize(dtype);
  const int cv = Cs / VE;
  TX = 1;
  while (TX < cv && TX < 256) TX <<= 1;
  TY = 256 / TX;
  nblk = 512;
  if ((long long)nblk * TY * 4 > npix) nblk = (int)((npix + (long long)TY * 4 - 1) / ((long long)TY * 4));
  if (nblk < 1) nblk = 1;
  ppb = (npix + nblk - 1) / nblk;
  nblk = (int)((npix + ppb - 1) / ppb);
}

size_t jpdse_channel_sum_workspace_size(int64_t npix, int32_t C) {
  return (size_t)512 * cpad(C) * sizeof(float);
}

int jpdse_channel_sum(int32_t dtype, int64_t npix, int32_t C, const void* dy, float* out, void* ws, size_t ws_bytes,
                      void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && dy && out && npix > 0 && C > 0, "channel_sum: bad argument");
  const int Cs = cpad(C);
  if (ws == nullptr || ws_bytes < jpdse_channel_sum_workspace_size(npix, C))
    return set_error(JPDSE_EWORKSPACE, "channel_sum: workspace too small");
  int TX, TY, nblk;
  long long ppb;
  csum_geom(dtype, npix, Cs, TX, TY, ppb, nblk);
  const int VE = 16 / (int)esize(dtype);
  const int col_blocks = (Cs / VE + TX - 1) / TX;
  float* partial = reinterpret_cast<float*>(ws);
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((channel_sum_kernel<bf16_t>), dim3(nblk * col_blocks), dim3(256), 0, as_stream(stream),
                       cptr<bf16_t>(dy), partial, (long long)npix, Cs, TX, TY, ppb);
  else
    hipLaunchKernelGGL((channel_sum_kernel<float>), dim3(nblk * col_blocks), dim3(256), 0, as_stream(stream),
                       cptr<float>(dy), partial, (long long)npix, Cs, TX, TY, ppb);
  if (int rc = check_launch("channel_sum")) return rc;
  hipLaunchKernelGGL(channel_sum_final_kernel, dim3((Cs + 63) / 64), dim3(256), 0, as_stream(stream), partial, out,
                     Cs, nblk);
  return check_launch("channel_sum_final");
}

int jpdse_channel_copy(int32_t dtype, int64_t npix, const void* src, int32_t src_cs, int32_t src_c0, void* dst,
                       int32_t dst_cs, int32_t dst_c0, int32_t nch, void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && src && dst && npix > 0 && nch > 0, "channel_copy: bad argument");
  JPDSE_REQUIRE(src_c0 >= 0 && dst_c0 >= 0 && src_c0 + nch <= src_cs && dst_c0 + nch <= dst_cs,
                "channel_copy: channel range out of bounds");
  const long long total = (long long)npix * nch;
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((channel_copy_kernel<bf16_t>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream),
                       cptr<bf16_t>(src), src_cs, src_c0, mptr<bf16_t>(dst), dst_cs, dst_c0, nch, total);
  else
    hipLaunchKernelGGL((channel_copy_kernel<float>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream),
                       cptr<float>(src), src_cs, src_c0, mptr<float>(dst), dst_cs, dst_c0, nch, total);
  return check_launch("channel_copy");
}

int jpdse_concat_channels(int32_t dtype, int64_t npix, const void* base, int32_t cs, const void* img, int32_t img_cs,
                          int32_t c0, int32_t nch, void* out, void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && base && img && out && npix > 0 && nch > 0, "concat_channels: bad argument");
  JPDSE_REQUIRE(cs % 8 == 0 && c0 >= 0 && c0 + nch <= cs && nch <= img_cs, "concat_channels: channel range out of bounds");
  const long long tv = (long long)npix * (cs / (16 / (int)esize(dtype)));
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((concat_channels_kernel<bf16_t>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream),
                       cptr<bf16_t>(base), cptr<bf16_t>(img), mptr<bf16_t>(out), cs, img_cs, c0, nch, tv);
  else
    hipLaunchKernelGGL((concat_channels_kernel<float>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream),
                       cptr<float>(base), cptr<float>(img), mptr<float>(out), cs, img_cs, c0, nch, tv);
  return check_launch("concat_channels");
}

int jpdse_nchw_to_nhwc(int32_t dtype, int32_t N, int32_t C, int32_t H, int32_t W, const float* src, void* dst,
                       void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && src && dst && N > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad argument");
  const int Cs = cpad(C);
  const long long HW = (long long)H * W, total = (long long)N * Cs * HW;
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), src,
                       mptr<bf16_t>(dst), C, Cs, HW, total);
  else
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), src,
                       mptr<float>(dst), C, Cs, HW, total);
  return check_launch("nchw_to_nhwc");
}

int jpdse_nhwc_to_nchw(int32_t dtype, int32_t N, int32_t C, int32_t H, int32_t W, const void* src, float* dst,
                       void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && src && dst && N > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw: bad argument");
  const int Cs = cpad(C);
  const long long HW = (long long)H * W, total = (long long)N * C * HW;
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<bf16_t>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream),
                       cptr<bf16_t>(src), dst, C, Cs, HW, total);
  else
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<float>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream),
                       cptr<float>(src), dst, C, Cs, HW, total);
  return check_launch("nhwc_to_nchw");
}

int jpdse_onehot_edge(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t num_labels, const float* label,
                      const int64_t* instance, void* dst, int32_t cs, void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && label && instance && dst && N > 0 && H > 0 && W > 0, "onehot_edge: bad argument");
  JPDSE_REQUIRE(num_labels > 0 && num_labels < cs && cs % 8 == 0, "onehot_edge: need num_labels < cs, cs %% 8 == 0");
  const int VE = 16 / (int)esize(dtype);
  const long long tv = (long long)N * H * W * (cs / VE);
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((onehot_edge_kernel<bf16_t>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream), label,
                       reinterpret_cast<const long long*>(instance), mptr<bf16_t>(dst), H, W, num_labels, cs, tv);
  else
    hipLaunchKernelGGL((onehot_edge_kernel<float>), dim3(ew_blocks(tv)), dim3(256), 0, as_stream(stream), label,
                       reinterpret_cast<const long long*>(instance), mptr<float>(dst), H, W, num_labels, cs, tv);
  return check_launch("onehot_edge");
}

int jpdse_input_builder(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t num_labels, const float* label,
                        const int64_t* instance, int32_t n_dst, void* const* dst, const void* const* img, int32_t cs,
                        int32_t img_cs, int32_t c0, int32_t nch, void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && label && instance && dst && img && N > 0 && H > 0 && W > 0, "input_builder: bad argument");
  JPDSE_REQUIRE(n_dst >= 1 && n_dst <= 3, "input_builder: 1..3 destinations");
  JPDSE_REQUIRE(num_labels >= 0 && num_labels < cs && cs % 8 == 0 && img_cs % 8 == 0, "input_builder: bad channel counts");
  JPDSE_REQUIRE(nch > 0 && nch <= img_cs && c0 >= 0 && c0 + nch <= cs, "input_builder: image channels [%d, %d) outside 0..%d", c0, c0 + nch, cs);
  BuilderArgs a = {};
  a.label = label;
  a.inst = reinterpret_cast<const long long*>(instance);
  for (int i = 0; i < n_dst; ++i) {
    JPDSE_REQUIRE(dst[i] != nullptr, "input_builder: destination %d is null", i);
    a.dst[i] = dst[i];
    a.img[i] = img[i];
  }
  a.n_dst = n_dst;
  a.H = H;
  a.W = W;
  a.nlab = num_labels;
  a.cs = cs;
  a.img_cs = img_cs;
  a.c0 = c0;
  a.nch = nch;
  const long long npix = (long long)N * H * W;
  const int cv = cs / (16 / (int)esize(dtype));
  JPDSE_REQUIRE(npix * cv < (1LL << 32), "input_builder: more than 2^32 vectors");
  const unsigned grid = (unsigned)((npix * cv + 255) / 256);
  // the vector count per pixel is a template parameter (cheap index arithmetic): 40 storage channels are the hot path
  // (5 bf16 / 10 fp32 vectors), other widths up to 64 channels have their own instantiation
#define JPDSE_BUILDER(T, CVN) hipLaunchKernelGGL((input_builder_kernel<T, CVN>), dim3(grid), dim3(256), 0, as_stream(stream), a, (unsigned)(npix * cv))
  if (dtype == JPDSE_BF16) {
    switch (cv) {
      case 1: JPDSE_BUILDER(bf16_t, 1); break;
      case 2: JPDSE_BUILDER(bf16_t, 2); break;
      case 3: JPDSE_BUILDER(bf16_t, 3); break;
      case 4: JPDSE_BUILDER(bf16_t, 4); break;
      case 5: JPDSE_BUILDER(bf16_t, 5); break;
      case 6: JPDSE_BUILDER(bf16_t, 6); break;
      case 7: JPDSE_BUILDER(bf16_t, 7); break;
      case 8: JPDSE_BUILDER(bf16_t, 8); break;
      default: return set_error(JPDSE_EINVAL, "input_builder: %d storage channels unsupported (<= 64)", cs);
    }
  } else {
    switch (cv) {
      case 2: JPDSE_BUILDER(float, 2); break;
      case 4: JPDSE_BUILDER(float, 4); break;
      case 6: JPDSE_BUILDER(float, 6); break;
      case 8: JPDSE_BUILDER(float, 8); break;
      case 10: JPDSE_BUILDER(float, 10); break;
      case 12: JPDSE_BUILDER(float, 12); break;
      case 14: JPDSE_BUILDER(float, 14); break;
      case 16: JPDSE_BUILDER(float, 16); break;
      default: return set_error(JPDSE_EINVAL, "input_builder: %d storage channels unsupported (<= 64)", cs);
    }
  }
#undef JPDSE_BUILDER
  return check_launch("input_builder");
}

int jpdse_insert_channels(int32_t dtype, int64_t npix, void* dst, int32_t cs, const void* img, int32_t img_cs, int32_t c0,
                          int32_t nch, void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && dst && img && npix > 0, "insert_channels: bad argument");
  JPDSE_REQUIRE(cs % 8 == 0 && img_cs % 8 == 0 && nch > 0 && nch <= img_cs && c0 >= 0 && c0 + nch <= cs,
                "insert_channels: channels [%d, %d) outside 0..%d", c0, c0 + nch, cs);
  const int VE = 16 / (int)esize(dtype);
  const int v0 = c0 / VE, v1 = (c0 + nch - 1) / VE, nv = v1 - v0 + 1;
  const long long total = (long long)npix * nv;
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((insert_channels_kernel<bf16_t>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream),
                       mptr<bf16_t>(dst), cptr<bf16_t>(img), cs, img_cs, c0, nch, v0, nv, total);
  else
    hipLaunchKernelGGL((insert_channels_kernel<float>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream),
                       mptr<float>(dst), cptr<float>(img), cs, img_cs, c0, nch, v0, nv, total);
  return check_launch("insert_channels");
}

int jpdse_copy(int64_t nbytes, const void* src, void* dst, void* stream) {
  JPDSE_REQUIRE(src && dst && nbytes > 0 && nbytes % 16 == 0, "copy: %lld bytes (a positive multiple of 16 is required)", (long long)nbytes);
  JPDSE_REQUIRE((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16 == 0, "copy: pointers must be 16-byte aligned");
  const long long nvec = nbytes / 16;
  hipLaunchKernelGGL(copy16_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, as_stream(stream), reinterpret_cast<const u32x4*>(src),
                     reinterpret_cast<u32x4*>(dst), nvec);
  return check_launch("copy");
}

size_t jpdse_loss_workspace_size(int64_t n) { return kRedBlocks * sizeof(float); }

}  // extern "C"

template <int MODE>
static int loss_fwd(const char* name, int dtype, long long total, long long count, const void* a, const void* b,
                    float target, int cs, float* out, void* ws, size_t ws_bytes, void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && a && total > 0 && count > 0, "%s: bad argument", name);
  if (ws == nullptr || ws_bytes < kRedBlocks * sizeof(float))
    return set_error(JPDSE_EWORKSPACE, "%s: workspace too small", name);
  int grid = ew_blocks(total);
  if (grid > kRedBlocks) grid = kRedBlocks;
  float* partial = reinterpret_cast<float*>(ws);
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((loss_partial_kernel<bf16_t, MODE>), dim3(grid), dim3(256), 0, as_stream(stream),
                       cptr<bf16_t>(a), cptr<bf16_t>(b), target, cs, partial, total);
  else
    hipLaunchKernelGGL((loss_partial_kernel<float, MODE>), dim3(grid), dim3(256), 0, as_stream(stream), cptr<float>(a),
                       cptr<float>(b), target, cs, partial, total);
  if (int rc = check_launch(name)) return rc;
  if (out == nullptr) return JPDSE_OK;                  // deferred: jpdse_loss_finalize reduces the partials
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, grid,
                     1.f / (float)count, out);
  return check_launch(name);
}

template <int MODE>
static int loss_bwd(const char* name, int dtype, long long total, long long count, const void* a, const void* b,
                    float target, int cs, const float* gout, float scale, void* da, void* stream, int relu_a = 0) {
  JPDSE_REQUIRE(!bad_dtype(dtype) && a && gout && da && total > 0 && count > 0, "%s: bad argument", name);
  const float sc = scale / (float)count;
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((loss_bwd_kernel<bf16_t, MODE>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream),
                       cptr<bf16_t>(a), cptr<bf16_t>(b), target, cs, gout, sc, mptr<bf16_t>(da), total, relu_a);
  else
    hipLaunchKernelGGL((loss_bwd_kernel<float, MODE>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream),
                       cptr<float>(a), cptr<float>(b), target, cs, gout, sc, mptr<float>(da), total, relu_a);
  return check_launch(name);
}

static int vec_count(const char* name, int dtype, int64_t n, long long* tv) {
  JPDSE_REQUIRE(!bad_dtype(dtype), "%s: bad dtype", name);
  const int VE = 16 / (int)esize(dtype);
  JPDSE_REQUIRE(n > 0 && n % VE == 0, "%s: n=%lld not a positive multiple of %d", name, (long long)n, VE);
  *tv = n / VE;
  return JPDSE_OK;
}

extern "C" {

int32_t jpdse_loss_partial_count(int64_t work_items) {
  if (work_items <= 0) return 0;
  const int grid = ew_blocks(work_items);
  return grid > kRedBlocks ? kRedBlocks : grid;
}

int jpdse_loss_finalize(const jpdse_loss_term* terms, int32_t n_terms, void* stream) {
  JPDSE_REQUIRE(terms != nullptr && n_terms > 0, "loss_finalize: bad argument");
  for (int32_t t0 = 0; t0 < n_terms; t0 += 32) {
    LossTermTable tab = {};
    const int n = n_terms - t0 < 32 ? n_terms - t0 : 32;
    for (int i = 0; i < n; ++i) {
      const jpdse_loss_term& e = terms[t0 + i];
      JPDSE_REQUIRE(e.partial != nullptr && e.out != nullptr && e.n > 0 && e.n <= kRedBlocks, "loss_finalize: term %d is malformed", t0 + i);
      tab.t[i] = e;
    }
    hipLaunchKernelGGL(loss_final_many_kernel, dim3(n), dim3(256), 0, as_stream(stream), tab);
    if (int rc = check_launch("loss_finalize")) return rc;
  }
  return JPDSE_OK;
}

int jpdse_l1_fwd(int32_t dtype, int64_t n, int64_t count, const void* a, const void* b, float* out, void* ws,
                 size_t ws_bytes, void* stream) {
  long long tv;
  if (int rc = vec_count("l1_fwd", dtype, n, &tv)) return rc;
  JPDSE_REQUIRE(b != nullptr, "l1_fwd: null b");
  return loss_fwd<RED_L1>("l1_fwd", dtype, tv, count, a, b, 0.f, 0, out, ws, ws_bytes, stream);
}
int jpdse_l1_bwd(int32_t dtype, int64_t n, int64_t count, const void* a, const void* b, const float* gout,
                 float scale, void* da, void* stream) {
  long long tv;
  if (int rc = vec_count("l1_bwd", dtype, n, &tv)) return rc;
  JPDSE_REQUIRE(b != nullptr, "l1_bwd: null b");
  return loss_bwd<RED_L1>("l1_bwd", dtype, tv, count, a, b, 0.f, 0, gout, scale, da, stream);
}
int jpdse_l1_fwd_bwd(int32_t dtype, int64_t n, int64_t count, const void* a, const void* b, float* out, float scale,
                     int32_t relu_a, void* da, void* ws, size_t ws_bytes, void* stream) {
  long long tv;
  if (int rc = vec_count("l1_fwd_bwd", dtype, n, &tv)) return rc;
  JPDSE_REQUIRE(a && b && da && count > 0, "l1_fwd_bwd: bad argument");
  if (ws == nullptr || ws_bytes < kRedBlocks * sizeof(float))
    return set_error(JPDSE_EWORKSPACE, "l1_fwd_bwd: workspace too small");
  int grid = ew_blocks(tv);
  if (grid > kRedBlocks) grid = kRedBlocks;
  float* partial = reinterpret_cast<float*>(ws);
  const float gs = scale / (float)count;
  if (dtype == JPDSE_BF16)
    hipLaunchKernelGGL((loss_partial_kernel<bf16_t, RED_L1>), dim3(grid), dim3(256), 0, as_stream(stream),
                       cptr<bf16_t>(a), cptr<bf16_t>(b), 0.f, 0, partial, tv, mptr<bf16_t>(da), gs, relu_a);
  else
    hipLaunchKernelGGL((loss_partial_kernel<float, RED_L1>), dim3(grid), dim3(256), 0, as_stream(stream),
                       cptr<float>(a), cptr<float>(b), 0.f, 0, partial, tv, mptr<float>(da), gs, relu_a);
  if (int rc = check_launch("l1_fwd_bwd")) return rc;
  if (out == nullptr) return JPDSE_OK;                  // deferred: jpdse_loss_finalize reduces the partials
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, grid, 1.f / (float)count, out);
  return check_launch("l1_fwd_bwd");
}
int jpdse_l1_bwd_relu(int32_t dtype, int64_t n, int64_t count, const void* a, const void* b, const float* gout,
                      float scale, void* da, void* stream) {
  long long tv;
  if (int rc = vec_count("l1_bwd_relu", dtype, n, &tv)) return rc;
  JPDSE_REQUIRE(b != nullptr, "l1_bwd_relu: null b");
  return loss_bwd<RED_L1>("l1_bwd_relu", dtype, tv, count, a, b, 0.f, 0, gout, scale, da, stream, 1);
}
int jpdse_mse_fwd(int32_t dtype, int64_t n, int64_t count, const void* a, const void* b, float* out, void* ws,
                  size_t ws_bytes, void* stream) {
  long long tv;
  if (int rc = vec_count("mse_fwd", dtype, n, &tv)) return rc;
  JPDSE_REQUIRE(b != nullptr, "mse_fwd: null b");
  return loss_fwd<RED_MSE>("mse_fwd", dtype, tv, count, a, b, 0.f, 0, out, ws, ws_bytes, stream);
}
int jpdse_mse_bwd(int32_t dtype, int64_t n, int64_t count, const void* a, const void* b, const float* gout,
                  float scale, void* da, void* stream) {
  long long tv;
  if (int rc = vec_count("mse_bwd", dtype, n, &tv)) return rc;
  JPDSE_REQUIRE(b != nullptr, "mse_bwd: null b");
  return loss_bwd<RED_MSE>("mse_bwd", dtype, tv, count, a, b, 0.f, 0, gout, scale, da, stream);
}
int jpdse_mse_const_fwd(int32_t dtype, int64_t npix, int32_t cs, float target, const void* x, float* out, void* ws,
                        size_t ws_bytes, void* stream) {
  JPDSE_REQUIRE(cs > 0, "mse_const_fwd: bad cs");
  return loss_fwd<RED_MSE_CONST>("mse_const_fwd", dtype, npix, npix, x, nullptr, target, cs, out, ws, ws_bytes, stream);
}
int jpdse_mse_const_bwd(int32_t dtype, int64_t npix, int32_t cs, float target, const void* x, const float* gout,
                        float scale, void* dx, void* stream) {
  JPDSE_REQUIRE(cs > 0, "mse_const_bwd: bad cs");
  return loss_bwd<RED_MSE_CONST>("mse_const_bwd", dtype, (long long)npix * cs, npix, x, nullptr, target, cs, gout,
                                 scale, dx, stream);
}

int jpdse_adam_step(const jpdse_adam_entry* table, int32_t n_entries, int64_t total_blocks, float lr, float beta1,
                    float beta2, float eps, int32_t step, float grad_scale, void* stream) {
  JPDSE_REQUIRE(table && n_entries > 0 && total_blocks > 0 && step >= 1, "adam_step: bad argument");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int pslot = hbm_prof_begin(as_stream(stream));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), table, n_entries,
                     (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale);
  // algorithmic bytes: p, m, v read and written, g read = 28 B per parameter (blocks of 1024 parameters; the optional bf16
  // panel copy adds 2 B per parameter of the layers that have one and is not counted)
  hbm_prof_end(pslot, JPDSE_HBM_ADAM, 28.0 * 1024.0 * (double)total_blocks, as_stream(stream));
  return check_launch("adam_step");
}

int jpdse_cast(int32_t src_dtype, int32_t dst_dtype, int64_t n, const void* src, void* dst, void* stream) {
  JPDSE_REQUIRE(!bad_dtype(src_dtype) && !bad_dtype(dst_dtype) && src_dtype != dst_dtype, "cast: fp32 <-> bf16 only");
  JPDSE_REQUIRE(src && dst && n > 0 && n % 8 == 0, "cast: n=%lld must be a positive multiple of 8", (long long)n);
  const long long t8 = n / 8;
  if (src_dtype == JPDSE_F32)
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_blocks(t8)), dim3(256), 0, as_stream(stream), cptr<float>(src),
                       mptr<bf16_t>(dst), t8);
  else
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_blocks(t8)), dim3(256), 0, as_stream(stream), cptr<bf16_t>(src),
                       mptr<float>(dst), t8);
  return check_launch("cast");
}

size_t jpdse_quant_loss_workspace_size(void) { return kRedBlocks * sizeof(double); }

}  // extern "C"

template <typename TA, typename TB>
static int quant_loss_launch(int mse, const void* a, const void* b, int C, int cs, long long npix, const QuantParams& qp,
                             double* partial, int grid, hipStream_t s) {
  if (mse)
    hipLaunchKernelGGL((quant_loss_partial_kernel<TA, TB, RED_MSE>), dim3(grid), dim3(256), 0, s, cptr<TA>(a), cptr<TB>(b),
                       C, cs, npix, qp, partial);
  else
    hipLaunchKernelGGL((quant_loss_partial_kernel<TA, TB, RED_L1>), dim3(grid), dim3(256), 0, s, cptr<TA>(a), cptr<TB>(b),
                       C, cs, npix, qp, partial);
  return check_launch("quant_loss");
}

extern "C" {

int jpdse_quant_loss(int32_t dtype_a, int32_t dtype_b, int64_t npix, int32_t C, const void* a, const void* b,
                     const double* mean, const double* std, int32_t mse, float* out, void* ws, size_t ws_bytes,
                     void* stream) {
  JPDSE_REQUIRE(!bad_dtype(dtype_a) && !bad_dtype(dtype_b) && a && b && out && mean && std && npix > 0,
                "quant_loss: bad argument");
  JPDSE_REQUIRE(C >= 1 && C <= 8, "quant_loss: %d channels (1..8 supported)", C);
  if (ws == nullptr || ws_bytes < kRedBlocks * sizeof(double))
    return set_error(JPDSE_EWORKSPACE, "quant_loss: workspace too small");
  QuantParams qp = {};
  for (int c = 0; c < C; ++c) { qp.mean[c] = mean[c]; qp.std[c] = std[c]; }
  int grid = ew_blocks(npix);
  if (grid > kRedBlocks) grid = kRedBlocks;
  double* partial = reinterpret_cast<double*>(ws);
  hipStream_t s = as_stream(stream);
  const int cs = cpad(C);
  int rc;
  if (dtype_a == JPDSE_BF16 && dtype_b == JPDSE_BF16) rc = quant_loss_launch<bf16_t, bf16_t>(mse, a, b, C, cs, npix, qp, partial, grid, s);
  else if (dtype_a == JPDSE_BF16) rc = quant_loss_launch<bf16_t, float>(mse, a, b, C, cs, npix, qp, partial, grid, s);
  else if (dtype_b == JPDSE_BF16) rc = quant_loss_launch<float, bf16_t>(mse, a, b, C, cs, npix, qp, partial, grid, s);
  else rc = quant_loss_launch<float, float>(mse, a, b, C, cs, npix, qp, partial, grid, s);
  if (rc) return rc;
  hipLaunchKernelGGL(quant_loss_final_kernel, dim3(1), dim3(256), 0, s, partial, grid, 1.0 / ((double)npix * C), out);
  return check_launch("quant_loss");
}

}  // extern "C"
