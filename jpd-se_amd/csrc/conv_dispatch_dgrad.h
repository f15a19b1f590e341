// Data-gradient dispatch of the convolution family (also the forward of ConvTranspose2d): reflect ring path, stride-phase
// programs, single-phase layers.  Part of conv_gemm.hip (one translation unit).
#pragma once

namespace jpdse {

// Reflect-padded 3x3 stride-1 data gradient = zero-padded data gradient (halo kernel, written to dx)
// + the ring of the padded domain folded back: padded row -1 -> image row 1, row H -> H-2, column -1 -> 1,
// column W -> W-2.  The four ring strips are small split-K GEMMs (fp32 slabs, FastArgs::no_finish); this
// kernel sums their slabs and adds them into dx.  One thread = 8 channels of one target pixel; targets are
// enumerated without duplicates: rows {1, H-2} completely, columns {1, W-2} without those two rows.
struct RingFoldArgs {
  bf16_t* dx;
  const float* top; const float* bot; const float* left; const float* right;   // slabs [splits][M_q][Cs]
  int splits_tb, splits_lr;
  int N, H, W, Cs;
};
__device__ __forceinline__ void ring_acc(float (&acc)[8], const float* slab, int splits, long long slab_elems,
                                         long long row, int Cs, int c0) {
  // slabs are read four at a time before they are added (same order): one L2 round trip per split otherwise
  const float* base = slab + row * Cs + c0;
  int sp = 0;
  for (; sp + 4 <= splits; sp += 4) {
    float4 lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4* src = reinterpret_cast<const float4*>(base + (sp + u) * slab_elems);
      lo[u] = src[0];
      hi[u] = src[1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] += lo[u].x; acc[1] += lo[u].y; acc[2] += lo[u].z; acc[3] += lo[u].w;
      acc[4] += hi[u].x; acc[5] += hi[u].y; acc[6] += hi[u].z; acc[7] += hi[u].w;
    }
  }
  for (; sp < splits; ++sp) {
    const float4* src = reinterpret_cast<const float4*>(base + sp * slab_elems);
    const float4 lo = src[0], hi = src[1];
    acc[0] += lo.x; acc[1] += lo.y; acc[2] += lo.z; acc[3] += lo.w;
    acc[4] += hi.x; acc[5] += hi.y; acc[6] += hi.z; acc[7] += hi.w;
  }
}
__global__ __launch_bounds__(256) void ring_fold_kernel(const RingFoldArgs a, long long total_vec) {
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= total_vec) return;
  const int cv = a.Cs >> 3;
  const int c0 = (int)(v % cv) * 8;
  long long t = v / cv;
  const int per_n = 2 * a.W + 2 * (a.H - 2);
  const int n = (int)(t / per_n);
  int e = (int)(t - (long long)n * per_n);
  const long long Mtb = (long long)a.N * (a.W + 2), Mlr = (long long)a.N * a.H;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  int i_row, j_col;
  if (e < 2 * a.W) {                      // row targets
    const bool is_top = e < a.W;
    const int j = is_top ? e : e - a.W;
    i_row = is_top ? 1 : a.H - 2;
    j_col = j;
    const float* slab = is_top ? a.top : a.bot;
    const long long rb = (long long)n * (a.W + 2);
    ring_acc(acc, slab, a.splits_tb, Mtb * a.Cs, rb + j + 1, a.Cs, c0);
    if (j == 1) ring_acc(acc, slab, a.splits_tb, Mtb * a.Cs, rb, a.Cs, c0);                  // corner b = 0
    if (j == a.W - 2) ring_acc(acc, slab, a.splits_tb, Mtb * a.Cs, rb + a.W + 1, a.Cs, c0);  // corner b = W+1
    if (j == 1) ring_acc(acc, a.left, a.splits_lr, Mlr * a.Cs, (long long)n * a.H + i_row, a.Cs, c0);
    if (j == a.W - 2) ring_acc(acc, a.right, a.splits_lr, Mlr * a.Cs, (long long)n * a.H + i_row, a.Cs, c0);
  } else {                                // column targets, rows other than 1 and H-2
    e -= 2 * a.W;
    const bool is_left = e < a.H - 2;
    int i = is_left ? e : e - (a.H - 2);  // index into the H-2 remaining rows
    i = i == 0 ? 0 : i + 1;               // rows 0, 2, 3, ..., H-3, H-1
    if (i >= a.H - 2) i += 1;
    i_row = i;
    j_col = is_left ? 1 : a.W - 2;
    ring_acc(acc, is_left ? a.left : a.right, a.splits_lr, Mlr * a.Cs, (long long)n * a.H + i, a.Cs, c0);
  }
  bf16_t* dst = a.dx + (((long long)n * a.H + i_row) * a.W + j_col) * a.Cs + c0;
  float cur[8];
  Vec16<bf16_t>::load(dst, cur);
#pragma unroll
  for (int q = 0; q < 8; ++q) cur[q] += acc[q];
  Vec16<bf16_t>::store(dst, cur);
}

// The folded frame of dy for the one-launch reflect data gradient (gemm_halo.h, VIRT): per image 2 rows of W + 2 pixels
// (above row 0 / below row H-1), then 2 columns of H pixels (left of column 0 / right of column W-1):
//   row frame t, column b-1:  dy[r0][c] + dy[r1][c] with (r0, r1) = (0, 2) | (H-3, H-1); at b = 0 and b = W+1 the corner, summed
//                             over the column pair (0, 2) | (W-3, W-1) as well
//   column frame t, row h:    dy[h][c0] + dy[h][c1]
// fp32 sums, rounded to bf16 once.  One thread = 8 channels of one frame pixel.
__global__ __launch_bounds__(256) void ring_frame_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ V, int N, int H, int W,
                                                         int Cs, long long total_vec) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total_vec) return;
  const int cv = Cs >> 3;
  const int c0 = (int)(t % cv) * 8;
  long long q = t / cv;
  const int per_n = 2 * (W + 2) + 2 * H;
  const int n = (int)(q / per_n);
  const int e = (int)(q - (long long)n * per_n);
  int rows[2], cols[2], nr, nc;
  if (e < 2 * (W + 2)) {
    const int top = e < W + 2, b = top ? e : e - (W + 2);
    rows[0] = top ? 0 : H - 3;
    rows[1] = top ? 2 : H - 1;
    nr = 2;
    if (b == 0) { cols[0] = 0; cols[1] = 2; nc = 2; }
    else if (b == W + 1) { cols[0] = W - 3; cols[1] = W - 1; nc = 2; }
    else { cols[0] = b - 1; cols[1] = 0; nc = 1; }
  } else {
    const int f = e - 2 * (W + 2);
    const int left = f < H;
    rows[0] = left ? f : f - H;
    rows[1] = 0;
    nr = 1;
    cols[0] = left ? 0 : W - 3;
    cols[1] = left ? 2 : W - 1;
    nc = 2;
  }
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int r = 0; r < nr; ++r)
    for (int c = 0; c < nc; ++c) {
      float v[8];
      Vec16<bf16_t>::load(dy + (((long long)n * H + rows[r]) * W + cols[c]) * Cs + c0, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
  Vec16<bf16_t>::store(V + ((long long)n * per_n + e) * Cs + c0, acc);
}

// dx = (dx + addend) * (mask > 0 ? 1 : slope), 16-byte vectors: the unfused form of the data-gradient epilogue
// extras (either pointer may be null); the sum is rounded to T before the slope, as the fused epilogues do
template <typename T>
__global__ void relu_mask_kernel(T* __restrict__ dx, const T* __restrict__ mask, long long total_vec,
                                 const T* __restrict__ addend = nullptr, float slope = 0.f) {
  constexpr int VE = Vec16<T>::N;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total_vec;
       i += (long long)gridDim.x * blockDim.x) {
    float v[VE], m[VE];
    Vec16<T>::load(dx + i * VE, v);
    if (addend != nullptr) {
      Vec16<T>::load(addend + i * VE, m);
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] += m[e];
    }
    if (mask != nullptr) {
      Vec16<T>::load(mask + i * VE, m);
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const float r = sizeof(T) == 2 ? bf16_round(v[e]) : v[e];
        v[e] = m[e] > 0.f ? v[e] : (slope == 0.f ? 0.f : r * slope);
      }
    }
    Vec16<T>::store(dx + i * VE, v);
  }
}

// The InstanceNorm (+ activation) whose backward consumes this data gradient (jpdse_conv_dgrad_fused_nsums): its input, its
// (mean, rstd), its activation, and where the per-block sums of its backward go (gemm_halo.h NSUM).
struct NormSink {
  const void* x;
  const float* stats;
  float* sums;
  int act;
  float slope;
};

// blocks per image whose norm-backward sums the data gradient of this layer writes (0: its kernel has no such epilogue):
// the reflect-padded 3x3 convs on the folded-frame halo kernel with 128-channel output tiles -- the ResnetBlock convs
static int dgrad_nsum_slots(const jpdse_conv_desc* d, const ConvPlan& p) {
  if (d->dtype != JPDSE_BF16 || d->pad_mode != JPDSE_PAD_REFLECT || d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1) return 0;
  if (!g_ring_enabled || !g_ring_virt || d->H < 8 || p.ph[0].Lk != 3 * p.Ks || p.Ks < 128 || p.Cs <= 64 || p.Cs % 128 != 0 || d->C != p.Cs) return 0;
  if (!halo_ok(3, 3, 1, d->H, d->W, p.Ks, p.Cs)) return 0;
  return (d->H / 4) * (d->W / 64);
}

template <typename T>
static int conv_dgrad_t(const jpdse_conv_desc* d, const ConvPlan& p, const void* dy, const void* pack, void* dx,
                        void* ws, hipStream_t s, const void* mask = nullptr, const void* addend = nullptr, float* mom = nullptr,
                        float mask_slope = 0.f, const NormSink* sink = nullptr) {
  if (sink != nullptr && (sizeof(T) != 2 || dgrad_nsum_slots(d, p) == 0 || (mask != nullptr && mask_slope != 0.f)))
    return set_error(JPDSE_EINVAL, "conv_dgrad: this layer's data-gradient kernel writes no norm-backward sums (jpdse_conv_dgrad_nsum_slots == 0)");
  char* wsb = reinterpret_cast<char*>(ws);
  void* dyp = wsb;
  void* dxp = wsb + p.dypad_bytes;
  const bool refl = d->pad_mode == JPDSE_PAD_REFLECT;
  const int st = d->stride;
  // LeakyReLU-backward epilogue (mask_slope != 0): only the generic tile kernels and the unfused pass apply it
  const bool lrelu = mask != nullptr && mask_slope != 0.f;
  if constexpr (sizeof(T) == 2) {
    if (!lrelu && !refl && p.nph == 1 && st == 1 && p.ph[0].cnth == d->H && p.ph[0].cntw == d->W &&
        rows_ok(p.ph[0].Uh, p.ph[0].Uw, 1, 0, JPDSE_ACT_NONE, d->H, d->W, p.Ks, p.Cs) && p.ph[0].Lk == 3 * p.Ks) {
      const Phase& f = p.ph[0];
      RowsArgs r = {};
      r.X = reinterpret_cast<const bf16_t*>(dy);
      r.B = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pack) + f.pack_off);
      r.Y = reinterpret_cast<bf16_t*>(dx);
      r.N = d->N;
      r.OH = d->H;
      r.OW = d->W;
      r.IH = p.OH;
      r.IW = p.OW;
      r.py = (f.Uh - 1) - f.i0h;
      r.px = (f.Uw - 1) - f.i0w;
      r.Kout = d->C;
      r.Ks = p.Cs;
      r.b_rows = p.Cs;
      r.out_sn = (long long)d->H * d->W * p.Cs;
      r.out_sh = (long long)d->W * p.Cs;
      r.out_sw = p.Cs;
      r.out_base = 0;
      r.act = JPDSE_ACT_NONE;
      r.mask = reinterpret_cast<const bf16_t*>(mask);
      r.addend = reinterpret_cast<const bf16_t*>(addend);
      return launch_rows(r, 1, s);
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (!lrelu && !refl && p.nph == 1 && p.ph[0].cnth == d->H && p.ph[0].cntw == d->W &&
        halo_ok(p.ph[0].Uh, p.ph[0].Uw, st, d->H, d->W, p.Ks, p.Cs)) {
      const Phase& f = p.ph[0];
      HaloArgs h = {};
      h.X = reinterpret_cast<const bf16_t*>(dy);
      h.B = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pack) + f.pack_off);
      h.bias = nullptr;
      h.Y = reinterpret_cast<bf16_t*>(dx);
      h.N = d->N;
      h.OH = d->H;
      h.OW = d->W;
      h.IH = p.OH;
      h.IW = p.OW;
      h.Cs = p.Ks;
      h.py = (f.Uh - 1) - f.i0h;
      h.px = (f.Uw - 1) - f.i0w;
      h.reflect = 0;
      h.Kout = d->C;
      h.Ks = p.Cs;
      h.b_rows = p.Cs;
      h.out_sn = (long long)d->H * d->W * p.Cs;
      h.out_sh = (long long)d->W * p.Cs;
      h.out_sw = p.Cs;
      h.out_base = 0;
      h.act = JPDSE_ACT_NONE;
      h.mask = reinterpret_cast<const bf16_t*>(mask);
      h.addend = reinterpret_cast<const bf16_t*>(addend);
      return p.Cs > 64 ? launch_halo_cfg<2>(h, s) : launch_halo_cfg<1>(h, s);
    }
    // the same for grids of 8 x 32 patches (W = 32): tap-program kernel, nine taps, split-K over the dy channel slabs
    if (!lrelu && !refl && p.nph == 1 && st == 1 && p.ph[0].cnth == d->H && p.ph[0].cntw == d->W && p.ph[0].Lk == p.ph[0].Uw * p.Ks &&
        taps9_shape_ok(p.ph[0].Uh, p.ph[0].Uw, 1, d->H, d->W, p.Ks, p.Cs, (long long)d->N * p.OH * p.OW * p.Ks, (long long)p.Cs * 9 * p.Ks)) {
      const Phase& f = p.ph[0];
      Taps4View v = {};
      v.X = reinterpret_cast<const bf16_t*>(dy);
      v.B = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pack) + f.pack_off);
      v.Y = reinterpret_cast<bf16_t*>(dx);
      v.N = d->N;
      v.IH = p.OH;
      v.IW = p.OW;
      v.Cin_s = p.Ks;
      v.OH = d->H;
      v.OW = d->W;
      v.py = (f.Uh - 1) - f.i0h;
      v.px = (f.Uw - 1) - f.i0w;
      v.Kout = d->C;
      v.Ks_out = p.Cs;
      v.ktot = (long long)f.Uh * f.Lk;
      v.tap_r = f.Lk;
      v.tap_s = p.Ks;
      v.act = JPDSE_ACT_NONE;
      v.addend = reinterpret_cast<const bf16_t*>(addend);
      v.mask = reinterpret_cast<const bf16_t*>(mask);
      return launch_taps9(v, reinterpret_cast<float*>(wsb + p.splitk_off), s);
    }
  }
  if constexpr (sizeof(T) == 2) {
    const bool ring_halo = halo_ok(3, 3, 1, d->H, d->W, p.Ks, p.Cs);
    const bool ring_taps = !ring_halo && taps9_shape_ok(3, 3, 1, d->H, d->W, p.Ks, p.Cs, (long long)d->N * d->H * d->W * p.Ks,
                                                        (long long)p.Cs * 9 * p.Ks);
    if (!lrelu && refl && g_ring_enabled && d->R == 3 && d->S == 3 && st == 1 && d->pad == 1 && d->H >= 8 &&
        (ring_halo || ring_taps) && p.ph[0].Lk == 3 * p.Ks) {
      // (1) zero-padded data gradient straight into dx: halo kernel, or the nine-tap program on 8 x 32 patches (W = 32)
      if (ring_taps) {
        Taps4View v = {};
        v.X = reinterpret_cast<const bf16_t*>(dy);
        v.B = reinterpret_cast<const bf16_t*>(pack);
        v.Y = reinterpret_cast<bf16_t*>(dx);
        v.N = d->N;
        v.IH = d->H;
        v.IW = d->W;
        v.Cin_s = p.Ks;
        v.OH = d->H;
        v.OW = d->W;
        v.py = v.px = 1;
        v.Kout = d->C;
        v.Ks_out = p.Cs;
        v.ktot = 3LL * p.ph[0].Lk;
        v.tap_r = p.ph[0].Lk;
        v.tap_s = p.Ks;
        v.act = JPDSE_ACT_NONE;
        v.addend = reinterpret_cast<const bf16_t*>(addend);
        if (g_ring_virt && d->H >= 16) {
          // one launch (+ the split-K finish): the ring rides in the frame of dy (gemm_taps.h VIRT)
          bf16_t* frame = reinterpret_cast<bf16_t*>(wsb);       // in front of the split-K slabs (splitk_off lies behind the padded-dy region)
          const long long fv = (long long)d->N * (2 * (d->W + 2) + 2 * d->H) * (p.Ks / 8);
          if ((size_t)fv * 16 > p.splitk_off) return set_error(JPDSE_EWORKSPACE, "reflect data gradient: no room for the frame in front of the slabs");
          hipLaunchKernelGGL(ring_frame_kernel, dim3(ew_blocks(fv)), dim3(256), 0, s, v.X, frame, d->N, d->H, d->W, p.Ks, fv);
          if (int rc = check_launch("ring_frame_kernel")) return rc;
          v.frame = frame;
          v.mask = reinterpret_cast<const bf16_t*>(mask);
          return launch_taps9(v, reinterpret_cast<float*>(wsb + p.splitk_off), s);
        }
        if (int rc = launch_taps9(v, reinterpret_cast<float*>(wsb + p.splitk_off), s)) return rc;
      }
      HaloArgs h = {};
      h.X = reinterpret_cast<const bf16_t*>(dy);
      h.B = reinterpret_cast<const bf16_t*>(pack);
      h.Y = reinterpret_cast<bf16_t*>(dx);
      h.N = d->N;
      h.OH = d->H;
      h.OW = d->W;
      h.IH = d->H;
      h.IW = d->W;
      h.Cs = p.Ks;
      h.py = h.px = 1;
      h.Kout = d->C;
      h.Ks = p.Cs;
      h.b_rows = p.Cs;
      h.out_sn = (long long)d->H * d->W * p.Cs;
      h.out_sh = (long long)d->W * p.Cs;
      h.out_sw = p.Cs;
      h.act = JPDSE_ACT_NONE;
      h.addend = reinterpret_cast<const bf16_t*>(addend);
      if (ring_halo && g_ring_virt && (p.Ks >= 128 || (p.Ks == 64 && g_halo_single))) {
        // one launch: the ring rides in the frame of dy (gemm_halo.h, VIRT); the mask, if any, in the same epilogue
        bf16_t* frame = reinterpret_cast<bf16_t*>(wsb);
        const long long fv = (long long)d->N * (2 * (d->W + 2) + 2 * d->H) * (p.Ks / 8);
        const int pslot = (p.Cs == g_prof.Ks && 9LL * p.Ks == g_prof.kdim) ? prof_begin(s) : -1;
        hipLaunchKernelGGL(ring_frame_kernel, dim3(ew_blocks(fv)), dim3(256), 0, s, h.X, frame, d->N, d->H, d->W, p.Ks, fv);
        int rc = check_launch("ring_frame_kernel");
        prof_end(pslot, 1, 0.0, s);
        if (rc) return rc;
        h.V = frame;
        h.mask = reinterpret_cast<const bf16_t*>(mask);
        if (sink != nullptr) {
          h.nx = reinterpret_cast<const bf16_t*>(sink->x);
          h.nstats = sink->stats;
          h.nsums = sink->sums;
          h.nact = sink->act;
          h.nslope = sink->slope;
          h.mom_slots = (d->H / 4) * (d->W / 64);
        }
        return p.Cs > 64 ? launch_halo_cfg<2>(h, s) : launch_halo_cfg<1>(h, s);
      }
      if (ring_halo)
        if (int rc = p.Cs > 64 ? launch_halo_cfg<2>(h, s) : launch_halo_cfg<1>(h, s)) return rc;
      // (2) the four ring strips of the reflect-padded domain as split-K GEMMs into fp32 slabs
      const int H = d->H, W = d->W, Ks = p.Ks, Lk = p.ph[0].Lk;
      const bf16_t* dyb = reinterpret_cast<const bf16_t*>(dy);
      const bf16_t* pk = reinterpret_cast<const bf16_t*>(pack);
      const int Mtb = d->N * (W + 2), Mlr = d->N * H;
      const int nt = (p.Cs + 127) / 128;
      // the strips have few rows (N (W + 2) and N H); 128-row tiles (two blocks per CU) measured slower (round-2 mode 31, retired)
      const int bm = 256;
      const int tiles = 2 * ((Mtb + bm - 1) / bm) * nt + 2 * ((Mlr + bm - 1) / bm) * nt;
      const int kt = 3 * Ks / 64;
      int sp = 256 / tiles;
      if (sp > kt / 8) sp = kt / 8;
      if (sp > 8) sp = 8;
      if (sp < 1) sp = 1;
      float* slab = reinterpret_cast<float*>(wsb);
      const size_t tb_elems = (size_t)sp * Mtb * p.Cs, lr_elems = (size_t)sp * Mlr * p.Cs;
      FastBatch rb = {};
      rb.small_m = 0;
      for (int q = 0; q < 4; ++q) {
        FastArgs g = {};
        const bool row_strip = q < 2;         // 0 top, 1 bottom, 2 left, 3 right
        g.X = dyb + (q == 1 ? (long long)(H - 1) * W * Ks : (q == 3 ? (long long)(W - 1) * Ks : 0));
        g.x_sn = (long long)H * W * Ks;
        g.x_sh = (long long)W * Ks;
        g.x_extent = (long long)d->N * H * W * Ks - (g.X - dyb);
        g.IH = row_strip ? 1 : H;
        g.IW = row_strip ? W : 1;
        g.Cs = Ks;
        g.R = row_strip ? 1 : 3;
        g.S = row_strip ? 3 : 1;
        g.sy = g.sx = 1;
        g.py = row_strip ? 0 : 1;
        g.px = row_strip ? 2 : 0;
        g.OH = row_strip ? 1 : H;
        g.OW = row_strip ? W + 2 : 1;
        g.M = row_strip ? Mtb : Mlr;
        // panel [c][u'][w'][k] with u' = 2 - r, w' = 2 - s: top r=0 -> u'=2, bottom u'=0, left s=0 -> w'=2, right w'=0
        g.B = pk + (q == 0 ? 2LL * Lk : (q == 2 ? 2LL * Ks : 0));
        g.b_stride = 3LL * Lk;
        g.b_tap_r = Lk;
        g.b_tap_s = Ks;
        g.Kout = d->C;
        g.Ks = p.Cs;
        g.b_rows = p.Cs;
        g.act = JPDSE_ACT_NONE;
        g.splits = sp;
        g.no_finish = 1;
        g.partial = slab + (q == 0 ? 0 : (q == 1 ? tb_elems : (q == 2 ? 2 * tb_elems : 2 * tb_elems + lr_elems)));
        rb.p[rb.n++] = g;
      }
      const int pslot = (p.Cs == g_prof.Ks && 9LL * p.Ks == g_prof.kdim) ? prof_begin(s) : -1;
      if (int rc = launch_fast_batch(rb, s)) return rc;
      // (3) fold the ring into rows 1 / H-2 and columns 1 / W-2 of dx
      RingFoldArgs rf = {};
      rf.dx = reinterpret_cast<bf16_t*>(dx);
      rf.top = rb.p[0].partial;
      rf.bot = rb.p[1].partial;
      rf.left = rb.p[2].partial;
      rf.right = rb.p[3].partial;
      rf.splits_tb = rf.splits_lr = sp;
      rf.N = d->N;
      rf.H = H;
      rf.W = W;
      rf.Cs = p.Cs;
      const long long tv = (long long)d->N * (2 * W + 2 * (H - 2)) * (p.Cs / 8);
      hipLaunchKernelGGL(ring_fold_kernel, dim3(ew_blocks(tv)), dim3(256), 0, s, rf, tv);
      int rc = check_launch("ring_fold_kernel");
      if (rc == JPDSE_OK && mask != nullptr) {
        const long long total_vec = (long long)d->N * H * W * (p.Cs / 8);
        hipLaunchKernelGGL((relu_mask_kernel<T>), dim3(ew_blocks(total_vec)), dim3(256), 0, s, reinterpret_cast<T*>(dx),
                           reinterpret_cast<const T*>(mask), total_vec);
        rc = check_launch("relu_mask_kernel");
      }
      prof_end(pslot, 1, 0.0, s);
      return rc;
    }
  }
  if constexpr (sizeof(T) == 2) {
    // few INPUT channels (VGG conv1_1: 3 <- 64): the data gradient is itself a conv with <= 3 output channels; the
    // single-phase dgrad panel [c][u'][w'][k] is exactly the "plain forward panel" head_fwd_kernel expects
    if (g_fast_enabled && g_head_fwd_enabled && !refl && st == 1 && d->R == 3 && d->S == 3 && d->pad == 1 && d->C <= 3 &&
        p.Cs == 8 && p.Ks == 64 && p.nph == 1 && p.ph[0].Lk == 3 * p.Ks && mask == nullptr && addend == nullptr) {
      HeadFwdArgs h = {};
      h.X = reinterpret_cast<const bf16_t*>(dy);
      h.Wp = reinterpret_cast<const bf16_t*>(pack);
      h.Y = reinterpret_cast<bf16_t*>(dx);
      h.N = d->N;
      h.H = p.OH;
      h.W = p.OW;
      h.OH = d->H;
      h.OW = d->W;
      h.K = d->C;
      h.Ks_out = p.Cs;
      h.R = 3;
      h.S = 3;
      h.pad = 1;
      h.act = JPDSE_ACT_NONE;
      h.tiles_w = (d->W + 63) / 64;
      h.tiles_h = (d->H + kHeadTH - 1) / kHeadTH;
      if (head_rows_ok(h, 64)) return launch_head_rows<3>(h, s);
      return launch_head_fwd<64, 3, 3, 3>(h, s);
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (g_fast_enabled && g_rows_enabled && !refl && st == 2 && d->R == 3 && d->S == 3 && d->pad == 1 && p.Ks == 128 && p.Cs == 64 &&
        d->C == 64 && d->H == 2 * p.OH && d->W == 2 * p.OW && p.OW % 64 == 0 && p.OH % 4 == 0 && p.nph == 4 &&
        mask == nullptr && addend == nullptr) {
      Dgrad2Args g = {};
      g.DY = reinterpret_cast<const bf16_t*>(dy);
      for (int i = 0; i < 4; ++i) g.P[i] = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pack) + p.ph[i].pack_off);
      g.DX = reinterpret_cast<bf16_t*>(dx);
      g.N = d->N;
      g.OH = p.OH;
      g.OW = p.OW;
      g.mom = mom;
      return launch_dgrad2_rows(g, s);
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (g_fast_enabled && g_rows_enabled && !refl && st == 2 && d->R == 4 && d->S == 4 && d->pad == 2 && p.Cs == 8 && d->C <= 3 &&
        p.Ks == 64 && d->H % 8 == 0 && d->W % 256 == 0 && p.OH == d->H / 2 + 1 && p.OW == d->W / 2 + 1 && p.nph == 4 &&
        p.ph[0].Lk == 128 && mask == nullptr && addend == nullptr) {
      ThinDgrad2Args g = {};
      g.DY = reinterpret_cast<const bf16_t*>(dy);
      for (int i = 0; i < 4; ++i) g.P[i] = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pack) + p.ph[i].pack_off);
      g.DX = reinterpret_cast<bf16_t*>(dx);
      g.N = d->N;
      g.OH = p.OH;
      g.OW = p.OW;
      g.H = d->H;
      g.W = d->W;
      g.K = d->C;
      return launch_thin_dgrad2_rows(g, s);
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (g_fast_enabled && g_rows_enabled && refl && st == 1 && d->R == 7 && d->S == 7 && d->pad == 3 && p.Ks == 8 && p.Cs == 64 &&
        d->C == 64 && d->H >= 8 && d->W >= 8 && p.nph == 1 && p.ph[0].Lk == 64 && mask == nullptr && addend == nullptr) {
      ThinInArgs g = {};
      g.DY = reinterpret_cast<const bf16_t*>(dy);
      g.P = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pack) + p.ph[0].pack_off);
      g.DX = reinterpret_cast<bf16_t*>(dx);
      g.DXP = reinterpret_cast<bf16_t*>(dxp);
      g.N = d->N;
      g.H = d->H;
      g.W = d->W;
      g.OH = d->H + 6;
      g.OW = d->W + 6;
      g.py = g.px = 6;
      g.act = JPDSE_ACT_NONE;
      return launch_thin_in_rows<7, true>(g, s);
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (taps_dgrad2_ok(d, p, mask, addend, mom)) return launch_taps_dgrad2(d, p, dy, pack, dx, s, mask, addend, mask_slope);
    if (!lrelu && !refl && p.nph == 1 && st == 1 && mom == nullptr && p.ph[0].cnth == d->H &&
        p.ph[0].cntw == d->W && p.ph[0].Lk == p.ph[0].Uw * p.Ks &&
        taps4_shape_ok(p.ph[0].Uh, p.ph[0].Uw, 1, d->H, d->W, p.Ks, p.Cs, (long long)d->N * p.OH * p.OW * p.Ks, (long long)p.Cs * 16 * p.Ks)) {
      const Phase& f = p.ph[0];
      Taps4View v = {};
      v.X = reinterpret_cast<const bf16_t*>(dy);
      v.B = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pack) + f.pack_off);
      v.Y = reinterpret_cast<bf16_t*>(dx);
      v.N = d->N;
      v.IH = p.OH;
      v.IW = p.OW;
      v.Cin_s = p.Ks;
      v.OH = d->H;
      v.OW = d->W;
      v.py = (f.Uh - 1) - f.i0h;
      v.px = (f.Uw - 1) - f.i0w;
      v.Kout = d->C;
      v.Ks_out = p.Cs;
      v.ktot = (long long)f.Uh * f.Lk;
      v.tap_r = f.Lk;
      v.tap_s = p.Ks;
      v.act = JPDSE_ACT_NONE;
      v.addend = reinterpret_cast<const bf16_t*>(addend);
      v.mask = reinterpret_cast<const bf16_t*>(mask);
      return launch_taps4(v, ws, s);
    }
  }
  if constexpr (sizeof(T) == 2) {
    // one output channel (PatchGAN 512 -> 1): dx is written once by an FMA kernel (thin_out1.h)
    if (thin1_shape_ok(d, p.Cs, p.Ks) && p.nph == 1 && mask == nullptr && mom == nullptr && p.ph[0].Uh == 4 && p.ph[0].Uw == 4 &&
        p.ph[0].cnth == d->H && p.ph[0].cntw == d->W) {
      const Phase& f = p.ph[0];
      Thin1DgradArgs t = {};
      t.DY = reinterpret_cast<const bf16_t*>(dy);
      t.B = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pack) + f.pack_off);
      t.taps = reinterpret_cast<float*>(wsb);        // 16 x Cs floats: inside the padded-dy region of every plan
      t.DX = reinterpret_cast<bf16_t*>(dx);
      t.addend = reinterpret_cast<const bf16_t*>(addend);
      t.N = d->N;
      t.H = d->H;
      t.W = d->W;
      t.Cs = p.Cs;
      t.OH = p.OH;
      t.OW = p.OW;
      t.py = (f.Uh - 1) - f.i0h;
      t.px = (f.Uw - 1) - f.i0w;
      t.Lk = f.Lk;
      t.b_stride = (long long)f.Uh * f.Lk;
      return launch_thin1_dgrad(t, s);
    }
  }
  bool fast = false;
  int nlive_phases = 0;
  if constexpr (sizeof(T) == 2) {
    // all stride phases go into ONE launch of the fast kernel: judge the merged grid
    fast = p.Ks % 64 == 0;
    long long tiles = 0;
    int kt_max = 0, nlive = 0, m_single = 0;
    const int bn = p.Cs > 64 ? 128 : (p.Cs > 32 ? 64 : 32);
    for (int i = 0; i < p.nph; ++i) {
      if (p.ph[i].cnth <= 0 || p.ph[i].cntw <= 0) continue;
      const int Mi = d->N * p.ph[i].cnth * p.ph[i].cntw;
      tiles += (long long)((Mi + 255) / 256) * ((p.Cs + bn - 1) / bn);
      const int kt = p.ph[i].Uh * p.ph[i].Uw * p.Ks / 64;
      kt_max = kt > kt_max ? kt : kt_max;
      m_single = Mi;
      ++nlive;
    }
    nlive_phases = nlive;
    if (nlive == 1) fast = fast && fast_pays(m_single, p.Cs, kt_max);
    else fast = fast && g_fast_enabled && p.Cs > 32 && kt_max >= g_merge_min_kt && tiles >= g_merge_min_tiles;   // few tiles / short K loops: generic wins
  }
  FastBatch batch = {};
  int rc = JPDSE_OK;
  if (!fast) {
    rc = launch_pad<T>(dy, dyp, d->N, p.OH, p.OW, p.Ks, p.PT, p.PB, p.PL, p.PR, JPDSE_PAD_ZERO, s);
    if (rc) return rc;
  }
  for (int i = 0; i < p.nph; ++i) {
    const Phase& f = p.ph[i];
    if (f.cnth <= 0 || f.cntw <= 0) continue;
    if constexpr (sizeof(T) == 2) {
      if (fast) {
        FastArgs g = {};
        g.X = reinterpret_cast<const bf16_t*>(dy);
        g.B = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pack) + f.pack_off);
        g.bias = nullptr;
        g.M = d->N * f.cnth * f.cntw;
        g.OH = f.cnth;
        g.OW = f.cntw;
        g.IH = p.OH;
        g.IW = p.OW;
        g.Cs = p.Ks;
        g.R = f.Uh;
        g.S = f.Uw;
        g.sy = g.sx = 1;
        g.py = (f.Uh - 1) - f.i0h;
        g.px = (f.Uw - 1) - f.i0w;
        g.reflect = 0;
        g.Kout = d->C;
        g.Ks = p.Cs;
        g.b_rows = p.Cs;
        if (refl) {
          g.Y = reinterpret_cast<bf16_t*>(dxp);
          g.out_sn = (long long)p.Hp * p.Wp * p.Cs;
          g.out_sh = (long long)st * p.Wp * p.Cs;
          g.out_sw = (long long)st * p.Cs;
          g.out_base = ((long long)(st * f.i0h + f.qh) * p.Wp + (st * f.i0w + f.qw)) * p.Cs;
        } else {
          g.Y = reinterpret_cast<bf16_t*>(dx);
          g.out_sn = (long long)d->H * d->W * p.Cs;
          g.out_sh = (long long)st * d->W * p.Cs;
          g.out_sw = (long long)st * p.Cs;
          g.out_base = ((long long)(st * f.i0h + f.qh - d->pad) * d->W + (st * f.i0w + f.qw - d->pad)) * p.Cs;
        }
        g.act = JPDSE_ACT_NONE;
        g.slope = 0.f;
        g.mask = refl ? nullptr : reinterpret_cast<const bf16_t*>(mask);
        g.mask_slope = mask_slope;
        g.addend = refl ? nullptr : reinterpret_cast<const bf16_t*>(addend);
        g.splits = nlive_phases == 1 ? splitk_for(g.M, p.Cs, f.Uh * f.Uw * p.Ks / 64) : 1;
        g.partial = reinterpret_cast<float*>(wsb + p.splitk_off);
        batch.p[batch.n++] = g;
        continue;
      }
    }
    GemmFwdArgs a = {};
    a.A = dyp;
    a.B = reinterpret_cast<const char*>(pack) + f.pack_off;
    a.bias = nullptr;
    a.M = d->N * f.cnth * f.cntw;
    a.OH = f.cnth;
    a.OW = f.cntw;
    a.Kout = d->C;
    a.Ks = p.Cs;
    a.R = f.Uh;
    a.cpr = f.Lk / p.BKE;
    a.b_rows = p.Cs;
    a.b_row_stride = (long long)f.Uh * f.Lk;
    a.in_sn = (long long)p.DH * p.DW * p.Ks;
    a.in_sh = (long long)p.DW * p.Ks;
    a.in_sw = p.Ks;
    a.in_sr = (long long)p.DW * p.Ks;
    a.in_base = ((long long)(f.i0h + p.PT - (f.Uh - 1)) * p.DW + (f.i0w + p.PL - (f.Uw - 1))) * p.Ks;
    if (refl) {
      a.Y = dxp;
      a.out_sn = (long long)p.Hp * p.Wp * p.Cs;
      a.out_sh = (long long)st * p.Wp * p.Cs;
      a.out_sw = (long long)st * p.Cs;
      a.out_base = ((long long)(st * f.i0h + f.qh) * p.Wp + (st * f.i0w + f.qw)) * p.Cs;
    } else {
      a.Y = dx;
      a.out_sn = (long long)d->H * d->W * p.Cs;
      a.out_sh = (long long)st * d->W * p.Cs;
      a.out_sw = (long long)st * p.Cs;
      a.out_base = ((long long)(st * f.i0h + f.qh - d->pad) * d->W + (st * f.i0w + f.qw - d->pad)) * p.Cs;
    }
    a.act = JPDSE_ACT_NONE;
    a.slope = 0.f;
    a.partial = reinterpret_cast<float*>(wsb + p.splitk_off);     // split-K slabs, reused by the phases (stream order)
    a.partial_cap = p.splitk_bytes;
    rc = launch_fwd<T>(a, s);
    if (rc) return rc;
  }
  if (batch.n > 0) {
    rc = launch_fast_batch(batch, s);
    if (rc) return rc;
  }
  if (refl) {
    const int VE = 16 / (int)sizeof(T);
    const long long total_vec = (long long)d->N * d->H * d->W * (p.Cs / VE);
    hipLaunchKernelGGL((reflect_fold_kernel<T>), dim3(ew_blocks(total_vec)), dim3(256), 0, s,
                       reinterpret_cast<const T*>(dxp), reinterpret_cast<T*>(dx), d->N, d->H, d->W, p.Cs, d->pad,
                       total_vec);
    rc = check_launch("reflect_fold_kernel");
  }
  if (rc == JPDSE_OK && (mask != nullptr || addend != nullptr) && !(fast && !refl)) {
    const int VE = 16 / (int)sizeof(T);
    const long long total_vec = (long long)d->N * d->H * d->W * (p.Cs / VE);
    hipLaunchKernelGGL((relu_mask_kernel<T>), dim3(ew_blocks(total_vec)), dim3(256), 0, s, reinterpret_cast<T*>(dx),
                       reinterpret_cast<const T*>(mask), total_vec, reinterpret_cast<const T*>(addend), mask_slope);
    rc = check_launch("relu_mask_kernel");
  }
  return rc;
}

}  // namespace jpdse
