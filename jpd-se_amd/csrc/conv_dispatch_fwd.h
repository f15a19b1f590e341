// Forward dispatch of the convolution family: picks the kernel of a layer (order matters: the moment-slot query of
// conv_launch.h mirrors it).  Part of conv_gemm.hip (one translation unit).
#pragma once

namespace jpdse {

template <typename T>
static int conv_fwd_t(const jpdse_conv_desc* d, const ConvPlan& p, const void* x, const void* pack,
                      const float* bias, void* y, void* ws, hipStream_t s, float* mom = nullptr, void* pool = nullptr,
                      bool* pooled = nullptr) {
  // pool != nullptr: the caller wants MaxPool2d(2, 2) of y as well; the branch that can write it from its epilogue sets *pooled
  // mom != nullptr: the caller asked jpdse_conv_moment_slots first, so the branch taken below is one that writes them
  if constexpr (sizeof(T) == 2) {
    if (g_fast_enabled && g_rows_enabled && p.Cs == 8 && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 &&
        d->pad_mode == JPDSE_PAD_ZERO && p.Ks == 64 && d->K == 64 && p.Lk_fwd == 32 &&
        (d->act == JPDSE_ACT_NONE || d->act == JPDSE_ACT_RELU || d->act == JPDSE_ACT_LRELU)) {
      ThinInArgs g = {};                       // VGG conv1_1: plain forward panel [k][r][(s, c8) 24 -> 32]
      g.DY = reinterpret_cast<const bf16_t*>(x);
      g.P = reinterpret_cast<const bf16_t*>(pack);
      g.DX = reinterpret_cast<bf16_t*>(y);
      g.bias = bias;
      g.act = d->act;
      g.slope = d->slope;
      g.N = d->N;
      g.H = d->H;
      g.W = d->W;
      g.OH = p.OH;
      g.OW = p.OW;
      g.py = g.px = 1;
      return launch_thin_in_rows<3, false>(g, s);
    }
    ThinFwdGeom tg;
    if (thin_fwd_geom(d, p, &tg)) {
      ThinFwdArgs t = {};
      t.X = reinterpret_cast<const bf16_t*>(x);
      t.Wt = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pack) + p.thin_pack_off);
      t.bias = bias;
      t.Y = reinterpret_cast<bf16_t*>(y);
      t.N = d->N;
      t.H = d->H;
      t.W = d->W;
      t.OH = p.OH;
      t.OW = p.OW;
      t.Cs = p.Cs;
      t.K = d->K;
      t.Ks = p.Ks;
      t.R = d->R;
      t.S = d->S;
      t.pad = d->pad;
      t.reflect = d->pad_mode == JPDSE_PAD_REFLECT;
      t.act = d->act;
      t.slope = d->slope;
      t.KP = p.KP_thin;
      t.ksteps = (p.KP_thin - 8) / 16;
      t.strip_units = tg.strip_units;
      t.w_units = tg.w_units;
      t.tiles_w = (p.OW + tg.TW - 1) / tg.TW;
      t.tiles_h = (p.OH + tg.TH - 1) / tg.TH;
      if (mom != nullptr && !thin_rows_takes(d, p)) {
        t.mom = mom;
        t.mom_slots = t.tiles_w * t.tiles_h;
      }
      if (g_rows_enabled && d->stride == 2 && d->R == 4 && d->S == 4 && p.Cs == 40 && p.Ks == 64 && d->K == 64 && !t.reflect &&
          p.KP_thin == 168 && (d->act == JPDSE_ACT_NONE || d->act == JPDSE_ACT_RELU || d->act == JPDSE_ACT_LRELU))
        return launch_thin_rows(t, s);
      if (d->stride == 1) {
        if (tg.TH == 8) return p.Ks == 64 ? launch_thin_fwd<2, 8, 1, 64>(t, tg.lds, s) : launch_thin_fwd<1, 8, 1, 64>(t, tg.lds, s);
        return tg.TW == 64 ? launch_thin_fwd<1, 4, 1, 64>(t, tg.lds, s) : launch_thin_fwd<1, 4, 1, 32>(t, tg.lds, s);
      }
      if (tg.TH == 8) return p.Ks == 64 ? launch_thin_fwd<2, 8, 2, 64>(t, tg.lds, s) : launch_thin_fwd<1, 8, 2, 64>(t, tg.lds, s);
      return tg.TW == 64 ? launch_thin_fwd<1, 4, 2, 64>(t, tg.lds, s) : launch_thin_fwd<1, 4, 2, 32>(t, tg.lds, s);
    }
    if (head_fwd_ok(d, p)) {
      HeadFwdArgs h = {};
      h.X = reinterpret_cast<const bf16_t*>(x);
      h.Wp = reinterpret_cast<const bf16_t*>(pack);
      h.bias = bias;
      h.Y = reinterpret_cast<bf16_t*>(y);
      h.N = d->N;
      h.H = d->H;
      h.W = d->W;
      h.OH = p.OH;
      h.OW = p.OW;
      h.K = d->K;
      h.Ks_out = p.Ks;
      h.R = d->R;
      h.S = d->S;
      h.pad = d->pad;
      h.reflect = d->pad_mode == JPDSE_PAD_REFLECT;
      h.act = d->act;
      h.slope = d->slope;
      h.tiles_w = (p.OW + 63) / 64;
      h.tiles_h = (p.OH + kHeadTH - 1) / kHeadTH;
      if (head_rows_ok(h, p.Cs)) return p.Cs == 64 ? launch_head_rows<7>(h, s) : launch_head_rows<7, 32>(h, s);
      return p.Cs == 64 ? launch_head_fwd<64>(h, s) : launch_head_fwd<32>(h, s);
    }
    if (tapsum_ok(d, p)) {
      const int cols = d->K * d->R * d->S, zs = (cols + 7) / 8 * 8;
      FastArgs f = {};
      f.X = reinterpret_cast<const bf16_t*>(x);
      f.B = reinterpret_cast<const bf16_t*>(pack);     // row k of the plain panel = [R*S][Cs]: K*R*S rows of Cs
      f.M = d->N * d->H * d->W;
      f.OH = d->H;
      f.OW = d->W;
      f.IH = d->H;
      f.IW = d->W;
      f.Cs = p.Cs;
      f.R = f.S = 1;
      f.sy = f.sx = 1;
      f.Kout = cols;
      f.Ks = zs;
      f.b_rows = cols;
      f.act = JPDSE_ACT_NONE;
      f.splits = 1;
      f.no_finish = 1;
      f.partial = reinterpret_cast<float*>(ws);
      if (int rc = launch_fast(f, s)) return rc;
      const long long total = (long long)d->N * p.OH * p.OW * p.Ks;
      hipLaunchKernelGGL(tapsum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, f.partial, bias,
                         reinterpret_cast<bf16_t*>(y), d->N, d->H, d->W, p.OH, p.OW, d->K, p.Ks, d->R, d->S, d->pad,
                         d->pad_mode == JPDSE_PAD_REFLECT ? 1 : 0, zs, d->act, d->slope, total);
      return check_launch("tapsum_kernel");
    }
    if (rows_ok(d->R, d->S, d->stride, d->pad_mode == JPDSE_PAD_REFLECT, d->act, p.OH, p.OW, p.Cs, p.Ks)) {
      RowsArgs r = {};
      r.X = reinterpret_cast<const bf16_t*>(x);
      r.B = reinterpret_cast<const bf16_t*>(pack);
      r.bias = bias;
      r.Y = reinterpret_cast<bf16_t*>(y);
      r.N = d->N;
      r.OH = p.OH;
      r.OW = p.OW;
      r.IH = d->H;
      r.IW = d->W;
      r.py = r.px = d->pad;
      r.Kout = d->K;
      r.Ks = p.Ks;
      r.b_rows = p.Ks;
      r.out_sn = (long long)p.OH * p.OW * p.Ks;
      r.out_sh = (long long)p.OW * p.Ks;
      r.out_sw = p.Ks;
      r.out_base = 0;
      r.act = d->act;
      r.slope = d->slope;
      r.mom = mom;
      return launch_rows(r, d->stride, s);
    }
    if (d->pad_mode != JPDSE_PAD_REFLECT && p.Lk_fwd == d->S * p.Cs && mom == nullptr &&
        taps4_shape_ok(d->R, d->S, d->stride, p.OH, p.OW, p.Cs, p.Ks, (long long)d->N * d->H * d->W * p.Cs, (long long)p.Ks * 16 * p.Cs)) {
      Taps4View v = {};
      v.X = reinterpret_cast<const bf16_t*>(x);
      v.B = reinterpret_cast<const bf16_t*>(pack);
      v.bias = bias;
      v.Y = reinterpret_cast<bf16_t*>(y);
      v.N = d->N;
      v.IH = d->H;
      v.IW = d->W;
      v.Cin_s = p.Cs;
      v.OH = p.OH;
      v.OW = p.OW;
      v.py = v.px = d->pad;
      v.Kout = d->K;
      v.Ks_out = p.Ks;
      v.ktot = (long long)d->R * p.Lk_fwd;
      v.tap_r = p.Lk_fwd;
      v.tap_s = p.Cs;
      v.act = d->act;
      v.slope = d->slope;
      return launch_taps4(v, ws, s);
    }
    if (p.Lk_fwd == d->S * p.Cs && mom == nullptr &&
        taps9_shape_ok(d->R, d->S, d->stride, p.OH, p.OW, p.Cs, p.Ks, (long long)d->N * d->H * d->W * p.Cs, (long long)p.Ks * 9 * p.Cs)) {
      Taps4View v = {};
      v.X = reinterpret_cast<const bf16_t*>(x);
      v.B = reinterpret_cast<const bf16_t*>(pack);
      v.bias = bias;
      v.Y = reinterpret_cast<bf16_t*>(y);
      v.N = d->N;
      v.IH = d->H;
      v.IW = d->W;
      v.Cin_s = p.Cs;
      v.OH = p.OH;
      v.OW = p.OW;
      v.py = v.px = d->pad;
      v.reflect = d->pad_mode == JPDSE_PAD_REFLECT;
      v.Kout = d->K;
      v.Ks_out = p.Ks;
      v.ktot = (long long)d->R * p.Lk_fwd;
      v.tap_r = p.Lk_fwd;
      v.tap_s = p.Cs;
      v.act = d->act;
      v.slope = d->slope;
      return launch_taps9(v, reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + p.splitk_off), s);
    }
    if (halo_ok(d->R, d->S, d->stride, p.OH, p.OW, p.Cs, p.Ks)) {
      HaloArgs h = {};
      h.X = reinterpret_cast<const bf16_t*>(x);
      h.B = reinterpret_cast<const bf16_t*>(pack);
      h.bias = bias;
      h.Y = reinterpret_cast<bf16_t*>(y);
      h.N = d->N;
      h.OH = p.OH;
      h.OW = p.OW;
      h.IH = d->H;
      h.IW = d->W;
      h.Cs = p.Cs;
      h.py = h.px = d->pad;
      h.reflect = d->pad_mode == JPDSE_PAD_REFLECT;
      h.Kout = d->K;
      h.Ks = p.Ks;
      h.b_rows = p.Ks;
      h.out_sn = (long long)p.OH * p.OW * p.Ks;
      h.out_sh = (long long)p.OW * p.Ks;
      h.out_sw = p.Ks;
      h.out_base = 0;
      h.act = d->act;
      h.slope = d->slope;
      if (mom != nullptr) {
        h.mom = mom;
        h.mom_slots = (p.OH / 4) * (p.OW / 64);
      }
      if (pool != nullptr && mom == nullptr && pooled != nullptr && !g_halo_abl) {
        h.pool = reinterpret_cast<bf16_t*>(pool);
        *pooled = true;
      }
#ifdef JPDSE_DEV
      if (g_halo_abl && p.Ks > 64) {      // timing-only ablations (scripts/bench_conv.py --fast 11..)
        switch (g_halo_abl) {
          case 1: return launch_halo_cfg<2, 1>(h, s);
          case 2: return launch_halo_cfg<2, 2>(h, s);
          case 4: return launch_halo_cfg<2, 4>(h, s);
          case 9: return launch_halo_cfg<2, 9>(h, s);
          case 11: return launch_halo_cfg<2, 11>(h, s);
          case 15: return launch_halo_cfg<2, 15>(h, s);
          case 16: return launch_halo_cfg<2, 16>(h, s);
          case 32: return launch_halo_cfg<2, 32>(h, s);
          case 48: return launch_halo_cfg<2, 48>(h, s);
          default: break;
        }
      }
#endif
      return p.Ks > 64 ? launch_halo_cfg<2>(h, s) : launch_halo_cfg<1>(h, s);
    }
    if (p.Cs % 64 == 0 && fast_pays(d->N * p.OH * p.OW, p.Ks, d->R * d->S * p.Cs / 64)) {
      FastArgs f = {};
      f.X = reinterpret_cast<const bf16_t*>(x);
      f.B = reinterpret_cast<const bf16_t*>(pack);
      f.bias = bias;
      f.Y = reinterpret_cast<bf16_t*>(y);
      f.M = d->N * p.OH * p.OW;
      f.OH = p.OH;
      f.OW = p.OW;
      f.IH = d->H;
      f.IW = d->W;
      f.Cs = p.Cs;
      f.R = d->R;
      f.S = d->S;
      f.sy = f.sx = d->stride;
      f.py = f.px = d->pad;
      f.reflect = d->pad_mode == JPDSE_PAD_REFLECT;
      f.Kout = d->K;
      f.Ks = p.Ks;
      f.b_rows = p.Ks;
      f.out_sn = (long long)p.OH * p.OW * p.Ks;
      f.out_sh = (long long)p.OW * p.Ks;
      f.out_sw = p.Ks;
      f.out_base = 0;
      f.act = d->act;
      f.slope = d->slope;
      f.splits = splitk_for(f.M, p.Ks, d->R * d->S * p.Cs / 64);
      f.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + p.splitk_off);
      return launch_fast(f, s);
    }
  }
  // generic path: staged through the workspace, the GEMM loaders rely on the zeroed slack behind it
  if (int rc = launch_pad<T>(x, ws, d->N, d->H, d->W, p.Cs, d->pad, d->pad, d->pad, d->pad, d->pad_mode, s))
    return rc;
  const void* xin = ws;
  if (p.toep && g_fast_enabled && g_toep_enabled && p.OW % 4 == 0) {
    // head: 4 output pixels x 8 channels per 32-wide GEMM row (see pack_fwd_toep_kernel); the [M/4][32]
    // result IS the NHWC output
    GemmFwdArgs a = {};
    a.A = xin;
    a.B = reinterpret_cast<const char*>(pack) + p.fwd_pack_plain_bytes;
    a.bias = bias;
    a.Y = y;
    a.M = d->N * p.OH * (p.OW / 4);
    a.OH = p.OH;
    a.OW = p.OW / 4;
    a.Kout = 32;
    a.Ks = 32;
    a.R = d->R;
    a.cpr = p.Lk_toep / p.BKE;
    a.b_rows = 32;
    a.b_row_stride = (long long)d->R * p.Lk_toep;
    a.in_sn = (long long)p.Hp * p.Wp * p.Cs;
    a.in_sh = (long long)p.Wp * p.Cs;
    a.in_sw = 4LL * p.Cs;
    a.in_sr = (long long)p.Wp * p.Cs;
    a.in_base = 0;
    a.out_sn = (long long)p.OH * p.OW * p.Ks;
    a.out_sh = (long long)p.OW * p.Ks;
    a.out_sw = 4LL * p.Ks;
    a.out_base = 0;
    a.act = d->act;
    a.slope = d->slope;
    a.col_mod = 8;
    a.k_real = d->K;
    return launch_fwd<T>(a, s);
  }
  GemmFwdArgs a = {};
  a.A = xin;
  a.B = pack;
  a.bias = bias;
  a.Y = y;
  a.M = d->N * p.OH * p.OW;
  a.OH = p.OH;
  a.OW = p.OW;
  a.Kout = d->K;
  a.Ks = p.Ks;
  a.R = d->R;
  a.cpr = p.Lk_fwd / p.BKE;
  a.b_rows = p.Ks;
  a.b_row_stride = (long long)d->R * p.Lk_fwd;
  a.in_sn = (long long)p.Hp * p.Wp * p.Cs;
  a.in_sh = (long long)d->stride * p.Wp * p.Cs;
  a.in_sw = (long long)d->stride * p.Cs;
  a.in_sr = (long long)p.Wp * p.Cs;
  a.in_base = 0;
  a.out_sn = (long long)p.OH * p.OW * p.Ks;
  a.out_sh = (long long)p.OW * p.Ks;
  a.out_sw = p.Ks;
  a.out_base = 0;
  a.act = d->act;
  a.slope = d->slope;
  a.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + p.splitk_off);   // split-K slabs (fp32 layers with few tiles)
  a.partial_cap = p.splitk_bytes;
  return launch_fwd<T>(a, s);
}

}  // namespace jpdse
