// bf16 weight gradient of the 3x3 stride-1 pad-1 convs with many channels (the 18 ResnetBlock convs of the generator:
// 1024 -> 1024, ctu/models/pix2pixHD_networks/networks.py:275-298):
//     dW[k][r][s][c] = sum_p dy[p][k] * x[p + (r-1, s-1)][c]
// without atomics and without partial tiles.
//
// wgrad_row_kernel (round 1) owned 256 k x 128 c x 3 taps per block: 96 tiles for a 1024 x 1024 filter, so the pixel
// range had to be cut stream-K fashion over the 256 CUs and the partial tiles met in fp32 atomics -- 20-25 % of the
// kernel's time (~100 MB of atomic adds at the chip-wide atomic rate) and run-to-run nondeterministic.  Here a block
// owns a 64 k x 64 c tile of ALL NINE taps: 16 x 16 = 256 tiles for 1024 x 1024, exactly one per CU, each reducing
// over every pixel and storing its 36,864 results once (37.7 MB per launch, no memset, no finish pass).
//
// Fill per chunk (64 consecutive pixels of one output row): the dy tile (64 px x 64 k = 8 KiB) and ONE new input row
// (66 px x 64 c, 8.3 KiB): the three input rows h-1, h, h+1 live in a ring of row slots that rolls down the image --
// consecutive output rows share two of their three input rows, so every input pixel is staged once instead of three
// times (9 taps, 16.3 KiB per chunk: 288 FLOP per staged byte against 253 for the per-row kernel).  Reflect / zero
// padding: columns are resolved per pixel by the loader, rows by choosing the slot (reflect: row -1 is row 1; zero:
// a slot of zeros).  Tap (r, s) = a transposed read (ds_read_b64_tr_b16) of slot r at pixel offset s.
//
// 8 waves = 2 (k) x 2 (c) x 2 (pixel halves of the chunk); wave tile 32 k x 32 c x 9 taps = 9 accumulator tiles; the
// two pixel halves are summed through LDS in a fixed order at the end (deterministic).  3-stage dy ring / 5 row
// slots, counted vmcnt, one raw barrier per chunk (as wgrad_row.h).  When the filter has fewer tiles than the chip
// has CUs, the pixel range is cut into `splits` equal parts, each block stores its partial tile into its own fp32
// slab and wgrad_nine_reduce_kernel adds the slabs in a fixed order.
//
// W32 instantiation (round 4): images 32 pixels wide -- the 1024-channel trunk of the LocalEnhancer at 16 x 32 (BASELINE config
// 3), where the per-tap kernel ran 144 blocks of 256 x 256 on 256 CUs (492 TFLOP/s).  A chunk is a PAIR of image rows (2j, 2j+1):
// 64 consecutive pixels of dy, so the dy side is unchanged; the wave's pixel half (ph) IS the row of the pair, so tap (r, s) of a
// wave reads ONE real input row, 2j + ph + r - 1, at pixel offset s: a wave-uniform slot choice as before.  Input rows are staged
// as 34-pixel rows (40-pixel slots), two new ones per chunk (global rows 2t+1, 2t+2 for chunk t: the stream runs seamlessly
// across images because H is even), eight slots (row G lives in slot G mod 8).  Reflect padding only: row -1 is row 1 (first
// pair of an image, ph 0), row H is row H-2 (last pair, ph 1); columns by the loader.
#pragma once
#include "common.h"
#include "gemm_fast.h"
#include "wgrad_fast.h"

namespace jpdse {

struct NineWgArgs {
  const bf16_t* X;    // [N][H][W][Cs] unpadded
  const bf16_t* DY;   // [N][H][W][Ks]
  float* DW;          // fp32 KRSC [K][3][3][C]
  float* partial;     // [splits][K][3][3][C] when splits > 1
  int N, H, W, Cs, C, Ks, K;
  int strips;         // W / 64 column strips; chunk order: (n, strip, h)
  int chunks_total;   // N * strips * H
  int k_tiles, c_tiles, splits, chunks_per_split;
  int xcd_map;        // tiles dealt to the 8 XCDs in 4 x 8 groups (speed only)
};

static constexpr int kNineStage = 64 * 128;           // dy stage: 64 px x 128 B
static constexpr int kNineSlot = 72 * 128;            // input row slot: 72 px x 128 B (66 live)
static constexpr int kNineSlots = 5;
static constexpr int kNineLoopLds = 3 * kNineStage + (kNineSlots + 1) * kNineSlot;   // + one slot of zeros
static constexpr int kNineRedLds = 4 * 9 * 16 * 64 * 4;                              // pixel-half reduction
static constexpr int kNineLds = kNineLoopLds > kNineRedLds ? kNineLoopLds : kNineRedLds;
static constexpr int kNine32Slot = 40 * 128;          // W32: input row slot, 40 px x 128 B (34 live: columns -1 .. 32)
static constexpr int kNine32Slots = 8;
static_assert(3 * kNineStage + kNine32Slots * kNine32Slot <= kNineLds, "the W32 loop fits the same allocation");

// SCHED: where the DMA issue of chunk t+2 sits in iteration t (see the loop).  ABL (timing-only): 1 = no epilogue
template <bool REFLECT, int SCHED = 3, int ABL = 0, bool W32 = false>
__global__ __launch_bounds__(512) void wgrad_nine_kernel(const NineWgArgs a) {
  static_assert(!W32 || REFLECT, "the 32-pixel-wide form is built for reflect padding");
  constexpr int NW = 8, BKP = 64, ROWB = 128;
  constexpr int A_STAGE = kNineStage, B_SLOT = W32 ? kNine32Slot : kNineSlot, NSLOT = W32 ? kNine32Slots : kNineSlots;
  constexpr int B_BASE = 3 * A_STAGE, ZERO_OFF = B_BASE + NSLOT * B_SLOT;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wid & 1, wc = (wid >> 1) & 1, ph = wid >> 2;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  const uint32_t lds0 = lds_addr_of(smem);

  // ---- block -> (tile, split) ---------------------------------------------------------------------
  const int tiles = a.k_tiles * a.c_tiles;
  int tile = blockIdx.x % tiles;
  const int sp = blockIdx.x / tiles;
  int kt, ct;
  if (a.xcd_map) {
    // blocks b, b + 8, ... share an XCD (observed dispatch; speed only): give each XCD a 4 (k) x 8 (c) group of tiles,
    // so that its L2 serves 4 dy panels + 8 x panels instead of 2 + 16
    const int xcd = tile & 7, idx = tile >> 3;
    const int gk = a.k_tiles >> 2;               // groups along k
    const int grp = xcd + 8 * (idx >> 5);        // 32 tiles per group
    const int in = idx & 31;
    kt = (grp % gk) * 4 + (in & 3);
    ct = (grp / gk) * 8 + (in >> 2);
  } else {
    ct = tile % a.c_tiles;
    kt = tile / a.c_tiles;
  }
  const int k0 = kt * 64, c0 = ct * 64;
  const int t0 = sp * a.chunks_per_split;
  int t1 = t0 + a.chunks_per_split;
  t1 = t1 < a.chunks_total ? t1 : a.chunks_total;

  // ---- zero slot (zero padding rows) ----------------------------------------------------------------
  if constexpr (!REFLECT) {
    for (int i = tid; i < B_SLOT / 16; i += 512) *reinterpret_cast<u32x4*>(smem + ZERO_OFF + i * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  // ---- transposed fragment offsets (lane roles as in wgrad_fast.h) ----------------------------------
  uint32_t a_rd, b_rd[3];
  {
    const int g = lane >> 4, li = lane & 15, h2 = g >> 1, cb = g & 1, q = li >> 2, p = li & 3;
    const int pix = 32 * ph + 8 * h2 + q;                    // this wave's pixel half; + 16 * ks + 4 * half: swizzle unchanged
    const int ch = wk * 32 + cb * 16 + 4 * p;
    a_rd = lds0 + pix * ROWB + ((((ch >> 3) ^ trswz<ROWB>(pix)) << 4) | ((ch & 7) << 1));
    const int chb = wc * 32 + cb * 16 + 4 * p;
    const int pixb = W32 ? pix - 32 * ph : pix;            // W32: pixel inside the wave's own input row
#pragma unroll
    for (int s = 0; s < 3; ++s)
      b_rd[s] = lds0 + B_BASE + (pixb + s) * ROWB + ((((chb >> 3) ^ trswz<ROWB>(pixb + s)) << 4) | ((chb & 7) << 1));
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // ---- loader state --------------------------------------------------------------------------------
  // dy: unit `wid` = pixels 8*wid .. 8*wid+7 of the chunk; x row: unit `wid` = row pixels 8*wid .. +7 (pixel q of
  // the row = image column 64*strip - 1 + q), wave 0 also unit 8 (pixels 64..71, 64 and 65 live)
  const int pl = lane >> 3, slot16 = lane & 7;
  const int a_pix = wid * 8 + pl;
  const int a_loff = a_pix * a.Ks + k0 + ((slot16 ^ trswz<ROWB>(a_pix)) << 3);
  const int b_q = wid * 8 + pl;
  const int b_sw = c0 + ((slot16 ^ trswz<ROWB>(b_q)) << 3);       // swizzle depends on (q >> 1) & 1: the same for q + 64
  const int Wm1 = a.W - 1, Hm1 = a.H - 1;
  // Cursors of the next dy chunk / next x row to issue: a 64-bit base pointer each (SGPR pair) advanced by precomputed
  // deltas through selects -- no multiplication, division or branch per chunk (the first version spent ~100 scalar
  // instructions and 8 branches per chunk behind the barrier, where they idle the matrix pipes of the whole CU).
  // The row stream is indexed like the chunks: row g <-> (n, strip, h) of chunk g.
  const long long d_row = (long long)a.W * a.Ks, d_strip = 64LL * a.Ks - (long long)a.H * a.W * a.Ks,
                  d_img = -(long long)(a.strips - 1) * 64 * a.Ks;
  const long long r_row = (long long)a.W * a.Cs, r_img_rows = (long long)a.H * a.W * a.Cs;
  const bf16_t* d_ptr;
  const bf16_t* r_ptr;      // first pixel of the input ROW (column 0), the strip's column offset is r_st * 64 - 1
  int d_h, d_st, r_h, r_st, r_left;     // r_left: rows of the stream that still exist (past the end: stay on the last)
  {
    const int ns0 = t0 / a.H, n0 = ns0 / a.strips, st0 = ns0 - n0 * a.strips, h0 = t0 - ns0 * a.H;
    d_st = __builtin_amdgcn_readfirstlane(st0);
    d_h = __builtin_amdgcn_readfirstlane(h0);
    d_ptr = a.DY + ((long long)(n0 * a.H + h0) * a.W + st0 * BKP) * a.Ks;
    const int g = t0 > 0 ? t0 - 1 : 0;        // the stream starts at row t0 - 1 (t0 = 0: that row is skipped below)
    const int ns1 = g / a.H, n1 = ns1 / a.strips, st1 = ns1 - n1 * a.strips, h1 = g - ns1 * a.H;
    r_st = __builtin_amdgcn_readfirstlane(st1);
    r_h = __builtin_amdgcn_readfirstlane(h1);
    r_ptr = a.X + (long long)(n1 * a.H + h1) * a.W * a.Cs;
    r_left = __builtin_amdgcn_readfirstlane(a.chunks_total - 1 - g);
  }
  int r_slot = __builtin_amdgcn_readfirstlane((t0 + NSLOT - 1) % NSLOT);     // slot of row g = g mod 5, g = t0 - 1
  int d_stage = 0;

  auto issue_dy = [&]() {
    glds16(d_ptr + (unsigned)a_loff, smem + d_stage * A_STAGE + wid * 1024);
    d_stage = d_stage == 2 ? 0 : d_stage + 1;
    // (n, strip, h) -> next chunk; the caller never issues past chunk t1 - 1 <= the last chunk
    const bool wrap_h = d_h == Hm1;
    const bool wrap_s = d_st == a.strips - 1;
    d_ptr += d_row + (wrap_h ? (wrap_s ? d_img : d_strip) : 0LL);
    d_h = wrap_h ? 0 : d_h + 1;
    d_st = wrap_h ? (wrap_s ? 0 : d_st + 1) : d_st;
  };
  auto issue_row = [&]() {
    const bf16_t* const row = r_ptr;
    char* const dst = smem + B_BASE + r_slot * B_SLOT;
    const int col0 = r_st * BKP - 1;
    auto unit = [&](int q, int sw, int u) {
      int iw = col0 + q;
      if constexpr (REFLECT) {
        iw = iw < 0 ? -iw : iw;
        iw = iw > Wm1 ? 2 * Wm1 - iw : iw;
        iw = iw < 0 ? 0 : iw;                                // pixels 66..71 of the last strip: never consumed
        glds16(row + (unsigned)(__mul24(iw, a.Cs) + sw), dst + u * 1024);
      } else {
        const bool ok = (unsigned)iw < (unsigned)a.W;
        const bf16_t* const src = row + (unsigned)(__mul24(ok ? iw : 0, a.Cs) + sw);
        glds16(ok ? src : zero, dst + u * 1024);
      }
    };
    unit(b_q, b_sw, wid);
    if (wid == 0) {
      int l3 = lane >> 3;                       // rebuilt from the lane id in place (a hoisted copy is spilled and its
      asm volatile("" : "+v"(l3));              // reload drains this wave's DMAs: see wgrad_thin.h / round-1 notes)
      const int q = 64 + l3;
      unit(q, c0 + (((lane & 7) ^ trswz<ROWB>(q)) << 3), 8);
    }
    r_slot = r_slot == NSLOT - 1 ? 0 : r_slot + 1;
    // next row of the stream; the rows after the tensor's last one (issued, never consumed) stay on the last row
    const bool more = r_left > 0;
    const bool wrap_h = r_h == Hm1;
    const bool wrap_s = r_st == a.strips - 1;
    // same strip: next image row; strip done: row 0 of the next strip of this image; image done: row 0 of the next image
    const long long step = wrap_h ? (wrap_s ? r_row : r_row - r_img_rows) : r_row;
    r_ptr += more ? step : 0LL;
    r_h = more ? (wrap_h ? 0 : r_h + 1) : r_h;
    r_st = more ? (wrap_h ? (wrap_s ? 0 : r_st + 1) : r_st) : r_st;
    r_left -= more ? 1 : 0;
  };

  // ---- W32: chunk t = row pair (2j, 2j+1) of an image; its dy is the t-th run of 64 pixels of the tensor; the row stream brings
  // global rows (2g+1, 2g+2) as "pair g" (pair -1 = rows (-1, 0): row -1 is never read -- reflect -- and loads row 0 twice).
  // Units of a pair: 2 rows x 5 units of 8 pixels; wave w takes unit w, waves 0 and 1 also units 8 and 9.
  long long p_row = 0;       // W32: global row index of the next pair's FIRST row (2g + 1)
  const long long rows_all = (long long)a.N * a.H;
  auto issue_pair = [&]() {
    auto unit = [&](int v) {
      const int rsel = v >= 5 ? 1 : 0, u = v - 5 * rsel;
      long long gr = p_row + rsel;
      gr = gr < 0 ? 0 : (gr > rows_all - 1 ? rows_all - 1 : gr);          // rows outside the tensor are never consumed
      const int q = u * 8 + pl;
      int iw = q - 1;
      iw = iw < 0 ? -iw : iw;
      iw = iw > Wm1 ? 2 * Wm1 - iw : iw;
      iw = iw < 0 ? 0 : iw;                                                // pixels 34..39: never consumed
      const int sw = c0 + ((slot16 ^ trswz<ROWB>(q)) << 3);
      const int slot = (int)((p_row + rsel) & (NSLOT - 1));
      glds16(a.X + gr * ((long long)a.W * a.Cs) + (unsigned)(__mul24(iw, a.Cs) + sw), smem + B_BASE + slot * B_SLOT + u * 1024);
    };
    unit(wid);
    if (wid < 2) {
      int w2 = wid;
      asm volatile("" : "+s"(w2));
      unit(8 + w2);
    }
    p_row += 2;
  };
  auto issue_dy32 = [&]() {
    glds16(d_ptr + (unsigned)a_loff, smem + d_stage * A_STAGE + wid * 1024);
    d_stage = d_stage == 2 ? 0 : d_stage + 1;
    d_ptr += 64LL * a.Ks;
  };

  // ---- pipeline: group G_t = {dy(t), row(t+1)}; prologue G_t0 also carries rows t0-1 and t0 ------------
  __syncthreads();        // zero slot written (no DMA in flight yet)
  int c_stage = 0;
  int c_slot = __builtin_amdgcn_readfirstlane(t0 % NSLOT);       // slot of row t
  int c_h = __builtin_amdgcn_readfirstlane(W32 ? t0 % (a.H >> 1) : t0 % a.H);
  if constexpr (W32) {
    d_ptr = a.DY + (long long)t0 * 64 * a.Ks;
    p_row = 2LL * t0 - 1;
    issue_pair();           // rows 2 t0 - 1, 2 t0
    issue_pair();           // rows 2 t0 + 1, 2 t0 + 2
    issue_dy32();           // dy t0
    if (t0 + 1 < t1) {
      issue_dy32();         // G_{t0+1}: dy t0 + 1, rows 2 t0 + 3, 2 t0 + 4
      issue_pair();
    }
  } else {
  if (t0 > 0) issue_row();                                 // row t0 - 1
  else r_slot = r_slot == NSLOT - 1 ? 0 : r_slot + 1;      // there is no row -1: chunk 0 is a first image row
  issue_row();            // row t0
  issue_dy();             // dy t0
  issue_row();            // row t0 + 1
  if (t0 + 1 < t1) {
    issue_dy();           // G_{t0+1}
    issue_row();
  }
  }

  auto kstep = [&]<int KS>(uint32_t a_addr, const uint32_t (&boff)[3]) {
    s16x8 af[1], bf[9];
    af[0] = tr_frag_asm<KS * 16 * ROWB, KS * 16 * ROWB + 4 * ROWB>(a_addr);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s)
        bf[r * 3 + s] = tr_frag_asm<KS * 16 * ROWB, KS * 16 * ROWB + 4 * ROWB>(b_rd[s] + boff[r]);
    tr_wait(af);
    tr_wait(bf);
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
      acc[t9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[t9], acc[t9], 0, 0, 0);
  };

  // SCHED 3: the six (k-step, filter-row) groups of a chunk as a software pipeline -- the transposed reads of group g+1
  // are in flight while the three MFMAs of group g issue, with COUNTED lgkmcnt waits (LDS operations return in order; the
  // loop holds no scalar-memory loads, which would share the counter -- checked in the .s).  The plain form waits for
  // lgkmcnt(0) after each k-step's 20 reads: two exposed LDS round trips per chunk and wave.
  auto lgkm_wait = [&]<int N>(s16x8& f0, s16x8& f1, s16x8& f2, s16x8& f3) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "n"(N));
  };
  auto chunk_pipelined = [&](uint32_t a_addr, const uint32_t (&boff)[3], auto&& mid) {
    s16x8 a0, a1, b0[3], b1[3], b2[3];
    constexpr int K1 = 16 * ROWB;
    a0 = tr_frag_asm<0, 4 * ROWB>(a_addr);
#pragma unroll
    for (int s = 0; s < 3; ++s) b0[s] = tr_frag_asm<0, 4 * ROWB>(b_rd[s] + boff[0]);
#pragma unroll
    for (int s = 0; s < 3; ++s) b1[s] = tr_frag_asm<0, 4 * ROWB>(b_rd[s] + boff[1]);
    mid();                                                     // (half 1: DMA issue under the first reads' latency)
    lgkm_wait.template operator()<6>(a0, b0[0], b0[1], b0[2]);
#pragma unroll
    for (int s = 0; s < 3; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0[s], acc[s], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 3; ++s) b2[s] = tr_frag_asm<0, 4 * ROWB>(b_rd[s] + boff[2]);
    lgkm_wait.template operator()<6>(a0, b1[0], b1[1], b1[2]);
#pragma unroll
    for (int s = 0; s < 3; ++s) acc[3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1[s], acc[3 + s], 0, 0, 0);
    a1 = tr_frag_asm<K1, K1 + 4 * ROWB>(a_addr);
#pragma unroll
    for (int s = 0; s < 3; ++s) b0[s] = tr_frag_asm<K1, K1 + 4 * ROWB>(b_rd[s] + boff[0]);
    lgkm_wait.template operator()<8>(a0, b2[0], b2[1], b2[2]);
#pragma unroll
    for (int s = 0; s < 3; ++s) acc[6 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2[s], acc[6 + s], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 3; ++s) b1[s] = tr_frag_asm<K1, K1 + 4 * ROWB>(b_rd[s] + boff[1]);
    lgkm_wait.template operator()<6>(a1, b0[0], b0[1], b0[2]);
#pragma unroll
    for (int s = 0; s < 3; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0[s], acc[s], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 3; ++s) b2[s] = tr_frag_asm<K1, K1 + 4 * ROWB>(b_rd[s] + boff[2]);
    lgkm_wait.template operator()<6>(a1, b1[0], b1[1], b1[2]);
#pragma unroll
    for (int s = 0; s < 3; ++s) acc[3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1[s], acc[3 + s], 0, 0, 0);
    lgkm_wait.template operator()<0>(a1, b2[0], b2[1], b2[2]);
#pragma unroll
    for (int s = 0; s < 3; ++s) acc[6 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2[s], acc[6 + s], 0, 0, 0);
  };

  for (int t = t0; t < t1; ++t) {
    if (t + 1 < t1) wait_vmcnt<2>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    const bool more = t + 2 < t1;
    // row slots of the three filter rows
    int off0, off1, off2;
    if constexpr (W32) {
      // this wave's output row is 2t + ph (global row index; slot = index mod 8): filter rows read 2t + ph - 1 .. 2t + ph + 1
      const int g0 = 2 * t + ph;
      const bool first = c_h == 0 && ph == 0, last = c_h == (a.H >> 1) - 1 && ph == 1;
      off0 = (((first ? g0 + 1 : g0 - 1)) & (NSLOT - 1)) * B_SLOT;          // row -1 is row 1
      off1 = (g0 & (NSLOT - 1)) * B_SLOT;
      off2 = (((last ? g0 - 1 : g0 + 1)) & (NSLOT - 1)) * B_SLOT;           // row H is row H - 2
    } else {
    const int s_prev = c_slot == 0 ? NSLOT - 1 : c_slot - 1, s_next = c_slot == NSLOT - 1 ? 0 : c_slot + 1;
    off0 = s_prev * B_SLOT;
    off1 = c_slot * B_SLOT;
    off2 = s_next * B_SLOT;
    if constexpr (REFLECT) {
      off0 = c_h == 0 ? off2 : off0;                       // row -1 is row 1
      off2 = c_h == Hm1 ? s_prev * B_SLOT : off2;          // row H is row H - 2
    } else {
      off0 = c_h == 0 ? NSLOT * B_SLOT : off0;             // the slot of zeros
      off2 = c_h == Hm1 ? NSLOT * B_SLOT : off2;
    }
    }
    const uint32_t boff[3] = {(uint32_t)off0, (uint32_t)off1, (uint32_t)off2};
    const uint32_t a_addr = a_rd + c_stage * A_STAGE;
    // The six (k-step, filter-row) groups of a chunk run as a software pipeline (chunk_pipelined); the DMA issue of chunk t+2
    // sits inside it for the second pixel half and behind it for the first.  (The unpipelined loop forms -- DMA issue right
    // behind the barrier, between the two k-steps, alternating between the two waves of a SIMD: round-2 developer modes
    // 21 / 22 / 24 -- measured 7-9 % slower and were removed in round 4; A/B record: DESIGN.md 4.1 (vii).)
    static_assert(SCHED == 3, "only the software-pipelined loop form is built");
    __builtin_amdgcn_s_setprio(1);
    if constexpr (W32) chunk_pipelined(a_addr, boff, [&]() { if (ph == 1 && more) { issue_dy32(); issue_pair(); } });
    else chunk_pipelined(a_addr, boff, [&]() { if (ph == 1 && more) { issue_dy(); issue_row(); } });
    __builtin_amdgcn_s_setprio(0);
    if constexpr (W32) { if (ph == 0 && more) { issue_dy32(); issue_pair(); } }
    else { if (ph == 0 && more) { issue_dy(); issue_row(); } }
    c_stage = c_stage == 2 ? 0 : c_stage + 1;
    c_slot = c_slot == NSLOT - 1 ? 0 : c_slot + 1;
    c_h = W32 ? (c_h == (a.H >> 1) - 1 ? 0 : c_h + 1) : (c_h == Hm1 ? 0 : c_h + 1);
  }

  // ---- sum the two pixel halves through LDS (fixed order), store -----------------------------------------------
  __syncthreads();                                    // every wave is done with the stages
  float* const red = reinterpret_cast<float*>(smem) + (wid & 3) * (9 * 16 * 64);
  if (ph == 1) {
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const f32x4 v = {acc[t9][4 * e4], acc[t9][4 * e4 + 1], acc[t9][4 * e4 + 2], acc[t9][4 * e4 + 3]};
        *reinterpret_cast<f32x4*>(red + ((t9 * 4 + e4) * 64 + lane) * 4) = v;
      }
  }
  __syncthreads();
  if (ph == 0 && !(ABL & 1)) {
    float* const out = a.splits > 1 ? a.partial + (long long)sp * a.K * 9 * a.C : a.DW;
    const int cc = c0 + wc * 32 + (lane & 31);
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((t9 * 4 + e4) * 64 + lane) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = 4 * e4 + j;
          const int k = k0 + wk * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          if (k < a.K && cc < a.C) out[((long long)k * 9 + t9) * a.C + cc] = acc[t9][e] + v[j];
        }
      }
  }
}

// DW = sum over the splits' slabs, in split order (16-byte vectors; n4 = K*9*C / 4)
__global__ __launch_bounds__(256) void wgrad_nine_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                               long long n4, int splits) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 s = reinterpret_cast<const f32x4*>(partial)[i];
    for (int sp = 1; sp < splits; ++sp) {
      const f32x4 v = reinterpret_cast<const f32x4*>(partial)[i + (long long)sp * n4];
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    reinterpret_cast<f32x4*>(dw)[i] = s;
  }
}

}  // namespace jpdse
