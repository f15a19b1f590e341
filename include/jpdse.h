/*
 * jpdse.h -- C ABI of libjpdse_hip.so: the MI355X (gfx950) kernels behind the JPD-SE
 * training hot path.
 *
 * The reference (SenseBrain/JPD-SE) is pure Python on torch.nn -> cuDNN and has no FFI of
 * its own; each entry point below replaces the torch.nn call sites cited next to it
 * (paths relative to the reference root, `networks.py` =
 * ctu/models/pix2pixHD_networks/networks.py, `model.py` = ctu/models/pix2pixHD_model.py).
 *
 * Conventions
 *   - every function returns 0 on success or a negative JPDSE_E* code; the message is
 *     available from jpdse_last_error() (thread local);
 *   - nothing here allocates, frees, synchronises or throws: all buffers are caller-owned
 *     device memory, borrowed for the duration of the enqueue on `stream` (hipStream_t
 *     passed as void*);
 *   - activations are NHWC, channel count stored rounded up to a multiple of 8
 *     (JPDSE_CPAD): storage channels >= logical channels, the padding lanes are zero;
 *   - filters are passed as fp32 "master" tensors in KRSC order (= the memory order of a
 *     torch OIHW tensor in channels_last format; for ConvTranspose2d's IOHW weight the same
 *     memory order is [Cin][R][S][Cout]) and packed per step into compute-dtype GEMM panels;
 *   - dtype: JPDSE_F32 computes on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains),
 *     JPDSE_BF16 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; statistics, loss
 *     reductions, weight gradients, Adam state are always fp32.
 */
#ifndef JPDSE_H_
#define JPDSE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JPDSE_ABI_VERSION 2   /* 2 (round 4): + jpdse_conv_dgrad_nsum_slots / _fused_nsums, jpdse_inorm_bwd_from_sums; round 3 had already added
                                * jpdse_loss_finalize, jpdse_conv_fwd_pool, jpdse_conv_dgrad_fused_lrelu, jpdse_input_builder, jpdse_copy, jpdse_prof_hbm_* under version 1 */

enum { JPDSE_F32 = 0, JPDSE_BF16 = 1 };
enum { JPDSE_PAD_ZERO = 0, JPDSE_PAD_REFLECT = 1 };
enum { JPDSE_ACT_NONE = 0, JPDSE_ACT_RELU = 1, JPDSE_ACT_LRELU = 2, JPDSE_ACT_TANH = 3 };
enum {
  JPDSE_OK = 0,
  JPDSE_EINVAL = -1,      /* bad descriptor / null pointer / unsupported combination */
  JPDSE_EWORKSPACE = -2,  /* workspace smaller than *_workspace_size() */
  JPDSE_ELAUNCH = -3,     /* hipLaunchKernel / runtime error (message carries hipGetErrorString) */
  JPDSE_EARCH = -4        /* device is not gfx950 */
};

#define JPDSE_CPAD(c) (((c) + 7) & ~7)

/* ---- library ------------------------------------------------------------------------ */
int jpdse_version(void);
const char* jpdse_last_error(void);
/* 0 when device `device` is gfx950; JPDSE_EARCH otherwise (no reference counterpart). */
int jpdse_arch_check(int device);

/* Kernel timer used by bench.py's roofline figure (no reference counterpart): after
 * jpdse_prof_select(1, Ks, kdim, max) every implicit-GEMM launch with CPAD(out channels) == Ks
 * and GEMM reduction length == kdim is bracketed by hipEvents on its own stream;
 * jpdse_prof_collect waits for them and returns the summed kernel time, the summed algorithmic
 * FLOPs (2*M*Ks*kdim per launch) and the number of launches, then resets the log.
 * Not thread safe; select with enable = 0 to switch it off. */
int jpdse_prof_select(int32_t enable, int32_t Ks, int64_t kdim, int32_t max_launches);
int jpdse_prof_collect(double* total_ms, double* total_flops, int64_t* launches);
/* The same sums for the other regions of the selected layer, WITHOUT resetting the log (call before jpdse_prof_collect):
 * cls 1 = the reflect ring of the data gradient -- ring_frame_kernel, or the strip GEMMs + fold (time only, their FLOPs belong to the data
 * gradient counted in class 0), cls 2 = the weight-gradient launches (K == Ks, 9 * CPAD(C) == kdim). */
int jpdse_prof_collect_class(int32_t cls, double* total_ms, double* total_flops, int64_t* launches);
/* The same for the HBM-bound calls (bench.py "roofline_hbm"; north_star: "HBM GB/s on the norm/activation kernels"): after
 * jpdse_prof_hbm_select(1, max) every jpdse_inorm_fwd / jpdse_inorm_fwd_from_moments (class JPDSE_HBM_INORM_FWD),
 * jpdse_inorm_bwd (JPDSE_HBM_INORM_BWD) and jpdse_adam_step (JPDSE_HBM_ADAM) call is bracketed by a hipEvent pair on its own
 * stream (all kernels of the call: moments, finalize, apply) until `max` regions are used.  jpdse_prof_hbm_collect waits for
 * the regions of one class and returns their summed time, their summed ALGORITHMIC bytes -- InstanceNorm forward 3 x the
 * tensor (x read by the moment pass and by the apply pass, y written; + the residual when there is one; 2 x when the moments
 * came from the conv epilogue), backward 5 x (x and dy read twice, dx written), Adam 28 B per parameter (+ 2 B when it also
 * writes the bf16 forward panel) -- and their count.  jpdse_prof_hbm_select(0, 0) switches the timer off and clears the log. */
#define JPDSE_HBM_INORM_FWD 0
#define JPDSE_HBM_INORM_BWD 1
#define JPDSE_HBM_ADAM 2
int jpdse_prof_hbm_select(int32_t enable, int32_t max_regions);
int jpdse_prof_hbm_collect(int32_t cls, double* total_ms, double* total_bytes, int64_t* regions);

/* ---- convolution family ------------------------------------------------------------- */
/* One descriptor covers nn.Conv2d as used by:
 *   ReflectionPad2d(3)+Conv2d 7x7            networks.py:210,246,160,175
 *   Conv2d 3x3 stride 2 pad 1                networks.py:215,162
 *   ReflectionPad2d(1)+Conv2d 3x3 (ResBlock) networks.py:275,283,291,298
 *   Conv2d 4x4 stride 2|1 pad 2 (PatchGAN)   networks.py:430,437,444,449
 *   VGG19 Conv2d 3x3 pad 1 + ReLU            networks.py:477-492
 * and, with the roles of fwd/dgrad swapped by the caller, nn.ConvTranspose2d 3x3 stride 2
 * pad 1 output_padding 1 (networks.py:244,170): convT.forward == dgrad of the Conv2d with
 * K=Cin_T, C=Cout_T; convT.dgrad == that Conv2d's fwd; convT.wgrad == its wgrad with
 * (x, dy) = (dy_T, x_T).  jpdse_convT_* below are exactly those aliases. */
typedef struct jpdse_conv_desc {
  int32_t dtype;      /* JPDSE_F32 | JPDSE_BF16: activations and packed filters */
  int32_t N, H, W, C; /* input  [N,H,W,CPAD(C)] */
  int32_t K;          /* output channels; output is [N,OH,OW,CPAD(K)] */
  int32_t R, S;       /* filter height, width */
  int32_t stride;     /* 1 | 2 */
  int32_t pad;        /* symmetric padding */
  int32_t pad_mode;   /* JPDSE_PAD_ZERO | JPDSE_PAD_REFLECT (reflect requires stride 1) */
  int32_t act;        /* fwd epilogue after bias: JPDSE_ACT_* */
  float slope;        /* LeakyReLU negative slope */
} jpdse_conv_desc;

int jpdse_conv_out_shape(const jpdse_conv_desc* d, int32_t* OH, int32_t* OW);
/* Introspection of the host-side plan (no device work; callable without a GPU).  Writes 54
 * int32: {Cs,Ks,Hp,Wp,OH,OW,Lk_fwd,n_phases,PT,PB,PL,PR,DH,DW} then for each of 4 stride
 * phases of the data gradient {qh,qw,Uh,Uw,i0h,cnth,i0w,cntw,Lk,pack_offset_bytes}. */
int jpdse_conv_plan_query(const jpdse_conv_desc* d, int32_t* out, int32_t n);
/* bytes of the packed forward / data-gradient filter panels */
size_t jpdse_conv_fwd_pack_size(const jpdse_conv_desc* d);
size_t jpdse_conv_dgrad_pack_size(const jpdse_conv_desc* d);
/* master fp32 KRSC -> compute-dtype panels (either output pointer may be NULL to skip) */
int jpdse_conv_pack_weights(const jpdse_conv_desc* d, const float* w_krsc, void* fwd_pack,
                            void* dgrad_pack, void* stream);
/* Batched re-packing of the data-gradient panels of many layers in ONE launch (per optimizer step every
 * conv of the stepped network needs a fresh panel).  jpdse_conv_pack_entries (host only, no GPU work)
 * appends one entry per stride phase of layer `d` to `out` and returns their number, or -1 when the
 * layer cannot be batched (a phase whose panel rows are padded): use jpdse_conv_pack_weights
 * for those.  The caller fills `block0` with the running sum of `blocks` over its whole table, uploads
 * the table once and calls jpdse_conv_pack_run(table_dev, n, total_blocks) after each optimizer step. */
typedef struct jpdse_pack_entry {
  const float* w;   /* fp32 KRSC master */
  void* out;        /* this phase's panel inside the layer's dgrad pack */
  int32_t K, Ks, C, Cs, R, S, st, qh, qw, Uh, Uw, Lk, gx, gy;
  int32_t out_f32;  /* 1: the panel is fp32 (JPDSE_F32 layers), 0: bf16 (ABI version 2) */
  int32_t reserved_;
  int64_t blocks;   /* thread blocks this entry needs */
  int64_t block0;   /* first block of this entry in the launch */
} jpdse_pack_entry;
int jpdse_conv_pack_entries(const jpdse_conv_desc* d, const float* w_krsc, void* dgrad_pack,
                            jpdse_pack_entry* out, int32_t max_entries);
int jpdse_conv_pack_run(const jpdse_pack_entry* table_dev, int32_t n_entries, int64_t total_blocks,
                        void* stream);
/* workspace needed by fwd / dgrad / wgrad (max over the three) */
size_t jpdse_conv_workspace_size(const jpdse_conv_desc* d);
/* y = act(conv(pad(x)) + bias); bias may be NULL (convs feeding an affine-less
 * InstanceNorm: the bias is cancelled exactly by the mean subtraction). */
int jpdse_conv_fwd(const jpdse_conv_desc* d, const void* x, const void* fwd_pack,
                   const float* bias, void* y, void* ws, size_t ws_bytes, void* stream);
/* Forward of a conv that feeds an affine-less InstanceNorm, with the norm's moment pass fused into the conv's epilogue
 * (networks.py:204,216,245: conv -> norm): also writes moments[n][c][slot] = (mean, M2 = sum (y - mean)^2) of the STORED
 * (rounded) values over the pixels of block `slot` of image n -- every slot covers H*W / slots pixels; taken about a per-block
 * pilot value, so a channel with |mean| >> std keeps its variance --, fp32 [N][CPAD(K)][slots][2] with slots = jpdse_conv_moment_slots(d) (0: this layer's kernel has no such
 * epilogue -- use jpdse_conv_fwd + jpdse_inorm_fwd).  No bias (the norm cancels it), d->act must be JPDSE_ACT_NONE.
 * jpdse_inorm_fwd_from_moments then replaces jpdse_inorm_fwd's moment pass. */
int32_t jpdse_conv_moment_slots(const jpdse_conv_desc* d);
int jpdse_conv_fwd_moments(const jpdse_conv_desc* d, const void* x, const void* fwd_pack, void* y, float* moments,
                           void* ws, size_t ws_bytes, void* stream);
/* dx = d(loss)/d(x) given dy = d(loss)/d(pre-activation output) */
/* y as jpdse_conv_fwd and, in the same call, y_pool = MaxPool2d(2, 2)(y) [N][OH/2][OW/2][CPAD(K)] -- the conv -> ReLU ->
 * MaxPool2d chain of VGG19 (networks.py:477-492).  Kernels that keep their output tile in LDS write both from one epilogue
 * (the pool pass over y is saved); the others run the pool kernel behind the conv.  Equal to jpdse_conv_fwd + jpdse_maxpool2_fwd. */
int jpdse_conv_fwd_pool(const jpdse_conv_desc* d, const void* x, const void* fwd_pack, const float* bias,
                        void* y, void* y_pool, void* ws, size_t ws_bytes, void* stream);
int jpdse_conv_dgrad(const jpdse_conv_desc* d, const void* dy, const void* dgrad_pack, void* dx,
                     void* ws, size_t ws_bytes, void* stream);
/* dx = conv_dgrad(dy) * (x > 0): `x` is this conv's own INPUT [N,H,W,CPAD(C)], which must be a
 * ReLU output (or a max-pool of one).  Equals jpdse_conv_dgrad followed by the backward of the
 * ReLU that produced x -- the chain conv -> ReLU(inplace) -> conv of VGG19 (networks.py:477-492)
 * -- with the mask applied in the GEMM epilogue instead of a separate pass over dx. */
int jpdse_conv_dgrad_relu(const jpdse_conv_desc* d, const void* dy, const void* dgrad_pack,
                          const void* x, void* dx, void* ws, size_t ws_bytes, void* stream);
/* dx = (conv_dgrad(dy) + addend) * (x > 0); `x` (ReLU mask, see above) and `addend` (same shape as dx)
 * may each be NULL.  The addend is the other branch of a gradient fan-in: the skip connection of a
 * ResnetBlock (`out = x + self.conv_block(x)`, networks.py:303-305) or the loss gradient arriving at
 * a VGG19 tap -- summed in the GEMM epilogue instead of a separate add pass. */
int jpdse_conv_dgrad_fused(const jpdse_conv_desc* d, const void* dy, const void* dgrad_pack,
                           const void* x, const void* addend, void* dx, void* ws, size_t ws_bytes,
                           void* stream);
/* LeakyReLU form of the above: dx = r * (x > 0 ? 1 : slope) with r = conv_dgrad(dy) + addend rounded to
 * the tensor dtype.  `x` (required) is this conv's own input, the OUTPUT of LeakyReLU(slope) -- the chain
 * Conv2d -> LeakyReLU(0.2) -> Conv2d of NLayerDiscriminator (networks.py:430-437) -- so dx is the
 * gradient w.r.t. that LeakyReLU's pre-activation; equal, bit for bit, to jpdse_conv_dgrad_fused(x = NULL)
 * followed by jpdse_act_bwd(JPDSE_ACT_LRELU). */
int jpdse_conv_dgrad_fused_lrelu(const jpdse_conv_desc* d, const void* dy, const void* dgrad_pack,
                                 const void* x, float slope, const void* addend, void* dx, void* ws,
                                 size_t ws_bytes, void* stream);
/* jpdse_conv_dgrad_fused whose output dx is the gradient w.r.t. the OUTPUT of an InstanceNorm2d (+ activation): the chains
 * conv -> InstanceNorm -> ReLU -> [pad] conv and x + conv_block(x) of ResnetBlock (networks.py:283-305).  The norm's backward
 * needs, per (image, channel), sum dz and sum dz * yhat with yhat = (norm_x - mean) rstd and dz = dx act'(yhat); the epilogue
 * that writes dx forms them per block from the values it stores and norm_x (the norm's INPUT, same shape as dx), norm_stats
 * ([N][CPAD(C)][2] = (mean, rstd) of jpdse_inorm_fwd) and writes sums [N][CPAD(C)][slots][2], slots =
 * jpdse_conv_dgrad_nsum_slots(d) (0: this layer's data-gradient kernel has no such epilogue -- use jpdse_conv_dgrad_fused +
 * jpdse_inorm_bwd).  jpdse_inorm_bwd_from_sums then replaces jpdse_inorm_bwd's own pass over (x, dy) for the sums. */
int32_t jpdse_conv_dgrad_nsum_slots(const jpdse_conv_desc* d);
int jpdse_conv_dgrad_fused_nsums(const jpdse_conv_desc* d, const void* dy, const void* dgrad_pack, const void* x,
                                 const void* addend, void* dx, const void* norm_x, const float* norm_stats,
                                 int32_t norm_act, float norm_slope, float* sums, void* ws, size_t ws_bytes, void* stream);
/* dw (fp32, KRSC master layout) = d(loss)/d(w); overwritten (beta = 0) */
int jpdse_conv_wgrad(const jpdse_conv_desc* d, const void* x, const void* dy, float* dw_krsc,
                     void* ws, size_t ws_bytes, void* stream);

/* nn.ConvTranspose2d aliases: `d` describes the *underlying* Conv2d (its input is the
 * transposed conv's OUTPUT: N,H,W,C = output geometry, K = transposed conv's Cin). */
int jpdse_convT_fwd(const jpdse_conv_desc* d, const void* x, const void* dgrad_pack, void* y,
                    void* ws, size_t ws_bytes, void* stream);
/* jpdse_convT_fwd with the moments of its output (see jpdse_conv_fwd_moments); slots for the UNDERLYING conv descriptor */
int32_t jpdse_convT_moment_slots(const jpdse_conv_desc* d);
int jpdse_convT_fwd_moments(const jpdse_conv_desc* d, const void* x, const void* dgrad_pack, void* y, float* moments,
                            void* ws, size_t ws_bytes, void* stream);
int jpdse_convT_dgrad(const jpdse_conv_desc* d, const void* dy, const void* fwd_pack, void* dx,
                      void* ws, size_t ws_bytes, void* stream);
int jpdse_convT_wgrad(const jpdse_conv_desc* d, const void* x, const void* dy, float* dw,
                      void* ws, size_t ws_bytes, void* stream);

/* ---- InstanceNorm2d(affine=False, eps) fused with activation / residual --------------- */
/* networks.py:31 (norm), :204,:216,:229,:245 (ReLU), :438,:446 (LeakyReLU 0.2),
 * :304 (x + conv_block(x): residual added after the second norm of a ResnetBlock). */
typedef struct jpdse_inorm_desc {
  int32_t dtype;
  int32_t N, H, W, C;
  int32_t act;          /* NONE | RELU | LRELU */
  float slope;
  float eps;            /* 1e-5 */
  int32_t has_residual; /* y = act(norm(x)) + residual */
} jpdse_inorm_desc;
size_t jpdse_inorm_workspace_size(const jpdse_inorm_desc* d);
/* stats: fp32 [N][CPAD(C)][2] = (mean, rstd), kept for backward */
int jpdse_inorm_fwd(const jpdse_inorm_desc* d, const void* x, const void* residual, void* y,
                    float* stats, void* ws, size_t ws_bytes, void* stream);
/* jpdse_inorm_fwd with the moment pass replaced by the per-block moments a conv epilogue wrote (jpdse_conv_fwd_moments):
 * a finalize over the `slots` blocks of each (image, channel) -- Chan's parallel-variance merge of the (mean, M2) slots in a
 * fixed order --, then the apply pass. */
int jpdse_inorm_fwd_from_moments(const jpdse_inorm_desc* d, const void* x, const float* moments, int32_t slots,
                                 const void* residual, void* y, float* stats, void* stream);
/* jpdse_inorm_bwd with the per-channel sums taken from the per-block slots a data-gradient epilogue wrote
 * (jpdse_conv_dgrad_fused_nsums): the slots of each (image, channel) are added in slot order, then the apply pass.
 * ws: jpdse_inorm_workspace_size(d) bytes. */
int jpdse_inorm_bwd_from_sums(const jpdse_inorm_desc* d, const void* x, const float* stats, const void* dy,
                              const float* sums, int32_t slots, void* dx, void* ws, size_t ws_bytes, void* stream);
/* dx from (x, stats, dy); the residual branch's gradient is dy itself */
int jpdse_inorm_bwd(const jpdse_inorm_desc* d, const void* x, const float* stats, const void* dy,
                    void* dx, void* ws, size_t ws_bytes, void* stream);

/* ---- pooling ------------------------------------------------------------------------ */
/* nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False)  networks.py:180,387 */
int jpdse_avgpool3s2_fwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, const void* x,
                         void* y, void* stream);
int jpdse_avgpool3s2_bwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, const void* dy,
                         void* dx, void* stream);
/* nn.MaxPool2d(2,2) inside VGG19 features (networks.py:477) */
int jpdse_maxpool2_fwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, const void* x,
                       void* y, void* stream);
int jpdse_maxpool2_bwd(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, const void* x,
                       const void* dy, void* dx, void* stream);

/* ---- elementwise pieces of the loss graph ----------------------------------------------- */
/* dz = dy * act'(.) evaluated from the activation OUTPUT y (ReLU, LeakyReLU, Tanh backward) */
int jpdse_act_bwd(int32_t dtype, int64_t n, int32_t act, float slope, const void* y, const void* dy,
                  void* dz, void* stream);
/* out = a + b (gradient fan-in where autograd would add) */
int jpdse_add(int32_t dtype, int64_t n, const void* a, const void* b, void* out, void* stream);
/* out[c] = sum over pixels of dy[p][c]  (bias gradients), fp32 [CPAD(C)] */
size_t jpdse_channel_sum_workspace_size(int64_t npix, int32_t C);
int jpdse_channel_sum(int32_t dtype, int64_t npix, int32_t C, const void* dy, float* out, void* ws,
                      size_t ws_bytes, void* stream);
/* dst[p][dst_c0 + i] = src[p][src_c0 + i], i < nch: torch.cat along channels
 * (model.py:455,595,733) and its backward (slice). */
int jpdse_channel_copy(int32_t dtype, int64_t npix, const void* src, int32_t src_cs, int32_t src_c0,
                       void* dst, int32_t dst_cs, int32_t dst_c0, int32_t nch, void* stream);
/* out = base with channels [c0, c0+nch) taken from img: `torch.cat((input_label, image), dim=1)`
 * (pix2pixHD_model.py:456,595) when the label/edge channels of `base` are already in place --
 * one pass instead of a copy of base plus a channel copy.  cs / img_cs: channel storage of
 * base,out / img. */
int jpdse_concat_channels(int32_t dtype, int64_t npix, const void* base, int32_t cs, const void* img,
                          int32_t img_cs, int32_t c0, int32_t nch, void* out, void* stream);
/* dst = src, nbytes a multiple of 16, both 16-byte aligned (assembling the batched [fake ; real] VGG19 input: networks.py:124-139) */
int jpdse_copy(int64_t nbytes, const void* src, void* dst, void* stream);
/* fill n elements with zero */
int jpdse_zero(int32_t dtype, int64_t n, void* p, void* stream);
/* dst = (dst_dtype) src, fp32 <-> bf16, n a multiple of 8: the gradient buckets of the optional bf16 all-reduce
 * (no reference counterpart: the reference has no collective, ctu/parsers/base_parser.py:234-237) */
int jpdse_cast(int32_t src_dtype, int32_t dst_dtype, int64_t n, const void* src, void* dst, void* stream);

/* ---- API-boundary layout conversion ---------------------------------------------------- */
/* fp32 NCHW (the x_dict tensors of ctu_dataset.py:124-128) -> NHWC compute dtype, C padded */
int jpdse_nchw_to_nhwc(int32_t dtype, int32_t N, int32_t C, int32_t H, int32_t W, const float* src,
                       void* dst, void* stream);
int jpdse_nhwc_to_nchw(int32_t dtype, int32_t N, int32_t C, int32_t H, int32_t W, const void* src,
                       float* dst, void* stream);
/* One-hot scatter of the label map + 4-neighbour instance-edge map, written to channels
 * [0, num_labels] of an NHWC tensor with `cs` storage channels (model.py:375-394,774-783).
 * label: fp32 [N,1,H,W] integer-valued; instance: int64 [N,1,H,W]. */
int jpdse_onehot_edge(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t num_labels,
                      const float* label, const int64_t* instance, void* dst, int32_t cs,
                      void* stream);

/* The whole input builder of a train step in ONE pass (model.py:375-394 one-hot + edges, :595 and :456 the two torch.cat):
 * for i < n_dst (1..3), dst[i] ([N,H,W,cs] NHWC) receives channels [0, num_labels) = one-hot(label), channel num_labels =
 * instance edge, channels [c0, c0 + nch) = img[i] ([N,H,W,img_cs] NHWC, same dtype; NULL: those lanes are zeroed and filled
 * in later by jpdse_insert_channels), every other lane zero.  The generator input and both halves of the discriminator
 * input are three destinations of one call: the label planes are read once and no intermediate "base" tensor exists. */
int jpdse_input_builder(int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t num_labels, const float* label,
                        const int64_t* instance, int32_t n_dst, void* const* dst, const void* const* img, int32_t cs,
                        int32_t img_cs, int32_t c0, int32_t nch, void* stream);
/* dst[..., c0 : c0 + nch] = img[..., 0 : nch] in place (the generated image into the discriminator input, model.py:456):
 * touches only the 16-byte vectors of dst that hold those channels. */
int jpdse_insert_channels(int32_t dtype, int64_t npix, void* dst, int32_t cs, const void* img, int32_t img_cs, int32_t c0,
                          int32_t nch, void* stream);

/* ---- losses ------------------------------------------------------------------------- */
/* All reductions are fp32 and deterministic (two-stage).  `count` is the LOGICAL element
 * count the mean divides by (padding lanes are zero in both operands).
 * out[0] = mean |a-b|  (nn.L1Loss: networks.py:131, model.py:207,218) */
size_t jpdse_loss_workspace_size(int64_t n);
/* Deferred second stage.  Every jpdse_*_fwd / jpdse_l1_fwd_bwd below accepts out == NULL: the block partials then stay in `ws`
 * (which must not be reused before they are consumed) and ONE jpdse_loss_finalize call reduces any number of such terms --
 * out[0] = inv_count * sum(partial[0 .. n)) in index order, per term -- instead of one tiny launch per term (the train step has 20:
 * pix2pixHD_model.py:205-221).  n = jpdse_loss_partial_count(work items): 16-byte vectors of `a` for the L1 / MSE terms, pixels for
 * jpdse_mse_const_fwd.  `terms` is a HOST array (copied into the launch). */
typedef struct jpdse_loss_term {
  const float* partial;
  int32_t n;
  float inv_count;
  float* out;
} jpdse_loss_term;
int32_t jpdse_loss_partial_count(int64_t work_items);
int jpdse_loss_finalize(const jpdse_loss_term* terms, int32_t n_terms, void* stream);
int jpdse_l1_fwd(int32_t dtype, int64_t n, int64_t count, const void* a, const void* b, float* out,
                 void* ws, size_t ws_bytes, void* stream);
/* da = scale * (*gout) * sign(a-b) / count ; gout is a DEVICE scalar (no host sync) */
int jpdse_l1_bwd(int32_t dtype, int64_t n, int64_t count, const void* a, const void* b,
                 const float* gout, float scale, void* da, void* stream);
/* as jpdse_l1_bwd when `a` is the output of a ReLU and the gradient is wanted w.r.t. its
 * pre-activation: zero where a <= 0 (VGGLoss taps relu{1..5}_1, networks.py:124-139 -- fuses the
 * ReLU backward of torchvision's in-place nn.ReLU into the loss gradient). */
int jpdse_l1_bwd_relu(int32_t dtype, int64_t n, int64_t count, const void* a, const void* b,
                      const float* gout, float scale, void* da, void* stream);
/* L1Loss forward and backward in ONE pass over a, b: out[0] = mean|a-b| and
 * da = scale/count * sign(a-b) (* [a > 0] when relu_a).  Valid because the upstream gradient of every
 * L1 term of loss_G is a constant known at forward time (the loss weights of
 * pix2pixHD_trainer.py:48-56): saves re-reading both operands in the backward pass. */
int jpdse_l1_fwd_bwd(int32_t dtype, int64_t n, int64_t count, const void* a, const void* b, float* out,
                     float scale, int32_t relu_a, void* da, void* ws, size_t ws_bytes, void* stream);
/* out[0] = mean (a-b)^2 (nn.MSELoss distortion: model.py:219-220) */
int jpdse_mse_fwd(int32_t dtype, int64_t n, int64_t count, const void* a, const void* b, float* out,
                  void* ws, size_t ws_bytes, void* stream);
int jpdse_mse_bwd(int32_t dtype, int64_t n, int64_t count, const void* a, const void* b,
                  const float* gout, float scale, void* da, void* stream);
/* LSGAN: out[0] = mean over the LOGICAL channel 0 of (x - target)^2 (networks.py:90,112-119).
 * x is the 1-channel PatchGAN map stored with `cs` channels. */
int jpdse_mse_const_fwd(int32_t dtype, int64_t npix, int32_t cs, float target, const void* x,
                        float* out, void* ws, size_t ws_bytes, void* stream);
int jpdse_mse_const_bwd(int32_t dtype, int64_t npix, int32_t cs, float target, const void* x,
                        const float* gout, float scale, void* dx, void* stream);

/* Evaluation distortion on de-normalised, clipped, uint8-TRUNCATED images (ctu/utils/misc.py:64-95 `tensor2im`,
 * pix2pixHD_model.py:636-641): q(x) = uint8(clip((x * std[c] + mean[c]) * 255, 0, 255)) evaluated in IEEE double as
 * numpy does, then out[0] = mean |q(a) - q(b)| (mse = 0) or mean (q(a) - q(b))^2 (mse = 1) over npix * C elements,
 * on the 0..255 scale.  a, b: NHWC images with CPAD(C) storage channels, each fp32 or bf16; mean / std: HOST arrays
 * of C doubles (opt.normalize_mean / opt.normalize_std).  Replaces two device->host copies + numpy per call. */
size_t jpdse_quant_loss_workspace_size(void);
int jpdse_quant_loss(int32_t dtype_a, int32_t dtype_b, int64_t npix, int32_t C, const void* a, const void* b,
                     const double* mean, const double* std, int32_t mse, float* out, void* ws, size_t ws_bytes,
                     void* stream);

/* ---- optimizer ---------------------------------------------------------------------- */
/* torch.optim.Adam (model.py:275,279) over a table of tensors, one launch.  `table` is a
 * device array of jpdse_adam_entry; bias corrections are computed from `step` (1-based). */
typedef struct jpdse_adam_entry {
  float* p;       /* parameter (fp32 master) */
  const float* g; /* gradient */
  float* m;       /* exp_avg */
  float* v;       /* exp_avg_sq */
  int64_t n;      /* elements */
  int64_t block0; /* first 1024-element block of this tensor in the launch */
  void* cast_bf16; /* optional: bf16 copy of the UPDATED parameter, same element order (the forward
                    * GEMM panel of a conv whose panel is a plain cast of the KRSC master); NULL = none */
} jpdse_adam_entry;
int jpdse_adam_step(const jpdse_adam_entry* table, int32_t n_entries, int64_t total_blocks,
                    float lr, float beta1, float beta2, float eps, int32_t step, float grad_scale,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JPDSE_H_ */
