/* atomai_amd.h — C ABI of libatomai_amd.so: the MI355X (gfx950) hot path of pycroscopy/atomai.
 *
 * The reference has NO native/FFI interface (it is pure Python on stock PyTorch), so
 * the seam these entry points replace is the set of ATen operator calls made by the reference's
 * nn.Modules and loss callables on the Segmentor / rVAE / DKL paths.  Each entry point cites the
 * reference lines whose work it takes over.  The Python host (atomai_amd/*.py, mirroring the
 * reference's module/trainer API) binds this header with ctypes (atomai_amd/_lib.py parses THIS file).
 *
 * Conventions
 *   - plain C: pointers, ints, floats; no torch / C++ types.  Every function returns int:
 *       0 = ok, >0 = hipError_t from the launch, <0 = -(index of the offending argument group).
 *   - the caller owns all memory (device pointers of torch-allocated buffers, incl. workspaces);
 *     nothing allocates, nothing synchronises; the last argument is the hipStream_t to launch on.
 *   - activations are NHWC fp32 with the channel count padded to a multiple of 4 ("Cs"), padding
 *     channels hold zeros.  Weights stay OIHW fp32 at the ABI (state-dict layout) and are re-imaged by
 *     amx_pack_weights.  "scale/shift" pairs are the producer's BatchNorm affine, applied by the
 *     consumer while loading (zero padding stays zero after the affine).
 */
#ifndef ATOMAI_AMD_H
#define ATOMAI_AMD_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- error text: message of the last failed call on the CALLING THREAD, e.g.
 * "conv2d_common: bad argument group (code -3)" or the hipGetErrorString of a failed launch; "" if none.
 * The pointer stays valid for the life of the thread; amx_clear_error resets it. */
const char* amx_last_error(void);
int amx_clear_error(void);

/* ---- launch-plan switches (no reference counterpart: stock PyTorch picks its kernels inside ATen).
 * Every AMX_* environment variable the library understands is a row of ONE table (csrc/knobs.hip; DESIGN.md §4
 * "Switches").  The table is read once, on the first call that needs a plan, under a lock, into an immutable plan that
 * every launch reads afterwards — no getenv on the launch path, the same plan on every host thread / stream.
 *   amx_knob(name)      value of the resolved switch `name` ("AMX_CONV_WS", ...); INT_MIN for an unknown name
 *   amx_knob_count()    number of rows;  amx_knob_name(i)  environment name of row i (NULL out of range)
 *   amx_knobs_reload()  re-reads the environment: the hook of in-process A/B scripts and plan-comparison tests
 *   amx_knobs_generation()  how often the table has been resolved so far (a cache of amx_knob values holds for one value)
 * Parsing: a variable that is unset OR empty takes the row's default; anything else goes through atoi (so "0" is a value,
 * "" is not — rounds 1-4 read "" as 0). */
int amx_knob(const char* name);
long amx_knobs_generation(void);
int amx_knob_count(void);
const char* amx_knob_name(int i);
int amx_knobs_reload(void);

/* ---- convolution: nn.Conv2d(k=3|1, padding=dilation) + bias + LeakyReLU + BN batch statistics,
 * reading torch.cat([src0, src1], 1) with each source's BN affine applied on load.
 * atomai/nets/blocks.py:61-76 (ConvBlock), :122-132 (UpsampleBlock 1x1), :300-318 (DilatedBlock);
 * atomai/nets/fcnn.py:132-138,223 (cat).  Also the data-gradient engine (weights packed with mode 1):
 * y/y1 are then the gradients w.r.t. src0/src1.  stats: [amx_conv2d_stats_rows][2][round_up(cout,16)], one row
 * (sum, M2) per 16-wide x th-high pixel strip (th = amx_conv2d_tile_h: the rows one wave owns); amx_bn_finalize /
 * amx_bn_stats_merge take th in rows_pix (mode 0).  stats and addend are mutually exclusive. */
int amx_conv2d_fwd(const float* x0, const float* sc0, const float* sh0, int C0s,
                   const float* x1, const float* sc1, const float* sh1, int C1s,
                   const float* wpk, const float* bias, const float* addend,
                   float* y, int Y0s, float* y1, int Y1s, float* stats,
                   int N, int H, int W, int cout, int taps, int dil, float slope, void* stream);
/* Eval-mode last layer with the classification head fused into its epilogue (atomai/nets/fcnn.py:139-142, 224-226 +
 * the sigmoid / softmax / permute of atomai/predictors/predictor.py:219-229): out = head(lrelu(conv + bias)), the
 * activation itself is not stored.  hw [K][round_up(cout,16)] / hb [K]: the final 1x1 convolution with the layer's own
 * eval-mode BatchNorm affine folded in (hw[k][c] = Wpx[k][c] * scale[c], hb[k] = bpx[k] + sum_c Wpx[k][c] * shift[c]).
 * mode 0: logits NCHW; 1: probabilities NHWC.  K <= 3.  Only for layers amx_conv2d_head_supported returns 1 for
 * (plain 3x3, one cout block, <= 32 couts). */
int amx_conv2d_head_supported(int Cin_s, int cout, int taps, int dil, int H);
int amx_conv2d_fwd_head(const float* x0, const float* sc0, const float* sh0, int C0s,
                        const float* x1, const float* sc1, const float* sh1, int C1s,
                        const float* wpk, const float* bias, const float* hw, const float* hb, float* out,
                        int K, int mode, int N, int H, int W, int cout, float slope, void* stream);
/* Eval-mode LAST layer of a DilatedBlock with the block's output fused into its epilogue (atomai/nets/blocks.py:321-329:
 * the block returns the sum of every sub-layer output): y = sum over the block's layers l of
 * [pre_l + a_l + (a_l * sc_l + sh_l)], pre_l = inverse LeakyReLU of a_l.  prev: n (1..3) device pointers to the earlier
 * layers' activations (shape of y); sc / sh: n + 1 eval-mode BatchNorm affines (round_up(cout,4) floats each, this layer's
 * last; zero vectors without BatchNorm).  Only for layers amx_conv2d_dsum_supported returns 1 for. */
int amx_conv2d_dsum_supported(int Cin_s, int cout, int taps, int dil, int H);
int amx_conv2d_fwd_dsum(const float* x0, const float* sc0, const float* sh0, int C0s, const float* wpk,
                        const float* bias, const float* const* prev, const float* const* sc, const float* const* sh,
                        int n, float* y, int N, int H, int W, int cout, int dil, float slope, void* stream);
int amx_conv2d_dgrad(const float* dpre, int Cs, const float* wpk, const float* addend, float* y, int Y0s, float* y1,
                     int Y1s, int N, int H, int W, int taps, int dil, void* stream);
/* Data gradient of a conv -> LeakyReLU -> BatchNorm layer with the BatchNorm / LeakyReLU backward formed by the loader:
 * the convolution input is dpre = lrelu'(a) * (k1*dy + k2*a + k3) (amx_bn_bwd_apply's arithmetic, bit-identical), read
 * from dy (gradient w.r.t. the layer output) and the saved activation a, never written to HBM.  Only for launches the
 * wave-specialised kernel takes (amx_conv2d_dgrad_fused_supported == 1: plain 3x3, 16 / 32 channels either side, image
 * sides in multiples of 16, no addend); together with amx_conv2d_wgrad_fused it replaces the amx_bn_bwd_apply pass of
 * those layers (autograd of blocks.py:61-76 through nn.BatchNorm2d / nn.LeakyReLU). */
int amx_conv2d_dgrad_fused_supported(int Cs, int Y0s, int Y1s, int N, int H, int W, int taps, int dil);
int amx_conv2d_dgrad_fused(const float* dy, const float* aux, const float* k1, const float* k2, const float* k3,
                           float bslope, int Cs, const float* wpk, float* y, int Y0s, float* y1, int Y1s,
                           int N, int H, int W, int taps, int dil, void* stream);
/* amx_conv2d_dgrad_fused for a launch with ONE output that is the complete gradient dy of a conv -> LeakyReLU -> BatchNorm
 * layer's output: the epilogue also reads that layer's saved activation bs_a (shape of y) and writes
 * amx_conv2d_dgrad_bsum_rows rows of (sum dy, sum dy * a) per channel to bs_part [rows][2][Y0s] — the input of
 * amx_bn_bwd_finalize, so amx_bn_bwd_reduce (a pass over both tensors) is not launched for that layer (autograd of
 * nn.BatchNorm2d, blocks.py:71-75).  amx_conv2d_dgrad_bsum_rows returns 0 where the form does not exist (AMX_BWD_SUMS=0,
 * or a launch the wave-specialised kernel does not take). */
int amx_conv2d_dgrad_bsum_rows(int Cs, int Y0s, int N, int H, int W, int taps, int dil);
int amx_conv2d_dgrad_fused_bsum(const float* dy, const float* aux, const float* k1, const float* k2, const float* k3,
                                float bslope, int Cs, const float* wpk, float* y, int Y0s, int N, int H, int W, int taps,
                                int dil, const float* bs_a, float* bs_part, void* stream);
int amx_conv2d_tile_h(int Cin_s, int cout, int taps, int dil, int H);
int amx_conv2d_num_tiles(int N, int H, int W, int th);
/* Dilations 2 / 4 / 6 run as d*d plain 3x3 convolutions on the residue-class sub-images x[ry::d, rx::d]; their
 * statistics rows are ordered [n][ry][rx][strip][tx].  amx_conv2d_stats_lattice: d for such a layer, else 0;
 * amx_conv2d_stats_rows: the number of rows amx_conv2d_fwd writes (== amx_conv2d_num_tiles(N,H,W,tile_h) when 0). */
/* Stride of the x-packed lattice tiles a dilated launch of this width uses (0: one residue-class sub-image per tile axis):
 * the `dil` sub-images of a residue row share one tile axis when that saves tile columns and no statistics are written
 * (AMX_CONV_XPACK, default 1; results are bit-identical either way). */
int amx_conv2d_lattice_xpack(int W, int dil, int has_stats);
int amx_conv2d_stats_lattice(int taps, int dil);
int amx_conv2d_stats_rows(int Cin_s, int cout, int taps, int dil, int N, int H, int W);
/* Diagnostic: launches of amx_conv2d_fwd / amx_conv2d_dgrad that the wave-specialised kernel of the thin plain-3x3
 * layers (conv_ws.hip) has taken since the library was loaded (AMX_CONV_WS=0 routes them to the general kernel). */
long amx_conv2d_ws_launches(void);
/* Diagnostic: launches taken by the remainder-column classes (conv_kernel.h, REM: widths of 25 / 50 filters in one cout
 * block of 16 + 3 x 4 / 3 x 16 + 4 columns, the 4-wide blocks on v_mfma_f32_4x4x1; AMX_CONV_REM=0 restores the padded
 * 32 / 2 x 32 column plan). */
long amx_conv2d_rem_launches(void);

/* weight gradient of the same convolution (autograd of nn.Conv2d; trainer.py:205 loss.backward()).
 * part: [amx_conv2d_wgrad_rows][taps][round_up(C0s+C1s,16)][round_up(cout,16)] partial rows. */
int amx_conv2d_wgrad(const float* x0, const float* sc0, const float* sh0, int C0s,
                     const float* x1, const float* sc1, const float* sh1, int C1s,
                     const float* dpre, int Dos, float* part, int N, int H, int W, int cout,
                     int taps, int dil, void* stream);
int amx_conv2d_wgrad_fused(const float* x0, const float* sc0, const float* sh0, int C0s,
                           const float* x1, const float* sc1, const float* sh1, int C1s,
                           const float* dy, const float* aux, const float* k1, const float* k2, const float* k3,
                           float bslope, int Dos, float* part, float* bpart, int N, int H, int W, int cout,
                           int taps, int dil, void* stream);
/* ResBlock (atomai/nets/blocks.py:135-214: conv -> BatchNorm -> LeakyReLU, residual add): the forward conv and the
 * weight gradient with a LeakyReLU applied after the on-load BatchNorm affine of each source (slope 1.0f = none) */
int amx_conv2d_fwd_act(const float* x0, const float* sc0, const float* sh0, float in_slope0, int C0s,
                       const float* x1, const float* sc1, const float* sh1, float in_slope1, int C1s,
                       const float* wpk, const float* bias, const float* addend,
                       float* y, int Y0s, float* y1, int Y1s, float* stats,
                       int N, int H, int W, int cout, int taps, int dil, float slope, void* stream);
int amx_conv2d_wgrad_act(const float* x0, const float* sc0, const float* sh0, float in_slope0, int C0s,
                         const float* x1, const float* sc1, const float* sh1, float in_slope1, int C1s,
                         const float* dy, const float* aux, const float* k1, const float* k2, const float* k3,
                         float bslope, int Dos, float* part, float* bpart, int N, int H, int W, int cout,
                         int taps, int dil, void* stream);
/* out = LeakyReLU(t*scale + shift + r)  (bn2 affine + residual add + activation, blocks.py:210-213), and
 * din = dout * LeakyReLU'(ref*scale + shift)  (scale == NULL: the sign of ref itself); din2 = optional copy */
int amx_res_out_fwd(const float* t, const float* scale, const float* shift, const float* r, float slope,
                    long npix, int Cs, float* out, void* stream);
int amx_lrelu_bwd(const float* dout, const float* ref, const float* scale, const float* shift, float slope,
                  long npix, int Cs, float* din, float* din2, void* stream);
/* ResHedNet side outputs (atomai/nets/fcnn.py:283-295): F.interpolate(score, size=(H, W), mode) with the producer's
 * pending BatchNorm affine applied on load, written into channels [coff, coff + C) of the concatenated NHWC tensor
 * dst (N,H,W,Cd); and its deterministic gather-form backward (dsrc: (N,h,w,Cs), gradient after the affine).
 * mode 0 bilinear (align_corners=False), 1 nearest. */
int amx_resize_cat_fwd(const float* src, const float* scale, const float* shift, int N, int h, int w, int Cs, int C,
                       float* dst, int H, int W, int Cd, int coff, int mode, void* stream);
int amx_resize_cat_bwd(const float* ddst, int N, int H, int W, int Cd, int coff, int C, float* dsrc, int h, int w,
                       int Cs, int mode, void* stream);
int amx_conv2d_wgrad_rows(int N, int H, int W, int Cin_s, int cout, int taps, int dil);
int amx_conv2d_wgrad_ksplit(int N, int H, int W, int Cin_s, int cout, int taps, int dil);
/* Diagnostic: launches of amx_conv2d_wgrad* that the wave-specialised kernel of the plain 3x3 classes (wgrad_ws.hip:
 * consumer waves issue MFMAs only, producer waves load / transform / stage the next tile) has taken since the library
 * was loaded (AMX_WGRAD_WS=0 routes them to wgrad_kernel.h; the partial rows are bit-identical either way). */
long amx_conv2d_wgrad_ws_launches(void);
int amx_wgrad_reduce(const float* part, int rows, int taps, int ci_pad, int co_pad, int C0, int C0s,
                     int C1, int Cout, float* dw, void* stream);

/* first layer, Cin == 1 (Conv2d(1, F, 3) of c1: fcnn.py:66-69,186-189) and its weight gradient.
 * in_sub / in_div: the input is read as (x - in_sub) / in_div — the predictor's stack normalisation
 * (utils/preproc.py:822-823) without a separate pass; (0, 1) = as is. */
int amx_conv1_fwd(const float* x, const float* w, const float* bias, float* y, float* stats,
                  int N, int H, int W, int Cout, int Cs, int dil, float slope, int rows, int rows_pix,
                  float in_sub, float in_div, void* stream);
/* Eval-mode first layer followed by F.max_pool2d(x, 2, 2) (nets/fcnn.py:123, 219 — Unet / dilnet pool c1 directly):
 * y as amx_conv1_fwd without statistics AND pooled = max_pool2d(y * pscale + pshift) (N,H/2,W/2,Cs), bit-identical
 * to amx_pool2x2_fwd(y, pscale, pshift); saves reading the net's largest activation back.  Only for shapes
 * amx_conv1_fwd_pool_supported reports (even H, W; whole row pairs per block). */
int amx_conv1_fwd_pool_supported(int H, int W, int dil, int rows_pix);
int amx_conv1_fwd_pool(const float* x, const float* w, const float* bias, float* y, float* pooled,
                       const float* pscale, const float* pshift, int N, int H, int W, int Cout, int Cs, int dil,
                       float slope, int rows, int rows_pix, float in_sub, float in_div, void* stream);
int amx_conv1_wgrad(const float* x, const float* dpre, float* part, int N, int H, int W, int Cs,
                    int dil, int rows, int rows_pix, void* stream);
int amx_conv1_wgrad_fused(const float* x, const float* dy, const float* aux, const float* k1, const float* k2,
                          const float* k3, float bslope, float* part, int N, int H, int W, int Cs, int dil,
                          int rows, int rows_pix, void* stream);

/* OIHW -> MFMA weight image (mode 0 forward, 1 dgrad) and NCHW<->NHWC at the module boundary */
int amx_pack_weights(const float* w_oihw, float* dst, int cout, int C0, int C0s, int C1, int C1s,
                     int taps, int mode, void* stream);
long amx_pack_weights_size(int cout, int C0s, int C1s, int taps, int mode);
/* amx_pack_weights of the sliced weight w[:, ci_off : ci_off + Cn] (one source of a concatenation), without the slice:
 * the image of a convolution over that source alone (size amx_pack_weights_size(cout, Cns, 0, taps, mode)). */
int amx_pack_weights_range(const float* w_oihw, float* dst, int cout, int cin_total, int ci_off, int Cn, int Cns,
                           int taps, int mode, void* stream);
/* n images in one launch per 8 jobs; w / dst: host arrays of n device pointers, desc: n x (cout, C0, C0s, C1, C1s, taps,
 * mode) */
int amx_pack_weights_batch(const void* const* w, void* const* dst, const int* desc, int n, void* stream);
int amx_nchw_to_nhwc(const float* src, float* dst, int N, int C, int Cs, int H, int W, void* stream);
int amx_nhwc_to_nchw(const float* src, float* dst, int N, int C, int Cs, int H, int W, void* stream);
int amx_add_inplace(float* dst, const float* src, long n, void* stream);
/* y = (x - sub) / div: the global min-max normalisation of torch_format_image (utils/preproc.py:798-825) for a
 * chunk already on the device (two correctly rounded fp32 ops = numpy's float32 arithmetic) */
int amx_sub_div(const float* x, float* y, long n, float sub, float div, void* stream);
/* contiguous 16-byte-granular copy on at most max_wgs workgroups; dst may be pinned host memory (the predictor's
 * download of a chunk: atomai/predictors/predictor.py:99-104 copies each batch to the host) */
int amx_copy16(const void* src, void* dst, long nbytes, int max_wgs, void* stream);

/* ---- BatchNorm2d after LeakyReLU (blocks.py:71-75; torch defaults eps 1e-5, momentum 0.1) */
int amx_bn_finalize(const float* stats, int rows, int cop, int mode, int N, int H, int W, int rows_pix,
                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, int C, int Cs, float* scale, float* shift,
                    float* save_mean, float* save_invstd, void* stream);
int amx_bn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv,
                       float eps, int C, int Cs, float* scale, float* shift, void* stream);
int amx_affine_nhwc(const float* a, const float* scale, const float* shift, float* y, long npix, int Cs,
                    void* stream);
int amx_rows_for(long npix);
int amx_rows_pix(long npix);
int amx_bn_bwd_reduce(const float* dy, const float* a, long npix, int Cs, float* part, void* stream);
int amx_bn_bwd_finalize(const float* part, int rows, int stride, int Cs, int C, long npix, const float* gamma,
                        const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                        float* k1, float* k2, float* k3, void* stream);
int amx_bn_bwd_apply(const float* dy, const float* a, const float* gx, const float* k1, const float* k2,
                     const float* k3, float slope, long npix, int Cs, float* dpre, float* part,
                     void* stream);
int amx_reduce_rows(const float* part, int rows, int stride, int C, float scale, float* out, void* stream);
/* mode 0: conv rows [n][strip][tx]; 1: rows of rows_pix consecutive pixels; 3: lattice-mode conv rows
 * [n][ry][rx][strip][tx] (lat = amx_conv2d_stats_lattice).  out: [nchunks][3][cop] (sum, M2, count) = finalize mode 2. */
int amx_bn_stats_merge(const float* stats, int rows, int cop, int mode, int N, int H, int W, int rows_pix, int lat,
                       int nchunks, float* out, void* stream);
int amx_reduce_rows_chunked(const float* part, int rows, long ncols, int nchunks, float* out, void* stream);
/* column sums of up to 16 partial-row tensors in two launches (chunk sums in row order, then the chunks in order: the
 * arithmetic of amx_reduce_rows_chunked applied twice): row r of segment k starts at parts[k] + r * strides[k] and has
 * cols[k] columns; outs[k]: cols[k] floats; tmp: nchunks * sum(cols) floats.  Serves the per-sample gradient partials
 * of amx_rdecoder_bwd (one segment per parameter, each summed straight into its gradient buffer). */
int amx_reduce_rows_segments(const float* const* parts, const long* cols, const long* strides, float* const* outs,
                             int nseg, int rows, int nchunks, float* tmp, void* stream);

/* ---- F.max_pool2d(x,2,2) (fcnn.py:123-127,219), F.interpolate x2 (blocks.py:130-131), DilatedBlock sum
 * (blocks.py:321-329).  mode: 0 bilinear (align_corners=False), 1 nearest. */
int amx_pool2x2_fwd(const float* a, const float* scale, const float* shift, float* d, int N, int H, int W,
                    int Cs, void* stream);
int amx_pool2x2_bwd(const float* g, const float* a, const float* scale, const float* shift,
                    const float* skip, float* dy, float* bstats, int N, int H, int W, int Cs, void* stream);
/* amx_pool2x2_bwd for the output of the net's FIRST layer (conv(1 -> F) 3x3 -> LeakyReLU -> BatchNorm -> max-pool,
 * atomai/nets/fcnn.py:83-84,121-122; the input needs no gradient) fused with that layer's weight gradient: dy is formed in
 * registers and never written; besides bstats it emits part3 [rows][3][10][Cs], the three sums the first-layer weight /
 * bias gradient is linear in (dW = k1 S1 + k2 S2 + k3 S3 with the constants amx_bn_bwd_finalize derives from bstats).
 * x: the net input [N][H][W]; rows = amx_pool2x2_bwd_rows.  amx_conv1_wgrad_combine: [rows][3][10][Cs] (after an optional
 * amx_reduce_rows_chunked) -> [10][Cs] in the row layout of amx_conv1_wgrad_fused's column sums (k1 == NULL: S1 alone). */
int amx_pool2x2_bwd_wgrad1_supported(int H, int W, int Cs, int dil);
int amx_pool2x2_bwd_wgrad1(const float* g, const float* a, const float* scale, const float* shift, const float* skip,
                           const float* x, float slope, float* bstats, float* part3, int N, int H, int W, int Cs,
                           void* stream);
int amx_conv1_wgrad_combine(const float* part3, int rows, int Cs, const float* k1, const float* k2, const float* k3,
                            float* out, void* stream);
int amx_pool2x2_bwd_rows(int N, int H, int W, int Cs);
int amx_upsample2x_fwd(const float* v, float* u, int N, int h, int w, int Cs, int mode, void* stream);
/* UpsampleBlock forward in ONE pass (atomai/nets/blocks.py:122-132; round 5): y[N][2h][2w][Cs_out] = x2 interpolation
 * (mode 0 bilinear align_corners=False / 1 nearest) of the 1x1 convolution W[Cout][Cin] (+ bias) of x[N][h][w][Cs_in] with
 * the producer's BatchNorm affine (sc, sh: Cs_in floats each, or both NULL) applied on load.  Bit-identical to
 * amx_conv2d_fwd(taps = 1) followed by amx_upsample2x_fwd; the low-resolution result never reaches HBM.
 * amx_upconv1x1_supported: Cout <= 64 (not 33..48), stored input channels = 0 or 4 mod 16 (the shapes whose summation
 * order equals the two-kernel path's), the layer's weights + one tile fit the LDS. */
int amx_upconv1x1_supported(int Cin, int Cs_in, int Cout, int Cs_out);
int amx_upconv1x1_fwd(const float* x, const float* sc, const float* sh, const float* w, const float* bias, float* y, int N,
                      int h, int w_, int Cin, int Cs_in, int Cout, int Cs_out, int mode, void* stream);
int amx_upsample2x_bwd(const float* du, float* dv, int N, int h, int w, int Cs, int mode, void* stream);
int amx_dilated_sum(const float* const* a, const float* const* scale, const float* const* shift, int n,
                    float slope, int accumulate, float* out, long npix, int Cs, void* stream);
/* the same with weights on the pre-activation / activation terms: a Dropout layer inside DilatedBlock is one more
 * summed sub-layer (blocks.py:311-312, 321-329) — eval: wpre = 2; training: an extra (wpre 1, wact 0, no scale) call
 * over the un-dropped convolution outputs. */
int amx_dilated_sum_ex(const float* const* a, const float* const* scale, const float* const* shift, int n,
                       float slope, int accumulate, float wpre, float wact, float* out, long npix, int Cs,
                       void* stream);

/* ---- head: px = Conv2d(F, nb_classes, 1) (fcnn.py:115,212); losses select_loss('ce')
 * (losses_metrics/losses.py:152-155) fused fwd+bwd; mode 1 of px = SegPredictor.forward_ probabilities
 * (predictors/predictor.py:219-229) */
int amx_px_fwd(const float* a, const float* scale, const float* shift, const float* w, const float* b,
               float* out, int N, int H, int W, int C, int Cs, int K, int mode, void* stream);
int amx_px_bwd(const float* dl, const float* a, const float* scale, const float* shift, const float* w,
               float* dxn, float* part, float* partb, float* bstats, int N, int H, int W, int C, int Cs, int K,
               int rows, int rows_pix, void* stream);
int amx_ce_fwd_bwd(const float* logits, const long long* target, float* dlogits, float* part, int rows,
                   int N, int K, long HW, void* stream);
int amx_bce_fwd_bwd(const float* logits, const float* target, float* dlogits, float* part, int rows,
                    long numel, void* stream);
/* loss.backward() hands the loss node an upstream gradient g (a device scalar, 1.0 unless the caller scaled the loss:
 * trainers/trainer.py:203-206).  x *= *g in place — and no pass at all when *g == 1 (the multiplication by one was a
 * 200 MB read + write of the logits gradient in every training step). */
int amx_scale_unless_one(float* x, const float* g, long n, void* stream);
/* the same for up to four tensors in one launch (unused slots: NULL / 0) */
int amx_scale_unless_one_multi(float* x0, long n0, float* x1, long n1, float* x2, long n2, float* x3, long n3,
                               const float* g, void* stream);
/* px forward + loss + px backward of a training step (trainers/trainer.py:201-207: prob = net(x); loss =
 * criterion(prob, y); loss.backward()) in ONE pass over the last activation: neither the logits nor their gradient are
 * written.  K >= 2: CrossEntropyLoss against target int64 [N][H][W]; K == 1: BCEWithLogitsLoss against target_f float
 * [N][1][H][W] (select_loss('ce', nb_classes), losses_metrics/losses.py:152-155).  Outputs as amx_px_bwd (dxn, part
 * [rows][K][Cs], partb [rows][K], bstats [rows][2][Cs] or NULL) for an upstream gradient of 1, plus lpart [rows] =
 * per-workgroup sums of the pixel losses (mean = sum / npix).  amx_px_ce_train_supported: 1 <= K <= 4 and Cs / 4 a power
 * of two. */
int amx_px_ce_train_supported(int Cs, int K);
int amx_px_ce_train(const float* a, const float* scale, const float* shift, const float* w, const float* b,
                    const long long* target, const float* target_f, float* dxn, float* part, float* partb, float* bstats,
                    float* lpart, int N, int H, int W, int C, int Cs, int K, int rows, int rows_pix, void* stream);
/* IoU of SegTrainer.accuracy_fn (trainers/trainer.py:727-737 -> losses_metrics/metrics.py:16-95): per-image K x K
 * confusion counts of (label, thresholded softmax / sigmoid class map) in one pass over the NCHW logits, replacing the
 * reference's host round trip (cv2.threshold per image + squeeze_channels + torch.bincount).  Exactly one of truth_i64
 * ([N][HW] class maps, K > 1) / truth_f32 ([N][HW] binary masks, K == 1) is non-NULL; hist is int32
 * [N][Kc][Kc], Kc = max(K, 2), zeroed by the caller.  activation 1: `logits` are logits (softmax / sigmoid applied
 * here, metrics.py:37-41); 0: they are probabilities already (IoU(..., activation=False)).  Any class count (more than
 * 8 classes re-read a pixel's scores instead of holding them in registers). */
int amx_iou_hist(const float* logits, const long long* truth_i64, const float* truth_f32, int N, int K, long HW,
                 float thresh, int activation, int* hist, void* stream);

/* ---- training-mode nn.Dropout of ConvBlock (Conv2d -> Dropout(p) -> LeakyReLU -> BatchNorm2d, atomai/nets/blocks.py:
 * 59-76): applied AFTER the fused conv + LeakyReLU (LeakyReLU is positively homogeneous, the mask multiplier is >= 0).
 * amx_dropout_fwd: a *= mask in place ([npix][Cs]), mask (values 0 or 1/(1-p)) kept for backward; mask_in != NULL
 * replaces the Philox generator (tests); stats != NULL: BatchNorm (sum, M2) rows of the MASKED tensor in the
 * rows_pix-pixels-per-row layout (amx_bn_finalize mode 1; rows = amx_rows_for(npix), rows_pix = amx_rows_pix(npix)).
 * amx_dropout_bwd: dpre *= mask in place. */
int amx_dropout_fwd(float* a, float* mask, const float* mask_in, float p, long seed, float* stats, long npix, int Cs,
                    int cop, int rows, int rows_pix, void* stream);
int amx_dropout_bwd(float* dpre, const float* mask, long n, void* stream);

/* ---- on-the-fly augmentation of the resident batch (datatransform.run and its apply_* steps,
 * atomai/transforms/imaug.py:108-358; hook atomai/trainers/trainer.py:339-341).  x / y are [N][H][W] fp32.
 * amx_aug_minmax: out = (min, max) of x (work: 2 * amx_aug_minmax_blocks(n) floats).
 * amx_aug_point: flip / 90-degree rotation, (x - min) / ptp, gaussian noise + clip, poisson, salt & pepper, gamma,
 *   2-D gaussian background in one pass; params [N][12] per image = (flip code -1|0|1|2 ccw|3 cw|4 none, gauss sigma,
 *   poisson scale, s&p amount, gamma, bg x0, y0, a, b, fwhm, bg amplitude, 0); 0 switches a step off.  Per-pixel
 *   randomness: Philox4x32-10 keyed by (seed, image, pixel, step), or the caller's fields f_* ([N][H][W]; NULL = use
 *   the generator): standard normals, poisson draws, and the two uniforms of salt & pepper.  jitter ([N][H] ints or
 *   NULL): row y of image n is rolled by jitter[n][y] pixels after the gaussian-noise step (apply_jitter, np.roll).
 * amx_aug_blur: one axis (0 rows, 1 columns) of scipy.ndimage.gaussian_filter(sigma[n], mode='reflect', truncate=4).
 * amx_aug_labels: the same flip / rotation on int64 class maps; present[n] |= 1 << class (caller zeroes it). */
int amx_aug_minmax_blocks(long n);
int amx_aug_minmax(const float* x, long n, float* work, float* out, void* stream);
int amx_aug_point(const float* x, float* y, const float* params, const float* mnmx, const float* f_gauss,
                  const float* f_pois, const float* f_sp1, const float* f_sp2, const int* jitter, int N, int H, int W,
                  long seed, void* stream);
int amx_aug_renorm(float* x, long n, const float* mnmx, void* stream);
int amx_aug_blur(const float* x, float* y, const float* sigma, int N, int H, int W, int axis, void* stream);
int amx_aug_labels(const long long* t, long long* out, const float* params, int* present, int N, int H, int W,
                   void* stream);
/* apply_zoom / apply_imresize (transforms/imaug.py:195-227, 276-300): cv2.resize of a per-image source window win[n] =
 * (y0, x0, h, w) of the [Hs][Ws] frame to [Hd][Wd]; mode 0 = INTER_LINEAR (what apply_imresize executes: its method
 * argument lands in cv2.resize's `dst` slot), 1 = INTER_CUBIC (a = -0.75; apply_zoom and the shrinking branch of
 * utils/img.py:cv_resize), 2 = INTER_AREA in its enlarging form (linear kernel with area-mode coefficients: the other
 * branch of cv_resize, predictors/predictor.py:203-204).  clip01: np.clip(img, 0, 1);
 * round_out: np.around (masks).  int64 class maps travel as the reference moves them: amx_aug_onehot expands them to K
 * float planes ([N][K][HW], unsqueeze_channels imaug.py:396-403), the planes are resampled like images with round_out,
 * amx_aug_squeeze forms label = sum_c c * mask_c (squeeze_channels, imaug.py:361-393) and values[n] |= 1 << label (caller
 * zeroes it; the reference keeps a pair iff exactly K distinct values occur).
 * OpenCV's arithmetic restated from its documentation: UNPINNED against cv2 (absent), see oracle/aug_oracle.py. */
int amx_aug_resample(const float* x, float* y, const int* win, int N, int Hs, int Ws, int Hd, int Wd, int mode,
                     int clip01, int round_out, void* stream);
int amx_aug_onehot(const long long* t, float* masks, int N, int K, long HW, void* stream);
int amx_aug_squeeze(const float* masks, long long* out, int* values, int N, int K, long HW, void* stream);

/* ---- dense layers (nn.Linear + Tanh / ReLU of fcEncoderNet / fcDecoderNet / the convEncoderNet heads,
 * atomai/nets/ed.py:231-343, 530-580, and of fcFeatureExtractor, atomai/nets/gp.py:14-26): one fp32-MFMA GEMM
 * C[M][N] = act(A * B + bias) with operand strides — element (m, k) of A at A[m*sam + k*sak], (k, n) of B at
 * B[k*sbk + n*sbn], (m, n) of C at C[m*scm + n] — so that forward (x W^T), data gradient (dpre W) and weight
 * gradient (dpre^T x) are the same entry point.  act: 0 none, 1 tanh, 2 relu (applied after the bias).
 * amx_act_bwd: dpre = dy * act'(y) from the layer OUTPUT y (tanh: 1 - y^2, relu: y > 0). */
int amx_gemm_f32(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long scm,
                 const float* bias, int M, int N, int K, int act, void* stream);
/* the same with the k range cut into `splits` slices (amx_gemm_f32_splits(M, N, K): 1 = not worth it) whose raw partial
 * sums go to work[splits][M][N] and are added in slice order before bias / activation — deterministic; for long-K layers
 * with few output tiles (the 512 x 128 x 4096 first encoder layer of the VAEs: 64 workgroups of 256 serial k steps). */
int amx_gemm_f32_splits(int M, int N, int K);
int amx_gemm_f32_splitk(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long scm,
                        const float* bias, int M, int N, int K, int act, float* work, int splits, void* stream);
int amx_act_bwd(const float* dy, const float* y, float* out, long n, int act, void* stream);

/* ---- rVAE spatial decoder (rDecoderNet + coord_latent, atomai/nets/ed.py:583-687): per-pixel MLP with all
 * hidden activations kept in LDS; backward recomputes per tile and emits per-sample partial rows.
 * Coordinates: theta == NULL -> coords is the explicit [B][n][2] tensor the reference's forward takes (gradient
 * in dcoords); theta = [B][3] (phi, dx, dy) -> coords is the shared grid [n][2] (imcoordgrid,
 * atomai/utils/coords.py:37-54) and transform_coordinates (:57-83: rotation by phi, then translation) is applied per
 * pixel in the kernel, its gradient reduced per sample into dtheta [B][3] — no (B, n, 2) tensor exists in either
 * direction.  C output channels (<= 4): xrec / dxrec are [B][n][C] (channel-last, as ed.py:639-642 reshapes),
 * Wo [C][hid], bo [C], pWo [B][C][hid], pbo [B][C].  hid in {32, 64, 128}, 1 <= NL <= 5. */
int amx_rdecoder_fwd(const float* coords, const float* theta, const float* z, const float* Wc, const float* bc,
                     const float* Wz, const float* W, const float* b, const float* Wo, const float* bo, float* xrec,
                     int B, int n, int L, int hid, int NL, int skip, int C, void* stream);
int amx_rdecoder_bwd(const float* coords, const float* theta, const float* z, const float* Wc, const float* bc,
                     const float* Wz, const float* W, const float* Wt, const float* b, const float* Wo,
                     const float* bo, const float* dxrec, float* dcoords, float* dtheta, float* dz, float* pW,
                     float* pb, float* pWo, float* pbo, float* pWc, float* pbc, float* pWz, int B, int n, int L,
                     int hid, int NL, int skip, int C, void* stream);
/* Saved-activation variant of the pair (training): the forward additionally writes the NL post-activation hidden images
 * to hsave ([B][NL][hid/4][npad][4] floats, npad = n rounded up to 128; size from amx_rdecoder_hsave_floats) and the
 * backward reads them back one tile ahead instead of recomputing the hidden layers (1/3 of its MFMA work and all of its
 * tanh epilogues) — bit-identical gradients.  hsave / hsaved == NULL: exactly the functions above. */
long amx_rdecoder_hsave_floats(int B, int n, int hid, int NL);
int amx_rdecoder_fwd_save(const float* coords, const float* theta, const float* z, const float* Wc, const float* bc,
                          const float* Wz, const float* W, const float* b, const float* Wo, const float* bo,
                          float* xrec, float* hsave, int B, int n, int L, int hid, int NL, int skip, int C,
                          void* stream);
int amx_rdecoder_bwd_saved(const float* coords, const float* theta, const float* z, const float* Wc, const float* bc,
                           const float* Wz, const float* W, const float* Wt, const float* b, const float* Wo,
                           const float* bo, const float* dxrec, const float* hsaved, float* dcoords, float* dtheta,
                           float* dz, float* pW, float* pb, float* pWo, float* pbo, float* pWc, float* pbc, float* pWz,
                           int B, int n, int L, int hid, int NL, int skip, int C, void* stream);

/* ---- ELBO terms of vae_loss / rvae_loss (atomai/losses_metrics/vi_losses.py:13-137), fwd and bwd.
 * recon_kind 0 = 'mse' (0.5 * sum (xrec - x)^2, vi_losses.py:23-26; recon_scale unused), 1 = 'ce' (sum of
 * binary_cross_entropy_with_logits(xrec, x), vi_losses.py:27-34) times recon_scale: 1 for a 2-D in_dim, 1 / (H*W) for a
 * 3-D one, where the reference sums over the channels only and its .mean() runs over samples x pixels. */
int amx_elbo_terms_fwd(const float* x, const float* xrec, const float* zmean, const float* zlogsd, int B, int n,
                       int Z, int rot, float phi_prior, int recon_kind, float recon_scale, float* recon, float* klz,
                       float* klrot, void* stream);
int amx_elbo_terms_bwd(const float* x, const float* xrec, const float* zmean, const float* zlogsd,
                       const float* g_recon, const float* g_klz, const float* g_klrot, int B, int n, int Z,
                       int rot, float phi_prior, int recon_kind, float recon_scale, float* dxrec, float* dmean,
                       float* dlogsd, void* stream);
/* Scalar ELBO without a capacity term (vi_losses.py:105-108, 129-137): out[0] = -mean(recon) - mean(klz) - mean(klrot)
 * (klrot may be NULL), fixed-order fp64 sum; amx_elbo_bwd_scalar = amx_elbo_terms_bwd with all three per-sample upstream
 * gradients equal to coef * gscalar[0] (gscalar: the device scalar autograd hands to the loss; coef = -1 / B). */
int amx_elbo_combine(const float* recon, const float* klz, const float* klrot, int B, float* out, void* stream);
int amx_elbo_bwd_scalar(const float* x, const float* xrec, const float* zmean, const float* zlogsd, const float* gscalar,
                        float coef, int B, int n, int Z, int rot, float phi_prior, int recon_kind, float recon_scale,
                        float* dxrec, float* dmean, float* dlogsd, void* stream);
/* rVAE latent plumbing (models/dgm/rvae.py:118-137) in one pass each way: z = mean + exp(logsd) * eps; theta [B][3] =
 * (z0, z1 * dx_prior, z2 * dx_prior) (translation) or (z0, 0, 0); zc [B][Z - 3 | Z - 1] = the content latents. */
int amx_rvae_latent_fwd(const float* zmean, const float* zlogsd, const float* eps, int B, int Z, int translation,
                        float dx_prior, float* theta, float* zc, void* stream);
int amx_rvae_latent_bwd(const float* zlogsd, const float* eps, const float* dtheta, const float* dzc, int B, int Z,
                        int translation, float dx_prior, float* dmean, float* dlogsd, void* stream);


/* The rDecoder kernels' activation function, element-wise: y[i] = tanh(x[i]) as the fused kernels evaluate it
 * (hardware exp + reciprocal, absolute error <= 2e-7; nn.Tanh of atomai/nets/ed.py:613-616).  For tests of that bound. */
int amx_rdec_tanh_probe(const float* x, float* y, long n, void* stream);

/* ---- DKL covariance evaluation: ScaleKernel(RBFKernel(ard)) / MaternKernel(2.5) on the embeddings
 * (selected at atomai/nets/gp.py:41-46,95-106; arithmetic in gpytorch).  kind 0 = RBF, 1 = Matern-5/2;
 * inv_ls = 1/lengthscale per dim; is_double selects fp64 buffers. */
int amx_kernel_matrix(const void* X1, const void* X2, const void* inv_ls, double outputscale, int kind,
                      double noise, int N, int M, int D, int is_double, void* K, void* stream);
int amx_kernel_matvec(const void* X1, const void* X2, const void* inv_ls, double outputscale, int kind, int N,
                      int M, int D, int R, int is_double, const void* V, void* Y, void* stream);
int amx_kernel_matrix_bwd(const void* X, const void* inv_ls, double outputscale, int kind, int N, int D,
                          int is_double, const void* G, void* dX, void* part, void* stream);
/* amx_kernel_matrix_bwd with G = (alpha alpha^T - Kinv) * gscale formed while Kinv is read (Kinv N x N symmetric,
 * alpha N): the gradient of the exact marginal log likelihood that gpytorch's ExactMarginalLogLikelihood gives the
 * reference's optimizer (atomai/trainers/gptrainer.py:126-137, 285-303), without the N x N temporaries. */
int amx_kernel_matrix_bwd_mll(const void* X, const void* inv_ls, double outputscale, int kind, int N, int D,
                              int is_double, const void* Kinv, const void* alpha, double gscale, void* dX, void* part,
                              void* stream);

/* ---- KISS-GP (structured kernel interpolation): the reference's covariance module is
 * gpytorch.kernels.GridInterpolationKernel(base_kernel, num_dims=embedim, grid_size=50) (atomai/nets/gp.py:41-46), trained
 * through ExactMarginalLogLikelihood (atomai/trainers/gptrainer.py:126-137, 303) and predicted with fast_pred_var
 * (atomai/models/dklgp/dklgpr.py:146-156):  K(X, X') = W(X) K_UU W(X')^T with K_UU = amx_kernel_matrix on a regular grid
 * of G nodes per embedding dimension and W the cubic-convolution interpolation weights (4 nodes per dimension).  D = 1 or
 * 2; node index = i0 * G + i1; `base` [N][D] = first stencil node per dimension, `w` / `dw` [N][D][4] = weights and their
 * derivatives w.r.t. the coordinate; C <= 8 right-hand sides; is_double selects fp64 buffers.  No float atomics: sums run
 * in a fixed order over `order` (point indices sorted by cell = base[.][0] * (G - 3) + base[.][1]) and `cell_start`
 * [(G-3)^D + 1] (first position of each cell in `order`). */
int amx_ski_weights(const void* Z, const void* g0, const void* inv_delta, int N, int D, int G, int is_double, int* base,
                    void* w, void* dw, int* cell, void* stream);      /* cell [N] (or NULL): the point's grid cell, the sort key */
/* elements (of the value type) of amx_ski_gram's workspace `ws`, or -1 */
long amx_ski_gram_workspace(int D, int G, int C);
/* A [m][m] = W^T W (or NULL; ZERO on entry: only the band |u_d - v_d| <= 3 is written) and b [C][m] = W^T r for r [C][N],
 * m = G^D */
int amx_ski_gram(const void* w, const void* r, const int* order, const int* cell_start, int N, int D, int G, int C,
                 int is_double, void* ws, void* A, void* b, void* stream);
/* backward of amx_ski_gram: GA = dL/dA [m][m] (symmetric), gb = dL/db [C][m]  ->  dZ [N][D], dr [C][N] */
int amx_ski_gram_bwd(const int* base, const void* w, const void* dw, const void* r, const void* GA, const void* gb, int N,
                     int D, int G, int C, int is_double, void* dZ, void* dr, void* stream);
/* Y [C][N] = W V^T for node vectors V [C][m] (predictive means) */
int amx_ski_interp(const int* base, const void* w, const void* V, int N, int D, int G, int C, int is_double, void* Y,
                   void* stream);
/* out [N1][N2] = scale * W1 Q W2^T for Q [m][m]; diag != 0: out [N1] = its diagonal (N1 == N2) */
int amx_ski_cov(const int* base1, const void* w1, int N1, const int* base2, const void* w2, int N2, const void* Q, int D,
                int G, double scale, int diag, int is_double, void* out, void* stream);

/* ---- Locator: thresholded class maps -> blob centres (atomai/predictors/predictor.py:531-639 Locator.run /
 *      rem_edge_coord; atomai/utils/img.py:554-564 cv_thresh; atomai/utils/coords.py:21-34 find_com =
 *      scipy.ndimage.label + center_of_mass).  prob: (B,H,W,C) fp32 NHWC probabilities; the first `nch`
 *      channels are located (the reference skips the last = background class).  amx_locate_label labels the
 *      4-connected components of prob > thr, accumulates the centres and writes the number of centres that
 *      survive the dist_edge filter to *count (device int).  amx_locate_emit then writes them, ordered by
 *      (frame, class, first pixel in raster order) = the reference's dictionary order: coords (cap,2) fp64
 *      (row, col), meta (cap,2) int32 (frame, class).  Workspace: amx_locate_workspace_bytes (negative when
 *      B*nch*H*W does not fit int32 labels: chunk the stack). */
long amx_locate_workspace_bytes(int B, int H, int W, int nch);
int amx_locate_label(const float* prob, int B, int H, int W, int C, int nch, float thr, int dist_edge, void* work,
                     int* count, void* stream);
int amx_locate_emit(const void* work, int B, int H, int W, int nch, int dist_edge, double* coords, int* meta,
                    long cap, void* stream);

/* ---- torch.optim.Adam defaults as one flat launch (trainer.py:539, vitrainer.py:218).  b1, b2 and the bias
 * corrections bc1 = 1-b1^t, bc2 = 1-b2^t are DOUBLES: torch forms 1-beta in double before rounding to fp32
 * (1 - float(0.999) is off by 1.3e-5 relative, a systematic error in every second moment). */
int amx_adam_flat(float* p, const float* g, float* m, float* v, long n, float lr, double b1, double b2,
                  float eps, double bc1, double bc2, float gscale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
