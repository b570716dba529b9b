/*
 * libpfa_hip.so - C ABI of the MI355X (gfx950) kernels behind the Pix2Pix GAN step.
 *
 * The reference (Affi-Amine/GAN-AUG-PFA) has no FFI layer: its hot path calls PyTorch
 * operators from Python (models.py:177-247, train_gan.py:42-71).  This header is the
 * boundary a maintainer binds instead: one entry point per operator
 * family, raw device pointers + sizes + a hipStream_t, no torch types.
 *
 * Conventions
 *  - Every function returns 0 on success, a negative value for an argument / shape error
 *    detected before launch, or a positive hipError_t.  pfa_last_error_string() gives the
 *    message (thread local).  No exception crosses the ABI.
 *  - Launch-only: no hipMalloc / hipFree / synchronisation inside; every buffer (workspaces
 *    included) is owned by the caller and must outlive the stream work.  All launches are
 *    ordered on the stream argument and are hipGraph-capturable.
 *  - Activations are NHWC ("channels last"), element type pfa_dtype, channel counts padded
 *    to PFA_CH(dtype) = 16 bytes / sizeof(element) (8 for bf16, 4 for f32); pad lanes hold 0.
 *  - Convolution weights for the kernels ("shadows") are [rows][16 taps][channels] with the
 *    contraction channel contiguous; tap = ky*4 + kx.  Master parameters stay float32 in the
 *    reference's OIHW / IOHW logical shape (channels-last memory), see pfa_weight_shadow.
 */
#ifndef PFA_HIP_H
#define PFA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pfa_stream;                 /* hipStream_t */

enum pfa_dtype { PFA_F32 = 0, PFA_BF16 = 1 };
#define PFA_CH(dtype) ((dtype) == PFA_BF16 ? 8 : 4)

enum pfa_act { PFA_ACT_NONE = 0, PFA_ACT_LRELU = 1, PFA_ACT_RELU = 2, PFA_ACT_TANH = 3 };

/* Geometry of an implicit-GEMM gather (no im2col buffer is ever materialised). */
enum pfa_gather_mode {
    PFA_CONV_S2 = 0,        /* nn.Conv2d(k4,s2,p1) forward (models.py:177,223,230); also ConvTranspose2d dgrad */
    PFA_CONV_S1 = 1,        /* nn.Conv2d(k4,s1,p1) forward (models.py:238,243)                                   */
    PFA_TCONV_S2 = 2,       /* nn.ConvTranspose2d(k4,s2,p1) forward (models.py:184,189,194) as 4 output-parity   */
                            /* 2x2-tap convolutions; also the dgrad of the stride-2 Conv2d                      */
    PFA_CONV_S1_DGRAD = 3   /* dgrad of the stride-1 Conv2d                                                      */
};

int pfa_version(void);
const char* pfa_last_error_string(void);
/* Number of kernel launches issued by this process through the library (tests use it to
 * prove the HIP path, not a fallback, produced a result). */
uint64_t pfa_launch_count(void);
/* Developer tuning knobs for A/B measurements (never needed for correctness). id 0: gather tile (0 = automatic,
 * 4 = 256x128 two-stage / 8 = 256x128 ring where n >= 128, 6 = 128x64 everywhere, 7 = no ring tiles); id 1: in-kernel ablation bits (1 no stores, 2 no statistics, 4 no MFMA, 8 no DMA,
 * 32 ring: a stage's DMA in one burst; 64 fused backward epilogue without its raw / g2 traffic; 128 staggered ring: no LDS fragment reads
 * (bits 4, 8 and 128 apply to the two-stage loops and the rings that honour them - results are wrong when set); 1024 the slice-combining passes
 * (k_splitk_finish_vec, pfa_finish_bn_*) wait one round trip per slice, as before round 3; 4096 no four-image tiles for 8 x 8 maps
 * on the stride-2 / transposed patch kernels (they go back to the implicit GEMM's ring); 16384 the 256-row bf16 launches of the implicit GEMM on the
 * row kernel of conv_rows.hip (the patch kernels' K loop over gathered rows: built and tested, measured slower inside the step)); 32768 the general weight-gradient kernel stages row-aligned
 * geometries with per-lane offsets fixed for the loop (k_wgrad<...,rows>: same bits, faster per layer, no gain in the step);
 * id 2: gather K-loop (0 = default per tile, 1 = fragment reads before the next tile's DMA, 2 = DMA first, 3 = DMA spread
 * over the MFMAs); id 4: weight-gradient K-loop (0 = default: K-group pairs - two 4-wave groups per output tile -
 * from two reduction slices on, 1 = reads first, 2 = DMA first, 3 = 256-row 3-stage ring tile for >= 256 direct channels,
 * 5 = no K-group pairs, 6 = pairs from four slices on, 7 = never the patch kernel of wgrad_patch.hip, 8 = the patch kernel
 * whenever the geometry fits);
 * id 3: the input-patch kernels (conv_patch.hip: stride-1 convolutions, forward and input gradient; conv_patch2.hip: stride-2
 * convolutions and transposed convolutions): 0 = launches large enough to fill the chip, 1 = never, 2 = whenever the geometry
 * fits, 3 = the stride-1 kernel only, 4 = no four-class transposed workgroups (conv_tconv4.hip), 5 = patch kernels and
 * four-class workgroups whenever the geometry fits, 6 / 8 = the stride-1 kernel's earlier forms (two-phase loop on MFMA
 * 16x16x32 / 32x32x16), 7 = 64-channel tiles of the stride-1 kernel whenever the geometry fits (tests; by default they are taken where 128-channel tiles
 * would need K slices to fill the chip), 9 = never (the K-sliced plan of round 3);
 * id 5: gather work-list order over the XCDs (0 = byte-count heuristic, 1..3 = force order 0..2); id 6: minimum number
 * of 256-row tiles for the ring tile, and of workgroups a stride-2 patch launch wants before it narrows its channel tile
 * or slices K (0 = defaults); id 7: 2 = also use the 256x64 ring for 16 < n <= 64, 4 = half-step software-pipelined
 * ring on the 256x128 tile, 5 = ring on the 128x64 tile, 6 / 7 = staggered ring on the 256x128 tile (the two waves of a SIMD
 * in opposite read / MFMA phases; one phase pair per K-step / per 32-wide half), 11..17 = diagnostic instantiations of the
 * stride-2 patch kernels (conv and weight gradient) without MFMAs (bit 1), fragment reads (2), DMA (4), 18 / 19 = the conv kernel without its
 * counted vmcnt wait / without that wait and the step's barrier: results are wrong;
 * id 8: workgroups a weight-gradient patch launch wants (0 = 128); id 9: workgroup count from which transposed
 * convolutions take four-class workgroups, id 10: 128-channel tile count below which stride-1 convolutions slice K (0 = 128); id 11: Adam grid (0 = default);
 * id 12: reduction slices the image-side weight gradient (wgrad_image.hip; off with id 4 = 7 | 9) wants (0 = 256);
 * id 13: two tap rows per workgroup in the patch weight gradient (wgrad_patch.hip): 2 = always, 3 = where its ring keeps
 * four stages (stride 1, 4x16 tile); id 14: partial rows up to which the BatchNorm coefficient launch is folded into the pass
 * that applies it (pfa_bn_act_stats / pfa_bn_bwd_apply_stats; 0 = 512, -1 = never); id 15: 1 = the launches that follow run
 * alone on the GPU (set and cleared by the engines around such passes, recorded in launch lists): plans that leave CUs to
 * other streams - the un-sliced stride-1 input gradient, 128-channel patch tiles on 128 workgroups - fill the chip instead. */
int pfa_set_knob(int id, int value);
/* Developer timing: device memory (caller-owned; NULL = off) into which the stride-2 patch kernels write clock stamps per
 * workgroup: conv (uint64[workgroups][16]: 8 shader-clock stamps - entry, first fragments, K loop done, accumulators in LDS,
 * rows issued, statistics written, epilogue done, stores accepted - and 2 of the 100 MHz real-time counter) and weight
 * gradient (uint64[workgroups][8]).  Never changes a result; off by default. */
int pfa_set_debug_buffer(void* dev, size_t bytes);
/* Kernel instantiation the calling thread's most recent pfa_gather_gemm launched, e.g. "k_gather_gemm<bf16,256,128,4,2,3>"
 * (tile rows, tile columns, wave grid, K-loop variant): lets a profiler script attribute a launch to a kernel name. */
const char* pfa_last_gather_kernel(void);
/* Per-launch timing for measurements (bench.py's roofline object): while enabled, every pfa_gather_gemm and pfa_wgrad
 * call records one HIP event on its launch stream right before and one right after its MFMA kernel (the finish kernel of a
 * split-K call and the reduce kernel of a weight gradient stay outside).  pfa_gather_timing(1) clears the store and
 * starts, (0) stops; after synchronising the stream, pfa_gather_timing_get(i, ...) returns launch i's kernel time in ms,
 * its instantiation name, its algorithmic FLOPs (2 x the MACs of the convolution on REAL channels and real taps: the
 * count of SURVEY 8d), its unique operand + result bytes and its K-slice (reduction-slice) count.  At most 65536
 * launches are kept per enable. */
int pfa_gather_timing(int enable);
int pfa_gather_timing_count(void);
int pfa_gather_timing_get(int i, float* ms, const char** name, double* flop, double* bytes, int* splitk);
/* Start of timed launch i relative to the start of timed launch `ref`, in ms (launches of different streams compare: the
 * events are device timestamps) - an in-situ timeline of the MFMA kernels of a multi-stream step (tools/insitu_gantt.py). */
int pfa_gather_timing_start(int i, int ref, float* start_ms);
/* workgroups of timed launch i where the library recorded them (the stride-1 patch kernel), else 0 */
int pfa_gather_timing_grid(int i, int* nwg);

/* ------------------------------------------------------------------ implicit GEMM, MFMA
 * out[pix(m)][n] = act( bias[n] + sum_{tap,c} src[pix(m)+tap][c] * w[n][tap][c] )
 * Replaces: Conv2d / ConvTranspose2d forward and their input gradients.
 * rows m enumerate the position grid [nb][hq][wq]; for PFA_TCONV_S2 the grid is the SOURCE
 * grid and each of the 4 parity classes writes output pixel (2qy+py, 2qx+px).
 * The source may be a virtual concat [src1 (c1 ch) | src2 (c2 ch)] (torch.cat of models.py:208
 * is never materialised); the output columns may be split [out1 (n1 cols) | out2].
 */
typedef struct pfa_gather_args {
    int32_t struct_size;
    int32_t mode, dtype;
    int32_t nb, hq, wq;          /* position grid                                   */
    int32_t hs, ws;              /* source spatial size                             */
    int32_t c1, c2;              /* padded source channels, multiples of PFA_CH     */
    int32_t n;                   /* GEMM N = rows of w that exist                   */
    int32_t n1;                  /* columns [0,n1) -> out1, [n1,..) -> out2         */
    int32_t ld1, ld2;            /* channel stride (padded channels) of out1 / out2 */
    int32_t act;                 /* pfa_act applied after bias                      */
    float slope;                 /* LeakyReLU slope                                 */
    int32_t splitk;              /* 1, or number of K slices (partials -> slab)     */
    int32_t ntiles_stat;         /* out: rows of psum/psq written (may be NULL use) */
    const void* src1;
    const void* src2;
    const void* w;               /* [n][16][c1+c2]                                  */
    const float* bias;           /* [n] or NULL                                     */
    void* out1;
    void* out2;
    float* psum;                 /* optional BatchNorm partial sums   [tiles][n]    */
    float* psq;                  /* optional partial sums of squares  [tiles][n]    */
    float* slab;                 /* splitk>1: float32 workspace of pfa_gather_slab_floats() elements */
    /* fuse_out == 3, eval-mode BatchNorm folded into a forward launch (generate_synthetic_data.py:55; K-sliced launches
     * apply it in their finish pass, which needs n % 4 == 0, n/4 dividing 256 and ld1 == n; no
     * bias, n1 >= n): y = fuse_scale[n]*acc + fuse_shift[n]; out1 = act(y); out2, if not NULL, = fuse_act2(y) over the
     * same columns (the LeakyReLU and ReLU readers of one encoder activation, models.py:204-208). */
    /* Fused activation / BatchNorm backward (input-gradient launches, no bias/act; a K-sliced launch applies it in its
     * finish pass, which needs n % 4 == 0, n/4 dividing 256 and, with one output, ld1 == n): the chosen output
     * receives dy = act1'(y)*g + act2'(y)*g2 instead of g, with y = scale*raw + shift (scale NULL: y = raw) and raw the
     * conv output that output is the gradient of (same shape).  With mean/invstd, psum/psq receive the per-tile
     * sums of dy and dy*xhat over the fused output's columns (row length = its ld) for pfa_bn_bwd_finalize. */
    int32_t fuse_out;            /* 0 = off, 1 = out1, 2 = out2, 3 = affine epilogue (below), 4 = PFA_FUSE_DEFER */
    int32_t fuse_act1, fuse_act2;
    int32_t c_real;              /* real (unpadded) contraction channels of src1|src2 for the FLOP accounting of      */
                                 /* pfa_gather_timing only; 0 = c1 + c2.  Never changes a result.                     */
    const void* fuse_raw;
    const void* fuse_g2;         /* optional second gradient w.r.t. the same tensor              */
    const float* fuse_scale;
    const float* fuse_shift;
    const float* fuse_mean;
    const float* fuse_invstd;
} pfa_gather_args;

/* Tiles of partial statistics a call will write (rows of psum/psq) for this geometry. */
int pfa_gather_stat_tiles(const pfa_gather_args* a);
/* Floats the slab of this call must hold (0 when splitk <= 1). */
size_t pfa_gather_slab_floats(const pfa_gather_args* a);
int pfa_gather_gemm(pfa_gather_args* a, pfa_stream s);

/* Sum split-K slabs [splitk][pixels][slab_ld], add bias, emit statistics partials, activate, store. */
int pfa_splitk_finish(const float* slab, int slab_ld, int splitk, int pixels, int n, int n1, const float* bias,
                      int act, float slope, int dtype, void* out1, int ld1, void* out2, int ld2, float* psum,
                      float* psq, int* ntiles_stat, pfa_stream s);

/* ------------------------------------------------------------------ weight gradient, MFMA
 * dw[i][tap][c] (+)= sum_m direct[m][i] * gathered[pix(m)+tap][c]     (float32 accumulate)
 * Replaces: autograd's weight gradient of Conv2d (direct = dOut, gathered = input) and of
 * ConvTranspose2d (direct = input incl. the virtual concat, gathered = dOut).
 * Gathered pixel = (qy*stride - 1 + ky, qx*stride - 1 + kx).  dw is laid out like the
 * float32 master parameter: [i_real][16][c_real] (pad channels of either side are skipped).
 */
typedef struct pfa_wgrad_args {
    int32_t struct_size;
    int32_t dtype, stride;
    int32_t nb, hq, wq;          /* direct operand grid (reduction rows)                       */
    int32_t hg, wg;              /* gathered operand spatial size                              */
    int32_t d1p, d1r, d2p, d2r;  /* direct channels: padded / real, for src1 and src2 (d2*=0)  */
    int32_t g1p, g1r, g2p, g2r;  /* gathered channels likewise                                 */
    int32_t msplit;              /* number of slices of the reduction (>=1)                    */
    int32_t fresh;               /* nonzero: dw holds zeros (a fresh gradient buffer, written once): a launch with  */
                                 /* ONE reduction slice then stores its tile instead of adding to it with atomics */
    const void* dir1;
    const void* dir2;
    const void* gat1;
    const void* gat2;
    float* dw;                   /* [d1r+d2r][16][g1r+g2r]; the result is ADDED to it (zero_grad semantics)   */
    float* slab;                 /* optional workspace of pfa_wgrad_slab_floats(a) floats: reduction slices are  */
                                 /* combined in a fixed order (bitwise reproducible) instead of float atomics   */
} pfa_wgrad_args;
size_t pfa_wgrad_slab_floats(const pfa_wgrad_args* a);
/* Reduction slices the call will really use (the library may take another kernel, with its own slice count, than the one
 * a->msplit was chosen for): callers pass a slab when many slices would otherwise meet in float atomics. */
int pfa_wgrad_slices(const pfa_wgrad_args* a);
int pfa_wgrad(const pfa_wgrad_args* a, pfa_stream s);

/* ------------------------------------------------------------------ the PatchGAN logit layer
 * Conv2d(512, 1, kernel 4, stride 1, padding 1) + bias: the discriminator's last layer (models.py:243), bf16.
 * One output channel makes the layer bandwidth-shaped; these three vector kernels replace its launches on the MFMA
 * kernels (pfa_gather_gemm CONV_S1 / CONV_S1_DGRAD + pfa_bn_bwd_reduce, pfa_wgrad + pfa_colsum).
 *   x        [nb][h][w][c] activations entering the layer; logits / dlogits [nb][h-1][w-1][ld], value in channel 0
 *   w        [16 taps][c] bf16: the layer's weights, tap = ky*4+kx (the engine's "same" shadow of the [1][c][4][4] master)
 * pfa_logit_supported: 1 when the kernels are built for this shape (else callers keep the general kernels). */
int pfa_logit_supported(int dtype, int c, int ld, int h, int w);
/* logits[.., 0] = bias[0] + sum_{tap,c} x[oy-1+ky][ox-1+kx][c] * w[tap][c]; channels 1..ld-1 are written as zeros. */
int pfa_logit_fwd(const void* x, const void* w, const float* bias, void* logits, int nb, int h, int wd, int c, int ld,
                  pfa_stream s);
/* pfa_logit_fwd with nn.BCEWithLogitsLoss (train_gan.py:42,58,60,67: mean reduction, constant target) in its epilogue -
 * what pfa_bce_logits_parts computes from the stored logits, without the launch in between: dlogits[.., 0] = gscale *
 * (sigmoid(x) - target) / count (pad channels zero), parts[workgroup] = lscale * (sum of the workgroup's loss terms) / count
 * for pfa_loss_finish; pfa_logit_fwd_nparts(nb, h, wd) partials are written. */
int pfa_logit_fwd_nparts(int nb, int h, int wd);
int pfa_logit_fwd_bce(const void* x, const void* w, const float* bias, void* logits, int nb, int h, int wd, int c, int ld,
                      float target, float lscale, float gscale, float* parts, void* dlogits, pfa_stream s);
/* g[y][x][c] = sum_tap dlogits[y+1-ky][x+1-kx] * w[tap][c], then the backward of what feeds the layer, as
 * pfa_bn_bwd_reduce computes it: dy = act'(scale*raw + shift) * g (scale == NULL: identity map), and when mean != NULL
 * the partial sums of dy and dy*xhat -> part[2][pfa_logit_dgrad_tiles(nb,h,wd)][c] for pfa_bn_bwd_finalize. */
int pfa_logit_dgrad_tiles(int nb, int h, int wd);
int pfa_logit_dgrad(const void* dlogits, int ld, const void* w, const void* raw, const float* scale, const float* shift,
                    const float* mean, const float* invstd, int act, float slope, void* dy, float* part, int nb, int h,
                    int wd, int c, pfa_stream s);
/* dw[tap][c] += sum_pixels dlogits * x[pixel + tap][c], db[0] += sum dlogits (db may be NULL).  slab: caller-owned scratch
 * of pfa_logit_wgrad_slab_floats(nb, h) floats - the workgroups' partial sums go there and a second launch adds them in a
 * fixed order (bitwise reproducible); NULL: float atomics into dw / db, whose last bits vary from run to run. */
size_t pfa_logit_wgrad_slab_floats(int nb, int h);
int pfa_logit_wgrad(const void* dlogits, int ld, const void* x, float* dw, float* db, int nb, int h, int wd, int c,
                    float* slab, pfa_stream s);

/* ------------------------------------------------------------------ transposed conv onto <= 8 channels
 * out[b][2*iy-1+ky][2*ix-1+kx][co] = act( bias[co] + sum_c src[b][iy][ix][c] * w[co][ky*4+kx][c] )
 * Replaces, where the output has only a few channels: the generator's outermost
 * ConvTranspose2d(2*ngf, 3, 4, 2, 1) + Tanh (models.py:184-186) and the input gradient of the
 * discriminator's first Conv2d(6, ndf, 4, 2, 1) (models.py:223, needed by train_gan.py:66-70).
 * Same result as pfa_gather_gemm(PFA_TCONV_S2) with the same operands; the input is staged once
 * per patch (pointwise MFMA GEMM + col2im in LDS) instead of once per tap and parity class.
 * src1|src2: NHWC [nb][hs][ws][c1|c2] virtual concat, c1 and c2 multiples of one 128-byte chunk
 * (64 bf16 / 32 f32 channels; c2 may be 0); w: [n][16][c1+c2] (rows >= n are never read);
 * out: NHWC [nb][2hs][2ws][ld], lanes [n, ld) written as zeros.  pfa_tconv_scatter_supported
 * returns 1 when these constraints hold (callers fall back to pfa_gather_gemm otherwise).
 */
typedef struct pfa_scatter_args {
    int32_t struct_size;
    int32_t dtype;
    int32_t nb, hs, ws;
    int32_t c1, c2;
    int32_t n, ld;
    int32_t act;
    float slope;
    int32_t relu_src1;           /* != 0: source 1 enters through ReLU (max(x, 0) on the staged operand): the eval forward keeps ONE copy of the
                                  * outermost encoder activation, the LeakyReLU one, and relu(lrelu(y)) == relu(y) (models.py:180,185) */
    const void* src1;
    const void* src2;
    const void* w;
    const float* bias;           /* [n] or NULL */
    void* out;
    /* Optional fused epilogue of the generator phase (train_gan.py:66-70): the launch computes dL/dfake of the discriminator's
     * first layer (this transposed form, no bias / activation), and with l1_fake != NULL `out` receives, instead of it,
     * the gradient w.r.t. the generator's pre-Tanh output
     *   dpre = (l1_lambda * sign(fake - real) / (pixels * l1_creal) + stored(dL/dfake)) * (1 - fake^2)   for channels < l1_creal, 0 beyond
     * - what pfa_l1_tanh_bwd computes from the stored dL/dfake (same bits: the value is rounded to the storage type first).
     * l1_fake / l1_real: [nb][2 hs][2 ws][ld], the output's geometry; l1_lambda = LAMBDA_L1 (train_gan.py:33).  The L1 loss
     * value itself is not produced here (pfa_l1_tanh_bwd_parts with dpre == NULL, off the critical path). */
    const void* l1_fake;
    const void* l1_real;
    float l1_lambda;
    int32_t l1_creal;
} pfa_scatter_args;
int pfa_tconv_scatter_supported(int dtype, int c1, int c2, int n, int ld);
int pfa_tconv_scatter(const pfa_scatter_args* a, pfa_stream s);

/* ------------------------------------------------------------------ BatchNorm2d (training)
 * Replaces nn.BatchNorm2d (models.py:179,181,231,239): eps 1e-5, momentum 0.1, biased batch
 * variance for normalisation, unbiased for running_var.  Statistics arrive as per-tile
 * partial sums written by the GEMM epilogue; `repeats` applies the running-buffer update that
 * many times (the reference runs the generator forward twice per step, train_gan.py:56,65).
 */
int pfa_bn_finalize(const float* psum, const float* psq, int ntiles, int c, double count, const float* gamma,
                    const float* beta, float eps, float momentum, int repeats, float* running_mean,
                    float* running_var, int64_t* num_batches_tracked, float* scale, float* shift, float* mean,
                    float* invstd, pfa_stream s);
/* Eval mode (generate_synthetic_data.py:55): scale/shift from the running buffers. */
int pfa_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, int c, float* scale, float* shift, pfa_stream s);
/* y = scale*raw + shift (identity when scale==NULL); out_l = LeakyReLU(y), out_r = ReLU(y); either may be NULL. */
int pfa_bn_act(const void* raw, int64_t rows, int c, int dtype, const float* scale, const float* shift, float slope,
               void* out_l, void* out_r, pfa_stream s);
/* pfa_bn_finalize + pfa_bn_act in ONE launch: every workgroup sums the per-tile partials of its own 64-channel group in
 * its prologue (the same fixed order in every workgroup: the same bits, no hand-off between workgroups), normalises and
 * activates its rows; the workgroups of row block 0 write scale / shift / mean / invstd (read by the backward pass) and
 * update the running buffers `repeats` times.  Built for pfa_bn_stats_fused_supported(dtype, c, ntiles) != 0 (c a multiple
 * of 64, or a smaller power-of-two chunk count; at most 512 partial rows: pfa_set_knob(14, limit | -1 = never)); other
 * shapes keep the two launches.  raw is [rows][c] (unpadded rows: BatchNorm layers have c % 16 bytes == 0). */
int pfa_bn_stats_fused_supported(int dtype, int c, int ntiles);
int pfa_bn_act_stats(const void* raw, int64_t rows, int c, int dtype, const float* psum, const float* psq, int ntiles,
                     double count, const float* gamma, const float* beta, float eps, float momentum, int repeats,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                     float* mean, float* invstd, float slope, void* out_l, void* out_r, pfa_stream s);
/* pfa_bn_bwd_finalize + pfa_bn_bwd_apply in ONE launch, same scheme: dr = BatchNorm backward of dy (in place allowed),
 * dgamma / dbeta += the channel sums (row block 0). */
int pfa_bn_bwd_apply_stats(const void* dy, const void* raw, int64_t rows, int c, int dtype, const float* part, int ntiles,
                           double count, const float* gamma, const float* mean, const float* invstd, float* dgamma,
                           float* dbeta, void* dr, pfa_stream s);
/* dy = act1'(y)*g1 (+ act2'(y)*g2), y = scale*raw+shift; partial sums of dy and dy*xhat -> part[2][tiles][c].
 * With mean==NULL (no norm after this conv) only dy is written (then dy IS the conv-output gradient). */
int pfa_bn_bwd_reduce(const void* raw, int64_t rows, int c, int dtype, const float* scale, const float* shift,
                      const float* mean, const float* invstd, const void* g1, int act1, const void* g2, int act2,
                      float slope, void* dy, float* part, int* ntiles, pfa_stream s);
/* Rows of `part` (per half) a pfa_bn_bwd_reduce call over [rows][c] writes; -1 if c is unsupported. */
int pfa_bn_bwd_tiles(int64_t rows, int c, int dtype);
int pfa_bn_bwd_finalize(const float* part, int ntiles, int c, double count, const float* gamma, const float* mean,
                        const float* invstd, float* dgamma, float* dbeta, float* coef, pfa_stream s);
/* dr = coef[0][c]*dy + coef[1][c]*raw + coef[2][c]   (may run in place on dy) */
int pfa_bn_bwd_apply(const void* dy, const void* raw, int64_t rows, int c, int dtype, const float* coef, void* dr,
                     pfa_stream s);

/* ------------------------------------------------------------------ losses (train_gan.py:42-43,58-69) */
/* BCEWithLogitsLoss, mean over `count` logits (channel 0 of a [count][ld] tensor), constant target.
 * loss_acc[0] += lscale * mean_loss; dlogits = gscale * (sigmoid(x)-target)/count (pad channels 0). */
int pfa_bce_logits(const void* logits, int64_t count, int ld, int dtype, float target, float lscale, float gscale,
                   float* loss_acc, void* dlogits, pfa_stream s);
/* L1Loss mean over count*c_real elements + Tanh backward, fused:
 * d = (dgan ? dgan[.., dgan_off + c] : 0) + lambda * sign(fake-real)/(count*c_real); dpre = d*(1-fake^2);
 * loss_acc[0] += lambda * mean|fake-real|. */
int pfa_l1_tanh_bwd(const void* fake, const void* real, int64_t count, int ld, int c_real, int dtype,
                    const void* dgan, int dgan_ld, int dgan_off, float lambda, float* loss_acc, void* dpre,
                    pfa_stream s);
/* The same two kernels with REPRODUCIBLE loss sums: every workgroup stores its partial - parts[0 .. n) with n =
 * pfa_bce_logits_nparts(count) <= 64 / pfa_l1_tanh_bwd_nparts(count, ld, dtype) <= 1024 - instead of adding to an
 * accumulator with float atomics (whose order varies run to run); pfa_loss_finish then adds ranges of partials in index
 * order: out[0] = sum parts[b0, e0) + sum parts[b1, e1), out[1] = sum parts[b2, e2) + sum parts[b3, e3) (an empty range:
 * b == e; b2 < 0: out[0] only).  The fused step (GanTrainer) uses these: its loss scalars are bitwise the same on every run. */
int pfa_bce_logits_nparts(int64_t count);
int pfa_bce_logits_parts(const void* logits, int64_t count, int ld, int dtype, float target, float lscale, float gscale,
                         float* parts, void* dlogits, pfa_stream s);
int pfa_l1_tanh_bwd_nparts(int64_t count, int ld, int dtype);
int pfa_l1_tanh_bwd_parts(const void* fake, const void* real, int64_t count, int ld, int c_real, int dtype,
                          const void* dgan, int dgan_ld, int dgan_off, float lambda, float* parts, void* dpre,
                          pfa_stream s);
int pfa_loss_finish(const float* parts, int b0, int e0, int b1, int e1, int b2, int e2, int b3, int e3, float* out,
                    pfa_stream s);

/* out[j] += sum_rows x[row][j] for j < c_out <= c (bias gradients; accumulates like every gradient output).
 * ws: >= pfa_colsum_ws(rows,c) floats. */
size_t pfa_colsum_ws(int64_t rows, int c);
int pfa_colsum(const void* x, int64_t rows, int c, int c_out, int dtype, float* out, float* ws, pfa_stream s);

/* ------------------------------------------------------------------ Adam (train_gan.py:140-141) */
/* torch.optim.Adam, weight_decay 0, amsgrad off, over one flat float32 arena.  `step` is the 1-based
 * step count of THIS update; when step_dev != NULL the count is read from device memory instead
 * (hipGraph replay) after being incremented by pfa_adam_tick.  The gradient is grad_scale * (g + g2); g2 (optional) lets
 * two backward passes that ran concurrently on different streams keep separate, race-free gradient arenas. */
int pfa_adam(float* p, const float* g, const float* g2, float* m, float* v, int64_t n, float lr, float beta1,
             float beta2, float eps, int step, const int32_t* step_dev, float grad_scale, pfa_stream s);
int pfa_adam_tick(int32_t* step_dev, pfa_stream s);

/* ------------------------------------------------------------------ layout / precision plumbing */
/* Master float32 parameter [rows][16][c1r+c2r] (channels-last memory of the OIHW / IOHW tensor) ->
 *   same: [rows][16][c1p+c2p] in `dtype` (pad channels zero), and
 *   tr  : [c1p+c2p][16][rows_p] in `dtype` (rows of pad channels zero).   Either may be NULL. */
int pfa_weight_shadow(const float* master, int rows, int c1r, int c1p, int c2r, int c2p, int rows_p, int dtype,
                      void* same, void* tr, pfa_stream s);
/* All layers of a network (at most 64) in one launch that reads each master once and writes both shadows: a
 * device-resident table of descriptors (same fields as above, plus the first block index of each layer - tr_blk0, the
 * running sum of pfa_shadow_blocks' tr_blocks; layers in ascending block order; same / tr may be NULL). */
typedef struct pfa_shadow_desc {
    const float* master;
    void* same;
    void* tr;
    int32_t rows, c1r, c1p, c2r, c2p, rows_p;
    int32_t same_blk0, tr_blk0;
} pfa_shadow_desc;
int pfa_shadow_blocks(int rows, int c1p, int c2p, int rows_p, int* same_blocks, int* tr_blocks);
int pfa_weight_shadow_batch(const pfa_shadow_desc* descs_dev, int n, int same_blocks, int tr_blocks, int dtype,
                            pfa_stream s);
/* optim.Adam (train_gan.py:140-141) and the kernel-side weight copies in ONE pass over the parameters: the launch of
 * pfa_weight_shadow_batch applies the update of pfa_adam (the same bits) to the master elements of each tile before it emits
 * the copies - p, g (+ g2), m, v are the flat arenas, every descriptor's `master` points into p - and a second small launch
 * updates the arena ranges that are not convolution weights (gaps_dev: ngaps pairs {first element, length}: BatchNorm gamma /
 * beta, biases).  Replaces pfa_adam + pfa_weight_shadow_batch over a range of the arena (28 + 8 -> 28 + 4 bytes per parameter). */
int pfa_adam_shadow_batch(const pfa_shadow_desc* descs_dev, int n, int tr_blocks, int dtype, const int64_t* gaps_dev, int ngaps,
                          float* p, const float* g, const float* g2, float* m, float* v, float lr, float beta1, float beta2,
                          float eps, int step, const int32_t* step_dev, float grad_scale, pfa_stream s);
/* NCHW float32 [n][c][h][w] <-> NHWC `dtype` [n][h][w][cp] (cp >= c, pad lanes zero / ignored). */
int pfa_pack_nhwc(const float* src, int n, int c, int h, int w, int cp, int dtype, void* dst, pfa_stream s);
int pfa_unpack_nchw(const void* src, int n, int c, int h, int w, int cp, int dtype, float* dst, pfa_stream s);
/* Inference output (generate_synthetic_data.py:66-71): NHWC [pixels][cp] -> packed uint8 [pixels][c],
 * u8 = byte((x*mul + add) * 255) with torch's rounding order (x*0.5+0.5, then TF.to_pil_image's mul(255).byte()). */
int pfa_unpack_u8(const void* src, int64_t pixels, int c, int cp, int dtype, float mul, float add, uint8_t* dst,
                  pfa_stream s);
int pfa_fill_zero(void* p, size_t bytes, pfa_stream s);
/* y[i] += alpha * x[i], float32 (loss bookkeeping of train_gan.py:63,71,74-75 kept on the device). */
int pfa_axpy(float* y, const float* x, int64_t n, float alpha, pfa_stream s);

/* ------------------------------------------------------------------ whole-batch finish of small K-sliced layers
 * pfa_gather_gemm with fuse_out = PFA_FUSE_DEFER runs the K slices only and leaves the float32 partial planes
 * [splitk][rows][n] in the slab (rows = output pixels; on return a->splitk holds the number of planes written).  For layers
 * with at most 4096 rows and n %% 16 == 0 (pfa_finish_bn_supported) the two entry points below then finish them in ONE
 * launch in which a workgroup owns all rows of a 16-column group:
 *   _fwd  conv output (bf16 `raw`) + BatchNorm2d training forward (nn.BatchNorm2d of models.py:179,181: batch statistics,
 *         running buffers updated `repeats` times, scale/shift/mean/invstd for the backward) + LeakyReLU -> out_l and/or
 *         ReLU -> out_r: replaces the finish pass + pfa_bn_finalize + pfa_bn_act.
 *   _bwd  input-gradient finish + the backward of the node the gradient belongs to, as fuse_out 1|2 + pfa_bn_bwd_finalize +
 *         pfa_bn_bwd_apply compute it: columns [0, n1) are a plain sum -> `plain`; columns [n1, n): d = act1'(y) g +
 *         act2'(y) g2 (y = scale raw + shift), and with mean / invstd the BatchNorm backward of d (dgamma / dbeta += ) -> dr. */
#define PFA_FUSE_DEFER 4
int pfa_finish_bn_supported(int dtype, int rows, int n);
int pfa_finish_bn_fwd(const float* slab, int splitk, int rows, int n, const float* gamma, const float* beta, float eps,
                      float momentum, int repeats, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                      int dtype, void* raw, int ld, void* out_l, void* out_r, float slope, float* scale, float* shift,
                      float* mean, float* invstd, pfa_stream s);
int pfa_finish_bn_bwd(const float* slab, int splitk, int rows, int n, int n1, void* plain, int ld1, const void* raw,
                      const void* g2, int act1, int act2, float slope, const float* scale, const float* shift,
                      const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta, int dtype,
                      void* dr, int ld2, pfa_stream s);

/* The same two finishes for the mid-size maps (up to 4096 rows, n %% 64 == 0: the generator's 8 x 8 and 16 x 16 levels at
 * batch 16), on up to 256 workgroups that meet once inside the launch (csrc/finish_coop.hip: partial sums as write-through
 * {epoch, value} granules, a BOUNDED wait).  Same arguments and results as pfa_finish_bn_fwd / _bwd (the statistics are
 * summed per block of 32-128 rows, then over the blocks; _bwd needs the node's BatchNorm, i.e. mean and invstd, and
 * n1 %% 64 == 0) plus two caller-owned workspaces: `part` (pfa_finish_coop_part_floats floats, 8-byte aligned) and `bar`
 * (pfa_finish_coop_bar_words 32-bit words), BOTH ZERO before the first launch that uses them (bar: word 0 is raised if a
 * launch gave up waiting - its results are then garbage; word 1 counts the launches).  A workspace serves one stream at a
 * time.  pfa_gather_plan_splitk: the slices pfa_gather_gemm will run the call in (the planners may keep fewer than asked
 * for; <= 1: nothing to finish, do not defer). */
int pfa_gather_plan_splitk(const pfa_gather_args* a);
int pfa_finish_coop_supported(int dtype, int rows, int n);
size_t pfa_finish_coop_part_floats(int rows, int n);
size_t pfa_finish_coop_bar_words(int n);
int pfa_finish_coop_fwd(const float* slab, int splitk, int rows, int n, const float* gamma, const float* beta, float eps,
                        float momentum, int repeats, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                        int dtype, void* raw, int ld, void* out_l, void* out_r, float slope, float* scale, float* shift,
                        float* mean, float* invstd, float* part, uint32_t* bar, pfa_stream s);
int pfa_finish_coop_bwd(const float* slab, int splitk, int rows, int n, int n1, void* plain, int ld1, const void* raw,
                        const void* g2, int act1, int act2, float slope, const float* scale, const float* shift,
                        const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta, int dtype,
                        void* dr, int ld2, float* part, uint32_t* bar, pfa_stream s);

/* ------------------------------------------------------------------ data-parallel exchange (one process per GPU)
 * Replaces, for this path, the gradient all-reduce DistributedDataParallel runs behind loss.backward()
 * (/root/reference/train_gan.py:62-63,70-71 under a torch.distributed launcher; BASELINE config 3): an RCCL communicator
 * over the ranks and an in-place SUM-all-reduce of a gradient range on the caller's stream - stream-ordered like every
 * other entry point, so the range's last producer and its first consumer are ordered by the stream (or by events) and a
 * launch list can hold the call.  The mean is the optimizer's grad_scale = 1 / ranks (pfa_adam).  RCCL is resolved with
 * dlopen at the first call (no link-time dependency; inside a PyTorch process the already-mapped librccl.so.1 is used).
 *   pfa_comm_unique_id  rank 0 draws the 128-byte identifier; the host transports it to the other ranks (any channel)
 *   pfa_comm_create     every rank, with its device current (hipSetDevice): collective over the ranks
 *   pfa_allreduce       buf[0..count) of dtype PFA_F32 | PFA_BF16 := sum over the ranks, in place, on stream s; every
 *                       rank issues the same calls in the same order
 * Measured on one rank only in this repository (a one-GPU box cannot hold two ranks of one communicator); the Python host
 * keeps torch.distributed as its default transport (GanTrainer(collectives="c") selects this one). */
typedef struct pfa_comm pfa_comm;
int pfa_comm_unique_id(void* out, int bytes);                 /* bytes must be 128 */
int pfa_comm_create(const void* unique_id, int bytes, int nranks, int rank, pfa_comm** out);
int pfa_comm_destroy(pfa_comm* c);
int pfa_comm_size(const pfa_comm* c);
int pfa_comm_rank(const pfa_comm* c);
int pfa_allreduce(pfa_comm* c, void* buf, int64_t count, int dtype, pfa_stream s);

/* ------------------------------------------------------------------ launch lists
 * A training step is ~210 launches on three streams with the same arguments every iteration (static shapes, persistent
 * buffers).  Issuing them from Python costs ~2.5 ms per step; a launch list is that sequence recorded once - every call of
 * an entry point above with its argument values, plus the event records / stream waits between the streams - and re-issued
 * by ONE call.  (hipGraph replay of the same fork/join pattern serialises the branches; this keeps the three streams.)
 *   PFA_OP_CALL    fn = pfa_replay_index(name); a = the call's arguments laid out as a C struct in declaration order
 *                  (natural alignment; a pointer-to-struct argument points at a caller-kept copy of that struct)
 *   PFA_OP_RECORD  hipEventRecord((hipEvent_t)a, (hipStream_t)b)
 *   PFA_OP_WAIT    hipStreamWaitEvent((hipStream_t)b, (hipEvent_t)a)
 * pfa_replay issues ops[0..n) in order on the calling thread; on a failing op it stops, stores its index in *failed_at (if
 * given) and returns that op's return code.  The caller keeps every buffer, stream and event of the list alive. */
enum { PFA_OP_CALL = 0, PFA_OP_RECORD = 1, PFA_OP_WAIT = 2 };
typedef struct pfa_replay_op {
    int32_t kind;
    int32_t fn;
    const void* a;
    void* b;
} pfa_replay_op;
int pfa_replay_index(const char* name);       /* -1: not a launching entry point */
int pfa_replay(const pfa_replay_op* ops, int n, int* failed_at);

#ifdef __cplusplus
}
#endif
#endif /* PFA_HIP_H */
