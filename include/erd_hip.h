/*
 * erd_hip.h -- C ABI of liberd_hip.so: the MI355X (gfx950) kernels of the ERD
 * incremental-detection training step.
 *
 * The reference (Hi-FT/ERD, a pure-Python fork of MMDetection 3.0.0) has no C
 * ABI of its own: every native kernel its hot path executes is reached through
 * a torch ATen / mmcv operator dispatch.  Each
 * entry point below therefore names the reference *call site* (file:line under
 * /root/reference) whose operator dispatch it replaces.
 *
 * Conventions:
 *   - plain device pointers + explicit sizes; fp32 activations are NHWC, weights
 *     are [Cout][kh][kw][Cin] (= OIHW tensors in channels_last memory format);
 *   - no allocation inside: the caller passes workspaces;
 *   - every call takes the hipStream_t it is enqueued on (as void*);
 *   - returns 0 on success, a negative ERD_E* argument error, or a positive
 *     hipError_t; never throws; message via erd_last_error() (thread-local);
 *   - no global mutable state; re-entrant.
 */
#ifndef ERD_HIP_H_
#define ERD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ERD_ABI_VERSION 6
#define ERD_MAX_SEG 5   /* FPN levels batched in one launch */
#define ERD_MAX_TAPS 9

#define ERD_EINVAL (-1)
#define ERD_EUNSUPPORTED (-2)

typedef void* erd_stream_t; /* hipStream_t */

/* Storage type of feature maps and their gradients in HBM.  ERD_F32: the headline configuration.  ERD_BF16: BASELINE.json
 * configs[2] -- what the reference's AMP switch does to conv outputs (tools/train.py:85-97): values are rounded to
 * nearest even when stored, widened exactly when loaded; every reduction, statistic, scale/shift and loss stays fp32
 * (GroupNorm statistics in fp64 as before).  Entry points that take maps as `void*` take a map_type next to them;
 * strides, offsets and extents are always in ELEMENTS. */
enum { ERD_F32 = 0, ERD_BF16 = 1 };

int erd_abi_version(void);
/* ABI v6.  CUs that every later launch leaves free: the persistent Winograd grids, the stream-K grids and the activation-stationary
 * kernel's grid are sized for (device CUs - reserve) -- data-parallel ranks leave room for RCCL's resident kernels, which otherwise cost
 * a whole-chip static grid a dispatch round (ERDTrainer.tune_cu_reserve picks the value with a short probe when a process group exists;
 * default 0).  n < 0 only queries; returns the previous reserve.  erd_usable_cus: device CUs - reserve (the caller sizes its one-round
 * weight-gradient splits with it). */
int erd_set_cu_reserve(int n);
int erd_usable_cus(void);
/* 0 for the product build; 1 when the library contains a timing / accuracy / trace probe variant of a kernel (csrc/erd_probes.h):
 * such builds exist for same-box A/B measurements only and must never be the library a training run loads */
int erd_probe_build(void);
const char* erd_last_error(void);
/* ABI v5.  sha256 (hex) over the sources this library was built from (csrc/Makefile: every .hip / .h of csrc/, the Makefile and this
 * header, concatenated in sorted path order; tools/csrc_sha.py computes the same value from a checkout): ties measurements to code */
const char* erd_csrc_sha(void);

/* ---- convolution as implicit GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32) --------------------
 * One "segment" = one feature map (level); a launch may batch up to ERD_MAX_SEG
 * segments that share weights (the GFL head, gfl_head.py:156-177 "weights shared
 * across the 5 levels").  The iteration grid of a segment is N x GH x GW points
 * (a,b); tap t reads in[n, a*in_stride+dy[t], b*in_stride+dx[t], :] (zero outside
 * [0,IH)x[0,IW)) and the result goes to out[n, a*out_stride+oy, b*out_stride+ox, :].
 * Forward conv, stride-1 dgrad and the four parity classes of a stride-2 dgrad are
 * all instances of this one form. */
typedef struct {
    const float* in;     /* [N][IH][IW][Cin], image stride in_nstride (elements) */
    float* out;          /* [N][OH][OW][Cout], image stride out_nstride */
    const float* res;    /* optional residual, same geometry as out (may alias out) */
    const float* alpha;  /* optional device scalar multiplied in the epilogue */
    const float* mask;   /* optional, same geometry as out: the result is zeroed where mask <= 0 (ReLU backward
                          * of the tensor whose gradient this launch produces) */
    int N, IH, IW, GH, GW, OH, OW;
    int64_t in_nstride, out_nstride, res_nstride;
    /* optional per-segment tap set (ntaps > 0): this segment uses taps [tap0, tap0 + ntaps) of the launch's dy/dx/wk
     * arrays and its own output offset (oy, ox) -- the four parity classes of a stride-2 input gradient (1 + 2 + 2 + 4
     * taps of a 3x3 kernel) then run as four segments of ONE launch; erd_conv_desc::ntaps must be the largest count.
     * ntaps == 0: the launch-wide taps and (oy, ox). */
    int tap0, ntaps, oy, ox;
} erd_conv_seg;

typedef struct {
    int nseg;
    erd_conv_seg seg[ERD_MAX_SEG];
    const float* w;      /* [Cout][wrow] ; tap t uses columns [wk[t], wk[t]+Cin) */
    int Cin, Cout, wrow;
    int ntaps;
    int dy[ERD_MAX_TAPS], dx[ERD_MAX_TAPS], wk[ERD_MAX_TAPS];
    int in_stride, out_stride, oy, ox;
    const float* scale;  /* optional [Cout]: v = acc*scale[co]            */
    const float* shift;  /* optional [Cout]: v += shift[co]  (bias / folded BN) */
    int relu;            /* v = max(v,0) */
    float* colsum;       /* optional [colsum_copies][Cout]: += column sums of the stored result (atomic) */
    /* optional stream-K workspace (>= erd_conv_igemm_ws_bytes()): lets the launch split the K loop of
     * boundary tiles across workgroups so that all CUs finish together; NULL = one workgroup per tile.
     * The ticket area (the last max_tiles*4 bytes) must be ZERO on entry; the kernel leaves it zero. */
    void* sk_ws;
    size_t sk_ws_bytes;
    /* optional: the same weights rounded to bf16, [Cout][wrow] (erd_to_bf16).  When set the launch runs on the
     * bf16 matrix cores (v_mfma_f32_32x32x16_bf16): activations are rounded to bf16 as they are staged, products
     * are exact, accumulation / epilogue / output stay fp32 (BASELINE.json configs[2]; `w` is then unused). */
    const void* w_bf16;
    /* 0 / 1: one row of sums.  2^k: workgroup b adds into row (b mod 2^k) -- same-address float atomics retire at about
     * six per microsecond, a 1000-tile launch would otherwise wait for them; the consumer folds the rows (erd_bn_dgamma). */
    int colsum_copies;
    /* bf16 storage of the maps (only with w_bf16): in_bf16 = every seg.in holds bf16 values, out_bf16 = every seg.out /
     * seg.res / seg.mask holds bf16 values (round to nearest even on store).  Strides and extents stay in ELEMENTS.
     * 0 / 0 = fp32 maps.  Cin % 8 == 0 with in_bf16.  The reference's AMP switch (tools/train.py:85-97) stores conv
     * outputs in the low-precision type the same way. */
    int in_bf16, out_bf16;
    /* optional ("f32x3" mode): the same weights as three bf16 LIMB planes, [3][Cout][wrow] (erd_split3 / erd_weight_transpose_x3):
     * plane 0 = each fp32 weight rounded to bf16 (nearest even), plane 1 = the remainder rounded to bf16, plane 2 = what is left
     * (at most 8 significant bits) -- w == p0 + p1 + p2 EXACTLY, |p1| <= 2^-8 |w|, |p2| <= 2^-16 |w|, remainders sign-symmetric.  When set (and Cin % 4 == 0, Cout % 4 == 0; otherwise the launch needs `w`) maps, accumulation
     * and results stay fp32 but every product a*w is formed on the bf16 matrix cores as the sum of the six limb products of
     * weight >= 2^-16 (the activation is split the same way in registers): the dropped terms are zero-mean and below 2^-23 |a*w|,
     * fp32's own rounding of that product (the round-3 library split by truncation: dropped terms up to 2^-21 |a*w|, all of the product's sign).  gfx950's fp32 MFMA runs at 1/16 of the bf16 rate; six bf16 MFMAs cost 3/8 of one fp32 MFMA. */
    const void* w_x3;
} erd_conv_desc;

/* replaces: F.conv2d dispatches at resnet.py:268-283, res_layer.py:57-63, fpn.py:196,215-220,
 * gfl_head.py:224-229 (+ fused frozen-stat BN resnet.py:268-300 / bias / ReLU / residual add);
 * and, run on dz with transformed weights, their convolution_backward (input grad). */
int erd_conv_igemm(const erd_conv_desc* d, erd_stream_t stream);
size_t erd_conv_igemm_ws_bytes(int max_tiles);
/* ABI v4.  erd_conv_igemm sends three-limb 1x1 launches with Cin in {64, 128} and Cout % 32 == 0 to an activation-stationary
 * kernel (csrc/conv_thin.hip: the pixel rows' limb fragments stay in registers, the output channels stream past them in blocks
 * of 32; the same MFMA sequence per accumulator as the stream-K kernel, i.e. bit-identical results).  on = 0 / 1 switches that
 * dispatch for the process (A/B runs and the bit-identity test), on < 0 only queries; returns the previous setting.
 * Default: on (environment ERD_THIN=0: off). */
int erd_conv_thin_enable(int on);
/* ABI v6.  1 when erd_conv_igemm would run THIS launch on the activation-stationary kernel (conv_thin_x3_kernel), 0 when on the stream-K
 * kernel (conv_igemm_kernel): the dispatch predicate itself, for callers that book launches per kernel symbol (bench.py's per-kernel
 * roofline rows must agree with what a profiler sees). */
int erd_conv_thin_ok(const erd_conv_desc* d);
/* dst[i] = bf16(src[i]) (round to nearest even), n elements */
int erd_to_bf16(const float* src, void* dst, int64_t n, erd_stream_t stream);
/* the three bf16 limbs of every value, each rounded to nearest even: dst[0][i] + dst[1][i] + dst[2][i] == src[i] exactly; dst = [3][n] bf16
 * (erd_conv_desc::w_x3 of a forward convolution) */
int erd_split3(const float* src, void* dst, int64_t n, erd_stream_t stream);

/* ---- Winograd F(2x2,3x3) for the stride-1 3x3 convolutions (fp32 matrix cores, 2.25x fewer multiplications) ----
 * erd_wino_weights: U = G g G^T of w [Cout][3][3][Cin] in the tiled layout the kernel streams
 * ([16 positions][ceil(Cout/16)][Cin/16][4 k-quads][16 couts][4 k]: one MFMA B-fragment of a wave is 1 KB contiguous;
 * erd_wino_weights_elems = 16 * ceil16(Cout) * Cin floats, rows past Cout are zero).  erd_wino_conv3x3: out = epi(conv3x3(in)),
 * stride 1, padding 1, over up to ERD_MAX_SEG maps sharing U (segments as in erd_conv_desc: in/out/N/IH/IW/
 * in_nstride/out_nstride; OH == IH, OW == IW).  Cin % 16 == 0, Cin >= 64, fp32 maps only.  The input gradient of such a layer is the same
 * call on dz with U built from the transposed weights ([Cin][3][3][Cout]) with flip = 1.
 * replaces: the F.conv2d dispatches of gfl_head.py:219-229 (towers), fpn.py:215 (outputs), resnet.py:270-274 (conv2). */
size_t erd_wino_weights_elems(int Cout, int Cin);
int erd_wino_weights(const float* w_ohwi, float* U, int Cout, int Cin, int flip /* reverse the 9 taps */,
                     erd_stream_t stream);
/* epilogue: v = acc*scale + shift (+ seg.res) ; ReLU ; zero where seg.mask <= 0 ; colsum[co] += v (atomic) */
int erd_wino_conv3x3(const erd_conv_seg* segs, int nseg, const float* U, int Cin, int Cout, const float* scale,
                     const float* shift, int relu, float* colsum, int colsum_copies /* see erd_conv_desc */,
                     int* sched /* optional 2 ints, zero on entry (left zero): dynamic item scheduling of the
                                   persistent grid; NULL = static split */,
                     erd_stream_t stream);

/* The THREE-LIMB form of the same algorithm (ABI v4; "f32x3"): fp32 maps, accumulation and results, the 16 transform-domain GEMMs on
 * the bf16 matrix cores (six limb products per fp32 product, v_mfma_f32_32x32x16_bf16).  erd_wino_weights_x3 builds U = G g G^T with
 * erd_wino_weights' arithmetic and stores its three bf16 limbs as MFMA fragments ([16 positions][3 limbs][ceil(Cout/32)][Cin/16]
 * [64 lanes][8]: erd_wino_weights_x3_elems bf16 values); erd_wino_conv3x3_x3 takes that image, everything else as
 * erd_wino_conv3x3 (same item list, epilogue, scheduling counters). */
size_t erd_wino_weights_x3_elems(int Cout, int Cin);
int erd_wino_weights_x3(const float* w_ohwi, void* U3, int Cout, int Cin, int flip, erd_stream_t stream);
int erd_wino_conv3x3_x3(const erd_conv_seg* segs, int nseg, const void* U3, int Cin, int Cout, const float* scale,
                        const float* shift, int relu, float* colsum, int colsum_copies, int* sched, erd_stream_t stream);
/* ABI v5.  erd_wino_conv3x3_x3 runs an item of 32 tiles x 128 output channels (wino_x3p_kernel: every tile block is transformed once
 * per 128 couts, eight waves that each multiply and transform) where Cout % 128 == 0 and that needs fewer dispatch rounds of the
 * persistent grid, 32 tiles x 64 couts (wino_x3_kernel) otherwise; results are bit-identical either way.
 * erd_wino_x3_couts_per_item tells the caller which (128 / 64; a profiler sees two kernel symbols), < 0 on bad segments. */
int erd_wino_x3_couts_per_item(const erd_conv_seg* segs, int nseg, int Cout);
/* ABI v6.  erd_wino_conv3x3_x3 on a PLAIN convolution (no residual / mask / column sums) that also produces the raw material of the
 * GroupNorm(32) statistics of what it stores (gfl_head.py:158-177: conv -> GN -> ReLU; replaces the statistics pass over the conv output,
 * its zero fill and the finalize launch): every item's output stage writes its 16 groups' (sum, sum of squares) to `gn_part` (a workspace
 * of erd_wino_x3_gn_ws_bytes; no atomics, deterministic).  erd_wino_gn_finalize (same segments) adds them in f64 and writes
 * mean_rstd[N][nseg][Cout / 8][2] = (mean, 1 / sqrt(var + eps)); erd_gn_relu_apply consumes mean_rstd.  Served by the 128-couts-per-item
 * kernel only (erd_wino_x3_couts_per_item == 128; erd_wino_x3_gn_ws_bytes returns 0 otherwise). */
int erd_wino_conv3x3_x3_gn(const erd_conv_seg* segs, int nseg, const void* U3, int Cin, int Cout, const float* scale,
                           const float* shift, int relu, int* sched, float* gn_part, size_t gn_part_bytes, erd_stream_t stream);
int erd_wino_gn_finalize(const erd_conv_seg* segs, int nseg, int Cout, const float* gn_part, float* mean_rstd, float eps,
                         erd_stream_t stream);
size_t erd_wino_x3_gn_ws_bytes(const erd_conv_seg* segs, int nseg, int Cout);

/* weight gradient: G[co][t][ci] = sum_p dz[p,co] * x[p shifted by tap t, ci], split-K over
 * pixels into `nsplit` partial slabs part[s][Cout][ntaps][Cin] (deterministic two-stage reduce).
 * Up to ERD_MAX_SEG feature maps that share the weights (the head's five levels) are summed in ONE
 * launch: their pixels are concatenated along K; every segment is given by element offsets from the
 * common base pointers x / dz (the level-concatenated buffers).
 * replaces: convolution_backward (weight grad) of the same call sites. */
typedef struct {
    int64_t x_off, dz_off;   /* element offsets of this map inside x / dz */
    int N, IH, IW, GH, GW, OH, OW;
    int64_t x_nstride, dz_nstride;
} erd_wgrad_seg;

typedef struct {
    const float* x;      /* base of the input maps  [N][IH][IW][Cin] */
    const float* dz;     /* base of the output-gradient maps [N][OH][OW][Cout], read at (a*out_stride+oy, ...) */
    int64_t x_elems, dz_elems;   /* extent of the two allocations (for the buffer descriptors) */
    int nseg;
    erd_wgrad_seg seg[ERD_MAX_SEG];
    int Cin, Cout, ntaps;
    int dy[ERD_MAX_TAPS], dx[ERD_MAX_TAPS];
    int in_stride, out_stride, oy, ox;
    float* part;         /* [nsplit][Cout][ntaps][Cin] */
    int nsplit;
    int bf16_multiplicands; /* 1: round dz and x to bf16 as they are staged and use the bf16 matrix cores (fp32 partials) */
    int x_bf16, dz_bf16;    /* 1: the x / dz maps are stored as bf16 (offsets, strides, extents stay in elements); only
                             * with bf16_multiplicands */
    int limbs3;             /* 1 ("f32x3"): fp32 maps and fp32 partial slabs, but every product dz * x is formed on the bf16 matrix
                             * cores from exact three-limb splits of both values (see erd_conv_desc::w_x3): both operands are
                             * split in the kernel's transposing loader (4 pixels x 4 channels per thread).  3x3 / stride-1 layers:
                             * three taps of a kernel row per workgroup (the taps are register shifts of the packed pixel vectors);
                             * every other layer: one tap, 128 x 128 channels per workgroup.  Cin % 4 == 0. */
} erd_wgrad_desc;
int erd_conv_wgrad(const erd_wgrad_desc* d, erd_stream_t stream);
/* > 0: the layer takes the three-taps-per-workgroup kernel (3x3, stride 1, pad 1, fp32) and this is its number of
 * K-slices (16-pixel row chunks) -- the caller sizes `nsplit` from it; 0: generic kernel. */
int erd_wgrad_row3_slices(const erd_wgrad_desc* d);

/* dW (+)= rowscale[co] * sum_s part[s]; optional rowdot[co] = <W[co,:], sum_s part[s][co,:]>
 * (used for d gamma of a frozen-statistics BN: resnet.py:648-657 keeps BN in eval while
 * gamma/beta train).  accumulate: bit 0 = add into dW instead of storing; bit 1 = rowdot already holds zeros
 * (otherwise it is cleared here). */
int erd_wgrad_reduce(const float* part, int nsplit, int Cout, int K, const float* w,
                     const float* rowscale, float* dW, int accumulate, float* rowdot,
                     erd_stream_t stream);

/* dst[ci][t'][co] = rowscale[co] * w[co][t][ci]  (t' = flip ? ntaps-1-t : t): weights of the
 * input-gradient convolution. */
int erd_weight_transpose(const float* w, const float* rowscale, float* dst, int Cout, int ntaps,
                         int Cin, int flip, erd_stream_t stream);
/* the same, rounded to bf16 (round to nearest even) for erd_conv_desc::w_bf16 */
int erd_weight_transpose_bf16(const float* w, const float* rowscale, void* dst, int Cout, int ntaps,
                              int Cin, int flip, erd_stream_t stream);
/* the same as three bf16 limb planes [3][Cin][ntaps][Cout] for erd_conv_desc::w_x3 of an input-gradient convolution */
int erd_weight_transpose_x3(const float* w, const float* rowscale, void* dst, int Cout, int ntaps,
                            int Cin, int flip, erd_stream_t stream);

/* Many weight transforms in ONE launch: erd_weight_transpose (kind 0), erd_weight_transpose_bf16 (kind 1),
 * erd_wino_weights (kind 2: w is [Cout][3][3][Cin], ntaps = 9, rowscale unused), erd_split3 (kind 3: the Cout * ntaps * Cin
 * values at w, rowscale / flip unused) or erd_wino_weights_x3 (kind 4, as kind 2) per item, same arithmetic.  `items_dev` is a
 * DEVICE array sorted by block0; item i owns blocks [block0, block0 + erd_weight_prep_blocks(kind, Cout, ntaps, Cin)) of the
 * launch, total_blocks is their sum.  Items of one launch must not depend on each other (a Winograd image of a transposed
 * weight goes into a second launch).  The trainer prepares everything the step derives from the parameters alone this
 * way, right after the optimizer update. */
typedef struct {
    const float* w;
    const float* rowscale;
    void* dst;
    int Cout, ntaps, Cin, flip;
    int kind;
    int block0;
} erd_weight_prep_item;
int erd_weight_prep_blocks(int kind, int Cout, int ntaps, int Cin);
int erd_weight_prep_batch(const erd_weight_prep_item* items_dev, int nitems, int total_blocks, erd_stream_t stream);

/* ---- stem (resnet.py:636-639): conv7x7/2 (3->64) + frozen BN + ReLU, then maxpool 3x3/2 ---- */
int erd_stem_conv7x7_bn_relu(const float* x_nchw, const float* w_ohwi, const float* scale,
                             const float* shift, float* out_nhwc, int N, int H, int W,
                             erd_stream_t stream);
int erd_maxpool3x3s2(const float* in, void* out, int N, int H, int W, int C, int out_type /* ERD_F32 | ERD_BF16 */,
                     erd_stream_t stream);

/* ---- DetDataPreprocessor (data_preprocessor.py:110-183; mmengine ImgDataPreprocessor): one CHW image (uint8 or
 * fp32, device) -> out[3][H][W]: optional channel flip (bgr_to_rgb), (x - mean) / std, pad_value outside (h, w).
 * mean3 / std3 are HOST pointers to 3 floats. */
int erd_preprocess_image(const void* img, int is_uint8, int h, int w, float* out, int H, int W,
                         const float* mean3, const float* std3, int flip_channels, float pad_value,
                         erd_stream_t stream);

/* Resize(keep_ratio, bilinear) + RandomFlip + DetDataPreprocessor in one pass over the padded output slot
 * (configs/gfl_increment/ *.py:13-19 pipeline; mmcv.imrescale backend cv2, restated and UNPINNED vs cv2): src is the decoded
 * uint8 image [sh][sw][3] on the device, {x,y}ofs / {x,y}coef the per-output-pixel source index and the two 11-bit
 * fixed-point weights along each axis (host-built, device arrays), (nh, nw) the resized size inside the (H, W) slot. */
int erd_resize_normalize(const void* src_hwc_u8, int sh, int sw, const int* xofs, const short* xcoef,
                         const int* yofs, const short* ycoef, int nh, int nw, float* out, int H, int W,
                         const float* mean3, const float* std3, int flip, int swap_rb, float pad_value,
                         erd_stream_t stream);

/* ---- frozen-statistics BN helpers ---------------------------------------------------------- */
/* scale = gamma*rsqrt(var+eps), shift = beta-mean*scale over n channels (resnet.py:268-300, eval BN) */
int erd_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                float eps, float* scale, float* shift, int64_t n, erd_stream_t stream);
/* the same for many BNs in ONE launch (the trainable BNs of the student are re-folded once per step, right after the
 * optimizer update: 42 five-microsecond launches on the forward pass's critical path otherwise).  `items_dev` is a DEVICE
 * array of nitems entries, max_n the largest n among them. */
typedef struct {
    const float* gamma;
    const float* beta;
    const float* mean;
    const float* var;
    float* scale;
    float* shift;
    int n;
    float eps;
} erd_bn_fold_item;
int erd_bn_fold_batch(const erd_bn_fold_item* items_dev, int nitems, int max_n, erd_stream_t stream);
/* use_relu: dz = dy * (y > 0) (dz may alias dy); else dz is not written (dz == dy semantically);
 * colsum[c] += sum_p dz[p,c].  Rows are [npix][C] with an image stride (level views). */
int erd_relu_bwd_colsum(const void* y, const void* dy, void* dz, int64_t npix, int C,
                        int64_t nstride_rows, int64_t rows_per_img, float* colsum, int use_relu,
                        int map_type /* of y, dy, dz */, erd_stream_t stream);
/* dbeta_sum = sum over the `copies` rows of dbeta[copies][C] (stored to dbeta_out when given);
 * dgamma = rsqrt(var+eps) * (rowdot - mean*dbeta_sum) */
int erd_bn_dgamma(const float* rowdot, const float* dbeta, int copies, const float* mean, const float* var,
                  float eps, float* dgamma, float* dbeta_out, int accumulate, int C, erd_stream_t stream);

/* ---- GroupNorm(32)+ReLU over level-concatenated [N][A][C] maps (gfl_head.py:158-177) -------- */
typedef struct {
    int nseg;
    int64_t off[ERD_MAX_SEG]; /* first row of the level inside an image's A rows */
    int64_t cnt[ERD_MAX_SEG]; /* rows (pixels) of the level per image */
} erd_levels;
int erd_gn_relu_fwd(const void* c, void* y, const float* gamma, const float* beta, double* stats_ws,
                    float* mean_rstd, int N, int64_t A, int C, int G, const erd_levels* lv, float eps,
                    int map_type /* of c, y */, erd_stream_t stream);
/* ABI v6.  The normalisation pass of erd_gn_relu_fwd alone, for statistics that came from the producing convolution
 * (erd_wino_conv3x3_x3_gn writes mean_rstd[N][nseg][G][2]): one launch, y = ReLU((c - mean) * rstd * gamma + beta). */
int erd_gn_relu_apply(const void* c, void* y, const float* gamma, const float* beta, const float* mean_rstd, int N,
                      int64_t A, int C, int G, const erd_levels* lv, int map_type /* of c, y */, erd_stream_t stream);
int erd_gn_relu_bwd(const void* c, const void* dy, const float* gamma, const float* beta,
                    const float* mean_rstd, double* stats_ws, void* dc, float* dgamma, float* dbeta,
                    int N, int64_t A, int C, int G, const erd_levels* lv, int map_type /* of c, dy, dc */,
                    erd_stream_t stream);

/* ---- FPN top-down (fpn.py:181-191): lat[l-1] += nearest2x(lat[l]) and its adjoint -------------- */
int erd_upsample2x_add(void* fine, const void* coarse, int N, int H, int W, int C, int h, int w,
                       int64_t fine_nstride, int64_t coarse_nstride, int map_type, erd_stream_t stream);
int erd_upsample2x_add_bwd(const void* dfine, void* dcoarse, int N, int H, int W, int C, int h, int w,
                           int64_t fine_nstride, int64_t coarse_nstride, int map_type, erd_stream_t stream);

/* ---- small dense helpers -------------------------------------------------------------------- */
int erd_colsum(const void* x, int64_t rows, int C, float* out, int accumulate, int map_type /* of x */, erd_stream_t stream);
/* y[n][a][:] = x[n][a][:] * alphas[level(a)] (gfl_head.py:229, one learnable Scale per level) and adjoint */
int erd_level_scale(const float* x, const float* alphas, float* y, int N, int64_t A, int C,
                    const erd_levels* lv, erd_stream_t stream);
int erd_level_scale_bwd(const float* x, const float* dy, const float* alphas, float* dx, float* dalphas,
                        int N, int64_t A, int C, const erd_levels* lv, erd_stream_t stream);
/* SGD(momentum, weight decay) over one flat buffer (torch.optim.SGD; config :112-114) */
int erd_sgd_momentum(float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                     float weight_decay, float grad_scale, int first_step, erd_stream_t stream);

/* ---- ERS (gfl_increment_erd.py:143-163) -------------------------------------------------------- */
/* per image: m_c = max_k sigmoid(cls[a,k]), keep iff m_c > mean + 2*std (unbiased); m_b = max_j
 * bbox[a,j] likewise.  Writes masks [N][A] (uint8), ascending index lists idx_*[N][A] (int64) and
 * counts[N][2] (int32), thresholds thr[N][2]. */
int erd_ers_select(const float* cls, const float* bbox, int N, int64_t A, int Ccls, int Cbox,
                   uint8_t* mask_cls, uint8_t* mask_bbox, int64_t* idx_cls, int64_t* idx_bbox,
                   int32_t* counts, float* thr, double* ws, erd_stream_t stream);

/* ---- anchors / ATSS (anchor_generator.py:259-301; atss_assigner.py:74-254) ---------------------- */
int erd_grid_anchors(float* anchors, const int* hs, const int* ws, const int* strides, int nlvl,
                     int octave_scale, erd_stream_t stream);
/* labels[N][A] int64 (bg = num_classes), label_weights[N][A], bbox_targets[N][A][4], num_pos[N].
 * gt boxes of all images concatenated, gt_off[N+1] (device); max_gt = max boxes per image (host);
 * valid = optional [N][A] flags (anchor_generator.py:415-476); ws = N*A*8 bytes. */
int erd_atss_assign(const float* anchors, const uint8_t* valid, const int64_t* lvl_off, int nlvl,
                    int64_t A, const float* gt_boxes, const int64_t* gt_labels, const int32_t* gt_off,
                    int N, int max_gt, int topk, int num_classes, int64_t* labels, float* label_weights,
                    float* bbox_targets, int32_t* num_pos, void* ws, erd_stream_t stream);

/* ---- supervised losses on the new classes (gfl_head_increment_erd.py:225-322) ------------------- */
/* One launch over all N*A anchors (level-concatenated).  Produces per-level sums
 * out[nlvl][4] = {qfl_sum, giou_sum(w-weighted), dfl_sum(w-weighted), sum w} and, in the same pass,
 * the gradients d/d cls[:, c_old:] and d/d bbox of (qfl_sum*gq + giou_sum*gb[l] + dfl_sum*gd[l]). */
int erd_gfl_losses_fwd(const float* cls, const float* bbox, const float* anchors, const int64_t* labels,
                       const float* label_weights, const float* bbox_targets, const int64_t* lvl_off,
                       const int* strides, int nlvl, int N, int64_t A, int c_old, int c_all,
                       float* score_ws, float* wt_ws, double* out_sums, erd_stream_t stream);
int erd_gfl_losses_bwd(const float* cls, const float* bbox, const float* anchors, const int64_t* labels,
                       const float* label_weights, const float* bbox_targets, const int64_t* lvl_off,
                       const int* strides, int nlvl, int N, int64_t A, int c_old, int c_all,
                       const float* score_ws, const float* wt_ws, const float* coef /* erd_loss_finalize layout */,
                       float* dcls, float* dbbox, erd_stream_t stream);

/* ---- response distillation (gfl_head_increment_erd.py:142-223) ------------------------------------ */
/* L2 over ERS-selected rows: sums[n] = sum_{a in idx_cls[n], k<c_old} (s-t)^2 ; bwd adds coef[n]*2(s-t) */
int erd_l2_distill(const float* s_cls, const float* t_cls, const int64_t* idx_cls, const int32_t* counts,
                   int N, int64_t A, int c_s, int c_t, int c_old, double* sums, erd_stream_t stream);
int erd_l2_distill_bwd(const float* s_cls, const float* t_cls, const int64_t* idx_cls,
                       const int32_t* counts, const float* coef, int N, int64_t A, int c_s, int c_t,
                       int c_old, float* dcls, erd_stream_t stream);
/* teacher boxes (D8 unit mix) + class-offset NMS (mmcv.ops.batched_nms restated, UNPINNED) */
int erd_distill_nms(const float* t_cls, const float* t_bbox, const float* anchors,
                    const int64_t* idx_bbox, const int32_t* counts, int N, int64_t A, int c_t,
                    float iou_thr, uint8_t* keep_mask, int32_t* keep_count, float* ws, size_t ws_bytes,
                    erd_stream_t stream);
/* KD-KL(T) over kept rows (kd_loss.py:12-37): sums[n] = sum_r w_r * T^2 * mean_j KL ; bwd adds grads */
int erd_kd_kl(const float* s_bbox, const float* t_bbox, const float* s_cls, const uint8_t* keep_mask,
              int N, int64_t A, int c_s, int c_old, float T, double* sums, erd_stream_t stream);
int erd_kd_kl_bwd(const float* s_bbox, const float* t_bbox, const float* s_cls, const uint8_t* keep_mask,
                  const float* coef, int N, int64_t A, int c_s, int c_old, float T, float* dbbox,
                  erd_stream_t stream);

/* loss vector [3*nlvl + 2N] = loss_cls[l] | loss_bbox[l] | loss_dfl[l] | loss_dist_cls[n] | loss_dist_bbox[n]
 * from the kernel sums (gfl_head_increment_erd.py:390-409,436-454; losses/utils.py:30-65) and/or the
 * backward coefficients coef[4*nlvl + 2N] given the upstream gradient of the loss vector.
 * avg[2] = {reduce_mean(sum num_pos), reduce_mean(sum weight_targets)} (device). */
int erd_loss_finalize(const double* lvl_sums, const float* avg, const double* l2_sums, const double* kd_sums,
                      const int32_t* counts, int nlvl, int N, int c_old, float dist_loss_weight,
                      float lw_cls, float lw_bbox, float lw_dfl, float lw_ld, const float* upstream,
                      float* losses, float* coef, erd_stream_t stream);

/* avg[0] = sum_n max(num_pos[n],1), avg[1] = sum_l lvl_sums[l][3]  (the two normalisers that
 * reduce_mean all-reduces: gfl_head_increment_erd.py:390-391,405-407) */
int erd_loss_avg(const int32_t* num_pos, int N, const double* lvl_sums, int nlvl, float* avg,
                 erd_stream_t stream);

/* ---- inference post-processing ----------------------------------------------
 * GFLHead._predict_by_feat_single (gfl_head.py:408-502) + filter_scores_and_topk (models/utils/misc.py:308-354):
 * per (image, level) the nms_pre highest sigmoid scores above score_thr (score desc, then (anchor, class) index
 * asc), decoded with Integral x stride around the anchor centre and clamped to img_hw[n] = (H, W).
 * cls [N][A][C], bbox [N][A][68] level-concatenated; outputs are level-major per image:
 * boxes [N][nlvl*nms_pre][4], scores / labels [N][nlvl*nms_pre], num[N] = valid entries. */
size_t erd_predict_ws_bytes(int N, int nlvl, int nms_pre);
int erd_predict_topk(const float* cls, const float* bbox, const float* anchors, int N, int64_t A, int C,
                     const erd_levels* lv, const int* strides, const float* img_hw, float score_thr,
                     int nms_pre, float* boxes, float* scores, int32_t* labels, int32_t* num, void* ws,
                     size_t ws_bytes, erd_stream_t stream);
/* BaseDenseHead._bbox_post_process (base_dense_head.py:424-486): boxes * inv_scale[n] = (1/sw, 1/sh), drop
 * w or h <= min_bbox_size (min_bbox_size < 0: keep all), mmcv.ops.batched_nms restated (class offsets in fp32,
 * IoU > iou_thr suppresses; UNPINNED vs mmcv), first max_per_img survivors in score order.
 * dets [N][max_per_img][5] = (x1,y1,x2,y2,score), det_labels [N][max_per_img] int64, det_num[N].
 * ws: N*max_cols*48 bytes. */
int erd_predict_nms(const float* boxes, const float* scores, const int32_t* labels, const int32_t* num,
                    int N, int max_cols, const float* inv_scale, float min_bbox_size, float iou_thr,
                    int max_per_img, float* dets, int64_t* det_labels, int32_t* det_num, void* ws,
                    size_t ws_bytes, erd_stream_t stream);

/* ---- stand-alone leaf operators: what the registered loss / coder / assigner MODULES run when a caller invokes them
 * directly (the training step itself uses the fused erd_gfl_losses_* / erd_kd_kl* kernels above).  Row-parallel; every
 * loss comes as `rows` (the reference's reduction='none' value, after the per-row sum / mean the reference applies) plus
 * a backward that takes coef[i] = upstream * loss_weight * weight[i] / denominator.
 *   erd_qfl_rows / erd_qfl_bwd   quality_focal_loss, beta = 2 (losses/gfocal_loss.py:12-53): rows[i] = sum_k loss[i][k]
 *   erd_dfl                      distribution_focal_loss over nb bins (gfocal_loss.py:143-165); rows and/or dpred
 *   erd_kd_kl_rows               knowledge_distillation_kl_div_loss, temperature T (kd_loss.py:12-37)
 *   erd_giou                     giou_loss rows 1 - GIoU and d/dpred (iou_loss.py:110-126; eps as bbox_overlaps clamps it)
 *   erd_bbox_overlaps            bbox_overlaps mode 0 'iou' / 1 'giou', aligned [A] or pairwise [A][G] (bbox_overlaps.py:13-199)
 *   erd_integral                 Integral: softmax over nb bins . [0..nb-1] (gfl_head.py:29-62); y and/or dx
 *   erd_distance2bbox            DistancePointBBoxCoder.decode / distance2bbox (transforms.py:147-198), max_w < 0: no clamp;
 *                                out and/or the gradient w.r.t. the distances
 *   erd_bbox2distance            encode / bbox2distance (transforms.py:201-230), max_dis < 0: no clamp
 *   erd_weighted_sum             out[0] = scale * sum rows[i] * weight[i]  (weight_reduce_loss, losses/utils.py:30-65), f64 sums
 *   erd_loss_coef                coef[i] = upstream[0] * scale * weight[i]
 *   erd_rows_mul                 out[i] = rows[i] * scale * weight[i]  (reduction='none' and its backward)
 *   erd_atss_result              AssignResult fields (gt_inds 0 / g+1, max_overlaps with -1e8 for unassigned priors, labels
 *                                -1 / class) of ONE image from the key workspace erd_atss_assign leaves behind
 *                                (atss_assigner.py:238-254) */
int erd_qfl_rows(const float* pred, const int64_t* label, const float* score, int64_t n, int C, float* rows, erd_stream_t stream);
int erd_qfl_bwd(const float* pred, const int64_t* label, const float* score, const float* coef, int64_t n, int C, float* dpred,
                erd_stream_t stream);
int erd_dfl(const float* pred, const float* target, const float* coef, int64_t m, int nb, float* rows, float* dpred,
            erd_stream_t stream);
int erd_kd_kl_rows(const float* pred, const float* soft, const float* coef, int64_t m, int nb, float T, float* rows, float* dpred,
                   erd_stream_t stream);
int erd_giou(const float* pred, const float* target, const float* coef, int64_t n, float eps, float* rows, float* dpred,
             erd_stream_t stream);
int erd_bbox_overlaps(const float* b1, const float* b2, int64_t A, int64_t G, int aligned, int mode, float eps, float* out,
                      erd_stream_t stream);
int erd_integral(const float* x, const float* dy, int64_t m, int nb, float* y, float* dx, erd_stream_t stream);
int erd_distance2bbox(const float* points, const float* dist, const float* dout, int64_t n, float max_h, float max_w, float* out,
                      float* ddist, erd_stream_t stream);
int erd_bbox2distance(const float* points, const float* boxes, int64_t n, float max_dis, float eps, float* out, erd_stream_t stream);
int erd_weighted_sum(const float* rows, const float* weight, int64_t n, double scale, float* out, erd_stream_t stream);
int erd_loss_coef(const float* upstream, const float* weight, int64_t n, float scale, float* coef, erd_stream_t stream);
int erd_rows_mul(const float* rows, const float* weight, int64_t n, float scale, float* out, erd_stream_t stream);
int erd_atss_result(const void* assign_ws, const int64_t* gt_labels, int64_t A, int64_t* gt_inds, float* max_overlaps,
                    int64_t* labels, erd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ERD_HIP_H_ */
