// ERD-specific per-anchor kernels: Elastic Response Selection, anchors, ATSS assignment, the new-class
// QFL/GIoU/DFL losses (forward + analytic backward), L2 / NMS-filtered KD-KL response distillation.
// All are HBM scans over level-concatenated [N][A][C] fp32 maps (coalesced float4 rows, wavefront
// reductions, f64 accumulation of the statistics that feed data-dependent decisions).
// Built with -ffp-contract=off: index/mask decisions must not depend on FMA contraction.
#include "erd_common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ int64_t imin64(int64_t a, int64_t b) { return a < b ? a : b; }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float softplusf_(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = erd::wave_sum_d(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double t = 0;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// ------------------------------------------------------------------------------------------------
// ERS (gfl_increment_erd.py:143-163)
// ------------------------------------------------------------------------------------------------
// stage 1: row maxima.  A block handles 256 consecutive rows: the [256][C] slab is read with coalesced
// float4 loads, float4-wise maxima parked in LDS, then thread r reduces row r.
template <bool SIGMOID>
__global__ __launch_bounds__(256) void ers_rowmax_kernel(const float* __restrict__ x, int64_t rows, int C,
                                                          float* __restrict__ m, int64_t A, double* __restrict__ sums,
                                                          int which) {
    extern __shared__ float part[];  // [256 * C/4]
    __shared__ double red[4];
    const int C4 = C >> 2;
    const int64_t r0 = (int64_t)blockIdx.x * 256;
    const int nrows = (int)imin64(256, rows - r0);
    const float4* src = reinterpret_cast<const float4*>(x + r0 * C);
    for (int i = threadIdx.x; i < nrows * C4; i += 256) {
        const float4 v = src[i];
        part[i] = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    }
    __syncthreads();
    float mv = 0.f;
    const bool ok = threadIdx.x < nrows;
    if (ok) {
        mv = part[threadIdx.x * C4];
        for (int i = 1; i < C4; ++i) mv = fmaxf(mv, part[threadIdx.x * C4 + i]);
        if (SIGMOID) mv = sigmoidf_(mv);  // max_k sigmoid(x_k) == sigmoid(max_k x_k): sigmoid is monotone
        m[r0 + threadIdx.x] = mv;
    }
    // per-image sum (rows of one block may straddle two images only if A % 256 != 0: handle per row)
    const int64_t n_first = r0 / A, n_last = (r0 + nrows - 1) / A;
    if (n_first == n_last) {
        const double t = block_sum_d(ok ? (double)mv : 0.0, red);
        if (threadIdx.x == 0) atomicAdd(sums + n_first * 4 + which, t);
    } else if (ok) {
        atomicAdd(sums + ((r0 + threadIdx.x) / A) * 4 + which, (double)mv);
    }
}

// the same for a channel count that is not a multiple of 4 (e.g. 70 old classes): one thread per row
template <bool SIGMOID>
__global__ __launch_bounds__(256) void ers_rowmax_generic_kernel(const float* __restrict__ x, int64_t rows, int C,
                                                                  float* __restrict__ m, int64_t A,
                                                                  double* __restrict__ sums, int which) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* z = x + r * C;
    float mv = z[0];
    for (int i = 1; i < C; ++i) mv = fmaxf(mv, z[i]);
    if (SIGMOID) mv = sigmoidf_(mv);
    m[r] = mv;
    atomicAdd(sums + (r / A) * 4 + which, (double)mv);
}

// stage 2: sum of squared deviations from the (f64) mean
__global__ __launch_bounds__(256) void ers_var_kernel(const float* __restrict__ m, int64_t A, double* __restrict__ sums,
                                                       int which) {
    __shared__ double red[4];
    const int n = blockIdx.y;
    const double mean = sums[n * 4 + which] / (double)A;
    double s = 0;
    for (int64_t a = blockIdx.x * 256ll + threadIdx.x; a < A; a += (int64_t)gridDim.x * 256) {
        const double d = (double)m[n * A + a] - mean;
        s += d * d;
    }
    const double t = block_sum_d(s, red);
    if (threadIdx.x == 0) atomicAdd(sums + n * 4 + 2 + which, t);
}

// stage 3: one block per (image, which): threshold (fp32 arithmetic like torch: mean + 2*std), strict '>'
// mask and ascending index compaction.
__global__ __launch_bounds__(1024) void ers_compact_kernel(const float* __restrict__ m_c, const float* __restrict__ m_b,
                                                            int64_t A, const double* __restrict__ sums,
                                                            uint8_t* __restrict__ mask_c, uint8_t* __restrict__ mask_b,
                                                            int64_t* __restrict__ idx_c, int64_t* __restrict__ idx_b,
                                                            int32_t* __restrict__ counts, float* __restrict__ thr_out) {
    const int n = blockIdx.x, which = blockIdx.y;
    const float* m = (which ? m_b : m_c) + n * A;
    uint8_t* mask = (which ? mask_b : mask_c) + n * A;
    int64_t* idx = (which ? idx_b : idx_c) + n * A;
    const double mean = sums[n * 4 + which] / (double)A;
    const double var = A > 1 ? sums[n * 4 + 2 + which] / (double)(A - 1) : NAN;   // unbiased (torch default)
    const float thr = (float)mean + 2.0f * (float)sqrt(var);
    __shared__ int wsum[16];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t a0 = 0; a0 < A; a0 += 1024) {
        const int64_t a = a0 + threadIdx.x;
        const bool sel = a < A && m[a] > thr;
        if (a < A) mask[a] = sel ? 1 : 0;
        const unsigned long long bal = __ballot(sel);
        const int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int i = 0; i < w; ++i) off += wsum[i];
        if (sel) idx[off + within] = a;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int i = 0; i < 16; ++i) t += wsum[i];
            base += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counts[n * 2 + which] = base;
        thr_out[n * 2 + which] = thr;
    }
}

// ------------------------------------------------------------------------------------------------
// anchors (anchor_generator.py:161-205,259-301): side = octave_scale*stride, centre (x*s, y*s)
// ------------------------------------------------------------------------------------------------
struct Levels5 {
    int n;
    int h[ERD_MAX_SEG], w[ERD_MAX_SEG], s[ERD_MAX_SEG];
    int64_t off[ERD_MAX_SEG + 1];
};

__global__ void anchors_kernel(float4* __restrict__ anchors, Levels5 lv, int octave) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= lv.off[lv.n]) return;
    int l = 0;
    while (l < lv.n - 1 && i >= lv.off[l + 1]) ++l;
    const int64_t r = i - lv.off[l];
    const int y = r / lv.w[l], x = r % lv.w[l];
    const float half = 0.5f * (float)(lv.s[l] * octave);
    const float sx = (float)x * (float)lv.s[l], sy = (float)y * (float)lv.s[l];
    anchors[i] = make_float4(sx - half, sy - half, sx + half, sy + half);
}

// ------------------------------------------------------------------------------------------------
// ATSS (atss_assigner.py:74-254).  Block = one (gt, image).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float iou_xyxy(const float4 a, const float4 b, float eps) {
    const float area1 = (a.z - a.x) * (a.w - a.y);
    const float area2 = (b.z - b.x) * (b.w - b.y);
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
    const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
    const float ov = w * h;
    const float uni = fmaxf(area1 + area2 - ov, eps);
    return ov / uni;
}

constexpr int ATSS_MAXC = ERD_MAX_SEG * 16;  // topk <= 16

__global__ __launch_bounds__(256) void atss_candidates_kernel(const float4* __restrict__ anchors,
                                                              const uint8_t* __restrict__ valid, Levels5 lv, int64_t A,
                                                              const float4* __restrict__ gt_boxes,
                                                              const int32_t* __restrict__ gt_off, int topk,
                                                              unsigned long long* __restrict__ best) {
    const int n = blockIdx.y;
    const int g0 = gt_off[n], G = gt_off[n + 1] - g0;
    const int g = blockIdx.x;
    if (g >= G) return;
    const float4 gt = gt_boxes[g0 + g];
    const float gcx = (gt.x + gt.z) / 2.0f, gcy = (gt.y + gt.w) / 2.0f;
    const uint8_t* vmask = valid ? valid + (int64_t)n * A : nullptr;

    __shared__ float s_d[4];
    __shared__ int s_i[4];
    __shared__ int cand[ATSS_MAXC];
    __shared__ float cand_iou[ATSS_MAXC];
    __shared__ int ncand;
    __shared__ float prev_d;
    __shared__ int prev_i;
    if (threadIdx.x == 0) ncand = 0;
    __syncthreads();

    for (int l = 0; l < lv.n; ++l) {
        const int64_t lo = lv.off[l], hi = lv.off[l + 1];
        if (threadIdx.x == 0) { prev_d = -1.f; prev_i = -1; }
        __syncthreads();
        for (int k = 0; k < topk; ++k) {
            const float pd = prev_d;
            const int pi = prev_i;
            float bd = INFINITY;
            int bi = 0x7fffffff;
            for (int64_t a = lo + threadIdx.x; a < hi; a += 256) {
                if (vmask && !vmask[a]) continue;
                const float4 an = anchors[a];
                const float dx = (an.x + an.z) / 2.0f - gcx, dy = (an.y + an.w) / 2.0f - gcy;
                const float d = sqrtf(dx * dx + dy * dy);
                // strictly after the previously selected (distance, index) pair; smallest such pair wins
                const bool after = d > pd || (d == pd && (int)a > pi);
                if (after && (d < bd || (d == bd && (int)a < bi))) { bd = d; bi = (int)a; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float od = __shfl_xor(bd, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
            }
            if ((threadIdx.x & 63) == 0) { s_d[threadIdx.x >> 6] = bd; s_i[threadIdx.x >> 6] = bi; }
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int i = 1; i < 4; ++i)
                    if (s_d[i] < s_d[0] || (s_d[i] == s_d[0] && s_i[i] < s_i[0])) { s_d[0] = s_d[i]; s_i[0] = s_i[i]; }
                if (s_i[0] != 0x7fffffff) {
                    cand[ncand] = s_i[0];
                    cand_iou[ncand] = iou_xyxy(anchors[s_i[0]], gt, 1e-6f);
                    ++ncand;
                }
                prev_d = s_d[0];
                prev_i = s_i[0];
            }
            __syncthreads();
            if (prev_i == 0x7fffffff) break;  // level exhausted (fewer than topk valid anchors)
        }
        __syncthreads();
    }
    // mean + std (unbiased) of the candidate IoUs, positives, centre-inside test, conflict resolution
    if (threadIdx.x == 0 && ncand > 0) {
        double s = 0;
        for (int i = 0; i < ncand; ++i) s += cand_iou[i];
        const double mean = s / ncand;
        double v = 0;
        for (int i = 0; i < ncand; ++i) { const double d = cand_iou[i] - mean; v += d * d; }
        const float stdv = ncand > 1 ? (float)sqrt(v / (ncand - 1)) : NAN;
        const float thr = (float)mean + stdv;
        for (int i = 0; i < ncand; ++i) {
            const float4 an = anchors[cand[i]];
            const float cx = (an.x + an.z) / 2.0f, cy = (an.y + an.w) / 2.0f;
            const float mn = fminf(fminf(cx - gt.x, cy - gt.y), fminf(gt.z - cx, gt.w - cy));
            if (cand_iou[i] >= thr && mn > 0.01f) {
                // max IoU wins; ties -> lowest gt index
                const unsigned long long key =
                    ((unsigned long long)__float_as_uint(cand_iou[i]) << 32) | (unsigned long long)(0xffffffffu - (unsigned)g);
                atomicMax(best + (int64_t)n * A + cand[i], key);
            }
        }
    }
}

__global__ __launch_bounds__(256) void atss_targets_kernel(const unsigned long long* __restrict__ best,
                                                           const uint8_t* __restrict__ valid, int64_t A,
                                                           const float4* __restrict__ gt_boxes,
                                                           const int64_t* __restrict__ gt_labels,
                                                           const int32_t* __restrict__ gt_off, int num_classes,
                                                           int64_t* __restrict__ labels, float* __restrict__ lw,
                                                           float4* __restrict__ bt, int32_t* __restrict__ num_pos) {
    const int n = blockIdx.y;
    const int64_t a = blockIdx.x * 256ll + threadIdx.x;
    bool pos = false;
    if (a < A) {
        const int64_t i = (int64_t)n * A + a;
        const unsigned long long k = best[i];
        const bool v = valid ? valid[i] != 0 : true;
        if (k != 0ull) {
            const int g = gt_off[n] + (int)(0xffffffffu - (unsigned)(k & 0xffffffffull));
            labels[i] = gt_labels[g];
            lw[i] = 1.0f;
            bt[i] = gt_boxes[g];
            pos = true;
        } else {
            labels[i] = num_classes;
            lw[i] = v ? 1.0f : 0.0f;
            bt[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const unsigned long long bal = __ballot(pos);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(num_pos + n, (int)__popcll(bal));
}

// ------------------------------------------------------------------------------------------------
// supervised losses on the new classes (gfl_head_increment_erd.py:225-322)
// ------------------------------------------------------------------------------------------------
struct SoftmaxInt {  // softmax over 17 bins and its expectation (Integral, gfl_head.py:29-62)
    float p[17];
    float d;
    float lse;
};
__device__ __forceinline__ void softmax17(const float* z, SoftmaxInt& r) {
    float mx = z[0];
#pragma unroll
    for (int j = 1; j < 17; ++j) mx = fmaxf(mx, z[j]);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 17; ++j) { r.p[j] = expf(z[j] - mx); s += r.p[j]; }
    const float inv = 1.0f / s;
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 17; ++j) { r.p[j] *= inv; d += r.p[j] * (float)j; }
    r.d = d;
    r.lse = mx + logf(s);
}

// d/da max(a,b): torch.maximum splits the gradient on ties
__device__ __forceinline__ float dmax_a(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float dmin_a(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

// loss = 1 - giou(pred, tgt) and its gradient wrt pred (x1,y1,x2,y2); eps as GIoULoss (1e-6)
__device__ __forceinline__ float giou_loss_grad(const float4 p, const float4 t, float eps, float* g /*[4] or null*/,
                                                float* iou_out) {
    const float pw = p.z - p.x, ph = p.w - p.y;
    const float area1 = pw * ph, area2 = (t.z - t.x) * (t.w - t.y);
    const float ltx = fmaxf(p.x, t.x), lty = fmaxf(p.y, t.y), rbx = fminf(p.z, t.z), rby = fminf(p.w, t.w);
    const float w0 = rbx - ltx, h0 = rby - lty;
    const float w = fmaxf(w0, 0.f), h = fmaxf(h0, 0.f);
    const float ov = w * h;
    const float uni0 = area1 + area2 - ov;
    const float uni = fmaxf(uni0, eps);
    const float iou = ov / uni;
    const float ex1 = fminf(p.x, t.x), ey1 = fminf(p.y, t.y), ex2 = fmaxf(p.z, t.z), ey2 = fmaxf(p.w, t.w);
    const float ew0 = ex2 - ex1, eh0 = ey2 - ey1;
    const float ew = fmaxf(ew0, 0.f), eh = fmaxf(eh0, 0.f);
    const float ea0 = ew * eh;
    const float ea = fmaxf(ea0, eps);
    const float giou = iou - (ea - uni) / ea;
    if (iou_out) *iou_out = iou;
    if (g) {
        // forward-mode per coordinate q in {x1,y1,x2,y2}
        const float cw = w0 >= 0.f ? 1.f : 0.f, chh = h0 >= 0.f ? 1.f : 0.f;      // clamp(min=0) passes grad at >=
        const float cew = ew0 >= 0.f ? 1.f : 0.f, ceh = eh0 >= 0.f ? 1.f : 0.f;
        const float cu = uni0 > eps ? 1.f : (uni0 == eps ? 0.5f : 0.f);
        const float cea = ea0 > eps ? 1.f : (ea0 == eps ? 0.5f : 0.f);
        // d area1
        const float da1[4] = {-ph, -pw, ph, pw};
        // d w, d h (through lt/rb selections)
        const float dw[4] = {-dmax_a(p.x, t.x) * cw, 0.f, dmin_a(p.z, t.z) * cw, 0.f};
        const float dh[4] = {0.f, -dmax_a(p.y, t.y) * chh, 0.f, dmin_a(p.w, t.w) * chh};
        const float dew[4] = {-dmin_a(p.x, t.x) * cew, 0.f, dmax_a(p.z, t.z) * cew, 0.f};
        const float deh[4] = {0.f, -dmin_a(p.y, t.y) * ceh, 0.f, dmax_a(p.w, t.w) * ceh};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float dov = dw[q] * h + w * dh[q];
            const float duni = (da1[q] - dov) * cu;
            const float diou = (dov * uni - ov * duni) / (uni * uni);
            const float dea = (dew[q] * eh + ew * deh[q]) * cea;
            // giou = iou - 1 + uni/ea
            const float dgiou = diou + (duni * ea - uni * dea) / (ea * ea);
            g[q] = -dgiou;
        }
    }
    return 1.0f - giou;
}

__device__ __forceinline__ int level_of(const Levels5& lv, int64_t a) {
    int l = 0;
    while (l < lv.n - 1 && a >= lv.off[l + 1]) ++l;
    return l;
}

// Forward: one wave per anchor row group.  Thread-per-anchor over the dense QFL part would make
// strided row reads; instead a block takes 64 consecutive anchors, stages their cls rows through LDS
// with coalesced float4 loads, and thread t (<64) finishes anchor t.  Positives (rare) do the box work.
constexpr int GL_ROWS = 64;

template <bool BWD>
__global__ __launch_bounds__(256) void gfl_losses_kernel(const float* __restrict__ cls, const float* __restrict__ bbox,
                                                          const float4* __restrict__ anchors,
                                                          const int64_t* __restrict__ labels,
                                                          const float* __restrict__ lweights,
                                                          const float4* __restrict__ btargets, Levels5 lv, int64_t A,
                                                          int c_old, int c_all, float* __restrict__ score_ws,
                                                          float* __restrict__ wt_ws, double* __restrict__ out_sums,
                                                          const float* __restrict__ coef, float* __restrict__ dcls,
                                                          float* __restrict__ dbbox) {
    extern __shared__ float sm[];  // [GL_ROWS][c_all] logits (then grads)
    const int n = blockIdx.y;
    const int64_t a0 = (int64_t)blockIdx.x * GL_ROWS;
    const int nrows = (int)imin64(GL_ROWS, A - a0);
    const int C = c_all, C4 = C >> 2, cn = c_all - c_old;
    const float4* src = reinterpret_cast<const float4*>(cls + ((int64_t)n * A + a0) * C);
    for (int i = threadIdx.x; i < nrows * C4; i += 256) reinterpret_cast<float4*>(sm)[i] = src[i];
    __syncthreads();

    // 4 threads per anchor row: thread q of row r handles new channels q, q+4, ...
    const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
    const bool ok = r < nrows;
    const int64_t a = a0 + r;
    const int64_t gi = (int64_t)n * A + a;
    int lab = 0;
    float lw = 0.f;
    int l = 0;
    if (ok) {
        lab = (int)labels[gi];
        if (lab == c_all) lab = cn;      // bg label C_all -> C_new (:270-271)
        lw = lweights[gi];
        l = level_of(lv, a);
    }
    const bool is_pos = ok && lab >= 0 && lab < cn;
    float* row = sm + r * C;

    // ---- weight_targets = max_k sigmoid(cls_new) (detached) --------------------------------------
    float mx = -INFINITY;
    if (ok)
        for (int k = q; k < cn; k += 4) mx = fmaxf(mx, row[c_old + k]);
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    const float wt = is_pos ? sigmoidf_(mx) : 0.f;

    // ---- positives: decode, IoU score, GIoU, DFL (thread q == side q of the box) -------------------
    float score = 0.f, giou_l = 0.f, dfl_l = 0.f;
    float gd[4] = {0.f, 0.f, 0.f, 0.f};  // dloss_giou / d distance (only q==0 keeps all four)
    SoftmaxInt sx;
    float ycorner = 0.f;
    if (is_pos) {
        const float stride = (float)lv.s[l];
        const float4 an = anchors[a];
        const float cx = ((an.z + an.x) / 2.0f) / stride, cy = ((an.w + an.y) / 2.0f) / stride;
        float z[17];
        const float* zb = bbox + gi * 68 + q * 17;
#pragma unroll
        for (int j = 0; j < 17; ++j) z[j] = zb[j];
        softmax17(z, sx);
        const float d0 = __shfl(sx.d, (threadIdx.x & ~3) + 0, 64), d1 = __shfl(sx.d, (threadIdx.x & ~3) + 1, 64);
        const float d2 = __shfl(sx.d, (threadIdx.x & ~3) + 2, 64), d3 = __shfl(sx.d, (threadIdx.x & ~3) + 3, 64);
        const float4 pred = make_float4(cx - d0, cy - d1, cx + d2, cy + d3);
        const float4 bt = btargets[gi];
        const float4 tg = make_float4(bt.x / stride, bt.y / stride, bt.z / stride, bt.w / stride);
        float iou;
        giou_l = giou_loss_grad(pred, tg, 1e-6f, BWD ? gd : nullptr, &iou);
        score = iou;
        // bbox2distance(..., max_dis=16, eps=.1): clamp to [0, 15.9]
        const float raw = q == 0 ? cx - tg.x : (q == 1 ? cy - tg.y : (q == 2 ? tg.z - cx : tg.w - cy));
        ycorner = fminf(fmaxf(raw, 0.f), 16.0f - 0.1f);
        const int tl = (int)ycorner;            // .long() truncation of a non-negative value
        const float wl = (float)(tl + 1) - ycorner, wr = ycorner - (float)tl;
        dfl_l = (sx.lse - z[tl]) * wl + (sx.lse - z[tl + 1]) * wr;
        if (BWD) {
            const float cb = coef[4 * l + 1] * wt, cd = coef[4 * l + 2] * wt;
            // dgiou/dpred -> d distance: x1=cx-d0, y1=cy-d1, x2=cx+d2, y2=cy+d3
            const float gq = (q == 0 ? -gd[0] : (q == 1 ? -gd[1] : (q == 2 ? gd[2] : gd[3]))) * cb;
            float* gb = dbbox + gi * 68 + q * 17;
#pragma unroll
            for (int j = 0; j < 17; ++j) {
                float gz = gq * sx.p[j] * ((float)j - sx.d);                       // through Integral
                gz += cd * (sx.p[j] - (j == tl ? wl : 0.f) - (j == tl + 1 ? wr : 0.f));  // DFL
                gb[j] = gz;
            }
        }
    } else if (BWD && ok) {
        float* gb = dbbox + gi * 68 + q * 17;
#pragma unroll
        for (int j = 0; j < 17; ++j) gb[j] = 0.f;
    }
    if (!BWD && ok && q == 0) { score_ws[gi] = score; wt_ws[gi] = wt; }
    if (BWD && ok) score = score_ws[gi];

    // ---- QFL over the new channels (gfocal_loss.py:12-53) ----------------------------------------
    float qsum = 0.f;
    if (ok) {
        const float cq = BWD ? coef[4 * l] * lw : 0.f;
        for (int k = q; k < cn; k += 4) {
            const float x = row[c_old + k];
            const float sg = sigmoidf_(x);
            const float sp = softplusf_(x);
            if (is_pos && k == lab) {
                const float bce = fmaxf(x, 0.f) - x * score + log1pf(expf(-fabsf(x)));
                const float u = score - sg;
                qsum += bce * (u * u);
                if (BWD) row[c_old + k] = cq * (-(u * u * u) - 2.0f * bce * u * sg * (1.0f - sg));
            } else {
                qsum += sp * (sg * sg);
                if (BWD) row[c_old + k] = cq * (sg * sg * sg + 2.0f * sp * sg * sg * (1.0f - sg));
            }
        }
        if (BWD)
            for (int k = q; k < c_old; k += 4) row[k] = 0.f;   // old channels: filled by the distillation bwd
    }
    if (BWD) {
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(dcls + ((int64_t)n * A + a0) * C);
        for (int i = threadIdx.x; i < nrows * C4; i += 256) dst[i] = reinterpret_cast<float4*>(sm)[i];
        return;
    }
    // ---- forward reductions: per level {qfl, giou*w, dfl*w, w}: LDS partials, one global atomic per block ----
    qsum += __shfl_xor(qsum, 1, 64);
    qsum += __shfl_xor(qsum, 2, 64);
    dfl_l += __shfl_xor(dfl_l, 1, 64);
    dfl_l += __shfl_xor(dfl_l, 2, 64);
    __shared__ double lsum[ERD_MAX_SEG * 4];
    if (threadIdx.x < ERD_MAX_SEG * 4) lsum[threadIdx.x] = 0.0;
    __syncthreads();
    if (ok && q == 0) {
        atomicAdd(&lsum[l * 4 + 0], (double)(qsum * lw));
        if (is_pos) {
            atomicAdd(&lsum[l * 4 + 1], (double)(giou_l * wt));
            atomicAdd(&lsum[l * 4 + 2], (double)(dfl_l * wt));
            atomicAdd(&lsum[l * 4 + 3], (double)wt);
        }
    }
    __syncthreads();
    if (threadIdx.x < lv.n * 4 && lsum[threadIdx.x] != 0.0) atomicAdd(out_sums + threadIdx.x, lsum[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// distillation
// ------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void l2_distill_kernel(const float* __restrict__ s_cls, const float* __restrict__ t_cls,
                                                          const int64_t* __restrict__ idx, const int32_t* __restrict__ counts,
                                                          int64_t A, int c_s, int c_t, int c_old,
                                                          double* __restrict__ sums, const float* __restrict__ coef,
                                                          float* __restrict__ dcls) {
    __shared__ double red[4];
    const int n = blockIdx.y;
    const int K = counts[n * 2 + 0];
    const int64_t total = (int64_t)K * c_old;
    double acc = 0;
    const float cf = BWD ? coef[n] : 0.f;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t a = idx[(int64_t)n * A + i / c_old];
        const int k = (int)(i % c_old);
        const float d = s_cls[((int64_t)n * A + a) * c_s + k] - t_cls[((int64_t)n * A + a) * c_t + k];
        if (BWD) dcls[((int64_t)n * A + a) * c_s + k] += cf * 2.0f * d;
        else acc += (double)(d * d);
    }
    if (!BWD) {
        const double t = block_sum_d(acc, red);
        if (threadIdx.x == 0 && t != 0.0) atomicAdd(sums + n, t);
    }
}

constexpr int NMS_LDS_K = 2560;
#ifndef ERD_NMS_LDS
#define ERD_NMS_LDS 1      // 0: timing probe (tools/build_probe.sh), the greedy pass always out of the global workspace
#endif
// teacher boxes for NMS: pixel-unit centres + stride-unit distances (D8, :189-192), score = max sigmoid,
// id = first argmax.  ws layout per image: boxes[A][4], scores[A], ids[A] (int), order[A] (int), removed[A]
__global__ __launch_bounds__(1024) void distill_nms_kernel(const float* __restrict__ t_cls, const float* __restrict__ t_bbox,
                                                            const float4* __restrict__ anchors,
                                                            const int64_t* __restrict__ idx_bbox,
                                                            const int32_t* __restrict__ counts, int64_t A, int c_t,
                                                            float iou_thr, uint8_t* __restrict__ keep_mask,
                                                            int32_t* __restrict__ keep_count, float* __restrict__ ws) {
    const int n = blockIdx.x;
    const int K = counts[n * 2 + 1];
    float4* boxes = reinterpret_cast<float4*>(ws + (int64_t)n * A * 8);
    float* scores = reinterpret_cast<float*>(boxes + A);
    int* ids = reinterpret_cast<int*>(scores + A);
    int* order = ids + A;
    int* removed = order + A;
    const int64_t* idx = idx_bbox + (int64_t)n * A;
    uint8_t* km = keep_mask + (int64_t)n * A;
    for (int64_t a = threadIdx.x; a < A; a += 1024) km[a] = 0;
    __shared__ float red[16];
    __shared__ int s_cnt;
    __shared__ float4 s_box[NMS_LDS_K];        // 40 + 10 + 10 + 2.5 KB: the greedy pass of K <= NMS_LDS_K boxes runs out of LDS
    __shared__ float s_sc[NMS_LDS_K];
    __shared__ int s_anchor[NMS_LDS_K];
    __shared__ uint8_t s_rem[NMS_LDS_K];
    float lmax = -INFINITY;
    for (int i = threadIdx.x; i < K; i += 1024) {
        const int64_t a = idx[i];
        const float* zc = t_cls + ((int64_t)n * A + a) * c_t;
        float mx = zc[0];
        int am = 0;
        for (int k = 1; k < c_t; ++k)
            if (zc[k] > mx) { mx = zc[k]; am = k; }
        float d[4];
        for (int q = 0; q < 4; ++q) {
            float z[17];
            const float* zb = t_bbox + ((int64_t)n * A + a) * 68 + q * 17;
#pragma unroll
            for (int j = 0; j < 17; ++j) z[j] = zb[j];
            SoftmaxInt sx;
            softmax17(z, sx);
            d[q] = sx.d;
        }
        const float4 an = anchors[a];
        const float cx = (an.z + an.x) / 2.0f, cy = (an.w + an.y) / 2.0f;
        const float4 b = make_float4(cx - d[0], cy - d[1], cx + d[2], cy + d[3]);
        boxes[i] = b;
        scores[i] = sigmoidf_(mx);
        ids[i] = am;
        removed[i] = 0;
        lmax = fmaxf(lmax, fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
    }
    lmax = erd::wave_max(lmax);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lmax;
    __syncthreads();
    float maxc = red[0];
    for (int i = 1; i < 16; ++i) maxc = fmaxf(maxc, red[i]);
    const float offs = maxc + 1.0f;   // boxes.max() + 1
    __syncthreads();
    // class offsets in fp32 (batched_nms), then rank sort: score desc, stable
    for (int i = threadIdx.x; i < K; i += 1024) {
        const float o = (float)ids[i] * offs;
        float4 b = boxes[i];
        b.x += o; b.y += o; b.z += o; b.w += o;
        boxes[i] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_cnt = 0;
    if (ERD_NMS_LDS && K <= NMS_LDS_K) {
        // The greedy pass is one barrier per surviving box with the suppression flags read back right behind it: in global memory
        // every step is an L2 round trip (~0.6 us x K: 500 of the launch's 520 us at the step's K = 800), in LDS a tenth of that.
        // Boxes in rank order, same arithmetic, same order of decisions: the keep set is the global path's bit for bit.
        for (int i = threadIdx.x; i < K; i += 1024) s_sc[i] = scores[i];
        __syncthreads();
        for (int i = threadIdx.x; i < K; i += 1024) {
            const float si = s_sc[i];
            int rank = 0;
            for (int j = 0; j < K; ++j) {
                const float sj = s_sc[j];
                rank += (sj > si || (sj == si && j < i)) ? 1 : 0;
            }
            s_box[rank] = boxes[i];
            s_anchor[rank] = (int)idx[i];
            s_rem[rank] = 0;
        }
        __syncthreads();
        for (int oi = 0; oi < K; ++oi) {
            if (s_rem[oi]) continue;          // uniform: every thread reads the same flag after the last barrier
            if (threadIdx.x == 0) { km[s_anchor[oi]] = 1; ++s_cnt; }
            const float4 bi = s_box[oi];
            const float ai = (bi.z - bi.x) * (bi.w - bi.y);
            for (int oj = oi + 1 + threadIdx.x; oj < K; oj += 1024) {
                if (s_rem[oj]) continue;
                const float4 bj = s_box[oj];
                const float w = fmaxf(fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x), 0.f);
                const float h = fmaxf(fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y), 0.f);
                const float inter = w * h;
                const float aj = (bj.z - bj.x) * (bj.w - bj.y);
                const float iou = inter / (ai + aj - inter);
                if (iou > iou_thr) s_rem[oj] = 1;
            }
            __syncthreads();
        }
    } else {
        for (int i = threadIdx.x; i < K; i += 1024) {
            const float si = scores[i];
            int rank = 0;
            for (int j = 0; j < K; ++j) {
                const float sj = scores[j];
                rank += (sj > si || (sj == si && j < i)) ? 1 : 0;
            }
            order[rank] = i;
        }
        __syncthreads();
        for (int oi = 0; oi < K; ++oi) {
            const int i = order[oi];
            if (removed[i]) continue;         // uniform: every thread reads the same flag after the last barrier
            if (threadIdx.x == 0) { km[idx[i]] = 1; ++s_cnt; }
            const float4 bi = boxes[i];
            const float ai = (bi.z - bi.x) * (bi.w - bi.y);
            for (int oj = oi + 1 + threadIdx.x; oj < K; oj += 1024) {
                const int j = order[oj];
                if (removed[j]) continue;
                const float4 bj = boxes[j];
                const float w = fmaxf(fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x), 0.f);
                const float h = fmaxf(fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y), 0.f);
                const float inter = w * h;
                const float aj = (bj.z - bj.x) * (bj.w - bj.y);
                const float iou = inter / (ai + aj - inter);
                if (iou > iou_thr) removed[j] = 1;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) keep_count[n] = s_cnt;
}

// KD-KL with temperature over kept anchors.  4 threads per anchor (one per box side).
template <bool BWD>
__global__ __launch_bounds__(256) void kd_kl_kernel(const float* __restrict__ s_bbox, const float* __restrict__ t_bbox,
                                                     const float* __restrict__ s_cls, const uint8_t* __restrict__ keep,
                                                     int64_t A, int c_s, int c_old, float T, double* __restrict__ sums,
                                                     const float* __restrict__ coef, float* __restrict__ dbbox) {
    __shared__ double red[4];
    const int n = blockIdx.y;
    const int64_t a = blockIdx.x * 64ll + (threadIdx.x >> 2);
    const int q = threadIdx.x & 3;
    double contrib = 0;
    if (a < A && keep[(int64_t)n * A + a]) {
        const int64_t gi = (int64_t)n * A + a;
        // weight = max_k sigmoid(student old-class logits) (detached, :217-218)
        const float* zc = s_cls + gi * c_s;
        float mx = -INFINITY;
        for (int k = q; k < c_old; k += 4) mx = fmaxf(mx, zc[k]);
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        const float wt = sigmoidf_(mx);
        float zs[17], zt[17];
        const float* ps = s_bbox + gi * 68 + q * 17;
        const float* pt = t_bbox + gi * 68 + q * 17;
        float ms = -INFINITY, mt = -INFINITY;
#pragma unroll
        for (int j = 0; j < 17; ++j) {
            zs[j] = ps[j] / T;
            zt[j] = pt[j] / T;
            ms = fmaxf(ms, zs[j]);
            mt = fmaxf(mt, zt[j]);
        }
        float ss = 0.f, st = 0.f;
#pragma unroll
        for (int j = 0; j < 17; ++j) { ss += expf(zs[j] - ms); st += expf(zt[j] - mt); }
        const float lss = ms + logf(ss), lst = mt + logf(st);
        float kl = 0.f;
        float* gb = BWD ? dbbox + gi * 68 + q * 17 : nullptr;
        const float cf = BWD ? coef[n] * wt * T / 17.0f : 0.f;
#pragma unroll
        for (int j = 0; j < 17; ++j) {
            const float logp = zs[j] - lss, logt = zt[j] - lst;
            const float t = expf(logt);
            if (BWD) gb[j] += cf * (expf(logp) - t);
            else kl += t > 0.f ? t * (logt - logp) : 0.f;
        }
        contrib = (double)(kl / 17.0f * (T * T) * wt);
    }
    if (!BWD) {
        const double t = block_sum_d(contrib, red);
        if (threadIdx.x == 0 && t != 0.0) atomicAdd(sums + n, t);
    }
}

// losses[3*nlvl + 2N] (loss_cls[l], loss_bbox[l], loss_dfl[l], loss_dist_cls[n], loss_dist_bbox[n]) and the
// backward coefficients coef: [4l+0]=d total/d qfl_sum_l, [4l+1]=.../d giou_sum_l, [4l+2]=.../d dfl_sum_l,
// [4*nlvl+n] = d/d l2_sum_n, [4*nlvl+N+n] = d/d kd_sum_n.  `up` = upstream grads of the loss vector.
__global__ void finalize_kernel(const double* __restrict__ sums /*[nlvl][4]*/, const float* __restrict__ avg /*[2]*/,
                                const double* __restrict__ l2s, const double* __restrict__ kds,
                                const int32_t* __restrict__ counts, int nlvl, int N, int c_old, float w_dist,
                                float lw_cls, float lw_bbox, float lw_dfl, float lw_ld, const float* __restrict__ up,
                                float* __restrict__ losses, float* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float eps = 1.1920928955078125e-07f;  // finfo(float32).eps (losses/utils.py:59-61)
    const float a1 = avg[0];
    const float a2 = fmaxf(avg[1], 1.0f);       // .clamp_(min=1) (:406)
    for (int l = 0; l < nlvl; ++l) {
        const float q = (float)sums[l * 4 + 0], b = (float)sums[l * 4 + 1], d = (float)sums[l * 4 + 2];
        if (losses) {
            losses[l] = lw_cls * (q / (a1 + eps));
            losses[nlvl + l] = lw_bbox * (b / (1.0f + eps)) / a2;
            losses[2 * nlvl + l] = lw_dfl * (d / (4.0f + eps)) / a2;
        }
        if (coef) {
            coef[4 * l + 0] = lw_cls / (a1 + eps) * (up ? up[l] : 1.f);
            coef[4 * l + 1] = lw_bbox / (1.0f + eps) / a2 * (up ? up[nlvl + l] : 1.f);
            coef[4 * l + 2] = lw_dfl / (4.0f + eps) / a2 * (up ? up[2 * nlvl + l] : 1.f);
            coef[4 * l + 3] = 0.f;
        }
    }
    for (int n = 0; n < N; ++n) {
        const int kc = counts[n * 2 + 0];
        const float denom = (float)kc * (float)c_old;
        const float l2 = kc > 0 ? (float)l2s[n] / denom : 0.f;   // D10: empty selection -> 0 (reference: NaN)
        if (losses) {
            losses[3 * nlvl + n] = w_dist * l2;
            losses[3 * nlvl + N + n] = w_dist * (lw_ld * ((float)kds[n] / (4.0f + eps)));
        }
        if (coef) {
            coef[4 * nlvl + n] = kc > 0 ? w_dist / denom * (up ? up[3 * nlvl + n] : 1.f) : 0.f;
            coef[4 * nlvl + N + n] = w_dist * lw_ld / (4.0f + eps) * (up ? up[3 * nlvl + N + n] : 1.f);
        }
    }
}

Levels5 make_levels(const int64_t* lvl_off, const int* strides, int nlvl, const int* hs, const int* ws) {
    Levels5 lv;
    lv.n = nlvl;
    for (int i = 0; i < ERD_MAX_SEG; ++i) { lv.h[i] = lv.w[i] = lv.s[i] = 0; lv.off[i] = 0; }
    for (int i = 0; i < nlvl; ++i) {
        lv.off[i] = lvl_off[i];
        if (strides) lv.s[i] = strides[i];
        if (hs) lv.h[i] = hs[i];
        if (ws) lv.w[i] = ws[i];
    }
    for (int i = nlvl; i <= ERD_MAX_SEG; ++i) lv.off[i] = lvl_off[nlvl];
    return lv;
}

}  // namespace

extern "C" int erd_ers_select(const float* cls, const float* bbox, int N, int64_t A, int Ccls, int Cbox,
                              uint8_t* mask_cls, uint8_t* mask_bbox, int64_t* idx_cls, int64_t* idx_bbox,
                              int32_t* counts, float* thr, double* ws, erd_stream_t stream) {
    ERD_REQUIRE(cls && bbox && mask_cls && mask_bbox && idx_cls && idx_bbox && counts && thr && ws, "ers: null");
    ERD_REQUIRE(Ccls > 0 && Cbox > 0 && N > 0 && A > 0, "ers: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    // ws: double sums[N][4] | float m_c[N*A] | float m_b[N*A]
    double* sums = ws;
    float* m_c = reinterpret_cast<float*>(ws + (size_t)N * 4);
    float* m_b = m_c + (size_t)N * A;
    ERD_ZERO_ASYNC(sums, sizeof(double) * 4 * N, st);
    const int64_t rows = (int64_t)N * A;
    const unsigned nb = (unsigned)((rows + 255) / 256);
    if (Ccls % 4 == 0)
        hipLaunchKernelGGL(ers_rowmax_kernel<true>, dim3(nb), dim3(256), 256 * (Ccls / 4) * sizeof(float), st, cls, rows,
                           Ccls, m_c, A, sums, 0);
    else
        hipLaunchKernelGGL(ers_rowmax_generic_kernel<true>, dim3(nb), dim3(256), 0, st, cls, rows, Ccls, m_c, A, sums, 0);
    if (Cbox % 4 == 0)
        hipLaunchKernelGGL(ers_rowmax_kernel<false>, dim3(nb), dim3(256), 256 * (Cbox / 4) * sizeof(float), st, bbox,
                           rows, Cbox, m_b, A, sums, 1);
    else
        hipLaunchKernelGGL(ers_rowmax_generic_kernel<false>, dim3(nb), dim3(256), 0, st, bbox, rows, Cbox, m_b, A, sums, 1);
    const unsigned vb = (unsigned)std::min<int64_t>(64, (A + 255) / 256);
    hipLaunchKernelGGL(ers_var_kernel, dim3(vb, N), dim3(256), 0, st, m_c, A, sums, 0);
    hipLaunchKernelGGL(ers_var_kernel, dim3(vb, N), dim3(256), 0, st, m_b, A, sums, 1);
    hipLaunchKernelGGL(ers_compact_kernel, dim3(N, 2), dim3(1024), 0, st, m_c, m_b, A, sums, mask_cls, mask_bbox,
                       idx_cls, idx_bbox, counts, thr);
    return erd::check_launch("ers_select");
}

extern "C" int erd_grid_anchors(float* anchors, const int* hs, const int* ws, const int* strides, int nlvl,
                                int octave_scale, erd_stream_t stream) {
    ERD_REQUIRE(anchors && hs && ws && strides && nlvl >= 1 && nlvl <= ERD_MAX_SEG, "anchors: bad args");
    int64_t off[ERD_MAX_SEG + 1];
    off[0] = 0;
    for (int i = 0; i < nlvl; ++i) off[i + 1] = off[i] + (int64_t)hs[i] * ws[i];
    const Levels5 lv = make_levels(off, strides, nlvl, hs, ws);
    hipLaunchKernelGGL(anchors_kernel, dim3((unsigned)((off[nlvl] + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(anchors), lv, octave_scale);
    return erd::check_launch("grid_anchors");
}

extern "C" int erd_atss_assign(const float* anchors, const uint8_t* valid, const int64_t* lvl_off, int nlvl, int64_t A,
                               const float* gt_boxes, const int64_t* gt_labels, const int32_t* gt_off_dev,
                               int N, int max_gt, int topk, int num_classes, int64_t* labels, float* label_weights,
                               float* bbox_targets, int32_t* num_pos, void* ws, erd_stream_t stream) {
    ERD_REQUIRE(anchors && lvl_off && gt_off_dev && labels && label_weights && bbox_targets && num_pos && ws,
                "atss: null");
    ERD_REQUIRE(nlvl >= 1 && nlvl <= ERD_MAX_SEG && topk >= 1 && topk <= 16, "atss: nlvl=%d topk=%d", nlvl, topk);
    hipStream_t st = (hipStream_t)stream;
    const Levels5 lv = make_levels(lvl_off, nullptr, nlvl, nullptr, nullptr);
    unsigned long long* best = reinterpret_cast<unsigned long long*>(ws);
    ERD_ZERO_ASYNC(best, sizeof(unsigned long long) * (size_t)N * A, st);
    ERD_ZERO_ASYNC(num_pos, sizeof(int32_t) * N, st);
    if (max_gt > 0)
        hipLaunchKernelGGL(atss_candidates_kernel, dim3(max_gt, N), dim3(256), 0, st,
                           reinterpret_cast<const float4*>(anchors), valid, lv, A,
                           reinterpret_cast<const float4*>(gt_boxes), gt_off_dev, topk, best);
    hipLaunchKernelGGL(atss_targets_kernel, dim3((unsigned)((A + 255) / 256), N), dim3(256), 0, st, best, valid, A,
                       reinterpret_cast<const float4*>(gt_boxes), gt_labels, gt_off_dev, num_classes, labels,
                       label_weights, reinterpret_cast<float4*>(bbox_targets), num_pos);
    return erd::check_launch("atss_assign");
}

extern "C" int erd_gfl_losses_fwd(const float* cls, const float* bbox, const float* anchors, const int64_t* labels,
                                  const float* label_weights, const float* bbox_targets, const int64_t* lvl_off,
                                  const int* strides, int nlvl, int N, int64_t A, int c_old, int c_all,
                                  float* score_ws, float* wt_ws, double* out_sums, erd_stream_t stream) {
    ERD_REQUIRE(cls && bbox && anchors && labels && label_weights && bbox_targets && lvl_off && strides && score_ws &&
                    wt_ws && out_sums, "gfl_losses_fwd: null");
    ERD_REQUIRE(c_all % 4 == 0 && c_old >= 0 && c_old < c_all && nlvl <= ERD_MAX_SEG, "gfl_losses_fwd: channels");
    hipStream_t st = (hipStream_t)stream;
    const Levels5 lv = make_levels(lvl_off, strides, nlvl, nullptr, nullptr);
    ERD_ZERO_ASYNC(out_sums, sizeof(double) * 4 * nlvl, st);
    hipLaunchKernelGGL(gfl_losses_kernel<false>, dim3((unsigned)((A + GL_ROWS - 1) / GL_ROWS), N), dim3(256),
                       GL_ROWS * c_all * sizeof(float), st, cls, bbox, reinterpret_cast<const float4*>(anchors), labels,
                       label_weights, reinterpret_cast<const float4*>(bbox_targets), lv, A, c_old, c_all, score_ws,
                       wt_ws, out_sums, nullptr, nullptr, nullptr);
    return erd::check_launch("gfl_losses_fwd");
}

extern "C" int erd_gfl_losses_bwd(const float* cls, const float* bbox, const float* anchors, const int64_t* labels,
                                  const float* label_weights, const float* bbox_targets, const int64_t* lvl_off,
                                  const int* strides, int nlvl, int N, int64_t A, int c_old, int c_all,
                                  const float* score_ws, const float* wt_ws, const float* coef, float* dcls,
                                  float* dbbox, erd_stream_t stream) {
    ERD_REQUIRE(cls && bbox && anchors && labels && label_weights && bbox_targets && lvl_off && strides && score_ws &&
                    coef && dcls && dbbox, "gfl_losses_bwd: null");
    const Levels5 lv = make_levels(lvl_off, strides, nlvl, nullptr, nullptr);
    hipLaunchKernelGGL(gfl_losses_kernel<true>, dim3((unsigned)((A + GL_ROWS - 1) / GL_ROWS), N), dim3(256),
                       GL_ROWS * c_all * sizeof(float), (hipStream_t)stream, cls, bbox,
                       reinterpret_cast<const float4*>(anchors), labels, label_weights,
                       reinterpret_cast<const float4*>(bbox_targets), lv, A, c_old, c_all,
                       const_cast<float*>(score_ws), const_cast<float*>(wt_ws), nullptr, coef, dcls, dbbox);
    return erd::check_launch("gfl_losses_bwd");
}

extern "C" int erd_l2_distill(const float* s_cls, const float* t_cls, const int64_t* idx_cls, const int32_t* counts,
                              int N, int64_t A, int c_s, int c_t, int c_old, double* sums, erd_stream_t stream) {
    ERD_REQUIRE(s_cls && t_cls && idx_cls && counts && sums, "l2: null");
    hipStream_t st = (hipStream_t)stream;
    ERD_ZERO_ASYNC(sums, sizeof(double) * N, st);
    hipLaunchKernelGGL(l2_distill_kernel<false>, dim3(64, N), dim3(256), 0, st, s_cls, t_cls, idx_cls, counts, A, c_s,
                       c_t, c_old, sums, nullptr, nullptr);
    return erd::check_launch("l2_distill");
}

extern "C" int erd_l2_distill_bwd(const float* s_cls, const float* t_cls, const int64_t* idx_cls,
                                  const int32_t* counts, const float* coef, int N, int64_t A, int c_s, int c_t,
                                  int c_old, float* dcls, erd_stream_t stream) {
    ERD_REQUIRE(s_cls && t_cls && idx_cls && counts && coef && dcls, "l2_bwd: null");
    hipLaunchKernelGGL(l2_distill_kernel<true>, dim3(64, N), dim3(256), 0, (hipStream_t)stream, s_cls, t_cls, idx_cls,
                       counts, A, c_s, c_t, c_old, nullptr, coef, dcls);
    return erd::check_launch("l2_distill_bwd");
}

extern "C" int erd_distill_nms(const float* t_cls, const float* t_bbox, const float* anchors, const int64_t* idx_bbox,
                               const int32_t* counts, int N, int64_t A, int c_t, float iou_thr, uint8_t* keep_mask,
                               int32_t* keep_count, float* ws, size_t ws_bytes, erd_stream_t stream) {
    ERD_REQUIRE(t_cls && t_bbox && anchors && idx_bbox && counts && keep_mask && keep_count && ws, "nms: null");
    ERD_REQUIRE(ws_bytes >= (size_t)N * A * 8 * sizeof(float), "nms: workspace too small (need N*A*32 bytes)");
    hipLaunchKernelGGL(distill_nms_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, t_cls, t_bbox,
                       reinterpret_cast<const float4*>(anchors), idx_bbox, counts, A, c_t, iou_thr, keep_mask,
                       keep_count, ws);
    return erd::check_launch("distill_nms");
}

extern "C" int erd_kd_kl(const float* s_bbox, const float* t_bbox, const float* s_cls, const uint8_t* keep_mask, int N,
                         int64_t A, int c_s, int c_old, float T, double* sums, erd_stream_t stream) {
    ERD_REQUIRE(s_bbox && t_bbox && s_cls && keep_mask && sums, "kd_kl: null");
    hipStream_t st = (hipStream_t)stream;
    ERD_ZERO_ASYNC(sums, sizeof(double) * N, st);
    hipLaunchKernelGGL(kd_kl_kernel<false>, dim3((unsigned)((A + 63) / 64), N), dim3(256), 0, st, s_bbox, t_bbox, s_cls,
                       keep_mask, A, c_s, c_old, T, sums, nullptr, nullptr);
    return erd::check_launch("kd_kl");
}

extern "C" int erd_kd_kl_bwd(const float* s_bbox, const float* t_bbox, const float* s_cls, const uint8_t* keep_mask,
                             const float* coef, int N, int64_t A, int c_s, int c_old, float T, float* dbbox,
                             erd_stream_t stream) {
    ERD_REQUIRE(s_bbox && t_bbox && s_cls && keep_mask && coef && dbbox, "kd_kl_bwd: null");
    hipLaunchKernelGGL(kd_kl_kernel<true>, dim3((unsigned)((A + 63) / 64), N), dim3(256), 0, (hipStream_t)stream,
                       s_bbox, t_bbox, s_cls, keep_mask, A, c_s, c_old, T, nullptr, coef, dbbox);
    return erd::check_launch("kd_kl_bwd");
}

extern "C" int erd_loss_finalize(const double* lvl_sums, const float* avg, const double* l2_sums, const double* kd_sums,
                                 const int32_t* counts, int nlvl, int N, int c_old, float dist_loss_weight,
                                 float lw_cls, float lw_bbox, float lw_dfl, float lw_ld, const float* upstream,
                                 float* losses, float* coef, erd_stream_t stream) {
    ERD_REQUIRE(lvl_sums && avg && (N == 0 || (l2_sums && kd_sums && counts)) && (losses || coef), "finalize: null");
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lvl_sums, avg, l2_sums, kd_sums,
                       counts, nlvl, N, c_old, dist_loss_weight, lw_cls, lw_bbox, lw_dfl, lw_ld, upstream, losses, coef);
    return erd::check_launch("loss_finalize");
}

namespace {
// avg[0] = sum_n max(num_pos_n, 1) (sampling_result.py:96-100 summed, gfl_head.py:545-546);
// avg[1] = sum_l sum weight_targets (gfl_head_increment_erd.py:405)
__global__ void loss_avg_kernel(const int32_t* __restrict__ num_pos, int N, const double* __restrict__ lvl_sums,
                                int nlvl, float* __restrict__ avg) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int t = 0;
    for (int n = 0; n < N; ++n) t += num_pos[n] > 1 ? num_pos[n] : 1;
    double w = 0;
    for (int l = 0; l < nlvl; ++l) w += lvl_sums[l * 4 + 3];
    avg[0] = (float)t;
    avg[1] = (float)w;
}
}  // namespace

extern "C" int erd_loss_avg(const int32_t* num_pos, int N, const double* lvl_sums, int nlvl, float* avg,
                            erd_stream_t stream) {
    ERD_REQUIRE(num_pos && lvl_sums && avg, "loss_avg: null");
    hipLaunchKernelGGL(loss_avg_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, num_pos, N, lvl_sums, nlvl, avg);
    return erd::check_launch("loss_avg");
}
