// Stand-alone leaf operators of the GFL / ERD loss path: what a reference-side caller reaches through the registered
// loss / coder / assigner MODULES when it invokes them directly (self.loss_cls(...), coder.decode(...), assigner.assign(...)).
// The training step itself runs the fused kernels of losses.hip; these are the same formulas one operator at a time,
// row-parallel HBM streams (one thread per row, or one per element for the per-anchor dense QFL), each with its
// analytic backward.  Reductions (`weight_reduce_loss`, losses/utils.py:30-65) are a separate f64-accumulating pass.
//
// replaces (reference file:line under /root/reference/mmdet):
//   quality_focal_loss            models/losses/gfocal_loss.py:12-53
//   distribution_focal_loss       models/losses/gfocal_loss.py:143-165
//   knowledge_distillation_kl_div models/losses/kd_loss.py:12-37
//   giou_loss / bbox_overlaps     models/losses/iou_loss.py:110-126, structures/bbox/bbox_overlaps.py:13-199
//   Integral                      models/dense_heads/gfl_head.py:29-62
//   distance2bbox / bbox2distance structures/bbox/transforms.py:147-230
//   weight_reduce_loss            models/losses/utils.py:30-65
//   AssignResult fields           models/task_modules/assigners/atss_assigner.py:238-254
#include "erd_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// binary_cross_entropy_with_logits(x, t) = max(x,0) - x t + log(1 + exp(-|x|))
__device__ __forceinline__ float bce_logits(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }

// ---- QFL: rows[i] = sum_k loss[i][k]; one wave per row (C <= a few hundred) ---------------------------------------
// loss[i][k] = BCE(x, 0) * sigma^2; at (i, label_i) for 0 <= label_i < C: BCE(x, q_i) * (q_i - sigma)^2   (beta = 2)
__global__ __launch_bounds__(256) void qfl_rows_kernel(const float* __restrict__ pred, const int64_t* __restrict__ label,
                                                       const float* __restrict__ score, int64_t n, int C,
                                                       float* __restrict__ rows) {
    const int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t lb = label[i];
    const float q = score[i];
    float s = 0.f;
    for (int k = lane; k < C; k += 64) {
        const float x = pred[i * C + k];
        const float sg = sigmoidf_(x);
        if (k == lb) {
            const float d = q - sg;
            s += bce_logits(x, q) * d * d;
        } else {
            s += bce_logits(x, 0.f) * sg * sg;
        }
    }
    s = erd::wave_sum(s);
    if (lane == 0) rows[i] = s;
}

// dpred[i][k] = coef[i] * d loss[i][k] / d x
__global__ __launch_bounds__(256) void qfl_bwd_kernel(const float* __restrict__ pred, const int64_t* __restrict__ label,
                                                      const float* __restrict__ score, const float* __restrict__ coef,
                                                      int64_t n, int C, float* __restrict__ dpred) {
    const int64_t e = blockIdx.x * 256ll + threadIdx.x;
    if (e >= n * C) return;
    const int64_t i = e / C;
    const int k = (int)(e - i * C);
    const float x = pred[e];
    const float sg = sigmoidf_(x);
    float g;
    if (k == label[i]) {
        const float q = score[i];
        const float d = sg - q;                       // dBCE/dx = sigma - q ; d(q - sigma)^2/dx = 2 d sigma (1 - sigma)
        g = d * d * d + bce_logits(x, q) * 2.f * d * sg * (1.f - sg);
    } else {
        g = sg * sg * sg + bce_logits(x, 0.f) * 2.f * sg * sg * (1.f - sg);
    }
    dpred[e] = coef[i] * g;
}

// ---- rows over `nb` bins (nb <= 32): DFL, KD-KL, Integral ----------------------------------------------------------
constexpr int MAXB = 32;

__global__ __launch_bounds__(256) void dfl_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                  const float* __restrict__ coef, int64_t m, int nb,
                                                  float* __restrict__ rows, float* __restrict__ dpred) {
    const int64_t r = blockIdx.x * 256ll + threadIdx.x;
    if (r >= m) return;
    float z[MAXB];
    float mx = -INFINITY;
    for (int b = 0; b < nb; ++b) { z[b] = pred[r * nb + b]; mx = fmaxf(mx, z[b]); }
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += expf(z[b] - mx);
    const float lse = mx + logf(s);
    const float y = target[r];
    const int dl = (int)y;                            // label.long(): floor for y >= 0
    const int dr = dl + 1;
    const float wl = (float)dr - y, wr = y - (float)dl;
    if (rows) rows[r] = (lse - z[dl]) * wl + (lse - z[min(dr, nb - 1)]) * wr;
    if (dpred) {
        const float c = coef[r];
        for (int b = 0; b < nb; ++b) {
            const float p = expf(z[b] - lse);
            dpred[r * nb + b] = c * (p * (wl + wr) - (b == dl ? wl : 0.f) - (b == dr ? wr : 0.f));
        }
    }
}

// F.kl_div(log_softmax(pred / T), softmax(soft / T), reduction='none').mean(1) * T * T
__global__ __launch_bounds__(256) void kdkl_kernel(const float* __restrict__ pred, const float* __restrict__ soft,
                                                   const float* __restrict__ coef, int64_t m, int nb, float T,
                                                   float* __restrict__ rows, float* __restrict__ dpred) {
    const int64_t r = blockIdx.x * 256ll + threadIdx.x;
    if (r >= m) return;
    float zs[MAXB], zt[MAXB];
    float ms = -INFINITY, mt = -INFINITY;
    for (int b = 0; b < nb; ++b) {
        zs[b] = pred[r * nb + b] / T; zt[b] = soft[r * nb + b] / T;
        ms = fmaxf(ms, zs[b]); mt = fmaxf(mt, zt[b]);
    }
    float ss = 0.f, st = 0.f;
    for (int b = 0; b < nb; ++b) { ss += expf(zs[b] - ms); st += expf(zt[b] - mt); }
    const float lses = ms + logf(ss), lset = mt + logf(st);
    float acc = 0.f;
    for (int b = 0; b < nb; ++b) {
        const float lt = zt[b] - lset, t = expf(lt);
        acc += t > 0.f ? t * (lt - (zs[b] - lses)) : 0.f;      // xlogy convention of F.kl_div
    }
    if (rows) rows[r] = acc / (float)nb * (T * T);
    if (dpred) {
        const float c = coef[r] * T / (float)nb;               // d/dpred = (softmax(pred/T) - t) / T / nb * T^2
        for (int b = 0; b < nb; ++b) dpred[r * nb + b] = c * (expf(zs[b] - lses) - expf(zt[b] - lset));
    }
}

// Integral: y[r] = sum_b softmax(x[r])[b] * b ; dx = p_b (b - y) dy
__global__ __launch_bounds__(256) void integral_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t m,
                                                       int nb, float* __restrict__ y, float* __restrict__ dx) {
    const int64_t r = blockIdx.x * 256ll + threadIdx.x;
    if (r >= m) return;
    float z[MAXB];
    float mx = -INFINITY;
    for (int b = 0; b < nb; ++b) { z[b] = x[r * nb + b]; mx = fmaxf(mx, z[b]); }
    float s = 0.f;
    for (int b = 0; b < nb; ++b) { z[b] = expf(z[b] - mx); s += z[b]; }
    const float inv = 1.0f / s;
    float e = 0.f;
    for (int b = 0; b < nb; ++b) { z[b] *= inv; e += z[b] * (float)b; }
    if (y) y[r] = e;
    if (dx) {
        const float g = dy[r];
        for (int b = 0; b < nb; ++b) dx[r * nb + b] = z[b] * ((float)b - e) * g;
    }
}

// ---- boxes ----------------------------------------------------------------------------------------------------------
struct IouParts { float iou, giou; };
// bbox_overlaps.py:151-199 (aligned and pairwise share the arithmetic)
__device__ __forceinline__ IouParts overlaps(float4 a, float4 b, float eps) {
    const float area1 = (a.z - a.x) * (a.w - a.y), area2 = (b.z - b.x) * (b.w - b.y);
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f), h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
    const float overlap = w * h;
    const float uni = fmaxf(area1 + area2 - overlap, eps);
    IouParts r;
    r.iou = overlap / uni;
    const float ew = fmaxf(fmaxf(a.z, b.z) - fminf(a.x, b.x), 0.f), eh = fmaxf(fmaxf(a.w, b.w) - fminf(a.y, b.y), 0.f);
    const float enclose = fmaxf(ew * eh, eps);
    r.giou = r.iou - (enclose - uni) / enclose;
    return r;
}

// mode 0: iou, 1: giou.  aligned: out[i] over rows; else out[i][j] over A x G
__global__ __launch_bounds__(256) void overlaps_kernel(const float4* __restrict__ b1, const float4* __restrict__ b2, int64_t A,
                                                       int64_t Gn, int aligned, int mode, float eps, float* __restrict__ out) {
    const int64_t e = blockIdx.x * 256ll + threadIdx.x;
    const int64_t total = aligned ? A : A * Gn;
    if (e >= total) return;
    const int64_t i = aligned ? e : e / Gn, j = aligned ? e : e - i * Gn;
    const IouParts r = overlaps(b1[i], b2[j], eps);
    out[e] = mode ? r.giou : r.iou;
}

// GIoU loss rows (1 - giou) and the analytic gradient w.r.t. the predicted box (the target is detached)
__global__ __launch_bounds__(256) void giou_kernel(const float4* __restrict__ pred, const float4* __restrict__ target,
                                                   const float* __restrict__ coef, int64_t n, float eps,
                                                   float* __restrict__ rows, float4* __restrict__ dpred) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n) return;
    const float4 a = pred[i], b = target[i];
    const float area1 = (a.z - a.x) * (a.w - a.y), area2 = (b.z - b.x) * (b.w - b.y);
    const float ltx = fmaxf(a.x, b.x), lty = fmaxf(a.y, b.y), rbx = fminf(a.z, b.z), rby = fminf(a.w, b.w);
    const float w = fmaxf(rbx - ltx, 0.f), h = fmaxf(rby - lty, 0.f);
    const float overlap = w * h;
    const float uni_raw = area1 + area2 - overlap;
    const float uni = fmaxf(uni_raw, eps);
    const float iou = overlap / uni;
    const float ex1 = fminf(a.x, b.x), ey1 = fminf(a.y, b.y), ex2 = fmaxf(a.z, b.z), ey2 = fmaxf(a.w, b.w);
    const float ew = fmaxf(ex2 - ex1, 0.f), eh = fmaxf(ey2 - ey1, 0.f);
    const float enc_raw = ew * eh;
    const float enc = fmaxf(enc_raw, eps);
    if (rows) rows[i] = 1.f - (iou - (enc - uni) / enc);
    if (!dpred) return;
    // loss = 1 - overlap/uni + 1 - uni/enc  =>  dloss = -d(overlap)/uni + overlap/uni^2 d(uni) - d(uni)/enc + uni/enc^2 d(enc)
    // partials w.r.t. the four coordinates of the predicted box (x1, y1, x2, y2)
    // torch.max / torch.min split the gradient evenly at ties (boxes clipped to the same image border tie often);
    // clamp(min=0) passes the gradient where its argument is >= 0
    auto gmax = [](float x, float y) { return x > y ? 1.f : (x == y ? 0.5f : 0.f); };      // d max(x, y) / dx
    auto gmin = [](float x, float y) { return x < y ? 1.f : (x == y ? 0.5f : 0.f); };      // d min(x, y) / dx
    const float pw = (rbx - ltx) >= 0.f ? 1.f : 0.f, ph = (rby - lty) >= 0.f ? 1.f : 0.f;
    const float d_ov[4] = {-gmax(a.x, b.x) * pw * h, -gmax(a.y, b.y) * ph * w, gmin(a.z, b.z) * pw * h, gmin(a.w, b.w) * ph * w};
    const float d_a1[4] = {-(a.w - a.y), -(a.z - a.x), (a.w - a.y), (a.z - a.x)};
    const float qw = (ex2 - ex1) >= 0.f ? 1.f : 0.f, qh = (ey2 - ey1) >= 0.f ? 1.f : 0.f;
    float d_enc[4] = {0.f, 0.f, 0.f, 0.f};
    if (enc_raw > eps) {
        d_enc[0] = -gmin(a.x, b.x) * qw * eh;
        d_enc[1] = -gmin(a.y, b.y) * qh * ew;
        d_enc[2] = gmax(a.z, b.z) * qw * eh;
        d_enc[3] = gmax(a.w, b.w) * qh * ew;
    }
    const float c = coef[i];
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float d_uni = uni_raw > eps ? d_a1[q] - d_ov[q] : 0.f;
        g[q] = c * (-d_ov[q] / uni + overlap / (uni * uni) * d_uni - d_uni / enc + uni / (enc * enc) * d_enc[q]);
    }
    dpred[i] = make_float4(g[0], g[1], g[2], g[3]);
}

// distance2bbox (decode): x1 = px - l, y1 = py - t, x2 = px + r, y2 = py + b (+ optional clamp to [0, W] x [0, H])
__global__ __launch_bounds__(256) void d2b_kernel(const float2* __restrict__ pts, const float4* __restrict__ dist, int64_t n,
                                                  float max_h, float max_w, float4* __restrict__ out) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n) return;
    const float2 p = pts[i];
    const float4 d = dist[i];
    float4 b = make_float4(p.x - d.x, p.y - d.y, p.x + d.z, p.y + d.w);
    if (max_w >= 0.f) {
        b.x = fminf(fmaxf(b.x, 0.f), max_w); b.z = fminf(fmaxf(b.z, 0.f), max_w);
        b.y = fminf(fmaxf(b.y, 0.f), max_h); b.w = fminf(fmaxf(b.w, 0.f), max_h);
    }
    out[i] = b;
}
// gradient of decode w.r.t. the distances: (-g.x1, -g.y1, g.x2, g.y2), zero where the clamp was active
__global__ __launch_bounds__(256) void d2b_bwd_kernel(const float2* __restrict__ pts, const float4* __restrict__ dist,
                                                      const float4* __restrict__ dout, int64_t n, float max_h, float max_w,
                                                      float4* __restrict__ ddist) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n) return;
    const float2 p = pts[i];
    const float4 d = dist[i], g = dout[i];
    float4 r = make_float4(-g.x, -g.y, g.z, g.w);
    if (max_w >= 0.f) {
        const float x1 = p.x - d.x, y1 = p.y - d.y, x2 = p.x + d.z, y2 = p.y + d.w;
        if (x1 < 0.f || x1 > max_w) r.x = 0.f;
        if (y1 < 0.f || y1 > max_h) r.y = 0.f;
        if (x2 < 0.f || x2 > max_w) r.z = 0.f;
        if (y2 < 0.f || y2 > max_h) r.w = 0.f;
    }
    ddist[i] = r;
}
// bbox2distance (encode): l = px - x1 ... clamped to [0, max_dis - eps] when max_dis >= 0
__global__ __launch_bounds__(256) void b2d_kernel(const float2* __restrict__ pts, const float4* __restrict__ box, int64_t n,
                                                  float max_dis, float eps, float4* __restrict__ out) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n) return;
    const float2 p = pts[i];
    const float4 b = box[i];
    float4 d = make_float4(p.x - b.x, p.y - b.y, b.z - p.x, b.w - p.y);
    if (max_dis >= 0.f) {
        const float hi = max_dis - eps;
        d.x = fminf(fmaxf(d.x, 0.f), hi); d.y = fminf(fmaxf(d.y, 0.f), hi);
        d.z = fminf(fmaxf(d.z, 0.f), hi); d.w = fminf(fmaxf(d.w, 0.f), hi);
    }
    out[i] = d;
}

// out[0] = scale * sum_i rows[i] * (weight ? weight[i] : 1), accumulated in f64 (one block)
__global__ __launch_bounds__(1024) void wsum_kernel(const float* __restrict__ rows, const float* __restrict__ weight, int64_t n,
                                                    double scale, float* __restrict__ out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)rows[i] * (weight ? (double)weight[i] : 1.0);
    s = erd::wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += red[q];
        out[0] = (float)(t * scale);
    }
}
// coef[i] = upstream[0] * scale * (weight ? weight[i] : 1)
__global__ __launch_bounds__(256) void coef_kernel(const float* __restrict__ upstream, const float* __restrict__ weight, int64_t n,
                                                   float scale, float* __restrict__ coef) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i < n) coef[i] = upstream[0] * scale * (weight ? weight[i] : 1.f);
}

// out[i] = rows[i] * scale * (weight ? weight[i] : 1)   (reduction='none', forward and backward)
__global__ __launch_bounds__(256) void rows_mul_kernel(const float* __restrict__ rows, const float* __restrict__ weight, int64_t n,
                                                       float scale, float* __restrict__ out) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i < n) out[i] = rows[i] * scale * (weight ? weight[i] : 1.f);
}

// AssignResult fields from the assigner's (IoU, gt) keys (erd_atss_assign workspace): gt_inds (0 = unassigned,
// g + 1), max_overlaps (IoU of the assigned gt, -INF = -1e8 for unassigned: atss_assigner.py:238-246), labels (-1 / class)
__global__ __launch_bounds__(256) void atss_result_kernel(const unsigned long long* __restrict__ best,
                                                          const int64_t* __restrict__ gt_labels, int64_t A,
                                                          int64_t* __restrict__ gt_inds, float* __restrict__ max_ov,
                                                          int64_t* __restrict__ labels) {
    const int64_t a = blockIdx.x * 256ll + threadIdx.x;
    if (a >= A) return;
    const unsigned long long k = best[a];
    if (k != 0ull) {
        const int g = (int)(0xffffffffu - (unsigned)(k & 0xffffffffull));
        gt_inds[a] = g + 1;
        max_ov[a] = __uint_as_float((unsigned)(k >> 32));
        labels[a] = gt_labels[g];
    } else {
        gt_inds[a] = 0;
        max_ov[a] = -100000000.0f;
        labels[a] = -1;
    }
}

inline unsigned blocks_for(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace

extern "C" int erd_qfl_rows(const float* pred, const int64_t* label, const float* score, int64_t n, int C, float* rows,
                            erd_stream_t stream) {
    ERD_REQUIRE(pred && label && score && rows && C > 0, "qfl_rows: bad args");
    if (n > 0) hipLaunchKernelGGL(qfl_rows_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, (hipStream_t)stream, pred, label, score, n, C, rows);
    return erd::check_launch("qfl_rows");
}
extern "C" int erd_qfl_bwd(const float* pred, const int64_t* label, const float* score, const float* coef, int64_t n, int C,
                           float* dpred, erd_stream_t stream) {
    ERD_REQUIRE(pred && label && score && coef && dpred && C > 0, "qfl_bwd: bad args");
    if (n > 0) hipLaunchKernelGGL(qfl_bwd_kernel, dim3(blocks_for(n * C)), dim3(256), 0, (hipStream_t)stream, pred, label, score, coef, n, C, dpred);
    return erd::check_launch("qfl_bwd");
}
extern "C" int erd_dfl(const float* pred, const float* target, const float* coef, int64_t m, int nb, float* rows, float* dpred,
                       erd_stream_t stream) {
    ERD_REQUIRE(pred && target && nb >= 2 && nb <= MAXB && (rows || (dpred && coef)), "dfl: bad args");
    if (m > 0) hipLaunchKernelGGL(dfl_kernel, dim3(blocks_for(m)), dim3(256), 0, (hipStream_t)stream, pred, target, coef, m, nb, rows, dpred);
    return erd::check_launch("dfl");
}
extern "C" int erd_kd_kl_rows(const float* pred, const float* soft, const float* coef, int64_t m, int nb, float T, float* rows,
                              float* dpred, erd_stream_t stream) {
    ERD_REQUIRE(pred && soft && nb >= 2 && nb <= MAXB && T > 0.f && (rows || (dpred && coef)), "kd_kl_rows: bad args");
    if (m > 0) hipLaunchKernelGGL(kdkl_kernel, dim3(blocks_for(m)), dim3(256), 0, (hipStream_t)stream, pred, soft, coef, m, nb, T, rows, dpred);
    return erd::check_launch("kd_kl_rows");
}
extern "C" int erd_integral(const float* x, const float* dy, int64_t m, int nb, float* y, float* dx, erd_stream_t stream) {
    ERD_REQUIRE(x && nb >= 2 && nb <= MAXB && (y || (dx && dy)), "integral: bad args");
    if (m > 0) hipLaunchKernelGGL(integral_kernel, dim3(blocks_for(m)), dim3(256), 0, (hipStream_t)stream, x, dy, m, nb, y, dx);
    return erd::check_launch("integral");
}
extern "C" int erd_bbox_overlaps(const float* b1, const float* b2, int64_t A, int64_t Gn, int aligned, int mode, float eps,
                                 float* out, erd_stream_t stream) {
    const int64_t total = aligned ? A : A * Gn;
    ERD_REQUIRE(((b1 && b2 && out) || total == 0) && (mode == 0 || mode == 1) && (!aligned || A == Gn), "bbox_overlaps: bad args");
    if (total > 0)
        hipLaunchKernelGGL(overlaps_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float4*>(b1), reinterpret_cast<const float4*>(b2), A, Gn, aligned, mode, eps, out);
    return erd::check_launch("bbox_overlaps");
}
extern "C" int erd_giou(const float* pred, const float* target, const float* coef, int64_t n, float eps, float* rows, float* dpred,
                        erd_stream_t stream) {
    ERD_REQUIRE(pred && target && (rows || (dpred && coef)), "giou: bad args");
    if (n > 0)
        hipLaunchKernelGGL(giou_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(pred),
                           reinterpret_cast<const float4*>(target), coef, n, eps, rows, reinterpret_cast<float4*>(dpred));
    return erd::check_launch("giou");
}
extern "C" int erd_distance2bbox(const float* points, const float* dist, const float* dout, int64_t n, float max_h, float max_w,
                                 float* out, float* ddist, erd_stream_t stream) {
    ERD_REQUIRE(points && dist && (out || (ddist && dout)), "distance2bbox: bad args");
    if (n > 0 && out)
        hipLaunchKernelGGL(d2b_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2*>(points),
                           reinterpret_cast<const float4*>(dist), n, max_h, max_w, reinterpret_cast<float4*>(out));
    if (n > 0 && ddist)
        hipLaunchKernelGGL(d2b_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2*>(points),
                           reinterpret_cast<const float4*>(dist), reinterpret_cast<const float4*>(dout), n, max_h, max_w,
                           reinterpret_cast<float4*>(ddist));
    return erd::check_launch("distance2bbox");
}
extern "C" int erd_bbox2distance(const float* points, const float* boxes, int64_t n, float max_dis, float eps, float* out,
                                 erd_stream_t stream) {
    ERD_REQUIRE(points && boxes && out, "bbox2distance: bad args");
    if (n > 0)
        hipLaunchKernelGGL(b2d_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2*>(points),
                           reinterpret_cast<const float4*>(boxes), n, max_dis, eps, reinterpret_cast<float4*>(out));
    return erd::check_launch("bbox2distance");
}
extern "C" int erd_weighted_sum(const float* rows, const float* weight, int64_t n, double scale, float* out, erd_stream_t stream) {
    ERD_REQUIRE(out && (rows || n == 0), "weighted_sum: bad args");
    hipLaunchKernelGGL(wsum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rows, weight, n, scale, out);
    return erd::check_launch("weighted_sum");
}
extern "C" int erd_loss_coef(const float* upstream, const float* weight, int64_t n, float scale, float* coef, erd_stream_t stream) {
    ERD_REQUIRE(upstream && (coef || n == 0), "loss_coef: bad args");
    if (n > 0) hipLaunchKernelGGL(coef_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, upstream, weight, n, scale, coef);
    return erd::check_launch("loss_coef");
}
extern "C" int erd_rows_mul(const float* rows, const float* weight, int64_t n, float scale, float* out, erd_stream_t stream) {
    ERD_REQUIRE((rows && out) || n == 0, "rows_mul: bad args");
    if (n > 0) hipLaunchKernelGGL(rows_mul_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, rows, weight, n, scale, out);
    return erd::check_launch("rows_mul");
}
extern "C" int erd_atss_result(const void* assign_ws, const int64_t* gt_labels, int64_t A, int64_t* gt_inds, float* max_overlaps,
                               int64_t* labels, erd_stream_t stream) {
    ERD_REQUIRE(assign_ws && gt_inds && max_overlaps && labels, "atss_result: bad args");
    if (A > 0)
        hipLaunchKernelGGL(atss_result_kernel, dim3(blocks_for(A)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const unsigned long long*>(assign_ws), gt_labels, A, gt_inds, max_overlaps, labels);
    return erd::check_launch("atss_result");
}
