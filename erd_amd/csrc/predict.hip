// Inference post-processing of the GFL head, gfx950 only.
//
// Replaces, per image, GFLHead._predict_by_feat_single (gfl_head.py:408-502) + filter_scores_and_topk
// (models/utils/misc.py:308-354) + BaseDenseHead._bbox_post_process (base_dense_head.py:424-486) +
// mmcv.ops.batched_nms:
//   per level: sigmoid -> score > score_thr -> the nms_pre highest scores (score desc, then (anchor, class)
//   index asc -- the reference's sort is unstable, this is one of its valid outcomes) -> Integral x stride ->
//   distance2bbox around the anchor centre, clamped to img_shape;
//   per image: x 1/scale_factor, drop boxes with w or h <= min_bbox_size, class-offset NMS (IoU > thr
//   suppresses), first max_per_img survivors in score order.
//
// The top-k is an exact 3-pass radix select (11/11/10 bits of the score's fp32 pattern) over the level-concatenated
// [N][A][C] logits: pure HBM scans (C*4 bytes per anchor per pass) + tiny per-(image, level) scans; the <= nms_pre
// survivors are sorted by one workgroup per (image, level) in LDS.  No host synchronisation anywhere.
#include "erd_common.h"

namespace {

constexpr int HB = 2048;            // histogram bins per pass
constexpr int ROWS_PER_BLOCK = 64;  // anchors per scanning workgroup
constexpr int KP_MAX = 4096;        // sort capacity per (image, level)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct Sel {             // per (image, level) selection state, 8 x int32
    uint32_t prefix;     // key bits fixed so far / after pass 2: the k-th largest key T
    int32_t need;        // how many are still wanted inside the current prefix bucket / after pass 2: r (== T wanted)
    int32_t total;       // candidates above score_thr
    int32_t all;         // total <= k: take everything
    int32_t c_eq;        // candidates with key == T
    int32_t n_gt;        // append cursor of key > T
    int32_t n_eq;        // append cursor of key == T
    int32_t count;       // min(k, total): what the level contributes
};

struct LevelMap {
    int nseg;
    int64_t off[ERD_MAX_SEG], cnt[ERD_MAX_SEG];
    int chunk0[ERD_MAX_SEG + 1];   // first scanning block of each level
    float stride[ERD_MAX_SEG];
};

__device__ __forceinline__ uint32_t digit_of(uint32_t key, int pass) {
    return pass == 0 ? (key >> 21) : (pass == 1 ? ((key >> 10) & 2047u) : (key & 1023u));
}
__device__ __forceinline__ bool prefix_ok(uint32_t key, uint32_t prefix, int pass) {
    return pass == 0 ? true : (pass == 1 ? (key >> 21) == (prefix >> 21) : (key >> 10) == (prefix >> 10));
}

// ---- pass p histogram: grid (chunks, N) -------------------------------------------------------------------------
template <int PASS>
__global__ __launch_bounds__(256) void pred_hist_kernel(const float* __restrict__ cls, int64_t A, int C, LevelMap lm,
                                                        float thr, const Sel* __restrict__ sel,
                                                        uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[HB];
    const int n = blockIdx.y;
    int l = 0;
    while (l + 1 < lm.nseg && (int)blockIdx.x >= lm.chunk0[l + 1]) ++l;
    const int seg = n * lm.nseg + l;
    const Sel st = sel[seg];
    if (PASS > 0 && st.all) return;
    for (int i = threadIdx.x; i < HB; i += 256) h[i] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)(blockIdx.x - lm.chunk0[l]) * ROWS_PER_BLOCK;
    const int64_t r1 = min(lm.cnt[l], r0 + ROWS_PER_BLOCK);
    const float* base = cls + ((int64_t)n * A + lm.off[l] + r0) * C;
    const int64_t nel = (r1 - r0) * C;
    for (int64_t i = threadIdx.x; i < nel; i += 256) {
        const float s = sigmoidf_(base[i]);
        if (!(s > thr)) continue;
        const uint32_t key = __float_as_uint(s);
        if (!prefix_ok(key, st.prefix, PASS)) continue;
        atomicAdd(&h[digit_of(key, PASS)], 1u);
    }
    __syncthreads();
    uint32_t* g = hist + (int64_t)seg * HB;
    for (int i = threadIdx.x; i < HB; i += 256)
        if (h[i]) atomicAdd(&g[i], h[i]);
}

// ---- pass p scan: grid (N * nseg), 256 threads; finds the bin holding the `need`-th largest, clears the histogram
template <int PASS>
__global__ __launch_bounds__(256) void pred_scan_kernel(uint32_t* __restrict__ hist, Sel* __restrict__ sel, int k) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t incl[256];
    const int seg = blockIdx.x;
    Sel st = sel[seg];
    uint32_t* g = hist + (int64_t)seg * HB;
    if (PASS > 0 && st.all) return;
    // thread t owns bins [HB-1-8t-7, HB-1-8t] (descending)
    uint32_t loc[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { loc[j] = g[HB - 1 - (8 * (int)threadIdx.x + j)]; sum += loc[j]; }
    part[threadIdx.x] = sum;
    incl[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t v = threadIdx.x >= (unsigned)o ? incl[threadIdx.x - o] : 0u;
        __syncthreads();
        incl[threadIdx.x] += v;
        __syncthreads();
    }
    const uint32_t total = incl[255];
    if (PASS == 0) {
        if (threadIdx.x == 0) {
            st.total = (int)total;
            st.all = (int)total <= k ? 1 : 0;
            st.need = k;
            st.prefix = 0;
            st.c_eq = 0;
            st.n_gt = 0;
            st.n_eq = 0;
            st.count = (int)total <= k ? (int)total : k;
            if (st.all) sel[seg] = st;
        }
        if ((int)total <= k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[HB - 1 - (8 * (int)threadIdx.x + j)] = 0;
            return;
        }
        st.need = k;
        st.total = (int)total;
        st.all = 0;
        st.prefix = 0;
        st.c_eq = 0;
        st.n_gt = 0;
        st.n_eq = 0;
        st.count = k;
    }
    const uint32_t need = (uint32_t)st.need;
    const uint32_t before = incl[threadIdx.x] - part[threadIdx.x];
    if (before < need && need <= incl[threadIdx.x]) {      // exactly one thread
        uint32_t cum = before;
        int d = 0;
        uint32_t hd = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (cum < need && need <= cum + loc[j]) { d = HB - 1 - (8 * (int)threadIdx.x + j); hd = loc[j]; break; }
            cum += loc[j];
        }
        const int shift = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0);
        st.prefix |= (uint32_t)d << shift;
        st.need = (int)(need - cum);          // wanted inside bin d
        if (PASS == 2) st.c_eq = (int)hd;     // prefix is now the exact k-th key; need = r of the c_eq equal keys
        sel[seg] = st;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) g[HB - 1 - (8 * (int)threadIdx.x + j)] = 0;
}

__device__ __forceinline__ uint64_t compose(uint32_t key, uint32_t flat) {
    return ((uint64_t)key << 32) | (uint64_t)(0xFFFFFFFFu - flat);
}

// ---- compaction: grid (chunks, N).  key > T always; key == T here only when all of them fit -------------------
__global__ __launch_bounds__(256) void pred_compact_kernel(const float* __restrict__ cls, int64_t A, int C, LevelMap lm,
                                                           float thr, Sel* __restrict__ sel,
                                                           uint64_t* __restrict__ cand, int KP) {
    const int n = blockIdx.y;
    int l = 0;
    while (l + 1 < lm.nseg && (int)blockIdx.x >= lm.chunk0[l + 1]) ++l;
    const int seg = n * lm.nseg + l;
    const Sel st = sel[seg];
    const uint32_t T = st.all ? 0u : st.prefix;
    const bool eq_here = !st.all && st.c_eq == st.need;
    const int c_gt = st.count - (st.all ? 0 : st.need);
    const int64_t r0 = (int64_t)(blockIdx.x - lm.chunk0[l]) * ROWS_PER_BLOCK;
    const int64_t r1 = min(lm.cnt[l], r0 + ROWS_PER_BLOCK);
    const float* base = cls + ((int64_t)n * A + lm.off[l] + r0) * C;
    const int64_t nel = (r1 - r0) * C;
    uint64_t* out = cand + (int64_t)seg * KP;
    for (int64_t i = threadIdx.x; i < nel; i += 256) {
        const float s = sigmoidf_(base[i]);
        if (!(s > thr)) continue;
        const uint32_t key = __float_as_uint(s);
        const uint32_t flat = (uint32_t)(r0 * C + i);
        if (key > T) {
            const int slot = atomicAdd(&sel[seg].n_gt, 1);
            out[slot] = compose(key, flat);
        } else if (key == T && eq_here) {
            const int slot = c_gt + atomicAdd(&sel[seg].n_eq, 1);
            out[slot] = compose(key, flat);
        }
    }
}

// ---- ties cut by the k limit: the lowest (anchor, class) indices win.  grid (N*nseg), 1024 threads; no-op unless
// more keys equal T than are wanted
__global__ __launch_bounds__(1024) void pred_tie_kernel(const float* __restrict__ cls, int64_t A, int C, LevelMap lm,
                                                        float thr, const Sel* __restrict__ sel,
                                                        uint64_t* __restrict__ cand, int KP) {
    __shared__ int wsum[16];
    __shared__ int running;
    const int seg = blockIdx.x;
    const int n = seg / lm.nseg, l = seg % lm.nseg;
    const Sel st = sel[seg];
    if (st.all || st.c_eq == st.need) return;
    const uint32_t T = st.prefix;
    const int r = st.need, c_gt = st.count - st.need;
    const float* base = cls + ((int64_t)n * A + lm.off[l]) * C;
    const int64_t nel = lm.cnt[l] * C;
    uint64_t* out = cand + (int64_t)seg * KP;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t b = 0; b < nel; b += 1024) {
        const int64_t i = b + threadIdx.x;
        bool hit = false;
        if (i < nel) {
            const float s = sigmoidf_(base[i]);
            hit = s > thr && __float_as_uint(s) == T;
        }
        const unsigned long long m = __ballot(hit);
        const int before_in_wave = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int wbefore = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) wbefore += wsum[w]; tot += wsum[w]; }
        const int start = running;
        const int pos = start + wbefore + before_in_wave;
        if (hit && pos < r) out[c_gt + pos] = compose(T, (uint32_t)i);
        __syncthreads();
        if (threadIdx.x == 0) running = start + tot;
        __syncthreads();
        if (start + tot >= r) break;
    }
}

// ---- sort + decode: grid (nseg, N), 1024 threads --------------------------------------------------------------
__global__ __launch_bounds__(1024) void pred_sort_decode_kernel(const uint64_t* __restrict__ cand, int KP,
                                                                const Sel* __restrict__ sel,
                                                                const float* __restrict__ bbox,
                                                                const float4* __restrict__ anchors,
                                                                const float* __restrict__ img_hw, int64_t A, int C,
                                                                LevelMap lm, int k, float4* __restrict__ boxes,
                                                                float* __restrict__ scores,
                                                                int32_t* __restrict__ labels,
                                                                int32_t* __restrict__ num) {
    extern __shared__ uint64_t keys[];
    const int l = blockIdx.x, n = blockIdx.y;
    const int seg = n * lm.nseg + l;
    const int cnt = sel[seg].count;
    int base = 0;
    for (int q = 0; q < l; ++q) base += sel[n * lm.nseg + q].count;
    if (l == lm.nseg - 1 && threadIdx.x == 0) num[n] = base + cnt;
    const uint64_t* in = cand + (int64_t)seg * KP;
    for (int i = threadIdx.x; i < KP; i += 1024) keys[i] = i < cnt ? in[i] : 0ull;
    __syncthreads();
    // bitonic sort, descending
    for (int size = 2; size <= KP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < KP / 2; t += 1024) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    const float H = img_hw[2 * n], W = img_hw[2 * n + 1];
    const float stride_l = lm.stride[l];
    const int max_cols = k * lm.nseg;
    for (int i = threadIdx.x; i < cnt; i += 1024) {
        const uint64_t kv = keys[i];
        const float s = __uint_as_float((uint32_t)(kv >> 32));
        const uint32_t flat = 0xFFFFFFFFu - (uint32_t)(kv & 0xFFFFFFFFull);
        const int64_t a = lm.off[l] + flat / (uint32_t)C;
        const int lab = (int)(flat % (uint32_t)C);
        float d[4];
        const float* zb = bbox + ((int64_t)n * A + a) * 68;
        for (int q = 0; q < 4; ++q) {
            float mx = zb[q * 17];
#pragma unroll
            for (int j = 1; j < 17; ++j) mx = fmaxf(mx, zb[q * 17 + j]);
            float e[17], sum = 0.f;
#pragma unroll
            for (int j = 0; j < 17; ++j) { e[j] = expf(zb[q * 17 + j] - mx); sum += e[j]; }
            const float inv = 1.0f / sum;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 17; ++j) acc += (e[j] * inv) * (float)j;
            d[q] = acc * stride_l;
        }
        const float4 an = anchors[a];
        const float cx = (an.z + an.x) / 2.0f, cy = (an.w + an.y) / 2.0f;
        float4 b = make_float4(cx - d[0], cy - d[1], cx + d[2], cy + d[3]);
        b.x = fminf(fmaxf(b.x, 0.f), W); b.z = fminf(fmaxf(b.z, 0.f), W);
        b.y = fminf(fmaxf(b.y, 0.f), H); b.w = fminf(fmaxf(b.w, 0.f), H);
        const int64_t o = (int64_t)n * max_cols + base + i;
        boxes[o] = b;
        scores[o] = s;
        labels[o] = lab;
    }
}

// ---- per image: rescale, size filter, class-offset NMS, first max_per_img survivors -----------------------------
// ws per image: fb[M] float4 (rescaled, filtered), ob[M] float4 (offset boxes), fs[M], fl[M] int, order[M] int,
// removed[M] int
__global__ __launch_bounds__(1024) void pred_nms_kernel(const float4* __restrict__ boxes,
                                                        const float* __restrict__ scores,
                                                        const int32_t* __restrict__ labels,
                                                        const int32_t* __restrict__ num, int max_cols,
                                                        const float* __restrict__ inv_scale, float min_size,
                                                        float iou_thr, int max_per_img, float* __restrict__ dets,
                                                        int64_t* __restrict__ det_labels,
                                                        int32_t* __restrict__ det_num, float* __restrict__ ws) {
    __shared__ int wsum[16];
    __shared__ float red[16];
    __shared__ int s_running, s_kept;
    const int n = blockIdx.x;
    const int M = num[n];
    float* w0 = ws + (int64_t)n * max_cols * 12;
    float4* fb = reinterpret_cast<float4*>(w0);
    float4* ob = fb + max_cols;
    float* fs = reinterpret_cast<float*>(ob + max_cols);
    int* fl = reinterpret_cast<int*>(fs + max_cols);
    int* order = fl + max_cols;
    int* removed = order + max_cols;
    const float sx = inv_scale[2 * n], sy = inv_scale[2 * n + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_running = 0; s_kept = 0; }
    __syncthreads();
    // ordered compaction of the boxes that survive the size filter (results[valid_mask])
    float lmax = -INFINITY;
    for (int b0 = 0; b0 < M; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        bool ok = false;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < M) {
            b = boxes[(int64_t)n * max_cols + i];
            b.x *= sx; b.y *= sy; b.z *= sx; b.w *= sy;
            ok = min_size < 0.f || ((b.z - b.x) > min_size && (b.w - b.y) > min_size);
        }
        const unsigned long long m = __ballot(ok);
        const int inw = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int wb = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) wb += wsum[w]; tot += wsum[w]; }
        const int start = s_running;
        if (ok) {
            const int p = start + wb + inw;
            fb[p] = b;
            fs[p] = scores[(int64_t)n * max_cols + i];
            fl[p] = labels[(int64_t)n * max_cols + i];
            removed[p] = 0;
            lmax = fmaxf(lmax, fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
        }
        __syncthreads();
        if (threadIdx.x == 0) s_running = start + tot;
        __syncthreads();
    }
    const int K = s_running;
    lmax = erd::wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    float maxc = red[0];
    for (int i = 1; i < 16; ++i) maxc = fmaxf(maxc, red[i]);
    const float offs = maxc + 1.0f;       // boxes.max() + 1 (batched_nms)
    for (int i = threadIdx.x; i < K; i += 1024) {
        const float o = (float)fl[i] * offs;
        float4 b = fb[i];
        b.x += o; b.y += o; b.z += o; b.w += o;
        ob[i] = b;
    }
    __syncthreads();
    // rank sort: score desc, position asc
    for (int i = threadIdx.x; i < K; i += 1024) {
        const float si = fs[i];
        int rank = 0;
        for (int j = 0; j < K; ++j) {
            const float sj = fs[j];
            rank += (sj > si || (sj == si && j < i)) ? 1 : 0;
        }
        order[rank] = i;
    }
    __syncthreads();
    for (int oi = 0; oi < K; ++oi) {
        const int i = order[oi];
        if (removed[i]) continue;               // uniform
        if (threadIdx.x == 0) {
            const int p = s_kept;
            const float4 b = fb[i];
            float* d = dets + ((int64_t)n * max_per_img + p) * 5;
            d[0] = b.x; d[1] = b.y; d[2] = b.z; d[3] = b.w; d[4] = fs[i];
            det_labels[(int64_t)n * max_per_img + p] = fl[i];
        }
        const int kept_now = s_kept + 1;        // read before the barrier, written after it
        const float4 bi = ob[i];
        const float ai = (bi.z - bi.x) * (bi.w - bi.y);
        if (kept_now < max_per_img) {
            for (int oj = oi + 1 + threadIdx.x; oj < K; oj += 1024) {
                const int j = order[oj];
                if (removed[j]) continue;
                const float4 bj = ob[j];
                const float w = fmaxf(fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x), 0.f);
                const float h = fmaxf(fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y), 0.f);
                const float inter = w * h;
                const float aj = (bj.z - bj.x) * (bj.w - bj.y);
                const float iou = inter / (ai + aj - inter);
                if (iou > iou_thr) removed[j] = 1;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_kept = kept_now;
        __syncthreads();
        if (kept_now >= max_per_img) break;
    }
    __syncthreads();
    if (threadIdx.x == 0) det_num[n] = s_kept;
}

inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

}  // namespace

extern "C" size_t erd_predict_ws_bytes(int N, int nlvl, int nms_pre) {
    const int KP = next_pow2(nms_pre < 2 ? 2 : nms_pre);
    const size_t segs = (size_t)N * nlvl;
    size_t b = segs * HB * sizeof(uint32_t) + segs * sizeof(Sel);
    b = (b + 255) / 256 * 256;
    b += segs * KP * sizeof(uint64_t);
    b = (b + 255) / 256 * 256;
    b += (size_t)N * nlvl * nms_pre * 12 * sizeof(float);      // NMS scratch
    return b;
}

extern "C" int erd_predict_topk(const float* cls, const float* bbox, const float* anchors, int N, int64_t A, int C,
                                const erd_levels* lv, const int* strides, const float* img_hw, float score_thr,
                                int nms_pre, float* boxes, float* scores, int32_t* labels, int32_t* num, void* ws,
                                size_t ws_bytes, erd_stream_t stream) {
    ERD_REQUIRE(cls && bbox && anchors && lv && strides && img_hw && boxes && scores && labels && num && ws,
                "predict_topk: null");
    ERD_REQUIRE(N > 0 && lv->nseg > 0 && lv->nseg <= ERD_MAX_SEG, "predict_topk: bad N / levels");
    ERD_REQUIRE(nms_pre > 0 && nms_pre <= KP_MAX, "predict_topk: nms_pre must be in [1, %d]", KP_MAX);
    ERD_REQUIRE(score_thr >= 0.f, "predict_topk: score_thr must be >= 0 (scores are compared by bit pattern)");
    ERD_REQUIRE(ws_bytes >= erd_predict_ws_bytes(N, lv->nseg, nms_pre), "predict_topk: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int KP = next_pow2(nms_pre < 2 ? 2 : nms_pre);
    const size_t segs = (size_t)N * lv->nseg;
    LevelMap lm;
    lm.nseg = lv->nseg;
    lm.chunk0[0] = 0;
    for (int l = 0; l < lv->nseg; ++l) {
        lm.off[l] = lv->off[l];
        lm.cnt[l] = lv->cnt[l];
        lm.stride[l] = (float)strides[l];
        ERD_REQUIRE(lv->cnt[l] * C < (1ll << 31), "predict_topk: level too large");
        lm.chunk0[l + 1] = lm.chunk0[l] + (int)((lv->cnt[l] + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    }
    char* w = reinterpret_cast<char*>(ws);
    uint32_t* hist = reinterpret_cast<uint32_t*>(w);
    Sel* sel = reinterpret_cast<Sel*>(w + segs * HB * sizeof(uint32_t));
    size_t o = (segs * HB * sizeof(uint32_t) + segs * sizeof(Sel) + 255) / 256 * 256;
    uint64_t* cand = reinterpret_cast<uint64_t*>(w + o);
    ERD_ZERO_ASYNC(w, segs * HB * sizeof(uint32_t) + segs * sizeof(Sel), st);
    const dim3 gs(lm.chunk0[lv->nseg], N), gseg((unsigned)segs);
    hipLaunchKernelGGL(pred_hist_kernel<0>, gs, dim3(256), 0, st, cls, A, C, lm, score_thr, sel, hist);
    hipLaunchKernelGGL(pred_scan_kernel<0>, gseg, dim3(256), 0, st, hist, sel, nms_pre);
    hipLaunchKernelGGL(pred_hist_kernel<1>, gs, dim3(256), 0, st, cls, A, C, lm, score_thr, sel, hist);
    hipLaunchKernelGGL(pred_scan_kernel<1>, gseg, dim3(256), 0, st, hist, sel, nms_pre);
    hipLaunchKernelGGL(pred_hist_kernel<2>, gs, dim3(256), 0, st, cls, A, C, lm, score_thr, sel, hist);
    hipLaunchKernelGGL(pred_scan_kernel<2>, gseg, dim3(256), 0, st, hist, sel, nms_pre);
    hipLaunchKernelGGL(pred_compact_kernel, gs, dim3(256), 0, st, cls, A, C, lm, score_thr, sel, cand, KP);
    hipLaunchKernelGGL(pred_tie_kernel, gseg, dim3(1024), 0, st, cls, A, C, lm, score_thr, sel, cand, KP);
    hipLaunchKernelGGL(pred_sort_decode_kernel, dim3(lv->nseg, N), dim3(1024), KP * sizeof(uint64_t), st, cand, KP,
                       sel, bbox, reinterpret_cast<const float4*>(anchors), img_hw, A, C, lm, nms_pre,
                       reinterpret_cast<float4*>(boxes), scores, labels, num);
    return erd::check_launch("predict_topk");
}

extern "C" int erd_predict_nms(const float* boxes, const float* scores, const int32_t* labels, const int32_t* num,
                               int N, int max_cols, const float* inv_scale, float min_bbox_size, float iou_thr,
                               int max_per_img, float* dets, int64_t* det_labels, int32_t* det_num, void* ws,
                               size_t ws_bytes, erd_stream_t stream) {
    ERD_REQUIRE(boxes && scores && labels && num && inv_scale && dets && det_labels && det_num && ws,
                "predict_nms: null");
    ERD_REQUIRE(N > 0 && max_cols > 0 && max_per_img > 0, "predict_nms: bad sizes");
    ERD_REQUIRE(ws_bytes >= (size_t)N * max_cols * 12 * sizeof(float), "predict_nms: workspace too small");
    hipLaunchKernelGGL(pred_nms_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(boxes), scores, labels, num, max_cols, inv_scale, min_bbox_size,
                       iou_thr, max_per_img, dets, det_labels, det_num, reinterpret_cast<float*>(ws));
    return erd::check_launch("predict_nms");
}
