// HBM-bound kernels around the convolutions: stem, max-pool, frozen-BN helpers, GroupNorm(32)+ReLU
// forward/backward over level-concatenated [N][A][C] maps, FPN top-down add, bias/scale helpers, SGD.
// All activations NHWC fp32; 16-byte (float4) accesses, channels fastest => fully coalesced rows.
#include "erd_common.h"

namespace erd {
thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace erd

extern "C" int erd_abi_version(void) { return ERD_ABI_VERSION; }
#ifndef ERD_CSRC_SHA
#define ERD_CSRC_SHA "unknown"
#endif
extern "C" const char* erd_csrc_sha(void) { return ERD_CSRC_SHA; }
// 1 when any translation unit of this library was compiled with a probe of erd_probes.h (a timing / accuracy / trace variant)
extern "C" __attribute__((weak)) int erd_probe_build_marker;
extern "C" int erd_probe_build(void) { return &erd_probe_build_marker != nullptr ? erd_probe_build_marker : 0; }
extern "C" const char* erd_last_error(void) { return erd::g_err; }
namespace erd {
static int g_cu_reserve = 0;
int usable_cus(int physical) { return physical - g_cu_reserve >= 8 ? physical - g_cu_reserve : (physical < 8 ? physical : 8); }
}  // namespace erd
// ABI v6: CUs every later launch leaves free (persistent Winograd grids, stream-K grids, the activation-stationary kernel's grid, and --
// through erd_usable_cus -- the caller's one-round weight-gradient splits); n < 0 only queries.  Returns the previous reserve.
extern "C" int erd_set_cu_reserve(int n) {
    const int prev = erd::g_cu_reserve;
    if (n >= 0) erd::g_cu_reserve = n;
    return prev;
}
extern "C" int erd_usable_cus(void) {
    int dev = 0, n = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
    return erd::usable_cus(n);
}

namespace {

// ------------------------------------------------------------------------------------------------
// stem: conv 7x7 stride 2 pad 3, 3->64, input NCHW (the detector's public input layout), output NHWC,
// frozen BN + ReLU fused (resnet.py:636-638).  Implicit GEMM on the fp32 matrix cores: M = 8x32 output pixels per
// block (one 32-pixel row per MFMA row block, two per wave), N = 64 channels, K = 7*7*3 = 147 (+1 zero row) in
// the weights' own (kh, kw, c) order.  The 21x69x3 input patch and the [148][64] weight image live in LDS; the A
// operand is gathered straight from the patch (lane = pixel, the k-th patch offset is a compile-time constant per
// lane half), the B operand is a conflict-free row read.
// ------------------------------------------------------------------------------------------------
constexpr int ST_TH = 8, ST_TW = 32;
constexpr int ST_PH = ST_TH * 2 + 5, ST_PW = ST_TW * 2 + 5;  // 21 x 69
constexpr int ST_PWP = ST_PW + 1;                             // padded row
constexpr int ST_K = 148;

__host__ __device__ constexpr int stem_patch_off(int k) {     // (kh, kw, c) -> offset inside patch[3][21][70]
    return k >= 147 ? 0 : (k % 3) * ST_PH * ST_PWP + (k / 21) * ST_PWP + (k / 3) % 7;
}

__global__ __launch_bounds__(256, 2) void stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      float* __restrict__ out, int N, int H, int W, int OH, int OW) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int WLD = 65;                    // padded weight row: conflict-free transposing store and row reads
    constexpr int PATCH = 3 * ST_PH * ST_PWP, NLOAD = (3 * ST_PH * ST_PW + 255) / 256;
    __shared__ float wl[ST_K * WLD];           // [(kh,kw,c)][co], row 147 = 0
    __shared__ float patch[2][PATCH];
    // scale / shift in LDS: as global loads inside the store loop each of them sat between two stores, and gfx950 retires loads and
    // stores through one in-order counter -- every store waited for the previous one's acknowledgement (EXPERIMENTS 7f); registers
    // do not hold them either (the kernel sits at 256)
    __shared__ __attribute__((aligned(16))) float ssl[128];
    const int tid = threadIdx.x;
    if (tid < 128) ssl[tid] = tid < 64 ? scale[tid] : shift[tid - 64];
    // weights arrive as [co][kh][kw][c] (OHWI); LDS wants [(kh,kw,c)][co].  Loaded ONCE: the grid is persistent.
    for (int i = tid; i < 147 * 64; i += 256) {
        const int co = i / 147, k = i - co * 147;
        wl[k * WLD + co] = w[i];
    }
    if (tid < 64) wl[147 * WLD + tid] = 0.f;
    const int tx_n = (OW + ST_TW - 1) / ST_TW, ty_n = (OH + ST_TH - 1) / ST_TH;
    const int total = tx_n * ty_n * N;

    float pv[NLOAD];
    auto fetch = [&](int t) {                  // the 21x69x3 input patch of tile t -> registers
        const int n = t / (tx_n * ty_n), rem = t - n * tx_n * ty_n;
        const int ih0 = (rem / tx_n) * ST_TH * 2 - 3, iw0 = (rem % tx_n) * ST_TW * 2 - 3;
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int i = tid + q * 256;
            const int c = i / (ST_PH * ST_PW);
            const int r = (i / ST_PW) % ST_PH;
            const int col = i % ST_PW;
            const int ih = ih0 + r, iw = iw0 + col;
            float v = 0.f;
            if (i < 3 * ST_PH * ST_PW && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                v = x[((int64_t)(n * 3 + c) * H + ih) * W + iw];
            pv[q] = v;
        }
    };
    auto stash = [&](float* dst) {
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int i = tid + q * 256;
            if (i < 3 * ST_PH * ST_PW) dst[(i / ST_PW) * ST_PWP + i % ST_PW] = pv[q];     // (c*PH + r) rows of PWP
        }
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, h = lane >> 5;
    int t = blockIdx.x;
    if (t < total) { fetch(t); stash(patch[0]); }
    __syncthreads();
    for (int it = 0; t < total; t += gridDim.x, ++it) {
        const float* pc = patch[it & 1];
        const bool more = t + (int)gridDim.x < total;
        if (more) fetch(t + gridDim.x);        // in flight behind the MFMAs of this tile
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // pixel (row 2*wave + i, column li) of the tile -> patch origin (2*row, 2*col)
        const float* pa = pc + (2 * (2 * wave)) * ST_PWP + 2 * li;
        const float* pb = wl + h * WLD + li;
        // software pipeline: the operands of step ks+1 are read while the four MFMAs of step ks run; the scheduling
        // barrier keeps the compiler from hoisting all 296 LDS reads to the top (256 VGPRs + spills otherwise)
        const int off0 = h ? stem_patch_off(1) : stem_patch_off(0);
        float a0 = pa[off0], a1 = pa[off0 + 2 * ST_PWP], b0 = pb[0], b1 = pb[32];
#pragma unroll
        for (int ks = 0; ks < ST_K / 2; ++ks) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (ks + 1 < ST_K / 2) {
                const int off = h ? stem_patch_off(2 * ks + 3) : stem_patch_off(2 * ks + 2);
                na0 = pa[off]; na1 = pa[off + 2 * ST_PWP];
                nb0 = pb[(ks + 1) * 2 * WLD]; nb1 = pb[(ks + 1) * 2 * WLD + 32];
            }
            // weights are the A operand (rows = channels), pixels the B operand (columns): a lane then owns four
            // CONSECUTIVE channels of its pixel per accumulator quad -> 16-byte stores
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        if (more) stash(patch[(it & 1) ^ 1]);  // that buffer was last read one tile ago (barrier in between)
        // acc[cb][pr][4g + q] of lane (li, h): channel cb*32 + 8g + 4h + q of pixel (row 2*wave + pr, column li)
        const int n = t / (tx_n * ty_n), rem = t - n * tx_n * ty_n;
        const int oh0 = (rem / tx_n) * ST_TH, ow0 = (rem % tx_n) * ST_TW;
        const int ow = ow0 + li;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int oh = oh0 + 2 * wave + pr;
            if (oh >= OH || ow >= OW) continue;
            float* o = out + ((int64_t)(n * OH + oh) * OW + ow) * 64 + 4 * h;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = cb * 32 + 8 * g;
                    const float4 sc = *reinterpret_cast<const float4*>(ssl + ch + 4 * h);
                    const float4 sh = *reinterpret_cast<const float4*>(ssl + 64 + ch + 4 * h);
                    float4 v;
                    v.x = fmaxf(acc[cb][pr][4 * g + 0] * sc.x + sh.x, 0.f);
                    v.y = fmaxf(acc[cb][pr][4 * g + 1] * sc.y + sh.y, 0.f);
                    v.z = fmaxf(acc[cb][pr][4 * g + 2] * sc.z + sh.z, 0.f);
                    v.w = fmaxf(acc[cb][pr][4 * g + 3] * sc.w + sh.w, 0.f);
                    *reinterpret_cast<float4*>(o + ch) = v;
                }
        }
        __syncthreads();
    }
}

template <typename T>      // T: storage cell of the output map (float | erd::bf16s)
__global__ __launch_bounds__(256) void maxpool_kernel(const float4* __restrict__ in, T* __restrict__ out, int N,
                                                      int H, int W, int C4, int OH, int OW) {
    const int64_t total = (int64_t)N * OH * OW * C4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = i % C4;
        int64_t r = i / C4;
        const int ow = r % OW; r /= OW;
        const int oh = r % OH;
        const int n = r / OH;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ih = oh * 2 - 1 + dy;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int iw = ow * 2 - 1 + dx;
                if ((unsigned)iw >= (unsigned)W) continue;
                const float4 v = in[((int64_t)(n * H + ih) * W + iw) * C4 + c];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        erd::st4(out + i * 4, m);
    }
}

__global__ void bn_fold_kernel(const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ m,
                               const float* __restrict__ v, float eps, float* __restrict__ scale,
                               float* __restrict__ shift, int64_t n) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i < n) {
        const float s = g[i] * (1.0f / sqrtf(v[i] + eps));
        scale[i] = s;
        shift[i] = b[i] - m[i] * s;
    }
}

// dz = dy * (y>0) (dz may alias dy); colsum[c] += sum over rows of dz.  rows are [npix][C] with an image
// stride (level views of [N][A][C] buffers): row r -> img r / rows_per_img.
// Block (x = row range, y = column group of C4w float4s): thread (column c4 = tid % C4w, row lane = tid / C4w) streams
// float4s down its column, 4 rows in flight; the block combines its row lanes through LDS and issues ONE atomic per
// channel.  Same-address float atomics retire at ~6 per microsecond, so the grid is cut into 64-channel column groups
// x long row ranges (a few hundred adds per channel) rather than into many short full-width row ranges.
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_colsum_kernel(const T* __restrict__ y, const T* dy, T* dz,
                                                               int64_t npix, int C, int64_t nstride,
                                                               int64_t rows_per_img, float* __restrict__ colsum,
                                                               int use_relu, int rows_per_block, int C4w) {
    __shared__ float4 red[256];
    const int lanes = 256 / C4w;
    const int rl = threadIdx.x / C4w;
    const bool dense = nstride == rows_per_img * C;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = min(npix, r_begin + rows_per_block);
    {
        const int c4 = blockIdx.y * C4w + threadIdx.x % C4w;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rl < lanes) {
            auto offset = [&](int64_t r) -> int64_t {
                if (dense) return r * C + c4 * 4;
                const int64_t img = r / rows_per_img;
                return img * nstride + (r - img * rows_per_img) * C + c4 * 4;
            };
            int64_t r = r_begin + rl;
            for (; r + 3 * lanes < r_end; r += 4 * lanes) {
                int64_t o[4];
                float4 g[4], yy[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[q] = offset(r + q * lanes);
                    g[q] = erd::ld4(dy + o[q]);
                    if (use_relu) yy[q] = erd::ld4(y + o[q]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (use_relu) {
                        g[q].x = yy[q].x > 0.f ? g[q].x : 0.f; g[q].y = yy[q].y > 0.f ? g[q].y : 0.f;
                        g[q].z = yy[q].z > 0.f ? g[q].z : 0.f; g[q].w = yy[q].w > 0.f ? g[q].w : 0.f;
                        erd::st4(dz + o[q], g[q]);
                    }
                    s.x += g[q].x; s.y += g[q].y; s.z += g[q].z; s.w += g[q].w;
                }
            }
            for (; r < r_end; r += lanes) {
                const int64_t o = offset(r);
                float4 g = erd::ld4(dy + o);
                if (use_relu) {
                    const float4 yy = erd::ld4(y + o);
                    g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
                    g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
                    erd::st4(dz + o, g);
                }
                s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            }
        }
        if (colsum) {
            __syncthreads();
            red[threadIdx.x] = s;
            __syncthreads();
            if (threadIdx.x < C4w) {
                float4 t = red[threadIdx.x];
                for (int j = 1; j < lanes; ++j) {
                    const float4 v = red[threadIdx.x + j * C4w];
                    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
                }
                atomicAdd(colsum + c4 * 4 + 0, t.x);
                atomicAdd(colsum + c4 * 4 + 1, t.y);
                atomicAdd(colsum + c4 * 4 + 2, t.z);
                atomicAdd(colsum + c4 * 4 + 3, t.w);
            }
        }
    }
}

// dbeta: [copies][C] replicated column sums (the input-gradient epilogues add into row blockIdx & (copies - 1)).  32 channels per
// workgroup, eight row lanes each: lane g folds rows g, g + 8, ... with four loads in flight (a single thread per channel walking 64
// rows is a chain of dependent L2 round trips: 20 us per launch on the trailing stream), then thread (0, c) adds the eight shares in
// row-lane order -- for copies <= 8 the sum order of the one-thread loop this replaces.
__global__ void __launch_bounds__(256) bn_dgamma_kernel(const float* __restrict__ rowdot, const float* __restrict__ dbeta, int copies,
                                                        const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta_out, int accumulate, int C) {
    __shared__ float part[8][32];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;
    float s = 0.f;
    if (i < C) {
        int r = g;
        for (; r + 24 < copies; r += 32) {
            const float a0 = dbeta[(int64_t)r * C + i], a1 = dbeta[(int64_t)(r + 8) * C + i];
            const float a2 = dbeta[(int64_t)(r + 16) * C + i], a3 = dbeta[(int64_t)(r + 24) * C + i];
            s += (a0 + a1) + (a2 + a3);
        }
        for (; r < copies; r += 8) s += dbeta[(int64_t)r * C + i];
    }
    part[g][c] = s;
    __syncthreads();
    if (g == 0 && i < C) {
        float db = part[0][c];
#pragma unroll
        for (int j = 1; j < 8; ++j) db += part[j][c];
        if (dbeta_out) dbeta_out[i] = db;
        const float v = (1.0f / sqrtf(var[i] + eps)) * (rowdot[i] - mean[i] * db);
        dgamma[i] = accumulate ? dgamma[i] + v : v;
    }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(G groups) + ReLU over [N][A][C] with per-(image, level, group) statistics.
// stats_ws: double [N][nseg][G][2] (sum, sumsq) accumulated with f64 atomics, then folded to
// mean_rstd float [N][nseg][G][2].
// ------------------------------------------------------------------------------------------------
constexpr int GN_ROWS = 128;       // rows per block of the streaming (apply) kernels
constexpr int GN_STAT_ROWS = 512;  // rows per block of the statistics kernels (fewer atomics per byte)

__device__ __forceinline__ int find_level(const erd_levels& lv, int64_t a) {
    int s = 0;
#pragma unroll 1
    for (; s < lv.nseg - 1; ++s)
        if (a < lv.off[s] + lv.cnt[s]) break;
    return s;
}

// grid: (chunks over A rows, N).  Each chunk of GN_ROWS rows lies inside one level (host pads the
// chunk table per level: blockIdx.x -> (level, first row) through cumulative chunk counts).
struct GnChunks {
    int nseg;
    int start[ERD_MAX_SEG + 1];
};
__device__ __forceinline__ void gn_chunk(const erd_levels& lv, const GnChunks& ch, int bx, int& s, int64_t& r0,
                                         int64_t& r1, int rows = GN_ROWS) {
    s = 0;
#pragma unroll 1
    for (; s < ch.nseg - 1; ++s)
        if (bx < ch.start[s + 1]) break;
    r0 = lv.off[s] + (int64_t)(bx - ch.start[s]) * rows;
    r1 = min(lv.off[s] + lv.cnt[s], r0 + rows);
}

// Statistics kernels: grid (row chunks, N, 4 column groups).  A block owns 16 float4 columns (64 channels = 8 groups)
// and 16 row lanes over a long row range: the same bytes in flight as a full-width block over a quarter of the rows,
// but a quarter of the same-address atomics (those retire at ~6 per microsecond and were the limiter).
template <int C, int G, typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ c, double* __restrict__ stats,
                                                       int64_t A, erd_levels lv, GnChunks ch) {
    constexpr int CPG = C / G;         // channels per group (8)
    static_assert(CPG == 8 && C == 256, "tuned for C=256, G=32");
    const int n = blockIdx.y;
    int s; int64_t r0, r1;
    gn_chunk(lv, ch, blockIdx.x, s, r0, r1, GN_STAT_ROWS);
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;  // 16 columns x 16 row lanes
    const int c4 = blockIdx.z * 16 + cl;
    float sum = 0.f, sq = 0.f;
    int64_t r = r0 + rl;
    for (; r + 48 < r1; r += 64) {
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = erd::ld4(c + ((int64_t)n * A + r + 16 * q) * C + c4 * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sum += v[q].x + v[q].y + v[q].z + v[q].w;
            sq += v[q].x * v[q].x + v[q].y * v[q].y + v[q].z * v[q].z + v[q].w * v[q].w;
        }
    }
    for (; r < r1; r += 16) {
        const float4 v = erd::ld4(c + ((int64_t)n * A + r) * C + c4 * 4);
        sum += v.x + v.y + v.z + v.w;
        sq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    // a group = 2 adjacent float4 columns -> combine lane pairs
    sum += __shfl_xor(sum, 1, 64);
    sq += __shfl_xor(sq, 1, 64);
    __shared__ float red[16][8][2];
    if ((cl & 1) == 0) { red[rl][cl >> 1][0] = sum; red[rl][cl >> 1][1] = sq; }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int g = threadIdx.x >> 1, k = threadIdx.x & 1;
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += (double)red[q][g][k];
        atomicAdd(stats + (((int64_t)n * lv.nseg + s) * G + blockIdx.z * 8 + g) * 2 + k, t);
    }
}

__global__ void gn_finalize_kernel(const double* __restrict__ stats, float* __restrict__ mean_rstd, int N, int G,
                                   erd_levels lv, int cpg, float eps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int total = N * lv.nseg * G;
    if (i < total) {
        const int s = (i / G) % lv.nseg;
        const double m = (double)lv.cnt[s] * cpg;
        const double mean = stats[i * 2] / m;
        double var = stats[i * 2 + 1] / m - mean * mean;
        if (var < 0) var = 0;
        mean_rstd[i * 2] = (float)mean;
        mean_rstd[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

template <int C, int G, typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ c, T* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean_rstd, int64_t A, erd_levels lv,
                                                       GnChunks ch) {
    const int n = blockIdx.y;
    int s; int64_t r0, r1;
    gn_chunk(lv, ch, blockIdx.x, s, r0, r1);
    const int c4 = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int g = c4 >> 1;
    const float2 mr = *reinterpret_cast<const float2*>(mean_rstd + (((int64_t)n * lv.nseg + s) * G + g) * 2);
    const float4 ga = reinterpret_cast<const float4*>(gamma)[c4];
    const float4 be = reinterpret_cast<const float4*>(beta)[c4];
    for (int64_t r = r0 + rl; r < r1; r += 4) {
        const int64_t off = ((int64_t)n * A + r) * C + c4 * 4;
        const float4 v = erd::ld4(c + off);
        float4 o;
        o.x = fmaxf((v.x - mr.x) * mr.y * ga.x + be.x, 0.f);
        o.y = fmaxf((v.y - mr.x) * mr.y * ga.y + be.y, 0.f);
        o.z = fmaxf((v.z - mr.x) * mr.y * ga.z + be.z, 0.f);
        o.w = fmaxf((v.w - mr.x) * mr.y * ga.w + be.w, 0.f);
        erd::st4(y + off, o);
    }
}

// backward pass 1: per (n,level,group) s1 = sum dyh, s2 = sum dyh*xhat (dyh = dy*mask*gamma);
// per channel dgamma += sum dy*mask*xhat, dbeta += sum dy*mask
template <int C, int G, typename T>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const T* __restrict__ c, const T* __restrict__ dy,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ mean_rstd,
                                                           double* __restrict__ stats, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int64_t A, erd_levels lv,
                                                           GnChunks ch) {
    const int n = blockIdx.y;
    int s; int64_t r0, r1;
    gn_chunk(lv, ch, blockIdx.x, s, r0, r1, GN_STAT_ROWS);
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;  // 16 columns x 16 row lanes (see gn_stats_kernel)
    const int c4 = blockIdx.z * 16 + cl;
    const int g = c4 >> 1;
    const float2 mr = *reinterpret_cast<const float2*>(mean_rstd + (((int64_t)n * lv.nseg + s) * G + g) * 2);
    const float4 ga = reinterpret_cast<const float4*>(gamma)[c4];
    const float4 be = reinterpret_cast<const float4*>(beta)[c4];
    float s1 = 0.f, s2 = 0.f;
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = make_float4(0.f, 0.f, 0.f, 0.f);
#define GN_ONE(F)                                                          \
        {                                                                  \
            const float xh = (v.F - mr.x) * mr.y;                          \
            const float dm = (xh * ga.F + be.F > 0.f) ? d.F : 0.f;         \
            dg.F += dm * xh;                                               \
            db.F += dm;                                                    \
            const float dh = dm * ga.F;                                    \
            s1 += dh;                                                      \
            s2 += dh * xh;                                                 \
        }
    int64_t r = r0 + rl;
    for (; r + 16 < r1; r += 32) {          // two rows in flight per thread
        const int64_t off0 = ((int64_t)n * A + r) * C + c4 * 4, off1 = off0 + (int64_t)16 * C;
        const float4 v0 = erd::ld4(c + off0), d0 = erd::ld4(dy + off0);
        const float4 v1 = erd::ld4(c + off1), d1 = erd::ld4(dy + off1);
        { const float4 v = v0, d = d0; GN_ONE(x) GN_ONE(y) GN_ONE(z) GN_ONE(w) }
        { const float4 v = v1, d = d1; GN_ONE(x) GN_ONE(y) GN_ONE(z) GN_ONE(w) }
    }
    for (; r < r1; r += 16) {
        const int64_t off = ((int64_t)n * A + r) * C + c4 * 4;
        const float4 v = erd::ld4(c + off);
        const float4 d = erd::ld4(dy + off);
        GN_ONE(x) GN_ONE(y) GN_ONE(z) GN_ONE(w)
    }
#undef GN_ONE
    s1 += __shfl_xor(s1, 1, 64);
    s2 += __shfl_xor(s2, 1, 64);
    __shared__ float red[16][8][2];
    __shared__ float4 redc[16][16][2];
    if ((cl & 1) == 0) { red[rl][cl >> 1][0] = s1; red[rl][cl >> 1][1] = s2; }
    redc[rl][cl][0] = dg;
    redc[rl][cl][1] = db;
    __syncthreads();
    if (threadIdx.x < 16) {
        const int gg = threadIdx.x >> 1, k = threadIdx.x & 1;
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += (double)red[q][gg][k];
        atomicAdd(stats + (((int64_t)n * lv.nseg + s) * G + blockIdx.z * 8 + gg) * 2 + k, t);
    } else if (threadIdx.x >= 64 && threadIdx.x < 80) {
        const int cc = threadIdx.x - 64;
        float4 a = redc[0][cc][0], b = redc[0][cc][1];
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            a.x += redc[q][cc][0].x; a.y += redc[q][cc][0].y; a.z += redc[q][cc][0].z; a.w += redc[q][cc][0].w;
            b.x += redc[q][cc][1].x; b.y += redc[q][cc][1].y; b.z += redc[q][cc][1].z; b.w += redc[q][cc][1].w;
        }
        const int ch0 = (blockIdx.z * 16 + cc) * 4;
        atomicAdd(dgamma + ch0 + 0, a.x); atomicAdd(dgamma + ch0 + 1, a.y);
        atomicAdd(dgamma + ch0 + 2, a.z); atomicAdd(dgamma + ch0 + 3, a.w);
        atomicAdd(dbeta + ch0 + 0, b.x); atomicAdd(dbeta + ch0 + 1, b.y);
        atomicAdd(dbeta + ch0 + 2, b.z); atomicAdd(dbeta + ch0 + 3, b.w);
    }
}

// backward pass 2: dc = rstd * (dyh - (s1 + xhat*s2)/m)
template <int C, int G, typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ c, const T* __restrict__ dy,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ mean_rstd,
                                                           const double* __restrict__ stats, T* __restrict__ dc,
                                                           int64_t A, erd_levels lv, GnChunks ch) {
    const int n = blockIdx.y;
    int s; int64_t r0, r1;
    gn_chunk(lv, ch, blockIdx.x, s, r0, r1);
    const int c4 = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int g = c4 >> 1;
    const int64_t si = ((int64_t)n * lv.nseg + s) * G + g;
    const float2 mr = *reinterpret_cast<const float2*>(mean_rstd + si * 2);
    const float inv_m = (float)(1.0 / ((double)lv.cnt[s] * (C / G)));
    const float m1 = (float)stats[si * 2] * inv_m, m2 = (float)stats[si * 2 + 1] * inv_m;
    const float4 ga = reinterpret_cast<const float4*>(gamma)[c4];
    const float4 be = reinterpret_cast<const float4*>(beta)[c4];
    for (int64_t r = r0 + rl; r < r1; r += 4) {
        const int64_t off = ((int64_t)n * A + r) * C + c4 * 4;
        const float4 v = erd::ld4(c + off);
        const float4 d = erd::ld4(dy + off);
        float4 o;
#define GN_ONE(F)                                                          \
        {                                                                  \
            const float xh = (v.F - mr.x) * mr.y;                          \
            const float dh = (xh * ga.F + be.F > 0.f) ? d.F * ga.F : 0.f;  \
            o.F = mr.y * (dh - m1 - xh * m2);                              \
        }
        GN_ONE(x) GN_ONE(y) GN_ONE(z) GN_ONE(w)
#undef GN_ONE
        erd::st4(dc + off, o);
    }
}

GnChunks make_chunks(const erd_levels* lv, int rows = GN_ROWS) {
    GnChunks ch;
    ch.nseg = lv->nseg;
    int acc = 0;
    for (int s = 0; s < lv->nseg; ++s) {
        ch.start[s] = acc;
        acc += (int)((lv->cnt[s] + rows - 1) / rows);
    }
    ch.start[lv->nseg] = acc;
    for (int s = lv->nseg + 1; s <= ERD_MAX_SEG; ++s) ch.start[s] = acc;
    return ch;
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void upsample_add_kernel(T* __restrict__ fine, const T* __restrict__ coarse,
                                                           int N, int H, int W, int C4, int h, int w, int64_t fns,
                                                           int64_t cns) {
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = i % C4;
        int64_t r = i / C4;
        const int x = r % W; r /= W;
        const int y = r % H;
        const int n = r / H;
        // F.interpolate(mode='nearest', size=(H,W)): src = floor(dst * (h/H)); exact 2x => dst>>1
        const int sy = min((int)((int64_t)y * h / H), h - 1), sx = min((int)((int64_t)x * w / W), w - 1);
        T* f = fine + n * fns + (((int64_t)y * W + x) * C4 + c) * 4;
        const float4 cv = erd::ld4(coarse + n * cns + (((int64_t)sy * w + sx) * C4 + c) * 4);
        float4 v = erd::ld4(f);
        v.x += cv.x; v.y += cv.y; v.z += cv.z; v.w += cv.w;
        erd::st4(f, v);
    }
}

// adjoint: dcoarse[sy,sx] += sum of dfine over the pixels that map to it
template <typename T>
__global__ __launch_bounds__(256) void upsample_add_bwd_kernel(const T* __restrict__ dfine,
                                                               T* __restrict__ dcoarse, int N, int H, int W,
                                                               int C4, int h, int w, int64_t fns, int64_t cns) {
    const int64_t total = (int64_t)N * h * w * C4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = i % C4;
        int64_t r = i / C4;
        const int sx = r % w; r /= w;
        const int sy = r % h;
        const int n = r / h;
        // fine rows y with floor(y*h/H) == sy
        const int y0 = (int)(((int64_t)sy * H + h - 1) / h), y1 = (int)((((int64_t)sy + 1) * H + h - 1) / h);
        const int x0 = (int)(((int64_t)sx * W + w - 1) / w), x1 = (int)((((int64_t)sx + 1) * W + w - 1) / w);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int y = y0; y < min(y1, H); ++y)
            for (int x = x0; x < min(x1, W); ++x) {
                const float4 v = erd::ld4(dfine + n * fns + (((int64_t)y * W + x) * C4 + c) * 4);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        T* d = dcoarse + n * cns + (((int64_t)sy * w + sx) * C4 + c) * 4;
        float4 o = erd::ld4(d);
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        erd::st4(d, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t rows, int C,
                                                     float* __restrict__ out, int rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int64_t r = r0; r < r1; ++r) s += erd::ld1(x + r * C + c);
        atomicAdd(out + c, s);
    }
}

// y = x * alpha[level]  over [N][A][C] (gfl_head.py:229 `scale(self.gfl_reg(reg_feat))`, one Scale per level)
__global__ __launch_bounds__(256) void level_scale_kernel(const float* __restrict__ x, const float* __restrict__ alphas,
                                                          float* __restrict__ y, int64_t A, int C, erd_levels lv,
                                                          GnChunks ch) {
    const int n = blockIdx.y;
    int s; int64_t r0, r1;
    gn_chunk(lv, ch, blockIdx.x, s, r0, r1);
    const float a = alphas[s];
    const int64_t base = ((int64_t)n * A + r0) * C, cnt = (r1 - r0) * C;
    for (int64_t i = threadIdx.x; i < cnt; i += 256) y[base + i] = x[base + i] * a;
}

__global__ __launch_bounds__(256) void level_scale_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float* __restrict__ alphas, float* __restrict__ dx,
                                                              float* __restrict__ dalphas, int64_t A, int C,
                                                              erd_levels lv, GnChunks ch) {
    const int n = blockIdx.y;
    int s; int64_t r0, r1;
    gn_chunk(lv, ch, blockIdx.x, s, r0, r1);
    const float a = alphas[s];
    const int64_t base = ((int64_t)n * A + r0) * C, cnt = (r1 - r0) * C;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < cnt; i += 256) {
        const float g = dy[base + i];
        acc += g * x[base + i];
        dx[base + i] = g * a;
    }
    acc = erd::wave_sum(acc);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dalphas + s, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sgd_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                  float4* __restrict__ buf, int64_t n4, float lr, float mom, float wd,
                                                  float gs, int first) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pp = p[i];
        const float4 gg = g[i];
        float4 d;
        d.x = gg.x * gs + wd * pp.x; d.y = gg.y * gs + wd * pp.y;
        d.z = gg.z * gs + wd * pp.z; d.w = gg.w * gs + wd * pp.w;
        float4 b;
        if (first) {
            b = d;
        } else {
            b = buf[i];
            b.x = mom * b.x + d.x; b.y = mom * b.y + d.y; b.z = mom * b.z + d.z; b.w = mom * b.w + d.w;
        }
        buf[i] = b;
        pp.x -= lr * b.x; pp.y -= lr * b.y; pp.z -= lr * b.z; pp.w -= lr * b.w;
        p[i] = pp;
    }
}

inline int grid_for(int64_t n, int cap = 2048) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int erd_stem_conv7x7_bn_relu(const float* x, const float* w, const float* scale, const float* shift,
                                        float* out, int N, int H, int W, erd_stream_t stream) {
    ERD_REQUIRE(x && w && scale && shift && out && N > 0, "stem: bad args");
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    const int64_t tiles = (int64_t)((OW + ST_TW - 1) / ST_TW) * ((OH + ST_TH - 1) / ST_TH) * N;
    ERD_REQUIRE(tiles < (1ll << 31), "stem: too many tiles");
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)(tiles < 512 ? tiles : 512)), dim3(256), 0,       // persistent: 2 per CU
                       (hipStream_t)stream, x, w, scale, shift, out, N, H, W, OH, OW);
    return erd::check_launch("stem");
}

// map_type dispatch of the templated kernels: ERD_MAP(T, statement using T)
#define ERD_MAP(map_type, ...)                                   \
    do {                                                         \
        if ((map_type) == ERD_BF16) { using T = erd::bf16s; __VA_ARGS__; } \
        else { using T = float; __VA_ARGS__; }                   \
    } while (0)
#define ERD_MAP_OK(map_type) ((map_type) == ERD_F32 || (map_type) == ERD_BF16)

extern "C" int erd_maxpool3x3s2(const float* in, void* out, int N, int H, int W, int C, int out_type, erd_stream_t stream) {
    ERD_REQUIRE(in && out && C % 4 == 0 && ERD_MAP_OK(out_type), "maxpool: bad args");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * OH * OW * (C / 4);
    ERD_MAP(out_type, hipLaunchKernelGGL(maxpool_kernel<T>, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream,
                                         reinterpret_cast<const float4*>(in), reinterpret_cast<T*>(out), N, H, W, C / 4, OH, OW));
    return erd::check_launch("maxpool");
}

namespace {
// DetDataPreprocessor (data_preprocessor.py:110-183 + mmengine ImgDataPreprocessor): one image [3][h][w] (uint8 or
// fp32, CHW) -> its slot [3][H][W] of the batch: optional channel flip, float, (x - mean) / std, pad_value beyond
// (h, w).  The subtraction and the division stay two IEEE operations (no reciprocal, no fma), so the result is
// bit-identical to the host arithmetic.
template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const T* __restrict__ img, int h, int w, float* __restrict__ out,
                                                         int H, int W, float m0, float m1, float m2, float s0, float s1,
                                                         float s2, int flip, float pad_value) {
    const int64_t plane = (int64_t)H * W;
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= plane) return;
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    const bool inside = y < h && x < w;
    const int64_t src = (int64_t)y * w + x;
    const int64_t sp = (int64_t)h * w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = pad_value;
        if (inside) {
            const float px = (float)img[(flip ? 2 - c : c) * sp + src];
            const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
            v = __fdiv_rn(__fsub_rn(px, mean), sd);
        }
        out[c * plane + i] = v;
    }
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void to_bf16_kernel(const float4* __restrict__ src, uint2* __restrict__ dst, int64_t n4,
                                                      const float* __restrict__ tail_src, unsigned short* __restrict__ tail_dst,
                                                      int ntail) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = src[i];
        const f2 a = {v.x, v.y}, b = {v.z, v.w};
        uint2 o;
        o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf2));
        o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf2));
        dst[i] = o;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) {
        const f2 a = {tail_src[threadIdx.x], 0.f};
        tail_dst[threadIdx.x] = (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf2)) & 0xffffu);
    }
}
}  // namespace

extern "C" int erd_to_bf16(const float* src, void* dst, int64_t n, erd_stream_t stream) {
    ERD_REQUIRE(src && dst && n >= 0, "to_bf16: bad args");
    if (n == 0) return 0;
    const int64_t n4 = n / 4;
    const int ntail = (int)(n - n4 * 4);
    hipLaunchKernelGGL(to_bf16_kernel, dim3(grid_for(n4 > 0 ? n4 : 1, 4096)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(src), reinterpret_cast<uint2*>(dst), n4, src + n4 * 4,
                       reinterpret_cast<unsigned short*>(dst) + n4 * 4, ntail);
    return erd::check_launch("to_bf16");
}

extern "C" int erd_preprocess_image(const void* img, int is_uint8, int h, int w, float* out, int H, int W,
                                    const float* mean3, const float* std3, int flip_channels, float pad_value,
                                    erd_stream_t stream) {
    ERD_REQUIRE(img && out && mean3 && std3, "preprocess: null");
    ERD_REQUIRE(h > 0 && w > 0 && H >= h && W >= w, "preprocess: image %dx%d does not fit the %dx%d slot", h, w, H, W);
    const int64_t plane = (int64_t)H * W;
    const dim3 grid((unsigned)((plane + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (is_uint8)
        hipLaunchKernelGGL(preprocess_kernel<uint8_t>, grid, dim3(256), 0, st, reinterpret_cast<const uint8_t*>(img), h, w,
                           out, H, W, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], flip_channels, pad_value);
    else
        hipLaunchKernelGGL(preprocess_kernel<float>, grid, dim3(256), 0, st, reinterpret_cast<const float*>(img), h, w, out,
                           H, W, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], flip_channels, pad_value);
    return erd::check_launch("preprocess");
}

namespace {
// Resize(keep_ratio) + RandomFlip + DetDataPreprocessor fused: one thread per OUTPUT pixel of the
// padded [3][H][W] slot.  The 8-bit bilinear resize follows OpenCV's fixed-point two-pass scheme (coefficient tables are
// built on the host exactly as the test oracle builds them): rows = S[x0]*a0 + S[x1]*a1 (int, x2048), then
// u8 = (((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2, and only then float, (v - mean) / std.
__global__ __launch_bounds__(256) void resize_normalize_kernel(const uint8_t* __restrict__ src, int sh, int sw,
                                                               const int* __restrict__ xofs, const short* __restrict__ xco,
                                                               const int* __restrict__ yofs, const short* __restrict__ yco,
                                                               int nh, int nw, float* __restrict__ out, int H, int W, float m0,
                                                               float m1, float m2, float s0, float s1, float s2, int flip,
                                                               int swap_rb, float pad_value) {
    const int64_t plane = (int64_t)H * W;
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= plane) return;
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    if (y >= nh || x >= nw) {
        out[i] = pad_value; out[plane + i] = pad_value; out[2 * plane + i] = pad_value;
        return;
    }
    const int xs = flip ? nw - 1 - x : x;                       // flipping the resized image == reading it mirrored
    const int x0 = xofs[xs], x1 = min(x0 + 1, sw - 1);
    const int y0 = yofs[y], y1 = min(y0 + 1, sh - 1);
    const int a0 = xco[2 * xs], a1 = xco[2 * xs + 1], b0 = yco[2 * y], b1 = yco[2 * y + 1];
    const uint8_t* r0 = src + (int64_t)y0 * sw * 3;
    const uint8_t* r1 = src + (int64_t)y1 * sw * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int cs = swap_rb ? 2 - c : c;
        const int t0 = r0[x0 * 3 + cs] * a0 + r0[x1 * 3 + cs] * a1;
        const int t1 = r1[x0 * 3 + cs] * a0 + r1[x1 * 3 + cs] * a1;
        int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        out[c * plane + i] = __fdiv_rn(__fsub_rn((float)v, mean), sd);
    }
}
}  // namespace

extern "C" int erd_resize_normalize(const void* src_hwc_u8, int sh, int sw, const int* xofs, const short* xcoef,
                                    const int* yofs, const short* ycoef, int nh, int nw, float* out, int H, int W,
                                    const float* mean3, const float* std3, int flip, int swap_rb, float pad_value,
                                    erd_stream_t stream) {
    ERD_REQUIRE(src_hwc_u8 && xofs && xcoef && yofs && ycoef && out && mean3 && std3, "resize_normalize: null");
    ERD_REQUIRE(sh > 0 && sw > 0 && nh > 0 && nw > 0 && H >= nh && W >= nw, "resize_normalize: bad sizes");
    const int64_t plane = (int64_t)H * W;
    hipLaunchKernelGGL(resize_normalize_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint8_t*>(src_hwc_u8), sh, sw, xofs, xcoef, yofs, ycoef, nh, nw, out, H, W,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], flip, swap_rb, pad_value);
    return erd::check_launch("resize_normalize");
}

extern "C" int erd_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                           float* scale, float* shift, int64_t n, erd_stream_t stream) {
    ERD_REQUIRE(gamma && beta && mean && var && scale && shift, "bn_fold: null");
    if (n == 0) return 0;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gamma,
                       beta, mean, var, eps, scale, shift, n);
    return erd::check_launch("bn_fold");
}

namespace {
__global__ __launch_bounds__(256) void bn_fold_batch_kernel(const erd_bn_fold_item* __restrict__ items) {
    const erd_bn_fold_item it = items[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < it.n) {
        const float s = it.gamma[i] * (1.0f / sqrtf(it.var[i] + it.eps));      // (the arithmetic of bn_fold_kernel)
        it.scale[i] = s;
        it.shift[i] = it.beta[i] - it.mean[i] * s;
    }
}
}  // namespace

extern "C" int erd_bn_fold_batch(const erd_bn_fold_item* items_dev, int nitems, int max_n, erd_stream_t stream) {
    ERD_REQUIRE(items_dev && nitems >= 0 && max_n >= 0, "bn_fold_batch: bad args");
    if (nitems == 0 || max_n == 0) return 0;
    hipLaunchKernelGGL(bn_fold_batch_kernel, dim3((unsigned)((max_n + 255) / 256), (unsigned)nitems), dim3(256), 0,
                       (hipStream_t)stream, items_dev);
    return erd::check_launch("bn_fold_batch");
}

extern "C" int erd_relu_bwd_colsum(const void* y, const void* dy, void* dz, int64_t npix, int C,
                                   int64_t nstride_rows, int64_t rows_per_img, float* colsum, int use_relu,
                                   int map_type, erd_stream_t stream) {
    ERD_REQUIRE(dy && C % 4 == 0 && (!use_relu || (y && dz)) && ERD_MAP_OK(map_type), "relu_bwd: bad args");
    if (npix == 0) return 0;
    const int C4 = C / 4;
    ERD_REQUIRE(C4 % 16 == 0 || 256 % C4 == 0, "relu_bwd: C=%d unsupported", C);
    const int cw4 = C4 % 16 == 0 ? 16 : C4, gy = C4 / cw4, lanes = 256 / cw4;
    static const int blocks_env = getenv("ERD_RELU_BLOCKS") ? atoi(getenv("ERD_RELU_BLOCKS")) : 0;   // tuning aid
    const int want = blocks_env > 0 ? blocks_env : 512;
    const int unit = 4 * lanes;                                     // rows one pass of the unrolled loop covers
    int64_t rpb = (npix + (want / gy > 0 ? want / gy : 1) - 1) / (want / gy > 0 ? want / gy : 1);
    rpb = (rpb + unit - 1) / unit * unit;
    ERD_MAP(map_type, hipLaunchKernelGGL(relu_bwd_colsum_kernel<T>, dim3((unsigned)((npix + rpb - 1) / rpb), gy), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)y, (const T*)dy, (T*)dz, npix, C, nstride_rows,
                                         rows_per_img, colsum, use_relu, (int)rpb, cw4));
    return erd::check_launch("relu_bwd_colsum");
}

extern "C" int erd_bn_dgamma(const float* rowdot, const float* dbeta, int copies, const float* mean, const float* var,
                             float eps, float* dgamma, float* dbeta_out, int accumulate, int C, erd_stream_t stream) {
    ERD_REQUIRE(rowdot && dbeta && mean && var && dgamma && copies >= 1, "bn_dgamma: bad args");
    hipLaunchKernelGGL(bn_dgamma_kernel, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, rowdot, dbeta, copies,
                       mean, var, eps, dgamma, dbeta_out, accumulate, C);
    return erd::check_launch("bn_dgamma");
}

extern "C" int erd_gn_relu_fwd(const void* c, void* y, const float* gamma, const float* beta, double* stats_ws,
                               float* mean_rstd, int N, int64_t A, int C, int G, const erd_levels* lv, float eps,
                               int map_type, erd_stream_t stream) {
    ERD_REQUIRE(c && y && gamma && beta && stats_ws && mean_rstd && lv && ERD_MAP_OK(map_type), "gn_fwd: bad args");
    ERD_REQUIRE(C == 256 && G == 32, "gn_fwd: only C=256,G=32 (gfl_head.py:109-110) is built");
    hipStream_t st = (hipStream_t)stream;
    const GnChunks ch = make_chunks(lv);
    const GnChunks chs = make_chunks(lv, GN_STAT_ROWS);
    const int nst = N * lv->nseg * G;
    ERD_ZERO_ASYNC(stats_ws, sizeof(double) * 2 * nst, st);
#ifndef ERD_GN_NOSTATS      // timing probe (tools/build_probe.sh): the upper bound of what statistics fused into the producing
                            // convolution's output stage could save -- the pass is simply not run (results are wrong)
    ERD_MAP(map_type, hipLaunchKernelGGL((gn_stats_kernel<256, 32, T>), dim3(chs.start[lv->nseg], N, 4), dim3(256), 0, st,
                                         (const T*)c, stats_ws, A, *lv, chs));
#endif
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((nst + 255) / 256), dim3(256), 0, st, stats_ws, mean_rstd, N, G, *lv,
                       C / G, eps);
    ERD_MAP(map_type, hipLaunchKernelGGL((gn_apply_kernel<256, 32, T>), dim3(ch.start[lv->nseg], N), dim3(256), 0, st,
                                         (const T*)c, (T*)y, gamma, beta, mean_rstd, A, *lv, ch));
    return erd::check_launch("gn_relu_fwd");
}

// the normalisation pass alone: mean_rstd[N][nseg][G][2] comes from the producing convolution (erd_wino_conv3x3_x3_gn)
extern "C" int erd_gn_relu_apply(const void* c, void* y, const float* gamma, const float* beta, const float* mean_rstd, int N,
                                 int64_t A, int C, int G, const erd_levels* lv, int map_type, erd_stream_t stream) {
    ERD_REQUIRE(c && y && gamma && beta && mean_rstd && lv && ERD_MAP_OK(map_type), "gn_apply: bad args");
    ERD_REQUIRE(C == 256 && G == 32, "gn_apply: only C=256,G=32 (gfl_head.py:109-110) is built");
    const GnChunks ch = make_chunks(lv);
    ERD_MAP(map_type, hipLaunchKernelGGL((gn_apply_kernel<256, 32, T>), dim3(ch.start[lv->nseg], N), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)c, (T*)y, gamma, beta, mean_rstd, A, *lv, ch));
    return erd::check_launch("gn_relu_apply");
}

extern "C" int erd_gn_relu_bwd(const void* c, const void* dy, const float* gamma, const float* beta,
                               const float* mean_rstd, double* stats_ws, void* dc, float* dgamma, float* dbeta, int N,
                               int64_t A, int C, int G, const erd_levels* lv, int map_type, erd_stream_t stream) {
    ERD_REQUIRE(c && dy && gamma && beta && mean_rstd && stats_ws && dc && dgamma && dbeta && lv && ERD_MAP_OK(map_type),
                "gn_bwd: bad args");
    ERD_REQUIRE(C == 256 && G == 32, "gn_bwd: only C=256,G=32 is built");
    hipStream_t st = (hipStream_t)stream;
    const GnChunks ch = make_chunks(lv);
    const GnChunks chs = make_chunks(lv, GN_STAT_ROWS);
    const int nst = N * lv->nseg * G;
    ERD_ZERO_ASYNC(stats_ws, sizeof(double) * 2 * nst, st);
    ERD_MAP(map_type, hipLaunchKernelGGL((gn_bwd_stats_kernel<256, 32, T>), dim3(chs.start[lv->nseg], N, 4), dim3(256), 0, st,
                                         (const T*)c, (const T*)dy, gamma, beta, mean_rstd, stats_ws, dgamma, dbeta, A, *lv, chs));
    ERD_MAP(map_type, hipLaunchKernelGGL((gn_bwd_apply_kernel<256, 32, T>), dim3(ch.start[lv->nseg], N), dim3(256), 0, st,
                                         (const T*)c, (const T*)dy, gamma, beta, mean_rstd, stats_ws, (T*)dc, A, *lv, ch));
    return erd::check_launch("gn_relu_bwd");
}

extern "C" int erd_upsample2x_add(void* fine, const void* coarse, int N, int H, int W, int C, int h, int w,
                                  int64_t fns, int64_t cns, int map_type, erd_stream_t stream) {
    ERD_REQUIRE(fine && coarse && C % 4 == 0 && fns % 4 == 0 && cns % 4 == 0 && ERD_MAP_OK(map_type), "upsample_add: bad args");
    const int64_t total = (int64_t)N * H * W * (C / 4);
    ERD_MAP(map_type, hipLaunchKernelGGL(upsample_add_kernel<T>, dim3(grid_for(total, 4096)), dim3(256), 0, (hipStream_t)stream,
                                         (T*)fine, (const T*)coarse, N, H, W, C / 4, h, w, fns, cns));
    return erd::check_launch("upsample_add");
}

extern "C" int erd_upsample2x_add_bwd(const void* dfine, void* dcoarse, int N, int H, int W, int C, int h, int w,
                                      int64_t fns, int64_t cns, int map_type, erd_stream_t stream) {
    ERD_REQUIRE(dfine && dcoarse && C % 4 == 0 && fns % 4 == 0 && cns % 4 == 0 && ERD_MAP_OK(map_type), "upsample_add_bwd: bad args");
    const int64_t total = (int64_t)N * h * w * (C / 4);
    ERD_MAP(map_type, hipLaunchKernelGGL(upsample_add_bwd_kernel<T>, dim3(grid_for(total, 4096)), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)dfine, (T*)dcoarse, N, H, W, C / 4, h, w, fns, cns));
    return erd::check_launch("upsample_add_bwd");
}

extern "C" int erd_colsum(const void* x, int64_t rows, int C, float* out, int accumulate, int map_type, erd_stream_t stream) {
    ERD_REQUIRE(x && out && C > 0 && ERD_MAP_OK(map_type), "colsum: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) ERD_ZERO_ASYNC(out, sizeof(float) * C, st);
    if (rows == 0) return 0;
    int rpb = 64;
    while ((rows + rpb - 1) / rpb > 2048) rpb *= 2;
    ERD_MAP(map_type, hipLaunchKernelGGL(colsum_kernel<T>, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, st,
                                         (const T*)x, rows, C, out, rpb));
    return erd::check_launch("colsum");
}

extern "C" int erd_level_scale(const float* x, const float* alphas, float* y, int N, int64_t A, int C,
                               const erd_levels* lv, erd_stream_t stream) {
    ERD_REQUIRE(x && alphas && y && lv, "level_scale: null");
    const GnChunks ch = make_chunks(lv);
    hipLaunchKernelGGL(level_scale_kernel, dim3(ch.start[lv->nseg], N), dim3(256), 0, (hipStream_t)stream, x, alphas, y,
                       A, C, *lv, ch);
    return erd::check_launch("level_scale");
}

extern "C" int erd_level_scale_bwd(const float* x, const float* dy, const float* alphas, float* dx, float* dalphas,
                                   int N, int64_t A, int C, const erd_levels* lv, erd_stream_t stream) {
    ERD_REQUIRE(x && dy && alphas && dx && dalphas && lv, "level_scale_bwd: null");
    hipStream_t st = (hipStream_t)stream;
    const GnChunks ch = make_chunks(lv);
    ERD_ZERO_ASYNC(dalphas, sizeof(float) * lv->nseg, st);
    hipLaunchKernelGGL(level_scale_bwd_kernel, dim3(ch.start[lv->nseg], N), dim3(256), 0, st, x, dy, alphas, dx,
                       dalphas, A, C, *lv, ch);
    return erd::check_launch("level_scale_bwd");
}

extern "C" int erd_sgd_momentum(float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                                float weight_decay, float grad_scale, int first_step, erd_stream_t stream) {
    ERD_REQUIRE(p && g && buf && n % 4 == 0, "sgd: bad args (n must be a multiple of 4)");
    if (n == 0) return 0;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n / 4, 4096)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(p), reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(buf),
                       n / 4, lr, momentum, weight_decay, grad_scale, first_step);
    return erd::check_launch("sgd");
}
