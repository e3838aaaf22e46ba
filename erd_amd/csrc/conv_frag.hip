// 1x1 convolutions in the three-limb form with the WEIGHTS AS MFMA FRAGMENTS STRAIGHT FROM L2 (K = Cin >= 256).
//
// Why a third implicit-GEMM kernel.  Two measurements of round 4: (i) a wave gets 11-15 B/clk out of the LDS
// whatever the other waves do, and the stream-K three-limb kernel (conv_mfma.hip) issues 38 LDS instructions per wave and
// 32-channel slice -- 24 of them reads of weight-plane fragments that every one of its four waves (a 4 x 1 grid: each owns 32
// pixel rows x all 128 couts) fetches separately; (ii) 1 KB-contiguous wave loads from L2 run at 94 B/clk/CU
// (tools/wino_x3_skeleton.hip), four times what the 128-byte row segments of the global -> LDS operand loaders reach
// (tools/load_path.hip).  So: the weight limb planes are stored PRE-TILED as MFMA B-fragments
// (erd_weight_frag_x3: [plane][cout / 32][K / 16][lane][8 bf16], 1 KB per fragment) and go global -> registers with no LDS in
// between; only the activations are staged through LDS (fp32, 32-channel slices, the stream-K kernel's swizzle).  The four
// waves form a 2 x 2 grid (64 rows x 64 couts each): a wave needs the fragments of two cout blocks only (6 KB per k16 step
// instead of 12), splits two row blocks of activations (88 VALU per 24 MFMAs) and reads four 16-byte LDS fragments per step.
//   * MFMA sequence per accumulator: the stream-K kernel's (k16 steps ascending; hi x lo, hi x mid, hi x hi, mid x mid,
//     mid x hi, lo x hi) -- results are bit-identical to it (tests/test_gpu_frag.py);
//   * weight fragments: a ring two k16 steps deep (2 x 2 cout blocks x 3 planes x 4 registers), re-loaded a whole step ahead;
//   * epilogue: the wave's 64 x 64 block through its own LDS block into 256-byte row segments (scale / shift / residual /
//     ReLU / mask / column sums fused, as in the other kernels);
//   * work: (pixel tile, 128-cout block) tiles, whole K per tile, a persistent grid of two workgroups per CU over equal
//     contiguous tile ranges in the XCD-aware order.
// Serves: one tap (1x1), Cin % 32 == 0, Cin >= 256, Cout % 64 == 0, fp32 maps, stride 1 or 2, forward and the input-gradient
// forms; erd_conv_igemm dispatches here when erd_conv_desc::w_x3f is set (ERD_FRAG=0: off).
#include <algorithm>
#include "erd_common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x7fffffffu;

__device__ __forceinline__ u4v buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ u4v buf_load16_s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

struct FRow {
    int in_off;    // element offset of the row's input pixel (channel 0); -1: zeros
    int out_off;   // element offset of the row's output pixel; -1: row past the end
};

constexpr int WSLD = 68;                               // floats per staged row of a wave's 64 x 64 block
constexpr int A_STAGE = 128 * 8 * 16;                  // bytes of one activation slice in LDS: 128 rows x 32 channels fp32

__global__ __launch_bounds__(256, 2) void conv_frag_x3_kernel(const erd_conv_desc p, const int mtiles, const int ntn, const int xcd_order) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: operand region [2][128 rows][8 chunks] float4 (32 KB), re-used by the epilogue's four wave blocks (4 x 64 x WSLD x 4 = 68 KB);
    //      behind them the row table [128] and the column-sum scratch [2 row halves][128]
    constexpr int STAGE_B = 4 * 64 * WSLD * 4;
    float4* As = reinterpret_cast<float4*>(smem);
    FRow* rows = reinterpret_cast<FRow*>(smem + STAGE_B);
    float* red = reinterpret_cast<float*>(rows + 128);

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                      // (wave-uniform: the weight-fragment offsets are scalar)
    const int G = gridDim.x;
    int wg = blockIdx.x;
    if (xcd_order) {
        const int q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long long T = (long long)mtiles * ntn;
    const long long t_begin = T * wg / G, t_end = T * (wg + 1) / G;
    const int K = p.Cin, nks = K / 16, nsl = K / 32, ncb = p.Cout / 32;
    const unsigned per_plane = (unsigned)ncb * (unsigned)nks * 1024u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_x3f), 0, (int)(3u * per_plane), 0x00020000);
    const unsigned w_lane = (unsigned)lane * 16u;
    auto swzc = [](int row, int c) { return c ^ ((row >> 1) & 7); };
    const int chunk = tid & 7, r0 = tid >> 3;                    // activation staging: row r0 + 32 j, 16-byte chunk `chunk`
    float* const cs_row = p.colsum ? p.colsum + (p.colsum_copies > 1 ? (int64_t)(blockIdx.x & (p.colsum_copies - 1)) * p.Cout : 0) : nullptr;
    const float alpha_dummy = 1.f;

#pragma unroll 1
    for (long long t = t_begin; t < t_end; ++t) {
        const int nt = (int)(t % ntn);
        int mt = (int)(t / ntn);
        int s = 0;
#pragma unroll 1
        for (; s < p.nseg - 1; ++s) {
            const int M = p.seg[s].N * p.seg[s].GH * p.seg[s].GW;
            const int tiles = (M + 127) / 128;
            if (mt < tiles) break;
            mt -= tiles;
        }
        const erd_conv_seg& sg = p.seg[s];
        __syncthreads();                       // the previous tile's epilogue is done with LDS
        if (tid < 128) {
            const int GHW = sg.GH * sg.GW, M = sg.N * GHW, m = mt * 128 + tid;
            FRow ri;
            ri.in_off = -1;
            ri.out_off = -1;
            if (m < M) {
                const int n = m / GHW, rem = m - n * GHW, a = rem / sg.GW, b = rem - a * sg.GW;
                const int ih = a * p.in_stride + p.dy[0], iw = b * p.in_stride + p.dx[0];
                if ((unsigned)ih < (unsigned)sg.IH && (unsigned)iw < (unsigned)sg.IW)
                    ri.in_off = (int)(n * sg.in_nstride) + (ih * sg.IW + iw) * p.Cin;
                ri.out_off = (int)(n * sg.out_nstride) + ((a * p.out_stride + p.oy) * sg.OW + (b * p.out_stride + p.ox)) * p.Cout;
            }
            rows[tid] = ri;
        }
        __syncthreads();
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
            erd::uniform_ptr(const_cast<float*>(sg.in)), 0, erd::uniform_int((int)((long long)sg.N * sg.in_nstride * 4)), 0x00020000);
        unsigned a_base[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int io = rows[r0 + 32 * j].in_off;
            a_base[j] = io < 0 ? OOB : (unsigned)(io + chunk * 4) * 4u;
        }
        // this wave's weight fragments: cout blocks 4 nt + 2 wn + {0, 1}; byte offset of (cout block, k16 step 0) inside a plane
        const unsigned wcb = (unsigned)((nt * 4 + wn * 2) * nks) * 1024u;
        u4v ub[2][2][3];                        // [k16 step parity][cout block][plane]
        auto load_w = [&](const int slot, const int ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    ub[slot][j][pl] = buf_load16_s(rs_w, w_lane, (unsigned)pl * per_plane + wcb + (unsigned)(j * nks + ks) * 1024u);
        };
        float4 ra[4];
        auto load_a = [&](int sl) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u4v v = buf_load16(rs_in, a_base[j] == OOB ? OOB : a_base[j] + (unsigned)sl * 128u);
                ra[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            }
        };
        auto store_a = [&](int buf) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = r0 + 32 * j;
                As[buf * 1024 + row * 8 + swzc(row, chunk)] = ra[j];
            }
        };
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        load_a(0);
        load_w(0, 0);
        load_w(1, 1);
        store_a(0);
        __syncthreads();
#pragma unroll 1
        for (int sl = 0; sl < nsl; ++sl) {
            const int buf = sl & 1;
            const bool more = sl + 1 < nsl;
            if (more) load_a(sl + 1);
            const float4* Ab = As + buf * 1024;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {             // the two k16 steps of the slice; weight ring slot = kk
                float x[2][8];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int arow = wm * 64 + i * 32 + li, ca = 4 * kk + 2 * h;
                    const float4 x0 = Ab[arow * 8 + swzc(arow, ca)], x1 = Ab[arow * 8 + swzc(arow, ca + 1)];
                    x[i][0] = x0.x; x[i][1] = x0.y; x[i][2] = x0.z; x[i][3] = x0.w; x[i][4] = x1.x; x[i][5] = x1.y; x[i][6] = x1.z; x[i][7] = x1.w;
                }
                u4v ph[2], pm[2], pl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        unsigned hi, mid, lo;
                        erd::limbs3_pair(x[i][2 * e], x[i][2 * e + 1], hi, mid, lo);
                        ph[i][e] = hi; pm[i][e] = mid; pl[i][e] = lo;
                    }
                // per accumulator: (hi, lo) (hi, mid) (hi, hi) (mid, mid) (mid, hi) (lo, hi) -- activation limb first, weight plane second
#define ERD_FX3(AL, WP)                                                                                                   \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AL[i]),                 \
                                                                        __builtin_bit_cast(bf16x8, ub[kk][j][WP]), acc[i][j], 0, 0, 0);
                ERD_FX3(ph, 2) ERD_FX3(ph, 1) ERD_FX3(ph, 0) ERD_FX3(pm, 1) ERD_FX3(pm, 0) ERD_FX3(pl, 0)
#undef ERD_FX3
                // the slot is free: the same step parity of the next slice (two k16 steps = a whole slice ahead)
                if (more) load_w(kk, 2 * (sl + 1) + kk);
            }
            if (more) store_a(buf ^ 1);
            __syncthreads();
        }

        // ---- epilogue: the wave's 64 x 64 block -> its own LDS block -> 256-byte row segments -------------------------------------
        float* wst = reinterpret_cast<float*>(smem) + wave * 64 * WSLD;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) wst[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * WSLD + j * 32 + li] = acc[i][j][r];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int c4 = lane & 15, rsub = lane >> 4;                  // 16 float4 columns x 4 rows per wave instruction
        const int co = nt * 128 + wn * 64 + c4 * 4;
        const float* res = sg.res;
        const float* msk = sg.mask;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + co);
        if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + co);
        const float alpha = *(sg.alpha ? sg.alpha : &alpha_dummy);
        float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int rr = q * 4 + rsub;
            const int oo = rows[wm * 64 + rr].out_off;
            if (oo < 0) continue;
            float4 v = *reinterpret_cast<const float4*>(wst + rr * WSLD + c4 * 4);
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
            if (sg.alpha) { v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha; }
            if (res) { const float4 rv = *reinterpret_cast<const float4*>(res + oo + co); v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (msk) {
                const float4 mv = *reinterpret_cast<const float4*>(msk + oo + co);
                v.x = mv.x > 0.f ? v.x : 0.f; v.y = mv.y > 0.f ? v.y : 0.f;
                v.z = mv.z > 0.f ? v.z : 0.f; v.w = mv.w > 0.f ? v.w : 0.f;
            }
            *reinterpret_cast<float4*>(sg.out + oo + co) = v;
            csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w;
        }
        if (cs_row) {      // lanes that share a column group (lane bits 4, 5), the two row halves through LDS, one atomic per channel
            csum.x += __shfl_xor(csum.x, 16, 64); csum.y += __shfl_xor(csum.y, 16, 64);
            csum.z += __shfl_xor(csum.z, 16, 64); csum.w += __shfl_xor(csum.w, 16, 64);
            csum.x += __shfl_xor(csum.x, 32, 64); csum.y += __shfl_xor(csum.y, 32, 64);
            csum.z += __shfl_xor(csum.z, 32, 64); csum.w += __shfl_xor(csum.w, 32, 64);
            if (lane < 16) *reinterpret_cast<float4*>(red + wm * 128 + wn * 64 + lane * 4) = csum;
            __syncthreads();
            if (tid < 128) atomicAdd(cs_row + nt * 128 + tid, red[tid] + red[128 + tid]);
        }
    }
}

// dst[plane][co / 32][k / 16][lane = (k % 16 / 8) * 32 + co % 32][k % 8] = limb `plane` of w[co * wrow + k]   (one thread per value)
__global__ __launch_bounds__(256) void weight_frag_x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int Cout, int K,
                                                              int wrow) {
    const int64_t idx = blockIdx.x * 256ll + threadIdx.x;
    if (idx >= (int64_t)Cout * K) return;
    const int co = (int)(idx / K), k = (int)(idx % K);
    unsigned short hi, mid, lo;
    erd::limbs3(w[(int64_t)co * wrow + k], hi, mid, lo);
    const int nks = K / 16;
    const int64_t per_plane = (int64_t)(Cout / 32) * nks * 512;
    const int64_t o = (((int64_t)(co / 32) * nks + k / 16) * 64 + ((k % 16) / 8) * 32 + (co % 32)) * 8 + (k % 8);
    dst[o] = hi;
    dst[per_plane + o] = mid;
    dst[2 * per_plane + o] = lo;
}

int num_cus_frag() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

}  // namespace

namespace erd {

static int g_frag_on = -1;

int conv_frag_enable(int on) {
    if (g_frag_on < 0) g_frag_on = getenv("ERD_FRAG") ? atoi(getenv("ERD_FRAG")) : 1;
    const int prev = g_frag_on;
    if (on >= 0) g_frag_on = on ? 1 : 0;
    return prev;
}

bool conv_frag_x3_ok(const erd_conv_desc* d) {
    if (!conv_frag_enable(-1) || !d->w_x3f || d->w_bf16 || d->in_bf16 || d->out_bf16 || d->ntaps != 1) return false;
    if (d->Cin % 32 != 0 || d->Cin < 256 || d->Cout % 128 != 0 || d->wk[0] != 0) return false;
    for (int s = 0; s < d->nseg; ++s)
        if (d->seg[s].ntaps > 0) return false;
    return true;
}

int conv_frag_x3(const erd_conv_desc* d, hipStream_t st) {
    int mtiles = 0;
    for (int s = 0; s < d->nseg; ++s) mtiles += (int)(((int64_t)d->seg[s].N * d->seg[s].GH * d->seg[s].GW + 127) / 128);
    const int ntn = d->Cout / 128;
    if (mtiles == 0) return 0;
    const size_t lds = (size_t)4 * 64 * WSLD * 4 + 128 * sizeof(FRow) + 2 * 128 * 4;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_frag_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    static const int xcd = getenv("ERD_XCD") ? atoi(getenv("ERD_XCD")) : 1;
    const long long T = (long long)mtiles * ntn;
    const int G = (int)std::min<long long>(T, 2ll * num_cus_frag());
    hipLaunchKernelGGL(conv_frag_x3_kernel, dim3(G), dim3(256), lds, st, *d, mtiles, ntn, xcd ? 1 : 0);
    return erd::check_launch("conv_frag_x3");
}

}  // namespace erd

extern "C" int erd_conv_frag_enable(int on) { return erd::conv_frag_enable(on); }

extern "C" size_t erd_weight_frag_x3_elems(int Cout, int K) { return (size_t)3 * Cout * K; }

extern "C" int erd_weight_frag_x3(const float* w, void* dst, int Cout, int K, int wrow, erd_stream_t stream) {
    ERD_REQUIRE(w && dst && Cout > 0 && K > 0 && Cout % 32 == 0 && K % 16 == 0 && wrow >= K, "weight_frag_x3: Cout=%d K=%d wrow=%d", Cout, K, wrow);
    const int64_t n = (int64_t)Cout * K;
    hipLaunchKernelGGL(weight_frag_x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<unsigned short*>(dst), Cout, K, wrow);
    return erd::check_launch("weight_frag_x3");
}
