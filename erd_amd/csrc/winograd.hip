// Winograd F(2x2, 3x3) convolution for the stride-1 3x3 layers, fp32 on the gfx950 matrix cores.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray 2016; 2.25x fewer multiplications than direct)
//
// The 16 transform positions xi = (i, j) are 16 independent GEMMs  M_xi[tile][cout] = sum_cin V_xi[tile][cin] U_xi[cout][cin].
// One workgroup (256 threads, 4 waves) owns 32 output tiles (4 x 8 tiles = 8 x 16 output pixels of one image of one
// level) x 64 output channels; wave w owns the four positions of transform row i = w.  Per 16-channel K slice:
//   global -> LDS   : the raw 10 x 18 pixel patch (zeros outside the image come from the buffer out-of-range rule)
//   LDS -> LDS      : B^T d B per (tile, 4 channels) into V[xi][tile][16 cin] rows (64 B, XOR swizzle)
//   MFMA            : A fragments from V (ds_read_b128), B fragments straight from global: U is stored pre-tiled
//                     [xi][cout/32][cin/4][32][4] so that a wave's fragment load is one contiguous 1 KB
// and after the K loop the accumulators go through LDS once more for A^T M A, the optional scale/shift/ReLU epilogue
// and coalesced stores.  The input gradient of the same layers is this kernel run on dz with the flipped, transposed
// weights.  (The weight gradient stays a direct GEMM over pixels.)
#include "erd_common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x7fffffffu;

__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ float4 buf_load16_s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

constexpr int TBH = 4, TBW = 8;               // tiles per workgroup: 4 rows x 8 cols = 32 (one MFMA M-tile)
constexpr int PR = 2 * TBH + 2, PC = 2 * TBW + 2;   // raw patch 10 x 18 pixels
constexpr int KS = 16;                        // input channels per K slice
constexpr int BN = 64;                        // output channels per workgroup
constexpr int RAW_F4 = PR * PC * (KS / 4);    // float4s of the raw patch
constexpr int RCS = 6;                        // LDS chunks per raw pixel: 4 data + 2 pad -> the strided 4x4-patch reads of the
                                              // transform are bank-conflict free (4 would be 8-way)
constexpr int RAW_LDS_F4 = PR * PC * RCS;
constexpr int V_F4 = 16 * 32 * (KS / 4);      // float4s of the 16 transformed tiles
constexpr int RAW2_LDS_F4 = 192 * RCS;       // second-generation kernel: 3 x 256 staged float4s = 192 pixel slots (180 used)

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// U[xi][cout/32][cin/4][cout%32][cin%4] = (G g G^T)[xi] ; g = w[cout][kh][kw][cin]
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout,
                                                          int Cin, int flip) {
    const int64_t idx = blockIdx.x * 256ll + threadIdx.x;
    if (idx >= (int64_t)Cout * Cin) return;
    const int co = (int)(idx / Cin), ci = (int)(idx % Cin);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = w[((int64_t)co * 9 + (flip ? 8 - (a * 3 + b) : a * 3 + b)) * Cin + ci];
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int cob = Cout / 32 + (Cout % 32 ? 1 : 0);
    const int64_t per_xi = (int64_t)cob * (Cin / 4) * 128;
    const int64_t base = ((int64_t)(co / 32) * (Cin / 4) + ci / 4) * 128 + (co % 32) * 4 + (ci % 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float u0 = t[i][0], u1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]), u2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]),
                    u3 = t[i][2];
        U[(i * 4 + 0) * per_xi + base] = u0;
        U[(i * 4 + 1) * per_xi + base] = u1;
        U[(i * 4 + 2) * per_xi + base] = u2;
        U[(i * 4 + 3) * per_xi + base] = u3;
    }
}

struct WinoSeg {
    const float* in;
    float* out;
    const float* res;        // optional, geometry of out: added before ReLU / mask (may alias out)
    const float* mask;       // optional, geometry of out: result zeroed where mask <= 0
    int N, H, W;
    int tbh, tbw;            // workgroup grid of one image: ceil(ceil(H/2)/4) x ceil(ceil(W/2)/8)
    int64_t in_nstride, out_nstride;
    int block0;              // first workgroup (per cout block) of this segment
};
struct WinoDesc {
    int nseg;
    WinoSeg seg[ERD_MAX_SEG];
    const float* U;
    int Cin, Cout;
    const float* scale;
    const float* shift;
    int relu;
    float* colsum;           // optional [colsum_copies][Cout]: += column sums of the stored result
    int colsum_copies;       // 0/1 or a power of two: workgroup b adds into row b mod copies
    int blocks_per_nb;       // workgroups per cout block
    int nitems;              // workgroup items = blocks_per_nb * cout blocks
    int* sched;              // optional {next-item counter, finished-workgroup counter}, zero on entry and on exit
    int dbg;                 // ablation switches (ERD_WINO_DBG): 1 no transform, 2 no weight loads, 4 no MFMA, 8 no raw loads
};

// One (tile block, cout block) work item: where it lives
struct WinoItem {           // all fields are wave-uniform; decode() pins them to scalar registers
    int s, n, y0, x0, cout0, cb0, cb1;
};

__global__ __launch_bounds__(512, 2) void wino_conv_kernel(const WinoDesc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* raw0 = reinterpret_cast<float4*>(smem);         // 2 x [PR][PC][RCS]
    float4* raw1 = raw0 + RAW_LDS_F4;
    float4* Vs0 = raw1 + RAW_LDS_F4;                         // 2 x [16][32 tiles][KS/4] swizzled
    float4* Vs1 = Vs0 + V_F4;
    float* Ms = reinterpret_cast<float*>(smem);              // output staging Z[4][2][32 tiles][64 couts] floats (64 KB)
    int* sh_next = reinterpret_cast<int*>(Vs1 + V_F4);       // next item of this workgroup (dynamic scheduling)

    const int tid = threadIdx.x;
    const int Cin = p.Cin;
    const int cob_all = (p.Cout + 31) / 32;
    const int64_t per_xi = (int64_t)cob_all * (Cin / 4) * 128;            // floats per transform position of U
    const int nks = Cin / KS;
    const int nitems = p.nitems;

    // PERSISTENT workgroups (one per CU): the first item is blockIdx.x, further ones are claimed from a global counter (a
    // workgroup that starts late -- its CU was busy with another stream's kernel or an RCCL channel -- simply claims
    // fewer; with a static item += gridDim.x split it would hold the whole launch back).  Items are cout-block major, so at any time
    // the whole chip works on one 64-channel slice of U (L2 resident).  The first two raw slices and the first weight
    // fragments of the NEXT item are requested before the output stage of the current one: its global latency and the
    // workgroup launch disappear behind work that exists anyway.
    auto decode = [&](int item) {
        WinoItem it;
        const int nb = item / p.blocks_per_nb;
        int b = item - nb * p.blocks_per_nb;
        int s = 0;
        while (s + 1 < p.nseg && b >= p.seg[s + 1].block0) ++s;
        b -= p.seg[s].block0;
        const int per_img = p.seg[s].tbh * p.seg[s].tbw;
        const int n = b / per_img;
        const int rem = b - n * per_img;
        const int tyb = rem / p.seg[s].tbw, txb = rem - tyb * p.seg[s].tbw;
        it.s = __builtin_amdgcn_readfirstlane(s);
        it.n = __builtin_amdgcn_readfirstlane(n);
        it.y0 = __builtin_amdgcn_readfirstlane(tyb * 2 * TBH);
        it.x0 = __builtin_amdgcn_readfirstlane(txb * 2 * TBW);
        it.cout0 = __builtin_amdgcn_readfirstlane(nb * BN);
        it.cb0 = __builtin_amdgcn_readfirstlane(min((nb * BN) >> 5, cob_all - 1));
        it.cb1 = __builtin_amdgcn_readfirstlane(min(((nb * BN) >> 5) + 1, cob_all - 1));
        return it;
    };

    // Wave specialisation: waves 0-3 only issue MFMAs (wave w owns transform row i = w), waves 4-7 only move and
    // transform data.  Every SIMD hosts one wave of each kind, so the hardware interleaves the matrix pipe with the
    // VALU / LDS / VMEM work of the transform without any help from the instruction scheduler; one barrier per slice.
    const int wave = tid >> 6, lane = tid & 63;
    const bool is_mma = wave < 4;
    const int li = lane & 31, h = lane >> 5;
    const int dt = tid & 255;
    constexpr int NRAW = (RAW_F4 + 255) / 256;
    const int t_chunk = dt & 3, t_tile = (dt >> 2) & 31, t_half = dt >> 7;
    const int t_ty = t_tile >> 3, t_tx = t_tile & 7;
    auto vswz = [](int row, int c) { return c ^ ((row >> 2) & 3); };

    auto raw_offsets = [&](const WinoItem& it, unsigned (&roff)[NRAW]) {
#pragma unroll
        for (int i = 0; i < NRAW; ++i) {
            const int idx = dt + 256 * i;
            roff[i] = OOB;
            if (idx < RAW_F4) {
                const int chunk = idx & 3, pix = idx >> 2;
                const int pr = pix / PC, pc = pix - pr * PC;
                const int iy = it.y0 - 1 + pr, ix = it.x0 - 1 + pc;
                if ((unsigned)iy < (unsigned)p.seg[it.s].H && (unsigned)ix < (unsigned)p.seg[it.s].W)
                    roff[i] = (unsigned)(it.n * p.seg[it.s].in_nstride + ((int64_t)iy * p.seg[it.s].W + ix) * Cin + chunk * 4) * 4u;
            }
        }
    };
    auto issue_raw = [&](const WinoItem& it, const unsigned (&roff)[NRAW], int ks_, float4* dst) {
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.seg[it.s].in), 0, (int)((long long)p.seg[it.s].N * p.seg[it.s].in_nstride * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < NRAW; ++i)
            dst[i] = (ks_ < nks && roff[i] != OOB) ? buf_load16(rs_in, roff[i] + (unsigned)(ks_ * KS * 4))
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_raw = [&](const float4* src, float4* rawbuf) {
#pragma unroll
        for (int i = 0; i < NRAW; ++i) {
            const int idx = dt + 256 * i;
            if (idx < RAW_F4) rawbuf[(idx >> 2) * RCS + (idx & 3)] = src[i];
        }
    };
    // rows (2*HALF, 2*HALF+1) of B^T d B for this thread's (tile, 4 channels); waves 4,5 do HALF 0, waves 6,7 HALF 1
    auto transform_half = [&](const float4* rawbuf, float4* V, auto half_tag) {
        constexpr int HALF = decltype(half_tag)::value;
        // HALF 0 needs patch rows 0,1,2 (r0 = d0 - d2, r1 = d1 + d2); HALF 1 rows 1,2,3 (r2 = d2 - d1, r3 = d1 - d3)
        float4 rr[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 d0 = rawbuf[((2 * t_ty + 0 + HALF) * PC + (2 * t_tx + c)) * RCS + t_chunk];
            const float4 d1 = rawbuf[((2 * t_ty + 1 + HALF) * PC + (2 * t_tx + c)) * RCS + t_chunk];
            const float4 d2 = rawbuf[((2 * t_ty + 2 + HALF) * PC + (2 * t_tx + c)) * RCS + t_chunk];
            if (HALF == 0) { rr[0][c] = f4sub(d0, d2); rr[1][c] = f4add(d1, d2); }
            else           { rr[0][c] = f4sub(d1, d0); rr[1][c] = f4sub(d0, d2); }
        }
        const int col = vswz(t_tile, t_chunk);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int i = 2 * HALF + a;
            V[((i * 4 + 0) * 32 + t_tile) * 4 + col] = f4sub(rr[a][0], rr[a][2]);
            V[((i * 4 + 1) * 32 + t_tile) * 4 + col] = f4add(rr[a][1], rr[a][2]);
            V[((i * 4 + 2) * 32 + t_tile) * 4 + col] = f4sub(rr[a][2], rr[a][1]);
            V[((i * 4 + 3) * 32 + t_tile) * 4 + col] = f4sub(rr[a][1], rr[a][3]);
        }
    };
    auto transform = [&](const float4* rawbuf, float4* V) {
        if (t_half == 0) transform_half(rawbuf, V, std::integral_constant<int, 0>{});
        else transform_half(rawbuf, V, std::integral_constant<int, 1>{});
    };
    // Weight fragments: buffer loads with ONE resource, a per-(position, cout sub-tile) 32-bit lane offset that is fixed for
    // an item, and the slice / k-step as the scalar offset operand -- 8 VGPRs of addressing in the MFMA waves (64-bit
    // pointer arithmetic per load spilled registers once the kernel became persistent).
    const __amdgpu_buffer_rsrc_t rs_U = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.U), 0, (int)(16 * per_xi * 4), 0x00020000);
    auto fb_offsets = [&](const WinoItem& it, unsigned (&fbo)[4][2]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned base = (unsigned)(((wave & 3) * 4 + j) * (per_xi / 4));
            fbo[j][0] = (base + (unsigned)(it.cb0 * (Cin / 4) + h) * 32u + (unsigned)li) * 16u;
            fbo[j][1] = (base + (unsigned)(it.cb1 * (Cin / 4) + h) * 32u + (unsigned)li) * 16u;
        }
    };
    auto load_fb = [&](const unsigned (&fbo)[4][2], int ks_, int kk, float4 (&f)[4][2], bool two) {
        const unsigned soff = (unsigned)(min(ks_, nks - 1) * (KS / 4) + 2 * kk) * 512u;      // 32 lanes x 16 B per chunk
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j][0] = buf_load16_s(rs_U, fbo[j][0], soff);
        if (two) {
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j][1] = buf_load16_s(rs_U, fbo[j][1], soff);
        }
    };

    // Output stage.  Wave w holds M[i = w][j = 0..3]; the column half of A^T M A, z_q = sum_j M[w][j] A[j][q], is taken in
    // registers, so only Z[i][q] goes through LDS: Zs[4 i][2 q][32 tiles][64 couts] = 64 KB, ONE pass.  All 512 threads
    // then finish y[p][q] = sum_i A^T[p][i] Z[i][q], the epilogue and the stores.
    const int o_c = tid & 63;
    auto emit = [&](const WinoItem& it, int rep) -> float {
        const int o_tile = (tid >> 6) + 8 * rep;
        const int co = it.cout0 + o_c;
        if (co >= p.Cout) return 0.f;
        const WinoSeg& sg = p.seg[it.s];
        const float sc = p.scale ? p.scale[co] : 1.f, sh = p.shift ? p.shift[co] : 0.f;
        float z[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) z[i][q] = Ms[((i * 2 + q) * 32 + o_tile) * 64 + o_c];
        float y[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            y[0][q] = z[0][q] + z[1][q] + z[2][q];
            y[1][q] = z[1][q] - z[2][q] - z[3][q];
        }
        const int ty = o_tile >> 3, tx = o_tile & 7;
        float csum = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int oy = it.y0 + 2 * ty + a, ox = it.x0 + 2 * tx + c;
                if (oy < sg.H && ox < sg.W) {
                    const int64_t o = it.n * sg.out_nstride + ((int64_t)oy * sg.W + ox) * p.Cout + co;
                    float v = y[a][c] * sc + sh;
                    if (sg.res) v += sg.res[o];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (sg.mask) v = sg.mask[o] > 0.f ? v : 0.f;
                    sg.out[o] = v;
                    csum += v;
                }
            }
        return csum;
    };
    auto emit_all = [&](const WinoItem& it) {
        float csum = 0.f;
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) csum += emit(it, rep);
        if (p.colsum && it.cout0 + o_c < p.Cout)
            atomicAdd(p.colsum + (p.colsum_copies > 1 ? (blockIdx.x & (p.colsum_copies - 1)) * p.Cout : 0) + it.cout0 + o_c, csum);
    };

    // The two roles run separate code paths with the SAME barrier sequence per item (2 + nks + 2); keeping them apart lets
    // the register allocator give the 128 accumulator registers to the MFMA waves only.
    int item = blockIdx.x;
    if (item >= nitems) return;
    WinoItem cur = decode(item);
    if (is_mma) {
        float4 fb[2][4][2];                                 // [k-step][position][cout sub-tile]
        unsigned fbo[4][2], nfbo[4][2];
        fb_offsets(cur, fbo);
        // an item whose second 32-channel sub-block lies beyond Cout (Cout = 40 / 68 / 80 heads: 64+4, 64+16) skips that
        // sub-block's weight fragments and MFMAs: half the matrix work of the item
        bool two = cur.cout0 + 32 < p.Cout;
        load_fb(fbo, 0, 0, fb[0], true);
        load_fb(fbo, 0, 1, fb[1], true);
        for (;;) {
            f32x16 acc[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][q][r] = 0.f;
            if (tid == 0) *sh_next = p.sched ? (int)gridDim.x + atomicAdd(p.sched, 1) : item + (int)gridDim.x;
            __syncthreads();
            __syncthreads();
            const int nxt_item = __builtin_amdgcn_readfirstlane(*sh_next);
            const bool has_next = nxt_item < nitems;
            const WinoItem nxt = has_next ? decode(nxt_item) : cur;
            fb_offsets(nxt, nfbo);
            for (int ks = 0; ks < nks; ++ks) {
                const float4* Vc = (ks & 1) ? Vs1 : Vs0;
                const bool last = ks + 1 == nks;
                // k-step 0, then its weight fragments are re-loaded for the next slice (of this item or, on the last
                // slice, slice 0 of the NEXT item); same for k-step 1.  (Re-loading each position right after its own
                // MFMAs, or interleaved one group late, measured 40-45 % slower.)
                float4 fa[2][4];                     // both k-steps' tile fragments up front: one LDS latency per slice
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) fa[kk][j] = Vc[((wave * 4 + j) * 32 + li) * 4 + vswz(li, 2 * kk + h)];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    if (!(p.dbg & 4)) {
#define ERD_WMFMA(m)                                                                                              \
                        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                 \
                            acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][j].m, fb[kk][j][0].m, acc[j][0], 0, 0, 0); \
                        if (two) {                                                                                    \
                            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
                                acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][j].m, fb[kk][j][1].m, acc[j][1], 0, 0, 0); \
                        }
                        ERD_WMFMA(x) ERD_WMFMA(y) ERD_WMFMA(z) ERD_WMFMA(w)
#undef ERD_WMFMA
                    }
                    if (!(p.dbg & 2)) {
                        if (last) load_fb(nfbo, 0, kk, fb[kk], true);      // (the next item may need both sub-blocks)
                        else load_fb(fbo, ks + 1, kk, fb[kk], two);
                    }
                }
                __syncthreads();
            }
            __syncthreads();                                // every wave is done with the operand buffers
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float m0 = acc[0][q2][r], m1 = acc[1][q2][r], m2 = acc[2][q2][r], m3 = acc[3][q2][r];
                    Ms[((wave * 2 + 0) * 32 + row) * 64 + q2 * 32 + li] = m0 + m1 + m2;
                    Ms[((wave * 2 + 1) * 32 + row) * 64 + q2 * 32 + li] = m1 - m2 - m3;
                }
            __syncthreads();
            emit_all(cur);
            if (!has_next) {
                if (tid == 0 && p.sched) {                   // the last workgroup to leave re-arms the counters
                    if (atomicAdd(p.sched + 1, 1) == (int)gridDim.x - 1) { p.sched[0] = 0; p.sched[1] = 0; }
                }
                break;
            }
            item = nxt_item;
            cur = nxt;
            two = cur.cout0 + 32 < p.Cout;
#pragma unroll
            for (int j = 0; j < 4; ++j) { fbo[j][0] = nfbo[j][0]; fbo[j][1] = nfbo[j][1]; }
            __syncthreads();                                // Zs consumed: the operand buffers may be refilled
        }
    } else {
        unsigned roff[NRAW], nroff[NRAW];
        float4 rv[NRAW], rvb[NRAW];
        raw_offsets(cur, roff);
        issue_raw(cur, roff, 0, rv);
        issue_raw(cur, roff, 1, rvb);
        for (;;) {
            store_raw(rv, raw0);
            __syncthreads();
            transform(raw0, Vs0);
            store_raw(rvb, raw1);
            issue_raw(cur, roff, 2, rv);
            __syncthreads();
            const int nxt_item = __builtin_amdgcn_readfirstlane(*sh_next);
            const bool has_next = nxt_item < nitems;
            const WinoItem nxt = has_next ? decode(nxt_item) : cur;
            // state: V[0] = V(0), raw1 = raw(1), rv = raw(2) in flight
            for (int ks = 0; ks < nks; ++ks) {
                const int c = ks & 1;
                float4* Vn = c ? Vs0 : Vs1;
                float4* rawc = c ? raw1 : raw0;      // held raw(ks): consumed before the last barrier -> receives raw(ks+2)
                float4* rawn = c ? raw0 : raw1;      // raw(ks+1)
                store_raw(rv, rawc);
                if (!(p.dbg & 8)) issue_raw(cur, roff, ks + 3, rv);
                if (!(p.dbg & 1)) transform(rawn, Vn);
                __syncthreads();
            }
            if (has_next) {                          // the next item's first two raw slices travel during the output stage
                raw_offsets(nxt, nroff);
                issue_raw(nxt, nroff, 0, rv);
                issue_raw(nxt, nroff, 1, rvb);
            }
            __syncthreads();
            __syncthreads();
            emit_all(cur);
            if (!has_next) break;
            item = nxt_item;
            cur = nxt;
#pragma unroll
            for (int i = 0; i < NRAW; ++i) roff[i] = nroff[i];
            __syncthreads();
        }
    }
}


// =============================================================================================
// Second-generation kernel (default; ERD_WINO_GEN=1 selects the first one above for A/B runs).
//
// Same workgroup item (32 tiles x 64 couts, 16-channel slices, 4 data waves + 4 MFMA waves) and the same data-wave
// pipeline, but the matrix work is laid out so that NOTHING but operands ever crosses a wave:
//   * MFMA wave w owns ALL 16 transform positions of couts [16w, 16w+16) x 32 tiles on v_mfma_f32_16x16x4_f32
//     (rows = couts from U, columns = tiles from V; 16 positions x 2 tile halves x 4 accumulator registers = 128).
//     A lane therefore holds M_xi[cout][tile] for every xi of its (4 couts, 1 tile) cell, and A^T M A, the epilogue and
//     the 16-byte stores of 4 consecutive couts happen in registers -- no Z staging through LDS, no barrier, and the
//     data waves (and the other MFMA waves) run on into the next item meanwhile.
//   * The slices of consecutive items form ONE stream: exactly one barrier per slice ("V(g+1) is complete and V(g) has
//     been read"), placed one position before the end of a slice so that the first tile fragments of the next slice
//     travel behind the last position's MFMAs; the weight fragments form a 16-deep register ring (one per position),
//     each re-loaded for the next slice right after its own MFMAs.
//   * U layout [xi][cout/16][cin/16][kq][cout%16][4]: lane (i = l & 15, kq = l >> 4) fetches 16 B = channels
//     16 ks + 4 kq + {0..3} of cout i; the MFMA k index of step s is kq, i.e. channel 16 ks + 4 kq + s -- the V rows
//     are read with the same lane map (tile j = l & 15, 16-byte chunk kq), swizzled so that every 16-lane service
//     group of ds_read_b128 covers all 64 banks.
// =============================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef ERD_WINO_ABL
#define ERD_WINO_ABL 0      // compile-time ablation switches of wino2_conv_kernel (tools/_abl builds): 1 no transform, 2 no weight loads, 4 no MFMA, 8 no raw loads, 16 no output stage
#endif

__device__ __forceinline__ int vswz2(int row, int c) { return c ^ ((0 - (row >> 2)) & 3); }

__global__ __launch_bounds__(256) void wino_weight2_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout,
                                                           int Cin, int flip) {
    const int cop = (Cout + 15) / 16 * 16;
    const int64_t idx = blockIdx.x * 256ll + threadIdx.x;
    if (idx >= (int64_t)cop * Cin) return;
    const int co = (int)(idx / Cin), ci = (int)(idx % Cin);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            g[a][b] = co < Cout ? w[((int64_t)co * 9 + (flip ? 8 - (a * 3 + b) : a * 3 + b)) * Cin + ci] : 0.f;
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int nks = Cin / 16;
    const int64_t per_xi = (int64_t)(cop / 16) * nks * 256;
    const int64_t base = (((int64_t)(co / 16) * nks + ci / 16) * 4 + (ci % 16) / 4) * 64 + (co % 16) * 4 + (ci % 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float u0 = t[i][0], u1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]), u2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]),
                    u3 = t[i][2];
        U[(i * 4 + 0) * per_xi + base] = u0;
        U[(i * 4 + 1) * per_xi + base] = u1;
        U[(i * 4 + 2) * per_xi + base] = u2;
        U[(i * 4 + 3) * per_xi + base] = u3;
    }
}

#if ERD_WINO_ABL & 32
__device__ unsigned long long g_wino_trace[256 * 8];       // per workgroup: [0..3] matrix wave 0, [4..7] data wave 4
#define ERD_T0(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define ERD_TACC(acc, v) acc += __builtin_amdgcn_s_memtime() - v
#else
#define ERD_T0(v)
#define ERD_TACC(acc, v)
#endif
__global__ __launch_bounds__(512, 2) void wino2_conv_kernel(const WinoDesc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* raw0 = reinterpret_cast<float4*>(smem);         // 2 x [192 pixel slots][RCS]
    float4* raw1 = raw0 + RAW2_LDS_F4;
    float4* Vs0 = raw1 + RAW2_LDS_F4;                        // 2 x [16][32 tiles][KS/4] swizzled
    float4* Vs1 = Vs0 + V_F4;
    float* sh_ss = reinterpret_cast<float*>(Vs1 + V_F4);     // [item k & 3][scale 64 | shift 64]
    int* sh_item = reinterpret_cast<int*>(sh_ss + 512);      // [2]: item k of this workgroup's sequence lives in slot k & 1

    const int tid = threadIdx.x;
    const int Cin = p.Cin;
    const int nks = Cin / KS;
    const int nitems = p.nitems;
    const int ncb16 = (p.Cout + 15) / 16;

    auto decode = [&](int item) {
        WinoItem it;
        const int nb = item / p.blocks_per_nb;
        int b = item - nb * p.blocks_per_nb;
        int s = 0;
        while (s + 1 < p.nseg && b >= p.seg[s + 1].block0) ++s;
        b -= p.seg[s].block0;
        const int per_img = p.seg[s].tbh * p.seg[s].tbw;
        const int n = b / per_img;
        const int rem = b - n * per_img;
        const int tyb = rem / p.seg[s].tbw, txb = rem - tyb * p.seg[s].tbw;
        it.s = __builtin_amdgcn_readfirstlane(s);
        it.n = __builtin_amdgcn_readfirstlane(n);
        it.y0 = __builtin_amdgcn_readfirstlane(tyb * 2 * TBH);
        it.x0 = __builtin_amdgcn_readfirstlane(txb * 2 * TBW);
        it.cout0 = __builtin_amdgcn_readfirstlane(nb * BN);
        it.cb0 = 0;
        it.cb1 = 0;
        return it;
    };
    // item k + 1 of this workgroup's sequence: claimed from the launch's counter, or (no counter) a static stride
    auto claim = [&](int k) -> int {
        return p.sched ? (int)gridDim.x + atomicAdd(p.sched, 1) : (int)blockIdx.x + (k + 1) * (int)gridDim.x;
    };

#ifndef ERD_WINO_SWAP
#define ERD_WINO_SWAP 1
#endif
    // waves 0-3 move data, waves 4-7 issue MFMAs (the hardware's issue arbitration favours the older waves of a SIMD:
    // with the matrix waves first, the data waves got about one instruction per MFMA)
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool is_mma = ERD_WINO_SWAP ? wave_id >= 4 : wave_id < 4;
    const int wave = wave_id & 3;
    int item0 = blockIdx.x;
    if (item0 >= nitems) return;

    if (is_mma) {
        // ------------------------------------------------------------------ matrix waves
        const int j = lane & 15, kq = lane >> 4;
        const __amdgpu_buffer_rsrc_t rs_U = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.U), 0, (int)((size_t)16 * ncb16 * nks * 1024), 0x00020000);
        const unsigned u_lane = (unsigned)lane * 16u;
        const unsigned per_xi_b = (unsigned)ncb16 * (unsigned)nks * 1024u;          // bytes per transform position
        // tile-fragment address inside a V buffer: row (xi, tile) = 64 B, chunk kq swizzled by the tile index
        const unsigned v_lane = (ERD_WINO_ABL & 128) ? (unsigned)lane * 16u : (unsigned)(j * 64 + vswz2(j, kq) * 16);   // (128: linear = conflict-free reference)
        const char* vbase0 = reinterpret_cast<const char*>(Vs0);

        WinoItem cur = decode(item0);
        int k_item = 0;
        unsigned long long t_bar = 0, t_out = 0, t_sw = 0; (void)t_bar; (void)t_out; (void)t_sw;
        ERD_T0(t_begin);
        unsigned u_item = (unsigned)__builtin_amdgcn_readfirstlane(((cur.cout0 >> 4) + wave) * nks * 1024);   // byte offset of (cb, ks = 0) inside a position
        f32x4 acc[16][2];
        float4 ub[16];                                                              // weight-fragment ring, one per position
        float4 va[2][2][2];                                                         // tile fragments [pair parity][position in pair][tile half]
#pragma unroll
        for (int q = 0; q < 16; ++q) ub[q] = buf_load16_s(rs_U, u_lane, (unsigned)q * per_xi_b + u_item);
        __syncthreads();                                    // P   (data waves: raw slice 0 is in LDS)
        __syncthreads();                                    // B_0 (V(0) complete)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            va[0][e][0] = *reinterpret_cast<const float4*>(vbase0 + v_lane + e * 2048);
            va[0][e][1] = *reinterpret_cast<const float4*>(vbase0 + v_lane + e * 2048 + 1024);
        }
        int g = 0;
        for (;;) {
            const int nxt_item = __builtin_amdgcn_readfirstlane(sh_item[(k_item + 1) & 1]);
            const bool has_next = nxt_item < nitems;
            const WinoItem nxt = has_next ? decode(nxt_item) : cur;
            const unsigned u_next = (unsigned)__builtin_amdgcn_readfirstlane(((nxt.cout0 >> 4) + wave) * nks * 1024);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc[q][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[q][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            for (int ks = 0; ks < nks; ++ks, ++g) {
                const char* vc = vbase0 + v_lane + ((g & 1) ? V_F4 * 16 : 0);      // one VGPR; positions are immediate offsets
                const char* vn = vbase0 + v_lane + ((g & 1) ? 0 : V_F4 * 16);
                int lastflag = __builtin_amdgcn_readfirstlane(ks + 1 == nks ? 1 : 0);
                asm volatile("" : "+s"(lastflag));          // (opaque: keeps the compiler from peeling the last slice)
                const unsigned u_reload = (unsigned)__builtin_amdgcn_readfirstlane(
                    (int)(lastflag ? u_next : u_item + (unsigned)(ks + 1) * 1024u));
                // positions in pairs: the fragments of pair P+1 are requested before the 16 MFMAs of pair P are issued
#pragma unroll
                for (int P = 0; P < 8; ++P) {
                    if (P == 7) {                           // B_{g+1}: V(g+1) complete; every read of V(g) has returned
                        ERD_T0(tb);
                        __syncthreads();
                        ERD_TACC(t_bar, tb);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // (the fragments of pair 7 are waited for by the MFMAs below anyway)
                    const char* src = P < 7 ? vc + (2 * P + 2) * 2048 : vn;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        va[(P + 1) & 1][e][0] = *reinterpret_cast<const float4*>(src + e * 2048);
                        va[(P + 1) & 1][e][1] = *reinterpret_cast<const float4*>(src + e * 2048 + 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);      // reads first: they travel behind this pair's 16 MFMAs
                    const int q0 = 2 * P, q1 = 2 * P + 1;
                    const float4 a00 = va[P & 1][0][0], a01 = va[P & 1][0][1], a10 = va[P & 1][1][0], a11 = va[P & 1][1][1];
                    const float4 u0 = ub[q0], u1 = ub[q1];
#define ERD_W2(m)                                                                                        \
                    acc[q0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.m, a00.m, acc[q0][0], 0, 0, 0); \
                    acc[q0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0.m, a01.m, acc[q0][1], 0, 0, 0); \
                    acc[q1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.m, a10.m, acc[q1][0], 0, 0, 0); \
                    acc[q1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1.m, a11.m, acc[q1][1], 0, 0, 0);
                    if (!(ERD_WINO_ABL & 4)) { ERD_W2(x) ERD_W2(y) ERD_W2(z) ERD_W2(w) }
#undef ERD_W2
                    if (!(ERD_WINO_ABL & 2)) {
                    ub[q0] = buf_load16_s(rs_U, u_lane, (unsigned)q0 * per_xi_b + u_reload);
                    ub[q1] = buf_load16_s(rs_U, u_lane, (unsigned)q1 * per_xi_b + u_reload);
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise hoists a whole slice of fragment reads
                                                            //  to the top and sinks the barrier and the reloads to the bottom)
                }
            }
            // ---- output stage, in registers: y = A^T M A per (tile, cout), epilogue, 16-byte stores
            ERD_T0(to);
            if (!(ERD_WINO_ABL & 16)) {
                const WinoSeg& sg = p.seg[cur.s];
                const int co0 = cur.cout0 + 16 * wave + 4 * kq;
                const float* ss = sh_ss + (k_item & 3) * 128;
                const int cl = 16 * wave + 4 * kq;                            // 0..63 inside the item's cout block
                const float4 sc = *reinterpret_cast<const float4*>(ss + cl);
                const float4 sh = *reinterpret_cast<const float4*>(ss + 64 + cl);
                const bool vec_ok = (p.Cout & 3) == 0;
                float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
                if (vec_ok && !sg.res && !sg.mask && !p.colsum) {
                    // the common case (forward convolutions: scale/shift/ReLU only), branch-free: 32-bit offsets into a
                    // buffer resource of the output map, pixels outside the map (and couts beyond Cout) get an offset
                    // past the buffer -- the hardware drops those stores
                    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
                        sg.out, 0, (int)((long long)sg.N * sg.out_nstride * 4), 0x00020000);
                    const float lo = p.relu ? 0.f : -__builtin_inff();
                    const unsigned px_b = (unsigned)p.Cout * 4u, row_b = (unsigned)sg.W * px_b;
                    const int ty0 = j >> 3, tx = j & 7;                          // tile t = th * 16 + j -> (ty0 + 2 th, tx)
                    const int ox0 = cur.x0 + 2 * tx;
                    const unsigned off_t = (unsigned)(cur.n * sg.out_nstride + ((int64_t)(cur.y0 + 2 * ty0) * sg.W + ox0) * p.Cout + co0) * 4u;
                    const bool cok = co0 < p.Cout;
                    const bool xok0 = cok && ox0 < sg.W, xok1 = cok && ox0 + 1 < sg.W;
#pragma unroll
                    for (int th = 0; th < 2; ++th) {
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            float4 z[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                float4 m[3];
#pragma unroll
                                for (int jj = 0; jj < 3; ++jj) {
                                    const f32x4 a = acc[i * 4 + jj + c][th];
                                    m[jj] = make_float4(a[0], a[1], a[2], a[3]);
                                }
                                z[i] = c == 0 ? f4add(f4add(m[0], m[1]), m[2]) : f4sub(f4sub(m[0], m[1]), m[2]);
                            }
#pragma unroll
                            for (int a = 0; a < 2; ++a) {
                                const float4 yv = a == 0 ? f4add(f4add(z[0], z[1]), z[2]) : f4sub(f4sub(z[1], z[2]), z[3]);
                                const int oy = cur.y0 + 2 * ty0 + 4 * th + a;
                                const bool ok = (c == 0 ? xok0 : xok1) && oy < sg.H;
                                const unsigned off = ok ? off_t + (unsigned)(4 * th + a) * row_b + (unsigned)c * px_b : 0x80000000u;
                                u32x4 v;
                                v.x = __float_as_uint(fmaxf(yv.x * sc.x + sh.x, lo));
                                v.y = __float_as_uint(fmaxf(yv.y * sc.y + sh.y, lo));
                                v.z = __float_as_uint(fmaxf(yv.z * sc.z + sh.z, lo));
                                v.w = __float_as_uint(fmaxf(yv.w * sc.w + sh.w, lo));
                                __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, off, 0, 0);
                            }
                        }
                    }
                } else
#pragma unroll
                for (int th = 0; th < 2; ++th) {
                    const int t = th * 16 + j, ty = t >> 3, tx = t & 7;
                    // One output column c at a time: z[i] = (M A)[i][c], then the two rows y[a][c] -- 24 live registers.
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        float4 z[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float4 m[3];
#pragma unroll
                            for (int jj = 0; jj < 3; ++jj) {
                                const f32x4 a = acc[i * 4 + jj + c][th];
                                m[jj] = make_float4(a[0], a[1], a[2], a[3]);
                            }
                            z[i] = c == 0 ? f4add(f4add(m[0], m[1]), m[2]) : f4sub(f4sub(m[0], m[1]), m[2]);
                        }
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            const float4 yv = a == 0 ? f4add(f4add(z[0], z[1]), z[2]) : f4sub(f4sub(z[1], z[2]), z[3]);
                            const int oy = cur.y0 + 2 * ty + a, ox = cur.x0 + 2 * tx + c;
                            if (oy < sg.H && ox < sg.W && co0 < p.Cout) {
                                const int64_t o = cur.n * sg.out_nstride + ((int64_t)oy * sg.W + ox) * p.Cout + co0;
                                float4 v = make_float4(yv.x * sc.x + sh.x, yv.y * sc.y + sh.y, yv.z * sc.z + sh.z,
                                                       yv.w * sc.w + sh.w);
                                if (vec_ok) {
                                    if (sg.res) v = f4add(v, *reinterpret_cast<const float4*>(sg.res + o));
                                    if (p.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                                    if (sg.mask) {
                                        const float4 mk = *reinterpret_cast<const float4*>(sg.mask + o);
                                        v = make_float4(mk.x > 0.f ? v.x : 0.f, mk.y > 0.f ? v.y : 0.f, mk.z > 0.f ? v.z : 0.f,
                                                        mk.w > 0.f ? v.w : 0.f);
                                    }
                                    *reinterpret_cast<float4*>(sg.out + o) = v;
                                } else {
                                    float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        if (co0 + r < p.Cout) {
                                            float e = vv[r];
                                            if (sg.res) e += sg.res[o + r];
                                            if (p.relu) e = fmaxf(e, 0.f);
                                            if (sg.mask) e = sg.mask[o + r] > 0.f ? e : 0.f;
                                            sg.out[o + r] = e;
                                            vv[r] = e;
                                        } else vv[r] = 0.f;
                                    }
                                    v = make_float4(vv[0], vv[1], vv[2], vv[3]);
                                }
                                cs = f4add(cs, v);
                            }
                        }
                    }
                }
                if (p.colsum) {                                               // (Cout % 4 == 0 is required with colsum)
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) {
                        cs.x += __shfl_xor(cs.x, o, 64); cs.y += __shfl_xor(cs.y, o, 64);
                        cs.z += __shfl_xor(cs.z, o, 64); cs.w += __shfl_xor(cs.w, o, 64);
                    }
                    if (j == 0 && co0 < p.Cout) {
                        float* cp = p.colsum + (p.colsum_copies > 1 ? (blockIdx.x & (p.colsum_copies - 1)) * p.Cout : 0) + co0;
                        atomicAdd(cp + 0, cs.x); atomicAdd(cp + 1, cs.y); atomicAdd(cp + 2, cs.z); atomicAdd(cp + 3, cs.w);
                    }
                }
            }
            ERD_TACC(t_out, to);
            if (!has_next) {
#if ERD_WINO_ABL & 32
                if (wave == 0 && lane == 0 && blockIdx.x < 256) {
                    g_wino_trace[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime() - t_begin;
                    g_wino_trace[blockIdx.x * 8 + 1] = t_bar;
                    g_wino_trace[blockIdx.x * 8 + 2] = t_out;
                    g_wino_trace[blockIdx.x * 8 + 3] = (unsigned long long)(k_item + 1);
                }
#endif
                if (wave == 0 && lane == 0 && p.sched) {     // the last workgroup to leave re-arms the counters
                    if (atomicAdd(p.sched + 1, 1) == (int)gridDim.x - 1) { p.sched[0] = 0; p.sched[1] = 0; }
                }
                break;
            }
            cur = nxt;
            u_item = u_next;
            ++k_item;
        }
    } else {
        // ------------------------------------------------------------------ data waves
        // (Every instruction here competes with the matrix waves for the SIMD's issue slots -- measured: the loop took ~3x
        //  longer beside the MFMA stream than alone -- so the steady state is kept to the bare minimum: 3 unpredicated
        //  buffer loads (out-of-image and padding entries carry an offset beyond the buffer: they return zeros), LDS
        //  addresses that are one VGPR base + immediates (the buffer parity and the transform half are template
        //  parameters), the buffer resource of the look-ahead item in SGPRs.)
#ifndef ERD_WINO_PRIO
#define ERD_WINO_PRIO 3
#endif
        __builtin_amdgcn_s_setprio(ERD_WINO_PRIO);
        const int dt = tid & 255;
        const int t_chunk = dt & 3, t_tile = (dt >> 2) & 31, t_half = __builtin_amdgcn_readfirstlane(dt >> 7);
        const int t_ty = t_tile >> 3, t_tx = t_tile & 7;
        WinoItem la = decode(item0);                          // the item of the look-ahead pointer (3 slices ahead of g)
        unsigned long long t_bar = 0, t_ent = 0, t_wait = 0; (void)t_bar; (void)t_ent; (void)t_wait;
        ERD_T0(t_begin);
        int la_ks = 0, k_la = 0;
        unsigned la_soff = 0;
        bool la_valid = true;
        int slices_total = nks;
        constexpr unsigned OOBV = 0x80000000u;
        unsigned roff[3];
        float4 rv[3], rvb[3];
        __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.seg[0].in), 0, 0, 0x00020000);
        float pend_sc = 1.f, pend_sh = 0.f;
        int pend_claim = 0, pend_k = -1;
        // entering item k of the sequence: its buffer resource and patch offsets, its scale / shift slice, the claim of item k + 1
        auto enter_item = [&]() {
            const WinoSeg& sg = p.seg[la.s];
            rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.in), 0, (int)((long long)sg.N * sg.in_nstride * 4),
                                                      0x00020000);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int idx = dt + 256 * i;
                const int chunk = idx & 3, pix = idx >> 2;
                const int pr = pix / PC, pc = pix - pr * PC;
                const int iy = la.y0 - 1 + pr, ix = la.x0 - 1 + pc;
                roff[i] = OOBV;
                if (idx < RAW_F4 && (unsigned)iy < (unsigned)sg.H && (unsigned)ix < (unsigned)sg.W)
                    roff[i] = (unsigned)(la.n * sg.in_nstride + ((int64_t)iy * sg.W + ix) * Cin + chunk * 4) * 4u;
            }
            // requested here, written to LDS one slice later (flush_pending, after the wait the raw slice needs anyway):
            // neither the atomic's round trip nor the two loads ever stall the data waves
            if (dt < 64) {
                const int co = la.cout0 + dt;
                pend_sc = (p.scale && co < p.Cout) ? p.scale[co] : 1.f;
                pend_sh = (p.shift && co < p.Cout) ? p.shift[co] : 0.f;
            }
            if (dt == 64) pend_claim = claim(k_la);
            pend_k = k_la;
        };
        auto flush_pending = [&]() {
            if (pend_k >= 0) {
                if (dt < 64) {
                    float* ss = sh_ss + (pend_k & 3) * 128;
                    ss[dt] = pend_sc;
                    ss[64 + dt] = pend_sh;
                }
                if (dt == 64) sh_item[(pend_k + 1) & 1] = pend_claim;
                pend_k = -1;
            }
        };
        auto issue_next = [&](float4* dst) {
            if (la_valid) {
                if (!(ERD_WINO_ABL & 8)) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) dst[i] = buf_load16_s(rs_in, roff[i], la_soff);
                }
                la_soff += KS * 4;
                if (++la_ks == nks) {                         // the pointer leaves item k_la
                    const int nx = __builtin_amdgcn_readfirstlane(sh_item[(k_la + 1) & 1]);
                    if (nx < nitems) {
                        la = decode(nx);
                        la_ks = 0;
                        la_soff = 0;
                        ++k_la;
                        slices_total += nks;
                        enter_item();
                    } else la_valid = false;
                }
            }
        };
        char* const sm = smem;
        constexpr unsigned RAWB = RAW2_LDS_F4 * 16, VOFF = 2 * RAWB, VB = V_F4 * 16;
        const unsigned st_base = (unsigned)(((dt >> 2) * RCS + (dt & 3)) * 16);                       // + i * 64 pixels
        const unsigned rd_base = (unsigned)((((2 * t_ty + t_half) * PC + 2 * t_tx) * RCS + t_chunk) * 16);
        const unsigned wr_base = VOFF + (unsigned)((((2 * t_half * 4) * 32 + t_tile) * 4 + vswz2(t_tile, t_chunk)) * 16);
        auto store_raw = [&](const float4* src, auto par_tag) {
            constexpr unsigned PAR = decltype(par_tag)::value;
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<float4*>(sm + PAR * RAWB + st_base + i * (64 * RCS * 16)) = src[i];
        };
        // rows (2 HALF, 2 HALF + 1) of B^T d B for this thread's (tile, 4 channels): raw buffer RPAR -> V buffer VPAR
        auto transform = [&](auto rpar_tag, auto vpar_tag, auto half_tag) {
            constexpr unsigned RPAR = decltype(rpar_tag)::value, VPAR = decltype(vpar_tag)::value;
            constexpr int HALF = decltype(half_tag)::value;
            // HALF 0 needs patch rows 0,1,2 (r0 = d0 - d2, r1 = d1 + d2); HALF 1 rows 1,2,3 (r2 = d2 - d1, r3 = d1 - d3)
            float4 rr[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 d0 = *reinterpret_cast<const float4*>(sm + RPAR * RAWB + rd_base + (0 * PC + c) * (RCS * 16));
                const float4 d1 = *reinterpret_cast<const float4*>(sm + RPAR * RAWB + rd_base + (1 * PC + c) * (RCS * 16));
                const float4 d2 = *reinterpret_cast<const float4*>(sm + RPAR * RAWB + rd_base + (2 * PC + c) * (RCS * 16));
                if (HALF == 0) { rr[0][c] = f4sub(d0, d2); rr[1][c] = f4add(d1, d2); }
                else           { rr[0][c] = f4sub(d1, d0); rr[1][c] = f4sub(d0, d2); }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                char* v = sm + wr_base + VPAR * VB + a * (4 * 2048);
                *reinterpret_cast<float4*>(v + 0 * 2048) = f4sub(rr[a][0], rr[a][2]);
                *reinterpret_cast<float4*>(v + 1 * 2048) = f4add(rr[a][1], rr[a][2]);
                *reinterpret_cast<float4*>(v + 2 * 2048) = f4sub(rr[a][2], rr[a][1]);
                *reinterpret_cast<float4*>(v + 3 * 2048) = f4sub(rr[a][1], rr[a][3]);
            }
        };
        using P0 = std::integral_constant<unsigned, 0>;
        using P1 = std::integral_constant<unsigned, 1>;
        auto run = [&](auto half_tag) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { rv[i] = make_float4(0.f, 0.f, 0.f, 0.f); rvb[i] = rv[i]; }
            enter_item();
            flush_pending();
            issue_next(rv);                                   // raw(0)
            issue_next(rvb);                                  // raw(1)
            store_raw(rv, P0{});
            __syncthreads();                                  // P
            transform(P0{}, P0{}, half_tag);
            store_raw(rvb, P1{});
            flush_pending();
            issue_next(rv);                                   // raw(2)
            __syncthreads();                                  // B_0
            // state at the top of iteration g: V(g) complete, raw(g+1) in raw[(g+1)&1], rv = raw(g+2) in flight;
            // iteration g stores raw(g+2) over raw(g), requests raw(g+3), transforms raw(g+1) -> V(g+1); barrier B_{g+1}
            auto iter = [&](auto par_tag, auto npar_tag) {
                ERD_T0(ts);
                store_raw(rv, par_tag);
                flush_pending();
                issue_next(rv);
                ERD_TACC(t_ent, ts);                          // (trace builds: slot 6 = store + issue incl. enter_item)
                ERD_T0(tt);
                if (!(ERD_WINO_ABL & 1)) transform(npar_tag, npar_tag, half_tag);
#if ERD_WINO_ABL & 32
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                ERD_TACC(t_wait, tt);                         // (slot 7 = transform)
                ERD_T0(tb);
                __syncthreads();                              // B_{g+1}
                ERD_TACC(t_bar, tb);
            };
            for (int g = 0;;) {
                if (g >= slices_total) break;
                iter(P0{}, P1{});
                if (++g >= slices_total) break;
                iter(P1{}, P0{});
                ++g;
            }
        };
        if (t_half == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, 1>{});
#if ERD_WINO_ABL & 32
        if (wave == 0 && lane == 0 && blockIdx.x < 256) {
            g_wino_trace[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memtime() - t_begin;
            g_wino_trace[blockIdx.x * 8 + 5] = t_bar;
            g_wino_trace[blockIdx.x * 8 + 6] = t_ent;
            g_wino_trace[blockIdx.x * 8 + 7] = t_wait;
        }
#endif
    }
}

}  // namespace

static int wino_gen() {
    static const int gen = getenv("ERD_WINO_GEN") ? atoi(getenv("ERD_WINO_GEN")) : 2;
    return gen;
}

extern "C" int erd_wino_weights(const float* w_ohwi, float* U, int Cout, int Cin, int flip, erd_stream_t stream) {
    ERD_REQUIRE(w_ohwi && U && Cout > 0 && Cin > 0 && Cin % 4 == 0, "wino_weights: bad args");
    if (wino_gen() == 2) {
        ERD_REQUIRE(Cin % KS == 0, "wino_weights: Cin=%d must be a multiple of %d", Cin, KS);
        const int64_t n = (int64_t)((Cout + 15) / 16 * 16) * Cin;
        hipLaunchKernelGGL(wino_weight2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_ohwi,
                           U, Cout, Cin, flip);
        return erd::check_launch("wino_weights");
    }
    const int64_t n = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_ohwi, U,
                       Cout, Cin, flip);
    return erd::check_launch("wino_weights");
}

#if ERD_WINO_ABL & 32
extern "C" int erd_wino_trace(unsigned long long* out) {       // debug builds only (tools/build_abl.sh 32)
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino_trace), sizeof(g_wino_trace));
}
#endif

extern "C" size_t erd_wino_weights_elems(int Cout, int Cin) {
    if (wino_gen() == 2) return (size_t)16 * ((Cout + 15) / 16 * 16) * Cin;
    return (size_t)16 * ((Cout + 31) / 32) * (Cin / 4) * 128;
}

extern "C" int erd_wino_conv3x3(const erd_conv_seg* segs, int nseg, const float* U, int Cin, int Cout,
                                const float* scale, const float* shift, int relu, float* colsum, int colsum_copies,
                                int* sched, erd_stream_t stream) {
    ERD_REQUIRE(segs && U && nseg >= 1 && nseg <= ERD_MAX_SEG, "wino: bad args");
    ERD_REQUIRE(Cin % KS == 0 && Cout > 0, "wino: Cin=%d must be a multiple of %d", Cin, KS);
    WinoDesc d;
    d.nseg = nseg;
    d.U = U;
    d.Cin = Cin;
    d.Cout = Cout;
    d.scale = scale;
    d.shift = shift;
    d.relu = relu;
    d.colsum = colsum;
    ERD_REQUIRE(colsum_copies >= 0 && (colsum_copies & (colsum_copies - 1)) == 0, "wino: colsum_copies must be a power of two");
    d.colsum_copies = colsum_copies;
    d.sched = sched;
    ERD_REQUIRE(!colsum || Cout % 4 == 0, "wino: colsum needs Cout %% 4 == 0");
    static const int dbg = getenv("ERD_WINO_DBG") ? atoi(getenv("ERD_WINO_DBG")) : 0;
    d.dbg = dbg;
    int blocks = 0;
    for (int s = 0; s < nseg; ++s) {
        const erd_conv_seg& g = segs[s];
        ERD_REQUIRE(g.in && g.out && g.IH == g.OH && g.IW == g.OW, "wino: segment %d is not a stride-1 same-size map", s);
        ERD_REQUIRE((int64_t)g.N * g.in_nstride < (1ll << 29), "wino: segment %d too large", s);
        WinoSeg& w = d.seg[s];
        w.in = g.in;
        w.out = g.out;
        w.res = g.res;
        w.mask = g.mask;
        ERD_REQUIRE(!g.alpha, "wino: per-level scalars are not supported");
        w.N = g.N;
        w.H = g.IH;
        w.W = g.IW;
        w.tbh = ((g.IH + 1) / 2 + TBH - 1) / TBH;
        w.tbw = ((g.IW + 1) / 2 + TBW - 1) / TBW;
        w.in_nstride = g.in_nstride;
        w.out_nstride = g.out_nstride;
        w.block0 = blocks;
        blocks += g.N * w.tbh * w.tbw;
    }
    d.blocks_per_nb = blocks;
    const int ncb = (Cout + BN - 1) / BN;
    const bool gen2 = wino_gen() == 2;
    const size_t lds = gen2 ? (size_t)2 * (RAW2_LDS_F4 + V_F4) * sizeof(float4) + 2048 + 16
                            : (size_t)2 * (RAW_LDS_F4 + V_F4) * sizeof(float4) + 16;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino2_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_done = true;
    }
    if (blocks == 0) return 0;
    d.nitems = blocks * ncb;
    int ncu = 0;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        static int cached = 0;
        if (cached == 0 && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cached = prop.multiProcessorCount;
        ncu = cached > 0 ? cached : 256;
    }
    static const int persist = getenv("ERD_WINO_PERSIST") ? atoi(getenv("ERD_WINO_PERSIST")) : 1;
    const int grid = persist ? (d.nitems < ncu ? d.nitems : ncu) : d.nitems;
    if (persist != 1) d.sched = nullptr;      // 2: persistent grid, static item stride (gen 2 only)
    if (gen2) hipLaunchKernelGGL(wino2_conv_kernel, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, d);
    else hipLaunchKernelGGL(wino_conv_kernel, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, d);
    return erd::check_launch("wino_conv3x3");
}
