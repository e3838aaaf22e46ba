// Winograd F(2x2, 3x3) convolution for the stride-1 3x3 layers, fp32 on the gfx950 matrix cores.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray 2016; 2.25x fewer multiplications than direct)
//
// The 16 transform positions xi = (i, j) are 16 independent GEMMs  M_xi[cout][tile] = sum_cin U_xi[cout][cin] V_xi[tile][cin].
// One workgroup item = 32 output tiles (a bh x bw block of 2x2-pixel tiles of one image of one level, bh * bw = 32) x 64
// output channels, walked in 16-channel K slices by 8 waves with fixed roles:
//   * 4 DATA waves: raw (2bh+2) x (2bw+2) pixel patch global -> registers -> LDS (out-of-image and padding entries carry
//     an offset beyond the buffer: they return zeros), then B^T d B per (tile, 4 channels) LDS -> LDS into
//     V[xi][tile][16 cin] rows (64 B, XOR swizzle), double-buffered, raw slices requested three slices ahead;
//   * 4 MATRIX waves: wave w owns ALL 16 positions of couts [16w, 16w+16) x 32 tiles on v_mfma_f32_16x16x4_f32 (rows =
//     couts from U, columns = tiles from V; 16 positions x 2 tile halves x 4 accumulator registers = 128).  A lane
//     therefore holds M_xi[cout][tile] for every xi of its (4 couts, 1 tile) cell, and A^T M A, the epilogue and the
//     16-byte stores of 4 consecutive couts happen in registers -- no staging through LDS, no barrier, and the data waves
//     run on into the next item meanwhile.  Weight fragments come straight from L2: U is stored pre-tiled
//     [xi][cout/16][cin/16][kq][cout%16][4] so that a fragment load is one contiguous 1 KB (lane (i = l & 15, kq = l >> 4)
//     fetches channels 16 ks + 4 kq + {0..3} of cout i; the MFMA k index of step s is kq, i.e. channel 16 ks + 4 kq + s);
//     they form a 16-deep register ring (one per position), each re-loaded for the next slice right after its own MFMAs.
//     The V rows are read with the same lane map (tile j = l & 15, 16-byte chunk kq), swizzled so that every 16-lane
//     service group of ds_read_b128 covers all 64 banks; positions go in pairs, the fragments of pair P+1 are requested
//     before the 16 MFMAs of pair P are issued.
//   * The slices of consecutive items form ONE stream with exactly one barrier per slice ("V(g+1) is complete and V(g)
//     has been read"), placed one pair before the end of a slice so that the first fragments of the next slice travel
//     behind the last pair's MFMAs.  Items are claimed from a global counter (persistent grid, one workgroup per CU).
//   * Block shapes: the tile grid of a map is covered by 4x8 blocks in its interior, and its ragged bottom / right strips
//     by 2x16 / 1x32 resp. 8x4 / 16x2 / 32x1 blocks (always 32 tiles): 179 instead of 202 blocks on the five head levels
//     at 800x1344 (the plain 4x8 cover wastes 15 % of the matrix work in partly empty blocks).
// The input gradient of the same layers is this kernel run on dz with the flipped, transposed weights.  (The weight
// gradient stays a direct GEMM over pixels.)
//
// Measured on the head-tower launch (4 x 22400 pixels, 256 -> 256; s_memtime traces, -DERD_WINO_TRACE): the matrix
// waves spend 78 % of their cycles issuing MFMAs at the pipe's rate; what is left is item switches (output stage 3 %),
// barrier waits at item boundaries (6 %) and issue slots lost to the data waves' instructions (the fp32 MFMA runs at
// the vector rate, every VALU / LDS instruction of the co-resident data wave competes with it -- which is why the data
// waves' steady state is kept to ~75 instructions per slice).
#include "erd_common.h"
#include <type_traits>
#include <stdlib.h>
#include <algorithm>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOBV = 0x80000000u;        // beyond num_records of every buffer built here: loads return zeros, stores are dropped

__device__ __forceinline__ float4 buf_load16_s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

constexpr int KS = 16;                        // input channels per K slice
constexpr int BN = 64;                        // output channels per workgroup item
constexpr int RCS = 6;                        // LDS chunks per raw pixel: 4 data + 2 pad -> the strided 4x4-patch reads of the
                                              // transform are bank-conflict free (4 would be 8-way)
#ifndef ERD_WINO_NCH
#define ERD_WINO_NCH 4
#endif
constexpr int NCH = ERD_WINO_NCH;             // staged chunks of 256 float4 (= 64 pixels x 16 channels) per slice
constexpr int MAXPIX = 64 * NCH;              // raw pixel slots (a 2x16 block needs 6 x 34 = 204; 1x32 blocks are not used)
constexpr int RAW_LDS_F4 = MAXPIX * RCS;
constexpr int V_F4 = 16 * 32 * (KS / 4);      // float4s of the 16 transformed tiles
constexpr int MAXREG = 16;                    // block regions per launch (<= 3 per map)

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ int vswz2(int row, int c) { return c ^ ((0 - (row >> 2)) & 3); }

#ifdef ERD_WINO_TRACE       // debug builds only (tools/build_abl.sh): per-workgroup cycle accounting of both wave roles
__device__ unsigned long long g_wino_trace[256 * 8];       // [0..3] matrix wave 0, [4..7] data wave 0
__device__ unsigned long long g_wino_trace_p[256 * 16];    // wino_x3p_kernel: [0..7] wave 0, [8..15] wave 4
#define ERD_T0(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define ERD_TACC(acc, v) acc += __builtin_amdgcn_s_memtime() - v
#else
#define ERD_T0(v)
#define ERD_TACC(acc, v)
#endif

// U[xi][cout/16][cin/16][kq][cout%16][4] = (G g G^T)[xi] ; g = w[cout][kh][kw][cin]  (couts padded to 16 with zeros)
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout,
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


struct WinoSeg {
    const float* in;
    float* out;
    const float* res;        // optional, geometry of out: added before ReLU / mask (may alias out)
    const float* mask;       // optional, geometry of out: result zeroed where mask <= 0
    int N, H, W;
    int64_t in_nstride, out_nstride;
};
// a rectangle of the tile grid of one map, covered by nby x nbx blocks of (32 >> lbw) x (1 << lbw) tiles
struct WinoRegion {
    int seg, ty0, tx0, nby, nbx, lbw;
    int block0;              // first block (per cout block) of this region
    int ty1, tx1;            // tiles at or beyond (ty1, tx1) belong to another region: computed (whole blocks) but not stored,
                             // so every output pixel is written -- and column-summed, and read as a residual -- exactly once
};
struct WinoDesc {
    int nseg, nreg;
    WinoSeg seg[ERD_MAX_SEG];
    WinoRegion reg[MAXREG];
    const float* U;
    const void* U3;          // three-limb form (wino_x3_kernel): the bf16 limb image of U, erd_wino_weights_x3
    int Cin, Cout;
    const float* scale;
    const float* shift;
    int relu;
    float* colsum;           // optional [colsum_copies][Cout]: += column sums of the stored result
    int colsum_copies;       // 0/1 or a power of two: workgroup b adds into row b mod copies
    int blocks_per_nb;       // blocks per cout block
    int nitems;              // workgroup items = blocks_per_nb * cout blocks
    int* sched;              // optional {next-item counter, finished-workgroup counter}, zero on entry and on exit
    float* gn_part;          // optional (wino_x3p_kernel): [nitems][16][2] = (sum, sum of squares) of what the item STORED, per group of 8
                             // channels -- the GroupNorm(32) statistics of gfl_head.py:158-177 from the producer (wino_gn_finalize_kernel)
};

// One (tile block, cout block) work item: where it lives.  All fields are wave-uniform (scalar registers).
struct WinoItem {
    int s, n, y0, x0, cout0, lbw;       // y0, x0: first output pixel of the block; block = (32 >> lbw) x (1 << lbw) tiles
    int yl, xl;                         // output pixels at or beyond (yl, xl) are not this item's to store
};

__global__ __launch_bounds__(512, 2) void wino_conv_kernel(const WinoDesc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: raw0 | raw1 (2 x [MAXPIX pixel slots][RCS] float4) | V0 | V1 (2 x [16][32 tiles][KS/4] float4, swizzled) |
    //      sh_ss [item k & 3][scale 64 | shift 64] | sh_item [2] (item k of this workgroup's sequence lives in slot k & 1)
    constexpr unsigned RAWB = RAW_LDS_F4 * 16, VOFF = 2 * RAWB, VB = V_F4 * 16;
    float* sh_ss = reinterpret_cast<float*>(smem + VOFF + 2 * VB);
    int* sh_item = reinterpret_cast<int*>(sh_ss + 512);

    const int tid = threadIdx.x;
    const int Cin = p.Cin;
    const int nks = Cin / KS;
    const int nitems = p.nitems;
    const int ncb16 = (p.Cout + 15) / 16;

    auto decode = [&](int item) {
        WinoItem it;
        const int nb = item / p.blocks_per_nb;
        int b = item - nb * p.blocks_per_nb;
        int r = 0;
        while (r + 1 < p.nreg && b >= p.reg[r + 1].block0) ++r;
        const WinoRegion& rg = p.reg[r];
        b -= rg.block0;
        const int per_img = rg.nby * rg.nbx;
        const int n = b / per_img;
        const int rem = b - n * per_img;
        const int by = rem / rg.nbx, bx = rem - by * rg.nbx;
        const int lbw = rg.lbw;
        it.s = __builtin_amdgcn_readfirstlane(rg.seg);
        it.n = __builtin_amdgcn_readfirstlane(n);
        it.y0 = __builtin_amdgcn_readfirstlane(2 * (rg.ty0 + by * (32 >> lbw)));
        it.x0 = __builtin_amdgcn_readfirstlane(2 * (rg.tx0 + (bx << lbw)));
        it.cout0 = __builtin_amdgcn_readfirstlane(nb * BN);
        it.lbw = __builtin_amdgcn_readfirstlane(lbw);
        it.yl = __builtin_amdgcn_readfirstlane(min(p.seg[rg.seg].H, 2 * rg.ty1));
        it.xl = __builtin_amdgcn_readfirstlane(min(p.seg[rg.seg].W, 2 * rg.tx1));
        return it;
    };
    // item k + 1 of this workgroup's sequence: claimed from the launch's counter, or (no counter) a static stride
    auto claim = [&](int k) -> int {
        return p.sched ? (int)gridDim.x + atomicAdd(p.sched, 1) : (int)blockIdx.x + (k + 1) * (int)gridDim.x;
    };

    // waves 0-3 move data, waves 4-7 issue MFMAs; every SIMD hosts one wave of each kind
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool is_mma = wave_id >= 4;
    const int wave = wave_id & 3;
    const int item0 = blockIdx.x;
    if (item0 >= nitems) return;

    if (is_mma) {
        // ------------------------------------------------------------------ matrix waves
#ifdef ERD_WINO_MMA_PRIO
        __builtin_amdgcn_s_setprio(ERD_WINO_MMA_PRIO);
#endif
        const int j = lane & 15, kq = lane >> 4;
        const __amdgpu_buffer_rsrc_t rs_U = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.U), 0, (int)((size_t)16 * ncb16 * nks * 1024), 0x00020000);
        const unsigned u_lane = (unsigned)lane * 16u;
        const unsigned per_xi_b = (unsigned)ncb16 * (unsigned)nks * 1024u;          // bytes per transform position
        // tile-fragment address inside a V buffer: row (xi, tile) = 64 B, chunk kq swizzled by the tile index
        const unsigned v_lane = (unsigned)(j * 64 + vswz2(j, kq) * 16);
        const char* vbase0 = smem + VOFF;

        WinoItem cur = decode(item0);
        int k_item = 0;
        unsigned long long t_bar = 0, t_out = 0; (void)t_bar; (void)t_out;
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
                const char* vc = vbase0 + v_lane + ((g & 1) ? VB : 0);             // one VGPR; positions are immediate offsets
                const char* vn = vbase0 + v_lane + ((g & 1) ? 0 : VB);
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
                    ERD_W2(x) ERD_W2(y) ERD_W2(z) ERD_W2(w)
#undef ERD_W2
                    ub[q0] = buf_load16_s(rs_U, u_lane, (unsigned)q0 * per_xi_b + u_reload);
                    ub[q1] = buf_load16_s(rs_U, u_lane, (unsigned)q1 * per_xi_b + u_reload);
                    __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise hoists a whole slice of fragment reads
                                                            //  to the top and sinks the barrier and the reloads to the bottom)
                }
            }
            // ---- output stage, in registers: y = A^T M A per (tile, cout), epilogue, 16-byte stores
            ERD_T0(to);
            {
                const WinoSeg& sg = p.seg[cur.s];
                const int co0 = cur.cout0 + 16 * wave + 4 * kq;
                const float* ss = sh_ss + (k_item & 3) * 128;
                const int cl = 16 * wave + 4 * kq;                            // 0..63 inside the item's cout block
                const float4 sc = *reinterpret_cast<const float4*>(ss + cl);
                const float4 sh = *reinterpret_cast<const float4*>(ss + 64 + cl);
                const bool vec_ok = (p.Cout & 3) == 0;
                const int lbw = cur.lbw, bwm = (1 << lbw) - 1;                // tile t of the block -> (t >> lbw, t & bwm)
                float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
                if (vec_ok && !sg.res && !sg.mask && !p.colsum) {
                    // the common case (forward convolutions: scale/shift/ReLU only), branch-free: 32-bit offsets into a
                    // buffer resource of the output map, pixels outside the map (and couts beyond Cout) get an offset
                    // past the buffer -- the hardware drops those stores
                    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
                        sg.out, 0, (int)((long long)sg.N * sg.out_nstride * 4), 0x00020000);
                    const float lo = p.relu ? 0.f : -__builtin_inff();
                    const unsigned px_b = (unsigned)p.Cout * 4u, row_b = (unsigned)sg.W * px_b;
                    const unsigned off_n = (unsigned)(cur.n * sg.out_nstride + ((int64_t)cur.y0 * sg.W + cur.x0) * p.Cout + co0) * 4u;
                    const bool cok = co0 < p.Cout;
#pragma unroll
                    for (int th = 0; th < 2; ++th) {
                        const int t = th * 16 + j, ty = t >> lbw, tx = t & bwm;
                        const unsigned off_t = off_n + (unsigned)(2 * ty) * row_b + (unsigned)(2 * tx) * px_b;
                        const int oy0 = cur.y0 + 2 * ty, ox0 = cur.x0 + 2 * tx;
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
                                const bool ok = cok && oy0 + a < cur.yl && ox0 + c < cur.xl;
                                const unsigned off = ok ? off_t + (unsigned)a * row_b + (unsigned)c * px_b : OOBV;
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
                    const int t = th * 16 + j, ty = t >> lbw, tx = t & bwm;
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
                            if (oy < cur.yl && ox < cur.xl && co0 < p.Cout) {
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
#ifdef ERD_WINO_TRACE
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
        //  longer beside the MFMA stream than alone -- so the steady state is kept to the bare minimum: unpredicated
        //  buffer loads (out-of-image and padding entries carry an offset beyond the buffer: they return zeros), LDS
        //  addresses that are VGPR bases + immediates (the buffer parity and the transform half are template parameters),
        //  the buffer resource of the look-ahead item in SGPRs.)
#ifndef ERD_WINO_DATA_PRIO
#define ERD_WINO_DATA_PRIO 3
#endif
        __builtin_amdgcn_s_setprio(ERD_WINO_DATA_PRIO);
        const int dt = tid & 255;
        const int t_chunk = dt & 3, t_tile = (dt >> 2) & 31, t_half = __builtin_amdgcn_readfirstlane(dt >> 7);
        WinoItem la = decode(item0);                          // the item of the look-ahead pointer (3 slices ahead of g)
        unsigned long long t_bar = 0, t_ent = 0, t_wait = 0; (void)t_bar; (void)t_ent; (void)t_wait;
        ERD_T0(t_begin);
        int la_ks = 0, k_la = 0;
        unsigned la_soff = 0;
        bool la_valid = true;
        int slices_total = nks;
        unsigned roff[NCH];                                    // raw staging: float4 idx = dt + 256 i  (pixel idx >> 2, chunk idx & 3)
        float4 rv[NCH], rvb[NCH];
        __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.seg[0].in), 0, 0, 0x00020000);
        float pend_sc = 1.f, pend_sh = 0.f;
        int pend_claim = 0, pend_k = -1;
        // geometry of the transform (it runs two slices behind the look-ahead pointer: at most two items are alive)
        unsigned rd0 = 0, rd1 = 0, rd2 = 0;                    // this thread's three patch rows (LDS bytes inside a raw buffer)
        unsigned nrd0 = 0, nrd1 = 0, nrd2 = 0;                 // ... of the item the look-ahead pointer has entered
        int tr_left = 0;                                       // slices the transform still has to do in its current item
        // Entering an item costs ~200 instructions (decode, four patch offsets, the transform's patch rows): done in one
        // iteration it made that iteration longer than a slice and the matrix waves waited at the barrier (10 k cycles
        // per item).  It is therefore STAGED over the iterations before the pointer gets there: the id of item k + 1 is
        // readable two iterations after the pointer entered item k (claim -> flush -> barrier), so
        //   la_ks == 3: decode(item k + 1)        la_ks == 4: its offsets / patch rows / buffer resource
        //   la_ks == nks: switch (register moves), request its scale / shift slice and claim item k + 2.
        WinoItem nx_it = la;
        bool nx_valid = false;
        unsigned nx_roff[NCH], nx_rd0 = 0, nx_rd1 = 0, nx_rd2 = 0;
        __amdgpu_buffer_rsrc_t nx_rs = rs_in;
        auto item_geometry = [&](const WinoItem& it, unsigned (&ro)[NCH], unsigned& g0, unsigned& g1, unsigned& g2,
                                 __amdgpu_buffer_rsrc_t& rs) {
            const WinoSeg& sg = p.seg[it.s];
            rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.in), 0, (int)((long long)sg.N * sg.in_nstride * 4), 0x00020000);
            const int lbw = it.lbw, bw = 1 << lbw, bh = 32 >> lbw;
            const int pc_n = 2 * bw + 2, npix = (2 * bh + 2) * pc_n;
            const int recip = (65536 + pc_n - 1) / pc_n;              // (scalar)
            const unsigned base_n = (unsigned)(it.n * sg.in_nstride);
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int idx = dt + 256 * i;
                const int chunk = idx & 3, pix = idx >> 2;
                const int pr = (pix * recip) >> 16, pc = pix - pr * pc_n;         // = pix / pc_n for pix < 512 (checked for every pitch)
                const int iy = it.y0 - 1 + pr, ix = it.x0 - 1 + pc;
                ro[i] = OOBV;
                if (pix < npix && (unsigned)iy < (unsigned)sg.H && (unsigned)ix < (unsigned)sg.W)
                    ro[i] = (base_n + (unsigned)((iy * sg.W + ix) * Cin + chunk * 4)) * 4u;       // (< 2^31: checked by the host)
            }
            const int t_ty = t_tile >> lbw, t_tx = t_tile & (bw - 1);
            g0 = (unsigned)((((2 * t_ty + t_half) * pc_n + 2 * t_tx) * RCS + t_chunk) * 16);
            g1 = g0 + (unsigned)(pc_n * RCS * 16);
            g2 = g1 + (unsigned)(pc_n * RCS * 16);
        };
        // its scale / shift slice and the claim of the following item are REQUESTED on entry and written to LDS one slice
        // later (flush_pending, after the wait the raw slice needs anyway): neither the atomic's round trip nor the two
        // loads ever stall the data waves
        auto request_item_data = [&]() {
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
#pragma unroll
                for (int i = 0; i < NCH; ++i) dst[i] = buf_load16_s(rs_in, roff[i], la_soff);
                la_soff += KS * 4;
                ++la_ks;
                if (la_ks == 3) {                             // stage 1: which item comes next
                    const int nx = __builtin_amdgcn_readfirstlane(sh_item[(k_la + 1) & 1]);
                    nx_valid = nx < nitems;
                    if (nx_valid) nx_it = decode(nx);
                }
                if (la_ks == 4 && nx_valid) item_geometry(nx_it, nx_roff, nx_rd0, nx_rd1, nx_rd2, nx_rs);    // stage 2
                if (la_ks == nks) {                           // the pointer leaves item k_la
                    if (nx_valid) {
                        la = nx_it;
                        rs_in = nx_rs;
#pragma unroll
                        for (int i = 0; i < NCH; ++i) roff[i] = nx_roff[i];
                        nrd0 = nx_rd0; nrd1 = nx_rd1; nrd2 = nx_rd2;
                        la_ks = 0;
                        la_soff = 0;
                        ++k_la;
                        slices_total += nks;
                        request_item_data();
                    } else la_valid = false;
                }
            }
        };
        char* const sm = smem;
        const unsigned st_base = (unsigned)(((dt >> 2) * RCS + (dt & 3)) * 16);                       // + i * 64 pixels
        const unsigned wr_base = VOFF + (unsigned)((((2 * t_half * 4) * 32 + t_tile) * 4 + vswz2(t_tile, t_chunk)) * 16);
        // (all chunks are always loaded and stored: entries beyond a smaller patch carry OOBV offsets -> zeros nobody reads)
        auto store_raw = [&](const float4* src, auto par_tag) {
            constexpr unsigned PAR = decltype(par_tag)::value;
#pragma unroll
            for (int i = 0; i < NCH; ++i) *reinterpret_cast<float4*>(sm + PAR * RAWB + st_base + i * (64 * RCS * 16)) = src[i];
        };
        // rows (2 HALF, 2 HALF + 1) of B^T d B for this thread's (tile, 4 channels): raw buffer RPAR -> V buffer VPAR
        auto transform = [&](auto rpar_tag, auto vpar_tag, auto half_tag) {
            constexpr unsigned RPAR = decltype(rpar_tag)::value, VPAR = decltype(vpar_tag)::value;
            constexpr int HALF = decltype(half_tag)::value;
            // HALF 0 needs patch rows 0,1,2 (r0 = d0 - d2, r1 = d1 + d2); HALF 1 rows 1,2,3 (r2 = d2 - d1, r3 = d1 - d3)
            const char* r0 = sm + RPAR * RAWB + rd0;
            const char* r1 = sm + RPAR * RAWB + rd1;
            const char* r2 = sm + RPAR * RAWB + rd2;
            float4 rr[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 d0 = *reinterpret_cast<const float4*>(r0 + c * (RCS * 16));
                const float4 d1 = *reinterpret_cast<const float4*>(r1 + c * (RCS * 16));
                const float4 d2 = *reinterpret_cast<const float4*>(r2 + c * (RCS * 16));
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
            if (--tr_left == 0) {                             // the next slice belongs to the item the pointer entered last
                rd0 = nrd0;
                rd1 = nrd1;
                rd2 = nrd2;
                tr_left = nks;
            }
        };
        using P0 = std::integral_constant<unsigned, 0>;
        using P1 = std::integral_constant<unsigned, 1>;
        auto run = [&](auto half_tag) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) { rv[i] = make_float4(0.f, 0.f, 0.f, 0.f); rvb[i] = rv[i]; }
            item_geometry(la, roff, nrd0, nrd1, nrd2, rs_in);
            request_item_data();
            rd0 = nrd0;
            rd1 = nrd1;
            rd2 = nrd2;
            tr_left = nks;
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
#ifdef ERD_WINO_GNPROBE     // timing probe only (tools/build_probe.sh): what a GroupNorm-apply + ReLU on the raw slice would cost the
                            // data waves -- one scale and NCH shift float4 loads per slice (the shift row of a padding entry would be a
                            // zero row) and fma + max per element.  The values are placeholders: results are NOT a convolution of anything.
                {
                    const float4 gsc = buf_load16_s(rs_in, (unsigned)(dt & 3) * 16u, la_soff & 1023u);
#pragma unroll
                    for (int i = 0; i < NCH; ++i) {
                        const float4 gsh = buf_load16_s(rs_in, (unsigned)(dt & 3) * 16u + 64u * i, la_soff & 1023u);
                        rv[i].x = fmaxf(rv[i].x * gsc.x + gsh.x, 0.f); rv[i].y = fmaxf(rv[i].y * gsc.y + gsh.y, 0.f);
                        rv[i].z = fmaxf(rv[i].z * gsc.z + gsh.z, 0.f); rv[i].w = fmaxf(rv[i].w * gsc.w + gsh.w, 0.f);
                    }
                }
#endif
                store_raw(rv, par_tag);
                flush_pending();
                issue_next(rv);
                ERD_TACC(t_ent, ts);                          // (trace builds: slot 6 = store + issue incl. enter_item)
                ERD_T0(tt);
                transform(npar_tag, npar_tag, half_tag);
#ifdef ERD_WINO_TRACE
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
#ifdef ERD_WINO_TRACE
        if (wave == 0 && lane == 0 && blockIdx.x < 256) {
            g_wino_trace[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memtime() - t_begin;
            g_wino_trace[blockIdx.x * 8 + 5] = t_bar;
            g_wino_trace[blockIdx.x * 8 + 6] = t_ent;
            g_wino_trace[blockIdx.x * 8 + 7] = t_wait;
        }
#endif
    }
}


// =====================================================================================================================
// The THREE-LIMB form of the same algorithm ("f32x3"): fp32 maps, fp32 accumulation, fp32 results, the 16 transform-domain
// GEMMs on the bf16 matrix cores.  U = G g G^T arrives pre-split into three bf16 limb planes (erd_wino_weights_x3), V = B^T d B
// is split by the data waves right after the transform (erd::limbs3_pair: exact sums, round-to-nearest limbs), and every
// product u v is formed as the six limb products of weight >= 2^-16 on v_mfma_f32_32x32x16_bf16 -- 6 x 32 cycles per
// (position, 32 couts x 32 tiles x 16 channels) against 128 x 32 cycles of v_mfma_f32_16x16x4_f32 for all 16 positions of
// 16 couts: a slice is 1 536 matrix cycles per wave instead of 4 096 (tools/wino_x3_skeleton.hip measured the matrix side
// alone -- MFMAs + weight-fragment stream -- at 1 892 cycles per slice, profiles/r04_wino_x3_skeleton.txt).
// What changes against wino_conv_kernel:
//   * item = 32 tiles x 64 couts as before, but a matrix wave owns 32 couts x 32 tiles (the MFMA's shape) of HALF the
//     positions: wave w -> cout block w >> 1, positions 8 (w & 1) .. + 7 (= rows i = 2 (w & 1), 2 (w & 1) + 1 of the 4 x 4
//     transform grid): 8 x 16 = 128 accumulator registers.  A^T M A needs all four rows: each wave forms z[i][c] =
//     (M A)[i][c] for its two rows, the pair swaps ONE row each through LDS (wave A sends z1 and finishes output row 0 =
//     (z0 + z1) + z2, wave B sends z2 and finishes row 1 = (z1 - z2) - z3: the fp32 kernel's summation order) in eight
//     rounds of 4 registers, double-buffered, handed over with LDS flags (no workgroup barrier: the data waves run on);
//   * V lives in LDS as [position][limb][k half][tile][8 channels] bf16 (a wave's B-fragment read is 1 KB, every 16-lane
//     service group 256 contiguous bytes: conflict-free), 48 KB per slice, double-buffered; the data waves write 8 bytes per
//     (position, limb);
//   * weight fragments: U3[position][limb][cout block][slice] is 1 KB contiguous; a ring of FOUR positions (x 3 limbs),
//     re-loaded four positions ahead.
// Everything else -- item list, block shapes, raw staging, look-ahead, the single barrier per slice -- is the fp32 kernel's.
constexpr int VX_B = 16 * 3 * 1024;            // bytes of one V buffer (three-limb form)
#ifndef ERD_WX3_RD
#define ERD_WX3_RD 4                          // weight-fragment ring depth in positions (4: half a slice ahead; 8: a whole slice)
#endif
constexpr int XCH_B = 1024;                    // bytes of one exchange chunk (4 registers x 64 lanes)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 2) void wino_x3_kernel(const WinoDesc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: raw0 | raw1 | V0 | V1 (2 x VX_B) | exchange [4 waves][2][XCH_B] | flags [4] | sh_ss [4][128] | sh_item [2]
    constexpr unsigned RAWB = RAW_LDS_F4 * 16, VOFF = 2 * RAWB, VB = VX_B, XOFF = VOFF + 2 * VB;
    volatile int* xflag = reinterpret_cast<volatile int*>(smem + XOFF + 4 * 2 * XCH_B);
    float* sh_ss = reinterpret_cast<float*>(smem + XOFF + 4 * 2 * XCH_B + 64);
    int* sh_item = reinterpret_cast<int*>(sh_ss + 512);

    const int tid = threadIdx.x;
    const int Cin = p.Cin;
    const int nks = Cin / KS;
    const int nitems = p.nitems;
    const int ncb32 = (p.Cout + 31) / 32;

    auto decode = [&](int item) {
        WinoItem it;
        const int nb = item / p.blocks_per_nb;
        int b = item - nb * p.blocks_per_nb;
        int r = 0;
        while (r + 1 < p.nreg && b >= p.reg[r + 1].block0) ++r;
        const WinoRegion& rg = p.reg[r];
        b -= rg.block0;
        const int per_img = rg.nby * rg.nbx;
        const int n = b / per_img;
        const int rem = b - n * per_img;
        const int by = rem / rg.nbx, bx = rem - by * rg.nbx;
        const int lbw = rg.lbw;
        it.s = __builtin_amdgcn_readfirstlane(rg.seg);
        it.n = __builtin_amdgcn_readfirstlane(n);
        it.y0 = __builtin_amdgcn_readfirstlane(2 * (rg.ty0 + by * (32 >> lbw)));
        it.x0 = __builtin_amdgcn_readfirstlane(2 * (rg.tx0 + (bx << lbw)));
        it.cout0 = __builtin_amdgcn_readfirstlane(nb * BN);
        it.lbw = __builtin_amdgcn_readfirstlane(lbw);
        it.yl = __builtin_amdgcn_readfirstlane(min(p.seg[rg.seg].H, 2 * rg.ty1));
        it.xl = __builtin_amdgcn_readfirstlane(min(p.seg[rg.seg].W, 2 * rg.tx1));
        return it;
    };
    auto claim = [&](int k) -> int {
        return p.sched ? (int)gridDim.x + atomicAdd(p.sched, 1) : (int)blockIdx.x + (k + 1) * (int)gridDim.x;
    };

    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool is_mma = wave_id >= 4;
    const int wave = wave_id & 3;
    const int item0 = blockIdx.x;
    if (item0 >= nitems) return;
    if (tid < 4) xflag[tid] = 0;                 // (published by the first workgroup barrier, long before the first exchange)

    if (is_mma) {
        // ------------------------------------------------------------------ matrix waves
        const int li = lane & 31, h = lane >> 5;
        const int cb = wave >> 1, ph = wave & 1;
        const __amdgpu_buffer_rsrc_t rs_U = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(p.U3), 0, (int)((size_t)16 * 3 * ncb32 * nks * 1024), 0x00020000);
        const unsigned u_lane = (unsigned)lane * 16u;
        const unsigned per_xl_b = (unsigned)ncb32 * (unsigned)nks * 1024u;          // bytes per (position, limb)
        const unsigned pos0_b = (unsigned)(ph * 8 * 3) * per_xl_b;                  // this wave's first position
        // B-fragment address inside a V buffer: (position, limb) block of 1 KB, tile row 32 B, k half h
        // (layout of a (position, limb) block: [k half h][tile][8 channels = 16 B], the tiles of half 1 XOR 8 -- see the data waves)
        const unsigned v_lane = (unsigned)(ph * 8 * 3 * 1024 + h * 512 + ((li ^ (h * 8)) * 16));
        const char* vbase0 = smem + VOFF;
        char* const xmy = smem + XOFF + wave * 2 * XCH_B + lane * 16;
        const char* const xpartner = smem + XOFF + (wave ^ 1) * 2 * XCH_B + lane * 16;
        int xr = 0;                                                                 // exchange rounds done

        WinoItem cur = decode(item0);
        int k_item = 0;
        unsigned long long t_bar = 0, t_out = 0, t_xch = 0; (void)t_bar; (void)t_out; (void)t_xch;
        ERD_T0(t_begin);
        unsigned u_item = (unsigned)__builtin_amdgcn_readfirstlane(((cur.cout0 >> 5) + cb) * nks * 1024);   // byte offset of (cout block, ks = 0) inside a plane
        f32x16 acc[8];
        u32x4 ub[ERD_WX3_RD][3];                                                             // weight-fragment ring: position q lives in slot q mod ERD_WX3_RD
        bf16x8 vf[2][3];                                                            // tile fragments [position parity][limb]
        auto load_u = [&](const int q, const unsigned soff) {                      // q: compile-time after unrolling
#pragma unroll
            for (int l = 0; l < 3; ++l)
                ub[q & (ERD_WX3_RD - 1)][l] = __builtin_amdgcn_raw_buffer_load_b128(rs_U, u_lane, pos0_b + (unsigned)(q * 3 + l) * per_xl_b + soff, 0);
        };
#pragma unroll
        for (int q = 0; q < ERD_WX3_RD; ++q) load_u(q, u_item);
        __syncthreads();                                    // P   (data waves: raw slice 0 is in LDS)
        __syncthreads();                                    // B_0 (V(0) complete)
#pragma unroll
        for (int l = 0; l < 3; ++l) vf[0][l] = *reinterpret_cast<const bf16x8*>(vbase0 + v_lane + l * 1024);
        int g = 0;
        for (;;) {
            const int nxt_item = __builtin_amdgcn_readfirstlane(sh_item[(k_item + 1) & 1]);
            const bool has_next = nxt_item < nitems;
            const WinoItem nxt = has_next ? decode(nxt_item) : cur;
            const unsigned u_next = (unsigned)__builtin_amdgcn_readfirstlane(((nxt.cout0 >> 5) + cb) * nks * 1024);
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
            for (int ks = 0; ks < nks; ++ks, ++g) {
                const char* vc = vbase0 + v_lane + ((g & 1) ? VB : 0);
                const char* vn = vbase0 + v_lane + ((g & 1) ? 0 : VB);
                int lastflag = __builtin_amdgcn_readfirstlane(ks + 1 == nks ? 1 : 0);
                asm volatile("" : "+s"(lastflag));          // (opaque: keeps the compiler from peeling the last slice)
                const unsigned u_cur = (unsigned)__builtin_amdgcn_readfirstlane((int)(u_item + (unsigned)ks * 1024u));
                const unsigned u_reload = (unsigned)__builtin_amdgcn_readfirstlane(
                    (int)(lastflag ? u_next : u_item + (unsigned)(ks + 1) * 1024u));
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (q == 7) {                           // B_{g+1}: V(g+1) complete; every read of V(g) has been issued
                        ERD_T0(tb);
                        __syncthreads();
                        ERD_TACC(t_bar, tb);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const char* src = q < 7 ? vc + (q + 1) * 3 * 1024 : vn;
#ifdef ERD_WX3_VREAD1      // timing probe: one of the three limb fragments is read (results are wrong)
                    vf[(q + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(src);
                    vf[(q + 1) & 1][1] = vf[(q + 1) & 1][0]; vf[(q + 1) & 1][2] = vf[(q + 1) & 1][0];
#else
#pragma unroll
                    for (int l = 0; l < 3; ++l) vf[(q + 1) & 1][l] = *reinterpret_cast<const bf16x8*>(src + l * 1024);
#endif
                    __builtin_amdgcn_sched_barrier(0);      // reads first: they travel behind this position's MFMAs
                    const bf16x8 uh = __builtin_bit_cast(bf16x8, ub[q & (ERD_WX3_RD - 1)][0]), um = __builtin_bit_cast(bf16x8, ub[q & (ERD_WX3_RD - 1)][1]),
                                 ul = __builtin_bit_cast(bf16x8, ub[q & (ERD_WX3_RD - 1)][2]);
                    const bf16x8 vh = vf[q & 1][0], vm = vf[q & 1][1], vl = vf[q & 1][2];
                    // rows = couts (U), columns = tiles (V); smallest terms first, as everywhere in the three-limb kernels
#ifdef ERD_WX3_NOMFMA       // timing probe: everything but the matrix instructions (results are wrong)
                    acc[q][0] += __builtin_bit_cast(float4, ul).x * __builtin_bit_cast(float4, vh).x + __builtin_bit_cast(float4, um).x * __builtin_bit_cast(float4, vm).x +
                                 __builtin_bit_cast(float4, uh).x * __builtin_bit_cast(float4, vl).x;
#else
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ul, vh, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(um, vh, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vl, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(um, vm, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vm, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vh, acc[q], 0, 0, 0);
#endif
                    // the slot is free: position q + 4 of this slice, or position q - 4 of the next one
#ifndef ERD_WX3_NOLOAD      // (timing probe: the ring keeps the first slice's fragments)
                    if (q + ERD_WX3_RD < 8) load_u(q + ERD_WX3_RD, u_cur); else load_u(q + ERD_WX3_RD - 8, u_reload);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- output stage: z[i][c] = (M A)[i][c] for this wave's two rows, one row swapped with the partner wave, y = A^T z
            ERD_T0(to);
            {
                const WinoSeg& sg = p.seg[cur.s];
                const float* ss = sh_ss + (k_item & 3) * 128;
                const int lbw = cur.lbw, bwm = (1 << lbw) - 1;
                const int ty = li >> lbw, tx = li & bwm;
                const int oy = cur.y0 + 2 * ty + ph;                          // this wave finishes output row a = ph of every tile
                const bool simple = !sg.res && !sg.mask && !p.colsum;
                const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
                    sg.out, 0, (int)((long long)sg.N * sg.out_nstride * 4), 0x00020000);
                const float lo = p.relu ? 0.f : -__builtin_inff();
                float4 cs[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) cs[gq] = make_float4(0.f, 0.f, 0.f, 0.f);
                // z[il][c] = (M A)[i][c] of this wave's rows i = 2 ph + il: the accumulators die here, row by row
                f32x16 z[2][2];
                z[0][0] = (acc[0] + acc[1]) + acc[2];
                z[0][1] = (acc[1] - acc[2]) - acc[3];
                z[1][0] = (acc[4] + acc[5]) + acc[6];
                z[1][1] = (acc[5] - acc[6]) - acc[7];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f32x16 za = z[0][c], zb = z[1][c];
                    f32x16 recv;
#pragma unroll
                    for (int qr = 0; qr < 4; ++qr) {                          // A sends z1, B sends z2: four registers per round
                        char* mine = xmy + (xr & 1) * XCH_B;
                        const float4 s4 = ph == 0 ? make_float4(zb[4 * qr], zb[4 * qr + 1], zb[4 * qr + 2], zb[4 * qr + 3])
                                                  : make_float4(za[4 * qr], za[4 * qr + 1], za[4 * qr + 2], za[4 * qr + 3]);
                        *reinterpret_cast<float4*>(mine) = s4;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (lane == 0) xflag[wave] = xr + 1;
                        ERD_T0(tx);
                        while (xflag[wave ^ 1] < xr + 1) __builtin_amdgcn_s_sleep(1);
                        ERD_TACC(t_xch, tx);
                        const float4 r0 = *reinterpret_cast<const float4*>(xpartner + (xr & 1) * XCH_B);
                        recv[4 * qr + 0] = r0.x; recv[4 * qr + 1] = r0.y; recv[4 * qr + 2] = r0.z; recv[4 * qr + 3] = r0.w;
                        ++xr;
                    }
                    // wave A: y[0][c] = (z0 + z1) + z2 ; wave B: y[1][c] = (z1 - z2) - z3   (the fp32 kernel's order)
                    const f32x16 yv = ph == 0 ? (za + zb) + recv : (recv - za) - zb;
                    const int ox = cur.x0 + 2 * tx + c;
                    const bool pix_ok = oy < cur.yl && ox < cur.xl;
                    const int64_t opix = cur.n * sg.out_nstride + ((int64_t)oy * sg.W + ox) * p.Cout;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {                          // registers 4 gq .. 4 gq + 3: couts 8 gq + 4 h + {0..3} of the block
                        const int cl = 32 * cb + 8 * gq + 4 * h;              // 0..63 inside the item's cout block
                        const int co0 = cur.cout0 + cl;
                        const float4 sc = *reinterpret_cast<const float4*>(ss + cl);
                        const float4 sh = *reinterpret_cast<const float4*>(ss + 64 + cl);
                        float4 v = make_float4(yv[4 * gq] * sc.x + sh.x, yv[4 * gq + 1] * sc.y + sh.y, yv[4 * gq + 2] * sc.z + sh.z,
                                               yv[4 * gq + 3] * sc.w + sh.w);
                        if (simple && (p.Cout & 3) == 0) {
                            const bool ok = pix_ok && co0 < p.Cout;
                            u32x4 o;
                            o.x = __float_as_uint(fmaxf(v.x, lo)); o.y = __float_as_uint(fmaxf(v.y, lo));
                            o.z = __float_as_uint(fmaxf(v.z, lo)); o.w = __float_as_uint(fmaxf(v.w, lo));
                            __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, ok ? (unsigned)((opix + co0) * 4) : OOBV, 0, 0);
                        } else if (pix_ok && co0 < p.Cout) {
                            const int64_t o = opix + co0;
                            if ((p.Cout & 3) == 0) {
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
                            cs[gq] = f4add(cs[gq], v);
                        }
                    }
                }
                if (p.colsum) {                                               // (Cout % 4 == 0 is required with colsum)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
                        for (int o = 16; o > 0; o >>= 1) {
                            cs[gq].x += __shfl_xor(cs[gq].x, o, 64); cs[gq].y += __shfl_xor(cs[gq].y, o, 64);
                            cs[gq].z += __shfl_xor(cs[gq].z, o, 64); cs[gq].w += __shfl_xor(cs[gq].w, o, 64);
                        }
                        const int co0 = cur.cout0 + 32 * cb + 8 * gq + 4 * h;
                        if (li == 0 && co0 < p.Cout) {
                            float* cp = p.colsum + (p.colsum_copies > 1 ? (blockIdx.x & (p.colsum_copies - 1)) * p.Cout : 0) + co0;
                            atomicAdd(cp + 0, cs[gq].x); atomicAdd(cp + 1, cs[gq].y); atomicAdd(cp + 2, cs[gq].z); atomicAdd(cp + 3, cs[gq].w);
                        }
                    }
                }
            }
            ERD_TACC(t_out, to);
            if (!has_next) {
#ifdef ERD_WINO_TRACE
                if (wave == 0 && lane == 0 && blockIdx.x < 256) {
                    g_wino_trace[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime() - t_begin;
                    g_wino_trace[blockIdx.x * 8 + 1] = t_bar;
                    g_wino_trace[blockIdx.x * 8 + 2] = t_out;
                    g_wino_trace[blockIdx.x * 8 + 3] = t_xch;             // (x3: the exchange polls inside the output stage, not the item count)
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
        // ------------------------------------------------------------------ data waves (wino_conv_kernel's, with the limb split
        // behind the transform and 8-byte stores into the [position][limb][tile][16 channels] planes)
        __builtin_amdgcn_s_setprio(ERD_WINO_DATA_PRIO);
        const int dt = tid & 255;
        const int t_chunk = dt & 3, t_tile = (dt >> 2) & 31, t_half = __builtin_amdgcn_readfirstlane(dt >> 7);
        WinoItem la = decode(item0);
        unsigned long long t_bar = 0, t_ent = 0, t_wait = 0; (void)t_bar; (void)t_ent; (void)t_wait;
        ERD_T0(t_begin);
        int la_ks = 0, k_la = 0;
        unsigned la_soff = 0;
        bool la_valid = true;
        int slices_total = nks;
        unsigned roff[NCH];
        float4 rv[NCH], rvb[NCH];
        __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.seg[0].in), 0, 0, 0x00020000);
        float pend_sc = 1.f, pend_sh = 0.f;
        int pend_claim = 0, pend_k = -1;
        unsigned rd0 = 0, rd1 = 0, rd2 = 0;
        unsigned nrd0 = 0, nrd1 = 0, nrd2 = 0;
        int tr_left = 0;
        WinoItem nx_it = la;
        bool nx_valid = false;
        unsigned nx_roff[NCH], nx_rd0 = 0, nx_rd1 = 0, nx_rd2 = 0;
        __amdgpu_buffer_rsrc_t nx_rs = rs_in;
        auto item_geometry = [&](const WinoItem& it, unsigned (&ro)[NCH], unsigned& g0, unsigned& g1, unsigned& g2,
                                 __amdgpu_buffer_rsrc_t& rs) {
            const WinoSeg& sg = p.seg[it.s];
            rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.in), 0, (int)((long long)sg.N * sg.in_nstride * 4), 0x00020000);
            const int lbw = it.lbw, bw = 1 << lbw, bh = 32 >> lbw;
            const int pc_n = 2 * bw + 2, npix = (2 * bh + 2) * pc_n;
            const int recip = (65536 + pc_n - 1) / pc_n;
            const unsigned base_n = (unsigned)(it.n * sg.in_nstride);
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int idx = dt + 256 * i;
                const int chunk = idx & 3, pix = idx >> 2;
                const int pr = (pix * recip) >> 16, pc = pix - pr * pc_n;
                const int iy = it.y0 - 1 + pr, ix = it.x0 - 1 + pc;
                ro[i] = OOBV;
                if (pix < npix && (unsigned)iy < (unsigned)sg.H && (unsigned)ix < (unsigned)sg.W)
                    ro[i] = (base_n + (unsigned)((iy * sg.W + ix) * Cin + chunk * 4)) * 4u;
            }
            const int t_ty = t_tile >> lbw, t_tx = t_tile & (bw - 1);
            g0 = (unsigned)((((2 * t_ty + t_half) * pc_n + 2 * t_tx) * RCS + t_chunk) * 16);
            g1 = g0 + (unsigned)(pc_n * RCS * 16);
            g2 = g1 + (unsigned)(pc_n * RCS * 16);
        };
        auto request_item_data = [&]() {
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
#ifdef ERD_WX3_NORAW       // timing probe: no global loads of the raw patch (results are wrong)
#pragma unroll
                for (int i = 0; i < NCH; ++i) dst[i] = make_float4((float)la_soff, 1.f, 2.f, 3.f);
#else
#pragma unroll
                for (int i = 0; i < NCH; ++i) dst[i] = buf_load16_s(rs_in, roff[i], la_soff);
#endif
                la_soff += KS * 4;
                ++la_ks;
                if (la_ks == 3) {
                    const int nx = __builtin_amdgcn_readfirstlane(sh_item[(k_la + 1) & 1]);
                    nx_valid = nx < nitems;
                    if (nx_valid) nx_it = decode(nx);
                }
                if (la_ks == 4 && nx_valid) item_geometry(nx_it, nx_roff, nx_rd0, nx_rd1, nx_rd2, nx_rs);
                if (la_ks == nks) {
                    if (nx_valid) {
                        la = nx_it;
                        rs_in = nx_rs;
#pragma unroll
                        for (int i = 0; i < NCH; ++i) roff[i] = nx_roff[i];
                        nrd0 = nx_rd0; nrd1 = nx_rd1; nrd2 = nx_rd2;
                        la_ks = 0;
                        la_soff = 0;
                        ++k_la;
                        slices_total += nks;
                        request_item_data();
                    } else la_valid = false;
                }
            }
        };
        char* const sm = smem;
        const unsigned st_base = (unsigned)(((dt >> 2) * RCS + (dt & 3)) * 16);
        // this thread's 8 bytes inside a (position, limb) block: tile row 32 B, channels 4 t_chunk .. + 3
        // V block of a (position, limb): 1 KB = [k half h = channel / 8][tile 32][16 B].  A matrix wave's B-fragment read (lane = tile,
        // h) then covers 256 CONTIGUOUS bytes per 16-lane service group -- with 32-byte tile rows ([tile][16 channels]) the sixteen
        // 16-byte pieces of a group were spread over 512 bytes and every bank was hit twice.  The tiles of half 1 are XORed with 8
        // so that the 8-byte stores of a data wave's 16-lane group (4 tiles x 4 channel quads: both halves) land on different banks.
        const unsigned wr_base = VOFF + (unsigned)((t_chunk >> 1) * 512 + ((t_tile ^ ((t_chunk >> 1) * 8)) * 16) + (t_chunk & 1) * 8);
        auto store_raw = [&](const float4* src, auto par_tag) {
            constexpr unsigned PAR = decltype(par_tag)::value;
#pragma unroll
            for (int i = 0; i < NCH; ++i) *reinterpret_cast<float4*>(sm + PAR * RAWB + st_base + i * (64 * RCS * 16)) = src[i];
        };
        auto put = [&](char* dst, const float4 v) {        // four channels of one position -> three limb words of 8 bytes
            uint2 hi, mid, lo;
#ifdef ERD_WX3_NOSPLIT      // timing probe (tools/build_probe.sh): what the limb split costs the data waves (results are wrong)
            hi.x = __builtin_amdgcn_perm(__float_as_uint(v.y), __float_as_uint(v.x), 0x07060302u);
            hi.y = __builtin_amdgcn_perm(__float_as_uint(v.w), __float_as_uint(v.z), 0x07060302u);
            mid = hi; lo = hi;
#else
            erd::limbs3_pair(v.x, v.y, hi.x, mid.x, lo.x);
            erd::limbs3_pair(v.z, v.w, hi.y, mid.y, lo.y);
#endif
            *reinterpret_cast<uint2*>(dst) = hi;
#ifdef ERD_WX3_VWRITE1     // timing probe: only the first limb is stored (results are wrong)
            asm volatile("" :: "v"(mid.x), "v"(mid.y), "v"(lo.x), "v"(lo.y));
#else
            *reinterpret_cast<uint2*>(dst + 1024) = mid;
            *reinterpret_cast<uint2*>(dst + 2048) = lo;
#endif
        };
        // the transform in two halves so that its 12 LDS reads can be issued BEFORE the raw store / next loads of the same
        // iteration (they touch the other raw buffer) and travel under them: (1) the patch rows of this thread's (tile, 4 channels)
        float4 pd[3][4];
        auto transform_read = [&](auto rpar_tag) {
            constexpr unsigned RPAR = decltype(rpar_tag)::value;
            const char* r0 = sm + RPAR * RAWB + rd0;
            const char* r1 = sm + RPAR * RAWB + rd1;
            const char* r2 = sm + RPAR * RAWB + rd2;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                pd[0][c] = *reinterpret_cast<const float4*>(r0 + c * (RCS * 16));
                pd[1][c] = *reinterpret_cast<const float4*>(r1 + c * (RCS * 16));
                pd[2][c] = *reinterpret_cast<const float4*>(r2 + c * (RCS * 16));
            }
            if (--tr_left == 0) {                             // the next slice belongs to the item the pointer entered last
                rd0 = nrd0;
                rd1 = nrd1;
                rd2 = nrd2;
                tr_left = nks;
            }
        };
        // (2) rows (2 HALF, 2 HALF + 1) of B^T d B, split into limbs, into V buffer VPAR
        auto transform_write = [&](auto vpar_tag, auto half_tag) {
            constexpr unsigned VPAR = decltype(vpar_tag)::value;
            constexpr int HALF = decltype(half_tag)::value;
            float4 rr[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 d0 = pd[0][c], d1 = pd[1][c], d2 = pd[2][c];
                if (HALF == 0) { rr[0][c] = f4sub(d0, d2); rr[1][c] = f4add(d1, d2); }
                else           { rr[0][c] = f4sub(d1, d0); rr[1][c] = f4sub(d0, d2); }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {                  // positions (2 HALF + a) * 4 + {0, 1, 2, 3}: 3 KB apart
                char* v = sm + wr_base + VPAR * VB + ((2 * HALF + a) * 4) * 3 * 1024;
                put(v + 0 * 3072, f4sub(rr[a][0], rr[a][2]));
                put(v + 1 * 3072, f4add(rr[a][1], rr[a][2]));
                put(v + 2 * 3072, f4sub(rr[a][2], rr[a][1]));
                put(v + 3 * 3072, f4sub(rr[a][1], rr[a][3]));
            }
        };
        using P0 = std::integral_constant<unsigned, 0>;
        using P1 = std::integral_constant<unsigned, 1>;
        auto run = [&](auto half_tag) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) { rv[i] = make_float4(0.f, 0.f, 0.f, 0.f); rvb[i] = rv[i]; }
            item_geometry(la, roff, nrd0, nrd1, nrd2, rs_in);
            request_item_data();
            rd0 = nrd0;
            rd1 = nrd1;
            rd2 = nrd2;
            tr_left = nks;
            flush_pending();
            issue_next(rv);
            issue_next(rvb);
            store_raw(rv, P0{});
            __syncthreads();                                  // P
            transform_read(P0{});
            transform_write(P0{}, half_tag);
            store_raw(rvb, P1{});
            flush_pending();
            issue_next(rv);
            __syncthreads();                                  // B_0
            auto iter = [&](auto par_tag, auto npar_tag) {
                ERD_T0(ts);
                transform_read(npar_tag);                     // raw(g+1): requested first, consumed after the store / issue below
                __builtin_amdgcn_sched_barrier(0);
                store_raw(rv, par_tag);
                flush_pending();
                issue_next(rv);
                ERD_TACC(t_ent, ts);
                ERD_T0(tt);
                transform_write(npar_tag, half_tag);
#ifdef ERD_WINO_TRACE
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                ERD_TACC(t_wait, tt);
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
#ifdef ERD_WINO_TRACE
        if (wave == 0 && lane == 0 && blockIdx.x < 256) {
            g_wino_trace[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memtime() - t_begin;
            g_wino_trace[blockIdx.x * 8 + 5] = t_bar;
            g_wino_trace[blockIdx.x * 8 + 6] = t_ent;
            g_wino_trace[blockIdx.x * 8 + 7] = t_wait;
        }
#endif
    }
}

// =====================================================================================================================
// The three-limb form for 128 OUTPUT CHANNELS PER ITEM with eight PHASED waves (round 5; the default where Cout % 128 == 0 and
// it saves dispatch rounds: wino_launch).  wino_x3_kernel transforms every 32-tile block once per 64 output channels -- raw patch load,
// B^T d B and the limb split run Cout / 64 times per tile (four times on the 256 -> 256 head towers) -- and its four data waves
// (3 240 busy cycles per slice) are the critical path next to 1 536 cycles of matrix work (profiles/r04_wino_x3_trace.txt).
// Here an item is 32 tiles x 128 couts: the same data work per slice feeds TWICE the matrix work.  128 couts x 32 tiles x 16
// positions are 128 accumulator registers on each of EIGHT waves, so there is no room for separate data waves (twelve waves
// would leave 168 registers each): every wave does both jobs, in two PHASES per slice --
//   M: its share of slice g: wave w owns transform ROW w & 3 (positions 4 (w & 3) .. + 3) of cout blocks 2 (w >> 2), + 1 (of
//      four): a position's V fragments (three limbs, 3 KB of LDS reads) feed TWELVE MFMAs (two cout blocks x six limb products) --
//      (a first version with wino_x3_kernel's ownership -- eight positions x one cout block, 24 KB of fragment reads per wave and
//      slice, the flag-polled pair exchange -- ran the head towers in 363 us against 350: profiles/r05_wino_x3p_trace.txt);
//      weight fragments: a ring of four (position, cout block) units straight from L2 as before;
//   D: its eighth of the data work for slice g + 1 (thread t: tile (t >> 2) & 31, channels 4 (t & 3) .., transform row t >> 7:
//      two patch rows from the raw slice, row pass, column pass, limb split, twelve 8-byte stores; 2 x 16 bytes of raw staging);
// -- every wave M then D, one workgroup barrier per slice as before (V(g+1) complete, V(g) and raw(g+1) consumed).  Measured
// alternatives (profiles/r05_wino_x3p_trace.txt): the two waves of a SIMD in OPPOSITE order (one multiplies while the other
// transforms) is 7 % SLOWER -- a VALU wave beside an MFMA stream gets one issue slot per 8 cycles (tools/mfma_valu_coissue.hip) and
// its dependent chains crawl (a data phase takes 3 950 cycles beside a matrix phase, 2 000 beside another data phase); D then M, an
// extra barrier between the phases, raised priority for the transform: level or slower; FOUR waves with 512 registers, one per
// SIMD, the data work woven into the MFMA stream (wino_x3s_kernel, commit 64a9de0): 410 us against 350 -- a lone wave cannot issue
// the ~400 non-matrix instructions of a slice in less than 6 100 cycles.
// Output: z[c] = (M A)[row][c] is local to a wave (the fp32 kernel's order (m0 + m1) + m2, (m1 - m2) - m3); y = A^T z needs three
// rows: wave (row i) finishes output pixel (a, c) = (i >> 1, i & 1) of every tile for its two cout blocks from its own z and two
// others', which travel IN BULK through the V buffer the last slice has just released (two rounds -- one per cout block -- of
// 6 x 4 KB per cout pair, a barrier behind the writes and one behind the reads) instead of wino_x3_kernel's flag-polled
// four-register rounds.  Same V values, same MFMA sequence per accumulator, same order of every sum: results are BIT-identical
// to wino_x3_kernel (tests/test_gpu_wino_x3.py).
constexpr int BNP = 128;                       // output channels per item
constexpr int HPIX = 208;                      // raw pixel slots (the largest patch, 6 x 34, needs 204)

__global__ __launch_bounds__(512, 2) void wino_x3p_kernel(const WinoDesc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: raw0 | raw1 (2 x [HPIX][RCS] float4) | V0 | V1 (2 x VX_B) | sh_ss [4][scale 128 | shift 128] | sh_item [2]
    constexpr unsigned RAWB = HPIX * RCS * 16, VOFF = 2 * RAWB, VB = VX_B, XOFF = VOFF + 2 * VB;
    float* sh_ss = reinterpret_cast<float*>(smem + XOFF);
    int* sh_item = reinterpret_cast<int*>(sh_ss + 4 * 2 * BNP);
    float* sh_gn = reinterpret_cast<float*>(sh_item + 4);      // [4 transform rows][16 groups][2]: the waves' group sums of the finished item
    char* const sm = smem;

    const int tid = threadIdx.x;
    const int Cin = p.Cin;
    const int nks = Cin / KS;
    const int nitems = p.nitems;
    const int ncb32 = (p.Cout + 31) / 32;
    int cur_item = blockIdx.x;

    auto decode = [&](int item) {
        WinoItem it;
        const int nb = item / p.blocks_per_nb;
        int b = item - nb * p.blocks_per_nb;
        int r = 0;
        while (r + 1 < p.nreg && b >= p.reg[r + 1].block0) ++r;
        const WinoRegion& rg = p.reg[r];
        b -= rg.block0;
        const int per_img = rg.nby * rg.nbx;
        const int n = b / per_img;
        const int rem = b - n * per_img;
        const int by = rem / rg.nbx, bx = rem - by * rg.nbx;
        const int lbw = rg.lbw;
        it.s = __builtin_amdgcn_readfirstlane(rg.seg);
        it.n = __builtin_amdgcn_readfirstlane(n);
        it.y0 = __builtin_amdgcn_readfirstlane(2 * (rg.ty0 + by * (32 >> lbw)));
        it.x0 = __builtin_amdgcn_readfirstlane(2 * (rg.tx0 + (bx << lbw)));
        it.cout0 = __builtin_amdgcn_readfirstlane(nb * BNP);
        it.lbw = __builtin_amdgcn_readfirstlane(lbw);
        it.yl = __builtin_amdgcn_readfirstlane(min(p.seg[rg.seg].H, 2 * rg.ty1));
        it.xl = __builtin_amdgcn_readfirstlane(min(p.seg[rg.seg].W, 2 * rg.tx1));
        return it;
    };
    auto claim = [&](int k) -> int {
        return p.sched ? (int)gridDim.x + atomicAdd(p.sched, 1) : (int)blockIdx.x + (k + 1) * (int)gridDim.x;
    };

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int item0 = blockIdx.x;
    if (item0 >= nitems) return;

    // ---- roles -----------------------------------------------------------------------------------------------------------
    const int li = lane & 31, h = lane >> 5;
    const int ri = wave & 3, cp = wave >> 2;                         // matrix work: transform row (four positions), cout pair (of two)
    const int t_chunk = tid & 3, t_tile = (tid >> 2) & 31;          // data work: (tile, 4 channels) ...
    const int t_row = __builtin_amdgcn_readfirstlane(tid >> 7);     // ... and transform row (uniform per wave)
    // row pass of B^T d B: R = d[A] +- d[B] with (A, B, sign) = (0, 2, -), (1, 2, +), (2, 1, -), (1, 3, -)
    const int rowA = t_row == 0 ? 0 : (t_row == 2 ? 2 : 1), rowB = t_row == 2 ? 1 : (t_row == 3 ? 3 : 2);

    // ---- matrix side state (wino_x3_kernel's matrix waves) ------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rs_U = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.U3), 0, (int)((size_t)16 * 3 * ncb32 * nks * 1024), 0x00020000);
    const unsigned u_lane = (unsigned)lane * 16u;
    const unsigned per_xl_b = (unsigned)ncb32 * (unsigned)nks * 1024u;          // bytes per (position, limb)
    const unsigned pos0_b = (unsigned)(ri * 4 * 3) * per_xl_b;                  // this wave's first position
    const unsigned cb_b = (unsigned)nks * 1024u;                                // bytes from a cout block to the next inside a plane
    const unsigned v_lane = (unsigned)(ri * 4 * 3 * 1024 + h * 512 + ((li ^ (h * 8)) * 16));
    WinoItem cur = decode(item0);
    int k_item = 0;
    unsigned u_item = (unsigned)__builtin_amdgcn_readfirstlane(((cur.cout0 >> 5) + 2 * cp) * nks * 1024);
    // unit u = 2 j + b: position 4 ri + j, cout block 2 cp + b
    f32x16 acc[8];
    u32x4 ub[4][3];                                                             // weight-fragment ring: unit u lives in slot u & 3
    bf16x8 vf[2][3];                                                            // tile fragments [position parity][limb]
    auto load_u = [&](const int u, const unsigned soff) {                      // u: compile-time after unrolling
#pragma unroll
        for (int l = 0; l < 3; ++l)
            ub[u & 3][l] = __builtin_amdgcn_raw_buffer_load_b128(rs_U, u_lane, pos0_b + (unsigned)((u >> 1) * 3 + l) * per_xl_b + soff + (u & 1) * cb_b, 0);
    };

    // ---- data side state (wino_conv_kernel's data waves, on 512 threads) --------------------------------------------------
    // The look-ahead pointer (raw slices are requested three slices ahead of the matrix slice) enters the next item inside ONE
    // data phase: decode + this thread's two patch offsets + its two patch rows are ~60 instructions once per item (the role-split
    // kernels stage them over three slices because their data waves are the critical path; here every wave carries an eighth).
    WinoItem la = cur;                                   // the item of the look-ahead pointer
    int la_ks = 0, k_la = 0;
    unsigned la_soff = 0;
    bool la_valid = true;
    unsigned roff[2];
    float4 rv[2];
    __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.seg[0].in), 0, 0, 0x00020000);
    unsigned rdA = 0, rdB = 0, nrdA = 0, nrdB = 0;       // the transform's patch rows (it runs two slices behind the pointer) / of the item entered last
    int tr_left = 0;
    auto item_geometry = [&](const WinoItem& it, unsigned (&ro)[2], unsigned& gA, unsigned& gB, __amdgpu_buffer_rsrc_t& rs) {
        const WinoSeg& sg = p.seg[it.s];
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.in), 0, (int)((long long)sg.N * sg.in_nstride * 4), 0x00020000);
        const int lbw = it.lbw, bw = 1 << lbw, bh = 32 >> lbw;
        const int pc_n = 2 * bw + 2, npix = (2 * bh + 2) * pc_n;
        const int recip = (65536 + pc_n - 1) / pc_n;
        const unsigned base_n = (unsigned)(it.n * sg.in_nstride);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 512 * i;
            const int chunk = idx & 3, pix = idx >> 2;
            const int pr = (pix * recip) >> 16, pc = pix - pr * pc_n;
            const int iy = it.y0 - 1 + pr, ix = it.x0 - 1 + pc;
            ro[i] = OOBV;
            if (pix < npix && (unsigned)iy < (unsigned)sg.H && (unsigned)ix < (unsigned)sg.W)
                ro[i] = (base_n + (unsigned)((iy * sg.W + ix) * Cin + chunk * 4)) * 4u;
        }
        const int t_ty = t_tile >> lbw, t_tx = t_tile & (bw - 1);
        gA = (unsigned)((((2 * t_ty + rowA) * pc_n + 2 * t_tx) * RCS + t_chunk) * 16);
        gB = (unsigned)((((2 * t_ty + rowB) * pc_n + 2 * t_tx) * RCS + t_chunk) * 16);
    };
    // the scale / shift slice of the item the pointer has just entered and the claim of the item after it: requested on entry,
    // written to LDS at the end of the same phase (readable behind the slice's barrier)
    struct Pending { float sc, sh; int claim, k; };
    auto request_item_data = [&](Pending& pe) {
        if (tid < BNP) {
            const int co = la.cout0 + tid;
            pe.sc = (p.scale && co < p.Cout) ? p.scale[co] : 1.f;
            pe.sh = (p.shift && co < p.Cout) ? p.shift[co] : 0.f;
        }
        if (tid == BNP) pe.claim = claim(k_la);
        pe.k = k_la;
    };
    auto flush_pending = [&](const Pending& pe) {
        if (pe.k >= 0) {
            if (tid < BNP) {
                float* ss = sh_ss + (pe.k & 3) * (2 * BNP);
                ss[tid] = pe.sc;
                ss[BNP + tid] = pe.sh;
            }
            if (tid == BNP) sh_item[(pe.k + 1) & 1] = pe.claim;
        }
    };
    // request the next raw slice of the look-ahead item ...
    auto issue_next = [&](float4* dst) {
        if (la_valid) {
#pragma unroll
            for (int i = 0; i < 2; ++i) dst[i] = buf_load16_s(rs_in, roff[i], la_soff);
            la_soff += KS * 4;
            ++la_ks;
        }
    };
    // ... and, at the END of the phase (the transform's registers are dead), move the pointer on if that was its item's last slice
    auto advance_item = [&](Pending& pe) {
        if (la_valid && la_ks == nks) {                   // the pointer leaves item k_la (its successor was claimed >= 3 slices ago)
            const int nx = __builtin_amdgcn_readfirstlane(sh_item[(k_la + 1) & 1]);
            if (nx < nitems) {
                la = decode(nx);
                item_geometry(la, roff, nrdA, nrdB, rs_in);
                la_ks = 0;
                la_soff = 0;
                ++k_la;
                request_item_data(pe);
            } else la_valid = false;
        }
    };
    const bool st1_ok = (tid >> 2) + 128 < HPIX;
    auto store_raw = [&](const float4* src, const unsigned par) {
        // pixel tid >> 2 (+ 128 for the second chunk); recomputed per slice (three instructions) rather than kept in a register
        // across the matrix phase -- the allocator otherwise spills it and its reload waits for EVERY load in flight
        int t = tid;
        asm volatile("" : "+v"(t));
        const unsigned st_base = (unsigned)(((t >> 2) * RCS + (t & 3)) * 16);
        *reinterpret_cast<float4*>(sm + par * RAWB + st_base) = src[0];
        if (st1_ok) *reinterpret_cast<float4*>(sm + par * RAWB + st_base + 128 * RCS * 16) = src[1];
    };
    float4 pd[2][4];
    auto transform_read = [&](const unsigned rpar) {
        const char* rA = sm + rpar * RAWB + rdA;
        const char* rB = sm + rpar * RAWB + rdB;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            pd[0][c] = *reinterpret_cast<const float4*>(rA + c * (RCS * 16));
            pd[1][c] = *reinterpret_cast<const float4*>(rB + c * (RCS * 16));
        }
        if (--tr_left == 0) {                             // the next slice belongs to the item the pointer entered last
            rdA = nrdA;
            rdB = nrdB;
            tr_left = nks;
        }
    };
    // this thread's 8 bytes inside a (position, limb) block of 1 KB = [k half][tile ^ 8 (k half)][8 channels] (wino_x3_kernel's layout)
    // (recomputed where it is used -- eight instructions per slice -- for the same reason as store_raw's base)
    auto v_write_base = [&]() -> unsigned {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int c = t & 3, tl = (t >> 2) & 31;
        return VOFF + (unsigned)(t_row * 4 * 3072 + (c >> 1) * 512 + ((tl ^ ((c >> 1) * 8)) * 16) + (c & 1) * 8);
    };
    auto put = [&](char* dst, const float4 v) {        // four channels of one position -> three limb words of 8 bytes
        uint2 hi, mid, lo;
        erd::limbs3_pair(v.x, v.y, hi.x, mid.x, lo.x);
        erd::limbs3_pair(v.z, v.w, hi.y, mid.y, lo.y);
        *reinterpret_cast<uint2*>(dst) = hi;
        *reinterpret_cast<uint2*>(dst + 1024) = mid;
        *reinterpret_cast<uint2*>(dst + 2048) = lo;
    };
    // D phase: raw(g+1) -> V(g+1) (this thread's row of four positions), raw(g+2) regs -> LDS over raw(g), request raw(g+3) (last)
    // row pass of B^T d B for this wave's transform row (uniform branch: a packed add or a packed subtract, no sign operand)
    auto row_pass = [&](float4 (&R)[4]) __attribute__((always_inline)) {
        if (t_row == 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) R[c] = f4add(pd[0][c], pd[1][c]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) R[c] = f4sub(pd[0][c], pd[1][c]);
        }
    };
    unsigned long long t_d1 = 0, t_d2 = 0, t_d3 = 0, t_out = 0; (void)t_d1; (void)t_d2; (void)t_d3; (void)t_out;
    auto data_phase = [&](const unsigned par, const unsigned u_tail) __attribute__((always_inline)) {
        const unsigned npar = par ^ 1u;
        ERD_T0(ta);
        transform_read(npar);
        __builtin_amdgcn_sched_barrier(0);
        store_raw(rv, par);
#ifdef ERD_WINO_TRACE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        ERD_TACC(t_d1, ta);
        ERD_T0(tb2);
        float4 R[4];
        row_pass(R);
        char* v = sm + v_write_base() + npar * VB;
        put(v + 0 * 3072, f4sub(R[0], R[2]));
        put(v + 1 * 3072, f4add(R[1], R[2]));
        put(v + 2 * 3072, f4sub(R[2], R[1]));
        put(v + 3 * 3072, f4sub(R[1], R[3]));
        __builtin_amdgcn_sched_barrier(0);
#ifdef ERD_WINO_TRACE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        ERD_TACC(t_d2, tb2);
        ERD_T0(tc3);
        Pending pe;
        pe.sc = 1.f; pe.sh = 0.f; pe.claim = 0; pe.k = -1;
        advance_item(pe);
        flush_pending(pe);
        // the ring slots of positions 2 and 3 are empty across this phase (its registers are the transform's); their fragments
        // for the NEXT matrix phase are requested here, two positions (~400 matrix cycles) ahead of their use
        __builtin_amdgcn_sched_barrier(0);
        load_u(2, u_tail);
        load_u(3, u_tail);
        // raw(g+3) is requested BEHIND the ring's fragments: loads retire in issue order, so the next matrix phase's wait for slot 2
        // would otherwise also wait for these HBM reads, which nobody needs before the next data phase (fpn P3 262 -> 251 us,
        // head towers 324 -> 322, bit-identical: profiles/r05_thin_forms.txt section 6)
        issue_next(rv);
        ERD_TACC(t_d3, tc3);
    };
    // M phase: this wave's four positions x two cout blocks of slice g; the ring slot of unit u is re-loaded with unit u + 4 of
    // this slice or u - 4 of the next one as soon as its six MFMAs have been issued
    auto matrix_phase = [&](const unsigned par, const unsigned u_cur, const unsigned u_reload) __attribute__((always_inline)) {
        const char* vc = sm + VOFF + v_lane + par * VB;
#pragma unroll
        for (int l = 0; l < 3; ++l) vf[0][l] = *reinterpret_cast<const bf16x8*>(vc + l * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 3) {
#pragma unroll
                for (int l = 0; l < 3; ++l) vf[(j + 1) & 1][l] = *reinterpret_cast<const bf16x8*>(vc + (j + 1) * 3072 + l * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);      // reads first: they travel behind this position's MFMAs
            const bf16x8 vh = vf[j & 1][0], vm = vf[j & 1][1], vl = vf[j & 1][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int u = 2 * j + b;
                const bf16x8 uh = __builtin_bit_cast(bf16x8, ub[u & 3][0]), um = __builtin_bit_cast(bf16x8, ub[u & 3][1]),
                             ul = __builtin_bit_cast(bf16x8, ub[u & 3][2]);
                // rows = couts (U), columns = tiles (V); smallest terms first, as everywhere in the three-limb kernels
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ul, vh, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(um, vh, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vl, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(um, vm, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vm, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vh, acc[u], 0, 0, 0);
                // the slot is free: unit u + 4 of this slice, or unit u - 4 of the next one for u = 4, 5 (slots 2 and 3 stay
                // empty until the end of the following data phase)
                if (u + 4 < 8) load_u(u + 4, u_cur); else if (u < 6) load_u(u - 4, u_reload);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // Output stage of the finished item (behind the last slice's barrier: V buffer xpar is free, every wave is here).
    // Exchange slots of a cout pair (4 KB each = [4 register quads][64 lanes][16 B]): 0: z1[0], 1: z2[0], 2: z3[0], 3: z0[1],
    // 4: z2[1], 5: z1[1]  (z0[0] and z3[1] are only needed by their owners).
    auto output_stage = [&](const unsigned xpar) __attribute__((always_inline)) {
        const WinoSeg& sg = p.seg[cur.s];
        const float* ss = sh_ss + (k_item & 3) * (2 * BNP);
        const int lbw = cur.lbw, bwm = (1 << lbw) - 1;
        const int ty = li >> lbw, tx = li & bwm;
        const int fa = ri >> 1, fc = ri & 1;                           // this wave finishes output pixel (fa, fc) of every tile
        const int oy = cur.y0 + 2 * ty + fa, ox = cur.x0 + 2 * tx + fc;
        const bool pix_ok = oy < cur.yl && ox < cur.xl;
        const int64_t opix = cur.n * sg.out_nstride + ((int64_t)oy * sg.W + ox) * p.Cout;
        const bool simple = !sg.res && !sg.mask && !p.colsum;      // (gn_stats: the simple path only -- wino_launch checks)
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
            sg.out, 0, (int)((long long)sg.N * sg.out_nstride * 4), 0x00020000);
        const float lo = p.relu ? 0.f : -__builtin_inff();
        char* const xb = sm + VOFF + xpar * VB + cp * (6 * 4096) + lane * 16;
        // slots this wave writes (its z[0] / z[1]; -1: nobody else needs it) and reads (the two other rows of its output, ascending)
        const int w0 = ri == 1 ? 0 : ri == 2 ? 1 : ri == 3 ? 2 : -1;
        const int w1 = ri == 0 ? 3 : ri == 1 ? 5 : ri == 2 ? 4 : -1;
        const int rA = ri == 0 ? 0 : ri == 1 ? 3 : ri == 2 ? 0 : 5;
        const int rB = ri == 0 ? 1 : ri == 1 ? 4 : ri == 2 ? 2 : 4;
        // residual / mask values of BOTH rounds are requested before the first round's stores: gfx950 retires loads and stores through one
        // in-order counter (vmcnt), so a load issued behind a store cannot be waited for without waiting for that store's
        // acknowledgement too -- the per-channel-group "load, wait, store" order of rounds 1-4 serialized eight acknowledgements per
        // item (the masked input-gradient launches ran 1.2-1.7x their forward twins).  Pixels outside the map carry the offset OOBV.
        const int map_bytes = (int)((long long)sg.N * sg.out_nstride * 4);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.res ? sg.res : sg.out), 0, sg.res ? map_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_msk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sg.mask ? sg.mask : sg.out), 0, sg.mask ? map_bytes : 0, 0x00020000);
        float4 pr[2][4], pm[2][4];
        auto out_off = [&](int b, int gq) -> unsigned {
            return pix_ok ? (unsigned)((opix + cur.cout0 + 32 * (2 * cp + b) + 8 * gq + 4 * h) * 4) : OOBV;
        };
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            // z[c] = (M A)[row][c] of cout block b: the accumulators die here
            const f32x16 z0 = (acc[0 + b] + acc[2 + b]) + acc[4 + b];
            const f32x16 z1 = (acc[2 + b] - acc[4 + b]) - acc[6 + b];
            if (w0 >= 0) {
#pragma unroll
                for (int qr = 0; qr < 4; ++qr)
                    *reinterpret_cast<float4*>(xb + w0 * 4096 + qr * 1024) = make_float4(z0[4 * qr], z0[4 * qr + 1], z0[4 * qr + 2], z0[4 * qr + 3]);
            }
            if (w1 >= 0) {
#pragma unroll
                for (int qr = 0; qr < 4; ++qr)
                    *reinterpret_cast<float4*>(xb + w1 * 4096 + qr * 1024) = make_float4(z1[4 * qr], z1[4 * qr + 1], z1[4 * qr + 2], z1[4 * qr + 3]);
            }
            __syncthreads();
            f32x16 sa, sb;
#pragma unroll
            for (int qr = 0; qr < 4; ++qr) {
                const float4 a4 = *reinterpret_cast<const float4*>(xb + rA * 4096 + qr * 1024);
                const float4 b4 = *reinterpret_cast<const float4*>(xb + rB * 4096 + qr * 1024);
                sa[4 * qr] = a4.x; sa[4 * qr + 1] = a4.y; sa[4 * qr + 2] = a4.z; sa[4 * qr + 3] = a4.w;
                sb[4 * qr] = b4.x; sb[4 * qr + 1] = b4.y; sb[4 * qr + 2] = b4.z; sb[4 * qr + 3] = b4.w;
            }
            // y[0][c] = (z0 + z1) + z2 ; y[1][c] = (z1 - z2) - z3 with (own, sa, sb) put back in row order (the fp32 kernel's sums)
            f32x16 yv;
            if (ri == 0) yv = (z0 + sa) + sb;            // own = z0[0], sa = z1[0], sb = z2[0]
            else if (ri == 1) yv = (sa + z1) + sb;       // sa = z0[1], own = z1[1], sb = z2[1]
            else if (ri == 2) yv = (sa - z0) - sb;       // sa = z1[0], own = z2[0], sb = z3[0]
            else yv = (sa - sb) - z1;                    // sa = z1[1], sb = z2[1], own = z3[1]
            if (!simple && b == 0) {                     // (here, not earlier: until now the accumulators of both rounds were live)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        if (sg.res) pr[bb][gq] = buf_load16_s(rs_res, out_off(bb, gq), 0);
                        if (sg.mask) pm[bb][gq] = buf_load16_s(rs_msk, out_off(bb, gq), 0);
                    }
            }
            float4 cs[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {                          // registers 4 gq .. 4 gq + 3: couts 8 gq + 4 h + {0..3} of the block
                const int cl = 32 * (2 * cp + b) + 8 * gq + 4 * h;   // 0..127 inside the item's cout block
                const int co0 = cur.cout0 + cl;
                const float4 sc = *reinterpret_cast<const float4*>(ss + cl);
                const float4 sh = *reinterpret_cast<const float4*>(ss + BNP + cl);
                float4 v = make_float4(yv[4 * gq] * sc.x + sh.x, yv[4 * gq + 1] * sc.y + sh.y, yv[4 * gq + 2] * sc.z + sh.z,
                                       yv[4 * gq + 3] * sc.w + sh.w);
                cs[gq] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (simple) {
                    v = make_float4(fmaxf(v.x, lo), fmaxf(v.y, lo), fmaxf(v.z, lo), fmaxf(v.w, lo));
                    u32x4 o;
                    o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, pix_ok ? (unsigned)((opix + co0) * 4) : OOBV, 0, 0);
                    if (pix_ok) cs[gq] = v;
                } else {
                    if (sg.res) v = f4add(v, pr[b][gq]);
                    if (p.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                    if (sg.mask) {
                        const float4 mk = pm[b][gq];
                        v = make_float4(mk.x > 0.f ? v.x : 0.f, mk.y > 0.f ? v.y : 0.f, mk.z > 0.f ? v.z : 0.f,
                                        mk.w > 0.f ? v.w : 0.f);
                    }
                    u32x4 o;
                    o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, out_off(b, gq), 0, 0);
                    if (pix_ok) cs[gq] = v;
                }
            }
            if (p.gn_part) {
                // GroupNorm statistics from the producer: a group is 8 consecutive channels = the registers 4 gq .. + 3 of the lanes
                // h = 0 and h = 1; this wave holds one pixel of each of its 32 tiles (every pixel of the map is stored -- and counted --
                // exactly once: pix_ok).  Per (cout block, gq): two wave sums into LDS; behind the stage's last barrier 32 threads add
                // the four transform rows' shares in a fixed order and write the item's 16 x 2 sums -- no atomics (a first version
                // with f64 atomics per wave ran the head towers 1.35x SLOWER: ~500 same-address atomics per statistic, and every later
                // load of the wave waits for them in the in-order counter), deterministic.
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 c4 = cs[gq];
                    const float s1 = erd::wave_sum_dpp((c4.x + c4.y) + (c4.z + c4.w));
                    const float s2 = erd::wave_sum_dpp((c4.x * c4.x + c4.y * c4.y) + (c4.z * c4.z + c4.w * c4.w));
                    if (lane == 0) *reinterpret_cast<float2*>(sh_gn + ((ri * 16) + (2 * cp + b) * 4 + gq) * 2) = make_float2(s1, s2);
                }
            }
            if (p.colsum) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) {
                        cs[gq].x += __shfl_xor(cs[gq].x, o, 64); cs[gq].y += __shfl_xor(cs[gq].y, o, 64);
                        cs[gq].z += __shfl_xor(cs[gq].z, o, 64); cs[gq].w += __shfl_xor(cs[gq].w, o, 64);
                    }
                    const int co0 = cur.cout0 + 32 * (2 * cp + b) + 8 * gq + 4 * h;
                    if (li == 0) {
                        float* cpt = p.colsum + (p.colsum_copies > 1 ? (blockIdx.x & (p.colsum_copies - 1)) * p.Cout : 0) + co0;
                        atomicAdd(cpt + 0, cs[gq].x); atomicAdd(cpt + 1, cs[gq].y); atomicAdd(cpt + 2, cs[gq].z); atomicAdd(cpt + 3, cs[gq].w);
                    }
                }
            }
            __syncthreads();                              // the slots are free: next round's writes / the next slice's V stores
        }
        if (p.gn_part && tid < 32) {                      // (behind the last round's barrier: all 4 x 16 x 2 shares are in LDS)
            const float t = ((sh_gn[tid] + sh_gn[32 + tid]) + sh_gn[64 + tid]) + sh_gn[96 + tid];
            p.gn_part[(int64_t)cur_item * 32 + tid] = t;
        }
    };

    // ---- prologue ---------------------------------------------------------------------------------------------------------
    {
        float4 rvb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { rv[i] = make_float4(0.f, 0.f, 0.f, 0.f); rvb[i] = rv[i]; }
        item_geometry(la, roff, nrdA, nrdB, rs_in);
        Pending pe;
        pe.sc = 1.f; pe.sh = 0.f; pe.claim = 0; pe.k = -1;
        request_item_data(pe);
        rdA = nrdA;
        rdB = nrdB;
        tr_left = nks;
        flush_pending(pe);
        pe.k = -1;
        issue_next(rv);                                   // raw(0)
        issue_next(rvb);                                  // raw(1)   (nks >= 4: the pointer cannot leave the first item here)
#pragma unroll
        for (int q = 0; q < 4; ++q) load_u(q, u_item);
        store_raw(rv, 0);
        __syncthreads();                                  // raw(0) is in LDS, the first claim is published
        transform_read(0);
        float4 R[4];
        row_pass(R);
        char* v = sm + v_write_base();
        put(v + 0 * 3072, f4sub(R[0], R[2]));
        put(v + 1 * 3072, f4add(R[1], R[2]));
        put(v + 2 * 3072, f4sub(R[2], R[1]));
        put(v + 3 * 3072, f4sub(R[1], R[3]));
        store_raw(rvb, 1);
        issue_next(rv);                                   // raw(2)
        __syncthreads();                                  // V(0) complete, raw(1) in LDS
    }

    // ---- the slice stream: iteration g multiplies slice g (M) and transforms raw(g+1) -> V(g+1), stores raw(g+2), requests
    //      raw(g+3) (D); waves 0-3: M then D, waves 4-7: D then M (uniform branches around the two call sites of D) ----
    unsigned long long t_bar = 0, t_m = 0, t_d = 0; (void)t_bar; (void)t_m; (void)t_d;
    ERD_T0(t_begin);
    int g = 0;
    for (;;) {
        const int nxt_item = __builtin_amdgcn_readfirstlane(sh_item[(k_item + 1) & 1]);
        const bool has_next = nxt_item < nitems;
        const WinoItem nxt = has_next ? decode(nxt_item) : cur;
        const unsigned u_next = (unsigned)__builtin_amdgcn_readfirstlane(((nxt.cout0 >> 5) + 2 * cp) * nks * 1024);
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        for (int ks = 0; ks < nks; ++ks, ++g) {
            const unsigned par = (unsigned)(g & 1);
            const bool last = ks + 1 == nks;
            const unsigned u_cur = (unsigned)__builtin_amdgcn_readfirstlane((int)(u_item + (unsigned)ks * 1024u));
            const unsigned u_reload = (unsigned)__builtin_amdgcn_readfirstlane(
                (int)(last ? u_next : u_item + (unsigned)(ks + 1) * 1024u));
            { ERD_T0(tm); matrix_phase(par, u_cur, u_reload); ERD_TACC(t_m, tm); }
            __builtin_amdgcn_sched_barrier(0);
            { ERD_T0(td); data_phase(par, u_reload); ERD_TACC(t_d, td); }
            { ERD_T0(tb); __syncthreads(); ERD_TACC(t_bar, tb); }  // V(g+1) complete, V(g) and raw(g+1) consumed, raw(g+2) stored
        }
        { ERD_T0(to); output_stage((unsigned)((g - 1) & 1)); ERD_TACC(t_out, to); }
        if (!has_next) {
#ifdef ERD_WINO_TRACE       // [0..7] wave 0, [8..15] wave 4 (its SIMD mate): total, barrier wait, matrix phases, data phases (of which:
                            // reads + raw store, transform + V stores, item switch + ring tail), output stage
            if ((wave & 3) == 0 && lane == 0 && blockIdx.x < 256) {
                unsigned long long* tr = g_wino_trace_p + blockIdx.x * 16 + (wave >> 2) * 8;
                tr[0] = __builtin_amdgcn_s_memtime() - t_begin;
                tr[1] = t_bar;
                tr[2] = t_m;
                tr[3] = t_d;
                tr[4] = t_d1;
                tr[5] = t_d2;
                tr[6] = t_d3;
                tr[7] = t_out;
            }
#endif
            if (tid == 0 && p.sched) {                       // the last workgroup to leave re-arms the counters
                if (atomicAdd(p.sched + 1, 1) == (int)gridDim.x - 1) { p.sched[0] = 0; p.sched[1] = 0; }
            }
            break;
        }
        cur = nxt;
        cur_item = nxt_item;
        u_item = u_next;
        ++k_item;
    }
}

// GroupNorm statistics of a wino_x3p_kernel launch with gn_part: block (n, s, cout block nb) adds the (sum, sum of squares) that the items
// of image n in every region of segment s wrote for its 16 groups -- 16 item lanes x (16 groups x 2 moments) threads, every lane a fixed
// subsequence of the items, the lanes combined in a fixed order, all in f64: deterministic -- and folds them to (mean, 1 / std) with
// gn_finalize_kernel's arithmetic: mean_rstd[N][nseg][G][2].  (One THREAD per group walking its ~130 items was a 45 us latency chain.)
__global__ __launch_bounds__(512) void wino_gn_finalize_kernel(const WinoDesc p, float* __restrict__ mean_rstd, float eps) {
    const int G = p.Cout >> 3;
    const int nb = blockIdx.x, s = blockIdx.y, n = blockIdx.z;
    if (n >= p.seg[s].N) return;
    const int q = threadIdx.x & 31, il = threadIdx.x >> 5;      // q = group-in-item * 2 + moment; il: item lane (16)
    double acc = 0.0;
    for (int r = 0; r < p.nreg; ++r) {
        const WinoRegion& rg = p.reg[r];
        if (rg.seg != s) continue;
        const int per_img = rg.nby * rg.nbx;
        const int64_t it0 = (int64_t)nb * p.blocks_per_nb + rg.block0 + (int64_t)n * per_img;
        for (int b = il; b < per_img; b += 16) acc += (double)p.gn_part[(it0 + b) * 32 + q];
    }
    __shared__ double red[16][32];
    red[il][q] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        const int gi = threadIdx.x;
        double s1 = 0.0, s2 = 0.0;
        for (int l = 0; l < 16; ++l) { s1 += red[l][2 * gi]; s2 += red[l][2 * gi + 1]; }
        const double m = (double)p.seg[s].H * p.seg[s].W * 8.0;
        const double mean = s1 / m;
        double var = s2 / m - mean * mean;
        if (var < 0) var = 0;
        const int64_t i = ((int64_t)n * p.nseg + s) * G + nb * 16 + gi;
        mean_rstd[i * 2] = (float)mean;
        mean_rstd[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__global__ __launch_bounds__(256) void wino_weight_x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ U3, int Cout,
                                                              int Cin, int flip) {
    const int cop = (Cout + 31) / 32 * 32;
    const int64_t idx = blockIdx.x * 256ll + threadIdx.x;
    if (idx >= (int64_t)cop * Cin) return;
    erd::wino_x3_weight_item(w, U3, Cout, Cin, flip, (int)(idx / Cin), (int)(idx % Cin));
}

}  // namespace

extern "C" int erd_wino_weights(const float* w_ohwi, float* U, int Cout, int Cin, int flip, erd_stream_t stream) {
    ERD_REQUIRE(w_ohwi && U && Cout > 0 && Cin > 0 && Cin % KS == 0, "wino_weights: Cin=%d must be a multiple of %d", Cin, KS);
    const int64_t n = (int64_t)((Cout + 15) / 16 * 16) * Cin;
    hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_ohwi, U,
                       Cout, Cin, flip);
    return erd::check_launch("wino_weights");
}

#ifdef ERD_WINO_TRACE
extern "C" int erd_wino_trace(unsigned long long* out) {       // debug builds only
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino_trace), sizeof(g_wino_trace));
}
extern "C" int erd_wino_trace_p(unsigned long long* out) {     // debug builds only
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino_trace_p), sizeof(g_wino_trace_p));
}
#endif

extern "C" size_t erd_wino_weights_elems(int Cout, int Cin) { return (size_t)16 * ((Cout + 15) / 16 * 16) * Cin; }

namespace {
// the launch both forms share: descriptor (segments, block regions, item count) and the persistent grid
// segments -> descriptor (block regions, blocks per cout block) and the choice between 64 and 128 couts per item; 0 on success
int wino_plan(WinoDesc& d, const erd_conv_seg* segs, int nseg, bool x3, int Cout, int& ncu, bool& phased) {
    ERD_REQUIRE(segs && nseg >= 1 && nseg <= ERD_MAX_SEG, "wino: bad args");
    d.nseg = nseg;
    static const int shapes = getenv("ERD_WINO_SHAPES") ? atoi(getenv("ERD_WINO_SHAPES")) : 1;   // 0: plain 4x8 cover (A/B aid)
    int blocks = 0, nreg = 0;
    auto add_region = [&](int s, int N, int ty0, int tx0, int nby, int nbx, int lbw, int ty1, int tx1) {
        if (nby <= 0 || nbx <= 0) return;
        WinoRegion& r = d.reg[nreg++];
        r.seg = s; r.ty0 = ty0; r.tx0 = tx0; r.nby = nby; r.nbx = nbx; r.lbw = lbw; r.ty1 = ty1; r.tx1 = tx1;
        r.block0 = blocks;
        blocks += N * nby * nbx;
    };
    for (int s = 0; s < nseg; ++s) {
        const erd_conv_seg& g = segs[s];
        ERD_REQUIRE(g.in && g.out && g.IH == g.OH && g.IW == g.OW, "wino: segment %d is not a stride-1 same-size map", s);
        ERD_REQUIRE((int64_t)g.N * g.in_nstride < (1ll << 29) && (int64_t)g.N * g.out_nstride < (1ll << 29),
                    "wino: segment %d too large (32-bit byte offsets)", s);
        WinoSeg& w = d.seg[s];
        w.in = g.in;
        w.out = g.out;
        w.res = g.res;
        w.mask = g.mask;
        ERD_REQUIRE(!g.alpha, "wino: per-level scalars are not supported");
        w.N = g.N;
        w.H = g.IH;
        w.W = g.IW;
        w.in_nstride = g.in_nstride;
        w.out_nstride = g.out_nstride;
        const int TH = (g.IH + 1) / 2, TW = (g.IW + 1) / 2;          // 2x2-pixel tiles of the map
        if (!shapes) {
            add_region(s, g.N, 0, 0, (TH + 3) / 4, (TW + 7) / 8, 3, TH, TW);
            continue;
        }
        // interior: 4x8 blocks; ragged bottom rows: the flattest block that still covers them (1x32 / 2x16 / 4x8);
        // ragged right columns: the narrowest (32x1 / 16x2 / 8x4 / 4x8)
        const int nby = TH / 4, nbx = TW / 8;
        add_region(s, g.N, 0, 0, nby, nbx, 3, 4 * nby, 8 * nbx);
        const int rb = TH - 4 * nby;
        if (rb > 0) {
            const int bh = rb <= 2 ? 2 : 4, bw = 32 / bh;        // (1x32 blocks would need a 4 x 66 patch: not staged)
            add_region(s, g.N, 4 * nby, 0, 1, (TW + bw - 1) / bw, bw == 16 ? 4 : 3, TH, TW);    // the bottom rows, full width
        }
        const int cbw = TW - 8 * nbx;
        if (cbw > 0 && nby > 0) {
            const int bw = cbw <= 2 ? 2 : cbw <= 4 ? 4 : 8, bh = 32 / bw;
            // the right columns of the interior rows only: a tall block may reach below them, into the bottom region
            add_region(s, g.N, 0, 8 * nbx, (4 * nby + bh - 1) / bh, 1, bw == 2 ? 1 : bw == 4 ? 2 : 3, 4 * nby, TW);
        }
    }
    d.nreg = nreg;
    d.blocks_per_nb = blocks;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        static int cached = 0;
        if (cached == 0 && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cached = prop.multiProcessorCount;
        ncu = erd::usable_cus(cached > 0 ? cached : 256);
    }
    // 128 couts per item (wino_x3p_kernel) where the cout count allows it and it needs fewer dispatch rounds: an item of 128 couts
    // costs ~1.6 items of 64 (head towers: 2 800 / 1 400 items, 1.22x faster; 50 x 84 maps: 560 / 280 items on 256 CUs = 3 against 2
    // rounds, 0.97x -- stays on wino_x3_kernel).  ERD_WINO_P=0: never, 2: whenever Cout % 128 == 0 (A/B aids; read per launch, not
    // cached: tests/test_gpu_wino_x3.py flips it inside one process to compare the two kernels bit for bit)
    const char* const e_mode = getenv("ERD_WINO_P");
    const int p_mode = e_mode ? atoi(e_mode) : 1;
    const int64_t i64 = (int64_t)blocks * ((Cout + BN - 1) / BN), i128 = (int64_t)blocks * (Cout / BNP);
    phased = x3 && p_mode != 0 && Cout % BNP == 0 && blocks > 0 &&
             (p_mode >= 2 || 16 * ((i128 + ncu - 1) / ncu) <= 10 * ((i64 + ncu - 1) / ncu));
    return 0;
}

int wino_launch(const erd_conv_seg* segs, int nseg, const float* U, const void* U3, int Cin, int Cout, const float* scale,
                const float* shift, int relu, float* colsum, int colsum_copies, int* sched, hipStream_t stream,
                float* gn_part = nullptr, size_t gn_part_bytes = 0) {
    ERD_REQUIRE(segs && (U || U3) && nseg >= 1 && nseg <= ERD_MAX_SEG, "wino: bad args");
    ERD_REQUIRE(Cin % KS == 0 && Cin >= 4 * KS && Cout > 0, "wino: Cin=%d must be a multiple of %d and at least %d", Cin, KS, 4 * KS);
    WinoDesc d;
    d.U = U;
    d.U3 = U3;
    d.Cin = Cin;
    d.Cout = Cout;
    d.scale = scale;
    d.shift = shift;
    d.relu = relu;
    d.colsum = colsum;
    ERD_REQUIRE(colsum_copies >= 0 && (colsum_copies & (colsum_copies - 1)) == 0, "wino: colsum_copies must be a power of two");
    d.colsum_copies = colsum_copies;
    d.sched = sched;
    d.gn_part = gn_part;
    ERD_REQUIRE(!colsum || Cout % 4 == 0, "wino: colsum needs Cout %% 4 == 0");
    int ncu = 0;
    bool phased = false;
    if (const int rc = wino_plan(d, segs, nseg, U3 != nullptr, Cout, ncu, phased)) return rc;
    const int blocks = d.blocks_per_nb;
    if (blocks == 0) return 0;
    const int ncb = phased ? Cout / BNP : (Cout + BN - 1) / BN;
    d.nitems = blocks * ncb;
    // persistent grid (one workgroup per CU), items claimed from `sched`; ERD_WINO_PERSIST=2: static item stride,
    // 0: one workgroup per item (A/B aids)
    static const int persist = getenv("ERD_WINO_PERSIST") ? atoi(getenv("ERD_WINO_PERSIST")) : 1;
    const int grid = persist ? (d.nitems < ncu ? d.nitems : ncu) : d.nitems;
    if (persist != 1) d.sched = nullptr;
    if (gn_part) {
        ERD_REQUIRE(Cout % 8 == 0, "wino: GroupNorm statistics need Cout %% 8 == 0");
        ERD_REQUIRE(phased && !colsum, "wino: the fused GroupNorm statistics need the 128-couts-per-item kernel (erd_wino_x3_couts_per_item) and no column sums");
        for (int q = 0; q < nseg; ++q) {
            ERD_REQUIRE(!segs[q].res && !segs[q].mask, "wino: the fused GroupNorm statistics serve plain convolutions (segment %d has a residual / mask)", q);
        }
        ERD_REQUIRE(gn_part_bytes >= (size_t)d.nitems * 32 * sizeof(float), "wino: GroupNorm partial-sum workspace too small (%zu bytes for %d items)",
                    gn_part_bytes, d.nitems);
    }
    if (phased) {
        const size_t lds = (size_t)2 * HPIX * RCS * 16 + 2 * VX_B + 4 * 2 * BNP * sizeof(float) + 16 + 4 * 16 * 2 * sizeof(float);
        static bool attrp_done = false;
        if (!attrp_done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_x3p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attrp_done = true;
        }
        hipLaunchKernelGGL(wino_x3p_kernel, dim3((unsigned)grid), dim3(512), lds, stream, d);
        return erd::check_launch("wino_conv3x3_x3p");
    }
    if (U3) {
        const size_t lds = (size_t)2 * RAW_LDS_F4 * sizeof(float4) + 2 * VX_B + 4 * 2 * XCH_B + 64 + 2048 + 16;
        static bool attr3_done = false;
        if (!attr3_done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr3_done = true;
        }
        hipLaunchKernelGGL(wino_x3_kernel, dim3((unsigned)grid), dim3(512), lds, stream, d);
        return erd::check_launch("wino_conv3x3_x3");
    }
    const size_t lds = (size_t)2 * (RAW_LDS_F4 + V_F4) * sizeof(float4) + 2048 + 16;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(wino_conv_kernel, dim3((unsigned)grid), dim3(512), lds, stream, d);
    return erd::check_launch("wino_conv3x3");
}
}  // namespace

extern "C" int erd_wino_conv3x3(const erd_conv_seg* segs, int nseg, const float* U, int Cin, int Cout,
                                const float* scale, const float* shift, int relu, float* colsum, int colsum_copies,
                                int* sched, erd_stream_t stream) {
    ERD_REQUIRE(U, "wino: null U");
    return wino_launch(segs, nseg, U, nullptr, Cin, Cout, scale, shift, relu, colsum, colsum_copies, sched, (hipStream_t)stream);
}

extern "C" size_t erd_wino_weights_x3_elems(int Cout, int Cin) { return (size_t)16 * 3 * ((Cout + 31) / 32 * 32) * Cin; }

extern "C" int erd_wino_weights_x3(const float* w_ohwi, void* U3, int Cout, int Cin, int flip, erd_stream_t stream) {
    ERD_REQUIRE(w_ohwi && U3 && Cout > 0 && Cin > 0 && Cin % KS == 0, "wino_weights_x3: Cin=%d must be a multiple of %d", Cin, KS);
    const int64_t n = (int64_t)((Cout + 31) / 32 * 32) * Cin;
    hipLaunchKernelGGL(wino_weight_x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_ohwi,
                       reinterpret_cast<unsigned short*>(U3), Cout, Cin, flip);
    return erd::check_launch("wino_weights_x3");
}

extern "C" int erd_wino_x3_couts_per_item(const erd_conv_seg* segs, int nseg, int Cout) {
    WinoDesc d;
    int ncu = 0;
    bool phased = false;
    if (wino_plan(d, segs, nseg, true, Cout, ncu, phased)) return -1;
    return phased ? BNP : BN;
}

extern "C" int erd_wino_conv3x3_x3(const erd_conv_seg* segs, int nseg, const void* U3, int Cin, int Cout,
                                   const float* scale, const float* shift, int relu, float* colsum, int colsum_copies,
                                   int* sched, erd_stream_t stream) {
    ERD_REQUIRE(U3, "wino_x3: null U3");
    return wino_launch(segs, nseg, nullptr, U3, Cin, Cout, scale, shift, relu, colsum, colsum_copies, sched, (hipStream_t)stream);
}

/* ABI v6: erd_wino_conv3x3_x3 on a plain convolution (no residual / mask / column sums) that ALSO produces the raw material of the
 * GroupNorm statistics of its result: the output stage of every item writes its 16 groups' (sum, sum of squares) to gn_part (workspace,
 * erd_wino_x3_gn_ws_bytes).  Needs the 128-couts-per-item kernel (erd_wino_x3_couts_per_item == 128).  erd_wino_gn_finalize folds the
 * partial sums to mean_rstd, erd_gn_relu_apply consumes that. */
extern "C" int erd_wino_conv3x3_x3_gn(const erd_conv_seg* segs, int nseg, const void* U3, int Cin, int Cout, const float* scale,
                                      const float* shift, int relu, int* sched, float* gn_part, size_t gn_part_bytes,
                                      erd_stream_t stream) {
    ERD_REQUIRE(U3 && gn_part, "wino_x3_gn: null U3 / gn_part");
    return wino_launch(segs, nseg, nullptr, U3, Cin, Cout, scale, shift, relu, nullptr, 0, sched, (hipStream_t)stream, gn_part, gn_part_bytes);
}

/* the fold: mean_rstd[N][nseg][Cout / 8][2] = (mean, 1 / sqrt(var + eps)) from the items' partial sums of the launch with the same segments */
extern "C" int erd_wino_gn_finalize(const erd_conv_seg* segs, int nseg, int Cout, const float* gn_part, float* mean_rstd, float eps,
                                    erd_stream_t stream) {
    ERD_REQUIRE(segs && gn_part && mean_rstd && Cout % BNP == 0, "wino_gn_finalize: bad args");
    WinoDesc d;
    int ncu = 0, nmax = 0;
    bool phased = false;
    if (const int rc = wino_plan(d, segs, nseg, true, Cout, ncu, phased)) return rc;
    ERD_REQUIRE(phased, "wino_gn_finalize: these segments run on the 64-couts-per-item kernel (no partial sums exist)");
    d.Cout = Cout;
    d.gn_part = const_cast<float*>(gn_part);
    for (int q = 0; q < nseg; ++q) nmax = std::max(nmax, segs[q].N);
    hipLaunchKernelGGL(wino_gn_finalize_kernel, dim3((unsigned)(Cout / BNP), (unsigned)nseg, (unsigned)nmax), dim3(512), 0, (hipStream_t)stream, d,
                       mean_rstd, eps);
    return erd::check_launch("wino_gn_finalize");
}

extern "C" size_t erd_wino_x3_gn_ws_bytes(const erd_conv_seg* segs, int nseg, int Cout) {
    WinoDesc d;
    int ncu = 0;
    bool phased = false;
    if (!segs || wino_plan(d, segs, nseg, true, Cout, ncu, phased) != 0 || !phased) return 0;
    return (size_t)d.blocks_per_nb * (Cout / BNP) * 32 * sizeof(float);
}
