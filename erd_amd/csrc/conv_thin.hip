// Thin-K 1x1 convolutions in the three-limb form ("f32x3") with the ACTIVATIONS STATIONARY in registers.
//
// Why a second kernel.  The stream-K implicit GEMM (conv_mfma.hip) computes one 128 x 128 output tile per workgroup pass:
// tile set-up (row table, first loads), a K loop, an epilogue -- serial phases, two workgroups per CU.  On the 1x1 layers with
// K = Cin <= 128 the K loop is 2-4 slices: of a tile's 31.5 k cycles (128 -> 512 on 100 x 168, round-3 phase trace) 15.0 k are
// K loop and 16.5 k are set-up and epilogue during which that workgroup's matrix pipe share idles; the activation tile is
// re-read and re-split into limbs once per 128 output channels (Cout / 128 times), and every launch ends in a ragged
// dispatch round.  Measured 103 us against ~21 us of matrix work / ~21 us of HBM traffic.
//
// This kernel turns the loop nest around for K <= 128:
//   * a workgroup (4 waves, wave w owns pixel rows 32 w .. 32 w + 31 of a 128-row tile) loads its activation rows ONCE, splits
//     them into bf16 limbs ONCE (erd::limbs3_pair) and keeps all K / 16 MFMA A-fragments of all three limbs in registers
//     (K = 128: 96 VGPRs);
//   * it then walks the output channels in blocks of 32: the block's three weight planes (3 x 32 x K bf16 = 24 KB at K = 128)
//     arrive through a double-buffered LDS ring, one barrier per block, 6 x K / 16 MFMAs per wave and block -- the SAME MFMA
//     sequence per accumulator as the stream-K kernel (k16 steps ascending; per step hi x lo, hi x mid, hi x hi, mid x mid,
//     mid x hi, lo x hi), so results are bit-identical to it;
//   * a block's epilogue (accumulators -> the wave's private LDS block -> 128-byte row segments with scale / shift / residual /
//     ReLU / mask / column sums, 16-byte stores) is issued by the wave right behind the block's MFMAs and drains while the next
//     block's MFMAs run: no tile-level phases.  The block body is STRAIGHT-LINE (round 5): the residual / mask / scale / shift
//     loads of a block are requested before its MFMAs, its four stores are unconditional buffer stores issued back to back (rows
//     past the end carry an out-of-range offset), and the kernel is instantiated per (residual, mask) presence -- gfx950 retires
//     loads and stores through one in-order counter, and with guarded stores and loads between them every store waited for the
//     previous one's acknowledgement (35-75 % of a block in the phase trace, -DERD_THIN_TRACE; profiles/r05_thin_forms.txt);
//   * work = (pixel tile, cout block) pairs in tile-major order, cut into G equal contiguous ranges for a persistent grid of
//     two workgroups per CU: no ragged round, and no fix-up (an output block is owned by one workgroup).
// Bytes per 128 rows x N couts: the activation rows once (K x 512 B) + the weight planes once (6 N K B) instead of N / 128
// times both.  LDS 68 KB -> two workgroups per CU.
//
// Serves: one tap (1x1), Cin in {64, 128}, Cout % 32 == 0, fp32 maps, w_x3 planes; forward (stride 1 or 2) and the
// input-gradient forms (res / mask / colsum / alpha / accumulate) through the same descriptor as erd_conv_igemm, which
// dispatches here (ERD_THIN=0: off, A/B aid).
#include <algorithm>
#include "erd_common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x7fffffffu;      // past num_records of any buffer we build: the load returns zeros

__device__ __forceinline__ u4v buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

struct TRow {
    int in_off;    // element offset of the row's input pixel (channel 0); -1: outside the map / past the end -> zeros
    int out_off;   // element offset of the row's output pixel; -1: row past the end
};

constexpr int SLD = 36;                    // floats per staged row (32 couts + 4: rows land on different banks)

#ifdef ERD_THIN_TRACE      // phase trace (wave 0 of every workgroup): cycles summed per phase over the workgroup's blocks
__device__ unsigned long long g_thin_trace[1024 * 8];
#define THIN_T0(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define THIN_TACC(acc, v) acc += __builtin_amdgcn_s_memtime() - v
#else
#define THIN_T0(v)
#define THIN_TACC(acc, v)
#endif

__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), r, byte_off, 0, 0);
}

// K / 16: MFMA k-steps of the whole reduction (4: Cin = 64, 8: Cin = 128); 32-row groups per wave (1; 2 = 256-row tiles at K = 64,
// half the weight traffic and barriers per output, measured 8 % SLOWER: two workgroups per CU instead of three); residual / mask rows present
template <int KS, int RG, bool RES, bool MSK>
__global__ __launch_bounds__(256, 2) void conv_thin_x3_kernel(const erd_conv_desc p, const int mtiles, const int nb, const int xcd_order) {
    constexpr int K = KS * 16;
    constexpr int CPR = K / 8;                         // 16-byte chunks (8 bf16) per weight row
    constexpr int UNIT_B = 3 * 32 * K * 2;             // one cout block's three planes
    constexpr int NQ = UNIT_B / 16 / 256;              // 16-byte loads per thread and block (6 at K = 128, 3 at K = 64)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                                     // [2][UNIT_B]
    constexpr int TR = 128 * RG;                       // pixel rows of a tile (wave w: rows 32 RG w .. + 32 RG - 1)
    float* stage = reinterpret_cast<float*>(smem + 2 * UNIT_B);            // [4 waves][32][SLD]
    TRow* rows = reinterpret_cast<TRow*>(smem + 2 * UNIT_B + 4 * 32 * SLD * 4);   // [TR]
    float* red = reinterpret_cast<float*>(rows + TR);                    // [2][4 waves][32]: column sums on their way to one atomic per channel

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int G = gridDim.x;
    int wg = blockIdx.x;
    if (xcd_order) {       // workgroup b runs on XCD b % 8: a contiguous chunk of the work list per XCD (conv_igemm_kernel)
        const int q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long long T = (long long)mtiles * nb;
    const long long t_begin = T * wg / G, t_end = T * (wg + 1) / G;
    if (t_begin >= t_end) return;

    const unsigned plane_b = (unsigned)((long long)p.Cout * p.wrow * 2);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_x3), 0, (int)(3u * plane_b), 0x00020000);
    // this thread's share of a block's weight planes: (plane, row, chunk) of load q, the same for every block
    unsigned w_off[NQ];
    int l_off[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int idx = q * 256 + tid;
        const int pl = idx / (32 * CPR), rem = idx - pl * 32 * CPR, row = rem / CPR, ch = rem - row * CPR;
        w_off[q] = (unsigned)pl * plane_b + (unsigned)(row * p.wrow + p.wk[0] + ch * 8) * 2u;
        const int sw = CPR == 16 ? (ch ^ (row & 15)) : (ch ^ ((row >> 1) & 7));     // conflict-free 16-lane fragment reads
        l_off[q] = ((pl * 32 + row) * CPR + sw) * 16;
    }
    u4v rb[NQ];
    auto load_unit = [&](int cbn) {        // cout block cbn's planes
        const unsigned cb_off = (unsigned)(cbn * 32 * p.wrow) * 2u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) rb[q] = buf_load16(rs_w, w_off[q] + cb_off);
    };
    auto store_unit = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) *reinterpret_cast<u4v*>(ring + buf * UNIT_B + l_off[q]) = rb[q];
    };

    u4v ah[RG][KS], am[RG][KS], al[RG][KS];    // the wave's activation rows: limb fragments of every k16 step
    int pend_cb = -1;                      // cout block whose column sums wait in `red` for their atomics
    float* const cs_row = p.colsum ? p.colsum + (p.colsum_copies > 1 ? (int64_t)(blockIdx.x & (p.colsum_copies - 1)) * p.Cout : 0) : nullptr;

#ifdef ERD_THIN_TRACE
    unsigned long long t_tile = 0, t_ring = 0, t_pre = 0, t_mma = 0, t_stage = 0, t_out = 0;
    const unsigned long long t_begin_clk = __builtin_amdgcn_s_memtime();
#endif
    load_unit((int)(t_begin % nb));
    int it = 0;                            // blocks done by this workgroup: parity of the LDS ring / of `red`
#pragma unroll 1
    for (int mt = (int)(t_begin / nb); (long long)mt * nb < t_end; ++mt) {
        // ---- a pixel tile: row table, then this wave's activation rows -> limb fragments ------------------------------------
        int seg_i = 0, mt_in_seg = mt;
        {
            THIN_T0(tt);
#pragma unroll 1
            for (; seg_i < p.nseg - 1; ++seg_i) {
                const int M = p.seg[seg_i].N * p.seg[seg_i].GH * p.seg[seg_i].GW;
                const int tiles = (M + TR - 1) / TR;
                if (mt_in_seg < tiles) break;
                mt_in_seg -= tiles;
            }
            const erd_conv_seg& sg = p.seg[seg_i];
            __syncthreads();               // the previous tile's epilogues are done with the row table
            if (tid < TR) {
                const int GHW = sg.GH * sg.GW, M = sg.N * GHW, m = mt_in_seg * TR + tid;
                TRow ri;
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
            // lane (li, h) holds channels 16 s + 8 h .. + 7 of pixel row li for every step s: two 16-byte loads per step
            u4v xa[RG][KS], xb[RG][KS];
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                const int ibase = rows[(wave * RG + g) * 32 + li].in_off;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const unsigned o = ibase < 0 ? OOB : (unsigned)(ibase + 16 * s + 8 * h) * 4u;
                    xa[g][s] = buf_load16(rs_in, o);
                    xb[g][s] = buf_load16(rs_in, ibase < 0 ? OOB : o + 16u);
                }
            }
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {      // value pair e of the step: words (2 e, 2 e + 1) of the eight fp32 values
                    const u4v& x = e < 2 ? xa[g][s] : xb[g][s];
                    unsigned hi, mid, lo;
                    erd::limbs3_pair(__uint_as_float(x[(e & 1) * 2]), __uint_as_float(x[(e & 1) * 2 + 1]), hi, mid, lo);
                    ah[g][s][e] = hi; am[g][s][e] = mid; al[g][s][e] = lo;
                }
            }
            THIN_TACC(t_tile, tt);
        }
        const erd_conv_seg& sg = p.seg[seg_i];
        // the segment's output-side maps as buffers: rows past the end carry the offset OOB -- their loads return zeros, their stores
        // are dropped, and the block's body below is straight-line code (see the note on s_waitcnt at the stores)
        const int out_bytes = erd::uniform_int((int)((long long)sg.N * sg.out_nstride * 4));
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(erd::uniform_ptr(sg.out), 0, out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(erd::uniform_ptr(const_cast<float*>(RES ? sg.res : sg.out)), 0, out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_msk = __builtin_amdgcn_make_buffer_rsrc(erd::uniform_ptr(const_cast<float*>(MSK ? sg.mask : sg.out)), 0, out_bytes, 0x00020000);
        const bool has_alpha = sg.alpha != nullptr;
        const float alpha = has_alpha ? *sg.alpha : 1.f;      // (read once per tile: a load inside the block would order itself behind the block's stores)
        const long long tile_t0 = (long long)mt * nb;
        const int cb_begin = (int)(t_begin > tile_t0 ? t_begin - tile_t0 : 0), cb_end = (int)(t_end < tile_t0 + nb ? t_end - tile_t0 : nb);
#pragma unroll 1
        for (int cb = cb_begin; cb < cb_end; ++cb, ++it) {
        const long long t = tile_t0 + cb;
        const int buf = it & 1;
        THIN_T0(tr);
        store_unit(buf);                   // this block's planes: registers -> LDS (the buffer was last read two blocks ago)
        if (t + 1 < t_end) load_unit(cb + 1 == nb ? 0 : cb + 1);       // (tile-major order: the next unit is the next block, or block 0 of the next tile)
        __syncthreads();
        THIN_TACC(t_ring, tr);
        THIN_T0(tp);
        // ---- column sums of the PREVIOUS block: four waves' partial rows -> one atomic per channel ------------------------
        if (pend_cb >= 0 && tid < 32) {
            const float* rp = red + ((it - 1) & 1) * 128;
            atomicAdd(cs_row + pend_cb * 32 + tid, (rp[tid] + rp[32 + tid]) + (rp[64 + tid] + rp[96 + tid]));
        }
        // ---- residual / mask rows of this block are requested before its MFMAs ---------------------------------------------
        const int c4 = lane & 7, rsub = lane >> 3;
        const int co = cb * 32 + c4 * 4;
        unsigned off[RG][4];
        u4v pfr[RG][4], pfm[RG][4];
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = rows[(wave * RG + g) * 32 + q * 8 + rsub].out_off;
            off[g][q] = o < 0 ? OOB : (unsigned)(o + co) * 4u;
            if constexpr (RES) pfr[g][q] = buf_load16(rs_res, off[g][q]);
            if constexpr (MSK) pfm[g][q] = buf_load16(rs_msk, off[g][q]);
        }
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + co);
        if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + co);
        // ---- the block's products: 6 MFMAs per k16 step and row group, the stream-K kernel's order -----------------------------
        f32x16 acc[RG];
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        THIN_TACC(t_pre, tp);
        THIN_T0(tm);
        const char* Bb = ring + buf * UNIT_B;
        // (weight fragments are read ONE step ahead by hand and the schedule is pinned per step: left alone the compiler hoists
        //  every step's LDS reads to the top of the block -- 96 more live registers -- and spills the limb fragments)
        bf16x8 wf[2][3];
        auto read_w = [&](int s, int slot) {
            const int ch = 2 * s + h;
            const int sw = CPR == 16 ? (ch ^ (li & 15)) : (ch ^ ((li >> 1) & 7));
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wf[slot][pl] = *reinterpret_cast<const bf16x8*>(Bb + ((pl * 32 + li) * CPR + sw) * 16);
        };
        read_w(0, 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) read_w(s + 1, (s + 1) & 1);
            const bf16x8 w0 = wf[s & 1][0], w1 = wf[s & 1][1], w2 = wf[s & 1][2];
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, ah[g][s]), a1 = __builtin_bit_cast(bf16x8, am[g][s]), a2 = __builtin_bit_cast(bf16x8, al[g][s]);
#ifdef ERD_THIN_NOMFMA     // timing probe: everything but the matrix instructions (results are wrong)
                acc[g][0] += __builtin_bit_cast(float4, a0).x * __builtin_bit_cast(float4, w2).x + __builtin_bit_cast(float4, a1).x * __builtin_bit_cast(float4, w1).x +
                             __builtin_bit_cast(float4, a2).x * __builtin_bit_cast(float4, w0).x;
#else
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w2, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w1, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w0, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w1, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w0, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w0, acc[g], 0, 0, 0);
#endif
#ifdef ERD_X3_NINE        // accuracy probe, as in conv_igemm_kernel: the three dropped limb products
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w2, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w1, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w2, acc[g], 0, 0, 0);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        THIN_TACC(t_mma, tm);
        // ---- epilogue of the block, wave-private: accumulators -> LDS -> 128-byte row segments ------------------------------
        float* wst = stage + wave * 32 * SLD;
        float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            THIN_T0(tg);
#pragma unroll
            for (int r = 0; r < 16; ++r) wst[((r & 3) + 8 * (r >> 2) + 4 * h) * SLD + li] = acc[g][r];
            __builtin_amdgcn_wave_barrier();               // (LDS operations of one wave execute in order)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            THIN_TACC(t_stage, tg);
            THIN_T0(to);
            // (straight-line on purpose: gfx950 counts loads AND stores in vmcnt, in order.  With the stores under `if (row valid)`
            //  and the residual / mask / alpha loads under branches the compiler had to put `s_waitcnt vmcnt(0)` in front of every
            //  one of the four stores and in front of the next block's ring store -- each store waited for the previous one's
            //  acknowledgement, 2.6-2.9 k of a block's 7.4 k cycles in the phase trace.  Now: every load of the block before its
            //  first store, the stores back to back, and the next block's ring wait leaves them in flight.)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = *reinterpret_cast<const float4*>(wst + (q * 8 + rsub) * SLD + c4 * 4);
                v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                if (has_alpha) { v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha; }
                if constexpr (RES) { const float4 rv = __builtin_bit_cast(float4, pfr[g][q]); v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if constexpr (MSK) {
                    const float4 mv = __builtin_bit_cast(float4, pfm[g][q]);
                    v.x = mv.x > 0.f ? v.x : 0.f; v.y = mv.y > 0.f ? v.y : 0.f;
                    v.z = mv.z > 0.f ? v.z : 0.f; v.w = mv.w > 0.f ? v.w : 0.f;
                }
#ifdef ERD_THIN_NOSTORE    // timing probe: nothing is written (results are wrong)
                asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
#else
                buf_store16(rs_out, off[g][q], v);
#endif
                if (off[g][q] != OOB) { csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w; }
            }
            __builtin_amdgcn_wave_barrier();               // the staging block is re-used by the next accumulators
            THIN_TACC(t_out, to);
        }
        if (cs_row) {      // lanes that share a column group (lane bits 3..5), then the wave's row of `red`
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                csum.x += __shfl_xor(csum.x, o, 64); csum.y += __shfl_xor(csum.y, o, 64);
                csum.z += __shfl_xor(csum.z, o, 64); csum.w += __shfl_xor(csum.w, o, 64);
            }
            if (lane < 8) *reinterpret_cast<float4*>(red + (it & 1) * 128 + wave * 32 + lane * 4) = csum;
            pend_cb = cb;
        }
        }
    }
#ifdef ERD_THIN_TRACE
    if (tid == 0 && blockIdx.x < 1024) {
        unsigned long long* tr = g_thin_trace + blockIdx.x * 8;
        tr[0] = __builtin_amdgcn_s_memtime() - t_begin_clk; tr[1] = t_tile; tr[2] = t_ring; tr[3] = t_pre; tr[4] = t_mma; tr[5] = t_stage; tr[6] = t_out;
        tr[7] = (unsigned long long)it;
    }
#endif
    if (pend_cb >= 0) {
        __syncthreads();
        if (tid < 32) {
            const float* rp = red + ((it - 1) & 1) * 128;
            atomicAdd(cs_row + pend_cb * 32 + tid, (rp[tid] + rp[32 + tid]) + (rp[64 + tid] + rp[96 + tid]));
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------------------------
// The same loop nest for the bf16 mode (BASELINE configs[2]: bf16 multiplicands, exact products, fp32 accumulation; maps STORED as bf16;
// round 6).  There the stream-K kernel's thin launches are all set-up and epilogue (128 -> 512 on 100 x 168: 53 us
// of which 47 remain with its K loop's loads removed, EXPERIMENTS 7a) next to ~17 us of HBM traffic.  One limb: a lane's 16-byte load
// of eight stored bf16 channels IS its MFMA A-fragment (no conversion, K / 16 x 4 registers for the whole reduction); one weight plane
// (erd_conv_desc::w_bf16) through the LDS ring, ONE MFMA per k16 step in the stream-K kernel's order -- bit-identical results
// (tests/test_gpu_thin.py) --; the block's epilogue as above with 8-byte row pieces (four bf16 channels), residual / mask rows stored
// bf16 too.  36 KB of LDS, ~100 registers: four workgroups per CU.  With one limb the whole reduction of K = 256 fits as well (64
// fragment registers, 32 KB of ring: three workgroups per CU): layer3's expanding convolutions, the input gradients of its reducing
// ones, layer2's first block -- launches that are set-up and epilogue in the stream-K kernel's bf16 form too.
// KH = 2 (K = 512): a cout block's weights pass through the ring in two K-halves (16 KB each) -- the ring stays at 32 KB, the 128 fragment
// registers of the whole reduction leave room for two workgroups per CU.
template <int KS, bool RES, bool MSK, int KH = (KS > 16 ? 2 : 1)>
__global__ __launch_bounds__(256, (KS <= 8 ? 4 : (KS > 16 || (RES && MSK) ? 2 : 3))) void conv_thin_bf16_kernel(const erd_conv_desc p, const int mtiles, const int nb, const int xcd_order) {
    constexpr int K = KS * 16;
    constexpr int KU = K / KH, KSU = KS / KH;          // reduction length / k16 steps of one ring unit
    constexpr int CPR = KU / 8;                        // 16-byte chunks (8 bf16) per weight row of a unit
    constexpr int UNIT_B = 32 * KU * 2;                // one cout block (one K-half of it) of the one weight plane
    constexpr int NQ = (UNIT_B / 16 + 255) / 256;      // 16-byte loads per thread and block (2 at K = 128, 1 at K = 64)
    constexpr int NCH = UNIT_B / 16;                   // chunks of a block (512 / 256)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                                     // [2][UNIT_B]
    float* stage = reinterpret_cast<float*>(smem + 2 * UNIT_B);            // [4 waves][32][SLD]
    TRow* rows = reinterpret_cast<TRow*>(smem + 2 * UNIT_B + 4 * 32 * SLD * 4);   // [128]
    float* red = reinterpret_cast<float*>(rows + 128);                   // [2][4 waves][32]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int G = gridDim.x;
    int wg = blockIdx.x;
    if (xcd_order) {
        const int q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long long T = (long long)mtiles * nb;
    const long long t_begin = T * wg / G, t_end = T * (wg + 1) / G;
    if (t_begin >= t_end) return;

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_bf16), 0, (int)((long long)p.Cout * p.wrow * 2), 0x00020000);
    unsigned w_off[NQ];
    int l_off[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int idx = q * 256 + tid;
        const int row = idx / CPR, ch = idx - row * CPR;
        w_off[q] = idx < NCH ? (unsigned)(row * p.wrow + p.wk[0] + ch * 8) * 2u : OOB;
        const int sw = CPR >= 16 ? (ch ^ (row & 15)) : (ch ^ ((row >> 1) & 7));      // (rows of 256 B and more: the low four chunk bits)
        l_off[q] = idx < NCH ? (row * CPR + sw) * 16 : -1;
    }
    u4v rb[NQ];
    auto load_unit = [&](int cbn, int khn) {
        const unsigned cb_off = (unsigned)(cbn * 32 * p.wrow + khn * KU) * 2u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) rb[q] = buf_load16(rs_w, w_off[q] == OOB ? OOB : w_off[q] + cb_off);
    };
    auto store_unit = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (l_off[q] >= 0) *reinterpret_cast<u4v*>(ring + buf * UNIT_B + l_off[q]) = rb[q];
    };

    u4v ah[KS];                            // the wave's activation rows: the stored bf16 values ARE the fragments
    int pend_cb = -1;
    float* const cs_row = p.colsum ? p.colsum + (p.colsum_copies > 1 ? (int64_t)(blockIdx.x & (p.colsum_copies - 1)) * p.Cout : 0) : nullptr;

    load_unit((int)(t_begin % nb), 0);
    int it = 0, ib = 0;                    // ring steps / cout blocks done by this workgroup (parities of the LDS ring / of `red`)
#pragma unroll 1
    for (int mt = (int)(t_begin / nb); (long long)mt * nb < t_end; ++mt) {
        int seg_i = 0, mt_in_seg = mt;
#pragma unroll 1
        for (; seg_i < p.nseg - 1; ++seg_i) {
            const int M = p.seg[seg_i].N * p.seg[seg_i].GH * p.seg[seg_i].GW;
            const int tiles = (M + 127) / 128;
            if (mt_in_seg < tiles) break;
            mt_in_seg -= tiles;
        }
        const erd_conv_seg& sg = p.seg[seg_i];
        __syncthreads();               // the previous tile's epilogues are done with the row table
        if (tid < 128) {
            const int GHW = sg.GH * sg.GW, M = sg.N * GHW, m = mt_in_seg * 128 + tid;
            TRow ri;
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
        {
            const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
                erd::uniform_ptr(const_cast<float*>(sg.in)), 0, erd::uniform_int((int)((long long)sg.N * sg.in_nstride * 2)), 0x00020000);
            const int ibase = rows[wave * 32 + li].in_off;
#pragma unroll
            for (int s = 0; s < KS; ++s) ah[s] = buf_load16(rs_in, ibase < 0 ? OOB : (unsigned)(ibase + 16 * s + 8 * h) * 2u);
        }
        const int out_bytes = erd::uniform_int((int)((long long)sg.N * sg.out_nstride * 2));
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(erd::uniform_ptr(sg.out), 0, out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(erd::uniform_ptr(const_cast<float*>(RES ? sg.res : sg.out)), 0, out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_msk = __builtin_amdgcn_make_buffer_rsrc(erd::uniform_ptr(const_cast<float*>(MSK ? sg.mask : sg.out)), 0, out_bytes, 0x00020000);
        const bool has_alpha = sg.alpha != nullptr;
        const float alpha = has_alpha ? *sg.alpha : 1.f;
        const long long tile_t0 = (long long)mt * nb;
        const int cb_begin = (int)(t_begin > tile_t0 ? t_begin - tile_t0 : 0), cb_end = (int)(t_end < tile_t0 + nb ? t_end - tile_t0 : nb);
#pragma unroll 1
        for (int cb = cb_begin; cb < cb_end; ++cb, ++ib) {
            const long long t = tile_t0 + cb;
            store_unit(it & 1);
            if (KH > 1) load_unit(cb, 1); else if (t + 1 < t_end) load_unit(cb + 1 == nb ? 0 : cb + 1, 0);
            __syncthreads();
            if (pend_cb >= 0 && tid < 32) {
                const float* rp = red + ((ib - 1) & 1) * 128;
                atomicAdd(cs_row + pend_cb * 32 + tid, (rp[tid] + rp[32 + tid]) + (rp[64 + tid] + rp[96 + tid]));
            }
            const int c4 = lane & 7, rsub = lane >> 3;
            const int co = cb * 32 + c4 * 4;
            unsigned off[4];
            uint2 pfr[4], pfm[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = rows[wave * 32 + q * 8 + rsub].out_off;
                off[q] = o < 0 ? OOB : (unsigned)(o + co) * 2u;
                if constexpr (RES) pfr[q] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs_res, off[q], 0, 0));
                if constexpr (MSK) pfm[q] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs_msk, off[q], 0, 0));
            }
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + co);
            if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + co);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            bf16x8 wf[2];
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                if (kh > 0) {              // the block's second K-half: the ring's other buffer (its loads were requested a half ago)
                    store_unit(it & 1);
                    if (t + 1 < t_end) load_unit(cb + 1 == nb ? 0 : cb + 1, 0);
                    __syncthreads();
                }
                const char* Bb = ring + (it & 1) * UNIT_B;
                auto read_w = [&](int s, int slot) {
                    const int ch = 2 * s + h;
                    const int sw = CPR >= 16 ? (ch ^ (li & 15)) : (ch ^ ((li >> 1) & 7));
                    wf[slot] = *reinterpret_cast<const bf16x8*>(Bb + (li * CPR + sw) * 16);
                };
                read_w(0, 0);
#pragma unroll
                for (int s = 0; s < KSU; ++s) {
                    if (s + 1 < KSU) read_w(s + 1, (s + 1) & 1);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[kh * KSU + s]), wf[s & 1], acc, 0, 0, 0);
                }
                ++it;
            }
            float* wst = stage + wave * 32 * SLD;
            float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 16; ++r) wst[((r & 3) + 8 * (r >> 2) + 4 * h) * SLD + li] = acc[r];
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = *reinterpret_cast<const float4*>(wst + (q * 8 + rsub) * SLD + c4 * 4);
                v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                if (has_alpha) { v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha; }
                if constexpr (RES) { const float4 rv = erd::unpack4_bf16(pfr[q]); v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if constexpr (MSK) {
                    const float4 mv = erd::unpack4_bf16(pfm[q]);
                    v.x = mv.x > 0.f ? v.x : 0.f; v.y = mv.y > 0.f ? v.y : 0.f;
                    v.z = mv.z > 0.f ? v.z : 0.f; v.w = mv.w > 0.f ? v.w : 0.f;
                }
                typedef unsigned int u2v __attribute__((ext_vector_type(2)));
                const uint2 pk = erd::pack4_bf16(v);
                u2v o2; o2.x = pk.x; o2.y = pk.y;
                __builtin_amdgcn_raw_buffer_store_b64(o2, rs_out, off[q], 0, 0);
                if (off[q] != OOB) { csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w; }
            }
            __builtin_amdgcn_wave_barrier();
            if (cs_row) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
                    csum.x += __shfl_xor(csum.x, o, 64); csum.y += __shfl_xor(csum.y, o, 64);
                    csum.z += __shfl_xor(csum.z, o, 64); csum.w += __shfl_xor(csum.w, o, 64);
                }
                if (lane < 8) *reinterpret_cast<float4*>(red + (ib & 1) * 128 + wave * 32 + lane * 4) = csum;
                pend_cb = cb;
            }
        }
    }
    if (pend_cb >= 0) {
        __syncthreads();
        if (tid < 32) {
            const float* rp = red + ((ib - 1) & 1) * 128;
            atomicAdd(cs_row + pend_cb * 32 + tid, (rp[tid] + rp[32 + tid]) + (rp[64 + tid] + rp[96 + tid]));
        }
    }
}

int num_cus_thin() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int KS, int RG, bool RES, bool MSK>
int launch_thin(const erd_conv_desc* d, hipStream_t st) {
    constexpr int K = KS * 16, TR = 128 * RG;
    int mtiles = 0;
    for (int s = 0; s < d->nseg; ++s) mtiles += (int)(((int64_t)d->seg[s].N * d->seg[s].GH * d->seg[s].GW + TR - 1) / TR);
    const int nb = d->Cout / 32;
    if (mtiles == 0) return 0;
    const size_t lds = (size_t)2 * (3 * 32 * K * 2) + 4 * 32 * SLD * 4 + TR * sizeof(TRow) + 2 * 4 * 32 * 4;
    auto kern = conv_thin_x3_kernel<KS, RG, RES, MSK>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    static const int xcd = getenv("ERD_XCD") ? atoi(getenv("ERD_XCD")) : 1;
    const long long T = (long long)mtiles * nb;
    // two workgroups per CU at K = 128 (226 registers, 68 KB of LDS); three at K = 64 (154 registers, 45 KB).  ERD_THIN_WGS: A/B aid
    static const int wgs_env = getenv("ERD_THIN_WGS") ? atoi(getenv("ERD_THIN_WGS")) : 0;
    const int per_cu = wgs_env > 0 ? wgs_env : (KS == 4 && RG == 1 ? 3 : 2);
    const int G = (int)std::min<long long>(T, (long long)per_cu * erd::usable_cus(num_cus_thin()));
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), lds, st, *d, mtiles, nb, xcd ? 1 : 0);
    return erd::check_launch("conv_thin_x3");
}

template <int KS, bool RES, bool MSK>
int launch_thin_bf16(const erd_conv_desc* d, hipStream_t st) {
    constexpr int K = KS * 16;
    int mtiles = 0;
    for (int s = 0; s < d->nseg; ++s) mtiles += (int)(((int64_t)d->seg[s].N * d->seg[s].GH * d->seg[s].GW + 127) / 128);
    const int nb = d->Cout / 32;
    if (mtiles == 0) return 0;
    const size_t lds = (size_t)2 * (32 * (K > 256 ? K / 2 : K) * 2) + 4 * 32 * SLD * 4 + 128 * sizeof(TRow) + 2 * 4 * 32 * 4;
    static const int xcd = getenv("ERD_XCD") ? atoi(getenv("ERD_XCD")) : 1;
    const long long T = (long long)mtiles * nb;
    static const int wgs_env = getenv("ERD_THIN_BF16_WGS") ? atoi(getenv("ERD_THIN_BF16_WGS")) : 0;      // A/B aid
    const int per_cu = wgs_env > 0 ? wgs_env : (KS <= 8 ? 4 : (KS > 16 || (RES && MSK) ? 2 : 3));      // (K = 256 with residual AND mask rows: 168 registers would spill)
    const int G = (int)std::min<long long>(T, (long long)per_cu * erd::usable_cus(num_cus_thin()));
    static bool attr_done = false;      // (K = 256: 52 KB of dynamic LDS)
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_thin_bf16_kernel<KS, RES, MSK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_thin_bf16_kernel<KS, RES, MSK>), dim3(G), dim3(256), lds, st, *d, mtiles, nb, xcd ? 1 : 0);
    return erd::check_launch("conv_thin_bf16");
}

}  // namespace

namespace erd {

// does this launch fit the activation-stationary kernel?  (erd_conv_igemm asks before it picks a stream-K variant)
static int g_thin_on = -1;      // -1: not decided yet (ERD_THIN, default on)

int conv_thin_enable(int on) {
    if (g_thin_on < 0) g_thin_on = getenv("ERD_THIN") ? atoi(getenv("ERD_THIN")) : 1;
    const int prev = g_thin_on;
    if (on >= 0) g_thin_on = on ? 1 : 0;
    return prev;
}

bool conv_thin_x3_ok(const erd_conv_desc* d) {
    const int on = conv_thin_enable(-1);
    if (!on || !d->w_x3 || d->w_bf16 || d->in_bf16 || d->out_bf16 || d->ntaps != 1) return false;
    if (!(d->Cin == 64 || d->Cin == 128) || d->Cout % 32 != 0 || d->wrow % 8 != 0 || d->wk[0] % 8 != 0) return false;
    for (int s = 0; s < d->nseg; ++s) {
        const erd_conv_seg& g = d->seg[s];
        if (g.ntaps > 0) return false;
        // the kernel is instantiated per (residual, mask) presence and addresses both through the output's row offsets
        if ((g.res != nullptr) != (d->seg[0].res != nullptr) || (g.mask != nullptr) != (d->seg[0].mask != nullptr)) return false;
        if (g.res && g.res_nstride != g.out_nstride) return false;
        if ((long long)g.N * g.out_nstride * 4 >= 0x7fffffffLL || (long long)g.N * g.in_nstride * 4 >= 0x7fffffffLL) return false;
    }
    return true;
}

// the bf16 mode's thin launches: one tap, Cin in {64, 128, 256, 512}, Cout % 32 == 0, maps stored bf16 on both sides
bool conv_thin_bf16_ok(const erd_conv_desc* d) {
    const int on = conv_thin_enable(-1);
    if (!on || !d->w_bf16 || d->w_x3 || !d->in_bf16 || !d->out_bf16 || d->ntaps != 1) return false;
    static const int k256 = getenv("ERD_THIN_BF16_K256") ? atoi(getenv("ERD_THIN_BF16_K256")) : 1;      // A/B aids
    static const int k512 = getenv("ERD_THIN_BF16_K512") ? atoi(getenv("ERD_THIN_BF16_K512")) : 1;
    if (!(d->Cin == 64 || d->Cin == 128 || (d->Cin == 256 && k256) || (d->Cin == 512 && k512)) || d->Cout % 32 != 0 || d->wrow % 8 != 0 || d->wk[0] % 8 != 0) return false;
    for (int s = 0; s < d->nseg; ++s) {
        const erd_conv_seg& g = d->seg[s];
        if (g.ntaps > 0) return false;
        if ((g.res != nullptr) != (d->seg[0].res != nullptr) || (g.mask != nullptr) != (d->seg[0].mask != nullptr)) return false;
        if (g.res && g.res_nstride != g.out_nstride) return false;
        if ((long long)g.N * g.out_nstride * 2 >= 0x7fffffffLL || (long long)g.N * g.in_nstride * 2 >= 0x7fffffffLL) return false;
    }
    return true;
}

int conv_thin_bf16(const erd_conv_desc* d, hipStream_t st) {
    const bool r = d->seg[0].res != nullptr, m = d->seg[0].mask != nullptr;
    if (d->Cin == 64) return r ? (m ? launch_thin_bf16<4, true, true>(d, st) : launch_thin_bf16<4, true, false>(d, st))
                               : (m ? launch_thin_bf16<4, false, true>(d, st) : launch_thin_bf16<4, false, false>(d, st));
    if (d->Cin == 512) return r ? (m ? launch_thin_bf16<32, true, true>(d, st) : launch_thin_bf16<32, true, false>(d, st))
                                : (m ? launch_thin_bf16<32, false, true>(d, st) : launch_thin_bf16<32, false, false>(d, st));
    if (d->Cin == 256) return r ? (m ? launch_thin_bf16<16, true, true>(d, st) : launch_thin_bf16<16, true, false>(d, st))
                                : (m ? launch_thin_bf16<16, false, true>(d, st) : launch_thin_bf16<16, false, false>(d, st));
    return r ? (m ? launch_thin_bf16<8, true, true>(d, st) : launch_thin_bf16<8, true, false>(d, st))
             : (m ? launch_thin_bf16<8, false, true>(d, st) : launch_thin_bf16<8, false, false>(d, st));
}

int conv_thin_x3(const erd_conv_desc* d, hipStream_t st) {
    const bool r = d->seg[0].res != nullptr, m = d->seg[0].mask != nullptr;
    if (d->Cin == 64) return r ? (m ? launch_thin<4, 1, true, true>(d, st) : launch_thin<4, 1, true, false>(d, st))
                               : (m ? launch_thin<4, 1, false, true>(d, st) : launch_thin<4, 1, false, false>(d, st));
    return r ? (m ? launch_thin<8, 1, true, true>(d, st) : launch_thin<8, 1, true, false>(d, st))
             : (m ? launch_thin<8, 1, false, true>(d, st) : launch_thin<8, 1, false, false>(d, st));
}

}  // namespace erd

extern "C" int erd_conv_thin_enable(int on) { return erd::conv_thin_enable(on); }
extern "C" int erd_conv_thin_ok(const erd_conv_desc* d) { return d && (erd::conv_thin_x3_ok(d) || erd::conv_thin_bf16_ok(d)) ? 1 : 0; }
#ifdef ERD_THIN_TRACE
extern "C" int erd_thin_trace(unsigned long long* out) {       // trace builds only
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_thin_trace), sizeof(g_thin_trace));
}
#endif
