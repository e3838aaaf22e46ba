// Implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Data layout: activations NHWC fp32, weights [Cout][taps][Cin] (K contiguous for both GEMM
// operands).  One workgroup = 256 threads = 4 waves computes a BM x BN tile of
// out[pixel][cout] = sum_{tap,ci} in[pixel shifted by tap][ci] * w[cout][tap][ci].
//
//  * A (pixels) and B (couts) K-slices of 32 floats (=128 B rows) are staged global->VGPR->LDS
//    with a 16-B-chunk XOR swizzle (chunk ^= (row>>1)&7) so that the ds_read_b128 fragment reads
//    (16-lane groups of distinct rows, same chunk) and the ds_write_b128 stores (8 lanes = one row)
//    are bank-conflict free; double-buffered, one barrier per K-slice, next slice's global loads
//    issued before the MFMAs of the current one.
//  * each lane reads 4 consecutive k of its row with one ds_read_b128 and feeds them to 4 MFMAs:
//    the k-pairing (k, k+4) differs from memory order but is the same for A and B, which is all a
//    dot product needs.
//  * the epilogue applies folded-BN scale/shift | bias, a per-level scalar, a residual and ReLU
//    while the accumulators are still in registers, and writes 128-B rows (32 couts of one pixel).
//
// Forward conv, stride-1 input-gradient and the 4 parity classes of a stride-2 input-gradient are
// all expressed through the tap list of erd_conv_desc (include/erd_hip.h).
#include <algorithm>
#include <type_traits>
#include "erd_common.h"
#include <stdlib.h>
#ifndef ERD_SGB
#define ERD_SGB 0
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// two fp32 -> one dword of two bf16 (round to nearest even: v_cvt_pk_bf16_f32)
__device__ __forceinline__ float pack_bf16(float a, float b) {
    const f32x2v v = {a, b};
    return __builtin_bit_cast(float, __builtin_convertvector(v, bf16x2v));
}

constexpr int NTHREADS = 256;

#ifdef ERD_IGEMM_TRACE      // debug builds only: per-workgroup phase cycles (tools/dbg/igemm_trace.py)
__device__ unsigned long long g_igemm_trace[1024 * 8];
#define IG_T0(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define IG_ACC(slot, v) if (threadIdx.x == 0 && blockIdx.x < 1024) g_igemm_trace[blockIdx.x * 8 + slot] += __builtin_amdgcn_s_memtime() - v
#define IG_SET(slot, val) if (threadIdx.x == 0 && blockIdx.x < 1024) g_igemm_trace[blockIdx.x * 8 + slot] = (val)
#else
#define IG_T0(v)
#define IG_ACC(slot, v)
#define IG_SET(slot, val)
#endif

struct RowInfo {
    int in_off;   // element offset of image n in `in`
    int ih0, iw0; // a*in_stride, b*in_stride
    int out_off;  // element offset of the output pixel (-1: row past the end)
};

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// Work decomposition ("stream-K"): the launch is a PERSISTENT grid of G workgroups.  All output tiles have
// the same number of K-slices nkt, so the work is U = tiles*nkt equal units, and workgroup b takes the
// contiguous unit range [b*U/G, (b+1)*U/G): whole tiles in the middle, at most one partial tile at each end.
// A partial tile's accumulators go to a per-(workgroup, slot) fp32 slab; the LAST contributor to arrive (a
// relaxed agent-scope ticket behind an agent-scope release; acquire on the reducer -- cdna_hip_programming
// Guideline 16 counter form, no spinning, no co-residency assumption) sums all slabs of the tile in
// contributor order (deterministic) and runs the epilogue.  With G == tiles this is plain data-parallel.
// Why: at bs=4 most ERD layers have 0.5..4 "CU-rounds" of 128x128 tiles; tile-granular dispatch wastes
// 10-48 % of the matrix pipes in the last round, unit-granular dispatch wastes < 1/nkt.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x7fffffffu;   // past num_records of any buffer we build: the load returns zeros

__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// LDS-DMA (`buffer_load_dwordx4 ... lds`): 16 B per lane from the lane's own buffer offset to LDS address M0 + 16 * lane, no register in
// between; out-of-range offsets arrive as zeros (tools/glds_probe.hip).  Issued through inline assembly ON PURPOSE: the compiler's own
// builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) makes it wait `vmcnt(0)` in front of the next LDS read that might alias -- in the
// middle of the K-slice that is supposed to hide the transfer.  An asm load is absent from the compiler's s_waitcnt bookkeeping, which
// is safe in one direction only: loads retire in order, so the compiler's counted waits for ITS loads can only become stricter; the
// DMA's own completion is waited for explicitly (`glds_wait_all`) in front of the barrier that publishes the buffer.  M0 is written
// in the same statement (the compiler keeps nothing in M0 across statements); `s_nop 0`: the M0-write -> LDS-DMA hazard.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 rsrc_words(const void* ptr, int bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void glds16(const i32x4 rs, unsigned lds_byte, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_byte), "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void glds_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct SkWs {
    int* cnt;        // [tiles] arrival tickets (zeroed before the launch)
    float* slabs;    // [2*G][BM*BN]
    int xcd_order;   // 1: XCD-aware work order (default); 0: dispatch order (A/B aid, ERD_XCD=0)
    int whole_tiles; // 1: persistent grid over WHOLE tiles (workgroup b owns tiles [b T / G, (b+1) T / G): no partial tiles, no fix-up)
};

int xcd_order_enabled() {
    static const int v = getenv("ERD_XCD") ? atoi(getenv("ERD_XCD")) : 1;
    return v;
}

// BF = true: the same kernel on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16): a 16-B LDS chunk holds 8 bf16
// k-values instead of 4 floats (a K-slice is CH*8 values), activations are read as fp32 (two 16-B loads per chunk)
// and rounded to bf16 on their way into LDS, weights come pre-rounded (erd_conv_desc::w_bf16); accumulation, the
// stream-K hand-over and the epilogue stay fp32.
// AB / OB (bf16 mode only): the input maps / the output, residual and mask maps are STORED as bf16 (erd_conv_desc::
// in_bf16 / out_bf16): a 16-B load then carries a whole 8-value LDS chunk and needs no conversion, and the epilogue
// moves 8-B groups of four channels.  Accumulation, scale/shift, the stream-K slabs and the column sums stay fp32.
// X3 = true ("f32x3", erd_conv_desc::w_x3): fp32 maps, fp32 accumulation, fp32 results -- but the PRODUCTS run on the bf16 matrix
// cores, which on gfx950 are 16x faster than the fp32 ones (v_mfma_f32_32x32x2_f32 runs at the vector rate).  Every fp32
// value is the exact sum of three bf16 limbs (8 + 8 + 8 significand bits: hi = rne(x), mid = rne(x - hi),
// lo = x - hi - mid; erd_common.h); a product a*b is then the sum of nine exact limb products, of which the six with weight >= 2^-16 are
// accumulated (a_hi b_hi, a_hi b_mid, a_mid b_hi, a_hi b_lo, a_mid b_mid, a_lo b_hi: what is dropped is below 2^-23 of
// |a*b|, the size of fp32's own rounding of that product).  Weights arrive pre-split (three bf16 planes); activations stay
// fp32 in HBM and in LDS and are split in registers when a wave reads its fragments.  The waves form a 4 x 1 grid (each
// owns 32 pixel rows x all BN couts) so that every activation value is split by exactly one wave: 44 VALU operations per 24
// MFMAs.  One K-slice = 32 channels = two k16 steps.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BKT, int MINW, bool BF = false, bool ST = false, bool AB = false,
          bool OB = false, bool X3 = false, bool RES = false, bool MSK = false,     // RES / MSK (f32x3 only): some segment has a residual / a mask
          bool GL = false>     // GL (f32x3): operand slices travel global -> LDS by LDS-DMA (buffer_load ... lds), see below
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, MINW) void conv_igemm_kernel(const erd_conv_desc p, const int total_tiles,
                                                                     const SkWs ws) {
    constexpr int NT = WAVES_M * WAVES_N * 64;   // threads: four waves, or eight on the 256-row tiles of the bf16 mode
    constexpr int FM = BM / (WAVES_M * 32);
    constexpr int FN = BN / (WAVES_N * 32);
    constexpr int CH = BKT / 4;              // 16-B chunks per K-slice row (8 for BK=32, 4 for BK=16)
    constexpr int RPP = NT / CH;       // rows staged per pass
    constexpr int AJ = BM / RPP;             // float4 loads per thread for A
    constexpr int BJ = BN / RPP;
    constexpr int KPC = BF ? 8 : 4;          // k-values per 16-B chunk
    constexpr int BK = CH * KPC;             // k-values per K-slice
    constexpr bool SGB = ERD_SGB;
    static_assert(WAVES_M * WAVES_N == 4 || (WAVES_M * WAVES_N == 8 && BF && !X3), "4 waves (8: bf16 matrix cores only)");
    static_assert(BF || (!AB && !OB), "bf16 storage only with the bf16 matrix cores");
    static_assert(!X3 || (!BF && BKT == 32 && WAVES_N == 1 && FM == 1), "f32x3: fp32 maps, 32-channel slices, 4 x 1 waves");
    static_assert(!GL || X3, "LDS-DMA staging: the three-limb kernel (both operands are stored in LDS exactly as they lie in memory)");
    constexpr unsigned ABYTES = AB ? 2u : 4u;   // bytes per stored input value
    constexpr int CHB = X3 ? 4 : CH;            // 16-B chunks per K-slice row of ONE weight plane (f32x3: 8 bf16 per chunk)
    constexpr int NPL = X3 ? 3 : 1;             // weight planes in LDS
    constexpr int RPPB = NT / CHB;        // weight rows staged per pass
    constexpr int BJX = BN / RPPB;              // 16-B loads per thread and plane for B (f32x3)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* As = reinterpret_cast<float4*>(smem);                 // [2][BM*CH]
    float4* Bs = As + 2 * BM * CH;                                 // [2][BN*CH]   (f32x3: [2][3 planes][BN*4])
    constexpr int OPER_BYTES = 2 * (BM * CH + NPL * BN * CHB) * 16;
    constexpr int STAGE_BYTES = 64 * (BN + 4) * 4 + NT * 16;  // epilogue staging (+ column-sum scratch) re-uses the operand region
    constexpr int REGION = OPER_BYTES > STAGE_BYTES ? OPER_BYTES : STAGE_BYTES;
    // f32x3: the operand region is 80 KB at 128 x 128 -- exactly half of the CU's LDS, two workgroups per CU only if nothing
    // else is allocated.  The row table therefore lives INSIDE the region: at its start while a tile is set up (before the
    // first operand slice is stored) and behind the epilogue's staging area once the K loop is over (recomputed: 128 threads,
    // two divisions each); the broadcast word of the stream-K fix-up sits next to it.
    // (f32x3: the epilogue stages each wave's 32 rows in its own block: 4 x 32 x (BN + 4) floats)
    constexpr int WSTAGE_BYTES = 32 * (BN + 4) * 4;
    constexpr int ROWS_EPI_OFF = X3 ? ((4 * WSTAGE_BYTES + 255) / 256 * 256) : REGION;
    static_assert(!X3 || ROWS_EPI_OFF + BM * (int)sizeof(RowInfo) + 64 + 4 * (BN / 4) * 16 <= OPER_BYTES,
                  "f32x3: row table + column-sum scratch do not fit behind the staging blocks");
    RowInfo* rows = reinterpret_cast<RowInfo*>(smem + (X3 ? 0 : REGION));     // [BM]
    RowInfo* rows_epi = reinterpret_cast<RowInfo*>(smem + ROWS_EPI_OFF);
    int* bcast = reinterpret_cast<int*>(reinterpret_cast<RowInfo*>(smem + ROWS_EPI_OFF) + BM);                // [4]

    const int tid = threadIdx.x;
    const int ntn = (p.Cout + BN - 1) / BN;
    const int Cin = p.Cin;
    const int cpt = (Cin + BK - 1) / BK;  // K-slices per tap (last one zero-filled past Cin)
    const int nkt = p.ntaps * cpt;
    const float* __restrict__ w = p.w;

    const long long U = (long long)total_tiles * nkt;
    const int G = gridDim.x;
    // XCD-aware order: workgroup b runs on XCD b % 8 (each XCD has its own L2).  Give every XCD one CONTIGUOUS
    // chunk of the work list, so that the N-tiles of one pixel tile and spatially adjacent pixel tiles (3x3 halo)
    // hit the same L2 instead of being fetched through the fabric by up to 8 of them.  Bijective for any G.
    int wg = blockIdx.x;
    if (ws.xcd_order) {
        const int q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long long u_begin = ws.whole_tiles ? ((long long)total_tiles * wg / G) * nkt : (U * wg) / G;
    const long long u_end = ws.whole_tiles ? ((long long)total_tiles * (wg + 1) / G) * nkt : (U * (wg + 1)) / G;

    const int chunk = tid % CH;
    const int r0 = tid / CH;
    auto swzc = [](int row, int c) { return CH == 8 ? (c ^ ((row >> 1) & 7)) : (c ^ ((row >> 2) & 3)); };
    const int wave = tid >> 6, lane = tid & 63;
    // GL: an LDS-DMA instruction writes base + 16 * lane -- the LDS image is lane-linear, so the swizzle goes on the SOURCE side: the
    // thread that fills position `chunk` of row r fetches the logical chunk that the swizzle keeps there (an involution; the term of
    // the row, (r >> 1) & 7 resp. (r >> 2) & 3, is the same for all of a thread's row passes: they are 32 resp. 64 rows apart)
    const int chunk_l = GL ? (chunk ^ ((r0 >> 1) & 7)) : chunk;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, h = lane >> 5;

    IG_T0(t_kernel);
    IG_SET(0, t_kernel);
    for (int q_ = 1; q_ < 8; ++q_) { IG_SET(q_, 0ull); }
    for (long long u = u_begin; u < u_end;) {
        IG_T0(t_pro);
        const int tt = (int)(u / nkt);
        const int ks = (int)(u - (long long)tt * nkt);
        int ke = (int)min((long long)nkt, ks + (u_end - u));
        u += ke - ks;
        const int nt = tt % ntn;
        int mt = tt / ntn;

        // ---- which segment (level) does this M-tile belong to -------------------------------------
        int s = 0;
#pragma unroll 1
        for (; s < p.nseg - 1; ++s) {
            const int M = p.seg[s].N * p.seg[s].GH * p.seg[s].GW;
            const int tiles = (M + BM - 1) / BM;
            if (mt < tiles) break;
            mt -= tiles;
        }
        const erd_conv_seg& sg = p.seg[s];
        const int IH = sg.IH, IW = sg.IW;
        const float* __restrict__ in = sg.in;
        // per-segment tap set (parity classes of a stride-2 input gradient in one launch): fewer K-slices for this tile
        // (ST: only the instantiation that serves such launches carries the extra scalars)
        const int tap_lo = (ST && sg.ntaps) ? sg.tap0 : 0, nt_s = (ST && sg.ntaps) ? sg.ntaps : p.ntaps;
        const int oy_s = (ST && sg.ntaps) ? sg.oy : p.oy, ox_s = (ST && sg.ntaps) ? sg.ox : p.ox;
        const int nkt_t = ST ? nt_s * cpt : nkt;
        if (ST) {
            ke = min(ke, nkt_t);
            if (ks >= ke) continue;
        }

        __syncthreads();   // previous tile's epilogue / fragment reads are done with LDS
        auto fill_rows = [&](RowInfo* dst) {
            if (tid < BM) {
                const int GHW = sg.GH * sg.GW;
                const int M = sg.N * GHW;
                const int m = mt * BM + tid;
                RowInfo ri;
                if (m < M) {
                    const int n = m / GHW;
                    const int rem = m - n * GHW;
                    const int a = rem / sg.GW;
                    const int b = rem - a * sg.GW;
                    ri.in_off = (int)(n * sg.in_nstride);
                    ri.ih0 = a * p.in_stride;
                    ri.iw0 = b * p.in_stride;
                    ri.out_off = (int)(n * sg.out_nstride) +
                                 ((a * p.out_stride + oy_s) * sg.OW + (b * p.out_stride + ox_s)) * p.Cout;
                } else {
                    ri.in_off = 0;
                    ri.ih0 = -(1 << 28);
                    ri.iw0 = -(1 << 28);
                    ri.out_off = -1;
                }
                dst[tid] = ri;
            }
        };
        if constexpr (!GL) {
            fill_rows(rows);
            __syncthreads();
        }

        // per-row byte offset of tap (0,0) and a validity bit per tap: the K loop then needs one add + one select
        // per 16-B load, and zero padding comes from the buffer's out-of-range rule (no branches, no zero fill)
        unsigned a_base[AJ], a_mask[AJ];
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            RowInfo ri;
            if constexpr (GL) {      // no table, no barriers: a thread decodes its own AJ rows (the operand region is about to be filled by DMA)
                const int GHW = sg.GH * sg.GW;
                const int m = mt * BM + r0 + RPP * j;
                if (m < sg.N * GHW) {
                    const int n = m / GHW;
                    const int rem = m - n * GHW;
                    const int a = rem / sg.GW;
                    ri.in_off = (int)(n * sg.in_nstride);
                    ri.ih0 = a * p.in_stride;
                    ri.iw0 = (rem - a * sg.GW) * p.in_stride;
                } else {
                    ri.in_off = 0;
                    ri.ih0 = -(1 << 28);
                    ri.iw0 = -(1 << 28);
                }
            } else {
                ri = rows[r0 + RPP * j];
            }
            a_base[j] = (unsigned)(ri.in_off + (ri.ih0 * IW + ri.iw0) * Cin + chunk_l * KPC) * ABYTES;
            unsigned m = 0;
            for (int t = 0; t < nt_s; ++t) {
                const int ih = ri.ih0 + p.dy[tap_lo + t], iw = ri.iw0 + p.dx[tap_lo + t];
                m |= ((unsigned)ih < (unsigned)IH && (unsigned)iw < (unsigned)IW) ? (1u << t) : 0u;
            }
            a_mask[j] = m;
        }
        const int n0 = nt * BN;
        unsigned b_base[X3 ? 1 : BJ];
        unsigned bx_base[X3 ? BJX : 1];            // f32x3: byte offset of (cout row, 8-value chunk) inside ONE bf16 weight plane
        const int chunkb = tid % CHB, rb0 = tid / CHB;
        const int chunkb_l = GL ? (chunkb ^ ((rb0 >> 2) & 3)) : chunkb;      // (GL: the logical chunk behind this thread's LDS position)
        if constexpr (X3) {
#pragma unroll
            for (int j = 0; j < BJX; ++j) {
                const int co = n0 + rb0 + RPPB * j;
                bx_base[j] = co < p.Cout ? (unsigned)(co * p.wrow + chunkb_l * 8) * 2u : OOB;
            }
        } else {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int co = n0 + r0 + RPP * j;
                b_base[j] = co < p.Cout ? (unsigned)(co * p.wrow + chunk * KPC) * (BF ? 2u : 4u) : OOB;
            }
        }
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(in), 0, (int)((long long)sg.N * sg.in_nstride * ABYTES), 0x00020000);
        const unsigned plane_b = (unsigned)((long long)p.Cout * p.wrow * 2);      // f32x3: bytes of one weight plane
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
            X3 ? const_cast<void*>(p.w_x3) : (BF ? const_cast<void*>(p.w_bf16) : (void*)const_cast<float*>(w)), 0,
            (int)((long long)p.Cout * p.wrow * (X3 ? 6 : (BF ? 2 : 4))), 0x00020000);
        if constexpr (X3 && !GL) __syncthreads();         // the row table (start of the operand region) has been read by everybody

        // bf16 matrix cores: a K-slice is ~250 ns of MFMA work, far less than the latency of the global loads that
        // feed the next one -> TWO slices are kept in flight in registers (set = slice parity): slice kt+2 is requested
        // while slice kt is multiplied and slice kt+1 waits in its registers for the LDS buffer that slice kt-1 has left.
        constexpr bool PD2 = BF && MINW <= 2;      // (the four-per-CU tuning variant has no registers to spare)
        constexpr int NSET = PD2 ? 2 : 1;
        constexpr int NBL = X3 ? 3 * BJX : BJ;      // 16-B weight loads per thread and K-slice
        float4 ra[GL ? 1 : NSET][GL ? 1 : AJ], rb[GL ? 1 : NSET][GL ? 1 : NBL];      // (GL: no staging registers)
        float4 ra1[NSET][(BF && !AB) ? AJ : 1];     // bf16 mode on fp32 maps: the second half (k+4..k+7) of each 8-value chunk
        bool cokb = false;                          // f32x3: this thread's 8-value weight chunk lies inside Cin
        int tap = ks / cpt, cc = ks - tap * cpt;
        // wave-uniform description of the K-slice being fetched
        int adelta = 0, bdelta = 0, ctap = 0;
        bool cok = false, cok1 = false;
        auto slice_begin = [&]() {
            const int cb = cc * BK;
            ctap = tap;
            adelta = ((p.dy[tap_lo + tap] * IW + p.dx[tap_lo + tap]) * Cin + cb) * (int)ABYTES;    // bytes, relative to tap (0,0)
            bdelta = (p.wk[tap_lo + tap] + cb) * ((BF || X3) ? 2 : 4);
            cok = cb + chunk_l * KPC < Cin;   // Cin % 4 == 0: a 4-value group is all-in or all-out
            cokb = X3 && cb + chunkb_l * 8 < Cin;      // (a chunk that starts inside Cin may end past it: see erd_conv_igemm)
            cok1 = BF && !AB && cb + chunk * KPC + 4 < Cin;
            if (++cc == cpt) { cc = 0; ++tap; }
        };
        // GL: LDS-DMA.  `buffer_load_dwordx4 ... lds` moves 16 B per lane from the lane's own buffer offset to LDS address M0 + 16 * lane --
        // a wave fills 1 KB: 8 activation rows of 128 B, or 16 rows of one 64-B weight plane -- without passing through registers:
        // no staging registers (40 of them), no ds_write_b128 pass (13 LDS-port cycles each: 10 per thread and slice), and out-of-range
        // offsets (padding taps, rows past the end, channels past Cin) arrive as ZEROS exactly as with loads to registers
        // (tools/glds_probe.hip: source permutation, zero fill, destinations up to 160 KB).  Same LDS image, same MFMA sequence:
        // bit-identical results.
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;      // LDS byte address of the operand region
        const i32x4 rw_in = GL ? rsrc_words(in, (int)((long long)sg.N * sg.in_nstride * ABYTES)) : i32x4{0, 0, 0, 0};
        const i32x4 rw_w = GL ? rsrc_words(p.w_x3, (int)((long long)p.Cout * p.wrow * 6)) : i32x4{0, 0, 0, 0};
        auto glds_a = [&](int j, const int buf) {
            const bool ok = cok && ((a_mask[j] >> ctap) & 1u);
            glds16(rw_in, lds0 + (unsigned)(((buf * BM + RPP * j + 8 * wave_s) * CH) * 16), ok ? a_base[j] + (unsigned)adelta : OOB);
        };
        auto glds_b = [&](int j, const int buf) {      // j = plane * BJX + row pass
            const int pl = j / BJX, jj = j - pl * BJX;
            glds16(rw_w, lds0 + (unsigned)((2 * BM * CH + (buf * 3 + pl) * BN * CHB + (RPPB * jj + 16 * wave_s) * CHB) * 16),
                   cokb ? bx_base[X3 ? jj : 0] + (unsigned)bdelta + (unsigned)pl * plane_b : OOB);
        };
        auto load_a = [&](int j, const int set) {
            const bool ok = cok && ((a_mask[j] >> ctap) & 1u);
            ra[set][j] = buf_load16(rs_in, ok ? a_base[j] + (unsigned)adelta : OOB);
            if (BF && !AB) ra1[set][j] = buf_load16(rs_in, (ok && cok1) ? a_base[j] + (unsigned)adelta + 16u : OOB);
        };
        auto load_b = [&](int j, const int set) {
            if constexpr (X3) {                     // j = plane * BJX + row pass
                const int pl = j / BJX, jj = j - pl * BJX;
                // (a row past Cout carries OOB = 0x7fffffff: adding the slice / plane offsets keeps it beyond the buffer's extent)
                rb[set][j] = buf_load16(rs_w, cokb ? bx_base[jj] + (unsigned)bdelta + (unsigned)pl * plane_b : OOB);
            } else {
                rb[set][j] = buf_load16(rs_w, cok ? b_base[X3 ? 0 : j] + (unsigned)bdelta : OOB);
            }
        };
        auto store_lds = [&](int buf, const int set) {
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const int row = r0 + RPP * j;
                if (BF && !AB)      // (weights past Cin inside the chunk meet zeros here, so partial chunks are exact)
                    As[buf * BM * CH + row * CH + swzc(row, chunk)] =
                        make_float4(pack_bf16(ra[set][j].x, ra[set][j].y), pack_bf16(ra[set][j].z, ra[set][j].w),
                                    pack_bf16(ra1[set][(BF && !AB) ? j : 0].x, ra1[set][(BF && !AB) ? j : 0].y),
                                    pack_bf16(ra1[set][(BF && !AB) ? j : 0].z, ra1[set][(BF && !AB) ? j : 0].w));
                else
                    As[buf * BM * CH + row * CH + swzc(row, chunk)] = ra[set][j];
            }
            if constexpr (X3) {
#pragma unroll
                for (int j = 0; j < 3 * BJX; ++j) {
                    const int pl = j / BJX, row = rb0 + RPPB * (j - pl * BJX);
                    Bs[(buf * 3 + pl) * BN * CHB + row * CHB + (chunkb ^ ((row >> 2) & 3))] = rb[set][j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < BJ; ++j) {
                    const int row = r0 + RPP * j;
                    Bs[buf * BN * CH + row * CH + swzc(row, chunk)] = rb[set][X3 ? 0 : j];
                }
            }
        };

        f32x16 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        slice_begin();
        if constexpr (GL) {
#pragma unroll
            for (int j = 0; j < NBL; ++j) glds_b(j, 0);
#pragma unroll
            for (int j = 0; j < AJ; ++j) glds_a(j, 0);
        } else {
#pragma unroll
        for (int j = 0; j < AJ; ++j) load_a(j, 0);
#pragma unroll
        for (int j = 0; j < NBL; ++j) load_b(j, 0);
        }
        if (PD2 && ks + 1 < ke) {
            slice_begin();
#pragma unroll
            for (int j = 0; j < AJ; ++j) load_a(j, NSET - 1);
#pragma unroll
            for (int j = 0; j < NBL; ++j) load_b(j, NSET - 1);
        }
        if constexpr (GL) glds_wait_all(); else store_lds(0, 0);
        __syncthreads();

        IG_ACC(2, t_pro);
        IG_T0(t_loop);
        constexpr int KSTEPS = X3 ? 2 : CH / 2;      // one k-step = the two chunks (h = 0 / 1) a wave's lanes read (f32x3: 16 channels)
        constexpr int APS = (AJ + KSTEPS - 1) / KSTEPS, BPS = (NBL + KSTEPS - 1) / KSTEPS;   // loads per k-step
        // f32x3: one K-slice = two k16 steps.  A lane owns pixel row li of its wave's 32 rows and the 8 channels 16 s + 8 h ..
        // + 7 of step s: two 16-B reads of fp32 values, split into three bf16x8 limbs (round-to-nearest: cvt_pk / sub, exact), and
        // three 16-B reads per cout block of the pre-split weights; then six MFMAs per block, smallest terms first.
        auto k_slice_x3 = [&](const int kt, const int par) {
            const int buf = par;
            const bool more = kt + 1 < ke;
            if (more) slice_begin();
            const float4* Ab = As + buf * BM * CH;
            const float4* Bb = Bs + buf * 3 * BN * CHB;
            const int arow = wm * 32 + li;
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            // limb of eight values: rounded to nearest even and packed pairwise (v_cvt_pk_bf16_f32); `rest` = v - limb, exact
            // (erd::limbs3_pair spelled out so that the remainders can be formed under the upper limb's MFMAs)
            auto limb = [&](const float (&v)[8], u4v& P) {
#pragma unroll
                for (int e = 0; e < 4; ++e) P[e] = erd::pack2_bf16(v[2 * e], v[2 * e + 1]);
                return __builtin_bit_cast(bf16x8, P);
            };
            auto rest = [&](const float (&v)[8], const u4v& P, float (&r)[8]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r[2 * e] = erd::scalar_op(v[2 * e] - erd::bf16_lo(P[e]));
                    r[2 * e + 1] = erd::scalar_op(v[2 * e + 1] - erd::bf16_hi(P[e]));
                }
            };
            auto read_a = [&](int kk, float (&x)[8]) {
                const int ca = 4 * kk + 2 * h;                 // fp32 chunks ca, ca + 1 of the pixel row: channels 16 kk + 8 h .. + 7
                const float4 x0 = Ab[arow * CH + swzc(arow, ca)], x1 = Ab[arow * CH + swzc(arow, ca + 1)];
                x[0] = x0.x; x[1] = x0.y; x[2] = x0.z; x[3] = x0.w; x[4] = x1.x; x[5] = x1.y; x[6] = x1.z; x[7] = x1.w;
            };
            float4 wb[FN][3];
            auto read_b = [&](int kk, const int pl) {
                const int cbk = 2 * kk + h;                    // bf16 chunk of the weight rows
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int row = j * 32 + li;
                    wb[j][pl] = Bb[pl * BN * CHB + row * CHB + (cbk ^ ((row >> 2) & 3))];
                }
            };
#ifdef ERD_X3_NOMFMA      // timing probe: everything but the matrix instructions (results are garbage)
#define ERD_X3(AV, PL)                                                                                              \
            _Pragma("unroll") for (int j = 0; j < FN; ++j) acc[0][j][PL] += wb[j][PL].x * __builtin_bit_cast(float4, AV).x;
#else
#define ERD_X3(AV, PL)                                                                                              \
            _Pragma("unroll") for (int j = 0; j < FN; ++j)                                                          \
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AV, __builtin_bit_cast(bf16x8, wb[j][PL]), acc[0][j], 0, 0, 0);
#endif
            float x[8], xn[8];
            read_a(0, x);
            read_b(0, 2);
            read_b(0, 1);
            read_b(0, 0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // Order of the six limb products: the three that need only the UPPER limb of the activation first (four
                // v_perm and the fragments are ready), the remainders r1 = x - hi, r2 = r1 - mid are computed under their
                // MFMAs.  A weight plane's fragments are re-read for the next k16 step as soon as its last product is issued.
                if (kk == 0) read_a(1, xn);
#ifndef ERD_X3_NOLOAD     // (timing probe: the K loop re-uses the first slice's registers)
                if (more) {      // the next slice's global loads, half of them per k16 step, issued under this step's MFMAs
#pragma unroll
                    for (int q = 0; q < APS; ++q)
                        if (kk * APS + q < AJ) { if constexpr (GL) glds_a(kk * APS + q, buf ^ 1); else load_a(kk * APS + q, 0); }
#pragma unroll
                    for (int q = 0; q < BPS; ++q)
                        if (kk * BPS + q < NBL) { if constexpr (GL) glds_b(kk * BPS + q, buf ^ 1); else load_b(kk * BPS + q, 0); }
                }
#endif
                u4v ph, pm, pl;
                const bf16x8 ah = limb(x, ph);
                ERD_X3(ah, 2)
#ifndef ERD_X3_NOBREAD    // (timing probe: the second k16 step re-uses the first one's weight fragments)
                if (kk == 0) read_b(1, 2);
#endif
                ERD_X3(ah, 1)
                ERD_X3(ah, 0)
                float r1[8], r2[8];
#ifdef ERD_X3_NOVALU      // timing probe: the upper limb stands in for the other two (results are wrong)
                const bf16x8 am = ah, al = ah;
                (void)r1; (void)r2;
#else
                rest(x, ph, r1);
                const bf16x8 am = limb(r1, pm);
#endif
                ERD_X3(am, 1)
#ifndef ERD_X3_NOBREAD
                if (kk == 0) read_b(1, 1);
#endif
                ERD_X3(am, 0)
#ifndef ERD_X3_NOVALU
                rest(r1, pm, r2);
                const bf16x8 al = limb(r2, pl);
#endif
                ERD_X3(al, 0)
#ifdef ERD_X3_NINE        // accuracy probe (tools/build_probe.sh): the three limb products the production form drops (weight 2^-24 and below)
                ERD_X3(am, 2)
                ERD_X3(al, 1)
                ERD_X3(al, 2)
#endif
                if (kk == 0) {
#ifndef ERD_X3_NOBREAD
                    read_b(1, 0);
#endif
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = xn[e];
                }
            }
#undef ERD_X3
#ifndef ERD_X3_NOSYNC     // (timing probe, with ERD_X3_NOLOAD: no LDS refill and no barrier between slices)
            if constexpr (GL) glds_wait_all(); else { if (more) store_lds(buf ^ 1, 0); }
            __syncthreads();      // (GL: everybody's DMA into the other buffer has landed, and this buffer is free)
#endif
        };
        // one K-slice; par = (kt - ks) & 1 selects the LDS buffer (and, with two slices in flight, the register set)
        auto k_slice = [&](const int kt, const int par) {
            const int buf = par;
            const bool more = kt + 1 < ke;
            const bool fetch = PD2 ? kt + 2 < ke : more;       // is there a slice left to request?
            const int fset = PD2 ? par : 0;                    // its register set: the one slice kt came through
            if (fetch) slice_begin();
            const float4* Ab = As + buf * BM * CH;
            const float4* Bb = Bs + buf * BN * CH;
            // fragment reads run one k-step ahead of the MFMAs that consume them (register double buffer)
            float4 fa[2][FM], fb[2][FN];
            auto read_frags = [&](int kk, int slot) {
                const int c = 2 * kk + h;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int row = (wm * FM + i) * 32 + li;
                    fa[slot][i] = Ab[row * CH + swzc(row, c)];
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int row = (wn * FN + j) * 32 + li;
                    fb[slot][j] = Bb[row * CH + swzc(row, c)];
                }
            };
            read_frags(0, 0);
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const int cur = kk & 1;
#ifndef ERD_IG_NOFRAG      // (timing probe: every k-step re-uses the first one's fragments)
                if (kk + 1 < KSTEPS) read_frags(kk + 1, cur ^ 1);
#endif
                // next slice's global loads, a few per k-step, issued under this step's MFMAs
#ifdef ERD_IG_NOLOAD       // (timing probe: the K loop re-uses the first slices' registers)
                if (false) {
#else
                if (fetch) {
#endif
#pragma unroll
                    for (int q = 0; q < APS; ++q)
                        if (kk * APS + q < AJ) load_a(kk * APS + q, fset);
#pragma unroll
                    for (int q = 0; q < BPS; ++q)
                        if (kk * BPS + q < BJ) load_b(kk * BPS + q, fset);
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        if (BF) {
#ifdef ERD_IG_NOMFMA       // (timing probe: everything but the matrix instructions)
                            acc[i][j][kk] += fa[cur][i].x * fb[cur][j].y;
                            continue;
#endif
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, fa[cur][i]), __builtin_bit_cast(bf16x8, fb[cur][j]),
                                acc[i][j], 0, 0, 0);
                            continue;
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].x, fb[cur][j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].y, fb[cur][j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].z, fb[cur][j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].w, fb[cur][j].w, acc[i][j], 0, 0, 0);
                    }
                if (SGB) {
                    __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0);      // DS reads of the next step first
                    __builtin_amdgcn_sched_group_barrier(0x20, APS + BPS, 0);     // then the next slice's VMEM reads
                    __builtin_amdgcn_sched_group_barrier(0x8, FM * FN * 4, 0);    // then this step's MFMAs
                }
            }
#ifndef ERD_IG_NOSYNC      // (timing probe, with ERD_IG_NOLOAD: no LDS refill and no barrier between slices)
            if (more) store_lds(buf ^ 1, PD2 ? (par ^ 1) : 0);
            __syncthreads();
#endif
        };
        if constexpr (X3) {
            for (int kt = ks; kt < ke; ++kt) k_slice_x3(kt, (kt - ks) & 1);
        } else if constexpr (PD2) {
            for (int kt = ks; kt < ke; kt += 2) {       // unrolled by two: register sets are named at compile time
                k_slice(kt, 0);
                if (kt + 1 < ke) k_slice(kt + 1, 1);
            }
        } else {
            for (int kt = ks; kt < ke; ++kt) k_slice(kt, (kt - ks) & 1);
        }

        IG_ACC(3, t_loop);
        IG_SET(6, g_igemm_trace[blockIdx.x * 8 + 6] + (unsigned long long)(ke - ks));
        IG_T0(t_fix);
        // ---- partial tile: hand the accumulators over; the last contributor to arrive reduces ----------
        if (ks != 0 || ke != nkt_t) {
            const long long t0 = (long long)tt * nkt;
            const int first_b = (int)(((t0 + 1) * G - 1) / U);
            const int last_b = (int)(((t0 + nkt) * G - 1) / U);
            const int ncontrib = last_b - first_b + 1;
            const int my_slot = (tt == (int)(u_begin / nkt)) ? 0 : 1;
            // slab layout: float4 q = (fragment, register quad) of lane tid at [q][tid]: 16-B stores/loads, coalesced
            // Slabs travel with sc1 (write-through) stores and sc1 loads: visible across the XCDs' L2s without the
            // agent-scope release / acquire fences, whose `buffer_wbl2 sc1` writes back EVERY dirty line of the XCD's L2
            // (the output rows other workgroups have just stored) -- measured 27k cycles per workgroup in this hand-over
            // on the 264-tile 1x1 layers, a seventh of the kernel (cdna_hip_programming Guideline 16, sc1 form).
            const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(
                ws.slabs, 0, (int)std::min<size_t>((size_t)2 * G * BM * BN * 4, 0x7fffffffu), 0x00020000);
            const unsigned slab_off = (unsigned)((size_t)(2 * wg + my_slot) * (BM * BN) * 4) + (unsigned)tid * 16u;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        u32x4 v;
                        v.x = __float_as_uint(acc[i][j][4 * g]); v.y = __float_as_uint(acc[i][j][4 * g + 1]);
                        v.z = __float_as_uint(acc[i][j][4 * g + 2]); v.w = __float_as_uint(acc[i][j][4 * g + 3]);
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs_slab, slab_off + (unsigned)(((i * FN + j) * 4 + g) * NT * 16), 0, 16 /* sc1 */);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) bcast[0] = __hip_atomic_fetch_add(ws.cnt + tt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (bcast[0] != ncontrib - 1) { IG_ACC(4, t_fix); IG_SET(1, __builtin_amdgcn_s_memtime()); continue; }   // somebody else finishes this tile
            // Two contributors (the common case): a + b == b + a bit for bit, so the reducer keeps its own
            // accumulators and adds the other slab.  Three or more: re-sum every slab in K order from zero so the
            // result does not depend on who arrived last.
            const bool pair = ncontrib == 2;
            if (!pair) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
            for (int c = 0; c < ncontrib; ++c) {
                const int bb = first_b + c;
                if (pair && bb == wg) continue;
                const int bfirst_tile = (int)(((U * bb) / G) / nkt);
                const unsigned so = (unsigned)((size_t)(2 * bb + (tt == bfirst_tile ? 0 : 1)) * (BM * BN) * 4) + (unsigned)tid * 16u;
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                                rs_slab, so + (unsigned)(((i * FN + j) * 4 + g) * NT * 16), 0, 16 /* sc1 */);
                            acc[i][j][4 * g] += __uint_as_float(v.x); acc[i][j][4 * g + 1] += __uint_as_float(v.y);
                            acc[i][j][4 * g + 2] += __uint_as_float(v.z); acc[i][j][4 * g + 3] += __uint_as_float(v.w);
                        }
            }
            if (tid == 0) ws.cnt[tt] = 0;   // leave the ticket zeroed for the next launch (stream-ordered)
        }

        IG_ACC(4, t_fix);
        IG_T0(t_epi);
        if constexpr (X3) fill_rows(rows_epi);      // (behind the staging area; published by the barrier that opens the first pass)
        // ---- epilogue: accumulators -> LDS (64 rows at a time) -> coalesced float4 rows --------------------
        // A lane owns one output column, so direct stores would be 64 dword stores per lane (store-issue bound).
        // Staging the tile through the (now idle) operand LDS turns them into 16-B stores of whole 512-B rows and
        // lets scale/shift/residual/mask be applied on float4s.
        using OutT = typename std::conditional<OB, erd::bf16s, float>::type;      // storage cell of out / res / mask
        OutT* __restrict__ out = reinterpret_cast<OutT*>(sg.out);
        const OutT* res = reinterpret_cast<const OutT*>(sg.res);
        const OutT* msk = reinterpret_cast<const OutT*>(sg.mask);
        const float alpha = sg.alpha ? *sg.alpha : 1.f;
        const bool has_alpha = sg.alpha != nullptr;
        float* stage = reinterpret_cast<float*>(smem);          // [64][BN + 4]
        constexpr int SLD = BN + 4;
        constexpr int C4N = BN / 4;                             // float4 columns of the tile
        constexpr int RPS = NT / C4N;                     // rows stored per sweep
        if constexpr (X3) {
            // f32x3 (4 x 1 waves): every wave owns 32 complete output rows, so it stages them in its OWN LDS block and writes them
            // out by itself -- all four waves at once and without workgroup barriers between passes (the shared form below is
            // four sequential passes with two barriers each: 14 k of a tile's 56 k cycles on the K = 256 layers).
            {   // (Cout % 4 == 0: erd_conv_igemm sends other launches to the fp32 kernel)
                __syncthreads();                                // the row table behind the staging blocks is complete
                float* wst = reinterpret_cast<float*>(smem + wave * WSTAGE_BYTES);
                constexpr int RPW = 64 / C4N, NIT = 32 / RPW;   // rows per wave-instruction, instructions per block
                const int c4 = lane % C4N, rsub = lane / C4N;
                const int co = n0 + c4 * 4;
                const bool cvalid = co < p.Cout;
                // gfx950 counts loads AND stores in vmcnt, in issue order: a wait for a load that was issued behind a store also waits
                // for that store's acknowledgement (~650 cycles under load), and a store under `if (row valid)` is not counted as
                // "younger" by the compiler, which then waits vmcnt(0) in front of every store (16 serialized acknowledgements per
                // tile until round 5).  So: every residual / mask / scale / shift load of the tile is issued BEFORE its first
                // store, the stores are unconditional buffer stores (rows / columns past the end carry the offset OOB and are
                // dropped), and nothing but arithmetic sits between them.
                const int out_bytes = erd::uniform_int((int)((long long)sg.N * sg.out_nstride * 4));
                const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(erd::uniform_ptr(sg.out), 0, out_bytes, 0x00020000);
                // (a segment without the residual / mask the launch's other segments have gets an EMPTY buffer: its loads return zeros)
                const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
                    erd::uniform_ptr(const_cast<float*>(sg.res ? sg.res : sg.out)), 0, erd::uniform_int(sg.res ? out_bytes : 0), 0x00020000);
                const __amdgpu_buffer_rsrc_t rs_msk = __builtin_amdgcn_make_buffer_rsrc(
                    erd::uniform_ptr(const_cast<float*>(sg.mask ? sg.mask : sg.out)), 0, erd::uniform_int(sg.mask ? out_bytes : 0), 0x00020000);
                const bool no_msk = sg.mask == nullptr;
                auto row_off = [&](int q) -> unsigned {
                    const int oo = rows_epi[wave * 32 + q * RPW + rsub].out_off;
                    return (oo < 0 || !cvalid) ? OOB : (unsigned)(oo + co) * 4u;
                };
                // The wave's rows go out in phases of NP rows; a phase's residual / mask rows (NP x 16 B per lane and map) are requested
                // in front of the previous phase's stores at the earliest -- phase 0's before the accumulators are staged (they hold
                // 128 of the 256 registers: NPQ rows only, the rest once they are in LDS).  A later phase's loads sit behind the
                // previous phase's stores, so their wait covers one round of store acknowledgements: one per tile with a residual OR
                // a mask, three with both, none without.
                constexpr int NP = (RES && MSK) ? NIT / 4 : NIT / 2, NPQ = NP < 4 ? NP : 4;
                float4 pr[RES ? NP : 1], pm[MSK ? NP : 1];
                if constexpr (RES) {
#pragma unroll
                    for (int q = 0; q < NPQ; ++q) pr[q] = buf_load16(rs_res, row_off(q));
                } else if constexpr (MSK) {
#pragma unroll
                    for (int q = 0; q < NPQ; ++q) pm[q] = buf_load16(rs_msk, row_off(q));
                }
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) wst[((r & 3) + 8 * (r >> 2) + 4 * h) * SLD + j * 32 + li] = acc[0][j][r];
                if constexpr (RES) {
#pragma unroll
                    for (int q = NPQ; q < NP; ++q) pr[q] = buf_load16(rs_res, row_off(q));
                }
                if constexpr (MSK) {
#pragma unroll
                    for (int q = RES ? 0 : NPQ; q < NP; ++q) pm[q] = buf_load16(rs_msk, row_off(q));
                }
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cvalid && p.scale) sc = *reinterpret_cast<const float4*>(p.scale + co);
                if (cvalid && p.shift) sh = *reinterpret_cast<const float4*>(p.shift + co);
                __builtin_amdgcn_wave_barrier();                // (LDS operations of one wave execute in order)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int ph = 0; ph < NIT / NP; ++ph) {
                    if (ph > 0) {
                        if constexpr (RES) {
#pragma unroll
                            for (int q = 0; q < NP; ++q) pr[q] = buf_load16(rs_res, row_off(ph * NP + q));
                        }
                        if constexpr (MSK) {
#pragma unroll
                            for (int q = 0; q < NP; ++q) pm[q] = buf_load16(rs_msk, row_off(ph * NP + q));
                        }
                    }
#pragma unroll
                    for (int qq = 0; qq < NP; ++qq) {
                        const int q = ph * NP + qq;
                        const int rr = q * RPW + rsub;
                        float4 v = *reinterpret_cast<const float4*>(wst + rr * SLD + c4 * 4);
                        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                        if (has_alpha) { v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha; }
                        if constexpr (RES) { const float4 rv = pr[qq]; v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
                        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        if constexpr (MSK) {
                            const float4 mv = pm[qq];
                            v.x = (mv.x > 0.f || no_msk) ? v.x : 0.f; v.y = (mv.y > 0.f || no_msk) ? v.y : 0.f;
                            v.z = (mv.z > 0.f || no_msk) ? v.z : 0.f; v.w = (mv.w > 0.f || no_msk) ? v.w : 0.f;
                        }
                        const unsigned o = row_off(q);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, o, 0, 0);
                        if (o != OOB) { csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w; }
                    }
                }
                if (p.colsum) {      // lanes that share a column group, then the four waves through LDS, one atomic per channel
#pragma unroll
                    for (int o = C4N; o < 64; o <<= 1) {
                        csum.x += __shfl_xor(csum.x, o, 64); csum.y += __shfl_xor(csum.y, o, 64);
                        csum.z += __shfl_xor(csum.z, o, 64); csum.w += __shfl_xor(csum.w, o, 64);
                    }
                    float4* red = reinterpret_cast<float4*>(smem + ROWS_EPI_OFF + BM * sizeof(RowInfo) + 64);      // [4][C4N], behind table + broadcast word
                    if (lane < C4N) red[wave * C4N + lane] = cvalid ? csum : make_float4(0.f, 0.f, 0.f, 0.f);
                    __syncthreads();
                    if (tid < C4N && n0 + tid * 4 < p.Cout) {
                        float4 t = red[tid];
                        for (int q = 1; q < 4; ++q) { const float4 v = red[q * C4N + tid]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
                        float* cs = p.colsum + (p.colsum_copies > 1 ? (int64_t)(blockIdx.x & (p.colsum_copies - 1)) * p.Cout : 0);
                        const int cc0 = n0 + tid * 4;
                        atomicAdd(cs + cc0 + 0, t.x); atomicAdd(cs + cc0 + 1, t.y); atomicAdd(cs + cc0 + 2, t.z); atomicAdd(cs + cc0 + 3, t.w);
                    }
                }
            }
        }
#pragma unroll
        for (int half = 0; half < (X3 ? 0 : WAVES_M); ++half) {
            __syncthreads();                                    // operand tiles / previous half fully consumed
            // residual / ReLU-mask rows of this half are requested NOW: their latency hides behind the staging below
            // (inside the store loop every row would wait for its own load: ~1 us x 8 dependent iterations per half)
            constexpr int NPF = (FM * 32) / RPS;
            constexpr bool PREFETCH = MINW <= 3;                // (four workgroups per CU: 128 registers, no room)
            float4 pf[PREFETCH ? NPF : 1];                      // residual rows, or mask rows when there is no residual
            const OutT* pf_src = res ? res : msk;
            if (PREFETCH && pf_src && (p.Cout & 3) == 0 && n0 + (tid % C4N) * 4 < p.Cout) {
#pragma unroll
                for (int q = 0; q < NPF; ++q) {
                    const int oo = rows_epi[half * FM * 32 + tid / C4N + q * RPS].out_off;
                    if (oo >= 0) pf[PREFETCH ? q : 0] = erd::ld4(pf_src + oo + n0 + (tid % C4N) * 4);
                }
            }
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            stage[row * SLD + (wn * FN + j) * 32 + li] = acc[i][j][r];
                        }
            }
            __syncthreads();
            const int c4 = tid % C4N;
            const int co = n0 + c4 * 4;
            float4 csum_keep = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < p.Cout && (p.Cout & 3) != 0) {
                // Cout not a multiple of 4 (e.g. a 70-class teacher head): rows are not 16-B aligned -> scalar stores
                float csum[4] = {0.f, 0.f, 0.f, 0.f};
                for (int rr = tid / C4N; rr < FM * 32; rr += RPS) {
                    const int oo = rows_epi[half * FM * 32 + rr].out_off;
                    if (oo < 0) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (co + e >= p.Cout) break;
                        float x = stage[rr * SLD + c4 * 4 + e];
                        x = x * (p.scale ? p.scale[co + e] : 1.f) + (p.shift ? p.shift[co + e] : 0.f);
                        if (has_alpha) x *= alpha;
                        if (res) x += erd::ld1(res + oo + co + e);
                        if (p.relu) x = fmaxf(x, 0.f);
                        if (msk) x = erd::ld1(msk + oo + co + e) > 0.f ? x : 0.f;
                        erd::st1(out + oo + co + e, x);
                        csum[e] += x;
                    }
                }
                if (p.colsum) csum_keep = make_float4(csum[0], csum[1], csum[2], csum[3]);
            } else if (co < p.Cout) {
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + co);
                if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + co);
                float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < NPF; ++q) {
                    const int rr = tid / C4N + q * RPS;
                    const int oo = rows_epi[half * FM * 32 + rr].out_off;
                    if (oo < 0) continue;
                    float4 v = *reinterpret_cast<const float4*>(stage + rr * SLD + c4 * 4);
                    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                    if (has_alpha) { v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha; }
                    if (res) {
                        const float4 rv = PREFETCH ? pf[PREFETCH ? q : 0] : erd::ld4(res + oo + co);
                        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                    }
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (msk) {      // gradient of the ReLU that produced this tensor's forward twin (fused dz = dy*(y>0))
                        const float4 mv = (PREFETCH && !res) ? pf[PREFETCH ? q : 0] : erd::ld4(msk + oo + co);
                        v.x = mv.x > 0.f ? v.x : 0.f; v.y = mv.y > 0.f ? v.y : 0.f;
                        v.z = mv.z > 0.f ? v.z : 0.f; v.w = mv.w > 0.f ? v.w : 0.f;
                    }
                    erd::st4(out + oo + co, v);
                    csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w;
                }
                if (p.colsum) csum_keep = csum;
            }
            if (p.colsum) {     // per-channel sum of what was stored (d beta of the upstream BN): LDS-combine the row
                                // lanes, then one atomic per channel per half tile
                float4* red = reinterpret_cast<float4*>(stage + 64 * SLD);      // [RPS][C4N], behind the staged rows
                __syncthreads();
                red[tid] = (co < p.Cout) ? csum_keep : make_float4(0.f, 0.f, 0.f, 0.f);
                __syncthreads();
                if (tid < C4N && co < p.Cout) {
                    float4 t = red[tid];
                    for (int q = 1; q < RPS; ++q) {
                        const float4 v = red[q * C4N + tid];
                        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
                    }
                    float* cs = p.colsum + (p.colsum_copies > 1 ? (int64_t)(blockIdx.x & (p.colsum_copies - 1)) * p.Cout : 0);
                    atomicAdd(cs + co + 0, t.x);
                    if (co + 1 < p.Cout) atomicAdd(cs + co + 1, t.y);
                    if (co + 2 < p.Cout) atomicAdd(cs + co + 2, t.z);
                    if (co + 3 < p.Cout) atomicAdd(cs + co + 3, t.w);
                }
            }
        }
        IG_ACC(5, t_epi);
        IG_SET(1, __builtin_amdgcn_s_memtime());
    }
}

// -------------------------------------------------------------------------------------------------
// weight gradient: G[co][t][ci] = sum_p dz[p][co] * x[p+tap t][ci]; K = pixels (split over gridDim.z)
// LDS tiles are k-major ([32 px][128 ch], exactly the global layout); fragments by ds_read_b32.
// -------------------------------------------------------------------------------------------------
template <int BM, int BN, int BKW, int MINW>
__global__ __launch_bounds__(NTHREADS, MINW) void conv_wgrad_kernel(const erd_wgrad_desc p, const int xcd_order) {
    constexpr int BK = BKW;            // pixels per K-slice
    constexpr int FM = BM / 64, FN = BN / 64;  // 2x2 waves
    constexpr int AC = BM / 4, BC = BN / 4;    // float4 chunks per row
    constexpr int AJ = (BK * AC) / NTHREADS, BJ = (BK * BC) / NTHREADS;
    constexpr int AR = NTHREADS / AC, BR = NTHREADS / BC;  // rows covered per pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);   // [2][BK][BM]
    float* Bs = As + 2 * BK * BM;                 // [2][BK][BN]
    int2* offs = reinterpret_cast<int2*>(Bs + 2 * BK * BN);   // [2][BK]: (dz offset, x offset) per pixel, -1 = zero row

    const int tid = threadIdx.x;
    const int nci = (p.Cin + BN - 1) / BN;
    // 1-D grid; consecutive workgroups = the (tap, ci-tile, co-tile) combinations of one pixel range (split).  Left in
    // dispatch order (spread over the 8 XCDs): pinning a split to one XCD's L2 measured 3-20 % SLOWER here (36
    // workgroups hammering the same L2 lines), unlike the igemm kernel (ERD_XCD=2 re-enables it for A/B runs).
    int wg = blockIdx.x;
    if (xcd_order) {
        const int G = gridDim.x, q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nxb = nci * p.ntaps, nyb = (p.Cout + BM - 1) / BM;
    const int bx = wg % nxb, by = (wg / nxb) % nyb, bz = wg / (nxb * nyb);
    const int tap = bx / nci;
    const int ci0 = (bx % nci) * BN;
    const int co0 = by * BM;
    int P = 0;
    for (int l = 0; l < p.nseg; ++l) P += p.seg[l].N * p.seg[l].GH * p.seg[l].GW;
    const int nkt_total = (P + BK - 1) / BK;
    const int per = (nkt_total + p.nsplit - 1) / p.nsplit;
    const int kt_begin = bz * per;
    const int kt_end = min(nkt_total, kt_begin + per);
    const int dyt = p.dy[tap], dxt = p.dx[tap];
    const float* __restrict__ x = p.x;
    const float* __restrict__ dz = p.dz;

    const int achunk = tid % AC, ar0 = tid / AC;
    const int bchunk = tid % BC, br0 = tid / BC;
    const bool a_cok = co0 + achunk * 4 < p.Cout;
    const bool b_cok = ci0 + bchunk * 4 < p.Cin;
    const int a_col = co0 + achunk * 4, b_col = ci0 + bchunk * 4;

    // pixel -> element offsets of one K-slice, computed once by 32 threads (the divisions live here only)
    auto compute_offsets = [&](int kt, int slot) {
        if (tid < BK) {
            int pp = kt * BK + tid;
            int2 o = make_int2(-1, -1);
            if (pp < P && kt < kt_end) {
                int l = 0;
#pragma unroll 1
                for (; l < p.nseg - 1; ++l) {      // which map does this pixel of the concatenated K axis belong to
                    const int pl = p.seg[l].N * p.seg[l].GH * p.seg[l].GW;
                    if (pp < pl) break;
                    pp -= pl;
                }
                const erd_wgrad_seg& g = p.seg[l];
                const int GHW = g.GH * g.GW;
                const int n = pp / GHW;
                const int rem = pp - n * GHW;
                const int a = rem / g.GW;
                const int b = rem - a * g.GW;
                o.x = (int)(g.dz_off + n * g.dz_nstride) +
                      ((a * p.out_stride + p.oy) * g.OW + (b * p.out_stride + p.ox)) * p.Cout;
                const int ih = a * p.in_stride + dyt, iw = b * p.in_stride + dxt;
                if ((unsigned)ih < (unsigned)g.IH && (unsigned)iw < (unsigned)g.IW)
                    o.y = (int)(g.x_off + n * g.x_nstride) + (ih * g.IW + iw) * p.Cin;
            }
            offs[slot * BK + tid] = o;
        }
    };

    const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(dz), 0, (int)(p.dz_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x), 0, (int)(p.x_elems * 4), 0x00020000);
    float4 ra[AJ], rb[BJ];
    auto load_global = [&](int slot) {   // branch-free: zero rows come from the buffer's out-of-range rule
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int o = offs[slot * BK + ar0 + AR * j].x;
            ra[j] = buf_load16(rs_dz, (o >= 0 && a_cok) ? (unsigned)(o + a_col) * 4u : OOB);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int o = offs[slot * BK + br0 + BR * j].y;
            rb[j] = buf_load16(rs_x, (o >= 0 && b_cok) ? (unsigned)(o + b_col) * 4u : OOB);
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            *reinterpret_cast<float4*>(As + (buf * BK + ar0 + AR * j) * BM + achunk * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            *reinterpret_cast<float4*>(Bs + (buf * BK + br0 + BR * j) * BN + bchunk * 4) = rb[j];
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt_begin < kt_end) {
        compute_offsets(kt_begin, 0);
        __syncthreads();
        load_global(0);
        compute_offsets(kt_begin + 1, 1);
        store_lds(0);
        __syncthreads();
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int buf = (kt - kt_begin) & 1;
            const bool more = kt + 1 < kt_end;
            if (more) load_global(buf ^ 1);     // offsets of slice kt+1 were published before the last barrier
            compute_offsets(kt + 2, buf);       // slot `buf` was last read while loading slice kt (previous iteration)
            const float* Ab = As + buf * BK * BM;
            const float* Bb = Bs + buf * BK * BN;
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                const int k = 2 * ks + h;
                float fa[FM], fb[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) fa[i] = Ab[k * BM + (wm * FM + i) * 32 + li];
#pragma unroll
                for (int j = 0; j < FN; ++j) fb[j] = Bb[k * BN + (wn * FN + j) * 32 + li];
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
            if (more) store_lds(buf ^ 1);
            __syncthreads();
        }
    }
    // partial slab [z][Cout][ntaps][Cin]
    float* __restrict__ part = p.part + (int64_t)bz * p.Cout * p.ntaps * p.Cin;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int ci = ci0 + (wn * FN + j) * 32 + li;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wm * FM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < p.Cout) part[((int64_t)co * p.ntaps + tap) * p.Cin + ci] = acc[i][j][r];
            }
    }
}

// -------------------------------------------------------------------------------------------------
// weight gradient of a 3x3 stride-1 pad-1 convolution, THREE taps (one kernel row ky) per workgroup.
// A K-slice is 16 consecutive output pixels of ONE image row; the x slice carries a one-pixel halo on both sides
// (18 entries, out-of-image entries are zero rows), so tap kx simply reads entries k+kx: the dz slice and the x slice
// are fetched and staged once for three 128x128 products (the generic kernel fetches both once per tap: 2.8x the
// global->LDS traffic and LDS writes per MFMA, and 3x the barriers).  192 accumulator registers -> two workgroups
// per CU.  Partial slabs / split-K / reduce kernel are shared with the generic path.
// -------------------------------------------------------------------------------------------------
// Wave tiling <WAVES_M, FM, FN>: the 4 waves form a WAVES_M x (4 / WAVES_M) grid, each owning FM x FN 32x32 blocks per tap.
// <2,2,2>: 128 output channels x 128 input channels.  <1,3,1> / <1,2,1>: 96 / 64 output channels (the heads' 80- and
// 68-channel convolutions would waste 37-47 % of a 128-row tile).
template <int WAVES_M, int FM, int FN>
__global__ __launch_bounds__(NTHREADS, 2) void conv_wgrad_row3_kernel(const erd_wgrad_desc p, const int nslices_xcd) {
    const int nslices = nslices_xcd & 0x3fffffff;       // (bit 30: XCD-aware work order, see below)
    constexpr int BM = 128, BN = 128, BK = 16, BX = BK + 2;     // LDS tile widths (loads beyond BME rows are masked)
    constexpr int WAVES_N = 4 / WAVES_M, BME = WAVES_M * FM * 32;
    static_assert(WAVES_N * FN * 32 == BN && BME <= BM, "wave tiling");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);        // [2][BK][BM]   dz
    float* Bs = As + 2 * BK * BM;                      // [2][BX][BN]   x with halo
    int* offa = reinterpret_cast<int*>(Bs + 2 * BX * BN);   // [2][BK] dz element offsets (-1: zero row)
    int* offb = offa + 2 * BK;                         // [2][BX] x element offsets (-1: zero row)

    const int tid = threadIdx.x;
    const int nci = (p.Cin + BN - 1) / BN, nco = (p.Cout + BME - 1) / BME;
    // The (cin block, kernel row, cout block) workgroups of one K split read the same dz / x slices.  Workgroup b runs on XCD
    // b % 8 (each XCD has its own L2): give every XCD a CONTIGUOUS range of the work list so that the siblings of a split
    // share one L2 instead of pulling the slices through the fabric up to eight times.
    int wg = blockIdx.x;
    if (nslices_xcd >> 30) {
        const int G = gridDim.x, q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bx = wg % (nci * 3), by = (wg / (nci * 3)) % nco, bz = wg / (nci * 3 * nco);
    const int ky = bx / nci;
    const int ci0 = (bx % nci) * BN;
    const int co0 = by * BME;
    const int per = (nslices + p.nsplit - 1) / p.nsplit;
    const int kt_begin = bz * per;
    const int kt_end = min(nslices, kt_begin + per);
    const float* __restrict__ x = p.x;
    const float* __restrict__ dz = p.dz;

    const int chunk = tid & 31, r0 = tid >> 5;          // 32 float4 chunks per 128-channel row, 8 rows per pass
    const bool a_cok = chunk * 4 < BME && co0 + chunk * 4 < p.Cout;
    const bool b_cok = ci0 + chunk * 4 < p.Cin;
    const int a_col = co0 + chunk * 4, b_col = ci0 + chunk * 4;

    // slice -> (map, image, row, 16-pixel chunk) -> element offsets; 18 threads, the divisions live here only
    auto compute_offsets = [&](int kt, int slot) {
        if (tid < BX) {
            int oa = -1, ob = -1;
            if (kt < kt_end) {
                int l = 0, q = kt;
#pragma unroll 1
                for (; l < p.nseg - 1; ++l) {
                    const int cnt = p.seg[l].N * p.seg[l].GH * ((p.seg[l].GW + BK - 1) / BK);
                    if (q < cnt) break;
                    q -= cnt;
                }
                const erd_wgrad_seg& g = p.seg[l];
                const int cpr = (g.GW + BK - 1) / BK;
                const int c = q % cpr;
                const int rowi = q / cpr;
                const int a = rowi % g.GH, n = rowi / g.GH;
                const int bcol = c * BK + tid - 1;             // x column of entry `tid` (halo of one on each side)
                const int ih = a + ky - 1;
                if ((unsigned)bcol < (unsigned)g.IW && (unsigned)ih < (unsigned)g.IH)
                    ob = (int)(g.x_off + n * g.x_nstride) + (ih * g.IW + bcol) * p.Cin;
                const int zcol = c * BK + tid;
                if (tid < BK && zcol < g.GW) oa = (int)(g.dz_off + n * g.dz_nstride) + (a * g.OW + zcol) * p.Cout;
            }
            if (tid < BK) offa[slot * BK + tid] = oa;
            offb[slot * BX + tid] = ob;
        }
    };

    const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(dz), 0, (int)(p.dz_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x), 0, (int)(p.x_elems * 4), 0x00020000);
    float4 ra[2], rb[2], rh;
    auto load_global = [&](int slot) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = offa[slot * BK + r0 + 8 * j];
            ra[j] = buf_load16(rs_dz, (o >= 0 && a_cok) ? (unsigned)(o + a_col) * 4u : OOB);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = offb[slot * BX + r0 + 8 * j];
            rb[j] = buf_load16(rs_x, (o >= 0 && b_cok) ? (unsigned)(o + b_col) * 4u : OOB);
        }
        if (tid < 64) {                                  // the two halo-side entries 16, 17
            const int o = offb[slot * BX + 16 + r0];
            rh = buf_load16(rs_x, (o >= 0 && b_cok) ? (unsigned)(o + b_col) * 4u : OOB);
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<float4*>(As + (buf * BK + r0 + 8 * j) * BM + chunk * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<float4*>(Bs + (buf * BX + r0 + 8 * j) * BN + chunk * 4) = rb[j];
        if (tid < 64) *reinterpret_cast<float4*>(Bs + (buf * BX + 16 + r0) * BN + chunk * 4) = rh;
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, h = lane >> 5;
    f32x16 acc[3][FM][FN];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

    if (kt_begin < kt_end) {
        compute_offsets(kt_begin, 0);
        __syncthreads();
        load_global(0);
        compute_offsets(kt_begin + 1, 1);
        store_lds(0);
        __syncthreads();
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int buf = (kt - kt_begin) & 1;
            const bool more = kt + 1 < kt_end;
            if (more) load_global(buf ^ 1);
            compute_offsets(kt + 2, buf);
            const float* Ab = As + buf * BK * BM + wm * FM * 32 + li;
            const float* Bb = Bs + buf * BX * BN + wn * FN * 32 + li;
            // lane half h walks pixels k = 2*ks + h; tap kx reads x entry k + kx.  Entry k+2 of this step is entry
            // (k+2)+0 of the next one: carried over in registers.
            float fb0[FN], fb1[FN], fb2[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) { fb0[j] = Bb[h * BN + j * 32]; fb1[j] = Bb[(h + 1) * BN + j * 32]; }
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                const int k = 2 * ks + h;
                float fa[FM];
#pragma unroll
                for (int i = 0; i < FM; ++i) fa[i] = Ab[k * BM + i * 32];
#pragma unroll
                for (int j = 0; j < FN; ++j) fb2[j] = Bb[(k + 2) * BN + j * 32];
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb0[j], acc[0][i][j], 0, 0, 0);
                        acc[1][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb1[j], acc[1][i][j], 0, 0, 0);
                        acc[2][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb2[j], acc[2][i][j], 0, 0, 0);
                    }
                if (ks + 1 < BK / 2) {
#pragma unroll
                    for (int j = 0; j < FN; ++j) { fb0[j] = fb2[j]; fb1[j] = Bb[(k + 3) * BN + j * 32]; }
                }
            }
            if (more) store_lds(buf ^ 1);
            __syncthreads();
        }
    }
    float* __restrict__ part = p.part + (int64_t)bz * p.Cout * 9 * p.Cin;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int ci = ci0 + (wn * FN + j) * 32 + li;
            if (ci >= p.Cin) continue;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (wm * FM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (co < p.Cout) part[((int64_t)co * 9 + ky * 3 + t) * p.Cin + ci] = acc[t][i][j][r];
                }
        }
}

// -------------------------------------------------------------------------------------------------
// The three-tap weight gradient in the THREE-LIMB form (erd_wgrad_desc::limbs3): fp32 dz and x, fp32 partial slabs, every
// product formed on the bf16 matrix cores from exact limb splits of BOTH operands (see conv_igemm_kernel<..., X3>).
//   * A K-slice is 16 output pixels of one image row (x: 18 entries with the one-pixel halo), as in the fp32 kernel.  The
//     bf16 MFMA wants 8 consecutive PIXELS of one channel per lane while memory is pixel-major: each thread loads a
//     4-pixel x 4-channel micro-tile (four 16-B loads), splits its 16 values into limbs (cvt_pk / sub / cvt_pk / sub, exact) and
//     stores them transposed -- 8 bytes (4 pixels of one channel) per limb plane -- into channel-major LDS rows:
//     dz [3 planes][BMR co][16 px] (32-B rows, chunk XOR-swizzled), x [3 planes][64 ci][18 px] (48-B rows: 16 lanes x 12
//     banks tile all 64 banks).  Every value is split exactly once.
//   * tap kx of the kernel row reads x entries k + kx.  A lane holds entries 8h .. 8h+7 as four packed dwords plus the dword
//     of entries 8h+8, 8h+9: kx = 0 is (d0..d3), kx = 2 is (d1..d4) -- a renaming --, kx = 1 is four v_alignbit_b32.
//   * 128 (or 64) output channels x 64 input channels x 3 taps per workgroup, 2 x 2 waves of (64 | 32) x 32: 96 accumulator
//     registers, 36 MFMAs per k16 step and wave for 9 LDS reads and 12 shift operations; LDS double-buffered (42 KB).
// Partial slabs / split-K / reduce kernel are shared with the other weight-gradient kernels.
// -------------------------------------------------------------------------------------------------
#ifndef ERD_W3X3_MINW
#define ERD_W3X3_MINW 3      // three workgroups per CU (168 registers, 42 KB of LDS each): 181 vs 171 TF at two
#endif
// ROW3 = false: the same machinery for every OTHER layer (1x1, stride 2): one tap per workgroup, K-slices = 16 consecutive pixels
// of the concatenated pixel axis with a per-pixel offset table (as conv_wgrad_kernel), no halo, and -- with a third of the
// products per loaded byte -- 128 input channels per workgroup (FN = 2: 2 x 2 waves of 64 x 64).
template <int FM, int FN, bool ROW3>
__global__ __launch_bounds__(NTHREADS, ERD_W3X3_MINW) void conv_wgrad_row3_x3_kernel(const erd_wgrad_desc p, const int nslices_xcd) {
    const int nslices = nslices_xcd & 0x3fffffff;
    constexpr int NT = ROW3 ? 3 : 1;
    constexpr int BK = 16, BX = ROW3 ? BK + 2 : BK, BMR = FM * 64, BNR = FN * 64;
    constexpr int A_ROWB = 32, B_ROWB = ROW3 ? 48 : 32;        // bytes per LDS row of one plane
    constexpr int A_PL = BMR * A_ROWB, B_PL = BNR * B_ROWB;    // bytes per plane
    constexpr int BUF = 3 * (A_PL + B_PL);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* offa = reinterpret_cast<int*>(smem + 2 * BUF);        // [2][BK]
    int* offb = offa + 2 * BK;                                 // [2][BX]

    const int tid = threadIdx.x;
    const int nci = (p.Cin + BNR - 1) / BNR, nco = (p.Cout + BMR - 1) / BMR;
    int wg = blockIdx.x;
    if (nslices_xcd >> 30) {
        const int G = gridDim.x, q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nbx = nci * (ROW3 ? 3 : p.ntaps);
    const int bx = wg % nbx, by = (wg / nbx) % nco, bz = wg / (nbx * nco);
    const int ky = bx / nci;                                   // ROW3: kernel row; otherwise the tap index
    const int ci0 = (bx % nci) * BNR;
    const int co0 = by * BMR;
    const int per = (nslices + p.nsplit - 1) / p.nsplit;
    const int kt_begin = bz * per;
    const int kt_end = min(nslices, kt_begin + per);

    int P = 0;                                                 // (generic form: pixels of all maps)
    if (!ROW3)
        for (int l = 0; l < p.nseg; ++l) P += p.seg[l].N * p.seg[l].GH * p.seg[l].GW;
    auto compute_offsets = [&](int kt, int slot) {
        if (!ROW3) {
            if (tid < BK) {
                int pp = kt * BK + tid, oa = -1, ob = -1;
                if (pp < P && kt < kt_end) {
                    int l = 0;
#pragma unroll 1
                    for (; l < p.nseg - 1; ++l) {
                        const int pl = p.seg[l].N * p.seg[l].GH * p.seg[l].GW;
                        if (pp < pl) break;
                        pp -= pl;
                    }
                    const erd_wgrad_seg& g = p.seg[l];
                    const int GHW = g.GH * g.GW;
                    const int n = pp / GHW;
                    const int rem = pp - n * GHW;
                    const int a = rem / g.GW;
                    const int b = rem - a * g.GW;
                    oa = (int)(g.dz_off + n * g.dz_nstride) + ((a * p.out_stride + p.oy) * g.OW + (b * p.out_stride + p.ox)) * p.Cout;
                    const int ih = a * p.in_stride + p.dy[ky], iw = b * p.in_stride + p.dx[ky];
                    if ((unsigned)ih < (unsigned)g.IH && (unsigned)iw < (unsigned)g.IW)
                        ob = (int)(g.x_off + n * g.x_nstride) + (ih * g.IW + iw) * p.Cin;
                }
                offa[slot * BK + tid] = oa;
                offb[slot * BX + tid] = ob;
            }
            return;
        }
        if (tid < BX) {
            int oa = -1, ob = -1;
            if (kt < kt_end) {
                int l = 0, q = kt;
#pragma unroll 1
                for (; l < p.nseg - 1; ++l) {
                    const int cnt = p.seg[l].N * p.seg[l].GH * ((p.seg[l].GW + BK - 1) / BK);
                    if (q < cnt) break;
                    q -= cnt;
                }
                const erd_wgrad_seg& g = p.seg[l];
                const int cpr = (g.GW + BK - 1) / BK;
                const int c = q % cpr;
                const int rowi = q / cpr;
                const int a = rowi % g.GH, n = rowi / g.GH;
                const int bcol = c * BK + tid - 1;
                const int ih = a + ky - 1;
                if ((unsigned)bcol < (unsigned)g.IW && (unsigned)ih < (unsigned)g.IH)
                    ob = (int)(g.x_off + n * g.x_nstride) + (ih * g.IW + bcol) * p.Cin;
                const int zcol = c * BK + tid;
                if (tid < BK && zcol < g.GW) oa = (int)(g.dz_off + n * g.dz_nstride) + (a * g.OW + zcol) * p.Cout;
            }
            if (tid < BK) offa[slot * BK + tid] = oa;
            offb[slot * BX + tid] = ob;
        }
    };

    const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dz), 0, (int)(p.dz_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)(p.x_elems * 4), 0x00020000);

    // ---- staging roles: threads [0, NA) own a dz micro-tile (4 px x 4 co), threads [128, 128 + 80) an x micro-tile (4 entries x 4 ci)
    constexpr int NA = (BMR / 4) * 4;                           // dz micro-tiles: (BMR / 4) channel groups x 4 pixel groups
    constexpr int NBG = (BX + 3) / 4;                           // x entry groups: 5 with the halo, 4 without
    constexpr int NB = (BNR / 4) * ((NBG + 1) / 2 * 2);        // x micro-tiles: 16 x 5 with the halo (a sixth, idle, group pads the lane pairs), 32 x 4 without
    static_assert(NA <= 128 && NB <= 128, "staging roles");
    const int tb = tid - 128;
    // Both operands: neighbouring lanes alternate between two pixel groups and the LDS rows are PERMUTED (channel 4 cg + c lives in
    // row cg + (rows / 4) c, undone when the partial slab is written): the 8-byte limb stores of a half-wave then hit 32 different
    // bank pairs.  With channel-major rows and lanes = consecutive channel groups every store instruction touched 8 banks (row stride
    // 128 B): SQ_LDS_BANK_CONFLICT 0.17 of the wave cycles against 0.00 for the implicit GEMM (tools/pmc_waves.py)
    constexpr int QA = BMR / 4, QB = BNR / 4;
    const bool is_a = tid < NA;
    const int cg = is_a ? (tid >> 1) % QA : (tb >> 1) % QB;     // channel group
    const int pg = is_a ? (tid & 1) + 2 * (tid / (2 * QA)) : (tb & 1) + 2 * (tb / (2 * QB));     // group of 4 pixels (dz: 0..3) / entries (x: 0..4)
    const bool is_b = tid >= 128 && tid < 128 + NB && pg < NBG;
    const int col = is_a ? co0 + cg * 4 : ci0 + cg * 4;
    const bool cok = is_a ? (col < p.Cout) : (is_b && col < p.Cin);
    // a wave loads dz OR x (NA is a multiple of 64): the descriptor is picked on the scalar side -- picked per lane (`is_a ? rs_dz : rs_x`)
    // the compiler wraps every load of the K loop into a readfirstlane "waterfall" loop of a dozen instructions
    static_assert(NA % 64 == 0, "staging roles are wave-uniform");
    const bool a_wave = __builtin_amdgcn_readfirstlane(is_a ? 1 : 0) != 0;
    const __amdgpu_buffer_rsrc_t rs_mine = a_wave ? rs_dz : rs_x;
    float4 rv[4];
    auto load_global = [&](int slot) {
        if (is_a || is_b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = pg * 4 + i;
                const int o = is_a ? offa[slot * BK + e] : (e < BX ? offb[slot * BX + e] : -1);
                rv[i] = buf_load16(rs_mine, (o >= 0 && cok) ? (unsigned)(o + col) * 4u : OOB);
            }
        }
    };
    auto store_lds = [&](int buf) {
        if (!(is_a || is_b)) return;
        const float v[4][4] = {{rv[0].x, rv[0].y, rv[0].z, rv[0].w}, {rv[1].x, rv[1].y, rv[1].z, rv[1].w},
                               {rv[2].x, rv[2].y, rv[2].z, rv[2].w}, {rv[3].x, rv[3].y, rv[3].z, rv[3].w}};     // [pixel][channel]
        char* base = smem + buf * BUF + (is_a ? 0 : 3 * A_PL);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = is_a ? cg + QA * c : cg + QB * c;
            uint2 hi, mid, lo;      // four pixels of channel c per limb plane (round-to-nearest limbs, erd_common.h)
            erd::limbs3_pair(v[0][c], v[1][c], hi.x, mid.x, lo.x);
            erd::limbs3_pair(v[2][c], v[3][c], hi.y, mid.y, lo.y);
            // dz: 32-B rows, 16-B chunk (pg >> 1) swizzled by (row >> 3) & 1; x: 48-B rows, entry e at byte 2 e
            const int off = (is_a || !ROW3) ? row * A_ROWB + (((pg >> 1) ^ ((row >> 3) & 1)) << 4) + ((pg & 1) << 3) : row * B_ROWB + pg * 8;
            const int plane = is_a ? A_PL : B_PL;
            *reinterpret_cast<uint2*>(base + off) = hi;
            *reinterpret_cast<uint2*>(base + plane + off) = mid;
            *reinterpret_cast<uint2*>(base + 2 * plane + off) = lo;
        }
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;
    f32x16 acc[NT][FM][FN];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    if (kt_begin < kt_end) {
        compute_offsets(kt_begin, 0);
        __syncthreads();
        load_global(0);
        compute_offsets(kt_begin + 1, 1);
        store_lds(0);
        __syncthreads();
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int buf = (kt - kt_begin) & 1;
            const bool more = kt + 1 < kt_end;
            // (the offset table of slice kt + 2 BEFORE the requests of slice kt + 1: at 168 registers its index arithmetic reloads a
            //  spilled value from scratch, scratch loads retire through the same in-order counter as the global ones, and the
            //  reload's wait would cover the four requests just issued -- the whole memory latency in front of this slice's MFMAs)
            compute_offsets(kt + 2, buf);
            if (more) load_global(buf ^ 1);
            const char* Ab = smem + buf * BUF;
            const char* Bb = Ab + 3 * A_PL;
            // fragments: dz rows (wm * FM + i) * 32 + li, chunk h; x row wn * 32 + li, entries 8h .. 8h + 9
            u4v fa[FM][3];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int row = (wm * FM + i) * 32 + li;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fa[i][pl] = *reinterpret_cast<const u4v*>(Ab + pl * A_PL + row * A_ROWB + ((h ^ ((row >> 3) & 1)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int brow = (wn * FN + j) * 32 + li;
                u4v fb[3];
                unsigned fe[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    if constexpr (ROW3) {
                        fb[pl] = *reinterpret_cast<const u4v*>(Bb + pl * B_PL + brow * B_ROWB + (h << 4));
                        fe[pl] = *reinterpret_cast<const unsigned*>(Bb + pl * B_PL + brow * B_ROWB + (h << 4) + 16);
                    } else {
                        fb[pl] = *reinterpret_cast<const u4v*>(Bb + pl * B_PL + brow * B_ROWB + ((h ^ ((brow >> 3) & 1)) << 4));
                        fe[pl] = 0u;
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    bf16x8 xb[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        u4v s_;
                        if (t == 0) s_ = fb[pl];
                        else if (t == 2) { s_[0] = fb[pl][1]; s_[1] = fb[pl][2]; s_[2] = fb[pl][3]; s_[3] = fe[pl]; }
                        else {
                            s_[0] = __builtin_amdgcn_alignbit(fb[pl][1], fb[pl][0], 16);
                            s_[1] = __builtin_amdgcn_alignbit(fb[pl][2], fb[pl][1], 16);
                            s_[2] = __builtin_amdgcn_alignbit(fb[pl][3], fb[pl][2], 16);
                            s_[3] = __builtin_amdgcn_alignbit(fe[pl], fb[pl][3], 16);
                        }
                        xb[pl] = __builtin_bit_cast(bf16x8, s_);
                    }
#define ERD_W3(APL, BPL)                                                                                              \
                    _Pragma("unroll") for (int i = 0; i < FM; ++i)                                                    \
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][APL]), xb[BPL], acc[t][i][j], 0, 0, 0);
                    ERD_W3(0, 2) ERD_W3(0, 1) ERD_W3(1, 1) ERD_W3(2, 0) ERD_W3(1, 0) ERD_W3(0, 0)
#undef ERD_W3
                }
            }
            if (more) store_lds(buf ^ 1);
            __syncthreads();
        }
    }
    const int ntaps_all = ROW3 ? 9 : p.ntaps;
    float* __restrict__ part = p.part + (int64_t)bz * p.Cout * ntaps_all * p.Cin;
    if ((p.Cin & 3) == 0) {
        // ---- partial slab through LDS, one tap (and at most 128 x 64 / 64 x 128 values) at a time: the accumulators' one-float-per-lane
        // layout would be 16 x FM x FN x taps four-byte stores per lane (8 % of a launch, timing probe ERD_WG3_NOSLAB); staged, the
        // (co, ci) permutations of the operand rows are undone on the way and every thread writes whole 16-byte pieces of slab rows
        constexpr int SLDW = BNR + 4;                                            // floats per staged row
        constexpr int RPASS = (BMR * SLDW * 4 <= 2 * BUF) ? BMR : BMR / 2;       // accumulator rows per pass
        constexpr int NPASS = BMR / RPASS, C4 = BNR / 4;
        static_assert(RPASS * SLDW * 4 <= 2 * BUF && RPASS % 32 == 0, "slab staging");
        float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                __syncthreads();               // the K loop's fragment reads / the previous pass are done with the operand buffers
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    if (((wm * FM + i) * 32) / RPASS != ps) continue;            // (wave-uniform)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        const int brow = (wn * FN + j) * 32 + li;                  // LDS row of the x operand -> input channel of the tile
                        const int col = 4 * (brow % QB) + brow / QB;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int arow = (wm * FM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            stage[(arow - ps * RPASS) * SLDW + col] = acc[t][i][j][r];
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < RPASS * C4 / NTHREADS; ++q) {
                    const int idx = q * NTHREADS + tid;
                    const int row = idx / C4, c4 = idx - row * C4;
                    const int arow = ps * RPASS + row;                              // LDS row of the dz operand -> output channel
                    const int co = co0 + 4 * (arow % QA) + arow / QA, ci = ci0 + 4 * c4;
                    if (co < p.Cout && ci < p.Cin)
                        *reinterpret_cast<float4*>(part + ((int64_t)co * ntaps_all + (ROW3 ? ky * 3 + t : ky)) * p.Cin + ci) =
                            *reinterpret_cast<const float4*>(stage + row * SLDW + 4 * c4);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int brow = (wn * FN + j) * 32 + li;              // LDS row of the x operand ...
        const int ci = ci0 + 4 * (brow % QB) + brow / QB;      // ... holds this input channel
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int arow = (wm * FM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;      // LDS row of the dz operand ...
                    const int co = co0 + 4 * (arow % QA) + arow / QA;                          // ... holds this output channel
#ifdef ERD_WG3_NOSLAB      // timing probe: the partial slab is not written (results are wrong; read against the build BEFORE the staged slab write)
                    if (co < p.Cout) asm volatile("" :: "v"(acc[t][i][j][r]));
#else
                    if (co < p.Cout) part[((int64_t)co * ntaps_all + (ROW3 ? ky * 3 + t : ky)) * p.Cin + ci] = acc[t][i][j][r];
#endif
                }
    }
}

// -------------------------------------------------------------------------------------------------
// weight gradient on the bf16 matrix cores (erd_wgrad_desc::bf16_multiplicands): same GEMM over pixels, but the MFMA
// wants 8 consecutive K values (pixels) of ONE channel per lane while memory is pixel-major.  Each thread therefore
// loads an 8-pixel x 4-channel micro-tile (8 coalesced 16-B loads), rounds to bf16 and transposes it in registers
// into four 16-B vectors (one channel, 8 pixels each) that go to channel-major LDS rows [128 ch][64 px] with the
// 128-B-row XOR swizzle; fragments are then single ds_read_b128s and one v_mfma_f32_32x32x16_bf16 covers 16 pixels.
// Partial slabs, split-K and the reduce kernel are shared with the fp32 path.
// -------------------------------------------------------------------------------------------------
// XB / DB: x / dz are STORED as bf16 (erd_wgrad_desc::x_bf16 / dz_bf16): the micro-tile is then eight 8-B loads (4 channels of
// one pixel each) and the register transpose is a v_perm_b32 per pixel pair instead of a v_cvt_pk_bf16_f32.
template <bool XB, bool DB>
__global__ __launch_bounds__(NTHREADS, 2) void conv_wgrad_bf16_kernel(const erd_wgrad_desc p) {
    constexpr int BM = 128, BN = 128, BK = 64;          // BK pixels per K-slice = 8 chunks of 8 pixels
    constexpr int FM = 2, FN = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* As = reinterpret_cast<float4*>(smem);          // [2][BM rows][8 chunks]
    float4* Bs = As + 2 * BM * 8;                          // [2][BN rows][8 chunks]
    int2* offs = reinterpret_cast<int2*>(Bs + 2 * BN * 8); // [2][BK]

    const int tid = threadIdx.x;
    const int nci = (p.Cin + BN - 1) / BN;
    const int wg = blockIdx.x;
    const int nxb = nci * p.ntaps, nyb = (p.Cout + BM - 1) / BM;
    const int bx = wg % nxb, by = (wg / nxb) % nyb, bz = wg / (nxb * nyb);
    const int tap = bx / nci;
    const int ci0 = (bx % nci) * BN;
    const int co0 = by * BM;
    int P = 0;
    for (int l = 0; l < p.nseg; ++l) P += p.seg[l].N * p.seg[l].GH * p.seg[l].GW;
    const int nkt_total = (P + BK - 1) / BK;
    const int per = (nkt_total + p.nsplit - 1) / p.nsplit;
    const int kt_begin = bz * per;
    const int kt_end = min(nkt_total, kt_begin + per);
    const int dyt = p.dy[tap], dxt = p.dx[tap];

    const int cc = tid & 31, pg = tid >> 5;               // 4-channel group, 8-pixel group of this thread's micro-tiles
    const bool a_cok = co0 + cc * 4 < p.Cout;
    const bool b_cok = ci0 + cc * 4 < p.Cin;
    const int a_col = co0 + cc * 4, b_col = ci0 + cc * 4;

    auto compute_offsets = [&](int kt, int slot) {
        if (tid < BK) {
            int pp = kt * BK + tid;
            int2 o = make_int2(-1, -1);
            if (pp < P && kt < kt_end) {
                int l = 0;
#pragma unroll 1
                for (; l < p.nseg - 1; ++l) {
                    const int pl = p.seg[l].N * p.seg[l].GH * p.seg[l].GW;
                    if (pp < pl) break;
                    pp -= pl;
                }
                const erd_wgrad_seg& g = p.seg[l];
                const int GHW = g.GH * g.GW;
                const int n = pp / GHW;
                const int rem = pp - n * GHW;
                const int a = rem / g.GW;
                const int b = rem - a * g.GW;
                o.x = (int)(g.dz_off + n * g.dz_nstride) +
                      ((a * p.out_stride + p.oy) * g.OW + (b * p.out_stride + p.ox)) * p.Cout;
                const int ih = a * p.in_stride + dyt, iw = b * p.in_stride + dxt;
                if ((unsigned)ih < (unsigned)g.IH && (unsigned)iw < (unsigned)g.IW)
                    o.y = (int)(g.x_off + n * g.x_nstride) + (ih * g.IW + iw) * p.Cin;
            }
            offs[slot * BK + tid] = o;
        }
    };

    const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dz), 0, (int)(p.dz_elems * (DB ? 2 : 4)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)(p.x_elems * (XB ? 2 : 4)), 0x00020000);
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    float4 ra[DB ? 1 : 8], rb[XB ? 1 : 8];          // fp32-stored operand: 4 channels of one pixel per load
    u32x2 ua[DB ? 8 : 1], ub[XB ? 8 : 1];           // bf16-stored operand: the same 4 channels in 8 bytes
    auto load_global = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int2 o = offs[slot * BK + pg * 8 + i];
            if (DB) ua[DB ? i : 0] = __builtin_amdgcn_raw_buffer_load_b64(rs_dz, (o.x >= 0 && a_cok) ? (unsigned)(o.x + a_col) * 2u : OOB, 0, 0);
            else ra[DB ? 0 : i] = buf_load16(rs_dz, (o.x >= 0 && a_cok) ? (unsigned)(o.x + a_col) * 4u : OOB);
            if (XB) ub[XB ? i : 0] = __builtin_amdgcn_raw_buffer_load_b64(rs_x, (o.y >= 0 && b_cok) ? (unsigned)(o.y + b_col) * 2u : OOB, 0, 0);
            else rb[XB ? 0 : i] = buf_load16(rs_x, (o.y >= 0 && b_cok) ? (unsigned)(o.y + b_col) * 4u : OOB);
        }
    };
    auto swz8 = [](int row, int c) { return c ^ ((row >> 1) & 7); };
    auto store_lds = [&](int buf) {
        float4* Ad = As + buf * BM * 8;
        float4* Bd = Bs + buf * BN * 8;
#define ERD_T4(v, m) make_float4(pack_bf16(v[0].m, v[1].m), pack_bf16(v[2].m, v[3].m), pack_bf16(v[4].m, v[5].m), \
                                 pack_bf16(v[6].m, v[7].m))
        // bf16-stored: dword m of pixel i holds channels (2m, 2m+1); channel vectors gather the low / high halves of
        // the eight pixels (v_perm_b32: bytes 0-3 come from the second operand, 4-7 from the first)
#define ERD_P4(u, m, sel) make_float4(__uint_as_float(__builtin_amdgcn_perm(u[1].m, u[0].m, sel)),   \
                                      __uint_as_float(__builtin_amdgcn_perm(u[3].m, u[2].m, sel)),   \
                                      __uint_as_float(__builtin_amdgcn_perm(u[5].m, u[4].m, sel)),   \
                                      __uint_as_float(__builtin_amdgcn_perm(u[7].m, u[6].m, sel)))
        constexpr unsigned LO = 0x05040100u, HI = 0x07060302u;
        const int r = cc * 4;
        if constexpr (DB) {
            Ad[(r + 0) * 8 + swz8(r + 0, pg)] = ERD_P4(ua, x, LO);
            Ad[(r + 1) * 8 + swz8(r + 1, pg)] = ERD_P4(ua, x, HI);
            Ad[(r + 2) * 8 + swz8(r + 2, pg)] = ERD_P4(ua, y, LO);
            Ad[(r + 3) * 8 + swz8(r + 3, pg)] = ERD_P4(ua, y, HI);
        } else {
            Ad[(r + 0) * 8 + swz8(r + 0, pg)] = ERD_T4(ra, x);
            Ad[(r + 1) * 8 + swz8(r + 1, pg)] = ERD_T4(ra, y);
            Ad[(r + 2) * 8 + swz8(r + 2, pg)] = ERD_T4(ra, z);
            Ad[(r + 3) * 8 + swz8(r + 3, pg)] = ERD_T4(ra, w);
        }
        if constexpr (XB) {
            Bd[(r + 0) * 8 + swz8(r + 0, pg)] = ERD_P4(ub, x, LO);
            Bd[(r + 1) * 8 + swz8(r + 1, pg)] = ERD_P4(ub, x, HI);
            Bd[(r + 2) * 8 + swz8(r + 2, pg)] = ERD_P4(ub, y, LO);
            Bd[(r + 3) * 8 + swz8(r + 3, pg)] = ERD_P4(ub, y, HI);
        } else {
            Bd[(r + 0) * 8 + swz8(r + 0, pg)] = ERD_T4(rb, x);
            Bd[(r + 1) * 8 + swz8(r + 1, pg)] = ERD_T4(rb, y);
            Bd[(r + 2) * 8 + swz8(r + 2, pg)] = ERD_T4(rb, z);
            Bd[(r + 3) * 8 + swz8(r + 3, pg)] = ERD_T4(rb, w);
        }
#undef ERD_P4
#undef ERD_T4
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt_begin < kt_end) {
        compute_offsets(kt_begin, 0);
        __syncthreads();
        load_global(0);
        compute_offsets(kt_begin + 1, 1);
        store_lds(0);
        __syncthreads();
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int buf = (kt - kt_begin) & 1;
            const bool more = kt + 1 < kt_end;
            if (more) load_global(buf ^ 1);
            compute_offsets(kt + 2, buf);
            const float4* Ab = As + buf * BM * 8;
            const float4* Bb = Bs + buf * BN * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = 2 * ks + h;
                float4 fa[FM], fb[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int row = (wm * FM + i) * 32 + li;
                    fa[i] = Ab[row * 8 + swz8(row, c)];
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int row = (wn * FN + j) * 32 + li;
                    fb[j] = Bb[row * 8 + swz8(row, c)];
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
            }
            if (more) store_lds(buf ^ 1);
            __syncthreads();
        }
    }
    float* __restrict__ part = p.part + (int64_t)bz * p.Cout * p.ntaps * p.Cin;
    if ((p.Cin & 3) == 0) {
        // partial slab through the (now idle) operand buffers, 64 accumulator rows at a time, written as 16-byte pieces of whole slab
        // rows instead of 64 four-byte stores per lane (conv_wgrad_row3_x3_kernel does the same)
        constexpr int SLDW = BN + 4, RPASS = 64, C4 = BN / 4;
        static_assert(RPASS * SLDW * 4 <= 2 * (BM + BN) * 8 * 16 && FM * 32 == RPASS, "slab staging");
        float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int ps = 0; ps < BM / RPASS; ++ps) {
            __syncthreads();
            if (wm == ps) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            stage[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * SLDW + (wn * FN + j) * 32 + li] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < RPASS * C4 / NTHREADS; ++q) {
                const int idx = q * NTHREADS + tid;
                const int row = idx / C4, c4 = idx - row * C4;
                const int co = co0 + ps * RPASS + row, ci = ci0 + 4 * c4;
                if (co < p.Cout && ci < p.Cin)
                    *reinterpret_cast<float4*>(part + ((int64_t)co * p.ntaps + tap) * p.Cin + ci) =
                        *reinterpret_cast<const float4*>(stage + row * SLDW + 4 * c4);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int ci = ci0 + (wn * FN + j) * 32 + li;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wm * FM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < p.Cout) part[((int64_t)co * p.ntaps + tap) * p.Cin + ci] = acc[i][j][r];
            }
    }
}

// dW[co][k] (+)= rowscale[co] * sum_s part[s][co][k];  rowdot[co] += sum_k w[co][k] * G[co][k]
// grid (Cout, K/1024): every thread sums one float4 column of the nsplit slabs (coalesced across the block);
// rowdot is accumulated with one atomic per block (the caller zeroes it).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, int Cout,
                                                            int K, const float* __restrict__ w,
                                                            const float* __restrict__ rowscale,
                                                            float* __restrict__ dW, int accumulate,
                                                            float* __restrict__ rowdot) {
    const int co = blockIdx.x;
    const int k = (blockIdx.y * 256 + threadIdx.x) * 4;
    const int64_t slab = (int64_t)Cout * K;
    const float rs = rowscale ? rowscale[co] : 1.f;
    float dot = 0.f;
    if (k < K) {
        const int64_t o = (int64_t)co * K + k;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        int s = 0;
        for (; s + 3 < nsplit; s += 4) {
            const float4 v0 = *reinterpret_cast<const float4*>(part + (s + 0) * slab + o);
            const float4 v1 = *reinterpret_cast<const float4*>(part + (s + 1) * slab + o);
            const float4 v2 = *reinterpret_cast<const float4*>(part + (s + 2) * slab + o);
            const float4 v3 = *reinterpret_cast<const float4*>(part + (s + 3) * slab + o);
            g.x += (v0.x + v1.x) + (v2.x + v3.x); g.y += (v0.y + v1.y) + (v2.y + v3.y);
            g.z += (v0.z + v1.z) + (v2.z + v3.z); g.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; s < nsplit; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(part + s * slab + o);
            g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
        }
        if (rowdot) {
            const float4 ww = *reinterpret_cast<const float4*>(w + o);
            dot = ww.x * g.x + ww.y * g.y + ww.z * g.z + ww.w * g.w;
        }
        float4 o4 = make_float4(rs * g.x, rs * g.y, rs * g.z, rs * g.w);
        if (accumulate) {
            const float4 old = *reinterpret_cast<const float4*>(dW + o);
            o4.x += old.x; o4.y += old.y; o4.z += old.z; o4.w += old.w;
        }
        *reinterpret_cast<float4*>(dW + o) = o4;
    }
    if (rowdot) {
        __shared__ float red[4];
        dot = erd::wave_sum(dot);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(rowdot + co, red[0] + red[1] + red[2] + red[3]);
    }
}

// dst[ci][t'][co] = rowscale[co] * w[co][t][ci]
template <typename OutT>
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* __restrict__ w,
                                                                const float* __restrict__ rowscale,
                                                                OutT* __restrict__ dst, int Cout, int ntaps, int Cin,
                                                                int flip) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int td = flip ? ntaps - 1 - t : t;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[((int64_t)co * ntaps + t) * Cin + ci] * (rowscale ? rowscale[co] : 1.f);
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (co < Cout && ci < Cin) dst[((int64_t)ci * ntaps + td) * Cout + co] = (OutT)tile[tx][r];
    }
}

using erd::limbs3;      // three bf16 limbs of an fp32 value, round-to-nearest (erd_common.h): hi + mid + lo == x exactly

// dst[plane][i] = limb `plane` of src[i]
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int64_t n) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n) return;
    unsigned short h, m, l;
    limbs3(src[i], h, m, l);
    dst[i] = h;
    dst[n + i] = m;
    dst[2 * n + i] = l;
}

// dst[plane][ci][t'][co] = limb `plane` of rowscale[co] * w[co][t][ci]
__global__ __launch_bounds__(256) void weight_transpose_x3_kernel(const float* __restrict__ w, const float* __restrict__ rowscale,
                                                                   unsigned short* __restrict__ dst, int Cout, int ntaps, int Cin,
                                                                   int flip) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int td = flip ? ntaps - 1 - t : t;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[((int64_t)co * ntaps + t) * Cin + ci] * (rowscale ? rowscale[co] : 1.f);
        tile[r][tx] = v;
    }
    __syncthreads();
    const int64_t plane = (int64_t)Cin * ntaps * Cout;
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (co < Cout && ci < Cin) {
            unsigned short h, m, l;
            limbs3(tile[tx][r], h, m, l);
            const int64_t o = ((int64_t)ci * ntaps + td) * Cout + co;
            dst[o] = h;
            dst[plane + o] = m;
            dst[2 * plane + o] = l;
        }
    }
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int BM, int BN, int WM, int WN, int BKT, int MINW, bool BF = false, bool ST = false, bool AB = false, bool OB = false,
          bool X3 = false, bool RES = false, bool MSK = false, bool GL = false>
int launch_igemm(const erd_conv_desc* d, hipStream_t st) {
    constexpr int NT = WM * WN * 64;
    constexpr int BK = (BKT / 4) * (BF ? 8 : 4);
    int tiles = 0;
    for (int s = 0; s < d->nseg; ++s) {
        const int64_t M = (int64_t)d->seg[s].N * d->seg[s].GH * d->seg[s].GW;
        tiles += (int)((M + BM - 1) / BM);
    }
    const int ntn = (d->Cout + BN - 1) / BN;
    tiles *= ntn;
    if (tiles == 0) return 0;
    const int nkt = d->ntaps * ((d->Cin + BK - 1) / BK);
    const size_t oper = X3 ? (size_t)2 * (BM * 8 + 3 * BN * 4) * sizeof(float4) : (size_t)2 * (BM + BN) * (BKT / 4) * sizeof(float4);
    const size_t stage = (size_t)64 * (BN + 4) * 4 + NT * 16;
    // (f32x3: the row table and the fix-up's broadcast word live inside the operand region -- see the kernel)
    static const size_t lds_pad = getenv("ERD_IG_LDS_PAD") ? (size_t)atoi(getenv("ERD_IG_LDS_PAD")) : 0;   // occupancy experiments
    const size_t lds = (X3 ? oper : (oper > stage ? oper : stage) + BM * sizeof(RowInfo) + 16) + lds_pad;
    auto kern = conv_igemm_kernel<BM, BN, WM, WN, BKT, MINW, BF, ST, AB, OB, X3, RES, MSK, GL>;
    static bool attr_done = false;  // idempotent, value never changes: benign race
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    // persistent stream-K grid when a workspace is supplied and tile-granular dispatch would leave a
    // ragged last round; otherwise one workgroup per tile.
    const int slots = MINW * erd::usable_cus(num_cus());
    int G = tiles;
    SkWs ws{nullptr, nullptr, xcd_order_enabled(), 0};
    // workspace layout (fixed, independent of this launch's tile count): [slabs: 2*slots*128*128 floats][tickets]
    const size_t slab_bytes = (size_t)2 * 4 * num_cus() * 128 * 128 * sizeof(float);
    const size_t need = slab_bytes + (size_t)tiles * sizeof(int);
    const bool ragged = tiles < 8 * slots && (tiles % slots) != 0;
    // short K loops (1x1 convs on few channels) are latency/HBM-bound: they want many independent workgroups
    // bf16 matrix cores: the K loop is 8x shorter, the fix-up is not -> split K only when the tiles fill at most half
    // of the resident slots (measured: whole step 105 -> 118 img/s with tile-parallel launches everywhere else;
    // 264-tile layers 64 -> 41 us without the split, 132-tile 3x3 layers 134 -> 80 us with it)
    const bool sk_pays = BF ? tiles * 2 <= slots : true;
    // tiny launches (the stride-2 P6 / P7 convolutions: 6-18 tiles with 72 K-slices each) still split K, but every
    // workgroup keeps at least `min_slices` slices: 6 workgroups walking 72 slices each took 181 us for 2 MFLOP
    static const int min_slices = getenv("ERD_SK_MIN_SLICES") ? atoi(getenv("ERD_SK_MIN_SLICES")) : 8;
    const int64_t units = (int64_t)tiles * nkt;
    const bool tiny = units < slots && min_slices > 0 && nkt >= 2 * min_slices && !BF;
    bool seg_taps = false;           // per-segment tap sets: tiles have different K lengths -> whole tiles only,
    for (int s = 0; s < d->nseg; ++s) seg_taps |= d->seg[s].ntaps > 0;     // dealt round-robin over the XCDs (a contiguous
    if (seg_taps) ws.xcd_order = 0;  // chunk per XCD would hand one XCD all the 4-tap tiles and another all the 1-tap ones)
    static const int sk_min_k = getenv("ERD_SK_MIN_K") ? atoi(getenv("ERD_SK_MIN_K")) : 512;
    if (d->sk_ws && d->sk_ws_bytes >= need && ragged && sk_pays && !seg_taps && nkt * BKT >= sk_min_k && (units >= slots || tiny)) {
        G = (int)std::min<int64_t>(slots, min_slices > 0 ? std::max<int64_t>(tiles, units / min_slices) : slots);
        ws.slabs = reinterpret_cast<float*>(d->sk_ws);
        ws.cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(d->sk_ws) + slab_bytes);
        // tickets are zero on entry: the workspace is zero-initialised by its owner and every reducer re-zeroes
        // the ticket it consumed
    }
    // persistent whole tiles (experiment, ERD_PT=1): launches that stay tile-parallel and have more tiles than resident slots
    static const int pt = getenv("ERD_PT") ? atoi(getenv("ERD_PT")) : 0;
    if (pt && !ws.slabs && !seg_taps && tiles > slots) {
        G = slots;
        ws.whole_tiles = 1;
    }
    hipLaunchKernelGGL(kern, dim3(G), dim3(NT), lds, st, *d, tiles, ws);
    return erd::check_launch("conv_igemm");
}

}  // namespace

extern "C" int erd_conv_igemm(const erd_conv_desc* d, erd_stream_t stream) {
    ERD_REQUIRE(d != nullptr, "conv: null desc");
    ERD_REQUIRE(d->nseg >= 1 && d->nseg <= ERD_MAX_SEG, "conv: nseg=%d", d->nseg);
    ERD_REQUIRE(d->ntaps >= 1 && d->ntaps <= ERD_MAX_TAPS, "conv: ntaps=%d", d->ntaps);
    ERD_REQUIRE(d->Cin > 0 && d->Cin % 4 == 0, "conv: Cin=%d must be a multiple of 4", d->Cin);
    ERD_REQUIRE(d->Cout > 0 && d->wrow % 4 == 0, "conv: Cout=%d wrow=%d", d->Cout, d->wrow);
    ERD_REQUIRE(d->w_bf16 || d->w || d->w_x3, "conv: no weights");
    ERD_REQUIRE(!(d->w_x3 && d->w_bf16), "conv: w_x3 and w_bf16 are exclusive");
    ERD_REQUIRE(d->w_bf16 || !(d->in_bf16 || d->out_bf16), "conv: bf16 maps need the bf16 matrix-core mode (w_bf16)");
    ERD_REQUIRE(!d->in_bf16 || d->Cin % 8 == 0, "conv: bf16 input maps need Cin %% 8 == 0 (Cin=%d)", d->Cin);
    ERD_REQUIRE(d->colsum_copies >= 0 && (d->colsum_copies & (d->colsum_copies - 1)) == 0, "conv: colsum_copies=%d must be a power of two",
                d->colsum_copies);
    for (int s = 0; s < d->nseg; ++s) {
        const erd_conv_seg& g = d->seg[s];
        ERD_REQUIRE(g.in && g.out, "conv: null tensor in segment %d", s);
        ERD_REQUIRE(g.ntaps >= 0 && g.tap0 >= 0 && g.tap0 + g.ntaps <= ERD_MAX_TAPS && g.ntaps <= d->ntaps,
                    "conv: segment %d tap set [%d, %d) (launch ntaps %d)", s, g.tap0, g.tap0 + g.ntaps, d->ntaps);
        // (the three-limb kernels also STORE through 32-bit buffer byte offsets: their output maps must stay below 2 GiB as well)
        ERD_REQUIRE((int64_t)g.N * g.in_nstride < (1ll << 29) && (int64_t)g.N * g.out_nstride < (d->w_x3 ? (1ll << 29) : (1ll << 31)),
                    "conv: segment %d too large (input%s must stay below 2 GiB: 32-bit buffer byte offsets)", s, d->w_x3 ? " and output" : "");
    }
    hipStream_t st = (hipStream_t)stream;
    // Variant choice (measured, tools/bench_conv.py): long K loops are MFMA-bound and want the BK=32 / stream-K
    // kernel; short ones (1x1 convs on <=256 channels) are prologue/epilogue-latency bound and want many small
    // co-resident workgroups (BK=16, half the LDS and staging registers -> 4 workgroups per CU).
    static const int variant = getenv("ERD_IGEMM_VARIANT") ? atoi(getenv("ERD_IGEMM_VARIANT")) : 0;   // tuning aid
    bool seg_taps_any = false;
    for (int s = 0; s < d->nseg; ++s) seg_taps_any |= d->seg[s].ntaps > 0;
    if (d->w_bf16 && erd::conv_thin_bf16_ok(d)) return erd::conv_thin_bf16(d, st);      // thin 1x1 layers on bf16 maps: activations stationary
    if (d->w_bf16) {    // bf16 matrix cores: the loaders, not the MFMAs, set the pace -> the plain 2-workgroup variant
        // storage of the maps (in_bf16 / out_bf16) picks the instantiation: 0 = fp32 in HBM, 1 = bf16 in HBM
#define ERD_BF_LAUNCH(AB_, OB_)                                                                       \
        do {                                                                                          \
            if (seg_taps_any) return launch_igemm<128, 128, 2, 2, 32, 2, true, true, AB_, OB_>(d, st);  \
            if (variant == 2) return launch_igemm<128, 128, 2, 2, 16, 4, true, false, AB_, OB_>(d, st); \
            if (d->Cout <= 64) return launch_igemm<128, 64, 2, 2, 32, 2, true, false, AB_, OB_>(d, st); \
            return launch_igemm<128, 128, 2, 2, 32, 2, true, false, AB_, OB_>(d, st);                   \
        } while (0)
        // (256 x 128 tiles on eight waves, one workgroup per CU -- launch_igemm<256, 128, 4, 2, 32, 1, true, ...>, 25 % fewer operand
        //  bytes per flop -- measured 6 % SLOWER than two co-resident 128 x 128 workgroups: DESIGN 7a)
        if (d->in_bf16 && d->out_bf16) ERD_BF_LAUNCH(true, true);
        if (d->in_bf16) ERD_BF_LAUNCH(true, false);
        if (d->out_bf16) ERD_BF_LAUNCH(false, true);
        ERD_BF_LAUNCH(false, false);
#undef ERD_BF_LAUNCH
    }
    // (Cin % 4 == 0 suffices: an 8-value weight chunk that straddles the end of a tap's channels meets activation zeros there --
    //  the fp32 activation chunks are 4 wide and zero-filled past Cin -- and 16-byte buffer loads need dword alignment only)
    // short K loops (K = taps x Cin below ERD_X3_MIN_K) gain nothing from a faster loop: they are set-up / epilogue bound and
    // want the fp32 kernel's four small workgroups per CU (the three-limb kernel holds 80 KB of LDS: two per CU)
    // thin 1x1 layers (Cin <= 128): activations stationary in registers, weights streamed (conv_thin.hip; bit-identical results)
    if (erd::conv_thin_x3_ok(d)) return erd::conv_thin_x3(d, st);
    static const int x3_min_k = getenv("ERD_X3_MIN_K") ? atoi(getenv("ERD_X3_MIN_K")) : 0;
    if (d->w_x3 && d->Cin % 4 == 0 && d->wrow % 2 == 0 && (d->Cout % 4 == 0 || !d->w) && (d->ntaps * d->Cin >= x3_min_k || !d->w)) {
        ERD_REQUIRE(d->Cout % 4 == 0, "conv: the three-limb kernel stores 16-byte rows (Cout %% 4 == 0); pass `w` for Cout=%d", d->Cout);
        // "f32x3": fp32 maps and results, products on the bf16 matrix cores through exact three-limb splits (see the kernel)
        // the epilogue is instantiated per (residual, mask) presence: its loads are unconditional and sit in front of its stores
        bool any_res = false, any_msk = false;
        for (int s = 0; s < d->nseg; ++s) { any_res |= d->seg[s].res != nullptr; any_msk |= d->seg[s].mask != nullptr; }
#define ERD_X3_EPI(BN_, ST_, GL_)                                                                                                  \
        (any_res ? (any_msk ? launch_igemm<128, BN_, 4, 1, 32, 2, false, ST_, false, false, true, true, true, GL_>(d, st)         \
                            : launch_igemm<128, BN_, 4, 1, 32, 2, false, ST_, false, false, true, true, false, GL_>(d, st))       \
                 : (any_msk ? launch_igemm<128, BN_, 4, 1, 32, 2, false, ST_, false, false, true, false, true, GL_>(d, st)        \
                            : launch_igemm<128, BN_, 4, 1, 32, 2, false, ST_, false, false, true, false, false, GL_>(d, st)))
        // operand slices by LDS-DMA (GL; ERD_IG_GLDS=0: through registers + ds_write, the form of rounds 3-5; bit-identical results)
        const char* const e_glds = getenv("ERD_IG_GLDS");      // (read per launch: tests/test_gpu_f32x3.py flips it inside one process)
        const int glds = e_glds ? atoi(e_glds) : 1;
        if (glds) {
            if (seg_taps_any) return ERD_X3_EPI(128, true, true);
            if (d->Cout <= 64) return ERD_X3_EPI(64, false, true);
            return ERD_X3_EPI(128, false, true);
        }
        if (seg_taps_any) return ERD_X3_EPI(128, true, false);
        if (d->Cout <= 64) return ERD_X3_EPI(64, false, false);
        return ERD_X3_EPI(128, false, false);
#undef ERD_X3_EPI
    }
    ERD_REQUIRE(d->w, "conv: this launch needs the fp32 weights (w_x3 serves Cin %% 4 == 0 only)");
    if (seg_taps_any)   // per-segment tap sets (merged parity classes of a stride-2 input gradient): a dedicated instantiation
        return launch_igemm<128, 128, 2, 2, 32, 2, false, true>(d, st);
    if (variant == 9) return launch_igemm<128, 128, 2, 2, 32, 1>(d, st);
    if (variant == 1) return launch_igemm<128, 128, 2, 2, 16, 3>(d, st);
    if (variant == 2) return launch_igemm<128, 128, 2, 2, 16, 4>(d, st);
    if (d->Cout <= 64) return launch_igemm<128, 64, 2, 2, 32, 2>(d, st);
    int64_t mtiles = 0;
    for (int s = 0; s < d->nseg; ++s) mtiles += ((int64_t)d->seg[s].N * d->seg[s].GH * d->seg[s].GW + 127) / 128;
    const int64_t tiles = mtiles * ((d->Cout + 127) / 128);
    // short K loops, or so many tiles that tile-granular dispatch is already balanced: four small workgroups per
    // CU hide each other's staging/barrier phases best (measured 136 vs 131 TF on 8192 tiles x K=2048)
    // K = 256 onto >= 512 output channels (layer3's expanding 1x1 convolutions and the input gradients of its reducing ones):
    // 1056 tiles of 128x128 run as one full dispatch round plus a 37 % full one whose workgroups do not run faster for
    // being alone; 128x64 tiles halve that tail (121.7 -> 107.4 us per launch, +0.9 % on the step; K = 64 / 128 measured
    // equal or worse).  ERD_IGEMM_SHORTK=0: off (A/B aid).
    static const int force = getenv("ERD_IG_FORCE") ? atoi(getenv("ERD_IG_FORCE")) : 0;      // experiment: one variant for every short-K launch
    if (force && d->ntaps * d->Cin <= 256) {
        if (force == 1) return launch_igemm<128, 128, 2, 2, 32, 2>(d, st);
        if (force == 2) return launch_igemm<128, 128, 2, 2, 16, 3>(d, st);
        if (force == 3) return launch_igemm<128, 128, 2, 2, 16, 4>(d, st);
        if (force == 4) return launch_igemm<128, 64, 2, 2, 32, 2>(d, st);
        if (force == 5) return launch_igemm<128, 64, 2, 2, 16, 4>(d, st);
        if (force == 6) return launch_igemm<128, 64, 2, 2, 32, 3>(d, st);
    }
    static const int shortk = getenv("ERD_IGEMM_SHORTK") ? atoi(getenv("ERD_IGEMM_SHORTK")) : 1;
    if (shortk && d->ntaps * d->Cin == 256 && d->Cout >= 512) return launch_igemm<128, 64, 2, 2, 32, 2>(d, st);
    if (d->ntaps * d->Cin <= 256) {
        // Tile-parallel launches finish in whole dispatch rounds: pick the co-residency whose LAST round is fullest.
        // cost = rounds x (workgroups per CU / steady-state efficiency at that co-residency: 0.83 / 0.85 / 0.87 measured)
        static const int pick = getenv("ERD_IGEMM_PICK") ? atoi(getenv("ERD_IGEMM_PICK")) : 1;   // 0: always four per CU
        const int64_t cus = erd::usable_cus(num_cus());
        const double c2 = (double)((tiles + 2 * cus - 1) / (2 * cus)) * (2.0 / 0.83);
        const double c3 = (double)((tiles + 3 * cus - 1) / (3 * cus)) * (3.0 / 0.85);
        const double c4 = (double)((tiles + 4 * cus - 1) / (4 * cus)) * (4.0 / 0.87);
        if (pick && c3 < c4 && c3 <= c2) return launch_igemm<128, 128, 2, 2, 16, 3>(d, st);
        if (pick && c2 < c4 && c2 < c3) return launch_igemm<128, 128, 2, 2, 32, 2>(d, st);
        return launch_igemm<128, 128, 2, 2, 16, 4>(d, st);
    }
    if (tiles >= 16 * 2 * num_cus()) return launch_igemm<128, 128, 2, 2, 16, 4>(d, st);
    return launch_igemm<128, 128, 2, 2, 32, 2>(d, st);
}

namespace {
template <int BKW, int MINW>
int launch_wgrad(const erd_wgrad_desc* d, hipStream_t st) {
    constexpr int BM = 128, BN = 128;
    const int nci = (d->Cin + BN - 1) / BN, nco = (d->Cout + BM - 1) / BM;
    const size_t lds = (size_t)2 * BKW * (BM + BN) * sizeof(float) + 2 * BKW * sizeof(int2);
    auto kern = conv_wgrad_kernel<BM, BN, BKW, MINW>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    // XCD-aware work order for the split-K siblings too (same speed, fabric traffic 321 -> 178 MB per launch; ERD_XCD=0: off)
    hipLaunchKernelGGL(kern, dim3(nci * d->ntaps * nco * d->nsplit), dim3(NTHREADS), lds, st, *d, xcd_order_enabled() ? 1 : 0);
    return erd::check_launch("conv_wgrad");
}
}  // namespace

#ifdef ERD_IGEMM_TRACE
extern "C" int erd_igemm_trace(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_igemm_trace), sizeof(g_igemm_trace));
}
#endif

extern "C" size_t erd_conv_igemm_ws_bytes(int max_tiles) {
    return (size_t)2 * 4 * num_cus() * 128 * 128 * sizeof(float) + (size_t)max_tiles * sizeof(int);
}

// K-slices of the three-tap kernel (0: the layer is not a 3x3 / stride 1 / pad 1 convolution in tap order ky*3+kx)
extern "C" int erd_wgrad_row3_slices(const erd_wgrad_desc* d) {
    if (!d || d->bf16_multiplicands || d->ntaps != 9 || d->in_stride != 1 || d->out_stride != 1 || d->oy || d->ox) return 0;
    for (int t = 0; t < 9; ++t)
        if (d->dy[t] != t / 3 - 1 || d->dx[t] != t % 3 - 1) return 0;
    int64_t n = 0;
    for (int l = 0; l < d->nseg; ++l) {
        const erd_wgrad_seg& g = d->seg[l];
        if (g.GH != g.IH || g.GW != g.IW || g.OH != g.GH || g.OW != g.GW) return 0;
        n += (int64_t)g.N * g.GH * ((g.GW + 15) / 16);
    }
    return n < (1ll << 30) ? (int)n : 0;
}

extern "C" int erd_conv_wgrad(const erd_wgrad_desc* d, erd_stream_t stream) {
    ERD_REQUIRE(d != nullptr && d->x && d->dz && d->part, "wgrad: null pointer");
    ERD_REQUIRE(d->ntaps >= 1 && d->ntaps <= ERD_MAX_TAPS, "wgrad: ntaps=%d", d->ntaps);
    ERD_REQUIRE(d->Cin % 4 == 0 && d->Cout % 4 == 0, "wgrad: Cin=%d Cout=%d must be multiples of 4", d->Cin, d->Cout);
    ERD_REQUIRE(d->nsplit >= 1 && d->nsplit <= 65535, "wgrad: nsplit=%d", d->nsplit);
    ERD_REQUIRE(d->nseg >= 1 && d->nseg <= ERD_MAX_SEG, "wgrad: nseg=%d", d->nseg);
    int64_t npix = 0;
    for (int l = 0; l < d->nseg; ++l) npix += (int64_t)d->seg[l].N * d->seg[l].GH * d->seg[l].GW;
    ERD_REQUIRE(npix < (1ll << 31), "wgrad: too many pixels");
    ERD_REQUIRE(d->x_elems > 0 && d->dz_elems > 0 && d->x_elems < (1ll << 29) && d->dz_elems < (1ll << 29),
                "wgrad: tensors must stay below 2 GiB (32-bit buffer byte offsets)");
    // K-slices of 16 pixels: half the LDS / staging registers of a 32-pixel slice -> four workgroups per CU hide each
    // other's staging and barrier phases (measured +10 % over 32-pixel slices at two per CU, tools/bench_conv.py)
    ERD_REQUIRE(d->bf16_multiplicands || !(d->x_bf16 || d->dz_bf16), "wgrad: bf16-stored maps need bf16_multiplicands");
    if (d->bf16_multiplicands) {
        const int nci = (d->Cin + 127) / 128, nco = (d->Cout + 127) / 128;
        const size_t lds = (size_t)2 * (128 + 128) * 8 * sizeof(float4) + 2 * 64 * sizeof(int2);
        const dim3 grid(nci * d->ntaps * nco * d->nsplit);
#define ERD_WG_LAUNCH(XB_, DB_)                                                                                        \
        do {                                                                                                           \
            static bool attr_done = false;                                                                             \
            if (!attr_done) {                                                                                          \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16_kernel<XB_, DB_>),             \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
                attr_done = true;                                                                                      \
            }                                                                                                          \
            hipLaunchKernelGGL((conv_wgrad_bf16_kernel<XB_, DB_>), grid, dim3(NTHREADS), lds, (hipStream_t)stream, *d); \
        } while (0)
        if (d->x_bf16 && d->dz_bf16) ERD_WG_LAUNCH(true, true);
        else if (d->x_bf16) ERD_WG_LAUNCH(true, false);
        else if (d->dz_bf16) ERD_WG_LAUNCH(false, true);
        else ERD_WG_LAUNCH(false, false);
#undef ERD_WG_LAUNCH
        return erd::check_launch("conv_wgrad_bf16");
    }
    static const int variant = getenv("ERD_WGRAD_VARIANT") ? atoi(getenv("ERD_WGRAD_VARIANT")) : 1;   // tuning aid
    const char* row3_env = getenv("ERD_WGRAD_ROW3");          // read per call: tests flip it in-process
    const int row3 = row3_env ? atoi(row3_env) : 1;
    if (row3 && d->limbs3 && d->Cin % 4 == 0 && erd_wgrad_row3_slices(d) > 0) {
        // three-limb form: 128 (Cout > 64) or 64 output channels x 64 input channels x 3 taps per workgroup
        const int nslices = erd_wgrad_row3_slices(d);
        const int fm = d->Cout > 64 ? 2 : 1;
        const int nci = (d->Cin + 63) / 64, nco = (d->Cout + fm * 64 - 1) / (fm * 64);
        const size_t lds = (size_t)2 * 3 * (fm * 64 * 32 + 64 * 48) + 2 * (16 + 18) * sizeof(int);
        void (*kern)(const erd_wgrad_desc, const int) = fm == 2 ? conv_wgrad_row3_x3_kernel<2, 1, true> : conv_wgrad_row3_x3_kernel<1, 1, true>;
        static bool attr_done3[2] = {false, false};
        if (!attr_done3[fm - 1]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_done3[fm - 1] = true;
        }
        static const int row3_xcd = getenv("ERD_WGRAD_XCD") ? atoi(getenv("ERD_WGRAD_XCD")) : 1;
        hipLaunchKernelGGL(kern, dim3(nci * 3 * nco * d->nsplit), dim3(NTHREADS), lds, (hipStream_t)stream, *d,
                           nslices | (row3_xcd ? (1 << 30) : 0));
        return erd::check_launch("conv_wgrad_row3_x3");
    }
    if (d->limbs3 && d->Cin % 4 == 0) {
        // three-limb form of every other layer (1x1, stride 2): one tap, 128 x 128 channels per workgroup, 16-pixel slices of the
        // concatenated pixel axis
        const int nslices = (int)((npix + 15) / 16);
        const int nci = (d->Cin + 127) / 128, nco = (d->Cout + 127) / 128;
        const size_t lds = (size_t)2 * 3 * (128 * 32 + 128 * 32) + 2 * (16 + 16) * sizeof(int);
        void (*kern)(const erd_wgrad_desc, const int) = conv_wgrad_row3_x3_kernel<2, 2, false>;
        static bool attr_done1 = false;
        if (!attr_done1) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_done1 = true;
        }
        hipLaunchKernelGGL(kern, dim3(nci * d->ntaps * nco * d->nsplit), dim3(NTHREADS), lds, (hipStream_t)stream, *d,
                           nslices | (xcd_order_enabled() ? (1 << 30) : 0));
        return erd::check_launch("conv_wgrad_x3");
    }
    if (row3 && erd_wgrad_row3_slices(d) > 0) {
        const int nslices = erd_wgrad_row3_slices(d);
        const int bme = d->Cout <= 64 ? 64 : (d->Cout <= 96 ? 96 : 128);
        const int nci = (d->Cin + 127) / 128, nco = (d->Cout + bme - 1) / bme;
        const size_t lds = (size_t)2 * (16 * 128 + 18 * 128) * sizeof(float) + 2 * (16 + 18) * sizeof(int);
        void (*kern)(const erd_wgrad_desc, const int) =
            bme == 64 ? conv_wgrad_row3_kernel<1, 2, 1> : (bme == 96 ? conv_wgrad_row3_kernel<1, 3, 1> : conv_wgrad_row3_kernel<2, 2, 2>);
        static bool attr_done[3] = {false, false, false};
        const int vi = bme == 64 ? 0 : (bme == 96 ? 1 : 2);
        if (!attr_done[vi]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_done[vi] = true;
        }
        // XCD-aware work order (ERD_WGRAD_XCD=0: dispatch order): same speed, fabric traffic 647 -> 239 MB per launch (PMC)
        static const int row3_xcd = getenv("ERD_WGRAD_XCD") ? atoi(getenv("ERD_WGRAD_XCD")) : 1;
        hipLaunchKernelGGL(kern, dim3(nci * 3 * nco * d->nsplit), dim3(NTHREADS), lds, (hipStream_t)stream, *d,
                           nslices | (row3_xcd ? (1 << 30) : 0));
        return erd::check_launch("conv_wgrad_row3");
    }
    if (variant == 0) return launch_wgrad<32, 2>(d, (hipStream_t)stream);
    return launch_wgrad<16, 4>(d, (hipStream_t)stream);
}

extern "C" int erd_wgrad_reduce(const float* part, int nsplit, int Cout, int K, const float* w,
                                const float* rowscale, float* dW, int accumulate, float* rowdot,
                                erd_stream_t stream) {
    ERD_REQUIRE(part && dW && nsplit >= 1 && K % 4 == 0, "wgrad_reduce: bad args");
    ERD_REQUIRE(!rowdot || w, "wgrad_reduce: rowdot needs w");
    // accumulate: bit 0 = add into dW, bit 1 = rowdot is already zero (comes from the caller's zero arena: no memset)
    if (rowdot && !(accumulate & 2)) (void)hipMemsetAsync(rowdot, 0, sizeof(float) * Cout, (hipStream_t)stream);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(Cout, (K + 1023) / 1024), dim3(256), 0, (hipStream_t)stream, part,
                       nsplit, Cout, K, w, rowscale, dW, accumulate & 1, rowdot);
    return erd::check_launch("wgrad_reduce");
}

extern "C" int erd_weight_transpose(const float* w, const float* rowscale, float* dst, int Cout, int ntaps,
                                    int Cin, int flip, erd_stream_t stream) {
    ERD_REQUIRE(w && dst && Cout > 0 && Cin > 0 && ntaps > 0, "weight_transpose: bad args");
    hipLaunchKernelGGL(weight_transpose_kernel<float>, dim3((Cin + 31) / 32, (Cout + 31) / 32, ntaps), dim3(256), 0,
                       (hipStream_t)stream, w, rowscale, dst, Cout, ntaps, Cin, flip);
    return erd::check_launch("weight_transpose");
}

extern "C" int erd_split3(const float* src, void* dst, int64_t n, erd_stream_t stream) {
    ERD_REQUIRE(src && dst && n > 0, "split3: bad args");
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<unsigned short*>(dst), n);
    return erd::check_launch("split3");
}

extern "C" int erd_weight_transpose_x3(const float* w, const float* rowscale, void* dst, int Cout, int ntaps, int Cin, int flip,
                                       erd_stream_t stream) {
    ERD_REQUIRE(w && dst && Cout > 0 && Cin > 0 && ntaps > 0, "weight_transpose_x3: bad args");
    hipLaunchKernelGGL(weight_transpose_x3_kernel, dim3((Cin + 31) / 32, (Cout + 31) / 32, ntaps), dim3(256), 0, (hipStream_t)stream, w,
                       rowscale, reinterpret_cast<unsigned short*>(dst), Cout, ntaps, Cin, flip);
    return erd::check_launch("weight_transpose_x3");
}

extern "C" int erd_weight_transpose_bf16(const float* w, const float* rowscale, void* dst, int Cout, int ntaps,
                                         int Cin, int flip, erd_stream_t stream) {
    ERD_REQUIRE(w && dst && Cout > 0 && Cin > 0 && ntaps > 0, "weight_transpose_bf16: bad args");
    hipLaunchKernelGGL(weight_transpose_kernel<__bf16>, dim3((Cin + 31) / 32, (Cout + 31) / 32, ntaps), dim3(256), 0,
                       (hipStream_t)stream, w, rowscale, reinterpret_cast<__bf16*>(dst), Cout, ntaps, Cin, flip);
    return erd::check_launch("weight_transpose_bf16");
}
