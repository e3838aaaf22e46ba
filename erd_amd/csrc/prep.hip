// Per-step parameter preparation in two launches (erd_weight_prep_batch): everything the step derives from the student's
// convolution weights alone -- the transposed (and BN-scaled) weights of the input-gradient convolutions, in fp32 or bf16,
// and the Winograd weight images U = G g G^T of the forward and input-gradient forms -- depends only on the parameters,
// so the trainer builds it right after the optimizer update instead of ~105 five-microsecond launches scattered over
// the forward and backward passes (58 transposes + 47 weight images per step).  The arithmetic of every item is the
// arithmetic of erd_weight_transpose(_bf16) / erd_wino_weights (tests compare bit for bit).
#include "erd_common.h"

namespace {

__device__ __forceinline__ void transpose_block(const erd_weight_prep_item& it, int local, float (*tile)[33]) {
    // grid of the single-item kernel: (ceil(Cin/32), ceil(Cout/32), ntaps), 256 threads = 32 x 8
    const int nbx = (it.Cin + 31) / 32, nby = (it.Cout + 31) / 32;
    const int bx = local % nbx, by = (local / nbx) % nby, t = local / (nbx * nby);
    const int td = it.flip ? it.ntaps - 1 - t : t;
    const int ci0 = bx * 32, co0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* __restrict__ w = it.w;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.f;
        if (co < it.Cout && ci < it.Cin) v = w[((int64_t)co * it.ntaps + t) * it.Cin + ci] * (it.rowscale ? it.rowscale[co] : 1.f);
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (co < it.Cout && ci < it.Cin) {
            const int64_t o = ((int64_t)ci * it.ntaps + td) * it.Cout + co;
            if (it.kind == 1) reinterpret_cast<__bf16*>(it.dst)[o] = (__bf16)tile[tx][r];
            else reinterpret_cast<float*>(it.dst)[o] = tile[tx][r];
        }
    }
}

__device__ __forceinline__ void wino_block(const erd_weight_prep_item& it, int local) {
    const int Cout = it.Cout, Cin = it.Cin, flip = it.flip;
    const float* __restrict__ w = it.w;
    float* __restrict__ U = reinterpret_cast<float*>(it.dst);
    const int cop = (Cout + 15) / 16 * 16;
    const int64_t idx = local * 256ll + threadIdx.x;
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

// kind 3: the three bf16 limb planes of w (n = Cout * ntaps * Cin values, any layout): dst[plane][i], arithmetic of erd_split3
__device__ __forceinline__ void split3_block(const erd_weight_prep_item& it, int local) {
    const int64_t n = (int64_t)it.Cout * it.ntaps * it.Cin;
    const int64_t i = local * 256ll + threadIdx.x;
    if (i >= n) return;
    unsigned short h, m, l;
    erd::limbs3(it.w[i], h, m, l);
    unsigned short* d = reinterpret_cast<unsigned short*>(it.dst);
    d[i] = h;
    d[n + i] = m;
    d[2 * n + i] = l;
}

// kind 4: the Winograd weight image in the three-limb layout (erd_wino_weights_x3): one thread per (co < ceil32(Cout), ci)
__device__ __forceinline__ void wino_x3_block(const erd_weight_prep_item& it, int local) {
    const int cop = (it.Cout + 31) / 32 * 32;
    const int64_t idx = local * 256ll + threadIdx.x;
    if (idx >= (int64_t)cop * it.Cin) return;
    erd::wino_x3_weight_item(it.w, reinterpret_cast<unsigned short*>(it.dst), it.Cout, it.Cin, it.flip, (int)(idx / it.Cin), (int)(idx % it.Cin));
}

__global__ __launch_bounds__(256) void weight_prep_kernel(const erd_weight_prep_item* __restrict__ items, int nitems) {
    __shared__ float tile[32][33];
    __shared__ int s_item;
    if (threadIdx.x == 0) {          // items are sorted by block0: the last one that starts at or before this block
        int lo = 0, hi = nitems - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (items[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        s_item = lo;
    }
    __syncthreads();
    const erd_weight_prep_item it = items[s_item];
    const int local = (int)blockIdx.x - it.block0;
    if (it.kind == 2) wino_block(it, local);
    else if (it.kind == 3) split3_block(it, local);
    else if (it.kind == 4) wino_x3_block(it, local);
    else transpose_block(it, local, tile);
}

}  // namespace

extern "C" int erd_weight_prep_blocks(int kind, int Cout, int ntaps, int Cin) {
    if (kind == 2) return (int)(((int64_t)((Cout + 15) / 16 * 16) * Cin + 255) / 256);
    if (kind == 3) return (int)(((int64_t)Cout * ntaps * Cin + 255) / 256);
    if (kind == 4) return (int)(((int64_t)((Cout + 31) / 32 * 32) * Cin + 255) / 256);
    return ((Cin + 31) / 32) * ((Cout + 31) / 32) * ntaps;
}

extern "C" int erd_weight_prep_batch(const erd_weight_prep_item* items_dev, int nitems, int total_blocks, erd_stream_t stream) {
    ERD_REQUIRE(items_dev && nitems >= 0 && total_blocks >= 0, "weight_prep_batch: bad args");
    if (nitems == 0 || total_blocks == 0) return 0;
    hipLaunchKernelGGL(weight_prep_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, nitems);
    return erd::check_launch("weight_prep_batch");
}
