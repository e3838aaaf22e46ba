// tools/store_pattern.hip -- what the WRITE side of a GEMM-shaped grid costs on its own: no loads, no MFMAs, only the stores of
// conv_thin_x3_kernel's epilogue in its order, next to other orders of the same bytes.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
//
// The output map is [M rows][C floats] (NHWC, C = Cout); a persistent grid of G workgroups x 4 waves walks (128-row tile, 32-cout
// block) pairs in tile-major order, the thin-K kernel's work split:
//   mode 0  reference: the same bytes as ONE contiguous stream per workgroup (16 B per lane, 1 KB per wave instruction)
//   mode 1  the shipped epilogue: per (tile, block) a wave stores 32 rows x 128 B as 4 instructions of 8 rows x 128 B (dwordx4)
//   mode 2  the direct epilogue: 16 dword instructions of 2 rows x 128 B (accumulator layout of the 32x32 MFMA)
//   mode 3  four blocks back to back: 32 rows x 512 B as 16 dwordx4 instructions of 2 rows x 512 B
//   mode 4  whole rows: a wave instruction = one row of C floats (C = 256) / half a row (C = 512); a tile's rows in order
// Output: microseconds, TB/s, bytes per clock and CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ out, int M, int C, long long* clk) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int G = gridDim.x, wg = blockIdx.x;
    const int mtiles = (M + 127) / 128, nb = C / 32;
    const long long T = (long long)mtiles * nb, t0 = T * wg / G, t1 = T * (wg + 1) / G;
    const long long c0 = clock64();
    const float4 v = make_float4((float)tid, 1.f, 2.f, 3.f);
    if (MODE == 0) {
        // the workgroup's share of the bytes as one stream
        const long long bytes = (long long)M * C * 4, b0 = bytes * wg / G / 4096 * 4096, b1 = (wg + 1 == G) ? bytes : bytes * (wg + 1) / G / 4096 * 4096;
        for (long long b = b0 + tid * 16; b < b1; b += 4096) *reinterpret_cast<float4*>(reinterpret_cast<char*>(out) + b) = v;
    } else if (MODE == 4) {
        const int per = C / 4 <= 64 ? 1 : C / 4 / 64;       // instructions per row
        for (int mt = (int)(t0 / nb); (long long)mt * nb < t1; ++mt) {
            for (int r = 0; r < 32; ++r) {
                const long long row = (long long)mt * 128 + wave * 32 + r;
                if (row >= M) break;
                for (int i = 0; i < per; ++i) *reinterpret_cast<float4*>(out + row * C + (i * 64 + lane) * 4) = v;
            }
        }
    } else {
        for (long long t = t0; t < t1; ++t) {
            const int mt = (int)(t / nb), cb = (int)(t % nb);
            const long long row0 = (long long)mt * 128 + wave * 32;
            if (MODE == 1) {
                const int c4 = lane & 7, rsub = lane >> 3;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long long row = row0 + q * 8 + rsub;
                    if (row < M) *reinterpret_cast<float4*>(out + row * C + cb * 32 + c4 * 4) = v;
                }
            } else if (MODE == 2) {
                const int li = lane & 31, h = lane >> 5;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < M) out[row * C + cb * 32 + li] = v.x;
                }
            } else if (MODE == 3) {
                if (cb & 3) continue;                        // (the block's three successors ride along)
                const int c = lane & 31, rsub = lane >> 5;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const long long row = row0 + q * 2 + rsub;
                    if (row < M) *reinterpret_cast<float4*>(out + row * C + cb * 32 + c * 4) = v;
                }
            }
        }
    }
    if (tid == 0) clk[wg] = clock64() - c0;
}

template <int MODE>
void run(const char* name, float* out, int M, int C, int G, long long* clk) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(G), dim3(256), 0, 0, out, M, C, clk);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms / 10 < best) best = ms / 10;
    }
    static long long h[4096];
    CK(hipMemcpy(h, clk, sizeof(long long) * G, hipMemcpyDeviceToHost));
    double cyc = 0; for (int i = 0; i < G; ++i) cyc += (double)h[i]; cyc /= G;
    const double bytes = (double)M * C * 4;
    printf("  %-34s G %4d: %7.1f us  %5.2f TB/s  (%.0f s_memtime ticks per workgroup)\n", name, G, best * 1e3, bytes / best / 1e9, cyc);
}

int main() {
    float* out; long long* clk;
    const size_t cap = (size_t)300 << 20;
    CK(hipMalloc(&out, cap)); CK(hipMalloc(&clk, sizeof(long long) * 4096));
    const int shapes[3][2] = {{4 * 200 * 336, 256}, {4 * 100 * 168, 512}, {4 * 200 * 336, 64}};
    for (auto& s : shapes) {
        const int M = s[0], C = s[1];
        printf("[%d rows][%d floats] = %.0f MB\n", M, C, (double)M * C * 4 / 1e6);
        for (int G : {512, 768, 1024, 2048}) {
            run<0>("0 contiguous stream", out, M, C, G, clk);
            run<1>("1 shipped: 8 rows x 128 B, dwordx4", out, M, C, G, clk);
            run<2>("2 direct: 2 rows x 128 B, dword", out, M, C, G, clk);
            if (C >= 128) run<3>("3 four blocks: 2 rows x 512 B", out, M, C, G, clk);
            run<4>("4 whole rows", out, M, C, G, clk);
        }
    }
    CK(hipMemset(out, 0, cap));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a)); for (int i = 0; i < 10; ++i) CK(hipMemsetAsync(out, 0, 275251200, 0)); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("hipMemset of 275 MB: %.1f us  %.2f TB/s\n", ms * 100, 275.2512 / (ms / 10) / 1e3);
    return 0;
}
