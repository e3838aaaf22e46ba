// Sustained matrix-core rate and the clock the chip holds while doing it (evidence for DESIGN.md's roofline peaks).
//   hipcc --offload-arch=gfx950 -O3 -w tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Every CU runs WG workgroups of 4 waves; each wave issues back-to-back MFMAs on 4 independent accumulators (no memory
// traffic).  Reported per variant: TFLOP/s from hipEvent time, and the shader clock = s_memtime ticks / s_memrealtime
// (100 MHz constant-rate) ticks measured inside the kernel by one wave per workgroup (median over workgroups).
// `random operands` variants: the same loops on pseudo-random operand registers (four different A and four different B fragments
// per lane, visited round-robin, every bit position toggling) -- constant operands flatter the pipe: the chip's power
// management lowers the clock under the switching activity of real data, and that sustained rate, not the constant-operand
// one, is what a GEMM on real maps can reach.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* sink, unsigned long long* ticks, int iters) {
    f32x16 acc[KIND == 1 ? 1 : NACC];
    f32x4 acc4[KIND == 1 ? NACC : 1];
    for (int q = 0; q < (KIND == 1 ? 1 : NACC); ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int q = 0; q < (KIND == 1 ? NACC : 1); ++q) acc4[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < 0; ++q) {
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        acc4[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float a = 1e-3f * (threadIdx.x & 7), b = 0.5f;
    bf16x8 ab, bb;
    for (int r = 0; r < 8; ++r) { ab[r] = (__bf16)a; bb[r] = (__bf16)b; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            if constexpr (KIND == 0) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
            if constexpr (KIND == 1) acc4[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[q], 0, 0, 0);
            if constexpr (KIND == 2) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[q], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int q = 0; q < (KIND == 1 ? 1 : NACC); ++q) s += acc[q][0] + acc[q][7];
    for (int q = 0; q < (KIND == 1 ? NACC : 1); ++q) s += acc4[q][0] + acc4[q][3];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = w1 - w0; }
}

// hand-placed stream (inline asm, accumulators in VGPRs): NACC independent accumulators visited round-robin, 4 different
// A operands and NACC different B operands -- the issue pattern of the Winograd kernel's MFMA waves is NACC = 4
template <int NACC, int W32>
__global__ __launch_bounds__(256) void mfma_asm_loop(float* sink, unsigned long long* ticks, int iters) {
    f32x4 acc[NACC];
    f32x16 acc32[NACC];
    float a[4], b[NACC];
    for (int q = 0; q < NACC; ++q) {
        acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < 16; ++r) acc32[q][r] = 0.f;
        b[q] = 0.25f + 1e-3f * ((threadIdx.x + q) & 15);
    }
    for (int m = 0; m < 4; ++m) a[m] = 1e-3f * ((threadIdx.x * 7 + m) & 31);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < NACC; ++q) {
                if constexpr (W32) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc32[q]) : "v"(a[m]), "v"(b[q]));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a[m]), "v"(b[q]));
            }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][3] + acc32[q][5];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int NACC, int W32>
void run_asm(int wg_per_cu, int iters) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount * wg_per_cu;
    float* sink;
    unsigned long long* ticks;
    hipMalloc(&sink, 4);
    hipMalloc(&ticks, sizeof(unsigned long long) * 2 * grid);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_asm_loop<NACC, W32>), dim3(grid), dim3(256), 0, 0, sink, ticks, iters / 10);
    hipDeviceSynchronize();
    std::vector<double> tfs;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_asm_loop<NACC, W32>), dim3(grid), dim3(256), 0, 0, sink, ticks, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = W32 ? 32.0 * 32 * 2 * 2 : 16.0 * 16 * 4 * 2;
        tfs.push_back((double)grid * 4 * 4.0 * NACC * iters * flop / (ms * 1e-3) / 1e12);
    }
    std::sort(tfs.begin(), tfs.end());
    printf("asm stream %-24s %2d accumulators round-robin (VGPR), %d WG/CU x 4 waves: %8.1f TFLOP/s (median of 5; min %.1f max %.1f)\n",
           W32 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32", NACC, wg_per_cu, tfs[2], tfs[0], tfs[4]);
    hipFree(sink);
    hipFree(ticks);
}

template <int KIND, int NACC>
void run(const char* name, double flop_per_mfma, int wg_per_cu, int iters) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount * wg_per_cu;
    float* sink;
    unsigned long long* ticks;
    hipMalloc(&sink, 4);
    hipMalloc(&ticks, sizeof(unsigned long long) * 2 * grid);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<KIND, NACC>), dim3(grid), dim3(256), 0, 0, sink, ticks, iters / 10);   // warm-up
    hipDeviceSynchronize();
    std::vector<double> tfs, clks;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<KIND, NACC>), dim3(grid), dim3(256), 0, 0, sink, ticks, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * grid);
        hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
        std::vector<double> c;
        for (int b = 0; b < grid; ++b) c.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100e6 / 1e9);
        std::sort(c.begin(), c.end());
        tfs.push_back((double)grid * 4 /*waves*/ * (double)NACC * iters * flop_per_mfma / (ms * 1e-3) / 1e12);
        clks.push_back(c[c.size() / 2]);
    }
    std::sort(tfs.begin(), tfs.end());
    std::sort(clks.begin(), clks.end());
    printf("%-26s %2d independent accumulators, %d WG/CU x 4 waves, %d MFMAs/wave: %8.1f TFLOP/s (median of 5; min %.1f max %.1f), shader clock %.3f GHz "
           "(s_memtime / s_memrealtime, median workgroup; min %.3f max %.3f over runs)\n",
           name, NACC, wg_per_cu, NACC * iters, tfs[2], tfs[0], tfs[4], clks[2], clks[0], clks[4]);
    hipFree(sink);
    hipFree(ticks);
}

// random operands: KIND 0 = v_mfma_f32_32x32x2_f32, KIND 2 = v_mfma_f32_32x32x16_bf16; NACC independent accumulators,
// 4 A x 4 B operand fragments per lane from a hash of (thread, index) in [-1, 1)
__device__ inline float hash_unit(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(int)x * (1.0f / 2147483648.0f);
}
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void mfma_random_loop(float* sink, unsigned long long* ticks, int iters, unsigned seed) {
    f32x16 acc[NACC];
    for (int q = 0; q < NACC; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const unsigned t = (blockIdx.x * 256u + threadIdx.x) * 64u + seed;
    float af[4], bf[4];
    bf16x8 ab[4], bb[4];
    for (int m = 0; m < 4; ++m) {
        af[m] = hash_unit(t + m);
        bf[m] = hash_unit(t + 8 + m);
        for (int r = 0; r < 8; ++r) {
            ab[m][r] = (__bf16)hash_unit(t + 16 + m * 8 + r);
            bb[m][r] = (__bf16)hash_unit(t + 48 + m * 8 + r);
        }
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < NACC; ++q) {
                if constexpr (KIND == 0) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bf[(m + q) & 3], acc[q], 0, 0, 0);
                if constexpr (KIND == 2) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[m], bb[(m + q) & 3], acc[q], 0, 0, 0);
            }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][7];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int KIND, int NACC>
void run_random(const char* name, double flop_per_mfma, int wg_per_cu, int iters, int reps) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount * wg_per_cu;
    float* sink;
    unsigned long long* ticks;
    hipMalloc(&sink, 4);
    hipMalloc(&ticks, sizeof(unsigned long long) * 2 * grid);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_random_loop<KIND, NACC>), dim3(grid), dim3(256), 0, 0, sink, ticks, iters / 10, 1u);
    hipDeviceSynchronize();
    std::vector<double> tfs, clks;
    for (int rep = 0; rep < reps; ++rep) {      // back to back: the later repetitions run on a chip that is already warm / throttled
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_random_loop<KIND, NACC>), dim3(grid), dim3(256), 0, 0, sink, ticks, iters, 7u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * grid);
        hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
        std::vector<double> c;
        for (int b = 0; b < grid; ++b) c.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100e6 / 1e9);
        std::sort(c.begin(), c.end());
        tfs.push_back((double)grid * 4 * 4.0 * NACC * iters * flop_per_mfma / (ms * 1e-3) / 1e12);
        clks.push_back(c[c.size() / 2]);
    }
    printf("%-26s RANDOM operands, %2d accumulators, %d WG/CU x 4 waves, %.1f ms per launch: TFLOP/s per repetition", name, NACC, wg_per_cu,
           (double)grid * 4 * 4.0 * NACC * iters * flop_per_mfma / (tfs[reps - 1] * 1e12) * 1e3);
    for (double v : tfs) printf(" %.0f", v);
    printf("; shader clock GHz");
    for (double v : clks) printf(" %.3f", v);
    std::sort(tfs.begin(), tfs.end());
    printf("; median %.1f TFLOP/s\n", tfs[reps / 2]);
    hipFree(sink);
    hipFree(ticks);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("%s, %d CUs, clockRate %.0f MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1e3);
    for (int wg = 1; wg <= 2; ++wg) {
        run_asm<2, 0>(wg, 20000);
        run_asm<4, 0>(wg, 10000);
        run_asm<8, 0>(wg, 5000);
        run_asm<16, 0>(wg, 2500);
        run_asm<2, 1>(wg, 10000);
        run_asm<4, 1>(wg, 5000);
    }
    for (int wg = 1; wg <= 2; ++wg) {
        run<0, 4>("v_mfma_f32_32x32x2_f32", 32.0 * 32 * 2 * 2, wg, 40000);
        run<0, 2>("v_mfma_f32_32x32x2_f32", 32.0 * 32 * 2 * 2, wg, 80000);
        run<1, 2>("v_mfma_f32_16x16x4_f32", 16.0 * 16 * 4 * 2, wg, 160000);
        run<1, 4>("v_mfma_f32_16x16x4_f32", 16.0 * 16 * 4 * 2, wg, 80000);
        run<1, 8>("v_mfma_f32_16x16x4_f32", 16.0 * 16 * 4 * 2, wg, 40000);
        run<1, 16>("v_mfma_f32_16x16x4_f32", 16.0 * 16 * 4 * 2, wg, 20000);
        run<2, 4>("v_mfma_f32_32x32x16_bf16", 32.0 * 32 * 16 * 2, wg, 80000);
        run<2, 8>("v_mfma_f32_32x32x16_bf16", 32.0 * 32 * 16 * 2, wg, 40000);
    }
    // sustained rates on random operands (each launch ~10-40 ms, 9 back to back)
    for (int wg = 1; wg <= 2; ++wg) {
        run_random<0, 4>("v_mfma_f32_32x32x2_f32", 32.0 * 32 * 2 * 2, wg, 20000, 9);
        run_random<2, 4>("v_mfma_f32_32x32x16_bf16", 32.0 * 32 * 16 * 2, wg, 40000, 9);
        run_random<2, 8>("v_mfma_f32_32x32x16_bf16", 32.0 * 32 * 16 * 2, wg, 20000, 9);
    }
    return 0;
}
