// Do VALU instructions of one wave issue while ANOTHER wave of the same SIMD runs a chain of dependent bf16 MFMAs?  (gfx950)
// Workgroups of 512 threads = 8 waves = two per SIMD (wave w -> SIMD w & 3 assumed: the table tells): waves 0-3 run `mf` dependent
// v_mfma_f32_32x32x16_bf16 per iteration (0 = idle spin on s_sleep), waves 4-7 run 64 independent-chain VALU instructions per
// iteration; every wave reports its cycles per iteration.
//   hipcc --offload-arch=gfx950 -O3 -w tools/mfma_valu_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: MFMA waves idle, 1: MFMA waves run, 2: MFMA waves run, VALU waves idle, 3: ALL eight waves run VALU, 4: all eight run MFMA
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool mf = MODE == 4 || (MODE != 3 && wave < 4);
    f32x16 acc[2];
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f + i); }
    float x[8], y[8];
    for (int i = 0; i < 8; ++i) { x[i] = 1.0f + threadIdx.x * 1e-3f + i; y[i] = 0.5f + i; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mf) {
        if (MODE == 0) { for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(8); }
        else
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 16; ++m) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);   // one dependent chain
            }
    } else {
        if (MODE == 2) { for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(8); }
        else
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(y[i]));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 8 * 8 * 256);
    const int iters = 2048;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    static unsigned long long h[8 * 256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m[8] = {0};
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) m[w] += h[b * 8 + w] / 256.0 / iters;
    printf("%-58s cycles per iteration, waves 0-3:", name);
    for (int w = 0; w < 4; ++w) printf(" %7.1f", m[w]);
    printf("   waves 4-7:");
    for (int w = 4; w < 8; ++w) printf(" %7.1f", m[w]);
    printf("\n");
}
int main() {
    printf("per iteration: an MFMA wave issues 16 dependent v_mfma_f32_32x32x16_bf16 (512 matrix cycles), a VALU wave 64 v_sub_f32\n");
    run<2>("MFMA waves alone (VALU waves sleep)");
    run<0>("VALU waves alone (MFMA waves sleep)");
    run<1>("both: one MFMA wave + one VALU wave per SIMD");
    run<3>("all eight waves VALU (two per SIMD)");
    run<4>("all eight waves MFMA (two per SIMD)");
    return 0;
}
