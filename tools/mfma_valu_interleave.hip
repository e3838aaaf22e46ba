// Same-wave interleave on gfx950: a chain of dependent v_mfma_f32_32x32x16_bf16 with K independent VALU instructions (and
// optionally LDS traffic) placed between consecutive MFMAs, one or two such waves per SIMD.  Reports matrix-pipe cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -w tools/mfma_valu_interleave.hip -o /tmp/il && /tmp/il
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// K: VALU per MFMA; CH: number of independent VALU chains (1 = every VALU depends on the previous one); LDSOPS: 0 none, 1: one ds_read_b128 + one ds_write_b64 per MFMA
// NACC: number of independent MFMA accumulator chains rotated (1 = fully dependent)
template <int K, int CH, int LDSOPS, int NACC>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ float4 sh[2048];
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f + i); }
    float x[8], y[8];
    for (int i = 0; i < 8; ++i) { x[i] = 1.0f + threadIdx.x * 1e-3f + i; y[i] = 0.5f + i; }
    sh[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
    sh[threadIdx.x + 512] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    float4 ld = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < K; ++v) {
                const int c = (m * K + v) % CH;
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x[c]) : "v"(x[c]), "v"(y[c]));
            }
            if (LDSOPS) {
                const float4 t = sh[(threadIdx.x + m * 64) & 1023];
                ld.x += t.x;
                *reinterpret_cast<float2*>(&sh[1024 + ((threadIdx.x + m * 32) & 1023)]) = make_float2(x[0], x[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = ld.x;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int K, int CH, int LDSOPS, int NACC>
void run() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 8 * 8 * 256);
    const int iters = 1024;
    printf("K=%d VALU per MFMA, %d VALU chain(s), LDS ops %d, %d acc chain(s):", K, CH, LDSOPS, NACC);
    for (int waves = 4; waves <= 8; waves += 4) {       // 256 threads = one wave per SIMD, 512 = two
        hipLaunchKernelGGL((k<K, CH, LDSOPS, NACC>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        static unsigned long long h[8 * 256];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) m += h[b * 8 + w];
        m /= 256.0 * waves * iters * 16;
        printf("   %d wave%s/SIMD: %6.1f cycles per MFMA of a wave (%5.1f per SIMD)", waves / 4, waves == 4 ? " " : "s", m, m / (waves / 4));
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 1, 0, 1>(); run<2, 8, 0, 1>(); run<3, 8, 0, 1>(); run<4, 8, 0, 1>(); run<6, 8, 0, 1>(); run<8, 8, 0, 1>();
    run<5, 8, 0, 1>(); run<5, 8, 1, 1>(); run<5, 8, 1, 4>(); run<6, 8, 1, 1>();
    run<3, 1, 0, 1>(); run<3, 2, 0, 1>(); run<4, 2, 0, 1>(); run<4, 4, 0, 1>();
    run<3, 8, 1, 1>(); run<4, 8, 1, 1>();
    run<0, 1, 0, 4>(); run<3, 8, 0, 4>(); run<4, 8, 0, 4>(); run<3, 8, 1, 4>(); run<6, 8, 1, 4>();
    return 0;
}
