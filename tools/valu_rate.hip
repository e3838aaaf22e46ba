// Issue rate of the VALU instructions the limb split is made of (gfx950): cycles per wave-instruction with 1 / 2 / 4 waves per SIMD
// (1024 workgroups of 64 / 128 / 256 threads = four workgroups per CU; thread 0 of each workgroup times its own wave).
//   hipcc --offload-arch=gfx950 -O3 -w tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    float a[8], b[8];
    unsigned u[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0f + threadIdx.x * 1e-3f + i; b[i] = 0.5f + i; u[i] = threadIdx.x * 2654435761u + i; p[i] = (f2){a[i], b[i]}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {       // eight independent chains
            if (KIND == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
            if (KIND == 1) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]), "v"(0x07060302u));
            if (KIND == 2) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(p[(i + 1) & 7]));
            if (KIND == 4) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[i]) : "v"(u[i]));
            if (KIND == 5) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(0xffff0000u));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i] + p[i][0] + p[i][1];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 8 * 1024);
    const int iters = 4096;
    printf("%-22s", name);
    for (int waves = 1; waves <= 4; waves *= 2) {       // waves per SIMD: workgroups of 64 x waves threads, four per CU
        hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(64 * waves), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        unsigned long long h[1024];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 1024; ++i) m += h[i];
        const double per = m / 1024 / (iters * 8.0);
        printf("  %d wave%s/SIMD: %5.2f cycles per instruction of a wave (%4.2f per SIMD)", waves, waves > 1 ? "s" : " ", per, per / waves);
    }
    printf("\n");
}
int main() {
    run<2>("v_sub_f32"); run<3>("v_pk_add_f32"); run<0>("v_cvt_pk_bf16_f32"); run<1>("v_perm_b32"); run<4>("v_lshlrev_b32"); run<5>("v_and_b32");
    return 0;
}
