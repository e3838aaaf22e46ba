#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// tools/glds_probe.hip: does `buffer_load_dwordx4 ... lds` (LDS-DMA; __builtin_amdgcn_raw_ptr_buffer_load_lds) (a) take a per-lane SOURCE
// permutation, (b) write ZEROS for out-of-range offsets (the zero padding of the implicit GEMM's taps), (c) reach LDS addresses above
// 64 KB (M0 carries the destination base; gfx950 has 160 KB per CU)?  Run: hipcc --offload-arch=gfx950 -O3 -w tools/glds_probe.hip && ./a.out
__global__ void k(const float* src, float* dst, int n_valid_bytes, int base) {
    extern __shared__ __attribute__((aligned(16))) char smem0[];
    char* smem = smem0 + base;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = -1.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n_valid_bytes, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // lane p of wave w fetches 16 B at byte offset (w*64 + (p ^ 5)) * 16  (a source-side permutation), odd lanes of wave 1 out of range
    unsigned off = (unsigned)((wave * 64 + (lane ^ 5)) * 16);
    if (wave == 1 && (lane & 1)) off = 0x7fffffffu;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) dst[i] = reinterpret_cast<float*>(smem)[i];
}
int main() {
    std::vector<float> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *s, *d; hipMalloc(&s, 4096 * 4); hipMalloc(&d, 1024 * 4);
    hipMemcpy(s, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int rc = 0;
  for (int base : {0, 60 * 1024, 100 * 1024, 140 * 1024}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), base + 16384, 0, s, d, 4096 * 4, base);
    std::vector<float> o(1024); hipMemcpy(o.data(), d, 1024 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w) for (int p = 0; p < 64; ++p) for (int e = 0; e < 4; ++e) {
        float want = (w == 1 && (p & 1)) ? 0.f : (float)((w * 64 + (p ^ 5)) * 4 + e);
        float got = o[(w * 64 + p) * 4 + e];
        if (got != want) { if (bad < 8) printf("w%d p%d e%d got %g want %g\n", w, p, e, got, want); ++bad; }
    }
    printf("glds test, LDS base %d KB: %d mismatches\n", base / 1024, bad);
    rc |= bad != 0;
  }
    return rc;
}
