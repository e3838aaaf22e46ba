// test-only helper of tools/bench_wgrad_sensitivity.py (compiled there with hipcc, NOT part of liberd_hip.so):
// `blocks` workgroups of 1024 threads that each hold a CU slot for `cycles` shader cycles -- a stand-in for RCCL channels
// that are resident while a compute kernel launches.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void occupy_kernel(long long cycles, unsigned long long* sink) {
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    unsigned long long acc = 0;
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < cycles) {
        acc += 1;
        __builtin_amdgcn_s_sleep(32);
    }
    if (acc == 0xffffffffffffull) sink[0] = acc;
}
extern "C" int occupy_cus(int blocks, long long cycles, void* sink, void* stream) {
    // 1024 threads + 64 KB of LDS per workgroup: one workgroup takes half of a CU's wave slots and LDS
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(1024), 65536, (hipStream_t)stream, cycles, (unsigned long long*)sink);
    return (int)hipGetLastError();
}
