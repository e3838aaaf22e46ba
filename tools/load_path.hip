// tools/load_path.hip -- what the global -> LDS operand path of the implicit-GEMM kernels can deliver, without any MFMA.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/load_path.hip -o /tmp/load_path && /tmp/load_path
//
// A workgroup of 256 threads refills a [128 rows][128 B] LDS stage per step from 128-B row segments of a row-major matrix
// (8 lanes x 16 B per segment: the A-operand pattern of conv_igemm_kernel), 2 workgroups per CU, one barrier per step:
//   mode 0  registers, one step ahead      (load -> wait -> ds_write_b128 -> barrier: the fp32 / f32x3 loaders)
//   mode 1  registers, two steps ahead     (the bf16 loaders)
//   mode 2  LDS-DMA (buffer_load_dwordx4 ... lds), one step ahead, 2 stages
//   mode 3  LDS-DMA, three steps ahead, 4 stages
// and these footprints:
//   kind 0  every workgroup walks the same 4 MB, consecutive 128-B segments (L2 hits, ideal addresses)
//   kind 3  the same 4 MB as a [rows][stride] matrix: a step reads one 128-B column segment of 128 consecutive rows (L2 hits,
//           strided addresses: what a K-slice of an NHWC map looks like)
//   kind 1  workgroup b walks its own 128-row panels of a [rows][stride] matrix, slice by slice (1x1 convolution: HBM stream)
//   kind 2  like 1, but each segment is read nine times in a row (3x3 taps: 1 HBM miss + 8 L1/L2 hits)
// Output: bytes per clock and CU (at the measured clock: s_memtime / wall) and TB/s chip-wide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int ROWS_PER_STEP>
__global__ __launch_bounds__(256, 2) void loader(const float* __restrict__ src, float* __restrict__ sink, int src_kind, int steps,
                                                 long long row_stride_b, long long bytes_total, long long region_b) {
    constexpr int STAGE_B = ROWS_PER_STEP * 128;             // bytes per stage
    constexpr int NJ = ROWS_PER_STEP / 32;                   // 16-B loads per thread and step (32 rows per pass)
    constexpr int NSTAGE = MODE == 3 ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r0 = tid >> 3, chunk = (tid & 7) ^ ((r0 >> 1) & 7);      // swizzle on the SOURCE side (the LDS image is lane-linear)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)(bytes_total > 0x7fffffffll ? 0x7fffffff : bytes_total), 0x00020000);
    // address of (row r, 128-B segment s): src 0: segment index walks 4 MB linearly; src 1/2: panel rows x segments along K
    // (32-bit shift / mask arithmetic only: strides, region sizes and rows per region are powers of two -- a 64-bit division per
    //  load would make this an ALU benchmark)
    const int lstride = __builtin_ctzll((unsigned long long)row_stride_b), lspr = lstride - 7;
    const int lreg = region_b ? __builtin_ctzll((unsigned long long)region_b) : 0;
    const int lR = lreg - lstride;
    auto seg_off = [&](int step, int j) -> unsigned {
        const unsigned row = (unsigned)(r0 + 32 * j);
        if (src_kind == 0) return ((((unsigned)step * ROWS_PER_STEP + row) & ((1u << (lreg - 7)) - 1u)) << 7) + (unsigned)chunk * 16u;
        if (src_kind == 3) {
            const unsigned lin = (unsigned)step * ROWS_PER_STEP + row;
            return ((lin & ((1u << lR) - 1u)) << lstride) + (((lin >> lR) & ((1u << lspr) - 1u)) << 7) + (unsigned)chunk * 16u;
        }
        const unsigned s = src_kind == 2 ? (unsigned)step / 9u : (unsigned)step;
        const unsigned panel = blockIdx.x + gridDim.x * (s >> lspr);
        return ((panel * ROWS_PER_STEP + row) << lstride) + ((s & ((1u << lspr) - 1u)) << 7) + (unsigned)chunk * 16u;
    };
    float acc = 0.f;
    if constexpr (MODE <= 1) {
        constexpr int NSET = MODE == 1 ? 2 : 1;
        u32x4 r[NSET][NJ];
        auto issue = [&](int step, const int set) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) r[set][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, seg_off(step, j), 0, 0);
        };
        auto store = [&](int buf, const int set) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                *reinterpret_cast<u32x4*>(smem + buf * STAGE_B + ((r0 + 32 * j) * 8 + (tid & 7)) * 16) = r[set][j];
        };
        issue(0, 0);
        if (NSET == 2) issue(1, 1);
        for (int s = 0; s < steps; s += NSET) {
#pragma unroll
            for (int u = 0; u < NSET; ++u) {
                store((s + u) & 1, u);                       // (waits for set u)
                if (s + u + NSET < steps) issue(s + u + NSET, u);
                __syncthreads();
                acc += *reinterpret_cast<const float*>(smem + ((s + u) & 1) * STAGE_B + ((tid * 20) & (STAGE_B - 1) & ~3));
            }
        }
    } else {
        auto issue = [&](int step) {
            const int buf = step % NSTAGE;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // wave w of pass j fills rows [8 w + 32 j, +8): 1 KB, lane-linear
                char* dst = smem + buf * STAGE_B + ((8 * wave + 32 * j) * 8) * 16;
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, seg_off(step, j), 0, 0, 0);
#else
                (void)dst;
#endif
            }
        };
        constexpr int AHEAD = NSTAGE - 1;
        for (int s = 0; s < AHEAD && s < steps; ++s) issue(s);
        for (int s = 0; s < steps; ++s) {
            // stage s has landed once at most AHEAD - 1 younger steps' loads are outstanding
            if (AHEAD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (s + AHEAD - 1 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * NJ) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            acc += *reinterpret_cast<const float*>(smem + (s % NSTAGE) * STAGE_B + ((tid * 20) & (STAGE_B - 1) & ~3));
            __syncthreads();                                 // stage (s % NSTAGE) may be refilled
            if (s + AHEAD < steps) issue(s + AHEAD);
        }
    }
    if (acc == 12345.678f) sink[tid] = acc;
}

template <int MODE, int RPS>
static void run(const float* src, float* sink, int src_kind, int steps, long long row_stride_b, long long bytes_total, int ncu, const char* name, long long region_b) {
    const int lds = (MODE == 3 ? 4 : 2) * RPS * 128;
    auto k = loader<MODE, RPS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int G = 2 * ncu;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(G), dim3(256), lds, 0, src, sink, src_kind, steps, row_stride_b, bytes_total, region_b);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(G), dim3(256), lds, 0, src, sink, src_kind, steps, row_stride_b, bytes_total, region_b);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double bytes = (double)G * steps * RPS * 128;
    const double tbs = bytes / (best * 1e-3) / 1e12;
    printf("  %-34s stage %3d KB  %8.1f us  %6.2f TB/s  %5.1f B/clk/CU @2.4GHz\n", name, RPS * 128 / 1024, best * 1e3, tbs,
           bytes / (best * 1e-3) / ncu / 2.4e9);
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    const long long bytes_total = 1ll << 31;                  // 2 GiB source
    float* src; float* sink;
    CK(hipMalloc(&src, bytes_total)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(src, 0, bytes_total));
    struct Case { int kind; long long stride; int steps; const char* name; long long region; };
    const Case cases[] = {
        {0, 128, 256, "same 1 MB, consecutive segments (L2 hits)", 1 << 20},
        {3, 256, 256, "same 1 MB as [4096][256 B] rows, a column of segments per step (L2 hits)", 1 << 20},
        {3, 1024, 256, "same 1 MB as [1024][1 KB] rows (L2 hits)", 1 << 20},
        {3, 4096, 256, "same 1 MB as [256][4 KB] rows (L2 hits)", 1 << 20},
        {0, 128, 256, "same 4 MB, consecutive segments", 4 << 20},
        {3, 1024, 256, "same 4 MB as [4096][1 KB] rows", 4 << 20},
        {0, 128, 256, "same 64 MB, consecutive segments (Infinity Cache hits)", 64 << 20},
        {3, 1024, 256, "same 64 MB as [65536][1 KB] rows (Infinity Cache hits)", 64 << 20},
        {1, 4096, 128, "own panels, 4 KB rows = 1024 fp32 channels (HBM stream)", 0},
        {1, 1024, 128, "own panels, 1 KB rows = 256 fp32 channels (HBM stream)", 0},
        {2, 1024, 576, "own panels, 1 KB rows, each segment x9 (3x3 taps)", 0},
    };
    for (const Case& c : cases) {
        printf("source: %s\n", c.name);
        const int sk = c.kind, steps = c.steps;
        const long long rs = c.stride;
        run<0, 128>(src, sink, sk, steps, rs, bytes_total, ncu, "registers, 1 ahead", c.region);
        run<1, 128>(src, sink, sk, steps, rs, bytes_total, ncu, "registers, 2 ahead", c.region);
        run<2, 128>(src, sink, sk, steps, rs, bytes_total, ncu, "LDS-DMA, 1 ahead (2 stages)", c.region);
        run<3, 128>(src, sink, sk, steps, rs, bytes_total, ncu, "LDS-DMA, 3 ahead (4 stages)", c.region);
        run<0, 256>(src, sink, sk, steps, rs, bytes_total, ncu, "registers, 1 ahead", c.region);
    }
    return 0;
}
