// Would a THREE-LIMB Winograd F(2x2,3x3) feed the bf16 matrix cores?  (probe before building.)
//   hipcc --offload-arch=gfx950 -O3 -w tools/wino_x3_skeleton.hip -o /tmp/wino_x3 && /tmp/wino_x3
//
// The fp32 Winograd kernel (erd_amd/csrc/winograd.hip) works on items of 32 tiles x 64 couts: per 16-channel slice its four matrix
// waves issue 128 v_mfma_f32_16x16x4_f32 each (4 096 cycles) and stream 64 KB of U fragments (16 positions x 64 couts x 16
// channels x 4 B) from L2 into registers -- 16 B/clk/CU next to the ~22 B/clk/CU a CU's vector-memory path sustains
// (tools/load_path.hip).  The accumulators of an item, 16 positions x 32 tiles x 64 couts x 4 B = 128 KB, fill the 128
// accumulator registers of the four matrix waves; a larger item does not fit (8 waves x 256 registers per CU, half of them
// needed for everything else).
// In the three-limb form the same slice is 6 x 16 positions = 96 v_mfma_f32_32x32x16_bf16 per cout block of 32, i.e. 48 per
// matrix wave = 1 536 cycles -- but U now travels as three bf16 planes: 16 x 64 x 16 x 6 B = 96 KB per slice, 62 B/clk/CU.
// This skeleton measures exactly that and nothing else: four matrix waves per workgroup, one workgroup per CU, every wave
// streams its 24 KB of fragments per slice (8 positions x 3 planes x 1 KB, pre-tiled so that a wave's load is 1 KB contiguous:
// the friendliest layout) through a register ring and issues its 48 MFMAs; V fragments come from registers (no LDS traffic, no
// transform, no data waves, no output stage).  Variants: T = 64 tiles (every U fragment feeds two MFMAs; needs 256 accumulator
// registers per wave, i.e. no room for anything else), fp32 U split on chip (4 B per value, 64 KB per slice; 2.75 VALU per value
// on the MATRIX waves), and the MFMA-only / load-only halves.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ inline u4 buf_load16(const __amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}

// MODE 0: loads + MFMAs; 1: MFMAs only (fragments loaded once); 2: loads only (one MFMA per slice keeps the data live)
// TM: 32-tile blocks per item (1 = the item the register file allows, 2 = 64 tiles)
// PLANES: 3 = pre-split bf16 limb planes (6 B per U value); 2 = fp32 U (two 16-B loads carry 8 values), split in registers
template <int MODE, int TM, int PLANES>
__global__ __launch_bounds__(256, 1) void skeleton(const void* __restrict__ U, size_t ubytes, float* sink, unsigned long long* ticks,
                                                    int slices) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(U), 0, (int)ubytes, 0x00020000);
    f32x16 acc[8][TM];
    for (int p = 0; p < 8; ++p)
        for (int t = 0; t < TM; ++t)
            for (int r = 0; r < 16; ++r) acc[p][t][r] = 0.f;
    bf16x8 v[TM][3];
    for (int t = 0; t < TM; ++t)
        for (int l = 0; l < 3; ++l)
            for (int r = 0; r < 8; ++r) v[t][l][r] = (__bf16)(0.01f * ((lane * 7 + r * 3 + l + t) & 31) - 0.15f);
    // a wave's stream: slice-major, 8 positions x PLANES fragments of 1 KB each; workgroups of one XCD share a cout block's stream
    const unsigned per_slice = 8u * PLANES * 1024u;
    const unsigned wave_base = ((blockIdx.x & 7) * 4 + wave) * 16u * per_slice;      // 16 slices = one item's stream (Cin = 256)
    u4 f[8][PLANES];
    auto load_slice = [&](int s) {
        const unsigned base = wave_base + (unsigned)(s & 15) * per_slice + lane * 16u;
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int l = 0; l < PLANES; ++l) f[p][l] = buf_load16(rs, base + (p * PLANES + l) * 1024u);
    };
    load_slice(0);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < slices; ++s) {
        u4 g[8][PLANES];
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int l = 0; l < PLANES; ++l) g[p][l] = f[p][l];
        if (MODE != 1) load_slice(s + 1);          // the next slice's fragments travel under this slice's MFMAs
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            bf16x8 u[3];
            if constexpr (PLANES == 3) {
#pragma unroll
                for (int l = 0; l < 3; ++l) u[l] = __builtin_bit_cast(bf16x8, g[p][l]);
            } else {                                // fp32 U: eight values in two registers quads, split into limbs here (round to nearest)
                u4 h, m, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = __uint_as_float(g[p][e >> 1][(e & 1) * 2]), x1 = __uint_as_float(g[p][e >> 1][(e & 1) * 2 + 1]);
                    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    auto pk = [](float a, float b) { const f2 q = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(q, b2)); };
                    h[e] = pk(x0, x1);
                    const float r0 = x0 - __uint_as_float(h[e] << 16), r1 = x1 - __uint_as_float(h[e] & 0xffff0000u);
                    m[e] = pk(r0, r1);
                    lo[e] = pk(r0 - __uint_as_float(m[e] << 16), r1 - __uint_as_float(m[e] & 0xffff0000u));
                }
                u[0] = __builtin_bit_cast(bf16x8, h); u[1] = __builtin_bit_cast(bf16x8, m); u[2] = __builtin_bit_cast(bf16x8, lo);
            }
            if (MODE == 2) {
                if (p == 0) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[0][0], u[0], acc[0][0], 0, 0, 0);
                else acc[p][0][0] += __uint_as_float(g[p][0][0] ^ g[p][PLANES - 1][3]);
                continue;
            }
#pragma unroll
            for (int t = 0; t < TM; ++t) {          // the six limb products of weight >= 2^-16
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[t][0], u[2], acc[p][t], 0, 0, 0);
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[t][0], u[1], acc[p][t], 0, 0, 0);
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[t][0], u[0], acc[p][t], 0, 0, 0);
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[t][1], u[1], acc[p][t], 0, 0, 0);
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[t][1], u[0], acc[p][t], 0, 0, 0);
                acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[t][2], u[0], acc[p][t], 0, 0, 0);
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float sacc = 0.f;
    for (int p = 0; p < 8; ++p)
        for (int t = 0; t < TM; ++t) sacc += acc[p][t][0] + acc[p][t][9];
    if (sacc == 123.456f) sink[0] = sacc;
    if (threadIdx.x == 0) ticks[blockIdx.x] = c1 - c0;
}

template <int MODE, int TM, int PLANES>
void run(const char* what, const void* U, size_t ubytes, float* sink, unsigned long long* ticks, int grid, int slices) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((skeleton<MODE, TM, PLANES>), dim3(grid), dim3(256), 0, 0, U, ubytes, sink, ticks, slices / 8);
    hipDeviceSynchronize();
    std::vector<double> us, cyc;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((skeleton<MODE, TM, PLANES>), dim3(grid), dim3(256), 0, 0, U, ubytes, sink, ticks, slices);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        us.push_back(ms * 1e3);
        cyc.push_back((double)h[grid / 2] / slices);
    }
    std::sort(us.begin(), us.end());
    std::sort(cyc.begin(), cyc.end());
    const double mf = MODE == 2 ? 0.0 : 4.0 * 48 * TM * 32768.0;                  // MFMA flop per workgroup and slice
    const double by = MODE == 1 ? 0.0 : 4.0 * 8 * PLANES * 1024.0;                // fragment bytes per workgroup and slice
    const double ns = us[2] * 1e3 / slices;
    printf("%-66s %8.1f us  %7.0f ns/slice  %6.0f shader cycles/slice  ", what, us[2], ns, cyc[2]);
    if (mf > 0) printf("%7.0f TF executed (%5.1f fp32-equivalent Winograd-executed TF)  ", grid * mf / ns / 1e3, grid * mf / 6 / ns / 1e3);
    if (by > 0) printf("%5.1f TB/s of fragments = %4.1f B/clk/CU at the measured clock", grid * by / ns / 1e3, by / cyc[2]);
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount;
    const size_t ubytes = 8 * 4 * 16 * 8 * 3 * 1024;       // 8 XCD streams x 4 waves x 16 slices x 24 KB = 12.6 MB (L2 / Infinity Cache resident)
    void* U;
    float* sink;
    unsigned long long* ticks;
    hipMalloc(&U, ubytes);
    hipMalloc(&sink, 4);
    hipMalloc(&ticks, sizeof(unsigned long long) * grid);
    std::vector<unsigned short> h(ubytes / 2);
    unsigned x = 12345u;
    for (auto& e : h) { x = x * 1664525u + 1013904223u; e = (unsigned short)(0x3c00u + ((x >> 16) & 0x03ffu) + ((x >> 8) & 0x8000u)); }   // |u| in [0.0078, 0.0156)
    hipMemcpy(U, h.data(), ubytes, hipMemcpyHostToDevice);
    printf("%s, %d CUs; one workgroup (4 matrix waves) per CU; ideal matrix time of a slice: 48 MFMAs x 32 cycles = 1 536 cycles per wave\n",
           prop.name, grid);
    const int S = 16 * 64;
    run<1, 1, 3>("MFMAs only, 32 tiles x 64 couts (48 per wave and slice)", U, ubytes, sink, ticks, grid, S);
    run<2, 1, 3>("fragment loads only, bf16 limb planes (96 KB per slice)", U, ubytes, sink, ticks, grid, S);
    run<0, 1, 3>("loads + MFMAs, 32 tiles x 64 couts, bf16 limb planes", U, ubytes, sink, ticks, grid, S);
    run<0, 2, 3>("loads + MFMAs, 64 tiles x 64 couts (256 accumulators per wave)", U, ubytes, sink, ticks, grid, S / 2);
    run<2, 1, 2>("fragment loads only, fp32 U (64 KB per slice)", U, ubytes, sink, ticks, grid, S);
    run<0, 1, 2>("loads + MFMAs, fp32 U split on the matrix waves", U, ubytes, sink, ticks, grid, S);
    printf("for comparison: the fp32 Winograd kernel's slice is 4 096 matrix cycles for the same 32 tiles x 64 couts x 16 channels;\n"
           "a three-limb form is faster only if its slice stays well below that.\n");
    return 0;
}
