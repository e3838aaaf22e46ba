// Shared helpers for liberd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/erd_hip.h"
#include "erd_probes.h"

namespace erd {

void set_error(const char* fmt, ...);
// CUs the persistent / one-round grids are sized for: the device's count minus erd_set_cu_reserve()'s reserve (0 by default).  Data
// parallel ranks leave a few CUs to RCCL's resident kernels: a whole-chip static grid loses a dispatch round to any foreign workgroup
// (profiles/r05_cu_theft_step.txt).  Workspace LAYOUTS keep using the physical count.
int usable_cus(int physical);

// conv_thin.hip: the activation-stationary three-limb kernel for 1x1 convolutions with Cin <= 128 (erd_conv_igemm dispatches to it)
bool conv_thin_x3_ok(const erd_conv_desc* d);
int conv_thin_x3(const erd_conv_desc* d, hipStream_t st);
bool conv_thin_bf16_ok(const erd_conv_desc* d);      // ... and its bf16-mode twin (bf16 multiplicands, maps stored bf16)
int conv_thin_bf16(const erd_conv_desc* d, hipStream_t st);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

#define ERD_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            erd::set_error(__VA_ARGS__);  \
            return ERD_EINVAL;            \
        }                                 \
    } while (0)

// hipMemsetAsync(p, 0, bytes) whose failure is reported like a failed launch (the entry point returns the HIP error code)
#define ERD_ZERO_ASYNC(p, bytes, st)                                           \
    do {                                                                       \
        const hipError_t e_ = hipMemsetAsync((p), 0, (bytes), (st));           \
        if (e_ != hipSuccess) {                                                \
            erd::set_error("hipMemsetAsync: %s", hipGetErrorString(e_));       \
            return (int)e_;                                                    \
        }                                                                      \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// The sum over the wave in every lane, on the VECTOR pipe: four DPP steps (quad swaps, half-row and row mirrors) give every lane its
// row-of-16's sum, four v_readlane fetch the rows' sums.  wave_sum above compiles to six DEPENDENT ds_bpermute (the LDS crossbar,
// ~100 cycles each): fine for one sum per kernel, 10 000 cycles where an output stage needs sixteen (another summation order).
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define ERD_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    ERD_DPP_ADD(0xB1);      // quad_perm [1, 0, 3, 2]
    ERD_DPP_ADD(0x4E);      // quad_perm [2, 3, 0, 1]
    ERD_DPP_ADD(0x141);     // row_half_mirror
    ERD_DPP_ADD(0x140);     // row_mirror
#undef ERD_DPP_ADD
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- bf16 storage (BASELINE.json configs[2]: activations and their gradients live in HBM as bf16) ----------------
// Values are rounded to nearest even on the way out (v_cvt_pk_bf16_f32) and widened exactly on the way in.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {      // a -> bits 0..15 (lower address), b -> 16..31
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float4 unpack4_bf16(uint2 u) {
    return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
}
__device__ __forceinline__ uint2 pack4_bf16(float4 v) { return make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w)); }

// a pointer / integer that IS wave-uniform but that the compiler cannot prove uniform (a segment picked by a data-dependent loop):
// through v_readfirstlane, so that buffer resources built from it live in SGPRs (a resource in VGPRs turns every buffer
// load into a waterfall loop)
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- three-limb split ("f32x3"): an fp32 value as the EXACT sum of three bf16 values, each rounded to nearest even:
//   hi = rne(x), mid = rne(x - hi), lo = x - hi - mid      (x - hi has <= 16 significant bits, x - hi - mid <= 8: both exact)
// |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x| (half an ulp of the limb above), and the remainders are SIGN-SYMMETRIC around zero --
// a truncating split (the round-3 form) leaves remainders that all carry the sign of x, twice as large, so the limb products a
// three-limb GEMM drops (a_mid b_lo + a_lo b_mid + a_lo b_lo) were a bias towards zero of up to 2^-21 |a b| per product; here
// they are zero-mean and below 2^-23 |a b|.  Two values per call: v_cvt_pk_bf16_f32 rounds and packs a pair in one instruction.
// limb words: value 0 in bits 0..15, value 1 in bits 16..31 (the order the bf16 MFMA fragments and LDS rows want).
// (`scalar_op`: the value passes through an empty asm, so clang's SLP vectorizer cannot pair the subtraction that produced it with its
//  neighbour into a v_pk_add_f32 -- a packed fp32 instruction costs ~13 cycles over its issue slot beside an MFMA stream on gfx950, and the
//  limb split always runs beside one: profiles/r05_noslp_ab.txt.  No instruction is emitted.)
__device__ __forceinline__ float scalar_op(float v) {
    asm("" : "+v"(v));
    return v;
}
__device__ __forceinline__ void limbs3_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = pack2_bf16(x0, x1);
    const float r0 = scalar_op(x0 - bf16_lo(hi)), r1 = scalar_op(x1 - bf16_hi(hi));
    mid = pack2_bf16(r0, r1);
    lo = pack2_bf16(scalar_op(r0 - bf16_lo(mid)), scalar_op(r1 - bf16_hi(mid)));
}
__device__ __forceinline__ void limbs3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    unsigned h, m, l;
    limbs3_pair(x, 0.f, h, m, l);
    hi = (unsigned short)(h & 0xffffu);
    mid = (unsigned short)(m & 0xffffu);
    lo = (unsigned short)(l & 0xffffu);
}

// ---- Winograd F(2x2,3x3) weight image in the THREE-LIMB layout (winograd.hip wino_x3_kernel; prep.hip kind 4) ----------------
// U_xi = (G g G^T)[xi] of output channel co / input channel ci -- the arithmetic of wino_weight_kernel, value for value -- split
// into limbs and stored as MFMA A-fragments of v_mfma_f32_32x32x16_bf16: U3[xi][limb][co / 32][ci / 16][lane = (ci % 16 / 8) * 32 +
// co % 32][ci % 8], 1 KB per (xi, limb, cout block, slice); rows past Cout are zero.  One thread per (co < ceil32(Cout), ci).
__device__ __forceinline__ void wino_x3_weight_item(const float* __restrict__ w, unsigned short* __restrict__ U3, int Cout, int Cin,
                                                    int flip, int co, int ci) {
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
    const int nks = Cin / 16, ncb32 = (Cout + 31) / 32;
    const int64_t per_xl = (int64_t)ncb32 * nks * 512;                       // bf16 elements per (position, limb)
    const int64_t base = (((int64_t)(co / 32) * nks + ci / 16) * 64 + ((ci % 16) / 8) * 32 + (co % 32)) * 8 + (ci % 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float u[4] = {t[i][0], 0.5f * (t[i][0] + t[i][1] + t[i][2]), 0.5f * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned short hi, mid, lo;
            limbs3(u[j], hi, mid, lo);
            const int64_t o = (int64_t)((i * 4 + j) * 3) * per_xl + base;
            U3[o] = hi;
            U3[o + per_xl] = mid;
            U3[o + 2 * per_xl] = lo;
        }
    }
}

// Element type of a map: float or erd::bf16s (a 16-bit storage cell).  ld4 / st4 move four consecutive values
// (16-B / 8-B aligned), ld1 / st1 one.
struct bf16s { unsigned short bits; };
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16s* p) { return unpack4_bf16(*reinterpret_cast<const uint2*>(p)); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16s* p, float4 v) { *reinterpret_cast<uint2*>(p) = pack4_bf16(v); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16s* p) { return __uint_as_float((unsigned)p->bits << 16); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16s* p, float v) { p->bits = (unsigned short)(pack2_bf16(v, 0.f) & 0xffffu); }

}  // namespace erd
