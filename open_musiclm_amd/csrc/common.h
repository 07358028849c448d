// Shared device/host helpers for libomlm_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define OMLM_OK 0
#define OMLM_ERR_ARG (-1)
#define OMLM_ERR_LAUNCH (-2)
#define OMLM_ERR_UNSUPPORTED (-3)

extern "C" void omlm_set_error(const char* msg);

#define OMLM_CHECK_ARG(cond, msg)                                                     \
    do {                                                                              \
        if (!(cond)) {                                                                \
            omlm_set_error("bad argument: " msg " [" #cond "]");                      \
            return OMLM_ERR_ARG;                                                      \
        }                                                                             \
    } while (0)

static inline int omlm_post_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s: launch failed: %s", what, hipGetErrorString(e));
        omlm_set_error(buf);
        return OMLM_ERR_LAUNCH;
    }
    return OMLM_OK;
}

// ---- the 16-bit GEMM / attention operand type of this build -----------------------------------------------------------------
// Every source that touches 16-bit operands is compiled TWICE: once with h16_t = bf16 (precision "bf16"; also hosts the fp32 /
// "bf16x3" instantiations) and once with -DOMLM_FP16=1, h16_t = IEEE half (precision "fp16": same v_mfma_f32_32x32x16 rate, 11
// instead of 8 significand bits).  The two copies live in different namespaces (kernel host stubs of the same name must not
// merge) and the fp16 copy's entry points are hidden `<name>_h` functions that the public entry point forwards to when it is
// handed dtype code 2 (OMLM_DT_F16), so include/omlm.h has ONE function per operation.
#ifndef OMLM_FP16
#define OMLM_FP16 0
#endif
#define OMLM_DT_F32 0
#define OMLM_DT_BF16 1
#define OMLM_DT_F16 2
// dtype code as the fp16 copy sees it: its 16-bit type is "code 1" there
#define OMLM_H_CODE(c) ((c) == OMLM_DT_F16 ? 1 : (c))
#if OMLM_FP16
typedef _Float16 h16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 h16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
#define OMLM_NS omlm_f16
#define OMLM_API(name) __attribute__((visibility("hidden"))) name##_h
#define OMLM_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define OMLM_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#else
typedef __bf16 h16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 h16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 h16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 h16x2;
#define OMLM_NS omlm_bf16
#define OMLM_API(name) name
#define OMLM_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define OMLM_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ unsigned f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(unsigned u) { return __uint_as_float(u); }

// round-to-nearest-even fp32 -> bf16 bits (finite inputs)
__device__ __forceinline__ unsigned bf16_bits_rne(float f) {
    unsigned u = f2u(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ unsigned pack_h16_rne(float lo, float hi) {
    h16x2 v;
    v[0] = (h16_t)lo;
    v[1] = (h16_t)hi;
    return __builtin_bit_cast(unsigned, v);
}
#if OMLM_FP16
__device__ __forceinline__ float h16_lo_to_f(unsigned packed) { return (float)__builtin_bit_cast(h16x2, packed)[0]; }
__device__ __forceinline__ float h16_hi_to_f(unsigned packed) { return (float)__builtin_bit_cast(h16x2, packed)[1]; }
#else
__device__ __forceinline__ float h16_lo_to_f(unsigned packed) { return u2f(packed << 16); }
__device__ __forceinline__ float h16_hi_to_f(unsigned packed) { return u2f(packed & 0xFFFF0000u); }
#endif

// split fp32 pair into packed (truncated) hi bf16 pair and RNE lo bf16 pair: x ~= hi + lo
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    unsigned ua = f2u(a) & 0xFFFF0000u, ub = f2u(b) & 0xFFFF0000u;
    hi = ub | (ua >> 16);
#if OMLM_FP16
    lo = bf16_bits_rne(a - u2f(ua)) | (bf16_bits_rne(b - u2f(ub)) << 16);       // planes are bf16 by definition (the fp16 copy never runs them)
#else
    lo = pack_h16_rne(a - u2f(ua), b - u2f(ub));
#endif
}

// the same sum on the VALU: 4 DPP adds inside each row of 16 lanes + 4 readlanes, no LDS-crossbar round trips (six dependent
// ds_bpermute steps are ~0.35 us of pure latency -- visible in kernels that are ONE dependent chain, like the decode GEMVs)
template <int CTRL>
__device__ __forceinline__ float dpp_add_step(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = dpp_add_step<0xB1>(v);        // quad_perm [1,0,3,2]
    v = dpp_add_step<0x4E>(v);        // quad_perm [2,3,0,1]
    v = dpp_add_step<0x141>(v);       // row_half_mirror
    v = dpp_add_step<0x140>(v);       // row_mirror: every lane holds its row's sum
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
}
template <int CTRL>
__device__ __forceinline__ float dpp_max_step(float v) {
    return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false)));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = dpp_max_step<0xB1>(v);
    v = dpp_max_step<0x4E>(v);
    v = dpp_max_step<0x141>(v);
    v = dpp_max_step<0x140>(v);
    const int i = __builtin_bit_cast(int, v);
    return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16))),
                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48))));
}
// OMLM_WAVE_DPP=1: every wave_sum / wave_max of the library takes the DPP form (build-time A/B switch)
#ifndef OMLM_WAVE_DPP
#define OMLM_WAVE_DPP 0
#endif
__device__ __forceinline__ float wave_sum(float v) {
#if OMLM_WAVE_DPP
    return wave_sum_dpp(v);
#else
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#if OMLM_WAVE_DPP
    return wave_max_dpp(v);
#else
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
#endif
}

// d(bias) of the attention backward: the 63 diagonal sums of a 32x32 block of dS^T held in the MFMA C-layout (lane = query column
// q = lane & 31, bv[r] = key row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).  Output lane L stands for t = q - kr = L - 31 and pulls row
// kr's element from query column q = t + kr of the half-wave that holds that row: source lane (L - 31 + kr + 32 hh) mod 64 with
// hh = (kr >> 2) & 1.  ds_bpermute_b32 adds its immediate offset to the byte address, so ONE address register (4 * lane) serves all 32
// permutes; hipcc's __shfl form spends an index register plus two address VALU ops per permute (32 VGPRs, 64 of the ~130 VALU
// instructions of the loop, seen in the ISA; with the asm form the bf16 dQ kernel stops spilling: 20 -> 0 VGPRs).  The permutes are
// inline asm, i.e. invisible to hipcc's lgkmcnt bookkeeping: issued back to back and retired by the s_waitcnt of this function.
// MEASURED (MI355X, B = 32, N = 1116, H = 8): parity green (6 / 6 attention tests), backward 665.6 us against 665.4 us with __shfl --
// no gain: the ~150 us of d(bias) are the 32 LDS-crossbar permutes themselves, not their address arithmetic or the spills.  The asm
// form is therefore OFF by default (a hidden load is a liability where registers spill: the fp32 kernel still spills 48);
// -DOMLM_DIAG_ASM=1 selects it.
#ifndef OMLM_DIAG_ASM
#define OMLM_DIAG_ASM 0
#endif
#define OMLM_BPERM(KR) asm volatile("ds_bpermute_b32 %0, %1, %2 offset:%3" : "=v"(got[KR]) : "v"(base), \
        "v"(bv[4 * ((KR) >> 3) + ((KR) & 3)]), "i"(4 * (((KR) - 31 + 32 * (((KR) >> 2) & 1)) & 63)))
// Horner form on the VALU (no LDS crossbar): row kr's 32 values must move from lanes q (rows of the lower half-wave) / 32 + q (rows of
// the upper half-wave) to lanes q - kr + 31.  For the lower rows that is a right shift by 31 - kr, for the upper rows a left shift by
// kr + 1; taken in order of decreasing shift, each row is added after the running sum has been shifted by the difference (1 inside a
// group of four rows, 5 between groups): two chains of 15 DPP adds + 16 plain DPP shifts (wave_shr:1 / wave_shl:1, zero fill) on
// half-masked copies of the registers.  -DOMLM_DIAG_HORNER=1.
#ifndef OMLM_DIAG_HORNER
#define OMLM_DIAG_HORNER 1
#endif
template <int CTRL>
__device__ __forceinline__ float dpp_shift1(float v) {       // CTRL 0x138: lane i <- lane i - 1 (lane 0 <- 0); 0x130: lane i <- lane i + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float diag_sum_32x32_horner(const float (&bv)[16], int lane) {
    // half-masked copies by packed multiplies with {1, 0} lane constants (16 v_pk_mul_f32 instead of 32 v_cndmask_b32; dS is finite)
    const float mlo = lane < 32 ? 1.f : 0.f, mhi = 1.f - mlo;
    float va[16], vb[16];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 v = {bv[r], bv[r + 1]};
        const f32x2 x = v * f32x2{mlo, mlo}, y = v * f32x2{mhi, mhi};
        va[r] = x[0]; va[r + 1] = x[1]; vb[r] = y[0]; vb[r + 1] = y[1];
    }
    // chain A: rows of the lower half-wave, kr = (r & 3) + 8 (r >> 2) ascending = shift 31 - kr descending
    float a = va[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) {
        const int gap = (r & 3) == 0 ? 5 : 1;
#pragma unroll
        for (int g = 0; g < gap; ++g) a = dpp_shift1<0x138>(a);
        a += va[r];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) a = dpp_shift1<0x138>(a);    // the last row (kr = 27) needs 31 - 27
    // chain B: rows of the upper half-wave, kr = 4 + (r & 3) + 8 (r >> 2); shift left by kr + 1, largest first
    float b = vb[15];
#pragma unroll
    for (int r = 14; r >= 0; --r) {
        const int gap = (r & 3) == 3 ? 5 : 1;
#pragma unroll
        for (int g = 0; g < gap; ++g) b = dpp_shift1<0x130>(b);
        b += vb[r];
    }
#pragma unroll
    for (int g = 0; g < 5; ++g) b = dpp_shift1<0x130>(b);    // the last row (kr = 4) needs 4 + 1
    return a + b;
}
__device__ __forceinline__ float diag_sum_32x32(const float (&bv)[16], int lane) {
#if OMLM_DIAG_HORNER
    return diag_sum_32x32_horner(bv, lane);
#endif
    float dsum = 0.f;
#if OMLM_DIAG_ASM
    float got[32];
    const int base = lane << 2;
    OMLM_BPERM(0); OMLM_BPERM(1); OMLM_BPERM(2); OMLM_BPERM(3); OMLM_BPERM(4); OMLM_BPERM(5); OMLM_BPERM(6); OMLM_BPERM(7);
    OMLM_BPERM(8); OMLM_BPERM(9); OMLM_BPERM(10); OMLM_BPERM(11); OMLM_BPERM(12); OMLM_BPERM(13); OMLM_BPERM(14); OMLM_BPERM(15);
    OMLM_BPERM(16); OMLM_BPERM(17); OMLM_BPERM(18); OMLM_BPERM(19); OMLM_BPERM(20); OMLM_BPERM(21); OMLM_BPERM(22); OMLM_BPERM(23);
    OMLM_BPERM(24); OMLM_BPERM(25); OMLM_BPERM(26); OMLM_BPERM(27); OMLM_BPERM(28); OMLM_BPERM(29); OMLM_BPERM(30); OMLM_BPERM(31);
    // retire them: the wait names the first 16 destinations, the empty statement behind it the other 16, so that no consumer of
    // any of them is scheduled above the wait (cdna_hip_programming.md 5.7: loads hidden from hipcc, form (ii))
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(got[0]), "+v"(got[1]), "+v"(got[2]), "+v"(got[3]), "+v"(got[4]), "+v"(got[5]), "+v"(got[6]),
                 "+v"(got[7]), "+v"(got[8]), "+v"(got[9]), "+v"(got[10]), "+v"(got[11]), "+v"(got[12]), "+v"(got[13]), "+v"(got[14]), "+v"(got[15]));
    asm volatile("" : "+v"(got[16]), "+v"(got[17]), "+v"(got[18]), "+v"(got[19]), "+v"(got[20]), "+v"(got[21]), "+v"(got[22]), "+v"(got[23]),
                 "+v"(got[24]), "+v"(got[25]), "+v"(got[26]), "+v"(got[27]), "+v"(got[28]), "+v"(got[29]), "+v"(got[30]), "+v"(got[31]));
#pragma unroll
    for (int kr = 0; kr < 32; ++kr) {
        const int src = lane - 31 + kr;
        dsum += (src >= 0 && src < 32) ? got[kr] : 0.f;
    }
#else
#pragma unroll
    for (int kr = 0; kr < 32; ++kr) {
        const int r = 4 * (kr >> 3) + (kr & 3), hh = (kr >> 2) & 1;
        const int src = lane - 31 + kr;
        const float got = __shfl(bv[r], (32 * hh + src) & 63, 64);
        dsum += (src >= 0 && src < 32) ? got : 0.f;
    }
#endif
    return dsum;
}

// block-wide sum for blockDim.x == NT (multiple of 64); `red` is >= NT/64 floats of LDS.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// Philox-4x32-10 counter RNG: the dropout mask of element e is a pure function of (seed, e),
// so the backward kernel regenerates it instead of storing it.
template <int ROUNDS = 10>
__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                           unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        unsigned n1 = (unsigned)p1;
        unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned long long bytes) {
    unsigned n = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)n, 0x00020000);
}
#define OOB_OFF 0xFFFFFFF0u

template <typename T> struct elt_traits;
template <> struct elt_traits<float> { static constexpr bool precise = true; };
template <> struct elt_traits<h16_t> { static constexpr bool precise = false; };

__device__ __forceinline__ float load_as_float(const float* p) { return *p; }
__device__ __forceinline__ float load_as_float(const h16_t* p) { return (float)(*p); }
__device__ __forceinline__ void store_from_float(float* p, float v) { *p = v; }
__device__ __forceinline__ void store_from_float(h16_t* p, float v) { *p = (h16_t)v; }

#if !OMLM_FP16
// Files that exist once (elementwise casts, the optimizer's 16-bit weight shadow, the loss gradient) serve fp16 outputs from the
// same copy: dtype code 2 selects the half instantiation.
typedef _Float16 f16_t;
__device__ __forceinline__ void store_from_float(f16_t* p, float v) { *p = (f16_t)v; }
#endif

// 4 consecutive outputs as ONE store (16 B fp32 / 8 B bf16) -- scalar bf16 stores are 2-byte scatters
__device__ __forceinline__ void store4_from_float(float* p, float a, float b, float c, float d) { *(float4*)p = make_float4(a, b, c, d); }
__device__ __forceinline__ void store4_from_float(h16_t* p, float a, float b, float c, float d) {
    u32x2 o;
    o[0] = pack_h16_rne(a, b);
    o[1] = pack_h16_rne(c, d);
    *(u32x2*)p = o;
}

// ---- fp8 (e4m3) planes with one power-of-two scale per row: the operands of omlm_gemm_mx16's correction products (csrc/gemm_mx.hip) --------------
// four values -> four e4m3 bytes (RNE; |v| <= 448 by construction of the row scale)
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (unsigned)r;
}
// four values -> four bf8 (e5m2) bytes, RNE: the upper byte of a half, same exponent range -- no scale (the lo plane of h1)
__device__ __forceinline__ unsigned pack4_bf8(float a, float b, float c, float d) {
    int r = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, r, true);
    return (unsigned)r;
}
// exponent e of a row whose entries are bounded by `bound`: bound <= 2^(e + 8) (frexp: bound = m 2^ex with m < 1); an all-zero row takes e = -100
__device__ __forceinline__ int mx_row_exp(float bound) {
    int ex;
    (void)frexpf(bound, &ex);
    ex -= 8;
    return bound > 0.f && ex > -100 ? (ex > 100 ? 100 : ex) : -100;
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
