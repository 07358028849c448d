// FF-in / FF-out of precision "fp16ff" as ONE half product plus two fp8 correction products at twice the matrix rate (round 6).
//
//   C = A B^T with A = A_hi + A_lo, B = B_hi + B_lo (IEEE-half planes, x ~= hi + lo):
//   C ~= A_hi B_hi^T + A_hi B_lo^T + A_lo B_hi^T.  Round 5 ran the three products on v_mfma_f32_32x32x16_f16 (omlm_gemm_planes16: 3x the
//   k-tiles of a plain GEMM).  The two correction terms are 2^-11 of the main one, so a few significant bits are enough for them
//   (profiles/r06_error_budget_fp8corr.md: fp8 corrections leave 2.5e-5 / 4.2e-5 in the logits at depth 6 / 24, a sixth of what the rest
//   of the forward leaves): they run on v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands -- 2x the half rate, half the bytes --
//   and ONE power-of-two scale per operand row (E8M0 byte, constant along k: the instruction takes it from a VGPR, no per-k-tile traffic):
//       hi8 = fp8(hi 2^-e),  lo8 = fp8(lo 2^-(e - 11)),   e = the row's exponent byte - 127 (values <= 2^8 after scaling)
//       C += 2^(ea + eb - 11) (A_hi8 B_lo8^T + A_lo8 B_hi8^T)         [both corrections share the scale pair (ea, eb - 11)]
//   Loop tiles: K / 64 half tiles, then ceil(K / 128) fp8 tiles of (A_hi8, B_lo8), then the same of (A_lo8, B_hi8) -- 2x a plain GEMM.
//
// An fp8 plane has the row pitch of its half plane (lda * 2 bytes, k-th element at byte k, bytes [K, ceil128(K)) zero), so a 128-byte row
// chunk of it is addressed exactly like a 64-element chunk of the half plane: the LDS-DMA pieces, the LDS image and the ds_read_b128
// fragment reads of the half-tile-ring schedule (gemm.hip: gemm_tile8_body) are used UNCHANGED -- an fp8 tile is the same bytes covering
// twice the k -- and only the matrix instruction differs: two 16-byte fragments of a lane are the 32 k-values v_mfma_scale..32x32x64 wants
// (which 32 of the tile's k they are does not matter: A and B take the same ones, and the scale is constant along k).
// Measured layout of the instruction (tools/mx_probe*.hip, MI355X): lane l holds row / column l % 32; scale byte of lane l (selected by
// op_sel) applies to that row's k-block l / 32; value 2^(byte - 127); 4.5 PFLOP/s against 2.15 for v_mfma_f32_32x32x16_f16.
//
// Schedule, LDS images, epilogue: the half-tile ring of gemm.hip (two k-tiles of four 16-KiB half-tile slots, five half-tiles in flight
// behind counted vmcnt waits, waves 4-7 one barrier behind waves 0-3); see there for the RAW / WAR argument.
// The m-tile rows behind the last full machine round run as a slice-storing split-K of the same kernel (GemmArgs::c_split_stride) through a
// caller-owned workspace, summed in a fixed order by gemm_tail_reduce_kernel: deterministic, no atomics.
#include "gemm_common.h"

#if OMLM_FP16
namespace OMLM_NS {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

struct GemmMxArgs : GemmArgs {
    const void* A8; const void* B8;            // fp8 planes [hi8 | lo8], row pitch lda * 2 / ldb * 2 bytes, the lo8 plane a8_stride / b8_stride bytes behind the hi8 plane
    unsigned a8_stride, b8_stride;             // (every byte of both planes readable: rows are padded to whole 256-row tiles by the caller)
    const unsigned char* a_scale; const unsigned char* b_scale;               // E8M0 byte per row of A / B (of the hi8 plane; lo8: 2^-11 of it)
};

__device__ __forceinline__ i32x8 cat8(h16x8 x, h16x8 y) {
    const i32x4 a = __builtin_bit_cast(i32x4, x), b = __builtin_bit_cast(i32x4, y);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int HALF>
__device__ __forceinline__ h16x8 half8(i32x8 v) {
    const i32x4 h = HALF ? __builtin_shufflevector(v, v, 4, 5, 6, 7) : __builtin_shufflevector(v, v, 0, 1, 2, 3);
    return __builtin_bit_cast(h16x8, h);
}

// FUSE: one launch carries the full-round tiles (output type TOUT) AND the k-slices of the peeled tail (fp32 slices into the workspace): `tail`
// (uniform per workgroup) selects the epilogue; the k-loop is the same code for both.
template <typename TOUT, bool SLICE, bool FUSE = false>
__device__ __forceinline__ void gemm_mx_body(const GemmMxArgs& g, const int m0, const int n0, const int kt0, const int kt1, const bool split,
                                             const int ksplit, char* smem, const bool tail = false) {
    constexpr int A_BYTES = 256 * BK * 2, STAGE = 2 * A_BYTES;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned long long abytes = (unsigned long long)g.a_rows * g.lda * 2, bbytes = (unsigned long long)g.b_rows * g.ldb * 2;
    const dma_rsrc rsA = make_dma_rsrc(g.A, abytes), rsB = make_dma_rsrc(g.B, bbytes);
    const dma_rsrc rsA8 = make_dma_rsrc(g.A8, 2ull * g.a8_stride), rsB8 = make_dma_rsrc(g.B8, 2ull * g.b8_stride);
    DmaStagerT<false, 256, 8> sa, sb;
    sa.init(nullptr, g.lda, g.M, m0, wave, lane);
    sb.init(nullptr, g.ldb, g.N, n0, wave, lane);
    const int nk = kt1 - kt0;
    const int nk_half = g.K / BK, nk_main = (nk_half + 1) & ~1, nk8 = (g.K + 127) / 128;     // loop tiles: nk_main (half, padded to even) + 2 nk8 (fp8)
    const unsigned smem_lds = (unsigned)(size_t)LDS_PTR(char, smem);

    // the lane's row / column scales: byte (2 ha + i) of sA = row block (ha, i), byte hb of sB = column block hb (lo8 planes: exponent - 11, on B's side)
    unsigned sA = 0u, sB = 0u;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        const int row = m0 + (blk >> 1) * 128 + wr * 64 + 32 * (blk & 1) + (lane & 31);
        const unsigned e = row < g.M ? (unsigned)g.a_scale[row] : 127u;
        sA |= e << (8 * blk);
    }
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const int col = n0 + hb * 128 + wc * 32 + (lane & 31);
        const unsigned e = col < g.N ? (unsigned)g.b_scale[col] : 127u;
        sB |= (e > 11u ? e - 11u : 0u) << (8 * hb);
    }
    asm volatile("" : "+v"(sA), "+v"(sB));          // consumed HERE: hipcc's wait for these six loads sits in front of the DMA prologue, not inside the ring

    f32x16 acc[2][2][2];                    // [ha][hb][i]: rows ha * 128 + wr * 64 + 32 i, columns hb * 128 + wc * 32
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[x >> 2][(x >> 1) & 1][x & 1][e] = 0.f;

    // request event e_n, n = 4 T + J (J a compile-time constant): two pieces per wave; loop tile t -> (operand planes, byte offset along k)
    auto stage = [&](auto JC, const int T) {
        constexpr int J = decltype(JC)::value;
        constexpr bool isA = (J & 1) != 0;
        constexpr int h = J >> 1;
        const int t = kt0 + T;
        const bool live = T < nk && !(t == nk_half && nk_half < nk_main);           // (t == nk_half < nk_main: the pad tile of an odd half part)
        const bool mx = t >= nk_main, second = t >= nk_main + nk8;                 // (uniform)
        const int kb = t - (second ? nk_main + nk8 : (mx ? nk_main : 0));
        // A: half plane | hi8 (first correction) | lo8 (second);  B: half plane | lo8 | hi8.  An operand's two fp8 planes share ONE descriptor
        // (the lo8 plane sits a8_stride / b8_stride bytes behind the hi8 plane): the choice is an add on the instruction's SGPR offset
        const unsigned plane = isA ? (second ? g.a8_stride : 0u) : ((mx && !second) ? g.b8_stride : 0u);
        const unsigned koff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)kb * 128u + plane));      // bytes: 64 halves or 128 fp8 values per tile
        const dma_rsrc rs = isA ? (mx ? rsA8 : rsA) : (mx ? rsB8 : rsB);
        const unsigned base = smem_lds + (unsigned)((T & 1) * STAGE) + (isA ? 0u : (unsigned)A_BYTES);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = 2 * h + u;
            const int b = wave + 8 * i;
            if constexpr (isA) dma_issue_s(rs, base + (unsigned)(b * 1024), live ? sa.vfast[i] : OOB_OFF, koff);
            else               dma_issue_s(rs, base + (unsigned)(b * 1024), live ? sb.vfast[i] : OOB_OFF, koff);
        }
    };
    using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>;
    using J2 = std::integral_constant<int, 2>; using J3 = std::integral_constant<int, 3>;

    // fragments live as 8-dword tuples (k16 steps 2 s, 2 s + 1 of a lane): the fp8 instruction takes a tuple whole, the half instruction its two
    // 4-dword sub-tuples -- assembling tuples from separate 4-dword fragments in front of every fp8 MFMA cost copies and ~400 spilled registers
    i32x8 a[2][2], bX[2], bY[2], bZ[2];
    auto read_a = [&](const int T, const int ha) {
        const char* As = smem + (T & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                a[i][s] = cat8(read_frag<false>(As, ha * 128 + wr * 64 + 32 * i, 2 * s, lane), read_frag<false>(As, ha * 128 + wr * 64 + 32 * i, 2 * s + 1, lane));
    };
    auto read_b = [&](i32x8 (&b)[2], const int T, const int hb) {
        const char* Bs = smem + (T & 1) * STAGE + A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s)
            b[s] = cat8(read_frag<false>(Bs, hb * 128 + wc * 32, 2 * s, lane), read_frag<false>(Bs, hb * 128 + wc * 32, 2 * s + 1, lane));
    };
    // MXC: the tile is fp8 -- an 8-dword tuple is a lane's 32 k-values of a 64-deep step
    auto mma = [&](auto MXC, auto HAC, auto HBC, f32x16 (&c)[2], const i32x8 (&b)[2]) {
        constexpr bool MX = decltype(MXC)::value;
        constexpr int HA = decltype(HAC)::value, HB = decltype(HBC)::value;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (!MX) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i) c[i] = OMLM_MFMA_32x32x16(half8<0>(a[i][s]), half8<0>(b[s]), c[i]);
#pragma unroll
                for (int i = 0; i < 2; ++i) c[i] = OMLM_MFMA_32x32x16(half8<1>(a[i][s]), half8<1>(b[s]), c[i]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                c[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0][s], b[s], c[0], 0, 0, 2 * HA, (int)sA, HB, (int)sB);
                c[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[1][s], b[s], c[1], 0, 0, 2 * HA + 1, (int)sA, HB, (int)sB);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
#define T8_SYNC_IN()                                                           \
    do {                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                     \
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                      \
        __builtin_amdgcn_s_barrier();                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
        __builtin_amdgcn_sched_barrier(0);                                     \
    } while (0)
#define T8_SYNC_OUT()                                                          \
    do {                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                     \
        __builtin_amdgcn_s_barrier();                                          \
        __builtin_amdgcn_sched_barrier(0);                                     \
    } while (0)

    // prologue: e_0 .. e_6 = all of loop tile 0 and B0, A0, B1 of loop tile 1
    stage(J0{}, 0); stage(J1{}, 0); stage(J2{}, 0); stage(J3{}, 0);
    stage(J0{}, 1); stage(J1{}, 1); stage(J2{}, 1);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_b(bX, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();                                 // waves 4-7 run one barrier behind from here on

    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    auto ktile = [&](auto MXC, i32x8 (&b0)[2], i32x8 (&bn)[2], const int T) {
        read_a(T, 0);             stage(J3{}, T + 1);  T8_SYNC_IN();  mma(MXC, I0{}, I0{}, acc[0][0], b0);  T8_SYNC_OUT();
        read_b(bY, T, 1);         stage(J0{}, T + 2);  T8_SYNC_IN();  mma(MXC, I0{}, I1{}, acc[0][1], bY);  T8_SYNC_OUT();
        read_a(T, 1);             stage(J1{}, T + 2);  T8_SYNC_IN();  mma(MXC, I1{}, I1{}, acc[1][1], bY);  T8_SYNC_OUT();
        read_b(bn, T + 1, 0);     stage(J2{}, T + 2);  T8_SYNC_IN();  mma(MXC, I1{}, I0{}, acc[1][0], b0);  T8_SYNC_OUT();
    };
    // TWO loops, the half tiles first, then the fp8 tiles.  With both kinds inside ONE loop body (a branch per tile, or per phase around the
    // MFMAs) hipcc's register allocation fell apart -- 170 to 790 spilled registers, reloaded behind vmcnt(0) inside the ring; either kind alone
    // fits the 256.  The B0 fragments rotate bX -> bZ -> bX per pair of tiles, so the half loop must hand over after an EVEN number of tiles: the
    // half part of the tile numbering is padded to even with a dead tile (nk_main), and split-K slices start at even tiles (host).
    const int n1 = min(nk, max(0, nk_main - kt0));
    for (int T = 0; T < n1; T += 2) {
        ktile(std::false_type{}, bX, bZ, T);
        ktile(std::false_type{}, bZ, bX, T + 1);
    }
    for (int T = n1; T < nk; T += 2) {
        ktile(std::true_type{}, bX, bZ, T);
        if (T + 1 < nk) ktile(std::true_type{}, bZ, bX, T + 1);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();                                 // waves 0-3 wait for the others' last phase
#undef T8_SYNC_IN
#undef T8_SYNC_OUT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // the dead requests of the last phases (zeros into LDS) are drained
    __syncthreads();
    if constexpr (FUSE) {
        if (tail) {
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
                    tile_epilogue<2, 1, 32, float, 0, true, true>(g, reinterpret_cast<f32x16 (&)[2][1]>(acc[ha][hb]), smem, m0, n0, ha * 128 + wr * 64,
                                                                  hb * 128 + wc * 32, wave, lane, 0, true, nullptr, ksplit);
            return;
        }
    }
#pragma unroll
    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
            tile_epilogue<2, 1, 32, TOUT, 0, true, SLICE>(g, reinterpret_cast<f32x16 (&)[2][1]>(acc[ha][hb]), smem, m0, n0, ha * 128 + wr * 64,
                                                          hb * 128 + wc * 32, wave, lane, 0, split, nullptr, ksplit);
}

template <typename TOUT, bool SLICE>
__global__ __launch_bounds__(512) void gemm_mx_kernel(GemmMxArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 k-tiles][A | B]
    const int nwg = gridDim.x;
    const int lg = xcd_logical_id(blockIdx.y * nwg + blockIdx.x, nwg * (int)gridDim.y);
    const int tiles_m = (g.M + 255) / 256, tiles_n = (g.N + 255) / 256;
    const int ksplit = lg / nwg, bid = lg - ksplit * nwg;
    const int nk_all = ((g.K / BK + 1) & ~1) + 2 * ((g.K + 127) / 128);
    const int kt0 = ksplit * g.kt_per_split, kt1 = min(nk_all, kt0 + g.kt_per_split);
#ifndef OMLM_SUPER_ROWS
#define OMLM_SUPER_ROWS 1024
#endif
    constexpr int GROUP = OMLM_SUPER_ROWS / 256;
    const int gsz = GROUP * tiles_n;
    const int grp = bid / gsz, first_m = grp * GROUP;
    const int rows_in = min(GROUP, tiles_m - first_m);
    const int tm = first_m + (bid - grp * gsz) % rows_in, tn = (bid - grp * gsz) / rows_in;
    if (kt0 >= kt1) return;
    gemm_mx_body<TOUT, SLICE>(g, tm * 256, tn * 256, kt0, kt1, gridDim.y > 1, ksplit, smem);
}

template <typename TOUT, bool SLICE>
static int launch_mx(const GemmMxArgs& g, int splits, hipStream_t st) {
    constexpr size_t LDS = 2 * (size_t)(256 + 256) * BK * 2;
    const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    auto k = gemm_mx_kernel<TOUT, SLICE>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); attr = true; }
    hipLaunchKernelGGL(k, dim3(tiles, splits), dim3(512), LDS, st, g);
    return omlm_post_launch("omlm_gemm_mx16");
}

// The full-round tiles of `g` and the S k-slices of the peeled tail `t` (its C = the workspace, c_split_stride = one slice) in ONE grid: workgroups
// [0, main_tiles) are g's tiles in its XCD-aware order, the rest t's (tile, slice) pairs -- dispatched last, they fill the CUs the last round
// of g leaves idle instead of running as a launch of their own on a third of the machine (round 6: 41 us per FF GEMM).
struct GemmMxPair { GemmMxArgs g, t; int main_tiles, tail_tiles; };
template <typename TOUT>
__global__ __launch_bounds__(512) void gemm_mx_fused_kernel(GemmMxPair pr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool tail = (int)blockIdx.x >= pr.main_tiles;                      // (uniform)
    const GemmMxArgs& g = tail ? pr.t : pr.g;
    const int nwg = tail ? pr.tail_tiles : pr.main_tiles;
    const int lin = tail ? (int)blockIdx.x - pr.main_tiles : (int)blockIdx.x;
    const int total = tail ? (int)gridDim.x - pr.main_tiles : pr.main_tiles;
    const int lg = xcd_logical_id(lin, total);
    const int tiles_m = (g.M + 255) / 256, tiles_n = (g.N + 255) / 256;
    const int ksplit = lg / nwg, bid = lg - ksplit * nwg;
    const int nk_all = ((g.K / BK + 1) & ~1) + 2 * ((g.K + 127) / 128);
    const int kt0 = ksplit * g.kt_per_split, kt1 = min(nk_all, kt0 + g.kt_per_split);
    constexpr int GROUP = OMLM_SUPER_ROWS / 256;
    const int gsz = GROUP * tiles_n;
    const int grp = bid / gsz, first_m = grp * GROUP;
    const int rows_in = min(GROUP, tiles_m - first_m);
    const int tm = first_m + (bid - grp * gsz) % rows_in, tn = (bid - grp * gsz) / rows_in;
    if (kt0 >= kt1) return;
    gemm_mx_body<TOUT, false, true>(g, tm * 256, tn * 256, kt0, kt1, tail, ksplit, smem, tail);
}
template <typename TOUT>
static int launch_mx_fused(const GemmMxArgs& g, const GemmMxArgs& t, int S, hipStream_t st) {
    constexpr size_t LDS = 2 * (size_t)(256 + 256) * BK * 2;
    GemmMxPair pr;
    pr.g = g; pr.t = t;
    pr.main_tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    pr.tail_tiles = ((t.M + 255) / 256) * ((t.N + 255) / 256);
    auto k = gemm_mx_fused_kernel<TOUT>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); attr = true; }
    hipLaunchKernelGGL(k, dim3(pr.main_tiles + pr.tail_tiles * S), dim3(512), LDS, st, pr);
    return omlm_post_launch("omlm_gemm_mx16 (full rounds + tail slices)");
}
static bool mx_fuse_tail() { const char* e = getenv("OMLM_MX_FUSE_TAIL"); return !(e && e[0] == '0'); }      // (read per call: the kernel test toggles it)

static int mx_ncu() {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, n = 0;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return ncu;
}

// how the launch is cut: rows [0, M1) as whole machine rounds of 256 x 256 tiles, rows [M1, M) as S k-slices through the workspace
static void mx_plan(int M, int N, int K, long long ws_bytes, long long& M1, int& S, int& ktps) {
    const int ncu = mx_ncu();
    const int tiles_n = (N + 255) / 256, tiles_m = (M + 255) / 256, tiles = tiles_m * tiles_n;
    const int nk_all = ((K / BK + 1) & ~1) + 2 * ((K + 127) / 128);
    M1 = M; S = 1; ktps = nk_all;
    if (tiles <= ncu) return;
    const int rem = tiles % ncu;
    const int m_full = ((tiles / ncu) * ncu) / tiles_n;
    if (rem == 0 || m_full < 1 || m_full >= tiles_m) return;
    const int Mt = M - m_full * 256, Nw = (N + 3) / 4 * 4;
    const int tiles_t = ((Mt + 255) / 256) * tiles_n;
    int s = ncu / tiles_t;
    if (s > nk_all / 8) s = nk_all / 8;
    if (s > 8) s = 8;
    if (s < 2) return;
    const int per = ((nk_all + s - 1) / s + 1) & ~1;           // slices start at even loop tiles (gemm_mx_body's two loops)
    s = (nk_all + per - 1) / per;
    if (s < 2 || (long long)s * Mt * Nw * 4 > ws_bytes) return;
    M1 = (long long)m_full * 256; S = s; ktps = per;
}

}   // namespace OMLM_NS

using namespace OMLM_NS;

// Workspace the tail of an M x N x K launch wants (0: none); see omlm_gemm_mx16.
extern "C" long long omlm_gemm_mx16_workspace_bytes(int M, int N, int K) {
    long long M1; int S, ktps;
    mx_plan(M, N, K, (long long)1 << 62, M1, S, ktps);
    return S >= 2 ? (long long)S * (M - M1) * ((N + 3) / 4 * 4) * 4 : 0;
}

// C = A B^T (+ Cin) for IEEE-half operands given as planes (include/omlm.h).  A8 / B8: the operand's fp8 planes [hi8 | lo8] at the half plane's row
// pitch, the lo8 plane a8_stride / b8_stride bytes behind the hi8 plane, rows padded to a multiple of 256 (every byte of both planes readable;
// bytes [K, ceil128(K)) of every row zero).  C_lo != NULL: the result leaves as planes (no Cin) -- C the half hi plane, C_lo the lo plane as
// half (c_lo_bf8 == 0) or as bf8 bytes at the same element pitch (c_lo_bf8 != 0: GemmArgs::c_lo8); else fp32 (+ Cin).
// workspace: caller-owned, >= omlm_gemm_mx16_workspace_bytes(M, N, K) (or NULL: the tail runs unsplit); one per stream.
extern "C" int omlm_gemm_mx16(const void* A, const void* A8, long long a8_stride, const unsigned char* a_scale,
                              const void* B, const void* B8, long long b8_stride, const unsigned char* b_scale,
                              void* C, void* C_lo, int c_lo_bf8, const float* Cin, long long a_rows, long long b_rows,
                              int M, int N, int K, int lda, int ldb, int ldc, int ldcin,
                              void* workspace, long long workspace_bytes, void* stream) {
    if (M <= 0 || N <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(A && A8 && a_scale && B && B8 && b_scale && C, "null operand");
    OMLM_CHECK_ARG(K > 0 && K % BK == 0, "K must be a positive multiple of 64");
    OMLM_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0, "operand leading dimensions must be multiples of 8 elements");
    OMLM_CHECK_ARG(2 * lda >= (K + 127) / 128 * 128 && 2 * ldb >= (K + 127) / 128 * 128, "an fp8 plane row (pitch 2 * ld bytes) must hold K rounded up to 128 bytes");
    OMLM_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)A8 % 16) == 0 && ((uintptr_t)B8 % 16) == 0 &&
                   a8_stride % 16 == 0 && b8_stride % 16 == 0, "operands must be 16-byte aligned");
    OMLM_CHECK_ARG(a8_stride >= ((long long)M + 255) / 256 * 256 * lda * 2 && b8_stride >= ((long long)N + 255) / 256 * 256 * ldb * 2,
                   "an fp8 plane holds its rows padded to a multiple of 256");
    OMLM_CHECK_ARG((unsigned long long)a_rows * lda * 2 < 0xFFFFFFF0ull && (unsigned long long)b_rows * ldb * 2 < 0xFFFFFFF0ull &&
                   a8_stride < 0x7FFFFFF0ll && b8_stride < 0x7FFFFFF0ll, "operand exceeds the 4 GiB buffer-descriptor window");
    OMLM_CHECK_ARG(!(C_lo && Cin), "plane output takes no residual");
    OMLM_CHECK_ARG((workspace == nullptr) == (workspace_bytes == 0) && ((uintptr_t)workspace % 16) == 0, "workspace: 16-byte aligned buffer and its size, or NULL / 0");
    GemmMxArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.Cin = Cin; g.C_lo = C_lo; g.c_lo8 = (C_lo && c_lo_bf8) ? 1 : 0;
    OMLM_CHECK_ARG(!g.c_lo8 || (ldc % 8 == 0 && ((uintptr_t)C_lo % 8) == 0), "bf8 lo plane: 8-byte aligned, pitch a multiple of 8");
    g.a_rows = a_rows; g.b_rows = b_rows; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldcin = ldcin; g.alpha = 1.0f;
    g.A8 = A8; g.B8 = B8; g.a8_stride = (unsigned)a8_stride; g.b8_stride = (unsigned)b8_stride; g.a_scale = a_scale; g.b_scale = b_scale;
    const int nk_all = ((K / BK + 1) & ~1) + 2 * ((K + 127) / 128);
    g.kt_per_split = nk_all;
    hipStream_t st = as_stream(stream);
    long long M1; int S, ktps;
    mx_plan(M, N, K, workspace ? workspace_bytes : 0, M1, S, ktps);
    auto run = [&](const GemmMxArgs& ga) { return ga.C_lo ? launch_mx<h16pl_t, false>(ga, 1, st) : launch_mx<float, false>(ga, 1, st); };
    if (S < 2) return run(g);
    GemmMxArgs g1 = g, g2 = g;
    const size_t osz = C_lo ? 2 : 4;
    g1.M = (int)M1;
    g2.M = M - (int)M1;
    g2.A = (const char*)A + (size_t)M1 * lda * 2; g2.A8 = (const char*)A8 + (size_t)M1 * lda * 2;          // (both fp8 planes move: the stride stays)
    g2.a_scale = a_scale + M1;
    g2.a_rows = a_rows - M1;
    g2.C = (char*)C + (size_t)M1 * ldc * osz;
    if (C_lo) g2.C_lo = (char*)C_lo + (size_t)M1 * ldc * (g.c_lo8 ? 1 : osz);
    if (Cin) g2.Cin = Cin + (size_t)M1 * ldcin;
    const int Mt = g2.M, Nw = (N + 3) / 4 * 4;
    const long long slice = (long long)Mt * Nw;
    GemmMxArgs gw = g2;
    gw.C = workspace; gw.C_lo = nullptr; gw.Cin = nullptr; gw.ldc = Nw; gw.ldcin = 0; gw.c_split_stride = slice; gw.kt_per_split = ktps;
    int rc;
    // one grid where the last round of the full tiles leaves CUs idle (FF-in at B = 32: 3058 tiles = 11.95 rounds; the tail's 88 slices start on the
    // 14 idle CUs and finish ~20 us behind the round instead of 41 us as their own launch: 804 -> 781 us).  A main part of WHOLE rounds (FF-out:
    // 512 tiles) gains nothing from it (381 -> 392 us measured): two launches.  OMLM_MX_FUSE_TAIL=0: always two launches.
    const int main_tiles = (int)(M1 / 256) * ((N + 255) / 256);
    if (mx_fuse_tail() && main_tiles % mx_ncu() != 0) {
        rc = C_lo ? launch_mx_fused<h16pl_t>(g1, gw, S, st) : launch_mx_fused<float>(g1, gw, S, st);
        if (rc != OMLM_OK) return rc;
    } else {
        rc = run(g1);
        if (rc != OMLM_OK) return rc;
        rc = launch_mx<float, true>(gw, S, st);
        if (rc != OMLM_OK) return rc;
    }
    const long long quads = (long long)Mt * (Nw / 4);
    const int blocks = (int)((quads + 255) / 256 > 4096 ? 4096 : (quads + 255) / 256);
    if (C_lo && g.c_lo8) hipLaunchKernelGGL(gemm_tail_reduce_kernel<3>, dim3(blocks), dim3(256), 0, st, (const float*)workspace, S, slice, Mt, N, Nw, g2.C, g2.C_lo, ldc, (const float*)nullptr, 0);
    else if (C_lo) hipLaunchKernelGGL(gemm_tail_reduce_kernel<2>, dim3(blocks), dim3(256), 0, st, (const float*)workspace, S, slice, Mt, N, Nw, g2.C, g2.C_lo, ldc, (const float*)nullptr, 0);
    else      hipLaunchKernelGGL(gemm_tail_reduce_kernel<0>, dim3(blocks), dim3(256), 0, st, (const float*)workspace, S, slice, Mt, N, Nw, g2.C, (void*)nullptr, ldc, g2.Cin, ldcin);
    return omlm_post_launch("omlm_gemm_mx16 (tail reduce)");
}
#endif
