// MFMA GEMM for every dense contraction of the TokenConditionedTransformer path
// (to_q / to_kv / to_out, FF in/out, per-quantizer logit heads, and all their backward GEMMs).
//
//   C[m, n] = alpha * sum_k A(m, k) * B(n, k)  (+ Cin[m, n])
//
// A is either row-major [M, K] ("normal", k contiguous) or k-major [K, M] (A_KMAJ);
// B is either [N, K] (nn.Linear weight layout, k contiguous) or k-major [K, N] (B_KMAJ).
// With those two switches the forward (x W^T), the input-gradient (dY W) and the
// weight-gradient (dY^T X) contractions all read their operands exactly as the
// autograd graph leaves them in HBM -- no transposed copies are ever materialised.
// k-major tiles are staged row-for-row into LDS and fed to the matrix cores through the
// gfx950 hardware transpose read (ds_read_b64_tr_b16).
//
// Element type T selects the arithmetic:
//   T = bf16  : operands are bf16 in HBM, one v_mfma_f32_32x32x16_bf16 per k16 step  ("bf16")
//   T = float : operands are fp32 in HBM and are split on the fly into bf16 hi + bf16 lo
//               while being staged into LDS; hi*hi + hi*lo + lo*hi on the matrix cores
//               ("bf16x3": fp32-grade products at 1/3 of the bf16 MFMA rate, still ~5x the
//               fp32-MFMA rate).  Accumulation is fp32 in both modes.
//
// fp32 path (gemm_kernel): 128 x 128 x 64 per workgroup of 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 tiles;
// staging global -> registers (issued before the MFMA phase of the previous tile) -> split -> LDS after the barrier.
// LDS images are XOR-swizzled so that both ds_read_b128 (normal) and ds_read_b64_tr_b16 (k-major) fragment reads
// are conflict-free.
// Out-of-range rows/columns are read through a buffer descriptor with an out-of-bounds offset
// (hardware returns 0), so M, N and K may be ragged at 8-element granularity.
// Optional row maps (int32) gather A rows / k rows and scatter C rows: this is how the
// per-quantizer logit heads read the positions p = q (mod Q) of the hidden states in place.
//
// bf16 operands take the LDS-DMA path (gemm_bf16_tile_kernel: 256x256 / 256x128 / 128x128 tiles): tiles go HBM -> LDS directly with
// buffer_load_dwordx4 ... lds (no VGPR staging, no ds_write pass), the swizzle is applied on the
// per-lane SOURCE address (the DMA destination is lane-linear), LDS is double-buffered and there
// is ONE barrier per k-tile: the DMA of tile t+1 is in flight while tile t is on the matrix cores.
// fp32 operands (bf16x3) keep the register-staged path because they are split on the way in.
//
// Split-K: weight-gradient GEMMs have M x N = (out x in features) tiles only (32 tiles for to_q) but
// K = all tokens of the batch; gridDim.y splits K and the epilogue accumulates with fp32 atomics
// into C (these GEMMs are "C += ..." by construction: gradients accumulate over micro-batches).
#include "common.h"
#include "gemm_common.h"

namespace OMLM_NS {

// One workgroup's share of C = alpha A B^T (+ Cin).  lg: logical workgroup id inside this problem's (tiles x K-splits, split-major)
// space; split: partial sums are added to fp32 C with atomics; bal_wgs: workgroup count of the balanced split-K form (BAL only).
// FASTK: K is a multiple of the k-tile depth (host-checked): the DMA pieces take their k offset from an SGPR (DmaStagerT::issue_one)
template <int BM_, int BN_, int WM_, int WN_, bool A_KMAJ, bool B_KMAJ, typename TOUT, bool DBG, bool KMAP, bool BAL, bool SPLIT3 = false, bool FASTK = false, int EPI = 0>
__device__ __forceinline__ void gemm_tile_body(const GemmArgs& g, const int lg, const bool split, const int bal_wgs, char* smem) {
    const int dbg = DBG ? g.debug : 0;      // ablation switches exist only in the DBG instantiation (OMLM_GEMM_DEBUG set)
    constexpr int NWN = BN_ / WN_, NWAVES = (BM_ / WM_) * NWN;
    constexpr int MI = WM_ / 32, NJ = WN_ / 32;
    constexpr int A_BYTES = BM_ * BK * 2, B_BYTES = BN_ * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int UA = (BM_ / 8) / NWAVES, UB = (BN_ / 8) / NWAVES;   // DMA wave-instructions per k-tile per wave

    const int tiles_m = (g.M + BM_ - 1) / BM_, tiles_n = (g.N + BN_ - 1) / BN_;
    const int nwg = tiles_m * tiles_n;
    const int nk1 = (g.K + BK - 1) / BK;
    const int nk_all = SPLIT3 ? 3 * nk1 : nk1;            // split3: k-tile t reads planes (t / nk1) at k = (t % nk1) * BK
    constexpr bool bal = BAL;          // balanced split-K is its own instantiation: the plain kernels keep their code and registers
    // Balanced split-K: the (K chunk, tile, k-tile) units are numbered chunk-major and cut into bal_wgs equal contiguous
    // ranges; a workgroup walks its range segment by segment (a segment = consecutive k-tiles of one output tile) and adds each
    // partial tile to C.  Every workgroup does the same number of k-tiles (no partial last round), co-resident workgroups sit
    // in the same K chunk (shared panels), and a tile receives ~chunks + 1 partial sums instead of one per split.
    int u = 0, u1 = 0;                 // host guarantees tiles x k-tiles < 2^31 / workgroups
    if (bal) {
        const long long U = (long long)nwg * nk_all;
        u = (int)(U * lg / bal_wgs);
        u1 = (int)(U * (lg + 1) / bal_wgs);
        if (u >= u1) return;
    }
#ifndef OMLM_SUPER_ROWS
#define OMLM_SUPER_ROWS 1024       /* C rows per super-tile (tile rows walked column-major inside it) */
#endif
    constexpr int GROUP = OMLM_SUPER_ROWS / BM_;                       // tile rows per super-tile (same footprint as the 128 kernel's 8)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave / NWN) * WM_, wn = (wave % NWN) * WN_;
    const dma_rsrc rsA = make_dma_rsrc(g.A, (unsigned long long)g.a_rows * g.lda * 2);
    const dma_rsrc rsB = make_dma_rsrc(g.B, (unsigned long long)g.b_rows * g.ldb * 2);
    const dma_rsrc rsAl = make_dma_rsrc(SPLIT3 ? g.A_lo : g.A, (unsigned long long)g.a_rows * g.lda * 2);      // (plain kernels: never used)
    const dma_rsrc rsBl = make_dma_rsrc(SPLIT3 ? g.B_lo : g.B, (unsigned long long)g.b_rows * g.ldb * 2);

    for (;;) {
        int bid, kt0, kt1, ksplit_id = 0;
        if (bal) {
            const int per_chunk = nwg * g.bal_ck;
            int c = u / per_chunk;
            if (c > g.bal_chunks - 1) c = g.bal_chunks - 1;
            const int len_c = min(g.bal_ck, nk_all - c * g.bal_ck);
            const int r = u - c * per_chunk;
            bid = r / len_c;
            const int kk = r - bid * len_c;
            const int seg_end = min(u1, u + (len_c - kk));
            kt0 = c * g.bal_ck + kk;
            kt1 = kt0 + (seg_end - u);
            u = seg_end;
        } else {
            const int ksplit = lg / nwg;
            ksplit_id = ksplit;
            bid = lg - ksplit * nwg;
            kt0 = ksplit * g.kt_per_split;
            kt1 = min(nk_all, kt0 + g.kt_per_split);
        }
        const int gsz = GROUP * tiles_n;
        const int grp = bid / gsz, first_m = grp * GROUP;
        const int rows_in = min(GROUP, tiles_m - first_m);
        const int tm = first_m + (bid - grp * gsz) % rows_in, tn = (bid - grp * gsz) / rows_in;
        const int m0 = tm * BM_, n0 = tn * BN_;

        DmaStagerT<A_KMAJ, BM_, NWAVES> sa;
        DmaStagerT<B_KMAJ, BN_, NWAVES> sb;
        sa.init(g.a_map, g.lda, g.M, m0, wave, lane);
        sb.init(g.b_map, g.ldb, g.N, n0, wave, lane);

        f32x16 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // k-tile index -> contraction offset and operand planes
        // (SPLIT3 is its own instantiation: the plain kernels keep the code and register allocation they were measured with)
        // pa / pb: the k-tile reads A's / B's lo plane (uniform: the descriptor is chosen by scalar selects)
        auto tile_at = [&](int t, bool& pa, bool& pb) -> int {
            if constexpr (SPLIT3) {
                const int which = t >= 2 * nk1 ? 2 : (t >= nk1 ? 1 : 0);
                pa = which == 2;
                pb = which == 1;
                return (t - which * nk1) * BK;
            } else {
                pa = false; pb = false;
                return t * BK;
            }
        };
        if (kt0 < kt1) {
            bool pa, pb;
            const int k00 = tile_at(kt0, pa, pb);
            sa.template issue<KMAP, FASTK>(SPLIT3 && pa ? rsAl : rsA, g.a_map, g.lda, k00, g.K, smem, wave);
            sb.template issue<KMAP, FASTK>(SPLIT3 && pb ? rsBl : rsB, g.b_map, g.ldb, k00, g.K, smem + A_BYTES, wave);
        }
        // Rotated k-loop: the MFMAs of a tile's LAST k16 step run AFTER the next tile's barrier and first fragment reads (their
        // operands are in registers), so the matrix pipe has work while the barrier releases and the first LDS reads of the new
        // tile are in flight; the next tile's DMA issue is spread over that deferred step and steps 0, 1, and step 2 covers its
        // latency.  Same products in the same order per accumulator.
        {
            h16x8 a[2][MI], b[2][NJ];
            bool pending = false;                                                  // a[1], b[1] hold an unmultiplied last step
            constexpr int MPS = MI * NJ, NLOAD = UA + UB;
            constexpr int SPREAD = 2;          // phases (of 4) over which the next tile's DMA issue is spread: 2 measured best (1: -5 %, 3: -1..+3 %, 4: worse)
            constexpr int STRIDE = (SPREAD * MPS) / NLOAD > 0 ? (SPREAD * MPS) / NLOAD : 1;
            for (int kt = kt0; kt < kt1; ++kt) {
                const int cur = (kt - kt0) & 1;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();      // tile kt landed everywhere; every wave's reads of the other stage (incl. its last step) are complete
                const bool live = kt + 1 < kt1 && !(dbg & 1);        // dbg (DBG instantiation only): bit 0 = no DMA after the first tile, bit 1 = no MFMAs
                char* nxt = smem + (cur ^ 1) * STAGE;
                bool pan, pbn;
                const int knext = tile_at(kt + 1, pan, pbn);
                const dma_rsrc rsAn = SPLIT3 && pan ? rsAl : rsA, rsBn = SPLIT3 && pbn ? rsBl : rsB;
                const char* As = smem + cur * STAGE;
                const char* Bs = As + A_BYTES;
                // phase ph in 0..3 = (deferred last step of tile kt-1, step 0, step 1, step 2) of this iteration; fi = fragment set
                auto phase = [&](const int ph, const int fi, const bool mul) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            if (mul && !(dbg & 2)) acc[i][j] = OMLM_MFMA_32x32x16(a[fi][i], b[fi][j], acc[i][j]);
                            else if (mul) asm volatile("" :: "v"(a[fi][i]), "v"(b[fi][j]));
                            const int midx = ph * MPS + i * NJ + j;
                            if (midx % STRIDE == 0 && midx / STRIDE < NLOAD) {
                                const int l = midx / STRIDE;
                                __builtin_amdgcn_sched_barrier(0);
                                if (l < UA) sa.template issue_one<KMAP, FASTK>(l, rsAn, g.a_map, g.lda, knext, g.K, nxt, wave, live);
                                else        sb.template issue_one<KMAP, FASTK>(l - UA, rsBn, g.b_map, g.ldb, knext, g.K, nxt + A_BYTES, wave, live);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                };
#pragma unroll
                for (int i = 0; i < MI; ++i) a[0][i] = read_frag<A_KMAJ>(As, wm + 32 * i, 0, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[0][j] = read_frag<B_KMAJ>(Bs, wn + 32 * j, 0, lane);
                __builtin_amdgcn_sched_barrier(0);
                phase(0, 1, pending);                                              // last step of the previous tile (registers only)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int st = 0; st < 3; ++st) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) a[(st + 1) & 1][i] = read_frag<A_KMAJ>(As, wm + 32 * i, st + 1, lane);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) b[(st + 1) & 1][j] = read_frag<B_KMAJ>(Bs, wn + 32 * j, st + 1, lane);
                    __builtin_amdgcn_sched_barrier(0);
                    phase(st + 1, st & 1, true);
                    __builtin_amdgcn_sched_barrier(0);
                }
                pending = true;
            }
            if (pending) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = OMLM_MFMA_32x32x16(a[1][i], b[1][j], acc[i][j]);
            }
        }
        // The last iteration issued its "next tile" DMA pieces with out-of-bounds offsets (`live` false): no memory traffic, but the
        // hardware still writes their ZEROS into the other LDS stage -- the stage the epilogue below re-uses as its transpose patch
        // (smem + 0 .. for an even k-tile count, e.g. K = 1024 / 512).  They are inline asm, i.e. invisible to hipcc's vmcnt
        // bookkeeping, so without this wait nothing orders them before the epilogue's ds_writes: a late zero-fill wiped staged output
        // values (round 3: intermittent 30-70 % error in the rel-pos MLP's gradient -- its GEMMs run on the second stream next to the
        // trunk's HBM-bound kernels, whose traffic delays the pieces past the ~500 cycles the remaining MFMAs of a 128x128 tile take).
        // -DOMLM_GEMM_TAIL_WAIT=0 rebuilds the old behaviour (tests/stress_gemm_tail.py reproduces the defect with it).
#if OMLM_GEMM_TAIL_WAIT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        constexpr bool SLICE = BM_ == 128 && BN_ == 128 && std::is_same<TOUT, float>::value && !BAL && EPI == 0;
        tile_epilogue<MI, NJ, WN_, TOUT, EPI, (OMLM_EPI_CIN_AHEAD != 0) && (MI * NJ <= 4), SLICE>(g, acc, smem, m0, n0, wm, wn, wave, lane, dbg, bal || split, nullptr,
                                                                                                   SLICE ? ksplit_id : 0);
        if (!bal || u >= u1) break;
        __syncthreads();              // the non-split epilogue stages through LDS; the next segment's DMA must not overtake it
    }
}

// For split-K GEMMs the split-major XCD order puts all co-resident workgroups of an XCD on the SAME K range (they share A and B
// panels through its L2); with a tile-only remap an XCD held 3 unrelated K ranges at a time (measured L2 hit 48 % on the dW1 GEMM).
template <int BM_, int BN_, int WM_, int WN_, bool A_KMAJ, bool B_KMAJ, typename TOUT, bool DBG, bool KMAP, bool BAL = false, bool SPLIT3 = false, bool FASTK = false, int EPI = 0>
__global__ __launch_bounds__((BM_ / WM_) * (BN_ / WN_) * 64) void gemm_bf16_tile_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][A | B]
    const int nwg = gridDim.x;                                     // tiles (plain) / workgroups (balanced)
    const int total = BAL ? (int)gridDim.x : nwg * (int)gridDim.y;
    const int lg = xcd_logical_id(blockIdx.y * nwg + blockIdx.x, total);
    gemm_tile_body<BM_, BN_, WM_, WN_, A_KMAJ, B_KMAJ, TOUT, DBG, KMAP, BAL, SPLIT3, FASTK, EPI>(g, lg, gridDim.y > 1, (int)gridDim.x, smem);
}


// ---- persistent form of the wide tile kernel -----------------------------------------------------------------------------
// The 256x256 tiles run ONE workgroup per CU (128 KiB of LDS), so nothing overlaps a workgroup's prologue (first k-tile: a full L2 / HBM
// round trip), its epilogue and the dispatch of its successor: ~12 % of a K = 1024 tile (16 k-tiles), 32.5 us per tile against 28.6 us
// at the rate of the k-loop (round 4).  Here G = gridDim.x workgroups (one per CU, G a multiple of 8) each walk the tiles
// lin, lin + G, lin + 2G, ... of the XCD-aware order (lin & 7 is constant, so the walk stays inside its XCD's chunk and the 32 workgroups
// of an XCD move through consecutive super-tile patches together), and the LAST k-iteration of a tile requests the FIRST k-tile of the
// next one -- in the slots where the one-tile kernel issues its dead pieces -- so that tile's first round trip runs under this tile's
// epilogue.  The request lands in the stage the last k-tile did not use; the epilogue's transpose patches therefore live in the
// stage of the last k-tile (seven waves) and in 8.5 KiB behind the two stages (the eighth), never under a landing DMA piece.
// Order: [k-loop ... last iteration: DMA(next tile, k-tile 0) -> stage X] barrier [epilogue: patches in stage X^1 / behind] [next
// tile, iteration 0: vmcnt wait + barrier, DMA(k-tile 1) -> stage X^1, multiply from stage X].  The barrier of iteration 0 is what
// orders every wave's patch traffic before the first piece of k-tile 1 lands in stage X^1.
// Only the shape the host sends here is built: whole k-tiles (K % 64 == 0), no row / k-row / scatter maps, no split-K, plain epilogue.
#ifndef OMLM_PERSIST_COUNTED
#define OMLM_PERSIST_COUNTED 1     // 1: iteration 0 of a follow-up tile waits for its DMA pieces only (counted), not for the epilogue's stores behind them
#endif
// CIN: the epilogue adds a residual.  It is a template switch because the residual loads are what hipcc has to protect in the NEXT
// tile's first iteration (their registers are re-used; paths that skip a row leave a load unconsumed): with them in the code,
// iteration 0 carries compiler waits that drain the epilogue's stores whatever the run-time pointer is (seen in the ISA).
template <int BM_, int BN_, int WM_, int WN_, bool A_KMAJ, bool B_KMAJ, typename TOUT, bool CIN, int EPI = 0>
__global__ __launch_bounds__((BM_ / WM_) * (BN_ / WN_) * 64) void gemm_bf16_tile_persist_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][A | B] + one transpose patch
    constexpr int NWN = BN_ / WN_, NWAVES = (BM_ / WM_) * NWN;
    constexpr int MI = WM_ / 32, NJ = WN_ / 32;
    constexpr int A_BYTES = BM_ * BK * 2, B_BYTES = BN_ * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int UA = (BM_ / 8) / NWAVES, UB = (BN_ / 8) / NWAVES;
    constexpr int PATCH = 32 * (WN_ + 4) * 4;                      // one wave's transpose patch (tile_epilogue: 32 rows x SROW floats)
    static_assert((NWAVES - 1) * PATCH <= STAGE, "all but the last wave's patch must fit inside one stage");
    constexpr int GROUP = OMLM_SUPER_ROWS / BM_;
    constexpr int VEC = sizeof(TOUT) == 2 ? 8 : 4, NSTORE = MI * (32 / (64 / (WN_ / VEC)));   // 16-byte stores per wave of a full tile's epilogue

    const int tiles_m = (g.M + BM_ - 1) / BM_, tiles_n = (g.N + BN_ - 1) / BN_;
    const int total = tiles_m * tiles_n;
    const int nk = g.K / BK;                                       // host: K % 64 == 0, K > 0
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave / NWN) * WM_, wn = (wave % NWN) * WN_;
    const dma_rsrc rsA = make_dma_rsrc(g.A, (unsigned long long)g.a_rows * g.lda * 2);
    const dma_rsrc rsB = make_dma_rsrc(g.B, (unsigned long long)g.b_rows * g.ldb * 2);
    const unsigned smem_lds = (unsigned)(size_t)LDS_PTR(char, smem);

    auto origin = [&](int lin, int& m0, int& n0) {                 // the one-tile kernel's order (xcd_logical_id + super-tiles)
        const int bid = xcd_logical_id(lin, total);
        const int gsz = GROUP * tiles_n;
        const int grp = bid / gsz, first_m = grp * GROUP;
        const int rows_in = min(GROUP, tiles_m - first_m);
        m0 = (first_m + (bid - grp * gsz) % rows_in) * BM_;
        n0 = ((bid - grp * gsz) / rows_in) * BN_;
    };
    // one 1-KiB piece of k-tile k0 (same address form as DmaStagerT::issue_one<false, true>)
    auto piece = [&](int l, const unsigned (&va)[UA], const unsigned (&vb)[UB], int k0, unsigned stage_lds, bool live) {
        if (l < UA) {
            const int b = wave + NWAVES * l;
            dma_issue_s(rsA, stage_lds + (unsigned)(b * 1024), live ? va[l] : OOB_OFF,
                        A_KMAJ ? (unsigned)k0 * (unsigned)(g.lda * 2) : (unsigned)(k0 * 2));
        } else {
            const int b = wave + NWAVES * (l - UA);
            dma_issue_s(rsB, stage_lds + (unsigned)A_BYTES + (unsigned)(b * 1024), live ? vb[l - UA] : OOB_OFF,
                        B_KMAJ ? (unsigned)k0 * (unsigned)(g.ldb * 2) : (unsigned)(k0 * 2));
        }
    };

    int lin = blockIdx.x;
    const int G = gridDim.x;
    if (lin >= total) return;
    int m0, n0;
    origin(lin, m0, n0);
    unsigned va[UA], vb[UB];                                       // per-lane byte offsets of this tile's pieces at k = 0
    {
        DmaStagerT<A_KMAJ, BM_, NWAVES> sa;
        DmaStagerT<B_KMAJ, BN_, NWAVES> sb;
        sa.init(nullptr, g.lda, g.M, m0, wave, lane);
        sb.init(nullptr, g.ldb, g.N, n0, wave, lane);
#pragma unroll
        for (int i = 0; i < UA; ++i) va[i] = sa.vfast[i];
#pragma unroll
        for (int i = 0; i < UB; ++i) vb[i] = sb.vfast[i];
    }
    int par = 0;                                                   // stage of the current tile's k-tile 0
#pragma unroll
    for (int l = 0; l < UA + UB; ++l) piece(l, va, vb, 0, smem_lds, true);
    bool counted = false;                                          // the pieces of k-tile 0 are older than >= NSTORE stores of the previous epilogue

    for (;;) {
        const int lin_n = lin + G;
        const bool has_next = lin_n < total;
        int m0n = m0, n0n = n0;
        if (has_next) origin(lin_n, m0n, n0n);
        unsigned van[UA], vbn[UB];
        {
            DmaStagerT<A_KMAJ, BM_, NWAVES> sa;
            DmaStagerT<B_KMAJ, BN_, NWAVES> sb;
            sa.init(nullptr, g.lda, g.M, m0n, wave, lane);
            sb.init(nullptr, g.ldb, g.N, n0n, wave, lane);
#pragma unroll
            for (int i = 0; i < UA; ++i) van[i] = sa.vfast[i];
#pragma unroll
            for (int i = 0; i < UB; ++i) vbn[i] = sb.vfast[i];
        }
        f32x16 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        {   // the rotated k-loop of gemm_tile_body (same products in the same order per accumulator)
            h16x8 a[2][MI], b[2][NJ];
            bool pending = false;
            constexpr int MPS = MI * NJ, NLOAD = UA + UB;
            constexpr int SPREAD = 2;
            constexpr int STRIDE = (SPREAD * MPS) / NLOAD > 0 ? (SPREAD * MPS) / NLOAD : 1;
            for (int kt = 0; kt < nk; ++kt) {
                const int cur = (par + kt) & 1;
                if (OMLM_PERSIST_COUNTED && kt == 0 && counted) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSTORE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const bool last = kt + 1 == nk;
                const bool live = !last || has_next;
                const unsigned nxt = smem_lds + (unsigned)((cur ^ 1) * STAGE);
                const int knext = last ? 0 : (kt + 1) * BK;
                const char* As = smem + cur * STAGE;
                const char* Bs = As + A_BYTES;
                auto phase = [&](const int ph, const int fi, const bool mul) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            if (mul) acc[i][j] = OMLM_MFMA_32x32x16(a[fi][i], b[fi][j], acc[i][j]);
                            const int midx = ph * MPS + i * NJ + j;
                            if (midx % STRIDE == 0 && midx / STRIDE < NLOAD) {
                                const int l = midx / STRIDE;
                                __builtin_amdgcn_sched_barrier(0);
                                unsigned v;
                                if (l < UA) v = last ? van[l] : va[l]; else v = last ? vbn[l - UA] : vb[l - UA];
                                {
                                    unsigned one_a[UA], one_b[UB];
#pragma unroll
                                    for (int x = 0; x < UA; ++x) one_a[x] = v;
#pragma unroll
                                    for (int x = 0; x < UB; ++x) one_b[x] = v;
                                    piece(l, one_a, one_b, knext, nxt, live);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                };
#pragma unroll
                for (int i = 0; i < MI; ++i) a[0][i] = read_frag<A_KMAJ>(As, wm + 32 * i, 0, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[0][j] = read_frag<B_KMAJ>(Bs, wn + 32 * j, 0, lane);
                __builtin_amdgcn_sched_barrier(0);
                phase(0, 1, pending);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int st = 0; st < 3; ++st) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) a[(st + 1) & 1][i] = read_frag<A_KMAJ>(As, wm + 32 * i, st + 1, lane);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) b[(st + 1) & 1][j] = read_frag<B_KMAJ>(Bs, wn + 32 * j, st + 1, lane);
                    __builtin_amdgcn_sched_barrier(0);
                    phase(st + 1, st & 1, true);
                    __builtin_amdgcn_sched_barrier(0);
                }
                pending = true;
            }
            if (pending) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = OMLM_MFMA_32x32x16(a[1][i], b[1][j], acc[i][j]);
            }
        }
        const int cur_last = (par + nk - 1) & 1;
        // Last tile of the walk: its final iteration issued dead pieces (zeros) into the other stage, which nothing reads or patches again;
        // they are waited for all the same, so that the kernel never ends with LDS writes in flight.
        if (!has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                           // every wave's fragment reads of the last k-tile's stage are complete
        float* patch = wave < NWAVES - 1 ? (float*)(smem + cur_last * STAGE + wave * PATCH) : (float*)(smem + 2 * STAGE);
        GemmArgs ge = g;
        ge.c_map = nullptr;                                        // host: no scatter map on this route
        if constexpr (!CIN) ge.Cin = nullptr;
        tile_epilogue<MI, NJ, WN_, TOUT, EPI, false>(ge, acc, smem, m0, n0, wm, wn, wave, lane, 0, false, patch);    // (no second residual set: registers)
        if (!has_next) break;
        // a full interior tile's epilogue issued exactly NSTORE 16-byte stores per wave behind the next tile's pieces (more on the scalar
        // path): `vmcnt(NSTORE)` then covers the pieces; an edge tile skips stores, so its successor drains everything
        counted = m0 + BM_ <= g.M && n0 + BN_ <= g.N;
        par = (par + nk) & 1;
        lin = lin_n; m0 = m0n; n0 = n0n;
#pragma unroll
        for (int i = 0; i < UA; ++i) va[i] = van[i];
#pragma unroll
        for (int i = 0; i < UB; ++i) vb[i] = vbn[i];
    }
}


// ---- round 5: the wide tile on a half-tile ring with counted waits (two staggered wave groups) ----------------------------------
// What the k-loops above still pay (rounds 1-4): ONE barrier and a full `s_waitcnt vmcnt(0)` per k-tile with ONE tile of DMA in flight -- the
// youngest piece of k-tile t + 1 has ~1.5 of tile t's 4 k16 steps to land, so every L2 miss (and, under load, many hits) is exposed, and
// after each barrier all eight waves of the one workgroup per CU wait for their first fragments together (MFMA pipe busy 47 %).
// This body keeps the 256 x 256 x 64 tile, the LDS images and the fragment reads, and changes the schedule:
//   * the unit of staging is a HALF-tile (128 rows or columns x 64 k = 16 KiB = two 1-KiB DMA pieces per wave); LDS holds two k-tiles =
//     8 half-tile slots, and a slot is re-requested two phases after its last fragment read -- FIVE half-tiles (80 KiB) are in flight
//     behind a counted `s_waitcnt vmcnt(10)` per phase, never 0: every piece has ~5 phases (~1300 MFMA cycles) to land;
//   * a k-tile is four PHASES, one 64 x 32 quadrant of the wave's output per phase (8 MFMAs = 256 matrix-pipe cycles): the wave owns
//     rows ha * 128 + wr * 64 + [0, 64) and columns hb * 128 + wc * 32 + [0, 32) for ha, hb in {0, 1}, so quadrant (ha, hb) needs exactly
//     half-tiles A_ha and B_hb; quadrant order (0,0) (0,1) (1,1) (1,0): a phase reads ONE new half (A0: 8 ds_read_b128, B1: 4, A1: 8, and
//     in the last phase B0 of the NEXT k-tile: 4 -- the B fragments rotate through three register sets), requests one half-tile, waits
//     (counted) for the half the next phase reads, `s_barrier`, multiplies, `s_barrier`;
//   * waves 4-7 run ONE barrier behind waves 0-3: on every SIMD one wave is on the matrix pipe while its partner reads fragments and
//     issues DMA -- the fragment latency is covered by the partner's MFMAs instead of by a rotated loop.
// Event order (e_n: 4T = B0(T), 4T+1 = A0(T), 4T+2 = B1(T), 4T+3 = A1(T)): e_n is read in phase n - 1 and requested in phase n - 7
// (prologue: e_0 .. e_6); at the wait of phase g, e_(g+2) must have landed and e_(g+3) .. e_(g+7) stay in flight: vmcnt(10).
//   RAW: a wave's own pieces are covered by its counted wait, the other waves' by the barrier(s) behind their waits -- group 1's wait of
//        phase g sits one barrier later than group 0's, still in front of the barrier that opens group 0's phase g + 1 reads;
//   WAR: e_(g+7) overwrites the slot last read in phase g - 2 (A0: 4T -> requested in 4T + 2, etc.): both groups' reads of that phase have
//        been retired by their `lgkmcnt(0)` behind the phase's first barrier, two and one barriers before the request respectively.
// Requests past the last k-tile are issued dead (out-of-bounds offset: zeros, no traffic) so that the counts stay uniform; they are
// drained in front of the epilogue (the tail race of round 4).  Whole k-tiles only (K % 64 == 0), no row maps on the operands.
#ifndef OMLM_T8_SETPRIO
#define OMLM_T8_SETPRIO 1
#endif
template <bool A_KMAJ, bool B_KMAJ, typename TOUT, bool SPLIT3 = false>
__device__ __forceinline__ void gemm_tile8_body(const GemmArgs& g, const int m0, const int n0, const int kt0, const int kt1, const bool split, char* smem) {
    constexpr int A_BYTES = 256 * BK * 2, STAGE = 2 * A_BYTES;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const dma_rsrc rsA = make_dma_rsrc(g.A, (unsigned long long)g.a_rows * g.lda * 2);
    const dma_rsrc rsB = make_dma_rsrc(g.B, (unsigned long long)g.b_rows * g.ldb * 2);
    const dma_rsrc rsAl = make_dma_rsrc(SPLIT3 ? g.A_lo : g.A, (unsigned long long)g.a_rows * g.lda * 2);      // (plain kernels: never used)
    const dma_rsrc rsBl = make_dma_rsrc(SPLIT3 ? g.B_lo : g.B, (unsigned long long)g.b_rows * g.ldb * 2);
    DmaStagerT<A_KMAJ, 256, 8> sa;          // unit i of a wave is 1-KiB unit b = wave + 8 i of the tile image: i < 2 -> rows / columns [0, 128) = half 0
    DmaStagerT<B_KMAJ, 256, 8> sb;
    sa.init(nullptr, g.lda, g.M, m0, wave, lane);
    sb.init(nullptr, g.ldb, g.N, n0, wave, lane);
    const int nk = kt1 - kt0;
    const unsigned smem_lds = (unsigned)(size_t)LDS_PTR(char, smem);

    f32x16 acc[2][2][2];                    // [ha][hb][i]: rows ha * 128 + wr * 64 + 32 i, columns hb * 128 + wc * 32
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[x >> 2][(x >> 1) & 1][x & 1][e] = 0.f;

    // request event e_n, n = 4 T + J (J a compile-time constant): two pieces per wave
    auto stage = [&](auto JC, const int T) {
        constexpr int J = decltype(JC)::value;
        constexpr bool isA = (J & 1) != 0;
        constexpr int h = J >> 1;
        const bool live = T < nk;
        unsigned k0 = (unsigned)(kt0 + T) * BK;
        bool lo = false;                        // this event reads its operand's lo plane (uniform)
        if constexpr (SPLIT3) {
            // hi/lo operand planes ("bf16x3"): loop tile t reads planes (A_hi, B_hi), (A_hi, B_lo), (A_lo, B_hi) at k = (t mod nk1) * 64 -- gemm_tile_body's tile_at
            const int t = kt0 + T, nk1 = g.K / BK;
            const int which = t >= 2 * nk1 ? 2 : (t >= nk1 ? 1 : 0);
            k0 = (unsigned)(t - which * nk1) * BK;
            lo = isA ? which == 2 : which == 1;
        }
        const dma_rsrc rs = isA ? (SPLIT3 && lo ? rsAl : rsA) : (SPLIT3 && lo ? rsBl : rsB);
        if constexpr (SPLIT3) k0 = (unsigned)__builtin_amdgcn_readfirstlane((int)k0);      // (hipcc keeps the plane arithmetic on the VALU: the offset must reach the DMA in an SGPR)
        const unsigned base = smem_lds + (unsigned)((T & 1) * STAGE) + (isA ? 0u : (unsigned)A_BYTES);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = 2 * h + u;
            const int b = wave + 8 * i;
            if constexpr (isA) dma_issue_s(rs, base + (unsigned)(b * 1024), live ? sa.vfast[i] : OOB_OFF, A_KMAJ ? k0 * (unsigned)(g.lda * 2) : k0 * 2u);
            else               dma_issue_s(rs, base + (unsigned)(b * 1024), live ? sb.vfast[i] : OOB_OFF, B_KMAJ ? k0 * (unsigned)(g.ldb * 2) : k0 * 2u);
        }
    };
    using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>;
    using J2 = std::integral_constant<int, 2>; using J3 = std::integral_constant<int, 3>;

    h16x8 a[2][4], bX[4], bY[4], bZ[4];
    auto read_a = [&](const int T, const int ha) {
        const char* As = smem + (T & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int st = 0; st < 4; ++st) a[i][st] = read_frag<A_KMAJ>(As, ha * 128 + wr * 64 + 32 * i, st, lane);
    };
    auto read_b = [&](h16x8 (&b)[4], const int T, const int hb) {
        const char* Bs = smem + (T & 1) * STAGE + A_BYTES;
#pragma unroll
        for (int st = 0; st < 4; ++st) b[st] = read_frag<B_KMAJ>(Bs, hb * 128 + wc * 32, st, lane);
    };
    auto mma = [&](f32x16 (&c)[2], const h16x8 (&b)[4]) {
#if OMLM_T8_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int i = 0; i < 2; ++i) c[i] = OMLM_MFMA_32x32x16(a[i][st], b[st], c[i]);
#if OMLM_T8_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    // the synchronisation of one phase: (reads and the request were just issued) counted wait -> barrier -> fragments in -> MFMAs -> barrier
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

    // prologue: e_0 .. e_6 = all of k-tile 0 and B0, A0, B1 of k-tile 1
    stage(J0{}, 0); stage(J1{}, 0); stage(J2{}, 0); stage(J3{}, 0);
    stage(J0{}, 1); stage(J1{}, 1); stage(J2{}, 1);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                          // e_0 (B0 of k-tile 0) and e_1 (A0) are in
    __builtin_amdgcn_s_barrier();
    read_b(bX, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // (retired before anyone may re-request that slot: phase 1)
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();                                 // waves 4-7 run one barrier behind from here on

    // one k-tile: b0 holds B0 of this tile (read in the previous tile's last phase), bn receives B0 of the next tile
    auto ktile = [&](h16x8 (&b0)[4], h16x8 (&bn)[4], const int T) {
        read_a(T, 0);             stage(J3{}, T + 1);  T8_SYNC_IN();  mma(acc[0][0], b0);  T8_SYNC_OUT();     // phase 0: A0 -> (0, 0); request A1(T + 1)
        read_b(bY, T, 1);         stage(J0{}, T + 2);  T8_SYNC_IN();  mma(acc[0][1], bY);  T8_SYNC_OUT();     // phase 1: B1 -> (0, 1); request B0(T + 2)
        read_a(T, 1);             stage(J1{}, T + 2);  T8_SYNC_IN();  mma(acc[1][1], bY);  T8_SYNC_OUT();     // phase 2: A1 -> (1, 1); request A0(T + 2)
        read_b(bn, T + 1, 0);     stage(J2{}, T + 2);  T8_SYNC_IN();  mma(acc[1][0], b0);  T8_SYNC_OUT();     // phase 3: B0(T + 1) for later; (1, 0); request B1(T + 2)
    };
    for (int T = 0; T < nk; T += 2) {
        ktile(bX, bZ, T);
        if (T + 1 < nk) ktile(bZ, bX, T + 1);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();                                 // waves 0-3 wait for the others' last phase
#undef T8_SYNC_IN
#undef T8_SYNC_OUT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // the dead requests of the last phases (zeros into LDS) are drained
    __syncthreads();
    // epilogue: the wave's four 64 x 32 quadrants through its LDS patch (tile_epilogue, 32-column form)
#pragma unroll
    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
            tile_epilogue<2, 1, 32, TOUT>(g, reinterpret_cast<f32x16 (&)[2][1]>(acc[ha][hb]), smem, m0, n0, ha * 128 + wr * 64, hb * 128 + wc * 32,
                                          wave, lane, 0, split);
}

// one output tile per workgroup, the tile / split order of gemm_bf16_tile_kernel
template <bool A_KMAJ, bool B_KMAJ, typename TOUT, bool SPLIT3 = false>
__global__ __launch_bounds__(512) void gemm_tile8_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 k-tiles][A | B]
    const int nwg = gridDim.x;
    const int lg = xcd_logical_id(blockIdx.y * nwg + blockIdx.x, nwg * (int)gridDim.y);
    const int tiles_m = (g.M + 255) / 256, tiles_n = (g.N + 255) / 256;
    const int ksplit = lg / nwg, bid = lg - ksplit * nwg;
    const int nk_all = (SPLIT3 ? 3 : 1) * (g.K / BK);
    const int kt0 = ksplit * g.kt_per_split, kt1 = min(nk_all, kt0 + g.kt_per_split);
    constexpr int GROUP = OMLM_SUPER_ROWS / 256;
    const int gsz = GROUP * tiles_n;
    const int grp = bid / gsz, first_m = grp * GROUP;
    const int rows_in = min(GROUP, tiles_m - first_m);
    const int tm = first_m + (bid - grp * gsz) % rows_in, tn = (bid - grp * gsz) / rows_in;
    if (kt0 >= kt1) return;
    gemm_tile8_body<A_KMAJ, B_KMAJ, TOUT, SPLIT3>(g, tm * 256, tn * 256, kt0, kt1, gridDim.y > 1, smem);
}


// ---- grouped weight-gradient GEMM ------------------------------------------------------------------------------------
// All dW += dY^T X contractions of a backward pass (same K = tokens, small outputs) as ONE launch: the 30 separate GEMMs of a
// coarse-small step each had too few output tiles for 256 CUs and were split 5..31 ways along K with fp32 atomics
// (dW2: 44 tiles x 11 splits, 640 TFLOP/s); together they are ~900 full-K tiles, i.e. 3-4 machine rounds with few or no atomics.
// Nothing consumes a weight gradient before the optimizer, so the host defers them to the end of the backward.
#define OMLM_GROUP_MAX 48
struct omlm_gemm_wgrad_desc { const void* A; const void* B; float* C; const int* c_map; int M, N, K, lda, ldb, ldc; };   // include/omlm.h
struct GroupProb { const void* A; const void* B; float* C; const int* c_map; int M, N, K, lda, ldb, ldc, kt_per_split, start; };
struct GroupArgs { int n, total; GroupProb p[OMLM_GROUP_MAX]; };

// OMLM_GEMM_FASTK=0: the general DMA address form everywhere (A/B lever)
static bool gemm_fastk_off() {
    static int off = -1;
    if (off < 0) { const char* e = getenv("OMLM_GEMM_FASTK"); off = (e && e[0] == '0') ? 1 : 0; }
    return off == 1;
}

template <bool FASTK, bool T8 = false>
__global__ __launch_bounds__(512) void gemm_wgrad_group_kernel(GroupArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lg = xcd_logical_id(blockIdx.x, ga.total);
    int pi = 0;
    for (int i = 1; i < ga.n; ++i) if (lg >= ga.p[i].start) pi = i;      // uniform scalar scan (starts ascend)
    const GroupProb& q = ga.p[pi];
    GemmArgs g;
    g.A = q.A; g.B = q.B; g.C = q.C; g.Cin = q.C; g.a_map = nullptr; g.b_map = nullptr; g.c_map = q.c_map;
    g.a_rows = q.K; g.b_rows = q.K; g.M = q.M; g.N = q.N; g.K = q.K; g.lda = q.lda; g.ldb = q.ldb; g.ldc = q.ldc; g.ldcin = q.ldc;
    g.alpha = 1.f; g.kt_per_split = q.kt_per_split; g.bal_ck = 0; g.bal_chunks = 0; g.debug = 0; g.split3 = 0; g.A_lo = nullptr; g.B_lo = nullptr; g.C_lo = nullptr; g.c_lo8 = 0; g.c_split_stride = 0;
    g.epi_scale = nullptr; g.epi_norm = nullptr; g.epi_groups = 0; g.epi_ldnorm = 0; g.C2 = nullptr; g.c2_col0 = 0; g.ldc2 = 0;
    const int nk = (q.K + BK - 1) / BK;
    if constexpr (T8) {
        // the half-tile-ring schedule (round 5), same tile / split numbering as gemm_tile_body
        const int lq = lg - q.start;
        const int tiles_m = (q.M + 255) / 256, tiles_n = (q.N + 255) / 256, nwg = tiles_m * tiles_n;
        const int ksplit = lq / nwg, bid = lq - ksplit * nwg;
        const int kt0 = ksplit * q.kt_per_split, kt1 = min(nk, kt0 + q.kt_per_split);
        constexpr int GROUP = OMLM_SUPER_ROWS / 256;
        const int gsz = GROUP * tiles_n;
        const int grp = bid / gsz, first_m = grp * GROUP;
        const int rows_in = min(GROUP, tiles_m - first_m);
        const int tm = first_m + (bid - grp * gsz) % rows_in, tn = (bid - grp * gsz) / rows_in;
        if (kt0 < kt1) gemm_tile8_body<true, true, float>(g, tm * 256, tn * 256, kt0, kt1, q.kt_per_split < nk, smem);
    } else {
        gemm_tile_body<256, 256, 128, 64, true, true, float, false, false, false, false, FASTK>(g, lg - q.start, q.kt_per_split < nk, 0, smem);
    }
}


#ifdef OMLM_ISA_ONLY       /* tools/isa_audit.py-style inspection builds: the persistent kernels alone (seconds instead of minutes) */
template __global__ void gemm_bf16_tile_persist_kernel<256, 256, 128, 64, false, false, h16_t, false>(GemmArgs);
template __global__ void gemm_bf16_tile_persist_kernel<256, 256, 128, 64, false, true, h16_t, false>(GemmArgs);
template __global__ void gemm_bf16_tile_persist_kernel<256, 256, 128, 64, false, false, float, true>(GemmArgs);
template __global__ void gemm_bf16_tile_persist_kernel<256, 256, 128, 64, false, true, float, true>(GemmArgs);
template __global__ void gemm_bf16_tile_persist_kernel<128, 128, 64, 64, false, false, h16_t, false>(GemmArgs);
template __global__ void gemm_bf16_tile_persist_kernel<128, 128, 64, 64, false, false, float, true>(GemmArgs);
template __global__ void gemm_bf16_tile_persist_kernel<128, 128, 64, 64, false, false, h16_t, false, 1>(GemmArgs);
template __global__ void gemm_tile8_kernel<false, false, h16_t>(GemmArgs);
template __global__ void gemm_tile8_kernel<false, true, h16_t>(GemmArgs);
template __global__ void gemm_tile8_kernel<false, false, float>(GemmArgs);
template __global__ void gemm_tile8_kernel<true, true, float>(GemmArgs);
template __global__ void gemm_tile8_kernel<false, false, h16pl_t, true>(GemmArgs);      // the plane route of fp16ff: 3 products, planes out / fp32 out
template __global__ void gemm_tile8_kernel<false, false, float, true>(GemmArgs);
}   // namespace
#else
// workgroups of the persistent walk: one per CU, rounded down to a multiple of 8 (the walk's stride must keep a workgroup on its XCD);
// 0 = switched off (OMLM_GEMM_PERSIST=0)
static int gemm_persist_slots() {
    static int ncu8 = -1;
    if (ncu8 < 0) {
        int dev = 0, n = 0;
        ncu8 = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8) ? n / 8 * 8 : 0;
    }
    const char* e = getenv("OMLM_GEMM_PERSIST");                   // read per call like the other levers: tests toggle it inside one process
    return (e && e[0] == '0') ? 0 : ncu8;
}

// OMLM_GEMM_T8: the half-tile-ring schedule (gemm_tile8_body) for the 256 x 256 tiles: 0 = off, 1 = every eligible launch, 2 (default) =
// where it measured faster (profiles/r05b_gemm_t8_ab.md, same box, bit-identical results): the grouped weight gradients (557 k-tiles per
// tile: +6 %) and multi-round launches with K >= 2048 (d(xn2), K = 5504: +3 %; FF-out, K = 2752: +2 %; 8192^3: +15 %).  Short contractions
// (K = 1024: 16 k-tiles per tile) stay on the persistent walk of the rotated loop, which hides the per-tile prologue / epilogue that this
// one-tile-per-workgroup form exposes (FF-in 373 vs 399 us).  Read per call like the other levers (tests and tools/lib_ab toggle it).
static int gemm_t8_mode() {
    const char* e = getenv("OMLM_GEMM_T8");
    return e ? atoi(e) : OMLM_GEMM_T8_DEFAULT;
}
static bool gemm_t8_wanted(int K, int tiles, int splits) {
    const int mode = gemm_t8_mode();
    if (mode <= 0) return false;
    if (mode == 1) return true;
    static int ncu = 0;                                            // more than one round of one-workgroup-per-CU tiles (whatever OMLM_GEMM_PERSIST says)
    if (ncu == 0) {
        int dev = 0, n = 0;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return splits == 1 && K >= 2048 && tiles > ncu;
}

template <int BM_, int BN_, int WM_, int WN_, typename TOUT>
static int launch_tile(const GemmArgs& g, int a_kmaj, int b_kmaj, int splits, hipStream_t st) {
    constexpr int NTH = (BM_ / WM_) * (BN_ / WN_) * 64;
    constexpr size_t LDS = 2 * (size_t)(BM_ + BN_) * BK * 2;
    const int tiles = ((g.M + BM_ - 1) / BM_) * ((g.N + BN_ - 1) / BN_);
    dim3 grid(tiles, splits), block(NTH);
    if constexpr (BM_ == 256 && BN_ == 256) {
        // whole k-tiles, no maps on the operand side (a scatter map of C and split-K are fine), no ablation / plane / balanced modes
        // (the hi/lo-plane route runs a 3x k-loop: K >= 704 already is a long contraction for it; its instantiations exist in the bf16 copy only)
        if (gemm_t8_wanted(g.split3 ? 3 * g.K : g.K, tiles, splits) && g.K % BK == 0 && !g.a_map && !g.b_map && g.bal_ck == 0 && !g.debug && !gemm_fastk_off() &&
            (!g.split3 || !OMLM_FP16)) {
#define OMLM_T8_LAUNCH(AK, BKM)                                                                                       \
            do {                                                                                                       \
                auto k8 = gemm_tile8_kernel<AK, BKM, TOUT>;                                                            \
                static bool attr8 = false;                                                                             \
                if (!attr8) { (void)hipFuncSetAttribute((const void*)k8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); attr8 = true; } \
                if constexpr (!OMLM_FP16) {                                                                            \
                    auto k83 = gemm_tile8_kernel<AK, BKM, TOUT, true>;                                                 \
                    static bool attr83 = false;                                                                        \
                    if (!attr83) { (void)hipFuncSetAttribute((const void*)k83, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); attr83 = true; } \
                    if (g.split3) { hipLaunchKernelGGL(k83, grid, block, LDS, st, g); break; }                         \
                }                                                                                                      \
                hipLaunchKernelGGL(k8, grid, block, LDS, st, g);                                                       \
            } while (0)
            if (!a_kmaj && !b_kmaj)      OMLM_T8_LAUNCH(false, false);
            else if (!a_kmaj && b_kmaj)  OMLM_T8_LAUNCH(false, true);
            else if (a_kmaj && b_kmaj)   OMLM_T8_LAUNCH(true, true);
            else                         OMLM_T8_LAUNCH(true, false);
#undef OMLM_T8_LAUNCH
            return omlm_post_launch("omlm_gemm");
        }
    }
    if (g.bal_ck > 0) grid = dim3(splits, 1);          // balanced split-K: `splits` carries the workgroup count
    const bool need_kmap = (a_kmaj && g.a_map) || (b_kmaj && g.b_map);       // host routes these to the 128x128 tile
    if (need_kmap && BM_ != 128) { omlm_set_error("omlm_gemm: k-row maps are only built for the 128x128 tile"); return OMLM_ERR_UNSUPPORTED; }
    // The fp16 copy of this file (common.h: OMLM_FP16) instantiates only the production kernel and its k-row-map form: the ablation
    // (DBG), balanced split-K (BAL) and hi/lo-plane (SPLIT3) instantiations exist once, in the bf16 copy.
#define OMLM_TILE_LAUNCH(AK, BKM)                                                                                          \
    do {                                                                                                                    \
        auto kfn = gemm_bf16_tile_kernel<BM_, BN_, WM_, WN_, AK, BKM, TOUT, false, false>;                                 \
        auto kmap = gemm_bf16_tile_kernel<BM_, BN_, WM_, WN_, AK, BKM, TOUT, false, (AK || BKM) && BM_ == 128>;            \
        auto kfast = gemm_bf16_tile_kernel<BM_, BN_, WM_, WN_, AK, BKM, TOUT, false, false, false, false, true>;           \
        static bool attr = false;                                                                                           \
        if (!attr) {                                                                                                        \
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);             \
            (void)hipFuncSetAttribute((const void*)kfast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);           \
            (void)hipFuncSetAttribute((const void*)kmap, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);            \
        }                                                                                                                   \
        if constexpr (!OMLM_FP16) {                                                                                         \
            auto kdbg = gemm_bf16_tile_kernel<BM_, BN_, WM_, WN_, AK, BKM, TOUT, true, false>;                             \
            auto kbal = gemm_bf16_tile_kernel<BM_, BN_, WM_, WN_, AK, BKM, TOUT, false, false, sizeof(TOUT) == 4>;        \
            auto kbalmap = gemm_bf16_tile_kernel<BM_, BN_, WM_, WN_, AK, BKM, TOUT, false, (AK || BKM) && BM_ == 128, sizeof(TOUT) == 4>; \
            auto ks3 = gemm_bf16_tile_kernel<BM_, BN_, WM_, WN_, AK, BKM, TOUT, false, false, false, true>;                 \
            if (!attr) {                                                                                                    \
                (void)hipFuncSetAttribute((const void*)kdbg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);        \
                (void)hipFuncSetAttribute((const void*)kbal, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);        \
                (void)hipFuncSetAttribute((const void*)kbalmap, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);     \
                (void)hipFuncSetAttribute((const void*)ks3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);         \
            }                                                                                                               \
            attr = true;                                                                                                    \
            if (g.split3) { hipLaunchKernelGGL(ks3, grid, block, LDS, st, g); break; }                                     \
            if (g.bal_ck > 0 && sizeof(TOUT) == 4) { hipLaunchKernelGGL(need_kmap ? kbalmap : kbal, grid, block, LDS, st, g); break; } \
            if (g.debug && !need_kmap) { hipLaunchKernelGGL(kdbg, grid, block, LDS, st, g); break; }                       \
        }                                                                                                                   \
        attr = true;                                                                                                        \
        if (need_kmap) hipLaunchKernelGGL(kmap, grid, block, LDS, st, g);                                                  \
        else if (g.K % BK == 0 && !gemm_fastk_off()) hipLaunchKernelGGL(kfast, grid, block, LDS, st, g);                   \
        else           hipLaunchKernelGGL(kfn, grid, block, LDS, st, g);                                                   \
    } while (0)
    // Persistent walk (gemm_bf16_tile_persist_kernel) for the wide tile when the problem is more than one round of the machine:
    // whole k-tiles, k-contiguous A, no maps on the k side, no split, no ablation / plane modes.  OMLM_GEMM_PERSIST=0 keeps the one-tile grid.
    if constexpr ((BM_ == 256 && BN_ == 256) || (BM_ == 128 && BN_ == 128)) {
        const int slots = gemm_persist_slots() * (BM_ == 128 ? 2 : 1);             // 64 KiB tiles: two workgroups per CU
        if (slots > 0 && !a_kmaj && splits == 1 && g.bal_ck == 0 && !g.split3 && !g.debug && !g.a_map && !g.b_map && !g.c_map && g.K % BK == 0 &&
            !gemm_fastk_off() && tiles > slots) {
            constexpr size_t LDSP = LDS + 32 * (WN_ + 4) * 4;
            static bool pattr = false;
            auto k0 = gemm_bf16_tile_persist_kernel<BM_, BN_, WM_, WN_, false, false, TOUT, false>;
            auto k1 = gemm_bf16_tile_persist_kernel<BM_, BN_, WM_, WN_, false, true, TOUT, false>;
            auto k0c = gemm_bf16_tile_persist_kernel<BM_, BN_, WM_, WN_, false, false, TOUT, true>;
            auto k1c = gemm_bf16_tile_persist_kernel<BM_, BN_, WM_, WN_, false, true, TOUT, true>;
            if (!pattr) {
                (void)hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSP);
                (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSP);
                (void)hipFuncSetAttribute((const void*)k0c, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSP);
                (void)hipFuncSetAttribute((const void*)k1c, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSP);
                pattr = true;
            }
            hipLaunchKernelGGL(g.Cin ? (b_kmaj ? k1c : k0c) : (b_kmaj ? k1 : k0), dim3(slots), block, LDSP, st, g);
            return omlm_post_launch("omlm_gemm");
        }
    }
    if (!a_kmaj && !b_kmaj)      OMLM_TILE_LAUNCH(false, false);
    else if (!a_kmaj && b_kmaj)  OMLM_TILE_LAUNCH(false, true);
    else if (a_kmaj && b_kmaj)   OMLM_TILE_LAUNCH(true, true);
    else                         OMLM_TILE_LAUNCH(true, false);
#undef OMLM_TILE_LAUNCH
    return omlm_post_launch("omlm_gemm");
}

// The hi/lo-plane route of omlm_gemm_planes16 (row-major A [M, K] and B [N, K], no maps, no split-K): the half-tile-ring kernel for the
// 256 x 256 tiles (whole k-tiles), the rotated-loop SPLIT3 kernel otherwise.  Both copies of the file build it (TOUT: float, or h16pl_t =
// the result leaves as planes too).
template <int BM_, int BN_, int WM_, int WN_, typename TOUT>
static int launch_tile_s3(const GemmArgs& g, hipStream_t st, int splits = 1) {
    constexpr int NTH = (BM_ / WM_) * (BN_ / WN_) * 64;
    constexpr size_t LDS = 2 * (size_t)(BM_ + BN_) * BK * 2;
    const int tiles = ((g.M + BM_ - 1) / BM_) * ((g.N + BN_ - 1) / BN_);
    dim3 grid(tiles, splits), block(NTH);
    if constexpr (BM_ == 256 && BN_ == 256) {
        if (gemm_t8_mode() > 0 && g.K % BK == 0 && !gemm_fastk_off() && !g.a_map && splits == 1) {
            auto k8 = gemm_tile8_kernel<false, false, TOUT, true>;
            static bool attr8 = false;
            if (!attr8) { (void)hipFuncSetAttribute((const void*)k8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); attr8 = true; }
            hipLaunchKernelGGL(k8, grid, block, LDS, st, g);
            return omlm_post_launch("omlm_gemm_planes16");
        }
    }
    auto ks3 = gemm_bf16_tile_kernel<BM_, BN_, WM_, WN_, false, false, TOUT, false, false, false, true>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)ks3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); attr = true; }
    hipLaunchKernelGGL(ks3, grid, block, LDS, st, g);
    return omlm_post_launch("omlm_gemm_planes16");
}

template <typename T, typename TOUT>
static int launch_layout(const GemmArgs& g, int a_kmaj, int b_kmaj, int splits, hipStream_t st) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    const size_t lds = 4 * (size_t)(BM * BK * 2);            // fp32: 4 planes (A/B x hi/lo); bf16: 2 buffers x (A | B)
    dim3 grid(tiles, splits), block(NTHREADS);
    if constexpr (elt_traits<T>::precise) {
        if (!a_kmaj && !b_kmaj)      hipLaunchKernelGGL((gemm_kernel<T, false, false, TOUT>), grid, block, lds, st, g);
        else if (!a_kmaj && b_kmaj)  hipLaunchKernelGGL((gemm_kernel<T, false, true, TOUT>), grid, block, lds, st, g);
        else if (a_kmaj && b_kmaj)   hipLaunchKernelGGL((gemm_kernel<T, true, true, TOUT>), grid, block, lds, st, g);
        else                         hipLaunchKernelGGL((gemm_kernel<T, true, false, TOUT>), grid, block, lds, st, g);
    }
    return omlm_post_launch("omlm_gemm");
}

// ---- the peeled tail as a deterministic split-K (round 5) -------------------------------------------------------------------------
// The m-tile rows behind the last full round of 256 x 256 tiles run on 128 x 128 tiles (gemm_impl below): 48 ... 190 workgroups whose
// k-loops (86 k-tiles for d(xn2), 129 loop tiles for the FF-out plane route) are one serial chain each on a fraction of the machine
// (80 / 121 us for ~35 / ~50 us of work at the chip's rate).  With a workspace (a per-call argument: gemm_impl's tail_ws) the tail's K range is cut
// into S slices that fill the 2-per-CU slots, every slice STORES its fp32 partial tile to its own plane of the workspace (no atomics, no
// pre-filled C), and gemm_tail_reduce_kernel adds the planes in a fixed order together with the residual and writes the output type
// (fp32, 16-bit, or 16-bit hi/lo planes): deterministic, two launches.  The workspace belongs to the
// caller and to ONE stream at a time (the library keeps no pointer: two streams pass two buffers); OMLM_GEMM_TAIL_SPLIT=0 or no workspace keeps the one-launch tail.

static bool gemm_tail_split_on() {
    const char* e = getenv("OMLM_GEMM_TAIL_SPLIT");                // read per call like the other levers
    return !(e && e[0] == '0');
}

// dtype codes shared with the Python host: 0 = fp32, 1 = bf16
static int gemm_impl(const void* A, const void* B, void* C, const float* Cin,
                     const int* a_map, const int* b_map, const int* c_map,
                     long long a_rows, long long b_rows,
                     int M, int N, int K, int lda, int ldb, int ldc, int ldcin,
                     int a_kmajor, int b_kmajor, int in_dtype, int out_dtype, float alpha, void* stream,
                     int split3, const void* A_lo, const void* B_lo, void* C_lo = nullptr, bool s3_route = false,
                     float* tail_ws = nullptr, long long tail_ws_bytes = 0) {
    if (M <= 0 || N <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(A && B && C, "null operand");
    OMLM_CHECK_ARG(K > 0, "K must be positive");
    OMLM_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0, "operand leading dimensions must be multiples of 8 elements");
    OMLM_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "operands must be 16-byte aligned");
    OMLM_CHECK_ARG(in_dtype == 0 || in_dtype == 1, "in_dtype: 0=fp32 (bf16x3) or 1=bf16");
    OMLM_CHECK_ARG(out_dtype == 0 || out_dtype == 1, "out_dtype: 0=fp32 or 1=bf16");
    const size_t esz = in_dtype == 0 ? 4 : 2;
    OMLM_CHECK_ARG((unsigned long long)a_rows * lda * esz < 0xFFFFFFF0ull, "A exceeds the 4 GiB buffer-descriptor window");
    OMLM_CHECK_ARG((unsigned long long)b_rows * ldb * esz < 0xFFFFFFF0ull, "B exceeds the 4 GiB buffer-descriptor window");
    // contiguous dims are consumed in chunks of 8: a ragged contiguous extent must be zero-padded to 8 by the caller
    if (a_kmajor) OMLM_CHECK_ARG(lda >= ((M + 7) / 8) * 8, "k-major A: row pitch shorter than M padded to 8");
    if (b_kmajor) OMLM_CHECK_ARG(ldb >= ((N + 7) / 8) * 8, "k-major B: row pitch shorter than N padded to 8");
    if (!a_kmajor || !b_kmajor) OMLM_CHECK_ARG(K % 8 == 0, "k-contiguous operands need K % 8 == 0 (zero-pad the contraction)");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.Cin = Cin; g.a_map = a_map; g.b_map = b_map; g.c_map = c_map;
    g.a_rows = a_rows; g.b_rows = b_rows; g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldcin = ldcin; g.alpha = alpha;
    { const char* dbg = getenv("OMLM_GEMM_DEBUG"); g.debug = dbg ? atoi(dbg) : 0; }
    g.split3 = split3; g.A_lo = A_lo; g.B_lo = B_lo; g.C_lo = C_lo; g.c_lo8 = 0; g.c_split_stride = 0;
    g.epi_scale = nullptr; g.epi_norm = nullptr; g.epi_groups = 0; g.epi_ldnorm = 0; g.C2 = nullptr; g.c2_col0 = 0; g.ldc2 = 0;
    hipStream_t st = as_stream(stream);
    // tile shape (bf16 path): 256x256 when both output dims are wide, 256x128 for tall-narrow outputs, else 128x128
    int bm = BM, bn = BN;
    const char* force = getenv("OMLM_GEMM_TILE");
    const bool need_kmap = (a_kmajor && a_map) || (b_kmajor && b_map);
    if (in_dtype == 1 && !need_kmap) {
        if (force && force[0]) { if (!strcmp(force, "256x256")) { bm = 256; bn = 256; } else if (!strcmp(force, "256x128")) { bm = 256; bn = 128; } }
        // Short contractions onto narrow outputs (to_out, d(xn), d(x) of k | v: K <= 512, N <= 1024): with the persistent walk the 128x128
        // tiles (two walkers per CU, 4 x the tiles to balance) beat the wide ones -- to_out 103 -> 92 us, d(xn) 57 -> 50 us (round 4 probe).
        // OMLM_GEMM_SMALLK=0 keeps the old choice.
        else if (K <= 512 && N <= 1024 && K % BK == 0 && !a_kmajor && !a_map && !b_map && !c_map && !split3 && Cin != (const float*)C &&
                 ((M + 127) / 128) * ((N + 127) / 128) > 2 * gemm_persist_slots() && gemm_persist_slots() > 0 && !gemm_fastk_off() &&
                 !(getenv("OMLM_GEMM_SMALLK") && getenv("OMLM_GEMM_SMALLK")[0] == '0')) { bm = 128; bn = 128; }
        else if (M >= 1024 && N >= 1024) { bm = 256; bn = 256; }   // measured (probe, N = 1024): 256x256 514 us, 128x128 543, 256x128 657
        // N = 512 outputs (q-proj, d(o)): 128x128 (two workgroups per CU) measured 54 / 54 us against 60 / 59 for 256x128 (round 4 tile probe)
        else if (M >= 2048 && N > 512) { bm = 256; bn = 128; }
        else if (N >= 2048 && M >= 256) { bm = 256; bn = 256; }
    }
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n;
        else ncu = 256;
    }
    // split-K only for accumulate-into-C GEMMs with few output tiles (the weight-gradient contractions).  The 256-wide
    // tiles run one workgroup per CU, so the split count is chosen for whole rounds of the machine: e.g. dW1 has 88 tiles;
    // 12 splits = 1056 workgroups = 4.1 rounds (82 % of the last 5 used), 11 splits = 968 = 3.8 rounds (95 %).
    const int nk = ((K + BK - 1) / BK) * (split3 ? 3 : 1);     // k-tiles of the loop (three plane pairs per real k-tile when split3)
    const int tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    int splits = 1;
    // fp32 operands (register-staged kernel: the rel-pos MLP's 0.3-GFLOP GEMMs, 36 / 16 output tiles) are latency-bound per k-tile, not per byte:
    // they split down to TWO k-tiles per workgroup (54 -> ~20 us per launch; round 4), the 16-bit tile kernels keep their >= 8 k-tiles per split
    const bool fine = in_dtype == 0 && !split3;
    if (Cin == (const float*)C && out_dtype == 0 && tiles < 512 && nk >= (fine ? 4 : 16)) {
        const int slots = (bm == 256 ? 1 : 2) * ncu;              // co-resident workgroups (LDS: 128 KiB tiles 1 / CU, 64 KiB 2 / CU)
        int smax = fine ? nk / 2 : nk / 8; if (smax > 32) smax = 32; if (smax < 1) smax = 1;
        int smin = (slots + tiles - 1) / tiles; if (smin > smax) smin = smax; if (smin < 1) smin = 1;      // at least one full round
        float best = -1.f;
        for (int sp = smin; sp <= smax; ++sp) {
            const int total = tiles * sp, rounds = (total + slots - 1) / slots;
            // every split adds one atomic pass over C.  Re-measured after the k-major DMA fix (tools/splitk_probe.py): with the k-loop
            // faster the atomics weigh more -- 44 tiles (dW2): 5 / 11 splits = 233 / 267 us; 88 tiles (dW1): 5 / 8 / 11 = 479 / 478 / 515 us;
            // 128x128 tiles (dWq, dWkv: 64 KiB partials) keep the old weight: 16 / 32 splits stay best there.
            const float util = (float)total / (float)(rounds * slots) - (bm == 256 ? 0.02f : 0.012f) * (float)sp;
            if (util > best) { best = util; splits = sp; }
        }
    }
    { const char* e = getenv("OMLM_GEMM_SPLITS"); if (e && atoi(e) > 0 && splits > 1) splits = atoi(e); }     // tuning override
    g.kt_per_split = (nk + splits - 1) / splits;
    splits = (nk + g.kt_per_split - 1) / g.kt_per_split;
    // Balanced split-K for the same GEMMs (bf16 tile kernels): one workgroup per slot, every workgroup the same number of k-tiles,
    // K cut into round(slots / tiles) chunks so that co-resident workgroups read the same K range.
    g.bal_ck = 0; g.bal_chunks = 0;
    if (splits > 1 && in_dtype == 1 && !split3) {
        static int bal_on = -1;
        // measured (MI355X, dW1: 88 tiles x 558 k-tiles): 665 us balanced vs 642 us with the 8-split grid, train step 36.9 vs 36.7 ms --
        // the k-major main loop, not the partial last round or the atomic volume, is what holds these GEMMs at ~620 TFLOP/s.
        // Off by default; OMLM_GEMM_BAL=1 selects it.
        if (bal_on < 0) { const char* e = getenv("OMLM_GEMM_BAL"); bal_on = (e && e[0] == '1') ? 1 : 0; }
        if (bal_on && !OMLM_FP16) {
            const int slots = (bm == 256 ? 1 : 2) * ncu;
            const long long U = (long long)tiles * nk;
            long long G = slots;
            if (U / 24 < G) G = U / 24 > 0 ? U / 24 : 1;            // at least ~24 k-tiles per workgroup: prologue + atomic epilogue amortised
            int chunks = (int)((G + tiles / 2) / tiles); if (chunks < 1) chunks = 1; if (chunks > nk) chunks = nk;
            g.bal_ck = (nk + chunks - 1) / chunks;
            g.bal_chunks = (nk + g.bal_ck - 1) / g.bal_ck;
            splits = (int)G;
        }
    }
    if (in_dtype == 0) {
#if OMLM_FP16
        omlm_set_error("omlm_gemm: fp32 operands are served by the bf16 copy of the library");
        return OMLM_ERR_UNSUPPORTED;
#else
        static bool attr_done = false;   // 64 KiB dynamic LDS needs the opt-in attribute once per kernel
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)gemm_kernel<float, false, false, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            (void)hipFuncSetAttribute((const void*)gemm_kernel<float, false, true, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            (void)hipFuncSetAttribute((const void*)gemm_kernel<float, true, true, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            (void)hipFuncSetAttribute((const void*)gemm_kernel<float, true, false, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            attr_done = true;
        }
        OMLM_CHECK_ARG(out_dtype == 0, "fp32 operands produce fp32 output");
        return launch_layout<float, float>(g, a_kmajor, b_kmajor, splits, st);
#endif
    }
    auto launch = [&](const GemmArgs& ga, int tm_, int tn_, int sp) -> int {
        if (s3_route) {                  // omlm_gemm_planes16: splits == 1 (no accumulate-into-C form), out = fp32 or 16-bit planes
            if (tm_ == 256 && tn_ == 256)
                return ga.C_lo ? launch_tile_s3<256, 256, 128, 64, h16pl_t>(ga, st) : launch_tile_s3<256, 256, 128, 64, float>(ga, st);
            if (tm_ == 256 && tn_ == 128)
                return ga.C_lo ? launch_tile_s3<256, 128, 64, 64, h16pl_t>(ga, st) : launch_tile_s3<256, 128, 64, 64, float>(ga, st);
            return ga.C_lo ? launch_tile_s3<128, 128, 64, 64, h16pl_t>(ga, st) : launch_tile_s3<128, 128, 64, 64, float>(ga, st);
        }
        if (tm_ == 256 && tn_ == 256)
            return out_dtype == 0 ? launch_tile<256, 256, 128, 64, float>(ga, a_kmajor, b_kmajor, sp, st)
                                  : launch_tile<256, 256, 128, 64, h16_t>(ga, a_kmajor, b_kmajor, sp, st);
        if (tm_ == 256 && tn_ == 128)
            return out_dtype == 0 ? launch_tile<256, 128, 64, 64, float>(ga, a_kmajor, b_kmajor, sp, st)
                                  : launch_tile<256, 128, 64, 64, h16_t>(ga, a_kmajor, b_kmajor, sp, st);
        return out_dtype == 0 ? launch_tile<128, 128, 64, 64, float>(ga, a_kmajor, b_kmajor, sp, st)
                              : launch_tile<128, 128, 64, 64, h16_t>(ga, a_kmajor, b_kmajor, sp, st);
    };
    // Tail peeling for the one-workgroup-per-CU 256x256 tiles: dX-type GEMMs have 560 tiles = 2.19 rounds of 256 CUs, i.e. a
    // third round that is 19 % full.  The m-tile rows that fill whole rounds keep the 256x256 kernel; the remaining rows go to
    // the 128x128 kernel (2 workgroups per CU, ~1/3 of the time per tile).
    if (bm == 256 && bn == 256 && splits == 1 && !a_kmajor && !a_map && !c_map && !(force && force[0]) && tiles > ncu) {
        const int tiles_n = (N + 255) / 256, tiles_m = (M + 255) / 256;
        const int rem = tiles % ncu;
        const int m_full = ((tiles / ncu) * ncu) / tiles_n;
        if (rem > 0 && rem < ncu / 2 && m_full >= 1 && m_full < tiles_m) {
            const size_t osz = out_dtype == 0 ? 4 : 2;
            const long long M1 = (long long)m_full * 256;
            GemmArgs g1 = g, g2 = g;
            g1.M = (int)M1;
            g2.M = M - (int)M1;
            g2.A = (const char*)A + (size_t)M1 * lda * 2;
            g2.a_rows = a_rows - M1;
            g2.C = (char*)C + (size_t)M1 * ldc * osz;
            if (g.A_lo && !a_kmajor) g2.A_lo = (const char*)g.A_lo + (size_t)M1 * lda * 2;
            if (g.C_lo) g2.C_lo = (char*)g.C_lo + (size_t)M1 * ldc * osz;
            if (Cin) g2.Cin = Cin + (size_t)M1 * ldcin;
            const int rc = launch(g1, 256, 256, 1);
            if (rc != OMLM_OK) return rc;
            // the tail: deterministic split-K through the workspace when it pays (see gemm_tail_reduce_kernel)
            if (tail_ws && gemm_tail_split_on() && alpha == 1.0f) {
                const int Mt = g2.M, Nw = (N + 3) / 4 * 4;
                const int tiles_t = ((Mt + 127) / 128) * ((N + 127) / 128);
                int S = (2 * ncu) / tiles_t;
                if (S > nk / 8) S = nk / 8;
                if (S > 8) S = 8;
                if (S >= 2) {
                    const int ktps = (nk + S - 1) / S;
                    S = (nk + ktps - 1) / ktps;
                    const long long slice = (long long)Mt * Nw;
                    if (S >= 2 && (long long)S * slice * 4 <= tail_ws_bytes) {
                        GemmArgs gw = g2;
                        gw.C = tail_ws; gw.C_lo = nullptr; gw.Cin = nullptr; gw.ldc = Nw; gw.ldcin = 0;
                        gw.c_split_stride = slice; gw.kt_per_split = ktps;
                        const int rc2 = s3_route ? launch_tile_s3<128, 128, 64, 64, float>(gw, st, S)
                                                 : launch_tile<128, 128, 64, 64, float>(gw, a_kmajor, b_kmajor, S, st);
                        if (rc2 != OMLM_OK) return rc2;
                        const long long quads = (long long)Mt * (Nw / 4);
                        const int blocks = (int)((quads + 255) / 256 > 4096 ? 4096 : (quads + 255) / 256);
                        if (out_dtype == 0)
                            hipLaunchKernelGGL(gemm_tail_reduce_kernel<0>, dim3(blocks), dim3(256), 0, st, tail_ws, S, slice, Mt, N, Nw, g2.C, nullptr, ldc, g2.Cin, ldcin);
                        else if (g2.C_lo)
                            hipLaunchKernelGGL(gemm_tail_reduce_kernel<2>, dim3(blocks), dim3(256), 0, st, tail_ws, S, slice, Mt, N, Nw, g2.C, g2.C_lo, ldc, g2.Cin, ldcin);
                        else
                            hipLaunchKernelGGL(gemm_tail_reduce_kernel<1>, dim3(blocks), dim3(256), 0, st, tail_ws, S, slice, Mt, N, Nw, g2.C, nullptr, ldc, g2.Cin, ldcin);
                        return omlm_post_launch("omlm_gemm (tail reduce)");
                    }
                }
            }
            return launch(g2, 128, 128, 1);
        }
    }
    return launch(g, bm, bn, splits);
}

// Workspace of the peeled tail's deterministic split-K: a caller-owned scratch buffer handed to every call that may peel a tail (omlm_gemm,
// omlm_gemm_planes16, omlm_gemm_planes) -- ONE per stream that launches such GEMMs concurrently; NULL / 0: the one-launch tail.  The library keeps no
// pointer.  omlm_gemm_tail_workspace_bytes: an upper bound of what an M x N output needs (8 fp32 slices of the at most one-machine-round tail).
#if !OMLM_FP16
extern "C" long long omlm_gemm_tail_workspace_bytes(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const long long tiles_n = (N + 255) / 256, tiles_m = (M + 255) / 256;
    if (tiles_m * tiles_n <= ncu) return 0;
    long long tail_rows = ((long long)(ncu / 2) / tiles_n + 1) * 256;        // fewer than half a round of tiles ever go to the tail
    if (tail_rows > M) tail_rows = M;
    return 8ll * tail_rows * ((N + 3) / 4 * 4) * 4;
}
#endif

// in_dtype / out_dtype: 0 = fp32, 1 = bf16, 2 = fp16 (include/omlm.h).  fp16 operands (with fp32 or fp16 output) are served by the
// fp16 copy of this file; bf16 and fp32 operands here.
#if !OMLM_FP16
extern "C" int omlm_gemm_h(const void* A, const void* B, void* C, const float* Cin, const int* a_map, const int* b_map, const int* c_map,
                           long long a_rows, long long b_rows, int M, int N, int K, int lda, int ldb, int ldc, int ldcin,
                           int a_kmajor, int b_kmajor, int in_dtype, int out_dtype, float alpha, void* workspace, long long workspace_bytes, void* stream);
#endif
extern "C" int OMLM_API(omlm_gemm)(const void* A, const void* B, void* C, const float* Cin,
                         const int* a_map, const int* b_map, const int* c_map,
                         long long a_rows, long long b_rows,
                         int M, int N, int K, int lda, int ldb, int ldc, int ldcin,
                         int a_kmajor, int b_kmajor, int in_dtype, int out_dtype, float alpha, void* workspace, long long workspace_bytes, void* stream) {
#if !OMLM_FP16
    if (in_dtype == OMLM_DT_F16) {
        OMLM_CHECK_ARG(out_dtype == OMLM_DT_F32 || out_dtype == OMLM_DT_F16, "fp16 operands produce fp32 or fp16 output");
        return omlm_gemm_h(A, B, C, Cin, a_map, b_map, c_map, a_rows, b_rows, M, N, K, lda, ldb, ldc, ldcin, a_kmajor, b_kmajor,
                           1, OMLM_H_CODE(out_dtype), alpha, workspace, workspace_bytes, stream);
    }
#endif
    OMLM_CHECK_ARG((workspace == nullptr) == (workspace_bytes == 0) && workspace_bytes >= 0 && ((uintptr_t)workspace % 16) == 0,
                   "tail workspace: 16-byte aligned buffer and its size, or NULL / 0");
    return gemm_impl(A, B, C, Cin, a_map, b_map, c_map, a_rows, b_rows, M, N, K, lda, ldb, ldc, ldcin, a_kmajor, b_kmajor,
                     in_dtype, out_dtype, alpha, stream, 0, nullptr, nullptr, nullptr, false, (float*)workspace, workspace_bytes);
}

// q / k projections with the attention's l2-norm + learned scale folded into the epilogue (transformer.py:254-271): C = 16-bit
// [M, N] (+ C2 for columns >= c2_col0, e.g. v of the fused k | v projection), epi_groups leading 64-column heads normalised, their norms
// to norm_out [M, ldnorm] fp32 (what the backward needs instead of the fp32 pre-norm projections).  A [M, K], B [N, K] row-major 16-bit.
#if !OMLM_FP16
extern "C" int omlm_gemm_qknorm_h(const void* A, const void* B, void* C, void* C2, int c2_col0, int ldc2, const float* scale, float* norm_out,
                                  int ldnorm, int groups, long long a_rows, long long b_rows, int M, int N, int K, int lda, int ldb, int ldc,
                                  int dtype, void* stream);
#endif
extern "C" int OMLM_API(omlm_gemm_qknorm)(const void* A, const void* B, void* C, void* C2, int c2_col0, int ldc2, const float* scale,
                                          float* norm_out, int ldnorm, int groups, long long a_rows, long long b_rows, int M, int N, int K,
                                          int lda, int ldb, int ldc, int dtype, void* stream) {
#if !OMLM_FP16
    if (dtype == OMLM_DT_F16) return omlm_gemm_qknorm_h(A, B, C, C2, c2_col0, ldc2, scale, norm_out, ldnorm, groups, a_rows, b_rows, M, N, K, lda, ldb, ldc, 1, stream);
#endif
    if (M <= 0 || N <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(dtype == 1, "gemm_qknorm: operand dtype 1 (bf16) or 2 (fp16)");
    OMLM_CHECK_ARG(A && B && C && scale && norm_out && K > 0 && (K % 8) == 0, "gemm_qknorm: null operand / K");
    OMLM_CHECK_ARG((N % 64) == 0 && groups >= 0 && groups * 64 <= N && ldnorm >= groups, "gemm_qknorm: N must be whole 64-wide heads");
    OMLM_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && (ldc % 8) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0,
                   "gemm_qknorm: 16-byte aligned operands, pitches multiples of 8");
    OMLM_CHECK_ARG(!C2 || ((c2_col0 % 64) == 0 && (ldc2 % 8) == 0 && ((uintptr_t)C2 % 16) == 0), "gemm_qknorm: second output");
    OMLM_CHECK_ARG((unsigned long long)a_rows * lda * 2 < 0xFFFFFFF0ull && (unsigned long long)b_rows * ldb * 2 < 0xFFFFFFF0ull, "operand exceeds the 4 GiB window");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.a_rows = a_rows; g.b_rows = b_rows; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = 1.f; g.epi_scale = scale; g.epi_norm = norm_out; g.epi_groups = groups; g.epi_ldnorm = ldnorm; g.C2 = C2; g.c2_col0 = c2_col0; g.ldc2 = ldc2;
    const int nk = (K + BK - 1) / BK;
    g.kt_per_split = nk;
    constexpr size_t LDS = 2 * (size_t)(128 + 128) * BK * 2;
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    auto kfast = gemm_bf16_tile_kernel<128, 128, 64, 64, false, false, h16_t, false, false, false, false, true, 1>;
    auto kgen = gemm_bf16_tile_kernel<128, 128, 64, 64, false, false, h16_t, false, false, false, false, false, 1>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)kfast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        (void)hipFuncSetAttribute((const void*)kgen, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        attr = true;
    }
    const int slots = 2 * gemm_persist_slots();                    // persistent walk, two workgroups per CU (see gemm_bf16_tile_persist_kernel)
    if (slots > 0 && tiles > slots && K % BK == 0 && !gemm_fastk_off()) {
        constexpr size_t LDSP = LDS + 32 * (64 + 4) * 4;
        auto kp = gemm_bf16_tile_persist_kernel<128, 128, 64, 64, false, false, h16_t, false, 1>;
        static bool pattr = false;
        if (!pattr) { (void)hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSP); pattr = true; }
        hipLaunchKernelGGL(kp, dim3(slots), dim3(256), LDSP, as_stream(stream), g);
    }
    else if (K % BK == 0 && !gemm_fastk_off()) hipLaunchKernelGGL(kfast, dim3(tiles, 1), dim3(256), LDS, as_stream(stream), g);
    else                                        hipLaunchKernelGGL(kgen, dim3(tiles, 1), dim3(256), LDS, as_stream(stream), g);
    return omlm_post_launch("omlm_gemm_qknorm");
}

// C = A B^T (+ Cin) with BOTH operands as hi/lo planes of the 16-bit type `dtype` (1 = bf16, 2 = fp16): A [M, K] and B [N, K] row-major with
// their lo planes A_lo / B_lo in the same layout (separate allocations are fine), three products hi*hi + hi*lo + lo*hi accumulated in fp32
// in ONE launch (3 x the k-tiles).  The result is fp32 (C_lo NULL; Cin optional), or -- C_lo given -- leaves as planes of the same type:
// C = rne16(v), C_lo = rne16(v - C), so that the next consumer reads the un-rounded value and a 16-bit backward reads C alone.
// Precision "fp16ff" (round 5): the two ConvFeedForward GEMMs of the forward, which carry 86-88 % of the fp16 logits-error variance
// (profiles/r05_error_budget.md).
#if !OMLM_FP16
extern "C" int omlm_gemm_planes16_h(const void* A, const void* A_lo, const void* B, const void* B_lo, void* C, void* C_lo, const float* Cin,
                                    const int* a_map, const int* c_map, long long a_rows, long long b_rows, int M, int N, int K,
                                    int lda, int ldb, int ldc, int ldcin, int dtype, void* workspace, long long workspace_bytes, void* stream);
#endif
extern "C" int OMLM_API(omlm_gemm_planes16)(const void* A, const void* A_lo, const void* B, const void* B_lo, void* C, void* C_lo, const float* Cin,
                                            const int* a_map, const int* c_map, long long a_rows, long long b_rows, int M, int N, int K,
                                            int lda, int ldb, int ldc, int ldcin, int dtype, void* workspace, long long workspace_bytes, void* stream) {
#if !OMLM_FP16
    if (dtype == OMLM_DT_F16) return omlm_gemm_planes16_h(A, A_lo, B, B_lo, C, C_lo, Cin, a_map, c_map, a_rows, b_rows, M, N, K, lda, ldb, ldc, ldcin, 1,
                                                          workspace, workspace_bytes, stream);
#endif
    OMLM_CHECK_ARG((workspace == nullptr) == (workspace_bytes == 0) && workspace_bytes >= 0 && ((uintptr_t)workspace % 16) == 0,
                   "tail workspace: 16-byte aligned buffer and its size, or NULL / 0");
    OMLM_CHECK_ARG(dtype == 1, "gemm_planes16: operand dtype 1 (bf16) or 2 (fp16)");
    OMLM_CHECK_ARG(A_lo && B_lo, "gemm_planes16: null lo plane");
    OMLM_CHECK_ARG(((uintptr_t)A_lo % 16) == 0 && ((uintptr_t)B_lo % 16) == 0 && ((uintptr_t)C_lo % 16) == 0, "gemm_planes16: 16-byte aligned planes");
    OMLM_CHECK_ARG(!(C_lo && Cin), "gemm_planes16: plane output takes no residual");
    // a_map / c_map (optional): physical row of logical row m in A (both planes) / in C and Cin, as in omlm_gemm -- the logit heads
    return gemm_impl(A, B, C, Cin, a_map, nullptr, c_map, a_rows, b_rows, M, N, K, lda, ldb, ldc, ldcin, 0, 0,
                     1, C_lo ? 1 : 0, 1.0f, stream, 1, A_lo, B_lo, C_lo, true, (float*)workspace, workspace_bytes);
}

#if !OMLM_FP16
// fp32-grade GEMM on bf16 hi/lo planes ("bf16x3" through the LDS-DMA tile kernels).  A and B are bf16 hi planes in the layout
// omlm_gemm takes for bf16 operands; the matching lo plane lies a_plane_bytes / b_plane_bytes behind each (omlm_split_planes
// writes such a pair).  One launch whose k-loop is three times as long: hi*hi + hi*lo + lo*hi, accumulated in fp32 -- the same
// three products per element as the register-staged fp32 path (gemm_kernel<float>), at the bf16 kernels' rate.
extern "C" int omlm_gemm_planes(const void* A, long long a_plane_bytes, const void* B, long long b_plane_bytes, void* C, const float* Cin,
                                const int* a_map, const int* b_map, const int* c_map, long long a_rows, long long b_rows,
                                int M, int N, int K, int lda, int ldb, int ldc, int ldcin,
                                int a_kmajor, int b_kmajor, int out_dtype, float alpha, void* workspace, long long workspace_bytes, void* stream) {
    OMLM_CHECK_ARG((workspace == nullptr) == (workspace_bytes == 0) && workspace_bytes >= 0 && ((uintptr_t)workspace % 16) == 0,
                   "tail workspace: 16-byte aligned buffer and its size, or NULL / 0");
    OMLM_CHECK_ARG(a_plane_bytes > 0 && b_plane_bytes > 0 && (a_plane_bytes % 16) == 0 && (b_plane_bytes % 16) == 0, "plane strides");
    OMLM_CHECK_ARG(!(a_kmajor && a_map) && !(b_kmajor && b_map), "k-row maps are not supported on operand planes");
    // each plane has its own buffer descriptor (round 5): the lo planes are addressed by pointer, not by an offset inside A's / B's window
    return gemm_impl(A, B, C, Cin, a_map, b_map, c_map, a_rows, b_rows, M, N, K, lda, ldb, ldc, ldcin, a_kmajor, b_kmajor,
                     1, out_dtype, alpha, stream, 1, (const char*)A + a_plane_bytes, (const char*)B + b_plane_bytes, nullptr, false,
                     (float*)workspace, workspace_bytes);
}

// x [n] fp32 -> planes: hi[i] = x truncated to bf16 at planes[i], lo[i] = RNE(x - hi) at planes[plane_elems + i]  (x ~= hi + lo to 2^-17)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, unsigned short* __restrict__ planes,
                                                           long long n, long long plane_elems) {
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    const long long stride = (long long)gridDim.x * 256 * 8;
    for (long long i = i0; i < n; i += stride) {
        if (i + 8 <= n) {
            const float4 a = *(const float4*)(x + i), b = *(const float4*)(x + i + 4);
            unsigned h0, h1, h2, h3, l0, l1, l2, l3;
            split_pair(a.x, a.y, h0, l0); split_pair(a.z, a.w, h1, l1);
            split_pair(b.x, b.y, h2, l2); split_pair(b.z, b.w, h3, l3);
            u32x4 hi, lo;
            hi[0] = h0; hi[1] = h1; hi[2] = h2; hi[3] = h3; lo[0] = l0; lo[1] = l1; lo[2] = l2; lo[3] = l3;
            *(u32x4*)(planes + i) = hi;
            *(u32x4*)(planes + plane_elems + i) = lo;
        } else {
            for (long long j = i; j < n; ++j) {
                const unsigned u = f2u(x[j]) & 0xFFFF0000u;
                planes[j] = (unsigned short)(u >> 16);
                planes[plane_elems + j] = (unsigned short)bf16_bits_rne(x[j] - u2f(u));
            }
        }
    }
}
extern "C" int omlm_split_planes(const float* x, void* planes, long long n, long long plane_elems, void* stream) {
    if (n <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(x && planes && plane_elems >= n && (plane_elems % 8) == 0, "split_planes arguments");
    OMLM_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)planes % 16) == 0, "split_planes: 16-byte aligned buffers");
    long long blocks = (n / 8 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, (unsigned short*)planes, n, plane_elems);
    return omlm_post_launch("omlm_split_planes");
}

#endif   // !OMLM_FP16 (operand planes: bf16 copy only)

// C_i[M_i, N_i] += A_i^T B_i for `count` problems with 16-bit k-major operands A_i [K_i, M_i], B_i [K_i, N_i] in ONE launch of 256x256
// full-K (or lightly split) tiles.  splits: K-splits per tile for every problem (0 = chosen here for whole machine rounds);
// dtype: operand type of ALL problems (1 = bf16, 2 = fp16).
#if !OMLM_FP16
extern "C" int omlm_gemm_wgrad_group_h(const omlm_gemm_wgrad_desc* d, int count, int splits, int dtype, void* stream);
#endif
extern "C" int OMLM_API(omlm_gemm_wgrad_group)(const omlm_gemm_wgrad_desc* d, int count, int splits, int dtype, void* stream) {
    if (count <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(d, "null descriptor array");
#if !OMLM_FP16
    if (dtype == OMLM_DT_F16) return omlm_gemm_wgrad_group_h(d, count, splits, 1, stream);
#endif
    OMLM_CHECK_ARG(dtype == 1, "wgrad group: operand dtype must be 1 (bf16) or 2 (fp16)");
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n;
        else ncu = 256;
    }
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_wgrad_group_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)gemm_wgrad_group_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)gemm_wgrad_group_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        attr = true;
    }
    { const char* e = getenv("OMLM_GROUP_SPLITS"); if (e && atoi(e) > 0) splits = atoi(e); }
    for (int base = 0; base < count; base += OMLM_GROUP_MAX) {
        const int n = count - base < OMLM_GROUP_MAX ? count - base : OMLM_GROUP_MAX;
        long long units = 0;
        int nk_min = 1 << 30;
        bool fastk = !gemm_fastk_off();                       // every problem's K a multiple of the k-tile depth: SGPR-offset DMA form
        for (int i = 0; i < n; ++i) {
            const omlm_gemm_wgrad_desc& q = d[base + i];
            if (q.K % BK != 0) fastk = false;
            OMLM_CHECK_ARG(q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.K > 0, "wgrad group: bad problem");
            OMLM_CHECK_ARG((q.lda % 8) == 0 && (q.ldb % 8) == 0 && q.lda >= ((q.M + 7) / 8) * 8 && q.ldb >= ((q.N + 7) / 8) * 8,
                           "wgrad group: k-major operand pitches must be multiples of 8 covering the padded extent");
            OMLM_CHECK_ARG(((uintptr_t)q.A % 16) == 0 && ((uintptr_t)q.B % 16) == 0, "wgrad group: operands must be 16-byte aligned");
            OMLM_CHECK_ARG((unsigned long long)q.K * q.lda * 2 < 0xFFFFFFF0ull && (unsigned long long)q.K * q.ldb * 2 < 0xFFFFFFF0ull,
                           "wgrad group: operand exceeds the 4 GiB buffer-descriptor window");
            units += (long long)((q.M + 255) / 256) * ((q.N + 255) / 256);
            const int nk = (q.K + BK - 1) / BK;
            if (nk < nk_min) nk_min = nk;
        }
        int sp = splits;
        if (sp <= 0) {
            // a unit = one full-K tile; s splits cut it into s workgroups of 1/s the work and add s atomic passes over C
            float best = -1.f;
            sp = 1;
            for (int s = 1; s <= 4; ++s) {
                const long long wgs = units * s, rounds = (wgs + ncu - 1) / ncu;
                const float util = (float)wgs / (float)(rounds * ncu) - 0.02f * (float)(s - 1);
                if (util > best) { best = util; sp = s; }
            }
        }
        if (sp > nk_min / 8) sp = nk_min / 8 > 0 ? nk_min / 8 : 1;
        GroupArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.n = n;
        int start = 0;
        for (int i = 0; i < n; ++i) {
            const omlm_gemm_wgrad_desc& q = d[base + i];
            GroupProb& p = ga.p[i];
            p.A = q.A; p.B = q.B; p.C = q.C; p.c_map = q.c_map; p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
            const int nk = (q.K + BK - 1) / BK;
            p.kt_per_split = (nk + sp - 1) / sp;
            const int s_eff = (nk + p.kt_per_split - 1) / p.kt_per_split;
            p.start = start;
            start += ((q.M + 255) / 256) * ((q.N + 255) / 256) * s_eff;
        }
        ga.total = start;
        if (fastk && gemm_t8_mode() > 0) hipLaunchKernelGGL((gemm_wgrad_group_kernel<true, true>), dim3(start), dim3(512), 131072, as_stream(stream), ga);
        else if (fastk) hipLaunchKernelGGL(gemm_wgrad_group_kernel<true>, dim3(start), dim3(512), 131072, as_stream(stream), ga);
        else       hipLaunchKernelGGL(gemm_wgrad_group_kernel<false>, dim3(start), dim3(512), 131072, as_stream(stream), ga);
    }
    return omlm_post_launch("omlm_gemm_wgrad_group");
}

}   // namespace OMLM_NS
#endif   // OMLM_ISA_ONLY
