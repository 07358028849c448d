// Torch-free same-box A/B of two builds of libomlm_hip.so through the C ABI (include/omlm.h): attention backward and GEMM.
// Loads both libraries with dlopen(RTLD_LOCAL), runs the same device inputs through each, compares the outputs bit for bit
// (tolerance only where fp32 atomics make the order free: d(bias), split-K), and times each with HIP events.  A whole run is a few
// seconds -- no Python import on the GPU box.
//   build (here):  hipcc -O2 tools/lib_ab.cpp -o tools/lib_ab -ldl
//   run (GPU box): tools/lib_ab base.so [ENV=V@]variant.so ... -- attn attn_large gemm gemm_edge wgrad ffmid ln
#include <hip/hip_runtime.h>
#include "../include/omlm.h"        // struct layouts only (omlm_decode_args); every call goes through dlsym
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef int (*fwd_t)(const void*, const void*, const void*, const float*, const float*, const unsigned char*, void*, float*, int, int, int, float, int, int, void*);
typedef int (*bwd_t)(const void*, const void*, const void*, const float*, const float*, const unsigned char*, const void*, const void*, const float*, float*,
                     float*, float*, float*, float*, int, int, int, float, int, int, void*);
typedef int (*bwdw_t)(const void*, const void*, const void*, const float*, const float*, const unsigned char*, const void*, const void*, const float*, float*,
                      float*, float*, float*, float*, float*, int, int, int, float, int, int, void*);      // with the d(bias) workspace (round 3c on)
typedef long long (*bwsz_t)(int, int, int);
typedef int (*prep_t)(const float*, float*, int, int, int, const float*, const float*, float, float, void*);
typedef long long (*tbl_t)(int, int);
typedef int (*gemm_t)(const void*, const void*, void*, const float*, const int*, const int*, const int*, long long, long long, int, int, int, int, int, int, int,
                      int, int, int, int, float, void*, long long, void*);
typedef const char* (*err_t)(void);
struct wgrad_desc { const void* A; const void* B; float* C; const int* c_map; int M, N, K, lda, ldb, ldc; };
typedef int (*wgrad_t)(const wgrad_desc*, int, int, int, void*);
typedef int (*ffwd_t)(const void*, const void*, const void*, void*, float*, float*, int, int, int, int, float, float, unsigned long long,
                      const unsigned long long*, unsigned char*, void*, int, void*);
typedef long long (*fws_t)(int, int);
typedef int (*fbwd_t)(const void*, const void*, const void*, const void*, const float*, const float*, void*, void*, float*, float*, float*,
                      int, int, int, int, float, unsigned long long, const unsigned long long*, const unsigned char*, const void*, int, void*);
typedef int (*lnf_t)(const float*, const float*, void*, void*, float*, float*, int, int, int, float, int, void*);
typedef long long (*lnws_t)(int);
typedef int (*lnb_t)(const void*, const float*, const float*, const float*, const float*, const float*, float*, void*, float*, float*, int, int, float, int, int, void*);
typedef int (*dstep_t)(const omlm_decode_args*, const long long*, void*);
typedef int (*planes_t)(const void*, long long, const void*, long long, void*, const float*, const int*, const int*, const int*, long long, long long,
                        int, int, int, int, int, int, int, int, int, int, float, void*, long long, void*);

struct Lib {
    std::string path; void* h; fwd_t fwd; bwd_t bwd; bwdw_t bwdw; bwsz_t bwsz; prep_t prep; tbl_t tbl; gemm_t gemm; err_t err; wgrad_t wgrad; planes_t planes; ffwd_t ffwd; fws_t fws; fbwd_t fbwd; lnf_t lnf; lnws_t lnws; lnb_t lnb; dstep_t dstep;
    void load(const char* p) {
        path = p;
        h = dlopen(p, RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "dlopen %s: %s\n", p, dlerror()); exit(1); }
        fwd = (fwd_t)dlsym(h, "omlm_mqa_attn_fwd"); bwd = (bwd_t)dlsym(h, "omlm_mqa_attn_bwd"); bwdw = (bwdw_t)dlsym(h, "omlm_mqa_attn_bwd");
        bwsz = (bwsz_t)dlsym(h, "omlm_mqa_attn_bwd_workspace_bytes");       // absent in libraries older than the workspace form
        prep = (prep_t)dlsym(h, "omlm_attn_bias_prepare"); tbl = (tbl_t)dlsym(h, "omlm_attn_bias_table_floats");
        gemm = (gemm_t)dlsym(h, "omlm_gemm"); err = (err_t)dlsym(h, "omlm_last_error"); wgrad = (wgrad_t)dlsym(h, "omlm_gemm_wgrad_group"); planes = (planes_t)dlsym(h, "omlm_gemm_planes");
        ffwd = (ffwd_t)dlsym(h, "omlm_ffmid_fwd"); fws = (fws_t)dlsym(h, "omlm_ffmid_bwd_workspace_bytes"); fbwd = (fbwd_t)dlsym(h, "omlm_ffmid_bwd");
        lnf = (lnf_t)dlsym(h, "omlm_layernorm_fwd"); lnws = (lnws_t)dlsym(h, "omlm_layernorm_bwd_workspace_bytes"); lnb = (lnb_t)dlsym(h, "omlm_layernorm_bwd"); dstep = (dstep_t)dlsym(h, "omlm_decode_step");
        if (!fwd || !bwd || !prep || !tbl || !gemm || !err) { fprintf(stderr, "%s: missing symbol\n", p); exit(1); }
    }
    void ok(int rc, const char* what) { if (rc != 0) { fprintf(stderr, "%s: %s failed (%d): %s\n", path.c_str(), what, rc, err()); exit(1); } }
};

static uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

// cheap pseudo-random bf16 in (-1.73, 1.73) (unit variance): host generation must not dominate a GPU call
static void fast_fill(std::vector<uint16_t>& v, float scale, uint64_t seed) {
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    for (auto& e : v) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; const float u = (float)((x >> 40) & 0xFFFFFF) * (1.0f / 16777216.0f); e = bf16((2.f * u - 1.f) * 1.7320508f * scale); }
}

template <typename T> static T* dev(const std::vector<T>& h) { T* d; CK(hipMalloc(&d, h.size() * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
template <typename T> static T* dev_zero(size_t n) { T* d; CK(hipMalloc(&d, n * sizeof(T))); CK(hipMemset(d, 0, n * sizeof(T))); return d; }
template <typename T> static std::vector<T> host(const T* d, size_t n) { std::vector<T> h(n); CK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost)); return h; }

template <typename F> static float time_us(F fn, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fn(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

static std::string g_env_a, g_env_b;
static void apply_env(const std::string& e) {       // "NAME=VALUE" or empty: (un)set before a library's first use of the switch
    static std::string last;
    if (!last.empty()) unsetenv(last.substr(0, last.find('=')).c_str());
    last = e;
    if (!e.empty()) setenv(e.substr(0, e.find('=')).c_str(), e.substr(e.find('=') + 1).c_str(), 1);
}

struct Cmp { size_t n = 0, diff = 0; double maxabs = 0, maxref = 0; bool finite = true; };
static Cmp compare(const std::vector<float>& a, const std::vector<float>& b) {
    Cmp c; c.n = a.size();
    for (size_t i = 0; i < a.size(); ++i) {
        if (memcmp(&a[i], &b[i], 4) != 0) c.diff++;
        if (!std::isfinite(a[i]) || !std::isfinite(b[i])) c.finite = false;
        c.maxabs = std::fmax(c.maxabs, std::fabs((double)a[i] - (double)b[i]));
        c.maxref = std::fmax(c.maxref, std::fabs((double)b[i]));
    }
    return c;
}
static void report(const char* what, const Cmp& c, bool exact_expected) {
    printf("  %-6s %zu values, %zu differ bitwise, max |a-b| %.3e (max |b| %.3e)%s%s\n", what, c.n, c.diff, c.maxabs, c.maxref,
           c.finite ? "" : "  NON-FINITE", exact_expected ? (c.diff ? "  <-- EXPECTED IDENTICAL" : "  identical") : "");
}

static float bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void* dev_as(const std::vector<uint16_t>& h, int dtype) {      // dtype 1: the bf16 words; 0: the same values as fp32 ("bf16x3" operands)
    if (dtype == 1) return dev(h);
    std::vector<float> f(h.size()); for (size_t i = 0; i < h.size(); ++i) f[i] = bf16_to_f(h[i]);
    return dev(f);
}

static void attn_case(Lib& A, Lib& Bl, int B, int N, int H, int dtype = 1) {
    printf("== attention backward  B=%d N=%d H=%d (%s operands)\n", B, N, H, dtype ? "bf16" : "fp32 / bf16x3");
    const size_t M = (size_t)B * N; const int ld = (H + 7) / 8 * 8; const float scale = 8.0f;
    std::mt19937 g(1234); std::normal_distribution<float> nd(0.f, 1.f); std::uniform_real_distribution<float> ud(0.f, 1.f);
    std::vector<uint16_t> q(M * H * 64), k(M * 64), v(M * 64), dout(M * H * 64);
    auto unit = [&](uint16_t* dst) { float t[64]; double s = 0; for (int d = 0; d < 64; ++d) { t[d] = nd(g); s += (double)t[d] * t[d]; } const float r = 1.f / (float)std::sqrt(s); for (int d = 0; d < 64; ++d) dst[d] = bf16(t[d] * r); };
    for (size_t r = 0; r < M * H; ++r) unit(&q[r * 64]);
    for (size_t r = 0; r < M; ++r) unit(&k[r * 64]);
    for (auto& x : v) x = bf16(nd(g));
    for (auto& x : dout) x = bf16(nd(g));
    std::vector<float> bias((size_t)N * ld, 0.f);
    for (int r = 0; r < N; ++r) for (int h = 0; h < H; ++h) bias[(size_t)r * ld + h] = 0.1f * nd(g);
    std::vector<unsigned char> mask(M);
    for (size_t i = 0; i < M; ++i) mask[i] = (i % N == 0) ? 1 : (ud(g) > 0.15f);
    void *dq_ = dev_as(q, dtype), *dk_ = dev_as(k, dtype), *dv_ = dev_as(v, dtype), *ddo = dev_as(dout, dtype);
    float* dbias_in = dev(bias); unsigned char* dmask = dev(mask);
    void* out = dev_zero<unsigned char>(M * H * 64 * (dtype ? 2 : 4));
    float* lse = dev_zero<float>((size_t)B * H * N); float* delta = dev_zero<float>((size_t)B * H * N);
    std::vector<float> res[2][4]; float us[2]; float fwd_us[2]; float us_nb[2];
    Lib* libs[2] = {&A, &Bl};
    for (int li = 0; li < 2; ++li) {
        Lib& L = *libs[li];
        apply_env(li ? g_env_b : g_env_a);
        const long long tf = L.tbl(N, H);
        float* biasT = dev_zero<float>((size_t)tf);
        L.ok(L.prep(dbias_in, biasT, N, H, ld, nullptr, nullptr, 1.0f, scale, nullptr), "bias_prepare");
        L.ok(L.fwd(dq_, dk_, dv_, dbias_in, biasT, dmask, out, lse, B, N, H, scale, ld, dtype, nullptr), "attn_fwd");
        CK(hipDeviceSynchronize());
        float *gq = dev_zero<float>(M * H * 64), *gk = dev_zero<float>(M * 64), *gv = dev_zero<float>(M * 64), *gb = dev_zero<float>((size_t)N * ld);
        float* wsb = L.bwsz && !getenv("LIB_AB_NO_DBIAS_WS") ? dev_zero<float>((size_t)L.bwsz(B, N, H) / 4) : nullptr;
        auto bwd_call = [&](float* gbias) {
            return L.bwsz ? L.bwdw(dq_, dk_, dv_, dbias_in, biasT, dmask, out, ddo, lse, delta, gq, gk, gv, gbias, wsb, B, N, H, scale, ld, dtype, nullptr)
                          : L.bwd(dq_, dk_, dv_, dbias_in, biasT, dmask, out, ddo, lse, delta, gq, gk, gv, gbias, B, N, H, scale, ld, dtype, nullptr);
        };
        L.ok(bwd_call(gb), "attn_bwd");
        CK(hipDeviceSynchronize());
        res[li][0] = host(gq, M * H * 64); res[li][1] = host(gk, M * 64); res[li][2] = host(gv, M * 64); res[li][3] = host(gb, (size_t)N * ld);
        float* scratch_b = dev_zero<float>((size_t)N * ld);
        us[li] = time_us([&] { bwd_call(scratch_b); }, 10);
        us_nb[li] = time_us([&] { bwd_call(nullptr); }, 10);
        fwd_us[li] = time_us([&] { L.fwd(dq_, dk_, dv_, dbias_in, biasT, dmask, out, lse, B, N, H, scale, ld, dtype, nullptr); }, 10);
        CK(hipFree(gq)); CK(hipFree(gk)); CK(hipFree(gv)); CK(hipFree(gb)); CK(hipFree(scratch_b)); CK(hipFree(biasT)); if (wsb) CK(hipFree(wsb));
    }
    const double flops = 4.0 * H * 64 * (double)N * (N + 1) / 2 * B;
    printf("  A %-48s fwd %8.1f us  bwd %8.1f us (%6.1f TFLOP/s at 5 matmuls)  bwd without d(bias) %8.1f us\n", A.path.c_str(), fwd_us[0], us[0], 2.5 * flops / us[0] / 1e6, us_nb[0]);
    printf("  B %-48s fwd %8.1f us  bwd %8.1f us (%6.1f TFLOP/s)               bwd without d(bias) %8.1f us\n", Bl.path.c_str(), fwd_us[1], us[1], 2.5 * flops / us[1] / 1e6, us_nb[1]);
    report("dq", compare(res[0][0], res[1][0]), true);
    report("dk", compare(res[0][1], res[1][1]), true);
    report("dv", compare(res[0][2], res[1][2]), true);
    report("dbias", compare(res[0][3], res[1][3]), false);     // cross-workgroup fp32 atomics: order is free
    CK(hipFree(dq_)); CK(hipFree(dk_)); CK(hipFree(dv_)); CK(hipFree(ddo)); CK(hipFree(dbias_in)); CK(hipFree(dmask)); CK(hipFree(out)); CK(hipFree(lse)); CK(hipFree(delta));
}

static void gemm_case(Lib& A, Lib& Bl) {
    const int M = 35712, D = 1024, F2 = 5472;
    printf("== GEMM  M=%d D=%d F2=%d (bf16 operands)\n", M, D, F2);
    std::mt19937 g(99); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint16_t> X((size_t)M * D), W((size_t)F2 * D), dH((size_t)M * F2);
    fast_fill(X, 1.f, 1); fast_fill(W, 0.03f, 2); fast_fill(dH, 1.f, 3);
    uint16_t *dX_ = dev(X), *dW_ = dev(W), *ddH = dev(dH);
    Lib* libs[2] = {&A, &Bl};
    std::vector<float> r_nn[2], r_tn[2]; std::vector<uint16_t> r_nt[2]; float us[2][3];
    for (int li = 0; li < 2; ++li) {
        Lib& L = *libs[li];
        apply_env(li ? g_env_b : g_env_a);
        uint16_t* Hout = dev_zero<uint16_t>((size_t)M * F2); float* dXo = dev_zero<float>((size_t)M * D); float* dWo = dev_zero<float>((size_t)F2 * D);
        auto nt = [&] { L.ok(L.gemm(dX_, dW_, Hout, nullptr, nullptr, nullptr, nullptr, M, F2, M, F2, D, D, D, F2, 0, 0, 0, 1, 1, 1.f, nullptr, 0, nullptr), "gemm NT"); };
        auto nn = [&] { L.ok(L.gemm(ddH, dW_, dXo, nullptr, nullptr, nullptr, nullptr, M, F2, M, D, F2, F2, D, D, 0, 0, 1, 1, 0, 1.f, nullptr, 0, nullptr), "gemm NN"); };
        auto tn = [&] { L.ok(L.gemm(ddH, dX_, dWo, dWo, nullptr, nullptr, nullptr, M, M, F2, D, M, F2, D, D, D, 1, 1, 1, 0, 1.f, nullptr, 0, nullptr), "gemm TN"); };
        nt(); nn(); tn(); CK(hipDeviceSynchronize());
        r_nt[li] = host(Hout, (size_t)M * F2); r_nn[li] = host(dXo, (size_t)M * D); r_tn[li] = host(dWo, (size_t)F2 * D);
        us[li][0] = time_us(nt, 10); us[li][1] = time_us(nn, 10); us[li][2] = time_us(tn, 10);
        CK(hipFree(Hout)); CK(hipFree(dXo)); CK(hipFree(dWo));
    }
    const double f = 2.0 * M * D * (double)F2;
    for (int li = 0; li < 2; ++li)
        printf("  %c %-40s ffin_NT %7.1f us %6.0f TF | dX_NN %7.1f us %6.0f TF | dW_TN %7.1f us %6.0f TF\n", li ? 'B' : 'A', libs[li]->path.c_str(),
               us[li][0], f / us[li][0] / 1e6, us[li][1], f / us[li][1] / 1e6, us[li][2], f / us[li][2] / 1e6);
    size_t d = 0; for (size_t i = 0; i < r_nt[0].size(); ++i) d += r_nt[0][i] != r_nt[1][i];
    printf("  ffin_NT (bf16 out) %zu values, %zu differ bitwise%s\n", r_nt[0].size(), d, d ? "  <-- EXPECTED IDENTICAL" : "  identical");
    report("dX_NN", compare(r_nn[0], r_nn[1]), true);
    report("dW_TN", compare(r_tn[0], r_tn[1]), false);          // split-K fp32 atomics
    CK(hipFree(dX_)); CK(hipFree(dW_)); CK(hipFree(ddH));
}


// Calibration against the guide's reference ladder (cdna_hip_programming.md "Reference targets": square bf16 GEMMs on uniform random
// operands, NT layout, bf16 output): 4096^3 and 8192^3.
static void gemm_square_case(Lib& A, Lib& Bl) {
    Lib* libs[2] = {&A, &Bl};
    for (int n : {4096, 8192}) {
        std::vector<uint16_t> X((size_t)n * n), W((size_t)n * n);
        fast_fill(X, 1.f, 11); fast_fill(W, 1.f, 12);
        uint16_t *dX_ = dev(X), *dW_ = dev(W);
        printf("== GEMM %d^3 (bf16 operands, NT, bf16 out)\n", n);
        for (int li = 0; li < 2; ++li) {
            Lib& L = *libs[li];
            apply_env(li ? g_env_b : g_env_a);
            uint16_t* out = dev_zero<uint16_t>((size_t)n * n);
            auto nt = [&] { L.ok(L.gemm(dX_, dW_, out, nullptr, nullptr, nullptr, nullptr, n, n, n, n, n, n, n, n, 0, 0, 0, 1, 1, 1.f, nullptr, 0, nullptr), "gemm NT"); };
            nt(); CK(hipDeviceSynchronize());
            const float us = time_us(nt, 20);
            printf("  %c %-40s %8.1f us  %6.0f TFLOP/s\n", li ? 'B' : 'A', L.path.c_str(), us, 2.0 * n * (double)n * n / us / 1e6);
            CK(hipFree(out));
        }
        CK(hipFree(dX_)); CK(hipFree(dW_));
    }
}

// Odd shapes through every layout / output type: B's results must equal A's bit for bit (no split-K here: K is small enough
// that the host picks one slice, so there are no atomics).
static void gemm_edge_case(Lib& A, Lib& Bl) {
    printf("== GEMM edge shapes (bitwise B vs A)\n");
    struct Sh { int M, N, K; };
    const Sh shapes[] = {{300, 520, 200}, {257, 129, 72}, {1000, 1032, 1024}, {2232, 512, 1024}, {35, 40, 64}, {513, 1024, 2736}, {8, 8, 8}, {1116, 128, 1024}};
    std::mt19937 g(7); std::normal_distribution<float> nd(0.f, 1.f);
    Lib* libs[2] = {&A, &Bl};
    size_t bad = 0, total = 0;
    for (const Sh& sh : shapes)
        for (int ak = 0; ak < 2; ++ak) for (int bk = 0; bk < 2; ++bk) for (int od = 0; od < 2; ++od) for (int withc = 0; withc < 2; ++withc) {
            const int M = sh.M, N = sh.N, K = sh.K;
            const int lda = ak ? (M + 7) / 8 * 8 : (K + 7) / 8 * 8, ldb = bk ? (N + 7) / 8 * 8 : (K + 7) / 8 * 8, ldc = (N + 7) / 8 * 8;
            const long long ar = ak ? K : M, br = bk ? K : N;
            std::vector<uint16_t> a((size_t)ar * lda), b((size_t)br * ldb);
            for (auto& x : a) x = bf16(nd(g));
            for (auto& x : b) x = bf16(nd(g));
            std::vector<float> cin((size_t)M * ldc);
            for (auto& x : cin) x = nd(g);
            uint16_t *da = dev(a), *db = dev(b); float* dcin = dev(cin);
            std::vector<unsigned char> out[2];
            for (int li = 0; li < 2; ++li) {
                apply_env(li ? g_env_b : g_env_a);
                const size_t bytes = (size_t)M * ldc * (od ? 2 : 4);
                unsigned char* dc = dev_zero<unsigned char>(bytes);
                libs[li]->ok(libs[li]->gemm(da, db, dc, withc ? dcin : nullptr, nullptr, nullptr, nullptr, ar, br, M, N, K, lda, ldb, ldc, ldc, ak, bk, 1, od, 0.5f, nullptr, 0, nullptr), "gemm edge");
                CK(hipDeviceSynchronize());
                out[li] = host(dc, bytes);
                CK(hipFree(dc));
            }
            ++total;
            if (out[0] != out[1]) { ++bad; printf("  MISMATCH M=%d N=%d K=%d a_kmajor=%d b_kmajor=%d out=%s cin=%d\n", M, N, K, ak, bk, od ? "bf16" : "f32", withc); }
            CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dcin));
        }
    // row maps (gathered A rows, scattered / skipped C rows) and the hi/lo-plane route (3x k-loop), a few shapes each
    for (const Sh& sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K, lda = (K + 7) / 8 * 8, ldb = lda, ldc = (N + 7) / 8 * 8;
        const long long arows = M + 37, crows = M + 5;
        std::vector<uint16_t> a((size_t)arows * lda * 2), b((size_t)N * ldb * 2);          // two planes each (hi | lo)
        for (auto& x : a) x = bf16(nd(g));
        for (auto& x : b) x = bf16(nd(g));
        std::vector<int> am(M), cm(M);
        for (int r = 0; r < M; ++r) { am[r] = (int)(((long long)r * 7919 + 13) % arows); cm[r] = (r % 11 == 3) ? -1 : r; }   // gather; identity with skipped rows
        uint16_t *da = dev(a), *db = dev(b); int *dam = dev(am), *dcm = dev(cm);
        for (int mode = 0; mode < 2; ++mode) {                                                 // 0: maps through omlm_gemm, 1: planes
            std::vector<float> out[2];
            for (int li = 0; li < 2; ++li) {
                apply_env(li ? g_env_b : g_env_a);
                float* dc = dev_zero<float>((size_t)crows * ldc);
                if (mode == 0) libs[li]->ok(libs[li]->gemm(da, db, dc, nullptr, dam, nullptr, dcm, arows, N, M, N, K, lda, ldb, ldc, 0, 0, 0, 1, 0, 1.f, nullptr, 0, nullptr), "gemm maps");
                else libs[li]->ok(libs[li]->planes(da, (long long)arows * lda * 2, db, (long long)N * ldb * 2, dc, nullptr, nullptr, nullptr, nullptr, arows, N, M, N, K, lda, ldb, ldc, 0, 0, 0, 0, 1.f, nullptr, 0, nullptr), "gemm planes");
                CK(hipDeviceSynchronize());
                out[li] = host(dc, (size_t)crows * ldc);
                CK(hipFree(dc));
            }
            ++total;
            if (memcmp(out[0].data(), out[1].data(), out[0].size() * 4) != 0) { ++bad; printf("  MISMATCH %s M=%d N=%d K=%d\n", mode ? "planes" : "maps", M, N, K); }
        }
        CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dam)); CK(hipFree(dcm));
    }
    printf("  %zu configurations, %zu mismatching%s\n", total, bad, bad ? "  <-- EXPECTED IDENTICAL" : "  (all identical)");
}

// The grouped weight-gradient launch on one layer's five problems x `layers` (dW += dY^T X, K = B*N rows): same tile body as omlm_gemm.
static void wgrad_case(Lib& A, Lib& Bl, int kcut = 8) {
    const int K = 35712, D = 1024, layers = 3;
    printf("== grouped weight gradients  K=%d, %d layers x 5 problems\n", K - kcut, layers);
    const int Ms[5] = {512, 128, 1024, 5472, 1024}, Ns[5] = {1024, 1024, 512, 1024, 2736};
    std::vector<uint16_t> big((size_t)K * 5472); fast_fill(big, 1.f, 11);
    std::vector<uint16_t> big2((size_t)K * 2736); fast_fill(big2, 1.f, 12);
    uint16_t *dA = dev(big), *dB = dev(big2);
    Lib* libs[2] = {&A, &Bl};
    std::vector<float> res[2]; float us[2];
    size_t cfl = 0; for (int i = 0; i < 5; ++i) cfl += (size_t)Ms[i] * Ns[i];
    double flops = 0; for (int i = 0; i < 5; ++i) flops += 2.0 * Ms[i] * Ns[i] * (double)K; flops *= layers;
    for (int li = 0; li < 2; ++li) {
        apply_env(li ? g_env_b : g_env_a);
        float* C = dev_zero<float>(cfl * layers);
        std::vector<wgrad_desc> pr;
        size_t off = 0;
        for (int l = 0; l < layers; ++l)
            for (int i = 0; i < 5; ++i) {
                wgrad_desc d; d.A = dA + 8 * l; d.B = dB + 8 * l; d.C = C + off; d.c_map = nullptr;      // k-major operands: [K, M] / [K, N] views of the big buffers
                d.M = Ms[i]; d.N = Ns[i]; d.K = K - kcut; d.lda = 5472; d.ldb = 2736; d.ldc = Ns[i];
                off += (size_t)Ms[i] * Ns[i];
                pr.push_back(d);
            }
        auto run = [&] { libs[li]->ok(libs[li]->wgrad(pr.data(), (int)pr.size(), 0, 1, nullptr), "wgrad_group"); };
        run(); CK(hipDeviceSynchronize());
        res[li] = host(C, cfl * layers);
        us[li] = time_us(run, 5);
        CK(hipFree(C));
    }
    for (int li = 0; li < 2; ++li) printf("  %c %-48s %8.1f us  %6.0f TFLOP/s\n", li ? 'B' : 'A', libs[li]->path.c_str(), us[li], flops / us[li] / 1e6);
    report("dW", compare(res[0], res[1]), false);
    CK(hipFree(dA)); CK(hipFree(dB));
}

static void fill_f32(std::vector<float>& v, float scale, uint64_t seed) {
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    for (auto& e : v) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; e = ((float)((x >> 40) & 0xFFFFFF) * (2.0f / 16777216.0f) - 1.f) * 1.7320508f * scale; }
}
template <typename T> static size_t count_diff(const std::vector<T>& a, const std::vector<T>& b) { size_t d = 0; for (size_t i = 0; i < a.size(); ++i) d += memcmp(&a[i], &b[i], sizeof(T)) != 0; return d; }

// Round-5 shapes of the coarse-small step (Fp = 2752: every K a whole number of 64-deep k-tiles): the four large non-grouped GEMMs of a layer,
// A vs B bitwise (same products in the same order per accumulator) and timed in interleaved rounds.
static void gemm5_case(Lib& A, Lib& Bl) {
    const int M = 35712, D = 1024, Fp = 2752, F2 = 2 * Fp;
    printf("== GEMM round-5 shapes  M=%d D=%d Fp=%d (bf16 operands)\n", M, D, Fp);
    std::vector<uint16_t> X((size_t)M * D), W1((size_t)F2 * D), H1((size_t)M * F2), H2((size_t)M * Fp), W2((size_t)D * Fp);
    fast_fill(X, 1.f, 1); fast_fill(W1, 0.03f, 2); fast_fill(H1, 1.f, 3); fast_fill(H2, 1.f, 4); fast_fill(W2, 0.02f, 5);
    std::vector<float> R((size_t)M * D); fill_f32(R, 1.f, 6);
    uint16_t *dX = dev(X), *dW1 = dev(W1), *dH1 = dev(H1), *dH2 = dev(H2), *dW2 = dev(W2); float* dR = dev(R);
    Lib* libs[2] = {&A, &Bl};
    uint16_t *o_ffin[2], *o_dh2[2], *o_dxn2[2]; float* o_ffout[2];
    for (int li = 0; li < 2; ++li) { o_ffin[li] = dev_zero<uint16_t>((size_t)M * F2); o_dh2[li] = dev_zero<uint16_t>((size_t)M * Fp); o_dxn2[li] = dev_zero<uint16_t>((size_t)M * D); o_ffout[li] = dev_zero<float>((size_t)M * D); }
    auto ffin = [&](int li) { libs[li]->ok(libs[li]->gemm(dX, dW1, o_ffin[li], nullptr, nullptr, nullptr, nullptr, M, F2, M, F2, D, D, D, F2, 0, 0, 0, 1, 1, 1.f, nullptr, 0, nullptr), "ffin"); };
    auto dh2 = [&](int li) { libs[li]->ok(libs[li]->gemm(dX, dW2, o_dh2[li], nullptr, nullptr, nullptr, nullptr, M, D, M, Fp, D, D, Fp, Fp, 0, 0, 1, 1, 1, 1.f, nullptr, 0, nullptr), "dh2"); };       // dres [M, D] x W2p [D, Fp] k-major
    auto dxn2 = [&](int li) { libs[li]->ok(libs[li]->gemm(dH1, dW1, o_dxn2[li], nullptr, nullptr, nullptr, nullptr, M, F2, M, D, F2, F2, D, D, 0, 0, 1, 1, 1, 1.f, nullptr, 0, nullptr), "dxn2"); };  // dh1 [M, 2Fp] x W1p [2Fp, D] k-major
    auto ffout = [&](int li) { libs[li]->ok(libs[li]->gemm(dH2, dW2, o_ffout[li], dR, nullptr, nullptr, nullptr, M, D, M, D, Fp, Fp, Fp, D, D, 0, 0, 1, 0, 1.f, nullptr, 0, nullptr), "ffout"); };
    const char* names[4] = {"ffin_NT 16b", "dh2_NN 16b", "dxn2_NN 16b", "ffout_NT f32+res"};
    const double fl[4] = {2.0 * M * D * (double)F2, 2.0 * M * D * (double)Fp, 2.0 * M * D * (double)F2, 2.0 * M * D * (double)Fp};
    double best[2][4]; for (auto& r : best) for (auto& v : r) v = 1e30;
    double sum[2][4] = {{0}};
    const int rounds = 4;
    for (int r = 0; r < rounds + 1; ++r)
        for (int li = 0; li < 2; ++li) {
            apply_env(li ? g_env_b : g_env_a);
            for (int c = 0; c < 4; ++c) {
                auto fn = [&] { if (c == 0) ffin(li); else if (c == 1) dh2(li); else if (c == 2) dxn2(li); else ffout(li); };
                const float us = time_us(fn, 5);
                if (r > 0) { best[li][c] = std::fmin(best[li][c], (double)us); sum[li][c] += us; }
            }
        }
    CK(hipDeviceSynchronize());
    for (int c = 0; c < 4; ++c)
        printf("  %-18s A %7.1f us (min %7.1f) %5.0f TF | B %7.1f us (min %7.1f) %5.0f TF | B/A time %.3f\n", names[c], sum[0][c] / rounds, best[0][c], fl[c] / (sum[0][c] / rounds) / 1e6,
               sum[1][c] / rounds, best[1][c], fl[c] / (sum[1][c] / rounds) / 1e6, sum[1][c] / sum[0][c]);
    auto cmp16 = [&](const char* n, uint16_t* a, uint16_t* b, size_t cnt) { auto ha = host(a, cnt), hb = host(b, cnt); size_t d = count_diff(ha, hb); size_t nz = 0; for (auto v : hb) nz += v != 0; printf("  %-12s %zu values (%zu non-zero), %zu differ bitwise%s\n", n, cnt, nz, d, d ? "  <-- EXPECTED IDENTICAL" : "  identical"); };
    cmp16("ffin", o_ffin[0], o_ffin[1], (size_t)M * F2); cmp16("dh2", o_dh2[0], o_dh2[1], (size_t)M * Fp); cmp16("dxn2", o_dxn2[0], o_dxn2[1], (size_t)M * D);
    { auto ha = host(o_ffout[0], (size_t)M * D), hb = host(o_ffout[1], (size_t)M * D); report("ffout", compare(ha, hb), true); }
    for (int li = 0; li < 2; ++li) { CK(hipFree(o_ffin[li])); CK(hipFree(o_dh2[li])); CK(hipFree(o_dxn2[li])); CK(hipFree(o_ffout[li])); }
    CK(hipFree(dX)); CK(hipFree(dW1)); CK(hipFree(dH1)); CK(hipFree(dH2)); CK(hipFree(dW2)); CK(hipFree(dR));
}

// ConvFeedForward middle, bf16 operands, coarse-small micro-batch (M = 32 x 1116 rows, F = 2730 -> Fp = 2736), dropout 0.1 through
// the stored keep bits, gh saved: the configuration of the training step.
static void ffmid_case(Lib& A, Lib& Bl) {
    const int nseq = 1116, B = 32, M = B * nseq, F = 2730, Fp = 2736;
    printf("== ffmid  M=%d F=%d Fp=%d (bf16 operands, p = 0.1)\n", M, F, Fp);
    std::vector<uint16_t> h1((size_t)M * 2 * Fp), cw((size_t)3 * 2 * Fp), gm(Fp), dh2((size_t)M * Fp);
    fast_fill(h1, 1.f, 21); fast_fill(cw, 0.5f, 22); fast_fill(gm, 1.f, 23); fast_fill(dh2, 1.f, 24);
    for (int c = F; c < Fp; ++c) gm[c] = 0;
    uint16_t *dh1 = dev(h1), *dcw = dev(cw), *dgm = dev(gm), *ddh2 = dev(dh2);
    Lib* libs[2] = {&A, &Bl};
    std::vector<uint16_t> r_h2[2], r_gh[2], r_dh1[2]; std::vector<unsigned char> r_bits[2]; std::vector<float> r_stat[2], r_dg[2], r_dc[2]; float us[2][2];
    for (int li = 0; li < 2; ++li) {
        Lib& L = *libs[li];
        apply_env(li ? g_env_b : g_env_a);
        uint16_t *h2 = dev_zero<uint16_t>((size_t)M * Fp), *gh = dev_zero<uint16_t>((size_t)M * Fp), *du = dev_zero<uint16_t>((size_t)M * 2 * Fp), *g1 = dev_zero<uint16_t>((size_t)M * 2 * Fp);
        float *stat = dev_zero<float>((size_t)2 * M), *dg = dev_zero<float>(Fp), *dc = dev_zero<float>((size_t)2 * F * 3);
        unsigned char* bits = dev_zero<unsigned char>((size_t)M * Fp / 8);
        const long long wsb = L.fws(F, Fp);
        float* ws = dev_zero<float>((size_t)(wsb + 3) / 4 + 4);
        auto fwd = [&] { L.ok(L.ffwd(dh1, dcw, dgm, h2, stat, stat + M, M, nseq, F, Fp, 1e-5f, 0.1f, 1234ull, nullptr, bits, gh, 1, nullptr), "ffmid_fwd"); };
        auto bwd = [&] { L.ok(L.fbwd(ddh2, dh1, dcw, dgm, stat, stat + M, du, g1, dg, dc, ws, M, nseq, F, Fp, 0.1f, 1234ull, nullptr, bits, gh, 1, nullptr), "ffmid_bwd"); };
        fwd(); bwd(); CK(hipDeviceSynchronize());
        r_h2[li] = host(h2, (size_t)M * Fp); r_gh[li] = host(gh, (size_t)M * Fp); r_bits[li] = host(bits, (size_t)M * Fp / 8); r_stat[li] = host(stat, (size_t)2 * M);
        r_dh1[li] = host(g1, (size_t)M * 2 * Fp); r_dg[li] = host(dg, (size_t)Fp); r_dc[li] = host(dc, (size_t)2 * F * 3);
        us[li][0] = time_us(fwd, 10); us[li][1] = time_us(bwd, 10);
        CK(hipFree(h2)); CK(hipFree(gh)); CK(hipFree(du)); CK(hipFree(g1)); CK(hipFree(stat)); CK(hipFree(dg)); CK(hipFree(dc)); CK(hipFree(bits)); CK(hipFree(ws));
    }
    for (int li = 0; li < 2; ++li) printf("  %c %-48s fwd %8.1f us  bwd %8.1f us\n", li ? 'B' : 'A', libs[li]->path.c_str(), us[li][0], us[li][1]);
    printf("  differing values: h2 %zu, gh %zu, keep bits %zu, mean/rstd %zu, dh1 %zu (all expected 0)\n", count_diff(r_h2[0], r_h2[1]), count_diff(r_gh[0], r_gh[1]),
           count_diff(r_bits[0], r_bits[1]), count_diff(r_stat[0], r_stat[1]), count_diff(r_dh1[0], r_dh1[1]));
    report("dgamma", compare(r_dg[0], r_dg[1]), false);
    report("dconv", compare(r_dc[0], r_dc[1]), false);
    CK(hipFree(dh1)); CK(hipFree(dcw)); CK(hipFree(dgm)); CK(hipFree(ddh2));
}

// LayerNorm forward / backward at the residual-stream shape (M = 35712, D = 1024), bf16 output / bf16 dy as in the training step.
static void ln_case(Lib& A, Lib& Bl) {
    const int M = 35712, D = 1024;
    printf("== layernorm  M=%d D=%d (fp32 x, bf16 y / dy)\n", M, D);
    std::vector<float> x((size_t)M * D), gam(D), dres((size_t)M * D); std::vector<uint16_t> dy((size_t)M * D);
    fill_f32(x, 1.f, 31); fill_f32(gam, 1.f, 32); fill_f32(dres, 1.f, 33); fast_fill(dy, 1.f, 34);
    float *dx_ = dev(x), *dgam = dev(gam), *ddres = dev(dres); uint16_t* ddy = dev(dy);
    Lib* libs[2] = {&A, &Bl};
    std::vector<uint16_t> r_y[2], r_xc[2], r_dxc[2]; std::vector<float> r_st[2], r_dx[2], r_dg[2]; float us[2][2];
    for (int li = 0; li < 2; ++li) {
        Lib& L = *libs[li];
        apply_env(li ? g_env_b : g_env_a);
        uint16_t *y = dev_zero<uint16_t>((size_t)M * D), *xc = dev_zero<uint16_t>((size_t)M * D), *dxc = dev_zero<uint16_t>((size_t)M * D);
        float *st = dev_zero<float>((size_t)2 * M), *dxo = dev_zero<float>((size_t)M * D), *dg = dev_zero<float>(D);
        float* ws = dev_zero<float>((size_t)(L.lnws(D) + 3) / 4 + 4);
        auto fwd = [&] { L.ok(L.lnf(dx_, dgam, y, xc, st, st + M, M, D, D, 1e-5f, 1, nullptr), "ln_fwd"); };
        auto bwd = [&] { L.ok(L.lnb(ddy, dx_, dgam, st, st + M, ddres, dxo, dxc, dg, ws, M, D, 1.0f, 1, 1, nullptr), "ln_bwd"); };
        fwd(); bwd(); CK(hipDeviceSynchronize());
        r_y[li] = host(y, (size_t)M * D); r_xc[li] = host(xc, (size_t)M * D); r_dxc[li] = host(dxc, (size_t)M * D);
        r_st[li] = host(st, (size_t)2 * M); r_dx[li] = host(dxo, (size_t)M * D); r_dg[li] = host(dg, (size_t)D);
        us[li][0] = time_us(fwd, 10); us[li][1] = time_us(bwd, 10);
        CK(hipFree(y)); CK(hipFree(xc)); CK(hipFree(dxc)); CK(hipFree(st)); CK(hipFree(dxo)); CK(hipFree(dg)); CK(hipFree(ws));
    }
    for (int li = 0; li < 2; ++li) printf("  %c %-48s fwd %8.1f us  bwd %8.1f us\n", li ? 'B' : 'A', libs[li]->path.c_str(), us[li][0], us[li][1]);
    printf("  differing values: y %zu, xcast %zu, mean/rstd %zu, dx %zu, dxcast %zu (all expected 0)\n", count_diff(r_y[0], r_y[1]), count_diff(r_xc[0], r_xc[1]),
           count_diff(r_st[0], r_st[1]), count_diff(r_dx[0], r_dx[1]), count_diff(r_dxc[0], r_dxc[1]));
    report("dgamma", compare(r_dg[0], r_dg[1]), false);
    CK(hipFree(dx_)); CK(hipFree(dgam)); CK(hipFree(ddres)); CK(hipFree(ddy));
}

// One KV-cached decode step (coarse-small trunk: 6 layers, D = 1024, H = 8, F = 2730; bf16 weights) at row `pos` of an Nmax-row
// cache, repeated without advancing the row: us per id as the sampling loop sees it (32 launches), logits compared bit for bit.
static void decode_case(Lib& A, Lib& Bl, int B) {
    const int L = 6, D = 1024, H = 8, F = 2730, Fp = 2752 /* engine: F rounded up to 64 */, Nmax = 1116, pos = 600, V1 = 1025, ldV = 1032, HD = H * 64;
    printf("== decode step  B=%d  (row %d of %d, bf16 weights)\n", B, pos, Nmax);
    std::vector<const void*> Wq(L), Wkv(L), Wo(L), W1(L), W2(L);
    std::vector<const float*> ag(L), qs(L), ks(L), fg(L), cw(L), mg(L);
    std::vector<float*> Kc(L), Vc(L), hist(L);
    auto wdev = [&](size_t n, float sc, uint64_t seed) { std::vector<uint16_t> h(n); fast_fill(h, sc, seed); return (const void*)dev(h); };
    auto fdev = [&](size_t n, float sc, uint64_t seed, float add) { std::vector<float> h(n); fill_f32(h, sc, seed); for (auto& x : h) x += add; return dev(h); };
    for (int l = 0; l < L; ++l) {
        Wq[l] = wdev((size_t)HD * D, 0.03f, 100 + l); Wkv[l] = wdev((size_t)128 * D, 0.03f, 110 + l); Wo[l] = wdev((size_t)D * HD, 0.03f, 120 + l);
        W1[l] = wdev((size_t)2 * Fp * D, 0.03f, 130 + l); W2[l] = wdev((size_t)D * Fp, 0.02f, 140 + l);
        ag[l] = fdev(D, 0.05f, 150 + l, 1.f); qs[l] = fdev(64, 0.05f, 160 + l, 1.f); ks[l] = fdev(64, 0.05f, 170 + l, 1.f); fg[l] = fdev(D, 0.05f, 180 + l, 1.f);
        cw[l] = fdev((size_t)3 * 2 * Fp, 0.3f, 190 + l, 0.f);
        { std::vector<float> g(Fp); fill_f32(g, 0.05f, 200 + l); for (int c = 0; c < Fp; ++c) g[c] = c < F ? g[c] + 1.f : 0.f; mg[l] = dev(g); }
    }
    // the step WRITES the cache row and shifts the conv history: every library starts from its own fresh copy
    auto fresh_state = [&] {
        for (int l = 0; l < L; ++l) {
            Kc[l] = fdev((size_t)B * Nmax * 64, 0.1f, 210 + l, 0.f); Vc[l] = fdev((size_t)B * Nmax * 64, 1.f, 220 + l, 0.f); hist[l] = fdev((size_t)B * 2 * 2 * Fp, 1.f, 230 + l, 0.f);
        }
    };
    // (the per-layer pointer arrays of omlm_decode_args are HOST arrays of device pointers)
    float* table = fdev((size_t)Nmax * 8, 0.1f, 300, 0.f);
    float* fgam = fdev(D, 0.05f, 301, 1.f);
    const void* headW = wdev((size_t)ldV * D, 0.03f, 302);
    float* emb = fdev((size_t)(1024 * 3 + 1) * D, 0.5f, 303, 0.f);
    std::vector<long long> ids(B); for (int b = 0; b < B; ++b) ids[b] = 17 + 31 * b;
    long long* dids = dev(ids);
    std::vector<int> p1(1, pos); int* dpos = dev(p1);
    const int nsplit = (Nmax + 63) / 64;
    Lib* libs[2] = {&A, &Bl};
    std::vector<float> res[2]; float us[2];
    for (int li = 0; li < 2; ++li) {
        apply_env(li ? g_env_b : g_env_a);
        fresh_state();
        omlm_decode_args a; memset(&a, 0, sizeof(a));
        a.B = B; a.D = D; a.H = H; a.L = L; a.F = F; a.Fp = Fp; a.Nmax = Nmax; a.w_dtype = 1; a.round_bf16 = 1; a.nsplit = nsplit;
        a.eps = 1e-5f; a.scale = 8.0f; a.pos_dev = dpos;
        a.Wq = Wq.data(); a.Wkv = Wkv.data(); a.Wo = Wo.data(); a.W1p = W1.data(); a.W2p = W2.data();
        a.attn_gamma = ag.data(); a.q_scale = qs.data(); a.k_scale = ks.data(); a.ffin_gamma = fg.data(); a.convw = cw.data(); a.mid_gamma = mg.data();
        a.Kc = Kc.data(); a.Vc = Vc.data(); a.hist = hist.data();
        a.bias_table = table; a.bias_ld = 8; a.final_gamma = fgam; a.head_W = headW; a.V1 = V1; a.ldV = ldV;
        a.emb_table = emb; a.emb_row_offset = 1024; a.emb_rows = 1024 * 3 + 1;
        a.x = dev_zero<float>((size_t)B * D); a.x1 = dev_zero<float>((size_t)B * D); a.q = dev_zero<float>((size_t)B * HD);
        a.parts = dev_zero<float>((size_t)B * nsplit * H * 66); a.u = dev_zero<float>((size_t)B * Fp); a.logits = dev_zero<float>((size_t)B * ldV);
        if (!getenv("LIB_AB_NO_LN_PARTS")) a.ln_parts = dev_zero<float>((size_t)3 * OMLM_DECODE_LN_PARTS(D, Fp));      // (libraries older than the field never read it)
        a.advance_pos = nullptr; a.advance_step = nullptr;
        auto step = [&] { libs[li]->ok(libs[li]->dstep(&a, dids, nullptr), "decode_step"); };
        step(); CK(hipDeviceSynchronize());
        res[li] = host(a.logits, (size_t)B * ldV);
        us[li] = time_us(step, 50);
        CK(hipFree(a.x)); CK(hipFree(a.x1)); CK(hipFree(a.q)); CK(hipFree(a.parts)); CK(hipFree(a.u)); CK(hipFree(a.logits));
    }
    for (int li = 0; li < 2; ++li) printf("  %c %-48s %8.1f us per step  (%6.0f ids/s)\n", li ? 'B' : 'A', libs[li]->path.c_str(), us[li], B * 1e6 / us[li]);
    report("logits", compare(res[0], res[1]), true);
}

int main(int argc, char** argv) {
    // usage: lib_ab base.so [NAME=VALUE@]variant.so ... -- case ...      (NAME=VALUE is exported before that library's first call:
    //        the libraries cache their OMLM_* switches per instance, so the same code can be timed under two settings from two copies)
    std::vector<Lib*> libs; std::vector<std::string> envs; int i = 1;
    for (; i < argc && strcmp(argv[i], "--"); ++i) {
        std::string a = argv[i], env;
        const size_t at = a.find('@');
        if (at != std::string::npos) { env = a.substr(0, at); a = a.substr(at + 1); }
        Lib* L = new Lib; L->load(a.c_str()); if (!env.empty()) L->path = env + "@" + a;
        libs.push_back(L); envs.push_back(env);
    }
    if (libs.size() < 2 || i >= argc) { fprintf(stderr, "usage: %s base.so [ENV=V@]variant.so ... -- attn|attn_large|attn32|gemm|gemm_edge|wgrad|ffmid|ln|decode ...\n", argv[0]); return 2; }
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); printf("device: %s, %d CUs\n", pr.name, pr.multiProcessorCount);
    for (++i; i < argc; ++i)
        for (size_t v = 1; v < libs.size(); ++v) {
            g_env_a = envs[0]; g_env_b = envs[v];
            if (!strcmp(argv[i], "attn")) attn_case(*libs[0], *libs[v], 32, 1116, 8);
            else if (!strcmp(argv[i], "attn_large")) attn_case(*libs[0], *libs[v], 8, 1817, 16);
            else if (!strcmp(argv[i], "attn32")) attn_case(*libs[0], *libs[v], 32, 1116, 8, 0);
            else if (!strcmp(argv[i], "gemm")) gemm_case(*libs[0], *libs[v]);
            else if (!strcmp(argv[i], "gemm_edge")) gemm_edge_case(*libs[0], *libs[v]);
            else if (!strcmp(argv[i], "gemm_sq")) gemm_square_case(*libs[0], *libs[v]);
            else if (!strcmp(argv[i], "gemm5")) gemm5_case(*libs[0], *libs[v]);
            else if (!strcmp(argv[i], "wgrad")) wgrad_case(*libs[0], *libs[v]);
            else if (!strcmp(argv[i], "wgrad5")) wgrad_case(*libs[0], *libs[v], 64);        // K = 35648: whole 64-deep k-tiles (the step's own K = 35712 is one too)
            else if (!strcmp(argv[i], "ffmid")) ffmid_case(*libs[0], *libs[v]);
            else if (!strcmp(argv[i], "ln")) ln_case(*libs[0], *libs[v]);
            else if (!strcmp(argv[i], "decode")) { decode_case(*libs[0], *libs[v], 1); decode_case(*libs[0], *libs[v], 8); }
        }
    return 0;
}
