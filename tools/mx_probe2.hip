// Stand-alone probe (not product code): which (lane, byte) operand elements of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 x fp8) does the scale
// byte of lane ls apply to?  One wave per (ls, data lane ld): a one-hot A (or B) element against an all-ones partner, scale 2.0 in lane ls only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int SIDE>      // 0: probe A's scale, 1: probe B's scale
__global__ void probe(unsigned char* out) {
    const int ls = blockIdx.x, ld = blockIdx.y, l = threadIdx.x;
    v8i ones; for (int i = 0; i < 8; ++i) ones[i] = 0x38383838;
    const int sc = (l == ls) ? 0x80808080 : 0x7f7f7f7f;
    for (int jd = 0; jd < 32; ++jd) {
        v8i hot; for (int i = 0; i < 8; ++i) hot[i] = 0;
        if (l == ld) hot[jd >> 2] = 0x38 << (8 * (jd & 3));
        v16f acc = {};
        if (SIDE == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(hot, ones, acc, 0, 0, 0, sc, 0, 0x7f7f7f7f);
        else           acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, hot, acc, 0, 0, 0, 0x7f7f7f7f, 0, sc);
        float m = 0.f;
        for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[r]);
        for (int s = 1; s < 64; s <<= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
        if (l == 0) out[(ls * 64 + ld) * 32 + jd] = (unsigned char)(m + 0.5f);
    }
}

template <int OP>
__global__ void opsel_probe(float* out) {
    v8i ones; for (int i = 0; i < 8; ++i) ones[i] = 0x38383838;
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, ones, acc, 0, 0, OP, 0x8281807f, 0, 0x7f7f7f7f);
    if (threadIdx.x == 0) out[OP] = acc[0];
    v16f acc2 = {};
    acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, ones, acc2, 0, 0, 0, 0x7f7f7f7f, OP, 0x8281807f);
    if (threadIdx.x == 0) out[4 + OP] = acc2[0];
}

int main() {
    unsigned char* d; CK(hipMalloc(&d, 64 * 64 * 32));
    std::vector<unsigned char> h(64 * 64 * 32);
    for (int side = 0; side < 2; ++side) {
        if (side == 0) probe<0><<<dim3(64, 64), 64>>>(d); else probe<1><<<dim3(64, 64), 64>>>(d);
        CK(hipMemcpy(h.data(), d, h.size(), hipMemcpyDeviceToHost));
        printf("== scale of %s, lane ls -> operand elements (lane ld: byte mask) multiplied by 2\n", side ? "B" : "A");
        for (int ls = 0; ls < 64; ++ls) {
            printf("ls %2d:", ls);
            for (int ld = 0; ld < 64; ++ld) {
                unsigned mask = 0; int bad = 0;
                for (int j = 0; j < 32; ++j) { unsigned char v = h[(ls * 64 + ld) * 32 + j]; if (v == 2) mask |= 1u << j; else if (v != 1) ++bad; }
                if (mask || bad) printf(" ld %2d:%08x%s", ld, mask, bad ? "(!)" : "");
            }
            printf("\n");
            if (ls == 3) { ls = 15; }             // a sample of lanes: 0-3, 16-19, 32-35, 48-51
            else if (ls == 19) { ls = 31; }
            else if (ls == 35) { ls = 47; }
            else if (ls == 51) break;
        }
    }
    float* o; CK(hipMalloc(&o, 64)); float ho[8];
    opsel_probe<0><<<1, 64>>>(o); opsel_probe<1><<<1, 64>>>(o); opsel_probe<2><<<1, 64>>>(o); opsel_probe<3><<<1, 64>>>(o);
    CK(hipMemcpy(ho, o, 32, hipMemcpyDeviceToHost));
    printf("opsel (scale dword 0x8281807f, all ones): A-side %g %g %g %g  B-side %g %g %g %g (64 = byte 0x7f, 128 = 0x80, 256 = 0x81, 512 = 0x82)\n",
           ho[0], ho[1], ho[2], ho[3], ho[4], ho[5], ho[6], ho[7]);
    return 0;
}
