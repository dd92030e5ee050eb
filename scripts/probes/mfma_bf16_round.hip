// How does v_mfma_f32_16x16x32_bf16 round?  One wavefront, hand-made operands (all bf16-exact), results printed beside what
// round-to-nearest-even of the exact sum would give.
//   build: hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma_bf16_round.hip -o scripts/probes/mfma_bf16_round
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
// A[16][32], B[32][16] given as float (bf16-exact), C scalar broadcast: D = A B + C
__global__ void k(const float *A, const float *B, const float *C, float *D) {
    const int lane = threadIdx.x, lr = lane >> 4, lc = lane & 15;
    bf8_t a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)A[lc * 32 + 8 * lr + q]; b[q] = (__bf16)B[(8 * lr + q) * 16 + lc]; }
    f4_t c;
    for (int i = 0; i < 4; ++i) c[i] = C[(4 * lr + i) * 16 + lc];
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * lr + i) * 16 + lc] = c[i];
}
int main() {
    std::vector<float> A(16 * 32, 0.f), B(32 * 16, 0.f), C(256, 0.f), D(256);
    // row r of A x column 0 of B: case r.  B(:,0) = 1 for all k.
    for (int kk = 0; kk < 32; ++kk) B[kk * 16 + 0] = 1.0f;
    const float u = ldexpf(1.0f, -23);              // ulp of 1.0
    struct Case { const char *what; float c; std::vector<float> a; };
    std::vector<Case> cs = {
        { "1 + 0.75 ulp (one product)", 1.0f, { 0.75f * u } },
        { "1 + 0.5 ulp (tie, even below)", 1.0f, { 0.5f * u } },
        { "1+ulp + 0.5 ulp (tie, odd below)", 1.0f + u, { 0.5f * u } },
        { "1 + 0.25 ulp", 1.0f, { 0.25f * u } },
        { "-1 - 0.75 ulp", -1.0f, { -0.75f * u } },
        { "1 - 0.25 ulp(0.5)", 1.0f, { -0.25f * u * 0.5f } },
        { "1 + 32 x 1/32 x 0.75 ulp", 1.0f, std::vector<float>(32, 0.75f * u / 32) },
        { "1 + 3 x 0.25 ulp (three products)", 1.0f, { 0.25f * u, 0.25f * u, 0.25f * u } },
        { "1 + (0.5 ulp + 2^-40)", 1.0f, { 0.5f * u, ldexpf(1.0f, -40) } },
        { "0 + 1 + 2^-24 + 2^-24 (products only)", 0.0f, { 1.0f, ldexpf(1.0f, -24), ldexpf(1.0f, -24) } },
        { "0 + 1 + 2^-25 x 3 (products only)", 0.0f, { 1.0f, ldexpf(1.0f, -25), ldexpf(1.0f, -25), ldexpf(1.0f, -25) } },
        { "0 + big cancel: 256 - 256 + 2^-20", 0.0f, { 256.0f, -256.0f, ldexpf(1.0f, -20) } },
        { "0 + 1 + 2^-30 x 16", 0.0f, { 1.0f, ldexpf(1.f,-30), ldexpf(1.f,-30), ldexpf(1.f,-30), ldexpf(1.f,-30), ldexpf(1.f,-30), ldexpf(1.f,-30), ldexpf(1.f,-30), ldexpf(1.f,-30),
                                     ldexpf(1.f,-30), ldexpf(1.f,-30), ldexpf(1.f,-30), ldexpf(1.f,-30), ldexpf(1.f,-30), ldexpf(1.f,-30), ldexpf(1.f,-30), ldexpf(1.f,-30) } },
    };
    for (size_t r = 0; r < cs.size() && r < 16; ++r) {
        for (size_t q = 0; q < cs[r].a.size(); ++q) A[r * 32 + q] = cs[r].a[q];
        C[r * 16 + 0] = cs[r].c;
    }
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    for (size_t r = 0; r < cs.size() && r < 16; ++r) {
        long double ex = cs[r].c;
        for (float v : cs[r].a) ex += (long double)v;
        const float rn = (float)ex;
        uint32_t bd, bn; memcpy(&bd, &D[r * 16], 4); memcpy(&bn, &rn, 4);
        printf("%-40s mfma %.9g (0x%08x)   round-to-nearest of the exact sum %.9g (0x%08x)  %s\n", cs[r].what, D[r * 16], bd, rn, bn, bd == bn ? "same" : "DIFFERENT");
    }
    return 0;
}
