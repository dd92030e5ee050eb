// Probe: what rounding sequence does v_mfma_f64_16x16x4_f64 perform?  Writes A(16x4) B(4x16) C(16x16) D(16x16) per
// trial to a binary file; scripts/probes/mfma_f64_order.py compares D against candidate evaluation orders.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, const double* C, double* D) {
    const int lane = threadIdx.x, t = blockIdx.x;
    A += t * 64; B += t * 64; C += t * 256; D += t * 256;
    // A: lane holds A[row = lane&15][k = lane>>4];  B: lane holds B[k = lane>>4][col = lane&15]
    const double a = A[(lane & 15) * 4 + (lane >> 4)];
    const double b = B[(lane >> 4) * 16 + (lane & 15)];
    d4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[((lane >> 4) + 4 * r) * 16 + (lane & 15)];
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = d[r];
}
int main(int argc, char** argv) {
    const int trials = 64;
    std::mt19937_64 rng(12345);
    std::vector<double> A(trials * 64), B(trials * 64), C(trials * 256), D(trials * 256);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    for (int t = 0; t < trials; ++t) {
        // mixed magnitudes so that rounding order matters
        const double sa = (t % 4 == 0) ? 1.0 : (t % 4 == 1) ? 1e3 : (t % 4 == 2) ? 1e-3 : 1e8;
        for (int i = 0; i < 64; ++i) { A[t * 64 + i] = u(rng) * ((i % 4 == t % 3) ? sa : 1.0); B[t * 64 + i] = u(rng); }
        for (int i = 0; i < 256; ++i) C[t * 256 + i] = u(rng) * ((t % 5 == 0) ? 1e-6 : 1.0);
    }
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    k<<<trials, 64>>>(dA, dB, dC, dD);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
    FILE* f = fopen(argc > 1 ? argv[1] : "mfma_probe.bin", "wb");
    int hdr = trials; fwrite(&hdr, 4, 1, f);
    fwrite(A.data(), 8, A.size(), f); fwrite(B.data(), 8, B.size(), f); fwrite(C.data(), 8, C.size(), f); fwrite(D.data(), 8, D.size(), f);
    fclose(f);
    printf("wrote %d trials\n", trials);
    return 0;
}
