// v_mfma_f64_16x16x4_f64 operand / result layout probe (round 5, k_gcn_fc_mfma): A[i][k] = 1 at one (i, k), B[k][j] = 1 at one (k, j) --
// which lane and register of D lights up?   hipcc --offload-arch=gfx950 tools/micro/mfma_f64_layout.hip -o build_micro/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D) {      // A [16][4], B [4][16] row-major; assumed: lane l holds A[l % 16][l / 16], B[l / 16][l % 16]
    const int l = threadIdx.x;
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
int main() {
    double hA[64], hB[64], hD[256], *dA, *dB, *dD;
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) hA[i * 4 + kk] = (i + 1) * 1.0 + 0.001 * kk;     // distinct rows
    for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = (kk == 0) ? (j + 1) * 100.0 : 0.0;   // only k = 0 contributes: D[i][j] = (i + 1) * (j + 1) * 100
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int ok_a = 1, ok_b = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const double v = hD[l * 4 + r];
        const int i_a = 4 * (l / 16) + r, i_b = (l / 16) + 4 * r, j = l % 16;
        if (v != (i_a + 1) * (j + 1) * 100.0) ok_a = 0;
        if (v != (i_b + 1) * (j + 1) * 100.0) ok_b = 0;
    }
    printf("D layout: i = 4 * (lane / 16) + r : %s ;  i = lane / 16 + 4 * r : %s\n", ok_a ? "YES" : "no", ok_b ? "YES" : "no");
    printf("lane 0: %.0f %.0f %.0f %.0f   lane 17: %.0f %.0f %.0f %.0f\n", hD[0], hD[1], hD[2], hD[3], hD[68], hD[69], hD[70], hD[71]);
    return 0;
}
