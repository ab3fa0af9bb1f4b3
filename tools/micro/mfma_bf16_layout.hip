// Layout check + rate of v_mfma_f32_32x32x16_bf16 as used by the bf16x3-split conv2 of EdgeConv (kernels_edgeconv.hip):
//   A fragment: lane l holds A[m = l & 31][k = 8 * (l >> 5) + j], j = 0..7   (8 bf16 = 4 dwords)
//   B fragment: lane l holds B[k = 8 * (l >> 5) + j][n = l & 31]
//   D: lane l, reg q -> D[m = (q & 3) + 8 * (q >> 2) + 4 * (l >> 5)][n = l & 31]
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_bf16_layout.hip -o gpurun_out/mfma_bf16_layout && gpurun_out/mfma_bf16_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ inline unsigned short bf16_bits(float x) { return (unsigned short)(__float_as_uint(x) >> 16); }

__global__ void k_check(const float* A, const float* B, float* D) {   // A [32][16], B [16][32], D [32][32]
    const int l = threadIdx.x, m = l & 31, h = l >> 5;
    union { bf16x8 v; unsigned short s[8]; } a, b;
    for (int j = 0; j < 8; ++j) { a.s[j] = bf16_bits(A[m * 16 + 8 * h + j]); b.s[j] = bf16_bits(B[(8 * h + j) * 32 + m]); }
    f32x16 c;
    for (int q = 0; q < 16; ++q) c[q] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
    for (int q = 0; q < 16; ++q) D[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + m] = c[q];
}

__global__ void k_rate(float* out, int iters) {
    union { bf16x8 v; unsigned short s[8]; } a, b;
    for (int j = 0; j < 8; ++j) { a.s[j] = 0x3f80; b.s[j] = (unsigned short)(0x3c00 + threadIdx.x); }
    f32x16 c[4];
    for (int t = 0; t < 4; ++t) for (int q = 0; q < 16; ++q) c[t][q] = 0.f;
    for (int i = 0; i < iters; ++i)
        for (int t = 0; t < 4; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c[t], 0, 0, 0);
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int q = 0; q < 16; ++q) s += c[t][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), R(32 * 32, 0.f);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 7 + k * 3) % 13 - 6);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 5 + j * 11) % 17 - 8);     // asymmetric
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += A[i * 16 + k] * B[k * 32 + j];
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k_check<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += D[i] != R[i];
    printf("layout check: %d of 1024 elements differ\n", bad);
    float* dO; hipMalloc(&dO, 256 * 8 * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_rate<<<256 * 8, 256>>>(dO, 10);
    hipEventRecord(e0); k_rate<<<256 * 8, 256>>>(dO, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 32 * 32 * 16 * 4.0 * iters * (256.0 * 8 * 4);
    printf("v_mfma_f32_32x32x16_bf16: %.1f TFLOP/s (%.3f ms)\n", flop / ms / 1e9, ms);
    return bad != 0;
}
