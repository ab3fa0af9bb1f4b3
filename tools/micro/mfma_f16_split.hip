// v_mfma_f32_32x32x16_f16 for the fp16x2-split conv2 of EdgeConv (kernels_edgeconv.hip): does the matrix pipe keep fp16
// SUBNORMAL inputs (the low piece of a small value is one), what does the two-piece split cost in accuracy against the
// bf16x3 split and a float64 dot product, and the rate.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f16_split.hip -o gpurun_out/mfma_f16_split && gpurun_out/mfma_f16_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void k_subnormal(float* out) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f /* 2^-20 */; b[j] = (_Float16)1.0f; }
    f32x16 c;
    for (int q = 0; q < 16; ++q) c[q] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)(_Float16)(3.0e-6f); }
}

// D[m][n] = sum_k A[m][k] B[k][n], K = 64, operands fp32, three ways
__global__ void k_dot(const float* A, const float* B, float* D16, float* Dbf) {   // A [32][64], B [64][32]
    const int l = threadIdx.x, m = l & 31, h = l >> 5;
    f32x16 c, d;
    for (int q = 0; q < 16; ++q) { c[q] = 0.f; d[q] = 0.f; }
    for (int kb = 0; kb < 4; ++kb) {
        f16x8 a1, a2, b1, b2;
        union { bf16x8 v; unsigned short s[8]; } p1, p2, p3, q1, q2, q3;
        for (int j = 0; j < 8; j += 2) {
            const f32x2 av = {A[m * 64 + 16 * kb + 8 * h + j], A[m * 64 + 16 * kb + 8 * h + j + 1]};
            const f32x2 bv = {B[(16 * kb + 8 * h + j) * 32 + m], B[(16 * kb + 8 * h + j + 1) * 32 + m]};
            const f16x2 ah = __builtin_convertvector(av, f16x2), bh = __builtin_convertvector(bv, f16x2);
            const f16x2 al = __builtin_convertvector(av - __builtin_convertvector(ah, f32x2), f16x2);
            const f16x2 bl = __builtin_convertvector(bv - __builtin_convertvector(bh, f32x2), f16x2);
            a1[j] = ah[0]; a1[j + 1] = ah[1]; a2[j] = al[0]; a2[j + 1] = al[1];
            b1[j] = bh[0]; b1[j + 1] = bh[1]; b2[j] = bl[0]; b2[j + 1] = bl[1];
        }
        for (int j = 0; j < 8; ++j) {
            float v = A[m * 64 + 16 * kb + 8 * h + j];
            float x = __uint_as_float(__float_as_uint(v) & 0xffff0000u); float r = v - x;
            float y = __uint_as_float(__float_as_uint(r) & 0xffff0000u); float z = r - y;
            p1.s[j] = __float_as_uint(x) >> 16; p2.s[j] = __float_as_uint(y) >> 16; p3.s[j] = __float_as_uint(z) >> 16;
            v = B[(16 * kb + 8 * h + j) * 32 + m];
            x = __uint_as_float(__float_as_uint(v) & 0xffff0000u); r = v - x;
            y = __uint_as_float(__float_as_uint(r) & 0xffff0000u); z = r - y;
            q1.s[j] = __float_as_uint(x) >> 16; q2.s[j] = __float_as_uint(y) >> 16; q3.s[j] = __float_as_uint(z) >> 16;
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p3.v, q1.v, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p2.v, q2.v, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1.v, q3.v, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p2.v, q1.v, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1.v, q2.v, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1.v, q1.v, d, 0, 0, 0);
    }
    for (int q = 0; q < 16; ++q) {
        D16[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + m] = c[q];
        Dbf[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + m] = d[q];
    }
}

__global__ void k_rate(float* out, int iters) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)1.0f; b[j] = (_Float16)(0.001f * threadIdx.x); }
    f32x16 c[4];
    for (int t = 0; t < 4; ++t) for (int q = 0; q < 16; ++q) c[t][q] = 0.f;
    for (int i = 0; i < iters; ++i)
        for (int t = 0; t < 4; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[t], 0, 0, 0);
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int q = 0; q < 16; ++q) s += c[t][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* dS; hipMalloc(&dS, 16);
    k_subnormal<<<1, 64>>>(dS);
    float s[2]; hipMemcpy(s, dS, 8, hipMemcpyDeviceToHost);
    printf("subnormal A (2^-20) x 1.0, K = 16: %.9g (kept: %.9g, flushed: 0); cvt(3.0e-6) = %.9g\n", s[0], 16 * 9.5367431640625e-07, s[1]);
    for (int scale = 0; scale < 3; ++scale) {
        const float amp = scale == 0 ? 1.f : scale == 1 ? 0.01f : 30.f;
        std::vector<float> A(32 * 64), B(64 * 32), D16(1024), Dbf(1024);
        srand(7 + scale);
        for (auto& v : A) v = 0.2f * ((float)rand() / RAND_MAX - 0.5f);                    // weights
        for (auto& v : B) { v = amp * ((float)rand() / RAND_MAX - 0.3f); if (v < 0) v *= 0.2f; }   // LeakyReLU-shaped activations
        float *dA, *dB, *d1, *d2;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d1, 4096); hipMalloc(&d2, 4096);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        k_dot<<<1, 64>>>(dA, dB, d1, d2);
        hipMemcpy(D16.data(), d1, 4096, hipMemcpyDeviceToHost); hipMemcpy(Dbf.data(), d2, 4096, hipMemcpyDeviceToHost);
        double e16 = 0, ebf = 0, e32 = 0, mag = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double r = 0; float f = 0.f;
            for (int k = 0; k < 64; ++k) { r += (double)A[i * 64 + k] * B[k * 32 + j]; f = fmaf(A[i * 64 + k], B[k * 32 + j], f); }
            e16 = fmax(e16, fabs(D16[i * 32 + j] - r)); ebf = fmax(ebf, fabs(Dbf[i * 32 + j] - r)); e32 = fmax(e32, fabs((double)f - r));
            mag = fmax(mag, fabs(r));
        }
        printf("amp %-5g max|y| %.3g: max abs error  fp16x2 (3 products) %.3g | bf16x3 (6 products) %.3g | fp32 fmaf chain %.3g\n", amp, mag, e16, ebf, e32);
    }
    float* dO; hipMalloc(&dO, 256 * 8 * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_rate<<<256 * 8, 256>>>(dO, 10);
    hipEventRecord(e0); k_rate<<<256 * 8, 256>>>(dO, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 32 * 32 * 16 * 4.0 * iters * (256.0 * 8 * 4);
    printf("v_mfma_f32_32x32x16_f16: %.1f TFLOP/s (%.3f ms)\n", flop / ms / 1e9, ms);
    return 0;
}
