// Does independent VALU work hide under v_mfma_f32_32x32x2_f32 (64 cycles in the matrix pipe)?  V fma per MFMA, 1/2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc0, acc1;
    for (int q = 0; q < 16; ++q) { acc0[q] = (float)threadIdx.x; acc1[q] = 1.f; }
    float v[16];
    for (int q = 0; q < 16; ++q) v[q] = (float)(threadIdx.x + q);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int w = 0; w < V; ++w) v[(u * V + w) & 15] = __builtin_fmaf(v[(u * V + w) & 15], a, b);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
#pragma unroll
            for (int w = 0; w < V; ++w) v[(u * V + w + 8) & 15] = __builtin_fmaf(v[(u * V + w + 8) & 15], a, b);
        }
    }
    float s = 0.f;
    for (int q = 0; q < 16; ++q) s += acc0[q] + acc1[q] + v[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
void run(int blocks_per_cu, int cus) {
    const int blocks = blocks_per_cu * cus, iters = 2000;
    float* out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<V><<<blocks, 256>>>(out, 10, 1e-3f, 1e-3f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<V><<<blocks, 256>>>(out, iters, 1e-3f, 1e-3f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 16 * 32 * 32 * 2 * 2;
    printf("VALU per MFMA %2d  waves/SIMD %d: %.3f ms  %.1f TFLOP/s (MFMA only)\n", V, blocks_per_cu, ms, flops / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int w = 1; w <= 2; ++w) {
        run<0>(w, cus); run<2>(w, cus); run<4>(w, cus); run<8>(w, cus); run<12>(w, cus); run<16>(w, cus); run<24>(w, cus);
    }
    return 0;
}
