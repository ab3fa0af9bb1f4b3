// Practical ceiling of v_mfma_f32_32x32x2_f32 on this GPU: nothing but independent / dependent MFMA chains.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_peak.hip -o gpurun_out/mfma_f32_peak && ./gpurun_out/mfma_f32_peak
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int q = 0; q < 16; ++q) acc[c][q] = (float)(threadIdx.x + c);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int q = 0; q < 16; ++q) s += acc[c][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
void run(int blocks_per_cu, int cus) {
    const int blocks = blocks_per_cu * cus, iters = 4000;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS><<<blocks, 256>>>(out, 10, 1e-3f, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CHAINS><<<blocks, 256>>>(out, iters, 1e-3f, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 8 * CHAINS * 32 * 32 * 2 * 2;
    printf("chains %d  blocks/CU %d (waves/SIMD %d): %.3f ms  %.1f TFLOP/s\n", CHAINS, blocks_per_cu, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s  CUs %d  clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    run<1>(1, p.multiProcessorCount);
    run<1>(2, p.multiProcessorCount);
    run<2>(1, p.multiProcessorCount);
    run<2>(2, p.multiProcessorCount);
    run<4>(2, p.multiProcessorCount);
    run<2>(4, p.multiProcessorCount);
    return 0;
}
