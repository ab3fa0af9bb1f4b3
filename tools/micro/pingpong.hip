// Can the two waves of a SIMD overlap one's MFMA burst with the other's VALU phase when s_barrier keeps them in opposite phases?
// A workgroup of 8 waves (2 per SIMD); per trip every wave runs one MATRIX segment (24 x v_mfma_f32_32x32x16_f16, two accumulator chains) and one
// VALU segment (NV independent instructions), the segments separated by s_barrier.  Modes:
//   0  lockstep : every wave M, barrier, V, barrier               (no overlap possible: the M + V baseline)
//   1  ping-pong: waves 0-3 as above, waves 4-7 V, barrier, M, barrier   (one M wave beside one V wave on every SIMD)
//   2  free     : lockstep order without barriers
//   3  M only   4  V only
// hipcc --offload-arch=gfx950 -O3 -w tools/micro/pingpong.hip -o build_micro/pingpong && build_micro/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int NV8, int KIND>
__device__ __forceinline__ void valu_segment(float (&a)[8], float b, float c) {
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
        if (KIND == 0)
            asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));
        else   // the EdgeConv mix: conversions, packed multiplies, maxima
            asm volatile("v_cvt_pk_f16_f32 %0, %0, %8\nv_max_f32 %1, %1, %8\nv_cvt_f32_f16 %2, %2\nv_max_f32 %3, %3, %9\n"
                         "v_cvt_pk_f16_f32 %4, %4, %8\nv_max_f32 %5, %5, %8\nv_sub_f32 %6, %6, %9\nv_max_f32 %7, %7, %9"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));
    }
}
__device__ __forceinline__ void mfma_segment(f32x16& x, f32x16& y, f16x8 fa) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %2, %0\nv_mfma_f32_32x32x16_f16 %1, %2, %2, %1\nv_mfma_f32_32x32x16_f16 %0, %2, %2, %0\nv_mfma_f32_32x32x16_f16 %1, %2, %2, %1\n"
                     "v_mfma_f32_32x32x16_f16 %0, %2, %2, %0\nv_mfma_f32_32x32x16_f16 %1, %2, %2, %1\nv_mfma_f32_32x32x16_f16 %0, %2, %2, %0\nv_mfma_f32_32x32x16_f16 %1, %2, %2, %1"
                     : "+v"(x), "+v"(y) : "v"(fa));
}
template <int NV8, int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int* simd_of_wave) {
    float a[8] = {(float)threadIdx.x, 1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f};
    const float b = 1.0001f, c = 0.5f;
    f32x16 x = {0}, y = {0};
    const f16x8 fa = {1, 1, 1, 1, 1, 1, 1, 1};
    const int wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) simd_of_wave[wave] = (__builtin_amdgcn_s_getreg((2 << 11) | (4 << 6) | 4)) & 3;     // HW_ID bits 5:4 = simd_id
    const bool second = mode == 1 && wave >= 4;
    for (int i = 0; i < iters; ++i) {
        if (mode == 3) { mfma_segment(x, y, fa); continue; }
        if (mode == 4) { valu_segment<NV8, KIND>(a, b, c); continue; }
        if (!second) mfma_segment(x, y, fa); else valu_segment<NV8, KIND>(a, b, c);
        if (mode != 2) __builtin_amdgcn_s_barrier();
        if (!second) valu_segment<NV8, KIND>(a, b, c); else mfma_segment(x, y, fa);
        if (mode != 2) __builtin_amdgcn_s_barrier();
    }
    out[blockIdx.x * 512 + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7] + x[0] + y[1];
}
template <int NV8, int KIND>
void run(float* d, int* sw, int cus) {
    const int iters = 4000;
    const char* names[] = {"lockstep M|V", "ping-pong", "free-running", "M only", "V only"};
    double us[5];
    for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<NV8, KIND><<<cus, 512>>>(d, 10, mode, sw); hipDeviceSynchronize();
        hipEventRecord(e0); k<NV8, KIND><<<cus, 512>>>(d, iters, mode, sw); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        us[mode] = ms * 1e3 / iters;
    }
    printf("NV = %3d (%s): ", NV8 * 8, KIND ? "cvt / max / sub mix" : "v_fma_f32");
    for (int mode = 0; mode < 5; ++mode) printf("%s %.3f us  ", names[mode], us[mode]);
    printf("| ping-pong hides %.0f %% of the shorter segment\n", 100.0 * (us[0] - us[1]) / (us[3] < us[4] ? us[3] : us[4]));
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d; hipMalloc(&d, (size_t)cus * 512 * 4);
    int* sw; hipMalloc(&sw, 32);
    printf("per trip and wave: 24 MFMAs of 32x32x16 f16 + NV VALU; 8 waves per workgroup, one workgroup per CU (two waves per SIMD)\n");
    run<12, 0>(d, sw, cus); run<18, 0>(d, sw, cus); run<29, 0>(d, sw, cus);
    run<12, 1>(d, sw, cus); run<18, 1>(d, sw, cus); run<29, 1>(d, sw, cus);
    int h[8]; hipMemcpy(h, sw, 32, hipMemcpyDeviceToHost);
    printf("simd of waves 0..7 of block 0:"); for (int i = 0; i < 8; ++i) printf(" %d", h[i]); printf("\n");
    return 0;
}
