// THROUGHPUT of single instructions with the whole chip busy: W waves per SIMD (W = 1, 2, 4, 8), every wave a long stream of independent
// instructions (8 register chains), wall time by HIP events -> wave-instructions per SIMD-cycle at the measured clock.  Answers what a
// "VALU roof" is on gfx950: how many cycles of a SIMD one wave64 instruction costs once enough waves interleave, per instruction class,
// and whether MFMA and VALU streams of different waves overlap on one SIMD.
// hipcc --offload-arch=gfx950 -O3 -w tools/micro/issue_rates.hip -o build_micro/issue_rates && build_micro/issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
#define R8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, int mixed_mfma) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f, b = 1.0001f, c = 0.5f;
    double d0 = threadIdx.x, d1 = 1., d2 = 2., d3 = 3., d4 = 4., d5 = 5., d6 = 6., d7 = 7., e = 2.5;
    unsigned long long p0 = threadIdx.x, p1 = 1, p2 = 2, p3 = 3, p4 = 4, p5 = 5, p6 = 6, p7 = 7;
    f32x16 acc = {0}, acc_b = {0};
    f16x8 fa = {1, 1, 1, 1, 1, 1, 1, 1};
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(1, 2, 3, 4);
    __syncthreads();
    float4 l0, l1, l2, l3;
    const unsigned int la = threadIdx.x * 16;
    const bool do_mfma = mixed_mfma && ((threadIdx.x >> 6) + blockIdx.x) % 2 == 0;      // mixed mode: half of the waves run MFMAs only
    for (int i = 0; i < iters; ++i) {
        if (KIND == 9 || do_mfma) {
            // two independent accumulator chains, kept alive by asm: eight MFMAs per trip
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %2, %0\nv_mfma_f32_32x32x16_f16 %1, %2, %2, %1\nv_mfma_f32_32x32x16_f16 %0, %2, %2, %0\nv_mfma_f32_32x32x16_f16 %1, %2, %2, %1\n"
                         "v_mfma_f32_32x32x16_f16 %0, %2, %2, %0\nv_mfma_f32_32x32x16_f16 %1, %2, %2, %1\nv_mfma_f32_32x32x16_f16 %0, %2, %2, %0\nv_mfma_f32_32x32x16_f16 %1, %2, %2, %1"
                         : "+v"(acc), "+v"(acc_b) : "v"(fa));
            continue;
        }
        if (KIND == 0) { asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); }
        if (KIND == 1) { asm volatile("v_pk_fma_f32 %0, %0, %8, %8\nv_pk_fma_f32 %1, %1, %8, %8\nv_pk_fma_f32 %2, %2, %8, %8\nv_pk_fma_f32 %3, %3, %8, %8\nv_pk_fma_f32 %4, %4, %8, %8\nv_pk_fma_f32 %5, %5, %8, %8\nv_pk_fma_f32 %6, %6, %8, %8\nv_pk_fma_f32 %7, %7, %8, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(e)); }
        if (KIND == 2) { asm volatile("v_max_f64 %0, %0, %8\nv_max_f64 %1, %1, %8\nv_max_f64 %2, %2, %8\nv_max_f64 %3, %3, %8\nv_max_f64 %4, %4, %8\nv_max_f64 %5, %5, %8\nv_max_f64 %6, %6, %8\nv_max_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(e)); }
        if (KIND == 3) { asm volatile("v_max_f32 %0, %0, %8\nv_max_f32 %1, %1, %8\nv_max_f32 %2, %2, %8\nv_max_f32 %3, %3, %8\nv_max_f32 %4, %4, %8\nv_max_f32 %5, %5, %8\nv_max_f32 %6, %6, %8\nv_max_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
        if (KIND == 4) { asm volatile("v_cvt_pk_f16_f32 %0, %0, %8\nv_cvt_pk_f16_f32 %1, %1, %8\nv_cvt_pk_f16_f32 %2, %2, %8\nv_cvt_pk_f16_f32 %3, %3, %8\nv_cvt_pk_f16_f32 %4, %4, %8\nv_cvt_pk_f16_f32 %5, %5, %8\nv_cvt_pk_f16_f32 %6, %6, %8\nv_cvt_pk_f16_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
        if (KIND == 5) { asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc"); }
        if (KIND == 6) { asm volatile("ds_read_b128 %0, %4\nds_read_b128 %1, %4 offset:16\nds_read_b128 %2, %4 offset:32\nds_read_b128 %3, %4 offset:48\nds_read_b128 %0, %4 offset:64\nds_read_b128 %1, %4 offset:80\nds_read_b128 %2, %4 offset:96\nds_read_b128 %3, %4 offset:112\ns_waitcnt lgkmcnt(0)" : "=v"(l0), "=v"(l1), "=v"(l2), "=v"(l3) : "v"(la)); }
        if (KIND == 7) { asm volatile("ds_max_f32 %0, %1\nds_max_f32 %0, %2 offset:4\nds_max_f32 %0, %3 offset:8\nds_max_f32 %0, %4 offset:12\nds_max_f32 %0, %1 offset:1024\nds_max_f32 %0, %2 offset:1028\nds_max_f32 %0, %3 offset:1032\nds_max_f32 %0, %4 offset:1036" :: "v"(la), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory"); }
        if (KIND == 8) { asm volatile("v_fma_mixlo_f16 %0, %8, -1.0, %9 op_sel_hi:[1,0,0]\nv_fma_mixlo_f16 %1, %8, -1.0, %9 op_sel_hi:[1,0,0]\nv_fma_mixlo_f16 %2, %8, -1.0, %9 op_sel_hi:[1,0,0]\nv_fma_mixlo_f16 %3, %8, -1.0, %9 op_sel_hi:[1,0,0]\nv_fma_mixhi_f16 %4, %8, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\nv_fma_mixhi_f16 %5, %8, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\nv_fma_mixhi_f16 %6, %8, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\nv_fma_mixhi_f16 %7, %8, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float)(p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7) +
                                          acc[0] + acc_b[1] + l0.x + l1.x + l2.x + l3.x;
}
template <int KIND>
double run(int wps, int iters, int mixed, float* d, int cus) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<cus * wps, 256>>>(d, 10, mixed); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<cus * wps, 256>>>(d, iters, mixed); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d; hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
    const int iters = 20000;
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_max_f64", "v_max_f32", "v_cvt_pk_f16_f32", "v_cndmask_b32", "ds_read_b128", "ds_max_f32", "v_fma_mix{lo,hi}_f16", "v_mfma_f32_32x32x16_f16"};
    printf("%d CUs, clock %.2f GHz nominal; ns per wave-instruction per SIMD at W waves per SIMD (x GHz = SIMD cycles per instruction)\n", cus, p.clockRate / 1e6);
    for (int kind = 0; kind < 10; ++kind) {
        printf("%-26s", names[kind]);
        for (int wps : {1, 2, 4, 8}) {
            double ms = 0;
            switch (kind) { case 0: ms = run<0>(wps, iters, 0, d, cus); break; case 1: ms = run<1>(wps, iters, 0, d, cus); break; case 2: ms = run<2>(wps, iters, 0, d, cus); break;
                            case 3: ms = run<3>(wps, iters, 0, d, cus); break; case 4: ms = run<4>(wps, iters, 0, d, cus); break; case 5: ms = run<5>(wps, iters, 0, d, cus); break;
                            case 6: ms = run<6>(wps, iters, 0, d, cus); break; case 7: ms = run<7>(wps, iters, 0, d, cus); break; case 8: ms = run<8>(wps, iters, 0, d, cus); break;
                            case 9: ms = run<9>(wps, iters, 0, d, cus); break; }
            // per SIMD: wps waves x iters x 8 instructions
            printf("  W=%d: %6.2f ns", wps, ms * 1e6 / ((double)wps * iters * 8));
        }
        printf("\n");
    }
    // mixed: on every SIMD half of the waves run MFMAs only, the other half v_fma_f32 only: wall time vs the two alone
    for (int wps : {2, 4}) {
        const double m = run<0>(wps, iters, 1, d, cus), a = run<0>(wps / 2, iters, 0, d, cus), b = run<9>(wps / 2, iters, 0, d, cus);
        printf("mixed W=%d (half MFMA waves, half v_fma waves): %.3f ms; v_fma waves alone %.3f ms, MFMA waves alone %.3f ms\n", wps, m, a, b);
    }
    return 0;
}
