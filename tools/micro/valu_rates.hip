// Issue cost of the VALU instructions the kNN's sorted insertion is made of (wave64, one wave per SIMD, independent
// instructions): cycles per instruction = elapsed wall-clock cycles / instructions issued.
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o build_micro/valu_rates && build_micro/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define KERNEL(name, body)                                                                      \
    __global__ void name(unsigned long long* out, int iters) {                                  \
        unsigned long long a = threadIdx.x * 77ull + 1, b = threadIdx.x * 13ull + 5;            \
        unsigned int c = threadIdx.x, d = threadIdx.x * 3 + 1, r0 = 0, r1 = 1, r2 = 2, r3 = 3;  \
        float f = threadIdx.x, g = 1.5f, h0 = 0.f, h1 = 1.f;                                    \
        unsigned long long m;                                                                   \
        const long long t0 = clock64();                                                         \
        for (int i = 0; i < iters; ++i) { REP8(REP8(body)) }                                    \
        const long long t1 = clock64();                                                         \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned long long)(t1 - t0);        \
        out[1 + blockIdx.x * 64 + threadIdx.x] = a + b + c + d + r0 + r1 + r2 + r3 + (unsigned)f + (unsigned)h0 + (unsigned)h1 + m; \
    }
KERNEL(k_cmp_u64, asm volatile("v_cmp_gt_u64_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));)
KERNEL(k_cmp_u32, asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(c), "v"(d));)
KERNEL(k_cmp_f32, asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(f), "v"(g));)
KERNEL(k_cndmask, m = 0x5555555555555555ull; asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r0) : "v"(c), "v"(d), "s"(m));)
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(h0) : "v"(f), "v"(g)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(h1) : "v"(f), "v"(g));)
KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %1, %2" : "=v"(r1) : "v"(c), "v"(d));)
KERNEL(k_max_f32, asm volatile("v_max_f32 %0, %1, %2" : "=v"(h0) : "v"(f), "v"(g));)
KERNEL(k_pk_fma, asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a) : "v"(b));)
int main() {
    unsigned long long* d; hipMalloc(&d, (1 + 64 * 1024) * 8);
    struct { const char* n; void (*k)(unsigned long long*, int); int per; } ks[] = {
        {"v_cmp_gt_u64", k_cmp_u64, 64}, {"v_cmp_gt_u32", k_cmp_u32, 64}, {"v_cmp_gt_f32", k_cmp_f32, 64}, {"v_cndmask_b32", k_cndmask, 64},
        {"v_fma_f32 (x2 chains)", k_fma, 128}, {"v_add_u32", k_add_u32, 64}, {"v_max_f32", k_max_f32, 64}, {"v_pk_fma_f32", k_pk_fma, 64}};
    const int iters = 2000;
    for (auto& e : ks) {
        e.k<<<1, 64>>>(d, 10); hipDeviceSynchronize();
        e.k<<<1, 64>>>(d, iters); hipDeviceSynchronize();
        unsigned long long c; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
        printf("%-24s %.2f clock64 ticks per instruction (one wave alone)\n", e.n, (double)c / ((double)iters * e.per));
    }
    return 0;
}
