// The hand-scheduled EdgeConv slot loops (seggroup_amd/csrc/edgeconv_slots_gen.h) on their own: no weight staging, no base MFMAs, no
// statistics flush -- what does ONE tile's 20-slot loop cost, cache-hot (the same tile again and again) and with a fresh tile per trip?
//   hipcc --offload-arch=gfx950 -O3 -w -I seggroup_amd/csrc tools/micro/ec_slots_bench.hip -o build_micro/ec_slots_bench && build_micro/ec_slots_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "edgeconv_slots_gen.h"
#define SG_LDS __attribute__((address_space(3)))
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

__device__ __forceinline__ unsigned h2(float a, float b) {
    const _Float16 x = (_Float16)a, y = (_Float16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}

// MODE 2: S2X (one wave per SIMD), MODE 1: S1X plain statistics, MODE 3: S1X packed statistics
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_slots(const float* __restrict__ x9m, const int* __restrict__ knn, int ntiles, int trips, int fresh,
                                                               float* __restrict__ out, long long* __restrict__ cyc) {
    struct Img { u32x4 a1p[8][64]; float4 a1x[4][64]; u32x4 a2h[16][64]; };      // struct Lds of kernels_edgeconv.hip: a2h sits 12288 B behind a1p
    __shared__ Img img;
    auto& a2h = img.a2h;
    auto& a1p = img.a1p;
    __shared__ float4 bl[4][8][64];
    __shared__ int ids[4][20 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    for (int i = tid; i < 16 * 64; i += 256) { const float v = 0.01f * (float)((i * 37) % 101 - 50); (&a2h[0][0])[i] = u32x4{h2(v, -v), h2(v * 0.5f, v), h2(-v, v), h2(v, v * 0.25f)}; }
    for (int i = tid; i < 8 * 64; i += 256) { const float v = 0.02f * (float)((i * 53) % 89 - 44); (&a1p[0][0])[i] = u32x4{h2(v, -v), h2(v * 0.5f, v), h2(-v, v), h2(v, v * 0.25f)}; }
    for (int g = 0; g < 8; ++g) bl[wave][g][lane] = make_float4(0.1f * g, 0.01f * lane, -0.1f, 0.05f);
    __syncthreads();
    const float sd = 64.f;
    f32x16 tot = {0};
    long long loop_cycles = 0, loop_real = 0;
    const auto xp = (__attribute__((address_space(1))) const float*)x9m;
    for (int trip = 0; trip < trips; ++trip) {
        const int tile = ((blockIdx.x * 4 + wave) + (fresh ? trip * gridDim.x * 4 : 0)) % ntiles;
        const int pt = tile * 32 + (lane & 31);
        if (fresh || trip == 0) {
            const int* krow = knn + (size_t)pt * 20;
            for (int j = 0; j < 20; ++j) ids[wave][j * 64 + lane] = krow[j];
        }
        const float* xr = x9m + (size_t)pt * 12;
        const float xs0 = xr[4 * half] * sd, xs1 = xr[4 * half + 1] * sd, xs2 = xr[4 * half + 2] * sd, xs3 = xr[4 * half + 3] * sd, xs4 = xr[8] * sd;
        f32x16 ss0, ss1, sq0, sq1, bb0, bb1;
        const unsigned a_ids = (unsigned)(size_t)(SG_LDS const int*)(&ids[wave][lane]);
        const unsigned l16 = 16u * (unsigned)half;
        const long long t0 = __builtin_readcyclecounter();
        const long long r0 = __builtin_amdgcn_s_memrealtime();
        if constexpr (MODE == 2 || MODE == 4) {
            const unsigned a_base = (unsigned)(size_t)(SG_LDS const float4*)(&bl[wave][0][lane]);
            const unsigned a_frag = (unsigned)(size_t)(SG_LDS const u32x4*)(&a1p[0][lane]);
            const auto kp = (__attribute__((address_space(1))) const int*)knn;
            const unsigned koff = (unsigned)pt * 80u;
            const unsigned long long c02 = 0x3e4ccccd3e4ccccdull;             // 0.2f twice: v_pk_mul_f32's constant pair
#define SG_S2X_STATEMENT(prefix) \
            asm volatile(prefix SG_EC_S2X_SLOTS \
                         : "=&" SG_EC_S2X_STAT_S0(ss0), "=&" SG_EC_S2X_STAT_S1(ss1), "=&" SG_EC_S2X_STAT_Q0(sq0), "=&" SG_EC_S2X_STAT_Q1(sq1), \
                           "=&" SG_EC_S2X_BEST0(bb0), "=&" SG_EC_S2X_BEST1(bb1) \
                         : [c02] "s"(c02), [x9m] "s"(xp), [knn] "s"(kp), [sd] "s"(sd), [koff] "v"(koff), [l16] "v"(l16), [base] "v"(a_base), [frag] "v"(a_frag), \
                           [xs0] "v"(xs0), [xs1] "v"(xs1), [xs2] "v"(xs2), [xs3] "v"(xs3), [xs4] "v"(xs4) \
                         : "memory", SG_EC_S2X_SLOTS_CLOBBERS)
            // MODE 4: two textual copies of the statement taken in turn -- twice the code (2 x 52 KB against a 64 KB instruction cache per two CUs)
            if (MODE == 4 && (trip & 1)) SG_S2X_STATEMENT("; second copy\n\t");
            else SG_S2X_STATEMENT("");
        } else {
            f32x16 base0, base1;
            for (int q = 0; q < 16; ++q) { base0[q] = 0.01f * q + 0.001f * lane; base1[q] = -0.02f * q; }
            const u32x4 f0 = a1p[0][lane], f1 = a1p[1][lane], f2 = a1p[2][lane], f3 = a1p[3][lane];
            if constexpr (MODE == 1)
                asm volatile(SG_EC_S1X_SLOTS
                             : "=&" SG_EC_S1X_STAT_S0(ss0), "=&" SG_EC_S1X_STAT_S1(ss1), "=&" SG_EC_S1X_STAT_Q0(sq0), "=&" SG_EC_S1X_STAT_Q1(sq1),
                               "=&" SG_EC_S1X_BEST0(bb0), "=&" SG_EC_S1X_BEST1(bb1)
                             : [x9m] "s"(xp), [sd] "s"(sd), [knn] "s"((__attribute__((address_space(1))) const int*)knn), [koff] "v"((unsigned)pt * 80u), [l16] "v"(l16), [base] "v"((unsigned)(size_t)(SG_LDS const float4*)(&bl[wave][0][lane])), [frag] "v"((unsigned)(size_t)(SG_LDS const u32x4*)(&a1p[0][lane])),
                               [xs0] "v"(xs0), [xs1] "v"(xs1), [xs2] "v"(xs2), [xs3] "v"(xs3), [xs4] "v"(xs4)
                             : "memory", SG_EC_S1X_SLOTS_CLOBBERS);
            else
                asm volatile(SG_EC_S1X_SLOTS_PK
                             : "=&" SG_EC_S1X_STAT_S0(ss0), "=&" SG_EC_S1X_STAT_S1(ss1), "=&" SG_EC_S1X_STAT_Q0(sq0), "=&" SG_EC_S1X_STAT_Q1(sq1),
                               "=&" SG_EC_S1X_BEST0(bb0), "=&" SG_EC_S1X_BEST1(bb1)
                             : [x9m] "s"(xp), [sd] "s"(sd), [knn] "s"((__attribute__((address_space(1))) const int*)knn), [koff] "v"((unsigned)pt * 80u), [l16] "v"(l16), [base] "v"((unsigned)(size_t)(SG_LDS const float4*)(&bl[wave][0][lane])), [frag] "v"((unsigned)(size_t)(SG_LDS const u32x4*)(&a1p[0][lane])),
                               [xs0] "v"(xs0), [xs1] "v"(xs1), [xs2] "v"(xs2), [xs3] "v"(xs3), [xs4] "v"(xs4)
                             : "memory", SG_EC_S1X_SLOTS_PK_CLOBBERS);
        }
        loop_cycles += __builtin_readcyclecounter() - t0;
        loop_real += __builtin_amdgcn_s_memrealtime() - r0;
        tot += ss0 + ss1 + sq0 + sq1 + bb0 + bb1;
    }
    float s = 0;
    for (int q = 0; q < 16; ++q) s += tot[q];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) { cyc[blockIdx.x * 4 + wave] = loop_cycles; cyc[8192 + blockIdx.x * 4 + wave] = loop_real; }
}

template <int MODE>
void run(const char* name, const float* dx, const int* dk, int ntiles, float* dout, long long* dcyc, int blocks, int fresh) {
    const int trips = 40;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_slots<MODE><<<blocks, 256>>>(dx, dk, ntiles, 2, fresh, dout, dcyc); hipDeviceSynchronize();
    hipEventRecord(e0); k_slots<MODE><<<blocks, 256>>>(dx, dk, ntiles, trips, fresh, dout, dcyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(blocks * 4), rr(blocks * 4);
    hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(rr.data(), dcyc + 8192, rr.size() * 8, hipMemcpyDeviceToHost);
    double cs = 0; for (auto v : c) cs += (double)v;
    double rs = 0; for (auto v : rr) rs += (double)v;
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    printf("%-26s %-10s blocks %4d: %8.1f ns per slot and wave; x %.2f waves/SIMD -> %7.1f ns per slot and SIMD; inside the asm: %7.0f cycles (s_memtime), %7.1f ns (s_memrealtime, 100 MHz) per slot and wave\n",
           name, fresh ? "fresh tile" : "same tile", blocks, ms * 1e6 / (trips * 20.0), waves_per_simd, ms * 1e6 / (trips * 20.0) / waves_per_simd,
           cs / c.size() / (trips * 20.0), rs / rr.size() / (trips * 20.0) * 10.0);
}

int main() {
    const int N = 150000, ntiles = N / 32;
    std::vector<float> x((size_t)N * 12);
    std::vector<int> k((size_t)N * 20);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (auto& v : x) v = (float)(rnd() % 2001) / 1000.f - 1.f;
    for (int i = 0; i < N; ++i) for (int j = 0; j < 20; ++j) k[(size_t)i * 20 + j] = std::min(N - 1, (i / 150) * 150 + (int)(rnd() % 150));
    float* dx; int* dk; float* dout; long long* dcyc;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dk, k.size() * 4); hipMalloc(&dout, 4096 * 256 * 4); hipMalloc(&dcyc, 2 * 8192 * 8);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dk, k.data(), k.size() * 4, hipMemcpyHostToDevice);
    for (int fresh = 0; fresh < 2; ++fresh) {
        run<2>("S2X (MLP3) 2 waves/SIMD", dx, dk, ntiles, dout, dcyc, 512, fresh);
        run<2>("S2X (MLP3) 1 wave/SIMD", dx, dk, ntiles, dout, dcyc, 256, fresh);
        run<4>("S2X two copies 2 w/SIMD", dx, dk, ntiles, dout, dcyc, 512, fresh);
        run<4>("S2X two copies 1 w/SIMD", dx, dk, ntiles, dout, dcyc, 256, fresh);
        run<1>("S1X plain  2 waves/SIMD", dx, dk, ntiles, dout, dcyc, 512, fresh);
        run<3>("S1X packed 2 waves/SIMD", dx, dk, ntiles, dout, dcyc, 512, fresh);
        run<1>("S1X plain  1 wave/SIMD", dx, dk, ntiles, dout, dcyc, 256, fresh);
        run<3>("S1X packed 1 wave/SIMD", dx, dk, ntiles, dout, dcyc, 256, fresh);
    }
    return 0;
}
