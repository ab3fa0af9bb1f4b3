// Sorted insertion of a 64-bit key into a descending list of K = 20 keys held in registers, two forms:
//   (a) mask form (csrc/knn_device.h, round 2): K v_cmp_gt_u64 into SGPR pairs + 4 v_cndmask per slot  (~100 VALU)
//   (b) f64 form: keys laid out as doubles in [2^52, 2^53) (exponent 0x433, 52 payload bits = ordered score << 20 | ~index), so that
//       integer order == double order and   new[j] = max(old[j], min(old[j-1], x))   is two v_{min,max}_f64 per slot  (~39 VALU)
// Reports shader cycles per insertion with 1 / 2 / 4 waves per SIMD and checks that both forms produce the same lists.
// hipcc --offload-arch=gfx950 -O3 tools/micro/key_insert_rates.hip -o build_micro/key_insert_rates && build_micro/key_insert_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int K = 20;
__device__ __forceinline__ unsigned int sel_mask(unsigned int if_clear, unsigned int if_set, unsigned long long mask) {
    unsigned int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}
__device__ inline void insert_mask(unsigned long long (&kv)[K], unsigned long long x) {
    unsigned long long c[K];
#pragma unroll
    for (int j = 0; j < K; ++j) c[j] = __builtin_amdgcn_uicmpl(x, kv[j], 34);
    const unsigned int xl = (unsigned int)x, xh = (unsigned int)(x >> 32);
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
        const unsigned int pl = (unsigned int)kv[j - 1], ph = (unsigned int)(kv[j - 1] >> 32);
        const unsigned int ol = (unsigned int)kv[j], oh = (unsigned int)(kv[j] >> 32);
        const unsigned int tl = sel_mask(xl, pl, c[j - 1]), th = sel_mask(xh, ph, c[j - 1]);
        kv[j] = ((unsigned long long)sel_mask(oh, th, c[j]) << 32) | sel_mask(ol, tl, c[j]);
    }
    kv[0] = ((unsigned long long)sel_mask((unsigned int)(kv[0] >> 32), xh, c[0]) << 32) | sel_mask((unsigned int)kv[0], xl, c[0]);
}
// plain instructions: __builtin_fmax / fmin would put a canonicalising v_max_f64 x, x in front of every operand (NaN quieting)
__device__ __forceinline__ double max64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double min64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline void insert_f64(double (&kv)[K], double x) {
#pragma unroll
    for (int j = K - 1; j > 0; --j) kv[j] = max64(kv[j], min64(kv[j - 1], x));
    kv[0] = max64(kv[0], x);
}
__device__ inline unsigned long long rnd_key(unsigned long long& s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return 0x4330000000000000ull | ((s >> 12) & 0x000fffffffffffffull);
}
template <int FORM>
__global__ void k_insert(unsigned long long* out, long long* cyc, int n) {
    unsigned long long seed = (blockIdx.x * 256ull + threadIdx.x) * 7919ull + 17;
    unsigned long long a[K];
    double d[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { a[j] = 0x4330000000000000ull; d[j] = __longlong_as_double(0x4330000000000000ll); }
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        const unsigned long long x = rnd_key(seed);
        if (FORM == 0) insert_mask(a, x);
        else insert_f64(d, __longlong_as_double((long long)x));
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
#pragma unroll
    for (int j = 0; j < K; ++j) out[((size_t)blockIdx.x * blockDim.x + threadIdx.x) * K + j] = FORM == 0 ? a[j] : (unsigned long long)__double_as_longlong(d[j]);
}
int main() {
    const int n = 4000;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int wps : {1, 2, 4}) {                        // waves per SIMD: blocks of 256 threads = one wave per SIMD of a CU
        const int blocks = cus * wps;
        const size_t cnt = (size_t)blocks * 256 * K;
        unsigned long long *o0, *o1; long long* c;
        hipMalloc(&o0, cnt * 8); hipMalloc(&o1, cnt * 8); hipMalloc(&c, 8);
        double cyc[2];
        for (int f = 0; f < 2; ++f) {
            for (int rep = 0; rep < 2; ++rep) {
                if (f == 0) k_insert<0><<<blocks, 256>>>(o0, c, n); else k_insert<1><<<blocks, 256>>>(o1, c, n);
                hipDeviceSynchronize();
            }
            long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
            cyc[f] = (double)h / n;
        }
        std::vector<unsigned long long> h0(cnt), h1(cnt);
        hipMemcpy(h0.data(), o0, cnt * 8, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), o1, cnt * 8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < cnt; ++i) bad += h0[i] != h1[i];
        bool sorted = true;
        for (size_t i = 0; i + 1 < K; ++i) sorted = sorted && h1[i] >= h1[i + 1];
        printf("%d wave(s)/SIMD: mask form %.1f cycles per insertion (incl. ~8 for the key), f64 form %.1f; lists differ in %zu of %zu slots, sorted %d\n",
               wps, cyc[0], cyc[1], bad, cnt, (int)sorted);
        hipFree(o0); hipFree(o1); hipFree(c);
    }
    return 0;
}
