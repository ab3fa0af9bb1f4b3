// Device self-test of the DPP wave reductions (wave_ops.h) against the ds_bpermute butterflies they replace.
#include "knn_device.h"
#include "sg_common.h"
#include "wave_ops.h"

namespace {

__device__ inline unsigned int mix(unsigned int x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(64) void k_selftest_wave_ops(int* __restrict__ mismatches) {
    const int lane = threadIdx.x;
    int bad = 0;
    for (int round = 0; round < 64; ++round) {
        const unsigned int r = mix(blockIdx.x * 7919u + round * 64u + lane);
        // small integers: float / double sums are exact in any association order
        const float fv = (float)((int)(r & 0x3ff) - 512);
        const double dv = (double)((int)((r >> 10) & 0xfffff) - 500000);
        const float mv = __uint_as_float((r & 0x007fffffu) | 0x3f000000u) * ((r >> 31) ? -1.f : 1.f);
        const int iv = (int)(r >> 3) - (1 << 27);
        // values with MANY ties for the argmax rule
        float av = (float)((r >> 7) & 3);
        int ai = lane ^ (int)(round & 63);
        float sum_f = fv, max_f = mv, min_f = mv;
        double sum_d = dv;
        int max_i = iv;
        float arg_v = av;
        int arg_i = ai;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sum_f += __shfl_xor(sum_f, o);
            sum_d += __shfl_xor(sum_d, o);
            max_f = fmaxf(max_f, __shfl_xor(max_f, o));
            min_f = fminf(min_f, __shfl_xor(min_f, o));
            max_i = max(max_i, __shfl_xor(max_i, o));
            const float ov = __shfl_xor(arg_v, o);
            const int oi = __shfl_xor(arg_i, o);
            if (ov > arg_v || (ov == arg_v && oi < arg_i)) { arg_v = ov; arg_i = oi; }
        }
        sgw::wave_argmax(av, ai);
        bad += sgw::wave_sum(fv) != sum_f;
        bad += sgw::wave_sum(dv) != sum_d;
        bad += sgw::wave_max(mv) != max_f;
        bad += sgw::wave_min(mv) != min_f;
        bad += sgw::wave_max(iv) != max_i;
        bad += av != arg_v || ai != arg_i;
        bad += sgw::bcast(mv, round) != __shfl(mv, round);
        bad += sgw::bcast(iv, 63 - round) != __shfl(iv, 63 - round);
    }
    if (bad) atomicAdd(mismatches, bad);
}

// The top-K list in double form (knn_device.h: list_insert with v_min_f64 / v_max_f64) against the integer form it replaced
// (key_insert: one 64-bit compare and four selects per slot) and against list_merge12 (twelve keys at once): 64 lanes x 512 blocks, 300 keys each -- scores drawn from a handful of
// values (many exact ties, decided by the index), negative / zero / tiny positive scores, the empty key.
__global__ __launch_bounds__(64) void k_selftest_list_insert(int* __restrict__ mismatches) {
    using namespace sgknn;
    constexpr int K = 20;
    unsigned long long ki[K];
    double kd[K], km[K], b[12];
#pragma unroll
    for (int j = 0; j < K; ++j) { ki[j] = 0ull; kd[j] = list_empty(); km[j] = list_empty(); }
    int bad = 0;
    unsigned int r = mix(blockIdx.x * 64u + threadIdx.x + 12345u);
    int fill = (int)(r % 13u);                                          // keys of the coming batch of twelve that are real (per lane)
    for (int i = 0; i < 300; ++i) {
        r = mix(r + i);
        float sc;
        switch (r & 7) {
            case 0: sc = -(float)((r >> 8) & 7) * 0.125f; break;                 // eight values: ties
            case 1: sc = __uint_as_float(0x33000000u | ((r >> 9) & 0xffu)); break;    // tiny positive (the expanded score of a coincident pair)
            case 2: sc = -0.f; break;
            default: sc = -__uint_as_float(0x3a000000u + ((r >> 5) & 0x03ffffffu)); break;
        }
        const int idx = (int)((r >> 11) % (unsigned)kListMaxPoints);
        const unsigned long long key = ((r & 0x700u) == 0x700u || i % 12 >= fill) ? 0ull : make_key(sc, idx);
        key_insert<K>(ki, key);
        list_insert<K>(kd, key);
        // the list form built straight from (score, index) == the key's conversion; also for the index of a slab's padding lanes
        bad += __double_as_longlong(list_key(sc, idx)) != __double_as_longlong(to_list(make_key(sc, idx)));
        bad += __double_as_longlong(list_key(sc, 0x7fffffff)) != __double_as_longlong(to_list(make_key(sc, 0x7fffffff)));
        bad += __double_as_longlong(list_key(-INFINITY, idx)) != __double_as_longlong(to_list(make_key(-INFINITY, idx)));
        // ... and twelve at a time (list_merge12: the drain of the kNN kernels' append buffers), partly filled batches included
#pragma unroll
        for (int u = 0; u < 12; ++u)
            if (i % 12 == u) b[u] = to_list(key);
        if (i % 12 == 11) {
            list_merge12(km, b);
            fill = (int)(mix(r ^ 0x9e3779b9u) % 13u);
#pragma unroll
            for (int j = 0; j < K; ++j) bad += __double_as_longlong(km[j]) != __double_as_longlong(kd[j]);
        }
        if ((i & 15) == 15) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const bool empty = ki[j] == 0ull;
                bad += empty ? __double_as_longlong(kd[j]) != __double_as_longlong(list_empty())
                             : (from_list(kd[j]) != ki[j] || list_index(kd[j]) != key_index(ki[j]) || to_list(ki[j]) != kd[j]);
            }
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace

extern "C" int sg_selftest_list_insert(int* h_mismatches, void* stream) {
    SG_REQUIRE(h_mismatches, "sg_selftest_list_insert: null argument");
    int* d = nullptr;
    SG_HIP(hipMalloc((void**)&d, 4));
    hipStream_t st = sg::as_stream(stream);
    hipError_t e = hipMemsetAsync(d, 0, 4, st);
    if (e == hipSuccess) {
        k_selftest_list_insert<<<512, 64, 0, st>>>(d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_mismatches, d, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    SG_HIP(e);
    return SG_OK;
}

extern "C" int sg_selftest_wave_ops(int* h_mismatches, void* stream) {
    SG_REQUIRE(h_mismatches, "sg_selftest_wave_ops: null argument");
    int* d = nullptr;
    SG_HIP(hipMalloc((void**)&d, 4));
    hipStream_t st = sg::as_stream(stream);
    hipError_t e = hipMemsetAsync(d, 0, 4, st);
    if (e == hipSuccess) {
        k_selftest_wave_ops<<<512, 64, 0, st>>>(d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_mismatches, d, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    SG_HIP(e);
    return SG_OK;
}
