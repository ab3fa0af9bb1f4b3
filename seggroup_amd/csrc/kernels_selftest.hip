// Device self-test of the DPP wave reductions (wave_ops.h) against the ds_bpermute butterflies they replace.
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

}  // namespace

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
