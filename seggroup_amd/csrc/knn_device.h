// Device helpers shared by the in-cluster kNN kernels (kernels_knn.hip, kernels_knn_sorted.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace sgknn {

// knn() score in the reference's fp32 operation order (model.py:31-33); this code is compiled
// with -ffp-contract=off, the two FMAs are the ones MKL's K=3 dot product performs
__device__ inline float score4(const float4& me, const float4& p) {
    const float tt = __builtin_fmaf(me.z, p.z, __builtin_fmaf(me.y, p.y, me.x * p.x));
    const float inner = -2.0f * tt;
    return ((-p.w) - inner) - me.w;
}

// 64-bit key: order-preserving uint of the fp32 score << 32 | ~member index.  One unsigned compare implements
// the total order (score descending, index ascending), independent of arrival order.
__device__ inline unsigned long long make_key(float score, int idx) {
    const unsigned int u = __float_as_uint(score);
    const unsigned int o = u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
    return ((unsigned long long)o << 32) | (unsigned int)(0xffffffffu - (unsigned int)idx);
}
__device__ inline int key_index(unsigned long long key) { return (int)(0xffffffffu - (unsigned int)(key & 0xffffffffu)); }

// Sorted insertion in PARALLEL form: every slot decides independently from two compares (no K-step chain):
// new[j] = x > old[j] ? (x > old[j-1] ? old[j-1] : x) : old[j]
// The K compares are made ONCE, as wave masks (an SGPR pair each), and the selects take the masks as they are.  Written with
// plain bools the compiler keeps one VCC: it re-evaluates `x > old[j-1]` as a second 64-bit compare per slot and pays the
// VALU-writes-VCC -> v_cndmask wait states behind every one of them (2 compares + 2 s_nop + 4 selects per slot).
__device__ __forceinline__ unsigned int sel_mask(unsigned int if_clear, unsigned int if_set, unsigned long long mask) {
    unsigned int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}
template <int K>
__device__ inline void key_insert(unsigned long long (&kv)[K], unsigned long long x) {
    unsigned long long c[K];
#pragma unroll
    for (int j = 0; j < K; ++j) c[j] = __builtin_amdgcn_uicmpl(x, kv[j], 34 /* ICMP_UGT */);
    const unsigned int xl = (unsigned int)x, xh = (unsigned int)(x >> 32);
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
        const unsigned int pl = (unsigned int)kv[j - 1], ph = (unsigned int)(kv[j - 1] >> 32);
        const unsigned int ol = (unsigned int)kv[j], oh = (unsigned int)(kv[j] >> 32);
        const unsigned int tl = sel_mask(xl, pl, c[j - 1]), th = sel_mask(xh, ph, c[j - 1]);      // min(old[j-1], x)
        kv[j] = ((unsigned long long)sel_mask(oh, th, c[j]) << 32) | sel_mask(ol, tl, c[j]);
    }
    kv[0] = ((unsigned long long)sel_mask((unsigned int)(kv[0] >> 32), xh, c[0]) << 32) | sel_mask((unsigned int)kv[0], xl, c[0]);
}

// LIST form of a key (round 3): the same key as a double in [2^52, 2^53) -- exponent 0x433, payload = ordered score << 20 | the low
// 20 bits of ~index -- so that integer order == floating-point order and a sorted insertion is two plain instructions per slot,
//     new[j] = max(old[j], min(old[j-1], x)),
// instead of a 64-bit compare and four selects (tools/micro/key_insert_rates.hip: 248 vs 560 cycles per insertion into 20 slots,
// identical lists).  All keys share one exponent: no NaN, no denormal, no -0 can appear.  20 index bits: member positions inside one
// cluster, i.e. N <= 2^20 points per scene (the host entry points check it).  Key 0 ("nothing yet") <-> kListEmpty, the smallest.
constexpr int kListIndexBits = 20;
constexpr int kListMaxPoints = 1 << kListIndexBits;      // == SG_MAX_POINTS (include/seggroup_hip.h), checked in pipeline.cpp
__device__ __forceinline__ double list_empty() { return __hiloint2double(0x43300000, 0); }
__device__ __forceinline__ double to_list(unsigned long long key) {
    const unsigned int o = (unsigned int)(key >> 32), lo = (unsigned int)key;
    return __hiloint2double((int)(0x43300000u | (o >> 12)), (int)((o << 20) | (lo & 0xfffffu)));
}
__device__ __forceinline__ unsigned long long from_list(double d) {              // exact inverse for indices < 2^20 (empty -> a key below every real one)
    const unsigned int hi = (unsigned int)__double2hiint(d), lo = (unsigned int)__double2loint(d);
    return ((unsigned long long)((hi << 12) | (lo >> 20)) << 32) | (0xfff00000u | lo);
}
// the list form built straight from (score, index): == to_list(make_key(score, idx)), bit for bit (kernels_selftest.hip checks it); an index
// beyond 20 bits (the 0x7fffffff of a slab's padding lanes) keeps its low 20 bits, as to_list does
__device__ __forceinline__ double list_key(float score, int idx) {
    const unsigned int u = __float_as_uint(score);
    const unsigned int o = u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
    return __hiloint2double((int)(0x43300000u | (o >> 12)), (int)((o << 20) | (~(unsigned int)idx & 0xfffffu)));
}
// the order-preserving score bits of a list value (what key >> 32 is for a key): one v_alignbit_b32
__device__ __forceinline__ unsigned int list_score(double d) {
    return __builtin_amdgcn_alignbit((unsigned int)__double2hiint(d), (unsigned int)__double2loint(d), 20);
}
// the list value of "any key whose score bits reach o" minus one unit: k > list_floor(o) <=> the score bits of k are >= o
__device__ __forceinline__ double list_floor(unsigned int o) {
    return __longlong_as_double(__double_as_longlong(__hiloint2double((int)(0x43300000u | (o >> 12)), (int)(o << 20))) - 1ll);
}
__device__ __forceinline__ int list_index(double d) { return (int)(0xfffffu - ((unsigned int)__double2loint(d) & 0xfffffu)); }
// plain v_max_f64 / v_min_f64: fmax() / fmin() would put a canonicalising v_max_f64 x, x in front of every operand
__device__ __forceinline__ double list_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double list_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int K>
__device__ __forceinline__ void list_insert_l(double (&kv)[K], double x) {                 // x already in list form
#pragma unroll
    for (int j = K - 1; j > 0; --j) kv[j] = list_max(kv[j], list_min(kv[j - 1], x));
    kv[0] = list_max(kv[0], x);
}
template <int K>
__device__ __forceinline__ void list_insert(double (&kv)[K], unsigned long long key) {
    const double x = to_list(key);
#pragma unroll
    for (int j = K - 1; j > 0; --j) kv[j] = list_max(kv[j], list_min(kv[j - 1], x));
    kv[0] = list_max(kv[0], x);
}

// compare-exchange of two list-form keys: a <- the larger, b <- the smaller
__device__ __forceinline__ void list_cx(double& a, double& b) {
    const double hi = list_max(a, b), lo = list_min(a, b);
    a = hi; b = lo;
}
// TWELVE keys into a descending list of 20 at once (round 3): one insertion is 39 instructions whatever it changes, and a drain of the
// append buffers inserts for the busiest lane (2.6x the average lane).  Instead: sort the twelve with the 39-exchange
// network (depth 9; checked with the 0-1 principle in tests/test_knn_networks.py), think of them behind the list in ascending order --
// 20 descending + 12 ascending is a bitonic sequence of 32 -- and run the bitonic merge pruned to what the first 20 outputs need: the
// first stage pairs list slot i with slot i + 16 (slots 0-3 meet the list's own tail: already in order), the upper 16 are sorted by four
// full stages, of the lower 16 only the maxima of two halving steps and a 4-sorter survive.  78 + 108 instructions for up to twelve keys
// against 39 each; the result is the top 20 of the union, the same list sequential insertion leaves (keys are distinct: they carry
// the index).  Unused entries of `b` = list_empty().
__device__ __forceinline__ void list_merge12(double (&kv)[20], double (&b)[12]) {
    list_cx(b[0], b[8]); list_cx(b[1], b[7]); list_cx(b[2], b[6]); list_cx(b[3], b[11]); list_cx(b[4], b[10]); list_cx(b[5], b[9]);
    list_cx(b[0], b[1]); list_cx(b[2], b[5]); list_cx(b[3], b[4]); list_cx(b[6], b[9]); list_cx(b[7], b[8]); list_cx(b[10], b[11]);
    list_cx(b[0], b[2]); list_cx(b[1], b[6]); list_cx(b[5], b[10]); list_cx(b[9], b[11]);
    list_cx(b[0], b[3]); list_cx(b[1], b[2]); list_cx(b[4], b[6]); list_cx(b[5], b[7]); list_cx(b[8], b[11]); list_cx(b[9], b[10]);
    list_cx(b[1], b[4]); list_cx(b[3], b[5]); list_cx(b[6], b[8]); list_cx(b[7], b[10]);
    list_cx(b[1], b[3]); list_cx(b[2], b[5]); list_cx(b[6], b[9]); list_cx(b[8], b[10]);
    list_cx(b[2], b[3]); list_cx(b[4], b[5]); list_cx(b[6], b[7]); list_cx(b[8], b[9]);
    list_cx(b[4], b[6]); list_cx(b[5], b[7]);
    list_cx(b[3], b[4]); list_cx(b[5], b[6]); list_cx(b[7], b[8]);
    // stage 1 (distance 16): kv[4 + i] meets b[11 - i]; the smaller ones fall into the lower half lo[0..15] = {kv[16..19], b'...}
#pragma unroll
    for (int i = 0; i < 12; ++i) list_cx(kv[4 + i], b[11 - i]);
    // lower half, positions 16..31 = kv[16..19], b[11], b[10], ..., b[0]: its four largest, sorted, are the list's slots 16..19
#pragma unroll
    for (int i = 0; i < 4; ++i) kv[16 + i] = list_max(kv[16 + i], b[7 - i]);           // positions 16+i vs 24+i
#pragma unroll
    for (int i = 0; i < 4; ++i) b[11 - i] = list_max(b[11 - i], b[3 - i]);              // positions 20+i vs 28+i
#pragma unroll
    for (int i = 0; i < 4; ++i) kv[16 + i] = list_max(kv[16 + i], b[11 - i]);           // positions 16+i vs 20+i
    list_cx(kv[16], kv[18]); list_cx(kv[17], kv[19]); list_cx(kv[16], kv[17]); list_cx(kv[18], kv[19]);
    // upper half: bitonic, four full stages
#pragma unroll
    for (int d = 8; d > 0; d >>= 1)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if ((i & d) == 0) list_cx(kv[i], kv[i + d]);
}

// upper bound of the score of ANY point inside an axis-aligned box {min xyz, max xyz, max |p|^2}:
// -dmin^2 (shrunk by 1e-6) + 16 eps (|q|^2 + max |p|^2) -- the margin covers the fp32 rounding of score4()
__device__ inline float box_score_bound(const float4& me, const float* bx) {
    const float dx = fmaxf(fmaxf(bx[0] - me.x, me.x - bx[3]), 0.f);
    const float dy = fmaxf(fmaxf(bx[1] - me.y, me.y - bx[4]), 0.f);
    const float dz = fmaxf(fmaxf(bx[2] - me.z, me.z - bx[5]), 0.f);
    return -((dx * dx + dy * dy) + dz * dz) * 0.999999f + 9.6e-7f * (me.w + bx[6]);
}

}  // namespace sgknn
