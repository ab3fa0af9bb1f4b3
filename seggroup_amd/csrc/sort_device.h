// Device-side sorting for the pre-processing kernels (kernels_prepare.hip): a stable LSD radix sort over exactly the key bits that are
// in use, an exclusive scan and an adjacent-unique compaction.  (Round 4: these replace hipcub::DeviceRadixSort / DeviceSelect, which
// sorted all 64 key bits of the edge lists: eight 8-bit passes, where two 21-bit fields take four 11-bit passes.  sg_mesh_adjacency at
// F = 482,819 faces: 0.62 -> 0.39 ms, sg_segment_lists 0.31 -> 0.24 ms; outputs bit-identical, tests/test_gpu_prepare.py unchanged.)
//
//   radix pass (11 bits):  k_rs_hist      per-block digit counts                      hist[digit][block]
//                          k_rs_rows      one thread per digit: exclusive scan along its row of blocks + the digit total
//                          k_rs_scatter   stable scatter: a block walks its tile 256 keys at a time; a key's rank among the equal
//                                         digits of its wave comes from 11 ballots, the waves' counts are chained through LDS
//   Up to two independent lists ride in one launch (blockIdx.y): the raw and the resampled edge list of sg_mesh_adjacency.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace sgsort {

constexpr int kThreads = 256;
constexpr int kItems = 8;                      // items per thread and tile of the scans
constexpr int kTile = kThreads * kItems;       // 2048 items per block (scans, flags)
constexpr int kBits = 11;                      // radix digits: an edge-key field of 21 bits (ids below 2^20 + the dropped-row bit) is two passes
constexpr int kRadix = 1 << kBits;
constexpr int kSortItems = 16;                 // keys per thread and tile of the radix passes
constexpr int kSortTile = kThreads * kSortItems;    // 4096 keys per block (measured at 1.45 M edge keys: 16 -> 0.39 ms for sg_mesh_adjacency, 32 -> 0.41, 64 -> 0.52)

template <class K, class V>
struct Lists {                                  // up to two lists sorted side by side
    const K* kin[2];
    K* kout[2];
    const V* vin[2];
    V* vout[2];
    int* hist[2];                               // [nblocks][kRadix] counts -> exclusive scans over the blocks, then [kRadix] digit totals
    int n[2];
};

template <class K>
__device__ __forceinline__ unsigned digit_of(K k, int shift) { return (unsigned)(k >> shift) & (kRadix - 1); }

template <class K, class V>
__global__ __launch_bounds__(kThreads) void k_rs_hist(Lists<K, V> L, int shift, int nblocks) {
    __shared__ int cnt[kRadix];
    const int l = blockIdx.y, n = L.n[l];
    const size_t base = (size_t)blockIdx.x * kSortTile;
#pragma unroll
    for (int i = 0; i < kRadix / kThreads; ++i) cnt[threadIdx.x + i * kThreads] = 0;
    __syncthreads();
    if (base < (size_t)n) {
#pragma unroll 4
        for (int i = 0; i < kSortItems; ++i) {
            const size_t p = base + (size_t)i * kThreads + threadIdx.x;
            if (p < (size_t)n) atomicAdd(&cnt[digit_of(L.kin[l][p], shift)], 1);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kRadix / kThreads; ++i) {
        const int d = threadIdx.x + i * kThreads;
        L.hist[l][(size_t)blockIdx.x * kRadix + d] = cnt[d];                  // block-major: coalesced here and in k_rs_rows
    }
}

// one thread per digit: exclusive scan over the blocks' counts of its digit (hist[block][digit]: neighbouring threads read neighbouring
// words; the loads of eight blocks are issued together), the digit's total behind the matrix
template <class K, class V>
__global__ __launch_bounds__(kThreads) void k_rs_rows(Lists<K, V> L, int nblocks) {
    const int d = blockIdx.x * kThreads + threadIdx.x;
    int* col = L.hist[blockIdx.y] + d;
    int run = 0;
    int b = 0;
    for (; b + 8 <= nblocks; b += 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(b + u) * kRadix];
#pragma unroll
        for (int u = 0; u < 8; ++u) { col[(size_t)(b + u) * kRadix] = run; run += v[u]; }
    }
    for (; b < nblocks; ++b) { const int v = col[(size_t)b * kRadix]; col[(size_t)b * kRadix] = run; run += v; }
    L.hist[blockIdx.y][(size_t)kRadix * nblocks + d] = run;
}

template <class K, class V, bool kPairs>
__global__ __launch_bounds__(kThreads) void k_rs_scatter(Lists<K, V> L, int shift, int nblocks) {
    __shared__ int run[kRadix];                 // where the next key of each digit goes
    __shared__ int wcnt[kThreads / 64][kRadix];
    __shared__ int part[kThreads];
    const int l = blockIdx.y, n = L.n[l];
    const size_t base = (size_t)blockIdx.x * kSortTile;
    if (base >= (size_t)n) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // digit bases: exclusive scan of the digit totals (each thread owns kRadix / kThreads consecutive digits) + this block's row offsets
    constexpr int kPer = kRadix / kThreads;
    const int* tot = L.hist[l] + (size_t)kRadix * nblocks;
    int mine[kPer], s = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) { mine[i] = tot[threadIdx.x * kPer + i]; s += mine[i]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {
        const int x = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += x;
        __syncthreads();
    }
    int acc = part[threadIdx.x] - s;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int d = threadIdx.x * kPer + i;
        run[d] = acc + L.hist[l][(size_t)blockIdx.x * kRadix + d];
        acc += mine[i];
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) wcnt[w][d] = 0;
    }
    __syncthreads();
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int i = 0; i < kSortItems; ++i) {
        const size_t p = base + (size_t)i * kThreads + threadIdx.x;
        if (base + (size_t)i * kThreads >= (size_t)n) break;            // uniform: the tile ends here
        const bool live = p < (size_t)n;
        const K key = live ? L.kin[l][p] : (K)0;
        const unsigned d = digit_of(key, shift);
        // the lanes of this wave that hold the same digit
        unsigned long long peers = __builtin_amdgcn_ballot_w64(live);
#pragma unroll
        for (int b = 0; b < kBits; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(bit);
            peers &= bit ? m : ~m;
        }
        const int rank = __builtin_popcountll(peers & below);
        const int count = __builtin_popcountll(peers);
        const bool leader = live && rank == 0;
        if (leader) wcnt[wave][d] = count;
        __syncthreads();
        if (live) {
            int dst = run[d] + rank;
            for (int w = 0; w < wave; ++w) dst += wcnt[w][d];
            L.kout[l][dst] = key;
            if constexpr (kPairs) L.vout[l][dst] = L.vin[l][p];
        }
        __syncthreads();
        if (leader) { atomicAdd(&run[d], count); wcnt[wave][d] = 0; }
        __syncthreads();
    }
}

// exclusive scan of m ints in place by ONE block (tile sums of the long scans: a few thousand entries)
__device__ __forceinline__ void block_scan_inplace(int* __restrict__ a, int m) {
    __shared__ int part[1024];
    const int t = threadIdx.x, T = blockDim.x;
    const int per = (m + T - 1) / T;
    const int lo = min(t * per, m), hi = min(lo + per, m);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < T; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = lo; i < hi; ++i) { const int v = a[i]; a[i] = run; run += v; }
}

// ---- exclusive scan of a long int array (three launches) and the adjacent-unique compaction built on it ------------------------------
template <class K>
__global__ __launch_bounds__(kThreads) void k_head_flags(const K* __restrict__ keys, int n, int* __restrict__ flag) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}
__global__ __launch_bounds__(kThreads) void k_scan_tiles(int* __restrict__ a, int n, int* __restrict__ tile_sum) {
    __shared__ int part[kThreads];
    const size_t base = (size_t)blockIdx.x * kTile + (size_t)threadIdx.x * kItems;
    int v[kItems], s = 0;
#pragma unroll
    for (int i = 0; i < kItems; ++i) { v[i] = base + i < (size_t)n ? a[base + i] : 0; s += v[i]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {
        const int x = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += x;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
#pragma unroll
    for (int i = 0; i < kItems; ++i) { if (base + i < (size_t)n) a[base + i] = run; run += v[i]; }
    if (threadIdx.x == kThreads - 1) tile_sum[blockIdx.x] = part[threadIdx.x];
}
__global__ __launch_bounds__(1024) void k_scan_tile_sums(int* __restrict__ tile_sum, int ntiles, int* __restrict__ total) {
    // one slot past the tiles receives the grand total
    block_scan_inplace(tile_sum, ntiles + 1);
    __syncthreads();
    if (threadIdx.x == 0 && total) *total = tile_sum[ntiles];
}
template <class K>
__global__ __launch_bounds__(kThreads) void k_compact_heads(const K* __restrict__ keys, const int* __restrict__ pos, const int* __restrict__ tile_sum,
                                                           int n, K* __restrict__ out, int* __restrict__ head_index) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    if (i == 0 || keys[i] != keys[i - 1]) {
        const int dst = pos[i] + tile_sum[i / kTile];
        if (out) out[dst] = keys[i];
        if (head_index) head_index[dst] = i;
    }
}

inline int tiles_of(long long n) { return (int)((n + kTile - 1) / kTile); }
inline int sort_tiles_of(long long n) { return (int)((n + kSortTile - 1) / kSortTile); }

// ints of scratch a radix sort of `n` keys needs per list (row histograms + digit totals), and the unique compaction (flags + tile sums)
inline size_t hist_ints(long long n) { return (size_t)kRadix * (sort_tiles_of(n > 0 ? n : 1) + 1); }
inline size_t unique_ints(long long n) { return (size_t)(n > 0 ? n : 1) + tiles_of(n > 0 ? n : 1) + 2; }

// one pass over the digit at `shift` for every list; swaps (kin, kout) / (vin, vout) in L
template <class K, class V, bool kPairs>
inline void radix_pass(Lists<K, V>& L, int nlists, int shift, int nblocks, hipStream_t st) {
    k_rs_hist<K, V><<<dim3(nblocks, nlists), kThreads, 0, st>>>(L, shift, nblocks);
    k_rs_rows<K, V><<<dim3(kRadix / kThreads, nlists), kThreads, 0, st>>>(L, nblocks);
    k_rs_scatter<K, V, kPairs><<<dim3(nblocks, nlists), kThreads, 0, st>>>(L, shift, nblocks);
    for (int l = 0; l < nlists; ++l) {
        const K* k = L.kin[l]; L.kin[l] = L.kout[l]; L.kout[l] = const_cast<K*>(k);
        const V* v = L.vin[l]; L.vin[l] = L.vout[l]; L.vout[l] = const_cast<V*>(v);
    }
}

// Stable LSD sort of the bits [lo_bit, lo_bit + bits).  Returns the number of passes: after an odd number the result sits in what was
// (kout, vout), after an even number in (kin, vin) -- L is updated so that L.kin / L.vin always name the current result.
template <class K, class V, bool kPairs>
inline int radix_sort(Lists<K, V>& L, int nlists, int lo_bit, int bits, hipStream_t st) {
    long long nmax = 0;
    for (int l = 0; l < nlists; ++l) nmax = L.n[l] > nmax ? L.n[l] : nmax;
    if (nmax <= 0) return 0;
    const int nblocks = sort_tiles_of(nmax);
    int passes = 0;
    for (int shift = lo_bit; shift < lo_bit + bits; shift += kBits, ++passes) radix_pass<K, V, kPairs>(L, nlists, shift, nblocks, st);
    return passes;
}

// keys sorted ascending -> out[0..count) = the distinct keys (may be NULL), head_index[0..count) = where each run starts (may be NULL);
// *d_count = count.  scratch: unique_ints(n) ints.
template <class K>
inline void unique_sorted(const K* keys, int n, K* out, int* head_index, int* d_count, int* scratch, hipStream_t st) {
    const int nt = tiles_of(n);
    int* flag = scratch;
    int* tsum = scratch + n;
    k_head_flags<K><<<(n + kThreads - 1) / kThreads, kThreads, 0, st>>>(keys, n, flag);
    k_scan_tiles<<<nt, kThreads, 0, st>>>(flag, n, tsum);
    k_scan_tile_sums<<<1, 1024, 0, st>>>(tsum, nt, d_count);
    k_compact_heads<K><<<(n + kThreads - 1) / kThreads, kThreads, 0, st>>>(keys, flag, tsum, n, out, head_index);
}

}  // namespace sgsort
