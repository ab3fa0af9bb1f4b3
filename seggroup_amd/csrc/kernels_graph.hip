// Graph / label bookkeeping kernels (HBM-bound integer work): point-edge contraction (a3), member
// gathering (a2/a10), edge distance (a8), row/segment max (a10), label export (a16), metric counts (a17).
#include "engine_ctx.h"
#include "sg_common.h"
#include "wave_ops.h"

namespace {

using sg::SlotCtx;

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------
// a3: update_adj first call (model.py:291-302, 724-733).  One bit per (lo,hi) segment pair.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mark_pairs_body(const int64_t* __restrict__ adj, int E, const int32_t* __restrict__ seg, int N, int S,
                                                uint32_t* __restrict__ bitmap, int bid, int nblk) {
    for (int e = bid * blockDim.x + threadIdx.x; e < E; e += nblk * blockDim.x) {
        // 16-byte coalesced read of one edge row
        const longlong2 row = reinterpret_cast<const longlong2*>(adj)[e];
        const long long a = row.x, b = row.y;
        if (a < 0 || b < 0 || a >= N || b >= N) continue;
        int sa = seg[a], sb = seg[b];
        if (sa == sb || sa < 0 || sb < 0) continue;
        if (sa > sb) { int t = sa; sa = sb; sb = t; }
        const unsigned long long bit = (unsigned long long)sa * (unsigned)S + (unsigned)sb;
        const uint32_t mask = 1u << (bit & 31);
        uint32_t* w = bitmap + (bit >> 5);
        if ((__builtin_nontemporal_load(w) & mask) == 0) atomicOr(w, mask);   // ~60 edges hit each bit
    }
}
__global__ void k_mark_pairs(const int64_t* __restrict__ adj, int E, const int32_t* __restrict__ seg, int N, int S,
                             uint32_t* __restrict__ bitmap) {
    mark_pairs_body(adj, E, seg, N, S, bitmap, blockIdx.x, gridDim.x);
}
__global__ void k_mark_pairs_b(const SlotCtx* __restrict__ cx) {
    const SlotCtx& c = cx[blockIdx.y];
    mark_pairs_body(c.adj0, c.E0, c.seg_of_point, c.N, c.S, c.bitmap, blockIdx.x, gridDim.x);
}

constexpr int kWordsPerThread = 4;
constexpr int kWordsPerBlock = kBlock * kWordsPerThread;

__device__ inline int block_exclusive_scan(int v, int* total) {
    __shared__ int wsum[kBlock / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        if (w < wid) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__device__ __forceinline__ void count_bits_body(const uint32_t* __restrict__ bitmap, size_t words, int* __restrict__ block_count, int bid) {
    const size_t w0 = (size_t)bid * kWordsPerBlock + (size_t)threadIdx.x * kWordsPerThread;
    int c = 0;
#pragma unroll
    for (int i = 0; i < kWordsPerThread; ++i)
        if (w0 + i < words) c += __popc(bitmap[w0 + i]);
    int total;
    block_exclusive_scan(c, &total);
    if (threadIdx.x == 0) block_count[bid] = total;
}
__global__ void k_count_bits(const uint32_t* __restrict__ bitmap, size_t words, int* __restrict__ block_count) {
    count_bits_body(bitmap, words, block_count, blockIdx.x);
}
__global__ void k_count_bits_b(const SlotCtx* __restrict__ cx) {
    const SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x >= c.bits_blocks) return;
    count_bits_body(c.bitmap, (size_t)c.bitmap_words, c.block_count, blockIdx.x);
}

// single block: exclusive scan of the per-block counts, total -> *out_count
__device__ __forceinline__ void scan_blocks_body(int* __restrict__ block_count, int nblocks, int* __restrict__ out_count) {
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += kBlock) {
        const int i = base + threadIdx.x;
        const int v = i < nblocks ? block_count[i] : 0;
        int total;
        const int ex = block_exclusive_scan(v, &total);
        const int c = carry;
        if (i < nblocks) block_count[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_count = carry;
}
__global__ void k_scan_blocks(int* __restrict__ block_count, int nblocks, int* __restrict__ out_count) {
    scan_blocks_body(block_count, nblocks, out_count);
}
__global__ void k_scan_blocks_b(const SlotCtx* __restrict__ cx) {
    const SlotCtx& c = cx[blockIdx.y];
    scan_blocks_body(c.block_count, c.bits_blocks, c.count);
}

// kClear: the words are zeroed as they are read, so the bitmap is all-zero again when the launch ends (the engine's
// scenes then need no memset launch); out2 (may be null) receives a second copy of the first cap2 rows (the outbox)
template <bool kClear>
__device__ __forceinline__ void emit_pairs_body(uint32_t* __restrict__ bitmap, size_t words, const int* __restrict__ block_off, int S,
                                                int32_t* __restrict__ out, int capacity, int32_t* __restrict__ out2, int cap2, int bid) {
    const size_t w0 = (size_t)bid * kWordsPerBlock + (size_t)threadIdx.x * kWordsPerThread;
    uint32_t w[kWordsPerThread];
    int c = 0;
#pragma unroll
    for (int i = 0; i < kWordsPerThread; ++i) {
        w[i] = (w0 + i < words) ? bitmap[w0 + i] : 0u;
        if (kClear && w[i]) bitmap[w0 + i] = 0u;
        c += __popc(w[i]);
    }
    int total;
    int off = block_off[bid] + block_exclusive_scan(c, &total);
#pragma unroll
    for (int i = 0; i < kWordsPerThread; ++i) {
        uint32_t bits = w[i];
        while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            const unsigned long long bit = ((unsigned long long)(w0 + i) << 5) + b;
            const int32_t lo = (int32_t)(bit / (unsigned)S), hi = (int32_t)(bit % (unsigned)S);
            if (off < capacity) {
                out[2 * off] = lo;
                out[2 * off + 1] = hi;
            }
            if (out2 && off < cap2) {
                out2[2 * off] = lo;
                out2[2 * off + 1] = hi;
            }
            ++off;
        }
    }
}
__global__ void k_emit_pairs(uint32_t* __restrict__ bitmap, size_t words, const int* __restrict__ block_off, int S,
                             int32_t* __restrict__ out, int capacity) {
    emit_pairs_body<false>(bitmap, words, block_off, S, out, capacity, nullptr, 0, blockIdx.x);
}
__global__ void k_emit_pairs_b(const SlotCtx* __restrict__ cx) {
    const SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x >= c.bits_blocks) return;
    emit_pairs_body<true>(c.bitmap, (size_t)c.bitmap_words, c.block_count, c.S, c.adj1, c.cap1, c.adj1_out, c.out_rows, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// a2/a10: expand an ordered list of segments per cluster into point-level member arrays
// ------------------------------------------------------------------------------------------------
__global__ void k_gather_members(const int32_t* __restrict__ seg_points, const int32_t* __restrict__ seg_off,
                                 const int32_t* __restrict__ order, const int32_t* __restrict__ dst,
                                 const int32_t* __restrict__ cl, int32_t* __restrict__ members,
                                 int32_t* __restrict__ pos_of_point, int32_t* __restrict__ cluster_of_pos,
                                 int32_t* __restrict__ slot_of_pos) {
    const int i = blockIdx.x;                 // i-th segment in member order
    const int s = order[i];
    const int lo = seg_off[s], n = seg_off[s + 1] - lo, d = dst[i], c = cl[i];
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const int p = seg_points[lo + t];
        members[d + t] = p;
        if (pos_of_point) pos_of_point[p] = d + t;
        if (cluster_of_pos) cluster_of_pos[d + t] = c;
        if (slot_of_pos) slot_of_pos[d + t] = i;
    }
}

// ------------------------------------------------------------------------------------------------
// a8: calculate_distance (model.py:269-274).  One wave per edge, fp64 accumulation.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void edge_distance_one(const float* __restrict__ feat, int stride, int D, const int32_t* __restrict__ adj, int e,
                                                  float* __restrict__ dist, float* __restrict__ dist2, int cap2) {
    const int lane = threadIdx.x & 63;
    const float* a = feat + (size_t)adj[2 * e] * stride;
    const float* b = feat + (size_t)adj[2 * e + 1] * stride;
    double acc = 0.0;
    int k = lane;
    for (; k + 64 < D; k += 128) {                                // two strides per trip, their four loads in flight together
        const float a0 = a[k], a1 = a[k + 64], b0 = b[k], b1 = b[k + 64];
        double d = (double)a0 - (double)b0 + 1e-6; acc = fma(d, d, acc);
        d = (double)a1 - (double)b1 + 1e-6; acc = fma(d, d, acc);
    }
    for (; k < D; k += 64) {
        const double d = (double)a[k] - (double)b[k] + 1e-6;
        acc = fma(d, d, acc);
    }
    acc = sgw::wave_sum(acc);                                   // DPP network (wave_ops.h), fixed order
    if (lane == 0) {
        const float v = (float)sqrt(acc);
        dist[e] = v;
        if (dist2 && e < cap2) dist2[e] = v;
    }
}
__global__ void k_edge_distance(const float* __restrict__ feat, int stride, int D, const int32_t* __restrict__ adj, int E,
                                float* __restrict__ dist) {
    const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (e >= E) return;
    edge_distance_one(feat, stride, D, adj, e, dist, nullptr, 0);
}
// wave-stride over the edges: the edge count may live on the device (the contraction of the same phase produced it)
__global__ void k_edge_distance_b(const SlotCtx* __restrict__ cx) {
    const SlotCtx& c = cx[blockIdx.y];
    const int E = c.dist_E_dev ? min(*c.dist_E_dev, c.cap1) : c.dist_E;
    const int wpb = blockDim.x >> 6;
    for (int e = blockIdx.x * wpb + (threadIdx.x >> 6); e < E; e += gridDim.x * wpb)
        edge_distance_one(c.dist_feat, c.dist_stride, c.dist_D, c.dist_adj, e, c.dist, c.dist_copy, c.dist_copy_rows);
}

// ------------------------------------------------------------------------------------------------
// a10: aggregate_cluster_feature (model.py:278-288)
// ------------------------------------------------------------------------------------------------
// `fill_cols` further columns of every output row are set to -inf: the pipeline's next writer of those rows is the
// atomic point->cluster max, which then needs no fill launch of its own
__device__ __forceinline__ void group_max_rows_body(const float* __restrict__ rows, int row_stride, int D, const int32_t* __restrict__ goff,
                                                    const int32_t* __restrict__ gidx, float* __restrict__ out, int out_stride, int fill_cols, int g) {
    const int lo = goff[g], hi = goff[g + 1];
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
        float m = -INFINITY;
        int i = lo;
        for (; i + 3 < hi; i += 4) {                              // four child ids, then their four rows, in flight together
            const int a = gidx[i], b = gidx[i + 1], c = gidx[i + 2], d = gidx[i + 3];
            const float va = rows[(size_t)a * row_stride + k], vb = rows[(size_t)b * row_stride + k], vc = rows[(size_t)c * row_stride + k],
                        vd = rows[(size_t)d * row_stride + k];
            m = fmaxf(fmaxf(m, fmaxf(va, vb)), fmaxf(vc, vd));
        }
        for (; i < hi; ++i) m = fmaxf(m, rows[(size_t)gidx[i] * row_stride + k]);
        out[(size_t)g * out_stride + k] = m;
    }
    for (int k = threadIdx.x; k < fill_cols; k += blockDim.x) out[(size_t)g * out_stride + D + k] = -INFINITY;
}
__global__ void k_group_max_rows(const float* __restrict__ rows, int row_stride, int D, const int32_t* __restrict__ goff,
                                 const int32_t* __restrict__ gidx, float* __restrict__ out, int out_stride, int fill_cols) {
    group_max_rows_body(rows, row_stride, D, goff, gidx, out, out_stride, fill_cols, blockIdx.x);
}
__global__ void k_group_max_rows_b(const SlotCtx* __restrict__ cx) {
    const SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x >= c.C) return;
    group_max_rows_body(c.gm_rows, c.gm_stride, c.gm_D, c.goff, c.gidx, c.cat, c.Dcat, 64, blockIdx.x);
}

__device__ inline void atomic_max_float(float* addr, float v) {
    // order-preserving integer view: non-negative floats compare as ints, negative floats reversed as uints
    if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void k_fill(float* __restrict__ p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void k_fill_rows(float* __restrict__ p, int rows, int cols, int stride, float v) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[(i / cols) * stride + (i % cols)] = v;
}

// rows are in member order (clusters contiguous, cluster ids ascending).  Block = 16 channel quads x 16 row lanes over
// kRowsPerBlock rows: every thread issues its float4 loads (and the cluster ids) before it touches any of them; then, for
// each cluster id the block spans (usually one or two), the 16 row lanes are reduced through LDS and ONE atomic per
// (cluster, channel) leaves the block.  (The first version walked 128 rows per thread with one dependent 4-byte load per
// step -- 1,172 waves, 87 % of their cycles in s_waitcnt, 0.8 TB/s; per-thread flushes at this granularity drown in
// contended float-max atomics instead.)
constexpr int kRowsPerBlock = 64;
// a != nullptr: the rows are EdgeConv's pre-activation maxima E and every element is first mapped to LReLU(a_c * E + b_c)
// (the same fmaf + fmaxf as k_bn_lrelu_apply), which saves that kernel's launch and a read + write of the [N,64] array.
__device__ __forceinline__ void segment_max64_body(const float* __restrict__ rows, int N, const int32_t* __restrict__ cluster_of_pos,
                                                   float* __restrict__ out, int out_stride, const float* __restrict__ a,
                                                   const float* __restrict__ shift, int bid) {
    constexpr int kIter = kRowsPerBlock / 16;
    __shared__ float4 red[16][16];
    const int qi = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int b0 = bid * kRowsPerBlock, r0 = b0 + rl;
    float4 v[kIter];
    int c[kIter];
#pragma unroll
    for (int i = 0; i < kIter; ++i) {
        const int r = r0 + 16 * i;
        c[i] = r < N ? cluster_of_pos[r] : -1;
        v[i] = r < N ? *reinterpret_cast<const float4*>(rows + (size_t)r * 64 + qi * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (a) {
        const float4 a4 = *reinterpret_cast<const float4*>(a + qi * 4), s4 = *reinterpret_cast<const float4*>(shift + qi * 4);
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            float y;
            y = __builtin_fmaf(a4.x, v[i].x, s4.x); v[i].x = fmaxf(y, 0.2f * y);
            y = __builtin_fmaf(a4.y, v[i].y, s4.y); v[i].y = fmaxf(y, 0.2f * y);
            y = __builtin_fmaf(a4.z, v[i].z, s4.z); v[i].z = fmaxf(y, 0.2f * y);
            y = __builtin_fmaf(a4.w, v[i].w, s4.w); v[i].w = fmaxf(y, 0.2f * y);
        }
    }
    const int c_first = cluster_of_pos[b0], c_last = cluster_of_pos[min(b0 + kRowsPerBlock, N) - 1];
    for (int cl = c_first; cl <= c_last; ++cl) {
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int i = 0; i < kIter; ++i)
            if (c[i] == cl) { m.x = fmaxf(m.x, v[i].x); m.y = fmaxf(m.y, v[i].y); m.z = fmaxf(m.z, v[i].z); m.w = fmaxf(m.w, v[i].w); }
        __syncthreads();
        red[rl][qi] = m;
        __syncthreads();
        if (threadIdx.x < 64) {
            const float* col = &red[0][0].x + threadIdx.x;                 // channel threadIdx.x of row lane k at col[64 * k]
            float t = col[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) t = fmaxf(t, col[64 * k]);
            if (t > -INFINITY) atomic_max_float(out + (size_t)cl * out_stride + threadIdx.x, t);
        }
    }
}
__global__ __launch_bounds__(256) void k_segment_max64(const float* __restrict__ rows, int N, const int32_t* __restrict__ cluster_of_pos,
                                                       float* __restrict__ out, int out_stride, const float* __restrict__ a,
                                                       const float* __restrict__ shift) {
    segment_max64_body(rows, N, cluster_of_pos, out, out_stride, a, shift, blockIdx.x);
}
// (the engine has no batched twin of this kernel: its EdgeConv launches reduce E to the clusters' maxima themselves, kernels_edgeconv.hip)

// ------------------------------------------------------------------------------------------------
// a16: label export gather (model.py:525-605)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void export_body(const int32_t* __restrict__ unmap, int V, const int32_t* __restrict__ seg_of_point, int N,
                                            const int32_t* __restrict__ tables, int T, int S, int32_t* __restrict__ out, int bid, int nblk) {
    for (int v = bid * blockDim.x + threadIdx.x; v < V; v += nblk * blockDim.x) {
        const int p = unmap[v];
        const int s = (p >= 0 && p < N) ? seg_of_point[p] : -1;
        for (int t = 0; t < T; ++t) out[(size_t)t * V + v] = (s >= 0 && s < S) ? tables[(size_t)t * S + s] : -1;
    }
}
__global__ void k_export(const int32_t* __restrict__ unmap, int V, const int32_t* __restrict__ seg_of_point, int N,
                         const int32_t* __restrict__ tables, int T, int S, int32_t* __restrict__ out) {
    export_body(unmap, V, seg_of_point, N, tables, T, S, out, blockIdx.x, gridDim.x);
}
__global__ void k_eval_clear(uint32_t* __restrict__ cnt, int n0, int n1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n1) cnt[i] = i < n0 ? 0u : 0xffffffffu;
}
// also resets the metric counters the evaluate kernels of the same phase accumulate into (saves two memset launches)
__global__ void k_export_b(const SlotCtx* __restrict__ cx) {
    const SlotCtx& c = cx[blockIdx.y];
    const int n0 = 128 + 3 * c.max_ins, n1 = n0 + c.max_ins;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += gridDim.x * blockDim.x) c.cnt[i] = i < n0 ? 0u : 0xffffffffu;
    export_body(c.unmap, c.V, c.seg_of_point, c.N, c.tables, c.n_tables, c.S, c.labels, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// a17: evaluate (model.py:608-655): integer counts
// counters layout (uint32): [0..39] sem pred hist, [40..79] sem true hist, [80..119] sem both,
// [120..127] scalars {n_valid, sem_eq, ins_eq, n_semvalid, semvalid_eq, n_insvalid, insvalid_eq, -},
// then 4 arrays of max_ins: ins pred count, ins true count, ins both, first valid vertex (atomicMin)
// ------------------------------------------------------------------------------------------------
__device__ inline bool in_sem_valid(int c) {   // SEM_VALID_CLASS_IDS (model.py:27)
    const unsigned long long m = (1ull << 1) | (1ull << 2) | (1ull << 3) | (1ull << 4) | (1ull << 5) | (1ull << 6) | (1ull << 7) |
                                 (1ull << 8) | (1ull << 9) | (1ull << 10) | (1ull << 11) | (1ull << 12) | (1ull << 14) |
                                 (1ull << 16) | (1ull << 24) | (1ull << 28) | (1ull << 33) | (1ull << 34) | (1ull << 36) | (1ull << 39);
    return c >= 0 && c < 64 && ((m >> c) & 1ull);
}
__device__ inline bool in_ins_valid(int c) {   // INS_VALID_CLASS_IDS (model.py:28)
    return in_sem_valid(c) && c != 1 && c != 2;
}

// Per-thread register counters for the 7 scalars (one wave-reduced atomic per wave at the end), LDS
// histograms for the 3x40 semantic bins and -- when they fit (max_ins <= kInsLds) -- for the four
// per-instance arrays, flushed once per block: no hot global atomics.
constexpr int kInsLds = 2048;
__device__ __forceinline__ void eval_counts_body(const int32_t* __restrict__ gt, const int32_t* __restrict__ sem_pred,
                                                 const int32_t* __restrict__ ins_pred, int V, int max_ins,
                                                 uint32_t* __restrict__ cnt, int bid, int nblk) {
    extern __shared__ uint32_t dyn[];                       // [4 * max_ins] when max_ins <= kInsLds
    __shared__ uint32_t h[128];
    const bool lds_ins = max_ins <= kInsLds;
    for (int i = threadIdx.x; i < 128; i += blockDim.x) h[i] = 0;
    if (lds_ins)
        for (int i = threadIdx.x; i < 4 * max_ins; i += blockDim.x) dyn[i] = i < 3 * max_ins ? 0u : 0xffffffffu;
    __syncthreads();
    uint32_t* g_p = cnt + 128;
    uint32_t* g_t = g_p + max_ins;
    uint32_t* g_b = g_t + max_ins;
    uint32_t* g_f = g_b + max_ins;
    uint32_t* ins_p = lds_ins ? dyn : g_p;
    uint32_t* ins_t = lds_ins ? dyn + max_ins : g_t;
    uint32_t* ins_b = lds_ins ? dyn + 2 * max_ins : g_b;
    uint32_t* first = lds_ins ? dyn + 3 * max_ins : g_f;
    uint32_t sc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int v = bid * blockDim.x + threadIdx.x; v < V; v += nblk * blockDim.x) {
        const int2 g = reinterpret_cast<const int2*>(gt)[v];
        const int st = g.x, it = g.y;
        if (st == 0) continue;                                   // valid_idxs (model.py:615)
        const int sp = sem_pred[v], ip = ins_pred[v];
        sc[0] += 1;
        if (sp >= 1 && sp <= 40) atomicAdd(&h[sp - 1], 1u);
        if (st >= 1 && st <= 40) atomicAdd(&h[40 + st - 1], 1u);
        if (sp == st) {
            sc[1] += 1;
            if (sp >= 1 && sp <= 40) atomicAdd(&h[80 + sp - 1], 1u);
        }
        sc[2] += (ip == it);
        if (in_sem_valid(st)) { sc[3] += 1; sc[4] += (sp == st); }
        if (in_ins_valid(it)) { sc[5] += 1; sc[6] += (ip == it); }
        if (ip >= 0 && ip < max_ins) {
            atomicAdd(&ins_p[ip], 1u);
            atomicMin(&first[ip], (uint32_t)v);
            if (it == ip) atomicAdd(&ins_b[ip], 1u);
        }
        if (it >= 0 && it < max_ins) atomicAdd(&ins_t[it], 1u);
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        uint32_t x = sc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&h[120 + k], x);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 128; i += blockDim.x)
        if (h[i]) atomicAdd(&cnt[i], h[i]);
    if (lds_ins)
        for (int i = threadIdx.x; i < max_ins; i += blockDim.x) {
            if (dyn[i]) atomicAdd(&g_p[i], dyn[i]);
            if (dyn[max_ins + i]) atomicAdd(&g_t[i], dyn[max_ins + i]);
            if (dyn[2 * max_ins + i]) atomicAdd(&g_b[i], dyn[2 * max_ins + i]);
            if (dyn[3 * max_ins + i] != 0xffffffffu) atomicMin(&g_f[i], dyn[3 * max_ins + i]);
        }
}
__global__ __launch_bounds__(256) void k_eval_counts(const int32_t* __restrict__ gt, const int32_t* __restrict__ sem_pred,
                                                     const int32_t* __restrict__ ins_pred, int V, int max_ins,
                                                     uint32_t* __restrict__ cnt) {
    eval_counts_body(gt, sem_pred, ins_pred, V, max_ins, cnt, blockIdx.x, gridDim.x);
}
// dynamic LDS is sized for the launch's largest max_ins; a scene whose own max_ins fits uses the LDS histograms
__global__ __launch_bounds__(256) void k_eval_counts_b(const SlotCtx* __restrict__ cx) {
    const SlotCtx& c = cx[blockIdx.y];
    eval_counts_body(c.gt, c.labels + (size_t)c.sem_row * c.V, c.labels + (size_t)c.ins_row * c.V, c.V, c.max_ins, c.cnt, blockIdx.x, gridDim.x);
}

// semantic prediction at the first valid vertex of every predicted instance (model.py:636)
__device__ __forceinline__ void eval_first_sem_body(const int32_t* __restrict__ sem_pred, int max_ins, uint32_t* __restrict__ cnt, int bid) {
    const int i = bid * blockDim.x + threadIdx.x;
    if (i >= max_ins) return;
    const uint32_t* ins_p = cnt + 128;
    const uint32_t* first = ins_p + 3 * (size_t)max_ins;
    uint32_t* fsem = cnt + 128 + 4 * (size_t)max_ins;
    fsem[i] = ins_p[i] ? (uint32_t)sem_pred[first[i]] : 0xffffffffu;
}
__global__ void k_eval_first_sem(const int32_t* __restrict__ sem_pred, int max_ins, uint32_t* __restrict__ cnt) {
    eval_first_sem_body(sem_pred, max_ins, cnt, blockIdx.x);
}
__global__ void k_eval_first_sem_b(const SlotCtx* __restrict__ cx) {
    const SlotCtx& c = cx[blockIdx.y];
    eval_first_sem_body(c.labels + (size_t)c.sem_row * c.V, c.max_ins, c.cnt, blockIdx.x);
}

// arena-to-arena copies by a kernel (sg::copy_by_kernel): 16 bytes per thread and trip
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_copy4(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

// ================================================================================================
namespace sg {

int group_max_rows_fill(const float* d_rows, int row_stride, int D, const int32_t* d_goff, const int32_t* d_gidx, int G, float* d_out,
                        int out_stride, int fill_cols, void* stream) {
    if (G == 0) return SG_OK;
    k_group_max_rows<<<G, 64 * ((std::min(D, 256) + 63) / 64), 0, sg::as_stream(stream)>>>(d_rows, row_stride, D, d_goff, d_gidx, d_out,
                                                                                           out_stride, fill_cols);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

// d_out's 64 columns already hold -inf (or earlier maxima)
int segment_max_prefilled(const float* d_rows, int N, const int32_t* d_cluster_of_pos, float* d_out, int out_stride, void* stream,
                          const float* d_a, const float* d_shift) {
    if (N == 0) return SG_OK;
    k_segment_max64<<<sg::cdiv(N, kRowsPerBlock), 256, 0, sg::as_stream(stream)>>>(d_rows, N, d_cluster_of_pos, d_out, out_stride, d_a,
                                                                                   d_shift);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int copy_by_kernel(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return SG_OK;
    const bool wide = (((size_t)dst | (size_t)src | bytes) & 15) == 0;
    if (wide) {
        const size_t n16 = bytes / 16;
        k_copy16<<<(unsigned)std::min<size_t>((n16 + 255) / 256, 2048), 256, 0, st>>>(reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n16);
    } else {
        if ((((size_t)dst | (size_t)src | bytes) & 3) != 0) return sg::fail(SG_EINVAL, "copy_by_kernel: addresses and size must be multiples of 4");
        const size_t n4 = bytes / 4;
        k_copy4<<<(unsigned)std::min<size_t>((n4 + 255) / 256, 2048), 256, 0, st>>>(reinterpret_cast<const uint32_t*>(src), reinterpret_cast<uint32_t*>(dst), n4);
    }
    SG_LAUNCH_CHECK();
    return SG_OK;
}

// ---- batched launches (engine.cpp) ----
int b_contract(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st) {
    if (bd.nslots == 0) return SG_OK;
    const dim3 gy(1, bd.nslots);
    // ~1,024 workgroups per launch, each walking ~14 edges per thread: measured per launch of 8 scenes, blocks per scene 4,096 /
    // 1,024 / 512 / 256 / 128 / 64 -> 124 / 97 / 71 / 63 / 57 / 66 us (the edge rows stream, the two segment-id gathers per edge and
    // the bitmap atomics do better with fewer, longer-lived waves)
    if (bd.max_E0 > 0) {
        const int per_scene = std::min(sg::cdiv(bd.max_E0, kBlock), std::max(128, 1024 / std::max(bd.nslots, 1)));
        k_mark_pairs_b<<<dim3(per_scene, bd.nslots), kBlock, 0, st>>>(d_ctx);
    }
    k_count_bits_b<<<dim3(bd.max_bits_blocks, bd.nslots), kBlock, 0, st>>>(d_ctx);
    k_scan_blocks_b<<<gy, kBlock, 0, st>>>(d_ctx);
    k_emit_pairs_b<<<dim3(bd.max_bits_blocks, bd.nslots), kBlock, 0, st>>>(d_ctx);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int b_edge_distance(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st) {
    if (bd.nslots == 0) return SG_OK;
    const int blocks = std::max(1, std::min(sg::cdiv(std::max(bd.max_E, 1), 4), 512));
    k_edge_distance_b<<<dim3(blocks, bd.nslots), 256, 0, st>>>(d_ctx);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int b_group_max_fill(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st) {
    if (bd.nslots == 0 || bd.max_C == 0) return SG_OK;
    k_group_max_rows_b<<<dim3(bd.max_C, bd.nslots), 256, 0, st>>>(d_ctx);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int b_export_eval(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st) {
    if (bd.nslots == 0) return SG_OK;
    k_export_b<<<dim3(std::max(1, std::min(sg::cdiv(bd.max_V, 256), std::max(128, 1024 / std::max(bd.nslots, 1)))), bd.nslots), 256, 0, st>>>(d_ctx);
    const size_t dyn = (size_t)std::min(std::max(bd.max_ins, 1), kInsLds) * 16;     // every scene with max_ins <= kInsLds fits
    k_eval_counts_b<<<dim3(std::max(1, std::min(sg::cdiv(bd.max_V, 1024), 256)), bd.nslots), 256, dyn, st>>>(d_ctx);
    k_eval_first_sem_b<<<dim3(sg::cdiv(std::max(bd.max_ins, 1), 256), bd.nslots), 256, 0, st>>>(d_ctx);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

void eval_finish(const uint32_t* h, int max_ins, float* h_iou_sem, float* h_iou_ins, float* h_acc) {
    const uint32_t* ins_p = h + 128;
    const uint32_t* ins_t = ins_p + max_ins;
    const uint32_t* ins_b = ins_t + max_ins;
    const int32_t* first_sem = reinterpret_cast<const int32_t*>(ins_b + 2 * (size_t)max_ins);
    for (int c = 0; c < 40; ++c) {
        h_iou_sem[c] = (float)h[80 + c];                                  // I (model.py:626)
        h_iou_sem[40 + c] = (float)(h[c] + h[40 + c] - h[80 + c]);        // U (model.py:627)
    }
    for (int c = 0; c < 80; ++c) h_iou_ins[c] = 0.f;
    for (int i = 0; i < max_ins; ++i) {                                   // model.py:633-639
        if (!ins_p[i]) continue;
        int slot = first_sem[i] - 1;
        if (slot < 0) slot += 40;                                         // Python negative index wrap
        if (slot < 0 || slot >= 40) continue;
        h_iou_ins[slot] += (float)ins_b[i];
        h_iou_ins[40 + slot] += (float)(ins_p[i] + ins_t[i] - ins_b[i]);
    }
    auto ratio = [](uint32_t a, uint32_t b) { return b ? (float)((double)a / (double)b) : NAN; };
    h_acc[0] = ratio(h[121], h[120]);
    h_acc[1] = ratio(h[122], h[120]);
    h_acc[2] = ratio(h[124], h[123]);
    h_acc[3] = ratio(h[126], h[125]);
}

}  // namespace sg

extern "C" {

size_t sg_contract_ws_bytes(int S) {
    const size_t bits = (size_t)S * (size_t)S;
    const size_t words = (bits + 31) / 32;
    const size_t nblocks = (words + kWordsPerBlock - 1) / kWordsPerBlock;
    return sg::align_up(words * 4) + sg::align_up((nblocks + 1) * 4) + 256;
}

int sg_contract_point_edges(const int64_t* d_adj, int E, const int32_t* d_seg_of_point, int N, int S,
                            int32_t* d_out_adj, int out_capacity, int32_t* d_out_count,
                            void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(E >= 0 && N >= 0 && S > 0 && d_out_count && d_ws, "sg_contract_point_edges: bad arguments");
    const size_t bits = (size_t)S * (size_t)S;
    if (bits > (1ull << 31)) return sg::fail(SG_EUNSUP, "sg_contract_point_edges: S=%d exceeds the bitmap envelope (S*S <= 2^31)", S);
    const size_t words = (bits + 31) / 32;
    const int nblocks = (int)((words + kWordsPerBlock - 1) / kWordsPerBlock);
    sg::Carver cv(d_ws, ws_bytes);
    uint32_t* bitmap = cv.take<uint32_t>(words);
    int* block_count = cv.take<int>(nblocks + 1);
    if (!cv.ok) return sg::fail(SG_ENOMEM, "sg_contract_point_edges: workspace too small (%zu < %zu)", ws_bytes, sg_contract_ws_bytes(S));
    hipStream_t st = sg::as_stream(stream);
    SG_HIP(hipMemsetAsync(bitmap, 0, words * 4, st));
    if (E > 0) {
        const int grid = std::min(sg::cdiv(E, kBlock), 2048);
        k_mark_pairs<<<grid, kBlock, 0, st>>>(d_adj, E, d_seg_of_point, N, S, bitmap);
    }
    k_count_bits<<<nblocks, kBlock, 0, st>>>(bitmap, words, block_count);
    k_scan_blocks<<<1, kBlock, 0, st>>>(block_count, nblocks, d_out_count);
    k_emit_pairs<<<nblocks, kBlock, 0, st>>>(bitmap, words, block_count, S, d_out_adj, out_capacity);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_gather_members(const int32_t* d_seg_points, const int32_t* d_seg_off, int S, const int32_t* d_order,
                      const int32_t* d_dst, const int32_t* d_cl, int32_t* d_members, int32_t* d_pos_of_point,
                      int32_t* d_cluster_of_pos, int32_t* d_slot_of_pos, void* stream) {
    SG_REQUIRE(S >= 0 && d_members, "sg_gather_members: bad arguments");
    if (S == 0) return SG_OK;
    k_gather_members<<<S, 128, 0, sg::as_stream(stream)>>>(d_seg_points, d_seg_off, d_order, d_dst, d_cl, d_members,
                                                          d_pos_of_point, d_cluster_of_pos, d_slot_of_pos);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_edge_distance(const float* d_feat, int feat_stride, int D, const int32_t* d_adj, int E, float* d_dist, void* stream) {
    SG_REQUIRE(E >= 0 && D > 0 && feat_stride >= D, "sg_edge_distance: bad arguments");
    if (E == 0) return SG_OK;
    k_edge_distance<<<sg::cdiv(E, 4), 256, 0, sg::as_stream(stream)>>>(d_feat, feat_stride, D, d_adj, E, d_dist);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_group_max_rows(const float* d_rows, int row_stride, int D, const int32_t* d_goff, const int32_t* d_gidx, int G,
                      float* d_out, int out_stride, void* stream) {
    SG_REQUIRE(G >= 0 && D > 0, "sg_group_max_rows: bad arguments");
    if (G == 0) return SG_OK;
    return sg::group_max_rows_fill(d_rows, row_stride, D, d_goff, d_gidx, G, d_out, out_stride, 0, stream);
}

int sg_segment_max(const float* d_rows, int N, int D, const int32_t* d_cluster_of_pos, float* d_out, int out_stride, int C,
                   void* stream) {
    SG_REQUIRE(D == 64, "sg_segment_max: only D == 64 rows are supported (got %d)", D);
    if (N == 0 || C == 0) return SG_OK;
    k_fill_rows<<<std::min(sg::cdiv((long long)C * 64, 256), 1024), 256, 0, sg::as_stream(stream)>>>(d_out, C, 64, out_stride, -INFINITY);
    return sg::segment_max_prefilled(d_rows, N, d_cluster_of_pos, d_out, out_stride, stream);
}

int sg_export_labels(const int32_t* d_unmap, int V, const int32_t* d_seg_of_point, int N, const int32_t* d_tables, int T,
                     int S, int32_t* d_out, void* stream) {
    SG_REQUIRE(V >= 0 && T > 0 && S > 0, "sg_export_labels: bad arguments");
    if (V == 0) return SG_OK;
    k_export<<<std::min(sg::cdiv(V, 256), 2048), 256, 0, sg::as_stream(stream)>>>(d_unmap, V, d_seg_of_point, N, d_tables, T, S, d_out);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

}  // extern "C"

namespace sg {
// sg_evaluate with the counters' host landing buffer handed in (128 + 5 max_ins words): the pipeline passes a PINNED one -- with a pageable
// destination the copy in the tail of every single-scene forward was a staged, synchronous one
int evaluate_landing(const int32_t* d_gt, const int32_t* d_sem_pred, const int32_t* d_ins_pred, int V, int max_ins, float* h_iou_sem,
                     float* h_iou_ins, float* h_acc, void* d_ws, size_t ws_bytes, void* stream, uint32_t* h_landing) {
    SG_REQUIRE(V >= 0 && max_ins >= 1 && h_iou_sem && h_iou_ins && h_acc && h_landing, "sg_evaluate: bad arguments");
    const size_t n = 128 + 5 * (size_t)max_ins;
    if (ws_bytes < n * 4) return sg::fail(SG_ENOMEM, "sg_evaluate: workspace too small");
    uint32_t* cnt = (uint32_t*)d_ws;
    hipStream_t st = sg::as_stream(stream);
    {   // counters 0, first-occurrence slots all-ones: one launch instead of two memsets
        const int n0 = 128 + 3 * max_ins, n1 = n0 + max_ins;
        k_eval_clear<<<sg::cdiv(n1, 256), 256, 0, st>>>(cnt, n0, n1);
    }
    if (V > 0) {
        const size_t dyn = max_ins <= kInsLds ? (size_t)max_ins * 16 : 0;
        k_eval_counts<<<std::min(sg::cdiv(V, 1024), 256), 256, dyn, st>>>(d_gt, d_sem_pred, d_ins_pred, V, max_ins, cnt);
    }
    k_eval_first_sem<<<sg::cdiv(max_ins, 256), 256, 0, st>>>(d_sem_pred, max_ins, cnt);
    SG_LAUNCH_CHECK();
    SG_HIP(hipMemcpyAsync(h_landing, cnt, n * 4, hipMemcpyDeviceToHost, st));
    SG_HIP(hipStreamSynchronize(st));
    sg::eval_finish(h_landing, max_ins, h_iou_sem, h_iou_ins, h_acc);
    return SG_OK;
}
}  // namespace sg

extern "C" {

size_t sg_eval_ws_bytes(int max_ins) { return sg::align_up((size_t)(128 + 5 * (size_t)std::max(max_ins, 1)) * 4); }

int sg_evaluate(const int32_t* d_gt, const int32_t* d_sem_pred, const int32_t* d_ins_pred, int V, int max_ins,
                float* h_iou_sem, float* h_iou_ins, float* h_acc, void* d_ws, size_t ws_bytes, void* stream) {
    std::vector<uint32_t> h(128 + 5 * (size_t)std::max(max_ins, 1));      // a pageable landing buffer: the copy is a staged, synchronous one
    return sg::evaluate_landing(d_gt, d_sem_pred, d_ins_pred, V, max_ins, h_iou_sem, h_iou_ins, h_acc, d_ws, ws_bytes, stream, h.data());
}

}  // extern "C"
