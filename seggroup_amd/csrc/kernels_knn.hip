// a12: combine_centralized_pointcloud (reference seggroup/model.py:429-436) and
// a6+a11: get_knn / knn, k = 20 inside each cluster (model.py:30-36, 512-522).
//
// Everything here works in MEMBER order (position = index into the layer's member array), so a
// cluster is a contiguous range and the EdgeConv gathers that follow stay cluster-local in L2.
// Work unit = a "tile": <= 256 consecutive positions of ONE cluster (descriptor arrays built on
// the host by the pipeline from the cluster sizes).
//
// kNN: brute force per cluster, one query per lane, candidates streamed through LDS in 1024-point
// chunks ([x,y,z,|p|^2] float4, broadcast reads), top-20 kept sorted in registers.  Scores use the
// reference's exact fp32 operation order; -ffp-contract=off for this file.
#include <cstdlib>

#include "sg_common.h"

namespace {

constexpr int kTile = 256;
constexpr int kChunk = 1024;

__device__ inline double block_sum_256(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kTile) void k_center_tile_sums(const float* __restrict__ data, const int32_t* __restrict__ members,
                                                            const int32_t* __restrict__ tile_lo, const int32_t* __restrict__ tile_hi,
                                                            double* __restrict__ tile_sums) {
    __shared__ double red[4];
    const int t = blockIdx.x;
    const int pos = tile_lo[t] + threadIdx.x;
    double x = 0, y = 0, z = 0;
    if (pos < tile_hi[t]) {
        const float* row = data + (size_t)members[pos] * 6;
        x = row[0]; y = row[1]; z = row[2];
    }
    x = block_sum_256(x, red);
    y = block_sum_256(y, red);
    z = block_sum_256(z, red);
    if (threadIdx.x == 0) { tile_sums[3 * (size_t)t] = x; tile_sums[3 * (size_t)t + 1] = y; tile_sums[3 * (size_t)t + 2] = z; }
}

__global__ __launch_bounds__(kTile) void k_center_write(const float* __restrict__ data, const int32_t* __restrict__ members,
                                                        const int32_t* __restrict__ cl_off, const int32_t* __restrict__ tile_cl,
                                                        const int32_t* __restrict__ tile_lo, const int32_t* __restrict__ tile_hi,
                                                        const int32_t* __restrict__ cl_tile_off, const double* __restrict__ tile_sums,
                                                        float* __restrict__ x9m, float* __restrict__ xyzw) {
    __shared__ float mean[3];
    const int t = blockIdx.x;
    const int c = tile_cl[t];
    if (threadIdx.x < 3) {
        double s = 0.0;
        for (int u = cl_tile_off[c]; u < cl_tile_off[c + 1]; ++u) s += tile_sums[3 * (size_t)u + threadIdx.x];   // fixed order
        mean[threadIdx.x] = (float)(s / (double)(cl_off[c + 1] - cl_off[c]));
    }
    __syncthreads();
    const int pos = tile_lo[t] + threadIdx.x;
    if (pos >= tile_hi[t]) return;
    const float* row = data + (size_t)members[pos] * 6;
    const float x = row[0], y = row[1], z = row[2];
    float4* o = reinterpret_cast<float4*>(x9m + (size_t)pos * 12);
    o[0] = make_float4(x, y, z, row[3]);
    o[1] = make_float4(row[4], row[5], x - mean[0], y - mean[1]);
    o[2] = make_float4(z - mean[2], 0.f, 0.f, 0.f);
    reinterpret_cast<float4*>(xyzw)[pos] = make_float4(x, y, z, (x * x + y * y) + z * z);   // torch.sum(x**2, dim=1)
}

template <int K>
__device__ inline void topk_insert(float (&bv)[K], int (&bi)[K], float s, int id) {
    if (s > bv[K - 1]) {
        float v = s;
        bool placed = false;                 // once placed, everything below shifts down (stable for ties)
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (placed || v > bv[j]) {
                placed = true;
                const float tv = bv[j]; const int ti = bi[j];
                bv[j] = v; bi[j] = id; v = tv; id = ti;
            }
        }
    }
}

// Insertion into the sorted list in PARALLEL form: every slot decides independently from two compares
// (no 20-step dependency chain): new[j] = better(x, old[j]) ? (better(x, old[j-1]) ? old[j-1] : x) : old[j].
// Total order = (score descending, index ascending), so the result does not depend on arrival order.
template <int K>
__device__ inline void topk_insert_par(float (&bv)[K], int (&bi)[K], float s, int id) {
    bool c[K];
#pragma unroll
    for (int j = 0; j < K; ++j) c[j] = s > bv[j] || (s == bv[j] && id < bi[j]);
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
        const float v = c[j - 1] ? bv[j - 1] : s;
        const int i = c[j - 1] ? bi[j - 1] : id;
        bv[j] = c[j] ? v : bv[j];
        bi[j] = c[j] ? i : bi[j];
    }
    bv[0] = c[0] ? s : bv[0];
    bi[0] = c[0] ? id : bi[0];
}

__device__ inline float knn_score4(const float4& me, const float4& p) {
    const float tt = __builtin_fmaf(me.z, p.z, __builtin_fmaf(me.y, p.y, me.x * p.x));
    const float inner = -2.0f * tt;
    return ((-p.w) - inner) - me.w;
}

// Brute-force scan with a BUFFERED top-k.  A lane's sorted 20-entry list lives in registers; inserting
// costs ~100 VALU ops and, done inline, would run whenever ANY of the 64 lanes of a wave accepts a
// candidate (i.e. almost always).  Instead a lane that sees a candidate above its (possibly stale)
// threshold only appends (score, index) to its private LDS buffer -- two predicated ds_writes -- and
// the whole wave drains the buffers together once one of them is nearly full.  Draining re-checks
// each buffered candidate against the live list in index order, so the result is identical to the
// sequential scan (same "lower index wins ties" rule).
// profiling aid (sg_debug_knn_stats): [0] candidates scanned per wave, [1] buffer appends (lanes), [2] drain
// iterations (waves), [3] segments visited, [4] segments skipped
__device__ unsigned long long g_knn_stats[8];
__device__ unsigned long long g_knn_blocktime[8192];   // dbg & 8: per-(block,wave) runtime in shader clocks

constexpr int kBuf = 16;            // buffer slots per lane
constexpr int kQuad = 4;            // candidates examined between two fullness checks

template <int K>
__global__ __launch_bounds__(kTile) void k_cluster_knn(const float4* __restrict__ xyzw, const int32_t* __restrict__ cl_off,
                                                       const int32_t* __restrict__ tile_cl, const int32_t* __restrict__ tile_lo,
                                                       const int32_t* __restrict__ tile_hi, int pos0, int32_t* __restrict__ knn) {
    __shared__ float4 cand[kChunk + kQuad];
    __shared__ float buf_s[kBuf][kTile];
    __shared__ int buf_i[kBuf][kTile];
    const int t = blockIdx.x;
    const int c = tile_cl[t];
    const int clo = cl_off[c], n = cl_off[c + 1] - clo;
    const int tid = threadIdx.x;
    const int q = tile_lo[t] + tid;
    const bool active = q < tile_hi[t];
    if (n <= K) {                                            // model.py:516-518: all members, rest stays 0 (= point 0)
        if (active) {
            int32_t* o = knn + (size_t)q * K;
#pragma unroll
            for (int j = 0; j < K; ++j) o[j] = j < n ? clo + j : pos0;
        }
        return;
    }
    const float4 me = active ? xyzw[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    float bv[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bv[j] = -INFINITY; bi[j] = 0; }
    float thr = active ? -INFINITY : INFINITY;               // idle lanes never accept
    int cnt = 0;

    auto drain = [&]() {
        int mx = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
        for (int u = 0; u < mx; ++u) {
            if (u < cnt) topk_insert<K>(bv, bi, buf_s[u][tid], buf_i[u][tid]);
        }
        cnt = 0;
        if (active) thr = bv[K - 1];
    };

    for (int base = 0; base < n; base += kChunk) {
        const int m = min(kChunk, n - base);
        __syncthreads();
        for (int i = tid; i < m + kQuad; i += kTile)         // sentinels (|p|^2 = +inf -> score -inf) pad the last quad
            cand[i] = i < m ? xyzw[clo + base + i] : make_float4(0.f, 0.f, 0.f, INFINITY);
        __syncthreads();
        for (int i = 0; i < m; i += kQuad) {
#pragma unroll
            for (int u = 0; u < kQuad; ++u) {
                const float sc = knn_score4(me, cand[i + u]);
                if (sc > thr) {
                    buf_s[cnt][tid] = sc;
                    buf_i[cnt][tid] = base + i + u;
                    ++cnt;
                }
            }
            if (__any(cnt > kBuf - kQuad)) drain();
        }
    }
    drain();
    if (active) {
        int32_t* o = knn + (size_t)q * K;
#pragma unroll
        for (int j = 0; j < K; ++j) o[j] = clo + bi[j];
    }
}

// ------------------------------------------------------------------------------------------------
// Pruned kNN.  A cluster is an ordered list of ORIGINAL over-segments, and an over-segment is a small,
// spatially compact patch, so its axis-aligned box (computed once per scene) bounds the score of
// every point in it: score <= -dmin(query, box)^2 + rounding margin.  A wave (64 queries) walks the
// cluster's segments in member order and skips a whole segment when no lane can still be beaten.
// Visited candidates arrive in ascending member index, skipped ones are STRICTLY below the k-th score,
// so the result (ties included) is identical to the brute-force scan -- tests compare the two.
// Waves are independent: candidates are staged in a wave-private LDS slab, no workgroup barrier.
// ------------------------------------------------------------------------------------------------
__global__ void k_segment_boxes(const float* __restrict__ data, const int32_t* __restrict__ seg_points,
                                const int32_t* __restrict__ seg_off, float* __restrict__ box) {
    const int s = blockIdx.x, lane = threadIdx.x;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, xx = 0.f;
    for (int i = seg_off[s] + lane; i < seg_off[s + 1]; i += 64) {
        const float* r = data + (size_t)seg_points[i] * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], r[k]); mx[k] = fmaxf(mx[k], r[k]); }
        xx = fmaxf(xx, (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
        xx = fmaxf(xx, __shfl_xor(xx, o));
    }
    if (lane == 0) {
        float* b = box + (size_t)s * 8;
        b[0] = mn[0]; b[1] = mn[1]; b[2] = mn[2]; b[3] = mx[0]; b[4] = mx[1]; b[5] = mx[2]; b[6] = xx; b[7] = 0.f;
    }
}

// ---- pruned kernel, v4: one workgroup = ONE tile of <= 64 queries x FOUR candidate slices -------------------
// All four waves hold the same 64 queries; wave w walks every 4th segment of the cluster (rotation starting
// at the queries' own segment, so the first segments visited are the likeliest neighbours).  This cuts the
// longest wave -- which bounds the kernel, every tile being resident at once -- by 4x.  Each wave keeps its
// own sorted top-20 of its slice; the lanes publish their 20th-best key in LDS and prune / filter with the
// BEST published one (the global 20th best can only be better than any slice's).  A 4-way merge of the four
// sorted lists (LDS, wave 0) produces the row.
// Keys: 64-bit (order-preserving uint of the fp32 score << 32 | ~member index): one unsigned compare
// implements the total order (score descending, index ascending), independent of arrival order.
constexpr int kSlices = 4;
constexpr int kBuf4 = 20;           // buffer slots per lane; also sizes the merge area that aliases the buffers

__device__ inline unsigned long long knn_key(float score, int idx) {
    const unsigned int u = __float_as_uint(score);
    const unsigned int o = u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);       // monotone float -> uint
    return ((unsigned long long)o << 32) | (unsigned int)(0xffffffffu - (unsigned int)idx);
}

template <int K>
__device__ inline void key_insert_par(unsigned long long (&kv)[K], unsigned long long x) {
    bool c[K];
#pragma unroll
    for (int j = 0; j < K; ++j) c[j] = x > kv[j];
#pragma unroll
    for (int j = K - 1; j > 0; --j) kv[j] = c[j] ? (c[j - 1] ? kv[j - 1] : x) : kv[j];
    kv[0] = c[0] ? x : kv[0];
}

template <int K>
__global__ __launch_bounds__(64 * kSlices) void k_cluster_knn_pruned(
    const float4* __restrict__ xyzw, const int32_t* __restrict__ cl_off, const int32_t* __restrict__ tile_cl,
    const int32_t* __restrict__ tile_lo, const int32_t* __restrict__ tile_hi, const int32_t* __restrict__ cl_seg_off,
    const int32_t* __restrict__ order, const int32_t* __restrict__ dst, const int32_t* __restrict__ seg_off,
    const float* __restrict__ segbox, const int32_t* __restrict__ slot_of_pos, int pos0, int32_t* __restrict__ knn, int dbg_arg) {
    static_assert(kBuf4 >= K, "the merge area aliases the append buffers");
#ifdef SG_KNN_PROFILE
    const int dbg = dbg_arg;
#else
    constexpr int dbg = 0;     // release builds: the work counters below fold away (make PROFILE=1 keeps them)
    (void)dbg_arg;
#endif
    __shared__ float4 slab[kSlices][64 + kQuad];
    __shared__ unsigned long long buf[kBuf4][64 * kSlices];      // append buffers; later lists[slice][K][64]
    __shared__ unsigned int thr_pub[kSlices][64];             // published score part only: 32-bit LDS stores cannot tear
    const int t = blockIdx.x;
    const int c = tile_cl[t];
    const int clo = cl_off[c], n = cl_off[c + 1] - clo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = tile_lo[t] + lane;
    const bool active = q < tile_hi[t];
    if (n <= K) {                                            // model.py:516-518 (block-uniform)
        if (active && wave == 0) {
            int32_t* o = knn + (size_t)q * K;
#pragma unroll
            for (int j = 0; j < K; ++j) o[j] = j < n ? clo + j : pos0;
        }
        return;
    }
    const unsigned long long t_begin = (dbg & 8) ? __builtin_readcyclecounter() : 0ull;
    const float4 me = active ? xyzw[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned long long kv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) kv[j] = 0ull;
    unsigned long long thr = active ? 0ull : ~0ull;          // idle lanes never accept
    thr_pub[wave][lane] = (unsigned int)(thr >> 32);
    int cnt = 0;
    float4* cw = slab[wave];
    __syncthreads();

    // Best published threshold, as a key with the index part cleared (= the weakest key of that score: ties at
    // the threshold score are still accepted and settled exactly by the sorted lists).  Stale reads are fine:
    // thresholds only rise.
    auto best_thr = [&]() {
        unsigned int b = 0u;
#pragma unroll
        for (int w = 0; w < kSlices; ++w) b = max(b, thr_pub[w][lane]);
        const unsigned long long pub = (unsigned long long)b << 32;
        return pub > thr ? pub : thr;
    };
    auto drain = [&]() {
        int mxc = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mxc = max(mxc, __shfl_xor(mxc, o));
        if (dbg & 4) {
            unsigned long long tot = cnt;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
            if (lane == 0) { atomicAdd(&g_knn_stats[1], tot); atomicAdd(&g_knn_stats[2], (unsigned long long)mxc); }
        }
        for (int u = 0; u < mxc; ++u) key_insert_par<K>(kv, u < cnt ? buf[u][tid] : 0ull);
        cnt = 0;
        if (active) { thr = kv[K - 1]; thr_pub[wave][lane] = (unsigned int)(thr >> 32); }
    };

    const int so0 = cl_seg_off[c], nslots = cl_seg_off[c + 1] - so0;
    const int own = slot_of_pos[tile_lo[t]] - so0;

    auto slot_of = [&](int r) {                               // r-th segment of the rotation that starts at `own`
        int sl = own + r;
        if (sl >= nslots) sl -= nslots;
        return so0 + sl;
    };
    // scan points [p0, p1) of one segment slot (sg = its original segment id, prefetched by the caller)
    auto scan_range = [&](int slot, int sg, int p0, int p1) {
        const float* bx = segbox + (size_t)sg * 8;
        const unsigned long long use = best_thr();
        // upper bound of any score in this segment: -dmin^2 (shrunk) + 16 eps (|q|^2 + max|p|^2)
        const float dx = fmaxf(fmaxf(bx[0] - me.x, me.x - bx[3]), 0.f);
        const float dy = fmaxf(fmaxf(bx[1] - me.y, me.y - bx[4]), 0.f);
        const float dz = fmaxf(fmaxf(bx[2] - me.z, me.z - bx[5]), 0.f);
        const float ub = -((dx * dx + dy * dy) + dz * dz) * 0.999999f + 9.6e-7f * (me.w + bx[6]);
        if (!__any(knn_key(ub, 0) >= use)) return;             // index 0 = the best possible tie-break
        const int base = dst[slot] - clo;                     // local member index of the segment's first point
        float4 nxt = (p0 + lane < p1) ? xyzw[clo + base + p0 + lane] : make_float4(0.f, 0.f, 0.f, INFINITY);
        for (int sub = p0; sub < p1; sub += 64) {
            const int mm = min(64, p1 - sub);
            __builtin_amdgcn_wave_barrier();
            cw[lane] = nxt;
            if (lane < kQuad) cw[64 + lane] = make_float4(0.f, 0.f, 0.f, INFINITY);
            __builtin_amdgcn_wave_barrier();
            if (sub + 64 < p1)                                // next chunk's load flies while this one is scored
                nxt = (sub + 64 + lane < p1) ? xyzw[clo + base + sub + 64 + lane] : make_float4(0.f, 0.f, 0.f, INFINITY);
            const unsigned long long use2 = best_thr();       // thresholds rise while the segment is scanned
            for (int i = 0; i < mm; i += kQuad) {
#pragma unroll
                for (int u = 0; u < kQuad; ++u) {
                    const unsigned long long key = knn_key(knn_score4(me, cw[i + u]), base + sub + i + u);
                    if (key > use2 && key > thr) {
                        buf[cnt][tid] = key;
                        ++cnt;
                    }
                }
                if (__any(cnt > kBuf4 - kQuad)) drain();
            }
        }
    };

    // phase A: the queries' own segment (the likeliest neighbours), a quarter of its points per wave; a 4-way
    // merge of the four partial lists gives every lane its exact 20th best so far, published as the
    // threshold all slices filter and prune with -- without it every slice would first have to collect (and
    // insert) 20 candidates of its own.
    {
        const int slot = slot_of(0);
        const int sg = order[slot];
        const int m = seg_off[sg + 1] - seg_off[sg];
        const int per = (m + kSlices - 1) / kSlices;
        scan_range(slot, sg, min(m, wave * per), min(m, (wave + 1) * per));
        drain();
    }
    const unsigned long long t_a = (dbg & 8) ? __builtin_readcyclecounter() : 0ull;
    __syncthreads();
    {
        unsigned long long* lists = &buf[0][0];               // [slice][K][64]; the append buffers are empty now
#pragma unroll
        for (int j = 0; j < K; ++j) lists[((size_t)wave * K + j) * 64 + lane] = kv[j];
        __syncthreads();
        if (wave == 0) {
            int p[kSlices];
            unsigned long long h[kSlices], bk = 0ull;
#pragma unroll
            for (int w = 0; w < kSlices; ++w) { p[w] = 0; h[w] = lists[((size_t)w * K) * 64 + lane]; }
            for (int j = 0; j < K; ++j) {
                int bw = 0;
                bk = h[0];
#pragma unroll
                for (int w = 1; w < kSlices; ++w) if (h[w] > bk) { bk = h[w]; bw = w; }
#pragma unroll
                for (int w = 0; w < kSlices; ++w)
                    if (w == bw) { ++p[w]; h[w] = p[w] < K ? lists[((size_t)w * K + p[w]) * 64 + lane] : 0ull; }
            }
            if (active) thr_pub[0][lane] = max(thr_pub[0][lane], (unsigned int)(bk >> 32));   // 20th best of the union (0 if < 20)
        }
        __syncthreads();
    }
    const unsigned long long t_a2 = (dbg & 8) ? __builtin_readcyclecounter() : 0ull;
    // phase B: the remaining segments, every 4th one per wave, all filtered by the best published threshold
    {
        int r = 1 + wave;
        int sg_next = r < nslots ? order[slot_of(r)] : 0;
        for (; r < nslots; r += kSlices) {
            const int slot = slot_of(r), sg = sg_next;
            if (r + kSlices < nslots) sg_next = order[slot_of(r + kSlices)];     // prefetch the next descriptor
            scan_range(slot, sg, 0, seg_off[sg + 1] - seg_off[sg]);
        }
    }
    drain();
    const unsigned long long t_b = (dbg & 8) ? __builtin_readcyclecounter() : 0ull;
    __syncthreads();                                          // every wave is done with its append buffer
    const unsigned long long t_b2 = (dbg & 8) ? __builtin_readcyclecounter() : 0ull;
    unsigned long long* lists = &buf[0][0];                   // [slice][K][64]
#pragma unroll
    for (int j = 0; j < K; ++j) lists[((size_t)wave * K + j) * 64 + lane] = kv[j];
    __syncthreads();
    if (wave == 0 && active) {
        int p[kSlices];
        unsigned long long h[kSlices];
#pragma unroll
        for (int w = 0; w < kSlices; ++w) { p[w] = 0; h[w] = lists[((size_t)w * K) * 64 + lane]; }
        int32_t* o = knn + (size_t)q * K;
        for (int j = 0; j < K; ++j) {
            int bw = 0;
            unsigned long long bk = h[0];
#pragma unroll
            for (int w = 1; w < kSlices; ++w) if (h[w] > bk) { bk = h[w]; bw = w; }
            o[j] = (dbg & 3) ? clo + j : clo + (int)(0xffffffffu - (unsigned int)(bk & 0xffffffffu));
#pragma unroll
            for (int w = 0; w < kSlices; ++w)
                if (w == bw) { ++p[w]; h[w] = p[w] < K ? lists[((size_t)w * K + p[w]) * 64 + lane] : 0ull; }
        }
    }
    if ((dbg & 8) && lane == 0) {
        // [0..3]: wave-0 phase A, barrier wait after A (all waves), phase B work (all waves), barrier wait after B
        if (wave == 0) atomicAdd(&g_knn_stats[0], t_a - t_begin);
        atomicAdd(&g_knn_stats[1], t_a2 - t_a);
        atomicAdd(&g_knn_stats[2], t_b - t_a2);
        atomicAdd(&g_knn_stats[3], t_b2 - t_b);
        atomicAdd(&g_knn_stats[4], __builtin_readcyclecounter() - t_b2);
        atomicAdd(&g_knn_stats[5], 1ull);
        if (t * 4 + wave < 8192) g_knn_blocktime[(t * 4 + wave) & 8191] = __builtin_readcyclecounter() - t_begin;
    }
}

}  // namespace

extern "C" {

size_t sg_center_ws_bytes(int T, int C) { (void)C; return sg::align_up((size_t)std::max(T, 1) * 3 * 8); }

int sg_center_clusters(const float* d_data, int N, const int32_t* d_members, const int32_t* d_cl_off, int C,
                       const int32_t* d_tile_cl, const int32_t* d_tile_lo, const int32_t* d_tile_hi, int T,
                       const int32_t* d_cl_tile_off, float* d_x9m, float* d_xyzw, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(N >= 0 && C >= 0 && T >= 0 && d_x9m && d_xyzw, "sg_center_clusters: bad arguments");
    if (T == 0) return SG_OK;
    if (ws_bytes < (size_t)T * 24) return sg::fail(SG_ENOMEM, "sg_center_clusters: workspace too small");
    hipStream_t st = sg::as_stream(stream);
    double* sums = (double*)d_ws;
    k_center_tile_sums<<<T, kTile, 0, st>>>(d_data, d_members, d_tile_lo, d_tile_hi, sums);
    k_center_write<<<T, kTile, 0, st>>>(d_data, d_members, d_cl_off, d_tile_cl, d_tile_lo, d_tile_hi, d_cl_tile_off, sums, d_x9m, d_xyzw);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_cluster_knn(const float* d_xyzw, int N, const int32_t* d_cl_off, const int32_t* d_tile_cl, const int32_t* d_tile_lo,
                   const int32_t* d_tile_hi, int T, int k, int pos0, int32_t* d_knn, void* stream) {
    SG_REQUIRE(N >= 0 && T >= 0 && d_knn, "sg_cluster_knn: bad arguments");
    if (k != 20) return sg::fail(SG_EUNSUP, "sg_cluster_knn: only k == 20 is built (model.py:788,829), got %d", k);
    if (T == 0) return SG_OK;
    k_cluster_knn<20><<<T, kTile, 0, sg::as_stream(stream)>>>(reinterpret_cast<const float4*>(d_xyzw), d_cl_off, d_tile_cl, d_tile_lo,
                                                             d_tile_hi, pos0, d_knn);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

#ifdef SG_KNN_PROFILE
static int g_knn_dbg = getenv("SG_KNN_DEBUG") ? atoi(getenv("SG_KNN_DEBUG")) : 0;   // profiling knob: 1 no inserts, 2 no scan, 4 stats
#else
static constexpr int g_knn_dbg = 0;
#endif

int sg_segment_boxes(const float* d_data, const int32_t* d_seg_points, const int32_t* d_seg_off, int S, float* d_box, void* stream) {
    SG_REQUIRE(S >= 0 && d_box, "sg_segment_boxes: bad arguments");
    if (S == 0) return SG_OK;
    k_segment_boxes<<<S, 64, 0, sg::as_stream(stream)>>>(d_data, d_seg_points, d_seg_off, d_box);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_cluster_knn_pruned(const float* d_xyzw, int N, const int32_t* d_cl_off, const int32_t* d_tile_cl, const int32_t* d_tile_lo,
                          const int32_t* d_tile_hi, int T, const int32_t* d_cl_seg_off, const int32_t* d_order, const int32_t* d_dst,
                          const int32_t* d_seg_off, const float* d_segbox, const int32_t* d_slot_of_pos, int k, int pos0,
                          int32_t* d_knn, void* stream) {
    SG_REQUIRE(N >= 0 && T >= 0 && d_knn && d_segbox && d_slot_of_pos, "sg_cluster_knn_pruned: bad arguments");
    if (k != 20) return sg::fail(SG_EUNSUP, "sg_cluster_knn_pruned: only k == 20 is built (model.py:788,829), got %d", k);
    if (T == 0) return SG_OK;
    k_cluster_knn_pruned<20><<<T, 64 * kSlices, 0, sg::as_stream(stream)>>>(reinterpret_cast<const float4*>(d_xyzw), d_cl_off, d_tile_cl,
                                                                    d_tile_lo, d_tile_hi, d_cl_seg_off, d_order, d_dst, d_seg_off,
                                                                    d_segbox, d_slot_of_pos, pos0, d_knn, g_knn_dbg);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

#ifdef SG_KNN_PROFILE
int sg_debug_knn_blocktimes(unsigned long long* h_out, int count) {
    SG_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_knn_blocktime), sizeof(unsigned long long) * std::min(count, 8192)));
    unsigned long long* z = new unsigned long long[8192]();
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_knn_blocktime), z, sizeof(unsigned long long) * 8192);
    delete[] z;
    SG_HIP(e);
    return SG_OK;
}

// undocumented profiling aid: copies (and optionally clears) the pruned-kNN work counters
int sg_debug_knn_stats(unsigned long long* h_out, int reset) {
    SG_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_knn_stats), sizeof(unsigned long long) * 8));
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        SG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_knn_stats), z, sizeof z));
    }
    return SG_OK;
}

#endif

}  // extern "C"
