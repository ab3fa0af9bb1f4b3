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

}  // extern "C"
