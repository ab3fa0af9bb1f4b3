// raw scan -> hot-path inputs (reference seggroup/dataset/scannet/util.py).
//
//   sg_prep_sample_points   generate_pointcloud_pth 633-693 without the file I/O: gather + colour centring, the
//                           unmapper (last occurrence of a vertex in the mapper wins, 687-689), and for vertices that
//                           were not sampled the nearest sampled point (get_unmapper 538-550)
//   sg_nearest_point        get_unmapper / cal_pairwise_distance 530-550
//   sg_pointcloud_adjacency get_adj_from_pointcloud 814-834 (optional in the reference: nothing calls it): kNN graph of the cloud
//   sg_mesh_adjacency       get_adj_from_mesh 771-792: per-row sorted, lexicographically unique edge lists, raw and resampled
//   sg_segment_lists        generate_seg_labels_and_ds_set 174-220: compacted segment ids + member lists
//
// Sorting (round 4): csrc/sort_device.h -- a stable LSD radix sort over exactly the bits that carry an id (an edge key lo << 32 | hi of
// ids below 2^b is sorted by its two b + 1-bit fields: four 11-bit passes at b <= 20, where the library sort this file used before ran
// eight over all 64 bits), exclusive scans and an adjacent-unique compaction.
// All of it is index / byte work (HBM- and sort-bound) except the nearest-point search, a brute-force scan in the
// reference's exact fp32 formula  s_ij = ((-xx_i) - (-2 x_i.y_j)) - yy_j  (argmax, lowest j on ties): at ScanNet's worst
// case (380k unsampled vertices x 150k samples = 5.7e10 pairs) that is ~4.6e11 VALU lane-ops, ~15 ms on MI355X, so a
// spatial index is not worth its exactness proof.
#include "sg_common.h"
#include "sort_device.h"

namespace {

constexpr int kTile = 256;      // candidates staged per LDS tile = queries per block

// score of candidate c = (x, y, z, yy) for the query (qx, qy, qz) with squared norm qq, in the reference's order
__device__ inline float pair_score(float qx, float qy, float qz, float qq, const float4& c) {
    const float tt = __builtin_fmaf(qz, c.z, __builtin_fmaf(qy, c.y, qx * c.x));      // MKL's K=3 dot product
    const float inner = -2.0f * tt;
    return ((-qq) - inner) - c.w;
}

// [n,3] (row stride `stride` floats) -> float4 (x, y, z, (x*x + y*y) + z*z): torch.sum(y**2, dim=1)
__global__ void k_pack_xyzw(const float* __restrict__ p, int stride, int n, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = p[(size_t)i * stride], y = p[(size_t)i * stride + 1], z = p[(size_t)i * stride + 2];
    out[i] = make_float4(x, y, z, (x * x + y * y) + z * z);
}

// queries qid[0..U) (indices into qxyz rows, or the identity when qid == nullptr) against all N candidates.
// One query per thread; candidates stream through LDS in ascending order, so `>` keeps the lowest index of a tie.
__global__ __launch_bounds__(kTile) void k_nearest(const float* __restrict__ qxyz, int qstride, const int32_t* __restrict__ qid, int U,
                                                   const float4* __restrict__ cand, int N, int64_t* __restrict__ out, int scatter) {
    __shared__ float4 tile[kTile];
    const int u = blockIdx.x * kTile + threadIdx.x;
    const bool live = u < U;
    const int row = live ? (qid ? qid[u] : u) : 0;
    const float qx = qxyz[(size_t)row * qstride], qy = qxyz[(size_t)row * qstride + 1], qz = qxyz[(size_t)row * qstride + 2];
    const float qq = (qx * qx + qy * qy) + qz * qz;
    float best = -INFINITY;
    int arg = 0;
    for (int c0 = 0; c0 < N; c0 += kTile) {
        __syncthreads();
        tile[threadIdx.x] = c0 + threadIdx.x < N ? cand[c0 + threadIdx.x] : make_float4(0.f, 0.f, 0.f, INFINITY);
        __syncthreads();
        const int m = min(kTile, N - c0);
        int j = 0;
        for (; j + 4 <= m; j += 4) {
            const float s0 = pair_score(qx, qy, qz, qq, tile[j]), s1 = pair_score(qx, qy, qz, qq, tile[j + 1]);
            const float s2 = pair_score(qx, qy, qz, qq, tile[j + 2]), s3 = pair_score(qx, qy, qz, qq, tile[j + 3]);
            if (s0 > best) { best = s0; arg = c0 + j; }
            if (s1 > best) { best = s1; arg = c0 + j + 1; }
            if (s2 > best) { best = s2; arg = c0 + j + 2; }
            if (s3 > best) { best = s3; arg = c0 + j + 3; }
        }
        for (; j < m; ++j) {
            const float s = pair_score(qx, qy, qz, qq, tile[j]);
            if (s > best) { best = s; arg = c0 + j; }
        }
    }
    if (live) out[scatter ? row : u] = arg;
}

// get_adj_from_pointcloud (util.py:814-834): the KK = k + 1 best-scoring candidates of every point against the WHOLE cloud in the same
// exact formula (topk keeps descending score; equal scores: lower index first), then the k pairs (i, j_t), t = 1..k -- the top entry,
// normally the point itself, is dropped like the reference's `[:, 1:]` -- as sorted pair keys lo << 32 | hi.
template <int KK>
__global__ __launch_bounds__(kTile) void k_nearest_k(const float4* __restrict__ cand, int N, unsigned long long* __restrict__ keys) {
    __shared__ float4 tile[kTile];
    const int u = blockIdx.x * kTile + threadIdx.x;
    const bool live = u < N;
    const float4 me = cand[live ? u : 0];
    float bs[KK];
    int bi[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) { bs[t] = -INFINITY; bi[t] = 0x7fffffff; }
    for (int c0 = 0; c0 < N; c0 += kTile) {
        __syncthreads();
        tile[threadIdx.x] = c0 + threadIdx.x < N ? cand[c0 + threadIdx.x] : make_float4(0.f, 0.f, 0.f, INFINITY);
        __syncthreads();
        const int m = min(kTile, N - c0);
        for (int j = 0; j < m; ++j) {
            const float sc = pair_score(me.x, me.y, me.z, me.w, tile[j]);
            if (sc > bs[KK - 1]) {                             // candidates arrive in ascending index: `>` keeps the earlier one of a tie
                float v = sc;
                int id = c0 + j;
                bool placed = false;
#pragma unroll
                for (int t = 0; t < KK; ++t) {
                    if (placed || v > bs[t]) {
                        placed = true;
                        const float tv = bs[t]; const int ti = bi[t];
                        bs[t] = v; bi[t] = id; v = tv; id = ti;
                    }
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int t = 1; t < KK; ++t) {
            const int j = bi[t];
            keys[(size_t)u * (KK - 1) + (t - 1)] = ((unsigned long long)(unsigned)min(u, j) << 32) | (unsigned)max(u, j);
        }
    }
}

// pcl[i] = [xyz[m], rgb[m] / 127.5 - 1 evaluated in double, rounded to fp32]; unmap[m] = max i (last occurrence)
__global__ void k_sample_gather(const float* __restrict__ xyz, const uint8_t* __restrict__ rgb, const int64_t* __restrict__ mapper,
                                int Np, float* __restrict__ pcl, int32_t* __restrict__ last) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Np) return;
    const int m = (int)mapper[i];
    float* o = pcl + (size_t)i * 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = xyz[(size_t)m * 3 + k];
        o[3 + k] = (float)((double)rgb[(size_t)m * 3 + k] / 127.5 - 1.0);
    }
    atomicMax(&last[m], i);
}

__global__ void k_fill_i32(int32_t* p, int n, int32_t v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void k_unmap_finish(const int32_t* __restrict__ last, int V, int64_t* __restrict__ unmap, int32_t* __restrict__ missing,
                               int32_t* __restrict__ n_missing) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    unmap[v] = last[v];
    if (last[v] < 0) missing[atomicAdd(n_missing, 1)] = v;           // order is irrelevant: every entry owns its output slot
}

// edges (0,1), (0,2), (1,2) of every face; zero-length edges (util.py:783) and, for the resampled list, the same rows
// mapped through `unmap` -- packed as lo << 32 | hi; dropped rows become ~0 and sort to the end
__global__ void k_face_edges(const int32_t* __restrict__ faces, int F, const int64_t* __restrict__ unmap,
                             unsigned long long* __restrict__ raw, unsigned long long* __restrict__ res, long long id_limit, int* __restrict__ too_large) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int v[3] = {faces[(size_t)f * 3], faces[(size_t)f * 3 + 1], faces[(size_t)f * 3 + 2]};
    const int pa[3] = {0, 0, 1}, pb[3] = {1, 2, 2};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int a = v[pa[e]], b = v[pb[e]];
        const bool keep = a != b;
        raw[(size_t)f * 3 + e] = keep ? ((unsigned long long)(unsigned)min(a, b) << 32) | (unsigned)max(a, b) : ~0ull;
        if (res) {
            const long long ua = keep ? unmap[a] : 0, ub = keep ? unmap[b] : 0;
            if (ua >= id_limit || ub >= id_limit || ua < 0 || ub < 0) *too_large = 1;           // outside the bit range the sort covers
            res[(size_t)f * 3 + e] = keep ? ((unsigned long long)(unsigned)min(ua, ub) << 32) | (unsigned)max(ua, ub) : ~0ull;
        }
    }
}

__global__ void k_unpack_edges(const unsigned long long* __restrict__ keys, int n, int64_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[(size_t)i * 2] = (int64_t)(keys[i] >> 32);
    out[(size_t)i * 2 + 1] = (int64_t)(keys[i] & 0xffffffffull);
}

// raw_label[v] = rank of seg_indices[v] among the sorted unique ids (np.unique + seg_remapper, util.py:181-186)
__global__ void k_rank_labels(const int32_t* __restrict__ seg, int V, const int32_t* __restrict__ uniq, const int32_t* __restrict__ n_uniq,
                              int32_t* __restrict__ raw_label) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int x = seg[v];
    int lo = 0, hi = *n_uniq;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (uniq[mid] < x) lo = mid + 1; else hi = mid;
    }
    raw_label[v] = lo;
}

__global__ void k_sampled_labels(const int32_t* __restrict__ raw_label, const int64_t* __restrict__ mapper, int Np,
                                 int32_t* __restrict__ key, int32_t* __restrict__ val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Np) return;
    key[i] = raw_label[mapper[i]];
    val[i] = i;
}

int bits_for(long long n) {
    int b = 1;
    while ((1ll << b) < n) ++b;
    return b;
}

using EdgeLists = sgsort::Lists<unsigned long long, int>;

// sorts `nl` lists of n pair keys (lo << 32 | hi, ids < 2^idbits; dropped rows = ~0, which must come out last: one bit more than the
// ids carry is sorted per field) and leaves each list's distinct keys in out[l], their number in d_count[l].  a / b: the keys and a
// second buffer of the same size per list; hist: sgsort::hist_ints(n) ints per list; scratch: sgsort::unique_ints(n) ints.
int sort_unique_edges(unsigned long long* const* a, unsigned long long* const* b, unsigned long long* const* out, int nl, int n, int idbits,
                      int* const* hist, int* scratch, int* d_count, hipStream_t st) {
    EdgeLists L{};
    for (int l = 0; l < nl; ++l) { L.kin[l] = a[l]; L.kout[l] = b[l]; L.vin[l] = nullptr; L.vout[l] = nullptr; L.hist[l] = hist[l]; L.n[l] = n; }
    const int field = std::min(32, idbits + 1);
    // LSD: the hi field (bits 0 ..), then the lo field (bits 32 ..)
    sgsort::radix_sort<unsigned long long, int, false>(L, nl, 0, field, st);
    sgsort::radix_sort<unsigned long long, int, false>(L, nl, 32, field, st);
    for (int l = 0; l < nl; ++l) {
        if (out[l] == L.kin[l]) return sg::fail(SG_EINVAL, "sort_unique_edges: the output aliases the sorted keys");
        sgsort::unique_sorted<unsigned long long>(L.kin[l], n, out[l], nullptr, d_count + l, scratch, st);
    }
    return SG_OK;
}

}  // namespace

extern "C" {

size_t sg_nearest_point_ws_bytes(int N) { return sg::align_up((size_t)std::max(N, 1) * 16); }

int sg_nearest_point(const float* d_x, int U, const float* d_y, int y_stride, int N, int64_t* d_idx, void* d_ws, size_t ws_bytes,
                     void* stream) {
    SG_REQUIRE(U >= 0 && N > 0 && y_stride >= 3 && d_idx && d_ws, "sg_nearest_point: bad arguments");
    if (ws_bytes < sg_nearest_point_ws_bytes(N)) return sg::fail(SG_ENOMEM, "sg_nearest_point: workspace too small");
    if (U == 0) return SG_OK;
    hipStream_t st = sg::as_stream(stream);
    float4* cand = reinterpret_cast<float4*>(d_ws);
    k_pack_xyzw<<<sg::cdiv(N, 256), 256, 0, st>>>(d_y, y_stride, N, cand);
    k_nearest<<<sg::cdiv(U, kTile), kTile, 0, st>>>(d_x, 3, nullptr, U, cand, N, d_idx, 0);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

size_t sg_prep_sample_ws_bytes(int V, int Np) {
    return sg::align_up((size_t)std::max(Np, 1) * 16) + 2 * sg::align_up((size_t)std::max(V, 1) * 4) + 256;
}

int sg_prep_sample_points(const float* d_xyz, const uint8_t* d_rgb, int V, const int64_t* d_mapper, int Np, float* d_pcl,
                          int64_t* d_unmap, int* h_unsampled, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(V > 0 && Np > 0 && d_xyz && d_rgb && d_mapper && d_pcl && d_unmap && d_ws, "sg_prep_sample_points: bad arguments");
    sg::Carver cv(d_ws, ws_bytes);
    float4* cand = cv.take<float4>(Np);
    int32_t* last = cv.take<int32_t>(V);
    int32_t* missing = cv.take<int32_t>(V);
    int32_t* n_missing = cv.take<int32_t>(1);
    if (!cv.ok) return sg::fail(SG_ENOMEM, "sg_prep_sample_points: workspace too small (%zu < %zu)", ws_bytes, sg_prep_sample_ws_bytes(V, Np));
    hipStream_t st = sg::as_stream(stream);
    k_fill_i32<<<sg::cdiv(V, 256), 256, 0, st>>>(last, V, -1);
    SG_HIP(hipMemsetAsync(n_missing, 0, 4, st));
    k_sample_gather<<<sg::cdiv(Np, 256), 256, 0, st>>>(d_xyz, d_rgb, d_mapper, Np, d_pcl, last);
    k_unmap_finish<<<sg::cdiv(V, 256), 256, 0, st>>>(last, V, d_unmap, missing, n_missing);
    int nm = 0;
    SG_HIP(hipMemcpyAsync(&nm, n_missing, 4, hipMemcpyDeviceToHost, st));
    SG_HIP(hipStreamSynchronize(st));
    if (h_unsampled) *h_unsampled = nm;
    if (nm > 0) {
        k_pack_xyzw<<<sg::cdiv(Np, 256), 256, 0, st>>>(d_pcl, 6, Np, cand);
        k_nearest<<<sg::cdiv(nm, kTile), kTile, 0, st>>>(d_xyz, 3, missing, nm, cand, Np, d_unmap, 1);
    }
    SG_LAUNCH_CHECK();
    return SG_OK;
}

size_t sg_mesh_adjacency_ws_bytes(int F) {
    const size_t n = (size_t)std::max(F, 1) * 3;
    return 6 * sg::align_up(n * 8) + 2 * sg::align_up(sgsort::hist_ints((long long)n) * 4) + sg::align_up(sgsort::unique_ints((long long)n) * 4) + 512;
}

// d_adj_raw / d_adj_res: room for [3F,2] int64 each; the row counts come back through h_n_raw / h_n_res
int sg_mesh_adjacency(const int32_t* d_faces, int F, const int64_t* d_unmap, int V, int64_t* d_adj_raw, int* h_n_raw, int64_t* d_adj_res,
                      int* h_n_res, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(F >= 0 && V > 0 && d_adj_raw && h_n_raw && d_ws && (!d_adj_res || (d_unmap && h_n_res)), "sg_mesh_adjacency: bad arguments");
    *h_n_raw = 0;
    if (h_n_res) *h_n_res = 0;
    if (F == 0) return SG_OK;
    const int n = 3 * F;
    const bool res = d_adj_res != nullptr;
    sg::Carver cv(d_ws, ws_bytes);
    unsigned long long* k0 = cv.take<unsigned long long>(n);
    unsigned long long* k1 = cv.take<unsigned long long>(n);
    unsigned long long* k2 = cv.take<unsigned long long>(n);
    unsigned long long* r0 = cv.take<unsigned long long>(n);
    unsigned long long* r1 = cv.take<unsigned long long>(n);
    unsigned long long* r2 = cv.take<unsigned long long>(n);
    int* h0 = cv.take<int>(sgsort::hist_ints(n));
    int* h1 = cv.take<int>(sgsort::hist_ints(n));
    int* scratch = cv.take<int>(sgsort::unique_ints(n));
    int* d_count = cv.take<int>(4);
    if (!cv.ok) return sg::fail(SG_ENOMEM, "sg_mesh_adjacency: workspace too small (%zu < %zu)", ws_bytes, sg_mesh_adjacency_ws_bytes(F));
    hipStream_t st = sg::as_stream(stream);
    // ids below 2^idbits: the raw vertex ids by V; the resampled ids are checked by the kernel (sampled clouds stay below 2^20 points)
    const int idbits = std::max(bits_for((long long)V + 1), 20);
    SG_HIP(hipMemsetAsync(d_count, 0, 16, st));
    k_face_edges<<<sg::cdiv(F, 256), 256, 0, st>>>(d_faces, F, d_unmap, k0, res ? r0 : nullptr, 1ll << idbits, d_count + 2);
    unsigned long long* a[2] = {k0, r0};
    unsigned long long* b[2] = {k1, r1};
    unsigned long long* o[2] = {k2, r2};
    int* hh[2] = {h0, h1};
    const int rc = sort_unique_edges(a, b, o, res ? 2 : 1, n, idbits, hh, scratch, d_count, st);
    if (rc < 0) return rc;
    int cnt[3] = {0, 0, 0};
    unsigned long long tail_[2] = {0, 0};
    SG_HIP(hipMemcpyAsync(cnt, d_count, 12, hipMemcpyDeviceToHost, st));
    SG_HIP(hipStreamSynchronize(st));
    if (cnt[2]) return sg::fail(SG_EUNSUP, "sg_mesh_adjacency: a resampled vertex id does not fit %d bits", idbits);
    // the dropped rows, if any, collapsed into one trailing ~0 key
    SG_HIP(hipMemcpyAsync(&tail_[0], k2 + cnt[0] - 1, 8, hipMemcpyDeviceToHost, st));
    if (res) SG_HIP(hipMemcpyAsync(&tail_[1], r2 + cnt[1] - 1, 8, hipMemcpyDeviceToHost, st));
    SG_HIP(hipStreamSynchronize(st));
    if (tail_[0] == ~0ull) --cnt[0];
    if (res && tail_[1] == ~0ull) --cnt[1];
    if (cnt[0] > 0) k_unpack_edges<<<sg::cdiv(cnt[0], 256), 256, 0, st>>>(k2, cnt[0], d_adj_raw);
    if (res && cnt[1] > 0) k_unpack_edges<<<sg::cdiv(cnt[1], 256), 256, 0, st>>>(r2, cnt[1], d_adj_res);
    *h_n_raw = cnt[0];
    if (res) *h_n_res = cnt[1];
    SG_LAUNCH_CHECK();
    return SG_OK;
}

size_t sg_pointcloud_adjacency_ws_bytes(int N, int k) {
    const size_t n = (size_t)std::max(N, 1) * std::max(k, 1);
    return 3 * sg::align_up(n * 8) + sg::align_up(sgsort::hist_ints((long long)n) * 4) + sg::align_up(sgsort::unique_ints((long long)n) * 4) +
           sg::align_up((size_t)std::max(N, 1) * 16) + 512;
}

// get_adj_from_pointcloud (util.py:814-834): d_points rows of `stride` floats (xyz first), k in {5, 10, 20}; d_adj: room for [N*k,2]
// int64; *h_n = rows written (per-row sorted, lexicographically sorted, unique)
int sg_pointcloud_adjacency(const float* d_points, int stride, int N, int k, int64_t* d_adj, int* h_n, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(d_points && stride >= 3 && N > 0 && d_adj && h_n && d_ws, "sg_pointcloud_adjacency: bad arguments");
    if (k != 5 && k != 10 && k != 20) return sg::fail(SG_EUNSUP, "sg_pointcloud_adjacency: k = %d is not built (5, 10 = the reference's default, 20)", k);
    if (N <= k) return sg::fail(SG_EINVAL, "sg_pointcloud_adjacency: %d points for k = %d (topk(k + 1) raises in the reference)", N, k);
    *h_n = 0;
    const int n = N * k;
    sg::Carver cv(d_ws, ws_bytes);
    unsigned long long* k0 = cv.take<unsigned long long>(n);
    unsigned long long* k1 = cv.take<unsigned long long>(n);
    unsigned long long* k2 = cv.take<unsigned long long>(n);
    int* h0 = cv.take<int>(sgsort::hist_ints(n));
    int* scratch = cv.take<int>(sgsort::unique_ints(n));
    float4* cand = cv.take<float4>(N);
    int* d_count = cv.take<int>(2);
    if (!cv.ok) return sg::fail(SG_ENOMEM, "sg_pointcloud_adjacency: workspace too small (%zu < %zu)", ws_bytes, sg_pointcloud_adjacency_ws_bytes(N, k));
    hipStream_t st = sg::as_stream(stream);
    k_pack_xyzw<<<sg::cdiv(N, 256), 256, 0, st>>>(d_points, stride, N, cand);
    if (k == 5) k_nearest_k<6><<<sg::cdiv(N, kTile), kTile, 0, st>>>(cand, N, k0);
    else if (k == 10) k_nearest_k<11><<<sg::cdiv(N, kTile), kTile, 0, st>>>(cand, N, k0);
    else k_nearest_k<21><<<sg::cdiv(N, kTile), kTile, 0, st>>>(cand, N, k0);
    unsigned long long* a[1] = {k0};
    unsigned long long* b[1] = {k1};
    unsigned long long* o[1] = {k2};
    int* hh[1] = {h0};
    const int rc = sort_unique_edges(a, b, o, 1, n, bits_for((long long)N + 1), hh, scratch, d_count, st);
    if (rc < 0) return rc;
    int cnt = 0;
    SG_HIP(hipMemcpyAsync(&cnt, d_count, 4, hipMemcpyDeviceToHost, st));
    SG_HIP(hipStreamSynchronize(st));
    if (cnt > 0) k_unpack_edges<<<sg::cdiv(cnt, 256), 256, 0, st>>>(k2, cnt, d_adj);
    *h_n = cnt;
    SG_LAUNCH_CHECK();
    return SG_OK;
}

size_t sg_segment_lists_ws_bytes(int V, int Np) {
    const size_t n = (size_t)std::max(std::max(V, Np), 1);
    return 6 * sg::align_up(n * 4) + sg::align_up(sgsort::hist_ints((long long)n) * 4) + sg::align_up(sgsort::unique_ints((long long)n) * 4) + 512;
}

// d_raw_label [V]: compacted ids (the `.seg.txt` column).  d_seg_points [Np] / d_seg_off [G+1]: sampled points grouped
// by ascending compacted id, ascending point index inside a group (G <= number of raw segments: a segment none of whose
// vertices was sampled has no group).  h_counts = {number of raw segments, G}.
int sg_segment_lists(const int32_t* d_seg_indices, int V, const int64_t* d_mapper, int Np, int32_t* d_raw_label, int32_t* d_seg_points,
                     int32_t* d_seg_off, int* h_counts, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(V > 0 && Np > 0 && d_seg_indices && d_mapper && d_raw_label && d_seg_points && d_seg_off && h_counts && d_ws,
               "sg_segment_lists: bad arguments");
    const size_t n = (size_t)std::max(V, Np);
    sg::Carver cv(d_ws, ws_bytes);
    int32_t* a = cv.take<int32_t>(n);
    int32_t* b = cv.take<int32_t>(n);
    int32_t* c = cv.take<int32_t>(n);
    int32_t* d = cv.take<int32_t>(n);
    int32_t* e = cv.take<int32_t>(n);
    int32_t* f = cv.take<int32_t>(n);
    int* hist = cv.take<int>(sgsort::hist_ints((long long)n));
    int* scratch = cv.take<int>(sgsort::unique_ints((long long)n));
    int* d_cnt = cv.take<int>(2);
    if (!cv.ok) return sg::fail(SG_ENOMEM, "sg_segment_lists: workspace too small (%zu < %zu)", ws_bytes, sg_segment_lists_ws_bytes(V, Np));
    hipStream_t st = sg::as_stream(stream);
    using L32 = sgsort::Lists<unsigned int, int>;
    // 1. sorted unique raw ids (ids are non-negative in ScanNet: the sort is on the value's 32 bits)
    SG_HIP(hipMemcpyAsync(a, d_seg_indices, (size_t)V * 4, hipMemcpyDeviceToDevice, st));
    L32 L{};
    L.kin[0] = (const unsigned int*)a; L.kout[0] = (unsigned int*)b; L.hist[0] = hist; L.n[0] = V;
    sgsort::radix_sort<unsigned int, int, false>(L, 1, 0, 32, st);
    const unsigned int* sorted = L.kin[0];
    int32_t* uniq = (int32_t*)L.kout[0];
    sgsort::unique_sorted<unsigned int>(sorted, V, (unsigned int*)uniq, nullptr, d_cnt, scratch, st);
    k_rank_labels<<<sg::cdiv(V, 256), 256, 0, st>>>(d_seg_indices, V, uniq, d_cnt, d_raw_label);
    // 2. stable sort of (sampled label, point index): groups in ascending label order, ascending index inside (a label is below V: the
    // number of raw segments is not on the host yet, so the sort covers the bits a label can have)
    k_sampled_labels<<<sg::cdiv(Np, 256), 256, 0, st>>>(d_raw_label, d_mapper, Np, c, d);
    L32 P{};
    P.kin[0] = (const unsigned int*)c; P.kout[0] = (unsigned int*)e; P.vin[0] = d; P.vout[0] = f; P.hist[0] = hist; P.n[0] = Np;
    sgsort::radix_sort<unsigned int, int, true>(P, 1, 0, bits_for((long long)V + 1), st);
    const unsigned int* skeys = P.kin[0];
    const int32_t* svals = P.vin[0];
    SG_HIP(hipMemcpyAsync(d_seg_points, svals, (size_t)Np * 4, hipMemcpyDeviceToDevice, st));
    // 3. group starts: where the sorted label changes, already in ascending order (the compaction keeps the order)
    sgsort::unique_sorted<unsigned int>(skeys, Np, nullptr, d_seg_off, d_cnt + 1, scratch, st);
    int cnt[2] = {0, 0};
    SG_HIP(hipMemcpyAsync(cnt, d_cnt, 8, hipMemcpyDeviceToHost, st));
    SG_HIP(hipStreamSynchronize(st));
    const int G = cnt[1];
    SG_HIP(hipMemcpyAsync(d_seg_off + G, &Np, 4, hipMemcpyHostToDevice, st));
    SG_HIP(hipStreamSynchronize(st));
    h_counts[0] = cnt[0];
    h_counts[1] = G;
    SG_LAUNCH_CHECK();
    return SG_OK;
}

}  // extern "C"
