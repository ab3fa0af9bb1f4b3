// The reference's module-level functions with arguments the hot path never passes (missing #6): the Python surface keeps the
// reference's signatures (seggroup_amd/functional.py), so these configurations run too -- plain kernels, off every timed path:
//   sg_group_mean_rows   aggregate_cluster_feature(use_avg=True)                      model.py:278-288
//   sg_fps_general       farthest_point_sampling(initial_idx, skip_initial, distances) model.py:329-395
//   sg_knn_general       knn(x, k) for any channel count and any k <= 128             model.py:30-36
#include <cfloat>
#include "sg_common.h"

namespace {

// mean over each group's rows; the sum is carried in double (torch.mean's fp32 pairwise order is not reproduced: floats, 1e-4)
__global__ void k_group_mean_rows(const float* __restrict__ rows, int row_stride, int D, const int32_t* __restrict__ goff,
                                  const int32_t* __restrict__ gidx, float* __restrict__ out, int out_stride) {
    const int g = blockIdx.x, lo = goff[g], hi = goff[g + 1];
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
        double s = 0.0;
        for (int i = lo; i < hi; ++i) s += (double)rows[(size_t)gidx[i] * row_stride + k];
        out[(size_t)g * out_stride + k] = hi > lo ? (float)(s / (double)(hi - lo)) : __builtin_nanf("");     // torch.mean of nothing is NaN
    }
}

// One workgroup walks the k steps of model.py:369-394 over n points of `dim` coordinates: l2_norm's ((x - y)^2).sum(axis) in the order NumPy
// adds a short axis (sequentially), np.argmax's first-index ties.  min_d lives in global memory (n is not bounded by LDS).
constexpr int kFpsBlock = 256;
__global__ __launch_bounds__(kFpsBlock) void k_fps_general(const float* __restrict__ pts, int n, int dim, int k, int initial_idx, int skip_initial,
                                                           int32_t* __restrict__ indices, float* __restrict__ distances, float* __restrict__ min_d) {
    __shared__ float red_v[kFpsBlock];
    __shared__ int red_i[kFpsBlock];
    __shared__ int pick;
    const int tid = threadIdx.x;
    auto dist_to = [&](int far, int j) {
        float s = 0.f;
        for (int c = 0; c < dim; ++c) { const float d = pts[(size_t)far * dim + c] - pts[(size_t)j * dim + c]; s = c == 0 ? d * d : s + d * d; }
        return s;
    };
    auto argmax_min = [&]() {                                   // first index of the maximum of min_d
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int j = tid; j < n; j += kFpsBlock) { const float v = min_d[j]; if (v > bv || (v == bv && j < bi) || bi == 0x7fffffff) { if (v > bv || bi == 0x7fffffff) { bv = v; bi = j; } } }
        red_v[tid] = bv; red_i[tid] = bi;
        __syncthreads();
        for (int o = kFpsBlock / 2; o > 0; o >>= 1) {
            if (tid < o) {
                const float v = red_v[tid + o]; const int i = red_i[tid + o];
                if (i != 0x7fffffff && (red_i[tid] == 0x7fffffff || v > red_v[tid] || (v == red_v[tid] && i < red_i[tid]))) { red_v[tid] = v; red_i[tid] = i; }
            }
            __syncthreads();
        }
        if (tid == 0) pick = red_i[0];
        __syncthreads();
        return pick;
    };
    int far = initial_idx;
    for (int j = tid; j < n; j += kFpsBlock) min_d[j] = dist_to(far, j);
    __syncthreads();
    if (skip_initial) {                                         // model.py:382-386
        far = argmax_min();
        __syncthreads();
        for (int j = tid; j < n; j += kFpsBlock) min_d[j] = dist_to(far, j);
        __syncthreads();
    }
    if (tid == 0) indices[0] = far;
    if (distances) for (int j = tid; j < n; j += kFpsBlock) distances[j] = min_d[j];
    for (int i = 1; i < k; ++i) {
        far = argmax_min();
        if (tid == 0) indices[i] = far;
        __syncthreads();
        for (int j = tid; j < n; j += kFpsBlock) {
            const float d = dist_to(far, j);
            if (distances) distances[(size_t)i * n + j] = d;
            min_d[j] = fminf(min_d[j], d);
        }
        __syncthreads();
    }
}

// knn(): pairwise_distance[i][j] = (-xx_i - (-2 x_i . x_j)) - xx_j, top k per row in descending order, lower index first among equals (the
// build's defined tie rule).  One thread per query, the list in registers / local memory (k <= 128).
constexpr int kKnnMaxK = 128;
__global__ __launch_bounds__(128) void k_knn_general(const float* __restrict__ x, int B, int C, int n, int k, int64_t* __restrict__ idx) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (q >= n) return;
    const float* xb = x + (size_t)b * C * n;
    float best_v[kKnnMaxK];
    int best_i[kKnnMaxK];
    int have = 0;
    float xxq = 0.f;
    for (int c = 0; c < C; ++c) { const float v = xb[(size_t)c * n + q]; xxq = c == 0 ? v * v : xxq + v * v; }
    for (int j = 0; j < n; ++j) {
        float dot = 0.f, xxj = 0.f;
        for (int c = 0; c < C; ++c) {
            const float a = xb[(size_t)c * n + q], p = xb[(size_t)c * n + j];
            dot = c == 0 ? a * p : __builtin_fmaf(a, p, dot);
            xxj = c == 0 ? p * p : xxj + p * p;
        }
        const float s = ((-xxq) - (-2.0f * dot)) - xxj;
        if (have == k && !(s > best_v[k - 1])) continue;         // equal score, larger index: stays out
        int at = have < k ? have : k - 1;
        while (at > 0 && best_v[at - 1] < s) { best_v[at] = best_v[at - 1]; best_i[at] = best_i[at - 1]; --at; }
        best_v[at] = s; best_i[at] = j;
        if (have < k) ++have;
    }
    int64_t* o = idx + ((size_t)b * n + q) * k;
    for (int t = 0; t < k; ++t) o[t] = best_i[t];
}

}  // namespace

extern "C" {

int sg_group_mean_rows(const float* d_rows, int row_stride, int D, const int32_t* d_goff, const int32_t* d_gidx, int G, float* d_out,
                       int out_stride, void* stream) {
    SG_REQUIRE(G >= 0 && D > 0 && d_rows && d_goff && d_gidx && d_out, "sg_group_mean_rows: bad arguments");
    if (G == 0) return SG_OK;
    k_group_mean_rows<<<G, 64 * ((std::min(D, 256) + 63) / 64), 0, sg::as_stream(stream)>>>(d_rows, row_stride, D, d_goff, d_gidx, d_out, out_stride);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

size_t sg_fps_general_ws_bytes(int n) { return sg::align_up((size_t)std::max(n, 1) * sizeof(float)); }

int sg_fps_general(const float* d_pts, int n, int dim, int k, int initial_idx, int skip_initial, int32_t* d_indices, float* d_distances,
                   void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(d_pts && d_indices && n > 0 && dim > 0 && k > 0, "sg_fps_general: bad arguments");
    SG_REQUIRE(initial_idx >= 0 && initial_idx < n, "sg_fps_general: initial_idx %d outside [0, %d)", initial_idx, n);
    SG_REQUIRE(d_ws && ws_bytes >= sg_fps_general_ws_bytes(n), "sg_fps_general: workspace too small");
    k_fps_general<<<1, kFpsBlock, 0, sg::as_stream(stream)>>>(d_pts, n, dim, k, initial_idx, skip_initial ? 1 : 0, d_indices, d_distances,
                                                              static_cast<float*>(d_ws));
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_knn_general(const float* d_x, int B, int C, int n, int k, int64_t* d_idx, void* stream) {
    SG_REQUIRE(d_x && d_idx && B > 0 && C > 0 && n > 0, "sg_knn_general: bad arguments");
    if (k <= 0 || k > n || k > kKnnMaxK) return sg::fail(SG_EUNSUP, "sg_knn_general: k = %d (1 <= k <= min(n, %d); torch.topk raises beyond n)", k, kKnnMaxK);
    k_knn_general<<<dim3(sg::cdiv(n, 128), B), 128, 0, sg::as_stream(stream)>>>(d_x, B, C, n, k, d_idx);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

}  // extern "C"
