// the TRAIN-mode tail of SegModel.forward and the backward of the three cluster-level
// operators in front of it (reference seggroup/model.py:900-932, Classifier 154-166, util.py:12-29, train.py:160-170).
//
//   forward   Feat_5 [C,256] --max over the clusters of one weak instance--> Feat_6 [K,256]
//             -> linear1 (no bias) -> BatchNorm1d (batch statistics over the K instances) -> LeakyReLU(0.2)
//             -> Dropout(0.5) with a PINNED keep mask (the reference draws it from torch's RNG stream: parity is stated
//                with the same mask on both sides) -> linear2 -> label-smoothed cross entropy (sum)
//   backward  of exactly that, for loss = scale * loss_sum (train.py:166: scale = 1 / loss_num)
//   and       backward of aggregate_cluster_feature (group max), of the point->cluster max and of the GCN layer
//             (through the row-normalised similarity weights exp(-alpha * ||x_a - x_b + 1e-6||), which autograd differentiates
//             in the reference: model.py:262-265,305-309).
//
// Everything here is tiny (K <= a few hundred instances, C <= S clusters, D <= 256): one or a few blocks, fp64
// accumulation, no tuning -- the EdgeConv / BatchNorm2d backward (the part with real work) is kernels_train_edge.hip, the
// optimizers are at the end of this file, the chain is trainer.cpp.
#include <cmath>

#include "sg_common.h"

namespace {

constexpr int kH = 128, kCls = 40, kD5 = 256;
constexpr float kEps = 0.2f;                  // label smoothing (util.py:18)

// Feat_6[k] = max over the clusters c with group[c] == k (ascending c: torch.max keeps the first maximal row)
__global__ void k_tail_group_max(const float* __restrict__ feat5, int C, const int32_t* __restrict__ group, int K, float* __restrict__ feat6,
                                 int32_t* __restrict__ arg) {
    const int k = blockIdx.x, d = threadIdx.x;
    float m = -INFINITY;
    int a = -1;
    for (int c = 0; c < C; ++c)
        if (group[c] == k) {
            const float v = feat5[(size_t)c * kD5 + d];
            if (a < 0 || v > m) { m = v; a = c; }
        }
    feat6[(size_t)k * kD5 + d] = m;
    arg[(size_t)k * kD5 + d] = a;
}

// linear1 (no bias): one block per instance, one thread per output
__global__ __launch_bounds__(kH) void k_tail_linear1(const float* __restrict__ feat6, const float* __restrict__ w1, float* __restrict__ h) {
    __shared__ float f[kD5];
    const int k = blockIdx.x, j = threadIdx.x;
    for (int d = j; d < kD5; d += kH) f[d] = feat6[(size_t)k * kD5 + d];
    __syncthreads();
    double acc = 0.0;
    for (int d = 0; d < kD5; ++d) acc = fma((double)f[d], (double)w1[j * kD5 + d], acc);
    h[k * kH + j] = (float)acc;
}

// one block: classifier behind linear1 + loss.  Saved for the backward: h (pre-BN), xhat, pre-activation y, z (after dropout), softmax p.
__global__ __launch_bounds__(256) void k_tail_classifier(const float* __restrict__ feat6, int K, const int32_t* __restrict__ gold,
                                                         const float* __restrict__ keep, const float* __restrict__ w1, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ w2, const float* __restrict__ b2,
                                                         float* __restrict__ h, float* __restrict__ xhat, float* __restrict__ ypre, float* __restrict__ z,
                                                         float* __restrict__ prob, float* __restrict__ stat, float* __restrict__ logits_out,
                                                         float* __restrict__ loss_out) {
    __shared__ double red[256];
    const int tid = threadIdx.x;                              // h = linear1(feat6) comes from k_tail_linear1 (one block per instance)
    if (tid < kH) {                                           // BatchNorm1d, batch statistics (biased variance, eps 1e-5)
        double s = 0.0, q = 0.0;
        for (int k = 0; k < K; ++k) { const double v = h[k * kH + tid]; s += v; }
        const double mean = s / K;
        for (int k = 0; k < K; ++k) { const double v = h[k * kH + tid] - mean; q += v * v; }
        const double inv = 1.0 / sqrt(q / K + 1e-5);
        stat[tid] = (float)mean; stat[kH + tid] = (float)inv;
        for (int k = 0; k < K; ++k) {
            const double xh = (h[k * kH + tid] - mean) * inv;
            const float y = (float)(xh * (double)gamma[tid] + (double)beta[tid]);
            xhat[k * kH + tid] = (float)xh;
            ypre[k * kH + tid] = y;
            const float a = fmaxf(y, 0.2f * y);
            z[k * kH + tid] = keep ? a * keep[k * kH + tid] : a;
        }
    }
    __syncthreads();
    double local = 0.0;
    for (int k = tid; k < K; k += 256) {                      // linear2 + log-softmax + smoothed cross entropy, one instance per thread
        double lg[kCls], mx = -1e300;
        for (int c = 0; c < kCls; ++c) {
            double acc = (double)b2[c];
            for (int j = 0; j < kH; ++j) acc = fma((double)z[k * kH + j], (double)w2[c * kH + j], acc);
            lg[c] = acc;
            mx = fmax(mx, acc);
            if (logits_out) logits_out[k * kCls + c] = (float)acc;
        }
        double se = 0.0;
        for (int c = 0; c < kCls; ++c) se += exp(lg[c] - mx);
        const double lse = mx + log(se);
        const int g = gold[k];
        for (int c = 0; c < kCls; ++c) {
            const double lp = lg[c] - lse;
            prob[k * kCls + c] = (float)exp(lp);
            const double t = c == g ? 1.0 - (double)kEps : (double)kEps / (kCls - 1);
            local -= t * lp;
        }
    }
    red[tid] = local;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < 256; ++i) s += red[i];
        loss_out[0] = (float)s;
        loss_out[1] = (float)K;
    }
}

// backward of k_tail_classifier + k_tail_group_max for loss = scale * loss_sum; one block
__global__ __launch_bounds__(256) void k_tail_backward(const float* __restrict__ feat6, const int32_t* __restrict__ arg, int K, int C,
                                                       const int32_t* __restrict__ gold, const float* __restrict__ keep,
                                                       const float* __restrict__ w1, const float* __restrict__ gamma, const float* __restrict__ w2,
                                                       const float* __restrict__ xhat, const float* __restrict__ ypre, const float* __restrict__ z,
                                                       const float* __restrict__ prob, const float* __restrict__ stat, float scale,
                                                       float* __restrict__ dlog, float* __restrict__ dy, float* __restrict__ dh,
                                                       float* __restrict__ gw1, float* __restrict__ gg, float* __restrict__ gb, float* __restrict__ gw2,
                                                       float* __restrict__ gb2, float* __restrict__ gfeat5) {
    const int tid = threadIdx.x;
    for (int i = tid; i < K * kCls; i += 256) {               // d loss / d logits = scale * (softmax - smoothed target)
        const int k = i / kCls, c = i % kCls;
        const float t = c == gold[k] ? 1.f - kEps : kEps / (kCls - 1);
        dlog[i] = scale * (prob[i] - t);
    }
    __syncthreads();
    if (tid < kCls) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += dlog[k * kCls + tid];
        gb2[tid] = (float)s;
    }
    for (int i = tid; i < kCls * kH; i += 256) {              // gw2[c][j] = sum_k dlog[k][c] z[k][j]
        const int c = i / kH, j = i % kH;
        double s = 0.0;
        for (int k = 0; k < K; ++k) s = fma((double)dlog[k * kCls + c], (double)z[k * kH + j], s);
        gw2[i] = (float)s;
    }
    for (int i = tid; i < K * kH; i += 256) {                 // through linear2, dropout, LeakyReLU
        const int k = i / kH, j = i % kH;
        double s = 0.0;
        for (int c = 0; c < kCls; ++c) s = fma((double)dlog[k * kCls + c], (double)w2[c * kH + j], s);
        if (keep) s *= (double)keep[i];
        dy[i] = (float)(ypre[i] > 0.f ? s : 0.2 * s);
    }
    __syncthreads();
    if (tid < kH) {                                           // BatchNorm1d backward (batch statistics)
        double sdy = 0.0, sdyx = 0.0;
        for (int k = 0; k < K; ++k) { sdy += dy[k * kH + tid]; sdyx += (double)dy[k * kH + tid] * (double)xhat[k * kH + tid]; }
        gb[tid] = (float)sdy;
        gg[tid] = (float)sdyx;
        const double gi = (double)gamma[tid] * (double)stat[kH + tid];
        for (int k = 0; k < K; ++k)
            dh[k * kH + tid] = (float)(gi * ((double)dy[k * kH + tid] - sdy / K - (double)xhat[k * kH + tid] * sdyx / K));
    }
    __syncthreads();
}

// gw1[j][d] = sum_k dh[k][j] feat6[k][d]: block j, thread d
__global__ __launch_bounds__(kD5) void k_tail_backward_w1(const float* __restrict__ feat6, const float* __restrict__ dh, int K, float* __restrict__ gw1) {
    const int j = blockIdx.x, d = threadIdx.x;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s = fma((double)dh[k * kH + j], (double)feat6[(size_t)k * kD5 + d], s);
    gw1[j * kD5 + d] = (float)s;
}

// d Feat_6 = dh W1 -> the row of Feat_5 that won the max (gfeat5 zeroed before): block k, thread d
__global__ __launch_bounds__(kD5) void k_tail_backward_feat(const float* __restrict__ dh, const float* __restrict__ w1, const int32_t* __restrict__ arg,
                                                            float* __restrict__ gfeat5) {
    __shared__ float dhk[kH];
    const int k = blockIdx.x, d = threadIdx.x;
    if (d < kH) dhk[d] = dh[k * kH + d];
    __syncthreads();
    double s = 0.0;
    for (int j = 0; j < kH; ++j) s = fma((double)dhk[j], (double)w1[j * kD5 + d], s);
    const int a = arg[(size_t)k * kD5 + d];
    if (a >= 0) gfeat5[(size_t)a * kD5 + d] = (float)s;       // every (instance, channel) has its own winner row: no conflicts
}

// ---- backward of aggregate_cluster_feature (model.py:278-288): the gradient of a group's maximum goes to the first maximal row
__global__ void k_group_max_backward(const float* __restrict__ rows, int row_stride, int D, const int32_t* __restrict__ goff,
                                     const int32_t* __restrict__ gidx, const float* __restrict__ gout, int out_stride, float* __restrict__ grows,
                                     int grow_stride) {
    const int g = blockIdx.x;
    const int lo = goff[g], hi = goff[g + 1];
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
        float m = -INFINITY;
        int a = -1;
        for (int i = lo; i < hi; ++i) {
            const float v = rows[(size_t)gidx[i] * row_stride + k];
            if (a < 0 || v > m) { m = v; a = i; }
        }
        for (int i = lo; i < hi; ++i) grows[(size_t)gidx[i] * grow_stride + k] = i == a ? gout[(size_t)g * out_stride + k] : 0.f;
    }
}

// ---- backward of the point -> cluster max (model.py:793,834): rows in member order, cluster c = rows [cl_off[c], cl_off[c+1]).
// A floor or wall is one cluster of tens of thousands of rows, so the arg-max runs over ROWS in parallel: 64 rows per block and
// one thread per column reduce to one key, then one atomicMax per (block, cluster, column) on
//     key = order-preserving bits of the value << 32 | ~row          (largest value, FIRST row on ties, whatever the arrival order)
// and a second kernel sends the cluster's gradient to the winning row (everything else was zeroed by a memset).
__device__ __forceinline__ unsigned long long segmax_key(float v, int row) {
    const unsigned int u = __float_as_uint(v);
    const unsigned int o = u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
    return ((unsigned long long)o << 32) | (unsigned int)(0xffffffffu - (unsigned int)row);
}
__global__ void k_segment_max_keys(const float* __restrict__ rows, int N, int D, const int32_t* __restrict__ cl_off, int C,
                                   unsigned long long* __restrict__ keys) {
    const int r0 = blockIdx.x * 64, r1 = min(N, r0 + 64), k = threadIdx.x;
    int lo = 0, hi = C;                                       // the cluster of row r0: largest c with cl_off[c] <= r0
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cl_off[mid] <= r0) lo = mid; else hi = mid;
    }
    int c = lo, end = cl_off[c + 1];
    unsigned long long best = 0ull;
    for (int r = r0; r < r1; ++r) {
        if (r >= end) {
            if (best) atomicMax(&keys[(size_t)c * D + k], best);
            best = 0ull;
            while (r >= end) { ++c; end = cl_off[c + 1]; }
        }
        const unsigned long long key = segmax_key(rows[(size_t)r * D + k], r);
        best = key > best ? key : best;
    }
    if (best) atomicMax(&keys[(size_t)c * D + k], best);
}
__global__ void k_segment_max_scatter(const unsigned long long* __restrict__ keys, int D, const float* __restrict__ gout, int out_stride,
                                      float* __restrict__ grows) {
    const int c = blockIdx.x, k = threadIdx.x;
    const unsigned long long key = keys[(size_t)c * D + k];
    if (key == 0ull) return;                                  // an empty cluster
    const int row = (int)(0xffffffffu - (unsigned int)(key & 0xffffffffu));
    grows[(size_t)row * D + k] = gout[(size_t)c * out_stride + k];
}

// ---- GCN backward.  forward: s_e = exp(-alpha d_e), d_e = ||x_a - x_b + 1e-6||, r_i = 1 + sum_j s_ij,
//      agg_i = (x_i + sum_j s_ij x_j) / r_i, out = relu(agg W^T)
// step 1 (rows): recompute agg and out, G = gout * [out > 0], gagg = G W, partial gW
__global__ __launch_bounds__(256) void k_gcn_bwd_rows(const float* __restrict__ x, int S, int D, const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ col, const int32_t* __restrict__ eid, const int32_t* __restrict__ adj,
                                                      const float* __restrict__ w, float alpha, const float* __restrict__ gout, float* __restrict__ agg,
                                                      float* __restrict__ rsum, float* __restrict__ sw, float* __restrict__ dist, float* __restrict__ G,
                                                      float* __restrict__ gagg) {
    __shared__ float a_row[256], g_row[256];
    const int i = blockIdx.x, tid = threadIdx.x;
    const int lo = rowptr[i], hi = rowptr[i + 1];
    // edge weights of this row (each edge evaluated by both of its rows, identically)
    for (int e = lo + tid; e < hi; e += 256) {
        const int id = eid[e];
        const float* pa = x + (size_t)adj[2 * id] * D;
        const float* pb = x + (size_t)adj[2 * id + 1] * D;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) { const double d = (double)pa[k] - (double)pb[k] + 1e-6; acc = fma(d, d, acc); }
        const double dd = sqrt(acc);
        dist[id] = (float)dd;
        sw[id] = (float)exp(-dd * (double)alpha);
    }
    __syncthreads();
    double r = 1.0;
    for (int e = lo; e < hi; ++e) r += (double)sw[eid[e]];
    if (tid == 0) rsum[i] = (float)r;
    if (tid < D) {
        double acc = (double)x[(size_t)i * D + tid];
        for (int e = lo; e < hi; ++e) acc = fma((double)sw[eid[e]], (double)x[(size_t)col[e] * D + tid], acc);
        a_row[tid] = (float)(acc / r);
        agg[(size_t)i * D + tid] = a_row[tid];
    }
    __syncthreads();
    if (tid < D) {
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc = fma((double)a_row[k], (double)w[(size_t)tid * D + k], acc);
        g_row[tid] = acc > 0.0 ? gout[(size_t)i * D + tid] : 0.f;
        G[(size_t)i * D + tid] = g_row[tid];
    }
    __syncthreads();
    if (tid < D) {
        double acc = 0.0;
        for (int o = 0; o < D; ++o) acc = fma((double)g_row[o], (double)w[(size_t)o * D + tid], acc);
        gagg[(size_t)i * D + tid] = (float)acc;
    }
}
// step 2: gW[o][k] = sum_i G[i][o] agg[i][k]
__global__ void k_gcn_bwd_w(const float* __restrict__ G, const float* __restrict__ agg, int S, int D, float* __restrict__ gw) {
    const int o = blockIdx.x;
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
        double acc = 0.0;
        for (int i = 0; i < S; ++i) acc = fma((double)G[(size_t)i * D + o], (double)agg[(size_t)i * D + k], acc);
        gw[(size_t)o * D + k] = (float)acc;
    }
}
// step 3 (edges): gradient of the shared weight s_e from both rows -> gradient of d_e -> coefficient of (x_a - x_b + 1e-6)
__global__ void k_gcn_bwd_edges(const float* __restrict__ x, int D, const int32_t* __restrict__ adj, int E, const float* __restrict__ agg,
                                const float* __restrict__ rsum, const float* __restrict__ sw, const float* __restrict__ dist,
                                const float* __restrict__ gagg, float alpha, float* __restrict__ coef) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int a = adj[2 * e], b = adj[2 * e + 1];
    double ga = 0.0, gb = 0.0;
    for (int k = 0; k < D; ++k) {
        ga = fma((double)gagg[(size_t)a * D + k], (double)x[(size_t)b * D + k] - (double)agg[(size_t)a * D + k], ga);
        gb = fma((double)gagg[(size_t)b * D + k], (double)x[(size_t)a * D + k] - (double)agg[(size_t)b * D + k], gb);
    }
    const double gs = ga / (double)rsum[a] + gb / (double)rsum[b];
    const double gd = -(double)alpha * (double)sw[e] * gs;
    coef[e] = (float)(gd / (double)dist[e]);                  // d d_e / d x_a = (x_a - x_b + 1e-6) / d_e
}
// step 4 (rows): gx_i = gagg_i / r_i + sum_j (s_ij / r_j) gagg_j + sum_e coef_e * (+-)(x_a - x_b + 1e-6)
__global__ void k_gcn_bwd_x(const float* __restrict__ x, int D, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                            const int32_t* __restrict__ eid, const int32_t* __restrict__ adj, const float* __restrict__ rsum,
                            const float* __restrict__ sw, const float* __restrict__ gagg, const float* __restrict__ coef, float* __restrict__ gx) {
    const int i = blockIdx.x;
    const int lo = rowptr[i], hi = rowptr[i + 1];
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
        double acc = (double)gagg[(size_t)i * D + k] / (double)rsum[i];
        for (int e = lo; e < hi; ++e) {
            const int id = eid[e], j = col[e];
            acc = fma((double)sw[id] / (double)rsum[j], (double)gagg[(size_t)j * D + k], acc);
            const int a = adj[2 * id], b = adj[2 * id + 1];
            const double diff = (double)x[(size_t)a * D + k] - (double)x[(size_t)b * D + k] + 1e-6;
            acc = fma(i == a ? (double)coef[id] : -(double)coef[id], diff, acc);
        }
        gx[(size_t)i * D + k] = (float)acc;
    }
}

}  // namespace

namespace {

// torch.optim.SGD (train.py:96): g' = g + wd * p;  buf = first ? g' : momentum * buf + g';  p -= lr * buf
__global__ void k_sgd_step(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, int n, float lr, float momentum, float wd,
                           int first) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float gg = g[i] + wd * p[i];
        const float b = first ? gg : momentum * buf[i] + gg;
        buf[i] = b;
        p[i] = p[i] - lr * b;
    }
}

// torch.optim.Adam (train.py:98; betas 0.9 / 0.999, eps 1e-8, L2 weight decay added to the gradient), step = 1, 2, ...
__global__ void k_adam_step(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int n, float lr, float wd,
                            float beta1, float beta2, float eps, float bc1, float bc2_sqrt) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float gg = g[i] + wd * p[i];
        const float mi = beta1 * m[i] + (1.f - beta1) * gg;
        const float vi = beta2 * v[i] + (1.f - beta2) * gg * gg;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

__global__ void k_transpose_square(const float* __restrict__ src, float* __restrict__ dst, int D) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y)
        if (by + r < D && bx + (int)threadIdx.x < D) tile[r][threadIdx.x] = src[(size_t)(by + r) * D + bx + threadIdx.x];
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y)
        if (bx + r < D && by + (int)threadIdx.x < D) dst[(size_t)(bx + r) * D + by + threadIdx.x] = tile[threadIdx.x][r];
}

}  // namespace

namespace sg {
int transpose_square(const float* d_src, float* d_dst, int D, hipStream_t st) {
    k_transpose_square<<<dim3(cdiv(D, 32), cdiv(D, 32)), dim3(32, 8), 0, st>>>(d_src, d_dst, D);
    SG_LAUNCH_CHECK();
    return SG_OK;
}
}  // namespace sg

extern "C" {

size_t sg_train_tail_ws_bytes(int C, int K) {
    const size_t k = (size_t)std::max(K, 1);
    (void)C;
    return sg::align_up(k * kD5 * 4) * 2 + sg::align_up(k * kH * 4) * 6 + sg::align_up(k * kCls * 4) * 2 + sg::align_up(2 * kH * 4) + 1024;
}

namespace {
struct TailWs {
    float* feat6; int32_t* arg; float *h, *xhat, *ypre, *z, *dy, *dh, *prob, *dlog, *stat;
    bool ok;
};
TailWs carve_tail(void* d_ws, size_t bytes, int K) {
    sg::Carver cv(d_ws, bytes);
    TailWs t;
    t.feat6 = cv.take<float>((size_t)K * kD5); t.arg = cv.take<int32_t>((size_t)K * kD5);
    t.h = cv.take<float>((size_t)K * kH); t.xhat = cv.take<float>((size_t)K * kH); t.ypre = cv.take<float>((size_t)K * kH);
    t.z = cv.take<float>((size_t)K * kH); t.dy = cv.take<float>((size_t)K * kH); t.dh = cv.take<float>((size_t)K * kH);
    t.prob = cv.take<float>((size_t)K * kCls); t.dlog = cv.take<float>((size_t)K * kCls); t.stat = cv.take<float>(2 * kH);
    t.ok = cv.ok;
    return t;
}
}  // namespace

int sg_train_tail_forward(const float* d_feat5, int C, const int32_t* d_group, int K, const int32_t* d_gold, const float* d_keep,
                          const sg_classifier* cls, float* d_logits, float* d_loss, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(d_feat5 && d_group && d_gold && cls && d_loss && d_ws && C > 0, "sg_train_tail_forward: bad arguments");
    if (K < 2) return sg::fail(SG_EUNSUP, "sg_train_tail_forward: %d weak instance(s): BatchNorm1d in training mode needs more than one row (torch raises ValueError)", K);
    TailWs t = carve_tail(d_ws, ws_bytes, K);
    if (!t.ok) return sg::fail(SG_ENOMEM, "sg_train_tail_forward: workspace too small (%zu < %zu)", ws_bytes, sg_train_tail_ws_bytes(C, K));
    hipStream_t st = sg::as_stream(stream);
    k_tail_group_max<<<K, kD5, 0, st>>>(d_feat5, C, d_group, K, t.feat6, t.arg);
    k_tail_linear1<<<K, kH, 0, st>>>(t.feat6, cls->w1, t.h);
    k_tail_classifier<<<1, 256, 0, st>>>(t.feat6, K, d_gold, d_keep, cls->w1, cls->gamma, cls->beta, cls->w2, cls->b2, t.h, t.xhat, t.ypre, t.z, t.prob,
                                         t.stat, d_logits, d_loss);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_train_tail_backward(int C, int K, const int32_t* d_gold, const float* d_keep, const sg_classifier* cls, float scale, float* d_gw1,
                           float* d_ggamma, float* d_gbeta, float* d_gw2, float* d_gb2, float* d_gfeat5, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(d_gold && cls && d_gw1 && d_ggamma && d_gbeta && d_gw2 && d_gb2 && d_gfeat5 && d_ws && C > 0 && K >= 2, "sg_train_tail_backward: bad arguments");
    TailWs t = carve_tail(d_ws, ws_bytes, K);
    if (!t.ok) return sg::fail(SG_ENOMEM, "sg_train_tail_backward: workspace too small");
    hipStream_t st = sg::as_stream(stream);
    SG_HIP(hipMemsetAsync(d_gfeat5, 0, (size_t)C * kD5 * 4, st));
    k_tail_backward<<<1, 256, 0, st>>>(t.feat6, t.arg, K, C, d_gold, d_keep, cls->w1, cls->gamma, cls->w2, t.xhat, t.ypre, t.z, t.prob,
                                       t.stat, scale, t.dlog, t.dy, t.dh, d_gw1, d_ggamma, d_gbeta, d_gw2, d_gb2, d_gfeat5);
    k_tail_backward_w1<<<kH, kD5, 0, st>>>(t.feat6, t.dh, K, d_gw1);
    k_tail_backward_feat<<<K, kD5, 0, st>>>(t.dh, cls->w1, t.arg, d_gfeat5);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_group_max_rows_backward(const float* d_rows, int row_stride, int D, const int32_t* d_goff, const int32_t* d_gidx, int G,
                               const float* d_gout, int out_stride, float* d_grows, int grow_stride, void* stream) {
    SG_REQUIRE(G >= 0 && D > 0 && d_rows && d_gout && d_grows, "sg_group_max_rows_backward: bad arguments");
    if (G == 0) return SG_OK;
    k_group_max_backward<<<G, 256, 0, sg::as_stream(stream)>>>(d_rows, row_stride, D, d_goff, d_gidx, d_gout, out_stride, d_grows, grow_stride);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

size_t sg_segment_max_backward_ws_bytes(int C, int D) { return sg::align_up((size_t)std::max(C, 1) * std::max(D, 1) * 8); }

int sg_segment_max_backward(const float* d_rows, int N, int D, const int32_t* d_cl_off, int C, const float* d_gout, int out_stride,
                            float* d_grows, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(N >= 0 && C >= 0 && D > 0 && D <= 1024 && d_rows && d_cl_off && d_gout && d_grows && d_ws, "sg_segment_max_backward: bad arguments");
    if (C == 0 || N == 0) return SG_OK;
    if (ws_bytes < sg_segment_max_backward_ws_bytes(C, D)) return sg::fail(SG_ENOMEM, "sg_segment_max_backward: workspace too small");
    hipStream_t st = sg::as_stream(stream);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(d_ws);
    SG_HIP(hipMemsetAsync(keys, 0, (size_t)C * D * 8, st));
    SG_HIP(hipMemsetAsync(d_grows, 0, (size_t)N * D * 4, st));
    k_segment_max_keys<<<sg::cdiv(N, 64), D, 0, st>>>(d_rows, N, D, d_cl_off, C, keys);
    k_segment_max_scatter<<<C, D, 0, st>>>(keys, D, d_gout, out_stride, d_grows);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

size_t sg_gcn_backward_ws_bytes(int S, int D, int E) {
    const size_t s = (size_t)std::max(S, 1), e = (size_t)std::max(E, 1);
    return sg::align_up(s * D * 4) * 3 + sg::align_up(s * 4) + sg::align_up(e * 4) * 3 + 1024;
}

int sg_gcn_backward(const float* d_x, int S, int D, const int32_t* d_adj, int E, const int32_t* d_rowptr, const int32_t* d_col,
                    const int32_t* d_eid, const float* d_w, float alpha, const float* d_gout, float* d_gx, float* d_gw, void* d_ws, size_t ws_bytes,
                    void* stream) {
    SG_REQUIRE(S >= 0 && D > 0 && D <= 256 && E >= 0 && d_x && d_w && d_gout && d_gx && d_gw && d_ws, "sg_gcn_backward: bad arguments (D=%d must be <= 256)", D);
    if (S == 0) return SG_OK;
    sg::Carver cv(d_ws, ws_bytes);
    float* agg = cv.take<float>((size_t)S * D);
    float* G = cv.take<float>((size_t)S * D);
    float* gagg = cv.take<float>((size_t)S * D);
    float* rsum = cv.take<float>(S);
    float* sw = cv.take<float>(std::max(E, 1));
    float* dist = cv.take<float>(std::max(E, 1));
    float* coef = cv.take<float>(std::max(E, 1));
    if (!cv.ok) return sg::fail(SG_ENOMEM, "sg_gcn_backward: workspace too small (%zu < %zu)", ws_bytes, sg_gcn_backward_ws_bytes(S, D, E));
    hipStream_t st = sg::as_stream(stream);
    k_gcn_bwd_rows<<<S, 256, 0, st>>>(d_x, S, D, d_rowptr, d_col, d_eid, d_adj, d_w, alpha, d_gout, agg, rsum, sw, dist, G, gagg);
    k_gcn_bwd_w<<<D, 256, 0, st>>>(G, agg, S, D, d_gw);
    if (E > 0) k_gcn_bwd_edges<<<sg::cdiv(E, 128), 128, 0, st>>>(d_x, D, d_adj, E, agg, rsum, sw, dist, gagg, alpha, coef);
    k_gcn_bwd_x<<<S, 256, 0, st>>>(d_x, D, d_rowptr, d_col, d_eid, d_adj, rsum, sw, gagg, coef, d_gx);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

namespace {
__global__ void k_tail_bn_stats(const float* __restrict__ stat, float* __restrict__ out) {
    const int j = threadIdx.x;
    if (j < kH) {
        const double inv = (double)stat[kH + j];
        out[j] = stat[j];
        out[kH + j] = (float)fmax(1.0 / (inv * inv) - 1e-5, 0.0);
    }
}
}  // namespace

int sg_train_tail_bn_stats(void* d_ws, size_t ws_bytes, int K, float* d_out, void* stream) {
    SG_REQUIRE(d_ws && d_out && K >= 2, "sg_train_tail_bn_stats: bad arguments");
    TailWs t = carve_tail(d_ws, ws_bytes, K);
    if (!t.ok) return sg::fail(SG_ENOMEM, "sg_train_tail_bn_stats: workspace too small");
    k_tail_bn_stats<<<1, kH, 0, sg::as_stream(stream)>>>(t.stat, d_out);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

namespace {
// cross_entropy_loss (util.py:12-29): label smoothing eps = 0.2 over the C - 1 other classes, or the plain sum; one block
__global__ __launch_bounds__(256) void k_cross_entropy(const float* __restrict__ logits, int K, int C, const int32_t* __restrict__ gold, int smoothing,
                                                       float* __restrict__ prob, float* __restrict__ loss_out) {
    __shared__ double red[256];
    double local = 0.0;
    for (int k = threadIdx.x; k < K; k += 256) {
        double mx = -1e300;
        for (int c = 0; c < C; ++c) mx = fmax(mx, (double)logits[(size_t)k * C + c]);
        double se = 0.0;
        for (int c = 0; c < C; ++c) se += exp((double)logits[(size_t)k * C + c] - mx);
        const double lse = mx + log(se);
        const int g = gold[k];
        for (int c = 0; c < C; ++c) {
            const double lp = (double)logits[(size_t)k * C + c] - lse;
            prob[(size_t)k * C + c] = (float)exp(lp);
            const double t = smoothing ? (c == g ? 1.0 - (double)kEps : (double)kEps / (C - 1)) : (c == g ? 1.0 : 0.0);
            local -= t * lp;
        }
    }
    red[threadIdx.x] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < 256; ++i) s += red[i];
        loss_out[0] = (float)s;
    }
}
__global__ void k_cross_entropy_backward(const float* __restrict__ prob, int K, int C, const int32_t* __restrict__ gold, int smoothing, float scale,
                                         float* __restrict__ glogits) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K * C; i += gridDim.x * blockDim.x) {
        const int k = i / C, c = i - k * C;
        const float t = smoothing ? (c == gold[k] ? 1.f - kEps : kEps / (C - 1)) : (c == gold[k] ? 1.f : 0.f);
        glogits[i] = scale * (prob[i] - t);
    }
}
}  // namespace

int sg_cross_entropy_forward(const float* d_logits, int K, int C, const int32_t* d_gold, int smoothing, float* d_prob, float* d_loss, void* stream) {
    SG_REQUIRE(d_logits && d_gold && d_prob && d_loss && K > 0 && C > 1, "sg_cross_entropy_forward: bad arguments");
    k_cross_entropy<<<1, 256, 0, sg::as_stream(stream)>>>(d_logits, K, C, d_gold, smoothing, d_prob, d_loss);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_cross_entropy_backward(const float* d_prob, int K, int C, const int32_t* d_gold, int smoothing, float scale, float* d_glogits, void* stream) {
    SG_REQUIRE(d_prob && d_gold && d_glogits && K > 0 && C > 1, "sg_cross_entropy_backward: bad arguments");
    k_cross_entropy_backward<<<std::min(sg::cdiv((long long)K * C, 256), 1024), 256, 0, sg::as_stream(stream)>>>(d_prob, K, C, d_gold, smoothing, scale, d_glogits);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_optimizer_sgd(float* d_params, const float* d_grads, float* d_momentum_buf, int n, float lr, float momentum, float weight_decay,
                     int first_step, void* stream) {
    SG_REQUIRE(d_params && d_grads && d_momentum_buf && n >= 0, "sg_optimizer_sgd: bad arguments");
    if (n == 0) return SG_OK;
    k_sgd_step<<<std::min(sg::cdiv(n, 256), 1024), 256, 0, sg::as_stream(stream)>>>(d_params, d_grads, d_momentum_buf, n, lr, momentum, weight_decay, first_step);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_optimizer_adam(float* d_params, const float* d_grads, float* d_m, float* d_v, int n, float lr, float weight_decay, int step, void* stream) {
    SG_REQUIRE(d_params && d_grads && d_m && d_v && n >= 0 && step >= 1, "sg_optimizer_adam: bad arguments (step counts from 1)");
    if (n == 0) return SG_OK;
    const double b1 = 0.9, b2 = 0.999;
    const float bc1 = (float)(1.0 - std::pow(b1, step)), bc2s = (float)std::sqrt(1.0 - std::pow(b2, step));
    k_adam_step<<<std::min(sg::cdiv(n, 256), 1024), 256, 0, sg::as_stream(stream)>>>(d_params, d_grads, d_m, d_v, n, lr, weight_decay, 0.9f, 0.999f, 1e-8f, bc1, bc2s);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

}  // extern "C"

