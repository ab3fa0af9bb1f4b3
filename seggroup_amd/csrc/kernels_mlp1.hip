// a6+a7: MLP1 = knn(k=10) + get_graph_feature1 + conv1x1 6->64 + BatchNorm2d (batch statistics) +
// LeakyReLU(0.2) + max over k + [max | mean] over the 64 samples (reference seggroup/model.py:30-80).
//
// One wave per cluster: lane i owns sample i of the cluster's 64 samples; neighbours are exchanged
// with wave shuffles, nothing touches LDS.  BatchNorm needs statistics over ALL C*64*10 rows before
// any output exists, so the op is three launches:
//   1. k_mlp1_knn_moments : exact-order kNN-10 scores, top-10 per lane, first/second moments of the
//                           6-dim edge rows (fp64), per-cluster partials;
//   2. k_mlp1_finalize    : fixed-order sum of the partials -> per-channel mean/var of the conv output
//                           (the conv is linear, so they follow from the 6x6 input moments), folded into
//                           w' = a*w, b' = beta - a*mean;
//   3. k_mlp1_apply       : conv + affine + LeakyReLU + max_k, then wave-reduced max/mean over the lanes.
#include "engine_ctx.h"
#include "sg_common.h"
#include "knn_device.h"
#include "wave_ops.h"

namespace {

constexpr int K1 = 10;

// knn() score in the reference's fp32 operation order (model.py:31-33)
__device__ inline float knn_score(float xq, float yq, float zq, float xxq, float xc, float yc, float zc, float xxc) {
    const float t = __builtin_fmaf(zq, zc, __builtin_fmaf(yq, yc, xq * xc));
    const float inner = -2.0f * t;
    return ((-xxc) - inner) - xxq;
}

__device__ inline float sqnorm3(float x, float y, float z) { return (x * x + y * y) + z * z; }

__device__ __forceinline__ void mlp1_knn_moments_body(const float* __restrict__ samples, uint8_t* __restrict__ knn,
                                                      double* __restrict__ partial, int c) {
    const int lane = threadIdx.x;
    const float* row = samples + ((size_t)c * 64 + lane) * 6;
    float f[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) f[k] = row[k];
    const float xx = sqnorm3(f[0], f[1], f[2]);

    // top 10 in LIST form (knn_device.h): keys = (ordered score, ~candidate) as doubles of one exponent, a sorted insertion = 19 plain
    // v_min_f64 / v_max_f64 (round 2: 10 compares into wave masks + 40 selects per candidate).  Order: score descending, the EARLIER
    // candidate first among equal scores -- what the strict compares of the serial form gave.
    double kvl[K1];
#pragma unroll
    for (int t = 0; t < K1; ++t) kvl[t] = sgknn::list_empty();
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
        // candidate j is the same for every lane: v_readlane broadcasts instead of four ds_bpermute round trips
        const float s = knn_score(f[0], f[1], f[2], xx, sgw::bcast(f[0], j), sgw::bcast(f[1], j), sgw::bcast(f[2], j), sgw::bcast(xx, j));
        sgknn::list_insert<K1>(kvl, sgknn::make_key(s, j));
    }
    int bi[K1];
#pragma unroll
    for (int t = 0; t < K1; ++t) bi[t] = sgknn::list_index(kvl[t]);
    uint8_t* ko = knn + ((size_t)c * 64 + lane) * K1;
#pragma unroll
    for (int t = 0; t < K1; ++t) ko[t] = (uint8_t)bi[t];

    // edge rows: xyz := 10 * (nbr - mean_k nbr), rgb := nbr rgb  (model.py:57-60)
    double e[K1][6];
    double m0 = 0, m1 = 0, m2 = 0;
#pragma unroll
    for (int t = 0; t < K1; ++t) {
#pragma unroll
        for (int k = 0; k < 6; ++k) e[t][k] = (double)__shfl(f[k], bi[t]);
        m0 += e[t][0]; m1 += e[t][1]; m2 += e[t][2];
    }
    m0 /= K1; m1 /= K1; m2 /= K1;
    double acc[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) acc[q] = 0.0;
#pragma unroll
    for (int t = 0; t < K1; ++t) {
        e[t][0] = (e[t][0] - m0) * 10.0; e[t][1] = (e[t][1] - m1) * 10.0; e[t][2] = (e[t][2] - m2) * 10.0;
        int q = 6;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            acc[a] += e[t][a];
#pragma unroll
            for (int b = a; b < 6; ++b) acc[q++] += e[t][a] * e[t][b];
        }
    }
#pragma unroll
    for (int q = 0; q < 27; ++q) {
        const double v = sgw::wave_sum(acc[q]);                 // DPP network, fixed order
        if (lane == 0) partial[(size_t)c * 27 + q] = v;
    }
}
__global__ __launch_bounds__(64) void k_mlp1_knn_moments(const float* __restrict__ samples, uint8_t* __restrict__ knn,
                                                         double* __restrict__ partial) {
    mlp1_knn_moments_body(samples, knn, partial, blockIdx.x);
}
__global__ __launch_bounds__(64) void k_mlp1_knn_moments_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x >= c.S) return;
    mlp1_knn_moments_body(c.samples, c.m1_knn, c.m1_partial, blockIdx.x);
}

// one block of 27 x 32 threads: value q is summed by 32 lanes over clusters l, l+32, ... and the lane sums are
// combined by a fixed shuffle tree (reproducible); then thread c < 64 derives channel c's folded affine
__device__ __forceinline__ void mlp1_finalize_body(const double* __restrict__ partial, int C, const float* __restrict__ w,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float* __restrict__ folded) {
    __shared__ double mom[27];
    {
        const int q = threadIdx.x >> 5, l = threadIdx.x & 31;
        double s = 0.0;
        for (int c = l; c < C; c += 32) s += partial[(size_t)c * 27 + q];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);             // stays inside the 32-lane half
        if (l == 0) mom[q] = s / ((double)C * 64.0 * K1);
    }
    __syncthreads();
    const int ch = threadIdx.x;
    if (ch >= 64) return;
    double cov[6][6];
    int q = 6;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) { cov[a][b] = cov[b][a] = mom[q] - mom[a] * mom[b]; ++q; }
    double mean = 0.0, var = 0.0;
    for (int a = 0; a < 6; ++a) {
        mean += (double)w[ch * 6 + a] * mom[a];
        for (int b = 0; b < 6; ++b) var += (double)w[ch * 6 + a] * (double)w[ch * 6 + b] * cov[a][b];
    }
    const double a_ = (double)gamma[ch] / sqrt(var + 1e-5);
    for (int k = 0; k < 6; ++k) folded[ch * 6 + k] = (float)(a_ * (double)w[ch * 6 + k]);
    folded[384 + ch] = (float)((double)beta[ch] - a_ * mean);
}
__global__ __launch_bounds__(27 * 32) void k_mlp1_finalize(const double* __restrict__ partial, int C, const float* __restrict__ w,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ folded) {
    mlp1_finalize_body(partial, C, w, gamma, beta, folded);
}
__global__ __launch_bounds__(27 * 32) void k_mlp1_finalize_b(const sg::SlotCtx* __restrict__ cx, const float* __restrict__ w,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    mlp1_finalize_body(c.m1_partial, c.S, w, gamma, beta, c.m1_folded);
}

__device__ __forceinline__ void mlp1_apply_body(const float* __restrict__ samples, const uint8_t* __restrict__ knn,
                                                const float* __restrict__ folded, float* __restrict__ feat, int feat_stride, int c) {
    const int lane = threadIdx.x;
    const float* row = samples + ((size_t)c * 64 + lane) * 6;
    float f[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) f[k] = row[k];
    const uint8_t* ki = knn + ((size_t)c * 64 + lane) * K1;
    float e[K1][6];
    double m0 = 0, m1 = 0, m2 = 0;
    double ed[K1][3];
#pragma unroll
    for (int t = 0; t < K1; ++t) {
        const int j = ki[t];
#pragma unroll
        for (int k = 0; k < 6; ++k) e[t][k] = __shfl(f[k], j);
        ed[t][0] = e[t][0]; ed[t][1] = e[t][1]; ed[t][2] = e[t][2];
        m0 += ed[t][0]; m1 += ed[t][1]; m2 += ed[t][2];
    }
    m0 /= K1; m1 /= K1; m2 /= K1;
#pragma unroll
    for (int t = 0; t < K1; ++t) {
        e[t][0] = (float)((ed[t][0] - m0) * 10.0); e[t][1] = (float)((ed[t][1] - m1) * 10.0); e[t][2] = (float)((ed[t][2] - m2) * 10.0);
    }
    float* out = feat + (size_t)c * feat_stride;
    // h[sample][channel] goes through LDS once and lane `ch` then reduces its channel over the 64 samples serially: two LDS
    // operations per value instead of a cross-lane network per channel (64 x ~50 DPP / readlane instructions were a third of this
    // kernel).  Row stride 65: the column write and the row read are both conflict-free.
    __shared__ float hs[64 * 65];
    // the folded weights (384 + 64 floats) through LDS once: read per channel from global memory they were seven loads of one
    // address per lane and channel (the pointer comes out of a SlotCtx: flat loads), each waited for inside the channel loop
    __shared__ __attribute__((aligned(8))) float wl[448];
    // transposed on the way in: wl[k * 64 + ch], so that the weights of a channel PAIR are one 8-byte read and the pair's six FMAs per
    // neighbour are v_pk_fma_f32 (the neighbour's value feeds both halves): 6 + 2 instead of 12 + 2 instructions per (pair, neighbour),
    // the same IEEE operations in the same order per channel
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int e_ = lane + 64 * i;
        wl[e_ < 384 ? (e_ % 6) * 64 + e_ / 6 : e_] = folded[e_];
    }
    __builtin_amdgcn_wave_barrier();
    using f32x2 = __attribute__((ext_vector_type(2))) float;
#pragma unroll 2
    for (int ch = 0; ch < 64; ch += 2) {
        f32x2 w[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) w[k] = *reinterpret_cast<const f32x2*>(&wl[k * 64 + ch]);
        const f32x2 b = *reinterpret_cast<const f32x2*>(&wl[384 + ch]);
        float h0 = -INFINITY, h1 = -INFINITY;
#pragma unroll
        for (int t = 0; t < K1; ++t) {
            f32x2 y = b;
#pragma unroll
            for (int k = 0; k < 6; ++k) y = __builtin_elementwise_fma(w[k], f32x2{e[t][k], e[t][k]}, y);
            h0 = fmaxf(h0, y.x);                                 // max over k (model.py:76) ...
            h1 = fmaxf(h1, y.y);
        }
        // ... BEFORE LeakyReLU(0.2): x -> max(x, 0.2 x) is non-decreasing (also in fp32: rounding is monotone), so the maximum of the
        // activations is the activation of the maximum, bit for bit -- two instructions per channel instead of two per (channel, neighbour)
        hs[ch * 65 + lane] = fmaxf(h0, 0.2f * h0);
        hs[(ch + 1) * 65 + lane] = fmaxf(h1, 0.2f * h1);
    }
    __builtin_amdgcn_wave_barrier();
    {
        const float* row = hs + lane * 65;                        // lane = channel
        float mx = -INFINITY;
        double sm = 0.0;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) {
            const float v = row[i];
            mx = fmaxf(mx, v);
            sm += (double)v;
        }
        out[lane] = mx;                                          // model.py:77-79: [max | mean] over the 64 samples
        out[64 + lane] = (float)(sm / 64.0);
    }
}
__global__ __launch_bounds__(64) void k_mlp1_apply(const float* __restrict__ samples, const uint8_t* __restrict__ knn,
                                                   const float* __restrict__ folded, float* __restrict__ feat, int feat_stride) {
    mlp1_apply_body(samples, knn, folded, feat, feat_stride, blockIdx.x);
}
__global__ __launch_bounds__(64) void k_mlp1_apply_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x >= c.S) return;
    mlp1_apply_body(c.samples, c.m1_knn, c.m1_folded, c.feat1, 128, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// Training step: backward of MLP1 w.r.t. its parameters (the samples carry no gradient).
// BatchNorm2d is differentiated with its batch statistics over all C*64*10 rows; it is the only BatchNorm of the stack, so the
// dense part of its backward folds into the 6x6 input moments (see kernels_train_edge.hip's header):
//     dW = gamma/sigma ( sum da e^T - mean(da) sum e^T - mean(da xhat) sum xhat e^T ),   sum xhat e^T = (W See - mu Se^T) / sigma.
// `da` is nonzero on one of the 10 rows of every (sample, channel): the mean half of [max | mean] reaches every sample.
//   1. k_mlp1_knn_moments (the forward's)   2. k_mlp1_bwd_stats: mu, 1/sigma per channel + the reduced moments
//   3. k_mlp1_bwd_rows: one wave per cluster, per-cluster partials [64][8] = d beta, d gamma, sum da e[0..5]   4. k_mlp1_bwd_fold
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(27 * 32) void k_mlp1_bwd_stats(const double* __restrict__ partial, int C, const float* __restrict__ w,
                                                            double* __restrict__ mom_out, float* __restrict__ cst, float* __restrict__ stats_out) {
    __shared__ double mom[27];
    {
        const int q = threadIdx.x >> 5, l = threadIdx.x & 31;
        double s = 0.0;
        for (int c = l; c < C; c += 32) s += partial[(size_t)c * 27 + q];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (l == 0) { mom[q] = s / ((double)C * 64.0 * K1); mom_out[q] = mom[q]; }
    }
    __syncthreads();
    const int ch = threadIdx.x;
    if (ch >= 64) return;
    double cov[6][6];
    int q = 6;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) { cov[a][b] = cov[b][a] = mom[q] - mom[a] * mom[b]; ++q; }
    double mean = 0.0, var = 0.0;
    for (int a = 0; a < 6; ++a) {
        mean += (double)w[ch * 6 + a] * mom[a];
        for (int b = 0; b < 6; ++b) var += (double)w[ch * 6 + a] * (double)w[ch * 6 + b] * cov[a][b];
    }
    var = fmax(var, 0.0);
    cst[ch] = (float)mean;
    cst[64 + ch] = (float)(1.0 / sqrt(var + 1e-5));
    if (stats_out) { stats_out[ch] = (float)mean; stats_out[64 + ch] = (float)var; }
}

__global__ __launch_bounds__(64) void k_mlp1_bwd_rows(const float* __restrict__ samples, const uint8_t* __restrict__ knn, const float* __restrict__ w,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ cst,
                                                      const float* __restrict__ gfeat, int g_stride, double* __restrict__ rows_partial) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const float* row = samples + ((size_t)c * 64 + lane) * 6;
    float f[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) f[k] = row[k];
    const uint8_t* ki = knn + ((size_t)c * 64 + lane) * K1;
    float e[K1][6];
    double m0 = 0, m1 = 0, m2 = 0;
    double ed[K1][3];
#pragma unroll
    for (int t = 0; t < K1; ++t) {                                // the edge rows exactly as k_mlp1_apply builds them
        const int j = ki[t];
#pragma unroll
        for (int k = 0; k < 6; ++k) e[t][k] = __shfl(f[k], j);
        ed[t][0] = e[t][0]; ed[t][1] = e[t][1]; ed[t][2] = e[t][2];
        m0 += ed[t][0]; m1 += ed[t][1]; m2 += ed[t][2];
    }
    m0 /= K1; m1 /= K1; m2 /= K1;
#pragma unroll
    for (int t = 0; t < K1; ++t) {
        e[t][0] = (float)((ed[t][0] - m0) * 10.0); e[t][1] = (float)((ed[t][1] - m1) * 10.0); e[t][2] = (float)((ed[t][2] - m2) * 10.0);
    }
    const float* g = gfeat + (size_t)c * g_stride;
    double* out = rows_partial + (size_t)c * 512;
    for (int ch = 0; ch < 64; ++ch) {
        const float w0 = w[ch * 6 + 0], w1 = w[ch * 6 + 1], w2 = w[ch * 6 + 2], w3 = w[ch * 6 + 3], w4 = w[ch * 6 + 4], w5 = w[ch * 6 + 5];
        const float mu = cst[ch], inv = cst[64 + ch], ga = gamma[ch], be = beta[ch];
        float best = -INFINITY, bx = 0.f, ba = 0.f;
        float be_[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < K1; ++t) {
            float y = w0 * e[t][0];
            y = __builtin_fmaf(w1, e[t][1], y); y = __builtin_fmaf(w2, e[t][2], y);
            y = __builtin_fmaf(w3, e[t][3], y); y = __builtin_fmaf(w4, e[t][4], y); y = __builtin_fmaf(w5, e[t][5], y);
            const float xh = (y - mu) * inv;
            const float a = xh * ga + be;
            const float h = fmaxf(a, 0.2f * a);
            if (h > best) {                                       // first maximum over the 10 neighbours (torch.max)
                best = h; bx = xh; ba = a;
#pragma unroll
                for (int k = 0; k < 6; ++k) be_[k] = e[t][k];
            }
        }
        float mv = best;
        int mi = lane;
        sgw::wave_argmax(mv, mi);                                 // first maximal sample (model.py:77)
        const float dh = (lane == mi ? g[ch] : 0.f) + g[64 + ch] * (1.0f / 64.0f);
        const double da = (double)(dh * (ba > 0.f ? 1.f : 0.2f));
        const double v[8] = {da, da * (double)bx, da * (double)be_[0], da * (double)be_[1], da * (double)be_[2], da * (double)be_[3],
                             da * (double)be_[4], da * (double)be_[5]};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double r = sgw::wave_sum(v[q]);
            if (lane == 0) out[ch * 8 + q] = r;
        }
    }
}

// 512 threads: thread (ch, q) sums the clusters' partials in order; then dW, d gamma, d beta
__global__ __launch_bounds__(512) void k_mlp1_bwd_fold(const double* __restrict__ rows_partial, int nparts, int C, const double* __restrict__ mom,
                                                       const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ cst,
                                                       float* __restrict__ gw, float* __restrict__ gg, float* __restrict__ gb) {
    __shared__ double acc[512];
    const int t = threadIdx.x;
    double a = 0.0;
    for (int c = 0; c < nparts; ++c) a += rows_partial[(size_t)c * 512 + t];
    acc[t] = a;
    __syncthreads();
    const double rows = (double)C * 64.0 * K1;                    // mom holds MEANS (sum / rows)
    if (t < 64) { gb[t] = (float)acc[t * 8]; gg[t] = (float)acc[t * 8 + 1]; }
    if (t < 384) {
        const int ch = t / 6, j = t - ch * 6;
        const double inv = (double)cst[64 + ch], mu = (double)cst[ch];
        auto m2 = [&](int p, int q_) {                            // mean of e_p e_q from the packed upper triangle
            const int lo = p < q_ ? p : q_, hi = p < q_ ? q_ : p;
            return mom[6 + lo * 6 - lo * (lo - 1) / 2 + (hi - lo)];
        };
        double wsee = 0.0;
        for (int q_ = 0; q_ < 6; ++q_) wsee += (double)w[ch * 6 + q_] * m2(q_, j);
        const double xe = (wsee - mu * mom[j]) * inv * rows;      // sum_r xhat[r][ch] e[r][j]
        gw[t] = (float)((double)gamma[ch] * inv * (acc[ch * 8 + 2 + j] - acc[ch * 8] / rows * (mom[j] * rows) - acc[ch * 8 + 1] / rows * xe));
    }
}

}  // namespace

namespace sg {

int b_mlp1(const SlotCtx* d_ctx, const float* d_w, const float* d_g, const float* d_b, const BatchDims& bd, hipStream_t st) {
    if (bd.nslots == 0 || bd.max_S == 0) return SG_OK;
    k_mlp1_knn_moments_b<<<dim3(bd.max_S, bd.nslots), 64, 0, st>>>(d_ctx);
    k_mlp1_finalize_b<<<dim3(1, bd.nslots), 27 * 32, 0, st>>>(d_ctx, d_w, d_g, d_b);
    k_mlp1_apply_b<<<dim3(bd.max_S, bd.nslots), 64, 0, st>>>(d_ctx);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

}  // namespace sg

extern "C" {

size_t sg_mlp1_ws_bytes(int C) {
    const size_t c = (size_t)std::max(C, 1);
    return sg::align_up(c * 64 * K1) + sg::align_up(c * 27 * 8) + sg::align_up(448 * 4);
}

int sg_mlp1_forward(const float* d_samples, int C, const float* d_w, const float* d_gamma, const float* d_beta, float* d_feat,
                    int feat_stride, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(C >= 0 && feat_stride >= 128 && d_ws, "sg_mlp1_forward: bad arguments");
    if (C == 0) return SG_OK;
    sg::Carver cv(d_ws, ws_bytes);
    uint8_t* knn = cv.take<uint8_t>((size_t)C * 64 * K1);
    double* partial = cv.take<double>((size_t)C * 27);
    float* folded = cv.take<float>(448);
    if (!cv.ok) return sg::fail(SG_ENOMEM, "sg_mlp1_forward: workspace too small (%zu < %zu)", ws_bytes, sg_mlp1_ws_bytes(C));
    hipStream_t st = sg::as_stream(stream);
    k_mlp1_knn_moments<<<C, 64, 0, st>>>(d_samples, knn, partial);
    k_mlp1_finalize<<<1, 27 * 32, 0, st>>>(partial, C, d_w, d_gamma, d_beta, folded);
    k_mlp1_apply<<<C, 64, 0, st>>>(d_samples, knn, folded, d_feat, feat_stride);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

size_t sg_mlp1_backward_ws_bytes(int C) {
    const size_t c = (size_t)std::max(C, 1);
    return sg::align_up(c * 64 * K1) + sg::align_up(c * 27 * 8) + sg::align_up(c * 512 * 8) + sg::align_up(27 * 8) + sg::align_up(128 * 4) + sg::align_up(512 * 8);
}

int sg_mlp1_backward(const float* d_samples, int C, const float* d_w, const float* d_gamma, const float* d_beta, const float* d_gfeat,
                     int g_stride, float* d_gw, float* d_gg, float* d_gb, float* d_bn_stats, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(C > 0 && g_stride >= 128 && d_samples && d_w && d_gamma && d_beta && d_gfeat && d_gw && d_gg && d_gb && d_ws, "sg_mlp1_backward: bad arguments");
    sg::Carver cv(d_ws, ws_bytes);
    uint8_t* knn = cv.take<uint8_t>((size_t)C * 64 * K1);
    double* partial = cv.take<double>((size_t)C * 27);
    double* rows_partial = cv.take<double>((size_t)C * 512);
    double* mom = cv.take<double>(27);
    float* cst = cv.take<float>(128);
    double* red = cv.take<double>(512);
    if (!cv.ok) return sg::fail(SG_ENOMEM, "sg_mlp1_backward: workspace too small (%zu < %zu)", ws_bytes, sg_mlp1_backward_ws_bytes(C));
    hipStream_t st = sg::as_stream(stream);
    k_mlp1_knn_moments<<<C, 64, 0, st>>>(d_samples, knn, partial);
    k_mlp1_bwd_stats<<<1, 27 * 32, 0, st>>>(partial, C, d_w, mom, cst, d_bn_stats);
    k_mlp1_bwd_rows<<<C, 64, 0, st>>>(d_samples, knn, d_w, d_gamma, d_beta, cst, d_gfeat, g_stride, rows_partial);
    if (int rc = sg::reduce_partials(rows_partial, C, 512, 512, red, st)) return rc;
    k_mlp1_bwd_fold<<<1, 512, 0, st>>>(red, 1, C, mom, d_w, d_gamma, cst, d_gw, d_gg, d_gb);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

}  // extern "C"
