// Training step: backward of get_graph_feature2 + MLP2 / MLP3 (model.py:83-138) -- conv1x1 18->64
// (+ conv1x1 64->64), BatchNorm2d with BATCH statistics over all N*k rows, LeakyReLU(0.2), max over the k neighbours.
// The inputs (point coordinates / colours) carry no gradient, so the op returns parameter gradients only.
//
// What makes it a dense problem: with batch statistics,  dy = gamma/sigma (da - mean(da) - xhat mean(da xhat)).  `da` is
// nonzero on ONE of the k rows of a (point, channel) that carries output gradient, but the two mean terms reach every one
// of the N*k rows.  For the LAST BatchNorm of the stack that dense part is an affine function of the layer's own input
// and folds into input moments:
//     sum_r dy[r] e[r]^T = gamma/sigma ( sum_r da[r] e[r]^T - mean(da) sum_r e[r]^T - mean(da xhat) sum_r xhat[r] e[r]^T ),
//     sum_r xhat[r] e[r]^T = (W See - mu Se^T) / sigma                with Se = sum e, See = sum e e^T   (fp64).
// MLP2 (one conv) therefore needs no dense backward pass at all.  MLP3's second BatchNorm sits behind a LeakyReLU, so its
// dense part has to be carried through explicitly: one dense pass that recomputes conv1 -> h1 -> conv2 per 64-row tile
// and forms dy2, dW2 += dy2^T h1, dh1 = dy2 W2, da1 = dh1 lrelu'(a1), sum da1 e^T -- then the moment identity above for BN1.
//
// Passes.  Both: moments of e (the forward's own structured-moments kernel, kernels_edgeconv.hip) -> BN1 statistics.
//   layers == 1: one sparse pass (k_eb_sparse1), fold.
//   layers == 2: BN2 statistics + d beta2, d gamma2 + the arg-max k of every (point, channel) that carries gradient -- from the
//                forward's statistics and a sparse pass (k_eb_sparse2; what the training step does), or, without statistics,
//                from a dense forward pass (k_eb_forward, k_eb_last_bn) -- then the dense backward (k_eb_backward_mfma), fold.
// Arithmetic: fp32 (MFMA in the dense backward, FMAs on LDS tiles elsewhere), running sums flushed to fp64 every 16 tiles, block
// partials reduced in a fixed order (deterministic).  Conditioning: the XYZ of the x_i half is taken relative to row 0
// (BatchNorm is invariant to that shift, and sum_r dy[r] = 0 makes dW invariant too), like the forward kernels.
#include "sg_common.h"

namespace {

constexpr int kR = 64;        // rows per tile
constexpr int kES = 20;       // row stride of the edge-feature tile (18 used)
constexpr int kHS = 68;       // row stride of the 64-wide tiles
constexpr int kThreads = 256;
constexpr float kSlope = 0.2f;
constexpr double kBnEps = 1e-5;
constexpr int kMom = 420;     // 20 x 20 second moments + 20 sums
constexpr int kPart3 = 4096 + 64 * kES + 128;     // dense backward: dW2 | sum da1 e^T | sum da1, sum da1 xhat1
// per-channel constants (floats) the passes share
enum { MU1 = 0, INV1 = 64, MU2 = 128, INV2 = 192, DB2 = 256, DG2 = 320, X0 = 384, kCst = 388 };

struct Tile {
    float E[kR * kES];
    float H1[kR * kHS];
    float D2[kR * kHS];
    float W1t[kES * 64];      // [j][c]
    float W2[64 * kHS];       // [o][i]
    float G[4 * 64];          // output gradient of the tile's points (P <= 3 for k = 20; up to 4)
    int argk[4 * 64];
    int rown[kR];
};                                                                    // 64.8 KB: just inside the static LDS limit
__device__ __forceinline__ double* red_of(Tile& s) { return reinterpret_cast<double*>(s.H1); }   // [16][64] doubles, after the tile loop

__device__ __forceinline__ float lrelu(float a) { return a > 0.f ? a : kSlope * a; }
__device__ __forceinline__ float dlrelu(float a) { return a > 0.f ? 1.f : kSlope; }     // torch: x > 0 ? g : g * slope

// rows r < P*K of tile `tile`: point n = tile*P + r/K, neighbour k = r%K.  Rows past N (and r >= P*K) are zero, rown = -1.
// The two dependent gathers of a row (kNN entry, then that neighbour's 9 values) are requested ONE TILE AHEAD into registers
// (`fetch_edge_rows`) and turned into the LDS tile at the top of the next iteration (`store_edge_rows`): a block runs ~100 tiles
// back to back and used to wait out both round trips in front of every one of them (~17 us per tile for ~1 us of arithmetic).
struct RowFetch {
    float xi[9], xj[9];
    int n;
};
__device__ __forceinline__ void fetch_edge_rows(RowFetch& f, const float* __restrict__ x9, const int32_t* __restrict__ knn, int N, int K, int P,
                                                int tile, int ntiles) {
    const int t = threadIdx.x;
    f.n = -1;
    if (t < kR && tile < ntiles) {
        const int p = t / K, k = t - p * K, n = tile * P + p;
        if (p < P && n < N) {
            const int j = knn[(size_t)n * K + k];
            const float* xi = x9 + (size_t)n * 12;
            const float* xj = x9 + (size_t)j * 12;
#pragma unroll
            for (int c = 0; c < 9; ++c) { f.xi[c] = xi[c]; f.xj[c] = xj[c]; }
            f.n = n;
        }
    }
}
__device__ __forceinline__ void store_edge_rows(Tile& s, const RowFetch& f, const float* __restrict__ cst) {
    const int t = threadIdx.x;
    if (t < kR) {
        float v[kES];
#pragma unroll
        for (int j = 0; j < kES; ++j) v[j] = 0.f;
        if (f.n >= 0) {
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                v[c] = f.xj[c] - f.xi[c];
                v[9 + c] = f.xi[c];
            }
            v[9] -= cst[X0]; v[10] -= cst[X0 + 1]; v[11] -= cst[X0 + 2];
        }
        float4* dst = reinterpret_cast<float4*>(&s.E[t * kES]);
#pragma unroll
        for (int q = 0; q < kES / 4; ++q) dst[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        s.rown[t] = f.n;
    }
}

__device__ __forceinline__ void stage_weights(Tile& s, const float* __restrict__ w1, const float* __restrict__ w2) {
    for (int i = threadIdx.x; i < kES * 64; i += kThreads) {
        const int j = i >> 6, c = i & 63;
        s.W1t[i] = j < 18 ? w1[c * 18 + j] : 0.f;
    }
    if (w2)
        for (int i = threadIdx.x; i < 64 * 64; i += kThreads) s.W2[(i >> 6) * kHS + (i & 63)] = w2[i];
}

// y1 block of thread (tr, tc): rows 4 tr + i, columns 4 tc + j
__device__ __forceinline__ void conv1_block(const Tile& s, int tr, int tc, float (&acc)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
#pragma unroll
    for (int q = 0; q < kES / 4; ++q) {
        float4 e[4], w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = *reinterpret_cast<const float4*>(&s.E[(4 * tr + i) * kES + 4 * q]);
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const float4*>(&s.W1t[(4 * q + u) * 64 + 4 * tc]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ev[4] = {e[i].x, e[i].y, e[i].z, e[i].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[i][0] = fmaf(ev[u], w[u].x, acc[i][0]);
                acc[i][1] = fmaf(ev[u], w[u].y, acc[i][1]);
                acc[i][2] = fmaf(ev[u], w[u].z, acc[i][2]);
                acc[i][3] = fmaf(ev[u], w[u].w, acc[i][3]);
            }
        }
    }
}

// y2 block of thread (tr, tc): rows 4 tr + i, columns tc + 16 j   (W2 rows are read along their contiguous input index)
__device__ __forceinline__ void conv2_block(const Tile& s, int tr, int tc, float (&acc)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
#pragma unroll 2
    for (int k = 0; k < 64; k += 4) {
        float4 h[4], w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = *reinterpret_cast<const float4*>(&s.H1[(4 * tr + i) * kHS + k]);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const float4*>(&s.W2[(tc + 16 * j) * kHS + k]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = fmaf(h[i].w, w[j].w, fmaf(h[i].z, w[j].z, fmaf(h[i].y, w[j].y, fmaf(h[i].x, w[j].x, acc[i][j]))));
    }
}

// ordered block reduction of one double per thread over the 16 threads that share (t & 15) [column owner = tc] or (t & 63)
// -> `out[c]` for c < ncol; `part` = how many threads share a column (16: columns by tc..., 4: columns by t & 63)
__device__ __forceinline__ void store_partial(double* __restrict__ dst, const double* red, int ncol, int nshare, int t) {
    if (t < ncol) {
        double a = 0.0;
        for (int r = 0; r < nshare; ++r) a += red[r * 64 + t];
        dst[t] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pass 0: moments of the edge features
// ---------------------------------------------------------------------------------------------------------------
// the forward's structured moments (kernels_edgeconv.hip: per point a = sum_j d_j and D = sum_j d_j d_j^T over its 20 slots, one
// gather pass, ~6x cheaper than accumulating all 18 x 18 products per row) -> this file's [20 x 20 | 20] layout
__global__ void k_eb_unpack_moments(const double* __restrict__ m189, double* __restrict__ out) {
    const int t = threadIdx.x;
    if (t >= kMom) return;
    auto tri = [](int k, int l) {                               // index of (k <= l) in a 9 x 9 upper triangle stored row by row
        return k * 9 - k * (k - 1) / 2 + (l - k);
    };
    double v = 0.0;
    if (t >= 400) {
        const int i = t - 400;
        v = i < 18 ? m189[i] : 0.0;                              // [a | K x_i]
    } else {
        const int i = t / kES, j = t - i * kES;
        if (i < 18 && j < 18) {
            if (i < 9 && j < 9) v = m189[18 + tri(min(i, j), max(i, j))];            // d d^T
            else if (i < 9) v = m189[63 + 9 * i + (j - 9)];                           // d x^T
            else if (j < 9) v = m189[63 + 9 * j + (i - 9)];                           // x d^T
            else v = m189[144 + tri(min(i, j) - 9, max(i, j) - 9)];                    // x x^T
        }
    }
    out[t] = v;
}

// BN1 statistics from the moments: mu1 = W1 m, var1 = w^T Cov w  (fp64).  mom_out keeps the reduced moments for the fold.
__global__ void k_eb_fold1(const double* __restrict__ partial, int nblocks, double rows, const float* __restrict__ w1, float* __restrict__ cst,
                           double* __restrict__ mom_out, float* __restrict__ stats_out) {
    __shared__ double mom[kMom];
    for (int i = threadIdx.x; i < kMom; i += blockDim.x) {
        double a = 0.0;
        for (int b = 0; b < nblocks; ++b) a += partial[(size_t)b * kMom + i];
        mom[i] = a;
        mom_out[i] = a;
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c < 64) {
        double mu = 0.0, ex2 = 0.0, shift = 0.0;
        for (int i = 0; i < 18; ++i) {
            const double wi = w1[c * 18 + i];
            mu += wi * mom[400 + i] / rows;
            double row = 0.0;
            for (int j = 0; j < 18; ++j) row += (double)w1[c * 18 + j] * mom[i * kES + j];
            ex2 += wi * row / rows;
        }
        for (int a = 0; a < 3; ++a) shift += (double)w1[c * 18 + 9 + a] * (double)cst[X0 + a];
        const double var = fmax(ex2 - mu * mu, 0.0);
        cst[MU1 + c] = (float)mu;
        cst[INV1 + c] = (float)(1.0 / sqrt(var + kBnEps));
        if (stats_out) { stats_out[c] = (float)(mu + shift); stats_out[64 + c] = (float)var; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pass 1 (layers == 2, standalone operator without the forward's statistics): dense forward.  sum y2, sum y2^2 (block partials
// [64 | 64]) and, per (point, channel), the extreme of y2 over k in the direction of sign(gamma2) + the FIRST k that attains it.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads, 2) k_eb_forward(const float* __restrict__ x9, const int32_t* __restrict__ knn, int N, int K, int P, int ntiles,
                                                         const float* __restrict__ w1, const float* __restrict__ g1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ g2, const float* __restrict__ cst,
                                                         float* __restrict__ ext, uint8_t* __restrict__ argk, double* __restrict__ partial) {
    __shared__ Tile s;
    const int t = threadIdx.x, tr = t >> 4, tc = t & 15;
    stage_weights(s, w1, w2);
    float mu1[4], sc1[4], sh1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * tc + j;
        mu1[j] = cst[MU1 + c];
        sc1[j] = cst[INV1 + c] * g1[c];
        sh1[j] = b1[c];
    }
    double sy[4] = {0.0, 0.0, 0.0, 0.0}, sq[4] = {0.0, 0.0, 0.0, 0.0};
    const float* gl = g2;
    RowFetch rf;
    fetch_edge_rows(rf, x9, knn, N, K, P, blockIdx.x, ntiles);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        store_edge_rows(s, rf, cst);
        fetch_edge_rows(rf, x9, knn, N, K, P, tile + gridDim.x, ntiles);
        __syncthreads();
        float acc[4][4];
        conv1_block(s, tr, tc, acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 h;
            const bool valid = s.rown[4 * tr + i] >= 0;
            h.x = valid ? lrelu((acc[i][0] - mu1[0]) * sc1[0] + sh1[0]) : 0.f;
            h.y = valid ? lrelu((acc[i][1] - mu1[1]) * sc1[1] + sh1[1]) : 0.f;
            h.z = valid ? lrelu((acc[i][2] - mu1[2]) * sc1[2] + sh1[2]) : 0.f;
            h.w = valid ? lrelu((acc[i][3] - mu1[3]) * sc1[3] + sh1[3]) : 0.f;
            *reinterpret_cast<float4*>(&s.H1[(4 * tr + i) * kHS + 4 * tc]) = h;
        }
        __syncthreads();
        const float* src = s.D2;
        {
            conv2_block(s, tr, tc, acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool valid = s.rown[4 * tr + i] >= 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s.D2[(4 * tr + i) * kHS + tc + 16 * j] = acc[i][j];
                    if (valid) { sy[j] += (double)acc[i][j]; sq[j] += (double)acc[i][j] * (double)acc[i][j]; }
                }
            }
            __syncthreads();
        }
        if (t < P * 64) {
            const int p = t >> 6, o = t & 63, n = tile * P + p;
            if (n < N) {
                const float go = gl[o];
                const bool up = go >= 0.f;
                float best = src[(p * K) * kHS + o];
                int bk = 0;
                for (int k = 1; k < (go == 0.f ? 1 : K); ++k) {          // gamma == 0: every row ties, torch.max keeps the first
                    const float v = src[(p * K + k) * kHS + o];
                    if (up ? v > best : v < best) { best = v; bk = k; }
                }
                ext[(size_t)n * 64 + o] = best;
                argk[(size_t)n * 64 + o] = (uint8_t)bk;
            }
        }
    }
    {
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) red_of(s)[tr * 64 + tc + 16 * j] = which ? sq[j] : sy[j];
            __syncthreads();
            store_partial(partial + (size_t)blockIdx.x * 128 + which * 64, red_of(s), 64, 16, t);
        }
    }
}

__global__ void k_eb_fold2(const double* __restrict__ partial, int nblocks, double rows, float* __restrict__ cst, float* __restrict__ stats_out) {
    const int c = threadIdx.x;
    if (c >= 64) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nblocks; ++b) { s += partial[(size_t)b * 128 + c]; q += partial[(size_t)b * 128 + 64 + c]; }
    const double mu = s / rows, var = fmax(q / rows - mu * mu, 0.0);
    cst[MU2 + c] = (float)mu;
    cst[INV2 + c] = (float)(1.0 / sqrt(var + kBnEps));
    if (stats_out) { stats_out[128 + c] = (float)mu; stats_out[192 + c] = (float)var; }
}

// ---------------------------------------------------------------------------------------------------------------
// pass 2 (layers == 2, standalone operator; elementwise over [N,64]): da of the LAST BatchNorm at the extreme row -> d beta2,
// d gamma2 (block partials [64 | 64])
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_eb_last_bn(int N, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ cst,
                                                         const float* __restrict__ ext, const float* __restrict__ gout, double* __restrict__ partial) {
    __shared__ double red[4 * 64];
    const int t = threadIdx.x, p = t >> 6, o = t & 63;
    const float mu = cst[MU2 + o], inv = cst[INV2 + o], g = gamma[o], b = beta[o];
    double sdb = 0.0, sdg = 0.0;
    for (int n = blockIdx.x * 4 + p; n < N; n += gridDim.x * 4) {
        const float go = gout[(size_t)n * 64 + o];
        if (go == 0.f) continue;
        const float xh = (ext[(size_t)n * 64 + o] - mu) * inv;
        const float d = go * dlrelu(xh * g + b);
        sdb += (double)d;
        sdg += (double)d * (double)xh;
    }
    double* dst = partial + (size_t)blockIdx.x * 128;
#pragma unroll 1
    for (int q = 0; q < 2; ++q) {
        __syncthreads();
        red[p * 64 + o] = q ? sdg : sdb;
        __syncthreads();
        if (t < 64) dst[q * 64 + t] = ((red[t] + red[64 + t]) + red[128 + t]) + red[192 + t];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// layers == 1 (MLP2) needs no dense pass at all: the output gradient is nonzero on at most C x 64 (point, channel) pairs (the
// arg-maxima of the point -> cluster max), and for such a pair the 20 pre-activations of the point are recomputed on the spot:
// one wave per point, lane = channel (its weight row in registers), the point's 20 edge rows in LDS.  Points without gradient
// (three quarters of them) cost one load.  Block partials as k_eb_last_bn<1>: [d beta 64 | d gamma 64 | sum da e^T 64 x 20].
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_eb_sparse1(const float* __restrict__ x9, const int32_t* __restrict__ knn, int N, int K,
                                                         const float* __restrict__ w1, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ cst, const float* __restrict__ gout, double* __restrict__ partial) {
    __shared__ float E[4][32][kES];
    __shared__ double red[4 * 64];
    const int t = threadIdx.x, p = t >> 6, o = t & 63;
    const float mu = cst[MU1 + o], inv = cst[INV1 + o], g = gamma[o], b = beta[o];
    const float x0[3] = {cst[X0], cst[X0 + 1], cst[X0 + 2]};
    float w[18];
#pragma unroll
    for (int j = 0; j < 18; ++j) w[j] = w1[o * 18 + j];
    double sdb = 0.0, sdg = 0.0, ge[18];
#pragma unroll
    for (int j = 0; j < 18; ++j) ge[j] = 0.0;
    for (int n = blockIdx.x * 4 + p; n < N; n += gridDim.x * 4) {            // wave-uniform: one point per wave and iteration
        const float go = gout[(size_t)n * 64 + o];
        if (!__any(go != 0.f)) continue;
        __builtin_amdgcn_wave_barrier();
        if (o < K) {
            const int j = knn[(size_t)n * K + o];
            const float* xi = x9 + (size_t)n * 12;
            const float* xj = x9 + (size_t)j * 12;
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const float a = xi[c];
                E[p][o][c] = xj[c] - a;
                E[p][o][9 + c] = c < 3 ? a - x0[c] : a;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (go != 0.f) {
            float best = 0.f;
            int bk = 0;
            for (int k = 0; k < (g == 0.f ? 1 : K); ++k) {                   // gamma == 0: every row ties, torch.max keeps the first
                float y = 0.f;
#pragma unroll
                for (int j = 0; j < 18; ++j) y = fmaf(E[p][k][j], w[j], y);
                if (k == 0 || (g >= 0.f ? y > best : y < best)) { best = y; bk = k; }
            }
            const float xh = (best - mu) * inv;
            const float d = go * dlrelu(xh * g + b);
            sdb += (double)d;
            sdg += (double)d * (double)xh;
#pragma unroll
            for (int j = 0; j < 18; ++j) ge[j] += (double)d * (double)E[p][bk][j];
        }
    }
    double* dst = partial + (size_t)blockIdx.x * (128 + 64 * kES);
#pragma unroll 1
    for (int q = 0; q < 20; ++q) {
        __syncthreads();
        double v = q == 0 ? sdb : q == 1 ? sdg : 0.0;
        if (q >= 2) {
#pragma unroll
            for (int j = 0; j < 18; ++j) v = (q - 2 == j) ? ge[j] : v;
        }
        red[p * 64 + o] = v;
        __syncthreads();
        if (t < 64) {
            const double a = ((red[t] + red[64 + t]) + red[128 + t]) + red[192 + t];
            if (q < 2) dst[q * 64 + t] = a;
            else dst[128 + t * kES + (q - 2)] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// layers == 2 with the forward's BN2 statistics at hand (the training step's tape): the dense FORWARD pass is not needed -- what
// it produced is only read at (point, channel) pairs that carry gradient.  One wave per such point: its 20 edge rows and
// h1 = LReLU(BN1(conv1)) [20 x 64] in LDS, then every lane whose channel carries gradient evaluates its 20 conv2 outputs (same
// FMA order as the dense kernels: bit-equal y2), takes the extreme in the direction of sign(gamma2) and its first k.
// Writes argk for those pairs (the dense backward reads nothing else) and block partials [d beta2 64 | d gamma2 64].
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_eb_sparse2(const float* __restrict__ x9, const int32_t* __restrict__ knn, int N, int K,
                                                         const float* __restrict__ w1, const float* __restrict__ g1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ g2, const float* __restrict__ b2,
                                                         const float* __restrict__ cst, const float* __restrict__ gout, uint8_t* __restrict__ argk,
                                                         double* __restrict__ partial) {
    __shared__ float E[4][32][kES];
    __shared__ float H[4][32][64];
    __shared__ double red[4 * 64];
    const int t = threadIdx.x, p = t >> 6, o = t & 63;
    const float mu1 = cst[MU1 + o], sc1 = cst[INV1 + o] * g1[o], sh1 = b1[o];
    const float mu2 = cst[MU2 + o], inv2 = cst[INV2 + o], ga2 = g2[o], be2 = b2[o];
    const float x0[3] = {cst[X0], cst[X0 + 1], cst[X0 + 2]};
    float w[18];
#pragma unroll
    for (int j = 0; j < 18; ++j) w[j] = w1[o * 18 + j];
    double sdb = 0.0, sdg = 0.0;
    for (int n = blockIdx.x * 4 + p; n < N; n += gridDim.x * 4) {            // wave-uniform: one point per wave and iteration
        const float go = gout[(size_t)n * 64 + o];
        if (!__any(go != 0.f)) continue;
        __builtin_amdgcn_wave_barrier();
        if (o < K) {
            const int j = knn[(size_t)n * K + o];
            const float* xi = x9 + (size_t)n * 12;
            const float* xj = x9 + (size_t)j * 12;
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const float a = xi[c];
                E[p][o][c] = xj[c] - a;
                E[p][o][9 + c] = c < 3 ? a - x0[c] : a;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int k = 0; k < K; ++k) {                                        // h1[k][o], the dense kernels' expression and FMA order
            float y = 0.f;
#pragma unroll
            for (int j = 0; j < 18; ++j) y = fmaf(E[p][k][j], w[j], y);
            H[p][k][o] = lrelu((y - mu1) * sc1 + sh1);
        }
        __builtin_amdgcn_wave_barrier();
        if (go != 0.f) {
            float best = 0.f;
            int bk = 0;
            const float4* wr = reinterpret_cast<const float4*>(w2 + (size_t)o * 64);
            for (int k = 0; k < (ga2 == 0.f ? 1 : K); ++k) {
                float y = 0.f;
#pragma unroll 4
                for (int c = 0; c < 16; ++c) {
                    const float4 wv = wr[c];
                    const float4 hv = *reinterpret_cast<const float4*>(&H[p][k][4 * c]);
                    y = fmaf(hv.w, wv.w, fmaf(hv.z, wv.z, fmaf(hv.y, wv.y, fmaf(hv.x, wv.x, y))));
                }
                if (k == 0 || (ga2 >= 0.f ? y > best : y < best)) { best = y; bk = k; }
            }
            argk[(size_t)n * 64 + o] = (uint8_t)bk;
            const float xh = (best - mu2) * inv2;
            const float d = go * dlrelu(xh * ga2 + be2);
            sdb += (double)d;
            sdg += (double)d * (double)xh;
        }
    }
    double* dst = partial + (size_t)blockIdx.x * 128;
#pragma unroll 1
    for (int q = 0; q < 2; ++q) {
        __syncthreads();
        red[p * 64 + o] = q ? sdg : sdb;
        __syncthreads();
        if (t < 64) dst[q * 64 + t] = ((red[t] + red[64 + t]) + red[128 + t]) + red[192 + t];
    }
}

// BN2 statistics handed in (batch mean | biased variance) -> the constants the passes share
__global__ void k_eb_set_bn2(const float* __restrict__ stats_in, float* __restrict__ cst, float* __restrict__ stats_out) {
    const int c = threadIdx.x;
    if (c >= 64) return;
    cst[MU2 + c] = stats_in[c];
    cst[INV2 + c] = (float)(1.0 / sqrt((double)stats_in[64 + c] + kBnEps));
    if (stats_out) { stats_out[128 + c] = stats_in[c]; stats_out[192 + c] = stats_in[64 + c]; }
}

// d beta2, d gamma2 -> outputs + the per-row constants of the dense backward (mean terms of BN2's backward)
__global__ void k_eb_fold3(const double* __restrict__ partial, int nblocks, double rows, float* __restrict__ cst, float* __restrict__ gg2,
                           float* __restrict__ gb2) {
    const int c = threadIdx.x;
    if (c >= 64) return;
    double db = 0.0, dg = 0.0;
    for (int b = 0; b < nblocks; ++b) { db += partial[(size_t)b * 128 + c]; dg += partial[(size_t)b * 128 + 64 + c]; }
    gb2[c] = (float)db;
    gg2[c] = (float)dg;
    cst[DB2 + c] = (float)(db / rows);
    cst[DG2 + c] = (float)(dg / rows);
}

// ---------------------------------------------------------------------------------------------------------------
// pass 3 (layers == 2): dense backward on the matrix pipe.  Block partials: dW2 [64 x 64] | sum da1 e^T [64 x 20] | sum da1 [64] |
// sum da1 xhat1 [64].  Five contractions per 64-row tile as v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation): a VALU
// form with 4x4 register blocks (the first version: 2.75 ms) reads 16 bytes of LDS per 8 FMAs and is bound by LDS bandwidth at a
// quarter of the vector peak; an MFMA step takes two 4-byte LDS reads per lane for 2 x 32 x 32 FMAs.  Four waves, one 32 x 32 output block each:
//     y1^T [c][r] = W1 E^T          A = W1t[j][c]   B = E[r][j]        (10 steps)
//     y2^T [o][r] = W2 H1^T         A = W2[o][c]    B = H1[r][c]       (32 steps)
//     dW2  [o][c] += DY2^T H1       A = DY2[r][o]   B = H1[r][c]       (32 steps, over the tile's rows)
//     dh1^T[c][r] = W2^T DY2^T      A = W2[o][c]    B = DY2[r][o]      (32 steps)
//     gE   [c][j] += DA1^T E        A = DA1[r][c]   B = E[r][j]        (16 steps per wave: the row halves go to wave pairs)
// With M = channel and N = row an accumulator gives every lane 16 channels OF ITS OWN ROW, so BatchNorm, LeakyReLU' and the
// arg-max hit test are register work and conv1's xhat / sign bits are still at hand when dh1 arrives in the same layout.
// Tiles have an odd row stride (65 / 33 floats): both a tile's rows and its columns are then conflict-free for the scalar operand
// reads.  E carries a column of ones (j = 20), so sum da1 falls out of the last contraction.  The accumulators of the three
// running sums are added into fp64 registers after every tile.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kS = 65;        // row stride of the 64-wide tiles
constexpr int kFlush = 16;     // tiles between two fp64 flushes of the running sums
constexpr int kPartM = kPart3 + 8192;   // block partial of the MFMA kernel: the compact kPart3 part + its two fp64 scratch arrays
constexpr int kSE = 33;       // row stride of the edge-feature tile (32 columns: 18 features, 2 zeros, a one, zeros)

struct TileM {
    float E[kR * kSE];
    float H1[kR * kS];
    float D2[kR * kS];
    float W1t[kES * 64];      // [j][c]
    float W2[64 * kS];        // [o][c]
    float G[4 * 64];
    uint8_t argk[4 * 64];
    int rown[kR];
};                            // 63.5 KB

__global__ void __launch_bounds__(kThreads, 2) k_eb_backward_mfma(const float* __restrict__ x9, const int32_t* __restrict__ knn, int N, int K, int P,
                                                                  int ntiles, const float* __restrict__ w1, const float* __restrict__ g1,
                                                                  const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ g2,
                                                                  const float* __restrict__ b2, const float* __restrict__ cst,
                                                                  const uint8_t* __restrict__ argk, const float* __restrict__ gout,
                                                                  double* __restrict__ partial) {
    __shared__ TileM s;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, l32 = lane & 31, half = lane >> 5;
    const int mb = wave >> 1, nb = wave & 1;                  // this wave's output block: channels 32 mb.., rows / columns 32 nb..
    for (int i = t; i < kES * 64; i += kThreads) {
        const int j = i >> 6, c = i & 63;
        s.W1t[i] = j < 18 ? w1[c * 18 + j] : 0.f;
    }
    for (int i = t; i < 64 * 64; i += kThreads) s.W2[(i >> 6) * kS + (i & 63)] = w2[i];
    // channel of accumulator register v of this lane, and the per-channel constants of the two BatchNorms for it
    auto chan = [&](int v) { return 32 * mb + (v & 3) + 8 * (v >> 2) + 4 * half; };
    // The running sums of a block (~100 tiles x 64 rows) live in the fp32 accumulators of the matrix pipe; every kFlush tiles they are
    // added into the block's fp64 partials in global memory (every lane owns its slots: plain read-modify-write) and cleared, so fp32
    // only ever sums 512 rows.  fp64 copies in registers (96 more) would spill at two workgroups per CU.
    f32x16 a3, a5;
    float dx[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) { a3[v] = 0.f; a5[v] = 0.f; dx[v] = 0.f; }
    const int cb5 = wave & 1, rh5 = wave >> 1;                // last contraction: channel block, row half
    double* dst = partial + (size_t)blockIdx.x * kPartM;
    double* acc5 = dst + kPart3;                              // [row half][channel][32]
    double* accx = acc5 + 4096;                               // [wave][register][lane]
    // this lane's slots: register v -> channel (v & 3) + 8 (v >> 2) of its block (+ 4 half), i.e. a constant offset from three bases
    double* const q3 = dst + (size_t)(32 * mb + 4 * half) * 64 + 32 * nb + l32;              // dW2[o][c = 32 nb + l32]
    double* const q5 = acc5 + ((size_t)rh5 * 64 + 32 * cb5 + 4 * half) * 32 + l32;
    double* const qx = accx + (size_t)wave * 16 * 64 + lane;
    auto flush = [&](bool init) {
        // four slots at a time (a compiler barrier between the groups): all 48 read-modify-writes in flight at once need 96 registers
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int cv = (v & 3) + 8 * (v >> 2);
            q3[cv * 64] = init ? 0.0 : q3[cv * 64] + (double)a3[v];
            a3[v] = 0.f;
            if ((v & 3) == 3) asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int cv = (v & 3) + 8 * (v >> 2);
            q5[cv * 32] = init ? 0.0 : q5[cv * 32] + (double)a5[v];
            a5[v] = 0.f;
            if ((v & 3) == 3) asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            qx[v * 64] = init ? 0.0 : qx[v * 64] + (double)dx[v];
            dx[v] = 0.f;
            if ((v & 3) == 3) asm volatile("" ::: "memory");
        }
    };
    flush(true);
    int since = 0;

    RowFetch rf;
    fetch_edge_rows(rf, x9, knn, N, K, P, blockIdx.x, ntiles);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        if (t < kR) {
            float* e = &s.E[t * kSE];
#pragma unroll
            for (int j = 0; j < 32; ++j) e[j] = 0.f;
            if (rf.n >= 0) {
#pragma unroll
                for (int c = 0; c < 9; ++c) { e[c] = rf.xj[c] - rf.xi[c]; e[9 + c] = rf.xi[c]; }
                e[9] -= cst[X0]; e[10] -= cst[X0 + 1]; e[11] -= cst[X0 + 2];
                e[20] = 1.f;
            }
            s.rown[t] = rf.n;
        }
        fetch_edge_rows(rf, x9, knn, N, K, P, tile + gridDim.x, ntiles);
        if (t < P * 64) {
            const int n = tile * P + (t >> 6);
            s.G[t] = n < N ? gout[(size_t)n * 64 + (t & 63)] : 0.f;
            s.argk[t] = n < N ? argk[(size_t)n * 64 + (t & 63)] : (uint8_t)255;
        }
        __syncthreads();
        const int r = 32 * nb + l32;                          // this lane's row in the [channel][row] products
        const bool valid = s.rown[r] >= 0;
        const int pr = r / K, kr = r - pr * K;
        // ---- conv1: y1^T[c][r] ----
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
        for (int st = 0; st < 10; ++st)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s.W1t[(2 * st + half) * 64 + 32 * mb + l32], s.E[r * kSE + 2 * st + half], acc, 0, 0, 0);
        float xh1[16];
        unsigned pos1 = 0u;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int c = chan(v);
            const float mu1 = cst[MU1 + c], inv1 = cst[INV1 + c];
            xh1[v] = (acc[v] - mu1) * inv1;
            const float a = (acc[v] - mu1) * (inv1 * g1[c]) + b1[c];
            pos1 |= (a > 0.f ? 1u : 0u) << v;
            s.H1[r * kS + c] = valid ? lrelu(a) : 0.f;
        }
        __syncthreads();
        // ---- conv2: y2^T[o][r], then dy2 for ALL rows ----
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll 4
        for (int st = 0; st < 32; ++st)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s.W2[(32 * mb + l32) * kS + 2 * st + half], s.H1[r * kS + 2 * st + half], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int o = chan(v);
            const float inv2 = cst[INV2 + o], ga2 = g2[o];
            const float xh = (acc[v] - cst[MU2 + o]) * inv2;
            float da = 0.f;
            if (valid && (int)s.argk[pr * 64 + o] == kr) da = s.G[pr * 64 + o] * dlrelu(xh * ga2 + b2[o]);
            s.D2[r * kS + o] = valid ? ga2 * inv2 * ((da - cst[DB2 + o]) - xh * cst[DG2 + o]) : 0.f;
        }
        __syncthreads();
        // ---- dW2[o][c] += sum_r dy2[r][o] h1[r][c] ----
#pragma unroll 4
        for (int st = 0; st < 32; ++st)
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(s.D2[(2 * st + half) * kS + 32 * mb + l32], s.H1[(2 * st + half) * kS + 32 * nb + l32], a3, 0, 0, 0);
        // ---- dh1^T[c][r] = sum_o W2[o][c] dy2[r][o]; da1 ----
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll 4
        for (int st = 0; st < 32; ++st)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s.W2[(2 * st + half) * kS + 32 * mb + l32], s.D2[r * kS + 2 * st + half], acc, 0, 0, 0);
        float da1[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            da1[v] = valid ? acc[v] * (((pos1 >> v) & 1u) ? 1.f : kSlope) : 0.f;
            dx[v] = fmaf(da1[v], xh1[v], dx[v]);
        }
        __syncthreads();                                      // every reader of dy2 is done
#pragma unroll
        for (int v = 0; v < 16; ++v) s.D2[r * kS + chan(v)] = da1[v];
        __syncthreads();
        // ---- gE[c][j] += sum_r da1[r][c] e[r][j]  (j = 20: the ones column -> sum da1) ----
#pragma unroll 4
        for (int st = 0; st < 16; ++st) {
            const int row = 32 * rh5 + 2 * st + half;
            a5 = __builtin_amdgcn_mfma_f32_32x32x2f32(s.D2[row * kS + 32 * cb5 + l32], s.E[row * kSE + l32], a5, 0, 0, 0);
        }
        if (++since == kFlush) { flush(false); since = 0; }
    }
    flush(false);
    __threadfence_block();
    __syncthreads();
    // ---- the compact block partials k_eb_final reads: dW2 [64 x 64] (in place) | sum da1 e^T [64 x 20] | sum da1 [64] | sum da1 xhat1 [64] ----
    for (int i = t; i < 64 * 32; i += kThreads) {
        const int c = i >> 5, j = i & 31;
        const double sum = acc5[(size_t)c * 32 + j] + acc5[(size_t)(64 + c) * 32 + j];
        if (j < 18) dst[4096 + c * kES + j] = sum;
        else if (j == 20) dst[4096 + 64 * kES + c] = sum;
    }
    if (t < 64) {
        // sum da1 xhat1 of channel c: register v of the lanes of its half, over the 32 rows of a wave and the two row blocks, fixed order
        const int c = t, m = c >> 5, v = ((c & 31) & 3) + 4 * ((c & 31) >> 3), hf = ((c & 31) >> 2) & 1;
        double sum = 0.0;
        for (int n2 = 0; n2 < 2; ++n2)
            for (int l = 0; l < 32; ++l) sum += accx[((size_t)(2 * m + n2) * 16 + v) * 64 + 32 * hf + l];
        dst[4096 + 64 * kES + 64 + c] = sum;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fold: block partials -> gradients.  `off_e` / `off_s`: where sum da1 e^T [64 x 20] and sum da1 | sum da1 xhat1 sit in a
// block's partial; dW2 (layers == 2) is the first 4096 entries.  BN1's dense part comes from the moments (file header).
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_eb_final(const double* __restrict__ partial, int nblocks, int stride, int off_e, int off_s, int layers, double rows,
                           const double* __restrict__ mom, const float* __restrict__ w1, const float* __restrict__ g1, const float* __restrict__ cst,
                           float* __restrict__ gw1, float* __restrict__ gg1, float* __restrict__ gb1, float* __restrict__ gw2) {
    __shared__ double sda[64], sdx[64];
    const int t = threadIdx.x;
    if (layers == 2)
        for (int i = t; i < 4096; i += blockDim.x) {
            double a = 0.0;
            for (int b = 0; b < nblocks; ++b) a += partial[(size_t)b * stride + i];
            gw2[i] = (float)a;
        }
    if (t < 128) {
        double a = 0.0;
        for (int b = 0; b < nblocks; ++b) a += partial[(size_t)b * stride + off_s + t];
        if (t < 64) { sda[t] = a; gb1[t] = (float)a; }
        else { sdx[t - 64] = a; gg1[t - 64] = (float)a; }
    }
    __syncthreads();
    for (int i = t; i < 64 * 18; i += blockDim.x) {
        const int c = i / 18, j = i - c * 18;
        double ge = 0.0;
        for (int b = 0; b < nblocks; ++b) ge += partial[(size_t)b * stride + off_e + c * kES + j];
        const double inv = (double)cst[INV1 + c], mu = (double)cst[MU1 + c];
        double wsee = 0.0;
        for (int q = 0; q < 18; ++q) wsee += (double)w1[c * 18 + q] * mom[q * kES + j];
        const double xe = (wsee - mu * mom[400 + j]) * inv;                       // sum_r xhat1[r][c] e[r][j]
        gw1[i] = (float)((double)g1[c] * inv * (ge - sda[c] / rows * mom[400 + j] - sdx[c] / rows * xe));
    }
}

}  // namespace

namespace sg {
// out[i] = sum over b < nblocks of partial[b * stride + i], i < count: one thread per output (coalesced across the block), eight
// independent running sums combined in a fixed order -- the fold kernels used to walk the blocks' partials one dependent load at
// a time from a single workgroup (0.25-0.8 ms each)
__global__ void k_reduce_partials(const double* __restrict__ partial, int nblocks, int stride, int count, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int b = 0;
    for (; b + 8 <= nblocks; b += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += partial[(size_t)(b + u) * stride + i];
    }
    for (int u = 0; b < nblocks; ++b, ++u) a[u] += partial[(size_t)b * stride + i];
    out[i] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
int reduce_partials(const double* d_partial, int nblocks, int stride, int count, double* d_out, hipStream_t st) {
    k_reduce_partials<<<cdiv(count, 256), 256, 0, st>>>(d_partial, nblocks, stride, count, d_out);
    SG_LAUNCH_CHECK();
    return SG_OK;
}
}  // namespace sg

namespace {

__global__ void k_eb_init(const float* __restrict__ x9, float* __restrict__ cst) {
    if (threadIdx.x < 3) cst[X0 + threadIdx.x] = x9[threadIdx.x];
}

}  // namespace

extern "C" {

static int eb_blocks(int ntiles) { return std::max(1, std::min(ntiles, 512)); }

size_t sg_edgeconv_backward_ws_bytes(int N) {
    const size_t n = (size_t)std::max(N, 1);
    return sg::align_up(512 * (size_t)kPartM * 8) + sg::align_up(n * 64 * 4) + sg::align_up(n * 64) + sg::align_up(kCst * 4) + sg::align_up(kMom * 8) + sg::align_up(kPart3 * 8) + 4096;
}

int sg_edgeconv_backward(const float* d_x9m, const int32_t* d_knn, int N, int k, int layers, const float* d_w1, const float* d_g1, const float* d_b1,
                         const float* d_w2, const float* d_g2, const float* d_b2, const float* d_gout, float* d_gw1, float* d_gg1, float* d_gb1,
                         float* d_gw2, float* d_gg2, float* d_gb2, const float* d_bn2_in, float* d_bn_stats, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(N > 0 && k > 0 && k <= 32 && (layers == 1 || layers == 2) && d_x9m && d_knn && d_w1 && d_g1 && d_b1 && d_gout && d_gw1 && d_gg1 && d_gb1 && d_ws,
               "sg_edgeconv_backward: bad arguments (k = %d must be <= 32)", k);
    SG_REQUIRE(layers == 1 || (d_w2 && d_g2 && d_b2 && d_gw2 && d_gg2 && d_gb2), "sg_edgeconv_backward: layers == 2 needs the second conv's tensors");
    sg::Carver cv(d_ws, ws_bytes);
    double* partial = cv.take<double>(512 * (size_t)kPartM);
    float* ext = cv.take<float>((size_t)N * 64);
    uint8_t* argk = cv.take<uint8_t>((size_t)N * 64);
    float* cst = cv.take<float>(kCst);
    double* mom = cv.take<double>(kMom);
    double* red = cv.take<double>(kPart3);
    if (!cv.ok) return sg::fail(SG_ENOMEM, "sg_edgeconv_backward: workspace too small (%zu < %zu)", ws_bytes, sg_edgeconv_backward_ws_bytes(N));
    hipStream_t st = sg::as_stream(stream);
    const int P = std::min(kR / k, 4), ntiles = sg::cdiv(N, P), nb = eb_blocks(ntiles);
    const double rows = (double)N * (double)k;
    k_eb_init<<<1, 64, 0, st>>>(d_x9m, cst);
    {
        const int mb = sg::moments_blocks(N);
        double* m189 = red + kMom;                               // red has kPart3 doubles
        SG_REQUIRE((size_t)mb * 189 <= 512 * (size_t)kPart3, "sg_edgeconv_backward: partial buffer");
        if (int rc = sg::edge_moments_partials(d_x9m, d_knn, N, k, partial, st)) return rc;
        if (int rc = sg::reduce_partials(partial, mb, 189, 189, m189, st)) return rc;
        k_eb_unpack_moments<<<1, 512, 0, st>>>(m189, red);
    }
    k_eb_fold1<<<1, 256, 0, st>>>(red, 1, rows, d_w1, cst, mom, d_bn_stats);
    const int nb2 = std::max(1, std::min(sg::cdiv(N, 4), 1024));
    if (layers == 1) {
        const int stride = 128 + 64 * kES;
        SG_REQUIRE((size_t)nb2 * stride <= 512 * (size_t)kPart3, "sg_edgeconv_backward: partial buffer");
        k_eb_sparse1<<<nb2, kThreads, 0, st>>>(d_x9m, d_knn, N, k, d_w1, d_g1, d_b1, cst, d_gout, partial);
        if (int rc = sg::reduce_partials(partial, nb2, stride, stride, red, st)) return rc;
        k_eb_final<<<1, 1024, 0, st>>>(red, 1, stride, 128, 0, 1, rows, mom, d_w1, d_g1, cst, d_gw1, d_gg1, d_gb1, nullptr);
    } else {
        if (d_bn2_in) {
            // the forward's BN2 statistics are at hand (the training step's tape): no dense forward pass
            k_eb_set_bn2<<<1, 64, 0, st>>>(d_bn2_in, cst, d_bn_stats);
            k_eb_sparse2<<<nb2, kThreads, 0, st>>>(d_x9m, d_knn, N, k, d_w1, d_g1, d_b1, d_w2, d_g2, d_b2, cst, d_gout, argk, partial);
        } else {
            k_eb_forward<<<nb, kThreads, 0, st>>>(d_x9m, d_knn, N, k, P, ntiles, d_w1, d_g1, d_b1, d_w2, d_g2, cst, ext, argk, partial);
            if (int rc = sg::reduce_partials(partial, nb, 128, 128, red, st)) return rc;
            k_eb_fold2<<<1, 64, 0, st>>>(red, 1, rows, cst, d_bn_stats);
            k_eb_last_bn<<<nb2, kThreads, 0, st>>>(N, d_g2, d_b2, cst, ext, d_gout, partial);
        }
        if (int rc = sg::reduce_partials(partial, nb2, 128, 128, red, st)) return rc;
        k_eb_fold3<<<1, 64, 0, st>>>(red, 1, rows, cst, d_gg2, d_gb2);
        k_eb_backward_mfma<<<nb, kThreads, 0, st>>>(d_x9m, d_knn, N, k, P, ntiles, d_w1, d_g1, d_b1, d_w2, d_g2, d_b2, cst, argk, d_gout, partial);
        if (int rc = sg::reduce_partials(partial, nb, kPartM, kPart3, red, st)) return rc;
        k_eb_final<<<1, 1024, 0, st>>>(red, 1, kPart3, 4096, 4096 + 64 * kES, 2, rows, mom, d_w1, d_g1, cst, d_gw1, d_gg1, d_gb1, d_gw2);
    }
    SG_LAUNCH_CHECK();
    return SG_OK;
}

}  // extern "C"
