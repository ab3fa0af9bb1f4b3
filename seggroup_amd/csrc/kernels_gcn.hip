// a14: calculate_similarity + build_similarity_matrix + GCN (reference seggroup/model.py:262-265,
// 305-309, 141-151):   X' = relu( ((I + sym(exp(-alpha * d))) row-normalised) @ X @ W^T )
//
// The reference materialises the dense [S,S] matrix; the cluster graph has ~6 edges per row, so the
// row-normalised product is evaluated over the symmetric CSR built by the host grouping engine.
// S <= a few thousand and D <= 256: this stage is latency-, not bandwidth-bound; sums are carried in
// fp64 so the decision distances that follow are as close to exact arithmetic as fp32 features allow.
#include <cstdlib>

#include "engine_ctx.h"
#include "sg_common.h"
#include "wave_ops.h"

namespace {

__global__ void k_transpose(const float* __restrict__ w, int D, float* __restrict__ wt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D * D) wt[(i % D) * D + (i / D)] = w[i];      // wt[k][o] = w[o][k]
}

// one block per row: agg[i] = (x_i + sum_j s_ij x_j) / (1 + sum_j s_ij),  s_ij = exp(-alpha * dist_e).
// The distances of the row's edges are computed here (fp64 accumulation as in k_edge_distance, orientation (a - b + 1e-6)
// from the edge list, 16 lanes per edge): an edge is evaluated by both of its rows, by the same code and therefore to the
// same bits, which is cheaper than a launch of its own in front of this kernel.
// The row's edge list (edge id, neighbour, the edge's two endpoints) is staged in LDS with two rounds of parallel loads and
// every edge's weight is evaluated once -- walking rowptr -> eid -> adj -> rows -> dist per edge and per thread was a chain of
// dependent global round trips (72 us per launch for ~1,000 rows).  Rows with more than kDegCap edges take the direct path.
constexpr int kDegCap = 256;
__device__ __forceinline__ void gcn_aggregate_body(const float* __restrict__ x, int D, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const int32_t* __restrict__ eid, const int32_t* __restrict__ adj, float* __restrict__ dist, float alpha,
                                                   float* __restrict__ agg, int i) {
    __shared__ int s_id[kDegCap], s_col[kDegCap], s_a[kDegCap], s_b[kDegCap];
    __shared__ double s_w[kDegCap];
    const int lo = rowptr[i], hi = rowptr[i + 1], deg = hi - lo;
    const bool staged = deg <= kDegCap;
    if (staged) {
        for (int t = threadIdx.x; t < deg; t += blockDim.x) {
            const int id = eid[lo + t];
            s_id[t] = id;
            s_col[t] = col[lo + t];
            s_a[t] = adj[2 * id];
            s_b[t] = adj[2 * id + 1];
        }
        __syncthreads();
    }
    // 16 lanes per edge: a hub cluster has 100+ edges and one edge per wave made it the kernel's tail
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4, ng = blockDim.x >> 4;
    for (int t0 = 0; t0 < deg; t0 += ng) {
        const int t = t0 + grp;
        double acc = 0.0;
        int id = 0;
        if (t < deg) {
            id = staged ? s_id[t] : eid[lo + t];
            const float* a = x + (size_t)(staged ? s_a[t] : adj[2 * id]) * D;
            const float* b = x + (size_t)(staged ? s_b[t] : adj[2 * id + 1]) * D;
            int k = sub;
            for (; k + 48 < D; k += 64) {                         // four strides per trip, their eight loads in flight together
                const float a0 = a[k], a1 = a[k + 16], a2 = a[k + 32], a3 = a[k + 48], b0 = b[k], b1 = b[k + 16], b2 = b[k + 32], b3 = b[k + 48];
                double d = (double)a0 - (double)b0 + 1e-6; acc = fma(d, d, acc);
                d = (double)a1 - (double)b1 + 1e-6; acc = fma(d, d, acc);
                d = (double)a2 - (double)b2 + 1e-6; acc = fma(d, d, acc);
                d = (double)a3 - (double)b3 + 1e-6; acc = fma(d, d, acc);
            }
            for (; k < D; k += 16) {
                const double d = (double)a[k] - (double)b[k] + 1e-6;
                acc = fma(d, d, acc);
            }
        }
        // 16 lanes per edge = one DPP row: quad, quad pairs, then two row rotations leave the row's sum in every lane (a
        // `__shfl_xor` step is an LDS round trip, twice for a double)
        acc += sgw::dpp_d<sgw::kQuadXor1>(acc, acc);
        acc += sgw::dpp_d<sgw::kQuadXor2>(acc, acc);
        acc += sgw::dpp_d<sgw::kRowRor4>(acc, acc);
        acc += sgw::dpp_d<sgw::kRowRor8>(acc, acc);
        if (t < deg && sub == 0) {
            const float df = (float)sqrt(acc);
            dist[id] = df;
            if (staged) s_w[t] = exp(-(double)df * (double)alpha);
        }
    }
    __syncthreads();                                        // weights in LDS / this block's own stores to dist[] visible to it
    double rowsum = 1.0;
    if (staged) {
        for (int t = 0; t < deg; ++t) rowsum += s_w[t];
        for (int k = threadIdx.x; k < D; k += blockDim.x) {
            double acc = (double)x[(size_t)i * D + k];
            for (int t = 0; t < deg; ++t) acc = fma(s_w[t], (double)x[(size_t)s_col[t] * D + k], acc);
            agg[(size_t)i * D + k] = (float)(acc / rowsum);
        }
    } else {
        for (int e = lo; e < hi; ++e) rowsum += exp(-(double)dist[eid[e]] * (double)alpha);
        for (int k = threadIdx.x; k < D; k += blockDim.x) {
            double acc = (double)x[(size_t)i * D + k];
            for (int e = lo; e < hi; ++e)
                acc = fma(exp(-(double)dist[eid[e]] * (double)alpha), (double)x[(size_t)col[e] * D + k], acc);
            agg[(size_t)i * D + k] = (float)(acc / rowsum);
        }
    }
}
__global__ void k_gcn_aggregate(const float* __restrict__ x, int D, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                const int32_t* __restrict__ eid, const int32_t* __restrict__ adj, float* __restrict__ dist, float alpha,
                                float* __restrict__ agg) {
    gcn_aggregate_body(x, D, rowptr, col, eid, adj, dist, alpha, agg, blockIdx.x);
}
__global__ void k_gcn_aggregate_b(const sg::SlotCtx* __restrict__ cx, float alpha) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x >= c.C) return;
    gcn_aggregate_body(c.cat, c.Dcat, c.rowptr, c.col, c.eid, c.g_adj, c.g_dist, alpha, c.g_agg, blockIdx.x);
}

constexpr int kRows = 8;
// out[i][o] = relu(sum_k agg[i][k] * W[o][k]); block = kRows rows, thread o = output column
// out2 (may be null): a second copy of the result (the engine's outbox)
__device__ __forceinline__ void gcn_fc_body(const float* __restrict__ agg, int S, int D, const float* __restrict__ wt, float* __restrict__ out,
                                            float* __restrict__ out2, int bid) {
    // the block's rows once as DOUBLES in LDS (every thread used to convert the same row values again, per k and per row, behind a
    // 4-byte broadcast read each: 8 LDS reads + 8 conversions per 8 FMAs); two k per 16-byte broadcast read
    __shared__ __attribute__((aligned(16))) double rows[kRows][256];
    const int r0 = bid * kRows;
    for (int i = threadIdx.x; i < kRows * D; i += blockDim.x) {
        const int rr = i / D, k = i % D;
        rows[rr][k] = (r0 + rr < S) ? (double)agg[(size_t)(r0 + rr) * D + k] : 0.0;
    }
    __syncthreads();
    const int o = threadIdx.x;
    if (o >= D) return;
    double acc[kRows];
#pragma unroll
    for (int rr = 0; rr < kRows; ++rr) acc[rr] = 0.0;
    // the k order of the sum is unchanged (D is 192 or 256 on the hot path; an odd D takes the last k alone)
    // two pairs per trip: the four weight loads and the sixteen row reads of a trip are issued together (left to itself the compiler
    // waited out every load on its own; one scene = ~130 workgroups has nothing else to hide them with)
    int k = 0;
    for (; k + 3 < D; k += 4) {
        const float f0 = wt[(size_t)k * D + o], f1 = wt[(size_t)(k + 1) * D + o], f2 = wt[(size_t)(k + 2) * D + o], f3 = wt[(size_t)(k + 3) * D + o];
        double2 xa[kRows], xb[kRows];
#pragma unroll
        for (int rr = 0; rr < kRows; ++rr) { xa[rr] = *reinterpret_cast<const double2*>(&rows[rr][k]); xb[rr] = *reinterpret_cast<const double2*>(&rows[rr][k + 2]); }
        const double w0 = (double)f0, w1 = (double)f1, w2 = (double)f2, w3 = (double)f3;
#pragma unroll
        for (int rr = 0; rr < kRows; ++rr) {
            acc[rr] = fma(xa[rr].x, w0, acc[rr]);
            acc[rr] = fma(xa[rr].y, w1, acc[rr]);
            acc[rr] = fma(xb[rr].x, w2, acc[rr]);
            acc[rr] = fma(xb[rr].y, w3, acc[rr]);
        }
    }
    for (; k + 1 < D; k += 2) {
        const double w0 = (double)wt[(size_t)k * D + o], w1 = (double)wt[(size_t)(k + 1) * D + o];
#pragma unroll
        for (int rr = 0; rr < kRows; ++rr) {
            const double2 x = *reinterpret_cast<const double2*>(&rows[rr][k]);
            acc[rr] = fma(x.x, w0, acc[rr]);
            acc[rr] = fma(x.y, w1, acc[rr]);
        }
    }
    if (D & 1) {
        const double w = (double)wt[(size_t)(D - 1) * D + o];
#pragma unroll
        for (int rr = 0; rr < kRows; ++rr) acc[rr] = fma(rows[rr][D - 1], w, acc[rr]);
    }
#pragma unroll
    for (int rr = 0; rr < kRows; ++rr)
        if (r0 + rr < S) {
            const float v = fmaxf((float)acc[rr], 0.f);
            out[(size_t)(r0 + rr) * D + o] = v;
            if (out2) out2[(size_t)(r0 + rr) * D + o] = v;
        }
}
__global__ void k_gcn_fc(const float* __restrict__ agg, int S, int D, const float* __restrict__ wt, float* __restrict__ out) {
    gcn_fc_body(agg, S, D, wt, out, nullptr, blockIdx.x);
}
__global__ void k_gcn_fc_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x * kRows >= c.C) return;
    gcn_fc_body(c.g_agg, c.C, c.Dcat, c.g_wt, c.g_out, c.g_out_copy, blockIdx.x);
}


// Round 5: the fc on the fp64 MATRIX pipe (v_mfma_f64_16x16x4_f64; layout probed by tools/micro/mfma_f64_layout.hip: A lane l = [l % 16][l / 16],
// B lane l = [l / 16][l % 16], D lane l, register r = [l / 16 + 4 r][l % 16]).  k_gcn_fc (thread = one output column, the block's 8 rows
// broadcast out of LDS) pulls 1 KB through the LDS return path per wave and (row, k pair): ~43 / 36 us per launch of 8 scenes for the two
// layers against ~10 us of fp64 work.  Here a block owns 16 rows, its four waves a quarter of the D output columns each (3 | 4 tiles of
// 16): per four k one 8-byte LDS read of the rows (as doubles, row stride D + 2: conflict-free), one coalesced weight load per column tile
// (W^T, float -> double), one MFMA per tile; the next four k's operands are requested before this step's MFMAs.  fp64 products and sums
// like the VALU fc, each output's k in ascending order inside the pipe; the results go through the same (float) cast and ReLU.
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int D>
__device__ __forceinline__ void gcn_fc_mfma_body(const float* __restrict__ agg, int S, const float* __restrict__ wt, float* __restrict__ out,
                                                 float* __restrict__ out2, int bid) {
    constexpr int kT = D / 64;                                    // column tiles per wave
    static_assert(D % 64 == 0, "four waves x kT tiles of 16 columns");
    __shared__ double rows[16][D + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = bid * 16;
    for (int i = tid; i < 16 * D; i += 256) {
        const int rr = i / D, k = i - rr * D;
        rows[rr][k] = (r0 + rr < S) ? (double)agg[(size_t)(r0 + rr) * D + k] : 0.0;
    }
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;                     // A: row li, k offset lk;  B: k offset lk, column li
    const int c0 = wave * (kT * 16);
    f64x4 acc[kT];
#pragma unroll
    for (int t = 0; t < kT; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
    const float* wp = wt + (size_t)lk * D + c0 + li;             // + 4 b D per k block, + 16 t per tile
    // the weights of EIGHT k blocks in flight (a ring of 8 x kT registers, statically indexed: the loop is unrolled by eight): a layer of 134
    // clusters is 9 blocks per scene, one wave per SIMD, and with one block of lookahead every MFMA group waited out an L2 round trip
    constexpr int kPF = 8, kB = D / 4;
    static_assert(kB % kPF == 0, "k blocks in groups of eight");
    float bq[kPF][kT];
#pragma unroll
    for (int u = 0; u < kPF; ++u)
#pragma unroll
        for (int t = 0; t < kT; ++t) bq[u][t] = wp[(size_t)u * 4 * D + 16 * t];
    for (int b0 = 0; b0 < kB; b0 += kPF) {
#pragma unroll
        for (int u = 0; u < kPF; ++u) {
            const int b = b0 + u;
            const double a = rows[li][4 * b + lk];
            double bd_[kT];
#pragma unroll
            for (int t = 0; t < kT; ++t) bd_[t] = (double)bq[u][t];
            if (b + kPF < kB) {
#pragma unroll
                for (int t = 0; t < kT; ++t) bq[u][t] = wp[(size_t)(b + kPF) * 4 * D + 16 * t];
            }
#pragma unroll
            for (int t = 0; t < kT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bd_[t], acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + lk + 4 * r;
        if (row < S)
#pragma unroll
            for (int t = 0; t < kT; ++t) {
                const float v = fmaxf((float)acc[t][r], 0.f);
                out[(size_t)row * D + c0 + 16 * t + li] = v;
                if (out2) out2[(size_t)row * D + c0 + 16 * t + li] = v;
            }
    }
}
template <int D>
__global__ __launch_bounds__(256) void k_gcn_fc_mfma(const float* __restrict__ agg, int S, const float* __restrict__ wt, float* __restrict__ out) {
    gcn_fc_mfma_body<D>(agg, S, wt, out, nullptr, blockIdx.x);
}
template <int D>
__global__ __launch_bounds__(256) void k_gcn_fc_mfma_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x * 16 >= c.C) return;
    gcn_fc_mfma_body<D>(c.g_agg, c.C, c.g_wt, c.g_out, c.g_out_copy, blockIdx.x);
}

}  // namespace

namespace sg {

// d_wt = W^T ([k][o]); workspace as sg_gcn_forward (its transpose slot stays unused)
int gcn_forward_wt(const float* d_x, int S, int D, const int32_t* d_adj, int E, const int32_t* d_rowptr, const int32_t* d_col,
                   const int32_t* d_eid, const float* d_wt, float alpha, float* d_out, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(S >= 0 && D > 0 && D <= 256 && E >= 0 && d_ws, "sg_gcn_forward: bad arguments (D=%d must be <= 256)", D);
    if (S == 0) return SG_OK;
    sg::Carver cv(d_ws, ws_bytes);
    float* dist = cv.take<float>(std::max(E, 1));
    float* agg = cv.take<float>((size_t)S * D);
    if (!cv.ok) return sg::fail(SG_ENOMEM, "sg_gcn_forward: workspace too small (%zu < %zu)", ws_bytes, sg_gcn_ws_bytes(S, D, E));
    hipStream_t st = sg::as_stream(stream);
    static const bool valu_fc = getenv("SG_GCN_VALU_FC") != nullptr;      // the thread = column fc of round 4, kept as the cross-check
    k_gcn_aggregate<<<S, 64 * sg::cdiv(D, 64), 0, st>>>(d_x, D, d_rowptr, d_col, d_eid, d_adj, dist, alpha, agg);
    if (D == 192 && !valu_fc) k_gcn_fc_mfma<192><<<sg::cdiv(S, 16), 256, 0, st>>>(agg, S, d_wt, d_out);
    else if (D == 256 && !valu_fc) k_gcn_fc_mfma<256><<<sg::cdiv(S, 16), 256, 0, st>>>(agg, S, d_wt, d_out);
    else k_gcn_fc<<<sg::cdiv(S, kRows), 256, 0, st>>>(agg, S, D, d_wt, d_out);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

// every slot of a launch is at the same layer, hence the same feature width (192 | 256); the aggregate kernel writes the
// edge distances of g_adj into g_dist as a by-product (sg::gcn_forward_wt)
int b_gcn(const SlotCtx* d_ctx, const BatchDims& bd, float alpha, hipStream_t st) {
    if (bd.nslots == 0 || bd.max_C == 0) return SG_OK;
    static const bool valu_fc = getenv("SG_GCN_VALU_FC") != nullptr;
    k_gcn_aggregate_b<<<dim3(bd.max_C, bd.nslots), 256, 0, st>>>(d_ctx, alpha);
    const dim3 g16(sg::cdiv(bd.max_C, 16), bd.nslots);
    if (bd.gcn_D == 192 && !valu_fc) k_gcn_fc_mfma_b<192><<<g16, 256, 0, st>>>(d_ctx);
    else if (bd.gcn_D == 256 && !valu_fc) k_gcn_fc_mfma_b<256><<<g16, 256, 0, st>>>(d_ctx);
    else k_gcn_fc_b<<<dim3(sg::cdiv(bd.max_C, kRows), bd.nslots), 256, 0, st>>>(d_ctx);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

}  // namespace sg

extern "C" {

size_t sg_gcn_ws_bytes(int S, int D, int E) {
    return sg::align_up((size_t)std::max(E, 1) * 4) + sg::align_up((size_t)std::max(S, 1) * D * 4) + sg::align_up((size_t)D * D * 4);
}

int sg_gcn_forward(const float* d_x, int S, int D, const int32_t* d_adj, int E, const int32_t* d_rowptr, const int32_t* d_col,
                   const int32_t* d_eid, const float* d_w, float alpha, float* d_out, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(S >= 0 && D > 0 && D <= 256 && E >= 0 && d_ws, "sg_gcn_forward: bad arguments (D=%d must be <= 256)", D);
    if (S == 0) return SG_OK;
    sg::Carver cv(d_ws, ws_bytes);
    (void)cv.take<float>(std::max(E, 1));
    (void)cv.take<float>((size_t)S * D);
    float* wt = cv.take<float>((size_t)D * D);
    if (!cv.ok) return sg::fail(SG_ENOMEM, "sg_gcn_forward: workspace too small (%zu < %zu)", ws_bytes, sg_gcn_ws_bytes(S, D, E));
    k_transpose<<<sg::cdiv(D * D, 256), 256, 0, sg::as_stream(stream)>>>(d_w, D, wt);
    return sg::gcn_forward_wt(d_x, S, D, d_adj, E, d_rowptr, d_col, d_eid, wt, alpha, d_out, d_ws, ws_bytes, stream);
}

}  // extern "C"
