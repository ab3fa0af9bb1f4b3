// a4+a5: get_cluster_pointcloud + farthest_point_sampling (reference seggroup/model.py:319-426).
//
// One workgroup per cluster.  The cluster's XYZ and the running min-distance array live in LDS (up
// to kLdsCap points; larger clusters spill to an L2-resident global scratch).  Each of the P%n
// sampling steps is one strided pass (distance to the newest pick, min with the running array,
// first-index argmax) followed by a wave-shuffle + LDS reduction.  Small clusters run on ONE wave
// (no workgroup barriers at all); large ones on 16 waves.
//
// Bit-exactness: squared distances are evaluated exactly like NumPy does for
// ((a - b) ** 2).sum(axis=2) on float32 (model.py:326): three individually rounded squares added as
// (dx2 + dy2) + dz2 -- this translation unit is compiled with -ffp-contract=off so no FMA is formed.
#include <climits>
#include <type_traits>

#include "engine_ctx.h"
#include "sg_common.h"
#include "wave_ops.h"

namespace {

constexpr int kLdsCap = 8192;        // points whose xyz+min-distance fit the LDS carve (16 B each)
constexpr int kSmallMax = 2048;      // clusters up to this size run on a single wave

struct Best {
    float v;
    int i;
};

__device__ inline Best better(Best a, Best b) {   // larger value wins, ties -> lower index (np.argmax)
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

template <int BLOCK>
__device__ inline Best block_argmax(Best b, Best* red) {
    sgw::wave_argmax(b.v, b.i);                                  // DPP network + readlane (same total order as better())
    if constexpr (BLOCK > 64) {
        const int wid = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) red[wid] = b;
        __syncthreads();
        Best r = red[0];
#pragma unroll
        for (int w = 1; w < BLOCK / 64; ++w) r = better(r, red[w]);
        __syncthreads();
        return r;
    }
    return b;
}

template <int BLOCK, bool RESET>
__device__ inline Best fps_pass(const float* X, const float* Y, const float* Z, float* M, int n, float qx, float qy, float qz,
                                Best* red) {
    Best b{-INFINITY, INT_MAX};
    for (int i = threadIdx.x; i < n; i += BLOCK) {
        const float dx = X[i] - qx, dy = Y[i] - qy, dz = Z[i] - qz;
        float d = (dx * dx + dy * dy) + dz * dz;
        if (!RESET) d = fminf(M[i], d);
        M[i] = d;
        if (d > b.v) { b.v = d; b.i = i; }
    }
    return block_argmax<BLOCK>(b, red);
}

template <int BLOCK>
__device__ __forceinline__ void fps_sample_body(const float* __restrict__ data, int N, int ch_in,
                                                const int32_t* __restrict__ members, const int32_t* __restrict__ cl_off,
                                                int P, int ch_out, int transform, int n_lo, int n_hi, int lds_pts,
                                                float* __restrict__ samples, int32_t* __restrict__ sel,
                                                float* __restrict__ ws, int c, const int32_t* __restrict__ sperm = nullptr,
                                                const float* __restrict__ chunk_box = nullptr, const int32_t* __restrict__ seg_chunk_off = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lo = cl_off[c], n = cl_off[c + 1] - lo;
    if (n < n_lo || n > n_hi || n <= 0) return;
    int* picks = reinterpret_cast<int*>(smem);                       // [P]
    Best* red = reinterpret_cast<Best*>(picks + P);                  // [16]
    float* lds_f = reinterpret_cast<float*>(red + 16);               // [4 * lds_pts]
    const int rep = P / n, rem = P % n;
    const int tid = threadIdx.x;

    bool sampled = false;
    if constexpr (BLOCK == 64) {
        // Segments of <= 256 points (almost all of them): coordinates and running minima live in REGISTERS, four points per
        // lane (point i = lane + 64 u, the same assignment as the strided loop below, so the first-index argmax is unchanged);
        // the newest pick's coordinates come from its owner lane through v_readlane.  The 64 dependent steps of a segment
        // were bound by LDS round trips (4 reads + 1 write per point and step, then 3 broadcast reads).
        // kU = points per lane, picked by the segment's size (wave-uniform): a 100-point segment steps over two slots, not four
        auto in_registers = [&](auto ku) {
            constexpr int kU = decltype(ku)::value;
            float X[kU], Y[kU], Z[kU], M[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int i = tid + 64 * u;
                X[u] = Y[u] = Z[u] = 0.f; M[u] = 0.f;
                if (i < n) {
                    const float* row = data + (size_t)members[lo + i] * ch_in;
                    X[u] = row[0]; Y[u] = row[1]; Z[u] = row[2];
                }
            }
            // branch-free on purpose: written with `if (i < n)` / `if (d > b.v)` the step compiled to 35 branches and 45 exec-mask
            // saves for ~60 VALU of arithmetic, and a segment is a chain of 64 such steps.  A slot past the segment's end carries
            // the running minimum -inf: it never wins the argmax (slot 0 always holds point 0, whose distance is >= 0).
            bool in[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) in[u] = tid + 64 * u < n;
            auto pass = [&](float qx, float qy, float qz, bool reset) {
                Best b{-INFINITY, INT_MAX};
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int i = tid + 64 * u;
                    const float dx = X[u] - qx, dy = Y[u] - qy, dz = Z[u] - qz;
                    float d = (dx * dx + dy * dy) + dz * dz;
                    d = reset ? d : fminf(M[u], d);
                    d = in[u] ? d : -INFINITY;
                    M[u] = d;
                    const bool better_ = d > b.v;                 // strict: the lower index (u ascending) keeps a tie
                    b.v = better_ ? d : b.v;
                    b.i = better_ ? i : b.i;
                }
                sgw::wave_argmax(b.v, b.i);
                return b;
            };
            auto coords = [&](int cur, float& qx, float& qy, float& qz) {     // cur is wave-uniform
                const int u = cur >> 6, l = cur & 63;
                float sx = X[0], sy = Y[0], sz = Z[0];
#pragma unroll
                for (int w = 1; w < kU; ++w) { sx = u == w ? X[w] : sx; sy = u == w ? Y[w] : sy; sz = u == w ? Z[w] : sz; }
                qx = sgw::bcast(sx, l); qy = sgw::bcast(sy, l); qz = sgw::bcast(sz, l);
            };
            float qx, qy, qz;
            coords(0, qx, qy, qz);
            Best b = pass(qx, qy, qz, true);                          // start at member 0 (model.py:382-386)
            int cur = __builtin_amdgcn_readfirstlane(b.i);      // wave-uniform: say so, or `coords` becomes a branch tree
            if (tid == 0) picks[0] = cur;
            coords(cur, qx, qy, qz);
            b = pass(qx, qy, qz, true);
            for (int it = 1; it < rem; ++it) {                       // model.py:389-394
                cur = __builtin_amdgcn_readfirstlane(b.i);
                if (tid == 0) picks[it] = cur;
                if (it + 1 < rem) {
                    coords(cur, qx, qy, qz);
                    b = pass(qx, qy, qz, false);
                }
            }
        };
        if (rem > 0 && n <= 256) {
            if (n <= 64) in_registers(std::integral_constant<int, 1>{});
            else if (n <= 128) in_registers(std::integral_constant<int, 2>{});
            else in_registers(std::integral_constant<int, 4>{});
            sampled = true;
        }
    }
    if constexpr (BLOCK == 1024) {
        // Segments that do not fit the LDS carve (ScanNet floors and walls: 10k-40k points) with the Morton order and the 32-point
        // chunk boxes of k_bigseg_sort_boxes at hand: a sampling step only has to touch the chunks the new pick can still
        // improve.  Per chunk the running maximum of its points' min-distances is kept (cm, with the member index ci that
        // np.argmax would return for it and its sorted position cr); a chunk whose box is farther from the pick than cm keeps
        // every minimum (d >= box distance >= cm >= M_i: the 1e-6 margin covers the fp32 rounding of both sides), and the step's
        // argmax is the argmax over the chunk maxima (ties -> lowest member index, as in the plain pass).  The first two passes
        // (start point, then the reset to the first pick: model.py:382-386) visit everything.  One block walking 40k points 64 times
        // through L2 was 0.55 ms per launch.
        if (rem > 0 && !sampled && n > lds_pts && sperm && (n + 31) / 32 <= lds_pts) {
            const int nch = (n + 31) / 32, c0 = seg_chunk_off[c];
            float* cm = lds_f;                                                     // the carve is unused on this path: 4 x lds_pts words
            int* ci = reinterpret_cast<int*>(lds_f + lds_pts);
            int* cr = reinterpret_cast<int*>(lds_f + 2 * (size_t)lds_pts);
            int* wl = reinterpret_cast<int*>(lds_f + 3 * (size_t)lds_pts);
            __shared__ int wl_n;
            __shared__ float rv[16];
            __shared__ int ri[16], rr[16];
            float* Xs = ws + lo; float* Ys = ws + (size_t)N + lo; float* Zs = ws + 2 * (size_t)N + lo; float* Ms = ws + 3 * (size_t)N + lo;
            for (int r = tid; r < n; r += BLOCK) {
                const float* row = data + (size_t)members[sperm[lo + r]] * ch_in;
                Xs[r] = row[0]; Ys[r] = row[1]; Zs[r] = row[2];
            }
            __syncthreads();
            const int lane = tid & 63, wave = tid >> 6, hl = lane & 31, half = lane >> 5;
            // one chunk per half wave: distances to q, running minima, the chunk's (max, member index, sorted position)
            auto visit = [&](int ch, float qx, float qy, float qz, bool reset) {
                const int r = 32 * ch + hl;
                const bool in = r < n;
                float d = -INFINITY;
                int li = INT_MAX;
                if (in) {
                    const float dx = Xs[r] - qx, dy = Ys[r] - qy, dz = Zs[r] - qz;
                    d = (dx * dx + dy * dy) + dz * dz;
                    if (!reset) d = fminf(Ms[r], d);
                    Ms[r] = d;
                    li = sperm[lo + r] - lo;
                }
                // maxima / minima of the 32 lanes of each half: rows of 16 on the DPP path, row 0 -> 1 and 2 -> 3, then lanes 31 / 63
                float m = d;
                m = fmaxf(m, sgw::dpp_f<sgw::kQuadXor1>(m, m)); m = fmaxf(m, sgw::dpp_f<sgw::kQuadXor2>(m, m));
                m = fmaxf(m, sgw::dpp_f<sgw::kRowRor4>(m, m));  m = fmaxf(m, sgw::dpp_f<sgw::kRowRor8>(m, m));
                m = fmaxf(m, sgw::dpp_f<sgw::kRowBcast15, 0xA>(m, m));
                const float mh = half ? sgw::bcast(m, 63) : sgw::bcast(m, 31);
                int k = d == mh ? li : INT_MAX;
                k = min(k, sgw::dpp_i<sgw::kQuadXor1>(k, k)); k = min(k, sgw::dpp_i<sgw::kQuadXor2>(k, k));
                k = min(k, sgw::dpp_i<sgw::kRowRor4>(k, k));  k = min(k, sgw::dpp_i<sgw::kRowRor8>(k, k));
                k = min(k, sgw::dpp_i<sgw::kRowBcast15, 0xA>(k, k));
                const int kh = half ? sgw::bcast(k, 63) : sgw::bcast(k, 31);
                if (in && li == kh) { cm[ch] = mh; ci[ch] = kh; cr[ch] = r; }     // exactly one lane of the half
            };
            // argmax over the chunk maxima: larger value, ties -> lower member index; every thread returns the winner
            auto pick = [&](float& bv, int& bi, int& br) {
                float v = -INFINITY; int i = INT_MAX, r = 0;
                for (int ch = tid; ch < nch; ch += BLOCK) {
                    const float cv = cm[ch]; const int cix = ci[ch];
                    if (cv > v || (cv == v && cix < i)) { v = cv; i = cix; r = cr[ch]; }
                }
                const float wm = sgw::wave_max(v);
                const int wi = sgw::wave_min(v == wm ? i : INT_MAX);
                const int wr = sgw::wave_min(v == wm && i == wi ? r : INT_MAX);
                if (lane == 0) { rv[wave] = wm; ri[wave] = wi; rr[wave] = wr; }
                __syncthreads();
                bv = rv[0]; bi = ri[0]; br = rr[0];
#pragma unroll
                for (int w = 1; w < 16; ++w)
                    if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; br = rr[w]; }
                __syncthreads();
            };
            auto full_pass = [&](float qx, float qy, float qz) {
                for (int e = 2 * wave; e < nch; e += 32) visit(min(e + half, nch - 1), qx, qy, qz, true);   // both halves together (see below)
                __syncthreads();
            };
            float bv; int bi, br;
            {
                const float* row0 = data + (size_t)members[lo] * ch_in;           // start at member 0 (model.py:382-386)
                full_pass(row0[0], row0[1], row0[2]);
                pick(bv, bi, br);
            }
            int cur = bi;
            if (tid == 0) picks[0] = cur;
            full_pass(Xs[br], Ys[br], Zs[br]);                                    // reset to the first pick
            pick(bv, bi, br);
            for (int it = 1; it < rem; ++it) {                                    // model.py:389-394
                cur = bi;
                if (tid == 0) { picks[it] = cur; wl_n = 0; }
                if (it + 1 < rem) {
                    const float qx = Xs[br], qy = Ys[br], qz = Zs[br];
                    __syncthreads();
                    for (int ch = tid; ch < nch; ch += BLOCK) {
                        const float* bx = chunk_box + (size_t)(c0 + ch) * 8;
                        const float gx = fmaxf(fmaxf(bx[0] - qx, qx - bx[3]), 0.f), gy = fmaxf(fmaxf(bx[1] - qy, qy - bx[4]), 0.f),
                                    gz = fmaxf(fmaxf(bx[2] - qz, qz - bx[5]), 0.f);
                        if (((gx * gx + gy * gy) + gz * gz) * 0.999999f < cm[ch]) wl[atomicAdd(&wl_n, 1)] = ch;
                    }
                    __syncthreads();
                    const int nw = wl_n;
                    // both halves of a wave must run visit() together (its reductions are wave-wide DPP networks): an odd tail
                    // revisits the last listed chunk, which changes nothing
                    for (int e = 2 * wave; e < nw; e += 32) visit(wl[min(e + half, nw - 1)], qx, qy, qz, false);
                    __syncthreads();
                    pick(bv, bi, br);
                }
            }
            sampled = true;
        }
    }
    if (rem > 0 && !sampled) {
        float *X, *Y, *Z, *M;
        if (n <= lds_pts) { X = lds_f; Y = X + lds_pts; Z = Y + lds_pts; M = Z + lds_pts; }
        else { X = ws + lo; Y = ws + (size_t)N + lo; Z = ws + 2 * (size_t)N + lo; M = ws + 3 * (size_t)N + lo; }
        for (int i = tid; i < n; i += BLOCK) {
            const float* row = data + (size_t)members[lo + i] * ch_in;
            X[i] = row[0]; Y[i] = row[1]; Z[i] = row[2];
        }
        __syncthreads();
        // start at member 0; first pick = farthest from it, min-distance array RESET to that pick (model.py:382-386)
        Best b = fps_pass<BLOCK, true>(X, Y, Z, M, n, X[0], Y[0], Z[0], red);
        int cur = b.i;
        if (tid == 0) picks[0] = cur;
        __syncthreads();
        b = fps_pass<BLOCK, true>(X, Y, Z, M, n, X[cur], Y[cur], Z[cur], red);
        for (int it = 1; it < rem; ++it) {                           // model.py:389-394
            cur = b.i;
            if (tid == 0) picks[it] = cur;
            if (it + 1 < rem) {
                __syncthreads();
                b = fps_pass<BLOCK, false>(X, Y, Z, M, n, X[cur], Y[cur], Z[cur], red);
            }
        }
    }
    if (rem > 0) {
        __syncthreads();
        if (tid == 0 && picks[rem - 1] == 0) {                       // trailing-zero fix-up (model.py:407-412)
            int j = 1;
            while (j <= rem && picks[rem - j] == 0) ++j;
            if (j > rem) j = rem;
            const int invalid = j - 1;
            // NumPy copies overlapping slices through a temporary: walk downwards (dst index > src index)
            for (int t = invalid - 1; t >= 0; --t) picks[rem - invalid + t] = picks[t];
        }
        __syncthreads();
    }

    // rows: members tiled rep times, then the picks (model.py:413-420)
    float* out = samples + (size_t)c * P * ch_out;
    if constexpr (BLOCK == 64) {
        // One wave, P = 64 rows, six channels (the structural layer): lane r owns row r, so the transform below runs on registers.
        // The general path writes the rows, reads them back for the mean (12 threads, 16 DEPENDENT global loads each: as long as
        // the 64 sampling steps themselves), rewrites them centred, rereads and rewrites them scaled.  The mean's association
        // order -- four interleaved accumulators over the rows, added in order: see below -- is reproduced with v_readlane.
        if (P == 64 && transform && ch_out == 6 && !sel) {
            const int r = tid;
            const int local = r < rep * n ? r % n : picks[r - rep * n];
            const float* row = data + (size_t)members[lo + local] * ch_in;
            float v[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] = row[k];
            float mean[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 16; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] += sgw::bcast(v[ch], 4 * q + j);
                mean[ch] = (((a[0] + a[1]) + a[2]) + a[3]) / (float)P;
            }
            const float x = v[0] - mean[0], y = v[1] - mean[1], z = v[2] - mean[2];
            const float amax = sgw::wave_max(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))));
            float* q = out + (size_t)r * 6;
            q[0] = x / amax; q[1] = y / amax; q[2] = z / amax; q[3] = v[3]; q[4] = v[4]; q[5] = v[5];
            return;
        }
    }
    for (int r = tid; r < P; r += BLOCK) {
        const int local = r < rep * n ? r % n : picks[r - rep * n];
        const int p = members[lo + local];
        if (sel) sel[(size_t)c * P + r] = p;
        const float* row = data + (size_t)p * ch_in;
        for (int k = 0; k < ch_out; ++k) out[(size_t)r * ch_out + k] = row[k];
    }
    if (!transform) return;

    // model.py:421-423: subtract the mean XYZ of the P rows, divide by the scalar max |XYZ| -- in torch's fp32 arithmetic.
    // `cluster_data[:, :3].mean(0)` reduces the [P, 3] view over its outer dimension with FOUR interleaved fp32
    // accumulators (rows j, j+4, j+8, ... for j = 0..3) that are then added in order, ((a0 + a1) + a2) + a3, and divides by
    // P (probed against torch 2.10 CPU, 900 / 900 columns bit-equal; a plain or pairwise sum matches ~30-60 %).  The order is
    // load-bearing: for a segment whose samples all coincide (a 1-point segment tiled 64 times) the reference's output is
    // the rounding error of this very sum, scaled to O(1) by the division below -- a float64 mean gives 0 / 0 = NaN there
    // and the NaN spreads through MLP1's batch statistics to every distance of the scene.
    __shared__ float part4[4][3];
    __shared__ float mean3[3];
    __shared__ float wmax[16];
    __syncthreads();                                               // the rows above are visible to the whole block
    if (tid < 12) {
        const int j = tid & 3, ch = tid >> 2;
        float a = 0.f;
        for (int r = j; r < P; r += 4) a += out[(size_t)r * ch_out + ch];
        part4[j][ch] = a;
    }
    __syncthreads();
    if (tid < 3) mean3[tid] = (((part4[0][tid] + part4[1][tid]) + part4[2][tid]) + part4[3][tid]) / (float)P;
    __syncthreads();
    const float mx = mean3[0], my = mean3[1], mz = mean3[2];
    float amax = 0.f;
    for (int r = tid; r < P; r += BLOCK) {
        float* q = out + (size_t)r * ch_out;
        const float x = q[0] - mx, y = q[1] - my, z = q[2] - mz;
        q[0] = x; q[1] = y; q[2] = z;
        amax = fmaxf(amax, fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))));
    }
    amax = sgw::wave_max(amax);
    if constexpr (BLOCK > 64) {
        if ((tid & 63) == 0) wmax[tid >> 6] = amax;
        __syncthreads();
        amax = 0.f;
        for (int w = 0; w < BLOCK / 64; ++w) amax = fmaxf(amax, wmax[w]);
    }
    for (int r = tid; r < P; r += BLOCK) {
        float* q = out + (size_t)r * ch_out;
        q[0] = q[0] / amax; q[1] = q[1] / amax; q[2] = q[2] / amax;
    }
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_fps_sample(const float* __restrict__ data, int N, int ch_in,
                                                      const int32_t* __restrict__ members, const int32_t* __restrict__ cl_off,
                                                      int P, int ch_out, int transform, int n_lo, int n_hi, int lds_pts,
                                                      float* __restrict__ samples, int32_t* __restrict__ sel,
                                                      float* __restrict__ ws) {
    fps_sample_body<BLOCK>(data, N, ch_in, members, cl_off, P, ch_out, transform, n_lo, n_hi, lds_pts, samples, sel, ws, blockIdx.x);
}

// structural layer of several scenes: FPS-64 over the original over-segments, all six channels, transformed
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_fps_sample_b(const sg::SlotCtx* __restrict__ cx, int n_lo, int n_hi, int lds_pts, int sorted) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x >= c.S) return;
    fps_sample_body<BLOCK>(c.data, c.N, 6, c.seg_points, c.seg_off, 64, 6, 1, n_lo, n_hi, lds_pts, c.samples, nullptr, c.ws_fps, blockIdx.x,
                           sorted ? c.sperm : nullptr, c.chunk_box, c.seg_chunk_off);
}

}  // namespace

namespace sg {

// max_n < 0: cluster sizes unknown to the host -> launch both size classes with worst-case LDS.
int fps_sample_hint(const float* d_data, int N, int ch_in, const int32_t* d_members, const int32_t* d_cl_off, int C, int P,
                    int ch_out, int transform, float* d_samples, int32_t* d_sel, void* d_ws, size_t ws_bytes, void* stream,
                    int max_n) {
    SG_REQUIRE(N >= 0 && C >= 0 && P > 0 && P <= 4096 && ch_in >= 3 && ch_out >= 3 && ch_out <= ch_in,
               "sg_fps_sample: bad arguments (P=%d ch_in=%d ch_out=%d)", P, ch_in, ch_out);
    if (C == 0) return SG_OK;
    if (ws_bytes < (size_t)N * 16) return sg::fail(SG_ENOMEM, "sg_fps_sample: workspace too small");
    hipStream_t st = sg::as_stream(stream);
    const size_t head = (size_t)P * 4 + 16 * sizeof(Best);
    static bool attr_set = false;
    if (!attr_set) {
        SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fps_sample<1024>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(head + (size_t)kLdsCap * 16 + 4096 * 4)));
        attr_set = true;
    }
    // single-wave class: clusters with n <= kSmallMax, LDS carve sized to the largest of them
    const int small_pts = max_n < 0 ? kSmallMax : std::max(64, std::min(max_n, kSmallMax));
    k_fps_sample<64><<<C, 64, head + (size_t)small_pts * 16, st>>>(d_data, N, ch_in, d_members, d_cl_off, P, ch_out, transform, 1,
                                                                 kSmallMax, small_pts, d_samples, d_sel, (float*)d_ws);
    // 16-wave class: everything larger (blocks whose cluster is small exit immediately); skipped when the
    // host knows there is none
    if (max_n < 0 || max_n > kSmallMax) {
        const int big_pts = max_n < 0 ? kLdsCap : std::min(max_n, kLdsCap);
        k_fps_sample<1024><<<C, 1024, head + (size_t)big_pts * 16, st>>>(d_data, N, ch_in, d_members, d_cl_off, P, ch_out, transform,
                                                                        kSmallMax + 1, INT_MAX, big_pts, d_samples, d_sel,
                                                                        (float*)d_ws);
    }
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int b_fps64(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st, bool sorted) {
    if (bd.nslots == 0 || bd.max_S == 0) return SG_OK;
    const int P = 64;
    const size_t head = (size_t)P * 4 + 16 * sizeof(Best);
    static bool attr_set = false;
    if (!attr_set) {
        SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fps_sample_b<1024>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(head + (size_t)kLdsCap * 16 + 4096 * 4)));
        attr_set = true;
    }
    const int small_pts = std::max(64, std::min(bd.max_seg, kSmallMax));
    k_fps_sample_b<64><<<dim3(bd.max_S, bd.nslots), 64, head + (size_t)small_pts * 16, st>>>(d_ctx, 1, kSmallMax, small_pts, 0);
    if (bd.max_seg > kSmallMax) {
        const int big_pts = std::min(bd.max_seg, kLdsCap);
        k_fps_sample_b<1024><<<dim3(bd.max_S, bd.nslots), 1024, head + (size_t)big_pts * 16, st>>>(d_ctx, kSmallMax + 1, INT_MAX, big_pts, sorted ? 1 : 0);
    }
    SG_LAUNCH_CHECK();
    return SG_OK;
}

}  // namespace sg

extern "C" {

size_t sg_fps_ws_bytes(int N) { return sg::align_up((size_t)std::max(N, 1) * 16); }

int sg_fps_sample(const float* d_data, int N, int ch_in, const int32_t* d_members, const int32_t* d_cl_off, int C, int P,
                  int ch_out, int transform, float* d_samples, int32_t* d_sel, void* d_ws, size_t ws_bytes, void* stream) {
    return sg::fps_sample_hint(d_data, N, ch_in, d_members, d_cl_off, C, P, ch_out, transform, d_samples, d_sel, d_ws, ws_bytes,
                               stream, -1);
}

}  // extern "C"
