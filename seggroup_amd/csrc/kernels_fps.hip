// a4+a5: get_cluster_pointcloud + farthest_point_sampling (reference seggroup/model.py:319-426).
//
// One workgroup per cluster, three size classes: <= 512 points on ONE wave (no workgroup barriers at all), <= 2,048 on four waves,
// larger ones on sixteen.  Up to 8 x BLOCK points the cluster's XYZ and running min-distances live in REGISTERS (eight points per
// thread); beyond that (floors of a scan: 10k-40k points) a step visits only the 32-point chunks of the Morton-sorted segment that the
// new pick can still improve, or -- without the sorted order -- walks the points in LDS / an L2-resident global scratch.  Each of the
// P%n sampling steps is one pass (distance to the newest pick, min with the running value, first-index argmax) followed by a DPP wave
// reduction and, with several waves, one barrier.
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

constexpr int kLdsCap = 8192;        // entries of the LDS carve (16 B each): the chunk-pruned path's per-chunk records, the fallback's points
constexpr int kSmallMax = sg::kSegSmallMax;       // clusters up to this size run on a single wave (eight points per lane in registers)
constexpr int kMidMax = sg::kSegMidMax;        // ... up to this size on four waves, beyond on sixteen
#ifndef SG_FPS_FOUR
#define SG_FPS_FOUR 256
#endif
constexpr int kFourInFlight = SG_FPS_FOUR;
constexpr int kChunkCap = 2600;      // chunks of a segment whose per-chunk records (52 B) fit the LDS carve of the chunk-pruned path

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

#ifdef SG_FPS_SELFCHECK
// debugging aid (make SELFCHECK=1): [0] half-wave maxima that a shuffle butterfly computes differently, [1] the same for the first index,
// [2] picks that a second, serial evaluation by one wave finds differently, [3] picks checked, [4] Ms values read back differently
__device__ unsigned long long g_fps_check[8];
#endif
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
    {
        // Segments of <= 8 x BLOCK points (BLOCK = 64: almost all of them): coordinates and running minima live in REGISTERS, up to eight
        // points per thread (point i = tid + BLOCK u, the same assignment as the strided loop of the fallback below, so the first-index
        // argmax is unchanged).  The 64 dependent steps of a segment were bound by LDS round trips (4 reads + 1 write per point and step).
        // One wave: the newest pick's coordinates come from its owner lane through v_readlane.  Several waves (round 5: 256 threads for
        // 513-2,048 points, 1,024 for up to 8,192 -- the walls of a scan; one wave walking 2,048 points through LDS 63 times was 200 us):
        // every wave reduces to its own winner and parks (distance, index, x, y, z) in LDS, ONE barrier, every thread picks the best of
        // the BLOCK / 64 entries (larger distance, ties -> lower index: np.argmax).  The entries are double-buffered by step parity, so
        // the next step's writes cannot overtake this step's reads and no second barrier is needed.
        // kU = points per thread, picked by the segment's size (block-uniform): a 100-point segment steps over two slots, not eight
        constexpr int kWv = BLOCK / 64;
        struct Cand { float v; int i; float x, y, z; };
        __shared__ Cand cand[2][kWv > 1 ? kWv : 1];
        auto in_registers = [&](auto ku) {
            constexpr int kU = decltype(ku)::value;
            float X[kU], Y[kU], Z[kU], M[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int i = tid + BLOCK * u;
                X[u] = Y[u] = Z[u] = 0.f; M[u] = 0.f;
                if (i < n) {
                    const float* row = data + (size_t)members[lo + i] * ch_in;
                    X[u] = row[0]; Y[u] = row[1]; Z[u] = row[2];
                }
            }
            // branch-free on purpose: written with `if (i < n)` / `if (d > b.v)` the step compiled to 35 branches and 45 exec-mask
            // saves for ~60 VALU of arithmetic, and a segment is a chain of 64 such steps.  A slot past the segment's end carries
            // the running minimum -inf: it never wins the argmax (slot 0 of thread 0 always holds point 0, whose distance is >= 0).
            bool in[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) in[u] = tid + BLOCK * u < n;
            int parity = 0;
            // one sampling step: distances to q, running minima, the block's argmax; returns the pick and leaves ITS coordinates in q
            auto step = [&](float& qx, float& qy, float& qz, bool reset) {
                Best b{-INFINITY, INT_MAX};
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int i = tid + BLOCK * u;
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
                // the wave's winner is wave-uniform: say so, or selecting its coordinates becomes a branch tree
                const int wi = __builtin_amdgcn_readfirstlane(b.i);
                const int wu = wi < n ? wi / BLOCK : 0, wl = wi < n ? (wi % BLOCK) & 63 : 0;
                float sx = X[0], sy = Y[0], sz = Z[0];
#pragma unroll
                for (int w = 1; w < kU; ++w) { sx = wu == w ? X[w] : sx; sy = wu == w ? Y[w] : sy; sz = wu == w ? Z[w] : sz; }
                sx = sgw::bcast(sx, wl); sy = sgw::bcast(sy, wl); sz = sgw::bcast(sz, wl);
                if constexpr (kWv == 1) {
                    qx = sx; qy = sy; qz = sz;
                    return wi;
                } else {
                    if ((tid & 63) == 0) cand[parity][tid >> 6] = Cand{b.v, wi, sx, sy, sz};
                    __syncthreads();
                    Cand r = cand[parity][0];
#pragma unroll
                    for (int w = 1; w < kWv; ++w) {
                        const Cand c_ = cand[parity][w];
                        const bool take = c_.v > r.v || (c_.v == r.v && c_.i < r.i);
                        r.v = take ? c_.v : r.v; r.i = take ? c_.i : r.i; r.x = take ? c_.x : r.x; r.y = take ? c_.y : r.y; r.z = take ? c_.z : r.z;
                    }
                    parity ^= 1;
                    qx = r.x; qy = r.y; qz = r.z;
                    return r.i;
                }
            };
            const float* row0 = data + (size_t)members[lo] * ch_in;       // start at member 0 (model.py:382-386)
            float qx = row0[0], qy = row0[1], qz = row0[2];
            int cur = step(qx, qy, qz, true);                             // the first pick: farthest from member 0 ...
            if (tid == 0) picks[0] = cur;
            int nxt = step(qx, qy, qz, true);                             // ... and the running minima RESET to it
            for (int it = 1; it < rem; ++it) {                            // model.py:389-394
                cur = nxt;
                if (tid == 0) picks[it] = cur;
                if (it + 1 < rem) nxt = step(qx, qy, qz, false);
            }
        };
        if (rem > 0 && n <= 8 * BLOCK) {
            if (n <= BLOCK) in_registers(std::integral_constant<int, 1>{});
            else if (n <= 2 * BLOCK) in_registers(std::integral_constant<int, 2>{});
            else if (n <= 4 * BLOCK) in_registers(std::integral_constant<int, 4>{});
            else in_registers(std::integral_constant<int, 8>{});
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
        if (rem > 0 && !sampled && sperm && (n + 31) / 32 <= lds_pts) {
            // Round 5: a step used to cost ~7 us alone on the GPU (a 29k-point floor: 436 us per launch) -- the chunk boxes and the new
            // pick's coordinates came from global memory, every half wave waited out one chunk's loads at a time, the survivors were
            // listed one LDS atomic each.  Now the carve holds 13 words per chunk (running maximum, its member index and coordinates,
            // the chunk's box, the work list), the list is appended per wave, and a half wave has two chunks' loads in flight (four in
            // the two full passes): one L2 round trip and three barriers per step.
            const int nch = (n + 31) / 32, c0 = seg_chunk_off[c];
            float* cm = lds_f;
            int* ci = reinterpret_cast<int*>(lds_f + lds_pts);
            float* cx = lds_f + 2 * (size_t)lds_pts; float* cy = lds_f + 3 * (size_t)lds_pts; float* cz = lds_f + 4 * (size_t)lds_pts;
            int* wl = reinterpret_cast<int*>(lds_f + 5 * (size_t)lds_pts);
            float* bxl = lds_f + 6 * (size_t)lds_pts;                              // [6][lds_pts]: min xyz, max xyz
            __shared__ int wl_n;
            struct ChunkCand { float v; int i; int ch; };
            __shared__ ChunkCand cc[2][16];
            float* Xs = ws + lo; float* Ys = ws + (size_t)N + lo; float* Zs = ws + 2 * (size_t)N + lo; float* Ms = ws + 3 * (size_t)N + lo;
#pragma unroll 4
            for (int r = tid; r < n; r += BLOCK) {
                const float* row = data + (size_t)members[sperm[lo + r]] * ch_in;
                Xs[r] = row[0]; Ys[r] = row[1]; Zs[r] = row[2];
            }
            for (int ch = tid; ch < nch; ch += BLOCK) {
                const float* bx = chunk_box + (size_t)(c0 + ch) * 8;
#pragma unroll
                for (int k = 0; k < 6; ++k) bxl[(size_t)k * lds_pts + ch] = bx[k];
            }
            if (tid == 0) wl_n = 0;
            __syncthreads();
            const int lane = tid & 63, wave = tid >> 6, hl = lane & 31, half = lane >> 5;
            // kV chunks per half wave at a time: all their loads first, then distances, running minima and each chunk's (max, member index,
            // coordinates).  `chs` may repeat a chunk (odd tails: both halves of a wave run the DPP reductions together): the same values
            // are written again.
            auto visit = [&](auto kv_, const int* chs, float qx, float qy, float qz, bool reset) {
                constexpr int kV = decltype(kv_)::value;
                float x[kV], y[kV], z[kV], mo[kV];
                int li[kV];
#pragma unroll
                for (int v = 0; v < kV; ++v) {
                    const int r = min(32 * chs[v] + hl, n - 1);
                    x[v] = Xs[r]; y[v] = Ys[r]; z[v] = Zs[r];
                    // (device-scope load and store: past the CU's vector L1.  A chunk's minima are read and written by another wave every step,
                    // and a chunk shares its first and last cache line with its neighbours -- which other waves visit in the SAME step: a line one
                    // wave fetches while another wave writes its part of it can sit in the L1 with that part stale, and the next step's visitor
                    // of the neighbour then reads old minima.  Round 5: one sample set in ~5,000 big segments differed from run to run.)
                    mo[v] = reset ? 0.f : __hip_atomic_load(&Ms[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    li[v] = sperm[lo + r] - lo;
                }
#pragma unroll
                for (int v = 0; v < kV; ++v) {
                    const int r = 32 * chs[v] + hl;
                    const bool in = r < n;
                    const float dx = x[v] - qx, dy = y[v] - qy, dz = z[v] - qz;
                    float d = (dx * dx + dy * dy) + dz * dz;
                    if (!reset) d = fminf(mo[v], d);
                    if (in) __hip_atomic_store(&Ms[r], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    d = in ? d : -INFINITY;
                    const int lix = in ? li[v] : INT_MAX;
                    // maxima / minima of the 32 lanes of each half: rows of 16 on the DPP path, row 0 -> 1 and 2 -> 3, then lanes 31 / 63
                    float m = d;
                    m = fmaxf(m, sgw::dpp_f<sgw::kQuadXor1>(m, m)); m = fmaxf(m, sgw::dpp_f<sgw::kQuadXor2>(m, m));
                    m = fmaxf(m, sgw::dpp_f<sgw::kRowRor4>(m, m));  m = fmaxf(m, sgw::dpp_f<sgw::kRowRor8>(m, m));
                    m = fmaxf(m, sgw::dpp_f<sgw::kRowBcast15, 0xA>(m, m));
                    const float mh = half ? sgw::bcast(m, 63) : sgw::bcast(m, 31);
                    int k = d == mh ? lix : INT_MAX;
                    k = min(k, sgw::dpp_i<sgw::kQuadXor1>(k, k)); k = min(k, sgw::dpp_i<sgw::kQuadXor2>(k, k));
                    k = min(k, sgw::dpp_i<sgw::kRowRor4>(k, k));  k = min(k, sgw::dpp_i<sgw::kRowRor8>(k, k));
                    k = min(k, sgw::dpp_i<sgw::kRowBcast15, 0xA>(k, k));
                    const int kh = half ? sgw::bcast(k, 63) : sgw::bcast(k, 31);
#ifdef SG_FPS_SELFCHECK
                    {
                        float m2 = d;
                        for (int o = 16; o > 0; o >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, o));        // butterfly inside the 32 lanes of the half
                        int k2 = d == m2 ? lix : INT_MAX;
                        for (int o = 16; o > 0; o >>= 1) k2 = min(k2, __shfl_xor(k2, o));
                        if (m2 != mh && hl == 0) atomicAdd(&g_fps_check[0], 1ull);
                        if (k2 != kh && hl == 0) atomicAdd(&g_fps_check[1], 1ull);
                    }
#endif
                    if (in && lix == kh) { const int ch = chs[v]; cm[ch] = mh; ci[ch] = kh; cx[ch] = x[v]; cy[ch] = y[v]; cz[ch] = z[v]; }   // one lane of the half
                }
            };
            // argmax over the chunk maxima: larger value, ties -> lower member index; every thread returns the winner and its coordinates
            int parity = 0;
            auto pick = [&](int& bi, float& qx, float& qy, float& qz) {
                float v = -INFINITY; int i = INT_MAX, cch = 0;
                for (int ch = tid; ch < nch; ch += BLOCK) {
                    const float cv = cm[ch]; const int cix = ci[ch];
                    if (cv > v || (cv == v && cix < i)) { v = cv; i = cix; cch = ch; }
                }
                const float wm = sgw::wave_max(v);
                const int wi = sgw::wave_min(v == wm ? i : INT_MAX);
                const unsigned long long own = __ballot(v == wm && i == wi);
                if (lane == __ffsll((unsigned long long)own) - 1) cc[parity][wave] = ChunkCand{wm, wi, cch};
                __syncthreads();
                ChunkCand r = cc[parity][0];
#pragma unroll
                for (int w = 1; w < 16; ++w) {
                    const ChunkCand o = cc[parity][w];
                    const bool take = o.v > r.v || (o.v == r.v && o.i < r.i);
                    r.v = take ? o.v : r.v; r.i = take ? o.i : r.i; r.ch = take ? o.ch : r.ch;
                }
                parity ^= 1;
                bi = r.i; qx = cx[r.ch]; qy = cy[r.ch]; qz = cz[r.ch];
#ifdef SG_FPS_SELFCHECK
                if (tid == 0) {
                    float bv = -INFINITY; int bix = INT_MAX;
                    for (int ch = 0; ch < nch; ++ch) { const float cv = cm[ch]; const int cix = ci[ch]; if (cv > bv || (cv == bv && cix < bix)) { bv = cv; bix = cix; } }
                    atomicAdd(&g_fps_check[3], 1ull);
                    if (bix != bi) atomicAdd(&g_fps_check[2], 1ull);
                }
                __syncthreads();
#endif
            };
            auto full_pass = [&](float qx, float qy, float qz) {
                // A pass rewrites every chunk's record, the winner's coordinates included -- which the waves behind the last pick()'s barrier may still
                // be reading (qx = cx[r.ch] ...).  Round 5: without this barrier about one sample set in 4,000 big segments was wrong, a wave having
                // read the first pick's x from the old record and its y or z from the new one (found by tools/r05_repro.py with SG_ENGINE_HASH=1,
                // which samples every scene a second time through the single-scene entry).  The steps of the main loop need none: their
                // visits come behind the work list's barrier.
                __syncthreads();
                for (int e = 2 * wave; e < nch; e += 128) {                       // four chunks per half wave in flight
                    const int chs[4] = {min(e + half, nch - 1), min(e + 32 + half, nch - 1), min(e + 64 + half, nch - 1), min(e + 96 + half, nch - 1)};
                    visit(std::integral_constant<int, 4>{}, chs, qx, qy, qz, true);
                }
                __syncthreads();
            };
            int bi;
            float qx, qy, qz;
            {
                const float* row0 = data + (size_t)members[lo] * ch_in;           // start at member 0 (model.py:382-386)
                full_pass(row0[0], row0[1], row0[2]);
                pick(bi, qx, qy, qz);
            }
            int cur = bi;
            if (tid == 0) picks[0] = cur;
            full_pass(qx, qy, qz);                                                // reset to the first pick
            pick(bi, qx, qy, qz);
            for (int it = 1; it < rem; ++it) {                                    // model.py:389-394
                cur = bi;
                if (tid == 0) picks[it] = cur;
                if (it + 1 < rem) {
                    // the chunks the new pick can still improve, appended per wave (the barrier inside the last pick() separates this
                    // step's list from the previous step's readers; wl_n is reset behind the barrier below)
                    for (int ch0 = 0; ch0 < nch; ch0 += BLOCK) {
                        const int ch = ch0 + tid;
                        bool need = false;
                        if (ch < nch) {
                            const float gx = fmaxf(fmaxf(bxl[ch] - qx, qx - bxl[3 * (size_t)lds_pts + ch]), 0.f),
                                        gy = fmaxf(fmaxf(bxl[(size_t)lds_pts + ch] - qy, qy - bxl[4 * (size_t)lds_pts + ch]), 0.f),
                                        gz = fmaxf(fmaxf(bxl[2 * (size_t)lds_pts + ch] - qz, qz - bxl[5 * (size_t)lds_pts + ch]), 0.f);
                            need = ((gx * gx + gy * gy) + gz * gz) * 0.999999f < cm[ch];
                        }
                        const unsigned long long mask = __ballot(need);
                        int base = 0;
                        if (lane == 0 && mask) base = atomicAdd(&wl_n, __popcll(mask));
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (need) wl[base + __popcll(mask & ((1ull << lane) - 1ull))] = ch;
                    }
                    __syncthreads();
                    const int nw = wl_n;
                    // both halves of a wave must run visit() together (its reductions are wave-wide DPP networks): an odd tail
                    // revisits the last listed chunk, which changes nothing
                    // (long lists -- the first steps touch most of the segment -- four chunks per half wave in flight, short ones two or one;
                    // SG_FPS_FOUR: the list length from which four are in flight: a build-time knob, 256 since round 5)
                    if (nw > kFourInFlight) {
                        for (int e = 2 * wave; e < nw; e += 128) {
                            const int chs[4] = {wl[min(e + half, nw - 1)], wl[min(e + 32 + half, nw - 1)], wl[min(e + 64 + half, nw - 1)], wl[min(e + 96 + half, nw - 1)]};
                            visit(std::integral_constant<int, 4>{}, chs, qx, qy, qz, false);
                        }
                    } else {
                        for (int e = 2 * wave; e < nw; e += 64) {
                            const int chs[2] = {wl[min(e + half, nw - 1)], wl[min(e + 32 + half, nw - 1)]};
                            if (e + 32 < nw) visit(std::integral_constant<int, 2>{}, chs, qx, qy, qz, false);
                            else visit(std::integral_constant<int, 1>{}, chs, qx, qy, qz, false);
                        }
                    }
                    __syncthreads();
                    if (tid == 0) wl_n = 0;
                    pick(bi, qx, qy, qz);
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
    int seg = blockIdx.x;
    if constexpr (BLOCK == 64) { if (seg >= c.S) return; }
    else if constexpr (BLOCK == 256) { if (seg >= c.n_mid) return; seg = c.mid_segs[seg]; }      // the host's list of this class's segments
    else { if (seg >= c.n_big) return; seg = c.big_segs[seg]; }
    fps_sample_body<BLOCK>(c.data, c.N, 6, c.seg_points, c.seg_off, 64, 6, 1, n_lo, n_hi, lds_pts, c.samples, nullptr, c.ws_fps, seg,
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
                                   (int)(head + std::max((size_t)kLdsCap * 16, (size_t)kChunkCap * 52) + 4096 * 4)));
        attr_set = true;
    }
    // single-wave class: clusters with n <= kSmallMax (registers: no carve)
    k_fps_sample<64><<<C, 64, head, st>>>(d_data, N, ch_in, d_members, d_cl_off, P, ch_out, transform, 1, kSmallMax, 0, d_samples, d_sel, (float*)d_ws);
    // four- and sixteen-wave classes (blocks whose cluster is not theirs exit immediately); skipped when the host knows there is none
    if (max_n < 0 || max_n > kSmallMax)
        k_fps_sample<256><<<C, 256, head, st>>>(d_data, N, ch_in, d_members, d_cl_off, P, ch_out, transform, kSmallMax + 1, kMidMax, 0, d_samples, d_sel,
                                               (float*)d_ws);
    if (max_n < 0 || max_n > kMidMax) {
        // beyond 8 x 1024 points (no sorted order on this entry): the points in the carve while they fit, in the global scratch otherwise
        const int big_pts = max_n < 0 ? kLdsCap : std::min(max_n, kLdsCap);
        k_fps_sample<1024><<<C, 1024, head + (size_t)big_pts * 16, st>>>(d_data, N, ch_in, d_members, d_cl_off, P, ch_out, transform,
                                                                        kMidMax + 1, INT_MAX, big_pts, d_samples, d_sel,
                                                                        (float*)d_ws);
    }
    SG_LAUNCH_CHECK();
    return SG_OK;
}

bool fps_has_big_class(const BatchDims& bd) { return bd.max_big > 0; }

int b_fps64(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st, bool sorted, int which) {
    if (bd.nslots == 0 || bd.max_S == 0) return SG_OK;
    const int P = 64;
    const size_t head = (size_t)P * 4 + 16 * sizeof(Best);
    static bool attr_set = false;
    if (!attr_set) {
        SG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fps_sample_b<1024>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(head + std::max((size_t)kLdsCap * 16, (size_t)kChunkCap * 52) + 4096 * 4)));
        attr_set = true;
    }
    if (which & 1) {
        k_fps_sample_b<64><<<dim3(bd.max_S, bd.nslots), 64, head, st>>>(d_ctx, 1, kSmallMax, 0, 0);
        if (bd.max_mid > 0) k_fps_sample_b<256><<<dim3(bd.max_mid, bd.nslots), 256, head, st>>>(d_ctx, kSmallMax + 1, kMidMax, 0, 0);
    }
    if ((which & 2) && bd.max_big > 0) {
        // the carve: 13 words per chunk for the chunk-pruned path (segments beyond 8,192 points, Morton-sorted: up to kChunkCap chunks =
        // ~83k points; a larger segment takes the fallback), or the fallback's points (16 bytes each)
        const int big_pts = sorted ? std::max(64, std::min(sg::cdiv(bd.max_seg, 32), kChunkCap)) : std::max(64, std::min(bd.max_seg, kLdsCap));
        k_fps_sample_b<1024><<<dim3(bd.max_big, bd.nslots), 1024, head + (size_t)big_pts * (sorted ? 52 : 16), st>>>(d_ctx, kMidMax + 1, INT_MAX, big_pts,
                                                                                                                  sorted ? 1 : 0);
    }
    SG_LAUNCH_CHECK();
    return SG_OK;
}

}  // namespace sg

extern "C" {

#ifdef SG_FPS_SELFCHECK
int sg_debug_fps_check(unsigned long long* h_out) {
    SG_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_fps_check), sizeof(unsigned long long) * 8));
    unsigned long long z[8] = {0};
    SG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fps_check), z, sizeof z));
    return SG_OK;
}
#endif

size_t sg_fps_ws_bytes(int N) { return sg::align_up((size_t)std::max(N, 1) * 16); }

int sg_fps_sample(const float* d_data, int N, int ch_in, const int32_t* d_members, const int32_t* d_cl_off, int C, int P,
                  int ch_out, int transform, float* d_samples, int32_t* d_sel, void* d_ws, size_t ws_bytes, void* stream) {
    return sg::fps_sample_hint(d_data, N, ch_in, d_members, d_cl_off, C, P, ch_out, transform, d_samples, d_sel, d_ws, ws_bytes,
                               stream, -1);
}

}  // extern "C"
