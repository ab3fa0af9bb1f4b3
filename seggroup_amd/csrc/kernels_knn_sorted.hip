// In-cluster kNN-20 with spatially sorted candidates (reference seggroup/model.py:512-522, 30-36).
//
// Same result as k_cluster_knn / k_cluster_knn_pruned (bit-identical tables), much less work:
//   * once per scene the points of every ORIGINAL over-segment are put in Morton order (one device radix sort
//     keyed by segment | 30-bit Morton code inside the segment's box) and every run of 32 sorted points gets a
//     bounding box;
//   * per layer the kNN operand [x, y, z, |p|^2] and each point's member position are laid out in that order
//     (clusters stay contiguous: only the order INSIDE a segment changes);
//   * a workgroup = 64 spatially adjacent queries x 4 waves; the cluster's 32-point chunks are dealt round-robin
//     to the waves and a chunk is scanned only if its box can still beat some lane's 20th best.  Adjacent queries
//     share their neighbourhood, so after the queries' own segment almost every chunk is rejected by its box.
// The 64-bit keys carry the MEMBER index, so ties resolve exactly as in the member-order scan ("lower member
// index wins"), whatever order candidates arrive in.
#include <cstdlib>


#include "engine_ctx.h"
#include "knn_device.h"
#include "sg_common.h"
#include "wave_ops.h"

namespace {

using namespace sgknn;

// Work counters / cycle stamps of the kNN kernels exist only in profiling builds (make PROFILE=1): in a release build
// `dbg` folds to 0 at compile time, the statistics code and its device globals disappear, and the library has no
// process-wide mutable state.
#ifdef SG_KNN_PROFILE
constexpr bool kKnnProfile = true;
#else
constexpr bool kKnnProfile = false;
#endif

constexpr int kChunkPts = 32;
constexpr int kQuadS = 4;

// Morton code of a point inside its segment's box, 10 bits per axis.  Every axis is quantised by its own extent (balanced
// splits: measurably tighter chunk boxes on compact segments than isotropic cells, ~10 % of the kNN time) UNLESS that extent
// is less than a quarter of the largest one: such a thin axis is quantised by the largest extent instead.  Otherwise the thin
// axis of a planar segment (a floor: 8 m x 6 m x 2 cm of scanner noise) contributes random high bits, which scatters spatial
// neighbours over the order and leaves the 32-point chunk boxes ~10x larger than they need be.
__device__ inline unsigned int spread10(unsigned int v);
__device__ inline unsigned int morton30(const float* __restrict__ r, const float* __restrict__ b) {
    const float e[3] = {b[3] - b[0], b[4] - b[1], b[5] - b[2]};
    const float emax = fmaxf(fmaxf(e[0], e[1]), e[2]);
    unsigned int q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float ext = e[k] >= 0.25f * emax ? e[k] : emax;
        const float t = ext > 0.f ? (r[k] - b[k]) / ext : 0.f;
        q[k] = (unsigned int)fminf(fmaxf(t * 1023.f, 0.f), 1023.f);
    }
    return spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
}

__device__ inline unsigned int spread10(unsigned int v) {      // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// Segment box + Morton sort + chunk boxes in one launch (round 1: a key kernel, a library radix sort = 9 launches, a box kernel), for scenes
// whose largest segment has <= kSortCap points: block s owns segment s, computes its box, sorts (morton30 << 32 | local
// index) in LDS with a bitonic network -- the local index in the low bits reproduces the radix sort's stable order exactly
// -- and boxes its 32-point chunks.  At ~1000 scenes/s the pipelines issue ~100k runtime calls per second, so launches
// saved are throughput.
constexpr int kSortCap = sg::kSegMidMax;      // = the boundary of the engine's list of big segments (SlotCtx::big_segs)
__device__ __forceinline__ void segment_sort_boxes_body(const float* __restrict__ data, const int32_t* __restrict__ seg_points,
                                                        const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_chunk_off,
                                                        float* __restrict__ segbox, int32_t* __restrict__ sperm,
                                                        float* __restrict__ chunk_box, double* __restrict__ seg_sums, int s) {
    __shared__ unsigned long long key[kSortCap];
    __shared__ float red[4][8];
    __shared__ double dred[4][3];
    __shared__ float bx[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lo = seg_off[s], n = seg_off[s + 1] - lo;
    if (n > kSortCap) return;                                  // k_bigseg_sort_boxes' share (block-uniform)
    // 1. segment box {min xyz, max xyz, max |p|^2}
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, xx = 0.f;
    double sm[3] = {0.0, 0.0, 0.0};                          // coordinate sums: the host builds cluster centroids from them
    for (int i = tid; i < n; i += 256) {
        const float* r = data + (size_t)seg_points[lo + i] * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], r[k]); mx[k] = fmaxf(mx[k], r[k]); sm[k] += (double)r[k]; }
        xx = fmaxf(xx, (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
    }
    // wave reductions on the DPP path (wave_ops.h; a `__shfl_xor` step is an LDS round trip, twice for a double)
#pragma unroll
    for (int k = 0; k < 3; ++k) { mn[k] = sgw::wave_min(mn[k]); mx[k] = sgw::wave_max(mx[k]); }
    xx = sgw::wave_max(xx);
    if (seg_sums) {
#pragma unroll
        for (int k = 0; k < 3; ++k) sm[k] = sgw::wave_sum(sm[k]);
        if (lane == 0) { dred[wave][0] = sm[0]; dred[wave][1] = sm[1]; dred[wave][2] = sm[2]; }
    }
    if (lane == 0) { red[wave][0] = mn[0]; red[wave][1] = mn[1]; red[wave][2] = mn[2]; red[wave][3] = mx[0]; red[wave][4] = mx[1]; red[wave][5] = mx[2]; red[wave][6] = xx; }
    __syncthreads();
    if (tid < 7) {
        float v = red[0][tid];
        for (int w = 1; w < 4; ++w) v = tid < 3 ? fminf(v, red[w][tid]) : fmaxf(v, red[w][tid]);
        bx[tid] = v;
        segbox[(size_t)s * 8 + tid] = v;
    }
    if (tid == 7) segbox[(size_t)s * 8 + 7] = 0.f;
    if (seg_sums && tid >= 8 && tid < 11) seg_sums[(size_t)s * 3 + (tid - 8)] = ((dred[0][tid - 8] + dred[1][tid - 8]) + dred[2][tid - 8]) + dred[3][tid - 8];
    __syncthreads();
    // 2. keys (morton30 inside the box just computed), padded to a power of two with ~0
    int m2 = 64;
    while (m2 < n) m2 <<= 1;
    for (int i = tid; i < m2; i += 256) {
        unsigned long long kk = ~0ull;
        if (i < n) {
            const float* r = data + (size_t)seg_points[lo + i] * 6;
            const unsigned int m = morton30(r, bx);
            kk = ((unsigned long long)m << 32) | (unsigned int)i;
        }
        key[i] = kk;
    }
    __syncthreads();
    // 3. bitonic sort, ascending
    for (int k = 2; k <= m2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (m2 >> 1); t += 256) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                const unsigned long long a = key[i], b = key[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { key[i] = b; key[p] = a; }
            }
            __syncthreads();
        }
    // 4. sorted position -> index into the segment CSR
    for (int r = tid; r < n; r += 256) sperm[lo + r] = lo + (int)(key[r] & 0xffffffffull);
    // 5. boxes of the 32-point chunks (two per wave step)
    const int c0 = seg_chunk_off[s], half = lane >> 5, l = lane & 31;
    for (int j = 2 * wave + half; j * kChunkPts < n; j += 8) {
        const int t = j * kChunkPts + l;
        float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY}, cxx = 0.f;
        if (t < n) {
            const float* r = data + (size_t)seg_points[lo + (int)(key[t] & 0xffffffffull)] * 6;
#pragma unroll
            for (int k = 0; k < 3; ++k) { cmn[k] = r[k]; cmx[k] = r[k]; }
            cxx = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { cmn[k] = sgw::half_min(cmn[k], lane); cmx[k] = sgw::half_max(cmx[k], lane); }
        cxx = sgw::half_max(cxx, lane);
        if (l == 0) {
            float* b = chunk_box + (size_t)(c0 + j) * 8;
            b[0] = cmn[0]; b[1] = cmn[1]; b[2] = cmn[2]; b[3] = cmx[0]; b[4] = cmx[1]; b[5] = cmx[2]; b[6] = cxx; b[7] = 0.f;
        }
    }
}
__global__ __launch_bounds__(256) void k_segment_sort_boxes(const float* __restrict__ data, const int32_t* __restrict__ seg_points,
                                                            const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_chunk_off,
                                                            float* __restrict__ segbox, int32_t* __restrict__ sperm,
                                                            float* __restrict__ chunk_box, double* __restrict__ seg_sums) {
    segment_sort_boxes_body(data, seg_points, seg_off, seg_chunk_off, segbox, sperm, chunk_box, seg_sums, blockIdx.x);
}
__global__ __launch_bounds__(256) void k_segment_sort_boxes_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x >= c.S) return;
    segment_sort_boxes_body(c.data, c.seg_points, c.seg_off, c.seg_chunk_off, c.segbox, c.sperm, c.chunk_box, c.seg_sums, blockIdx.x);
}

// Segments of more than kSortCap points (floors, walls: 10k-40k points in ScanNet's over-segmentation) -- same outputs, same order
// (ascending (morton30, index)), no library sort.  Round 5: three launches instead of one block per segment doing everything
// (a 29k-point floor kept one workgroup busy for a millisecond: a serial run table by one thread and ~20 LDS sorts one after another):
//   k_bigseg_box / k_bigseg_keys / k_bigseg_scatter (round 6; round 5: one 1024-thread block per big segment did all three):  box + coordinate sums;
//       keys into global scratch + histogram of the TOP 12 Morton bits (16^3 cells); scan; scatter into cell order (keysB); the cells' first positions
//       go to the segment's scratch record (BigScratch below);
//   k_bigseg_runs (one block per window of kWin = 1,024 positions of the scene's sorted CSR):  the RUN that starts in the window -- from the
//       first cell that starts in it to the first cell that starts in a later window: whole cells, <= kWin + the last cell's size - 1 keys
//       -- is sorted in LDS by the full 64-bit key (cells are already in order, so the concatenation of the runs is the total order) and
//       emitted as sorted positions.  Runs are independent: they sort side by side on as many CUs.  A cell of more than kWin points (1/4096
//       of the segment's box: massive duplication) makes a run that is cut into kSortCap pieces sorted one by one -- the order inside such a
//       cell is then not the full sort, which only loosens the chunk boxes there (the kNN tables do not depend on this order);
//   k_bigseg_boxes (same windows):  boxes of the 32-point chunks that start in the window.
constexpr int kBigBlock = 1024, kBins = 4096, kWin = 1024;
// Round 6: the bucket step is SPLIT OVER kBigPieces WORKGROUPS PER SEGMENT, in three launches (box | keys + histogram | scan + scatter).  One
// 1,024-thread block per segment did all of it: three passes of dependent gathers (seg_points[i] -> data row) over 30-40k points through ONE
// CU's memory pipeline, 176-188 us per launch of 8 ScanNet-shaped scenes -- the longest link but one of phase P0's chain of skinny launches
// (DESIGN.md 5c).  The passes are embarrassingly parallel but for three hand-overs (the segment's box before any key, all counts before the scan,
// the scan before the scatter), and a launch boundary is the cheapest device-wide hand-over there is.  Per big segment a scratch record
// (BigScratch, indexed by lo / (kSortCap + 1): two big segments start at least kSortCap + 1 positions apart) behind the two key arrays holds
// the pieces' partial boxes / sums, the cell histogram, the scatter cursors and the cells' first positions (what k_bigseg_runs looks its runs
// up in: round 5 kept that table in the head of the segment's own keysA range, written by the one block when it was done with it).
// The order INSIDE a cell is now the order in which pieces' atomics land -- k_bigseg_runs sorts every run by the full 64-bit key, so the sorted
// order, the boxes and everything behind them are the same bits as before; the coordinate sums are combined piece by piece in a fixed order.
constexpr int kBigPieces = 16, kPieceBlock = 256;
struct BigScratch {
    float box[kBigPieces][8];          // per piece: min xyz | max xyz | max |p|^2 | -
    double sums[kBigPieces][4];        // per piece: sum x | y | z | -
    int hist[kBins];                   // points per cell (top 12 Morton bits)
    int cursor[kBins];                 // scatter cursors
    int first[kBins];                  // exclusive scan of hist: first position of every cell
};
static_assert(kBins % kBigPieces == 0 && kBins / kBigPieces == kPieceBlock, "every piece zeroes its own slice of the histogram: one entry per thread");
__host__ __device__ inline size_t bigseg_scratch_slots(int N) { return (size_t)N / (kSortCap + 1) + 1; }
__device__ __forceinline__ BigScratch& bigseg_scratch(unsigned long long* keysA, int N, int lo) {
    return reinterpret_cast<BigScratch*>(keysA + 2 * (size_t)N)[lo / (kSortCap + 1)];
}
__device__ __forceinline__ void piece_range(int n, int piece, int& i0, int& i1) {
    const int L = (n + kBigPieces - 1) / kBigPieces;
    i0 = min(n, piece * L); i1 = min(n, i0 + L);
}

// launch 1: partial box + coordinate sums of one piece; the piece's slice of the histogram and of the cursors is cleared
__device__ __forceinline__ void bigseg_box_body(const float* __restrict__ data, const int32_t* __restrict__ seg_points, const int32_t* __restrict__ seg_off,
                                                unsigned long long* __restrict__ keysA, int N, int s, int piece) {
    const int lo = seg_off[s], n = seg_off[s + 1] - lo;
    if (n <= kSortCap) return;
    BigScratch& sc = bigseg_scratch(keysA, N, lo);
    __shared__ float red[kPieceBlock / 64][8];
    __shared__ double dred[kPieceBlock / 64][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kW = kPieceBlock / 64;
    int i0, i1;
    piece_range(n, piece, i0, i1);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, xx = 0.f;
    double sm[3] = {0.0, 0.0, 0.0};
#pragma unroll 4
    for (int i = i0 + tid; i < i1; i += kPieceBlock) {
        const float* r = data + (size_t)seg_points[lo + i] * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], r[k]); mx[k] = fmaxf(mx[k], r[k]); sm[k] += (double)r[k]; }
        xx = fmaxf(xx, (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); sm[k] += __shfl_xor(sm[k], o); }
        xx = fmaxf(xx, __shfl_xor(xx, o));
    }
    if (lane == 0) {
        red[wave][0] = mn[0]; red[wave][1] = mn[1]; red[wave][2] = mn[2]; red[wave][3] = mx[0]; red[wave][4] = mx[1]; red[wave][5] = mx[2]; red[wave][6] = xx;
        dred[wave][0] = sm[0]; dred[wave][1] = sm[1]; dred[wave][2] = sm[2];
    }
    sc.hist[piece * kPieceBlock + tid] = 0;
    sc.cursor[piece * kPieceBlock + tid] = 0;
    __syncthreads();
    if (tid < 7) {
        float v = red[0][tid];
        for (int w = 1; w < kW; ++w) v = tid < 3 ? fminf(v, red[w][tid]) : fmaxf(v, red[w][tid]);
        sc.box[piece][tid] = v;
    }
    if (tid >= 8 && tid < 11) {
        double t = 0.0;
        for (int w = 0; w < kW; ++w) t += dred[w][tid - 8];
        sc.sums[piece][tid - 8] = t;
    }
}

// launch 2: the segment's box from the pieces' (every piece combines them itself, in piece order; piece 0 publishes box and sums), then the
// piece's keys and their cells' counts
__device__ __forceinline__ void bigseg_keys_body(const float* __restrict__ data, const int32_t* __restrict__ seg_points, const int32_t* __restrict__ seg_off,
                                                 float* __restrict__ segbox, double* __restrict__ seg_sums, unsigned long long* __restrict__ keysA, int N,
                                                 int s, int piece) {
    const int lo = seg_off[s], n = seg_off[s + 1] - lo;
    if (n <= kSortCap) return;
    BigScratch& sc = bigseg_scratch(keysA, N, lo);
    __shared__ float bx[8];
    const int tid = threadIdx.x;
    if (tid < 7) {
        float v = sc.box[0][tid];
        for (int p = 1; p < kBigPieces; ++p) v = tid < 3 ? fminf(v, sc.box[p][tid]) : fmaxf(v, sc.box[p][tid]);
        bx[tid] = v;
        if (piece == 0) segbox[(size_t)s * 8 + tid] = v;
    }
    if (piece == 0 && tid == 7) segbox[(size_t)s * 8 + 7] = 0.f;
    if (piece == 0 && seg_sums && tid >= 8 && tid < 11) {
        // NOTE: the sum is carried in double; its grouping (pieces x waves here, 4 waves in the small kernel) does not show in the fp32
        // centroids the host forms from it
        double t = 0.0;
        for (int p = 0; p < kBigPieces; ++p) t += sc.sums[p][tid - 8];
        seg_sums[(size_t)s * 3 + (tid - 8)] = t;
    }
    // the piece's cell counts in LDS first: a floor's points share a few hundred of the 4,096 cells, and one global atomic per POINT made this launch
    // (and the scatter) wait on ~150 atomics per address; now one per (piece, occupied cell)
    __shared__ int lh[kBins];
    for (int b = tid; b < kBins; b += kPieceBlock) lh[b] = 0;
    __syncthreads();
    int i0, i1;
    piece_range(n, piece, i0, i1);
#pragma unroll 4
    for (int i = i0 + tid; i < i1; i += kPieceBlock) {
        const float* r = data + (size_t)seg_points[lo + i] * 6;
        const unsigned int m = morton30(r, bx);
        keysA[lo + i] = ((unsigned long long)m << 32) | (unsigned int)i;
        atomicAdd(&lh[m >> 18], 1);
    }
    __syncthreads();
    for (int b = tid; b < kBins; b += kPieceBlock) {
        const int c = lh[b];
        if (c) atomicAdd(&sc.hist[b], c);
    }
}

// launch 3: exclusive scan of the 4096 counts (every piece for itself: 16 KB out of L2; piece 0 publishes the cells' first positions), then
// the piece's keys go to their cells
__device__ __forceinline__ void bigseg_scatter_body(const int32_t* __restrict__ seg_off, unsigned long long* __restrict__ keysA,
                                                    unsigned long long* __restrict__ keysB, int N, int s, int piece) {
    const int lo = seg_off[s], n = seg_off[s + 1] - lo;
    if (n <= kSortCap) return;
    BigScratch& sc = bigseg_scratch(keysA, N, lo);
    __shared__ int first[kBins];
    __shared__ int wsum[kPieceBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kPer = kBins / kPieceBlock;                   // 16 consecutive cells per thread
    int c[kPer], sum = 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) { c[u] = sc.hist[kPer * tid + u]; sum += c[u]; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = incl - sum;
    for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
    for (int u = 0; u < kPer; ++u) { first[kPer * tid + u] = off; if (piece == 0) sc.first[kPer * tid + u] = off; off += c[u]; }
    // the piece reserves its keys' places cell by cell (one global atomic per occupied cell: see k_bigseg_keys), then hands them out from LDS
    __shared__ int lh[kBins];
    for (int b = tid; b < kBins; b += kPieceBlock) lh[b] = 0;
    __syncthreads();
    int i0, i1;
    piece_range(n, piece, i0, i1);
#pragma unroll 4
    for (int i = i0 + tid; i < i1; i += kPieceBlock) atomicAdd(&lh[(unsigned int)(keysA[lo + i] >> 32) >> 18], 1);
    __syncthreads();
    for (int b = tid; b < kBins; b += kPieceBlock) {
        const int c = lh[b];
        if (c) first[b] += atomicAdd(&sc.cursor[b], c);          // first[b] = where THIS piece's keys of cell b start
        lh[b] = 0;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = i0 + tid; i < i1; i += kPieceBlock) {
        const unsigned long long k = keysA[lo + i];
        const unsigned int cell = (unsigned int)(k >> 32) >> 18;
        keysB[lo + first[cell] + atomicAdd(&lh[cell], 1)] = k;
    }
}

// the segment that holds position p of the sorted CSR (seg_off ascending, seg_off[S] = N; empty segments are skipped by the upper bound)
__device__ __forceinline__ int segment_of_position(const int32_t* __restrict__ seg_off, int S, int p) {
    int a = 0, b = S;                                          // largest s with seg_off[s] <= p
    while (b - a > 1) {
        const int m = (a + b) >> 1;
        if (seg_off[m] <= p) a = m; else b = m;
    }
    return a;
}
__device__ __forceinline__ int first_cell_at_or_behind(const int* __restrict__ cells, int x) {     // lower bound over the 4096 cell starts
    int a = 0, b = kBins;
    while (a < b) {
        const int m = (a + b) >> 1;
        if (cells[m] < x) a = m + 1; else b = m;
    }
    return a;
}

__device__ __forceinline__ void bigseg_runs_body(const int32_t* __restrict__ seg_off, int S, int N, const unsigned long long* __restrict__ keysA,
                                                 const unsigned long long* __restrict__ keysB, int32_t* __restrict__ sperm, int win) {
    __shared__ unsigned long long key[kSortCap];
    const int tid = threadIdx.x;
    const int w0 = win * kWin, w1 = min(N, w0 + kWin);
    if (w0 >= N) return;
    const int sa = segment_of_position(seg_off, S, w0), sb = segment_of_position(seg_off, S, w1 - 1);
    // only the first and the last segment of a window can be big (those in between lie inside it: <= kWin points)
    for (int pass = 0; pass < 2; ++pass) {
        const int s = pass == 0 ? sa : sb;
        if (pass == 1 && sb == sa) break;
        const int lo = seg_off[s], n = seg_off[s + 1] - lo;
        if (n <= kSortCap) continue;                             // block-uniform
        const int* cells = bigseg_scratch(const_cast<unsigned long long*>(keysA), N, lo).first;
        const int b0 = first_cell_at_or_behind(cells, max(w0 - lo, 0));
        const int h0 = b0 < kBins ? cells[b0] : n;
        if (h0 >= n || lo + h0 >= w1) continue;                  // no cell starts in this window
        const int b1 = first_cell_at_or_behind(cells, w1 - lo);
        const int h1 = b1 < kBins ? cells[b1] : n;
        for (int p0 = h0; p0 < h1; p0 += kSortCap) {             // one piece unless a cell holds more than kWin points
            const int m = min(kSortCap, h1 - p0);
            int m2 = 64;
            while (m2 < m) m2 <<= 1;
            for (int i = tid; i < m2; i += kBigBlock) key[i] = i < m ? keysB[lo + p0 + i] : ~0ull;
            __syncthreads();
            for (int k = 2; k <= m2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = tid; t < (m2 >> 1); t += kBigBlock) {
                        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                        const unsigned long long a = key[i], b = key[p];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { key[i] = b; key[p] = a; }
                    }
                    __syncthreads();
                }
            for (int i = tid; i < m; i += kBigBlock) sperm[lo + p0 + i] = lo + (int)(key[i] & 0xffffffffull);
            __syncthreads();
        }
    }
}

__device__ __forceinline__ void bigseg_boxes_body(const float* __restrict__ data, const int32_t* __restrict__ seg_points, const int32_t* __restrict__ seg_off,
                                                  int S, int N, const int32_t* __restrict__ seg_chunk_off, const int32_t* __restrict__ sperm,
                                                  float* __restrict__ chunk_box, int win) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l = lane & 31;
    const int w0 = win * kWin, w1 = min(N, w0 + kWin);
    if (w0 >= N) return;
    const int sa = segment_of_position(seg_off, S, w0), sb = segment_of_position(seg_off, S, w1 - 1);
    for (int pass = 0; pass < 2; ++pass) {
        const int s = pass == 0 ? sa : sb;
        if (pass == 1 && sb == sa) break;
        const int lo = seg_off[s], n = seg_off[s + 1] - lo;
        if (n <= kSortCap) continue;
        const int c0 = seg_chunk_off[s];
        const int j0 = (max(w0 - lo, 0) + kChunkPts - 1) / kChunkPts;              // chunks whose first position lies in [w0, w1)
        const int j1 = min((n + kChunkPts - 1) / kChunkPts, (w1 - lo + kChunkPts - 1) / kChunkPts);
        for (int j = j0 + 2 * wave + half; j < j1 + half; j += 2 * (blockDim.x >> 6)) {      // both halves of a wave run the reductions together
            const bool live = j < j1;
            const int t = j * kChunkPts + l;
            float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY}, cxx = 0.f;
            if (live && t < n) {
                const float* r = data + (size_t)seg_points[sperm[lo + t]] * 6;
#pragma unroll
                for (int k = 0; k < 3; ++k) { cmn[k] = r[k]; cmx[k] = r[k]; }
                cxx = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) { cmn[k] = sgw::half_min(cmn[k], lane); cmx[k] = sgw::half_max(cmx[k], lane); }
            cxx = sgw::half_max(cxx, lane);
            if (live && l == 0) {
                float* b = chunk_box + (size_t)(c0 + j) * 8;
                b[0] = cmn[0]; b[1] = cmn[1]; b[2] = cmn[2]; b[3] = cmx[0]; b[4] = cmx[1]; b[5] = cmx[2]; b[6] = cxx; b[7] = 0.f;
            }
        }
    }
}

// grid (segments, kBigPieces): a block is one piece of one segment (small segments exit at once)
__global__ __launch_bounds__(kPieceBlock) void k_bigseg_box(const float* __restrict__ data, const int32_t* __restrict__ seg_points, const int32_t* __restrict__ seg_off,
                                                            unsigned long long* __restrict__ keysA, int N) {
    bigseg_box_body(data, seg_points, seg_off, keysA, N, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(kPieceBlock) void k_bigseg_keys(const float* __restrict__ data, const int32_t* __restrict__ seg_points, const int32_t* __restrict__ seg_off,
                                                             float* __restrict__ segbox, double* __restrict__ seg_sums, unsigned long long* __restrict__ keysA, int N) {
    bigseg_keys_body(data, seg_points, seg_off, segbox, seg_sums, keysA, N, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(kPieceBlock) void k_bigseg_scatter(const int32_t* __restrict__ seg_off, unsigned long long* __restrict__ keysA,
                                                                unsigned long long* __restrict__ keysB, int N) {
    bigseg_scatter_body(seg_off, keysA, keysB, N, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(kBigBlock) void k_bigseg_runs(const int32_t* __restrict__ seg_off, int S, int N, const unsigned long long* __restrict__ keysA,
                                                           const unsigned long long* __restrict__ keysB, int32_t* __restrict__ sperm) {
    bigseg_runs_body(seg_off, S, N, keysA, keysB, sperm, blockIdx.x);
}
__global__ __launch_bounds__(512) void k_bigseg_boxes(const float* __restrict__ data, const int32_t* __restrict__ seg_points, const int32_t* __restrict__ seg_off,
                                                      int S, int N, const int32_t* __restrict__ seg_chunk_off, const int32_t* __restrict__ sperm,
                                                      float* __restrict__ chunk_box) {
    bigseg_boxes_body(data, seg_points, seg_off, S, N, seg_chunk_off, sperm, chunk_box, blockIdx.x);
}
// batched: grid (big segment of the host's list x piece, scene)
__global__ __launch_bounds__(kPieceBlock) void k_bigseg_box_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    const int b = blockIdx.x / kBigPieces;
    if (b >= c.n_big) return;                                    // the host's list of segments beyond kSortCap points
    bigseg_box_body(c.data, c.seg_points, c.seg_off, c.sort_keys, c.N, c.big_segs[b], blockIdx.x % kBigPieces);
}
__global__ __launch_bounds__(kPieceBlock) void k_bigseg_keys_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    const int b = blockIdx.x / kBigPieces;
    if (b >= c.n_big) return;
    bigseg_keys_body(c.data, c.seg_points, c.seg_off, c.segbox, c.seg_sums, c.sort_keys, c.N, c.big_segs[b], blockIdx.x % kBigPieces);
}
__global__ __launch_bounds__(kPieceBlock) void k_bigseg_scatter_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    const int b = blockIdx.x / kBigPieces;
    if (b >= c.n_big) return;
    bigseg_scatter_body(c.seg_off, c.sort_keys, c.sort_keys + c.N, c.N, c.big_segs[b], blockIdx.x % kBigPieces);
}
__global__ __launch_bounds__(kBigBlock) void k_bigseg_runs_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    bigseg_runs_body(c.seg_off, c.S, c.N, c.sort_keys, c.sort_keys + c.N, c.sperm, blockIdx.x);
}
__global__ __launch_bounds__(512) void k_bigseg_boxes_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    bigseg_boxes_body(c.data, c.seg_points, c.seg_off, c.S, c.N, c.seg_chunk_off, c.sperm, c.chunk_box, blockIdx.x);
}

// coordinate sums of every segment (the library-sort fallback of sg_segment_sort_boxes; same values as the fused kernel's)
__global__ __launch_bounds__(256) void k_segment_sums(const float* __restrict__ data, const int32_t* __restrict__ seg_points,
                                                      const int32_t* __restrict__ seg_off, double* __restrict__ seg_sums) {
    __shared__ double dred[4][3];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lo = seg_off[s], n = seg_off[s + 1] - lo;
    double sm[3] = {0.0, 0.0, 0.0};
    for (int i = tid; i < n; i += 256) {
        const float* r = data + (size_t)seg_points[lo + i] * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) sm[k] += (double)r[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) sm[k] += __shfl_xor(sm[k], o);
    if (lane == 0) { dred[wave][0] = sm[0]; dred[wave][1] = sm[1]; dred[wave][2] = sm[2]; }
    __syncthreads();
    if (tid < 3) seg_sums[(size_t)s * 3 + tid] = ((dred[0][tid] + dred[1][tid]) + dred[2][tid]) + dred[3][tid];
}

// Everything a layer needs laid out per point, in ONE launch (block i = i-th segment in member order): the member arrays
// of sg_gather_members, the centred rows of sg_center_clusters (the cluster centroids come in, the host sums the
// per-segment coordinate sums of the sort kernel) and the sorted kNN operands of sg_knn_operands.
__device__ __forceinline__ void layer_layout_body(const float* __restrict__ data, const int32_t* __restrict__ seg_points, const int32_t* __restrict__ seg_off,
                                                  const int32_t* __restrict__ sperm, const int32_t* __restrict__ order, const int32_t* __restrict__ dst,
                                                  const int32_t* __restrict__ cl, const float* __restrict__ cl_mean, int32_t* __restrict__ members,
                                                  int32_t* __restrict__ pos_of_point, int32_t* __restrict__ cluster_of_pos, int32_t* __restrict__ slot_of_pos,
                                                  float* __restrict__ x9m, float4* __restrict__ sxyzw, int32_t* __restrict__ smpos,
                                                  float4* __restrict__ point_rec, int32_t* __restrict__ seed_id, unsigned int* __restrict__ range_bits,
                                                  int i, int r_begin = 0, int r_end = 0x7fffffff) {
    const int s = order[i];
    const int lo = seg_off[s], n = min(seg_off[s + 1] - lo, r_end), d = dst[i], c = cl[i];
    const float mx = cl_mean[3 * c], my = cl_mean[3 * c + 1], mz = cl_mean[3 * c + 2];
    float amax = 0.f;                                           // largest |centred xyz| / |feature| of the rows this thread lays out
    for (int r = r_begin + threadIdx.x; r < n; r += blockDim.x) {
        const int p = seg_points[lo + r];
        members[d + r] = p;
        if (pos_of_point) pos_of_point[p] = d + r;                // (the engine passes null: its kernels read point_rec instead)
        cluster_of_pos[d + r] = c;
        slot_of_pos[d + r] = i;
        const float* row = data + (size_t)p * 6;
        const float x = row[0], y = row[1], z = row[2];
        // XYZ + this layer's member position of the point in ONE 16-byte record: what the seeded kNN of the layer gathers per seed
        // (indexed by point id without seed ids -- the records then scatter over the whole array)
        if (point_rec && !seed_id) point_rec[p] = make_float4(x, y, z, __int_as_float(d + r));
        float4* o = reinterpret_cast<float4*>(x9m + (size_t)(d + r) * 12);
        o[0] = make_float4(x, y, z, row[3]);
        o[1] = make_float4(row[4], row[5], x - mx, y - my);
        o[2] = make_float4(z - mz, 0.f, 0.f, 0.f);
        amax = fmaxf(fmaxf(fmaxf(amax, fabsf(x - mx)), fmaxf(fabsf(y - my), fabsf(z - mz))), fmaxf(fmaxf(fabsf(row[3]), fabsf(row[4])), fabsf(row[5])));
        const int ci = sperm[lo + r];                          // r-th point of the segment in Morton order
        const float* q = data + (size_t)seg_points[ci] * 6;
        sxyzw[d + r] = make_float4(q[0], q[1], q[2], (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);     // torch.sum(x**2, dim=1)
        smpos[d + r] = d + (ci - lo);
        // Seed ids (round 3): the id of a point in the seed tables = its position in the Morton-sorted CSR of the over-segmentation,
        // lo + r -- the same in every layer, consecutive for the queries of a tile and close for points that are close in space.  The
        // records are then written in order (coalesced; by point id they scattered 16 bytes per point) and a query's 20 seed gathers
        // hit a few neighbouring lines (by point id: 51 MB fetched per scene for 14 MB of kNN operands).
        if (seed_id) {
            seed_id[d + (ci - lo)] = lo + r;
            if (point_rec) point_rec[lo + r] = make_float4(q[0], q[1], q[2], __int_as_float(d + (ci - lo)));
        }
    }
    // EdgeConv's range (kernels_edgeconv.hip, conv1 on fp16 pieces): every difference x_j - x_i inside a cluster is bounded by twice the
    // largest |centred coordinate| / |feature| of the layer (raw and centred XYZ differ by the cluster's mean: the same differences).
    // Non-negative floats order like their bits: one atomic max per wave.
    if (range_bits) {
        amax = sgw::wave_max(amax);
        // 48,000 waves per launch: the range is kept as sg::kRangeWords words (their maximum counts) so that the atomics spread over
        // as many addresses -- one word per scene serialised them in L2 (layout 57 -> 134 us per launch of 8)
        if ((threadIdx.x & 63) == 0 && amax > 0.f)
            atomicMax(range_bits + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (sg::kRangeWords - 1)), __float_as_uint(amax));
    }
}
__global__ void k_layer_layout(const float* __restrict__ data, const int32_t* __restrict__ seg_points, const int32_t* __restrict__ seg_off,
                               const int32_t* __restrict__ sperm, const int32_t* __restrict__ order, const int32_t* __restrict__ dst,
                               const int32_t* __restrict__ cl, const float* __restrict__ cl_mean, int32_t* __restrict__ members,
                               int32_t* __restrict__ pos_of_point, int32_t* __restrict__ cluster_of_pos, int32_t* __restrict__ slot_of_pos,
                               float* __restrict__ x9m, float4* __restrict__ sxyzw, int32_t* __restrict__ smpos, float4* __restrict__ point_rec,
                               int32_t* __restrict__ seed_id, unsigned int* __restrict__ range_bits) {
    layer_layout_body(data, seg_points, seg_off, sperm, order, dst, cl, cl_mean, members, pos_of_point, cluster_of_pos, slot_of_pos, x9m, sxyzw,
                      smpos, point_rec, seed_id, range_bits, blockIdx.x);
}
// The engine's launches: a block lays out the first kLayoutPiece rows of its segment; the rest of a larger segment (ScanNet floors
// and walls: 10k-40k points) is cut into pieces of kLayoutPiece rows that the host lists per layer (lay_big = (slot, first row)
// pairs) and a second launch spreads over the GPU -- one block walking 40k rows alone was 0.46 ms per launch.
__global__ void k_layer_layout_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x >= c.S) return;
    layer_layout_body(c.data, c.seg_points, c.seg_off, c.sperm, c.order, c.dst, c.cl, c.cl_mean, c.members, nullptr, c.cluster_of_pos,
                      c.slot_of_pos, c.x9m, c.sxyzw, c.smpos, c.point_rec, c.seed_id, c.ec_range, blockIdx.x, 0, sg::kLayoutPiece);
}
__global__ void k_layer_layout_big_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x >= c.lay_nbig) return;
    const int i = c.lay_big[2 * blockIdx.x], r0 = c.lay_big[2 * blockIdx.x + 1];
    layer_layout_body(c.data, c.seg_points, c.seg_off, c.sperm, c.order, c.dst, c.cl, c.cl_mean, c.members, nullptr, c.cluster_of_pos,
                      c.slot_of_pos, c.x9m, c.sxyzw, c.smpos, c.point_rec, c.seed_id, c.ec_range, i, r0, r0 + sg::kLayoutPiece);
}

// per layer: block i = i-th segment in member order; writes the operand and the member position in SORTED order
__global__ void k_knn_operands(const float* __restrict__ data, const int32_t* __restrict__ seg_points,
                               const int32_t* __restrict__ seg_off, const int32_t* __restrict__ sperm,
                               const int32_t* __restrict__ order, const int32_t* __restrict__ dst,
                               float4* __restrict__ sxyzw, int32_t* __restrict__ smpos) {
    const int i = blockIdx.x;
    const int s = order[i];
    const int lo = seg_off[s], n = seg_off[s + 1] - lo, d = dst[i];
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        const int ci = sperm[lo + r];
        const float* row = data + (size_t)seg_points[ci] * 6;
        const float x = row[0], y = row[1], z = row[2];
        sxyzw[d + r] = make_float4(x, y, z, (x * x + y * y) + z * z);     // torch.sum(x**2, dim=1)
        smpos[d + r] = d + (ci - lo);
    }
}

// profiling aid (SG_KNN_DEBUG & 16): [0] blocks, [1] cycles phase A, [2] merge A, [3] phase B, [4] final merge+write,
// [5] chunks scanned (wave level), [6] chunks tested, [7] segments tested, [8] lane appends, [9] drain iterations,
// [10] drains, [11] drains with nothing buffered, [12] drains merged twelve at a time
__device__ unsigned long long g_knn5_stats[16];
#ifdef SG_KNN_SELFCHECK
// debugging aid (make SELFCHECK=1): [0] seeded lists that were not in order right after seeding, [1] lists not in order at the output stage,
// [2] tiles checked; [8..] one example: the lane's 20 seed ids, record positions and score bits
__device__ unsigned long long g_knn_check[8 + 128];
#endif

// kSlices = waves per 64-query tile (1, 2 or 4): the cluster's candidate chunks are dealt round-robin to them.  More
// slices shorten a tile's critical path but every slice warms up its own top-K list: at 150k points one tile costs
// 124 list insertions per query with 4 slices and about half of that with 1, and a launch with >= 2048 tiles fills
// the GPU without slicing (bench: 584 -> 670 scenes/s at 4 -> 1 slices; the host picks by tile count).
//
// kSeeded (one wave per tile only): the clusters of this layer are unions of the clusters the PREVIOUS kNN layer ran
// on, the score of a pair depends on the two points' raw coordinates only, and union() appends whole member lists
// (model.py:191), so the relative member order -- the tie rule -- inside a former cluster is unchanged.  A query's
// previous list therefore IS the exact top K among the candidates of its former cluster: the kernel starts from it
// (scores recomputed by the same formula, indices mapped to this layer's member positions), with an already tight
// threshold, and skips every chunk whose segment belongs to the query's former cluster.  Former clusters of <= K
// points have no kNN list (model.py:516-518 pads them): their segments carry seg_prevcl = -1, their queries start empty
// and nobody skips them.
template <int K, int kSlices, bool kSeeded>
__device__ __forceinline__ void cluster_knn_sorted_body(
    sg::gptr<const float4> sxyzw, sg::gptr<const int32_t> smpos, sg::gptr<const int32_t> cl_off,
    sg::gptr<const int32_t> tile_cl, sg::gptr<const int32_t> tile_lo, sg::gptr<const int32_t> tile_hi,
    sg::gptr<const int32_t> cl_seg_off, sg::gptr<const int32_t> order, sg::gptr<const int32_t> dst,
    sg::gptr<const int32_t> seg_off, sg::gptr<const int32_t> seg_chunk_off, sg::gptr<const float> segbox,
    sg::gptr<const float> chunk_box, sg::gptr<const int32_t> slot_of_pos, int pos0, sg::gptr<int32_t> knn, int dbg_arg,
    sg::gptr<const int32_t> seed, sg::gptr<const int32_t> seg_prevcl, sg::gptr<const int32_t> members,
    sg::gptr<const float4> point_rec, int t, sg::gptr<int32_t> seed_out = nullptr) {
    // seed_out (one wave per tile): the table once more with rows and entries as POINT ids -- what the next layer's seeded launch
    // reads (a separate pass over the finished table re-read 12 MB per scene for it)
    static_assert(!kSeeded || kSlices == 1, "seeding is built for one wave per tile");
    const int dbg = kKnnProfile ? dbg_arg : 0;
    // LDS per wave decides how many tiles a CU keeps in flight, and this kernel waits on memory ~45 % of the time:
    // one wave per tile needs neither the merge area (>= K slots per lane) nor a 128-entry descriptor batch
    constexpr int kBufS = kSlices == 1 ? 12 : 20;
    constexpr int kSlotBatch = kSlices == 1 ? 32 : 128;
    static_assert(kSlices == 1 || kBufS >= K, "the merge area aliases the append buffers");
    __shared__ float4 slab[kSlices][kChunkPts + kQuadS];
    __shared__ int slab_i[kSlices][kChunkPts + kQuadS];
    __shared__ unsigned long long buf[kBufS][64 * kSlices];     // append buffers; later lists[slice][K][64]
    __shared__ unsigned int thr_pub[kSlices][64];             // score part of each slice's 20th best
    __shared__ unsigned int thr5_pub[kSlices][64];            // score part of each slice's (K / kSlices)-th best
    __shared__ __attribute__((aligned(16))) float chunkbox_lds[kSlices][kSlices == 1 ? 256 : 64];      // one wave per tile: 32 boxes at a time
    __shared__ int st_m[kSlotBatch], st_c0[kSlotBatch], st_d[kSlotBatch], st_pc[kSlotBatch];
    __shared__ __attribute__((aligned(16))) float st_box[kSlotBatch][8];
    const int c = tile_cl[t];
    const int clo = cl_off[c], n = cl_off[c + 1] - clo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = tile_lo[t] + lane;                          // SORTED position
    const bool active = q < tile_hi[t];
    const int myrow = active ? smpos[q] : 0;                  // member position = output row
    if (n <= K) {                                            // model.py:516-518 (block-uniform)
        if (active && wave == 0) {
            // (its own look-up of the row: the compiler lays this block out behind the main path and, sharing `myrow` with it, kept the
            // register alive -- in scratch -- through both phases; see the output stage below)
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int row_e = smpos[tile_lo[t] + lane_e];
            const sg::gptr<int32_t> o = knn + (size_t)row_e * K;
#pragma unroll
            for (int j = 0; j < K; ++j) o[j] = j < n ? clo + j : pos0;
            if (seed_out) {
                const sg::gptr<int32_t> so = seed_out + (size_t)members[row_e] * K;
#pragma unroll
                for (int j = 0; j < K; ++j) so[j] = members[j < n ? clo + j : pos0];
            }
        }
        return;
    }
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) me = sxyzw[q];
    // the running top K in LIST form (knn_device.h): doubles whose order is the keys' order, inserted with v_min_f64 / v_max_f64
    double kv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) kv[j] = list_empty();
    // Round 5: thresholds and candidates live in LIST form throughout (knn_device.h: list_key builds a candidate from score and index in the
    // instructions make_key + to_list took together; list_score gives a value's score bits back in one instruction).  The drain used to convert
    // its twelve buffer entries per lane, five instructions each, wave-wide and whatever the lanes held -- a fifth of a drain.
    // thr_l = the lane's 20th best (empty: everything is accepted); idle lanes hold the largest list value and never accept
    double thr_l = active ? list_empty() : to_list(~0ull);
    int myprev = -1;                                          // former cluster whose candidates are already in kv
    if (kSeeded && active) {
        myprev = seg_prevcl[order[slot_of_pos[myrow]]];
        if (myprev >= 0) {
            const sg::gptr<const int32_t> sp = seed + (size_t)members[myrow] * K;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const float4 rec = point_rec[sp[j]];              // XYZ + this layer's member position: one gather per seed
                kv[j] = to_list(make_key(score4(me, make_float4(rec.x, rec.y, rec.z, (rec.x * rec.x + rec.y * rec.y) + rec.z * rec.z)),
                                         __float_as_int(rec.w) - clo));
            }
            thr_l = kv[K - 1];
#ifdef SG_KNN_SELFCHECK
            bool bad_ = false;
#pragma unroll
            for (int j = 0; j + 1 < K; ++j) bad_ |= from_list(kv[j]) <= from_list(kv[j + 1]);
            if (bad_ && atomicAdd(&g_knn_check[0], 1ull) == 0ull) {
                for (int j = 0; j < K; ++j) {
                    const float4 rec = point_rec[sp[j]];
                    g_knn_check[8 + j] = ((unsigned long long)(unsigned)sp[j] << 32) | (unsigned)(__float_as_int(rec.w) - clo);
                    g_knn_check[8 + 20 + j] = from_list(kv[j]);
                    const float again = score4(me, make_float4(rec.x, rec.y, rec.z, (rec.x * rec.x + rec.y * rec.y) + rec.z * rec.z));
                    g_knn_check[8 + 64 + j] = ((unsigned long long)__float_as_uint(again) << 32) | __float_as_uint(rec.x);
                    g_knn_check[8 + 84 + j] = ((unsigned long long)__float_as_uint(rec.y) << 32) | __float_as_uint(rec.z);
                }
                g_knn_check[8 + 44] = ((unsigned long long)__float_as_uint(me.x) << 32) | __float_as_uint(me.y);
                g_knn_check[8 + 45] = ((unsigned long long)__float_as_uint(me.z) << 32) | __float_as_uint(me.w);
                { const float4 m2 = sxyzw[q]; g_knn_check[8 + 46] = ((unsigned long long)__float_as_uint(m2.x) << 32) | __float_as_uint(m2.w); }
                g_knn_check[8 + 40] = ((unsigned long long)(unsigned)myrow << 32) | (unsigned)members[myrow];
                g_knn_check[8 + 41] = ((unsigned long long)(unsigned)clo << 32) | (unsigned)n;
                g_knn_check[8 + 42] = ((unsigned long long)(unsigned)myprev << 32) | (unsigned)t;
            }
#endif
        }
    }
#ifdef SG_KNN_SELFCHECK
    if (kSeeded && lane == 0) atomicAdd(&g_knn_check[2], 1ull);
#endif
    thr_pub[wave][lane] = list_score(thr_l);
    thr5_pub[wave][lane] = 0u;
    int cnt = 0;
    float4* cw = slab[wave];
    int* ci = slab_i[wave];
    __syncthreads();

    // Lower bounds of the query's final 20th-best key that need no merge: (a) any slice's own 20th best, (b) the
    // weakest of the slices' (K / kSlices)-th bests (kSlices x K / kSlices = K candidates are at least that good).  Published score parts
    // only rise, so stale reads are safe; the index part is cleared (ties at the bound are still accepted).
    // In list form: the larger of the lane's own 20th best and list_floor(published score bits) -- a key beats the latter iff its score bits
    // reach the published ones, whatever its index.  One wave per tile publishes nothing its own list does not know.
    auto best_thr = [&]() {
        if constexpr (kSlices == 1) return thr_l;
        unsigned int b = 0u, m5 = 0xffffffffu;
#pragma unroll
        for (int w = 0; w < kSlices; ++w) { b = max(b, thr_pub[w][lane]); m5 = min(m5, thr5_pub[w][lane]); }
        return list_max(list_floor(max(b, m5)), thr_l);
    };
    auto drain = [&]() {
        const int mxc = sgw::wave_max(cnt);                    // DPP + readlane: wave-uniform (an SGPR drives the loop below)
        if (dbg & 32) {
            unsigned long long tot = cnt;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
            if (lane == 0) {
                atomicAdd(&g_knn5_stats[8], tot); atomicAdd(&g_knn5_stats[9], (unsigned long long)mxc);
                atomicAdd(&g_knn5_stats[10], 1ull);                                           // drains, of them: empty, merged twelve at a time
                if (mxc == 0) atomicAdd(&g_knn5_stats[11], 1ull);
                if (kBufS == 12 && K == 20 && mxc > 0) atomicAdd(&g_knn5_stats[12], 1ull);
            }
        }
        if constexpr (kBufS == 12 && K == 20) {
            // One wave per tile: all twelve buffer entries at once (knn_device.h, list_merge12: ~280 instructions whatever the count, against
            // ~50 per key one at a time; 14 of a tile's 15.5 drains have 9-12 keys in the busiest lane, tools/knn_counters.py).  Written as
            // a loop of at most one trip whose count the compiler cannot see, and with no one-at-a-time alternative beside it: the network
            // leaves the list in other registers than it found it in, and as straight-line code (or with a second way to update the list)
            // the register allocator paid for that with 20 64-bit moves on the path that does NOT drain -- once per four candidates, more
            // than the merge saves.  A loop carries the list in place and keeps the moves on its own back edge.
            int trips = __builtin_amdgcn_readfirstlane(mxc > 0 ? 1 : 0);
            asm volatile("" : "+s"(trips));
            for (; trips > 0; --trips) {
                double b[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const double k = __longlong_as_double((long long)buf[u][tid]);
                    b[u] = u < cnt ? k : list_empty();
                }
                list_merge12(kv, b);
            }
        } else {
            // the next buffered key is read while the current one is inserted (an insertion is ~50 VALU, an LDS read ~100 cycles)
            double nxt = __longlong_as_double((long long)buf[0][tid]);
            for (int u = 0; u < mxc; ++u) {
                const double cur = u < cnt ? nxt : list_empty();
                nxt = __longlong_as_double((long long)buf[min(u + 1, kBufS - 1)][tid]);
                list_insert_l<K>(kv, cur);
            }
        }
        cnt = 0;
        if (active) {
            thr_l = kv[K - 1];
            thr_pub[wave][lane] = list_score(thr_l);
            thr5_pub[wave][lane] = list_score(kv[K / kSlices - 1]);
        }
    };
    bool ok = true;                                           // seeded: false while the segment at hand belongs to my former cluster
    int dbg_scanned = 0;                                      // profiling builds: chunks this tile scanned
    // the fp32 score a candidate must reach to be worth a key: the score part of the larger threshold (key 0 = nothing yet = -inf;
    // an idle lane's all-ones key decodes to NaN, which no score reaches; +inf while the segment at hand is already covered)
    // Score bits BELOW -inf's (0x007fffff) are not scores: 0 is the empty list, and with several waves per tile a slice that holds fewer than 20
    // real candidates publishes the score of a slab's padding candidate (-inf), whose list_floor() carries score bits 0x007ffffe -- decoded, a NaN,
    // which no candidate's score reaches: the lane accepted nothing more and its padding entries reached the table (round 5: scenes of 3,000
    // points came out wrong; tests/test_gpu_ops.py::test_multi_wave_knn_with_slices_short_of_candidates).  All of them mean "anything goes".
    auto score_bound = [](double bound_l, bool open) {
        const unsigned int o = list_score(bound_l);
#ifdef SG_KNN_R5_NAN_BOUND          // the round-5 fault, kept buildable: tools/build_knn_gate_lib.sh makes a library with it, and the gates above must fail on it
        const float f = o == 0u ? -INFINITY : __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
#else
        const float f = o < 0x00800000u ? -INFINITY : __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
#endif
        return open ? f : INFINITY;
    };
    // one 32-point chunk: sorted positions [p0, p0 + m)
    // one 32-point chunk: sorted positions [p0, p0 + m)
    auto scan_chunk = [&](const float* bx, int p0, int m) {
        const double use_l = best_thr();
        if ((dbg & 32) && lane == 0) atomicAdd(&g_knn5_stats[6], 1ull);
        if (!__any(ok && list_key(box_score_bound(me, bx), 0) >= use_l)) return;
        if ((dbg & 32) && lane == 0) atomicAdd(&g_knn5_stats[5], 1ull);
        if (dbg & 16) ++dbg_scanned;
        __builtin_amdgcn_wave_barrier();
        if (lane < kChunkPts + kQuadS) {
            const bool in = lane < m;
            float4 cand = make_float4(0.f, 0.f, 0.f, INFINITY);
            if (in) cand = sxyzw[p0 + lane];
            cw[lane] = cand;
            ci[lane] = in ? smpos[p0 + lane] - clo : 0x7fffffff;
        }
        __builtin_amdgcn_wave_barrier();
        // Most candidates lose against the running 20th best: they are turned away on the fp32 SCORE alone (one compare instead of
        // building the 64-bit key and comparing it twice: 7 instead of 13 VALU per candidate and lane); a score equal to the bound
        // goes on to the exact key comparison, which also settles the index order of ties.  `fbound` is refreshed wherever the
        // thresholds move (chunk start, drain).
        float fbound = score_bound(list_max(use_l, thr_l), ok);
        for (int i = 0; i < m; i += kQuadS) {
            // all kQuadS operand reads first: behind the first conditional store the scheduler would issue them one candidate
            // at a time and every candidate would wait out its own LDS round trip (with the reads together the compiler also
            // pairs the candidates' arithmetic into v_pk_mul / v_pk_fma_f32: same IEEE operations, fewer issue slots; a slab laid
            // out in candidate pairs to drop the packing moves measured no faster)
            float4 c4[kQuadS];
#pragma unroll
            for (int u = 0; u < kQuadS; ++u) c4[u] = cw[i + u];
            // all kQuadS scores first, ONE branch for the group: late in a scan every lane turns every candidate away, and a taken
            // `s_cbranch_execz` per candidate (7 VALU of work between two of them) costs about as much as the work itself
            float sc4[kQuadS];
#pragma unroll
            for (int u = 0; u < kQuadS; ++u) sc4[u] = score4(me, c4[u]);
            static_assert(kQuadS == 4, "the group test below is written for four candidates");
            if (fmaxf(fmaxf(sc4[0], sc4[1]), fmaxf(sc4[2], sc4[3])) >= fbound) {
#pragma unroll
                for (int u = 0; u < kQuadS; ++u) {
                    if (sc4[u] >= fbound) {
                        const double key = list_key(sc4[u], ci[i + u]);                   // the index is only read for a survivor
                        if (key > use_l && key > thr_l) {
                            buf[cnt][tid] = (unsigned long long)__double_as_longlong(key);
                            ++cnt;
                        }
                    }
                }
            }
            if (__any(cnt > kBufS - kQuadS)) {
                drain();
                fbound = score_bound(list_max(use_l, thr_l), ok);
            }
        }
    };
    const int so0 = cl_seg_off[c], nslots = cl_seg_off[c + 1] - so0;
    const int own = slot_of_pos[active ? myrow : smpos[tile_lo[t]]] - so0;   // lanes of a tile may span segments: lane 0 decides
    const int own0 = __shfl(own, 0);
    auto slot_of = [&](int r) {
        int sl = own0 + r;
        if (sl >= nslots) sl -= nslots;
        return so0 + sl;
    };
    // all chunks of one segment (descriptor already in registers / LDS); chunk number `item` decides which wave
    // takes it.  The segment's chunk boxes are staged 8 at a time through a wave-private LDS strip (one coalesced
    // load instead of one dependent global load per chunk).
    float* cbx = &chunkbox_lds[wave][0];
    // Large segments (a floor of 40k points is 1,300 chunks): testing every chunk box lane-uniformly costs ~25 instructions
    // per chunk and tile.  One wave per tile: 64 chunk boxes are tested AT ONCE, one per lane, against the box of the tile's
    // 64 queries and the weakest threshold any lane still accepts (the same conservative bound as box_score_bound, taken for
    // the whole tile); only the survivors go through the exact per-query test.
    float qlo[3] = {0.f, 0.f, 0.f}, qhi[3] = {0.f, 0.f, 0.f}, qw = 0.f;
    if (kSlices == 1) {
        const float4 m0 = active ? me : make_float4(sgw::bcast(me.x, 0), sgw::bcast(me.y, 0), sgw::bcast(me.z, 0), sgw::bcast(me.w, 0));
        qlo[0] = sgw::wave_min(m0.x); qlo[1] = sgw::wave_min(m0.y); qlo[2] = sgw::wave_min(m0.z);
        qhi[0] = sgw::wave_max(m0.x); qhi[1] = sgw::wave_max(m0.y); qhi[2] = sgw::wave_max(m0.z);
        qw = sgw::wave_max(m0.w);
    }
    // `start` (one wave per tile only): the chunk to begin with -- the tile's own chunk in its own segment, so that the first
    // 64 candidates are the queries' immediate neighbours and every later chunk meets a tight threshold.  (Walking a 1,300-chunk
    // floor from chunk 0 approaches the tile along the space-filling curve: nearly every chunk on the way is a little closer
    // than the best so far, passes its box test and is scanned.)
    // `two_sided`: the ring is walked OUTWARDS from `start` in both directions (start, start - 1, start + 1, start - 2, ...).  On a Z curve a
    // query's spatial neighbours lie before AND behind it; walking forwards only, a lane whose neighbours are behind `start` keeps a loose
    // threshold until the walk has come all the way round, and every chunk inside that loose radius is scanned for it (round 5: the slowest
    // tile of a ScanNet-shaped scene, in a 29k-point floor, scanned 215 chunks where ~30 hold its neighbours -- and a launch is as long as
    // its slowest tile).  The visiting order cannot change the result (exact top K under a total order).
    auto ring_chunk = [](int start, int e, int nch, bool two_sided) {
        int ch = start + (two_sided ? ((e & 1) ? -((e + 1) >> 1) : (e >> 1)) : e);
        if (ch >= nch) ch -= nch;
        if (ch < 0) ch += nch;
        return ch;
    };
    auto scan_segment = [&](int sg_m, int sg_c0, int d, const float* sbox, int pc, int& item, int start, bool two_sided = false) {
        const int nch = (sg_m + kChunkPts - 1) / kChunkPts;
        const double use_l = best_thr();
        if (kSeeded) ok = pc < 0 || pc != myprev;
        if ((dbg & 32) && lane == 0 && wave == 0) atomicAdd(&g_knn5_stats[7], 1ull);
        if (!__any(ok && list_key(box_score_bound(me, sbox), 0) >= use_l)) { item += nch; return; }
        if constexpr (kSlices == 1) {
            // 32 chunk boxes per coalesced load (one per lane), staged in the wave's LDS strip; ONE call site of scan_chunk
            for (int j0 = 0; j0 < nch; j0 += 32) {
                const int nb = min(32, nch - j0);
                const int mych = ring_chunk(start, min(j0 + lane, nch - 1), nch, two_sided);       // ring order from `start`
                bool pass = lane < nb;
                // the weakest score any lane still accepts (thresholds only rise: refreshed per 32 chunks; all lanes take part)
                float weakest = -INFINITY;
                if (nch > 16) {
                    const unsigned int o = list_score(best_thr());
                    float mine = INFINITY;
                    if (active && ok) mine = o < 0x00800000u ? -INFINITY : __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
                    weakest = sgw::wave_min(mine);
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < nb) {
                    const sg::gptr<const float4> bp = (sg::gptr<const float4>)(chunk_box + (size_t)(sg_c0 + mych) * 8);
                    const float4 b0 = bp[0], b1 = bp[1];
                    reinterpret_cast<float4*>(cbx)[2 * lane] = b0;
                    reinterpret_cast<float4*>(cbx)[2 * lane + 1] = b1;
                    if (nch > 16) {
                        const float gx = fmaxf(fmaxf(b0.x - qhi[0], qlo[0] - b0.w), 0.f);
                        const float gy = fmaxf(fmaxf(b0.y - qhi[1], qlo[1] - b1.x), 0.f);
                        const float gz = fmaxf(fmaxf(b0.z - qhi[2], qlo[2] - b1.y), 0.f);
                        pass = -((gx * gx + gy * gy) + gz * gz) * 0.999999f + 9.6e-7f * (qw + b1.z) >= weakest;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                unsigned long long live = __ballot(pass);
                while (live) {
                    const int j = __ffsll((unsigned long long)live) - 1;
                    live &= live - 1;
                    const int ch = ring_chunk(start, j0 + j, nch, two_sided);
                    scan_chunk(cbx + j * 8, d + ch * kChunkPts, min(kChunkPts, sg_m - ch * kChunkPts));
                }
            }
            item += nch;
            return;
        }
        for (int j0 = 0; j0 < nch; j0 += 8) {
            const int nb = min(8, nch - j0);
            __builtin_amdgcn_wave_barrier();
            if (lane < nb * 8) cbx[lane] = chunk_box[(size_t)(sg_c0 + j0) * 8 + lane];
            __builtin_amdgcn_wave_barrier();
            for (int j = 0; j < nb; ++j, ++item)
                if ((item & (kSlices - 1)) == wave)
                    scan_chunk(cbx + j * 8, d + (j0 + j) * kChunkPts, min(kChunkPts, sg_m - (j0 + j) * kChunkPts));
        }
    };

    int item = 0;
    const unsigned long long t0 = (dbg & 16) ? __builtin_readcyclecounter() : 0ull;
    // phase A: the queries' own segment, its chunks dealt to the four waves; afterwards best_thr() already is a tight
    // bound (weakest of the four 5th bests) without merging the four partial lists
    {
        const int slot = slot_of(0), sg = order[slot];
        float sbox[8];
#pragma unroll
        for (int k = 0; k < 7; ++k) sbox[k] = segbox[(size_t)sg * 8 + k];
        const int own_chunk = max(0, min((tile_lo[t] - dst[slot]) / kChunkPts, (seg_off[sg + 1] - seg_off[sg] - 1) / kChunkPts));
        scan_segment(seg_off[sg + 1] - seg_off[sg], seg_chunk_off[sg], dst[slot], sbox, kSeeded ? seg_prevcl[sg] : -1, item, own_chunk, kSlices == 1);
    }
    drain();
    const unsigned long long t1 = (dbg & 16) ? __builtin_readcyclecounter() : 0ull;
    __syncthreads();                                          // every slice has published its 5th / 20th best of the own segment
    const unsigned long long t2 = (dbg & 16) ? __builtin_readcyclecounter() : 0ull;
    // phase B: every other segment of the cluster.  The segment descriptors (id, member offset, size, first chunk,
    // box) are staged kSlotBatch at a time in LDS by the whole workgroup: walking the list straight from global
    // memory costs two dependent round trips per segment and dominated the kernel for clusters of many segments.
    for (int r0 = 1; r0 < nslots; r0 += kSlotBatch) {
        const int nb = min(kSlotBatch, nslots - r0);
        __syncthreads();                                      // previous batch fully consumed
        for (int e = tid; e < nb; e += 64 * kSlices) {
            const int slot = slot_of(r0 + e), sg = order[slot];
            st_m[e] = seg_off[sg + 1] - seg_off[sg];
            st_c0[e] = seg_chunk_off[sg];
            st_d[e] = dst[slot];
            if (kSeeded) st_pc[e] = seg_prevcl[sg];
            const sg::gptr<const float4> bp = (sg::gptr<const float4>)(segbox + (size_t)sg * 8);
            reinterpret_cast<float4*>(&st_box[e][0])[0] = bp[0];
            reinterpret_cast<float4*>(&st_box[e][0])[1] = bp[1];
        }
        __syncthreads();
        for (int i = 0; i < nb; ++i) scan_segment(st_m[i], st_c0[i], st_d[i], &st_box[i][0], kSeeded ? st_pc[i] : -1, item, 0);
    }
    drain();
    const unsigned long long t3 = (dbg & 16) ? __builtin_readcyclecounter() : 0ull;
#ifdef SG_KNN_SELFCHECK
    if (kSeeded && active) {
        bool bad_ = false;
#pragma unroll
        for (int j = 0; j + 1 < K; ++j) bad_ |= from_list(kv[j]) <= from_list(kv[j + 1]);
        if (bad_) atomicAdd(&g_knn_check[1], 1ull);
    }
#endif
    if (kSlices == 1) {
        if (active) {
            // the output row is looked up AGAIN (from a laundered lane number: not the load at the top): kept from there it was the one
            // register too many of the seeded kernel at 128 VGPRs and lived in scratch across both phases (tests/test_build.py keeps the
            // kernels of the inference path free of scratch)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int row_o = smpos[tile_lo[t] + lane_o];
            const sg::gptr<int32_t> o = knn + (size_t)row_o * K;
#pragma unroll
            for (int j = 0; j < K; ++j) o[j] = clo + list_index(kv[j]);
            if (seed_out) {
                const sg::gptr<int32_t> so = seed_out + (size_t)members[row_o] * K;
#pragma unroll
                for (int j = 0; j < K; ++j) so[j] = members[clo + list_index(kv[j])];
            }
        }
    } else {
        __syncthreads();                                          // every wave is done with its append buffer
        unsigned long long* lists = &buf[0][0];
    #pragma unroll
        for (int j = 0; j < K; ++j) lists[((size_t)wave * K + j) * 64 + lane] = from_list(kv[j]);
        __syncthreads();
        if (wave == 0 && active) {
            int p[kSlices];
            unsigned long long h[kSlices];
    #pragma unroll
            for (int w = 0; w < kSlices; ++w) { p[w] = 0; h[w] = lists[((size_t)w * K) * 64 + lane]; }
            const sg::gptr<int32_t> o = knn + (size_t)myrow * K;
            for (int j = 0; j < K; ++j) {
                int bw = 0;
                unsigned long long bk = h[0];
    #pragma unroll
                for (int w = 1; w < kSlices; ++w) if (h[w] > bk) { bk = h[w]; bw = w; }
                o[j] = clo + key_index(bk);
    #pragma unroll
                for (int w = 0; w < kSlices; ++w)
                    if (w == bw) { ++p[w]; h[w] = p[w] < K ? lists[((size_t)w * K + p[w]) * 64 + lane] : 0ull; }
            }
        }
    }
    if ((dbg & 16) && lane == 0) {
        const unsigned long long t4 = __builtin_readcyclecounter();
        if (wave == 0) atomicAdd(&g_knn5_stats[0], 1ull);
        atomicAdd(&g_knn5_stats[1], t1 - t0);
        atomicAdd(&g_knn5_stats[2], t2 - t1);
        atomicAdd(&g_knn5_stats[3], t3 - t2);
        atomicAdd(&g_knn5_stats[4], t4 - t3);
        // the slowest tile: cycles (24 bits, saturated) | cluster points / 64 (14) | segments of the cluster (12) | chunks scanned (14)
        atomicMax(&g_knn5_stats[13], (min(t4 - t0, 0xffffffull) << 40) | ((unsigned long long)min(n >> 6, 0x3fff) << 26) |
                                     ((unsigned long long)min(nslots, 0xfff) << 14) | (unsigned long long)min(dbg_scanned, 0x3fff));
        if (n > 2048) { atomicAdd(&g_knn5_stats[14], t4 - t0); atomicAdd(&g_knn5_stats[15], 1ull); }
    }
}
template <int K, int kSlices, bool kSeeded = false>
__global__ __launch_bounds__(64 * kSlices, kSlices == 1 ? 4 : 1) void k_cluster_knn_sorted(
    const float4* __restrict__ sxyzw, const int32_t* __restrict__ smpos, const int32_t* __restrict__ cl_off,
    const int32_t* __restrict__ tile_cl, const int32_t* __restrict__ tile_lo, const int32_t* __restrict__ tile_hi,
    const int32_t* __restrict__ cl_seg_off, const int32_t* __restrict__ order, const int32_t* __restrict__ dst,
    const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_chunk_off, const float* __restrict__ segbox,
    const float* __restrict__ chunk_box, const int32_t* __restrict__ slot_of_pos, int pos0, int32_t* __restrict__ knn, int dbg_arg,
    const int32_t* __restrict__ seed = nullptr, const int32_t* __restrict__ seg_prevcl = nullptr, const int32_t* __restrict__ members = nullptr,
    const float4* __restrict__ point_rec = nullptr) {
    using sg::as_global;
    cluster_knn_sorted_body<K, kSlices, kSeeded>(as_global(sxyzw), as_global(smpos), as_global(cl_off), as_global(tile_cl), as_global(tile_lo),
                                                 as_global(tile_hi), as_global(cl_seg_off), as_global(order), as_global(dst), as_global(seg_off),
                                                 as_global(seg_chunk_off), as_global(segbox), as_global(chunk_box), as_global(slot_of_pos), pos0,
                                                 as_global(knn), dbg_arg, as_global(seed), as_global(seg_prevcl), as_global(members),
                                                 as_global(point_rec), blockIdx.x);
}
// one wave per tile: these waves wait on memory two thirds of their life (PMC: 35 % issuing).  The seeded kernel gets a
// fourth resident wave per SIMD by capping it at 128 VGPRs (129 -> 118, no spill: 952 -> 848 us per launch of 8 scenes); a
// fifth wave for the unseeded one (104 -> 96 VGPRs) costs 10 spilled registers and is slower (973 -> 991 us).
template <int K, int kSlices, bool kSeeded>
__global__ __launch_bounds__(64 * kSlices, kSlices == 1 ? 4 : 1) void k_cluster_knn_sorted_b(const sg::SlotCtx* __restrict__ cx, int write_seed) {
    const sg::SlotCtx& c = cx[blockIdx.x];                    // grid = (scenes, tiles): one scene per XCD (kernels_edgeconv.hip, k_edgeconv_b)
    if ((int)blockIdx.y >= c.T) return;
    // pointers read out of a SlotCtx are generic to the compiler (flat_load: vmcnt AND lgkmcnt); sg_common.h, gptr
    using sg::as_global;
    cluster_knn_sorted_body<K, kSlices, kSeeded>(as_global(c.sxyzw), as_global(c.smpos), as_global(c.cl_pt_off), as_global(c.tile_cl),
                                                 as_global(c.tile_lo), as_global(c.tile_hi), as_global(c.cl_seg_off), as_global(c.order),
                                                 as_global(c.dst), as_global(c.seg_off), as_global(c.seg_chunk_off), as_global(c.segbox),
                                                 as_global(c.chunk_box), as_global(c.slot_of_pos), c.pos0, as_global(c.knn), 0,
                                                 as_global(c.knn_seed), as_global(c.seg_prevcl), as_global(c.seed_id), as_global(c.point_rec),
                                                 blockIdx.y, kSlices == 1 && !kSeeded && write_seed ? as_global(c.knn_seed) : nullptr);
}


// kNN table of one layer (rows and entries = member positions) -> rows and entries = point ids, for seeding the next layer
__global__ void k_knn_seed_points(const int32_t* __restrict__ knn, const int32_t* __restrict__ members, int N, int K,
                                  int32_t* __restrict__ seed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * K) return;
    seed[(size_t)members[i / K] * K + i % K] = members[knn[i]];
}
__global__ void k_knn_seed_points_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.N * 20) return;
    c.knn_seed[(size_t)c.seed_id[i / 20] * 20 + i % 20] = c.seed_id[c.knn[i]];
}

// ---------------------------------------------------------------------------------------------------------------
// Two-pass kernel over a cluster-ordered chunk table, one wave per 64-query tile (sg_knn_set_variant(0); faster than
// the kernel above on large scenes and large segments, on par at 150k / 1.5k -- numbers at sg::knn_variant_for).
// Two measurements of the kernel above shaped it (150k points, one wave per tile):
//   * ~3/4 of its instructions are sorted insertions of 64-bit (score, index) keys -- one compare and four selects
//     per list slot, ~105 instructions per insertion, executed wave-wide for the largest count of any lane (158 per
//     tile for 71 appended keys per lane);
//   * yet a pass that inserts almost nothing costs the same ~180k cycles per tile: every segment that survives its box
//     test costs a dependent global load of its chunk boxes, every surviving chunk another one of its operands -- about
//     25 exposed round trips per tile with 2-3 waves per SIMD to hide them.
// Hence
//   * k_knn_chunk_table lays the chunks of every cluster out contiguously per layer: {box min, box max, max |p|^2,
//     first sorted position << 6 | points - 1}.  A lane tests one chunk per 32-byte coalesced load against the
//     tile's query box and weakest threshold, one ballot names the survivors of 64 chunks, each survivor's descriptor is
//     broadcast with v_readlane for the exact per-query test, and the operands of the NEXT survivor are already on
//     their way to registers while the current chunk is scanned out of LDS;
//   * pass 1 keeps only the K best SCORES per query: inserting into a descending list of floats is one v_med3_f32 per
//     slot, new[j] = med3(old[j-1], old[j], x).  Its result s_K is the exact K-th best score;
//   * pass 2 rescans with the fixed floor s_K: only candidates with score >= s_K are appended (K of them plus ties at
//     the floor) and only those go through the exact 64-bit insertion; most chunks fail the box test at once.
// The key order refines the float order (ties: lower member index wins; -0.0 < +0.0), so the K best keys all have
// score >= s_K under float comparison and pass 2 sees every one of them: the table is bit-identical to the kernels above.
// ---------------------------------------------------------------------------------------------------------------
template <int K>
__device__ inline void score_insert(float (&sv)[K], float x) {
#pragma unroll
    for (int j = K - 1; j > 0; --j) sv[j] = __builtin_amdgcn_fmed3f(sv[j - 1], sv[j], x);
    sv[0] = fmaxf(sv[0], x);
}

// one block per cluster slot (a segment inside its cluster): copy its chunk boxes into cluster order
__global__ void k_knn_chunk_table(const int32_t* __restrict__ order, const int32_t* __restrict__ dst,
                                  const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_chunk_off,
                                  const float* __restrict__ chunk_box, const int32_t* __restrict__ slot_chunk0,
                                  float* __restrict__ cc) {
    const int slot = blockIdx.x, sg = order[slot];
    const int m = seg_off[sg + 1] - seg_off[sg], d = dst[slot];
    const int nch = (m + kChunkPts - 1) / kChunkPts;
    const float* src = chunk_box + (size_t)seg_chunk_off[sg] * 8;
    float* out = cc + (size_t)slot_chunk0[slot] * 8;
    for (int i = threadIdx.x; i < nch * 8; i += blockDim.x) {
        const int j = i >> 3;
        out[i] = (i & 7) == 7 ? __int_as_float(((d + j * kChunkPts) << 6) | (min(kChunkPts, m - j * kChunkPts) - 1)) : src[i];
    }
}

template <int K>
__global__ __launch_bounds__(64) void k_cluster_knn_2pass(
    const float4* __restrict__ sxyzw, const int32_t* __restrict__ smpos, const int32_t* __restrict__ cl_off,
    const int32_t* __restrict__ tile_cl, const int32_t* __restrict__ tile_lo, const int32_t* __restrict__ tile_hi,
    const int32_t* __restrict__ tile_chunk0, const int32_t* __restrict__ cl_chunk_off, const float4* __restrict__ cc,
    int pos0, int32_t* __restrict__ knn, int dbg_arg) {
    const int dbg = kKnnProfile ? dbg_arg : 0;
    __shared__ float4 cw[kChunkPts + kQuadS];
    __shared__ int ci[kChunkPts + kQuadS];
    constexpr int kBuf2 = 12;                                 // pass 2 appends ~K keys per lane in total; pass 1 gets 2x the slots
    __shared__ unsigned long long buf[kBuf2][64];             // pass 2 append buffer; pass 1 uses it as float[2 * kBuf2][64]
    const int t = blockIdx.x;
    const int c = tile_cl[t];
    const int clo = cl_off[c], n = cl_off[c + 1] - clo;
    const int lane = threadIdx.x;
    const int q = tile_lo[t] + lane;                          // SORTED position
    const bool active = q < tile_hi[t];
    const int myrow = active ? smpos[q] : 0;                  // member position = output row
    if (n <= K) {                                            // model.py:516-518 (block-uniform)
        if (active) {
            int32_t* o = knn + (size_t)myrow * K;
#pragma unroll
            for (int j = 0; j < K; ++j) o[j] = j < n ? clo + j : pos0;
        }
        return;
    }
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) me = sxyzw[q];
    const int ch0 = cl_chunk_off[c], nch = cl_chunk_off[c + 1] - ch0;
    const int start = tile_chunk0[t];                         // cluster-relative chunk holding the tile's first query
    // the tile's query box (idle lanes repeat lane 0) and largest |q|^2, for the coarse chunk test
    float qlo[3], qhi[3], qw;
    {
        const float4 m0 = active ? me : make_float4(__shfl(me.x, 0), __shfl(me.y, 0), __shfl(me.z, 0), __shfl(me.w, 0));
        qlo[0] = qhi[0] = m0.x; qlo[1] = qhi[1] = m0.y; qlo[2] = qhi[2] = m0.z; qw = m0.w;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { qlo[k] = fminf(qlo[k], __shfl_xor(qlo[k], o)); qhi[k] = fmaxf(qhi[k], __shfl_xor(qhi[k], o)); }
            qw = fmaxf(qw, __shfl_xor(qw, o));
        }
    }
    float* fbuf = reinterpret_cast<float*>(&buf[0][0]);       // [2 * kBuf2][64]
    constexpr int kBufF = 2 * kBuf2;

    // The walk over the cluster's chunks (ring order from the tile's own chunk) is the same in both passes.
    //   weakest()            smallest score any lane still accepts (coarse test, one value per wave)
    //   reject(box)          exact test: no lane can use a point of this box
    //   consider(score, i)   the pass's sink; full() / drain() its buffer management
    auto walk = [&](auto weakest, auto reject, auto consider, auto full, auto drain) {
        for (int b0 = 0; b0 < nch; b0 += 64) {
            const int nb = min(64, nch - b0);
            int rel = start + b0 + lane;
            if (rel >= nch) rel -= nch;
            const float4 d0 = lane < nb ? cc[(size_t)(ch0 + rel) * 2] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 d1 = lane < nb ? cc[(size_t)(ch0 + rel) * 2 + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
            // coarse: box-to-box distance, the margin of box_score_bound for the largest |q|^2 of the tile
            const float gx = fmaxf(fmaxf(d0.x - qhi[0], qlo[0] - d0.w), 0.f);
            const float gy = fmaxf(fmaxf(d0.y - qhi[1], qlo[1] - d1.x), 0.f);
            const float gz = fmaxf(fmaxf(d0.z - qhi[2], qlo[2] - d1.y), 0.f);
            const float coarse = -((gx * gx + gy * gy) + gz * gz) * 0.999999f + 9.6e-7f * (qw + d1.z);
            unsigned long long live = __ballot(lane < nb && coarse >= weakest());
            // software pipeline: the operands of the next live chunk are requested before the current one is scanned
            float4 pre_v = make_float4(0.f, 0.f, 0.f, INFINITY);
            int pre_i = 0x7fffffff;
            auto request = [&](int j) {
                const int w = __float_as_int(__shfl(d1.w, j)), p0 = w >> 6, m = (w & 63) + 1;
                const bool in = lane < m;
                pre_v = in ? sxyzw[p0 + lane] : make_float4(0.f, 0.f, 0.f, INFINITY);
                pre_i = in ? smpos[p0 + lane] - clo : 0x7fffffff;
            };
            if (live) request(__ffsll((unsigned long long)live) - 1);
            while (live) {
                const int j = __ffsll((unsigned long long)live) - 1;
                live &= live - 1;
                const float4 nv = pre_v;
                const int ni = pre_i;
                if (live) request(__ffsll((unsigned long long)live) - 1);
                float bx[7] = {__shfl(d0.x, j), __shfl(d0.y, j), __shfl(d0.z, j), __shfl(d0.w, j), __shfl(d1.x, j), __shfl(d1.y, j), __shfl(d1.z, j)};
                if (reject(bx)) continue;
                const int m = (__float_as_int(__shfl(d1.w, j)) & 63) + 1;
                __builtin_amdgcn_wave_barrier();
                if (lane < kChunkPts + kQuadS) { cw[lane] = nv; ci[lane] = ni; }
                __builtin_amdgcn_wave_barrier();
                for (int e = 0; e < m; e += kQuadS) {
#pragma unroll
                    for (int u = 0; u < kQuadS; ++u) consider(score4(me, cw[e + u]), ci[e + u]);
                    if (full()) drain();
                }
            }
        }
        drain();
    };

    const unsigned long long t0 = (dbg & 16) ? __builtin_readcyclecounter() : 0ull;
    // ---- pass 1: the K best scores
    float floor_score;
    {
        float sv[K];
#pragma unroll
        for (int j = 0; j < K; ++j) sv[j] = -INFINITY;
        float thr = active ? -INFINITY : INFINITY;            // idle lanes never accept
        int cnt = 0;
        auto drain = [&]() {
            int mxc = cnt;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mxc = max(mxc, __shfl_xor(mxc, o));
            for (int u = 0; u < mxc; ++u) score_insert<K>(sv, u < cnt ? fbuf[u * 64 + lane] : -INFINITY);
            cnt = 0;
            if (active) thr = sv[K - 1];
        };
        walk([&]() {
                 float w = thr;
#pragma unroll
                 for (int o = 32; o > 0; o >>= 1) w = fminf(w, __shfl_xor(w, o));
                 return w;
             },
             [&](const float* bx) { return !__any(box_score_bound(me, bx) >= thr); },
             [&](float sc, int) {
                 if (sc > thr) {
                     fbuf[cnt * 64 + lane] = sc;
                     ++cnt;
                 }
             },
             [&]() { return __any(cnt > kBufF - kQuadS - 1); }, drain);
        floor_score = sv[K - 1];
    }
    const unsigned long long t1 = (dbg & 16) ? __builtin_readcyclecounter() : 0ull;
    // ---- pass 2: exact keys of everything at or above the floor
    unsigned long long kv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) kv[j] = 0ull;
    {
        // lowest key of any score that compares equal to the floor (+-0 compare equal, -0.0 has the lower key)
        const unsigned long long floor_key = make_key(floor_score == 0.f ? -0.f : floor_score, 0x7fffffff) & 0xffffffff00000000ull;
        unsigned long long thr = active ? floor_key - 1ull : ~0ull;
        float wk = active ? floor_score : INFINITY;           // the floor never moves: one reduction for the whole pass
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wk = fminf(wk, __shfl_xor(wk, o));
        int cnt = 0;
        auto drain = [&]() {
            int mxc = cnt;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mxc = max(mxc, __shfl_xor(mxc, o));
            for (int u = 0; u < mxc; ++u) key_insert<K>(kv, u < cnt ? buf[u][lane] : 0ull);
            cnt = 0;
            if (active && kv[K - 1] > thr) thr = kv[K - 1];
        };
        walk([&]() { return wk; },
             [&](const float* bx) { return !__any(make_key(box_score_bound(me, bx), 0) > thr); },
             [&](float sc, int idx) {
                 const unsigned long long key = make_key(sc, idx);
                 if (key > thr) {
                     buf[cnt][lane] = key;
                     ++cnt;
                 }
             },
             [&]() { return __any(cnt > kBuf2 - kQuadS - 1); }, drain);
    }
    if (active) {
        int32_t* o = knn + (size_t)myrow * K;
#pragma unroll
        for (int j = 0; j < K; ++j) o[j] = clo + key_index(kv[j]);
    }
    if ((dbg & 16) && lane == 0) {
        const unsigned long long t2 = __builtin_readcyclecounter();
        atomicAdd(&g_knn5_stats[0], 1ull);
        atomicAdd(&g_knn5_stats[1], t1 - t0);
        atomicAdd(&g_knn5_stats[3], t2 - t1);
    }
}

}  // namespace

// Which in-cluster kNN kernel the pipeline launches for a layer of T tiles: 0 = two-pass over the chunk table,
// 1 / 2 / 4 = one-pass with that many waves per tile, 8 = one-pass x1 seeded from the previous kNN layer where there is
// one.  Measured on MI355X (ms per launch, layers 2 + 3, solo):
//   150k pts / 1.5k segs   one-pass x1 0.36 + 0.46   two-pass 0.38 + 0.50   (x4: 0.32 + 0.50, but 584 vs 670 scenes/s in the bench)
//   500k pts / 5k segs     one-pass x1 0.70 + 1.10   two-pass 0.59 + 1.06
//   150k pts / 100 segs    one-pass x1 1.27 + 1.42   two-pass 1.13 + 1.23
// and the same bench throughput for x1 and two-pass (650-690 scenes/s).  Default (override < 0): one wave per tile once
// the launch fills the GPU, seeded.  `override` is a per-pipeline setting (sg_pipeline_set_knn_variant): the library
// keeps no process-wide state.
int sg::knn_variant_for(int T, int override) {
    if (override == 0 || override == 1 || override == 2 || override == 4 || override == 8) return override;
    return T >= 2048 ? 8 : T >= 1024 ? 2 : 4;
}

namespace sg {

// the over-segments of every slot must fit the one-launch LDS sort (max_seg <= sort cap); the engine routes scenes with
// larger segments through the single-scene library-sort path
bool sort_boxes_fits_lds(int max_seg) { return max_seg <= kSortCap; }

// which: 1 = the segments of <= kSortCap points, 2 = the larger ones, 3 = both (the engine runs the two chains on two streams: engine.cpp, phase P0)
int b_sort_boxes(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st, int which) {
    if (bd.nslots == 0 || bd.max_S == 0) return SG_OK;
    if (which & 1) k_segment_sort_boxes_b<<<dim3(bd.max_S, bd.nslots), 256, 0, st>>>(d_ctx);
    if ((which & 2) && bd.max_big > 0) {                                       // windows without a big segment exit at once
        const int wins = sg::cdiv(bd.max_N, kWin);
        k_bigseg_box_b<<<dim3(bd.max_big * kBigPieces, bd.nslots), kPieceBlock, 0, st>>>(d_ctx);
        k_bigseg_keys_b<<<dim3(bd.max_big * kBigPieces, bd.nslots), kPieceBlock, 0, st>>>(d_ctx);
        k_bigseg_scatter_b<<<dim3(bd.max_big * kBigPieces, bd.nslots), kPieceBlock, 0, st>>>(d_ctx);
        k_bigseg_runs_b<<<dim3(wins, bd.nslots), kBigBlock, 0, st>>>(d_ctx);
        k_bigseg_boxes_b<<<dim3(wins, bd.nslots), 512, 0, st>>>(d_ctx);
    }
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int b_layer_layout(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st) {
    if (bd.nslots == 0 || bd.max_S == 0) return SG_OK;
    // 256 threads per segment although a segment averages 100 rows: blocks of 128 / 64 threads measured 9.4 / 11.1 instead of 8.7 us per scene
    // (the per-block chain order -> seg_off -> rows is latency, and fewer lanes per block means more trips through it)
    k_layer_layout_b<<<dim3(bd.max_S, bd.nslots), 256, 0, st>>>(d_ctx);
    if (bd.max_lay_big > 0) k_layer_layout_big_b<<<dim3(bd.max_lay_big, bd.nslots), 256, 0, st>>>(d_ctx);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

// write_seed: also emit the table in point ids (knn_seed) for the next layer's seeded launch; the one-wave-per-tile kernel does it
// in its output stage, the multi-wave variants leave it to b_knn_seed_points (*wrote_seed tells the caller)
int b_cluster_knn(const SlotCtx* d_ctx, const BatchDims& bd, int waves_per_tile, bool seeded, hipStream_t st, bool write_seed, bool* wrote_seed) {
    if (wrote_seed) *wrote_seed = false;
    if (bd.nslots == 0 || bd.max_T == 0) return SG_OK;
    if (bd.max_N > sgknn::kListMaxPoints) return sg::fail(SG_EUNSUP, "in-cluster kNN: %d points in a scene, the list keys carry 20 index bits (N <= %d)", bd.max_N, sgknn::kListMaxPoints);
    const dim3 grid(bd.nslots, bd.max_T);
    if (seeded) k_cluster_knn_sorted_b<20, 1, true><<<grid, 64, 0, st>>>(d_ctx, 0);
    else if (waves_per_tile == 1) {
        k_cluster_knn_sorted_b<20, 1, false><<<grid, 64, 0, st>>>(d_ctx, write_seed ? 1 : 0);
        if (wrote_seed) *wrote_seed = write_seed;
    } else if (waves_per_tile == 2) k_cluster_knn_sorted_b<20, 2, false><<<grid, 128, 0, st>>>(d_ctx, 0);
    else k_cluster_knn_sorted_b<20, 4, false><<<grid, 256, 0, st>>>(d_ctx, 0);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int b_knn_seed_points(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st) {
    if (bd.nslots == 0 || bd.max_N == 0) return SG_OK;
    k_knn_seed_points_b<<<dim3(sg::cdiv((long long)bd.max_N * 20, 256), bd.nslots), 256, 0, st>>>(d_ctx);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

}  // namespace sg

#ifdef SG_KNN_PROFILE
static int g_knn5_dbg = getenv("SG_KNN_DEBUG") ? atoi(getenv("SG_KNN_DEBUG")) : 0;   // profiling knob (16 = counters + cycle stamps)
#else
static constexpr int g_knn5_dbg = 0;
#endif

extern "C" {

#ifdef SG_KNN_PROFILE
// profiling builds only: copies and clears the sorted-kNN work counters
int sg_debug_knn5_stats(unsigned long long* h_out) {
    SG_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_knn5_stats), sizeof(unsigned long long) * 16));
    unsigned long long z[16] = {0};
    SG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_knn5_stats), z, sizeof z));
    return SG_OK;
}
#endif

#ifdef SG_KNN_SELFCHECK
int sg_debug_knn_check(unsigned long long* h_out) {
    SG_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_knn_check), sizeof(unsigned long long) * 136));
    unsigned long long z[136] = {0};
    SG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_knn_check), z, sizeof z));
    return SG_OK;
}
#endif

// two N-key arrays + one BigScratch record per possible big segment (k_bigseg_box / keys / scatter)
size_t sg_segment_sort_ws_bytes(int N) { return sg::align_up((size_t)std::max(N, 1) * 16, 64) + bigseg_scratch_slots(std::max(N, 1)) * sizeof(BigScratch) + 64; }

int sg_layer_layout(const float* d_data, int N, const int32_t* d_seg_points, const int32_t* d_seg_off, const int32_t* d_sperm, int S,
                    const int32_t* d_order, const int32_t* d_dst, const int32_t* d_cl, const float* d_cl_mean, int32_t* d_members,
                    int32_t* d_pos_of_point, int32_t* d_cluster_of_pos, int32_t* d_slot_of_pos, float* d_x9m, float* d_sxyzw,
                    int32_t* d_smpos, float* d_point_rec, int32_t* d_seed_id, unsigned int* d_range_bits, void* stream) {
    SG_REQUIRE(N >= 0 && S >= 0 && d_sperm && d_cl_mean && d_members && d_pos_of_point && d_cluster_of_pos && d_slot_of_pos && d_x9m &&
                   d_sxyzw && d_smpos, "sg_layer_layout: bad arguments");
    if (S == 0) return SG_OK;
    k_layer_layout<<<S, 128, 0, sg::as_stream(stream)>>>(d_data, d_seg_points, d_seg_off, d_sperm, d_order, d_dst, d_cl, d_cl_mean, d_members,
                                                        d_pos_of_point, d_cluster_of_pos, d_slot_of_pos, d_x9m,
                                                        reinterpret_cast<float4*>(d_sxyzw), d_smpos, reinterpret_cast<float4*>(d_point_rec), d_seed_id, d_range_bits);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_segment_sort_boxes(const float* d_data, int N, const int32_t* d_seg_points, const int32_t* d_seg_off,
                          const int32_t* d_seg_of_point, int S, const int32_t* d_seg_chunk_off, int max_seg, float* d_segbox,
                          int32_t* d_sperm, float* d_chunk_box, double* d_seg_sums, void* d_ws, size_t ws_bytes, void* stream) {
    SG_REQUIRE(N >= 0 && S >= 0 && max_seg >= 0 && d_segbox && d_sperm && d_chunk_box, "sg_segment_sort_boxes: bad arguments");
    if (N == 0 || S == 0) return SG_OK;
    k_segment_sort_boxes<<<S, 256, 0, sg::as_stream(stream)>>>(d_data, d_seg_points, d_seg_off, d_seg_chunk_off, d_segbox, d_sperm, d_chunk_box,
                                                               d_seg_sums);
    if (max_seg > kSortCap) {                                  // segments beyond one block's LDS: cell-bucketed LDS sort, scratch = 2 x N keys
        if (!d_ws || ws_bytes < sg_segment_sort_ws_bytes(N)) return sg::fail(SG_ENOMEM, "sg_segment_sort_boxes: workspace too small (%zu < %zu)", ws_bytes, sg_segment_sort_ws_bytes(N));
        unsigned long long* keys = reinterpret_cast<unsigned long long*>(d_ws);
        const int wins = sg::cdiv(N, kWin);
        k_bigseg_box<<<dim3(S, kBigPieces), kPieceBlock, 0, sg::as_stream(stream)>>>(d_data, d_seg_points, d_seg_off, keys, N);
        k_bigseg_keys<<<dim3(S, kBigPieces), kPieceBlock, 0, sg::as_stream(stream)>>>(d_data, d_seg_points, d_seg_off, d_segbox, d_seg_sums, keys, N);
        k_bigseg_scatter<<<dim3(S, kBigPieces), kPieceBlock, 0, sg::as_stream(stream)>>>(d_seg_off, keys, keys + N, N);
        k_bigseg_runs<<<wins, kBigBlock, 0, sg::as_stream(stream)>>>(d_seg_off, S, N, keys, keys + N, d_sperm);
        k_bigseg_boxes<<<wins, 512, 0, sg::as_stream(stream)>>>(d_data, d_seg_points, d_seg_off, S, N, d_seg_chunk_off, d_sperm, d_chunk_box);
    }
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_knn_operands(const float* d_data, const int32_t* d_seg_points, const int32_t* d_seg_off, const int32_t* d_sperm, int S,
                    const int32_t* d_order, const int32_t* d_dst, float* d_sxyzw, int32_t* d_smpos, void* stream) {
    SG_REQUIRE(S >= 0 && d_sxyzw && d_smpos, "sg_knn_operands: bad arguments");
    if (S == 0) return SG_OK;
    k_knn_operands<<<S, 128, 0, sg::as_stream(stream)>>>(d_data, d_seg_points, d_seg_off, d_sperm, d_order, d_dst,
                                                        reinterpret_cast<float4*>(d_sxyzw), d_smpos);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_knn_seed_points(const int32_t* d_knn, const int32_t* d_members, int N, int k, int32_t* d_seed, void* stream) {
    SG_REQUIRE(N >= 0 && k > 0 && d_knn && d_members && d_seed, "sg_knn_seed_points: bad arguments");
    if (N == 0) return SG_OK;
    k_knn_seed_points<<<sg::cdiv((long long)N * k, 256), 256, 0, sg::as_stream(stream)>>>(d_knn, d_members, N, k, d_seed);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_cluster_knn_seeded(const float* d_sxyzw, const int32_t* d_smpos, int N, const int32_t* d_cl_off, const int32_t* d_tile_cl,
                          const int32_t* d_tile_lo, const int32_t* d_tile_hi, int T, const int32_t* d_cl_seg_off,
                          const int32_t* d_order, const int32_t* d_dst, const int32_t* d_seg_off, const int32_t* d_seg_chunk_off,
                          const float* d_segbox, const float* d_chunk_box, const int32_t* d_slot_of_pos, const int32_t* d_seed,
                          const int32_t* d_seg_prevcl, const int32_t* d_members, const float* d_point_rec,
                          int k, int pos0, int32_t* d_knn, void* stream) {
    SG_REQUIRE(N >= 0 && T >= 0 && d_knn && d_sxyzw && d_smpos && d_seed && d_seg_prevcl && d_members && d_point_rec,
               "sg_cluster_knn_seeded: bad arguments");
    SG_REQUIRE(N <= sgknn::kListMaxPoints, "sg_cluster_knn_seeded: the list keys carry 20 index bits (N <= 2^20)");
    if (k != 20) return sg::fail(SG_EUNSUP, "sg_cluster_knn_seeded: only k == 20 is built (model.py:788,829), got %d", k);
    if (T == 0) return SG_OK;
    k_cluster_knn_sorted<20, 1, true><<<T, 64, 0, sg::as_stream(stream)>>>(
        reinterpret_cast<const float4*>(d_sxyzw), d_smpos, d_cl_off, d_tile_cl, d_tile_lo, d_tile_hi, d_cl_seg_off, d_order, d_dst,
        d_seg_off, d_seg_chunk_off, d_segbox, d_chunk_box, d_slot_of_pos, pos0, d_knn, g_knn5_dbg, d_seed, d_seg_prevcl, d_members,
        reinterpret_cast<const float4*>(d_point_rec));
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_knn_chunk_table(const int32_t* d_order, const int32_t* d_dst, const int32_t* d_seg_off, const int32_t* d_seg_chunk_off,
                       const float* d_chunk_box, int S, const int32_t* d_slot_chunk0, float* d_cc, void* stream) {
    SG_REQUIRE(S >= 0 && d_cc && d_slot_chunk0, "sg_knn_chunk_table: bad arguments");
    if (S == 0) return SG_OK;
    k_knn_chunk_table<<<S, 64, 0, sg::as_stream(stream)>>>(d_order, d_dst, d_seg_off, d_seg_chunk_off, d_chunk_box, d_slot_chunk0, d_cc);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_cluster_knn_2pass(const float* d_sxyzw, const int32_t* d_smpos, int N, const int32_t* d_cl_off, const int32_t* d_tile_cl,
                         const int32_t* d_tile_lo, const int32_t* d_tile_hi, const int32_t* d_tile_chunk0, int T,
                         const int32_t* d_cl_chunk_off, const float* d_cc, int k, int pos0, int32_t* d_knn, void* stream) {
    SG_REQUIRE(N >= 0 && T >= 0 && d_knn && d_sxyzw && d_smpos && d_cc, "sg_cluster_knn_2pass: bad arguments");
    SG_REQUIRE(N < (1 << 25), "sg_cluster_knn_2pass: sorted positions are packed in 25 bits (N < 2^25)");
    if (k != 20) return sg::fail(SG_EUNSUP, "sg_cluster_knn_2pass: only k == 20 is built (model.py:788,829), got %d", k);
    if (T == 0) return SG_OK;
    k_cluster_knn_2pass<20><<<T, 64, 0, sg::as_stream(stream)>>>(reinterpret_cast<const float4*>(d_sxyzw), d_smpos, d_cl_off, d_tile_cl,
                                                                 d_tile_lo, d_tile_hi, d_tile_chunk0, d_cl_chunk_off,
                                                                 reinterpret_cast<const float4*>(d_cc), pos0, d_knn, g_knn5_dbg);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_cluster_knn_sorted_w(const float* d_sxyzw, const int32_t* d_smpos, int N, const int32_t* d_cl_off, const int32_t* d_tile_cl,
                            const int32_t* d_tile_lo, const int32_t* d_tile_hi, int T, const int32_t* d_cl_seg_off,
                            const int32_t* d_order, const int32_t* d_dst, const int32_t* d_seg_off, const int32_t* d_seg_chunk_off,
                            const float* d_segbox, const float* d_chunk_box, const int32_t* d_slot_of_pos, int k, int pos0,
                            int waves_per_tile, int32_t* d_knn, void* stream) {
    SG_REQUIRE(N >= 0 && T >= 0 && d_knn && d_sxyzw && d_smpos, "sg_cluster_knn_sorted: bad arguments");
    SG_REQUIRE(N <= sgknn::kListMaxPoints, "sg_cluster_knn_sorted: the list keys carry 20 index bits (N <= 2^20)");
    if (k != 20) return sg::fail(SG_EUNSUP, "sg_cluster_knn_sorted: only k == 20 is built (model.py:788,829), got %d", k);
    if (T == 0) return SG_OK;
    const int slices = (waves_per_tile == 1 || waves_per_tile == 2 || waves_per_tile == 4) ? waves_per_tile : (T >= 2048 ? 1 : T >= 1024 ? 2 : 4);
#define SG_KNN_LAUNCH(S)                                                                                                  \
    k_cluster_knn_sorted<20, S><<<T, 64 * S, 0, sg::as_stream(stream)>>>(                                                 \
        reinterpret_cast<const float4*>(d_sxyzw), d_smpos, d_cl_off, d_tile_cl, d_tile_lo, d_tile_hi, d_cl_seg_off, d_order, d_dst, \
        d_seg_off, d_seg_chunk_off, d_segbox, d_chunk_box, d_slot_of_pos, pos0, d_knn, g_knn5_dbg)
    if (slices == 1) SG_KNN_LAUNCH(1);
    else if (slices == 2) SG_KNN_LAUNCH(2);
    else SG_KNN_LAUNCH(4);
#undef SG_KNN_LAUNCH
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int sg_cluster_knn_sorted(const float* d_sxyzw, const int32_t* d_smpos, int N, const int32_t* d_cl_off, const int32_t* d_tile_cl,
                          const int32_t* d_tile_lo, const int32_t* d_tile_hi, int T, const int32_t* d_cl_seg_off,
                          const int32_t* d_order, const int32_t* d_dst, const int32_t* d_seg_off, const int32_t* d_seg_chunk_off,
                          const float* d_segbox, const float* d_chunk_box, const int32_t* d_slot_of_pos, int k, int pos0,
                          int32_t* d_knn, void* stream) {
    return sg_cluster_knn_sorted_w(d_sxyzw, d_smpos, N, d_cl_off, d_tile_cl, d_tile_lo, d_tile_hi, T, d_cl_seg_off, d_order, d_dst, d_seg_off,
                                   d_seg_chunk_off, d_segbox, d_chunk_box, d_slot_of_pos, k, pos0, 0, d_knn, stream);
}

}  // extern "C"
