// a13: get_graph_feature2 + MLP2 / MLP3 (reference seggroup/model.py:83-138): the EdgeConv stage.
//
//   edge row (point i, neighbour slot j):  e = [x_j - x_i , x_i]              (18 = 2 x 9 channels)
//   MLP2:  max_j LReLU(BN1(W1 e))                                              (64 channels)
//   MLP3:  max_j LReLU(BN2(W2 LReLU(BN1(W1 e))))
//   BN = BatchNorm2d in TRAIN mode: per-channel batch mean / biased variance over all N*k rows
//   (infer.py never calls eval()).
//
// This is the only dense contraction on the hot path and it is MFMA-bound (39 GFLOP/scene, ~280
// flop/B): it runs on the matrix pipe -- the per-point part on v_mfma_f32_32x32x2_f32 (exact fp32, bit-equal to an fmaf chain),
// the per-edge part on 16-bit MFMAs over split fp32 operands (below).
//
// The x_i half of the edge feature does not depend on the neighbour: its contribution to conv1 (plus the folded shift) is
// evaluated once per point and seeds every slot's accumulator, so conv1 contracts 9 inputs per slot, not 18.
//
// Mapping (one wave = one tile of 32 points, loop over the k=20 neighbour slots):
//   D^T[ch][row] = sum_k W[ch][k] * E[row][k]:  A operand = weights (M = 32 channels per tile, two
//   tiles), B operand = edge features (N = 32 rows).  Lane l holds row (l & 31) and k-parity (l >> 5);
//   the accumulator gives every lane 16 channels of ITS OWN row per tile, so
//     * LeakyReLU / max-over-k / statistics are element-wise on registers (rows never cross lanes),
//     * the conv1 accumulator can be fed straight back as the B operand of conv2: MFMA step r pairs
//       channel c(r)   (lanes 0-31) with channel c(r)+4 (lanes 32-63), which is exactly how the
//       accumulator is laid out (row = (r&3) + 8(r>>2) + 4(lane>>5)); only the A operand (W2) has to
//       be fetched in that permuted k order, and it comes pre-arranged from LDS.
//   Inner BatchNorm (layer 1 of MLP3) is folded: w' = a*w (a = gamma/sqrt(var+eps)) and the shift
//   b' = beta - a*mean enters as the accumulator's initial value.
//
// Statistics need a global barrier, but the LAST layer does not need a second evaluation: per channel,
// y -> LReLU(a*y + b') is monotone (non-decreasing for a >= 0, non-increasing for a < 0, also in fp32:
// rounding is monotone), so  max_j LReLU(a*y_j + b') == LReLU(a*ext_j y_j + b')  with ext = max or min by
// the sign of gamma -- known before the statistics are.  The sign is folded into the weight ROWS while
// they are staged in LDS (negation commutes exactly with an fma chain), so the pass that accumulates the
// statistics of y' = sgn(gamma)*y also keeps  E = max_j y'_j  per (point, channel); a one-pass element-wise
// epilogue then applies  LReLU(|a|*E + b').  The INNER BatchNorm of MLP3 needs conv1's statistics before conv2 can
// run; conv1 is linear, so they follow from the first and second moments of the 18-channel edge features
// (k_edge_moments: VALU only, no MFMA pass).  MLP2 and MLP3 are therefore ONE MFMA pass each (S1X, S2X): the 38.4 GFLOP
// of a single evaluation, instead of 2 and 3 passes (82 GFLOP) or materialising the 768 MB [N,20,64] tensor.
// Per-block fp64 partial sums are combined in fixed order by a one-block finalize kernel, so the
// result is bit-reproducible run to run.
//
// conv1 and conv2 run on the 16-bit matrix pipe at fp32 accuracy: the fp32 MFMA issues at the fp32 VECTOR rate (64 cycles per
// 32x32x2 on a SIMD), v_mfma_f32_32x32x16_{bf16,f16} moves 8x the k-depth in half the cycles.  Every fp32 operand is cut into
// 16-bit pieces whose products are exact in fp32, and the matrix pipe accumulates the products that matter in fp32:
//   conv1 (d = x_j - x_i, raw data of unknown range): three bf16 pieces by truncation (x = x1 + x2 + x3, 8 + 8 + 8 significand
//     bits, the full fp32 exponent range), six products
//         w x ~= w1 x1 + (w1 x2 + w2 x1) + (w1 x3 + w2 x2 + w3 x1)        dropped: w2 x3 + w3 x2 + w3 x3 <= 3 * 2^-24 |w x|
//   conv2 (64 -> 64, 78 % of the stage's flops; its operand h = LReLU(BN1(.)) has a PROVABLE bound): two fp16 pieces by
//     round-to-nearest (hi = rn16(x), lo = rn16(x - hi): 11 + 11 bits and the sign of the remainder, |x - hi - lo| <= 2^-22 |x|),
//     three products
//         w x ~= w_hi x_hi + (w_hi x_lo + w_lo x_hi)                        dropped: w_lo x_lo ~ 2^-22 |w x|
//     Half the MFMAs of the bf16 split and a cheaper cut (v_cvt_pk_f16_f32 rounds and packs a pair in one instruction).  fp16's
//     exponent range is narrow, so both operands are moved into it by powers of two (exact; divided out of the statistics and
//     the maxima at the end): see k_bn_fold_moments, which also writes the pre-split weight image once per scene.  The matrix
//     pipe keeps fp16 subnormal inputs (tools/micro/mfma_f16_split.hip), so a small value's low piece costs at most 2^-25 / T
//     of absolute error.  Measured (same micro test, |y| ~ 0.5, K = 64): max error 1.6e-7 vs 0.9e-7 (bf16 x 3, six products)
//     and 1.3e-7 (plain fp32 fmaf chain); against the float64 oracle the point features stay within 2e-5 and every fixture's
//     labels are unchanged.
// The accumulator of conv1 is fed straight back as the B operand of conv2: a 16-deep k block takes 8 consecutive accumulator
// registers of a lane (lanes 0-31: k 0-7, lanes 32-63: k 8-15), and the weights are staged in LDS in exactly that channel
// order, already split.  conv1's six products are packed along K (struct Lds, a1p): 54 of 64 k-slots of four MFMAs per output tile instead of
// six MFMAs with 9 of 16.  Per neighbour slot: 8 (conv1) + 24 (conv2) MFMAs of 32 cycles instead of 10 + 64 fp32 MFMAs of 64.
//
// Conditioning: channels 0..2 of x_i are ABSOLUTE coordinates.  BatchNorm is invariant to a per-channel constant added to
// its input, so every kernel here evaluates the x_i half of conv1 on x_i - [c, 0] with c = the XYZ of row 0 (any fixed point
// near the cloud): the statistics (sum y, sum y^2 and the edge-feature moments) are then sums of values of the cloud's
// extent, not of its distance from the origin, and var = E[y^2] - mean^2 does not cancel for a scan that sits 100 m away
// from the origin.  The differences d = x_j - x_i are formed from the raw coordinates exactly as the reference does.
#include <cstddef>
#include "engine_ctx.h"
#include "sg_common.h"
#include "wave_ops.h"
#include "edgeconv_slots_gen.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// Profiling builds (make PROFILE=1): where a workgroup's time goes -- 100 MHz stamps (s_memrealtime) at the phase edges of edgeconv_body, summed
// per wave into g_ec_phase[mode][phase]; sg_debug_ec_phases() copies and clears them.  Release builds compile all of it out.
#ifdef SG_KNN_PROFILE
constexpr bool kEcProfile = true;
__device__ unsigned long long g_ec_phase[3][8];
#else
constexpr bool kEcProfile = false;
#endif
struct EcStamp {
    unsigned long long t;
    int mode;
    __device__ __forceinline__ void start(int m) { mode = m; if (kEcProfile) t = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void mark(int phase) {
#ifdef SG_KNN_PROFILE
        const unsigned long long n = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_ec_phase[mode][phase], n - t);
        t = n;
#endif
    }
};

enum { S1X = 1, S2X = 2 };     // conv1 statistics + extremum (MLP2) | conv1' -> conv2 statistics + extremum (MLP3)

constexpr int kWaves = sg::kEdgeWaves;
constexpr int kStagger1 = 0, kStagger2 = 0;    // defaults of the start offset between the two waves of a SIMD (see edgeconv_body)


__device__ inline int acc_channel(int tile, int reg, int half) { return 32 * tile + (reg & 3) + 8 * (reg >> 2) + 4 * half; }

struct Lds {
    // A fragments, four MFMA steps per ds_read_b128 (conflict-free: consecutive lanes, 16 B each)
    // conv1, d columns (used every neighbour slot): the six (weight piece, d piece) products of the bf16 x 3 split PACKED along K -- 54 of
    // the 64 k-slots of four 16-deep MFMAs per output tile carry a product (the 9-deep contraction alone fills 9 of 16).  Lanes 0-31
    // (k 0..7) own d0..d3, lanes 32-63 (k 8..15) own d4..d7, both own d8; per lane and MFMA m four units of two bf16:
    //   m = 0: (w1 | w1)   m = 1: (w2 | w1)   m = 2: (w2 | w3)   meeting B = (x1 | x2), (x1 | x3), (x2 | x1) of the lane's four values
    //   m = 3: d8's products -- lanes 0-31: w1, w1, w1, w2 ; lanes 32-63: w3, w2, 0, 0   meeting B = (x1, x2, x3, x1) of d8
    // a1p[m][t][lane] = the 8 bf16 of W1[32t + (lane&31)][.] in that order (pieces by truncation: w = w1 + w2 + w3)
    u32x4 a1p[4][2][64];
    float4 a1x[2][2][64];    // a1x[t][s>>2][lane][s&3] = W1[32t + (lane&31)][9 + k]    the x_i columns, used once per point
    // conv2, fp16 pieces (k_bn_fold_moments prepares the image once per scene): a2h[piece][ot][kb][lane] = 8 halves = piece of
    // S * sgn(gamma2) * W2[32ot + (lane&31)][acc_channel(kb>>1, 8(kb&1) + j, lane>>5)], j = 0..7
    u32x4 a2h[2][2][4][64];
    float sh1r[2][2][16];    // folded BN1 shift in accumulator-register order: sh1r[tile][half][reg] (b128 reads)
    double acc[kWaves][128]; // per-wave statistics: [0,64) sum, [64,128) sum of squares
};

__device__ __forceinline__ float pow2_scale(float bound, float target) {      // largest power of two p with p * bound <= target
    if (!(bound > 0.f) || !(bound < INFINITY)) return 1.f;
    int e;
    const float m = frexpf(target / bound, &e);                                // target / bound = m 2^e, m in [0.5, 1)
    (void)m;
    return ldexpf(1.f, min(max(e - 1, -40), 40));
}

// kF16 (MLP2, round 3): conv1's operand d = x_j - x_i on TWO fp16 pieces (round to nearest: hi = rn16(d), lo = rn16(d - hi), three
// products w_hi d_hi + w_hi d_lo + w_lo d_hi) instead of three bf16 pieces and six products: 27 k-slots = two MFMAs per output tile
// (4 per neighbour slot instead of 8) and 20 instead of 38 VALU for the cut.  fp16's narrow exponent range is handled with powers of
// two, which are exact: d is scaled by Sd with 2 Sd * (largest |centred coordinate| / |feature| of the layer: `range_bits`, raised by
// the layout kernel) <= 2^12 -- every difference inside a cluster is bounded by twice that value, so no piece can overflow; small
// differences land on fp16 subnormals, which the matrix pipe keeps (absolute error 2^-25 / Sd: 1e-11 of the range) -- and the weights by
// Sw with max |Sw w| in [2^11, 2^12); 1 / (Sw Sd) is divided out of the statistics and the maxima like S2X's scales.
// kAsm (round 4): the K = 20 slot loop of the fp16 variants as ONE hand-scheduled asm statement (edgeconv_slots_gen.h, written by
// tools/gen_edgeconv_asm.py: every MFMA followed by up to six instructions of other slots' VALU stages, software-pipelined over the 20
// slots).  Same instructions on the same values in the same order per accumulator as the C++ loop below it -- bit-identical results;
// the C++ loop stays for K != 20 and as the cross-check (sg_edgeconv_forward_x, flag 1).  MLP3's version keeps the 20 A fragments in
// AGPRs and the point's base accumulator in LDS: 256 VGPRs + 80 AGPRs, ONE wave per SIMD.
constexpr int kAsmK = 20;                  // the neighbour count the generated slot loops (edgeconv_slots_gen.h) are unrolled for
template <int MODE, bool REREAD_A, bool kFused = false, bool kF16 = false, bool kAsm = false>
__device__ __forceinline__ void edgeconv_body(sg::gptr<const float> x9m, sg::gptr<const int32_t> knn, int N, int K,
                                              sg::gptr<const float> w1, sg::gptr<const float> shift1,
                                              sg::gptr<const u32x4> w2img, sg::gptr<const float> scales,
                                              sg::gptr<const float> gamma_last,
                                              sg::gptr<float> ext, sg::gptr<double> partial, int bid,
                                              sg::gptr<const int32_t> cluster_of_pos = nullptr, int ext_stride = 64, int stagger = 0,
                                              sg::gptr<const unsigned int> range_bits = nullptr, int nblk = 1 << 28, int walk_mode = 0) {
    using sg::gptr;
    __shared__ Lds lds;
    // fused epilogue (round 5): the maximum of the wave's current run of tiles inside ONE cluster, per channel, and that cluster's id
    __shared__ float carry_m[kFused ? kWaves : 1][64];
    __shared__ int carry_c[kFused ? kWaves : 1];
    // the fused epilogue's stage (32 rows x 64 maxima per wave); during the slot loop of the fp16 variants: the lanes' neighbour ids [K][64]
    __shared__ float stage_or_ids[(kFused || kF16) ? kWaves : 1][(kFused || kF16) ? 32 * 65 : 1];
    __shared__ float flush_sums[kWaves][128];                     // a tile's 64 sums + 64 sums of squares on their way into lds.acc
    constexpr bool kTwo = MODE == S2X;
    // kAsm, MLP3: the stage strip holds the point's base accumulator during the slot loop ([8][64] float4); the asm reads the neighbour ids
    // straight from the kNN table (one global load per slot, a slot ahead): no LDS is left for an id strip at two workgroups per CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    [[maybe_unused]] const int r = lane & 31, half = lane >> 5;      // (shadowed inside the walk: re-derived from the laundered thread id)
    EcStamp stamp;
    stamp.start(kAsm ? MODE : 0);
    // the two scales of the fp16 conv1
    float Sd = 1.f, Sw = 1.f;
    if (kF16 && kTwo) Sd = scales[3];                           // MLP3: k_bn_fold_moments chose it and wrote the image (w2img + 1024)
    if (kF16 && !kTwo) {
        __shared__ float wred[kWaves], rred[kWaves];
        static_assert(sg::kRangeWords == 64 * kWaves, "one range word per thread");
        float rm = __uint_as_float(range_bits[tid]);              // non-negative floats: the largest word is the range
        float wm = 0.f;
        for (int i = tid; i < 64 * 9; i += 64 * kWaves) wm = fmaxf(wm, fabsf(w1[(i / 9) * 18 + i % 9]));
        wm = sgw::wave_max(wm);
        rm = sgw::wave_max(rm);
        if (lane == 0) { wred[wave] = wm; rred[wave] = rm; }
        __syncthreads();
        wm = wred[0]; rm = rred[0];
#pragma unroll
        for (int w_ = 1; w_ < kWaves; ++w_) { wm = fmaxf(wm, wred[w_]); rm = fmaxf(rm, rred[w_]); }
        Sw = pow2_scale(wm, 4095.f);
        Sd = pow2_scale(2.f * rm, 4096.f);
    }

    // the last layer's rows carry the sign of its BN gamma (see the header): y' = sgn(gamma) * y exactly
    for (int i = tid; i < 2 * 8 * 64; i += 64 * kWaves) {           // x_i columns: [t][step 0..7][lane]; steps 5..7 and k = 9 stay zero
        const int l = i & 63, s = (i >> 6) & 7, t = (i >> 9) & 1;
        const int ch = 32 * t + (l & 31), k = 2 * s + (l >> 5);
        float v = (s < 5 && k < 9) ? w1[ch * 18 + 9 + k] : 0.f;
        if (MODE == S1X && gamma_last[ch] < 0.f) v = -v;
        if (kF16 && !kTwo) v *= Sw * Sd;                            // MLP2: the x_i half joins an accumulator that runs at Sw Sd
        (&lds.a1x[t][s >> 2][l].x)[s & 3] = v;
    }
    if (kF16 && kTwo) {
        for (int i = tid; i < 2 * 2 * 64; i += 64 * kWaves) (&lds.a1p[0][0][0])[i] = w2img[1024 + 1 + i];      // behind conv2's image and the four scales
    } else if (kF16) {
        // fp16 image, two MFMAs per output tile: m = 0: (w_hi | w_hi) meeting B = (d_hi | d_lo) of the lane's four values;
        // m = 1: (w_lo | d8's products) meeting B = (d_hi | d8_hi, d8_lo): lanes 0-31 w_hi[8], w_hi[8]; lanes 32-63 w_lo[8], 0
        for (int i = tid; i < 2 * 2 * 64; i += 64 * kWaves) {
            const int l = i & 63, t = (i >> 6) & 1, m = i >> 7;
            const int ch = 32 * t + (l & 31), hf = l >> 5, c0 = 4 * hf;
            const float sg_ = gamma_last[ch] < 0.f ? -Sw : Sw;
            auto piece = [&](int k, int p) -> unsigned int {          // fp16 bits of the high (p = 1) / low (p = 2) piece of Sw W1[ch][k]
                if (p == 0) return 0u;
                const float v = sg_ * w1[ch * 18 + k];
                const _Float16 h = (_Float16)v;
                const _Float16 lo_ = (_Float16)(v - (float)h);
                return (unsigned int)__builtin_bit_cast(unsigned short, p == 1 ? h : lo_);
            };
            unsigned int u[4];
            const int pa = m == 0 ? 1 : 2;
            u[0] = piece(c0, pa) | (piece(c0 + 1, pa) << 16); u[1] = piece(c0 + 2, pa) | (piece(c0 + 3, pa) << 16);
            if (m == 0) { u[2] = u[0]; u[3] = u[1]; }
            else { u[2] = piece(8, hf ? 2 : 1); u[3] = piece(8, hf ? 0 : 1); }
            lds.a1p[m][t][l] = u32x4{u[0], u[1], u[2], u[3]};
        }
    } else
    for (int i = tid; i < 4 * 2 * 64; i += 64 * kWaves) {           // d columns: the packed product image (struct Lds)
        const int l = i & 63, t = (i >> 6) & 1, m = i >> 7;
        const int ch = 32 * t + (l & 31), hf = l >> 5;
        const bool neg = MODE == S1X && gamma_last[ch] < 0.f;
        auto piece = [&](int k, int p) -> unsigned int {              // bf16 bits of piece p (1..3) of W1[ch][k], 0 for p == 0
            if (p == 0) return 0u;
            float v = w1[ch * 18 + k];
            if (neg) v = -v;
            const float a1 = __uint_as_float(__float_as_uint(v) & 0xffff0000u);
            const float r1 = v - a1;
            const float a2 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
            const float r2 = r1 - a2;
            return __float_as_uint(p == 1 ? a1 : p == 2 ? a2 : r2) >> 16;
        };
        unsigned int u[4];
        if (m < 3) {
            const int pa = m == 0 ? 1 : 2, pb = m == 2 ? 3 : 1;       // weight piece of units 0-1 | units 2-3
            const int c0 = 4 * hf;
            u[0] = piece(c0, pa) | (piece(c0 + 1, pa) << 16); u[1] = piece(c0 + 2, pa) | (piece(c0 + 3, pa) << 16);
            u[2] = piece(c0, pb) | (piece(c0 + 1, pb) << 16); u[3] = piece(c0 + 2, pb) | (piece(c0 + 3, pb) << 16);
        } else {
            const int pw[2][4] = {{1, 1, 1, 2}, {3, 2, 0, 0}};
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = piece(8, pw[hf][q]);  // (w[8] piece, 0)
        }
        lds.a1p[m][t][l] = u32x4{u[0], u[1], u[2], u[3]};
    }
    if (kTwo) {
        for (int i = tid; i < 2 * 2 * 4 * 64; i += 64 * kWaves) (&lds.a2h[0][0][0][0])[i] = w2img[i];
    }
    if (tid < 64) {
        const int t_ = tid >> 5, h_ = (tid >> 4) & 1, q_ = tid & 15;
        lds.sh1r[t_][h_][q_] = shift1 ? shift1[acc_channel(t_, q_, h_)] : 0.f;
    }
    for (int i = tid; i < kWaves * 128; i += 64 * kWaves) (&lds.acc[0][0])[i] = 0.0;
    __syncthreads();
    stamp.mark(0);                                            // weights staged

    // Two waves share a SIMD (different workgroups, started together, running the same loop): they fall into lockstep -- both in
    // their MFMA phase, then both in their VALU phase -- and the matrix pipe idles while both cut operands and add statistics.  The
    // wave in the odd slot of its SIMD (HW_ID.wave_id) starts `stagger` x 64 cycles late, about half a neighbour slot, so that one
    // wave's MFMA phase meets the other's VALU phase.
    if (stagger > 0 && (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 1)) {
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
    // S2X runs on scaled operands (conv1' x T, W2 x S, both powers of two: see k_bn_fold_moments); y2 comes out x S T
    const float unscale = kTwo ? scales[0] : kF16 ? 1.f / (Sw * Sd) : 1.f;
    constexpr bool kFrag = kF16 && !kTwo;                       // MLP2 keeps conv1's four A fragments in registers, MLP3 re-reads them from LDS
    // A workgroup walks the tile groups bid, bid + nblk, ... of its scene (round 4: the weights are staged once per workgroup, not once per
    // four tiles -- staging was 18 % / 10 % of a wave's time in MLP2 / MLP3, tools/ec_phases.py); its waves run their tiles independently
    // (no barrier inside the walk).  The next tile's own row is requested before this tile's statistics flush, which hides the round trip.
    const int ngroups = (N + 32 * kWaves - 1) / (32 * kWaves);
    // row 0's XYZ (the conditioning shift): read once -- the slot-loop statement clobbers "memory", so inside the walk the compiler would
    // re-read them (three scalar round trips) for every tile
    const float cx0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)x9m[0])));
    const float cx1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)x9m[1])));
    const float cx2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)x9m[2])));
    // Which tiles a workgroup walks.  walk_mode 0 (round 4): tile groups bid, bid + nblk, ...  walk_mode 1 (round 5, the engine's launches):
    // workgroup bid owns ONE contiguous range of the scene's tile groups and each of its waves a contiguous quarter of the range's tiles, so
    // that (a) a wave's consecutive tiles are consecutive rows -- in member order mostly the same cluster: the fused epilogue carries a
    // cluster's maxima from tile to tile and leaves one atomic per (run, channel) instead of one per (tile, channel): 16 MB of atomic traffic
    // per scene-launch in round 4 -- and (b) the ranges of the workgroups that share an XCD (linear workgroup id mod 8 = bid mod 8 when
    // nblk is a multiple of 8) are neighbours: an L2 sees one eighth of a scene's rows and its gathers, not all of them.
    int g0 = 0, glen = 0;
    if (walk_mode) {
        const int rho = (nblk & 7) == 0 ? (bid & 7) * (nblk >> 3) + (bid >> 3) : bid;
        g0 = (int)((long long)rho * ngroups / nblk);
        glen = (int)((long long)(rho + 1) * ngroups / nblk) - g0;
    }
    const int walk_n = walk_mode ? glen : (bid < ngroups ? (ngroups - bid + nblk - 1) / nblk : 0);
    auto tile_of = [&](int it, int w_) { return walk_mode ? g0 * kWaves + w_ * glen + it : (bid + it * nblk) * kWaves + w_; };
    // (the lane's identity comes in as arguments: captured from the outer scope it stayed alive -- in scratch -- across the slot-loop statement)
    auto own_row = [&](int it, int w_, int r_) { const int p_ = tile_of(it, w_) * 32 + r_; return (gptr<const float4>)(x9m + (size_t)(p_ < N ? p_ : 0) * 12); };
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
    if (walk_n > 0 && !(kAsm && kTwo)) { const gptr<const float4> xr = own_row(0, wave, r); q0 = xr[0]; q1 = xr[1]; q2 = xr[2]; }
    const int tid_outer = tid;
    // Round 5 (S2X: 84-108 bytes of scratch per lane -- the kernel's 17 MB of HBM writes per scene-launch were spill traffic, not atomics): what a
    // lane knows about itself is rebuilt from v_mbcnt (a volatile statement: not common-subexpression'd across the slot loop) and the wave's
    // number in an SGPR, instead of a copy of the thread id kept alive -- in scratch -- across the slot-loop statement
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto fresh_tid = [&]() {
        if constexpr (kAsm && kTwo) {
            int l_;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
            return wave_s * 64 + l_;
        } else {
            int t_ = tid_outer;
            asm volatile("" : "+v"(t_));
            return t_;
        }
    };
    if constexpr (kFused) { if (lane == 0) carry_c[wave] = -1; }
    // one atomic max per (cluster, channel): order-preserving integer view -- non-negative floats compare as ints, negative floats reversed as uints
    // (`c` is wave-uniform at every call: the row's address stays in SGPRs and the lane adds its channel as a 32-bit offset -- written on a
    // generic pointer the 64-bit lane address kept a zero register alive in scratch across the whole walk)
    auto cluster_max_out = [&](int c, float v, int ch) {
        v *= unscale;                                             // one multiply per (cluster, channel) instead of 32 per lane and tile
        const gptr<float> row = ext + (size_t)__builtin_amdgcn_readfirstlane(c) * ext_stride;
        const unsigned off = 4u * (unsigned)ch;
        // (as statements: from C++ the compiler builds a 64-bit lane address whose zero high half it parks in scratch for the whole walk)
        if (v >= 0.0f) asm volatile("global_atomic_smax %0, %1, %2" :: "v"(off), "v"(v), "s"(row) : "memory");
        else asm volatile("global_atomic_umin %0, %1, %2" :: "v"(off), "v"(v), "s"(row) : "memory");
    };
    for (int it = 0; it < walk_n; ++it) {
    // Everything a lane knows about itself is re-derived from the thread id here and again behind the slot loop: the statement of the
    // hand-scheduled loop leaves the compiler ten registers, and what is live across it goes to scratch and back (45 values per tile before)
    const int tid = fresh_tid();
    const int lane = tid & 63, wave = (kAsm && kTwo) ? wave_s : tid >> 6, r = lane & 31, half = lane >> 5;     // S2X: the wave's number stays scalar
    const int tile = tile_of(it, wave);
    const int pt = tile * 32 + r;
    const bool valid = pt < N;
    [[maybe_unused]] const float vmask = valid ? 1.f : 0.f;
    const int ptc = valid ? pt : 0;
    float4 nq0, nq1, nq2;                                       // the next tile's own row (requested behind the slot loop)
    constexpr bool kRowAtTop = kAsm && kTwo;                     // S2X: no row carried round the loop (it lived in scratch: see fresh_tid)
    if constexpr (kRowAtTop) { const gptr<const float4> xr = own_row(it, wave, r); q0 = xr[0]; q1 = xr[1]; q2 = xr[2]; }

    if (tile * 32 < N) {
        // x_i (9 of the 12 floats of the padded row)
        const float xi[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x};
        // B operands, k = 2s + half: the x_i half of the edge feature is the same for all K neighbours of a point, so its
        // contribution W1[:, 9:18] x_i (+ the folded BN1 shift) is evaluated ONCE into `base` and every slot's accumulator
        // starts from it (first MFMA: C = base, D = acc1) -- 10 instead of 18 conv1 MFMAs per slot
        const float xsel[5] = {half ? xi[1] : xi[0], half ? xi[3] : xi[2], half ? xi[5] : xi[4], half ? xi[7] : xi[6], half ? 0.f : xi[8]};
        // d = x_j - x_i: lanes 0-31 cut d0..d3, lanes 32-63 d4..d7, both d8 (struct Lds, a1p)
        const float xs[5] = {half ? xi[4] : xi[0], half ? xi[5] : xi[1], half ? xi[6] : xi[2], half ? xi[7] : xi[3], xi[8]};
        const float xs_s[5] = {xs[0] * Sd, xs[1] * Sd, xs[2] * Sd, xs[3] * Sd, xs[4] * Sd};        // kF16 only
        // the x_i half sees coordinates relative to row 0 (see "Conditioning" in the header); d below uses the raw ones
        const float xcen[5] = {half ? xi[1] - cx1 : xi[0] - cx0, half ? xi[3] : xi[2] - cx2, xsel[2], xsel[3], xsel[4]};
        f32x16 base[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (kTwo) {
                const float4* sp = reinterpret_cast<const float4*>(&lds.sh1r[t][half][0]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = sp[g];
                    base[t][4 * g] = v.x; base[t][4 * g + 1] = v.y; base[t][4 * g + 2] = v.z; base[t][4 * g + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) base[t][q] = 0.f;
            }
        }
#pragma unroll
        for (int s4 = 0; s4 < 2; ++s4) {
            const float4 wa = lds.a1x[0][s4][lane], wb = lds.a1x[1][s4][lane];
            const float* pa = &wa.x;
            const float* pb = &wb.x;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s_ = 4 * s4 + u;
                if (s_ < 5) {
                    base[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[u], xcen[s_], base[0], 0, 0, 0);
                    base[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pb[u], xcen[s_], base[1], 0, 0, 0);
                }
            }
        }

        asm volatile("" :: "v"(base[0]), "v"(base[1]));
        stamp.mark(1);                                        // own row, base accumulator
        float stat_s[32], stat_q[32];
        f32x16 best[2];
        // Round 5 (MLP2 on fp16 pieces): the SUMS over the K slots are taken on the input side -- conv1 is linear: sum_j y_j = (K - 1) base +
        // (base + conv1(sum_j d_j)) -- five adds per slot into dsum instead of 32 on the outputs, one more conv1 behind the last slot
        // (`sums_from_dsum` below; the hand-scheduled loop does the same inside its statement: gen_edgeconv_asm.py, sums_s1x)
        constexpr bool kSumsLinear = kF16 && !kTwo;
        [[maybe_unused]] float dsum[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 32; ++q) { stat_s[q] = 0.f; stat_q[q] = 0.f; }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) best[t][q] = -INFINITY;

        const gptr<const int32_t> krow = knn + (size_t)ptc * K;
        // the fp16 conv1's four A fragments stay in registers for the whole tile (MLP2 has the room: no conv2 accumulators)
        u32x4 fr16[4] = {};
        if (kFrag && !kAsm) { fr16[0] = lds.a1p[0][0][lane]; fr16[1] = lds.a1p[0][1][lane]; fr16[2] = lds.a1p[1][0][lane]; fr16[3] = lds.a1p[1][1][lane]; }
        // one neighbour slot, given the neighbour's row
        auto slot_body = [&](const float4& n0, const float4& n1, const float4& n2) {
            // S2X holds 64 statistics + 32 maxima + 48 accumulator registers: do not let the compiler also park the 82
            // loop-invariant A fragments in VGPRs (it then spills ~60 of them to scratch: 0.56 vs 0.48 ms per MLP3 at
            // 150k points); re-read them from LDS per slot
            if (REREAD_A) asm volatile("" ::: "memory");
            // conv1 on the bf16 pipe: the lane's five d values cut into three bf16 pieces by truncation (d = x1 + x2 + x3), the six products
            // that matter packed along K (struct Lds): 8 MFMAs of 32 cycles on top of the point's base accumulator
            f32x16 acc1[2];
            if constexpr (kF16) {
                // d scaled by Sd (fma(n, Sd, -x_i Sd) = (n - x_i) Sd exactly: Sd is a power of two), cut into fp16 hi / lo, three products
                // packed into two MFMAs per output tile (see the image above).  n0 = the 16 bytes of the neighbour's row this lane's half
                // works on (the loads below fetch exactly those), n2.x = channel 8
                const float nv[5] = {n0.x, n0.y, n0.z, n0.w, n2.x};
                float ds[6];
#pragma unroll
                for (int q = 0; q < 5; ++q) ds[q] = __builtin_fmaf(nv[q], Sd, -xs_s[q]);
                ds[5] = 0.f;
                if constexpr (kSumsLinear) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) dsum[q] += ds[q];
                }
                unsigned int ph[3], pl_[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const f32x2 v = {ds[2 * u], ds[2 * u + 1]};
                    const f16x2 hi = __builtin_convertvector(v, f16x2);
                    const f16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x2), f16x2);      // v - hi is exact in fp32
                    ph[u] = __builtin_bit_cast(unsigned int, hi);
                    pl_[u] = __builtin_bit_cast(unsigned int, lo);
                }
                const f16x8 x0 = __builtin_bit_cast(f16x8, u32x4{ph[0], ph[1], pl_[0], pl_[1]});      // (d_hi | d_lo) of the lane's four values
                const f16x8 x1 = __builtin_bit_cast(f16x8, u32x4{ph[0], ph[1], ph[2], pl_[2]});       // (d_hi | d8_hi, d8_lo)
                const f16x8 wa0 = __builtin_bit_cast(f16x8, kFrag ? fr16[0] : lds.a1p[0][0][lane]), wb0 = __builtin_bit_cast(f16x8, kFrag ? fr16[1] : lds.a1p[0][1][lane]);
                const f16x8 wa1 = __builtin_bit_cast(f16x8, kFrag ? fr16[2] : lds.a1p[1][0][lane]), wb1 = __builtin_bit_cast(f16x8, kFrag ? fr16[3] : lds.a1p[1][1][lane]);
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa1, x1, base[0], 0, 0, 0);           // the smaller terms first
                acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb1, x1, base[1], 0, 0, 0);
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa0, x0, acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb0, x0, acc1[1], 0, 0, 0);
            } else {
                const float nv[5] = {half ? n1.x : n0.x, half ? n1.y : n0.y, half ? n1.z : n0.z, half ? n1.w : n0.w, n2.x};
                unsigned int v1[5], v2[5], v3[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const float v = nv[q] - xs[q];
                    const float r = v - __uint_as_float(__float_as_uint(v) & 0xffff0000u);
                    const float c = r - __uint_as_float(__float_as_uint(r) & 0xffff0000u);
                    v1[q] = __float_as_uint(v); v2[q] = __float_as_uint(r); v3[q] = __float_as_uint(c);
                }
                // units of two bf16 = the top halves of two registers (v_perm_b32)
                const unsigned int a1 = __builtin_amdgcn_perm(v1[1], v1[0], 0x07060302u), b1 = __builtin_amdgcn_perm(v1[3], v1[2], 0x07060302u);
                const unsigned int a2 = __builtin_amdgcn_perm(v2[1], v2[0], 0x07060302u), b2 = __builtin_amdgcn_perm(v2[3], v2[2], 0x07060302u);
                const unsigned int a3 = __builtin_amdgcn_perm(v3[1], v3[0], 0x07060302u), b3 = __builtin_amdgcn_perm(v3[3], v3[2], 0x07060302u);
                const unsigned int c1 = v1[4] >> 16, c2 = v2[4] >> 16, c3 = v3[4] >> 16;
                const bf16x8 x0 = __builtin_bit_cast(bf16x8, u32x4{a1, b1, a2, b2});       // (x1 | x2)
                const bf16x8 x1 = __builtin_bit_cast(bf16x8, u32x4{a1, b1, a3, b3});       // (x1 | x3)
                const bf16x8 x2 = __builtin_bit_cast(bf16x8, u32x4{a2, b2, a1, b1});       // (x2 | x1)
                const bf16x8 x3 = __builtin_bit_cast(bf16x8, u32x4{c1, c2, c3, c1});       // d8: x1, x2, x3, x1
                const bf16x8 wa0 = __builtin_bit_cast(bf16x8, lds.a1p[0][0][lane]), wb0 = __builtin_bit_cast(bf16x8, lds.a1p[0][1][lane]);
                const bf16x8 wa1 = __builtin_bit_cast(bf16x8, lds.a1p[1][0][lane]), wb1 = __builtin_bit_cast(bf16x8, lds.a1p[1][1][lane]);
                const bf16x8 wa2 = __builtin_bit_cast(bf16x8, lds.a1p[2][0][lane]), wb2 = __builtin_bit_cast(bf16x8, lds.a1p[2][1][lane]);
                const bf16x8 wa3 = __builtin_bit_cast(bf16x8, lds.a1p[3][0][lane]), wb3 = __builtin_bit_cast(bf16x8, lds.a1p[3][1][lane]);
                // smallest terms first: (w2 x2, w3 x1), (w2 x1, w1 x3), d8's six, (w1 x1, w1 x2)
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa2, x2, base[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb2, x2, base[1], 0, 0, 0);
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa1, x1, acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb1, x1, acc1[1], 0, 0, 0);
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa3, x3, acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb3, x3, acc1[1], 0, 0, 0);
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa0, x0, acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb0, x0, acc1[1], 0, 0, 0);
            }
            if (!kTwo) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        // rows past N (the last tile) are masked when the lanes are summed, not here; register pairs written out so that
                        // the packed adds / fmas sit on aligned pairs (left to the vectoriser they came out shifted by one, with a move
                        // per operand)
                        const f32x2 y = {acc1[t][q], acc1[t][q + 1]};
                        const f32x2 q2 = __builtin_elementwise_fma(y, y, f32x2{stat_q[16 * t + q], stat_q[16 * t + q + 1]});
                        if constexpr (!kSumsLinear) {
                            const f32x2 s2 = f32x2{stat_s[16 * t + q], stat_s[16 * t + q + 1]} + y;
                            stat_s[16 * t + q] = s2.x; stat_s[16 * t + q + 1] = s2.y;
                        }
                        stat_q[16 * t + q] = q2.x; stat_q[16 * t + q + 1] = q2.y;
                        best[t][q] = fmaxf(best[t][q], y.x); best[t][q + 1] = fmaxf(best[t][q + 1], y.y);
                    }
            } else {
                // LeakyReLU(BN1(.)) in place -> B operand of conv2
                // (the 0.2 x as v_pk_mul_f32, two values per issue slot: every VALU instruction costs a SIMD ~4 cycles whatever it does,
                // tools/micro/issue_rates.hip, and this kernel is bound by its instruction count)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        const f32x2 v = {acc1[t][q], acc1[t][q + 1]};
                        const f32x2 w = v * f32x2{0.2f, 0.2f};
                        acc1[t][q] = fmaxf(v.x, w.x); acc1[t][q + 1] = fmaxf(v.y, w.y);
                    }
                // conv2 on the fp16 matrix pipe (see the header): per 16-deep k block the 8 accumulator registers of this lane are cut
                // into two fp16 pieces each (round-to-nearest: v_cvt_pk_f16_f32) and meet the pre-split weights in three MFMAs per
                // output tile; the two output tiles are independent accumulator chains, interleaved
                f32x16 acc2[2];
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc2[ot][q] = 0.f;
                // The wave issues in order, and an MFMA that finds the matrix pipe busy blocks everything behind it: with the split
                // of a k block written in front of its MFMAs, split and MFMAs alternate and the pipe idles during every split.
                // The split of block kb + 1 only needs conv1's accumulator, so it is computed WHILE block kb's MFMAs run: one MFMA,
                // then four of the next block's VALU instructions in its 32-cycle shadow, pinned with sched_group_barrier
                // (0x8 = MFMA, 0x2 = VALU).  (Pipelining further -- the next slot's conv1 under this slot's statistics, the d split
                // under the last k block -- needs both accumulators live across the loop edge: 100-300 B of scratch per lane at
                // 2 waves/SIMD, and 1 wave/SIMD is 1.5x slower; measured, not kept.)
                // ALL of the operand's pieces first (one VALU phase), then the 24 MFMAs back to back (one matrix phase): with the cut of block
                // kb + 1 interleaved into block kb's MFMAs (round 2) the two waves of a SIMD were both in mixed MFMA / VALU streams all the
                // time and the matrix pipe sat half idle; pure phases let one wave's VALU phase run beside the other's MFMA burst -- the way
                // the statistics behind the last MFMA always did (leaving them out changed nothing, leaving the cut out saved 14 %).
                unsigned int xh[4][4], xl[4][4];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const f32x2 v = {acc1[kb >> 1][8 * (kb & 1) + 2 * jj], acc1[kb >> 1][8 * (kb & 1) + 2 * jj + 1]};
                        const f16x2 hi = __builtin_convertvector(v, f16x2);
                        const f16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x2), f16x2);     // v - hi is exact in fp32
                        xh[kb][jj] = __builtin_bit_cast(unsigned int, hi);
                        xl[kb][jj] = __builtin_bit_cast(unsigned int, lo);
                    }
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const f16x8 x1 = __builtin_bit_cast(f16x8, u32x4{xh[kb][0], xh[kb][1], xh[kb][2], xh[kb][3]});
                    const f16x8 x2 = __builtin_bit_cast(f16x8, u32x4{xl[kb][0], xl[kb][1], xl[kb][2], xl[kb][3]});
                    const f16x8 wa1 = __builtin_bit_cast(f16x8, lds.a2h[0][0][kb][lane]), wb1 = __builtin_bit_cast(f16x8, lds.a2h[0][1][kb][lane]);
                    const f16x8 wa2 = __builtin_bit_cast(f16x8, lds.a2h[1][0][kb][lane]), wb2 = __builtin_bit_cast(f16x8, lds.a2h[1][1][kb][lane]);
                    // smallest terms first
                    acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa2, x1, acc2[0], 0, 0, 0);
                    acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb2, x1, acc2[1], 0, 0, 0);
                    acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa1, x2, acc2[0], 0, 0, 0);
                    acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb1, x2, acc2[1], 0, 0, 0);
                    acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa1, x1, acc2[0], 0, 0, 0);
                    acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb1, x1, acc2[1], 0, 0, 0);
                }
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float z = acc2[ot][q];
                        stat_s[16 * ot + q] += z;
                        stat_q[16 * ot + q] = __builtin_fmaf(z, z, stat_q[16 * ot + q]);
                        best[ot][q] = fmaxf(best[ot][q], z);
                    }
            }
        };
        // software pipeline: the next slot's neighbour row is requested before this slot's MFMAs start.  (Hoisting the tile's
        // first loads -- own row, first ids, first neighbour row: three dependent round trips -- in front of the weight staging
        // and the base MFMAs measured SLOWER, S1X 421 -> 457 us: the co-resident wave of the other workgroup already covers them.)
        auto load_row = [&](int nb, float4& r0, float4& r1, float4& r2) {
            const gptr<const float4> xq = (gptr<const float4>)(x9m + (size_t)nb * 12);
            r0 = xq[0]; r1 = xq[1]; r2 = xq[2];
        };
        if constexpr (kF16) {
            // kNB row buffers that take turns, kNB - 1 neighbour rows in flight (MLP2: three; its waves sat in s_waitcnt 37 % of their cycles with one
            // -- a gather of 64 x 2 scattered lines misses L2 more often than not once eight scenes share it; MLP3 has no registers left for more
            // than one).  A lane needs the four values of its half (d0..d3 | d4..d7) and d8 of the neighbour's row: one 16-byte and one 4-byte load.
            // The requests of a slot are written as instructions: left to the compiler they are SUNK to their first use (register pressure), which
            // turns the lookahead into a memory round trip per slot; with the buffers handed over at the loop edge the register allocator copied
            // freshly loaded registers right behind their load.  `arrive` waits until only the younger requests are outstanding (loads return in
            // order) and ties the registers to the wait.  The K neighbour ids of the lane's row are parked in the wave's strip of the epilogue's
            // stage area (unused until the tile's maxima are written there): a request must not depend on a load that is itself in flight.
            using f32x4 = __attribute__((ext_vector_type(4))) float;
            constexpr int kNB = kTwo ? 2 : 4;
            int* idl = reinterpret_cast<int*>(&stage_or_ids[wave][0]);
            if constexpr (!kAsm)                                  // the hand-scheduled loops read the ids from the kNN table themselves
                for (int j = 0; j < K; ++j) idl[j * 64 + lane] = krow[j];
            constexpr bool slots_done = kAsm;                       // the kAsm kernels are only launched with K == 20 (the host checks)
            if constexpr (kAsm) {
                {
                    f32x16 ss0, ss1, sq0, sq1, bb0, bb1;
                    const float sd_u = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, Sd)));
                    // base -> the wave's stage strip, [t * 4 + g][lane] float4 = base[t][4 g .. 4 g + 3]: conv1's C operand.  MLP3 re-reads it every
                    // slot; MLP2 reads it (and its four A fragments) ONCE, at the head of the statement -- handing 48 registers over as operands
                    // made the compiler copy every one of them into place (~300 v_mov per tile)
                    float4* bl = reinterpret_cast<float4*>(&stage_or_ids[wave][0]);
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            bl[(t * 4 + g) * 64 + lane] = make_float4(base[t][4 * g], base[t][4 * g + 1], base[t][4 * g + 2], base[t][4 * g + 3]);
                    // The statement's operands are derived HERE, behind the base stores, from laundered copies of the thread id and the own row:
                    // computed earlier they sit in the registers base is built in, and the compiler parks them in scratch and fetches them back
                    // (7 scratch round trips in front of every tile's slot loop)
                    int tid_c = fresh_tid();
                    using f32x4 = __attribute__((ext_vector_type(4))) float;
                    f32x4 c0 = {q0.x, q0.y, q0.z, q0.w}, c1 = {q1.x, q1.y, q1.z, q1.w};
                    float c2x = q2.x;
                    asm volatile("" : "+v"(tid_c), "+v"(c0), "+v"(c1), "+v"(c2x));
                    const int lane_c = tid_c & 63, half_c = lane_c >> 5, r_c = lane_c & 31, wave_c = (kAsm && kTwo) ? wave_s : tid_c >> 6;
                    const int pt_c = tile_of(it, wave_c) * 32 + r_c;
                    const unsigned koff = (unsigned)(pt_c < N ? pt_c : 0) * 80u;    // the lane's row of the kNN table (K = 20 ids)
                    const unsigned l16 = 16u * (unsigned)half_c;
                    const float xs_c[5] = {(half_c ? c1.x : c0.x) * Sd, (half_c ? c1.y : c0.y) * Sd, (half_c ? c1.z : c0.z) * Sd, (half_c ? c1.w : c0.w) * Sd,
                                           c2x * Sd};
                    const unsigned a_base = (unsigned)(size_t)(SG_LDS const float4*)(reinterpret_cast<float4*>(&stage_or_ids[wave_c][0]) + lane_c);
                    // the A fragments stay in LDS (MLP3: they pass through a ring of four register tuples); a2h sits 12288 B behind a1p (struct Lds)
                    static_assert(offsetof(Lds, a2h) - offsetof(Lds, a1p) == 12288, "edgeconv_slots_gen.h addresses conv2's fragments relative to conv1's");
                    const unsigned a_frag = (unsigned)(size_t)(SG_LDS const u32x4*)(&lds.a1p[0][0][lane_c]);
                    if constexpr (!kTwo) {
                        // The PLAIN variant of the generated loop (sums of squares as 32 v_fma_f32 per slot), not the packed one (16 v_pk_fma_f32: ~0.2 us per
                        // scene faster): packed fp32 arithmetic is not dependable on the MI355X boxes of this pool while MFMA kernels start and stop on the SIMD
                        //  -- a lost v_pk_fma here is one y^2 missing from a BatchNorm variance, and a scene's labels then depend
                        // on the run.  -DSG_EC_S1X_PACKED builds the packed loop (measurements only).
#ifdef SG_EC_S1X_PACKED
                        asm volatile(SG_EC_S1X_SLOTS_PK
#else
                        asm volatile(SG_EC_S1X_SLOTS
#endif
                                     : "=&" SG_EC_S1X_STAT_S0(ss0), "=&" SG_EC_S1X_STAT_S1(ss1), "=&" SG_EC_S1X_STAT_Q0(sq0), "=&" SG_EC_S1X_STAT_Q1(sq1),
                                       "=&" SG_EC_S1X_BEST0(bb0), "=&" SG_EC_S1X_BEST1(bb1)
                                     : [x9m] "s"(x9m), [knn] "s"(knn), [sd] "s"(sd_u), [koff] "v"(koff), [l16] "v"(l16), [base] "v"(a_base), [frag] "v"(a_frag),
                                       [xs0] "v"(xs_c[0]), [xs1] "v"(xs_c[1]), [xs2] "v"(xs_c[2]), [xs3] "v"(xs_c[3]), [xs4] "v"(xs_c[4])
#ifdef SG_EC_S1X_PACKED
                                     : "memory", SG_EC_S1X_SLOTS_PK_CLOBBERS);
#else
                                     : "memory", SG_EC_S1X_SLOTS_CLOBBERS);
#endif
                    } else {
                        const unsigned long long c02 = 0x3e4ccccd3e4ccccdull;     // LeakyReLU's 0.2f twice: v_pk_mul_f32 takes no literal
                        asm volatile(SG_EC_S2X_SLOTS
                                     : "=&" SG_EC_S2X_STAT_S0(ss0), "=&" SG_EC_S2X_STAT_S1(ss1), "=&" SG_EC_S2X_STAT_Q0(sq0), "=&" SG_EC_S2X_STAT_Q1(sq1),
                                       "=&" SG_EC_S2X_BEST0(bb0), "=&" SG_EC_S2X_BEST1(bb1)
                                     : [c02] "s"(c02), [x9m] "s"(x9m), [knn] "s"(knn), [sd] "s"(sd_u), [koff] "v"(koff), [l16] "v"(l16), [base] "v"(a_base), [frag] "v"(a_frag),
                                       [xs0] "v"(xs_c[0]), [xs1] "v"(xs_c[1]), [xs2] "v"(xs_c[2]), [xs3] "v"(xs_c[3]), [xs4] "v"(xs_c[4])
                                     : "memory", SG_EC_S2X_SLOTS_CLOBBERS);
                    }
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        stat_s[q] = ss0[q]; stat_s[16 + q] = ss1[q];
                        stat_q[q] = sq0[q]; stat_q[16 + q] = sq1[q];
                    }
                    best[0] = bb0; best[1] = bb1;
                }
            }
            if constexpr (!slots_done) {
                // (`cur` = the row the coming slot works on: re-defined by the request so that the slot's code cannot be scheduled in front of
                // it; `done` = a statistic the finished slot wrote: re-defined by the wait so that the wait cannot be scheduled in front of it)
                auto request = [&](int nb, f32x4& r0, float& e8, f32x4& cur) {
                    const gptr<const float> row = x9m + (size_t)nb * 12;
                    const gptr<const float> p16 = row + 4 * half, p4 = row + 8;
                    asm volatile("global_load_dwordx4 %0, %2, off" : "=v"(r0), "+v"(cur) : "v"(p16));
                    asm volatile("global_load_dword %0, %1, off" : "=v"(e8) : "v"(p4));
                };
                auto arrive = [&](f32x4& r0, float& e8, float& done) {
                    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r0), "+v"(e8), "+v"(done) : "n"(2 * (kNB - 2)));
                };
                f32x4 rb[kNB];
                float eb[kNB];
                f32x4 none = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < kNB - 1; ++u) request(idl[min(u, K - 1) * 64 + lane], rb[u], eb[u], none);
                int idn = idl[min(kNB - 1, K - 1) * 64 + lane];          // the id of the next request, read a slot ahead of it
                for (int j = 0; j < K; j += kNB) {
#pragma unroll
                    for (int u = 0; u < kNB; ++u) {
                        arrive(rb[u], eb[u], stat_q[31]);
                        request(idn, rb[(u + kNB - 1) % kNB], eb[(u + kNB - 1) % kNB], rb[u]);      // slot j + u + kNB - 1 (beyond the last slot: re-reads it)
                        idn = idl[min(j + u + kNB, K - 1) * 64 + lane];
                        if (u == 0 || j + u < K) slot_body(make_float4(rb[u].x, rb[u].y, rb[u].z, rb[u].w), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(eb[u], 0.f, 0.f, 0.f));
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb[0]), "+v"(rb[1]), "+v"(eb[0]), "+v"(eb[1]), "+v"(stat_q[31]));       // nothing may still be in flight into registers the code below reuses
                if constexpr (kNB > 2) asm volatile("" : "+v"(rb[kNB - 2]), "+v"(rb[kNB - 1]), "+v"(eb[kNB - 2]), "+v"(eb[kNB - 1]));
                if constexpr (kSumsLinear) {
                    // sums_from_dsum: dsum cut like a slot's d, the slot's four MFMAs on top of base
                    unsigned int ph[3], pl_[3];
                    // (scaled by 2^-5 first, exactly: one slot's scaled d fits fp16, twenty of them -- a row padded with point 0 repeats one far
                    // neighbour, model.py:513 -- need not; acc = base + conv1(dsum / 32), sum = 32 acc + (K - 32) base)
                    const float dq[6] = {dsum[0] * 0.03125f, dsum[1] * 0.03125f, dsum[2] * 0.03125f, dsum[3] * 0.03125f, dsum[4] * 0.03125f, 0.f};
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const f32x2 v = {dq[2 * u], dq[2 * u + 1]};
                        const f16x2 hi = __builtin_convertvector(v, f16x2);
                        const f16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x2), f16x2);
                        ph[u] = __builtin_bit_cast(unsigned int, hi);
                        pl_[u] = __builtin_bit_cast(unsigned int, lo);
                    }
                    const f16x8 x0 = __builtin_bit_cast(f16x8, u32x4{ph[0], ph[1], pl_[0], pl_[1]});
                    const f16x8 x1 = __builtin_bit_cast(f16x8, u32x4{ph[0], ph[1], ph[2], pl_[2]});
                    const f16x8 wa0 = __builtin_bit_cast(f16x8, kFrag ? fr16[0] : lds.a1p[0][0][lane]), wb0 = __builtin_bit_cast(f16x8, kFrag ? fr16[1] : lds.a1p[0][1][lane]);
                    const f16x8 wa1 = __builtin_bit_cast(f16x8, kFrag ? fr16[2] : lds.a1p[1][0][lane]), wb1 = __builtin_bit_cast(f16x8, kFrag ? fr16[3] : lds.a1p[1][1][lane]);
                    f32x16 accs[2];
                    accs[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa1, x1, base[0], 0, 0, 0);
                    accs[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb1, x1, base[1], 0, 0, 0);
                    accs[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa0, x0, accs[0], 0, 0, 0);
                    accs[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb0, x0, accs[1], 0, 0, 0);
                    const float km32 = (float)(K - 32);
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < 16; ++q) stat_s[16 * t + q] = __builtin_fmaf(accs[t][q], 32.0f, km32 * base[t][q]);
                }
            }
            __builtin_amdgcn_wave_barrier();                      // the ids are dead: the strip is the epilogue's now
        } else {
            int nb_next = krow[0];
            float4 p0, p1, p2;
            load_row(nb_next, p0, p1, p2);
            nb_next = K > 1 ? krow[1] : 0;
            for (int j = 0; j < K; ++j) {
                const float4 n0 = p0, n1 = p1, n2 = p2;
                if (j + 1 < K) {
                    load_row(nb_next, p0, p1, p2);
                    if (j + 2 < K) nb_next = krow[j + 2];
                }
                slot_body(n0, n1, n2);
            }
        }

        asm volatile("" :: "v"(stat_q[31]), "v"(best[0]));
        stamp.mark(2);                                        // the slot loop
        const int tid = fresh_tid();
        const int lane = tid & 63, wave = (kAsm && kTwo) ? wave_s : tid >> 6, r = lane & 31, half = lane >> 5;
        const int tile = tile_of(it, wave);
        const int pt = tile * 32 + r;
        const bool valid = pt < N;
        const float vmask = valid ? 1.f : 0.f;
        const int ptc = valid ? pt : 0;
        if (!kRowAtTop && it + 1 < walk_n) { const gptr<const float4> xn = own_row(it + 1, wave, r); nq0 = xn[0]; nq1 = xn[1]; nq2 = xn[2]; }
        int myc = 0;
        if constexpr (kFused) myc = cluster_of_pos[ptc];      // asked for here, needed behind the statistics flush
        // sum over the 32 rows of each half on the DPP path (wave_ops.h; `__shfl_xor` is an LDS round trip per step on gfx950): quads,
        // 16-lane rows, then row_bcast15 folds row 0 into row 1 and row 2 into row 3 -- lanes 31 and 63 hold their half's sums.  They park
        // them in LDS and lane = channel adds them into the wave's fp64 accumulators: two parallel read-modify-writes per lane instead of
        // 64 serial ones by two lanes (each an LDS round trip: the flush was 12 % of a wave's time in MLP2; same values, same order)
        float* fs = &flush_sums[wave][0];
        if (tile * 32 + 32 > N) {                              // rows past N (the scene's last tile only) do not count
#pragma unroll
            for (int q = 0; q < 32; ++q) { stat_s[q] *= vmask; stat_q[q] *= vmask; }
        }
#ifndef SG_EC_FLUSH_CHAINS
        // A HALVING reduction over the 32 lanes of each half (round 4): a lane does not carry all 64 values through all five steps -- at every
        // step the two lanes of a pair split what is left, each keeps one half (its own half + the partner's copy of it) and passes the other
        // on: 32 + 16 + 8 + 4 + 2 additions instead of 5 x 64, and at the end every lane owns TWO of its half's 64 sums.  The pairings are
        // the ones the DPP modifiers of gfx9 offer as involutions: row_mirror (i <-> 15 - i, split by lane bit 3), row_half_mirror (i <-> 7 - i,
        // bit 2), quad permutes xor 1 / xor 2 (bits 0, 1), then lane xor 16 through the LDS crossbar (bit 4).  (The 64 independent chains this
        // replaces were 320 v_add_f32_dpp + 64 v_mov_dpp + 64 stores by two lanes per tile: with the flush compiled out the kernels run 3.4 /
        // 3.0 us per scene faster.)  The fp32 sums of a tile are associated differently than before; they enter fp64 right behind.
        {
            const bool k3 = lane & 8, k2 = lane & 4, k0 = lane & 1, k1 = lane & 2, k4 = lane & 16;
            float w32[32], w16[16], w8[8], w4[4], w2[2];
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const float keep = k3 ? stat_q[q] : stat_s[q], send = k3 ? stat_s[q] : stat_q[q];
                w32[q] = keep + sgw::dpp_f<0x140>(send, send);                      // row_mirror
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float keep = k2 ? w32[q + 16] : w32[q], send = k2 ? w32[q] : w32[q + 16];
                w16[q] = keep + sgw::dpp_f<0x141>(send, send);                      // row_half_mirror
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float keep = k0 ? w16[q + 8] : w16[q], send = k0 ? w16[q] : w16[q + 8];
                w8[q] = keep + sgw::dpp_f<sgw::kQuadXor1>(send, send);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float keep = k1 ? w8[q + 4] : w8[q], send = k1 ? w8[q] : w8[q + 4];
                w4[q] = keep + sgw::dpp_f<sgw::kQuadXor2>(send, send);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float keep = k4 ? w4[q + 2] : w4[q], send = k4 ? w4[q] : w4[q + 2];
                // lane xor 16 as a swizzle (bit mode: and 0x1f, xor 0x10): no address register -- __shfl_xor's lane id is loop-invariant to the
                // compiler, which kept it in scratch across the slot-loop statement
                w2[q] = keep + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x401F));
            }
            // which two: value index Q = 32 k3 + 16 k2 + 8 k0 + 4 k1 + 2 k4 + {0, 1}; Q < 32 = sum of accumulator register Q, else sum of squares of Q - 32
            const int reg = (k2 ? 16 : 0) + (k0 ? 8 : 0) + (k1 ? 4 : 0) + (k4 ? 2 : 0);
            const int at = (k3 ? 64 : 0) + acc_channel(reg >> 4, reg & 15, half);     // registers reg, reg + 1 -> channels at, at + 1
            fs[at] = w2[0];
            fs[at + 1] = w2[1];
        }
#else
        // the 64 reductions are independent chains: written without a branch in between, the compiler interleaves them and the two wait
        // states a DPP read needs behind the write of its source cost nothing (one exec-mask region per value left 99 s_nop per tile)
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            float s = stat_s[q], v = stat_q[q];
            s += sgw::dpp_f<sgw::kQuadXor1>(s, s); v += sgw::dpp_f<sgw::kQuadXor1>(v, v);
            s += sgw::dpp_f<sgw::kQuadXor2>(s, s); v += sgw::dpp_f<sgw::kQuadXor2>(v, v);
            s += sgw::dpp_f<sgw::kRowRor4>(s, s);  v += sgw::dpp_f<sgw::kRowRor4>(v, v);
            s += sgw::dpp_f<sgw::kRowRor8>(s, s);  v += sgw::dpp_f<sgw::kRowRor8>(v, v);
            s += sgw::dpp_f<sgw::kRowBcast15, 0xA>(0.f, s); v += sgw::dpp_f<sgw::kRowBcast15, 0xA>(0.f, v);
            stat_s[q] = s; stat_q[q] = v;
        }
        if (r == 31) {
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int ch = acc_channel(q >> 4, q & 15, half);
                fs[ch] = stat_s[q];
                fs[64 + ch] = stat_q[q];
            }
        }
#endif
        __builtin_amdgcn_wave_barrier();
        {
            int ub = __builtin_bit_cast(int, unscale);
            if constexpr (kAsm && kTwo) asm volatile("" : "+s"(ub));    // keeps the two conversions inside the loop (hoisted, the doubles lived in scratch)
            const double ud = (double)__builtin_bit_cast(float, ub);
            lds.acc[wave][lane] += (double)fs[lane] * ud;                                  // power of two: exact
            lds.acc[wave][64 + lane] += (double)fs[64 + lane] * (ud * ud);
        }
        __builtin_amdgcn_wave_barrier();
        stamp.mark(3);                                        // statistics flush
        if constexpr (!kFused) {
            if (valid) {
                // E = max_j y'_j : 4 consecutive channels per float4 store
                // the lane's BYTE offset in 32 bits (N <= 2^20 rows of 256 bytes) on top of the uniform base: the stores take the base from SGPRs --
                // a 64-bit lane address kept its zero high half alive, in scratch, across the slot-loop statement
                const gptr<char> orow = (gptr<char>)ext + ((unsigned)pt * 256u + 16u * (unsigned)half);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 v = make_float4(best[t][4 * g] * unscale, best[t][4 * g + 1] * unscale, best[t][4 * g + 2] * unscale,
                                                     best[t][4 * g + 3] * unscale);
                        *(gptr<float4>)(orow + 128 * t + 32 * g) = v;
                    }
            }
        } else {
            // Fused point -> cluster max (the scene engine): the consumer of E is max over each cluster's rows of LReLU(|a| E + b'),
            // and that map is monotone, so the cluster maximum of E itself is enough -- the [N,64] array (38 MB per scene and
            // layer, written here and read back by the segment-max kernel) never exists.  Rows are in member order (a tile spans
            // one cluster, sometimes two or three): the wave parks its 32 x 64 maxima in LDS, then lane = channel walks the rows
            // and leaves ONE atomic max per (cluster, channel) -- into the columns of the layer's feature matrix that the fill
            // kernel set to -inf (`ext` = cat + gm_D, stride Dcat); k_cluster_affine applies the activation once the fold is known.
            float* st = &stage_or_ids[wave][0];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 16; ++q) st[r * 65 + acc_channel(t, q, half)] = best[t][q];     // x unscale (> 0) behind the maximum: monotone, same bits
            __builtin_amdgcn_wave_barrier();
            const int rows_here = min(32, N - tile * 32);
            const int cfirst = __builtin_amdgcn_readfirstlane(myc);
            // the run carried over from the wave's previous tile: its cluster (wave-uniform) and, per channel, its maximum so far
            const int cc = __builtin_amdgcn_readfirstlane(carry_c[wave]);
            if (rows_here == 32 && __builtin_amdgcn_ballot_w64(myc != cfirst) == 0) {
                // the usual case -- all 32 rows in one cluster: the channel's 32 values are requested together and folded by a tree
                // (the walk below is a chain of 32 dependent LDS round trips)
                float col[32];
#pragma unroll
                for (int rr = 0; rr < 32; ++rr) col[rr] = st[rr * 65 + lane];
#pragma unroll
                for (int w_ = 16; w_ >= 1; w_ >>= 1)
#pragma unroll
                    for (int rr = 0; rr < w_; ++rr) col[rr] = fmaxf(col[rr], col[rr + w_]);
                if (cc == cfirst) carry_m[wave][lane] = fmaxf(carry_m[wave][lane], col[0]);      // the run goes on
                else {
                    if (cc >= 0) cluster_max_out(cc, carry_m[wave][lane], lane);
                    carry_m[wave][lane] = col[0];
                    if (lane == 0) carry_c[wave] = cfirst;
                }
            } else {
                int cprev = cc >= 0 ? cc : cfirst;
                float m = cc >= 0 ? carry_m[wave][lane] : -INFINITY;
                for (int rr = 0; rr < rows_here; ++rr) {
                    const int cr = __builtin_amdgcn_readlane(myc, rr);
                    const float v = st[rr * 65 + lane];
                    if (cr != cprev) { cluster_max_out(cprev, m, lane); m = v; cprev = cr; }
                    else m = fmaxf(m, v);
                }
                carry_m[wave][lane] = m;
                if (lane == 0) carry_c[wave] = cprev;
            }
            __builtin_amdgcn_wave_barrier();                    // the strip is the next tile's
        }
    }
    stamp.mark(4);                                            // maxima out (store / cluster maxima)
    if constexpr (!kRowAtTop) { q0 = nq0; q1 = nq1; q2 = nq2; }
    }   // tile groups
    const int tid_end = fresh_tid();                              // (not the outer copy: that one would live in scratch across the walk)
    if constexpr (kFused) {                                       // the wave's last run
        __builtin_amdgcn_wave_barrier();
        const int wave_e = (kAsm && kTwo) ? wave_s : tid_end >> 6, lane_e = tid_end & 63;
        const int cc = __builtin_amdgcn_readfirstlane(carry_c[wave_e]);
        if (cc >= 0) cluster_max_out(cc, carry_m[wave_e][lane_e], lane_e);
    }

    __syncthreads();
    if (tid_end < 128) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) s += lds.acc[w][tid_end];
        partial[(size_t)bid * 128 + tid_end] = s;
    }
    stamp.mark(5);                                            // barrier + partial sums
}
template <int MODE, bool REREAD_A = false, bool kF16 = false>
__global__ __launch_bounds__(64 * kWaves, 8 / kWaves) void k_edgeconv(const float* __restrict__ x9m, const int32_t* __restrict__ knn, int N, int K,
                                                          const float* __restrict__ w1, const float* __restrict__ shift1,
                                                          const u32x4* __restrict__ w2img, const float* __restrict__ scales,
                                                          const float* __restrict__ gamma_last,
                                                          float* __restrict__ ext, double* __restrict__ partial,
                                                          const unsigned int* __restrict__ range_bits = nullptr) {
    using sg::as_global;
    edgeconv_body<MODE, REREAD_A, false, kF16>(as_global(x9m), as_global(knn), N, K, as_global(w1), as_global(shift1), as_global(w2img),
                                               as_global(scales), as_global(gamma_last), as_global(ext), as_global(partial), blockIdx.x, nullptr, 64,
                                               0, as_global(range_bits));
}
// S1X: the scene's MLP2 weights as they are; S2X: its folded conv1 (ec_w1f, ec_sh1) + the raw conv2
// the engine's launches: E goes straight into the clusters' maxima (kFused above), c.pf is not written
// the hand-scheduled slot loops (kAsm), two waves per SIMD like the kernels above
template <int MODE>
__global__ __launch_bounds__(64 * kWaves, 2) void k_edgeconv_h(const float* __restrict__ x9m, const int32_t* __restrict__ knn, int N, int K,
                                                          const float* __restrict__ w1, const float* __restrict__ shift1,
                                                          const u32x4* __restrict__ w2img, const float* __restrict__ scales,
                                                          const float* __restrict__ gamma_last,
                                                          float* __restrict__ ext, double* __restrict__ partial,
                                                          const unsigned int* __restrict__ range_bits) {
    using sg::as_global;
    edgeconv_body<MODE, MODE == S2X, false, true, true>(as_global(x9m), as_global(knn), N, K, as_global(w1), as_global(shift1), as_global(w2img),
                                                        as_global(scales), as_global(gamma_last), as_global(ext), as_global(partial), blockIdx.x, nullptr, 64,
                                                        0, as_global(range_bits), gridDim.x);
}
template <int MODE>
__global__ __launch_bounds__(64 * kWaves, 2) void k_edgeconv_hb(const sg::SlotCtx* __restrict__ cx, int walk) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    // gridDim.x workgroups walk the scene's ec_blocks tile groups (b_edgeconv sizes the grid by what is resident at once); a scene with
    // fewer groups than that uses one workgroup per group.  The workgroups that run are the ones that leave a row of partial sums.
    const int nblk = min((int)gridDim.x, c.ec_blocks);
    if ((int)blockIdx.x >= nblk) return;
    using sg::as_global;
    if constexpr (MODE == S1X) edgeconv_body<MODE, false, true, true, true>(as_global(c.x9m), as_global(c.knn), c.N, c.K, as_global(c.ec_w1), nullptr, nullptr, nullptr,
                                                               as_global(c.ec_g1), as_global(c.cat + c.gm_D), as_global(c.ec_partial), blockIdx.x,
                                                               as_global(c.cluster_of_pos), c.Dcat, 0, as_global((const unsigned int*)c.ec_range), nblk, walk);
    else edgeconv_body<MODE, true, true, true, true>(as_global(c.x9m), as_global(c.knn), c.N, c.K, as_global((const float*)c.ec_w1f),
                                             as_global((const float*)c.ec_sh1), as_global(reinterpret_cast<const u32x4*>(c.ec_w2img)),
                                             as_global((const float*)c.ec_scale), as_global(c.ec_g2), as_global(c.cat + c.gm_D),
                                             as_global(c.ec_partial), blockIdx.x, as_global(c.cluster_of_pos), c.Dcat, 0, nullptr, nblk, walk);
}
template <int MODE, bool REREAD_A>
__global__ __launch_bounds__(64 * kWaves, 8 / kWaves) void k_edgeconv_b(const sg::SlotCtx* __restrict__ cx, int stagger) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x >= c.ec_blocks) return;
    // pointers read out of a SlotCtx are generic to the compiler (sg_common.h, gptr): hand them over as global memory
    using sg::as_global;
    if constexpr (MODE == S1X) edgeconv_body<MODE, REREAD_A, true, true>(as_global(c.x9m), as_global(c.knn), c.N, c.K, as_global(c.ec_w1), nullptr, nullptr, nullptr,
                                                               as_global(c.ec_g1), as_global(c.cat + c.gm_D), as_global(c.ec_partial), blockIdx.x,
                                                               as_global(c.cluster_of_pos), c.Dcat, stagger, as_global((const unsigned int*)c.ec_range));
    else edgeconv_body<MODE, REREAD_A, true, true>(as_global(c.x9m), as_global(c.knn), c.N, c.K, as_global((const float*)c.ec_w1f),
                                             as_global((const float*)c.ec_sh1), as_global(reinterpret_cast<const u32x4*>(c.ec_w2img)),
                                             as_global((const float*)c.ec_scale), as_global(c.ec_g2), as_global(c.cat + c.gm_D),
                                             as_global(c.ec_partial), blockIdx.x, as_global(c.cluster_of_pos), c.Dcat, stagger);
}
// ... and the activation on the [C,64] maxima, in place, once the fold of the layer's last BatchNorm is known
__global__ __launch_bounds__(256) void k_cluster_affine_b(const sg::SlotCtx* __restrict__ cx, int layers) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;                   // (cluster, channel quad)
    if (i >= c.C * 16) return;
    const float* a = layers == 1 ? c.ec_w1f : c.ec_w2f;
    const float* b = layers == 1 ? c.ec_sh1 : c.ec_sh2;
    const int ch = (i & 15) * 4;
    float4* p = reinterpret_cast<float4*>(c.cat + c.gm_D + (size_t)(i >> 4) * c.Dcat + ch);
    float4 v = *p;
    float y;
    y = __builtin_fmaf(a[ch], v.x, b[ch]);         v.x = fmaxf(y, 0.2f * y);
    y = __builtin_fmaf(a[ch + 1], v.y, b[ch + 1]); v.y = fmaxf(y, 0.2f * y);
    y = __builtin_fmaf(a[ch + 2], v.z, b[ch + 2]); v.z = fmaxf(y, 0.2f * y);
    y = __builtin_fmaf(a[ch + 3], v.w, b[ch + 3]); v.w = fmaxf(y, 0.2f * y);
    *p = v;
}

// Batch statistics of conv1's output WITHOUT evaluating it (the inner BN of MLP3): y = W1 e is linear, so
//   mean_c = w_c . E[e]      E[y_c^2] = w_c^T E[e e^T] w_c
// and with e = [d, x_i], d = x_j - x_i, the 18 + 171 moments split into per-point terms and one per-slot term:
//   sum_j e      = [a, K x_i]                         a = sum_j d_j
//   sum_j e e^T  = [[D, a x_i^T], [x_i a^T, K x_i x_i^T]]   D = sum_j d_j d_j^T   (45 products per slot)
// One point per lane: 63 VALU instructions per neighbour slot instead of 18 MFMAs per 32 points (~10x cheaper; under
// the bench's load the MFMA statistics pass took 0.35 ms per scene).  Layout of the 189 partial sums per block:
// [0,9) a | [9,18) K x_i | [18,63) D upper triangle | [63,144) a x_i^T row-major | [144,189) K x_i x_i^T upper triangle.
constexpr int kMom = 189;

__device__ __forceinline__ void edge_moments_body(const float* __restrict__ x9m, const int32_t* __restrict__ knn, int N, int K,
                                                  double* __restrict__ partial, int bid) {
    __shared__ double red[16][kMom];                            // one partial per 16-lane row (4 waves x 4 rows), summed over the block's chunks
    const int tid = threadIdx.x, lane = tid & 63;
    for (int chunk = 0; chunk < sg::kMomPts / 256; ++chunk) {
    const int pt = (bid * (sg::kMomPts / 256) + chunk) * 256 + tid;
    if (chunk > 0 && (bid * (sg::kMomPts / 256) + chunk) * 256 >= N) break;      // block-uniform: nothing left
    const bool valid = pt < N;
    float a[9], D[45], xi[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { a[k] = 0.f; xi[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < 45; ++k) D[k] = 0.f;
    if (valid) {
        const float4* xr = reinterpret_cast<const float4*>(x9m + (size_t)pt * 12);
        const float4 q0 = xr[0], q1 = xr[1], q2 = xr[2];
        xi[0] = q0.x; xi[1] = q0.y; xi[2] = q0.z; xi[3] = q0.w; xi[4] = q1.x; xi[5] = q1.y; xi[6] = q1.z; xi[7] = q1.w; xi[8] = q2.x;
        const float cx = x9m[0], cy = x9m[1], cz = x9m[2];       // moments of [d, x_i - [c, 0]]: see "Conditioning" in the header
        // this kernel waits on memory 70 % of its life (PMC): neighbour rows are requested FOUR slots at a time (12 loads of
        // 16 B in flight per lane instead of 3), the neighbour ids of the next group while the current one is consumed
        const int32_t* krow = knn + (size_t)pt * K;
        constexpr int kG = 4;
        int ids[kG];
#pragma unroll
        for (int u = 0; u < kG; ++u) ids[u] = u < K ? krow[u] : 0;
        for (int j0 = 0; j0 < K; j0 += kG) {
            float4 r0[kG], r1[kG];
            float r2[kG];                                            // channel 8 alone: the row's last 12 bytes are padding (same time, 12 fewer registers)
            // (Round 5, measured wrong and not kept: taking d8 from d2 -- channels 6..8 are channels 0..2 minus the cluster's mean -- to save the
            // third gather.  It holds inside a cluster, but rows padded with POINT 0 (clusters of <= K points, model.py:513) have a neighbour from
            // another cluster, with another mean: 19 GPU tests failed.)
#pragma unroll
            for (int u = 0; u < kG; ++u) {
                const float* xrow = x9m + (size_t)ids[u] * 12;
                const float4* xq = reinterpret_cast<const float4*>(xrow);
                r0[u] = xq[0]; r1[u] = xq[1]; r2[u] = xrow[8];
            }
#pragma unroll
            for (int u = 0; u < kG; ++u) ids[u] = j0 + kG + u < K ? krow[j0 + kG + u] : 0;
#pragma unroll
            for (int u = 0; u < kG; ++u) {
                if (j0 + u >= K) break;
                const float4 n0 = r0[u], n1 = r1[u];
                const float d[9] = {n0.x - xi[0], n0.y - xi[1], n0.z - xi[2], n0.w - xi[3], n1.x - xi[4],
                                    n1.y - xi[5], n1.z - xi[6], n1.w - xi[7], r2[u] - xi[8]};
                int t = 0;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    a[k] += d[k];
#pragma unroll
                    for (int l = k; l < 9; ++l) { D[t] = __builtin_fmaf(d[k], d[l], D[t]); ++t; }
                }
            }
        }
        xi[0] -= cx; xi[1] -= cy; xi[2] -= cz;                   // d is done with the raw coordinates; the x_i terms are centred
    }
    const float Kf = valid ? (float)K : 0.f;
    // 189 sums per wave: fp32 over the 16 lanes of a row on a fixed DPP network (four steps: every lane of the row ends up with the row's
    // sum), then doubles -- the block's 16 rows in fixed order below, the blocks in fixed order in k_bn_fold_moments.  (Round 2 carried every
    // sum on to lane 63 and read it back: two more DPP steps and a v_readlane per value, a sixth of this kernel's instructions.)
    const int row16 = tid >> 4;
    auto wsum = [&](float v, int slot) {
        v += sgw::dpp_f<sgw::kQuadXor1>(v, v);
        v += sgw::dpp_f<sgw::kQuadXor2>(v, v);
        v += sgw::dpp_f<sgw::kRowRor4>(v, v);
        v += sgw::dpp_f<sgw::kRowRor8>(v, v);
        if ((lane & 15) == 0) red[row16][slot] = chunk == 0 ? (double)v : red[row16][slot] + (double)v;      // the row's own entry: no other thread touches it
    };
    int t = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) wsum(a[k], k);
#pragma unroll
    for (int k = 0; k < 9; ++k) wsum(Kf * xi[k], 9 + k);
#pragma unroll
    for (int k = 0; k < 45; ++k) wsum(D[k], 18 + k);
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int l = 0; l < 9; ++l) wsum(a[k] * xi[l], 63 + 9 * k + l);
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int l = k; l < 9; ++l) { wsum(Kf * xi[k] * xi[l], 144 + t); ++t; }
    }   // chunks
    __syncthreads();
    if (tid < kMom) {
        double t16 = red[0][tid];
#pragma unroll
        for (int i = 1; i < 16; ++i) t16 += red[i][tid];
        partial[(size_t)bid * kMom + tid] = t16;
    }
}
__global__ __launch_bounds__(256) void k_edge_moments(const float* __restrict__ x9m, const int32_t* __restrict__ knn, int N, int K,
                                                      double* __restrict__ partial) {
    edge_moments_body(x9m, knn, N, K, partial, blockIdx.x);
}
__global__ __launch_bounds__(256) void k_edge_moments_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    if ((int)blockIdx.x >= c.ec_mblocks) return;
    edge_moments_body(c.x9m, c.knn, c.N, c.K, c.ec_partial, blockIdx.x);
}

// moments -> folded conv1 weights and shift (one block; fixed-order sum of the per-block partials)
// ... and the operands of S2X's conv2 on the fp16 matrix pipe.  conv2 multiplies h = LReLU(conv1'(e)) by W2 as two fp16 pieces per
// value (hi = rn16(v), lo = rn16(v - hi): 22+ significand bits; products hi*hi + hi*lo + lo*hi accumulated in fp32).  fp16 has a
// narrow exponent range, so both operands are moved into it by POWERS OF TWO (exact; S2X divides them out again):
//   T: |h_c| <= |beta_c| + |gamma_c| sqrt(rows)  (a standardised value is at most sqrt(rows - 1); LReLU only shrinks), so with
//      T = 2^floor(log2(32768 / max_c bound_c)) no T h can overflow fp16 (65504) whatever the data is; T multiplies the folded conv1
//      weights and shift (the accumulator of conv1' is then T x the unscaled one bit for bit);
//   S: max |S W2| in [2^12, 2^13): the low pieces of all but vanishing weights are fp16 normals.
// The pre-split weight image (the layout S2X copies to LDS, rows already carrying sgn(gamma2)) is written here, once per scene,
// instead of once per workgroup.
__device__ __forceinline__ void bn_fold_moments_body(const double* __restrict__ partial, int nblocks, double rows,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ w, float* __restrict__ w_folded, float* __restrict__ shift,
                                                     const float* __restrict__ w2, const float* __restrict__ gamma2,
                                                     u32x4* __restrict__ w2img, float* __restrict__ scales,
                                                     const unsigned int* __restrict__ range_bits = nullptr, u32x4* __restrict__ c1img = nullptr) {
    __shared__ double part[4][256];
    __shared__ float sbound[64], swmax[16], sT, sS;
    __shared__ double tot[kMom];
    __shared__ double mu[18], M[18][18];
    const int v = threadIdx.x & 255, g = threadIdx.x >> 8;
    // eight independent chains per thread keep eight loads in flight (one chain = one L2 round trip per add)
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (v < kMom) {
        int b = g;
        for (; b + 28 < nblocks; b += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += partial[(size_t)(b + 4 * u) * kMom + v];
        }
        for (; b < nblocks; b += 4) acc[0] += partial[(size_t)b * kMom + v];
    }
    part[g][v] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (threadIdx.x < kMom) tot[threadIdx.x] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 18) mu[threadIdx.x] = tot[threadIdx.x] / rows;
    if (threadIdx.x < 324) {
        const int k = threadIdx.x / 18, l = threadIdx.x % 18;
        auto tri = [](int i, int j) { return i * 9 - i * (i - 1) / 2 + (j - i); };       // upper-triangle index, i <= j < 9
        double m;
        if (k < 9 && l < 9) m = tot[18 + (k <= l ? tri(k, l) : tri(l, k))];
        else if (k < 9) m = tot[63 + 9 * k + (l - 9)];
        else if (l < 9) m = tot[63 + 9 * l + (k - 9)];
        else m = tot[144 + (k <= l ? tri(k - 9, l - 9) : tri(l - 9, k - 9))];
        M[k][l] = m / rows;
    }
    __syncthreads();
    const int ch = threadIdx.x & 63;
    double mean = 0.0, ey2 = 0.0;
    for (int k = 0; k < 18; ++k) {
        const double wk = (double)w[ch * 18 + k];
        mean += wk * mu[k];
        double r = 0.0;
        for (int l = 0; l < 18; ++l) r += (double)w[ch * 18 + l] * M[k][l];
        ey2 += wk * r;
    }
    const double var = ey2 - mean * mean;
    const double a = (double)gamma[ch] / sqrt(var + 1e-5);
    float T = 1.f;
    if (w2img) {
        if (threadIdx.x < 64) sbound[ch] = fabsf(beta[ch]) + fabsf(gamma[ch]) * (float)sqrt(rows);
        float wm = 0.f;
        for (int i = threadIdx.x; i < 64 * 64; i += 1024) wm = fmaxf(wm, fabsf(w2[i]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
        if ((threadIdx.x & 63) == 0) swmax[threadIdx.x >> 6] = wm;
        __syncthreads();
        if (threadIdx.x == 0) {
            float b = 0.f, m = 0.f;
            for (int c = 0; c < 64; ++c) b = fmaxf(b, sbound[c]);
            for (int c = 0; c < 16; ++c) m = fmaxf(m, swmax[c]);
            sT = pow2_scale(b, 32768.f);
            sS = pow2_scale(m, 8191.f);
            scales[0] = 1.f / (sT * sS);                           // what S2X multiplies its results by
            scales[1] = sT;
            scales[2] = sS;
        }
        __syncthreads();
        T = sT;
        const float S = sS;
        if (threadIdx.x < 2 * 4 * 64) {                            // one (ot, kb, lane) fragment per thread: 8 weights -> hi | lo pieces
            const int l = threadIdx.x & 63, kb = (threadIdx.x >> 6) & 3, ot = threadIdx.x >> 8;
            const int oc = 32 * ot + (l & 31);
            const float sgn = gamma2[oc] < 0.f ? -S : S;           // the last layer's rows carry the sign of its BN gamma (see the header)
            unsigned int hi[4], lo[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x2 v = {sgn * w2[oc * 64 + acc_channel(kb >> 1, 8 * (kb & 1) + 2 * jj, l >> 5)],
                                 sgn * w2[oc * 64 + acc_channel(kb >> 1, 8 * (kb & 1) + 2 * jj + 1, l >> 5)]};
                const f16x2 h = __builtin_convertvector(v, f16x2);
                const f16x2 r = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), f16x2);
                hi[jj] = __builtin_bit_cast(unsigned int, h);
                lo[jj] = __builtin_bit_cast(unsigned int, r);
            }
            w2img[threadIdx.x] = u32x4{hi[0], hi[1], hi[2], hi[3]};
            w2img[512 + threadIdx.x] = u32x4{lo[0], lo[1], lo[2], lo[3]};
        }
    }
    if (c1img) {
        // Round 3: conv1' d-columns as an fp16 image too (S2X's conv1 on two fp16 pieces like MLP2's: 4 instead of 8 MFMAs per slot).  The
        // operand d is scaled by Sd (2 Sd range <= 2^12, range = the layer's range words), the folded weights a w T by 1 / Sd -- so the
        // accumulator stays T x conv1' -- and if a weight would leave fp16 that way (|a w T / Sd| > 2^14: a degenerate variance), T gives:
        // a smaller T only costs conv2's low pieces precision in that degenerate case.
        __shared__ double sa[64];
        __shared__ float samax[64], srange[sg::kRangeWords / 64], sSd;
        if (threadIdx.x < 64) {
            sa[ch] = a;
            float m = 0.f;
            for (int k = 0; k < 9; ++k) m = fmaxf(m, fabsf((float)(a * (double)w[ch * 18 + k])));
            samax[ch] = m;
        }
        if (threadIdx.x < sg::kRangeWords) {                     // one range word per thread (one thread walking all 256: 17 us of dependent loads)
            const float rw = sgw::wave_max(__uint_as_float(range_bits[threadIdx.x]));
            if ((threadIdx.x & 63) == 0) srange[threadIdx.x >> 6] = rw;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float am = 0.f, rm = 0.f;
            for (int c = 0; c < 64; ++c) am = fmaxf(am, samax[c]);
            for (int i = 0; i < sg::kRangeWords / 64; ++i) rm = fmaxf(rm, srange[i]);
            float Sd = pow2_scale(2.f * rm, 4096.f);
            const float lim = pow2_scale(am * sT / Sd, 16384.f);
            if (lim < 1.f) { sT *= lim; scales[0] = 1.f / (sT * sS); scales[1] = sT; }
            // the product only sees (weight / Sd) (d Sd): move up to 2^7 of the scale from d to the weights while the largest weight stays
            // below 2^12, so that the weights' low pieces are not cut off at fp16's smallest subnormal (2^-24); d keeps an absolute
            // resolution of 2^-25 2^7 / Sd <= 2^-30 of the layer's range, finer than the fp32 coordinates it is the difference of
            const float room = pow2_scale(am * sT / Sd, 4095.f);
            if (room > 1.f) Sd /= fminf(room, 128.f);
            sSd = Sd;
            scales[3] = Sd;
        }
        __syncthreads();
        T = sT;
        if (threadIdx.x < 2 * 2 * 64) {                            // one (m, t, lane) fragment per thread, the layout of edgeconv_body's fp16 image
            const int l = threadIdx.x & 63, t = (threadIdx.x >> 6) & 1, m = threadIdx.x >> 7;
            const int oc = 32 * t + (l & 31), hf = l >> 5, c0 = 4 * hf;
            const float inv = 1.f / sSd;
            auto piece = [&](int k, int p) -> unsigned int {
                if (p == 0) return 0u;
                const float v = (float)(sa[oc] * (double)w[oc * 18 + k]) * T * inv;       // w_folded's value / Sd: exact (a power of two)
                const _Float16 h = (_Float16)v;
                const _Float16 lo_ = (_Float16)(v - (float)h);
                return (unsigned int)__builtin_bit_cast(unsigned short, p == 1 ? h : lo_);
            };
            unsigned int u[4];
            const int pa = m == 0 ? 1 : 2;
            u[0] = piece(c0, pa) | (piece(c0 + 1, pa) << 16); u[1] = piece(c0 + 2, pa) | (piece(c0 + 3, pa) << 16);
            if (m == 0) { u[2] = u[0]; u[3] = u[1]; }
            else { u[2] = piece(8, hf ? 2 : 1); u[3] = piece(8, hf ? 0 : 1); }
            c1img[threadIdx.x] = u32x4{u[0], u[1], u[2], u[3]};
        }
    }
    for (int k = threadIdx.x >> 6; k < 18; k += 16) w_folded[ch * 18 + k] = (float)(a * (double)w[ch * 18 + k]) * T;
    if (threadIdx.x < 64) shift[ch] = (float)((double)beta[ch] - a * mean) * T;
}
__global__ __launch_bounds__(1024) void k_bn_fold_moments(const double* __restrict__ partial, int nblocks, double rows,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ w, float* __restrict__ w_folded, float* __restrict__ shift,
                                                          const float* __restrict__ w2, const float* __restrict__ gamma2,
                                                          u32x4* __restrict__ w2img, float* __restrict__ scales,
                                                          const unsigned int* __restrict__ range_bits, u32x4* __restrict__ c1img) {
    bn_fold_moments_body(partial, nblocks, rows, gamma, beta, w, w_folded, shift, w2, gamma2, w2img, scales, range_bits, c1img);
}
__global__ __launch_bounds__(1024) void k_bn_fold_moments_b(const sg::SlotCtx* __restrict__ cx) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    bn_fold_moments_body(c.ec_partial, c.ec_mblocks, (double)c.N * 20.0, c.ec_g1, c.ec_b1, c.ec_w1, c.ec_w1f, c.ec_sh1, c.ec_w2, c.ec_g2,
                         reinterpret_cast<u32x4*>(c.ec_w2img), c.ec_scale, c.ec_range, reinterpret_cast<u32x4*>(c.ec_scale + 4));
}

// last layer of an MLP: fixed-order reduction of the per-block partials -> |a| = |gamma| / sqrt(var + eps) and the shift.
// 1024 threads: thread (value v = tid & 127, lane group g = tid >> 7) sums blocks g, g+8, g+16, ... and the
// eight group sums are added in a fixed order, so the result does not depend on scheduling.
__device__ __forceinline__ void bn_fold_body(const double* __restrict__ partial, int nblocks, double rows,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float* __restrict__ a_out, float* __restrict__ shift, float* __restrict__ stats = nullptr) {
    __shared__ double part[8][128];
    __shared__ double tot[128];
    const int v = threadIdx.x & 127, g = threadIdx.x >> 7;
    // four independent chains per thread keep ~4 loads in flight (a single chain is one L2 round trip per add)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = g;
    for (; b + 24 < nblocks; b += 32) {
        s0 += partial[(size_t)b * 128 + v];
        s1 += partial[(size_t)(b + 8) * 128 + v];
        s2 += partial[(size_t)(b + 16) * 128 + v];
        s3 += partial[(size_t)(b + 24) * 128 + v];
    }
    for (; b < nblocks; b += 8) s0 += partial[(size_t)b * 128 + v];
    part[g][v] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < 128) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        // the statistics are those of y' = sgn(gamma)*y (mean' = sgn*mean, same variance), so
        // a*y + (beta - a*mean) == |a|*y' + (beta - |a|*mean')
        const int ch = threadIdx.x;
        const double mean = tot[ch] / rows;
        const double var = tot[64 + ch] / rows - mean * mean;
        const double a = fabs((double)gamma[ch]) / sqrt(var + 1e-5);
        a_out[ch] = (float)a;
        shift[ch] = (float)((double)beta[ch] - a * mean);
        if (stats) {                                              // batch mean of y itself | biased variance (the training step's tape)
            stats[ch] = (float)(gamma[ch] < 0.f ? -mean : mean);
            stats[64 + ch] = (float)var;
        }
    }
}
__global__ __launch_bounds__(1024) void k_bn_fold(const double* __restrict__ partial, int nblocks, double rows,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float* __restrict__ a_out, float* __restrict__ shift, float* __restrict__ stats,
                                                  unsigned int* __restrict__ range_bits) {
    bn_fold_body(partial, nblocks, rows, gamma, beta, a_out, shift, stats);
    if (range_bits && threadIdx.x < sg::kRangeWords) range_bits[threadIdx.x] = 0u;      // the layer is done with its range words: the next layout starts from zero
}
// layers == 1: MLP2's only BN (-> ec_w1f = |a|, ec_sh1); layers == 2: MLP3's last BN (-> ec_w2f, ec_sh2)
__global__ __launch_bounds__(1024) void k_bn_fold_b(const sg::SlotCtx* __restrict__ cx, int layers, int max_rows) {
    const sg::SlotCtx& c = cx[blockIdx.y];
    const int rows = min(c.ec_blocks, max_rows);             // the workgroups of the EdgeConv launch that walked this scene
    if (layers == 1) bn_fold_body(c.ec_partial, rows, (double)c.N * 20.0, c.ec_g1, c.ec_b1, c.ec_w1f, c.ec_sh1);
    else bn_fold_body(c.ec_partial, rows, (double)c.N * 20.0, c.ec_g2, c.ec_b2, c.ec_w2f, c.ec_sh2);
    if (threadIdx.x < sg::kRangeWords) c.ec_range[threadIdx.x] = 0u;      // the layer is done with its range words (k_layer_layout raises them again)
}

// a caller without sg_layer_layout's range word: the largest |x9m[p][c] - x9m[0][c]| over all rows and the 9 channels (twice that bounds
// every difference between two rows), raised into *range_bits like the layout kernel does
__global__ __launch_bounds__(256) void k_edge_range(const float* __restrict__ x9m, int N, unsigned int* __restrict__ range_bits) {
    float m = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < N; p += gridDim.x * 256)
#pragma unroll
        for (int c = 0; c < 9; ++c) m = fmaxf(m, fabsf(x9m[(size_t)p * 12 + c] - x9m[c]));
    m = sgw::wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(range_bits, __float_as_uint(m));
}

// epilogue of the last layer, in place: out = LReLU(|a| * E + b')
__global__ __launch_bounds__(256) void k_bn_lrelu_apply(float* __restrict__ e, size_t n4, const float* __restrict__ a,
                                                        const float* __restrict__ shift) {
    __shared__ float sa[64], sb[64];
    if (threadIdx.x < 64) { sa[threadIdx.x] = a[threadIdx.x]; sb[threadIdx.x] = shift[threadIdx.x]; }
    __syncthreads();
    float4* p = reinterpret_cast<float4*>(e);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & 15) * 4;
        float4 v = p[i];
        float y;
        y = __builtin_fmaf(sa[c], v.x, sb[c]);         v.x = fmaxf(y, 0.2f * y);
        y = __builtin_fmaf(sa[c + 1], v.y, sb[c + 1]); v.y = fmaxf(y, 0.2f * y);
        y = __builtin_fmaf(sa[c + 2], v.z, sb[c + 2]); v.z = fmaxf(y, 0.2f * y);
        y = __builtin_fmaf(sa[c + 3], v.w, sb[c + 3]); v.w = fmaxf(y, 0.2f * y);
        p[i] = v;
    }
}

}  // namespace

namespace sg {

// workgroups of 64 * kWaves threads resident at once at two per CU (what the hand-scheduled kernels are built for): the grid of a launch
// whose workgroups walk their scene's tile groups.  SG_EC_WALK (development) overrides the factor: 0 = one group per workgroup.
static int resident_workgroups() {
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const char* e = getenv("SG_EC_WALK");
        const double f = e ? atof(e) : 2.0;
        return f <= 0.0 ? (1 << 28) : std::max(1, (int)(cus * f));
    }();
    return n;
}

// SG_EC_COMPILER_LOOP=1 (development / A-B timing): every EdgeConv launch of the process takes the compiler-scheduled slot loop
static const bool g_compiler_loop = getenv("SG_EC_COMPILER_LOOP") ? atoi(getenv("SG_EC_COMPILER_LOOP")) != 0 : false;

// the epilogue alone: d_dst = LReLU(|a| * d_e + b') (debug taps of the pipeline, which otherwise fuses it into the segment max)
int edgeconv_apply(const float* d_e, int N, const float* d_a, const float* d_shift, float* d_dst, void* stream) {
    if (N == 0) return SG_OK;
    hipStream_t st = sg::as_stream(stream);
    if (d_dst != d_e) SG_HIP(hipMemcpyAsync(d_dst, d_e, (size_t)N * 64 * 4, hipMemcpyDeviceToDevice, st));
    const size_t n4 = (size_t)N * 16;
    k_bn_lrelu_apply<<<(int)std::min<size_t>((n4 + 255) / 256, 2048), 256, 0, st>>>(d_dst, n4, d_a, d_shift);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int edgeconv_forward_marked(const float* d_x9m, const int32_t* d_knn, int N, int k, int layers, const float* d_w1, const float* d_g1,
                            const float* d_b1, const float* d_w2, const float* d_g2, const float* d_b2, float* d_out, void* d_ws,
                            size_t ws_bytes, void* stream, const std::function<void(int)>& mark, const float** d_affine,
                            unsigned int* d_range_bits, unsigned flags) {
    SG_REQUIRE(N >= 0 && k > 0 && (layers == 1 || layers == 2) && d_ws, "sg_edgeconv_forward: bad arguments");
    SG_REQUIRE(!d_range_bits || k <= 32, "sg_edgeconv_forward_r: at most 32 neighbours per point (their ids are parked in a 32 x 65-word LDS strip)");
    if (d_affine) { d_affine[0] = nullptr; d_affine[1] = nullptr; d_affine[2] = nullptr; }
    if (N == 0) return SG_OK;
    const int ngroups = sg::cdiv(sg::cdiv(N, 32), kWaves);
    const bool hand = k == 20 && d_range_bits && !(flags & SG_EDGECONV_COMPILER_LOOP) && !g_compiler_loop;    // the hand-scheduled slot loop is written for K = 20
    const int nblocks = hand ? std::min(ngroups, resident_workgroups()) : ngroups;       // = rows of partial sums
    sg::Carver cv(d_ws, ws_bytes);
    double* partial = cv.take<double>(std::max((size_t)nblocks * 128, (size_t)sg::moments_blocks(N) * kMom));
    float* fold = cv.take<float>(sg::kEdgeFoldFloats + 128);
    if (!cv.ok) return sg::fail(SG_ENOMEM, "sg_edgeconv_forward: workspace too small (%zu < %zu)", ws_bytes, sg_edgeconv_ws_bytes(N));
    float* w1f = fold;
    float* sh1 = w1f + 64 * 18;
    float* w2f = sh1 + 64;
    float* sh2 = w2f + 64 * 64;
    u32x4* w2img = reinterpret_cast<u32x4*>(sh2 + 64);           // S2X's pre-split conv2 weights + {1/(S T), T, S}
    float* scales = sh2 + 64 + 4096;
    float* stats_last = fold + sg::kEdgeFoldFloats;              // batch mean | variance of the LAST BatchNorm's input
    hipStream_t st = sg::as_stream(stream);
    const double rows = (double)N * (double)k;
    const dim3 grid(nblocks), block(64 * kWaves);
    const size_t n4 = (size_t)N * 16;
    const int egrid = (int)std::min<size_t>((n4 + 255) / 256, 2048);
    if (layers == 1) {
        if (d_range_bits && hand) k_edgeconv_h<S1X><<<grid, block, 0, st>>>(d_x9m, d_knn, N, k, d_w1, nullptr, nullptr, nullptr, d_g1, d_out, partial, d_range_bits);
        else if (d_range_bits) k_edgeconv<S1X, false, true><<<grid, block, 0, st>>>(d_x9m, d_knn, N, k, d_w1, nullptr, nullptr, nullptr, d_g1, d_out, partial, d_range_bits);
        else k_edgeconv<S1X><<<grid, block, 0, st>>>(d_x9m, d_knn, N, k, d_w1, nullptr, nullptr, nullptr, d_g1, d_out, partial);
        k_bn_fold<<<1, 1024, 0, st>>>(partial, nblocks, rows, d_g1, d_b1, w1f, sh1, stats_last, d_range_bits);
        if (mark) mark(0);
        if (d_affine) { d_affine[0] = w1f; d_affine[1] = sh1; d_affine[2] = stats_last; }        // the caller applies LReLU(|a| E + b') where it consumes E
        else k_bn_lrelu_apply<<<egrid, 256, 0, st>>>(d_out, n4, w1f, sh1);
        if (mark) mark(1);
    } else {
        const int mblocks = sg::moments_blocks(N);
        k_edge_moments<<<mblocks, 256, 0, st>>>(d_x9m, d_knn, N, k, partial);
        k_bn_fold_moments<<<1, 1024, 0, st>>>(partial, mblocks, rows, d_g1, d_b1, d_w1, w1f, sh1, d_w2, d_g2, w2img, scales, d_range_bits,
                                              d_range_bits ? reinterpret_cast<u32x4*>(scales + 4) : nullptr);
        if (mark) mark(0);
        if (d_range_bits && hand) k_edgeconv_h<S2X><<<grid, block, 0, st>>>(d_x9m, d_knn, N, k, w1f, sh1, w2img, scales, d_g2, d_out, partial, d_range_bits);
        else if (d_range_bits) k_edgeconv<S2X, true, true><<<grid, block, 0, st>>>(d_x9m, d_knn, N, k, w1f, sh1, w2img, scales, d_g2, d_out, partial, d_range_bits);
        else k_edgeconv<S2X, true><<<grid, block, 0, st>>>(d_x9m, d_knn, N, k, w1f, sh1, w2img, scales, d_g2, d_out, partial);
        k_bn_fold<<<1, 1024, 0, st>>>(partial, nblocks, rows, d_g2, d_b2, w2f, sh2, stats_last, d_range_bits);
        if (mark) mark(1);
        if (d_affine) { d_affine[0] = w2f; d_affine[1] = sh2; d_affine[2] = stats_last; }
        else k_bn_lrelu_apply<<<egrid, 256, 0, st>>>(d_out, n4, w2f, sh2);
        if (mark) mark(2);
    }
    SG_LAUNCH_CHECK();
    return SG_OK;
}

// the edge-feature moments on their own, for the training step's BatchNorm backward (kernels_train_edge.hip): per-block partials
// [moments_blocks(N)][189] in edge_moments_body's layout (a | K x_i | D upper | a x_i^T | K x_i x_i^T upper; x_i XYZ relative to row 0)
int edge_moments_partials(const float* d_x9m, const int32_t* d_knn, int N, int K, double* d_partial, hipStream_t st) {
    if (N <= 0) return SG_OK;
    k_edge_moments<<<sg::moments_blocks(N), 256, 0, st>>>(d_x9m, d_knn, N, K, d_partial);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

int b_edgeconv(const SlotCtx* d_ctx, const BatchDims& bd, int layers, void (*mark)(void*, int), void* mark_arg, hipStream_t st) {
    if (bd.nslots == 0 || bd.max_N == 0) return SG_OK;
    const int ngroups = sg::cdiv(sg::cdiv(bd.max_N, 32), kWaves);
    // the generated slot loops are unrolled for K = 20 (neighbour rows at a fixed 80-byte stride): any other K in the group takes the
    // compiler loop, which reads SlotCtx::K (nothing inside k_edgeconv_hb checks it)
    const bool hand = !g_compiler_loop && bd.min_K == kAsmK && bd.max_K == kAsmK;
    const int nblocks = hand ? std::min(ngroups, std::max(1, resident_workgroups() / bd.nslots)) : ngroups;
    const dim3 grid(nblocks, bd.nslots), one(1, bd.nslots);
    static const int walk = getenv("SG_EC_WALK_MODE") ? atoi(getenv("SG_EC_WALK_MODE")) : 0;      // 0 = stride (default); 1 = contiguous ranges, XCD-grouped (edgeconv_body): measured, see there
    // development knobs: SG_EC_STAGGER1 / SG_EC_STAGGER2 = start offset of the odd wave slot in units of 64 cycles
    static const int stagger1 = getenv("SG_EC_STAGGER1") ? atoi(getenv("SG_EC_STAGGER1")) : kStagger1;
    static const int stagger2 = getenv("SG_EC_STAGGER2") ? atoi(getenv("SG_EC_STAGGER2")) : kStagger2;
    if (layers == 1) {
        if (mark) mark(mark_arg, 2);                                  // 2 / 3: in front of / behind the EdgeConv launch itself
        if (hand) k_edgeconv_hb<S1X><<<grid, 64 * kWaves, 0, st>>>(d_ctx, walk);
        else k_edgeconv_b<S1X, false><<<grid, 64 * kWaves, 0, st>>>(d_ctx, stagger1);
        if (mark) mark(mark_arg, 3);
        k_bn_fold_b<<<one, 1024, 0, st>>>(d_ctx, 1, nblocks);
        k_cluster_affine_b<<<dim3(sg::cdiv(bd.max_C * 16, 256), bd.nslots), 256, 0, st>>>(d_ctx, 1);
        if (mark) mark(mark_arg, 0);
    } else {
        k_edge_moments_b<<<dim3(sg::moments_blocks(bd.max_N), bd.nslots), 256, 0, st>>>(d_ctx);
        k_bn_fold_moments_b<<<one, 1024, 0, st>>>(d_ctx);
        if (mark) mark(mark_arg, 0);
        if (hand) k_edgeconv_hb<S2X><<<grid, 64 * kWaves, 0, st>>>(d_ctx, walk);
        else k_edgeconv_b<S2X, true><<<grid, 64 * kWaves, 0, st>>>(d_ctx, stagger2);
        if (mark) mark(mark_arg, 3);
        k_bn_fold_b<<<one, 1024, 0, st>>>(d_ctx, 2, nblocks);
        k_cluster_affine_b<<<dim3(sg::cdiv(bd.max_C * 16, 256), bd.nslots), 256, 0, st>>>(d_ctx, 2);
        if (mark) mark(mark_arg, 1);
    }
    SG_LAUNCH_CHECK();
    return SG_OK;
}

}  // namespace sg

extern "C" {

size_t sg_edgeconv_ws_bytes(int N) {
    const size_t nblocks = (size_t)sg::cdiv(sg::cdiv(std::max(N, 1), 32), kWaves);
    const size_t mblocks = (size_t)sg::moments_blocks(std::max(N, 1));
    return sg::align_up(std::max(nblocks * 128, mblocks * 189) * 8) + sg::align_up((sg::kEdgeFoldFloats + 128) * 4);
}

int sg_edgeconv_forward(const float* d_x9m, const int32_t* d_knn, int N, int k, int layers, const float* d_w1, const float* d_g1,
                        const float* d_b1, const float* d_w2, const float* d_g2, const float* d_b2, float* d_out, void* d_ws,
                        size_t ws_bytes, void* stream) {
    return sg::edgeconv_forward_marked(d_x9m, d_knn, N, k, layers, d_w1, d_g1, d_b1, d_w2, d_g2, d_b2, d_out, d_ws, ws_bytes, stream,
                                       nullptr, nullptr, nullptr, 0u);
}

int sg_edgeconv_forward_r(const float* d_x9m, const int32_t* d_knn, int N, int k, int layers, const float* d_w1, const float* d_g1,
                          const float* d_b1, const float* d_w2, const float* d_g2, const float* d_b2, float* d_out, void* d_ws,
                          size_t ws_bytes, unsigned int* d_range_bits, void* stream) {
    return sg::edgeconv_forward_marked(d_x9m, d_knn, N, k, layers, d_w1, d_g1, d_b1, d_w2, d_g2, d_b2, d_out, d_ws, ws_bytes, stream,
                                       nullptr, nullptr, d_range_bits, 0u);
}

int sg_edgeconv_forward_x(const float* d_x9m, const int32_t* d_knn, int N, int k, int layers, const float* d_w1, const float* d_g1,
                          const float* d_b1, const float* d_w2, const float* d_g2, const float* d_b2, float* d_out, void* d_ws,
                          size_t ws_bytes, unsigned int* d_range_bits, unsigned int flags, void* stream) {
    return sg::edgeconv_forward_marked(d_x9m, d_knn, N, k, layers, d_w1, d_g1, d_b1, d_w2, d_g2, d_b2, d_out, d_ws, ws_bytes, stream,
                                       nullptr, nullptr, d_range_bits, flags);
}

#ifdef SG_KNN_PROFILE
// profiling builds only: [3][8] sums of 10 ns ticks per (0 = compiler loop, 1 = MLP2 hand, 2 = MLP3 hand) x phase; cleared by the call
int sg_debug_ec_phases(unsigned long long* h_out) {
    SG_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_ec_phase), sizeof(unsigned long long) * 24));
    unsigned long long z[24] = {0};
    SG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ec_phase), z, sizeof z));
    return SG_OK;
}
#endif

int sg_edge_range(const float* d_x9m, int N, unsigned int* d_range_bits, void* stream) {
    SG_REQUIRE(N >= 0 && d_x9m && d_range_bits, "sg_edge_range: bad arguments");
    if (N == 0) return SG_OK;
    k_edge_range<<<std::min(sg::cdiv(N, 256), 1024), 256, 0, sg::as_stream(stream)>>>(d_x9m, N, d_range_bits);
    SG_LAUNCH_CHECK();
    return SG_OK;
}

}  // extern "C"
