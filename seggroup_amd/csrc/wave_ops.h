// Wave64 reductions on the DPP data path (CDNA / GFX9 encodings: quad_perm, row_ror, row_bcast15 / 31) with the result
// handed back through v_readlane -- no LDS traffic.  `__shfl_xor` compiles to ds_bpermute_b32 on gfx950: every butterfly
// step is an LDS round trip (~100 cycles of latency in a dependent chain), which is what the one-wave-per-segment kernels
// of the structural layer (FPS, MLP1) spent most of their time on.
//
//   step 1-2  quad_perm [1,0,3,2], [2,3,0,1] : every lane holds its quad's result
//   step 3-4  row_ror 4, row_ror 8           : every lane holds its 16-lane row's result
//   step 5    row_bcast15 (rows 1 and 3)     : lane 15 of row r-1 is combined into row r
//   step 6    row_bcast31 (rows 2 and 3)     : lane 31 is combined into rows 2-3 -> lane 63 holds the wave's result
//   readlane 63                              : the result, wave-uniform (an SGPR)
// The association order of a floating-point sum is fixed by this network (deterministic, but different from a xor
// butterfly's).  All 64 lanes must be active.
#pragma once
#include <hip/hip_runtime.h>

namespace sgw {

// Full row mask: every lane receives a value (quad_perm / row_ror have no out-of-range source), so `old` is irrelevant -- and it
// must not be a live register: with old = v the compiler emits copy + v_mov_b32_dpp + op (4 instructions per step with the
// hazard nop); with old = 0 and bound_ctrl it folds the move into the consumer (`v_add_f32_dpp v, v, v quad_perm:...`: one).
// A partial row mask (the row_bcast steps) keeps `old` for the lanes that are not written.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i(int old, int v) {
    if constexpr (ROW_MASK == 0xF) return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
    else return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float old, float v) {
    return __int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(old), __float_as_int(v)));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_d(double old, double v) {
    const long long o = __double_as_longlong(old), x = __double_as_longlong(v);
    const int lo = dpp_i<CTRL, ROW_MASK>((int)(o & 0xffffffffll), (int)(x & 0xffffffffll));
    const int hi = dpp_i<CTRL, ROW_MASK>((int)(o >> 32), (int)(x >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kRowRor4 = 0x124, kRowRor8 = 0x128, kRowBcast15 = 0x142, kRowBcast31 = 0x143;

// generic network: `op(a, b)` must be associative and commutative; `self` is what a lane outside the row mask of the two
// broadcast steps sees as the incoming value (the identity of op, or -- for idempotent ops -- the lane's own value)
#define SGW_REDUCE(T, DPP, v, OP, IDENT)                                           \
    do {                                                                           \
        v = OP(v, DPP<kQuadXor1>(v, v));                                           \
        v = OP(v, DPP<kQuadXor2>(v, v));                                           \
        v = OP(v, DPP<kRowRor4>(v, v));                                            \
        v = OP(v, DPP<kRowRor8>(v, v));                                            \
        v = OP(v, DPP<kRowBcast15, 0xA>(IDENT, v));                                \
        v = OP(v, DPP<kRowBcast31, 0xC>(IDENT, v));                                \
    } while (0)

__device__ __forceinline__ float op_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float op_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float op_add(float a, float b) { return a + b; }
__device__ __forceinline__ double op_addd(double a, double b) { return a + b; }
__device__ __forceinline__ int op_maxi(int a, int b) { return max(a, b); }

__device__ __forceinline__ float wave_max(float v) {
    SGW_REDUCE(float, dpp_f, v, op_max, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min(float v) {
    SGW_REDUCE(float, dpp_f, v, op_min, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_max(int v) {
    SGW_REDUCE(int, dpp_i, v, op_maxi, v);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float wave_sum(float v) {
    SGW_REDUCE(float, dpp_f, v, op_add, 0.f);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_sum(double v) {
    SGW_REDUCE(double, dpp_d, v, op_addd, 0.0);
    const long long x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(x & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(x >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ int op_mini(int a, int b) { return min(a, b); }
__device__ __forceinline__ int wave_min(int v) {
    SGW_REDUCE(int, dpp_i, v, op_mini, v);
    return __builtin_amdgcn_readlane(v, 63);
}

// argmax with "larger value wins, ties -> lower index" (np.argmax), in two plain reductions: the wave's maximum (every step
// folds into one v_max_f32_dpp), then the smallest index among the lanes that hold it (v_min_i32_dpp).  The pairwise network on
// (value, index) pairs it replaces cost 9 instructions per step (two moves, three compares, two mask operations, two selects):
// 54 of the ~105 instructions of an FPS step.  Values must not be NaN (a NaN equals nothing: no lane would own the maximum).
__device__ __forceinline__ void wave_argmax(float& val, int& idx) {
    const float m = wave_max(val);
    idx = wave_min(val == m ? idx : 0x7fffffff);
    val = m;
}

// reductions over the 32 lanes of each HALF of the wave (two 32-point chunks per wave step): rows of 16 on the DPP path, row 0 -> 1
// and row 2 -> 3 with row_bcast15, then the half's result is read from its last lane (31 / 63) for every lane of the half
__device__ __forceinline__ float half_reduce_pick(float v, int lane) {
    const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return (lane & 32) ? hi : lo;
}
__device__ __forceinline__ float half_max(float v, int lane) {
    v = fmaxf(v, dpp_f<kQuadXor1>(v, v)); v = fmaxf(v, dpp_f<kQuadXor2>(v, v));
    v = fmaxf(v, dpp_f<kRowRor4>(v, v));  v = fmaxf(v, dpp_f<kRowRor8>(v, v));
    v = fmaxf(v, dpp_f<kRowBcast15, 0xA>(v, v));
    return half_reduce_pick(v, lane);
}
__device__ __forceinline__ float half_min(float v, int lane) {
    v = fminf(v, dpp_f<kQuadXor1>(v, v)); v = fminf(v, dpp_f<kQuadXor2>(v, v));
    v = fminf(v, dpp_f<kRowRor4>(v, v));  v = fminf(v, dpp_f<kRowRor8>(v, v));
    v = fminf(v, dpp_f<kRowBcast15, 0xA>(v, v));
    return half_reduce_pick(v, lane);
}

// lane `j` (wave-uniform) of v for every lane: v_readlane_b32 instead of a ds_bpermute round trip
__device__ __forceinline__ float bcast(float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); }
__device__ __forceinline__ int bcast(int v, int j) { return __builtin_amdgcn_readlane(v, j); }

}  // namespace sgw
