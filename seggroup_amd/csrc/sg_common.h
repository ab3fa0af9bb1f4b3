// Shared helpers for libseggroup_hip.so (error plumbing, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <functional>
#include <vector>

#include "seggroup_hip.h"

namespace sg {

// A pointer loaded from a descriptor struct (SlotCtx) is a GENERIC pointer to the compiler: every access through it is a
// flat_load / flat_store -- counted on vmcnt AND lgkmcnt, so LDS waits and memory waits block each other, plus an aperture check
// per access.  Device code that mixes LDS traffic with gathers therefore takes its pointers in address space 1 (gptr<T>):
// global_load / global_store, vmcnt only.  (A cast back to a generic pointer loses the information again.)
#if defined(__HIPCC__)
#if defined(__HIP_DEVICE_COMPILE__)
#define SG_GLOBAL __attribute__((address_space(1)))
#define SG_LDS __attribute__((address_space(3)))
#else
#define SG_GLOBAL                    /* the host pass only parses device code */
#define SG_LDS
#endif
template <class T>
using gptr = SG_GLOBAL T*;
template <class T>
__host__ __device__ __forceinline__ gptr<T> as_global(T* p) { return (gptr<T>)p; }
#endif


char* err_buf();                       // thread-local message buffer (defined in capi.cpp)
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// roctx ranges (capi.cpp): no-ops unless SG_ROCTX is set when the library loads and librocprofiler-sdk-roctx.so can be opened;
// `rocprofv3 --kernel-trace --marker-trace` then shows the engine's phases and stages per group thread (bench.py --profile)
void roctx_push(const char* name);
void roctx_pop();
struct RoctxRange {
    explicit RoctxRange(const char* name) { roctx_push(name); }
    ~RoctxRange() { roctx_pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

#define SG_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess)                                                            \
            return ::sg::fail(SG_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

#define SG_LAUNCH_CHECK()                                                                 \
    do {                                                                                  \
        hipError_t e__ = hipGetLastError();                                               \
        if (e__ != hipSuccess)                                                            \
            return ::sg::fail(SG_EHIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

#define SG_REQUIRE(cond, ...)                                                             \
    do {                                                                                  \
        if (!(cond)) return ::sg::fail(SG_EINVAL, __VA_ARGS__);                           \
    } while (0)

// internal variant of sg_fps_sample with the largest cluster size known to the host (kernels_fps.hip)
int evaluate_landing(const int32_t* d_gt, const int32_t* d_sem_pred, const int32_t* d_ins_pred, int V, int max_ins, float* h_iou_sem,
                     float* h_iou_ins, float* h_acc, void* d_ws, size_t ws_bytes, void* stream, uint32_t* h_landing);   // kernels_graph.hip
int knn_variant_for(int T, int override);   // kernels_knn_sorted.hip: 0 = two-pass over the chunk table, 1/2/4 = waves per tile of the one-pass kernel, 8 = x1 seeded
int fps_sample_hint(const float* d_data, int N, int ch_in, const int32_t* d_members, const int32_t* d_cl_off, int C, int P,
                    int ch_out, int transform, float* d_samples, int32_t* d_sel, void* d_ws, size_t ws_bytes, void* stream,
                    int max_n);

// internal variant of sg_edgeconv_forward: `mark(i)` is called after step i so the pipeline can time the steps
// separately (kernels_edgeconv.hip).  layers == 1: 0 = S1X + fold, 1 = epilogue.  layers == 2: 0 = edge-feature
// moments + fold, 1 = S2X + fold, 2 = epilogue.
int edgeconv_forward_marked(const float* d_x9m, const int32_t* d_knn, int N, int k, int layers, const float* d_w1, const float* d_g1,
                            const float* d_b1, const float* d_w2, const float* d_g2, const float* d_b2, float* d_out, void* d_ws,
                            size_t ws_bytes, void* stream, const std::function<void(int)>& mark, const float** d_affine,
                            unsigned int* d_range_bits = nullptr, unsigned flags = 0u);
// d_range_bits != nullptr (sg_layer_layout's range word, or sg_edge_range's): MLP2's conv1 runs on fp16 pieces scaled by that range
// (4 instead of 8 MFMAs per neighbour slot); the word is cleared when the op is done with it.
// d_affine != nullptr: the last BatchNorm + LeakyReLU is NOT applied; d_out holds E = max_k sgn(gamma) y_k and
// d_affine[0..2] receive the device pointers of |a| [64], b' [64] and the last BatchNorm's batch mean | variance [128] (inside d_ws) for a consumer that applies
// LReLU(|a| E + b') itself (segment_max_prefilled does); edgeconv_apply is the epilogue on its own.
int edgeconv_apply(const float* d_e, int N, const float* d_a, const float* d_shift, float* d_dst, void* stream);

// pipeline-internal forms that save launches (kernels_graph.hip, kernels_gcn.hip): group max that also pre-fills the
// columns the point->cluster max writes next; that max without its own fill; GCN with the weight already transposed
int group_max_rows_fill(const float* d_rows, int row_stride, int D, const int32_t* d_goff, const int32_t* d_gidx, int G, float* d_out,
                        int out_stride, int fill_cols, void* stream);
int segment_max_prefilled(const float* d_rows, int N, const int32_t* d_cluster_of_pos, float* d_out, int out_stride, void* stream,
                          const float* d_a = nullptr, const float* d_shift = nullptr);     // rows -> LReLU(a * row + shift) first
int gcn_forward_wt(const float* d_x, int S, int D, const int32_t* d_adj, int E, const int32_t* d_rowptr, const int32_t* d_col,
                   const int32_t* d_eid, const float* d_wt, float alpha, float* d_out, void* d_ws, size_t ws_bytes, void* stream);

// EdgeConv's fold buffer: w1f [64*18] | sh1 [64] | w2f [64*64] | sh2 [64] | S2X's fp16 weight image [4096] | its scales [4]
// k_edge_moments: points per workgroup, in chunks of 256.  Measured (round 5, per launch of 8 scenes): four chunks per block leave the one-block-
// per-scene fold kernel a quarter of the partial rows [189] to add up (k_bn_fold_moments_b 33.5 -> 22.1 us) but take the gather-bound moments
// kernel's latency cover with them (149 -> 224 us): one chunk per block stays.
constexpr int kMomPts = 256;
inline int moments_blocks(int N) { return (N + kMomPts - 1) / kMomPts; }
constexpr int kRangeWords = 256;          // EdgeConv's range: the maximum of this many words (k_layer_layout spreads its atomics over them)
constexpr int kEdgeFoldFloats = 64 * 18 + 64 + 64 * 64 + 64 + 4096 + 4 + 1024;    // ... | conv2' image | scales | conv1' fp16 image
int edge_moments_partials(const float* d_x9m, const int32_t* d_knn, int N, int K, double* d_partial, hipStream_t st);   // kernels_edgeconv.hip, [cdiv(N,256)][189]
int reduce_partials(const double* d_partial, int nblocks, int stride, int count, double* d_out, hipStream_t st);   // kernels_train_edge.hip
int transpose_square(const float* d_src, float* d_dst, int D, hipStream_t st);      // kernels_train.hip
// bytes moved by a KERNEL on `st` (kernels_graph.hip): either side may be pinned host memory (mapped into the device's address space) --
// the engine's small per-phase transfers then do not queue behind bulk copies on the copy engines (engine.cpp).  16-byte aligned.
int copy_by_kernel(void* dst, const void* src, size_t bytes, hipStream_t st);

// device -> pinned-host copies on the copy engines through the HSA runtime (sdma.cpp); the caller falls back to hipMemcpyAsync on any failure
bool sdma_available();
int sdma_copy_d2h(void* const* dst, const void* const* src, const size_t* bytes, int n);
struct SdmaTicket { unsigned long long signal = 0; };
int sdma_issue(void* dst, const void* src, size_t bytes, SdmaTicket* t);     // one copy, either direction between device and pinned host memory
int sdma_wait(SdmaTicket* t);

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Bump allocator over a caller-provided workspace.
struct Carver {
    char* base;
    size_t cap, off = 0;
    bool ok = true;
    Carver(void* p, size_t n) : base((char*)p), cap(n) {}
    template <class T>
    T* take(size_t count) {
        size_t bytes = align_up(count * sizeof(T));
        if (off + bytes > cap) { ok = false; return nullptr; }
        T* r = reinterpret_cast<T*>(base + off);
        off += bytes;
        return r;
    }
};

}  // namespace sg
