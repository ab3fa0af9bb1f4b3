// Private layout of sg_pipeline (shared by pipeline.cpp, the single-scene path, and engine.cpp, which drives several
// pipelines' buffers with batched launches).  Not part of the C ABI.
#pragma once
#include "sg_common.h"

namespace sgp {

constexpr int kNumEvents = 48;
static const char* const kStageNames[] = {"contract_edges", "fps64", "mlp1", "dist1+d2h",
                             "l2.gather", "l2.center", "l2.knn", "l2.edgeconv", "l2.segmax", "l2.gcn+dist",
                             "l3.gather", "l3.center", "l3.knn", "l3.edgeconv", "l3.segmax", "l3.gcn+dist",
                             "fallback_fps1024", "export", "evaluate",
                             // sub-passes of the two EdgeConv stages (their sum is l2.edgeconv / l3.edgeconv)
                             "l2.edgeconv.stats1", "l2.edgeconv.final", "l3.edgeconv.stats1", "l3.edgeconv.stats2",
                             "l3.edgeconv.final",
                             // the scene engine only: the EdgeConv launch ALONE (events right before and right after it; its stage above also
                             // holds the fold and affine launches behind it)
                             "kernel.l2.edgeconv", "kernel.l3.edgeconv"};
constexpr int kNumStages = sizeof(kStageNames) / sizeof(kStageNames[0]);

// A buffer either owns its allocation (alloc) or is a view into its pipeline's arena (view): sg_pipeline_create asks for ~45 device and
// ~10 pinned buffers per pipeline, the engine creates 80 pipelines -- as individual hipMalloc / hipHostMalloc calls that was most of the
// driver's start-up and tear-down (round 4: one device arena and one pinned arena per pipeline).
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool own = false;
    int alloc(size_t count) {
        n = count ? count : 1;
        own = hipMalloc((void**)&p, n * sizeof(T)) == hipSuccess;
        return own ? 0 : -1;
    }
    void view(void* at, size_t count) { p = static_cast<T*>(at); n = count ? count : 1; own = false; }
    ~DevBuf() { if (p && own) (void)hipFree(p); }
};
template <class T>
struct PinBuf {
    T* p = nullptr;
    size_t n = 0;
    bool own = false;
    int alloc(size_t count) {
        n = count ? count : 1;
        own = hipHostMalloc((void**)&p, n * sizeof(T), hipHostMallocDefault) == hipSuccess;
        return own ? 0 : -1;
    }
    void view(void* at, size_t count) { p = static_cast<T*>(at); n = count ? count : 1; own = false; }
    ~PinBuf() { if (p && own) (void)hipHostFree(p); }
};

}  // namespace sgp

struct sg_pipeline {
    int maxN = 0, maxS = 0, maxE = 0, maxV = 0, maxT = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    size_t dev_bytes = 0, pin_bytes = 0;
    char* dev_arena = nullptr;          // every DevBuf / PinBuf below (but the weights) is a view into one of these two allocations
    char* pin_arena = nullptr;
    ~sg_pipeline() {
        if (dev_arena) (void)hipFree(dev_arena);
        if (pin_arena) (void)hipHostFree(pin_arena);
    }

    // weights (device)
    sgp::DevBuf<float> w;   // all parameters, offsets below
    size_t o_m1w, o_m1g, o_m1b, o_m2w, o_m2g, o_m2b, o_g2, o_m3w1, o_m3g1, o_m3b1, o_m3w2, o_m3g2, o_m3b2, o_g3, o_g2t, o_g3t;

    // device work buffers
    sgp::DevBuf<char> ws_contract, ws_fps, ws_mlp1, ws_edge, ws_gcn, ws_eval, ws_sort;
    sgp::DevBuf<int32_t> adj1, count, members, pos_of_point, cluster_of_pos, slot_of_pos, sperm, smpos, seg_chunk_off, knn, knn_seed, seed_id, desc, tables, labels;
    sgp::DevBuf<unsigned int> ec_range;     // one word: the layer's EdgeConv range (k_layer_layout raises it, the layer's last fold kernel clears it)
    sgp::DevBuf<float> samples, samples_big, feat1, featA, featB, dist, x9m, xyzw, pf, segbox, chunk_box, chunk_table, point_rec;

    // pinned host staging
    sgp::PinBuf<int32_t> h_adj, h_desc, h_tables, h_count, h_chunk_off, h_eval;      // h_eval: the metric counters' landing buffer
    sgp::PinBuf<float> h_dist, h_feat, h_samples;
    sgp::DevBuf<double> seg_sums;          // [S,3] coordinate sums of every over-segment (layer-invariant)
    sgp::PinBuf<double> h_seg_sums;

    // The FPS-1024 fallback of the final clustering (model.py:479; an unlabeled, disconnected component: rare) samples into [S,1024,3] floats on
    // the device and on the host: 18 MB each at S = 1,500.  Allocated on first use -- as part of every pipeline's arenas the eighty slots of an
    // engine pinned 1.5 GB for it at start-up (round 5: most of the 0.17 s `sg_engine_create` took).
    int need_fallback_buffers() {
        const size_t n = (size_t)maxS * 1024 * 3;
        if (!samples_big.p && samples_big.alloc(n) != 0) return sg::fail(SG_ENOMEM, "FPS-1024 fallback: device allocation of %zu floats failed", n);
        if (!h_samples.p && h_samples.alloc(n) != 0) return sg::fail(SG_ENOMEM, "FPS-1024 fallback: pinned allocation of %zu floats failed", n);
        return SG_OK;
    }

    hipEvent_t ev[sgp::kNumEvents];
    hipEvent_t ev_count = nullptr;      // behind the D2H of the contracted edge count: the host waits for THAT, not for the whole structural layer
    // label rows leave as soon as their tables exist (round 4): H2D of a layer's table rows, the export kernel for them and the D2H of the vectors run
    // on this second stream while the next layer computes; ev_side orders it behind the scene's start and the metric kernels behind the last export
    hipStream_t side = nullptr;
    hipEvent_t ev_side = nullptr;
    int ev_stage[sgp::kNumEvents];
    int n_ev = 0;
    float stage_ms[sgp::kNumStages];

    // stage timing (sg_pipeline_set_timing): 2 = an event after every stage, 1 = only around the in-cluster kNN and the
    // EdgeConv passes (what bench.py's roofline needs: ~10 instead of ~25 events per scene, which cost ~6 % of the
    // throughput with 16 pipelines in flight), 0 = none
    int timing = 2;
    int knn_variant = -1;        // sg_pipeline_set_knn_variant: < 0 = by tile count
    static bool kernel_stage(int stage) { return stage == 6 || stage == 12 || (stage >= 19 && stage <= 23); }
    void mark_kernel_start() {                                // level 1: the event a timed kernel stage is measured from
        if (timing == 1) record(-1);
    }
    void mark(int stage) {
        if (timing == 2 || (timing == 1 && kernel_stage(stage))) record(stage);
    }
    void record(int stage) {
        if (n_ev < sgp::kNumEvents) {
            (void)hipEventRecord(ev[n_ev], stream);
            ev_stage[n_ev] = stage;
            ++n_ev;
        }
    }
};

namespace sgp {

struct LayerDesc {               // host view of one frozen numbering + what the device needs for it
    int C = 0, T = 0;
    std::vector<int32_t> root, cl_of_seg, order, cl_seg_off, cl_pt_off, dst;
};

inline int freeze_layer(const sg_partition* part, int S, LayerDesc& L) {
    L.root.resize(S); L.cl_of_seg.resize(S); L.order.resize(S); L.cl_seg_off.resize(S + 1); L.cl_pt_off.resize(S + 1); L.dst.resize(S);
    L.C = sg_partition_layer(part, L.root.data(), L.cl_of_seg.data(), L.order.data(), L.cl_seg_off.data(), L.cl_pt_off.data(), L.dst.data());
    return L.C;
}

// Device descriptor block of one layer, carved from ONE pinned buffer and shipped in ONE H2D copy.
struct DescOffsets {
    size_t order, dst, cl, cl_pt_off, cl_seg_off, tile_cl, tile_lo, tile_hi, goff, gidx, adj, rowptr, col, eid,
        slot_chunk0, cl_chunk_off, tile_chunk0, seg_prevcl, cl_mean, total;
};

}  // namespace sgp

// What the training step's backward needs from one forward (trainer.cpp): the pipeline's work buffers are
// reused from layer to layer, so the per-layer operands are copied out (device to device, on the pipeline's stream) while
// the forward runs.  Owned and sized by the trainer; `sg_debug::tape` hands it to sg_pipeline_forward.
struct sg_tape {
    struct Layer {
        sgp::DevBuf<float> x9m, pf, cat, gcn;      // [N,12] centred rows | [N,64] pre-activation extremes | [C,Dcat] GCN input | [C,Dcat] GCN output
        sgp::DevBuf<float> bn_last;                // [128] batch mean | variance of the EdgeConv stack's last BatchNorm (from the forward's fold)
        sgp::DevBuf<int32_t> knn, desc;            // [N,20] | the layer's descriptor block (offsets in `o`)
        sgp::DescOffsets o;
        int C = 0, Cprev = 0, Dcat = 0, Dprev = 0, E = 0;
    } layer[2];
    int N = 0, S = 0;
    // final clusters (after group_unlabeled): row j of layer[1].gcn belongs to final cluster fin_gidx^-1; CSR over final clusters
    std::vector<int32_t> fin_goff, fin_gidx, ins5, sem5;
    std::vector<float> feat5;                       // [C6,256]
    int C6 = 0;
    bool filled = false;
};

