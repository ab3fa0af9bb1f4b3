// SegModel.forward (reference seggroup/model.py:684-897) for one scene on one HIP stream.
//
// The reference alternates device math with serial, order-dependent Python bookkeeping; here the
// bookkeeping is the host grouping engine (grouping.cpp, segment-level, microseconds) and every
// per-point / per-edge computation is a kernel.  Per scene the host blocks on the stream only where
// the serial grouping needs distances: after MLP1, after each of the two semantic layers, optionally
// after the FPS-1024 fallback, and at the end (3-5 synchronisations).  Several pipelines on distinct
// streams (one per in-flight scene) overlap their host phases with each other's kernels.
#include <atomic>
#include <chrono>
#include <cmath>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include <string>

#include "pipeline_priv.h"

namespace {

using namespace sgp;

// SG_HOST_PROFILE=1: wall time of sg_pipeline_forward split into 'blocked in hipStreamSynchronize' and the rest (host work +
// launches), printed per sg_batch_forward call -- a development aid
std::atomic<long long> g_prof_total_ns{0}, g_prof_sync_ns{0}, g_prof_scenes{0}, g_prof_sec[8];
const char* const kProfSec[8] = {"setup+launch0", "regroup", "label tables", "descriptors", "layer launches", "final clustering", "export+eval", "other"};
const bool g_host_profile = getenv("SG_HOST_PROFILE") != nullptr;

}  // namespace

extern "C" {

const char* sg_pipeline_stage_name(int i) { return (i >= 0 && i < kNumStages) ? kStageNames[i] : nullptr; }

int sg_pipeline_stage_times(const sg_pipeline* pl, float* h_ms, int capacity) {
    if (!pl || !h_ms) return sg::fail(SG_EINVAL, "sg_pipeline_stage_times: null argument");
    for (int i = 0; i < kNumStages && i < capacity; ++i) h_ms[i] = pl->stage_ms[i];
    return kNumStages;
}

int sg_pipeline_set_timing(sg_pipeline* pl, int level) {
    if (!pl || level < 0 || level > 2) return sg::fail(SG_EINVAL, "sg_pipeline_set_timing: bad arguments");
    const int prev = pl->timing;
    pl->timing = level;
    return prev;
}

int sg_pipeline_set_knn_variant(sg_pipeline* pl, int variant) {
    if (!pl) return sg::fail(SG_EINVAL, "sg_pipeline_set_knn_variant: null pipeline");
    const int prev = pl->knn_variant;
    pl->knn_variant = (variant == 0 || variant == 1 || variant == 2 || variant == 4 || variant == 8) ? variant : -1;
    return prev;
}

size_t sg_pipeline_device_bytes(const sg_pipeline* pl) { return pl ? pl->dev_bytes : 0; }

void sg_pipeline_destroy(sg_pipeline* pl) {
    if (!pl) return;
    for (int i = 0; i < kNumEvents; ++i) (void)hipEventDestroy(pl->ev[i]);
    if (pl->ev_count) (void)hipEventDestroy(pl->ev_count);
    if (pl->ev_side) (void)hipEventDestroy(pl->ev_side);
    if (pl->side) (void)hipStreamDestroy(pl->side);
    delete pl;
}

sg_pipeline* sg_pipeline_create(int maxN, int maxS, int maxE, int maxV, const sg_weights* wt, void* stream) {
    if (maxN <= 0 || maxS <= 0 || maxE < 0 || maxV <= 0 || !wt) {
        sg::fail(SG_EINVAL, "sg_pipeline_create: bad arguments");
        return nullptr;
    }
    if (maxN > SG_MAX_POINTS) {
        sg::fail(SG_EUNSUP, "sg_pipeline_create: maxN = %d; a scene holds at most %d points (the kNN list keys carry 20 index bits)", maxN, SG_MAX_POINTS);
        return nullptr;
    }
    if (sg_device_count() <= 0) {
        sg::fail(SG_EHIP, "sg_pipeline_create: no HIP device visible -- the SegGroup hot path has no CPU fallback");
        return nullptr;
    }
    std::unique_ptr<sg_pipeline> pl(new sg_pipeline());
    if (hipGetDevice(&pl->device) != hipSuccess) { sg::fail(SG_EHIP, "hipGetDevice failed"); return nullptr; }
    pl->maxN = maxN; pl->maxS = maxS; pl->maxE = maxE; pl->maxV = maxV;
    pl->maxT = maxN / 64 + maxS + 1;
    pl->stream = sg::as_stream(stream);
    for (int i = 0; i < kNumEvents; ++i)
        if (hipEventCreate(&pl->ev[i]) != hipSuccess) { sg::fail(SG_EHIP, "hipEventCreate failed"); return nullptr; }
    if (hipEventCreateWithFlags(&pl->ev_count, hipEventDisableTiming) != hipSuccess) { sg::fail(SG_EHIP, "hipEventCreate failed"); return nullptr; }
    if (hipEventCreateWithFlags(&pl->ev_side, hipEventDisableTiming) != hipSuccess || hipStreamCreateWithFlags(&pl->side, hipStreamNonBlocking) != hipSuccess) {
        sg::fail(SG_EHIP, "sg_pipeline_create: side stream / event creation failed");
        sg_pipeline_destroy(pl.release());
        return nullptr;
    }
    for (float& m : pl->stage_ms) m = 0.f;

    int bad = 0;
    size_t dev = 0, pin = 0;
    // D / P only RECORD a request (the buffer's element count is known at once: later requests read it); the two arenas are allocated when
    // all requests are in and every buffer becomes a 256-byte aligned view (sgp::DevBuf::view)
    struct Req { std::function<void(char*)> set; size_t off; };
    std::vector<Req> dreq, preq;
    auto D = [&](auto& buf, size_t count) {
        count = count ? count : 1;
        buf.n = count;
        auto* b = &buf;
        dreq.push_back({[b, count](char* at) { b->view(at, count); }, dev});
        dev += (count * sizeof(*buf.p) + 255) / 256 * 256;
    };
    auto P = [&](auto& buf, size_t count) {
        count = count ? count : 1;
        buf.n = count;
        auto* b = &buf;
        preq.push_back({[b, count](char* at) { b->view(at, count); }, pin});
        pin += (count * sizeof(*buf.p) + 255) / 256 * 256;
    };

    // weights
    size_t off = 0;
    auto slot = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };
    pl->o_m1w = slot(64 * 6); pl->o_m1g = slot(64); pl->o_m1b = slot(64);
    pl->o_m2w = slot(64 * 18); pl->o_m2g = slot(64); pl->o_m2b = slot(64);
    pl->o_g2 = slot(192 * 192);
    pl->o_m3w1 = slot(64 * 18); pl->o_m3g1 = slot(64); pl->o_m3b1 = slot(64);
    pl->o_m3w2 = slot(64 * 64); pl->o_m3g2 = slot(64); pl->o_m3b2 = slot(64);
    pl->o_g3 = slot(256 * 256);
    pl->o_g2t = slot(192 * 192); pl->o_g3t = slot(256 * 256);        // W^T of the two GCN layers (k_gcn_fc reads [k][o])
    bad |= pl->w.alloc(off);                             // (owned: filled right below)
    if (!bad) {
        std::vector<float> hw(off, 0.f);
        auto put = [&](size_t o, const float* src, size_t n) { if (src) std::copy(src, src + n, hw.begin() + o); else bad = 1; };
        put(pl->o_m1w, wt->mlp1_w, 64 * 6); put(pl->o_m1g, wt->mlp1_g, 64); put(pl->o_m1b, wt->mlp1_b, 64);
        put(pl->o_m2w, wt->mlp2_w, 64 * 18); put(pl->o_m2g, wt->mlp2_g, 64); put(pl->o_m2b, wt->mlp2_b, 64);
        put(pl->o_g2, wt->gcn2_w, 192 * 192);
        put(pl->o_m3w1, wt->mlp3_w1, 64 * 18); put(pl->o_m3g1, wt->mlp3_g1, 64); put(pl->o_m3b1, wt->mlp3_b1, 64);
        put(pl->o_m3w2, wt->mlp3_w2, 64 * 64); put(pl->o_m3g2, wt->mlp3_g2, 64); put(pl->o_m3b2, wt->mlp3_b2, 64);
        put(pl->o_g3, wt->gcn3_w, 256 * 256);
        auto put_t = [&](size_t o, const float* src, int D) {
            if (!src) { bad = 1; return; }
            for (int r = 0; r < D; ++r) for (int k = 0; k < D; ++k) hw[o + (size_t)k * D + r] = src[(size_t)r * D + k];
        };
        put_t(pl->o_g2t, wt->gcn2_w, 192); put_t(pl->o_g3t, wt->gcn3_w, 256);
        if (bad) { sg::fail(SG_EINVAL, "sg_pipeline_create: null weight pointer"); return nullptr; }
        if (hipMemcpy(pl->w.p, hw.data(), off * 4, hipMemcpyHostToDevice) != hipSuccess) { sg::fail(SG_EHIP, "weight upload failed"); return nullptr; }
    }

    const size_t S = maxS, N = maxN, E = maxE, V = maxV, T = pl->maxT;
    const size_t maxE1 = std::min<size_t>(E, S * (S - 1) / 2 + 1);
    D(pl->ws_contract, sg_contract_ws_bytes(maxS));
    D(pl->ws_fps, sg_fps_ws_bytes(maxN));
    D(pl->ws_mlp1, sg_mlp1_ws_bytes(maxS));
    D(pl->ws_edge, sg_edgeconv_ws_bytes(maxN));
    D(pl->ws_gcn, sg_gcn_ws_bytes(maxS, 256, (int)maxE1));
    D(pl->ws_eval, sg_eval_ws_bytes(maxS + 2));
    D(pl->adj1, 2 * maxE1); D(pl->count, 4);
    D(pl->members, N); D(pl->pos_of_point, N); D(pl->cluster_of_pos, N); D(pl->slot_of_pos, N);
    D(pl->knn, N * 20); D(pl->knn_seed, N * 20); D(pl->seed_id, N); D(pl->ec_range, sg::kRangeWords);
    D(pl->desc, 15 * S + 16 + 4 * T + 2 * maxE1 + 4 * maxE1 + 128);
    D(pl->tables, SG_NUM_LABEL_VECTORS * S); D(pl->labels, SG_NUM_LABEL_VECTORS * V);
    D(pl->samples, S * 64 * 6);                                 // (samples_big / h_samples: sg_pipeline::need_fallback_buffers, on first use)
    D(pl->feat1, S * 128); D(pl->featA, S * 256); D(pl->featB, S * 256);
    D(pl->seg_sums, S * 3); P(pl->h_seg_sums, S * 3); D(pl->segbox, S * 8); D(pl->chunk_box, (N / 32 + S + 1) * 8); D(pl->chunk_table, (N / 32 + S + 1) * 8); D(pl->sperm, N); D(pl->smpos, N); D(pl->seg_chunk_off, S + 1);
    D(pl->ws_sort, sg_segment_sort_ws_bytes(maxN)); D(pl->dist, maxE1); D(pl->x9m, N * 12); D(pl->xyzw, N * 4); D(pl->pf, N * 64); D(pl->point_rec, N * 4);
    P(pl->h_adj, 2 * maxE1); P(pl->h_desc, pl->desc.n); P(pl->h_tables, SG_NUM_LABEL_VECTORS * S); P(pl->h_count, 4); P(pl->h_chunk_off, S + 1); P(pl->h_eval, pl->ws_eval.n / 4 + 16);
    P(pl->h_dist, maxE1); P(pl->h_feat, S * 256);
    if (!bad && hipMalloc((void**)&pl->dev_arena, dev) != hipSuccess) bad = 1;
    if (!bad && hipHostMalloc((void**)&pl->pin_arena, pin, hipHostMallocDefault) != hipSuccess) bad = 1;
    if (bad) { sg::fail(SG_ENOMEM, "sg_pipeline_create: device/pinned allocation failed (N=%d S=%d E=%d V=%d)", maxN, maxS, maxE, maxV); return nullptr; }
    for (auto& r : dreq) r.set(pl->dev_arena + r.off);
    for (auto& r : preq) r.set(pl->pin_arena + r.off);
    dev += pl->w.n * sizeof(float);
    if (hipMemset(pl->ec_range.p, 0, sg::kRangeWords * sizeof(unsigned int)) != hipSuccess) { sg::fail(SG_EHIP, "sg_pipeline_create: hipMemset failed"); return nullptr; }
    pl->dev_bytes = dev; pl->pin_bytes = pin;
    return pl.release();
}

static hipError_t timed_sync(hipStream_t st) {
    if (!g_host_profile) return hipStreamSynchronize(st);
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipStreamSynchronize(st);
    g_prof_sync_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return e;
}

static hipError_t timed_event_sync(hipEvent_t ev) {
    if (!g_host_profile) return hipEventSynchronize(ev);
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipEventSynchronize(ev);
    g_prof_sync_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return e;
}

#define PL_CHECK(call) do { int rc__ = (call); if (rc__ < 0) { sg_partition_destroy(part); return rc__; } } while (0)
#define PL_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { sg_partition_destroy(part); \
    return sg::fail(SG_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); } } while (0)

// Small transfers between the pipeline's pinned arena and the device go through a KERNEL on the scene's stream (sg::copy_by_kernel; pinned
// memory is mapped into the device's address space): on the copy engines each of them cost a queue round trip of its own, and behind bulk
// copies of other streams much more (engine.cpp, arena_copy).  SG_ENGINE_COPY=sdma restores hipMemcpyAsync.  Bulk data (the label vectors)
// stays on the copy engines.
static bool pl_sdma() { static const bool v = getenv("SG_ENGINE_COPY") && std::string(getenv("SG_ENGINE_COPY")) == "sdma"; return v; }
#define PL_COPY(dst, src, bytes, kind, stream_)                                                                              \
    do {                                                                                                                     \
        if (pl_sdma()) { PL_HIP(hipMemcpyAsync((dst), (src), (bytes), (kind), (stream_))); }                                 \
        else { PL_CHECK(sg::copy_by_kernel((void*)(dst), (const void*)(src), (size_t)(bytes), (stream_))); }                 \
    } while (0)

int sg_pipeline_forward(sg_pipeline* pl, const sg_scene* sc, int mode, sg_result* out, sg_debug* dbg) {
    if (!pl || !sc || !out) return sg::fail(SG_EINVAL, "sg_pipeline_forward: null argument");
    SG_REQUIRE(mode == SG_MODE_INS_INFER || mode == SG_MODE_SEM_INFER, "sg_pipeline_forward: bad mode %d", mode);
    const int N = sc->N, S = sc->S, E0 = sc->E0, V = sc->V;
    SG_REQUIRE(N > 0 && S > 0 && V > 0 && E0 >= 0, "sg_pipeline_forward: empty scene");
    SG_REQUIRE(N <= pl->maxN && S <= pl->maxS && E0 <= pl->maxE && V <= pl->maxV,
               "sg_pipeline_forward: scene (N=%d S=%d E0=%d V=%d) exceeds the pipeline capacity (N=%d S=%d E0=%d V=%d)", N, S, E0, V,
               pl->maxN, pl->maxS, pl->maxE, pl->maxV);
    SG_REQUIRE(out->h_labels, "sg_pipeline_forward: out->h_labels is null");
    SG_HIP(hipSetDevice(pl->device));                     // the calling thread may be a fresh worker thread
    struct ProfScope {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~ProfScope() {
            if (!g_host_profile) return;
            g_prof_total_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            ++g_prof_scenes;
        }
    } prof_scope;
    auto prof_last = std::chrono::steady_clock::now();
    long long prof_sync_seen = g_host_profile ? 0 : 0;
    (void)prof_sync_seen;
    auto lap = [&](int sec) {                                // host-profile: time since the previous lap goes to section `sec` (< 0: dropped)
        if (!g_host_profile) return;
        const auto now = std::chrono::steady_clock::now();
        if (sec >= 0) g_prof_sec[sec] += std::chrono::duration_cast<std::chrono::nanoseconds>(now - prof_last).count();
        prof_last = now;
    };
    hipStream_t st = pl->stream;
    void* stv = (void*)st;
    sg_tape* tape = dbg ? reinterpret_cast<sg_tape*>(dbg->tape) : nullptr;
    if (tape) tape->filled = false;
    SG_REQUIRE(!tape || mode == SG_MODE_INS_INFER, "the training tape needs the full (ins_infer) forward");
    const float* W = pl->w.p;
    pl->n_ev = 0;
    out->stalled = 0; out->used_fallback = 0;
    for (int i = 0; i < 5; ++i) out->trace[i] = 0;

    // (the partition itself -- S member lists -- is built further down, behind the first launches: the GPU is idle until those are queued,
    // and one scene alone pays for every microsecond the host spends in front of them)
    sg_partition* part = nullptr;
    int max_ins = 1;
    for (int s = 0; s < S; ++s) max_ins = std::max(max_ins, sc->h_seg_ins[s] + 2);
    if (sg_eval_ws_bytes(max_ins) > pl->ws_eval.n)
        return sg::fail(SG_EUNSUP, "weak instance ids up to %d exceed the pipeline's metric workspace (max_segments + 2)", max_ins - 2);

    int32_t* tab = pl->h_tables.p;                       // [14,S]
    // The label vectors of a layer leave as soon as its tables exist: table rows H2D, k_export for those rows, vectors D2H -- all on the
    // pipeline's second stream, beside the next layer's kernels (one scene alone used to wait for 8.4 MB / 28 MB of labels to cross PCIe
    // behind its last kernel: 0.17 / 0.55 ms at 150k / 500k points).  The rows of h_tables / tables / labels of different layers are disjoint.
    hipStream_t side = pl->side;
    struct SideGuard {                                     // an early return must not leave copies into out->h_labels in flight
        hipStream_t s;
        ~SideGuard() { (void)hipStreamSynchronize(s); }
    } side_guard{side};
    SG_HIP(hipEventRecord(pl->ev_side, st));               // the scene's arrays are ready for `st`: so they are for `side`
    SG_HIP(hipStreamWaitEvent(side, pl->ev_side, 0));
    // tables_for() only fills the host rows (the partition changes with the next grouping pass); the second stream's calls -- ~20 API calls per
    // scene -- are issued by flush_exports() right in front of the NEXT stream sync, i.e. while the layer just launched runs (issued where the
    // tables are made they cost a scene 0.24 ms of host time with the GPU idle)
    int pend_row[5], pend_rows[5], npend = 0;
    auto tables_for = [&](int first_row, bool with_seg) -> int {
        int32_t* a = tab + (size_t)first_row * S;
        const int rc = sg_partition_export_tables(part, with_seg ? a : nullptr, with_seg ? a + S : a, with_seg ? a + 2 * (size_t)S : a + S);
        if (rc < 0) return rc;
        pend_row[npend] = first_row; pend_rows[npend] = with_seg ? 3 : 2; ++npend;
        return SG_OK;
    };
    auto flush_exports = [&](bool last) -> int {
        for (int i = 0; i + 1 < npend;) {                   // consecutive row ranges leave as one (layer_4.* + final.* at the end: five rows, three calls)
            if (pend_row[i] + pend_rows[i] == pend_row[i + 1]) {
                pend_rows[i] += pend_rows[i + 1];
                for (int k = i + 1; k + 1 < npend; ++k) { pend_row[k] = pend_row[k + 1]; pend_rows[k] = pend_rows[k + 1]; }
                --npend;
            } else ++i;
        }
        for (int i = 0; i < npend; ++i) {
            const int rows = pend_rows[i];
            const size_t t0 = (size_t)pend_row[i] * S, l0 = (size_t)pend_row[i] * V;
            if (hipMemcpyAsync(pl->tables.p + t0, tab + t0, (size_t)rows * S * 4, hipMemcpyHostToDevice, side) != hipSuccess)
                return sg::fail(SG_EHIP, "label tables: H2D failed");
            const int rc2 = sg_export_labels(sc->d_unmap, V, sc->d_seg_of_point, N, pl->tables.p + t0, rows, S, pl->labels.p + l0, (void*)side);
            if (rc2 < 0) return rc2;
            if (last && i == npend - 1 && hipEventRecord(pl->ev_side, side) != hipSuccess)       // behind the LAST export kernel: the metric kernels wait for it
                return sg::fail(SG_EHIP, "label tables: event record failed");
            if (hipMemcpyAsync(out->h_labels + l0, pl->labels.p + l0, (size_t)rows * V * 4, hipMemcpyDeviceToHost, side) != hipSuccess)
                return sg::fail(SG_EHIP, "label vectors: D2H failed");
        }
        npend = 0;
        return SG_OK;
    };

    // ---------------- graph initialisation + structural grouping layer (model.py:710-783) ---------------
    const int cap1 = (int)(pl->adj1.n / 2);
    pl->mark(-1);
    PL_CHECK(sg_contract_point_edges(sc->d_adj, E0, sc->d_seg_of_point, N, S, pl->adj1.p, cap1, pl->count.p, pl->ws_contract.p,
                                     pl->ws_contract.n, stv));
    PL_COPY(pl->h_count.p, pl->count.p, 4, hipMemcpyDeviceToHost, st);
    PL_HIP(hipEventRecord(pl->ev_count, st));
    pl->mark(0);
    int max_seg = 0;
    for (int s = 0; s < S; ++s) max_seg = std::max(max_seg, sc->h_seg_size[s]);
    PL_CHECK(sg::fps_sample_hint(sc->d_data, N, 6, sc->d_seg_points, sc->d_seg_off, S, 64, 6, 1, pl->samples.p, nullptr, pl->ws_fps.p,
                                 pl->ws_fps.n, stv, max_seg));
    if (mode == SG_MODE_INS_INFER) {
        // once per scene: segment boxes, Morton order inside every over-segment + boxes of its 32-point chunks (kNN
        // pruning) -- one launch unless a segment exceeds a block's LDS
        int32_t* co = pl->h_chunk_off.p;
        co[0] = 0;
        for (int s = 0; s < S; ++s) co[s + 1] = co[s] + (sc->h_seg_size[s] + 31) / 32;
        PL_COPY(pl->seg_chunk_off.p, co, (size_t)(S + 1) * 4, hipMemcpyHostToDevice, st);
        PL_CHECK(sg_segment_sort_boxes(sc->d_data, N, sc->d_seg_points, sc->d_seg_off, sc->d_seg_of_point, S, pl->seg_chunk_off.p, max_seg,
                                       pl->segbox.p, pl->sperm.p, pl->chunk_box.p, pl->seg_sums.p, pl->ws_sort.p, pl->ws_sort.n, stv));
        PL_COPY(pl->h_seg_sums.p, pl->seg_sums.p, (size_t)S * 3 * 8, hipMemcpyDeviceToHost, st);     // ready at the sync below
    }
    pl->mark(1);
    PL_CHECK(sg_mlp1_forward(pl->samples.p, S, W + pl->o_m1w, W + pl->o_m1g, W + pl->o_m1b, pl->feat1.p, 128, pl->ws_mlp1.p,
                             pl->ws_mlp1.n, stv));
    pl->mark(2);
    part = sg_partition_create(S, sc->h_seg_first, sc->h_seg_size, sc->h_seg_ins, sc->h_seg_sem);
    if (!part) return SG_EINVAL;
    lap(0);
    // only the edge count is needed here, and it has been on the host since the first kernel finished: the sampling / sorting / MLP1 launches
    // above keep the GPU busy while the distance launch and its copies are queued behind them (one full stream sync less per scene)
    PL_HIP(timed_event_sync(pl->ev_count));
    lap(-1);
    int E1 = pl->h_count.p[0];
    if (E1 > cap1) { sg_partition_destroy(part); return sg::fail(SG_ENOMEM, "adjacency capacity exceeded (%d > %d)", E1, cap1); }
    PL_CHECK(sg_edge_distance(pl->feat1.p, 128, 128, pl->adj1.p, E1, pl->dist.p, stv));
    PL_COPY(pl->h_adj.p, pl->adj1.p, (size_t)E1 * 8, hipMemcpyDeviceToHost, st);
    PL_COPY(pl->h_dist.p, pl->dist.p, (size_t)E1 * 4, hipMemcpyDeviceToHost, st);
    if (dbg) {
        if (dbg->d_samples1) PL_HIP(hipMemcpyAsync(dbg->d_samples1, pl->samples.p, (size_t)S * 64 * 6 * 4, hipMemcpyDeviceToDevice, st));
        if (dbg->d_feat1) PL_HIP(hipMemcpyAsync(dbg->d_feat1, pl->feat1.p, (size_t)S * 128 * 4, hipMemcpyDeviceToDevice, st));
    }
    pl->mark(3);

    LayerDesc Lcur, Lnew;
    freeze_layer(part, S, Lcur);                          // layer 1: every segment its own cluster
    out->trace[0] = Lcur.C;
    PL_CHECK(tables_for(0, true));                        // layer_1.{seg,ins,sem}
    PL_CHECK(flush_exports(false));
    lap(2);
    PL_HIP(timed_sync(st));
    lap(-1);

    std::vector<int32_t> adj(pl->h_adj.p, pl->h_adj.p + 2 * (size_t)E1), adj_next;
    std::vector<uint8_t> connected, keep;
    int E = E1;
    auto tap_adj = [&](int i, const std::vector<int32_t>& a, int rows) {
        if (dbg && dbg->h_adj[i]) std::copy(a.begin(), a.begin() + 2 * (size_t)rows, dbg->h_adj[i]);
        if (dbg) dbg->n_adj[i] = rows;
    };
    auto tap_dist = [&](int i, int rows) { if (dbg && dbg->h_dist[i]) std::copy(pl->h_dist.p, pl->h_dist.p + rows, dbg->h_dist[i]); };
    tap_adj(0, adj, E);
    tap_dist(0, E);

    // group + re-index + contract; returns the new layer in Lnew and the contracted adjacency in adj
    auto regroup = [&](float th) -> int {
        connected.assign(std::max(E, 1), 0);
        int rc = sg_partition_group_nearby(part, Lcur.root.data(), Lcur.C, pl->h_dist.p, adj.data(), E, th, connected.data());
        if (rc == SG_ESTALL) { out->stalled = 1; rc = SG_OK; sg::err_buf()[0] = 0; }   // downgraded: no stale message stays behind
        if (rc < 0) return rc;
        keep.resize(connected.size());
        for (size_t i = 0; i < connected.size(); ++i) keep[i] = !connected[i];
        adj_next.resize(2 * (size_t)std::max(E, 1));
        const int En = sg_partition_contract(part, Lcur.root.data(), adj.data(), E, keep.data(), adj_next.data());
        if (En < 0) return En;
        freeze_layer(part, S, Lnew);
        adj.assign(adj_next.begin(), adj_next.begin() + 2 * (size_t)En);
        E = En;
        return SG_OK;
    };

    PL_CHECK(regroup(mode == SG_MODE_SEM_INFER ? 3.0f : 6.0f));
    lap(1);
    out->trace[1] = Lnew.C;
    PL_CHECK(tables_for(3, true));                        // layer_2.*
    lap(2);
    tap_adj(1, adj, E);

    int n_tables = 6;
    int ins_row = 4, sem_row = 5;

    if (mode == SG_MODE_INS_INFER) {
        // ---------------- semantic grouping layers (model.py:786-865) ----------------------------------
        const float* feat_prev = pl->feat1.p;             // features of the PREVIOUS numbering (rows = Lcur clusters)
        int feat_prev_stride = 128, feat_prev_dim = 128;
        float* cat = pl->featA.p;
        float* gcn_out = pl->featB.p;
        bool have_seed = false;
        for (int layer = 0; layer < 2; ++layer) {
            const int C = Lnew.C, Dcat = feat_prev_dim + 64;
            const int sb = 4 + 6 * layer;                 // stage index base
            pl->mark(-1);
            // ---- descriptor block ----
            std::vector<int32_t> tile_cl, tile_lo, tile_hi, cl_tile_off(C + 1);
            for (int c = 0; c < C; ++c) {
                cl_tile_off[c] = (int)tile_cl.size();
                for (int lo = Lnew.cl_pt_off[c]; lo < Lnew.cl_pt_off[c + 1]; lo += 64) {   // 64 queries per tile (kNN v4)
                    tile_cl.push_back(c); tile_lo.push_back(lo); tile_hi.push_back(std::min(lo + 64, Lnew.cl_pt_off[c + 1]));
                }
            }
            cl_tile_off[C] = (int)tile_cl.size();
            const int T = (int)tile_cl.size();
            // cluster-ordered chunk table of the two-pass kNN: chunk numbers per slot / cluster, and per tile the
            // (cluster-relative) chunk that holds its first sorted position
            int knn_variant = sg::knn_variant_for(T, pl->knn_variant);
            // seeded (8): layer 3 starts from layer 2's table; a former cluster of <= 20 points has no kNN list (-1)
            const bool seeded = knn_variant == 8 && layer == 1 && have_seed;
            const int waves_per_tile = pl->knn_variant == 1 || pl->knn_variant == 2 || pl->knn_variant == 4 ? pl->knn_variant : 0;
            if (knn_variant == 8) knn_variant = 1;
            std::vector<int32_t> seg_prevcl(seeded ? S : 0);
            for (int sg = 0; seeded && sg < S; ++sg) {
                const int pc = Lcur.cl_of_seg[sg];
                seg_prevcl[sg] = Lcur.cl_pt_off[pc + 1] - Lcur.cl_pt_off[pc] > 20 ? pc : -1;
            }
            std::vector<int32_t> slot_chunk0(S + 1, 0), cl_chunk_off(C + 1, 0), tile_chunk0(T, 0);
            if (knn_variant == 0) {
                for (int i = 0; i < S; ++i) slot_chunk0[i + 1] = slot_chunk0[i] + (sc->h_seg_size[Lnew.order[i]] + 31) / 32;
                for (int c = 0; c <= C; ++c) cl_chunk_off[c] = slot_chunk0[Lnew.cl_seg_off[c]];
                for (int c = 0; c < C; ++c) {
                    int slot = Lnew.cl_seg_off[c];
                    for (int tt = cl_tile_off[c]; tt < cl_tile_off[c + 1]; ++tt) {
                        const int pos = tile_lo[tt];
                        while (slot + 1 < Lnew.cl_seg_off[c + 1] && Lnew.dst[slot + 1] <= pos) ++slot;
                        tile_chunk0[tt] = slot_chunk0[slot] + (pos - Lnew.dst[slot]) / 32 - cl_chunk_off[c];
                    }
                }
            }
            // parents: old clusters (Lcur numbering) absorbed by each new cluster, in old order (model.py:766-768)
            std::vector<int32_t> goff(C + 1, 0), gidx(Lcur.C), cl_of_order(S);
            for (int j = 0; j < Lcur.C; ++j) ++goff[Lnew.cl_of_seg[Lcur.root[j]] + 1];
            for (int c = 0; c < C; ++c) goff[c + 1] += goff[c];
            {
                std::vector<int32_t> fill(goff.begin(), goff.end() - 1);
                for (int j = 0; j < Lcur.C; ++j) gidx[fill[Lnew.cl_of_seg[Lcur.root[j]]]++] = j;
            }
            for (int i = 0; i < S; ++i) cl_of_order[i] = Lnew.cl_of_seg[Lnew.order[i]];
            // cluster centroids (combine_centralized_pointcloud, model.py:429-436) from the per-segment coordinate sums:
            // the sum is carried in double, so its grouping does not show in the fp32 result
            std::vector<int32_t> cl_mean_bits(3 * (size_t)C);
            for (int c = 0; c < C; ++c) {
                double sx = 0.0, sy = 0.0, sz = 0.0;
                for (int i = Lnew.cl_seg_off[c]; i < Lnew.cl_seg_off[c + 1]; ++i) {
                    const double* q = pl->h_seg_sums.p + 3 * (size_t)Lnew.order[i];
                    sx += q[0]; sy += q[1]; sz += q[2];
                }
                const double cnt = (double)(Lnew.cl_pt_off[c + 1] - Lnew.cl_pt_off[c]);
                const float m[3] = {(float)(sx / cnt), (float)(sy / cnt), (float)(sz / cnt)};
                std::memcpy(&cl_mean_bits[3 * (size_t)c], m, 12);
            }
            // symmetric CSR of the cluster graph
            std::vector<int32_t> rowptr(C + 1, 0), col(2 * (size_t)E), eid(2 * (size_t)E);
            for (int e = 0; e < E; ++e) { ++rowptr[adj[2 * e] + 1]; ++rowptr[adj[2 * e + 1] + 1]; }
            for (int c = 0; c < C; ++c) rowptr[c + 1] += rowptr[c];
            {
                std::vector<int32_t> fill(rowptr.begin(), rowptr.end() - 1);
                for (int e = 0; e < E; ++e) {
                    const int a = adj[2 * e], b = adj[2 * e + 1];
                    col[fill[a]] = b; eid[fill[a]++] = e;
                    col[fill[b]] = a; eid[fill[b]++] = e;
                }
            }
            DescOffsets o;
            size_t cur = 0;
            auto put = [&](const std::vector<int32_t>& v, size_t count) {
                const size_t at = cur;
                std::copy(v.begin(), v.begin() + count, pl->h_desc.p + at);
                cur += (count + 3) / 4 * 4;
                return at;
            };
            o.order = put(Lnew.order, S); o.dst = put(Lnew.dst, S); o.cl = put(cl_of_order, S); o.cl_pt_off = put(Lnew.cl_pt_off, C + 1);
            o.cl_seg_off = put(Lnew.cl_seg_off, C + 1);
            o.tile_cl = put(tile_cl, T); o.tile_lo = put(tile_lo, T); o.tile_hi = put(tile_hi, T);
            o.seg_prevcl = put(seg_prevcl, seg_prevcl.size());
            o.cl_mean = put(cl_mean_bits, cl_mean_bits.size());
            o.slot_chunk0 = put(slot_chunk0, S + 1); o.cl_chunk_off = put(cl_chunk_off, C + 1); o.tile_chunk0 = put(tile_chunk0, T);
            o.goff = put(goff, C + 1); o.gidx = put(gidx, Lcur.C); o.adj = put(adj, 2 * (size_t)E);
            o.rowptr = put(rowptr, C + 1); o.col = put(col, 2 * (size_t)E); o.eid = put(eid, 2 * (size_t)E);
            o.total = cur;
            if (o.total > pl->desc.n) { sg_partition_destroy(part); return sg::fail(SG_ENOMEM, "descriptor buffer too small"); }
            lap(3);
            PL_COPY(pl->desc.p, pl->h_desc.p, o.total * 4, hipMemcpyHostToDevice, st);
            const int32_t* dd = pl->desc.p;

            // member arrays + centred rows + sorted kNN operands of the layer: one launch
            PL_CHECK(sg_layer_layout(sc->d_data, N, sc->d_seg_points, sc->d_seg_off, pl->sperm.p, S, dd + o.order, dd + o.dst, dd + o.cl,
                                     reinterpret_cast<const float*>(dd + o.cl_mean), pl->members.p, pl->pos_of_point.p, pl->cluster_of_pos.p,
                                     pl->slot_of_pos.p, pl->x9m.p, pl->xyzw.p, pl->smpos.p, pl->point_rec.p, pl->seed_id.p, pl->ec_range.p, stv));
            // + -inf into the 64 columns the point->cluster max fills below
            PL_CHECK(sg::group_max_rows_fill(feat_prev, feat_prev_stride, feat_prev_dim, dd + o.goff, dd + o.gidx, C, cat, Dcat, 64, stv));
            pl->mark(sb + 0);
            pl->mark(sb + 1);                              // (centring is part of the layout kernel now)
            // point 0 is the first member of segment 0; its member-order position is that segment's dst
            int pos0 = 0;
            for (int i = 0; i < S; ++i) if (Lnew.order[i] == 0) { pos0 = Lnew.dst[i]; break; }
            pl->mark_kernel_start();
            if (knn_variant == 0) {
                PL_CHECK(sg_knn_chunk_table(dd + o.order, dd + o.dst, sc->d_seg_off, pl->seg_chunk_off.p, pl->chunk_box.p, S,
                                            dd + o.slot_chunk0, pl->chunk_table.p, stv));
                PL_CHECK(sg_cluster_knn_2pass(pl->xyzw.p, pl->smpos.p, N, dd + o.cl_pt_off, dd + o.tile_cl, dd + o.tile_lo, dd + o.tile_hi,
                                              dd + o.tile_chunk0, T, dd + o.cl_chunk_off, pl->chunk_table.p, 20, pos0, pl->knn.p, stv));
            } else if (seeded) {
                PL_CHECK(sg_cluster_knn_seeded(pl->xyzw.p, pl->smpos.p, N, dd + o.cl_pt_off, dd + o.tile_cl, dd + o.tile_lo, dd + o.tile_hi, T,
                                               dd + o.cl_seg_off, dd + o.order, dd + o.dst, sc->d_seg_off, pl->seg_chunk_off.p, pl->segbox.p,
                                               pl->chunk_box.p, pl->slot_of_pos.p, pl->knn_seed.p, dd + o.seg_prevcl, pl->seed_id.p,
                                               pl->point_rec.p, 20, pos0, pl->knn.p, stv));
            } else {
                PL_CHECK(sg_cluster_knn_sorted_w(pl->xyzw.p, pl->smpos.p, N, dd + o.cl_pt_off, dd + o.tile_cl, dd + o.tile_lo, dd + o.tile_hi, T,
                                                 dd + o.cl_seg_off, dd + o.order, dd + o.dst, sc->d_seg_off, pl->seg_chunk_off.p, pl->segbox.p,
                                                 pl->chunk_box.p, pl->slot_of_pos.p, 20, pos0, waves_per_tile, pl->knn.p, stv));
            }
            if (layer == 0) {                              // the next kNN layer may start from this table
                PL_CHECK(sg_knn_seed_points(pl->knn.p, pl->seed_id.p, N, 20, pl->knn_seed.p, stv));
                have_seed = true;
            }
            pl->mark(sb + 2);
            // sub-pass marks: the last pass is marked with the stage id itself, so "lN.edgeconv" keeps meaning the
            // time of the LAST pass here and the reporting side adds the sub-passes up (see sg_pipeline_stage_times)
            const int sub0 = layer == 0 ? 19 : 21;
            auto mark_pass = [&](int i) { pl->mark(sub0 + i); };
            // pf = pre-activation maxima; the last BN + LeakyReLU is applied inside the segment max (affine[0..1] = |a|, b')
            const float* affine[3];
            if (layer == 0)
                PL_CHECK(sg::edgeconv_forward_marked(pl->x9m.p, pl->knn.p, N, 20, 1, W + pl->o_m2w, W + pl->o_m2g, W + pl->o_m2b, nullptr,
                                                     nullptr, nullptr, pl->pf.p, pl->ws_edge.p, pl->ws_edge.n, stv, mark_pass, affine, pl->ec_range.p));
            else
                PL_CHECK(sg::edgeconv_forward_marked(pl->x9m.p, pl->knn.p, N, 20, 2, W + pl->o_m3w1, W + pl->o_m3g1, W + pl->o_m3b1,
                                                     W + pl->o_m3w2, W + pl->o_m3g2, W + pl->o_m3b2, pl->pf.p, pl->ws_edge.p,
                                                     pl->ws_edge.n, stv, mark_pass, affine, pl->ec_range.p));
            PL_CHECK(sg::segment_max_prefilled(pl->pf.p, N, pl->cluster_of_pos.p, cat + feat_prev_dim, Dcat, stv, affine[0], affine[1]));
            pl->mark(sb + 4);
            PL_CHECK(sg::gcn_forward_wt(cat, C, Dcat, dd + o.adj, E, dd + o.rowptr, dd + o.col, dd + o.eid, W + (layer == 0 ? pl->o_g2t : pl->o_g3t),
                                    0.125f, gcn_out, pl->ws_gcn.p, pl->ws_gcn.n, stv));
            PL_CHECK(sg_edge_distance(gcn_out, Dcat, Dcat, dd + o.adj, E, pl->dist.p, stv));
            PL_COPY(pl->h_dist.p, pl->dist.p, (size_t)E * 4, hipMemcpyDeviceToHost, st);
            if (layer == 1 || (dbg && dbg->h_gcn[layer]))
                PL_COPY(pl->h_feat.p, gcn_out, (size_t)C * Dcat * 4, hipMemcpyDeviceToHost, st);
            if (dbg) {
                if (dbg->d_pointfeat[layer]) PL_CHECK(sg::edgeconv_apply(pl->pf.p, N, affine[0], affine[1], dbg->d_pointfeat[layer], stv));
                if (dbg->d_knn[layer]) PL_HIP(hipMemcpyAsync(dbg->d_knn[layer], pl->knn.p, (size_t)N * 20 * 4, hipMemcpyDeviceToDevice, st));
                if (dbg->d_members[layer]) PL_HIP(hipMemcpyAsync(dbg->d_members[layer], pl->members.p, (size_t)N * 4, hipMemcpyDeviceToDevice, st));
            }
            if (tape) {
                sg_tape::Layer& TL = tape->layer[layer];
                TL.o = o; TL.C = C; TL.Cprev = Lcur.C; TL.Dcat = Dcat; TL.Dprev = feat_prev_dim; TL.E = E;
                if ((size_t)N * 12 > TL.x9m.n || (size_t)N * 20 > TL.knn.n || (size_t)N * 64 > TL.pf.n || o.total > TL.desc.n || (size_t)C * Dcat > TL.cat.n ||
                    (size_t)C * Dcat > TL.gcn.n) {
                    sg_partition_destroy(part);
                    return sg::fail(SG_ENOMEM, "training tape smaller than the scene");
                }
                PL_HIP(hipMemcpyAsync(TL.x9m.p, pl->x9m.p, (size_t)N * 12 * 4, hipMemcpyDeviceToDevice, st));
                PL_HIP(hipMemcpyAsync(TL.knn.p, pl->knn.p, (size_t)N * 20 * 4, hipMemcpyDeviceToDevice, st));
                PL_HIP(hipMemcpyAsync(TL.pf.p, pl->pf.p, (size_t)N * 64 * 4, hipMemcpyDeviceToDevice, st));
                PL_HIP(hipMemcpyAsync(TL.desc.p, pl->desc.p, o.total * 4, hipMemcpyDeviceToDevice, st));
                PL_HIP(hipMemcpyAsync(TL.cat.p, cat, (size_t)C * Dcat * 4, hipMemcpyDeviceToDevice, st));
                PL_HIP(hipMemcpyAsync(TL.gcn.p, gcn_out, (size_t)C * Dcat * 4, hipMemcpyDeviceToDevice, st));
                PL_HIP(hipMemcpyAsync(TL.bn_last.p, affine[2], 128 * 4, hipMemcpyDeviceToDevice, st));
            }
            pl->mark(sb + 5);
            PL_CHECK(flush_exports(false));                  // the finished layers' label rows, beside this layer's kernels
            lap(4);
            PL_HIP(timed_sync(st));
            lap(-1);
            if (dbg && dbg->h_gcn[layer]) std::copy(pl->h_feat.p, pl->h_feat.p + (size_t)C * Dcat, dbg->h_gcn[layer]);
            tap_dist(1 + layer, E);

            // ---- grouping on the GCN features (model.py:802-815 / 843-856) ----
            Lcur = Lnew;
            PL_CHECK(regroup(2.0f));
            lap(1);
            out->trace[2 + layer] = Lnew.C;
            PL_CHECK(tables_for(6 + 3 * layer, true));    // layer_3.* / layer_4.*
            lap(2);
            tap_adj(2 + layer, adj, E);
            // next layer: previous features = this GCN output (featB); its concat goes to featA again and its
            // GCN output back into featB -- safe, the stream runs group_max_rows(featB -> featA) before gcn writes featB
            feat_prev = gcn_out; feat_prev_stride = Dcat; feat_prev_dim = Dcat;
        }

        // ---------------- final clustering (model.py:868-888) ------------------------------------------
        // Feat_4 = max over absorbed rows of the gcn_3 output (host copy), adj_4 = current adj
        const int D4 = 256;
        std::vector<float> feat4((size_t)Lnew.C * D4, -INFINITY);
        for (int j = 0; j < Lcur.C; ++j) {
            float* dstp = &feat4[(size_t)Lnew.cl_of_seg[Lcur.root[j]] * D4];
            const float* src = pl->h_feat.p + (size_t)j * D4;
            for (int k = 0; k < D4; ++k) dstp[k] = std::max(dstp[k], src[k]);
        }
        std::vector<int32_t> root5(Lnew.root.begin(), Lnew.root.begin() + Lnew.C);
        root5.resize(S);
        int C5 = Lnew.C, E5 = E;
        adj.resize(2 * (size_t)std::max(E, 1));
        const int need_fallback = sg_partition_group_unlabeled(part, root5.data(), &C5, feat4.data(), D4, adj.data(), &E5);
        if (need_fallback < 0) { sg_partition_destroy(part); return need_fallback; }
        if (need_fallback) {
            // FPS-1024 over the current clusters (model.py:479), XYZ only, no transform
            LayerDesc L5;
            freeze_layer(part, S, L5);
            std::vector<int32_t> cl_of_order(S);
            for (int i = 0; i < S; ++i) cl_of_order[i] = L5.cl_of_seg[L5.order[i]];
            size_t cur = 0;
            auto put = [&](const std::vector<int32_t>& v, size_t count) {
                const size_t at = cur;
                std::copy(v.begin(), v.begin() + count, pl->h_desc.p + at);
                cur += (count + 3) / 4 * 4;
                return at;
            };
            const size_t o_order = put(L5.order, S), o_dst = put(L5.dst, S), o_cl = put(cl_of_order, S), o_off = put(L5.cl_pt_off, L5.C + 1);
            pl->mark(-1);
            PL_COPY(pl->desc.p, pl->h_desc.p, cur * 4, hipMemcpyHostToDevice, st);
            const int32_t* dd = pl->desc.p;
            PL_CHECK(sg_gather_members(sc->d_seg_points, sc->d_seg_off, S, dd + o_order, dd + o_dst, dd + o_cl, pl->members.p, nullptr, nullptr, nullptr, stv));
            int max_cl = 0;
            for (int c = 0; c < L5.C; ++c) max_cl = std::max(max_cl, L5.cl_pt_off[c + 1] - L5.cl_pt_off[c]);
            PL_CHECK(pl->need_fallback_buffers());
            PL_CHECK(sg::fps_sample_hint(sc->d_data, N, 6, pl->members.p, dd + o_off, L5.C, 1024, 3, 0, pl->samples_big.p, nullptr,
                                         pl->ws_fps.p, pl->ws_fps.n, stv, max_cl));
            PL_HIP(hipMemcpyAsync(pl->h_samples.p, pl->samples_big.p, (size_t)L5.C * 1024 * 3 * 4, hipMemcpyDeviceToHost, st));
            pl->mark(16);
            lap(5);
            PL_HIP(timed_sync(st));
            lap(-1);
            PL_CHECK(sg_partition_unlabeled_fallback(part, L5.root.data(), L5.C, pl->h_samples.p, 1024));
            out->used_fallback = 1;
        }
        out->trace[4] = sg_partition_num_clusters(part);
        if (tape || (dbg && dbg->h_feat5 && dbg->h_ins5 && dbg->h_sem5)) {
            // Feat_5 + the weak labels of the final clusters: what the train-mode tail consumes (model.py:900-914).  After the
            // FPS-1024 fallback the reference max-aggregates once more into the final numbering (model.py:495-507).
            LayerDesc L6;
            const int C6 = freeze_layer(part, S, L6);
            std::vector<float> f6((size_t)C6 * D4, -INFINITY);
            for (int j = 0; j < C5; ++j) {
                float* dstp = &f6[(size_t)L6.cl_of_seg[root5[j]] * D4];
                const float* src = &feat4[(size_t)j * D4];
                for (int k = 0; k < D4; ++k) dstp[k] = std::max(dstp[k], src[k]);
            }
            std::vector<int32_t> ins6(C6), sem6(C6);
            for (int c = 0; c < C6; ++c) {
                double np_ = 0.0;
                PL_CHECK(sg_partition_label(part, L6.root[c], &ins6[c], &sem6[c], &np_));
            }
            if (dbg->h_feat5 && dbg->h_ins5 && dbg->h_sem5) {
                std::copy(f6.begin(), f6.end(), dbg->h_feat5);
                std::copy(ins6.begin(), ins6.end(), dbg->h_ins5);
                std::copy(sem6.begin(), sem6.end(), dbg->h_sem5);
                dbg->n5 = C6;
            }
            if (tape) {
                // rows of the last GCN output (Lcur numbering) -> final cluster: the composition of the max-aggregations above
                tape->C6 = C6; tape->N = N; tape->S = S;
                tape->fin_goff.assign(C6 + 1, 0);
                tape->fin_gidx.resize(Lcur.C);
                for (int j = 0; j < Lcur.C; ++j) ++tape->fin_goff[L6.cl_of_seg[Lcur.root[j]] + 1];
                for (int c = 0; c < C6; ++c) tape->fin_goff[c + 1] += tape->fin_goff[c];
                std::vector<int32_t> fill(tape->fin_goff.begin(), tape->fin_goff.end() - 1);
                for (int j = 0; j < Lcur.C; ++j) tape->fin_gidx[fill[L6.cl_of_seg[Lcur.root[j]]]++] = j;
                tape->feat5.swap(f6);
                tape->ins5.swap(ins6); tape->sem5.swap(sem6);
                tape->filled = true;
            }
        }
        PL_CHECK(tables_for(12, false));                  // final.{ins,sem}
        n_tables = 14; ins_row = 12; sem_row = 13;
    }

    // ---------------- export + evaluate (model.py:525-655) -------------------------------------------
    pl->mark(-1);
    (void)n_tables;
    PL_CHECK(flush_exports(true));                         // whatever is left (at least the final rows)
    PL_HIP(hipStreamWaitEvent(st, pl->ev_side, 0));        // the metric kernels read the LAST exported rows on the device
    pl->mark(17);
    PL_CHECK(sg::evaluate_landing(sc->d_gt, pl->labels.p + (size_t)sem_row * V, pl->labels.p + (size_t)ins_row * V, V, max_ins, out->iou_sem,
                                  out->iou_ins, out->acc, pl->ws_eval.p, pl->ws_eval.n, stv, reinterpret_cast<uint32_t*>(pl->h_eval.p)));
    pl->mark(18);
    lap(6);
    PL_HIP(timed_sync(st));
    PL_HIP(timed_sync(side));                             // the last rows' D2H
    lap(-1);
    sg_partition_destroy(part);
    part = nullptr;

    for (float& m : pl->stage_ms) m = 0.f;
    for (int i = 1; i < pl->n_ev; ++i) {
        float ms = 0.f;
        if (pl->ev_stage[i] >= 0 && hipEventElapsedTime(&ms, pl->ev[i - 1], pl->ev[i]) == hipSuccess) pl->stage_ms[pl->ev_stage[i]] += ms;
    }
    pl->stage_ms[7] = pl->stage_ms[19] + pl->stage_ms[20];
    pl->stage_ms[13] = pl->stage_ms[21] + pl->stage_ms[22] + pl->stage_ms[23];
    return SG_OK;
}

// Batch driver (infer.py:149-152 loop body for many scenes): `npipes` host threads, one per pipeline, pull scene
// indices from an atomic counter until all `count` scenes are done.  Everything between two scenes' kernels --
// the grouping engine, descriptor building, result hand-over -- runs in these native threads, so one scene's
// host phase overlaps the others' kernels without the Python interpreter in the loop.
static const char* kLabelNames[SG_NUM_LABEL_VECTORS] = {"layer_1.seg", "layer_1.ins", "layer_1.sem", "layer_2.seg", "layer_2.ins",
                                                        "layer_2.sem", "layer_3.seg", "layer_3.ins", "layer_3.sem", "layer_4.seg",
                                                        "layer_4.ins", "layer_4.sem", "final.ins", "final.sem"};

int sg_batch_forward(sg_pipeline* const* pipes, int npipes, const sg_scene* scenes, int count, int mode, sg_result* results,
                     float* h_stage_ms_sum, sg_writer* writer, const char* const* out_dirs, int formats) {
    if (!pipes || npipes <= 0 || count < 0 || (count > 0 && (!scenes || !results))) return sg::fail(SG_EINVAL, "sg_batch_forward: bad arguments");
    std::atomic<int> next(0);
    std::atomic<int> first_err(0);
    std::mutex mu;
    std::string msg;
    std::vector<float> sums(kNumStages, 0.f);
    auto worker = [&](int w) {
        std::vector<float> mine(kNumStages, 0.f);
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= count || first_err.load() != 0) break;
            const int rc = sg_pipeline_forward(pipes[w], &scenes[i], mode, &results[i], nullptr);
            if (rc < 0) {
                int zero = 0;
                if (first_err.compare_exchange_strong(zero, rc)) { std::lock_guard<std::mutex> g(mu); msg = sg_last_error(); }
                break;
            }
            for (int k = 0; k < kNumStages; ++k) mine[k] += pipes[w]->stage_ms[k];
            if (writer && out_dirs && out_dirs[i]) {          // hand the label vectors to the writer pool (it copies them)
                const int nvec = mode == SG_MODE_INS_INFER ? SG_NUM_LABEL_VECTORS : 6;
                for (int v = 0; v < nvec && rc >= 0; ++v) {
                    const std::string base = std::string(out_dirs[i]) + "/" + kLabelNames[v];
                    const int wrc = sg_writer_submit(writer, base.c_str(), results[i].h_labels + (size_t)v * scenes[i].V, scenes[i].V, formats);
                    if (wrc < 0) {
                        int zero = 0;
                        if (first_err.compare_exchange_strong(zero, wrc)) { std::lock_guard<std::mutex> g(mu); msg = sg_last_error(); }
                        break;
                    }
                }
            }
        }
        std::lock_guard<std::mutex> g(mu);
        for (int k = 0; k < kNumStages; ++k) sums[k] += mine[k];
    };
    std::vector<std::thread> th;
    const int nt = std::min(npipes, std::max(count, 1));
    for (int w = 1; w < nt; ++w) th.emplace_back(worker, w);
    worker(0);
    for (auto& t : th) t.join();
    if (g_host_profile && g_prof_scenes.load() > 0) {
        const double n = (double)g_prof_scenes.exchange(0);
        const double tot = g_prof_total_ns.exchange(0) / n * 1e-6, syn = g_prof_sync_ns.exchange(0) / n * 1e-6;
        fprintf(stderr, "[sg host profile] %d scenes on %d pipelines: forward %.3f ms per scene = %.3f ms blocked in stream syncs + %.3f ms host work / launches:",
                (int)n, npipes, tot, syn, tot - syn);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.3f", kProfSec[i], g_prof_sec[i].exchange(0) / n * 1e-6);
        fprintf(stderr, "\n");
    }
    if (h_stage_ms_sum)
        for (int k = 0; k < kNumStages; ++k) h_stage_ms_sum[k] += sums[k];
    if (first_err.load() != 0) return sg::fail(first_err.load(), "sg_batch_forward: %s", msg.c_str());
    return SG_OK;
}

}  // extern "C"
