// Scene engine: SegModel.forward (reference seggroup/model.py:684-897) for MANY scenes, with the scene index as a grid
// dimension.
//
// sg_pipeline_forward runs one scene per host thread and stream: ~45 launches, ~10 small copies and 4-5 stream
// synchronisations per scene, and 16 such threads hammering one HIP runtime is what bounded round 1 (trivial kernels
// waited 100+ us for their turn).  Here a GROUP of up to B scenes advances in lock-step through the same four phases
//
//     P0   contract mesh edges | FPS-64 | Morton sort + boxes | MLP1 | edge distances          (device-side edge count)
//     L2   layout | group max | in-cluster kNN | seed table | EdgeConv MLP2 | segment max | GCN | edge distances
//     L3   the same with MLP3 (edge-feature moments first) and the seeded kNN
//     END  export of the 14 label vectors | metric counters
//
// and every kernel is launched ONCE per phase for all B scenes (grid.y = scene, arguments from a SlotCtx array: see
// engine_ctx.h).  Per phase a group issues ONE host-to-device copy (SlotCtx array + every scene's descriptors), the
// launches, ONE device-to-host copy (the "outbox": distances, contracted adjacency, segment sums, counters) and ONE
// stream synchronisation; the serial grouping of the B scenes (grouping.cpp) runs between phases on the group's host
// thread while the other groups' kernels occupy the GPU.  ~35 launches per B scenes instead of ~45 per scene.
//
// The engine owns G groups (one persistent host thread + one HIP stream each) that pull scenes from a job queue:
// sg_engine_submit() returns at once, so a driver can queue the next batch before the previous one has drained.
// Kernel bodies are shared with the single-scene path, so results are bit-identical to sg_pipeline_forward (tested).
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <memory>
#include <map>
#include <mutex>
#include <string>
#include <thread>

#include "engine_ctx.h"
#include "pipeline_priv.h"

namespace {

using namespace sgp;
using sg::BatchDims;
using sg::SlotCtx;

constexpr int kMaxGroupEvents = 24;
#define EG_CHECK_RET(call) do { int rc__ = (call); if (rc__ < 0) return rc__; } while (0)

// SG_ENGINE_PROFILE=1: wall time of the group threads split into host work (grouping, descriptors, launch calls), time
// blocked in hipStreamSynchronize and time idle waiting for work; printed by sg_engine_destroy -- a development aid
const bool g_profile_print = getenv("SG_ENGINE_PROFILE") != nullptr;     // print at sg_engine_destroy
bool g_profile = g_profile_print;
thread_local long long tl_ns_sync = 0;
inline long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline hipError_t timed_sync(hipStream_t st) {
    if (!g_profile) return hipStreamSynchronize(st);
    const long long t0 = now_ns();
    const hipError_t e = hipStreamSynchronize(st);
    tl_ns_sync += now_ns() - t0;
    return e;
}

// consecutive roctx ranges of a phase (sg_common.h: no-ops unless SG_ROCTX): next() closes the open one; the destructor closes the last, also
// on an early return
struct Stages {
    bool open = false;
    void next(const char* name) { if (open) sg::roctx_pop(); sg::roctx_push(name); open = true; }
    ~Stages() { if (open) sg::roctx_pop(); }
};

// bump allocator over a pinned buffer whose device twin has the same layout
struct Arena {
    char* h = nullptr;
    char* d = nullptr;
    size_t cap = 0, used = 0;
    bool ok = true;
    void reset() { used = 0; ok = true; }
    template <class T>
    T* take(size_t count, T** dev) {
        const size_t at = (used + 63) / 64 * 64, bytes = count * sizeof(T);
        if (at + bytes > cap) { ok = false; *dev = nullptr; return nullptr; }
        used = at + bytes;
        *dev = reinterpret_cast<T*>(d + at);
        return reinterpret_cast<T*>(h + at);
    }
};

struct Job {
    int id = 0;
    const sg_scene* scenes = nullptr;
    sg_result* results = nullptr;
    int count = 0, mode = 0;
    sg_writer* writer = nullptr;
    const char* const* out_dirs = nullptr;
    int formats = 0;
    int next = 0, done = 0;
    int err = 0;
    std::string msg;
};

// one scene in flight in one slot of a group
struct Run {
    sg_pipeline* pl = nullptr;
    const sg_scene* sc = nullptr;
    sg_result* out = nullptr;
    const char* out_dir = nullptr;
    sg_partition* part = nullptr;
    int max_ins = 1, max_seg = 0, cap1 = 0, out_rows = 0;
    std::vector<int32_t> lay_big;
    int E = 0;
    std::vector<int32_t> adj, adj_next;
    std::vector<uint8_t> connected, keep;
    LayerDesc Lcur, Lnew;
    const float* feat_prev = nullptr;
    int feat_prev_stride = 128, feat_prev_dim = 128;
    int n_tables = 6, ins_row = 4, sem_row = 5;
    std::vector<int32_t> tab;                  // [14,S] export tables
    SlotCtx ctx;                               // master copy; written into the params arena before every phase
    // host views of this phase's outbox
    int32_t* o_count = nullptr;
    double* o_seg_sums = nullptr;
    int32_t* o_adj1 = nullptr;
    float* o_dist = nullptr;
    float* o_feat = nullptr;
    uint32_t* o_cnt = nullptr;
    bool dist_in_outbox = true;
    std::vector<double> seg_sums;              // [S,3] kept for the cluster centroids of both layers
    std::vector<int32_t> chunk_off;            // [S+1] first 32-point chunk of every segment (re-shipped with every phase's parameters)
    // fixed carve of the slot's workspaces
    uint32_t* bitmap = nullptr;
    int* block_count = nullptr;
    uint8_t* m1_knn = nullptr;
    double* m1_partial = nullptr;
    float* m1_folded = nullptr;
    double* ec_partial = nullptr;
    float* ec_fold = nullptr;
    float* g_dist = nullptr;
    float* g_agg = nullptr;
};

}  // namespace

struct sg_engine {
    int G = 0, B = 0, device = 0;
    int maxN = 0, maxS = 0, maxE = 0, maxV = 0;
    int timing = 0;
    int label_compact = 0;                      // sg_engine_set_label_transfer
    size_t dev_bytes = 0;

    struct Group {
        sg_engine* eng = nullptr;
        int index = 0;
        hipStream_t stream = nullptr;
        hipStream_t side = nullptr;                 // phase P0's big-segment chain (sort + FPS of segments beyond 2,048 points): only with SG_ENGINE_FORK
        hipStream_t heavy = nullptr;                // SG_ENGINE_CUMASK=heavy:<n>[:knn]: the EdgeConv (and kNN) launches on a stream confined to n CUs
        bool heavy_knn = false;
        hipEvent_t ev_fork = nullptr, ev_join = nullptr;
        std::vector<sg_pipeline*> slots;
        Arena par, box;
        hipEvent_t ev[kMaxGroupEvents];
        int ev_stage[kMaxGroupEvents];
        int n_ev = 0;
        std::vector<Run> runs;
        std::thread th;

        void mark(int stage) {                                // stage < 0: a start mark
            if (eng->timing == 0 || n_ev >= kMaxGroupEvents) return;
            (void)hipEventRecord(ev[n_ev], stream);
            ev_stage[n_ev++] = stage;
        }
        int superstep(Run* r, int n, int mode, sg_writer* writer, int formats, long long tag);
        // A phase's parameter block (H2D) and outbox (D2H), a few hundred KB each.  Round 5: moved by a KERNEL on the group's stream
        // (pinned memory is mapped into the device's address space; the arenas are 64-byte aligned and sized in 64-byte steps) instead of
        // hipMemcpyAsync: on the copy engines they queued behind whatever bulk transfers were in flight -- with the pack loader uploading
        // 13 MB scenes beside the engine, every phase of every group waited for a few of those (tools/exp_h2d_interference.py).
        // SG_ENGINE_COPY=sdma restores the copy-engine path.
        // SG_ENGINE_COPY=hsa | hsa_par | hsa_box (experiment, round 6): the parameter block and / or the outbox through the HSA runtime's copy interface
        // (sdma.cpp) -- the parameter block issued and waited for in front of the phase's launches, the outbox fetched BEHIND the phase's stream sync
        static int hsa_mode() {
            static const int m = [] { const char* e = getenv("SG_ENGINE_COPY"); const std::string v = e ? e : ""; return v == "hsa" ? 3 : v == "hsa_par" ? 1 : v == "hsa_box" ? 2 : 0; }();
            return m;
        }
        int outbox_and_sync() {                                   // the phase's results on the host, the stream drained
            if ((hsa_mode() & 2) && sg::sdma_available()) {
                if (timed_sync(stream) != hipSuccess) return sg::fail(SG_EHIP, "engine: stream sync failed");
                sg::SdmaTicket t;
                const long long t0 = g_profile ? now_ns() : 0;
                if (box.used && (sg::sdma_issue(box.h, box.d, (box.used + 15) / 16 * 16, &t) != SG_OK || sg::sdma_wait(&t) != SG_OK)) {
                    sg::err_buf()[0] = 0;
                    EG_CHECK_RET(sg::copy_by_kernel(box.h, box.d, (box.used + 15) / 16 * 16, stream));
                    if (timed_sync(stream) != hipSuccess) return sg::fail(SG_EHIP, "engine: stream sync failed");
                }
                if (g_profile) tl_ns_sync += now_ns() - t0;
                return SG_OK;
            }
            const int rc = arena_copy(box.h, box.d, box.used, false);
            if (rc < 0) return rc;
            return timed_sync(stream) == hipSuccess ? SG_OK : sg::fail(SG_EHIP, "engine: stream sync failed");
        }
        int arena_copy(void* dst, const void* src, size_t bytes, bool to_device) {
            static const bool sdma = getenv("SG_ENGINE_COPY") && std::string(getenv("SG_ENGINE_COPY")) == "sdma";
            if (bytes == 0) return SG_OK;
            if (to_device && (hsa_mode() & 1) && sg::sdma_available()) {
                sg::SdmaTicket t;
                if (sg::sdma_issue(dst, src, (bytes + 15) / 16 * 16, &t) == SG_OK && sg::sdma_wait(&t) == SG_OK) return SG_OK;
                sg::err_buf()[0] = 0;
            }
            if (sdma) {
                if (hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, stream) != hipSuccess) return sg::fail(SG_EHIP, "engine: arena copy failed");
                return SG_OK;
            }
            return sg::copy_by_kernel(dst, src, (bytes + 15) / 16 * 16, stream);      // (arenas: 64-byte aligned blocks inside allocations with slack)
        }
        int phase_p0(Run* r, int n, int mode);
        int phase_layer(Run* r, int n, int layer);
        int phase_end(Run* r, int n, int mode);
        void collect_times(int n);
        void loop();
    };
    std::vector<std::unique_ptr<Group>> groups;

    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<std::shared_ptr<Job>> jobs;
    int next_id = 0;
    int idle_groups = 0;                  // groups waiting for work (under mu)
    bool stop = false;

    std::atomic<long long> ns_sync{0}, ns_step{0}, ns_idle{0}, n_steps{0}, n_step_scenes{0};     // SG_ENGINE_PROFILE

    std::mutex mu_times;
    double stage_ms_sum[kNumStages] = {0};
    long long scenes_timed = 0;
};

namespace {

#define EG_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) \
    return sg::fail(SG_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); } while (0)
#define EG_CHECK(call) do { int rc__ = (call); if (rc__ < 0) return rc__; } while (0)

// SG_ENGINE_HASH=1 in a DEBUG build (make DEBUG=1: -DSG_ENGINE_DEBUG; ADVICE round 5 -- the release library carries none of this): after every phase,
// one line per scene on stderr with FNV-1a digests of the phase's device results -- two runs of the same scenes must print the same lines, whatever the
// group shape (tools/r05_repro.py compares them).  The first sample set / kNN table seen for a scene is kept and later ones are compared with it; the
// store is keyed by a digest of the scene's INPUT (device slots are reused for other scenes by the pack loader: a pointer is not a scene) and is emptied
// when an engine is destroyed.
#ifdef SG_ENGINE_DEBUG
static const bool g_hash = getenv("SG_ENGINE_HASH") != nullptr;
thread_local hipStream_t tl_hash_stream = nullptr;          // the calling group's stream: the copies must not touch the null stream (it would serialise the groups)
int fetch(void* h, const void* d, size_t bytes) {
    if (bytes == 0) return 0;
    if (hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, tl_hash_stream) != hipSuccess) return -1;
    return hipStreamSynchronize(tl_hash_stream) == hipSuccess ? 0 : -1;
}
uint64_t dev_digest(const void* d, size_t bytes) {
    std::vector<unsigned char> h(bytes);
    if (fetch(h.data(), d, bytes) != 0) return 0;
    uint64_t x = 1469598103934665603ull;
    for (size_t i = 0; i < bytes; ++i) { x ^= h[i]; x *= 1099511628211ull; }
    return x;
}
std::mutex g_dbg_mu;
std::map<uint64_t, std::vector<float>> g_dbg_samples;                           // scene digest -> first sample set seen
std::map<std::pair<uint64_t, int>, std::vector<int32_t>> g_dbg_knn;               // (scene digest, layer) -> first kNN table seen
uint64_t scene_key(const sg_scene* sc) { return dev_digest(sc->d_data, (size_t)sc->N * 24) ^ ((uint64_t)sc->N << 40) ^ (uint64_t)sc->S; }
void debug_reset() { std::lock_guard<std::mutex> g(g_dbg_mu); g_dbg_samples.clear(); g_dbg_knn.clear(); }
#else
inline void debug_reset() {}
#endif

int tables_for(Run& r, int first_row, bool with_seg) {
    const int S = r.sc->S;
    int32_t* a = r.tab.data() + (size_t)first_row * S;
    return sg_partition_export_tables(r.part, with_seg ? a : nullptr, with_seg ? a + S : a, with_seg ? a + 2 * (size_t)S : a + S);
}

// group + re-index + contract (model.py:218-258, 291-302, 759-768): new layer in Lnew, contracted adjacency in adj
int regroup(Run& r, const float* h_dist, float th) {
    r.connected.assign(std::max(r.E, 1), 0);
    int rc = sg_partition_group_nearby(r.part, r.Lcur.root.data(), r.Lcur.C, h_dist, r.adj.data(), r.E, th, r.connected.data());
    if (rc == SG_ESTALL) { r.out->stalled = 1; rc = SG_OK; sg::err_buf()[0] = 0; }
    if (rc < 0) return rc;
    r.keep.resize(r.connected.size());
    for (size_t i = 0; i < r.connected.size(); ++i) r.keep[i] = !r.connected[i];
    r.adj_next.resize(2 * (size_t)std::max(r.E, 1));
    const int En = sg_partition_contract(r.part, r.Lcur.root.data(), r.adj.data(), r.E, r.keep.data(), r.adj_next.data());
    if (En < 0) return En;
    freeze_layer(r.part, r.sc->S, r.Lnew);
    r.adj.assign(r.adj_next.begin(), r.adj_next.begin() + 2 * (size_t)En);
    r.E = En;
    return SG_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// P0: graph initialisation + structural layer up to its decision distances (model.py:710-757)
// ---------------------------------------------------------------------------------------------------------------------
int sg_engine::Group::phase_p0(Run* runs_, int n, int mode) {
    Stages rg;
    rg.next("P0.describe");
    par.reset(); box.reset();
    SlotCtx* d_ctx = nullptr;
    SlotCtx* h_ctx = par.take<SlotCtx>(n, &d_ctx);
    BatchDims bd;
    bd.nslots = n;
    for (int i = 0; i < n; ++i) {
        Run& r = runs_[i];
        sg_pipeline* pl = r.pl;
        const sg_scene* sc = r.sc;
        const int N = sc->N, S = sc->S;
        r.out->stalled = 0; r.out->used_fallback = 0;
        for (int k = 0; k < 5; ++k) r.out->trace[k] = 0;
        r.part = sg_partition_create(S, sc->h_seg_first, sc->h_seg_size, sc->h_seg_ins, sc->h_seg_sem);
        if (!r.part) return SG_EINVAL;
        r.max_ins = 1; r.max_seg = 0;
        for (int s = 0; s < S; ++s) { r.max_ins = std::max(r.max_ins, sc->h_seg_ins[s] + 2); r.max_seg = std::max(r.max_seg, sc->h_seg_size[s]); }
        if (r.max_ins > pl->maxS + 2)
            return sg::fail(SG_EUNSUP, "weak instance ids up to %d exceed the engine's metric workspace (max_segments + 2)", r.max_ins - 2);
        r.tab.resize((size_t)SG_NUM_LABEL_VECTORS * S);
        r.cap1 = (int)(pl->adj1.n / 2);
        r.out_rows = std::min(r.cap1, 8 * S + 64);
        r.n_tables = 6; r.ins_row = 4; r.sem_row = 5;

        SlotCtx& c = r.ctx;
        std::memset(&c, 0, sizeof c);
        c.data = sc->d_data; c.adj0 = sc->d_adj; c.seg_of_point = sc->d_seg_of_point; c.seg_points = sc->d_seg_points; c.seg_off = sc->d_seg_off;
        c.unmap = sc->d_unmap; c.gt = sc->d_gt;
        c.N = N; c.S = S; c.E0 = sc->E0; c.V = sc->V;
        const size_t bits = (size_t)S * (size_t)S;
        if (bits > (1ull << 31)) return sg::fail(SG_EUNSUP, "S=%d exceeds the bitmap envelope (S*S <= 2^31)", S);
        c.bitmap = r.bitmap; c.bitmap_words = (bits + 31) / 32; c.block_count = r.block_count;
        c.bits_blocks = (int)((c.bitmap_words + 1023) / 1024);
        c.adj1 = pl->adj1.p; c.cap1 = r.cap1; c.out_rows = r.out_rows;
        // outbox of this phase: count | segment sums | first out_rows adjacency rows | their distances
        r.o_count = box.take<int32_t>(4, &c.count);
        r.o_seg_sums = box.take<double>((size_t)S * 3, &c.seg_sums);
        r.o_adj1 = box.take<int32_t>(2 * (size_t)r.out_rows, &c.adj1_out);
        r.o_dist = box.take<float>((size_t)r.out_rows, &c.dist1_out);
        c.samples = pl->samples.p; c.ws_fps = reinterpret_cast<float*>(pl->ws_fps.p);
        {
            // the segments of the larger size classes (kernels_fps.hip: 513-2,048 points on four waves, beyond on sixteen; kernels_knn_sorted.hip:
            // beyond 2,048 points the cell-bucketed sort): their launches index these lists instead of testing every segment
            int n_mid = 0, n_big = 0;
            for (int s = 0; s < S; ++s) { const int z = sc->h_seg_size[s]; n_mid += z > sg::kSegSmallMax && z <= sg::kSegMidMax; n_big += z > sg::kSegMidMax; }
            int32_t *d_mid = nullptr, *d_big = nullptr;
            int32_t* mid = par.take<int32_t>((size_t)std::max(n_mid, 1), &d_mid);
            int32_t* big = par.take<int32_t>((size_t)std::max(n_big, 1), &d_big);
            if (mid && big) {
                int a = 0, b = 0;
                for (int s = 0; s < S; ++s) { const int z = sc->h_seg_size[s]; if (z > sg::kSegMidMax) big[b++] = s; else if (z > sg::kSegSmallMax) mid[a++] = s; }
            }
            c.mid_segs = d_mid; c.big_segs = d_big; c.n_mid = n_mid; c.n_big = n_big;
            bd.max_mid = std::max(bd.max_mid, n_mid); bd.max_big = std::max(bd.max_big, n_big);
        }
        if (mode == SG_MODE_INS_INFER) {
            r.chunk_off.resize((size_t)S + 1);
            r.chunk_off[0] = 0;
            for (int s = 0; s < S; ++s) r.chunk_off[s + 1] = r.chunk_off[s] + (sc->h_seg_size[s] + 31) / 32;
            int32_t* d_co = nullptr;
            int32_t* co = par.take<int32_t>((size_t)S + 1, &d_co);
            if (co) std::memcpy(co, r.chunk_off.data(), ((size_t)S + 1) * 4);
            c.seg_chunk_off = d_co;                               // lives in THIS phase's parameter block only
            c.segbox = pl->segbox.p; c.sperm = pl->sperm.p; c.chunk_box = pl->chunk_box.p;
            c.sort_keys = reinterpret_cast<unsigned long long*>(pl->ws_sort.p);
        }
        c.m1_knn = r.m1_knn; c.m1_partial = r.m1_partial; c.m1_folded = r.m1_folded; c.feat1 = pl->feat1.p;
        c.dist_feat = pl->feat1.p; c.dist_stride = 128; c.dist_D = 128; c.dist_adj = pl->adj1.p; c.dist_E_dev = c.count; c.dist_E = 0;
        c.dist = pl->dist.p; c.dist_copy = c.dist1_out; c.dist_copy_rows = r.out_rows;
        c.labels = pl->labels.p;
        bd.max_N = std::max(bd.max_N, N); bd.max_S = std::max(bd.max_S, S); bd.max_E0 = std::max(bd.max_E0, sc->E0); bd.max_V = std::max(bd.max_V, sc->V);
        bd.max_bits_blocks = std::max(bd.max_bits_blocks, c.bits_blocks); bd.max_seg = std::max(bd.max_seg, r.max_seg);
        bd.max_E = std::max(bd.max_E, r.out_rows);
    }
    if (!par.ok || !box.ok) return sg::fail(SG_ENOMEM, "engine: phase P0 exceeds the group's parameter / outbox arena");
    for (int i = 0; i < n; ++i) h_ctx[i] = runs_[i].ctx;
    EG_CHECK(arena_copy(par.d, par.h, par.used, true));
    mark(-1);
    rg.next("P0.contract");
    EG_CHECK(sg::b_contract(d_ctx, bd, stream));
    mark(0);
    rg.next("P0.sort_boxes+fps64");
    // Segments beyond 2,048 points (floors and walls of a scan) take a chain of skinny launches -- bucket / runs / boxes of the Morton sort, then a
    // chunk-pruned FPS that one workgroup per segment walks for hundreds of microseconds -- that has nothing to do with the thousand small segments'
    // sort and sampling: the two chains run side by side, the big one on the group's SIDE stream (fork behind the parameter block and the
    // contraction's launches, join in front of MLP1, which reads every segment's samples).  Solo, per launch of 8 ScanNet-shaped scenes: ~680 us of
    // critical path instead of ~860.  Batches without such a segment launch nothing on the side stream.
    // MEASURED AND NOT THE DEFAULT (round 5, bench --seg-profile scannet, three regions each): side by side 2,977-2,990 scenes/s, one after
    // the other 3,009-3,054 -- under the load of ten groups the chain's latency is covered anyway and a second stream per group only adds
    // launches that wait for each other.  SG_ENGINE_FORK=1 turns it on (single-group latency experiments).
    static const bool want_fork = getenv("SG_ENGINE_FORK") != nullptr;
    const bool fork = want_fork && side && sg::fps_has_big_class(bd);
    if (fork) {
        EG_HIP(hipEventRecord(ev_fork, stream));
        EG_HIP(hipStreamWaitEvent(side, ev_fork, 0));
        if (mode == SG_MODE_INS_INFER) EG_CHECK(sg::b_sort_boxes(d_ctx, bd, side, 2));
        EG_CHECK(sg::b_fps64(d_ctx, bd, side, mode == SG_MODE_INS_INFER, 2));
        EG_HIP(hipEventRecord(ev_join, side));
        if (mode == SG_MODE_INS_INFER) EG_CHECK(sg::b_sort_boxes(d_ctx, bd, stream, 1));
        EG_CHECK(sg::b_fps64(d_ctx, bd, stream, mode == SG_MODE_INS_INFER, 1));
        EG_HIP(hipStreamWaitEvent(stream, ev_join, 0));
    } else {
        if (mode == SG_MODE_INS_INFER) {
            EG_CHECK(sg::b_sort_boxes(d_ctx, bd, stream));          // first: the sampling of the largest segments walks its chunk boxes
        }
        EG_CHECK(sg::b_fps64(d_ctx, bd, stream, mode == SG_MODE_INS_INFER));
    }
    mark(1);
    rg.next("P0.mlp1");
    sg_pipeline* p0 = runs_[0].pl;
    EG_CHECK(sg::b_mlp1(d_ctx, p0->w.p + p0->o_m1w, p0->w.p + p0->o_m1g, p0->w.p + p0->o_m1b, bd, stream));
    mark(2);
    rg.next("P0.edge_distance");
    EG_CHECK(sg::b_edge_distance(d_ctx, bd, stream));
    mark(3);
    rg.next("P0.sync");
    EG_CHECK(outbox_and_sync());
    rg.next("P0.host_grouping");

    // ---- host: layer 1 tables, structural grouping, layer 2 tables ----
    for (int i = 0; i < n; ++i) {
        Run& r = runs_[i];
        const int S = r.sc->S;
        const int E1 = r.o_count[0];
#ifdef SG_ENGINE_DEBUG
        if (g_hash) tl_hash_stream = stream;
        if (g_hash) {
            // the first sample set seen for a scene is kept; later ones are compared segment by segment
            const uint64_t skey = scene_key(r.sc);
            std::vector<float> smp((size_t)S * 64 * 6);
            (void)fetch(smp.data(), r.pl->samples.p, smp.size() * 4);
            {
                // the same sampling once more through the single-scene entry (no Morton order: its large segments take the plain strided passes)
                float* d_ref = nullptr; void* d_wsr = nullptr;
                if (hipMalloc((void**)&d_ref, smp.size() * 4) == hipSuccess && hipMalloc(&d_wsr, (size_t)r.sc->N * 16) == hipSuccess &&
                    sg::fps_sample_hint(r.sc->d_data, r.sc->N, 6, r.sc->d_seg_points, r.sc->d_seg_off, S, 64, 6, 1, d_ref, nullptr, d_wsr, (size_t)r.sc->N * 16,
                                        (void*)stream, r.max_seg) == SG_OK) {
                    std::vector<float> ref(smp.size());
                    (void)fetch(ref.data(), d_ref, ref.size() * 4);
                    for (int sg_ = 0; sg_ < S; ++sg_)
                        if (std::memcmp(&smp[(size_t)sg_ * 384], &ref[(size_t)sg_ * 384], 384 * 4) != 0)
                            std::fprintf(stderr, "SGROW %p P0 segment %d (%d points): the engine's samples differ from the single-scene entry's\n", (const void*)r.sc->d_data, sg_, r.sc->h_seg_size[sg_]);
                }
                (void)hipStreamSynchronize(stream);
                if (d_ref) (void)hipFree(d_ref);
                if (d_wsr) (void)hipFree(d_wsr);
            }
            std::lock_guard<std::mutex> g(g_dbg_mu);
            auto it = g_dbg_samples.find(skey);
            if (it == g_dbg_samples.end()) g_dbg_samples[skey] = smp;
            else {
                int shown = 0;
                for (int sg_ = 0; sg_ < S; ++sg_) {
                    const float* a = &smp[(size_t)sg_ * 384];
                    const float* b = &it->second[(size_t)sg_ * 384];
                    if (std::memcmp(a, b, 384 * 4) == 0) continue;
                    int rows = 0, first_row = -1;
                    for (int k = 0; k < 64; ++k) if (std::memcmp(a + 6 * k, b + 6 * k, 24) != 0) { ++rows; if (first_row < 0) first_row = k; }
                    if (shown++ < 6) std::fprintf(stderr, "SGROW %p P0 segment %d (%d points): %d of its 64 sample rows differ, the first is row %d\n", (const void*)r.sc->d_data, sg_, r.sc->h_seg_size[sg_], rows, first_row);
                }
            }
        }
        if (g_hash)
            std::fprintf(stderr, "SGHASH %p P0 E1=%d sperm=%016llx samples=%016llx feat1=%016llx adj1=%016llx dist=%016llx\n", (const void*)r.sc->d_data, E1,
                         (unsigned long long)(mode == SG_MODE_INS_INFER ? dev_digest(r.pl->sperm.p, (size_t)r.sc->N * 4) : 0),
                         (unsigned long long)dev_digest(r.pl->samples.p, (size_t)S * 64 * 6 * 4), (unsigned long long)dev_digest(r.pl->feat1.p, (size_t)S * 128 * 4),
                         (unsigned long long)dev_digest(r.pl->adj1.p, (size_t)std::min(E1, r.cap1) * 8), (unsigned long long)dev_digest(r.pl->dist.p, (size_t)std::min(E1, r.cap1) * 4));
#endif
        if (E1 > r.cap1) return sg::fail(SG_ENOMEM, "adjacency capacity exceeded (%d > %d)", E1, r.cap1);
        const float* h_dist = r.o_dist;
        if (E1 > r.out_rows) {                                    // rare: denser than the outbox assumes -- fetch the full arrays
            // into the pipeline's PINNED landing buffers (sized for the adjacency capacity): a pageable destination makes the copy a
            // staged, synchronous one on the group's stream
            EG_HIP(hipMemcpyAsync(r.pl->h_adj.p, r.pl->adj1.p, (size_t)E1 * 8, hipMemcpyDeviceToHost, stream));
            EG_HIP(hipMemcpyAsync(r.pl->h_dist.p, r.pl->dist.p, (size_t)E1 * 4, hipMemcpyDeviceToHost, stream));
            EG_HIP(timed_sync(stream));
            r.adj.assign(r.pl->h_adj.p, r.pl->h_adj.p + 2 * (size_t)E1);
            h_dist = r.pl->h_dist.p;
        } else {
            r.adj.assign(r.o_adj1, r.o_adj1 + 2 * (size_t)E1);
        }
        r.E = E1;
        if (mode == SG_MODE_INS_INFER) r.seg_sums.assign(r.o_seg_sums, r.o_seg_sums + (size_t)S * 3);
        freeze_layer(r.part, S, r.Lcur);                          // layer 1: every segment its own cluster
        r.out->trace[0] = r.Lcur.C;
        EG_CHECK(tables_for(r, 0, true));                         // layer_1.{seg,ins,sem}
        EG_CHECK(regroup(r, h_dist, mode == SG_MODE_SEM_INFER ? 3.0f : 6.0f));
        r.out->trace[1] = r.Lnew.C;
        EG_CHECK(tables_for(r, 3, true));                         // layer_2.*
        r.feat_prev = r.pl->feat1.p; r.feat_prev_stride = 128; r.feat_prev_dim = 128;
    }
    return SG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// L2 / L3: one semantic grouping layer (model.py:786-865)
// ---------------------------------------------------------------------------------------------------------------------
int sg_engine::Group::phase_layer(Run* runs_, int n, int layer) {
    Stages rg;
    rg.next(layer == 0 ? "L2.describe" : "L3.describe");
    par.reset(); box.reset();
    SlotCtx* d_ctx = nullptr;
    SlotCtx* h_ctx = par.take<SlotCtx>(n, &d_ctx);
    BatchDims bd;
    bd.nslots = n;
    long long tiles_total = 0;
    std::vector<int32_t> tmp;
    for (int i = 0; i < n; ++i) {
        Run& r = runs_[i];
        sg_pipeline* pl = r.pl;
        const sg_scene* sc = r.sc;
        const int N = sc->N, S = sc->S, C = r.Lnew.C, E = r.E, Dcat = r.feat_prev_dim + 64;
        const LayerDesc& Ln = r.Lnew;
        const LayerDesc& Lc = r.Lcur;
        SlotCtx& c = r.ctx;
        auto put = [&](const int32_t* v, size_t count) -> const int32_t* {
            int32_t* d = nullptr;
            int32_t* h = par.take<int32_t>(std::max<size_t>(count, 1), &d);
            if (h && count) std::memcpy(h, v, count * 4);
            return d;
        };
        // tiles of 64 consecutive member positions of one cluster, LARGEST clusters first: a tile's cost grows with its cluster
        // (more segments and chunks to test, more candidates), one wave runs one tile, and the launch ends when its slowest
        // wave does -- the long tiles must start first (the tile order is free: every tile names its cluster)
        int T = 0;
        for (int cc = 0; cc < C; ++cc) T += (Ln.cl_pt_off[cc + 1] - Ln.cl_pt_off[cc] + 63) / 64;
        {
            tmp.resize(C);
            for (int cc = 0; cc < C; ++cc) tmp[cc] = cc;
            std::stable_sort(tmp.begin(), tmp.end(), [&](int a, int b) {
                return Ln.cl_pt_off[a + 1] - Ln.cl_pt_off[a] > Ln.cl_pt_off[b + 1] - Ln.cl_pt_off[b];
            });
            int32_t *d_cl = nullptr, *d_lo = nullptr, *d_hi = nullptr;
            int32_t* t_cl = par.take<int32_t>(std::max(T, 1), &d_cl);
            int32_t* t_lo = par.take<int32_t>(std::max(T, 1), &d_lo);
            int32_t* t_hi = par.take<int32_t>(std::max(T, 1), &d_hi);
            if (t_cl && t_lo && t_hi) {
                int t = 0;
                for (int k = 0; k < C; ++k) {
                    const int cc = tmp[k];
                    for (int lo = Ln.cl_pt_off[cc]; lo < Ln.cl_pt_off[cc + 1]; lo += 64, ++t) {
                        t_cl[t] = cc; t_lo[t] = lo; t_hi[t] = std::min(lo + 64, Ln.cl_pt_off[cc + 1]);
                    }
                }
            }
            c.tile_cl = d_cl; c.tile_lo = d_lo; c.tile_hi = d_hi; c.T = T;
        }
        tiles_total += T;
        c.order = put(Ln.order.data(), S); c.dst = put(Ln.dst.data(), S);
        {
            std::vector<int32_t>& big = r.lay_big;                // a block lays out kLayoutPiece rows: list the rest of larger segments
            big.clear();
            if (r.max_seg > sg::kLayoutPiece)
                for (int k = 0; k < S; ++k)
                    for (int r0 = sg::kLayoutPiece; r0 < sc->h_seg_size[Ln.order[k]]; r0 += sg::kLayoutPiece) { big.push_back(k); big.push_back(r0); }
            c.lay_big = put(big.data(), big.size()); c.lay_nbig = (int)(big.size() / 2);
            bd.max_lay_big = std::max(bd.max_lay_big, c.lay_nbig);
        }
        c.seg_chunk_off = put(r.chunk_off.data(), (size_t)S + 1);
        tmp.resize(S);
        for (int k = 0; k < S; ++k) tmp[k] = Ln.cl_of_seg[Ln.order[k]];
        c.cl = put(tmp.data(), S);
        c.cl_pt_off = put(Ln.cl_pt_off.data(), C + 1); c.cl_seg_off = put(Ln.cl_seg_off.data(), C + 1);
        // layer 3 may start from layer 2's table: former clusters of <= 20 points have no kNN list (-1)
        if (layer == 1) {
            for (int sg = 0; sg < S; ++sg) {
                const int pc = Lc.cl_of_seg[sg];
                tmp[sg] = Lc.cl_pt_off[pc + 1] - Lc.cl_pt_off[pc] > 20 ? pc : -1;
            }
            c.seg_prevcl = put(tmp.data(), S);
        } else c.seg_prevcl = nullptr;
        // cluster centroids (combine_centralized_pointcloud, model.py:429-436) from the per-segment coordinate sums
        {
            float* d_m = nullptr;
            float* m = par.take<float>(3 * (size_t)std::max(C, 1), &d_m);
            if (m)
                for (int cc = 0; cc < C; ++cc) {
                    double sx = 0.0, sy = 0.0, sz = 0.0;
                    for (int k = Ln.cl_seg_off[cc]; k < Ln.cl_seg_off[cc + 1]; ++k) {
                        const double* q = r.seg_sums.data() + 3 * (size_t)Ln.order[k];
                        sx += q[0]; sy += q[1]; sz += q[2];
                    }
                    const double cnt = (double)(Ln.cl_pt_off[cc + 1] - Ln.cl_pt_off[cc]);
                    m[3 * cc] = (float)(sx / cnt); m[3 * cc + 1] = (float)(sy / cnt); m[3 * cc + 2] = (float)(sz / cnt);
                }
            c.cl_mean = d_m;
        }
        // parents: old clusters (Lcur numbering) absorbed by each new cluster, in old order (model.py:766-768)
        {
            int32_t *d_goff = nullptr, *d_gidx = nullptr;
            int32_t* goff = par.take<int32_t>((size_t)C + 1, &d_goff);
            int32_t* gidx = par.take<int32_t>(std::max(Lc.C, 1), &d_gidx);
            if (goff && gidx) {
                std::fill(goff, goff + C + 1, 0);
                for (int j = 0; j < Lc.C; ++j) ++goff[Ln.cl_of_seg[Lc.root[j]] + 1];
                for (int cc = 0; cc < C; ++cc) goff[cc + 1] += goff[cc];
                tmp.assign(goff, goff + C);
                for (int j = 0; j < Lc.C; ++j) gidx[tmp[Ln.cl_of_seg[Lc.root[j]]]++] = j;
            }
            c.goff = d_goff; c.gidx = d_gidx;
        }
        // adjacency + symmetric CSR of the cluster graph
        c.g_adj = put(r.adj.data(), 2 * (size_t)E);
        {
            int32_t *d_rp = nullptr, *d_col = nullptr, *d_eid = nullptr;
            int32_t* rowptr = par.take<int32_t>((size_t)C + 1, &d_rp);
            int32_t* col = par.take<int32_t>(std::max<size_t>(2 * (size_t)E, 1), &d_col);
            int32_t* eid = par.take<int32_t>(std::max<size_t>(2 * (size_t)E, 1), &d_eid);
            if (rowptr && col && eid) {
                std::fill(rowptr, rowptr + C + 1, 0);
                for (int e = 0; e < E; ++e) { ++rowptr[r.adj[2 * e] + 1]; ++rowptr[r.adj[2 * e + 1] + 1]; }
                for (int cc = 0; cc < C; ++cc) rowptr[cc + 1] += rowptr[cc];
                tmp.assign(rowptr, rowptr + C);
                for (int e = 0; e < E; ++e) {
                    const int a = r.adj[2 * e], b = r.adj[2 * e + 1];
                    col[tmp[a]] = b; eid[tmp[a]++] = e;
                    col[tmp[b]] = a; eid[tmp[b]++] = e;
                }
            }
            c.rowptr = d_rp; c.col = d_col; c.eid = d_eid;
        }
        c.E = E; c.C = C; c.Dcat = Dcat;
        c.members = pl->members.p; c.pos_of_point = pl->pos_of_point.p; c.point_rec = layer == 1 ? reinterpret_cast<float4*>(pl->point_rec.p) : nullptr;   /* only the seeded (layer-3) kNN reads the records */ c.cluster_of_pos = pl->cluster_of_pos.p; c.slot_of_pos = pl->slot_of_pos.p; c.seed_id = pl->seed_id.p;
        c.x9m = pl->x9m.p; c.sxyzw = reinterpret_cast<float4*>(pl->xyzw.p); c.smpos = pl->smpos.p;
        c.gm_rows = r.feat_prev; c.gm_stride = r.feat_prev_stride; c.gm_D = r.feat_prev_dim; c.cat = pl->featA.p;
        // point 0 is the first member of segment 0; its member-order position is that segment's dst
        c.pos0 = 0;
        for (int k = 0; k < S; ++k) if (Ln.order[k] == 0) { c.pos0 = Ln.dst[k]; break; }
        c.knn = pl->knn.p; c.knn_seed = pl->knn_seed.p;
        const float* W = pl->w.p;
        if (layer == 0) { c.ec_w1 = W + pl->o_m2w; c.ec_g1 = W + pl->o_m2g; c.ec_b1 = W + pl->o_m2b; c.ec_w2 = nullptr; c.ec_g2 = nullptr; c.ec_b2 = nullptr; }
        else { c.ec_w1 = W + pl->o_m3w1; c.ec_g1 = W + pl->o_m3g1; c.ec_b1 = W + pl->o_m3b1; c.ec_w2 = W + pl->o_m3w2; c.ec_g2 = W + pl->o_m3g2; c.ec_b2 = W + pl->o_m3b2; }
        c.ec_partial = r.ec_partial; c.ec_w1f = r.ec_fold; c.ec_sh1 = c.ec_w1f + 64 * 18; c.ec_w2f = c.ec_sh1 + 64; c.ec_sh2 = c.ec_w2f + 64 * 64;
        c.ec_w2img = c.ec_sh2 + 64; c.ec_scale = c.ec_w2img + 4096; c.ec_range = pl->ec_range.p;
        c.K = 20;
        bd.min_K = bd.min_K ? std::min(bd.min_K, c.K) : c.K; bd.max_K = std::max(bd.max_K, c.K);
        bd.gcn_D = Dcat;
        c.pf = pl->pf.p; c.ec_blocks = sg::cdiv(sg::cdiv(N, 32), sg::kEdgeWaves); c.ec_mblocks = sg::moments_blocks(N);
        c.g_wt = W + (layer == 0 ? pl->o_g2t : pl->o_g3t); c.g_dist = r.g_dist; c.g_agg = r.g_agg; c.g_out = pl->featB.p;
        // outbox: decision distances (+ the GCN output of layer 3, which the final clustering reads on the host)
        r.dist_in_outbox = E <= r.out_rows;
        float* d_dist = nullptr;
        r.o_dist = box.take<float>(r.dist_in_outbox ? std::max(E, 1) : 1, &d_dist);
        c.dist_feat = pl->featB.p; c.dist_stride = Dcat; c.dist_D = Dcat; c.dist_adj = c.g_adj; c.dist_E_dev = nullptr; c.dist_E = E;
        c.dist = r.dist_in_outbox ? d_dist : pl->dist.p; c.dist_copy = nullptr; c.dist_copy_rows = 0;
        if (layer == 1) r.o_feat = box.take<float>((size_t)std::max(C, 1) * Dcat, &c.g_out_copy);
        else { r.o_feat = nullptr; c.g_out_copy = nullptr; }
        bd.max_N = std::max(bd.max_N, N); bd.max_S = std::max(bd.max_S, S); bd.max_C = std::max(bd.max_C, C); bd.max_T = std::max(bd.max_T, T);
        bd.max_E = std::max(bd.max_E, E);
    }
    if (!par.ok || !box.ok) return sg::fail(SG_ENOMEM, "engine: a layer phase exceeds the group's parameter / outbox arena");
    for (int i = 0; i < n; ++i) h_ctx[i] = runs_[i].ctx;
    EG_CHECK(arena_copy(par.d, par.h, par.used, true));
    const int sb = 4 + 6 * layer;
    mark(-1);
    rg.next(layer == 0 ? "L2.layout" : "L3.layout");
    EG_CHECK(sg::b_layer_layout(d_ctx, bd, stream));
    EG_CHECK(sg::b_group_max_fill(d_ctx, bd, stream));              // + -inf into the 64 columns the point->cluster max fills below
    mark(sb + 0);
    // which kNN kernel: by the tiles of the whole launch (every variant gives the same table)
    int variant = sg::knn_variant_for((int)std::min<long long>(tiles_total, 1 << 30), runs_[0].pl->knn_variant);
    if (variant == 0) variant = 8;                                  // the two-pass kernel has no batched twin
    const bool seeded = variant == 8 && layer == 1;
    const int waves = variant == 8 ? 1 : variant;
    bool wrote_seed = false;
    // SG_ENGINE_CUMASK=heavy:<n>[:knn] (experiment): the issue-bound launches run on the group's CU-confined stream, so that
    // 256 - n CUs stay free of them for the other groups' latency-bound launches
    auto hop = [&](hipStream_t from, hipStream_t to, hipEvent_t ev) -> int {
        EG_HIP(hipEventRecord(ev, from));
        EG_HIP(hipStreamWaitEvent(to, ev, 0));
        return SG_OK;
    };
    const bool knn_heavy = heavy && heavy_knn;
    rg.next(layer == 0 ? "L2.knn" : "L3.knn");
    if (knn_heavy) EG_CHECK(hop(stream, heavy, ev_fork));
    EG_CHECK(sg::b_cluster_knn(d_ctx, bd, waves, seeded, knn_heavy ? heavy : stream, layer == 0, &wrote_seed));
    if (layer == 0 && !wrote_seed) EG_CHECK(sg::b_knn_seed_points(d_ctx, bd, knn_heavy ? heavy : stream));
    if (!heavy) mark(sb + 2);
    // marks 0 / 1 close the sub-passes (statistics pass(es)), 2 / 3 bracket the EdgeConv launch itself: an interval that starts at mark 2
    // belongs to the stage in front of it (the kNN), the one that ends at mark 3 is the kernel alone
    struct MarkArg { Group* g; int sub0, before, kernel; } ma{this, layer == 0 ? 19 : 21, sb + 2, 24 + layer};
    rg.next(layer == 0 ? "L2.edgeconv" : "L3.edgeconv");
    if (heavy) {
        if (!knn_heavy) EG_CHECK(hop(stream, heavy, ev_fork));
        EG_CHECK(sg::b_edgeconv(d_ctx, bd, layer + 1, nullptr, nullptr, heavy));       // (no stage marks: they are events on the main stream)
        EG_CHECK(hop(heavy, stream, ev_join));
    } else
    EG_CHECK(sg::b_edgeconv(d_ctx, bd, layer + 1, [](void* a, int i) {
        auto* m = static_cast<MarkArg*>(a);
        m->g->mark(i == 2 ? m->before : i == 3 ? m->kernel : m->sub0 + i);
    }, &ma, stream));
    // (the point -> cluster max rides inside the EdgeConv launches, the last BN + LeakyReLU in b_edgeconv's k_cluster_affine)
    mark(sb + 4);
    rg.next(layer == 0 ? "L2.gcn+edge_distance" : "L3.gcn+edge_distance");
    EG_CHECK(sg::b_gcn(d_ctx, bd, 0.125f, stream));
    EG_CHECK(sg::b_edge_distance(d_ctx, bd, stream));
    for (int i = 0; i < n; ++i) {
        Run& r = runs_[i];
        if (!r.dist_in_outbox) {
            EG_HIP(hipMemcpyAsync(r.pl->h_dist.p, r.pl->dist.p, (size_t)r.E * 4, hipMemcpyDeviceToHost, stream));     // pinned (see above)
        }
    }
    mark(sb + 5);
    rg.next(layer == 0 ? "L2.sync" : "L3.sync");
    EG_CHECK(outbox_and_sync());
    rg.next(layer == 0 ? "L2.host_grouping" : "L3.host_grouping");

    // ---- host: grouping on the GCN features (model.py:802-815 / 843-856) ----
    for (int i = 0; i < n; ++i) {
        Run& r = runs_[i];
        const int Dcat = r.feat_prev_dim + 64;
#ifdef SG_ENGINE_DEBUG
        if (g_hash) tl_hash_stream = stream;
        if (g_hash)
            std::fprintf(stderr, "SGHASH %p L%d C=%d E=%d knn=%016llx x9m=%016llx fold=%016llx cat=%016llx gcn=%016llx dist=%016llx\n", (const void*)r.sc->d_data, layer, r.Lnew.C, r.E,
                         (unsigned long long)dev_digest(r.pl->knn.p, (size_t)r.sc->N * 20 * 4), (unsigned long long)dev_digest(r.pl->x9m.p, (size_t)r.sc->N * 12 * 4),
                         (unsigned long long)(layer == 0 ? dev_digest(r.ctx.ec_w1f, 64 * 4) ^ (dev_digest(r.ctx.ec_sh1, 64 * 4) << 1)      /* MLP2: |a| [64] and the shifts [64] only */
                                                                 : dev_digest(r.ctx.ec_w1f, (size_t)((r.ctx.ec_scale + 4) - r.ctx.ec_w1f) * 4)),
                         (unsigned long long)dev_digest(r.pl->featA.p, (size_t)r.Lnew.C * Dcat * 4), (unsigned long long)dev_digest(r.pl->featB.p, (size_t)r.Lnew.C * Dcat * 4),
                         (unsigned long long)dev_digest(r.dist_in_outbox ? (const void*)r.ctx.dist : (const void*)r.pl->dist.p, (size_t)r.E * 4));
        if (g_hash) {
            // the first table seen for (scene, layer) is kept; later ones are compared row by row
            const int N = r.sc->N;
            std::vector<int32_t> knn((size_t)N * 20), seed((size_t)N * 20), cop(N), sid(N);
            (void)fetch(knn.data(), r.pl->knn.p, knn.size() * 4);
            (void)fetch(seed.data(), r.pl->knn_seed.p, seed.size() * 4);
            (void)fetch(cop.data(), r.pl->cluster_of_pos.p, cop.size() * 4);
            (void)fetch(sid.data(), r.pl->seed_id.p, sid.size() * 4);
            std::fprintf(stderr, "SGHASH %p S%d seed=%016llx rec=%016llx sid=%016llx sxyzw=%016llx smpos=%016llx\n", (const void*)r.sc->d_data, layer,
                         (unsigned long long)dev_digest(r.pl->knn_seed.p, (size_t)N * 80), (unsigned long long)(layer == 1 ? dev_digest(r.pl->point_rec.p, (size_t)N * 16) : 0),
                         (unsigned long long)dev_digest(r.pl->seed_id.p, (size_t)N * 4), (unsigned long long)dev_digest(r.pl->xyzw.p, (size_t)N * 16),
                         (unsigned long long)dev_digest(r.pl->smpos.p, (size_t)N * 4));
            const auto key = std::make_pair(scene_key(r.sc), layer);
            std::lock_guard<std::mutex> g(g_dbg_mu);
            auto it = g_dbg_knn.find(key);
            if (it == g_dbg_knn.end()) g_dbg_knn[key] = knn;
            else {
                int shown = 0, rows = 0;
                for (int q = 0; q < N; ++q) {
                    if (std::memcmp(&knn[(size_t)q * 20], &it->second[(size_t)q * 20], 80) == 0) continue;
                    ++rows;
                    if (shown++ >= 3) continue;
                    const int cc = cop[q], n = r.Lnew.cl_pt_off[cc + 1] - r.Lnew.cl_pt_off[cc];
                    std::fprintf(stderr, "SGROW %p L%d row %d cluster %d (%d points, from %d) seed id %d\n   now  :", (const void*)r.sc->d_data, layer, q, cc, n, r.Lnew.cl_pt_off[cc], sid[q]);
                    for (int j = 0; j < 20; ++j) std::fprintf(stderr, " %d", knn[(size_t)q * 20 + j]);
                    std::fprintf(stderr, "\n   first:");
                    for (int j = 0; j < 20; ++j) std::fprintf(stderr, " %d", it->second[(size_t)q * 20 + j]);
                    std::fprintf(stderr, "\n   seeds:");
                    for (int j = 0; j < 20; ++j) std::fprintf(stderr, " %d", seed[(size_t)sid[q] * 20 + j]);
                    std::fprintf(stderr, "\n");
                }
                if (rows) std::fprintf(stderr, "SGROW %p L%d: %d rows differ\n", (const void*)r.sc->d_data, layer, rows);
            }
        }
#endif
        r.Lcur = r.Lnew;
        EG_CHECK(regroup(r, r.dist_in_outbox ? r.o_dist : r.pl->h_dist.p, 2.0f));
        r.out->trace[2 + layer] = r.Lnew.C;
        EG_CHECK(tables_for(r, 6 + 3 * layer, true));             // layer_3.* / layer_4.*
        // next layer: previous features = this GCN output (featB); its concat goes to featA again and its GCN output back
        // into featB -- safe, the stream runs the group max (featB -> featA) before the GCN writes featB
        r.feat_prev = r.pl->featB.p; r.feat_prev_stride = Dcat; r.feat_prev_dim = Dcat;
    }
    return SG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// final clustering (host; FPS-1024 fallback scene by scene, rare), export + evaluate (model.py:868-897, 525-655)
// ---------------------------------------------------------------------------------------------------------------------
int sg_engine::Group::phase_end(Run* runs_, int n, int mode) {
    Stages rg;
    rg.next("END.final_clustering");
    if (mode == SG_MODE_INS_INFER) {
        for (int i = 0; i < n; ++i) {
            Run& r = runs_[i];
            sg_pipeline* pl = r.pl;
            const sg_scene* sc = r.sc;
            const int S = sc->S, D4 = 256;
            // Feat_4 = max over absorbed rows of the gcn_3 output (outbox copy), adj_4 = current adj
            std::vector<float> feat4((size_t)r.Lnew.C * D4, -INFINITY);
            for (int j = 0; j < r.Lcur.C; ++j) {
                float* dstp = &feat4[(size_t)r.Lnew.cl_of_seg[r.Lcur.root[j]] * D4];
                const float* src = r.o_feat + (size_t)j * D4;
                for (int k = 0; k < D4; ++k) dstp[k] = std::max(dstp[k], src[k]);
            }
            std::vector<int32_t> root5(r.Lnew.root.begin(), r.Lnew.root.begin() + r.Lnew.C);
            root5.resize(S);
            int C5 = r.Lnew.C, E5 = r.E;
            r.adj.resize(2 * (size_t)std::max(r.E, 1));
            const int need_fallback = sg_partition_group_unlabeled(r.part, root5.data(), &C5, feat4.data(), D4, r.adj.data(), &E5);
            if (need_fallback < 0) return need_fallback;
            if (need_fallback) {
                // FPS-1024 over the current clusters (model.py:479), XYZ only, no transform: this scene alone, on the group's stream
                LayerDesc L5;
                freeze_layer(r.part, S, L5);
                std::vector<int32_t> cl_of_order(S);
                for (int k = 0; k < S; ++k) cl_of_order[k] = L5.cl_of_seg[L5.order[k]];
                size_t cur = 0;
                auto put = [&](const std::vector<int32_t>& v, size_t count) {
                    const size_t at = cur;
                    std::copy(v.begin(), v.begin() + count, pl->h_desc.p + at);
                    cur += (count + 3) / 4 * 4;
                    return at;
                };
                const size_t o_order = put(L5.order, S), o_dst = put(L5.dst, S), o_cl = put(cl_of_order, S), o_off = put(L5.cl_pt_off, L5.C + 1);
                EG_HIP(hipMemcpyAsync(pl->desc.p, pl->h_desc.p, cur * 4, hipMemcpyHostToDevice, stream));
                const int32_t* dd = pl->desc.p;
                EG_CHECK(sg_gather_members(sc->d_seg_points, sc->d_seg_off, S, dd + o_order, dd + o_dst, dd + o_cl, pl->members.p, nullptr, nullptr,
                                           nullptr, (void*)stream));
                int max_cl = 0;
                for (int cc = 0; cc < L5.C; ++cc) max_cl = std::max(max_cl, L5.cl_pt_off[cc + 1] - L5.cl_pt_off[cc]);
                EG_CHECK(pl->need_fallback_buffers());
                EG_CHECK(sg::fps_sample_hint(sc->d_data, sc->N, 6, pl->members.p, dd + o_off, L5.C, 1024, 3, 0, pl->samples_big.p, nullptr,
                                             pl->ws_fps.p, pl->ws_fps.n, (void*)stream, max_cl));
                EG_HIP(hipMemcpyAsync(pl->h_samples.p, pl->samples_big.p, (size_t)L5.C * 1024 * 3 * 4, hipMemcpyDeviceToHost, stream));
                EG_HIP(timed_sync(stream));
                EG_CHECK(sg_partition_unlabeled_fallback(r.part, L5.root.data(), L5.C, pl->h_samples.p, 1024));
                r.out->used_fallback = 1;
            }
            r.out->trace[4] = sg_partition_num_clusters(r.part);
            EG_CHECK(tables_for(r, 12, false));                   // final.{ins,sem}
            r.n_tables = 14; r.ins_row = 12; r.sem_row = 13;
        }
    }
    par.reset(); box.reset();
    SlotCtx* d_ctx = nullptr;
    SlotCtx* h_ctx = par.take<SlotCtx>(n, &d_ctx);
    BatchDims bd;
    bd.nslots = n;
    for (int i = 0; i < n; ++i) {
        Run& r = runs_[i];
        const int S = r.sc->S;
        SlotCtx& c = r.ctx;
        int32_t* d_tab = nullptr;
        int32_t* tab = par.take<int32_t>((size_t)r.n_tables * S, &d_tab);
        if (tab) std::memcpy(tab, r.tab.data(), (size_t)r.n_tables * S * 4);
        c.tables = d_tab; c.n_tables = r.n_tables; c.labels = r.pl->labels.p; c.sem_row = r.sem_row; c.ins_row = r.ins_row; c.max_ins = r.max_ins;
        r.o_cnt = box.take<uint32_t>(128 + 5 * (size_t)r.max_ins, &c.cnt);
        bd.max_V = std::max(bd.max_V, r.sc->V); bd.max_ins = std::max(bd.max_ins, r.max_ins);
    }
    if (!par.ok || !box.ok) return sg::fail(SG_ENOMEM, "engine: the export phase exceeds the group's parameter / outbox arena");
    for (int i = 0; i < n; ++i) h_ctx[i] = runs_[i].ctx;
    EG_CHECK(arena_copy(par.d, par.h, par.used, true));
    static const bool want_sdma = !(getenv("SG_ENGINE_LABEL_COPY") && std::string(getenv("SG_ENGINE_LABEL_COPY")) == "hip");
    const bool label_sdma = want_sdma && sg::sdma_available();
    std::vector<void*> sd_dst;
    std::vector<const void*> sd_src;
    std::vector<size_t> sd_bytes;
    mark(-1);
    rg.next("END.export+evaluate");
    EG_CHECK(sg::b_export_eval(d_ctx, bd, stream));
    for (int i = 0; i < n; ++i) {
        Run& r = runs_[i];
        // compact transfer: the vectors stay on the device (the evaluate kernels read two of them there); the host has the tables already
        const bool compact = eng->label_compact && r.sc->h_seg_of_vertex;
        if (!compact) {
            if (!r.out->h_labels) return sg::fail(SG_EINVAL, "sg_engine: results[].h_labels is null (and the compact label transfer is off, or the scene has no h_seg_of_vertex)");
            // The label vectors (8.4 MB per 150k-vertex scene).  hipMemcpyAsync moves them with a blit KERNEL on this ROCm -- waves on the CUs waiting for
            // PCIe, 6-7 % of the engine's throughput (sdma.cpp) -- so by default they go over the copy engines instead, behind the stream's sync below
            // (SG_ENGINE_LABEL_COPY=hip restores the stream copy; it is also the fallback when the HSA path is not there or refuses a pointer)
            if (label_sdma) { sd_dst.push_back(r.out->h_labels); sd_src.push_back(r.pl->labels.p); sd_bytes.push_back((size_t)r.n_tables * r.sc->V * 4); }
            else EG_HIP(hipMemcpyAsync(r.out->h_labels, r.pl->labels.p, (size_t)r.n_tables * r.sc->V * 4, hipMemcpyDeviceToHost, stream));
        }
        if (r.out->h_tables) std::memcpy(r.out->h_tables, r.tab.data(), (size_t)r.n_tables * r.sc->S * 4);
    }
    mark(18);
    rg.next("END.sync");
    EG_CHECK(outbox_and_sync());
    if (!sd_dst.empty()) {
        rg.next("END.labels_sdma");
        const long long t0 = g_profile ? now_ns() : 0;
        if (sg::sdma_copy_d2h(sd_dst.data(), sd_src.data(), sd_bytes.data(), (int)sd_dst.size()) != SG_OK) {
            sg::err_buf()[0] = 0;                                    // not an error of the scene: the stream copy does the same job
            for (size_t i = 0; i < sd_dst.size(); ++i) EG_HIP(hipMemcpyAsync(sd_dst[i], sd_src[i], sd_bytes[i], hipMemcpyDeviceToHost, stream));
            EG_HIP(timed_sync(stream));
        }
        if (g_profile) tl_ns_sync += now_ns() - t0;
    }
    for (int i = 0; i < n; ++i) {
        Run& r = runs_[i];
        sg::eval_finish(r.o_cnt, r.max_ins, r.out->iou_sem, r.out->iou_ins, r.out->acc);
    }
    return SG_OK;
}

void sg_engine::Group::collect_times(int n) {
    if (eng->timing == 0 || n_ev < 2) { n_ev = 0; return; }
    double ms_stage[kNumStages] = {0};
    for (int i = 1; i < n_ev; ++i) {
        float ms = 0.f;
        if (ev_stage[i] >= 0 && ev_stage[i] < kNumStages && hipEventElapsedTime(&ms, ev[i - 1], ev[i]) == hipSuccess) ms_stage[ev_stage[i]] += ms;
    }
    n_ev = 0;
    ms_stage[19] += ms_stage[24];                                // the kernel's own interval is part of its statistics sub-pass
    ms_stage[22] += ms_stage[25];
    ms_stage[7] = ms_stage[19] + ms_stage[20];
    ms_stage[13] = ms_stage[21] + ms_stage[22] + ms_stage[23];
    std::lock_guard<std::mutex> g(eng->mu_times);
    for (int k = 0; k < kNumStages; ++k) eng->stage_ms_sum[k] += ms_stage[k];
    eng->scenes_timed += n;
}

int sg_engine::Group::superstep(Run* r, int n, int mode, sg_writer* writer, int formats, long long tag) {
    n_ev = 0;
    int rc = phase_p0(r, n, mode);
    if (rc >= 0) collect_times(0);                               // phases are timed one by one: the event pool is small
    if (rc >= 0 && mode == SG_MODE_INS_INFER) {
        for (int layer = 0; layer < 2 && rc >= 0; ++layer) {
            rc = phase_layer(r, n, layer);
            if (rc >= 0) collect_times(0);
        }
    }
    if (rc >= 0) rc = phase_end(r, n, mode);
    if (rc >= 0) collect_times(n);
    if (rc >= 0 && writer) {
        const int nvec = mode == SG_MODE_INS_INFER ? SG_NUM_LABEL_VECTORS : 6;
        for (int i = 0; i < n && rc >= 0; ++i) {
            if (!r[i].out_dir) continue;
            if (eng->label_compact && r[i].sc->h_seg_of_vertex)     // tables + seg_of_vertex, copied: the worker expands while it formats
                rc = sg_writer_submit_scene_tables(writer, r[i].out_dir, r[i].tab.data(), r[i].sc->S, r[i].sc->h_seg_of_vertex, r[i].sc->V, nvec, formats, tag);
            else   // by reference: the vectors stay in the caller's buffer, which the caller reuses only behind sg_writer_wait_tag(ticket)
                rc = sg_writer_submit_scene(writer, r[i].out_dir, r[i].out->h_labels, r[i].sc->V, nvec, formats, tag);
        }
    }
    for (int i = 0; i < n; ++i) {
        if (r[i].part) sg_partition_destroy(r[i].part);
        r[i].part = nullptr;
    }
    if (rc < 0) {
        // an aborted phase may leave set bits in the contraction bitmaps (they are only cleared by k_emit_pairs): restore the
        // all-zero invariant and drain the stream before the slots are reused
        for (int i = 0; i < n; ++i) (void)hipMemsetAsync(r[i].bitmap, 0, ((size_t)r[i].pl->maxS * r[i].pl->maxS + 31) / 32 * 4, stream);
        (void)hipStreamSynchronize(stream);
        n_ev = 0;
    }
    return rc;
}

void sg_engine::Group::loop() {
    (void)hipSetDevice(eng->device);
    for (;;) {
        std::shared_ptr<Job> job;
        int first = 0, take = 0;
        {
            const long long t_idle = g_profile ? now_ns() : 0;
            std::unique_lock<std::mutex> lk(eng->mu);
            ++eng->idle_groups;
            eng->cv_work.wait(lk, [&] {
                if (eng->stop) return true;
                for (auto& j : eng->jobs) if (j->next < j->count && j->err == 0) return true;
                return false;
            });
            if (eng->stop) { --eng->idle_groups; return; }
            for (auto& j : eng->jobs) if (j->next < j->count && j->err == 0) { job = j; break; }
            // a full group's worth, unless other groups are idle too and the queue is short: then what is queued is shared
            // with them (start of a run, last scenes of a run) instead of leaving them without work
            long long queued = 0;
            for (auto& j : eng->jobs) if (j->err == 0) queued += j->count - j->next;
            const long long share = (queued + eng->idle_groups - 1) / std::max(eng->idle_groups, 1);
            take = (int)std::min<long long>(std::min<long long>((long long)slots.size(), job->count - job->next), std::max<long long>(1, share));
            --eng->idle_groups;
            first = job->next;
            job->next += take;
            if (g_profile) eng->ns_idle += now_ns() - t_idle;
        }
        const long long t_step = g_profile ? now_ns() : 0;
        tl_ns_sync = 0;
        for (int i = 0; i < take; ++i) {
            Run& r = runs[i];
            r.sc = &job->scenes[first + i];
            r.out = &job->results[first + i];
            r.out_dir = job->out_dirs ? job->out_dirs[first + i] : nullptr;
        }
        int rc = SG_OK;
        for (int i = 0; i < take && rc >= 0; ++i) {
            const sg_scene* sc = runs[i].sc;
            if (!(sc->N > 0 && sc->S > 0 && sc->V > 0 && sc->E0 >= 0)) rc = sg::fail(SG_EINVAL, "sg_engine: empty scene");
            else if (sc->N > eng->maxN || sc->S > eng->maxS || sc->E0 > eng->maxE || sc->V > eng->maxV)
                rc = sg::fail(SG_EINVAL, "sg_engine: scene (N=%d S=%d E0=%d V=%d) exceeds the engine capacity (N=%d S=%d E0=%d V=%d)", sc->N, sc->S,
                              sc->E0, sc->V, eng->maxN, eng->maxS, eng->maxE, eng->maxV);
            else if (!runs[i].out->h_labels && !(eng->label_compact && sc->h_seg_of_vertex))
                rc = sg::fail(SG_EINVAL, "sg_engine: results[%d].h_labels is null", first + i);
        }
        if (rc >= 0) rc = superstep(runs.data(), take, job->mode, job->writer, job->formats, job->id);
        if (g_profile) { eng->ns_step += now_ns() - t_step; eng->ns_sync += tl_ns_sync; ++eng->n_steps; eng->n_step_scenes += take; }
        {
            std::lock_guard<std::mutex> lk(eng->mu);
            if (rc < 0 && job->err == 0) { job->err = rc; job->msg = sg_last_error(); }
            job->done += take;
            if (job->err != 0) {                                   // unclaimed scenes of a failed job are dropped
                job->done += job->count - job->next;
                job->next = job->count;
            }
        }
        eng->cv_done.notify_all();
    }
}

namespace {

// The group's streams.  Plain: one non-blocking stream.  The SIDE stream of SG_ENGINE_FORK (phase P0's big-segment chain; measured slower,
// off by default) and its events exist only when that variable is set (no handles for a path that is off).
// SG_ENGINE_CUMASK (experiments of round 6; hipExtStreamCreateWithCUMask -- bit i of the mask is CU i / 8 of XCD i % 8 on
// this part, tools/micro/cu_mask_probe.hip):
//   all:<n>          every group's stream confined to CUs [0, n)                      (control: must cost 256 / n)
//   rot:<h>          group g's stream excludes the h CUs [g h, g h + h) mod 256       (every group leaves a different slice free)
//   halfcu | halfxcd groups alternate between the two halves of every XCD | between XCDs 0-3 and 4-7
//   heavy:<n>[:knn]  main streams unconfined; EdgeConv (and with :knn the in-cluster kNN) on a second stream per group confined to CUs [0, n)
int create_group_streams(sg_engine::Group& grp, int g, int groups) {
    static const char* spec = getenv("SG_ENGINE_CUMASK");
    auto masked = [](hipStream_t* st, const uint32_t* m) {
        return hipExtStreamCreateWithCUMask(st, 8, m) == hipSuccess ? SG_OK : sg::fail(SG_EHIP, "hipExtStreamCreateWithCUMask failed");
    };
    auto range_mask = [](uint32_t* m, int lo, int hi, bool set) {          // bits [lo, hi) mod 256
        for (int b = lo; b < hi; ++b) { const int i = ((b % 256) + 256) % 256; if (set) m[i / 32] |= 1u << (i % 32); else m[i / 32] &= ~(1u << (i % 32)); }
    };
    uint32_t m[8];
    const std::string sp = spec ? spec : "";
    int rc = SG_OK;
    bool want_events = getenv("SG_ENGINE_FORK") != nullptr;
    if (sp.rfind("all:", 0) == 0) {
        std::fill(m, m + 8, 0u); range_mask(m, 0, std::max(8, std::min(256, atoi(sp.c_str() + 4))), true);
        rc = masked(&grp.stream, m);
    } else if (sp.rfind("rot:", 0) == 0) {
        const int h = std::max(0, std::min(224, atoi(sp.c_str() + 4)));
        std::fill(m, m + 8, ~0u); range_mask(m, g * h, g * h + h, false);
        rc = masked(&grp.stream, m);
    } else if (sp == "halfcu") {
        std::fill(m, m + 8, 0u); range_mask(m, (g & 1) * 128, (g & 1) * 128 + 128, true);
        rc = masked(&grp.stream, m);
    } else if (sp == "halfxcd") {
        for (int w = 0; w < 8; ++w) m[w] = (g & 1) ? 0xF0F0F0F0u : 0x0F0F0F0Fu;
        rc = masked(&grp.stream, m);
    } else {
        if (hipStreamCreateWithFlags(&grp.stream, hipStreamNonBlocking) != hipSuccess) rc = sg::fail(SG_EHIP, "hipStreamCreate failed");
        if (rc == SG_OK && sp.rfind("heavy:", 0) == 0) {
            std::fill(m, m + 8, 0u); range_mask(m, 0, std::max(8, std::min(256, atoi(sp.c_str() + 6))), true);
            rc = masked(&grp.heavy, m);
            grp.heavy_knn = sp.find(":knn") != std::string::npos;
            want_events = true;
        } else if (rc == SG_OK && !sp.empty() && sp != "0") rc = sg::fail(SG_EINVAL, "SG_ENGINE_CUMASK=%s: not one of all:<n> rot:<h> halfcu halfxcd heavy:<n>[:knn]", sp.c_str());
    }
    if (rc != SG_OK) return rc;
    (void)groups;
    if (getenv("SG_ENGINE_FORK") && hipStreamCreateWithFlags(&grp.side, hipStreamNonBlocking) != hipSuccess) return sg::fail(SG_EHIP, "hipStreamCreate failed");
    if (want_events && (hipEventCreateWithFlags(&grp.ev_fork, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&grp.ev_join, hipEventDisableTiming) != hipSuccess)) return sg::fail(SG_EHIP, "hipEventCreate failed");
    return SG_OK;
}
}  // namespace

extern "C" {

void sg_engine_destroy(sg_engine* e) {
    if (!e) return;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        e->stop = true;
    }
    e->cv_work.notify_all();
    for (auto& g : e->groups) if (g->th.joinable()) g->th.join();
    debug_reset();
    if (g_profile_print && e->n_steps.load() > 0) {
        const double n = (double)e->n_steps.load(), sc = (double)e->n_step_scenes.load();
        const double step = e->ns_step.load() / n * 1e-6, syn = e->ns_sync.load() / n * 1e-6, idle = e->ns_idle.load() / n * 1e-6;
        fprintf(stderr, "[sg engine profile] %d groups x %d: %.0f super-steps, %.2f scenes each: %.3f ms per super-step = %.3f ms blocked in stream syncs + "
                "%.3f ms host work / launch calls; %.3f ms idle between super-steps\n", e->G, e->B, n, sc / n, step, syn, step - syn, idle);
    }
    for (auto& g : e->groups) {
        for (sg_pipeline* p : g->slots) sg_pipeline_destroy(p);
        for (int i = 0; i < kMaxGroupEvents; ++i) if (g->ev[i]) (void)hipEventDestroy(g->ev[i]);
        if (g->par.h) (void)hipHostFree(g->par.h);
        if (g->par.d) (void)hipFree(g->par.d);
        if (g->box.h) (void)hipHostFree(g->box.h);
        if (g->box.d) (void)hipFree(g->box.d);
        if (g->side) (void)hipStreamDestroy(g->side);
        if (g->heavy) (void)hipStreamDestroy(g->heavy);
        if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
        if (g->ev_join) (void)hipEventDestroy(g->ev_join);
        if (g->stream) (void)hipStreamDestroy(g->stream);
    }
    delete e;
}

sg_engine* sg_engine_create(int maxN, int maxS, int maxE, int maxV, const sg_weights* wt, int groups, int scenes_per_group) {
    if (maxN <= 0 || maxS <= 0 || maxE < 0 || maxV <= 0 || !wt || groups <= 0 || scenes_per_group <= 0 || groups > 64 || scenes_per_group > 64) {
        sg::fail(SG_EINVAL, "sg_engine_create: bad arguments");
        return nullptr;
    }
    if (maxN > SG_MAX_POINTS) {
        sg::fail(SG_EUNSUP, "sg_engine_create: maxN = %d; a scene holds at most %d points (the kNN list keys carry 20 index bits)", maxN, SG_MAX_POINTS);
        return nullptr;
    }
    if (sg_device_count() <= 0) {
        sg::fail(SG_EHIP, "sg_engine_create: no HIP device visible -- the SegGroup hot path has no CPU fallback");
        return nullptr;
    }
    std::unique_ptr<sg_engine, void (*)(sg_engine*)> e(new sg_engine(), sg_engine_destroy);
    e->G = groups; e->B = scenes_per_group;
    e->maxN = maxN; e->maxS = maxS; e->maxE = maxE; e->maxV = maxV;
    if (hipGetDevice(&e->device) != hipSuccess) { sg::fail(SG_EHIP, "hipGetDevice failed"); return nullptr; }
    const size_t S = maxS;
    const size_t maxE1 = std::min<size_t>((size_t)maxE, S * (S - 1) / 2 + 1);
    const size_t out_rows = std::min<size_t>(maxE1, 8 * S + 64);
    for (int g = 0; g < groups; ++g) {
        std::unique_ptr<sg_engine::Group> grp(new sg_engine::Group());
        grp->eng = e.get(); grp->index = g;
        for (int i = 0; i < kMaxGroupEvents; ++i) grp->ev[i] = nullptr;
        if (create_group_streams(*grp, g, groups) != SG_OK) return nullptr;
        for (int i = 0; i < kMaxGroupEvents; ++i)
            if (hipEventCreate(&grp->ev[i]) != hipSuccess) { sg::fail(SG_EHIP, "hipEventCreate failed"); return nullptr; }
        grp->runs.resize(scenes_per_group);
        for (int b = 0; b < scenes_per_group; ++b) {
            sg_pipeline* pl = sg_pipeline_create(maxN, maxS, maxE, maxV, wt, (void*)grp->stream);
            if (!pl) return nullptr;
            grp->slots.push_back(pl);
            e->dev_bytes += pl->dev_bytes;
            Run& r = grp->runs[b];
            r.pl = pl;
            // fixed carve of the slot's workspaces (the single-scene entry points carve the same buffers per call)
            const size_t words = (S * S + 31) / 32, nblk = (words + 1023) / 1024;
            sg::Carver cc(pl->ws_contract.p, pl->ws_contract.n);
            r.bitmap = cc.take<uint32_t>(words);
            r.block_count = cc.take<int>(nblk + 1);
            sg::Carver cm(pl->ws_mlp1.p, pl->ws_mlp1.n);
            r.m1_knn = cm.take<uint8_t>(S * 64 * 10);
            r.m1_partial = cm.take<double>(S * 27);
            r.m1_folded = cm.take<float>(448);
            sg::Carver ce(pl->ws_edge.p, pl->ws_edge.n);
            const size_t nb = (size_t)sg::cdiv(sg::cdiv(maxN, 32), sg::kEdgeWaves), mb = (size_t)sg::cdiv(maxN, 256);
            r.ec_partial = ce.take<double>(std::max(nb * 128, mb * 189));
            r.ec_fold = ce.take<float>(sg::kEdgeFoldFloats);
            sg::Carver cg(pl->ws_gcn.p, pl->ws_gcn.n);
            r.g_dist = cg.take<float>(std::max<size_t>(maxE1, 1));
            r.g_agg = cg.take<float>(S * 256);
            if (!cc.ok || !cm.ok || !ce.ok || !cg.ok) { sg::fail(SG_ENOMEM, "sg_engine_create: workspace carve failed"); return nullptr; }
            if (hipMemset(r.bitmap, 0, words * 4) != hipSuccess) { sg::fail(SG_EHIP, "hipMemset failed"); return nullptr; }
        }
        // parameter arena: SlotCtx array + every slot's descriptors (the single-scene pipeline's descriptor capacity + tables)
        const size_t T = (size_t)maxN / 64 + S + 1;
        const size_t par_slot = sizeof(SlotCtx) + (21 * S + 192 + 3 * T + 6 * maxE1 + 256 + 2 * ((size_t)maxN / sg::kLayoutPiece + 1)) * 4 + SG_NUM_LABEL_VECTORS * S * 4 + 2048;
        const size_t box_slot = 256 + S * 24 + out_rows * 12 + S * 256 * 4 + (128 + 5 * (S + 2)) * 4 + 1024;
        grp->par.cap = (par_slot * scenes_per_group + 255) / 256 * 256;      // whole 256-byte steps: arena_copy moves 16 bytes per thread
        grp->box.cap = (box_slot * scenes_per_group + 255) / 256 * 256;
        if (hipHostMalloc((void**)&grp->par.h, grp->par.cap, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&grp->par.d, grp->par.cap) != hipSuccess ||
            hipHostMalloc((void**)&grp->box.h, grp->box.cap, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&grp->box.d, grp->box.cap) != hipSuccess) {
            sg::fail(SG_ENOMEM, "sg_engine_create: arena allocation failed");
            return nullptr;
        }
        e->dev_bytes += grp->par.cap + grp->box.cap;
        e->groups.push_back(std::move(grp));
    }
    if (hipDeviceSynchronize() != hipSuccess) { sg::fail(SG_EHIP, "hipDeviceSynchronize failed"); return nullptr; }
    for (auto& g : e->groups) {
        sg_engine::Group* gp = g.get();
        gp->th = std::thread([gp] { gp->loop(); });
    }
    return e.release();
}

int sg_engine_submit(sg_engine* e, const sg_scene* scenes, int count, int mode, sg_result* results, sg_writer* writer,
                     const char* const* out_dirs, int formats) {
    if (!e || count < 0 || (count > 0 && (!scenes || !results))) return sg::fail(SG_EINVAL, "sg_engine_submit: bad arguments");
    SG_REQUIRE(mode == SG_MODE_INS_INFER || mode == SG_MODE_SEM_INFER, "sg_engine_submit: bad mode %d", mode);
    auto job = std::make_shared<Job>();
    job->scenes = scenes; job->results = results; job->count = count; job->mode = mode;
    job->writer = (writer && out_dirs) ? writer : nullptr; job->out_dirs = out_dirs; job->formats = formats;
    int id;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        id = job->id = e->next_id++;
        e->jobs.push_back(job);
    }
    e->cv_work.notify_all();
    return id;
}

int sg_engine_wait(sg_engine* e, int ticket) {
    if (!e) return sg::fail(SG_EINVAL, "sg_engine_wait: null engine");
    std::shared_ptr<Job> job;
    {
        std::unique_lock<std::mutex> lk(e->mu);
        for (auto& j : e->jobs) if (j->id == ticket) { job = j; break; }
        if (!job) return sg::fail(SG_EINVAL, "sg_engine_wait: unknown ticket %d", ticket);
        e->cv_done.wait(lk, [&] { return job->done >= job->count; });
        for (auto it = e->jobs.begin(); it != e->jobs.end(); ++it) if ((*it)->id == ticket) { e->jobs.erase(it); break; }
    }
    if (job->err != 0) return sg::fail(job->err, "sg_engine: %s", job->msg.c_str());
    return SG_OK;
}

int sg_engine_set_label_transfer(sg_engine* e, int compact) {
    if (!e || (compact != 0 && compact != 1)) return sg::fail(SG_EINVAL, "sg_engine_set_label_transfer: bad arguments");
    const int prev = e->label_compact;
    e->label_compact = compact;
    return prev;
}

int sg_engine_set_timing(sg_engine* e, int level) {
    if (!e || level < 0 || level > 2) return sg::fail(SG_EINVAL, "sg_engine_set_timing: bad arguments");
    const int prev = e->timing;
    e->timing = level;
    return prev;
}

int sg_engine_set_knn_variant(sg_engine* e, int variant) {
    if (!e) return sg::fail(SG_EINVAL, "sg_engine_set_knn_variant: null engine");
    int prev = -1;
    for (auto& g : e->groups) for (sg_pipeline* p : g->slots) prev = sg_pipeline_set_knn_variant(p, variant);
    return prev;
}

long long sg_engine_stage_times(sg_engine* e, double* h_ms_sum, int capacity, int reset) {
    if (!e) return sg::fail(SG_EINVAL, "sg_engine_stage_times: null engine");
    std::lock_guard<std::mutex> g(e->mu_times);
    for (int k = 0; k < kNumStages && k < capacity && h_ms_sum; ++k) h_ms_sum[k] = e->stage_ms_sum[k];
    const long long n = e->scenes_timed;
    if (reset) {
        for (double& v : e->stage_ms_sum) v = 0.0;
        e->scenes_timed = 0;
    }
    return n;
}

/* development aid: where the group threads' wall time went since the last reset -- out[0] super-steps, [1] scenes, [2] ms
 * inside super-steps, [3] of which blocked in stream synchronisation, [4] ms waiting for work.  enable != 0 switches the
 * accounting on (it is off unless SG_ENGINE_PROFILE is set). */
int sg_engine_profile(sg_engine* e, double* out, int reset, int enable) {
    if (!e) return sg::fail(SG_EINVAL, "sg_engine_profile: null engine");
    if (enable) g_profile = true;
    if (out) {
        out[0] = (double)e->n_steps.load(); out[1] = (double)e->n_step_scenes.load();
        out[2] = e->ns_step.load() * 1e-6; out[3] = e->ns_sync.load() * 1e-6; out[4] = e->ns_idle.load() * 1e-6;
    }
    if (reset) { e->n_steps = 0; e->n_step_scenes = 0; e->ns_step = 0; e->ns_sync = 0; e->ns_idle = 0; }
    return SG_OK;
}

size_t sg_engine_device_bytes(const sg_engine* e) { return e ? e->dev_bytes : 0; }

}  // extern "C"
