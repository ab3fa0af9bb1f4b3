// Batched launches of the scene engine (engine.cpp): ONE launch covers the same stage of several scenes.
//
// Every per-scene kernel of the pipeline has a twin `k_*_b(const SlotCtx* ctx)` whose grid is (blocks of the largest
// scene, scenes): block (x, y) runs the unchanged kernel body for block x of scene y with the arguments read from
// ctx[y] (a uniform address: scalar loads).  A scene whose own grid is smaller lets the surplus blocks exit at once.
// The engine rewrites the SlotCtx array of a group of scenes before each phase and ships it, together with the
// scenes' descriptor blocks, in ONE host-to-device copy ("scene index as grid dimension, per-scene
// CSR offsets").  Host-bound results of a phase (edge distances, contracted adjacency, segment sums, metric counters)
// are written by the kernels into one contiguous "outbox" per group and come back in ONE device-to-host copy.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace sg {

struct SlotCtx {
    // ---- scene inputs (sg_scene) ----
    const float* data;                 // [N,6]
    const int64_t* adj0;               // [E0,2]
    const int32_t* seg_of_point;       // [N]
    const int32_t* seg_points;         // [N]
    const int32_t* seg_off;            // [S+1]
    const int32_t* unmap;              // [V]
    const int32_t* gt;                 // [V,2]
    int N, S, E0, V;

    // ---- a3: point-edge contraction (bitmap is all-zero between scenes: k_emit_pairs clears what it reads) ----
    uint32_t* bitmap;
    unsigned long long bitmap_words;
    int* block_count;
    int bits_blocks;
    int32_t* adj1;                     // [cap1,2] full-capacity result
    int cap1;
    int32_t* adj1_out;                 // outbox copy of the first out_rows rows
    float* dist1_out;                  // outbox copy of the first out_rows distances
    int out_rows;
    int32_t* count;                    // outbox: [0] = E1

    // ---- a4/a5: FPS-64 over the original segments ----
    float* samples;                    // [S,64,6]
    float* ws_fps;

    // ---- once per scene: Morton order inside segments, chunk boxes, segment sums ----
    const int32_t* seg_chunk_off;      // [S+1] (params block)
    // the segments of the two larger size classes, listed by the host (params block): a launch over ALL segments dispatched ~12,000 workgroups of
    // 1,024 threads per batch for the ~60 that had work (round 5's k_bigseg_bucket_b: 183 us against 88 us for the same segments one scene at a time)
    const int32_t* mid_segs;           // segments of 513-2,048 points (FPS on four waves)
    const int32_t* big_segs;           // segments beyond 2,048 points (sixteen-wave FPS, cell-bucketed Morton sort)
    int n_mid, n_big;
    float* segbox;                     // [S,8]
    int32_t* sperm;                    // [N]
    float* chunk_box;
    double* seg_sums;                  // outbox [S,3]
    unsigned long long* sort_keys;     // [2N] scratch of the big-segment sort

    // ---- a6/a7: MLP1 ----
    uint8_t* m1_knn;
    double* m1_partial;
    float* m1_folded;
    float* feat1;                      // [S,128]

    // ---- a8: edge distance of the phase (E from the device when d_E != nullptr) ----
    const float* dist_feat;
    int dist_stride, dist_D;
    const int32_t* dist_adj;
    const int32_t* dist_E_dev;
    int dist_E;
    float* dist;                       // full-capacity / outbox destination
    float* dist_copy;                  // optional second destination (first dist_copy_rows entries)
    int dist_copy_rows;

    // ---- layer layout (member arrays, centred rows, sorted kNN operands) ----
    const int32_t* order;              // params block
    const int32_t* dst;
    const int32_t* lay_big;            // (slot, first row) pairs: the pieces of segments beyond kLayoutPiece rows
    int lay_nbig;
    const int32_t* cl;
    const float* cl_mean;
    int32_t* members;
    int32_t* pos_of_point;
    float4* point_rec;                 // [N] by seed id: xyz + this layer's member position (seeded kNN)
    int32_t* seed_id;                  // [N] by member position: the point's id in the seed tables = its place in the Morton-sorted CSR of the over-segmentation
    int32_t* cluster_of_pos;
    int32_t* slot_of_pos;
    float* x9m;                        // [N,12]
    float4* sxyzw;                     // [N]
    int32_t* smpos;                    // [N]

    // ---- a10: group max of the previous layer's features into the concat buffer ----
    const float* gm_rows;
    int gm_stride, gm_D;
    const int32_t* goff;
    const int32_t* gidx;
    int C;                             // clusters of this layer
    float* cat;                        // [C, Dcat]
    int Dcat;

    // ---- a6/a11: in-cluster kNN ----
    const int32_t* cl_pt_off;
    const int32_t* tile_cl;
    const int32_t* tile_lo;
    const int32_t* tile_hi;
    int T;
    const int32_t* cl_seg_off;
    int pos0;
    int32_t* knn;                      // [N,20]
    int32_t* knn_seed;                 // [N,20] seed ids, rows by seed id (written after layer 2, read by layer 3)
    const int32_t* seg_prevcl;

    // ---- a13: EdgeConv ----
    const float* ec_w1;                // layer weights (device, shared by all scenes)
    const float* ec_g1;
    const float* ec_b1;
    const float* ec_w2;
    const float* ec_g2;
    const float* ec_b2;
    double* ec_partial;
    float* ec_w1f;                     // folded conv1 weights / |a| of a one-layer MLP
    float* ec_sh1;
    float* ec_w2f;                     // |a| of the last layer of MLP3
    float* ec_sh2;
    float* ec_w2img;                   // S2X's conv2 weights, pre-split into fp16 pieces in LDS order (16 KB, k_bn_fold_moments)
    float* ec_scale;                   // {1 / (S T), T, S}: the power-of-two scales of that pass
    unsigned int* ec_range;            // bits of the largest |centred coordinate| / |feature| of the layer (k_layer_layout writes, the last fold kernel of the layer clears)
    float* pf;                         // [N,64] pre-activation maxima
    int K;                             // neighbours per point (20); a run-time value on purpose: as a literal the moments kernel unrolls all slots
    int ec_blocks;                     // ceil(ceil(N/32)/4)
    int ec_mblocks;                    // sg::moments_blocks(N)

    // ---- a14: GCN ----
    const int32_t* g_adj;
    const int32_t* rowptr;
    const int32_t* col;
    const int32_t* eid;
    int E;
    const float* g_wt;
    float* g_dist;
    float* g_agg;
    float* g_out;                      // [C,Dcat]
    float* g_out_copy;                 // outbox copy (layer 3 only) or nullptr

    // ---- a16/a17: export + metric counters ----
    const int32_t* tables;             // params block [n_tables,S]
    int n_tables;
    int32_t* labels;                   // [n_tables,V]
    int sem_row, ins_row;
    int max_ins;
    uint32_t* cnt;                     // outbox [128 + 5 max_ins]
};

// ---- batched launchers: one launch per call, grid.y = nslots -------------------------------------------------------
constexpr int kEdgeWaves = 4;             // tiles (waves) per EdgeConv workgroup: grid and partial-sum sizing (kernels_edgeconv.hip)
constexpr int kLayoutPiece = 1024;       // rows of a segment one layout block walks (k_layer_layout_b / _big_b)

// size classes of the over-segments (kernels_fps.hip, kernels_knn_sorted.hip; the engine lists the classes' segments on the host)
constexpr int kSegSmallMax = 512, kSegMidMax = 2048;

struct BatchDims {                       // maxima over the slots of a group (grid.x sizes)
    int nslots = 0;
    int max_N = 0, max_S = 0, max_E0 = 0, max_V = 0;
    int max_bits_blocks = 0, max_seg = 0;
    int max_C = 0, max_T = 0, max_E = 0, max_ins = 0;
    int max_prevC = 0;
    int max_lay_big = 0;                 // pieces of segments beyond kLayoutPiece rows (layout kernel)
    int gcn_D = 0;                       // feature width of the layer's GCN (every slot of a launch is at the same layer: 192 | 256)
    int max_mid = 0, max_big = 0;        // largest SlotCtx::n_mid / n_big of the group
    int min_K = 0, max_K = 0;            // range of SlotCtx::K over the slots: the hand-scheduled EdgeConv loops are unrolled for K = 20 only
};

int b_contract(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st);
// sorted: k_*_sort_boxes of the same scenes has run on this stream (segments beyond the LDS carve then sample chunk-pruned)
int b_fps64(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st, bool sorted, int which = 3);    // which: 1 = segments <= 2,048 points, 2 = larger, 3 = all
bool fps_has_big_class(const BatchDims& bd);
bool sort_boxes_fits_lds(int max_seg);
int b_sort_boxes(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st, int which = 3);      // which: 1 = segments <= 2,048 points, 2 = larger, 3 = all
int b_mlp1(const SlotCtx* d_ctx, const float* d_w, const float* d_g, const float* d_b, const BatchDims& bd, hipStream_t st);
int b_edge_distance(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st);
int b_layer_layout(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st);
int b_group_max_fill(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st);
int b_cluster_knn(const SlotCtx* d_ctx, const BatchDims& bd, int waves_per_tile, bool seeded, hipStream_t st, bool write_seed = false,
                  bool* wrote_seed = nullptr);
int b_knn_seed_points(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st);
// layers == 1: S1X + fold.  layers == 2: moments + fold, S2X + fold.  `mark(i)` after step i (stage timing).
int b_edgeconv(const SlotCtx* d_ctx, const BatchDims& bd, int layers, void (*mark)(void*, int), void* mark_arg, hipStream_t st);
int b_gcn(const SlotCtx* d_ctx, const BatchDims& bd, float alpha, hipStream_t st);
int b_export_eval(const SlotCtx* d_ctx, const BatchDims& bd, hipStream_t st);

// host half of sg_evaluate: counters (k_eval_counts layout) -> the three reference return values
void eval_finish(const uint32_t* h_cnt, int max_ins, float* h_iou_sem, float* h_iou_ins, float* h_acc);

}  // namespace sg
