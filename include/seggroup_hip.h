/*
 * seggroup_hip.h -- C ABI of libseggroup_hip.so: the MI355X (gfx950) implementation of the SegGroup
 * pseudo-label generation hot path (reference: antao97/SegGroup, seggroup/model.py + seggroup/infer.py).
 *
 * The reference has no FFI; its operator seam is the set of module-level functions that
 * SegModel.forward resolves by global name.  Each entry point below replaces one (or a
 * fused run) of those functions; the citation names the reference lines it stands in for.  A
 * maintainer binds them from Python with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C types only; `d_` pointers are DEVICE pointers (tensor.data_ptr()), `h_` pointers are HOST.
 *   - every function returns 0 on success or a negative SG_E* code; sg_last_error() gives the
 *     message of the calling thread's last failure.  Nothing throws, nothing calls exit().
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  Device entry
 *     points only ENQUEUE work unless documented otherwise.
 *   - indices are int32 inside the library (N < 2^31); the Python layer converts from the
 *     reference's int64.  Features are float32.
 *   - functions never allocate or free caller-visible memory.  Scratch comes from the caller
 *     (`d_ws`, sizes from sg_*_ws_bytes) except for the sg_pipeline_* object, which owns its
 *     buffers between create and destroy.
 */
#ifndef SEGGROUP_HIP_H
#define SEGGROUP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_OK          0
#define SG_EINVAL     -1   /* bad argument / inconsistent sizes                                   */
#define SG_EHIP       -2   /* a HIP runtime call failed                                            */
#define SG_ENOMEM     -3   /* workspace too small / allocation failed                              */
#define SG_ESTALL     -4   /* reference would loop forever (model.py:228-239)       */
#define SG_EUNSUP     -5   /* size outside the supported envelope (documented per function)        */

#define SG_MODE_INS_INFER 0   /* infer.py --ins_infer : all five layers (model.py:684-897)         */
#define SG_MODE_SEM_INFER 1   /* infer.py --sem_infer : returns after layer 2 (model.py:781-783)   */

#define SG_NUM_LABEL_VECTORS 14 /* layer_{1..4}.{seg,ins,sem} + final.{ins,sem} (model.py:525-605) */
#define SG_RANGE_WORDS 256      /* words of every d_range_bits buffer below (zero-initialised by the caller)  */
#define SG_MAX_POINTS (1 << 20) /* points of one scene: the in-cluster kNN's list keys carry 20 index bits; sg_pipeline_create,
                                 * sg_engine_create and the trainers return NULL / SG_EUNSUP above it (the reference has no such limit;
                                 * ScanNet scenes stay below 600k points)                                     */

const char* sg_last_error(void);
int  sg_version(void);
/* number of visible HIP devices (0 when no GPU): lets callers fail loudly before first use */
int  sg_device_count(void);
/* device self-test of the DPP wave reductions the structural-layer kernels use (csrc/wave_ops.h) against the ds_bpermute
 * butterflies they replace; *h_mismatches = 0 on a healthy build.  Synchronises the stream. */
int  sg_selftest_wave_ops(int* h_mismatches, void* stream);
/* device self-test: the kNN's top-K list in double form (v_min_f64 / v_max_f64 insertion) against the 64-bit integer form, ties, signed
 * zeros, tiny positive scores and the empty key included; *h_mismatches = 0 when they agree */
int  sg_selftest_list_insert(int* h_mismatches, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a3  update_adj, first call (model.py:291-302 with model.py:724-733): contract the point-level
 * mesh adjacency through the over-segmentation.  d_adj is the [E,2] int64 tensor of <scene>.adj.pth;
 * d_seg_of_point maps point -> segment number (rank of the segment's first point).  Output rows are
 * (lo,hi) int32 pairs, lexicographically sorted and unique -- the order torch.unique(dim=0) gives.
 * Implementation: one bit per (lo,hi) pair in an S*S bitmap, then an ordered compaction.
 * Supported: S*S <= 2^31 bits.  d_ws needs sg_contract_ws_bytes(S).
 * ------------------------------------------------------------------------------------------- */
size_t sg_contract_ws_bytes(int S);
int sg_contract_point_edges(const int64_t* d_adj, int E, const int32_t* d_seg_of_point, int N, int S,
                            int32_t* d_out_adj, int out_capacity, int32_t* d_out_count,
                            void* d_ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a2/a10  member lists of one layer's clusters (DisjointSet.indexs / get_cluster_list, model.py:
 * 169-214; re-index blocks model.py:759-768).  A cluster is an ORDERED list of original segments
 * (model.py:191 concatenates member lists), so its member list is the concatenation of those
 * segments' ascending point lists.  h-side arrays come from sg_partition_layer().
 *   d_seg_points/d_seg_off : CSR of the original over-segmentation (points ascending per segment)
 *   d_order[S]  original segment ids in cluster-concatenated order
 *   d_dst[S]    destination offset (in points) of each of those segments
 *   d_cl[S]     cluster number of each of those segments
 * writes d_members[N] (point ids in member order), d_pos_of_point[N] (inverse permutation),
 * d_cluster_of_pos[N] and d_slot_of_pos[N] (index into d_order of the segment holding each position);
 * the last three may be NULL.
 * ------------------------------------------------------------------------------------------- */
int sg_gather_members(const int32_t* d_seg_points, const int32_t* d_seg_off, int S,
                      const int32_t* d_order, const int32_t* d_dst, const int32_t* d_cl,
                      int32_t* d_members, int32_t* d_pos_of_point, int32_t* d_cluster_of_pos,
                      int32_t* d_slot_of_pos, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a4+a5  get_cluster_pointcloud + farthest_point_sampling (model.py:319-426).  For every cluster
 * (member CSR d_members/d_cl_off, C clusters): rows = members tiled P/n times, then P%n FPS picks
 * (start at member 0, first pick = farthest from it with the min-distance array RESET to that pick,
 * first-index argmax, fp32 distances (dx2+dy2)+dz2 with individually rounded squares, trailing-zero
 * fix-up of model.py:407-412).  d_data is [N,ch_in] float32 (XYZ first); ch_out (3 or 6) channels are
 * copied.  transform != 0 applies model.py:421-423 (subtract mean XYZ of the P rows, divide by the
 * scalar max |XYZ|).  d_sel (may be NULL) receives the chosen point ids [C,P].
 * d_ws needs sg_fps_ws_bytes(N) (min-distance scratch for clusters that do not fit in LDS).
 * ------------------------------------------------------------------------------------------- */
size_t sg_fps_ws_bytes(int N);
int sg_fps_sample(const float* d_data, int N, int ch_in, const int32_t* d_members, const int32_t* d_cl_off, int C,
                  int P, int ch_out, int transform, float* d_samples, int32_t* d_sel,
                  void* d_ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a6+a7  MLP1 = knn(k=10) + get_graph_feature1 + conv1x1 6->64 + BatchNorm2d(batch statistics over
 * all C*64*10 rows, eps 1e-5, biased variance) + LeakyReLU(0.2) + max_k + [max | mean] over the 64
 * points (model.py:30-80).  d_samples [C,64,6] -> d_feat rows of 128 floats with row stride
 * feat_stride (floats).  d_w [64,6], d_gamma/d_beta [64].  d_ws needs sg_mlp1_ws_bytes(C).
 * ------------------------------------------------------------------------------------------- */
size_t sg_mlp1_ws_bytes(int C);
int sg_mlp1_forward(const float* d_samples, int C, const float* d_w, const float* d_gamma, const float* d_beta,
                    float* d_feat, int feat_stride, void* d_ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a8  calculate_distance (model.py:269-274): d[e] = || F[a] - F[b] + 1e-6 ||_2 for every edge.
 * ------------------------------------------------------------------------------------------- */
int sg_edge_distance(const float* d_feat, int feat_stride, int D, const int32_t* d_adj, int E,
                     float* d_dist, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a10  aggregate_cluster_feature (model.py:278-288): out[g] = element-wise max of rows[idx] over
 * group g (CSR d_goff[G+1], d_gidx).  Columns [0,D) of each output row (stride out_stride).
 * ------------------------------------------------------------------------------------------- */
int sg_group_max_rows(const float* d_rows, int row_stride, int D, const int32_t* d_goff, const int32_t* d_gidx,
                      int G, float* d_out, int out_stride, void* stream);

/* a10 point->cluster max (model.py:793,834): rows are in member order, i.e. clusters are contiguous ranges and
 * d_cluster_of_pos is non-decreasing along the rows (sg_gather_members produces exactly that) */
int sg_segment_max(const float* d_rows, int N, int D, const int32_t* d_cluster_of_pos,
                   float* d_out, int out_stride, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a12  combine_centralized_pointcloud (model.py:429-436), written in MEMBER order:
 *   d_x9m [N,12]  = [XYZ, RGB, XYZ - mean XYZ of the point's cluster, 0,0,0]   (row pos)
 *   d_xyzw [N,4]  = [X, Y, Z, fl(fl(X2+Y2)+Z2)]                              (kNN operand)
 * d_tile_cl / d_tile_lo / d_tile_hi describe T tiles (<=256 consecutive positions of one cluster),
 * d_cl_tile_off[C+1] the tiles of each cluster (from sg_partition_layer); the two-level sum keeps
 * the mean bit-reproducible.  d_ws needs sg_center_ws_bytes(T, C).
 * ------------------------------------------------------------------------------------------- */
size_t sg_center_ws_bytes(int T, int C);
int sg_center_clusters(const float* d_data, int N, const int32_t* d_members, const int32_t* d_cl_off, int C,
                       const int32_t* d_tile_cl, const int32_t* d_tile_lo, const int32_t* d_tile_hi, int T,
                       const int32_t* d_cl_tile_off, float* d_x9m, float* d_xyzw,
                       void* d_ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a6+a11  get_knn (model.py:512-522) with knn (model.py:30-36), k = 20, in member-order positions.
 * Scores are evaluated in the reference's exact fp32 order:
 *   s_ij = ((-xx_j) - (-2 * fma(z_i,z_j, fma(y_i,y_j, fl(x_i*x_j))))) - xx_i
 * Clusters with n <= k list all members in member order and leave the remaining columns pointing at
 * GLOBAL POINT 0 (position pos0) -- the zero-initialised table quirk of model.py:513.
 * d_knn [N,k] int32 positions, row = query position, descending score.
 * ------------------------------------------------------------------------------------------- */
int sg_cluster_knn(const float* d_xyzw, int N, const int32_t* d_cl_off,
                   const int32_t* d_tile_cl, const int32_t* d_tile_lo, const int32_t* d_tile_hi, int T,
                   int k, int pos0, int32_t* d_knn, void* stream);

/* Same result as sg_cluster_knn (bit-identical tables, ties included), with whole over-segments skipped
 * when their bounding box proves that none of their points can enter any lane's top-k.  Tiles for THIS
 * entry point hold <= 64 positions (one workgroup = 64 queries x 4 candidate slices):
 *   d_segbox [S,8]   = {min xyz, max xyz, max |p|^2, 0} per ORIGINAL segment (sg_segment_boxes, once per scene)
 *   d_cl_seg_off[C+1], d_order[S], d_dst[S] : the layer's ordered segment lists (sg_partition_layer)
 *   d_seg_off[S+1]   : CSR offsets of the original segmentation (segment sizes)
 *   d_slot_of_pos[N] : from sg_gather_members (each wave scans its own segment first)                  */
int sg_segment_boxes(const float* d_data, const int32_t* d_seg_points, const int32_t* d_seg_off, int S,
                     float* d_box, void* stream);
int sg_cluster_knn_pruned(const float* d_xyzw, int N, const int32_t* d_cl_off,
                          const int32_t* d_tile_cl, const int32_t* d_tile_lo, const int32_t* d_tile_hi, int T,
                          const int32_t* d_cl_seg_off, const int32_t* d_order, const int32_t* d_dst,
                          const int32_t* d_seg_off, const float* d_segbox, const int32_t* d_slot_of_pos,
                          int k, int pos0, int32_t* d_knn, void* stream);

/* Fastest variant (same tables again).  Once per scene sg_segment_sort_boxes puts the points of every original
 * segment in Morton order (d_sperm[N]: sorted position -> index into d_seg_points; key = 30-bit Morton code inside the
 * segment's box, ties by CSR index) and boxes every run of 32 sorted points (d_chunk_box [(sum_s ceil(size_s/32)), 8];
 * d_seg_chunk_off[S+1] = first chunk of each segment, computed by the caller from the segment sizes).  Per layer
 * sg_knn_operands lays the kNN operand out in that order (d_sxyzw [N,4], d_smpos [N] = member position of each sorted
 * position) and sg_cluster_knn_sorted scans only chunks whose box can still beat a lane's 20th best.  Tiles hold <= 64
 * SORTED positions of one cluster (same ranges as member order: sorting only permutes inside a segment).
 * One launch (a block per segment: box, LDS bitonic sort, chunk boxes) when the largest segment (max_seg points, known
 * to the host) fits a block's LDS (2048 points); segments beyond that take a second launch that buckets by the top 12
 * Morton bits in d_ws (sg_segment_sort_ws_bytes(N) = 16 N bytes; unused otherwise) and sorts every run of cells in LDS.
 * Outputs: d_segbox [S,8], d_sperm [N], d_chunk_box.  d_seg_sums (may be NULL): [S,3] double, the sum of every
 * segment's xyz -- layer-invariant, so the host can form any cluster's centroid from it. */
size_t sg_segment_sort_ws_bytes(int N);
int sg_segment_sort_boxes(const float* d_data, int N, const int32_t* d_seg_points, const int32_t* d_seg_off,
                          const int32_t* d_seg_of_point, int S, const int32_t* d_seg_chunk_off, int max_seg, float* d_segbox,
                          int32_t* d_sperm, float* d_chunk_box, double* d_seg_sums, void* d_ws, size_t ws_bytes, void* stream);
/* One launch for what sg_gather_members + sg_center_clusters + sg_knn_operands produce for a layer (same arrays, same
 * bits): d_cl[i] = cluster of the i-th segment in member order, d_cl_mean [C,3] = the clusters' centroids as fp32
 * (= (float)(sum of the members' xyz in double / count), e.g. from sg_segment_sort_boxes' d_seg_sums).
 * d_point_rec (may be NULL): [N,4] = {x, y, z, bits of the point's member position in this layer}: the 16-byte record
 * sg_cluster_knn_seeded gathers per seed (instead of a row of d_data and an entry of d_pos_of_point).
 * d_seed_id (may be NULL): [N] by member position = the point's SEED ID, its place in the Morton-sorted CSR of the
 * over-segmentation (d_seg_off[s] + rank inside segment s): the same in every layer, consecutive for the queries of a kNN tile,
 * close for points that are close in space.  With d_seed_id the records are indexed by seed id (written in order), without by
 * point id.  Seed tables may use either id space -- whatever map the caller hands to sg_knn_seed_points / sg_cluster_knn_seeded.
 * d_range_bits (may be NULL): SG_RANGE_WORDS words, atomically raised to the bits of the largest |centred coordinate| / |feature| laid out
 * (the atomics are spread over the words; the range is their maximum, and 2 * range bounds every edge difference of the layer): the scale
 * sg_edgeconv_forward_r cuts conv1's operand with. */
int sg_layer_layout(const float* d_data, int N, const int32_t* d_seg_points, const int32_t* d_seg_off, const int32_t* d_sperm, int S,
                    const int32_t* d_order, const int32_t* d_dst, const int32_t* d_cl, const float* d_cl_mean, int32_t* d_members,
                    int32_t* d_pos_of_point, int32_t* d_cluster_of_pos, int32_t* d_slot_of_pos, float* d_x9m, float* d_sxyzw,
                    int32_t* d_smpos, float* d_point_rec, int32_t* d_seed_id, unsigned int* d_range_bits, void* stream);
int sg_knn_operands(const float* d_data, const int32_t* d_seg_points, const int32_t* d_seg_off, const int32_t* d_sperm, int S,
                    const int32_t* d_order, const int32_t* d_dst, float* d_sxyzw, int32_t* d_smpos, void* stream);
int sg_cluster_knn_sorted(const float* d_sxyzw, const int32_t* d_smpos, int N, const int32_t* d_cl_off,
                          const int32_t* d_tile_cl, const int32_t* d_tile_lo, const int32_t* d_tile_hi, int T,
                          const int32_t* d_cl_seg_off, const int32_t* d_order, const int32_t* d_dst,
                          const int32_t* d_seg_off, const int32_t* d_seg_chunk_off, const float* d_segbox,
                          const float* d_chunk_box, const int32_t* d_slot_of_pos, int k, int pos0, int32_t* d_knn, void* stream);
/* Seeded variant for a layer whose clusters are unions of the clusters of the PREVIOUS kNN layer (model.py:829 after
 * 788: the score of a pair depends on raw coordinates only and union() appends whole member lists, so a query's
 * previous list is the exact top k inside its former cluster).  sg_knn_seed_points turns the previous table (rows and
 * entries = member positions of that layer, d_members = its position -> id map: point ids (sg_layer_layout's d_members)
 * or seed ids (its d_seed_id: what the pipeline and the engine pass)) into d_seed [N,k] indexed by and holding those ids.
 * sg_cluster_knn_seeded = sg_cluster_knn_sorted with one wave per tile that starts every query from its seeds (d_members:
 * THIS layer's position -> id map in the same id space, d_point_rec: sg_layer_layout's records of THIS layer by id) and skips the chunks of
 * segments whose former cluster (d_seg_prevcl[S]) is the query's own; d_seg_prevcl = -1 marks segments of former
 * clusters with <= k points, which have no kNN list (model.py:516-518).  Same table as every other variant. */
int sg_knn_seed_points(const int32_t* d_knn, const int32_t* d_members, int N, int k, int32_t* d_seed, void* stream);
int sg_cluster_knn_seeded(const float* d_sxyzw, const int32_t* d_smpos, int N, const int32_t* d_cl_off,
                          const int32_t* d_tile_cl, const int32_t* d_tile_lo, const int32_t* d_tile_hi, int T,
                          const int32_t* d_cl_seg_off, const int32_t* d_order, const int32_t* d_dst,
                          const int32_t* d_seg_off, const int32_t* d_seg_chunk_off, const float* d_segbox,
                          const float* d_chunk_box, const int32_t* d_slot_of_pos, const int32_t* d_seed,
                          const int32_t* d_seg_prevcl, const int32_t* d_members, const float* d_point_rec,
                          int k, int pos0, int32_t* d_knn, void* stream);

/* Two-pass kernel over a cluster-ordered chunk table (same tables once more; variant 0 of sg_pipeline_set_knn_variant:
 * faster on 500k-point scenes and on large segments, on par with the one-pass kernel at 150k / 1.5k).  Per layer the host provides d_slot_chunk0[S+1] (exclusive prefix, in cluster slot order, of the
 * 32-point chunk counts of the segments d_order[slot]), d_cl_chunk_off[C+1] (= slot_chunk0 at each cluster's first
 * slot) and d_tile_chunk0[T] (cluster-relative number of the chunk holding the tile's first sorted position).
 * sg_knn_chunk_table writes d_cc [(slot_chunk0[S]), 8]: {box min xyz, box max xyz, max |p|^2, bits: first sorted
 * position << 6 | points - 1} of every chunk in that order (needs N < 2^25).  sg_cluster_knn_2pass: pass 1 keeps the
 * k best SCORES per query (one v_med3_f32 per list slot), pass 2 rescans with that floor and inserts only the ~k
 * survivors as exact (score, index) keys; 64 chunk boxes are tested per coalesced load and the next surviving
 * chunk's operands are prefetched while the current one is scanned. */
int sg_knn_chunk_table(const int32_t* d_order, const int32_t* d_dst, const int32_t* d_seg_off, const int32_t* d_seg_chunk_off,
                       const float* d_chunk_box, int S, const int32_t* d_slot_chunk0, float* d_cc, void* stream);
int sg_cluster_knn_2pass(const float* d_sxyzw, const int32_t* d_smpos, int N, const int32_t* d_cl_off,
                         const int32_t* d_tile_cl, const int32_t* d_tile_lo, const int32_t* d_tile_hi,
                         const int32_t* d_tile_chunk0, int T, const int32_t* d_cl_chunk_off, const float* d_cc,
                         int k, int pos0, int32_t* d_knn, void* stream);
/* sg_cluster_knn_sorted with an explicit number of waves per 64-query tile (1, 2 or 4; anything else = by tile count:
 * 1 when T >= 2048 tiles fill the GPU, else 2 or 4).  More waves shorten a tile's critical path, but every wave warms
 * up its own top-k list.  Same table for every choice (the GPU tests compare them). */
int sg_cluster_knn_sorted_w(const float* d_sxyzw, const int32_t* d_smpos, int N, const int32_t* d_cl_off,
                            const int32_t* d_tile_cl, const int32_t* d_tile_lo, const int32_t* d_tile_hi, int T,
                            const int32_t* d_cl_seg_off, const int32_t* d_order, const int32_t* d_dst,
                            const int32_t* d_seg_off, const int32_t* d_seg_chunk_off, const float* d_segbox,
                            const float* d_chunk_box, const int32_t* d_slot_of_pos, int k, int pos0, int waves_per_tile,
                            int32_t* d_knn, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a13  get_graph_feature2 + MLP2 / MLP3 (model.py:83-138): edge features [x_j - x_i, x_i] over the
 * k=20 table, conv1x1 18->64 (+ conv1x1 64->64 for layers == 2), BatchNorm2d with batch statistics
 * over all N*k rows, LeakyReLU(0.2), max over k.  fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * d_out [N,64] in member order.  d_w2/d_g2/d_b2 are ignored when layers == 1.
 * d_ws needs sg_edgeconv_ws_bytes(N).
 * ------------------------------------------------------------------------------------------- */
size_t sg_edgeconv_ws_bytes(int N);
int sg_edgeconv_forward(const float* d_x9m, const int32_t* d_knn, int N, int k, int layers,
                        const float* d_w1, const float* d_g1, const float* d_b1,
                        const float* d_w2, const float* d_g2, const float* d_b2,
                        float* d_out, void* d_ws, size_t ws_bytes, void* stream);
/* The same with conv1's operand on fp16 pieces scaled by the layer's range (4 instead of 8 MFMAs per neighbour slot, same accuracy: the
 * pieces keep 22 bits and the scale is a power of two).  d_range_bits: a ZERO-INITIALISED buffer of SG_RANGE_WORDS (256) words whose
 * maximum is the bits of a float R with |x_j - x_i| <= 2 R for every edge and channel -- sg_layer_layout raises them while it writes
 * d_x9m, sg_edge_range computes one from d_x9m alone (max |x9m[p][c] - x9m[0][c]|; the words must be 0 or smaller bounds before).
 * BOTH layer counts use the range: MLP2 scales conv1's operand with it, MLP3's fold kernel reads it to choose the scale of conv1' and
 * writes the fp16 weight image accordingly.  Every kernel involved reads or clears all SG_RANGE_WORDS words (a smaller buffer is read
 * and written out of bounds); the op clears them when it is done.  NULL = sg_edgeconv_forward (bf16 pieces, no range needed).
 * With k == 20 the neighbour-slot loop runs as a hand-scheduled gfx950 instruction stream (csrc/edgeconv_slots_gen.h); other k take
 * the compiler-scheduled loop.  Both compute the same bits. */
int sg_edgeconv_forward_r(const float* d_x9m, const int32_t* d_knn, int N, int k, int layers, const float* d_w1, const float* d_g1,
                          const float* d_b1, const float* d_w2, const float* d_g2, const float* d_b2, float* d_out, void* d_ws,
                          size_t ws_bytes, unsigned int* d_range_bits, void* stream);
/* sg_edgeconv_forward_r with flags: SG_EDGECONV_COMPILER_LOOP forces the compiler-scheduled slot loop also for k == 20 (the
 * cross-check of the hand-scheduled one: tests compare the two bit for bit). */
#define SG_EDGECONV_COMPILER_LOOP 1u
int sg_edgeconv_forward_x(const float* d_x9m, const int32_t* d_knn, int N, int k, int layers, const float* d_w1, const float* d_g1,
                          const float* d_b1, const float* d_w2, const float* d_g2, const float* d_b2, float* d_out, void* d_ws,
                          size_t ws_bytes, unsigned int* d_range_bits, unsigned int flags, void* stream);
int sg_edge_range(const float* d_x9m, int N, unsigned int* d_range_bits, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a14  calculate_similarity + build_similarity_matrix + GCN (model.py:262-265,305-309,141-151):
 * out = relu( ((I + sym(exp(-alpha*d))) row-normalised) @ X @ W^T ), evaluated sparsely over the
 * symmetric CSR (d_rowptr[S+1], d_col, d_eid -> index into the edge list) built by the host.
 * d_ws needs sg_gcn_ws_bytes(S, D, E).
 * ------------------------------------------------------------------------------------------- */
size_t sg_gcn_ws_bytes(int S, int D, int E);
int sg_gcn_forward(const float* d_x, int S, int D, const int32_t* d_adj, int E,
                   const int32_t* d_rowptr, const int32_t* d_col, const int32_t* d_eid,
                   const float* d_w, float alpha, float* d_out, void* d_ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a16  export_{segment,instance,semantic}_label (model.py:525-605), gather only:
 *   out[t][v] = tables[t][ seg_of_point[ unmap[v] ] ]      t < T label vectors, v < V raw vertices
 * d_tables [T,S] int32 holds each original segment's exported value for that vector.
 * ------------------------------------------------------------------------------------------- */
int sg_export_labels(const int32_t* d_unmap, int V, const int32_t* d_seg_of_point, int N,
                     const int32_t* d_tables, int T, int S, int32_t* d_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a17  evaluate (model.py:608-655): integer counts on the device, ratios on the host.
 * d_gt [V,2] int32 (sem, ins), predictions int32 [V].  Writes the three reference return values
 * to HOST buffers h_iou_sem[2*40], h_iou_ins[2*40], h_acc[4]; SYNCHRONISES the stream.
 * max_ins = upper bound (exclusive) of predicted instance ids.  d_ws needs sg_eval_ws_bytes(max_ins).
 * ------------------------------------------------------------------------------------------- */
size_t sg_eval_ws_bytes(int max_ins);
int sg_evaluate(const int32_t* d_gt, const int32_t* d_sem_pred, const int32_t* d_ins_pred, int V, int max_ins,
                float* h_iou_sem, float* h_iou_ins, float* h_acc, void* d_ws, size_t ws_bytes, void* stream);

/* =============================================================================================
 * Host-side grouping engine (a2, a3 later calls, a9, a10 re-index, a15): the serial, order-dependent
 * core of the reference, restated over SEGMENT-level arrays (S <= a few thousand), plain C++.
 * ============================================================================================= */
typedef struct sg_partition sg_partition;

/* DisjointSet + graph initialisation (model.py:169-214,712-721).  seg_first[s] = index of the first
 * point of segment s (strictly ascending), seg_size[s] its point count, seg_ins/seg_sem the weak
 * labels of that first point (-1 = unlabeled). */
sg_partition* sg_partition_create(int S, const int32_t* h_seg_first, const int32_t* h_seg_size,
                                  const int32_t* h_seg_ins, const int32_t* h_seg_sem);
void sg_partition_destroy(sg_partition* p);
int  sg_partition_num_clusters(const sg_partition* p);

/* DisjointSet.union(id1,id2) on cluster roots given as SEGMENT numbers (model.py:181-192);
 * returns 1 if points moved, 0 if no-op / vetoed. */
int sg_partition_union(sg_partition* p, int seg_root1, int seg_root2);
/* find(): root segment of the cluster that currently owns segment s (model.py:178-179) */
int sg_partition_find(const sg_partition* p, int s);
/* current weak labels / point count of the cluster rooted at segment r (stale for dead roots) */
int sg_partition_label(const sg_partition* p, int r, int32_t* ins, int32_t* sem, double* npts);

/* Freeze the current numbering (get_cluster_list order, model.py:209-214; re-index blocks 759-768):
 *   h_root[C]        root segment of cluster c (ascending)
 *   h_cl_of_seg[S]   cluster number of every original segment
 *   h_order[S]       original segments in cluster-concatenated (member) order
 *   h_cl_seg_off[C+1] range of h_order belonging to cluster c
 *   h_cl_pt_off[C+1]  range of member positions (points) belonging to cluster c
 *   h_dst[S]         point offset of h_order[i] in the member array
 * returns C, or a negative error. */
int sg_partition_layer(const sg_partition* p, int32_t* h_root, int32_t* h_cl_of_seg, int32_t* h_order,
                       int32_t* h_cl_seg_off, int32_t* h_cl_pt_off, int32_t* h_dst);

/* group_nearby_clusters (model.py:218-258).  adj rows index the layer frozen in h_root (C clusters).
 * h_connected[E] receives 1 for edges whose endpoints ended in one cluster.  Returns 0, or SG_ESTALL
 * when the reference's pass 2 would never terminate (the sweep that made no progress is the last). */
int sg_partition_group_nearby(sg_partition* p, const int32_t* h_root, int C, const float* h_dist,
                              const int32_t* h_adj, int E, float th, uint8_t* h_connected);

/* update_adj for layers >= 2 (model.py:291-302): contract the edges with h_keep[e] != 0 (NULL = all)
 * from the numbering h_root_old to the CURRENT partition's numbering; sorted unique rows.
 * Returns the number of rows written (<= E) or a negative error. */
int sg_partition_contract(const sg_partition* p, const int32_t* h_root_old, const int32_t* h_adj, int E,
                          const uint8_t* h_keep, int32_t* h_adj_out);

/* group_unlabeled_clusters, first loop (model.py:447-477): iterated nearest-feature-neighbour merges
 * of unlabeled clusters.  h_feat [C,D] / h_adj [E,2] are updated IN PLACE to the new numbering
 * (feature = max of absorbed rows); *C_io / *E_io likewise.  Returns 1 if an unlabeled cluster
 * remains (the caller must then sample with P=1024 and call ..._fallback), 0 if none. */
int sg_partition_group_unlabeled(sg_partition* p, int32_t* h_root_io, int* C_io, float* h_feat, int D,
                                 int32_t* h_adj, int* E_io);
/* second half (model.py:479-507): h_samples [C,1024,3] fp32 XYZ of every cluster in the numbering
 * left by the call above. */
int sg_partition_unlabeled_fallback(sg_partition* p, const int32_t* h_root, int C, const float* h_samples, int P);

/* Export tables for one layer (model.py:525-605): per ORIGINAL segment the values that
 * export_segment_label / export_instance_label / export_semantic_label write for its points. */
int sg_partition_export_tables(const sg_partition* p, int32_t* h_seg_tab, int32_t* h_ins_tab, int32_t* h_sem_tab);

/* =============================================================================================
 * Whole-scene pipeline: SegModel.forward (model.py:684-897) for one scene on one stream.
 * ============================================================================================= */
typedef struct sg_pipeline sg_pipeline;

typedef struct sg_weights {       /* HOST pointers, float32, row-major (checkpoint contract) */
    const float* mlp1_w;  const float* mlp1_g;  const float* mlp1_b;      /* [64,6]  [64] [64]        */
    const float* mlp2_w;  const float* mlp2_g;  const float* mlp2_b;      /* [64,18] [64] [64]        */
    const float* gcn2_w;                                                  /* [192,192]                */
    const float* mlp3_w1; const float* mlp3_g1; const float* mlp3_b1;     /* [64,18] [64] [64]        */
    const float* mlp3_w2; const float* mlp3_g2; const float* mlp3_b2;     /* [64,64] [64] [64]        */
    const float* gcn3_w;                                                  /* [256,256]                */
} sg_weights;

typedef struct sg_scene {         /* one scene, DEVICE-resident inputs (staged by the loader)          */
    int N, S, E0, V;
    const float*   d_data;          /* [N,6]   <scene>.pcl.pth                 (data.py:34)            */
    const int64_t* d_adj;           /* [E0,2]  <scene>.adj.pth                 (model.py:724)          */
    const int32_t* d_seg_of_point;  /* [N]     segment number per point        (.seg.json, model.py:714)*/
    const int32_t* d_seg_points;    /* [N]     CSR of the .seg.json lists: points ascending per segment */
    const int32_t* d_seg_off;       /* [S+1]                                                            */
    const int32_t* d_unmap;         /* [V]     <scene>.unmap.pth               (model.py:533)          */
    const int32_t* d_gt;            /* [V,2]   label/real/raw .label.pth (sem, ins) (model.py:612)     */
    /* HOST, segment level (derived by the loader from weak_label and the segment lists) */
    const int32_t* h_seg_first;     /* [S] first point of each segment                                 */
    const int32_t* h_seg_size;      /* [S]                                                             */
    const int32_t* h_seg_ins;       /* [S] weak instance label of the first point (weak_label[:,1])   */
    const int32_t* h_seg_sem;       /* [S] weak semantic label of the first point (weak_label[:,0])   */
    /* HOST, optional (NULL = not given): the over-segment of every raw vertex, seg_of_point[unmap[v]] (-1 where unmap[v] is not a
     * point).  Needed by the engine's compact label transfer (sg_engine_set_label_transfer): a label vector is a table look-up
     * through this array, so the look-up can run on the host and only the [14,S] tables have to cross PCIe. */
    const int32_t* h_seg_of_vertex; /* [V]                                                             */
} sg_scene;

typedef struct sg_result {        /* HOST outputs                                                       */
    int32_t* h_labels;              /* [14,V] int32, pinned or pageable; order: layer_1.seg, layer_1.ins,
                                       layer_1.sem, layer_2.*, layer_3.*, layer_4.*, final.ins, final.sem.
                                       sem_infer fills the first 6 and leaves the rest untouched.        */
    float iou_sem[80];              /* [2,40] I then U  (model.py:628)                                   */
    float iou_ins[80];              /* [2,40]           (model.py:640)                                   */
    float acc[4];                   /* model.py:654                                                      */
    int32_t trace[5];               /* cluster counts of layers 1..5                                     */
    int32_t stalled;                /* 1 if a pass-2 sweep was cut short (SG_ESTALL condition)           */
    int32_t used_fallback;          /* 1 if the FPS-1024 fallback of model.py:479-494 ran                */
    int32_t* h_tables;              /* optional [14,S] int32 (NULL = not wanted): the label tables the vectors are looked up in,
                                       h_labels[t][v] = s >= 0 ? h_tables[t][s] : -1 with s = h_seg_of_vertex[v] (sg_expand_labels) */
} sg_result;

typedef struct sg_debug {         /* optional taps for stage-level parity tests (every pointer may be NULL) */
    float*   d_samples1;            /* DEVICE [S,64,6]                                                   */
    float*   d_feat1;               /* DEVICE [S,128]                                                    */
    float*   d_pointfeat[2];        /* DEVICE [N,64] MLP2 / MLP3 output, MEMBER order of that layer      */
    int32_t* d_knn[2];              /* DEVICE [N,20] member-order positions                              */
    int32_t* d_members[2];          /* DEVICE [N]    point id at each member-order position              */
    float*   h_gcn[2];              /* HOST [S2,192] / [S3,256]                                          */
    float*   h_dist[3];             /* HOST distance vectors of the three group_nearby calls            */
    int32_t* h_adj[4];              /* HOST adjacency lists adj_1..adj_4 ([E,2])                         */
    int32_t  n_adj[4];              /* rows written to h_adj[i]                                          */
    /* train-mode inputs of the classifier tail (model.py:900-914), ins_infer forward only: */
    float*   h_feat5;               /* HOST [S,256] Feat_5: features of the final clusters                */
    int32_t* h_ins5;                /* HOST [S] weak instance label of every final cluster (-1 = none)    */
    int32_t* h_sem5;                /* HOST [S] weak semantic label of every final cluster                */
    int32_t  n5;                    /* final clusters written                                             */
    void*    tape;                  /* library-internal (sg_trainer): the training step's record of this forward; NULL otherwise */
} sg_debug;

sg_pipeline* sg_pipeline_create(int max_points, int max_segments, int max_edges, int max_vertices,
                                const sg_weights* w, void* stream);
void sg_pipeline_destroy(sg_pipeline* pl);
/* device + pinned bytes owned by the pipeline (for capacity planning against 288 GB HBM) */
size_t sg_pipeline_device_bytes(const sg_pipeline* pl);

/* SegModel.forward for one scene.  Blocks the calling thread until the results are in `out`
 * (it synchronises `stream` at each of the 3-5 points where the serial grouping needs distances).
 * Thread-safe across DIFFERENT pipeline objects (one pipeline per in-flight scene). */
int sg_pipeline_forward(sg_pipeline* pl, const sg_scene* scene, int mode, sg_result* out, sg_debug* dbg);

/* Many scenes through `npipes` pipelines: one native host thread per pipeline pulls scenes until all `count` are
 * done (the infer.py:149-152 loop without the interpreter in it).  results[i].h_labels must point at [14,V_i]
 * int32 host buffers, vector v at h_labels + v * V_i (pinned for full D2H speed).  h_stage_ms_sum (may be NULL) accumulates the per-stage device
 * times of every forward (same order as sg_pipeline_stage_name).  Blocks until every scene is done. */
struct sg_writer;   /* asynchronous label-file writer pool, declared below */
/* writer / out_dirs / formats (all optional): after each scene its label vectors are submitted to `writer` as
 * <out_dirs[i]>/<layer_k.seg|ins|sem, final.ins|sem>.{txt,npy} (formats: 1 txt | 2 npy); directories must exist. */
int sg_batch_forward(sg_pipeline* const* pipes, int npipes, const sg_scene* scenes, int count, int mode,
                     sg_result* results, float* h_stage_ms_sum,
                     struct sg_writer* writer, const char* const* out_dirs, int formats);

/* =============================================================================================
 * Scene engine: SegModel.forward for MANY scenes with the scene index as a grid dimension (infer.py:149-152 loop body,
 * model.py:684-897).  `groups` persistent host threads (one HIP stream each) pull up to `scenes_per_group` scenes at a
 * time from a job queue and advance them in lock-step: every kernel is launched ONCE per phase for all scenes of the
 * group (csrc/engine_ctx.h), with one host->device copy, one device->host copy and one stream synchronisation per
 * phase; the serial grouping of a group's scenes runs on its thread while the other groups' kernels occupy the GPU.
 * groups x scenes_per_group scenes are in flight.  Results are bit-identical to sg_pipeline_forward (same kernel
 * bodies; tests/test_gpu_scene.py).
 *
 * sg_engine_submit enqueues `count` scenes and returns a ticket at once (the arrays must stay valid until the ticket
 * has been waited for; results[i].h_labels as for sg_batch_forward); sg_engine_wait blocks until every scene of the
 * ticket is done and returns the first error.  A driver keeps the GPU busy across batches by submitting batch k+1
 * before it waits for batch k.  writer / out_dirs / formats as for sg_batch_forward.
 * ============================================================================================= */
typedef struct sg_engine sg_engine;
sg_engine* sg_engine_create(int max_points, int max_segments, int max_edges, int max_vertices, const sg_weights* w,
                            int groups, int scenes_per_group);
void sg_engine_destroy(sg_engine* e);
int sg_engine_submit(sg_engine* e, const sg_scene* scenes, int count, int mode, sg_result* results,
                     struct sg_writer* writer, const char* const* out_dirs, int formats);
int sg_engine_wait(sg_engine* e, int ticket);
/* 0 = no stage timing (default), 1 | 2 = HIP events around the stages of every batched launch.  Returns the previous level. */
int sg_engine_set_timing(sg_engine* e, int level);
/* 0 (default): every scene's 14 label vectors are copied to results[i].h_labels (8.4 MB per 150k-vertex scene).  1 = compact: the
 * vectors stay on the device; results[i].h_tables (if given) receives the [14,S] tables, the writer pool is handed tables + the scene's
 * h_seg_of_vertex and expands while it writes, results[i].h_labels is left untouched (may be NULL).  Scenes without h_seg_of_vertex
 * fall back to the full copy.  Returns the previous setting.  (Round 4: at the headline rate the full copy is 24 GB/s of D2H per GPU.) */
int sg_engine_set_label_transfer(sg_engine* e, int compact);
/* sg_pipeline_set_knn_variant for every slot (the two-pass kernel, 0, has no batched twin: the seeded kernel runs instead) */
int sg_engine_set_knn_variant(sg_engine* e, int variant);
/* accumulated device time per stage (ms; a batched launch counts once, whatever the number of scenes in it) since the
 * last reset, in sg_pipeline_stage_name order; returns the number of scenes those launches covered */
long long sg_engine_stage_times(sg_engine* e, double* h_ms_sum, int capacity, int reset);
size_t sg_engine_device_bytes(const sg_engine* e);
/* development aid: wall time of the group threads since the last reset -- out[0] super-steps, [1] scenes, [2] ms inside
 * super-steps, [3] of which blocked in stream synchronisation, [4] ms waiting for work; enable != 0 turns the accounting on */
int sg_engine_profile(sg_engine* e, double* out, int reset, int enable);

/* per-stage device time of the last forward, in milliseconds (HIP events on the pipeline's stream);
 * names via sg_pipeline_stage_name(i), count returned. */
int sg_pipeline_stage_times(const sg_pipeline* pl, float* h_ms, int capacity);
/* How many HIP events a forward records for sg_pipeline_stage_times: 2 (default) one after every stage, 1 only around
 * the in-cluster kNN and the EdgeConv passes (the other stages read 0), 0 none.  With many pipelines in flight the
 * ~25 events of level 2 cost ~6 % of the throughput.  Returns the previous level. */
int sg_pipeline_set_timing(sg_pipeline* pl, int level);
/* Which in-cluster kNN kernel THIS pipeline uses for a layer of T tiles (all variants give the same table; the GPU
 * tests compare them).  Per pipeline: the library keeps no process-wide state.
 *   -1  by tile count (default): 8 when T >= 2048 tiles fill the GPU, else one-pass with 2 or 4 waves per tile
 *    0  two-pass (sg_cluster_knn_2pass)
 *    1 | 2 | 4  one-pass (sg_cluster_knn_sorted_w) with that many waves per 64-query tile
 *    8  one-pass, 1 wave per tile, seeded from the previous kNN layer where there is one (sg_cluster_knn_seeded:
 *       layer 3 starts from layer 2's table)
 * Returns the previous setting. */
int sg_pipeline_set_knn_variant(sg_pipeline* pl, int variant);
const char* sg_pipeline_stage_name(int i);

/* =============================================================================================
 * Output writers (a16 file side, model.py:536-547): one decimal integer per line, '\n' terminated;
 * and the .npy twin (v1.0 header, '<i4', shape (V,)).  Host only.
 * ============================================================================================= */
int sg_write_label_txt(const char* path, const int32_t* h_vec, int V);
int sg_write_label_npy(const char* path, const int32_t* h_vec, int V);

/* Asynchronous writer pool: submit() copies h_vec and returns; `formats` = 1 txt | 2 npy | 3 both, written to
 * <path_without_ext>.txt / .npy by one of `threads` native threads.  At most `max_queue` vectors are pending
 * (submit blocks beyond that).  flush() waits for everything submitted so far and reports the first error. */
/* ---------------------------------------------------------------------------------------------
 * Native scene-pack loader (csrc/loader.cpp; reference data.py:28-38 + the per-forward file reads of model.py:696-724).
 * `threads` workers, each with a pinned staging buffer of `slot_bytes` (>= the largest pack file) and a copy stream; `slots` device blobs
 * of 3 x `slot_bytes` (one allocation) made at creation (nothing allocates per scene): packs written from round 4 on store the [E0,2] adjacency as int32
 * (half the bytes of the reference's int64 rows) and the loader widens it to int64 behind the upload, inside the slot -- sg_scene.d_adj is
 * int64 either way; packs with int64 rows still load.  sg_loader_submit queues a `.sgpack` (seggroup_amd/cache.py) and returns a ticket at once;
 * sg_loader_wait blocks until that pack is resident and fills *out (device arrays inside the slot's blob, the four per-segment host
 * arrays and h_seg_of_vertex owned by the slot), *slot and the scene's name.  The slot belongs to the caller until
 * sg_loader_release(slot): release it when the engine has finished the scene.  A worker takes a job only when a slot is free.
 * Call on the thread / device context the engine uses (the device current at creation is the loader's).
 * ------------------------------------------------------------------------------------------- */
typedef struct sg_loader sg_loader;
sg_loader* sg_loader_create(int threads, int slots, size_t slot_bytes);
/* The same with the device slots sized for packs of at most `max_edges` adjacency rows (slot_bytes + 16 max_edges instead of 3 x slot_bytes;
 * 0 = unknown = sg_loader_create).  A pack with more edges fails its ticket with SG_ENOMEM. */
sg_loader* sg_loader_create_sized(int threads, int slots, size_t slot_bytes, size_t max_edges);
/* At most `uploads_in_flight` workers (default 2; env SG_LOADER_COPIES) are between the start and the end of a pack's host-to-device copy at a
 * time: more bulk copies in flight saturate the copy engines and starve the scene engine's own small transfers (csrc/loader.cpp). */
int  sg_loader_set_copy_limit(sg_loader* l, int uploads_in_flight);
int  sg_loader_submit(sg_loader* l, const char* pack_path);
int  sg_loader_wait(sg_loader* l, int ticket, sg_scene* out, int* slot, char* name, int name_capacity);
int  sg_loader_release(sg_loader* l, int slot);
void sg_loader_destroy(sg_loader* l);

typedef struct sg_writer sg_writer;
sg_writer* sg_writer_create(int threads, int max_queue);
int  sg_writer_submit(sg_writer* w, const char* path_without_ext, const int32_t* h_vec, int V, int formats);
/* A whole scene BY REFERENCE (model.py:533-547: the 14 files of one export directory): h_labels = nvec vectors of V values, stride V,
 * written as <out_dir>/<layer_1.seg ... final.sem>.{txt,npy} by one worker (openat on the directory's descriptor, writev).  Nothing is
 * copied: the buffer must stay untouched until sg_writer_wait_tag(w, tag) or sg_writer_flush returned.  tag >= 0, ascending over time
 * (the engine passes its ticket number); sg_writer_wait_tag blocks until every scene with a tag <= `tag` is on disk. */
int  sg_writer_submit_scene(sg_writer* w, const char* out_dir, const int32_t* h_labels, int V, int nvec, int formats, long long tag);
/* The same scene given as label TABLES [nvec,S] + the over-segment of every vertex [V] (both copied: ~0.7 MB instead of a reference
 * to 8.4 MB of vectors that had to cross PCIe first); the worker expands vector by vector while it formats.  Files byte-identical to
 * sg_writer_submit_scene's. */
int  sg_writer_submit_scene_tables(sg_writer* w, const char* out_dir, const int32_t* h_tables, int S, const int32_t* h_seg_of_vertex, int V,
                                   int nvec, int formats, long long tag);
/* h_out[t][v] = (s >= 0 && s < S) ? h_tables[t][s] : -1, s = h_seg_of_vertex[v]: what k_export computes on the device (model.py:525-605) */
int  sg_expand_labels(const int32_t* h_tables, int nvec, int S, const int32_t* h_seg_of_vertex, int V, int32_t* h_out);
int  sg_writer_wait_tag(sg_writer* w, long long tag);
int  sg_writer_flush(sg_writer* w);
void sg_writer_destroy(sg_writer* w);

/* =============================================================================================
 * Raw scan -> hot-path inputs (reference seggroup/dataset/scannet/util.py).  The compute parts
 * of the reference's offline pre-processing; file formats stay on the Python side (seggroup_amd/prepare.py).
 * Index types follow the reference's tensors: mapper / unmapper / adjacency are int64 (LongTensor), the raw
 * inputs (PLY faces, segIndices) int32.  Every function synchronises the stream before it returns host counts.
 * ============================================================================================= */

/* get_unmapper (util.py:538-550, cal_pairwise_distance 530-535): d_idx[u] = argmax_j ((-|x_u|^2) - (-2 x_u.y_j)) - |y_j|^2
 * in the reference's fp32 operation order, lowest j among equal maxima.  d_x [U,3]; d_y rows of `y_stride` floats
 * (>= 3: xyz first).  Brute force: U*N pair evaluations. */
size_t sg_nearest_point_ws_bytes(int N);
int sg_nearest_point(const float* d_x, int U, const float* d_y, int y_stride, int N, int64_t* d_idx,
                     void* d_ws, size_t ws_bytes, void* stream);

/* generate_pointcloud_pth (util.py:633-693) without the file I/O and the random draw: given the mapper (which
 * raw vertex every sampled point copies), d_pcl [Np,6] = [xyz, rgb / 127.5 - 1 (evaluated in double, stored fp32)]
 * and d_unmap [V] = the LAST sampled point that copies the vertex (687-689) or, for a vertex that was not sampled,
 * its nearest sampled point (sg_nearest_point against d_pcl).  *h_unsampled = number of such vertices. */
size_t sg_prep_sample_ws_bytes(int V, int Np);
int sg_prep_sample_points(const float* d_xyz, const uint8_t* d_rgb, int V, const int64_t* d_mapper, int Np,
                          float* d_pcl, int64_t* d_unmap, int* h_unsampled, void* d_ws, size_t ws_bytes, void* stream);

/* get_adj_from_pointcloud (dataset/scannet/util.py:814-834; optional in the reference -- nothing calls it): the k nearest
 * neighbours of every point against the whole cloud in cal_pairwise_distance's exact fp32 formula (the best-scoring entry of
 * topk(k + 1), normally the point itself, is dropped), as per-row sorted, lexicographically sorted, unique [*, 2] int64 rows.
 * d_points rows of `stride` floats, xyz first; k in {5, 10, 20}; d_adj needs room for [N*k,2]; *h_n = rows written.
 * Equal scores: the lower index ranks first (torch.topk leaves that order unspecified). */
size_t sg_pointcloud_adjacency_ws_bytes(int N, int k);
int sg_pointcloud_adjacency(const float* d_points, int stride, int N, int k, int64_t* d_adj, int* h_n, void* d_ws, size_t ws_bytes, void* stream);

/* get_adj_from_mesh (util.py:771-792).  d_faces [F,3]: edges (0,1), (0,2), (1,2) of every face, zero-length ones
 * dropped (783); d_adj_raw = ascending ids per row, unique rows in lexicographic order; d_adj_res (may be NULL)
 * = the same rows mapped through d_unmap first (rows that collapse to (a, a) only then are kept, as in the
 * reference).  Both need room for [3F,2] int64; the row counts come back in *h_n_raw / *h_n_res. */
size_t sg_mesh_adjacency_ws_bytes(int F);
int sg_mesh_adjacency(const int32_t* d_faces, int F, const int64_t* d_unmap, int V, int64_t* d_adj_raw, int* h_n_raw,
                      int64_t* d_adj_res, int* h_n_res, void* d_ws, size_t ws_bytes, void* stream);

/* generate_seg_labels_and_ds_set (util.py:174-220) without the file I/O.  d_seg_indices [V] (non-negative raw
 * segment ids) -> d_raw_label [V] = rank of the id among the sorted unique ids (the `.seg.txt` column), and the
 * member lists of the SAMPLED cloud as a CSR: d_seg_points [Np] grouped by ascending compacted id, ascending point
 * index inside a group, d_seg_off [G+1] (room for min(V, Np) + 1).  h_counts[0] = number of raw segments,
 * h_counts[1] = G (a segment none of whose vertices was sampled has no group). */
size_t sg_segment_lists_ws_bytes(int V, int Np);
int sg_segment_lists(const int32_t* d_seg_indices, int V, const int64_t* d_mapper, int Np, int32_t* d_raw_label,
                     int32_t* d_seg_points, int32_t* d_seg_off, int* h_counts, void* d_ws, size_t ws_bytes, void* stream);

/* `.seg.json` exactly as json.dump writes it (util.py:205-220): one list per sampled point, a segment's members at
 * the index of its smallest member, [] elsewhere.  Host arrays of sg_segment_lists (any group order). */
int sg_write_seg_json(const char* path, const int32_t* h_seg_points, const int32_t* h_seg_off, int G, int Np);

/* =============================================================================================
 * Training step, operator level: the train-mode tail of SegModel.forward + its backward, and the backward
 * of every operator in front of it (group max, point->cluster max, GCN, EdgeConv MLP2 / MLP3, MLP1); further down the whole step
 * as one object (sg_trainer_*) and the optimizers.  The DDP gradient all-reduce (train.py:88) is the host side's: one
 * torch.distributed all-reduce of the flat gradient vector over RCCL (seggroup_amd/trainer.py).
 * ============================================================================================= */
typedef struct sg_classifier {    /* DEVICE pointers, float32, row-major (model.py:154-166)                         */
    const float* w1;                /* linear1.weight [128,256] (no bias)                                          */
    const float* gamma;             /* bn1.weight [128]                                                            */
    const float* beta;              /* bn1.bias   [128]                                                            */
    const float* w2;                /* linear2.weight [40,128]                                                     */
    const float* b2;                /* linear2.bias   [40]                                                         */
} sg_classifier;

/* model.py:900-932 with Classifier.forward (154-166) and cross_entropy_loss(smoothing=True) (util.py:12-29):
 *   d_feat5 [C,256]   Feat_5, the final clusters' features
 *   d_group [C]       instance slot of every final cluster: rank of its weak instance label among the sorted unique labels
 *                     (np.unique(ins_list), model.py:909; an unlabeled cluster's -1 is a label like any other)
 *   d_gold  [K]       semantic class of every slot (the first cluster's, model.py:913-914)
 *   d_keep  [K,128]   dropout keep mask ALREADY scaled by 1 / (1 - p) (0 or 2 for p = 0.5); NULL = no dropout.  The reference
 *                     draws it from torch's RNG stream; parity is stated with the same mask on both sides.
 * BatchNorm1d uses batch statistics over the K instances (K < 2: SG_EUNSUP, torch raises ValueError there).
 * Writes d_loss[2] = {loss_sum, K} (the reference's `loss [1,2]`), optionally d_logits [K,40]; keeps the activations the
 * backward needs in d_ws (sg_train_tail_ws_bytes). */
size_t sg_train_tail_ws_bytes(int C, int K);
int sg_train_tail_forward(const float* d_feat5, int C, const int32_t* d_group, int K, const int32_t* d_gold, const float* d_keep,
                          const sg_classifier* cls, float* d_logits, float* d_loss, void* d_ws, size_t ws_bytes, void* stream);
/* gradients of loss = scale * loss_sum (train.py:166: scale = 1 / loss_num) w.r.t. the five classifier tensors and Feat_5
 * [C,256]; d_ws as left by the forward call of the same scene */
int sg_train_tail_backward(int C, int K, const int32_t* d_gold, const float* d_keep, const sg_classifier* cls, float scale,
                           float* d_gw1, float* d_ggamma, float* d_gbeta, float* d_gw2, float* d_gb2, float* d_gfeat5,
                           void* d_ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The reference's module-level functions with arguments its own forward never passes (the Python
 * surface keeps the reference's signatures: seggroup_amd/functional.py).  Plain kernels, off every
 * timed path (csrc/kernels_general.hip).
 *   sg_group_mean_rows : aggregate_cluster_feature(use_avg=True), model.py:282-284 -- the mean of each
 *       group's rows (arguments as sg_group_max_rows); the caller concatenates [max | mean].
 *   sg_fps_general     : farthest_point_sampling(pts, k, initial_idx, skip_initial), model.py:329-395 for one
 *       cloud d_pts [n,dim]: d_indices [k]; d_distances [k,n] or NULL (the reference's second return value);
 *       l2_norm in NumPy's order, np.argmax's first-index ties.  d_ws needs sg_fps_general_ws_bytes(n).
 *   sg_knn_general     : knn(x, k), model.py:30-36 for x [B,C,n] of any channel count: d_idx [B,n,k] int64, scores
 *       (-|a|^2 - (-2 a.b)) - |b|^2 descending, lower index first among equal scores (torch.topk leaves that open);
 *       1 <= k <= min(n, 128), otherwise SG_EUNSUP.
 * ------------------------------------------------------------------------------------------- */
int sg_group_mean_rows(const float* d_rows, int row_stride, int D, const int32_t* d_goff, const int32_t* d_gidx, int G,
                       float* d_out, int out_stride, void* stream);
size_t sg_fps_general_ws_bytes(int n);
int sg_fps_general(const float* d_pts, int n, int dim, int k, int initial_idx, int skip_initial, int32_t* d_indices,
                   float* d_distances, void* d_ws, size_t ws_bytes, void* stream);
int sg_knn_general(const float* d_x, int B, int C, int n, int k, int64_t* d_idx, void* stream);

/* backward of sg_group_max_rows (aggregate_cluster_feature, model.py:278-288): the gradient of a group's maximum goes to its
 * first maximal row (torch.max), every other row of the group gets 0; d_grows rows of `grow_stride` floats */
int sg_group_max_rows_backward(const float* d_rows, int row_stride, int D, const int32_t* d_goff, const int32_t* d_gidx, int G,
                               const float* d_gout, int out_stride, float* d_grows, int grow_stride, void* stream);
/* backward of sg_segment_max (model.py:793,834): rows [N,D] in member order, cluster c = rows [d_cl_off[c], d_cl_off[c+1]); the
 * gradient of a cluster's maximum goes to its first maximal row.  d_ws needs sg_segment_max_backward_ws_bytes(C, D). */
size_t sg_segment_max_backward_ws_bytes(int C, int D);
int sg_segment_max_backward(const float* d_rows, int N, int D, const int32_t* d_cl_off, int C, const float* d_gout, int out_stride,
                            float* d_grows, void* d_ws, size_t ws_bytes, void* stream);
/* backward of sg_gcn_forward, INCLUDING the path through the similarity weights exp(-alpha ||x_a - x_b + 1e-6||) and their
 * row normalisation (autograd differentiates them in the reference: model.py:262-265,305-309): d_gx [S,D], d_gw [D,D] */
size_t sg_gcn_backward_ws_bytes(int S, int D, int E);
int sg_gcn_backward(const float* d_x, int S, int D, const int32_t* d_adj, int E, const int32_t* d_rowptr, const int32_t* d_col,
                    const int32_t* d_eid, const float* d_w, float alpha, const float* d_gout, float* d_gx, float* d_gw,
                    void* d_ws, size_t ws_bytes, void* stream);

/* backward of sg_mlp1_forward w.r.t. its parameters (MLP1, model.py:39-80; batch-statistics BatchNorm2d over all C*64*10 rows):
 *   d_gfeat rows of g_stride floats: gradient w.r.t. Feat_1 [C,128] ([max | mean] over the 64 samples)
 *   d_gw [64,6], d_gg [64], d_gb [64];  d_bn_stats [128] or NULL = batch mean | biased variance of the conv output
 * d_ws needs sg_mlp1_backward_ws_bytes(C). */
size_t sg_mlp1_backward_ws_bytes(int C);
int sg_mlp1_backward(const float* d_samples, int C, const float* d_w, const float* d_gamma, const float* d_beta, const float* d_gfeat,
                     int g_stride, float* d_gw, float* d_gg, float* d_gb, float* d_bn_stats, void* d_ws, size_t ws_bytes, void* stream);

/* backward of sg_edgeconv_forward (get_graph_feature2 + MLP2 / MLP3, model.py:83-138; autograd differentiates it in the reference's
 * training step, train.py:167).  The inputs carry no gradient: parameter gradients only.
 *   d_x9m [N,12], d_knn [N,k]   as for the forward call
 *   d_gout [N,64]               gradient w.r.t. the op's OUTPUT (the post-activation max over k), e.g. from sg_segment_max_backward
 *   d_gw1 [64,18] d_gg1 d_gb1 [64]; layers == 2: d_gw2 [64,64] d_gg2 d_gb2 [64]
 *   d_bn2_in [128] or NULL      layers == 2: batch mean | biased variance of the SECOND BatchNorm's input as the forward computed
 *                               them (sg_trainer's tape); given, the dense forward pass that would recompute them is skipped
 *   d_bn_stats [256] or NULL    batch mean1 | biased var1 | mean2 | var2 (for the running-statistics update, momentum 0.1)
 * BatchNorm2d is differentiated WITH its batch statistics over all N*k rows (training mode).  Deterministic (ordered fp64
 * reductions).  d_ws needs sg_edgeconv_backward_ws_bytes(N). */
size_t sg_edgeconv_backward_ws_bytes(int N);
int sg_edgeconv_backward(const float* d_x9m, const int32_t* d_knn, int N, int k, int layers, const float* d_w1, const float* d_g1,
                         const float* d_b1, const float* d_w2, const float* d_g2, const float* d_b2, const float* d_gout, float* d_gw1,
                         float* d_gg1, float* d_gb1, float* d_gw2, float* d_gg2, float* d_gb2, const float* d_bn2_in, float* d_bn_stats,
                         void* d_ws, size_t ws_bytes, void* stream);

/* cross_entropy_loss (util.py:12-29) on its own: sum over the K rows of -sum_c t_c log softmax(logits)_c with the smoothed target
 * (0.8 on the gold class, 0.2 / (C - 1) elsewhere) or, smoothing == 0, the one-hot target.  d_prob [K,C] receives the softmax
 * (what the backward needs); backward: d_glogits = scale * (softmax - target). */
int sg_cross_entropy_forward(const float* d_logits, int K, int C, const int32_t* d_gold, int smoothing, float* d_prob, float* d_loss, void* stream);
int sg_cross_entropy_backward(const float* d_prob, int K, int C, const int32_t* d_gold, int smoothing, float scale, float* d_glogits, void* stream);

/* batch mean | biased variance [128 | 128] of classifier.bn1 as left in d_ws by sg_train_tail_forward (running-statistics update) */
int sg_train_tail_bn_stats(void* d_ws, size_t ws_bytes, int K, float* d_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One training step (train.py:160-168 around SegModel.forward in train mode).  Parameters and gradients are flat DEVICE
 * vectors of SG_NUM_PARAMS floats in the reference's `named_parameters()` order (sg_param_slot), owned by the caller: the
 * host side all-reduces the gradient vector over RCCL (train.py:88 DistributedDataParallel averages it over the ranks) and
 * applies the optimizer to the same vectors.
 * ------------------------------------------------------------------------------------------- */
#define SG_NUM_PARAM_TENSORS 19
#define SG_NUM_PARAMS 147880
#define SG_NUM_BN_STATS 768
typedef struct sg_trainer sg_trainer;
int sg_param_slot(int index, const char** name, int* offset, int* count);
sg_trainer* sg_trainer_create(int max_points, int max_segments, int max_edges, int max_vertices, float* d_params, float* d_grads, void* stream);
void sg_trainer_destroy(sg_trainer* tr);
size_t sg_trainer_device_bytes(const sg_trainer* tr);
/* SegModel.forward in train mode up to the classifier (model.py:684-914) with the CURRENT parameter vector: pseudo labels and
 * metrics as in ins_infer (`out`), the record the backward needs, *final_clusters = rows of Feat_5, *instances = K rows of Feat_6 */
int sg_trainer_forward(sg_trainer* tr, const sg_scene* scene, sg_result* out, int* final_clusters, int* instances);
/* Classifier + label-smoothed cross entropy (model.py:916-930) of the last sg_trainer_forward: h_loss[2] = {loss_sum, K} (the
 * reference's `loss [1,2]`); d_keep: DEVICE dropout keep mask [K,128] already scaled by 1 / (1 - p), or NULL (no dropout);
 * d_logits_out [K,40] may be NULL.  K < 2: SG_EUNSUP (BatchNorm1d in training mode raises in the reference). */
int sg_trainer_loss(sg_trainer* tr, const float* d_keep, float* h_loss, float* d_logits_out);
/* loss.backward() for loss = scale * loss_sum (scale <= 0: 1 / K, train.py:164-166), after sg_trainer_loss with the same mask:
 * fills the whole gradient vector. */
int sg_trainer_backward(sg_trainer* tr, const float* d_keep, float scale);
/* HOST h_out[SG_NUM_BN_STATS]: batch mean | biased variance of mlp_1.bn1 [64|64], mlp_2.bn1, mlp_3.bn1, mlp_3.bn2, classifier.bn1
 * [128|128] of the last step (after sg_trainer_backward); h_rows[5] = rows each statistic ran over (running_var is unbiased) */
int sg_trainer_bn_stats(const sg_trainer* tr, float* h_out, double* h_rows);

/* torch.optim.SGD (train.py:96: lr * 100, momentum, weight_decay 1e-4; dampening 0, no Nesterov) on flat vectors */
int sg_optimizer_sgd(float* d_params, const float* d_grads, float* d_momentum_buf, int n, float lr, float momentum, float weight_decay,
                     int first_step, void* stream);
/* torch.optim.Adam (train.py:98: betas 0.9 / 0.999, eps 1e-8, weight_decay added to the gradient); step counts from 1 */
int sg_optimizer_adam(float* d_params, const float* d_grads, float* d_m, float* d_v, int n, float lr, float weight_decay, int step, void* stream);

/* =============================================================================================
 * Readers for the reference's on-disk inputs.  Host only.
 * ============================================================================================= */

/* `<scene>.seg.json` (model.py:713-714; written by dataset/scannet/util.py:205-220): one JSON list per sampled point,
 * non-empty iff the point is the first member of an over-segment.  Fills h_seg_of_point[N] with segment numbers (rank
 * of the segment's first point) and returns the number of segments, or a negative error (malformed JSON, a list that
 * does not start at its own index, a member out of range / claimed twice, an uncovered point). */
int sg_parse_seg_json(const char* path, int N, int32_t* h_seg_of_point);

/* Segment number per point (ranks of the segments' first points) -> CSR of the over-segmentation (h_seg_points [N] ascending
 * inside every segment, h_seg_off [S+1]) + h_seg_first [S], h_seg_size [S]: the host side of DisjointSet's initial state
 * (model.py:712-721) in one counting pass. */
int sg_stage_segments(const int32_t* h_seg_of_point, int N, int S, int32_t* h_seg_points, int32_t* h_seg_off,
                      int32_t* h_seg_first, int32_t* h_seg_size);

/* One scene pack (`<scene>.sgpack`, seggroup_amd/cache.py: magic | header | the staged arrays, 64-byte aligned) from the reference's six per-scene
 * files, natively: src_paths6 = {<s>.pcl.pth, <s>.unmap.pth, weak <s>.label.pth, <s>.seg.json, raw <s>.label.pth, <s>.adj.pth} (data.py:28-38,
 * model.py:696-724 read the same files).  The `.pth` containers are torch.save's STORED zip + a protocol-2 pickle of one tensor; anything
 * else (compressed members, other pickles, float index tensors, names that need JSON escaping) is SG_EINVAL and the caller builds that pack in
 * Python.  Written atomically (temporary file + rename); byte-identical to cache.write_pack's output.
 * sg_pack_build_many: n scenes (src_paths = n x 6 paths) on `threads` plain threads; h_status[i] = SG_OK or the scene's error; returns the number built. */
int sg_pack_build(const char* const* src_paths6, const char* name, const char* out_path);
int sg_pack_build_many(const char* const* src_paths, const char* const* names, const char* const* out_paths, int n, int threads,
                       int32_t* h_status);

#ifdef __cplusplus
}
#endif
#endif /* SEGGROUP_HIP_H */
