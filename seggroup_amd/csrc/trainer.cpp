// One training step of the reference (train.py:160-168 around SegModel.forward in train mode, model.py:684-932) for one scene:
// forward through the pipeline with a tape, the classifier tail, and the backward chain
//
//   loss -> classifier tail -> Feat_5 -> (max-aggregations of the final clustering) -> GCN_3 -> [ Feat_3 | max over points of MLP3 ]
//        -> GCN_2 -> [ Feat_2 | max over points of MLP2 ] -> (max over merged segments) -> MLP1
//
// every link a HIP kernel (kernels_train.hip, kernels_train_edge.hip, kernels_mlp1.hip).  The discrete structure (FPS
// samples, kNN tables, clusterings, adjacency) is whatever the forward produced: like autograd in the reference, the
// backward treats it as constant.  Parameters and gradients are flat device vectors in the reference's
// `named_parameters()` order (sg_param_slot), owned by the caller -- the host side all-reduces the gradient vector over
// RCCL (train.py's DistributedDataParallel) and applies sg_optimizer_sgd / sg_optimizer_adam to the same vectors.
#include <memory>

#include "pipeline_priv.h"

namespace {

struct Slot { const char* name; int off, count; };
// the reference model's named_parameters() order (BatchNorm before its conv: model.py:42-44,88-90,121-126)
const Slot kSlots[SG_NUM_PARAM_TENSORS] = {
    {"mlp_1.bn1.weight", 0, 64}, {"mlp_1.bn1.bias", 64, 64}, {"mlp_1.conv1.0.weight", 128, 384},
    {"mlp_2.bn1.weight", 512, 64}, {"mlp_2.bn1.bias", 576, 64}, {"mlp_2.conv1.0.weight", 640, 1152},
    {"gcn_2.fc.weight", 1792, 36864},
    {"mlp_3.bn1.weight", 38656, 64}, {"mlp_3.bn1.bias", 38720, 64}, {"mlp_3.conv1.0.weight", 38784, 1152},
    {"mlp_3.bn2.weight", 39936, 64}, {"mlp_3.bn2.bias", 40000, 64}, {"mlp_3.conv2.0.weight", 40064, 4096},
    {"gcn_3.fc.weight", 44160, 65536},
    {"classifier.linear1.weight", 109696, 32768}, {"classifier.bn1.weight", 142464, 128}, {"classifier.bn1.bias", 142592, 128},
    {"classifier.linear2.weight", 142720, 5120}, {"classifier.linear2.bias", 147840, 40}};
enum { M1G, M1B, M1W, M2G, M2B, M2W, G2, M3G1, M3B1, M3W1, M3G2, M3B2, M3W2, G3, CW1, CG, CB, CW2, CB2 };

}  // namespace

struct sg_trainer {
    sg_pipeline* pl = nullptr;
    sg_tape tape;
    hipStream_t st = nullptr;
    float* params = nullptr;            // caller-owned, [SG_NUM_PARAMS]
    float* grads = nullptr;
    int maxN = 0, maxS = 0;
    sgp::DevBuf<float> feat5, g_feat5, gA, gB, gC, g_pf, bn, loss, logits;
    sgp::DevBuf<int32_t> group, gold, fin;
    sgp::DevBuf<char> ws_tail, ws_gcn, ws_edge, ws_mlp1, ws_seg;
    int C6 = 0, K = 0;
    bool have_forward = false, have_loss = false;
    const float* P(int slot) const { return params + kSlots[slot].off; }
    float* G(int slot) const { return grads + kSlots[slot].off; }
};

#define TR_CHECK(call)                 \
    do {                               \
        const int rc__ = (call);       \
        if (rc__ < 0) return rc__;     \
    } while (0)

extern "C" {

int sg_param_slot(int index, const char** name, int* offset, int* count) {
    if (index < 0 || index >= SG_NUM_PARAM_TENSORS) return sg::fail(SG_EINVAL, "sg_param_slot: index %d outside [0, %d)", index, SG_NUM_PARAM_TENSORS);
    if (name) *name = kSlots[index].name;
    if (offset) *offset = kSlots[index].off;
    if (count) *count = kSlots[index].count;
    return SG_OK;
}

void sg_trainer_destroy(sg_trainer* tr) {
    if (!tr) return;
    if (tr->pl) sg_pipeline_destroy(tr->pl);
    delete tr;
}

sg_trainer* sg_trainer_create(int maxN, int maxS, int maxE, int maxV, float* d_params, float* d_grads, void* stream) {
    if (!d_params || !d_grads) { sg::fail(SG_EINVAL, "sg_trainer_create: null parameter / gradient vector"); return nullptr; }
    std::unique_ptr<sg_trainer, void (*)(sg_trainer*)> tr(new sg_trainer(), sg_trainer_destroy);   // a failed allocation below also frees tr->pl
    // the pipeline wants host weights at creation; they are replaced from d_params before every forward
    std::vector<float> zero(65536, 0.f);
    sg_weights w;
    const float* z = zero.data();
    w.mlp1_w = z; w.mlp1_g = z; w.mlp1_b = z; w.mlp2_w = z; w.mlp2_g = z; w.mlp2_b = z; w.gcn2_w = z; w.mlp3_w1 = z; w.mlp3_g1 = z; w.mlp3_b1 = z;
    w.mlp3_w2 = z; w.mlp3_g2 = z; w.mlp3_b2 = z; w.gcn3_w = z;
    tr->pl = sg_pipeline_create(maxN, maxS, maxE, maxV, &w, stream);
    if (!tr->pl) return nullptr;
    sg_pipeline_set_timing(tr->pl, 0);                        // no per-stage events in the training loop
    tr->st = sg::as_stream(stream);
    tr->params = d_params; tr->grads = d_grads; tr->maxN = maxN; tr->maxS = maxS;
    int bad = 0;
    const size_t N = maxN, S = maxS;
    for (sg_tape::Layer& L : tr->tape.layer) {
        bad |= L.x9m.alloc(N * 12) | L.pf.alloc(N * 64) | L.knn.alloc(N * 20) | L.desc.alloc(tr->pl->desc.n) | L.cat.alloc(S * 256) | L.gcn.alloc(S * 256) | L.bn_last.alloc(128);
    }
    const size_t maxE1 = tr->pl->adj1.n / 2;
    bad |= tr->feat5.alloc(S * 256) | tr->g_feat5.alloc(S * 256) | tr->gA.alloc(S * 256) | tr->gB.alloc(S * 256) | tr->gC.alloc(S * 256) | tr->g_pf.alloc(N * 64);
    bad |= tr->bn.alloc(SG_NUM_BN_STATS) | tr->loss.alloc(2) | tr->logits.alloc(S * 40) | tr->group.alloc(S) | tr->gold.alloc(S) | tr->fin.alloc(2 * S + 2);
    bad |= tr->ws_tail.alloc(sg_train_tail_ws_bytes(maxS, maxS)) | tr->ws_gcn.alloc(sg_gcn_backward_ws_bytes(maxS, 256, (int)maxE1));
    bad |= tr->ws_seg.alloc(sg_segment_max_backward_ws_bytes(maxS, 64));
    bad |= tr->ws_edge.alloc(sg_edgeconv_backward_ws_bytes(maxN)) | tr->ws_mlp1.alloc(sg_mlp1_backward_ws_bytes(maxS));
    if (bad) { sg::fail(SG_ENOMEM, "sg_trainer_create: device allocation failed (N=%d S=%d)", maxN, maxS); return nullptr; }
    return tr.release();
}

/* SegModel.forward in train mode up to (not including) the classifier: the pseudo labels and metrics of the scene like an
 * ins_infer forward (model.py:684-897) with the CURRENT parameter vector, the tape for the backward, and the tail's inputs.
 * -> *final_clusters = rows of Feat_5, *instances = K (rows of Feat_6 = distinct weak instance labels among them). */
int sg_trainer_forward(sg_trainer* tr, const sg_scene* scene, sg_result* out, int* final_clusters, int* instances) {
    SG_REQUIRE(tr && scene && out, "sg_trainer_forward: null argument");
    sg_pipeline* pl = tr->pl;
    hipStream_t st = tr->st;
    tr->have_forward = false;
    tr->have_loss = false;
    SG_HIP(hipSetDevice(pl->device));
    // parameters -> the pipeline's weight block (+ the transposed GCN matrices its forward kernel reads)
    struct { int slot; size_t off; } map[] = {{M1W, pl->o_m1w}, {M1G, pl->o_m1g}, {M1B, pl->o_m1b}, {M2W, pl->o_m2w}, {M2G, pl->o_m2g}, {M2B, pl->o_m2b},
                                              {G2, pl->o_g2}, {M3W1, pl->o_m3w1}, {M3G1, pl->o_m3g1}, {M3B1, pl->o_m3b1}, {M3W2, pl->o_m3w2},
                                              {M3G2, pl->o_m3g2}, {M3B2, pl->o_m3b2}, {G3, pl->o_g3}};
    for (const auto& m : map)
        SG_HIP(hipMemcpyAsync(pl->w.p + m.off, tr->P(m.slot), (size_t)kSlots[m.slot].count * 4, hipMemcpyDeviceToDevice, st));
    TR_CHECK(sg::transpose_square(tr->P(G2), pl->w.p + pl->o_g2t, 192, st));
    TR_CHECK(sg::transpose_square(tr->P(G3), pl->w.p + pl->o_g3t, 256, st));
    sg_debug dbg;
    std::memset(&dbg, 0, sizeof dbg);
    dbg.tape = &tr->tape;
    TR_CHECK(sg_pipeline_forward(pl, scene, SG_MODE_INS_INFER, out, &dbg));
    sg_tape& T = tr->tape;
    if (!T.filled) return sg::fail(SG_EINVAL, "sg_trainer_forward: the forward left no tape");
    // model.py:900-914: instance slots = ranks among the sorted distinct weak instance labels (-1 counts), class of a slot =
    // the first cluster's
    const int C6 = T.C6;
    std::vector<int32_t> uniq(T.ins5);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    const int K = (int)uniq.size();
    std::vector<int32_t> group(C6), gold(K, -1);
    for (int c = 0; c < C6; ++c) {
        group[c] = (int)(std::lower_bound(uniq.begin(), uniq.end(), T.ins5[c]) - uniq.begin());
        if (gold[group[c]] < 0) {
            if (T.sem5[c] < 0 || T.sem5[c] >= 40)
                return sg::fail(SG_EINVAL, "weak semantic label %d outside 0..39 (cross_entropy_loss's scatter raises in the reference)", T.sem5[c]);
            gold[group[c]] = T.sem5[c];
        }
    }
    SG_HIP(hipMemcpyAsync(tr->feat5.p, T.feat5.data(), (size_t)C6 * 256 * 4, hipMemcpyHostToDevice, st));
    SG_HIP(hipMemcpyAsync(tr->group.p, group.data(), (size_t)C6 * 4, hipMemcpyHostToDevice, st));
    SG_HIP(hipMemcpyAsync(tr->gold.p, gold.data(), (size_t)K * 4, hipMemcpyHostToDevice, st));
    SG_HIP(hipMemcpyAsync(tr->fin.p, T.fin_goff.data(), (size_t)(C6 + 1) * 4, hipMemcpyHostToDevice, st));
    SG_HIP(hipMemcpyAsync(tr->fin.p + C6 + 1, T.fin_gidx.data(), T.fin_gidx.size() * 4, hipMemcpyHostToDevice, st));
    SG_HIP(hipStreamSynchronize(st));                      // the host vectors above go out of scope
    tr->C6 = C6; tr->K = K;
    tr->have_forward = true;
    if (final_clusters) *final_clusters = C6;
    if (instances) *instances = K;
    return SG_OK;
}

/* Classifier + label-smoothed cross entropy (model.py:916-930) of the scene of the last sg_trainer_forward: h_loss[2] =
 * {loss_sum, K} (the reference's `loss [1,2]`).  d_keep: DEVICE dropout keep mask [K,128], already scaled by 1 / (1 - p), or NULL. */
int sg_trainer_loss(sg_trainer* tr, const float* d_keep, float* h_loss, float* d_logits_out) {
    SG_REQUIRE(tr && tr->have_forward, "sg_trainer_loss: no forward");
    hipStream_t st = tr->st;
    SG_HIP(hipSetDevice(tr->pl->device));
    sg_classifier cls;
    cls.w1 = tr->P(CW1); cls.gamma = tr->P(CG); cls.beta = tr->P(CB); cls.w2 = tr->P(CW2); cls.b2 = tr->P(CB2);
    float* logits = d_logits_out ? d_logits_out : tr->logits.p;
    TR_CHECK(sg_train_tail_forward(tr->feat5.p, tr->C6, tr->group.p, tr->K, tr->gold.p, d_keep, &cls, logits, tr->loss.p, tr->ws_tail.p, tr->ws_tail.n, (void*)st));
    TR_CHECK(sg_train_tail_bn_stats(tr->ws_tail.p, tr->ws_tail.n, tr->K, tr->bn.p + 512, (void*)st));
    if (h_loss) {
        SG_HIP(hipMemcpyAsync(h_loss, tr->loss.p, 8, hipMemcpyDeviceToHost, st));
        SG_HIP(hipStreamSynchronize(st));
    }
    tr->have_loss = true;
    return SG_OK;
}

/* loss.backward() for loss = scale * loss_sum (scale <= 0: 1 / K, train.py:164-166) after sg_trainer_loss with the same mask:
 * fills the WHOLE gradient vector. */
int sg_trainer_backward(sg_trainer* tr, const float* d_keep, float scale) {
    SG_REQUIRE(tr && tr->have_forward && tr->have_loss, "sg_trainer_backward: sg_trainer_forward and sg_trainer_loss come first");
    sg_pipeline* pl = tr->pl;
    sg_tape& T = tr->tape;
    hipStream_t st = tr->st;
    void* sv = (void*)st;
    SG_HIP(hipSetDevice(pl->device));
    const int C6 = tr->C6, K = tr->K, N = T.N, S = T.S;
    if (scale <= 0.f) scale = 1.0f / (float)K;
    sg_classifier cls;
    cls.w1 = tr->P(CW1); cls.gamma = tr->P(CG); cls.beta = tr->P(CB); cls.w2 = tr->P(CW2); cls.b2 = tr->P(CB2);
    TR_CHECK(sg_train_tail_backward(C6, K, tr->gold.p, d_keep, &cls, scale, tr->G(CW1), tr->G(CG), tr->G(CB), tr->G(CW2), tr->G(CB2), tr->g_feat5.p,
                                    tr->ws_tail.p, tr->ws_tail.n, sv));
    // Feat_5 <- rows of the last GCN output
    const sg_tape::Layer& L1 = T.layer[1];
    const sg_tape::Layer& L0 = T.layer[0];
    TR_CHECK(sg_group_max_rows_backward(L1.gcn.p, 256, 256, tr->fin.p, tr->fin.p + C6 + 1, C6, tr->g_feat5.p, 256, tr->gA.p, 256, sv));
    // ---- semantic layer 2 (GCN_3, MLP3) ----
    {
        const int32_t* dd = L1.desc.p;
        const sgp::DescOffsets& o = L1.o;
        TR_CHECK(sg_gcn_backward(L1.cat.p, L1.C, 256, dd + o.adj, L1.E, dd + o.rowptr, dd + o.col, dd + o.eid, tr->P(G3), 0.125f, tr->gA.p, tr->gB.p, tr->G(G3),
                                 tr->ws_gcn.p, tr->ws_gcn.n, sv));
        TR_CHECK(sg_group_max_rows_backward(L0.gcn.p, 192, 192, dd + o.goff, dd + o.gidx, L1.C, tr->gB.p, 256, tr->gC.p, 192, sv));
        TR_CHECK(sg_segment_max_backward(L1.pf.p, N, 64, dd + o.cl_pt_off, L1.C, tr->gB.p + 192, 256, tr->g_pf.p, tr->ws_seg.p, tr->ws_seg.n, sv));
        TR_CHECK(sg_edgeconv_backward(L1.x9m.p, L1.knn.p, N, 20, 2, tr->P(M3W1), tr->P(M3G1), tr->P(M3B1), tr->P(M3W2), tr->P(M3G2), tr->P(M3B2), tr->g_pf.p,
                                      tr->G(M3W1), tr->G(M3G1), tr->G(M3B1), tr->G(M3W2), tr->G(M3G2), tr->G(M3B2), L1.bn_last.p, tr->bn.p + 256, tr->ws_edge.p, tr->ws_edge.n, sv));
    }
    // ---- semantic layer 1 (GCN_2, MLP2) ----
    {
        const int32_t* dd = L0.desc.p;
        const sgp::DescOffsets& o = L0.o;
        TR_CHECK(sg_gcn_backward(L0.cat.p, L0.C, 192, dd + o.adj, L0.E, dd + o.rowptr, dd + o.col, dd + o.eid, tr->P(G2), 0.125f, tr->gC.p, tr->gB.p, tr->G(G2),
                                 tr->ws_gcn.p, tr->ws_gcn.n, sv));
        TR_CHECK(sg_group_max_rows_backward(pl->feat1.p, 128, 128, dd + o.goff, dd + o.gidx, L0.C, tr->gB.p, 192, tr->gA.p, 128, sv));
        TR_CHECK(sg_segment_max_backward(L0.pf.p, N, 64, dd + o.cl_pt_off, L0.C, tr->gB.p + 128, 192, tr->g_pf.p, tr->ws_seg.p, tr->ws_seg.n, sv));
        TR_CHECK(sg_edgeconv_backward(L0.x9m.p, L0.knn.p, N, 20, 1, tr->P(M2W), tr->P(M2G), tr->P(M2B), nullptr, nullptr, nullptr, tr->g_pf.p, tr->G(M2W),
                                      tr->G(M2G), tr->G(M2B), nullptr, nullptr, nullptr, nullptr, tr->bn.p + 128, tr->ws_edge.p, tr->ws_edge.n, sv));
    }
    // ---- structural layer (MLP1) ----
    TR_CHECK(sg_mlp1_backward(pl->samples.p, S, tr->P(M1W), tr->P(M1G), tr->P(M1B), tr->gA.p, 128, tr->G(M1W), tr->G(M1G), tr->G(M1B), tr->bn.p, tr->ws_mlp1.p,
                              tr->ws_mlp1.n, sv));
    SG_HIP(hipStreamSynchronize(st));
    return SG_OK;
}

/* batch statistics of the five BatchNorm layers of the last step (for the running_mean / running_var update, momentum 0.1):
 * HOST h_out[SG_NUM_BN_STATS] = mlp_1.bn1 mean[64] var[64] | mlp_2.bn1 | mlp_3.bn1 | mlp_3.bn2 | classifier.bn1 mean[128] var[128];
 * biased variances, rows per layer in h_rows[5] (the unbiased running_var needs them) */
int sg_trainer_bn_stats(const sg_trainer* tr, float* h_out, double* h_rows) {
    SG_REQUIRE(tr && h_out && tr->have_loss, "sg_trainer_bn_stats: no step to report");
    if (h_rows) {
        h_rows[0] = (double)tr->tape.S * 640.0;
        h_rows[1] = h_rows[2] = h_rows[3] = (double)tr->tape.N * 20.0;
        h_rows[4] = (double)tr->K;
    }
    SG_HIP(hipMemcpyAsync(h_out, tr->bn.p, SG_NUM_BN_STATS * 4, hipMemcpyDeviceToHost, tr->st));
    SG_HIP(hipStreamSynchronize(tr->st));
    return SG_OK;
}

size_t sg_trainer_device_bytes(const sg_trainer* tr) {
    if (!tr) return 0;
    size_t b = sg_pipeline_device_bytes(tr->pl);
    for (const sg_tape::Layer& L : tr->tape.layer) b += (L.x9m.n + L.pf.n + L.cat.n + L.gcn.n) * 4 + (L.knn.n + L.desc.n) * 4;
    return b + (tr->feat5.n + tr->g_feat5.n + tr->gA.n + tr->gB.n + tr->gC.n + tr->g_pf.n) * 4 + tr->ws_tail.n + tr->ws_gcn.n + tr->ws_edge.n + tr->ws_mlp1.n;
}

}  // extern "C"
