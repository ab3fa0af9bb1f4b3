// Host-side grouping engine: the serial, order-dependent core of SegGroup restated over
// SEGMENT-level arrays (reference seggroup/model.py:169-258, 291-302, 439-509, 525-605).
//
// The reference keys its DisjointSet by point index and moves whole point lists on every union
// (model.py:186-192).  Points never leave their original over-segment, so a cluster is fully
// described by an ORDERED list of original segments; this engine keeps exactly that (S <= a few
// thousand entries) and the device expands it to point level with sg_gather_members().
// Segment numbers ascend with the index of the segment's first point, so "ascending root point id"
// (get_cluster_list, model.py:209-214) is "ascending root segment number" here.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

#include "sg_common.h"

struct sg_partition {
    int S = 0;
    std::vector<int32_t> seg_first, seg_size;
    std::vector<int32_t> owner;               // find(): root segment of every segment
    std::vector<int64_t> ins, sem;            // per root segment (stale on dead roots, like the reference)
    std::vector<double> npts;                 // point_num (float64 in the reference, model.py:176)
    // ordered member segments of each live root as singly linked lists over the segment ids (head[r] < 0: dead root).  Round 4: S small
    // vectors cost a scene 1,500 allocations when the partition is built and as many frees when it goes, plus a copy per union
    // (0.1 ms of a 1.9 ms single-scene forward, 0.36 ms at 5,000 segments); appending list a to list b is two stores here
    std::vector<int32_t> head, tail, next;

    bool live(int r) const { return head[r] >= 0; }

    bool unite(int a, int b) {                // DisjointSet.union(id1=a, id2=b), model.py:181-192
        if (a == b) return false;
        const int64_t ia = ins[a], ib = ins[b];
        if (ia != -1 && ib != -1 && ia != ib) return false;      // label veto (184-185)
        const bool moved = head[a] >= 0;
        for (int32_t s = head[a]; s >= 0; s = next[s]) owner[s] = b;   // 186
        npts[b] += npts[a];                                      // 187 (also for stale dead roots)
        if (ia != ib) {                                          // 188-190
            ins[b] = -ia * ib;
            sem[b] = -sem[a] * sem[b];
        }
        if (moved) {                                             // 191-192: b's members first, then a's, both in their own order
            if (head[b] < 0) head[b] = head[a];
            else next[tail[b]] = head[a];
            tail[b] = tail[a];
            head[a] = -1;
        }
        return moved;
    }

    // current numbering: cluster number of every ROOT segment (-1 for dead roots); returns count
    int numbering(std::vector<int32_t>& num_of_root) const {
        num_of_root.assign(S, -1);
        int c = 0;
        for (int r = 0; r < S; ++r)
            if (head[r] >= 0) num_of_root[r] = c++;
        return c;
    }
};

extern "C" {

sg_partition* sg_partition_create(int S, const int32_t* h_seg_first, const int32_t* h_seg_size,
                                  const int32_t* h_seg_ins, const int32_t* h_seg_sem) {
    if (S < 0 || (S > 0 && (!h_seg_first || !h_seg_size || !h_seg_ins || !h_seg_sem))) {
        sg::fail(SG_EINVAL, "sg_partition_create: null input");
        return nullptr;
    }
    for (int s = 1; s < S; ++s)
        if (h_seg_first[s] <= h_seg_first[s - 1]) {
            sg::fail(SG_EINVAL, "sg_partition_create: seg_first must be strictly ascending (segment %d)", s);
            return nullptr;
        }
    auto* p = new sg_partition();
    p->S = S;
    p->seg_first.assign(h_seg_first, h_seg_first + S);
    p->seg_size.assign(h_seg_size, h_seg_size + S);
    p->owner.resize(S);
    std::iota(p->owner.begin(), p->owner.end(), 0);
    p->ins.assign(h_seg_ins, h_seg_ins + S);
    p->sem.assign(h_seg_sem, h_seg_sem + S);
    p->npts.resize(S);
    p->head.resize(S); p->tail.resize(S); p->next.assign(S, -1);
    for (int s = 0; s < S; ++s) {
        p->npts[s] = (double)h_seg_size[s];
        p->head[s] = p->tail[s] = s;
    }
    return p;
}

void sg_partition_destroy(sg_partition* p) { delete p; }

int sg_partition_num_clusters(const sg_partition* p) {
    if (!p) return sg::fail(SG_EINVAL, "null partition");
    int c = 0;
    for (int r = 0; r < p->S; ++r) c += p->head[r] >= 0;
    return c;
}

int sg_partition_union(sg_partition* p, int a, int b) {
    if (!p || a < 0 || b < 0 || a >= p->S || b >= p->S) return sg::fail(SG_EINVAL, "sg_partition_union: bad root");
    return p->unite(a, b) ? 1 : 0;
}

int sg_partition_find(const sg_partition* p, int s) {
    if (!p || s < 0 || s >= p->S) return sg::fail(SG_EINVAL, "sg_partition_find: bad segment");
    return p->owner[s];
}

int sg_partition_label(const sg_partition* p, int r, int32_t* ins, int32_t* sem, double* npts) {
    if (!p || r < 0 || r >= p->S) return sg::fail(SG_EINVAL, "sg_partition_label: bad root");
    if (ins) *ins = (int32_t)p->ins[r];
    if (sem) *sem = (int32_t)p->sem[r];
    if (npts) *npts = p->npts[r];
    return SG_OK;
}

int sg_partition_layer(const sg_partition* p, int32_t* h_root, int32_t* h_cl_of_seg, int32_t* h_order,
                       int32_t* h_cl_seg_off, int32_t* h_cl_pt_off, int32_t* h_dst) {
    if (!p) return sg::fail(SG_EINVAL, "null partition");
    int c = 0, so = 0, po = 0;
    for (int r = 0; r < p->S; ++r) {
        if (p->head[r] < 0) continue;
        if (h_root) h_root[c] = r;
        if (h_cl_seg_off) h_cl_seg_off[c] = so;
        if (h_cl_pt_off) h_cl_pt_off[c] = po;
        for (int32_t s = p->head[r]; s >= 0; s = p->next[s]) {
            if (h_cl_of_seg) h_cl_of_seg[s] = c;
            if (h_order) h_order[so] = s;
            if (h_dst) h_dst[so] = po;
            ++so;
            po += p->seg_size[s];
        }
        ++c;
    }
    if (h_cl_seg_off) h_cl_seg_off[c] = so;
    if (h_cl_pt_off) h_cl_pt_off[c] = po;
    return c;
}

int sg_partition_group_nearby(sg_partition* p, const int32_t* h_root, int C, const float* h_dist,
                              const int32_t* h_adj, int E, float th, uint8_t* h_connected) {
    if (!p || (E > 0 && (!h_root || !h_dist || !h_adj))) return sg::fail(SG_EINVAL, "sg_partition_group_nearby: null input");
    for (int e = 0; e < 2 * E; ++e)
        if (h_adj[e] < 0 || h_adj[e] >= C) return sg::fail(SG_EINVAL, "sg_partition_group_nearby: edge endpoint out of range");
    // pass 1 (model.py:219-226): thresholded unions in edge order; `Dist > th` skips, NaN merges
    for (int e = 0; e < E; ++e) {
        if (h_dist[e] > th) continue;
        p->unite(p->owner[h_root[h_adj[2 * e]]], p->owner[h_root[h_adj[2 * e + 1]]]);
    }
    // pass 2 (model.py:228-239): absorb clusters with < 5 points until a sweep sees none
    // A cluster's point count only grows (union adds, and find() leads from a dead root to the grown one), so an edge that finds both of its
    // clusters at >= 5 points never fires again: after the first full sweep only the edges that fired are walked, in their original order --
    // the same unions in the same order as the reference's full sweeps (a 10k-edge sweep per round was most of this function).
    int rc = SG_OK;
    thread_local std::vector<int32_t> live;
    live.clear();
    bool first = true;
    for (;;) {
        bool small = false, moved = false;
        size_t kept = 0;
        const size_t n = first ? (size_t)E : live.size();
        for (size_t i = 0; i < n; ++i) {
            const int e = first ? (int)i : live[i];
            const int a = p->owner[h_root[h_adj[2 * e]]], b = p->owner[h_root[h_adj[2 * e + 1]]];
            if (p->npts[a] < 5 || p->npts[b] < 5) {
                moved |= p->unite(a, b);
                small = true;
                if (first) live.push_back(e); else live[kept] = e;
                ++kept;
            }
        }
        if (!first) live.resize(kept);
        first = false;
        if (!small) break;
        if (!moved) { rc = SG_ESTALL; break; }   // the reference never terminates here
    }
    // pass 3 (model.py:241-258)
    if (h_connected)
        for (int e = 0; e < E; ++e)
            h_connected[e] = p->owner[h_root[h_adj[2 * e]]] == p->owner[h_root[h_adj[2 * e + 1]]];
    if (rc == SG_ESTALL) sg::fail(SG_ESTALL, "group_nearby: a <5-point cluster cannot merge (reference would loop forever)");
    return rc;
}

int sg_partition_contract(const sg_partition* p, const int32_t* h_root_old, const int32_t* h_adj, int E,
                          const uint8_t* h_keep, int32_t* h_adj_out) {
    if (!p || (E > 0 && (!h_root_old || !h_adj || !h_adj_out))) return sg::fail(SG_EINVAL, "sg_partition_contract: null input");
    // update_adj (model.py:291-302): map both endpoints to the current numbering, drop loops, order each pair, then
    // unique rows in lexicographic order.  Endpoints are < C <= S, so two stable counting passes (by b, then by a) sort
    // the pairs in O(E + C) -- a comparison sort of the ~10k first-layer edges was the largest host cost of a forward.
    thread_local std::vector<int32_t> num, pa, pb, qa, qb, cnt;
    const int C = p->numbering(num);
    pa.clear(); pb.clear();
    for (int e = 0; e < E; ++e) {
        if (h_keep && !h_keep[e]) continue;
        int32_t a = num[p->owner[h_root_old[h_adj[2 * e]]]], b = num[p->owner[h_root_old[h_adj[2 * e + 1]]]];
        if (a == b) continue;
        if (a > b) std::swap(a, b);
        pa.push_back(a); pb.push_back(b);
    }
    const size_t n = pa.size();
    qa.resize(n); qb.resize(n);
    auto pass = [&](const std::vector<int32_t>& key, const std::vector<int32_t>& ia, const std::vector<int32_t>& ib,
                    std::vector<int32_t>& oa, std::vector<int32_t>& ob) {
        cnt.assign((size_t)C + 1, 0);
        for (size_t i = 0; i < n; ++i) ++cnt[key[i] + 1];
        for (int c = 0; c < C; ++c) cnt[c + 1] += cnt[c];
        for (size_t i = 0; i < n; ++i) {
            const int at = cnt[key[i]]++;
            oa[at] = ia[i]; ob[at] = ib[i];
        }
    };
    pass(pb, pa, pb, qa, qb);          // by second endpoint
    pass(qa, qa, qb, pa, pb);          // stable by first endpoint: lexicographic
    int out = 0;
    for (size_t i = 0; i < n; ++i) {
        if (i && pa[i] == pa[i - 1] && pb[i] == pb[i - 1]) continue;
        h_adj_out[2 * out] = pa[i];
        h_adj_out[2 * out + 1] = pb[i];
        ++out;
    }
    return out;
}

static float pair_distance(const float* a, const float* b, int D) {   // calculate_distance, model.py:269-274
    double acc = 0.0;
    for (int k = 0; k < D; ++k) {
        const double d = (double)a[k] - (double)b[k] + 1e-6;
        acc += d * d;
    }
    return (float)std::sqrt(acc);
}

// max-aggregate rows into the current numbering (aggregate_cluster_feature, model.py:278-288) and
// contract the edges; old numbering = root_old[0..C)
static void renumber(const sg_partition* p, std::vector<int32_t>& root, std::vector<float>& feat, int D,
                     std::vector<int32_t>& adj) {
    std::vector<int32_t> num;
    const int Cn = p->numbering(num);
    const int Co = (int)root.size();
    std::vector<float> nf((size_t)Cn * D, -INFINITY);
    for (int j = 0; j < Co; ++j) {
        const int c = num[p->owner[root[j]]];
        float* dst = &nf[(size_t)c * D];
        const float* src = &feat[(size_t)j * D];
        for (int k = 0; k < D; ++k) dst[k] = std::max(dst[k], src[k]);
    }
    std::vector<int32_t> nadj(adj.size());
    const int En = sg_partition_contract(p, root.data(), adj.data(), (int)adj.size() / 2, nullptr, nadj.data());
    nadj.resize((size_t)2 * std::max(En, 0));
    std::vector<int32_t> nroot(Cn);
    for (int r = 0; r < p->S; ++r)
        if (num[r] >= 0) nroot[num[r]] = r;
    root.swap(nroot);
    feat.swap(nf);
    adj.swap(nadj);
}

int sg_partition_group_unlabeled(sg_partition* p, int32_t* h_root_io, int* C_io, float* h_feat, int D,
                                 int32_t* h_adj, int* E_io) {
    if (!p || !h_root_io || !C_io || !h_feat || !E_io) return sg::fail(SG_EINVAL, "sg_partition_group_unlabeled: null input");
    std::vector<int32_t> root(h_root_io, h_root_io + *C_io);
    std::vector<float> feat(h_feat, h_feat + (size_t)*C_io * D);
    std::vector<int32_t> adj(h_adj, h_adj + (size_t)2 * *E_io);
    int count_old = *C_io;
    for (;;) {                                                    // model.py:447-477
        const int C = (int)root.size(), E = (int)adj.size() / 2;
        std::vector<float> M((size_t)C * C, 1000.0f);             // build_distance_matrix, model.py:312-316
        for (int e = 0; e < E; ++e) {
            const int a = adj[2 * e], b = adj[2 * e + 1];
            const float d = pair_distance(&feat[(size_t)a * D], &feat[(size_t)b * D], D);
            M[(size_t)a * C + b] = d;
            M[(size_t)b * C + a] = d;
        }
        std::vector<int32_t> nearest(C, 0);
        for (int i = 0; i < C; ++i) {                             // torch.min(dim=-1): first index on ties
            const float* row = &M[(size_t)i * C];
            int best = 0;
            for (int j = 1; j < C; ++j)
                if (row[j] < row[best]) best = j;
            nearest[i] = best;
        }
        for (int i = 0; i < C; ++i) {                             // model.py:453-458
            const int c1 = p->owner[root[i]];
            if (p->ins[c1] != -1) continue;
            p->unite(c1, p->owner[root[nearest[i]]]);
        }
        renumber(p, root, feat, D, adj);                          // model.py:460-473
        if ((int)root.size() == count_old) break;                 // model.py:474-477
        count_old = (int)root.size();
    }
    *C_io = (int)root.size();
    *E_io = (int)adj.size() / 2;
    std::copy(root.begin(), root.end(), h_root_io);
    std::copy(feat.begin(), feat.end(), h_feat);
    std::copy(adj.begin(), adj.end(), h_adj);
    for (int32_t r : root)
        if (p->ins[p->owner[r]] == -1) return 1;
    return 0;
}

int sg_partition_unlabeled_fallback(sg_partition* p, const int32_t* h_root, int C, const float* h_samples, int P) {
    if (!p || !h_root || !h_samples || P <= 0) return sg::fail(SG_EINVAL, "sg_partition_unlabeled_fallback: null input");
    std::vector<float> dmin(C);
    std::vector<int32_t> order(C);
    for (int i = 0; i < C; ++i) {                                 // model.py:481-494
        const int c1 = p->owner[h_root[i]];
        if (p->ins[p->owner[c1]] != -1) continue;
        double m64[3] = {0, 0, 0};
        const float* si = h_samples + (size_t)i * P * 3;
        for (int k = 0; k < P; ++k)
            for (int d = 0; d < 3; ++d) m64[d] += si[3 * k + d];
        const float m[3] = {(float)(m64[0] / P), (float)(m64[1] / P), (float)(m64[2] / P)};
        for (int j = 0; j < C; ++j) {                             // l2_norm + min over samples (fp32, 319-326)
            const float* sj = h_samples + (size_t)j * P * 3;
            float best = INFINITY;
            for (int k = 0; k < P; ++k) {
                const float dx = m[0] - sj[3 * k], dy = m[1] - sj[3 * k + 1], dz = m[2] - sj[3 * k + 2];
                volatile float xx = dx * dx, yy = dy * dy, zz = dz * dz;   // individually rounded squares
                volatile float s = xx + yy;
                const float v = s + zz;
                best = std::min(best, v);
            }
            dmin[j] = best;
        }
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return dmin[a] < dmin[b]; });
        for (int j : order) {
            if (j == i) continue;
            const int c2 = p->owner[h_root[j]];
            if (p->ins[p->owner[c2]] == -1) continue;
            p->unite(c1, c2);                                     // later calls are the reference's stale no-ops
        }
    }
    return SG_OK;
}

int sg_partition_export_tables(const sg_partition* p, int32_t* h_seg_tab, int32_t* h_ins_tab, int32_t* h_sem_tab) {
    if (!p) return sg::fail(SG_EINVAL, "null partition");
    for (int s = 0; s < p->S; ++s) {
        const int r = p->owner[s];
        if (h_seg_tab) h_seg_tab[s] = p->seg_first[r];                                   // model.py:530-531
        if (h_ins_tab) h_ins_tab[s] = p->ins[r] != -1 ? (int32_t)(p->ins[r] + 1) : -1;   // model.py:557-559
        if (h_sem_tab) h_sem_tab[s] = p->sem[r] != -1 ? (int32_t)(p->sem[r] + 1) : -1;   // model.py:585-587
    }
    return SG_OK;
}

}  // extern "C"
