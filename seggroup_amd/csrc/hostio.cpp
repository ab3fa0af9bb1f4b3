// Host-side readers for the reference's on-disk scene formats: the parts of input staging that
// were GIL-bound Python (`json.load` of a 150k-entry list of lists, `torch.load` of zip-wrapped tensors).  Host only;
// also built under ASan/UBSan (`make asan`).
#include <cerrno>
#include <string>

#include "sg_common.h"

namespace {

// whole file -> memory (seg.json is ~1.5 MB, a .pth ~4 MB)
int slurp(const char* path, std::string& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return sg::fail(SG_EINVAL, "cannot open %s: %s", path, strerror(errno));
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return sg::fail(SG_EINVAL, "cannot seek %s", path); }
    const long n = ftell(f);
    if (n < 0) { fclose(f); return sg::fail(SG_EINVAL, "cannot size %s", path); }
    rewind(f);
    out.resize((size_t)n);
    const size_t got = n ? fread(&out[0], 1, (size_t)n, f) : 0;
    fclose(f);
    if (got != (size_t)n) return sg::fail(SG_EINVAL, "short read of %s", path);
    return SG_OK;
}

inline bool is_ws(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; }

}  // namespace

extern "C" {

// `<scene>.seg.json` (written by util.py:205-220, read by model.py:713-714): a JSON list with one list per sampled
// point; list i is non-empty iff point i is the FIRST member of an over-segment and then holds that segment's members.
// Fills h_seg_of_point[N] with segment numbers (rank of the segment's first point, ascending) and returns the number
// of segments, or a negative error: malformed JSON, a list that does not start at its own index (model.py:715-720
// would mis-key its DisjointSet), a member out of range or claimed twice, or a point no list covers (the reference
// raises KeyError in update_adj then).  Same checks as seggroup_amd.scene.seg_from_lists, ~100x faster than json.load.
int sg_parse_seg_json(const char* path, int N, int32_t* h_seg_of_point) {
    if (!path || N <= 0 || !h_seg_of_point) return sg::fail(SG_EINVAL, "sg_parse_seg_json: bad arguments");
    static thread_local std::string buf;                      // the calling thread's own, kept between files (the pack builder parses thousands, csrc/packbuild.cpp)
    const int rc = slurp(path, buf);
    if (rc < 0) return rc;
    const char* p = buf.data();
    const char* const end = p + buf.size();
    auto skip = [&]() { while (p < end && is_ws(*p)) ++p; };
    for (int i = 0; i < N; ++i) h_seg_of_point[i] = -1;
    skip();
    if (p >= end || *p != '[') return sg::fail(SG_EINVAL, "sg_parse_seg_json: %s does not start with '['", path);
    ++p;
    int index = 0, S = 0;
    long long covered = 0;
    skip();
    if (p < end && *p == ']') { ++p; index = 0; }
    else {
        for (;;) {
            skip();
            if (p >= end || *p != '[') return sg::fail(SG_EINVAL, "sg_parse_seg_json: %s: expected '[' for list %d", path, index);
            ++p;
            skip();
            int count = 0;
            if (p < end && *p == ']') ++p;
            else {
                for (;;) {
                    skip();
                    bool neg = false;
                    if (p < end && *p == '-') { neg = true; ++p; }
                    if (p >= end || *p < '0' || *p > '9') return sg::fail(SG_EINVAL, "sg_parse_seg_json: %s: expected an integer in list %d", path, index);
                    long long v = 0;
                    while (p < end && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); if (v > 0x7fffffffLL) break; ++p; }
                    if (neg) v = -v;
                    if (v < 0 || v >= N) return sg::fail(SG_EINVAL, "sg_parse_seg_json: %s: member %lld of list %d is outside [0, %d)", path, v, index, N);
                    if (count == 0 && v != index)
                        return sg::fail(SG_EINVAL, "seg.json list %d does not start at its own index (got %lld)", index, v);
                    if (index >= N) return sg::fail(SG_EINVAL, "sg_parse_seg_json: %s: more than %d lists", path, N);
                    if (h_seg_of_point[v] >= 0) return sg::fail(SG_EINVAL, "sg_parse_seg_json: %s: point %lld is a member of two lists", path, v);
                    h_seg_of_point[v] = S;
                    ++count;
                    ++covered;
                    skip();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == ']') { ++p; break; }
                    return sg::fail(SG_EINVAL, "sg_parse_seg_json: %s: expected ',' or ']' in list %d", path, index);
                }
            }
            if (count) ++S;
            ++index;
            skip();
            if (p < end && *p == ',') { ++p; continue; }
            if (p < end && *p == ']') { ++p; break; }
            return sg::fail(SG_EINVAL, "sg_parse_seg_json: %s: expected ',' or ']' after list %d", path, index - 1);
        }
    }
    skip();
    if (p != end) return sg::fail(SG_EINVAL, "sg_parse_seg_json: %s: trailing bytes after the list", path);
    if (covered != N) return sg::fail(SG_EINVAL, "seg.json does not cover every point (%lld of %d; the reference raises KeyError in update_adj here)", covered, N);
    return S;
}


// Segment number per point -> the CSR of the over-segmentation (points ascending inside every segment) + per-segment first
// point and size: one counting pass instead of a stable argsort of N keys (the largest item of scene staging after the JSON
// parse).  Segment numbers must be the ranks of the segments' first points (what sg_parse_seg_json returns).
int sg_stage_segments(const int32_t* h_seg_of_point, int N, int S, int32_t* h_seg_points, int32_t* h_seg_off, int32_t* h_seg_first,
                      int32_t* h_seg_size) {
    if (!h_seg_of_point || N <= 0 || S <= 0 || !h_seg_points || !h_seg_off || !h_seg_first || !h_seg_size)
        return sg::fail(SG_EINVAL, "sg_stage_segments: bad arguments");
    for (int s = 0; s < S; ++s) { h_seg_size[s] = 0; h_seg_first[s] = -1; }
    for (int i = 0; i < N; ++i) {
        const int s = h_seg_of_point[i];
        if (s < 0 || s >= S) return sg::fail(SG_EINVAL, "sg_stage_segments: point %d has segment %d outside [0, %d)", i, s, S);
        if (h_seg_size[s]++ == 0) h_seg_first[s] = i;
    }
    h_seg_off[0] = 0;
    for (int s = 0; s < S; ++s) {
        if (h_seg_size[s] == 0) return sg::fail(SG_EINVAL, "sg_stage_segments: segment %d has no points", s);
        if (s && h_seg_first[s] <= h_seg_first[s - 1]) return sg::fail(SG_EINVAL, "segment numbers must ascend with each segment's first point");
        h_seg_off[s + 1] = h_seg_off[s] + h_seg_size[s];
    }
    std::vector<int32_t> cur(h_seg_off, h_seg_off + S);
    for (int i = 0; i < N; ++i) h_seg_points[cur[h_seg_of_point[i]]++] = i;
    return SG_OK;
}

}  // extern "C"
