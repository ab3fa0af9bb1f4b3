// Error plumbing, version / device probe, and the label-file writers (a16 file side,
// reference seggroup/model.py:536-547) of libseggroup_hip.so.
#include <cerrno>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>

#include "sg_common.h"

namespace sg {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace sg

extern "C" {

const char* sg_last_error(void) { return sg::err_buf(); }

int sg_version(void) { return 100; }   // 0.1.0

int sg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// '%d\n' per value (model.py:541-546 writes every label with '%d\n').  14 vectors x V values per scene make this the
// largest host cost of a scene once the GPU part takes < 1 ms, so: digits come two at a time from a 200-byte table, values
// below 100 (semantic / instance labels) take a branch without any division, and the output buffer is a reused
// thread-local allocation (a fresh std::string would zero-fill ~2 MB per vector first).
namespace {
const char kDigits2[201] =
    "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
    "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";

inline char* put_u32(char* o, uint32_t v) {
    if (v < 10) { *o++ = (char)('0' + v); return o; }
    if (v < 100) { o[0] = kDigits2[2 * v]; o[1] = kDigits2[2 * v + 1]; return o + 2; }
    char tmp[10];
    int n = 0;
    while (v >= 100) { const uint32_t q = v / 100, r = v - q * 100; tmp[n++] = kDigits2[2 * r + 1]; tmp[n++] = kDigits2[2 * r]; v = q; }
    if (v >= 10) { tmp[n++] = kDigits2[2 * v + 1]; tmp[n++] = kDigits2[2 * v]; }
    else tmp[n++] = (char)('0' + v);
    while (n) *o++ = tmp[--n];
    return o;
}

struct TlBuf {
    char* p = nullptr;
    size_t cap = 0;
    ~TlBuf() { free(p); }
    char* need(size_t n) {
        if (n > cap) { free(p); p = (char*)malloc(n); cap = p ? n : 0; }
        return p;
    }
};
}  // namespace

int sg_write_label_txt(const char* path, const int32_t* h_vec, int V) {
    if (!path || (V > 0 && !h_vec) || V < 0) return sg::fail(SG_EINVAL, "sg_write_label_txt: bad arguments");
    static thread_local TlBuf tl;
    char* const buf = tl.need((size_t)V * 12 + 16);
    if (!buf) return sg::fail(SG_ENOMEM, "sg_write_label_txt: out of memory");
    char* o = buf;
    for (int i = 0; i < V; ++i) {
        const int32_t v = h_vec[i];
        if (v < 0) { *o++ = '-'; o = put_u32(o, (uint32_t)(-(int64_t)v)); }
        else o = put_u32(o, (uint32_t)v);
        *o++ = '\n';
    }
    FILE* f = fopen(path, "wb");
    if (!f) return sg::fail(SG_EINVAL, "sg_write_label_txt: cannot open %s: %s", path, strerror(errno));
    const size_t len = (size_t)(o - buf);
    const size_t w = fwrite(buf, 1, len, f);
    if (fclose(f) != 0 || w != len) return sg::fail(SG_EINVAL, "sg_write_label_txt: short write to %s", path);
    return SG_OK;
}

// NumPy .npy v1.0, little-endian int32, C order, shape (V,)
int sg_write_label_npy(const char* path, const int32_t* h_vec, int V) {
    if (!path || (V > 0 && !h_vec) || V < 0) return sg::fail(SG_EINVAL, "sg_write_label_npy: bad arguments");
    char dict[128];
    int n = snprintf(dict, sizeof dict, "{'descr': '<i4', 'fortran_order': False, 'shape': (%d,), }", V);
    const int unpadded = 10 + n + 1;                        // magic(6)+ver(2)+len(2)+dict+'\n'
    const int pad = (64 - unpadded % 64) % 64;
    std::string hdr("\x93NUMPY\x01\x00", 8);
    const uint16_t hlen = (uint16_t)(n + pad + 1);
    hdr.push_back((char)(hlen & 0xff));
    hdr.push_back((char)(hlen >> 8));
    hdr.append(dict, n);
    hdr.append((size_t)pad, ' ');
    hdr.push_back('\n');
    FILE* f = fopen(path, "wb");
    if (!f) return sg::fail(SG_EINVAL, "sg_write_label_npy: cannot open %s: %s", path, strerror(errno));
    size_t w = fwrite(hdr.data(), 1, hdr.size(), f);
    w += fwrite(h_vec, 4, (size_t)V, f);
    if (fclose(f) != 0 || w != hdr.size() + (size_t)V) return sg::fail(SG_EINVAL, "sg_write_label_npy: short write to %s", path);
    return SG_OK;
}

// `.seg.json` (util.py:205-220): json.dump of a list with one entry per sampled point -- the ascending member list of a
// segment at the index of its smallest member, [] everywhere else -- byte for byte as Python writes it ("[[], [3, 7], []]").
// Groups come as a CSR (any group order); members ascending inside a group.
int sg_write_seg_json(const char* path, const int32_t* h_seg_points, const int32_t* h_seg_off, int G, int Np) {
    if (!path || G < 0 || Np < 0 || (G > 0 && (!h_seg_points || !h_seg_off))) return sg::fail(SG_EINVAL, "sg_write_seg_json: bad arguments");
    std::vector<int32_t> group_at(Np, -1);
    for (int g = 0; g < G; ++g) {
        const int first = h_seg_points[h_seg_off[g]];
        if (h_seg_off[g + 1] <= h_seg_off[g] || first < 0 || first >= Np) return sg::fail(SG_EINVAL, "sg_write_seg_json: bad group %d", g);
        group_at[first] = g;
    }
    std::string buf;
    buf.reserve((size_t)Np * 10 + 16);
    auto put_int = [&](int64_t v) {
        char tmp[12];
        int n = 0;
        do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (n) buf.push_back(tmp[--n]);
    };
    buf.push_back('[');
    for (int i = 0; i < Np; ++i) {
        if (i) buf.append(", ");
        buf.push_back('[');
        const int g = group_at[i];
        if (g >= 0)
            for (int k = h_seg_off[g]; k < h_seg_off[g + 1]; ++k) {
                if (k > h_seg_off[g]) buf.append(", ");
                put_int(h_seg_points[k]);
            }
        buf.push_back(']');
    }
    buf.push_back(']');
    FILE* f = fopen(path, "wb");
    if (!f) return sg::fail(SG_EINVAL, "sg_write_seg_json: cannot open %s: %s", path, strerror(errno));
    const size_t w = fwrite(buf.data(), 1, buf.size(), f);
    if (fclose(f) != 0 || w != buf.size()) return sg::fail(SG_EINVAL, "sg_write_seg_json: short write to %s", path);
    return SG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Asynchronous label writer: formatting 14 x V integers as text is ~8 MB per 150k scene, far
// slower than the forward itself, so the files are written by a small pool of native threads while the next
// scenes are already on the GPU.  submit() copies the vector, so the caller's (pinned) buffer is free at once.
// ---------------------------------------------------------------------------------------------------------
struct sg_writer {
    struct Job { std::string base; std::vector<int32_t> vec; int formats; };
    std::mutex mu;
    std::condition_variable cv_job, cv_idle;
    std::deque<Job> q;
    std::vector<std::thread> threads;
    size_t max_queue = 64;
    int busy = 0;
    bool stop = false;
    int first_err = 0;
    std::string err;

    void run() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                j = std::move(q.front());
                q.pop_front();
                ++busy;
            }
            cv_idle.notify_all();
            int rc = 0;
            if (j.formats & 1) rc = sg_write_label_txt((j.base + ".txt").c_str(), j.vec.data(), (int)j.vec.size());
            if (rc == 0 && (j.formats & 2)) rc = sg_write_label_npy((j.base + ".npy").c_str(), j.vec.data(), (int)j.vec.size());
            {
                std::lock_guard<std::mutex> lk(mu);
                if (rc < 0 && first_err == 0) { first_err = rc; err = sg_last_error(); }
                --busy;
            }
            cv_idle.notify_all();
        }
    }
};

extern "C" {

sg_writer* sg_writer_create(int threads, int max_queue) {
    if (threads <= 0 || max_queue <= 0) { sg::fail(SG_EINVAL, "sg_writer_create: bad arguments"); return nullptr; }
    auto* w = new sg_writer();
    w->max_queue = (size_t)max_queue;
    for (int i = 0; i < threads; ++i) w->threads.emplace_back([w] { w->run(); });
    return w;
}

int sg_writer_submit(sg_writer* w, const char* path_without_ext, const int32_t* h_vec, int V, int formats) {
    if (!w || !path_without_ext || (V > 0 && !h_vec) || V < 0 || !(formats & 3)) return sg::fail(SG_EINVAL, "sg_writer_submit: bad arguments");
    sg_writer::Job j{path_without_ext, std::vector<int32_t>(h_vec, h_vec + V), formats};
    {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv_idle.wait(lk, [&] { return w->q.size() < w->max_queue; });     // back-pressure: bounded memory
        w->q.push_back(std::move(j));
    }
    w->cv_job.notify_one();
    return SG_OK;
}

int sg_writer_flush(sg_writer* w) {
    if (!w) return sg::fail(SG_EINVAL, "sg_writer_flush: null writer");
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv_idle.wait(lk, [&] { return w->q.empty() && w->busy == 0; });
    if (w->first_err) {
        const int rc = w->first_err;
        const std::string msg = w->err;
        w->first_err = 0;
        return sg::fail(rc, "sg_writer: %s", msg.c_str());
    }
    return SG_OK;
}

void sg_writer_destroy(sg_writer* w) {
    if (!w) return;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->stop = true;
    }
    w->cv_job.notify_all();
    for (auto& t : w->threads) t.join();
    delete w;
}

}  // extern "C"
