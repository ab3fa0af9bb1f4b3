// Error plumbing, version / device probe, and the label-file writers (a16 file side,
// reference seggroup/model.py:536-547) of libseggroup_hip.so.
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <fcntl.h>
#include <sys/uio.h>
#include <unistd.h>
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

// roctx (sg_common.h): resolved once, at the first range; SG_ROCTX unset = two null pointers and nothing else happens
namespace {
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi() {
        if (!getenv("SG_ROCTX")) return;
        void* h = nullptr;
        for (const char* n : {"librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so"})
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const RoctxApi& roctx_api() { static const RoctxApi a; return a; }
}  // namespace
void roctx_push(const char* name) { if (roctx_api().push) (void)roctx_api().push(name); }
void roctx_pop() { if (roctx_api().pop) (void)roctx_api().pop(); }

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

// Round 4: nearly every label is small -- segment numbers below S, instance / semantic ids, -1 -- so the text of every value in [-1, 9999]
// comes ready-made from a table: "digits + newline" left-aligned in eight bytes and its length.  A value is then one 8-byte store (the
// bytes behind the newline are overwritten by the next value; the buffer has slack) and an add: ~1.5 ns instead of ~5 per value
// (14 x 150k values per scene: 11 -> ~3.5 ms of one core).  Larger values take put_u32.
struct SmallText {
    uint64_t word[10001];
    uint8_t len[10001];
    SmallText() {
        for (int i = 0; i <= 10000; ++i) {
            char t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            char* o = t;
            const int v = i - 1;
            if (v < 0) { *o++ = '-'; *o++ = '1'; }
            else o = put_u32(o, (uint32_t)v);
            *o++ = '\n';
            memcpy(&word[i], t, 8);
            len[i] = (uint8_t)(o - t);
        }
    }
};
const SmallText kSmallText;

struct TlBuf {
    char* p = nullptr;
    size_t cap = 0;
    ~TlBuf() { free(p); }
    char* need(size_t n) {
        if (n > cap) { free(p); p = (char*)malloc(n); cap = p ? n : 0; }
        return p;
    }
};
int write_vector_files(int dfd, const char* name, const int32_t* vec, int V, int formats, TlBuf& tl);      // below, with the writer pool
}  // namespace

int sg_write_label_txt(const char* path, const int32_t* h_vec, int V) {
    if (!path || (V > 0 && !h_vec) || V < 0) return sg::fail(SG_EINVAL, "sg_write_label_txt: bad arguments");
    const size_t n = strlen(path);
    if (n < 4 || strcmp(path + n - 4, ".txt") != 0) return sg::fail(SG_EINVAL, "sg_write_label_txt: %s does not end in .txt", path);
    static thread_local TlBuf tl;
    return write_vector_files(AT_FDCWD, std::string(path, n - 4).c_str(), h_vec, V, 1, tl);
}

// NumPy .npy v1.0, little-endian int32, C order, shape (V,)
int sg_write_label_npy(const char* path, const int32_t* h_vec, int V) {
    if (!path || (V > 0 && !h_vec) || V < 0) return sg::fail(SG_EINVAL, "sg_write_label_npy: bad arguments");
    const size_t n = strlen(path);
    if (n < 4 || strcmp(path + n - 4, ".npy") != 0) return sg::fail(SG_EINVAL, "sg_write_label_npy: %s does not end in .npy", path);
    static thread_local TlBuf tl;
    return write_vector_files(AT_FDCWD, std::string(path, n - 4).c_str(), h_vec, V, 2, tl);
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
// Round 3: a job is a SCENE -- its label vectors stay where the engine wrote them (the caller's pinned buffer: no copy; the caller
// reuses that buffer only behind sg_writer_wait_tag), the worker creates the scene's files itself with openat / writev on the
// directory's descriptor (36,000 file creations per second at the bench's rate: one path walk per scene instead of per file, no
// stdio buffer copy).  sg_writer_submit (one vector, copied) stays for callers that hand over a temporary.
namespace {
const char* const kWriterNames[SG_NUM_LABEL_VECTORS] = {"layer_1.seg", "layer_1.ins", "layer_1.sem", "layer_2.seg", "layer_2.ins", "layer_2.sem",
                                                         "layer_3.seg", "layer_3.ins", "layer_3.sem", "layer_4.seg", "layer_4.ins", "layer_4.sem",
                                                         "final.ins", "final.sem"};

int write_all(int fd, const struct iovec* iov, int cnt) {
    struct iovec v[4];
    for (int i = 0; i < cnt; ++i) v[i] = iov[i];
    int first = 0;
    while (first < cnt) {
        const ssize_t w = writev(fd, v + first, cnt - first);
        if (w < 0) { if (errno == EINTR) continue; return -1; }
        size_t left = (size_t)w;
        while (first < cnt && left >= v[first].iov_len) { left -= v[first].iov_len; ++first; }
        if (first < cnt) { v[first].iov_base = (char*)v[first].iov_base + left; v[first].iov_len -= left; }
    }
    return 0;
}

int npy_header(char* hdr, int V) {                              // NumPy .npy v1.0, little-endian int32, C order, shape (V,): bytes written
    char dict[128];
    const int n = snprintf(dict, sizeof dict, "{'descr': '<i4', 'fortran_order': False, 'shape': (%d,), }", V);
    const int unpadded = 10 + n + 1;                            // magic(6)+ver(2)+len(2)+dict+'\n'
    const int pad = (64 - unpadded % 64) % 64;
    memcpy(hdr, "\x93NUMPY\x01\x00", 8);
    const uint16_t hlen = (uint16_t)(n + pad + 1);
    hdr[8] = (char)(hlen & 0xff); hdr[9] = (char)(hlen >> 8);
    memcpy(hdr + 10, dict, n);
    memset(hdr + 10 + n, ' ', pad);
    hdr[10 + n + pad] = '\n';
    return 10 + n + pad + 1;
}

// one label vector as <dir>/<name>.txt and / or .npy; dfd = the directory's descriptor (or AT_FDCWD with a full path in `name`)
int write_vector_files(int dfd, const char* name, const int32_t* vec, int V, int formats, TlBuf& tl) {
    std::string fname_s;                                        // no fixed-size buffer: a long path must not lose its extension
    const char* fname = nullptr;
    if (formats & 2) {
        fname_s.assign(name).append(".npy");
        fname = fname_s.c_str();
        const int fd = openat(dfd, fname, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
        if (fd < 0) return sg::fail(SG_EINVAL, "label writer: cannot open %s: %s", fname, strerror(errno));
        char hdr[256];
        const int hl = npy_header(hdr, V);
        const struct iovec iov[2] = {{hdr, (size_t)hl}, {(void*)vec, (size_t)V * 4}};
        const int rc = write_all(fd, iov, V > 0 ? 2 : 1);
        if (close(fd) != 0 || rc != 0) return sg::fail(SG_EINVAL, "label writer: short write to %s", fname);
    }
    if (formats & 1) {
        char* const buf = tl.need((size_t)V * 12 + 16);
        if (!buf) return sg::fail(SG_ENOMEM, "label writer: out of memory");
        char* o = buf;
        for (int i = 0; i < V; ++i) {
            const int32_t v = vec[i];
            const uint32_t idx = (uint32_t)v + 1u;                   // -1 -> 0 ... 9999 -> 10000; anything else wraps beyond the table
            if (idx <= 10000u) {
                memcpy(o, &kSmallText.word[idx], 8);
                o += kSmallText.len[idx];
                continue;
            }
            if (v < 0) { *o++ = '-'; o = put_u32(o, (uint32_t)(-(int64_t)v)); }
            else o = put_u32(o, (uint32_t)v);
            *o++ = '\n';
        }
        fname_s.assign(name).append(".txt");
        fname = fname_s.c_str();
        const int fd = openat(dfd, fname, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
        if (fd < 0) return sg::fail(SG_EINVAL, "label writer: cannot open %s: %s", fname, strerror(errno));
        const struct iovec iov[1] = {{buf, (size_t)(o - buf)}};
        const int rc = write_all(fd, iov, 1);
        if (close(fd) != 0 || rc != 0) return sg::fail(SG_EINVAL, "label writer: short write to %s", fname);
    }
    return SG_OK;
}
}  // namespace

struct sg_writer {
    // a scene (dir + nvec vectors of V values at `base`, stride V; not owned) or one owned vector (`own`, full path without extension in dir)
    // ... or a scene as tables: `own` = [nvec * S] tables followed by [V] seg_of_vertex, S > 0
    struct Job { std::string dir; const int32_t* base = nullptr; int V = 0, nvec = 0, formats = 0, S = 0; long long tag = 0; std::vector<int32_t> own; };
    std::mutex mu;
    std::condition_variable cv_job, cv_idle;
    std::deque<Job> q;
    std::vector<std::thread> threads;
    std::map<long long, int> live;                              // tag -> jobs queued or being written
    size_t max_queue = 64;
    int busy = 0;
    bool stop = false;
    int first_err = 0;
    std::string err;

    void run() {
        TlBuf tl;
        std::vector<int32_t> expand;
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
            if (j.S > 0) {
                // tables + seg_of_vertex: expand one vector at a time into this thread's buffer, then format / write it like any other
                const int dfd = open(j.dir.c_str(), O_RDONLY | O_DIRECTORY | O_CLOEXEC);
                if (dfd < 0) rc = sg::fail(SG_EINVAL, "label writer: cannot open directory %s: %s", j.dir.c_str(), strerror(errno));
                const int32_t* tab = j.own.data();
                const int32_t* sov = tab + (size_t)j.nvec * j.S;
                expand.resize((size_t)std::max(j.V, 1));
                for (int v = 0; v < j.nvec && rc == 0; ++v) {
                    const int32_t* t = tab + (size_t)v * j.S;
                    for (int i = 0; i < j.V; ++i) { const int s_ = sov[i]; expand[i] = (s_ >= 0 && s_ < j.S) ? t[s_] : -1; }
                    rc = write_vector_files(dfd, kWriterNames[v], expand.data(), j.V, j.formats, tl);
                }
                if (dfd >= 0) close(dfd);
            } else if (j.base) {
                const int dfd = open(j.dir.c_str(), O_RDONLY | O_DIRECTORY | O_CLOEXEC);
                if (dfd < 0) rc = sg::fail(SG_EINVAL, "label writer: cannot open directory %s: %s", j.dir.c_str(), strerror(errno));
                for (int v = 0; v < j.nvec && rc == 0; ++v) rc = write_vector_files(dfd, kWriterNames[v], j.base + (size_t)v * j.V, j.V, j.formats, tl);
                if (dfd >= 0) close(dfd);
            } else {
                rc = write_vector_files(AT_FDCWD, j.dir.c_str(), j.own.data(), (int)j.own.size(), j.formats, tl);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (rc < 0 && first_err == 0) { first_err = rc; err = sg_last_error(); }
                --busy;
                auto it = live.find(j.tag);
                if (it != live.end() && --it->second == 0) live.erase(it);
            }
            cv_idle.notify_all();
        }
    }
    int push(Job&& j) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_idle.wait(lk, [&] { return q.size() < max_queue; });           // back-pressure: bounded queue
            ++live[j.tag];
            q.push_back(std::move(j));
        }
        cv_job.notify_one();
        return SG_OK;
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
    sg_writer::Job j;
    j.dir = path_without_ext; j.own.assign(h_vec, h_vec + V); j.formats = formats; j.tag = -1;
    return w->push(std::move(j));
}

int sg_writer_submit_scene(sg_writer* w, const char* out_dir, const int32_t* h_labels, int V, int nvec, int formats, long long tag) {
    if (!w || !out_dir || (V > 0 && !h_labels) || V < 0 || nvec < 0 || nvec > SG_NUM_LABEL_VECTORS || !(formats & 3))
        return sg::fail(SG_EINVAL, "sg_writer_submit_scene: bad arguments");
    sg_writer::Job j;
    j.dir = out_dir; j.base = h_labels; j.V = V; j.nvec = nvec; j.formats = formats; j.tag = tag;
    return w->push(std::move(j));
}

int sg_writer_submit_scene_tables(sg_writer* w, const char* out_dir, const int32_t* h_tables, int S, const int32_t* h_seg_of_vertex, int V, int nvec,
                                  int formats, long long tag) {
    if (!w || !out_dir || !h_tables || S <= 0 || (V > 0 && !h_seg_of_vertex) || V < 0 || nvec < 0 || nvec > SG_NUM_LABEL_VECTORS || !(formats & 3))
        return sg::fail(SG_EINVAL, "sg_writer_submit_scene_tables: bad arguments");
    sg_writer::Job j;
    j.dir = out_dir; j.V = V; j.nvec = nvec; j.formats = formats; j.S = S;
    j.tag = -1;                                                  // owns its data: nothing of the caller's to wait for
    (void)tag;
    j.own.resize((size_t)nvec * S + (size_t)V);
    std::memcpy(j.own.data(), h_tables, (size_t)nvec * S * 4);
    if (V > 0) std::memcpy(j.own.data() + (size_t)nvec * S, h_seg_of_vertex, (size_t)V * 4);
    return w->push(std::move(j));
}

int sg_expand_labels(const int32_t* h_tables, int nvec, int S, const int32_t* h_seg_of_vertex, int V, int32_t* h_out) {
    if (!h_tables || !h_out || nvec < 0 || S <= 0 || V < 0 || (V > 0 && !h_seg_of_vertex)) return sg::fail(SG_EINVAL, "sg_expand_labels: bad arguments");
    for (int t = 0; t < nvec; ++t) {
        const int32_t* tab = h_tables + (size_t)t * S;
        int32_t* o = h_out + (size_t)t * V;
        for (int i = 0; i < V; ++i) { const int s_ = h_seg_of_vertex[i]; o[i] = (s_ >= 0 && s_ < S) ? tab[s_] : -1; }
    }
    return SG_OK;
}

int sg_writer_wait_tag(sg_writer* w, long long tag) {
    if (!w) return sg::fail(SG_EINVAL, "sg_writer_wait_tag: null writer");
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv_idle.wait(lk, [&] {
        auto it = w->live.lower_bound(0);                       // tags < 0: owned copies, nothing to wait for
        return it == w->live.end() || it->first > tag;
    });
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
