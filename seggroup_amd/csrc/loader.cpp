// Native scene-pack loader (reference data.py:28-38 + the per-forward file reads of model.py:696-724).
//
// Round 3's driver staged scenes from Python threads: `cache.load_pack` reads a pack into a pinned buffer, uploads it and builds torch
// views -- 719 scenes/s on one thread and no faster with many (the interpreter lock), plus a hipMalloc per scene until torch's caching
// allocator had its working set.  Here a pool of native threads does the whole thing:
//
//     sg_loader_submit(path) -> ticket          worker: open + read the JSON header, pread the payload into ITS pinned buffer, one
//     sg_loader_wait(ticket, &scene)                    hipMemcpyAsync into a pre-allocated DEVICE SLOT on the worker's copy stream, the
//     ... sg_engine_submit(scene) ...                   four per-segment host arrays and seg_of_vertex copied / derived into the slot's
//     sg_loader_release(slot)                           host side, stream sync, done.
//
// Slots (device blob + host arrays) are allocated once, at sg_loader_create: nothing allocates per scene.  A slot is owned by the caller
// from sg_loader_wait until sg_loader_release (the engine reads the device arrays until the scene's ticket is done).
// Pack format: seggroup_amd/cache.py ("SGPACK01" | u32 header length | JSON header | 64-byte aligned little-endian arrays).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "sg_common.h"

namespace {

struct ArrayRef { size_t off = 0, bytes = 0; bool found = false; };

// packs store the [E0,2] adjacency as int32 (half of a pack's bytes were the int64 form: 8.6 of 17 MB at 150k points); the library's
// kernels take the reference's int64 rows (model.py:724).  The worker widens them IN ITS PINNED BUFFER, behind the file's bytes, and the one
// upload carries both (through round 6 a kernel on the worker's stream did it on the device: 40k elements that waited 0.5-1.5 ms for a CU
// among the engine's persistent grids, with the worker waiting for them)

// the header is written by json.dumps: {"name": "...", "N": n, ..., "arrays": {"data": ["<f4", [N, 6], off], ...}}
bool find_int(const std::string& h, const char* key, long long* out) {
    const std::string k = std::string("\"") + key + "\":";
    size_t p = h.find(k);
    if (p == std::string::npos) return false;
    p += k.size();
    while (p < h.size() && h[p] == ' ') ++p;
    char* e = nullptr;
    *out = strtoll(h.c_str() + p, &e, 10);
    return e != h.c_str() + p;
}
bool find_name(const std::string& h, std::string* out) {
    const std::string k = "\"name\":";
    size_t p = h.find(k);
    if (p == std::string::npos) return false;
    p = h.find('"', p + k.size());
    if (p == std::string::npos) return false;
    const size_t q = h.find('"', p + 1);
    if (q == std::string::npos) return false;
    *out = h.substr(p + 1, q - p - 1);
    return true;
}
// "key": ["<dt>", [d0, d1], off]
bool find_array(const std::string& h, const char* key, ArrayRef* a) {
    const size_t arrays = h.find("\"arrays\":");
    if (arrays == std::string::npos) return false;
    const std::string k = std::string("\"") + key + "\": [";
    size_t p = h.find(k, arrays);
    if (p == std::string::npos) return false;
    p += k.size();
    if (p >= h.size() || h[p] != '"') return false;
    const size_t q = h.find('"', p + 1);
    if (q == std::string::npos) return false;
    const std::string dt = h.substr(p + 1, q - p - 1);
    size_t item = 0;
    if (dt == "<f4" || dt == "<i4") item = 4;
    else if (dt == "<i8") item = 8;
    else return false;
    size_t lb = h.find('[', q);
    const size_t rb = h.find(']', lb);
    if (lb == std::string::npos || rb == std::string::npos) return false;
    size_t count = 1;
    const char* c = h.c_str() + lb + 1;
    while (c < h.c_str() + rb) {
        char* e = nullptr;
        const long long d = strtoll(c, &e, 10);
        if (e == c) break;
        count *= (size_t)d;
        c = e;
        while (c < h.c_str() + rb && (*c == ',' || *c == ' ')) ++c;
    }
    const char* o = h.c_str() + rb + 1;
    while (*o == ',' || *o == ' ') ++o;
    char* e = nullptr;
    const long long off = strtoll(o, &e, 10);
    if (e == o) return false;
    a->off = (size_t)off; a->bytes = count * item; a->found = true;
    return true;
}

}  // namespace

struct sg_loader {
    struct Slot {
        char* d_blob = nullptr;
        std::vector<int32_t> seg_first, seg_size, seg_ins, seg_sem, seg_of_vertex;
        std::string name;
        bool busy = false;
    };
    struct Job { int ticket = 0; std::string path; int slot = -1; int rc = 0; std::string err; bool done = false; sg_scene sc{}; };
    int device = 0;
    size_t slot_bytes = 0;              // a pack's bytes
    size_t pin_bytes = 0;               // pinned read buffer of every worker: the pack + its adjacency widened to int64
    size_t blob_bytes = 0;              // a device slot: the same
    char* arena = nullptr;              // all slots: one allocation
    std::vector<Slot> slots;
    std::deque<int> free_slots;
    std::deque<std::shared_ptr<Job>> queue;
    std::map<int, std::shared_ptr<Job>> jobs;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_work, cv_done, cv_slot;
    bool stop = false;
    int next_ticket = 1;
    // Bulk uploads in flight at once.  Every worker has its own copy stream, and with 6-8 of them pushing 13 MB packs the copy engines
    // saturate (55 GB/s) and the scene engine's own small transfers -- a parameter block and an outbox per phase -- wait behind them:
    // tools/exp_h2d_interference.py, engine on resident scenes: 3,004 scenes/s alone, 2,911 / 2,804 / 2,135 / 1,156 with 1 / 2 / 4 / 8
    // threads copying 13 MB buffers back to back (2,334 / 3,522 / 4,243 / 3,980 copies/s).  Workers still read their files side by side;
    // only `copy_limit` of them (default 2) are between hipMemcpyAsync and the end of their upload at a time.
    int copy_limit = 2, copies = 0;
    // SG_LOADER_PROFILE=1: where a worker's time goes (ns summed over all packs): slot wait | open + header | pread | gate wait | issue | host arrays | sync
    bool profile = false;
    // SG_LOADER_DRY=1 (tools/host_scale_rehearsal.py): no HIP call at all -- plain host buffers, the upload replaced by one pass of reads over the
    // staging buffer (what the copy engine's DMA does to host memory); the scene's device pointers are null.  Eight pretend ranks can then
    // share a box with one GPU (or none) and exercise the HOST side of the loader: file reads, header parsing, seg_of_vertex.
    bool dry = false;
    std::atomic<unsigned long long> dry_sink{0};
    std::atomic<long long> prof[8] = {};
    static long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    std::condition_variable cv_copy;
    struct CopyGate {
        sg_loader* L;
        explicit CopyGate(sg_loader* l) : L(l) {
            std::unique_lock<std::mutex> lk(L->mu);
            L->cv_copy.wait(lk, [&] { return L->copies < L->copy_limit; });
            ++L->copies;
        }
        ~CopyGate() {
            { std::lock_guard<std::mutex> lk(L->mu); --L->copies; }
            L->cv_copy.notify_one();
        }
    };

    void run() {
        hipStream_t st = nullptr;
        char* pin = nullptr;
        hipEvent_t ev = nullptr;
        bool ok;
        if (dry) { pin = static_cast<char*>(aligned_alloc(4096, pin_bytes)); ok = pin != nullptr; if (ok) memset(pin, 0, pin_bytes); }
        else {
            (void)hipSetDevice(device);
            // a HIGH-priority stream: under the engine's load an upload took 0.85 ms against 0.3 ms on an idle GPU (tools/time_loader.py,
            // SG_LOADER_PROFILE) -- whatever moves the bytes waits for its turn among ten groups' kernels
            int pr_lo = 0, pr_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi);
            static const bool flat = getenv("SG_LOADER_FLAT_PRIORITY") != nullptr;
            ok = (flat ? hipStreamCreateWithFlags(&st, hipStreamNonBlocking) : hipStreamCreateWithPriority(&st, hipStreamNonBlocking, pr_hi)) == hipSuccess &&
                 hipHostMalloc((void**)&pin, pin_bytes, hipHostMallocDefault) == hipSuccess &&
                 hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
        }
        for (;;) {
            std::shared_ptr<Job> j;
            const long long t_idle = profile ? now_ns() : 0;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || (!queue.empty() && !free_slots.empty()); });
                if (stop) break;
                j = queue.front(); queue.pop_front();
                j->slot = free_slots.front(); free_slots.pop_front();
                slots[j->slot].busy = true;
            }
            if (profile) prof[0] += now_ns() - t_idle;
            int rc = ok ? load(*j, st, pin, ev) : sg::fail(SG_EHIP, "sg_loader: no stream / pinned buffer for this worker");
            {
                std::lock_guard<std::mutex> lk(mu);
                j->rc = rc;
                if (rc < 0) {
                    j->err = sg_last_error();
                    slots[j->slot].busy = false;
                    free_slots.push_back(j->slot);
                    j->slot = -1;
                }
                j->done = true;
            }
            cv_done.notify_all();
            if (rc < 0) cv_work.notify_one();
        }
        if (pin) { if (dry) free(pin); else (void)hipHostFree(pin); }
        if (ev) (void)hipEventDestroy(ev);
        if (st) (void)hipStreamDestroy(st);
    }

    int load(Job& j, hipStream_t st, char* pin, hipEvent_t ev) {
        long long t_ = profile ? now_ns() : 0;
        auto lap = [&](int k) { if (profile) { const long long n_ = now_ns(); prof[k] += n_ - t_; t_ = n_; } };
        const int fd = open(j.path.c_str(), O_RDONLY | O_CLOEXEC);
        if (fd < 0) return sg::fail(SG_EINVAL, "sg_loader: cannot open %s: %s", j.path.c_str(), strerror(errno));
        struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
        char head[12];
        if (pread(fd, head, 12, 0) != 12 || memcmp(head, "SGPACK01", 8) != 0) return sg::fail(SG_EINVAL, "sg_loader: %s is not a scene pack", j.path.c_str());
        uint32_t hlen = 0;
        memcpy(&hlen, head + 8, 4);
        if (hlen == 0 || hlen > (1u << 20)) return sg::fail(SG_EINVAL, "sg_loader: %s: bad header length", j.path.c_str());
        std::string hdr(hlen, '\0');
        if (pread(fd, &hdr[0], hlen, 12) != (ssize_t)hlen) return sg::fail(SG_EINVAL, "sg_loader: %s: short header", j.path.c_str());
        long long N = 0, S = 0, E0 = 0, V = 0;
        Slot& sl = slots[j.slot];
        if (!find_int(hdr, "N", &N) || !find_int(hdr, "S", &S) || !find_int(hdr, "E0", &E0) || !find_int(hdr, "V", &V) || !find_name(hdr, &sl.name))
            return sg::fail(SG_EINVAL, "sg_loader: %s: malformed header", j.path.c_str());
        static const char* kNames[11] = {"data", "adj", "seg_of_point", "seg_points", "seg_off", "unmap", "gt", "seg_first", "seg_size", "seg_ins", "seg_sem"};
        ArrayRef a[11];
        for (int i = 0; i < 11; ++i)
            if (!find_array(hdr, kNames[i], &a[i])) return sg::fail(SG_EINVAL, "sg_loader: %s: array %s missing from the header", j.path.c_str(), kNames[i]);
        const bool adj32 = a[1].bytes == (size_t)E0 * 8 && E0 > 0;                       // int32 pairs (packs written from round 4 on) | int64 pairs
        const size_t expect[11] = {(size_t)N * 24, adj32 ? (size_t)E0 * 8 : (size_t)E0 * 16, (size_t)N * 4, (size_t)N * 4, (size_t)(S + 1) * 4, (size_t)V * 4, (size_t)V * 8,
                                   (size_t)S * 4, (size_t)S * 4, (size_t)S * 4, (size_t)S * 4};
        for (int i = 0; i < 11; ++i)
            if (a[i].bytes != expect[i]) return sg::fail(SG_EINVAL, "sg_loader: %s: array %s has %zu bytes, expected %zu", j.path.c_str(), kNames[i], a[i].bytes, expect[i]);
        struct stat stt;
        if (fstat(fd, &stt) != 0) return sg::fail(SG_EINVAL, "sg_loader: cannot stat %s", j.path.c_str());
        const size_t base = 12 + (size_t)hlen;
        const size_t size = (size_t)stt.st_size - base;
        if (size > slot_bytes) return sg::fail(SG_ENOMEM, "sg_loader: %s holds %zu bytes, a slot %zu", j.path.c_str(), size, slot_bytes);
        const size_t wide_off = (size + 255) / 256 * 256;                                  // the widened adjacency sits behind the file's bytes
        if (adj32 && wide_off + (size_t)E0 * 16 > pin_bytes)
            return sg::fail(SG_ENOMEM, "sg_loader: %s: no room for the widened adjacency (%zu + %zu > %zu)", j.path.c_str(), wide_off, (size_t)E0 * 16, pin_bytes);
        const size_t up_bytes = adj32 ? wide_off + (size_t)E0 * 16 : size;                  // what goes up: the file's bytes (+ the widened rows)
        for (int i = 0; i < 11; ++i)
            if (a[i].off + a[i].bytes > size) return sg::fail(SG_EINVAL, "sg_loader: %s: array %s runs past the end of the file", j.path.c_str(), kNames[i]);
        lap(1);
        size_t got = 0;
        while (got < size) {
            const ssize_t r = pread(fd, pin + got, size - got, (off_t)(base + got));
            if (r < 0) { if (errno == EINTR) continue; return sg::fail(SG_EINVAL, "sg_loader: read error on %s: %s", j.path.c_str(), strerror(errno)); }
            if (r == 0) return sg::fail(SG_EINVAL, "sg_loader: %s is truncated", j.path.c_str());
            got += (size_t)r;
        }
        if (adj32) {
            const int32_t* src = reinterpret_cast<const int32_t*>(pin + a[1].off);
            long long* dst = reinterpret_cast<long long*>(pin + wide_off);
            for (size_t i = 0, n2 = (size_t)E0 * 2; i < n2; ++i) dst[i] = (long long)src[i];
        }
        // one upload; the arrays are typed views into the slot's blob (every array starts on a 64-byte boundary of the file)
        lap(2);
        std::unique_ptr<CopyGate> gate(new CopyGate(this));
        lap(3);
        sg::SdmaTicket sdma;
        bool by_sdma = false;
        if (dry) {
            unsigned long long acc = 0;
            const unsigned long long* w = reinterpret_cast<const unsigned long long*>(pin);
            for (size_t i = 0; i < size / 8; ++i) acc += w[i];
            dry_sink += acc;
        } else {
        // Round 6, SG_LOADER_COPY=sdma: the pack goes up through the HSA runtime's copy interface (sdma.cpp) instead of hipMemcpyAsync.  For the engine's
        // label vectors (device -> host) that path is worth 6 % of the headline; for the uploads it measured +2-3 % (.npy: 1,894-1,928 against 1,840-1,875
        // scenes/s steady) and nothing with .txt output -- not enough to make it the default (files byte-identical either way: tests/test_gpu_loader.py).
        static const bool want_sdma = getenv("SG_LOADER_COPY") && std::string(getenv("SG_LOADER_COPY")) == "sdma";
        if (want_sdma && sg::sdma_available() && sg::sdma_issue(sl.d_blob, pin, up_bytes, &sdma) == SG_OK) by_sdma = true;
        else {
        sg::err_buf()[0] = 0;
        if (hipMemcpyAsync(sl.d_blob, pin, up_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return sg::fail(SG_EHIP, "sg_loader: upload of %s failed", j.path.c_str());
        if (hipEventRecord(ev, st) != hipSuccess) return sg::fail(SG_EHIP, "sg_loader: upload of %s failed", j.path.c_str());
        }
        }
        lap(4);
        auto host = [&](int i, std::vector<int32_t>& v) { v.resize((size_t)S); memcpy(v.data(), pin + a[i].off, (size_t)S * 4); };
        host(7, sl.seg_first); host(8, sl.seg_size); host(9, sl.seg_ins); host(10, sl.seg_sem);
        // seg_of_vertex[v] = seg_of_point[unmap[v]] (-1 where unmap[v] is not a point): the host-side look-up of the compact label transfer
        sl.seg_of_vertex.resize((size_t)V);
        {
            const int32_t* sop = reinterpret_cast<const int32_t*>(pin + a[2].off);
            const int32_t* um = reinterpret_cast<const int32_t*>(pin + a[5].off);
            for (long long v = 0; v < V; ++v) { const int32_t p_ = um[v]; sl.seg_of_vertex[(size_t)v] = (p_ >= 0 && p_ < N) ? sop[p_] : -1; }
        }
        lap(5);
        if (!dry && by_sdma) {
            if (sg::sdma_wait(&sdma) != SG_OK) return sg::fail(SG_EHIP, "sg_loader: upload of %s failed (copy engine)", j.path.c_str());
        } else
        if (!dry && hipEventSynchronize(ev) != hipSuccess) return sg::fail(SG_EHIP, "sg_loader: upload of %s failed", j.path.c_str());
        gate.reset();                                                // the pinned buffer has been read: the next worker's copy may start
        lap(7);
        if (!dry && hipStreamSynchronize(st) != hipSuccess) return sg::fail(SG_EHIP, "sg_loader: upload of %s failed", j.path.c_str());
        lap(6);
        sg_scene& sc = j.sc;
        sc.N = (int)N; sc.S = (int)S; sc.E0 = (int)E0; sc.V = (int)V;
        sc.d_data = reinterpret_cast<const float*>(sl.d_blob + a[0].off);
        sc.d_adj = reinterpret_cast<const int64_t*>(sl.d_blob + (adj32 ? wide_off : a[1].off));
        sc.d_seg_of_point = reinterpret_cast<const int32_t*>(sl.d_blob + a[2].off);
        sc.d_seg_points = reinterpret_cast<const int32_t*>(sl.d_blob + a[3].off);
        sc.d_seg_off = reinterpret_cast<const int32_t*>(sl.d_blob + a[4].off);
        sc.d_unmap = reinterpret_cast<const int32_t*>(sl.d_blob + a[5].off);
        sc.d_gt = reinterpret_cast<const int32_t*>(sl.d_blob + a[6].off);
        sc.h_seg_first = sl.seg_first.data(); sc.h_seg_size = sl.seg_size.data(); sc.h_seg_ins = sl.seg_ins.data(); sc.h_seg_sem = sl.seg_sem.data();
        sc.h_seg_of_vertex = sl.seg_of_vertex.data();
        return SG_OK;
    }
};

extern "C" {

sg_loader* sg_loader_create(int threads, int slots, size_t slot_bytes) { return sg_loader_create_sized(threads, slots, slot_bytes, 0); }

sg_loader* sg_loader_create_sized(int threads, int slots, size_t slot_bytes, size_t max_edges) {
    if (threads <= 0 || slots <= 0 || slot_bytes == 0) { sg::fail(SG_EINVAL, "sg_loader_create: bad arguments"); return nullptr; }
    auto* L = new sg_loader();
    L->dry = getenv("SG_LOADER_DRY") != nullptr;
    if (!L->dry && hipGetDevice(&L->device) != hipSuccess) { sg::fail(SG_EHIP, "sg_loader_create: no HIP device"); delete L; return nullptr; }
    L->slot_bytes = (slot_bytes + 4095) / 4096 * 4096;
    if (const char* e = getenv("SG_LOADER_COPIES")) L->copy_limit = std::max(1, atoi(e));
    L->profile = getenv("SG_LOADER_PROFILE") != nullptr;
    // a slot holds a pack and, behind it, its int32 adjacency widened to int64 (16 bytes per edge).  max_edges = 0: the caller does not know
    // its packs' edge counts -- the adjacency is at most the whole file, i.e. three times the pack at worst (~9 GB for 256
    // slots of 150k-point packs; with the edge count it is ~1.7 x the file).  ONE allocation for all slots (one hipMalloc / hipFree instead
    // of `slots` of them: start-up and tear-down of the driver)
    // a multiple of 256: every slot starts where the first one does modulo a cache line, so the packs' "every array starts on a 64-byte boundary"
    // holds on the device for slots 1 .. n - 1 too (the sized stride was only a multiple of 16)
    L->pin_bytes = sg::align_up(max_edges > 0 ? L->slot_bytes + max_edges * 16 + 8192 : 3 * L->slot_bytes + 4096, 4096);
    L->blob_bytes = L->pin_bytes;
    L->slots.resize((size_t)slots);
    if (L->dry) L->blob_bytes = 0;                               // no device side: the scenes' device pointers are offsets from null
    if (!L->dry && hipMalloc((void**)&L->arena, (size_t)slots * L->blob_bytes) != hipSuccess) {
        sg::fail(SG_ENOMEM, "sg_loader_create: cannot allocate %d device slots of %zu bytes", slots, L->blob_bytes);
        delete L;
        return nullptr;
    }
    for (int i = 0; i < slots; ++i) {
        L->slots[i].d_blob = L->arena + (size_t)i * L->blob_bytes;
        L->free_slots.push_back(i);
    }
    for (int i = 0; i < threads; ++i) L->threads.emplace_back([L] { L->run(); });
    return L;
}

int sg_loader_submit(sg_loader* L, const char* path) {
    if (!L || !path) return sg::fail(SG_EINVAL, "sg_loader_submit: bad arguments");
    auto j = std::make_shared<sg_loader::Job>();
    j->path = path;
    {
        std::lock_guard<std::mutex> lk(L->mu);
        j->ticket = L->next_ticket++;
        L->jobs[j->ticket] = j;
        L->queue.push_back(j);
    }
    L->cv_work.notify_one();
    return j->ticket;
}

int sg_loader_wait(sg_loader* L, int ticket, sg_scene* out, int* slot, char* name, int name_cap) {
    if (!L || !out || !slot) return sg::fail(SG_EINVAL, "sg_loader_wait: bad arguments");
    std::shared_ptr<sg_loader::Job> j;
    {
        std::unique_lock<std::mutex> lk(L->mu);
        auto it = L->jobs.find(ticket);
        if (it == L->jobs.end()) return sg::fail(SG_EINVAL, "sg_loader_wait: unknown ticket %d", ticket);
        j = it->second;
        L->cv_done.wait(lk, [&] { return j->done; });
        L->jobs.erase(ticket);
    }
    if (j->rc < 0) return sg::fail(j->rc, "%s", j->err.c_str());
    *out = j->sc;
    *slot = j->slot;
    if (name && name_cap > 0) {
        const std::string& n = L->slots[j->slot].name;
        const size_t c = std::min((size_t)name_cap - 1, n.size());
        memcpy(name, n.data(), c);
        name[c] = '\0';
    }
    return SG_OK;
}

int sg_loader_set_copy_limit(sg_loader* L, int uploads_in_flight) {
    if (!L || uploads_in_flight < 1) return sg::fail(SG_EINVAL, "sg_loader_set_copy_limit: bad arguments");
    { std::lock_guard<std::mutex> lk(L->mu); L->copy_limit = uploads_in_flight; }
    L->cv_copy.notify_all();
    return SG_OK;
}

int sg_loader_release(sg_loader* L, int slot) {
    if (!L || slot < 0 || slot >= (int)L->slots.size()) return sg::fail(SG_EINVAL, "sg_loader_release: bad slot");
    {
        std::lock_guard<std::mutex> lk(L->mu);
        if (!L->slots[slot].busy) return sg::fail(SG_EINVAL, "sg_loader_release: slot %d is not in use", slot);
        L->slots[slot].busy = false;
        L->free_slots.push_back(slot);
    }
    L->cv_work.notify_one();
    return SG_OK;
}

void sg_loader_destroy(sg_loader* L) {
    if (!L) return;
    {
        std::lock_guard<std::mutex> lk(L->mu);
        L->stop = true;
    }
    L->cv_work.notify_all();
    for (auto& t : L->threads) if (t.joinable()) t.join();
    if (L->profile)
        fprintf(stderr, "[sg_loader profile] %zu threads, ms summed over all packs: idle / slot wait %.1f | open + header %.1f | pread %.1f | gate wait %.1f | issue %.1f | host arrays %.1f | copy done %.1f | widen + sync %.1f\n",
                L->threads.size(), L->prof[0] / 1e6, L->prof[1] / 1e6, L->prof[2] / 1e6, L->prof[3] / 1e6, L->prof[4] / 1e6, L->prof[5] / 1e6, L->prof[7] / 1e6, L->prof[6] / 1e6);
    if (L->arena && !L->dry) (void)hipFree(L->arena);
    delete L;
}

}  // extern "C"
