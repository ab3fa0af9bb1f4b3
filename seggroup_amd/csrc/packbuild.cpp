// Native scene-pack builder (missing #4): the reference's per-scene files -> one `.sgpack`, without Python.
//
// The first run over a reference-format tree builds a pack per scene (seggroup_amd/cache.py).  In Python that is ~2,000 interpreter-level calls
// per scene under the GIL (zipfile, pickle, json, NumPy glue): ~260 scenes/s however many threads share it, 12x under what the engine then
// consumes.  Here a scene is: six files read once, the `.pth` containers taken apart in place (a STORED zip -- torch.save's own writer -- holding
// `data.pkl`, a protocol-2 pickle of ONE tensor, and the raw storage), the seg.json lists parsed by sg_parse_seg_json's code, the staging
// of cache.stage_arrays, and the pack written exactly as cache.write_pack writes it (byte for byte: tests/test_cache.py holds the two together).
// sg_pack_build_many runs a pool of plain threads over a list of scenes.  Untrusted input: every offset, size and stride read from a file is
// checked against the bytes that are there; anything this reader does not know (a compressed member, another pickle shape, a dtype it has no
// conversion for) is an error, and the caller falls back to the Python builder for that scene.  Host only; built under ASan/UBSan too.
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

#include "sg_common.h"

namespace {

struct Err { std::string msg; };
#define PB_FAIL(...) do { char b__[512]; snprintf(b__, sizeof b__, __VA_ARGS__); throw Err{b__}; } while (0)

// A source file, read into a buffer the calling thread keeps between packs.  (Round 6, first form: fread into a zero-filled std::string -- a 19 MB fill + a
// 19 MB copy per scene; second form: mmap / munmap per file -- no fill, no copy, but with every large std::vector below also an mmap / munmap of its own
// (glibc's malloc maps blocks beyond 128 KB) a pack was ~20 map / unmap pairs and a few thousand page faults on ONE address space: 32 builder threads
// spent their time on the process's mmap lock and on TLB shootdowns, 1,002 packs/s on 8 threads, 1,189-1,438 on 32.  With the allocator told to keep
// large blocks (MALLOC_MMAP_THRESHOLD_) the same build ran 1,549 on 32.  Now nothing large is mapped or freed per pack: files are pread into pooled
// buffers, the staging vectors are the thread's own.)  The subset of std::string's interface the parsers below use.
struct RawPool {
    std::vector<std::pair<char*, size_t>> idle;
    ~RawPool() { for (auto& r : idle) free(r.first); }
    char* get(size_t n, size_t* cap) {
        size_t best = idle.size();
        for (size_t k = 0; k < idle.size(); ++k)
            if (idle[k].second >= n && (best == idle.size() || idle[k].second < idle[best].second)) best = k;
        if (best < idle.size()) { char* p = idle[best].first; *cap = idle[best].second; idle[best] = idle.back(); idle.pop_back(); return p; }
        const size_t c = std::max<size_t>(n + n / 4, (size_t)1 << 16);        // room for the next, slightly larger, file of the kind
        char* p = static_cast<char*>(malloc(c));
        if (!p) PB_FAIL("out of memory (%zu bytes)", c);
        *cap = c;
        return p;
    }
    void put(char* p, size_t cap) { if (idle.size() < 16) idle.emplace_back(p, cap); else free(p); }
};
thread_local RawPool t_pool;

struct Buf {
    const char* p = nullptr;
    size_t n = 0;
    char* raw = nullptr;
    size_t cap = 0;
    Buf() = default;
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;
    ~Buf() { if (raw) t_pool.put(raw, cap); }                // (created and destroyed by the thread that builds the pack)
    size_t size() const { return n; }
    const char* data() const { return p; }
    char operator[](size_t at) const { return p[at]; }
    std::string substr(size_t at, size_t len) const { return std::string(p + at, len); }
    size_t find(char c, size_t from) const {
        if (from >= n) return std::string::npos;
        const void* q = memchr(p + from, c, n - from);
        return q ? (size_t)(static_cast<const char*>(q) - p) : std::string::npos;
    }
};

std::shared_ptr<Buf> slurp(const std::string& path) {
    auto b = std::make_shared<Buf>();
    const int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) PB_FAIL("cannot open %s: %s", path.c_str(), strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 0) { close(fd); PB_FAIL("cannot size %s", path.c_str()); }
    b->n = (size_t)st.st_size;
    if (b->n) {
        try { b->raw = t_pool.get(b->n, &b->cap); } catch (...) { close(fd); throw; }
        size_t got = 0;
        while (got < b->n) {
            const ssize_t r = pread(fd, b->raw + got, b->n - got, (off_t)got);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) break;
            got += (size_t)r;
        }
        if (got != b->n) { close(fd); PB_FAIL("short read of %s", path.c_str()); }
        b->p = b->raw;
    }
    close(fd);
    return b;
}

inline uint16_t rd16(const Buf& b, size_t at) { if (at + 2 > b.size()) PB_FAIL("zip: read past the end"); return (uint16_t)((uint8_t)b[at] | ((uint8_t)b[at + 1] << 8)); }
inline uint32_t rd32(const Buf& b, size_t at) { if (at + 4 > b.size()) PB_FAIL("zip: read past the end"); uint32_t v; memcpy(&v, b.data() + at, 4); return v; }
inline uint64_t rd64(const Buf& b, size_t at) { if (at + 8 > b.size()) PB_FAIL("zip: read past the end"); uint64_t v; memcpy(&v, b.data() + at, 8); return v; }

struct Member { size_t data_at = 0, size = 0; };

// name -> (offset of the member's bytes, size) for every STORED member of a zip archive held in `b`
std::map<std::string, Member> zip_members(const Buf& b, const std::string& path) {
    if (b.size() < 22) PB_FAIL("%s: not a zip archive", path.c_str());
    size_t eocd = std::string::npos;
    for (size_t at = b.size() - 22, stop = b.size() > 65557 ? b.size() - 65557 : 0;; --at) {
        if (rd32(b, at) == 0x06054b50u) { eocd = at; break; }
        if (at == stop) break;
    }
    if (eocd == std::string::npos) PB_FAIL("%s: no end-of-central-directory record", path.c_str());
    uint64_t entries = rd16(b, eocd + 10), cd_size = rd32(b, eocd + 12), cd_off = rd32(b, eocd + 16);
    if (entries == 0xffffu || cd_size == 0xffffffffu || cd_off == 0xffffffffu) {                 // the values live in the zip64 record
        if (eocd < 20 || rd32(b, eocd - 20) != 0x07064b50u) PB_FAIL("%s: zip64 locator missing", path.c_str());
        const uint64_t rec = rd64(b, eocd - 20 + 8);
        if (rec + 56 > b.size() || rd32(b, rec) != 0x06064b50u) PB_FAIL("%s: zip64 record missing", path.c_str());
        entries = rd64(b, rec + 32); cd_size = rd64(b, rec + 40); cd_off = rd64(b, rec + 48);
    }
    if (cd_off > b.size() || cd_size > b.size() - cd_off) PB_FAIL("%s: central directory outside the file", path.c_str());
    std::map<std::string, Member> out;
    size_t at = cd_off;
    for (uint64_t e = 0; e < entries; ++e) {
        if (rd32(b, at) != 0x02014b50u) PB_FAIL("%s: bad central directory entry", path.c_str());
        const uint16_t method = rd16(b, at + 10), nlen = rd16(b, at + 28), xlen = rd16(b, at + 30), clen = rd16(b, at + 32);
        const uint32_t csize = rd32(b, at + 20), usize = rd32(b, at + 24), lho = rd32(b, at + 42);
        if (at + 46 + nlen > b.size()) PB_FAIL("%s: member name outside the file", path.c_str());
        const std::string name = b.substr(at + 46, nlen);
        at += 46u + nlen + xlen + clen;
        if (csize == 0xffffffffu || usize == 0xffffffffu || lho == 0xffffffffu) PB_FAIL("%s: member %s beyond 4 GB", path.c_str(), name.c_str());
        if (method != 0 || csize != usize) continue;                                               // compressed: not ours (the caller will miss it)
        if (rd32(b, lho) != 0x04034b50u) PB_FAIL("%s: bad local header of %s", path.c_str(), name.c_str());
        const size_t data_at = (size_t)lho + 30u + rd16(b, lho + 26) + rd16(b, lho + 28);
        if (data_at > b.size() || usize > b.size() - data_at) PB_FAIL("%s: member %s outside the file", path.c_str(), name.c_str());
        out[name] = Member{data_at, usize};
    }
    return out;
}

// ---- the pickle of torch.save(tensor): a stack machine over the dozen opcodes such a file holds ----
struct Val {
    enum Kind { NONE, INT, STR, GLOBAL, TUPLE, BOOL, STORAGE, TENSOR, DICT, MARK } kind = NONE;
    long long i = 0;
    std::string s;                       // STR / GLOBAL ("module name") / STORAGE key
    std::vector<Val> items;              // TUPLE
    int item_size = 0;                   // STORAGE: bytes per element; TENSOR keeps the storage's
    char type = 0;                       // STORAGE / TENSOR: 'f' float, 'i' signed, 'u' unsigned, 'b' bool
    long long offset = 0;                // TENSOR
    std::vector<long long> size, stride;
};

struct Tensor {
    char type = 0; int item = 0;
    std::vector<long long> shape;
    std::shared_ptr<Buf> file;           // the archive's bytes (kept alive: a contiguous tensor is a view into them)
    std::string own;                     // ... or the gathered copy of a strided view
    const char* data = nullptr;          // contiguous, little endian
    long long numel() const { long long n = 1; for (long long v : shape) n *= v; return n; }
};

Val storage_of(const Val& t) {
    if (t.kind != Val::TUPLE || t.items.size() != 5 || t.items[0].kind != Val::STR || t.items[0].s != "storage" || t.items[1].kind != Val::GLOBAL ||
        t.items[2].kind != Val::STR) PB_FAIL("pickle: unexpected persistent id");
    static const struct { const char* n; char t; int sz; } kinds[] = {{"torch FloatStorage", 'f', 4}, {"torch DoubleStorage", 'f', 8}, {"torch LongStorage", 'i', 8},
        {"torch IntStorage", 'i', 4}, {"torch ShortStorage", 'i', 2}, {"torch CharStorage", 'i', 1}, {"torch ByteStorage", 'u', 1}, {"torch BoolStorage", 'b', 1}};
    for (const auto& k : kinds)
        if (t.items[1].s == k.n) { Val v; v.kind = Val::STORAGE; v.s = t.items[2].s; v.type = k.t; v.item_size = k.sz; return v; }
    PB_FAIL("pickle: storage type %s is not supported", t.items[1].s.c_str());
}

std::vector<long long> ints_of(const Val& t) {
    if (t.kind != Val::TUPLE) PB_FAIL("pickle: size / stride is not a tuple");
    std::vector<long long> out;
    for (const Val& v : t.items) { if (v.kind != Val::INT) PB_FAIL("pickle: size / stride entry is not an integer"); out.push_back(v.i); }
    return out;
}

Val unpickle(const Buf& b, size_t at, size_t end) {
    std::vector<Val> st;
    std::map<long long, Val> memo;
    auto need = [&](size_t n) { if (at + n > end) PB_FAIL("pickle: truncated"); };
    auto pop = [&]() { if (st.empty()) PB_FAIL("pickle: stack underflow"); Val v = std::move(st.back()); st.pop_back(); return v; };
    auto pop_to_mark = [&]() {
        std::vector<Val> items;
        for (;;) { Val v = pop(); if (v.kind == Val::MARK) break; items.push_back(std::move(v)); }
        std::reverse(items.begin(), items.end());
        return items;
    };
    for (int steps = 0; steps < 100000; ++steps) {
        need(1);
        const uint8_t op = (uint8_t)b[at++];
        switch (op) {
        case 0x80: need(1); if ((uint8_t)b[at++] > 5) PB_FAIL("pickle: protocol too new"); break;           // PROTO
        case 'c': {                                                                                        // GLOBAL module\nname\n
            const size_t a = b.find('\n', at); if (a == std::string::npos || a >= end) PB_FAIL("pickle: GLOBAL without a module line");
            const size_t c = b.find('\n', a + 1); if (c == std::string::npos || c >= end) PB_FAIL("pickle: GLOBAL without a name line");
            Val v; v.kind = Val::GLOBAL; v.s = b.substr(at, a - at) + " " + b.substr(a + 1, c - a - 1); st.push_back(std::move(v)); at = c + 1; break; }
        case 'q': need(1); if (st.empty()) PB_FAIL("pickle: BINPUT on an empty stack"); memo[(uint8_t)b[at++]] = st.back(); break;
        case 'r': need(4); if (st.empty()) PB_FAIL("pickle: LONG_BINPUT on an empty stack"); memo[rd32(b, at)] = st.back(); at += 4; break;
        case 'h': { need(1); auto it = memo.find((uint8_t)b[at++]); if (it == memo.end()) PB_FAIL("pickle: BINGET of an unknown memo"); st.push_back(it->second); break; }
        case 'j': { need(4); auto it = memo.find(rd32(b, at)); at += 4; if (it == memo.end()) PB_FAIL("pickle: LONG_BINGET of an unknown memo"); st.push_back(it->second); break; }
        case '(': { Val v; v.kind = Val::MARK; st.push_back(v); break; }
        case 'X': { need(4); const uint32_t n = rd32(b, at); at += 4; need(n); Val v; v.kind = Val::STR; v.s = b.substr(at, n); at += n; st.push_back(std::move(v)); break; }
        case 0x8c: { need(1); const uint8_t n = (uint8_t)b[at++]; need(n); Val v; v.kind = Val::STR; v.s = b.substr(at, n); at += n; st.push_back(std::move(v)); break; }
        case 'K': { need(1); Val v; v.kind = Val::INT; v.i = (uint8_t)b[at++]; st.push_back(v); break; }
        case 'M': { need(2); Val v; v.kind = Val::INT; v.i = rd16(b, at); at += 2; st.push_back(v); break; }
        case 'J': { need(4); Val v; v.kind = Val::INT; v.i = (int32_t)rd32(b, at); at += 4; st.push_back(v); break; }
        case 0x8a: { need(1); const uint8_t n = (uint8_t)b[at++]; need(n); if (n > 8) PB_FAIL("pickle: integer beyond 64 bits");     // LONG1
            unsigned long long u = 0; for (int k = 0; k < n; ++k) u |= (unsigned long long)(uint8_t)b[at + k] << (8 * k);
            if (n && n < 8 && ((uint8_t)b[at + n - 1] & 0x80)) u |= ~0ull << (8 * n);
            at += n; Val v; v.kind = Val::INT; v.i = (long long)u; st.push_back(v); break; }
        case 't': { Val v; v.kind = Val::TUPLE; v.items = pop_to_mark(); st.push_back(std::move(v)); break; }
        case ')': { Val v; v.kind = Val::TUPLE; st.push_back(v); break; }
        case 0x85: case 0x86: case 0x87: { const int n = op - 0x84; Val v; v.kind = Val::TUPLE; v.items.resize(n); for (int k = n - 1; k >= 0; --k) v.items[k] = pop(); st.push_back(std::move(v)); break; }
        case 0x88: case 0x89: { Val v; v.kind = Val::BOOL; v.i = op == 0x88; st.push_back(v); break; }
        case 'N': { Val v; st.push_back(v); break; }
        case 'Q': { Val t = pop(); st.push_back(storage_of(t)); break; }                                    // BINPERSID
        case 'R': {                                                                                        // REDUCE
            Val args = pop(), fn = pop();
            if (fn.kind != Val::GLOBAL || args.kind != Val::TUPLE) PB_FAIL("pickle: REDUCE of something that is not a known constructor");
            if (fn.s == "collections OrderedDict" && args.items.empty()) { Val v; v.kind = Val::DICT; st.push_back(v); break; }
            if (fn.s == "torch._utils _rebuild_tensor_v2" && args.items.size() >= 4 && args.items[0].kind == Val::STORAGE && args.items[1].kind == Val::INT) {
                Val v = args.items[0]; v.kind = Val::TENSOR; v.offset = args.items[1].i; v.size = ints_of(args.items[2]); v.stride = ints_of(args.items[3]);
                st.push_back(std::move(v)); break;
            }
            PB_FAIL("pickle: %s is not part of a plain tensor file", fn.s.c_str()); }
        case '.': { Val v = pop(); if (v.kind != Val::TENSOR) PB_FAIL("pickle: the file does not hold a single tensor"); return v; }
        default: PB_FAIL("pickle: opcode 0x%02x is not part of a plain tensor file", op);
        }
    }
    PB_FAIL("pickle: too long");
}

// torch.load(path).numpy(), made contiguous
Tensor read_pth(const std::string& path) {
    Tensor out;
    out.file = slurp(path);
    const Buf& b = *out.file;
    const auto members = zip_members(b, path);
    std::string prefix;
    int found = 0;
    for (const auto& kv : members) {
        const std::string& n = kv.first;
        if (n.size() >= 9 && n.compare(n.size() - 9, 9, "/data.pkl") == 0) { prefix = n.substr(0, n.size() - 9); ++found; }
    }
    if (found != 1) PB_FAIL("%s: not a torch zip archive with one data.pkl", path.c_str());
    auto bo = members.find(prefix + "/byteorder");
    if (bo != members.end()) {
        std::string v = b.substr(bo->second.data_at, bo->second.size);
        while (!v.empty() && (v.back() == '\n' || v.back() == ' ')) v.pop_back();
        if (v != "little") PB_FAIL("%s: big-endian storage", path.c_str());
    }
    const Member pk = members.at(prefix + "/data.pkl");
    const Val t = unpickle(b, pk.data_at, pk.data_at + pk.size);
    auto sm = members.find(prefix + "/data/" + t.s);
    if (sm == members.end()) PB_FAIL("%s: storage %s is missing (or compressed)", path.c_str(), t.s.c_str());
    const long long have = (long long)(sm->second.size / (size_t)t.item_size);
    if (t.size.size() != t.stride.size() || t.size.size() > 8 || t.offset < 0) PB_FAIL("%s: bad tensor view", path.c_str());
    out.type = t.type; out.item = t.item_size; out.shape = t.size;
    long long reach = t.offset, n = 1;
    for (size_t k = 0; k < t.size.size(); ++k) {
        if (t.size[k] < 0 || t.stride[k] < 0 || t.size[k] > (1ll << 40) || t.stride[k] > (1ll << 40)) PB_FAIL("%s: tensor view with a bad size / stride", path.c_str());
        if (t.size[k] > 0) reach += (t.size[k] - 1) * t.stride[k];
        n *= t.size[k];
        if (n > (1ll << 40)) PB_FAIL("%s: tensor too large", path.c_str());
    }
    if (n > 0 && reach >= have) PB_FAIL("%s: tensor view reaches beyond its storage", path.c_str());
    const char* src = b.data() + sm->second.data_at;
    out.data = src;
    if (n == 0) return out;
    bool contig = true;
    { long long want = 1; for (size_t k = t.size.size(); k-- > 0;) { if (t.size[k] != 1 && t.stride[k] != want) contig = false; want *= t.size[k]; } }
    if (contig) { out.data = src + (size_t)t.offset * t.item_size; return out; }
    out.own.resize((size_t)n * t.item_size);
    std::vector<long long> idx(t.size.size(), 0);
    for (long long e = 0; e < n; ++e) {
        long long at = t.offset;
        for (size_t k = 0; k < idx.size(); ++k) at += idx[k] * t.stride[k];
        memcpy(&out.own[(size_t)e * t.item_size], src + (size_t)at * t.item_size, t.item_size);
        for (size_t k = idx.size(); k-- > 0;) { if (++idx[k] < t.size[k]) break; idx[k] = 0; }
    }
    out.data = out.own.data();
    return out;
}

// element e of a tensor as a long long / double, whatever it is stored as
inline long long int_at(const Tensor& t, size_t e) {
    const char* p = t.data + e * t.item;
    if (t.type == 'f') { if (t.item == 4) { float v; memcpy(&v, p, 4); return (long long)v; } double v; memcpy(&v, p, 8); return (long long)v; }
    if (t.type == 'u' || t.type == 'b') return (uint8_t)*p;
    switch (t.item) { case 1: return (int8_t)*p; case 2: { int16_t v; memcpy(&v, p, 2); return v; } case 4: { int32_t v; memcpy(&v, p, 4); return v; } default: { long long v; memcpy(&v, p, 8); return v; } }
}

// ndarray.astype(np.int32): wraps like NumPy's cast from int64; *fits = every value in [0, 2^31) (the adjacency's test, same pass)
void as_i32(const Tensor& t, std::vector<int32_t>& out, bool* fits = nullptr) {
    const size_t n = (size_t)t.numel();
    out.resize(n);
    long long lo = 0, hi = 0;
    if (t.type == 'i' && t.item == 8) {                      // the usual case (torch.long): a tight loop, unaligned-safe
        const char* p = t.data;
        for (size_t e = 0; e < n; ++e) { long long v; memcpy(&v, p + 8 * e, 8); lo = std::min(lo, v); hi = std::max(hi, v); out[e] = (int32_t)(uint32_t)(unsigned long long)v; }
    } else if (t.type == 'i' && t.item == 4) {
        if (n) memcpy(out.data(), t.data, n * 4);
        for (size_t e = 0; e < n; ++e) lo = std::min<long long>(lo, out[e]);
    } else {
        for (size_t e = 0; e < n; ++e) { const long long v = int_at(t, e); lo = std::min(lo, v); hi = std::max(hi, v); out[e] = (int32_t)(uint32_t)(unsigned long long)v; }
    }
    if (fits) *fits = lo >= 0 && hi < (1ll << 31);
}

struct Blob { std::string dtype; std::vector<long long> shape; const void* p; size_t bytes; };

void build_one(const char* const* src, const char* name, const char* out_path) {
    for (const char* c = name; *c; ++c)
        if (!((*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z') || (*c >= '0' && *c <= '9') || *c == '_' || *c == '-' || *c == '.'))
            PB_FAIL("scene name %s needs JSON escaping", name);
    // source_files(): pcl, unmap, weak label, seg.json, raw label, adjacency (seggroup_amd/cache.py)
    const Tensor pcl = read_pth(src[0]), unmap_t = read_pth(src[1]), weak = read_pth(src[2]), gt_t = read_pth(src[4]), adj_t = read_pth(src[5]);
    if (pcl.shape.size() != 2 || pcl.shape[1] != 6 || pcl.type != 'f') PB_FAIL("%s: the point cloud is not a float [N,6] tensor", src[0]);
    const long long N = pcl.shape[0];
    if (N <= 0 || N > 0x7fffffffLL) PB_FAIL("%s: bad point count", src[0]);
    if (weak.shape.size() != 2 || weak.shape[0] != N || weak.shape[1] != 2 || weak.type == 'f') PB_FAIL("stage_arrays: inconsistent input shapes");
    // the staging arrays: the thread's own, resized per pack (see Buf above: nothing large is allocated or freed per pack)
    struct Scratch { std::vector<float> data_conv; std::vector<int32_t> seg, order, off, first, counts, adj32, unmap, gt, seg_ins, seg_sem; std::vector<int64_t> adj64; };
    thread_local Scratch t_s;
    std::vector<float>& data_conv = t_s.data_conv;
    const float* data_p = reinterpret_cast<const float*>(pcl.data);                  // float32: written straight from the archive's bytes
    if (pcl.item != 4) {
        data_conv.resize((size_t)N * 6);
        for (size_t e = 0; e < data_conv.size(); ++e) { double v; memcpy(&v, pcl.data + 8 * e, 8); data_conv[e] = (float)v; }
        data_p = data_conv.data();
    }
    std::vector<int32_t>& seg = t_s.seg;
    seg.resize((size_t)N);
    const int S = sg_parse_seg_json(src[3], (int)N, seg.data());
    if (S <= 0) PB_FAIL("%s", sg_last_error());
    std::vector<int32_t>&order = t_s.order, &off = t_s.off, &first = t_s.first, &counts = t_s.counts;
    order.resize((size_t)N); off.resize((size_t)S + 1); first.resize((size_t)S); counts.resize((size_t)S);
    if (sg_stage_segments(seg.data(), (int)N, S, order.data(), off.data(), first.data(), counts.data()) < 0) PB_FAIL("%s", sg_last_error());
    if (adj_t.type == 'f' || adj_t.numel() % 2 != 0) PB_FAIL("%s: the adjacency is not an integer [E,2] tensor", src[5]);
    const long long E0 = adj_t.numel() / 2;
    bool fits = true;
    std::vector<int32_t>& adj32 = t_s.adj32;
    as_i32(adj_t, adj32, &fits);                             // write_pack: point indices that fit are stored as int32
    std::vector<int64_t>& adj64 = t_s.adj64;
    adj64.clear();
    if (!fits) { adj64.resize((size_t)(2 * E0)); for (size_t e = 0; e < adj64.size(); ++e) adj64[e] = int_at(adj_t, e); }
    if (unmap_t.type == 'f') PB_FAIL("%s: the unmapper is not an integer tensor", src[1]);
    std::vector<int32_t>&unmap = t_s.unmap, &gt = t_s.gt;
    as_i32(unmap_t, unmap); as_i32(gt_t, gt);
    const long long V = unmap_t.numel();
    std::vector<int32_t>&seg_ins = t_s.seg_ins, &seg_sem = t_s.seg_sem;
    seg_ins.resize((size_t)S); seg_sem.resize((size_t)S);
    for (int s = 0; s < S; ++s) { seg_ins[s] = (int32_t)int_at(weak, (size_t)first[s] * 2 + 1); seg_sem[s] = (int32_t)int_at(weak, (size_t)first[s] * 2); }

    const Blob blobs[11] = {{"<f4", {N, 6}, data_p, (size_t)N * 24},
                            fits ? Blob{"<i4", {E0, 2}, adj32.data(), adj32.size() * 4} : Blob{"<i8", {E0, 2}, adj64.data(), adj64.size() * 8},
                            {"<i4", {N}, seg.data(), seg.size() * 4}, {"<i4", {N}, order.data(), order.size() * 4}, {"<i4", {(long long)S + 1}, off.data(), off.size() * 4},
                            {"<i4", unmap_t.shape, unmap.data(), unmap.size() * 4}, {"<i4", gt_t.shape, gt.data(), gt.size() * 4},
                            {"<i4", {(long long)S}, first.data(), first.size() * 4}, {"<i4", {(long long)S}, counts.data(), counts.size() * 4},
                            {"<i4", {(long long)S}, seg_ins.data(), seg_ins.size() * 4}, {"<i4", {(long long)S}, seg_sem.data(), seg_sem.size() * 4}};
    static const char* const names[11] = {"data", "adj", "seg_of_point", "seg_points", "seg_off", "unmap", "gt", "seg_first", "seg_size", "seg_ins", "seg_sem"};
    // json.dumps of cache.write_pack's header, character for character
    std::string hdr = std::string("{\"name\": \"") + name + "\", \"N\": " + std::to_string(N) + ", \"S\": " + std::to_string(S) + ", \"E0\": " + std::to_string(E0) +
                      ", \"V\": " + std::to_string(V) + ", \"arrays\": {";
    size_t at = 0;
    for (int k = 0; k < 11; ++k) {
        hdr += std::string(k ? ", " : "") + "\"" + names[k] + "\": [\"" + blobs[k].dtype + "\", [";
        for (size_t d = 0; d < blobs[k].shape.size(); ++d) hdr += (d ? ", " : "") + std::to_string(blobs[k].shape[d]);
        hdr += "], " + std::to_string(at) + "]";
        at += (blobs[k].bytes + 63) / 64 * 64;
    }
    hdr += "}}";
    const size_t pad = (64 - (8 + 4 + hdr.size()) % 64) % 64;
    hdr.append(pad, ' ');
    const std::string tmp = std::string(out_path) + ".tmp" + std::to_string((long long)getpid()) + "." + std::to_string(std::hash<std::thread::id>()(std::this_thread::get_id()) % 1000000);
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) PB_FAIL("cannot create %s: %s", tmp.c_str(), strerror(errno));
    static const char zeros[64] = {0};
    const uint32_t hlen = (uint32_t)hdr.size();
    bool ok = fwrite("SGPACK01", 1, 8, f) == 8 && fwrite(&hlen, 4, 1, f) == 1 && fwrite(hdr.data(), 1, hdr.size(), f) == hdr.size();
    for (int k = 0; k < 11 && ok; ++k) {
        ok = blobs[k].bytes == 0 || fwrite(blobs[k].p, 1, blobs[k].bytes, f) == blobs[k].bytes;
        const size_t z = (64 - blobs[k].bytes % 64) % 64;
        ok = ok && (z == 0 || fwrite(zeros, 1, z, f) == z);
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), out_path) != 0) { remove(tmp.c_str()); PB_FAIL("cannot write %s: %s", out_path, strerror(errno)); }
}

}  // namespace

extern "C" {

int sg_pack_build(const char* const* src_paths6, const char* name, const char* out_path) {
    if (!src_paths6 || !name || !out_path) return sg::fail(SG_EINVAL, "sg_pack_build: bad arguments");
    for (int k = 0; k < 6; ++k) if (!src_paths6[k]) return sg::fail(SG_EINVAL, "sg_pack_build: bad arguments");
    try { build_one(src_paths6, name, out_path); }
    catch (const Err& e) { return sg::fail(SG_EINVAL, "sg_pack_build: %s", e.msg.c_str()); }
    catch (const std::exception& e) { return sg::fail(SG_EINVAL, "sg_pack_build: %s", e.what()); }
    return SG_OK;
}

int sg_pack_build_many(const char* const* src_paths, const char* const* names, const char* const* out_paths, int n, int threads, int32_t* h_status) {
    if (n < 0 || (n > 0 && (!src_paths || !names || !out_paths || !h_status))) return sg::fail(SG_EINVAL, "sg_pack_build_many: bad arguments");
    std::atomic<int> next{0}, built{0};
    std::mutex mu;
    std::string first_err;
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            const int rc = sg_pack_build(src_paths + 6 * (size_t)i, names[i], out_paths[i]);
            h_status[i] = rc;
            if (rc == SG_OK) ++built;
            else { std::lock_guard<std::mutex> g(mu); if (first_err.empty()) first_err = sg_last_error(); }
        }
    };
    const int T = std::max(1, std::min(threads, n));
    std::vector<std::thread> pool;
    for (int t = 1; t < T; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (!first_err.empty()) snprintf(sg::err_buf(), 512, "%s", first_err.c_str());      // the caller reads it when a status is negative
    return built.load();
}

}  // extern "C"
