// Device -> pinned-host copies on the COPY ENGINES (SDMA), issued through the HSA runtime the process's HIP runtime sits on (round 6).
//
// `hipMemcpyAsync(device -> pinned host)` on this ROCm is executed by a blit KERNEL (`__amd_rocclr_copyBuffer` in every kernel trace of the bench:
// two per scene, ~150 us each for a scene's 8.4 MB of label vectors at PCIe speed) -- waves that sit on the CUs waiting for PCIe writes.  With the
// engine GPU-bound that is not free: the bench with only the [14,S] tables crossing PCIe runs 6-7 % faster than with the full vectors (3,455-3,486
// against 3,204-3,325 scenes/s).  The copy engines move the same bytes without a wave: hsa_amd_memory_async_copy, agents taken
// from hsa_amd_pointer_info of the two pointers, one completion signal per copy, the caller's thread waits for them.  No link-time dependency: the
// symbols are looked up in the already-loaded runtime; if anything is missing or any call fails, the caller falls back to hipMemcpyAsync (and this
// path stays off for the rest of the process).
#include <atomic>
#include <dlfcn.h>
#include <mutex>

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include "sg_common.h"

namespace sg {

namespace {

struct HsaApi {
    decltype(&hsa_init) init = nullptr;
    decltype(&hsa_amd_pointer_info) pointer_info = nullptr;
    decltype(&hsa_signal_create) signal_create = nullptr;
    decltype(&hsa_signal_destroy) signal_destroy = nullptr;
    decltype(&hsa_signal_store_relaxed) signal_store = nullptr;
    decltype(&hsa_signal_wait_scacquire) signal_wait = nullptr;
    decltype(&hsa_amd_memory_async_copy) async_copy = nullptr;
    bool ok = false;
    HsaApi() {
        void* h = RTLD_DEFAULT;
        auto sym = [&](const char* n) -> void* {
            void* p = dlsym(h, n);
            if (!p) {
                for (const char* lib : {"libhsa-runtime64.so.1", "libhsa-runtime64.so"}) {
                    void* hl = dlopen(lib, RTLD_NOW | RTLD_NOLOAD);
                    if (hl && (p = dlsym(hl, n)) != nullptr) break;
                }
            }
            return p;
        };
        init = reinterpret_cast<decltype(init)>(sym("hsa_init"));
        pointer_info = reinterpret_cast<decltype(pointer_info)>(sym("hsa_amd_pointer_info"));
        signal_create = reinterpret_cast<decltype(signal_create)>(sym("hsa_signal_create"));
        signal_destroy = reinterpret_cast<decltype(signal_destroy)>(sym("hsa_signal_destroy"));
        signal_store = reinterpret_cast<decltype(signal_store)>(sym("hsa_signal_store_relaxed"));
        signal_wait = reinterpret_cast<decltype(signal_wait)>(sym("hsa_signal_wait_scacquire"));
        async_copy = reinterpret_cast<decltype(async_copy)>(sym("hsa_amd_memory_async_copy"));
        ok = init && pointer_info && signal_create && signal_destroy && signal_store && signal_wait && async_copy && init() == HSA_STATUS_SUCCESS;
    }
};
const HsaApi& api() { static const HsaApi a; return a; }
std::atomic<bool> g_disabled{false};

struct SignalPool {                       // per thread: a group thread reuses its signals for every super-step
    std::vector<hsa_signal_t> s;
    ~SignalPool() { for (hsa_signal_t x : s) (void)api().signal_destroy(x); }
};

}  // namespace

bool sdma_available() { return !g_disabled.load(std::memory_order_relaxed) && api().ok; }

namespace {
// the two agents of a copy; false when either pointer is not memory the HSA runtime allocated or locked (a pageable buffer: not an error of the path)
bool agents_of(const HsaApi& a, void* dst, const void* src, hsa_agent_t* da, hsa_agent_t* sa) {
    hsa_amd_pointer_info_t ps, pd;
    ps.size = sizeof ps; pd.size = sizeof pd;
    if (a.pointer_info(const_cast<void*>(src), &ps, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS ||
        a.pointer_info(dst, &pd, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return false;
    auto known = [](const hsa_amd_pointer_info_t& p) { return p.type == HSA_EXT_POINTER_TYPE_HSA || p.type == HSA_EXT_POINTER_TYPE_LOCKED; };
    if (!known(ps) || !known(pd)) return false;
    *da = pd.agentOwner; *sa = ps.agentOwner;
    return true;
}
}  // namespace

// ONE copy between device memory and pinned host memory (either direction), issued now; sdma_wait blocks until it is done.  The ticket owns a signal.
int sdma_issue(void* dst, const void* src, size_t bytes, SdmaTicket* t) {
    t->signal = 0;
    if (bytes == 0) return SG_OK;
    if (!sdma_available()) return sg::fail(SG_EUNSUP, "sdma: the HSA runtime's copy interface is not available");
    const HsaApi& a = api();
    hsa_agent_t da, sa;
    if (!agents_of(a, dst, src, &da, &sa)) return sg::fail(SG_EUNSUP, "sdma: a pointer the HSA runtime does not know (host side not pinned?)");
    hsa_signal_t sig;
    if (a.signal_create(1, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) { g_disabled = true; return sg::fail(SG_EHIP, "sdma: hsa_signal_create failed"); }
    if (a.async_copy(dst, da, src, sa, bytes, 0, nullptr, sig) != HSA_STATUS_SUCCESS) {
        (void)a.signal_destroy(sig);
        g_disabled = true;
        return sg::fail(SG_EHIP, "sdma: hsa_amd_memory_async_copy failed");
    }
    t->signal = sig.handle;
    return SG_OK;
}

int sdma_wait(SdmaTicket* t) {
    if (!t->signal) return SG_OK;
    const HsaApi& a = api();
    hsa_signal_t sig; sig.handle = t->signal;
    const hsa_signal_value_t v = a.signal_wait(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
    (void)a.signal_destroy(sig);
    t->signal = 0;
    if (v < 0) { g_disabled = true; return sg::fail(SG_EHIP, "sdma: a copy reported an error"); }
    return SG_OK;
}

// n copies (device -> pinned host in the engine; any direction works), all issued, then all waited for.  SG_OK, or a negative code with NOTHING guaranteed about the destinations
// (the caller repeats the copies its own way).
int sdma_copy_d2h(void* const* dst, const void* const* src, const size_t* bytes, int n) {
    if (n <= 0) return SG_OK;
    if (!sdma_available()) return sg::fail(SG_EUNSUP, "sdma: the HSA runtime's copy interface is not available");
    const HsaApi& a = api();
    static thread_local SignalPool pool;
    while ((int)pool.s.size() < n) {
        hsa_signal_t sig;
        if (a.signal_create(1, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) { g_disabled = true; return sg::fail(SG_EHIP, "sdma: hsa_signal_create failed"); }
        pool.s.push_back(sig);
    }
    int issued = 0;
    int rc = SG_OK;
    for (int i = 0; i < n && rc == SG_OK; ++i) {
        if (bytes[i] == 0) continue;
        hsa_agent_t da, sa;
        if (!agents_of(a, dst[i], src[i], &da, &sa)) {
            rc = sg::fail(SG_EUNSUP, "sdma: a pointer the HSA runtime does not know (destination not pinned?)");      // this call only: the path stays on
            break;
        }
        a.signal_store(pool.s[i], 1);
        if (a.async_copy(dst[i], da, src[i], sa, bytes[i], 0, nullptr, pool.s[i]) != HSA_STATUS_SUCCESS) {
            rc = sg::fail(SG_EHIP, "sdma: hsa_amd_memory_async_copy failed");
            g_disabled = true;
            break;
        }
        issued = i + 1;
    }
    for (int i = 0; i < issued; ++i) {
        if (bytes[i] == 0) continue;
        const hsa_signal_value_t v = a.signal_wait(pool.s[i], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
        if (v < 0 && rc == SG_OK) { rc = sg::fail(SG_EHIP, "sdma: a copy reported an error"); g_disabled = true; }
    }
    return rc;
}

}  // namespace sg
