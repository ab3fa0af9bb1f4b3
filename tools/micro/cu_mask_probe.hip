// Where do the bits of a HIP CU mask land on an MI355X?  (round 6)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cu_mask_probe.hip -o build_micro/cu_mask_probe && build_micro/cu_mask_probe
// A grid of 4,096 one-wave workgroups that each spin ~20 us is launched on streams made by hipExtStreamCreateWithCUMask with a few masks;
// every workgroup records hwreg(HW_REG_XCC_ID) and the SE / SH / CU fields of hwreg(HW_REG_HW_ID).  Printed per mask: how many distinct
// (XCD, SE, CU) places ran a workgroup, per XCD, and the launch's duration against the unmasked one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#include <map>

__global__ void k_where(uint32_t* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = clock64();
    while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(const char* name, const uint32_t* mask) {
    hipStream_t st;
    if (mask) { if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed\n", name); return; } }
    else (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int nb = 4096;
    uint32_t* d; (void)hipMalloc(&d, nb * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_where<<<nb, 64, 0, st>>>(d, 2000);                      // warm
    (void)hipEventRecord(a, st);
    k_where<<<nb, 64, 0, st>>>(d, 2000);                      // 2,000 ticks of the 100 MHz counter = 20 us
    (void)hipEventRecord(b, st);
    (void)hipStreamSynchronize(st);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<uint32_t> h(2 * nb);
    (void)hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<int, std::set<int>> per_xcc;
    for (int i = 0; i < nb; ++i) {
        const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 15;
        const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc].insert(se * 32 + sh * 16 + cu);
    }
    int total = 0;
    printf("%-34s %7.3f ms |", name, ms);
    for (auto& kv : per_xcc) { printf(" x%d:%2zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf(" | %d CUs\n", total);
    if (mask && total <= 40) {
        for (auto& kv : per_xcc) { printf("      xcd %d (se.cu):", kv.first); for (int v : kv.second) printf(" %d.%d", v / 32, v % 16); printf("\n"); }
    }
    (void)hipFree(d); (void)hipStreamDestroy(st);
}

int main() {
    uint32_t m[8];
    run("no mask", nullptr);
    for (int i = 0; i < 8; ++i) m[i] = ~0u;
    run("all 256 bits", m);
    for (int i = 0; i < 8; ++i) m[i] = i < 7 ? ~0u : 0u;
    run("bits [0,224)", m);
    for (int i = 0; i < 8; ++i) m[i] = i < 4 ? ~0u : 0u;
    run("bits [0,128)", m);
    for (int i = 0; i < 8; ++i) m[i] = i == 0 ? ~0u : 0u;
    run("bits [0,32)", m);
    for (int i = 0; i < 8; ++i) m[i] = i == 7 ? ~0u : 0u;
    run("bits [224,256)", m);
    for (int i = 0; i < 8; ++i) m[i] = i == 0 ? 0xFFu : 0u;
    run("bits [0,8)", m);
    for (int i = 0; i < 8; ++i) m[i] = 0x01010101u;
    run("bits i % 8 == 0", m);
    for (int i = 0; i < 8; ++i) m[i] = 0x0F0F0F0Fu;
    run("bits i % 8 < 4", m);
    return 0;
}
