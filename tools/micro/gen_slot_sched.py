#!/usr/bin/env python3
"""Generates tools/micro/slot_sched.hip: what does a hand-placed EdgeConv neighbour slot cost on one SIMD?

A "slot" of k_edgeconv<S2X> is 28 v_mfma_f32_32x32x16_f16 (two accumulator chains) and ~230-330 VALU instructions.  The kernels written
here run that mix from ONE or TWO waves per SIMD with the VALU instructions placed by hand:
  f fillers behind every MFMA (in its 32-cycle shadow), the rest of the slot's VALU budget in one lump behind the last MFMA.
Every pattern is one asm statement (the compiler neither reorders nor pads it); fillers are independent of the MFMAs and of each other
(8 registers in rotation), so the numbers are issue / pipe costs, not dependency stalls.

    python3 tools/micro/gen_slot_sched.py > tools/micro/slot_sched.hip
    hipcc --offload-arch=gfx950 -O3 -w tools/micro/slot_sched.hip -o build_micro/slot_sched && build_micro/slot_sched
"""
import sys

NM = 28
FILL = {
    # name -> list of instruction templates over the rotating registers (%{i} = one of a0..a7, b = %8, c = %9)
    "fma": ["v_fma_f32 {a}, {a}, %12, %13"],
    "max": ["v_max_f32 {a}, {a}, %12"],
    "mix": ["v_cvt_pk_f16_f32 {a}, {a}, %12", "v_max_f32 {a}, {a}, %12", "v_cvt_f32_f16 {a}, {a}", "v_max_f32 {a}, {a}, %13",
            "v_add_f32 {a}, {a}, %12", "v_fma_f32 {a}, {a}, %12, %13", "v_sub_f32 {a}, {a}, %13", "v_max_f32 {a}, {a}, %13"],
    "pkfma": ["v_pk_fma_f32 {p}, {p}, {q}, {q}"],
    "pkadd": ["v_pk_add_f32 {p}, {p}, {q}"],
    "pkmul": ["v_pk_mul_f32 {p}, {p}, {q}"],
    "fmamix": ["v_fma_mix_f32 {a}, {a}, %12, %13 op_sel_hi:[1,0,0]"],
    "cvtpk": ["v_cvt_pk_f16_f32 {a}, {a}, %12"],
}


def filler(kind, n, start):
    out = []
    t = FILL[kind]
    for i in range(n):
        k = start + i
        a = "%%%d" % (k % 8)
        ins = t[k % len(t)]
        # packed forms work on the pairs (%10, %11): two 64-bit registers in rotation
        ins = ins.format(a=a, p="%%%d" % (8 + k % 2), q="%14")
        out.append(ins)
    return out


def slot(f, total, kind, chains=2):
    """28 MFMAs, f fillers behind each, then (total - 28 f) in a lump."""
    lines = []
    used = 0
    for m in range(NM):
        acc = "%%%d" % (10 + m % chains)
        lines.append("v_mfma_f32_32x32x16_f16 {0}, %15, %15, {0}".format(acc))
        n = min(f, max(total - used, 0))
        lines += filler(kind, n, used)
        used += n
    lines += filler(kind, max(total - used, 0), used)
    return lines


def kernel_rot(name, agpr):
    """28 MFMAs, two chains, A and B operands rotating over 8 different register tuples (A optionally in AGPRs)"""
    lines = []
    for m in range(NM):
        acc = "%%%d" % (m % 2)
        a = ("a[%d:%d]" if agpr else "v[%d:%d]") % ((100 if not agpr else 0) + 4 * (m % 8), (100 if not agpr else 0) + 4 * (m % 8) + 3)
        b = "v[%d:%d]" % (140 + 4 * ((m * 3) % 8), 140 + 4 * ((m * 3) % 8) + 3)
        lines.append("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (acc, a, b, acc))
    body = "\\n\"\n        \"".join(lines)
    init = "".join("v_accvgpr_write_b32 a%d, %%2\\n" % i for i in range(32)) if agpr else "".join("v_mov_b32 v%d, %%2\\n" % (100 + i) for i in range(32))
    init += "".join("v_mov_b32 v%d, %%2\\n" % (140 + i) for i in range(32))
    clob = ", ".join(['"v%d"' % i for i in range(100, 172)] + (['"a%d"' % i for i in range(32)] if agpr else []))
    return """
__global__ __launch_bounds__(512) void %s(float* out, int iters) {
    extern __shared__ float dyn[];
    f32x16 x = {0}, y = {0};
    const unsigned one = 0x3c003c00u;
    asm volatile("%s" : "+v"(x), "+v"(y) : "v"(one) : %s);
    for (int i = 0; i < iters; ++i)
        asm volatile("%s" : "+v"(x), "+v"(y) : "v"(one) : %s);
    if (iters < 0) dyn[threadIdx.x] = 1.f;
    out[blockIdx.x * 512 + threadIdx.x] = x[0] + y[1];
}
""" % (name, init, clob, body, clob)


def kernel(name, lines):
    body = "\\n\"\n        \"".join(lines)
    return """
__global__ __launch_bounds__(512) void %s(float* out, int iters) {
    extern __shared__ float dyn[];
    float a[8] = {(float)threadIdx.x, 1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f};
    const float b = 1.0001f, c = 0.5f;
    f32x2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, q = {1.0001f, 0.9999f};
    f32x16 x = {0}, y = {0};
    const f16x8 fa = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int i = 0; i < iters; ++i)
        asm volatile("%s"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(p0), "+v"(p1), "+v"(x), "+v"(y)
                     : "v"(b), "v"(c), "v"(q), "v"(fa));
    if (iters < 0) dyn[threadIdx.x] = 1.f;
    out[blockIdx.x * 512 + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7] + x[0] + y[1] + p0.x + p1.y;
}
""" % (name, body)


def main():
    ks = []      # (name, description, total VALU)
    src = []
    def add(name, desc, lines, nv):
        ks.append((name, desc, nv))
        src.append(kernel(name, lines))
    add("k_m_only", "28 MFMA only", slot(0, 0, "fma"), 0)
    add("k_m_only1", "28 MFMA only, one chain", slot(0, 0, "fma", chains=1), 0)
    for f in (1, 2, 3, 4, 5, 6, 7, 8):
        add("k_fma_f%d" % f, "v_fma_f32 x %d behind every MFMA" % f, slot(f, NM * f, "fma"), NM * f)
    for f in (4, 5, 6, 7):
        add("k_mix_f%d" % f, "cvt/max/add mix x %d behind every MFMA" % f, slot(f, NM * f, "mix"), NM * f)
    for total in (140, 230, 330):
        add("k_lump_%d" % total, "28 MFMA, then %d mix in one lump" % total, slot(0, total, "mix"), total)
        for f in (4, 5, 6):
            if NM * f < total:
                add("k_mix_f%d_t%d" % (f, total), "mix: %d behind every MFMA, rest of %d in a lump" % (f, total), slot(f, total, "mix"), total)
    for kind in ("pkfma", "pkadd", "pkmul", "fmamix", "cvtpk", "max"):
        for f in (2, 4):
            add("k_%s_f%d" % (kind, f), "%s x %d behind every MFMA" % (kind, f), slot(f, NM * f, kind), NM * f)
    ks.append(("k_rot_vgpr", "28 MFMA, A/B operands rotating over 8 VGPR tuples", 0)); src.append(kernel_rot("k_rot_vgpr", False))
    ks.append(("k_rot_agpr", "28 MFMA, A in AGPRs (8 tuples), B rotating", 0)); src.append(kernel_rot("k_rot_agpr", True))
    for kind in ("fma", "mix", "pkfma", "pkadd", "fmamix", "cvtpk"):
        lines = filler(kind, 224, 0)
        add("k_v_only_%s" % kind, "224 %s, no MFMA" % kind, lines, 224)

    print("// GENERATED by tools/micro/gen_slot_sched.py -- do not edit.  See that file for what is measured.")
    print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstring>")
    print("using f32x16 = __attribute__((ext_vector_type(16))) float;\nusing f32x2 = __attribute__((ext_vector_type(2))) float;")
    print("using f16x8 = __attribute__((ext_vector_type(8))) _Float16;")
    print("".join(src))
    print("""
typedef void (*kern_t)(float*, int);
struct Entry { const char* name; const char* desc; kern_t fn; int nv; };
static Entry entries[] = {""")
    for name, desc, nv in ks:
        print('    {"%s", "%s", %s, %d},' % (name, desc, name, nv))
    print("""};
static double run(kern_t fn, float* d, int cus, int threads, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 96 * 1024;                        // one workgroup per CU
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fn, dim3(cus), dim3(threads), lds, 0, d, 10); hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(fn, dim3(cus), dim3(threads), lds, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e6 / iters;                           // ns per slot (per wave: the waves of a SIMD run their slots side by side)
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d; hipMalloc(&d, (size_t)cus * 512 * 4);
    const int iters = 3000;
    const double m1 = run(entries[0].fn, d, cus, 256, iters);
    printf("slot = 28 x v_mfma_f32_32x32x16_f16 (+ VALU); ns per slot and SIMD; cycles at the clock that makes MFMA-only = 32 cycles per MFMA (%.3f GHz)\\n", 28 * 32 / m1);
    printf("%-22s %-52s %6s | %9s %8s %8s | %9s %8s %8s\\n", "kernel", "pattern", "VALU", "1w ns", "cyc", "cyc/MFMA", "2w ns/2", "cyc", "vs 1w");
    for (auto& e : entries) {
        const double t1 = run(e.fn, d, cus, 256, iters);           // one wave per SIMD
        const double t2 = run(e.fn, d, cus, 512, iters) / 2;       // two waves per SIMD: per slot of either wave
        const double ghz = 28 * 32 / m1;
        printf("%-22s %-52s %6d | %9.1f %8.0f %8.1f | %9.1f %8.0f %8.2f\\n", e.name, e.desc, e.nv, t1, t1 * ghz, t1 * ghz / 28, t2, t2 * ghz, t2 / t1);
    }
    return 0;
}""")


if __name__ == "__main__":
    main()
