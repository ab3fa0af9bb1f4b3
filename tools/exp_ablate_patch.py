#!/usr/bin/env python3
"""Development only: patches seggroup_amd/csrc/kernels_edgeconv.hip IN PLACE so that the high bits of SG_EC_STAGGER2 (value = flags << 8)
leave parts of the S2X slot out -- 1 conv2's MFMAs, 2 the fp16 cut, 4 the statistics, 8 conv1's MFMAs, 16 LeakyReLU, 32 the neighbour gathers
(results are garbage; the launch time is what is measured: tools/exp_ablate.sh).  The branches disturb the instruction schedule (the patched
kernel is ~40 % slower with no flag set), so only differences between flag sets mean something.  Undo: git checkout seggroup_amd/csrc/kernels_edgeconv.hip.

Written against the slot loop of commit 55849f3 (MLP3's conv1 still on three bf16 pieces, full-row loads): the attribution quoted in DESIGN.md
section 5 was measured there.  Against a later kernel the script stops with "anchor not found" and changes nothing -- check that commit out to
repeat the measurement."""
import os
import sys
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "seggroup_amd", "csrc", "kernels_edgeconv.hip")
s = open(p).read()


def rep(old, new):
    global s
    if old not in s:
        sys.exit("anchor not found (the kernel changed): " + old[:70])
    s = s.replace(old, new, 1)


rep("""    if (stagger > 0 && (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 1)) {""",
    """    const int ablate = stagger >> 8;
    stagger &= 255;
    if (stagger > 0 && (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 1)) {""")
rep("""            if (j + 1 < K) {
                const gptr<const float4> xq""", """            if (j + 1 < K && !(ablate & 32)) {
                const gptr<const float4> xq""")
rep("""                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa2, x2, base[0], 0, 0, 0);""",
    """                if (ablate & 8) { acc1[0] = base[0]; acc1[1] = base[1]; acc1[0][0] += __uint_as_float(a1 ^ b2 ^ a3 ^ b3 ^ c1 ^ c2 ^ c3 ^ b1 ^ a2) * 1e-30f; }
                else {
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa2, x2, base[0], 0, 0, 0);""")
rep("""                acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb0, x0, acc1[1], 0, 0, 0);
            }""", """                acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb0, x0, acc1[1], 0, 0, 0);
                }
            }""")
rep("""#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        const f32x2 v = {acc1[t][q], acc1[t][q + 1]};""", """                if (!(ablate & 16))
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        const f32x2 v = {acc1[t][q], acc1[t][q + 1]};""")
rep("""                        const f16x2 hi = __builtin_convertvector(v, f16x2);
                        const f16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x2), f16x2);     // v - hi is exact in fp32
                        xh[kb][jj]""", """                        if (ablate & 2) { xh[kb][jj] = __float_as_uint(v.x); xl[kb][jj] = __float_as_uint(v.y); continue; }
                        const f16x2 hi = __builtin_convertvector(v, f16x2);
                        const f16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x2), f16x2);     // v - hi is exact in fp32
                        xh[kb][jj]""")
rep("""                    // smallest terms first
                    acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa2, x1, acc2[0], 0, 0, 0);""",
    """                    // smallest terms first
                    if (ablate & 1) { acc2[0][kb] += __uint_as_float(xh[kb][0] ^ xl[kb][1] ^ xh[kb][2] ^ xl[kb][3] ^ xh[kb][1] ^ xl[kb][0] ^ xh[kb][3] ^ xl[kb][2]) * 1e-30f; continue; }
                    acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa2, x1, acc2[0], 0, 0, 0);""")
rep("""                        const float z = acc2[ot][q];
                        stat_s[16 * ot + q] += z;""", """                        const float z = acc2[ot][q];
                        if (ablate & 4) { if (q == 0) best[ot][0] += z; continue; }
                        stat_s[16 * ot + q] += z;""")
open(p, "w").write(s)
print("patched", p)
