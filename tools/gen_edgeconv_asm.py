#!/usr/bin/env python3
"""Writes seggroup_amd/csrc/edgeconv_slots_gen.h: the hand-scheduled neighbour-slot loops of the EdgeConv kernels (gfx950).

Why: k_edgeconv<S2X> ran 28 MFMAs (896 cycles of matrix pipe) and ~330 VALU instructions per neighbour slot in ~2,400 cycles per SIMD --
the compiler's schedule puts the VALU phases (operand cut, LeakyReLU, statistics) BETWEEN the MFMA bursts, and the two add up.
tools/micro/slot_sched.hip (round 4) shows what the hardware allows: up to SIX independent VALU instructions placed behind every
v_mfma_f32_32x32x16_f16 cost nothing (28 MFMAs + 168 VALU = 393 ns, MFMAs alone 387 ns), from one wave per SIMD as well as from two;
packed fp32 VALU (v_pk_*) beside MFMAs is the exception (+4 ns each).  So the slot loop is written here as ONE asm statement per tile:
the 20 slots fully unrolled, software-pipelined at stage granularity by the order the stages are listed in (`program_*`), and
interleaved instruction by instruction by a small list scheduler (`schedule`) that
  * keeps the MFMAs in program order and puts up to FILL other instructions behind each,
  * derives every dependency from the registers an instruction reads and writes (RAW / WAR / WAW on physical registers),
  * inserts the s_waitcnt vmcnt / lgkmcnt each consumer of a load needs (loads return in order: the count is the number of younger
    loads of the class already issued) and the wait states the hardware does not interlock (the numbers hipcc itself pads with):
        MFMA result -> VALU / LDS / VMEM read or overwrite   12 states      VALU write -> MFMA operand   2 states
        MFMA source -> overwritten by VALU / a load           10 states
The arithmetic, its order and therefore every result bit are those of edgeconv_body's C++ slot loop (kernels_edgeconv.hip), which
stays in the library as the K != 20 path and as the cross-check (tests/test_gpu_ops.py compares the two bit for bit).

    python3 tools/gen_edgeconv_asm.py            # rewrites the header; `make` does not run it (the header is committed)
"""
import os
import sys
from collections import defaultdict

K = 20
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("SG_EC_OUT") or os.path.join(HERE, "..", "seggroup_amd", "csrc", "edgeconv_slots_gen.h")
# The SG_EC_* experiment knobs below (packed statistics, other register files, omitted stages ...) change the emitted code: they are read ONLY
# when the generator is started with --experimental (tools/build_ec_variant.sh does), so that a stray variable in the environment of a build --
# the Makefile regenerates the header when this file is newer -- cannot put packed fp32 back into the library.  SG_EC_OUT
# (where to write) is not an experiment.
EXPERIMENTAL = "--experimental" in sys.argv[1:]


def knob(name, default=""):
    return os.environ.get(name, default) if EXPERIMENTAL else default


MFMA_TO_READ = 14      # instructions between an MFMA and a non-MFMA reader / overwriter of its D (hipcc: s_nop 11 = 12 states; + margin)
VALU_TO_MFMA = 3       # instructions between a VALU write and the MFMA reading it (hipcc: s_nop 1 = 2 states)
MFMA_SRC_WAR = 12      # instructions between an MFMA and an overwrite of its C operand (read pass by pass)
MFMA_AB_WAR = 2        # ... of its A / B operand (read when the MFMA issues: hipcc pads nothing here)


class Op:
    __slots__ = ("kind", "text", "reads", "writes", "cls", "tag", "idx", "deps", "fill", "frag", "cregs")

    def __init__(self, kind, text, reads=(), writes=(), cls=None, tag="", fill=True):
        self.kind = kind            # mfma | valu | lds | vmem | salu
        self.text = text
        self.reads = tuple(reads)
        self.writes = tuple(writes)
        self.cls = cls              # 'vm' | 'lgkm' for loads (asynchronous register writes)
        self.tag = tag
        self.fill = fill
        self.frag = None
        self.cregs = frozenset()


def vr(n, cnt=1):
    return ["v%d" % (n + i) for i in range(cnt)]


def ar(n, cnt=1):
    if knob("SG_EC_AVGPR"):
        return ["v%d" % (12 + n + i) for i in range(cnt)]
    return ["a%d" % (n + i) for i in range(cnt)]


def vt(n, cnt):
    assert cnt == 1 or n % 2 == 0, "VGPR tuples must be 64-bit aligned: v%d x %d" % (n, cnt)
    return "v[%d:%d]" % (n, n + cnt - 1) if cnt > 1 else "v%d" % n


A_IN_VGPR = bool(knob("SG_EC_AVGPR"))       # timing experiment (with SG_EC_OMIT=L,S,...): the A fragments in v[12:91] instead of AGPRs


def at(n, cnt):
    if A_IN_VGPR:
        return "v[%d:%d]" % (12 + n, 12 + n + cnt - 1)
    return "a[%d:%d]" % (n, n + cnt - 1)


class Prog:
    def __init__(self):
        self.ops = []

    def add(self, *a, **k):
        self.ops.append(Op(*a, **k))

    # ---- instruction helpers (register numbers are physical VGPRs unless the operand is a string such as "%[xs0]") ----
    def valu(self, text, reads, writes, tag=""):
        self.add("valu", text, reads, writes, tag=tag)

    def mfma(self, d, a_txt, a_regs, b, c, tag=""):
        """v_mfma_f32_32x32x16_f16 v[d:d+15], A, v[b:b+3], C    (C = None: inline 0)"""
        ctxt = "0" if c is None else vt(c, 16)
        reads = list(a_regs) + vr(b, 4) + ([] if c is None else vr(c, 16))
        self.add("mfma", "v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (vt(d, 16), a_txt, vt(b, 4), ctxt), reads, vr(d, 16), tag=tag)
        self.ops[-1].cregs = frozenset([] if c is None else vr(c, 16))

    def mfma_frag(self, d, off, b, c, tag=""):
        """the same with the A fragment at LDS offset `off` taken from the ring (assign_frag_ring fills in the registers)"""
        self.mfma(d, "@A@", [], b, c, tag)
        self.ops[-1].frag = FragUse(off)


class FragUse:
    """placeholder in an MFMA's A operand: the fragment at LDS byte offset `off` (from %[frag]), to be found in a ring slot"""
    def __init__(self, off):
        self.off = off


def assign_frag_ring(ops, ring):
    """A fragments live in LDS and pass through `ring` (VGPR tuples of 4): every MFMA built with a FragUse gets the slot its fragment
    sits in; a fragment that is not resident is loaded into the slot that was read longest ago, and the load is listed right behind
    that read -- as early as the registers allow, so the scheduler can issue it that many MFMAs ahead (the 20 fragments of a slot come
    round every 28 MFMAs: whatever is evicted misses next time anyway, so the oldest slot gives the longest run-up)."""
    uses = [(i, o.frag.off) for i, o in enumerate(ops) if getattr(o, "frag", None) is not None]
    nxt = {}                                             # position in `uses` -> position of the next use of the same fragment
    last = {}
    for k in range(len(uses) - 1, -1, -1):
        nxt[k] = last.get(uses[k][1], 1 << 30)
        last[uses[k][1]] = k
    slot_frag = [None] * len(ring)
    slot_next = [(-1)] * len(ring)                       # next use (position in `uses`) of the slot's content; -1 = empty
    slot_last_op = [-1] * len(ring)                      # op index of the last MFMA that read the slot
    inserts = []                                         # (after op index, Op)
    for k, (i, off) in enumerate(uses):
        op = ops[i]
        if off in slot_frag:
            sl = slot_frag.index(off)
        else:
            sl = min(range(len(ring)), key=lambda q: slot_last_op[q])
            ld = Op("lds", "ds_read_b128 %s, %%[frag] offset:%d" % (vt(ring[sl], 4), off), ["%[frag]"], vr(ring[sl], 4), cls="lgkm", tag="F")
            inserts.append((slot_last_op[sl], ld))
            slot_frag[sl] = off
        slot_next[sl] = nxt[k]
        slot_last_op[sl] = i
        op.text = op.text.replace("@A@", vt(ring[sl], 4))
        op.reads = tuple(list(op.reads) + vr(ring[sl], 4))
    out = []
    by_pos = defaultdict(list)
    for pos, ld in inserts:
        by_pos[pos].append(ld)
    out.extend(by_pos.get(-1, []))
    for i, o in enumerate(ops):
        out.append(o)
        out.extend(by_pos.get(i, []))
    return out


def operand(x):
    return x if isinstance(x, str) else "v%d" % x


def rd(x):
    return [] if (isinstance(x, str) and not x.startswith("v")) else ([x] if isinstance(x, str) else ["v%d" % x])


# ------------------------------------------------------------------------------------------------------------------------------
# the scheduler
# ------------------------------------------------------------------------------------------------------------------------------
def build_deps(ops):
    last_write = {}
    readers = defaultdict(list)
    for i, op in enumerate(ops):
        op.idx = i
        deps = set()
        for r in op.reads:
            if r in last_write:
                deps.add(last_write[r])
        for r in op.writes:
            if r in last_write:
                deps.add(last_write[r])
            for q in readers[r]:
                deps.add(q)
        deps.discard(i)
        op.deps = deps
        for r in op.reads:
            readers[r].append(i)
        for r in op.writes:
            last_write[r] = i
            readers[r] = []


DEP_DIST = int(knob("SG_EC_DEP_DIST", "2"))     # keep a VALU this many instructions away from the VALU whose result it reads
LAT = {"lgkm": 32, "vm": 250}          # instructions after which a load of the class is taken to have landed (placement only; the s_waitcnt is exact)


def schedule(ops, fill, window=96):
    """Merges the MFMA stream (program order) with the other instructions (program order, look-ahead `window`)."""
    build_deps(ops)
    n = len(ops)
    mf = [o.idx for o in ops if o.kind == "mfma"]
    pend = [o.idx for o in ops if o.kind != "mfma"]
    emitted_at = {}                 # op idx -> position in the output (counting s_nop states)
    out = []                        # (text, comment)
    order = []
    pos = 0                         # states issued so far
    mem_issued = {"vm": [], "lgkm": []}       # op idx of loads in issue order
    waited = {"vm": -1, "lgkm": -1}           # every load of the class up to this index in mem_issued has landed
    pending_reg = {}                # register -> (cls, index in mem_issued) of the load that will write it
    stats = {"nops": 0, "waits": 0, "gaps": defaultdict(int)}
    maxcnt = {"vm": 63, "lgkm": 15}

    def min_pos(op):
        """earliest position at which op may issue given the hazards the hardware does not interlock"""
        need = 0
        ow, orr = set(op.writes), set(op.reads)
        for d in op.deps:
            p = ops[d]
            at_ = emitted_at[d]
            if p.kind == "mfma" and op.kind != "mfma":
                w = set(p.writes)
                if w & orr or w & ow:
                    need = max(need, at_ + MFMA_TO_READ)
                elif p.cregs & ow:
                    need = max(need, at_ + MFMA_SRC_WAR)
                elif set(p.reads) & ow:
                    need = max(need, at_ + MFMA_AB_WAR)
            elif p.kind == "mfma" and op.kind == "mfma":
                w = set(p.writes)
                # the accumulate chain (C = the producer's whole D, same D) issues back to back; any other read of a result waits
                if w & orr and not (w <= orr and ow == w):
                    need = max(need, at_ + MFMA_TO_READ)
            elif p.kind == "valu" and op.kind == "mfma":
                if set(p.writes) & orr:
                    need = max(need, at_ + VALU_TO_MFMA)
        return need

    def waits_for(op):
        """{cls: index in mem_issued}: the youngest load per class op has to wait for"""
        res = {}
        for r in list(op.reads) + list(op.writes):
            if r in pending_reg:
                cls, k = pending_reg[r]
                if k > waited[cls]:
                    res[cls] = max(res.get(cls, -1), k)
        return res

    recent = []                     # the last VALU instructions issued (op idx): a dependent VALU issued right behind its producer stalls

    def cost(i):
        """states that issuing op i now is expected to lose: hazard padding + what is left of its loads' latency"""
        op = ops[i]
        c = max(0, min_pos(op) - pos)
        for cls, k in waits_for(op).items():
            c = max(c, emitted_at[mem_issued[cls][k]] + LAT[cls] - pos)
        if c == 0 and op.kind == "valu" and DEP_DIST:
            for back, d in enumerate(reversed(recent[-DEP_DIST:])):
                if d in op.deps and emitted_at[d] >= pos - DEP_DIST:
                    c = max(c, 0.25 * (DEP_DIST - back))         # a fraction of a state: any truly free instruction is preferred
        return c

    def emit(op):
        nonlocal pos
        for cls, k in waits_for(op).items():
            younger = min(len(mem_issued[cls]) - 1 - k, maxcnt[cls])
            out.append(("s_waitcnt %s(%d)" % ("vmcnt" if cls == "vm" else "lgkmcnt", younger), ""))
            pos += 1
            waited[cls] = k
            stats["waits"] += 1
        need = min_pos(op)
        if need > pos:
            gap = need - pos
            while gap > 0:
                g = min(gap, 16)
                out.append(("s_nop %d" % (g - 1), ""))
                gap -= g
            stats["nops"] += need - pos
            pos = need
        out.append((op.text, op.tag))
        order.append(op.idx)
        emitted_at[op.idx] = pos
        pos += 1
        if op.kind == "valu":
            recent.append(op.idx)
        if op.cls:
            mem_issued[op.cls].append(op.idx)
            for r in op.writes:
                pending_reg[r] = (op.cls, len(mem_issued[op.cls]) - 1)

    done = [False] * n
    # an instruction is never hoisted above an MFMA that precedes it in program order (it may sink below later ones): the order the
    # stages are listed in decides which MFMAs' shadows a stage is offered to
    not_before = {}
    seen = 0
    for o in ops:
        if o.kind == "mfma":
            seen += 1
        else:
            not_before[o.idx] = seen

    def ready(i):
        return all(done[d] for d in ops[i].deps)

    im = 0
    since = fill                     # fillers emitted behind the last MFMA
    while im < len(mf) or pend:
        cm = cost(mf[im]) if im < len(mf) and ready(mf[im]) else None
        if cm == 0 and since >= fill:
            emit(ops[mf[im]]); done[mf[im]] = True; im += 1
            stats["gaps"][since if since < 99 else 99] += 1
            since = 0
            continue
        pick, best = None, None
        for want_mem in (True, False):               # loads first: their latency is what the instructions behind them hide
            for j, i in enumerate(pend[:8 * window if want_mem else window]):
                if not_before[i] > im:
                    break
                if (ops[i].cls is not None) != want_mem or not ready(i):
                    continue
                c = cost(i)
                if c == 0:
                    pick = j
                    break
                if best is None or c < best[0]:
                    best = (c, j)
            if pick is not None:
                break
        if pick is None:
            if cm is not None and (best is None or cm <= best[0]):
                emit(ops[mf[im]]); done[mf[im]] = True; im += 1
                stats["gaps"][since if since < 99 else 99] += 1
                since = 0
                continue
            if best is None:
                raise RuntimeError("scheduler stuck at MFMA %d of %d, %d others left" % (im, len(mf), len(pend)))
            pick = best[1]
        i = pend.pop(pick)
        emit(ops[i]); done[i] = True
        since += 1
    verify(ops, order)
    stats["gaps"] = dict(sorted(stats["gaps"].items()))
    # a rough timeline (4-cycle issue states): every instruction one state, the matrix pipe 8 states per MFMA, a waited-for load lands
    # 32 (LDS) / 200 (global) states after its issue
    t, pipe = 0, 0
    issue_t = {}
    lat = {"lgkm": 32, "vm": 200}
    q = {"vm": [], "lgkm": []}
    for text, _ in out:
        opn = text.split()[0]
        if opn.startswith("v_mfma"):
            t = max(t, pipe)
            pipe = t + 8
        elif opn == "s_waitcnt":
            cls = "vm" if "vmcnt" in text else "lgkm"
            cnt = int(text[text.index("(") + 1:text.index(")")])
            if len(q[cls]) > cnt:
                t = max(t, q[cls][len(q[cls]) - 1 - cnt] + lat[cls])
        elif opn == "s_nop":
            t += int(text.split()[1])
        elif opn.startswith("ds_"):
            q["lgkm"].append(t)
        elif opn.startswith("global_"):
            q["vm"].append(t)
        t += 1
    stats["est_states"] = max(t, pipe)
    return out, stats


def verify(ops, out_order_idx):
    """every dependency edge of the program order is respected by the emitted order"""
    where = {i: k for k, i in enumerate(out_order_idx)}
    for op in ops:
        for d in op.deps:
            assert where[d] < where[op.idx], (ops[d].text, op.text)


# ------------------------------------------------------------------------------------------------------------------------------
# S1X (MLP2): conv1 on two fp16 pieces + statistics + maxima.  Two waves per SIMD (<= 256 VGPRs), base and the four A fragments in VGPRs.
# ------------------------------------------------------------------------------------------------------------------------------
class MapS1X:
    # Two waves per SIMD; base and the four A fragments in VGPRs; ONE conv1 accumulator (the partner wave's MFMAs cover the statistics
    # phase; a second accumulator left the compiler 13 registers around the statement and ~90 spilled values per tile)
    FREE = 12                     # v0..v11 stay with the compiler (inputs, values live across the statement), and everything from END on
    STAT_S = 12                   # 32: stat_s[16 t + q]
    STAT_Q = 44
    BEST = 76
    ACC = (108, 108)              # conv1's accumulator (32)
    BASE = 140                    # 32: the x_i half of conv1 (C operand of every slot's first MFMAs)
    FRAG = 172                    # 16: fr16[0..3]
    ROW = 188                     # 3 x 6: the neighbour rows in flight (16 B on an even register + channel 8 + one unused)
    NROW = 3
    RSTRIDE = 6
    X0 = 206                      # (d_hi | d_lo)
    X1 = 210                      # (d_hi | d8_hi, d8_lo)
    DS = 214                      # 5
    IDV = 219                     # 2 ids (alternating)
    IDSTRIDE = 1
    GLOBAL_IDS = True             # ids straight from the kNN table, a slot ahead (staging them in LDS cost 20 loads + 20 stores + their round trip per tile)
    T48 = 221
    OFF = 222
    END = 223
    # Round 5: the sums over the 20 slots are not accumulated per OUTPUT (16 v_pk_add_f32 = 32 issue slots per slot) but per INPUT: conv1 is
    # linear, sum_j y_j = 19 base + conv1(sum_j d_j): five v_add_f32 per slot into DSUM, one more conv1 behind the last slot.  DSUM lives in
    # the last five registers of the STAT_S output range (written only by the epilogue, behind the cut that consumes DSUM).
    DSUM = 12 + 27


def d_cut(p, m, j, row, sd, xs):
    if "D" in OMIT:
        return
    _d_cut(p, m, j, row, sd, xs)


def _d_cut(p, m, j, row, sd, xs):
    """the lane's five d values of slot j (row registers row..row+4) -> x0, x1 (fp16 hi / lo pieces); 18 VALU"""
    ds = m.DS
    for q in range(5):
        p.valu("v_fma_f32 v%d, v%d, %s, -%s" % (ds + q, row + q, sd, xs[q]), vr(row + q) + rd(xs[q]), vr(ds + q), tag="D%d" % j)
    if m is MapS1X:                                        # sum_j d_j (scaled by Sd like d): slot 0 writes it
        for q in range(5):
            if j == 0:
                p.valu("v_add_f32 v%d, 0, v%d" % (m.DSUM + q, ds + q), vr(ds + q), vr(m.DSUM + q))
            else:
                p.valu("v_add_f32 v%d, v%d, v%d" % (m.DSUM + q, m.DSUM + q, ds + q), vr(m.DSUM + q) + vr(ds + q), vr(m.DSUM + q))
    _cut5(p, m, ds, j)


def _cut5(p, m, ds, j):
    """five scaled values in registers ds..ds+4 -> x0, x1 (fp16 hi / lo pieces), destroying ds"""
    # hi pieces straight into x1[0..2]
    p.valu("v_cvt_pk_f16_f32 v%d, v%d, v%d" % (m.X1, ds, ds + 1), vr(ds, 2), vr(m.X1), tag="D%s" % j)
    p.valu("v_cvt_pk_f16_f32 v%d, v%d, v%d" % (m.X1 + 1, ds + 2, ds + 3), vr(ds + 2, 2), vr(m.X1 + 1))
    p.valu("v_cvt_pk_f16_f32 v%d, v%d, 0" % (m.X1 + 2, ds + 4), vr(ds + 4), vr(m.X1 + 2))
    if MIXLO and m is MapS2X:
        p.valu("v_mov_b32 v%d, v%d" % (m.X0, m.X1), vr(m.X1), vr(m.X0))
        p.valu("v_mov_b32 v%d, v%d" % (m.X0 + 1, m.X1 + 1), vr(m.X1 + 1), vr(m.X0 + 1))
        for q in range(5):                                 # lo = rn16(d - hi) as fp16 halves (see _lrelu_cut); the upper half of x1[3] stays 0 (init_outputs)
            src = m.X1 + q // 2
            dst = (m.X0 + 2 + q // 2) if q < 4 else (m.X1 + 3)
            p.valu("v_fma_mix%s_f16 v%d, v%d, -1.0, v%d op_sel:[%d,0,0] op_sel_hi:[1,0,0]" % ("hi" if q & 1 else "lo", dst, src, ds + q, q & 1),
                   vr(src) + vr(ds + q) + vr(dst), vr(dst))
        return
    # lo = d - hi (exact in fp32), in place
    for q in range(5):
        src = m.X1 + q // 2
        p.valu("v_fma_mix_f32 v%d, v%d, -1.0, v%d op_sel:[%d,0,0] op_sel_hi:[1,0,0]" % (ds + q, src, ds + q, q & 1), vr(src) + vr(ds + q), vr(ds + q))
    p.valu("v_mov_b32 v%d, v%d" % (m.X0, m.X1), vr(m.X1), vr(m.X0))
    p.valu("v_mov_b32 v%d, v%d" % (m.X0 + 1, m.X1 + 1), vr(m.X1 + 1), vr(m.X0 + 1))
    p.valu("v_cvt_pk_f16_f32 v%d, v%d, v%d" % (m.X0 + 2, ds, ds + 1), vr(ds, 2), vr(m.X0 + 2))
    p.valu("v_cvt_pk_f16_f32 v%d, v%d, v%d" % (m.X0 + 3, ds + 2, ds + 3), vr(ds + 2, 2), vr(m.X0 + 3))
    p.valu("v_cvt_pk_f16_f32 v%d, v%d, 0" % (m.X1 + 3, ds + 4), vr(ds + 4), vr(m.X1 + 3))


def init_outputs(p, m):
    if S2X_PKSTAT and m is MapS2X:                       # (experiment knob: the packed S2X statistics still accumulate from slot 0)
        for q in range(32):
            p.valu("v_mov_b32 v%d, 0" % (m.STAT_S + q), [], vr(m.STAT_S + q), tag="init" if q == 0 else "")
            p.valu("v_mov_b32 v%d, 0" % (m.STAT_Q + q), [], vr(m.STAT_Q + q))
            p.valu("v_mov_b32 v%d, 0xff800000" % (m.BEST + q), [], vr(m.BEST + q))
    # otherwise nothing to initialise: slot 0's statistics WRITE the sums, sums of squares and maxima (stats_s1x / stats_s2x)
    if MIXLO and m is MapS2X:
        p.valu("v_mov_b32 v%d, 0" % (m.X1 + 3), [], vr(m.X1 + 3))


def id_reg(m, j):
    return m.IDV + m.IDSTRIDE * (j & 1)


def id_read(p, m, j):
    if "G" in OMIT:
        return
    if getattr(m, "GLOBAL_IDS", False):                  # straight from the kNN table (MLP3: no LDS left for an id strip at two workgroups per CU)
        p.add("vmem", "global_load_dword v%d, %%[koff], %%[knn] offset:%d" % (id_reg(m, j), ID_STRIDE * j), ["%[koff]"], vr(id_reg(m, j)), cls="vm", tag="ID%d" % j)
    else:
        p.add("lds", "ds_read_b32 v%d, %%[ids] offset:%d" % (id_reg(m, j), j * 256), ["%[ids]"], vr(id_reg(m, j)), cls="lgkm", tag="ID%d" % j)


def gather(p, m, j):
    if "G" in OMIT:
        return
    _gather(p, m, j)


def _gather(p, m, j):
    row = m.ROW + m.RSTRIDE * (j % m.NROW)
    idv = id_reg(m, j)
    p.valu("v_mul_u32_u24 v%d, 48, v%d" % (m.T48, idv), vr(idv), vr(m.T48), tag="G%d" % j)
    p.valu("v_add_u32 v%d, v%d, %%[l16]" % (m.OFF, m.T48), vr(m.T48), vr(m.OFF))
    p.add("vmem", "global_load_dwordx4 %s, v%d, %%[x9m]" % (vt(row, 4), m.OFF), vr(m.OFF), vr(row, 4), cls="vm")
    p.add("vmem", "global_load_dword v%d, v%d, %%[x9m] offset:32" % (row + 4, m.T48), vr(m.T48), vr(row + 4), cls="vm")


def program_s1x(pk_stats):
    m = MapS1X
    p = Prog()
    xs = ["%%[xs%d]" % q for q in range(5)]
    # prologue: base and the four A fragments from LDS (once per tile), the first rows (an id is read one request ahead of its own: two
    # id registers take turns)
    for g in range(8):
        p.add("lds", "ds_read_b128 %s, %%[base] offset:%d" % (vt(m.BASE + 4 * g, 4), g * 1024), ["%[base]"], vr(m.BASE + 4 * g, 4), cls="lgkm", tag="B")
    for i in range(4):
        p.add("lds", "ds_read_b128 %s, %%[frag] offset:%d" % (vt(m.FRAG + 4 * i, 4), i * 1024), ["%[frag]"], vr(m.FRAG + 4 * i, 4), cls="lgkm", tag="F")
    init_outputs(p, m)
    id_read(p, m, 0)
    for j in range(min(K, m.NROW)):
        if j + 1 < K:
            id_read(p, m, j + 1)
        gather(p, m, j)
    for j in range(K):
        if j + m.NROW + 1 < K:
            id_read(p, m, j + m.NROW + 1)
        row = m.ROW + m.RSTRIDE * (j % m.NROW)
        d_cut(p, m, j, row, "%[sd]", xs)
        if j + m.NROW < K:
            gather(p, m, j + m.NROW)                     # re-uses the row registers d_cut has just read
        acc = m.ACC[j & 1]
        # statistics + maxima of the slot before, if it has an accumulator of its own: listed in front of conv1, i.e. offered to its shadows
        if j >= 1 and m.ACC[0] != m.ACC[1]:
            stats_s1x(p, m, j - 1, pk_stats)
        # conv1: the smaller terms first, like the C++ loop
        for t in range(2):
            p.mfma(acc + 16 * t, vt(m.FRAG + 4 * (2 + t), 4), vr(m.FRAG + 4 * (2 + t), 4), m.X1, m.BASE + 16 * t, tag="C1 %d" % j)
        for t in range(2):
            p.mfma(acc + 16 * t, vt(m.FRAG + 4 * t, 4), vr(m.FRAG + 4 * t, 4), m.X0, acc + 16 * t)
        if m.ACC[0] == m.ACC[1]:
            stats_s1x(p, m, j, pk_stats)
    if m.ACC[0] != m.ACC[1]:
        stats_s1x(p, m, K - 1, pk_stats)
    sums_s1x(p, m)
    return p


def stats_s1x(p, m, j, pk):
    """slot j's 32 outputs into the running sums of squares / maxima (the SUMS come from DSUM: sums_s1x).  Slot 0 WRITES them (y * y, y: the same
    bits as accumulating into zeroed registers, without the moves of an initialisation per tile)"""
    acc = m.ACC[j & 1]
    for q in range(0, 32, 2):
        y0, y1 = acc + q, acc + q + 1
        if j == 0:
            for y in (y0, y1):
                o = y - acc
                if not pk:
                    p.valu("v_mul_f32 v%d, v%d, v%d" % (m.STAT_Q + o, y, y), vr(y), vr(m.STAT_Q + o), tag="S%d" % j)
                p.valu("v_mov_b32 v%d, v%d" % (m.BEST + o, y), vr(y), vr(m.BEST + o), tag="S%d" % j)
            if pk:
                p.valu("v_pk_mul_f32 %s, %s, %s" % (vt(m.STAT_Q + q, 2), vt(y0, 2), vt(y0, 2)), vr(y0, 2), vr(m.STAT_Q + q, 2))
            continue
        if pk:
            p.valu("v_pk_fma_f32 %s, %s, %s, %s" % (vt(m.STAT_Q + q, 2), vt(y0, 2), vt(y0, 2), vt(m.STAT_Q + q, 2)), vr(m.STAT_Q + q, 2) + vr(y0, 2), vr(m.STAT_Q + q, 2), tag="S%d" % j)
        else:
            for y in (y0, y1):
                o = y - acc
                p.valu("v_fma_f32 v%d, v%d, v%d, v%d" % (m.STAT_Q + o, y, y, m.STAT_Q + o), vr(m.STAT_Q + o) + vr(y), vr(m.STAT_Q + o), tag="S%d" % j)
        p.valu("v_max_f32 v%d, v%d, v%d" % (m.BEST + q, m.BEST + q, y0), vr(m.BEST + q) + vr(y0), vr(m.BEST + q))
        p.valu("v_max_f32 v%d, v%d, v%d" % (m.BEST + q + 1, m.BEST + q + 1, y1), vr(m.BEST + q + 1) + vr(y1), vr(m.BEST + q + 1))


def sums_s1x(p, m):
    """behind the last slot: sum_j y_j = K base + conv1(sum_j d_j).  DSUM is scaled by 2^-5 first (exact): a slot's scaled d fits fp16 (2^16),
    the sum of twenty of them need not -- rows padded with point 0 (clusters of <= K points, model.py:513) repeat one far neighbour up to twenty
    times -- and 20 x 2^16 x 2^-5 does.  Then the slot's cut and four MFMAs on top of a ZERO accumulator... the C operand is `base` like every
    slot's (the fragments and the operand registers are the slot's), so acc = base + conv1(DSUM / 32), and
    sum = 32 acc + (K - 32) base: one v_mul_f32 + one v_fmamk_f32 per output, K = 20"""
    for q in range(5):
        p.valu("v_mul_f32 v%d, 0x3d000000, v%d" % (m.DSUM + q, m.DSUM + q), vr(m.DSUM + q), vr(m.DSUM + q), tag="SE")       # x 2^-5
    _cut5(p, m, m.DSUM, "E")
    acc = m.ACC[0]
    for t in range(2):
        p.mfma(acc + 16 * t, vt(m.FRAG + 4 * (2 + t), 4), vr(m.FRAG + 4 * (2 + t), 4), m.X1, m.BASE + 16 * t, tag="C1 E")
    for t in range(2):
        p.mfma(acc + 16 * t, vt(m.FRAG + 4 * t, 4), vr(m.FRAG + 4 * t, 4), m.X0, acc + 16 * t)
    for q in range(32):
        p.valu("v_mul_f32 v%d, 0x%08x, v%d" % (m.STAT_S + q, 0xc1400000, m.BASE + q), vr(m.BASE + q), vr(m.STAT_S + q), tag="SE")                    # (K - 32) base = -12 base
        p.valu("v_fmamk_f32 v%d, v%d, 0x%08x, v%d" % (m.STAT_S + q, acc + q, 0x42000000, m.STAT_S + q), vr(acc + q) + vr(m.STAT_S + q), vr(m.STAT_S + q))   # + 32 acc


# ------------------------------------------------------------------------------------------------------------------------------
# S2X (MLP3): conv1' -> LeakyReLU -> cut -> conv2 -> statistics + maxima.  One wave per SIMD: 256 VGPRs + the 20 A fragments in AGPRs.
# ------------------------------------------------------------------------------------------------------------------------------
class MapS2X:
    # Two waves per SIMD, VGPRs only.  (First version: one wave per SIMD with the 20 A fragments in AGPRs.  MFMAs that read an operand
    # from an AGPR run the chip at ~1.65 instead of ~1.97 GHz -- same cycle count per slot, 19 % more time; tools/micro/ec_slots_bench.hip,
    # DESIGN.md section 5 -- and one wave per SIMD leaves a tile's prologue and statistics flush uncovered.)  The A fragments stay in LDS
    # and pass through a ring of four register tuples, each requested as soon as the tuple's last reader has issued.  ONE neighbour row
    # in flight: a slot lasts ~1.4 us per wave here, several memory round trips.
    FREE = 16                     # v0..v15 stay with the compiler (nine inputs + what lives across the statement)
    STAT_S = 16
    STAT_Q = 48
    BEST = 80
    ACC2 = 112                    # 32: tile 0 | tile 1
    BUF = (144, 180)              # conv1 accumulator -> conv2's operand pieces, in place (32) + 4 extra for the first high block
    EXT = (176, 212)
    ROW = 216                     # 16 B on an even register + channel 8; the sixth register holds an id
    NROW = 1
    RSTRIDE = 6
    IDV = 221                     # 221, 222
    IDSTRIDE = 1
    T48 = 223
    X0 = 224
    X1 = 228
    DS = 232                      # 5
    OFF = 237
    TMP = 238                     # 2: 0.2 x
    NTMP = 2
    RING = (240, 244, 248, 252)   # A fragments in flight
    GLOBAL_IDS = True
    END = 256
    # LDS byte offsets from %[frag] = &lds.a1p[0][0][lane]: a1p[m][t] at 1024 (2 m + t); a2h[piece][ot][kb] behind a1p and a1x
    A1 = 0
    A2 = 8192 + 4096


def base_load(p, m, j):
    if "B" in OMIT:
        return
    _base_load(p, m, j)


def _base_load(p, m, j):
    """conv1's C operand (the x_i half + folded shift) of slot j from the wave's LDS strip into the slot's buffer"""
    buf = m.BUF[j & 1]
    for g in range(8):
        p.add("lds", "ds_read_b128 %s, %%[base] offset:%d" % (vt(buf + 4 * g, 4), g * 1024), ["%[base]"], vr(buf + 4 * g, 4), cls="lgkm", tag="B%d" % j)


def conv1_s2x(p, m, j, fillers=()):
    """the four conv1 MFMAs of slot j; `fillers` (ops) are listed six behind each of them, the rest behind the last"""
    buf = m.BUF[j & 1]
    fillers = list(fillers)
    k = 0
    for x, base_a in ((m.X1, 2), (m.X0, 0)):
        for t in range(2):
            p.mfma_frag(buf + 16 * t, m.A1 + 1024 * (base_a + t), x, buf + 16 * t, tag="C1 %d" % j if (x == m.X1 and t == 0) else "")
            k += 1
            take = fillers[:6] if k < 4 else fillers
            fillers = fillers[len(take):]
            p.ops.extend(take)


def lrelu_cut(p, m, j):
    if "L" in OMIT:
        return
    _lrelu_cut(p, m, j)


def _lrelu_cut(p, m, j):
    """LeakyReLU in place, then every 8 accumulator registers (one 16-deep k block of conv2) become 4 registers of high and 4 of
    low fp16 pieces: xl[kb] = regs 0..3 of the block, xh[kb] = regs 4..7 of the block BEFORE it (the extra block for kb = 0)."""
    buf, ext = m.BUF[j & 1], m.EXT[j & 1]
    for kb in range(4):
        r = buf + 8 * kb
        hi = ext if kb == 0 else buf + 8 * (kb - 1) + 4
        if LRELU_PK:
            # 0.2 x for two channels at once (the constant pair sits in SGPRs: VOP3P takes no literal), one v_max each
            for u in range(0, 8, 2):
                t = m.TMP
                p.valu("v_pk_mul_f32 %s, %s, %%[c02]" % (vt(t, 2), vt(r + u, 2)), vr(r + u, 2), vr(t, 2), tag="L%d.%d" % (j, kb))
                p.valu("v_max_f32 v%d, v%d, v%d" % (r + u, r + u, t), vr(r + u) + vr(t), vr(r + u))
                p.valu("v_max_f32 v%d, v%d, v%d" % (r + u + 1, r + u + 1, t + 1), vr(r + u + 1) + vr(t + 1), vr(r + u + 1))
        for u in range(0 if LRELU_PK else 8):
            t = m.TMP + (u % m.NTMP)
            p.valu("v_mul_f32 v%d, 0x3e4ccccd, v%d" % (t, r + u), vr(r + u), vr(t), tag="L%d.%d" % (j, kb))
            p.valu("v_max_f32 v%d, v%d, v%d" % (r + u, r + u, t), vr(r + u) + vr(t), vr(r + u))
        for u in range(4):
            a, b = r + 2 * u, r + 2 * u + 1
            p.valu("v_cvt_pk_f16_f32 v%d, v%d, v%d" % (hi + u, a, b), vr(a) + vr(b), vr(hi + u))
            if MIXLO:
                # lo = rn16(x - hi) written as fp16 halves of the destination by the mixed-precision FMA itself (x - hi is exact in fp32,
                # so the one rounding is cvt_pk's): two instructions per pair instead of three.  A half write keeps the other half: the
                # destination counts as read.  Order: u ascending, so r+u (= a for u = 0, an already consumed b / a otherwise) is free.
                p.valu("v_fma_mixlo_f16 v%d, v%d, -1.0, v%d op_sel:[0,0,0] op_sel_hi:[1,0,0]" % (r + u, hi + u, a), vr(hi + u) + vr(a) + vr(r + u), vr(r + u))
                p.valu("v_fma_mixhi_f16 v%d, v%d, -1.0, v%d op_sel:[1,0,0] op_sel_hi:[1,0,0]" % (r + u, hi + u, b), vr(hi + u) + vr(b) + vr(r + u), vr(r + u))
                continue
            p.valu("v_fma_mix_f32 v%d, v%d, -1.0, v%d op_sel:[0,0,0] op_sel_hi:[1,0,0]" % (a, hi + u, a), vr(hi + u) + vr(a), vr(a))
            p.valu("v_fma_mix_f32 v%d, v%d, -1.0, v%d op_sel:[1,0,0] op_sel_hi:[1,0,0]" % (b, hi + u, b), vr(hi + u) + vr(b), vr(b))
            p.valu("v_cvt_pk_f16_f32 v%d, v%d, v%d" % (r + u, a, b), vr(a) + vr(b), vr(r + u))


def conv2(p, m, j):
    """the 24 conv2 MFMAs of slot j: the two output tiles are independent accumulator chains and ALTERNATE -- a chain's next MFMA
    issued behind VALU fillers with no other MFMA in between pays ~40 cycles (measured: 12 + 12 tile-sequential MFMAs with six fillers
    each ran 860 ns per slot, the matrix pipe's share being 390)"""
    buf, ext = m.BUF[j & 1], m.EXT[j & 1]
    first = True
    for kb in range(4):
        xl = buf + 8 * kb
        xh = ext if kb == 0 else buf + 8 * (kb - 1) + 4
        for (piece, x) in ((1, xh), (0, xl), (0, xh)):                # w_lo x_hi, w_hi x_lo, w_hi x_hi: smallest terms first
            for ot in range(2):
                d = m.ACC2 + 16 * ot
                p.mfma_frag(d, m.A2 + 1024 * ((piece * 2 + ot) * 4 + kb), x, None if first else d, tag="C2 %d" % j if (first and ot == 0) else "")
            first = False


EXTRA_MFMA = int(knob("SG_EC_EXTRA_MFMA", "0"))       # timing experiment (with SG_EC_OMIT=Q): this many more MFMAs per slot, accumulating into the freed
                                                      # sum / sum-of-squares registers, A fragments from LDS like conv2's -- the ISSUE MIX of "BatchNorm 2's
                                                      # statistics from second moments of h1 on the matrix pipe" without its transposes (results are garbage)


def stats_s2x(p, m, j, ot):
    if "S" in OMIT:
        return
    if "Q" in OMIT:                                    # timing experiment: the maxima only (no sums, no sums of squares: 64 of a slot's 244 VALU)
        for q in range(16):
            z = m.ACC2 + 16 * ot + q
            o = 16 * ot + q
            if j == 0:
                p.valu("v_mov_b32 v%d, v%d" % (m.BEST + o, z), vr(z), vr(m.BEST + o), tag="S%d.%d" % (j, ot) if q == 0 else "")
            else:
                p.valu("v_max_f32 v%d, v%d, v%d" % (m.BEST + o, m.BEST + o, z), vr(m.BEST + o) + vr(z), vr(m.BEST + o), tag="S%d.%d" % (j, ot) if q == 0 else "")
        if ot == 1:
            buf, ext = m.BUF[j & 1], m.EXT[j & 1]
            for e in range(EXTRA_MFMA):
                d = (m.STAT_S if (e & 1) == 0 else m.STAT_Q) + 16 * ((e >> 1) & 1)
                x = ext if (e % 4) == 0 else buf + 8 * ((e % 4) - 1) + 4
                p.mfma_frag(d, m.A2 + 1024 * (e % 24), x, None if (j == 0 and e < 4) else d, tag="X%d" % j if e == 0 else "")
        return
    for q in range(0, 16 if S2X_PKSTAT else 0, 2):
        z = m.ACC2 + 16 * ot + q
        o = 16 * ot + q
        p.valu("v_pk_add_f32 %s, %s, %s" % (vt(m.STAT_S + o, 2), vt(m.STAT_S + o, 2), vt(z, 2)), vr(m.STAT_S + o, 2) + vr(z, 2), vr(m.STAT_S + o, 2), tag="S%d.%d" % (j, ot) if q == 0 else "")
        p.valu("v_pk_fma_f32 %s, %s, %s, %s" % (vt(m.STAT_Q + o, 2), vt(z, 2), vt(z, 2), vt(m.STAT_Q + o, 2)), vr(m.STAT_Q + o, 2) + vr(z, 2), vr(m.STAT_Q + o, 2))
        p.valu("v_max_f32 v%d, v%d, v%d" % (m.BEST + o, m.BEST + o, z), vr(m.BEST + o) + vr(z), vr(m.BEST + o))
        p.valu("v_max_f32 v%d, v%d, v%d" % (m.BEST + o + 1, m.BEST + o + 1, z + 1), vr(m.BEST + o + 1) + vr(z + 1), vr(m.BEST + o + 1))
    for q in range(16 if (j == 0 and not S2X_PKSTAT) else 0):           # slot 0 writes (see stats_s1x)
        z = m.ACC2 + 16 * ot + q
        o = 16 * ot + q
        p.valu("v_add_f32 v%d, 0, v%d" % (m.STAT_S + o, z), vr(z), vr(m.STAT_S + o), tag="S%d.%d" % (j, ot) if q == 0 else "")
        p.valu("v_mul_f32 v%d, v%d, v%d" % (m.STAT_Q + o, z, z), vr(z), vr(m.STAT_Q + o))
        p.valu("v_mov_b32 v%d, v%d" % (m.BEST + o, z), vr(z), vr(m.BEST + o))
    for q in range(0 if (S2X_PKSTAT or j == 0) else 16):
        z = m.ACC2 + 16 * ot + q
        o = 16 * ot + q
        p.valu("v_add_f32 v%d, v%d, v%d" % (m.STAT_S + o, m.STAT_S + o, z), vr(m.STAT_S + o) + vr(z), vr(m.STAT_S + o), tag="S%d.%d" % (j, ot) if q == 0 else "")
        p.valu("v_fma_f32 v%d, v%d, v%d, v%d" % (m.STAT_Q + o, z, z, m.STAT_Q + o), vr(m.STAT_Q + o) + vr(z), vr(m.STAT_Q + o))
        p.valu("v_max_f32 v%d, v%d, v%d" % (m.BEST + o, m.BEST + o, z), vr(m.BEST + o) + vr(z), vr(m.BEST + o))


ID_STRIDE = int(knob("SG_EC_ID_STRIDE", "4"))            # bytes between a lane's consecutive neighbour ids: 4 = row-major [N][20] table; 128 = slot-major per 32-row tile
MIXLO = bool(int(knob("SG_EC_MIXLO", "0")))             # S2X: the low fp16 pieces of conv2's operand straight from v_fma_mixlo/hi_f16
LRELU_PK = bool(int(knob("SG_EC_LRELU_PK", "0")))       # S2X: LeakyReLU's 0.2 x as v_pk_mul_f32 (asm operand %[c02] = the constant twice, in SGPRs)
S2X_PKSTAT = bool(int(knob("SG_EC_S2X_PKSTAT", "0")))   # S2X: packed sums / sums of squares
OMIT = set(x for x in knob("SG_EC_OMIT", "").split(",") if x)       # timing experiments: leave stages out (results are garbage)


def program_s2x():
    m = MapS2X
    p = Prog()
    xs = ["%%[xs%d]" % q for q in range(5)]
    init_outputs(p, m)
    id_read(p, m, 0)
    for j in range(min(K, m.NROW)):
        if j + 1 < K:
            id_read(p, m, j + 1)
        gather(p, m, j)

    def front(j, fillers=()):                            # everything of slot j up to conv1
        if j >= K:
            p.ops.extend(fillers)
            return
        if j + m.NROW + 1 < K:
            id_read(p, m, j + m.NROW + 1)
        base_load(p, m, j)
        d_cut(p, m, j, m.ROW + m.RSTRIDE * (j % m.NROW), "%[sd]", xs)
        if j + m.NROW < K:
            gather(p, m, j + m.NROW)
        conv1_s2x(p, m, j, fillers)

    front(0)
    front(1)
    lrelu_cut(p, m, 0)
    # steady state, slot j: conv2 of slot j (24 MFMAs, the two tiles alternating) and conv1 of slot j + 2 (4 MFMAs).  LeakyReLU + cut of
    # slot j + 1 is listed in FRONT of conv2 and so offered to its shadows six at a time; the statistics of slot j have to sit between
    # conv2 of slot j and conv2 of slot j + 1 (one accumulator pair): they run while conv1 of slot j + 2 waits for its base rows -- the
    # buffer those land in is conv2's operand until the last MFMA of slot j has issued -- and in conv1's shadows.
    for j in range(K):
        if j + 1 < K:
            lrelu_cut(p, m, j + 1)
        conv2(p, m, j)
        q = Prog()
        stats_s2x(q, m, j, 0)
        stats_s2x(q, m, j, 1)
        front(j + 2, q.ops)                              # the slot's statistics: in the shadows of conv1's four MFMAs and while its base rows load
    p.ops = assign_frag_ring(p.ops, m.RING)
    return p


# ------------------------------------------------------------------------------------------------------------------------------
def render(name, sched, clobber_v, clobber_a, doc):
    lines = []
    lines.append("// %s" % doc)
    lines.append("#define %s \\" % name)
    for text, tag in sched:
        lines.append('    "%s\\n\\t"%s \\' % (text, ("  /* %s */" % tag) if tag else ""))
    lines.append('    ""')
    cl = ", ".join('"v%d"' % i for i in clobber_v) + (", " if clobber_a else "") + ", ".join('"a%d"' % i for i in clobber_a)
    lines.append("#define %s_CLOBBERS %s" % (name, cl))
    return "\n".join(lines)


def count(sched):
    c = defaultdict(int)
    for text, _ in sched:
        c[text.split()[0]] += 1
    return c


def main():
    fill = int(knob("SG_EC_FILL", "6"))
    out = []
    out.append("// GENERATED by tools/gen_edgeconv_asm.py -- do not edit by hand (re-run the generator).")
    out.append("// The neighbour-slot loops of k_edgeconv_h<S1X / S2X> (kernels_edgeconv.hip) as hand-scheduled gfx950 instruction streams.")
    out.append("#pragma once")
    report = []
    for pk in (0, 1):
        p = program_s1x(bool(pk))
        sched, st = schedule(p.ops, fill)
        c = count(sched)
        nm = "SG_EC_S1X_SLOTS_PK" if pk else "SG_EC_S1X_SLOTS"
        m = MapS1X
        outs = list(range(m.STAT_S, m.BEST + 32))
        clob = [i for i in range(m.FREE, m.END) if i not in outs]
        out.append(render(nm, sched, clob, [], "MLP2, K = %d slots, %s statistics: %d instructions (%d MFMA, %d s_nop states, %d s_waitcnt)"
                          % (K, "packed" if pk else "plain", len(sched), c["v_mfma_f32_32x32x16_f16"], st["nops"], st["waits"])))
        report.append((nm, len(sched), dict(c), st))
    p = program_s2x()
    sched, st = schedule(p.ops, fill)
    c = count(sched)
    m = MapS2X
    outs = list(range(m.STAT_S, m.BEST + 32))
    clob = [i for i in range(m.FREE, m.END) if i not in outs]
    out.append(render("SG_EC_S2X_SLOTS", sched, clob, [], "MLP3, K = %d slots: %d instructions (%d MFMA, %d s_nop states, %d s_waitcnt)"
                      % (K, len(sched), c["v_mfma_f32_32x32x16_f16"], st["nops"], st["waits"])))
    report.append(("SG_EC_S2X_SLOTS", len(sched), dict(c), st))
    for cls, mm in (("S1X", MapS1X), ("S2X", MapS2X)):
        for k in ("STAT_S", "STAT_Q", "BEST"):
            v = getattr(mm, k)
            out.append('#define SG_EC_%s_%s0 "{v[%d:%d]}"' % (cls, k, v, v + 15))
            out.append('#define SG_EC_%s_%s1 "{v[%d:%d]}"' % (cls, k, v + 16, v + 31))
    out.append('#define SG_EC_S1X_BASE0 "{v[%d:%d]}"' % (MapS1X.BASE, MapS1X.BASE + 15))
    out.append('#define SG_EC_S1X_BASE1 "{v[%d:%d]}"' % (MapS1X.BASE + 16, MapS1X.BASE + 31))
    for i in range(4):
        out.append('#define SG_EC_S1X_FRAG%d "{v[%d:%d]}"' % (i, MapS1X.FRAG + 4 * i, MapS1X.FRAG + 4 * i + 3))
    with open(OUT, "w") as f:
        f.write("\n".join(out) + "\n")
    for nm, n, c, st in report:
        print(nm, n, "instructions;", {k: v for k, v in sorted(c.items())}, st, "-> ~%d cycles per slot" % (st["est_states"] * 4 // K), file=sys.stderr)


if __name__ == "__main__":
    main()
