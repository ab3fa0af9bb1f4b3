#!/usr/bin/env python3
"""Independent audit of the hand-scheduled EdgeConv slot loops ("audit the wait states gen_edgeconv_asm.py emits between an MFMA
result and its first VALU reader"; software-managed hazards on gfx9xx).

Reads the COMMITTED text of seggroup_amd/csrc/edgeconv_slots_gen.h -- not the generator's internal op list -- parses every instruction's register
operands, replays the stream counting issue states (an instruction = 1, `s_nop N` = N + 1) and checks, for every v_mfma:

  D -> other  a non-MFMA instruction that reads or writes a register of the MFMA's destination comes >= 12 states behind it (hipcc pads `s_nop 11`
              between this 8-pass MFMA and a VALU reader of its result; the generator asks for 13)
  D -> MFMA   another MFMA that reads the destination as A / B, or as C without being the accumulate chain (same C and D), likewise
  VALU -> A/B/C  a VALU write to an operand of an MFMA comes >= 2 states in front of it (hipcc: `s_nop 1`)
  C WAR       a non-MFMA write to the MFMA's C operand comes >= 11 states behind it (C is read pass by pass; the generator asks for 12)
  A/B WAR     a write to its A / B operand >= 1 state behind it

and for every asynchronous load (ds_read / global_load): no instruction reads or overwrites its destination before an `s_waitcnt` of its class
whose count covers it (loads return in order per class).  Prints one line per macro; exits non-zero on a violation.

    python3 tools/audit_edgeconv_asm.py [path/to/edgeconv_slots_gen.h]
"""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT = os.path.join(HERE, "..", "seggroup_amd", "csrc", "edgeconv_slots_gen.h")
D_TO_OTHER, VALU_TO_MFMA, C_WAR, AB_WAR = 12, 2, 11, 1

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(operand):
    out = []
    for m in REG.finditer(operand):
        if m.group(1):
            out.append(m.group(1) + m.group(2))
        else:
            out += [m.group(3) + str(i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]
    return out


def parse_macros(text):
    """{macro name: [instruction text]} for the *_SLOTS macros (instruction strings only)"""
    macros, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"#define (SG_EC_\w*SLOTS\w*)\s*\\", line)
        if m and not m.group(1).endswith("CLOBBERS"):
            cur = m.group(1)
            macros[cur] = []
            continue
        if cur is None:
            continue
        s = re.match(r'\s*"(.*?)\\n\\t"', line)
        if s:
            macros[cur].append(s.group(1))
        elif not line.rstrip().endswith("\\"):
            cur = None
    return macros


def split_ops(ins):
    mn, _, rest = ins.partition(" ")
    rest = re.sub(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[[^\]]*\]", "", rest)      # modifiers hold digits, not registers
    rest = re.sub(r"\boffset:\d+", "", rest)
    return mn, [o.strip() for o in rest.split(",") if o.strip()]


def audit(name, lines):
    problems = []
    pos = 0
    mfmas = []                          # live records: dict(pos, D, A, B, C, chain)
    last_valu_write = {}                # reg -> pos
    pending = {"vm": [], "lgkm": []}    # loads in issue order: (dest regs)
    n_mfma = n_ins = 0
    for ins in lines:
        mn, ops = split_ops(ins)
        if mn == "s_nop":
            pos += int(ops[0]) + 1
            continue
        if mn == "s_waitcnt":
            for cls, key in (("vm", "vmcnt"), ("lgkm", "lgkmcnt")):
                m = re.search(key + r"\((\d+)\)", ins)
                if m:
                    keep = int(m.group(1))
                    pending[cls] = pending[cls][len(pending[cls]) - keep:] if keep else []
            pos += 1
            continue
        n_ins += 1
        is_mfma = mn.startswith("v_mfma")
        is_load = mn.startswith("ds_read") or mn.startswith("global_load")
        is_store = mn.startswith("ds_write") or mn.startswith("global_store") or mn.startswith("global_atomic")
        if is_store:
            writes, reads = [], [r for o in ops for r in regs_of(o)]
        else:
            writes, reads = regs_of(ops[0]) if ops else [], [r for o in ops[1:] for r in regs_of(o)]
        # asynchronous results must have been waited for
        for cls in ("vm", "lgkm"):
            for k, dest in enumerate(pending[cls]):
                hit = set(dest) & (set(reads) | set(writes))
                if hit:
                    problems.append(f"{name}: `{ins}` touches {sorted(hit)[:3]} while the {cls} load #{k} of {len(pending[cls])} outstanding that writes them has not been waited for")
        if is_mfma:
            n_mfma += 1
            D, A, B, C = regs_of(ops[0]), regs_of(ops[1]), regs_of(ops[2]), regs_of(ops[3]) if len(ops) > 3 else []
            for r in A + B + C:
                if r in last_valu_write and pos - last_valu_write[r] < VALU_TO_MFMA:
                    problems.append(f"{name}: `{ins}` reads {r} {pos - last_valu_write[r]} state(s) behind the VALU write to it (needs {VALU_TO_MFMA})")
            for p in mfmas:
                dist = pos - p["pos"]
                if dist >= D_TO_OTHER + 2:
                    continue
                d = set(p["D"])
                chain = set(C) == d and set(D) == d                      # accumulate chain: back to back is fine
                if (d & set(A + B)) or ((d & set(C)) and not chain) or ((d & set(D)) and not chain):
                    if dist < D_TO_OTHER:
                        problems.append(f"{name}: `{ins}` uses the result of the MFMA {dist} states in front of it (needs {D_TO_OTHER})")
            mfmas.append({"pos": pos, "D": D, "A": A, "B": B, "C": C})
            mfmas = [p for p in mfmas if pos - p["pos"] < 24]
        else:
            for p in mfmas:
                dist = pos - p["pos"]
                if dist >= 24:
                    continue
                d = set(p["D"])
                if dist < D_TO_OTHER and (d & set(reads) or d & set(writes)):
                    problems.append(f"{name}: `{ins}` touches the destination of the MFMA {dist} states in front of it (needs {D_TO_OTHER})")
                if dist < C_WAR and set(p["C"]) & set(writes) and not set(p["C"]) <= d:
                    problems.append(f"{name}: `{ins}` overwrites the C operand of the MFMA {dist} states in front of it (needs {C_WAR})")
                if dist < AB_WAR and set(p["A"] + p["B"]) & set(writes):
                    problems.append(f"{name}: `{ins}` overwrites an A / B operand of the MFMA {dist} states in front of it")
            if is_load:
                pending["vm" if mn.startswith("global_load") else "lgkm"].append(writes)
            elif mn.startswith("v_"):
                for r in writes:
                    last_valu_write[r] = pos
        pos += 1
    return n_ins, n_mfma, pos, problems


def main(path=DEFAULT):
    macros = parse_macros(open(path).read())
    if not macros:
        print("no *_SLOTS macro found in", path)
        return 2
    bad = 0
    for name, lines in macros.items():
        n_ins, n_mfma, states, problems = audit(name, lines)
        print(f"{name}: {n_ins} instructions ({n_mfma} MFMA), {states} issue states: {'OK' if not problems else '%d VIOLATION(S)' % len(problems)}")
        for p in problems[:10]:
            print("   " + p)
        bad += len(problems)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(*sys.argv[1:2]))
