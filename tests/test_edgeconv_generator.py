"""The committed seggroup_amd/csrc/edgeconv_slots_gen.h IS what tools/gen_edgeconv_asm.py writes (the Makefile
lists the header as a prerequisite with no rule behind it -- this test is the rule).  CPU only: the generator is pure Python."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_slot_loops_equal_the_generators_output(tmp_path):
    out = tmp_path / "edgeconv_slots_gen.h"
    env = {k: v for k, v in os.environ.items() if not k.startswith("SG_EC_")}          # no development knob may leak in
    env["SG_EC_OUT"] = str(out)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_edgeconv_asm.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    committed = open(os.path.join(ROOT, "seggroup_amd", "csrc", "edgeconv_slots_gen.h")).read()
    generated = out.read_text()
    assert generated == committed, "edgeconv_slots_gen.h is stale: run `python3 tools/gen_edgeconv_asm.py` and commit the result"


def test_emitted_streams_hold_the_mfma_hazard_distances_and_waits():
    """the MFMA -> VALU hazards of gfx9xx are software-managed, and the slot loops are `asm volatile` text the compiler does not
    check.  tools/audit_edgeconv_asm.py re-derives them from the COMMITTED text alone (register operands parsed out of every instruction, issue
    states counted): an MFMA's destination is not touched for 12 states, a VALU result is not fed to an MFMA for 2, C / A / B operands are not
    overwritten early, and no asynchronous load's destination is touched before a covering s_waitcnt.  The audit must also SEE a violation:
    with the s_nops or the waits taken out of a stream it has to fail."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_edgeconv_asm as A
    macros = A.parse_macros(open(os.path.join(ROOT, "seggroup_amd", "csrc", "edgeconv_slots_gen.h")).read())
    assert {"SG_EC_S1X_SLOTS", "SG_EC_S2X_SLOTS"} <= set(macros)
    for name, lines in macros.items():
        n_ins, n_mfma, _, problems = A.audit(name, lines)
        assert n_mfma in (84, 560) and n_ins > 1500, (name, n_ins, n_mfma)
        assert not problems, problems[:5]
    s2x = macros["SG_EC_S2X_SLOTS"]
    assert any(l.startswith("s_nop") for l in s2x) and any(l.startswith("s_waitcnt") for l in s2x)
    _, _, _, p_nop = A.audit("no nops", [l for l in s2x if not l.startswith("s_nop")])
    assert any("MFMA" in p for p in p_nop)
    _, _, _, p_wait = A.audit("no waits", [l for l in s2x if not l.startswith("s_waitcnt")])
    assert any("has not been waited for" in p for p in p_wait)
