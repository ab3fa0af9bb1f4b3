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
