#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the kernels of one scene forward (one PMC pass, no trace domains besides kernel-trace).
#   gpurun -- 'bash tools/pmc_kernel.sh <outdir-name> [N S]'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmc_sq}
N=${2:-150000}; S=${3:-1500}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $OUT -- python3 $R/tools/time_scene.py $N $S > $OUT.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m_ = re.search(r"(k_[a-z0-9_]+(<[^(]*>)?)", r["Kernel_Name"])
    k = m_.group(1) if m_ else r["Kernel_Name"][:40]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    rows.append((m.get("SQ_WAVE_CYCLES", 0), k, m, len(next(iter(c.values())))))
for wc, k, m, n in sorted(rows, reverse=True)[:16]:
    w = max(m.get("SQ_WAVES", 1), 1)
    print("%-50s launches %3d waves %7d | per wave: VALU %7.0f SALU %6.0f LDS %6.0f | wave-cycles(x4) %8.0f  active %4.1f%% wait_any %4.1f%% wait_inst %4.1f%%" % (
        k[-50:], n, w, m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_SALU", 0) / w, m.get("SQ_INSTS_LDS", 0) / w, 4 * wc / w,
        100 * m.get("SQ_ACTIVE_INST_ANY", 0) / max(wc, 1), 100 * m.get("SQ_WAIT_ANY", 0) / max(wc, 1), 100 * m.get("SQ_WAIT_INST_ANY", 0) / max(wc, 1)))
PY
