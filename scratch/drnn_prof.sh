#!/bin/bash
# DialogueRNN BiModel (configs[3]) step: kernel stats under rocprofv3
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/drnn_prof
mkdir -p $OUT
python3 $R/scratch/drnn_steps.py 5 > $OUT/steps_persist.log 2>&1
python3 $R/scratch/drnn_steps.py 5 graph > $OUT/steps_persist_graph.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o drnn -- python3 $R/scratch/drnn_steps.py 3 > $OUT/prof.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open("$OUT/kernel_stats_top.txt", "w") as o:
    for r in rows[:25]:
        o.write(f"{r['Name'][:90]:90s} calls={r['Calls']:>7s} total_ms={float(r['TotalDurationNs'])/1e6:10.2f} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}\n")
PY
cat $OUT/steps_persist.log $OUT/steps_persist_graph.log | grep BiModel
head -16 $OUT/kernel_stats_top.txt
