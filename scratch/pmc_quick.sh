#!/bin/bash
# usage: pmc_quick.sh <tag> [MSER_OPTIONS]   -- FETCH/WRITE passes of a short eager bench, prints cell_* traffic
tag=$1; export MSER_OPTIONS=$2
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/${tag}_fetch -o fetch -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-variants --no-graph --no-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/${tag}_write -o write -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-variants --no-graph --no-roofline > /dev/null 2>&1
python profiles/pmc_traffic.py gpurun_out/prof/${tag}_fetch gpurun_out/prof/${tag}_write > gpurun_out/prof/${tag}_pmc_traffic.json
python - <<PY
import json
j=json.load(open("gpurun_out/prof/${tag}_pmc_traffic.json"))
for k,v in j["kernels"].items():
    if "cell_bwd" in k or "cell_fwd" in k: print("${tag}", k[:40], round(v["traffic_bytes_per_launch"]/1e9,3), "GB/launch")
PY
