#!/usr/bin/env python3
"""Per-kernel SQ counter summary (MFMA pipe utilisation, where the waves wait) from one rocprofv3 PMC pass:

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \\
              SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/sq -o sq -- \\
              python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-variants
    python profiles/sq_counters.py gpurun_out/sq > profiles/r01_f_sq_counters.json

Units (MI355X_MICROARCH.md "rocprofv3 PMC slots"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over
the waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs.  Reported per kernel (averages over its launches):
  mfma_busy_frac   = MFMA_BUSY_CYCLES / (SIMDs the grid can occupy * kernel duration * 2.4 GHz nominal clock): a LOWER bound on
                     the pipe utilisation (the real clock under load is lower) and directly comparable with achieved / peak FLOP/s
  wait_any_frac    = SQ_WAIT_ANY / SQ_WAVE_CYCLES        (waves parked at s_waitcnt / s_barrier)
  wait_inst_frac   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   (issue stalls: MFMA dependency / pipe busy)
  active_frac      = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
Counter collection serialises kernels, so the durations here are isolated-kernel durations, not the overlapped ones of a step."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def main():
    f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    meta = {}
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
        if not k.startswith("mser::"):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
        acc[k]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        n[k]["_ns"] += 1
        meta[k] = (int(r["Grid_Size"]), int(r["Workgroup_Size"]))
    out = {}
    for k in acc:
        a = {c: acc[k][c] / max(n[k][c], 1) for c in acc[k]}
        grid, wg = meta[k]
        waves = grid // 64
        wgs = grid // wg
        simds = min(1024, waves)                               # SIMDs that can hold a wave of this grid (1024 on the chip)
        us = a["_ns"] / 1e3
        wave_cycles = 4.0 * a.get("SQ_WAVE_CYCLES", 0.0)       # quad-cycles -> cycles, summed over waves
        clk_ghz = wave_cycles / max(waves, 1) / max(a["_ns"], 1)     # lower bound: waves need not live for the whole kernel
        kcycles = a["_ns"] * 2.4                                      # nominal 2.4 GHz
        wc = max(a.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        out[k] = dict(launches=n[k]["_ns"], avg_us=round(us, 1), workgroups=wgs, waves=waves,
                      wave_clock_ghz_lower_bound=round(clk_ghz, 2),
                      mfma_busy_frac=round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(simds * kcycles, 1.0), 3),
                      wait_any_frac=round(a.get("SQ_WAIT_ANY", 0.0) / wc, 3),
                      wait_inst_frac=round(a.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                      active_frac=round(a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3),
                      lds_bank_conflict_cycles=int(a.get("SQ_LDS_BANK_CONFLICT", 0.0)))
    json.dump(dict(note="see profiles/sq_counters.py for units and definitions", kernels=out), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
