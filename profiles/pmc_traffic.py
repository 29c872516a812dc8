#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950: TCC has 4
slots, they cost 3 + 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/fetch -o fetch -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/write -o write -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph
    python profiles/pmc_traffic.py gpurun_out/pmc/fetch gpurun_out/pmc/write > profiles/pmc_traffic.json

Units and gfx950 corrections exactly as the guide's HBM section prescribes: both counters are in KiB; FETCH_SIZE tallies
128-byte requests at 64 bytes on gfx950, so the read side is DOUBLED; WRITE_SIZE is taken as is.
traffic_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the launches of the kernel.
bench.py reads the committed JSON to fill roofline.traffic (PMC collection serialises the kernels, it cannot run inside the
timed region)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def collect(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return acc


def main():
    fe = collect(sys.argv[1], "FETCH_SIZE")
    wr = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fe) | set(wr)):
        f_kib = fe[k][0] / max(fe[k][1], 1) if k in fe else 0.0
        w_kib = wr[k][0] / max(wr[k][1], 1) if k in wr else 0.0
        out[k] = dict(launches=max(fe[k][1] if k in fe else 0, wr[k][1] if k in wr else 0),
                      fetch_size_kib_raw=round(f_kib, 2), write_size_kib=round(w_kib, 2),
                      traffic_bytes_per_launch=int((2.0 * f_kib + w_kib) * 1024))
    json.dump(dict(note="(2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch; FETCH_SIZE doubled per the gfx950 correction of "
                        "MI355X_MICROARCH.md (HBM section); separate --pmc passes", kernels=out), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
