"""Print the key fields of bench.py JSON lines: jl.py file..."""
import json, sys
for f in sys.argv[1:]:
    for line in open(f):
        line = line.strip()
        if not line.startswith("{"):
            continue
        j = json.loads(line)
        r = j.get("roofline") or {}
        fw = r.get("lsthm_forward") or {}
        print(f"{f}: {j['ms_per_step']:.4f} ms/step [{j['config'].get('launch')}]  bwd {r.get('avg_launch_us')} us  fwd {fw.get('avg_launch_us')} us  frac {r.get('frac')}")
        v = j.get("variants") or {}
        for k, x in v.items():
            if isinstance(x, dict):
                print("   ", k, {a: b for a, b in x.items() if a in ("ms_per_step", "launch", "error", "weight_stream_GBps", "utterances_per_s")})
