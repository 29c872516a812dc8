import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0, 1e18, 0])
with open(f) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"][:60]
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        a = agg[key]; a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot/1e6:.2f} ms")
for k, v in rows[:45]:
    print(f"{v[1]/1e6:8.3f} ms  n={v[0]:5d} avg={v[1]/v[0]/1e3:8.2f}us min={v[2]/1e3:7.2f} max={v[3]/1e3:7.2f}  grid=({k[1]},{k[2]},{k[3]}) wg={k[4]}  {k[0]}")
