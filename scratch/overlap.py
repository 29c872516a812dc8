import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
t0 = None
pers = [r for r in rows if "persist" in r["Kernel_Name"]]
last = pers[-4:]
base = min(float(r["Start_Timestamp"]) for r in last)
for r in last:
    print(f'{r["Kernel_Name"][11:40]:30s} start {(float(r["Start_Timestamp"])-base)/1e3:9.1f} us  end {(float(r["End_Timestamp"])-base)/1e3:9.1f} us  dur {(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3:8.1f}')
# whole last step: from first kernel after previous adam to this adam
adam = [i for i, r in enumerate(rows) if "adam_flat_dev_kernel" in r["Kernel_Name"]]
a, b = adam[-2], adam[-1]
st = float(rows[a]["End_Timestamp"]); en = float(rows[b]["End_Timestamp"])
print(f"last step wall (adam to adam): {(en-st)/1e3:.1f} us, {b-a} kernels")
