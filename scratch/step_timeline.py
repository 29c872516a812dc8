"""Print the kernel timeline of one steady-state eager training step from a rocprofv3 kernel trace (csv).
usage: step_timeline.py <dir-with-*kernel_trace.csv> [step-from-end]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_flat_dev_kernel" in r["Kernel_Name"]]
a, b = adam[-k - 1], adam[-k]
seg = rows[a + 1:b + 1]
base = int(rows[a]["End_Timestamp"])
qs = sorted(set(r["Queue_Id"] for r in seg))
for r in seg:
    st = (int(r["Start_Timestamp"]) - base) / 1e3
    du = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    nm = r["Kernel_Name"].replace("void mser::", "").replace("mser::", "").replace("(anonymous namespace)::", "")[:36]
    print(f"q{qs.index(r['Queue_Id'])} {st:8.1f} +{du:7.1f} {nm} g={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
print("step wall (adam end -> adam end): %.1f us, %d kernels" % ((int(rows[b]["End_Timestamp"]) - base) / 1e3, len(seg)))
