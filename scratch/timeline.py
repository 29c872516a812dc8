import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
adam = [i for i, r in enumerate(rows) if "adam_flat_dev_kernel" in r["Kernel_Name"]]
a, b = adam[-2], adam[-1]
seg = sorted(rows[a + 1:b + 1], key=lambda r: float(r["Start_Timestamp"]))
base = float(seg[0]["Start_Timestamp"])
qs = sorted(set(r["Queue_Id"] for r in seg))
print("queues:", qs)
# per queue busy intervals summary + list of long kernels
for r in seg:
    st = (float(r["Start_Timestamp"]) - base) / 1e3; du = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    nm = r["Kernel_Name"].replace("void mser::", "").replace("mser::", "")[:34]
    if du > 40 or "persist" in nm:
        print(f"q{qs.index(r['Queue_Id'])} {st:9.1f} +{du:8.1f}  {nm}")
# coarse: for each queue, first start / last end / sum of durations
for q in qs:
    rs = [r for r in seg if r["Queue_Id"] == q]
    st = min(float(r["Start_Timestamp"]) for r in rs); en = max(float(r["End_Timestamp"]) for r in rs)
    busy = sum(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rs)
    print(f"queue {qs.index(q)}: n={len(rs):4d} first {(st-base)/1e3:8.1f} last {(en-base)/1e3:8.1f} busy {busy/1e3:8.1f} us")
# gaps on the critical milestones
def find(sub):
    return [r for r in seg if sub in r["Kernel_Name"]]
for sub in ("spk_fwd_persist", "lsthm_fwd_persist", "logsoftmax_tb_fwd", "masked_nll_bwd", "lsthm_bwd_persist", "spk_bwd_persist", "adam_flat_dev"):
    for r in find(sub):
        print(f"{sub:22s} start {(float(r['Start_Timestamp'])-base)/1e3:8.1f} end {(float(r['End_Timestamp'])-base)/1e3:8.1f}")
