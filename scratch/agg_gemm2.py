import csv, glob
f = glob.glob("/tmp/pg/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm_kernel" in r["Kernel_Name"]]
Ks = (0, 16, 32, 48, 64, 128, 256, 512, 1024)
for gi in range(len(rows) // 20):
    ds = sorted(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows[gi*20:(gi+1)*20])
    nm = rows[gi*20]["Kernel_Name"][:36]
    print("M=%d K=%5d: median %7.2f us  min %7.2f  %s" % (4096 if gi < 9 else 64, Ks[gi % 9], ds[10]/1e3, ds[0]/1e3, nm))
