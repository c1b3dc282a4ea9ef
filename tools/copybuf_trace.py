"""Where do the __amd_rocclr_copyBuffer dispatches of a training step come from?  Reads a rocprofv3 --kernel-trace CSV
(gpurun_out/cb/**/_kernel_trace.csv), prints the copy dispatches per queue with the kernels before / after them."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "copyBuffer" in n]
print("dispatches", len(rows), "copyBuffer", len(idx))
ctx = collections.Counter()
for i in idx:
    q = rows[i]["Queue_Id"]
    prev = next((names[j][:60] for j in range(i - 1, max(i - 40, -1), -1) if rows[j]["Queue_Id"] == q and "copyBuffer" not in names[j]), "-")
    nxt = next((names[j][:60] for j in range(i + 1, min(i + 40, len(rows))) if rows[j]["Queue_Id"] == q and "copyBuffer" not in names[j]), "-")
    ctx[(q, prev, nxt, rows[i].get("Grid_Size", "?"))] += 1
for k, v in ctx.most_common(25):
    print(v, k)
