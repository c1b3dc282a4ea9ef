cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in x mf; do
rm -rf gpurun_out/skt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/skt -- python3 tools/swin_in_process.py $w > /dev/null 2>&1
python3 - $w <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/skt/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("==", sys.argv[1])
for r in rows[:400]:
    n = r['Name']
    if any(k in n for k in ('gemm_big_kernel<', 'wgrad_big', 'attn', 'Li128ELi128ELi0ELi1E', 'gelu', 'window')) and 'float' not in n and 'ILf' not in n and 'IfL' not in n:
        print(f"{int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:9.2f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  {n[:70]}")
PY
done
rm -rf gpurun_out/skt
