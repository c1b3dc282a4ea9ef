cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in x mf; do
rm -rf gpurun_out/skt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/skt -- python3 tools/swin_in_process.py $w > /dev/null 2>&1
python3 - $w <<'PY'
import csv, glob, sys, collections
f = glob.glob('gpurun_out/skt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("==", sys.argv[1], list(rows[0].keys())[:16])
c = collections.Counter()
for r in rows[-6000:]:
    n = r['Kernel_Name']
    if 'wgrad_big' in n or 'gemm_big_kernel<6, false, 0, 0' in n:
        c[(n[:40], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Workgroup_Size_X') or r.get('Workgroup_Size'), r['Queue_Id'])] += 1
for k, v in c.most_common(12): print(v, k)
PY
done
rm -rf gpurun_out/skt
