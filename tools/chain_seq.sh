# sequence of kernels of ONE EfficientSATRN step on the chain's queue with start / duration / gap (GPU box): bash tools/chain_seq.sh > out.txt
mkdir -p gpurun_out/cs_tmp && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cs_tmp -- python3 bench.py --steps 3 --warmup 3 --no-decode --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/cs_tmp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the busiest queue = the chain
q = collections.Counter(r['Queue_Id'] for r in rows).most_common(1)[0][0]
ch = [r for r in rows if r['Queue_Id'] == q]
# last step: from the last adamw backwards to the previous adamw
idx = [i for i, r in enumerate(ch) if 'adamw' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
prev_end = None
t0 = int(ch[a]['Start_Timestamp'])
for r in ch[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {gap:6.2f}  {r['Kernel_Name'][:100]}")
    prev_end = e
PY
rm -rf gpurun_out/cs_tmp
