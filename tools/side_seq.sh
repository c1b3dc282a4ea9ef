# the SIDE queue (weight gradients) of one EfficientSATRN step beside the chain: start / duration / gap per kernel + busy totals (GPU box):
#   bash tools/side_seq.sh > out.txt
mkdir -p gpurun_out/ss_tmp && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ss_tmp -- python3 bench.py --steps 3 --warmup 3 --no-decode --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/ss_tmp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
qs = collections.Counter(r['Queue_Id'] for r in rows).most_common(2)
chain_q, side_q = qs[0][0], qs[1][0]
ch = [r for r in rows if r['Queue_Id'] == chain_q]
idx = [i for i, r in enumerate(ch) if 'adamw' in r['Kernel_Name']]
t_lo, t_hi = int(ch[idx[-2]]['End_Timestamp']), int(ch[idx[-1]]['End_Timestamp'])
t0 = int(ch[idx[-2] + 1]['Start_Timestamp'])
side = [r for r in rows if r['Queue_Id'] == side_q and t_lo <= int(r['Start_Timestamp']) < t_hi]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in side)
cbusy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in ch[idx[-2] + 1: idx[-1] + 1])
print(f"step window {(t_hi - t0) / 1e3:.1f} us; chain busy {cbusy / 1e3:.1f} us; side queue: {len(side)} kernels, busy {busy / 1e3:.1f} us, first start {(int(side[0]['Start_Timestamp']) - t0) / 1e3:.1f}, last end {(int(side[-1]['End_Timestamp']) - t0) / 1e3:.1f}")
agg = collections.defaultdict(lambda: [0, 0])
prev_end = None
for r in side:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {gap:7.2f}  {r['Kernel_Name'][:90]}")
    prev_end = e
    a = agg[r['Kernel_Name'][:60]]; a[0] += 1; a[1] += e - s
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"TOTAL {v[0]:4d} x  {v[1] / 1e3:9.1f} us  {k}")
PY
rm -rf gpurun_out/ss_tmp
