# idle time of the chain's queue at the step boundaries of the bench loop (GPU box):  bash tools/step_boundary.sh [extra bench args]
mkdir -p gpurun_out/sb_tmp && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sb_tmp -- python3 bench.py --steps 12 --warmup 3 --no-decode --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/sb_tmp/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
q = collections.Counter(r['Queue_Id'] for r in rows).most_common(1)[0][0]
ch = [r for r in rows if r['Queue_Id'] == q]
idx = [i for i, r in enumerate(ch) if 'pack_all' in r['Kernel_Name']]
for i in idx[:-1]:
    end = int(ch[i]['End_Timestamp'])
    # first convolution of the next step
    j = next(k for k in range(i + 1, len(ch)) if 'stem_conv' in ch[k]['Kernel_Name'])
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in ch[i + 1:j])
    names = [r['Kernel_Name'][:28] for r in ch[i + 1:j]]
    print(f"boundary: {(int(ch[j]['Start_Timestamp']) - end) / 1e3:7.1f} us from pack_all end to stem conv start, {busy / 1e3:5.1f} us of it in kernels {names}")
PY
rm -rf gpurun_out/sb_tmp
