# per-kernel durations of the autoregressive training step under rocprofv3 --stats (GPU box):  bash tools/ar_kstats.sh [B] [T]
mkdir -p gpurun_out/ark
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ark/run
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ark/run -- python3 tools/ar_time.py ${1:-32} ${2:-128} > gpurun_out/ark/time.log 2>&1
cat gpurun_out/ark/time.log
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/ark/run/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"{int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:10.2f} us  total {float(r['TotalDurationNs'])/1e6:9.3f} ms  {r['Name'][:100]}")
PY
rm -rf gpurun_out/ark/run
