# per-kernel average durations of the training step under rocprofv3 --stats for env variants (run on the GPU box):
#   bash tools/kstats.sh <kernel-name-regex> VAR=1 X=1 ...
mkdir -p gpurun_out/ks
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export $v
  rm -rf gpurun_out/ks/run
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks/run -- python3 bench.py --steps 10 --warmup 3 --no-decode --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "== $v"
  python3 - "$pat" <<'PY'
import csv, glob, re, sys
pat = re.compile(sys.argv[1])
for f in glob.glob('gpurun_out/ks/run/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat.search(r['Name']):
            print(f"{int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.2f} us  total {float(r['TotalDurationNs'])/1e6:8.3f} ms  {r['Name'][:90]}")
PY
  unset ${v%%=*}
done
rm -rf gpurun_out/ks/run
