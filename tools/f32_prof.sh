# kernel table of the f32 (parity) mode's training step on the benchmark workload (GPU box):  bash tools/f32_prof.sh <out.csv>
mkdir -p gpurun_out/f32p && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f32p -- python3 bench.py --dtype f32 --steps 4 --warmup 2 --no-decode --no-cpu-baseline --no-extras > gpurun_out/f32p/line.log 2>&1
cp "$(ls gpurun_out/f32p/*/*kernel_stats.csv | head -1)" "$1"
grep '^{' gpurun_out/f32p/line.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 step', d['ms_per_step'], 'ms;', [(k['kernel'],k['launches'],k['ms']) for k in d['kernel_breakdown'][:8]])"
rm -rf gpurun_out/f32p/*/
