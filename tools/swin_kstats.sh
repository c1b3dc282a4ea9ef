# per-kernel durations of the SwinTRN step under rocprofv3 (GPU box): bash tools/swin_kstats.sh <out.csv> [VAR=1 ...]
out=$1; shift
for v in "$@"; do export $v; done
mkdir -p gpurun_out/swk_tmp && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/swk_tmp -- python3 tools/swin_time.py > gpurun_out/swk_tmp.log 2>&1
cp "$(ls gpurun_out/swk_tmp/*/*kernel_stats.csv | head -1)" "$out"
grep -o '"ms_per_step": [0-9.]*' gpurun_out/swk_tmp.log
rm -rf gpurun_out/swk_tmp
