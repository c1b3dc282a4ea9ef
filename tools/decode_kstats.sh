# per-kernel durations of tools/decode_time.py under rocprofv3 (GPU box): bash tools/decode_kstats.sh <out.csv> [VAR=1 ...]
out=$1; shift
for v in "$@"; do export $v; done
mkdir -p gpurun_out/dk_tmp && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dk_tmp -- python3 tools/decode_time.py > gpurun_out/dk_tmp.log 2>&1
cp "$(ls gpurun_out/dk_tmp/*/*kernel_stats.csv | head -1)" "$out"
grep "per batch" gpurun_out/dk_tmp.log
rm -rf gpurun_out/dk_tmp
