#!/bin/bash
# Round-end evidence, run ON the GPU box from the repo root (gpurun -- 'bash tools/round_profile.sh r01_v13'):
#   1. rocprofv3 --kernel-trace --stats of the bench command  -> profiles/<tag>_bench_kernel_stats.csv + _bench_line.log
#   2. two PMC passes (FETCH_SIZE / WRITE_SIZE, kernel-trace only) -> profiles/r01_pmc_traffic.json (read by bench.py)
#   3. the default bench line                                      -> profiles/<tag>_bench_default.json
set -e -o pipefail
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" gpurun_out/pmc_fetch gpurun_out/pmc_write
mkdir -p "$OUT" profiles
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 bench.py --steps 10 --warmup 4 --no-decode --no-cpu-baseline > "$OUT/bench_line.log" 2>&1
cp "$(ls "$OUT"/*/*kernel_stats.csv | head -1)" "profiles/${TAG}_bench_kernel_stats.csv"
grep '^{' "$OUT/bench_line.log" > "profiles/${TAG}_bench_line.log"
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 2 --no-decode --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 2 --no-decode --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1
echo "pmc write done"
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json
cp profiles/r01_pmc_traffic.json gpurun_out/
python3 bench.py > "$OUT/default.log" 2>&1
grep '^{' "$OUT/default.log" > "profiles/${TAG}_bench_default.json"
cp profiles/${TAG}_* gpurun_out/
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write "$OUT"/*/  # raw traces are large; the summaries are what is kept
echo "round profile done"
