#!/bin/bash
# Round-end evidence, run ON the GPU box from the repo root (gpurun -- 'bash tools/round_profile.sh r02'):
#   1. rocprofv3 --kernel-trace --stats of the bench command  -> profiles/<tag>_bench_kernel_stats.csv + _bench_line.log
#   2. two PMC passes (FETCH_SIZE / WRITE_SIZE, kernel-trace only) -> profiles/<tag>_pmc_traffic.json (read by bench.py)
#   3. MFMA-busy PMC pass of the whole step and of the encoder self-attention region alone -> profiles/<tag>_mfma_busy*.json
#   4. kernel stats of the SwinTRN step and of the greedy decode     -> profiles/<tag>_swin_kernel_stats.csv, _decode_kernel_stats.csv
#   4b. kernel stats of the autoregressive training step                -> profiles/<tag>_ar_kernel_stats.csv, _ar_time.log
#   5. the default bench line                                      -> profiles/<tag>_bench_default.json
# Every rocprofv3 command has the program itself after `--` and never mixes --pmc with the trace domains gpurun refuses.
set -e -o pipefail
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/pmc_region
mkdir -p "$OUT" profiles
BENCH="bench.py --steps 10 --warmup 4 --no-decode --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $BENCH > "$OUT/bench_line.log" 2>&1
cp "$(ls "$OUT"/*/*kernel_stats.csv | head -1)" "profiles/${TAG}_bench_kernel_stats.csv"
grep '^{' "$OUT/bench_line.log" > "profiles/${TAG}_bench_line.log"
echo "stats done"
SHORT="bench.py --steps 2 --warmup 2 --no-decode --no-cpu-baseline --no-extras"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 $SHORT > gpurun_out/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 $SHORT > gpurun_out/pmc_write.log 2>&1
echo "pmc write done"
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write "profiles/${TAG}_pmc_traffic.json"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python3 $SHORT > gpurun_out/pmc_mfma.log 2>&1
python3 tools/pmc_mfma.py gpurun_out/pmc_mfma "profiles/${TAG}_mfma_busy_step.json" "whole training step ($SHORT), bf16 B=32"
echo "pmc mfma (step) done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_region -- python3 tools/enc_attn_region.py --iters 20 > gpurun_out/pmc_region.log 2>&1
python3 tools/pmc_mfma.py gpurun_out/pmc_region "profiles/${TAG}_mfma_busy.json" "encoder self-attention region alone (tools/enc_attn_region.py --iters 20: LN -> QKV -> attention -> out-proj, fwd + bwd), bf16 B=32"
python3 tools/enc_attn_region.py > "profiles/${TAG}_enc_attn_region.json" 2> /dev/null
python3 tools/enc_attn_region.py --batch 1024 --iters 50 > "profiles/${TAG}_enc_attn_region_b1024.json" 2> /dev/null
echo "pmc mfma (region) done"
# greedy decode (64 x 231, pipelined): memory-side bytes of the one decode launch
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_dfetch -- python3 tools/decode_time.py > gpurun_out/pmc_dfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_dwrite -- python3 tools/decode_time.py > gpurun_out/pmc_dwrite.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_dfetch gpurun_out/pmc_dwrite "profiles/${TAG}_pmc_decode_traffic.json" "greedy decode 64 x 231 (tools/decode_time.py: per-image kernel then pipelined decoder, 2 + 4 launches each); bytes are per LAUNCH = per whole decode"
rm -rf gpurun_out/pmc_dfetch gpurun_out/pmc_dwrite
echo "pmc decode done"
# SwinTRN step and greedy decode: per-kernel tables (the two workloads beside the headline one)
rm -rf gpurun_out/kt_swin gpurun_out/kt_dec
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_swin -- python3 tools/swin_time.py > "profiles/${TAG}_swin_step.json.log" 2>&1
cp "$(ls gpurun_out/kt_swin/*/*kernel_stats.csv | head -1)" "profiles/${TAG}_swin_kernel_stats.csv"
grep '^{' "profiles/${TAG}_swin_step.json.log" > "profiles/${TAG}_swin_step.json"; rm -f "profiles/${TAG}_swin_step.json.log"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_swin -- python3 tools/swin_time.py > gpurun_out/pmc_swin.log 2>&1
python3 tools/pmc_mfma.py gpurun_out/pmc_swin "profiles/${TAG}_mfma_busy_swin.json" "SwinTRN training step (tools/swin_time.py), bf16 bs16 384x384"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_dec -- python3 tools/decode_time.py > "profiles/${TAG}_decode_time.log" 2>&1
cp "$(ls gpurun_out/kt_dec/*/*kernel_stats.csv | head -1)" "profiles/${TAG}_decode_kernel_stats.csv"
rm -rf gpurun_out/kt_swin gpurun_out/kt_dec gpurun_out/pmc_swin
echo "swin / decode kernel stats done"
# autoregressive training branch (bs32, T = 128): its two launches beside the rest of the step
rm -rf gpurun_out/kt_ar
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_ar -- python3 tools/ar_time.py > "profiles/${TAG}_ar_time.log" 2>&1
cp "$(ls gpurun_out/kt_ar/*/*kernel_stats.csv | head -1)" "profiles/${TAG}_ar_kernel_stats.csv"
rm -rf gpurun_out/kt_ar
echo "ar kernel stats done"
python3 bench.py > "$OUT/default.log" 2>&1
grep '^{' "$OUT/default.log" > "profiles/${TAG}_bench_default.json"
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/pmc_region "$OUT"/*/  # raw traces are large; the summaries are what is kept
echo "round profile done"
