# ablation of the big GEMM on the SwinTRN stage-3 shapes (run on the GPU box): tile height and the timing-experiment bits
export SHAPES="9216,1536,384;9216,384,1536;9216,1152,384;9216,384,384;9216,2048,512"
for mt in 2 3 4; do
  for dbg in 0 1 2 4 8; do
    echo "== MT=$mt DBG=$dbg"
 SATRN_KNOBS=gemm_big=2,gemm_big_mt=$mt SATRN_TIMING=big_dbg=$dbg python3 tools/gemm_big.py 2>&1 | grep "M="
  done
done
echo "== old kernel"
SATRN_KNOBS=gemm_big=0 python3 tools/gemm_big.py 2>&1 | grep "M="
