# ablation of the big weight-gradient kernel on the SwinTRN stage-3 shapes (GPU box): timing-experiment bits, with / without bias gradient
export SHAPES="9216,1536,384;9216,384,1536;9216,1152,384;9216,384,384;36864,768,192;147456,384,96"
for wb in 1 ""; do
for dbg in 0 1 2 4 8; do
  echo "== WITH_BIAS=$wb DBG=$dbg"
 WITH_BIAS=$wb SATRN_TIMING=big_dbg=$dbg python3 tools/wgrad_big.py 2>&1 | grep "M="
done
done
echo "== old kernel"
SATRN_KNOBS=wgrad_big=0 python3 tools/wgrad_big.py 2>&1 | grep "M="
