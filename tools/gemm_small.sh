export SMALLN=1
for f in "" 64x32x4 64x32x2 64x32x1 64x64x2 64x64x4 128x64x2; do SATRN_KNOBS=gemm_force=$f python tools/gemm_bench.py fwd 2>&1 | grep "M=  1536 N=  256 K= 1536\|M=  6144 N=  160 K=  960"; done
