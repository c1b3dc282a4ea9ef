# dense bf16 GEMM: every forced tile configuration on the large shapes (run on the GPU box)
for f in "" 128x128x1 128x128x2 128x64x1 128x64x2 64x64x1 64x64x2 64x64x4; do SATRN_GEMM_FORCE=$f python tools/gemm_big.py 2>&1 | grep "^M=" ; done
