# dense bf16 GEMM: every forced tile configuration on the shapes of tools/gemm_big.py (run on the GPU box;
# SATRN_SHAPES=1 / LATE_SHAPES=1 select the other shape lists)
for f in "" 128x128x1 128x128x2 128x64x1 128x64x2 64x64x1 64x64x2 64x64x4 64x32x1 64x32x2 64x32x4; do SATRN_KNOBS=gemm_force=$f python tools/gemm_big.py 2>&1 | grep "^M=" ; done
