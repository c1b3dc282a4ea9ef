# per-CU LDS fill rate of the persistent GEMM against the number of CUs in use (GPU box): is the main loop bound per CU or by the shared L2 / fabric?
export SHAPES="9216,2048,512;9216,1536,384;4096,4096,4096"
for g in 256 128 64 32 16; do for dbg in 1; do echo "== GRID=$g DBG=$dbg (no epilogue)"; SATRN_KNOBS=big_grid=$g,gemm_big=2,gemm_big_mt=3 SATRN_TIMING=big_dbg=$dbg python3 tools/gemm_big.py 2>&1 | grep "M="; done; done
