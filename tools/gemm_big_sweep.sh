# EfficientSATRN step against the routing threshold of the big GEMM, and the late-stage shapes old vs big in isolation (GPU box)
for g in 0.3 0.6 1.0 1.5 2.0; do
  echo "== MIN_GFLOP=$g"
 SATRN_KNOBS=gemm_big_min_gflop=$g python3 bench.py --steps 20 --warmup 5 --no-decode --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(r['ms_per_step'], r.get('roofline'))"
done
echo "== late shapes old"
LATE_SHAPES=1 SATRN_KNOBS=gemm_big=0 python3 tools/gemm_big.py 2>&1 | grep "M="
for mt in 2 3; do
echo "== late shapes big MT=$mt"
LATE_SHAPES=1 SATRN_KNOBS=gemm_big=2,gemm_big_mt=$mt python3 tools/gemm_big.py 2>&1 | grep "M="
done
