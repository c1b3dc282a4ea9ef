# A/B of engine environment knobs on the training step (run on the GPU box): prints ms/step, median and host issue time
mkdir -p gpurun_out/ab
B="python3 bench.py --steps 40 --warmup 6 --no-decode --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" $B 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$name', d['ms_per_step'], d['ms_per_step_median'], 'host', d.get('host_issue_ms_per_step'))"; }
for v in "$@"; do run "$v" $v; done
