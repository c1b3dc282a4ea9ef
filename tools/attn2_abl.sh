# where the fused attention backward spends its time inside the SwinTRN step (GPU box): kernel averages under the SATRN_TIMING=a2_dbg=<bits> experiments
for d in 0 6 14 30 62 2 4; do
 bash tools/swin_kstats.sh gpurun_out/a2abl_$d.csv SATRN_OFF=side_stream SATRN_TIMING=a2_dbg=$d > /dev/null 2>&1
  python3 - $d <<'PY'
import csv, sys
d = sys.argv[1]
for r in csv.DictReader(open(f'gpurun_out/a2abl_{d}.csv')):
    if 'attn2_bwd_kernel<32>' in r['Name']:
        print(f"DBG={d:>3s}  attn2_bwd<32>: calls {int(r['Calls'])}  avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  max {float(r['MaxNs'])/1e3:8.2f}")
PY
done
