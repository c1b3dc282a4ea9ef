"""Per-kernel shader-sequencer counters of one rocprofv3 --pmc pass (where do a kernel's wave cycles go?):
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU \
            SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...
  python tools/pmc_sq.py <dir> [name filter]
Prints each counter as a fraction of SQ_WAVE_CYCLES (cycles summed over resident waves) per kernel."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(set)
for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70] + " grid " + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r.get("Dispatch_Id"))
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
    if flt and flt not in k:
        continue
    wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    print(f"{k[:100]:100s} x{len(n[k]):3d} wave_cycles {wc:.3e} | " + " ".join(f"{nm.replace('SQ_', '')}={v / wc:.3f}" for nm, v in sorted(c.items()) if nm != "SQ_WAVE_CYCLES"))
