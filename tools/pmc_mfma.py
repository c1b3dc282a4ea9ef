"""Aggregate a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace pass into per-kernel MFMA
utilisation -> JSON.   python tools/pmc_mfma.py <rocprof dir> <out.json> [label]

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (wall cycles x 1024 SIMDs), wall cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the
counter over the 8 XCDs: MI355X_MICROARCH.md, DVFS give-back) -- the fraction of the chip's matrix-pipe cycles that were busy
while the kernel ran.  busy_util uses SQ_BUSY_CYCLES (cycles with any wave resident, summed over the shader engines) as the
denominator instead: matrix-pipe share of the time the shader array was occupied at all."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<.*>)?", name)
    if name.startswith("_Z"):
        return name[:90]
    return (m.group(1) + (m.group(2) or ""))[:90] if m else name[:90]


def main():
    d, out = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else ""
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key)
                n[k] += 1
    dur = defaultdict(float)
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    rows = {}
    for k, c in acc.items():
        mf, busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        wall = gui / 8.0
        rows[k] = dict(dispatches=n[k], mfma_busy_cycles=mf, sq_busy_cycles=busy, grbm_gui_active=gui,
                       avg_us=round(dur.get(k, 0.0) / max(n[k], 1) / 1e3, 2),
                       mfma_util=round(mf / (wall * 1024.0), 5) if wall else None,
                       busy_util=round(mf / busy, 5) if busy else None)
    rows = dict(sorted(rows.items(), key=lambda kv: -kv[1]["mfma_busy_cycles"]))
    json.dump({"_note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace; " + label +
               "; mfma_util = MFMA_BUSY / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)", "kernels": rows}, open(out, "w"), indent=1)
    for k, v in list(rows.items())[:12]:
        print(k[:70], v["dispatches"], v["avg_us"], v["mfma_util"], v["busy_util"])


if __name__ == "__main__":
    main()
