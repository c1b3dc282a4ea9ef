"""Per-kernel timeline of ONE training step from a rocprofv3 --kernel-trace CSV (run on the GPU box):

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 3 --warmup 2 \
        --no-decode --no-cpu-baseline --no-extras
    python3 tools/chain_trace.py gpurun_out/trace > gpurun_out/trace_summary.txt

Prints, for the last step (delimited by the AdamW launches), every dispatch of the busiest queue (the main chain) in start
order: start offset, duration, gap to the previous dispatch's end, grid, name -- plus totals of execution time and gaps.  The
question it answers: is a late-stage kernel's ~10 us its own execution or the seam between two dependent launches?"""
import csv
import glob
import sys
from collections import Counter, defaultdict


def short(name):
    name = name.replace("_Z", "")
    for k in ("gemm_kernel", "wgrad_kernel", "bn_act_kernel", "bn_bwd_apply_kernel", "colreduce_kernel", "conv3x3_halo_kernel",
              "se_fwd_kernel", "se_bwd_a_kernel", "se_bwd_b_kernel", "dwconv_s1_kernel", "dwconv_kernel", "hw_reduce_kernel", "bcast_kernel",
              "layernorm_bwd_kernel", "layernorm_kernel", "attn_kernel", "attn_bwd", "adamw_kernel", "colsum_kernel", "act_bwd_kernel"):
        if k in name:
            i = name.find(k)
            return name[i:i + 60]
    return name[:60]


def main():
    d = sys.argv[1]
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
    # steps end with a burst of AdamW launches: split on gaps between the bursts
    bursts = []
    for i in ad:
        if not bursts or i - bursts[-1][-1] > 50:
            bursts.append([i])
        else:
            bursts[-1].append(i)
    if len(bursts) < 2:
        print("need two optimizer bursts in the trace")
        return
    lo, hi = bursts[-2][-1] + 1, bursts[-1][-1] + 1
    step = rows[lo:hi]
    t0 = int(step[0]["Start_Timestamp"])
    qcount = Counter(r["Queue_Id"] for r in step)
    mainq = qcount.most_common(1)[0][0]
    print(f"# step = dispatches {lo}..{hi} ({len(step)}), queues {dict(qcount)}, main queue {mainq}, wall {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
    prev_end = None
    texec = tgap = 0.0
    per = defaultdict(lambda: [0, 0.0, 0.0])
    allq = len(sys.argv) > 2 and sys.argv[2] == "all"
    for r in step:
        if r["Queue_Id"] != mainq:
            if allq:
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                print(f"{(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.2f}   [queue {r['Queue_Id']}] end {(e - t0) / 1e3:9.1f} {short(r['Kernel_Name'])}")
            continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        dur = (e - s) / 1e3
        prev_end = e
        texec += dur
        tgap += max(gap, 0.0)
        nm = short(r["Kernel_Name"])
        p = per[nm]
        p[0] += 1; p[1] += dur; p[2] += max(gap, 0.0)
        grid = r.get("Grid_Size_X", "?"); wg = r.get("Workgroup_Size_X", "?")
        print(f"{(s - t0) / 1e3:9.1f} dur {dur:7.2f} gap {gap:6.2f} grid {grid:>8s}/{wg:>4s} {nm}")
    print(f"# main queue: exec {texec / 1e3:.3f} ms, gaps {tgap / 1e3:.3f} ms")
    print("# per kernel on the main queue: launches, exec us, gap-before us")
    for nm, p in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"# {p[0]:5d} {p[1]:9.1f} {p[2]:9.1f}  {nm}")


if __name__ == "__main__":
    main()
