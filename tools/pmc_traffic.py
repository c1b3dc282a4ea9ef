"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into HBM-side
bytes per launch for every kernel family -> profiles/r01_pmc_traffic.json (read by bench.py for roofline.traffic).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 2 --no-decode --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 2 --no-decode --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Counter unit: KB.  FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 B, MI355X_MICROARCH.md "HBM")."""
import csv
import glob
import json
import sys
from collections import defaultdict

FAMILIES = [("gemm_kernel", "launch_gemm"), ("conv3x3_halo_kernel", "launch_gemm"), ("gemm_skinny_kernel", "launch_gemm"), ("wgrad_kernel", "launch_wgrad"), ("bn_act_kernel", "launch_bn_act"),
            ("bn_bwd_apply_kernel", "launch_bn_bwd_apply"), ("BnBwdRedF", "launch_bn_bwd_reduce"), ("StatsF", "launch_colstats"),
            ("bn_act_pool_kernel", "launch_bn_act_pool"), ("bn_dw_img_kernel", "launch_bn_dwconv"), ("dw_bwd_img_kernel", "launch_dwconv_bwd_bn"),
            ("se_mlp_scale_kernel", "launch_se_mlp_scale"), ("se_bwd_gate_ds_kernel", "launch_se_bwd_wide"), ("se_bwd_pool_kernel", "launch_se_bwd_wide"),
            ("DwWgradF", "launch_dwconv_wgrad"), ("dwconv", "launch_dwconv"), ("se_fwd_kernel", "launch_se_fwd"),
            ("se_bwd_a_kernel", "launch_se_bwd"), ("se_bwd_b_kernel", "launch_se_bwd_weights"), ("attn_kernel", "launch_attn"),
            ("layernorm_bwd", "launch_layernorm_bwd"), ("layernorm", "launch_layernorm"), ("adamw_kernel", "launch_adamw"),
            ("pack_all_kernel", "launch_pack_all"), ("ce_", "launch_ce"), ("decode_pipe_kernel", "decode_pipe"), ("decode_greedy_kernel", "decode_greedy")]


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return None


def load(d, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            fam = family(r["Kernel_Name"])
            if fam:
                tot[fam] += float(r["Counter_Value"]) * 1024.0
                n[fam] += 1
    return tot, n


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    ft, fn = load(fetch_dir, "FETCH_SIZE")
    wt, wn = load(write_dir, "WRITE_SIZE")
    fams = {}
    for fam in sorted(ft, key=lambda k: -(2 * ft[k] + wt.get(k, 0.0))):
        nl = fn[fam]
        fb, wb = 2.0 * ft[fam] / nl, wt.get(fam, 0.0) / max(wn.get(fam, 1), 1)
        fams[fam] = dict(launches=nl, fetch_bytes_per_launch=round(fb), write_bytes_per_launch=round(wb), hbm_bytes_per_launch=round(fb + wb))
    note = (sys.argv[4] + " -- " if len(sys.argv) > 4 else "") + ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 2 --no-decode "
            "--no-cpu-baseline --no-extras` (5 steps incl. the profile step), bf16 B=32; aggregated by tools/pmc_traffic.py. FETCH_SIZE doubled per "
            "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); counter unit KB; Infinity-Cache hits are counted, so these "
            "are memory-side (fabric) bytes, an upper bound on HBM bytes.")
    json.dump({"_note": note, "families": fams}, open(out, "w"), indent=1)
    for k, v in list(fams.items())[:10]:
        print(k, v)


if __name__ == "__main__":
    main()
