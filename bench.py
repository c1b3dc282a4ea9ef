#!/usr/bin/env python
"""Headline benchmark: EfficientSATRN training step (BASELINE.json configs[1]): bf16, batch 32 per GPU, 1x128x384
synthetic images, teacher-forced seq_len 128, dropout 0.1, CE + backward + clip_grad_norm_(2.0) + AdamW(lr 5e-4).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Rank 0 prints ONE JSON line (contract in the round prompt): whole-job images/s, plus
  roofline     : the dominant kernel family of the step, timed live with HIP events on the engine's stream
  cpu_baseline : the CPU oracle (oracle/satrn_oracle.py, kind "port") timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16 (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
FLOP_PER_IMG_TRAIN = 20.17e9  # conv+GEMM, 2xMAC, forward 6.725 GFLOP/img x 3

CFG = dict(network="EfficientSATRN", rgb=1, enc_hidden=512, enc_filter=512, enc_heads=8, enc_layers=2, dec_src=512,
           dec_hidden=256, dec_filter=1024, dec_heads=8, dec_layers=3, num_classes=245)


class _DS:
    token_to_id = {"<SOS>": 0, "<EOS>": 1, "<PAD>": 2}
    id_to_token = {i: str(i) for i in range(245)}


def make_model(dtype, H, W, dropout):
    import satrn_amd
    flags = satrn_amd.Flags(dict(
        network="EfficientSATRN", input_size=dict(height=H, width=W),
        SATRN=dict(encoder=dict(hidden_dim=512, filter_dim=512, layer_num=2, head_num=8),
                   decoder=dict(src_dim=512, hidden_dim=256, filter_dim=1024, layer_num=3, head_num=8)),
        data=dict(rgb=1), dropout_rate=dropout)).get()
    return satrn_amd.EfficientSATRN(flags, _DS(), None, dtype=dtype)


def synth(B, H, W, T, seed, device):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 1, H, W, generator=g)
    exp = torch.randint(3, 245, (B, T + 1), generator=g)
    exp[:, 0] = 0
    exp[:, -1] = 1
    return img.to(device), exp.to(device)


def cpu_baseline(B, H, W, T, budget_s=15.0, max_steps=48):
    """The CPU oracle (PyTorch fp32 restatement pinned to the reference) on the host cores: fwd + CE + bwd + clip + AdamW."""
    from oracle import satrn_oracle as O
    # the GPU box gives a one-GPU job a 16-CPU share: more threads than that only oversubscribes
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, ncpu)))
    cfg = dict(O.CFG_EFF)
    sd = O.det_state_dict(cfg, 0)
    img, exp = O.det_inputs(B, 1, H, W, T, seed=21)
    names = O.trainable_names(cfg)
    m = {n: torch.zeros_like(sd[n]) for n in names}
    v = {n: torch.zeros_like(sd[n]) for n in names}
    times = []
    t_all = time.time()
    for it in range(max_steps):
        t0 = time.time()
        _, _, grads, _ = O.forward_backward(img, exp, sd, cfg)
        p = {n: sd[n] for n in names}
        O.clip_adamw_step(p, grads, m, v, it + 1, 5e-4)
        times.append(time.time() - t0)
        if time.time() - t_all > budget_s:
            break
    best = min(times[1:]) if len(times) > 1 else times[0]
    med = sorted(times[1:])[len(times[1:]) // 2] if len(times) > 1 else times[0]
    return dict(value=round(B / med, 3), unit="images/s", cores=torch.get_num_threads(), cpu=_cpu_model(), kind="port", best=round(B / best, 3),
                sample=f"oracle fp32 train step (fwd+CE+bwd+clip+AdamW), batch {B} of the same 1x{H}x{W}/T={T} workload, "
                       f"median of {len(times)} steps in {time.time() - t_all:.1f} s of CPU time")


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_decode_baseline(H, W, B=8, steps=231):
    """second half of the metric on the host cores: the oracle's reference-semantics greedy decode (like the reference it
    re-projects the whole output history every step, oracle.decoder_greedy_forward) for a bounded sample."""
    from oracle import satrn_oracle as O
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, ncpu)))
    cfg = dict(O.CFG_EFF)
    sd = O.det_state_dict(cfg, 0)
    img, _ = O.det_inputs(B, 1, H, W, 4, seed=5)
    t0 = time.time()
    with torch.no_grad():
        src = O.encoder_forward(img, sd, cfg, False)
        t1 = time.time()
        O.decoder_greedy_forward(src, steps, sd, cfg)
    t2 = time.time()
    return dict(value=round(B * steps / (t2 - t0), 1), unit="tokens/s", cores=torch.get_num_threads(), cpu=_cpu_model(), kind="port",
                sample=f"oracle fp32 greedy decode, batch {B} x {steps} steps of the same 1x{H}x{W} workload: encoder {t1 - t0:.2f} s + decoder {t2 - t1:.2f} s")


def precision_report(H, W, T, B, dev):
    """(a) what bf16 storage costs on the benchmarked configuration: the same weights and batch through the f32 engine (the
    parity mode: exact-f32 MFMA, <= 1e-3 of the reference's CPU path) and the bf16 engine, dropout off -- relative error of the
    logits, the loss and the flat gradient; (b) the f32 engine's own step time on the benchmark workload."""
    img, exp = synth(B, H, W, T, 21, dev)
    outs = {}
    f32 = None
    for dt in ("f32", "bf16"):
        torch.manual_seed(21)
        m = make_model(dt, H, W, 0.0).to(dev)
        m.train()
        logits = m(img, exp, True, 1.0)
        loss = m.criterion(logits.transpose(1, 2), exp[:, 1:])
        m.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        outs[dt] = (logits.detach().float().clone(), float(loss.item()), m.flat_grad().detach().clone())
        del m, logits, loss
        if dt == "f32":
            torch.manual_seed(21)
            m2 = make_model("f32", H, W, 0.1).to(dev)
            m2.train()
            for _ in range(3):
                m2.train_step(img, exp, 5e-4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 6
            for _ in range(n):
                m2.train_step(img, exp, 5e-4)
            torch.cuda.synchronize()
            fms = (time.perf_counter() - t0) / n * 1e3
            f32 = dict(ms_per_step=round(fms, 3), images_per_s=round(B / fms * 1e3, 1), steps=n,
                       mfma_frac=round(FLOP_PER_IMG_TRAIN * B / (fms * 1e-3) / 1e12 / PEAK_F32_TFLOPS, 4), peak_tflops=PEAK_F32_TFLOPS,
                       note="dtype f32: v_mfma_f32_16x16x4_f32 (exact f32), deterministic fixed-order reductions; the mode that carries the 1e-3 parity claim")
            del m2
    lf, lb = outs["f32"][0], outs["bf16"][0]
    gf, gb = outs["f32"][2], outs["bf16"][2]
    res = dict(logits_max_abs_err=round((lb - lf).abs().max().item(), 5), logits_rel_err=round(((lb - lf).abs().max() / lf.abs().max()).item(), 5),
               loss_f32=round(outs["f32"][1], 5), loss_bf16=round(outs["bf16"][1], 5),
               grad_rel_l2_err=round(((gb - gf).norm() / gf.norm()).item(), 5),
               grad_cosine=round((torch.dot(gb, gf) / (gb.norm() * gf.norm())).item(), 6),
               config="same weights / batch as the benchmark (B=32, 1x128x384, T=128), dropout off, one training forward + backward")
    return res, f32


def _decode_traffic():
    """memory-side bytes per decode STEP of the pipelined decoder, from the committed PMC passes (None if absent)"""
    for pf in DECODE_TRAFFIC_FILES:
        try:
            d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pf)))
            return round(d["families"]["decode_pipe"]["hbm_bytes_per_launch"] / 231.0)
        except Exception:  # noqa: BLE001
            continue
    return None


DECODE_TRAFFIC_FILES = ("r03_pmc_decode_traffic.json", "r02_pmc_decode_traffic.json")


def _decode_traffic_source():
    for pf in DECODE_TRAFFIC_FILES:
        if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pf)):
            return f"profiles/{pf} (memory-side bytes of the whole 231-step decode launch / 231)"
    return None


def swin_report(dev, B=16, T=128):
    """BASELINE configs[3]: SwinTRN (networks/SWIN.py: Swin-B/384 encoder + SWIN.yaml decoder) training step, bf16, batch 16.
    The reference cannot run 256x256 (PatchEmbed asserts 384, 64 patches % window 12 != 0: SURVEY section 2 row 3), so this is
    the reference's own 384x384 geometry."""
    import satrn_amd
    flags = satrn_amd.Flags(dict(network="SWIN", input_size=dict(height=384, width=384),
                                 SATRN=dict(encoder=dict(hidden_dim=300, filter_dim=600, layer_num=6, head_num=8),
                                            decoder=dict(src_dim=1024, hidden_dim=512, filter_dim=512, layer_num=4, head_num=8)),
                                 data=dict(rgb=3), dropout_rate=0.1)).get()
    torch.manual_seed(21)
    m = satrn_amd.SWIN(flags, _DS(), True, dtype="bf16").to(dev)
    m.train()
    g = torch.Generator().manual_seed(5)
    img = torch.randn(B, 3, 384, 384, generator=g).to(dev)
    exp = torch.randint(3, 245, (B, T + 1), generator=g)
    exp[:, 0] = 0
    exp[:, -1] = 1
    exp = exp.to(dev)
    for _ in range(3):
        m.train_step(img, exp, 5e-4)
    torch.cuda.synchronize()
    n = 8
    t0 = time.perf_counter()
    for _ in range(n):
        m.train_step(img, exp, 5e-4)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    loss = m.read_loss()[0]
    prof = m.profile_step(img, exp)
    fl = sum(p_["flops"] for p_ in prof)
    by = sum(p_["bytes"] for p_ in prof)
    top = [dict(kernel=p_["kernel"], launches=p_["launches"], ms=round(p_["ms"], 3), gflop=round(p_["flops"] / 1e9, 1),
                mfma_frac=round(p_["flops"] / max(p_["ms"], 1e-6) / 1e9 / PEAK_BF16_TFLOPS, 4)) for p_ in prof[:14]]
    del m
    return dict(workload="SwinTRN train step (fwd+CE+bwd+clip+AdamW), bs16, 3x384x384, teacher-forced T=128, dropout 0.1, drop_path 0.5 (BASELINE configs[3] at the reference's 384 geometry)",
                ms_per_step=round(ms, 3), images_per_s=round(B / ms * 1e3, 1), steps=n, dtype="bf16", final_loss=round(loss, 4),
                step_gflop=round(fl / 1e9, 1), step_algorithmic_mbytes=round(by / 1e6, 1),
                mfma_frac=round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), hbm_frac=round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                launches=sum(p_["launches"] for p_ in prof), top_families=top)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def visible_gpu_count():
    """GPUs this process may use, from sysfs (/sys/class/kfd/kfd/topology/nodes/*/properties: simd_count > 0) clipped by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES; None when the topology cannot be read.  No HIP call."""
    import glob
    total = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        # no readable topology: 0 only when the compute device node is missing too (no amdgpu compute driver at all); a container that
        # passes /dev/kfd through without the kfd sysfs tree skips the check and lets the children report
        return None if os.path.exists("/dev/kfd") else 0
    for f in nodes:
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            total += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            total = min(total, len([x for x in v.split(",") if x.strip() != ""]))
    return total


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a child, pass its stderr through, print the ONE JSON line rank 0 produced and check it really came from N ranks."""
    import socket
    import subprocess
    n = args.gpus
    # Count GPUs WITHOUT the HIP runtime: this process goes on to start a launcher, and a parent that has opened /dev/kfd
    # (torch.cuda.device_count() can, through hipGetDeviceCount) must not spawn from that state on this pool.  The KFD topology in
    # sysfs lists every node; GPU nodes have simd_count > 0.  If sysfs is unreadable the check is skipped: the children fail and say so.
    have = visible_gpu_count()
    if have is not None and have < n:
        log(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible")
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            log(ln)
    if proc.returncode != 0 or line is None:
        log(f"bench.py: the {n}-rank job failed (exit code {proc.returncode}, JSON line {'found' if line else 'missing'})")
        return proc.returncode or 3
    got = json.loads(line)
    if got.get("n_gpus") != n or got.get("rccl_ranks") != n:
        log(f"bench.py: asked for {n} ranks, the line reports n_gpus={got.get('n_gpus')} rccl_ranks={got.get('rccl_ranks')}")
        return 4
    print(line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step as one hipGraph (single chain) instead of eager two-stream launches")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the bf16-vs-f32 error figures and the f32-mode timing")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched bare (no torchrun): start the N ranks ourselves as CHILD processes -- before this process touches the
        # GPU -- relay rank 0's JSON line and leave with the launcher's exit code (never exec-replace: see the round notes)
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    H, W, T, B = 128, 384, 128, args.batch
    torch.manual_seed(21)
    model = make_model(args.dtype, H, W, 0.1).to(dev)
    if world > 1:  # identical initial weights on every rank
        model._ensure_bound(dev)
        dist.broadcast(model.flat_params(), 0)
    model.train()
    img, exp = synth(B, H, W, T, 21 + rank, dev)
    lr = 5e-4
    graph = args.graph

    from satrn_amd import dp

    def step():
        dp.dp_train_step(model, img, exp, lr, use_graph=graph)

    for i in range(max(args.warmup, 2)):
        step()
        torch.cuda.synchronize()
        if rank == 0:
            log(f"warmup {i} done, loss {model.read_loss()[0]:.4f}")
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # one event per step on the stream the step is issued on (the engine's streams are joined to it at the end of every step):
    # per-step GPU times for the median without a host sync inside the timed region
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    # host time per step inside the back-to-back timed loop.  This is NOT what the host needs: once the launch queues are full the
    # host blocks until the GPU retires packets, so in steady state this figure converges to the GPU's own step time
    # (back-pressure).  The host's real cost is host_issue_ms_unblocked below (one step issued into an idle GPU, no sync).
    host_issue_ms = (time.perf_counter() - t0) / args.steps * 1e3
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss, cnt, gnorm = model.read_loss()
    if rank == 0:
        log(f"timed {args.steps} steps in {dt:.3f}s")
    # outside the timed region: the host's own cost of issuing ONE step (GPU idle before, no sync after): median of 7
    unb = []
    for _ in range(7):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        unb.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    host_issue_unblocked_ms = sorted(unb)[len(unb) // 2]
    exposed = dp.last_exchange_exposed_ms() if world > 1 else None

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
        ms_median = per_step[len(per_step) // 2]
        # ---- roofline of the dominant kernel family: live HIP-event timing of every launch of one eager step
        prof = model.profile_step(img, exp)
        log("profile:", json.dumps(prof))
        tot_ms = sum(p["ms"] for p in prof)
        dom = prof[0]
        # the dominant family is priced against BOTH limits; the binding one is the one it sits closer to (for this network's
        # GEMMs -- small K, small N, huge M in the early stages; tiny everywhere else -- that is HBM, not the matrix pipe)
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        ach_f = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        ach_b = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        if ach_f / peak >= ach_b / PEAK_HBM_GBS:
            roof = dict(bound="mfma", achieved=round(ach_f, 3), peak=peak, unit="TFLOP/s", frac=round(ach_f / peak, 5), traffic=None)
        else:
            roof = dict(bound="hbm", achieved=round(ach_b, 2), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach_b / PEAK_HBM_GBS, 5), traffic=None)
        roof["mfma_frac"] = round(ach_f / peak, 5)
        roof["hbm_frac"] = round(ach_b / PEAK_HBM_GBS, 5)
        roof["kernel"] = dom["kernel"]
        pmc_all = {}
        for pf in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:  # HBM bytes per launch of every kernel family from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE)
                pmc_all = json.load(open(os.path.join(ROOT, "profiles", pf)))["families"]
                roof["traffic_source"] = f"profiles/{pf} (rocprofv3 --pmc passes of this command, collected with tools/round_profile.sh; NOT measured in this run)"
                break
            except Exception:
                continue
        if dom["kernel"] in pmc_all:
            roof["traffic"] = pmc_all[dom["kernel"]]["hbm_bytes_per_launch"]
        roof["algorithmic_bytes_per_launch"] = round(dom["bytes"] / max(dom["launches"], 1))
        # every kernel family of the step against the roofline that bounds it (flops -> MFMA peak of the dtype, bytes -> HBM):
        # frac = the larger of the two fractions, i.e. how close the family runs to whichever limit it is nearer to
        fams = []
        peak_fl = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        for p_ in prof:
            sec = max(p_["ms"], 1e-6) * 1e-3
            ff, fb = p_["flops"] / sec / 1e12 / peak_fl, p_["bytes"] / sec / 1e9 / PEAK_HBM_GBS
            fams.append(dict(kernel=p_["kernel"], launches=p_["launches"], ms=round(p_["ms"], 3), gflop=round(p_["flops"] / 1e9, 2),
                             mbytes=round(p_["bytes"] / 1e6, 1), mfma_frac=round(ff, 4), hbm_frac=round(fb, 4),
                             bound="mfma" if ff >= fb else "hbm", frac=round(max(ff, fb), 4),
                             pmc_hbm_mbytes_per_launch=(round(pmc_all[p_["kernel"]]["hbm_bytes_per_launch"] / 1e6, 2) if p_["kernel"] in pmc_all else None)))
        roof["families"] = fams
        roof["step_algorithmic_mbytes"] = round(sum(p_["bytes"] for p_ in prof) / 1e6, 1)
        roof["step_hbm_bound_ms"] = round(sum(p_["bytes"] for p_ in prof) / (PEAK_HBM_GBS * 1e9) * 1e3, 3)
        roof["step_mfma_bound_ms"] = round(FLOP_PER_IMG_TRAIN * B / (peak_fl * 1e12) * 1e3, 3)
        roof["launches_per_step"] = dom["launches"]
        roof["avg_launch_us"] = round(dom["ms"] * 1e3 / max(dom["launches"], 1), 3)
        roof["share_of_step_kernel_time"] = round(dom["ms"] / max(tot_ms, 1e-9), 4)
        roof["whole_step_mfma_frac"] = round(FLOP_PER_IMG_TRAIN * B / (ms * 1e-3) / 1e12 / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS), 5)
        out = dict(metric="train images/sec (whole node) EfficientSATRN bs32/GPU 128x384", value=round(value, 2), unit="images/s",
                   n_gpus=world, rccl_ranks=(dist.get_world_size() if dist is not None else 1), steps=args.steps, warmup=args.warmup, ms_per_step=round(ms, 3), ms_per_step_median=round(ms_median, 3), host_issue_ms_per_step=round(host_issue_ms, 3), host_issue_ms_unblocked=round(host_issue_unblocked_ms, 3),
                   host_issue_note="host_issue_ms_per_step is measured inside the back-to-back loop and includes queue back-pressure (the host waits for the GPU once the launch queues fill); host_issue_ms_unblocked is one step issued into an idle GPU without a sync = what the host itself needs",
                   exchange_exposed_ms=(round(exposed, 3) if exposed is not None else None), higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype=args.dtype, data="synthetic",
                   config=dict(workload="EfficientSATRN train step (fwd+CE+bwd+clip+AdamW), bs32/GPU, 1x128x384, teacher-forced T=128, dropout 0.1 (BASELINE configs[1])",
                               global_batch=world * B, seq_len=T, parallelism=f"dp{world}", hipgraph=graph, streams=1 if graph else 2,
                               exchange=("none" if world == 1 else ("one flat all-reduce" if graph else "backward cut after the last backbone stage: 74 % of the gradient all-reduced (async RCCL) beside the rest of the backward"))),
                   roofline=roof, final_loss=round(loss, 4), grad_norm=round(gnorm, 4),
                   kernel_breakdown=[dict(kernel=p["kernel"], launches=p["launches"], ms=round(p["ms"], 3)) for p in prof[:12]])
        if world == 1 and not args.no_decode:
            # the extra measurements must never cost the headline line: a failure is reported IN the JSON (and on stderr)
            try:
                # second half of BASELINE's metric: KV-cached greedy decode, EfficientSATRN, batch 64, 231 steps (configs[4])
                log("greedy decode ...")
                model.eval()
                dimg, _ = synth(64, H, W, 4, 5, dev)
                model.greedy(dimg, 231)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                reps = 3
                for _ in range(reps):
                    model.greedy(dimg, 231)
                torch.cuda.synchronize()
                dsec = (time.perf_counter() - t1) / reps
                dpath, dgive, dnote = model.last_decode_path()
                if dpath != "pipe" or dgive:
                    raise RuntimeError(f"the benchmarked decode did not run on the role pipeline: path {dpath!r}, give-ups {dgive}, note {dnote!r}")
                out["greedy_decode"] = dict(value=round(64 * 231 / dsec, 1), unit="tokens/s", batch=64, steps=231,
                                            ms_per_batch=round(dsec * 1e3, 2), includes="encoder + 231 decoder steps",
                                            decoder_path=dpath, pipe_giveups=dgive)
                # HBM roofline of the decode (38 MB of weights + mean KV history per 64-image step, bf16), with the
                # encoder pass timed on its own and taken out
                model.encode(dimg)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    model.encode(dimg)
                torch.cuda.synchronize()
                esec = (time.perf_counter() - t1) / reps
                step_us = max(dsec - esec, 1e-9) / 231 * 1e6
                out["greedy_decode"]["roofline"] = dict(bound="hbm", algorithmic_bytes_per_step=38.0e6, us_per_step=round(step_us, 1),
                                                        achieved=round(38.0e6 / step_us / 1e3, 1), peak=8000.0, unit="GB/s",
                                                        frac=round(38.0e6 / step_us / 1e3 / 8000.0, 4), encoder_ms=round(esec * 1e3, 2),
                                                        kv_only=dict(algorithmic_bytes_per_step=32.2e6, achieved=round(32.2e6 / step_us / 1e3, 1), frac=round(32.2e6 / step_us / 1e3 / 8000.0, 4),
                                                                     note="what a weight-stationary decoder must move per 64-image step: self-attention K/V history 22.8 MB (mean over 231 steps) + cross-attention K/V 9.4 MB; the 5.8 MB of weights stay in LDS"),
                                                        traffic=_decode_traffic(), traffic_source=_decode_traffic_source(),
                                                        kernel="decode_pipe_kernel (one persistent workgroup per decoder role, weights resident in LDS, images pipelined through the roles)",
                                                        note="latency-bound, not HBM-bound: a token is a dependent chain of 13 role hops per step (3 layers x [Q/K/V, self-attention + out-projection, LayerNorm + cross-attention, LayerNorm + feed-forward] + generator), and at batch 64 the 244 role workgroups are ~80 % busy; the weights never leave LDS, so the 38 MB/step figure is what a weight-streaming decoder would move, kept as the algorithmic unit of SURVEY 8d")
                # the per-image kernel of round 1 (one workgroup per image streams every weight each step), kept as the
                # fallback for shapes the pipeline does not take; timed beside it
                from satrn_amd import switches as sw
                sw.off("decode_pipe")
                try:
                    model.greedy(dimg, 231)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    model.greedy(dimg, 231)
                    torch.cuda.synchronize()
                    out["greedy_decode"]["per_image_kernel"] = dict(ms_per_batch=round((time.perf_counter() - t1) * 1e3, 2), decoder_path=model.last_decode_path()[0],
                                                                    note="round-1 decoder (fallback path), same batch")
                finally:
                    sw.on("decode_pipe")
                # the same decode with the DecodingManager rules evaluated inside the decode kernel (the reference's default
                # at inference, inference.py:48); rule table = the reference RULES as compiled into tests/golden/rules.npz
                rules_npz = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "rules.npz")
                if os.path.exists(rules_npz):
                    import numpy as np
                    import satrn_amd
                    table = np.load(rules_npz)["table"]

                    class _M:
                        tokens = ["<SOS>", "<EOS>"] + [f"t{i}" for i in range(len(table) - 10)]
                        rules = {}
                    mgr = satrn_amd.DeviceDecodingManager(_M())
                    mgr._table_host = table.astype(np.int32)
                    model.decoder.manager = mgr
                    model.greedy(dimg, 231)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(reps):
                        model.greedy(dimg, 231)
                    torch.cuda.synchronize()
                    msec = (time.perf_counter() - t1) / reps
                    model.decoder.manager = None
                    out["greedy_decode"]["with_decoding_manager"] = dict(value=round(64 * 231 / msec, 1), unit="tokens/s",
                                                                         ms_per_batch=round(msec * 1e3, 2))
                # larger batches, reported beside the BASELINE batch, not instead of it: 112 images is the most the pipeline's
                # mailboxes take; above that the engine falls back to the per-image kernel (one workgroup per image)
                dimg112 = torch.cat([dimg, dimg[:48]])
                model.greedy(dimg112, 231)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                model.greedy(dimg112, 231)
                torch.cuda.synchronize()
                d112 = time.perf_counter() - t1
                out["greedy_decode"]["batch_112"] = dict(value=round(112 * 231 / d112, 1), unit="tokens/s", ms_per_batch=round(d112 * 1e3, 2), decoder_path=model.last_decode_path()[0])
                dimg4 = torch.cat([dimg] * 4)
                model.greedy(dimg4, 231)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                model.greedy(dimg4, 231)
                torch.cuda.synchronize()
                d4 = time.perf_counter() - t1
                out["greedy_decode"]["batch_256"] = dict(value=round(256 * 231 / d4, 1), unit="tokens/s", ms_per_batch=round(d4 * 1e3, 2), decoder_path=model.last_decode_path()[0])
                # best-first beam search of the same 64 images (EfficientSATRN.beam_search, beam 5, max_sequence 230): one launch,
                # at most 229 decoder-step expansions per image
                class _Loader:
                    class dataset:
                        token_to_id = {"<SOS>": 0, "<EOS>": 1, "<PAD>": 2}
                model.beam_search(dimg, _Loader, beam_width=5, max_sequence=230)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                model.beam_search(dimg, _Loader, beam_width=5, max_sequence=230)
                torch.cuda.synchronize()
                db = time.perf_counter() - t1
                out["greedy_decode"]["beam_search"] = dict(beam_width=5, max_sequence=230, batch=64, ms_per_batch=round(db * 1e3, 2),
                                                           includes="encoder + up to 229 expansions per image + back-trace + D2H")
                model.train()
            except Exception as ex:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                out.setdefault("greedy_decode", {})["error"] = repr(ex)
                model.train()
        if world == 1 and not args.no_extras:
            # the reference's own training schedule flips a coin per batch (teacher_forcing_ratio 0.8 -> 0.3 over the epochs,
            # configs/EfficientSATRN.yaml:35-37, train_modules/train_single_opt.py:75): the non-teacher-forced branch
            # (networks/EfficientSATRN.py:496-525) through the same fused step, and the schedule's mean step at ratio 0.55
            try:
                model.train()
                for _ in range(2):
                    model.train_step(img, exp, lr, teacher_forced=False)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                nar = 4
                for _ in range(nar):
                    model.train_step(img, exp, lr, teacher_forced=False)
                torch.cuda.synchronize()
                ar_ms = (time.perf_counter() - t1) / nar * 1e3
                out["train_tf_schedule"] = dict(teacher_forced_ms=round(ms, 3), autoregressive_ms=round(ar_ms, 3), ratio=round(ar_ms / ms, 2),
                                                mean_ms_at_tf_0_55=round(0.55 * ms + 0.45 * ar_ms, 3),
                                                images_per_s_at_tf_0_55=round(B / (0.55 * ms + 0.45 * ar_ms) * 1e3, 1),
                                                note="autoregressive branch = 127 dependent decoder steps with gradients (networks/EfficientSATRN.py:496-525) in two launches (kernels_ar.hip): forward = 4 weight slices per image, one exchange per block; backward = one workgroup per image and layer, pipelined over the steps; weight gradients as products over [B*T]-row slabs. The operator-level form (SATRN_OFF=ar_fused, ~126 launches per step) measured 95 ms. Reachable from the fused / data-parallel step (train_step(teacher_forcing_ratio=...), rank-shared coin)")
            except Exception as ex:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                out["train_tf_schedule"] = dict(error=repr(ex))
            try:
                out["accuracy_bf16_vs_f32"], out["f32_mode"] = precision_report(H, W, T, B, dev)
            except Exception as ex:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                out["accuracy_bf16_vs_f32"] = dict(error=repr(ex))
        if world == 1 and not args.no_extras:
            log("swin ...")
            try:
                out["swin_trn"] = swin_report(dev)
            except Exception as ex:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                out["swin_trn"] = dict(error=repr(ex))
        if not args.no_cpu_baseline and world == 1 and not args.no_decode:
            log("cpu decode baseline ...")
            try:
                out.setdefault("greedy_decode", {})["cpu_baseline"] = cpu_decode_baseline(H, W)
            except Exception as ex:  # noqa: BLE001
                out.setdefault("greedy_decode", {})["cpu_baseline"] = dict(error=repr(ex))
        if not args.no_cpu_baseline and world == 1:
            log("cpu baseline ...")
            try:
                out["cpu_baseline"] = cpu_baseline(8, H, W, T)
            except Exception as ex:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                out["cpu_baseline"] = dict(error=repr(ex))
        # the second half of BASELINE's metric goes LAST in the line (a reader that keeps only the tail still has it)
        gd = out.get("greedy_decode", {})
        if "value" in gd:
            out["greedy_decode_summary"] = dict(metric="greedy-decode tok/s (EfficientSATRN, batch 64, max_sequence 230, KV-cached HIP decoder; BASELINE configs[4])",
                                                value=gd["value"], unit="tokens/s", ms_per_batch=gd.get("ms_per_batch"), decoder_path=gd.get("decoder_path"),
                                                pipe_giveups=gd.get("pipe_giveups"))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
