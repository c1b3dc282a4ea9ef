"""Time the BatchNorm-apply (+activation, +residual) pass on the backbone's shapes, cache-cold (rotating buffer sets)."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satrn_amd
lib = satrn_amd._lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
NSET = int(os.environ.get("NSET", 12))
for M, C, res, act in [(98304, 48, 1, 0), (98304, 48, 0, 0), (385056, 24, 1, 0), (385056, 24, 0, 2), (98304, 192, 0, 2), (24576, 256, 0, 2),
                       (6144, 960, 0, 2), (1536, 1536, 0, 2), (1536, 256, 1, 0)]:
    sets = []
    for _ in range(NSET):
        y = torch.randn(M, C, device="cuda").bfloat16()
        r = torch.randn(M, C, device="cuda").bfloat16() if res else None
        z = torch.empty_like(y)
        sets.append((y, r, z))
    w, b, rm, rv = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    nbt = torch.zeros(1, dtype=torch.int64, device="cuda"); scratch = torch.zeros(6 * C, device="cuda")
    def run(k):
        y, r, z = sets[k % NSET]
        lib.satrn_batchnorm_act_fwd(1, P(y), P(w), P(b), P(rm), P(rv), P(nbt), ctypes.c_float(1e-3), 0, act, P(r), P(z), M, C, P(scratch), st())
    for k in range(NSET): run(k)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 5 * NSET
    a.record()
    for k in range(iters): run(k)
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / iters * 1e3
    nbytes = M * C * 2 * (3 if res else 2)
    print(f"bn_act M={M:7d} C={C:5d} res={res} act={act}: {us:7.1f} us  {nbytes/us/1e6:6.2f} TB/s", flush=True)
