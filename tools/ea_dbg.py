"""Phase timing of the fused encoder self-attention region kernel (SATRN_TIMING=ea_dbg=N: leave after phase N; wrong results)."""
import ctypes, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satrn_amd
lib = satrn_amd._lib.load()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, L, D, H = 32, 48, 512, 8
M = B * L
bf = torch.bfloat16
x = torch.randn(M, D, device="cuda").to(bf)
lnw, lnb = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
wqkv = (torch.randn(3 * D, D, device="cuda") * 0.03).to(bf); bqkv = torch.zeros(3 * D, device="cuda")
wo = (torch.randn(D, D, device="cuda") * 0.03).to(bf); bo = torch.zeros(D, device="cuda")
e = lambda *s: torch.empty(*s, dtype=bf, device="cuda")
y1, qkv, att, o, y2, parts = e(M, D), e(M, 3 * D), e(M, D), e(M, D), e(M, D), e(H // 2, M, D)
mr1, mr2, lse = torch.empty(2 * M, device="cuda"), torch.empty(2 * M, device="cuda"), torch.empty(B * H * L, device="cuda")
def run():
    lib.satrn_enc_attn_region_fwd(P(x), P(lnw), P(lnb), P(wqkv), P(bqkv), P(wo), P(bo), B, L, D, H, 0.0, 0.0, None, 0, 0, P(y1), P(mr1), P(qkv), P(att), P(lse), P(parts), P(o), P(y2), P(mr2), st())
for dbg in (1, 2, 3, 0):
    os.environ["SATRN_TIMING"] = "ea_dbg=" + str(str(dbg))
    for _ in range(5): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): run()
    b.record(); torch.cuda.synchronize()
    print(f"leave after phase {dbg if dbg else 'end'}: {a.elapsed_time(b) * 10:.1f} us per (region kernel + LN fold)")
