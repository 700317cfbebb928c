import sys, torch
sys.path.insert(0, '.')
from freepose_amd import ops
import statistics
B, n_tok = 64, 1374
npad = (n_tok + 15) // 16 * 16
qk = torch.randn(B * npad, 2048, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, 16, 64, npad, device="cuda").to(torch.bfloat16)
o = torch.empty(B * npad, 1024, device="cuda", dtype=torch.bfloat16)
def t():
    ops.attention(qk, vt, n_tok, out=o); torch.cuda.synchronize()
    tm = ops.Timer(); tm.start()
    for _ in range(10): ops.attention(qk, vt, n_tok, out=o)
    tm.stop(); return tm.elapsed_ms() / 10
r = [t() for _ in range(5)]
fl = 4.0 * B * n_tok * n_tok * 1024
print(f"attention B={B} n={n_tok}: median {statistics.median(r):.3f} ms = {fl / statistics.median(r) / 1e9:.0f} TF (best {fl / min(r) / 1e9:.0f})")
