"""Is running the V^T GEMM beside the q/k GEMM on a second stream worth its two cross-stream waits per layer?  22 x (qk GEMM, V^T GEMM,
attention, proj GEMM) at small batch sizes: one stream vs qk on the main stream + V^T on a side stream (fork / join through events).
    python tools/two_stream_probe.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from freepose_amd import ops  # noqa: E402

L, D, H = 22, 1024, 16


def run(B, n_tok, two_streams, iters=20):
    npad = (n_tok + 15) // 16 * 16
    M = B * npad
    g = torch.Generator().manual_seed(0)
    x = torch.randn((M, D), generator=g).to(torch.bfloat16).cuda()
    ws = [(torch.randn((3 * D, D), generator=g) * 0.03).to(torch.bfloat16).cuda() for _ in range(L)]
    wp = [(torch.randn((D, D), generator=g) * 0.03).to(torch.bfloat16).cuda() for _ in range(L)]
    bias3, bias1 = torch.zeros(3 * D, dtype=torch.bfloat16).cuda(), torch.zeros(D, dtype=torch.bfloat16).cuda()
    qk = torch.empty((M, 2 * D), dtype=torch.bfloat16, device="cuda")
    vt = torch.zeros((B, H, 64, npad), dtype=torch.bfloat16, device="cuda")
    ao = torch.empty((M, D), dtype=torch.bfloat16, device="cuda")
    y = torch.empty((M, D), dtype=torch.bfloat16, device="cuda")
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    evs = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(L)]

    def forward():
        cur = x
        for i in range(L):
            if two_streams:
                evs[i][0].record(main)                         # `cur` is ready
                with torch.cuda.stream(side):
                    side.wait_event(evs[i][0])
                    ops.gemm_vt(cur, ws[i][2 * D:], bias1, npad, H, out=vt)
                    evs[i][1].record(side)
                ops.gemm(cur, ws[i][:2 * D], bias3[:2 * D], 0, out=qk)
                main.wait_event(evs[i][1])
            else:
                ops.gemm(cur, ws[i][:2 * D], bias3[:2 * D], 0, out=qk)
                ops.gemm_vt(cur, ws[i][2 * D:], bias1, npad, H, out=vt)
            ops.attention(qk, vt, n_tok, out=ao)
            ops.gemm(ao, wp[i], bias1, 0, out=y)
            cur = y
    for _ in range(3):
        forward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        forward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, vt.float().abs().sum().item()


for B, n_tok in ((1, 905), (1, 1374), (2, 1374), (5, 905), (21, 905)):
    run(B, n_tok, False, iters=2)                          # first touch of the shapes (workspaces, kernel attributes) outside the comparison
    b, cb = run(B, n_tok, True)
    a, ca = run(B, n_tok, False)
    print(f"B={B} n_tok={n_tok}: one stream {a:.3f} ms | V^T on a side stream {b:.3f} ms   x{a / b:.3f}   same result: {ca == cb}")
