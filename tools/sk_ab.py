"""Balanced GEMM tier (stream-K) against the plain small tiers, per ViT layer shape and launch size (lab build).

    rocprofv3 --kernel-trace ... -- python tools/sk_ab.py run labels.json     # every (shape, M, form) N_REP times, separated by marker kernels
    python tools/sk_ab.py read <kernel_trace.csv> labels.json                  # GEMM kernel time per call and form

Forms: plain = the product's dispatch without the balanced tier (gemm_sk = 1); sk128/G, sk64/G = the balanced tier forced on 128x128 /
64x64 units with G workgroups; ring128/G = 128x128 units on a K-tile ring (one workgroup per CU).  Weights rotate over enough copies
that no launch finds its operand in L2 / MALL (the ViT streams 550 MB of weights per forward).  The run also checks every form's result
against the plain tiers: identical wherever no tile is shared, <= 1 bf16 ulp of the fp32 accumulation order elsewhere."""
import csv
import json
import re
import sys
from pathlib import Path

N_REP = 6
SHAPES = {  # name: (N, K, epi)   epi 0 bias (qk), 1 bias+GELU (fc1), 2 LayerScale+residual (proj / fc2), 4 transposed V
    "qk": (2048, 1024, 0), "v": (1024, 1024, 4), "proj": (1024, 1024, 2), "fc1": (4096, 1024, 1), "fc2": (1024, 4096, 2)}
if __import__("os").environ.get("SK_SHAPES") == "stats":   # the forward's proj / fc2 launches: LayerScale + residual + row statistics (epilogue 8), in place
    SHAPES = {"proj": (1024, 1024, 2), "proj+stats": (1024, 1024, 8), "fc2": (1024, 4096, 2), "fc2+stats": (1024, 4096, 8)}
MS = {"B1@518": (1376, 1376), "B2@518": (2752, 1376), "B5@420": (4560, 912), "B21@420 remainder": (2768, 0), "B21@420": (19152, 912), "B8@420": (7296, 912)}


def forms(M, N, K):
    """(name, gemm_sk, gemm_sk_grid, gemm_ring cap, hot operands)"""
    import os
    if os.environ.get("SK_FORMS") == "probe":      # what bounds a small-tier K step: operands hot / cold, ring depth, one tile per workgroup
        t128 = -(-M // 128) * -(-N // 128)
        return [("plain", 1, 0, -1, False), ("plain hot", 1, 0, -1, True), (f"sk128/{t128} (1 tile each)", 2, t128, -1, False),
                ("sk128/256", 2, 256, -1, False), ("sk128/256 hot", 2, 256, -1, True), ("sk128/512", 2, 512, -1, False), ("sk128/512 hot", 2, 512, -1, True),
                ("ring128/256 r4", 4, 256, 4, False), ("ring128/256 r3", 4, 256, 3, False), ("ring128/256 r4 hot", 4, 256, 4, True)]
    if os.environ.get("SK_FORMS") == "waves":      # the 128x128 tier on 8 waves (two per SIMD from one workgroup), with / without the 64x64 tier
        return [("plain", 1, 0, -1, False), ("8 waves", 1, 0, -1, False, 238 | 262144), ("8 waves, no 64x64 tier", 1, 0, -1, False, 238 | 262144 | 2048),
                ("4 waves, no 64x64 tier", 1, 0, -1, False, 238 | 2048), ("8 waves, no split", 1, 0, -1, False, 238 | 262144 | 2048 | 4096)]
    if os.environ.get("SK_FORMS") == "spread":     # DMA pieces issued between the MFMAs of the second half-step
        return [("plain", 1, 0, -1, False), ("spread", 1, 0, -1, False, 238 | 524288), ("spread, 8 waves", 1, 0, -1, False, 238 | 524288 | 262144),
                ("spread, no split", 1, 0, -1, False, 238 | 524288 | 4096), ("spread, 8 waves, no split", 1, 0, -1, False, 238 | 524288 | 262144 | 4096)]
    if os.environ.get("SK_FORMS") == "asm":        # the hand-scheduled one-wave-per-SIMD 256x256 kernel on part-filled grids
        return [("plain", 1, 0, -1, False), ("no split", 1, 0, -1, False, 238 | 4096), ("asm, no split", 1, 0, -1, False, 238 | 4096 | 16384),
                ("asm", 1, 0, -1, False, 238 | 16384)]
    if os.environ.get("SK_FORMS") == "asms":       # the hand-scheduled 128x128 kernel below the big tier
        return [("plain", 1, 0, -1, False), ("asm 128x128", 1, 0, -1, False, 238 | 1048576), ("asm 128x128, no split", 1, 0, -1, False, 238 | 1048576 | 4096)]
    f = [("plain", 1, 0, -1, False)]
    for g in (256, 512):
        f.append((f"sk128/{g}", 2, g, -1, False))
    for g in (512, 1024):
        f.append((f"sk64/{g}", 3, g, -1, False))
    f.append(("ring128/256", 4, 256, -1, False))
    return f


def run(label_path):
    import torch
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from freepose_amd import _lib
    _lib.use_lab()
    from freepose_amd import ops
    g = torch.Generator().manual_seed(3)
    labels = []
    sel = sys.argv[3].split(",") if len(sys.argv) > 3 else list(MS)
    marker = torch.zeros(256, device="cuda", dtype=torch.bfloat16)
    import os
    flush = torch.zeros(256 << 20, device="cuda", dtype=torch.float32) if os.environ.get("SK_COLDX") == "1" else None   # operands cold on every call
    end_x, end_g = torch.zeros((1, 64), device="cuda", dtype=torch.bfloat16), torch.zeros(64, device="cuda", dtype=torch.bfloat16)
    for mname in sel:
        M, npad = MS[mname]
        for sname, (N, K, epi) in SHAPES.items():
            if epi == 4 and npad == 0:
                continue
            ncopy = max(2, (300 << 20) // (N * K * 2))
            ws = [(torch.randn((N, K), generator=g) * 0.03).to(torch.bfloat16).cuda() for _ in range(min(ncopy, 8))]
            while len(ws) < ncopy:
                ws.append(ws[len(ws) % 8].clone())
            x = torch.randn((M, K), generator=g).to(torch.bfloat16).cuda()
            bias = torch.randn((N,), generator=g).to(torch.bfloat16).cuda()
            gamma = torch.randn((N,), generator=g).to(torch.bfloat16).cuda()
            resid = torch.randn((M, N), generator=g).to(torch.bfloat16).cuda()
            ref = None
            for fname, mode, grid, ring, hot, *rest in forms(M, N, K):
                ops.set_option("gemm_variant", rest[0] if rest else -1)
                ops.set_option("gemm_sk", mode)
                ops.set_option("gemm_sk_grid", grid)
                ops.set_option("gemm_ring", ring)

                def call(w):
                    if epi == 4:
                        return ops.gemm_vt(x, w, bias, npad, 16)
                    if epi == 8:
                        return ops.gemm_stats(x, w, bias, gamma, resid)[0]
                    return ops.gemm(x, w, bias, epi, gamma=gamma, resid=resid)
                out = call(ws[0]).float()
                if ref is None:
                    ref = out
                else:
                    d = (out - ref).abs()
                    tol = (ref.abs() + (resid.float().abs() if epi in (2, 8) else 0.0)) * 2.0 ** -6 + 1e-3   # (epi 2 cancels against the residual)
                    bad = int((d > tol).sum())
                    neq = int((out != ref).sum())
                    print(f"  check {mname} {sname} {fname}: {neq} of {out.numel()} elements differ from plain, {bad} beyond 1 bf16 ulp", flush=True)
                    assert bad == 0, (mname, sname, fname)
                torch.cuda.synchronize()
                ops.gelu_direct(marker)
                for r in range(N_REP):
                    if flush is not None:
                        flush.add_(1)                   # 1 GiB read + written: X, the residual and the weights leave L2 and the Infinity Cache
                    call(ws[0] if hot else ws[(r * 7 + 1) % len(ws)])
                ops.layernorm(end_x, end_g, end_g)      # end marker: what follows (the next form's check call) is not timed
                torch.cuda.synchronize()
                labels.append([mname, sname, fname, M, N, K])
            ops.set_option("gemm_sk", -1)
            ops.set_option("gemm_sk_grid", -1)
            ops.set_option("gemm_ring", -1)
            ops.set_option("gemm_variant", -1)
            del ws
    torch.cuda.synchronize()
    json.dump(labels, open(label_path, "w"))
    print("labels:", len(labels))


def read(trace, label_path):
    labels = json.load(open(label_path))
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    groups, cur = [], None
    for r in rows:
        k = r["Kernel_Name"]
        if "gelu_direct_kernel" in k:
            cur = []
        elif "layernorm_kernel" in k:
            if cur is not None:
                groups.append(cur)
            cur = None
        elif cur is not None and re.search(r"gemm_(bf16|asm)_kernel", k):
            cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), k))
    assert len(groups) == len(labels), (len(groups), len(labels))
    table = {}
    for (mname, sname, fname, M, N, K), grp in zip(labels, groups):
        # the first call of a group is the correctness call's twin only for the marker order above: all N_REP calls follow the marker
        t = sum(d for d, _ in grp) / N_REP / 1e3
        kinds = sorted({re.sub(r".*gemm_(bf16|asm)_kernel<([^>]*)>.*", r"\2", k) for _, k in grp})
        table.setdefault((mname, sname, M, N, K), []).append((fname, t, len(grp) / N_REP, kinds))
    for (mname, sname, M, N, K), fs in table.items():
        base = fs[0][1]
        best = min(fs, key=lambda f: f[1])
        print(f"{mname:18s} {sname:5s} M={M:6d} N={N:5d} K={K:5d}: " + " | ".join(
            f"{fn} {t:6.1f} us ({2.0 * M * N * K / t / 1e6:5.0f} TF)" for fn, t, _, _ in fs) + f"   best {best[0]} x{base / best[1]:.2f}   [plain = {fs[0][3]}]")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        read(sys.argv[2], sys.argv[3])
