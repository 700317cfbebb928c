"""GPU-busy fraction of the bank-building loop on shards on disk (BASELINE config 2 through scripts.extract_retrieval_features.process).

    rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python tools/bank_build_prof.py run [n_meshes] [--no_prefetch]
    python tools/bank_build_prof.py read <kernel_trace.csv>

`run` builds the synthetic workspace of bench.py's CLI legs, processes mesh 0 as a warm-up, then `n_meshes` meshes; two marker kernels
(gelu_direct_kernel) bracket the steady part — from the moment the first timed mesh's file is written to the end — in which every mesh's host
stage (tar reads, PNG decode, host->device copy) has had a predecessor's ViT calls to hide under.  `read` reports, between the markers,
the time the GPU spent inside kernels / the wall span."""
import csv
import os
import shutil
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def run(n_meshes, no_prefetch):
    import torch
    import bench
    from freepose_amd import ops
    from freepose_amd.src.dataloader.template import WebTemplateDataset
    from freepose_amd.src.pipeline.retrieval.dino import DINOv2FeatureExtractor
    from scripts import extract_retrieval_features as erf
    vit = ops.ViT("dinov2_vitl14_reg", seed=0)
    fe = DINOv2FeatureExtractor.__new__(DINOv2FeatureExtractor)
    torch.nn.Module.__init__(fe)
    fe.model_name, fe.model, fe.num_register_tokens = "dinov2_vitl14_reg", vit, vit.n_reg
    root, names = bench.make_cli_workspace(n_meshes, 600)
    cwd = os.getcwd()
    marker = torch.zeros(256, device="cuda", dtype=torch.bfloat16)
    try:
        os.chdir(root)
        a = erf.build_parser().parse_args(["--batch_size", "256", "--n_views", "600"] + (["--no_prefetch"] if no_prefetch else []))
        fdir = root / "data" / "datasets" / "feat"
        fdir.mkdir(parents=True, exist_ok=True)
        ds = WebTemplateDataset((root / "data" / "datasets" / "objaverse_shards").as_posix(), (root / "data" / "mesh_cache.csv").as_posix(),
                                crop=False, n_views=600, cache_meshes=0)
        erf.process(fe, ds, [0], a, fdir, quiet=True)
        torch.cuda.synchronize()

        class Stamps(list):
            def append(self, t):
                if not self:
                    ops.gelu_direct(marker)            # first timed mesh done: the steady part starts
                super().append(t)
        st = Stamps()
        t0 = time.perf_counter()
        erf.process(fe, ds, list(range(n_meshes)), a, fdir, quiet=True, stamps=st)
        ops.gelu_direct(marker)
        torch.cuda.synchronize()
        sec = time.perf_counter() - t0
        iv = [b - a_ for a_, b in zip(st[:-1], st[1:])]
        print(f"bank build, {'no prefetch' if no_prefetch else 'prefetch'}: {n_meshes} meshes in {sec:.2f} s = {n_meshes / sec:.2f} meshes/s; "
              f"steady interval median {sorted(iv)[len(iv) // 2]:.3f} s per mesh; host stage {ds.decode_seconds / (n_meshes + 1):.3f} s per mesh", flush=True)
    finally:
        os.chdir(cwd)
        shutil.rmtree(root, ignore_errors=True)


def read(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "gelu_direct_kernel" in r["Kernel_Name"]]
    assert len(marks) >= 2, len(marks)
    a, b = marks[-2], marks[-1]
    sel = rows[a + 1:b]
    span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"])
    busy, last_end = 0, int(rows[a]["End_Timestamp"])
    for r in sel:                                      # union of kernel intervals (streams may overlap: the side-stream copies are not kernels)
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e > last_end:
            busy += e - max(s, last_end)
            last_end = e
    print(f"{path}: steady part {span / 1e6:.1f} ms, {len(sel)} kernels, GPU inside kernels {busy / 1e6:.1f} ms = {100.0 * busy / span:.1f} % busy")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 8, "--no_prefetch" in sys.argv)
    else:
        read(sys.argv[2])
