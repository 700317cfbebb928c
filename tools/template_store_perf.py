"""Template store measurement on a shard in the reference's on-disk format (tests/synth_shard.py: 2 meshes x 600
views, 420^2 RGB + u16 depth PNGs): time per mesh for a cold load (tar read + threaded PNG decode + device crops) and for a
store hit, against a single-threaded decode (the reference's loop, template.py:65-72).  Development probe."""
import sys
import tempfile
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import synth_shard  # noqa: E402
from freepose_amd.src.dataloader.template import WebTemplateDataset  # noqa: E402


def main():
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        t0 = time.perf_counter()
        names = synth_shard.write_shard(td)
        print(f"wrote the synthetic shard in {time.perf_counter() - t0:.1f} s ({(td / 'shards' / 'shard-000000.tar').stat().st_size / 1e6:.0f} MB)")
        for threads in (1, None):
            ds = WebTemplateDataset(str(td / "shards"), str(td / "mesh_cache.csv"), bbox_extend=0.05, decode_threads=threads)
            ds.get_template_by_name(names[0])           # first touch also writes the member index
            torch.cuda.synchronize()
            ds._store.clear()
            t0 = time.perf_counter()
            ds.get_template_by_name(names[1])
            torch.cuda.synchronize()
            cold = time.perf_counter() - t0
            t0 = time.perf_counter()
            for _ in range(20):
                ds.get_template_by_name(names[1])
            torch.cuda.synchronize()
            hit = (time.perf_counter() - t0) / 20
            print(f"decode threads {ds._threads:3d}: cold load {cold:.2f} s/mesh ({1 / cold:.2f} meshes/s; host decode {ds.decode_seconds / 2:.2f} s), "
                  f"store hit {hit * 1e6:.0f} us/mesh", flush=True)


if __name__ == "__main__":
    main()
