"""python -m scripts.merge_features --features_folder objaverse_shards_ffa_22

Drop-in for scripts/merge_features.py:17-35: per-mesh [<=600, D] descriptor files -> one bank row per mesh (mean over
views, fp32) -> data/<folder>.npy.  Unlike the reference, which silently skips missing/NaN meshes and thereby shifts
every later row against data/mesh_cache.txt, the kept ids are written next to the bank
(data/<folder>.ids.txt) so row i is always attributable."""
from __future__ import annotations

import argparse
from pathlib import Path

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--features_folder", type=str, default="objaverse_features_ffa_22")
    ap.add_argument("--filelist", type=str, default="mesh_cache.txt")
    args = ap.parse_args(argv)
    folder = Path("data/datasets/").resolve() / args.features_folder
    ids = Path(f"data/{args.filelist}").read_text(encoding="utf-8").splitlines()
    rows, kept = [], []
    for mesh_id in ids:
        f = folder / f"{mesh_id}.npy"
        if not f.exists():
            print(f"Feature {f} does not exist")
            continue
        row = np.mean(np.load(f), axis=0)
        if np.isnan(row).any():
            print(f"Feature {f} contains NaNs")
            continue
        rows.append(row)
        kept.append(mesh_id)
    np.save(f"data/{args.features_folder}.npy", np.stack(rows, axis=0))
    Path(f"data/{args.features_folder}.ids.txt").write_text("\n".join(kept) + "\n")


if __name__ == "__main__":
    main()
