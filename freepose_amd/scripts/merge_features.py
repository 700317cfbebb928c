"""python -m scripts.merge_features --features_folder objaverse_shards_ffa_22

Drop-in for scripts/merge_features.py:17-35: per-mesh [<=600, D] descriptor files -> one bank row per mesh (mean over
views, fp32) -> data/<folder>.npy.  Unlike the reference, which silently skips missing/NaN meshes and thereby shifts
every later row against data/mesh_cache.txt, the kept ids are written next to the bank
(data/<folder>.ids.txt) so row i is always attributable."""
from __future__ import annotations

import argparse
from pathlib import Path
from typing import Iterable, List, Optional, Tuple

import numpy as np

DATA = Path("data")


def bank_row(path: Path) -> Optional[np.ndarray]:
    """mean descriptor of one mesh, or None (with the reference's console message) when the file is missing or has NaNs"""
    if not path.exists():
        print(f"Feature {path} does not exist")
        return None
    row = np.load(path).mean(axis=0)
    if np.isnan(row).any():
        print(f"Feature {path} contains NaNs")
        return None
    return row


def collect(folder: Path, mesh_ids: Iterable[str]) -> Tuple[np.ndarray, List[str]]:
    kept, rows = [], []
    for mesh_id in mesh_ids:
        row = bank_row(folder / f"{mesh_id}.npy")
        if row is not None:
            kept.append(mesh_id)
            rows.append(row)
    return np.stack(rows, axis=0), kept


def main(argv=None):
    cli = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    cli.add_argument("--features_folder", type=str, default="objaverse_features_ffa_22")
    cli.add_argument("--filelist", type=str, default="mesh_cache.txt")
    opt = cli.parse_args(argv)
    mesh_ids = (DATA / opt.filelist).read_text(encoding="utf-8").splitlines()
    bank, kept = collect((DATA / "datasets").resolve() / opt.features_folder, mesh_ids)
    np.save(DATA / f"{opt.features_folder}.npy", bank)
    (DATA / f"{opt.features_folder}.ids.txt").write_text("\n".join(kept) + "\n")


if __name__ == "__main__":
    main()
