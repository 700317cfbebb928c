"""python -m scripts.merge_results --dataset ycbv

Drop-in for scripts/merge_results.py:12-29 — the gather step of the static-image driver: every sub-folder of `data/results/<dataset>/`
holds the CSVs that `scripts.dino_inference` wrote per SLURM array task (`pose_outputs_<task>.csv`) and, under `--gpus N` /
torch.distributed.run, per rank (`pose_outputs_<task>_r<rank>.csv`); each folder becomes ONE BOP results file in the working directory,
named like the reference's (`_`<dataset>-<split>` moved to the end, the remaining `_` turned into `-`).

The reference concatenates the files in directory-listing order (unspecified); here they are taken in (task, rank) order, so the merged
file does not depend on the file system — with one rank per task that is the images in dataset order.  Empty files and rows with a
missing value are dropped as there (:21-26)."""
from __future__ import annotations

import argparse
import re
from pathlib import Path

import pandas as pd


def _order(path: Path):
    """pose_outputs_<task>[_r<rank>].csv -> (task, rank); anything else after them, by name"""
    m = re.fullmatch(r"pose_outputs_(\d+)(?:_r(\d+))?\.csv", path.name)
    return (0, int(m.group(1)), int(m.group(2) or 0), "") if m else (1, 0, 0, path.name)


def merged_name(folder_name: str, dataset: str, split: str) -> str:
    return folder_name.replace(f"_{dataset}-{split}", "").replace("_", "-") + f"_{dataset}-{split}.csv"


def merge_folder(folder: Path) -> pd.DataFrame:
    parts = []
    for f in sorted((p for p in folder.iterdir() if p.is_file()), key=_order):
        df = pd.read_csv(f)
        if not df.empty:
            parts.append(df)
    if not parts:
        raise ValueError(f"no rows to merge in {folder}")             # (the reference fails here too: pd.concat of nothing)
    return pd.concat(parts).dropna()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", type=str)
    ap.add_argument("--split", type=str, default="test")
    args = ap.parse_args(argv)
    results = Path("./data/results/").resolve() / args.dataset
    written = []
    for folder in sorted(results.iterdir()):
        if folder.is_file():
            continue
        out = Path(merged_name(folder.name, args.dataset, args.split))
        merge_folder(folder).to_csv(out, header=True, index=False)
        written.append(out)
    return written


if __name__ == "__main__":
    main()
