"""scripts.merge_results (the gather step of the static-image driver, reference scripts/merge_results.py:12-29): per-task / per-rank
CSVs of a results folder -> one BOP results file, rows in (task, rank) order, empty files and incomplete rows dropped, reference file name."""
import pandas as pd

from freepose_amd.scripts import merge_results as mr
from freepose_amd.scripts.dino_inference import CSV_COLUMNS


def _rows(ids):
    return pd.DataFrame([{"scene_id": 48, "im_id": i, "obj_id": f"m{i}", "score": 0.5, "R": "1 0 0 0 1 0 0 0 1", "t": "0 0 700.0",
                          "bbox_visib": "1 2 3 4", "scale": 0.1, "time": 0.2} for i in ids], columns=CSV_COLUMNS)


def test_merge_results_orders_by_task_and_rank_and_names_like_the_reference(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    folder = tmp_path / "data" / "results" / "ycbv" / "props-ground-box-0.3-text-0.5-ffa-22-top-0_ycbv-test_dinopose_layer_22_bbext_0.05_depth_zoedepth_cache_50"
    folder.mkdir(parents=True)
    _rows([31, 33]).to_csv(folder / "pose_outputs_1_r0.csv", index=False)
    _rows([32]).to_csv(folder / "pose_outputs_1_r1.csv", index=False)
    _rows([1, 2, 3]).to_csv(folder / "pose_outputs_0.csv", index=False)
    _rows([301]).to_csv(folder / "pose_outputs_10.csv", index=False)           # task 10 sorts after task 1 (not lexicographically)
    _rows([]).to_csv(folder / "pose_outputs_2.csv", index=False)               # a task without detections: header only
    broken = _rows([99])
    broken.loc[0, "t"] = None
    broken.to_csv(folder / "pose_outputs_3.csv", index=False)                  # an incomplete row is dropped (dropna, :26)
    (tmp_path / "data" / "results" / "ycbv" / "props.json").write_text("[]")   # files beside the folders are skipped (:15-16)
    out = mr.main(["--dataset", "ycbv"])
    assert [p.name for p in out] == ["props-ground-box-0.3-text-0.5-ffa-22-top-0-dinopose-layer-22-bbext-0.05-depth-zoedepth-cache-50_ycbv-test.csv"]
    df = pd.read_csv(out[0])
    assert list(df.columns) == CSV_COLUMNS
    assert df["im_id"].tolist() == [1, 2, 3, 31, 33, 32, 301]
    assert mr.merged_name("a_b_ycbv-test_c", "ycbv", "test") == "a-b-c_ycbv-test.csv"
