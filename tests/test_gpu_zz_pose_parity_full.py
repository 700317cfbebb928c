"""BASELINE config 4 at ITS OWN size in the default `-m gpu` run: 576 pose hypotheses, 518^2 crops, ViT-L/14-reg
layer 22, oracle ViT in the reference's bf16 regime (pose_estimator.py:21,85-118; online_pose_estimator.py:66-96) — every one of the
576 x Q HIP scores within 3 bf16 ulp of the oracle's, decisive queries pick the oracle's hypothesis (re = 0, te <= 1e-6 m).
The file name sorts last so the table closes the suite's output.  2 queries by default (578 oracle ViT-L forwards at ~0.4 s each on
16 threads: ~5 minutes); FP_PARITY_QUERIES=6 / FP_PARITY_FP32=1 for the long form (log of one: profiles/r05_pose_parity_full.log)."""
import os

import pytest

from tests.test_gpu_pose_parity import run_pose_parity

pytestmark = pytest.mark.gpu


def test_pose_parity_config4_full_size(capsys):
    regimes = ("bf16", "fp32") if os.environ.get("FP_PARITY_FP32") == "1" else ("bf16",)
    run_pose_parity(capsys, 576, int(os.environ.get("FP_PARITY_QUERIES", "2")), 518, regimes)
