"""The lab build (-DFP_LAB: alternative GEMM loops, attention ring depths, measurement hooks, FP_* toggles) is exercised in its
own process — tools/lab_selfcheck.py — so that the product library and the lab library are never loaded together."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_lab_build_variants_agree_with_the_product_kernels():
    from freepose_amd import _lib
    if not _lib.LAB_LIB_PATH.exists():
        pytest.skip("lab library not built (python -m freepose_amd.build --lab)")
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "lab_selfcheck.py")], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "LAB_SELFCHECK_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_product_rejects_lab_options():
    from freepose_amd import ops
    with pytest.raises(RuntimeError, match="lab build"):
        ops.set_option("gemm_variant", 6)
    with pytest.raises(RuntimeError, match="unknown option"):
        ops.check(ops._lib.load().fp_ctx_set_option(ops.context(), b"gemm_dbg", 8), "fp_ctx_set_option")
