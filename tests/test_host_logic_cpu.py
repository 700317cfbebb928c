"""Host-side policy code that needs no GPU."""
import pytest

from freepose_amd import ops


def _rounds(b, n_tok, n_cu=256):
    npad = (n_tok + 15) // 16 * 16
    return -(-(-(-b * npad // 256) * 4) // n_cu)


@pytest.mark.parametrize("n,n_tok,max_batch", [(576, 1374, 192), (600, 905, 192), (577, 1374, 64), (19, 1374, 192),
                                               (1, 1374, 192), (256, 1374, 256), (1000, 261, 128)])
def test_plan_vit_batches(n, n_tok, max_batch):
    plan = ops.plan_vit_batches(n, n_tok, max_batch)
    assert sum(plan) == n and all(b > 0 for b in plan)
    assert max(plan) <= max_batch + max(1, max_batch // 8)
    naive = [max_batch] * (n // max_batch) + ([n % max_batch] if n % max_batch else [])
    assert sum(_rounds(b, n_tok) for b in plan) <= sum(_rounds(b, n_tok) for b in naive)


def test_plan_vit_batches_empty():
    assert ops.plan_vit_batches(0, 1374) == []


def test_unresizable_box_names_what_the_reference_cannot_crop():
    """CropResizePad's host check (bbox_utils.unresizable_box): the reference's torch code raises on a box whose resized side is 0 px
    (found by oracle/fuzz_vs_reference.py: box [43, 87, 44, 132] in a 146 x 125 image, target 30, extension 0.5 -> 2 x 82 px -> 0 x 29),
    on an empty crop, and cannot use an exactly square crop that comes out one pixel short of the target; everything else passes"""
    import numpy as np
    from freepose_amd.src.utils.bbox_utils import unresizable_box
    assert unresizable_box(np.array([[43, 87, 44, 132]]), 146, 125, 30, 0.5) == 0
    assert unresizable_box(np.array([[10, 10, 60, 60], [43, 87, 44, 132]]), 146, 125, 30, 0.5) == 1
    assert unresizable_box(np.array([[10, 10, 10, 40]]), 146, 125, 30, 0.0) == 0            # empty crop
    assert unresizable_box(np.array([[200, 10, 260, 40]]), 146, 125, 30, 0.0) == 0          # entirely outside the image
    assert unresizable_box(np.array([[0, 0, 125, 146], [10, 10, 60, 60], [5, 7, 75, 47]]), 146, 125, 30, 0.1) == -1
    assert unresizable_box(np.zeros((0, 4), dtype=np.int64), 146, 125, 30, 0.1) == -1
    # every side length at the pipeline's sizes: which exactly-square crops come out one pixel short is pinned by the oracle (and the
    # oracle by the reference: profiles/r05_oracle_fuzz_vs_reference.log)
    from oracle import fp_oracle as fo
    for target in (420, 224, 56):
        for side in range(1, 700):
            b = np.array([[0, 0, side, side]], dtype=np.int32)
            try:
                fo.crop_resize_pad(np.zeros((1, 1, 700, 700), np.float32), b, target, 0.0)
                ok = True
            except ValueError:
                ok = False
            assert (unresizable_box(b, 700, 700, target, 0.0) == -1) == ok, (target, side)
