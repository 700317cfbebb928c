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
