"""Property tests of the host logic (hypothesis; no GPU): the partitions the N-rank drivers rely on, the ViT batch planner, the
proposals-JSON mask codec, and the host-side box check against the oracle's refusal rule."""
import numpy as np
from hypothesis import given, settings, strategies as st

from freepose_amd import ops, parallel
from freepose_amd.src.pipeline.utils import mask_to_rle_pytorch, rle_to_mask
from freepose_amd.src.utils.bbox_utils import unresizable_box
from oracle import fp_oracle as fo

FAST = settings(max_examples=300, deadline=None)


@FAST
@given(st.integers(0, 5000), st.integers(1, 64))
def test_rank_partitions_cover_every_item_exactly_once(n, world):
    """shard_items (round-robin: proposals, frames, objects) and shard_chunk (contiguous clip stretches): every id on exactly one rank,
    in ascending order, sizes within one of each other; chunks are contiguous and ordered by rank"""
    rr = [parallel.shard_items(n, r, world) for r in range(world)]
    ch = [parallel.shard_chunk(n, r, world) for r in range(world)]
    for parts in (rr, ch):
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(n))
        assert all(p == sorted(p) for p in parts)
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 1
    cat = [i for p in ch for i in p]
    assert cat == list(range(n))                                     # rank order == clip order
    for r in range(world):
        lo, hi = parallel.shard_range(n, r, world)
        assert ch[r] == list(range(lo, hi))


@FAST
@given(st.integers(0, 3000), st.sampled_from([261, 905, 1374, 1449, 17]), st.integers(1, 400))
def test_vit_batch_plan_is_a_partition_within_bounds(n, n_tok, max_batch):
    plan = ops.plan_vit_batches(n, n_tok, max_batch)
    assert sum(plan) == n and all(b > 0 for b in plan)
    assert plan == sorted(plan, reverse=True)
    if n:
        assert max(plan) <= max_batch + max(1, max_batch // 8)
        if len(plan) > 1:                                            # several batches: none is a sliver below half the nominal size
            assert min(plan) >= max(1, max_batch // 2) or len(plan) == 2


@FAST
@given(st.integers(1, 40), st.integers(1, 40), st.integers(0, 2 ** 32 - 1), st.sampled_from([0.0, 0.05, 0.5, 0.95, 1.0]))
def test_rle_codec_round_trips(h, w, seed, density):
    """uncompressed COCO RLE, column-major, first run = background (sam2/utils/amg.py:109-151): decode(encode(m)) == m, the runs add up
    to the mask size, alternate, and only the first may be empty"""
    m = np.random.Generator(np.random.PCG64(seed)).random((2, h, w)) < density
    for rle, mask in zip(mask_to_rle_pytorch(m), m):
        assert rle["size"] == [h, w] and sum(rle["counts"]) == h * w
        assert all(c > 0 for c in rle["counts"][1:])
        assert (rle["counts"][0] == 0) == bool(mask[0, 0])
        assert np.array_equal(rle_to_mask(rle), mask)
        assert len(rle["counts"]) == 1 + int(np.count_nonzero(np.diff(mask.T.reshape(-1).astype(np.int8)))) + int(mask[0, 0])


@settings(max_examples=1500, deadline=None)
@given(st.integers(8, 300), st.integers(8, 300), st.integers(-20, 320), st.integers(-20, 320), st.integers(-5, 330), st.integers(-5, 330),
       st.sampled_from([14, 30, 98, 224, 420]), st.sampled_from([0.0, 0.05, 0.1, 0.2, 0.5]))
def test_host_box_check_agrees_with_the_oracle(H, W, x0, y0, bw, bh, target, ext):
    """bbox_utils.unresizable_box (what makes CropResizePad / Proposals raise like the reference's torch code) refuses exactly the boxes the
    oracle refuses — boxes inside, across and outside the image, empty and inverted ones, every extension"""
    box = np.array([[x0, y0, x0 + bw, y0 + bh]], dtype=np.int32)
    try:
        fo.crop_resize_pad(np.zeros((1, 1, H, W), np.float32), box, target, ext)
        ok = True
    except ValueError:
        ok = False
    assert (unresizable_box(box, H, W, target, ext) == -1) == ok, (H, W, box.tolist(), target, ext)
