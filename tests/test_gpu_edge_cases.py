"""Empty / degenerate inputs through the C ABI: pipeline-level entries return empty results, kernel-level entries and
impossible requests fail loudly with the library's error text (no silent fallback)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_empty_inputs_give_empty_outputs():
    import bench
    from freepose_amd import ops
    from freepose_amd.retrieval import TemplateBank
    bf = torch.bfloat16
    vit = ops.ViT("dinov2_vits14_reg", seed=1)
    assert vit(torch.zeros((0, 3, 224, 224), dtype=bf).cuda(), layer=22, feature_type="patch").shape == (0, 256, 384)
    tb = TemplateBank(bench.synthetic_bank(300, 384, seed=3), shard=False)
    s, i = tb.topk(torch.zeros((0, 384), dtype=bf).cuda(), 10)
    assert s.shape == (0, 10) and i.shape == (0, 10)
    s, i = tb.topk(ops.l2_normalize(torch.ones((1, 384), dtype=bf).cuda()), 1000)      # k is clamped to the bank size
    assert s.shape == (1, 300) and len(set(i[0].tolist())) == 300
    assert ops.ffa(torch.zeros((0, 256, 384), dtype=bf).cuda(), torch.zeros((0, 224, 224), dtype=torch.uint8).cuda(),
                   cell=14, normalize=True).shape == (0, 384)
    assert ops.l2_normalize(torch.zeros((0, 384), dtype=bf).cuda()).shape == (0, 384)
    v, f, c = bench.synthetic_mesh(2)
    rgb, depth = ops.rasterize(ops.Mesh(v, f, c), torch.zeros((0, 3, 3)).cuda(), 0.25, 600, 600, 210, 210, 420, 420)
    assert rgb.shape == (0, 420, 420, 3) and depth.shape == (0, 420, 420)
    assert ops.depth_extents(torch.zeros((0, 420, 420)).cuda(), 600, 600, 210, 210).shape == (0, 8)
    assert ops.crop_resize_pad(torch.zeros((1, 3, 64, 64)).cuda(), torch.zeros((0, 4), dtype=torch.int32).cuda(), 32, 0.0).shape == (0, 3, 32, 32)
    assert ops.template_score(torch.zeros((0, 256, 384), dtype=bf).cuda(), torch.zeros((256, 384), dtype=bf).cuda()).shape == (0,)


def test_impossible_requests_fail_loudly():
    from freepose_amd import ops
    bf = torch.bfloat16
    with pytest.raises(RuntimeError, match="topk_merge"):
        ops.topk_merge(torch.zeros((1, 5)).cuda(), torch.zeros((1, 5), dtype=torch.int32).cuda(), 10)
    with pytest.raises(RuntimeError, match="gemm"):
        ops.gemm(torch.zeros((0, 64), dtype=bf).cuda(), torch.zeros((64, 64), dtype=bf).cuda(), torch.zeros(64, dtype=bf).cuda(), 0)
    with pytest.raises(RuntimeError, match="gemm"):      # K not a multiple of 64
        ops.gemm(torch.zeros((8, 40), dtype=bf).cuda(), torch.zeros((64, 40), dtype=bf).cuda(), torch.zeros(64, dtype=bf).cuda(), 0)
    with pytest.raises(RuntimeError, match="attention"):
        ops.attention(torch.zeros((0, 128), dtype=bf).cuda(), torch.zeros((0, 1, 64, 16), dtype=bf).cuda(), 16)


def test_fully_masked_and_degenerate_boxes():
    """a proposal whose mask is empty gives a NaN descriptor (0/0 like the reference's masked mean); a zero-area box gives an
    all-zero crop instead of reading out of bounds"""
    from freepose_amd import ops
    bf = torch.bfloat16
    feats = torch.randn((1, 256, 384)).to(bf).cuda()
    d = ops.ffa(feats, torch.zeros((1, 224, 224), dtype=torch.uint8).cuda(), cell=14, normalize=False)
    assert torch.isnan(d.float()).all()
    img = torch.rand((1, 3, 64, 64)).cuda()
    boxes = torch.tensor([[10, 10, 10, 10], [-50, -50, 5, 5], [60, 60, 200, 200]], dtype=torch.int32).cuda()
    out = ops.crop_resize_pad(img, boxes, 32, 0.0)
    assert out.shape == (3, 3, 32, 32) and torch.isfinite(out).all()


def test_detection_the_reference_cannot_crop_raises_like_the_reference():
    """a detection whose crop resizes to a side of 0 px ends the reference's script inside F.interpolate (bbox_utils.py:35); the mirror
    classes raise too (host check on the host-resident boxes) instead of handing the ViT the kernel's all-zero crop"""
    import numpy as np
    from freepose_amd.src.pipeline.utils import Proposals
    from freepose_amd.src.utils.bbox_utils import CropResizePad
    img = np.zeros((146, 125, 3), dtype=np.uint8)
    masks = np.zeros((2, 146, 125), dtype=bool)
    masks[0, 20:60, 20:60] = True
    masks[1, 87:132, 43] = True
    dets = {"masks": torch.from_numpy(masks), "boxes": torch.tensor([[20, 20, 60, 60], [43, 87, 44, 132]])}
    with pytest.raises(RuntimeError, match="detection 1"):
        Proposals(img, dets, 30, 0, 0, bbox_extend=0.5)
    ok = Proposals(img, {"masks": dets["masks"][:1], "boxes": dets["boxes"][:1]}, 30, 0, 0, bbox_extend=0.5)
    assert ok.proposals.shape == (1, 3, 30, 30)
    with pytest.raises(RuntimeError, match="box 0"):
        CropResizePad(30, (146, 125), bbox_extend=0.5)(torch.zeros((1, 3, 146, 125)), torch.tensor([[43, 87, 44, 132]]))
