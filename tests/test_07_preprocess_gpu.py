"""Occlusion / out-of-bounds masks on the GPU (dvd_flow_consistency_mask) against the oracle's restatement of
scripts/preprocess/davis/generate_flows.py:57-82,139-148.  Integer masks: bit-exact (zero mismatching pixels),
on flows whose forward/backward error straddles the 1-pixel threshold and whose targets straddle the border."""
import numpy as np
import pytest
import torch

from oracle import preprocess as OP

pytestmark = pytest.mark.gpu


def _pair(H, W, seed, noise):
    """A smooth forward flow, its approximate inverse, plus noise of about the threshold's size."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    f12 = torch.stack([6.0 * torch.sin(yy / 17.0) + 0.02 * xx, 4.0 * torch.cos(xx / 23.0) - 0.03 * yy], -1)
    f21 = -f12 + noise * torch.randn(H, W, 2, generator=g)
    f12 = f12 + 0.3 * noise * torch.randn(H, W, 2, generator=g)
    f12[:5] += 40.0                        # a band whose targets leave the image
    return f12.contiguous(), f21.contiguous()


@pytest.mark.parametrize('H,W,seed,noise', [(48, 64, 1, 0.6), (96, 168, 2, 0.9), (192, 384, 3, 0.5), (33, 51, 4, 1.5)])
def test_masks_are_bit_identical_to_the_reference_formulas(H, W, seed, noise):
    from dvd_hip import preprocess as P
    f12, f21 = _pair(H, W, seed, noise)
    want_1, want_2 = OP.consistency_masks(f12.numpy(), f21.numpy())
    got_1, got_2 = P.flow_consistency_masks(f12.cuda(), f21.cuda())
    for name, got, want in (('mask_1', got_1, want_1), ('mask_2', got_2, want_2)):
        got = got.cpu().numpy()
        assert got.dtype == np.uint8 and got.shape == want.shape
        assert 0.05 < want.mean() < 0.95, 'the case must exercise both outcomes'
        assert (got != want).sum() == 0, '%s: %d pixels differ' % (name, (got != want).sum())


def test_batched_call_and_training_mask_convention():
    from dvd_hip import preprocess as P
    pairs = [_pair(40, 56, s, 0.7) for s in (5, 6, 7)]
    f12 = torch.stack([p[0] for p in pairs]).cuda()
    f21 = torch.stack([p[1] for p in pairs]).cuda()
    m1, m2 = P.flow_consistency_masks(f12, f21)
    for b, (a, c) in enumerate(pairs):
        w1, w2 = OP.consistency_masks(a.numpy(), c.numpy())
        assert (m1[b].cpu().numpy() != w1).sum() == 0 and (m2[b].cpu().numpy() != w2).sum() == 0
    t1, t2 = P.training_masks(m1, m2)            # generate_sequence_midas.py:144-147: 1 = valid, [B,H,W,1,1]
    assert t1.shape == (3, 40, 56, 1, 1) and t1.dtype == torch.float32
    assert torch.equal(t2[..., 0, 0], 1.0 - m2.float())
