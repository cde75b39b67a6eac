"""Occlusion / out-of-bounds masks on the GPU (dvd_flow_consistency_mask) against (a) tests/golden/flow_masks.npz --
masks computed by the REFERENCE'S OWN code (`get_oob_mask`, `backward_flow_warp` and the mask statements of
`generate_pair_data`, scripts/preprocess/davis/generate_flows.py:57-82,139-148, cut out of its source and executed as
they are: tests/ref_exec.py, tests/golden/make_golden.py) -- and (b) the oracle's restatement of the same lines.
Integer masks: bit-exact (zero mismatching pixels), on flows whose forward/backward error straddles the 1-pixel
threshold and whose targets straddle the border."""
import numpy as np
import pytest
import torch

import helpers
from helpers import flow_pair as _pair
from oracle import preprocess as OP

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('H,W,seed,noise', helpers.FLOW_MASK_CASES)
def test_masks_are_bit_identical_to_the_reference_formulas(H, W, seed, noise):
    from dvd_hip import preprocess as P
    f12, f21 = _pair(H, W, seed, noise)
    want_1, want_2 = OP.consistency_masks(f12.numpy(), f21.numpy())
    got_1, got_2 = P.flow_consistency_masks(f12.cuda(), f21.cuda())
    gd = helpers.load_golden('flow_masks')
    assert np.allclose(gd['flow_crc_%dx%d' % (H, W)], [float(f12.double().sum()), float(f21.double().sum())], rtol=1e-12), \
        'the seeded flows differ from the ones the fixture was generated on'
    for name, got, want in (('mask_1', got_1, want_1), ('mask_2', got_2, want_2)):
        got = got.cpu().numpy()
        assert got.dtype == np.uint8 and got.shape == want.shape
        assert 0.05 < want.mean() < 0.95, 'the case must exercise both outcomes'
        assert (got != want).sum() == 0, '%s: %d pixels differ from the oracle' % (name, (got != want).sum())
        ref = np.unpackbits(gd['%s_%dx%d' % (name, H, W)])[:H * W].reshape(H, W)
        assert (got != ref).sum() == 0, '%s: %d pixels differ from the reference code' % (name, (got != ref).sum())


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
