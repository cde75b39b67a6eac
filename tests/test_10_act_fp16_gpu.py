"""fp16 ACTIVATION storage (BASELINE configs[4]: "fp16 activations with fp32 loss accumulation") -- the kernels.

What is checked, and against what:
  * helper kernels (BatchNorm+ReLU, bilinear up-sampling, the 8-per-group 3x3) with fp16 storage do the SAME fp32 arithmetic as
    their fp32 versions and only round when they store: out16 == fp16(out32 of the same fp16-valued inputs), BIT-EXACT;
  * the matrix kernels (csrc/xconv.hip IN16 / OUT16, csrc/xwgrad3.hip H16) against float64 convolutions (the reference's
    nn.Conv2d, third_party/midas_blocks.py:102-168) of the SAME fp16-valued inputs: what remains is the two-term weight split
    (2^-22 per product) and, for activations, the fp16 rounding of the stored result: |err| <= 2^-11 |y| + 4e-6 max|y|;
    weight gradients (fp32 out) 2e-5 of max, the bound of the fp32 path;
  * the depth head / loss-scale boundary (csrc/a16.hip) against torch, and the loss-scale policy's transitions;
  * the MiDaS net with fp16 activations against the same net with fp32 activations: depth to 2e-3, parameter-gradient norms to
    2e-2 (the error budget of fp16 storage pinned by tests/test_split_bf16_cpu.py::test_fp16_activation_arithmetic_error_budget:
    2e-4 of max|y| per K = 2304 layer)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import log_measured, seeded_fill_

pytestmark = pytest.mark.gpu

H_EPS = 2.0 ** -11


def _h(t):
    """fp16-valued fp32 tensor (what the fp16 kernels see) and its fp16 copy on the GPU."""
    t16 = t.half()
    return t16.float(), t16.cuda()


def _state():
    from dvd_hip import conv as C, ops
    st = ops.gscale_new(torch.device('cuda'))
    C.set_grad_scale_state(st)
    return st


@pytest.fixture(autouse=True)
def _reset_state():
    yield
    from dvd_hip import conv as C
    C.set_grad_scale_state(None)


# ---- helper kernels: same arithmetic, rounded once at the store -------------------------------------------------------
def test_bnrelu_fp16_storage_is_the_fp32_kernel_rounded_once():
    from dvd_hip import conv as C
    torch.manual_seed(0)
    st = _state()
    for (N, Cc, H, W, res, relu) in ((2, 24, 12, 20, True, True), (3, 16, 7, 9, False, True), (2, 8, 64, 70, True, False)):
        bn = seeded_fill_(torch.nn.BatchNorm2d(Cc), 3).cuda().eval()
        x32, x16 = _h(torch.randn(N, Cc, H, W))
        r32, r16 = _h(torch.randn(N, Cc, H, W))
        g32, g16 = _h(torch.randn(N, Cc, H, W))
        outs = []
        for x, r, g in ((x32.cuda(), r32.cuda(), g32.cuda()), (x16, r16, g16)):
            x = x.requires_grad_(True)
            r = r.requires_grad_(True)
            bn.zero_grad()
            y = C.bn_eval_relu(bn, x, residual=r if res else None, relu=relu)
            y.backward(g)
            outs.append((y.detach(), x.grad, r.grad if res else None, bn.weight.grad.clone(), bn.bias.grad.clone()))
        (y32, gx32, gr32, gw32, gb32), (y16, gx16, gr16, gw16, gb16) = outs
        assert y16.dtype == torch.float16 and torch.equal(y16, y32.half())
        # backward: the ReLU mask comes from the STORED y (fp16 in one run, fp32 in the other): identical signs unless y32 is a
        # positive value that rounds to 0 in fp16 (none at these magnitudes)
        assert torch.equal(gx16, gx32.half())
        if res:
            assert torch.equal(gr16, gr32.half())
        # parameter gradients: fp32 sums of the same values (1 / S = 1 here: the state is fresh)
        assert torch.allclose(gw16, gw32, rtol=1e-5, atol=1e-5) and torch.allclose(gb16, gb32, rtol=1e-5, atol=1e-5)
    assert float(st[0]) == 1.0


@pytest.mark.parametrize('align', [False, True])
def test_upsample_fp16_storage_is_the_fp32_kernel_rounded_once(align):
    from dvd_hip import conv as C
    torch.manual_seed(1)
    for (N, Cc, H, W) in ((2, 8, 12, 20), (1, 5, 7, 10), (1, 3, 24, 42)):
        x32, x16 = _h(torch.randn(N, Cc, H, W))
        g32, g16 = _h(torch.randn(N, Cc, 2 * H, 2 * W))
        a = x32.cuda().requires_grad_(True)
        b = x16.requires_grad_(True)
        ya, yb = C.upsample_bilinear2x(a, align), C.upsample_bilinear2x(b, align)
        ya.backward(g32.cuda())
        yb.backward(g16)
        assert yb.dtype == torch.float16 and torch.equal(yb.detach(), ya.detach().half())
        assert torch.equal(b.grad, a.grad.half())


def test_gconv_c8_fp16_storage_is_the_fp32_kernel_rounded_once():
    from dvd_hip import conv as C
    torch.manual_seed(2)
    _state()
    for (N, Cc, H, W) in ((2, 32, 12, 20), (1, 64, 17, 70)):
        m = C.GroupedConv3x3C8(Cc).cuda()
        x32, x16 = _h(torch.randn(N, Cc, H, W))
        g32, g16 = _h(torch.randn(N, Cc, H, W))
        res = []
        for x, g in ((x32.cuda(), g32.cuda()), (x16, g16)):
            x = x.requires_grad_(True)
            m.zero_grad()
            y = m(x)
            y.backward(g)
            res.append((y.detach(), x.grad, m.weight.grad.clone()))
        assert torch.equal(res[1][0], res[0][0].half()) and torch.equal(res[1][1], res[0][1].half())
        assert torch.allclose(res[1][2], res[0][2], rtol=1e-5, atol=1e-5)


# ---- matrix kernels against float64 -----------------------------------------------------------------------------------
XCASES = [
    # N, Cin, Cout, H, W, KS, groups
    (2, 256, 256, 24, 42, 3, 1),
    (2, 64, 256, 24, 42, 1, 1),
    (1, 1024, 256, 12, 21, 3, 1),
    (1, 256, 128, 20, 36, 3, 1),
    (1, 128, 32, 30, 70, 3, 1),
    (1, 512, 512, 13, 22, 1, 1),
    (2, 256, 256, 12, 20, 3, 8),      # grouped, 32 per group (ResNeXt stage 3)
    (1, 512, 512, 9, 14, 3, 8),       # 64 per group (stage 4)
    (2, 16, 256, 7, 9, 3, 1),
    (2, 256, 256, 96, 128, 1, 1),
]


def _chk(name, got, want, rel, of_max):
    d = (got.double().cpu() - want).abs()
    bound = rel * want.abs() + of_max * float(want.abs().max())
    worst = float((d / bound).max())
    log_measured(name, worst, 1.0)
    assert worst <= 1.0, '%s: worst error / bound = %.3g' % (name, worst)


@pytest.mark.parametrize('N,Cin,Cout,H,W,KS,G', XCASES)
def test_conv_fp16_activations_forward_input_and_weight_gradient(N, Cin, Cout, H, W, KS, G):
    from dvd_hip import conv as C
    torch.manual_seed(Cin + Cout + KS + G)
    st = _state()
    conv = torch.nn.Conv2d(Cin, Cout, KS, padding=KS // 2, groups=G, bias=True)
    x32, x16 = _h(torch.randn(N, Cin, H, W))
    g32, g16 = _h(torch.randn(N, Cout, H, W))
    r32, r16 = _h(torch.randn(N, Cout, H, W))
    xd = x32.double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    bd = conv.bias.detach().double().requires_grad_(True)
    yd = F.conv2d(xd.relu(), wd, bd, padding=KS // 2, groups=G) + r32.double()
    yd.backward(g32.double())
    conv = conv.cuda()
    xg = x16.requires_grad_(True)
    y = C.xconv2d(conv, xg, relu_in=True, residual=r16)
    assert y.dtype == torch.float16
    y.backward(g16)
    tag = 'a16 conv %s' % ((N, Cin, Cout, H, W, KS, G),)
    _chk(tag + ' y', y.detach(), yd.detach(), 1.01 * H_EPS, 4e-6)
    _chk(tag + ' gx', xg.grad, xd.grad, 1.01 * H_EPS, 4e-6)
    _chk(tag + ' gw', conv.weight.grad, wd.grad, 0.0, 2e-5)
    _chk(tag + ' gb', conv.bias.grad, bd.grad, 0.0, 2e-5)
    # the backward-data epilogue reported the largest gradient magnitude it wrote to the loss-scale state
    assert float(st[3]) >= 0.99 * float(xg.grad.float().abs().max())


@pytest.mark.parametrize('N,Cc,G,H,W', [(2, 128, 4, 26, 45), (1, 128, 2, 50, 90), (1, 64, 4, 13, 22), (1, 256, 1, 21, 30)])
def test_stride_two_conv_fp16_activations(N, Cc, G, H, W):
    """The 3x3 / stride 2 convolutions at the entries of ResNeXt stages 2-4 (torchvision Bottleneck.conv2 through
    third_party/midas_blocks.py:35-50) on the strided kernels with fp16 activations (csrc/xconv.hip XArgs::S2 / ZI, IN16):
    forward, input gradient and weight gradient against float64; 16 per group through conv.GroupedConv3x3C16."""
    from dvd_hip import conv as C
    torch.manual_seed(Cc + G + H)
    st = _state()
    if Cc // G == 16:
        mod = C.GroupedConv3x3C16(Cc, stride=2)
    else:
        mod = C.XConv2d(Cc, Cc, 3, stride=2, padding=1, groups=G, bias=False)
    x32, x16 = _h(torch.randn(N, Cc, H, W))
    g32, g16 = _h(torch.randn(N, Cc, (H + 1) // 2, (W + 1) // 2))
    xd = x32.double().requires_grad_(True)
    wd = mod.weight.detach().double().requires_grad_(True)
    yd = F.conv2d(xd, wd, None, stride=2, padding=1, groups=G)
    yd.backward(g32.double())
    mod = mod.cuda()
    xg = x16.requires_grad_(True)
    y = mod(xg)
    assert y.dtype == torch.float16 and y.shape == yd.shape
    assert 'XConvS2' in type(y.grad_fn).__name__ or any('XConvS2' in type(n).__name__ for n, _ in y.grad_fn.next_functions)
    y.backward(g16)
    tag = 'a16 stride-2 conv %s' % ((N, Cc, G, H, W),)
    _chk(tag + ' y', y.detach(), yd.detach(), 1.01 * H_EPS, 4e-6)
    _chk(tag + ' gx', xg.grad, xd.grad, 1.01 * H_EPS, 4e-6)
    _chk(tag + ' gw', mod.weight.grad, wd.grad, 0.0, 2e-5)
    assert float(st[3]) >= 0.99 * float(xg.grad.float().abs().max())


def test_conv_bn_relu_fused_fp16():
    """conv + eval-mode BatchNorm + residual + ReLU in one launch, fp16 in / out, and its backward through the masked pass."""
    from dvd_hip import conv as C
    torch.manual_seed(5)
    _state()
    N, Cin, Cout, H, W = 2, 64, 256, 12, 20
    conv = torch.nn.Conv2d(Cin, Cout, 1, bias=False)
    bn = seeded_fill_(torch.nn.BatchNorm2d(Cout), 7).eval()
    x32, x16 = _h(torch.randn(N, Cin, H, W))
    g32, g16 = _h(torch.randn(N, Cout, H, W))
    r32, r16 = _h(torch.randn(N, Cout, H, W))
    cd, bd = conv.double(), bn.double()
    xd = x32.double().requires_grad_(True)
    yd = F.relu(bd(cd(xd)) + r32.double())
    yd.backward(g32.double())
    want = {k: v.grad.clone() for k, v in (('w', cd.weight), ('gamma', bd.weight), ('beta', bd.bias))}
    conv, bn = conv.float().cuda(), bn.float().cuda()
    conv.zero_grad(), bn.zero_grad()
    xg = x16.requires_grad_(True)
    y = C.conv_bn_act(conv, bn, xg, residual=r16, relu=True)
    y.backward(g16)
    _chk('a16 convbn y', y.detach(), yd.detach(), 1.01 * H_EPS, 4e-6)
    # ReLU' is taken from the fp16 y: elements whose float64 value is within fp16 rounding of 0 may flip; compare where |y| is clear
    clear = (yd.detach().abs() > 1e-3) | (yd.detach() == 0)
    d = ((xg.grad.double().cpu() - xd.grad).abs())
    assert float(d.max()) <= 3e-3 * float(xd.grad.abs().max())
    for k, p in (('w', conv.weight), ('gamma', bn.weight), ('beta', bn.bias)):
        err = float((p.grad.double().cpu() - want[k]).abs().max() / want[k].abs().max())
        assert err <= 2e-3, (k, err)
    assert bool(clear.any())


def test_wide_epilogue_of_the_1x1_kernels_is_bitwise_the_narrow_one_fp16():
    """dvd_xconv_select 6 / 7 switch the 1x1 kernels' two-chunk loop / LDS-transposed 16-byte epilogue off: same fp16 bits, with
    BatchNorm, residual and ReLU fused (forward) and on the masked backward-data pass."""
    from dvd_hip import _lib, conv as C
    torch.manual_seed(77)
    _state()
    N, Cin, Cout, H, W = 2, 256, 512, 12, 20
    conv = torch.nn.Conv2d(Cin, Cout, 1, bias=False).cuda()
    bn = seeded_fill_(torch.nn.BatchNorm2d(Cout), 9).eval().cuda()
    _, x16 = _h(torch.randn(N, Cin, H, W))
    _, g16 = _h(torch.randn(N, Cout, H, W))
    _, r16 = _h(torch.randn(N, Cout, H, W))
    lib = _lib.load()
    outs = []
    try:
        for cfg in (0, 6, 7):
            _lib.check(lib.dvd_xconv_select(cfg), 'dvd_xconv_select')
            xg = x16.clone().requires_grad_(True)
            y = C.conv_bn_act(conv, bn, xg, residual=r16, relu=True)
            y.backward(g16)
            outs.append((y.detach().clone(), xg.grad.clone()))
    finally:
        _lib.check(lib.dvd_xconv_select(0), 'dvd_xconv_select')
    for y, gx in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(gx, outs[0][1])


def test_weight_gradient_is_unscaled_by_the_loss_scale():
    """The fp16 gradients carry S; every parameter gradient is multiplied by state[1] = 1 / S where it is produced."""
    from dvd_hip import conv as C
    torch.manual_seed(6)
    st = _state()
    S = 256.0
    st[0], st[1] = S, 1.0 / S
    N, Cin, Cout, H, W = 2, 64, 64, 12, 20
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1, bias=True).cuda()
    x32, x16 = _h(torch.randn(N, Cin, H, W))
    g32, _ = _h(1e-3 * torch.randn(N, Cout, H, W))
    g16 = (g32 * S).half().cuda()                                  # what the backward hands around: S * g
    xd, wd = x32.double(), conv.weight.detach().double().cpu().requires_grad_(True)
    bd = conv.bias.detach().double().cpu().requires_grad_(True)
    F.conv2d(xd, wd, bd, padding=1).backward((g16.float().cpu() / S).double())
    y = C.xconv2d(conv, x16.requires_grad_(True))
    y.backward(g16)
    assert float((conv.weight.grad.double().cpu() - wd.grad).abs().max() / wd.grad.abs().max()) < 2e-5
    assert float((conv.bias.grad.double().cpu() - bd.grad).abs().max() / bd.grad.abs().max()) < 1e-3   # (fp16 torch.sum inputs)


# ---- the boundary and the loss-scale policy --------------------------------------------------------------------------
def test_head_boundary_forward_backward_and_scale():
    from dvd_hip import conv as C
    torch.manual_seed(7)
    st = _state()
    N, Cc, H, W = 2, 32, 12, 20
    conv = torch.nn.Conv2d(Cc, 1, 1).cuda()
    x32, x16 = _h(torch.randn(N, Cc, H, W))
    gy = (1e-6 * torch.randn(N, 1, H, W)).cuda()                    # tiny output gradient: would underflow fp16 unscaled
    xr = x32.cuda().requires_grad_(True)
    ref = F.conv2d(F.relu(xr), conv.weight, conv.bias)
    ref.backward(gy)
    want_w, want_b = conv.weight.grad.clone(), conv.bias.grad.clone()
    conv.zero_grad()
    xg = x16.requires_grad_(True)
    y = C.head1x1(conv, xg, relu_in=True)
    assert y.dtype == torch.float32 and torch.allclose(y, ref.detach(), rtol=1e-5, atol=1e-6)
    y.backward(gy)
    S = float(st[0])
    m = float(gy.abs().max() * conv.weight.detach().abs().max())
    assert S == 2.0 ** round(np.log2(S)) and 2.0 ** 3 < m * S <= 2.0 ** 4 and float(st[1]) == 1.0 / S
    got = xg.grad.float() / S
    assert float((got - xr.grad).abs().max()) <= 1.01 * H_EPS * float(xr.grad.abs().max())
    assert torch.allclose(conv.weight.grad, want_w, rtol=1e-4, atol=1e-12) and torch.allclose(conv.bias.grad, want_b, rtol=1e-4)
    assert abs(float(st[3]) - float(xg.grad.float().abs().max())) <= 1e-3 * float(st[3])


def test_loss_scale_policy_transitions_and_guarded_adam():
    from dvd_hip import ops
    st = ops.gscale_new(torch.device('cuda'))
    assert st.tolist() == [1.0, 1.0, 4.0] + [0.0] * 13      # 16 floats since ABI 7
    p = torch.ones(8, device='cuda')
    g, m, v = torch.ones(8, device='cuda'), torch.zeros(8, device='cuda'), torch.zeros(8, device='cuda')
    st[3] = 10000.0                                  # 2^13.3: centred, nothing changes, no skip
    ops.gscale_end(st)
    assert st[2:6].tolist() == [4.0, 0.0, 0.0, 0.0]
    ops.adam_step(p, g, m, v, 1, 0.1, 0.5, 0.9, skip_ptr=st[4:5])
    assert float(p[0]) < 1.0
    st[3] = 1000.0                                   # 2^9.97: three octaves of unused range -> +3
    ops.gscale_end(st)
    assert st[2:6].tolist() == [7.0, 0.0, 0.0, 0.0]
    st[3] = 1.0                                      # far below: at most +4 per step
    ops.gscale_end(st)
    assert st[2:6].tolist() == [11.0, 0.0, 0.0, 0.0]
    st[3] = 40000.0                                  # 2^15.3: still representable, but only 0.7 octaves left -> -2, no skip
    ops.gscale_end(st)
    assert st[2:6].tolist() == [9.0, 0.0, 0.0, 0.0]
    st[3] = 2.0 ** 20                                # seven octaves over 2^13: skip + back off by exactly that
    ops.gscale_end(st)
    assert st[2:6].tolist() == [2.0, 0.0, 1.0, 1.0]
    before = (p.clone(), m.clone(), v.clone())
    ops.adam_step(p, g, m, v, 2, 0.1, 0.5, 0.9, skip_ptr=st[4:5])
    assert torch.equal(p, before[0]) and torch.equal(m, before[1]) and torch.equal(v, before[2])
    st[3] = float('inf')
    ops.gscale_end(st)
    assert st[2:6].tolist() == [-6.0, 0.0, 1.0, 2.0]
    st[3] = 8192.0
    ops.gscale_end(st)
    assert st[2:6].tolist() == [-6.0, 0.0, 0.0, 2.0]
    # --- forward monitor.  An activation beyond fp16's range skips the step for BOTH networks -- [4] and the
    # activation-only pair [8], [9] -- without touching the target; consecutive such skips are counted in [10]
    st[3], st[6] = 8192.0, 70000.0
    ops.gscale_end(st)
    assert st[2:7].tolist() == [-6.0, 0.0, 1.0, 3.0, 0.0] and st[8:11].tolist() == [1.0, 1.0, 1.0]
    st[3], st[6] = 8192.0, float('nan')
    ops.gscale_end(st)
    assert st[4:6].tolist() == [1.0, 4.0] and st[8:11].tolist() == [1.0, 2.0, 2.0]
    # a mere loss-scale overflow skips the depth net's update only: the activation pair stays clear and its run ends
    st[3] = 2.0 ** 20
    ops.gscale_end(st)
    assert st[4:6].tolist() == [1.0, 5.0] and st[8:11].tolist() == [0.0, 2.0, 0.0]
    before = (p.clone(), m.clone(), v.clone())
    ops.adam_step(p, g, m, v, 3, 0.1, 0.5, 0.9, skip_ptr=st[4:5])          # depth net: skipped
    assert torch.equal(p, before[0]) and torch.equal(m, before[1])
    ops.adam_step(p, g, m, v, 3, 0.1, 0.5, 0.9, skip_ptr=st[8:10])         # scene-flow net: not skipped
    assert not torch.equal(p, before[0])
    # a stale overflow left behind by a pass that never reaches dvd_gscale_end (validation, warm-up, inference) is cleared by
    # the training step's first launch and does not skip that step
    st[6] = float('inf')
    ops.gscale_step_begin(st)
    assert float(st[6]) == 0.0
    st[3] = 8192.0
    ops.gscale_end(st)
    assert st[4:6].tolist() == [0.0, 5.0] and st[8:11].tolist() == [0.0, 2.0, 0.0]


def test_checkpoints_carry_the_effective_adam_step():
    """FlatNet.step_count advances on skipped steps too (the host never learns of a skip inside the step); the guarded Adam
    kernel subtracts the device-side skip count.  A checkpoint must carry step - skipped, the meaning torch.optim.Adam /
    GradScaler give it, and a resume must continue from there."""
    from dvd_hip import flat
    net = torch.nn.Linear(4, 4).cuda()
    fn = flat.FlatNet(net, 1e-3, (0.5, 0.9))
    skipped = torch.tensor([3.0], device='cuda')
    fn.skip_count = skipped
    fn.step_count = 10
    sd = fn.state_dict()
    assert float(sd['state'][0]['step']) == 7.0
    fn2 = flat.FlatNet(torch.nn.Linear(4, 4).cuda(), 1e-3, (0.5, 0.9))
    fn2.skip_count = torch.tensor([2.0], device='cuda')          # a process that has skipped two steps of its own
    fn2.load_state_dict(sd)
    assert fn2.step_count - fn2._skipped() == 7
    assert float(fn2.state_dict()['state'][0]['step']) == 7.0


# ---- the network ------------------------------------------------------------------------------------------------------
def test_midas_fp16_activations_against_fp32_activations():
    from dvd_hip import conv as C, ops
    from dvd_hip.third_party.MiDaS import MidasNet, calibrate_head_for_random_init
    torch.manual_seed(0)
    net = calibrate_head_for_random_init(MidasNet(non_negative=True, normalize_input=True)).cuda().eval()
    x = torch.rand(2, 3, 64, 96, device='cuda')
    gd = torch.randn(2, 1, 64, 96, device='cuda') * 1e-3
    res = []
    for dt in (torch.float32, torch.float16, torch.float16):
        net.act_dtype = dt
        C.set_grad_scale_state(ops.gscale_new(x.device) if dt == torch.float16 else None)
        net.zero_grad()
        d = net(x)
        d.backward(gd)
        # (refinenet4.resConfUnit1 is never used -- refinenet4 has one input, third_party/MiDaS.py:228 -- in either mode)
        missing = sorted(k for k, p in net.named_parameters() if p.grad is None)
        assert all(k.startswith('scratch.refinenet4.resConfUnit1.') for k in missing), (dt, missing[:8])
        res.append((d.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}))
    (d32, g32), (d16, g16), (d16b, g16b) = res
    assert d16.dtype == torch.float32
    # the fp16 kernels are deterministic: two passes give bit-identical depths and parameter gradients (the 7x7 stem's weight
    # gradient runs on MIOpen, whose reduction order is not fixed: compared to 1e-5)
    assert torch.equal(d16, d16b)
    for k in g16:
        if k.startswith('pretrained.layer1.0.'):
            assert torch.allclose(g16[k], g16b[k], rtol=1e-5, atol=1e-5 * float(g16[k].abs().max())), k
        else:
            assert torch.equal(g16[k], g16b[k]), k
    e_d = float((d16 - d32).abs().max() / d32.abs().max())
    worst, name = 0.0, None
    for k in g32:
        n32 = float(g32[k].double().norm())
        if n32 == 0.0:
            continue
        r = abs(float(g16[k].double().norm()) - n32) / n32
        if r > worst:
            worst, name = r, k
    log_measured('midas fp16 activations vs fp32: depth rel', e_d, 2e-3)
    log_measured('midas fp16 activations vs fp32: worst gradient-norm rel (%s)' % name, worst, 2e-2)
    print('fp16 activations: depth %.2e, worst gradient norm %.2e (%s)' % (e_d, worst, name))
    assert e_d < 2e-3 and worst < 2e-2


# ---- the step ---------------------------------------------------------------------------------------------------------
def test_full_step_fp16_activations_against_the_reference_fixture():
    """`--act_fp16` step against the REAL reference's fp32 CPU step (tests/golden/fullstep_midas_b2_192x384_train.npz, the
    BASELINE configs[0] shape): stated tolerance of the fp16-activation mode -- losses rtol 2e-3, per-parameter gradient norms
    5e-2 (the fp32-storage step meets 1e-5 / 1.5e-3 on the same fixture, tests/test_30_full_step_gpu.py); no skipped step."""
    import test_30_full_step_gpu as T30
    import helpers
    gd = helpers.load_golden('fullstep_midas_b2_192x384_train')
    model, opt, batch = T30._build(gd, act_fp16=True)
    log = model._train_on_batch(int(gd['epoch']), 0, helpers.loader_batch(batch))
    torch.cuda.synchronize()
    st = model._gscale.tolist()
    assert st[4] == 0.0 and st[5] == 0.0, 'the step overflowed fp16: %s' % (st,)
    loss_rel = max(abs(log[k] - float(gd['log_' + k])) / abs(float(gd['log_' + k]))
                   for k in ('loss', 'flow_loss_1_2', 'disp_loss_1_2', 'sf_loss'))
    names = [str(n) for n in gd['param_names']]
    want_g = dict(zip(names, gd['grad_norms']))
    worst, wname = 0.0, None
    for prefix, net in (('depth', model.net_depth), ('sf', model.net_sceneflow)):
        for k, p in net.named_parameters():
            w = want_g[prefix + '/' + k]
            if w == 0.0:
                continue
            r = abs(float(p.grad.double().norm()) - w) / w
            if r > worst:
                worst, wname = r, prefix + '/' + k
    log_measured('fp16-activation step vs reference fixture: loss rel', loss_rel, 2e-3)
    log_measured('fp16-activation step vs reference fixture: worst gradient-norm rel (%s)' % wname, worst, 5e-2)
    print('fp16-activation step: loss rel %.2e, worst gradient norm %.2e (%s), loss scale 2^%d, observed max %.0f' % (
        loss_rel, worst, wname, int(np.log2(st[0])), st[3]))
    assert loss_rel < 2e-3 and worst < 5e-2


def test_fp16_step_graph_replay_matches_eager():
    """Two steps with captured graphs (kept slots) equal two steps run eagerly: the loss-scale state, the head boundary and the
    fp16 kernels are all capture-safe (every scalar they read is written inside the same graph or lives in the persistent state)."""
    import test_30_full_step_gpu as T30
    import helpers
    gd = helpers.load_golden('fullstep_midas_b1_64x96_train')
    logs = []
    for graphs in (0, 0, 1):                  # eager twice (the mode itself is deterministic), then replayed graphs
        model, opt, batch = T30._build(gd, act_fp16=True, depth_graphs=graphs)
        trace = []
        for i in range(3):
            log = model._train_on_batch(int(gd['epoch']), i, helpers.loader_batch(dict(batch)))
            trace.append((log['loss'], [round(v, 3) for v in model._gscale.tolist()[:6]]))
        torch.cuda.synchronize()
        logs.append((log['loss'], float(model._flat_depth.grad.double().norm()), trace))
        del model
    print('\n'.join(str(l) for l in logs))
    # Steps 1 and 2 (same weights up to Adam's sign-like first update) are identical in all three runs; from step 3 on two EAGER
    # runs differ by ~1e-4 themselves: the stem's MIOpen weight gradient is not bitwise reproducible (1e-7), and fp16 STORAGE
    # is a discontinuous function of the weights -- a 1e-7 perturbation moves some activations across an fp16 rounding boundary
    # (2^-11 each).  The replayed graphs must agree with eager execution to that same noise level, and exactly in the loss-scale
    # decisions.
    for a, b in ((0, 1), (0, 2)):
        assert [t[0] for t in logs[a][2][:2]] == [t[0] for t in logs[b][2][:2]], (a, b)
        assert [t[1] for t in logs[a][2]] == [t[1] for t in logs[b][2]], 'loss-scale decisions differ'
        assert abs(logs[a][0] - logs[b][0]) <= 5e-4 * abs(logs[a][0]) and abs(logs[a][1] - logs[b][1]) <= 2e-2 * logs[a][1], logs


@pytest.mark.parametrize('where,value', [('activation', float('inf')), ('activation', float('nan')),
                                         ('gradient', float('nan')), ('gradient', float('inf'))])
def test_non_finite_fp16_values_skip_the_step(where, value):
    """the overflow guard must see NaNs and FORWARD overflows.  A non-finite value planted in an fp16
    activation (a forward hook on a decoder convolution) or in an fp16 activation gradient (a tensor hook) must set the skip
    flag -- every observed-maximum reduction records a NaN as +Inf (csrc/dvd_common.h amax_acc), the fp16-output convolution
    epilogues and the depth head fold max|y| into the forward monitor (state[6]) -- and the guarded Adam step must leave the
    depth net's parameters untouched.  A clean step on the same model right after is NOT skipped."""
    import test_30_full_step_gpu as T30
    import helpers
    gd = helpers.load_golden('fullstep_midas_b1_64x96_train')
    model, opt, batch = T30._build(gd, act_fp16=True, depth_graphs=0)
    net = model.net_depth
    # the first convolution of the output head: fp16 in and out, and its consumers (bilinear up-sampling, a convolution WITHOUT
    # an input ReLU) pass a non-finite value on.  (Behind most other layers a NaN ACTIVATION is absorbed: the fused kernels
    # evaluate ReLU as fmaxf(x, 0), which maps NaN to 0 in the forward, in the masks and in the weight gradients alike --
    # nothing is poisoned and no step needs skipping; torch.relu would have propagated it.)
    target = net.scratch.output_conv[0]
    state = {'armed': True}

    def plant(t):
        t = t.clone()
        t.view(-1)[t.numel() // 3] = value
        return t

    def fwd_hook(mod, inp, out):
        if not state['armed']:
            return None
        assert out.dtype == torch.float16
        if where == 'activation':
            return plant(out)
        if out.requires_grad:                 # (phase 1 runs the forward without autograd; phase 3 recomputes it with)
            out.register_hook(lambda g: plant(g) if state['armed'] else g)
        return None
    h = target.register_forward_hook(fwd_hook)
    before = [p.detach().clone() for p in net.parameters()]
    log = model._train_on_batch(int(gd['epoch']), 0, helpers.loader_batch(dict(batch)))
    torch.cuda.synchronize()
    st = model._gscale.tolist()
    assert st[4] == 1.0 and st[5] == 1.0, 'the step with a non-finite %s was not skipped: state %r' % (where, st)
    # an activation overflow also invalidates the scene-flow network's gradient (flag [8]); a non-finite fp16 GRADIENT of the
    # depth net does not: the depth maps were finite, that network's fp32 update goes through
    assert st[8] == (1.0 if where == 'activation' else 0.0) and st[9] == st[8], st
    assert log['steps_skipped'] == 1
    for p, b in zip(net.parameters(), before):
        assert torch.equal(p.detach(), b), 'a skipped step changed the depth net'
    state['armed'] = False
    log = model._train_on_batch(int(gd['epoch']), 1, helpers.loader_batch(dict(batch)))
    torch.cuda.synchronize()
    st = model._gscale.tolist()
    h.remove()
    assert st[4] == 0.0 and st[5] == 1.0 and np.isfinite(log['loss']), st
    assert any(not torch.equal(p.detach(), b) for p, b in zip(net.parameters(), before))
    # a stale forward monitor -- what a validation pass, a warm-up epoch or an inference call with an overflowing activation
    # leaves behind, none of them reaches dvd_gscale_end -- must not skip the NEXT training step
    model._gscale[6] = float('inf')
    log = model._train_on_batch(int(gd['epoch']), 2, helpers.loader_batch(dict(batch)))
    torch.cuda.synchronize()
    st = model._gscale.tolist()
    assert st[4] == 0.0 and st[5] == 1.0 and st[10] == 0.0 and log['steps_skipped'] == 1, st


def test_fp16_mlp_stash_overflow_skips_the_step_instead_of_poisoning_the_weights():
    """Round 6: the scene-flow MLP's fp16 stash (implied by --act_fp16) stores the hidden activations as _Float16; one beyond
    65504 is stashed as Inf and the weight gradients contracted against it are not finite -- before the guard, one such step
    turned every MLP parameter into NaN for good (this fixture's seeded MLP weights put its activations next to fp16's range:
    the second step overflows).  The MLP forward now folds max |h| into the step's forward monitor (dvd_mlp_desc.fwd_monitor =
    slot [6] of the loss-scale state): the step is skipped for BOTH networks, every parameter stays finite, and the run says so
    (steps_skipped in the log, a warning from the third consecutive skip on)."""
    import warnings
    import test_30_full_step_gpu as T30
    import helpers
    gd = helpers.load_golden('fullstep_midas_b1_64x96_train')
    model, opt, batch = T30._build(gd, act_fp16=True, depth_graphs=1, depth_chunk=1)
    batch['flow_1_2'] = batch['flow_1_2'] * 0.25
    batch['flow_2_1'] = batch['flow_2_1'] * 0.25
    skipped, warned = [], 0
    for i in range(5):
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter('always')
            log = model._train_on_batch(int(gd['epoch']), i, helpers.loader_batch(dict(batch)))
        torch.cuda.synchronize()
        warned += sum('activation' in str(w.message) for w in rec)
        skipped.append(log['steps_skipped'])
        assert bool(torch.isfinite(model._flat_sf.flat).all()) and bool(torch.isfinite(model._flat_depth.flat).all()), \
            'step %d left non-finite parameters (state %r)' % (i, model._gscale.tolist())
    st = model._gscale.tolist()
    print('fp16 MLP stash overflow: steps skipped', skipped, 'state', [round(v, 1) for v in st[:11]], 'warnings', warned)
    assert skipped[0] == 0 and skipped[-1] >= 1, skipped
    assert st[9] >= 1.0, 'no step was skipped for an activation overflow: %r' % st
    if st[10] >= 3.0:
        assert warned >= 1


# ---- the scene-flow MLP's fp16 stash ----------------------------------------------------------------------------------
def test_mlp_fp16_stash_changes_only_the_weight_gradients():
    """dvd_mlp_desc.stash_f16: the hidden activations h_0 .. h_4 of the stash are stored as fp16 (networks/sceneflow_field.py:43-53
    is the layer stack they belong to).  The forward output and the input gradient are BIT-identical to the fp32 stash (neither
    reads the stored activations: the dX chain uses the sign bits); the weight gradients contract fp32 gradients against
    fp16-rounded activations: <= 8e-4 of max|g| per tensor (measured on MI355X: 3.3e-4 worst, layer 3), biases exact."""
    from dvd_hip import ops
    from oracle import sceneflow_mlp as M
    sd = M.init_params(seed=4)
    B, H, W = 2, 24, 40
    g = torch.Generator().manual_seed(11)
    p = (2.0 * torch.randn(B, 3, H, W, generator=g)).cuda()
    ts = torch.rand(B, 1, 1, 1, generator=g).expand(B, 1, H, W).contiguous().cuda()
    gout = torch.randn(B, 3, H, W, generator=g).cuda()
    n_pix = B * H * W
    res = []
    for s16 in (False, True):
        k = ops.SceneFlowMLPKernels('cuda', 16, 16, True, stash_f16=s16)
        k.pack([sd['convs.%d.conv.weight' % i].cuda() for i in range(6)], [sd['convs.%d.conv.bias' % i].cuda() for i in range(6)])
        stash, gst = k.new_stash(n_pix), k.new_gstash(n_pix)
        sf, g_p = torch.empty_like(p), torch.empty_like(p)
        k.forward(p, ts, 0.0, 0.01, sf_out=sf, stash=stash)
        dims = [k.c_in] + [256] * 5
        gW = [torch.zeros(256 if i < 5 else 3, dims[i], device='cuda') for i in range(6)]
        gb = [torch.zeros(256 if i < 5 else 3, device='cuda') for i in range(6)]
        k.backward_dx(stash, 0.01, gout, g_p, gst, gW[5], gb[5], (B, H, W))
        k.backward_dw(stash, gst, n_pix, gW[:5], gb[:5])
        res.append((sf, g_p, gW, gb, stash.numel()))
    (sf32, gp32, gW32, gb32, n32), (sf16, gp16, gW16, gb16, n16) = res
    assert n16 < 0.6 * n32
    assert torch.equal(sf32, sf16) and torch.equal(gp32, gp16)
    for i in range(6):
        assert torch.allclose(gb32[i], gb16[i], rtol=1e-6, atol=1e-7 * float(gb32[i].abs().max()))      # (layer 5: atomics)
        e = float((gW32[i] - gW16[i]).abs().max() / gW32[i].abs().max())
        log_measured('mlp fp16 stash: dW_%d vs fp32 stash, of max' % i, e, 8e-4)
        assert e <= (1e-6 if i == 0 else 8e-4), (i, e)          # layer 0 contracts against the fp32 embedding
