"""Host-side logic of the flat parameter / gradient buffers (dvd_hip/flat.py) that needs no GPU:
views stay attached, multi-tensor gradient accumulation over several backward passes equals autograd's
own per-parameter accumulation, checkpoints carry the fused-Adam state."""
import torch

from dvd_hip import flat


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))


def test_parameters_and_gradients_are_views_of_the_flat_buffers():
    net = _net()
    before = [p.detach().clone() for p in net.parameters()]
    fn = flat.FlatNet(net, 1e-3, (0.5, 0.9))
    assert fn.numel % 4 == 0 and all(o % 4 == 0 for o in fn.offsets)          # 16-byte aligned segments
    for p, o, b in zip(net.parameters(), fn.offsets, before):
        assert torch.equal(p.detach(), b)
        assert p.data_ptr() == fn.flat.data_ptr() + 4 * o and p.grad.data_ptr() == fn.grad.data_ptr() + 4 * o
    fn.flat.mul_(2.0)                                                          # one op on the flat buffer moves every parameter
    for p, b in zip(net.parameters(), before):
        assert torch.equal(p.detach(), 2 * b)


def test_multi_tensor_accumulation_equals_autograd_accumulation():
    net = _net().eval()
    fn = flat.FlatNet(net, 1e-3, (0.5, 0.9))
    xs = [torch.randn(2, 3, 8, 8) for _ in range(3)]
    fn.zero_grad()
    for x in xs:                                  # what the depth-net backward does per chunk
        fn.detach_grads()
        net(x).square().sum().backward()
        fn.absorb_grads()
    got = fn.grad.clone()
    assert all(p.grad.data_ptr() == fn.grad.data_ptr() + 4 * o for p, o in zip(net.parameters(), fn.offsets))
    fn.zero_grad()
    for x in xs:
        net(x).square().sum().backward()          # autograd adds into the attached views itself
    assert torch.allclose(got, fn.grad, rtol=1e-6, atol=1e-7)
    # a parameter that received no gradient keeps its view and its accumulated value
    fn.zero_grad()
    fn.grad.fill_(1.0)
    fn.detach_grads()
    net[0].weight.grad = torch.ones_like(net[0].weight)
    fn.absorb_grads()
    assert float(fn.view(fn.grad, 0).min()) == 2.0 and float(fn.view(fn.grad, 1).max()) == 1.0


def test_state_dict_round_trip():
    a, b = flat.FlatNet(_net(), 1e-3, (0.5, 0.9)), flat.FlatNet(_net(), 1e-3, (0.5, 0.9))
    a.exp_avg.normal_()
    a.exp_avg_sq.uniform_()
    a.step_count = 7
    b.load_state_dict(a.state_dict())
    assert b.step_count == 7
    for i in range(len(a.params)):               # (the alignment padding between segments carries no state)
        assert torch.equal(a.view(a.exp_avg, i), b.view(b.exp_avg, i))
        assert torch.equal(a.view(a.exp_avg_sq, i), b.view(b.exp_avg_sq, i))
