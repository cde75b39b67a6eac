"""Times the three scene-flow MLP kernels at the step's launch size (16 pairs of 384x672) through the C ABI."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'dynamic-video-depth_amd'))
from dvd_hip import ops  # noqa: E402


def timeit(fn, it=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def main():
    B, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 384, 672)))
    dev = 'cuda'
    nw = int(os.environ.get('MLP_NW', '0'))          # waves per workgroup of forward / dX (dvd_sf_mlp_select)
    f16 = os.environ.get('MLP_STASH_F16', '0') == '1'
    from dvd_hip import _lib
    _lib.check(_lib.load().dvd_sf_mlp_select(nw), 'dvd_sf_mlp_select')
    k = ops.SceneFlowMLPKernels(dev, 16, 16, True, stash_f16=f16)
    g = torch.Generator(device=dev).manual_seed(0)
    dims = [k.c_in] + [256] * 5
    Ws = [torch.randn(256 if i < 5 else 3, dims[i], device=dev, generator=g) / dims[i] ** 0.5 for i in range(6)]
    bs = [0.05 * torch.randn(256 if i < 5 else 3, device=dev, generator=g) for i in range(6)]
    k.pack(Ws, bs)
    p = 3 * torch.randn(B, 3, H, W, device=dev, generator=g)
    t = torch.rand(B, 1, H, W, device=dev, generator=g)
    n_pix = B * H * W
    if os.environ.get('MLP_UNCACHED', '0') != '0':
        # timing study: the stashes in memory the L2 does not cache (hipExtMallocWithFlags), so that their streams do not
        # evict the packed weights
        import ctypes
        hip = ctypes.CDLL('libamdhip64.so')
        flag = int(os.environ['MLP_UNCACHED'])           # 3 = hipDeviceMallocUncached, 1 = fine grained

        class Raw(object):
            def __init__(self, n):
                self.ptr = ctypes.c_void_p()
                rc = hip.hipExtMallocWithFlags(ctypes.byref(self.ptr), ctypes.c_size_t(4 * n), ctypes.c_uint(flag))
                assert rc == 0, rc
                self.__cuda_array_interface__ = {'shape': (n,), 'typestr': '<f4', 'data': (self.ptr.value, False), 'version': 2}
        raws = [Raw(k.stash_floats(n_pix)), Raw(k.gstash_floats(n_pix))]
        st, gst = (torch.as_tensor(r, device=dev) for r in raws)
    else:
        st, gst = k.new_stash(n_pix), k.new_gstash(n_pix)
    sf, gp = torch.empty_like(p), torch.empty_like(p)
    gout = torch.randn(B, 3, H, W, device=dev, generator=g)
    gW = [torch.zeros_like(w) for w in Ws]
    gb = [torch.zeros_like(b) for b in bs]
    flop = 593408.0 * n_pix
    rec = {'uncached': os.environ.get('MLP_UNCACHED', '0'), 'waves_per_workgroup': nw or 'default', 'stash_f16': f16, 'pixels': n_pix, 'stash_GB': st.numel() * 4 / 1e9, 'gstash_GB': gst.numel() * 4 / 1e9}
    ms = timeit(lambda: k.forward(p, t, 0.0, 0.01, sf_out=sf, stash=st))
    rec.update(fwd_ms=ms, fwd_tfs=flop / ms / 1e9)
    ms = timeit(lambda: k.forward(p, t, 0.0, 0.01, sf_out=sf))
    rec.update(fwd_nostash_ms=ms, fwd_nostash_tfs=flop / ms / 1e9)
    ms = timeit(lambda: k.backward_dx(st, 0.01, gout, gp, gst, gW[5], gb[5], (B, H, W)))
    rec.update(dx_ms=ms, dx_tfs=flop / ms / 1e9)
    ms = timeit(lambda: k.backward_dw(st, gst, n_pix, gW[:5], gb[:5]))
    rec.update(dw_ms=ms, dw_tfs=flop / ms / 1e9)
    # reproducibility of the weight gradients: two runs from zero must agree bitwise
    outs = []
    for _ in range(2):
        for x in gW + gb:
            x.zero_()
        k.backward_dw(st, gst, n_pix, gW[:5], gb[:5])
        outs.append([x.clone() for x in gW[:5] + gb[:5]])
    rec['dw_bitwise_reproducible'] = all(torch.equal(a, b) for a, b in zip(*outs))
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
