"""The arithmetic every matrix kernel of this package uses (csrc/xconv.hip, xwgrad3.hip, sf_mlp.hip), emulated in numpy:
an fp32 operand is split exactly into three bf16 terms x = h + m + l (round to nearest even, like v_cvt_pk_bf16_f32) and
a product is the six largest of the nine partial products, accumulated in fp32, small terms first.  Pins the claims made
in DESIGN.md section 5.2: the split is exact, and a K = 2304 dot product (a 3x3 convolution over 256 channels) lands in
the error class of an fp32 FMA chain -- orders of magnitude below a plain bf16 product."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bf16 (ties to even), returned as float32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    return h, m, bf16_rne(r2)


def dot6(a, b):
    """sum_k a[k] * b[k] over the last axis with the kernels' term order and fp32 accumulation per term."""
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    acc = np.zeros(a.shape[:-1], dtype=np.float32)
    for x, y in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):      # (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
        # one MFMA accumulates 16 products per K step in fp32; chunked sums model that
        p = (x.astype(np.float32) * y.astype(np.float32)).astype(np.float32)        # bf16 x bf16 is exact in fp32
        for k0 in range(0, p.shape[-1], 16):
            acc = (acc + p[..., k0:k0 + 16].sum(-1, dtype=np.float32)).astype(np.float32)
    return acc


def test_three_term_split_is_exact():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(100000).astype(np.float32) * s for s in (1e-6, 1.0, 3e4)])
    h, m, l = split3(x)
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)).astype(np.float32), x)
    # every term is representable in bf16 (low 16 bits clear)
    for t in (h, m, l):
        assert not np.any(t.view(np.uint32) & 0xFFFF)


def test_six_term_product_is_fp32_class():
    rng = np.random.default_rng(1)
    K, n = 2304, 4096
    a = rng.standard_normal((n, K)).astype(np.float32)
    b = (rng.standard_normal((n, K)) / np.sqrt(K)).astype(np.float32)
    exact = (a.astype(np.float64) * b.astype(np.float64)).sum(-1)
    scale = np.abs(exact).max()
    err6 = np.abs(dot6(a, b).astype(np.float64) - exact).max() / scale
    fp32 = np.zeros(n, dtype=np.float32)
    for k in range(K):                                    # a sequential fp32 multiply-add chain
        fp32 = (fp32 + a[:, k] * b[:, k]).astype(np.float32)
    err32 = np.abs(fp32.astype(np.float64) - exact).max() / scale
    err_bf16 = np.abs((bf16_rne(a).astype(np.float64) * bf16_rne(b).astype(np.float64)).sum(-1) - exact).max() / scale
    print('six-term split %.2e, fp32 chain %.2e, plain bf16 %.2e of max|y|' % (err6, err32, err_bf16))
    assert err6 < 4e-6                      # the tolerance of tests/test_06_xconv_gpu.py
    assert err6 < 4 * err32                 # same class as an fp32 accumulation of the same data
    assert err_bf16 > 100 * err6            # what a plain bf16 product would cost
