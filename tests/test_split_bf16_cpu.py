"""The arithmetic of the matrix kernels of this package (csrc/xconv.hip, xwgrad3.hip, sf_mlp.hip; csrc/dvd_split.h), emulated
in numpy.  Round 3 (what the kernels run): every operand tensor is scaled by a power of two taken from its max|x| and split
into TWO fp16 terms (22 bits), a product is THREE partial products l*h' + h*l' + h*h' accumulated in fp32, small terms first
(pow2_scale / split2_f16 / dot3 below).  Round 2 (kept as the yardstick): an exact split into three bf16 terms and the six
largest of nine partial products (split3 / dot6).  Pins the claims of DESIGN.md section 5.0: the bf16 split is exact, and a
K = 2304 dot product (a 3x3 convolution over 256 channels), a heavy-tailed gradient and a K = 16384 weight gradient land,
in BOTH arithmetics, in the error class of an fp32 multiply-add chain -- orders of magnitude below a plain 16-bit product."""
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


# ---- round 3: two fp16 terms of the power-of-two-scaled operand, three partial products (csrc/dvd_split.h) ----------------
def pow2_scale(amax):
    """2^e with amax * 2^e in [2^13, 2^14): the device function of csrc/dvd_split.h."""
    amax = np.float32(amax)
    if not (amax > 0) or not np.isfinite(amax):
        return np.float32(1.0)
    e = int((amax.view(np.uint32) >> 23) & 0xff)
    se = min(max(267 - e, 1), 254)
    return np.uint32(se << 23).view(np.float32)


def split2_f16(x, s):
    xs = (np.asarray(x, dtype=np.float32) * s).astype(np.float32)          # exact: s is a power of two
    h = xs.astype(np.float16)                                               # round to nearest even, like v_cvt_pk_f16_f32
    l = (xs - h.astype(np.float32)).astype(np.float32).astype(np.float16)   # xs - h is exact in fp32
    return h.astype(np.float32), l.astype(np.float32)


def dot3(a, b):
    """sum_k a[k] * b[k] with the kernels' arithmetic: l*h' + h*l' + h*h', fp32 accumulation per 16-product MFMA step,
    unscaled by the exact power of two at the end."""
    sa, sb = pow2_scale(np.abs(a).max()), pow2_scale(np.abs(b).max())
    ah, al = split2_f16(a, sa)
    bh, bl = split2_f16(b, sb)
    acc = np.zeros(a.shape[:-1], dtype=np.float32)
    for x, y in ((al, bh), (ah, bl), (ah, bh)):
        p = (x * y).astype(np.float32)                                      # fp16 x fp16 is exact in fp32
        for k0 in range(0, p.shape[-1], 16):
            acc = (acc + p[..., k0:k0 + 16].sum(-1, dtype=np.float32)).astype(np.float32)
    return acc.astype(np.float64) / (np.float64(sa) * np.float64(sb))


def test_pow2_scale_puts_the_maximum_below_2_to_14():
    for amax in (1e-30, 3e-7, 0.02, 1.0, 1.9999, 2.0, 77.0, 65504.0, 3e20):
        s = pow2_scale(amax)
        assert 2.0 ** 13 <= np.float32(amax) * s < 2.0 ** 14 or s in (np.float32(2.0 ** 127), np.float32(2.0 ** -126))
        assert np.log2(float(s)) == np.round(np.log2(float(s)))
    assert pow2_scale(0.0) == 1.0 and pow2_scale(np.inf) == 1.0 and pow2_scale(np.nan) == 1.0


def test_two_fp16_terms_carry_22_bits():
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(200000) * np.exp(3 * rng.standard_normal(200000))).astype(np.float32) * np.float32(1e-5)
    s = pow2_scale(np.abs(x).max())
    h, l = split2_f16(x, s)
    back = (h.astype(np.float64) + l.astype(np.float64)) / np.float64(s)
    err = np.abs(back - x.astype(np.float64))
    big = np.abs(x) >= np.abs(x).max() * 2.0 ** -17             # l is a normal fp16 number: full 22 bits
    assert (err[big] <= np.abs(x[big]) * 2.0 ** -21).all()
    assert (err <= np.abs(x).max() * 2.0 ** -38).all() or (err[~big] <= 2.0 ** -25 / np.float64(s)).all()


def test_three_product_fp16_pair_is_fp32_class():
    """The error of the round-3 arithmetic against float64 on the shapes of the depth net (K = 2304: a 3x3 convolution over
    256 channels; K = 256: a 1x1; K = 16384 pixels: a weight gradient) is in the class of the round-2 six-product bf16
    arithmetic and BELOW a sequential fp32 multiply-add chain -- with half the MFMAs."""
    rng = np.random.default_rng(1)
    cases = (
        ('3x3 conv', lambda: (rng.standard_normal((1024, 2304)).astype(np.float32),
                              (rng.standard_normal((1024, 2304)) / np.sqrt(2304)).astype(np.float32))),
        ('relu x small weights', lambda: ((np.maximum(rng.standard_normal((1024, 2304)), 0) * 3).astype(np.float32),
                                          (rng.standard_normal((1024, 2304)) * 0.02).astype(np.float32))),
        ('heavy-tailed gradients', lambda: ((rng.standard_normal((2048, 256)) * np.exp(2 * rng.standard_normal((2048, 256))) * 1e-6).astype(np.float32),
                                            (rng.standard_normal((2048, 256)) * 0.05).astype(np.float32))),
        ('weight gradient', lambda: ((rng.standard_normal((256, 16384)) * 1e-5).astype(np.float32),
                                     np.maximum(rng.standard_normal((256, 16384)), 0).astype(np.float32))),
    )
    for name, gen in cases:
        a, b = gen()
        exact = (a.astype(np.float64) * b.astype(np.float64)).sum(-1)
        scale = np.abs(exact).max()
        e3 = np.abs(dot3(a, b) - exact).max() / scale
        e6 = np.abs(dot6(a, b).astype(np.float64) - exact).max() / scale
        chain = np.zeros(a.shape[0], dtype=np.float32)
        prod = (a * b).astype(np.float32)
        for k in range(a.shape[1]):
            chain = (chain + prod[:, k]).astype(np.float32)
        e32 = np.abs(chain.astype(np.float64) - exact).max() / scale
        print('%-24s fp16 pair x3 %.2e | bf16 triple x6 %.2e | fp32 chain %.2e of max|y|' % (name, e3, e6, e32))
        assert e3 < 4e-6                 # the tolerance of tests/test_06_xconv_gpu.py
        assert e3 < 4 * e6 + 1e-8        # same class as the six-product arithmetic
        assert e3 < 1.5 * e32            # not worse than an fp32 multiply-add chain over the same K


# ---- groundwork for BASELINE configs[4] (fp16 activations, fp32 accumulation; NOT built) -------------
def dot2_fp16_activations(a16, b):
    """sum_k a[k] * b[k] for an activation tensor a that is STORED in fp16 (one term, scaled by a power of two from its
    maximum like every operand) against a two-term weight: two partial products a*l' + a*h' -- two MFMAs per product
    instead of three, and half the activation bytes."""
    sa, sb = pow2_scale(np.abs(a16).max()), pow2_scale(np.abs(b).max())
    a1 = (a16.astype(np.float32) * sa).astype(np.float16).astype(np.float32)
    bh, bl = split2_f16(b, sb)
    acc = np.zeros(a16.shape[:-1], dtype=np.float32)
    for x, y in ((a1, bl), (a1, bh)):
        p = (x * y).astype(np.float32)
        for k0 in range(0, p.shape[-1], 16):
            acc = (acc + p[..., k0:k0 + 16].sum(-1, dtype=np.float32)).astype(np.float32)
    return acc.astype(np.float64) / (np.float64(sa) * np.float64(sb))


def test_fp16_activation_arithmetic_error_budget():
    """What configs[4] would cost in accuracy, so that its parity tolerance can be stated before it is built: with
    activations rounded ONCE to fp16 (11 bits) and weights kept at 22 bits, (1) the matrix arithmetic itself adds nothing
    measurable to that rounding (against the float64 product of the ROUNDED activations it stays in the fp32 class), and
    (2) against fp32 activations a K = 2304 convolution output moves by ~1e-4 of max|y| (the 2^-12 relative rounding of every
    activation, averaged over K terms) -- three orders above this package's fp32-class bound, the price of the format."""
    rng = np.random.default_rng(11)
    a = (np.maximum(rng.standard_normal((512, 2304)), 0) * 3).astype(np.float32)          # post-ReLU activations
    b = (rng.standard_normal((512, 2304)) * 0.02).astype(np.float32)
    sa = pow2_scale(np.abs(a).max())
    a16 = ((a * sa).astype(np.float16).astype(np.float32) / sa).astype(np.float32)        # what HBM would hold (scaled fp16)
    exact_rounded = (a16.astype(np.float64) * b.astype(np.float64)).sum(-1)
    exact_fp32 = (a.astype(np.float64) * b.astype(np.float64)).sum(-1)
    got = dot2_fp16_activations(a16, b)
    scale = np.abs(exact_fp32).max()
    e_arith = np.abs(got - exact_rounded).max() / scale
    e_format = np.abs(exact_rounded - exact_fp32).max() / scale
    print('fp16 activations: arithmetic %.2e, storage format %.2e of max|y|' % (e_arith, e_format))
    assert e_arith < 4e-6                       # two products against two-term weights: still fp32 class
    assert 1e-5 < e_format < 1e-3               # the activation rounding dominates: ~1e-4
