"""The index arithmetic of the scene-flow MLP's stash layout ("T8", csrc/sf_mlp.hip) as an executable
specification: no GPU, no library -- a numpy model of the accumulator layout of v_mfma_f32_32x32x16_f16, of
v_permlane16_swap and of the three places that address a stash block (the forward / dX epilogue's stores, the weight-gradient
kernel's chunk loads, the dX kernel's row-block reads for the last layer's weight gradient).  What it pins: every (channel,
pixel) of a layer's [256][64] block is written exactly once, to the address `t8_off(channel, t8_pos(pixel))`; the readers find
it there; the two operands of a weight-gradient chunk hold the SAME 16 pixels in the same order.

The GPU side of the same statement is tests/test_02_sf_mlp_gpu.py (gradients against the oracle / the real reference's
fixture, 4-wave and 8-wave kernels bit-identical)."""
import numpy as np

KTM = 64          # pixels per tile
WIDTH = 256       # hidden channels


def t8_pos(m):
    return (m & 15) * 4 + (m >> 4)


def t8_off(n, pos):
    return (n >> 3) * 512 + (pos >> 4) * 128 + (n & 7) * 16 + (pos & 15)


def permlane16_swap(vdst, vsrc):
    """v_permlane16_swap_b32: the odd rows (16 lanes each) of vdst are exchanged with the even rows of vsrc."""
    a, b = vdst.copy().reshape(4, 16), vsrc.copy().reshape(4, 16)
    a[1], b[0] = vsrc.reshape(4, 16)[0], vdst.reshape(4, 16)[1]
    a[3], b[2] = vsrc.reshape(4, 16)[2], vdst.reshape(4, 16)[3]
    return a.reshape(64), b.reshape(64)


def accumulator_quad(rt, q, ct):
    """What register 4q + e (e = 0..3) of the accumulator of (row tile rt, column tile ct) holds in lane L: channel
    32 rt + 8 q + 4 (L >> 5) + e, pixel 32 ct + (L & 31) -- encoded as channel * 64 + pixel."""
    lanes = np.arange(64)
    return [(32 * rt + 8 * q + 4 * (lanes >> 5) + e) * KTM + 32 * ct + (lanes & 31) for e in range(4)]


def test_pixel_position_map_is_a_permutation():
    pos = [t8_pos(m) for m in range(KTM)]
    assert sorted(pos) == list(range(KTM))
    # a run of four positions = the pixels jl, jl + 16, jl + 32, jl + 48
    for jl in range(16):
        assert [pos.index(4 * jl + k) for k in range(4)] == [jl + 16 * k for k in range(4)]


def test_block_offsets_are_a_bijection():
    offs = {t8_off(n, p) for n in range(WIDTH) for p in range(KTM)}
    assert offs == set(range(WIDTH * KTM))


def test_epilogue_stores_cover_the_block_once_at_the_documented_addresses():
    """store_t8: per (row tile, q) four exchanges (two channel pairs x two column tiles), then two 16-byte stores per lane."""
    mem = np.full(WIDTH * KTM, -1, dtype=np.int64)
    lanes = np.arange(64)
    jl, b4, hh = lanes & 15, (lanes >> 4) & 1, lanes >> 5
    lane_off = (jl >> 2) * 128 + (4 * hh + b4) * 16 + (jl & 3) * 4          # t8_lane_off
    for rt in range(8):
        for q in range(4):
            x, y = accumulator_quad(rt, q, 0), accumulator_quad(rt, q, 1)
            blk = t8_off(32 * rt + 8 * q, 0)
            for e1 in range(2):
                s0 = permlane16_swap(x[2 * e1], x[2 * e1 + 1])
                s1 = permlane16_swap(y[2 * e1], y[2 * e1 + 1])
                o = [s0[0], s0[1], s1[0], s1[1]]
                for k in range(4):
                    addr = blk + lane_off + 32 * e1 + k
                    assert (mem[addr] == -1).all()                           # written once
                    mem[addr] = o[k]
                # what the comment in csrc/sf_mlp.hip states about the lane's values
                for k in range(4):
                    ch, px = o[k] // KTM, o[k] % KTM
                    assert (ch == 32 * rt + 8 * q + 4 * hh + 2 * e1 + b4).all()
                    assert (px == jl + 16 * k).all()
    assert (mem >= 0).all()
    ch, px = mem // KTM, mem % KTM
    want = np.array([t8_off(int(c), t8_pos(int(p))) for c, p in zip(ch, px)])
    assert (want == np.arange(WIDTH * KTM)).all()


def _block_in_layout():
    mem = np.zeros(WIDTH * KTM, dtype=np.int64)
    for n in range(WIDTH):
        for m in range(KTM):
            mem[t8_off(n, t8_pos(m))] = n * KTM + m
    return mem


def test_weight_gradient_chunk_loads_see_the_same_pixels_in_both_operands():
    """dw_body::stage_load: thread (row = i * 128 + (tid >> 2), quad = tid & 3) reads four consecutive positions of chunk c;
    the embedding rows ([channel][64 positions]) are addressed row * 64 + 16 c + 4 quad."""
    mem = _block_in_layout()
    emb = np.zeros(WIDTH * KTM, dtype=np.int64)                # an embedding row stores pixel m at position t8_pos(m)
    for n in range(WIDTH):
        for m in range(KTM):
            emb[n * KTM + t8_pos(m)] = n * KTM + m
    seen = set()
    for chunk in range(4):
        pixels_of_chunk = None
        for i in range(2):
            for tid in range(512):
                row, quad = i * 128 + (tid >> 2), tid & 3
                got = mem[t8_off(row, chunk * 16 + quad * 4):][:4]
                assert (got // KTM == row).all()
                px = tuple(got % KTM)
                e = emb[row * KTM + chunk * 16 + quad * 4:][:4]
                assert tuple(e % KTM) == px and (e // KTM == row).all()        # both layouts: the same pixels, same order
                if row == 0:
                    pixels_of_chunk = (pixels_of_chunk or {})
                    pixels_of_chunk[quad] = px
                else:
                    assert pixels_of_chunk[quad] == px                      # every row: the same 16 pixels per chunk
                seen.update((row, p) for p in px)
        # 32 consecutive threads read 512 contiguous bytes (fp32) of one 8-channel block
        a = [t8_off(i32 >> 2, chunk * 16 + (i32 & 3) * 4) for i32 in range(32)]
        assert sorted(a) == list(range(a[0], a[0] + 128, 4)) and a[0] % 128 == 0
    assert len(seen) == WIDTH * KTM


def test_last_layer_weight_gradient_reads():
    """mlp_bwd_dx_kernel, dW5: wave w reads its channels' h4 blocks 16 bytes per lane and load; the comment states which
    channel and which pixel positions a lane gets (fp32 stash: two loads per 8-channel block; fp16: one)."""
    mem = _block_in_layout()
    lanes = np.arange(64)
    for nw in (4, 8):
        kcw = WIDTH // nw
        for w in range(nw):
            base = kcw * w * KTM                                  # elements (T8 blocks of a wave's channels are contiguous)
            covered = set()
            # fp32: 4 elements per lane and load, 256 elements per load
            for i in range(kcw // 4):
                for L in lanes:
                    got = mem[base + 256 * i + 4 * L:][:4]
                    ch = kcw * w + 8 * (i >> 1) + ((L >> 2) & 7)
                    p0 = 16 * (2 * (i & 1) + (L >> 5)) + 4 * (L & 3)
                    assert (got // KTM == ch).all()
                    assert [t8_pos(int(p)) for p in got % KTM] == [p0 + k for k in range(4)]
                    covered.update(int(g) for g in got)
            assert len(covered) == kcw * KTM
            # fp16: 8 elements per lane and load, 512 per load
            covered = set()
            for i in range(kcw // 8):
                for L in lanes:
                    got = mem[base + 512 * i + 8 * L:][:8]
                    ch = kcw * w + 8 * i + ((L >> 1) & 7)
                    p0 = 16 * (L >> 4) + 8 * (L & 1)
                    assert (got // KTM == ch).all()
                    assert [t8_pos(int(p)) for p in got % KTM] == [p0 + k for k in range(8)]
                    covered.update(int(g) for g in got)
            assert len(covered) == kcw * KTM
