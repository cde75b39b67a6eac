"""ops.new_scalar / the max|.| scalar pool across back-to-back HIP-graph captures.

A scalar chunk lives in the private pool of the graph that allocated it, and its zero-fill node replays with that graph
only.  Two captures in a row (the kept slots of Model._keep_slot: g_f, g_b, then the next slot's g_f, g_b with no eager
kernel call in between) must therefore never share a chunk, and a scalar tagged in one capture must not be served in
another.  No GPU here: the capture state and the allocation are mocked."""
import pytest
import torch

from dvd_hip import ops


@pytest.fixture
def fake_capture(monkeypatch):
    state = {'capturing': False, 'fills': []}
    monkeypatch.setattr(torch.cuda, 'is_current_stream_capturing', lambda: state['capturing'])
    real_zeros = torch.zeros

    def zeros(n, device=None, dtype=None):
        t = real_zeros(n, dtype=dtype)
        state['fills'].append((ops._capture_state[0] if state['capturing'] else 0, t))
        return t
    monkeypatch.setattr(ops.torch, 'zeros', zeros)
    ops._scalar_pool.clear()
    yield state
    ops._scalar_pool.clear()


def _storage(t):
    return t.untyped_storage().data_ptr()


def test_back_to_back_captures_never_share_a_chunk(fake_capture):
    st = fake_capture
    chunks = []
    for _ in range(4):                       # g_f, g_b of slot 0, g_f, g_b of slot 1: no eager call in between
        ops.begin_capture()
        st['capturing'] = True
        a, b = ops.new_scalar('cpu'), ops.new_scalar('cpu')
        assert _storage(a) == _storage(b)                   # one chunk inside a capture
        chunks.append(_storage(a))
        st['capturing'] = False
    assert len(set(chunks)) == 4
    # every chunk was zero-filled inside the capture that first hands it out
    assert [g for g, _ in st['fills']] == [1 + ops._capture_state[0] - 4 + i for i in range(4)]


def test_scalars_are_valid_in_their_own_capture_only(fake_capture):
    st = fake_capture
    t = torch.ones(4)
    ops.set_amax(t, torch.ones(1))           # eager tag
    assert ops.known_amax(t) is not None
    ops.begin_capture()
    st['capturing'] = True
    assert ops.known_amax(t) is None         # an eager scalar must not be baked into a graph
    ops.set_amax(t, torch.ones(1))
    assert ops.known_amax(t) is not None
    st['capturing'] = False
    ops.begin_capture()
    st['capturing'] = True
    assert ops.known_amax(t) is None         # ... nor another capture's
    st['capturing'] = False
    assert ops.known_amax(t) is None         # ... nor used eagerly


def test_eager_chunks_are_reused_across_captures(fake_capture):
    st = fake_capture
    a = ops.new_scalar('cpu')
    ops.begin_capture()
    st['capturing'] = True
    ops.new_scalar('cpu')
    st['capturing'] = False
    b = ops.new_scalar('cpu')
    assert _storage(a) == _storage(b) and a.data_ptr() != b.data_ptr()


def test_unannounced_captures_get_generations_of_their_own(fake_capture):
    """a `torch.cuda.graph` capture that does not call ops.begin_capture() (a test, a tool, a future capture
    site) must not continue the previous capture's generation: the first scalar call inside a capture that follows a call
    outside one opens a generation by itself."""
    st = fake_capture
    ops.begin_capture()
    st['capturing'] = True
    a = ops.new_scalar('cpu')
    g1 = ops._capture_gen()
    st['capturing'] = False
    ops.new_scalar('cpu')                    # an eager call between the two captures
    st['capturing'] = True                   # a capture nobody announced
    b = ops.new_scalar('cpu')
    g2 = ops._capture_gen()
    st['capturing'] = False
    assert g2 != g1 and g2 != 0
    assert _storage(a) != _storage(b)
    # an announced capture right after an unannounced one is not counted twice
    ops.begin_capture()
    st['capturing'] = True
    g3 = ops._capture_gen()
    st['capturing'] = False
    assert g3 == g2 + 1
