"""dvd_flop_counters / ops.executed_flops (ABI 5): the algorithmic-work accounting behind bench.py's `roofline_mfma`.  Host-side
only -- the counters live in the library and are written at launch time, so without a GPU they can be read and reset, and the
Python bookkeeping of replayed graphs can be exercised."""
import ctypes
import re
import os

from dvd_hip import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_classes_match_the_header_and_the_trace_join_table():
    text = open(os.path.join(ROOT, 'include', 'dvd_hip.h')).read()
    m = re.search(r'DVD_FLOP_CLASSES\s*=\s*(\d+)', text)
    assert m and int(m.group(1)) == len(ops.FLOP_CLASSES)
    enum = re.findall(r'DVD_FLOP_([A-Z0-9_]+)\s*=\s*(\d+),', text)
    assert [int(v) for _, v in enum] == list(range(len(ops.FLOP_CLASSES)))
    assert [n.lower() for n, _ in enum] == [c.replace('bwd_', '') for c in ops.FLOP_CLASSES]
    assert set(ops.FLOP_CLASS_KERNELS) == set(ops.FLOP_CLASSES)
    # every class can be found in both spellings of a kernel name (rocpd database: mangled, rocprofv3 csv: demangled)
    demangled = 'void dvd::xconv_kernel<4, 2, 2, 2, 11, true, false, false, false>(dvd::XArgs)'
    mangled = '_ZN3dvd12xconv_kernelILi4ELi2ELi2ELi2ELi11ELb1ELb0ELb0ELb0EEEvNS_5XArgsE.kd'
    for name in (demangled, mangled):
        hits = [c for c, frags in ops.FLOP_CLASS_KERNELS.items() if any(f in name for f in frags)]
        assert hits == ['xconv_1x1_wide'], (name, hits)


def test_counters_read_reset_and_replay_bookkeeping():
    lib = _lib.load()
    buf = (ctypes.c_double * len(ops.FLOP_CLASSES))()
    assert lib.dvd_flop_counters(ctypes.cast(buf, ctypes.c_void_p), len(ops.FLOP_CLASSES), 1) == 0      # read + reset
    assert all(v == 0.0 for v in ops.flop_counters().values())
    assert lib.dvd_flop_counters(None, 3, 0) != 0                                                        # null output: an error code
    before = ops.executed_flops()
    ops.note_replay({'xconv_wide': 2.0e9, 'xwgrad3': 1.0e9})
    ops.note_replay(None)                                                                                # an uncounted graph
    ops.note_replay({'xconv_wide': 2.0e9})
    after = ops.executed_flops()
    d = {k: after[k] - before[k] for k in after}
    assert d['xconv_wide'] == 4.0e9 and d['xwgrad3'] == 1.0e9 and sum(d.values()) == 5.0e9
    assert ops.flops_since(ops.flop_counters()) == {k: 0.0 for k in ops.ALL_CLASSES}


def test_recomputed_work_is_counted_apart():
    """ops.counting_recomputed(): the work launched or replayed inside the context is executed AND recorded as a re-computation
    (bench.py: roofline_mfma.needed_TFLOP_per_step = executed - recomputed; SURVEY 8d does not count recompute); nested and
    repeated regions add up, work outside is untouched."""
    r0, e0 = dict(ops.RECOMPUTED), ops.executed_flops()
    ops.note_replay({'xconv_wide': 3.0e9})                               # needed work
    with ops.counting_recomputed():
        ops.note_replay({'xconv_wide': 1.0e9, 'mlp_fwd': 2.0e9})         # a chunk's forward / a stash-free MLP pass, again
    with ops.counting_recomputed():
        ops.note_replay({'mlp_fwd': 0.5e9})
    e1 = ops.executed_flops()
    rec = {k: ops.RECOMPUTED[k] - r0[k] for k in r0}
    exe = {k: e1[k] - e0[k] for k in e1}
    assert rec['xconv_wide'] == 1.0e9 and rec['mlp_fwd'] == 2.5e9 and sum(rec.values()) == 3.5e9
    assert exe['xconv_wide'] == 4.0e9 and exe['mlp_fwd'] == 2.5e9
    assert sum(exe.values()) - sum(rec.values()) == 3.0e9


def test_head_room_knob(monkeypatch):
    from dvd_hip.models.scene_flow_motion_field import head_room_fraction, keep_slot_fits
    monkeypatch.delenv('DVD_HEAD_ROOM_GB', raising=False)
    assert head_room_fraction(1) == 0.08 and head_room_fraction(8) == 0.10
    monkeypatch.setenv('DVD_HEAD_ROOM_GB', '48')
    assert abs(head_room_fraction(1) - 48.0 / 288.0) < 1e-12 and head_room_fraction(8) == head_room_fraction(1)
    monkeypatch.setenv('DVD_HEAD_ROOM_GB', '10')                 # never below the default
    assert head_room_fraction(8) == 0.10
    gb = 2 ** 30
    # a 60 GB slot next to 130 GB of stashes on a 288 GB device with 225 GB free: fits at 8 %, not with 48 GB of head room
    assert keep_slot_fits(60 * gb, 225 * gb, 288 * gb, 130 * gb, 0, 0, 150 * gb, 0.08)
    assert not keep_slot_fits(60 * gb, 225 * gb, 288 * gb, 130 * gb, 0, 0, 150 * gb, 48.0 / 288.0)


def test_profiles_joined_by_the_bench_line_carry_a_source_digest():
    """bench.py joins three committed profiles into its line (HBM traffic and SQ counters of the warp+loss sequence, per-class
    kernel times); each is stamped by the tool that wrote it with dvd_hip.build.source_digest of the sources it was collected
    from, and the line says `profile_is_of_this_binary`.  Here: the stamp exists, has the digest's form, and the warp+loss
    profiles -- the kernel `north_star` puts a number on -- ARE of the committed sources (a kernel edit without a re-collected
    profile fails here, not silently on the bench line)."""
    import json
    import os
    import re
    from dvd_hip import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    warp = build.source_digest(build.WARP_UNITS)
    assert re.fullmatch(r'[0-9a-f]{16}', warp) and warp != build.source_digest(None)
    for name in ('warp_loss_pmc.json', 'warp_loss_sq.json'):
        rec = json.load(open(os.path.join(root, 'profiles', name)))
        assert rec.get('source_digest') == warp, '%s was collected from other warp+loss sources (%s, now %s): re-collect it ' \
            '(tools/gpu_visit.sh stages pmc sq)' % (name, rec.get('source_digest'), warp)
    rec = json.load(open(os.path.join(root, 'profiles', 'mfma_roofline.json')))
    for mode in ('fp32', 'fp16'):
        assert re.fullmatch(r'[0-9a-f]{16}', rec[mode]['source_digest']) and rec[mode]['helpers']
