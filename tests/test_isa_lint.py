"""Static checks of the compiled gfx950 ISA (CPU only: hipcc cross-compiles).  The LDS-DMA apply kernel reads its fragments by
inline assembly with hand-counted `s_waitcnt lgkmcnt` (scfgp_amd/csrc/apply.hip); nothing may touch the destination of such a
read before a wait has covered it, and its k loops must stay free of scratch traffic."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
pytestmark = pytest.mark.skipif(not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')


def test_inflight_checker_catches_a_planted_hazard():
    import isa_inflight
    ok = ['ds_read_b64 v[2:3], v0 offset:0', 'ds_read_b64 v[4:5], v0 offset:8', 's_waitcnt lgkmcnt(1)',
          'v_mfma_f32_16x16x4_f32 v[8:11], v2, v3, v[8:11]', 's_waitcnt lgkmcnt(0)', 'v_mov_b32_e32 v6, v4']
    assert isa_inflight.check(list(enumerate(ok)), 'ok') == []
    bad = ['ds_read_b64 v[2:3], v0 offset:0', 'ds_read_b64 v[4:5], v0 offset:8', 's_waitcnt lgkmcnt(1)', 'v_mov_b32_e32 v6, v4']
    hits = isa_inflight.check(list(enumerate(bad)), 'bad')
    assert len(hits) == 1 and hits[0][2] == [4]
    overwrite = ['ds_read_b64 v[2:3], v0 offset:0', 'ds_read_b64 v[2:3], v1 offset:0']
    assert len(isa_inflight.check(list(enumerate(overwrite)), 'overwrite')) == 1


def test_apply_dma_kernels_never_touch_a_fragment_in_flight_and_keep_their_loops_out_of_scratch():
    import isa_inflight
    import isa_loops
    assert isa_inflight.main('apply', 'apply_dma_kernel', []) == 0
    loops = [(n, b) for n, b in isa_loops.census('apply', 'apply_dma_kernel', []) if 'mfma' in b]
    assert len(loops) >= 12 * 2                                  # 12 instantiations: the steady loop and the tail loop of each
    for name, body in loops:
        assert body['scratch'] == 0 and body['barrier'] == 1 and body['ds_read'] in (12, 16), (name, body)


def test_gram_kernels_never_touch_a_fragment_in_flight():
    """the pipelined Gram tiles (gram_pipe_dma: 256 x 128 and 64 x 512) inside the persistent gram_kernel: the flush of a chunk must find the
    pipeline drained (no spill or copy of a fragment register whose read has not been waited for)"""
    import isa_inflight
    assert isa_inflight.main('gram', 'gram_kernel', []) == 0
