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


def test_ring_checker_catches_a_planted_count():
    """a three-slot ring's counted wait must be the fetch count of a trip around the loop: round 4's vmcnt(3) over a two-instruction share"""
    import isa_inflight
    loop = lambda n: ['.LBB0_1:', 's_waitcnt vmcnt(%d)' % n, 's_barrier', 'global_load_lds_dwordx4 v[0:1], off',
                      'global_load_lds_dwordx4 v[2:3], off', 'v_mfma_f32_16x16x4_f32 v[8:11], v2, v3, v[8:11]', 's_cbranch_scc1 .LBB0_1']
    assert isa_inflight.ring_check(loop(2)) == ([], 1)
    bad, n = isa_inflight.ring_check(loop(3))
    assert n == 1 and bad == [(0, 6, [2], [3])]
    # a rotated loop: the wave that also fetches the row weights takes a second back edge; its count is 3, the others' 2
    rot = ['.LBB0_1:', 's_cbranch_vccz .LBB0_2', 's_waitcnt vmcnt(2)', '.LBB0_2:', 's_cbranch_vccnz .LBB0_3', 's_waitcnt vmcnt(3)',
           '.LBB0_3:', 's_barrier', 'global_load_lds_dwordx4 v[0:1], off', 'global_load_lds_dwordx4 v[2:3], off',
           'v_mfma_f32_16x16x4_f32 v[8:11], v2, v3, v[8:11]', 's_cbranch_vccnz .LBB0_1', 'global_load_lds_dword v4, off', 's_branch .LBB0_1']
    assert isa_inflight.ring_check(rot) == ([], 1)


# The checks below read the gfx950 code objects INSIDE scfgp_amd/lib/libscfgp_hip.so -- the library that ships and that the GPU
# tests load -- not a recompilation with flags of their own, and assert how much they inspected.
def test_apply_dma_kernels_never_touch_a_fragment_in_flight_and_keep_their_loops_out_of_scratch():
    import isa_inflight
    import isa_loops
    import isa_source
    r = isa_inflight.run(None, 'apply_dma_kernel', shipped=isa_source.SHIPPED)
    assert r['functions'] == 20 and r['inflight'] == 0 and r['ring'] == 0 and r['ring_loops'] == 20, r       # 8 fp64 + 12 fp32 instantiations
    loops = [(n, b) for n, b in isa_loops.census(None, 'apply_dma_kernel', shipped=isa_source.SHIPPED) if 'mfma' in b]
    assert len(loops) >= 20 * 2                                  # the steady loop and the tail loop of each
    for name, body in loops:
        assert body['scratch'] == 0 and body['barrier'] == 1 and body['ds_read'] in (12, 16), (name, body)


def test_gram_kernels_never_touch_a_fragment_in_flight():
    """the pipelined Gram tiles (gram_pipe_dma: 256 x 128 tall, 64 x 512 wide, two 128 x 128 on one image, 128 x 128) and the fp64
    128 x 128 tiles inside the persistent gram_kernel: the flush of a chunk must find the pipeline drained (no spill or copy of a
    fragment register whose read has not been waited for), and every counted vmcnt of the three-slot rings is its loop's fetch
    count.  Asserts that all four instantiations and their fetch loops were in fact inspected."""
    import isa_inflight
    import isa_loops
    import isa_source
    r = isa_inflight.run(None, 'gram_kernel', shipped=isa_source.SHIPPED)
    assert r['functions'] == 4 and r['inflight'] == 0 and r['ring'] == 0, r
    assert r['ring_loops'] >= 2 * 4 + 2 * 12 and r['ds_reads'] > 400, r       # fp64: 128 x 128 +- diagonal; fp32: four DMA tile kinds, +- weights / side sums
    per_kernel = {}
    for name, b in isa_loops.census(None, 'gram_kernel', shipped=isa_source.SHIPPED):
        if 'mfma' in b and b['global_load'] > 0 and b['barrier'] == 1:
            per_kernel.setdefault(name, []).append(b)
    f32 = [v for k, v in per_kernel.items() if 'TileCfg<float' in k]
    assert len(f32) == 2 and all(len(v) >= 4 for v in f32)      # a steady single-barrier loop per pipelined tile kind
    # the steady loops of the pipelined fp32 tiles: 8 MFMA tiles x 2 halves x TM ... per trip, no scratch in the tall / wide / pair bodies
    assert sum(b['scratch'] == 0 for v in f32 for b in v) >= 2 * 4


def test_f16x3_kernels_never_touch_a_fragment_in_flight_and_keep_their_loops_out_of_scratch():
    """apply_f16_kernel (6 instantiations: two-slot ring, 16 hand-issued ds_read_b128 and 48 matrix instructions per stage) and
    gram_f16_kernel (three-slot ring whose counted vmcnt is the wave's three fetches, 24 transposing reads, 24 matrix instructions):
    the fragment reads are inline assembly released by counted lgkmcnt waits, so the same replay applies."""
    import isa_inflight
    import isa_loops
    import isa_source
    r = isa_inflight.run(None, 'f16_kernel', shipped=isa_source.SHIPPED)
    assert r['functions'] == 7 and r['inflight'] == 0 and r['ring'] == 0 and r['ring_loops'] >= 7 and r['ds_reads'] >= 6 * 16 + 24, r
    loops = [(n, b) for n, b in isa_loops.census(None, 'f16_kernel', shipped=isa_source.SHIPPED) if 'mfma' in b]
    assert len(loops) >= 7
    for name, body in loops:
        gram = name.startswith('gram')
        # (the Gram's loop has two back edges -- waves without stored outputs skip the multiplication -- so a layout range may end before the barrier)
        assert body['scratch'] == 0 and body['barrier'] == (1 if not gram else body['barrier'] & 1), (name, body)
        assert (body['mfma'], body['ds_read']) == ((24, 24) if gram else (48, 16)), (name, body)
