"""csrc/isa_guard.py: the build-time check that no 16-byte MUBUF store with a register soffset sits next to a vector write of
its data registers (the pattern that stored stale lanes on MI355X in round 2). CPU only: it reads the
cross-compiled library."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'autoencoder_based_image_compression_amd', 'csrc'))

# what the reverted build (commit 7851e75) looked like around the faulty store, and three harmless neighbours
BAD = """
_Z6kernelv:
	v_pk_mul_f32 v[18:19], v[24:25], v[18:19]
	v_pk_mul_f32 v[20:21], v[24:25], v[20:21]
	buffer_store_dwordx4 v[18:21], v1, s[4:7], s0 offen
	s_endpgm
"""
AFTER = """
_Z6kernelv:
	buffer_store_dwordx4 v[18:21], v1, s[4:7], s0 offen
	s_nop 0
	v_mov_b32_e32 v19, 0
	s_endpgm
"""
FINE = """
_Z6kernelv:
	v_pk_mul_f32 v[20:21], v[24:25], v[20:21]
	buffer_store_dwordx4 v[18:21], v1, s[4:7], 0 offen offset:16
	v_pk_mul_f32 v[30:31], v[24:25], v[30:31]
	s_nop 2
	buffer_store_dwordx4 v[28:31], v1, s[4:7], s0 offen
	v_add_f32_e32 v40, v41, v42
	buffer_store_dwordx4 v[50:53], v1, s[4:7], s0 offen
	v_cmp_lt_f32_e32 vcc, v50, v51
	buffer_store_dwordx2 v[60:61], v1, s[4:7], s0 offen
	v_mov_b32_e32 v60, 0
_Z7kernel2v:
	v_mov_b32_e32 v50, 0
	s_endpgm
"""


def test_the_pattern_is_found_in_both_directions_and_nowhere_else():
    import isa_guard
    assert len(isa_guard.scan(BAD, 'bad')) == 1 and 'v20' in isa_guard.scan(BAD, 'bad')[0]
    assert len(isa_guard.scan(AFTER, 'after')) == 1 and 'ahead of' in isa_guard.scan(AFTER, 'after')[0]
    assert isa_guard.scan(FINE, 'fine') == []


def test_the_shipped_library_is_clean():
    import isa_guard
    lib = os.path.join(ROOT, 'autoencoder_based_image_compression_amd', 'lib', 'libeae_hip.so')
    if not os.path.isfile(lib):
        pytest.skip('libeae_hip.so not built')
    blobs = isa_guard.code_objects(lib)
    assert len(blobs) >= 10                       # one code object per kernel file
    assert isa_guard.check([lib]) == []
    # the scan sees the stores it is about: the cut tiles' parked accumulators ARE 16-byte stores with a register soffset
    text = ''.join(isa_guard.disassemble(b) for b in blobs)
    assert 'buffer_store_dwordx4' in text
