"""Pins the CPU oracle of the lossless coder (oracle/coder_oracle.c) against
(1) golden vectors produced by the REAL reference coder (tests/golden/coder_golden.npz, made by oracle/gen_golden.py
    from kodak_tensorflow/lossless/c++/source through oracle/ref_shim.cpp), including the reference's own known-answer
    cases (tests.cpp:69-376, test_lossless.py:96-101), and
(2) when the reference build oracle/_ref is present, the reference itself on random inputs."""
import os

import numpy
import pytest

from oracle import coder as oc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'coder_golden.npz')


@pytest.fixture(scope='module')
def gold():
    with numpy.load(GOLD) as data:
        return {k: data[k] for k in data.files}


@pytest.fixture(scope='module')
def orc():
    return oc.CoderLib('oracle')


def test_known_answer_compress_lossless(gold, orc):
    """tests.cpp:354-376: 104 bits = 51 BAC + 53 bypass; BAC e6ffffff1ffe05, bypass fef3f6c67f0d02."""
    (rec, nb, st) = orc.compress_lossless(gold['ka_compress_in'], gold['ka_compress_p'], want_streams=True)
    assert nb == 104 == int(gold['ka_compress_bits'])
    assert numpy.array_equal(rec, gold['ka_compress_in'])
    assert (st['bac_bits'], st['bypass_bits']) == (51, 53)
    assert st['bac_bytes'].tobytes().hex() == 'e6ffffff1ffe05' == gold['ka_compress_bac'].tobytes().hex()
    assert st['bypass_bytes'].tobytes().hex() == 'fef3f6c67f0d02' == gold['ka_compress_byp'].tobytes().hex()


def test_known_answer_bac_20_bits(gold, orc):
    """tests.cpp:69-132: 20 bits with 20 probabilities -> 31 bits, round trip."""
    c = orc.coder(72, gold['ka_bac_p'])
    for (b, p) in zip(gold['ka_bac_bits_in'], gold['ka_bac_p']):
        c.bac_encoding(int(b), float(p))
    c.stop_bac_encoding()
    assert c.written_bac() == 31 == int(gold['ka_bac_nbits'])
    assert numpy.array_equal(c.bytes_bac(), gold['ka_bac_stream'])
    c.start_bac_decoding()
    assert [c.bac_decoding(float(p)) for p in gold['ka_bac_p']] == list(gold['ka_bac_bits_in'])


def test_known_answer_signed_ueg0(gold, orc):
    """tests.cpp:280-352: 64 BAC bits + 49 bypass bits."""
    c = orc.coder(200, numpy.full(8, 0.5))
    for v in gold['ka_sueg0_in']:
        c.write_signed_ueg0(int(v))
    c.stop_bac_encoding()
    assert (c.written_bac(), c.written_bypass()) == (64, 49)
    assert numpy.array_equal(c.bytes_bac(), gold['ka_sueg0_bac']) and numpy.array_equal(c.bytes_bypass(), gold['ka_sueg0_byp'])
    c.start_bac_decoding()
    assert [c.read_signed_ueg0() for _ in gold['ka_sueg0_in']] == list(gold['ka_sueg0_in'])


def test_known_answer_eg0_and_truncated_unary(gold, orc):
    """tests.cpp:164-209 (79 bypass bits) and :211-278 (40 BAC bits, decodes 0 1 2 8 8 8 8)."""
    c = orc.coder(231, numpy.full(8, 0.1))
    for v in gold['ka_eg0_in']:
        c.write_eg0(int(v))
    assert c.written_bypass() == 79 and numpy.array_equal(c.bytes_bypass(), gold['ka_eg0_byp'])
    assert [c.read_eg0() for _ in gold['ka_eg0_in']] == list(gold['ka_eg0_in'])
    c = orc.coder(56, numpy.full(8, 0.5))
    for v in gold['ka_eg0_in']:
        c.write_truncated_unary(int(v))
    c.stop_bac_encoding()
    assert c.written_bac() == 40 and numpy.array_equal(c.bytes_bac(), gold['ka_tu_bac'])
    c.start_bac_decoding()
    assert [c.read_truncated_unary() for _ in gold['ka_eg0_in']] == [0, 1, 2, 8, 8, 8, 8] == list(gold['ka_tu_decoded'])


def test_known_answer_flattened_map(gold, orc):
    """test_lossless.py:96-101: the reference's only real assert; 20 bits."""
    (rec, nb) = orc.compress_lossless(gold['ka_flat_in'], numpy.array([0.5, 0.5, 0.5]))
    assert nb == 20 == int(gold['ka_flat_bits']) and numpy.array_equal(rec, gold['ka_flat_in'])


def test_count_nb_bits_table(gold, orc):
    table = numpy.array([orc.count_nb_bits(i) for i in range(65537)], dtype=numpy.uint8)
    assert numpy.array_equal(table, gold['nb_bits_0_65536'])


def test_golden_streams(gold, orc):
    for i in range(int(gold['nb_cases'])):
        x = gold['case{}_in'.format(i)]
        p = gold['case{}_p'.format(i)]
        (rec, nb, st) = orc.compress_lossless(x, p, want_streams=True)
        assert numpy.array_equal(rec, x), i
        assert st['bac_bits'] == int(gold['case{}_bac_bits'.format(i)]) and st['bypass_bits'] == int(gold['case{}_byp_bits'.format(i)]), i
        assert numpy.array_equal(st['bac_bytes'], gold['case{}_bac'.format(i)]), i
        assert numpy.array_equal(st['bypass_bytes'], gold['case{}_byp'.format(i)]), i
        assert nb == st['bac_bits'] + st['bypass_bits']


def test_golden_errors(gold, orc):
    for i in range(int(gold['nb_err_cases'])):
        expected = str(gold['err_messages'][i])
        try:
            (rec, nb) = orc.compress_lossless(gold['err{}_in'.format(i)], gold['err{}_p'.format(i)])
            got = 'ok:{}'.format(nb)
        except Exception as exc:
            got = '{0}:{1}'.format(type(exc).__name__, exc)
        assert got == expected, i


def test_truncated_unary_length_zero(orc):
    """m_probabilities.at(0) throws std::out_of_range when L == 0 (LosslessCoder.cpp:173,189) -> IndexError."""
    c = orc.coder(64, numpy.zeros(0))
    with pytest.raises(IndexError):
        c.write_truncated_unary(0)


@pytest.mark.skipif(not oc.available('ref'), reason='oracle/_ref not built (needs /root/reference)')
def test_oracle_equals_reference_on_random_maps(orc):
    ref = oc.CoderLib('ref')
    rng = numpy.random.RandomState(123)
    for t in range(200):
        n = int(rng.randint(1, 300))
        L = int(rng.randint(1, 41))
        scale = rng.choice([0.3, 1, 3, 10, 100, 3000])
        x = numpy.clip(numpy.round(rng.laplace(size=n)*scale), -32767, 32767).astype(numpy.int16)
        p = numpy.clip(rng.rand(L), 0.01, 0.99)
        outcomes = []
        for lib in (orc, ref):
            try:
                (rec, nb) = lib.compress_lossless(x, p)
                outcomes.append((nb, rec.tobytes()))
            except RuntimeError as exc:
                outcomes.append(str(exc))
        assert outcomes[0] == outcomes[1]
        if not isinstance(outcomes[0], str):
            c = ref.coder(n*max(32, L), p)
            for v in x:
                c.write_signed_ueg0(int(v))
            c.stop_bac_encoding()
            st = orc.compress_lossless(x, p, want_streams=True)[2]
            assert numpy.array_equal(c.bytes_bac(), st['bac_bytes']) and numpy.array_equal(c.bytes_bypass(), st['bypass_bytes'])
