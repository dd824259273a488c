"""TensorFlow checkpoint ingestion (kodak/eae/graph/tf_checkpoint.py), CPU only.

The reference restores `eae/results/<suffix>/model_<i>.ckpt` through `tf.train.Saver` (EntropyAutoencoder.py:454-458,
IsolatedDecoder.py:123-124). Pinned here: the names / dtypes / shapes / saver version the reference's own
`.ckpt.meta` files declare (tests/golden/ckpt_variables.json, made by oracle/gen_ckpt_variables.py), the CRC-32C and
Snappy known answers, the on-disk framing constants, and writer -> reader round trips of both layouts.
"""
import json
import os
import struct

import numpy
import pytest

from autoencoder_based_image_compression_amd.kodak.eae.graph import tf_checkpoint as ckpt
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ckpt_variables.json')


def reference_like_variables(are_bin_widths_learned, seed):
    """Everything a reference checkpoint holds (model + optimiser slots + schedule), with the golden shapes."""
    rng = numpy.random.RandomState(seed)
    variables = var.random_variables(1., are_bin_widths_learned, seed=seed, bias_std=0.1)
    for name in list(variables):
        if name == var.BIN_WIDTHS_NAME:
            continue                                   # SGD-trained in the reference: no Adam slots in its graphs
        variables[name + '/Adam'] = rng.standard_normal(variables[name].shape).astype(numpy.float32)
        variables[name + '/Adam_1'] = rng.standard_normal(variables[name].shape).astype(numpy.float32)
    variables['beta1_power'] = numpy.array(0.9**5, dtype=numpy.float32)
    variables['beta2_power'] = numpy.array(0.999**5, dtype=numpy.float32)
    variables['decaying_lr/global_step'] = numpy.array(400000, dtype=numpy.int32)
    variables['piecewise_linear_function/grid'] = numpy.linspace(-10.5, 10.5, 671).astype(numpy.float32)
    variables['piecewise_linear_function/parameters'] = rng.uniform(size=(128, 671)).astype(numpy.float32)
    variables['piecewise_linear_function/nb_intervals_per_side'] = numpy.array(10, dtype=numpy.int64)
    return variables


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors, also the ones in TensorFlow's and LevelDB's crc32c tests
    assert ckpt.crc32c(b'123456789') == 0xe3069283
    assert ckpt.crc32c(bytes(32)) == 0x8a9136aa
    assert ckpt.crc32c(bytes([0xff]*32)) == 0x62a8ab43
    assert ckpt.crc32c(bytes(range(32))) == 0x46dd794e
    assert ckpt.crc32c(bytes(range(31, -1, -1))) == 0x113fdb5c
    data = numpy.random.RandomState(0).randint(0, 256, size=100003, dtype=numpy.uint8).tobytes()
    for cut in (0, 1, 7, 8, 9, 50000, len(data)):
        assert ckpt.crc32c(data[cut:], ckpt.crc32c(data[:cut])) == ckpt.crc32c(data)      # Extend()
    # bitwise definition (reflected polynomial 0x82F63B78) on unaligned slices
    def bitwise(buffer):
        crc = 0xffffffff
        for byte in buffer:
            crc ^= byte
            for _ in range(8):
                crc = (crc >> 1) ^ (0x82f63b78 if crc & 1 else 0)
        return crc ^ 0xffffffff
    for (start, stop) in ((0, 0), (1, 2), (3, 20), (5, 77)):
        assert ckpt.crc32c(data[start:stop]) == bitwise(data[start:stop])
    for crc in (0, 1, 0xe3069283, 0xffffffff):
        assert ckpt.unmask_crc(ckpt.mask_crc(crc)) == crc
        assert ckpt.mask_crc(crc) != crc


def test_snappy_known_answers():
    # hand-assembled streams following the Snappy format description: varint length, then literal / copy elements
    assert ckpt.snappy_uncompress(b'\x00') == b''
    assert ckpt.snappy_uncompress(bytes([5, 4 << 2]) + b'hello') == b'hello'
    # "abcabcabcabc": literal "abc" then a 1-byte-offset copy of 9 bytes from 3 back (overlapping: run-length behaviour)
    assert ckpt.snappy_uncompress(bytes([12, 2 << 2]) + b'abc' + bytes([((9 - 4) << 2) | 1, 3])) == b'abc'*4
    # 2-byte-offset copy; literal with an explicit one-byte length (tag 60 << 2)
    literal = bytes(range(100))
    stream = bytes([110, 60 << 2, 99]) + literal + bytes([((10 - 1) << 2) | 2, 100, 0])
    assert ckpt.snappy_uncompress(stream) == literal + literal[:10]
    with pytest.raises(ckpt.CheckpointError):
        ckpt.snappy_uncompress(bytes([4, 0 << 2]) + b'a' + bytes([(4 - 4) << 2 | 1, 9]))     # copy before the start
    with pytest.raises(ckpt.CheckpointError):
        ckpt.snappy_uncompress(bytes([9, 4 << 2]) + b'hello')                                  # announced length differs


def test_table_framing_and_prefix_compression(tmp_path):
    path = str(tmp_path/'table')
    pairs = [(b'', b'header')] + [('encoder/weights_{:03d}'.format(i).encode(), bytes([i % 251])*(i % 40)) for i in range(300)]
    ckpt.write_table(path, pairs)
    assert ckpt.read_table(path) == pairs
    with open(path, 'rb') as file:
        data = file.read()
    assert struct.unpack('<Q', data[-8:])[0] == 0xdb4775248b80fb57
    assert len(data) > ckpt.FOOTER_LENGTH
    # the shared-prefix encoding is in use: the common stem is not repeated 300 times
    assert data.count(b'encoder/weights_') < 300//ckpt.RESTART_INTERVAL + 3
    # a flipped payload byte is caught by the block checksum
    corrupted = bytearray(data)
    corrupted[40] ^= 0x10
    with open(path, 'wb') as file:
        file.write(corrupted)
    with pytest.raises(ckpt.CheckpointError):
        ckpt.read_table(path)
    assert len(ckpt.read_table(path, verify_checksums=False)) == len(pairs)
    with open(path, 'wb') as file:
        file.write(data[:-1] + b'\x00')
    with pytest.raises(ckpt.CheckpointError):
        ckpt.read_table(path)
    with pytest.raises(ValueError):
        ckpt.write_table(path, [(b'b', b''), (b'a', b'')])


def test_table_with_many_blocks(tmp_path, monkeypatch):
    monkeypatch.setattr(ckpt, 'BLOCK_SIZE', 512)
    path = str(tmp_path/'table')
    pairs = [('k{:05d}'.format(i).encode(), os.urandom(37)) for i in range(400)]
    ckpt.write_table(path, pairs)
    assert ckpt.read_table(path) == pairs


def test_snappy_compressed_block_is_read(tmp_path):
    # one data block stored with compression type 1 whose Snappy body is a single literal
    builder = ckpt._BlockBuilder()
    builder.add(b'', b'h')
    builder.add(b'name', b'value')
    block = builder.finish()
    body = ckpt.write_varint(len(block)) + bytes([60 << 2, len(block) - 1]) + block
    out = bytearray()

    def emit(contents, kind):
        handle = ckpt.write_varint(len(out)) + ckpt.write_varint(len(contents))
        out.extend(contents + bytes([kind]) + struct.pack('<I', ckpt.mask_crc(ckpt.crc32c(contents + bytes([kind])))))
        return handle
    data_handle = emit(body, ckpt.SNAPPY_COMPRESSION)
    index = ckpt._BlockBuilder()
    index.add(b'name', data_handle)
    meta_handle = emit(ckpt._BlockBuilder().finish(), ckpt.NO_COMPRESSION)
    index_handle = emit(index.finish(), ckpt.NO_COMPRESSION)
    footer = meta_handle + index_handle
    out.extend(footer + bytes(40 - len(footer)) + struct.pack('<Q', ckpt.TABLE_MAGIC))
    path = str(tmp_path/'table')
    with open(path, 'wb') as file:
        file.write(out)
    assert ckpt.read_table(path) == [(b'', b'h'), (b'name', b'value')]


def test_golden_names_match_the_model_surface():
    """The names, dtypes and shapes this package restores are the ones in the reference's `.ckpt.meta` graphs."""
    with open(GOLDEN) as file:
        golden = json.load(file)
    assert len(golden) == 9
    for (key, value) in golden.items():
        learned = key.startswith('learning_bw')
        assert value['saver_def'] == {'version': 1}              # SaverDef.V1, not sharded: one `model_<i>.ckpt` file
        declared = value['variables']
        for name in var.model_names(learned):
            assert declared[name]['dtype_enum'] == 1, name        # DT_FLOAT
            assert tuple(declared[name]['shape']) == var.SHAPES[name], name
        for name in var.ENCODER_NAMES_FIXED_BW + var.DECODER_NAMES_FIXED_BW:
            assert (name in declared) == (not learned)
        assert 'decoder/biases_6' not in declared
        assert declared['decaying_lr/global_step'] == {'dtype_enum': 3, 'shape': []}
        assert declared['piecewise_linear_function/nb_intervals_per_side'] == {'dtype_enum': 9, 'shape': []}
        assert all(v['dtype_enum'] in ckpt.DTYPES for v in declared.values())


@pytest.mark.parametrize('layout', ['v1', 'v2'])
@pytest.mark.parametrize('learned', [False, True])
def test_round_trip_of_a_reference_like_checkpoint(tmp_path, layout, learned):
    variables = reference_like_variables(learned, seed=11)
    with open(GOLDEN) as file:
        golden = json.load(file)['learning_bw_0dot5_12000/model_10.ckpt.meta' if learned else '1_24000/model_10.ckpt.meta']
    assert set(variables) == set(golden['variables'])             # the synthetic checkpoint has the reference's inventory
    prefix = str(tmp_path/'model_10.ckpt')
    (ckpt.save_checkpoint_v1 if layout == 'v1' else ckpt.save_checkpoint)(prefix, variables)
    assert ckpt.checkpoint_format(prefix) == (1 if layout == 'v1' else 2)
    assert sorted(os.listdir(str(tmp_path))) == (['model_10.ckpt'] if layout == 'v1' else
                                                ['model_10.ckpt.data-00000-of-00001', 'model_10.ckpt.index'])
    listed = ckpt.list_variables(prefix)
    assert set(listed) == set(variables)
    for (name, array) in variables.items():
        assert listed[name] == (array.dtype, array.shape), name
    everything = ckpt.load_checkpoint(prefix)
    assert set(everything) == set(variables)
    for (name, array) in variables.items():
        assert everything[name].dtype == array.dtype and everything[name].shape == array.shape, name
        assert numpy.array_equal(everything[name], array), name
    # the model surface: exactly the inference variables, float32, C-contiguous
    for side in ('encoder', 'decoder', 'both'):
        restored = var.restore_variables(prefix, learned, side)
        assert list(restored) == var.model_names(learned, side)
        for (name, array) in restored.items():
            assert array.flags['C_CONTIGUOUS'] and numpy.array_equal(array, variables[name])
    if learned:
        with pytest.raises(KeyError):
            var.restore_variables(prefix, False)                   # a fixed-bin-width graph needs gamma_3 / gamma_4
    with pytest.raises(KeyError):
        ckpt.load_checkpoint(prefix, names=['decoder/biases_6'])


def test_v1_key_encoding():
    # ordered-code layout of EncodeTensorNameSlice: num(0) | escaped name | 00 01 | num(rank) | (start, length) per dim
    assert ckpt.encode_tensor_name_slice('ab', 0) == b'\x00ab\x00\x01\x00'
    assert ckpt.encode_tensor_name_slice('ab', 2) == b'\x00ab\x00\x01\x01\x02' + b'\x80\x7f'*2
    assert ckpt.encode_tensor_name_slice('a\x00b', 1) == b'\x00a\x00\xffb\x00\x01\x01\x01\x80\x7f'
    names = ['encoder/weights_1', 'encoder/weights_1/Adam', 'encoder/weights_1/Adam_1', 'encoder/weights_2', 'beta1_power']
    keys = [ckpt.encode_tensor_name_slice(name, 4) for name in names]
    assert [names[i] for i in numpy.argsort(numpy.array(keys, dtype=object))] == sorted(names)


def test_v1_partial_slices_and_raw_content(tmp_path):
    """A variable stored as two row slices, one through `tensor_content`, is reassembled."""
    full = numpy.arange(24, dtype=numpy.float32).reshape(6, 4)
    f = ckpt._field
    ld = ckpt._length_delimited

    def extent(start=None, length=None):
        body = b''
        if start:
            body += f(1, 0, ckpt.write_varint(start))
        if length is not None:
            body += f(2, 0, ckpt.write_varint(length))
        return f(1, 2, ld(body))
    shape = ckpt.encode_tensor_shape(full.shape)
    meta = f(1, 2, ld(f(1, 2, ld(b'w')) + f(2, 2, ld(shape)) + f(3, 0, ckpt.write_varint(1))))
    header = f(1, 2, ld(meta))

    def saved(rows, payload_field):
        part = full[rows]
        tensor = f(1, 0, ckpt.write_varint(1)) + f(2, 2, ld(ckpt.encode_tensor_shape(part.shape)))
        tensor += f(payload_field, 2, ld(part.tobytes()))
        piece = extent(rows.start, rows.stop - rows.start) + extent()
        return f(2, 2, ld(f(1, 2, ld(b'w')) + f(2, 2, ld(piece)) + f(3, 2, ld(tensor))))
    prefix = str(tmp_path/'model.ckpt')
    ckpt.write_table(prefix, [(b'', header), (b'\x00w\x00\x01a', saved(slice(0, 2), 5)), (b'\x00w\x00\x01b', saved(slice(2, 6), 4))])
    loaded = ckpt.load_checkpoint(prefix)
    assert numpy.array_equal(loaded['w'], full)


def test_corruption_and_missing_files(tmp_path):
    variables = {'encoder/biases_1': numpy.arange(128, dtype=numpy.float32)}
    prefix = str(tmp_path/'model_3.ckpt')
    ckpt.save_checkpoint(prefix, variables)
    data_path = ckpt.data_filename(prefix, 0, 1)
    with open(data_path, 'r+b') as file:
        file.seek(17)
        file.write(b'\x55')
    with pytest.raises(ckpt.CheckpointError):
        ckpt.load_checkpoint(prefix)
    assert ckpt.load_checkpoint(prefix, verify_checksums=False)['encoder/biases_1'].shape == (128,)
    with open(data_path, 'r+b') as file:
        file.truncate(100)
    with pytest.raises(ckpt.CheckpointError):
        ckpt.load_checkpoint(prefix)
    os.remove(data_path)
    with pytest.raises(IOError):
        ckpt.load_checkpoint(prefix)
    assert not ckpt.exists(str(tmp_path/'absent.ckpt'))
    with pytest.raises(IOError):
        var.restore_variables(str(tmp_path/'absent.ckpt'), False)
    # an `.npz` is not mistaken for a V1 table
    npz = str(tmp_path/'model_4.npz')
    var.save_variables(npz, var.random_variables(1., False, seed=1))
    assert ckpt.checkpoint_format(npz) == 0
    assert set(var.restore_variables(str(tmp_path/'model_4.ckpt'), False)) == set(var.model_names(False))
    assert set(var.restore_variables(npz, False, 'decoder')) == set(var.model_names(False, 'decoder'))


def test_latest_checkpoint(tmp_path):
    directory = str(tmp_path)
    assert ckpt.latest_checkpoint(directory) is None
    ckpt.save_checkpoint_v1(os.path.join(directory, 'model_10.ckpt'), {'a': numpy.zeros(3, dtype=numpy.float32)})
    ckpt.write_checkpoint_state(directory, 'model_10.ckpt')
    with open(os.path.join(directory, 'checkpoint')) as file:      # same text as eae/results/*/checkpoint in the reference
        assert file.read() == 'model_checkpoint_path: "model_10.ckpt"\nall_model_checkpoint_paths: "model_10.ckpt"\n'
    assert ckpt.latest_checkpoint(directory) == os.path.join(directory, 'model_10.ckpt')
