"""TensorFlow checkpoints read (and written) without TensorFlow: the V1 format and the V2 "tensor bundle".

The reference restores its trained models with `tf.train.Saver().restore(sess, 'eae/results/<suffix>/model_<i>.ckpt')`
(kodak_tensorflow/eae/graph/EntropyAutoencoder.py:390, :454-458; IsolatedDecoder.py:107, :123-124) and writes them with
`Saver.save` (EntropyAutoencoder.py:465-482). The `SaverDef` inside every `model_*.ckpt.meta` of the reference says
`version: V1` (tests/golden/ckpt_variables.json), so the authors' checkpoints are V1 files:

  V1   model_<i>.ckpt                       ONE sorted string table (the LevelDB table format):
                                            key ""                      -> SavedTensorSlices{meta: names, shapes, dtypes}
                                            key ordered-code(name,slice) -> SavedTensorSlices{data: SavedSlice{name, slice,
                                                                           TensorProto with float_val / int_val / ...}}
  V2   model_<i>.ckpt.index                 a table: key "" -> BundleHeaderProto; key name -> BundleEntryProto (dtype,
                                            shape, shard_id, offset, size, crc32c)
       model_<i>.ckpt.data-0000k-of-0000n   the raw little-endian tensor bytes, addressed by (shard_id, offset, size)

Both are read here; `load_checkpoint` picks by what is on disk, like `Saver.restore` does.

TensorFlow itself is a third-party dependency of the reference that is absent from this image (the reference pins no
version; its README asks for TF 1.x), and the trained `model_*.ckpt` files are absent from the reference mount (only
the `.ckpt.meta` graphs are there). The formats are therefore restated from their published definitions (TensorFlow
`core/util/{tensor_bundle,tensor_slice_writer,saved_tensor_slice_util}.cc`, `core/protobuf/tensor_bundle.proto`,
`core/util/saved_tensor_slice.proto`, `core/framework/tensor.proto`, `core/lib/io/{format,block,table_builder}.cc`,
`core/lib/strings/ordered_code.cc`; LevelDB `doc/table_format.md`): PARITY UNPINNED against a TensorFlow-written file.
What is pinned: the variable names, dtypes and shapes and the saver version (tests/golden/ckpt_variables.json,
extracted from the reference's own `.ckpt.meta` files by oracle/gen_ckpt_variables.py), the CRC-32C and Snappy known
answers, and writer -> reader round trips of both layouts.

Host logic only (numpy + the CRC-32C of libeae_coder.so); nothing here touches the GPU.
"""
import os
import struct

import numpy

TABLE_MAGIC = 0xdb4775248b80fb57          # io/format.h kTableMagicNumber (same as LevelDB)
FOOTER_LENGTH = 48                        # two padded block handles (2 * 20 bytes) + the 8-byte magic
BLOCK_TRAILER_LENGTH = 5                  # compression type (1 byte) + masked CRC-32C (4 bytes)
NO_COMPRESSION = 0
SNAPPY_COMPRESSION = 1
CRC_MASK_DELTA = 0xa282ead8               # lib/hash/crc32c.h kMaskDelta
RESTART_INTERVAL = 16                     # table::Options::block_restart_interval
BLOCK_SIZE = 256*1024                     # the writer starts a new data block beyond this many bytes

# tensorflow/core/framework/types.proto
DTYPES = {1: numpy.dtype('<f4'), 2: numpy.dtype('<f8'), 3: numpy.dtype('<i4'), 4: numpy.dtype('u1'),
          5: numpy.dtype('<i2'), 6: numpy.dtype('i1'), 9: numpy.dtype('<i8'), 10: numpy.dtype('bool'),
          17: numpy.dtype('<u2'), 22: numpy.dtype('<u4'), 23: numpy.dtype('<u8')}
DTYPE_ENUMS = {dtype: enum for (enum, dtype) in DTYPES.items()}


class CheckpointError(IOError):
    """A malformed, truncated or corrupted checkpoint file."""


# ---- CRC-32C --------------------------------------------------------------------------------------------------------

def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of a bytes-like object, via `eae_crc32c` of the host library."""
    import ctypes
    from .... import _native
    buffer = numpy.frombuffer(data, dtype=numpy.uint8) if not isinstance(data, numpy.ndarray) else data.reshape(-1).view(numpy.uint8)
    buffer = numpy.ascontiguousarray(buffer)
    return int(_native.coder().eae_crc32c(buffer.ctypes.data_as(ctypes.c_void_p), buffer.size, crc))


def mask_crc(crc):
    """lib/hash/crc32c.h Mask: rotate right by 15 bits and add a constant (CRCs of data that embeds CRCs)."""
    return (((crc >> 15) | (crc << 17)) + CRC_MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
    rotated = (masked - CRC_MASK_DELTA) & 0xffffffff
    return ((rotated >> 17) | (rotated << 15)) & 0xffffffff


# ---- protocol-buffer wire format (just what the two bundle messages and a GraphDef walk need) --------------------------

def read_varint(buffer, position):
    result = 0
    shift = 0
    while True:
        if position >= len(buffer):
            raise CheckpointError('Truncated varint.')
        byte = buffer[position]
        position += 1
        result |= (byte & 0x7f) << shift
        if byte < 0x80:
            return (result, position)
        shift += 7
        if shift > 63:
            raise CheckpointError('Varint longer than 64 bits.')


def write_varint(value):
    if value < 0:
        value += 1 << 64
    out = bytearray()
    while value >= 0x80:
        out.append((value & 0x7f) | 0x80)
        value >>= 7
    out.append(value)
    return bytes(out)


def iterate_fields(buffer):
    """Yields (field number, wire type, value) of one message; value is an int (varint, fixed) or a memoryview."""
    buffer = memoryview(buffer)
    position = 0
    while position < len(buffer):
        (tag, position) = read_varint(buffer, position)
        (number, wire) = (tag >> 3, tag & 7)
        if wire == 0:
            (value, position) = read_varint(buffer, position)
        elif wire == 1:
            value = struct.unpack_from('<Q', buffer, position)[0]
            position += 8
        elif wire == 2:
            (length, position) = read_varint(buffer, position)
            if position + length > len(buffer):
                raise CheckpointError('Truncated length-delimited field.')
            value = buffer[position:position + length]
            position += length
        elif wire == 5:
            value = struct.unpack_from('<I', buffer, position)[0]
            position += 4
        else:
            raise CheckpointError('Unsupported protocol-buffer wire type {}.'.format(wire))
        yield (number, wire, value)


def _signed64(value):
    return value - (1 << 64) if value >= 1 << 63 else value


def parse_tensor_shape(buffer):
    """TensorShapeProto (framework/tensor_shape.proto): repeated Dim dim = 2 { int64 size = 1 }; bool unknown_rank = 3."""
    dims = []
    for (number, wire, value) in iterate_fields(buffer):
        if number == 2 and wire == 2:
            size = 0
            for (n2, w2, v2) in iterate_fields(value):
                if n2 == 1 and w2 == 0:
                    size = _signed64(v2)
            dims.append(size)
        elif number == 3 and value:
            raise CheckpointError('A tensor of unknown rank cannot be stored in a bundle.')
    return tuple(dims)


def parse_bundle_header(buffer):
    """BundleHeaderProto: int32 num_shards = 1; Endianness endianness = 2 (0 little, 1 big); VersionDef version = 3."""
    header = {'num_shards': 0, 'endianness': 0, 'producer': 0}
    for (number, wire, value) in iterate_fields(buffer):
        if number == 1 and wire == 0:
            header['num_shards'] = value
        elif number == 2 and wire == 0:
            header['endianness'] = value
        elif number == 3 and wire == 2:
            for (n2, w2, v2) in iterate_fields(value):
                if n2 == 1 and w2 == 0:
                    header['producer'] = v2
    return header


def parse_bundle_entry(buffer):
    """BundleEntryProto: dtype = 1; TensorShapeProto shape = 2; shard_id = 3; offset = 4; size = 5; fixed32 crc32c = 6;
    repeated TensorSliceProto slices = 7 (partitioned variables: not produced by the reference's Saver)."""
    entry = {'dtype': 0, 'shape': (), 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': 0, 'nb_slices': 0}
    for (number, wire, value) in iterate_fields(buffer):
        if number == 1 and wire == 0:
            entry['dtype'] = value
        elif number == 2 and wire == 2:
            entry['shape'] = parse_tensor_shape(value)
        elif number == 3 and wire == 0:
            entry['shard_id'] = value
        elif number == 4 and wire == 0:
            entry['offset'] = _signed64(value)
        elif number == 5 and wire == 0:
            entry['size'] = _signed64(value)
        elif number == 6 and wire == 5:
            entry['crc32c'] = value
        elif number == 7:
            entry['nb_slices'] += 1
    return entry


def _field(number, wire, payload):
    return write_varint((number << 3) | wire) + payload


def encode_tensor_shape(shape):
    return b''.join(_field(2, 2, _length_delimited(_field(1, 0, write_varint(int(size))))) for size in shape)


def _length_delimited(payload):
    return write_varint(len(payload)) + payload


def encode_bundle_header(num_shards, producer=1):
    # proto3: zero-valued scalars (endianness LITTLE = 0) are not serialised
    return _field(1, 0, write_varint(num_shards)) + _field(3, 2, _length_delimited(_field(1, 0, write_varint(producer))))


def encode_bundle_entry(dtype_enum, shape, shard_id, offset, size, masked_crc):
    out = _field(1, 0, write_varint(dtype_enum))
    out += _field(2, 2, _length_delimited(encode_tensor_shape(shape)))
    if shard_id:
        out += _field(3, 0, write_varint(shard_id))
    if offset:
        out += _field(4, 0, write_varint(offset))
    out += _field(5, 0, write_varint(size))
    out += _field(6, 5, struct.pack('<I', masked_crc))
    return out


# ---- Snappy (raw format), decoder only: TensorFlow writes bundle indices uncompressed, other writers may not ----------

def snappy_uncompress(data):
    data = memoryview(data)
    (length, position) = read_varint(data, 0)
    out = bytearray()
    while position < len(data):
        tag = data[position]
        position += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            size = tag >> 2
            if size >= 60:
                nb_bytes = size - 59
                size = int.from_bytes(bytes(data[position:position + nb_bytes]), 'little')
                position += nb_bytes
            size += 1
            if position + size > len(data):
                raise CheckpointError('Truncated Snappy literal.')
            out += data[position:position + size]
            position += size
            continue
        if kind == 1:                                   # copy, 1-byte offset
            size = ((tag >> 2) & 7) + 4
            offset = ((tag >> 5) << 8) | data[position]
            position += 1
        elif kind == 2:                                 # copy, 2-byte offset
            size = (tag >> 2) + 1
            offset = data[position] | (data[position + 1] << 8)
            position += 2
        else:                                           # copy, 4-byte offset
            size = (tag >> 2) + 1
            offset = int.from_bytes(bytes(data[position:position + 4]), 'little')
            position += 4
        if offset == 0 or offset > len(out):
            raise CheckpointError('Snappy copy reaches before the start of the output.')
        for _ in range(size):                           # byte by byte: source and destination may overlap
            out.append(out[-offset])
    if len(out) != length:
        raise CheckpointError('Snappy stream decodes to {0} bytes, its header announces {1}.'.format(len(out), length))
    return bytes(out)


# ---- sorted string table ---------------------------------------------------------------------------------------------

def _read_block(data, offset, size, verify):
    end = offset + size + BLOCK_TRAILER_LENGTH
    if offset < 0 or end > len(data):
        raise CheckpointError('A block handle points outside the table.')
    contents = data[offset:offset + size]
    kind = data[offset + size]
    if verify:
        stored = unmask_crc(struct.unpack_from('<I', data, offset + size + 1)[0])
        if crc32c(data[offset:offset + size + 1]) != stored:
            raise CheckpointError('Block checksum mismatch in the table.')
    if kind == NO_COMPRESSION:
        return bytes(contents)
    if kind == SNAPPY_COMPRESSION:
        return snappy_uncompress(contents)
    raise CheckpointError('Unknown block compression type {}.'.format(kind))


def _iterate_block(block):
    """(key, value) pairs of one block: entries are (shared, non_shared, value_length) varints + key suffix + value;
    the tail holds the restart offsets and their count, which a forward scan does not need."""
    if len(block) < 4:
        raise CheckpointError('Block too short.')
    nb_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4*nb_restarts
    if limit < 0:
        raise CheckpointError('Bad restart array.')
    position = 0
    key = b''
    while position < limit:
        (shared, position) = read_varint(block, position)
        (non_shared, position) = read_varint(block, position)
        (value_length, position) = read_varint(block, position)
        if shared > len(key) or position + non_shared + value_length > limit:
            raise CheckpointError('Corrupted block entry.')
        key = key[:shared] + block[position:position + non_shared]
        position += non_shared
        yield (key, block[position:position + value_length])
        position += value_length


def _decode_handle(buffer, position=0):
    (offset, position) = read_varint(buffer, position)
    (size, position) = read_varint(buffer, position)
    return (offset, size, position)


def read_table(path, verify_checksums=True):
    """All (key, value) pairs of a table file, in key order, as a list of (bytes, bytes)."""
    with open(path, 'rb') as file:
        data = file.read()
    if len(data) < FOOTER_LENGTH:
        raise CheckpointError('"{}" is too short to be a table.'.format(path))
    footer = data[-FOOTER_LENGTH:]
    if struct.unpack_from('<Q', footer, FOOTER_LENGTH - 8)[0] != TABLE_MAGIC:
        raise CheckpointError('"{}" is not a table (bad magic number): a V1 checkpoint or another file.'.format(path))
    (_, _, position) = _decode_handle(footer)                       # metaindex block: unused by bundles
    (index_offset, index_size, _) = _decode_handle(footer, position)
    pairs = []
    for (_, handle) in _iterate_block(_read_block(data, index_offset, index_size, verify_checksums)):
        (offset, size, _) = _decode_handle(handle)
        pairs.extend(_iterate_block(_read_block(data, offset, size, verify_checksums)))
    return pairs


class _BlockBuilder(object):
    def __init__(self):
        self.buffer = bytearray()
        self.restarts = [0]
        self.counter = 0
        self.last_key = b''

    def add(self, key, value):
        shared = 0
        if self.counter < RESTART_INTERVAL:
            limit = min(len(key), len(self.last_key))
            while shared < limit and key[shared] == self.last_key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buffer))
            self.counter = 0
        self.buffer += write_varint(shared) + write_varint(len(key) - shared) + write_varint(len(value))
        self.buffer += key[shared:] + value
        self.last_key = key
        self.counter += 1

    def finish(self):
        return bytes(self.buffer) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def write_table(path, pairs):
    """Writes sorted (key, value) pairs as an uncompressed table (what BundleWriter::Finish asks of its TableBuilder)."""
    keys = [key for (key, _) in pairs]
    if keys != sorted(keys) or len(set(keys)) != len(keys):
        raise ValueError('Table keys must be strictly increasing.')
    out = bytearray()

    def emit(block):
        handle = write_varint(len(out)) + write_varint(len(block))
        out.extend(block)
        out.append(NO_COMPRESSION)
        out.extend(struct.pack('<I', mask_crc(crc32c(block + bytes([NO_COMPRESSION])))))
        return handle

    index = _BlockBuilder()
    builder = _BlockBuilder()
    for (key, value) in pairs:
        builder.add(key, value)
        if len(builder.buffer) >= BLOCK_SIZE:
            index.add(builder.last_key, emit(builder.finish()))       # any separator >= the block's last key is valid
            builder = _BlockBuilder()
    if builder.counter or not pairs:
        index.add(builder.last_key, emit(builder.finish()))
    metaindex_handle = emit(_BlockBuilder().finish())
    index_handle = emit(index.finish())
    footer = metaindex_handle + index_handle
    out.extend(footer + b'\x00'*(FOOTER_LENGTH - 8 - len(footer)) + struct.pack('<Q', TABLE_MAGIC))
    with open(path, 'wb') as file:
        file.write(out)


# ---- bundles ----------------------------------------------------------------------------------------------------------

def data_filename(prefix, shard_id, num_shards):
    return '{0}.data-{1:05d}-of-{2:05d}'.format(prefix, shard_id, num_shards)


def _has_table_magic(path):
    with open(path, 'rb') as file:
        file.seek(0, os.SEEK_END)
        if file.tell() < FOOTER_LENGTH:
            return False
        file.seek(-8, os.SEEK_END)
        return struct.unpack('<Q', file.read(8))[0] == TABLE_MAGIC


def checkpoint_format(prefix):
    """2 if `prefix.index` exists (V2 bundle), 1 if `prefix` or its `-?????-of-?????` shards exist (V1), else 0.
    The order is the one `Saver.restore` probes in."""
    if os.path.isfile(prefix + '.index'):
        return 2
    files = v1_files(prefix)
    return 1 if files and all(_has_table_magic(path) for path in files) else 0


def exists(prefix):
    """`tf.train.checkpoint_exists`: True if `prefix` names a V1 or a V2 checkpoint."""
    return checkpoint_format(prefix) != 0


def list_variables(prefix, verify_checksums=True):
    """{name: (numpy dtype or None, shape)} like `tf.train.list_variables`."""
    version = checkpoint_format(prefix)
    if version == 1:
        meta = {}
        for path in v1_files(prefix):
            pairs = read_table(path, verify_checksums)
            if not pairs or pairs[0][0] != b'':
                raise CheckpointError('"{}" has no tensor-slice metadata entry.'.format(path))
            meta.update(_read_v1_meta(pairs[0][1]))
        return {name: (DTYPES.get(enum), shape) for (name, (enum, shape)) in meta.items()}
    (_, entries) = _read_index(prefix, verify_checksums)
    return {name: (DTYPES.get(entry['dtype']), entry['shape']) for (name, entry) in entries.items()}


def _read_index(prefix, verify_checksums):
    if not os.path.isfile(prefix + '.index'):
        raise IOError('The checkpoint "{}" does not exist.'.format(prefix))
    pairs = read_table(prefix + '.index', verify_checksums)
    if not pairs or pairs[0][0] != b'':
        raise CheckpointError('"{}.index" has no bundle header.'.format(prefix))
    header = parse_bundle_header(pairs[0][1])
    if header['endianness'] != 0:
        raise CheckpointError('Big-endian bundles are not supported.')
    if header['num_shards'] < 1:
        raise CheckpointError('The bundle header announces no data shard.')
    entries = {key.decode('utf-8'): parse_bundle_entry(value) for (key, value) in pairs[1:]}
    return (header, entries)


def load_checkpoint(prefix, names=None, verify_checksums=True):
    """Reads tensors of the checkpoint `prefix` (e.g. ".../model_10.ckpt") into a dict of numpy arrays.

    `names`: the variables wanted (KeyError if one is missing); None reads every tensor of a supported dtype (the
    reference's checkpoints also hold the optimiser slots, `decaying_lr/global_step` and the piecewise-linear
    function's variables). V1 or V2 is picked by what is on disk (`checkpoint_format`).
    """
    if checkpoint_format(prefix) == 1:
        return _load_v1(prefix, None if names is None else set(names), verify_checksums)[1]
    (header, entries) = _read_index(prefix, verify_checksums)
    wanted = list(entries) if names is None else list(names)
    shards = {}
    result = {}
    for name in wanted:
        if name not in entries:
            raise KeyError('The variable "{0}" is not in the checkpoint "{1}".'.format(name, prefix))
        entry = entries[name]
        if entry['dtype'] not in DTYPES or entry['nb_slices']:
            if names is None:
                continue
            raise CheckpointError('The variable "{}" has an unsupported dtype or is partitioned.'.format(name))
        dtype = DTYPES[entry['dtype']]
        nb_elements = int(numpy.prod(entry['shape'], dtype=numpy.int64)) if entry['shape'] else 1
        if nb_elements*dtype.itemsize != entry['size']:
            raise CheckpointError('The variable "{}": byte size and shape disagree.'.format(name))
        shard_id = entry['shard_id']
        if shard_id not in shards:
            path = data_filename(prefix, shard_id, header['num_shards'])
            if not os.path.isfile(path):
                raise IOError('The data file "{}" does not exist.'.format(path))
            shards[shard_id] = numpy.memmap(path, dtype=numpy.uint8, mode='r')
        shard = shards[shard_id]
        if entry['offset'] < 0 or entry['offset'] + entry['size'] > shard.size:
            raise CheckpointError('The variable "{}" reaches beyond its data file.'.format(name))
        raw = numpy.array(shard[entry['offset']:entry['offset'] + entry['size']])
        if verify_checksums and crc32c(raw) != unmask_crc(entry['crc32c']):
            raise CheckpointError('Checksum mismatch for the variable "{}".'.format(name))
        result[name] = raw.view(dtype).reshape(entry['shape']).astype(dtype.newbyteorder('='), copy=False)
    return result


def _dtype_enum(name, array):
    for (enum, dtype) in DTYPES.items():
        if (array.dtype.kind, array.dtype.itemsize) == (dtype.kind, dtype.itemsize):
            return enum
    raise ValueError('The variable "{0}" has the unsupported dtype {1}.'.format(name, array.dtype))


def save_checkpoint(prefix, variables):
    """Writes {name: numpy array} as a one-shard V2 bundle: `prefix.index` + `prefix.data-00000-of-00001`."""
    pairs = [(b'', encode_bundle_header(1))]
    offset = 0
    with open(data_filename(prefix, 0, 1), 'wb') as file:
        for name in sorted(variables, key=lambda n: n.encode('utf-8')):
            array = numpy.asarray(variables[name], order='C')
            enum = _dtype_enum(name, array)
            raw = array.astype(DTYPES[enum], copy=False).tobytes()
            file.write(raw)
            pairs.append((name.encode('utf-8'), encode_bundle_entry(enum, array.shape, 0, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + '.index', pairs)


# ---- V1 checkpoints (TensorSliceWriter) ----------------------------------------------------------------------------------

def _packed(value, wire, fmt, out):
    """One occurrence of a repeated numeric field: packed (length-delimited) or a single element."""
    if wire == 2:
        if fmt == 'varint':
            position = 0
            while position < len(value):
                (element, position) = read_varint(value, position)
                out.append(_signed64(element))
        else:
            out.extend(numpy.frombuffer(bytes(value), dtype=fmt).tolist())
    elif fmt == 'varint':
        out.append(_signed64(value))
    else:
        size = numpy.dtype(fmt).itemsize
        out.append(numpy.frombuffer(int(value).to_bytes(size, 'little'), dtype=fmt)[0].item())


def parse_tensor_proto(buffer):
    """TensorProto (framework/tensor.proto) -> (dtype enum, shape or None, flat numpy array).

    dtype = 1; tensor_shape = 2; tensor_content = 4 (raw bytes); float_val = 5; double_val = 6; int_val = 7;
    int64_val = 10; bool_val = 11. The V1 writer fills the typed `*_val` lists (saved_tensor_slice_util.h)."""
    (dtype_enum, shape, content) = (0, None, None)
    floats = []
    values = []
    for (number, wire, value) in iterate_fields(buffer):
        if number == 1 and wire == 0:
            dtype_enum = value
        elif number == 2 and wire == 2:
            shape = parse_tensor_shape(value)
        elif number == 4 and wire == 2:
            content = bytes(value)
        elif number == 5:
            if wire == 2:
                floats.append(numpy.frombuffer(bytes(value), dtype='<f4'))
            else:
                floats.append(numpy.frombuffer(struct.pack('<I', value), dtype='<f4'))
        elif number == 6:
            _packed(value, wire, '<f8', values)
        elif number in (7, 10, 11):
            _packed(value, wire, 'varint', values)
    if dtype_enum not in DTYPES:
        raise CheckpointError('Unsupported tensor dtype {}.'.format(dtype_enum))
    dtype = DTYPES[dtype_enum]
    if content is not None:
        flat = numpy.frombuffer(content, dtype=dtype)
    elif floats:
        flat = numpy.concatenate(floats).astype(dtype, copy=False)
    else:
        flat = numpy.array(values).astype(dtype) if values else numpy.zeros(0, dtype=dtype)
    return (dtype_enum, shape, flat)


def _parse_slice(buffer, shape):
    """TensorSliceProto: repeated Extent extent = 1 { int64 start = 1; int64 length = 2 (absent = the full dimension) }."""
    extents = []
    for (number, wire, value) in iterate_fields(buffer):
        if number == 1 and wire == 2:
            (start, length) = (0, None)
            for (n2, w2, v2) in iterate_fields(value):
                if n2 == 1 and w2 == 0:
                    start = _signed64(v2)
                elif n2 == 2 and w2 == 0:
                    length = _signed64(v2)
            extents.append((start, length))
    if len(extents) != len(shape):
        raise CheckpointError('A slice has {0} extents for a tensor of rank {1}.'.format(len(extents), len(shape)))
    return tuple(slice(start, dim if length is None or length < 0 else start + length)
                 for ((start, length), dim) in zip(extents, shape))


def _read_v1_meta(buffer):
    """SavedTensorSlices.meta (field 1) -> {name: (dtype enum, shape)}; SavedSliceMeta: name = 1; shape = 2; type = 3."""
    tensors = {}
    for (number, wire, value) in iterate_fields(buffer):
        if number != 1 or wire != 2:
            continue
        for (n1, w1, v1) in iterate_fields(value):
            if n1 != 1 or w1 != 2:                                   # SavedTensorSliceMeta.tensor
                continue
            (name, shape, dtype_enum) = (None, (), 0)
            for (n2, w2, v2) in iterate_fields(v1):
                if n2 == 1:
                    name = bytes(v2).decode('utf-8')
                elif n2 == 2:
                    shape = parse_tensor_shape(v2)
                elif n2 == 3:
                    dtype_enum = v2
            tensors[name] = (dtype_enum, shape)
    return tensors


def v1_files(prefix):
    """The table file(s) of a V1 checkpoint: `prefix` itself, or the shards `prefix-?????-of-?????` of a sharded Saver."""
    if os.path.isfile(prefix):
        return [prefix]
    directory = os.path.dirname(prefix) or '.'
    base = os.path.basename(prefix)
    if not os.path.isdir(directory):
        return []
    shards = [name for name in sorted(os.listdir(directory))
              if name.startswith(base + '-') and len(name) == len(base) + 15 and name[len(base) + 6:len(base) + 10] == '-of-']
    return [os.path.join(directory, name) for name in shards]


def _load_v1(prefix, names, verify_checksums):
    meta = {}
    arrays = {}
    for path in v1_files(prefix):
        pairs = read_table(path, verify_checksums)
        if not pairs or pairs[0][0] != b'':
            raise CheckpointError('"{}" has no tensor-slice metadata entry.'.format(path))
        local = _read_v1_meta(pairs[0][1])
        meta.update(local)
        for (_, value) in pairs[1:]:
            for (number, wire, saved) in iterate_fields(value):
                if number != 2 or wire != 2:                         # SavedTensorSlices.data
                    continue
                (name, extents, tensor) = (None, None, None)
                for (n2, w2, v2) in iterate_fields(saved):           # SavedSlice: name = 1; slice = 2; data = 3
                    if n2 == 1:
                        name = bytes(v2).decode('utf-8')
                    elif n2 == 2:
                        extents = v2
                    elif n2 == 3:
                        tensor = v2
                if name not in local:
                    raise CheckpointError('The slice of "{}" has no metadata.'.format(name))
                if names is not None and name not in names:
                    continue
                (dtype_enum, shape) = local[name]
                if dtype_enum not in DTYPES:
                    if names is None:
                        continue
                    raise CheckpointError('The variable "{}" has an unsupported dtype.'.format(name))
                (_, _, flat) = parse_tensor_proto(tensor)
                if name not in arrays:
                    arrays[name] = numpy.zeros(shape, dtype=DTYPES[dtype_enum].newbyteorder('='))
                index = _parse_slice(extents, shape) if extents is not None else tuple(slice(0, d) for d in shape)
                target = arrays[name][index] if shape else arrays[name]
                if flat.size != target.size:
                    raise CheckpointError('The variable "{0}": {1} stored elements for a slice of {2}.'.format(
                        name, flat.size, target.size))
                if shape:
                    arrays[name][index] = flat.reshape(target.shape)
                else:
                    arrays[name][...] = flat.reshape(())
    if names is not None:
        for name in names:
            if name not in arrays:
                raise KeyError('The variable "{0}" is not in the checkpoint "{1}".'.format(name, prefix))
    return (meta, arrays)


def _ordered_number(value):
    """strings/ordered_code.cc WriteNumIncreasing: a length byte, then the big-endian bytes without leading zeros."""
    payload = value.to_bytes((value.bit_length() + 7)//8, 'big')
    return bytes([len(payload)]) + payload


def _ordered_signed_small(value):
    """WriteSignedNumIncreasing for -64 <= value < 64 (one byte), all a full-extent slice needs (start 0, length -1)."""
    if not -64 <= value < 64:
        raise ValueError('Only one-byte signed ordered codes are written.')
    return bytes([0x80 ^ (value & 0xff)])


def encode_tensor_name_slice(name, rank):
    """saved_tensor_slice_util.cc EncodeTensorNameSlice for the slice covering a whole tensor of this rank."""
    escaped = b''.join(b'\x00\xff' if byte == 0 else (b'\xff\x00' if byte == 0xff else bytes([byte]))
                       for byte in name.encode('utf-8')) + b'\x00\x01'
    key = _ordered_number(0) + escaped + _ordered_number(rank)
    for _ in range(rank):
        key += _ordered_signed_small(0) + _ordered_signed_small(-1)
    return key


def save_checkpoint_v1(prefix, variables):
    """Writes {name: numpy array} as a single-file V1 checkpoint (what `Saver(write_version=V1).save` lays out)."""
    meta = b''
    pairs = []
    for name in sorted(variables):
        array = numpy.asarray(variables[name], order='C')
        enum = _dtype_enum(name, array)
        shape = encode_tensor_shape(array.shape)
        full_slice = b''.join(_field(1, 2, _length_delimited(b'')) for _ in array.shape)
        meta += _field(1, 2, _length_delimited(_field(1, 2, _length_delimited(name.encode('utf-8'))) +
                                               _field(2, 2, _length_delimited(shape)) + _field(3, 0, write_varint(enum)) +
                                               _field(4, 2, _length_delimited(full_slice))))
        flat = array.reshape(-1)
        if array.dtype.kind == 'f' and array.dtype.itemsize == 4:
            payload = _field(5, 2, _length_delimited(flat.astype('<f4').tobytes()))
        elif array.dtype.kind == 'f':
            payload = _field(6, 2, _length_delimited(flat.astype('<f8').tobytes()))
        else:
            number = 10 if enum == 9 else (11 if enum == 10 else 7)
            payload = _field(number, 2, _length_delimited(b''.join(write_varint(int(v)) for v in flat)))
        tensor = _field(1, 0, write_varint(enum)) + _field(2, 2, _length_delimited(shape)) + payload
        saved = (_field(1, 2, _length_delimited(name.encode('utf-8'))) + _field(2, 2, _length_delimited(full_slice)) +
                 _field(3, 2, _length_delimited(tensor)))
        pairs.append((encode_tensor_name_slice(name, array.ndim), _field(2, 2, _length_delimited(saved))))
    versions = _field(1, 0, write_varint(1))
    header = _field(1, 2, _length_delimited(meta + _field(2, 2, _length_delimited(versions))))
    write_table(prefix, [(b'', header)] + sorted(pairs))


def latest_checkpoint(directory):
    """`tf.train.latest_checkpoint`: the prefix named by `model_checkpoint_path` in `<directory>/checkpoint`, or None."""
    state = os.path.join(directory, 'checkpoint')
    if not os.path.isfile(state):
        return None
    with open(state, 'r') as file:
        for line in file:
            if line.startswith('model_checkpoint_path:'):
                name = line.split(':', 1)[1].strip().strip('"')
                prefix = name if os.path.isabs(name) else os.path.join(directory, name)
                return prefix if exists(prefix) else None
    return None


def write_checkpoint_state(directory, prefix_basename):
    """The `checkpoint` text file a Saver keeps beside its bundles (CheckpointState in text format)."""
    with open(os.path.join(directory, 'checkpoint'), 'w') as file:
        file.write('model_checkpoint_path: "{0}"\nall_model_checkpoint_paths: "{0}"\n'.format(prefix_basename))
