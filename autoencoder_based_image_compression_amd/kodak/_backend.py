"""numpy <-> HBM plumbing shared by the reference-shaped modules. No CPU fallback: without a GPU every op raises.

The reference's call surface is numpy in / numpy out at every call (eae/batching.py:44-53, 94-99; tools/tools.py;
lossless/compression.py), and its script hands what one call returned to the next ones (reconstructing_eae_kodak.py:185-225:
`quantize_per_map` -> `count_nb_deads`, `rescale_compress_lossless_maps(cq[j])`, `rate_3d(cq[j])`; `decode_mini_batches` ->
`psnr_2d(ref[j], rec[j])`). The signatures stay; what a call produced stays RESIDENT in HBM next to the numpy array it
returned, so the next call that is handed that array -- or a C-contiguous view of it, like `cq[j, :, :, :]` -- finds it on the
device instead of sending it back:

* `publish(tensor)` copies a device tensor into a fresh numpy array, marks that array READ-ONLY and registers it (by identity,
  through a weak reference: the device copy is dropped when the array dies);
* `resident(array)` finds the registered array an argument is, or is a view of (the `.base` chain), and returns the matching
  slice of the device copy -- only while the registered array is still read-only: an array somebody made writable again is
  forgotten and uploaded like any other (numpy refuses writes through every view of a read-only array, so the device copy
  cannot go stale behind the registry's back short of raw pointer access);
* `to_device(array)` = `resident` or an upload.

The returned arrays being read-only is the one visible difference from the reference's functions (an in-place write into a
returned array raises `ValueError: assignment destination is read-only` instead of succeeding); the reference's own scripts
never write into them. `EAE_SURFACE_RESIDENT=0` turns the registry off: plain writable arrays, every call uploads.
"""
import os
import weakref

import numpy
import torch

RESIDENT_ENABLED = os.environ.get('EAE_SURFACE_RESIDENT', '1') != '0'


class NoDeviceError(RuntimeError):
    pass


def device():
    if not torch.cuda.is_available():
        raise NoDeviceError('No MI355X visible: the product path runs on the gfx950 kernels only (there is no CPU '
                            'fallback; the CPU oracle under oracle/ is test infrastructure).')
    return torch.device('cuda', torch.cuda.current_device())


class Resident(object):
    """The device side of one published array: `tensor` (same shape and dtype as the array) and `extras`, whatever the
    publishing function or a later call computed from it on the device (symbols, coder results ...), keyed by the caller."""

    __slots__ = ('tensor', 'extras', 'nbytes', '__weakref__')

    def __init__(self, tensor):
        self.tensor = tensor
        self.extras = {}
        self.nbytes = tensor.numel()*tensor.element_size()


_REGISTRY = {}        # id(array) -> (weak reference to the array, Resident)
statistics = {'published': 0, 'hits': 0, 'uploads': 0, 'forgotten_writable': 0, 'buffers_reused': 0}

# Host memory of the published arrays: a Kodak set's latents are 18.9 MB, and a fresh allocation of that size costs more in
# first-touch page faults (about 1.2 ms) than the device -> host copy that fills it (0.35 ms). The bytes of a published array
# that has died (no view of it is left either: views keep it alive) go back to this pool and serve the next array of that size.
_POOL = {}            # nbytes -> list of free bytearrays
_POOL_LIMIT_BYTES = int(os.environ.get('EAE_SURFACE_POOL_BYTES', str(512 << 20)))
_pool_bytes = [0]


def _forget(key, buffer=None):
    _REGISTRY.pop(key, None)
    if buffer is not None and _pool_bytes[0] + len(buffer) <= _POOL_LIMIT_BYTES:
        _POOL.setdefault(len(buffer), []).append(buffer)
        _pool_bytes[0] += len(buffer)


def _host_array(shape, dtype):
    """(numpy array of that shape over pooled bytes, the bytearray). The array's `.base` is the bytearray -- not another
    ndarray -- so that every view a caller takes has the ARRAY as its base (numpy collapses view chains onto the first array
    that sits on foreign memory): views keep it alive, and `_owner` finds it."""
    nbytes = int(numpy.prod(shape, dtype=numpy.int64))*numpy.dtype(dtype).itemsize
    free = _POOL.get(nbytes)
    if free:
        buffer = free.pop()
        _pool_bytes[0] -= nbytes
        statistics['buffers_reused'] += 1
    else:
        buffer = bytearray(max(nbytes, 1))
    return (numpy.ndarray(shape, dtype=dtype, buffer=buffer), buffer)


def register(array, tensor, buffer=None):
    """Registers `tensor` as the device copy of `array` (which becomes read-only). Returns the `Resident`. `buffer`: the pooled
    bytes under `array` (`_host_array`), handed back to the pool when the array dies."""
    record = Resident(tensor)
    if not RESIDENT_ENABLED:
        return record
    array.flags.writeable = False
    key = id(array)
    _REGISTRY[key] = (weakref.ref(array, lambda _ref, key=key, buffer=buffer: _forget(key, buffer)), record)
    statistics['published'] += 1
    return record


def publish(tensor):
    """Device tensor -> numpy array with the same contents (one synchronous device -> host copy), registered. Returns
    (array, Resident)."""
    shape = tuple(tensor.shape)
    if not RESIDENT_ENABLED:
        array = numpy.empty(shape, dtype=_numpy_dtype(tensor.dtype))
        torch.from_numpy(array).copy_(tensor)
        return (array, Resident(tensor))
    (array, buffer) = _host_array(shape, _numpy_dtype(tensor.dtype))
    torch.from_numpy(array).copy_(tensor)
    return (array, register(array, tensor, buffer))


_NUMPY_DTYPES = {torch.float32: numpy.float32, torch.float64: numpy.float64, torch.uint8: numpy.uint8, torch.int16: numpy.int16,
                 torch.int32: numpy.int32, torch.int64: numpy.int64}


def _numpy_dtype(torch_dtype):
    return _NUMPY_DTYPES[torch_dtype]


def _registered(array):
    entry = _REGISTRY.get(id(array))
    if entry is None:
        return None
    (ref, record) = entry
    if ref() is not array:               # an id that has been reused before the callback ran
        _forget(id(array))
        return None
    if array.flags.writeable:            # made writable again by its owner: it may have changed, never trust it again
        _forget(id(array))
        statistics['forgotten_writable'] += 1
        return None
    return record


def _owner(array):
    """(registered array, its Resident) for `array` itself or the first registered array on its `.base` chain, else None."""
    node = array
    for _ in range(8):
        if not isinstance(node, numpy.ndarray):
            return None
        record = _registered(node)
        if record is not None:
            return (node, record)
        node = node.base
    return None


def resident(array):
    """(Resident of the registered array `array` is or is a view of, first element of `array` in it) or None. Only
    C-contiguous views of the same dtype that lie inside the registered array qualify."""
    if not RESIDENT_ENABLED or not _REGISTRY or not isinstance(array, numpy.ndarray):
        return None
    found = _owner(array)
    if found is None:
        return None
    (owner, record) = found
    if array is owner:
        return (record, 0)
    if not array.flags.c_contiguous or array.dtype != owner.dtype or array.size == 0:
        return None
    delta = array.ctypes.data - owner.ctypes.data
    if delta < 0 or delta % owner.itemsize or delta//owner.itemsize + array.size > owner.size:
        return None
    return (record, delta//owner.itemsize)


def resident_tensor(array, dtype=None):
    """The device tensor holding `array`'s values, shaped like `array`, if `array` is (a view of) a published array on the
    current device -- else None."""
    found = resident(array)
    if found is None:
        return None
    (record, first) = found
    tensor = record.tensor
    if tensor.device != device() or (dtype is not None and _numpy_dtype(tensor.dtype) != numpy.dtype(dtype)):
        return None
    statistics['hits'] += 1
    if first == 0 and tuple(tensor.shape) == array.shape:
        return tensor
    return tensor.reshape(-1)[first:first + array.size].view(array.shape)


def to_device(array, dtype=None):
    """Device tensor with `array`'s values (converted to `dtype`): the resident copy when there is one, else an upload."""
    if isinstance(array, numpy.ndarray) and (dtype is None or array.dtype == numpy.dtype(dtype)):
        tensor = resident_tensor(array, dtype)
        if tensor is not None:
            return tensor
    a = numpy.ascontiguousarray(array, dtype=dtype)
    if not a.flags.writeable:          # e.g. numpy.asarray(PIL image), or a published array: torch refuses to alias read-only
        view = a.view()                # memory silently; the alias below is only read (one host -> device copy)
        try:
            view.flags.writeable = True
            a = view
        except ValueError:             # a view of foreign read-only memory: numpy will not lift the flag
            a = a.copy()
    statistics['uploads'] += 1
    return torch.from_numpy(a).to(device(), non_blocking=False)


def to_host(tensor):
    return tensor.cpu().numpy()
