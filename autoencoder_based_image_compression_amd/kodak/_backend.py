"""numpy <-> HBM plumbing shared by the reference-shaped modules. No CPU fallback: without a GPU every op raises."""
import numpy
import torch


class NoDeviceError(RuntimeError):
    pass


def device():
    if not torch.cuda.is_available():
        raise NoDeviceError('No MI355X visible: the product path runs on the gfx950 kernels only (there is no CPU '
                            'fallback; the CPU oracle under oracle/ is test infrastructure).')
    return torch.device('cuda', torch.cuda.current_device())


def to_device(array, dtype=None):
    a = numpy.ascontiguousarray(array, dtype=dtype)
    if not a.flags.writeable:          # e.g. numpy.asarray(PIL image): torch refuses to alias read-only memory silently
        a = a.copy()
    return torch.from_numpy(a).to(device(), non_blocking=False)


def to_host(tensor):
    return tensor.cpu().numpy()
