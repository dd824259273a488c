"""Device-resident ops: torch tensors (used only as HBM buffers + streams) -> include/eae_hip.h entry points.

Every function launches asynchronously on torch's current stream and returns device tensors. There is no CPU
fallback: a missing libeae_hip.so or a non-CUDA tensor raises.
"""
import torch

from . import _native

NORM_NONE, NORM_GDN, NORM_IGDN = 0, 1, 2
NB_MAPS = 128


class HipError(RuntimeError):
    pass


def _check(status, what):
    if status != 0:
        raise HipError('{0} failed with status {1}'.format(what, status))


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(on=None):
    """hipStream_t of torch's current stream on the current device (every launch of this module goes there). `on`: the
    launch's main tensor -- it must live on the current device (a kernel launched on another device's stream with foreign
    pointers faults or silently serialises): select the device with `torch.cuda.device(...)` / `set_device` first."""
    current = torch.cuda.current_device()
    if on is not None and on.device.index != current:
        raise HipError('tensor on {0} but the current device is cuda:{1}: wrap the call in `with torch.cuda.device({0!r})`'.format(
            on.device, current))
    if _raw_stream is not None:                      # one C call instead of building a torch.cuda.Stream object per launch
        return _raw_stream(current)
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError('expected a device tensor')
    if not t.is_contiguous():
        raise HipError('expected a contiguous tensor')
    return t.data_ptr()


def device_info():
    import ctypes
    name = ctypes.create_string_buffer(128)
    cus = ctypes.c_int(0)
    mhz = ctypes.c_int(0)
    mem = ctypes.c_int64(0)
    _check(_native.hip().eae_hip_device_info(name, 128, ctypes.byref(cus), ctypes.byref(mhz), ctypes.byref(mem)), 'device_info')
    return {'name': name.value.decode(), 'compute_units': cus.value, 'clock_mhz': mhz.value, 'hbm_bytes': mem.value}


def partition_info():
    """{'compute_units', 'xcds', 'whole_device'} of the current logical device (include/eae_hip.h: eae_hip_partition_info)."""
    import ctypes
    (cus, xcds, whole) = (ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0))
    _check(_native.hip().eae_hip_partition_info(ctypes.byref(cus), ctypes.byref(xcds), ctypes.byref(whole)), 'partition_info')
    return {'compute_units': cus.value, 'xcds': xcds.value, 'whole_device': bool(whole.value)}


class Model(object):
    """eae_hip_model (include/eae_hip.h, whole-path entry points): the variables of one trained entropy autoencoder resident
    on the current device in the kernels' layouts. `variables`: dict of numpy arrays keyed by the TensorFlow variable names
    (kodak/eae/graph/variables.py); encoder-only and decoder-only dicts are accepted."""

    _FIELDS = ('weights_1', 'biases_1', 'gamma_1', 'beta_1', 'weights_2', 'biases_2', 'gamma_2', 'beta_2', 'weights_3', 'biases_3',
               'gamma_3', 'beta_3', 'gamma_4', 'beta_4', 'weights_4', 'biases_4', 'gamma_5', 'beta_5', 'weights_5', 'biases_5',
               'gamma_6', 'beta_6', 'weights_6')

    def __init__(self, variables, are_bin_widths_learned):
        import ctypes
        import numpy
        self.are_bin_widths_learned = bool(are_bin_widths_learned)
        self.device = torch.device('cuda', torch.cuda.current_device())
        pointers = (ctypes.c_void_p*len(self._FIELDS))()
        keep = []
        for (i, field) in enumerate(self._FIELDS):
            name = ('encoder/' if int(field[-1]) <= 3 else 'decoder/') + field
            if name in variables and not (self.are_bin_widths_learned and field in ('gamma_3', 'beta_3', 'gamma_4', 'beta_4')):
                array = numpy.ascontiguousarray(variables[name], dtype=numpy.float32)
                keep.append(array)
                pointers[i] = array.ctypes.data
        handle = ctypes.c_void_p()
        _check(_native.hip().eae_hip_model_create(ctypes.cast(pointers, ctypes.c_void_p), 1 if self.are_bin_widths_learned else 0,
                                                  ctypes.byref(handle)), 'eae_hip_model_create')
        self._handle = handle
        # the failure word of a call sits behind the conv workspace at the head of its scratch block (eae_hip_transform_status);
        # calls are asynchronous, so the word is published to pinned memory behind each call and looked at later (`check`)
        self._status_offset = (int(_native.hip().eae_hip_conv_workspace_bytes()) + 255)//256*256
        self._pending = []          # (event, pinned word) of calls nobody has looked at yet
        self._free_words = []

    def _track(self, scratch):
        word = scratch[self._status_offset:self._status_offset + 4].view(torch.int32)
        pinned = self._free_words.pop() if self._free_words else torch.zeros(1, dtype=torch.int32).pin_memory()
        publish_to_host(word, pinned)
        event = torch.cuda.Event()
        event.record()
        self._pending.append((event, pinned))

    def check(self, wait=False):
        """Raises `SplitHandOffTimeout` if an encode / decode call that has completed (wait=True: any call issued so far) left
        tiles unfinished (include/eae_hip.h: eae_hip_conv_workspace_collect) -- its outputs are then invalid. Every encode /
        decode looks at the completed calls first; the reference-shaped `sess.run` nodes wait, right after their device -> host copy."""
        (still, failed) = ([], 0)
        for (event, pinned) in self._pending:
            if wait:
                event.synchronize()
            if event.query():
                failed += int(pinned[0])
                self._free_words.append(pinned)
            else:
                still.append((event, pinned))
        self._pending = still
        if failed:
            raise SplitHandOffTimeout('{} tiles of a cut conv launch were not handed over (eae_hip_transform_status): the outputs '
                                      'of that call are invalid'.format(failed))

    def close(self):
        if getattr(self, '_handle', None):
            _native.hip().eae_hip_model_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, images_u8, out=None):
        """uint8 [N,H,W] (device) -> latents f32 [N,H/16,W/16,128]: `sess.run(node_y)` of eae/batching.py:96-99 in one call.
        out: a contiguous float32 tensor of that shape to write into (e.g. a mini-batch's slice of the whole set's latents)."""
        if images_u8.dtype != torch.uint8:
            raise TypeError('`images_u8.dtype` is not `torch.uint8`.')
        (n, h, wd) = images_u8.shape[:3]
        nbytes = int(_native.hip().eae_hip_encode_scratch_bytes(n, h, wd))
        if nbytes == 0:
            raise ValueError('The image size is not divisible by the product of the three strides.')
        if images_u8.device != self.device:
            raise HipError('images on {0} but the model lives on {1}'.format(images_u8.device, self.device))
        self.check()
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=images_u8.device)
        latents = out if out is not None else torch.empty((n, h//16, wd//16, NB_MAPS), dtype=torch.float32, device=images_u8.device)
        if latents.dtype != torch.float32 or tuple(latents.shape) != (n, h//16, wd//16, NB_MAPS):
            raise HipError('`out` must be float32 of shape (N, H/16, W/16, 128)')
        if not latents.is_contiguous() or latents.device != self.device:
            # the kernels get a raw pointer and write N x h x w x 128 floats flat behind it
            raise HipError('`out` must be a contiguous tensor on {0} (got {1}, {2})'.format(
                self.device, 'contiguous' if latents.is_contiguous() else 'non-contiguous', latents.device))
        _check(_native.hip().eae_hip_encode(self._handle, _p(images_u8), n, h, wd, _p(latents), _p(scratch), nbytes, _stream(images_u8)),
               'eae_hip_encode')
        self._track(scratch)
        return latents

    def decode(self, quantized_latents, want_f32=False, want_u8=True, ref_u8=None, sse=None, out_u8=None):
        """f32 [N,h,w,128] (device) -> (f32 [N,16h,16w] or None, uint8 or None, sse or None): `sess.run(node_reconstruction)` +
        `tls.cast_bt601` of eae/batching.py:49-53 (+ the squared error of tls.psnr_2d) in one call. out_u8: a contiguous uint8
        tensor of N x 16h x 16w elements to write the reconstruction into."""
        (n, h, wd, c) = quantized_latents.shape
        d = quantized_latents.device
        if d != self.device:
            raise HipError('latents on {0} but the model lives on {1}'.format(d, self.device))
        self.check()
        nbytes = int(_native.hip().eae_hip_decode_scratch_bytes(n, h, wd))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=d)
        out_f32 = torch.empty((n, 16*h, 16*wd), dtype=torch.float32, device=d) if want_f32 else None
        if out_u8 is not None and (out_u8.dtype != torch.uint8 or out_u8.numel() != n*16*h*16*wd):
            raise HipError('`out_u8` must hold N x 16h x 16w uint8 elements')
        if out_u8 is not None and (not out_u8.is_contiguous() or out_u8.device != self.device):
            raise HipError('`out_u8` must be a contiguous tensor on {0} (got {1}, {2})'.format(
                self.device, 'contiguous' if out_u8.is_contiguous() else 'non-contiguous', out_u8.device))
        if out_u8 is None:
            out_u8 = torch.empty((n, 16*h, 16*wd), dtype=torch.uint8, device=d) if want_u8 else None
        if ref_u8 is not None and sse is None:
            sse = torch.zeros(n, dtype=torch.int64, device=d)
        _check(_native.hip().eae_hip_decode(self._handle, _p(quantized_latents), n, h, wd, _p(out_f32), _p(out_u8), _p(ref_u8), _p(sse),
                                            _p(scratch), nbytes, _stream(quantized_latents)), 'eae_hip_decode')
        self._track(scratch)
        return out_f32, out_u8, sse


def conv9x9s4_u8(x_u8, w_packed, bias, gamma_packed=None, beta=None, out=None):
    """conv_1 + bias_add (+ gdn_1). x_u8: uint8 [N,H,W] or [N,H,W,1] -> f32 [N,H/4,W/4,128].
    w_packed from `pack_conv9x9s4_weights`, gamma_packed from `pack_gamma`."""
    if x_u8.dtype != torch.uint8:
        raise TypeError('`x_u8.dtype` is not `torch.uint8`.')
    (n, h, wd) = x_u8.shape[:3]
    if out is None:
        out = torch.empty((n, h//4, wd//4, NB_MAPS), dtype=torch.float32, device=x_u8.device)
    _check(_native.hip().eae_hip_conv9x9s4_u8(_p(x_u8), _p(w_packed), _p(bias), _p(gamma_packed), _p(beta), _p(out), n, h, wd, _stream(x_u8)),
           'eae_hip_conv9x9s4_u8')
    return out


def conv_workspace(device):
    """Zeroed scratch that lets conv5x5s2 / tconv5x5s2 cut the last tiles of a launch (include/eae_hip.h): every launch leaves
    it zeroed, so one allocation serves any number of launches that cannot overlap each other (one per stream / batch slot).
    Whoever passes one to a launch owns its error word: `conv_workspace_collect` behind the launches, and a look at the word
    before their outputs are trusted (codec.BatchCodec does both per batch)."""
    return torch.zeros(int(_native.hip().eae_hip_conv_workspace_bytes())//4, dtype=torch.int32, device=device)


class SplitHandOffTimeout(HipError):
    """A cut conv launch left tiles unfinished (include/eae_hip.h: eae_hip_conv_workspace_collect): its outputs are invalid."""


def conv_workspace_collect(workspace, error_word):
    """Stream-ordered: adds the number of tiles the launches on `workspace` left unfinished since the last collect to
    `error_word` (int32 device tensor of one element, or a pinned host one) and restores the all-zero workspace."""
    if error_word.numel() != 1 or error_word.element_size() != 4:
        raise HipError('`error_word` must be one 32-bit word')
    word = error_word.data_ptr() if not error_word.is_cuda else _p(error_word)
    _check(_native.hip().eae_hip_conv_workspace_collect(_p(workspace), word, _stream(workspace)), 'eae_hip_conv_workspace_collect')


def conv5x5s2(x, w_packed, bias, norm=NORM_NONE, gamma_packed=None, beta=None, out=None, workspace=None):
    """conv_2 / conv_3 + bias_add (+ gdn). workspace: a `conv_workspace` (the launch may cut its last tiles: same bits, no
    partly empty last round; the caller collects its error word, see `conv_workspace_collect`); None / False -> the entry
    point without a workspace (whole tiles only, nothing to collect)."""
    (n, h, wd, c) = x.shape
    if out is None:
        out = torch.empty((n, h//2, wd//2, NB_MAPS), dtype=torch.float32, device=x.device)
    if workspace is None or workspace is False:
        _check(_native.hip().eae_hip_conv5x5s2(_p(x), _p(w_packed), _p(bias), norm, _p(gamma_packed), _p(beta), _p(out), n, h, wd, _stream(x)),
               'eae_hip_conv5x5s2')
        return out
    _check(_native.hip().eae_hip_conv5x5s2_ws(_p(x), _p(w_packed), _p(bias), norm, _p(gamma_packed), _p(beta), _p(out), n, h, wd,
                                                  _p(workspace), _stream(x)), 'eae_hip_conv5x5s2_ws')
    return out


def gdn(x, gamma_packed, beta, inverse=False, out=None):
    if out is None:
        out = torch.empty_like(x)
    rows = x.numel()//NB_MAPS
    _check(_native.hip().eae_hip_gdn(_p(x), _p(gamma_packed), _p(beta), 1 if inverse else 0, _p(out), rows, _stream(x)), 'eae_hip_gdn')
    return out


def tconv5x5s2(x, w_packed, bias, norm=NORM_NONE, gamma_packed=None, beta=None, out=None, workspace=None):
    """transpose_conv_1 / _2 + bias_add (+ inverse gdn); `workspace` as in conv5x5s2."""
    (n, h, wd, c) = x.shape
    if out is None:
        out = torch.empty((n, 2*h, 2*wd, NB_MAPS), dtype=torch.float32, device=x.device)
    if workspace is None or workspace is False:
        _check(_native.hip().eae_hip_tconv5x5s2(_p(x), _p(w_packed), _p(bias), norm, _p(gamma_packed), _p(beta), _p(out), n, h, wd, _stream(x)),
               'eae_hip_tconv5x5s2')
        return out
    _check(_native.hip().eae_hip_tconv5x5s2_ws(_p(x), _p(w_packed), _p(bias), norm, _p(gamma_packed), _p(beta), _p(out), n, h, wd,
                                                   _p(workspace), _stream(x)), 'eae_hip_tconv5x5s2_ws')
    return out


def tconv9x9s4_luma(x, w_phase, want_f32=False, want_u8=True, ref_u8=None, sse=None):
    """transpose_conv_3 (+ cast_bt601, + squared error vs ref_u8). Returns (f32 or None, u8 or None, sse or None)."""
    (n, h, wd, c) = x.shape
    out_f32 = torch.empty((n, 4*h, 4*wd), dtype=torch.float32, device=x.device) if want_f32 else None
    out_u8 = torch.empty((n, 4*h, 4*wd), dtype=torch.uint8, device=x.device) if want_u8 else None
    if ref_u8 is not None and sse is None:
        sse = torch.zeros(n, dtype=torch.int64, device=x.device)
    _check(_native.hip().eae_hip_tconv9x9s4_luma(_p(x), _p(w_phase), _p(out_f32), _p(out_u8), _p(ref_u8), _p(sse), n, h, wd, _stream(x)),
           'eae_hip_tconv9x9s4_luma')
    return out_f32, out_u8, sse


def pack_conv_weights(w_hwio):
    """HWIO [k,k,128,128] -> kernel layout [k*k,128,packed out] (include/eae_hip.h)."""
    (k, k2, ci, co) = w_hwio.shape
    out = torch.empty((k*k2, ci, co), dtype=torch.float32, device=w_hwio.device)
    _check(_native.hip().eae_hip_pack_conv_weights(_p(w_hwio), _p(out), k*k2, _stream()), 'eae_hip_pack_conv_weights')
    return out


def pack_conv9x9s4_weights(w_tf):
    """conv_1 filter [9,9,1,128] -> kernel layout [82, packed out] (row 81 is zero)."""
    out = torch.empty((82, NB_MAPS), dtype=torch.float32, device=w_tf.device)
    _check(_native.hip().eae_hip_pack_conv9x9s4_weights(_p(w_tf), _p(out), _stream()), 'eae_hip_pack_conv9x9s4_weights')
    return out


def pack_tconv_weights(w_tf):
    """TF conv2d_transpose filter [k,k,out,in] -> kernel layout [k*k,in,packed out]."""
    (k, k2, co, ci) = w_tf.shape
    out = torch.empty((k*k2, ci, co), dtype=torch.float32, device=w_tf.device)
    _check(_native.hip().eae_hip_pack_tconv_weights(_p(w_tf), _p(out), k*k2, _stream()), 'eae_hip_pack_tconv_weights')
    return out


def pack_gamma(gamma):
    """gamma [128,128] -> [128, packed c]."""
    out = torch.empty_like(gamma)
    _check(_native.hip().eae_hip_pack_gamma(_p(gamma), _p(out), _stream()), 'eae_hip_pack_gamma')
    return out


def packed_channel_order():
    """numpy index array `perm` with packed[..., perm[c]] = plain[..., c] (c -> (c % 32)*4 + c//32)."""
    import numpy
    c = numpy.arange(NB_MAPS)
    return (c % 32)*4 + c//32


def pack_tconv9x9s4_weights(w_tf):
    """[9,9,1,128] -> [9,128,16]."""
    out = torch.empty((9, NB_MAPS, 16), dtype=torch.float32, device=w_tf.device)
    _check(_native.hip().eae_hip_pack_tconv9x9s4_weights(_p(w_tf), _p(out), _stream()), 'eae_hip_pack_tconv9x9s4_weights')
    return out


def quantize_maps(y, bin_widths, map_mean=None, want_cq=False, want_shifted=False, want_symbols=False, want_flags=False,
                  out_symbols=None, out_flags=None, out_checks=None):
    """One pass over y [N,h,w,C] (or [N,hw,C]); see include/eae_hip.h. Returns a dict of the requested device tensors.

    `checks` (int32 [3]) always comes back: [0] int16 range violations, [1] "quantization was omitted" count,
    [2] "lossless compression altered the data" count. `out_symbols`: a preallocated int16 [N, C, hw] buffer to write
    the planar symbols into (a caller that hands them to another stream keeps them out of the caching allocator).
    """
    n = y.shape[0]
    c = y.shape[-1]
    hw = y.numel()//(n*c)
    d = y.device
    cq = torch.empty_like(y) if want_cq else None
    shifted = torch.empty_like(y) if want_shifted else None
    symbols = (out_symbols if out_symbols is not None else torch.empty((n, c, hw), dtype=torch.int16, device=d)) if want_symbols else None
    if symbols is not None and (symbols.dtype != torch.int16 or symbols.numel() != n*c*hw):
        raise TypeError('`out_symbols` must be an int16 tensor of N x C x hw elements.')
    # out_flags / out_checks: preallocated int32 buffers ALREADY ZEROED by the caller (the kernel only sets / adds)
    flags = (out_flags if out_flags is not None else torch.zeros((n, c), dtype=torch.int32, device=d)) if want_flags else None
    checks = out_checks if out_checks is not None else torch.zeros(3, dtype=torch.int32, device=d)
    _check(_native.hip().eae_hip_quantize_maps(_p(y), _p(map_mean), _p(bin_widths), _p(cq), _p(shifted), _p(symbols), _p(flags),
                                               _p(checks), n, hw, c, _stream(y)), 'eae_hip_quantize_maps')
    return {'cq': cq, 'shifted': shifted, 'symbols': symbols, 'nonzero_flags': flags, 'checks': checks}


def latent_stage(x, bin_widths, map_mean=None, gdn_in=None, igdn_out=None, want_y=False, want_shifted=False, want_symbols=True,
                 want_flags=False, out_symbols=None, out_flags=None, out_checks=None):
    """conv_3 output x [N,h,w,128] (bias added, not normalised) -> gdn_3 -> quantiser -> (+mean) inverse_gdn_4, one kernel.
    gdn_in / igdn_out: (gamma_packed, beta) pairs or None. Returns a dict like quantize_maps plus 'y' and 't' (the input of
    transpose_conv_1). out_* buffers: preallocated, flags / checks already zeroed (see quantize_maps)."""
    n = x.shape[0]
    c = x.shape[-1]
    hw = x.numel()//(n*c)
    d = x.device
    y = torch.empty_like(x) if want_y else None
    shifted = torch.empty_like(x) if want_shifted else None
    t = torch.empty_like(x) if igdn_out is not None else None
    symbols = (out_symbols if out_symbols is not None else torch.empty((n, c, hw), dtype=torch.int16, device=d)) if want_symbols else None
    flags = (out_flags if out_flags is not None else torch.zeros((n, c), dtype=torch.int32, device=d)) if want_flags else None
    checks = out_checks if out_checks is not None else torch.zeros(3, dtype=torch.int32, device=d)
    (g_in, b_in) = gdn_in if gdn_in is not None else (None, None)
    (g_out, b_out) = igdn_out if igdn_out is not None else (None, None)
    _check(_native.hip().eae_hip_latent_stage(_p(x), _p(g_in), _p(b_in), _p(map_mean), _p(bin_widths), _p(g_out), _p(b_out), _p(y),
                                              _p(shifted), _p(t), _p(symbols), _p(flags), _p(checks), n, hw, _stream(x)),
           'eae_hip_latent_stage')
    return {'y': y, 'shifted': shifted, 't': t, 'symbols': symbols, 'nonzero_flags': flags, 'checks': checks}


def conv5x5s2_latent(x, w_packed, bias, bin_widths, map_mean=None, gdn_in=None, igdn_out=None, want_y=False, want_shifted=False,
                     want_flags=False, out_symbols=None, out_flags=None, out_checks=None, workspace=None):
    """conv_3 + bias followed by `latent_stage` in one launch (include/eae_hip.h: eae_hip_conv5x5s2_latent): x is the gdn_2
    output [N,h,w,128]; returns latent_stage's dict ('t' for the fixed-bin-width model, 'shifted' for the learned one is the
    synthesis transform's input). Same bits as conv5x5s2(..., NORM_NONE) + latent_stage."""
    (n, h, wd, c) = x.shape
    d = x.device
    hw = (h//2)*(wd//2)
    fixed = gdn_in is not None
    if fixed != (igdn_out is not None):
        raise ValueError('`gdn_in` and `igdn_out` go together (the fixed-bin-width model has both, the learned one neither).')
    shape = (n, h//2, wd//2, c)
    y = torch.empty(shape, dtype=torch.float32, device=d) if want_y else None
    shifted = torch.empty(shape, dtype=torch.float32, device=d) if (want_shifted or not fixed) else None
    t = torch.empty(shape, dtype=torch.float32, device=d) if fixed else None
    symbols = out_symbols if out_symbols is not None else torch.empty((n, c, hw), dtype=torch.int16, device=d)
    flags = (out_flags if out_flags is not None else torch.zeros((n, c), dtype=torch.int32, device=d)) if want_flags else None
    checks = out_checks if out_checks is not None else torch.zeros(3, dtype=torch.int32, device=d)
    (g_in, b_in) = gdn_in if fixed else (None, None)
    (g_out, b_out) = igdn_out if fixed else (None, None)
    if workspace is None:
        workspace = False          # never cut without a workspace whose owner collects its error word
    _check(_native.hip().eae_hip_conv5x5s2_latent(_p(x), _p(w_packed), _p(bias), _p(g_in), _p(b_in), _p(map_mean), _p(bin_widths), _p(g_out),
                                                  _p(b_out), _p(y), _p(shifted), _p(t), _p(symbols), _p(flags), _p(checks), n, h, wd,
                                                  _p(workspace) if workspace is not False else None, _stream(x)),
           'eae_hip_conv5x5s2_latent')
    return {'y': y, 'shifted': shifted, 't': t, 'symbols': symbols, 'nonzero_flags': flags, 'checks': checks}


def map_means(y):
    """float32 per-map means over every other axis of y [..., C]: `numpy.mean(y, axis=(0, 1, 2))` of lossless/stats.py:306 bit
    for bit (float32 accumulator per map, rows ascending, then / float32(rows))."""
    c = y.shape[-1]
    rows = y.numel()//c
    means = torch.empty(c, dtype=torch.float32, device=y.device)
    _check(_native.hip().eae_hip_map_means(_p(y), _p(means), rows, c, _stream(y)), 'eae_hip_map_means')
    return means


def map_minmax(y):
    """Per-map minimum and maximum over every other axis of y [..., C]: float32 [2, C]."""
    c = y.shape[-1]
    out = torch.empty((2, c), dtype=torch.float32, device=y.device)
    keys = torch.empty(2*c, dtype=torch.int32, device=y.device)
    _check(_native.hip().eae_hip_map_minmax(_p(y), _p(out), _p(keys), y.numel()//c, c, _stream()), 'eae_hip_map_minmax')
    return out


def floor_histograms(y, radius):
    """Per-map histogram of floor(y) for y [..., C]: (hist int32 [C, 2*radius+1] with bin floor(y)+radius, overflow int32 [C])."""
    c = y.shape[-1]
    hist = torch.zeros((c, 2*radius + 1), dtype=torch.int32, device=y.device)
    overflow = torch.zeros(c, dtype=torch.int32, device=y.device)
    _check(_native.hip().eae_hip_floor_histograms(_p(y), _p(hist), radius, _p(overflow), y.numel()//c, c, _stream()),
           'eae_hip_floor_histograms')
    return hist, overflow


def nonzero_flags(x):
    """x [N, ..., C] -> int32 [N, C], 1 where map (n, c) has a non-zero element."""
    n = x.shape[0]
    c = x.shape[-1]
    flags = torch.zeros((n, c), dtype=torch.int32, device=x.device)
    _check(_native.hip().eae_hip_nonzero_flags(_p(x), _p(flags), n, x.numel()//(n*c), c, _stream()), 'eae_hip_nonzero_flags')
    return flags


def cast_int16(x):
    """int16(round_half_even(x)) and the count of out-of-range elements (int32 [1])."""
    out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    range_error = torch.zeros(1, dtype=torch.int32, device=x.device)
    _check(_native.hip().eae_hip_cast_int16(_p(x), _p(out), x.numel(), _p(range_error), _stream()), 'eae_hip_cast_int16')
    return out, range_error


def symbol_histograms(symbols_planar, radius, out=None, first_map=0, map_step=1, zero=True):
    """symbols [..., map_size] int16 -> (hist int32 [n_maps, 2*radius+1], overflow int32 [n_maps]); `out` = that pair,
    preallocated (zeroed here unless zero=False: the kernel accumulates). first_map / map_step: only every map_step-th
    map starting at first_map (e.g. the exception map of every image of a batch)."""
    map_size = symbols_planar.shape[-1]
    n_maps = (symbols_planar.numel()//map_size - first_map + map_step - 1)//map_step
    if out is None:
        hist = torch.zeros((n_maps, 2*radius + 1), dtype=torch.int32, device=symbols_planar.device)
        overflow = torch.zeros(n_maps, dtype=torch.int32, device=symbols_planar.device)
    else:
        (hist, overflow) = out
        if zero:
            hist.zero_()
            overflow.zero_()
    _check(_native.hip().eae_hip_symbol_histograms_strided(_p(symbols_planar), _p(hist), radius, _p(overflow), n_maps, map_size,
                                                           first_map, map_step, _stream(symbols_planar)), 'eae_hip_symbol_histograms_strided')
    return hist, overflow


def cast_bt601(x):
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _check(_native.hip().eae_hip_cast_bt601(_p(x), _p(out), x.numel(), _stream()), 'eae_hip_cast_bt601')
    return out


def rgb_to_ycbcr(rgb_uint8, want_ycbcr=True, want_luma=False):
    """uint8 [..., 3] RGB -> (ycbcr uint8 [..., 3] or None, luma uint8 [...] or None), ITU-R BT.601 (tools.py:1019-1083)."""
    count = rgb_uint8.numel()//3
    ycbcr = torch.empty_like(rgb_uint8) if want_ycbcr else None
    luma = torch.empty(rgb_uint8.shape[:-1], dtype=torch.uint8, device=rgb_uint8.device) if want_luma else None
    _check(_native.hip().eae_hip_rgb_to_ycbcr(_p(rgb_uint8), _p(ycbcr), _p(luma), count, _stream()), 'eae_hip_rgb_to_ycbcr')
    return ycbcr, luma


def sse_u8(a, b):
    """Per-image sum of squared differences of two uint8 stacks [N, ...] -> int64 [N]."""
    n = a.shape[0]
    sse = torch.zeros(n, dtype=torch.int64, device=a.device)
    _check(_native.hip().eae_hip_sse_u8(_p(a), _p(b), _p(sse), n, a.numel()//n, _stream()), 'eae_hip_sse_u8')
    return sse


# ---- lossless coder on the device (include/eae_hip.h, "lossless coder on the device") ---------------------------------

CODER_ROUNDTRIP, CODER_ENCODE_ONLY, CODER_ROUNDTRIP_VERIFY = 0, 1, 2


class CoderStreams(object):
    """Per-map streams of one batch, resident in HBM, in the layout of eae_coder_encode_maps (include/eae_coder.h):
    map m owns `streams[m]` (stride bytes): BAC bytes at +0, bypass bytes at +stride/2."""

    def __init__(self, n_maps, map_size, truncated_unary_length, device, results=None):
        self.n_maps = n_maps
        self.map_size = map_size
        self.truncated_unary_length = truncated_unary_length
        self.stride = int(_native.hip().eae_hip_coder_stream_stride_bytes(map_size, truncated_unary_length))
        self.streams = torch.empty((n_maps, self.stride), dtype=torch.uint8, device=device)
        # one allocation so that a caller can fetch all four per-map results with a single device -> host copy
        self.results = results if results is not None else torch.zeros((4, n_maps), dtype=torch.int32, device=device)
        if self.results.shape != (4, n_maps) or self.results.dtype != torch.int32 or not self.results.is_contiguous():
            raise TypeError('`results` must be a contiguous int32 tensor of shape (4, n_maps).')
        (self.bac_bits, self.bypass_bits, self.status, self.stage) = self.results.unbind(0)

    def nb_bits(self):
        return self.bac_bits + self.bypass_bits


def coder_compress_maps(symbols_planar, probabilities, prob_row, truncated_unary_length, mode=CODER_ROUNDTRIP_VERIFY,
                        out=None, lanes_per_wave=0):
    """symbols [..., map_size] int16 (device), probabilities [rows, L] float64 (device), prob_row int32 [n_maps] or None
    -> (CoderStreams, reconstruction or None). Asynchronous: per-map errors are in `streams.status`."""
    map_size = symbols_planar.shape[-1]
    n_maps = symbols_planar.numel()//map_size
    if symbols_planar.dtype != torch.int16 or probabilities.dtype != torch.float64:
        raise TypeError('`symbols_planar` must be int16 and `probabilities` float64.')
    if out is None:
        out = CoderStreams(n_maps, map_size, truncated_unary_length, symbols_planar.device)
    reconstruction = torch.empty_like(symbols_planar) if mode == CODER_ROUNDTRIP else None
    _check(_native.hip().eae_hip_coder_compress_maps(n_maps, map_size, _p(symbols_planar), _p(reconstruction) if reconstruction is not None else None,
                                                     truncated_unary_length, _p(probabilities), _p(prob_row) if prob_row is not None else None,
                                                     _p(out.streams), out.stride, _p(out.bac_bits), _p(out.bypass_bits), _p(out.status),
                                                     _p(out.stage), mode, lanes_per_wave, _stream(symbols_planar)), 'eae_hip_coder_compress_maps')
    return out, reconstruction


def coder_decode_maps(streams, probabilities, prob_row, lanes_per_wave=0):
    """CoderStreams -> int16 [n_maps, map_size] (device); maps with prob_row < 0 are left untouched (zeros)."""
    out = torch.zeros((streams.n_maps, streams.map_size), dtype=torch.int16, device=streams.streams.device)
    _check(_native.hip().eae_hip_coder_decode_maps(streams.n_maps, streams.map_size, _p(out), streams.truncated_unary_length,
                                                   _p(probabilities), _p(prob_row) if prob_row is not None else None,
                                                   _p(streams.streams), streams.stride, _p(streams.bac_bits), _p(streams.bypass_bits),
                                                   _p(streams.status), _p(streams.stage), lanes_per_wave, _stream()),
           'eae_hip_coder_decode_maps')
    return out


def coder_verify_maps(streams, expected_symbols, probabilities, prob_row, lanes_per_wave=0):
    """Decodes every map of `streams` and compares with `expected_symbols` on the device; failures land in
    `streams.status` (6 = roundtrip mismatch)."""
    if expected_symbols.dtype != torch.int16 or expected_symbols.numel() != streams.n_maps*streams.map_size:
        raise TypeError('`expected_symbols` must hold n_maps x map_size int16 symbols.')
    _check(_native.hip().eae_hip_coder_verify_maps(streams.n_maps, streams.map_size, _p(expected_symbols), streams.truncated_unary_length,
                                                   _p(probabilities), _p(prob_row), _p(streams.streams), streams.stride,
                                                   _p(streams.bac_bits), _p(streams.bypass_bits), _p(streams.status), _p(streams.stage),
                                                   lanes_per_wave, _stream()), 'eae_hip_coder_verify_maps')


def publish_to_host(src_device, dst_pinned):
    """Stream-ordered copy of a small device tensor into a pinned host tensor of the same byte size, done by a kernel
    (never blocks the calling thread). Synchronise an event recorded after it before reading `dst_pinned`."""
    nbytes = src_device.numel()*src_device.element_size()
    if not dst_pinned.is_pinned() or dst_pinned.numel()*dst_pinned.element_size() != nbytes or not dst_pinned.is_contiguous():
        raise HipError('expected a contiguous pinned host tensor of the same size')
    _check(_native.hip().eae_hip_publish_to_host(_p(src_device), dst_pinned.data_ptr(), nbytes, _stream()), 'eae_hip_publish_to_host')


def publish_sequence(counter_device, word_pinned):
    """Stream-ordered: increments `counter_device` (int32 device tensor of one element) and leaves the new value in `word_pinned`
    (int32 pinned host tensor of one element), behind everything the current stream has done so far (include/eae_hip.h)."""
    if counter_device.numel() != 1 or counter_device.element_size() != 4 or word_pinned.numel() != 1 or word_pinned.element_size() != 4:
        raise HipError('expected two 32-bit words')
    if not word_pinned.is_pinned():
        raise HipError('expected a pinned host word')
    _check(_native.hip().eae_hip_publish_sequence(_p(counter_device), word_pinned.data_ptr(), _stream(counter_device)), 'eae_hip_publish_sequence')


def publish_step(src_device, dst_pinned, clear_from, tickets_device, counter_device, word_pinned, conv_ws=None, error_word=None):
    """One launch for the end of a step's side (include/eae_hip.h: eae_hip_publish_step): [`conv_workspace_collect(conv_ws,
    error_word)`] -> `publish_to_host(src_device, dst_pinned)` -> `src_device` zeroed from element `clear_from` on (its accumulators,
    for the step that uses it next) -> `publish_sequence(counter_device, word_pinned)`. `tickets_device`: a zeroed int32 on the device."""
    nbytes = src_device.numel()*src_device.element_size()
    if not dst_pinned.is_pinned() or dst_pinned.numel()*dst_pinned.element_size() != nbytes or not dst_pinned.is_contiguous() or not word_pinned.is_pinned():
        raise HipError('expected contiguous pinned host tensors, the first of the size of the source')
    if any(t.numel() != 1 or t.element_size() != 4 for t in (tickets_device, counter_device, word_pinned)):
        raise HipError('expected three 32-bit words')
    _check(_native.hip().eae_hip_publish_step(_p(src_device), dst_pinned.data_ptr(), nbytes, int(clear_from)*src_device.element_size(),
                                              _p(conv_ws), _p(error_word), _p(tickets_device), _p(counter_device),
                                              word_pinned.data_ptr(), _stream(src_device)), 'eae_hip_publish_step')


def coder_workspace(n_maps, map_size, truncated_unary_length, device):
    """Scratch for coder_encode_batch / coder_decode_batch (one per batch in flight)."""
    nbytes = int(_native.hip().eae_hip_coder_workspace_bytes(n_maps, map_size, truncated_unary_length))
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def coder_encode_batch(symbols_planar, probabilities, prob_row, truncated_unary_length, out=None, workspace=None):
    """The 64-maps-per-wavefront encoder (include/eae_hip.h). Same results as coder_compress_maps(mode=CODER_ENCODE_ONLY)."""
    map_size = symbols_planar.shape[-1]
    n_maps = symbols_planar.numel()//map_size
    if symbols_planar.dtype != torch.int16 or probabilities.dtype != torch.float64:
        raise TypeError('`symbols_planar` must be int16 and `probabilities` float64.')
    if out is None:
        out = CoderStreams(n_maps, map_size, truncated_unary_length, symbols_planar.device)
    if workspace is None:
        workspace = coder_workspace(n_maps, map_size, truncated_unary_length, symbols_planar.device)
    _check(_native.hip().eae_hip_coder_encode_batch(n_maps, map_size, _p(symbols_planar), truncated_unary_length, _p(probabilities),
                                                    _p(prob_row), _p(out.streams), out.stride, _p(out.bac_bits), _p(out.bypass_bits),
                                                    _p(out.status), _p(out.stage), _p(workspace), workspace.numel(), _stream(symbols_planar)),
           'eae_hip_coder_encode_batch')
    return out


def coder_decode_batch(streams, probabilities, prob_row, expected=None, workspace=None):
    """The 64-maps-per-wavefront decoder. expected=None: returns the decoded symbols [n_maps, map_size] (skipped maps
    zero). With `expected`: decodes into the workspace and compares on the device; failures land in `streams.status`.
    With a workspace (`coder_workspace`; always there with `expected`) the serial core leaves one prefix byte per symbol there
    and a data-parallel pass adds signs and suffixes (csrc/hip/coder_simd.hip); without, the general kernel decodes every
    map. Same results."""
    device = streams.streams.device
    if expected is None:
        out = torch.zeros((streams.n_maps, streams.map_size), dtype=torch.int16, device=device)
    else:
        out = None
        if expected.dtype != torch.int16 or expected.numel() != streams.n_maps*streams.map_size:
            raise TypeError('`expected` must hold n_maps x map_size int16 symbols.')
        if workspace is None:
            workspace = coder_workspace(streams.n_maps, streams.map_size, streams.truncated_unary_length, device)
    _check(_native.hip().eae_hip_coder_decode_batch(streams.n_maps, streams.map_size, _p(out), _p(expected), streams.truncated_unary_length,
                                                    _p(probabilities), _p(prob_row), _p(streams.streams), streams.stride,
                                                    _p(streams.bac_bits), _p(streams.bypass_bits), _p(streams.status), _p(streams.stage),
                                                    _p(workspace), workspace.numel() if workspace is not None else 0, _stream(streams.streams)),
           'eae_hip_coder_decode_batch')
    return out


class ExperimentalCoderMissing(HipError):
    """The library in use was built without -DEAE_EXPERIMENTAL_CODER (the product library always is: include/eae_hip.h)."""


def _experimental_coder():
    lib = _native.hip()
    if not hasattr(lib, 'eae_hip_coder_roundtrip_trailing'):
        raise ExperimentalCoderMissing(
            'the chunked / fused coder round trips are experimental and not part of lib/libeae_hip.so (both measured slower than '
            'encode_batch + decode_batch: DESIGN.md section 5); they live in lib/libeae_hip_test.so (EAE_HIP_LIB=test, or '
            '`make -C csrc EXTRA_HIPFLAGS=-DEAE_EXPERIMENTAL_CODER`)')
    return lib


def coder_trailing_workspace(n_maps, map_size, truncated_unary_length, device):
    """Scratch for coder_roundtrip_trailing (one per batch in flight). Experimental build only."""
    nbytes = int(_experimental_coder().eae_hip_coder_trailing_workspace_bytes(n_maps, map_size, truncated_unary_length))
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def coder_roundtrip_trailing(symbols_planar, probabilities, prob_row, truncated_unary_length, chunks=4, out=None, workspace=None):
    """Encode every map, decode it back, compare (coder_encode_batch + coder_decode_batch(expected=symbols)) with the serial
    chains cut into `chunks` launches so that emit pass and decoder trail the encoder core (include/eae_hip.h: for one or two
    images). Same streams, bit counts, statuses and stages. Returns the CoderStreams. Experimental build only."""
    map_size = symbols_planar.shape[-1]
    n_maps = symbols_planar.numel()//map_size
    if symbols_planar.dtype != torch.int16 or probabilities.dtype != torch.float64:
        raise TypeError('`symbols_planar` must be int16 and `probabilities` float64.')
    if out is None:
        out = CoderStreams(n_maps, map_size, truncated_unary_length, symbols_planar.device)
    if workspace is None:
        workspace = coder_trailing_workspace(n_maps, map_size, truncated_unary_length, symbols_planar.device)
    _check(_experimental_coder().eae_hip_coder_roundtrip_trailing(n_maps, map_size, _p(symbols_planar), truncated_unary_length, _p(probabilities),
                                                          _p(prob_row), _p(out.streams), out.stride, _p(out.bac_bits), _p(out.bypass_bits),
                                                          _p(out.status), _p(out.stage), _p(workspace), workspace.numel(), int(chunks),
                                                          _stream(symbols_planar)), 'eae_hip_coder_roundtrip_trailing')
    return out


def coder_roundtrip_fused(symbols_planar, probabilities, prob_row, truncated_unary_length, out=None, workspace=None):
    """Encode every map, decode it back, compare, with the three serial stages of a group of 64 maps as three wavefronts of one
    workgroup handing records and stream words to each other through LDS (include/eae_hip.h: eae_hip_coder_roundtrip_fused). Same
    streams, bit counts, statuses and stages as coder_encode_batch + coder_decode_batch(expected=symbols). Returns the CoderStreams.
    Experimental build only."""
    map_size = symbols_planar.shape[-1]
    n_maps = symbols_planar.numel()//map_size
    if symbols_planar.dtype != torch.int16 or probabilities.dtype != torch.float64:
        raise TypeError('`symbols_planar` must be int16 and `probabilities` float64.')
    if out is None:
        out = CoderStreams(n_maps, map_size, truncated_unary_length, symbols_planar.device)
    if workspace is None:
        workspace = coder_trailing_workspace(n_maps, map_size, truncated_unary_length, symbols_planar.device)
    _check(_experimental_coder().eae_hip_coder_roundtrip_fused(n_maps, map_size, _p(symbols_planar), truncated_unary_length, _p(probabilities),
                                                       _p(prob_row), _p(out.streams), out.stride, _p(out.bac_bits), _p(out.bypass_bits),
                                                       _p(out.status), _p(out.stage), _p(workspace), workspace.numel(), _stream(symbols_planar)),
           'eae_hip_coder_roundtrip_fused')
    return out


def coder_pack_streams(streams, offsets, payload_bytes):
    """Gathers the valid stream bytes of every map into one uint8 device tensor; `offsets` int64 [n_maps, 2] (device)."""
    payload = torch.zeros(max(int(payload_bytes), 1), dtype=torch.uint8, device=streams.streams.device)
    _check(_native.hip().eae_hip_coder_pack_streams(streams.n_maps, _p(streams.streams), streams.stride, _p(streams.bac_bits),
                                                    _p(streams.bypass_bits), _p(offsets), _p(payload), _stream()),
           'eae_hip_coder_pack_streams')
    return payload


def coder_unpack_streams(payload, offsets, bac_bits, bypass_bits, map_size, truncated_unary_length):
    """The inverse: a CoderStreams whose regions hold the bytes of `payload` (bit counts int32 [n_maps], device)."""
    n_maps = bac_bits.numel()
    streams = CoderStreams(n_maps, map_size, truncated_unary_length, payload.device)
    streams.bac_bits.copy_(bac_bits)
    streams.bypass_bits.copy_(bypass_bits)
    _check(_native.hip().eae_hip_coder_unpack_streams(n_maps, _p(payload), _p(offsets), _p(streams.bac_bits), _p(streams.bypass_bits),
                                                      _p(streams.streams), streams.stride, _stream()), 'eae_hip_coder_unpack_streams')
    return streams


def dequantize_maps(symbols_planar, bin_widths, map_mean=None, want_cq=False, want_shifted=True):
    """int16 symbols [N, 128, hw] -> float32 [N, hw, 128]: bw * symbol (and + map_mean), the arrays quantize_maps produced."""
    (n, c, hw) = symbols_planar.shape
    cq = torch.empty((n, hw, c), dtype=torch.float32, device=symbols_planar.device) if want_cq else None
    shifted = torch.empty((n, hw, c), dtype=torch.float32, device=symbols_planar.device) if want_shifted else None
    _check(_native.hip().eae_hip_dequantize_maps(_p(symbols_planar), _p(bin_widths), _p(map_mean), _p(cq), _p(shifted), n, hw, c, _stream(symbols_planar)),
           'eae_hip_dequantize_maps')
    return {'cq': cq, 'shifted': shifted}


# ---- SVHN float64 path (include/eae_hip.h, "SVHN path") -------------------------------------------------------------

def svhn_dense(x, w, b, leaky_relu):
    """act(x . w + b) in float64; x [n,k], w [k,m], b [m] or [1,m]."""
    (n, k) = x.shape
    m = w.shape[1]
    out = torch.empty((n, m), dtype=torch.float64, device=x.device)
    _check(_native.hip().eae_hip_svhn_dense_f64(_p(x), _p(w), _p(b), _p(out), n, k, m, 1 if leaky_relu else 0, _stream()),
           'eae_hip_svhn_dense_f64')
    return out


def svhn_preprocess(images_u8, mean, std_training):
    (n, d) = images_u8.shape
    out = torch.empty((n, d), dtype=torch.float64, device=images_u8.device)
    _check(_native.hip().eae_hip_svhn_preprocess(_p(images_u8), _p(mean), float(std_training), _p(out), n, d, _stream()),
           'eae_hip_svhn_preprocess')
    return out


def svhn_quantize(y, bin_width, want_q=True, want_symbols=False):
    q = torch.empty_like(y) if want_q else None
    symbols = torch.empty(y.shape, dtype=torch.int32, device=y.device) if want_symbols else None
    checks = torch.zeros(2, dtype=torch.int32, device=y.device)
    _check(_native.hip().eae_hip_svhn_quantize_f64(_p(y), float(bin_width), _p(q), _p(symbols), _p(checks), y.numel(), _stream()),
           'eae_hip_svhn_quantize_f64')
    return q, symbols, checks


def svhn_symbol_histogram(symbols):
    """Exact histogram of int32 symbols from their minimum to their maximum: (hist int64 numpy, minimum)."""
    minmax = torch.tensor([2147483647, -2147483648], dtype=torch.int32, device=symbols.device)
    _check(_native.hip().eae_hip_svhn_symbol_range(_p(symbols), symbols.numel(), _p(minmax), _stream()), 'eae_hip_svhn_symbol_range')
    (lo, hi) = minmax.cpu().tolist()
    nb = hi - lo + 1
    hist = torch.zeros(nb, dtype=torch.int32, device=symbols.device)
    overflow = torch.zeros(1, dtype=torch.int32, device=symbols.device)
    _check(_native.hip().eae_hip_svhn_symbol_histogram(_p(symbols), symbols.numel(), lo, nb, _p(hist), _p(overflow), _stream()),
           'eae_hip_svhn_symbol_histogram')
    if int(overflow.item()) != 0:
        raise HipError('symbol outside [min, max]')
    return hist.cpu().numpy().astype('int64'), lo


def svhn_postprocess(reconstruction, std_training, mean, ref_u8=None):
    (n, d) = reconstruction.shape
    out = torch.empty((n, d), dtype=torch.uint8, device=reconstruction.device)
    sse = torch.zeros(n, dtype=torch.int64, device=reconstruction.device) if ref_u8 is not None else None
    _check(_native.hip().eae_hip_svhn_postprocess(_p(reconstruction), float(std_training), _p(mean), _p(out), _p(ref_u8), _p(sse),
                                                  n, d, _stream()), 'eae_hip_svhn_postprocess')
    return out, sse
