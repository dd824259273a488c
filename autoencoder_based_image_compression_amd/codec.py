"""The whole hot path for a batch of images as one asynchronous call: what the body of `fix_gamma`
(kodak_tensorflow/reconstructing_eae_kodak.py:144-147, 170-225) does per image with `sess.run`, numpy and 127 Cython
calls, issued here as a dozen launches over arrays that stay in HBM:

    conv1+GDN1 -> conv2+GDN2 -> conv3 -> [GDN3 -> centre / quantise / int16 symbols / dead-map flags -> IGDN4]
    -> exception-map histograms
    -> (side stream) lossless coder: encode every map, decode it back, compare                 lossless/compression.py:84-154
    -> tconv1+IGDN5 -> tconv2+IGDN6 -> tconv3 + BT.601 cast + squared error against the input  tools.py:61-93, 831-881

`BatchCodec.submit` only enqueues; `Ticket.result()` returns, per image, the quantities `fix_gamma` stores: the number
of bits of the lossless code (coder bits of the 127 ordinary maps + ceil(h*w*entropy) of the exception map,
compression.py:68-81), the squared error (-> `tls.psnr_2d`), the number of dead maps (`tls.count_nb_deads`), and, on
request, the uint8 reconstruction. The values equal those of the reference-shaped functions of `kodak/` on the same
inputs (tests/test_gpu_codec.py); `bench.py` times exactly this class.

Concurrency: the coder is a few latency-bound wavefronts, so its launches go to side streams and overlap the synthesis
transforms of the same batch and the analysis transforms of the next ones; `nb_in_flight` batches of coder work may be
pending. Buffers that cross streams are preallocated per slot; results reach the host through a kernel that writes pinned
memory (no hipMemcpyAsync on the launch thread) and a worker thread turns them into per-image numbers; it polls its events
and sleeps in between instead of spinning in `synchronize()` (eight ranks share one host CPU quota). For small batches the
step can be replayed as three hipGraphs per slot (`use_graphs`) over several transform streams (`nb_transform_streams`).
"""
import contextlib
import os
import queue
import threading
import time
import weakref

import numpy
import torch

from . import device as dev
from . import pipeline
from .kodak.eae.graph import constants as csts
from .kodak.lossless import compression as lossless_compression

# HIP multiplexes streams onto 4 hardware queues: side streams are shared by every codec of the process so that a coder
# stream never ends up on the hardware queue of the stream the transforms run on. One list per device AND kind: a codec's
# coder streams are coder streams for every other codec too, whatever their `nb_in_flight`.
_SIDE_STREAMS = {}          # (device index, kind) -> list of streams
# The result worker polls its events and sleeps in between: `Event.synchronize()` was measured to spin a whole CPU per
# process (with blocking events too), and eight ranks share one 16-CPU quota. 0 restores synchronize().
_POLL_SECONDS = float(os.environ.get('EAE_WORKER_POLL_SECONDS', '0.0002'))
# How the result worker learns that a batch is through. 'sequence' (default): the last launch of the coder side and of the synthesis
# side bumps a step counter that lands in pinned host memory (device.publish_sequence); the worker, which knows how many times the
# slot has been submitted, reads that word -- no HIP call and no event recorded for it, so the worker can never disturb a stream
# capture (`_capture_all`), and a poll is a load. Measured against 'events' (round 3-4's way: one event per side per step,
# `event.query()` every 0.2 ms; profiles/r05_wait_modes.log, r05_host_cpu_threads.log): same throughput within 0.5 %, the worker's
# own CPU 0.7-0.9 ms per 3.0 ms step either way, and the 2.85 ms per step the HIP runtime's signal thread spends in
# kfd_wait_on_events (system time) does NOT move: it does not come from the host's events.
_WAIT_MODE = os.environ.get('EAE_WORKER_WAIT', 'sequence')
if _WAIT_MODE not in ('sequence', 'events'):
    raise ValueError('EAE_WORKER_WAIT={0!r}: expected "sequence" or "events"'.format(_WAIT_MODE))
# Where a batch's coder work starts. The coder's first kernel is wide (binarise: one wavefront per map, 3,048 of them for 24 Kodak
# images) and, launched the moment the symbols exist, runs exactly while transpose_conv_1 does -- the shortest of the conv GEMM
# launches, which it stretched from 0.29 to 0.34 ms. '1': the coder stream waits for transpose_conv_1 instead (its wide pass then
# falls into transpose_conv_2's 1.07 ms: tconv1 0.339 -> 0.304 ms, 0.66 -> 0.73 of peak; the step is the same within 0.1 %,
# profiles/r05_coder_behind_tconv1.log). For one or two images per batch the step IS the coder's chain and starting it a launch
# later only adds to it (one image 1.23 -> 1.37 ms): there the coder starts as soon as the symbols exist. '0' / '1' force either.
_CODER_BEHIND_TCONV1 = os.environ.get('EAE_CODER_BEHIND_TCONV1')
# (0.1 ms between polls, behind the one long sleep of `_Worker._wait_sequence`: one image at a time 1.09 -> 1.05 ms against 0.2 ms, 0.03 ms more CPU per step)
_SEQUENCE_POLL_SECONDS = float(os.environ.get('EAE_WORKER_SEQUENCE_POLL_SECONDS', '0.0001'))
_SEQUENCE_TIMEOUT_SECONDS = float(os.environ.get('EAE_WORKER_SEQUENCE_TIMEOUT_SECONDS', '60'))
_LONG_SLEEP = os.environ.get('EAE_WORKER_LONG_SLEEP', '1') != '0'      # the one long sleep in front of the polls (`_Worker._wait_sequence`)
# `Ticket.result()` called before anybody has started on the ticket's step does the waiting and the bookkeeping ITSELF (the caller is
# blocked anyway: no wake-up of the worker, no hand-over back), and around the moment the step is expected it watches the step counter
# without sleeping, for at most this long (seconds of one CPU per call; 0 = never: sleeps and polls like the worker)
_RESULT_SPIN_SECONDS = float(os.environ.get('EAE_RESULT_SPIN_SECONDS', '0.00015'))
_RESULT_BY_CALLER = os.environ.get('EAE_RESULT_BY_CALLER', '1') != '0'      # 0: results are always formed by the worker thread (rounds 1-5)
_EARLY_PUBLISH = os.environ.get('EAE_EARLY_PUBLISH', '1') != '0'            # 0: a small step's analysis-side blocks reach the host with the coder's only


def _short_sleeps_for_this_thread():
    """PR_SET_TIMERSLACK = 1 us for the calling thread (Linux rounds a sleep up by the slack, 50 us by default)."""
    try:
        import ctypes
        ctypes.CDLL(None, use_errno=True).prctl(29, 1000, 0, 0, 0)
    except Exception:
        pass


# Stream priorities (experiments: scratch/r04): EAE_TRANSFORM_STREAM_PRIORITY / EAE_CODER_STREAM_PRIORITY, torch's numbering
# (-1 high, 0 normal); unset = the runtime's default for both.
def _priority(name):
    """Parsed once at import: a malformed value fails here, not inside some constructor later."""
    text = os.environ.get(name)
    if text is None:
        return None
    try:
        return int(text)
    except ValueError:
        raise ValueError('{0}={1!r}: expected an integer stream priority (-1 high, 0 normal)'.format(name, text))


_PRIORITY = {'transform': _priority('EAE_TRANSFORM_STREAM_PRIORITY'), 'coder': _priority('EAE_CODER_STREAM_PRIORITY')}
# Codecs of this process that have not been closed, per device: a graph capture waits until the others are idle (`_capture_all`).
# Weak references: a codec dropped without close() is still collected (its __del__ closes it).
_LIVE = {}                  # device index -> weakref.WeakSet of BatchCodec
_LIVE_LOCK = threading.Lock()


def _side_streams(count, device, kind='coder'):
    """The first `count` streams of the process-wide list of the device for this kind ('coder' / 'transform')."""
    streams = _SIDE_STREAMS.setdefault((device.index, kind), [])
    while len(streams) < count:
        if _PRIORITY[kind] is not None:
            streams.append(torch.cuda.Stream(device=device, priority=_PRIORITY[kind]))
        else:
            streams.append(torch.cuda.Stream(device=device))
    return streams[:count]


def _step_streams(count, device):
    """`count` streams for `one_stream_steps`: the device's transform streams as far as they exist, then its existing coder streams,
    and only then new (transform) streams."""
    existing_transform = len(_SIDE_STREAMS.get((device.index, 'transform'), ()))
    from_coder = min(len(_SIDE_STREAMS.get((device.index, 'coder'), ())), max(0, count - existing_transform))
    return _side_streams(count - from_coder, device, kind='transform') + _side_streams(from_coder, device)


class Ticket(object):
    """Handle on one submitted batch."""

    def __init__(self, nb_images):
        self.nb_images = nb_images
        self._done = threading.Event()
        self._error = None
        self._values = None
        self.reconstruction_uint8 = None      # device tensor when the codec keeps reconstructions
        self.reconstruction_host = None       # numpy uint8 (a view of the slot's pinned buffer) with BatchCodec(fetch_reconstruction=True),
        #                                       after result(); valid until the slot comes round again (nb_slots submits later)
        self.fed_event = None                 # host input: recorded behind the host -> device copy (the caller's pinned batch may
        #                                       be rewritten once it has completed)
        self.decoded_event = None             # with keep_reconstruction: recorded behind the synthesis transform: wait for it on another
        #                                       stream before reading `reconstruction_uint8` there (valid until the slot comes round again)
        self._coder_span = None               # (start, stop) timing events on the coder stream (BatchCodec(time_coder=True))
        self._job = None                      # (_Job, _Worker) until somebody has started on the step's results

    def coder_ms(self):
        """Milliseconds the coder launches of this batch took on their stream (from the moment the symbols were ready to the
        publication of the results), other work sharing the GPU. Only with `BatchCodec(time_coder=True)`, after `result()`."""
        if self._coder_span is None:
            raise RuntimeError('the codec was not built with time_coder=True')
        return self._coder_span[0].elapsed_time(self._coder_span[1])

    def result(self):
        """Blocks until the batch is through; dict of numpy arrays, one entry per image:
        'nb_bits' int64 (lossless code incl. the exception map's entropy cost), 'coder_bits' int64, 'exception_bits' int64,
        'sse' int64 (sum of squared uint8 differences), 'nb_deads' int64.
        When the result worker has not started on this step yet (one step at a time: the caller is here a few microseconds after
        `submit`), the calling thread waits for the device and forms the results itself (`_Worker.process`)."""
        pending = self._job
        if pending is not None:
            self._job = None
            (job, worker) = pending
            if _RESULT_BY_CALLER and not self._done.is_set() and job.claim():
                worker.process(job, by_caller=True)
        self._done.wait()
        if self._error is not None:
            raise self._error
        return self._values


class StepTimeout(RuntimeError):
    """The device did not report a submitted step within EAE_WORKER_SEQUENCE_TIMEOUT_SECONDS: the codec is unusable from here on."""


class _Job(object):
    """What the results of one submitted step are made from; whoever claims it first -- the result worker taking it off its queue, or
    the caller's `Ticket.result()` -- waits for the device and forms the results."""
    __slots__ = ('fields', '_claimed')

    def __init__(self, *fields):
        self.fields = fields
        self._claimed = threading.Lock()

    def claim(self):
        return self._claimed.acquire(False)


_THREAD = threading.local()


class _Worker(threading.Thread):
    def __init__(self, map_size, nb_maps, host_probabilities, idx_map_exception, host_threads):
        super(_Worker, self).__init__(daemon=True)
        self.map_size = map_size
        self.nb_maps = nb_maps
        self.host_probabilities = host_probabilities
        self.idx_map_exception = idx_map_exception
        self.host_threads = host_threads
        self.jobs = queue.Queue()

    @staticmethod
    def _wait(event):
        if _POLL_SECONDS > 0.:
            while not event.query():
                time.sleep(_POLL_SECONDS)
        else:
            event.synchronize()

    _typical_wait = 0.      # seconds this thread lately had to wait per job (a running mean): `_wait_sequence` sleeps through most of it
    _alone = 0              # consecutive jobs that found nothing queued behind them (one step at a time)
    failed = None           # StepTimeout: the device stopped reporting; the codec refuses further work (BatchCodec.submit)
    _sleep = staticmethod(time.sleep)          # (a test drives `_wait_sequence` with a clock of its own: tests/test_host_logic.py)
    _now = staticmethod(time.monotonic)

    _caller_waits = ()      # the last waits of `_wait_sequence_by_caller` (seconds)

    def _wait_sequence_by_caller(self, words, expected, early=None):
        """`_wait_sequence` for the thread that called `Ticket.result()` on a step nobody had started on: that thread is blocked
        anyway, so nothing is handed over and what is left is to notice the device's last write as soon as it lands. ONE sleep up to
        shortly before the moment the step is expected to be through -- the shortest of the last three such waits, less a third of the
        spin budget: a lower bound as long as the regime lasts, and a change of regime (another shape of use, steps queued behind each
        other) shows as waits that differ, which switches the sleep off --, then the step counter is read in a loop for at most
        EAE_RESULT_SPIN_SECONDS of CPU, then polls like the worker's.
        early: called once the LAST counter (the synthesis side, which reports long before the coder's chains end when the step is
        small) has been reached, if that happens inside the sleep: the part of the results that does not come from the coder is
        formed while the coder still runs (`process`: 0.06 of the 0.09 ms between the device's last write and `result()` returning)."""
        started = self._now()
        waits = self._caller_waits
        steady = _LONG_SLEEP and len(waits) == 3 and max(waits) < 1.25*min(waits)
        wake = started + (min(waits) - _RESULT_SPIN_SECONDS/3.) if steady else started
        deadline = None
        if early is not None and len(expected) > 1:
            last = len(expected) - 1
            while wake - self._now() > 2.*_SEQUENCE_POLL_SECONDS:
                if ((int(words[last]) - expected[last]) & 0xFFFFFFFF) < 0x80000000:
                    early()
                    early = None
                    break
                self._sleep(min(2.*_SEQUENCE_POLL_SECONDS, wake - self._now()))
        if wake - self._now() > _SEQUENCE_POLL_SECONDS:
            self._sleep(wake - self._now())
        spin_until = self._now() + _RESULT_SPIN_SECONDS
        for (index, value) in enumerate(expected):
            while ((int(words[index]) - value) & 0xFFFFFFFF) >= 0x80000000:      # words[index] < value, wrap-around safe
                now = self._now()
                if now < spin_until:
                    continue
                self._sleep(0.5*_SEQUENCE_POLL_SECONDS)
                if deadline is None:
                    deadline = now + _SEQUENCE_TIMEOUT_SECONDS
                elif now > deadline:
                    raise StepTimeout('the device has not reported step {0} of this slot after {1:.0f} s (step counter at {2}): the slot\'s '
                                      'buffers may still be written to, so this codec takes no further batches -- close it'.format(
                                          value, _SEQUENCE_TIMEOUT_SECONDS, int(words[index])))
        self._caller_waits = (waits + (min(self._now() - started, 0.05),))[-3:]

    def _wait_sequence(self, words, expected):
        """Until the step counters the device leaves in pinned memory (`words`: numpy int32 view) have reached `expected`. Every
        sleep is a system call and a wake-up of this thread (a fifth of its time per step at twelve polls a step): the first sleep
        is three quarters of what the wait has lately been, the polls come after it."""
        deadline = None
        started = self._now()
        # (only in a steady regime -- more jobs queued behind this one, or one job at a time for a while: the last batches of a pipelined
        # run have the GPU to themselves and come back sooner than the mean says, and sleeping through that cost 20-step blocks 3 %:
        # 2,990 against 3,080 Mpx/s)
        queued = self.jobs.qsize()
        self._alone = self._alone + 1 if queued == 0 else 0
        steady = _LONG_SLEEP and (queued >= 2 or self._alone >= 4) and self._typical_wait > 4.*_SEQUENCE_POLL_SECONDS
        first = 0.75*self._typical_wait if steady else 0.
        poll = 0.5*_SEQUENCE_POLL_SECONDS if self._alone >= 4 else _SEQUENCE_POLL_SECONDS      # one step at a time: the caller is waiting for this
        for (index, value) in enumerate(expected):
            while ((int(words[index]) - value) & 0xFFFFFFFF) >= 0x80000000:      # words[index] < value, wrap-around safe
                self._sleep(max(first, poll))
                first = 0.
                if deadline is None:
                    deadline = self._now() + _SEQUENCE_TIMEOUT_SECONDS
                elif self._now() > deadline:
                    raise StepTimeout('the device has not reported step {0} of this slot after {1:.0f} s (step counter at {2}): the slot\'s '
                                      'buffers may still be written to, so this codec takes no further batches -- close it'.format(
                                          value, _SEQUENCE_TIMEOUT_SECONDS, int(words[index])))
        waited = min(self._now() - started, 0.05)
        if queued >= 2 or self._alone >= 3:      # (a regime change shows in the mean after a few jobs: `first` is a lower bound by then)
            self._typical_wait = waited if self._alone == 3 else self._typical_wait + 0.25*(waited - self._typical_wait)
        elif self._typical_wait == 0.:
            self._typical_wait = waited

    def run(self):
        _short_sleeps_for_this_thread()
        while True:
            job = self.jobs.get()
            if job is None:
                return
            if job.claim():          # (else the caller's `Ticket.result()` got there first and does it itself)
                self.process(job)

    def process(self, job, by_caller=False):
        """Waits until the device is through with the step, then forms the ticket's results from the slot's pinned blocks (or the
        exception `result()` raises), frees the slot. On the worker thread, or on the caller's (`Ticket.result()`)."""
        (ticket, events, views, symbols_host, slot_free, recount, fetch, sequence) = job.fields[:8]
        # the synthesis side's publication carried the analysis side's blocks too (exception-map histograms, dead-map flags, range
        # check): small steps, whose caller waits for each result -- `BatchCodec._early_publish`
        early_published = len(job.fields) > 8 and bool(job.fields[8])
        try:
            if by_caller and not getattr(_THREAD, 'short_sleeps', False):
                _short_sleeps_for_this_thread()
                _THREAD.short_sleeps = True
            arrays = [v if isinstance(v, numpy.ndarray) else v.numpy() for v in views]
            (results, hist, overflow, flags, checks, sse) = arrays
            early = {}

            def early_part():
                """Everything that does not come from the coder: in pinned memory once the synthesis side has reported, i.e. while the
                coder's chains still run when the step is small. Nothing is raised from here (the slot is still in use): an
                exception is kept and raised behind the waits, after the ones the old order put in front of it."""
                # the last word of the squared-error block: tiles that a cut conv launch of this batch left unfinished
                # (device.conv_workspace_collect); nothing of this batch can be trusted then
                early['unfinished'] = int(sse[-1])
                early['sse'] = sse[:-1].astype(numpy.int64).copy()
                early['out_of_range'] = int(checks[0]) != 0
                early['nb_deads'] = (flags == 0).sum(axis=1).astype(numpy.int64)
                early['exception_bits'] = numpy.zeros(ticket.nb_images, dtype=numpy.int64)
                try:
                    if hist.size and early['unfinished'] == 0 and not early['out_of_range']:
                        rows = hist
                        if int(overflow.sum()) != 0:
                            # a symbol beyond +-hist_radius: the reference's histogram runs from the smallest to the largest symbol
                            # whatever they are (lossless/compression.py:68-75, tools.py:376-388), so count again over all of int16
                            rows = recount()
                        early['exception_bits'] = lossless_compression.exception_maps_nb_bits(rows.astype(numpy.int64), self.map_size)
                except Exception as exc:
                    early['error'] = exc

            if sequence is not None:
                if by_caller:
                    self._wait_sequence_by_caller(*sequence, early=early_part if early_published else None)
                else:
                    self._wait_sequence(*sequence)
            for event in events:
                self._wait(event)
            if fetch is not None:
                # The copy back is issued HERE, behind events that have completed: on this runtime an asynchronous copy
                # whose stream still waits for an event holds the calling thread until it can start (the launch thread would
                # submit the next step only after this one is decoded: 4.5 instead of 3.2 ms per step).
                (reconstruction, pinned, stream) = fetch
                with torch.cuda.device(reconstruction.device), torch.cuda.stream(stream):
                    pinned.copy_(reconstruction, non_blocking=True)
                    copied = torch.cuda.Event()
                    copied.record()
                self._wait(copied)
                ticket.reconstruction_host = pinned.numpy()
            if not early:
                early_part()
            if early['unfinished'] != 0:
                raise dev.SplitHandOffTimeout('{} tiles of a cut conv launch were not handed over: the results of this batch '
                                              'are invalid (the workspace has been reset; later batches are unaffected)'.format(early['unfinished']))
            if symbols_host is not None:
                # encode + decode + compare per map on the host cores, like compress_lossless + the caller's assert
                (_, nb_bits) = lossless_compression.code_planar_symbols(symbols_host.numpy(), self.host_probabilities,
                                                                       self.idx_map_exception, nb_threads=self.host_threads,
                                                                       roundtrip=True, verify_only=True)
                results = numpy.zeros_like(results)
                results[0] = nb_bits.reshape(-1)
            if results[2].any():
                bad = int(numpy.flatnonzero(results[2])[0])
                if int(results[2, bad]) == 6:
                    raise AssertionError('\nArrays are not equal\nThe lossless compression has altered the centered quantized data.')
                from .kodak.lossless import interface_cython
                interface_cython.raise_for_status(int(results[2, bad]), int(results[3, bad]))
            if early['out_of_range']:
                raise AssertionError('The rounded array elements cannot be represented as 16-bit signed integers.')
            if 'error' in early:
                raise early['error']
            n = ticket.nb_images
            coder_bits = (results[0].astype(numpy.int64) + results[1].astype(numpy.int64)).reshape(n, self.nb_maps).sum(axis=1)
            exception_bits = early['exception_bits']
            ticket._values = {'nb_bits': coder_bits + exception_bits, 'coder_bits': coder_bits,
                              'exception_bits': exception_bits, 'sse': early['sse'], 'nb_deads': early['nb_deads']}
        except StepTimeout as exc:    # nothing says the device is through with this slot: no later submit may reuse it
            self.failed = exc
            ticket._error = exc
        except Exception as exc:      # surfaced by Ticket.result()
            ticket._error = exc
        finally:
            slot_free.set()
            ticket._done.set()


def default_nb_in_flight(h_in, w_in):
    """Batches of coder work to keep in flight when the caller does not say. A feature map is ONE serial chain (its symbols x
    about 0.2-0.4 us each to encode, the same again to decode, stretched next to the transforms), whatever the batch; the
    transforms of a batch take a time that grows with the batch instead. Measured on MI355X in the product mode
    (profiles/r03_depth_sweep.txt, profiles/r03_i_bench.json): 24 Kodak-sized images per step (maps of 1,536 symbols): three
    to five in flight are equal within 1 % at 0.2 bpp, five is the best at 1.4 bpp and 16 % ahead of three at 3.2 bpp; 64
    images of 256x256 (maps of 256 symbols: short chains, short steps) lose 10 % with five against three; a 2048x2048 image
    has maps of 16,384 symbols and needs eight. Round 5, on the shorter chains of that round's cores (profiles/r05_coder_waves_per_block.log):
    six against five for Kodak-sized maps is equal within noise up to 2 bpp and 2 % ahead at 3.2 bpp (seven no better), so: six."""
    map_size = (h_in//csts.STRIDE_PROD)*(w_in//csts.STRIDE_PROD)
    return int(min(8, 3 + map_size//512))


def default_coder_chunks(n_maps):
    """Launches the coder's serial chains are cut into (`BatchCodec(coder_chunks=None)`): ONE, i.e. off. For one or two images a
    step is as long as the coder's chains laid end to end (binarise, encoder core, emit, decoder core, debinarise: 0.65 of the 1.2 ms
    one Kodak image takes), and cut into chunks on three streams the decoder trails the encoder -- but on this runtime every
    cross-stream hop costs 60-100 us (launched directly or replayed as a hipGraph), more than a chunk saves at the headline's entropy:
    one image's round trip 0.65 ms whole, 0.84 / 0.95 / 1.24 ms in 2 / 3 / 4 chunks; at 2 bpp 2.46 ms whole, 2.29 / 2.18 in 4 / 8
    chunks (profiles/r05_trailing_alone.log). So the chunked form stays an option (`coder_chunks`, EAE_CODER_CHUNKS) for long chains."""
    forced = os.environ.get('EAE_CODER_CHUNKS')
    if forced:
        return max(1, min(16, int(forced)))
    return 1


PRODUCT_TRANSFORM_STREAMS = 3      # 2 / 3 / 4 / 5 streams: 3,020 / 3,075 / 3,060 / 3,030 Mpx/s for 24 Kodak images per step, 2,550 / 2,770 / 2,740
#                                    for 64 images of 256x256 (profiles/r03_transform_streams2.txt)


def product_mode(h_in, w_in):
    """The keyword arguments of the mode a deployment (and `bench.py`'s headline) runs `BatchCodec` in: consecutive batches go
    round three private transform streams, the step is replayed as three hipGraphs, the number of coder batches in flight follows
    the map size. `BatchCodec(..., **codec.product_mode(h, w))`; the constructor's own defaults are the conservative ones (every
    launch on the caller's stream, kernel by kernel: what the per-launch hooks and the roofline leg need)."""
    return {'nb_in_flight': default_nb_in_flight(h_in, w_in), 'nb_transform_streams': PRODUCT_TRANSFORM_STREAMS, 'use_graphs': True}


_BUDGET_WARNED = [False]


def stream_budget(nb_transform_streams, nb_in_flight, hw_queues=None, copies=0):
    """Caps a codec's busy streams to the hardware queues the process has: -> (nb_transform_streams, nb_in_flight, message or None).

    The HIP runtime gives a process GPU_MAX_HW_QUEUES hardware queues (4 unless the variable says otherwise; the package sets 16 at
    import when it still can: `__init__._configure_hw_queues`); streams beyond that share queues, busy ones with busy ones, which was
    measured to cost the product mode up to 30 %. One queue is left to the caller's own stream, `copies` to the feed / fetch streams.
    Coder streams (batches in flight) go first, down to two; then the transform streams, down to one; then the coder streams again."""
    if hw_queues is None:
        from autoencoder_based_image_compression_amd import HW_QUEUES
        hw_queues = HW_QUEUES[0]
    (nt, nf) = (max(1, int(nb_transform_streams)), max(1, int(nb_in_flight)))
    budget = max(2, int(hw_queues) - 1 - int(copies))
    (nt0, nf0) = (nt, nf)
    while nt + nf > budget and nf > 2:
        nf -= 1
    while nt + nf > budget and nt > 1:
        nt -= 1
    while nt + nf > budget and nf > 1:
        nf -= 1
    message = None
    if (nt, nf) != (nt0, nf0):
        message = ('BatchCodec: {0} transform + {1} coder streams asked for, but this process has {2} hardware queues (GPU_MAX_HW_QUEUES; '
                   'streams beyond them share queues and serialise): running {3} + {4}. Set GPU_MAX_HW_QUEUES=16 in the environment before '
                   'the first GPU call, or import this package before anything initialises the GPU.'.format(nt0, nf0, hw_queues, nt, nf))
    return (nt, nf, message)


class BatchCodec(object):
    """Encode -> quantise -> entropy-code (with round trip) -> decode -> squared error for batches of a fixed shape."""

    def __init__(self, variables, are_bin_widths_learned, bin_widths_test, map_mean, binary_probabilities, idx_map_exception,
                 batch_size, h_in, w_in, device='cuda', nb_in_flight=None, keep_reconstruction=False, launch_hook=None,
                 coder='device', host_coder_threads=0, hist_radius=2047, nb_transform_streams=1, use_graphs=False,
                 time_coder=False, fuse_latent=False, fetch_reconstruction=False, coder_chunks=None, one_stream_steps=False):
        """coder: 'device' (the coder kernels on side streams), 'host' (ONE device -> host copy of the symbols per batch, then
        the host C-ABI coder `eae_coder_compress_maps` on `host_coder_threads` threads: the shape BASELINE.json sketches) or
        'none' (transforms only; the bit counts come back as zeros).
        nb_in_flight: batches whose coder work may be pending at once, each on its own side stream (None:
        `default_nb_in_flight(h_in, w_in)`).
        nb_transform_streams: 1 = the transforms run on the caller's current stream; more = consecutive batches alternate
        between that many private streams (worth it only for small batches, whose kernels leave most of the GPU idle).
        one_stream_steps: a step's coder runs on the step's transform stream, behind its synthesis transform, instead of beside it on a
        coder stream: ONE graph launch per step, no event between streams. For one or two images per step, where consecutive steps on
        `nb_transform_streams` streams are what fills the GPU and a hop between streams costs 60-100 us on this runtime (one Kodak
        image per step on 14 streams: 0.295 -> 0.252 ms per image, the launching thread 0.15 -> 0.06 ms; a single step takes the
        coder's 0.2 ms longer).
        use_graphs: capture the launches of one step into three hipGraphs per slot on first use (analysis side, coder,
        synthesis side) and replay them afterwards: three host launches per step instead of about twenty. For small batches, where the launch thread is the
        bottleneck (one Kodak image per step); `launch_hook` is not called for replayed steps. Not for coder='host'.
        fuse_latent: run the latent stage (gdn_3, quantiser, inverse_gdn_4) as the epilogue of the conv_3 launch instead of as
        its own kernel (device.conv5x5s2_latent; same bits). One launch fewer, but at Kodak batch sizes conv_3 has one tile per
        SIMD and nothing to hide that epilogue behind: 3.41 against 3.44 ms per 24 images, with the conv_3 launch at 0.36 ms
        instead of 0.27 + 0.12. Pays for batches that give conv_3 several tiles per SIMD.
        fetch_reconstruction: the uint8 reconstructions are copied to pinned host memory by the result worker (on a stream of its
        own, once the batch is decoded) and `Ticket.reconstruction_host` holds them after `result()`: the fetch of the reference's
        `decode_mini_batches` (eae/batching.py:49-53). The feed is `submit()` with a pinned HOST tensor.
        coder_chunks: 2..16 cuts the coder's serial chains into that many launches each, so that the emit pass and the decoder run
        while the encoder core is still at work (`device.coder_roundtrip_trailing`; same results). None: `default_coder_chunks`
        (off: on this runtime the hops between the three streams cost more than the overlap saves, except for long chains).
        hist_radius: the exception map's entropy is formed from an exact histogram of its symbols over [-hist_radius,
        hist_radius]; when a symbol falls outside, the result worker counts that batch's exception maps again over the whole
        int16 range (like the image-by-image functions of `kodak/`; the reference's histogram has no bound,
        lossless/compression.py:68-75)."""
        if coder not in ('device', 'host', 'none'):
            raise ValueError('`coder` is neither "device" nor "host" nor "none".')
        if use_graphs and coder == 'host':
            raise ValueError('`use_graphs` needs the coder on the device (or none).')
        if h_in % csts.STRIDE_PROD != 0 or w_in % csts.STRIDE_PROD != 0:
            raise ValueError('The image size is not divisible by the product of the three strides.')
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        if self.device.index != torch.cuda.current_device():      # launches go to the current device's streams (device._stream)
            raise ValueError('`device` is {0} but the current device is cuda:{1}: build and use the codec under '
                             '`torch.cuda.device({0!r})`.'.format(self.device, torch.cuda.current_device()))
        self.learned = are_bin_widths_learned
        self.encoder = pipeline.DeviceEncoder(variables, are_bin_widths_learned, self.device)
        self.decoder = pipeline.DeviceDecoder(variables, are_bin_widths_learned, self.device)
        self.batch_size = batch_size
        (self.h_in, self.w_in) = (h_in, w_in)
        self.nb_maps = csts.NB_MAPS_3
        self.map_size = (h_in//csts.STRIDE_PROD)*(w_in//csts.STRIDE_PROD)
        self.idx_map_exception = idx_map_exception if 0 <= idx_map_exception < self.nb_maps else -1
        probabilities = numpy.ascontiguousarray(binary_probabilities, dtype=numpy.float64)
        if probabilities.ndim != 2 or probabilities.shape[0] != self.nb_maps:
            raise ValueError('`binary_probabilities` must have one row per map.')
        self.truncated_unary_length = probabilities.shape[1]
        if self.truncated_unary_length > 255:
            raise OverflowError('value too large to convert to numpy.uint8_t')   # interface_cython.pyx:49
        self.bin_widths = torch.from_numpy(numpy.ascontiguousarray(bin_widths_test, dtype=numpy.float32)).to(self.device)
        self.map_mean = torch.from_numpy(numpy.ascontiguousarray(map_mean, dtype=numpy.float32)).to(self.device)
        self.probabilities = torch.from_numpy(probabilities).to(self.device)
        prob_row = torch.arange(self.nb_maps, dtype=torch.int32).repeat(batch_size)
        if self.idx_map_exception >= 0:
            prob_row[self.idx_map_exception::self.nb_maps] = -1        # costed from its histogram (compression.py:68-75)
        self.prob_row = prob_row.to(self.device)
        self.fetch_reconstruction = bool(fetch_reconstruction)
        self.keep_reconstruction = bool(keep_reconstruction) or self.fetch_reconstruction
        self.hist_radius = int(hist_radius)
        self.coder = coder
        self.time_coder = bool(time_coder)      # Ticket.coder_ms(): the launch-by-launch path only
        self.fuse_latent = bool(fuse_latent)
        self.one_stream_steps = bool(one_stream_steps)
        # launch_hook(name, fn): called for every timed launch of the launch-by-launch path with name in ('conv1_gdn1',
        # 'conv2_gdn2', 'conv3', 'latent', 'tconv1_igdn5', 'tconv2_igdn6', 'tconv3', 'coder_encode', 'coder_decode') on the
        # stream the launch goes to (bench.py brackets them with HIP events); it must return fn()
        self.launch_hook = launch_hook if launch_hook is not None else (lambda name, fn: fn())
        n_maps = batch_size*self.nb_maps
        nb_hist = batch_size if self.idx_map_exception >= 0 else 0
        self._n_maps = n_maps
        # per-slot device block for the host: [coder results 4 x n_maps | exception histograms | overflow | flags | checks(4)]
        # followed by the squared errors (int64 per image, published on their own once the synthesis transform is through)
        self._layout = (4*n_maps, nb_hist*(2*self.hist_radius + 1), nb_hist, n_maps, 4)
        nb_words = sum(self._layout)
        assert nb_words % 2 == 0
        if nb_in_flight is None:
            nb_in_flight = default_nb_in_flight(h_in, w_in)
        if coder == 'device' and not one_stream_steps and os.environ.get('EAE_IGNORE_HW_QUEUES') != '1':
            # no more busy streams than the process has hardware queues (said once; EAE_IGNORE_HW_QUEUES=1: the experiments' switch)
            (nb_transform_streams, nb_in_flight, message) = stream_budget(nb_transform_streams, nb_in_flight,
                                                                          copies=1 if fetch_reconstruction else 0)
            if message is not None and not _BUDGET_WARNED[0]:
                _BUDGET_WARNED[0] = True
                import warnings
                warnings.warn(message, RuntimeWarning, stacklevel=2)
        self.nb_in_flight = nb_in_flight
        self.nb_transform_streams = nb_transform_streams
        self.nb_slots = nb_in_flight + 2
        nb_private = nb_transform_streams if (nb_transform_streams > 1 or use_graphs) else 0      # replays never go to the caller's stream
        if self.one_stream_steps:
            # no coder streams; the steps' streams come from both of the process's lists, so that a process that has run other codecs
            # makes as few new streams as it can (streams beyond GPU_MAX_HW_QUEUES share hardware queues, busy ones with busy ones)
            self._streams = []
            self._transform_streams = _step_streams(nb_private, self.device)
        else:
            self._streams = _side_streams(nb_in_flight, self.device)
            self._transform_streams = _side_streams(nb_private, self.device, kind='transform')
        # ... and behind the squared errors one more 64-bit word whose low half is the conv workspace's error word of the step
        self._slot_all = [torch.zeros(nb_words + 2*batch_size + 2, dtype=torch.int32, device=self.device) for _ in range(self.nb_slots)]
        self._slot_out = [t[:nb_words] for t in self._slot_all]
        self._pinned_out = [torch.zeros(nb_words, dtype=torch.int32).pin_memory() for _ in range(self.nb_slots)]
        self._slot_sse = [t[nb_words:].view(torch.int64) for t in self._slot_all]           # [batch_size] errors + [1] status
        self._slot_unfinished = [t[nb_words + 2*batch_size:nb_words + 2*batch_size + 1] for t in self._slot_all]
        self._pinned_sse = [torch.zeros(batch_size + 1, dtype=torch.int64).pin_memory() for _ in range(self.nb_slots)]
        self._symbols = [torch.empty((batch_size, self.nb_maps, self.map_size), dtype=torch.int16, device=self.device)
                         for _ in range(self.nb_slots)]
        self._coder_streams = [dev.CoderStreams(n_maps, self.map_size, self.truncated_unary_length, self.device,
                                                results=self._views(self._slot_out[i])[0]) for i in range(self.nb_slots)]
        self._coder_behind_tconv1 = (n_maps > 256) if _CODER_BEHIND_TCONV1 is None else _CODER_BEHIND_TCONV1 != '0'
        # one or two images per step: the caller usually waits for each result, and what it waits for last is the coder. The analysis
        # side's blocks (exception-map histograms, dead-map flags, range check) then travel with the synthesis side's publication
        # already -- one more copy launch on the transform stream, 0.4 ms before the coder ends -- and `Ticket.result()` turns them into
        # their share of the results while the coder still runs (`_Worker.process`).
        # (not with `one_stream_steps`: that mode is for many small steps in flight, where the launching thread's time per step is the rate)
        self._early_publish = n_maps <= 256 and _EARLY_PUBLISH and not one_stream_steps
        self.coder_chunks = int(default_coder_chunks(n_maps) if coder_chunks is None else coder_chunks)
        if self.coder_chunks > 1 and coder != 'device':
            self.coder_chunks = 1
        make_workspace = dev.coder_trailing_workspace if self.coder_chunks > 1 else dev.coder_workspace
        self._workspaces = [make_workspace(n_maps, self.map_size, self.truncated_unary_length, self.device) for _ in range(self.nb_slots)]
        # scratch that lets the conv GEMM launches cut their last tiles (device.conv_workspace): a slot's launches never overlap each other
        self._conv_ws = [dev.conv_workspace(self.device) for _ in range(self.nb_slots)]
        # step counters of every slot: [coder side, synthesis side] on the device (+ the two ticket words of device.publish_step), their
        # published values in pinned memory, and how many times the host has submitted each side (what the result worker waits for)
        self._seq_dev = [torch.zeros(4, dtype=torch.int32, device=self.device) for _ in range(self.nb_slots)]
        self._pinned_seq = [torch.zeros(2, dtype=torch.int32).pin_memory() for _ in range(self.nb_slots)]
        self._seq_host = [t.numpy() for t in self._pinned_seq]
        self._counts = [[0, 0] for _ in range(self.nb_slots)]
        self._slot_free = [threading.Event() for _ in range(self.nb_slots)]
        for event in self._slot_free:
            event.set()
        self._pinned_symbols = [torch.empty((batch_size, self.nb_maps, self.map_size), dtype=torch.int16).pin_memory()
                                if coder == 'host' else None for _ in range(self.nb_slots)]
        self._worker = _Worker(self.map_size, self.nb_maps, probabilities if coder == 'host' else None, self.idx_map_exception,
                               host_coder_threads)
        self._worker.start()
        self._index = 0
        self.use_graphs = bool(use_graphs)
        self._graphs = [None]*self.nb_slots          # per slot: (three graphs, static input, latents, reconstruction)
        self._warm = False
        self._recount_stream = None
        # host in / host out: a copy stream each way, a pinned buffer per slot for the reconstructions, device staging for the inputs
        self._feed_stream = None
        self._staging = [None]*self.nb_slots
        self._slot_views = [None]*self.nb_slots      # the views of a slot's pinned result block, made once (graph mode)
        self._fetch_stream = torch.cuda.Stream(device=self.device) if self.fetch_reconstruction else None
        self._pinned_rec = [torch.empty((batch_size, h_in, w_in), dtype=torch.uint8).pin_memory() if self.fetch_reconstruction else None
                            for _ in range(self.nb_slots)]
        assert not self.use_graphs or self._transform_streams      # replays never go to the caller's stream
        # last: a constructor that raised above leaves nothing half-built behind for another codec's `_capture_all` to drain
        with _LIVE_LOCK:
            _LIVE.setdefault(self.device.index, weakref.WeakSet()).add(self)

    def _views(self, t):
        out = []
        pos = 0
        for count in self._layout:
            out.append(t[pos:pos + count])
            pos += count
        (results, hist, overflow, flags, checks) = out
        nb_hist = overflow.numel()
        return (results.view(4, self._n_maps), hist.view(nb_hist, 2*self.hist_radius + 1) if nb_hist else hist, overflow,
                flags.view(self.batch_size, self.nb_maps), checks)

    def submit(self, luminances_uint8):
        """uint8 tensor (batch_size, h_in, w_in), on the device or in PINNED host memory -> Ticket. Everything is enqueued; nothing
        is waited for except a free slot (at most nb_in_flight + 2 batches are pending). A host batch is copied in on a copy
        stream of the codec's own (`Ticket.fed_event` is recorded behind the copy: leave the batch alone until then)."""
        if luminances_uint8.dtype != torch.uint8:
            raise TypeError('`luminances_uint8.dtype` is not equal to `torch.uint8`.')
        if tuple(luminances_uint8.shape) != (self.batch_size, self.h_in, self.w_in):
            raise ValueError('`luminances_uint8.shape` is not (batch_size, h_in, w_in).')
        if self._worker.failed is not None:
            raise StepTimeout('this codec stopped taking batches: {0}'.format(self._worker.failed))
        host = luminances_uint8.device.type == 'cpu'
        if host and not luminances_uint8.is_pinned():
            raise ValueError('a host batch must be in pinned memory (`torch.Tensor.pin_memory()`): its copy is asynchronous.')
        if self.use_graphs:
            return self._submit_graph(luminances_uint8)
        if not self._transform_streams:
            return self._submit(luminances_uint8)
        # small batches leave most of the GPU idle and a step is a chain of short dependent kernels: consecutive batches go
        # to different streams so that their chains overlap
        stream = self._transform_streams[self._index % len(self._transform_streams)]
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            ticket = self._submit(luminances_uint8)
        if not host:
            luminances_uint8.record_stream(stream)
        return ticket

    def _feed(self, host_batch, device_batch, ticket_holder, fresh=False):
        """Host -> device copy of a pinned batch on the codec's feed stream; the CURRENT stream waits for it. fresh: the
        destination has just come from the caching allocator, which may have handed out a block that kernels still queued on
        the current stream use (freed intermediates of the previous steps: safe to reuse in stream order only): the copy
        then goes behind everything the current stream holds."""
        if self._feed_stream is None:
            self._feed_stream = torch.cuda.Stream(device=self.device)
        if fresh:
            self._feed_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._feed_stream):
            device_batch.copy_(host_batch, non_blocking=True)
            fed = torch.cuda.Event()
            fed.record()
        torch.cuda.current_stream().wait_event(fed)
        ticket_holder.append(fed)

    def _submit_graph(self, luminances_uint8):
        """One step = three hipGraph launches: the analysis side (conv1 .. symbols) and the synthesis side on a transform
        stream, the coder on a coder stream between two events, exactly the stream structure of the launch-by-launch path.
        Slot s owns its buffers (symbols, streams, result blocks), so it owns its three graphs too: all captured at the first
        call (after one ordinary step that gets every lazy initialisation out of the way: `_capture_all`), replayed afterwards."""
        if not self._warm:
            self._submit(luminances_uint8).result()          # first launches: function attributes, lazy module loads
            try:
                self._capture_all(luminances_uint8)
            except BaseException:
                self._graphs = [None]*self.nb_slots          # a half-captured set is of no use: the next submit starts over
                raise
            self._warm = True
        slot = self._index % self.nb_slots
        stream = self._transform_streams[self._index % len(self._transform_streams)]
        coder_stream = self._streams[self._index % len(self._streams)] if self._streams else None
        self._index += 1
        self._slot_free[slot].wait()
        self._slot_free[slot].clear()
        sequence_mode = _WAIT_MODE == 'sequence'
        expected = (self._counts[slot][0] + 1, self._counts[slot][1] + 1)      # what the slot's step counters will show behind this step
        try:
            caller = torch.cuda.current_stream()
            if self._graphs[slot] is None:
                raise RuntimeError('slot {} has no captured graphs (a capture failed earlier)'.format(slot))
            (graphs, static_input, _, reconstruction) = self._graphs[slot]
            fed = []
            # the launch thread's time is what the pipelined small-batch rate is made of: the current stream is switched directly
            # (`with torch.cuda.stream(...)` costs three device-index resolutions per entry and exit) and restored in the `finally`
            try:
                torch.cuda.set_stream(stream)
                if luminances_uint8.device.type == 'cpu':
                    self._feed(luminances_uint8, static_input, fed)
                else:
                    stream.wait_stream(caller)
                    static_input.copy_(luminances_uint8, non_blocking=True)
                graphs[0].replay()
                (coded, decoded) = (None, None)
                if self.one_stream_steps:
                    if not sequence_mode:                     # the whole step is graphs[0]: both of the worker's events behind it
                        coded = torch.cuda.Event()
                        coded.record(stream)
                else:
                    quantized = torch.cuda.Event()
                    quantized.record(stream)
                    torch.cuda.set_stream(coder_stream)
                    coder_stream.wait_event(quantized)
                    graphs[1].replay()
                    if not sequence_mode:
                        coded = torch.cuda.Event()
                        coded.record(coder_stream)
                    torch.cuda.set_stream(stream)
                    graphs[2].replay()
                if not sequence_mode or self.keep_reconstruction:      # (a caller reading the reconstruction on another stream waits for it)
                    decoded = torch.cuda.Event()
                    decoded.record(stream)
            finally:
                torch.cuda.set_stream(caller)
            if not fed:
                luminances_uint8.record_stream(stream)
            ticket = Ticket(self.batch_size)
            ticket.decoded_event = decoded
            ticket.fed_event = fed[0] if fed else None
            if self.keep_reconstruction:
                ticket.reconstruction_uint8 = reconstruction       # valid until this slot is replayed again
            if self._slot_views[slot] is None:
                self._slot_views[slot] = self._views(self._pinned_out[slot]) + (self._pinned_sse[slot],)
            job = _Job(ticket, () if sequence_mode else (coded, decoded), self._slot_views[slot], None,
                       self._slot_free[slot], lambda: self._recount_exception_maps(slot), self._fetch_job(slot, reconstruction),
                       (self._seq_host[slot], expected) if sequence_mode else None, self._early_publish)
            ticket._job = (job, self._worker)
            self._worker.jobs.put(job)
            self._counts[slot] = list(expected)
            return ticket
        except BaseException:
            self._resync(slot)
            self._slot_free[slot].set()       # nobody will report on this slot: without this, drain() / close() wait for ever
            raise

    def _capture_all(self, like):
        """Captures the three graphs of EVERY slot now, while nothing of THIS codec is in flight and its result worker is idle
        (other codecs of the process share the side streams: capture before they are busy, or give them other streams): a capture that
        starts later shares its stream with replays whose events the worker is polling, and on this runtime a query of an
        event recorded on a capturing stream invalidates the capture (seen with one transform stream and 24 Kodak images:
        hipErrorStreamCaptureInvalidated at the first launch of the second slot's capture)."""
        # Codecs of this process share the device's side streams (and their slices move with `nb_in_flight`): one that is busy would
        # have its worker polling events on a stream this capture uses. Wait until every other live codec of the device is idle;
        # a caller that keeps submitting to another codec from another thread during a first submit here is outside the contract
        # (bring a device's codecs up one after the other).
        with _LIVE_LOCK:
            others = [c for c in _LIVE.get(self.device.index, ()) if c is not self and getattr(c, '_worker', None) is not None]
        for other in others:
            other.drain()
        torch.cuda.synchronize(self.device)
        for slot in range(self.nb_slots):
            # any of the codec's streams will do for the capture: a replay runs on the stream it is launched into, which
            # `_submit_graph` picks from the submission index (slots and streams go round at different periods)
            stream = self._transform_streams[slot % len(self._transform_streams)]
            coder_stream = self._streams[slot % len(self._streams)] if self._streams else None
            static_input = torch.empty(tuple(like.shape), dtype=torch.uint8, device=self.device)      # `like` may be a host batch
            if self.one_stream_steps:
                graphs = [torch.cuda.CUDAGraph()]
                with torch.cuda.graph(graphs[0], stream=stream, capture_error_mode='thread_local'):
                    latents = self._launch_analysis(static_input, slot, None)
                    reconstruction = self._launch_synthesis(latents, static_input, slot, None)
                    self._launch_coder(slot)
                self._graphs[slot] = (graphs, static_input, latents, reconstruction)
                continue
            graphs = [torch.cuda.CUDAGraph() for _ in range(3)]
            with torch.cuda.graph(graphs[0], stream=stream, capture_error_mode='thread_local'):
                latents = self._launch_analysis(static_input, slot, None)
                if self._coder_behind_tconv1:
                    latents = self._launch_synthesis_head(latents, slot, None)
            with torch.cuda.graph(graphs[1], stream=coder_stream, capture_error_mode='thread_local'):
                self._launch_coder(slot)
            with torch.cuda.graph(graphs[2], stream=stream, capture_error_mode='thread_local'):
                reconstruction = self._launch_synthesis(latents, static_input, slot, None, head_done=self._coder_behind_tconv1)
            self._graphs[slot] = (graphs, static_input, latents, reconstruction)

    def _submit(self, luminances_uint8):
        slot = self._index % self.nb_slots
        stream = self._streams[self._index % len(self._streams)] if self._streams else None
        self._index += 1
        self._slot_free[slot].wait()
        self._slot_free[slot].clear()
        sequence_mode = _WAIT_MODE == 'sequence'
        expected = (self._counts[slot][0] + 1, self._counts[slot][1] + 1)
        try:
            hook = self.launch_hook
            fed = []
            if luminances_uint8.device.type == 'cpu':
                fresh = self._staging[slot] is None
                if fresh:
                    self._staging[slot] = torch.empty((self.batch_size, self.h_in, self.w_in), dtype=torch.uint8, device=self.device)
                self._feed(luminances_uint8, self._staging[slot], fed, fresh)
                luminances_uint8 = self._staging[slot]
            latents = self._launch_analysis(luminances_uint8, slot, hook)
            head_done = self._coder_behind_tconv1 and not self.one_stream_steps
            if head_done:
                latents = self._launch_synthesis_head(latents, slot, hook)
            ticket = Ticket(self.batch_size)
            if self.one_stream_steps:                          # the coder behind the synthesis transform, on this stream
                reconstruction = self._launch_synthesis(latents, luminances_uint8, slot, hook)
            else:
                quantized = torch.cuda.Event()
                quantized.record()
            with (contextlib.nullcontext() if self.one_stream_steps else torch.cuda.stream(stream)):
                if not self.one_stream_steps:
                    stream.wait_event(quantized)
                if self.time_coder:
                    started = torch.cuda.Event(enable_timing=True)
                    started.record()
                self._launch_coder(slot, hook)
                (coded, decoded) = (None, None)
                if not sequence_mode or self.time_coder:
                    coded = torch.cuda.Event(enable_timing=self.time_coder)
                    coded.record()
                if self.time_coder:
                    ticket._coder_span = (started, coded)
            if not self.one_stream_steps:
                reconstruction = self._launch_synthesis(latents, luminances_uint8, slot, hook, head_done=head_done)
            if not sequence_mode or self.keep_reconstruction:
                decoded = torch.cuda.Event()
                decoded.record()
            ticket.decoded_event = decoded
            ticket.fed_event = fed[0] if fed else None
            if self.keep_reconstruction:
                ticket.reconstruction_uint8 = reconstruction
            job = _Job(ticket, () if sequence_mode else (coded, decoded), self._views(self._pinned_out[slot]) + (self._pinned_sse[slot],),
                       self._pinned_symbols[slot], self._slot_free[slot], lambda: self._recount_exception_maps(slot),
                       self._fetch_job(slot, reconstruction), (self._seq_host[slot], expected) if sequence_mode else None,
                       self._early_publish)
            ticket._job = (job, self._worker)
            self._worker.jobs.put(job)
            self._counts[slot] = list(expected)
            return ticket
        except BaseException:
            self._resync(slot)
            self._slot_free[slot].set()       # as in _submit_graph
            raise

    def _resync(self, slot):
        """A submit that raised may have launched some of its step: wait for whatever it did launch and take the slot's step
        counters from the device, so that the next job on this slot waits for the right values."""
        try:
            torch.cuda.synchronize(self.device)
            self._counts[slot] = [int(v) for v in self._seq_dev[slot][:2].cpu().tolist()]
            self._seq_dev[slot][2:].zero_()                    # ticket words of a publish that never ran to its end
            self._slot_all[slot][4*self._n_maps:].zero_()      # accumulators a publish would have cleared
        except Exception:      # the device itself is in trouble: the next submit will say so
            pass

    def _fetch_job(self, slot, reconstruction):
        """What the result worker needs to copy the slot's reconstruction to the host (None: not asked for)."""
        if not self.fetch_reconstruction:
            return None
        reconstruction.record_stream(self._fetch_stream)
        return (reconstruction, self._pinned_rec[slot], self._fetch_stream)

    def _recount_exception_maps(self, slot):
        """Histograms of the slot's exception maps over all of int16, as a host array [batch, 65535] (called by the result
        worker while it still owns the slot, on its own stream, when a symbol fell outside `hist_radius`)."""
        with torch.cuda.device(self.device):
            if self._recount_stream is None:
                self._recount_stream = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._recount_stream):
                (hist, overflow) = dev.symbol_histograms(self._symbols[slot].view(self._n_maps, self.map_size), 32767,
                                                         first_map=self.idx_map_exception, map_step=self.nb_maps)
                host = hist.cpu().numpy()
                if int(overflow.sum().item()) != 0:          # -32768: cast_float_to_int16 never produces it (tools.py:95-133)
                    raise RuntimeError('exception-map symbol outside [-32767, 32767]')
        return host

    @staticmethod
    def _no_hook(name, fn):
        return fn()

    def _launch_analysis(self, luminances_uint8, slot, hook):
        """conv1+GDN1 -> conv2+GDN2 -> conv3 -> latent stage (symbols, dead-map flags, decoder input) -> exception-map
        histograms, on the current stream. Returns the synthesis transform's input."""
        hook = hook or self._no_hook
        enc = self.encoder
        v = enc.v
        d = self.decoder.v
        gdn_1 = hook('conv1_gdn1', lambda: dev.conv9x9s4_u8(luminances_uint8, enc.w1, v['encoder/biases_1'], enc.g[1], v['encoder/beta_1']))
        ws = self._conv_ws[slot]
        gdn_2 = hook('conv2_gdn2', lambda: dev.conv5x5s2(gdn_1, enc.w2, v['encoder/biases_2'], dev.NORM_GDN, enc.g[2], v['encoder/beta_2'], workspace=ws))
        # (histograms, overflow, flags, checks, squared errors are accumulated into: zero when the slot is made, and zeroed again by the
        # launches that publish them, `_launch_coder` / `_launch_synthesis`)
        (_, hist, overflow, flags, checks) = self._views(self._slot_out[slot])
        gdn_in = None if self.learned else (enc.g[3], v['encoder/beta_3'])
        igdn_out = None if self.learned else (self.decoder.g[4], d['decoder/beta_4'])
        if self.fuse_latent:
            # conv_3 with the latent stage as its epilogue: one launch, the latents never go through HBM in between
            q = hook('conv3', lambda: dev.conv5x5s2_latent(gdn_2, enc.w3, v['encoder/biases_3'], self.bin_widths, self.map_mean, gdn_in=gdn_in,
                                                           igdn_out=igdn_out, want_flags=True, out_symbols=self._symbols[slot],
                                                           out_flags=flags, out_checks=checks[:3], workspace=ws))
        else:
            y_raw = hook('conv3', lambda: dev.conv5x5s2(gdn_2, enc.w3, v['encoder/biases_3'], dev.NORM_NONE, workspace=ws))
            # gdn_3 -> centre / quantise / symbols / dead-map flags -> de-centre -> inverse_gdn_4: one pass over the latents
            q = hook('latent', lambda: dev.latent_stage(y_raw, self.bin_widths, self.map_mean, gdn_in=gdn_in, igdn_out=igdn_out,
                                                        want_shifted=self.learned, want_symbols=True, want_flags=True,
                                                        out_symbols=self._symbols[slot], out_flags=flags, out_checks=checks[:3]))
        if self.idx_map_exception >= 0:
            dev.symbol_histograms(self._symbols[slot].view(self._n_maps, self.map_size), self.hist_radius, out=(hist, overflow),
                                  first_map=self.idx_map_exception, map_step=self.nb_maps, zero=False)
        if self._early_publish:
            # The analysis side's blocks are final here and reach pinned memory at once (the coder's publication, which also zeroes them
            # for the slot's next step, copies them again): in FRONT of the event the coder's stream waits for, so the zeroing cannot
            # overtake this copy. The host looks at them when the synthesis side has reported (`_Worker.process`).
            first = 4*self._n_maps
            dev.publish_to_host(self._slot_out[slot][first:], self._pinned_out[slot][first:])
        return q['shifted'] if self.learned else q['t']

    def _launch_coder(self, slot, hook=None):
        """The lossless coder over the slot's symbols (encode every map, decode it back, compare) and the publication of the
        slot's result block, on the current stream."""
        hook = hook or self._no_hook
        symbols = self._symbols[slot].view(self._n_maps, self.map_size)
        if self.coder == 'device' and self.coder_chunks > 1:
            hook('coder_roundtrip', lambda: dev.coder_roundtrip_trailing(symbols, self.probabilities, self.prob_row, self.truncated_unary_length,
                                                                         chunks=self.coder_chunks, out=self._coder_streams[slot],
                                                                         workspace=self._workspaces[slot]))
        elif self.coder == 'device':
            hook('coder_encode', lambda: dev.coder_encode_batch(symbols, self.probabilities, self.prob_row, self.truncated_unary_length,
                                                                out=self._coder_streams[slot], workspace=self._workspaces[slot]))
            hook('coder_decode', lambda: dev.coder_decode_batch(self._coder_streams[slot], self.probabilities, self.prob_row,
                                                                expected=symbols, workspace=self._workspaces[slot]))
        elif self.coder == 'host':
            self._pinned_symbols[slot].copy_(self._symbols[slot], non_blocking=True)
        else:
            self._views(self._slot_out[slot])[0].zero_()
        seq = self._seq_dev[slot]
        dev.publish_step(self._slot_out[slot], self._pinned_out[slot], 4*self._n_maps, seq[2:3], seq[0:1], self._pinned_seq[slot][0:1])

    def _launch_synthesis_head(self, latents, slot, hook):
        """tconv1+IGDN5 on the current stream (the first launch of the synthesis side, apart: `_coder_behind_tconv1`)."""
        hook = hook or self._no_hook
        dec = self.decoder
        d = dec.v
        return hook('tconv1_igdn5', lambda: dev.tconv5x5s2(latents, dec.w4, d['decoder/biases_4'], dev.NORM_IGDN, dec.g[5], d['decoder/beta_5'],
                                                           workspace=self._conv_ws[slot]))

    def _launch_synthesis(self, latents, luminances_uint8, slot, hook, head_done=False):
        """tconv1+IGDN5 -> tconv2+IGDN6 -> tconv3 + BT.601 cast + squared error against the input, and the publication of the
        squared errors, on the current stream. Returns the uint8 reconstruction. head_done: `latents` is already tconv1's output."""
        hook = hook or self._no_hook
        dec = self.decoder
        d = dec.v
        ws = self._conv_ws[slot]
        t = latents if head_done else self._launch_synthesis_head(latents, slot, hook)
        t = hook('tconv2_igdn6', lambda: dev.tconv5x5s2(t, dec.w5, d['decoder/biases_5'], dev.NORM_IGDN, dec.g[6], d['decoder/beta_6'], workspace=ws))
        (_, reconstruction, _) = hook('tconv3', lambda: dev.tconv9x9s4_luma(t, dec.w6, want_f32=False, want_u8=True, ref_u8=luminances_uint8,
                                                                            sse=self._slot_sse[slot][:self.batch_size]))
        # every conv launch of this step (analysis side too: same stream, same workspace) is behind us: its error word, and a
        # clean workspace for the slot's next step
        seq = self._seq_dev[slot]
        dev.publish_step(self._slot_sse[slot], self._pinned_sse[slot], 0, seq[3:4], seq[1:2], self._pinned_seq[slot][1:2],
                         conv_ws=ws, error_word=self._slot_unfinished[slot])
        return reconstruction

    def drain(self):
        """Waits until every submitted batch is through."""
        torch.cuda.synchronize(self.device)
        for event in self._slot_free:
            event.wait()

    def close(self):
        """Waits for the pending batches (whether or not their tickets failed) and joins the result worker: a worker still
        unwinding while the interpreter finalises aborts the process at exit. Idempotent."""
        if getattr(self, '_worker', None) is None:      # closed already, or a constructor that raised before the worker existed
            return
        try:
            self.drain()
        finally:
            self._worker.jobs.put(None)
            self._worker.join()
            self._worker = None
            with _LIVE_LOCK:
                _LIVE.get(self.device.index, weakref.WeakSet()).discard(self)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, traceback):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: nothing left to report to
            pass
