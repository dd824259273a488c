#!/usr/bin/env python
"""bench.py -- Mpixels/s of the compression inference path (encode -> quantise -> entropy-code -> decode) on MI355X.

One STEP = one pass of the whole hot path over one batch of Kodak-sized (512x768) synthetic luminance images that
are already resident in HBM:
    conv1+GDN1 -> conv2+GDN2 -> conv3 -> [GDN3 -> centre/quantise/int16 symbols (+dead-map flags) -> IGDN4] (one kernel)
    -> exception-map histogram
    -> lossless coder ON THE DEVICE (UEG0 + binary arithmetic coder, one map per lane: encode + decode + compare, on its
       own stream, concurrent with the synthesis transforms); streams stay in HBM, per-map bit counts go to the host
    -> tconv1+IGDN5 -> tconv2+IGDN6 -> tconv3 + BT.601 cast + squared error vs the input (PSNR).
Nothing is skipped or cached between steps. Multi-GPU: one process per GPU, each rank codes its own batch (weak
scaling, no data-path collective); one RCCL all-reduce sums the rate / PSNR statistics at the end of the timed region.

Prints ONE JSON line on rank 0 (see DESIGN.md section 6 for the fields).
"""
import argparse
import gc
import json
import os
import queue
import sys
import threading
import time

import numpy
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from autoencoder_based_image_compression_amd import device as dev            # noqa: E402
from autoencoder_based_image_compression_amd import pipeline                 # noqa: E402
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var   # noqa: E402
from autoencoder_based_image_compression_amd.kodak.lossless import compression as lossless_compression   # noqa: E402
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats   # noqa: E402
from autoencoder_based_image_compression_amd.kodak.tools import tools as tls   # noqa: E402

H_IN, W_IN = 512, 768          # Kodak luminance (datasets/kodak/kodak.py:10-83: uint8 (24, 512, 768))
IDX_MAP_EXCEPTION = 67         # lossless/results/1_10000/training_index_10/idx_map_exception.pkl
TRUNCATED_UNARY_LENGTH = 10    # collecting_stats_eae_extra.py:44
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup quota (the GPU box exposes 256 hardware
    threads behind a 16-CPU quota; OpenMP sized for 256 would only thrash)."""
    count = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            (quota, period) = f.read().split()
        if quota != 'max':
            count = min(count, max(1, int(int(quota)/int(period))))
    except (OSError, ValueError):
        pass
    return max(1, count)


def synthetic_images(seed, n, h, w):
    """RandomState(seed).randint(16, 236) low-pass filtered (3x box blur), uint8."""
    rng = numpy.random.RandomState(seed)
    x = rng.randint(16, 236, size=(n, h, w)).astype(numpy.float32)
    for _ in range(3):
        x = (x + numpy.roll(x, 1, 1) + numpy.roll(x, -1, 1) + numpy.roll(x, 1, 2) + numpy.roll(x, -1, 2))/numpy.float32(5.)
    return numpy.round(x).astype(numpy.uint8)


def synthetic_model(bin_width=1.):
    """Random-init weights of the fixed-bin-width architecture (no trained checkpoint exists in the reference mount)."""
    v = var.random_variables(bin_width, False, seed=0, bias_std=0.01)
    v['decoder/weights_6'] = (v['decoder/weights_6']*numpy.float32(30.)).astype(numpy.float32)   # leave the clip floor
    return v


class RateWorker(threading.Thread):
    """Host end of the rate measurement, off the launch thread: waits for a batch's (small) device -> host copy of the
    coder's per-map results and of the exception-map histograms, checks every status, sums the bit counts and forms the
    exception map's ceil(h*w*entropy) (compression.py:68-75) in numpy float64 like the reference."""

    def __init__(self, map_size, host_probabilities=None, host_threads=0):
        super(RateWorker, self).__init__(daemon=True)
        self.map_size = map_size
        self.host_probabilities = host_probabilities      # --coder host: the C-ABI host coder runs here
        self.host_threads = host_threads
        self.jobs = queue.Queue()
        self.coder_bits = 0
        self.exception_bits = 0
        self.dead_maps = 0
        self.error = None
        self.busy_s = 0.

    def run(self):
        while True:
            job = self.jobs.get()
            if job is None:
                return
            (event, results_host, hist_host, overflow_host, flags_host, checks_host, symbols_host, slot_free) = job
            try:
                event.synchronize()
                t0 = time.perf_counter()
                results = results_host.numpy()
                if symbols_host is not None:
                    # the north star's shape: ONE device -> host copy of the symbols, then the host C-ABI coder (encode +
                    # decode + compare per map like compress_lossless), threaded over maps
                    (_, nb_bits) = lossless_compression.code_planar_symbols(symbols_host.numpy(), self.host_probabilities,
                                                                           IDX_MAP_EXCEPTION, nb_threads=self.host_threads,
                                                                           roundtrip=True, verify_only=True)
                    results = numpy.zeros_like(results)
                    results[0] = nb_bits.reshape(-1)
                if results[2].any():
                    bad = int(numpy.flatnonzero(results[2])[0])
                    raise RuntimeError('device coder: map {0} failed with status {1} at stage {2}'.format(bad, results[2, bad], results[3, bad]))
                if int(overflow_host.numpy().sum()) != 0:
                    raise RuntimeError('exception-map symbols outside the histogram radius')
                if os.environ.get('EAE_CODER_DEBUG_CLOCKS'):
                    keep = results[1] > 0
                    sys.stderr.write('CLOCKS shader-cycles mean {:.0f} max {} refclk-ticks mean {:.0f} -> MHz {:.0f}\n'.format(
                        results[3][keep].mean(), results[3][keep].max(), results[1][keep].mean(),
                        100.*results[3][keep].mean()/results[1][keep].mean()))
                if int(checks_host[0]) != 0:
                    raise AssertionError('The rounded array elements cannot be represented as 16-bit signed integers.')
                self.dead_maps += int((flags_host.numpy() == 0).sum())        # tls.count_nb_deads (tools.py:294-320)
                self.coder_bits += int(results[0].astype(numpy.int64).sum()) + int(results[1].astype(numpy.int64).sum())
                self.exception_bits += sum(int(lossless_compression.exception_map_nb_bits(row, self.map_size))
                                           for row in hist_host.numpy().astype(numpy.int64))
                self.busy_s += time.perf_counter() - t0
            except Exception as exc:   # surfaced by the main thread
                self.error = exc
            finally:
                slot_free.set()


_STREAM_POOL = []


def run_pipeline(args, batch, steps, warmup, device, world, rank, cores, tracing, variables, coder_streams=None):
    """Builds the resident state for `batch` images per step, runs `warmup` untimed and `steps` timed steps, and returns
    what the report needs. Everything in here up to the first barrier is outside the timed region."""
    if world > 1:
        import torch.distributed as dist
    # ---- model, inputs, coder tables (outside the timed region) ---------------------------------------------------
    encoder = pipeline.DeviceEncoder(variables, False, device)
    decoder = pipeline.DeviceDecoder(variables, False, device)
    bin_widths = torch.from_numpy(variables[var.BIN_WIDTHS_NAME]).to(device)
    images = torch.from_numpy(synthetic_images(1000 + rank, batch, H_IN, W_IN)).to(device)
    (h_map, w_map) = (H_IN//16, W_IN//16)
    map_size = h_map*w_map
    # statistics that feed the coder, from this build's own a26/a27 path on the first batch (lossless/stats.py:306, :13-68)
    y0 = encoder(images)
    map_mean_host = dev.map_means(y0).cpu().numpy()
    probabilities = lossless_stats.compute_binary_probabilities(y0.cpu().numpy(), variables[var.BIN_WIDTHS_NAME], map_mean_host,
                                                                TRUNCATED_UNARY_LENGTH)
    map_mean = torch.from_numpy(map_mean_host).to(device)
    del y0
    probabilities_dev = torch.from_numpy(numpy.ascontiguousarray(probabilities, dtype=numpy.float64)).to(device)
    # row of `probabilities` per map of the batch; -1 = the exception map, costed from its histogram (compression.py:68-75)
    prob_row = torch.arange(128, dtype=torch.int32).repeat(batch)
    prob_row[IDX_MAP_EXCEPTION::128] = -1
    prob_row = prob_row.to(device)
    n_maps = batch*128
    nb_coder_streams = coder_streams or args.coder_streams
    nb_slots = nb_coder_streams + 2
    # everything the host needs from one batch, contiguous on the device:
    # [coder results 4 x n_maps | exception-map histograms | their overflow counts | non-zero flags of every map | 3 checks]
    nb_host_words = 4*n_maps + batch*511 + batch + n_maps + 3
    slot_out = [torch.zeros(nb_host_words, dtype=torch.int32, device=device) for _ in range(nb_slots)]
    pinned_out = [torch.zeros(nb_host_words, dtype=torch.int32).pin_memory() for _ in range(nb_slots)]

    def views(t):
        (a, b) = (4*n_maps, 4*n_maps + batch*511)
        return (t[:a].view(4, n_maps), t[a:b].view(batch, 511), t[b:b + batch],
                t[b + batch:b + batch + n_maps].view(batch, 128), t[b + batch + n_maps:])

    streams = [dev.CoderStreams(n_maps, map_size, TRUNCATED_UNARY_LENGTH, device, results=views(slot_out[i])[0]) for i in range(nb_slots)]
    slot_hist = [views(slot_out[i])[1:3] for i in range(nb_slots)]
    slot_flags = [views(slot_out[i])[3:] for i in range(nb_slots)]
    pinned_views = [views(pinned_out[i]) for i in range(nb_slots)]
    slot_symbols = [torch.empty((batch, 128, map_size), dtype=torch.int16, device=device) for _ in range(nb_slots)]
    workspaces = [dev.coder_workspace(n_maps, map_size, TRUNCATED_UNARY_LENGTH, device) for _ in range(nb_slots)]
    slot_free = [threading.Event() for _ in range(nb_slots)]
    for e in slot_free:
        e.set()
    host_coder = args.coder == 'host'
    coder_threads = args.coder_threads if args.coder_threads > 0 else max(1, cores//max(world, 1) - 2)
    worker = RateWorker(map_size, probabilities if host_coder else None, coder_threads)
    pinned_symbols = [torch.empty((batch, 128, map_size), dtype=torch.int16).pin_memory() if host_coder else None for _ in range(nb_slots)]
    worker.start()
    # HIP multiplexes streams onto 4 hardware queues: streams are reused across runs of this function so that a coder
    # stream never ends up sharing a queue with the transform stream
    while len(_STREAM_POOL) < nb_coder_streams:
        _STREAM_POOL.append(torch.cuda.Stream())
    coder_streams = _STREAM_POOL[:nb_coder_streams]
    sse_total = torch.zeros(batch, dtype=torch.int64, device=device)
    gemm_events = []            # (start, stop, launch name) around every conv_gemm launch of the timed region

    def timed_launch(name, fn, record):
        if not record:
            return fn()
        (a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        a.record()
        out = fn()
        b.record()
        gemm_events.append((a, b, name))
        return out

    host_marks = []

    def mark(tag):
        if tracing:
            host_marks.append((tag, time.perf_counter()))

    def step(index, record):
        mark('begin')
        v = encoder.v
        gdn_1 = dev.conv9x9s4_u8(images, encoder.w1, v['encoder/biases_1'], encoder.g[1], v['encoder/beta_1'])
        gdn_2 = timed_launch('conv2_gdn2', lambda: dev.conv5x5s2(gdn_1, encoder.w2, v['encoder/biases_2'], dev.NORM_GDN,
                                                                 encoder.g[2], v['encoder/beta_2']), record)
        y_raw = timed_launch('conv3', lambda: dev.conv5x5s2(gdn_2, encoder.w3, v['encoder/biases_3'], dev.NORM_NONE), record)
        mark('encoder')
        slot = index % nb_slots
        slot_free[slot].wait()
        slot_free[slot].clear()
        mark('slot')
        # buffers that cross to the coder streams are per-slot and preallocated (no caching-allocator traffic across streams)
        slot_out[slot][4*n_maps:].zero_()            # histograms, overflow, flags, checks: the kernels below accumulate into them
        # gdn_3 -> centre / quantise / symbols / dead-map flags -> de-centre -> inverse_gdn_4: one pass over the latents
        d = decoder.v
        q = dev.latent_stage(y_raw, bin_widths, map_mean, gdn_in=(encoder.g[3], v['encoder/beta_3']),
                             igdn_out=(decoder.g[4], d['decoder/beta_4']), want_symbols=True, want_flags=True,
                             out_symbols=slot_symbols[slot], out_flags=slot_flags[slot][0], out_checks=slot_flags[slot][1])
        # exception map of every image: exact histogram on the device; its entropy is formed on the host by the rate worker
        dev.symbol_histograms(q['symbols'].view(n_maps, map_size), 255, out=slot_hist[slot], first_map=IDX_MAP_EXCEPTION,
                              map_step=128, zero=False)
        quantized = torch.cuda.Event()
        quantized.record()
        # entropy coding off the transform stream, concurrent with the transforms of this and the next batches: every map
        # is encoded (streams left in HBM), then decoded back and compared in a second launch (what compress_lossless +
        # the assert of compression.py:146-153 do); then ONE small device -> host copy of the per-map bit counts /
        # statuses. The coder is a few latency-bound waves: several batches are kept in flight on separate streams.
        mark('quantize')
        symbols = q['symbols'].view(n_maps, map_size)
        coder_stream = coder_streams[index % len(coder_streams)]
        with torch.cuda.stream(coder_stream):
            coder_stream.wait_event(quantized)
            if host_coder:
                pinned_symbols[slot].copy_(q['symbols'], non_blocking=True)
            elif not os.environ.get('EAE_BENCH_NO_CODER'):    # diagnostic only: transforms without the coder
                if args.coder_lanes >= 0:                     # per-lane kernels (coder_device.hip), for comparison
                    dev.coder_compress_maps(symbols, probabilities_dev, prob_row, TRUNCATED_UNARY_LENGTH,
                                            mode=dev.CODER_ENCODE_ONLY, out=streams[slot], lanes_per_wave=args.coder_lanes)
                    dev.coder_verify_maps(streams[slot], symbols, probabilities_dev, prob_row, args.coder_lanes)
                else:                                         # 64 maps per wavefront in step (coder_simd.hip)
                    dev.coder_encode_batch(symbols, probabilities_dev, prob_row, TRUNCATED_UNARY_LENGTH, out=streams[slot],
                                           workspace=workspaces[slot])
                    mark('c_encode')
                    dev.coder_decode_batch(streams[slot], probabilities_dev, prob_row, expected=symbols, workspace=workspaces[slot])
                    mark('c_decode')
            dev.publish_to_host(slot_out[slot], pinned_out[slot])
            mark('c_copies')
            copied = torch.cuda.Event()
            copied.record()
        mark('coder')
        worker.jobs.put((copied,) + pinned_views[slot] + (pinned_symbols[slot], slot_free[slot]))
        t = timed_launch('tconv1_igdn5', lambda: dev.tconv5x5s2(q['t'], decoder.w4, d['decoder/biases_4'], dev.NORM_IGDN,
                                                                decoder.g[5], d['decoder/beta_5']), record)
        t = timed_launch('tconv2_igdn6', lambda: dev.tconv5x5s2(t, decoder.w5, d['decoder/biases_5'], dev.NORM_IGDN,
                                                                decoder.g[6], d['decoder/beta_6']), record)
        dev.tconv9x9s4_luma(t, decoder.w6, want_f32=False, want_u8=True, ref_u8=images, sse=sse_total)
        mark('decoder')

    def drain():
        torch.cuda.synchronize()
        for e in slot_free:
            e.wait()
        if worker.error is not None:
            raise worker.error

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i, False)
    drain()
    worker.coder_bits = 0
    worker.exception_bits = 0
    worker.dead_maps = 0
    sse_total.zero_()
    worker.busy_s = 0.

    # the launch thread allocates only short-lived wrappers: keep the cyclic collector (a 30 ms pause every ~75 steps) out of it
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    step_marks = []
    for i in range(steps):
        if tracing:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            step_marks.append((time.perf_counter() - t0, ev))
        step(i, True)
    drain()
    # the path's only exchange step: sum the rate / PSNR accumulators over ranks
    coder_bits = float(worker.coder_bits) + float(worker.exception_bits)
    stats = torch.tensor([coder_bits, float(sse_total.sum().item()), float(worker.dead_maps), float(steps*batch)],
                         dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    worker.jobs.put(None)
    return {'elapsed': elapsed, 'stats': stats, 'gemm_events': gemm_events, 'probabilities': probabilities,
            'map_mean_host': map_mean_host, 'host_coder': host_coder, 'coder_threads': coder_threads,
            'step_marks': step_marks, 'host_marks': host_marks}


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=100)
    parser.add_argument('--warmup', type=int, default=10)
    parser.add_argument('--batch', type=int, default=24, help='Kodak-sized images per GPU per step')
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--no-single-image', action='store_true', help='skip the one-image-per-step side measurement')
    parser.add_argument('--coder', choices=('device', 'host'), default='device',
                        help='device: the coder kernels on side streams (default). host: one device -> host copy of the symbols '
                             'per batch and the host C-ABI coder on a thread pool (the shape BASELINE.json sketches)')
    parser.add_argument('--coder-threads', type=int, default=0, help='host coder threads (0 = usable CPUs - 2)')
    parser.add_argument('--coder-streams', type=int, default=int(os.environ.get('EAE_CODER_STREAMS', '2')),
                        help='batches whose entropy coding may be in flight at once (each on its own HIP stream)')
    parser.add_argument('--coder-lanes', type=int, default=int(os.environ.get('EAE_CODER_LANES', '-1')),
                        help='-1: 64 maps per wavefront in step (default); >= 0: the per-lane kernels with that many maps per block')
    args = parser.parse_args()

    # two Python threads share the GIL (kernel launches; rate bookkeeping): hand it over quickly
    sys.setswitchinterval(1e-4)
    tracing = bool(os.environ.get('EAE_BENCH_TRACE'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU fallback for the product path.')
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', rank=rank, world_size=world)
    device = torch.device('cuda', local_rank)
    cores = usable_cpus()
    os.environ.setdefault('OMP_NUM_THREADS', str(cores))      # the oracle's OpenMP transforms (cpu_baseline only)

    variables = synthetic_model(1.)
    run = run_pipeline(args, args.batch, args.steps, args.warmup, device, world, rank, cores, tracing, variables)
    (elapsed, stats, gemm_events, probabilities, map_mean_host) = (run['elapsed'], run['stats'], run['gemm_events'],
                                                                    run['probabilities'], run['map_mean_host'])
    (host_coder, coder_threads, step_marks, host_marks) = (run['host_coder'], run['coder_threads'], run['step_marks'], run['host_marks'])

    # ---- derived figures (outside the timed region) ------------------------------------------------------------------
    pixels_per_step = args.batch*H_IN*W_IN
    total_pixels = pixels_per_step*args.steps*world
    value = total_pixels/elapsed/1e6
    nb_images_total = stats[3].item()
    bpp = stats[0].item()/(nb_images_total*H_IN*W_IN)
    mean_psnr = float(tls.psnr_from_sse(stats[1].item(), nb_images_total*H_IN*W_IN))   # PSNR of the pooled MSE
    flops = {'conv2_gdn2': pipeline.FLOP_PER_PIXEL['conv2_gdn2'], 'conv3': 2*1600,      # gdn_3 runs in the latent-stage kernel
             'tconv1_igdn5': pipeline.FLOP_PER_PIXEL['tconv1_igdn5'], 'tconv2_igdn6': pipeline.FLOP_PER_PIXEL['tconv2_igdn6']}
    per_launch_ms = {}
    for (a, b, name) in gemm_events:
        per_launch_ms.setdefault(name, []).append(a.elapsed_time(b))
    gemm_ms = sum(sum(v) for v in per_launch_ms.values())
    gemm_launches = sum(len(v) for v in per_launch_ms.values())
    gemm_flop = sum(flops[name]*pixels_per_step*len(v) for (name, v) in per_launch_ms.items())
    achieved = gemm_flop/(gemm_ms*1e-3)/1e12 if gemm_ms > 0 else 0.
    traffic = None
    traffic_file = os.path.join(ROOT, 'profiles', 'traffic_conv_gemm.json')
    if os.path.isfile(traffic_file):
        with open(traffic_file) as f:
            traffic = json.load(f).get('hbm_bytes_per_launch')
    line = {
        'metric': 'Mpixels/s encode+decode (Kodak 768x512 luma), bitstream bit-exact',
        'value': round(value, 3), 'unit': 'Mpixels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed/args.steps*1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'kodak_512x768_luma_batch{}_per_gpu_bin_width_1.0_lossless_roundtrip'.format(args.batch),
                   'images_per_gpu_per_step': args.batch, 'height': H_IN, 'width': W_IN, 'bin_width_multiplier': 1.0,
                   'truncated_unary_length': TRUNCATED_UNARY_LENGTH, 'idx_map_exception': IDX_MAP_EXCEPTION,
                   'weights': 'random-init fixed-bin-width architecture (trained checkpoints absent from the reference)',
                   'parallelism': 'image shards, one process per GPU' if world > 1 else 'single GPU',
                   'coder': 'device, 64 maps per wavefront, encode + decode + compare' if not host_coder else
                            'host C-ABI coder, {} threads, after one device -> host copy of the symbols'.format(coder_threads)},
        'images_per_s': round(nb_images_total/elapsed, 2),
        'rate_bpp': round(bpp, 5), 'psnr_db_pooled': round(mean_psnr, 4), 'dead_maps_per_image': round(stats[2].item()/nb_images_total, 3),
        'roofline': {'bound': 'mfma', 'kernel': 'conv_gemm_wave_kernel (conv2+GDN2, conv3, tconv1+IGDN5, tconv2+IGDN6)',
                     'achieved': round(achieved, 3), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(achieved/PEAK_F32_MFMA_TFLOPS, 4), 'traffic': traffic,
                     'avg_launch_ms': round(gemm_ms/max(gemm_launches, 1), 4),
                     'per_launch_ms': {k: round(sum(v)/len(v), 4) for (k, v) in per_launch_ms.items()},
                     'algorithmic_flop_per_launch': {k: flops[k]*pixels_per_step for k in flops}},
    }
    if step_marks:
        sections = {}
        slow = []
        for ((tag_a, t_a), (tag_b, t_b)) in zip(host_marks[:-1], host_marks[1:]):
            if tag_b == 'begin':
                continue
            sections.setdefault(tag_b, []).append((t_b - t_a)*1e3)
            if (t_b - t_a) > 4e-3:
                slow.append((tag_b, len(sections[tag_b]) - 1, round((t_b - t_a)*1e3, 1)))
        sys.stderr.write('TRACE host sections mean ms: {}\nTRACE host sections > 4 ms (section, step, ms): {}\n'.format(
            {k: round(sum(v_)/len(v_), 3) for (k, v_) in sections.items()}, slow))
        host = [round(m[0]*1e3, 2) for m in step_marks]
        gpu = [round(step_marks[0][1].elapsed_time(m[1]), 2) for m in step_marks]
        sys.stderr.write('TRACE host enqueue deltas (ms): {}\nTRACE gpu step deltas (ms): {}\nTRACE total ms {}\n'.format(
            [round(b - a, 1) for (a, b) in zip(host[:-1], host[1:])], [round(b - a, 1) for (a, b) in zip(gpu[:-1], gpu[1:])], round(elapsed*1e3, 2)))
    if rank == 0 and world == 1 and args.batch != 1 and not args.no_single_image:
        # BASELINE.json configs[1] is ONE Kodak image: the same path with one image per step (launch-bound, not the headline)
        del run, gemm_events
        one = run_pipeline(args, 1, 300, 30, device, world, rank, cores, False, variables, coder_streams=3)
        line['single_image'] = {'ms_per_image': round(one['elapsed']/300*1e3, 4),
                                'mpixels_per_s': round(300*H_IN*W_IN/one['elapsed']/1e6, 2), 'steps': 300, 'warmup': 30,
                                'note': 'one 512x768 image per step, steps pipelined back to back; host launch overhead dominates'}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(variables, probabilities, map_mean_host, cores)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(variables, probabilities, map_mean, cores):
    """The same path on the host cores, on a BOUNDED sample (checker code, timed only here, never shipped):
    transforms = oracle/transforms_oracle.c (plain-C restatement, OpenMP over all cores);
    coder = the reference's own C++ coder compiled into oracle/_ref (single thread, as the reference runs it),
    falling back to the oracle's C restatement when the reference build is absent; numpy quantiser / PSNR.
    One image calibrates, then as many images as fit in about 20 s of CPU work (2..64) are timed together."""
    from oracle import coder as oracle_coder
    from oracle import transforms as oracle_transforms
    import ctypes
    try:        # libgomp was initialised when torch was imported: set the team size for this thread explicitly
        ctypes.CDLL('libgomp.so.1').omp_set_num_threads(int(cores))
    except OSError:
        pass
    bw = variables[var.BIN_WIDTHS_NAME]
    kind_coder = 'ref' if oracle_coder.available('ref') else 'oracle'
    lib = oracle_coder.CoderLib(kind_coder)

    def run(n_img):
        x = synthetic_images(999, n_img, H_IN, W_IN)
        t = {}
        t0 = time.perf_counter()
        y = oracle_transforms.encoder(x.astype(numpy.float32)[..., None], variables, False)
        t['encoder_oracle_c_openmp'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        tiled = numpy.tile(bw.reshape(1, 1, 1, 128), y.shape[:3] + (1,))
        cq = tiled*numpy.round((y - map_mean)/tiled)
        sym = numpy.round(cq/tiled).astype(numpy.int16)
        t['quantiser_numpy'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        bits = 0
        for j in range(n_img):
            for c in range(128):
                if c == IDX_MAP_EXCEPTION:
                    continue
                (rec, nb) = lib.compress_lossless(numpy.ascontiguousarray(sym[j, :, :, c]).reshape(-1), probabilities[c])
                bits += nb
        t['coder_{}_single_thread'.format('reference_cpp' if kind_coder == 'ref' else 'oracle_c')] = time.perf_counter() - t0
        t0 = time.perf_counter()
        rec = oracle_transforms.decoder(cq + map_mean, variables, False)[..., 0]
        rec_u8 = numpy.round(rec.clip(min=16., max=235.)).astype(numpy.uint8)
        mse = numpy.mean((x.astype(numpy.float64) - rec_u8.astype(numpy.float64))**2)
        t['decoder_oracle_c_openmp_plus_psnr'] = time.perf_counter() - t0
        return (t, bits, float(mse))

    (t1, _, _) = run(1)
    n_img = int(max(2, min(64, round(20./max(sum(t1.values()), 1e-3)))))
    (t, bits, mse) = run(n_img)
    total = sum(t.values())
    return {'value': round(n_img*H_IN*W_IN/total/1e6, 4), 'unit': 'Mpixels/s', 'cores': cores, 'kind': 'port',
            'sample': '{} synthetic 512x768 images, encode+quantise+code(enc+dec)+decode+PSNR, {:.1f} s of CPU work; transforms '
                      'OpenMP on the {} usable CPUs, coder single-threaded like the reference'.format(n_img, total, cores),
            'seconds': {k: round(val, 3) for (k, val) in t.items()}, 'bits': int(bits), 'mse': round(mse, 4)}


if __name__ == '__main__':
    main()
