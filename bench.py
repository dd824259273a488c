#!/usr/bin/env python
"""bench.py -- Mpixels/s of the compression inference path (encode -> quantise -> entropy-code -> decode) on MI355X.

One STEP = one `codec.BatchCodec.submit`: one pass of the whole hot path over one batch of Kodak-sized (512x768) synthetic luminance images that
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
import math
import os
import pickle
import socket
import subprocess
import sys
import time


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1,
                        help='ranks = GPUs of this node. N > 1 without a launcher (no WORLD_SIZE in the environment): this '
                             'process starts N rank processes itself and waits for them; under torch.distributed.run it must '
                             'equal WORLD_SIZE. configs[3] of BASELINE.json is `--gpus 8 --height 256 --width 256 --batch 64`')
    parser.add_argument('--steps', type=int, default=100)
    parser.add_argument('--warmup', type=int, default=10)
    parser.add_argument('--min-seconds', type=float, default=1.0,
                        help='the block of --steps timed steps is repeated until this much time has been timed in total '
                             '(every block bracketed by barrier + synchronize); the MEDIAN block is reported')
    parser.add_argument('--max-blocks', type=int, default=25)
    parser.add_argument('--batch', type=int, default=24, help='images per GPU per step (default: the Kodak set)')
    parser.add_argument('--height', type=int, default=512, help='image height (default 512: Kodak)')
    parser.add_argument('--width', type=int, default=768, help='image width (default 768: Kodak); e.g. --height 256 --width 256 --batch 64 is one rank of BASELINE.json configs[3]')
    parser.add_argument('--bin-width', type=float, default=1.0,
                        help='quantisation bin width of the headline run (1.0: BASELINE.json configs[1]). Random-init weights '
                             'give a low-entropy latent at 1.0; the `realistic_entropy` side figures repeat the run at smaller '
                             'widths (more bits per pixel for the coder)')
    parser.add_argument('--kodak-npy', default=None, metavar='PATH',
                        help='real images instead of synthetic ones: the array `datasets/kodak/kodak.py:66-83` of the reference writes '
                             '(`kodak.npy`: uint8 (24, 512, 768) luminances; any uint8 (N, H, W) with H, W multiples of 16 is accepted). '
                             '--batch / --height / --width follow the array')
    parser.add_argument('--checkpoint', default=None, metavar='PATH',
                        help='a trained model instead of random-init weights: the ".ckpt" prefix `Saver.restore` takes '
                             '(eae/graph/EntropyAutoencoder.py:452-458; TensorFlow V1 / V2, read without TensorFlow) or an .npz keyed by '
                             'the TF variable names. --bin-width then multiplies the checkpoint\'s bin widths (reconstructing_eae_kodak.py:184)')
    parser.add_argument('--learned-bin-widths', action='store_true', help='--checkpoint is a learned-bin-width model (no GDN3 / IGDN4)')
    parser.add_argument('--stats-dir', default=None, metavar='PATH',
                        help='the coder\'s statistics instead of ones computed from the first batch: a directory with `map_mean.npy`, '
                             '`idx_map_exception.pkl` and `binary_probabilities_<multiplier>.npy` (lossless/stats.py:243-320; e.g. the '
                             'reference\'s lossless/results/1_10000/training_index_10/)')
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--no-transforms-alone', action='store_true',
                        help='skip the second launch-by-launch leg (the transforms without the coder: `roofline.transforms_alone`); the profile commands of '
                             'scratch/r06/collect_final.sh pass it so that a kernel\'s rocprofv3 average is over the launches `roofline` itself times')
    parser.add_argument('--no-dropin-surface', action='store_true',
                        help='skip the `dropin_surface` leg (the mirror of the reference\'s fix_gamma through the reference\'s own call '
                             'surface: numpy in, numpy out, batch_size 4)')
    parser.add_argument('--only-single-image-pipelined', action='store_true',
                        help='only the pipelined one-image-per-step figure of `single_image` (the default run starts this in a process of its own)')
    parser.add_argument('--only-library-user', action='store_true',
                        help='the headline workload as a user of the package gets it: GPU_MAX_HW_QUEUES not set by this file (the package asks for '
                             'its queues itself at import), a default product-mode BatchCodec; prints {"library_user": ...} (main() runs this in a '
                             'child process whose environment does not hold the variable)')
    parser.add_argument('--only-dropin-surface', action='store_true',
                        help='run the `dropin_surface` leg alone and print it (diagnostic: no headline, no roofline)')
    parser.add_argument('--no-single-image', '--no-side', dest='no_single_image', action='store_true',
                        help='skip the side measurements (one image per step, other shapes, PCIe-inclusive, realistic entropy, host coder)')
    parser.add_argument('--coder', choices=('device', 'host'), default='device',
                        help='device: the coder kernels on side streams (default). host: one device -> host copy of the symbols '
                             'per batch and the host C-ABI coder on a thread pool (the shape BASELINE.json sketches)')
    parser.add_argument('--coder-threads', type=int, default=0, help='host coder threads (0 = usable CPUs - 2)')
    parser.add_argument('--coder-streams', type=int, default=int(os.environ.get('EAE_CODER_STREAMS', '0')),
                        help='batches whose entropy coding may be in flight at once (each on its own HIP stream). 0 (default): '
                             'decided from the shape (`auto_coder_streams`: a map is one serial chain, so large maps need more '
                             'batches in flight for the transforms to cover it)')
    parser.add_argument('--transform-streams', type=int, default=int(os.environ.get('EAE_TRANSFORM_STREAMS', '0')),
                        help='0 (default): codec.PRODUCT_TRANSFORM_STREAMS = 3, the product mode: consecutive batches go round three private streams, so the tail of '
                             'one batch\'s kernel is filled by the next batches\' (2 / 3 / 4 / 5 streams: 3,020 / 3,075 / 3,060 / '
                             '3,030 Mpx/s at the default shape, 2,550 / 2,770 / 2,740 for 64 x 256x256: profiles/r03_transform_streams2.txt). '
                             '1: the transforms of consecutive batches back to back on one stream (what the `roofline` leg always '
                             'uses, so that the HIP events around a launch time that kernel alone)')
    parser.add_argument('--fuse-latent', action='store_true',
                        help='the latent stage as the epilogue of the conv_3 launch (codec.BatchCodec(fuse_latent=True)); the roofline '
                             'figure then counts gdn_3 and inverse_gdn_4 in that launch')
    parser.add_argument('--graphs', dest='graphs', action='store_true', default=True,
                        help='(default) replay three captured hipGraphs per step instead of launching kernel by kernel')
    parser.add_argument('--no-graphs', dest='graphs', action='store_false')
    parser.add_argument('--seed-offset', type=int, default=0, help='rank r codes the images of seed 1000 + r + this (tests)')
    parser.add_argument('--force-nccl', action='store_true',
                        help='join an RCCL process group even as a single rank, so that the barriers and the one statistics '
                             'all-reduce of the N-GPU run go through librccl on a one-GPU box (tests/test_bench_launcher.py)')
    parser.add_argument('--exchange', choices=('all_reduce', 'all_gather'), default='all_reduce',
                        help='the path\'s one exchange step at the end of every timed block (SURVEY.md 8(e)): all_reduce sums four float64 accumulators '
                             '(default); all_gather is the exact-parity mode -- every rank receives the per-image (bits, squared error, dead maps) '
                             'of all ranks in global image order (`sharding.gather_per_image`), so a mean over images is bit-identical to one process\'s')
    parser.add_argument('--dry-launch', action='store_true',
                        help='rendezvous check only (no GPU): every rank joins a gloo group, one all-reduce, rank 0 prints n_gpus')
    args = parser.parse_args(argv)
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        parser.error('--gpus and --steps must be at least 1, --warmup at least 0')
    if args.coder == 'host':
        args.graphs = False          # the host coder's copy is not part of a captured step
    return args


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher around it: start N FRESH rank processes (one per GPU; RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR, MASTER_PORT in their environment), wait for all of them and return the first non-zero exit
    code. This process has not touched the GPU (torch is not even imported yet) and never does; nothing is re-exec'ed."""
    port = free_port()
    children = []
    for rank in range(args.gpus):
        env = dict(os.environ)
        env.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(args.gpus), 'LOCAL_WORLD_SIZE': str(args.gpus),
                    'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'EAE_BENCH_SPAWNED': '1'})
        # The pool's host driver only supports dmabuf IPC: with the legacy mode RCCL's intra-node transport fails in
        # hipIpcGetMemHandle ("invalid argument"). The image exports this already; a caller's own setting wins.
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    code = 0
    try:
        pending = list(children)
        while pending:
            for child in list(pending):
                rc = child.poll()
                if rc is None:
                    continue
                pending.remove(child)
                if rc != 0 and code == 0:
                    code = rc
                    for other in pending:       # a rank died: the others would wait in a collective for ever
                        other.terminate()
            time.sleep(0.05)
    finally:
        for child in children:
            if child.poll() is None:
                child.kill()
    return code


if __name__ == '__main__':
    _ARGS = parse_args()
    if _ARGS.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(_ARGS, sys.argv[1:]))

# The HIP runtime multiplexes streams onto 4 hardware queues by default; streams that land on the same queue serialise.
# With three transform streams, 3-8 coder streams and copy streams that aliasing was measured to cost up to 30 % (and the
# one-image-per-step leg keeps 14 streams busy). Must be set before the runtime initialises; an explicit setting of the
# caller wins. (The package does the same for itself at import -- autoencoder_based_image_compression_amd/__init__.py --; `--only-library-user`
# leaves it to the package, to measure what a user who only imports it gets.)
if '--only-library-user' not in sys.argv:
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
# The pool's host driver only supports dmabuf IPC (RCCL's intra-node transport fails in hipIpcGetMemHandle with the legacy mode).
# ROCr reads this when it initialises, so it is set here, before torch is imported; launch_ranks gives it to its children too.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy          # noqa: E402
import torch          # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from autoencoder_based_image_compression_amd import codec                    # noqa: E402
from autoencoder_based_image_compression_amd import device as dev            # noqa: E402
from autoencoder_based_image_compression_amd import pipeline                 # noqa: E402
from autoencoder_based_image_compression_amd.kodak.eae.graph import variables as var   # noqa: E402
from autoencoder_based_image_compression_amd.kodak.lossless import stats as lossless_stats   # noqa: E402
from autoencoder_based_image_compression_amd.kodak.tools import tools as tls   # noqa: E402

IDX_MAP_EXCEPTION = 67         # lossless/results/1_10000/training_index_10/idx_map_exception.pkl
TRUNCATED_UNARY_LENGTH = 10    # collecting_stats_eae_extra.py:44
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_TBS = 8.0             # MI355X_MICROARCH.md: HBM3E
# bin widths of the `realistic_entropy` side figures: with the random-init weights of `synthetic_model` they put the rate
# near 1, 1.4, 2 and 3 bits per pixel, the range SURVEY.md 8(d) expects from trained models (the rates are measured and printed)
REALISTIC_BIN_WIDTHS = (0.25, 0.125, 0.05, 0.0125)
# Algorithmic work of every timed launch per INPUT pixel: FLOP (pipeline.FLOP_PER_PIXEL) and HBM bytes of the
# layer-boundary model (fp32 activations in and out once, uint8 image / reconstruction, int16 symbols)
BYTES_PER_PIXEL = {'conv1_gdn1': 1 + 32, 'conv2_gdn2': 32 + 8, 'conv3': 8 + 2, 'latent': 2 + 2 + 1, 'tconv1_igdn5': 2 + 8,
                   'tconv2_igdn6': 8 + 32, 'tconv3': 32 + 1 + 1}
GEMM_LAUNCHES = ('conv2_gdn2', 'conv3', 'tconv1_igdn5', 'tconv2_igdn6')


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


def auto_coder_streams(h, w):
    """Batches of coder work to keep in flight: the product's own default (codec.default_nb_in_flight)."""
    return codec.default_nb_in_flight(h, w)


class _IntegerOnlyUnpickler(pickle.Unpickler):
    """`idx_map_exception.pkl` is one pickled integer (lossless/stats.py:186-189 of the reference); a pickle can name any callable, and
    a statistics directory may come from anywhere: nothing is looked up, so a file that asks for a class or a function is refused."""

    def find_class(self, module, name):
        raise pickle.UnpicklingError('idx_map_exception.pkl must hold one integer, not a reference to {0}.{1}'.format(module, name))


def load_pickled_int(path):
    with open(path, 'rb') as f:
        value = _IntegerOnlyUnpickler(f).load()
    if isinstance(value, bool) or not isinstance(value, (int, numpy.integer)):
        raise SystemExit('bench.py: {0} must hold one integer (got {1})'.format(path, type(value).__name__))
    return int(value)


def load_inputs(args):
    """What --kodak-npy / --checkpoint / --stats-dir name, checked like the reference's loaders check it; None where a flag is absent.
    Returns {'images', 'variables', 'statistics': (map_mean, probabilities, idx_map_exception) or None, 'data': label for the line}."""
    import pickle
    out = {'images': None, 'variables': None, 'statistics': None, 'data': 'synthetic'}
    labels = []
    if args.kodak_npy:
        images = numpy.load(args.kodak_npy)
        if images.dtype != numpy.uint8 or images.ndim != 3:
            raise SystemExit('bench.py: --kodak-npy must hold uint8 luminances of shape (N, H, W) (datasets/kodak/kodak.py:66-83), '
                             'not {0} {1}'.format(images.dtype, images.shape))
        if images.shape[1] % 16 or images.shape[2] % 16:
            raise SystemExit('bench.py: --kodak-npy: height and width must be multiples of 16 (EntropyAutoencoder.py:77-80)')
        out['images'] = numpy.ascontiguousarray(images)
        labels.append('images: {0} ({1} x {2}x{3})'.format(os.path.basename(args.kodak_npy), *images.shape))
    if args.checkpoint:
        variables = var.restore_variables(args.checkpoint, bool(args.learned_bin_widths))
        variables[var.BIN_WIDTHS_NAME] = (args.bin_width*variables[var.BIN_WIDTHS_NAME]).astype(numpy.float32)
        out['variables'] = variables
        labels.append('weights: {}'.format(os.path.basename(args.checkpoint)))
    if args.stats_dir:
        map_mean = numpy.load(os.path.join(args.stats_dir, 'map_mean.npy'))
        idx_map_exception = load_pickled_int(os.path.join(args.stats_dir, 'idx_map_exception.pkl'))
        probabilities = numpy.load(os.path.join(args.stats_dir, 'binary_probabilities_{}.npy'.format(tls.float_to_str(float(args.bin_width)))))
        if map_mean.shape != (128,) or probabilities.ndim != 2 or probabilities.shape[0] != 128:
            raise SystemExit('bench.py: --stats-dir: map_mean.npy must be (128,) and binary_probabilities_*.npy (128, L)')
        out['statistics'] = (map_mean.astype(numpy.float32), numpy.ascontiguousarray(probabilities, dtype=numpy.float64), idx_map_exception)
        labels.append('coder statistics: {}'.format(args.stats_dir))
    if labels:
        out['data'] = '; '.join(labels) + ('' if (args.kodak_npy and args.checkpoint) else '; the rest synthetic')
    return out


class Context(object):
    """What every leg of the benchmark shares: the process group, the device, the CPU budget."""

    def __init__(self, args, device, world, rank, cores):
        (self.args, self.device, self.world, self.rank, self.cores) = (args, device, world, rank, cores)
        self.grouped = world > 1 or bool(getattr(args, 'force_nccl', False))      # a process group exists
        self.collectives = 0                                                       # barriers + all-reduces that went through it
        self.inputs = {'images': None, 'variables': None, 'statistics': None, 'data': 'synthetic'}

    def images(self, seed, batch, h, w):
        """uint8 (batch, h, w): the caller's images (--kodak-npy) when they have that size and there are enough of them -- every rank
        then codes the same images --, else the synthetic ones of `seed`."""
        mine = self.inputs['images']
        if mine is not None and mine.shape[1:] == (h, w) and mine.shape[0] >= batch:
            return mine[:batch]
        return synthetic_images(seed, batch, h, w)

    def statistics(self, y0, bin_widths, use_given=True):
        """(map_mean, probabilities, idx_map_exception): --stats-dir, else this build's own a26 / a27 path on the first batch's latents
        (lossless/stats.py:306, :13-68) and the reference model's exception index."""
        if use_given and self.inputs['statistics'] is not None:
            return self.inputs['statistics']
        map_mean_host = dev.map_means(y0).cpu().numpy()
        probabilities = lossless_stats.compute_binary_probabilities(y0.cpu().numpy(), bin_widths, map_mean_host, TRUNCATED_UNARY_LENGTH)
        return (map_mean_host, probabilities, IDX_MAP_EXCEPTION)

    def barrier(self):
        if self.grouped:
            self.collectives += 1
            import torch.distributed as dist
            if dist.get_backend() == 'nccl':
                dist.barrier(device_ids=[self.device.index])      # RCCL: name the device, no guess from the rank
            else:
                dist.barrier()
        torch.cuda.synchronize()

    def gather_per_image(self, local, nb_images_total):
        """float64 (images of this rank, k) -> (nb_images_total, k) on every rank, global image order (`sharding.gather_per_image`:
        all_gather through RCCL on the device, through gloo on the host)."""
        if not self.grouped:
            return local
        self.collectives += 1
        import torch.distributed as dist
        from autoencoder_based_image_compression_amd import sharding
        return sharding.gather_per_image(local, nb_images_total, device=self.device if dist.get_backend() == 'nccl' else None)

    def all_reduce(self, tensor, op):
        if self.grouped:
            self.collectives += 1
            import torch.distributed as dist
            if dist.get_backend() != 'nccl':      # gloo (the shared-GPU test hook) reduces host tensors
                host = tensor.cpu()
                dist.all_reduce(host, op=getattr(dist.ReduceOp, op))
                tensor.copy_(host)
            else:
                dist.all_reduce(tensor, op=getattr(dist.ReduceOp, op))
        return tensor


def run_pipeline(ctx, batch, steps, warmup, variables, h, w, coder='device', coder_streams=3, transform_streams=1, use_graphs=False,
                 min_seconds=0., max_blocks=1, record=False, coder_events=False, pcie=False, serial=False, given_statistics=True,
                 statistics=None, one_stream_steps=False):
    """Builds the resident state for `batch` images of h x w per step (codec.BatchCodec: weights, tables, per-slot buffers),
    runs `warmup` untimed steps, then BLOCKS of exactly `steps` timed steps -- each block bracketed by barrier + synchronize
    on both sides, its wall time the MAX over ranks -- until `min_seconds` have been timed (at most `max_blocks` blocks; the
    count is the same on every rank because it is decided from the all-reduced times). Everything up to the first barrier
    is outside the timed region.
    record: HIP events around every named launch of the step (launch-by-launch path, one transform stream: a launch then has
    the GPU to itself apart from the coder's side streams). pcie: the images of every step come from pinned host memory (uint8,
    one async copy on a copy stream) and the uint8 reconstructions go back to pinned host memory (the feed / fetch of the
    reference's `sess.run`, eae/batching.py:95-99, 49-53). serial: every step is waited for before the next is submitted (the
    latency of one step on an otherwise idle GPU instead of the throughput of the pipeline)."""
    args = ctx.args
    (device, world, rank) = (ctx.device, ctx.world, ctx.rank)
    images_host = torch.from_numpy(numpy.ascontiguousarray(ctx.images(1000 + rank + args.seed_offset, batch, h, w)))
    images = images_host.to(device)
    bin_widths = variables[var.BIN_WIDTHS_NAME]
    learned = var.ENCODER_NAMES_FIXED_BW[0] not in variables
    # statistics that feed the coder: the caller's (--stats-dir; `given_statistics` False for the legs at other bin widths, whose
    # tables the directory does not hold), `statistics` (a leg's own), else from the first batch
    if statistics is not None:
        (map_mean_host, probabilities, idx_map_exception) = statistics
    else:
        encoder = pipeline.DeviceEncoder(variables, learned, device)
        y0 = encoder(images)
        (map_mean_host, probabilities, idx_map_exception) = ctx.statistics(y0, bin_widths, use_given=given_statistics)
        encoder.check()
        del y0, encoder
    events = []            # (start, stop, launch name) around every named launch of the timed region
    recording = [False]

    def timed_launch(name, fn):
        if not recording[0]:
            return fn()
        (a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        a.record()
        out = fn()
        b.record()
        events.append((a, b, name))
        return out

    coder_mode = 'none' if os.environ.get('EAE_BENCH_NO_CODER') else coder      # 'none': diagnostic only
    coder_threads = args.coder_threads if args.coder_threads > 0 else max(1, ctx.cores//max(world, 1) - 2)
    with codec.BatchCodec(variables, learned, bin_widths, map_mean_host, probabilities, idx_map_exception, batch, h, w,
                          device=device, nb_in_flight=coder_streams, launch_hook=timed_launch if record else None,
                          coder=coder_mode, host_coder_threads=coder_threads, nb_transform_streams=transform_streams,
                          use_graphs=use_graphs, time_coder=coder_events, fuse_latent=args.fuse_latent,
                          fetch_reconstruction=pcie, one_stream_steps=one_stream_steps) as the_codec:
        if pcie:
            # the codec's own feed and fetch: a pinned host batch in (copied on its feed stream), the reconstruction back to
            # pinned host memory (copied by its result worker once the batch is decoded)
            pinned_in = images_host.pin_memory()

            def submit():
                return the_codec.submit(pinned_in)
        else:
            def submit():
                return the_codec.submit(images)
        for _ in range(warmup):
            submit()
        the_codec.drain()
        # the launch thread allocates only short-lived wrappers: keep the cyclic collector (a 30 ms pause every ~75 steps) out of it
        gc.collect()
        gc.disable()
        try:
            recording[0] = True
            (block_seconds, block_cpu) = ([], [])
            stats = torch.zeros(4, dtype=torch.float64, device=device)
            coder_ms = []
            while True:
                ctx.barrier()
                t0 = time.perf_counter()
                c0 = time.process_time()
                tickets = []
                for _ in range(steps):
                    tickets.append(submit())
                    if serial:
                        tickets[-1].result()
                the_codec.drain()
                results = [t.result() for t in tickets]          # raises here if any map of any batch failed
                # (`reconstruction_host` is the slot's pinned buffer: the contents belong to the ticket only until its slot is
                # submitted again, so all that can be checked once the block is through is that every copy was made)
                if pcie and any(t.reconstruction_host is None for t in tickets):
                    raise RuntimeError('a reconstruction did not reach the host')
                # the path's only exchange step: sum the rate / PSNR accumulators over ranks ...
                if args.exchange == 'all_reduce':
                    block_stats = torch.tensor([float(sum(int(r['nb_bits'].sum()) for r in results)), float(sum(int(r['sse'].sum()) for r in results)),
                                                float(sum(int(r['nb_deads'].sum()) for r in results)), float(steps*batch)],
                                               dtype=torch.float64, device=device)
                    ctx.all_reduce(block_stats, 'SUM')
                else:
                    # ... or, exact-parity mode, gather the per-image values of every rank in global image order
                    local = numpy.concatenate([numpy.stack([r['nb_bits'], r['sse'], r['nb_deads']], axis=1) for r in results]).astype(numpy.float64)
                    everything = ctx.gather_per_image(local, steps*batch*max(world, 1))
                    block_stats = torch.tensor(list(everything.sum(axis=0)) + [float(everything.shape[0])], dtype=torch.float64, device=device)
                ctx.barrier()
                elapsed = time.perf_counter() - t0
                cpu = time.process_time() - c0
                te = ctx.all_reduce(torch.tensor([elapsed], dtype=torch.float64, device=device), 'MAX')
                block_seconds.append(float(te.item()))
                block_cpu.append(cpu)
                stats += block_stats
                coder_ms.extend(t.coder_ms() for t in tickets if coder_events)
                if sum(block_seconds) >= min_seconds or len(block_seconds) >= max_blocks:
                    break
        finally:
            gc.enable()
    order = sorted(range(len(block_seconds)), key=lambda i: block_seconds[i])
    mid = order[(len(order) - 1)//2]      # an actual block (the lower median when the count is even)
    # host CPU of this process (all its threads: launch thread, result worker, runtime helpers) per step, every rank's figure
    cpu_ms = torch.zeros(max(world, 1), dtype=torch.float64, device=device)
    cpu_ms[rank] = block_cpu[mid]/steps*1e3
    ctx.all_reduce(cpu_ms, 'SUM')
    return {'elapsed': block_seconds[mid], 'block_seconds': block_seconds, 'stats': stats, 'events': events,
            'probabilities': probabilities, 'map_mean_host': map_mean_host, 'idx_map_exception': idx_map_exception, 'host_coder': coder_mode == 'host',
            'coder_threads': coder_threads, 'coder_ms': coder_ms, 'host_cpu_ms_per_step': [round(float(v), 4) for v in cpu_ms.tolist()]}


def rate_and_psnr(stats, h, w):
    nb_images = stats[3].item()
    bpp = stats[0].item()/(nb_images*h*w)
    psnr = float(tls.psnr_from_sse(stats[1].item(), nb_images*h*w))   # PSNR of the pooled MSE
    return (bpp, psnr)


def coder_alone_ms(ctx, batch, variables, h, w, repeats=20):
    """The coder chain of one batch (binarise + encode, decode + compare) timed with HIP events on an otherwise idle GPU:
    the serial depth the transforms of the batches in flight have to cover."""
    device = ctx.device
    images = torch.from_numpy(synthetic_images(1000 + ctx.rank, batch, h, w)).to(device)
    bin_widths = variables[var.BIN_WIDTHS_NAME]
    y = pipeline.DeviceEncoder(variables, False, device)(images)
    map_mean = dev.map_means(y)
    probabilities = lossless_stats.compute_binary_probabilities(y.cpu().numpy(), bin_widths, map_mean.cpu().numpy(), TRUNCATED_UNARY_LENGTH)
    q = dev.quantize_maps(y, torch.from_numpy(bin_widths).to(device), map_mean, want_symbols=True)
    symbols = q['symbols'].reshape(batch*128, -1)
    prob_row = torch.arange(128, dtype=torch.int32).repeat(batch)
    prob_row[IDX_MAP_EXCEPTION::128] = -1
    prob_row = prob_row.to(device)
    prob = torch.from_numpy(probabilities).to(device)
    streams = dev.CoderStreams(batch*128, symbols.shape[1], TRUNCATED_UNARY_LENGTH, device)
    workspace = dev.coder_workspace(batch*128, symbols.shape[1], TRUNCATED_UNARY_LENGTH, device)
    (enc_ms, dec_ms) = ([], [])
    for i in range(repeats + 3):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        marks[0].record()
        dev.coder_encode_batch(symbols, prob, prob_row, TRUNCATED_UNARY_LENGTH, out=streams, workspace=workspace)
        marks[1].record()
        dev.coder_decode_batch(streams, prob, prob_row, expected=symbols, workspace=workspace)
        marks[2].record()
        torch.cuda.synchronize()
        if i >= 3:
            enc_ms.append(marks[0].elapsed_time(marks[1]))
            dec_ms.append(marks[1].elapsed_time(marks[2]))
    results = streams.results.cpu().numpy()
    if results[2].any():
        raise RuntimeError('the coder reported a status on the benchmark symbols')
    return (sorted(enc_ms)[len(enc_ms)//2], sorted(dec_ms)[len(dec_ms)//2])


def launch_rooflines(run_events, pixels_per_step, fuse_latent, map_symbols):
    """Per-launch figures from the HIP events of the roofline leg: average duration, algorithmic FLOP/s and B/s (SURVEY.md
    8(d)'s per-pixel figures x the pixels of a launch), the larger of the two fractions of peak, and which roof that is."""
    flops = dict(pipeline.FLOP_PER_PIXEL)
    flops = {'conv1_gdn1': flops['conv1_gdn1'], 'conv2_gdn2': flops['conv2_gdn2'],
             # gdn_3 and inverse_gdn_4 run in the latent-stage kernel unless that stage is the conv_3 launch's epilogue
             'conv3': 2*1600 + (flops['conv3_gdn3'] - 2*1600 + flops['igdn4'] if fuse_latent else 0),
             'latent': flops['conv3_gdn3'] - 2*1600 + flops['igdn4'],
             'tconv1_igdn5': flops['tconv1_igdn5'], 'tconv2_igdn6': flops['tconv2_igdn6'], 'tconv3': flops['tconv3']}
    per_launch_ms = {}
    for (a, b, name) in run_events:
        per_launch_ms.setdefault(name, []).append(a.elapsed_time(b))
    out = {}
    for (name, values) in per_launch_ms.items():
        ms = sum(values)/len(values)
        entry = {'avg_ms': round(ms, 4), 'launches': len(values)}
        if name in flops:
            tf = flops[name]*pixels_per_step/(ms*1e-3)/1e12
            tb = BYTES_PER_PIXEL[name]*pixels_per_step/(ms*1e-3)/1e12
            (f_frac, b_frac) = (tf/PEAK_F32_MFMA_TFLOPS, tb/PEAK_HBM_TBS)
            entry.update({'flop_per_px': flops[name], 'bytes_per_px': BYTES_PER_PIXEL[name], 'tflops': round(tf, 2), 'tbytes_per_s': round(tb, 3),
                          'frac_mfma': round(f_frac, 4), 'frac_hbm': round(b_frac, 4), 'bound': 'mfma' if f_frac >= b_frac else 'hbm',
                          'frac': round(max(f_frac, b_frac), 4)})
        else:
            # the coder: a serial chain per map (64 maps per wavefront in step); neither roof applies, the figure is symbols / s
            entry.update({'bound': 'latency (one serial chain per feature map)', 'msymbols_per_s': round(map_symbols/(ms*1e-3)/1e6, 2),
                          'bytes_per_symbol': 2 if name == 'coder_encode' else 4,
                          'frac_hbm': round((2 if name == 'coder_encode' else 4)*map_symbols/(ms*1e-3)/1e12/PEAK_HBM_TBS, 6)})
        out[name] = entry
    return (out, per_launch_ms, flops)


def children_alone_on_the_gpu(args, h_in, w_in):
    """The two side figures that are measured in processes of their own, run one after the other BEFORE this process initialises the
    GPU, so that each has the GPU to itself: (1) `library_user`: what a user of the package gets who sets nothing -- a child whose
    environment does not hold GPU_MAX_HW_QUEUES imports the package (which asks for its queues itself) and runs the headline workload
    in the default product mode; (2) `single_image_pipelined`: one image per step with twenty steps in flight (twenty busy streams).
    Round 6 measured why they must not run while this process holds its context: the same `library_user` child gave 2,928 Mpx/s
    behind the parent's legs (sixteen hardware queues of a live process next to its own) and 3,153-3,160 alone
    (profiles/r06_library_user.md). Children, never an exec."""
    out = {}
    same_model = ['--bin-width', repr(args.bin_width)] + (['--checkpoint', args.checkpoint] if args.checkpoint else []) + (
        ['--learned-bin-widths'] if args.learned_bin_widths else [])
    try:
        env = {k: v for (k, v) in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
        child = subprocess.run([sys.executable, os.path.abspath(__file__), '--only-library-user', '--no-cpu-baseline', '--steps', str(args.steps),
                                '--warmup', str(args.warmup), '--batch', str(args.batch), '--height', str(h_in), '--width', str(w_in)] + same_model,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, check=True, text=True, env=env)
        out['library_user'] = json.loads([ln for ln in child.stdout.splitlines() if ln.startswith('{')][-1])['library_user']
    except Exception as exc:      # a side figure must never cost the run its headline
        out['library_user'] = {'error': '{0}: {1}'.format(type(exc).__name__, exc)}
    try:
        child = subprocess.run([sys.executable, os.path.abspath(__file__), '--only-single-image-pipelined', '--no-cpu-baseline'] + same_model,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, check=True, text=True,
                               env=dict(os.environ, GPU_MAX_HW_QUEUES='24'))      # twenty busy streams + the default one
        out['single_image_pipelined'] = json.loads([ln for ln in child.stdout.splitlines() if ln.startswith('{')][-1])['single_image_pipelined']
    except Exception as exc:
        out['single_image_pipelined'] = {'pipelined_error': '{0}: {1}'.format(type(exc).__name__, exc)}
    return out


def main(args):
    inputs = load_inputs(args)
    if inputs['images'] is not None:
        (args.batch, args.height, args.width) = inputs['images'].shape
    (h_in, w_in) = (args.height, args.width)
    if args.transform_streams <= 0:
        args.transform_streams = codec.PRODUCT_TRANSFORM_STREAMS

    # two Python threads share the GIL (kernel launches; the codec's result worker): hand it over quickly
    sys.setswitchinterval(1e-4)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus {0} but WORLD_SIZE is {1}: run `python bench.py --gpus N` (it starts its own N ranks) '
                         'or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`.'.format(args.gpus, world))
    share_gpu = bool(os.environ.get('EAE_BENCH_SHARE_GPU'))      # test hook: several ranks on one GPU (gloo instead of RCCL)
    if args.dry_launch:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if os.environ.get('EAE_BENCH_FAIL_RANK') == str(rank):      # test hook: a rank that dies before the rendezvous
            raise SystemExit(3)
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)
        count = torch.tensor([1.], dtype=torch.float64)
        dist.all_reduce(count)
        if rank == 0:
            print(json.dumps({'dry_launch': True, 'n_gpus': world, 'ranks_seen': int(count.item()), 'local_rank': local_rank,
                              'coder_streams': args.coder_streams or auto_coder_streams(h_in, w_in), 'usable_cpus': usable_cpus()}))
        dist.destroy_process_group()
        return
    early = {}
    if (rank == 0 and world == 1 and not args.no_single_image and not (args.only_single_image_pipelined or args.only_library_user or args.only_dropin_surface)
            and args.batch != 1 and (h_in, w_in) == (512, 768) and torch.cuda.device_count() > 0):
        early = children_alone_on_the_gpu(args, h_in, w_in)      # BEFORE this process initialises the GPU (counting devices does not)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU fallback for the product path.')
    if share_gpu:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit('bench.py: rank {0} has no GPU (this node shows {1}); one process per GPU.'.format(local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_nccl:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:
            os.environ.setdefault('MASTER_PORT', str(free_port()))
        dist.init_process_group(backend='gloo' if share_gpu else 'nccl', rank=rank, world_size=world,
                                device_id=None if share_gpu else torch.device('cuda', local_rank))
    device = torch.device('cuda', local_rank)
    cores = usable_cpus()
    os.environ.setdefault('OMP_NUM_THREADS', str(cores))      # the oracle's OpenMP transforms (cpu_baseline only)
    ctx = Context(args, device, world, rank, cores)
    ctx.inputs = inputs
    coder_streams = args.coder_streams or auto_coder_streams(h_in, w_in)

    def model_at(width):
        """The model with its bin widths at `width` times their trained values (random-init: `width` itself)."""
        if inputs['variables'] is None:
            return synthetic_model(width)
        scaled = dict(inputs['variables'])
        scaled[var.BIN_WIDTHS_NAME] = ((width/args.bin_width)*inputs['variables'][var.BIN_WIDTHS_NAME]).astype(numpy.float32)
        return scaled

    variables = model_at(args.bin_width)
    if args.only_single_image_pipelined:
        print(json.dumps({'single_image_pipelined': single_image_pipelined_leg(ctx, variables, h_in, w_in)}))
        return
    if args.only_library_user:
        import autoencoder_based_image_compression_amd as package
        mode = codec.product_mode(h_in, w_in)
        one = run_pipeline(ctx, args.batch, args.steps, args.warmup, variables, h_in, w_in, coder_streams=mode['nb_in_flight'],
                           transform_streams=mode['nb_transform_streams'], use_graphs=mode['use_graphs'], min_seconds=args.min_seconds,
                           max_blocks=args.max_blocks)
        print(json.dumps({'library_user': {
            'value': round(args.batch*h_in*w_in*args.steps/one['elapsed']/1e6, 3), 'unit': 'Mpixels/s',
            'ms_per_step': round(one['elapsed']/args.steps*1e3, 4), 'steps': args.steps,
            'hw_queues': package.HW_QUEUES[0], 'hw_queues_set_by': package.HW_QUEUES[1], 'env_GPU_MAX_HW_QUEUES': os.environ.get('GPU_MAX_HW_QUEUES'),
            'mode': 'codec.BatchCodec(**codec.product_mode(h, w)) in a fresh process, alone on the GPU, whose environment ' + ('did not hold GPU_MAX_HW_QUEUES' if package.HW_QUEUES[1] == 'package' else 'held GPU_MAX_HW_QUEUES')}}))
        return
    if args.only_dropin_surface:
        # diagnostic: the statistics that feed the coder as run_pipeline derives them, then the leg alone
        encoder = pipeline.DeviceEncoder(variables, bool(args.learned_bin_widths), device)
        y0 = encoder(torch.from_numpy(numpy.ascontiguousarray(ctx.images(1000 + rank + args.seed_offset, args.batch, h_in, w_in))).to(device))
        (map_mean_host, probabilities, idx_map_exception) = ctx.statistics(y0, variables[var.BIN_WIDTHS_NAME])
        encoder.check()
        del y0, encoder
        print(json.dumps({'dropin_surface': dropin_surface_leg(ctx, variables, probabilities, map_mean_host, idx_map_exception, h_in, w_in)}))
        return
    # ---- the roofline leg first, while the process has no other streams: the same steps launched kernel by kernel on ONE
    # transform stream (the default stream), HIP events around every launch (on the stream it goes to), so that a duration is
    # that kernel's own next to nothing but the coder's side streams ------------------------------------------------------
    roof = run_pipeline(ctx, args.batch, min(args.steps, 30), min(args.warmup, 5), variables, h_in, w_in, coder=args.coder,
                        coder_streams=min(coder_streams, 3), transform_streams=1, use_graphs=False, min_seconds=0.3, max_blocks=5, record=True)
    # (the same leg with the transforms ALONE, rank 0 of a one-GPU run: what each kernel reaches of its roof without the coder's
    # long-lived waves beside it -- `roofline.transforms_alone`; `roofline` itself stays the path as it runs)
    roof_alone = None
    if world == 1 and not args.no_transforms_alone and not os.environ.get('EAE_BENCH_NO_CODER'):
        os.environ['EAE_BENCH_NO_CODER'] = '1'
        try:
            roof_alone = run_pipeline(ctx, args.batch, min(args.steps, 30), min(args.warmup, 5), variables, h_in, w_in, coder=args.coder,
                                      coder_streams=min(coder_streams, 3), transform_streams=1, use_graphs=False, min_seconds=0.3, max_blocks=5, record=True)
        finally:
            del os.environ['EAE_BENCH_NO_CODER']
    # ---- the headline: the product's default mode (three transform streams, the step replayed as hipGraphs) -------------
    run = run_pipeline(ctx, args.batch, args.steps, args.warmup, variables, h_in, w_in, coder=args.coder, coder_streams=coder_streams,
                       transform_streams=args.transform_streams, use_graphs=args.graphs, min_seconds=args.min_seconds,
                       max_blocks=args.max_blocks)
    (elapsed, stats, probabilities, map_mean_host) = (run['elapsed'], run['stats'], run['probabilities'], run['map_mean_host'])
    (host_coder, coder_threads, idx_map_exception) = (run['host_coder'], run['coder_threads'], run['idx_map_exception'])
    # ---- derived figures (outside the timed region) ------------------------------------------------------------------
    pixels_per_step = args.batch*h_in*w_in
    value = pixels_per_step*args.steps*world/elapsed/1e6
    nb_images_total = stats[3].item()
    (bpp, mean_psnr) = rate_and_psnr(stats, h_in, w_in)
    (per_kernel, per_launch_ms, flops) = launch_rooflines(roof['events'], pixels_per_step, args.fuse_latent, args.batch*128*(h_in//16)*(w_in//16))
    gemm = {k: v for (k, v) in per_launch_ms.items() if k in GEMM_LAUNCHES}
    gemm_ms = sum(sum(v) for v in gemm.values())
    gemm_launches = sum(len(v) for v in gemm.values())
    gemm_flop = sum(flops[name]*pixels_per_step*len(v) for (name, v) in gemm.items())
    achieved = gemm_flop/(gemm_ms*1e-3)/1e12 if gemm_ms > 0 else 0.
    (traffic, traffic_source) = (None, None)
    traffic_file = os.path.join(ROOT, 'profiles', 'traffic_conv_gemm.json')
    if os.path.isfile(traffic_file):
        with open(traffic_file) as f:
            traffic = json.load(f).get('hbm_bytes_per_launch')
        traffic_source = ('profiles/traffic_conv_gemm.json: rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, separate runs) of this '
                          'command at the default shape, committed with the profile summaries -- a constant of the build, not measured by this run')
    blocks = run['block_seconds']
    roof_ms = roof['elapsed']/min(args.steps, 30)*1e3
    transforms_alone = None
    if roof_alone is not None:
        (pk_a, ms_a, fl_a) = launch_rooflines(roof_alone['events'], pixels_per_step, args.fuse_latent, args.batch*128*(h_in//16)*(w_in//16))
        g_a = {k: v for (k, v) in ms_a.items() if k in GEMM_LAUNCHES}
        g_a_ms = sum(sum(v) for v in g_a.values())
        g_a_flop = sum(fl_a[k]*pixels_per_step*len(v) for (k, v) in g_a.items())
        transforms_alone = {'note': 'the same launch-by-launch leg with no coder launched (EAE_BENCH_NO_CODER=1): each kernel without the coder\'s waves beside it',
                            'achieved': round(g_a_flop/(g_a_ms*1e-3)/1e12, 3) if g_a_ms > 0 else 0.,
                            'frac': round(g_a_flop/(g_a_ms*1e-3)/1e12/PEAK_F32_MFMA_TFLOPS, 4) if g_a_ms > 0 else 0.,
                            'one_stream_ms_per_step': round(roof_alone['elapsed']/min(args.steps, 30)*1e3, 4),
                            'per_kernel': {k: {'avg_ms': v['avg_ms'], 'frac': v.get('frac')} for (k, v) in pk_a.items() if not k.startswith('coder')}}
        del roof_alone
    line = {
        'metric': 'Mpixels/s encode+decode ({0} {1}x{2} luma), bitstream bit-exact'.format(
            'Kodak' if (h_in, w_in) == (512, 768) else 'synthetic', w_in, h_in),
        'value': round(value, 3), 'unit': 'Mpixels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed/args.steps*1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': inputs['data'],
        'config': {'workload': '{0}_{1}x{2}_luma_batch{3}_per_gpu_bin_width_{4}_lossless_roundtrip'.format(
                       'kodak' if (h_in, w_in) == (512, 768) else 'synthetic', h_in, w_in, args.batch, args.bin_width),
                   'images_per_gpu_per_step': args.batch, 'height': h_in, 'width': w_in, 'bin_width_multiplier': args.bin_width,
                   'truncated_unary_length': int(probabilities.shape[1]), 'idx_map_exception': idx_map_exception,
                   'weights': ('random-init fixed-bin-width architecture (trained checkpoints absent from the reference)' if inputs['variables'] is None
                               else '{0} ({1}-bin-width architecture)'.format(args.checkpoint, 'learned' if args.learned_bin_widths else 'fixed')),
                   'parallelism': 'image shards, one process per GPU' if world > 1 else 'single GPU',
                   'mode': '{0} transform stream(s), {1} batches of coder work in flight, {2}'.format(
                       args.transform_streams, coder_streams, 'three hipGraph launches per step' if args.graphs else 'launched kernel by kernel'),
                   'coder': 'device, 64 maps per wavefront, encode + decode + compare' if not host_coder else
                            'host C-ABI coder, {} threads, after one device -> host copy of the symbols'.format(coder_threads)},
        'timing': {'blocks': len(blocks), 'steps_per_block': args.steps, 'reported': 'median block',
                   'block_ms': [round(b*1e3, 3) for b in blocks], 'min_block_ms': round(min(blocks)*1e3, 3),
                   'max_block_ms': round(max(blocks)*1e3, 3), 'spread_pct': round((max(blocks) - min(blocks))/elapsed*100., 2),
                   'timed_seconds': round(sum(blocks), 4)},
        'images_per_s': round(args.batch*args.steps*world/elapsed, 2),
        # process CPU time (all threads) per step during the median block, one entry per rank: 8 ranks share the node's CPU quota
        'host_cpu_ms_per_step': run['host_cpu_ms_per_step'], 'usable_cpus': cores,
        # summed over ranks by the path's one all-reduce, over all timed blocks: exact integers (tests compare them across world sizes)
        'totals': {'bits': int(stats[0].item()), 'sse': int(stats[1].item()), 'dead_maps': int(stats[2].item()), 'images': int(nb_images_total)},
        'rate_bpp': round(bpp, 5), 'psnr_db_pooled': round(mean_psnr, 4), 'dead_maps_per_image': round(stats[2].item()/nb_images_total, 3),
        'one_stream_leg': {'ms_per_step': round(roof_ms, 4), 'value': round(pixels_per_step*world/roof_ms/1e3, 3), 'unit': 'Mpixels/s',
                           'flags': '--transform-streams 1 --no-graphs', 'host_cpu_ms_per_step': roof['host_cpu_ms_per_step'],
                           'note': 'source of `roofline`: every launch bracketed by HIP events on its own stream'},
        'roofline': {'bound': 'mfma', 'kernel': 'conv_gemm_split_kernel (conv2+GDN2, conv3, tconv1+IGDN5, tconv2+IGDN6)',
                     'achieved': round(achieved, 3), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(achieved/PEAK_F32_MFMA_TFLOPS, 4), 'traffic': traffic, 'traffic_source': traffic_source,
                     'avg_launch_ms': round(gemm_ms/max(gemm_launches, 1), 4),
                     'per_launch_ms': {k: round(sum(v)/len(v), 4) for (k, v) in gemm.items()},
                     'per_launch_frac': {k: round(flops[k]*pixels_per_step/(sum(v)/len(v)*1e-3)/1e12/PEAK_F32_MFMA_TFLOPS, 4)
                                         for (k, v) in gemm.items()},
                     'algorithmic_flop_per_launch': {k: flops[k]*pixels_per_step for k in GEMM_LAUNCHES},
                     'per_kernel': per_kernel, 'peak_hbm_tbytes_per_s': PEAK_HBM_TBS, 'transforms_alone': transforms_alone},
    }
    line['exchange'] = args.exchange
    # a scaling curve needs one GPU per rank: ranks sharing a GPU (EAE_BENCH_SHARE_GPU) or a single rank measure none
    line['scaling_measured'] = bool(world > 1 and not share_gpu)
    partition = dev.partition_info()
    line['device'] = dict(dev.device_info(), xcds=partition['xcds'], whole_device=partition['whole_device'])
    if ctx.grouped:
        import torch.distributed as dist
        # the collectives of this run (barriers around every timed block, the MAX of the block times, the one statistics
        # all-reduce) went through this backend; with --force-nccl also as a single rank
        line['process_group'] = {'backend': dist.get_backend(), 'world_size': world, 'collectives': ctx.collectives,
                                 'forced_single_rank': bool(world == 1)}
    del run, roof
    side = rank == 0 and world == 1 and not args.no_single_image
    if side and args.batch != 1 and (h_in, w_in) == (512, 768):
        # BASELINE.json configs[1] is ONE Kodak image: the same path with one image per step (launch-bound, not the headline)
        line['single_image'] = {'ms_per_image': None, 'mpixels_per_s': None, 'steps': 1000, 'warmup': 30, 'host_cpu_ms_per_image': None,      # filled in below
                                'latency_ms': None,
                                'latency_note': 'one image at a time, each waited for before the next is submitted (submit -> result '
                                                'on the host, blocks of 100 images, median block): what BASELINE.json configs[1] takes end to end; '
                                                'measured, like the pipelined figure, in a process of its own that has the GPU to itself (inside this '
                                                'process, behind its other legs and their fifteen streams, the same loop measures 0.06 ms more)',
                                'note': 'one 512x768 image per step; a step is a chain of short latency-bound kernels, so steps '
                                        'are pipelined: codec.BatchCodec(one_stream_steps=True), a stream per step in flight, one '
                                        'hipGraph launch per step (`steps_in_flight`); `latency_ms`: the default schedule (coder beside the synthesis transform)'}
    if side and (h_in, w_in, args.batch) == (512, 768, 24):
        # the other shapes BASELINE.json names, same default flags (what `python bench.py --height H --width W --batch B` prints)
        line['other_shapes'] = []
        for (b2, h2, w2, steps2, what) in ((64, 256, 256, 100, 'one rank of configs[3] (512 images of 256x256 over 8 GPUs)'),
                                           (2, 2048, 2048, 50, 'configs[4]: 2048x2048, untiled (fits HBM), two images per step')):
            leg = run_pipeline(ctx, b2, steps2, 8, variables, h2, w2, coder_streams=auto_coder_streams(h2, w2),
                               transform_streams=args.transform_streams, use_graphs=args.graphs, min_seconds=0.4, max_blocks=5)
            entry = {'workload': '{0}x{1}x{2}'.format(b2, h2, w2), 'what': what,
                     'value': round(steps2*b2*h2*w2/leg['elapsed']/1e6, 3), 'unit': 'Mpixels/s',
                     'ms_per_step': round(leg['elapsed']/steps2*1e3, 4), 'steps': steps2,
                     'coder_streams': auto_coder_streams(h2, w2), 'rate_bpp': round(rate_and_psnr(leg['stats'], h2, w2)[0], 5)}
            # the shape's own launch-by-launch leg (as for `roofline`): what its kernels reach of their roofs
            roof2 = run_pipeline(ctx, b2, 30, 5, variables, h2, w2, coder_streams=min(auto_coder_streams(h2, w2), 3), transform_streams=1,
                                 use_graphs=False, min_seconds=0.3, max_blocks=5, record=True)      # as the headline's own roofline leg
            (pk2, ms2, fl2) = launch_rooflines(roof2['events'], b2*h2*w2, args.fuse_latent, b2*128*(h2//16)*(w2//16))
            g2 = {k: v for (k, v) in ms2.items() if k in GEMM_LAUNCHES}
            g2_ms = sum(sum(v) for v in g2.values())
            g2_flop = sum(fl2[k]*b2*h2*w2*len(v) for (k, v) in g2.items())
            entry['roofline'] = {'bound': 'mfma', 'kernel': 'conv GEMM launches', 'unit': 'TFLOP/s', 'peak': PEAK_F32_MFMA_TFLOPS,
                                 'achieved': round(g2_flop/(g2_ms*1e-3)/1e12, 3) if g2_ms > 0 else 0.,
                                 'frac': round(g2_flop/(g2_ms*1e-3)/1e12/PEAK_F32_MFMA_TFLOPS, 4) if g2_ms > 0 else 0.,
                                 'one_stream_ms_per_step': round(roof2['elapsed']/30*1e3, 4),
                                 'per_kernel': {k: {'avg_ms': v['avg_ms'], 'frac': v.get('frac'), 'bound': v['bound']} for (k, v) in pk2.items()}}
            del roof2
            line['other_shapes'].append(entry)
    golden = os.path.join(ROOT, 'tests', 'golden', 'coder_golden.npz')
    if side and args.coder == 'device' and inputs['statistics'] is None and os.path.isfile(golden):
        # the coder exercised on the AUTHORS' statistics (the tables of lossless/results/1_10000/training_index_10/ of the reference,
        # held as data in tests/golden/): their map means, exception map and binary probabilities over this run's latents. The tables
        # were not made from these latents, so the rate is what a mismatched table costs; every map still round-trips.
        with numpy.load(golden) as g:
            authors = (g['real_map_mean'].astype(numpy.float32), g['real_probabilities_1'].copy(), int(g['real_idx_map_exception']))
        leg = run_pipeline(ctx, args.batch, 30, 5, variables, h_in, w_in, coder_streams=coder_streams, transform_streams=args.transform_streams,
                           use_graphs=args.graphs, min_seconds=0.4, max_blocks=5, statistics=authors)
        line['authors_statistics'] = {'value': round(30*pixels_per_step/leg['elapsed']/1e6, 3), 'unit': 'Mpixels/s', 'ms_per_step': round(leg['elapsed']/30*1e3, 4),
                                      'steps': 30, 'rate_bpp': round(rate_and_psnr(leg['stats'], h_in, w_in)[0], 5), 'idx_map_exception': authors[2],
                                      'tables': 'map_mean.npy, idx_map_exception.pkl, binary_probabilities_1.npy of the reference model 1_10000 / training_index_10 '
                                                '(tests/golden/coder_golden.npz); --stats-dir PATH runs the headline itself on a directory of such files'}
    if side and args.coder == 'device':
        # the feed / fetch of the reference's sess.run (uint8 images from pinned host memory in, uint8 reconstructions out)
        feed = run_pipeline(ctx, args.batch, 60, 10, variables, h_in, w_in, coder_streams=coder_streams, transform_streams=args.transform_streams,
                            use_graphs=args.graphs, min_seconds=0.4, max_blocks=5, pcie=True)
        line['pcie_inclusive'] = {'value': round(60*pixels_per_step/feed['elapsed']/1e6, 3), 'unit': 'Mpixels/s',
                                  'ms_per_step': round(feed['elapsed']/60*1e3, 4), 'steps': 60, 'warmup': 10,
                                  'bytes_per_step': {'host_to_device': pixels_per_step, 'device_to_host': pixels_per_step},
                                  'note': 'codec.BatchCodec(fetch_reconstruction=True).submit(pinned host batch): the uint8 batch is '
                                          'copied in on the codec\'s feed stream before every step, the uint8 reconstruction copied '
                                          'back to pinned host memory by its result worker; never the headline'}
        # north_star's original shape (one device -> host copy of the symbols, the host C-ABI coder on the CPUs the quota allows)
        host = run_pipeline(ctx, args.batch, 30, 5, variables, h_in, w_in, coder='host', coder_streams=3, min_seconds=0.5, max_blocks=5)
        line['host_coder'] = {'value': round(30*pixels_per_step/host['elapsed']/1e6, 3), 'unit': 'Mpixels/s',
                              'ms_per_step': round(host['elapsed']/30*1e3, 4), 'steps': 30, 'warmup': 5, 'blocks': len(host['block_seconds']),
                              'threads': host['coder_threads'], 'usable_cpus': cores, 'rate_bpp': round(rate_and_psnr(host['stats'], h_in, w_in)[0], 5),
                              'flags': '--coder host'}
        # a latent with the entropy of a trained model: the same path at smaller bin widths. How long the coder chain of a batch
        # is on its own, what a step costs with and without it, hence whether the transforms still hide it.
        line['realistic_entropy'] = []
        os.environ['EAE_BENCH_NO_CODER'] = '1'
        bare = run_pipeline(ctx, args.batch, 30, 5, variables, h_in, w_in, coder_streams=coder_streams, transform_streams=args.transform_streams,
                            use_graphs=args.graphs, min_seconds=0.3, max_blocks=5)
        del os.environ['EAE_BENCH_NO_CODER']
        ms_bare = bare['elapsed']/30*1e3
        for width in (args.bin_width,) + tuple(wd for wd in REALISTIC_BIN_WIDTHS if wd < args.bin_width):
            v_w = model_at(width)
            given = width == args.bin_width          # a --stats-dir table belongs to one multiplier
            leg = run_pipeline(ctx, args.batch, 30, 5, v_w, h_in, w_in, coder_streams=coder_streams, transform_streams=args.transform_streams,
                               use_graphs=args.graphs, min_seconds=0.5, max_blocks=5, given_statistics=given)
            # the coder's span in the pipeline needs events on its stream: the launch-by-launch path, one transform stream
            span = run_pipeline(ctx, args.batch, 20, 5, v_w, h_in, w_in, coder_streams=coder_streams, min_seconds=0., max_blocks=1, coder_events=True,
                                given_statistics=given)
            (enc_ms, dec_ms) = coder_alone_ms(ctx, args.batch, v_w, h_in, w_in)
            (leg_bpp, leg_psnr) = rate_and_psnr(leg['stats'], h_in, w_in)
            ms_step = leg['elapsed']/30*1e3
            in_pipe = sorted(span['coder_ms'])[len(span['coder_ms'])//2]
            line['realistic_entropy'].append({
                'bin_width': width, 'rate_bpp': round(leg_bpp, 4), 'psnr_db_pooled': round(leg_psnr, 3),
                'value': round(30*pixels_per_step/leg['elapsed']/1e6, 3), 'ms_per_step': round(ms_step, 4),
                'ms_per_step_without_coder': round(ms_bare, 4), 'step_over_no_coder_step': round(ms_step/ms_bare, 4),
                'coder_alone_ms_per_batch': {'binarise_encode': round(enc_ms, 4), 'decode_compare': round(dec_ms, 4)},
                'coder_in_pipeline_ms_per_batch': round(in_pipe, 4), 'coder_streams': coder_streams,
                # the span is measured in the launch-by-launch schedule on one transform stream (it needs events on the coder's stream),
                # `ms_step` in the product mode: an indication, not an identity -- the step ratio above is the measurement
                'coder_span_measured_in': 'launch-by-launch schedule, one transform stream',
                'coder_on_critical_path': bool(in_pipe > coder_streams*ms_step*0.95)})
    if rank == 0 and world == 1 and not args.no_dropin_surface and (h_in, w_in) == (512, 768):
        try:
            line['dropin_surface'] = dropin_surface_leg(ctx, variables, probabilities, map_mean_host, idx_map_exception, h_in, w_in)
        except Exception as exc:      # a side figure must never cost the run its headline
            line['dropin_surface'] = {'error': '{0}: {1}'.format(type(exc).__name__, exc)}
    if side and 'single_image' in line:
        # the pipelined figure of one image per step: measured by `children_alone_on_the_gpu` before this process touched the GPU
        line['single_image'].update(early.get('single_image_pipelined', {'pipelined_error': 'not run'}))
    if 'library_user' in early:
        line['library_user'] = early['library_user']
        if 'value' in line['library_user']:
            line['library_user']['over_headline'] = round(line['library_user']['value']/line['value'], 4)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            # the one leg of this file that runs checker code (oracle/): the CPU baseline and, with the same CPU transforms, what
            # the transforms' summation order does to symbols, bits and PSNR end to end
            (line['cpu_baseline'], line['order_sensitivity']) = cpu_baseline_leg(variables, probabilities, map_mean_host, cores, h_in, w_in, device)
            surface = line.get('dropin_surface', {})
            for key in ('code_lossless', 'approx', 'fix_gamma_batched'):
                if key in surface and line['cpu_baseline'].get('value'):
                    surface[key]['over_cpu_baseline'] = round(surface[key]['value']/line['cpu_baseline']['value'], 2)
            if 'code_lossless' in surface:
                surface['north_star_target'] = {'over_cpu_baseline': 50., 'met': bool(surface['code_lossless'].get('over_cpu_baseline', 0.) >= 50.)}
        print(json.dumps(line))
        sys.stdout.flush()
    if ctx.grouped:
        import torch.distributed as dist
        dist.destroy_process_group()


def single_image_pipelined_leg(ctx, variables, h, w, steps=1000):
    """One image per step, pipelined: fourteen to twenty steps in flight, each ONE graph launch on a stream of its own with the coder behind the
    synthesis transform instead of beside it (codec.BatchCodec(one_stream_steps=True)): no hop between streams, a third of the launching
    thread's work per step."""
    # first, with nothing else in the process: ONE image at a time, each waited for before the next is submitted (the default schedule,
    # coder beside the synthesis transform) -- `Ticket.result()` finds the step unclaimed and waits for the device itself (codec._Job)
    alone = run_pipeline(ctx, 1, 100, 10, variables, h, w, coder_streams=1, transform_streams=1, use_graphs=True, serial=True, min_seconds=1.5, max_blocks=21)
    streams = max(2, min(20, int(os.environ.get('GPU_MAX_HW_QUEUES', '4')) - 2))      # 14 / 20 streams with 16 / 24 queues: 0.250 / 0.228 ms per image; 28 with 32: 0.53
    one = run_pipeline(ctx, 1, steps, 30, variables, h, w, coder_streams=streams, transform_streams=streams, use_graphs=True, one_stream_steps=True)
    return {'ms_per_image': round(one['elapsed']/steps*1e3, 4), 'mpixels_per_s': round(steps*h*w/one['elapsed']/1e6, 2), 'steps': steps, 'warmup': 30,
            'host_cpu_ms_per_image': one['host_cpu_ms_per_step'][0], 'steps_in_flight': streams, 'gpu_max_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES'),
            'latency_ms': round(alone['elapsed']/100*1e3, 4), 'latency_host_cpu_ms_per_image': alone['host_cpu_ms_per_step'][0],
            'latency_blocks_ms_per_image': [round(b/100*1e3, 4) for b in alone['block_seconds']],
            'pipelined_in': 'a process of its own, alone on the GPU: started before the bench process initialised its GPU context (GPU_MAX_HW_QUEUES=24 python bench.py --only-single-image-pipelined)'}


def dropin_surface_leg(ctx, variables, probabilities, map_mean, idx_map_exception, h, w, nb_images=24, batch_size=4, repeats=5):
    """What a user of the reference's OWN call surface gets: the mirror of `fix_gamma` (reconstructing_eae_kodak.py:31-243) run as
    the reference's script runs it -- `nb_images` uint8 numpy images in, `batch_size = 4` (:624), one multiplier, numpy in / numpy
    out at every call, per-image `rescale_compress_lossless_maps` / `rate_3d` / `psnr_2d`, no PNG dumps -- with the modules imported
    under the reference's names from `dropin/` (the import-shadowing route of INTEGRATION.md section 2). Timed region = BASELINE.md
    section 3 item 4: the body of `fix_gamma` for one rate point without graph construction / checkpoint restore (the two
    `initialization` calls and the constructors are timed apart and subtracted). `code_lossless` on and off; `fix_gamma_batched`
    (the same arrays through codec.BatchCodec) beside it. Median of `repeats` calls after one warm-up call."""
    import importlib
    import pickle
    import shutil
    import tempfile
    from autoencoder_based_image_compression_amd.kodak import reconstructing_eae_kodak as mirror
    dropin = os.path.join(ROOT, 'autoencoder_based_image_compression_amd', 'dropin')
    sys.path.insert(0, dropin)
    try:
        by_reference_name = {name: importlib.import_module(name) for name in (
            'tensorflow', 'eae.batching', 'eae.graph.EntropyAutoencoder', 'eae.graph.IsolatedDecoder', 'lossless.compression', 'tools.tools')}
    finally:
        sys.path.remove(dropin)
    # the names the reference's script binds (reconstructing_eae_kodak.py:21-29) are the very objects the mirror calls
    same = (by_reference_name['eae.batching'].encode_mini_batches is mirror.batching.encode_mini_batches and
            by_reference_name['eae.batching'].decode_mini_batches is mirror.batching.decode_mini_batches and
            by_reference_name['tools.tools'].quantize_per_map is mirror.tls.quantize_per_map and
            by_reference_name['tools.tools'].rate_3d is mirror.tls.rate_3d and
            by_reference_name['tools.tools'].psnr_2d is mirror.tls.psnr_2d and
            by_reference_name['lossless.compression'].rescale_compress_lossless_maps is mirror.compression.rescale_compress_lossless_maps and
            by_reference_name['eae.graph.EntropyAutoencoder'].EntropyAutoencoder is mirror.EntropyAutoencoder and
            by_reference_name['eae.graph.IsolatedDecoder'].IsolatedDecoder is mirror.IsolatedDecoder and
            by_reference_name['tensorflow'].Session is mirror.tf.Session)
    if not same:
        raise RuntimeError('dropin/ does not re-export the objects the mirror harness calls')
    root = tempfile.mkdtemp(prefix='eae_dropin_surface_')
    try:
        learned = var.ENCODER_NAMES_FIXED_BW[0] not in variables
        bin_width_init = float(variables[var.BIN_WIDTHS_NAME][0])      # only names the directories (reconstructing_eae_kodak.py:88-93)
        suffix = '{0}{1}_10000'.format('learning_bw_' if learned else '', tls.float_to_str(bin_width_init))
        model_dir = os.path.join(root, 'eae/results', suffix)
        stats_dir = os.path.join(root, 'lossless/results', suffix, 'training_index_10')
        os.makedirs(model_dir)
        os.makedirs(stats_dir)
        var.save_variables(os.path.join(model_dir, 'model_10.npz'), variables)
        with open(os.path.join(model_dir, 'nb_itvs_per_side_10.pkl'), 'wb') as f:
            pickle.dump(91, f, protocol=2)
        numpy.save(os.path.join(stats_dir, 'map_mean.npy'), map_mean)
        with open(os.path.join(stats_dir, 'idx_map_exception.pkl'), 'wb') as f:
            pickle.dump(idx_map_exception, f, protocol=2)
        numpy.save(os.path.join(stats_dir, 'binary_probabilities_1.npy'), probabilities)
        images = numpy.ascontiguousarray(ctx.images(1000, nb_images, h, w))
        multipliers = numpy.array([1.], dtype=numpy.float32)

        # wall time spent inside named calls of the surface (cheap wrappers; the two `initialization`s are always wrapped: their
        # time is what the timed region leaves out)
        spent = {}

        def timed(name, fn):
            def wrapper(*a, **k):
                t0 = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    spent[name] = spent.get(name, 0.) + time.perf_counter() - t0
            return wrapper

        patches = []

        def patch(owner, attribute, name):
            original = getattr(owner, attribute)
            patches.append((owner, attribute, original))
            setattr(owner, attribute, timed(name, original))

        def unpatch():
            while patches:
                (owner, attribute, original) = patches.pop()
                setattr(owner, attribute, original)

        def call(is_lossless, batched=False):
            spent.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if batched:
                out = mirror.fix_gamma_batched(images, bin_width_init, multipliers, 10, 10000., batch_size, learned, root=root)
            else:
                out = mirror.fix_gamma(images, bin_width_init, multipliers, 10, 10000., batch_size, learned, is_lossless, root=root)
            total = time.perf_counter() - t0
            return (out, total, dict(spent))

        def measure(is_lossless, batched=False, breakdown=False):
            patch(mirror.EntropyAutoencoder, 'initialization', 'init')
            patch(mirror.IsolatedDecoder, 'initialization', 'init')
            if breakdown:
                patch(mirror.batching, 'encode_mini_batches', 'encode_mini_batches')
                patch(mirror.batching, 'decode_mini_batches', 'decode_mini_batches')
                patch(mirror.tls, 'quantize_per_map', 'quantize_per_map')
                patch(mirror.tls, 'count_nb_deads', 'count_nb_deads')
                patch(mirror.tls, 'rate_3d', 'rate_3d')
                patch(mirror.tls, 'psnr_2d', 'psnr_2d')
                patch(mirror.compression, 'rescale_compress_lossless_maps', 'rescale_compress_lossless_maps')
            try:
                call(is_lossless, batched)                                        # warm-up: module loads, allocator, file cache
                runs = [call(is_lossless, batched) for _ in range(1 if breakdown else repeats)]
            finally:
                unpatch()
            runs.sort(key=lambda r: r[1] - r[2].get('init', 0.))
            return runs[(len(runs) - 1)//2]

        pixels = nb_images*h*w
        out = {'workload': '{0} x {1}x{2} uint8 numpy, batch_size {3}, one multiplier (1.0), no PNG dumps'.format(nb_images, h, w, batch_size),
               'through': 'dropin/ (tensorflow, eae.batching, eae.graph.*, lossless.compression, tools.tools): the objects the mirror of '
                          'fix_gamma calls, checked by identity',
               'timed_region': 'the call of fix_gamma minus its two `initialization` calls (BASELINE.md section 3 item 4); median of {} calls'.format(repeats),
               'unit': 'Mpixels/s'}
        reference = {}
        for (key, is_lossless) in (('code_lossless', True), ('approx', False)):
            ((rate, psnr), total, sp) = measure(is_lossless)
            region = total - sp.get('init', 0.)
            (_, total_b, sp_b) = measure(is_lossless, breakdown=True)
            named = {k: round(v*1e3, 3) for (k, v) in sorted(sp_b.items()) if k != 'init'}
            named['harness_own_numpy_and_constructors'] = round((total_b - sum(sp_b.values()))*1e3, 3)
            out[key] = {'value': round(pixels/region/1e6, 3), 'images_per_s': round(nb_images/region, 2), 'ms_per_image': round(region/nb_images*1e3, 4),
                        'ms_timed_region': round(region*1e3, 3), 'ms_whole_call': round(total*1e3, 3), 'ms_initialization': round(sp.get('init', 0.)*1e3, 3),
                        'ms_per_call_of_the_surface': named, 'mean_rate_bpp': round(float(rate.mean()), 5), 'mean_psnr_db': round(float(psnr.mean()), 4)}
            reference[key] = (rate, psnr)
        ((rate_b, psnr_b), total, sp) = measure(True, batched=True)
        out['fix_gamma_batched'] = {'value': round(pixels/total/1e6, 3), 'images_per_s': round(nb_images/total, 2), 'ms_whole_call': round(total*1e3, 3),
                                    'note': 'same files, same returned arrays through codec.BatchCodec (whole call incl. checkpoint restore, codec '
                                            'construction and graph capture: it has no separate initialization)',
                                    'arrays_equal_fix_gamma': bool(numpy.array_equal(rate_b, reference['code_lossless'][0]) and
                                                                   numpy.array_equal(psnr_b, reference['code_lossless'][1]))}
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def cpu_baseline_leg(variables, probabilities, map_mean, cores, h, w, device):
    """bench.py's checker leg (rank 0, one GPU): `cpu_baseline` (the path timed on the host cores) and `order_sensitivity` (the HIP
    path held against the same graph in two other arithmetics on the host). Nothing outside this leg touches oracle/."""
    baseline = cpu_baseline(variables, probabilities, map_mean, cores, h, w)
    try:
        sensitivity = order_sensitivity(device, cores)
    except Exception as exc:      # a checker figure must never cost the run its headline
        sensitivity = {'error': '{0}: {1}'.format(type(exc).__name__, exc)}
    return (baseline, sensitivity)


def order_sensitivity(device, cores, configs=None):
    """north_star's float clause (symbols bit-exact, reconstruction within 1e-4 dB PSNR) stated end to end with TensorFlow absent: the
    HIP path's symbols, coded bits and PSNR on BASELINE.json configs[1] (one Kodak-sized image, bin width 1.0) and on a bounded
    share of configs[2] (Kodak-sized images at bin widths 0.5 / 1.0 / 2.0) against the same graph (i) in float64 and (ii) in float32
    in oneDNN's summation order (oracle/order_sensitivity.py: checker code, used only here and in tests/test_gpu_order_sensitivity.py).
    Coded bits: the product's host coder, the same probability tables for every arithmetic."""
    from oracle import order_sensitivity as checker
    from autoencoder_based_image_compression_amd.kodak.lossless import compression
    variables = synthetic_model(1.)
    encoder = pipeline.DeviceEncoder(variables, False, device)
    decoder = pipeline.DeviceDecoder(variables, False, device)
    configs = configs or (('configs[1]', 1, (1.0,)), ('configs[2] (4 of its 24 images)', 4, (0.5, 1.0, 2.0)))
    out = {'tolerance_psnr_db': 1e-4, 'tolerance_distance_from_a_rounding_boundary': 1e-4,
           'note': 'HIP path against the same graph in float64 and in float32 in oneDNN order (TensorFlow absent: its order cannot be run)'}
    t0 = time.perf_counter()
    for (label, nb_images, multipliers) in configs:
        images = synthetic_images(2000, nb_images, 512, 768)
        x = torch.from_numpy(images).to(device)
        y = encoder(x)
        map_mean = dev.map_means(y)
        (y_host, mean_host) = (y.cpu().numpy(), map_mean.cpu().numpy())
        (product, widths, tables) = ({}, {}, {})
        for m in multipliers:
            name = 'bin_width_{0}'.format(m)
            bw = numpy.full(128, m, dtype=numpy.float32)
            q = dev.quantize_maps(y, torch.from_numpy(bw).to(device), map_mean, want_shifted=True, want_symbols=True)
            (_, rec_u8, _) = decoder(q['shifted'], reference_uint8=x)
            planar = q['symbols'].cpu().numpy()                                             # [N][C][h w]
            product[name] = {'symbols': numpy.ascontiguousarray(planar.transpose(0, 2, 1)).reshape(y_host.shape),
                             'reconstruction': rec_u8.cpu().numpy()}
            widths[name] = bw
            tables[name] = lossless_stats.compute_binary_probabilities(y_host, bw, mean_host, TRUNCATED_UNARY_LENGTH)
        rows = {}
        for name in product:
            def count_bits(symbols_nhwc, table=tables[name]):
                n = symbols_nhwc.shape[0]
                planar = numpy.ascontiguousarray(symbols_nhwc.reshape(n, -1, 128).transpose(0, 2, 1).astype(numpy.int16))
                (_, nb_bits) = compression.code_planar_symbols(planar, table, IDX_MAP_EXCEPTION, nb_threads=max(1, cores - 1), verify_only=True)
                return nb_bits.astype(numpy.int64).sum(axis=1)
            one = checker.compare(images, {name: product[name]}, variables, False, {name: widths[name]}, mean_host, count_bits, threads=cores)
            for (path, per_width) in one['paths'].items():
                rows.setdefault(path, {})[name] = per_width[name]
        out[label] = {'images': nb_images, 'vs': rows}
    worst = [r for cfg in out.values() if isinstance(cfg, dict) and 'vs' in cfg for per in cfg['vs'].values() for r in per.values()]
    out['summary'] = {'symbols_compared': int(sum(r['symbols'] for r in worst)), 'symbols_differing': int(sum(r['symbols_differing'] for r in worst)),
                      'largest_distance_from_a_rounding_boundary': max(r['largest_distance_from_a_rounding_boundary'] for r in worst),
                      'delta_bits_per_image_max': max(r['delta_bits_per_image_max'] for r in worst),
                      'delta_psnr_db_per_image_max': max(r['delta_psnr_db_per_image_max'] for r in worst),
                      'within_tolerance': bool(max(r['delta_psnr_db_per_image_max'] for r in worst) <= 1e-4 and
                                               max(r['largest_distance_from_a_rounding_boundary'] for r in worst) < 1e-4),
                      'seconds': round(time.perf_counter() - t0, 1)}
    return out


def cpu_baseline(variables, probabilities, map_mean, cores, h, w):
    """The same path on the host cores, on a BOUNDED sample (checker code, timed only here, never shipped):
    transforms twice -- on torch-CPU (oneDNN convolutions, `cores` threads: the stand-in for the TensorFlow-CPU kernels behind the
    reference's sess.run, oracle/transforms_torch.py) and by the plain-C oracle (oracle/transforms_oracle.c, OpenMP, the bit-exact
    checker); coder = the reference's own C++ coder compiled into oracle/_ref (single thread, as the reference runs it),
    falling back to the oracle's C restatement when the reference build is absent; numpy quantiser / PSNR.
    `value` uses the FASTER transform leg. One image calibrates, then as many images as fit in about 20 s of CPU work."""
    from oracle import coder as oracle_coder
    from oracle import transforms as oracle_transforms
    from oracle import transforms_torch
    import ctypes
    try:        # libgomp was initialised when torch was imported: set the team size for this thread explicitly
        ctypes.CDLL('libgomp.so.1').omp_set_num_threads(int(cores))
    except OSError:
        pass
    bw = variables[var.BIN_WIDTHS_NAME]
    kind_coder = 'ref' if oracle_coder.available('ref') else 'oracle'
    lib = oracle_coder.CoderLib(kind_coder)
    cpu = transforms_torch.CpuTransforms(variables, False, threads=cores)
    coder_key = 'coder_{}_single_thread'.format('reference_cpp' if kind_coder == 'ref' else 'oracle_c')

    def run(n_img, with_oracle):
        x = synthetic_images(999, n_img, h, w)
        xf = x.astype(numpy.float32)[..., None]
        t = {}
        t0 = time.perf_counter()
        y = cpu.encoder(xf)
        t['encoder_torch_cpu'] = time.perf_counter() - t0
        if with_oracle:
            t0 = time.perf_counter()
            oracle_transforms.encoder(xf, variables, False)
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
        t[coder_key] = time.perf_counter() - t0
        t0 = time.perf_counter()
        rec = cpu.decoder(cq + map_mean)[..., 0]
        rec_u8 = numpy.round(rec.clip(min=16., max=235.)).astype(numpy.uint8)
        mse = numpy.mean((x.astype(numpy.float64) - rec_u8.astype(numpy.float64))**2)
        t['decoder_torch_cpu_plus_psnr'] = time.perf_counter() - t0
        if with_oracle:
            t0 = time.perf_counter()
            oracle_transforms.decoder(cq + map_mean, variables, False)
            t['decoder_oracle_c_openmp'] = time.perf_counter() - t0
        return (t, bits, float(mse))

    run(1, False)                 # first touch: oneDNN primitive creation, thread pool
    (t1, _, _) = run(1, True)
    per_image_torch = t1['encoder_torch_cpu'] + t1['quantiser_numpy'] + t1[coder_key] + t1['decoder_torch_cpu_plus_psnr']
    per_image_oracle = t1['encoder_oracle_c_openmp'] + t1['quantiser_numpy'] + t1[coder_key] + t1['decoder_oracle_c_openmp']
    with_oracle = per_image_oracle < per_image_torch          # the plain-C transforms are timed on the full sample only if they are the faster leg
    # FIVE repetitions of about 4 s of CPU work each (the median of >= 3 runs), the spread printed. A
    # repetition is `passes` passes over a sample of at most 24 images (one mini-batch of the headline), its times summed.
    per_image = max(min(per_image_torch, per_image_oracle), 1e-3)
    n_img = int(max(2, min(24, round(4./per_image))))
    passes = int(max(1, round(4./(n_img*per_image))))
    reps = []
    for _ in range(5):
        t = {}
        for _ in range(passes):
            (t_pass, bits, mse) = run(n_img, with_oracle)
            for (k, val) in t_pass.items():
                t[k] = t.get(k, 0.) + val
        torch_total = t['encoder_torch_cpu'] + t['quantiser_numpy'] + t[coder_key] + t['decoder_torch_cpu_plus_psnr']
        oracle_total = (t['encoder_oracle_c_openmp'] + t['quantiser_numpy'] + t[coder_key] + t['decoder_oracle_c_openmp']) if with_oracle else float('inf')
        reps.append((passes*n_img*h*w/min(torch_total, oracle_total)/1e6, t, torch_total, oracle_total))
    reps.sort(key=lambda r: r[0])
    (value, t, torch_total, oracle_total) = reps[len(reps)//2]
    return {'value': round(value, 4), 'unit': 'Mpixels/s', 'cores': cores, 'kind': 'port',
            'repetitions': [round(r[0], 4) for r in reps], 'spread': round((reps[-1][0] - reps[0][0])/value, 4),
            'transform_leg_used': 'torch_cpu' if torch_total <= oracle_total else 'oracle_c_openmp',
            'value_with_torch_cpu_transforms': round(passes*n_img*h*w/torch_total/1e6, 4),
            'value_with_oracle_c_transforms_one_image': round(h*w/per_image_oracle/1e6, 4),
            'threads': {'torch_intraop': cpu.threads, 'openmp': cores, 'coder': 1},
            'sample': ('median of 5 repetitions of {5} x {0} synthetic {1}x{2} images each, encode+quantise+code(enc+dec)+decode+PSNR, {3:.1f} s of CPU '
                       'work per repetition (`spread` = (max - min)/median of the five); transforms on torch-CPU/oneDNN ({4} threads: the stand-in for the reference\'s TensorFlow-CPU '
                       'kernels; the plain-C oracle (OpenMP, {4} threads) timed on one image beside it: `value_with_oracle_c_transforms_one_image`); '
                       'coder single-threaded like the reference').format(n_img, h, w, sum(t.values()), cores, passes),
            'seconds': {k: round(val, 3) for (k, val) in t.items()}, 'bits': int(bits), 'mse': round(mse, 4)}


if __name__ == '__main__':
    main(_ARGS)
