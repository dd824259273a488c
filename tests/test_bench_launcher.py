"""`python bench.py --gpus N` must run N ranks (missing #1: the flag was parsed and ignored, so the
driver's scaling run would have measured one GPU). The launcher starts N fresh rank processes before anything touches the
GPU; under an external launcher a mismatch between --gpus and WORLD_SIZE is an error."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def run_bench(argv, env_extra=None, timeout=900):
    env = {k: v for (k, v) in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    proc = subprocess.run([sys.executable, BENCH] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          universal_newlines=True, timeout=timeout, cwd=ROOT)
    lines = [line for line in proc.stdout.splitlines() if line.startswith('{')]
    return (proc, [json.loads(line) for line in lines])


@pytest.mark.parametrize('world', [2, 3])
def test_launcher_starts_one_process_per_rank(world):
    """CPU rendezvous check: N children, each with its own RANK / LOCAL_RANK, one JSON line from rank 0."""
    (proc, lines) = run_bench(['--gpus', str(world), '--dry-launch'])
    assert proc.returncode == 0, proc.stderr
    assert len(lines) == 1
    assert {k: lines[0][k] for k in ('dry_launch', 'n_gpus', 'ranks_seen', 'local_rank')} == {
        'dry_launch': True, 'n_gpus': world, 'ranks_seen': world, 'local_rank': 0}


def test_eight_ranks_of_configs_3_rendezvous():
    """BASELINE.json configs[3] as the driver will launch it on an 8-GPU node (`--gpus 8 --height 256 --width 256 --batch 64`),
    rendezvous only (no GPU here): eight children, eight ranks seen, the shape-derived number of coder batches in flight."""
    (proc, lines) = run_bench(['--gpus', '8', '--height', '256', '--width', '256', '--batch', '64', '--dry-launch'], timeout=600)
    assert proc.returncode == 0, proc.stderr
    assert len(lines) == 1
    assert lines[0]['n_gpus'] == 8 and lines[0]['ranks_seen'] == 8 and lines[0]['coder_streams'] == 3 and lines[0]['usable_cpus'] >= 1


def test_coder_batches_in_flight_follow_the_map_size():
    """codec.default_nb_in_flight (what `bench.py --coder-streams 0` and `BatchCodec(nb_in_flight=None)` use): three for the short
    chains of 256x256 images, six for Kodak-sized maps (five until the end of round 5), eight for 2048x2048 (profiles/r03_depth_sweep.txt,
    r03_small_depth.txt, r05_coder_waves_per_block.log)."""
    from autoencoder_based_image_compression_amd import codec
    assert [codec.default_nb_in_flight(h, w) for (h, w) in ((64, 96), (256, 256), (512, 768), (1024, 1024), (2048, 2048))] == [3, 3, 6, 8, 8]


def test_world_size_mismatch_is_an_error():
    (proc, lines) = run_bench(['--gpus', '4', '--dry-launch'], {'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert proc.returncode != 0 and not lines
    assert '--gpus 4 but WORLD_SIZE is 2' in proc.stderr


def test_a_failing_rank_fails_the_launch():
    """A rank that dies must not leave the others waiting in a collective: non-zero exit, promptly."""
    (proc, lines) = run_bench(['--gpus', '2', '--dry-launch'], {'EAE_BENCH_FAIL_RANK': '1'}, timeout=120)
    assert proc.returncode != 0 and not lines


def test_under_an_external_launcher_the_flag_must_agree():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` (the driver's form): no second level of children."""
    env = {k: v for (k, v) in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    proc = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                           '--master-port', '29533', BENCH, '--gpus', '2', '--dry-launch'], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, universal_newlines=True, timeout=600, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr
    lines = [json.loads(line) for line in proc.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1 and lines[0]['dry_launch'] is True and lines[0]['n_gpus'] == 2 and lines[0]['ranks_seen'] == 2


@pytest.mark.gpu
def test_two_ranks_through_the_flag_on_one_gpu():
    """The real path with two ranks sharing this box's one GPU (gloo instead of RCCL): n_gpus == 2, and the all-reduced
    integer totals are the sum of the two one-rank runs on the same images."""
    common = ['--steps', '3', '--warmup', '1', '--batch', '4', '--no-cpu-baseline', '--no-single-image', '--min-seconds', '0']
    (proc, lines) = run_bench(['--gpus', '2'] + common, {'EAE_BENCH_SHARE_GPU': '1'})
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert len(lines) == 1
    two = lines[0]
    assert two['n_gpus'] == 2 and two['scaling'] == 'weak' and two['timing']['blocks'] == 1
    singles = []
    for offset in (0, 1):
        (proc, lines) = run_bench(['--gpus', '1', '--seed-offset', str(offset)] + common)
        assert proc.returncode == 0, proc.stderr[-2000:]
        assert lines[0]['n_gpus'] == 1
        singles.append(lines[0]['totals'])
    for key in ('bits', 'sse', 'dead_maps', 'images'):
        assert two['totals'][key] == singles[0][key] + singles[1][key], key
    assert two['totals']['images'] == 2*3*4
    assert len(two['host_cpu_ms_per_step']) == 2 and all(v > 0. for v in two['host_cpu_ms_per_step'])


@pytest.mark.gpu
def test_four_ranks_share_the_gpu_and_the_host_cpu_budget_is_reported():
    """Four ranks through the flag on this box's one GPU (gloo): what an 8-GPU node asks of the HOST is 8 x one rank's CPU time per
    step inside the container's CPU quota -- the line reports every rank's `host_cpu_ms_per_step` and `usable_cpus`, and the
    all-reduced totals cover four ranks."""
    common = ['--steps', '3', '--warmup', '1', '--batch', '2', '--no-cpu-baseline', '--no-side', '--min-seconds', '0']
    (proc, lines) = run_bench(['--gpus', '4'] + common, {'EAE_BENCH_SHARE_GPU': '1'})
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert len(lines) == 1
    four = lines[0]
    assert four['n_gpus'] == 4 and four['totals']['images'] == 4*3*2
    assert len(four['host_cpu_ms_per_step']) == 4 and all(v > 0. for v in four['host_cpu_ms_per_step'])
    assert four['usable_cpus'] >= 1 and 'one_stream_leg' in four and four['roofline']['per_kernel']['conv1_gdn1']['avg_ms'] > 0.


@pytest.mark.gpu
def test_one_rank_through_rccl():
    """`--force-nccl`: the process group of the N-GPU run (backend 'nccl' = RCCL on ROCm), joined by ONE rank on this box's one GPU:
    librccl loads, `init_process_group` succeeds with HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment (what `launch_ranks` gives its
    children), the barriers of the timed blocks (`dist.barrier(device_ids=...)`) and the statistics all-reduce run on the device,
    and the integer totals equal those of the same run without a process group."""
    common = ['--gpus', '1', '--steps', '3', '--warmup', '1', '--batch', '4', '--no-cpu-baseline', '--no-single-image', '--min-seconds', '0']
    (proc, lines) = run_bench(common + ['--force-nccl'])
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert len(lines) == 1
    grouped = lines[0]
    assert grouped['process_group']['backend'] == 'nccl' and grouped['process_group']['world_size'] == 1
    assert grouped['process_group']['collectives'] >= 4          # barriers on both sides of a timed block, MAX of the times, the totals
    (proc, lines) = run_bench(common)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert 'process_group' not in lines[0]
    assert grouped['totals'] == lines[0]['totals'] and grouped['n_gpus'] == 1
    assert grouped['scaling_measured'] is False and grouped['device']['whole_device'] is True
    # the exact-parity exchange (all_gather of the per-image values, `sharding.gather_per_image`) through RCCL as well
    (proc, lines) = run_bench(common + ['--force-nccl', '--exchange', 'all_gather'])
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert lines[0]['exchange'] == 'all_gather' and lines[0]['process_group']['backend'] == 'nccl'
    assert lines[0]['totals'] == grouped['totals']


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_eight_ranks_of_configs_3_share_the_gpu():
    """The 8-GPU run of BASELINE.json configs[3] rehearsed on this box's one GPU (EAE_BENCH_SHARE_GPU: eight rank processes, gloo for
    the collectives, 64 images of 256x256 per rank and step) under the box's CPU quota: every rank codes its own images (seed offset
    = rank), the integer totals of the line equal the sums of eight one-rank runs, with either exchange; the summed process CPU of
    the eight ranks per step is printed. No scaling is measured by this (`scaling_measured` false)."""
    shape = ['--height', '256', '--width', '256', '--batch', '64', '--steps', '2', '--warmup', '1', '--min-seconds', '0',
             '--no-cpu-baseline', '--no-side', '--no-dropin-surface']
    (proc, lines) = run_bench(['--gpus', '8'] + shape, {'EAE_BENCH_SHARE_GPU': '1'}, timeout=1500)
    assert proc.returncode == 0, proc.stderr[-3000:]
    eight = lines[0]
    assert eight['n_gpus'] == 8 and eight['scaling_measured'] is False and len(eight['host_cpu_ms_per_step']) == 8
    (proc, lines) = run_bench(['--gpus', '8', '--exchange', 'all_gather'] + shape, {'EAE_BENCH_SHARE_GPU': '1'}, timeout=1500)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert lines[0]['totals'] == eight['totals']
    sums = {'bits': 0, 'sse': 0, 'dead_maps': 0, 'images': 0}
    for rank in range(8):
        (proc, lines) = run_bench(['--gpus', '1', '--seed-offset', str(rank)] + shape)
        assert proc.returncode == 0, proc.stderr[-2000:]
        for key in sums:
            sums[key] += lines[0]['totals'][key]
    assert sums == eight['totals'], (sums, eight['totals'])
    print('8 ranks sharing one GPU, 64 x 256x256 per rank and step: process CPU per step and rank (ms) {0}, summed {1:.2f} ms per {2:.2f} ms step '
          '= {3:.2f} CPUs of the {4} this box allows'.format(eight['host_cpu_ms_per_step'], sum(eight['host_cpu_ms_per_step']), eight['ms_per_step'],
                                                          sum(eight['host_cpu_ms_per_step'])/eight['ms_per_step'], eight['usable_cpus']))
