"""r04: one Kodak image at a time (submit -> result on the host, BASELINE.json configs[1] literally): the wall time per image, and under
rocprofv3 --kernel-trace (scratch/r04/single_latency.sh) the kernels of one image with their start offsets, durations and the gaps between them."""
import os, sys, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'analyse':
    f = glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
    starts = [i for (i, r) in enumerate(rows) if 'conv1_kernel' in r[2]]
    (a, b) = (starts[-3], starts[-2])                      # one image in the steady state
    t0 = rows[a][0]
    print('one image: conv1 start to the next conv1 start %.1f us; kernels %d' % ((rows[b][0] - t0)/1e3, b - a))
    end_prev = None
    busy = 0
    for (s, e, n) in rows[a:b]:
        short = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
        gap = '' if end_prev is None else '%+7.1f' % ((s - end_prev)/1e3)
        print('  %8.1f  %7.1f us  gap %8s  %s' % ((s - t0)/1e3, (e - s)/1e3, gap, short))
        end_prev = e if end_prev is None else max(end_prev, e)
        busy += e - s
    print('  sum of kernel durations %.1f us; last kernel ends at %.1f us' % (busy/1e3, (end_prev - t0)/1e3))
    sys.exit(0)
import bench, torch
args = bench.parse_args(['--no-cpu-baseline', '--no-side'])
device = torch.device('cuda', 0)
torch.cuda.set_device(device)
ctx = bench.Context(args, device, 1, 0, bench.usable_cpus())
variables = bench.synthetic_model(1.)
n = int(os.environ.get('N_IMAGES', '100'))
alone = bench.run_pipeline(ctx, 1, n, 10, variables, 512, 768, coder_streams=1, transform_streams=1, use_graphs=True, serial=True)
print('one image at a time: %.4f ms per image (submit -> result on the host)' % (alone['elapsed']/n*1e3))
