"""r04: what a 20-step block (the driver's --steps 20) pays at its two ends. Reads a rocprofv3 kernel trace of a few blocks, splits it at the
idle gaps between blocks, and prints per block: its span, the kernel-busy profile of its first and last milliseconds (how many transform
kernels run at once per 0.25 ms bucket), and the last kernels to finish.   python scratch/r04/block_edges.py <trace dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
def short(n):
    for k in ('conv1_kernel', 'conv_gemm_split_kernel<0', 'conv_gemm_split_kernel<1', 'conv_gemm_split_kernel<2', 'tconv3_kernel', 'latent_quarter', 'bac_encode', 'bac_decode',
              'binarise', 'debinarise', 'emit_kernel', 'publish', 'coder_maps', 'decoder_maps', 'hist_kernel', 'compare', 'mark', 'collect', 'map_means', 'quantize'):
        if k in n:
            return k.replace('conv_gemm_split_kernel<0', 'conv3').replace('conv_gemm_split_kernel<1', 'conv2').replace('conv_gemm_split_kernel<2', 'tconv1/2').replace('_kernel', '')
    return n[:24]
TRANSFORM = ('conv1', 'conv2', 'conv3', 'tconv1/2', 'tconv3', 'latent_quarter')
# blocks: split where nothing runs for > 300 us
blocks, cur, end = [], [], None
for (s, e, n) in rows:
    if end is not None and s - end > 300000 and cur:
        blocks.append(cur); cur = []
    cur.append((s, e, short(n)))
    end = e if end is None else max(end, e)
if cur:
    blocks.append(cur)
for b in blocks:
    nconv1 = sum(1 for k in b if k[2] == 'conv1')
    if nconv1 < 15:
        continue
    t0 = min(k[0] for k in b); t1 = max(k[1] for k in b)
    print('block: %d batches, span %.3f ms = %.4f ms per batch' % (nconv1, (t1 - t0)/1e6, (t1 - t0)/1e6/nconv1))
    # steady-state rate from the middle conv1 starts
    c1 = sorted(k[0] for k in b if k[2] == 'conv1')
    mid = (c1[-5] - c1[4])/1e6/(len(c1) - 9)
    print('  conv1 to conv1 in the middle of the block: %.4f ms;  span - batches x that = %.3f ms' % (mid, (t1 - t0)/1e6 - nconv1*mid))
    for (name, lo, hi) in (('first', t0, t0 + 6000000), ('last', t1 - 6000000, t1)):
        line = []
        for q in range(24):
            a, z = lo + q*250000, lo + (q + 1)*250000
            busy = collections.Counter()
            for (s, e, n) in b:
                ov = min(e, z) - max(s, a)
                if ov > 0:
                    busy['t' if n in TRANSFORM else 'c'] += ov
            line.append('%.1f/%.1f' % (busy['t']/250000., busy['c']/250000.))
        print('  %s 6 ms, transform / coder kernels running at once per 0.25 ms: %s' % (name, ' '.join(line)))
    tail = sorted(b, key=lambda k: k[1])[-14:]
    print('  last to finish (start, end before the block end, us): ' + '; '.join('%s %d..%d' % (n, (t1 - s)//1000, (t1 - e)//1000) for (s, e, n) in tail))
    head = sorted(b)[:10]
    print('  first to start (start, end after the block start, us): ' + '; '.join('%s %d..%d' % (n, (s - t0)//1000, (e - t0)//1000) for (s, e, n) in head))
