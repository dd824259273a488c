"""Builds profiles/traffic_conv_gemm.json from rocprofv3 --pmc passes over `bench.py` (one counter set per pass, as
MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).

usage: python profiles/make_traffic.py <dir with pmc_fetch/ pmc_write/ pmc_mfma/ sub-directories> <batch>

FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests of 16-B/lane loads at 64 B).
The four conv GEMM launches of a step are told apart by their template instance."""
import collections
import csv
import glob
import json
import os
import sys


def rows_of(directory):
    files = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit('no counter_collection.csv under ' + directory)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    return rows


# The four launches of a step are instances of conv_gemm_split_kernel<epilogue, waves per block> (0 none / 1 GDN / 2 IGDN; 4); the two transposed
# convolutions share an instance, so a launch is identified by (instance, grid size in work-items). Grid of a layer
# (conv_gemm_split.hip: launch_split): 8 x ceil((tiles of the largest XCD share + cut tiles) / 4) blocks of 256 threads.
def grid_items(batch, positions_per_image, phases, cut, cus=256):
    tiles = batch*positions_per_image//32
    share = -(-tiles//8)*phases
    simds = 4*cus
    d = 0
    if cut:
        k = max(1, min(3, tiles*phases//simds))
        d = min(share, (cus//8)*4*k)
    return -(-(share + d)//4)*8*256


def instances(batch):
    px = 512*768
    return {'conv2_gdn2': ('conv_gemm_split_kernel<1,', grid_items(batch, px//64, 1, True)),
            'conv3': ('conv_gemm_split_kernel<0,', grid_items(batch, px//256, 1, True)),
            'tconv1_igdn5': ('conv_gemm_split_kernel<2,', grid_items(batch, px//256, 4, False)),
            'tconv2_igdn6': ('conv_gemm_split_kernel<2,', grid_items(batch, px//64, 4, False))}


INSTANCES = {}


def per_kernel(directory, counter, name):
    """Counter values of every dispatch of one conv GEMM launch of the step."""
    (instance, grid) = INSTANCES[name]
    out = collections.OrderedDict()
    for r in rows_of(directory):
        if r['Counter_Name'] != counter or instance not in r['Kernel_Name'] or int(r['Grid_Size']) != grid:
            continue
        out[int(r['Dispatch_Id'])] = out.get(int(r['Dispatch_Id']), 0.) + float(r['Counter_Value'])
    return [out[k] for k in sorted(out)]


def main():
    (root, batch) = (sys.argv[1], int(sys.argv[2]))
    INSTANCES.update(instances(batch))
    names = ['conv2_gdn2', 'conv3', 'tconv1_igdn5', 'tconv2_igdn6']     # launch order inside one step
    pixels = batch*512*768
    # algorithmic bytes per launch: input activations + output activations, float32, each read / written once
    # (DESIGN.md section 5): conv2 128ch at /4 -> /8, conv3 /8 -> /16, tconv1 /16 -> /8, tconv2 /8 -> /4
    act = {4: pixels//16*128*4, 8: pixels//64*128*4, 16: pixels//256*128*4}
    algorithmic = {'conv2_gdn2': act[4] + act[8], 'conv3': act[8] + act[16], 'tconv1_igdn5': act[16] + act[8],
                   'tconv2_igdn6': act[8] + act[4]}
    result = {}
    for name in names:
        fetch = per_kernel(os.path.join(root, 'pmc_fetch'), 'FETCH_SIZE', name)
        write = per_kernel(os.path.join(root, 'pmc_write'), 'WRITE_SIZE', name)
        busy = per_kernel(os.path.join(root, 'pmc_mfma'), 'SQ_VALU_MFMA_BUSY_CYCLES', name)
        active = per_kernel(os.path.join(root, 'pmc_mfma'), 'GRBM_GUI_ACTIVE', name)
        result[name] = {
            'dispatches': len(fetch),
            'hbm_read_bytes': 2.*1024.*sum(fetch)/len(fetch),
            'hbm_write_bytes': 1024.*sum(write)/len(write),
            'algorithmic_bytes': algorithmic[name],
            'mfma_busy_fraction_of_kernel_cycles': round((sum(busy)/1024.)/(sum(active)/8.), 3) if active else None,
        }
    result['hbm_bytes_per_launch'] = sum(result[n]['hbm_read_bytes'] + result[n]['hbm_write_bytes'] for n in names)/4.
    result['algorithmic_bytes_per_launch'] = sum(algorithmic.values())/4.
    result['note'] = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE (separate passes) on `python3 bench.py '
                      '--steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-side --no-dropin-surface --no-transforms-alone --transform-streams 1 --no-graphs --coder-streams 3` (batch {}; scratch/r06/collect_final.sh); counters are in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md '
                      '(16-B/lane loads are tallied at half); FETCH_SIZE counts L2 misses including Infinity Cache hits; per-launch average '
                      'over the four conv GEMM launches of a step. MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs).'
                      .format(batch))
    json.dump(result, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'traffic_conv_gemm.json'), 'w'), indent=1)
    print(json.dumps(result, indent=1))


if __name__ == '__main__':
    main()
