"""Per-kernel averages of every counter of the rocprofv3 --pmc passes of a session (scratch/r03_session1.sh: one pass per
counter set, `pmc_<i>/` sub-directories) -> one JSON: {kernel: {counter: average per dispatch, 'dispatches': n}}.

usage: python profiles/make_pmc_summary.py <session dir> <out.json>

Units as rocprofv3 reports them: SQ_* cycle counters are summed over the chip's SEs/XCDs (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*
in quad-cycles per MI355X_MICROARCH.md; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES in cycles), GRBM_GUI_ACTIVE summed over the 8 XCDs,
FETCH_SIZE / WRITE_SIZE in KB (FETCH_SIZE to be doubled on gfx950 for 16-byte-per-lane loads, as profiles/make_traffic.py does).
Kernel names are shortened to the function name and its template arguments."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'\(.*$', '', name)
    name = re.sub(r'^void ', '', name)
    return name.strip()


def main(session, out_path):
    table = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))   # kernel -> counter -> dispatch -> value
    for f in glob.glob(os.path.join(session, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
        with open(f) as handle:
            for r in csv.DictReader(handle):
                table[short(r['Kernel_Name'])][r['Counter_Name']][(f, r['Dispatch_Id'])] += float(r['Counter_Value'])
    summary = {}
    for (kernel, counters) in sorted(table.items()):
        entry = {}
        for (counter, per_dispatch) in sorted(counters.items()):
            values = list(per_dispatch.values())
            entry[counter] = round(sum(values)/len(values), 3)
            entry['dispatches'] = len(values)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in entry and entry.get('GRBM_GUI_ACTIVE'):
            # MFMA busy as a fraction of the kernel's duration: the counter is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE
            # over the 8 XCDs (the normalisation of profiles/make_traffic.py; a register-only MFMA stream reads 0.98)
            entry['mfma_busy_frac'] = round((entry['SQ_VALU_MFMA_BUSY_CYCLES']/1024.)/(entry['GRBM_GUI_ACTIVE']/8.), 4)
        if 'SQ_WAVE_CYCLES' in entry:
            for key in ('SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_LDS'):
                if key in entry:
                    entry[key.lower() + '_per_wave_cycle'] = round(entry[key]/max(entry['SQ_WAVE_CYCLES'], 1.), 4)
        summary[kernel] = entry
    with open(out_path, 'w') as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    for (kernel, entry) in summary.items():
        print(kernel[:70], {k: v for (k, v) in entry.items() if k != 'dispatches'})


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
