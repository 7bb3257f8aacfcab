"""
Summarise the rocprofv3 PMC passes written by profiles/collect_pmc.sh into one JSON object per workload:
per kernel (mean over its dispatches after the warm-up evaluation) duration, FETCH_SIZE, WRITE_SIZE,
HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies 128-B read requests at 64 B:
MI355X_MICROARCH.md, HBM section; Infinity-Cache hits are included in both counters), MFMA busy
(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)) and the effective clock.

    python profiles/pmc_summarize.py gpurun_out/pmc_<tag> <key> [profiles/r02_pmc_traffic.json]
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([a-z_0-9]+)(<.*>)?', name)
    base = m.group(1) if m else name
    tiles = re.findall(r'TileCfg<(\w+), (\d+), (\d+), (\d+)', name)
    tag = '_'.join('%sx%sx%s' % t[1:] for t in tiles[:1])
    flags = re.findall(r'>, (\w+)(?:, (\w+))?>$', name)
    tail = ''
    m2 = re.search(r'>, ([\w, ]+)>\s*(\(|$)', name)
    if m2:
        tail = '_' + m2.group(1).replace(', ', '_')
    elif not tiles:
        m3 = re.match(r'[a-z_0-9]+<([\w, ]+)>\(', name)          # plain template arguments: apply_dma_kernel<0, 256>(...)
        if m3:
            tail = '_' + m3.group(1).replace(', ', '_')
    return base + ('_' + tag if tag else '') + tail


def read_pass(d):
    rows = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (int(r['Dispatch_Id']), r['Kernel_Name'])
            e = rows.setdefault(key, {'ns': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
            e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return rows


def per_kernel(rows, skip_first_half=True):
    """mean per kernel name over the dispatches of the LAST evaluation (the warm-up evaluation comes first)"""
    by = {}
    for (did, name), e in sorted(rows.items()):
        by.setdefault(name, []).append(e)
    out = {}
    for name, lst in by.items():
        if skip_first_half and len(lst) >= 2 and len(lst) % 2 == 0:
            lst = lst[len(lst) // 2:]
        agg = {'launches': len(lst)}
        for k in lst[0]:
            agg[k] = sum(e.get(k, 0.0) for e in lst) / len(lst)
        out[name] = agg
    return out


def main():
    d, key = sys.argv[1], sys.argv[2]
    dst = sys.argv[3] if len(sys.argv) > 3 else None
    fetch = per_kernel(read_pass(os.path.join(d, 'fetch')))
    write = per_kernel(read_pass(os.path.join(d, 'write')))
    busy = per_kernel(read_pass(os.path.join(d, 'busy')))
    kernels = {}
    for name in sorted(set(fetch) | set(write) | set(busy)):
        f, w, b = fetch.get(name, {}), write.get(name, {}), busy.get(name, {})
        ns = b.get('ns') or f.get('ns') or w.get('ns') or 0
        if ns < 20000 and 'featuremap' not in name and 'project' not in name:
            continue                                        # keep the summary to the kernels that matter
        e = {'launches_per_eval': int(max(f.get('launches', 0), w.get('launches', 0), b.get('launches', 0))),
             'ms_under_pmc': ns / 1e6}
        if 'FETCH_SIZE' in f:
            e['FETCH_SIZE_KB'] = f['FETCH_SIZE']
        if 'WRITE_SIZE' in w:
            e['WRITE_SIZE_KB'] = w['WRITE_SIZE']
        if 'FETCH_SIZE' in f and 'WRITE_SIZE' in w:
            e['GB_corrected'] = (2.0 * f['FETCH_SIZE'] + w['WRITE_SIZE']) * 1024 / 1e9
        if 'GRBM_GUI_ACTIVE' in b and b['GRBM_GUI_ACTIVE'] > 0:
            cyc = b['GRBM_GUI_ACTIVE'] / 8.0
            e['clock_GHz'] = cyc / b['ns']
            e['MFMA_BUSY'] = b.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cyc * 1024.0)
        kernels[short(name)] = e
    res = {'kernels': kernels,
           'correction': 'GB_corrected = (2 x FETCH_SIZE + WRITE_SIZE) KB: gfx950 tallies 128-B read requests at 64 B '
                         '(MI355X_MICROARCH.md, HBM); Infinity-Cache hits are included in both counters; one counter group per pass'}
    # one apply product = the full-tile launch + the ragged-remainder launch of the same epilogue (EPI: 0 Phi.B, 1 Phibar, 3 / 4 the
    # triangular products of the factor form); the figure is the mean over the products of one evaluation
    prod = {}
    for k, v in kernels.items():
        if 'GB_corrected' not in v:
            continue
        m = re.match(r'apply_dma_kernel_(?:float_|double_)?(\d)_', k) or re.match(r'apply_kernel_.*_(\d)$', k)
        if m:
            prod[m.group(1)] = prod.get(m.group(1), 0.0) + v['GB_corrected']      # each of a product's kernels runs once per evaluation
    if prod:
        res['apply_product_GB'] = {'EPI_' + e: gb for e, gb in sorted(prod.items())}
        main = [gb for e, gb in prod.items() if e in ('0', '1')] or list(prod.values())     # the two full N x K x K products
        res['apply_kernel_mean_GB_per_launch'] = sum(main) / len(main)
    gm = [v for k, v in kernels.items() if k.startswith('gram_kernel') and 'MFMA_BUSY' in v]
    if gm:
        res['gram_MFMA_BUSY'] = sum(v['MFMA_BUSY'] for v in gm) / len(gm)
    fm = [v for k, v in kernels.items() if (k.startswith('featuremap') or k.startswith('project')) and 'WRITE_SIZE_KB' in v]
    if fm:
        res['featuremap_WRITE_SIZE_GBs'] = sum(v['WRITE_SIZE_KB'] for v in fm) * 1024 / 1e9 / (sum(v['ms_under_pmc'] for v in fm) / 1e3)
    print(json.dumps(res, indent=1))
    if dst:
        allr = json.load(open(dst)) if os.path.exists(dst) else {}
        allr[key] = res
        json.dump(allr, open(dst, 'w'), indent=1)


if __name__ == '__main__':
    main()
