#!/usr/bin/env python
"""profiles/pmc_traffic.json (what bench.py's roofline.traffic reads) from the pmc_summary.json files scripts/pmc_run.sh leaves
under gpurun_out/<tag>/:   python scripts/make_pmc_traffic.py <round tag, e.g. r06_g>
expects gpurun_out/<tag>_pmc_c2 (the headline configuration) and optionally <tag>_pmc_c5; copies every <tag>_pmc_*/pmc_summary.md to
profiles/<tag>_pmc_<cfg>.md."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
CFG = {'c2': dict(users=10_000_000, items=1_000_000), 'c5': dict(users=12_500_000, items=125_000_000)}
WHAT = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* / TCC_* in separate passes over bench.py --steps 4 --warmup 1 --no-cpu-baseline ' \
       '--no-fit --no-probes --no-sharded-check --no-overlapped --no-configs%s; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB'


def entry(cfg):
    src = os.path.join(ROOT, 'gpurun_out', '%s_pmc_%s' % (tag, cfg), 'pmc_summary.json')
    rec = json.load(open(src))
    kern = {k: {'hbm_bytes_per_launch': v['hbm_bytes_per_launch'], 'FETCH_SIZE_KiB': v['FETCH_SIZE'], 'WRITE_SIZE_KiB': v['WRITE_SIZE']}
            for k, v in rec.items() if isinstance(v, dict) and 'hbm_bytes_per_launch' in v}
    return {'config': dict(CFG[cfg], dim=64, batch=1 << 20, loss='bpr', opt='adagrad', user_rows='doubled (slk_user_pingpong_begin): bench.py\'s default'),
            'source': 'profiles/%s_pmc_%s.md (%s)' % (tag, cfg, WHAT % ('' if cfg == 'c2' else ' --workload ' + cfg)), 'kernels': kern}


out = entry('c2')
out['others'] = [entry('c5')] if os.path.isdir(os.path.join(ROOT, 'gpurun_out', '%s_pmc_c5' % tag)) else []
json.dump(out, open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), 'w'), indent=1)
for d in sorted(os.listdir(os.path.join(ROOT, 'gpurun_out'))):
    if d.startswith(tag + '_pmc_'):
        shutil.copy(os.path.join(ROOT, 'gpurun_out', d, 'pmc_summary.md'), os.path.join(ROOT, 'profiles', d + '.md'))
print(json.dumps({k: round(v['hbm_bytes_per_launch'] / 1e9, 3) for k, v in out['kernels'].items()}))
