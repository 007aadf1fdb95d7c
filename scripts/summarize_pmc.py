"""Summarises rocprofv3 --pmc passes (scripts/pmc_run.sh) into profiles/<name>.md + .json.

usage: python scripts/summarize_pmc.py gpurun_out/<tag> profiles/<name> "<title>"

Per kernel class (user pass, item pass, sequence pass): mean counter value per launch.  HBM
traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB): on gfx950 FETCH_SIZE
tallies 128-B read requests at 64 B (MI355X_MICROARCH.md "HBM"; our own calibration in
gpurun_out/calib_*: a 1 GiB streaming copy reports FETCH 0.5 GiB / WRITE 1.0 GiB, a 1M x 256-B random
row gather reports 0.73 of its bytes), so 2 x FETCH is an upper bound for the random-row part."""
import json
import os
import re
import sqlite3
import sys


def cls(name):
    if 'k_item_pass' in name:  # BloomEmbedding tables: the rows pass (PART 1) and the bias pass (PART 2) are different kernels
        m = re.search(r'k_item_pass<\s*\d+,\s*\d+,\s*\d+,\s*\d+,\s*(\d+)', name)
        part = m.group(1) if m else '0'
        return 'k_item_pass' if part == '0' else 'k_item_pass<PART %s>' % part
    for key in ('k_user_pass', 'k_score_pass', 'k_adaptive_select', 'k_seq_pass', 'k_shard_user_pass', 'k_mt_jump', 'k_mt_stream', 'k_mt_prefix', 'k_rs_scatter', 'k_rs_hist',
                'k_score_gemm'):
        if key in name:
            return key
    return None


def main():
    src, dst, title = sys.argv[1:4]
    agg = {}
    for group in sorted(os.listdir(src)):
        d = os.path.join(src, group)
        if not os.path.isdir(d):
            continue
        dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith('.db')]
        for db in dbs:
            c = sqlite3.connect(db)
            q = 'select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name'
            for kname, cname, total, n in c.execute(q):
                k = cls(kname)
                if k is None:
                    continue
                a = agg.setdefault(k, {}).setdefault(cname, [0.0, 0])
                a[0] += total
                a[1] += n
    out = {}
    for k, cs in agg.items():
        out[k] = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
        out[k]['launches'] = max(v[1] for v in cs.values())
        if 'FETCH_SIZE' in out[k] and 'WRITE_SIZE' in out[k]:
            out[k]['hbm_bytes_per_launch'] = (2.0 * out[k]['FETCH_SIZE'] + out[k]['WRITE_SIZE']) * 1024.0
        # derived: how long a request of the L2 to the memory side stays outstanding (Little: level / requests, in L2 clocks)
        for kind in ('RD', 'WR'):
            req, lvl = out[k].get('TCC_EA0_%sREQ_sum' % kind), out[k].get('TCC_EA0_%sREQ_LEVEL_sum' % kind)
            if req and lvl:
                out[k]['derived_%s_request_latency_clocks' % kind.lower()] = lvl / req
    with open(dst + '.json', 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    with open(dst + '.md', 'w') as f:
        f.write('# %s\n\nrocprofv3 --pmc (separate passes per counter group; mean per launch)\n\n' % title)
        for k, cs in sorted(out.items()):
            f.write('## %s\n\n| counter | mean per launch |\n|---|---|\n' % k)
            for c, v in sorted(cs.items()):
                f.write('| %s | %.6g |\n' % (c, v))
            f.write('\n')
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
