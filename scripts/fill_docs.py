#!/usr/bin/env python
"""Regenerates the passages between <!--TAG--> ... <!--/TAG--> anchors of DESIGN.md / profiles/README.md / README.md from
profiles/r03_final_*.json (run after a verification pass has been copied into profiles/)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')


def L(name):
    return json.load(open(os.path.join(P, name)))


b = L('r03_final_bench.json')
r = b['roofline']
k, ov = r['kernels'], r['overlapped']
fit = b.get('fit_end_to_end', {})
sh = b.get('sharded_world1_consistency', {}).get('ms_per_step_second_call', {})
cpu = b.get('cpu_baseline', {})
stats = open(os.path.join(P, 'r03_final_kernel_stats.md')).read()
m_u = re.search(r'k_user_pass<[^|]*\| (\d+) \| [\d.]+ \| ([\d.]+)', stats)
m_i = re.search(r'k_item_pass<[^|]*\| (\d+) \| [\d.]+ \| ([\d.]+)', stats)
try:
    pk = L('r03_final_prof_bench.json')['roofline']['kernels']
    prof_note = ('; the bench line printed by that traced run (`r03_final_prof_bench.json`) has %.1f / %.1f µs from its own HIP events: the '
                 'two clocks agree, the tracer itself costs the kernels a few per cent' % (pk['user_pass']['avg_ms'] * 1e3, pk['item_pass']['avg_ms'] * 1e3))
except Exception:
    prof_note = ''
headline = (
    "Driver-style run (`bench.py --steps 20 --warmup 5`, `profiles/r03_final_bench.json`; a bare ctx: negatives, sorts and passes in "
    "order on one stream): **%.3f G interactions/s**, %.4f ms per minibatch, whole step **%.3f** of 8 TB/s on the 3136 algorithmic "
    "bytes (r02: 1.275 G/s, 0.822 ms, 0.500) = user pass %.4f ms + item pass %.4f ms + sampler %.3f + sorts %.3f (+ gaps); dominant "
    "kernel (user pass) 1576 B × 2²⁰ ÷ %.4f ms = %.2f TB/s = **%.3f**, item pass 1560 B × 2²⁰ ÷ %.4f ms = %.2f TB/s = **%.3f** (r02: "
    "0.635 / 0.631).  rocprofv3 of the same command (`r03_final_kernel_stats.md`): user pass %s µs, item pass %s µs (means over %s "
    "launches%s).  `roofline.overlapped`, the same minibatches the way `fit()` runs them (next chunk's negatives + sorts on a second "
    "stream; second call of its kind): **%.4f ms per step = %.3f G/s (%.3f)** with the passes at %.4f + %.4f ms beside the sorts.  "
    "`bench.py` times the FIRST 20-minibatch call of the process, which is 2–6 %% slower than the calls after it whether the prep "
    "overlaps or not (`r03_p_first_call_of_a_process.txt`): the GPU leaving its idle power state — after 0.5 s of idle a fourth call is "
    "as slow as the first (`r03_t_first_call_after_idle.txt`); W = 5 warm-up minibatches are 4 ms, so `bench.py` runs its copy / triad "
    "probes before the warm-up, which takes about half of it off.  `fit_end_to_end` (the drop-in `fit()`, 2²⁵ interactions × 10 epochs, id "
    "upload included): **%.3f G interactions/s** (r02: 0.81).  Row-sharded path at world 1: %.2f ms per step against %.2f fused (r02: "
    "2.39 / 0.94).  Probes: best copy %.2f TB/s (chunked, non-temporal), plain grid-stride copy %.2f, triad %.2f; the step's "
    "algorithmic accesses alone %.3f ms.  CPU baseline, same run: the reference itself %.3f M interactions/s at %d threads, the C "
    "port %.3f M/s on one core."
    % (b['value'] / 1e9, b['ms_per_step'], r['step_frac_of_peak'], k['user_pass']['avg_ms'], k['item_pass']['avg_ms'],
       r['other_ms_per_step']['sample'], r['other_ms_per_step']['prep'],
       k['user_pass']['avg_ms'], k['user_pass']['achieved_GBs'] / 1e3, k['user_pass']['achieved_GBs'] / 8000.0,
       k['item_pass']['avg_ms'], k['item_pass']['achieved_GBs'] / 1e3, k['item_pass']['achieved_GBs'] / 8000.0,
       m_u.group(2) if m_u else '?', m_i.group(2) if m_i else '?', m_u.group(1) if m_u else '?', prof_note,
       ov['ms_per_step'], ov['interactions_per_s'] / 1e9, ov['step_frac_of_peak'], ov['kernels']['user_pass']['avg_ms'],
       ov['kernels']['item_pass']['avg_ms'],
       fit.get('interactions_per_s', 0) / 1e9, sh.get('sharded_world1', 0), sh.get('fused', 0),
       r['measured']['copy_GBs'] / 1e3, r['measured']['variants_GBs']['copy_plain'] / 1e3, r['measured']['triad_GBs'] / 1e3,
       r['ceiling']['ms_per_step'], cpu.get('value', 0) / 1e6, cpu.get('cores', 0), cpu.get('port', {}).get('value', 0) / 1e6))


def zl(fmt, names, key):
    out = []
    for n in names:
        d = L(fmt % n)
        out.append('%.2f' % d['roofline']['kernels'][key]['avg_ms'])
    return ' / '.join(out)


zu = '**' + zl('r03_final_bench_userzipf_%s.json', ('0.8', '1.0', '1.2'), 'user_pass') + '**'
zi = '**' + zl('r03_final_bench_zipf_%s.json', ('0.8', '1.0', '1.2'), 'item_pass') + '**'
c3, c4, c5, sa = (L('r03_final_bench_%s.json' % w) for w in ('c3', 'c4', 'c5', 'sparse_adam'))
bt = {n: L('r03_final_bench_batch_%s.json' % n) for n in ('256', '1024', '65536', '1048576')}
other = ("C3 %.0f M interactions/s (%.2f; r02 138, 0.46), C4 %.2f G timesteps/s (%.2f; r02 1.31, 0.34 — the prep overlap and the key "
         "prefetch), C5 shard %.2f G/s (%.2f; r02 0.89, 0.35), SparseAdam on C2 %.2f G/s (%.2f of its 4696-B roofline).  Minibatch lines "
         "on the C2 tables: 256 → %.1f µs, 1024 → %.1f µs (both the persistent kernel), 65 536 → %.0f µs, 2²⁰ → %.0f µs."
         % (c3['value'] / 1e6, c3['roofline']['step_frac_of_peak'], c4['value'] / 1e9, c4['roofline']['step_frac_of_peak'],
            c5['value'] / 1e9, c5['roofline']['step_frac_of_peak'], sa['value'] / 1e9, sa['roofline']['step_frac_of_peak'],
            bt['256']['ms_per_step'] * 1e3, bt['1024']['ms_per_step'] * 1e3, bt['65536']['ms_per_step'] * 1e3, bt['1048576']['ms_per_step'] * 1e3))
try:
    ad = [json.loads(l) for l in open(os.path.join(P, 'r03_final_adaptive_small_routes.jsonl'))]
    g = {(d['shape'], d['batch'], d['route']): d['us_per_minibatch'] for d in ad}
    pn = L('r03_final_bench_poolnet_256x10.json')
    other += ("  Reference-default minibatches (256): adaptive hinge with 5 draws inside the persistent kernel %.1f µs on 10⁶ × 10⁵ tables / "
              "%.1f µs at the MovieLens-100K shape (launches %.1f / %.1f); PoolNet 256 sequences × 10 timesteps %.1f µs (launches: the "
              "persistent form is slower, §6)."
              % (g[('mid', 256, 'persistent kernel')], g[('c1', 256, 'persistent kernel')], g[('mid', 256, 'launches')],
                 g[('c1', 256, 'launches')], pn['ms_per_step'] * 1e3))
except Exception as e:
    print('small-batch lines not added:', repr(e))
final_row = ("`scripts/gpu_r03_final.sh`: the round's last verification pass on one box — `pytest -m gpu` tail, the default bench line (with "
             "`roofline.overlapped`, the probes, `fit_end_to_end`, the sharded world-1 check, the reference CPU baseline), rocprofv3 kernel stats and "
             "PMC traffic of the same workload (`pmc_traffic.json` is refreshed from it), C3 / C4 / C5, SparseAdam, the minibatch-size lines, "
             "Zipf items / users / both: **%.3f G interactions/s, %.4f ms per step (0.%s of the roofline), fit() %.2f G/s**"
             % (b['value'] / 1e9, b['ms_per_step'], ('%.3f' % r['step_frac_of_peak'])[2:], fit.get('interactions_per_s', 0) / 1e9))
readme = ("**%.2f G interactions/s** kernel-side as the driver measures it (the first 20-minibatch call of a process, everything in order on "
          "one stream: %.3f ms per step, **%.0f %%** of the 8 TB/s roofline on algorithmic bytes; user pass %.0f %%, item pass %.0f %%), "
          "**%.2f G/s** in the steady state with the next chunk's negatives and sorts on a second stream (what `fit()` runs: %.3f ms, %.0f %%); "
          "the drop-in `fit()` end to end **%.2f G/s** (2²⁵ interactions × 10 epochs, id upload included; round 2: 0.81).  The two passes move "
          "≈ 3.9 GB of real HBM traffic per step at 5.6–6.5 TB/s — the chip's best copy is %.1f TB/s — so what separates the step from the 70 %% "
          "target is bytes exactness needs (the pre-step user-row records) and the sorts, not idle bandwidth (DESIGN.md §7).  C3 %.2f G "
          "interactions/s, C4 %.2f G timesteps/s, per-GPU shard of the 1B-item table %.2f G/s, SparseAdam on C2 %.2f G/s.  `cpu_baseline` is "
          "Spotlight's own CPU PyTorch path timed on the box's host cores in the same run (%.2f M interactions/s at %d threads)."
          % (b['value'] / 1e9, b['ms_per_step'], 100 * r['step_frac_of_peak'], 100 * k['user_pass']['achieved_GBs'] / 8000.0,
             100 * k['item_pass']['achieved_GBs'] / 8000.0, ov['interactions_per_s'] / 1e9, ov['ms_per_step'], 100 * ov['step_frac_of_peak'],
             fit.get('interactions_per_s', 0) / 1e9, r['measured']['copy_GBs'] / 1e3, c3['value'] / 1e9, c4['value'] / 1e9, c5['value'] / 1e9,
             sa['value'] / 1e9, cpu.get('value', 0) / 1e6, cpu.get('cores', 0)))
subs = {'README_HEADLINE': readme, 'HEADLINE': headline, 'ZIPF_USERS': zu, 'ZIPF_ITEMS': zi, 'OTHER': other, 'FINAL': final_row}
for fn in ('DESIGN.md', 'profiles/README.md', 'README.md'):
    p = os.path.join(ROOT, fn)
    s = open(p).read()
    for tag, v in subs.items():
        s = re.sub(r'<!--%s-->.*?<!--/%s-->' % (tag, tag), lambda m: '<!--%s-->%s<!--/%s-->' % (tag, v, tag), s, flags=re.S)
    open(p, 'w').write(s)
print(headline)
print(other)
