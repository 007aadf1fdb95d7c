"""The `configs` object of the default bench line: every BASELINE.json configuration besides the headline
one -- and the two scoring workloads -- as a compact leg of a few steps each, so that the driver's own run records them.

Each leg is `bench.py --workload ...` (the very command a maintainer would run by hand) in a process of its own, with W + K steps
and the same bracket (device sync, exactly K steps, sync); a leg that fails costs its own entry, never the line.  The legs run
AFTER the headline measurement and its fit() leg, before the CPU baseline."""
import json
import os
import subprocess
import sys
import time

from benchlib.common import ROOT

_QUIET = ['--no-cpu-baseline', '--no-fit', '--no-probes', '--no-sharded-check', '--no-overlapped', '--no-configs']
# name -> (arguments, how to read the leg's own JSON line)
# Step counts (round 6): every training leg runs SEVERAL prep chunks per call (C2-sized minibatches: 20 steps = 3 chunks; C4: 40 steps
# = 4 chunks of 10 minibatches; C3: 64 steps = 2 chunks of 32), as any epoch of a training run does -- with one chunk per call (rounds
# 4-5: 6-8 steps) the chunk's negatives + sorts sit in front of its passes with nothing to overlap and the leg measured a start-up,
# not a training rate (same box: C4 0.591 -> 0.531 ms per step, C3 1.661 -> 1.606, C5 1.046 -> 1.026, SparseAdam 0.900 -> 0.881)
LEGS = [
    ('c3', ['--workload', 'c3', '--steps', '64', '--warmup', '16'], 'step'),
    ('c4', ['--workload', 'c4', '--steps', '40', '--warmup', '10'], 'step'),
    ('c5_shard', ['--workload', 'c5', '--steps', '20', '--warmup', '5'] + _QUIET, 'main'),
    ('c2_sparse_adam', ['--opt', 'sparse_adam', '--steps', '20', '--warmup', '5'] + _QUIET, 'main'),
    ('c2_b65536', ['--batch', '65536', '--steps', '128', '--warmup', '32'] + _QUIET, 'main'),
    # the reference's own operating points (VERDICT r05 missing 2): its test configuration end to end, its test minibatch (1024) and
    # its constructor defaults (batch_size=256, adaptive hinge's num_negative_samples=5: factorization/implicit.py:80,88) on the C2 tables
    ('c1', ['--workload', 'c1', '--steps', '3'], 'step'),
    ('c2_b1024', ['--batch', '1024', '--steps', '2048', '--warmup', '256'] + _QUIET, 'main'),
    ('c2_b256_adaptive', ['--batch', '256', '--loss', 'adaptive_hinge', '--steps', '2048', '--warmup', '256'] + _QUIET, 'main'),
    # SURVEY.md 8(d) "variants to report": 20 % left padding (C4), Zipf(1.0) positive items (C2)
    ('c4_padded', ['--workload', 'c4', '--pad-frac', '0.2', '--steps', '40', '--warmup', '10'], 'step'),
    ('c2_zipf_items', ['--item-zipf', '1.0', '--steps', '20', '--warmup', '5'] + _QUIET, 'main'),
    ('predict', ['--workload', 'predict', '--steps', '400', '--warmup', '40'], 'scoring'),
    ('eval', ['--workload', 'eval', '--steps', '5', '--warmup', '2'], 'scoring'),
]


def _compact(kind, rec):
    roof = rec.get('roofline', {})
    out = {'ms_per_step': round(rec['ms_per_step'], 5), 'value': float('%.5g' % rec['value']), 'unit': rec['unit'],
           'steps': rec['steps'], 'warmup': rec['warmup']}
    if kind == 'main':
        out.update({'alg_bytes_per_unit': roof.get('step_alg_bytes_per_interaction'), 'dominant_kernel': roof.get('kernel'),
                    'frac': round(roof.get('frac', 0.0), 4), 'step_frac': round(roof.get('step_frac_of_peak', 0.0), 4)})
        k = roof.get('kernels', {})
        out['kernel_ms'] = {n: round(v['avg_ms'], 4) for n, v in k.items()}
        layout = rec.get('config', {}).get('item_bias_layout', '')
        if layout and not layout.startswith('two arrays'):
            out['item_bias_layout'] = 'interleaved with its Adagrad accumulator for the run (slk_bias_shadow_begin, outside the timed region)'
        if rec.get('config', {}).get('user_row_layout', '').startswith('the user table doubled'):
            out['user_row_layout'] = 'doubled for the run (slk_user_pingpong_begin, outside the timed region): no pre-step-row record'
        if roof.get('single_occurrence_fast_path'):
            out['single_occurrence_fast_path'] = ('once-only items updated by the user pass (option item_single_min_items): frac = the user pass '
                                                  'on the whole step\'s algorithmic bytes')
        if roof.get('persistent_epoch_kernel'):
            out['persistent_us_per_minibatch'] = round(roof['persistent_epoch_kernel']['us_per_minibatch'], 2)
    elif kind == 'step':
        out.update({'alg_bytes_per_unit': roof.get('alg_bytes_per_interaction', roof.get('alg_bytes_per_timestep')),
                    'step_frac': round(roof.get('step_frac_of_peak', 0.0), 6 if roof.get('step_frac_of_peak', 1.0) < 0.01 else 4)})
        if 'variants' in roof:  # c1: the whole fit() under the reference's default optimizer and under row-sparse Adagrad
            out['variants'] = {n: {'fit_seconds': round(v['fit_seconds'], 5), 'interactions_per_s': float('%.5g' % v['interactions_per_s']),
                                   'us_per_minibatch_end_to_end': round(v['us_per_minibatch_end_to_end'], 2),
                                   'alg_bytes_per_minibatch': v['alg_bytes_per_minibatch'], 'frac': float('%.3g' % v['frac'])}
                               for n, v in roof['variants'].items()}
        if 'kernels' in roof:
            out['kernel_ms'] = {n: round(v['avg_ms'], 4) for n, v in roof['kernels'].items()}
        if 'ms_per_step_by_class' in roof:
            out['kernel_ms'] = {n: round(v, 4) for n, v in roof['ms_per_step_by_class'].items()}
    else:
        out.update({'bound': roof.get('bound'), 'dominant_kernel': roof.get('kernel'), 'frac': round(roof.get('frac', 0.0), 4),
                    'device_ms_per_call': round(roof.get('device_ms_per_call', 0.0), 5)})
    return out


def run_config_legs(extra_args=(), timeout=240):
    """{leg: compact record} + what they cost; `extra_args`: passed to every leg (e.g. --set options of the parent run)."""
    out, t_all = {}, time.perf_counter()
    for name, argv, kind in LEGS:
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py')] + argv + list(extra_args)
        t0 = time.perf_counter()
        try:
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            line = [l for l in res.stdout.decode().splitlines() if l.startswith('{')][-1]
            out[name] = _compact(kind, json.loads(line))
        except Exception as e:  # noqa: BLE001 -- a reported extra, never a reason to lose the line
            out[name] = {'error': repr(e)[:200]}
        out[name]['leg_seconds'] = round(time.perf_counter() - t0, 1)
    out['note'] = ('compact legs of the other BASELINE.json configurations and the scoring workloads: `python bench.py <args>` each in '
                   'its own process (benchlib/legs.py: LEGS), W + K steps, device-synchronised bracket; frac = dominant kernel, '
                   'step_frac = whole step, both of 8 TB/s on algorithmic bytes (eval: of the f32 MFMA peak)')
    out['seconds'] = round(time.perf_counter() - t_all, 1)
    return out
