"""The `roofline` object of the C2 / C5 line (bench.py): per-kernel figures on algorithmic bytes from the engine's hipEvent timers,
the overlapped leg, the probes, and for the row-sharded path the xGMI accounting."""
from benchlib.common import HBM_PEAK_GBS, algorithmic_bytes, pmc_traffic


def build_roofline(args, *, world, K, B, D, elapsed, prof, prof_ov, elapsed_ov, elapsed_ov_prof, probes, ceiling, trainer,
                   xgmi_rows, denominators, xgmi_bytes=None):
    """-> (value, roofline dict).  `prof` / `prof_ov`: {class: (launches, ms)} of the instrumented K steps (in line / overlapped);
    `trainer`: the ShardedBilinearTrainer of a row-sharded run or None; `denominators`: rank 0's one-GPU rates at N > 1."""
    value = world * K * B / elapsed
    s_words = 1 if args.opt == 'adagrad' else 2  # (adam_dense: the per-row figure; its full-table sweep is extra)
    ub, ib = algorithmic_bytes(D, s_words)
    kern = {}
    for name, per_int in (('user_pass', ub), ('item_pass', ib)):
        n, ms = prof[name]
        avg_s = ms / max(n, 1) * 1e-3
        kern[name] = {'launches': n, 'avg_ms': ms / max(n, 1), 'alg_bytes_per_launch': per_int * B,
                      'achieved_GBs': per_int * B / avg_s / 1e9 if avg_s > 0 else 0.0}
    step_bytes, step_bytes_note = ub + ib, None
    if args.loss == 'adaptive_hinge':
        # SURVEY.md 8(d), the C3 accounting on plain tables (H = 1): the forward reads 1 user row + (1 + n) item rows; the update
        # touches the user row and the 2 live item rows (write + state read + write each) -- (2 + n) * 4D + 3 * 12D + ~90 B
        n = args.n_neg
        step_bytes = (2 + n) * 4 * D + 36 * D * s_words + 90
        step_bytes_note = ('adaptive hinge, n_neg %d, plain tables: (2 + n) * 4D forward reads + 3 rows x (write + state R/W) + ~90 B of '
                           'ids and biases (SURVEY.md 8(d), the C3 formula with H = 1); the per-kernel figures keep the pair accounting' % n)
    dom = max(kern, key=lambda k: kern[k]['avg_ms'])
    roof = {'bound': 'hbm', 'kernel': 'k_' + dom, 'achieved': kern[dom]['achieved_GBs'],
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': kern[dom]['achieved_GBs'] / HBM_PEAK_GBS,
            'traffic': pmc_traffic(args, 'k_' + dom) if trainer is None else None,
            'traffic_source': 'committed file (profiles/pmc_traffic.json), not measured by this run',
            'traffic_note': 'HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE from profiles/pmc_traffic.json '
                            '(rocprofv3 --pmc passes over this workload: PMC counters cannot be read from inside the '
                            'benchmark process); null if no committed measurement matches',
            'kernels': kern,
            'step_alg_bytes_per_interaction': step_bytes,
            'step_frac_of_peak': value / world * step_bytes / (HBM_PEAK_GBS * 1e9),
            'other_ms_per_step': {k: prof[k][1] / K for k in ('sample', 'prep', 'exchange', 'dense_sweep', 'epoch')}}
    if step_bytes_note:
        roof['step_alg_bytes_note'] = step_bytes_note
    if prof['score'][0]:
        roof['other_ms_per_step']['score'] = prof['score'][1] / K
    if prof_ov is not None:
        kov = {}
        for name, per_int in (('user_pass', ub), ('item_pass', ib)):
            n, ms = prof_ov[name]
            avg_s = ms / max(n, 1) * 1e-3
            kov[name] = {'avg_ms': ms / max(n, 1), 'achieved_GBs': per_int * B / avg_s / 1e9 if avg_s > 0 else 0.0,
                         'frac': per_int * B / avg_s / 1e9 / HBM_PEAK_GBS if avg_s > 0 else 0.0}
        roof['overlapped'] = {'ms_per_step': elapsed_ov / K * 1e3, 'interactions_per_s': K * B / elapsed_ov,
                              'step_frac_of_peak': K * B / elapsed_ov * (ub + ib) / (HBM_PEAK_GBS * 1e9),
                              'kernels': kov, 'ms_per_step_with_kernel_timers': elapsed_ov_prof / K * 1e3,
                              'other_ms_per_step': {k: prof_ov[k][1] / K for k in ('sample', 'prep')},
                              'note': 'the same K minibatches with option overlap_prep = 1 (what fit() sets on its ctx): the next '
                                      'chunk\'s negatives + sorts on a second stream beside the passes; second call of its kind '
                                      '(the line\'s value is the same K minibatches in order on one stream).  Beside the sorts every pass runs '
                                      'longer, the step shorter.'}
    if prof['epoch'][0]:
        roof['persistent_epoch_kernel'] = {'launches': prof['epoch'][0], 'us_per_minibatch': prof['epoch'][1] / K * 1e3,
                                           'note': 'every minibatch of a chunk inside one cooperative launch (slk_epoch.hip)'}
        if not kern['user_pass']['launches'] and prof['epoch'][1] > 0:
            # minibatches <= 1024: no pass launches at all -- the dominant (only) training kernel is the persistent one, both
            # phases of every minibatch of the chunk; its algorithmic bytes are the whole step's
            ach = step_bytes * B * K / (prof['epoch'][1] * 1e-3) / 1e9
            roof.update({'kernel': 'k_bilinear_epoch', 'achieved': ach, 'frac': ach / HBM_PEAK_GBS, 'traffic': None,
                         'bound': 'hbm by its bytes; at this minibatch size the kernel is bound by its two grid barriers per minibatch'})
    if probes:
        roof['measured'] = probes
    if ceiling:
        # the step's algorithmic accesses alone (slk_probe_step_ceiling): what exact grouping + hand-over cost on top
        tot = ceiling['user_side_ms'] + ceiling['item_side_ms']
        ceiling.update({'ms_per_step': tot, 'step_frac_of_peak': (ub + ib) * B / (tot * 1e-3) / (HBM_PEAK_GBS * 1e9),
                        'user_side_frac_of_peak': ub * B / (ceiling['user_side_ms'] * 1e-3) / (HBM_PEAK_GBS * 1e9),
                        'note': 'algorithmic row accesses only (no sorts, records, ids, biases, duplicate handling) on the '
                                'same tables in the same lane layout; the item side touches each distinct item once'})
        roof['ceiling'] = ceiling
    if trainer is not None:
        kern_ms = sum(prof[k][1] for k in ('sample', 'prep', 'user_pass', 'item_pass', 'exchange')) / K
        roof['xgmi'] = {'rows_per_step_per_gpu': xgmi_rows[0] / K,
                        'bytes_per_step_per_gpu_each_way': xgmi_rows[0] / K * (2 * (D + 1) * 4 + 4),  # id + row + gradient
                        'slices_per_minibatch': trainer.slices, 'minibatches_per_chunk': args.shard_chunk,
                        'kernel_ms_per_step': kern_ms,
                        'exchange_and_host_ms_per_step': elapsed / K * 1e3 - kern_ms}
        if xgmi_bytes is not None:
            # rank 0's MEASURED figures (bytes its trainer handed to the collectives for other ranks) beside the modelled one above
            roof['xgmi']['bytes_per_step_per_gpu_each_way_measured'] = xgmi_bytes[0] / K
            roof['xgmi']['payload_bytes_per_step_per_gpu_each_way_measured'] = xgmi_bytes[1] / K
            roof['xgmi']['measured_bytes_per_interaction'] = xgmi_bytes[0] / K / B
        # the exchange bound: every byte leaves through one of the (world - 1) direct xGMI links of
        # this GPU (point-to-point mesh, ~76.8 GB/s per link and direction)
        xb = roof['xgmi']['bytes_per_step_per_gpu_each_way']
        peak = max(world - 1, 1) * 76.8
        roof['xgmi'].update({'link_peak_GBs_each_way': peak,
                             'achieved_GBs_each_way_over_the_whole_step': xb / (elapsed / K) / 1e9,
                             'min_ms_per_step_at_link_peak': xb / (peak * 1e9) * 1e3 if world > 1 else 0.0})
        if world > 1:
            # the curve is to be read against the wire, not against N x one GPU: every interaction moves, per GPU and direction,
            # 2 lookups x (N-1)/N remote x (id 4 B + row (D+1)*4 B, then gradient (D+1)*4 B) over N-1 links of 76.8 GB/s
            roof['xgmi']['bound_%d_gpus' % world] = {
                'interactions_per_s_at_link_peak': world * B / (xb / (peak * 1e9)),
                'note': 'whole-job rate at which the exchange alone saturates every xGMI link (exact fp32 rows on the wire)'}
            if denominators is not None:
                roof['xgmi']['denominators_1_gpu'] = denominators
                for k in ('fused', 'sharded_world1'):
                    if isinstance(denominators.get(k), dict):
                        denominators[k]['scaling_factor_of_this_run'] = value / denominators[k]['interactions_per_s']
        if world == 1:
            # a MODEL, not a measurement: what this rank's measured kernel time and the wire allow at 8 GPUs.  Per
            # direction a GPU moves, for 7/8 of its 2B lookups, the id + the row (as requester in, as owner out)
            # + the gradient (the other way round) over 7 links of 76.8 GB/s.
            wire_ms = 2 * B * 7 / 8 * (2 * (D + 1) * 4 + 4) / (7 * 76.8e9) * 1e3
            roof['xgmi']['model_8_gpus'] = {
                'kind': 'model (no 8-GPU hardware measured)', 'kernel_ms_per_step_measured_here': kern_ms,
                'exchange_ms_per_step_at_link_peak': wire_ms,
                'ms_per_step_exchange_fully_hidden_or_hiding': max(kern_ms, wire_ms),
                'ms_per_step_nothing_overlapped': kern_ms + wire_ms,
                'interactions_per_s_range': [8 * B / ((kern_ms + wire_ms) * 1e-3), 8 * B / (max(kern_ms, wire_ms) * 1e-3)]}
    return value, roof
