"""Constants and byte accounting shared by every workload of bench.py."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def algorithmic_bytes(dim, opt_state_words):
    """per interaction, fp32, int64 ids, no credit for cache hits/duplicates.
    user pass: user row param R+W + state R+W, two item rows read, user/pos/neg ids, user bias
    R+W(+state), two item-bias reads; item pass: two item rows written + state R+W, two item
    biases written + state R+W."""
    s = opt_state_words
    user_pass = (8 * dim + 8 * dim * s) + 2 * 4 * dim + 16 + (8 + 8 * s) + 8
    item_pass = 2 * (4 * dim + 8 * dim * s) + 2 * (4 + 8 * s)
    return user_pass, item_pass


def pmc_traffic(args, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes over this very
    workload (profiles/pmc_traffic.json, produced by scripts/pmc_run.sh + scripts/summarize_pmc.py:
    2 x FETCH_SIZE + WRITE_SIZE, separate passes; MI355X_MICROARCH.md "HBM").  PMC counters cannot
    be collected from inside the benchmark process, so this is null unless the committed
    measurement matches the workload being run."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    # (the committed passes ran bench.py's defaults: the user table doubled for the run -- a --no-user-pingpong run has no match)
    if getattr(args, 'no_user_pingpong', False) or getattr(args, 'user_pingpong_min_batch', 1 << 17) != 1 << 17:
        return None
    for r in [rec] + list(rec.get('others', [])):  # the headline configuration first, then the other measured ones
        cfg = r.get('config', {})
        if all(cfg.get(k) == getattr(args, k) for k in ('users', 'items', 'dim', 'batch', 'loss', 'opt')):
            return r.get('kernels', {}).get(kernel, {}).get('hbm_bytes_per_launch')
    return None
