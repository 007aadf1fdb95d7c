"""Ranking / accuracy metrics -- the callers on the far side of predict(), with the signatures and
semantics of spotlight/evaluation.py:9-244.

precision_recall_score / sequence_precision_recall_score take their top-k sets from score rows formed on the GPU a tile of
users at a time (torch.topk; a tie across a k boundary falls back to the reference's own per-user route, whose argsort
breaks it in its own way).  mrr_score and sequence_mrr_score have a fast path for this package's models: instead of one
predict() (a full pass over the item table) and one host scipy.stats.rankdata per user, the item
table streams through the matrix cores once per 64 held-out items (exact-fp32 MFMA: the same
scores as predict(), bit for bit) and every score is compared with its row's target score as it is
formed -- no score matrix (csrc/slk_eval.hip: slk_bilinear_rank / slk_poolnet_rank; models without
that route are scored a tile of rows at a time: slk_*_scores + slk_rank_targets).  Any other model
object (anything with the reference's predict()) takes the generic per-user route, which is also
what the tests compare the fast path with.
"""
import numpy as np
import scipy.stats as st

FLOAT_MAX = np.finfo(np.float32).max

_SCORE_BYTES = 256 << 20  # device memory for one tile of score rows
_STATS = {'topk_device': 0, 'topk_reference_route': 0}  # keys whose top-k sets came from the device / fell back (tests read it)


def _device_ranks(model, keys, num_items, exclude, targets):
    """ranks[k] = rankdata(-scores(keys[k]) with exclude[k] pushed last)[targets[k]], computed on
    the GPU.  keys: array of users (1-D) or sequences (2-D); exclude / targets: lists of index arrays."""
    import torch
    from spotlight_amd.factorization import implicit as host
    per_tile = max(1, _SCORE_BYTES // (4 * num_items))
    # score rows are model._num_items wide: every target / exclude index must address that row
    for lists in (targets, exclude):
        for x in lists:
            x = np.asarray(x)
            if x.size and (x.min() < 0 or x.max() >= num_items):  # numpy's error on predictions[indices]
                raise IndexError('index {} is out of bounds for axis 0 with size {}'.format(int(x.max()), num_items))
    fused = getattr(model, '_fused_ranks', None)
    if fused is not None:
        # one ROW per held-out item; per group the DISTINCT excluded items (predictions[ids] = FLOAT_MAX is idempotent)
        tl = [np.asarray(x).reshape(-1).astype(np.int64) for x in targets]
        el = [np.unique(np.asarray(x).reshape(-1).astype(np.int64)) for x in exclude]
        row_group = np.repeat(np.arange(len(keys), dtype=np.int64), [len(x) for x in tl])
        row_target = np.concatenate(tl) if tl else np.zeros(0, np.int64)
        if any(len(x) for x in el):
            exc_off = np.concatenate([[0], np.cumsum([len(x) for x in el])]).astype(np.int64)
            exc_items = np.concatenate(el)
        else:
            exc_off = exc_items = None
        ranks = fused(keys, row_group, row_target, exc_off, exc_items) if len(row_group) else np.zeros(0)
        if ranks is not None:
            bounds = np.concatenate([[0], np.cumsum([len(x) for x in tl])])
            return [ranks[bounds[k]:bounds[k + 1]] for k in range(len(keys))]
    out = []
    for lo in range(0, len(keys), per_tile):
        hi = min(lo + per_tile, len(keys))
        scores = model._batch_scores(keys[lo:hi])
        device = scores.device
        flat = lambda lists: (np.repeat(np.arange(hi - lo), [len(x) for x in lists]).astype(np.int64),
                              np.concatenate(lists).astype(np.int64) if lists else np.zeros(0, np.int64))
        er, ei = flat([np.asarray(x).reshape(-1) for x in exclude[lo:hi]])
        tr, ti = flat([np.asarray(x).reshape(-1) for x in targets[lo:hi]])
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        d_er, d_ei, d_tr, d_ti = dev(er), dev(ei), dev(tr), dev(ti)
        ranks = torch.empty(len(tr), dtype=torch.float64, device=device)
        host._engine_for(device).rank_targets(scores.data_ptr(), hi - lo, num_items, d_er.data_ptr(), d_ei.data_ptr(),
                                              len(er), d_tr.data_ptr(), d_ti.data_ptr(), len(tr), ranks.data_ptr(),
                                              host._stream_for(device))
        ranks = ranks.cpu().numpy()
        bounds = np.cumsum([0] + [len(np.asarray(x).reshape(-1)) for x in targets[lo:hi]])
        out.extend(ranks[bounds[k]:bounds[k + 1]] for k in range(hi - lo))
    return out


def _device_topk_sets(model, keys, num_items, exclude, ks):
    """For every key (user or sequence) and every k of `ks`: the SET of its k best items, train / preceding items pushed to
    the end -- what `predictions.argsort()[:k]` of the reference holds (evaluation.py:150-155, 212-217; precision and recall
    depend on the set only) -- from score rows formed on the GPU a tile of keys at a time (model._batch_scores: the bits of
    predict()) and torch.topk.  Returns a list of (order, ok): `order` the kmax best items, best first; `ok` False where a
    tie crosses one of the k boundaries (numpy's argsort breaks such ties in an unspecified way: the caller takes the
    reference's own route for that key)."""
    import torch
    kmax = int(max(ks))
    per_tile = max(1, _SCORE_BYTES // (4 * num_items))
    out = []
    for lo in range(0, len(keys), per_tile):
        hi = min(lo + per_tile, len(keys))
        scores = model._batch_scores(keys[lo:hi])
        ex = [np.asarray(x).reshape(-1).astype(np.int64) for x in exclude[lo:hi]]
        for x in ex:
            if x.size and (x.min() < 0 or x.max() >= num_items):  # numpy's error on predictions[indices]
                raise IndexError('index {} is out of bounds for axis 0 with size {}'.format(int(x.max()), num_items))
        if any(x.size for x in ex):
            er = torch.from_numpy(np.repeat(np.arange(hi - lo), [x.size for x in ex]).astype(np.int64)).to(scores.device)
            ei = torch.from_numpy(np.concatenate(ex)).to(scores.device)
            scores[er, ei] = -float(FLOAT_MAX)  # predictions = -scores; predictions[excluded] = FLOAT_MAX
        vals, idx = torch.topk(scores, kmax + 1, dim=1, largest=True, sorted=True)
        vals, idx = vals.cpu().numpy(), idx.cpu().numpy()
        for r in range(hi - lo):
            ok = all(vals[r, int(k) - 1] != vals[r, int(k)] for k in ks) and not np.isnan(vals[r]).any()
            _STATS['topk_device' if ok else 'topk_reference_route'] += 1
            out.append((idx[r, :kmax], ok))
    return out


def _has_fast_path(model):
    return getattr(model, '_batch_scores', None) is not None or getattr(model, '_fused_ranks', None) is not None


def mrr_score(model, test, train=None):
    """Mean reciprocal rank of each test user's held-out items among all items, train items pushed
    to the end of the ranking (evaluation.py:9-56).  One score per user with test interactions."""
    test = test.tocsr()
    train = train.tocsr() if train is not None else None
    users = np.array([u for u in range(test.shape[0]) if test.indptr[u + 1] > test.indptr[u]], dtype=np.int64)
    if _has_fast_path(model) and len(users):
        targets = [test[u].indices for u in users]
        exclude = [train[u].indices if train is not None else np.zeros(0, np.int64) for u in users]
        # the score rows' stride is the MODEL's item count (test may have been built with another num_items)
        ranks = _device_ranks(model, users, model._num_items, exclude, targets)
        return np.array([(1.0 / r).mean() for r in ranks])
    mrrs = []
    for user_id in users:
        predictions = -model.predict(int(user_id))
        if train is not None:
            predictions[train[user_id].indices] = FLOAT_MAX
        mrrs.append((1.0 / st.rankdata(predictions)[test[user_id].indices]).mean())
    return np.array(mrrs)


def sequence_mrr_score(model, test, exclude_preceding=False):
    """Reciprocal rank of the last element of every test sequence, predicted from the elements
    before it (evaluation.py:59-109)."""
    sequences = test.sequences[:, :-1]
    targets = test.sequences[:, -1:]
    if _has_fast_path(model) and len(sequences):
        exclude = [sequences[i] if exclude_preceding else np.zeros(0, np.int64) for i in range(len(sequences))]
        ranks = _device_ranks(model, sequences, model._num_items, exclude, [targets[i] for i in range(len(sequences))])
        return np.array([(1.0 / r).mean() for r in ranks])
    mrrs = []
    for i in range(len(sequences)):
        predictions = -model.predict(sequences[i])
        if exclude_preceding:
            predictions[sequences[i]] = FLOAT_MAX
        mrrs.append((1.0 / st.rankdata(predictions)[targets[i]]).mean())
    return np.array(mrrs)


def _get_precision_recall(predictions, targets, k):
    top = predictions[:k]
    hits = len(set(top).intersection(set(targets)))
    return float(hits) / len(top), float(hits) / len(targets)


def sequence_precision_recall_score(model, test, k=10, exclude_preceding=False):
    """Precision@k / recall@k of the last k elements of every test sequence, predicted from the
    elements before them (evaluation.py:112-162)."""
    sequences = test.sequences[:, :-k]
    targets = test.sequences[:, -k:]
    fast = None
    if getattr(model, '_batch_scores', None) is not None and len(sequences) and 0 < k < model._num_items:
        exclude = [sequences[i] if exclude_preceding else np.zeros(0, np.int64) for i in range(len(sequences))]
        fast = _device_topk_sets(model, sequences, model._num_items, exclude, [k])
    pairs = []
    for i in range(len(sequences)):
        if fast is not None and fast[i][1]:
            pairs.append(_get_precision_recall(fast[i][0], targets[i], k))
            continue
        predictions = -model.predict(sequences[i])
        if exclude_preceding:
            predictions[sequences[i]] = FLOAT_MAX
        pairs.append(_get_precision_recall(predictions.argsort()[:k], targets[i], k))
    pairs = np.array(pairs)
    return pairs[:, 0], pairs[:, 1]


def precision_recall_score(model, test, train=None, k=10):
    """Precision@k and recall@k per test user; k may be an array, giving one column per value
    (evaluation.py:172-223)."""
    test = test.tocsr()
    train = train.tocsr() if train is not None else None
    ks = np.array([k]) if np.isscalar(k) else k
    precision, recall = [], []
    fast = {}
    if getattr(model, '_batch_scores', None) is not None and len(ks) and 0 < int(min(ks)) and int(max(ks)) < model._num_items:
        users = np.array([u for u in range(test.shape[0]) if test.indptr[u + 1] > test.indptr[u]], dtype=np.int64)
        if len(users):
            exclude = [train[u].indices if train is not None else np.zeros(0, np.int64) for u in users]
            fast = dict(zip(users.tolist(), _device_topk_sets(model, users, model._num_items, exclude, [int(x) for x in ks])))
    for user_id, row in enumerate(test):
        if not len(row.indices):
            continue
        order, ok = fast.get(user_id, (None, False))
        if ok:
            p, r = zip(*[_get_precision_recall(order, row.indices, x) for x in ks])
            precision.append(p)
            recall.append(r)
            continue
        predictions = -model.predict(user_id)
        if train is not None:
            predictions[train[user_id].indices] = FLOAT_MAX
        order = predictions.argsort()
        p, r = zip(*[_get_precision_recall(order, row.indices, x) for x in ks])
        precision.append(p)
        recall.append(r)
    return np.array(precision).squeeze(), np.array(recall).squeeze()


def rmse_score(model, test):
    """Root mean squared error of predict(user_ids, item_ids) against test.ratings (evaluation.py:226-244)."""
    predictions = model.predict(test.user_ids, test.item_ids)
    return np.sqrt(((test.ratings - predictions) ** 2).mean())
