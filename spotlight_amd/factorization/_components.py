"""Id preprocessing for predict (mirrors spotlight/factorization/_components.py:8-25)."""
import numpy as np


def _predict_process_ids(user_ids, item_ids, num_items):
    """Returns (users int64 [1 or n], items int64 [n] or None for arange(num_items), n).

    `None` items mean "all items" and a scalar user is broadcast; both are resolved inside
    the kernel (slk_bilinear_predict) instead of materialising arange/expand tensors."""
    if np.isscalar(user_ids):
        user_ids = np.array(user_ids, dtype=np.int64)
    users = np.ascontiguousarray(np.asarray(user_ids).reshape(-1), dtype=np.int64)
    if item_ids is None:
        items, n = None, int(num_items)
    else:
        items = np.ascontiguousarray(np.asarray(item_ids).reshape(-1), dtype=np.int64)
        n = items.size
    if users.size != n:
        if users.size != 1:
            # the reference's `expand` raises for incompatible sizes
            raise RuntimeError('The expanded size of the tensor ({}) must match the existing size ({})'
                               .format(n, users.size))
    return users, items, n
