"""Dataset shuffling and train/test splits -- host-side callers of the training path, with the
same signatures and the same RandomState consumption as spotlight/cross_validation.py:20-176
(so a given seed yields the reference's exact split).  Pure numpy: nothing here touches the GPU.
"""
import numpy as np
from sklearn.utils import murmurhash3_32

from spotlight_amd.interactions import Interactions


def _take(interactions, index):
    """Interactions restricted to `index` (an index array, boolean mask or slice); optional
    arrays stay None."""
    opt = lambda a: None if a is None else a[index]
    return Interactions(interactions.user_ids[index], interactions.item_ids[index],
                        ratings=opt(interactions.ratings), timestamps=opt(interactions.timestamps),
                        weights=opt(interactions.weights), num_users=interactions.num_users,
                        num_items=interactions.num_items)


def shuffle_interactions(interactions, random_state=None):
    """Random permutation of the interactions (cross_validation.py:20-55): one legacy
    RandomState.shuffle of arange(n)."""
    if random_state is None:
        random_state = np.random.RandomState()
    order = np.arange(len(interactions.user_ids))
    random_state.shuffle(order)
    return _take(interactions, order)


def random_train_test_split(interactions, test_percentage=0.2, random_state=None):
    """Shuffle, then cut at int((1 - test_percentage) * n) (cross_validation.py:58-111)."""
    shuffled = shuffle_interactions(interactions, random_state=random_state)
    cutoff = int((1.0 - test_percentage) * len(shuffled))
    return _take(shuffled, slice(None, cutoff)), _take(shuffled, slice(cutoff, None))


def user_based_train_test_split(interactions, test_percentage=0.2, random_state=None):
    """All interactions of a user land on the same side: a user is a test user when
    murmurhash3_32(user, seed, positive=True) % 100 / 100 < test_percentage, the seed being one
    int64 draw over the uint32 range (cross_validation.py:114-176)."""
    if random_state is None:
        random_state = np.random.RandomState()
    seed = random_state.randint(np.iinfo(np.uint32).min, np.iinfo(np.uint32).max, dtype=np.int64)
    bucket = murmurhash3_32(interactions.user_ids, seed=seed, positive=True) % 100 / 100.0
    in_test = bucket < test_percentage
    return _take(interactions, np.logical_not(in_test)), _take(interactions, in_test)
