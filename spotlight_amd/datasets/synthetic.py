"""Synthetic sequential interactions -- same generator (and RandomState consumption) as
spotlight/datasets/synthetic.py:1-135, the data source of the reference's sequence-model tests:
items follow an n-th order Markov chain whose doubly-stochastic transition matrix is drawn from a
Dirichlet distribution.  Host-side numpy only.  (The download-based datasets of the reference need
network access and h5py and are out of scope.)
"""
import numpy as np

from spotlight_amd.interactions import Interactions


def _build_transition_matrix(num_items, concentration_parameter, random_state, atol=0.001):
    """Dirichlet rows, then Sinkhorn-style column/row normalisation until both marginals are
    within atol of 1 (at most 100 sweeps) (:12-34)."""
    matrix = random_state.dirichlet(np.repeat(concentration_parameter, num_items), num_items)
    for _ in range(100):
        if (np.all(np.abs(1.0 - matrix.sum(axis=0)) < atol)
                and np.all(np.abs(1.0 - matrix.sum(axis=1)) < atol)):
            break
        matrix /= matrix.sum(axis=0)
        matrix /= matrix.sum(1)[:, np.newaxis]
    return matrix


def _generate_sequences(num_steps, transition_matrix, order, random_state):
    """One chain of num_steps states: the next state is drawn from the mean of the cumulative
    transition rows of the last `order` states (:37-62)."""
    num_states = transition_matrix.shape[0]
    cumulative = np.cumsum(transition_matrix, axis=1)
    uniforms = random_state.rand(num_steps)
    state = random_state.randint(num_states, size=order, dtype=np.int64)
    out = np.empty(num_steps, dtype=np.int32)
    for k, u in enumerate(uniforms):
        nxt = min(num_states - 1, np.searchsorted(cumulative[state].mean(axis=0), u))
        state[:-1] = state[1:]
        state[-1] = nxt
        out[k] = nxt
    return out


def generate_sequential(num_users=100, num_items=1000, num_interactions=10000, concentration_parameter=0.1,
                        order=3, random_state=None):
    """Interactions(user_ids sorted, item_ids in [1, num_items), timestamps = arange) (:65-135)."""
    if random_state is None:
        random_state = np.random.RandomState()
    transition = _build_transition_matrix(num_items - 1, concentration_parameter, random_state)
    user_ids = np.sort(random_state.randint(0, num_users, num_interactions, dtype=np.int32))
    item_ids = _generate_sequences(num_interactions, transition, order, random_state) + 1
    timestamps = np.arange(len(user_ids), dtype=np.int32)
    ratings = np.ones(len(user_ids), dtype=np.float32)
    return Interactions(user_ids, item_ids, ratings=ratings, timestamps=timestamps, num_users=num_users,
                        num_items=num_items)
