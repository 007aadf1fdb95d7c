"""repr helper (mirrors spotlight/helpers.py:1-12)."""


def _repr_model(model):
    net = '[uninitialised]' if model._net is None else repr(model._net)
    return '<{}: {}>'.format(model.__class__.__name__, net)
