"""`spotlight` -- the reference's import paths, served by spotlight_amd.

The drop-in contract is "same import paths": code written against maciejkula/spotlight --

    from spotlight.interactions import Interactions
    from spotlight.factorization.implicit import ImplicitFactorizationModel
    from spotlight.sequence.implicit import ImplicitSequenceModel
    from spotlight.evaluation import mrr_score, sequence_mrr_score
    from spotlight.cross_validation import random_train_test_split
    from spotlight.datasets.synthetic import generate_sequential

-- runs unmodified on the MI355X engine when this repository root is on sys.path (or installed): every
`spotlight.<x>` module IS the module object `spotlight_amd.<x>` (an alias in sys.modules, not a second copy, so
classes, the per-device engine and isinstance checks are shared).  Modules the package does not provide
(`spotlight.datasets.movielens`, `.goodbooks`, `.amazon`: network downloads + HDF5) raise ModuleNotFoundError.
"""
import importlib
import importlib.abc
import importlib.util
import sys

import spotlight_amd as _impl

__version__ = 'v0.1.6'  # the reference release whose API this package mirrors (spotlight/__init__.py)
_PREFIX, _REAL = __name__ + '.', _impl.__name__ + '.'


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Resolves `spotlight.<x>` to the already-importable `spotlight_amd.<x>` and hands back that very module."""

    def find_spec(self, name, path=None, target=None):
        if not name.startswith(_PREFIX):
            return None
        real = _REAL + name[len(_PREFIX):]
        try:
            spec = importlib.util.find_spec(real)
        except (ImportError, ValueError):
            return None
        if spec is None:
            return None
        return importlib.util.spec_from_loader(name, self, is_package=spec.submodule_search_locations is not None)

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_PREFIX):])

    def exec_module(self, module):
        pass  # already executed under its spotlight_amd name


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
__path__ = []  # a package without files of its own: every submodule comes from the finder above
