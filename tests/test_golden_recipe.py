"""Guards of the parity recipe: the fixture generators under oracle/ must run THE REFERENCE, never
this repository's `spotlight/` alias package onto the product, and -- wherever the live reference exists -- regenerating
the fixtures must reproduce tests/golden bit for bit.  CPU only; each generator runs in its own interpreter."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GENERATORS = ['make_golden', 'make_golden_seq', 'make_golden_bloom', 'make_golden_explicit', 'make_golden_encoders',
              'make_golden_host', 'make_golden_floors']
LIVE = os.path.isfile('/root/reference/spotlight/__init__.py')
STAGED = os.path.isfile(os.path.join(ROOT, 'oracle', '_ref', 'spotlight', '__init__.py'))

needs_reference = pytest.mark.skipif(not (LIVE or STAGED), reason='the reference exists neither live nor staged here')


def _run(code, cwd=ROOT, env=None):
    e = dict(os.environ)
    e.pop('SLK_GOLDEN_OUT', None)
    e.update(env or {})
    return subprocess.run([sys.executable, '-c', code], cwd=cwd, env=e, capture_output=True, text=True, timeout=900)


@needs_reference
@pytest.mark.parametrize('gen', GENERATORS)
def test_generator_imports_the_reference_not_the_product(gen):
    # importing the generator module (not running it) performs its imports; cwd = repository root, i.e. the alias
    # package `spotlight/` is first on sys.path -- exactly the situation in which the old recipe loaded the product
    code = ('import sys, importlib; sys.path.insert(0, %r)\n'
            'm = importlib.import_module("oracle.%s")\n'
            'import spotlight\n'
            'from oracle.reference_import import assert_is_reference\n'
            'assert_is_reference(spotlight)\n'
            'mods = [n for n in sys.modules if n == "spotlight_amd" or n.startswith("spotlight_amd.")]\n'
            'assert not mods, mods\n'
            'for n, mod in list(sys.modules.items()):\n'
            '    if n.startswith("spotlight.") and mod is not None: assert_is_reference(mod)\n'
            'print("REFERENCE", spotlight.__file__)\n') % (ROOT, gen)
    r = _run(code)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'REFERENCE' in r.stdout and 'spotlight_amd' not in r.stdout.split('REFERENCE')[-1]


def test_guard_rejects_the_alias_package():
    # the guard itself must fire on the product's alias (without needing a HIP device: the alias module is faked)
    code = ('import sys, types; sys.path.insert(0, %r)\n'
            'fake = types.ModuleType("spotlight"); fake.__file__ = %r\n'
            'sys.modules["spotlight"] = fake\n'
            'from oracle.reference_import import import_reference\n'
            'try:\n'
            '    import_reference()\n'
            'except (AssertionError, RuntimeError) as e:\n'
            '    print("REJECTED")\n'
            'else:\n'
            '    raise SystemExit("the alias package was accepted as the reference")\n') % (
                ROOT, os.path.join(ROOT, 'spotlight', '__init__.py'))
    r = _run(code)
    assert r.returncode == 0 and 'REJECTED' in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not LIVE, reason='regeneration needs the live reference (/root/reference: build container only)')
@pytest.mark.parametrize('gen,args', [('make_golden', []), ('make_golden_seq', []), ('make_golden_seq', ['bloom']),
                                      ('make_golden_bloom', []), ('make_golden_explicit', []),
                                      ('make_golden_encoders', []), ('make_golden_host', [])])
def test_regenerated_fixtures_are_bit_identical(gen, args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', gen + '.py'), '--check'] + args, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'bit-identical to tests/golden' in r.stdout
