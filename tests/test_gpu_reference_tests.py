"""`-m gpu`: the reference's OWN test modules that need no dataset download, run UNMODIFIED (the copies staged by
oracle/make_ref.sh under the git-ignored oracle/_ref/tests) against this package through its `spotlight.*` import
aliases: tests/sequence/test_sequence_implicit.py (23 ImplicitSequenceModel configurations with MRR floors) and
tests/test_layers.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTS = os.path.join(ROOT, 'oracle', '_ref', 'tests')


@pytest.mark.parametrize('rel', ['test_layers.py', os.path.join('sequence', 'test_sequence_implicit.py')])
def test_reference_test_module_passes_unmodified(rel):
    path = os.path.join(REF_TESTS, rel)
    if not os.path.exists(path):
        pytest.skip('oracle/_ref/tests not staged (sh oracle/make_ref.sh where /root/reference exists)')
    # the repository root first on the path: `import spotlight` is this package's alias, never oracle/_ref/spotlight
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-p', 'no:cacheprovider', '--rootdir', REF_TESTS, path],
                       cwd=REF_TESTS, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-4000:]
    assert 'spotlight_amd' in subprocess.run(
        [sys.executable, '-c', 'import spotlight.sequence.implicit as m; print(m.__name__)'], env=env, cwd=REF_TESTS,
        stdout=subprocess.PIPE).stdout.decode()
