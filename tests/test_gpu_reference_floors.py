"""The reference's own end-to-end accuracy floors for the sequence path (tests/reference_floors.py), on cuda:0."""
import pytest

import reference_floors as rf

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', rf.CASES, ids=[c[0] for c in rf.CASES])
def test_reference_mrr_floor(case):
    rf.check_case(case, use_cuda=True)
