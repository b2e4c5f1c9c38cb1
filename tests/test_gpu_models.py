"""Model-level parity on the MI355X against the reference's golden vectors (tests/gpu_model_check.py)."""
import pytest

import gpu_model_check as mc


@pytest.mark.gpu
@pytest.mark.parametrize("case", mc.CASES, ids=[c.__name__ for c in mc.CASES])
def test_model_case(case):
    results = case()
    bad = [msg for ok, msg in results if not ok]
    assert not bad, "\n".join(bad)
