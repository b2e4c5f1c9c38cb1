"""AAS-VC / Conformer specific kernel parity on the MI355X (tests/gpu_kernel_check_aas.py)."""
import pytest

import gpu_kernel_check_aas as kc


@pytest.mark.gpu
@pytest.mark.parametrize("case", kc.CASES, ids=[c.__name__ for c in kc.CASES])
def test_kernel_case(case):
    results = case()
    bad = [msg for ok, msg in results if not ok]
    assert not bad, "\n".join(bad)
