"""Kernel-level parity on the MI355X: each case of tests/gpu_kernel_check.py as a pytest test."""
import pytest

import gpu_kernel_check as kc


@pytest.mark.gpu
@pytest.mark.parametrize("case", kc.CASES, ids=[c.__name__ for c in kc.CASES])
def test_kernel_case(case):
    results = case()
    bad = [msg for ok, msg in results if not ok]
    assert not bad, "\n".join(bad)
