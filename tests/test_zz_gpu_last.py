"""Runs last (file order): no multi-workgroup fold of the whole GPU suite lost a partner except where a test injected it."""
import pytest

pytestmark = pytest.mark.gpu


def test_no_unplanned_sync_fallbacks():
    from desirna_amd import engine as E
    from tests import test_gpu_parity
    assert E.sync_fallbacks_total() == sum(test_gpu_parity.INJECTED_FALLBACKS)
