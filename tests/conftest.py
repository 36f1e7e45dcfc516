import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE_DIR = os.path.join(ROOT, "oracle")
if ORACLE_DIR not in sys.path:
    sys.path.insert(0, ORACLE_DIR)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


NET_FORMS = {"auto": 0, "direct": 1, "winograd": 2, "team": 3, "wide": 4, "w43": 5}  # AZ_NET_FORM_* of include/azmcts.h


@pytest.fixture
def net_form():
    """Selects the kernel az_net_eval runs (by name, or by its AZ_NET_FORM_* number) for the
    duration of a test; the process-wide setting goes back to AUTO afterwards, pass or fail."""
    from alphazero_amd import engine as E

    lib = E.load_library()

    def choose(form):
        assert lib.az_net_set_form(NET_FORMS.get(form, form)) == 0

    yield choose
    lib.az_net_set_form(NET_FORMS["auto"])
