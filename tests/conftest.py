import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- background fp32 oracle passes of the two real-size stage tests (tests/oracle_prefetch.py) ----
_PREFETCH_TESTS = {"enc": "test_prodshape_encoder_first_frame_batch_vs_oracle_9x720x1280",
                   "dec": "test_prodshape_decoder_first_latent_batch_vs_oracle_9x720x1280"}


def pytest_collection_finish(session):
    """Start the oracle workers as soon as it is known that their consumers will run (selected, on a box with a GPU).  Never on the CPU-only
    container (the consumers are `gpu` tests), not with DOVE_TEST_BF16_YARDSTICK=1 (that mode runs both oracles inline) and not with
    DOVE_TEST_ORACLE_PREFETCH=0."""
    if os.environ.get("DOVE_TEST_ORACLE_PREFETCH", "1") != "1" or os.environ.get("DOVE_TEST_BF16_YARDSTICK", "0") == "1":
        return
    if session.config.option.collectonly:
        return
    names = {it.name for it in session.items}
    stages = [s for s, n in _PREFETCH_TESTS.items() if n in names]
    if not stages:
        return
    import torch
    if not torch.cuda.is_available():
        return
    tests_dir = os.path.join(ROOT, "tests")
    if tests_dir not in sys.path:
        sys.path.insert(0, tests_dir)
    import oracle_prefetch
    from test_parity_gpu import CONV_OUT_SCALE
    oracle_prefetch.start(stages, CONV_OUT_SCALE)


def pytest_sessionfinish(session, exitstatus):
    mod = sys.modules.get("oracle_prefetch")
    if mod is not None:
        mod.stop()
