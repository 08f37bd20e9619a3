import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption('--run-slow', action='store_true', default=False, help='also run the tests marked slow (minutes of CPU oracle each)')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: minutes of CPU oracle per test; skipped unless --run-slow is given')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if not config.getoption('--run-slow'):
        skip_slow = pytest.mark.skip(reason='slow: pass --run-slow')
        for item in items:
            if 'slow' in item.keywords:
                item.add_marker(skip_slow)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.hookimpl(trylast=True)
def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The parity figures of the full-size tests, printed after everything else (tests/parity_report.py)."""
    try:
        import parity_report
    except Exception:
        return
    if not parity_report.LINES:
        return
    terminalreporter.section('parity figures measured in this run (max-rel = max|gpu - oracle| / max|oracle|)')
    for line in parity_report.LINES:
        terminalreporter.write_line(line)
