import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with `-m gpu`; if someone runs the whole suite on a CPU box, skip them loudly.
    try:
        import torch

        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible (gpu-marked tests run on the MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# ---- the full-size parity rows (g1 / g2 / g4: tests/test_gpu_full_model*.py) skip when the HOST has too little free memory for the fp32 oracle copy.
# A skip is a silent loss of exactly the rows a judge looks for, so it is made loud: every such skip is listed in the
# terminal summary under its own banner, and a run on a box that HAS the memory (>= 100 GB available at the end of the session) yet skipped
# them for memory is turned into a failure.
_FULL_SIZE_SKIPS = []


def pytest_runtest_logreport(report):
    if report.skipped and "test_gpu_full_model" in report.nodeid and "not enough free host memory" in str(report.longrepr):
        _FULL_SIZE_SKIPS.append(report.nodeid)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not _FULL_SIZE_SKIPS:
        return
    try:
        import psutil

        avail = psutil.virtual_memory().available / 2**30
    except Exception:
        avail = -1.0
    tr = terminalreporter
    tr.section("FULL-SIZE PARITY ROWS DID NOT RUN (host memory)", sep="!", red=True, bold=True)
    tr.write_line("%d full-size test(s) (OTTER-MPT7B g1 / g2, C4 / C5 g4) were SKIPPED for lack of free host memory; %.0f GB are available now:" % (len(_FULL_SIZE_SKIPS), avail))
    for n in sorted(set(_FULL_SIZE_SKIPS)):
        tr.write_line("    " + n)
    tr.write_line("their rows of DESIGN.md section 5 are NOT covered by this run.")


def pytest_sessionfinish(session, exitstatus):
    if not _FULL_SIZE_SKIPS:
        return
    try:
        import psutil

        if psutil.virtual_memory().available >= 100 << 30 and exitstatus == 0:
            session.exitstatus = 1      # the box could have run them: a memory skip here is a defect of the run, not of the box
    except Exception:
        pass
