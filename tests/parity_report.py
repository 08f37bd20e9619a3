"""Parity figures the GPU tests measure, collected so that ``tests/conftest.py::pytest_terminal_summary`` can print them LAST:
``pytest -q`` shows dots only, and the driver keeps the tail of the output (GPUTEST_rNN.json) - this is how the figures of
the full-size parity tests become driver-witnessed numbers instead of builder-run ones."""
LINES = []


def record(line):
    LINES.append(line)
    print(line)
