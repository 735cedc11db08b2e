"""A short randomised differential run (tools/fuzz_parity.py: random indexes, shapes, ties, malformed rows, filters, beam
widths and kernel-routing options) through the C ABI against the oracle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_randomised_configurations_match_the_oracle(pkg, pyoracle):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(25.0, 1000)
