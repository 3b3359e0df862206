"""GPU: every pooling head, forward + backward, captured in a hipGraph (torch.cuda.graph) and replayed - INTEGRATION.md
says the C-ABI entry points are capturable (no host synchronisation, no allocation, the Newton-Schulz fork / join onto
the helper queues expressed with events); this is where that is checked.  tools/graph_rows.py does the work in a child
process (a failed capture can leave the stream in capture mode: it must not take the test session with it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_head_replays_bit_identically_from_a_graph():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'graph_rows.py'), '--quick'], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('[')][-1])
    assert len(rows) == 7                        # BCNN, CBCNN, MPN, AP-CNN, OSME, CIN at 7x7 and at 14x14 maps
    for r in rows:
        print(r)
        assert 'error' not in r, r
        assert r['bit_identical_to_eager'], r
        assert r['us_graph_replay'] > 0
