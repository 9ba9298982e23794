"""GPU: the reference's only runnable example (notebooks/demo.ipynb) end to end on the facade."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_notebook_demo_vem_improves_elbo(capsys):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import demo
    e0, e1, acc = demo.main(seed=0, vem_iters=5, verbose=False)
    # the notebook's stored traces on its own (unseeded) data: first VE step -1980.97 (M=8); -2226.8 -> -1207.2 over 5
    # iterations (M=6).  Same model, seeded data here: about -1871 -> -1138.
    assert e1 > e0 + 500.0 and -1500.0 < e1 < -900.0
    assert acc > 0.75
    out = capsys.readouterr().out
    assert "VE-step" in out and "VM-step" in out            # util.vem_algorithm reports every half-step, like the reference
    trace = [float(l.split("ELBO = [")[1].rstrip("]\n")) for l in out.splitlines() if "ELBO = [" in l]
    assert len(trace) == 10 and all(b >= a - 1e-6 * abs(a) for a, b in zip(trace, trace[1:]))   # monotone VEM
