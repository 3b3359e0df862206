"""GPU parity of the opt-in variants that were written after round 1's GPU budget was spent: validated against the
oracle on the CPU emulation tier (tests/test_emu_parity.py collects these cases too), first run on a real MI355X
by the round-end `-m gpu` pass.  The file sorts after the validated suites on purpose.  Each variant is OFF by default
in the library (environment switch named in the test) until it has a measured number.
"""
import numpy as np
import pytest
import torch

import hawkeye_oracle as O

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def rel(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = (b.detach() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b))).double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope='module')
def F():
    import hawkeye_amd.functional as F_
    from hawkeye_amd import _lib
    lib = _lib.load()
    assert b'gfx950' in lib.hk_version()
    return F_


@pytest.mark.parametrize('b,d,itn', [(3, 70, 5), (2, 128, 5), (9, 64, 3), (2, 33, 2)])
def test_ns_symmetric_tile_mode(F, b, d, itn, monkeypatch):
    """HK_NS_SYM=1: upper-triangle tiles only + mirrored stores for the symmetric products of the Newton-Schulz chain,
    and Z Y taken as the transpose of Y Z in the backward (38 -> 34 launches).  Same tolerances as the full products."""
    x = torch.relu(torch.randn(b, d, 6, 7, generator=torch.Generator().manual_seed(d))) + 0.01
    xo = x.clone().requires_grad_(True)
    yo = O.triuvec(O.sqrtm(O.covpool(xo), itn))
    wt = torch.randn(yo.shape, generator=torch.Generator().manual_seed(1))
    (yo * wt).sum().backward()
    res = []
    for flag in ('0', '1'):
        monkeypatch.setenv('HK_NS_SYM', flag)
        xg = x.clone().to(DEV).requires_grad_(True)
        cov = F.covpool(xg)
        s = F.sqrtm(cov, itn)
        yg = F.triuvec(s)
        (yg * wt.to(DEV)).sum().backward()
        assert rel(yg, yo) < 1e-5 and rel(xg.grad, xo.grad) < 1e-4
        if flag == '1':
            assert rel(s, s.transpose(1, 2)) < 1e-6            # off-diagonal tiles mirrored, diagonal tiles computed
        res.append((yg.detach(), xg.grad))
    assert rel(res[1][0], res[0][0]) < 5e-6 and rel(res[1][1], res[0][1]) < 5e-5
