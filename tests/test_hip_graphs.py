"""The launch-bound end of the path under HIP graphs: a captured mahal_and_det / decompose + solve
replays with new operand values written into the captured buffers and gives the eager results
(the library keeps no host-side state between calls besides one-time kernel attribute caches)."""
import numpy as np
import pytest
import torch

import _util
import cyclic_gps.cyclic_reduction as cr

pytestmark = pytest.mark.gpu


@pytest.fixture
def no_host_checks():
    old = cr.CHECK_POSITIVE_DEFINITE
    cr.CHECK_POSITIVE_DEFINITE = False          # the definiteness check is a device->host sync
    yield
    cr.CHECK_POSITIVE_DEFINITE = old


@pytest.mark.parametrize("n,d", [(1024, 2), (70001, 4)])
def test_graph_replay_matches_eager(n, d, no_host_checks):
    A = [t.cuda() for t in _util.conditioned_system(n, d, seed=1)[:3]]
    B = [t.cuda() for t in _util.conditioned_system(n, d, seed=2)[:3]]
    R, O, v = (t.clone() for t in A)

    def work():
        m, ld = cr.mahal_and_det(R, O, v)
        x = cr.solve(cr.decompose(R, O), v)
        return m, ld, x

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                # warm-up off the default stream, as capture requires
        for _ in range(2):
            work()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = work()
    for src in (B, A):
        for dst, s in zip((R, O, v), src):
            dst.copy_(s)
        g.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in outs]
        want = work()
        for a, b in zip(got, want):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-13, atol=1e-13)
