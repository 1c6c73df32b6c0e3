"""inverse_blocks at large N: block-diagonal of J Sigma = I (see fuzz_parity.py) and timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import _util
import cyclic_gps.cyclic_reduction as cr

for n, d, dtype in (((1 << 22) + 5, 4, torch.float64), (1 << 24, 4, torch.float64), ((1 << 23) + 77, 4, torch.float32),
                    ((1 << 22) + 3, 2, torch.float64), ((1 << 21) + 1, 5, torch.float32),
                    # the large-block passes: four lanes per row (8 x 8), Sigma of the tile in LDS (fp64 d = 6, 7)
                    (1 << 22, 8, torch.float32), ((1 << 21) + 129, 8, torch.float32), ((1 << 20) + 7, 8, torch.float64),
                    ((1 << 20) + 65, 6, torch.float64), ((1 << 19) + 3, 7, torch.float64)):
    Rs, Os, b, x_true, logdet = _util.conditioned_system(n, d, dtype=dtype, device="cuda", seed=n % 1000)
    dec = cr.decompose(Rs, Os)
    Sd, So = cr.inverse_blocks(dec)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        Sd, So = cr.inverse_blocks(dec)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    del dec
    worst = 0.0
    step = 1 << 20
    for a in range(0, n, step):          # chunked: keeps the temporaries small
        e = min(n, a + step)
        res = Rs[a:e] @ Sd[a:e]
        lo = max(a, 1)
        res[lo - a:] += Os[lo - 1:e - 1] @ So[lo - 1:e - 1].transpose(1, 2)
        hi = min(e, n - 1)
        res[:hi - a] += Os[a:hi].transpose(1, 2) @ So[a:hi]
        res -= torch.eye(d, dtype=dtype, device="cuda")
        worst = max(worst, float(res.abs().max()))
    print("n=%d d=%d %s: %.0f us, max |diag(J Sigma) - I| = %.2e" % (n, d, str(dtype)[6:], dt * 1e6, worst), flush=True)
    assert worst < (1e-9 if dtype == torch.float64 else 2e-3)
    del Rs, Os, Sd, So, res
    torch.cuda.empty_cache()
print("OK")
