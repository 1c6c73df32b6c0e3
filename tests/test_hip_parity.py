"""GPU parity tests: the HIP path (through the C ABI) against
  * golden vectors recorded from the unmodified reference,
  * the CPU oracle on seeded inputs (ragged sizes, every d, both dtypes),
  * closed-form properties at BASELINE.json's full sizes.
Tolerances: north_star asks for 1e-5 relative (fp64 log-det / solve) and 1e-4
(fp32 posterior mean); the fp64 checks here are far tighter (1e-9 .. 1e-11).
"""
import ctypes

import numpy as np
import pytest
import torch

import _util
from oracle import cr_oracle as O
import cyclic_gps.cyclic_reduction as cr

pytestmark = pytest.mark.gpu

CASES = _util.golden_cr_cases()
IDS = ["d%d_n%d" % (d, n) for d, n, _ in CASES]
T64 = dict(rtol=1e-9, atol=1e-11)


def _np(t):
    return t.detach().cpu().numpy()


def _catnp(lst, d):
    return np.concatenate([_np(t).reshape(-1, d, d) for t in lst], axis=0) if lst else np.zeros((0, d, d))


@pytest.mark.parametrize("d,n,path", CASES, ids=IDS)
@pytest.mark.parametrize("where", ["gpu", "cpu"])
def test_against_reference_golden(d, n, path, where):
    """Everything the reference computes for this system, recomputed by the HIP path.
    where="cpu" feeds CPU tensors like the reference's own tests do."""
    g = np.load(path)
    dev = "cuda" if where == "gpu" else "cpu"
    Rs, Os, v = (torch.from_numpy(g[k]).to(dev) for k in ("Rs", "Os", "v"))
    dec = cr.decompose(Rs, Os)
    ms, Ds, Fs, Gs = dec
    assert ms.device.type == "cpu" and ms.dtype == torch.int64
    assert np.array(ms).tolist() == g["ms"].tolist()
    assert all(t.device.type == dev for t in Ds)
    np.testing.assert_allclose(_catnp(Ds, d), g["Dcat"], **T64)
    np.testing.assert_allclose(_catnp(Fs, d), g["Fcat"].reshape(-1, d, d), **T64)
    np.testing.assert_allclose(_catnp(Gs, d), g["Gcat"].reshape(-1, d, d), **T64)
    assert all(float(torch.triu(D, 1).abs().max()) == 0.0 for D in Ds)   # true lower-triangular factors
    half = cr.halfsolve(dec, v)
    assert [h.shape[0] for h in half] == [(m + 1) // 2 for m in ms.tolist()]
    np.testing.assert_allclose(np.concatenate([_np(h) for h in half]), g["half"], **T64)
    np.testing.assert_allclose(_np(cr.solve(dec, v)), g["solve"], **T64)
    np.testing.assert_allclose(float(cr.mahal(dec, v)), float(g["mahal"]), rtol=1e-10)
    np.testing.assert_allclose(float(cr.det(dec)), float(g["det"]), rtol=1e-10, atol=1e-11)
    for levelwise in (False, True):
        m, ld = cr._mahal_and_det(Rs, Os, v, levelwise=levelwise)
        np.testing.assert_allclose([float(m), float(ld)], g["mad"], rtol=1e-10, atol=1e-11)
    vcrr = _util.split_levels(torch.from_numpy(g["vcrr"]).to(dev), [(m + 1) // 2 for m in ms.tolist()])
    np.testing.assert_allclose(_np(cr.backhalfsolve(dec, vcrr)), g["back"], **T64)
    Sd, So = cr.inverse_blocks(dec)
    np.testing.assert_allclose(_np(Sd), g["Sig_diag"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(_np(So), g["Sig_off"].reshape(-1, d, d), rtol=1e-8, atol=1e-10)
    if n > 1:
        (nn, D, F, G), (R1, O1) = cr.decompose_step(Rs, Os)
        assert nn == n
        for got, key in ((D, "step_D"), (F, "step_F"), (G, "step_G"), (R1, "step_R"), (O1, "step_O")):
            np.testing.assert_allclose(_np(got), g[key].reshape(got.shape), **T64)


def test_plain_tuple_factor_is_accepted():
    """A decomp built by someone else (plain tuple of per-level lists, e.g. the oracle's) works too."""
    g = np.load(CASES[-1][2])
    Rs, Os, v = (torch.from_numpy(g[k]) for k in ("Rs", "Os", "v"))
    dec = O.decompose(Rs, Os)
    np.testing.assert_allclose(_np(cr.solve(dec, v)), g["solve"], **T64)
    np.testing.assert_allclose(float(cr.det(dec)), float(g["det"]), rtol=1e-10)


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_ragged_sizes_against_oracle(d, dtype):
    """Every N from 1 to 70 plus sizes around the tile widths: even/odd counts at every level."""
    tol = dict(rtol=1e-9, atol=1e-10) if dtype == torch.float64 else dict(rtol=2e-4, atol=2e-4)
    sizes = list(range(1, 71)) + [127, 128, 129, 255, 256, 257, 511, 513, 1000, 1023, 1025, 2049, 4097, 5000]
    if d in (1, 4, 5, 6, 7, 8):   # tile-boundary cases of the multi-pass kernels: one-row ragged tiles at two passes
        sizes += [65537, 66049, 131073]   # (d = 8, fp64 d = 6: the several-lanes-per-row streaming kernels)
    if d >= 6:
        sizes += [2047, 2049, 4095, 4096, 8191, 8193, 12289, 16385]
    if d == 4:              # every rows-per-lane regime of the fused solve + log-det (1, 4, 8 rows per lane; 16: full-size tests)
        sizes += [65536, 262143, 262145, 300001, 524287]
    for n in sizes:
        Rs, Os, b, x_true, logdet = _util.conditioned_system(n, d, seed=100 + n)
        ref_m, ref_ld = O.mahal_and_det(Rs, Os, b)
        R, Oo, v = Rs.to(dtype).cuda(), Os.to(dtype).cuda(), b.to(dtype).cuda()
        for levelwise in (False, True):
            m, ld = cr._mahal_and_det(R, Oo, v, levelwise=levelwise)
            np.testing.assert_allclose([float(m), float(ld)], [float(ref_m), float(ref_ld)],
                                       rtol=tol["rtol"], atol=tol["atol"], err_msg="n=%d" % n)
        dec = cr.decompose(R, Oo)
        x = cr.solve(dec, v)
        np.testing.assert_allclose(_np(x).astype(np.float64), _np(x_true), err_msg="n=%d" % n, **tol)
        np.testing.assert_allclose(float(cr.det(dec)), logdet, rtol=tol["rtol"], atol=tol["atol"])
        if n in (1, 2, 3, 33, 70, 257, 1000):
            ref = O.decompose(Rs, Os)
            for mine, theirs in zip(dec[1:], ref[1:]):
                for a, bb in zip(mine, theirs):
                    np.testing.assert_allclose(_np(a).astype(np.float64), _np(bb), err_msg="n=%d" % n, **tol)
            Sd, So = cr.inverse_blocks(dec)
            rSd, rSo = O.inverse_blocks(ref)
            np.testing.assert_allclose(_np(Sd).astype(np.float64), _np(rSd), **tol)
            np.testing.assert_allclose(_np(So).astype(np.float64), _np(rSo), **tol)


@pytest.mark.parametrize("d,dtype", [(1, torch.float64), (2, torch.float64), (3, torch.float64), (4, torch.float64),
                                     (4, torch.float32), (5, torch.float32), (5, torch.float64), (6, torch.float64),
                                     (7, torch.float64), (8, torch.float32), (8, torch.float64)],
                         ids=["d1f64", "d2f64", "d3f64", "d4f64", "d4f32", "d5f32", "d5f64", "d6f64", "d7f64", "d8f32", "d8f64"])
def test_inverse_blocks_fused_passes(d, dtype):
    """inverse_blocks at sizes where the three-levels-per-launch passes run: one, two and three fused
    passes, ragged one-row tiles, rows of a level that are not a multiple of 8.  Blocks up to 200 bytes
    keep a tile's Sigma in registers, larger ones (fp32 d = 8, fp64 d = 6..8) in LDS."""
    tol = dict(rtol=1e-9, atol=1e-10) if dtype == torch.float64 else dict(rtol=3e-4, atol=3e-4)
    for n in (1024, 1025, 1031, 4097, 8191, 8200, 65537, 66049, 100003, 524289 if d == 4 else 70001):
        Rs, Os, _, _, _ = _util.conditioned_system(n, d, seed=7 + n)
        rSd, rSo = O.inverse_blocks(O.decompose(Rs, Os))
        dec = cr.decompose(Rs.to(dtype).cuda(), Os.to(dtype).cuda())
        Sd, So = cr.inverse_blocks(dec)
        np.testing.assert_allclose(_np(Sd).astype(np.float64), _np(rSd), err_msg="n=%d" % n, **tol)
        np.testing.assert_allclose(_np(So).astype(np.float64), _np(rSo), err_msg="n=%d" % n, **tol)


def test_known_answers_float32():
    """The reference's closed-form test (tests/test_cyclic_reduction.py:243-291), float32 like there."""
    Rs, Os, det_true, inv_true = _util.bab_blocks(10, 5, 2, dtype=torch.float32)
    x = torch.rand(10, 1, generator=torch.Generator().manual_seed(3))
    dec = cr.decompose(Rs, Os)
    assert np.allclose(np.log(det_true), float(cr.det(dec)))
    m, ld = cr.mahal_and_det(Rs, Os, x=x)
    assert np.allclose(np.log(det_true), float(ld))
    assert np.allclose(float(x[:, 0].double() @ torch.from_numpy(inv_true) @ x[:, 0].double()), float(m))
    Sd, So = cr.inverse_blocks(dec)
    assert np.allclose(_np(Sd).ravel(), np.diag(inv_true))
    assert np.allclose(_np(So).ravel(), np.diag(inv_true, -1))
    Rs, Os, logdet_true, inv_scale = _util.schur_gram_blocks(5, 1.0, 2.0, dtype=torch.float32)
    dec = cr.decompose(Rs, Os)
    assert np.allclose(logdet_true, float(cr.det(dec)))
    m, ld = cr.mahal_and_det(Rs, Os, x=x.reshape(5, 2))
    assert np.allclose(logdet_true, float(ld))
    assert np.allclose(inv_scale * float((x * x).sum()), float(m))


def test_not_positive_definite_raises():
    Rs, Os, b, _, _ = _util.conditioned_system(64, 3)
    Rs[17] = -Rs[17]
    with pytest.raises(cr.NotPSDError):
        cr.decompose(Rs.cuda(), Os.cuda())
    with pytest.raises(cr.NotPSDError):
        cr.mahal_and_det(Rs.cuda(), Os.cuda(), b.cuda())
    Rs[17] = float("nan")                         # the reference's psd_safe_cholesky raises NanError for NaN operands
    with pytest.raises(cr.NanError):
        cr.mahal_and_det(Rs.cuda(), Os.cuda(), b.cuda())
    with pytest.raises(cr.NanError):
        cr.decompose(Rs.cuda(), Os.cuda())
    # without the host-side check (no device sync) the results are poisoned, never plausible:
    # two negative pivots multiply to a positive "determinant"
    Rs, Os, b, _, _ = _util.conditioned_system(5000, 4)
    Rs[100] = -Rs[100]
    Rs[3000] = -Rs[3000]
    cr.CHECK_POSITIVE_DEFINITE = False
    try:
        m, ld = cr.mahal_and_det(Rs.cuda(), Os.cuda(), b.cuda())
    finally:
        cr.CHECK_POSITIVE_DEFINITE = True
    assert torch.isnan(m) and torch.isnan(ld)


@pytest.mark.parametrize("N,d,dtype,rtol", [
    (2 ** 20, 4, torch.float64, 1e-10),     # BASELINE config 2
    (2 ** 22, 8, torch.float32, 2e-5),      # BASELINE config 3
    (2 ** 20 + 12345, 4, torch.float64, 1e-10),
    (2 ** 21 + 1, 5, torch.float64, 1e-10),
    (2 ** 24, 4, torch.float64, 1e-10),     # BASELINE config 4 as ONE system on one GPU (one launch of long chunks; CGPS_S1_LONG=0: three launches)
    # the one-launch form (record stages inside the stage-1 kernel) for the other block sizes: with the
    # coherent-load hand-off (even d*d in 16-byte vectors) and with the acquire hand-off (the rest)
    (2 ** 19 + 12345, 1, torch.float64, 1e-10),
    (2 ** 20, 2, torch.float64, 1e-10),
    (2 ** 20 - 5, 3, torch.float64, 1e-10),
    (2 ** 19 + 77, 2, torch.float32, 2e-5),
    (2 ** 20, 3, torch.float32, 2e-5),
    (2 ** 20 - 5, 6, torch.float32, 2e-5),
    (2 ** 20, 7, torch.float32, 2e-5),
    (2 ** 20 + 3, 6, torch.float64, 1e-10),
    (2 ** 20, 8, torch.float64, 1e-10),
    # fp64 d = 7: decompose = levels of more than 2^17 rows one launch each, then 64-row tile passes (cgps_decompose.hip)
    (2 ** 18 + 2 ** 17 + 5, 7, torch.float64, 1e-10),
], ids=["c2_N2^20_d4_f64", "c3_N2^22_d8_f32", "ragged_d4_f64", "ragged_d5_f64", "c4_N2^24_d4_f64", "d1_f64", "d2_f64",
        "d3_f64", "d2_f32", "d3_f32", "d6_f32", "d7_f32", "d6_f64", "d8_f64", "d7_f64"])
def test_full_size_closed_form(N, d, dtype, rtol):
    """Size-independent properties at the benchmark sizes: J = L L^T with L block
    bidiagonal, so log|J| and the planted solution x_true are known in closed form."""
    Rs, Os, b, x_true, logdet = _util.conditioned_system(N, d, dtype=dtype, device="cuda")
    mahal_true = float((x_true.double() * b.double()).sum())
    for levelwise in (False, True):
        m, ld = cr._mahal_and_det(Rs, Os, b, levelwise=levelwise)
        assert abs(float(ld) - logdet) <= rtol * abs(logdet)
        assert abs(float(m) - mahal_true) <= max(rtol, 1e-9) * 10 * abs(mahal_true)
    dec = cr.decompose(Rs, Os)
    x = cr.solve(dec, b)
    err = float((x.double() - x_true.double()).abs().max())
    assert err <= (1e-9 if dtype == torch.float64 else 1e-3), err
    assert abs(float(cr.det(dec)) - logdet) <= rtol * abs(logdet)
    # the solve is linear: J^-1 (2b) == 2 J^-1 b
    x2 = cr.solve(dec, 2 * b)
    assert float((x2 - 2 * x).abs().max()) <= (1e-9 if dtype == torch.float64 else 1e-3)


@pytest.mark.parametrize("N,d,dtype", [(2 ** 22, 8, torch.float32), (2 ** 20 + 7, 8, torch.float64),
                                       (2 ** 20 + 65, 6, torch.float64), (2 ** 19 + 3, 7, torch.float64),
                                       (2 ** 21 + 5, 4, torch.float64), (2 ** 21 + 1, 5, torch.float32)],
                         ids=["c3_d8_f32_quads", "d8_f64_quads", "d6_f64_lds", "d7_f64_lds", "d4_f64_registers", "d5_f32_registers"])
def test_inverse_blocks_full_size_identity(N, d, dtype):
    """inverse_blocks at benchmark sizes (every form of the three-levels-per-launch pass: registers, LDS,
    four lanes per row): the block diagonal of J Sigma is the identity -- it involves every diagonal and
    every off-diagonal block of the result."""
    Rs, Os, _, _, _ = _util.conditioned_system(N, d, dtype=dtype, device="cuda", seed=N % 1000)
    Sd, So = cr.inverse_blocks(cr.decompose(Rs, Os))
    worst, step = 0.0, 1 << 19
    for a in range(0, N, step):                  # chunked: keeps the temporaries small
        e = min(N, a + step)
        res = Rs[a:e] @ Sd[a:e]
        lo = max(a, 1)
        res[lo - a:] += Os[lo - 1:e - 1] @ So[lo - 1:e - 1].transpose(1, 2)
        hi = min(e, N - 1)
        res[:hi - a] += Os[a:hi].transpose(1, 2) @ So[a:hi]
        res -= torch.eye(d, dtype=dtype, device="cuda")
        worst = max(worst, float(res.abs().max()))
    assert worst < (1e-9 if dtype == torch.float64 else 2e-4), worst


@pytest.mark.parametrize("n,d", [(2 ** 20, 4), (2 ** 20, 3), (2 ** 21, 4), (2 ** 20, 8)],
                         ids=["coherent_loads_d4", "acquire_d3", "acquire_two_per_cu_d4", "four_lanes_per_row_d8"])
def test_folded_final_stage_never_reads_stale_records(n, d):
    """At 2^19 < N <= 2^20 rows (d = 4, fp64) the whole reduction is ONE launch: stage-1 workgroups hand
    their records to the workgroup that arrives last, inside the launch (csrc/cgps_tile.h, fold_final).
    A stale read there would go unnoticed when the same system is solved again and again (the stale
    record equals the fresh one), so alternate between DIFFERENT systems that share one workspace --
    caches warm with the other system's records -- and check every single result, bit for bit,
    against the first result of that system and against the one-launch-per-level form."""
    systems = []
    for seed in (11, 12, 13):
        Rs, Os, b, x_true, logdet = _util.conditioned_system(n - 4096 * (seed - 11), d, seed=seed, device="cuda")
        ref = cr._mahal_and_det(Rs, Os, b, levelwise=True)
        systems.append((Rs, Os, b, float(ref[0]), float(ref[1]), logdet))
    first = [None] * len(systems)
    cr.CHECK_POSITIVE_DEFINITE = False
    try:
        outs = []
        for it in range(150 if d <= 4 else 45):
            k = (it * 7 + it // 5) % len(systems)
            Rs, Os, b, m_ref, ld_ref, logdet = systems[k]
            outs.append((k, cr.mahal_and_det(Rs, Os, b)))
        torch.cuda.synchronize()
    finally:
        cr.CHECK_POSITIVE_DEFINITE = True
    for k, (m, ld) in outs:
        Rs, Os, b, m_ref, ld_ref, logdet = systems[k]
        m, ld = float(m), float(ld)
        if first[k] is None:
            first[k] = (m, ld)
            assert abs(ld - logdet) <= 1e-10 * abs(logdet)
            assert abs(ld - ld_ref) <= 1e-11 * abs(ld_ref) and abs(m - m_ref) <= 1e-9 * abs(m_ref)
        assert (m, ld) == first[k], (k, m, ld, first[k])


@pytest.mark.parametrize("n,d", [(2 ** 20, 4), (2 ** 21, 4), (2 ** 19 + 77, 5)])
def test_in_launch_hand_off_under_uneven_load(n, d):
    """The same alternation of DIFFERENT systems through one workspace while a second stream keeps the memory system
    busy with unrelated traffic in bursts (copies of 512 MB, uneven load: the hand-off forms that drop the acquire are
    the ones that go stale only under load -- micro-architecture guide, inter-workgroup visibility), every result
    checked bit for bit against what the same system gave on the idle chip.  Run ONCE: a check, not a stress loop.
    Also: cgps_reset_counters() between two calls leaves the one-launch path working, and more workspaces than the
    library has counter slots keep it working (the slot of the least recently used workspace is reused)."""
    from cyclic_gps import _hip
    systems = []
    cr.CHECK_POSITIVE_DEFINITE = False
    try:
        for seed in (21, 22, 23):
            Rs, Os, b, x_true, logdet = _util.conditioned_system(n - 4096 * (seed - 21), d, seed=seed, device="cuda")
            m, ld = cr.mahal_and_det(Rs, Os, b)
            torch.cuda.synchronize()
            assert abs(float(ld) - logdet) <= 1e-10 * abs(logdet)
            systems.append((Rs, Os, b, float(m), float(ld)))
        side = torch.cuda.Stream()
        src = torch.empty(512 << 20, dtype=torch.uint8, device="cuda").random_(0, 255)
        dst = torch.empty_like(src)
        outs = []
        for it in range(60):
            if it % 3 != 2:                       # bursts, not a uniform background
                with torch.cuda.stream(side):
                    dst.copy_(src)
            k = (it * 7 + it // 5) % len(systems)
            outs.append((k, cr.mahal_and_det(*systems[k][:3])))
            if it == 30:
                _hip.check(_hip.lib().cgps_reset_counters(_hip.stream_ptr()))
        torch.cuda.synchronize()
        for k, (m, ld) in outs:
            assert (float(m), float(ld)) == systems[k][3:], (k, float(m), float(ld), systems[k][3:])
        # 1100 different workspaces (the library has 1024 slots): the results stay exact
        Rs, Os, b, m0, ld0 = systems[0]
        lib = _hip.lib()
        ws, nb = _hip.workspace(Rs.shape[0], d, Rs.dtype, _hip.OP_MAHAL_LOGDET, Rs.device)
        pool = torch.empty(nb + 256 * 1100, dtype=torch.uint8, device="cuda")
        out = torch.zeros(2, dtype=torch.float64, device="cuda")
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        for i in range(1100):
            wsp = ctypes.c_void_p(pool.data_ptr() + 256 * i)
            _hip.check(lib.cgps_mahal_logdet(_hip.ptr(Rs), _hip.ptr(Os), _hip.ptr(b), Rs.shape[0], d, _hip.dtype_code(Rs.dtype),
                                             wsp, nb, _hip.ptr(out), _hip.ptr(info), _hip.stream_ptr()))
            if i % 100 == 99 or i >= 1090:
                assert tuple(out.tolist()) == (m0, ld0), i
    finally:
        cr.CHECK_POSITIVE_DEFINITE = True


@pytest.mark.parametrize("d,dtype", [(1, torch.float64), (2, torch.float64), (3, torch.float32), (4, torch.float64),
                                     (5, torch.float64), (8, torch.float32)],
                         ids=["d1f64", "d2f64", "d3f32", "d4f64", "d5f64", "d8f32"])
@pytest.mark.parametrize("m", [2, 3, 4, 7, 8, 11])
def test_batched_right_hand_sides_against_oracle_columns(d, dtype, m):
    """solve / halfsolve / backhalfsolve / mahal with Y[N, d, m] (the reference's einsums carry a trailing
    "...", cyclic_reduction.py:52-57): every column equals the oracle's single-column result.  Sizes
    around the panel tiles (128 / 256 / 512 rows), several passes, ragged tiles."""
    tol = dict(rtol=1e-9, atol=1e-10) if dtype == torch.float64 else dict(rtol=3e-4, atol=3e-4)
    for n in (1, 2, 5, 127, 128, 129, 257, 1000, 4097, 70001):
        Rs, Os, b, _, _ = _util.conditioned_system(n, d, seed=31 + n + m)
        g = torch.Generator().manual_seed(n + 7 * m)
        Y = torch.randn(n, d, m, dtype=torch.float64, generator=g)
        ref_dec = O.decompose(Rs, Os)
        dec = cr.decompose(Rs.to(dtype).cuda(), Os.to(dtype).cuda())
        Yd = Y.to(dtype).cuda()
        X = cr.solve(dec, Yd)
        assert X.shape == (n, d, m)
        half = cr.halfsolve(dec, Yd)
        back = cr.backhalfsolve(dec, half)
        mah = float(cr.mahal(dec, Yd))
        ref_mah = 0.0
        for c in range(m):
            ref_half = O.halfsolve(ref_dec, Y[:, :, c])
            ref_x = O.backhalfsolve(ref_dec, ref_half)
            ref_mah += float(sum((h ** 2).sum() for h in ref_half))
            np.testing.assert_allclose(_np(X[:, :, c]).astype(np.float64), _np(ref_x), err_msg="n=%d c=%d" % (n, c), **tol)
            np.testing.assert_allclose(_np(back[:, :, c]).astype(np.float64), _np(ref_x), err_msg="n=%d c=%d" % (n, c), **tol)
            np.testing.assert_allclose(np.concatenate([_np(h[:, :, c]) for h in half]).astype(np.float64),
                                       np.concatenate([_np(h) for h in ref_half]), err_msg="n=%d c=%d" % (n, c), **tol)
        assert abs(mah - ref_mah) <= (1e-9 if dtype == torch.float64 else 1e-3) * max(1.0, abs(ref_mah))
    # trailing shape [N, d, 2, 3] behaves like m = 6
    Y4 = torch.randn(300, d, 2, 3, dtype=torch.float64, generator=g)
    Rs, Os, _, _, _ = _util.conditioned_system(300, d, seed=5)
    dec = cr.decompose(Rs.to(dtype).cuda(), Os.to(dtype).cuda())
    X4 = cr.solve(dec, Y4.to(dtype).cuda())
    assert X4.shape == Y4.shape
    np.testing.assert_allclose(_np(X4[:, :, 1, 2]).astype(np.float64), _np(O.solve(O.decompose(Rs, Os), Y4[:, :, 1, 2])), **tol)


def test_batched_solve_full_size_and_gradients():
    """Config-2 size: eight planted solutions at once; and autograd through solve with Y[N, d, m]."""
    n, d, m = 2 ** 20, 4, 8
    Rs, Os, b, x_true, _ = _util.conditioned_system(n, d, device="cuda")
    dec = cr.decompose(Rs, Os)
    scale = torch.arange(1, m + 1, dtype=torch.float64, device="cuda")
    X = cr.solve(dec, b[:, :, None] * scale)
    assert float((X - x_true[:, :, None] * scale).abs().max()) <= 1e-8
    # gradients: sum_c u_c^T J^-1 y_c against the same thing column by column
    n = 1000
    Rs, Os, _, _, _ = _util.conditioned_system(n, 3, device="cuda", seed=3)
    Y = torch.randn(n, 3, 4, dtype=torch.float64, device="cuda")
    U = torch.randn(n, 3, 4, dtype=torch.float64, device="cuda")
    grads = []
    for batched in (True, False):
        R, Oo, Yg = Rs.clone().requires_grad_(True), Os.clone().requires_grad_(True), Y.clone().requires_grad_(True)
        dec = cr.decompose(R, Oo)
        if batched:
            s = (U * cr.solve(dec, Yg)).sum()
        else:
            s = sum((U[:, :, c] * cr.solve(dec, Yg[:, :, c])).sum() for c in range(4))
        s.backward()
        grads.append((R.grad.clone(), Oo.grad.clone(), Yg.grad.clone()))
    for a, bb in zip(*grads):
        np.testing.assert_allclose(_np(a), _np(bb), rtol=1e-9, atol=1e-10)
