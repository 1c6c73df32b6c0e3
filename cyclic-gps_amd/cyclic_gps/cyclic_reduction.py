"""Drop-in for the reference's ``cyclic_gps/cyclic_reduction.py`` on MI355X.

Same names, argument names and return structures as the reference module
(reference file:line cited per function).  The path -- ``decompose_step``,
``decompose``, ``mahal_and_det``, ``halfsolve``, ``backhalfsolve``, ``solve``,
``det``, ``mahal``, ``inverse_blocks`` and their gradients -- runs in hand-written
HIP kernels for gfx950 reached through the C ABI of ``include/cgps.h``
(``_hip.py`` is the ctypes binding); PyTorch only owns the device memory and the
stream.  The six small banded-product helpers of the reference's surface
(``UU_T, Ux, U_Tx, SigU, UtV_diags, interleave``, :15-200) are NOT kernels of
this library: inside the path they are fused into the kernels above, and as
stand-alone names they are batched torch products on whatever device their
arguments live on.

* Inputs may live on the GPU (zero-copy) or on the CPU (they are staged to the
  current GPU and results are returned on the CPU, so the reference's own
  CPU-tensor tests run unchanged on a GPU box).  There is no CPU fallback: with
  no GPU or no ``libcgps.so`` the functions raise.
* ``decompose`` returns an indexable 4-tuple ``(ms, Ds, Fs, Gs)`` like the
  reference; the per-level tensors are views into three packed device buffers
  (``decomp.packed``) that the solve kernels use directly.
* ``ms`` stays a CPU int64 tensor (``np.array(ms)`` is used by callers,
  reference tests/test_cyclic_reduction.py:210).
"""
import ctypes

import numpy as np  # noqa: F401  (the reference's star-import exposes np and torch)
import torch

from . import _hip

JITTER = None  # reference :13 (passed to psd_safe_cholesky there; no jitter retry here)

try:  # raise the caller's own exception classes when gpytorch is installed
    from gpytorch.utils.errors import NanError, NotPSDError  # type: ignore
except Exception:  # pragma: no cover
    class NotPSDError(RuntimeError):
        """A diagonal block was not positive definite."""

    class NanError(RuntimeError):
        """The blocks handed in hold NaN (what psd_safe_cholesky raises in the reference, :227,306,429)."""

# Set to False to skip the device->host read of the `info` word after a
# factorisation (removes one stream synchronisation per call).  Nothing raises then:
# mahal_and_det returns NaN for both scalars on a block that is not positive definite, and a
# factor carries the NaNs of the failed square root.
CHECK_POSITIVE_DEFINITE = True


# ----------------------------------------------------------------------------
# plumbing
# ----------------------------------------------------------------------------
def _device():
    if not torch.cuda.is_available():
        raise _hip.CgpsError("no GPU visible: the cyclic-reduction kernels need an MI355X (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _stage(t, dtype=None):
    """Contiguous GPU tensor for t (copying from the CPU when needed)."""
    if t.device.type != "cuda":
        t = t.to(_device())
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _back(t, like):
    return t if like.device.type == "cuda" else t.to(like.device)


def _check_blocks(Rs, Os):
    if Rs.dim() != 3 or Os.dim() != 3:
        raise TypeError("Rs and Os must be [num_blocks, block_dim, block_dim] tensors")
    if Rs.shape[1] != Rs.shape[2] or (Os.shape[0] > 0 and tuple(Os.shape[1:]) != tuple(Rs.shape[1:])):
        raise TypeError("block_dim mismatch between Rs %s and Os %s" % (tuple(Rs.shape), tuple(Os.shape)))
    assert Rs.shape[0] == Os.shape[0] + 1          # reference :223
    _hip.dtype_code(Rs.dtype)


def _raise_if_not_pd(info, *blocks):
    """The reference's error behaviour at its psd_safe_cholesky calls: NanError when the operands hold NaN,
    NotPSDError otherwise (its jitter-and-retry in between is not reproduced).  `blocks` are looked at only
    on the failure path."""
    if CHECK_POSITIVE_DEFINITE:
        bad = int(info.item())
        if bad != 0:
            for t in blocks:
                nan = int(torch.isnan(t).sum().item()) if t is not None and t.numel() else 0
                if nan:
                    raise NanError("%d of %d elements of the %s tensor are NaN." % (nan, t.numel(), tuple(t.shape)))
            raise NotPSDError("block row %d is not positive definite" % (bad - 1))


class CRDecomp(tuple):
    """(ms, Ds, Fs, Gs) exactly as the reference returns it (:309), plus the packed
    device buffers the per-level lists are views of."""

    def __new__(cls, ms, Ds, Fs, Gs, packed=None, like=None):
        self = super().__new__(cls, (ms, Ds, Fs, Gs))
        self.packed = packed      # (Dp, Fp, Gp) on the GPU, or None
        self.like = like          # tensor whose device the results should follow
        self.inputs = None        # (Rs, Os) when they require grad (set by decompose)
        return self


def _views(Dp, Fp, Gp, N):
    ms, offD, offF, offG = _hip.level_layout(N)
    L = len(ms)

    def cut(P, off, count):                      # one split call instead of one slice per level
        sizes = [off[i + 1] - off[i] for i in range(count)]
        return list(P[:off[count]].split(sizes)) if count else []
    return torch.tensor(ms, dtype=torch.int64), cut(Dp, offD, L), cut(Fp, offF, L - 1), cut(Gp, offG, L - 1)


def _packed(decomp):
    """(Dp, Fp, Gp, N, d, like) for any decomp: ours, or a plain tuple of lists."""
    if isinstance(decomp, CRDecomp) and decomp.packed is not None:
        Dp, Fp, Gp = decomp.packed
        return Dp, Fp, Gp, int(decomp[0][0]), Dp.shape[-1], decomp.like
    ms, Ds, Fs, Gs = decomp
    like = Ds[0]
    d = like.shape[-1]
    N = int(ms[0])
    dev = _device()
    Dp = torch.cat([_stage(t) for t in Ds], dim=0)
    pad = torch.zeros((1, d, d), dtype=like.dtype, device=dev)
    Fp = torch.cat([_stage(t) for t in Fs] + [pad], dim=0)
    Gp = torch.cat([_stage(t) for t in Gs] + [pad], dim=0)
    return Dp, Fp, Gp, N, d, like


# ----------------------------------------------------------------------------
# factorisation
# ----------------------------------------------------------------------------
def decompose_step(Rs, Os):
    """One reduction level -> (n, D, F, G), (Rs', Os')   (reference :203-259)."""
    _check_blocks(Rs, Os)
    n, d = Rs.shape[0], Rs.shape[1]
    if n < 2:
        raise ValueError("decompose_step needs at least two diagonal blocks")
    R, O = _stage(Rs), _stage(Os)
    dev, dt = R.device, R.dtype
    new = lambda k: torch.empty((k, d, d), dtype=dt, device=dev)  # noqa: E731
    D, F, G, Rn, On = new((n + 1) // 2), new(n // 2), new((n - 1) // 2), new(n // 2), new(n // 2 - 1)
    info = torch.empty(1, dtype=torch.int32, device=dev)
    _hip.check(_hip.lib().cgps_decompose_step(
        _hip.ptr(R), _hip.ptr(O), n, d, _hip.dtype_code(dt), _hip.ptr(D), _hip.ptr(F), _hip.ptr(G),
        _hip.ptr(Rn), _hip.ptr(On), _hip.ptr(info), _hip.stream_ptr()))
    _raise_if_not_pd(info, R, O)
    b = lambda t: _back(t, Rs)  # noqa: E731
    return (n, b(D), b(F), b(G)), (b(Rn), b(On))


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def decompose(Rs, Os):
    """Full cyclic-reduction factorisation -> (ms, Ds, Fs, Gs)   (reference :287-309).

    The factor tensors themselves are plain (non-differentiable) device buffers; when Rs / Os
    require grad the returned decomp remembers them, and det(decomp) / solve(decomp, y) /
    mahal(decomp, y) differentiate with respect to them analytically (see _DetFn, _SolveFn)."""
    dec = _decompose_raw(Rs.detach(), Os.detach())
    if _needs_grad(Rs, Os):
        dec.inputs = (Rs, Os)
    return dec


def _decompose_raw(Rs, Os):
    _check_blocks(Rs, Os)
    N, d = Rs.shape[0], Rs.shape[1]
    R, O = _stage(Rs), _stage(Os)
    dev, dt = R.device, R.dtype
    Dp = torch.empty((N, d, d), dtype=dt, device=dev)
    Fp = torch.empty((N, d, d), dtype=dt, device=dev)
    Gp = torch.empty((N, d, d), dtype=dt, device=dev)
    info = torch.empty(1, dtype=torch.int32, device=dev)
    ws, nbytes = _hip.workspace(N, d, dt, _hip.OP_DECOMPOSE, dev)
    _hip.check(_hip.lib().cgps_decompose(
        _hip.ptr(R), _hip.ptr(O), N, d, _hip.dtype_code(dt), _hip.ptr(Dp), _hip.ptr(Fp), _hip.ptr(Gp),
        _hip.ptr(ws), nbytes, _hip.ptr(info), _hip.stream_ptr()))
    _raise_if_not_pd(info, R, O)
    ms, Ds, Fs, Gs = _views(Dp, Fp, Gp, N)
    if Rs.device.type != "cuda":
        Ds, Fs, Gs = ([_back(t, Rs) for t in lst] for lst in (Ds, Fs, Gs))
    return CRDecomp(ms, Ds, Fs, Gs, packed=(Dp, Fp, Gp), like=Rs)


def decompose_solve(Rs, Os, y):
    """(decompose(Rs, Os), solve(decomp, y)) in one call: what compute_insample_posterior does first
    (reference models.py:288-292: decompose, then solve), with the forward substitution of y riding along in the
    first pass of the factorisation -- the forward sweep never reads 7/8 of the factor (cgps_decompose_solve).
    An addition to the reference's surface (its callers that factor and solve together can switch to it);
    y: [N, d].  No autograd graph: with a gradient wanted, call decompose and solve."""
    _check_blocks(Rs, Os)
    N, d = Rs.shape[0], Rs.shape[1]
    R, O = _stage(Rs.detach()), _stage(Os.detach())
    dev, dt = R.device, R.dtype
    v = _stage(y.detach()).to(dt).reshape(N, d).contiguous()
    Dp, Fp, Gp = (torch.empty((N, d, d), dtype=dt, device=dev) for _ in range(3))
    xcrr = torch.empty((N, d), dtype=dt, device=dev)
    x = torch.empty((N, d), dtype=dt, device=dev)
    info = torch.empty(1, dtype=torch.int32, device=dev)
    ws, nbytes = _hip.workspace(N, d, dt, _hip.OP_DECOMPOSE_SOLVE, dev)
    _hip.check(_hip.lib().cgps_decompose_solve(
        _hip.ptr(R), _hip.ptr(O), _hip.ptr(v), N, d, _hip.dtype_code(dt), _hip.ptr(Dp), _hip.ptr(Fp), _hip.ptr(Gp),
        _hip.ptr(xcrr), _hip.ptr(x), _hip.ptr(ws), nbytes, _hip.ptr(info), _hip.stream_ptr()))
    _raise_if_not_pd(info, R, O)
    ms, Ds, Fs, Gs = _views(Dp, Fp, Gp, N)
    if Rs.device.type != "cuda":
        Ds, Fs, Gs = ([_back(t, Rs) for t in lst] for lst in (Ds, Fs, Gs))
    return CRDecomp(ms, Ds, Fs, Gs, packed=(Dp, Fp, Gp), like=Rs), _back(x.reshape(y.shape), y)


def mahal_and_det(Rs, Os, x):
    """(x^T J^-1 x, log|J|) in one fused sweep, factor not kept   (reference :380-438).
    Differentiable in Rs, Os and x (the reference trains through it, models.py:367-381)."""
    if _needs_grad(Rs, Os, x):
        return _MahalLogdetFn.apply(Rs, Os, x)
    return _mahal_and_det(Rs, Os, x, levelwise=False)


# ----------------------------------------------------------------------------
# analytic adjoints (checked against the reference's autograd in
# tests/test_oracle.py::test_oracle_gradients_match_reference_autograd).  With w = J^-1 x and
# Sig = J^-1:   m = x^T J^-1 x : dm/dx = 2w, dm/dR_i = -w_i w_i^T, dm/dO_i = -2 w_{i+1} w_i^T
#               l = log|J|     : dl/dR_i = Sig_ii, dl/dO_i = 2 Sig_{i+1,i}
#               s = u^T J^-1 y : with a = J^-1 u: ds/dy = a, ds/dR_i = -sym(a_i w_i^T),
#                                ds/dO_i = -(a_{i+1} w_i^T + w_{i+1} a_i^T)
# Every backward is solves and selected-inverse blocks on the same HIP kernels.
# ----------------------------------------------------------------------------
def _outer(a, b):
    return a.unsqueeze(-1) * b.unsqueeze(-2)


def _pair(a, b):
    """sum over the right-hand-side columns of a_c b_c^T per block row: [N, d] or [N, d, ...] -> [N, d, d]"""
    if a.dim() == 2:
        return _outer(a, b)
    return a.reshape(a.shape[0], a.shape[1], -1) @ b.reshape(b.shape[0], b.shape[1], -1).transpose(-1, -2)


class _MahalLogdetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Rs, Os, x):
        ctx.save_for_backward(Rs, Os, x)
        return _mahal_and_det(Rs.detach(), Os.detach(), x.detach(), levelwise=False)

    @staticmethod
    def backward(ctx, gm, gl):
        Rs, Os, x = ctx.saved_tensors
        need_blocks = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        gR = gO = None
        # factor and solve together: the forward substitution rides along in the factorisation (decompose_solve)
        dec, w = decompose_solve(Rs, Os, x.reshape(Rs.shape[0], Rs.shape[1]))
        if need_blocks:
            gR, gO = inverse_blocks(dec)
        if need_blocks:
            # gR = gl Sigma_diag - gm w w^T, gO = 2 (gl Sigma_off - gm w[1:] w[:-1]^T), written over
            # the blocks inverse_blocks just produced (one pass instead of ten element-wise kernels)
            if gR.is_cuda:
                N, d = gR.shape[0], gR.shape[1]
                wd = w.to(gR.dtype).contiguous()
                g2 = torch.stack([gm.reshape(()).to(gR.dtype), gl.reshape(()).to(gR.dtype)]).to(gR.device)
                _hip.check(_hip.lib().cgps_mahal_logdet_adjoint(
                    _hip.ptr(gR), _hip.ptr(gO), _hip.ptr(wd), N, d, _hip.dtype_code(gR.dtype),
                    _hip.ptr(g2[0:1]), _hip.ptr(g2[1:2]), _hip.stream_ptr()))
            else:       # blocks handed in as CPU tensors: the results were staged back to the host
                gR = gl * gR - gm * _outer(w, w)
                gO = 2 * (gl * gO - gm * _outer(w[1:], w[:-1]))
        return gR, gO, (2 * gm * w).reshape(x.shape)


class _DetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Rs, Os, dec):
        ctx.dec = dec
        return _det_raw(dec)

    @staticmethod
    def backward(ctx, g):
        Sd, So = inverse_blocks(ctx.dec)
        return g * Sd, 2 * g * So, None


class _SolveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Rs, Os, y, dec):
        w = _solve_raw(dec, y.detach())
        ctx.dec = dec
        ctx.save_for_backward(w)
        return w

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        a = _solve_raw(ctx.dec, g.contiguous())
        gR = gO = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            aw = _pair(a, w)
            gR = -0.5 * (aw + aw.transpose(-1, -2))
            gO = -(_pair(a[1:], w[:-1]) + _pair(w[1:], a[:-1]))
        return gR, gO, a, None


def _mahal_and_det(Rs, Os, x, levelwise):
    _check_blocks(Rs, Os)
    N, d = Rs.shape[0], Rs.shape[1]
    R, O = _stage(Rs), _stage(Os)
    dev, dt = R.device, R.dtype
    v = _stage(x, dt).reshape(N, d)
    out = torch.empty(2, dtype=torch.float64, device=dev)
    info = torch.empty(1, dtype=torch.int32, device=dev)
    ws, nbytes = _hip.workspace(N, d, dt, _hip.OP_MAHAL_LOGDET, dev)
    fn = _hip.lib().cgps_mahal_logdet_levelwise if levelwise else _hip.lib().cgps_mahal_logdet
    _hip.check(fn(_hip.ptr(R), _hip.ptr(O), _hip.ptr(v), N, d, _hip.dtype_code(dt), _hip.ptr(ws), nbytes,
                  _hip.ptr(out), _hip.ptr(info), _hip.stream_ptr()))
    _raise_if_not_pd(info, R, O)
    res = _back(out.to(dt), Rs)
    return res[0], res[1]


# ----------------------------------------------------------------------------
# operations on a stored factor
# ----------------------------------------------------------------------------
def _rhs(y, N, d, dt):
    """y [N, d] or [N, d, ...] (the reference's einsums carry a trailing "...", :52-57) as a contiguous
    [N, d, m] device tensor, plus m and the trailing shape to restore."""
    v = _stage(y, dt)
    tail = tuple(v.shape[2:])
    m = 1
    for s in tail:
        m *= s
    return v.reshape(N, d, m), m, tail


def halfsolve(decomp, y):
    """L^-1 (T y) as the per-level list ("CRR layout")   (reference :312-338).  y: [N, d] or [N, d, m]."""
    xs, _ = _halfsolve(decomp, y, want_mahal=False)
    return xs


def _halfsolve(decomp, y, want_mahal):
    Dp, Fp, Gp, N, d, like = _packed(decomp)
    dev, dt = Dp.device, Dp.dtype
    v, m, tail = _rhs(y, N, d, dt)
    xcrr = torch.empty((N, d, m), dtype=dt, device=dev)
    mah = torch.empty(1, dtype=torch.float64, device=dev) if want_mahal else None
    ws, nbytes = _hip.workspace(N, d, dt, _hip.OP_HALFSOLVE, dev, nrhs=m)
    _hip.check(_hip.lib().cgps_halfsolve(
        _hip.ptr(Dp), _hip.ptr(Fp), _hip.ptr(Gp), N, d, _hip.dtype_code(dt), m, _hip.ptr(v), _hip.ptr(xcrr),
        _hip.ptr(ws), nbytes, _hip.ptr(mah), _hip.stream_ptr()))
    ms, offD, _, _ = _hip.level_layout(N)
    xcrr = xcrr.reshape((N, d) + tail)
    xs = [_back(xcrr[offD[i]:offD[i + 1]], y) for i in range(len(ms))]
    return xs, mah


def backhalfsolve(decomp, ycrr):
    """T^T L^-T applied to a per-level list -> natural order   (reference :341-377)."""
    Dp, Fp, Gp, N, d, like = _packed(decomp)
    dev, dt = Dp.device, Dp.dtype
    src = ycrr[0]
    tail = tuple(src.shape[2:])
    b = torch.cat([_stage(t, dt).reshape(t.shape[0], d, -1) for t in ycrr], dim=0).contiguous()
    assert b.shape[0] == N
    m = b.shape[2]
    x = torch.empty((N, d, m), dtype=dt, device=dev)
    ws, nbytes = _hip.workspace(N, d, dt, _hip.OP_BACKSOLVE, dev, nrhs=m)
    _hip.check(_hip.lib().cgps_backsolve(
        _hip.ptr(Dp), _hip.ptr(Fp), _hip.ptr(Gp), N, d, _hip.dtype_code(dt), m, _hip.ptr(b), _hip.ptr(x),
        _hip.ptr(ws), nbytes, _hip.stream_ptr()))
    return _back(x.reshape((N, d) + tail), src)


def _decomp_inputs(decomp):
    inp = getattr(decomp, "inputs", None)
    return inp if inp is not None else (None, None)


def solve(decomp, y):
    """J^-1 y   (reference :441-444); y: [N, d] or [N, d, m] (up to eight columns share one read of the
    factor).  Differentiable in y and in the Rs / Os the factor came from."""
    Rs, Os = _decomp_inputs(decomp)
    if _needs_grad(Rs, Os, y):
        if Rs is None:       # only y carries grad: J^-1 is a constant symmetric operator
            like = _packed(decomp)[5]
            Rs = Os = torch.zeros(0, dtype=like.dtype, device=like.device)
        return _SolveFn.apply(Rs, Os, y, decomp)
    return _solve_raw(decomp, y)


def _solve_raw(decomp, y):
    Dp, Fp, Gp, N, d, like = _packed(decomp)
    dev, dt = Dp.device, Dp.dtype
    v, m, tail = _rhs(y, N, d, dt)
    x = torch.empty((N, d, m), dtype=dt, device=dev)
    ws, nbytes = _hip.workspace(N, d, dt, _hip.OP_SOLVE, dev, nrhs=m)
    _hip.check(_hip.lib().cgps_solve(
        _hip.ptr(Dp), _hip.ptr(Fp), _hip.ptr(Gp), N, d, _hip.dtype_code(dt), m, _hip.ptr(v), _hip.ptr(x),
        _hip.ptr(ws), nbytes, _hip.stream_ptr()))
    return _back(x.reshape((N, d) + tail), y)


def det(decomp):
    """log|J| from the factor (the reference's name; it is the log-determinant)   (reference :447-458).
    Differentiable in the Rs / Os the factor came from."""
    Rs, Os = _decomp_inputs(decomp)
    if _needs_grad(Rs, Os):
        return _DetFn.apply(Rs, Os, decomp)
    return _det_raw(decomp)


def _det_raw(decomp):
    Dp, Fp, Gp, N, d, like = _packed(decomp)
    dev, dt = Dp.device, Dp.dtype
    out = torch.empty(1, dtype=torch.float64, device=dev)
    ws, nbytes = _hip.workspace(N, d, dt, _hip.OP_LOGDET_FACTOR, dev)
    _hip.check(_hip.lib().cgps_logdet_factor(
        _hip.ptr(Dp), N, d, _hip.dtype_code(dt), _hip.ptr(ws), nbytes, _hip.ptr(out), _hip.stream_ptr()))
    return _back(out.to(dt), like)[0]


def mahal(decomp, y):
    """y^T J^-1 y = ||L^-1 T y||^2   (reference :461-467)."""
    _, m = _halfsolve(decomp, y, want_mahal=True)
    return _back(m.to(_packed(decomp)[0].dtype), y)[0]


def inverse_blocks(decomp):
    """Diagonal and lower off-diagonal blocks of J^-1   (reference :470-503)."""
    Dp, Fp, Gp, N, d, like = _packed(decomp)
    dev, dt = Dp.device, Dp.dtype
    Sd = torch.empty((N, d, d), dtype=dt, device=dev)
    So = torch.empty((N - 1, d, d), dtype=dt, device=dev)
    ws, nbytes = _hip.workspace(N, d, dt, _hip.OP_INVERSE_BLOCKS, dev)
    _hip.check(_hip.lib().cgps_inverse_blocks(
        _hip.ptr(Dp), _hip.ptr(Fp), _hip.ptr(Gp), N, d, _hip.dtype_code(dt), _hip.ptr(Sd), _hip.ptr(So),
        _hip.ptr(ws), nbytes, _hip.stream_ptr()))
    return _back(Sd, like), _back(So, like)


# ----------------------------------------------------------------------------
# banded products with the block upper-bidiagonal U (diagonal F, super-diagonal G).
# Thin device-side helpers of the reference surface (:15-200); they run as batched
# products wherever their inputs live and are not on the fused hot path.
# ----------------------------------------------------------------------------
def UU_T(diags, offdiags):
    """Block-tridiagonal part of U U^T -> (diagonal, lower off-diagonal)   (reference :15-37)."""
    nf, ng = diags.shape[0], offdiags.shape[0]
    dg = diags @ diags.transpose(-1, -2)
    dg[:ng] += offdiags @ offdiags.transpose(-1, -2)
    k = min(nf - 1, ng)
    return dg, diags[1:1 + k] @ offdiags[:k].transpose(-1, -2)


def Ux(diags, offdiags, x):
    """U @ x   (reference :40-60)."""
    nf, ng = diags.shape[0], offdiags.shape[0]
    out = torch.einsum("bij,bj...->bi...", diags, x[:nf])
    out[:ng] += torch.einsum("bij,bj...->bi...", offdiags, x[1:1 + ng])
    return out


def U_Tx(diags, offdiags, x):
    """U^T @ x   (reference :63-87)."""
    nf, ng = diags.shape[0], offdiags.shape[0]
    rows = nf + 1 if nf == ng else nf
    out = x.new_zeros((rows,) + tuple(x.shape[1:]))
    out[:nf] = torch.einsum("bji,bj...->bi...", diags, x)
    out[1:1 + ng] += torch.einsum("bji,bj...->bi...", offdiags, x[:ng])
    return out


def SigU(sig_dblocks, sig_offdblocks, u_dblocks, u_offdblocks):
    """Diagonal and upper-diagonal blocks of Sig @ U   (reference :90-136)."""
    nf, ng = u_dblocks.shape[0], u_offdblocks.shape[0]
    mid = sig_dblocks @ u_dblocks
    mid[1:] += sig_offdblocks @ u_offdblocks[:nf - 1]
    hi = sig_dblocks[:ng] @ u_offdblocks
    k = min(ng, nf - 1)
    hi[:k] += sig_offdblocks[:k].transpose(-1, -2) @ u_dblocks[1:1 + k]
    return mid, hi


def UtV_diags(u_dblocks, u_offdblocks, v_dblocks, v_offdblocks):
    """Diagonal blocks of U^T V   (reference :139-178)."""
    nf, ng = u_dblocks.shape[0], u_offdblocks.shape[0]
    rows = nf + 1 if nf == ng else nf
    out = u_dblocks.new_zeros((rows,) + tuple(u_dblocks.shape[1:]))
    out[:nf] = u_dblocks.transpose(-1, -2) @ v_dblocks
    out[1:1 + ng] += u_offdblocks.transpose(-1, -2) @ v_offdblocks
    return out


def interleave(a, b):
    """V[::2] = a; V[1::2] = b   (reference :181-200)."""
    n, m = a.shape[0], b.shape[0]
    k = min(n, m)
    out = a.new_empty((n + m,) + tuple(a.shape[1:]))
    out[0:2 * k:2] = a[:k]
    out[1:2 * k:2] = b[:k]
    out[2 * k:] = a[k:] if n > m else b[k:]
    return out
