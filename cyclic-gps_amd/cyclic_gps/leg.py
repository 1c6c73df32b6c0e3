"""Thin harness for BASELINE config 5: LEG marginal likelihood + in-sample posterior mean.

This is NOT a re-implementation of the reference's `LEGFamily` (models.py is out of scope,
SURVEY.md section 2); it is the minimum of the caller's math needed to FEED the cyclic-reduction
path with the operands the reference feeds it, restated in our own code:

    G            = N N^T + R - R^T + 1e-5 I                       (models.py:152-159)
    Sigma^-1     = PEG precision blocks from exp(-1/2 dt G)        (models.py:181-239)
    K            = Sigma^-1 + blockdiag(B^T (LL^T)^-1 B)           (models.py:254-268)
    v            = x (LL^T)^-1 B                                   (models.py:270-280)
    log p(x)     = -1/2 (mahal + logdet)                           (models.py:301-372)
    posterior    = solve(decompose(K), v), inverse_blocks(...)     (models.py:282-298)

The d x d assembly (matrix exponentials, two small solves per time gap) is embarrassingly
parallel batched device work done with torch ops; everything block-tridiagonal goes through
cyclic_gps.cyclic_reduction, i.e. the HIP kernels.  Tensors follow the device of `ts`.
"""
import math
import os

import torch

from . import cyclic_reduction as cr


_eyes = {}


def _scaled_eye(n, scale, dtype, device):
    """scale * I, built once per (n, scale, dtype, device): N ~ 500 is launch-bound and this is two launches per use."""
    key = (n, scale, dtype, device)
    e = _eyes.get(key)
    if e is None:
        e = _eyes[key] = scale * torch.eye(n, dtype=dtype, device=device)
    return e


_llw = {}


def _ll_weights(n, dtype, device):
    """weights of (x (LL^T)^-1 x, log LL^T, k_mahal, k_det, log|Sigma^-1|) in the log-likelihood: -1/2 (quad - k_mahal + n log LLT
    + k_det - sig) (the n log 2 pi of the observation term is a host constant)"""
    key = (n, dtype, device)
    w = _llw.get(key)
    if w is None:
        w = _llw[key] = torch.tensor([-0.5, -0.5 * n, 0.5, -0.5, 0.5], dtype=dtype, device=device)
    return w


class LEGMatrices:
    """The four model matrices as the reference registers them (models.py:135-178):
    N [d,d] lower triangular, R [d,d] strictly lower (G uses R - R^T), B [obs,d],
    Lambda [obs,obs] lower triangular with softplus already applied."""

    def __init__(self, N, R, B, Lambda):
        self.N, self.R, self.B, self.Lambda = N, R, B, Lambda

    def to(self, device):
        return LEGMatrices(*(t.to(device) for t in (self.N, self.R, self.B, self.Lambda)))

    @property
    def G(self):
        d = self.N.shape[0]
        return torch.addmm(self.R - self.R.T + _scaled_eye(d, 1e-5, self.N.dtype, self.N.device), self.N, self.N.T)

    @property
    def LLT_inv(self):
        """(Lambda Lambda^T + 1e-9 I)^-1, obs_dim x obs_dim, used through plain matrix products
        (torch.linalg.solve's GPU backward faults on ROCm 7.0 for a 1x1 system with hundreds of
        right-hand sides; the inverse's backward is matmul only)."""
        return self.inv_of(self.LLT)

    @staticmethod
    def inv_of(LLT):
        if LLT.shape[0] == 1:                         # a single output: no factorisation call (and none inside a HIP graph)
            return 1.0 / LLT
        return torch.linalg.inv_ex(LLT)[0]            # inv_ex: no error check, hence no device->host synchronisation

    @property
    def LLT(self):
        o = self.Lambda.shape[0]
        return torch.addmm(_scaled_eye(o, 1e-9, self.Lambda.dtype, self.Lambda.device), self.Lambda, self.Lambda.T)


def _peg_precision_hip(ts, G):
    """The same blocks from one HIP kernel (cgps_peg_precision, csrc/cgps_leg.h): one lane per block
    row, matrix exponential and the two small solves in registers.  No autograd graph."""
    from . import _hip
    n, d = ts.shape[0], G.shape[0]
    ts = ts.to(G.dtype).contiguous()
    G = G.contiguous()
    Rs = torch.empty(n, d, d, dtype=G.dtype, device=G.device)
    Os = torch.empty(max(n - 1, 0), d, d, dtype=G.dtype, device=G.device)
    info = torch.zeros(1, dtype=torch.int32, device=G.device)
    _hip.check(_hip.lib().cgps_peg_precision(_hip.ptr(ts), _hip.ptr(G), n, d, _hip.dtype_code(G.dtype), _hip.ptr(Rs),
                                             _hip.ptr(Os), _hip.ptr(info), _hip.stream_ptr()))
    if cr.CHECK_POSITIVE_DEFINITE:
        bad = int(info.item())
        if bad:
            raise cr.NotPSDError("time gap next to row %d gives a singular PEG block (zero-length gap?)" % (bad - 1))
    return Rs, Os


class _PegPrecisionFn(torch.autograd.Function):
    """cgps_peg_precision with its analytic adjoint (cgps_peg_precision_adjoint, csrc/cgps_leg.h): a
    training step assembles its operands with the same kernel as an evaluation."""

    @staticmethod
    def forward(ctx, ts, G):
        ctx.save_for_backward(ts, G)
        Rs, Os = _peg_precision_hip(ts.detach(), G.detach())
        ctx.mark_non_differentiable()
        return Rs, Os

    @staticmethod
    def backward(ctx, gRs, gOs):
        from . import _hip
        ts, G = ctx.saved_tensors
        n, d = ts.shape[0], G.shape[0]
        if n < 2:
            return None, torch.zeros_like(G)
        tsd = ts.detach().to(G.dtype).contiguous()
        gRs = torch.zeros(n, d, d, dtype=G.dtype, device=G.device) if gRs is None else gRs.to(G.dtype).contiguous()
        gOs = torch.zeros(n - 1, d, d, dtype=G.dtype, device=G.device) if gOs is None else gOs.to(G.dtype).contiguous()
        nb = (n - 1 + 63) // 64
        part = torch.empty(nb, d, d, dtype=G.dtype, device=G.device)
        want_ts = ctx.needs_input_grad[0]
        gtau = torch.empty(n - 1, dtype=G.dtype, device=G.device) if want_ts else None
        _hip.check(_hip.lib().cgps_peg_precision_adjoint(
            _hip.ptr(tsd), _hip.ptr(G.detach().contiguous()), n, d, _hip.dtype_code(G.dtype), _hip.ptr(gRs), _hip.ptr(gOs),
            _hip.ptr(part), _hip.ptr(gtau), _hip.stream_ptr()))
        gts = None
        if want_ts:
            z = gtau.new_zeros(1)
            gts = (torch.cat([z, gtau]) - torch.cat([gtau, z])).to(ts.dtype)
        return gts, part.sum(0)


def peg_precision(ts, G):
    """Diagonal and lower off-diagonal blocks of the PEG prior precision (models.py:181-239).
    On the GPU one HIP kernel, differentiable in G and ts through its analytic adjoint (a second
    kernel); on CPU tensors batched torch ops."""
    d = G.shape[0]
    wants_grad = torch.is_grad_enabled() and (G.requires_grad or ts.requires_grad)
    if G.is_cuda and ts.is_cuda and 1 <= d <= 8 and G.dtype in (torch.float32, torch.float64):
        if wants_grad and os.environ.get("CGPS_LEG_TORCH_ASSEMBLY") != "1":
            return _PegPrecisionFn.apply(ts, G)
        if not wants_grad:
            return _peg_precision_hip(ts, G)
    eye = torch.eye(d, dtype=G.dtype, device=G.device)
    dt = ts[1:] - ts[:-1]
    E = torch.matrix_exp(-0.5 * G.unsqueeze(0) * dt.reshape(-1, 1, 1))
    Et = E.transpose(-1, -2)
    a = torch.linalg.solve(eye - Et @ E, Et)          # (I - E^T E)^-1 E^T
    b = torch.linalg.solve(eye - E @ Et, E)           # (I - E E^T)^-1 E
    c1, c2 = E @ a, Et @ b
    Rs = eye.repeat(ts.shape[0], 1, 1)
    Rs[:-1] += c2
    Rs[1:] += c1
    return Rs.contiguous(), (-b).contiguous()


def fused_supported(ts, G):
    """The assembly-in-registers form (cgps_leg_mahal_logdet) exists for this case: GPU tensors, no gradient wanted, a
    block size whose first pass runs one lane per row."""
    d = G.shape[0]
    return (G.is_cuda and ts.is_cuda and G.dtype in (torch.float32, torch.float64) and 1 <= d <= 7 and
            not (d == 6 and G.dtype == torch.float64) and
            not (torch.is_grad_enabled() and (G.requires_grad or ts.requires_grad)) and
            os.environ.get("CGPS_LEG_UNFUSED") != "1")


def leg_mahal_and_det(ts, G, A=None, v=None):
    """(v^T J^-1 v, log|J|) of J = PEG precision(ts, G) + blockdiag(A) in ONE kernel launch that never writes the blocks
    of J to memory (cgps_leg_mahal_logdet, csrc/cgps_tile_leg.h): what `peg_precision` + `cr.mahal_and_det` compute
    (models.py:349-367).  A [d,d] or None; v [N,d] or None (zeros).  No autograd graph."""
    from . import _hip
    n, d, dt = ts.shape[0], G.shape[0], G.dtype
    ts = ts.to(dt).contiguous()
    G = G.contiguous()
    A = None if A is None else A.to(dt).contiguous()
    v = None if v is None else v.to(dt).contiguous()
    ws, ws_bytes = _hip.workspace(n, d, dt, _hip.OP_MAHAL_LOGDET, G.device)
    out = torch.empty(2, dtype=torch.float64, device=G.device)
    info = torch.zeros(1, dtype=torch.int32, device=G.device)
    _hip.check(_hip.lib().cgps_leg_mahal_logdet(_hip.ptr(ts), _hip.ptr(G), _hip.ptr(A), _hip.ptr(v), n, d, _hip.dtype_code(dt),
                                                _hip.ptr(ws), ws_bytes, _hip.ptr(out), _hip.ptr(info), _hip.stream_ptr()))
    if cr.CHECK_POSITIVE_DEFINITE:
        bad = int(info.item())
        if bad:
            raise cr.NotPSDError("LEG system: a block near row %d is not positive definite (or a time gap has zero length)" % (bad - 1))
    out = out.to(dt)
    return out[0], out[1]


_pair_ws = {}


def leg_loglik_reductions(ts, G, A, v):
    """The two reductions of a LEG log-likelihood in ONE launch (cgps_leg_mahal_logdet_pair): returns
    (v^T K^-1 v, log|K|, log|Sigma^-1|) with K = PEG precision(ts, G) + blockdiag(A), Sigma^-1 the PEG precision itself
    (models.py:349-367: decompose + det of Sigma^-1, mahal_and_det of K).  No autograd graph."""
    from . import _hip
    n, d, dt = ts.shape[0], G.shape[0], G.dtype
    ts = ts.to(dt).contiguous()
    G, A, v = G.contiguous(), A.to(dt).contiguous(), v.to(dt).contiguous()
    nbytes = 2 * ((_hip._workspace_bytes(int(n), int(d), _hip.dtype_code(dt), _hip.OP_MAHAL_LOGDET) + 255) // 256 * 256)
    key = (G.device, torch.cuda.current_stream().cuda_stream)
    ws = _pair_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _pair_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=G.device)
    out = torch.empty(4, dtype=torch.float64, device=G.device)
    info = torch.zeros(2, dtype=torch.int32, device=G.device)
    _hip.check(_hip.lib().cgps_leg_mahal_logdet_pair(_hip.ptr(ts), _hip.ptr(G), _hip.ptr(A), _hip.ptr(v), n, d, _hip.dtype_code(dt),
                                                     _hip.ptr(ws), ws.numel(), _hip.ptr(out), _hip.ptr(info), _hip.stream_ptr()))
    if cr.CHECK_POSITIVE_DEFINITE:
        bad = info.tolist()
        if bad[0] or bad[1]:
            raise cr.NotPSDError("LEG system: a block near row %d is not positive definite (or a time gap has zero length)"
                                 % ((bad[0] or bad[1]) - 1))
    out = out.to(dt)
    return out[0], out[1], out[3]


def posterior_precision(m, ts):
    Rs, Os = peg_precision(ts, m.G)
    BtLB = m.B.T @ m.LLT_inv @ m.B
    return Rs + BtLB.unsqueeze(0), Os


def compute_v(m, xs):
    return (xs @ m.LLT_inv @ m.B).contiguous()


def log_likelihood(m, ts, xs):
    """log p(xs | ts) of the LEG model (models.py:301-372)."""
    LLT = m.LLT
    Li = m.inv_of(LLT)
    xl = xs @ Li
    v = (xl @ m.B).contiguous()
    G = m.G
    n = xs.shape[0]
    if fused_supported(ts, G) and LLT.shape[0] == 1:
        # the two reductions (prior precision: log-det only; posterior precision: mahal + log-det) never see their blocks
        # in memory, and run side by side in one launch: they share nothing but ts and G.  The scalar terms around them
        # are one product, one log and one weighted sum (N ~ 500 is launch-bound: every small launch is ~2.5 us)
        k_mahal, k_det, sig_inv_det = leg_loglik_reductions(ts, G, m.B.T @ Li @ m.B, v)
        terms = torch.stack([torch.dot(xl.reshape(-1), xs.reshape(-1)), torch.log(LLT[0, 0]), k_mahal, k_det, sig_inv_det])
        return torch.dot(terms, _ll_weights(n, terms.dtype, terms.device)) - 0.5 * n * math.log(2 * math.pi)
    llt_mahal = (xl * xs).sum()
    llt_det = (torch.log(2 * math.pi * LLT[0, 0]) if LLT.shape[0] == 1 else torch.logdet(2 * math.pi * LLT)) * n
    if fused_supported(ts, G):
        k_mahal, k_det, sig_inv_det = leg_loglik_reductions(ts, G, m.B.T @ Li @ m.B, v)
        return -0.5 * ((llt_mahal - k_mahal) + (llt_det + k_det - sig_inv_det))
    Rs, Os = peg_precision(ts, G)
    _, sig_inv_det = cr.mahal_and_det(Rs, Os, torch.zeros_like(v))       # = det(decompose(Rs, Os)), fused
    K_Rs = Rs + (m.B.T @ Li @ m.B).unsqueeze(0)
    k_mahal, k_det = cr.mahal_and_det(Rs=K_Rs, Os=Os, x=v)
    return -0.5 * ((llt_mahal - k_mahal) + (llt_det + k_det - sig_inv_det))


class Graphed:
    """``fn(*args, **kwargs)`` of this harness (``log_likelihood``, ``insample_posterior``,
    ``predict.make_predictions`` ...) captured once in a HIP graph and replayed: an evaluation loop over
    fixed shapes then costs one graph launch instead of the function's 25-60 kernel launches and the Python
    between them (N ~ 500 is launch-bound: BASELINE config 5).

    The graph reads its inputs from the tensors given here: change them IN PLACE (``copy_``) between
    replays.  Positive-definiteness is not checked inside the graph (the check reads a device word on the
    host); a non-PD system shows up as NaN / inf in the results.  No gradient.  ``fn`` must not read
    device values on the host (``predict.make_predictions(..., check_sorted=False)``)."""

    def __init__(self, fn, *args, warmup=2, **kwargs):
        dev = next(a.device for a in args if isinstance(a, torch.Tensor))
        prev = cr.CHECK_POSITIVE_DEFINITE
        cr.CHECK_POSITIVE_DEFINITE = False
        try:
            with torch.no_grad():
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):                   # workspaces and library handles exist before capture
                    for _ in range(warmup):
                        fn(*args, **kwargs)
                torch.cuda.current_stream(dev).wait_stream(side)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.value = fn(*args, **kwargs)
        finally:
            cr.CHECK_POSITIVE_DEFINITE = prev

    def __call__(self):
        """Replay; returns what ``fn`` returned at capture (the same tensors every time)."""
        self.graph.replay()
        return self.value


class GraphedLogLikelihood(Graphed):
    """``log_likelihood(m, ts, xs)`` as a replayable HIP graph (see ``Graphed``); ``value`` is the 0-d result."""

    def __init__(self, m, ts, xs, warmup=2):
        super().__init__(log_likelihood, m, ts, xs, warmup=warmup)


class GraphedValueAndGrad:
    """One training evaluation -- ``ll = log_likelihood(m, ts, xs); ll.backward()`` -- captured in a HIP
    graph (forward through the fused kernels, backward through ``solve`` + ``inverse_blocks`` + the two
    analytic adjoints).  ``m``'s four matrices must be leaf tensors that require a gradient; the graph
    leaves d ll / d (N, R, B, Lambda) in their ``.grad`` (overwritten at every replay) and ll in ``value``.
    Update the matrices / data in place between replays (an optimiser's ``step()`` does).  As with
    ``GraphedLogLikelihood`` nothing is checked on the host inside the graph."""

    def __init__(self, m, ts, xs, warmup=3):
        self.params = [m.N, m.R, m.B, m.Lambda]
        if not all(p.is_leaf and p.requires_grad for p in self.params):
            raise ValueError("the four LEG matrices must be leaf tensors with requires_grad=True")
        prev = cr.CHECK_POSITIVE_DEFINITE
        cr.CHECK_POSITIVE_DEFINITE = False
        try:
            cur = torch.cuda.current_stream(ts.device)
            side = torch.cuda.Stream(device=ts.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    for p in self.params:
                        p.grad = None
                    log_likelihood(m, ts, xs).backward()
            cur.wait_stream(side)
            for p in self.params:
                p.grad = None                               # the capture allocates the gradients in the graph's pool
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.value = log_likelihood(m, ts, xs)
                self.value.backward()
            self.grads = [p.grad for p in self.params]
        finally:
            cr.CHECK_POSITIVE_DEFINITE = prev

    def __call__(self):
        """Replay; returns (ll, [d ll / d N, d ll / d R, d ll / d B, d ll / d Lambda]) -- the same tensors every time."""
        self.graph.replay()
        return self.value, self.grads


def insample_posterior(m, ts, xs):
    """Posterior mean [N,d] and (diag, lower off-diag) covariance blocks (models.py:282-298)."""
    K_Rs, K_Os = posterior_precision(m, ts)
    v = compute_v(m, xs)
    if K_Rs.is_cuda and not (torch.is_grad_enabled() and (K_Rs.requires_grad or K_Os.requires_grad or v.requires_grad)):
        dec, mean = cr.decompose_solve(K_Rs, K_Os, v)      # factor and solve together (cgps_decompose_solve)
    else:
        dec = cr.decompose(Rs=K_Rs, Os=K_Os)
        mean = cr.solve(dec, v)
    return mean, cr.inverse_blocks(dec)


# ---- the config-5 workload -----------------------------------------------------------------
def co2_like_series(rows=770, seed=0, dtype=torch.float64):
    """Mauna-Loa-shaped monthly series (decimal date, ppm): quadratic trend + annual and
    semi-annual cycles + small noise.  Stand-in for ../data/co2_mm_mlo.csv, which is not part of
    the reference repo (co2_data_experiments.py:17)."""
    g = torch.Generator().manual_seed(seed)
    t = 1958.2 + torch.arange(rows, dtype=dtype) / 12.0
    u = t - 1958.0
    x = 315.0 + 0.8 * u + 0.012 * u * u + 2.9 * torch.sin(2 * math.pi * t) + 0.8 * torch.sin(4 * math.pi * t + 0.6)
    x = x + 0.3 * torch.randn(rows, dtype=dtype, generator=g)
    return t, x.unsqueeze(-1)


def load_co2_csv(path, dtype=torch.float64):
    """The real file, when a user has it: columns as in co2_data_experiments.py:17-19."""
    import numpy as np
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#") or line[0].isalpha():
                continue
            rows.append([float(tok) for tok in line.replace(",", " ").split()])
    a = np.array(rows)
    return torch.tensor(a[:, 2], dtype=dtype), torch.tensor(a[:, 3], dtype=dtype).unsqueeze(-1)


def co2_workload(path=None, dtype=torch.float64):
    """(all_ts, all_xs, train_ts, train_xs) standardised and masked as the reference does
    (co2_data_experiments.py:21-30, dataset_process_utils.py:9-25)."""
    if path is not None and os.path.exists(path):
        t, x = load_co2_csv(path, dtype)
    else:
        t, x = co2_like_series(dtype=dtype)
    ts = 12 * (t - t.min())
    xs = x - x.mean()
    xs = xs / xs.std()
    train_ts = torch.cat([ts[:262], ts[502:-28]])
    train_xs = torch.cat([xs[:262], xs[502:-28]])
    return ts, xs, train_ts, train_xs
