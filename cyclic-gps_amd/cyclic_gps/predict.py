"""Prediction glue of the LEG model, batched over the target times (the step right AFTER the
cyclic-reduction path).

The reference walks the targets in a Python loop (models.py:467-514): per target one
``searchsorted`` hit, one or two matrix exponentials through an eigendecomposition of G
(model_utils.py:12-29) and one ``gaussian_stitch`` (model_utils.py:64-107) of a 2d x 2d or 3d x 3d
Gaussian.  Here the same arithmetic runs for ALL targets at once as batched tensor ops on the
device the in-sample posterior lives on (``searchsorted`` once, one batched ``matrix_exp``, one
batched d x d / 2d x 2d solve per branch): nothing comes back to the host inside
``make_predictions``.

Same names and argument meaning as the reference (methods of ``LEGFamily`` there, functions of a
``LEGMatrices`` here):

    gaussian_stitch(joint_mean, joint_cov, marginal_mean, marginal_cov)      model_utils.py:64-107
    build_2x2_block, build_3x3_block                                          model_utils.py:31-60
    forecast(eG, ip_mean, ip_cov)                                             models.py:394-408
    interpolate(eG1, eG2, prev_ip_mean, ..., next_ip_cov_diag)                models.py:410-452
    intercast(m, ip_mean, ip_cov, ts, target_ts)                              models.py:455-514
    predictive_posterior(m, ts, xs, target_ts)                                models.py:516-528
    make_predictions(m, ts, xs, target_ts)                                    models.py:530-546

The in-sample posterior itself (``decompose`` / ``solve`` / ``inverse_blocks``) is the HIP path
(``leg.insample_posterior``).  Pinned by ``tests/golden/leg_*.npz`` (``pp_mean``, ``pp_cov``,
``pred_mean``, ``pred_cov`` recorded from the reference's own ``make_predictions``).
"""
import os

import torch

from . import leg


def build_2x2_block(a, b, c, d):
    """[[a, b], [c, d]], batched over leading dimensions   (reference model_utils.py:31-50)."""
    return torch.cat([torch.cat([a, b], dim=-1), torch.cat([c, d], dim=-1)], dim=-2)


def build_3x3_block(a, b, c, d, e, f, g, h, i):
    """3 x 3 block matrix, batched   (reference model_utils.py:52-60)."""
    return torch.cat([torch.cat([a, b, c], dim=-1), torch.cat([d, e, f], dim=-1), torch.cat([g, h, i], dim=-1)], dim=-2)


def gaussian_stitch(joint_mean, joint_cov, marginal_mean, marginal_cov):
    """E and Cov of y under q(x, y) = p2(x) p1(y | x), p1 = N(joint_mean, joint_cov) over (x, y),
    p2 = N(marginal_mean, marginal_cov) over x   (reference model_utils.py:64-107).  Batched over
    leading dimensions (the reference transposes with ``.T``, i.e. is un-batched)."""
    m = marginal_cov.shape[-1]
    Axx, Ayx = joint_cov[..., :m, :m], joint_cov[..., m:, :m]
    # mean_transformer = C_yx C_xx^-1
    # (solve_ex: no error check, hence no device->host synchronisation; Axx = [[I, E^T], [E, I]] with
    # E a contraction, or I: never singular)
    Mt = torch.linalg.solve_ex(Axx.transpose(-1, -2), Ayx.transpose(-1, -2))[0].transpose(-1, -2)
    mean = joint_mean[..., m:] + (Mt @ marginal_mean[..., None])[..., 0]
    cond = joint_cov[..., m:, m:] - Mt @ joint_cov[..., :m, m:]
    return mean, cond + Mt @ marginal_cov @ Mt.transpose(-1, -2)


def compute_eG(G, diffs):
    """exp(-1/2 d G) for every d in diffs [m] -> [m, rank, rank].  (The reference goes through a
    complex eigendecomposition of G, model_utils.py:12-29; the matrix exponential itself is the
    same quantity without the detour through complex arithmetic.)"""
    return torch.matrix_exp(-0.5 * diffs.reshape(-1, 1, 1) * G.unsqueeze(0))


def forecast(eG, ip_mean, ip_cov):
    """One step away from an in-sample point: latent at the target given the in-sample posterior
    N(ip_mean, ip_cov) of its neighbour   (reference models.py:394-408).  Batched."""
    rank = eG.shape[-1]
    I = torch.eye(rank, dtype=eG.dtype, device=eG.device).expand(eG.shape)
    joint_mean = torch.zeros(eG.shape[:-2] + (2 * rank,), dtype=eG.dtype, device=eG.device)
    joint_cov = build_2x2_block(I, eG.transpose(-1, -2), eG, I)
    return gaussian_stitch(joint_mean, joint_cov, ip_mean, ip_cov)


def interpolate(eG1, eG2, prev_ip_mean, prev_ip_cov_diag, prev_ip_cov_offdiag, next_ip_mean, next_ip_cov_diag):
    """Latent at a target between two in-sample points   (reference models.py:410-452): eG1 spans
    previous -> target, eG2 target -> next.  Batched."""
    rank = eG1.shape[-1]
    I = torch.eye(rank, dtype=eG1.dtype, device=eG1.device).expand(eG1.shape)
    T = lambda a: a.transpose(-1, -2)  # noqa: E731
    eG3 = eG1 @ eG2
    joint_latent_mean = torch.zeros(eG1.shape[:-2] + (3 * rank,), dtype=eG1.dtype, device=eG1.device)
    joint_latent_cov = build_3x3_block(I, T(eG3), T(eG1),
                                       eG3, I, eG2,
                                       eG1, T(eG2), I)
    joint_ip_mean = torch.cat([prev_ip_mean, next_ip_mean], dim=-1)
    joint_ip_cov = build_2x2_block(prev_ip_cov_diag, T(prev_ip_cov_offdiag), prev_ip_cov_offdiag, next_ip_cov_diag)
    return gaussian_stitch(joint_latent_mean, joint_latent_cov, joint_ip_mean, joint_ip_cov)


def _intercast_hip(G, ip_mean, Rs, Os, ts, target_ts):
    """All targets in one HIP kernel (cgps_leg_intercast, csrc/cgps_leg.h): one lane per target, the two
    matrix exponentials and the stitch in registers.  No autograd graph (prediction is inference)."""
    from . import _hip
    dt, dev = ip_mean.dtype, ip_mean.device
    n, d, p = ip_mean.shape[0], G.shape[0], target_ts.shape[0]
    c = lambda t: t.detach().to(device=dev, dtype=dt).contiguous()   # noqa: E731
    ts_, tt_, G_, mu_, Rs_, Os_ = c(ts), c(target_ts), c(G), c(ip_mean), c(Rs), c(Os)
    means = torch.empty(p, d, dtype=dt, device=dev)
    covs = torch.empty(p, d, d, dtype=dt, device=dev)
    _hip.check(_hip.lib().cgps_leg_intercast(_hip.ptr(ts_), n, _hip.ptr(tt_), p, _hip.ptr(G_), d, _hip.dtype_code(dt),
                                             _hip.ptr(mu_), _hip.ptr(Rs_), _hip.ptr(Os_), _hip.ptr(means),
                                             _hip.ptr(covs), _hip.stream_ptr()))
    return means, covs


def intercast(m, ip_mean, ip_cov, ts, target_ts, thresh=1e-10, check_sorted=True):
    """Posterior of the latent at every target time from the in-sample posterior
    (reference models.py:455-514; ``thresh`` is accepted and unused there as well).

    ip_mean [n, rank]; ip_cov = {"Rs": [n, rank, rank], "Os": [n-1, rank, rank]} (or a pair);
    ts [n] sorted; target_ts [p] strictly increasing (asserted like the reference does, :471 -- the
    one device->host read in here; check_sorted=False skips it).  Returns (means [p, rank],
    covs [p, rank, rank]).  Branches per target, as the reference takes them: before the first
    observation (backward forecast), after the last (forward forecast), at the first / last
    observation (in-sample values), otherwise interpolation between the two neighbours.
    On the GPU: one HIP kernel, a lane per target (``_intercast_hip``).  On CPU tensors (or with a
    gradient wanted, or CGPS_LEG_TORCH_INTERCAST=1) batched torch ops: every branch is evaluated for
    every target with clamped (harmless) arguments and the results are selected with masks, so no
    branch decision leaves the device."""
    Rs, Os = (ip_cov["Rs"], ip_cov["Os"]) if isinstance(ip_cov, dict) else ip_cov
    G = m.G
    n, rank = ts.shape[0], G.shape[0]
    target_ts = target_ts.to(ts.dtype)
    if check_sorted:
        assert bool((target_ts[1:] - target_ts[:-1] > 0).all())      # reference :471
    p = target_ts.shape[0]
    if (ip_mean.is_cuda and 1 <= rank <= 8 and ip_mean.dtype in (torch.float32, torch.float64)
            and os.environ.get("CGPS_LEG_TORCH_INTERCAST") != "1"
            and not (torch.is_grad_enabled() and any(t.requires_grad for t in (ip_mean, Rs, Os, G, ts, target_ts)))):
        return _intercast_hip(G, ip_mean, Rs, Os, ts, target_ts)
    idx = torch.searchsorted(ts, target_ts)
    close = lambda a, b: (a - b).abs() <= 1e-8 + 1e-5 * b.abs()      # noqa: E731  torch.allclose(a, b) defaults
    at_first = (idx == 0) & close(target_ts, ts[0])
    at_last = (idx > 0) & close(target_ts, ts[-1])
    back = (idx == 0) & ~at_first
    fwd = (idx == n) & ~at_last
    zero = torch.zeros((), dtype=ts.dtype, device=ts.device)
    ex = lambda v, shape: v.expand(shape)                            # noqa: E731
    # forecasts from the first / last observation
    mb, cb = forecast(compute_eG(G, torch.maximum(ts[0] - target_ts, zero)).transpose(-1, -2),
                      ex(ip_mean[0], (p, rank)), ex(Rs[0], (p, rank, rank)))
    mf, cf = forecast(compute_eG(G, torch.maximum(target_ts - ts[-1], zero)),
                      ex(ip_mean[-1], (p, rank)), ex(Rs[-1], (p, rank, rank)))
    means = torch.where(back[:, None], mb, mf)
    covs = torch.where(back[:, None, None], cb, cf)
    if n > 1:                                                         # interpolation between neighbours
        j = idx.clamp(1, n - 1)
        mi, ci = interpolate(
            eG1=compute_eG(G, torch.maximum(target_ts - ts[j - 1], zero)),
            eG2=compute_eG(G, torch.maximum(ts[j] - target_ts, zero)),
            prev_ip_mean=ip_mean[j - 1], prev_ip_cov_diag=Rs[j - 1], prev_ip_cov_offdiag=Os[j - 1],
            next_ip_mean=ip_mean[j], next_ip_cov_diag=Rs[j])
        mid = ~(back | fwd)
        means = torch.where(mid[:, None], mi, means)
        covs = torch.where(mid[:, None, None], ci, covs)
    means = torch.where(at_first[:, None], ip_mean[0], torch.where(at_last[:, None], ip_mean[-1], means))
    covs = torch.where(at_first[:, None, None], Rs[0], torch.where(at_last[:, None, None], Rs[-1], covs))
    return means, covs


def predictive_posterior(m, ts, xs, target_ts, check_sorted=True):
    """E[z(t) | x] and Cov[z(t) | x] at every target   (reference models.py:516-528).  The in-sample
    posterior is decompose + solve + inverse_blocks on the HIP path."""
    mean, (cRs, cOs) = leg.insample_posterior(m, ts, xs)
    return intercast(m, mean, {"Rs": cRs, "Os": cOs}, ts, target_ts, check_sorted=check_sorted)


def make_predictions(m, ts, xs, target_ts, check_sorted=True):
    """Predicted observation mean [p, obs_dim] and covariance [p, obs_dim, obs_dim] at the targets
    (reference models.py:530-546: the latent's B-image; the observation noise is not added there).
    check_sorted=False skips the reference's assertion on the targets (:471), the one device->host read:
    the call is then capturable in a HIP graph (``leg.Graphed``)."""
    pm, pv = predictive_posterior(m, ts, xs, target_ts, check_sorted=check_sorted)
    return pm @ m.B.T, m.B.unsqueeze(0) @ pv @ m.B.T.unsqueeze(0)
