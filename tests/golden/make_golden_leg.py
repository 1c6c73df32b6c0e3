"""Golden vectors for the config-5 harness (cyclic_gps/leg.py), recorded by running the
UNMODIFIED reference LEGFamily (cyclic_gps/models.py) in the build container:
    python tests/golden/make_golden_leg.py
Writes leg_co2like.npz (rank 5, obs_dim 1, the CO2-shaped workload of co2_data_experiments.py
on our synthetic Mauna-Loa-like series) and leg_small_{regular,irregular}.npz (n = 33, rank 3,
obs_dim 2, with the reference's dense naive likelihood, model_utils.py:131-142).
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import _refload  # noqa: E402

warnings.filterwarnings("ignore")


def record(model, ts, xs, naive=None, target_ts=None):
    model.register_model_matrices_from_params()
    pred = {}
    if target_ts is not None:
        # prediction glue (models.py:394-546, model_utils.py:64-107): forecast / interpolate / exact hits
        with torch.no_grad():
            pp_mean, pp_cov = model.predictive_posterior(ts, xs, target_ts)
            p_mean, p_cov = model.make_predictions(ts, xs, target_ts)
        pred = dict(target_ts=target_ts.numpy(), pp_mean=pp_mean.numpy(), pp_cov=pp_cov.numpy(),
                    pred_mean=p_mean.numpy(), pred_cov=p_cov.numpy())
    with torch.no_grad():
        Sig_Rs, Sig_Os = model.compute_PEG_precision(ts)
        K_Rs, K_Os = model.compute_posterior_precision(ts)
        v = model.compute_v(xs)
        ll = model.log_likelihood(ts, xs)
        mean, cov = model.compute_insample_posterior(ts, xs)
    out = dict(ts=ts.numpy(), xs=xs.numpy(), N=model.N.numpy(), R=model.R.numpy(), B=model.B.detach().numpy(),
               Lambda=model.Lambda.numpy(), G=model.G.numpy(), Sig_Rs=Sig_Rs.numpy(), Sig_Os=Sig_Os.numpy(),
               K_Rs=K_Rs.numpy(), K_Os=K_Os.numpy(), v=v.numpy(), ll=float(ll), post_mean=mean.numpy(),
               post_cov_Rs=cov["Rs"].numpy(), post_cov_Os=cov["Os"].numpy())
    if naive is not None:
        out["naive_ll"] = float(naive)
    out.update(pred)
    return out


def record_grads(models, ts, xs, seed, rank, obs_dim):
    """d(log-likelihood)/d(model matrices) from the reference's autograd through its own cyclic
    reduction (training path, models.py:374-381), expressed per matrix ENTRY so that a harness
    holding the matrices directly can be compared: N and R entries are the parameters themselves
    (models.py:135-143); Lambda = softplus(params) so d/dLambda = d/dparams / sigmoid(params)."""
    torch.manual_seed(seed)
    m = models.LEGFamily(rank=rank, obs_dim=obs_dim, train=True, data_type=torch.float64)
    m.double()
    ll = m.log_likelihood(ts, xs)
    ll.backward()
    d = rank
    gN = torch.zeros(d, d, dtype=torch.float64)
    gN[m.N_idxs] = m.N_params.grad
    gR = torch.zeros(d, d, dtype=torch.float64)
    gR[m.R_idxs] = m.R_params.grad
    gL = torch.zeros(obs_dim, obs_dim, dtype=torch.float64)
    gL[m.Lambda_idxs] = m.Lambda_params.grad / torch.sigmoid(m.Lambda_params.detach())
    return dict(grad_ll=float(ll), gN=gN.numpy(), gR=gR.numpy(), gB=m.B.grad.numpy(), gLambda=gL.numpy())


def main():
    models = _refload.load_reference_models()
    # the product package shares the import name `cyclic_gps`; load our leg.py by path for the data recipe
    import importlib.util
    spec = importlib.util.spec_from_file_location("cgps_leg_data", os.path.join(ROOT, "cyclic-gps_amd", "cyclic_gps", "leg.py"))
    src = open(spec.origin).read().replace("from . import cyclic_reduction as cr", "cr = None")
    ns = {}
    exec(compile(src, spec.origin, "exec"), ns)
    all_ts, _, train_ts, train_xs = ns["co2_workload"]()
    # every month of the series (training months = exact hits, the masked gap = interpolation, the held-out
    # tail = forecast) plus a few times before the first observation and in between months
    co2_targets = torch.sort(torch.cat([all_ts, torch.tensor([-30.0, -7.5, -0.25], dtype=all_ts.dtype),
                                        all_ts[100:110] + 0.37, all_ts[-1:] + 17.0]))[0]

    torch.manual_seed(20240611)
    m = models.LEGFamily(rank=5, obs_dim=1, train=False, data_type=torch.float64)
    m.double()
    rec = record(m, train_ts, train_xs, target_ts=co2_targets)
    rec.update(record_grads(models, train_ts, train_xs, 20240611, 5, 1))
    np.savez_compressed(os.path.join(HERE, "leg_co2like.npz"), **rec)

    sys.path.insert(0, _refload.REFERENCE_ROOT)
    from cyclic_gps.model_utils import compute_log_marginal_likelihood
    sys.path.remove(_refload.REFERENCE_ROOT)
    g = torch.Generator().manual_seed(99)
    for spacing in ("regular", "irregular"):
        n = 33
        if spacing == "regular":
            ts = torch.cumsum(torch.ones(n, dtype=torch.float64), dim=0)
        else:
            ts = torch.cumsum(torch.empty(n, dtype=torch.float64).exponential_(1.0, generator=g) + 0.01, dim=0)
        xs = torch.randn(n, 2, dtype=torch.float64, generator=g).cumsum(0) * 0.1
        torch.manual_seed(7)
        m = models.LEGFamily(rank=3, obs_dim=2, train=False, data_type=torch.float64)
        m.double()
        naive = compute_log_marginal_likelihood(N=m.N, R=m.R, B=m.B.detach(), Lambda=m.calc_Lambda_Lambda_T(m.Lambda),
                                                ts=ts, xs=xs)
        targets = torch.sort(torch.cat([ts[:1] - 2.0, ts[:1], 0.5 * (ts[3:9] + ts[4:10]), ts[15:17], ts[-1:], ts[-1:] + 0.75,
                                        ts[-1:] + 6.0]))[0]
        np.savez_compressed(os.path.join(HERE, "leg_small_%s.npz" % spacing), **record(m, ts, xs, naive, target_ts=targets))
    print("wrote LEG golden files")


if __name__ == "__main__":
    main()
