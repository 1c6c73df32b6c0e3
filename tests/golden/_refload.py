"""Dev-only loader for the UNMODIFIED reference (build container only).

The reference's hot-path module imports three third-party packages that are not
installed in this image: ``torchtyping`` and ``typeguard``
(annotations / a decorator: no arithmetic) and ``gpytorch`` (one arithmetic
function, ``psd_safe_cholesky``, which on positive-definite input is
``torch.linalg.cholesky``).  ``models.py`` additionally wants
``pytorch_lightning.LightningModule`` (a ``torch.nn.Module`` with ``log``).
This file registers inert placeholders for those names in ``sys.modules`` and
then imports the reference's own files from ``/root/reference`` unchanged.

It works only where ``/root/reference`` exists.  Nothing here travels into the
product or runs on the GPU box; it exists so ``make_golden.py`` can record what
the reference itself computes.
"""
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("CGPS_REFERENCE_ROOT", "/root/reference")


def _install_placeholders():
    if "torchtyping" not in sys.modules:
        tt = types.ModuleType("torchtyping")

        class TensorType:  # annotation only
            def __class_getitem__(cls, item):
                return cls

            def __new__(cls, *a, **k):
                return cls

        tt.TensorType = TensorType
        tt.patch_typeguard = lambda *a, **k: None
        sys.modules["torchtyping"] = tt
    if "typeguard" not in sys.modules:
        tg = types.ModuleType("typeguard")
        tg.typechecked = lambda f=None, **k: f if f is not None else (lambda g: g)
        sys.modules["typeguard"] = tg
    if "gpytorch" not in sys.modules:
        gp = types.ModuleType("gpytorch")
        gpu = types.ModuleType("gpytorch.utils")
        gpc = types.ModuleType("gpytorch.utils.cholesky")
        gpe = types.ModuleType("gpytorch.utils.errors")

        class NotPSDError(RuntimeError):
            pass

        def psd_safe_cholesky(A, upper=False, out=None, jitter=None, max_tries=3):
            L, info = torch.linalg.cholesky_ex(A)
            if bool((info != 0).any()):
                raise NotPSDError("not positive definite (jitter path not reproduced)")
            return L

        gpc.psd_safe_cholesky = psd_safe_cholesky
        gpe.NotPSDError = NotPSDError
        gp.utils, gpu.cholesky, gpu.errors = gpu, gpc, gpe
        sys.modules.update({"gpytorch": gp, "gpytorch.utils": gpu,
                            "gpytorch.utils.cholesky": gpc, "gpytorch.utils.errors": gpe})
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")

        class LightningModule(torch.nn.Module):
            def log(self, *a, **k):
                return None

        pl.LightningModule = LightningModule
        sys.modules["pytorch_lightning"] = pl


def load_reference():
    """Return the reference's ``cyclic_gps.cyclic_reduction`` module."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise FileNotFoundError(REFERENCE_ROOT + " is not present (build container only)")
    _install_placeholders()
    # the product package has the same import name; make sure the reference wins here
    for k in [k for k in sys.modules if k == "cyclic_gps" or k.startswith("cyclic_gps.")]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import cyclic_gps.cyclic_reduction as ref_cr  # noqa
    finally:
        sys.path.remove(REFERENCE_ROOT)
    assert ref_cr.__file__.startswith(REFERENCE_ROOT), ref_cr.__file__
    return ref_cr


def load_reference_models():
    """Return the reference's ``cyclic_gps.models`` module (LEGFamily)."""
    load_reference()
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import cyclic_gps.models as ref_models  # noqa
    finally:
        sys.path.remove(REFERENCE_ROOT)
    return ref_models
