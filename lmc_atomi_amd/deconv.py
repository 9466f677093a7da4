"""Bayesian imaging deconvolution experiment -- the caller of the hot path, re-written (not ported) from the
sampling branch of the reference driver ``prox_lmc_deconv.py:40-135,447-735``: same flags, same nine models
(3 box blurs x {TV, MC-TV, ME-TV}), same step sizes, posterior means and SNR / PSNR / MSE.  The MAP branch
(``:138-445``) and all plotting are out of scope; results go to a dict / ``.npz`` instead of figures.

    python -m lmc_atomi_amd.deconv --alg MYULA --N 1000 --n_chains 64 --out means.npz

The image is an array argument (or a synthetic one): the reference loads ``skimage.data.camera()`` / a PNG.
"""
from __future__ import annotations

import argparse
import time

import numpy as np

from . import (Convolve2D, Gradient, L2, L21, TV, L2_ncvx_tv, MoreauYosidaUnadjustedLangevin, MoreauYosidaMetropolisAdjustedLangevin,
               UnadjustedLangevinPrimalDual, signal_noise_ratio, peak_signal_noise_ratio, mean_squared_error)


def synthetic_image(ny=512, nx=512, seed=1234):
    """Piecewise-constant blocks + ramp in [0, 255] (stands in for skimage.data.camera(), prox_lmc_deconv.py:48)."""
    rng = np.random.default_rng(seed)
    img = np.zeros((ny, nx))
    for _ in range(12):
        i0, j0 = rng.integers(0, ny - 8), rng.integers(0, nx - 8)
        i1, j1 = rng.integers(i0 + 4, ny + 1), rng.integers(j0 + 4, nx + 1)
        img[i0:i1, j0:j1] = rng.uniform(20, 235)
    img += np.linspace(0, 20, nx)[None, :]
    return np.clip(img, 0, 255)


def prox_lmc_deconv(gamma_mc=15., gamma_me=15., sigma=0.75, tau=0.3, N=1000, niter_l2=50, niter_tv=10, image=None,
                    alg='ULPDA', seed=0, n_chains=None, burn_in=0, thin=1, models=None, verbose=True, diagnostics=None, rtol=1e-4):
    """Posterior means of the nine models M1..M9 (prox_lmc_deconv.py:447-703) by ULPDA or MYULA on the GPU.

    ``rtol``: the early exit of the TV proxes AS THE REFERENCE IS CONFIGURED -- ``pyproximal.TV(dims, sigma, niter=niter_tv)`` leaves upstream's default
    ``rtol = 1e-4`` in force (:122) and ``algs.L2_ncvx_tv`` has it as its own default (algs.py:130,169).  Decided on the device, chain by chain (DESIGN
    3.0r); ``rtol=0``: every prox runs all its passes.  (MYMALA: the TV prior keeps the fixed count -- its Metropolis test needs one proposal map.)

    ``n_chains=None`` runs the reference's single chain (every iterate kept on the host, mean over iterates,
    ``:474``); ``n_chains=C`` runs C chains per model and averages over chains and kept iterations.
    Returns ``{"M1": {"mean", "snr", "psnr", "mse", "seconds"}, ...}``.
    """
    img = synthetic_image() if image is None else np.asarray(image, dtype=np.float64)
    ny, nx = img.shape
    rng = np.random.default_rng(seed)
    H = {}
    for k in (5, 6, 7):                                             # prox_lmc_deconv.py:55-69
        H[k] = Convolve2D((ny, nx), h=np.ones((k, k)) / (k * k), offset=(k // 2, k // 2))
    y = H[5].matvec(img.ravel()).reshape(ny, nx) + rng.normal(0, sigma, size=(ny, nx))   # :59 (all models share y)
    L = 1. / sigma ** 2                                             # :88-94
    tau0, mu0 = 0.95 / L, 1.
    gamma_myula = 1. / L
    tau_myula = 0.2 * gamma_myula
    Gop = Gradient(dims=(ny, nx))                                   # :98
    x0 = np.zeros(ny * nx)                                          # :135

    def data_term(k, kind):                                         # :101-113
        if kind == "tv":
            return L2(Op=H[k], b=y.ravel(), sigma=1 / sigma ** 2, niter=50, warm=True)
        if kind == "mc":
            return L2_ncvx_tv(dims=(ny, nx), Op=H[k], Op2=Gop, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau, gamma=gamma_mc,
                              isotropic=True, niter=niter_l2, warm=True)
        return L2_ncvx_tv(dims=(ny, nx), Op=H[k], b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau, gamma=gamma_me,
                          isotropic=True, niter=niter_l2, rtol=rtol, warm=True)

    order = [("M1", 5, "tv"), ("M2", 5, "mc"), ("M3", 5, "me"), ("M4", 6, "tv"), ("M5", 6, "mc"), ("M6", 6, "me"),
             ("M7", 7, "tv"), ("M8", 7, "mc"), ("M9", 7, "me")]
    out = {}
    for name, k, kind in order:
        if models is not None and name not in models:
            continue
        f = data_term(k, kind)
        t0 = time.time()
        if alg == 'ULPDA':                                          # :455-464
            res = UnadjustedLangevinPrimalDual(f, L21(ndim=2, sigma=tau), Gop, tau=tau0, mu=mu0, theta=1., x0=x0, gfirst=False,
                                               niter=N, seed=seed, n_chains=n_chains, burn_in=burn_in, thin=thin,
                                               **({"diagnostics": diagnostics} if (diagnostics and n_chains) else {}))
        elif alg == 'MYULA':                                        # :465-473
            res = MoreauYosidaUnadjustedLangevin(f, TV(dims=(ny, nx), sigma=tau, niter=niter_tv, rtol=rtol), tau=tau_myula,
                                                 gamma=gamma_myula, x0=x0, niter=N, seed=seed, n_chains=n_chains,
                                                 burn_in=burn_in, thin=thin,
                                                 **({"diagnostics": diagnostics} if (diagnostics and n_chains) else {}))
        elif alg == 'MYMALA':                                       # Metropolis-adjusted MYULA (generalises prox_lmc.py:134-158)
            res = MoreauYosidaMetropolisAdjustedLangevin(f, TV(dims=(ny, nx), sigma=tau, niter=niter_tv), tau=tau_myula,
                                                         gamma=gamma_myula, x0=x0, niter=N, seed=seed, n_chains=n_chains or 1,
                                                         burn_in=burn_in, thin=thin)
        else:
            raise ValueError("alg must be 'ULPDA', 'MYULA' or 'MYMALA'")
        mean = res.mean(axis=0) if isinstance(res, np.ndarray) else res.mean.cpu().numpy().ravel()      # :474
        out[name] = {"mean": mean.reshape(ny, nx),
                     "snr": float(signal_noise_ratio(img, mean, dims=(ny, nx))),            # :707-735
                     "psnr": float(peak_signal_noise_ratio(img, mean, dims=(ny, nx))),
                     "mse": float(mean_squared_error(img, mean, dims=(ny, nx))),
                     "seconds": time.time() - t0}
        diag = getattr(res, "diagnostics", None)
        if diag is not None:                                        # split R-hat / ESS across chains (diagnostics.py)
            out[name].update(rhat_max=diag["rhat_max"], ess_min=diag["ess_min"], rhat=diag["rhat"].cpu().numpy(),
                             ess=diag["ess"].cpu().numpy())
            if verbose:
                print(f"    {name}: {diag['n_chains']} chains x {diag['n_kept']} kept iterations, block means + energies: "
                      f"max split R-hat {diag['rhat_max']:.3f}, min ESS {diag['ess_min']:.0f}")
        if verbose:
            print(f"{alg} posterior mean {name} ({k}x{k} blur, {kind}): SNR {out[name]['snr']:.3f} dB  "
                  f"PSNR {out[name]['psnr']:.3f} dB  MSE {out[name]['mse']:.4f}  [{out[name]['seconds']:.1f} s]")
    out["_observation"] = y
    out["_image"] = img
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--gamma_mc", type=float, default=15.)
    ap.add_argument("--gamma_me", type=float, default=15.)
    ap.add_argument("--sigma", type=float, default=0.75)
    ap.add_argument("--tau", type=float, default=0.3)
    ap.add_argument("--N", type=int, default=1000)
    ap.add_argument("--niter_l2", type=int, default=50)
    ap.add_argument("--niter_tv", type=int, default=10)
    ap.add_argument("--rtol", type=float, default=1e-4, help="early exit of the TV proxes (1e-4: the reference as configured; 0: always all passes)")
    ap.add_argument("--alg", default="ULPDA", choices=["ULPDA", "MYULA", "MYMALA"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--n_chains", type=int, default=None)
    ap.add_argument("--burn_in", type=int, default=0)
    ap.add_argument("--thin", type=int, default=1)
    ap.add_argument("--diagnostics", action="store_true", help="with --n_chains (MYULA, ULPDA): split R-hat / ESS across chains (8x8 block means + energies)")
    ap.add_argument("--size", type=int, default=512, help="side of the synthetic test image")
    ap.add_argument("--image", default=None, help=".npy file with a 2-D grayscale image in [0, 255]")
    ap.add_argument("--models", default=None, help="comma-separated subset of M1..M9")
    ap.add_argument("--out", default=None, help="write posterior means and metrics to this .npz")
    a = ap.parse_args(argv)
    img = np.load(a.image) if a.image else synthetic_image(a.size, a.size)
    res = prox_lmc_deconv(a.gamma_mc, a.gamma_me, a.sigma, a.tau, a.N, a.niter_l2, a.niter_tv, img, a.alg, a.seed, a.n_chains,
                          a.burn_in, a.thin, a.models.split(",") if a.models else None, diagnostics=a.diagnostics, rtol=a.rtol)
    if a.out:
        flat = {}
        for k, v in res.items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    flat[f"{k}_{kk}"] = vv
            else:
                flat[k] = v
        np.savez_compressed(a.out, **flat)


if __name__ == "__main__":
    main()
