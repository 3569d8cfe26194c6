"""Golden fixture for the ILS convolution kernels: the REFERENCE's Measurement_0.lblconv / lblconvg / lblconv_fil /
lblconvg_fil (:3335, :3799, :3549, :3992) and their *_ngeom variants (:3444, :3685, :3614, :3912) on a seeded
spectrum, every ISHAPE (incl. the shapes whose result is 0/0).
Build container only.   python oracle/gen_golden_conv.py"""
import os
import sys
import io
import contextlib
import importlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    import_reference()
    M0 = importlib.import_module("archnemesis.Measurement_0")
    rng = np.random.default_rng(99)
    nwave, nx, nconv = 900, 6, 14
    vwave = 2000.0 + 0.01 * np.arange(nwave) + rng.uniform(-2e-3, 2e-3, nwave)
    vwave = np.sort(vwave)
    y = 1e-7 * (1.0 + 0.5 * np.sin(np.arange(nwave) / 17.0)) * rng.uniform(0.9, 1.1, nwave)
    dydx = rng.normal(size=(nwave, nx)) * 10.0 ** rng.uniform(-10, -7, (1, nx))
    vconv = np.linspace(2000.4, 2008.6, nconv)
    vconv[3] = vwave[440]                                  # a convolution point exactly on a grid point (Hamming k = 0)
    fwhm = 0.35
    out = dict(vwave=vwave, y=y, dydx=dydx, vconv=vconv, fwhm=fwhm)
    sink = io.StringIO()                                   # the un-jitted kernels print for every zero weight
    with contextlib.redirect_stdout(sink), np.errstate(all="ignore"):
        for ishape in range(5):
            out[f"conv_{ishape}"] = M0.lblconv(nwave, vwave, y, nconv, vconv, ishape, fwhm)
            yo, go = M0.lblconvg(nwave, vwave, y, dydx, nconv, vconv, ishape, fwhm)
            out[f"convg_{ishape}_y"] = yo; out[f"convg_{ishape}_g"] = go
        # tabulated filters: a different number of points per convolution point, zero-padded columns
        nfil = rng.integers(5, 12, nconv).astype(np.int32)
        NF = int(nfil.max())
        vfil = np.zeros((NF, nconv)); afil = np.zeros((NF, nconv))
        for j in range(nconv):
            half = rng.uniform(0.1, 0.4)
            vfil[:nfil[j], j] = np.linspace(vconv[j] - half, vconv[j] + half, nfil[j])
            a = np.exp(-np.linspace(-2, 2, nfil[j]) ** 2); a[0] = 0.0          # zero weight at one edge
            afil[:nfil[j], j] = a
        out.update(nfil=nfil, vfil=vfil, afil=afil)
        out["fil"] = M0.lblconv_fil(nwave, vwave, y, nconv, vconv, nfil, vfil, afil)
        yo, go = M0.lblconvg_fil(nwave, vwave, y, dydx, nconv, vconv, nfil, vfil, afil)
        out["filg_y"] = yo; out["filg_g"] = go
        # several geometries on one grid (the *_ngeom kernels)
        ngeom = 3
        y2 = np.column_stack([y * s for s in (1.0, 0.7, 1.9)]) * rng.uniform(0.95, 1.05, (nwave, ngeom))
        dydx3 = rng.normal(size=(nwave, ngeom, nx)) * 10.0 ** rng.uniform(-10, -7, (1, 1, nx))
        out.update(y_ngeom=y2, dydx_ngeom=dydx3)
        for ishape in range(5):
            out[f"ngconv_{ishape}"] = M0.lblconv_ngeom(nwave, vwave, y2, nconv, vconv, ishape, fwhm)
            yo, go = M0.lblconvg_ngeom(nwave, vwave, y2, dydx3, nconv, vconv, ishape, fwhm)
            out[f"ngconvg_{ishape}_y"] = yo; out[f"ngconvg_{ishape}_g"] = go
        out["ngfil_y"] = M0.lblconv_fil_ngeom(nwave, vwave, y2, nconv, vconv, nfil, vfil, afil)
        yo, go = M0.lblconvg_fil_ngeom(nwave, vwave, y2, dydx3, nconv, vconv, nfil, vfil, afil)
        out["ngfilg_y"] = yo; out["ngfilg_g"] = go
        # k-table methods Measurement_0.conv / convg, FWHM < 0 (filter per convolution point, bracketing window) and
        # FWHM == 0 (plain interpolation); afil[0] > 0 for some filters so the bracketing points carry weight
        Meas = M0.Measurement_0()
        Meas.NGEOM = 1; Meas.FWHM = -1.0; Meas.NCONV = np.array([nconv]); Meas.VCONV = vconv[:, None].copy()
        afil2 = afil.copy(); afil2[0, ::2] = 0.3
        Meas.NFIL = nfil; Meas.VFIL = vfil; Meas.AFIL = afil2
        out["afil_k"] = afil2
        out["kconv_fil"] = Meas.conv(vwave, y, IGEOM=0)
        yo, go = Meas.convg(vwave, y, dydx, IGEOM=0)
        out["kconvg_fil_y"] = yo; out["kconvg_fil_g"] = go
        Meas.FWHM = 0.0
        out["kconv_0"] = Meas.conv(vwave, y, IGEOM=0)
        yo, go = Meas.convg(vwave, y, dydx, IGEOM=0)
        out["kconvg_0_y"] = yo; out["kconvg_0_g"] = go
        # filter integrals (no normalisation): integrate_filter / integrate_filterg and the *_ngeom variants
        out["intf"] = M0.integrate_filter(nwave, vwave, y, nconv, vconv, nfil, vfil, afil)
        yo, go = M0.integrate_filterg(nwave, vwave, y, dydx, nconv, vconv, nfil, vfil, afil)
        out["intfg_y"] = yo; out["intfg_g"] = go
        out["ngintf"] = M0.integrate_filter_ngeom(nwave, vwave, y2, nconv, vconv, nfil, vfil, afil)
        yo, go = M0.integrate_filterg_ngeom(nwave, vwave, y2, dydx3, nconv, vconv, nfil, vfil, afil)
        out["ngintfg_y"] = yo; out["ngintfg_g"] = go
    np.savez_compressed(os.path.join(OUT, "ils_conv.npz"), **out)
    print({k: (np.shape(v), bool(np.isnan(v).any())) for k, v in out.items() if k.startswith(("conv", "fil", "ngconv", "ngfil", "kconv"))})


if __name__ == "__main__":
    main()
