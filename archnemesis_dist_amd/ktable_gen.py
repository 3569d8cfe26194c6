"""k-table generator on the GPU: host mirror of Spectroscopy_0.calc_ktable_chunk (Spectroscopy_0.py:3558-3655).

For one chunk of output bins and every (pressure, temperature) of the table the reference (1) sizes a line-by-line grid
from the narrowest Voigt width, (2) computes the absorption coefficient on it (`calc_klbl_online`, the runtime
line-by-line path -- on the GPU once `install_gpu_line_kernel` is active), (3) per bin sorts the coefficients, builds the
weighted cumulative distribution and reads k at the g-ordinates.  Step (3) -- Python loops with an argsort per bin -- is
`AnsfmEngine.kdist_bins` here (one segmented sort for all bins of the chunk); steps (1)-(2) keep the reference's
objects and method names, so the same Spectroscopy / LineData instances drive it."""
import numpy as np

from .forward_model import get_engine


def calc_ktable_chunk(iwaves, Spectroscopy, Spectroscopy_LBL, self_frac, Measurement, device=0, engine=None):
    """Drop-in for Spectroscopy_0.calc_ktable_chunk: -> k_coefficients (len(iwaves), NG, NP, NT)."""
    eng = engine if engine is not None else get_engine(device)
    iwaves = np.asarray(iwaves)
    iwavemin, iwavemax = iwaves[0], iwaves[-1]
    nwave = len(iwaves)
    WAVE = np.asarray(Spectroscopy.WAVE, dtype=np.float64)

    def half_width(iw):                                           # :3566-3568, :3627-3629
        return (Measurement.VFIL[0:Measurement.NFIL[iw], iw] - Measurement.VCONV[iw, 0]).max()

    if Measurement is not None:
        vchunkmin = WAVE[iwavemin] - half_width(iwavemin)
        vchunkmax = WAVE[iwavemax] + half_width(iwavemax)
        vbinmin = np.array([WAVE[iw] - half_width(iw) for iw in iwaves])
        vbinmax = np.array([WAVE[iw] + half_width(iw) for iw in iwaves])
    else:
        delwave = WAVE[1] - WAVE[0]
        vchunkmin = WAVE[iwavemin] - delwave / 2.
        vchunkmax = WAVE[iwavemax] + delwave / 2.
        vbinmin = WAVE[iwaves] - delwave / 2.
        vbinmax = WAVE[iwaves] + delwave / 2.
    vchunkmean = np.mean(WAVE[iwaves])

    linedata = Spectroscopy_LBL.LINE_DATA[0]
    lineparams = Spectroscopy_LBL.LINE_DATA_PARAMS[0]
    ispace = int(Spectroscopy_LBL.ISPACE)
    if ispace == 1:
        wnchunkmin = 1. / vchunkmax * 1.0e4
        wnchunkmax = 1. / vchunkmin * 1.0e4
    else:
        wnchunkmin, wnchunkmax = vchunkmin, vchunkmax
    linedata.set_params(vmin=wnchunkmin - lineparams.wn_approx_window * 2., vmax=wnchunkmax + lineparams.wn_approx_window * 2.,
                        wave_unit=0).fetch_linedata()
    linedata.fetch_partition_fn()
    k_coefficients = np.zeros((nwave, Spectroscopy.NG, Spectroscopy.NP, Spectroscopy.NT))
    if len(linedata.combined_line_data.NU) == 0:
        return k_coefficients

    G_ORD = np.asarray(Spectroscopy.G_ORD, dtype=np.float64)
    fil = None
    if Measurement is not None:                                   # np.interp(delta_wave, VFIL - VCONV, AFIL)  (:3641)
        nfil = np.asarray([Measurement.NFIL[iw] for iw in iwaves], dtype=np.int32)
        NF = int(nfil.max())
        dfil = np.zeros((NF, nwave)); afil = np.zeros((NF, nwave))
        for j, iw in enumerate(iwaves):
            dfil[:nfil[j], j] = Measurement.VFIL[0:nfil[j], iw] - Measurement.VCONV[iw, 0]
            afil[:nfil[j], j] = Measurement.AFIL[0:nfil[j], iw]
        fil = (WAVE[iwaves], nfil, dfil, afil)

    for ip in range(Spectroscopy.NP):
        for it in range(Spectroscopy.NT):
            pressx = Spectroscopy.PRESS[ip]
            tempx = Spectroscopy.TEMP[it]
            alpha_d = linedata.calculate_doppler_width(tempx, combined_output=True)
            gamma_l = linedata.calculate_lorentz_width(tempx, pressx, amb_frac=1. - self_frac, combined_output=True)
            hwhm_voigt = 0.5346 * gamma_l + np.sqrt(0.2166 * gamma_l ** 2. + alpha_d ** 2.)
            delwn_calc = np.min(hwhm_voigt) / 5.
            delv_calc = delwn_calc * (vchunkmean ** 2.) / 1.0e4 if ispace == 1 else delwn_calc
            ncalc = int((vchunkmax - vchunkmin) / delv_calc)
            wavecalc = np.linspace(vchunkmin, vchunkmax, ncalc)
            Spectroscopy_LBL.NWAVE = ncalc
            Spectroscopy_LBL.WAVE = wavecalc
            kabs = Spectroscopy_LBL.calc_klbl_online(1, [pressx], [tempx], amb_frac=1. - self_frac)[:, 0, 0]
            k_coefficients[:, :, ip, it] = eng.kdist_bins(wavecalc, kabs, vbinmin, vbinmax, G_ORD, fil)
    return k_coefficients


def install_gpu_ktable_generator(device=0):
    """Route Spectroscopy_0.calc_ktable_chunk (the worker calc_ktable fans out, :3338-3555) through the function above."""
    import importlib
    sp = importlib.import_module("archnemesis.Spectroscopy_0")
    if not hasattr(sp, "_ansfm_reference_calc_ktable_chunk"):
        sp._ansfm_reference_calc_ktable_chunk = sp.calc_ktable_chunk

    def chunk(iwaves, Spectroscopy, Spectroscopy_LBL, self_frac, Measurement):
        return calc_ktable_chunk(iwaves, Spectroscopy, Spectroscopy_LBL, self_frac, Measurement, device)

    sp.calc_ktable_chunk = chunk
    return chunk
