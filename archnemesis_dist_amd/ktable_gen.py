"""k-table generator on the GPU: host mirror of Spectroscopy_0.calc_ktable_chunk (Spectroscopy_0.py:3558-3655).

For one chunk of output bins and every (pressure, temperature) of the table the reference (1) sizes a line-by-line grid
from the narrowest Voigt width, (2) computes the absorption coefficient on it (`calc_klbl_online`, the runtime
line-by-line path -- on the GPU once `install_gpu_line_kernel` is active), (3) per bin sorts the coefficients, builds the
weighted cumulative distribution and reads k at the g-ordinates.  Step (3) -- Python loops with an argsort per bin -- is
`AnsfmEngine.kdist_bins` here (one segmented sort for all bins of the chunk); steps (1)-(2) keep the reference's
objects and method names, so the same Spectroscopy / LineData instances drive it."""
import numpy as np

from .forward_model import get_engine


def _bin_half_widths(bins, grid, meas):
    """half width of every output bin: half the table spacing, or the reach of the bin's filter beyond its centre"""
    if meas is None:
        return np.full(len(bins), 0.5 * (grid[1] - grid[0]))
    return np.array([np.max(meas.VFIL[:meas.NFIL[b], b] - meas.VCONV[b, 0]) for b in bins])


def _filter_tables(bins, meas):
    """(nfil, offsets, amplitudes) of the per-bin instrument functions, offsets measured from the bin centre"""
    count = np.array([meas.NFIL[b] for b in bins], dtype=np.int32)
    off = np.zeros((int(count.max()), len(bins)))
    amp = np.zeros_like(off)
    for col, b in enumerate(bins):
        n = count[col]
        off[:n, col] = meas.VFIL[:n, b] - meas.VCONV[b, 0]
        amp[:n, col] = meas.AFIL[:n, b]
    return count, off, amp


def calc_ktable_chunk(iwaves, Spectroscopy, Spectroscopy_LBL, self_frac, Measurement, device=0, engine=None):
    """Drop-in for Spectroscopy_0.calc_ktable_chunk (:3558): k-coefficients (len(iwaves), NG, NP, NT) of one chunk of
    output bins.  The host keeps what the reference does per (p, T) -- line-by-line grid step = a fifth of the narrowest
    Voigt half width (:3600-3607), spectrum from `calc_klbl_online` (:3613) -- and hands the per-bin sort / cumulative
    distribution / g-quantile step (:3620-3652) for ALL bins of the chunk to the engine in one call."""
    eng = get_engine(device) if engine is None else engine
    bins = np.asarray(iwaves)
    table, lbl = Spectroscopy, Spectroscopy_LBL
    centres = np.asarray(table.WAVE, dtype=np.float64)[bins]
    half = _bin_half_widths(bins, np.asarray(table.WAVE, dtype=np.float64), Measurement)
    lo, hi = centres - half, centres + half
    span_lo, span_hi = lo[0], hi[-1]                                   # the chunk: first bin's lower to last bin's upper edge
    in_wavelength = int(lbl.ISPACE) == 1
    wn_lo, wn_hi = (1.0e4 / span_hi, 1.0e4 / span_lo) if in_wavelength else (span_lo, span_hi)

    lines, margin = lbl.LINE_DATA[0], 2.0 * lbl.LINE_DATA_PARAMS[0].wn_approx_window
    lines.set_params(vmin=wn_lo - margin, vmax=wn_hi + margin, wave_unit=0).fetch_linedata()
    lines.fetch_partition_fn()
    out = np.zeros((bins.size, table.NG, table.NP, table.NT))
    if len(lines.combined_line_data.NU) == 0:                          # no lines in range: the table stays zero (:3595)
        return out

    g_ord = np.asarray(table.G_ORD, dtype=np.float64)
    fil = None if Measurement is None else (centres,) + _filter_tables(bins, Measurement)
    amb = 1.0 - self_frac
    for ip, it in np.ndindex(table.NP, table.NT):
        p_atm, t_k = table.PRESS[ip], table.TEMP[it]
        doppler = lines.calculate_doppler_width(t_k, combined_output=True)
        lorentz = lines.calculate_lorentz_width(t_k, p_atm, amb_frac=amb, combined_output=True)
        step = np.min(0.5346 * lorentz + np.sqrt(0.2166 * lorentz ** 2. + doppler ** 2.)) / 5.      # Voigt HWHM / 5
        if in_wavelength:
            step = step * (np.mean(centres) ** 2.) / 1.0e4
        npts = int((span_hi - span_lo) / step)
        lbl.NWAVE, lbl.WAVE = npts, np.linspace(span_lo, span_hi, npts)
        kabs = lbl.calc_klbl_online(1, [p_atm], [t_k], amb_frac=amb)[:, 0, 0]
        out[:, :, ip, it] = eng.kdist_bins(lbl.WAVE, kabs, lo, hi, g_ord, fil)
    return out


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
