"""Stand-ins that drive calc_ktable_chunk (the reference's, in oracle/gen_golden_ktable.py, and the GPU mirror in the
tests) without a line database: a Spectroscopy_LBL whose calc_klbl_online returns an analytic spectrum, a LineData with
the width methods the chunk function calls, and optionally a Measurement with one filter per bin.  Plain NumPy."""
from types import SimpleNamespace
import numpy as np


class FakeLineData:
    max_lines_or_bins = 8

    def __init__(self, centres):
        self.combined_line_data = SimpleNamespace(NU=np.asarray(centres, float))

    def set_params(self, **kw):
        self.params = kw
        return self

    def fetch_linedata(self):
        return self

    def fetch_partition_fn(self):
        return None

    def calculate_doppler_width(self, temp, combined_output=True):
        return 3.0e-3 * np.sqrt(temp / 200.0) * np.ones(self.combined_line_data.NU.size)

    def calculate_lorentz_width(self, temp, press, amb_frac=1.0, combined_output=True):
        return 0.07 * press * (296.0 / temp) ** 0.7 * (0.9 + 0.1 * amb_frac) * np.ones(self.combined_line_data.NU.size)


class FakeSpectroscopyLBL:
    """calc_klbl_online(npoints, press, temp, amb_frac) -> (NWAVE, npoints, 1): a sum of Lorentzians on self.WAVE"""
    ISPACE = 0

    def __init__(self, centres, strengths):
        self.centres = np.asarray(centres, float); self.strengths = np.asarray(strengths, float)
        self.LINE_DATA = [FakeLineData(centres)]
        self.LINE_DATA_PARAMS = [SimpleNamespace(wn_approx_window=25.0, s_min=0.0)]
        self.NWAVE = 0; self.WAVE = None

    def calc_klbl_online(self, npoints, press, temp, amb_frac=1.0):
        p, t = float(press[0]), float(temp[0])
        g = 0.07 * p * (296.0 / t) ** 0.7 + 3.0e-3 * np.sqrt(t / 200.0)
        w = np.asarray(self.WAVE, float)
        k = np.zeros_like(w) + 1.0e-27 * (1.0 + 0.01 * (w - w[0]))
        for c, s in zip(self.centres, self.strengths):
            k = k + s * (t / 296.0) ** -1.5 * (g / np.pi) / ((w - c) ** 2 + g * g)
        return k[:, None, None]


def make_case(with_filter, seed=11):
    rng = np.random.default_rng(seed)
    NB, NG, NP, NT = 7, 10, 3, 2
    x, _ = np.polynomial.legendre.leggauss(NG)
    S = SimpleNamespace(WAVE=1200.0 + 0.5 * np.arange(NB), NG=NG, NP=NP, NT=NT, G_ORD=0.5 * (x + 1.0),
                        PRESS=np.array([3.0e-3, 0.1, 1.5]), TEMP=np.array([150.0, 290.0]))
    centres = np.sort(rng.uniform(1199.5, 1203.5, 40)); strengths = 10.0 ** rng.uniform(-24, -20, 40)
    L = FakeSpectroscopyLBL(centres, strengths)
    M = None
    if with_filter:
        nfil = rng.integers(4, 9, NB)
        NF = int(nfil.max())
        VFIL = np.zeros((NF, NB)); AFIL = np.zeros((NF, NB))
        for j in range(NB):
            half = rng.uniform(0.3, 0.6)
            VFIL[:nfil[j], j] = S.WAVE[j] + np.linspace(-half, half, nfil[j])
            AFIL[:nfil[j], j] = np.exp(-np.linspace(-1.5, 1.5, nfil[j]) ** 2)
        M = SimpleNamespace(NFIL=nfil, VFIL=VFIL, AFIL=AFIL, VCONV=S.WAVE[:, None].copy())
    return np.arange(1, NB - 1), S, L, 0.15, M
