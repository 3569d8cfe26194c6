"""Rebuild the objects CIRSrad reads (SpectroscopyX, LayerX, PathX, ...) from the C1 golden fixture
(tests/golden/c1_cirsrad.npz: the reference's nemesisfm on the Jupiter CIRS nadir inputs with
synthetic k-tables, every 9th wavenumber kept)."""
import os
from types import SimpleNamespace

import numpy as np


class _Atm(SimpleNamespace):
    def locate_gas(self, gid, iso):
        return int(self._igas[(int(gid), int(iso))])


def load_c1(golden_dir, name="c1_cirsrad.npz"):
    z = np.load(os.path.join(golden_dir, name))
    W, G, NP, NT, S = z["K"].shape
    L = z["LAY_PRESS"].shape[0]
    spec = SimpleNamespace(NWAVE=W, NG=G, NP=NP, NT=NT, NGAS=S, WAVE=z["WAVE"], K=z["K"], PRESS=z["TPRESS"],
                           TEMP=z["TTEMP"], DELG=z["DELG"], G_ORD=z["G_ORD"], ID=z["ID"], ISO=z["ISO"],
                           ILBL=int(z["ILBL"]))
    layer = SimpleNamespace(NLAY=L, PRESS=z["LAY_PRESS"], TEMP=z["LAY_TEMP"], AMOUNT=z["LAY_AMOUNT"],
                            TOTAM=z["LAY_TOTAM"], TAUCIA=z["TAUCIA"], TAURAY=z["TAURAY"], TAUDUST=z["TAUDUST"])
    if "dTAUCON" in z.files:
        layer.dTAUCON = z["dTAUCON"]
    path = SimpleNamespace(NPATH=z["LAYINC"].shape[1], NLAYIN=z["NLAYIN"], LAYINC=z["LAYINC"], SCALE=z["SCALE"],
                           EMTEMP=z["EMTEMP"], IMOD=z["IMOD"], SOL_ANG=z["SOL_ANG"], EMISS_ANG=z["EMISS_ANG"])
    igas = {(int(i), int(s)): int(g) for i, s, g in zip(z["ID"], z["ISO"], z["IGAS"])}
    atm = _Atm(NVMR=int(z["NVMR"]), _igas=igas, RADIUS=0.0)
    surf = SimpleNamespace(TSURF=float(z["TSURF"]), GASGIANT=bool(z["GASGIANT"]), LOWBC=int(z["LOWBC"]), VEM=None,
                           EMISSIVITY=None)
    meas = SimpleNamespace(IFORM=int(z["IFORM"]), ISPACE=int(z["ISPACE"]))
    scat = SimpleNamespace(NDUST=int(z["NDUST"]))
    stel = SimpleNamespace(SOLEXIST=bool(z["SOLEXIST"]))
    return z, dict(SpectroscopyX=spec, LayerX=layer, PathX=path, AtmosphereX=atm, SurfaceX=surf, MeasurementX=meas,
                   ScatterX=scat, StellarX=stel, CIAX=None, EmissionsX=None)
