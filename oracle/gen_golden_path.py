"""Golden fixtures for the path geometry: the REFERENCE's AtmCalc_0 (AtmCalc_0.py:33-478) driven with a
minimal Layer stand-in holding the attributes it reads (build container only)."""
import os
import sys
import importlib
from types import SimpleNamespace
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference  # noqa: E402
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    ans = import_reference()
    AC = importlib.import_module("archnemesis.AtmCalc_0")
    z = np.load(os.path.join(OUT, "layer_average.npz"))
    RADIUS = float(z["RADIUS"]); BASEH = z["split1_BASEH"]; DELH = z["cg_nadir_DELH"]; TEMP = z["cg_nadir_TEMP"]; H = z["H"]
    Layer = SimpleNamespace(RADIUS=RADIUS, BASEH=BASEH, DELH=DELH, TEMP=TEMP, H=H, NLAY=BASEH.size)
    PC = ans.enum.PathCalcEnum
    cases = {"nadir0": dict(path_observer_pointing=1, BOTLAY=0, ANGLE=0.0, EMISS_ANG=0.0, path_calc=PC.THERMAL_EMISSION),
             "nadir40": dict(path_observer_pointing=1, BOTLAY=0, ANGLE=40.0, EMISS_ANG=40.0, path_calc=PC.THERMAL_EMISSION),
             "nadir_up": dict(path_observer_pointing=1, BOTLAY=0, ANGLE=150.0, EMISS_ANG=150.0, path_calc=PC.THERMAL_EMISSION),
             "limb5": dict(path_observer_pointing=0, BOTLAY=5, ANGLE=90.0, EMISS_ANG=90.0, path_calc=PC.THERMAL_EMISSION),
             "nadir_ipzen1": dict(path_observer_pointing=1, BOTLAY=2, ANGLE=30.0, EMISS_ANG=30.0, IPZEN=1, path_calc=PC.THERMAL_EMISSION),
             "nadir_wf": dict(path_observer_pointing=1, BOTLAY=0, ANGLE=10.0, EMISS_ANG=10.0, path_calc=PC.WEIGHTING_FUNCTION)}
    out = dict(RADIUS=RADIUS, BASEH=BASEH, DELH=DELH, TEMP=TEMP, H_top=H[-1], names=np.array(list(cases)))
    for n, kw in cases.items():
        a = AC.AtmCalc_0(Layer, **kw)
        out[n + "_args"] = np.array([kw["path_observer_pointing"], kw["BOTLAY"], kw["ANGLE"], kw["EMISS_ANG"], kw.get("IPZEN", 0), int(kw["path_calc"])], float)
        for k in ("NLAYIN", "LAYINC", "SCALE", "EMTEMP", "IMOD"):
            out[f"{n}_{k}"] = np.asarray(getattr(a, k))
        print(n, a.NPATH, a.NLAYIN)
    np.savez_compressed(os.path.join(OUT, "path_geometry.npz"), **out)


if __name__ == "__main__":
    main()
