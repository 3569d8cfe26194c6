"""The reference's own known-answer test for the layering step (tests/test_layer_class.py:12-157): Jupiter_test_layer
inputs -> subprofretg() -> calc_path(), compared there with literal arrays from the Fortran NEMESIS code at rtol 1e-2.
This script (build container only) runs that flow with the reference -- the input directory is copied to a temp dir and
its k-table list pointed at small synthetic .kta files written by the reference's write_ktable, because the real tables
are absent and read_input_files needs their headers -- records the arguments and results of the Layer_0.layer_split /
layer_average calls it makes, and takes the NEMESIS literal arrays out of the test's source with `ast` (data, not code).
    python oracle/gen_golden_nemesis_layers.py    -> tests/golden/nemesis_layers.npz"""
import ast
import os
import shutil
import sys
import tempfile
import importlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference, REFERENCE_ROOT  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def literals(path, func):
    """name -> ndarray for every `NAME = np.array([...])` assignment inside `func` of the test module."""
    tree = ast.parse(open(path).read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == func:
            for st in ast.walk(node):
                if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Name) and \
                        isinstance(st.value, ast.Call) and getattr(st.value.func, "attr", "") == "array":
                    out[st.targets[0].id] = np.array(ast.literal_eval(st.value.args[0]), dtype=float)
    return out


def main():
    ans = import_reference()
    sp_mod = sys.modules["archnemesis.Spectroscopy_0"]
    l0 = importlib.import_module("archnemesis.Layer_0")
    src = os.path.join(REFERENCE_ROOT, "tests", "files", "Jupiter_test_layer")
    work = tempfile.mkdtemp(prefix="ansfm_lay_")
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), os.path.join(work, f))
        os.chmod(os.path.join(work, f), 0o644)
    x, w = np.polynomial.legendre.leggauss(4)
    fn = os.path.join(work, "c2h2_synth.kta")
    sp_mod.write_ktable(fn, 26, 0, 0.5 * (x + 1), 0.5 * w, np.logspace(-6, 1, 4), np.linspace(80., 300., 3), 5, 600.0, 2.5, 0.0,
                        np.full((5, 4, 4, 3), 1e-22))
    with open(os.path.join(work, "cirstest.kls"), "w") as f:
        f.write(fn + "\n")
    rec = {}
    orig_split, orig_avg = l0.layer_split, l0.layer_average

    def split(*a, **k):
        r = orig_split(*a, **k)
        rec["split"] = (a, k, r)
        return r

    def avg(*a, **k):
        r = orig_avg(*a, **k)
        rec["avg"] = (a, k, r)
        return r

    l0.layer_split, l0.layer_average = split, avg
    cwd = os.getcwd()
    os.chdir(work)
    try:
        Atm, Meas, Spec, Scat, Stel, Surf, CIA, Lay, Var, Ret = ans.Files.read_input_files("cirstest")
        fm = ans.ForwardModel_0(runname="cirstest", Atmosphere=Atm, Surface=Surf, Measurement=Meas, Spectroscopy=Spec, Stellar=Stel,
                                Scatter=Scat, CIA=CIA, Layer=Lay, Variables=Var)
        fm.subprofretg()
        fm.calc_path()
    finally:
        os.chdir(cwd)
        l0.layer_split, l0.layer_average = orig_split, orig_avg
    out = {}
    names_s = ["RADIUS", "H", "P"]
    a, k, r = rec["split"]
    for n, v in zip(names_s, a):
        out["split_" + n] = np.asarray(v, float)
    for n, v in k.items():
        out["split_kw_" + n] = np.asarray(-1 if v is None else v, float)
    out["split_BASEH"], out["split_BASEP"] = (np.asarray(v, float) for v in r)
    a, k, r = rec["avg"]
    allk = dict(zip(["RADIUS", "H", "P", "T", "ID", "VMR", "DUST", "PARAH2", "BASEH", "BASEP"], a)); allk.update(k)
    for n, v in allk.items():
        if v is not None:
            out["avg_" + n] = np.asarray(v, float)
    for n, v in zip(["HEIGHT", "PRESS", "TEMP", "TOTAM", "AMOUNT", "PP", "CONT", "FRAC", "DELH", "BASET", "LAYSF"], r):
        out["ref_" + n] = np.asarray(v, float)
    for n, v in literals(os.path.join(REFERENCE_ROOT, "tests", "test_layer_class.py"), "test_layer_nemesis_units").items():
        out["nemesis_" + n] = v
    np.savez_compressed(os.path.join(OUT, "nemesis_layers.npz"), **out)
    print({k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
