"""CPU twin of archnemesis_dist_amd.profile_state.BatchedCKThermalModel -- TEST INFRASTRUCTURE ONLY.

The same chain nemesisfm runs for one state (ForwardModel_0.py:437-589: subprofretg -> calc_path -> CIRSrad), one state
at a time through the oracle's restatements (oracle.layer_average, oracle.calc_tau_rayleigh, oracle.cirsrad_ck_thermal),
so that a numerical Jacobian from the engine can be held against one built the reference's way (jacobian_nemesis
:2305-2359: nfm forward models, KK = (Y_i - Y_0) / (1.05 x_i - x_i)).  Only tests/, bench.py's checker leg and
__graft_entry__.smoke() import this.  The host-side layer grid / ray geometry (archnemesis_dist_amd.layering) is shared
with the product: it is pinned on its own by tests/golden/path_geometry.npz and layer_average.npz."""
import numpy as np

from . import oracle as orc


def spectra(model, K, TPRESS, TTEMP, WAVE, DELG, X):
    """Spectra (n, NY) of the states X (n, NX) for `model` (a BatchedCKThermalModel; only its configuration is read)."""
    from archnemesis_dist_amd import layering
    st, la, ge = model.state, model.lay, model.geo
    X = np.atleast_2d(X)
    out = []
    for x in X:
        T, VMR = st.profiles(x[None])
        T, VMR = T[0], VMR[0]
        (HEIGHT, PRESS, TEMP, TOTAM, AMOUNT, PP, CONT, FRAC, DELH, BASET, LAYSF) = orc.layer_average(
            model.RADIUS, st.H, st.P, T, model.ID, VMR, None, None, model.BASEH, model.BASEP, LAYANG=la["LAYANG"],
            LAYINT=la["LAYINT"], LAYHT=la["LAYHT"], NINT=la["NINT"])
        path = layering.calc_path(model.RADIUS, model.BASEH, DELH, TEMP, float(st.H[-1]), pointing=ge["pointing"],
                                  BOTLAY=ge["BOTLAY"], ANGLE=ge["ANGLE"], EMISS_ANG=ge["EMISS_ANG"], IPZEN=ge["IPZEN"])
        amount = AMOUNT[:, model.igas_map].T * 1.0e-4
        cont = np.zeros((WAVE.size, PRESS.size))
        if model.IRAY != 0:
            cont = cont + orc.calc_tau_rayleigh(model.IRAY, model.ISPACE, WAVE, TOTAM, ID=model.ID, ISO=model.ISO,
                                                VMR=PP / PRESS[:, None])[0]
        if model.extra is not None:
            cont = cont + model.extra
        spec = orc.cirsrad_ck_thermal(model.ISPACE, K, TPRESS, TTEMP, WAVE, DELG, PRESS, TEMP, amount, cont, path.NLAYIN,
                                      path.LAYINC, path.SCALE, path.EMTEMP, model.TSURF)
        out.append(spec.reshape(-1))
    return np.stack(out)


def jacobian(model, K, TPRESS, TTEMP, WAVE, DELG, columns=None):
    """(YN, KK[:, columns]) the reference's way; columns = state-vector elements to perturb (default: all)."""
    from archnemesis_dist_amd.jacobian import perturbed_states
    st = model.state
    st.calc_DSTEP()
    XN = np.array(st.XN, float)
    xnx = perturbed_states(XN, st.DSTEP)
    cols = np.arange(st.NX) if columns is None else np.asarray(columns)
    Y = spectra(model, K, TPRESS, TTEMP, WAVE, DELG, xnx[:, np.concatenate([[0], cols + 1])].T)
    xn1 = XN[cols] * 1.05
    xn1[xn1 == 0.0] = 0.05
    KK = ((Y[1:] - Y[0:1]) / (xn1 - XN[cols])[:, None]).T
    return Y[0], KK
