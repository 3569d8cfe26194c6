"""Dense linear algebra of the optimal-estimation step on the GPU.

OptimalEstimation_0.calc_gain_matrix (OptimalEstimation_0.py:545-563), calc_phiret (:573-610), calc_next_xn (:655-677)
and calc_serr (:690-716): after the Jacobian the
retrieval forms  M = KK SA KK^T + SE  (NY x NY, NY up to ~1e4) and solves for the gain matrix -- O(NY^3), the next
bottleneck once the forward models are fast.  Plain library work: float64 GEMMs (rocBLAS) and an LU solve (rocSOLVER)
through torch on the device; nothing hand-written, no CPU path (fails loudly without a GPU)."""
import numpy as np


def _dev(device):
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("oe_linalg needs a HIP device (no CPU fallback)")
    return torch, torch.device("cuda", device)


def calc_gain_matrix(KK, SA, SE, device=0):
    """dd = sa kk^T (kk sa kk^T + se)^-1 by a linear solve, aa = dd kk   (:551-563).  SE may be (1,1) (broadcast, as the
    reference).  Returns DD (NX, NY), AA (NX, NX)."""
    torch, dev = _dev(device)
    f8 = torch.float64
    kk = torch.as_tensor(np.ascontiguousarray(KK, dtype=np.float64), device=dev)
    sa = torch.as_tensor(np.ascontiguousarray(SA, dtype=np.float64), device=dev)
    se = torch.as_tensor(np.ascontiguousarray(SE, dtype=np.float64), device=dev)
    sa_kt = sa @ kk.T                                   # (NX, NY)
    M = kk @ sa_kt + se                                 # (NY, NY)
    X_T = torch.linalg.solve(M.T, sa_kt.T)              # (NY, NX)
    DD = X_T.T.contiguous()
    AA = DD @ kk
    return DD.cpu().numpy(), AA.cpu().numpy()


def calc_serr(DD, AA, SA, SE, simple=False, device=0):
    """sm = dd se dd^T, sn = (aa - I) sa (aa - I)^T, st = sn + sm   (:700-716).  Returns SM, SN, ST."""
    torch, dev = _dev(device)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    dd, aa, sa, se = t(DD), t(AA), t(SA), t(SE)
    a = dd * se[0, 0] if simple else dd @ se
    SM = a @ dd.T
    b = aa.clone()
    b.diagonal().sub_(1.0)
    SN = (b @ sa) @ b.T
    ST = SN + SM
    return SM.cpu().numpy(), SN.cpu().numpy(), ST.cpu().numpy()


def calc_phiret(Y, YN, XN, XA, SE, SA, device=0):
    """Cost function (:573-610): phi = (yn-y)^T SE^-1 (yn-y) + (xn-xa)^T SA^-1 (xn-xa), chisq = first term / NY.
    SE scalar (1,1), diagonal or full, as the reference distinguishes them.  Returns PHI, CHISQ."""
    torch, dev = _dev(device)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    b = t(np.asarray(YN) - np.asarray(Y)); d = t(np.asarray(XN) - np.asarray(XA))
    SE = np.asarray(SE, dtype=np.float64)
    if SE.shape == (1, 1):
        meas = float(torch.dot(b, b)) / float(SE[0, 0])
    else:
        se = t(SE)
        off = se - torch.diag(torch.diagonal(se))
        if not bool(off.any()):
            meas = float(torch.dot(b / torch.diagonal(se), b))
        else:
            meas = float(torch.dot(b, torch.linalg.solve(se, b)))
    apr = float(torch.dot(d, torch.linalg.solve(t(SA), d)))
    return meas + apr, meas / b.numel()


def calc_next_xn(XA, XN, Y, YN, DD, AA, device=0):
    """xn+1 = xa + dd (y - yn) - aa (xa - xn)   (:655-677)."""
    torch, dev = _dev(device)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    xa = t(XA)
    return (xa + t(DD) @ t(np.asarray(Y) - np.asarray(YN)) - t(AA) @ (xa - t(XN))).cpu().numpy()


def install_gpu_oe_linalg(device=0):
    """Route OptimalEstimation_0.calc_gain_matrix / calc_serr / calc_phiret / calc_next_xn through the functions above."""
    import importlib
    oe = importlib.import_module("archnemesis.OptimalEstimation_0")
    cls = oe.OptimalEstimation_0
    if not hasattr(cls, "_ansfm_reference_linalg"):
        cls._ansfm_reference_linalg = (cls.calc_gain_matrix, cls.calc_serr, cls.calc_phiret, cls.calc_next_xn)

    def _calc_gain_matrix(self):
        self.DD, self.AA = calc_gain_matrix(self.KK, self.SA, self.SE, device)

    def _calc_serr(self, simple=False):
        self.SM, self.SN, self.ST = calc_serr(self.DD, self.AA, self.SA, self.SE, simple, device)

    def _calc_phiret(self):
        self.PHI, self.CHISQ = calc_phiret(self.Y[:self.NY], self.YN[:self.NY], self.XN[:self.NX], self.XA[:self.NX], self.SE,
                                           self.SA, device)
        assert not np.isnan(self.PHI), "PHI cannot be NAN"
        assert not np.isnan(self.CHISQ), "CHISQ cannot be NAN"

    def _calc_next_xn(self):
        return calc_next_xn(self.XA, self.XN, self.Y, self.YN, self.DD, self.AA, device)[:self.NX]

    cls.calc_gain_matrix = _calc_gain_matrix
    cls.calc_serr = _calc_serr
    cls.calc_phiret = _calc_phiret
    cls.calc_next_xn = _calc_next_xn
