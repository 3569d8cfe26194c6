"""Sharding of the independent forward models of a numerical Jacobian across GPUs and the
single gather that brings the spectra back.

Reference: ForwardModel_0.jacobian_nemesis (ForwardModel_0.py:2184-2361).  The reference fans the
`nfm` forward models out to joblib workers in contiguous chunks (:2322-2330), each worker returns
a zero-padded (NY,nfm) array and the host sums them (:2336-2337).  Here rank i of n computes the
same contiguous chunk on its own GPU and ONE all_gather of the padded (nfm_max, NY) column blocks
(RCCL over xGMI when the backend is nccl) replaces the sum -- no all-reduce of a zero-padded
matrix, no other data-path collective.
"""
import numpy as np


def chunk_range(nfm, n_jobs, i):
    """[start, end) of worker i -- the reference's chunk arithmetic (ForwardModel_0.py:2322-2330)."""
    base = nfm // n_jobs
    rem = nfm % n_jobs
    return i * base + min(i, rem), (i + 1) * base + min(i + 1, rem)


def perturbed_states(XN, DSTEP):
    """xnx (NX, NX+1): column 0 = XN, column 1+i = XN + DSTEP_i e_i, zeros replaced by 0.05
    (ForwardModel_0.py:2234-2242)."""
    XN = np.asarray(XN, dtype=float)
    NX = XN.shape[0]
    xnx = np.zeros((NX, NX + 1))
    xnx[:, 0] = XN
    xnx[:, 1:] = np.repeat(XN[:, None], NX, axis=1) + np.diag(np.asarray(DSTEP, dtype=float))
    blk = xnx[:, 1:]
    blk[blk == 0] = 0.05
    return xnx


def finite_difference_jacobian(YNtot, XN, inum, iYN=0, FIX=None):
    """KK[:, inum[i]] = (YNtot[:, ifm] - YN) / (1.05*x - x)  (x==0 -> 0.05)   (:2348-2359).
    YNtot (NY, nfm) with column 0 the unperturbed spectrum when iYN == 0."""
    XN = np.asarray(XN, dtype=float)
    NY = YNtot.shape[0]
    KK = np.zeros((NY, XN.shape[0]))
    YN = YNtot[:, 0].copy()
    for i, ix in enumerate(inum):
        ifm = i + 1 if iYN == 0 else i
        xn1 = XN[ix] * 1.05
        if xn1 == 0.0:
            xn1 = 0.05
        if FIX is None or FIX[ix] == 0:
            KK[:, ix] = (YNtot[:, ifm] - YN) / (xn1 - XN[ix])
    return YN, KK


def gather_columns(local_block, nfm, rank, world_size, group=None):
    """All-gather of the per-rank spectra blocks.

    local_block: torch tensor (nfm_local, NY) holding forward models chunk_range(nfm, world, rank).
    Returns a (nfm, NY) tensor on every rank (same device as local_block).  One collective; ragged
    chunks are padded to the largest chunk."""
    import torch
    import torch.distributed as dist
    if world_size == 1:
        return local_block
    sizes = [chunk_range(nfm, world_size, r) for r in range(world_size)]
    nmax = max(e - s for s, e in sizes)
    NY = local_block.shape[1]
    pad = torch.zeros((nmax, NY), dtype=local_block.dtype, device=local_block.device)
    pad[: local_block.shape[0]] = local_block
    out = torch.empty((world_size * nmax, NY), dtype=local_block.dtype, device=local_block.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * nmax: r * nmax + (e - s)] for r, (s, e) in enumerate(sizes)]
    return torch.cat(parts, dim=0)
