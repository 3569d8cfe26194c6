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


def gather_columns(local_block, nfm, rank, world_size, group=None, force=False):
    """All-gather of the per-rank spectra blocks.

    local_block: torch tensor (nfm_local, NY) holding forward models chunk_range(nfm, world, rank).
    Returns a (nfm, NY) tensor on every rank (same device as local_block).  One collective; ragged
    chunks are padded to the largest chunk."""
    import torch
    import torch.distributed as dist
    if world_size == 1 and not force:     # force: run the collective anyway (exercises RCCL on a 1-GPU box)
        return local_block
    sizes = [chunk_range(nfm, world_size, r) for r in range(world_size)]
    nmax = max(e - s for s, e in sizes)
    NY = local_block.shape[1]
    pad = torch.zeros((nmax, NY), dtype=local_block.dtype, device=local_block.device)
    pad[: local_block.shape[0]] = local_block
    out = _all_gather(pad, world_size, group)
    parts = [out[r * nmax: r * nmax + (e - s)] for r, (s, e) in enumerate(sizes)]
    return torch.cat(parts, dim=0)


def _all_gather(pad, world_size, group):
    """dist.all_gather_into_tensor of equal blocks along dim 0.  The gloo backend (CPU tests, the one-GPU rehearsal of the
    N > 1 path) gathers host copies of device tensors; nccl (= RCCL) works on the device tensors themselves."""
    import torch
    import torch.distributed as dist
    via_host = pad.is_cuda and dist.get_backend(group) == "gloo"
    src = pad.cpu() if via_host else pad
    out = torch.empty((world_size * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src.contiguous(), group=group)
    return out.to(pad.device) if via_host else out


_PINNED = {}


def _to_host(t):
    """One device-to-host copy of a result matrix through a page-locked buffer kept per shape (a pageable copy of the 16 MB
    KK of a C3 Jacobian takes 3 ms, the pinned one 0.7 ms; the buffer is reused from call to call, the result is a copy)."""
    import torch
    if not t.is_cuda:
        return t.numpy().copy()
    key = (tuple(t.shape), t.dtype)
    buf = _PINNED.get(key)
    if buf is None:
        if len(_PINNED) > 8:
            _PINNED.clear()
        buf = _PINNED[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    buf.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return buf.numpy().copy()


def finite_difference_jacobian_dev(allY, XN, inum, FIX=None):
    """The same quotient on the device: allY torch (nfm, NY), row 0 the unperturbed spectrum -> YN (NY,), KK (NY, NX)
    as NumPy arrays; KK is assembled (transposed, the FIXed and analytic columns left at zero) on the device and crosses
    PCIe once."""
    import torch
    XN = np.asarray(XN, dtype=float)
    inum = np.asarray(inum)
    xn1 = XN[inum] * 1.05
    xn1[xn1 == 0.0] = 0.05
    keep = np.ones(len(inum), bool) if FIX is None else (np.asarray(FIX)[inum] == 0)
    den = torch.as_tensor(xn1 - XN[inum], dtype=allY.dtype, device=allY.device)
    NY = allY.shape[1]
    KK = torch.zeros((NY, XN.shape[0]), dtype=allY.dtype, device=allY.device)
    cols = torch.as_tensor(inum[keep], dtype=torch.long, device=allY.device)
    rows = torch.as_tensor(1 + np.nonzero(keep)[0], dtype=torch.long, device=allY.device)
    KK[:, cols] = ((allY[rows] - allY[0:1]) / den[rows - 1][:, None]).T
    return _to_host(allY[0].contiguous()), _to_host(KK)


def gather_wavenumber_blocks(local_block, ny_local_all, rank, world_size, group=None, force=False):
    """All-gather for the wavenumber-sharded mode: every rank holds ALL nfm forward models on ITS part of the spectral
    axis, local_block (nfm, NY_local); ny_local_all = NY_local of every rank (ragged parts are padded to the largest).
    -> (nfm, sum NY_local) on every rank, the parts side by side in rank order.  One collective."""
    import torch
    import torch.distributed as dist
    if world_size == 1 and not force:
        return local_block
    nmax = int(max(ny_local_all))
    nfm = local_block.shape[0]
    pad = torch.zeros((nfm, nmax), dtype=local_block.dtype, device=local_block.device)
    pad[:, : local_block.shape[1]] = local_block
    out = _all_gather(pad, world_size, group).view(world_size, nfm, nmax)
    return torch.cat([out[r, :, : int(ny_local_all[r])] for r in range(world_size)], dim=1)


def jacobian_nemesis_batched(model, rank=0, world_size=1, group=None, force_collective=False, analytical_gradient=False,
                             shard="states"):
    """jacobian_nemesis (ForwardModel_0.py:2184-2361, numerical part) with every forward model of a rank in ONE batched
    call.  `model` offers the state (`model.state`: XN, NX, NUM, FIX, calc_DSTEP()) and `spectra_batch(X (n, NX)) ->
    torch (n, NY)` on its GPU (profile_state.BatchedCKThermalModel).

    shard = "states": rank r of n takes the reference's contiguous chunk of the nfm = NX_run + 1 forward models
    (:2322-2330).  A rank whose chunk does not start with the unperturbed state puts it in front of its batch all the same:
    the engine shares every layer that is bit-identical to the FIRST state of a batch, and every perturbed state is one
    step away from the unperturbed one, not from its neighbour.  One all_gather of the (nfm_local, NY) blocks (RCCL over
    xGMI when the backend is nccl), then KK on the device and a single copy to the host.

    shard = "wavenumbers": every rank runs ALL nfm states on its own part of the spectral axis -- `model` was built on the
    rank's slice `chunk_range(NWAVE, n, r)` of the k-table (the table is split, not replicated: 1/n of the HBM and of the
    upload per GPU) and returns (nfm, NY_local).  With layer de-duplication on, this is the mode that scales: in the state
    mode every rank recomputes the ~L rows of the unperturbed state next to its 1/n of the ~3 nfm perturbed rows, here no
    row is computed twice anywhere (every wavenumber is independent up to the ILS).  The quotient is formed per
    rank and ONE all_gather of the (nfm, NY_local) blocks puts the parts side by side; `model.ny_local_all(world)` gives
    the split.

    analytical_gradient=True (jacobian_nemesis's own switch, :2262-2289): one forward model with analytic gradients
    (`model.jacobian_analytic`, the nemesisfmg route) instead of nfm forward models.  There is nothing to shard: every
    rank computes the same (YN, KK)."""
    if analytical_gradient:
        if not hasattr(model, "jacobian_analytic"):
            raise NotImplementedError("jacobian_nemesis_batched: this model has no analytic-gradient route")
        return model.jacobian_analytic()
    V = model.state
    V.calc_DSTEP()
    XN = np.array(V.XN, dtype=float)
    xnx = perturbed_states(XN, V.DSTEP)
    inum = np.where((np.ones_like(np.asarray(V.NUM)) == 1) & (np.asarray(V.FIX) == 0))[0]
    nfm = len(inum) + 1
    ixrun = np.zeros(nfm, dtype="int32")
    ixrun[1:nfm] = inum[:] + 1
    if shard == "wavenumbers":
        Y = model.spectra_batch(np.ascontiguousarray(xnx[:, ixrun].T))          # (nfm, NY_local): every state, my wavenumbers
        if Y is None:
            raise RuntimeError(f"Something went wrong when calculating forward models 1-{nfm}/{nfm}.")
        sizes = model.ny_local_all(world_size) if world_size > 1 else [Y.shape[1]]
        allY = gather_wavenumber_blocks(Y, sizes, rank, world_size, group=group, force=force_collective)
        return finite_difference_jacobian_dev(allY, XN, inum, FIX=np.asarray(V.FIX))
    if shard != "states":
        raise ValueError("shard must be 'states' or 'wavenumbers'")
    s, e = chunk_range(nfm, world_size, rank)
    cols = list(ixrun[s:e])
    if e == s:                                      # more ranks than forward models: nothing to compute, an empty block to gather
        import torch
        ny = int(model.ny()) if hasattr(model, "ny") else None
        if ny is None:
            raise RuntimeError("jacobian_nemesis_batched: a rank without forward models needs model.ny() to join the gather")
        dev = model.torch_device() if hasattr(model, "torch_device") else "cpu"
        block = torch.zeros((0, ny), dtype=torch.float64, device=dev)
    else:
        lead = 0 if s == 0 else 1                   # the unperturbed state as the batch's de-duplication reference
        X = xnx[:, [0] * lead + cols].T             # (lead + nfm_local, NX)
        Y = model.spectra_batch(np.ascontiguousarray(X))
        if Y is None:
            raise RuntimeError(f"Something went wrong when calculating forward models {s + 1}-{e}/{nfm}.")        # :2177
        block = Y[lead:]
    allY = gather_columns(block, nfm, rank, world_size, group=group, force=force_collective)
    return finite_difference_jacobian_dev(allY, XN, inum, FIX=np.asarray(V.FIX))


def jacobian_nemesis_sharded(fm, rank=0, world_size=1, device=None, analytical_gradient=False, group=None, **flags):
    """Numerical Jacobian with the forward models sharded over ranks (one process per GPU).

    Mirrors ForwardModel_0.jacobian_nemesis (ForwardModel_0.py:2184-2361) for the numerical part: builds the
    perturbed states (:2234-2242), picks ixrun (:2291-2302), gives rank r the reference's contiguous chunk
    (:2322-2330) -- each forward model is `fm.nemesisfm()` with `fm.Variables.XN` set like execute_fm does
    (:2154), i.e. the reference's host code with CIRSrad on this rank's GPU -- and replaces the sum of
    zero-padded matrices (:2336-2337) by one all_gather.  Returns (YN, KK) on every rank.

    `fm` needs: Variables.{XN, NX, NUM, FIX, calc_DSTEP(), DSTEP}, Measurement.{NY, NGEOM, NCONV}, nemesisfm().
    analytical_gradient=True defers to the reference's own jacobian_nemesis (nemesisfmg path)."""
    import torch
    if hasattr(fm, "spectra_batch"):                # a batched model: one call per rank instead of one per column
        return jacobian_nemesis_batched(fm, rank=rank, world_size=world_size, group=group,
                                        analytical_gradient=analytical_gradient,       # raises when the model has no such route
                                        shard=flags.pop("shard", "states"))
    V, M = fm.Variables, fm.Measurement
    if analytical_gradient:
        return fm.jacobian_nemesis(analytical_gradient=True, **flags)
    V.calc_DSTEP()
    XN = np.array(V.XN, dtype=float)
    xnx = perturbed_states(XN, V.DSTEP)
    inum = np.where((np.ones_like(np.asarray(V.NUM)) == 1) & (np.asarray(V.FIX) == 0))[0]   # NUM[:] = 1 (:2254-2255)
    nfm = len(inum) + 1
    ixrun = np.zeros(nfm, dtype="int32")
    ixrun[1:nfm] = inum[:] + 1
    s, e = chunk_range(nfm, world_size, rank)
    NY = int(M.NY)
    local = np.zeros((e - s, NY))
    for k, ifm in enumerate(range(s, e)):
        V.XN = xnx[:, ixrun[ifm]]
        SPECMOD = fm.nemesisfm()
        if SPECMOD is None:
            raise RuntimeError(f"Something went wrong when calculating forward model {ifm + 1}/{nfm}.")   # :2177
        ik = 0
        for igeom in range(M.NGEOM):
            nc = int(M.NCONV[igeom])
            local[k, ik:ik + nc] = SPECMOD[0:nc, igeom]
            ik += nc
    V.XN = XN
    dev = device if device is not None else "cpu"
    block = torch.as_tensor(local, dtype=torch.float64, device=dev)
    allY = gather_columns(block, nfm, rank, world_size, group=group).cpu().numpy()     # (nfm, NY)
    YN, KK = finite_difference_jacobian(np.ascontiguousarray(allY.T), XN, inum, iYN=0, FIX=np.asarray(V.FIX))
    return YN, KK
