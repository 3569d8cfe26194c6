"""One giant line-by-line model split over GPUs by wavenumber ("alternatively split nu into 8 contiguous
ranges (no exchange; line window halo handled by each rank reading lines within +-75 cm-1 of its range)").

`add_line_set_monochromatic_absorption` (LineData_0.py:280-357) gives every grid point the sum, in ascending line order, of
the lines whose +-wn_approx_window reaches it.  A rank that owns the grid points [i0, i1) therefore needs only the lines
within the window of ITS range (plus the largest pressure shift); handing it that sub-list in the same order gives the
same additions in the same order, i.e. the rank's slab is bit-identical to the same rows of the one-GPU result.  The slabs
are brought together by one all_gather of the (layers, points_per_rank) blocks (RCCL over xGMI when the backend is nccl)
-- or left where they are when the next step (ILS convolution per rank) wants them there."""
import numpy as np

from .jacobian import chunk_range


def grid_range(nw, world_size, rank):
    """[i0, i1) of rank's contiguous share of an nw-point grid (the reference's chunk arithmetic, ForwardModel_0.py:2322)."""
    return chunk_range(nw, world_size, rank)


def lines_reaching(nu, wn_lo, wn_hi, wn_approx_window, max_shift=0.0):
    """Indices (ascending, as a slice when nu is sorted) of the lines whose window can reach [wn_lo, wn_hi]."""
    nu = np.asarray(nu)
    reach = float(wn_approx_window) + abs(float(max_shift))
    if nu.size > 1 and np.all(nu[1:] >= nu[:-1]):
        a = int(np.searchsorted(nu, wn_lo - reach, side="left"))
        b = int(np.searchsorted(nu, wn_hi + reach, side="right"))
        return slice(a, b)
    return np.nonzero((nu >= wn_lo - reach) & (nu <= wn_hi + reach))[0]


def max_pressure_shift(broadening_params, p_calc, p_ref):
    """Upper bound of |line_shift| (LineData_0.line_shift :189: sum_b p/p_ref * delta_b * x_b, x_b <= 1)."""
    bp = np.asarray(broadening_params, float)
    delta = np.abs(bp[2::3]) if bp.shape[0] % 3 == 0 else np.abs(bp[2:3])
    return float(np.max(np.atleast_1d(p_calc)) / float(p_ref) * delta.sum(axis=0).max()) if delta.size else 0.0


def add_line_set_sharded(kernel, wn_grid, lineshape_id, t_calc, t_ref, p_calc, p_ref, q_ratio, isotopic_abundance,
                         isotopic_mass, mol_mix_frac, broadening_params, nu, sw, e_lower, stim_ref, rank=0, world_size=1,
                         s_floor=0.0, wn_calc_window=25.0, wn_approx_window=75.0, gather=True, group=None, device=None):
    """This rank's slab of k(nu) for the layers (t_calc, p_calc) -- `kernel` is AnsfmEngine.add_line_set_monochromatic_
    absorption (or any function with its signature) -- and, with gather=True, the whole (layers, nw) array on every rank.

    Returns (out, (i0, i1)): out is (L, nw) when gathered, else the local (L, i1 - i0) slab."""
    wn_grid = np.asarray(wn_grid, float)
    nw = wn_grid.size
    i0, i1 = grid_range(nw, world_size, rank)
    t = np.atleast_1d(np.asarray(t_calc, float))
    L = t.size
    slab = np.zeros((L, i1 - i0))
    if i1 > i0:
        shift = max_pressure_shift(broadening_params, p_calc, p_ref)
        sel = lines_reaching(nu, wn_grid[i0], wn_grid[i1 - 1], wn_approx_window, shift)
        bp = np.asarray(broadening_params)[:, sel]
        if np.asarray(nu)[sel].size:
            kernel(np.ascontiguousarray(wn_grid[i0:i1]), lineshape_id, t, t_ref, np.atleast_1d(p_calc), p_ref,
                   np.atleast_1d(q_ratio), isotopic_abundance, isotopic_mass, mol_mix_frac, np.ascontiguousarray(bp),
                   np.asarray(nu)[sel], np.asarray(sw)[sel], np.asarray(e_lower)[sel], np.asarray(stim_ref)[sel], slab,
                   s_floor=s_floor, wn_calc_window=wn_calc_window, wn_approx_window=wn_approx_window)
    if not gather or world_size == 1:
        return slab, (i0, i1)
    import torch
    import torch.distributed as dist
    sizes = [grid_range(nw, world_size, r) for r in range(world_size)]
    nmax = max(b - a for a, b in sizes)
    dev = device if device is not None else "cpu"
    pad = torch.zeros((L, nmax), dtype=torch.float64, device=dev)
    pad[:, : i1 - i0] = torch.as_tensor(slab, device=dev)
    out = torch.empty((world_size, L, nmax), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out.view(world_size * L, nmax), pad, group=group)
    full = torch.cat([out[r, :, : b - a] for r, (a, b) in enumerate(sizes)], dim=1)
    return full.cpu().numpy(), (i0, i1)
