"""Multi-GPU host logic: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests).

The per-time-step transform batch is indexed by b = (field, level) and every 2-D transform is
independent, so the batch is partitioned contiguously -- rank r of R owns
b in [r*B/R, (r+1)*B/R) -- with all tables replicated and NO data-path collective.  The only
spectral-space step that couples levels on its own is implicit_terms (implicit.f90:174-216): its inputs
are completed with one all-gather of (divdt, tdt) level slabs (sharded_implicit_terms).  A complete
level-sharded time step has more level couplings (the grid-space tendencies' vertical sums,
get_spectral_tendencies' dmean / sigma-dot, the hydrostatic integration): LevelComm.sharded_step_ is
the device path for it (include/spdy.h: spdy_sharded_step_dev), sharded_step_host its field-by-field
host mirror for the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(nitems, rank, world):
    """Contiguous block partition; ranks differ by at most one item."""
    lo = (nitems * rank) // world
    hi = (nitems * (rank + 1)) // world
    return lo, hi


def shard_sizes(nitems, world):
    return [shard_range(nitems, r, world)[1] - shard_range(nitems, r, world)[0] for r in range(world)]


def max_over_ranks(seconds, device=None):
    """bench.py contract: the job's time is the slowest rank's time."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_levels(local, kx):
    """local: [k_local, nx, mx] complex tensor holding this rank's contiguous level block (shard_range
    over kx).  Returns the full [kx, nx, mx] tensor on every rank.  Ragged blocks (kx % world != 0) are
    padded to the largest block for the collective and trimmed afterwards."""
    world = dist.get_world_size()
    sizes = shard_sizes(kx, world)
    kmax = max(sizes)
    pad = torch.zeros((kmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * kmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    # view_as_real: NCCL/RCCL has no complex dtype; the (re,im) pairs travel as doubles
    if local.is_complex():
        dist.all_gather_into_tensor(torch.view_as_real(out), torch.view_as_real(pad))
    else:
        dist.all_gather_into_tensor(out, pad)
    parts = [out[r * kmax: r * kmax + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


class _DevArray:
    """A plan-owned device buffer as something torch.as_tensor can wrap without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class LocalGroup:
    """Ranks inside this process (include/spdy.h: spdy_comm_group_*): one host thread and one plan per rank, on one or on
    several devices; collectives are peer copies between the ranks' streams, no RCCL.  How a single-process host drives
    several GPUs -- and how the multi-rank paths run on a 1-GPU box."""

    def __init__(self, lib, nranks):
        import ctypes
        from ._lib import check
        self.lib, self.nranks, self.h = lib, nranks, ctypes.c_void_p()
        check(lib.spdy_comm_group_create(nranks, ctypes.byref(self.h)))

    def close(self):
        if self.h:
            from ._lib import check
            check(self.lib.spdy_comm_group_destroy(self.h))
            self.h = None


class LevelComm:
    """The C-ABI communicator (include/spdy.h: spdy_comm_*): direct RCCL collectives on the plan's stream, no torch
    ops, graph-capturable.  One per process/GPU; the RCCL unique id travels over the already initialised
    torch.distributed group (any backend) -- a Fortran/MPI host would MPI_Bcast it instead.
    LevelComm(sp, group=LocalGroup, rank=r): rank r of an in-process group instead."""

    def __init__(self, sp, group=None, rank=None):
        import ctypes
        from ._lib import check
        h = ctypes.c_void_p()
        if group is not None:
            check(sp.lib.spdy_comm_create_local(sp.h, group.h, rank, ctypes.byref(h)))
            self.sp, self.h, self.rank, self.world, self.group = sp, h, rank, group.nranks, group
            return
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        ident = ctypes.create_string_buffer(128)
        if rank == 0:
            check(sp.lib.spdy_comm_unique_id(ident))
        if world > 1:
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0)
            ident = ctypes.create_string_buffer(box[0], 128)
        check(sp.lib.spdy_comm_create(sp.h, world, rank, ident, ctypes.byref(h)))
        self.sp, self.h, self.rank, self.world, self.group = sp, h, rank, world, None

    def level_range(self, nlev):
        return shard_range(nlev, self.rank, self.world)

    def allgather_levels_(self, *full):
        """In place: each tensor is a full [nlev, nx, mx] stack in which this rank has filled its own level block."""
        import ctypes
        from ._lib import check
        self.sp._sync_stream()
        arr = (ctypes.c_void_p * len(full))(*[t.data_ptr() for t in full])
        check(self.sp.lib.spdy_allgather_levels_dev(self.h, full[0].shape[0], len(full), arr))

    def implicit_terms_sharded_(self, divdt_full, tdt_full, psdt):
        from ._lib import check
        self.sp._sync_stream()
        check(self.sp.lib.spdy_implicit_terms_sharded_dev(self.h, divdt_full.data_ptr(), tdt_full.data_ptr(), psdt.data_ptr()))

    # ---- the complete level-sharded time step (spdy_sharded_step_*)
    def sharded_step_workspace(self):
        from ._lib import check
        check(self.sp.lib.spdy_sharded_step_workspace(self.h))

    def sharded_step_(self, vor, div, t, tr, ps, phis, tcorh, qcorh, sdrag, j1, j2, dt, eps, wil, phi, tend=None):
        """One adiabatic step, transforms sharded by level; the prognostics ([2, kx, nx, mx] / [2, nx, mx]) are updated in
        place on every rank; tend ([4 kx + 1, nx, mx], optional) receives vordt | divdt | tdt | trdt | psdt."""
        from ._lib import check
        self.sp._sync_stream()
        check(self.sp.lib.spdy_sharded_step_dev(self.h, vor.data_ptr(), div.data_ptr(), t.data_ptr(), tr.data_ptr(), ps.data_ptr(),
                                                phis.data_ptr(), tcorh.data_ptr(), qcorh.data_ptr(), sdrag, j1, j2, dt, eps, wil,
                                                phi.data_ptr(), tend.data_ptr() if tend is not None else None))

    def sharded_step_grid_(self, vor, div, t, tr, ps, j2):
        from ._lib import check
        self.sp._sync_stream()
        check(self.sp.lib.spdy_sharded_step_grid_dev(self.h, vor.data_ptr(), div.data_ptr(), t.data_ptr(), tr.data_ptr(), ps.data_ptr(), j2))

    def sharded_step_spectral_(self, vor, div, t, tr, ps, phis, tcorh, qcorh, sdrag, j1, dt, eps, wil, phi, tend=None):
        from ._lib import check
        self.sp._sync_stream()
        check(self.sp.lib.spdy_sharded_step_spectral_dev(self.h, vor.data_ptr(), div.data_ptr(), t.data_ptr(), tr.data_ptr(), ps.data_ptr(),
                                                         phis.data_ptr(), tcorh.data_ptr(), qcorh.data_ptr(), sdrag, j1, dt, eps, wil,
                                                         phi.data_ptr(), tend.data_ptr() if tend is not None else None))

    def sharded_step_operands(self):
        """(U, V, PL, lo, hi): this rank's direct-batch operands as torch views of the communicator's workspace -- where a host
        with physics adds its grid-space tendencies between the two halves of the step."""
        import ctypes
        from ._lib import check
        u, v, pl, lo, hi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        check(self.sp.lib.spdy_sharded_step_operands(self.h, ctypes.byref(u), ctypes.byref(v), ctypes.byref(pl), ctypes.byref(lo), ctypes.byref(hi)))
        nl, sp = hi.value - lo.value, self.sp
        dev = "cuda:%d" % sp.device
        mk = lambda ptr, n: torch.as_tensor(_DevArray(ptr.value, (n, sp.il, sp.ix), "<f8"), device=dev)
        return mk(u, 3 * nl), mk(v, 3 * nl), mk(pl, 3 * nl + 1), lo.value, hi.value

    def sharded_step_stacks(self):
        """The two exchanged level-block stacks (grid: 6 kx grids; spectral: 9 kx + nranks spectra) as flat float64 views."""
        import ctypes
        from ._lib import check
        g, gn, t, tn = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_void_p(), ctypes.c_size_t()
        check(self.sp.lib.spdy_sharded_step_stacks(self.h, ctypes.byref(g), ctypes.byref(gn), ctypes.byref(t), ctypes.byref(tn)))
        dev = "cuda:%d" % self.sp.device
        return (torch.as_tensor(_DevArray(g.value, (gn.value,), "<f8"), device=dev),
                torch.as_tensor(_DevArray(t.value, (tn.value,), "<f8"), device=dev))

    # ---- the transposed form (levels <-> point / coefficient ranges; include/spdy.h)
    def set_option(self, name, value):
        """spdy_comm_set_option: "transpose" 0 | 1 selects the form of the sharded step; "force", "dry" as the environment."""
        from ._lib import check
        check(self.sp.lib.spdy_comm_set_option(self.h, name.encode(), int(value)))

    def state_gather_(self, vor, div, t, tr, ps):
        """Transposed form: make the prognostic arrays whole on every rank again (spdy_sharded_state_gather_dev)."""
        from ._lib import check
        self.sp._sync_stream()
        check(self.sp.lib.spdy_sharded_state_gather_dev(self.h, vor.data_ptr(), div.data_ptr(), t.data_ptr(), tr.data_ptr(), ps.data_ptr()))

    def gather_ranges_(self, *arrays):
        """Transposed form: complete arrays of [rows, nx, mx] complex values that a step left current on this rank's coefficients
        only (phi, the final tendencies)."""
        import ctypes
        from ._lib import check
        self.sp._sync_stream()
        arr = (ctypes.c_void_p * len(arrays))(*[a.data_ptr() for a in arrays])
        rows = (ctypes.c_int * len(arrays))(*[int(a.numel() // (self.sp.nx * self.sp.mx)) for a in arrays])
        check(self.sp.lib.spdy_sharded_gather_ranges_dev(self.h, len(arrays), arr, rows))

    def describe(self):
        """spdy_comm_describe as a dict: route / librccl path, ranks, form, this rank's shares, bytes received per step."""
        import ctypes, json
        from ._lib import check
        buf = ctypes.create_string_buffer(2048)
        check(self.sp.lib.spdy_comm_describe(self.h, buf, 2048))
        return json.loads(buf.value.decode())

    def close(self):
        if self.h:
            self.sp.lib.spdy_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the level-block layout of the sharded step (csrc/spdy_kernels.hpp: LevelShard), mirrored for the host ---------------
def level_owner(k, kx, world):
    """Rank that owns level k when rank r owns [kx r / world, kx (r + 1) / world)."""
    return (world * (k + 1) - 1) // kx


def block_slab(f, k, kx, world, nfields, nextra=0):
    """Slab index of field f, level k in a level-block stack of `nfields` fields x kx levels (+ `nextra` level-free slabs per
    rank behind its block): block r starts at nfields * lo_r + nextra * r and is the rank's own [nfields][nl_r] operand."""
    r = level_owner(k, kx, world)
    lo, hi = shard_range(kx, r, world)
    return nfields * lo + nextra * r + f * (hi - lo) + (k - lo)


def allgather_blocks(local, sizes):
    """local: this rank's block (first axis = slabs, sizes[rank] of them); returns the concatenation of all ranks' blocks."""
    world = dist.get_world_size()
    nmax = max(sizes)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype)
    pad[: local.shape[0]] = local
    out = torch.empty((world * nmax,) + tuple(local.shape[1:]), dtype=local.dtype)
    if local.is_complex():
        dist.all_gather_into_tensor(torch.view_as_real(out), torch.view_as_real(pad))
    else:
        dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * nmax: r * nmax + sizes[r]] for r in range(world)], dim=0)


def sharded_step_host(ex, st, rank, world, j1, j2, dt, eps, wil, sdrag, gather=allgather_blocks):
    """The data flow of spdy_sharded_step_dev on the host, field by field, with `ex` as the executor of the single pieces (the
    reference's procedures: uvspec, spec_to_grid, grad, grid_tendencies, vdspec, grid_to_spec, tendency_combine,
    spectral_tendencies, implicit_terms, hdiff_step, step_field) -- what the CPU tests run with their host executor plugged in, over
    gloo.  Rank `rank` transforms ITS levels only; the level-block stacks G (6 kx grids) and T (9 kx + world spectra) are each
    completed by one gather of one contiguous block per rank, exactly as in the C implementation; the column procedures then
    run on full columns.  st: {vor, div, t, tr: [2, kx, nx, mx]; ps: [2, nx, mx]; phis, tcorh, qcorh: [nx, mx]} complex NumPy
    arrays, the full state on every rank.  Returns (new state, final tendencies dict)."""
    import numpy as np
    kx = st["vor"].shape[1]
    lo, hi = shard_range(kx, rank, world)
    nl, lv = hi - lo, j2 - 1
    sizes = shard_sizes(kx, world)
    # 1: inverse batch of the rank's levels -> its block [ug | vg | vorg | divg | tg | trg] (nl each); grad(ps) everywhere
    blk = []
    uv = [ex.uvspec(st["vor"][lv, k], st["div"][lv, k]) for k in range(lo, hi)]
    blk += [ex.spec_to_grid(u, 2) for u, _ in uv] + [ex.spec_to_grid(v, 2) for _, v in uv]
    for n in ("vor", "div", "t", "tr"):
        blk += [ex.spec_to_grid(st[n][lv, k], 1) for k in range(lo, hi)]
    dx, dy = ex.grad(st["ps"][lv])
    px, py = ex.spec_to_grid(dx, 2), ex.spec_to_grid(dy, 2)
    # 2: one gather of the grid stack
    G = gather(torch.from_numpy(np.stack(blk)), [6 * n for n in sizes]).numpy()
    col = lambda f: np.stack([G[block_slab(f, k, kx, world, 6)] for k in range(kx)])
    ug, vg, vorg, divg, tg, trg = (col(f) for f in range(6))
    # 3: grid tendencies on full columns; this rank keeps its own levels' operands
    U, V, PL = ex.grid_tendencies(ug, vg, tg, vorg, divg, trg, px, py)
    own = [g * kx + k for g in range(3) for k in range(lo, hi)]
    Ul, Vl, PLl = U[own], V[own], np.concatenate([PL[own], PL[3 * kx:]])
    # 4: direct batch of the rank's levels (+ the level-free ps tendency) -> its block [pvor | pdiv | pspec (3 nl each)] | psdt
    vd = [ex.vdspec(Ul[i], Vl[i], 2) for i in range(3 * nl)]
    blk = [x[0] for x in vd] + [x[1] for x in vd] + [ex.grid_to_spec(PLl[i]) for i in range(3 * nl + 1)]
    # 5: one gather of the spectral stack
    T = gather(torch.from_numpy(np.stack(blk)), [9 * n + 1 for n in sizes]).numpy()
    stack = lambda f0: np.stack([T[block_slab(f0 + g, k, kx, world, 9, 1)] for g in range(3) for k in range(kx)])
    pvor, pdiv = stack(0), stack(3)
    pspec = np.concatenate([stack(6), T[9 * sizes[0]][None]])                    # psdt: block 0's copy
    # 6: the spectral side on full columns (time level 1 feeds the tendencies and the diffusion)
    pdiv, pspec = ex.tendency_combine(pdiv, pspec)
    vordt, divdt, tdt, trdt, psdt = pvor[:kx], pdiv[:kx], pdiv[kx:2 * kx], pdiv[2 * kx:], pspec[3 * kx]
    divdt, tdt, psdt, phi = ex.spectral_tendencies(st["div"][0], st["t"][0], st["ps"][0], st["phis"], divdt, tdt, psdt)
    divdt, tdt, psdt = ex.implicit_terms(divdt, tdt, psdt)
    vordt, divdt, tdt, trdt = ex.hdiff_step(st["vor"][0], st["div"][0], st["t"][0], st["tr"][0], st["tcorh"], st["qcorh"], sdrag,
                                            vordt, divdt, tdt, trdt)
    new, fin = dict(st), {"phi": phi, "U": Ul, "V": Vl, "PL": PLl}
    new["ps"], fin["psdt"] = ex.step_field(j1, dt, eps, wil, st["ps"], psdt)
    for n, d in (("vor", vordt), ("div", divdt), ("t", tdt), ("tr", trdt)):
        new[n], fin[n + "dt"] = ex.step_field(j1, dt, eps, wil, st[n], d)
    return new, fin


def block_ranges(total, world, unit=16):
    """The transposed form's horizontal ranges (csrc/spdy_api_shard.hip: block_ranges): the `total` points / coefficients in
    blocks of `unit` (the column kernels' block size), rank r owns blocks [nblk r / world, nblk (r + 1) / world)."""
    nblk = (total + unit - 1) // unit
    out = []
    for r in range(world):
        b0, b1 = nblk * r // world, nblk * (r + 1) // world
        out.append((min(total, b0 * unit), min(total, b1 * unit)))
    return out


def exchange_pieces(pieces):
    """pieces[q] = what this rank sends to rank q (NumPy arrays); returns got[q] = what rank q sent here.  (An all-to-all spelled
    with all_gather_object: the CPU tests' gloo backend has no all_to_all, and the volumes are a few hundred KB.)"""
    world, rank = dist.get_world_size(), dist.get_rank()
    box = [None] * world
    dist.all_gather_object(box, pieces)
    return [box[q][rank] for q in range(world)]


def sharded_step_host_transposed(ex, st, rank, world, j1, j2, dt, eps, wil, sdrag, ranges_valid=False, exchange=exchange_pieces):
    """The data flow of the TRANSPOSED form of spdy_sharded_step_dev (include/spdy.h; csrc/spdy_api_shard.hip) on the host, with
    `ex` as the executor of the single procedures, over any torch.distributed backend: levels <-> point ranges around the
    grid-space column procedure, levels <-> coefficient ranges around the spectral side, four exchanges, nothing replicated.
    st: this rank's prognostic arrays ({vor, div, t, tr: [2, kx, nx, mx]; ps: [2, nx, mx]; phis, tcorh, qcorh}); with
    ranges_valid (every call but the first) they are current on (all levels x own coefficients) only and exchange 4 completes,
    in place, what the inverse batch reads.  The column procedures are independent in the horizontal, so the executor runs them
    on whole arrays that are zero outside the rank's range and only the range is kept.
    Returns (state current on (all levels x own coefficients), final tendencies on the own coefficients + this rank's operands)."""
    import numpy as np
    kx, nx, mx = st["vor"].shape[1:]
    lo, hi = shard_range(kx, rank, world)
    nl, lv = hi - lo, j2 - 1
    lev = [shard_range(kx, q, world) for q in range(world)]
    st = {n: np.array(v) for n, v in st.items()}
    flat = lambda a: a.reshape(a.shape[:-2] + (-1,))                            # [.., nx * mx] (or [.., il * ix]) view
    cr = block_ranges(nx * mx, world)
    c0, c1 = cr[rank]
    # exchange 4 (coefficient ranges -> levels) of the previous step, where its result is first read
    if ranges_valid and world > 1:
        for n in ("vor", "div", "t", "tr"):
            f = flat(st[n][lv])
            got = exchange([f[lev[q][0]:lev[q][1], c0:c1].copy() for q in range(world)])
            for q in range(world):
                f[lo:hi, cr[q][0]:cr[q][1]] = got[q]
        f = flat(st["ps"][lv])
        got = exchange([f[c0:c1].copy() for _ in range(world)])
        for q in range(world):
            f[cr[q][0]:cr[q][1]] = got[q]
    # inverse batch of the rank's levels -> its block [ug | vg | vorg | divg | tg | trg] (nl each); grad(ps) everywhere
    blk = []
    uv = [ex.uvspec(st["vor"][lv, k], st["div"][lv, k]) for k in range(lo, hi)]
    blk += [ex.spec_to_grid(u, 2) for u, _ in uv] + [ex.spec_to_grid(v, 2) for _, v in uv]
    for n in ("vor", "div", "t", "tr"):
        blk += [ex.spec_to_grid(st[n][lv, k], 1) for k in range(lo, hi)]
    dx, dy = ex.grad(st["ps"][lv])
    px, py = ex.spec_to_grid(dx, 2), ex.spec_to_grid(dy, 2)
    blk = np.stack(blk)
    il, ix = blk.shape[1:]
    pr = block_ranges(il * ix, world)
    p0, p1 = pr[rank]
    # exchange 1: levels -> point ranges
    got = exchange([flat(blk)[:, pr[q][0]:pr[q][1]].copy() for q in range(world)])
    cols = [np.zeros((kx, il * ix)) for _ in range(6)]                          # ug, vg, vorg, divg, tg, trg: all levels, own points
    for q in range(world):
        nq = lev[q][1] - lev[q][0]
        for f in range(6):
            cols[f][lev[q][0]:lev[q][1], p0:p1] = got[q][f * nq:(f + 1) * nq]
    ug, vg, vorg, divg, tg, trg = (c.reshape(kx, il, ix) for c in cols)
    U, V, PL = ex.grid_tendencies(ug, vg, tg, vorg, divg, trg, px, py)          # (every point is its own column: only [p0, p1) is kept)
    # exchange 2 back: every rank's direct-batch operands (3 nl | 3 nl | 3 nl + 1 grids) come home
    def operands_of(q):
        own = [g * kx + k for g in range(3) for k in range(lev[q][0], lev[q][1])]
        return np.concatenate([flat(U)[own, p0:p1], flat(V)[own, p0:p1], flat(PL)[own, p0:p1], flat(PL)[3 * kx:, p0:p1]])
    got = exchange([operands_of(q) for q in range(world)])
    ops = np.zeros((9 * nl + 1, il * ix))
    for q in range(world):
        ops[:, pr[q][0]:pr[q][1]] = got[q]
    ops = ops.reshape(9 * nl + 1, il, ix)
    Ul, Vl, PLl = ops[:3 * nl], ops[3 * nl:6 * nl], ops[6 * nl:]
    # direct batch of the rank's levels incl. vds (+ the level-free ps tendency) -> [pvor | pdiv | pspec (3 nl each)] | psdt
    vd = [ex.vdspec(Ul[i], Vl[i], 2) for i in range(3 * nl)]
    blk = np.stack([x[0] for x in vd] + [x[1] for x in vd] + [ex.grid_to_spec(PLl[i]) for i in range(3 * nl + 1)])
    # exchange 3: levels -> coefficient ranges
    got = exchange([flat(blk)[:, cr[q][0]:cr[q][1]].copy() for q in range(world)])
    pvor, pdiv, pspec = (np.zeros((3 * kx, nx * mx), complex) for _ in range(3))
    psdt_in = np.zeros(nx * mx, complex)
    for q in range(world):
        nq = lev[q][1] - lev[q][0]
        for g in range(3):
            rows = slice(g * kx + lev[q][0], g * kx + lev[q][1])
            pvor[rows, c0:c1] = got[q][g * nq:(g + 1) * nq]
            pdiv[rows, c0:c1] = got[q][(3 + g) * nq:(4 + g) * nq]
            pspec[rows, c0:c1] = got[q][(6 + g) * nq:(7 + g) * nq]
        if q == 0:
            psdt_in[c0:c1] = got[q][9 * nq]                                     # block 0's copy of the level-free tendency
    sh3 = lambda a: a.reshape(a.shape[0], nx, mx)
    pdiv, pspec = ex.tendency_combine(sh3(pdiv), np.concatenate([sh3(pspec), psdt_in.reshape(1, nx, mx)]))
    vordt, divdt, tdt, trdt, psdt = sh3(pvor)[:kx], pdiv[:kx], pdiv[kx:2 * kx], pdiv[2 * kx:], pspec[3 * kx]
    # the spectral side on all levels; every coefficient is independent: only [c0, c1) is kept
    keep = np.zeros(nx * mx, bool)
    keep[c0:c1] = True
    keep = keep.reshape(nx, mx)
    masked = lambda a: np.where(keep, a, 0.0)
    s1 = {n: masked(st[n]) for n in ("vor", "div", "t", "tr", "ps")}
    divdt, tdt, psdt, phi = ex.spectral_tendencies(s1["div"][0], s1["t"][0], s1["ps"][0], masked(st["phis"]), divdt, tdt, psdt)
    divdt, tdt, psdt = ex.implicit_terms(divdt, tdt, psdt)
    vordt, divdt, tdt, trdt = ex.hdiff_step(s1["vor"][0], s1["div"][0], s1["t"][0], s1["tr"][0], masked(st["tcorh"]), masked(st["qcorh"]), sdrag,
                                            vordt, divdt, tdt, trdt)
    new, fin = dict(st), {"phi": phi, "U": Ul, "V": Vl, "PL": PLl, "range": (c0, c1)}
    upd, fin["psdt"] = ex.step_field(j1, dt, eps, wil, s1["ps"], psdt)
    new["ps"] = np.where(keep, upd, st["ps"])
    for n, d in (("vor", vordt), ("div", divdt), ("t", tdt), ("tr", trdt)):
        upd, fin[n + "dt"] = ex.step_field(j1, dt, eps, wil, s1[n], d)
        new[n] = np.where(keep, upd, st[n])
    return new, fin


def sharded_implicit_terms(sp, divdt_local, tdt_local, psdt, comm=None):
    """Level-sharded semi-implicit correction (implicit.f90:168-217 couples all levels of a coefficient): complete
    the level stacks with one all-gather, run implicit_terms on the full columns (independent per spectral
    coefficient, a few microseconds, done redundantly on every rank), keep this rank's levels.
    sp: speedy_f90_amd.Spectral with initialize_implicit() done; tensors live on sp's device.
    comm: a LevelComm -> the gather is one grouped RCCL call on the same stream as the solve (include/spdy.h);
    without it the gather goes through torch.distributed.  Either way every kernel and collective is ordered on
    torch's current stream (Spectral follows it), so the returned tensors are safe to use from torch."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(sp.kx, rank, world)
    if comm is not None:
        shape = (sp.kx,) + tuple(divdt_local.shape[1:])
        div = torch.empty(shape, dtype=divdt_local.dtype, device=divdt_local.device)
        t = torch.empty(shape, dtype=tdt_local.dtype, device=tdt_local.device)
        div[lo:hi].copy_(divdt_local)
        t[lo:hi].copy_(tdt_local)
        ps = psdt.clone()
        comm.implicit_terms_sharded_(div, t, ps)
        return div[lo:hi], t[lo:hi], ps
    if world > 1:
        div = allgather_levels(divdt_local, sp.kx).contiguous()
        t = allgather_levels(tdt_local, sp.kx).contiguous()
    else:
        div, t = divdt_local.clone(), tdt_local.clone()
    ps = psdt.clone()
    sp.implicit_terms_dev(div, t, ps)
    return div[lo:hi], t[lo:hi], ps
