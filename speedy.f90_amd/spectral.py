"""Python mirror of the reference's `spectral` module (spectral.f90:8-11) over the C-ABI.

Same public names, argument meaning and result shapes as the Fortran module, plus batched
and device-resident variants.  Arrays are NumPy C-order views of the reference's column-major
arrays (axes reversed):

    grid  vorg(ix,il)          -> float64    [..., il, ix]
    spec  vorm(mx,nx) complex  -> complex128 [..., nx, mx]
    four  (2*mx,il)            -> float64    [..., il, 2*mx]

Leading dimensions are the batch (e.g. the kx levels of a (mx,nx,kx) array).  Host methods
accept/return NumPy arrays; *_dev methods take torch CUDA tensors (device memory stays where
it is; kernels run on torch's current stream unless use_own_stream() was called).
"""
import ctypes

import numpy as np

from ._lib import check, load

RESOLUTIONS = {"t30": (30, 96, 24), "t63": (63, 192, 48)}   # trunc, ix, iy


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Graph:
    """A captured sequence of device-resident calls (spdy_graph_* in include/spdy.h)."""

    def __init__(self, lib, handle, sp=None):
        self.lib, self.h, self.sp = lib, handle, sp

    def launch(self):
        """Replays on the plan's stream -- after re-pointing it at torch's current stream when the plan follows torch
        (the default), so that the replay is ordered with the surrounding torch ops like every ``*_dev`` call."""
        if self.sp is not None:
            self.sp._sync_stream()
        check(self.lib.spdy_graph_launch(self.h))

    def num_nodes(self):
        """Nodes of the captured graph: one per kernel launch / collective (spdy_graph_num_nodes)."""
        n = ctypes.c_int(0)
        check(self.lib.spdy_graph_num_nodes(self.h, ctypes.byref(n)))
        return n.value

    def close(self):
        if self.h:
            self.lib.spdy_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _GraphCapture:
    def __init__(self, sp):
        self.sp, self.graph = sp, Graph(sp.lib, None, sp)

    def __enter__(self):
        check(self.sp.lib.spdy_graph_begin(self.sp.h))
        self.sp._capturing = True
        return self.graph

    def __exit__(self, exc_type, exc, tb):
        h = ctypes.c_void_p()
        self.sp._capturing = False
        rc = self.sp.lib.spdy_graph_end(self.sp.h, ctypes.byref(h))
        if exc_type is None:
            check(rc)
            self.graph.h = h
        elif rc == 0:
            self.sp.lib.spdy_graph_destroy(h)
        return False


class Spectral:
    """One transform plan = the module state `initialize_spectral` builds (spectral.f90:20)."""

    def __init__(self, res="t30", kx=8, max_batch=64, device=0):
        trunc, ix, iy = RESOLUTIONS[res] if isinstance(res, str) else res
        self.lib = load()
        h = ctypes.c_void_p()
        check(self.lib.spdy_plan_create(trunc, ix, iy, kx, max_batch, device, ctypes.byref(h)))
        self.h = h
        self.trunc, self.ix, self.iy, self.il, self.kx = trunc, ix, iy, 2 * iy, kx
        self.nx, self.mx, self.max_batch, self.device = trunc + 2, trunc + 1, max_batch, device
        self.grid_shape, self.spec_shape = (self.il, self.ix), (self.nx, self.mx)
        self.four_shape = (self.il, 2 * self.mx)

    def close(self):
        if getattr(self, "h", None):
            self.lib.spdy_plan_destroy(self.h)
            self.h = None

    __del__ = close

    # ------------------------------------------------------------------ helpers
    def _in(self, a, shape, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        if a.shape[-len(shape):] != tuple(shape):
            raise ValueError("expected trailing shape %s, got %s" % (shape, a.shape))
        lead = a.shape[:-len(shape)]
        nb = int(np.prod(lead)) if lead else 1
        return a, lead, nb

    def table(self, name):
        n = check(self.lib.spdy_get_table(self.h, name.encode(), None, 0))
        out = np.zeros(n)
        check(self.lib.spdy_get_table(self.h, name.encode(), _p(out), n))
        return out

    @property
    def el2(self):
        """Public table of the reference module (spectral.f90:8)."""
        return self.table("el2").reshape(self.spec_shape)

    # ------------------------------------------------------------------ transforms (host arrays)
    def spec_to_grid(self, vorm, kcos=1):
        """spectral.f90:98 -- kcos: int or per-field sequence."""
        s, lead, nb = self._in(vorm, self.spec_shape, np.complex128)
        g = np.empty(lead + self.grid_shape)
        kc = np.ascontiguousarray(np.broadcast_to(np.asarray(kcos, np.int32), (nb,)))
        check(self.lib.spdy_spec_to_grid_batch(self.h, nb, _p(s), _p(kc), _p(g)))
        return g

    def grid_to_spec(self, vorg):
        """spectral.f90:112"""
        g, lead, nb = self._in(vorg, self.grid_shape, np.float64)
        s = np.empty(lead + self.spec_shape, np.complex128)
        check(self.lib.spdy_grid_to_spec_batch(self.h, nb, _p(g), _p(s)))
        return s

    def legendre_inv(self, s):
        s, lead, nb = self._in(s, self.spec_shape, np.complex128)
        f = np.empty(lead + self.four_shape)
        check(self.lib.spdy_legendre_inv(self.h, nb, _p(s), _p(f)))
        return f

    def legendre_dir(self, f):
        f, lead, nb = self._in(f, self.four_shape, np.float64)
        s = np.empty(lead + self.spec_shape, np.complex128)
        check(self.lib.spdy_legendre_dir(self.h, nb, _p(f), _p(s)))
        return s

    def fourier_inv(self, f, kcos=1):
        f, lead, nb = self._in(f, self.four_shape, np.float64)
        g = np.empty(lead + self.grid_shape)
        check(self.lib.spdy_fourier_inv(self.h, nb, _p(f), int(kcos), _p(g)))
        return g

    def fourier_dir(self, g):
        g, lead, nb = self._in(g, self.grid_shape, np.float64)
        f = np.empty(lead + self.four_shape)
        check(self.lib.spdy_fourier_dir(self.h, nb, _p(g), _p(f)))
        return f

    # ------------------------------------------------------------------ spectral operators
    def laplacian(self, a):
        a, lead, nb = self._in(a, self.spec_shape, np.complex128)
        o = np.empty_like(a)
        check(self.lib.spdy_laplacian(self.h, nb, _p(a), _p(o)))
        return o

    def inverse_laplacian(self, a):
        a, lead, nb = self._in(a, self.spec_shape, np.complex128)
        o = np.empty_like(a)
        check(self.lib.spdy_inverse_laplacian(self.h, nb, _p(a), _p(o)))
        return o

    def trunct(self, a):
        a, lead, nb = self._in(a, self.spec_shape, np.complex128)
        a = a.copy()
        check(self.lib.spdy_trunct(self.h, nb, _p(a)))
        return a

    def grad(self, psi):
        psi, lead, nb = self._in(psi, self.spec_shape, np.complex128)
        dx, dy = np.zeros_like(psi), np.zeros_like(psi)
        check(self.lib.spdy_grad(self.h, nb, _p(psi), _p(dx), _p(dy)))
        return dx, dy

    def vds(self, ucosm, vcosm):
        u, lead, nb = self._in(ucosm, self.spec_shape, np.complex128)
        v, _, _ = self._in(vcosm, self.spec_shape, np.complex128)
        vor, div = np.zeros_like(u), np.zeros_like(u)
        check(self.lib.spdy_vds(self.h, nb, _p(u), _p(v), _p(vor), _p(div)))
        return vor, div

    def uvspec(self, vorm, divm):
        a, lead, nb = self._in(vorm, self.spec_shape, np.complex128)
        b, _, _ = self._in(divm, self.spec_shape, np.complex128)
        u, v = np.zeros_like(a), np.zeros_like(a)
        check(self.lib.spdy_uvspec(self.h, nb, _p(a), _p(b), _p(u), _p(v)))
        return u, v

    def vdspec(self, ug, vg, kcos=2):
        ug, lead, nb = self._in(ug, self.grid_shape, np.float64)
        vg, _, _ = self._in(vg, self.grid_shape, np.float64)
        vor = np.zeros(lead + self.spec_shape, np.complex128)
        div = np.zeros_like(vor)
        check(self.lib.spdy_vdspec(self.h, nb, _p(ug), _p(vg), _p(vor), _p(div), int(kcos)))
        return vor, div

    def uvspec_to_grid(self, vorm, divm, kcos=2):
        """uvspec + spec_to_grid(., kcos) of both results in one call (tendencies.f90:98-100); leading batch dims allowed."""
        vorm, lead, nb = self._in(vorm, self.spec_shape, np.complex128)
        divm, _, _ = self._in(divm, self.spec_shape, np.complex128)
        ug, vg = np.zeros(lead + self.grid_shape), np.zeros(lead + self.grid_shape)
        check(self.lib.spdy_uvspec_to_grid(self.h, nb, _p(vorm), _p(divm), _p(ug), _p(vg), int(kcos)))
        return ug, vg

    def grad_to_grid(self, psi, kcos=2):
        """grad + spec_to_grid(., kcos) of both results in one call (tendencies.f90:121-123)."""
        psi, lead, nb = self._in(psi, self.spec_shape, np.complex128)
        gx, gy = np.zeros(lead + self.grid_shape), np.zeros(lead + self.grid_shape)
        check(self.lib.spdy_grad_to_grid(self.h, nb, _p(psi), _p(gx), _p(gy), int(kcos)))
        return gx, gy

    # ------------------------------------------------------------------ spectral-space tail
    def do_horizontal_diffusion(self, field, fdt_in, dmp, dmp1):
        """horizontal_diffusion.f90:86-105 (2-D or 3-D by the leading dimension)."""
        field, lead, nlev = self._in(field, self.spec_shape, np.complex128)
        fdt_in, _, _ = self._in(fdt_in, self.spec_shape, np.complex128)
        dmp = np.ascontiguousarray(dmp, np.float64)
        dmp1 = np.ascontiguousarray(dmp1, np.float64)
        out = np.empty_like(field)
        check(self.lib.spdy_hdiff(self.h, nlev, _p(field), _p(fdt_in), _p(dmp), _p(dmp1), _p(out)))
        return out

    def initialize_implicit(self, dt):
        """implicit.f90:36"""
        check(self.lib.spdy_implicit_init(self.h, float(dt)))

    def set_sigma(self, hsg):
        """Half levels hsg[kx+1] for a level count the reference defines no set for (geometry.f90:42-48)."""
        hsg = np.ascontiguousarray(hsg, np.float64)
        if hsg.shape != (self.kx + 1,):
            raise ValueError("hsg must have kx+1 entries")
        check(self.lib.spdy_plan_set_sigma(self.h, _p(hsg)))

    def get_geopotential(self, t, phis):
        """geopotential.f90:33 -- t [kx,nx,mx], phis [nx,mx] -> phi [kx,nx,mx]."""
        t = np.ascontiguousarray(t, np.complex128); phis = np.ascontiguousarray(phis, np.complex128)
        if t.shape != (self.kx,) + self.spec_shape or phis.shape != self.spec_shape:
            raise ValueError("get_geopotential expects (kx,nx,mx), (nx,mx)")
        phi = np.empty_like(t)
        check(self.lib.spdy_geopotential(self.h, _p(t), _p(phis), _p(phi)))
        return phi

    def step_field(self, j1, dt, eps, wil, field, fdt):
        """time_stepping.f90:121-167 step_field_2d/3d -- field [2,nlev,nx,mx] (or [2,nx,mx]), fdt [nlev,nx,mx] (or
        [nx,mx]); returns the updated copies (field, fdt)."""
        f = np.array(field, np.complex128, order="C"); d = np.array(fdt, np.complex128, order="C")
        nlev = 1 if d.ndim == 2 else d.shape[0]
        check(self.lib.spdy_step_field(self.h, nlev, int(j1), float(dt), float(eps), float(wil), _p(f), _p(d)))
        return f, d

    def implicit_terms(self, divdt, tdt, psdt):
        """implicit.f90:168 (returns the updated copies)."""
        d = np.array(divdt, np.complex128, order="C")
        t = np.array(tdt, np.complex128, order="C")
        p = np.array(psdt, np.complex128, order="C")
        if d.shape != (self.kx,) + self.spec_shape or t.shape != d.shape or p.shape != self.spec_shape:
            raise ValueError("implicit_terms expects (kx,nx,mx),(kx,nx,mx),(nx,mx)")
        check(self.lib.spdy_implicit_terms(self.h, _p(d), _p(t), _p(p)))
        return d, t, p

    # ------------------------------------------------------------------ device-resident batch (torch tensors)
    # Stream policy of the device-resident (`*_dev`) methods.  Default: follow torch -- before every call the plan is
    # switched to torch's current stream, so kernels are ordered with the torch ops that produce/consume the tensors
    # (torch's legacy default stream maps to the plan's own blocking stream, which HIP orders against the default
    # stream).  use_own_stream(): the plan keeps its own stream (bench.py, graph replay loops); the caller synchronises.
    def use_torch_stream(self):
        self._follow = True
        self._sync_stream()

    def use_own_stream(self):
        self._follow = False
        self._cur_stream = 0
        check(self.lib.spdy_plan_set_stream(self.h, None))

    def _sync_stream(self):
        # (while a capture is open the plan's stream is pinned: spdy_plan_set_stream would fail with SPDY_ERR_STATE)
        if not getattr(self, "_follow", True) or getattr(self, "_capturing", False):
            return
        import torch
        h = torch.cuda.current_stream().cuda_stream
        if h != getattr(self, "_cur_stream", 0):
            check(self.lib.spdy_plan_set_stream(self.h, ctypes.c_void_p(h) if h else None))
            self._cur_stream = h

    KERNEL_KINDS = ("legendre_inv", "fourier_inv", "fourier_dir", "legendre_dir", "s2g_fused", "g2s_fused")

    def set_fused(self, mode):
        """1 / -1 (default) = fused single-pass kernels (T30, T63) at every batch size, 0 = four-kernel path."""
        check(self.lib.spdy_plan_set_fused(self.h, int(mode)))

    def set_option(self, name, value):
        """Launch-policy switch of this plan (spdy_plan_set_option): "t30_part", "t30_split", "t63_split", "t63_stage",
        "t63_derive" (0 / 1), "t63_np2_from", "wt_min_mb"."""
        check(self.lib.spdy_plan_set_option(self.h, name.encode(), int(value)))

    def wave_placement(self):
        """(SIMD of waves 0..7 of workgroup 0, number of workgroups that violate the round-robin placement the T63 kernels'
        role assignment relies on) -- spdy_wave_placement."""
        import ctypes
        simd, bad = (ctypes.c_int * 8)(), ctypes.c_int(0)
        check(self.lib.spdy_wave_placement(self.h, simd, ctypes.byref(bad)))
        return list(simd), bad.value

    def set_profiling(self, on=True):
        check(self.lib.spdy_plan_set_profiling(self.h, 1 if on else 0))

    def get_profile(self):
        """{kernel kind: (total ms, launches)} measured with HIP events on the launch stream."""
        ms = (ctypes.c_double * 6)()
        cnt = (ctypes.c_int * 6)()
        check(self.lib.spdy_plan_get_profile(self.h, ms, cnt))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(self.KERNEL_KINDS)}

    def synchronize(self):
        check(self.lib.spdy_plan_synchronize(self.h))

    def graph_capture(self):
        self._sync_stream()
        return self._graph_capture()

    def _graph_capture(self):
        """Context manager: record the ``*_dev`` calls made inside it (nothing runs) and return a
        :class:`Graph` whose ``launch()`` replays them as one HIP graph launch on the plan's stream::

            with sp.graph_capture() as g:
                sp.spec_to_grid_dev(spec, grid); sp.grid_to_spec_dev(grid, spec2)
            g.launch(); sp.synchronize()

        The plan must be on its own stream (or another non-default one), not torch's legacy default stream."""
        return _GraphCapture(self)

    @staticmethod
    def _dp(t):
        return ctypes.c_void_p(t.data_ptr())

    def spec_to_grid_dev(self, d_spec, d_grid, kcos=1, d_kcos=None):
        """d_spec: [nb, nx, mx] complex128 (or [nb,nx,mx,2] float64) CUDA tensor; d_grid: [nb, il, ix] float64."""
        self._sync_stream()
        nb = d_grid.shape[0]
        check(self.lib.spdy_spec_to_grid_dev(self.h, nb, self._dp(d_spec), self._dp(d_kcos) if d_kcos is not None else None,
                                             int(kcos), self._dp(d_grid)))

    def grid_to_spec_dev(self, d_grid, d_spec):
        self._sync_stream()
        nb = d_grid.shape[0]
        check(self.lib.spdy_grid_to_spec_dev(self.h, nb, self._dp(d_grid), self._dp(d_spec)))

    def uvspec_dev(self, vor, div, u, v):
        self._sync_stream()
        check(self.lib.spdy_uvspec_dev(self.h, vor.shape[0], self._dp(vor), self._dp(div), self._dp(u), self._dp(v)))

    def vdspec_dev(self, ug, vg, vor, div, kcos=2):
        self._sync_stream()
        check(self.lib.spdy_vdspec_dev(self.h, ug.shape[0], self._dp(ug), self._dp(vg), self._dp(vor), self._dp(div), int(kcos)))

    def uvspec_to_grid_dev(self, vor, div, ug, vg, kcos=2):
        """uvspec followed by spec_to_grid(., kcos) of both results (tendencies.f90:98-100), one pass at T30."""
        self._sync_stream()
        check(self.lib.spdy_uvspec_to_grid_dev(self.h, vor.shape[0], self._dp(vor), self._dp(div), self._dp(ug), self._dp(vg), int(kcos)))

    def grad_to_grid_dev(self, psi, gx, gy, kcos=2):
        """grad followed by spec_to_grid(., kcos) of both results (tendencies.f90:121-123), one pass at T30."""
        self._sync_stream()
        check(self.lib.spdy_grad_to_grid_dev(self.h, psi.shape[0], self._dp(psi), self._dp(gx), self._dp(gy), int(kcos)))

    def inverse_batch_dev(self, vor, div, ug, vg, spec, grid, kcos_pairs=2, kcos=1, d_kcos=None):
        """uvspec + spec_to_grid(., kcos_pairs) of the (vor, div) pairs and spec_to_grid of `spec` in one launch."""
        self._sync_stream()
        check(self.lib.spdy_inverse_batch_dev(self.h, vor.shape[0], self._dp(vor), self._dp(div), self._dp(ug), self._dp(vg), int(kcos_pairs),
                                              spec.shape[0], self._dp(spec), self._dp(d_kcos) if d_kcos is not None else None, int(kcos),
                                              self._dp(grid)))

    def inverse_batch_grad_dev(self, vor, div, ug, vg, spec, grid, psi, gx, gy, kcos_pairs=2, kcos=1, d_kcos=None, kcos_grad=2):
        """inverse_batch_dev + grad_to_grid_dev(psi -> gx, gy): everything a step transforms to the grid, one fused launch at T63."""
        self._sync_stream()
        check(self.lib.spdy_inverse_batch_grad_dev(self.h, vor.shape[0], self._dp(vor), self._dp(div), self._dp(ug), self._dp(vg),
                                                   int(kcos_pairs), spec.shape[0], self._dp(spec),
                                                   self._dp(d_kcos) if d_kcos is not None else None, int(kcos), self._dp(grid),
                                                   psi.shape[0], self._dp(psi), self._dp(gx), self._dp(gy), int(kcos_grad)))

    def inverse_batch_segs_dev(self, vor, div, ug, vg, specs, grid, psi=None, gx=None, gy=None, kcos_pairs=2, kcos=1, d_kcos=None, kcos_grad=2):
        """inverse_batch_grad_dev with the plain spectra in up to four separate arrays `specs` (tendencies.f90:89-101 reads
        vor, div, t, tr from four prognostic arrays); their grids are the one stack `grid`.  psi / gx / gy optional."""
        self._sync_stream()
        import ctypes

        class Seg(ctypes.Structure):
            _fields_ = [("nb", ctypes.c_int), ("d_spec", ctypes.c_void_p)]
        segs = (Seg * len(specs))(*[Seg(int(x.shape[0]), self._dp(x)) for x in specs])
        ngrad = 0 if psi is None else psi.shape[0]
        check(self.lib.spdy_inverse_batch_segs_dev(self.h, vor.shape[0], self._dp(vor), self._dp(div), self._dp(ug), self._dp(vg),
                                                   int(kcos_pairs), len(specs), ctypes.cast(segs, ctypes.c_void_p),
                                                   self._dp(d_kcos) if d_kcos is not None else None, int(kcos), self._dp(grid),
                                                   ngrad, self._dp(psi) if ngrad else None, self._dp(gx) if ngrad else None,
                                                   self._dp(gy) if ngrad else None, int(kcos_grad)))

    def direct_batch_dev(self, ug, vg, vor, div, grid, spec, kcos=2):
        """vdspec of the (ug, vg) pairs and grid_to_spec of `grid` in one launch (a model step's direct batch)."""
        self._sync_stream()
        check(self.lib.spdy_direct_batch_dev(self.h, ug.shape[0], self._dp(ug), self._dp(vg), self._dp(vor), self._dp(div), int(kcos),
                                             grid.shape[0], self._dp(grid), self._dp(spec)))

    def implicit_terms_dev(self, divdt, tdt, psdt):
        self._sync_stream()
        check(self.lib.spdy_implicit_terms_dev(self.h, self._dp(divdt), self._dp(tdt), self._dp(psdt)))

    def grid_tendencies_dev(self, ug, vg, tg, vorg, divg, trg, px, py, u_out, v_out, plain_out):
        """tendencies.f90:105-197 on the gridded prognostics; outputs are the operands of one direct_batch_dev launch."""
        self._sync_stream()
        args = (ug, vg, tg, vorg, divg, trg, px, py, u_out, v_out, plain_out)
        check(self.lib.spdy_grid_tendencies_dev(self.h, *[self._dp(x) for x in args]))

    def tendency_combine_dev(self, pdiv, pspec):
        """In place on the direct batch's outputs: divdt -= laplacian(KE), tdt += ttend, trdt += trtend, psdt(1,1) = 0."""
        self._sync_stream()
        check(self.lib.spdy_tendency_combine_dev(self.h, self._dp(pdiv), self._dp(pspec)))

    def spectral_step_dev(self, pvor, pdiv, pspec, vor, div, t, tr, ps, phis, tcorh, qcorh, sdrag, j1, dt, eps, wil, phi):
        """Everything after the direct batch in one launch: tendency_combine, spectral_tendencies, implicit_terms, hdiff_step
        and step_fields of ps, vor, div, t, tr (prognostics [2,kx,nx,mx] / ps [2,nx,mx], both time levels)."""
        self._sync_stream()
        args = (pvor, pdiv, pspec, vor, div, t, tr, ps, phis, tcorh, qcorh)
        check(self.lib.spdy_spectral_step_dev(self.h, *[self._dp(x) for x in args], float(sdrag), int(j1), float(dt), float(eps),
                                              float(wil), self._dp(phi)))

    def direct_batch_spectral_step_dev(self, ug, vg, grid, pvor, pdiv, pspec, vor, div, t, tr, ps, phis, tcorh, qcorh, sdrag, j1, dt,
                                       eps, wil, phi, kcos=2):
        """direct_batch_dev(ug, vg [3kx] -> pvor, pdiv; grid [3kx+1] -> pspec) + spectral_step_dev as one call (at T63 vds is
        applied where the spectral step reads the pairs' spectra: one launch less)."""
        self._sync_stream()
        args = (pvor, pdiv, pspec, vor, div, t, tr, ps, phis, tcorh, qcorh)
        check(self.lib.spdy_direct_batch_spectral_step_dev(self.h, self._dp(ug), self._dp(vg), self._dp(grid), int(kcos),
                                                           *[self._dp(x) for x in args], float(sdrag), int(j1), float(dt), float(eps),
                                                           float(wil), self._dp(phi)))

    def output_batch_dev(self, vor, div, t, q, phi, ps, u_out, v_out, t_out, q_out, phi_out, ps_out):
        """input_output.f90:184-206 on device-resident state: complex128 [kx,nx,mx] (ps [nx,mx]) in, float32 [kx,il,ix]
        (ps_out [il,ix]) out."""
        self._sync_stream()
        args = (vor, div, t, q, phi, ps, u_out, v_out, t_out, q_out, phi_out, ps_out)
        check(self.lib.spdy_output_batch_dev(self.h, *[self._dp(x) for x in args]))

    def geopotential_dev(self, t, phis, phi):
        self._sync_stream()
        check(self.lib.spdy_geopotential_dev(self.h, self._dp(t), self._dp(phis), self._dp(phi)))

    def spectral_tendencies_dev(self, div, t, ps, phis, divdt, tdt, psdt, phi):
        """tendencies.f90:242-293 -- div, t, ps: time level j2 of the prognostics; divdt, tdt, psdt in place; phi out."""
        self._sync_stream()
        check(self.lib.spdy_spectral_tendencies_dev(self.h, *[self._dp(x) for x in (div, t, ps, phis, divdt, tdt, psdt, phi)]))

    def hdiff_step_dev(self, vor, div, t, tr, tcorh, qcorh, sdrag, vordt, divdt, tdt, trdt):
        """The diffusion block of step() (time_stepping.f90:62-96) in one launch; tendencies in place."""
        self._sync_stream()
        dp = lambda x: self._dp(x) if x is not None else None
        check(self.lib.spdy_hdiff_step_dev(self.h, dp(vor), dp(div), dp(t), dp(tr), dp(tcorh), dp(qcorh), float(sdrag),
                                           dp(vordt), dp(divdt), dp(tdt), dp(trdt)))

    def step_fields_dev(self, pairs, j1, dt, eps, wil):
        """step_field_2d/3d for several prognostic arrays in one launch: pairs = [(field [2,nlev,nx,mx], fdt [nlev,nx,mx]), ...]."""
        self._sync_stream()
        class Op(ctypes.Structure):
            _fields_ = [("nlev", ctypes.c_int), ("field", ctypes.c_void_p), ("fdt", ctypes.c_void_p)]
        arr = (Op * len(pairs))()
        for o, (f, d) in zip(arr, pairs):
            nlev = 1 if d.dim() == 2 else d.shape[0]
            assert f.numel() == 2 * d.numel()
            o.nlev, o.field, o.fdt = nlev, f.data_ptr(), d.data_ptr()
        check(self.lib.spdy_step_fields_dev(self.h, len(pairs), ctypes.cast(arr, ctypes.c_void_p), int(j1), float(dt), float(eps), float(wil)))

    def hdiff_multi_dev(self, ops):
        """ops: up to 8 tuples (field, fdt_in, dmp_name, dmp1_name, out) -- the diffusion calls of one time step
        (time_stepping.f90:63-96) in one launch."""
        self._sync_stream()
        class Op(ctypes.Structure):
            _fields_ = [("nlev", ctypes.c_int), ("field", ctypes.c_void_p), ("fdt_in", ctypes.c_void_p),
                        ("d_dmp", ctypes.c_void_p), ("d_dmp1", ctypes.c_void_p), ("fdt_out", ctypes.c_void_p)]
        arr = (Op * len(ops))()
        for o, (field, fdt_in, dmp_name, dmp1_name, out) in zip(arr, ops):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            check(self.lib.spdy_device_table(self.h, dmp_name.encode(), ctypes.byref(a)))
            check(self.lib.spdy_device_table(self.h, dmp1_name.encode(), ctypes.byref(b)))
            o.nlev, o.field, o.fdt_in, o.d_dmp, o.d_dmp1, o.fdt_out = field.shape[0], field.data_ptr(), fdt_in.data_ptr(), a.value, b.value, out.data_ptr()
        check(self.lib.spdy_hdiff_multi_dev(self.h, len(ops), ctypes.cast(arr, ctypes.c_void_p)))

    def hdiff_dev(self, field, fdt_in, dmp_name, dmp1_name, out):
        self._sync_stream()
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        check(self.lib.spdy_device_table(self.h, dmp_name.encode(), ctypes.byref(a)))
        check(self.lib.spdy_device_table(self.h, dmp1_name.encode(), ctypes.byref(b)))
        check(self.lib.spdy_hdiff_dev(self.h, field.shape[0], self._dp(field), self._dp(fdt_in), a, b, self._dp(out)))
