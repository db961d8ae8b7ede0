#!/usr/bin/env python3
"""Regenerate tests/golden/ref_{t30,t63}.npz from the REAL reference.

Runs only in the build container: it needs oracle/_ref/libspeedy_ref_*.so, i.e. the
reference's own Fortran hot-path modules compiled by flang -O2 from /root/reference/source
(oracle/build_ref.sh).  The committed .npz files are data only -- seeded inputs and the
reference's outputs for them -- and are what pins the C oracle (tests/test_oracle_golden.py)
and, through it, the HIP path on the GPU box where /root/reference does not exist.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import synth  # noqa: E402
from oracle.pyoracle import Reference, build  # noqa: E402

DTS = (1200.0, 2400.0, 4800.0)   # time_stepping.f90:15-23: initialize_implicit(dt/2, dt, 2dt), delt=2400


tail_inputs = synth.tail_inputs   # seeded (divdt, tdt, psdt)-shaped inputs for implicit_terms / do_horizontal_diffusion

L16_SUB = (slice(None), slice(None, None, 5), slice(None, None, 3))   # the part of a [16,65,64] array kept for T63 L16


STEP_SUB = (slice(None), slice(None, None, 2), slice(None, None, 2))   # [.., kx, nx, mx] -> every second n and m


def geop_inputs(kx, nx, mx):
    """Seeded spectral temperature [kx,nx,mx] (~300) and surface geopotential [nx,mx] (~1000)."""
    return synth.cfield((kx, nx, mx), 5, 300.0), synth.cfield((nx, mx), 6, 1000.0)


def make_extra():
    """ref_extra.npz: the reference builds with other level counts (oracle/build_ref.sh: kx = 5, 7 and the 16-level
    T63 build that gets its half levels through the reference's public geometry variables) + get_geopotential."""
    from oracle.pyoracle import Oracle
    d = {}
    for tag, sub in (("t30", None), ("t63", None), ("t30k5", None), ("t30k7", None), ("t63k16", L16_SUB)):
        r = Reference(tag)
        kx, nx, mx = r.kx, r.nx, r.mx
        cut = (lambda a: a[sub]) if sub else (lambda a: a)
        if tag == "t63k16":
            # half levels in, derived level tables from the C oracle (bit-equal to the reference's own at kx = 5, 7, 8)
            o = Oracle(r.trunc, r.ix, r.iy, kx)
            o.set_sigma(synth.SIGMA_L16)
            r.set_sigma(*[o.table(n) for n in ("hsg", "dhs", "fsg", "dhsr", "fsgr")])
        d[tag + "_coriol"] = r.coriol()
        d.update({tag + "_" + k: v for k, v in r.sigma().items()})
        T, phis = geop_inputs(kx, nx, mx)
        d[tag + "_geop"] = cut(r.geopotential(T, phis))
        div, t, ps = tail_inputs(kx, nx, mx)
        for dt in ((4800.0,) if tag.startswith("t63") else (1200.0, 4800.0)):
            key = "%s_dt%d_" % (tag, int(dt))
            r.tail_init(dt)
            d.update({key + k: v for k, v in r.tref_tables().items()})
            d.update({tag + "_" + k: v for k, v in r.corv().items()})
            if tag in ("t30", "t63"):
                continue                                  # their implicit/hdiff outputs are in ref_<tag>.npz
            a, b, c = r.implicit_terms(div, t, ps)
            d[key + "imp_div_out"], d[key + "imp_t_out"], d[key + "imp_ps_out"] = cut(a), cut(b), c
            dm = r.dmp_tables()
            d[key + "hdiff3d"] = cut(r.hdiff(t, div, dm["dmpd"], dm["dmp1d"]))
    out = os.path.join(HERE, "ref_extra.npz")
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB")


def step_inputs(kx, nx, mx):
    """Seeded prognostic-like stacks for step_field: field [2,kx,nx,mx], tendency [kx,nx,mx] (not band-limited, so that the
    reference's trunct of the tendency matters), and the 2-D pair."""
    return (synth.cfield((2, kx, nx, mx), 11), synth.cfield((kx, nx, mx), 12, 1e-4),
            synth.cfield((2, nx, mx), 13), synth.cfield((nx, mx), 14, 1e-4))


def make_step():
    """ref_step.npz: step_field_3d / step_field_2d (time_stepping.f90:126-167) of the flang-built reference for j1 = 1
    (forward step, eps = 0) and j1 = 2 (leapfrog + Robert-Asselin-Williams filter, eps = rob) at dt = delt and 2 delt."""
    d = {}
    for tag, sub in (("t30", None), ("t30k5", STEP_SUB), ("t63k16", L16_SUB)):
        r = Reference(tag)
        wil, rob = r.wil_rob()
        d[tag + "_wil_rob"] = np.array([wil, rob])
        F3, D3, F2, D2 = step_inputs(r.kx, r.nx, r.mx)
        for j1, dt, eps in ((1, 2400.0, 0.0), (2, 4800.0, rob)):
            key = "%s_j%d_" % (tag, j1)
            ss = STEP_SUB if (sub is None and j1 == 1) else sub       # T30 L8: j1 = 2 complete, j1 = 1 a sub-lattice
            cut = (lambda a, ss=ss: a[(Ellipsis,) + ss]) if ss else (lambda a: a)
            f3, d3 = r.step_field(j1, dt, eps, F3, D3)
            f2, d2 = r.step_field(j1, dt, eps, F2, D2)
            d[key + "f3"], d[key + "d3"], d[key + "f2"], d[key + "d2"] = cut(f3), cut(d3), f2, d2
    out = os.path.join(HERE, "ref_step.npz")
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB")


def spectend_inputs(kx, nx, mx):
    """Seeded inputs of get_spectral_tendencies: div, t [kx,nx,mx], ps, phis [nx,mx] (one time level) and the tendencies
    divdt, tdt [kx,nx,mx], psdt [nx,mx] it updates -- magnitudes of the model's fields."""
    return (synth.cfield((kx, nx, mx), 21, 1e-5), synth.cfield((kx, nx, mx), 22, 30.0), synth.cfield((nx, mx), 23, 0.05),
            synth.cfield((nx, mx), 24, 2000.0), synth.cfield((kx, nx, mx), 25, 1e-9), synth.cfield((kx, nx, mx), 26, 1e-3),
            synth.cfield((nx, mx), 27, 1e-7))


def make_spectend():
    """ref_spectend.npz: get_spectral_tendencies (tendencies.f90:241-293) of the flang-built reference -- the subroutine cut out
    of the reference file by oracle/build_ref.sh -- at 8, 5 and 16 levels, j2 = 1 and 2 (the time level only selects the slab of
    the reference's prognostic arrays the inputs are put into)."""
    from oracle.pyoracle import Oracle
    d = {}
    for tag, sub in (("t30", None), ("t30k5", None), ("t63k16", L16_SUB)):
        r = Reference(tag)
        kx, nx, mx = r.kx, r.nx, r.mx
        cut = (lambda a, sub=sub: a[sub]) if sub else (lambda a: a)
        if tag == "t63k16":
            o = Oracle(r.trunc, r.ix, r.iy, kx)
            o.set_sigma(synth.SIGMA_L16)
            r.set_sigma(*[o.table(n) for n in ("hsg", "dhs", "fsg", "dhsr", "fsgr")])
        r.tail_init(4800.0)
        div, t, ps, phis, divdt, tdt, psdt = spectend_inputs(kx, nx, mx)
        for j2 in (1, 2):
            a, b, c, phi = r.spectral_tendencies(div, t, ps, phis, divdt, tdt, psdt, j2=j2)
            key = "%s_j%d_" % (tag, j2)
            if j2 == 2 and sub is None:                   # the second slab must give the same numbers: a sub-lattice is enough
                cut = lambda a: a[STEP_SUB]
            d[key + "divdt"], d[key + "tdt"], d[key + "psdt"], d[key + "phi"] = cut(a), cut(b), c, cut(phi)
    out = os.path.join(HERE, "ref_spectend.npz")
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB")


DYN63_SUB = (slice(None), slice(None, None, 9), slice(None, None, 5))   # the part of a [16,65,64] array kept for the step fixture
DYNSTEP_CASES = ((1, 1, 1200.0), (2, 2, 4800.0))     # (j1, j2, dt): the forward half step of first_step, the leapfrog step


def make_dynstep():
    """ref_dynstep.npz: one ADIABATIC time step of the flang-built reference -- time_stepping.f90 step(j1, j2, dt), the file
    compiled unchanged on tendencies.f90 minus its three physics lines (oracle/build_ref.sh) -- from the seeded state of
    tests/dynstep.py at 8, 5 and 16 levels: both time levels of the five prognostics and phi (sub-lattices of the arrays)."""
    import dynstep
    from oracle.pyoracle import Oracle
    d = {}
    for tag, sub in (("t30", STEP_SUB), ("t30k5", STEP_SUB), ("t63k16", DYN63_SUB)):
        r = Reference(tag)
        o = Oracle(r.trunc, r.ix, r.iy, r.kx)
        if tag == "t63k16":
            o.set_sigma(synth.SIGMA_L16)
            r.set_sigma(*[o.table(n) for n in ("hsg", "dhs", "fsg", "dhsr", "fsgr")])
        st = dynstep.state(o, 8000)
        for j1, j2, dt in (DYNSTEP_CASES if tag == "t30" else DYNSTEP_CASES[1:]):
            r.tail_init(dt)
            new, phi = r.step(j1, j2, dt, st)
            key = "%s_j%d%d_" % (tag, j1, j2)
            for n in ("vor", "div", "t", "tr"):
                d[key + n] = new[n][(Ellipsis,) + sub[1:]]
            d[key + "ps"], d[key + "phi"] = new["ps"], phi[sub]
            vt = r.get_tendencies(j2, st)
            for n, a in zip(("vordt", "divdt", "tdt", "psdt", "trdt"), vt):
                d[key + n] = a if n == "psdt" else a[sub]
    out = os.path.join(HERE, "ref_dynstep.npz")
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB")


def make_longrun():
    """ref_run72.npz: first_step + 72 leapfrog steps (2 days at delt = 2400 s) of the flang-built reference's ADIABATIC step from
    the reference's own rest state over a seeded orography, and from the same state with a seeded wind field (tests/longrun.py):
    the prognostics after 1, 2, 4, 9, 18, 36 and 72 steps (ps whole, every second level / n / m of the 3-D fields)."""
    import longrun
    from oracle.pyoracle import Oracle
    r = Reference("t30")
    o = Oracle(r.trunc, r.ix, r.iy, r.kx)
    d = {}
    for case, amp in longrun.CASES.items():
        st = longrun.rest_state(o, wind=amp)
        out = longrun.run(lambda j1, j2, dt, s: r.step(j1, j2, dt, s)[0], r.tail_init, st)
        for n, state in out.items():
            for k, a in state.items():
                d["%s_%d_%s" % (case, n, k)] = longrun.cut(k, a)
    out = os.path.join(HERE, "ref_run72.npz")
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB")


OUT_SUB = {"t30": (slice(None), slice(None), slice(None, None, 2)), "t30k5": (slice(None), slice(None), slice(None, None, 2)),
           "t63k16": (slice(None), slice(None, None, 3), slice(None, None, 4))}   # [kx, il, ix] sub-lattices kept per build
OUT_SEED = 7000


def make_output():
    """ref_output.npz: the gridded snapshot of the flang-built reference -- the computing lines of input_output.f90's subroutine
    output (:183-205: uvspec, five inverse transforms per level, float32 conversions) cut out of the reference file as they are
    (oracle/build_ref.sh) -- from the seeded state of tests/dynstep.py at 8, 5 and 16 levels: float32 u, v, t, q, phi
    (sub-lattices) and ps (whole)."""
    import dynstep
    from oracle.pyoracle import Oracle
    d = {}
    for tag, sub in OUT_SUB.items():
        r = Reference(tag)
        o = Oracle(r.trunc, r.ix, r.iy, r.kx)
        if tag == "t63k16":
            o.set_sigma(synth.SIGMA_L16)
            r.set_sigma(*[o.table(n) for n in ("hsg", "dhs", "fsg", "dhsr", "fsgr")])
        st = dynstep.state(o, OUT_SEED)
        phi = r.geopotential(st["t"][0], st["phis"])
        outs = r.output(st["vor"][0], st["div"][0], st["t"][0], st["tr"][0], phi, st["ps"][0])
        for n, a in zip(("u", "v", "t", "q", "phi", "ps"), outs):
            d["%s_%s" % (tag, n)] = a if n == "ps" else a[sub]
    out = os.path.join(HERE, "ref_output.npz")
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB")


def make(tag, nb_grid, dts, imp_dts, lean):
    r = Reference(tag)
    tr, ix, il, kx, nx, mx = r.trunc, r.ix, r.il, r.kx, r.nx, r.mx
    d = {"dims": np.array([tr, ix, r.iy, il, kx, nx, mx], np.int32)}
    d.update({"tab_" + k: v for k, v in r.geometry().items() if k != "coa_half"})
    d["tab_coa_half"] = r.geometry()["coa_half"][: r.iy]
    d.update({"tab_" + k: v for k, v in r.sigma().items()})
    wa, ifac = r.rffti1()
    d["tab_work"], d["tab_ifac"] = wa, ifac
    d["tab_epsi"], d["tab_el2"] = r.epsi(), r.el2()

    # raw FFTPACK vectors
    v = synth.splitmix64(7, ix) - 0.5
    d["fft_in"], d["fft_b"], d["fft_f"] = v, r.rfftb1(v), r.rfftf1(v)

    S = synth.spectra(2, tr, full_rows=True)
    G = synth.grids(2, ix, il)
    d["S"], d["G"] = S, G[:max(nb_grid, 2)]
    # transform stages and API, field by field
    nb = nb_grid
    d["leginv"] = np.stack([r.legendre_inv(S[b]) for b in range(nb)])
    d["finv1"] = np.stack([r.fourier_inv(d["leginv"][b], 1) for b in range(nb)])
    d["finv2"] = np.stack([r.fourier_inv(d["leginv"][b], 2) for b in range(nb)])
    d["fdir"] = np.stack([r.fourier_dir(G[b]) for b in range(nb)])
    d["legdir"] = np.stack([r.legendre_dir(d["fdir"][b]) for b in range(nb)])
    d["s2g1"] = np.stack([r.spec_to_grid(S[b], 1) for b in range(nb)])
    d["s2g2"] = np.stack([r.spec_to_grid(S[b], 2) for b in range(nb)])
    d["g2s"] = np.stack([r.grid_to_spec(G[b]) for b in range(nb)])
    d["ones_g2s"] = r.grid_to_spec(np.ones((il, ix)))
    # spectral operators
    d["lap"], d["invlap"], d["trunct"] = r.laplacian(S[0]), r.inverse_laplacian(S[0]), r.trunct(S[0])
    d["grad_dx"], d["grad_dy"] = r.grad(S[0])
    d["vds_vor"], d["vds_div"] = r.vds(S[0], S[1])
    d["uv_u"], d["uv_v"] = r.uvspec(S[0], S[1])
    d["vdspec2_vor"], d["vdspec2_div"] = r.vdspec(G[0], G[1], 2)
    d["vdspec1_vor"], d["vdspec1_div"] = r.vdspec(G[0], G[1], 1)
    # spectral-space tail (horizontal diffusion + semi-implicit solve)
    imp_div, imp_t, imp_ps = tail_inputs(kx, nx, mx)
    if not lean:   # lean fixtures (T63) regenerate these from the seed via tail_inputs()
        d["imp_div"], d["imp_t"], d["imp_ps"] = imp_div, imp_t, imp_ps
    d["dts"] = np.array(dts)
    d["imp_dts"] = np.array(imp_dts)
    for dt in dts:
        key = "dt%d_" % int(dt)
        r.tail_init(dt)
        dm = r.dmp_tables()
        d.update({key + k: v for k, v in dm.items()})
        d.update({key + k: v for k, v in r.tref_tables().items()})
        if dt in imp_dts:
            a, b, c = r.implicit_terms(imp_div, imp_t, imp_ps)
            d[key + "imp_div_out"], d[key + "imp_t_out"], d[key + "imp_ps_out"] = a, b, c
        if dt == dts[-1]:
            if not lean:
                d[key + "hdiff3d"] = r.hdiff(imp_t, imp_div, dm["dmp"], dm["dmp1"])
            d[key + "hdiff2d"] = r.hdiff(imp_ps, 2 * imp_ps, dm["dmps"], dm["dmp1s"])
    out = os.path.join(HERE, "ref_%s.npz" % tag)
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB")


if __name__ == "__main__":
    build(quiet=True)
    if len(sys.argv) > 1 and sys.argv[1] == "step":      # only the step_field fixture
        make_step()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dynstep":   # only the adiabatic-step fixture
        make_dynstep()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "output":    # only the gridded-snapshot fixture
        make_output()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "longrun":   # only the 2-day run fixture
        make_longrun()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "spectend":  # only the get_spectral_tendencies fixture
        make_spectend()
        sys.exit(0)
    make("t30", nb_grid=2, dts=DTS, imp_dts=(1200.0, 4800.0), lean=False)
    make("t63", nb_grid=1, dts=(4800.0,), imp_dts=(4800.0,), lean=True)
    make_extra()
    make_step()
    make_spectend()
    make_dynstep()
    make_output()
    make_longrun()
