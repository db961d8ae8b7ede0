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


def tail_inputs(kx, nx, mx):
    """Seeded (divdt, tdt, psdt)-shaped inputs for implicit_terms / do_horizontal_diffusion."""
    u = synth.splitmix64(99, 2 * (2 * kx + 1) * nx * mx).reshape(2 * kx + 1, nx, mx, 2) * 2 - 1
    z = u[..., 0] + 1j * u[..., 1]
    return z[:kx] * 1e-6, z[kx:2 * kx] * 1e-3, z[2 * kx] * 1e-5


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
    make("t30", nb_grid=2, dts=DTS, imp_dts=(1200.0, 4800.0), lean=False)
    make("t63", nb_grid=1, dts=(4800.0,), imp_dts=(4800.0,), lean=True)
