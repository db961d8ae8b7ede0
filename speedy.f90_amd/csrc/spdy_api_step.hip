// C ABI (include/spdy.h), second half: the spectral-space tail of a time step (horizontal diffusion, semi-implicit
// solve, spectral tendencies, geopotential, leapfrog/RAW filter) and the output path.  (Multi-GPU: spdy_api_shard.hip.)
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "spdy_plan.hpp"

using spdy::HostTables;
using namespace spdy_detail;

extern "C" {

/* ---------------------------------------------------------------- sigma levels */
int spdy_plan_set_sigma(spdy_plan *p, const double *hsg)
{
    NEED_PLAN(p);
    NOT_CAPTURING(p, "spdy_plan_set_sigma");
    const std::string err = p->tab.set_sigma(hsg);
    if (!err.empty()) return fail(SPDY_ERR_ARG, "set_sigma: %s", err.c_str());
    if (p->device < 0) return SPDY_OK;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return upload_level_tables(p);
}

/* ---------------------------------------------------------------- horizontal diffusion */
int spdy_hdiff_dev(spdy_plan *p, int nlev, const double *field, const double *fdt_in, const double *d_dmp,
                   const double *d_dmp1, double *fdt_out)
{
    NEED_DEVICE(p);
    if (nlev < 0) return fail(SPDY_ERR_ARG, "nlev < 0");
    if (nlev && (!field || !fdt_in || !d_dmp || !d_dmp1 || !fdt_out)) return fail(SPDY_ERR_ARG, "null device pointer");
    KERNEL(spdy::launch_hdiff(p->dev, nlev, field, fdt_in, d_dmp, d_dmp1, fdt_out, p->stream));
    return SPDY_OK;
}

int spdy_hdiff_multi_dev(spdy_plan *p, int nops, const spdy_hdiff_op *ops)
{
    NEED_DEVICE(p);
    if (nops < 0 || nops > SPDY_HDIFF_MAX_OPS || (nops && !ops)) return fail(SPDY_ERR_ARG, "nops=%d outside [0, %d]", nops, (int)SPDY_HDIFF_MAX_OPS);
    spdy::HdiffOps h{};
    h.nops = nops;
    for (int i = 0; i < nops; ++i) {
        if (ops[i].nlev < 0 || (ops[i].nlev && (!ops[i].field || !ops[i].fdt_in || !ops[i].d_dmp || !ops[i].d_dmp1 || !ops[i].fdt_out)))
            return fail(SPDY_ERR_ARG, "hdiff op %d: bad argument", i);
        h.nlev[i] = ops[i].nlev; h.field[i] = ops[i].field; h.fdt[i] = ops[i].fdt_in;
        h.dmp[i] = ops[i].d_dmp; h.dmp1[i] = ops[i].d_dmp1; h.out[i] = ops[i].fdt_out;
    }
    KERNEL(spdy::launch_hdiff_multi(p->dev, h, p->stream));
    return SPDY_OK;
}

int spdy_hdiff(spdy_plan *p, int nlev, const double *field, const double *fdt_in, const double *dmp,
               const double *dmp1, double *fdt_out)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nlev));
    if (nlev && (!field || !fdt_in || !dmp || !dmp1 || !fdt_out)) return fail(SPDY_ERR_ARG, "null pointer");
    RC(ensure_staging(p));
    const size_t n = nlev * spec_elems(p), tn = (size_t)p->tab.mx * p->tab.nx;
    RC(h2d(p, p->stage_a, field, n));
    RC(h2d(p, p->stage_b, fdt_in, n));
    RC(h2d(p, p->stage_c, dmp, tn));            // both damping tables share stage_c (2*mx*nx <= ix*il) ...
    RC(h2d(p, p->stage_c + tn, dmp1, tn));
    RC(spdy_hdiff_dev(p, nlev, p->stage_a, p->stage_b, p->stage_c, p->stage_c + tn, p->stage_d));   // ... so the result has its own buffer
    RC(d2h(p, fdt_out, p->stage_d, n));
    return sync(p);
}

int spdy_device_table(spdy_plan *p, const char *name, const double **d_ptr)
{
    NEED_DEVICE(p);
    if (!name || !d_ptr) return fail(SPDY_ERR_ARG, "null argument");
    static const char *names[6] = {"dmp", "dmpd", "dmps", "dmp1", "dmp1d", "dmp1s"};
    for (int i = 0; i < 6; ++i)
        if (!std::strcmp(name, names[i])) {
            if (i >= 3 && !p->tab.implicit_ready) return fail(SPDY_ERR_STATE, "%s needs implicit_init first", name);
            *d_ptr = p->d_dmp[i];
            return SPDY_OK;
        }
    return fail(SPDY_ERR_ARG, "no device table '%s'", name);
}

/* ---------------------------------------------------------------- semi-implicit solve */
int spdy_implicit_init(spdy_plan *p, double dt)
{
    NEED_PLAN(p);
    NOT_CAPTURING(p, "spdy_implicit_init (host table build + blocking upload)");
    const std::string err = p->tab.build_implicit(dt);
    if (!err.empty()) return fail(SPDY_ERR_UNSUPPORTED, "implicit_init: %s", err.c_str());
    if (p->device < 0) return SPDY_OK;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    const HostTables &t = p->tab;
    const size_t tn = sizeof(double) * t.mx * t.nx;
    const std::vector<double> *d1[3] = {&t.dmp1, &t.dmp1d, &t.dmp1s};
    for (int i = 0; i < 3; ++i) HIP_TRY(hipMemcpy(p->d_dmp[3 + i], d1[i]->data(), tn, hipMemcpyHostToDevice));
    struct { double *dst; const std::vector<double> *src; } up[6] = {
        {p->d_xd, &t.xd}, {p->d_xc, &t.xc}, {p->d_xj, &t.xj}, {p->d_tref1, &t.tref1}, {p->d_dhsx, &t.dhsx}, {p->d_elz, &t.elz}};
    for (auto &u : up) HIP_TRY(hipMemcpy(u.dst, u.src->data(), u.src->size() * sizeof(double), hipMemcpyHostToDevice));
    {   // row-major, row-padded copies: element (k, k1) of a column-major kx x kx matrix at [k][k1]
        const int kx = t.kx, kxp = (kx + 1) & ~1, nl = t.mx + t.nx + 1;
        std::vector<double> xt((size_t)kx * kxp * (2 + nl), 0.0);
        auto tr = [&](const double *src, double *dst) {
            for (int k = 0; k < kx; ++k)
                for (int k1 = 0; k1 < kx; ++k1) dst[(size_t)k * kxp + k1] = src[k + (size_t)kx * k1];
        };
        tr(t.xd.data(), xt.data());
        tr(t.xc.data(), xt.data() + (size_t)kx * kxp);
        for (int l = 0; l < nl; ++l) tr(t.xj.data() + (size_t)kx * kx * l, xt.data() + (size_t)kx * kxp * (2 + l));
        HIP_TRY(hipMemcpy(p->d_xt, xt.data(), xt.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return upload_level_tables(p);
}

int spdy_implicit_terms_dev(spdy_plan *p, double *divdt, double *tdt, double *psdt)
{
    NEED_DEVICE(p);
    if (!p->tab.implicit_ready) return fail(SPDY_ERR_STATE, "implicit_terms before implicit_init");
    if (!divdt || !tdt || !psdt) return fail(SPDY_ERR_ARG, "null pointer");
    KERNEL(spdy::launch_implicit(p->dev, divdt, tdt, psdt, p->stream));
    return SPDY_OK;
}

int spdy_implicit_terms(spdy_plan *p, double *divdt, double *tdt, double *psdt)
{
    NEED_DEVICE(p);
    if (p->max_batch < p->tab.kx) return fail(SPDY_ERR_ARG, "max_batch must be >= kx for the host implicit_terms");
    if (!divdt || !tdt || !psdt) return fail(SPDY_ERR_ARG, "null pointer");
    RC(ensure_staging(p));
    const size_t n = p->tab.kx * spec_elems(p);
    RC(h2d(p, p->stage_a, divdt, n));
    RC(h2d(p, p->stage_b, tdt, n));
    RC(h2d(p, p->stage_c, psdt, spec_elems(p)));
    RC(spdy_implicit_terms_dev(p, p->stage_a, p->stage_b, p->stage_c));
    RC(d2h(p, divdt, p->stage_a, n));
    RC(d2h(p, tdt, p->stage_b, n));
    RC(d2h(p, psdt, p->stage_c, spec_elems(p)));
    return sync(p);
}

/* ---------------------------------------------------------------- spectral side of a time step */
int spdy_geopotential_dev(spdy_plan *p, const double *t, const double *phis, double *phi)
{
    NEED_DEVICE(p);
    if (!p->tab.sigma_ready) return fail(SPDY_ERR_STATE, "geopotential needs sigma levels (kx in {5,7,8} or spdy_plan_set_sigma)");
    if (!t || !phis || !phi) return fail(SPDY_ERR_ARG, "null device pointer");
    KERNEL(spdy::launch_geopotential(p->dev, t, phis, phi, p->stream));
    return SPDY_OK;
}

int spdy_geopotential(spdy_plan *p, const double *t, const double *phis, double *phi)
{
    NEED_DEVICE(p);
    if (p->max_batch < p->tab.kx) return fail(SPDY_ERR_ARG, "max_batch must be >= kx for the host get_geopotential");
    if (!t || !phis || !phi) return fail(SPDY_ERR_ARG, "null pointer");
    RC(ensure_staging(p));
    const size_t n = p->tab.kx * spec_elems(p);
    RC(h2d(p, p->stage_a, t, n));
    RC(h2d(p, p->stage_b, phis, spec_elems(p)));
    RC(spdy_geopotential_dev(p, p->stage_a, p->stage_b, p->stage_c));
    RC(d2h(p, phi, p->stage_c, n));
    return sync(p);
}

int spdy_spectral_tendencies_dev(spdy_plan *p, const double *div, const double *t, const double *ps, const double *phis,
                                 double *divdt, double *tdt, double *psdt, double *phi)
{
    NEED_DEVICE(p);
    if (!p->tab.implicit_ready) return fail(SPDY_ERR_STATE, "spectral_tendencies needs the reference temperature profile: call spdy_implicit_init first");
    if (!div || !t || !ps || !phis || !divdt || !tdt || !psdt || !phi) return fail(SPDY_ERR_ARG, "null device pointer");
    KERNEL(spdy::launch_spectral_tendencies(p->dev, div, t, ps, phis, divdt, tdt, psdt, phi, p->stream));
    return SPDY_OK;
}

int spdy_hdiff_step_dev(spdy_plan *p, const double *vor, const double *div, const double *t, const double *tr,
                        const double *d_tcorh, const double *d_qcorh, double sdrag,
                        double *vordt, double *divdt, double *tdt, double *trdt)
{
    NEED_DEVICE(p);
    if (!p->tab.implicit_ready) return fail(SPDY_ERR_STATE, "hdiff_step needs dmp1*: call spdy_implicit_init first");
    if (!p->tab.sigma_ready) return fail(SPDY_ERR_STATE, "hdiff_step needs sigma levels");
    if (!vor || !div || !t || !d_tcorh || !vordt || !divdt || !tdt) return fail(SPDY_ERR_ARG, "null device pointer");
    if ((tr != nullptr) != (trdt != nullptr) || (tr && !d_qcorh)) return fail(SPDY_ERR_ARG, "tracer arguments: tr, trdt and qcorh go together");
    spdy::HdiffStep h{vor, div, t, tr, vordt, divdt, tdt, trdt, d_tcorh, d_qcorh,
                      p->d_dmp[0], p->d_dmp[1], p->d_dmp[2], p->d_dmp[3], p->d_dmp[4], p->d_dmp[5], sdrag};
    KERNEL(spdy::launch_hdiff_step(p->dev, h, p->stream));
    return SPDY_OK;
}

int spdy_step_fields_dev(spdy_plan *p, int nops, const spdy_step_op *ops, int j1, double dt, double eps, double wil)
{
    NEED_DEVICE(p);
    if (nops < 0 || nops > SPDY_STEP_MAX_OPS || (nops && !ops)) return fail(SPDY_ERR_ARG, "nops=%d outside [0, %d]", nops, (int)SPDY_STEP_MAX_OPS);
    if (j1 != 1 && j1 != 2) return fail(SPDY_ERR_ARG, "j1 must be 1 or 2");
    spdy::StepOps s{};
    s.nops = nops;
    for (int i = 0; i < nops; ++i) {
        if (ops[i].nlev < 0 || (ops[i].nlev && (!ops[i].field || !ops[i].fdt))) return fail(SPDY_ERR_ARG, "step op %d: bad argument", i);
        s.nlev[i] = ops[i].nlev; s.field[i] = ops[i].field; s.fdt[i] = ops[i].fdt;
    }
    // time_stepping.f90:155-157: the tendency is truncated first whenever ix == 4*iy (true for T30 and T63)
    const int do_trunct = p->tab.ix == 4 * p->tab.iy;
    KERNEL(spdy::launch_step_fields(p->dev, s, j1, dt, eps, wil, do_trunct, p->stream));
    return SPDY_OK;
}

int spdy_step_field(spdy_plan *p, int nlev, int j1, double dt, double eps, double wil, double *field, double *fdt)
{
    NEED_DEVICE(p);
    RC(check_batch(p, 2 * nlev));
    if (nlev && (!field || !fdt)) return fail(SPDY_ERR_ARG, "null pointer");
    RC(ensure_staging(p));
    const size_t n = nlev * spec_elems(p);
    RC(h2d(p, p->stage_a, field, 2 * n));
    RC(h2d(p, p->stage_b, fdt, n));
    const spdy_step_op op{nlev, p->stage_a, p->stage_b};
    RC(spdy_step_fields_dev(p, 1, &op, j1, dt, eps, wil));
    RC(d2h(p, field, p->stage_a, 2 * n));
    RC(d2h(p, fdt, p->stage_b, n));
    return sync(p);
}


/* ---------------------------------------------------------------- grid-space dynamical tendencies */
int spdy_grid_tendencies_dev(spdy_plan *p, const double *ug, const double *vg, const double *tg, const double *vorg, const double *divg,
                             const double *trg, const double *px, const double *py, double *u_out, double *v_out, double *plain_out)
{
    NEED_DEVICE(p);
    if (!p->tab.implicit_ready) return fail(SPDY_ERR_STATE, "grid_tendencies needs the reference temperature profile: call spdy_implicit_init first");
    if (!ug || !vg || !tg || !vorg || !divg || !trg || !px || !py || !u_out || !v_out || !plain_out) return fail(SPDY_ERR_ARG, "null device pointer");
    const spdy::GridTend g{ug, vg, tg, vorg, divg, trg, px, py, u_out, v_out, plain_out};
    KERNEL(spdy::launch_grid_tendencies(p->dev, g, p->stream));
    return SPDY_OK;
}

int spdy_tendency_combine_dev(spdy_plan *p, double *pdiv, double *pspec)
{
    NEED_DEVICE(p);
    if (!pdiv || !pspec) return fail(SPDY_ERR_ARG, "null device pointer");
    KERNEL(spdy::launch_tendency_combine(p->dev, pdiv, pspec, p->stream));
    return SPDY_OK;
}

int spdy_spectral_step_dev(spdy_plan *p, double *pvor, double *pdiv, double *pspec, double *vor, double *div, double *t, double *tr,
                           double *ps, const double *phis, const double *d_tcorh, const double *d_qcorh, double sdrag, int j1, double dt,
                           double eps, double wil, double *phi)
{
    NEED_DEVICE(p);
    if (!p->tab.implicit_ready) return fail(SPDY_ERR_STATE, "spectral_step needs spdy_implicit_init first");
    if (!p->tab.sigma_ready) return fail(SPDY_ERR_STATE, "spectral_step needs sigma levels");
    if (!pvor || !pdiv || !pspec || !vor || !div || !t || !tr || !ps || !phis || !d_tcorh || !d_qcorh || !phi)
        return fail(SPDY_ERR_ARG, "null device pointer");
    if (j1 != 1 && j1 != 2) return fail(SPDY_ERR_ARG, "j1 must be 1 or 2");
    const int kx = p->tab.kx;
    if (kx > 16) {   // the fused kernel holds one level per thread row (64 x kx <= 1024 threads): issue the separate kernels
        RC(spdy_tendency_combine_dev(p, pdiv, pspec));
        const size_t L = (size_t)kx * spec_elems(p);
        double *divdt = pdiv, *tdt = pdiv + L, *trdt = pdiv + 2 * L, *psdt = pspec + 3 * L;
        RC(spdy_spectral_tendencies_dev(p, div, t, ps, phis, divdt, tdt, psdt, phi));
        RC(spdy_implicit_terms_dev(p, divdt, tdt, psdt));
        RC(spdy_hdiff_step_dev(p, vor, div, t, tr, d_tcorh, d_qcorh, sdrag, pvor, divdt, tdt, trdt));
        const spdy_step_op ops[5] = {{1, ps, psdt}, {kx, vor, pvor}, {kx, div, divdt}, {kx, t, tdt}, {kx, tr, trdt}};
        return spdy_step_fields_dev(p, 5, ops, j1, dt, eps, wil);
    }
    const spdy::SpecStep a{pvor, pdiv, pspec, vor, div, t, tr, ps, phis, d_tcorh, d_qcorh, phi, sdrag, dt, eps, wil, j1,
                           p->tab.ix == 4 * p->tab.iy, nullptr, nullptr};
    KERNEL(spdy::launch_spectral_step(p->dev, a, p->stream));
    return SPDY_OK;
}

int spdy_direct_batch_spectral_step_dev(spdy_plan *p, const double *ug, const double *vg, const double *grid, int kcos, double *pvor,
                                        double *pdiv, double *pspec, double *vor, double *div, double *t, double *tr, double *ps,
                                        const double *phis, const double *d_tcorh, const double *d_qcorh, double sdrag, int j1, double dt,
                                        double eps, double wil, double *phi)
{
    NEED_DEVICE(p);
    const int kx = p->tab.kx, P = 3 * kx;
    if (!ug || !vg || !grid || !pvor || !pdiv || !pspec) return fail(SPDY_ERR_ARG, "null device pointer");
    if (p->tab.trunc == 63 && p->fused_mode != 0 && kx <= 16 && P <= p->max_batch && p->tab.implicit_ready && p->tab.sigma_ready && vor && div && t && tr &&
        ps && phis && d_tcorh && d_qcorh && phi && (j1 == 1 || j1 == 2)) {
        // T63: the transform kernel leaves the pairs' spectra un-vds'ed in the plan's temporaries; the spectral step applies
        // vds where it reads them -- direct batch + everything after it = 2 launches instead of 3
        RC(direct_batch_raw63(p, P, ug, vg, kcos, P + 1, grid, pspec));
        const spdy::SpecStep a{pvor, pdiv, pspec, vor, div, t, tr, ps, phis, d_tcorh, d_qcorh, phi, sdrag, dt, eps, wil, j1,
                               p->tab.ix == 4 * p->tab.iy, p->tmp_c, p->tmp_d};
        KERNEL(spdy::launch_spectral_step(p->dev, a, p->stream));
        return SPDY_OK;
    }
    RC(spdy_direct_batch_dev(p, P, ug, vg, pvor, pdiv, kcos, P + 1, grid, pspec));
    return spdy_spectral_step_dev(p, pvor, pdiv, pspec, vor, div, t, tr, ps, phis, d_tcorh, d_qcorh, sdrag, j1, dt, eps, wil, phi);
}

/* ---------------------------------------------------------------- output path */
int spdy_output_workspace(spdy_plan *p)
{
    NEED_DEVICE(p);
    if (p->out_grid) return SPDY_OK;
    NOT_CAPTURING(p, "allocating the output workspace (call spdy_output_workspace before the capture)");
    const int kx = p->tab.kx;
    void *ptr;
    RC(dev_alloc(p, (size_t)(5 * kx + 1) * grid_elems(p) * sizeof(double), &ptr));
    p->out_grid = static_cast<double *>(ptr);
    RC(dev_alloc(p, (size_t)(3 * kx + 1) * spec_elems(p) * sizeof(double), &ptr));
    p->out_spec = static_cast<double *>(ptr);
    return SPDY_OK;
}

int spdy_output_batch_dev(spdy_plan *p, const double *vor, const double *div, const double *t, const double *q, const double *phi,
                          const double *ps, float *u_out, float *v_out, float *t_out, float *q_out, float *phi_out, float *ps_out)
{
    NEED_DEVICE(p);
    if (!vor || !div || !t || !q || !phi || !ps || !u_out || !v_out || !t_out || !q_out || !phi_out || !ps_out)
        return fail(SPDY_ERR_ARG, "null device pointer");
    const int kx = p->tab.kx;
    if (p->max_batch < 3 * kx + 1) return fail(SPDY_ERR_ARG, "max_batch must be >= 3*kx+1 for the output batch");
    RC(spdy_output_workspace(p));
    const size_t gs = grid_elems(p), ss = spec_elems(p);
    // t, q, phi levels and ps into one stack: the whole snapshot is then one transform launch
    spdy::GatherOps g{};
    g.nops = 4;
    const double *src[4] = {t, q, phi, ps};
    for (int i = 0; i < 4; ++i) { g.nfld[i] = i < 3 ? kx : 1; g.src[i] = src[i]; g.dst[i] = p->out_spec + (size_t)i * kx * ss; }
    KERNEL(spdy::launch_gather_spectra(p->dev, g, p->stream));
    double *ug = p->out_grid, *vg = ug + (size_t)kx * gs, *plain = vg + (size_t)kx * gs;
    RC(spdy_inverse_batch_dev(p, kx, vor, div, ug, vg, 2, 3 * kx + 1, p->out_spec, nullptr, 1, plain));
    // input_output.f90:200-206: u, v, t as they are; q*1.0e-3; phi/grav; p0*exp(ps) -- then real(., sp)
    spdy::OutputCast c{};
    c.nops = 6;
    float *dst[6] = {u_out, v_out, t_out, q_out, phi_out, ps_out};
    const int kind[6] = {0, 0, 0, 1, 2, 3};
    const double fac[6] = {1.0, 1.0, 1.0, static_cast<double>(1.0e-3f), p->tab.grav, static_cast<double>(1.e+5f)};
    for (int i = 0; i < 6; ++i) {
        c.nfld[i] = i < 5 ? kx : 1; c.kind[i] = kind[i]; c.factor[i] = fac[i];
        c.src[i] = p->out_grid + (size_t)i * kx * gs; c.dst[i] = dst[i];
    }
    KERNEL(spdy::launch_output_cast(p->dev, c, p->stream));
    return SPDY_OK;
}

}  // extern "C"
