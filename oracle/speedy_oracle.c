/*
 * speedy_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, scalar, CPU restatement of the reference's grid<->spectral hot path
 * (samhatfield/speedy.f90), used ONLY as the checker in tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.  The product (speedy.f90_amd/csrc) never includes,
 * links or calls anything in this directory.
 *
 * Parity pin: this restatement is checked bit-for-bit / to <=1e-15 against the real
 * reference compiled with flang -O2 (oracle/_ref, see build_ref.sh) and against the
 * golden vectors committed under tests/golden (generated from that build by
 * tests/golden/make_golden.py).  The reference ships no tests of its own.
 *
 * All arrays are Fortran column-major exactly as the reference declares them:
 *   grid  g(ix,il)        -> g[i + ix*j]
 *   spec  s(mx,nx) complex-> s[2*(m + mx*n) + {0,1}]   (re,im interleaved)
 *   four  f(2*mx,il)      -> f[r + 2*mx*j]
 * The reference is "FP64" in storage only: unsuffixed literals / float() are evaluated in
 * float32 first.  Those spots are marked  [f32].
 * Build with  -O2 -ffp-contract=off  (no FMA contraction, no reassociation).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

typedef struct orc_ctx {
    int trunc, ix, iy, il, kx, nx, mx;
    /* geometry.f90 */
    double *sia_half, *coa_half, *cosgr, *cosgr2;       /* iy, il, il, il */
    double *hsg, *dhs, *fsg, *dhsr, *fsgr;              /* kx+1, kx.. */
    /* fourier.f90 / fftpack.f90 */
    double *work; int ifac[15];
    /* legendre.f90 */
    double *epsi, *repsi;                               /* (mx+1)*(nx+1) */
    double *wt;                                         /* iy */
    double *poly;                                       /* mx*nx*iy  (cpol without re/im duplicate) */
    int *nsh2;                                          /* nx */
    /* spectral.f90 */
    double *el2, *elm2, *el4, *trfilt, *gradx, *gradym, *gradyp, *uvdx, *uvdym, *uvdyp, *vddym, *vddyp;
    /* horizontal_diffusion.f90 / implicit.f90 */
    double *dmp, *dmpd, *dmps, *dmp1, *dmp1d, *dmp1s;
    double *tref, *tref1, *tref2, *tref3, *xc, *xd, *xj, *dhsx, *elz;
    int tail_ready;
    /* geometry.f90:89 ; geopotential.f90:14-15 ; horizontal_diffusion.f90:27-28 */
    double *coriol, *xgeop1, *xgeop2, *tcorv, *qcorv;
    int sigma_ready;
} orc_ctx;

static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

/* physical_constants.f90:16-25, dynamical_constants.f90:12-22: float32 literals widened [f32] */
static const double REARTH = (double)6.371e+6f;
static const double GRAV = (double)9.81f;
static const double CP = (double)1004.0f;
static double akap_(void) { return (double)(2.0f / 7.0f); }
static const double GAMMA_LAPSE = (double)6.0f;
static const double THD = (double)2.4f, THDD = (double)2.4f, THDS = (double)12.0f;
static const double ALPH = (double)0.5f;               /* params.f90:34 */
static const double OMEGA = (double)7.292e-05f;        /* physical_constants.f90:17 */
static const double HSCALE = (double)7.5f, HSHUM = (double)2.5f;   /* dynamical_constants.f90:13-14 */
static const double P0 = (double)1.e+5f;               /* physical_constants.f90:21 */

/* Functions of the half levels: geometry.f90:51-60, initialize_geopotential (geopotential.f90:18-31) and the
 * vertical orographic-correction profiles of initialize_horizontal_diffusion (horizontal_diffusion.f90:68-82). */
static void sigma_functions(orc_ctx *c)
{
    const int kx = c->kx;
    const double rgas = akap_() * CP;
    double rgam, qexp;
    int k;
    for (k = 0; k < kx; ++k) {                                         /* geometry.f90:51-54 */
        c->dhs[k] = c->hsg[k + 1] - c->hsg[k];
        c->fsg[k] = 0.5 * (c->hsg[k + 1] + c->hsg[k]);
    }
    for (k = 0; k < kx; ++k) {                                         /* :57-60 */
        c->dhsr[k] = 0.5 / c->dhs[k];
        c->fsgr[k] = akap_() / (2.0 * c->fsg[k]);
    }
    for (k = 1; k <= kx; ++k) {                                        /* geopotential.f90:26-29, 1-based k */
        c->xgeop1[k - 1] = rgas * log(c->hsg[k] / c->fsg[k - 1]);
        if (k != kx) c->xgeop2[k] = rgas * log(c->fsg[k] / c->hsg[k]);
    }
    rgam = rgas * GAMMA_LAPSE / (1000.0 * GRAV);                       /* horizontal_diffusion.f90:70-71 */
    qexp = HSCALE / HSHUM;
    c->tcorv[0] = 0.0;
    c->qcorv[0] = 0.0;
    if (kx > 1) c->qcorv[1] = 0.0;
    for (k = 2; k <= kx; ++k) {                                        /* :77-80 */
        c->tcorv[k - 1] = pow(c->fsg[k - 1], rgam);
        if (k > 2) c->qcorv[k - 1] = pow(c->fsg[k - 1], qexp);
    }
    c->sigma_ready = 1;
}

/* Test hook for level counts the reference has no sigma set for (geometry.f90:42-48 knows kx = 5, 7, 8): the
 * half levels are supplied, everything derived from them follows the reference recipes above.                 */
ORC_API void orc_set_sigma(orc_ctx *c, const double *hsg)
{
    int k;
    for (k = 0; k <= c->kx; ++k) c->hsg[k] = hsg[k];
    sigma_functions(c);
    c->tail_ready = 0;
}

/* ------------------------------------------------------------------ geometry.f90:35-89 */
static void init_geometry(orc_ctx *c)
{
    static const float hs8[9] = {0.000f, 0.050f, 0.140f, 0.260f, 0.420f, 0.600f, 0.770f, 0.900f, 1.000f};
    static const float hs7[8] = {0.020f, 0.140f, 0.260f, 0.420f, 0.600f, 0.770f, 0.900f, 1.000f};
    static const float hs5[6] = {0.000f, 0.150f, 0.350f, 0.650f, 0.900f, 1.000f};
    const float *hs = c->kx == 8 ? hs8 : c->kx == 7 ? hs7 : c->kx == 5 ? hs5 : NULL;
    int k, j;
    for (k = 0; k <= c->kx; ++k) c->hsg[k] = hs ? (double)hs[k] : 0.0;   /* geometry.f90:42-48 [f32] */
    if (hs) sigma_functions(c);
    for (j = 1; j <= c->iy; ++j) {                                        /* :66-86 */
        int jj = c->il + 1 - j;
        /* whole right-hand side is default-real: float32 pi literal, float32 cos  [f32] */
        float arg = 3.141592654f * ((float)j - 0.25f) / ((float)c->il + 0.5f);
        double s = (double)cosf(arg);
        double co = sqrt(1.0 - s * s);
        c->sia_half[j - 1] = s;
        c->coa_half[j - 1] = co;      /* declared (il) but only 1..iy are ever set (geometry.f90:69) */
        c->cosgr[j - 1] = 1.0 / co;  c->cosgr[jj - 1] = 1.0 / co;
        c->cosgr2[j - 1] = 1.0 / (co * co);  c->cosgr2[jj - 1] = 1.0 / (co * co);
        c->coriol[j - 1] = 2.0 * OMEGA * (-s);                            /* :70-71, :89: coriol = 2.0*omega*sia */
        c->coriol[jj - 1] = 2.0 * OMEGA * s;
    }
}

/* ------------------------------------------------------------------ fftpack.f90:1-67 rffti1 */
ORC_API void orc_rffti1(int n, double *wa, int *ifac)
{
    static const int ntryh[4] = {4, 2, 3, 5};
    int nl = n, nf = 0, j = 0, ntry = 0, i;
    while (nl != 1) {                       /* peel factors trying 4,2,3,5,7,9,...            */
        ntry = (j < 4) ? ntryh[j] : ntry + 2;
        ++j;
        while (nl % ntry == 0) {
            ++nf;
            ifac[nf + 1] = ntry;
            nl /= ntry;
            if (ntry == 2 && nf != 1) {     /* a 2 found late is moved to the front (:29-35) */
                for (i = nf; i >= 2; --i) ifac[i + 1] = ifac[i];
                ifac[2] = 2;
            }
        }
    }
    ifac[0] = n;
    ifac[1] = nf;
    {
        double tpi = (double)(8.0f * atanf(1.0f));          /* :39  [f32] */
        double argh = tpi / (double)n;
        int is = 0, l1 = 1, k1;
        for (k1 = 1; k1 <= nf - 1; ++k1) {
            int ip = ifac[k1 + 1], ld = 0, l2 = l1 * ip, ido = n / l2, jj;
            for (jj = 1; jj <= ip - 1; ++jj) {
                double argld, fi = 0.0;
                int ii, p = is;
                ld += l1;
                argld = (double)ld * argh;
                for (ii = 3; ii <= ido; ii += 2) {
                    double arg;
                    p += 2;
                    fi = fi + 1.0;
                    arg = fi * argld;
                    wa[p - 2] = cos(arg);
                    wa[p - 1] = sin(arg);
                }
                is += ido;
            }
            l1 = l2;
        }
    }
}

/* Radix kernels, FFTPACK half-complex storage.  cc is the stage input, ch the output.
 * Backward: cc(ido,ip,l1) -> ch(ido,l1,ip); forward: cc(ido,l1,ip) -> ch(ido,ip,l1).
 * p = 0-based index of a real part (1,3,5,...), p+1 its imaginary part; q/q+1 the mirrored
 * pair (ido-2-p, ido-1-p).  Twiddle for that pair: (w[p-1], w[p]).                         */
#define B_CC(i, j, k) cc[(i) + ido * ((j) + ip * (k))]
#define B_CH(i, k, j) ch[(i) + ido * ((k) + l1 * (j))]
#define F_CC(i, k, j) cc[(i) + ido * ((k) + l1 * (j))]
#define F_CH(i, j, k) ch[(i) + ido * ((j) + ip * (k))]

/* fftpack.f90:204-254 */
static void radb2(int ido, int l1, const double *cc, double *ch, const double *w1)
{
    const int ip = 2; int k, p;
    for (k = 0; k < l1; ++k) {
        B_CH(0, k, 0) = B_CC(0, 0, k) + B_CC(ido - 1, 1, k);
        B_CH(0, k, 1) = B_CC(0, 0, k) - B_CC(ido - 1, 1, k);
    }
    if (ido < 2) return;
    for (k = 0; k < l1; ++k)
        for (p = 1; p + 1 < ido; p += 2) {
            int q = ido - 2 - p;
            double tr2, ti2;
            B_CH(p, k, 0) = B_CC(p, 0, k) + B_CC(q, 1, k);
            tr2 = B_CC(p, 0, k) - B_CC(q, 1, k);
            B_CH(p + 1, k, 0) = B_CC(p + 1, 0, k) - B_CC(q + 1, 1, k);
            ti2 = B_CC(p + 1, 0, k) + B_CC(q + 1, 1, k);
            B_CH(p, k, 1) = w1[p - 1] * tr2 - w1[p] * ti2;
            B_CH(p + 1, k, 1) = w1[p - 1] * ti2 + w1[p] * tr2;
        }
    if (ido % 2 == 1) return;
    for (k = 0; k < l1; ++k) {
        B_CH(ido - 1, k, 0) = B_CC(ido - 1, 0, k) + B_CC(ido - 1, 0, k);
        B_CH(ido - 1, k, 1) = -(B_CC(0, 1, k) + B_CC(0, 1, k));
    }
}

/* fftpack.f90:256-323 ; taur,taui are float32 values [f32] */
static void radb3(int ido, int l1, const double *cc, double *ch, const double *w1, const double *w2)
{
    const int ip = 3; int k, p;
    const double taur = (double)(-.5f), taui = (double)(.5f * sqrtf(3.f));
    for (k = 0; k < l1; ++k) {
        double tr2 = B_CC(ido - 1, 1, k) + B_CC(ido - 1, 1, k);
        double cr2 = B_CC(0, 0, k) + taur * tr2;
        double ci3;
        B_CH(0, k, 0) = B_CC(0, 0, k) + tr2;
        ci3 = taui * (B_CC(0, 2, k) + B_CC(0, 2, k));
        B_CH(0, k, 1) = cr2 - ci3;
        B_CH(0, k, 2) = cr2 + ci3;
    }
    if (ido == 1) return;
    for (k = 0; k < l1; ++k)
        for (p = 1; p + 1 < ido; p += 2) {
            int q = ido - 2 - p;
            double tr2 = B_CC(p, 2, k) + B_CC(q, 1, k);
            double cr2 = B_CC(p, 0, k) + taur * tr2;
            double ti2, ci2, cr3, ci3, dr2, dr3, di2, di3;
            B_CH(p, k, 0) = B_CC(p, 0, k) + tr2;
            ti2 = B_CC(p + 1, 2, k) - B_CC(q + 1, 1, k);
            ci2 = B_CC(p + 1, 0, k) + taur * ti2;
            B_CH(p + 1, k, 0) = B_CC(p + 1, 0, k) + ti2;
            cr3 = taui * (B_CC(p, 2, k) - B_CC(q, 1, k));
            ci3 = taui * (B_CC(p + 1, 2, k) + B_CC(q + 1, 1, k));
            dr2 = cr2 - ci3;  dr3 = cr2 + ci3;
            di2 = ci2 + cr3;  di3 = ci2 - cr3;
            B_CH(p, k, 1) = w1[p - 1] * dr2 - w1[p] * di2;
            B_CH(p + 1, k, 1) = w1[p - 1] * di2 + w1[p] * dr2;
            B_CH(p, k, 2) = w2[p - 1] * dr3 - w2[p] * di3;
            B_CH(p + 1, k, 2) = w2[p - 1] * di3 + w2[p] * dr3;
        }
}

/* fftpack.f90:325-424 ; sqrt2 float32 [f32] */
static void radb4(int ido, int l1, const double *cc, double *ch,
                  const double *w1, const double *w2, const double *w3)
{
    const int ip = 4; int k, p;
    const double sqrt2 = (double)sqrtf(2.f);
    for (k = 0; k < l1; ++k) {
        double tr1 = B_CC(0, 0, k) - B_CC(ido - 1, 3, k);
        double tr2 = B_CC(0, 0, k) + B_CC(ido - 1, 3, k);
        double tr3 = B_CC(ido - 1, 1, k) + B_CC(ido - 1, 1, k);
        double tr4 = B_CC(0, 2, k) + B_CC(0, 2, k);
        B_CH(0, k, 0) = tr2 + tr3;
        B_CH(0, k, 1) = tr1 - tr4;
        B_CH(0, k, 2) = tr2 - tr3;
        B_CH(0, k, 3) = tr1 + tr4;
    }
    if (ido < 2) return;
    for (k = 0; k < l1; ++k)
        for (p = 1; p + 1 < ido; p += 2) {
            int q = ido - 2 - p;
            double ti1 = B_CC(p + 1, 0, k) + B_CC(q + 1, 3, k);
            double ti2 = B_CC(p + 1, 0, k) - B_CC(q + 1, 3, k);
            double ti3 = B_CC(p + 1, 2, k) - B_CC(q + 1, 1, k);
            double tr4 = B_CC(p + 1, 2, k) + B_CC(q + 1, 1, k);
            double tr1 = B_CC(p, 0, k) - B_CC(q, 3, k);
            double tr2 = B_CC(p, 0, k) + B_CC(q, 3, k);
            double ti4 = B_CC(p, 2, k) - B_CC(q, 1, k);
            double tr3 = B_CC(p, 2, k) + B_CC(q, 1, k);
            double cr3, ci3, cr2, cr4, ci2, ci4;
            B_CH(p, k, 0) = tr2 + tr3;
            cr3 = tr2 - tr3;
            B_CH(p + 1, k, 0) = ti2 + ti3;
            ci3 = ti2 - ti3;
            cr2 = tr1 - tr4;  cr4 = tr1 + tr4;
            ci2 = ti1 + ti4;  ci4 = ti1 - ti4;
            B_CH(p, k, 1) = w1[p - 1] * cr2 - w1[p] * ci2;
            B_CH(p + 1, k, 1) = w1[p - 1] * ci2 + w1[p] * cr2;
            B_CH(p, k, 2) = w2[p - 1] * cr3 - w2[p] * ci3;
            B_CH(p + 1, k, 2) = w2[p - 1] * ci3 + w2[p] * cr3;
            B_CH(p, k, 3) = w3[p - 1] * cr4 - w3[p] * ci4;
            B_CH(p + 1, k, 3) = w3[p - 1] * ci4 + w3[p] * cr4;
        }
    if (ido % 2 == 1) return;
    for (k = 0; k < l1; ++k) {
        double ti1 = B_CC(0, 1, k) + B_CC(0, 3, k);
        double ti2 = B_CC(0, 3, k) - B_CC(0, 1, k);
        double tr1 = B_CC(ido - 1, 0, k) - B_CC(ido - 1, 2, k);
        double tr2 = B_CC(ido - 1, 0, k) + B_CC(ido - 1, 2, k);
        B_CH(ido - 1, k, 0) = tr2 + tr2;
        B_CH(ido - 1, k, 1) = sqrt2 * (tr1 - ti1);
        B_CH(ido - 1, k, 2) = ti2 + ti2;
        B_CH(ido - 1, k, 3) = -sqrt2 * (tr1 + ti1);
    }
}

/* fftpack.f90:722-772 */
static void radf2(int ido, int l1, const double *cc, double *ch, const double *w1)
{
    const int ip = 2; int k, p;
    for (k = 0; k < l1; ++k) {
        F_CH(0, 0, k) = F_CC(0, k, 0) + F_CC(0, k, 1);
        F_CH(ido - 1, 1, k) = F_CC(0, k, 0) - F_CC(0, k, 1);
    }
    if (ido < 2) return;
    for (k = 0; k < l1; ++k)
        for (p = 1; p + 1 < ido; p += 2) {
            int q = ido - 2 - p;
            double tr2 = w1[p - 1] * F_CC(p, k, 1) + w1[p] * F_CC(p + 1, k, 1);
            double ti2 = w1[p - 1] * F_CC(p + 1, k, 1) - w1[p] * F_CC(p, k, 1);
            F_CH(p + 1, 0, k) = F_CC(p + 1, k, 0) + ti2;
            F_CH(q + 1, 1, k) = ti2 - F_CC(p + 1, k, 0);
            F_CH(p, 0, k) = F_CC(p, k, 0) + tr2;
            F_CH(q, 1, k) = F_CC(p, k, 0) - tr2;
        }
    if (ido % 2 == 1) return;
    for (k = 0; k < l1; ++k) {
        F_CH(0, 1, k) = -F_CC(ido - 1, k, 1);
        F_CH(ido - 1, 0, k) = F_CC(ido - 1, k, 0);
    }
}

/* fftpack.f90:774-840 */
static void radf3(int ido, int l1, const double *cc, double *ch, const double *w1, const double *w2)
{
    const int ip = 3; int k, p;
    const double taur = (double)(-.5f), taui = (double)(.5f * sqrtf(3.f));
    for (k = 0; k < l1; ++k) {
        double cr2 = F_CC(0, k, 1) + F_CC(0, k, 2);
        F_CH(0, 0, k) = F_CC(0, k, 0) + cr2;
        F_CH(0, 2, k) = taui * (F_CC(0, k, 2) - F_CC(0, k, 1));
        F_CH(ido - 1, 1, k) = F_CC(0, k, 0) + taur * cr2;
    }
    if (ido == 1) return;
    for (k = 0; k < l1; ++k)
        for (p = 1; p + 1 < ido; p += 2) {
            int q = ido - 2 - p;
            double dr2 = w1[p - 1] * F_CC(p, k, 1) + w1[p] * F_CC(p + 1, k, 1);
            double di2 = w1[p - 1] * F_CC(p + 1, k, 1) - w1[p] * F_CC(p, k, 1);
            double dr3 = w2[p - 1] * F_CC(p, k, 2) + w2[p] * F_CC(p + 1, k, 2);
            double di3 = w2[p - 1] * F_CC(p + 1, k, 2) - w2[p] * F_CC(p, k, 2);
            double cr2 = dr2 + dr3, ci2 = di2 + di3, tr2, ti2, tr3, ti3;
            F_CH(p, 0, k) = F_CC(p, k, 0) + cr2;
            F_CH(p + 1, 0, k) = F_CC(p + 1, k, 0) + ci2;
            tr2 = F_CC(p, k, 0) + taur * cr2;
            ti2 = F_CC(p + 1, k, 0) + taur * ci2;
            tr3 = taui * (di2 - di3);
            ti3 = taui * (dr3 - dr2);
            F_CH(p, 2, k) = tr2 + tr3;
            F_CH(q, 1, k) = tr2 - tr3;
            F_CH(p + 1, 2, k) = ti2 + ti3;
            F_CH(q + 1, 1, k) = ti3 - ti2;
        }
}

/* fftpack.f90:842-936 ; hsqt2 float32 [f32] */
static void radf4(int ido, int l1, const double *cc, double *ch,
                  const double *w1, const double *w2, const double *w3)
{
    const int ip = 4; int k, p;
    const double hsqt2 = (double)(.5f * sqrtf(2.f));
    for (k = 0; k < l1; ++k) {
        double tr1 = F_CC(0, k, 1) + F_CC(0, k, 3);
        double tr2 = F_CC(0, k, 0) + F_CC(0, k, 2);
        F_CH(0, 0, k) = tr1 + tr2;
        F_CH(ido - 1, 3, k) = tr2 - tr1;
        F_CH(ido - 1, 1, k) = F_CC(0, k, 0) - F_CC(0, k, 2);
        F_CH(0, 2, k) = F_CC(0, k, 3) - F_CC(0, k, 1);
    }
    if (ido < 2) return;
    for (k = 0; k < l1; ++k)
        for (p = 1; p + 1 < ido; p += 2) {
            int q = ido - 2 - p;
            double cr2 = w1[p - 1] * F_CC(p, k, 1) + w1[p] * F_CC(p + 1, k, 1);
            double ci2 = w1[p - 1] * F_CC(p + 1, k, 1) - w1[p] * F_CC(p, k, 1);
            double cr3 = w2[p - 1] * F_CC(p, k, 2) + w2[p] * F_CC(p + 1, k, 2);
            double ci3 = w2[p - 1] * F_CC(p + 1, k, 2) - w2[p] * F_CC(p, k, 2);
            double cr4 = w3[p - 1] * F_CC(p, k, 3) + w3[p] * F_CC(p + 1, k, 3);
            double ci4 = w3[p - 1] * F_CC(p + 1, k, 3) - w3[p] * F_CC(p, k, 3);
            double tr1 = cr2 + cr4, tr4 = cr4 - cr2, ti1 = ci2 + ci4, ti4 = ci2 - ci4;
            double ti2 = F_CC(p + 1, k, 0) + ci3, ti3 = F_CC(p + 1, k, 0) - ci3;
            double tr2 = F_CC(p, k, 0) + cr3, tr3 = F_CC(p, k, 0) - cr3;
            F_CH(p, 0, k) = tr1 + tr2;
            F_CH(q, 3, k) = tr2 - tr1;
            F_CH(p + 1, 0, k) = ti1 + ti2;
            F_CH(q + 1, 3, k) = ti1 - ti2;
            F_CH(p, 2, k) = ti4 + tr3;
            F_CH(q, 1, k) = tr3 - ti4;
            F_CH(p + 1, 2, k) = tr4 + ti3;
            F_CH(q + 1, 1, k) = tr4 - ti3;
        }
    if (ido % 2 == 1) return;
    for (k = 0; k < l1; ++k) {
        double ti1 = -hsqt2 * (F_CC(ido - 1, k, 1) + F_CC(ido - 1, k, 3));
        double tr1 = hsqt2 * (F_CC(ido - 1, k, 1) - F_CC(ido - 1, k, 3));
        F_CH(ido - 1, 0, k) = tr1 + F_CC(ido - 1, k, 0);
        F_CH(ido - 1, 2, k) = F_CC(ido - 1, k, 0) - tr1;
        F_CH(0, 1, k) = ti1 - F_CC(ido - 1, k, 2);
        F_CH(0, 3, k) = ti1 + F_CC(ido - 1, k, 2);
    }
}

/* fftpack.f90:69-134 rfftb1: backward driver, factors in table order, ping-pong c<->ch.
 * Only radix 2/3/4 exist here: the reference never reaches radix 5/general for N=96/192. */
ORC_API int orc_rfftb1(int n, double *c, double *ch, const double *wa, const int *ifac)
{
    int nf = ifac[1], na = 0, l1 = 1, iw = 0, k1;
    for (k1 = 0; k1 < nf; ++k1) {
        int ip = ifac[k1 + 2], l2 = ip * l1, ido = n / l2;
        const double *in = na ? ch : c;
        double *out = na ? c : ch;
        if (ip == 4) radb4(ido, l1, in, out, wa + iw, wa + iw + ido, wa + iw + 2 * ido);
        else if (ip == 2) radb2(ido, l1, in, out, wa + iw);
        else if (ip == 3) radb3(ido, l1, in, out, wa + iw, wa + iw + ido);
        else return -1;
        na = 1 - na;
        l1 = l2;
        iw += (ip - 1) * ido;
    }
    if (na) memcpy(c, ch, sizeof(double) * (size_t)n);
    return 0;
}

/* fftpack.f90:136-202 rfftf1: forward driver, factors in reverse order, twiddles walked down */
ORC_API int orc_rfftf1(int n, double *c, double *ch, const double *wa, const int *ifac)
{
    int nf = ifac[1], na = 1, l2 = n, iw = n - 1, k1;
    for (k1 = 1; k1 <= nf; ++k1) {
        int ip = ifac[nf - k1 + 2], l1 = l2 / ip, ido = n / l2;
        const double *in; double *out;
        iw -= (ip - 1) * ido;
        na = 1 - na;
        in = na ? ch : c;
        out = na ? c : ch;
        if (ip == 4) radf4(ido, l1, in, out, wa + iw, wa + iw + ido, wa + iw + 2 * ido);
        else if (ip == 2) radf2(ido, l1, in, out, wa + iw);
        else if (ip == 3) radf3(ido, l1, in, out, wa + iw, wa + iw + ido);
        else return -1;
        l2 = l1;
    }
    if (na != 1) memcpy(c, ch, sizeof(double) * (size_t)n);
    return 0;
}

/* ------------------------------------------------------------------ fourier.f90:23-53 */
ORC_API void orc_fourier_inv(const orc_ctx *c, const double *f, int kcos, double *g)
{
    const int ix = c->ix, il = c->il, tm = 2 * c->mx;
    double *fvar = dalloc((size_t)ix), *ch = dalloc((size_t)ix);
    int j, m, i;
    for (j = 0; j < il; ++j) {
        const double *in = f + (size_t)tm * j;
        fvar[0] = in[0];
        for (m = 3; m <= tm; ++m) fvar[m - 2] = in[m - 1];     /* drops Im(m'=0) */
        for (m = tm; m <= ix; ++m) fvar[m - 1] = 0.0;
        orc_rfftb1(ix, fvar, ch, c->work, c->ifac);
        if (kcos == 1) for (i = 0; i < ix; ++i) g[i + (size_t)ix * j] = fvar[i];
        else for (i = 0; i < ix; ++i) g[i + (size_t)ix * j] = fvar[i] * c->cosgr[j];
    }
    free(fvar); free(ch);
}

/* ------------------------------------------------------------------ fourier.f90:56-82 */
ORC_API void orc_fourier_dir(const orc_ctx *c, const double *g, double *f)
{
    const int ix = c->ix, il = c->il, tm = 2 * c->mx;
    double *fvar = dalloc((size_t)ix), *ch = dalloc((size_t)ix);
    const double scale = (double)(1.0f / (float)ix);            /* fourier.f90:72 [f32] */
    int j, m;
    for (j = 0; j < il; ++j) {
        double *out = f + (size_t)tm * j;
        memcpy(fvar, g + (size_t)ix * j, sizeof(double) * (size_t)ix);
        orc_rfftf1(ix, fvar, ch, c->work, c->ifac);
        out[0] = fvar[0] * scale;
        out[1] = 0.0;
        for (m = 3; m <= tm; ++m) out[m - 1] = fvar[m - 2] * scale;
    }
    free(fvar); free(ch);
}

/* ------------------------------------------------------------------ legendre.f90:158-191 */
static void gauss_weights(const orc_ctx *c, double *w)
{
    const int iy = c->iy, n = 2 * iy;
    double z1 = 2.0, pp = 0.0;
    int i, j;
    for (i = 1; i <= iy; ++i) {
        double z = cos(3.141592654 * ((double)i - 0.25) / ((double)n + 0.5));
        while (fabs(z - z1) > 2.220446049250313e-16) {
            double p1 = 1.0, p2 = 0.0, p3;
            for (j = 1; j <= n; ++j) {
                p3 = p2;
                p2 = p1;
                p1 = ((2.0 * (double)j - 1.0) * z * p2 - ((double)j - 1.0) * p3) / (double)j;
            }
            pp = (double)n * (z * p1 - p2) / (z * z - 1.0);
            z1 = z;
            z = z1 - p1 / pp;
        }
        w[i - 1] = 2.0 / ((1.0 - z * z) * (pp * pp));
    }
}

/* ------------------------------------------------------------------ legendre.f90:194-237 */
static void legendre_poly(const orc_ctx *c, int j, double *poly /* mx*nx */)
{
    const int mx = c->mx, nx = c->nx, m1 = mx + 1;
    const double small = (double)1.e-30f;                       /* :200 [f32] */
    const double y = c->coa_half[j], x = c->sia_half[j];
    double *alp = dalloc((size_t)m1 * nx);
    int m, n;
#define ALP(m, n) alp[(m) + m1 * (n)]
#define EPS(m, n) c->epsi[(m) + m1 * (n)]
#define REPS(m, n) c->repsi[(m) + m1 * (n)]
    ALP(0, 0) = (double)sqrtf(0.5f);                            /* :212 [f32] */
    for (m = 1; m <= mx; ++m) {
        /* consq(m) = sqrt(0.5*(2.0*float(m)+1.0)/float(m)) entirely float32  (:208) [f32] */
        float fm = (float)m;
        double consq = (double)sqrtf(0.5f * (2.0f * fm + 1.0f) / fm);
        ALP(m, 0) = consq * y * ALP(m - 1, 0);
    }
    for (m = 0; m <= mx; ++m) ALP(m, 1) = (x * ALP(m, 0)) * REPS(m, 1);
    for (n = 2; n < nx; ++n)
        for (m = 0; m <= mx; ++m)
            ALP(m, n) = (x * ALP(m, n - 1) - EPS(m, n - 1) * ALP(m, n - 2)) * REPS(m, n);
    for (n = 0; n < nx; ++n)
        for (m = 0; m < mx; ++m) {
            double v = ALP(m, n);
            poly[m + mx * n] = (fabs(v) <= small) ? 0.0 : v;
        }
#undef ALP
    free(alp);
}

/* ------------------------------------------------------------------ legendre.f90:23-71 */
static void init_legendre(orc_ctx *c)
{
    const int mx = c->mx, nx = c->nx, m1 = mx + 1;
    int m, n, j;
    gauss_weights(c, c->wt);
    for (n = 1; n <= nx; ++n) {
        c->nsh2[n - 1] = 0;
        for (m = 1; m <= mx; ++m)
            if ((m - 1) + n - 1 <= c->trunc + 1 || c->ix != 4 * c->iy) c->nsh2[n - 1] += 2;
    }
    for (m = 1; m <= mx + 1; ++m)
        for (n = 1; n <= nx + 1; ++n) {
            double emm2 = (double)((float)(m - 1) * (float)(m - 1));
            double ell2 = (double)((float)(n + m - 2) * (float)(n + m - 2));
            double e;
            if (n == nx + 1) e = 0.0;
            else if (n == 1 && m == 1) e = 0.0;
            else e = sqrt((ell2 - emm2) / (4.0 * ell2 - 1.0));
            c->epsi[(m - 1) + m1 * (n - 1)] = e;
            c->repsi[(m - 1) + m1 * (n - 1)] = (e > 0.0) ? 1.0 / e : 0.0;
        }
    for (j = 0; j < c->iy; ++j) legendre_poly(c, j, c->poly + (size_t)mx * nx * j);
}
#define POLY(m, n, j) c->poly[(m) + mx * ((n) + nx * (j))]

/* ------------------------------------------------------------------ legendre.f90:74-111
 * in: s(2mx,nx) real view; out: f(2mx,il).  cpol(2m-1,n,j)=cpol(2m,n,j)=poly(m,n,j).      */
ORC_API void orc_legendre_inv(const orc_ctx *c, const double *s, double *f)
{
    const int mx = c->mx, nx = c->nx, tm = 2 * mx, il = c->il;
    double *even = dalloc((size_t)tm), *odd = dalloc((size_t)tm);
    int j, n, r;
    for (j = 0; j < c->iy; ++j) {
        int j1 = il - 1 - j;
        for (r = 0; r < tm; ++r) { even[r] = 0.0; odd[r] = 0.0; }
        for (n = 0; n < nx; n += 2)
            for (r = 0; r < c->nsh2[n]; ++r) even[r] = even[r] + s[r + tm * n] * POLY(r >> 1, n, j);
        for (n = 1; n < nx; n += 2)
            for (r = 0; r < c->nsh2[n]; ++r) odd[r] = odd[r] + s[r + tm * n] * POLY(r >> 1, n, j);
        for (r = 0; r < tm; ++r) {
            f[r + (size_t)tm * j1] = even[r] + odd[r];
            f[r + (size_t)tm * j] = even[r] - odd[r];
        }
    }
    free(even); free(odd);
}

/* ------------------------------------------------------------------ legendre.f90:114-155 */
ORC_API void orc_legendre_dir(const orc_ctx *c, const double *f, double *s)
{
    const int mx = c->mx, nx = c->nx, tm = 2 * mx, il = c->il, iy = c->iy;
    double *even = dalloc((size_t)tm * iy), *odd = dalloc((size_t)tm * iy);
    int j, n, r;
    for (r = 0; r < tm * nx; ++r) s[r] = 0.0;
    for (j = 0; j < iy; ++j) {
        int j1 = il - 1 - j;
        for (r = 0; r < tm; ++r) {
            even[r + tm * j] = (f[r + (size_t)tm * j1] + f[r + (size_t)tm * j]) * c->wt[j];
            odd[r + tm * j] = (f[r + (size_t)tm * j1] - f[r + (size_t)tm * j]) * c->wt[j];
        }
    }
    for (n = 0; n < c->trunc + 1; ++n) {
        const double *src = (n % 2 == 0) ? even : odd;
        for (r = 0; r < c->nsh2[n]; ++r) {
            double acc = 0.0;                           /* dot_product: sequential over j */
            for (j = 0; j < iy; ++j) acc = acc + POLY(r >> 1, n, j) * src[r + tm * j];
            s[r + tm * n] = acc;
        }
    }
    free(even); free(odd);
}

/* ------------------------------------------------------------------ spectral.f90:98-122 */
ORC_API void orc_spec_to_grid(const orc_ctx *c, const double *s, int kcos, double *g)
{
    double *f = dalloc((size_t)2 * c->mx * c->il);
    orc_legendre_inv(c, s, f);
    orc_fourier_inv(c, f, kcos, g);
    free(f);
}

ORC_API void orc_grid_to_spec(const orc_ctx *c, const double *g, double *s)
{
    double *f = dalloc((size_t)2 * c->mx * c->il);
    orc_fourier_dir(c, g, f);
    orc_legendre_dir(c, f, s);
    free(f);
}

/* ------------------------------------------------------------------ spectral.f90:20-82 */
static void init_spectral(orc_ctx *c)
{
    const int mx = c->mx, nx = c->nx, m1 = mx + 1;
    int m, n;
#define T2(a, m, n) c->a[(m) + mx * (n)]
    for (n = 0; n < nx; ++n)
        for (m = 0; m < mx; ++m) {
            int l = m + n;
            T2(el2, m, n) = (double)(float)(l * (l + 1)) / (REARTH * REARTH);
            T2(el4, m, n) = T2(el2, m, n) * T2(el2, m, n);
            T2(trfilt, m, n) = (l <= c->trunc) ? 1.0 : 0.0;
            T2(elm2, m, n) = (l == 0) ? 0.0 : 1.0 / T2(el2, m, n);
        }
    for (m = 0; m < mx; ++m)
        for (n = 0; n < nx; ++n) {
            double el1 = (double)(float)(m + n);
            double fm = (double)(float)m;
            if (n == 0) {
                c->gradx[m] = fm / REARTH;
                T2(uvdx, m, 0) = -REARTH / (double)(float)(m + 1);
                T2(uvdym, m, 0) = 0.0;
                T2(vddym, m, 0) = 0.0;
                T2(gradym, m, 0) = 0.0;                 /* left unset by the reference, never read */
            } else {
                T2(uvdx, m, n) = -REARTH * fm / (el1 * (el1 + 1.0));
                T2(gradym, m, n) = (el1 - 1.0) * EPS(m, n) / REARTH;
                T2(uvdym, m, n) = -REARTH * EPS(m, n) / el1;
                T2(vddym, m, n) = (el1 + 1.0) * EPS(m, n) / REARTH;
            }
            T2(gradyp, m, n) = (el1 + 2.0) * EPS(m, n + 1) / REARTH;
            T2(uvdyp, m, n) = -REARTH * EPS(m, n + 1) / (el1 + 1.0);
            T2(vddyp, m, n) = el1 * EPS(m, n + 1) / REARTH;
        }
    (void)m1;
}

/* complex helpers on interleaved (re,im) pairs; Fortran semantics: real*complex scales both
 * parts; complex*(0,1) is a full complex multiply.                                          */
typedef struct { double re, im; } cplx;
static cplx cget(const double *a, int idx) { cplx z = {a[2 * idx], a[2 * idx + 1]}; return z; }
static void cset(double *a, int idx, cplx z) { a[2 * idx] = z.re; a[2 * idx + 1] = z.im; }
static cplx rmul(double r, cplx z) { cplx o = {r * z.re, r * z.im}; return o; }
static cplx cadd(cplx a, cplx b) { cplx o = {a.re + b.re, a.im + b.im}; return o; }
static cplx csub(cplx a, cplx b) { cplx o = {a.re - b.re, a.im - b.im}; return o; }
static cplx cneg(cplx a) { cplx o = {-a.re, -a.im}; return o; }
static cplx muli(cplx a) { cplx o = {a.re * 0.0 - a.im * 1.0, a.re * 1.0 + a.im * 0.0}; return o; }
#define SI(m, n) ((m) + mx * (n))

/* spectral.f90:84-96, 229-233 */
ORC_API void orc_laplacian(const orc_ctx *c, const double *a, double *o)
{ int i, t = c->mx * c->nx; for (i = 0; i < t; ++i) cset(o, i, rmul(c->el2[i], cneg(cget(a, i)))); }
ORC_API void orc_inverse_laplacian(const orc_ctx *c, const double *a, double *o)
{ int i, t = c->mx * c->nx; for (i = 0; i < t; ++i) cset(o, i, rmul(c->elm2[i], cneg(cget(a, i)))); }
ORC_API void orc_trunct(const orc_ctx *c, double *a)
{ int i, t = c->mx * c->nx; for (i = 0; i < t; ++i) cset(a, i, rmul(c->trfilt[i], cget(a, i))); }

/* spectral.f90:124-144 */
ORC_API void orc_grad(const orc_ctx *c, const double *psi, double *psdx, double *psdy)
{
    const int mx = c->mx, nx = c->nx, tr1 = c->trunc + 1;
    int m, n;
    for (n = 0; n < nx; ++n)
        for (m = 0; m < mx; ++m) cset(psdx, SI(m, n), muli(rmul(c->gradx[m], cget(psi, SI(m, n)))));
    for (m = 0; m < mx; ++m) {
        cset(psdy, SI(m, 0), rmul(T2(gradyp, m, 0), cget(psi, SI(m, 1))));
        cset(psdy, SI(m, nx - 1), rmul(-T2(gradym, m, nx - 1), cget(psi, SI(m, tr1 - 1))));
    }
    for (n = 1; n < tr1; ++n)
        for (m = 0; m < mx; ++m)
            cset(psdy, SI(m, n), cadd(rmul(-T2(gradym, m, n), cget(psi, SI(m, n - 1))),
                                      rmul(T2(gradyp, m, n), cget(psi, SI(m, n + 1)))));
}

/* spectral.f90:146-171 */
ORC_API void orc_vds(const orc_ctx *c, const double *ucosm, const double *vcosm, double *vorm, double *divm)
{
    const int mx = c->mx, nx = c->nx, tr1 = c->trunc + 1;
    int m, n;
    for (m = 0; m < mx; ++m) {
        cplx zp0 = muli(rmul(c->gradx[m], cget(ucosm, SI(m, 0))));
        cplx zc0 = muli(rmul(c->gradx[m], cget(vcosm, SI(m, 0))));
        cset(vorm, SI(m, 0), csub(zc0, rmul(T2(vddyp, m, 0), cget(ucosm, SI(m, 1)))));
        cset(vorm, SI(m, nx - 1), rmul(T2(vddym, m, nx - 1), cget(ucosm, SI(m, tr1 - 1))));
        cset(divm, SI(m, 0), cadd(zp0, rmul(T2(vddyp, m, 0), cget(vcosm, SI(m, 1)))));
        cset(divm, SI(m, nx - 1), rmul(-T2(vddym, m, nx - 1), cget(vcosm, SI(m, tr1 - 1))));
    }
    for (n = 1; n < tr1; ++n)
        for (m = 0; m < mx; ++m) {
            cplx zp = muli(rmul(c->gradx[m], cget(ucosm, SI(m, n))));
            cplx zc = muli(rmul(c->gradx[m], cget(vcosm, SI(m, n))));
            cset(vorm, SI(m, n), cadd(csub(rmul(T2(vddym, m, n), cget(ucosm, SI(m, n - 1))),
                                           rmul(T2(vddyp, m, n), cget(ucosm, SI(m, n + 1)))), zc));
            cset(divm, SI(m, n), cadd(cadd(rmul(-T2(vddym, m, n), cget(vcosm, SI(m, n - 1))),
                                           rmul(T2(vddyp, m, n), cget(vcosm, SI(m, n + 1)))), zp));
        }
}

/* spectral.f90:173-196 */
ORC_API void orc_uvspec(const orc_ctx *c, const double *vorm, const double *divm, double *ucosm, double *vcosm)
{
    const int mx = c->mx, nx = c->nx, tr1 = c->trunc + 1;
    int m, n;
    for (m = 0; m < mx; ++m) {
        cplx zp0 = muli(rmul(T2(uvdx, m, 0), cget(vorm, SI(m, 0))));
        cplx zc0 = muli(rmul(T2(uvdx, m, 0), cget(divm, SI(m, 0))));
        cset(ucosm, SI(m, 0), csub(zc0, rmul(T2(uvdyp, m, 0), cget(vorm, SI(m, 1)))));
        cset(ucosm, SI(m, nx - 1), rmul(T2(uvdym, m, nx - 1), cget(vorm, SI(m, tr1 - 1))));
        cset(vcosm, SI(m, 0), cadd(zp0, rmul(T2(uvdyp, m, 0), cget(divm, SI(m, 1)))));
        cset(vcosm, SI(m, nx - 1), rmul(-T2(uvdym, m, nx - 1), cget(divm, SI(m, tr1 - 1))));
    }
    for (n = 1; n < tr1; ++n)
        for (m = 0; m < mx; ++m) {
            cplx zp = muli(rmul(T2(uvdx, m, n), cget(vorm, SI(m, n))));
            cplx zc = muli(rmul(T2(uvdx, m, n), cget(divm, SI(m, n))));
            cset(vcosm, SI(m, n), cadd(cadd(rmul(-T2(uvdym, m, n), cget(divm, SI(m, n - 1))),
                                            rmul(T2(uvdyp, m, n), cget(divm, SI(m, n + 1)))), zp));
            cset(ucosm, SI(m, n), cadd(csub(rmul(T2(uvdym, m, n), cget(vorm, SI(m, n - 1))),
                                            rmul(T2(uvdyp, m, n), cget(vorm, SI(m, n + 1)))), zc));
        }
}

/* spectral.f90:198-227 */
ORC_API void orc_vdspec(const orc_ctx *c, const double *ug, const double *vg, double *vorm, double *divm, int kcos)
{
    const int ix = c->ix, il = c->il, ns = 2 * c->mx * c->nx;
    double *u1 = dalloc((size_t)ix * il), *v1 = dalloc((size_t)ix * il);
    double *su = dalloc((size_t)ns), *sv = dalloc((size_t)ns);
    const double *sc = (kcos == 2) ? c->cosgr : c->cosgr2;
    int i, j;
    for (j = 0; j < il; ++j)
        for (i = 0; i < ix; ++i) {
            u1[i + ix * j] = ug[i + ix * j] * sc[j];
            v1[i + ix * j] = vg[i + ix * j] * sc[j];
        }
    orc_grid_to_spec(c, u1, su);
    orc_grid_to_spec(c, v1, sv);
    orc_vds(c, su, sv, vorm, divm);
    free(u1); free(v1); free(su); free(sv);
}

/* ------------------------------------------------------------------ horizontal_diffusion.f90:36-82 */
static void init_hdiff(orc_ctx *c)
{
    const int mx = c->mx, nx = c->nx;
    const double hdiff = 1.0 / (THD * 3600.0), hdifd = 1.0 / (THDD * 3600.0), hdifs = 1.0 / (THDS * 3600.0);
    const double rlap = (double)(1.0f / (float)(c->trunc * (c->trunc + 1)));   /* :55 [f32] */
    int j, k;
    for (j = 0; j < nx; ++j)
        for (k = 0; k < mx; ++k) {
            double twn = (double)(float)(k + j);
            double elap = twn * (twn + 1.0) * rlap;
            double e2 = elap * elap, elapn = e2 * e2;            /* elap**4, integer power */
            c->dmp[k + mx * j] = hdiff * elapn;
            c->dmpd[k + mx * j] = hdifd * elapn;
            c->dmps[k + mx * j] = hdifs * elap;
        }
}

/* horizontal_diffusion.f90:86-105: fdt_out = (fdt_in - dmp*field)*dmp1, nlev levels */
ORC_API void orc_hdiff(const orc_ctx *c, int nlev, const double *field, const double *fdt_in,
                       const double *dmp, const double *dmp1, double *fdt_out)
{
    const int t = c->mx * c->nx;
    int k, i;
    for (k = 0; k < nlev; ++k)
        for (i = 0; i < t; ++i) {
            cplx fl = cget(field, k * t + i), fd = cget(fdt_in, k * t + i);
            cset(fdt_out, k * t + i, rmul(dmp1[i], csub(fd, rmul(dmp[i], fl))));
        }
}

/* ------------------------------------------------------------------ matrix_inversion.f90:10-133
 * Crout LU with implicit scaling + partial pivoting, then column-by-column back substitution. */
static int lu_decompose(double *a, int n, int *indx)
{
    double vv[128];
    int i, j, k, imax = 0;
    for (i = 0; i < n; ++i) {
        double big = 0.0;
        for (j = 0; j < n; ++j) if (fabs(a[i + n * j]) > big) big = fabs(a[i + n * j]);
        if (big == 0.0) return -1;
        vv[i] = 1.0 / big;
    }
    for (j = 0; j < n; ++j) {
        double big = 0.0;
        for (i = 0; i < j; ++i) {
            double sum = a[i + n * j];
            for (k = 0; k < i; ++k) sum = sum - a[i + n * k] * a[k + n * j];
            a[i + n * j] = sum;
        }
        for (i = j; i < n; ++i) {
            double sum = a[i + n * j], dum;
            for (k = 0; k < j; ++k) sum = sum - a[i + n * k] * a[k + n * j];
            a[i + n * j] = sum;
            dum = vv[i] * fabs(sum);
            if (dum >= big) { imax = i; big = dum; }
        }
        if (j != imax) {
            for (k = 0; k < n; ++k) { double t = a[imax + n * k]; a[imax + n * k] = a[j + n * k]; a[j + n * k] = t; }
            vv[imax] = vv[j];
        }
        indx[j] = imax;
        if (j != n - 1) {
            double dum;
            if (a[j + n * j] == 0.0) a[j + n * j] = (double)1.0e-20f;
            dum = 1.0 / a[j + n * j];
            for (i = j + 1; i < n; ++i) a[i + n * j] = a[i + n * j] * dum;
        }
    }
    if (a[(n - 1) + n * (n - 1)] == 0.0) a[(n - 1) + n * (n - 1)] = (double)1.0e-20f;
    return 0;
}

static void lu_solve(const double *a, int n, const int *indx, double *b)
{
    int i, j, ii = -1;
    for (i = 0; i < n; ++i) {
        int ll = indx[i];
        double sum = b[ll];
        b[ll] = b[i];
        if (ii >= 0) for (j = ii; j < i; ++j) sum = sum - a[i + n * j] * b[j];
        else if (sum != 0.0) ii = i;
        b[i] = sum;
    }
    for (i = n - 1; i >= 0; --i) {
        double sum = b[i];
        for (j = i + 1; j < n; ++j) sum = sum - a[i + n * j] * b[j];
        b[i] = sum / a[i + n * i];
    }
}

/* ------------------------------------------------------------------ implicit.f90:36-165 */
ORC_API int orc_tail_init(orc_ctx *c, double dt)
{
    const int mx = c->mx, nx = c->nx, kx = c->kx, nl = mx + nx + 1;
    const double akap = akap_(), rgas = akap * CP;
    const double rgam = rgas * GAMMA_LAPSE / (1000.0 * GRAV);
    double *xa = dalloc((size_t)kx * kx), *xb = dalloc((size_t)kx * kx), *ya = dalloc((size_t)kx * kx);
    double *xe = dalloc((size_t)kx * kx), *xf = dalloc((size_t)kx * kx), *dsum = dalloc((size_t)kx);
    int *indx = (int *)calloc((size_t)kx, sizeof(int));
    double xi, xxi;
    int m, n, k, k1, k2, l, rc = 0;
#define M2(a, r, cc_) a[(r) + kx * (cc_)]
    if (!c->sigma_ready) return -2;                    /* geometry.f90:42-48 defines no sigma set for this kx (orc_set_sigma) */
    init_hdiff(c);
    for (m = 0; m < mx * nx; ++m) {
        c->dmp1[m] = 1.0 / (1.0 + c->dmp[m] * dt);
        c->dmp1d[m] = 1.0 / (1.0 + c->dmpd[m] * dt);
        c->dmp1s[m] = 1.0 / (1.0 + c->dmps[m] * dt);
    }
    for (k = 0; k < kx; ++k) {
        double fl = c->fsg[k], lo = (double)0.2f;               /* max(0.2,fsg): float32 0.2 [f32] */
        c->tref[k] = 288.0 * pow(fl > lo ? fl : lo, rgam);
        c->tref1[k] = rgas * c->tref[k];
        c->tref2[k] = akap * c->tref[k];
        c->tref3[k] = c->fsgr[k] * c->tref[k];
    }
    xi = dt * ALPH;
    xxi = xi / (REARTH * REARTH);
    for (k = 0; k < kx; ++k) c->dhsx[k] = xi * c->dhs[k];
    for (n = 0; n < nx; ++n)
        for (m = 0; m < mx; ++m)
            c->elz[m + mx * n] = (double)((float)(m + n) * (float)(m + n + 1)) * xxi;
    for (k = 0; k < kx; ++k)
        for (k1 = 0; k1 < kx; ++k1) M2(ya, k, k1) = -akap * c->tref[k] * c->dhs[k1];
    for (k = 1; k < kx; ++k)
        M2(xa, k, k - 1) = 0.5 * (akap * c->tref[k] / c->fsg[k] - (c->tref[k] - c->tref[k - 1]) / c->dhs[k]);
    for (k = 0; k < kx - 1; ++k)
        M2(xa, k, k) = 0.5 * (akap * c->tref[k] / c->fsg[k] - (c->tref[k + 1] - c->tref[k]) / c->dhs[k]);
    dsum[0] = c->dhs[0];
    for (k = 1; k < kx; ++k) dsum[k] = dsum[k - 1] + c->dhs[k];
    for (k = 0; k < kx - 1; ++k)
        for (k1 = 0; k1 < kx; ++k1) {
            M2(xb, k, k1) = c->dhs[k1] * dsum[k];
            if (k1 <= k) M2(xb, k, k1) = M2(xb, k, k1) - c->dhs[k1];
        }
    for (k = 0; k < kx; ++k)
        for (k1 = 0; k1 < kx; ++k1) {
            M2(c->xc, k, k1) = M2(ya, k, k1);
            for (k2 = 0; k2 < kx - 1; ++k2) M2(c->xc, k, k1) = M2(c->xc, k, k1) + M2(xa, k, k2) * M2(xb, k2, k1);
        }
    for (k = 0; k < kx * kx; ++k) c->xd[k] = 0.0;
    for (k = 0; k < kx; ++k)
        for (k1 = k + 1; k1 < kx; ++k1) M2(c->xd, k, k1) = rgas * log(c->hsg[k1 + 1] / c->hsg[k1]);
    for (k = 0; k < kx; ++k) M2(c->xd, k, k) = rgas * log(c->hsg[k + 1] / c->fsg[k]);
    for (k = 0; k < kx; ++k)
        for (k1 = 0; k1 < kx; ++k1) {
            M2(xe, k, k1) = 0.0;
            for (k2 = 0; k2 < kx; ++k2) M2(xe, k, k1) = M2(xe, k, k1) + M2(c->xd, k, k2) * M2(c->xc, k2, k1);
        }
    for (l = 1; l <= nl; ++l) {
        double xxx = (double)((float)l * (float)(l + 1)) / (REARTH * REARTH);
        double *xjl = c->xj + (size_t)kx * kx * (l - 1);
        for (k = 0; k < kx; ++k)
            for (k1 = 0; k1 < kx; ++k1)
                M2(xf, k, k1) = xi * xi * xxx * (rgas * c->tref[k] * c->dhs[k1] - M2(xe, k, k1));
        for (k = 0; k < kx; ++k) M2(xf, k, k) = M2(xf, k, k) + 1.0;
        for (k = 0; k < kx * kx; ++k) xjl[k] = 0.0;
        for (k = 0; k < kx; ++k) M2(xjl, k, k) = 1.0;
        if (lu_decompose(xf, kx, indx) != 0) { rc = -1; break; }
        for (k = 0; k < kx; ++k) lu_solve(xf, kx, indx, xjl + (size_t)kx * k);
    }
    for (k = 0; k < kx * kx; ++k) c->xc[k] = c->xc[k] * xi;
    c->tail_ready = (rc == 0);
    free(xa); free(xb); free(ya); free(xe); free(xf); free(dsum); free(indx);
    return rc;
}

/* ------------------------------------------------------------------ implicit.f90:168-217 */
ORC_API void orc_implicit_terms(const orc_ctx *c, double *divdt, double *tdt, double *psdt)
{
    const int mx = c->mx, nx = c->nx, kx = c->kx, t = mx * nx;
    double *ye = dalloc((size_t)2 * t * kx), *yf = dalloc((size_t)2 * t * kx);
    int k, k1, m, n, i;
    for (k1 = 0; k1 < kx; ++k1)
        for (k = 0; k < kx; ++k)
            for (i = 0; i < t; ++i)
                cset(ye, k * t + i, cadd(cget(ye, k * t + i), rmul(M2(c->xd, k, k1), cget(tdt, k1 * t + i))));
    for (k = 0; k < kx; ++k)
        for (i = 0; i < t; ++i)
            cset(ye, k * t + i, cadd(cget(ye, k * t + i), rmul(c->tref1[k], cget(psdt, i))));
    for (k = 0; k < kx; ++k)
        for (i = 0; i < t; ++i)
            cset(yf, k * t + i, cadd(cget(divdt, k * t + i), rmul(c->elz[i], cget(ye, k * t + i))));
    for (i = 0; i < 2 * t * kx; ++i) divdt[i] = 0.0;
    for (n = 0; n < nx; ++n)
        for (m = 0; m < mx; ++m) {
            int l = m + n;                              /* = (m1+n1-2) in 1-based terms */
            const double *xjl;
            if (l == 0) continue;
            xjl = c->xj + (size_t)kx * kx * (l - 1);
            for (k1 = 0; k1 < kx; ++k1)
                for (k = 0; k < kx; ++k)
                    cset(divdt, k * t + SI(m, n),
                         cadd(cget(divdt, k * t + SI(m, n)), rmul(M2(xjl, k, k1), cget(yf, k1 * t + SI(m, n)))));
        }
    for (k = 0; k < kx; ++k)
        for (i = 0; i < t; ++i) cset(psdt, i, csub(cget(psdt, i), rmul(c->dhsx[k], cget(divdt, k * t + i))));
    for (k = 0; k < kx; ++k)
        for (k1 = 0; k1 < kx; ++k1)
            for (i = 0; i < t; ++i)
                cset(tdt, k * t + i, cadd(cget(tdt, k * t + i), rmul(M2(c->xc, k, k1), cget(divdt, k1 * t + i))));
    free(ye); free(yf);
}

/* ------------------------------------------------------------------ geopotential.f90:33-57 */
ORC_API void orc_geopotential(const orc_ctx *c, const double *t, const double *phis, double *phi)
{
    const int mx = c->mx, nx = c->nx, kx = c->kx, sz = mx * nx;
    int k, i, n;
    for (i = 0; i < sz; ++i)                                             /* 1. bottom layer (:45) */
        cset(phi, (kx - 1) * sz + i, cadd(cget(phis, i), rmul(c->xgeop1[kx - 1], cget(t, (kx - 1) * sz + i))));
    for (k = kx - 2; k >= 0; --k)                                        /* 2. other layers (:48-51) */
        for (i = 0; i < sz; ++i)
            cset(phi, k * sz + i, cadd(cadd(cget(phi, (k + 1) * sz + i), rmul(c->xgeop2[k + 1], cget(t, (k + 1) * sz + i))),
                                       rmul(c->xgeop1[k], cget(t, k * sz + i))));
    for (k = 1; k <= kx - 2; ++k) {                                      /* 3. lapse-rate correction, m = 1 only (:54-57) */
        double corf = c->xgeop1[k] * 0.5 * log(c->hsg[k + 1] / c->fsg[k]) / log(c->fsg[k + 1] / c->fsg[k - 1]);
        for (n = 0; n < nx; ++n)
            cset(phi, k * sz + SI(0, n), cadd(cget(phi, k * sz + SI(0, n)),
                                              rmul(corf, csub(cget(t, (k + 1) * sz + SI(0, n)), cget(t, (k - 1) * sz + SI(0, n))))));
    }
}

/* ------------------------------------------------------------------ tendencies.f90:242-293 get_spectral_tendencies
 * PINNED since round 3: tendencies.f90 as a whole cannot be compiled here (get_grid_point_tendencies uses physics -> ... ->
 * netcdf), but this subroutine, cut out of the reference file as it is together with the declaration part of
 * prognostics.f90 (oracle/build_ref.sh), compiles with flang; tests/golden/ref_spectend.npz holds its outputs at 8, 5 and
 * 16 levels (tests/test_oracle_golden.py::test_spectral_tendencies_pinned).  div, t, ps: time level j2 of the prognostics. */
ORC_API void orc_spectral_tendencies(const orc_ctx *c, const double *div, const double *t, const double *ps, const double *phis,
                                     double *divdt, double *tdt, double *psdt, double *phi)
{
    const int mx = c->mx, nx = c->nx, kx = c->kx, sz = mx * nx;
    const double rgas = akap_() * CP;
    double *dmeanc = dalloc((size_t)2 * sz), *sigdtc = dalloc((size_t)2 * sz * (kx + 1)), *dumk = dalloc((size_t)2 * sz * (kx + 1));
    int k, i;
    for (k = 0; k < kx; ++k)                                             /* :257-260 */
        for (i = 0; i < sz; ++i) cset(dmeanc, i, cadd(cget(dmeanc, i), rmul(c->dhs[k], cget(div, k * sz + i))));
    for (i = 0; i < sz; ++i) cset(psdt, i, csub(cget(psdt, i), cget(dmeanc, i)));   /* :262 */
    psdt[0] = 0.0; psdt[1] = 0.0;                                        /* :263 */
    /* sigdtc(:,:,1) = sigdtc(:,:,kx+1) = 0 (calloc) ; :269-271 */
    for (k = 0; k < kx - 1; ++k)
        for (i = 0; i < sz; ++i)
            cset(sigdtc, (k + 1) * sz + i, csub(cget(sigdtc, k * sz + i), rmul(c->dhs[k], csub(cget(div, k * sz + i), cget(dmeanc, i)))));
    for (k = 1; k < kx; ++k)                                             /* :276-278 */
        for (i = 0; i < sz; ++i) cset(dumk, k * sz + i, rmul(c->tref[k] - c->tref[k - 1], cget(sigdtc, k * sz + i)));
    for (k = 0; k < kx; ++k)                                             /* :280-284 */
        for (i = 0; i < sz; ++i) {
            cplx a = csub(cget(tdt, k * sz + i), rmul(c->dhsr[k], cadd(cget(dumk, (k + 1) * sz + i), cget(dumk, k * sz + i))));
            cplx b = cadd(a, rmul(c->tref3[k], cadd(cget(sigdtc, (k + 1) * sz + i), cget(sigdtc, k * sz + i))));
            cset(tdt, k * sz + i, csub(b, rmul(c->tref2[k], cget(dmeanc, i))));
        }
    orc_geopotential(c, t, phis, phi);                                   /* :287 */
    for (k = 0; k < kx; ++k)                                             /* :289-291: divdt - laplacian(phi + rgas*tref(k)*ps) */
        for (i = 0; i < sz; ++i) {
            cplx x = cadd(cget(phi, k * sz + i), rmul(rgas * c->tref[k], cget(ps, i)));
            cset(divdt, k * sz + i, csub(cget(divdt, k * sz + i), rmul(c->el2[i], cneg(x))));
        }
    free(dmeanc); free(sigdtc); free(dumk);
}

/* ------------------------------------------------------------------ time_stepping.f90:62-96: the diffusion block of step()
 * PINNED since round 3 as part of the whole step: the reference's time_stepping.f90, compiled unchanged on tendencies.f90 minus
 * its three physics lines (oracle/build_ref.sh), reproduces the sequence this routine is part of bit for bit
 * (tests/dynstep.py, tests/golden/ref_dynstep.npz, test_dynamics_step_pinned).  vor, div, t, tr: time level 1; tcorh, qcorh: (mx,nx) complex; tr/trdt may be NULL.                                */
ORC_API void orc_hdiff_step(const orc_ctx *c, const double *vor, const double *div, const double *t, const double *tr,
                            const double *tcorh, const double *qcorh, double sdrag,
                            double *vordt, double *divdt, double *tdt, double *trdt)
{
    const int mx = c->mx, nx = c->nx, kx = c->kx, sz = mx * nx;
    double *ctmp = dalloc((size_t)2 * sz * kx), *tmp = dalloc((size_t)2 * sz * kx);
    int k, i, n;
    orc_hdiff(c, kx, vor, vordt, c->dmp, c->dmp1, tmp);  memcpy(vordt, tmp, sizeof(double) * 2 * sz * kx);    /* :63 */
    orc_hdiff(c, kx, div, divdt, c->dmpd, c->dmp1d, tmp); memcpy(divdt, tmp, sizeof(double) * 2 * sz * kx);   /* :64 */
    for (k = 0; k < kx; ++k)                                                                                 /* :66-72 */
        for (i = 0; i < sz; ++i) cset(ctmp, k * sz + i, cadd(cget(t, k * sz + i), rmul(c->tcorv[k], cget(tcorh, i))));
    orc_hdiff(c, kx, ctmp, tdt, c->dmp, c->dmp1, tmp);   memcpy(tdt, tmp, sizeof(double) * 2 * sz * kx);      /* :74 */
    for (n = 0; n < nx; ++n) {                                                                               /* :77-81 */
        cset(vordt, SI(0, n), csub(cget(vordt, SI(0, n)), rmul(sdrag, cget(vor, SI(0, n)))));
        cset(divdt, SI(0, n), csub(cget(divdt, SI(0, n)), rmul(sdrag, cget(div, SI(0, n)))));
    }
    orc_hdiff(c, kx, vor, vordt, c->dmps, c->dmp1s, tmp); memcpy(vordt, tmp, sizeof(double) * 2 * sz * kx);  /* :83-85 */
    orc_hdiff(c, kx, div, divdt, c->dmps, c->dmp1s, tmp); memcpy(divdt, tmp, sizeof(double) * 2 * sz * kx);
    orc_hdiff(c, kx, ctmp, tdt, c->dmps, c->dmp1s, tmp);  memcpy(tdt, tmp, sizeof(double) * 2 * sz * kx);
    if (tr) {
        for (k = 0; k < kx; ++k)                                                                             /* :88-94 */
            for (i = 0; i < sz; ++i) cset(ctmp, k * sz + i, cadd(cget(tr, k * sz + i), rmul(c->qcorv[k], cget(qcorh, i))));
        orc_hdiff(c, kx, ctmp, trdt, c->dmpd, c->dmp1d, tmp); memcpy(trdt, tmp, sizeof(double) * 2 * sz * kx);   /* :96 */
    }
    free(ctmp); free(tmp);
}

/* ------------------------------------------------------------------ time_stepping.f90:121-167 step_field_2d / _3d
 * PINNED since round 3 (the two functions, cut out of time_stepping.f90 as they are, compile with flang: tests/golden/
 * ref_step.npz, test_step_field_pinned).  field(mx,nx,nlev,2): both time levels; fdt(mx,nx,nlev) truncated in place (:155-157). */
ORC_API void orc_step_field(const orc_ctx *c, int nlev, int j1, double dt, double eps, double wil, double *field, double *fdt)
{
    const int sz = c->mx * c->nx;
    double *l1 = field, *l2 = field + (size_t)2 * sz * nlev, *lj = (j1 == 1) ? l1 : l2;
    double *fnew = dalloc((size_t)2 * sz);
    int k, i;
    for (k = 0; k < nlev; ++k) {
        if (c->ix == c->iy * 4) orc_trunct(c, fdt + (size_t)2 * sz * k);
        for (i = 0; i < sz; ++i) cset(fnew, i, cadd(cget(l1, k * sz + i), rmul(dt, cget(fdt, k * sz + i))));       /* :160 */
        for (i = 0; i < sz; ++i)                                                                                   /* :161 */
            cset(l1, k * sz + i, cadd(cget(lj, k * sz + i),
                                      rmul(wil * eps, cadd(csub(cget(l1, k * sz + i), rmul(2.0, cget(lj, k * sz + i))), cget(fnew, i)))));
        for (i = 0; i < sz; ++i)                                                                                   /* :164 */
            cset(l2, k * sz + i, csub(cget(fnew, i),
                                      rmul((1.0 - wil) * eps, cadd(csub(cget(l1, k * sz + i), rmul(2.0, cget(lj, k * sz + i))), cget(fnew, i)))));
    }
    free(fnew);
}

/* ------------------------------------------------------------------ input_output.f90:184-206: the gridded snapshot
 * PINNED since the end of round 3: the computing lines of subroutine output (:183-205, with its own declarations) are cut out of
 * the reference file as they are and compiled by flang (oracle/build_ref.sh: module output_fields_ref); this function equals
 * them bit for bit in float32 at 8, 5, 7 and 16 levels (tests/golden/ref_output.npz, test_output_fields_pinned/_live).
 * Inputs: time level 1 of vor, div, t, q = tr(:,:,:,1,1), and phi, all (mx,nx,kx); ps (mx,nx).
 * Outputs: float32 arrays u, v, t, q, phi (ix,il,kx) and ps (ix,il).                                              */
ORC_API void orc_output(const orc_ctx *c, const double *vor, const double *div, const double *t, const double *q,
                        const double *phi, const double *ps, float *u_out, float *v_out, float *t_out, float *q_out,
                        float *phi_out, float *ps_out)
{
    const int kx = c->kx, ns = 2 * c->mx * c->nx, ng = c->ix * c->il;
    double *ucos = dalloc((size_t)ns), *vcos = dalloc((size_t)ns), *g = dalloc((size_t)ng);
    int k, i;
    for (k = 0; k < kx; ++k) {
        orc_uvspec(c, vor + (size_t)ns * k, div + (size_t)ns * k, ucos, vcos);                     /* :185 */
        orc_spec_to_grid(c, ucos, 2, g); for (i = 0; i < ng; ++i) u_out[(size_t)ng * k + i] = (float)g[i];          /* :186, :201 */
        orc_spec_to_grid(c, vcos, 2, g); for (i = 0; i < ng; ++i) v_out[(size_t)ng * k + i] = (float)g[i];          /* :187, :202 */
        orc_spec_to_grid(c, t + (size_t)ns * k, 1, g); for (i = 0; i < ng; ++i) t_out[(size_t)ng * k + i] = (float)g[i];   /* :188, :203 */
        orc_spec_to_grid(c, q + (size_t)ns * k, 1, g);                                              /* :189, :204 q*1.0e-3 [f32 literal] */
        for (i = 0; i < ng; ++i) q_out[(size_t)ng * k + i] = (float)(g[i] * (double)1.0e-3f);
        orc_spec_to_grid(c, phi + (size_t)ns * k, 1, g);                                            /* :190, :205 phi/grav */
        for (i = 0; i < ng; ++i) phi_out[(size_t)ng * k + i] = (float)(g[i] / GRAV);
    }
    orc_spec_to_grid(c, ps, 1, g);                                                                  /* :192, :206 p0*exp(ps) */
    for (i = 0; i < ng; ++i) ps_out[i] = (float)(P0 * exp(g[i]));
    free(ucos); free(vcos); free(g);
}

/* ------------------------------------------------------------------ tendencies.f90:105-197: grid-space dynamical tendencies
 * PINNED since round 3 as part of the whole step and of get_tendencies (tendencies.f90 minus its three physics lines compiles
 * with flang: oracle/build_ref.sh; test_dynamics_step_pinned, bit for bit).  Inputs: gridded prognostics of time level j2 (vorg WITHOUT
 * Coriolis: added here as :103-107 does), px, py = spec_to_grid(grad(ps), 2).  Outputs in the layout of the direct batch:
 * u, v [3 kx] = (utend, vtend) | (-ug*tgg, -vg*tgg) | (-ug*trg, -vg*trg); plain [3 kx + 1] = KE | ttend | trtend | psdt grid. */
ORC_API void orc_grid_tendencies(const orc_ctx *c, const double *ug, const double *vg, const double *tg, const double *vorg_in,
                                 const double *divg, const double *trg, const double *px, const double *py,
                                 double *u, double *v, double *plain)
{
    const int kx = c->kx, g = c->ix * c->il, ix = c->ix;
    const double rgas = akap_() * CP, akap = akap_();
    double *vorg = dalloc((size_t)g * kx), *umean = dalloc(g), *vmean = dalloc(g), *dmean = dalloc(g);
    double *puv = dalloc((size_t)g * kx), *tgg = dalloc((size_t)g * kx);
    double *sigdt = dalloc((size_t)g * (kx + 1)), *sigm = dalloc((size_t)g * (kx + 1)), *temp = dalloc((size_t)g * (kx + 1));
    double *utend = u, *vtend = v, *ttend = plain + (size_t)g * kx, *trtend = plain + (size_t)2 * g * kx;
    int k, i;
#define A3(a, k_) ((a) + (size_t)g * (k_))
    for (k = 0; k < kx; ++k)
        for (i = 0; i < g; ++i) A3(vorg, k)[i] = A3(vorg_in, k)[i] + c->coriol[i / ix];                 /* :103-107 */
    for (k = 0; k < kx; ++k)                                                                             /* :113-117 */
        for (i = 0; i < g; ++i) {
            umean[i] = umean[i] + A3(ug, k)[i] * c->dhs[k];
            vmean[i] = vmean[i] + A3(vg, k)[i] * c->dhs[k];
            dmean[i] = dmean[i] + A3(divg, k)[i] * c->dhs[k];
        }
    for (i = 0; i < g; ++i) A3(plain, 3 * kx)[i] = -umean[i] * px[i] - vmean[i] * py[i];                /* :125 */
    for (k = 0; k < kx; ++k)                                                                             /* :135-137 */
        for (i = 0; i < g; ++i) A3(puv, k)[i] = (A3(ug, k)[i] - umean[i]) * px[i] + (A3(vg, k)[i] - vmean[i]) * py[i];
    for (k = 0; k < kx; ++k)                                                                             /* :139-142 */
        for (i = 0; i < g; ++i) {
            A3(sigdt, k + 1)[i] = A3(sigdt, k)[i] - c->dhs[k] * (A3(puv, k)[i] + A3(divg, k)[i] - dmean[i]);
            A3(sigm, k + 1)[i] = A3(sigm, k)[i] - c->dhs[k] * A3(puv, k)[i];
        }
    for (k = 0; k < kx; ++k)                                                                             /* :146-148 */
        for (i = 0; i < g; ++i) A3(tgg, k)[i] = A3(tg, k)[i] - c->tref[k];
    for (k = 1; k < kx; ++k)                                                                             /* :155-157 (temp(1) = temp(kx+1) = 0) */
        for (i = 0; i < g; ++i) A3(temp, k)[i] = A3(sigdt, k)[i] * (A3(ug, k)[i] - A3(ug, k - 1)[i]);
    for (k = 0; k < kx; ++k)                                                                             /* :159-162 */
        for (i = 0; i < g; ++i)
            A3(utend, k)[i] = A3(vg, k)[i] * A3(vorg, k)[i] - A3(tgg, k)[i] * rgas * px[i] - (A3(temp, k + 1)[i] + A3(temp, k)[i]) * c->dhsr[k];
    for (k = 1; k < kx; ++k)                                                                             /* :165-167 */
        for (i = 0; i < g; ++i) A3(temp, k)[i] = A3(sigdt, k)[i] * (A3(vg, k)[i] - A3(vg, k - 1)[i]);
    for (k = 0; k < kx; ++k)                                                                             /* :169-172 */
        for (i = 0; i < g; ++i)
            A3(vtend, k)[i] = -A3(ug, k)[i] * A3(vorg, k)[i] - A3(tgg, k)[i] * rgas * py[i] - (A3(temp, k + 1)[i] + A3(temp, k)[i]) * c->dhsr[k];
    for (k = 1; k < kx; ++k)                                                                             /* :175-178 */
        for (i = 0; i < g; ++i)
            A3(temp, k)[i] = A3(sigdt, k)[i] * (A3(tgg, k)[i] - A3(tgg, k - 1)[i]) + A3(sigm, k)[i] * (c->tref[k] - c->tref[k - 1]);
    for (k = 0; k < kx; ++k)                                                                             /* :180-184 */
        for (i = 0; i < g; ++i)
            A3(ttend, k)[i] = A3(tgg, k)[i] * A3(divg, k)[i] - (A3(temp, k + 1)[i] + A3(temp, k)[i]) * c->dhsr[k]
                              + c->fsgr[k] * A3(tgg, k)[i] * (A3(sigdt, k + 1)[i] + A3(sigdt, k)[i])
                              + c->tref3[k] * (A3(sigm, k + 1)[i] + A3(sigm, k)[i])
                              + akap * (A3(tg, k)[i] * A3(puv, k)[i] - A3(tgg, k)[i] * dmean[i]);
    for (k = 1; k < kx; ++k)                                                                             /* :187-189 */
        for (i = 0; i < g; ++i) A3(temp, k)[i] = A3(sigdt, k)[i] * (A3(trg, k)[i] - A3(trg, k - 1)[i]);
    for (k = 1; k <= 2 && k <= kx; ++k)                                                                  /* :191 temp(:,:,2:3) = 0 */
        for (i = 0; i < g; ++i) A3(temp, k)[i] = 0.0;
    for (k = 0; k < kx; ++k)                                                                             /* :193-195 */
        for (i = 0; i < g; ++i)
            A3(trtend, k)[i] = A3(trg, k)[i] * A3(divg, k)[i] - (A3(temp, k + 1)[i] + A3(temp, k)[i]) * c->dhsr[k];
    for (k = 0; k < kx; ++k)                                                                             /* operands of :216-232 */
        for (i = 0; i < g; ++i) {
            A3(plain, k)[i] = 0.5 * (A3(ug, k)[i] * A3(ug, k)[i] + A3(vg, k)[i] * A3(vg, k)[i]);
            A3(u, kx + k)[i] = -A3(ug, k)[i] * A3(tgg, k)[i];      A3(v, kx + k)[i] = -A3(vg, k)[i] * A3(tgg, k)[i];
            A3(u, 2 * kx + k)[i] = -A3(ug, k)[i] * A3(trg, k)[i];  A3(v, 2 * kx + k)[i] = -A3(vg, k)[i] * A3(trg, k)[i];
        }
#undef A3
    free(vorg); free(umean); free(vmean); free(dmean); free(puv); free(tgg); free(sigdt); free(sigm); free(temp);
}

/* tendencies.f90:125-126, 218-233 in spectral space, on the outputs of the direct batch (see spdy_tendency_combine_dev) */
ORC_API void orc_tendency_combine(const orc_ctx *c, double *pdiv, double *pspec)
{
    const int kx = c->kx, sz = c->mx * c->nx;
    int k, i;
    for (k = 0; k < kx; ++k)
        for (i = 0; i < sz; ++i) {
            cset(pdiv, k * sz + i, csub(cget(pdiv, k * sz + i), rmul(c->el2[i], cneg(cget(pspec, k * sz + i)))));
            cset(pdiv, (kx + k) * sz + i, cadd(cget(pdiv, (kx + k) * sz + i), cget(pspec, (kx + k) * sz + i)));
            cset(pdiv, (2 * kx + k) * sz + i, cadd(cget(pdiv, (2 * kx + k) * sz + i), cget(pspec, (2 * kx + k) * sz + i)));
        }
    pspec[2 * (size_t)3 * kx * sz] = 0.0; pspec[2 * (size_t)3 * kx * sz + 1] = 0.0;
}

/* ------------------------------------------------------------------ context */
ORC_API orc_ctx *orc_create(int trunc, int ix, int iy, int kx)
{
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    int mx, nx, t;
    if (!c) return NULL;
    c->trunc = trunc; c->ix = ix; c->iy = iy; c->il = 2 * iy; c->kx = kx;
    c->nx = nx = trunc + 2; c->mx = mx = trunc + 1; t = mx * nx;
    if (2 * mx > ix) { free(c); return NULL; }
    c->sia_half = dalloc(iy); c->coa_half = dalloc(c->il); c->cosgr = dalloc(c->il); c->cosgr2 = dalloc(c->il);
    c->hsg = dalloc(kx + 1); c->dhs = dalloc(kx); c->fsg = dalloc(kx); c->dhsr = dalloc(kx); c->fsgr = dalloc(kx);
    c->work = dalloc(ix);
    c->epsi = dalloc((size_t)(mx + 1) * (nx + 1)); c->repsi = dalloc((size_t)(mx + 1) * (nx + 1));
    c->wt = dalloc(iy); c->poly = dalloc((size_t)t * iy); c->nsh2 = (int *)calloc(nx, sizeof(int));
    c->el2 = dalloc(t); c->elm2 = dalloc(t); c->el4 = dalloc(t); c->trfilt = dalloc(t);
    c->gradx = dalloc(mx); c->gradym = dalloc(t); c->gradyp = dalloc(t);
    c->uvdx = dalloc(t); c->uvdym = dalloc(t); c->uvdyp = dalloc(t); c->vddym = dalloc(t); c->vddyp = dalloc(t);
    c->dmp = dalloc(t); c->dmpd = dalloc(t); c->dmps = dalloc(t);
    c->dmp1 = dalloc(t); c->dmp1d = dalloc(t); c->dmp1s = dalloc(t);
    c->tref = dalloc(kx); c->tref1 = dalloc(kx); c->tref2 = dalloc(kx); c->tref3 = dalloc(kx);
    c->xc = dalloc((size_t)kx * kx); c->xd = dalloc((size_t)kx * kx);
    c->xj = dalloc((size_t)kx * kx * (mx + nx + 1)); c->dhsx = dalloc(kx); c->elz = dalloc(t);
    c->coriol = dalloc(c->il); c->xgeop1 = dalloc(kx); c->xgeop2 = dalloc(kx); c->tcorv = dalloc(kx); c->qcorv = dalloc(kx);
    init_geometry(c);
    orc_rffti1(ix, c->work, c->ifac);
    init_legendre(c);
    init_spectral(c);
    init_hdiff(c);
    return c;
}

ORC_API void orc_destroy(orc_ctx *c)
{
    if (!c) return;
    free(c->sia_half); free(c->coa_half); free(c->cosgr); free(c->cosgr2);
    free(c->hsg); free(c->dhs); free(c->fsg); free(c->dhsr); free(c->fsgr); free(c->work);
    free(c->epsi); free(c->repsi); free(c->wt); free(c->poly); free(c->nsh2);
    free(c->el2); free(c->elm2); free(c->el4); free(c->trfilt); free(c->gradx); free(c->gradym); free(c->gradyp);
    free(c->uvdx); free(c->uvdym); free(c->uvdyp); free(c->vddym); free(c->vddyp);
    free(c->dmp); free(c->dmpd); free(c->dmps); free(c->dmp1); free(c->dmp1d); free(c->dmp1s);
    free(c->tref); free(c->tref1); free(c->tref2); free(c->tref3);
    free(c->xc); free(c->xd); free(c->xj); free(c->dhsx); free(c->elz);
    free(c->coriol); free(c->xgeop1); free(c->xgeop2); free(c->tcorv); free(c->qcorv);
    free(c);
}

/* Named table access for tests.  Returns the element count, or -1 for an unknown name.
 * "ifac" is returned as doubles.                                                          */
ORC_API int orc_get_table(const orc_ctx *c, const char *name, double *out)
{
    const int mx = c->mx, nx = c->nx, t = mx * nx, kx = c->kx;
    const double *src = NULL; int cnt = 0, i;
#define TBL(nm, ptr, count) if (!strcmp(name, nm)) { src = (ptr); cnt = (count); }
    TBL("sia_half", c->sia_half, c->iy) TBL("coa_half", c->coa_half, c->il)
    TBL("cosgr", c->cosgr, c->il) TBL("cosgr2", c->cosgr2, c->il)
    TBL("hsg", c->hsg, kx + 1) TBL("dhs", c->dhs, kx) TBL("fsg", c->fsg, kx)
    TBL("dhsr", c->dhsr, kx) TBL("fsgr", c->fsgr, kx)
    TBL("work", c->work, c->ix) TBL("epsi", c->epsi, (mx + 1) * (nx + 1)) TBL("wt", c->wt, c->iy)
    TBL("poly", c->poly, t * c->iy)
    TBL("el2", c->el2, t) TBL("elm2", c->elm2, t) TBL("el4", c->el4, t) TBL("trfilt", c->trfilt, t)
    TBL("gradx", c->gradx, mx) TBL("gradym", c->gradym, t) TBL("gradyp", c->gradyp, t)
    TBL("uvdx", c->uvdx, t) TBL("uvdym", c->uvdym, t) TBL("uvdyp", c->uvdyp, t)
    TBL("vddym", c->vddym, t) TBL("vddyp", c->vddyp, t)
    TBL("dmp", c->dmp, t) TBL("dmpd", c->dmpd, t) TBL("dmps", c->dmps, t)
    TBL("dmp1", c->dmp1, t) TBL("dmp1d", c->dmp1d, t) TBL("dmp1s", c->dmp1s, t)
    TBL("tref", c->tref, kx) TBL("tref1", c->tref1, kx) TBL("tref2", c->tref2, kx) TBL("tref3", c->tref3, kx)
    TBL("xc", c->xc, kx * kx) TBL("xd", c->xd, kx * kx) TBL("xj", c->xj, kx * kx * (mx + nx + 1))
    TBL("dhsx", c->dhsx, kx) TBL("elz", c->elz, t)
    TBL("coriol", c->coriol, c->il) TBL("xgeop1", c->xgeop1, kx) TBL("xgeop2", c->xgeop2, kx)
    TBL("tcorv", c->tcorv, kx) TBL("qcorv", c->qcorv, kx)
    if (!strcmp(name, "ifac")) { for (i = 0; i < 15; ++i) out[i] = (double)c->ifac[i]; return 15; }
    if (!strcmp(name, "nsh2")) { for (i = 0; i < nx; ++i) out[i] = (double)c->nsh2[i]; return nx; }
    if (!src) return -1;
    if (out) memcpy(out, src, sizeof(double) * (size_t)cnt);
    return cnt;
}

/* CPU baseline loop (bench.py cpu_baseline.kind = "port"): same execution model as the
 * reference -- one field at a time, grid_to_spec then spec_to_grid(.,1).                    */
ORC_API void orc_roundtrip_loop(const orc_ctx *c, int nf, int nrep, const double *g_in, double *g_out)
{
    const size_t gsz = (size_t)c->ix * c->il;
    double *s = dalloc((size_t)2 * c->mx * c->nx);
    int r, f;
    for (r = 0; r < nrep; ++r)
        for (f = 0; f < nf; ++f) {
            orc_grid_to_spec(c, g_in + gsz * f, s);
            orc_spec_to_grid(c, s, 1, g_out + gsz * f);
        }
    free(s);
}
