// gfx950 (MI355X, CDNA4) kernels for the speedy.f90 grid<->spectral path.
//
//   legendre_inv / legendre_dir : per-zonal-wavenumber contractions on the FP64 matrix cores
//                                 (v_mfma_f64_16x16x4_f64); batch (field x re/im) is the N dimension.
//   fourier_inv / fourier_dir   : FFTPACK-schedule real FFTs (N = 96: 2*4*4*3, N = 192: 4*4*4*3).
//                                 Each row is split over N/48 lanes; a lane runs a 48-point
//                                 radix-(4,4,3) sub-transform entirely in registers, the ido=48
//                                 stage is done through LDS, and LDS is also the transpose
//                                 buffer that makes every HBM access a full coalesced line.
//   spectral operators, horizontal diffusion, semi-implicit solve: one lane per coefficient.
//
// Numerics: same butterflies, same twiddle values, same float32-derived constants as the
// reference (tables come from spdy_tables.cpp); only FMA contraction and the order of the
// Legendre sums differ -> agreement ~1e-15 relative, bar 1e-12.
#include "spdy_kernels.hpp"
#include "spdy_cpx.hpp"
#include "spdy_t63_sched.hpp"

namespace spdy {

typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// FFT constants (uniform -> scalar loads).  96 and 192 share the ido=12 / ido=3 stage twiddles
// mathematically, but each resolution keeps its own copy so plans of both can coexist.
// ------------------------------------------------------------------------------------------
#ifdef SPDY_PAD
__device__ __attribute__((used)) char g_layout_pad[SPDY_PAD];   // code-object layout experiments
#endif
__constant__ FftConstants c_fft96;
__constant__ FftConstants c_fft192;

hipError_t upload_fft_constants(int ix, const FftConstants &c)
{
    if (ix == 96) return hipMemcpyToSymbol(HIP_SYMBOL(c_fft96), &c, sizeof(c));
    if (ix == 192) return hipMemcpyToSymbol(HIP_SYMBOL(c_fft192), &c, sizeof(c));
    return hipErrorInvalidValue;
}

template <int NF> __device__ __forceinline__ const FftConstants &fftc();
template <> __device__ __forceinline__ const FftConstants &fftc<96>() { return c_fft96; }
template <> __device__ __forceinline__ const FftConstants &fftc<192>() { return c_fft192; }

// ------------------------------------------------------------------------------------------
// Radix stages on register arrays.  Half-complex (FFTPACK) storage:
//   backward: in(ido,ip,l1) -> out(ido,l1,ip) ; forward: in(ido,l1,ip) -> out(ido,ip,l1)
// p = index of a real part (1,3,5..), p+1 its imaginary part, (q,q+1) = (ido-2-p, ido-1-p) the
// conjugate-mirrored pair.  All indices are compile-time after unrolling -> pure VGPR code.
// ------------------------------------------------------------------------------------------
template <int IDO, int L1, class W>
__device__ __forceinline__ void bwd4(const double (&in)[IDO * 4 * L1], double (&out)[IDO * 4 * L1],
                                     const W &w, double sqrt2)
{
#define I4(i, j, k) in[(i) + IDO * ((j) + 4 * (k))]
#define O4(i, k, j) out[(i) + IDO * ((k) + L1 * (j))]
    UNROLL for (int k = 0; k < L1; ++k) {
        const double tr1 = I4(0, 0, k) - I4(IDO - 1, 3, k), tr2 = I4(0, 0, k) + I4(IDO - 1, 3, k);
        const double tr3 = I4(IDO - 1, 1, k) + I4(IDO - 1, 1, k), tr4 = I4(0, 2, k) + I4(0, 2, k);
        O4(0, k, 0) = tr2 + tr3;  O4(0, k, 1) = tr1 - tr4;
        O4(0, k, 2) = tr2 - tr3;  O4(0, k, 3) = tr1 + tr4;
        UNROLL for (int p = 1; p + 1 < IDO; p += 2) {
            const int q = IDO - 2 - p;
            const double ti1 = I4(p + 1, 0, k) + I4(q + 1, 3, k), ti2 = I4(p + 1, 0, k) - I4(q + 1, 3, k);
            const double ti3 = I4(p + 1, 2, k) - I4(q + 1, 1, k), tr4b = I4(p + 1, 2, k) + I4(q + 1, 1, k);
            const double tr1b = I4(p, 0, k) - I4(q, 3, k), tr2b = I4(p, 0, k) + I4(q, 3, k);
            const double ti4 = I4(p, 2, k) - I4(q, 1, k), tr3b = I4(p, 2, k) + I4(q, 1, k);
            const double cr3 = tr2b - tr3b, ci3 = ti2 - ti3, cr2 = tr1b - tr4b, cr4 = tr1b + tr4b;
            const double ci2 = ti1 + ti4, ci4 = ti1 - ti4;
            O4(p, k, 0) = tr2b + tr3b;
            O4(p + 1, k, 0) = ti2 + ti3;
            O4(p, k, 1) = w[p - 1] * cr2 - w[p] * ci2;
            O4(p + 1, k, 1) = w[p - 1] * ci2 + w[p] * cr2;
            O4(p, k, 2) = w[IDO + p - 1] * cr3 - w[IDO + p] * ci3;
            O4(p + 1, k, 2) = w[IDO + p - 1] * ci3 + w[IDO + p] * cr3;
            O4(p, k, 3) = w[2 * IDO + p - 1] * cr4 - w[2 * IDO + p] * ci4;
            O4(p + 1, k, 3) = w[2 * IDO + p - 1] * ci4 + w[2 * IDO + p] * cr4;
        }
        if (IDO % 2 == 0) {
            const double ti1 = I4(0, 1, k) + I4(0, 3, k), ti2 = I4(0, 3, k) - I4(0, 1, k);
            const double ur1 = I4(IDO - 1, 0, k) - I4(IDO - 1, 2, k), ur2 = I4(IDO - 1, 0, k) + I4(IDO - 1, 2, k);
            O4(IDO - 1, k, 0) = ur2 + ur2;
            O4(IDO - 1, k, 1) = sqrt2 * (ur1 - ti1);
            O4(IDO - 1, k, 2) = ti2 + ti2;
            O4(IDO - 1, k, 3) = -sqrt2 * (ur1 + ti1);
        }
    }
#undef I4
#undef O4
}

// radix-3 backward with ido = 1 (the last backward stage): no twiddles
template <int L1>
__device__ __forceinline__ void bwd3_ido1(const double (&in)[3 * L1], double (&out)[3 * L1], double taui)
{
    UNROLL for (int k = 0; k < L1; ++k) {
        const double tr2 = in[1 + 3 * k] + in[1 + 3 * k];
        const double cr2 = in[3 * k] + (-0.5) * tr2;
        const double ci3 = taui * (in[2 + 3 * k] + in[2 + 3 * k]);
        out[k] = in[3 * k] + tr2;
        out[k + L1] = cr2 - ci3;
        out[k + 2 * L1] = cr2 + ci3;
    }
}

// radix-3 forward with ido = 1 (the first forward stage)
template <int L1>
__device__ __forceinline__ void fwd3_ido1(const double (&in)[3 * L1], double (&out)[3 * L1], double taui)
{
    UNROLL for (int k = 0; k < L1; ++k) {
        const double cr2 = in[k + L1] + in[k + 2 * L1];
        out[3 * k] = in[k] + cr2;
        out[2 + 3 * k] = taui * (in[k + 2 * L1] - in[k + L1]);
        out[1 + 3 * k] = in[k] + (-0.5) * cr2;
    }
}

template <int IDO, int L1, class W>
__device__ __forceinline__ void fwd4(const double (&in)[IDO * 4 * L1], double (&out)[IDO * 4 * L1],
                                     const W &w, double hsqt2)
{
#define I4(i, k, j) in[(i) + IDO * ((k) + L1 * (j))]
#define O4(i, j, k) out[(i) + IDO * ((j) + 4 * (k))]
    UNROLL for (int k = 0; k < L1; ++k) {
        const double tr1 = I4(0, k, 1) + I4(0, k, 3), tr2 = I4(0, k, 0) + I4(0, k, 2);
        O4(0, 0, k) = tr1 + tr2;
        O4(IDO - 1, 3, k) = tr2 - tr1;
        O4(IDO - 1, 1, k) = I4(0, k, 0) - I4(0, k, 2);
        O4(0, 2, k) = I4(0, k, 3) - I4(0, k, 1);
        UNROLL for (int p = 1; p + 1 < IDO; p += 2) {
            const int q = IDO - 2 - p;
            const double cr2 = w[p - 1] * I4(p, k, 1) + w[p] * I4(p + 1, k, 1);
            const double ci2 = w[p - 1] * I4(p + 1, k, 1) - w[p] * I4(p, k, 1);
            const double cr3 = w[IDO + p - 1] * I4(p, k, 2) + w[IDO + p] * I4(p + 1, k, 2);
            const double ci3 = w[IDO + p - 1] * I4(p + 1, k, 2) - w[IDO + p] * I4(p, k, 2);
            const double cr4 = w[2 * IDO + p - 1] * I4(p, k, 3) + w[2 * IDO + p] * I4(p + 1, k, 3);
            const double ci4 = w[2 * IDO + p - 1] * I4(p + 1, k, 3) - w[2 * IDO + p] * I4(p, k, 3);
            const double sr1 = cr2 + cr4, sr4 = cr4 - cr2, si1 = ci2 + ci4, si4 = ci2 - ci4;
            const double si2 = I4(p + 1, k, 0) + ci3, si3 = I4(p + 1, k, 0) - ci3;
            const double sr2 = I4(p, k, 0) + cr3, sr3 = I4(p, k, 0) - cr3;
            O4(p, 0, k) = sr1 + sr2;
            O4(q, 3, k) = sr2 - sr1;
            O4(p + 1, 0, k) = si1 + si2;
            O4(q + 1, 3, k) = si1 - si2;
            O4(p, 2, k) = si4 + sr3;
            O4(q, 1, k) = sr3 - si4;
            O4(p + 1, 2, k) = sr4 + si3;
            O4(q + 1, 1, k) = sr4 - si3;
        }
        if (IDO % 2 == 0) {
            const double ti1 = -hsqt2 * (I4(IDO - 1, k, 1) + I4(IDO - 1, k, 3));
            const double ur1 = hsqt2 * (I4(IDO - 1, k, 1) - I4(IDO - 1, k, 3));
            O4(IDO - 1, 0, k) = ur1 + I4(IDO - 1, k, 0);
            O4(IDO - 1, 2, k) = I4(IDO - 1, k, 0) - ur1;
            O4(0, 1, k) = ti1 - I4(IDO - 1, k, 2);
            O4(0, 3, k) = ti1 + I4(IDO - 1, k, 2);
        }
    }
#undef I4
#undef O4
}

// 48-point backward sub-transform: radix 4 (ido 12), radix 4 (ido 3), radix 3 (ido 1).
// Input in x (clobbered), natural-order result in y.
__device__ __forceinline__ void sub48_backward_c(const FftConstants &c, double (&x)[48], double (&y)[48])
{
    bwd4<12, 1>(x, y, c.a, c.sqrt2);
    bwd4<3, 4>(y, x, c.b, c.sqrt2);
    bwd3_ido1<16>(x, y, c.taui);
}
__device__ __forceinline__ void sub48_forward_c(const FftConstants &c, double (&x)[48], double (&y)[48])
{
    fwd3_ido1<16>(x, y, c.taui);
    fwd4<3, 4>(y, x, c.b, c.hsqt2);
    fwd4<12, 1>(x, y, c.a, c.hsqt2);
}

// The 49 stage constants of the 48-point sub-transforms ({a[36], b[9], taui, sqrt2, hsqt2, scale} of FftConstants) held in
// TWO VGPRs -- dword i in lane i -- and picked out with v_readlane (lane index = compile-time immediate, result = an SGPR
// operand of the FP64 instruction).  As scalar loads from constant memory the compiler fetches them just in time, in ~10
// groups per sub-transform (there are not 98 free SGPRs), and every group is an exposed scalar-cache round trip -- ~2 k
// ticks of a 48-point transform whose arithmetic takes 1.3 k.
struct LaneConst {
    unsigned t0, t1;
    __device__ __forceinline__ void load(const FftConstants &c, int lane)
    {
        const unsigned *d = reinterpret_cast<const unsigned *>(c.a);
        t0 = d[lane];
        t1 = lane < 34 ? d[64 + lane] : 0u;
    }
    __device__ __forceinline__ double get(int i) const
    {
        const int d = 2 * i;
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(d < 64 ? t0 : t1), d & 63);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(d + 1 < 64 ? t0 : t1), (d + 1) & 63);
        return __longlong_as_double(((unsigned long long)hi << 32) | lo);
    }
};
struct LaneTab {
    LaneConst c;
    int base;
    __device__ __forceinline__ double operator[](int i) const { return c.get(base + i); }
};
__device__ __forceinline__ void sub48_backward_l(const LaneConst &lc, double (&x)[48], double (&y)[48])
{
    const double sqrt2 = lc.get(46);
    bwd4<12, 1>(x, y, LaneTab{lc, 0}, sqrt2);
    bwd4<3, 4>(y, x, LaneTab{lc, 36}, sqrt2);
    bwd3_ido1<16>(x, y, lc.get(45));
}
__device__ __forceinline__ void sub48_forward_l(const LaneConst &lc, double (&x)[48], double (&y)[48])
{
    const double hsqt2 = lc.get(47);
    fwd3_ido1<16>(x, y, lc.get(45));
    fwd4<3, 4>(y, x, LaneTab{lc, 36}, hsqt2);
    fwd4<12, 1>(x, y, LaneTab{lc, 0}, hsqt2);
}

template <int NF>
__device__ __forceinline__ void sub48_backward(double (&x)[48], double (&y)[48])
{
    const FftConstants &c = fftc<NF>();
    bwd4<12, 1>(x, y, c.a, c.sqrt2);
    bwd4<3, 4>(y, x, c.b, c.sqrt2);
    bwd3_ido1<16>(x, y, c.taui);
}

// 48-point forward sub-transform: radix 3 (ido 1), radix 4 (ido 3), radix 4 (ido 12).
template <int NF>
__device__ __forceinline__ void sub48_forward(double (&x)[48], double (&y)[48])
{
    const FftConstants &c = fftc<NF>();
    fwd3_ido1<16>(x, y, c.taui);
    fwd4<3, 4>(y, x, c.b, c.hsqt2);
    fwd4<12, 1>(x, y, c.a, c.hsqt2);
}

// ------------------------------------------------------------------------------------------
// fourier_inv  (fourier.f90:23-53 + fftpack rfftb1): four[nb*il][fs] -> grid[nb*il][NF]
// Block = 64 rows x TPR lanes-per-row; thread = h*64 + r, so h (which 48-point column of the
// ido=48 stage this lane owns) is wave-uniform.
// ------------------------------------------------------------------------------------------
// rows per block: the LDS tile (rows x (N+1) doubles) decides how many blocks share a CU -- at N = 192 a
// 64-row tile is 98.8 KB (one block, 4 waves per CU: load, FFT and store phases cannot overlap); 32 rows
// give 3 blocks per CU.  (SPDY_FFT_ROWS_192 for experiments.)
#ifndef SPDY_FFT_ROWS_192
#define SPDY_FFT_ROWS_192 32
#endif
constexpr int fft_rows(int nf) { return nf == 192 ? SPDY_FFT_ROWS_192 : 64; }

template <int NF, int TWO_MX>
__global__ __launch_bounds__(fft_rows(NF) *(NF / 48), 2) void fourier_inv_kernel(
    const double *__restrict__ four, double *__restrict__ grid, const double *__restrict__ cosgr,
    const int *__restrict__ d_kcos, int kcos_all, int nrows, int il, int fs)
{
    constexpr int FFT_ROWS = fft_rows(NF), TPR = NF / 48, RS = NF + 1, NT = FFT_ROWS * TPR;
    constexpr int LASTF = TWO_MX - 2;   // fvar[i] = F[i+1] for 1 <= i <= LASTF, zero beyond
    constexpr int PER_ROW = (TWO_MX + 15) / 16 * 8, NPRE = FFT_ROWS * PER_ROW / NT;   // 16-byte pieces per row (= fs/2), per thread
    static_assert(FFT_ROWS * PER_ROW % NT == 0, "whole pieces per thread");
    __shared__ double tile[FFT_ROWS * RS];
    __shared__ double tw[(TPR - 1) * 48];   // ido=48 twiddles: the index depends on the lane's column h -> LDS, not scalar loads
    __shared__ double rowsc[FFT_ROWS];      // per-row output factor
    const int tid = threadIdx.x, r = tid & (FFT_ROWS - 1), h = tid / FFT_ROWS;
    const int ntiles = (nrows + FFT_ROWS - 1) / FFT_ROWS;
    for (int e = tid; e < (TPR - 1) * 48; e += NT) tw[e] = fftc<NF>().first[e];

    // The block walks over row tiles; the next tile's Fourier rows are fetched into registers (16 B per lane,
    // coalesced) while the current one is transformed: with one wave per 16 rows and 24 KB of LDS per wave only
    // ~6 waves fit a CU, so the loads have to overlap the arithmetic inside the wave.
    double2 pre[NPRE];
#define FINV_FETCH(tile_)                                                                          \
    do {                                                                                           \
        const long r0_ = (long)(tile_) * FFT_ROWS;                                                 \
        const double2 *src_ = reinterpret_cast<const double2 *>(four + r0_ * fs);                  \
        const long lim_ = ((long)nrows - r0_) * PER_ROW; /* pieces of rows that exist */           \
        UNROLL for (int i = 0; i < NPRE; ++i) {                                                    \
            const int e_ = tid + i * NT;                                                           \
            pre[i] = e_ < lim_ ? src_[e_] : make_double2(0.0, 0.0);                                \
        }                                                                                          \
    } while (0)
    if ((int)blockIdx.x < ntiles) FINV_FETCH(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long row0 = (long)t * FFT_ROWS;
    const int nvalid = (int)min((long)FFT_ROWS, (long)nrows - row0);

    // phase 1: the prefetched Fourier rows (first TWO_MX doubles of each fs-stride row) into the LDS tile
    __syncthreads();                                       // previous tile's store phase is done with the LDS tile
    int tl = tid;
    asm volatile("" : "+v"(tl));                           // per tile: 2 x NPRE hoisted LDS offsets would spill next to the FFT
    UNROLL for (int i = 0; i < NPRE; ++i) {
        const int e = tl + i * NT, rr = e / PER_ROW, c2 = e - rr * PER_ROW;
        if (2 * c2 < TWO_MX) {
            tile[rr * RS + 2 * c2] = pre[i].x;
            tile[rr * RS + 2 * c2 + 1] = pre[i].y;
        }
    }
    if (t + (int)gridDim.x < ntiles) FINV_FETCH(t + gridDim.x);
    __syncthreads();

    // phase 2: ido=48 stage (this lane's column h only), then the 48-point sub-transform
    double x[48], y[48];
    {
        const double *rowp = tile + r * RS;
        const FftConstants &c = fftc<NF>();
        auto fv = [&](int i) -> double { return i == 0 ? rowp[0] : (i <= LASTF ? rowp[i + 1] : 0.0); };
        if (TPR == 2) {   // radix 2: in(48,2) -> out(48,1,2)   (fftpack radb2)
            if (h == 0) {
                x[0] = fv(0) + fv(95);
                UNROLL for (int p = 1; p < 47; p += 2) {
                    const int q = 46 - p;
                    x[p] = fv(p) + fv(48 + q);
                    x[p + 1] = fv(p + 1) - fv(48 + q + 1);
                }
                x[47] = fv(47) + fv(47);
            } else {
                x[0] = fv(0) - fv(95);
                UNROLL for (int p = 1; p < 47; p += 2) {
                    const int q = 46 - p;
                    const double tr2 = fv(p) - fv(48 + q), ti2 = fv(p + 1) + fv(48 + q + 1);
                    x[p] = c.first[p - 1] * tr2 - c.first[p] * ti2;
                    x[p + 1] = c.first[p - 1] * ti2 + c.first[p] * tr2;
                }
                x[47] = -(fv(48) + fv(48));
            }
        } else {          // radix 4: in(48,4) -> out(48,1,4)   (fftpack radb4)
#define CCF(i, j) fv((i) + 48 * (j))
            {
                const double tr1 = CCF(0, 0) - CCF(47, 3), tr2 = CCF(0, 0) + CCF(47, 3);
                const double tr3 = CCF(47, 1) + CCF(47, 1), tr4 = CCF(0, 2) + CCF(0, 2);
                x[0] = h == 0 ? tr2 + tr3 : h == 1 ? tr1 - tr4 : h == 2 ? tr2 - tr3 : tr1 + tr4;
            }
            UNROLL for (int p = 1; p < 47; p += 2) {
                const int q = 46 - p;
                const double ti1 = CCF(p + 1, 0) + CCF(q + 1, 3), ti2 = CCF(p + 1, 0) - CCF(q + 1, 3);
                const double ti3 = CCF(p + 1, 2) - CCF(q + 1, 1), tr4 = CCF(p + 1, 2) + CCF(q + 1, 1);
                const double tr1 = CCF(p, 0) - CCF(q, 3), tr2 = CCF(p, 0) + CCF(q, 3);
                const double ti4 = CCF(p, 2) - CCF(q, 1), tr3 = CCF(p, 2) + CCF(q, 1);
                if (h == 0) {
                    x[p] = tr2 + tr3;
                    x[p + 1] = ti2 + ti3;
                } else {
                    const double cr = h == 1 ? tr1 - tr4 : h == 2 ? tr2 - tr3 : tr1 + tr4;
                    const double ci = h == 1 ? ti1 + ti4 : h == 2 ? ti2 - ti3 : ti1 - ti4;
                    const double wr = tw[48 * (h - 1) + p - 1], wi = tw[48 * (h - 1) + p];
                    x[p] = wr * cr - wi * ci;
                    x[p + 1] = wr * ci + wi * cr;
                }
            }
            {
                const double ti1 = CCF(0, 1) + CCF(0, 3), ti2 = CCF(0, 3) - CCF(0, 1);
                const double tr1 = CCF(47, 0) - CCF(47, 2), tr2 = CCF(47, 0) + CCF(47, 2);
                x[47] = h == 0 ? tr2 + tr2 : h == 1 ? c.sqrt2 * (tr1 - ti1) : h == 2 ? ti2 + ti2 : -c.sqrt2 * (tr1 + ti1);
            }
#undef CCF
        }
    }
    sub48_backward<NF>(x, y);
    __syncthreads();

    // phase 3: element t of column h is grid point TPR*t + h
    {
        double *rowp = tile + r * RS;
        UNROLL for (int t = 0; t < 48; ++t) rowp[TPR * t + h] = y[t];
        if (h == 0) {                                  // this row's output factor: 1 (kcos == 1) or 1/cos(lat)
            const int grow = (int)row0 + r, fld = grow / il, j = grow - fld * il;
            int kc = 1;
            if (r < nvalid) kc = d_kcos ? d_kcos[fld] : kcos_all;
            rowsc[r] = kc == 1 ? 1.0 : cosgr[j];
        }
    }
    __syncthreads();

    // phase 4: coalesced store (+ optional 1/cos(lat) scaling, fourier.f90:47-51; the factor of each row was put
    // into LDS by one lane per row -- a 64-bit divide per 16-byte piece used to dominate this phase)
    {
        double2 *dst = reinterpret_cast<double2 *>(grid + row0 * NF);
        constexpr int per_row = NF / 2;
        int tl4 = tid;
        asm volatile("" : "+v"(tl4));                  // per tile (no hoisting of the unrolled offsets into registers)
        UNROLL for (int i = 0; i < FFT_ROWS * per_row / NT; ++i) {
            const int e = tl4 + i * NT, rr = e / per_row, c2 = e - rr * per_row;
            if (rr < nvalid) {
                const double sc = rowsc[rr];
                dst[e] = make_double2(tile[rr * RS + 2 * c2] * sc, tile[rr * RS + 2 * c2 + 1] * sc);
            }
        }
    }
    }   // tile loop
#undef FINV_FETCH
}

// ------------------------------------------------------------------------------------------
// fourier_dir  (fourier.f90:56-82 + fftpack rfftf1): grid[nb*il][NF] -> four[nb*il][fs]
// ------------------------------------------------------------------------------------------
template <int NF, int TWO_MX>
__global__ __launch_bounds__(fft_rows(NF) *(NF / 48), 2) void fourier_dir_kernel(
    const double *__restrict__ grid, const double *__restrict__ gscale, double *__restrict__ four,
    int nrows, int il, int fs)
{
    constexpr int FFT_ROWS = fft_rows(NF), TPR = NF / 48, RS = NF + 1, NT = FFT_ROWS * TPR;
    constexpr int PER_ROW = NF / 2, NPRE = FFT_ROWS * PER_ROW / NT;   // 16-byte pieces per grid row, per thread
    static_assert(FFT_ROWS * PER_ROW % NT == 0, "whole pieces per thread");
    __shared__ double tile[FFT_ROWS * RS];
    __shared__ double tw[(TPR - 1) * 48];   // ido=48 twiddles, indexed per lane in phase 3 -> LDS, not constant-memory vector loads
    const int tid = threadIdx.x, r = tid & (FFT_ROWS - 1), h = tid / FFT_ROWS;
    const int ntiles = (nrows + FFT_ROWS - 1) / FFT_ROWS;
    const FftConstants &c = fftc<NF>();
    for (int e = tid; e < (TPR - 1) * 48; e += NT) tw[e] = c.first[e];

    // persistent over row tiles with a register prefetch of the next tile's grid rows (see fourier_inv_kernel)
    // the first NEARLY pieces are fetched before the FFT (as many as fit next to its 192 registers), the rest after it
    constexpr int NEARLY = NPRE / 2;
    double2 pre[NPRE];
#define FDIR_FETCH(tile_, I0, I1)                                                                  \
    do {                                                                                           \
        const long r0_ = (long)(tile_) * FFT_ROWS;                                                 \
        const double2 *src_ = reinterpret_cast<const double2 *>(grid + r0_ * NF);                  \
        const long lim_ = ((long)nrows - r0_) * PER_ROW;                                           \
        UNROLL for (int i = (I0); i < (I1); ++i) {                                                 \
            const int e_ = tid + i * NT;                                                           \
            pre[i] = e_ < lim_ ? src_[e_] : make_double2(0.0, 0.0);                                \
        }                                                                                          \
    } while (0)
    if ((int)blockIdx.x < ntiles) FDIR_FETCH(blockIdx.x, 0, NPRE);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long row0 = (long)t * FFT_ROWS;
    const int nvalid = (int)min((long)FFT_ROWS, (long)nrows - row0);

    // phase 1: the prefetched grid rows into the LDS tile (optionally scaled per latitude: vdspec)
    __syncthreads();                                       // previous tile's store phase is done with the LDS tile
    int tl = tid;
    asm volatile("" : "+v"(tl));                           // per tile: 2 x NPRE hoisted LDS offsets would spill next to the FFT
    UNROLL for (int i = 0; i < NPRE; ++i) {
        const int e = tl + i * NT, rr = e / PER_ROW, c2 = e - rr * PER_ROW;
        double2 v = pre[i];
        if (gscale) {
            const double sc = gscale[((int)row0 + rr) % il];
            v.x *= sc; v.y *= sc;
        }
        tile[rr * RS + 2 * c2] = v.x;
        tile[rr * RS + 2 * c2 + 1] = v.y;
    }
    if (t + (int)gridDim.x < ntiles) FDIR_FETCH(t + gridDim.x, 0, NEARLY);   // in flight during the FFT
    __syncthreads();

    // phase 2: decimated samples TPR*t + h -> 48-point forward sub-transform
    double x[48], y[48];
    {
        const double *rowp = tile + r * RS;
        UNROLL for (int t = 0; t < 48; ++t) x[t] = rowp[TPR * t + h];
    }
    sub48_forward<NF>(x, y);
    __syncthreads();
    {
        double *rowp = tile + r * RS + 48 * h;
        UNROLL for (int i = 0; i < 48; ++i) rowp[i] = y[i];
    }
    if (t + (int)gridDim.x < ntiles) FDIR_FETCH(t + gridDim.x, NEARLY, NPRE);   // the FFT registers are dead: the rest of the next tile
    __syncthreads();

    // phase 3: the ido=48 stage, in place.  Butterflies p and 46-p touch the same 4*TPR slots,
    // so one lane does both: 13 work items per row (item 0 = the i=0 and i=47 specials).
    {
        // r, h re-derived here: addresses hoisted out of the tile loop get spilled next to the FFT, and the
        // scratch reload's vmcnt wait would also wait for the next tile's prefetch issued just above
        int tid3 = tid;
        asm volatile("" : "+v"(tid3));
        const int r3 = tid3 & (FFT_ROWS - 1), h3 = tid3 / FFT_ROWS;
        double *z = tile + r3 * RS;
        for (int it = h3; it < 13; it += TPR) {
            if (TPR == 2) {        // fftpack radf2, ido=48, l1=1: in(48,1,2) -> out(48,2,1)
                if (it == 0) {
                    const double a0 = z[0], a1 = z[48], b0 = z[47], b1 = z[95];
                    z[0] = a0 + a1;  z[95] = a0 - a1;  z[48] = -b1;  z[47] = b0;
                } else {
                    const int p = 2 * it - 1, q = 46 - p;
                    const double wr = tw[p - 1], wi = tw[p];
                    const double a0r = z[p], a0i = z[p + 1], a1r = z[48 + p], a1i = z[48 + p + 1];
                    const double tr2 = wr * a1r + wi * a1i, ti2 = wr * a1i - wi * a1r;
                    if (p == q) {
                        z[p + 1] = a0i + ti2;  z[48 + q + 1] = ti2 - a0i;
                        z[p] = a0r + tr2;      z[48 + q] = a0r - tr2;
                    } else {
                        const double vr = tw[q - 1], vi = tw[q];
                        const double b0r = z[q], b0i = z[q + 1], b1r = z[48 + q], b1i = z[48 + q + 1];
                        const double sr2 = vr * b1r + vi * b1i, si2 = vr * b1i - vi * b1r;
                        z[p + 1] = a0i + ti2;  z[48 + q + 1] = ti2 - a0i;
                        z[p] = a0r + tr2;      z[48 + q] = a0r - tr2;
                        z[q + 1] = b0i + si2;  z[48 + p + 1] = si2 - b0i;
                        z[q] = b0r + sr2;      z[48 + p] = b0r - sr2;
                    }
                }
            } else {               // fftpack radf4, ido=48, l1=1: in(48,1,4) -> out(48,4,1)
                if (it == 0) {
                    const double a0 = z[0], a1 = z[48], a2 = z[96], a3 = z[144];
                    const double e0 = z[47], e1 = z[95], e2 = z[143], e3 = z[191];
                    const double tr1 = a1 + a3, tr2 = a0 + a2;
                    const double ti1 = -c.hsqt2 * (e1 + e3), ur1 = c.hsqt2 * (e1 - e3);
                    z[0] = tr1 + tr2;        z[144 + 47] = tr2 - tr1;
                    z[48 + 47] = a0 - a2;    z[96] = a3 - a1;
                    z[47] = ur1 + e0;        z[96 + 47] = e0 - ur1;
                    z[48] = ti1 - e2;        z[144] = ti1 + e2;
                } else {
                    const int p = 2 * it - 1, q = 46 - p;
                    // both butterflies' inputs first (each writes slots the other reads), as named scalars: an
                    // array indexed by the (early-exit) loop variable ended up in scratch memory
#define RADF4_LOAD(pp_, a0, a1, a2, a3, a4, a5, a6, a7)                                                   \
    const double a0 = z[pp_], a1 = z[(pp_) + 1], a2 = z[48 + (pp_)], a3 = z[48 + (pp_) + 1],               \
                 a4 = z[96 + (pp_)], a5 = z[96 + (pp_) + 1], a6 = z[144 + (pp_)], a7 = z[144 + (pp_) + 1]
#define RADF4_BFLY(pp_, qq_, a0, a1, a2, a3, a4, a5, a6, a7)                                              \
    do {                                                                                                  \
        const double cr2 = tw[(pp_) - 1] * a2 + tw[pp_] * a3, ci2 = tw[(pp_) - 1] * a3 - tw[pp_] * a2;     \
        const double cr3 = tw[48 + (pp_) - 1] * a4 + tw[48 + (pp_)] * a5, ci3 = tw[48 + (pp_) - 1] * a5 - tw[48 + (pp_)] * a4; \
        const double cr4 = tw[96 + (pp_) - 1] * a6 + tw[96 + (pp_)] * a7, ci4 = tw[96 + (pp_) - 1] * a7 - tw[96 + (pp_)] * a6; \
        const double sr1 = cr2 + cr4, sr4 = cr4 - cr2, si1 = ci2 + ci4, si4 = ci2 - ci4;                   \
        const double si2 = a1 + ci3, si3 = a1 - ci3, sr2 = a0 + cr3, sr3 = a0 - cr3;                       \
        z[pp_] = sr1 + sr2;            z[144 + (qq_)] = sr2 - sr1;                                         \
        z[(pp_) + 1] = si1 + si2;      z[144 + (qq_) + 1] = si1 - si2;                                     \
        z[96 + (pp_)] = si4 + sr3;     z[48 + (qq_)] = sr3 - si4;                                          \
        z[96 + (pp_) + 1] = sr4 + si3; z[48 + (qq_) + 1] = sr4 - si3;                                      \
    } while (0)
                    RADF4_LOAD(p, pa0, pa1, pa2, pa3, pa4, pa5, pa6, pa7);
                    RADF4_LOAD(q, qa0, qa1, qa2, qa3, qa4, qa5, qa6, qa7);
                    RADF4_BFLY(p, q, pa0, pa1, pa2, pa3, pa4, pa5, pa6, pa7);
                    if (p != q) RADF4_BFLY(q, p, qa0, qa1, qa2, qa3, qa4, qa5, qa6, qa7);
#undef RADF4_LOAD
#undef RADF4_BFLY
                }
            }
        }
    }
    __syncthreads();

    // phase 4: scale by float32(1/N), pack (a0, 0, re1, im1, ...) and store coalesced
    {
        double2 *dst = reinterpret_cast<double2 *>(four + row0 * fs);
        constexpr int per_row = (TWO_MX + 15) / 16 * 8;   // = fs / 2 (launcher checks)
        int tl4 = tid;
        asm volatile("" : "+v"(tl4));                  // per tile (no hoisting of the unrolled offsets into registers)
        UNROLL for (int i = 0; i < FFT_ROWS * per_row / NT; ++i) {
            const int e = tl4 + i * NT, rr = e / per_row, c2 = e - rr * per_row;
            if (rr < nvalid) {
                double2 v = make_double2(0.0, 0.0);
                const double *z = tile + rr * RS;
                if (c2 == 0) v.x = z[0] * c.scale;
                else if (2 * c2 < TWO_MX) { v.x = z[2 * c2 - 1] * c.scale; v.y = z[2 * c2] * c.scale; }
                dst[e] = v;
            }
        }
    }
    }   // tile loop
#undef FDIR_FETCH
}

// ------------------------------------------------------------------------------------------
// legendre_inv (legendre.f90:74-111) on the FP64 matrix cores.
// Block = (8 fields) x (8 zonal wavenumbers); wave w owns m' = 8*blockIdx.y + w.
//   D[j, c] = sum_n A[j, n] * B[n, c],  A = P(m', n, j) (16-latitude tile), B = X(n, c),
//   c = 2*field + (re|im); one accumulator pair (even-n, odd-n parity) per latitude tile.
// North/south rows: out(il-1-j) = even + odd, out(j) = even - odd.
// ------------------------------------------------------------------------------------------
constexpr int LEG_BT = 8;      // fields per block
constexpr int LEG_MG = 8;      // zonal wavenumbers per block (one 128-byte line of spec)
constexpr int LEG_NT = 64 * LEG_MG;

template <int JT>
__global__ __launch_bounds__(LEG_NT) void legendre_inv_kernel(DevPlan p, int nb, const double *__restrict__ spec,
                                                               double *__restrict__ four)
{
    extern __shared__ __attribute__((aligned(16))) double xs[];   // [w][parity][kp][16] (+2 pad per w)
    const int kp = 4 * p.ks_inv, wstride = 2 * kp * 16 + 2;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b0 = blockIdx.x * LEG_BT, m0 = blockIdx.y * LEG_MG;

    for (int e = tid; e < LEG_MG * wstride; e += LEG_NT) xs[e] = 0.0;
    __syncthreads();
    // stage X: one (re,im) pair per lane, 8 consecutive m' = one 128-byte line
    {
        const int total = LEG_BT * p.nx * LEG_MG;
        for (int e = tid; e < total; e += LEG_NT) {
            const int mi = e & (LEG_MG - 1), t = e >> 3, n = t % p.nx, b = t / p.nx;
            const int m = m0 + mi;
            if (b0 + b < nb && m < p.mx && m + n <= p.trunc + 1) {
                const double2 v = *reinterpret_cast<const double2 *>(spec + 2 * (((long)(b0 + b) * p.nx + n) * p.mx + m));
                double *d = xs + mi * wstride + ((n & 1) * kp + (n >> 1)) * 16 + 2 * b;
                d[0] = v.x;
                d[1] = v.y;
            }
        }
    }
    __syncthreads();

    const int m = m0 + w;
    const bool wvalid = m < p.mx;
    const int kact = p.nx - min(m, p.mx - 1);        // active n: 0 .. kact-1
    d4 acc[2][JT];
    UNROLL for (int q = 0; q < 2; ++q) {
        UNROLL for (int t = 0; t < JT; ++t) acc[q][t] = (d4){0.0, 0.0, 0.0, 0.0};
    }
    if (wvalid) {
        UNROLL for (int par = 0; par < 2; ++par) {
            const int cnt = (kact + 1 - par) >> 1, nks = (cnt + 3) >> 2;
            const double *bsrc = xs + w * wstride + par * kp * 16 + (lane >> 4) * 16 + (lane & 15);
            const double *asrc = p.pa_inv + ((long)(m * 2 + par) * p.ks_inv) * JT * 64 + lane;
            // A fragments stream from L2: fetch k-step ks+1 while the matrix cores work on ks
            double a_nxt[JT];
            UNROLL for (int t = 0; t < JT; ++t) a_nxt[t] = asrc[t * 64];
            for (int ks = 0; ks < nks; ++ks) {
                double a_cur[JT];
                UNROLL for (int t = 0; t < JT; ++t) a_cur[t] = a_nxt[t];
                const int kn = min(ks + 1, nks - 1);
                UNROLL for (int t = 0; t < JT; ++t) a_nxt[t] = asrc[(kn * JT + t) * 64];
                const double bv = bsrc[ks * 64];
                UNROLL for (int t = 0; t < JT; ++t)
                    acc[par][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[t], bv, acc[par][t], 0, 0, 0);
            }
        }
    }
    // Output through LDS, one 16-latitude tile (16 southern + 16 northern rows) at a time: the D layout
    // (row = (lane>>4) + 4*reg, col = lane&15 = 2*field + re/im) would give 8-byte stores with 16 contiguous
    // bytes each, which the memory system runs at a third of the rate of full lines (tools/store_issue.hip);
    // the block's 8 wavenumbers make one 128-byte line per (field, latitude row).
    // LDS image: [row 0..15 south j, 16..31 north il-1-j][field][18: 8 x (re, im) + pad]
    const int col = lane & 15, part = col & 1;
    UNROLL for (int t = 0; t < JT; ++t) {
        __syncthreads();                              // the B image / the previous tile's lines are no longer needed
        if (wvalid) {
            UNROLL for (int rg = 0; rg < 4; ++rg) {
                const int rowi = (lane >> 4) + 4 * rg;
                const double ev = acc[0][t][rg], od = acc[1][t][rg];
                xs[(rowi * LEG_BT + (col >> 1)) * 18 + 2 * w + part] = ev - od;           // row j
                xs[((16 + rowi) * LEG_BT + (col >> 1)) * 18 + 2 * w + part] = ev + od;    // row il-1-j
            }
        }
        __syncthreads();
        UNROLL for (int i = 0; i < 32 * LEG_BT * LEG_MG / LEG_NT; ++i) {
            const int e = tid + i * LEG_NT, mi = e & (LEG_MG - 1), fld = (e >> 3) & (LEG_BT - 1), rowi = e >> 6;
            const int jj = 16 * t + (rowi & 15), row = rowi < 16 ? jj : p.il - 1 - jj;
            if (jj < p.iy && b0 + fld < nb && m0 + mi < p.mx)
                *reinterpret_cast<double2 *>(four + ((long)(b0 + fld) * p.il + row) * p.fs + 2 * (m0 + mi)) =
                    *reinterpret_cast<const double2 *>(xs + (rowi * LEG_BT + fld) * 18 + 2 * mi);
        }
    }
}

// ------------------------------------------------------------------------------------------
// legendre_dir (legendre.f90:114-155) on the FP64 matrix cores.
//   D[n, c] = sum_j A[n, j] * B[j, c],  A = P(m', n, j)*wt(j) (16-row n tile of one parity),
//   B = F(il-1-j, c) +/- F(j, c) (+ for n even 0-based, - for n odd).
// ------------------------------------------------------------------------------------------
template <int NTD>
__global__ __launch_bounds__(LEG_NT) void legendre_dir_kernel(DevPlan p, int nb, const double *__restrict__ four,
                                                               double *__restrict__ spec)
{
    // The Fourier image of one tile (8 fields x il rows x 8 wavenumbers) fills most of the LDS, so only one block
    // fits a CU: the block is persistent over tiles and fetches the next tile's image into registers while the
    // matrix cores work on the current one (load, contraction and store phases would otherwise run back to back).
    constexpr int IL = NTD == 3 ? 96 : 48;                        // launcher checks p.il
    constexpr int NPRE = LEG_BT * IL * LEG_MG / LEG_NT;           // 16-byte pieces per thread and tile
    extern __shared__ __attribute__((aligned(16))) double fsm[];   // [w][il][16] (+2 pad per w)
    const int wstride = IL * 16 + 2;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nbx = (nb + LEG_BT - 1) / LEG_BT, ntiles = nbx * ((p.mx + LEG_MG - 1) / LEG_MG);
    double2 pre[NPRE];
#define LDIR_FETCH(tile_)                                                                                     \
    do {                                                                                                      \
        const int tb0_ = ((tile_) % nbx) * LEG_BT, tm0_ = ((tile_) / nbx) * LEG_MG;                           \
        UNROLL for (int i = 0; i < NPRE; ++i) {                                                               \
            const int e_ = tid + i * LEG_NT, mi_ = e_ & (LEG_MG - 1), t_ = e_ >> 3, row_ = t_ % IL, b_ = t_ / IL; \
            pre[i] = make_double2(0.0, 0.0);                                                                  \
            if (tb0_ + b_ < nb && tm0_ + mi_ < p.mx)                                                          \
                pre[i] = *reinterpret_cast<const double2 *>(four + ((long)(tb0_ + b_) * IL + row_) * p.fs + 2 * (tm0_ + mi_)); \
        }                                                                                                     \
    } while (0)
    if ((int)blockIdx.x < ntiles) LDIR_FETCH((int)blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b0 = (tile % nbx) * LEG_BT, m0 = (tile / nbx) * LEG_MG;
    __syncthreads();                                  // the previous tile's output lines have left the LDS
    {
        int tl = tid;
        asm volatile("" : "+v"(tl));                  // per tile: hoisted LDS offsets would cost 2 x NPRE registers
        UNROLL for (int i = 0; i < NPRE; ++i) {
            const int e = tl + i * LEG_NT, mi = e & (LEG_MG - 1), t = e >> 3, row = t % IL, b = t / IL;
            double *d = fsm + mi * wstride + row * 16 + 2 * b;
            d[0] = pre[i].x;
            d[1] = pre[i].y;
        }
    }
    if (tile + (int)gridDim.x < ntiles) LDIR_FETCH(tile + (int)gridDim.x);
    __syncthreads();

    const int m = m0 + w;
    const bool wvalid = m < p.mx;
    // active n (0-based): n <= trunc and m + n <= trunc + 1
    const int nact = min(p.trunc + 1, p.trunc + 2 - min(m, p.mx - 1));
    int ntile[2];
    UNROLL for (int par = 0; par < 2; ++par) ntile[par] = (((nact + 1 - par) >> 1) + 15) >> 4;
    d4 acc[2][NTD];
    UNROLL for (int q = 0; q < 2; ++q) {
        UNROLL for (int t = 0; t < NTD; ++t) acc[q][t] = (d4){0.0, 0.0, 0.0, 0.0};
    }
    if (wvalid) {
        // A fragments stream from L2 (they are shared by every field block): fetch k-step ks+1 while the matrix
        // cores work on ks, otherwise every k-step starts with an exposed L2 round trip
        const double *fw = fsm + w * wstride + (lane & 15);
        const double *abase = p.pa_dir + ((long)(m * 2) * p.nt_dir) * p.js_dir * 64 + lane;   // [(m*2+par)*nt_dir + t][ks][64]
        const long pstride = (long)p.nt_dir * p.js_dir * 64, tstride = (long)p.js_dir * 64;
        double a_nxt[2][NTD];
        UNROLL for (int par = 0; par < 2; ++par) {
            UNROLL for (int t = 0; t < NTD; ++t) a_nxt[par][t] = t < ntile[par] ? abase[par * pstride + t * tstride] : 0.0;
        }
        for (int ks = 0; ks < p.js_dir; ++ks) {
            double a_cur[2][NTD];
            UNROLL for (int par = 0; par < 2; ++par) {
                UNROLL for (int t = 0; t < NTD; ++t) a_cur[par][t] = a_nxt[par][t];
            }
            const int kn = min(ks + 1, p.js_dir - 1);
            UNROLL for (int par = 0; par < 2; ++par) {
                UNROLL for (int t = 0; t < NTD; ++t)
                    if (t < ntile[par]) a_nxt[par][t] = abase[par * pstride + t * tstride + kn * 64];
            }
            const int j = 4 * ks + (lane >> 4);
            const double south = fw[(p.il - 1 - j) * 16], north = fw[j * 16];
            const double bv[2] = {south + north, south - north};
            UNROLL for (int par = 0; par < 2; ++par) {
                UNROLL for (int t = 0; t < NTD; ++t) {
                    if (t < ntile[par])
                        acc[par][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[par][t], bv[par], acc[par][t], 0, 0, 0);
                }
            }
        }
    }
    // Output through LDS (see legendre_inv_kernel): image [n][field][18: 8 x (re, im) + pad], then one 128-byte
    // piece of a spectral row per 8 lanes.  Inactive coefficients are written as the zeros they are.
    __syncthreads();                                  // every wave is done with the Fourier image
    const int col = lane & 15, part = col & 1;
    if (wvalid) {
        UNROLL for (int par = 0; par < 2; ++par) {
            UNROLL for (int t = 0; t < NTD; ++t) {
                UNROLL for (int rg = 0; rg < 4; ++rg) {
                    const int n = 2 * (16 * t + (lane >> 4) + 4 * rg) + par;
                    if (n < p.nx) fsm[(n * LEG_BT + (col >> 1)) * 18 + 2 * w + part] = acc[par][t][rg];
                }
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < p.nx * LEG_BT * LEG_MG; e += LEG_NT) {
        const int mi = e & (LEG_MG - 1), fld = (e >> 3) & (LEG_BT - 1), n = e >> 6;
        if (b0 + fld < nb && m0 + mi < p.mx)
            *reinterpret_cast<double2 *>(spec + 2 * (((long)(b0 + fld) * p.nx + n) * p.mx + m0 + mi)) =
                *reinterpret_cast<const double2 *>(fsm + (n * LEG_BT + fld) * 18 + 2 * mi);
    }
    }   // tile loop
#undef LDIR_FETCH
}

// ------------------------------------------------------------------------------------------
// Spectral-space operators: one lane per complex coefficient (b, n, m).
// ------------------------------------------------------------------------------------------
// laplacian / inverse_laplacian / trunct (spectral.f90:84-96, 229-233)
__global__ void scale_op_kernel(DevPlan p, int op, long total, const double *__restrict__ in, double *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i % (p.mx * p.nx));
    const cpx z = ld(in, i);
    if (op == OP_LAPLACIAN) st(out, i, p.el2[e] * (-z));
    else if (op == OP_INV_LAPLACIAN) st(out, i, p.elm2[e] * (-z));
    else st(out, i, p.trfilt[e] * z);
}

// grad (spectral.f90:124-144)
__device__ __forceinline__ void grad_body(const DevPlan &p, long i, long total, const double *__restrict__ psi, double *__restrict__ dx,
                                          double *__restrict__ dy)
{
    if (i >= total) return;
    const int sz = p.mx * p.nx, e = (int)(i % sz), m = e % p.mx, n = e / p.mx;
    const long f0 = i - e;
    st(dx, i, times_i(p.gradx[m] * ld(psi, i)));
    cpx r;
    if (n == 0) r = p.gradyp[e] * ld(psi, f0 + m + p.mx);
    else if (n == p.nx - 1) r = (-p.gradym[e]) * ld(psi, f0 + m + (long)p.mx * p.trunc);
    else if (n <= p.trunc) r = (-p.gradym[e]) * ld(psi, i - p.mx) + p.gradyp[e] * ld(psi, i + p.mx);
    else return;
    st(dy, i, r);
}

__global__ void grad_kernel(DevPlan p, long total, const double *__restrict__ psi, double *__restrict__ dx, double *__restrict__ dy)
{
    grad_body(p, (long)blockIdx.x * blockDim.x + threadIdx.x, total, psi, dx, dy);
}

// vds (spectral.f90:146-171): (U,V) -> (vor,div)
__global__ void vds_kernel(DevPlan p, long total, const double *__restrict__ u, const double *__restrict__ v,
                           double *__restrict__ vor, double *__restrict__ dv)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int sz = p.mx * p.nx, e = (int)(i % sz), m = e % p.mx, n = e / p.mx;
    const long f0 = i - e;
    if (n == 0) {
        const cpx zp = times_i(p.gradx[m] * ld(u, i)), zc = times_i(p.gradx[m] * ld(v, i));
        st(vor, i, zc - p.vddyp[e] * ld(u, i + p.mx));
        st(dv, i, zp + p.vddyp[e] * ld(v, i + p.mx));
    } else if (n == p.nx - 1) {
        const long k = f0 + m + (long)p.mx * p.trunc;
        st(vor, i, p.vddym[e] * ld(u, k));
        st(dv, i, (-p.vddym[e]) * ld(v, k));
    } else if (n <= p.trunc) {
        const cpx zp = times_i(p.gradx[m] * ld(u, i)), zc = times_i(p.gradx[m] * ld(v, i));
        st(vor, i, (p.vddym[e] * ld(u, i - p.mx) - p.vddyp[e] * ld(u, i + p.mx)) + zc);
        st(dv, i, ((-p.vddym[e]) * ld(v, i - p.mx) + p.vddyp[e] * ld(v, i + p.mx)) + zp);
    }
}

// uvspec (spectral.f90:173-196): (vor,div) -> (U,V)
__device__ __forceinline__ void uvspec_body(const DevPlan &p, long i, long total, const double *__restrict__ vor,
                                            const double *__restrict__ dv, double *__restrict__ u, double *__restrict__ v)
{
    if (i >= total) return;
    const int sz = p.mx * p.nx, e = (int)(i % sz), m = e % p.mx, n = e / p.mx;
    const long f0 = i - e;
    if (n == 0) {
        const cpx zp = times_i(p.uvdx[e] * ld(vor, i)), zc = times_i(p.uvdx[e] * ld(dv, i));
        st(u, i, zc - p.uvdyp[e] * ld(vor, i + p.mx));
        st(v, i, zp + p.uvdyp[e] * ld(dv, i + p.mx));
    } else if (n == p.nx - 1) {
        const long k = f0 + m + (long)p.mx * p.trunc;
        st(u, i, p.uvdym[e] * ld(vor, k));
        st(v, i, (-p.uvdym[e]) * ld(dv, k));
    } else if (n <= p.trunc) {
        const cpx zp = times_i(p.uvdx[e] * ld(vor, i)), zc = times_i(p.uvdx[e] * ld(dv, i));
        st(v, i, ((-p.uvdym[e]) * ld(dv, i - p.mx) + p.uvdyp[e] * ld(dv, i + p.mx)) + zp);
        st(u, i, (p.uvdym[e] * ld(vor, i - p.mx) - p.uvdyp[e] * ld(vor, i + p.mx)) + zc);
    }
}

__global__ void uvspec_kernel(DevPlan p, long total, const double *__restrict__ vor, const double *__restrict__ dv,
                              double *__restrict__ u, double *__restrict__ v)
{
    uvspec_body(p, (long)blockIdx.x * blockDim.x + threadIdx.x, total, vor, dv, u, v);
}

// uvspec of one set of fields and grad of another in ONE launch (blockIdx.y): what a step applies before its inverse batch
__global__ void uvspec_grad_kernel(DevPlan p, long total_uv, const double *__restrict__ vor, const double *__restrict__ dv,
                                   double *__restrict__ u, double *__restrict__ v, long total_gr, const double *__restrict__ psi,
                                   double *__restrict__ dx, double *__restrict__ dy)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0) uvspec_body(p, i, total_uv, vor, dv, u, v);
    else grad_body(p, i, total_gr, psi, dx, dy);
}

// do_horizontal_diffusion (horizontal_diffusion.f90:86-105)
__global__ void hdiff_kernel(int sz, long total, const double *__restrict__ field, const double *__restrict__ fdt,
                             const double *__restrict__ dmp, const double *__restrict__ dmp1, double *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i % sz);
    st(out, i, dmp1[e] * (ld(fdt, i) - dmp[e] * ld(field, i)));
}

#include "spdy_fused_t30.inc"
#include "spdy_fused_t63.inc"

// ------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------
// grid limit of the persistent kernels: 3 blocks per CU (the FFT kernels' LDS footprint allows exactly that)
static int fft_grid_limit(const DevPlan &p) { return 3 * (p.num_cu > 0 ? p.num_cu : 256); }

static size_t legendre_inv_lds(const DevPlan &p) { return sizeof(double) * std::max(LEG_MG * (2 * 4 * p.ks_inv * 16 + 2), 32 * LEG_BT * 18); }
static size_t legendre_dir_lds(const DevPlan &p) { return sizeof(double) * std::max(LEG_MG * (p.il * 16 + 2), p.nx * LEG_BT * 18); }

__global__ void wave_placement_kernel(int *out);
// Function attributes are per device: the plan calls this once after hipSetDevice (spdy_plan_create), so a second
// plan on another GPU of the same process gets its own > 64 KB dynamic-LDS limits.
hipError_t prepare_device_kernels()
{
    struct { const void *fn; int bytes; } big[] = {
#define SPDY_K2(k_, m_, b_) {reinterpret_cast<const void *>(k_<m_, false>), b_}, {reinterpret_cast<const void *>(k_<m_, true>), b_}
        SPDY_K2(s2g_fused_t30_kernel, 0, t30::S2G_LDS), SPDY_K2(s2g_fused_t30_kernel, 1, t30::S2G_LDS),
        SPDY_K2(s2g_fused_t30_kernel, 2, t30::S2G_LDS), SPDY_K2(s2g_fused_t30_kernel, 3, t30::S2G_LDS),
        {reinterpret_cast<const void *>(s2g_fused_t30_kernel<0, false, true>), t30::S2G_LDS}, {reinterpret_cast<const void *>(s2g_fused_t30_kernel<1, false, true>), t30::S2G_LDS},
        {reinterpret_cast<const void *>(s2g_fused_t30_kernel<2, false, true>), t30::S2G_LDS}, {reinterpret_cast<const void *>(s2g_fused_t30_kernel<3, false, true>), t30::S2G_LDS},
        SPDY_K2(g2s_fused_t30_kernel, 0, t30::G2S_LDS), SPDY_K2(g2s_fused_t30_kernel, 1, t30::G2S_LDS),
        SPDY_K2(g2s_fused_t30_kernel, 2, t30::G2S_LDS), SPDY_K2(g2s_fused_t30_kernel, 3, t30::G2S_LDS),
        // the three-workgroups-per-tile form of small direct batches (every model-sized T30 batch, the captured step)
        {reinterpret_cast<const void *>(g2s_fused_t30_kernel<0, false, 3>), t30::G2S_LDS}, {reinterpret_cast<const void *>(g2s_fused_t30_kernel<1, false, 3>), t30::G2S_LDS},
        {reinterpret_cast<const void *>(g2s_fused_t30_kernel<2, false, 3>), t30::G2S_LDS}, {reinterpret_cast<const void *>(g2s_fused_t30_kernel<3, false, 3>), t30::G2S_LDS},
        SPDY_K2(g2s_fused_t63_kernel, 0, t63::LDS_BYTES), SPDY_K2(g2s_fused_t63_kernel, 1, t63::LDS_BYTES),
#undef SPDY_K2
        {reinterpret_cast<const void *>(s2g_fused_t63_kernel<false, false>), t63::LDS_BYTES}, {reinterpret_cast<const void *>(s2g_fused_t63_kernel<true, false>), t63::LDS_BYTES},
        {reinterpret_cast<const void *>(s2g_fused_t63_kernel<false, true>), t63::LDS_BYTES},
        {reinterpret_cast<const void *>(s2g_fused_t63_kernel<false, true, true>), t63::LDS_BYTES},
        // the two-workgroups-per-pair form of small direct batches (every model-sized T63 batch, the captured step)
        {reinterpret_cast<const void *>(g2s_fused_t63_kernel<0, false, true>), t63::LDS_BYTES}, {reinterpret_cast<const void *>(g2s_fused_t63_kernel<1, false, true>), t63::LDS_BYTES},
        {reinterpret_cast<const void *>(g2s_rows_t63_kernel<0>), (4 * 16 * t63::RS + t63::TW) * 8 + 4 * 68}, {reinterpret_cast<const void *>(g2s_rows_t63_kernel<1>), (4 * 16 * t63::RS + t63::TW) * 8 + 4 * 68},
        {reinterpret_cast<const void *>(wave_placement_kernel), t63::LDS_BYTES},
        {reinterpret_cast<const void *>(legendre_inv_kernel<3>), 104 * 1024}, {reinterpret_cast<const void *>(legendre_dir_kernel<3>), 104 * 1024}};   // T63: 73,856 / 98,432 B
    for (auto &b : big) {
        hipError_t e = hipFuncSetAttribute(b.fn, hipFuncAttributeMaxDynamicSharedMemorySize, b.bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_legendre_inv(const DevPlan &p, int nb, const double *spec, double *four, hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    const dim3 grid((nb + LEG_BT - 1) / LEG_BT, (p.mx + LEG_MG - 1) / LEG_MG);
    const size_t lds = legendre_inv_lds(p);   // B image / output lines
    if (p.jt == 2) hipLaunchKernelGGL(legendre_inv_kernel<2>, grid, dim3(LEG_NT), lds, s, p, nb, spec, four);
    else if (p.jt == 3) hipLaunchKernelGGL(legendre_inv_kernel<3>, grid, dim3(LEG_NT), lds, s, p, nb, spec, four);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_legendre_dir(const DevPlan &p, int nb, const double *four, double *spec, hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    if (p.il != (p.nt_dir == 3 ? 96 : 48)) return hipErrorInvalidValue;
    const int ntiles = ((nb + LEG_BT - 1) / LEG_BT) * ((p.mx + LEG_MG - 1) / LEG_MG);
    const size_t lds = legendre_dir_lds(p);   // Fourier image / output lines
    const dim3 grid(std::min(ntiles, (int)(160 * 1024 / lds) * fft_grid_limit(p) / 3));              // persistent: resident blocks only
    if (p.nt_dir == 1) hipLaunchKernelGGL(legendre_dir_kernel<1>, grid, dim3(LEG_NT), lds, s, p, nb, four, spec);
    else if (p.nt_dir == 3) hipLaunchKernelGGL(legendre_dir_kernel<3>, grid, dim3(LEG_NT), lds, s, p, nb, four, spec);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_fourier_inv(const DevPlan &p, int nb, const double *four, const int *d_kcos, int kcos_all,
                              double *grid, hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    const int nrows = nb * p.il, FFT_ROWS = fft_rows(p.ix);
    const int per_cu = 160 * 1024 / ((FFT_ROWS * (p.ix + 1) + 4 * 48 + FFT_ROWS) * 8);   // LDS-resident blocks per CU
    const int nblk = std::min((nrows + FFT_ROWS - 1) / FFT_ROWS, per_cu * fft_grid_limit(p) / 3);
    if (p.fs != (2 * p.mx + 15) / 16 * 16) return hipErrorInvalidValue;
    if (p.ix == 96 && p.mx == 31)
        hipLaunchKernelGGL((fourier_inv_kernel<96, 62>), dim3(nblk), dim3(FFT_ROWS * 2), 0, s, four, grid, p.cosgr,
                           d_kcos, kcos_all, nrows, p.il, p.fs);
    else if (p.ix == 192 && p.mx == 64)
        hipLaunchKernelGGL((fourier_inv_kernel<192, 128>), dim3(nblk), dim3(FFT_ROWS * 4), 0, s, four, grid, p.cosgr,
                           d_kcos, kcos_all, nrows, p.il, p.fs);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_fourier_dir(const DevPlan &p, int nb, const double *grid, const double *gscale, double *four,
                              hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    const int nrows = nb * p.il, FFT_ROWS = fft_rows(p.ix);
    const int per_cu = 160 * 1024 / ((FFT_ROWS * (p.ix + 1) + 4 * 48 + FFT_ROWS) * 8);   // LDS-resident blocks per CU
    const int nblk = std::min((nrows + FFT_ROWS - 1) / FFT_ROWS, per_cu * fft_grid_limit(p) / 3);
    if (p.fs != (2 * p.mx + 15) / 16 * 16) return hipErrorInvalidValue;
    if (p.ix == 96 && p.mx == 31)
        hipLaunchKernelGGL((fourier_dir_kernel<96, 62>), dim3(nblk), dim3(FFT_ROWS * 2), 0, s, grid, gscale, four,
                           nrows, p.il, p.fs);
    else if (p.ix == 192 && p.mx == 64)
        hipLaunchKernelGGL((fourier_dir_kernel<192, 128>), dim3(nblk), dim3(FFT_ROWS * 4), 0, s, grid, gscale, four,
                           nrows, p.il, p.fs);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

static inline dim3 blocks_for(long total) { return dim3((unsigned)((total + 255) / 256)); }

hipError_t launch_scale_op(const DevPlan &p, int op, int nb, const double *in, double *out, hipStream_t s)
{
    const long total = (long)nb * p.mx * p.nx;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(scale_op_kernel, blocks_for(total), dim3(256), 0, s, p, op, total, in, out);
    return hipGetLastError();
}

hipError_t launch_grad(const DevPlan &p, int nb, const double *psi, double *psdx, double *psdy, hipStream_t s)
{
    const long total = (long)nb * p.mx * p.nx;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(grad_kernel, blocks_for(total), dim3(256), 0, s, p, total, psi, psdx, psdy);
    return hipGetLastError();
}

hipError_t launch_vds(const DevPlan &p, int nb, const double *u, const double *v, double *vor, double *div, hipStream_t s)
{
    const long total = (long)nb * p.mx * p.nx;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(vds_kernel, blocks_for(total), dim3(256), 0, s, p, total, u, v, vor, div);
    return hipGetLastError();
}

hipError_t launch_uvspec(const DevPlan &p, int nb, const double *vor, const double *div, double *u, double *v, hipStream_t s)
{
    const long total = (long)nb * p.mx * p.nx;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(uvspec_kernel, blocks_for(total), dim3(256), 0, s, p, total, vor, div, u, v);
    return hipGetLastError();
}

hipError_t launch_uvspec_grad(const DevPlan &p, int nuv, const double *vor, const double *div, double *u, double *v, int ngr,
                              const double *psi, double *psdx, double *psdy, hipStream_t s)
{
    const long tuv = (long)nuv * p.mx * p.nx, tgr = (long)ngr * p.mx * p.nx;
    if (tuv <= 0 || tgr <= 0) return hipErrorInvalidValue;
    dim3 grd = blocks_for(tuv > tgr ? tuv : tgr);
    grd.y = 2;
    hipLaunchKernelGGL(uvspec_grad_kernel, grd, dim3(256), 0, s, p, tuv, vor, div, u, v, tgr, psi, psdx, psdy);
    return hipGetLastError();
}

// blockIdx.y selects the operation (the seven do_horizontal_diffusion calls of a time step in one launch)
__global__ void hdiff_multi_kernel(int sz, HdiffOps ops)
{
    const int op = blockIdx.y;
    const long total = (long)ops.nlev[op] * sz, i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i % sz);
    st(ops.out[op], i, ops.dmp1[op][e] * (ld(ops.fdt[op], i) - ops.dmp[op][e] * ld(ops.field[op], i)));
}
hipError_t launch_hdiff_multi(const DevPlan &p, const HdiffOps &ops, hipStream_t s)
{
    int maxlev = 0;
    for (int i = 0; i < ops.nops; ++i) maxlev = std::max(maxlev, ops.nlev[i]);
    const long total = (long)maxlev * p.mx * p.nx;
    if (ops.nops <= 0 || total <= 0) return hipSuccess;
    hipLaunchKernelGGL(hdiff_multi_kernel, dim3(blocks_for(total).x, ops.nops), dim3(256), 0, s, p.mx * p.nx, ops);
    return hipGetLastError();
}
hipError_t launch_hdiff(const DevPlan &p, int nlev, const double *field, const double *fdt, const double *dmp,
                        const double *dmp1, double *out, hipStream_t s)
{
    const long total = (long)nlev * p.mx * p.nx;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(hdiff_kernel, blocks_for(total), dim3(256), 0, s, p.mx * p.nx, total, field, fdt, dmp, dmp1, out);
    return hipGetLastError();
}

// Which SIMD the dispatcher put each of a 512-thread workgroup's eight waves on (HW_ID bits 5:4), launched with the fused T63
// kernels' LDS footprint, i.e. one workgroup per CU as they run.  The T63 kernels place their Legendre and FFT waves by SIMD
// on the assumption that hardware waves w and w + 4 share one (round-robin placement): spdy_wave_placement checks it.
__global__ __launch_bounds__(512) void wave_placement_kernel(int *out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const unsigned simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);      // hwreg(HW_REG_HW_ID, 4, 2) = SIMD_ID
    if ((threadIdx.x & 63) == 0) {
        lds[threadIdx.x >> 6] = 0.0;                                                // (the allocation is really made)
        out[8 * blockIdx.x + (threadIdx.x >> 6)] = (int)simd;
    }
}
hipError_t launch_wave_placement(int *d_out, int nwg, hipStream_t s)
{
    // (its dynamic-LDS limit is raised per device in prepare_device_kernels)
    hipLaunchKernelGGL(wave_placement_kernel, dim3(nwg), dim3(512), t63::LDS_BYTES, s, d_out);
    return hipGetLastError();
}


}  // namespace spdy
