// Compile-time work schedule of the fused T63 kernels, shared by the kernels (spdy_fused_t63.inc) and the host code
// that builds their MFMA A-operand images (spdy_api.hip).  Plain constexpr C++: usable on both sides.
#pragma once
namespace spdy {
namespace t63 {
constexpr int TRUNC = 63, MX = 64, NX = 65, IY = 48, IL = 96, IX = 192;
constexpr int CP = 8, NCH = IY / CP;                         // latitude pairs per chunk, chunks per field pair
constexpr int CROWS = 4 * CP;                                // LDS rows of a chunk: [field 2][half 2: lat, IL-1-lat][CP]
constexpr int RS = IX + 1, BUF = CROWS * RS;                 // row stride (odd), doubles per chunk buffer
constexpr int NLW = 4, NFW = 4, NTHR = 64 * (NLW + NFW);     // Legendre waves, FFT waves (two groups of two), 512 threads
constexpr int NBUF = 3;                                      // chunk buffers: FFT group A, FFT group B, Legendre
constexpr int NQ = MX / 4;                                   // 16 quads of zonal wavenumbers
constexpr int SPEC_C = MX * NX;
constexpr int TW = 192;                                      // ido = 48 radix-4 twiddles, per column: [0] = (1, 0) pairs, [1..3] = wa1..wa3
constexpr int LDS_BYTES = (NBUF * BUF + TW + IL) * 8 + 4 * 68; // 150,800 B: chunk buffers, twiddles, a per-latitude factor table, the zero-fill row table
constexpr int MAXS = 38;                                     // slots per Legendre wave (both directions)

// DIR = true: direct transform writes n <= trunc with m'+n <= trunc+1 (legendre.f90:142-154);
// DIR = false: inverse transform reads m'+n <= trunc+1, n up to nx-1 (legendre.f90:92-103)
constexpr int nmax(bool dir, int q) { return dir ? (TRUNC < TRUNC + 1 - 4 * q ? TRUNC : TRUNC + 1 - 4 * q) : TRUNC + 1 - 4 * q; }
constexpr int ncount(bool dir, int q, int par) { return par == 0 ? nmax(dir, q) / 2 + 1 : (nmax(dir, q) + 1) / 2; }
constexpr int ngrp(bool dir, int q, int par) { return (ncount(dir, q, par) + 3) / 4; }
// Quads of wave w at its four positions: {2w, 15-2w, 2w+1, 14-2w} -- equal work per wave (the quad indices add up to 30) and per
// half (positions {0,1} / {2,3}: 15 each), and the two quads of positions (0, 2) and of (3, 1) are NEIGHBOURS (an even quad and
// the next one), so that the direct kernel's write-out can store their coefficients as whole 128-byte lines (t63_dir_writeout).
constexpr int wave_quad(int w, int i) { return i == 0 ? 2 * w : i == 1 ? 15 - 2 * w : i == 2 ? 2 * w + 1 : 14 - 2 * w; }
// slot index of (quad position i, parity, n-group g) within wave w
constexpr int slot_of(bool dir, int w, int i, int par, int g)
{
    int s = 0;
    for (int ii = 0; ii < i; ++ii) s += ngrp(dir, wave_quad(w, ii), 0) + ngrp(dir, wave_quad(w, ii), 1);
    if (par) s += ngrp(dir, wave_quad(w, i), 0);
    return s + g;
}
constexpr int nslots(bool dir, int w) { return slot_of(dir, w, 4, 0, 0); }
// A-operand image: index (in 64-lane double2 fragments) of wave w's slot s for latitude chunk c.  T63_ALAYOUT 0: [w][s][c]
// (a chunk's fragments 6 KB apart), 1: [w][c][s] (a wave's stream for one chunk is one contiguous run)
#ifndef T63_ALAYOUT
#define T63_ALAYOUT 1
#endif
constexpr int afrag(int w, int s, int c) { return T63_ALAYOUT ? (w * NCH + c) * MAXS + s : (w * MAXS + s) * NCH + c; }
constexpr int AFRAG_SLOT_STRIDE = T63_ALAYOUT ? 1 : NCH;    // fragments between consecutive slots of one (wave, chunk)
static_assert(nslots(true, 0) <= MAXS && nslots(true, 1) <= MAXS && nslots(true, 2) <= MAXS && nslots(true, 3) <= MAXS, "direct slots");
static_assert(nslots(false, 0) <= MAXS && nslots(false, 1) <= MAXS && nslots(false, 2) <= MAXS && nslots(false, 3) <= MAXS, "inverse slots");
}  // namespace t63
}  // namespace spdy
