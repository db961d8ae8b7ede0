// Complex helpers shared by the elementwise spectral-space kernels.  Operation order follows the reference's
// Fortran expressions: real*complex scales both parts; "* (0.0, 1.0)" is the full complex multiply.
#pragma once
#include <hip/hip_runtime.h>

namespace spdy {

#ifndef UNROLL
#define UNROLL _Pragma("unroll")
#endif

struct cpx { double re, im; };
__device__ __forceinline__ cpx ld(const double *a, long i) { const double2 v = *reinterpret_cast<const double2 *>(a + 2 * i); return {v.x, v.y}; }
__device__ __forceinline__ void st(double *a, long i, cpx z) { *reinterpret_cast<double2 *>(a + 2 * i) = make_double2(z.re, z.im); }
// Write-through stores (sc0 sc1).  A kernel's ordinary stores stay dirty in the eight L2s until the end-of-kernel release
// flushes them -- a serial tail of the launch that grows with the bytes written; written through they leave
// as they are produced.  Worth it for outputs of several MB per launch; small outputs are better left to the write-back
// cache.
__device__ __forceinline__ void st1_wt(double *p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ cpx operator*(double r, cpx z) { return {r * z.re, r * z.im}; }
__device__ __forceinline__ cpx operator+(cpx a, cpx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cpx operator-(cpx a, cpx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cpx operator-(cpx a) { return {-a.re, -a.im}; }
__device__ __forceinline__ cpx times_i(cpx a) { return {a.re * 0.0 - a.im, a.re + a.im * 0.0}; }   // * (0,1)

}  // namespace spdy
