// GPU box: back-to-back latency of model-shaped launches from a C host (no Python call overhead).
// (An experimental fork/join of independent calls over internal streams doubled this time -- event record/wait
// costs more than the overlap gains -- and was dropped.)
//   hipcc -O2 tools/launch_latency.cpp -Iinclude -Lspeedy.f90_amd -lspdy -Wl,-rpath,$PWD/speedy.f90_amd -o /tmp/cc && /tmp/cc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include "spdy.h"
#define CK(x) do { int rc_ = (x); if (rc_) { printf("error %d at line %d: %s\n", rc_, __LINE__, spdy_last_error()); return 1; } } while (0)
int main()
{
    spdy_plan *p;
    CK(spdy_plan_create(30, 96, 24, 8, 128, 0, &p));
    const size_t G = 96 * 48, S = 2 * 31 * 32;
    double *ug, *vg, *vor, *dv, *g2, *s2, *sp, *gr;
    hipMalloc(&ug, 24 * G * 8); hipMalloc(&vg, 24 * G * 8); hipMalloc(&vor, 24 * S * 8); hipMalloc(&dv, 24 * S * 8);
    hipMalloc(&g2, 25 * G * 8); hipMalloc(&s2, 25 * S * 8); hipMalloc(&sp, 91 * S * 8); hipMalloc(&gr, 91 * G * 8);
    hipMemset(ug, 0, 24 * G * 8); hipMemset(vg, 0, 24 * G * 8); hipMemset(g2, 0, 25 * G * 8); hipMemset(sp, 0, 91 * S * 8);
    auto run = [&](bool conc, int n) {
        CK(spdy_plan_synchronize(p));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {
            CK(spdy_vdspec_dev(p, 24, ug, vg, vor, dv, 2));
            CK(spdy_grid_to_spec_dev(p, 25, g2, s2));
            CK(spdy_spec_to_grid_dev(p, 91, sp, nullptr, 1, gr));
        }
        CK(spdy_plan_synchronize(p));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        printf("%7.1f us per (vdspec 24 + g2s 25 + s2g 91), i.e. %.1f us per launch\n", us, us / 3);
        return 0;
    };
    for (int rep = 0; rep < 3; ++rep) if (run(false, 2000)) return 1;
    spdy_plan_destroy(p);
    return 0;
}
