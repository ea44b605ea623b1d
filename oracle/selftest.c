/* selftest.c -- sanitizer driver for the CPU oracle (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).
 * Built by tests/test_oracle_sanitizers.py with -fsanitize=address,undefined together with the oracle sources:
 * one random reset, one actuated control interval and every getter of the 2D and the 3D oracle on small grids,
 * so that out-of-bounds stencil reads, misaligned accesses or leaks in the restatement surface here (GPU
 * sanitizers are not available on the pool, so the checker at least is checked this way). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "rbc_oracle.h"

int main(void)
{
    int fail = 0;
    {
        rbco_config c = {96, 32, 6.283185307179586, 2.0, 1e4, 0.7, 1.0, 1.0, 12, 0.75, 0.03, 0.07, 0.01, 48, 8};
        rbco_sim *s = rbco_create(&c);
        if (!s) return 2;
        float act[12];
        for (int a = 0; a < 12; ++a) act[a] = (float)sin(1.0 + a);
        rbco_reset_random(s, 7);
        fail |= !rbco_step(s, act);
        size_t n = (size_t)c.nx * c.nz;
        double *b = malloc(n * sizeof *b), *u = malloc(n * sizeof *u), *w = malloc((n + c.nx) * sizeof *w), *st = malloc(5 * n * sizeof *st);
        float *st32 = malloc(5 * n * sizeof *st32), *ob32 = malloc(5 * (size_t)c.obs_nx * c.obs_nz * sizeof *ob32);
        double *tb = malloc(c.nx * sizeof *tb);
        rbco_get_fields(s, b, u, w); rbco_get_state(s, st, 5); rbco_get_state_f32(s, st32, 5); rbco_get_obs_f32(s, ob32, 5);
        rbco_get_tendencies(s, b, u, w); rbco_bottom_profile(s, tb); rbco_projected_rate(s, u, w, b);
        double t; int64_t step;
        rbco_get_info(s, &t, &step);
        fail |= !(rbco_max_divergence(s) < 1e-12) || !(rbco_nusselt(s, 1) == rbco_nusselt(s, 1)) || !(rbco_kinetic_energy(s) >= 0) || step != 2;
        rbco_reset_from_arrays(s, st, st + n, w);              /* b, u of the state dump + w */
        free(b); free(u); free(w); free(st); free(st32); free(ob32); free(tb);
        rbco_destroy(s);
    }
    {
        rbco3_config c = {16, 8, 8, 12.566370614359172, 6.283185307179586, 2.0, 3e3, 0.7, 1.0, 1.0, 4, 0.9, 0.01, 0.02, 0.05};
        rbco3_sim *s = rbco3_create(&c);
        if (!s) return 3;
        float act[16];
        for (int a = 0; a < 16; ++a) act[a] = (float)cos(0.5 + a);
        rbco3_reset_random(s, 9);
        fail |= !rbco3_step(s, act);
        size_t n = (size_t)c.nx * c.ny * c.nz, pl = (size_t)c.nx * c.ny;
        double *b = malloc(n * sizeof *b), *u = malloc(n * sizeof *u), *v = malloc(n * sizeof *v), *w = malloc((n + pl) * sizeof *w);
        float *st32 = malloc(4 * n * sizeof *st32);
        rbco3_get_fields(s, b, u, v, w); rbco3_get_state_f32(s, st32);
        double t; int64_t step;
        rbco3_get_info(s, &t, &step);
        fail |= !(rbco3_max_divergence(s) < 1e-11) || !(rbco3_nusselt(s) == rbco3_nusselt(s)) || step != 2;
        rbco3_reset_from_arrays(s, b, u, v, w);
        rbco3_get_tendencies(s, u, v, w, b);
        free(b); free(u); free(v); free(w); free(st32);
        rbco3_destroy(s);
    }
    printf(fail ? "selftest: FAILED\n" : "selftest: ok\n");
    return fail;
}
