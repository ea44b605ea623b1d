/*
 * rbc_oracle.c -- CPU oracle (fp64, one env) for the 2D Rayleigh-Benard hot path.
 * TEST INFRASTRUCTURE ONLY -- see rbc_oracle.h for the rules and the parity status.
 *
 * Every function cites the reference file:line it follows.  Where the arithmetic lives in
 * the un-vendored dependency Oceananigans.jl v0.92.0 the citation names the reference call
 * site plus the Oceananigans source file whose published algorithm is restated ([OC] tag).
 *
 * Index conventions (0-based here; the reference is 1-based):
 *   cell centres i=0..nx-1 (x_c=(i+1/2)dx), k=0..nz-1 (z_c=(k+1/2)dz)
 *   u lives on x-faces  (x_f=i*dx, z_c)   -> array u[k][i],  k=0..nz-1
 *   w lives on z-faces  (x_c, z_f=k*dz)   -> array w[k][i],  k=0..nz   (k=0, nz are the walls)
 *   b, pHY', pNHS live on centres.
 * All arrays carry a 3-cell halo in x and z like the reference grid (halo=(3,3,3) default).
 */
#include "rbc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define HALO 3
#define MAX_HEATERS 64

struct rbco_sim {
    rbco_config c;
    int nx, nz, sx, rows;     /* sx = row stride, rows = nz+1+2*HALO */
    double dx, dz, nu, kappa;
    double *u, *w, *b, *phy, *pnhs;       /* haloed */
    double *gu, *gw, *gb;                 /* G^n   (haloed layout, interior used) */
    double *gu0, *gw0, *gb0;              /* G^-   */
    double *rhs, *phi;                    /* interior nz*nx scratch for the Poisson solve */
    double action[MAX_HEATERS];
    double *tb;                           /* bottom wall temperature per column */
    double time;                          /* api-level time (rbc_sim2D_api.jl:13,87) */
    int64_t step;                         /* api-level step (rbc_sim2D_api.jl:12,67,88) */
    int var[RBCO_VAR_COUNT];
    /* Poisson tables */
    double *lamx, *lamz;                  /* eigenvalues */
    double *cosx, *sinx;                  /* twiddles for the x-DFT */
    double *dctz;                         /* dense DCT-II basis (variant 1) */
    double *tri_inv, *tri_l;              /* per-mode factorisation of the z tridiagonal systems */
};

#define IDX(s, i, k) (((k) + HALO) * (s)->sx + ((i) + HALO))

/* ------------------------------------------------------------------------------------------ */
/* RNG: counter-based, so the HIP library can reproduce the same deviates (own design; the    */
/* reference uses Julia's Xoshiro stream via randn(), rbc_sim2D.jl:164-167, not reproducible) */
/* ------------------------------------------------------------------------------------------ */
static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

double rbco_normal(uint64_t seed, uint32_t field, uint32_t index)
{
    uint64_t ctr = ((uint64_t)field << 32) | (uint64_t)index;
    uint64_t h0 = splitmix64(seed ^ splitmix64(ctr));
    uint64_t r1 = splitmix64(h0 + 0x9E3779B97F4A7C15ull);
    uint64_t r2 = splitmix64(h0 + 2 * 0x9E3779B97F4A7C15ull);
    double u1 = (double)((r1 >> 11) + 1) * (1.0 / 9007199254740992.0); /* (0,1] */
    double u2 = (double)(r2 >> 11) * (1.0 / 9007199254740992.0);       /* [0,1) */
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

/* ------------------------------------------------------------------------------------------ */
/* A10: heater profile. restates collate_actions_colin, rbc_sim2D.jl:87-133                   */
/* ------------------------------------------------------------------------------------------ */
static double heater_profile(const rbco_sim *s, double x)
{
    const int n = s->c.heaters;
    const double ampl = s->c.heater_limit;
    const double dx = 0.03;                                  /* :91 */
    double mean = 0.0, dev = 0.0;
    for (int a = 0; a < n; ++a) mean += ampl * s->action[a];
    mean /= n;                                               /* :94 */
    for (int a = 0; a < n; ++a) {
        double d = fabs(ampl * s->action[a] - mean);
        if (d > dev) dev = d;
    }
    double K2 = dev / ampl;                                  /* :95 */
    if (!(K2 > 1.0)) K2 = 1.0;
    const double seg = s->c.lx / n;                          /* :97 */
    int xs = (int)floor(x / seg) + 1;                        /* :100 (1-based) */
    if (xs > n) xs = n;
    int im1 = (xs == 1) ? n : xs - 1;                        /* :102-106 */
    int ip1 = (xs == n) ? 1 : xs + 1;                        /* :110-114 */
    double T0 = 2 + (ampl * s->action[im1 - 1] - mean) / K2;
    double T1 = 2 + (ampl * s->action[xs - 1] - mean) / K2;  /* :108 */
    double T2 = 2 + (ampl * s->action[ip1 - 1] - mean) / K2;
    double xp = x - (xs - 1) * seg;                          /* :117 */
    if (xp < dx)                                             /* :120-122 */
        return T0 + ((T0 - T1) / (4 * dx * dx * dx)) * (xp - 2 * dx) * (xp + dx) * (xp + dx);
    else if (xp >= seg - dx)                                 /* :124-126 */
        return T1 + ((T1 - T2) / (4 * dx * dx * dx)) * (xp - seg - 2 * dx) * (xp - seg + dx) * (xp - seg + dx);
    return T1;                                               /* :130 */
}

void rbco_set_action(rbco_sim *s, const float *action)
{
    /* rbc_sim2D_api.jl:77 `global action = actuation`; float32 promoted to Float64 in
       `ampl .* action` (rbc_sim2D.jl:93) */
    for (int a = 0; a < s->c.heaters; ++a) s->action[a] = action ? (double)action[a] : 0.0;
    for (int i = 0; i < s->nx; ++i) s->tb[i] = heater_profile(s, (i + 0.5) * s->dx); /* bottom_T at x_c, :135-138 */
}

void rbco_bottom_profile(const rbco_sim *s, double *tb) { memcpy(tb, s->tb, sizeof(double) * s->nx); }

/* ------------------------------------------------------------------------------------------ */
/* halo fill. [OC] BoundaryConditions/fill_halo_regions_{periodic,value_gradient,open}.jl     */
/* reference call sites: ValueBoundaryCondition rbc_sim2D.jl:141-146, Periodic/Bounded :77-81 */
/* ------------------------------------------------------------------------------------------ */
static void periodic_x(const rbco_sim *s, double *a)
{
    for (int k = -HALO; k < s->nz + 1 + HALO; ++k)
        for (int h = 1; h <= HALO; ++h) {
            a[IDX(s, -h, k)] = a[IDX(s, s->nx - h, k)];
            a[IDX(s, s->nx - 1 + h, k)] = a[IDX(s, h - 1, k)];
        }
}

static void fill_halo_u(const rbco_sim *s, double *u)
{
    const double dz = s->dz;
    for (int i = 0; i < s->nx; ++i) {
        /* Value BC 0 (no slip) at both walls: c_halo = c_I + grad*(-+dz), grad = (c_I - bc)/(dz/2) */
        double c1 = u[IDX(s, i, 0)], cN = u[IDX(s, i, s->nz - 1)];
        u[IDX(s, i, -1)] = c1 + ((c1 - 0.0) / (dz / 2)) * (-dz);
        u[IDX(s, i, s->nz)] = cN + ((0.0 - cN) / (dz / 2)) * dz;
    }
    periodic_x(s, u);
}

static void fill_halo_b(const rbco_sim *s, double *b)
{
    const double dz = s->dz;
    for (int i = 0; i < s->nx; ++i) {
        double c1 = b[IDX(s, i, 0)], cN = b[IDX(s, i, s->nz - 1)];
        b[IDX(s, i, -1)] = c1 + ((c1 - s->tb[i]) / (dz / 2)) * (-dz);       /* bottom = bottom_T(x) */
        b[IDX(s, i, s->nz)] = cN + ((s->c.min_b - cN) / (dz / 2)) * dz;     /* top = min_b          */
    }
    periodic_x(s, b);
}

static void fill_halo_w(const rbco_sim *s, double *w)
{
    for (int i = 0; i < s->nx; ++i) { /* impenetrable walls: boundary faces are set, not halos */
        w[IDX(s, i, 0)] = 0.0;
        w[IDX(s, i, s->nz)] = 0.0;
    }
    periodic_x(s, w);
}

static void fill_halo_p(const rbco_sim *s, double *p)
{
    for (int i = 0; i < s->nx; ++i) { /* default no-flux BC for a centred field on a Bounded axis */
        p[IDX(s, i, -1)] = p[IDX(s, i, 0)];
        p[IDX(s, i, s->nz)] = p[IDX(s, i, s->nz - 1)];
    }
    periodic_x(s, p);
}

/* ------------------------------------------------------------------------------------------ */
/* A6: reconstruction stencils. [OC] Advection/{upwind_biased,centered}_reconstruction.jl and */
/* topologically_conditional_interpolation.jl; reference call site rbc_sim2D.jl:151           */
/* (advection=UpwindBiasedFifthOrder()).  `p` points at the value with index `idx` of the     */
/* 1-D line, `st` is the stride.  idx1 is the 1-BASED target index as in the reference.       */
/* ------------------------------------------------------------------------------------------ */
/* periodic x: always the high-order stencil */
static inline double left5(const double *p, int st)   /* p -> psi[i] : psi[i-3..i+1] centre->face, or p->psi[i+1]: face->centre */
{ return (2 * p[-3 * st] - 13 * p[-2 * st] + 47 * p[-st] + 27 * p[0] - 3 * p[st]) / 60; }
static inline double right5(const double *p, int st)  /* psi[i-2..i+2] */
{ return (-3 * p[-2 * st] + 27 * p[-st] + 47 * p[0] - 13 * p[st] + 2 * p[2 * st]) / 60; }
static inline double left3(const double *p, int st)   /* psi[i-2..i] */
{ return (-p[-2 * st] + 5 * p[-st] + 2 * p[0]) / 6; }
static inline double right3(const double *p, int st)  /* psi[i-1..i+1] */
{ return (2 * p[-st] + 5 * p[0] - p[st]) / 6; }
static inline double sym4(const double *p, int st)    /* psi[i-2..i+1] */
{ return (-p[-2 * st] + 7 * p[-st] + 7 * p[0] - p[st]) / 12; }
static inline double sym2(const double *p, int st)    /* psi[i-1], psi[i] */
{ return (p[-st] + p[0]) / 2; }

/* buffer tests on a Bounded axis with N cells; k1 is 1-based; B = buffer size */
static inline int out_sym_f(const rbco_sim *s, int k1, int N, int B)
{ return s->var[RBCO_VAR_BOUNDS] == 2 ? (k1 > B && k1 < N + 1 - B) : (k1 >= B + 1 && k1 <= N + 1 - B); }
static inline int out_sym_c(const rbco_sim *s, int k1, int N, int B)
{ return s->var[RBCO_VAR_BOUNDS] == 2 ? (k1 > B && k1 < N + 1 - B) : (k1 >= B && k1 <= N + 1 - B); }
static inline int out_left_f(const rbco_sim *s, int k1, int N, int B)
{ if (s->var[RBCO_VAR_BOUNDS] == 0) return out_sym_f(s, k1, N, B); return s->var[RBCO_VAR_BOUNDS] == 2 ? (k1 > B && k1 < N + 1 - (B - 1)) : (k1 >= B + 1 && k1 <= N + 1 - (B - 1)); }
static inline int out_left_c(const rbco_sim *s, int k1, int N, int B)
{ if (s->var[RBCO_VAR_BOUNDS] == 0) return out_sym_c(s, k1, N, B); return s->var[RBCO_VAR_BOUNDS] == 2 ? (k1 > B - 1 && k1 < N + 1 - (B - 1)) : (k1 >= B && k1 <= N + 1 - (B - 1)); }
static inline int out_right_f(const rbco_sim *s, int k1, int N, int B)
{ if (s->var[RBCO_VAR_BOUNDS] == 0) return out_sym_f(s, k1, N, B); return s->var[RBCO_VAR_BOUNDS] == 2 ? (k1 > B - 1 && k1 < N + 1 - B) : (k1 >= B && k1 <= N + 1 - B); }
static inline int out_right_c(const rbco_sim *s, int k1, int N, int B)
{ if (s->var[RBCO_VAR_BOUNDS] == 0) return out_sym_c(s, k1, N, B); return s->var[RBCO_VAR_BOUNDS] == 2 ? (k1 > B - 2 && k1 < N + 1 - B) : (k1 >= B - 1 && k1 <= N + 1 - B); }

/* centre field -> z-face k (0-based face k sits between centres k-1 and k). p -> psi[k] */
static double zf_left(const rbco_sim *s, const double *p, int k)
{
    const int st = s->sx, k1 = k + 1, N = s->nz;
    if (out_left_f(s, k1, N, 3)) return left5(p, st);
    if (out_left_f(s, k1, N, 2)) return left3(p, st);
    return p[-st];
}
static double zf_right(const rbco_sim *s, const double *p, int k)
{
    const int st = s->sx, k1 = k + 1, N = s->nz;
    if (out_right_f(s, k1, N, 3)) return right5(p, st);
    if (out_right_f(s, k1, N, 2)) return right3(p, st);
    return p[0];
}
static double zf_sym(const rbco_sim *s, const double *p, int k)
{
    const int st = s->sx, k1 = k + 1, N = s->nz;
    if (s->var[RBCO_VAR_SYMLEVEL] == 0) {
        /* the test uses the UPWIND scheme's buffer (3); its buffer schemes' advecting
           velocity schemes are Centered(2) */
        if (out_sym_f(s, k1, N, 3)) return sym4(p, st);
        return sym2(p, st);
    }
    if (out_sym_f(s, k1, N, 2)) return sym4(p, st);
    return sym2(p, st);
}
/* face field -> z-centre k (between faces k and k+1). p -> psi[k+1] (so that the same
   stencil helpers apply: left5 reads p[-3..+1] = faces k-2..k+2) */
static double zc_left(const rbco_sim *s, const double *p, int k)
{
    const int st = s->sx, k1 = k + 1, N = s->nz;
    if (out_left_c(s, k1, N, 3)) return left5(p, st);
    if (out_left_c(s, k1, N, 2)) return left3(p, st);
    return p[-st];
}
static double zc_right(const rbco_sim *s, const double *p, int k)
{
    const int st = s->sx, k1 = k + 1, N = s->nz;
    if (out_right_c(s, k1, N, 3)) return right5(p, st);
    if (out_right_c(s, k1, N, 2)) return right3(p, st);
    return p[0];
}
static double zc_sym(const rbco_sim *s, const double *p, int k)
{
    const int st = s->sx, k1 = k + 1, N = s->nz;
    if (s->var[RBCO_VAR_SYMLEVEL] == 0) {
        if (out_sym_c(s, k1, N, 3)) return sym4(p, st);
        return sym2(p, st);
    }
    if (out_sym_c(s, k1, N, 2)) return sym4(p, st);
    return sym2(p, st);
}

/* [OC] Advection/upwind_biased_reconstruction.jl: upwind_biased_product */
static inline double upwind(double ut, double psiL, double psiR)
{ return ((ut + fabs(ut)) * psiL + (ut - fabs(ut)) * psiR) / 2; }

/* ------------------------------------------------------------------------------------------ */
/* advective fluxes. [OC] Advection/{momentum,tracer}_advection_operators.jl                  */
/* Areas on the (x,Flat,z) grid: Ax = dz, Az = dx, V = dx*dz.                                 */
/* ------------------------------------------------------------------------------------------ */
static double flux_uu(const rbco_sim *s, int i, int k)   /* at centre (i,k) */
{
    const double *u = s->u + IDX(s, i + 1, k);           /* p -> u[i+1] (face->centre helpers) */
    double ut = s->dz * sym4(u, 1);
    return upwind(ut, left5(u, 1), right5(u, 1));
}
static double flux_wu(const rbco_sim *s, int i, int k)   /* at (x-face i, z-face k) */
{
    double wt = s->dx * sym4(s->w + IDX(s, i, k), 1);    /* centre->face in x on w */
    const double *u = s->u + IDX(s, i, k);
    return upwind(wt, zf_left(s, u, k), zf_right(s, u, k));
}
static double flux_uw(const rbco_sim *s, int i, int k)   /* at (x-face i, z-face k) */
{
    double ut = s->dz * zf_sym(s, s->u + IDX(s, i, k), k);
    const double *w = s->w + IDX(s, i, k);
    return upwind(ut, left5(w, 1), right5(w, 1));
}
static double flux_ww(const rbco_sim *s, int i, int k)   /* at centre (i,k) */
{
    const double *w = s->w + IDX(s, i, k + 1);
    double wt = s->dx * zc_sym(s, w, k);
    return upwind(wt, zc_left(s, w, k), zc_right(s, w, k));
}
static double flux_bx(const rbco_sim *s, int i, int k)   /* at x-face i */
{
    const double *b = s->b + IDX(s, i, k);
    return s->dz * upwind(s->u[IDX(s, i, k)], left5(b, 1), right5(b, 1));
}
static double flux_bz(const rbco_sim *s, int i, int k)   /* at z-face k */
{
    const double *b = s->b + IDX(s, i, k);
    return s->dx * upwind(s->w[IDX(s, i, k)], zf_left(s, b, k), zf_right(s, b, k));
}

/* ------------------------------------------------------------------------------------------ */
/* A8: hydrostatic pressure anomaly. [OC] Models/NonhydrostaticModels/                        */
/* update_hydrostatic_pressure.jl; output channel 4 at rbc_sim2D_api.jl:114                   */
/* ------------------------------------------------------------------------------------------ */
static void update_hydrostatic_pressure(rbco_sim *s)
{
    const int nz = s->nz;
    for (int i = 0; i < s->nx; ++i) {
        /* b at face k = mean of the neighbouring centres (top one uses the halo cell) */
        s->phy[IDX(s, i, nz - 1)] = -(0.5 * (s->b[IDX(s, i, nz - 1)] + s->b[IDX(s, i, nz)])) * s->dz;
        for (int k = nz - 2; k >= 0; --k)
            s->phy[IDX(s, i, k)] = s->phy[IDX(s, i, k + 1)] - (0.5 * (s->b[IDX(s, i, k)] + s->b[IDX(s, i, k + 1)])) * s->dz;
    }
    periodic_x(s, s->phy);
}

/* ------------------------------------------------------------------------------------------ */
/* tendencies G^n. [OC] Models/NonhydrostaticModels/nonhydrostatic_tendency_kernel_functions  */
/* .jl + TurbulenceClosures (ScalarDiffusivity, isotropic, explicit). call site               */
/* rbc_sim2D.jl:150-158                                                                       */
/* ------------------------------------------------------------------------------------------ */
static void compute_tendencies(rbco_sim *s)
{
    const int nx = s->nx, nz = s->nz;
    const double dx = s->dx, dz = s->dz, V = dx * dz, nu = s->nu, ka = s->kappa;
    const double *u = s->u, *w = s->w, *b = s->b;
    for (int k = 0; k < nz; ++k)
        for (int i = 0; i < nx; ++i) {
            const int c = IDX(s, i, k);
            /* ---- u ---- */
            double adv_u = (flux_uu(s, i, k) - flux_uu(s, i - 1, k) + flux_wu(s, i, k + 1) - flux_wu(s, i, k)) / V;
            double visc_u;
            if (s->var[RBCO_VAR_VISCOUS] == 0) {
                /* -d_j tau_1j, tau_11 = -2 nu dxu (ccc), tau_13 = -nu (dzu + dxw) (fcf) */
                double t11_i = -2 * nu * (u[c + 1] - u[c]) / dx;
                double t11_im = -2 * nu * (u[c] - u[c - 1]) / dx;
                double t13_kp = -nu * ((u[c + s->sx] - u[c]) / dz + (w[c + s->sx] - w[c + s->sx - 1]) / dx);
                double t13_k = -nu * ((u[c] - u[c - s->sx]) / dz + (w[c] - w[c - 1]) / dx);
                visc_u = -((dz * t11_i - dz * t11_im) + (dx * t13_kp - dx * t13_k)) / V;
            } else {
                visc_u = nu * ((u[c + 1] - 2 * u[c] + u[c - 1]) / (dx * dx) + (u[c + s->sx] - 2 * u[c] + u[c - s->sx]) / (dz * dz));
            }
            double gu = -adv_u + visc_u;
            if (s->var[RBCO_VAR_BUOYANCY] == 0) gu -= (s->phy[c] - s->phy[c - 1]) / dx;
            s->gu[c] = gu;
            /* ---- b ---- */
            double adv_b = (flux_bx(s, i + 1, k) - flux_bx(s, i, k) + flux_bz(s, i, k + 1) - flux_bz(s, i, k)) / V;
            double qx_ip = -ka * (b[c + 1] - b[c]) / dx, qx_i = -ka * (b[c] - b[c - 1]) / dx;
            double qz_kp = -ka * (b[c + s->sx] - b[c]) / dz, qz_k = -ka * (b[c] - b[c - s->sx]) / dz;
            double diff_b = -((dz * qx_ip - dz * qx_i) + (dx * qz_kp - dx * qz_k)) / V;
            s->gb[c] = -adv_b + diff_b;
            /* ---- w (face k; the wall face k=0 never evolves) ---- */
            if (k == 0) { s->gw[c] = 0.0; continue; }
            double adv_w = (flux_uw(s, i + 1, k) - flux_uw(s, i, k) + flux_ww(s, i, k) - flux_ww(s, i, k - 1)) / V;
            double visc_w;
            if (s->var[RBCO_VAR_VISCOUS] == 0) {
                double t31_ip = -nu * ((u[c + 1] - u[c + 1 - s->sx]) / dz + (w[c + 1] - w[c]) / dx);
                double t31_i = -nu * ((u[c] - u[c - s->sx]) / dz + (w[c] - w[c - 1]) / dx);
                double t33_k = -2 * nu * (w[c + s->sx] - w[c]) / dz;
                double t33_km = -2 * nu * (w[c] - w[c - s->sx]) / dz;
                visc_w = -((dz * t31_ip - dz * t31_i) + (dx * t33_k - dx * t33_km)) / V;
            } else {
                visc_w = nu * ((w[c + 1] - 2 * w[c] + w[c - 1]) / (dx * dx) + (w[c + s->sx] - 2 * w[c] + w[c - s->sx]) / (dz * dz));
            }
            double gw = -adv_w + visc_w;
            if (s->var[RBCO_VAR_BUOYANCY] == 1) gw += 0.5 * (b[c] + b[c - s->sx]);
            s->gw[c] = gw;
        }
}

/* [OC] Models/NonhydrostaticModels/update_nonhydrostatic_model_state.jl: update_state! */
void rbco_update_state(rbco_sim *s)
{
    fill_halo_u(s, s->u);
    fill_halo_w(s, s->w);
    fill_halo_b(s, s->b);
    update_hydrostatic_pressure(s);
    compute_tendencies(s);
}

/* ------------------------------------------------------------------------------------------ */
/* A9: pressure solve. [OC] Solvers/fft_based_poisson_solver.jl + Models/NonhydrostaticModels */
/* /{solve_for_pressure,pressure_correction}.jl.  Solves the 5-point  lap(phi) = rhs  exactly */
/* with periodic x / homogeneous Neumann z; the mean of phi is 0 (zero mode set to 0).        */
/* ------------------------------------------------------------------------------------------ */
static void poisson_tables(rbco_sim *s)
{
    const int nx = s->nx, nz = s->nz;
    const double pi = 3.14159265358979323846;
    s->lamx = malloc(sizeof(double) * nx);
    s->lamz = malloc(sizeof(double) * nz);
    s->cosx = malloc(sizeof(double) * nx);
    s->sinx = malloc(sizeof(double) * nx);
    s->dctz = malloc(sizeof(double) * nz * nz);
    s->tri_inv = malloc(sizeof(double) * nx * nz);
    s->tri_l = malloc(sizeof(double) * nx * nz);
    for (int m = 0; m < nx; ++m) { /* poisson_eigenvalues(N, L, dim, ::Periodic) */
        double t = 2 * sin(m * pi / nx) / s->dx;
        s->lamx[m] = t * t;
        s->cosx[m] = cos(2 * pi * m / nx);
        s->sinx[m] = sin(2 * pi * m / nx);
    }
    for (int q = 0; q < nz; ++q) { /* ::Bounded */
        double t = 2 * sin(q * pi / (2 * nz)) / s->dz;
        s->lamz[q] = t * t;
        for (int k = 0; k < nz; ++k) s->dctz[q * nz + k] = cos(pi * q * (k + 0.5) / nz);
    }
    /* LU of the z operator for every x-mode m:  (phi[k-1] - 2phi[k] + phi[k+1])/dz^2 - lamx*phi = r
       with mirror (Neumann) ends. diag d_k = -(2 or 1)/dz^2 - lamx, off-diagonals 1/dz^2. */
    const double o = 1.0 / (s->dz * s->dz);
    for (int m = 0; m < nx; ++m) {
        double piv = 0.0;
        for (int k = 0; k < nz; ++k) {
            double d = -((k == 0 || k == nz - 1) ? 1.0 : 2.0) * o - s->lamx[m];
            if (m == 0 && k == nz - 1) d -= o; /* regularise the singular mean mode (mean removed afterwards) */
            double l = (k == 0) ? 0.0 : o / piv;
            piv = d - l * ((k == 0) ? 0.0 : o);
            s->tri_l[m * nz + k] = l;
            s->tri_inv[m * nz + k] = 1.0 / piv;
        }
    }
}

/* variant 0: real DFT in x (direct O(N^2) sums with exact twiddle table), Thomas in z */
static void poisson_fft_tridiag(rbco_sim *s)
{
    const int nx = s->nx, nz = s->nz, nh = nx / 2;
    const double o = 1.0 / (s->dz * s->dz);
    double *re = malloc(sizeof(double) * (nh + 1) * nz), *im = malloc(sizeof(double) * (nh + 1) * nz);
    for (int k = 0; k < nz; ++k)
        for (int m = 0; m <= nh; ++m) {
            double a = 0, bb = 0;
            for (int i = 0; i < nx; ++i) {
                int t = (int)(((long)m * i) % nx);
                a += s->rhs[k * nx + i] * s->cosx[t];
                bb -= s->rhs[k * nx + i] * s->sinx[t];
            }
            re[m * nz + k] = a;
            im[m * nz + k] = bb;
        }
    for (int m = 0; m <= nh; ++m)
        for (int part = 0; part < 2; ++part) {
            double *r = (part ? im : re) + m * nz;
            const double *l = s->tri_l + m * nz, *inv = s->tri_inv + m * nz;
            for (int k = 1; k < nz; ++k) r[k] -= l[k] * r[k - 1];
            r[nz - 1] *= inv[nz - 1];
            for (int k = nz - 2; k >= 0; --k) r[k] = (r[k] - o * r[k + 1]) * inv[k];
        }
    { /* zero the mean of the m=0 mode */
        double mean = 0;
        for (int k = 0; k < nz; ++k) mean += re[k];
        mean /= nz;
        for (int k = 0; k < nz; ++k) re[k] -= mean;
    }
    for (int k = 0; k < nz; ++k)
        for (int i = 0; i < nx; ++i) {
            double a = re[k];
            const int even = (nx % 2 == 0);      /* odd nx: modes 1..(nx-1)/2 each with their conjugate, no Nyquist mode */
            for (int m = 1; m < (even ? nh : nh + 1); ++m) {
                int t = (int)(((long)m * i) % nx);
                a += 2 * (re[m * nz + k] * s->cosx[t] - im[m * nz + k] * s->sinx[t]);
            }
            if (even) a += re[nh * nz + k] * ((i & 1) ? -1.0 : 1.0);
            s->phi[k * nx + i] = a / nx;
        }
    free(re);
    free(im);
}

/* variant 1: the reference solver's formulation -- eigenfunction expansion in both
   directions (DFT in x, DCT-II in z), phi_hat = -rhs_hat/(lamx+lamz), zero mode = 0 */
static void poisson_eigen(rbco_sim *s)
{
    const int nx = s->nx, nz = s->nz;
    double *re = calloc((size_t)nx * nz, sizeof(double)), *im = calloc((size_t)nx * nz, sizeof(double));
    double *tr = malloc(sizeof(double) * nx * nz);
    /* DCT-II in z */
    for (int q = 0; q < nz; ++q)
        for (int i = 0; i < nx; ++i) {
            double a = 0;
            for (int k = 0; k < nz; ++k) a += s->dctz[q * nz + k] * s->rhs[k * nx + i];
            tr[q * nx + i] = a;
        }
    for (int q = 0; q < nz; ++q)
        for (int m = 0; m < nx; ++m) {
            double a = 0, bb = 0;
            for (int i = 0; i < nx; ++i) {
                int t = (int)(((long)m * i) % nx);
                a += tr[q * nx + i] * s->cosx[t];
                bb -= tr[q * nx + i] * s->sinx[t];
            }
            double lam = s->lamx[m] + s->lamz[q];
            if (m == 0 && q == 0) { a = 0; bb = 0; lam = 1; }
            re[q * nx + m] = -a / lam;
            im[q * nx + m] = -bb / lam;
        }
    for (int q = 0; q < nz; ++q)
        for (int i = 0; i < nx; ++i) {
            double a = 0;
            for (int m = 0; m < nx; ++m) {
                int t = (int)(((long)m * i) % nx);
                a += re[q * nx + m] * s->cosx[t] - im[q * nx + m] * s->sinx[t];
            }
            tr[q * nx + i] = a / nx;
        }
    /* inverse of the unnormalised DCT-II: x_k = (1/N) X_0 + (2/N) sum_{q>=1} X_q cos(pi q (k+1/2)/N) */
    for (int k = 0; k < nz; ++k)
        for (int i = 0; i < nx; ++i) {
            double a = tr[i] / nz;
            for (int q = 1; q < nz; ++q) a += (2.0 / nz) * s->dctz[q * nz + k] * tr[q * nx + i];
            s->phi[k * nx + i] = a;
        }
    free(re);
    free(im);
    free(tr);
}

/* calculate_pressure_correction! + pressure_correct_velocities! for one stage of size dts */
static void pressure_project(rbco_sim *s, double dts)
{
    const int nx = s->nx, nz = s->nz;
    fill_halo_u(s, s->u);
    fill_halo_w(s, s->w);
    for (int k = 0; k < nz; ++k)
        for (int i = 0; i < nx; ++i) {
            const int c = IDX(s, i, k);
            /* div_ccc = (d_x(Ax u) + d_z(Az w))/V */
            double div = ((s->dz * s->u[c + 1] - s->dz * s->u[c]) + (s->dx * s->w[c + s->sx] - s->dx * s->w[c])) / (s->dx * s->dz);
            s->rhs[k * nx + i] = div / dts;
        }
    if (s->var[RBCO_VAR_POISSON] == 0) poisson_fft_tridiag(s); else poisson_eigen(s);
    for (int k = 0; k < nz; ++k)
        for (int i = 0; i < nx; ++i) s->pnhs[IDX(s, i, k)] = s->phi[k * nx + i];
    fill_halo_p(s, s->pnhs);
    for (int k = 0; k < nz; ++k)
        for (int i = 0; i < nx; ++i) {
            const int c = IDX(s, i, k);
            s->u[c] -= (s->pnhs[c] - s->pnhs[c - 1]) / s->dx * dts;
            s->w[c] -= (s->pnhs[c] - s->pnhs[c - s->sx]) / s->dz * dts;   /* k=0: halo mirror -> 0 */
        }
}

/* ------------------------------------------------------------------------------------------ */
/* A5: Le-Moin low-storage RK3. [OC] TimeSteppers/runge_kutta_3.jl; call site                 */
/* rbc_sim2D.jl:152 (timestepper=:RungeKutta3)                                                */
/* ------------------------------------------------------------------------------------------ */
static void rk3_stage(rbco_sim *s, double dt, double g, double z, int first)
{
    const int nx = s->nx, nz = s->nz;
    for (int k = 0; k < nz; ++k)
        for (int i = 0; i < nx; ++i) {
            const int c = IDX(s, i, k);
            if (first) {
                s->u[c] += dt * g * s->gu[c];
                s->w[c] += dt * g * s->gw[c];
                s->b[c] += dt * g * s->gb[c];
            } else {
                s->u[c] += dt * (g * s->gu[c] + z * s->gu0[c]);
                s->w[c] += dt * (g * s->gw[c] + z * s->gw0[c]);
                s->b[c] += dt * (g * s->gb[c] + z * s->gb0[c]);
            }
        }
}

static void store_tendencies(rbco_sim *s)
{
    size_t n = (size_t)s->rows * s->sx * sizeof(double);
    memcpy(s->gu0, s->gu, n);
    memcpy(s->gw0, s->gw, n);
    memcpy(s->gb0, s->gb, n);
}

void rbco_substep(rbco_sim *s, double dt)
{
    const double g1 = 8.0 / 15, g2 = 5.0 / 12, g3 = 3.0 / 4, z2 = -17.0 / 60, z3 = -5.0 / 12;
    rk3_stage(s, dt, g1, 0, 1);
    pressure_project(s, g1 * dt);
    store_tendencies(s);
    rbco_update_state(s);
    rk3_stage(s, dt, g2, z2, 0);
    pressure_project(s, (g2 + z2) * dt);
    store_tendencies(s);
    rbco_update_state(s);
    rk3_stage(s, dt, g3, z3, 0);
    pressure_project(s, (g3 + z3) * dt);
    rbco_update_state(s);
}

/* A13: step_contains_NaNs, rbc_sim2D.jl:223-228 */
static int contains_nan(const rbco_sim *s)
{
    for (int k = 0; k < s->nz; ++k)
        for (int i = 0; i < s->nx; ++i) {
            const int c = IDX(s, i, k);
            if (isnan(s->b[c]) || isnan(s->u[c]) || isnan(s->w[c])) return 1;
        }
    return 0;
}

/* A4: step_simulation, rbc_sim2D_api.jl:75-97.  run! advances the model clock from its
   current time to stop_time with aligned steps ([OC] Simulations/run.jl, simulation.jl:
   aligned_time_step = min(dt, stop_time - t)); run! re-initialises, i.e. update_state! with the
   NEW action before the first stage. */
int rbco_step(rbco_sim *s, const float *action)
{
    rbco_set_action(s, action);
    rbco_update_state(s);
    double t = 0.0;
    const double T = s->c.dt_control, dt0 = s->c.dt_solver;
    /* substeps: full dt while more than dt remains, final one clipped to land on T */
    int nfull = (int)floor(T / dt0 + 1e-9);
    for (int n = 0; n < nfull; ++n) { rbco_substep(s, dt0); }
    t = nfull * dt0;
    if (T - t > 1e-9 * dt0) rbco_substep(s, T - t);
    s->time += s->c.dt_control;   /* :87 */
    s->step += 1;                 /* :88 */
    return contains_nan(s) ? 0 : 1;
}

/* ------------------------------------------------------------------------------------------ */
/* create / reset                                                                             */
/* ------------------------------------------------------------------------------------------ */
rbco_sim *rbco_create(const rbco_config *cfg)
{
    if (cfg->heaters > MAX_HEATERS || cfg->heaters < 1) return NULL;
    rbco_sim *s = calloc(1, sizeof(*s));
    s->c = *cfg;
    s->nx = cfg->nx;
    s->nz = cfg->nz;
    s->sx = s->nx + 2 * HALO;
    s->rows = s->nz + 1 + 2 * HALO;
    s->dx = cfg->lx / cfg->nx;
    s->dz = cfg->lz / cfg->nz;
    s->nu = sqrt(cfg->pr / cfg->ra);          /* rbc_sim2D_api.jl:40 */
    s->kappa = 1 / sqrt(cfg->pr * cfg->ra);   /* :41 */
    size_t n = (size_t)s->rows * s->sx;
    double **all[] = {&s->u, &s->w, &s->b, &s->phy, &s->pnhs, &s->gu, &s->gw, &s->gb, &s->gu0, &s->gw0, &s->gb0};
    for (unsigned a = 0; a < sizeof(all) / sizeof(all[0]); ++a) *all[a] = calloc(n, sizeof(double));
    s->rhs = calloc((size_t)s->nx * s->nz, sizeof(double));
    s->phi = calloc((size_t)s->nx * s->nz, sizeof(double));
    s->tb = calloc(s->nx, sizeof(double));
    poisson_tables(s);
    s->step = 1;
    rbco_set_action(s, NULL);
    return s;
}

void rbco_destroy(rbco_sim *s)
{
    if (!s) return;
    double *all[] = {s->u, s->w, s->b, s->phy, s->pnhs, s->gu, s->gw, s->gb, s->gu0, s->gw0, s->gb0,
                     s->rhs, s->phi, s->tb, s->lamx, s->lamz, s->cosx, s->sinx, s->dctz, s->tri_inv, s->tri_l};
    for (unsigned a = 0; a < sizeof(all) / sizeof(all[0]); ++a) free(all[a]);
    free(s);
}

void rbco_set_variant(rbco_sim *s, int which, int value)
{
    if (which >= 0 && which < RBCO_VAR_COUNT) s->var[which] = value;
}

static void clear_fields(rbco_sim *s)
{
    size_t n = (size_t)s->rows * s->sx * sizeof(double);
    double *all[] = {s->u, s->w, s->b, s->phy, s->pnhs, s->gu, s->gw, s->gb, s->gu0, s->gw0, s->gb0};
    for (unsigned a = 0; a < sizeof(all) / sizeof(all[0]); ++a) memset(all[a], 0, n);
}

void rbco_load_raw(rbco_sim *s, const double *b, const double *u, const double *w)
{
    clear_fields(s);
    for (int k = 0; k < s->nz; ++k)
        for (int i = 0; i < s->nx; ++i) {
            s->b[IDX(s, i, k)] = b[k * s->nx + i];
            s->u[IDX(s, i, k)] = u[k * s->nx + i];
        }
    for (int k = 0; k <= s->nz; ++k)
        for (int i = 0; i < s->nx; ++i) s->w[IDX(s, i, k)] = w[k * s->nx + i];
}

/* [OC] Models/NonhydrostaticModels/set_nonhydrostatic_model.jl: set! fills halos, then
   enforces incompressibility with one projection of unit time step; call sites
   rbc_sim2D.jl:170 (random IC) and :184 (checkpoint IC); then initialize_simulation resets
   the api counters (rbc_sim2D_api.jl:47,67-68) */
static void finish_reset(rbco_sim *s)
{
    rbco_set_action(s, NULL);      /* api:47 action = zeros(actuators) */
    fill_halo_u(s, s->u);
    fill_halo_w(s, s->w);
    fill_halo_b(s, s->b);
    update_hydrostatic_pressure(s);
    pressure_project(s, 1.0);
    fill_halo_u(s, s->u);
    fill_halo_w(s, s->w);
    fill_halo_b(s, s->b);
    update_hydrostatic_pressure(s);
    s->step = 1;
    s->time = 0.0;
}

void rbco_reset_from_arrays(rbco_sim *s, const double *b, const double *u, const double *w)
{
    rbco_load_raw(s, b, u, w);
    finish_reset(s);
}

/* initialize_model, rbc_sim2D.jl:163-171 (distribution-level restatement: own RNG) */
void rbco_reset_random(rbco_sim *s, uint64_t seed)
{
    clear_fields(s);
    const double kick = s->c.random_kick, min_b = s->c.min_b, db = s->c.delta_b, Lz = s->c.lz;
    for (int k = 0; k < s->nz; ++k)
        for (int i = 0; i < s->nx; ++i) {
            uint32_t id = (uint32_t)(k * s->nx + i);
            s->u[IDX(s, i, k)] = kick * rbco_normal(seed, 0, id);
            double z = (k + 0.5) * s->dz;
            double v = min_b + (Lz - z) * db / 2 + kick * rbco_normal(seed, 2, id);
            s->b[IDX(s, i, k)] = v < min_b ? min_b : (v > min_b + db ? min_b + db : v);
        }
    for (int k = 0; k <= s->nz; ++k)
        for (int i = 0; i < s->nx; ++i) s->w[IDX(s, i, k)] = kick * rbco_normal(seed, 1, (uint32_t)(k * s->nx + i));
    finish_reset(s);
}

/* ------------------------------------------------------------------------------------------ */
/* outputs                                                                                    */
/* ------------------------------------------------------------------------------------------ */
void rbco_get_tendencies(const rbco_sim *s, double *gb, double *gu, double *gw)
{
    for (int k = 0; k < s->nz; ++k)
        for (int i = 0; i < s->nx; ++i) {
            gb[k * s->nx + i] = s->gb[IDX(s, i, k)];
            gu[k * s->nx + i] = s->gu[IDX(s, i, k)];
            gw[k * s->nx + i] = s->gw[IDX(s, i, k)];
        }
}

void rbco_get_fields(const rbco_sim *s, double *b, double *u, double *w)
{
    for (int k = 0; k < s->nz; ++k)
        for (int i = 0; i < s->nx; ++i) {
            b[k * s->nx + i] = s->b[IDX(s, i, k)];
            u[k * s->nx + i] = s->u[IDX(s, i, k)];
        }
    for (int k = 0; k <= s->nz; ++k)
        for (int i = 0; i < s->nx; ++i) w[k * s->nx + i] = s->w[IDX(s, i, k)];
}

/* d(u,w,b)/dt of the semi-discrete system after projection: used to test that the stored
   Ra=1e4 checkpoint states are fixed points of the restated operator */
void rbco_projected_rate(rbco_sim *s, double *du, double *dw, double *db)
{
    const int nx = s->nx, nz = s->nz;
    size_t n = (size_t)s->rows * s->sx;
    double *su = malloc(n * sizeof(double)), *sw = malloc(n * sizeof(double));
    memcpy(su, s->u, n * sizeof(double));
    memcpy(sw, s->w, n * sizeof(double));
    rbco_update_state(s);
    for (int k = 0; k < nz; ++k)
        for (int i = 0; i < nx; ++i) {
            db[k * nx + i] = s->gb[IDX(s, i, k)];
            s->u[IDX(s, i, k)] = s->gu[IDX(s, i, k)];
            s->w[IDX(s, i, k)] = s->gw[IDX(s, i, k)];
        }
    pressure_project(s, 1.0);
    for (int k = 0; k < nz; ++k)
        for (int i = 0; i < nx; ++i) {
            du[k * nx + i] = s->u[IDX(s, i, k)];
            dw[k * nx + i] = s->w[IDX(s, i, k)];
        }
    memcpy(s->u, su, n * sizeof(double));
    memcpy(s->w, sw, n * sizeof(double));
    free(su);
    free(sw);
    rbco_update_state(s);
}

/* get_state, rbc_sim2D_api.jl:102-118; python transposes to (C,z,x), rbc2D.py:184-189 */
void rbco_get_state(const rbco_sim *s, double *out, int nch)
{
    const double *ch[5] = {s->b, s->u, s->w, s->phy, s->pnhs};
    for (int c = 0; c < nch; ++c)
        for (int k = 0; k < s->nz; ++k)
            for (int i = 0; i < s->nx; ++i) out[((size_t)c * s->nz + k) * s->nx + i] = ch[c][IDX(s, i, k)];
}

void rbco_get_state_f32(const rbco_sim *s, float *out, int nch)
{
    const double *ch[5] = {s->b, s->u, s->w, s->phy, s->pnhs};
    for (int c = 0; c < nch; ++c)
        for (int k = 0; k < s->nz; ++k)
            for (int i = 0; i < s->nx; ++i) out[((size_t)c * s->nz + k) * s->nx + i] = (float)ch[c][IDX(s, i, k)];
}

/* get_observation, rbc_sim2D_api.jl:123-129: x index 1:Nx/No_x:Nx, z index 1:Nz/No_z:Nz */
void rbco_get_obs_f32(const rbco_sim *s, float *out, int nch)
{
    const double *ch[5] = {s->b, s->u, s->w, s->phy, s->pnhs};
    const int stx = s->nx / s->c.obs_nx, stz = s->nz / s->c.obs_nz;
    for (int c = 0; c < nch; ++c)
        for (int k = 0; k < s->c.obs_nz; ++k)
            for (int i = 0; i < s->c.obs_nx; ++i)
                out[((size_t)c * s->c.obs_nz + k) * s->c.obs_nx + i] = (float)ch[c][IDX(s, i * stx, k * stz)];
}

/* get_nusselt, rbc_sim2D_api.jl:142-163 with array_gradient rbc_sim2D.jl:206-220 */
double rbco_nusselt(const rbco_sim *s, int on_state)
{
    const int stx = on_state ? 1 : s->nx / s->c.obs_nx, stz = on_state ? 1 : s->nz / s->c.obs_nz;
    const int mx = on_state ? s->nx : s->c.obs_nx, mz = on_state ? s->nz : s->c.obs_nz;
    double q1 = 0;
    double *tx = malloc(sizeof(double) * mz);
    for (int k = 0; k < mz; ++k) {
        double row = 0;
        for (int i = 0; i < mx; ++i) {
            const int c = IDX(s, i * stx, k * stz);
            q1 += s->b[c] * s->w[c];
            row += s->b[c];
        }
        tx[k] = row / mx;
    }
    q1 /= (double)mx * mz;
    double g = 0;
    for (int k = 0; k < mz; ++k) {
        if (k == 0) g += tx[1] - tx[0];
        else if (k == mz - 1) g += tx[k] - tx[k - 1];
        else g += (tx[k + 1] - tx[k - 1]) / 2;
    }
    double q2 = s->kappa * (g / mz);
    free(tx);
    return (q1 - q2) / (s->kappa * s->c.delta_b / s->c.lz);
}

void rbco_get_info(const rbco_sim *s, double *t, int64_t *step)
{
    *t = s->time;
    *step = s->step;
}

double rbco_max_divergence(const rbco_sim *s)
{
    double m = 0;
    for (int k = 0; k < s->nz; ++k)
        for (int i = 0; i < s->nx; ++i) {
            int ip = (i + 1) % s->nx;
            double d = (s->u[IDX(s, ip, k)] - s->u[IDX(s, i, k)]) / s->dx + (s->w[IDX(s, i, k + 1)] - s->w[IDX(s, i, k)]) / s->dz;
            if (fabs(d) > m) m = fabs(d);
        }
    return m;
}

double rbco_kinetic_energy(const rbco_sim *s)
{
    double a = 0, bsum = 0;
    for (int k = 0; k < s->nz; ++k)
        for (int i = 0; i < s->nx; ++i) {
            a += s->u[IDX(s, i, k)] * s->u[IDX(s, i, k)];
            bsum += s->w[IDX(s, i, k)] * s->w[IDX(s, i, k)];
        }
    return 0.5 * (a + bsum) / ((double)s->nx * s->nz);
}
