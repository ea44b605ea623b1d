/*
 * rbc_oracle3d.c -- CPU oracle (fp64, one env) for the 3D Rayleigh-Benard hot path.
 * TEST INFRASTRUCTURE ONLY (same rules as rbc_oracle.c / rbc_oracle.h).
 *
 * Restates src/rbc_gym/sim/rbc_sim3D.jl + rbc_sim3D_api.jl on top of the same Oceananigans
 * v0.92.0 discretisation as the 2D oracle ([OC] tags, see rbc_oracle.c): grid
 * (Periodic, Periodic, Bounded), UpwindBiased(5) advection with the wall-adjacent order
 * reduction pinned on the 2D checkpoint data, ScalarDiffusivity stress divergence, split
 * hydrostatic pressure, Le-Moin RK3, exact FFT-xy + tridiagonal-z pressure projection.
 * PARITY STATUS: "parity unpinned" at trajectory level (no 3D checkpoint data ships with the
 * reference: data/checkpoints/.../3D_ckpt_ra2500.h5 are listed in .MISSING_LARGE_BLOBS); the only
 * 3D pin is statistical (experiments/flowstats/flowstats_ra.pkl).
 *
 * Index conventions: cells i<nx (x), j<ny (y), k<nz (z); u on x-faces, v on y-faces, w on z-faces
 * k=0..nz (walls at 0 and nz); arrays [k][j][i] with a 3-cell halo in every direction.
 */
#include "rbc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define H3 3
#define MAXH3 32


struct rbco3_sim {
    rbco3_config c;
    int nx, ny, nz, sx, sy, rows;
    size_t n;
    double dx, dy, dz, nu, kappa, tff;
    double *u, *v, *w, *b, *phy, *pnhs;
    double *g[4], *g0[4];                 /* tendencies of u,v,w,b and previous stage */
    double *rhs, *phi;                    /* interior nz*ny*nx */
    double action[MAXH3 * MAXH3];         /* preprocessed wall temperatures per segment */
    double time;
    int64_t step;
};

#define I3(s, i, j, k) ((((size_t)(k) + H3) * (s)->sy + ((j) + H3)) * (s)->sx + ((i) + H3))

/* ---- stencils (same as the 2D oracle, pinned variant) ------------------------------------- */
static inline double l5(const double *p, long st) { return (2 * p[-3 * st] - 13 * p[-2 * st] + 47 * p[-st] + 27 * p[0] - 3 * p[st]) / 60; }
static inline double r5(const double *p, long st) { return (-3 * p[-2 * st] + 27 * p[-st] + 47 * p[0] - 13 * p[st] + 2 * p[2 * st]) / 60; }
static inline double l3(const double *p, long st) { return (-p[-2 * st] + 5 * p[-st] + 2 * p[0]) / 6; }
static inline double r3(const double *p, long st) { return (2 * p[-st] + 5 * p[0] - p[st]) / 6; }
static inline double s4(const double *p, long st) { return (-p[-2 * st] + 7 * p[-st] + 7 * p[0] - p[st]) / 12; }
static inline double s2(const double *p, long st) { return (p[-st] + p[0]) / 2; }
static inline double upw(double ut, double L, double R) { return ((ut + fabs(ut)) * L + (ut - fabs(ut)) * R) / 2; }

/* z (Bounded): centre field -> face k (p -> psi[k]); 1-based face index k+1; one buffer test for
   both biases: 5th if 4<=k1<=N-2, 3rd if 3<=k1<=N-1, else 1st; advecting velocity C4 where 5th */
static double zf_L(const rbco3_sim *s, const double *p, int k, long st)
{ int k1 = k + 1, N = s->nz; if (k1 >= 4 && k1 <= N - 2) return l5(p, st); if (k1 >= 3 && k1 <= N - 1) return l3(p, st); return p[-st]; }
static double zf_R(const rbco3_sim *s, const double *p, int k, long st)
{ int k1 = k + 1, N = s->nz; if (k1 >= 4 && k1 <= N - 2) return r5(p, st); if (k1 >= 3 && k1 <= N - 1) return r3(p, st); return p[0]; }
static double zf_S(const rbco3_sim *s, const double *p, int k, long st)
{ int k1 = k + 1, N = s->nz; if (k1 >= 4 && k1 <= N - 2) return s4(p, st); return s2(p, st); }
/* face field -> centre k (p -> psi[k+1]); 5th if 3<=k1<=N-2, 3rd if 2<=k1<=N-1 */
static double zc_L(const rbco3_sim *s, const double *p, int k, long st)
{ int k1 = k + 1, N = s->nz; if (k1 >= 3 && k1 <= N - 2) return l5(p, st); if (k1 >= 2 && k1 <= N - 1) return l3(p, st); return p[-st]; }
static double zc_R(const rbco3_sim *s, const double *p, int k, long st)
{ int k1 = k + 1, N = s->nz; if (k1 >= 3 && k1 <= N - 2) return r5(p, st); if (k1 >= 2 && k1 <= N - 1) return r3(p, st); return p[0]; }
static double zc_S(const rbco3_sim *s, const double *p, int k, long st)
{ int k1 = k + 1, N = s->nz; if (k1 >= 3 && k1 <= N - 2) return s4(p, st); return s2(p, st); }

/* ---- A10 (3D): preprocess_action + bottom_T, rbc_sim3D.jl:111-141 ------------------------------ */
void rbco3_set_action(rbco3_sim *s, const float *action, int raw_zero)
{
    const int n = s->c.heaters, nn = n * n;
    if (raw_zero || !action) {           /* api:49: the global is raw zeros(8,8) until the first step */
        for (int a = 0; a < nn; ++a) s->action[a] = 0.0;
        return;
    }
    double mean = 0, mx = 0;
    for (int a = 0; a < nn; ++a) mean += (double)action[a];
    mean /= nn;
    for (int a = 0; a < nn; ++a) { double d = fabs((double)action[a] - mean); if (d > mx) mx = d; }
    double K = mx > 1 ? mx : 1;
    /* action arrives as numpy (n,n) row-major = a[row][col]; Julia indexes action[i,j] with i<-x, j<-y on the
       same memory viewed column-major is NOT what juliacall does: the array keeps its logical indices,
       so action[i,j] = a[i-1][j-1] */
    for (int a = 0; a < nn; ++a) s->action[a] = (s->c.min_b + s->c.delta_b) + (((double)action[a] - mean) / K) * s->c.heater_limit;
}

static double bottom_T3(const rbco3_sim *s, int i, int j)
{
    const int n = s->c.heaters;
    double x = (i + 0.5) * s->dx, y = (j + 0.5) * s->dy;
    int a = (int)floor(x / s->c.lx * n) + 1, bq = (int)floor(y / s->c.ly * n) + 1;
    if (a < 1) a = 1;
    if (a > n) a = n;
    if (bq < 1) bq = 1;
    if (bq > n) bq = n;
    return s->action[(a - 1) * n + (bq - 1)];
}

/* ---- halos ------------------------------------------------------------------------------------ */
static void periodic_xy(const rbco3_sim *s, double *a)
{
    for (int k = -H3; k < s->nz + 1 + H3; ++k) {
        for (int j = 0; j < s->ny; ++j)
            for (int h = 1; h <= H3; ++h) {
                a[I3(s, -h, j, k)] = a[I3(s, s->nx - h, j, k)];
                a[I3(s, s->nx - 1 + h, j, k)] = a[I3(s, h - 1, j, k)];
            }
        for (int i = -H3; i < s->nx + H3; ++i)
            for (int h = 1; h <= H3; ++h) {
                a[I3(s, i, -h, k)] = a[I3(s, i, s->ny - h, k)];
                a[I3(s, i, s->ny - 1 + h, k)] = a[I3(s, i, h - 1, k)];
            }
    }
}

static void fill_noslip(const rbco3_sim *s, double *f)
{
    const double dz = s->dz;
    for (int j = 0; j < s->ny; ++j)
        for (int i = 0; i < s->nx; ++i) {
            double c1 = f[I3(s, i, j, 0)], cN = f[I3(s, i, j, s->nz - 1)];
            f[I3(s, i, j, -1)] = c1 + ((c1 - 0.0) / (dz / 2)) * (-dz);
            f[I3(s, i, j, s->nz)] = cN + ((0.0 - cN) / (dz / 2)) * dz;
        }
    periodic_xy(s, f);
}

static void fill_b(const rbco3_sim *s, double *b)
{
    const double dz = s->dz;
    for (int j = 0; j < s->ny; ++j)
        for (int i = 0; i < s->nx; ++i) {
            double c1 = b[I3(s, i, j, 0)], cN = b[I3(s, i, j, s->nz - 1)];
            b[I3(s, i, j, -1)] = c1 + ((c1 - bottom_T3(s, i, j)) / (dz / 2)) * (-dz);
            b[I3(s, i, j, s->nz)] = cN + ((s->c.min_b - cN) / (dz / 2)) * dz;
        }
    periodic_xy(s, b);
}

static void fill_w(const rbco3_sim *s, double *w)
{
    for (int j = 0; j < s->ny; ++j)
        for (int i = 0; i < s->nx; ++i) { w[I3(s, i, j, 0)] = 0.0; w[I3(s, i, j, s->nz)] = 0.0; }
    periodic_xy(s, w);
}

static void fill_p(const rbco3_sim *s, double *p)
{
    for (int j = 0; j < s->ny; ++j)
        for (int i = 0; i < s->nx; ++i) { p[I3(s, i, j, -1)] = p[I3(s, i, j, 0)]; p[I3(s, i, j, s->nz)] = p[I3(s, i, j, s->nz - 1)]; }
    periodic_xy(s, p);
}

static void hydrostatic(rbco3_sim *s)
{
    for (int j = 0; j < s->ny; ++j)
        for (int i = 0; i < s->nx; ++i) {
            s->phy[I3(s, i, j, s->nz - 1)] = -(0.5 * (s->b[I3(s, i, j, s->nz - 1)] + s->b[I3(s, i, j, s->nz)])) * s->dz;
            for (int k = s->nz - 2; k >= 0; --k)
                s->phy[I3(s, i, j, k)] = s->phy[I3(s, i, j, k + 1)] - (0.5 * (s->b[I3(s, i, j, k)] + s->b[I3(s, i, j, k + 1)])) * s->dz;
        }
    periodic_xy(s, s->phy);
}

/* ---- tendencies --------------------------------------------------------------------------------- */
static void tendencies(rbco3_sim *s)
{
    const int nx = s->nx, ny = s->ny, nz = s->nz;
    const long X = 1, Y = s->sx, Z = (long)s->sx * s->sy;
    const double dx = s->dx, dy = s->dy, dz = s->dz, Ax = dy * dz, Ay = dx * dz, Az = dx * dy, V = dx * dy * dz;
    const double nu = s->nu, ka = s->kappa;
    const double *u = s->u, *v = s->v, *w = s->w, *b = s->b;
#define FUU(c)  upw(Ax * s4(u + (c) + X, X), l5(u + (c) + X, X), r5(u + (c) + X, X))                   /* centre i      */
#define FVU(c)  upw(Ay * s4(v + (c), X), l5(u + (c), Y), r5(u + (c), Y))                               /* (xf i, yf j)  */
#define FWU(c, k) upw(Az * s4(w + (c), X), zf_L(s, u + (c), k, Z), zf_R(s, u + (c), k, Z))              /* (xf i, zf k)  */
#define FUV(c)  upw(Ax * s4(u + (c), Y), l5(v + (c), X), r5(v + (c), X))                               /* (xf i, yf j)  */
#define FVV(c)  upw(Ay * s4(v + (c) + Y, Y), l5(v + (c) + Y, Y), r5(v + (c) + Y, Y))                   /* centre j      */
#define FWV(c, k) upw(Az * s4(w + (c), Y), zf_L(s, v + (c), k, Z), zf_R(s, v + (c), k, Z))              /* (yf j, zf k)  */
#define FUW(c, k) upw(Ax * zf_S(s, u + (c), k, Z), l5(w + (c), X), r5(w + (c), X))                      /* (xf i, zf k)  */
#define FVW(c, k) upw(Ay * zf_S(s, v + (c), k, Z), l5(w + (c), Y), r5(w + (c), Y))                      /* (yf j, zf k)  */
#define FWW(c, k) upw(Az * zc_S(s, w + (c) + Z, k, Z), zc_L(s, w + (c) + Z, k, Z), zc_R(s, w + (c) + Z, k, Z)) /* centre k */
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const size_t c = I3(s, i, j, k);
                /* u */
                double adv = (FUU(c) - FUU(c - X) + FVU(c + Y) - FVU(c) + FWU(c + Z, k + 1) - FWU(c, k)) / V;
                double t11e = -2 * nu * (u[c + X] - u[c]) / dx, t11w = -2 * nu * (u[c] - u[c - X]) / dx;
                double t12n = -nu * ((u[c + Y] - u[c]) / dy + (v[c + Y] - v[c + Y - X]) / dx);
                double t12s = -nu * ((u[c] - u[c - Y]) / dy + (v[c] - v[c - X]) / dx);
                double t13t = -nu * ((u[c + Z] - u[c]) / dz + (w[c + Z] - w[c + Z - X]) / dx);
                double t13b = -nu * ((u[c] - u[c - Z]) / dz + (w[c] - w[c - X]) / dx);
                double vis = -((Ax * t11e - Ax * t11w) + (Ay * t12n - Ay * t12s) + (Az * t13t - Az * t13b)) / V;
                s->g[0][c] = -adv + vis - (s->phy[c] - s->phy[c - X]) / dx;
                /* v */
                adv = (FUV(c + X) - FUV(c) + FVV(c) - FVV(c - Y) + FWV(c + Z, k + 1) - FWV(c, k)) / V;
                double t21e = -nu * ((u[c + X] - u[c + X - Y]) / dy + (v[c + X] - v[c]) / dx);
                double t21w = -nu * ((u[c] - u[c - Y]) / dy + (v[c] - v[c - X]) / dx);
                double t22n = -2 * nu * (v[c + Y] - v[c]) / dy, t22s = -2 * nu * (v[c] - v[c - Y]) / dy;
                double t23t = -nu * ((v[c + Z] - v[c]) / dz + (w[c + Z] - w[c + Z - Y]) / dy);
                double t23b = -nu * ((v[c] - v[c - Z]) / dz + (w[c] - w[c - Y]) / dy);
                vis = -((Ax * t21e - Ax * t21w) + (Ay * t22n - Ay * t22s) + (Az * t23t - Az * t23b)) / V;
                s->g[1][c] = -adv + vis - (s->phy[c] - s->phy[c - Y]) / dy;
                /* b */
                double fxe = Ax * upw(u[c + X], l5(b + c + X, X), r5(b + c + X, X)), fxw = Ax * upw(u[c], l5(b + c, X), r5(b + c, X));
                double fyn = Ay * upw(v[c + Y], l5(b + c + Y, Y), r5(b + c + Y, Y)), fys = Ay * upw(v[c], l5(b + c, Y), r5(b + c, Y));
                double fzt = Az * upw(w[c + Z], zf_L(s, b + c + Z, k + 1, Z), zf_R(s, b + c + Z, k + 1, Z));
                double fzb = Az * upw(w[c], zf_L(s, b + c, k, Z), zf_R(s, b + c, k, Z));
                double qxe = -ka * (b[c + X] - b[c]) / dx, qxw = -ka * (b[c] - b[c - X]) / dx;
                double qyn = -ka * (b[c + Y] - b[c]) / dy, qys = -ka * (b[c] - b[c - Y]) / dy;
                double qzt = -ka * (b[c + Z] - b[c]) / dz, qzb = -ka * (b[c] - b[c - Z]) / dz;
                s->g[3][c] = -((fxe - fxw) + (fyn - fys) + (fzt - fzb)) / V
                             - ((Ax * qxe - Ax * qxw) + (Ay * qyn - Ay * qys) + (Az * qzt - Az * qzb)) / V;
                /* w (face k; wall face never evolves) */
                if (k == 0) { s->g[2][c] = 0.0; continue; }
                adv = (FUW(c + X, k) - FUW(c, k) + FVW(c + Y, k) - FVW(c, k) + FWW(c, k) - FWW(c - Z, k - 1)) / V;
                double t31e = -nu * ((u[c + X] - u[c + X - Z]) / dz + (w[c + X] - w[c]) / dx);
                double t31w = -nu * ((u[c] - u[c - Z]) / dz + (w[c] - w[c - X]) / dx);
                double t32n = -nu * ((v[c + Y] - v[c + Y - Z]) / dz + (w[c + Y] - w[c]) / dy);
                double t32s = -nu * ((v[c] - v[c - Z]) / dz + (w[c] - w[c - Y]) / dy);
                double t33t = -2 * nu * (w[c + Z] - w[c]) / dz, t33b = -2 * nu * (w[c] - w[c - Z]) / dz;
                vis = -((Ax * t31e - Ax * t31w) + (Ay * t32n - Ay * t32s) + (Az * t33t - Az * t33b)) / V;
                s->g[2][c] = -adv + vis;
            }
}

void rbco3_update_state(rbco3_sim *s)
{
    fill_noslip(s, s->u);
    fill_noslip(s, s->v);
    fill_w(s, s->w);
    fill_b(s, s->b);
    hydrostatic(s);
    tendencies(s);
}

/* ---- pressure: DFT in x and y, tridiagonal in z -------------------------------------------------- */
static void solve_poisson(rbco3_sim *s)
{
    const int nx = s->nx, ny = s->ny, nz = s->nz;
    const double pi = 3.14159265358979323846, o = 1.0 / (s->dz * s->dz);
    size_t nn = (size_t)nx * ny * nz;
    double *re = malloc(nn * sizeof(double)), *im = malloc(nn * sizeof(double));
    double *tr = malloc(nn * sizeof(double)), *ti = malloc(nn * sizeof(double));
    double *cx = malloc(nx * sizeof(double)), *sxn = malloc(nx * sizeof(double)), *cy = malloc(ny * sizeof(double)), *syn = malloc(ny * sizeof(double));
    for (int m = 0; m < nx; ++m) { cx[m] = cos(2 * pi * m / nx); sxn[m] = sin(2 * pi * m / nx); }
    for (int m = 0; m < ny; ++m) { cy[m] = cos(2 * pi * m / ny); syn[m] = sin(2 * pi * m / ny); }
    /* forward x */
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int m = 0; m < nx; ++m) {
                double a = 0, bb = 0;
                for (int i = 0; i < nx; ++i) { int t = (int)(((long)m * i) % nx); double x = s->rhs[((size_t)k * ny + j) * nx + i]; a += x * cx[t]; bb -= x * sxn[t]; }
                tr[((size_t)k * ny + j) * nx + m] = a; ti[((size_t)k * ny + j) * nx + m] = bb;
            }
    /* forward y */
    for (int k = 0; k < nz; ++k)
        for (int n = 0; n < ny; ++n)
            for (int m = 0; m < nx; ++m) {
                double a = 0, bb = 0;
                for (int j = 0; j < ny; ++j) {
                    int t = (int)(((long)n * j) % ny);
                    double xr = tr[((size_t)k * ny + j) * nx + m], xi = ti[((size_t)k * ny + j) * nx + m];
                    a += xr * cy[t] + xi * syn[t]; bb += xi * cy[t] - xr * syn[t];
                }
                re[((size_t)k * ny + n) * nx + m] = a; im[((size_t)k * ny + n) * nx + m] = bb;
            }
    /* z: Thomas per (m,n); mean mode pinned then its z-mean removed */
    double *piv = malloc(nz * sizeof(double));
    for (int n = 0; n < ny; ++n)
        for (int m = 0; m < nx; ++m) {
            double tx = 2 * sin(m * pi / nx) / s->dx, ty = 2 * sin(n * pi / ny) / s->dy, lam = tx * tx + ty * ty;
            for (int part = 0; part < 2; ++part) {
                double *r = part ? im : re;
                double prev = 0;
                for (int k = 0; k < nz; ++k) {
                    double d = -((k == 0 || k == nz - 1) ? 1.0 : 2.0) * o - lam;
                    if (m == 0 && n == 0 && k == nz - 1) d -= o;
                    double l = (k == 0) ? 0.0 : o / piv[k - 1];
                    if (part == 0) piv[k] = d - l * ((k == 0) ? 0.0 : o);
                    size_t q = ((size_t)k * ny + n) * nx + m;
                    r[q] = r[q] - l * prev;
                    prev = r[q];
                }
                size_t qN = ((size_t)(nz - 1) * ny + n) * nx + m;
                r[qN] /= piv[nz - 1];
                for (int k = nz - 2; k >= 0; --k) {
                    size_t q = ((size_t)k * ny + n) * nx + m, qp = ((size_t)(k + 1) * ny + n) * nx + m;
                    r[q] = (r[q] - o * r[qp]) / piv[k];
                }
            }
            if (m == 0 && n == 0) {
                double mean = 0;
                for (int k = 0; k < nz; ++k) mean += re[(size_t)k * ny * nx];
                mean /= nz;
                for (int k = 0; k < nz; ++k) re[(size_t)k * ny * nx] -= mean;
            }
        }
    /* inverse y then x */
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int m = 0; m < nx; ++m) {
                double a = 0, bb = 0;
                for (int n = 0; n < ny; ++n) {
                    int t = (int)(((long)n * j) % ny);
                    double xr = re[((size_t)k * ny + n) * nx + m], xi = im[((size_t)k * ny + n) * nx + m];
                    a += xr * cy[t] - xi * syn[t]; bb += xi * cy[t] + xr * syn[t];
                }
                tr[((size_t)k * ny + j) * nx + m] = a / ny; ti[((size_t)k * ny + j) * nx + m] = bb / ny;
            }
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                double a = 0;
                for (int m = 0; m < nx; ++m) { int t = (int)(((long)m * i) % nx); a += tr[((size_t)k * ny + j) * nx + m] * cx[t] - ti[((size_t)k * ny + j) * nx + m] * sxn[t]; }
                s->phi[((size_t)k * ny + j) * nx + i] = a / nx;
            }
    free(re); free(im); free(tr); free(ti); free(cx); free(sxn); free(cy); free(syn); free(piv);
}

static void project(rbco3_sim *s, double dts)
{
    const int nx = s->nx, ny = s->ny, nz = s->nz;
    const long X = 1, Y = s->sx, Z = (long)s->sx * s->sy;
    fill_noslip(s, s->u); fill_noslip(s, s->v); fill_w(s, s->w);
    const double Ax = s->dy * s->dz, Ay = s->dx * s->dz, Az = s->dx * s->dy, V = s->dx * s->dy * s->dz;
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                size_t c = I3(s, i, j, k);
                double div = ((Ax * s->u[c + X] - Ax * s->u[c]) + (Ay * s->v[c + Y] - Ay * s->v[c]) + (Az * s->w[c + Z] - Az * s->w[c])) / V;
                s->rhs[((size_t)k * ny + j) * nx + i] = div / dts;
            }
    solve_poisson(s);
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) s->pnhs[I3(s, i, j, k)] = s->phi[((size_t)k * ny + j) * nx + i];
    fill_p(s, s->pnhs);
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                size_t c = I3(s, i, j, k);
                s->u[c] -= (s->pnhs[c] - s->pnhs[c - X]) / s->dx * dts;
                s->v[c] -= (s->pnhs[c] - s->pnhs[c - Y]) / s->dy * dts;
                s->w[c] -= (s->pnhs[c] - s->pnhs[c - Z]) / s->dz * dts;
            }
}

static void rk_stage(rbco3_sim *s, double dt, double g, double z, int first)
{
    double *f[4] = {s->u, s->v, s->w, s->b};
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < s->nz; ++k)
            for (int j = 0; j < s->ny; ++j)
                for (int i = 0; i < s->nx; ++i) {
                    size_t c = I3(s, i, j, k);
                    if (first) f[q][c] += dt * g * s->g[q][c];
                    else f[q][c] += dt * (g * s->g[q][c] + z * s->g0[q][c]);
                }
}

static void store(rbco3_sim *s) { for (int q = 0; q < 4; ++q) memcpy(s->g0[q], s->g[q], s->n * sizeof(double)); }

void rbco3_substep(rbco3_sim *s, double dt)
{
    const double g1 = 8.0 / 15, g2 = 5.0 / 12, g3 = 3.0 / 4, z2 = -17.0 / 60, z3 = -5.0 / 12;
    rk_stage(s, dt, g1, 0, 1); project(s, g1 * dt); store(s); rbco3_update_state(s);
    rk_stage(s, dt, g2, z2, 0); project(s, (g2 + z2) * dt); store(s); rbco3_update_state(s);
    rk_stage(s, dt, g3, z3, 0); project(s, (g3 + z3) * dt); rbco3_update_state(s);
}

/* step_simulation, rbc_sim3D_api.jl:77-101 */
int rbco3_step(rbco3_sim *s, const float *action)
{
    rbco3_set_action(s, action, 0);
    rbco3_update_state(s);
    const double T = s->c.dt_control * s->tff, dt0 = s->c.dt_solver * s->tff;
    int nfull = (int)floor(T / dt0 + 1e-9);
    for (int n = 0; n < nfull; ++n) rbco3_substep(s, dt0);
    if (T - nfull * dt0 > 1e-9 * dt0) rbco3_substep(s, T - nfull * dt0);
    s->time += s->c.dt_control * s->tff;   /* api:89 */
    s->step += 1;
    const double *f[4] = {s->u, s->v, s->w, s->b};
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < s->nz; ++k)
            for (int j = 0; j < s->ny; ++j)
                for (int i = 0; i < s->nx; ++i)
                    if (isnan(f[q][I3(s, i, j, k)])) return 0;
    return 1;
}

rbco3_sim *rbco3_create(const rbco3_config *cfg)
{
    if (cfg->heaters < 1 || cfg->heaters > MAXH3) return NULL;
    rbco3_sim *s = calloc(1, sizeof(*s));
    s->c = *cfg;
    s->nx = cfg->nx; s->ny = cfg->ny; s->nz = cfg->nz;
    s->sx = s->nx + 2 * H3; s->sy = s->ny + 2 * H3; s->rows = s->nz + 1 + 2 * H3;
    s->n = (size_t)s->rows * s->sy * s->sx;
    s->dx = cfg->lx / cfg->nx; s->dy = cfg->ly / cfg->ny; s->dz = cfg->lz / cfg->nz;
    s->nu = sqrt(cfg->pr / cfg->ra); s->kappa = 1 / sqrt(cfg->pr * cfg->ra);
    s->tff = cfg->lz * cfg->lz;                                   /* api:43 */
    double **all[] = {&s->u, &s->v, &s->w, &s->b, &s->phy, &s->pnhs, &s->g[0], &s->g[1], &s->g[2], &s->g[3],
                      &s->g0[0], &s->g0[1], &s->g0[2], &s->g0[3]};
    for (unsigned a = 0; a < sizeof(all) / sizeof(all[0]); ++a) *all[a] = calloc(s->n, sizeof(double));
    s->rhs = calloc((size_t)s->nx * s->ny * s->nz, sizeof(double));
    s->phi = calloc((size_t)s->nx * s->ny * s->nz, sizeof(double));
    s->step = 1;
    return s;
}

void rbco3_destroy(rbco3_sim *s)
{
    if (!s) return;
    double *all[] = {s->u, s->v, s->w, s->b, s->phy, s->pnhs, s->g[0], s->g[1], s->g[2], s->g[3], s->g0[0], s->g0[1], s->g0[2], s->g0[3], s->rhs, s->phi};
    for (unsigned a = 0; a < sizeof(all) / sizeof(all[0]); ++a) free(all[a]);
    free(s);
}

static void clear3(rbco3_sim *s)
{
    double *all[] = {s->u, s->v, s->w, s->b, s->phy, s->pnhs, s->g[0], s->g[1], s->g[2], s->g[3], s->g0[0], s->g0[1], s->g0[2], s->g0[3]};
    for (unsigned a = 0; a < sizeof(all) / sizeof(all[0]); ++a) memset(all[a], 0, s->n * sizeof(double));
}

static void finish_reset3(rbco3_sim *s)
{
    rbco3_set_action(s, NULL, 1);
    fill_noslip(s, s->u); fill_noslip(s, s->v); fill_w(s, s->w); fill_b(s, s->b);
    hydrostatic(s);
    project(s, 1.0);
    fill_noslip(s, s->u); fill_noslip(s, s->v); fill_w(s, s->w); fill_b(s, s->b);
    hydrostatic(s);
    s->step = 1; s->time = 0.0;
}

/* arrays [k][j][i]: b,u,v nz*ny*nx ; w (nz+1)*ny*nx */
void rbco3_reset_from_arrays(rbco3_sim *s, const double *b, const double *u, const double *v, const double *w)
{
    clear3(s);
    for (int k = 0; k <= s->nz; ++k)
        for (int j = 0; j < s->ny; ++j)
            for (int i = 0; i < s->nx; ++i) {
                size_t q = ((size_t)k * s->ny + j) * s->nx + i;
                s->w[I3(s, i, j, k)] = w[q];
                if (k < s->nz) { s->b[I3(s, i, j, k)] = b[q]; s->u[I3(s, i, j, k)] = u[q]; s->v[I3(s, i, j, k)] = v[q]; }
            }
    finish_reset3(s);
}

/* initialize_model, rbc_sim3D.jl:169-179 (own counter-based RNG, fields 0:u 1:v 2:w 3:b) */
void rbco3_reset_random(rbco3_sim *s, uint64_t seed)
{
    clear3(s);
    const double kick = s->c.random_kick, min_b = s->c.min_b, db = s->c.delta_b;
    for (int k = 0; k <= s->nz; ++k)
        for (int j = 0; j < s->ny; ++j)
            for (int i = 0; i < s->nx; ++i) {
                uint32_t id = (uint32_t)(((size_t)k * s->ny + j) * s->nx + i);
                s->w[I3(s, i, j, k)] = kick * rbco_normal(seed, 2, id);
                if (k < s->nz) {
                    s->u[I3(s, i, j, k)] = kick * rbco_normal(seed, 0, id);
                    s->v[I3(s, i, j, k)] = kick * rbco_normal(seed, 1, id);
                    double z = (k + 0.5) * s->dz;
                    double val = min_b + (s->c.lz - z) * db / 2 + kick * rbco_normal(seed, 3, id);
                    s->b[I3(s, i, j, k)] = val < min_b ? min_b : (val > min_b + db ? min_b + db : val);
                }
            }
    finish_reset3(s);
}

void rbco3_get_fields(const rbco3_sim *s, double *b, double *u, double *v, double *w)
{
    for (int k = 0; k <= s->nz; ++k)
        for (int j = 0; j < s->ny; ++j)
            for (int i = 0; i < s->nx; ++i) {
                size_t q = ((size_t)k * s->ny + j) * s->nx + i;
                w[q] = s->w[I3(s, i, j, k)];
                if (k < s->nz) { b[q] = s->b[I3(s, i, j, k)]; u[q] = s->u[I3(s, i, j, k)]; v[q] = s->v[I3(s, i, j, k)]; }
            }
}

/* tendencies of the current state: gu,gv,gw,gb each nz*ny*nx */
void rbco3_get_tendencies(const rbco3_sim *s, double *gu, double *gv, double *gw, double *gb)
{
    double *o[4] = {gu, gv, gw, gb};
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < s->nz; ++k)
            for (int j = 0; j < s->ny; ++j)
                for (int i = 0; i < s->nx; ++i) o[q][((size_t)k * s->ny + j) * s->nx + i] = s->g[q][I3(s, i, j, k)];
}

/* get_state (rbc_sim3D_api.jl:106-121) after rbc3D.py:229-232: (4, nz, ny, nx) float32, channels b,u,v,w */
void rbco3_get_state_f32(const rbco3_sim *s, float *out)
{
    const double *ch[4] = {s->b, s->u, s->v, s->w};
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < s->nz; ++k)
            for (int j = 0; j < s->ny; ++j)
                for (int i = 0; i < s->nx; ++i)
                    out[(((size_t)q * s->nz + k) * s->ny + j) * s->nx + i] = (float)ch[q][I3(s, i, j, k)];
}

/* get_nusselt, rbc_sim3D_api.jl:134-159 */
double rbco3_nusselt(const rbco3_sim *s)
{
    double acc = 0;
    for (int k = 0; k < s->nz; ++k) {
        double zc = (k + 0.5) / s->nz;
        double tc = (1 - zc) * s->c.delta_b + s->c.min_b;
        for (int j = 0; j < s->ny; ++j)
            for (int i = 0; i < s->nx; ++i) acc += (s->b[I3(s, i, j, k)] - tc) * s->w[I3(s, i, j, k)];
    }
    return 1 + (acc / ((double)s->nx * s->ny * s->nz)) / s->kappa;
}

void rbco3_get_info(const rbco3_sim *s, double *t, int64_t *step) { *t = s->time; *step = s->step; }

double rbco3_max_divergence(const rbco3_sim *s)
{
    double m = 0;
    for (int k = 0; k < s->nz; ++k)
        for (int j = 0; j < s->ny; ++j)
            for (int i = 0; i < s->nx; ++i) {
                int ip = (i + 1) % s->nx, jp = (j + 1) % s->ny;
                double d = (s->u[I3(s, ip, j, k)] - s->u[I3(s, i, j, k)]) / s->dx + (s->v[I3(s, i, jp, k)] - s->v[I3(s, i, j, k)]) / s->dy
                           + (s->w[I3(s, i, j, k + 1)] - s->w[I3(s, i, j, k)]) / s->dz;
                if (fabs(d) > m) m = fabs(d);
            }
    return m;
}
