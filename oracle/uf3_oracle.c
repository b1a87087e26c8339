/*
 * uf3_oracle.c -- CPU restatement of the UF3 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; it is the checker, never the product.  The product path is the
 * HIP library under uf3_amd/csrc and fails loudly when that library is missing.
 *
 * What is restated (plain C, fp64, single thread), with the reference lines
 * (/root/reference, uf3 v0.4.0) each function follows:
 *
 *   supercell tiling and ghost indexing      uf3/data/geometry.py:14-149
 *   2-body distances / energy features       uf3/representation/distances.py:19-75
 *                                            uf3/representation/bspline.py:810-849
 *   2-body force features                    distances.py:78-143,331-364; bspline.py:852-895
 *   3-body neighbour pairs                   uf3/representation/angles.py:289-346
 *   triplet enumeration, species sort, masks angles.py:424-514
 *   per-leg basis values / derivatives       angles.py:517-632; bspline.py:950-974
 *   4x4x4 scatter (energy, forces)           angles.py:104-139, 235-286, 142-232
 *   symmetry fold + template mask            bspline.py:664-690
 *   energy / force evaluation of a model     uf3/forcefield/calculator.py:183-343
 *
 * The reference evaluates basis functions through scipy.interpolate.BSpline
 * (basis_element, extrapolate=False, NaN -> 0) and, in the calculator, through
 * ndsplines.NDSpline; neither is vendored in the reference tree.  Here every
 * basis element is evaluated from the Cox-de Boor recursion on its own five
 * knots (0/0 := 0, half-open knot intervals [t_i, t_{i+1}); like scipy's basis_element the
 * value at the element's own last knot is 0, also at the 4-fold end knot).
 *
 * Deliberately kept close to the reference's formulation (explicit supercell,
 * ghost-centre loop for 3-body forces) so that it is independent of the product,
 * which uses periodic neighbour lists with image shifts and a per-atom gather.
 * The one liberty: candidate pairs come from a uniform grid over the supercell
 * instead of a dense cdist matrix; the accepted pairs and their order (row-major
 * over (i, j) supercell indices) are identical.
 *
 * Pinned against: tests/golden/rattled_steel_features.json (reference's own
 * golden), the literal H2O / CH4 vectors and calculator energies/forces of the
 * reference's tests, and captures made by importing the reference in the build
 * container (tests/golden/make_golden.py).  See tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n_species;            /* S, ascending Z */
    const int *species_z;     /* [S] */
    int n_pairs;              /* pair blocks in column order */
    const int *pair_z;        /* [P][2] atomic numbers, pair_z[p][0] <= pair_z[p][1] */
    const int *pair_nk;       /* [P] knots per pair */
    const double *pair_knots; /* concatenated */
    const double *pair_rmin;  /* [P] r_min_map (raw) */
    const double *pair_rmax;  /* [P] */
    const int *pair_col;      /* [P] column offset of block */
    int lead2, trail2, lead3, trail3;
    int n_trios;
    const int *trio_z;        /* [T][3] centre, n1 <= n2 */
    const int *trio_nk;       /* [T][3] */
    const double *trio_knots; /* concatenated l, m, n per trio */
    const int *trio_sym;      /* [T] 1, 2, 3 */
    const int *trio_col;      /* [T] column offset */
    const int *trio_ncol;     /* [T] compressed size */
    const int64_t *trio_mask; /* concatenated template_mask */
    const double *trio_w;     /* concatenated flat_weights */
    int n_feat;               /* F (1-body columns included) */
    double r_cut;             /* supercell radius (BSplineBasis.r_cut) */
} uf3o_spec;

typedef struct {
    int n_atoms;
    const double *pos;  /* [N][3] */
    const int *z;       /* [N] */
    const double *cell; /* [3][3] rows = lattice vectors */
    const int *pbc;     /* [3] */
} uf3o_frame;

/* ---------------------------------------------------------------- supercell */
typedef struct {
    int n_img, n_atoms, m;     /* m = n_img * n_atoms */
    int fac[3], cnt[3];
    int *shift;                /* [n_img][3] */
    double *pos;               /* [m][3] */
    int *z;                    /* [m] */
} supercell_t;

static void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* geometry.py:54-83: ceil(r_cut / |projection of a_i on the normal of the other two|) */
static void supercell_factors(const double *cell, double r_cut, int *fac) {
    int all_zero = 1, any_zero_vec = 0;
    for (int i = 0; i < 9; i++) if (cell[i] != 0.0) all_zero = 0;
    for (int i = 0; i < 3; i++) if (sqrt(dot3(cell + 3 * i, cell + 3 * i)) == 0.0) any_zero_vec = 1;
    if (all_zero || any_zero_vec) { fac[0] = fac[1] = fac[2] = 1; return; }
    const double *a = cell, *b = cell + 3, *c = cell + 6;
    double n[3][3];
    cross3(b, c, n[0]); cross3(a, c, n[1]); cross3(a, b, n[2]);
    for (int i = 0; i < 3; i++) {
        const double *v = cell + 3 * i;
        double s = dot3(v, n[i]) / dot3(n[i], n[i]);
        double p[3] = {n[i][0] * s, n[i][1] * s, n[i][2] * s};
        fac[i] = (int)ceil(r_cut / sqrt(dot3(p, p)));
    }
}

/* k-th entry of the per-axis image list 0, +1, -1, +2, -2, ... (geometry.py:131-138) */
static int axis_image(int k) { return (k == 0) ? 0 : ((k & 1) ? (k + 1) / 2 : -(k / 2)); }

static void build_supercell(const uf3o_frame *f, double r_cut, supercell_t *sc) {
    int any_pbc = f->pbc[0] || f->pbc[1] || f->pbc[2];
    sc->n_atoms = f->n_atoms;
    if (!any_pbc) { sc->fac[0] = sc->fac[1] = sc->fac[2] = 0; }
    else supercell_factors(f->cell, r_cut, sc->fac);
    for (int d = 0; d < 3; d++) sc->cnt[d] = (any_pbc && f->pbc[d]) ? 2 * sc->fac[d] + 1 : 1;
    sc->n_img = sc->cnt[0] * sc->cnt[1] * sc->cnt[2];
    sc->m = sc->n_img * f->n_atoms;
    sc->shift = (int *)malloc(sizeof(int) * 3 * (size_t)sc->n_img);
    sc->pos = (double *)malloc(sizeof(double) * 3 * (size_t)sc->m);
    sc->z = (int *)malloc(sizeof(int) * (size_t)sc->m);
    int img = 0;
    /* meshgrid(xy) flatten: b slowest, a middle, c fastest (geometry.py:108-114) */
    for (int ib = 0; ib < sc->cnt[1]; ib++)
        for (int ia = 0; ia < sc->cnt[0]; ia++)
            for (int ic = 0; ic < sc->cnt[2]; ic++, img++) {
                int s[3] = {axis_image(ia), axis_image(ib), axis_image(ic)};
                memcpy(sc->shift + 3 * img, s, sizeof(s));
                double off[3];
                for (int k = 0; k < 3; k++)
                    off[k] = s[0] * f->cell[k] + s[1] * f->cell[3 + k] + s[2] * f->cell[6 + k];
                for (int a = 0; a < f->n_atoms; a++) {
                    size_t j = (size_t)img * f->n_atoms + a;
                    for (int k = 0; k < 3; k++) sc->pos[3 * j + k] = f->pos[3 * a + k] + off[k];
                    sc->z[j] = f->z[a];
                }
            }
}
static void free_supercell(supercell_t *sc) { free(sc->shift); free(sc->pos); free(sc->z); }

/* ------------------------------------------------- uniform grid over supercell */
typedef struct {
    double lo[3], inv;
    int n[3];
    int *start;   /* [nbins+1] */
    int *items;   /* supercell indices, ascending within a bin */
} grid_t;

static void build_grid(const supercell_t *sc, double h, grid_t *g) {
    double hi[3] = {-1e300, -1e300, -1e300};
    g->lo[0] = g->lo[1] = g->lo[2] = 1e300;
    for (int j = 0; j < sc->m; j++)
        for (int k = 0; k < 3; k++) {
            double v = sc->pos[3 * (size_t)j + k];
            if (v < g->lo[k]) g->lo[k] = v;
            if (v > hi[k]) hi[k] = v;
        }
    if (h <= 0) h = 1.0;
    g->inv = 1.0 / h;
    size_t nb = 1;
    for (int k = 0; k < 3; k++) {
        g->n[k] = (int)floor((hi[k] - g->lo[k]) * g->inv) + 1;
        if (g->n[k] < 1) g->n[k] = 1;
        nb *= (size_t)g->n[k];
    }
    g->start = (int *)calloc(nb + 1, sizeof(int));
    g->items = (int *)malloc(sizeof(int) * (size_t)(sc->m > 0 ? sc->m : 1));
    int *bin = (int *)malloc(sizeof(int) * (size_t)(sc->m > 0 ? sc->m : 1));
    for (int j = 0; j < sc->m; j++) {
        int c[3];
        for (int k = 0; k < 3; k++) {
            c[k] = (int)floor((sc->pos[3 * (size_t)j + k] - g->lo[k]) * g->inv);
            if (c[k] < 0) c[k] = 0;
            if (c[k] >= g->n[k]) c[k] = g->n[k] - 1;
        }
        bin[j] = (c[0] * g->n[1] + c[1]) * g->n[2] + c[2];
        g->start[bin[j] + 1]++;
    }
    for (size_t b = 0; b < nb; b++) g->start[b + 1] += g->start[b];
    int *fill = (int *)malloc(sizeof(int) * nb);
    for (size_t b = 0; b < nb; b++) fill[b] = g->start[b];
    for (int j = 0; j < sc->m; j++) g->items[fill[bin[j]]++] = j;
    free(fill); free(bin);
}
static void free_grid(grid_t *g) { free(g->start); free(g->items); }

static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }

/* all supercell atoms within the 27 bins around x, ascending index (np.where order) */
static int gather_candidates(const supercell_t *sc, const grid_t *g, const double *x, int **buf, int *cap) {
    int c[3], n = 0;
    for (int k = 0; k < 3; k++) {
        c[k] = (int)floor((x[k] - g->lo[k]) * g->inv);
        if (c[k] < 0) c[k] = 0;
        if (c[k] >= g->n[k]) c[k] = g->n[k] - 1;
    }
    for (int a = c[0] - 1; a <= c[0] + 1; a++) {
        if (a < 0 || a >= g->n[0]) continue;
        for (int b = c[1] - 1; b <= c[1] + 1; b++) {
            if (b < 0 || b >= g->n[1]) continue;
            for (int d = c[2] - 1; d <= c[2] + 1; d++) {
                if (d < 0 || d >= g->n[2]) continue;
                int bi = (a * g->n[1] + b) * g->n[2] + d;
                for (int t = g->start[bi]; t < g->start[bi + 1]; t++) {
                    if (n == *cap) { *cap = *cap ? *cap * 2 : 256; *buf = (int *)realloc(*buf, sizeof(int) * (size_t)*cap); }
                    (*buf)[n++] = g->items[t];
                }
            }
        }
    }
    (void)sc;
    qsort(*buf, (size_t)n, sizeof(int), cmp_int);
    return n;
}

static double dist3(const double *a, const double *b) {
    double d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    return sqrt(d0 * d0 + d1 * d1 + d2 * d2);
}

/* ------------------------------------------------------------ basis elements */
/* Cox-de Boor on the element's own knots t[0..4]; degree k basis starting at knot s */
static double bspl(const double *t, int s, int k, double x) {
    if (k == 0) return (t[s] <= x && x < t[s + 1]) ? 1.0 : 0.0;
    double a = 0.0, b = 0.0, d1 = t[s + k] - t[s], d2 = t[s + k + 1] - t[s + 1];
    if (d1 > 0) a = (x - t[s]) / d1 * bspl(t, s, k - 1, x);
    if (d2 > 0) b = (t[s + k + 1] - x) / d2 * bspl(t, s + 1, k - 1, x);
    return a + b;
}
/* value (nu=0) or first derivative (nu=1) of the cubic element on knots t[0..4]; 0 outside [t0,t4] */
static double basis_element(const double *t, double x, int nu) {
    if (!(x >= t[0] && x <= t[4])) return 0.0;   /* extrapolate=False -> NaN -> 0 */
    if (nu == 0) return bspl(t, 0, 3, x);
    double a = 0.0, b = 0.0, d1 = t[3] - t[0], d2 = t[4] - t[1];
    if (d1 > 0) a = 3.0 / d1 * bspl(t, 0, 2, x);
    if (d2 > 0) b = 3.0 / d2 * bspl(t, 1, 2, x);
    return a - b;
}
/* np.searchsorted(knots, x, side='left') */
static int searchsorted_left(const double *t, int n, double x) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) / 2; if (t[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

static int species_index(const uf3o_spec *s, int z) {
    for (int i = 0; i < s->n_species; i++) if (s->species_z[i] == z) return i;
    return -1;
}
static int pair_index(const uf3o_spec *s, int za, int zb) {
    if (za > zb) { int t = za; za = zb; zb = t; }
    for (int p = 0; p < s->n_pairs; p++) if (s->pair_z[2 * p] == za && s->pair_z[2 * p + 1] == zb) return p;
    return -1;
}
static int trio_index(const uf3o_spec *s, int zc, int za, int zb) {
    for (int t = 0; t < s->n_trios; t++)
        if (s->trio_z[3 * t] == zc && s->trio_z[3 * t + 1] == za && s->trio_z[3 * t + 2] == zb) return t;
    return -1;
}

/* ------------------------------------------------------------------ 2-body */
/*
 * Energy row: sum over real i, supercell j of B_b(d) for the pair block of (Z_i, Z_j),
 * max(r_min,0) < d < r_max strict (distances.py:60-69), bases [lead, nb-trail).
 * Force rows: x[m,c,b] = -sum_p B'_b(r_p) (delta_mj - delta_mi)(R_j-R_i)_c / r_p over directed
 * supercell pairs with a real end (distances.py:116-141, bspline.py:877-895).  A (ghost x, real y)
 * pair is the periodic image of the (real y', ghost x') pair seen from y, so both delta terms of
 * atom m are collected while visiting m's own neighbours.
 * Optional dump: per pair block, (i, j) supercell indices in np.where order.
 */
static void two_body(const uf3o_spec *s, const supercell_t *sc, const grid_t *g,
                     double *xe, double *xf, int64_t *pair_cnt, int64_t *pair_ij, int64_t pair_cap) {
    int *cand = NULL, cap = 0, N = sc->n_atoms, F = s->n_feat;
    const double **knots = (const double **)malloc(sizeof(double *) * (size_t)(s->n_pairs + 1));
    const double *kp = s->pair_knots;
    for (int p = 0; p < s->n_pairs; p++) { knots[p] = kp; kp += s->pair_nk[p]; }
    if (pair_cnt) memset(pair_cnt, 0, sizeof(int64_t) * (size_t)s->n_pairs);
    for (int i = 0; i < N; i++) {
        const double *ri = sc->pos + 3 * (size_t)i;
        int nc = gather_candidates(sc, g, ri, &cand, &cap);
        for (int c = 0; c < nc; c++) {
            int j = cand[c];
            int p = pair_index(s, sc->z[i], sc->z[j]);
            if (p < 0) continue;
            const double *rj = sc->pos + 3 * (size_t)j;
            double d = dist3(ri, rj);
            double rmin = s->pair_rmin[p] > 0 ? s->pair_rmin[p] : 0.0;
            if (!(d > rmin && d < s->pair_rmax[p])) continue;
            if (pair_cnt) {
                if (pair_ij && pair_cnt[p] < pair_cap) {
                    pair_ij[2 * ((size_t)p * pair_cap + pair_cnt[p])] = i;
                    pair_ij[2 * ((size_t)p * pair_cap + pair_cnt[p]) + 1] = j;
                }
                pair_cnt[p]++;
            }
            int nb = s->pair_nk[p] - 4;
            for (int b = s->lead2; b < nb - s->trail2; b++) {
                const double *t = knots[p] + b;
                if (xe) xe[s->pair_col[p] + b] += basis_element(t, d, 0);
                if (xf && d > t[0] && d < t[4]) {           /* strict support mask, bspline.py:884 */
                    double dv = basis_element(t, d, 1);
                    for (int k = 0; k < 3; k++) {
                        double cosk = (rj[k] - ri[k]) / d;
                        /* pair (i,j): m=i term  -B'*(-1)*cos ; mirrored pair (j',i): m=i term -B'*(+1)*(-cos) */
                        double *dst = xf + ((size_t)i * 3 + k) * F + s->pair_col[p] + b;
                        *dst += dv * cosk;
                        *dst += dv * cosk;
                    }
                }
            }
        }
    }
    free(cand); free((void *)knots);
}

/* ------------------------------------------------------------------ 3-body */
typedef struct {
    const double *k[3];
    int nk[3], dim[3];
    const int64_t *mask;
    const double *w;
    int *inv_start;   /* [L*M*N+1] raw bin -> range in inv_col */
    int *inv_col;     /* compressed columns fed by the raw bin (with multiplicity) */
} trio_view;

/* images of (l,m,n) summed by compress_3B (bspline.py:674-687) */
static int sym_images(int sym, int l, int m, int n, int M, int N, int64_t *out) {
#define U(a, b, c) (((int64_t)(a) * M + (b)) * N + (c))
    if (sym == 1) { out[0] = U(l, m, n); return 1; }
    if (sym == 2) { out[0] = U(l, m, n); out[1] = U(m, l, n); return 2; }
    out[0] = U(l, m, n); out[1] = U(l, n, m); out[2] = U(m, l, n);
    out[3] = U(n, l, m); out[4] = U(m, n, l); out[5] = U(n, m, l);
    return 6;
#undef U
}

static void free_trio_views(const uf3o_spec *s, trio_view *v) {
    for (int t = 0; t < s->n_trios; t++) { free(v[t].inv_start); free(v[t].inv_col); }
    free(v);
}

static void trio_views(const uf3o_spec *s, trio_view *v, double *rmin3, double *rmax3) {
    const double *kp = s->trio_knots;
    const int64_t *mp = s->trio_mask;
    const double *wp = s->trio_w;
    double lo = 1e300, hi = -1e300;
    for (int t = 0; t < s->n_trios; t++) {
        for (int d = 0; d < 3; d++) {
            v[t].k[d] = kp; v[t].nk[d] = s->trio_nk[3 * t + d]; v[t].dim[d] = v[t].nk[d] - 4;
            for (int q = 0; q < v[t].nk[d]; q++) {
                if (kp[q] < lo) lo = kp[q];
                if (d < 2 && kp[q] > hi) hi = kp[q];   /* first two legs only, angles.py:322-325 */
            }
            kp += v[t].nk[d];
        }
        v[t].mask = mp; v[t].w = wp;
        /* invert "column c = sum of grid over the symmetry images of its bin" */
        int M = v[t].dim[1], N = v[t].dim[2];
        size_t sz = (size_t)v[t].dim[0] * M * N;
        v[t].inv_start = (int *)calloc(sz + 1, sizeof(int));
        v[t].inv_col = (int *)malloc(sizeof(int) * 6 * (size_t)(s->trio_ncol[t] + 1));
        for (int rep = 0; rep < 2; rep++) {
            int *fill = rep ? (int *)malloc(sizeof(int) * (sz + 1)) : NULL;
            if (rep) { for (size_t u = sz; u > 0; u--) v[t].inv_start[u] = v[t].inv_start[u - 1]; v[t].inv_start[0] = 0;
                       for (size_t u = 0; u < sz; u++) v[t].inv_start[u + 1] += v[t].inv_start[u];
                       memcpy(fill, v[t].inv_start, sizeof(int) * (sz + 1)); }
            for (int c = 0; c < s->trio_ncol[t]; c++) {
                int64_t u = mp[c], img[6];
                int l = (int)(u / ((int64_t)M * N)), m = (int)((u / N) % M), n = (int)(u % N);
                int ni = sym_images(s->trio_sym[t], l, m, n, M, N, img);
                for (int q = 0; q < ni; q++) {
                    if (!rep) v[t].inv_start[img[q]]++;
                    else v[t].inv_col[fill[img[q]]++] = c;
                }
            }
            free(fill);
        }
        mp += s->trio_ncol[t]; wp += s->trio_ncol[t];
    }
    *rmin3 = lo > 0 ? lo : 0.0;
    *rmax3 = hi;
}

/* values (and derivatives) of the four candidate bases of leg `d` at r (angles.py:545-566, 607-630) */
static int leg_values(const trio_view *v, int d, double r, int lead, int trail, double *val, double *der) {
    int first = searchsorted_left(v->k[d], v->nk[d], r) - 4;
    for (int a = 0; a < 4; a++) {
        int b = first + a;
        val[a] = 0.0; if (der) der[a] = 0.0;
        if (b < lead || b >= v->dim[d] - trail) continue;   /* trimmed, or wrapped negative index */
        val[a] = basis_element(v->k[d] + b, r, 0);
        if (der) der[a] = basis_element(v->k[d] + b, r, 1);
    }
    return first;
}

/* fold the symmetry images, keep template bins, weight (bspline.py:664-690), accumulate into out */
static void compress_add(const uf3o_spec *s, const trio_view *v, int t, const double *grid, double *out, double sign) {
    int M = v->dim[1], N = v->dim[2], sym = s->trio_sym[t];
    for (int c = 0; c < s->trio_ncol[t]; c++) {
        int64_t u = v->mask[c];
        int l = (int)(u / ((int64_t)M * N)), m = (int)((u / N) % M), n = (int)(u % N);
        int64_t img[6];
        int ni = sym_images(sym, l, m, n, M, N, img);
        double val = 0.0;
        for (int q = 0; q < ni; q++) val += grid[img[q]];
        out[s->trio_col[t] + c] += sign * val * v->w[c];
    }
}

static int wrap_idx(int i, int n) { return i < 0 ? i + n : i; }  /* numpy negative indexing */

/*
 * 3-body energy grid and (optionally) per-atom force grids.
 * Energy: real centres only (angles.py:339).  Forces: every supercell atom may be a centre; a ghost
 * centre keeps only triplets whose first-listed neighbour is real (angles.py:451-460).
 * Neighbour pairs: r_min3 < d <= r_max3.  j<k by supercell index, then the two neighbours are put in
 * ascending-Z order (stable), interaction = (Z_i; Z_j, Z_k); legs masked t[0] <= r <= t[-1].
 */
static void three_body(const uf3o_spec *s, const supercell_t *sc, const grid_t *g,
                       double *xe, double *xf, int64_t *n3_cnt, int64_t *n3_ij, int64_t n3_cap) {
    int T = s->n_trios, N = sc->n_atoms, F = s->n_feat;
    trio_view *v = (trio_view *)malloc(sizeof(trio_view) * (size_t)T);
    double rmin3, rmax3;
    trio_views(s, v, &rmin3, &rmax3);
    double **egrid = (double **)malloc(sizeof(double *) * (size_t)T);
    for (int t = 0; t < T; t++) egrid[t] = (double *)calloc((size_t)v[t].dim[0] * v[t].dim[1] * v[t].dim[2], sizeof(double));
    int *cand = NULL, cap = 0, *nbr = NULL, nbr_cap = 0;
    double *nbr_d = NULL;
    int64_t cnt = 0;

    int n_centres = xf ? sc->m : N;
    for (int pass = 0; pass < (xf ? 2 : 1); pass++) {
        /* pass 0: energy over real centres; pass 1: forces over all centres */
        int nc_max = pass == 0 ? N : n_centres;
        if (pass == 0 && !xe && !n3_cnt) continue;
        for (int i = 0; i < nc_max; i++) {
            const double *ri = sc->pos + 3 * (size_t)i;
            int nc = gather_candidates(sc, g, ri, &cand, &cap), nn = 0;
            for (int c = 0; c < nc; c++) {
                double d = dist3(ri, sc->pos + 3 * (size_t)cand[c]);
                if (d > rmin3 && d <= rmax3) {
                    if (nn == nbr_cap) {
                        nbr_cap = nbr_cap ? 2 * nbr_cap : 64;
                        nbr = (int *)realloc(nbr, sizeof(int) * (size_t)nbr_cap);
                        nbr_d = (double *)realloc(nbr_d, sizeof(double) * (size_t)nbr_cap);
                    }
                    nbr[nn] = cand[c]; nbr_d[nn] = d; nn++;
                    if (pass == 0 && n3_cnt) {
                        if (n3_ij && cnt < n3_cap) { n3_ij[2 * cnt] = i; n3_ij[2 * cnt + 1] = cand[c]; }
                        cnt++;
                    }
                }
            }
            if (pass == 0 && !xe) continue;
            int ghost = i >= N;
            for (int a = 0; a < nn; a++) {
                for (int b = 0; b < nn; b++) {
                    /* meshgrid(i_group_filtered, i_group): j from the (real-only, for ghost centres)
                       list, k from the full list, keep j < k */
                    int j = nbr[a], k = nbr[b];
                    if (!(j < k)) continue;
                    if (ghost && j >= N) continue;
                    double rij = nbr_d[a], rik = nbr_d[b];
                    int zj = sc->z[j], zk = sc->z[k];
                    if (zj > zk) { int ti = j; j = k; k = ti; ti = zj; zj = zk; zk = ti; double td = rij; rij = rik; rik = td; }
                    int t = trio_index(s, sc->z[i], zj, zk);
                    if (t < 0) continue;
                    const double *rj = sc->pos + 3 * (size_t)j, *rk = sc->pos + 3 * (size_t)k;
                    double rjk = dist3(rj, rk);
                    const trio_view *tv = v + t;
                    if (!(rij >= tv->k[0][0] && rij <= tv->k[0][tv->nk[0] - 1])) continue;
                    if (!(rik >= tv->k[1][0] && rik <= tv->k[1][tv->nk[1] - 1])) continue;
                    if (!(rjk >= tv->k[2][0] && rjk <= tv->k[2][tv->nk[2] - 1])) continue;
                    double vl[4], vm[4], vn[4], dl[4], dm[4], dn[4];
                    int L = tv->dim[0], M = tv->dim[1], Nn = tv->dim[2];
                    int il = leg_values(tv, 0, rij, s->lead3, s->trail3, vl, pass ? dl : NULL);
                    int im = leg_values(tv, 1, rik, s->lead3, s->trail3, vm, pass ? dm : NULL);
                    int in = leg_values(tv, 2, rjk, s->lead3, s->trail3, vn, pass ? dn : NULL);
                    if (pass == 0) {
                        for (int x = 0; x < 4; x++) for (int y = 0; y < 4; y++) for (int w = 0; w < 4; w++) {
                            double val = vl[x] * vm[y] * vn[w];
                            if (val == 0.0) continue;
                            egrid[t][((size_t)wrap_idx(il + x, L) * M + wrap_idx(im + y, M)) * Nn + wrap_idx(in + w, Nn)] += val;
                        }
                    } else {
                        /* direction cosines restricted to real atoms (distances.py:354-363) */
                        int idx[3] = {i, j, k};
                        for (int who = 0; who < 3; who++) {
                            int m = idx[who];
                            if (m >= N) continue;
                            /* an atom may appear once only in a triplet, but two of i,j,k can be images
                               of one real atom only as distinct supercell indices, so m matches one slot */
                            for (int c3 = 0; c3 < 3; c3++) {
                                double dij = ((m == j) - (m == i)) * (rj[c3] - ri[c3]) / rij;
                                double dik = ((m == k) - (m == i)) * (rk[c3] - ri[c3]) / rik;
                                double djk = ((m == k) - (m == j)) * (rk[c3] - rj[c3]) / rjk;
                                if (dij == 0 && dik == 0 && djk == 0) continue;
                                /* force_grid[m][c] -= val, then compress_3B: a raw bin u reaches every
                                   column whose symmetry images contain u, times that column's weight */
                                double *row = xf + ((size_t)m * 3 + c3) * F + s->trio_col[t];
                                for (int x = 0; x < 4; x++) for (int y = 0; y < 4; y++) for (int w = 0; w < 4; w++) {
                                    double val = dl[x] * vm[y] * vn[w] * dij + vl[x] * dm[y] * vn[w] * dik + vl[x] * vm[y] * dn[w] * djk;
                                    if (val == 0.0) continue;
                                    size_t u = ((size_t)wrap_idx(il + x, L) * M + wrap_idx(im + y, M)) * Nn + wrap_idx(in + w, Nn);
                                    for (int e = tv->inv_start[u]; e < tv->inv_start[u + 1]; e++)
                                        row[tv->inv_col[e]] -= val * tv->w[tv->inv_col[e]];
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if (xe) for (int t = 0; t < T; t++) compress_add(s, v + t, t, egrid[t], xe, 1.0);
    if (n3_cnt) *n3_cnt = cnt;
    for (int t = 0; t < T; t++) free(egrid[t]);
    free(egrid); free(cand); free(nbr); free(nbr_d); free_trio_views(s, v);
}

/* ------------------------------------------------------------------- entry */
/*
 * Feature rows of one frame (process.py:293-367 without the y column):
 *   xe [F]        1-body counts | 2-body | 3-body         (NULL to skip)
 *   xf [N][3][F]  1-body zeros  | 2-body | 3-body         (NULL to skip)
 * Index dumps (any may be NULL):
 *   pair_cnt [P], pair_ij [P][pair_cap][2]   2-body (i, j) per pair block
 *   n3_cnt [1],  n3_ij [n3_cap][2]           identify_ij(square=False) pairs
 *   sc_info [8]: n_img, m, fac[3], cnt[3]
 */
int uf3o_featurize(const uf3o_spec *s, const uf3o_frame *f, double *xe, double *xf,
                   int64_t *pair_cnt, int64_t *pair_ij, int64_t pair_cap,
                   int64_t *n3_cnt, int64_t *n3_ij, int64_t n3_cap, int64_t *sc_info) {
    supercell_t sc;
    grid_t g;
    for (int a = 0; a < f->n_atoms; a++) if (species_index(s, f->z[a]) < 0) return 2;
    build_supercell(f, s->r_cut, &sc);
    double rmin3 = 0, rmax3 = 0, h = 0;
    for (int p = 0; p < s->n_pairs; p++) if (s->pair_rmax[p] > h) h = s->pair_rmax[p];
    if (s->n_trios > 0) {
        trio_view *v = (trio_view *)malloc(sizeof(trio_view) * (size_t)s->n_trios);
        trio_views(s, v, &rmin3, &rmax3);
        free_trio_views(s, v);
        if (rmax3 > h) h = rmax3;
    }
    build_grid(&sc, h * (1.0 + 1e-9) + 1e-9, &g);
    size_t F = (size_t)s->n_feat;
    if (xe) {
        memset(xe, 0, sizeof(double) * F);
        for (int a = 0; a < f->n_atoms; a++) xe[species_index(s, f->z[a])] += 1.0;
    }
    if (xf) memset(xf, 0, sizeof(double) * F * 3 * (size_t)f->n_atoms);
    two_body(s, &sc, &g, xe, xf, pair_cnt, pair_ij, pair_cap);
    if (s->n_trios > 0) three_body(s, &sc, &g, xe, xf, n3_cnt, n3_ij, n3_cap);
    else if (n3_cnt) *n3_cnt = 0;
    if (sc_info) {
        sc_info[0] = sc.n_img; sc_info[1] = sc.m;
        for (int d = 0; d < 3; d++) { sc_info[2 + d] = sc.fac[d]; sc_info[5 + d] = sc.cnt[d]; }
    }
    free_grid(&g); free_supercell(&sc);
    return 0;
}

/* supercell positions / species in reference order; returns m (call with out=NULL to size) */
int64_t uf3o_supercell(const uf3o_frame *f, double r_cut, double *pos_out, int *z_out, int *shift_out) {
    supercell_t sc;
    build_supercell(f, r_cut, &sc);
    int64_t m = sc.m;
    if (pos_out) memcpy(pos_out, sc.pos, sizeof(double) * 3 * (size_t)sc.m);
    if (z_out) memcpy(z_out, sc.z, sizeof(int) * (size_t)sc.m);
    if (shift_out) memcpy(shift_out, sc.shift, sizeof(int) * 3 * (size_t)sc.n_img);
    free_supercell(&sc);
    return m;
}

/* ------------------------------------------------------------- evaluator */
/* tensor-product cubic spline value / first partials from a full coefficient grid */
static void spline3(const trio_view *tv, const double *c, double rl, double rm, double rn, double *val, double *grad) {
    double vl[4], vm[4], vn[4], dl[4], dm[4], dn[4];
    int L = tv->dim[0], M = tv->dim[1], N = tv->dim[2];
    int il = leg_values(tv, 0, rl, 0, 0, vl, dl), im = leg_values(tv, 1, rm, 0, 0, vm, dm), in = leg_values(tv, 2, rn, 0, 0, vn, dn);
    double v = 0, g0 = 0, g1 = 0, g2 = 0;
    for (int x = 0; x < 4; x++) for (int y = 0; y < 4; y++) for (int w = 0; w < 4; w++) {
        int a = il + x, b = im + y, d = in + w;
        if (a < 0 || b < 0 || d < 0 || a >= L || b >= M || d >= N) continue;
        double cc = c[((size_t)a * M + b) * N + d];
        v += cc * vl[x] * vm[y] * vn[w];
        g0 += cc * dl[x] * vm[y] * vn[w];
        g1 += cc * vl[x] * dm[y] * vn[w];
        g2 += cc * vl[x] * vm[y] * dn[w];
    }
    *val = v; grad[0] = g0; grad[1] = g1; grad[2] = g2;
}

/*
 * Energy and forces of a fitted model (calculator.py:156-343).
 *   c1 [S] one-body, c2 concatenated pair coefficient vectors (nk-4 each, ALL bases),
 *   c3 concatenated full L*M*N grids (decompress_3B output) per trio.
 */
int uf3o_eval(const uf3o_spec *s, const uf3o_frame *f, const double *c1, const double *c2, const double *c3,
              double *energy, double *forces) {
    supercell_t sc;
    grid_t g;
    for (int a = 0; a < f->n_atoms; a++) if (species_index(s, f->z[a]) < 0) return 2;
    build_supercell(f, s->r_cut, &sc);
    int T = s->n_trios, N = sc.n_atoms;
    trio_view *v = T ? (trio_view *)malloc(sizeof(trio_view) * (size_t)T) : NULL;
    double rmin3 = 0, rmax3 = 0, h = 0;
    for (int p = 0; p < s->n_pairs; p++) if (s->pair_rmax[p] > h) h = s->pair_rmax[p];
    if (T) { trio_views(s, v, &rmin3, &rmax3); if (rmax3 > h) h = rmax3; }
    build_grid(&sc, h * (1.0 + 1e-9) + 1e-9, &g);
    double e = 0.0;
    if (forces) memset(forces, 0, sizeof(double) * 3 * (size_t)N);
    for (int a = 0; a < N; a++) e += c1[species_index(s, f->z[a])];
    /* pair part */
    const double **pk = (const double **)malloc(sizeof(double *) * (size_t)(s->n_pairs + 1));
    const double **pc = (const double **)malloc(sizeof(double *) * (size_t)(s->n_pairs + 1));
    { const double *kp = s->pair_knots, *cp = c2;
      for (int p = 0; p < s->n_pairs; p++) { pk[p] = kp; pc[p] = cp; kp += s->pair_nk[p]; cp += s->pair_nk[p] - 4; } }
    int *cand = NULL, cap = 0;
    for (int i = 0; i < N; i++) {
        const double *ri = sc.pos + 3 * (size_t)i;
        int nc = gather_candidates(&sc, &g, ri, &cand, &cap);
        for (int c = 0; c < nc; c++) {
            int j = cand[c], p = pair_index(s, sc.z[i], sc.z[j]);
            if (p < 0) continue;
            const double *rj = sc.pos + 3 * (size_t)j;
            double d = dist3(ri, rj), rmin = s->pair_rmin[p] > 0 ? s->pair_rmin[p] : 0.0;
            if (!(d > rmin && d < s->pair_rmax[p])) continue;
            int nb = s->pair_nk[p] - 4, first = searchsorted_left(pk[p], s->pair_nk[p], d) - 4;
            double phi = 0, dphi = 0;
            for (int a = 0; a < 4; a++) {
                int b = first + a;
                if (b < 0 || b >= nb) continue;
                phi += pc[p][b] * basis_element(pk[p] + b, d, 0);
                dphi += pc[p][b] * basis_element(pk[p] + b, d, 1);
            }
            e += phi;
            if (forces) for (int k = 0; k < 3; k++) forces[3 * i + k] += 2.0 * dphi * (rj[k] - ri[k]) / d;
        }
    }
    /* trio part: real centres for the energy, all centres (ghost: real j) for the forces */
    if (T) {
        const double **tc = (const double **)malloc(sizeof(double *) * (size_t)T);
        { const double *cp = c3; for (int t = 0; t < T; t++) { tc[t] = cp; cp += (size_t)v[t].dim[0] * v[t].dim[1] * v[t].dim[2]; } }
        int *nbr = NULL, nbr_cap = 0; double *nbr_d = NULL;
        int n_centres = forces ? sc.m : N;
        for (int i = 0; i < n_centres; i++) {
            const double *ri = sc.pos + 3 * (size_t)i;
            int nc = gather_candidates(&sc, &g, ri, &cand, &cap), nn = 0;
            for (int c = 0; c < nc; c++) {
                double d = dist3(ri, sc.pos + 3 * (size_t)cand[c]);
                if (d > rmin3 && d <= rmax3) {
                    if (nn == nbr_cap) { nbr_cap = nbr_cap ? 2 * nbr_cap : 64; nbr = (int *)realloc(nbr, sizeof(int) * (size_t)nbr_cap); nbr_d = (double *)realloc(nbr_d, sizeof(double) * (size_t)nbr_cap); }
                    nbr[nn] = cand[c]; nbr_d[nn] = d; nn++;
                }
            }
            int ghost = i >= N;
            for (int a = 0; a < nn; a++) for (int b = 0; b < nn; b++) {
                int j = nbr[a], k = nbr[b];
                if (!(j < k)) continue;
                if (ghost && j >= N) continue;
                double rij = nbr_d[a], rik = nbr_d[b];
                int zj = sc.z[j], zk = sc.z[k];
                if (zj > zk) { int ti = j; j = k; k = ti; ti = zj; zj = zk; zk = ti; double td = rij; rij = rik; rik = td; }
                int t = trio_index(s, sc.z[i], zj, zk);
                if (t < 0) continue;
                const double *rj = sc.pos + 3 * (size_t)j, *rk = sc.pos + 3 * (size_t)k;
                double rjk = dist3(rj, rk);
                const trio_view *tv = v + t;
                if (!(rij >= tv->k[0][0] && rij <= tv->k[0][tv->nk[0] - 1])) continue;
                if (!(rik >= tv->k[1][0] && rik <= tv->k[1][tv->nk[1] - 1])) continue;
                if (!(rjk >= tv->k[2][0] && rjk <= tv->k[2][tv->nk[2] - 1])) continue;
                double val, gr[3];
                spline3(tv, tc[t], rij, rik, rjk, &val, gr);
                if (!ghost) e += val;
                if (forces) {
                    int idx[3] = {i, j, k};
                    for (int who = 0; who < 3; who++) {
                        int m = idx[who];
                        if (m >= N) continue;
                        for (int c3 = 0; c3 < 3; c3++) {
                            double dij = ((m == j) - (m == i)) * (rj[c3] - ri[c3]) / rij;
                            double dik = ((m == k) - (m == i)) * (rk[c3] - ri[c3]) / rik;
                            double djk = ((m == k) - (m == j)) * (rk[c3] - rj[c3]) / rjk;
                            forces[3 * m + c3] -= dij * gr[0] + dik * gr[1] + djk * gr[2];
                        }
                    }
                }
            }
        }
        free(nbr); free(nbr_d); free((void *)tc);
    }
    if (energy) *energy = e;
    free(cand); free((void *)pk); free((void *)pc); if (v) free_trio_views(s, v);
    free_grid(&g); free_supercell(&sc);
    return 0;
}
