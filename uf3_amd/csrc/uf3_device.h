// uf3_device.h -- device-side data layout and the small device functions shared by
// the kernels (cubic B-spline evaluation from per-interval records, periodic cell
// list traversal, wave64 helpers).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define UF3_MAX_SPECIES 8
#define UF3_MAX_PAIRS 36
#define WAVE 64

// One record per knot interval i (t_i < x <= t_{i+1}): the six knots and the six
// reciprocal knot differences the de Boor-Cox triangle needs.  96 bytes, 16-B aligned.
struct KnotRec {
    double t[6];   // t[i-2] .. t[i+3]
    double r[6];   // 1/(t[i+1]-t[i]); 1/(t[i+1]-t[i-1]), 1/(t[i+2]-t[i]);
                   // 1/(t[i+1]-t[i-2]), 1/(t[i+2]-t[i-1]), 1/(t[i+3]-t[i])
};

struct LegDev {
    int rec_off;       // first KnotRec of this knot vector (index = interval number)
    int nk;            // number of knots; valid intervals are 3 .. nk-5
    double t0, tlast;  // support of the leg
    double inv_h;      // (nk-7)/(tlast-t0): interval guess, exact for uniform knots
};

struct PairDev {
    LegDev leg;
    int col;           // first column of the block
    int nb;            // basis functions (nk-4)
    int sa, sb;        // species indices, sa <= sb
    double rmin, rmax; // strict range: max(r_min,0) < d < r_max
    double s_lo, s_hi; // the same test on the SQUARED distance s as the kernels form it: rmin < sqrt(s) < rmax  <=>  s_lo < s < s_hi
                       // (sqrt correctly rounded, hence monotonic: s_lo = the largest double whose root is <= rmin, s_hi the
                       // smallest whose root is >= rmax; found by uf3_basis_create)
};

// what the per-(atom, trio) dispatch of k_featurize looks at, in one 32-byte block at the head of TrioDev: ONE scalar load per
// trio and atom (read field by field, each behind its own branch, it was eight dependent scalar round trips)
// grouped: 0, or (layout + 1) | offset of the block's fold table in FeatArgs::gsrc << 8 (grouped windows), or
// 1 | first column tiles of the three bands << 8, 4 bits each (banded windows)
struct TrioHead { int dense, nsrc, ncol, sc, sa, sb, col, grouped; };

struct TrioDev {
    TrioHead head;     // copies of the fields below (filled by uf3_basis_create once they are final)
    LegDev leg[3];     // l (ij), m (ik), n (jk)
    int col, ncol;
    int lut_off;       // into BasisDev::lut (also offset of the full grid in c3)
    int dim_m, dim_n;  // M, N of the L x M x N grid
    int dim_l;
    int sc, sa, sb;    // species indices: centre, neighbours sa <= sb
    int nsrc, src_off; // symmetry images per column (1, 2, 6) and offset into the colsrc table
    // dense accumulation window = bounding box of the raw bins that feed a column.  dense = featurizer mode (6-9) of
    // the matrix-core specialisation that serves the box, 0 if none does
    int dense, lo[3], ext[3];
    // two column tiles (n-major columns): a record with r_n <= thr0 touches tile 0 only, with r_n > thr2 tile 1 only
    // (knot values of leg n, so the classes are exact for any knot sequence)
    double thr0, thr2;
    // grouped != 0 (3 x 3 x <=9 windows): the n bins are covered by three overlapping groups of five, [0,4] [2,6] [4,8];
    // a record's four n bins lie inside ONE group -- r_n <= gthr0: group 0, r_n > gthr2: group 2, else group 1 -- so
    // every step is a single MFMA into its group's accumulator tile (columns (n - group base, m): 15 of 16)
    double gthr0, gthr2;
    int grouped;
    int banded;        // MODE 9 force launches: the leg-n intervals fall into <= 3 bands of <= 3 column tiles each (gthr0 / gthr2
    int band_tile[3];  // separate them, band_tile[b] = first column tile of band b); TrioHead::grouped = 1 | tiles << 8
    int layout;        // grouped windows: number of this trio's window layout (legs' knot sequences, window box, thresholds)
    int gsrc_off;      // ... and where its fold table starts in FeatArgs::gsrc
    int wrow[3];       // ... and, per leg, the number of its first window row (one row per knot interval from 3 on, see uf3_basis_create)
};

struct BasisDev {
    int S, P, T, F;
    int lead2, trail2;
    double rmin3, rmax3;   // 3-body neighbour range: rmin3 < d <= rmax3
    double s3_lo, s3_hi;   // ... on the squared distance: s3_lo < s <= s3_hi (largest doubles whose roots are <= rmin3 / rmax3)
    double rmax2;          // largest pair r_max
    double rsearch;        // cell-list radius = max(rmax2, rmax3) (= BSplineBasis.r_cut)
    int trio_legs_uniform; // every trio has the legs (knot sequences) and grid dimensions of trio 0: the evaluator reads them once,
                           // through scalar loads, instead of per lane and triplet
    int pairs_uniform;     // every pair block has the knots, range and size of pair 0 (only its first column differs: pair_col)
    int pair_col[UF3_MAX_SPECIES * UF3_MAX_SPECIES];   // first column of the pair block of two species
    signed char z2s[120];  // atomic number -> species index or -1
    short pair_of[UF3_MAX_SPECIES * UF3_MAX_SPECIES];
    short trio_of[UF3_MAX_SPECIES * UF3_MAX_SPECIES * UF3_MAX_SPECIES];  // [centre][a][b], a<=b
    PairDev pairs[UF3_MAX_PAIRS];
    const TrioDev *trios;
    const KnotRec *recs;
    const int *lut;        // raw bin -> global column, or -1
};

// Per-frame geometry for the periodic cell list and the reference's ghost numbering.
struct FrameGeom {
    double cell[9];     // real lattice rows (image offsets)
    double inv[9];      // inverse of the effective (completed) cell: frac_k = sum_j x_j inv[3j+k]
    double binw[3];     // non-periodic axes: bin width in fractional units
    int nb[3];          // bins per axis
    int rad[3];         // search radius in bins (periodic) ; unused for non-periodic
    int per[3];         // periodic flag
    int fac[3];         // reference image range per axis (0 when not periodic)
    int cnt[3];         // images per axis in the reference supercell (2*fac+1 or 1)
    double win_lo[3], win_hi[3];   // fractional window around the cell: while every atom of the frame lies inside it, any two
                                   // atoms within r_cut of each other are images -fac .. fac of one another (TrioWalk::img_check)
    int bin_base;       // first global bin
    int atom_lo, atom_hi;
};

// one atom in bin order (32 B, one load): original position, batch-global index, packed wrap vector + species
struct __attribute__((aligned(16))) SlotRec {
    double x, y, z;
    int atom;
    int ws;                   // pack3(wrap) | species << 30 would not fit: wrap 3 x 9 bits, species bits 27..30
};
__device__ __forceinline__ int pack_ws(int w0, int w1, int w2, int spec) {
    return (w0 + 256) | ((w1 + 256) << 9) | ((w2 + 256) << 18) | (spec << 27);
}
__device__ __forceinline__ void unpack_ws(int p, int &w0, int &w1, int &w2, int &spec) {
    w0 = (p & 511) - 256; w1 = ((p >> 9) & 511) - 256; w2 = ((p >> 18) & 511) - 256; spec = (p >> 27) & 15;
}

struct CellList {
    const int *bin_start;     // [nbins_total+1] slots
    const SlotRec *slots;     // [natoms] atoms in bin order
    const int *atom_bin;      // [natoms] local bin id (within frame) per atom
    const int *atom_wrap;     // [natoms] packed wrap per atom
};

// one 3-body neighbour (48 B, 16-B aligned): vector from the centre to this image, its length, and the
// identifiers the triplet logic needs; lists are padded to `cap` entries per atom and sorted by
// (species, reference supercell index)
struct __attribute__((aligned(16))) N3Entry {
    double dx, dy, dz, r;
    int parent;     // batch-global atom index of the neighbour
    int shiftc;     // packed image shift relative to the centre's home cell
    int sidx;       // reference supercell index of the neighbour (seen from a real centre)
    int spec;       // species index
};

struct N3Lists {
    int cap;
    int *cnt;       // [natoms]
    int *spoff;     // [natoms][UF3_MAX_SPECIES+1] first entry of each species in the list
    N3Entry *ent;   // [natoms*cap]
    // extension (batches with atoms outside their cell only, k_build_n3_ext): neighbours whose image index lies beyond
    // the reference's range as seen from the atom, but inside it as seen from one of the atom's ghost images
    int xcap;
    int *xoff;      // [natoms][UF3_MAX_SPECIES+1]
    N3Entry *xent;  // [natoms*xcap]
};

__device__ __forceinline__ int pack3(int a, int b, int c) { return (a + 512) | ((b + 512) << 10) | ((c + 512) << 20); }
__device__ __forceinline__ void unpack3(int p, int &a, int &b, int &c) {
    a = (p & 1023) - 512; b = ((p >> 10) & 1023) - 512; c = ((p >> 20) & 1023) - 512;
}

// position of image index s in the reference's per-axis list 0, +1, -1, +2, -2, ...
__device__ __forceinline__ int image_pos(int s) { return s == 0 ? 0 : (s > 0 ? 2 * s - 1 : -2 * s); }

__device__ __forceinline__ int supercell_index(const FrameGeom &g, int s0, int s1, int s2, int local_atom) {
    int rank = (image_pos(s1) * g.cnt[0] + image_pos(s0)) * g.cnt[2] + image_pos(s2);
    return rank * (g.atom_hi - g.atom_lo) + local_atom;
}

// ---- cubic B-spline: interval search + de Boor-Cox triangle ---------------------
// interval i with t_i < x <= t_{i+1}, 3 <= i <= nk-5.  Caller guarantees t0 < x <= tlast.
__device__ __forceinline__ int find_interval(const KnotRec *recs, const LegDev &leg, double x) {
    int hi = leg.nk - 5;
    int i = 3 + (int)((x - leg.t0) * leg.inv_h);
    i = i < 3 ? 3 : (i > hi ? hi : i);
    const KnotRec *base = recs + leg.rec_off;
    for (;;) {
        double ti = base[i].t[2], ti1 = base[i].t[3];
        if (x > ti1 && i < hi) ++i;
        else if (x <= ti && i > 3) --i;
        else break;
    }
    return i;
}

// Same search, returning the interval's record as well: the whole record at the initial guess is fetched at once
// and only re-fetched when the guess was off (non-uniform knots, or x on an interval boundary), so the common
// case costs one memory round trip instead of two.
// AS: where the records are -- 3: the workgroup's LDS copy, 1: global memory, 0: not said (a generic pointer).  Said explicitly
// wherever the caller knows: through a generic pointer (a member of the kernel's argument struct) these become flat loads,
// slower than DS reads, and their results can only be waited for together with everything else in flight on both counters.
typedef double __attribute__((ext_vector_type(2))) KnotPair;
template <int AS> struct KnotPairs { typedef const KnotPair *type; };
template <> struct KnotPairs<1> { typedef const __attribute__((address_space(1))) KnotPair *type; };
template <> struct KnotPairs<3> { typedef const __attribute__((address_space(3))) KnotPair *type; };

template <int AS = 0>
__device__ __forceinline__ void load_knot_rec(const KnotRec *p, KnotRec &k) {
    // six 16-byte reads issued back to back (the compiler otherwise fetches t[3], tests it, and only then the rest)
    typename KnotPairs<AS>::type q = (typename KnotPairs<AS>::type)(const KnotPair *)p;
    const KnotPair v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4], v5 = q[5];
    __builtin_amdgcn_sched_group_barrier(0x120, 6, 0);             // DS or VMEM reads
    k.t[0] = v0.x; k.t[1] = v0.y; k.t[2] = v1.x; k.t[3] = v1.y; k.t[4] = v2.x; k.t[5] = v2.y;
    k.r[0] = v3.x; k.r[1] = v3.y; k.r[2] = v4.x; k.r[3] = v4.y; k.r[4] = v5.x; k.r[5] = v5.y;
}

template <int AS = 0>
__device__ __forceinline__ int load_interval(const KnotRec *recs, const LegDev &leg, double x, KnotRec &k) {
    const int hi = leg.nk - 5;
    int i = 3 + (int)((x - leg.t0) * leg.inv_h);
    i = i < 3 ? 3 : (i > hi ? hi : i);
    const KnotRec *base = recs + leg.rec_off;
    load_knot_rec<AS>(base + i, k);
    if (__builtin_expect(x > k.t[3] && i < hi, 0)) {
        do { ++i; load_knot_rec<AS>(base + i, k); } while (x > k.t[3] && i < hi);
    } else if (__builtin_expect(x <= k.t[2] && i > 3, 0)) {
        do { --i; load_knot_rec<AS>(base + i, k); } while (x <= k.t[2] && i > 3);
    }
    return i;
}

// The two halves of load_interval, for callers that look up several legs at once: every leg's record at its guessed
// interval goes out first (load_interval_guess), the rare corrections follow (load_interval_fix) -- three dependent
// memory round trips become one.
template <int AS = 0>
__device__ __forceinline__ int load_interval_guess(const KnotRec *recs, const LegDev &leg, double x, KnotRec &k) {
    const int hi = leg.nk - 5;
    int i = 3 + (int)((x - leg.t0) * leg.inv_h);
    i = i < 3 ? 3 : (i > hi ? hi : i);
    load_knot_rec<AS>(recs + leg.rec_off + i, k);
    return i;
}
template <int AS = 0>
__device__ __forceinline__ int load_interval_fix(const KnotRec *recs, const LegDev &leg, double x, int i, KnotRec &k) {
    const int hi = leg.nk - 5;
    const KnotRec *base = recs + leg.rec_off;
    if (__builtin_expect(x > k.t[3] && i < hi, 0)) {
        do { ++i; load_knot_rec<AS>(base + i, k); } while (x > k.t[3] && i < hi);
    } else if (__builtin_expect(x <= k.t[2] && i > 3, 0)) {
        do { --i; load_knot_rec<AS>(base + i, k); } while (x <= k.t[2] && i > 3);
    }
    return i;
}

// values v[0..3] and first derivatives d[0..3] of basis functions i-3 .. i at x
template <bool DERIV>
__device__ __forceinline__ void bspline4(const KnotRec &k, double x, double *v, double *d) {
    double l1 = x - k.t[2], l2 = x - k.t[1], l3 = x - k.t[0];
    double r1 = k.t[3] - x, r2 = k.t[4] - x, r3 = k.t[5] - x;
    // degree 1
    double tmp = k.r[0];
    double n0 = r1 * tmp, n1 = l1 * tmp;
    // degree 2
    tmp = n0 * k.r[1];
    double q0 = r1 * tmp, saved = l2 * tmp;
    tmp = n1 * k.r[2];
    double q1 = saved + r2 * tmp;
    double q2 = l1 * tmp;
    // degree 3
    tmp = q0 * k.r[3];
    v[0] = r1 * tmp; saved = l3 * tmp;
    tmp = q1 * k.r[4];
    v[1] = saved + r2 * tmp; saved = l2 * tmp;
    tmp = q2 * k.r[5];
    v[2] = saved + r3 * tmp;
    v[3] = l1 * tmp;
    if (DERIV) {
        double a = 3.0 * q0 * k.r[3], b = 3.0 * q1 * k.r[4], c = 3.0 * q2 * k.r[5];
        d[0] = -a; d[1] = a - b; d[2] = b - c; d[3] = c;
    }
}

// unfused |d|: ((dx*dx + dy*dy) + dz*dz), the order scipy's cdist uses, so that range
// comparisons at the cut-offs agree with the reference to the last bit.  (`__dmul_rn` / `__dadd_rn` are plain `*` / `+` in
// this toolchain's headers and the library is built with -ffp-contract=fast: until round 5 the compiler fused these into
// fma(dz, dz, fma(dx, dx, dy * dy)) -- one ulp off scipy's sum in about one distance in ten, found when the distance VALUES
// were first compared bit for bit (tests/test_module_surfaces.py).  `#pragma clang fp contract(off)` does not help under
// -ffp-contract=fast, where the backend fuses whatever it meets; the empty asm makes the three squares opaque to it.)
__device__ __forceinline__ double norm3_sq_rn(double dx, double dy, double dz) {        // the radicand of norm3_rn
    double xx = dx * dx, yy = dy * dy, zz = dz * dz;
    asm("" : "+v"(xx), "+v"(yy), "+v"(zz));
    const double xy = xx + yy;
    return xy + zz;
}
__device__ __forceinline__ double norm3_rn(double dx, double dy, double dz) {
    return sqrt(norm3_sq_rn(dx, dy, dz));
}

// |d| for the THIRD leg of a triplet (r_jk): not part of any neighbour-index decision -- it selects knot intervals, where a
// cubic B-spline is continuous, and is compared with the leg's ends, where the basis functions vanish -- so the last bit does
// not matter and the root comes from the hardware estimate + one Newton step + one correction (8 instructions; the IEEE
// library root with its scaling for subnormal arguments is 20, and the walk takes one per triplet role).
__device__ __forceinline__ double norm3_leg(double dx, double dy, double dz) {
    const double s = fma(dz, dz, fma(dy, dy, dx * dx));
    const double y = __builtin_amdgcn_rsq(s);
    double g = s * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    g = fma(fma(-g, g, s), h, g);
    return s > 0.0 ? g : 0.0;
}

// 1 / x for a distance: the hardware estimate + two Newton steps (about one ulp; 9 instructions where the IEEE quotient, with its
// scaling for extreme arguments, is ~30).  Used where a derivative is turned into a force along a bond: nothing there is
// compared or indexed, the tolerance on rows / forces is relative 1e-9.
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// inclusive prefix sum over the 64 lanes on the DPP network (row shifts, then row broadcasts): no LDS round trips
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2, 3
    return v;
}
// inclusive prefix maximum of non-negative values over the 64 lanes, same network
__device__ __forceinline__ int wave_scan_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
    return v;
}
__device__ __forceinline__ int mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// ---- periodic cell-list traversal -------------------------------------------------
// Calls f(slot, s0, s1, s2) for every (neighbour slot, image shift) candidate of atom m, all
// 64 lanes striding over the slots of one bin at a time.  Shifts are relative to the ORIGINAL
// (unwrapped) positions; images outside the reference's range (|s| > fac) are skipped, so the
// candidate set is exactly the reference's supercell (geometry.py:131-149).
template <class F>
__device__ __forceinline__ void for_each_candidate(const FrameGeom &g, const CellList &cl, int m, F f, int range_mult = 1) {
    const int lane = lane_id();
    const int lb = cl.atom_bin[m];
    const int b2 = lb % g.nb[2], b1 = (lb / g.nb[2]) % g.nb[1], b0 = lb / (g.nb[2] * g.nb[1]);
    int w0, w1, w2;
    unpack3(cl.atom_wrap[m], w0, w1, w2);
    int lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
        if (g.per[k]) { lo[k] = -g.rad[k]; hi[k] = g.rad[k]; }
        else if (g.nb[k] == 1) { lo[k] = 0; hi[k] = 0; }
        else if (g.nb[k] == 2) { lo[k] = 0; hi[k] = 1; }
        else { lo[k] = -1; hi[k] = 1; }
    }
    auto fdiv = [](int t, int n) { return t >= 0 ? t / n : -((n - 1 - t) / n); };
    // Bins along the fastest axis are contiguous in memory: a maximal range of them that shares one image
    // shift is a single slot range ("run").  Lanes describe the runs in parallel (one bin_start round trip for
    // all of them), a prefix sum lays their slots out in one flat candidate index, and the wave then walks that
    // index 64 candidates per step.
    const int n1 = hi[1] - lo[1] + 1, n01 = (hi[0] - lo[0] + 1) * n1;
    const int sh2_lo = fdiv(b2 + lo[2], g.nb[2]), nr2 = fdiv(b2 + hi[2], g.nb[2]) - sh2_lo + 1;
    const int n_runs = n01 * nr2;
    for (int r0 = 0; r0 < n_runs; r0 += WAVE) {
        const int r = r0 + lane;
        int len = 0, slot0 = 0, shp = 0;
        if (r < n_runs) {
            const int i01 = r / nr2, q = r - i01 * nr2, i0 = i01 / n1, i1 = i01 - i0 * n1;
            const int t0 = b0 + lo[0] + i0, t1 = b1 + lo[1] + i1;
            int sh0 = fdiv(t0, g.nb[0]), sh1 = fdiv(t1, g.nb[1]), sh2 = sh2_lo + q;
            const int c0 = t0 - sh0 * g.nb[0], c1 = t1 - sh1 * g.nb[1];
            const int o2a = max(lo[2], sh2 * g.nb[2] - b2), o2b = min(hi[2], (sh2 + 1) * g.nb[2] - 1 - b2);
            const int c2 = b2 + o2a - sh2 * g.nb[2];
            if (!g.per[0]) sh0 = 0;
            if (!g.per[1]) sh1 = 0;
            if (!g.per[2]) sh2 = 0;
            const int gb = g.bin_base + (c0 * g.nb[1] + c1) * g.nb[2] + c2;
            slot0 = cl.bin_start[gb];
            len = cl.bin_start[gb + (o2b - o2a + 1)] - slot0;
            shp = pack3(sh0, sh1, sh2);
        }
        const int incl = wave_scan_incl(len);
        const int total = __builtin_amdgcn_readlane(incl, WAVE - 1);
        const int rel = slot0 - (incl - len);                 // slot = rel + flat index, inside this run
        // (number of the first non-empty run behind each run; WAVE: none)
        const unsigned long long nonempty = __ballot(len > 0);
        const unsigned long long behind = lane < WAVE - 1 ? nonempty >> (lane + 1) : 0ull;
        const int next_ne = behind ? lane + 1 + __builtin_ctzll(behind) : WAVE;
        // slot record and packed image shift of the flat candidates wbase + lane (a window of 64).  The run of a candidate = the
        // first non-empty run after the last run that ends at or before it: every non-empty run that ends inside the window
        // pushes the number of its successor to the lane of its end (`ds_permute`: one crossbar trip, the ends of non-empty runs
        // are distinct), an inclusive maximum over the lanes (DPP) spreads it, and the lanes before the first end take the run
        // that contains wbase (one ballot).  (Before: a binary search over the lanes' prefix sums, six dependent shuffles.)
        // Indices past the end read slot 0 (never used).
        auto fetch = [&](int wbase, SlotRec &sr, int &shpk) {
            const int idx = wbase + lane;
            const int p = incl - wbase;
            const bool sender = (len > 0) & (p > 0) & (p < WAVE);
            const int got = __builtin_amdgcn_ds_permute((sender ? p : 0) << 2, sender ? next_ne + 1 : 0);
            const unsigned long long holds = __ballot((len > 0) & (p > 0));        // non-empty runs that end behind wbase
            const int first = holds ? __builtin_ctzll(holds) : 0;
            const int reached = wave_scan_max(lane == 0 ? 0 : got);
            const int run = min(reached ? reached - 1 : first, WAVE - 1);
            const int slot = __shfl(rel, run) + idx;
            shpk = __shfl(shp, run);
            sr = cl.slots[idx < total ? slot : 0];
        };
        // the records of step i + 1 are requested before step i is handed to f (its global round trip hides behind f's work)
        SlotRec sr_next;
        int shp_next;
        if (total > 0) fetch(0, sr_next, shp_next);
        for (int base = 0; base < total; base += WAVE) {
            const int idx = base + lane;
            SlotRec sr = sr_next;
            const int shpk = shp_next;
            if (base + WAVE < total) fetch(base + WAVE, sr_next, shp_next);
            int sh0, sh1, sh2;
            unpack3(shpk, sh0, sh1, sh2);
            bool ok = idx < total;
            int s0 = 0, s1 = 0, s2 = 0, sj = 0;
            if (ok) {
                int v0, v1, v2;
                unpack_ws(sr.ws, v0, v1, v2, sj);
                s0 = sh0 - v0 + w0; s1 = sh1 - v1 + w1; s2 = sh2 - v2 + w2;
                ok = (abs(s0) <= range_mult * g.fac[0]) && (abs(s1) <= range_mult * g.fac[1]) && (abs(s2) <= range_mult * g.fac[2]);
                if (ok && sr.atom == m && s0 == 0 && s1 == 0 && s2 == 0) ok = false;
            } else { sr.x = sr.y = sr.z = 0.0; sr.atom = 0; sr.ws = 0; }
            f(ok, sr, sj, s0, s1, s2);
        }
    }
}
