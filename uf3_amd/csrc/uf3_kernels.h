// uf3_kernels.h -- the gfx950 kernels of the UF3 hot path.
//
//   k_frame_bins      atom -> (frame, wrapped fractional bin)            [HBM-trivial]
//   k_bin_start / k_gather_sorted   cell list over radix-sorted atoms
//   k_build_n3        per-atom 3-body neighbour lists with image shifts, sorted by
//                     (species, reference supercell index)               one wave / atom
//   k_featurize       energy row + 3 force rows per atom                 one wave / atom
//   k_eval            energy + forces of a fitted model                  one wave / atom
//   k_gram_mfma       X^T X on the fp64 matrix cores
//
// Formulation: every atom m GATHERS all pair terms and all
// triplet terms it takes part in -- as centre, or as one of the two neighbours of
// a centre c in N3(m) -- so its three force-feature rows are accumulated in LDS and
// written exactly once, coalesced, with no global atomics.  A triplet is visited by
// each of its three atoms; translation invariance makes the three visits the three
// slices of the reference's arrange_deriv_3b (angles.py:235-286) output.
#pragma once
#include "uf3_device.h"

// ---------------------------------------------------------------------------------
// cell list
// ---------------------------------------------------------------------------------
__global__ void k_frame_bins(const BasisDev *B, const FrameGeom *geoms, const int64_t *atom_offsets,
                             int n_frames, int natoms, const double *pos, const int32_t *z,
                             int *frame_of, int *atom_bin, int *atom_wrap, signed char *spec,
                             int *sort_key, int *sort_val, int *err_flag) {
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= natoms) return;
    int lo = 0, hi = n_frames - 1;                // frame with atom_offsets[f] <= a < atom_offsets[f+1]
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (atom_offsets[mid] <= a) lo = mid; else hi = mid - 1;
    }
    const FrameGeom &g = geoms[lo];
    frame_of[a] = lo;
    int zz = z[a];
    int s = (zz >= 0 && zz < 120) ? B->z2s[zz] : -1;
    if (s < 0) { atomicExch(err_flag, 2); s = 0; }
    spec[a] = (signed char)s;
    double x = pos[3 * (size_t)a], y = pos[3 * (size_t)a + 1], w = pos[3 * (size_t)a + 2];
    int bin[3], wrap[3];
    for (int k = 0; k < 3; k++) {
        double f = x * g.inv[k] + y * g.inv[3 + k] + w * g.inv[6 + k];
        if (g.per[k]) {
            double fl = floor(f);
            int b = (int)((f - fl) * g.nb[k]);
            bin[k] = b >= g.nb[k] ? g.nb[k] - 1 : (b < 0 ? 0 : b);
            wrap[k] = (int)fl;
            if (wrap[k] < -500 || wrap[k] > 500) { atomicExch(err_flag, 1); wrap[k] = 0; }
        } else {
            long long q = (long long)floor(f / g.binw[k]);
            int b = (int)(q % g.nb[k]);
            bin[k] = b < 0 ? b + g.nb[k] : b;
            wrap[k] = 0;
        }
    }
    int lb = (bin[0] * g.nb[1] + bin[1]) * g.nb[2] + bin[2];
    atom_bin[a] = lb;
    atom_wrap[a] = pack3(wrap[0], wrap[1], wrap[2]);
    sort_key[a] = g.bin_base + lb;
    sort_val[a] = a;
}

__global__ void k_bin_start(const int *sorted_key, int natoms, int nbins, int *bin_start) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nbins) return;
    int lo = 0, hi = natoms;                      // first slot with key >= b
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (sorted_key[mid] < b) lo = mid + 1; else hi = mid;
    }
    bin_start[b] = lo;
}

__global__ void k_gather_sorted(const int *sorted_val, int natoms, const double *pos, const int *atom_wrap,
                                const signed char *spec, int *s_atom, double *s_pos, int *s_wrap,
                                signed char *s_spec) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= natoms) return;
    int a = sorted_val[s];
    s_atom[s] = a;
    s_pos[3 * (size_t)s] = pos[3 * (size_t)a];
    s_pos[3 * (size_t)s + 1] = pos[3 * (size_t)a + 1];
    s_pos[3 * (size_t)s + 2] = pos[3 * (size_t)a + 2];
    s_wrap[s] = atom_wrap[a];
    s_spec[s] = spec[a];
}

// vector from atom m (original position pm) to the image (slot, shift) of a neighbour
__device__ __forceinline__ void image_delta(const FrameGeom &g, const CellList &cl, int slot, int s0, int s1, int s2,
                                            const double *pm, double &dx, double &dy, double &dz) {
    double off[3];
    for (int k = 0; k < 3; k++) off[k] = s0 * g.cell[k] + s1 * g.cell[3 + k] + s2 * g.cell[6 + k];
    // (p_j + offset) - p_i, as the reference tiles positions first (geometry.py:146-148)
    dx = (cl.s_pos[3 * (size_t)slot] + off[0]) - pm[0];
    dy = (cl.s_pos[3 * (size_t)slot + 1] + off[1]) - pm[1];
    dz = (cl.s_pos[3 * (size_t)slot + 2] + off[2]) - pm[2];
}

// ---------------------------------------------------------------------------------
// 3-body neighbour lists: one wave per atom
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_build_n3(const BasisDev *B, const FrameGeom *geoms, const int *frame_of, CellList cl, N3Lists n3,
           const double *pos, int natoms, int *overflow_need) {
    extern __shared__ __align__(16) unsigned char smem[];
    int cap = n3.cap;
    unsigned long long *key = (unsigned long long *)smem;
    double *ex = (double *)(key + cap), *ey = ex + cap, *ez = ey + cap, *er = ez + cap;
    int *eparent = (int *)(er + cap), *eshift = eparent + cap, *esidx = eshift + cap, *espec = esidx + cap;

    int m = blockIdx.x;
    if (m >= natoms) return;
    int lane = lane_id();
    const FrameGeom g = geoms[frame_of[m]];
    double pm[3] = {pos[3 * (size_t)m], pos[3 * (size_t)m + 1], pos[3 * (size_t)m + 2]};
    double rmin3 = B->rmin3, rmax3 = B->rmax3;
    int count = 0;
    for_each_candidate(g, cl, m, [&](bool ok, int slot, int s0, int s1, int s2) {
        double dx = 0, dy = 0, dz = 0, d = 0;
        if (ok) {
            image_delta(g, cl, slot, s0, s1, s2, pm, dx, dy, dz);
            d = norm3_rn(dx, dy, dz);
            ok = (d > rmin3) && (d <= rmax3);            // angles.py:340: lower strict, upper inclusive
        }
        unsigned long long mask = __ballot(ok);
        if (ok) {
            int e = count + mbcnt(mask);
            if (e < cap) {
                int j = cl.s_atom[slot];
                int sidx = supercell_index(g, s0, s1, s2, j - g.atom_lo);
                int sp = cl.s_spec[slot];
                key[e] = ((unsigned long long)sp << 32) | (unsigned)sidx;
                ex[e] = dx; ey[e] = dy; ez[e] = dz; er[e] = d;
                eparent[e] = j; eshift[e] = pack3(s0, s1, s2); esidx[e] = sidx; espec[e] = sp;
            }
        }
        count += __popcll(mask);
    });
    __syncthreads();
    if (count > cap) { if (lane == 0) atomicMax(overflow_need, count); count = cap; }
    if (lane == 0) n3.cnt[m] = count;
    size_t base = (size_t)m * cap;
    for (int e = lane; e < count; e += WAVE) {     // rank sort by (species, supercell index)
        unsigned long long k = key[e];
        int rank = 0;
        for (int f = 0; f < count; f++) rank += key[f] < k;
        size_t o = base + rank;
        n3.parent[o] = eparent[e]; n3.shiftc[o] = eshift[e]; n3.sidx[o] = esidx[e]; n3.spec[o] = espec[e];
        n3.dx[o] = ex[e]; n3.dy[o] = ey[e]; n3.dz[o] = ez[e]; n3.r[o] = er[e];
    }
}


// Which of the two neighbours (m itself, or k) of centre c is the reference's "j" (leg l)?
// Different species: the lower atomic number.  Same species: the reference keeps pairs j < k by
// supercell index (angles.py:474).  Its force loop numbers atoms in the TRUE supercell even when
// the centre is a ghost, where j is additionally restricted to real atoms (angles.py:451-460):
//   centre real (shift 0)      -> compare the supercell indices seen from the centre;
//   centre ghost, k real       -> compare the two real indices;
//   centre ghost, k ghost      -> m (real) is always j.
// Only matters for symmetry-1 trios with equal neighbour species; reproduced for parity.
__device__ __forceinline__ bool neighbour_is_first(const FrameGeom &g, int sm, int ksp, int s0, int s1, int s2,
                                                   int m_local, int msidx, int ksidx, int kshift, int k_local) {
    (void)g; (void)msidx;
    if (sm != ksp) return sm < ksp;
    if (s0 == 0 && s1 == 0 && s2 == 0) return m_local < ksidx;
    int k0, k1, k2;
    unpack3(kshift, k0, k1, k2);
    bool k_real = (k0 + s0 == 0) && (k1 + s1 == 0) && (k2 + s2 == 0);
    return k_real ? (m_local < k_local) : true;
}

// ---------------------------------------------------------------------------------
// featurizer
// ---------------------------------------------------------------------------------
#ifdef UF3_PROFILE
#define PROF_DECL long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long prof_c = 0;
#define PROF_T0 prof_c = clock64();
#define PROF_ADD(k) { long long now__ = clock64(); prof_t[k] += now__ - prof_c; prof_c = now__; }
#else
#define PROF_DECL
#define PROF_T0
#define PROF_ADD(k)
#endif
#define NWAVES 8          // waves of the workgroup that cooperates on one atom
#define HALF 32           // triplet records staged per wave at a time
#define ITEM_STRIDE 38    // doubles per staged record (16-B aligned records)
// record layout (doubles): 0-7 (Bl,B'l)[4], 8-15 (Bm,B'm)[4], 16-23 (Bn,B'n)[4],
// 24-26 A1, 27-29 A2, 30-32 A3, 34-35 int4 {lut base, stride of l, stride of m, energy flag}

struct FeatArgs {
    const BasisDev *B;
    const TrioDev *trios;     // explicit global pointers (no flat loads through the struct)
    const KnotRec *recs;
    const int *lut;
    const FrameGeom *geoms;
    const int *frame_of;
    CellList cl;
    N3Lists n3;
    const double *pos;
    const signed char *spec;
    double *x_e;        // [n_frames][F] or null
    double *x_f;        // [natoms][3][F] or null
    int natoms, atoms_per_block;
    int col_lo, col_hi; // column window held in LDS
    int lut_len;        // > 0: the uint16 copy of the LUT lives in LDS
    long long *prof;    // UF3_PROFILE builds: [NWAVES][8] cycle counters
    int skip;           // profiling ablations (UF3_DEBUG_SKIP): 1 two-body, 2 centre role, 4 neighbour role, 8 scatter
};

__device__ __forceinline__ void lds_add(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS traffic inside one wave is in order; only the compiler has to be held back
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One triplet seen from atom m.  rl, rm, rn: leg lengths in the trio's (l, m, n) order;
// a1/a2/a3: -(d r_leg / d R_m) for the three legs (zero vector when the leg does not move with m).
struct TripletGeom {
    double rl, rm, rn;
    double a1[3], a2[3], a3[3];
    int trio;
    bool centre;
};

struct TripletRec {
    double v[3][4], d[3][4];
    int meta[4];
};

template <bool WANT_F>
__device__ __forceinline__ bool eval_triplet(const FeatArgs &A, const TripletGeom &t, bool valid, TripletRec &r) {
    if (!valid || t.trio < 0) return false;
    const TrioDev *td = A.trios + t.trio;
    // leg masks t[0] <= r <= t[-1] (angles.py:502-508).  r == t[0] selects no basis function
    // (searchsorted - 4 < 0) and at r == t[-1] every selected scipy element evaluates to 0
    // (half-open last interval), so both ends contribute nothing: open interval here.
    if (!((t.rl > td->leg[0].t0) && (t.rl < td->leg[0].tlast) && (t.rm > td->leg[1].t0) && (t.rm < td->leg[1].tlast) &&
          (t.rn > td->leg[2].t0) && (t.rn < td->leg[2].tlast))) return false;
    int il = find_interval(A.recs, td->leg[0], t.rl);
    int im = find_interval(A.recs, td->leg[1], t.rm);
    int in = find_interval(A.recs, td->leg[2], t.rn);
    bspline4<WANT_F>(A.recs[td->leg[0].rec_off + il], t.rl, r.v[0], r.d[0]);
    bspline4<WANT_F>(A.recs[td->leg[1].rec_off + im], t.rm, r.v[1], r.d[1]);
    bspline4<WANT_F>(A.recs[td->leg[2].rec_off + in], t.rn, r.v[2], r.d[2]);
    int mn = td->dim_m * td->dim_n;
    r.meta[0] = td->lut_off + (il - 3) * mn + (im - 3) * td->dim_n + (in - 3);
    r.meta[1] = mn;
    r.meta[2] = td->dim_n;
    r.meta[3] = t.centre ? 1 : 0;
    return true;
}

// 64 evaluated triplets (one per lane) -> two half-batches through the wave's 32-record LDS stage,
// each scattered with lanes <-> the 4x4x4 block of basis products of one record at a time.
template <bool WANT_E, bool WANT_F>
__device__ __forceinline__ void stage_and_scatter(const FeatArgs &A, const TripletGeom &t, const TripletRec &r, bool valid,
                                                  double *stage, double *rowbuf, double *erow,
                                                  const unsigned short *lut16) {
    const int lane = lane_id();
    const int a = lane >> 4, b = (lane >> 2) & 3, c = lane & 3;
    const int W = A.col_hi - A.col_lo;
    for (int half = 0; half < 2; half++) {
        bool mine = valid && ((lane >> 5) == half);
        unsigned long long mask = __ballot(mine);
        if (mask == 0) continue;
        if (mine) {
            double *rec = stage + (size_t)mbcnt(mask) * ITEM_STRIDE;
            for (int leg = 0; leg < 3; leg++)
                for (int q = 0; q < 4; q++) {
                    rec[8 * leg + 2 * q] = r.v[leg][q];
                    if (WANT_F) rec[8 * leg + 2 * q + 1] = r.d[leg][q];
                }
            if (WANT_F) for (int q = 0; q < 3; q++) { rec[24 + q] = t.a1[q]; rec[27 + q] = t.a2[q]; rec[30 + q] = t.a3[q]; }
            int4 mt = make_int4(r.meta[0], r.meta[1], r.meta[2], r.meta[3]);
            *(int4 *)(rec + 34) = mt;
        }
        wave_sync();
        const int n_staged = (A.skip & 8) ? 0 : __popcll(mask);
        for (int q = 0; q < n_staged; q++) {
            const double *rec = stage + (size_t)q * ITEM_STRIDE;
            const int4 mt = *(const int4 *)(rec + 34);
            const int raw = mt.x + a * mt.y + b * mt.z + c;
            int col = lut16 ? (int)lut16[raw] : A.lut[raw];
            if (lut16) col = (col == 0xFFFF) ? -1 : col;
            if (col < A.col_lo || col >= A.col_hi) continue;
            col -= A.col_lo;
            const double2 L = *(const double2 *)(rec + 2 * a);
            const double2 M = *(const double2 *)(rec + 8 + 2 * b);
            const double2 N = *(const double2 *)(rec + 16 + 2 * c);
            const double z = L.x * M.x;
            if (WANT_E) { if (mt.w) lds_add(erow + col, z * N.x); }
            if (WANT_F) {
                const double p1 = L.y * (M.x * N.x), p2 = M.y * (L.x * N.x), p3 = N.y * z;
                const double2 a01 = *(const double2 *)(rec + 24), a23 = *(const double2 *)(rec + 26),
                              a45 = *(const double2 *)(rec + 28), a67 = *(const double2 *)(rec + 30);
                const double a8 = rec[32];
                // A1 = (a01.x, a01.y, a23.x)  A2 = (a23.y, a45.x, a45.y)  A3 = (a67.x, a67.y, a8)
                lds_add(rowbuf + col, p1 * a01.x + p2 * a23.y + p3 * a67.x);
                lds_add(rowbuf + W + col, p1 * a01.y + p2 * a45.x + p3 * a67.y);
                lds_add(rowbuf + 2 * W + col, p1 * a23.x + p2 * a45.y + p3 * a8);
            }
        }
        wave_sync();
    }
}

template <bool WANT_E, bool WANT_F>
__global__ void __launch_bounds__(NWAVES * WAVE)
k_featurize(FeatArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    const BasisDev *B = A.B;
    const int W = A.col_hi - A.col_lo, cap = A.n3.cap;
    double *rowbuf = (double *)smem;                               // [3][W]   (WANT_F)
    double *erow = rowbuf + (WANT_F ? 3 * W : 0);                  // [W]      (WANT_E)
    double *ox = erow + (WANT_E ? W : 0), *oy = ox + cap, *oz = oy + cap, *orr = oz + cap;   // own neighbour list
    double *stage_all = orr + cap + (cap & 1);                     // [NWAVES][HALF][ITEM_STRIDE], 16-B aligned
    int *oparent = (int *)(stage_all + (size_t)NWAVES * HALF * ITEM_STRIDE), *oshift = oparent + cap,
        *osidx = oshift + cap, *ospec = osidx + cap, *ooff = ospec + cap;        // ooff [cap+1]
    unsigned short *lut_lds = (unsigned short *)(ooff + cap + 1 + ((cap + 1) & 1));
    const unsigned short *lut16 = A.lut_len > 0 ? lut_lds : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *stage = stage_all + (size_t)wave * HALF * ITEM_STRIDE;
    const int nacc = (WANT_F ? 3 * W : 0) + (WANT_E ? W : 0);
    for (int q = tid; q < nacc; q += NWAVES * WAVE) rowbuf[q] = 0.0;
    for (int q = tid; q < A.lut_len; q += NWAVES * WAVE) { int v = A.lut[q]; lut_lds[q] = v < 0 ? 0xFFFF : (unsigned short)v; }
    __syncthreads();

    const int a0 = blockIdx.x * A.atoms_per_block, a1 = min(a0 + A.atoms_per_block, A.natoms);
    int cur_frame = -1;
    PROF_DECL
    for (int m = a0; m < a1; m++) {
        PROF_T0
        const int fr = A.frame_of[m];
        if (WANT_E && fr != cur_frame && cur_frame >= 0) {
            for (int q = tid; q < W; q += NWAVES * WAVE) {
                double v = erow[q];
                if (v != 0.0) unsafeAtomicAdd(A.x_e + (size_t)cur_frame * B->F + A.col_lo + q, v);
                erow[q] = 0.0;
            }
            __syncthreads();
        }
        cur_frame = fr;
        const FrameGeom g = A.geoms[fr];
        const int sm = A.spec[m];
        const double pm[3] = {A.pos[3 * (size_t)m], A.pos[3 * (size_t)m + 1], A.pos[3 * (size_t)m + 2]};
        if (WANT_E && tid == 0 && sm >= A.col_lo && sm < A.col_hi) lds_add(erow + (sm - A.col_lo), 1.0);   // 1-body count

        // ---- own 3-body neighbour list -> LDS --------------------------------------------
        int n = 0;
        if (B->T > 0) {
            n = A.n3.cnt[m];
            size_t base = (size_t)m * cap;
            for (int e = tid; e < n; e += NWAVES * WAVE) {
                ox[e] = A.n3.dx[base + e]; oy[e] = A.n3.dy[base + e]; oz[e] = A.n3.dz[base + e]; orr[e] = A.n3.r[base + e];
                oparent[e] = A.n3.parent[base + e]; oshift[e] = A.n3.shiftc[base + e];
                osidx[e] = A.n3.sidx[base + e]; ospec[e] = A.n3.spec[base + e];
            }
        }

        PROF_ADD(6)
        // ---- 2-body: the waves share the neighbour bins ------------------------------------
        if (!(A.skip & 1)) for_each_candidate_strided(g, A.cl, m, wave, NWAVES, [&](bool ok, int slot, int s0, int s1, int s2) {
            if (!ok) return;
            int sj = A.cl.s_spec[slot];
            const PairDev &pd = B->pairs[B->pair_of[sm * UF3_MAX_SPECIES + sj]];
            if (pd.col + pd.nb <= A.col_lo || pd.col >= A.col_hi) return;
            double dx, dy, dz;
            image_delta(g, A.cl, slot, s0, s1, s2, pm, dx, dy, dz);
            double d = norm3_rn(dx, dy, dz);
            if (!(d > pd.rmin && d < pd.rmax)) return;        // distances.py:66 strict both sides
            int i = find_interval(A.recs, pd.leg, d);
            double v[4], dv[4];
            bspline4<WANT_F>(A.recs[pd.leg.rec_off + i], d, v, dv);
            double inv = 2.0 / d;
            for (int q = 0; q < 4; q++) {
                int bidx = i - 3 + q;
                if (bidx < B->lead2 || bidx >= pd.nb - B->trail2) continue;   // bspline.py:840,880
                int col = pd.col + bidx - A.col_lo;
                if (col < 0 || col >= W) continue;
                if (WANT_E) lds_add(erow + col, v[q]);
                if (WANT_F) {
                    // -sum_p B'(r_p) (delta_mj - delta_mi)(R_j-R_i)/r over both directed images of the bond
                    double s = dv[q] * inv;
                    lds_add(rowbuf + col, s * dx);
                    lds_add(rowbuf + W + col, s * dy);
                    lds_add(rowbuf + 2 * W + col, s * dz);
                }
            }
        });
        PROF_ADD(0)
        __syncthreads();
        PROF_ADD(5)

        // ---- 3-body ---------------------------------------------------------------------
        if (B->T > 0) {
            // (a) m is the centre: neighbour pairs a < b of its own (species, index)-sorted list
            const int n_pairs = (A.skip & 2) ? 0 : n * (n - 1) / 2;
            // items are dealt to the waves in equal contiguous shares (a share is walked 64 at a time)
            const int share_c = (n_pairs + NWAVES - 1) / NWAVES;
            const int end_c = min(n_pairs, (wave + 1) * share_c);
            for (int p0 = wave * share_c; p0 < end_c; p0 += WAVE) {
                int p = p0 + lane;
                bool valid = p < end_c;
                TripletGeom t;
                TripletRec r;
                t.centre = true; t.trio = -1;
                if (valid) {
                    int bb = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
                    while (bb * (bb - 1) / 2 > p) --bb;
                    while ((bb + 1) * bb / 2 <= p) ++bb;
                    int aa = p - bb * (bb - 1) / 2;
                    t.rl = orr[aa]; t.rm = orr[bb];
                    double ex = ox[bb] - ox[aa], ey = oy[bb] - oy[aa], ez = oz[bb] - oz[aa];
                    t.rn = norm3_rn(ex, ey, ez);
                    t.trio = B->trio_of[(sm * UF3_MAX_SPECIES + ospec[aa]) * UF3_MAX_SPECIES + ospec[bb]];
                    if (WANT_F) {
                        double il = 1.0 / t.rl, im = 1.0 / t.rm;
                        t.a1[0] = ox[aa] * il; t.a1[1] = oy[aa] * il; t.a1[2] = oz[aa] * il;
                        t.a2[0] = ox[bb] * im; t.a2[1] = oy[bb] * im; t.a2[2] = oz[bb] * im;
                        t.a3[0] = t.a3[1] = t.a3[2] = 0.0;
                    }
                }
                PROF_ADD(1)
                valid = eval_triplet<WANT_F>(A, t, valid, r);
                PROF_ADD(2)
                stage_and_scatter<WANT_E, WANT_F>(A, t, r, valid, stage, rowbuf, erow, lut16);
                PROF_ADD(3)
            }
            // (b) m is a neighbour of centre c = own entry e; the other neighbour k runs over N3(c)
            if (WANT_F) {
                if (wave == 0) {
                    int total = 0;
                    for (int e0 = 0; e0 < n; e0 += WAVE) {           // exclusive scan of |N3(parent_e)|
                        int e = e0 + lane;
                        int cnt = e < n ? A.n3.cnt[oparent[e]] : 0;
                        int incl = cnt;
                        for (int sh = 1; sh < WAVE; sh <<= 1) { int o = __shfl_up(incl, sh); if (lane >= sh) incl += o; }
                        if (e < n) ooff[e] = total + incl - cnt;
                        total += __shfl(incl, WAVE - 1);
                    }
                    if (lane == 0) ooff[n] = total;
                }
                __syncthreads();
                const int total = (A.skip & 4) ? 0 : ooff[n];
                const int m_local = m - g.atom_lo;
                const int share_n = (total + NWAVES - 1) / NWAVES;
                const int end_n = min(total, (wave + 1) * share_n);
                for (int p0 = wave * share_n; p0 < end_n; p0 += WAVE) {
                    int p = p0 + lane;
                    bool valid = p < end_n;
                    TripletGeom t;
                    TripletRec r;
                    t.centre = false; t.trio = -1;
                    if (valid) {
                        int lo = 0, hi = n - 1;                  // entry e with ooff[e] <= p < ooff[e+1]
                        while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (ooff[mid] <= p) lo = mid; else hi = mid - 1; }
                        int e = lo, kk = p - ooff[e];
                        int pc = oparent[e];
                        size_t kb = (size_t)pc * cap + kk;
                        int s0, s1, s2;
                        unpack3(oshift[e], s0, s1, s2);
                        int rev_shift = pack3(-s0, -s1, -s2);
                        int kparent = A.n3.parent[kb], kshift = A.n3.shiftc[kb];
                        valid = !(kparent == m && kshift == rev_shift);       // k is m itself
                        if (valid) {
                            int ksp = A.n3.spec[kb], ksidx = A.n3.sidx[kb];
                            int msidx = supercell_index(g, -s0, -s1, -s2, m_local);  // m as numbered from c
                            double vx = A.n3.dx[kb], vy = A.n3.dy[kb], vz = A.n3.dz[kb], rk = A.n3.r[kb];
                            double ex = ox[e] + vx, ey = oy[e] + vy, ez = oz[e] + vz;   // m -> k
                            t.rn = norm3_rn(ex, ey, ez);
                            bool m_first = neighbour_is_first(g, sm, ksp, s0, s1, s2, m_local, msidx, ksidx, kshift,
                                                              kparent - g.atom_lo);
                            int sc = ospec[e];
                            double ie = 1.0 / orr[e], in = 1.0 / t.rn;
                            double ue[3] = {ox[e] * ie, oy[e] * ie, oz[e] * ie};
                            t.a3[0] = ex * in; t.a3[1] = ey * in; t.a3[2] = ez * in;
                            if (m_first) {
                                t.rl = orr[e]; t.rm = rk;
                                t.trio = B->trio_of[(sc * UF3_MAX_SPECIES + sm) * UF3_MAX_SPECIES + ksp];
                                for (int q = 0; q < 3; q++) { t.a1[q] = ue[q]; t.a2[q] = 0.0; }
                            } else {
                                t.rl = rk; t.rm = orr[e];
                                t.trio = B->trio_of[(sc * UF3_MAX_SPECIES + ksp) * UF3_MAX_SPECIES + sm];
                                for (int q = 0; q < 3; q++) { t.a1[q] = 0.0; t.a2[q] = ue[q]; }
                            }
                        }
                    }
                    PROF_ADD(4)
                    valid = eval_triplet<WANT_F>(A, t, valid, r);
                    PROF_ADD(2)
                    stage_and_scatter<false, WANT_F>(A, t, r, valid, stage, rowbuf, erow, lut16);
                    PROF_ADD(3)
                }
            }
        }
        PROF_ADD(7)
        __syncthreads();
        PROF_ADD(5)
        if (WANT_F) {   // the three rows of atom m leave the chip once, coalesced
            double *dst = A.x_f + (size_t)m * 3 * B->F + A.col_lo;
            for (int q = tid; q < 3 * W; q += NWAVES * WAVE) {
                int comp = q / W, col = q - comp * W;
                dst[(size_t)comp * B->F + col] = rowbuf[q];
                rowbuf[q] = 0.0;
            }
            __syncthreads();
        }
        PROF_ADD(5)
    }
#ifdef UF3_PROFILE
    if (lane == 0 && A.prof) for (int q = 0; q < 8; q++) atomicAdd((unsigned long long *)A.prof + wave * 8 + q, (unsigned long long)prof_t[q]);
#endif
    if (WANT_E && cur_frame >= 0) {
        for (int q = tid; q < W; q += NWAVES * WAVE) {
            double v = erow[q];
            if (v != 0.0) unsafeAtomicAdd(A.x_e + (size_t)cur_frame * B->F + A.col_lo + q, v);
        }
    }
}

// ---------------------------------------------------------------------------------
// evaluator: energy + forces of a fitted model, one wave per atom, lanes <-> items
// ---------------------------------------------------------------------------------
struct EvalArgs {
    const BasisDev *B;
    const FrameGeom *geoms;
    const int *frame_of;
    CellList cl;
    N3Lists n3;
    const double *pos;
    const signed char *spec;
    const double *c1, *c2, *c3;   // device copies; c2 indexed by (pair col - S) + b, c3 by lut offset + raw
    double *e_atom;               // [natoms]
    double *forces;               // [natoms][3] or null
    int natoms;
};

// V and its three leg partials at (rl, rm, rn) from the full coefficient grid of a trio
__device__ __forceinline__ bool trio_value(const BasisDev *B, const double *c3, int trio, double rl, double rm, double rn,
                                           bool want_grad, double &val, double *grad) {
    if (trio < 0) return false;
    const TrioDev *td = B->trios + trio;
    if (!((rl > td->leg[0].t0) && (rl < td->leg[0].tlast) && (rm > td->leg[1].t0) && (rm < td->leg[1].tlast) &&
          (rn > td->leg[2].t0) && (rn < td->leg[2].tlast))) return false;
    int il = find_interval(B->recs, td->leg[0], rl), im = find_interval(B->recs, td->leg[1], rm),
        in = find_interval(B->recs, td->leg[2], rn);
    double vl[4], vm[4], vn[4], dl[4], dm[4], dn[4];
    bspline4<true>(B->recs[td->leg[0].rec_off + il], rl, vl, dl);
    bspline4<true>(B->recs[td->leg[1].rec_off + im], rm, vm, dm);
    bspline4<true>(B->recs[td->leg[2].rec_off + in], rn, vn, dn);
    int mn = td->dim_m * td->dim_n;
    const double *c = c3 + td->lut_off + (il - 3) * mn + (im - 3) * td->dim_n + (in - 3);
    double v = 0, g0 = 0, g1 = 0, g2 = 0;
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) {
            double s = 0, sd = 0;
            const double *row = c + a * mn + b * td->dim_n;
            for (int q = 0; q < 4; q++) { double cc = row[q]; s += cc * vn[q]; sd += cc * dn[q]; }
            v += vl[a] * vm[b] * s;
            if (want_grad) { g0 += dl[a] * vm[b] * s; g1 += vl[a] * dm[b] * s; g2 += vl[a] * vm[b] * sd; }
        }
    val = v; grad[0] = g0; grad[1] = g1; grad[2] = g2;
    return true;
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int sh = 32; sh > 0; sh >>= 1) v += __shfl_xor(v, sh);
    return v;
}

__global__ void __launch_bounds__(64)
k_eval(EvalArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    const BasisDev *B = A.B;
    const int cap = A.n3.cap;
    double *ox = (double *)smem, *oy = ox + cap, *oz = oy + cap, *orr = oz + cap;
    int *oparent = (int *)(orr + cap), *oshift = oparent + cap, *osidx = oshift + cap, *ospec = osidx + cap,
        *ooff = ospec + cap;
    int m = blockIdx.x;
    if (m >= A.natoms) return;
    int lane = lane_id();
    const FrameGeom g = A.geoms[A.frame_of[m]];
    const int sm = A.spec[m];
    const bool want_f = A.forces != nullptr;
    double pm[3] = {A.pos[3 * (size_t)m], A.pos[3 * (size_t)m + 1], A.pos[3 * (size_t)m + 2]};
    double e = 0.0, fx = 0.0, fy = 0.0, fz = 0.0;
    if (lane == 0) e = A.c1[sm];
    for_each_candidate(g, A.cl, m, [&](bool ok, int slot, int s0, int s1, int s2) {
        if (!ok) return;
        int sj = A.cl.s_spec[slot];
        const PairDev &pd = B->pairs[B->pair_of[sm * UF3_MAX_SPECIES + sj]];
        double dx, dy, dz;
        image_delta(g, A.cl, slot, s0, s1, s2, pm, dx, dy, dz);
        double d = norm3_rn(dx, dy, dz);
        if (!(d > pd.rmin && d < pd.rmax)) return;
        int i = find_interval(B->recs, pd.leg, d);
        double v[4], dv[4];
        bspline4<true>(B->recs[pd.leg.rec_off + i], d, v, dv);
        const double *c = A.c2 + (pd.col - B->S) + (i - 3);
        double phi = 0, dphi = 0;
        for (int q = 0; q < 4; q++) { phi += c[q] * v[q]; dphi += c[q] * dv[q]; }
        e += phi;
        double s = 2.0 * dphi / d;
        fx += s * dx; fy += s * dy; fz += s * dz;
    });
    if (B->T > 0) {
        int n = A.n3.cnt[m];
        size_t base = (size_t)m * cap;
        for (int q = lane; q < n; q += WAVE) {
            ox[q] = A.n3.dx[base + q]; oy[q] = A.n3.dy[base + q]; oz[q] = A.n3.dz[base + q]; orr[q] = A.n3.r[base + q];
            oparent[q] = A.n3.parent[base + q]; oshift[q] = A.n3.shiftc[base + q];
            osidx[q] = A.n3.sidx[base + q]; ospec[q] = A.n3.spec[base + q];
        }
        __syncthreads();
        int n_pairs = n * (n - 1) / 2;
        for (int p = lane; p < n_pairs; p += WAVE) {
            int bb = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
            while (bb * (bb - 1) / 2 > p) --bb;
            while ((bb + 1) * bb / 2 <= p) ++bb;
            int aa = p - bb * (bb - 1) / 2;
            double rl = orr[aa], rm = orr[bb];
            double rn = norm3_rn(ox[bb] - ox[aa], oy[bb] - oy[aa], oz[bb] - oz[aa]);
            int trio = B->trio_of[(sm * UF3_MAX_SPECIES + ospec[aa]) * UF3_MAX_SPECIES + ospec[bb]];
            double val, gr[3];
            if (!trio_value(B, A.c3, trio, rl, rm, rn, want_f, val, gr)) continue;
            e += val;
            if (want_f) {   // F_m = -dV/dR_m = gl * u_ij + gm * u_ik
                double a = gr[0] / rl, b = gr[1] / rm;
                fx += a * ox[aa] + b * ox[bb]; fy += a * oy[aa] + b * oy[bb]; fz += a * oz[aa] + b * oz[bb];
            }
        }
        if (want_f) {
            int total = 0;
            for (int e0 = 0; e0 < n; e0 += WAVE) {
                int q = e0 + lane;
                int cnt = q < n ? A.n3.cnt[oparent[q]] : 0;
                int incl = cnt;
                for (int sh = 1; sh < WAVE; sh <<= 1) { int o = __shfl_up(incl, sh); if (lane >= sh) incl += o; }
                if (q < n) ooff[q] = total + incl - cnt;
                total += __shfl(incl, WAVE - 1);
            }
            if (lane == 0) ooff[n] = total;
            __syncthreads();
            int m_local = m - g.atom_lo;
            for (int p = lane; p < total; p += WAVE) {
                int lo = 0, hi = n - 1;
                while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (ooff[mid] <= p) lo = mid; else hi = mid - 1; }
                int q = lo, kk = p - ooff[q];
                size_t kb = (size_t)oparent[q] * cap + kk;
                int s0, s1, s2;
                unpack3(oshift[q], s0, s1, s2);
                if (A.n3.parent[kb] == m && A.n3.shiftc[kb] == pack3(-s0, -s1, -s2)) continue;
                int ksp = A.n3.spec[kb], ksidx = A.n3.sidx[kb];
                int msidx = supercell_index(g, -s0, -s1, -s2, m_local);
                double vx = A.n3.dx[kb], vy = A.n3.dy[kb], vz = A.n3.dz[kb], rk = A.n3.r[kb];
                double ex = ox[q] + vx, ey = oy[q] + vy, ez = oz[q] + vz;
                double rn = norm3_rn(vx - (-ox[q]), vy - (-oy[q]), vz - (-oz[q]));
                bool m_first = neighbour_is_first(g, sm, ksp, s0, s1, s2, m_local, msidx, ksidx, A.n3.shiftc[kb],
                                                  A.n3.parent[kb] - g.atom_lo);
                int sc = ospec[q];
                double val, gr[3];
                int trio; double rl, rm;
                if (m_first) { rl = orr[q]; rm = rk; trio = B->trio_of[(sc * UF3_MAX_SPECIES + sm) * UF3_MAX_SPECIES + ksp]; }
                else { rl = rk; rm = orr[q]; trio = B->trio_of[(sc * UF3_MAX_SPECIES + ksp) * UF3_MAX_SPECIES + sm]; }
                if (!trio_value(B, A.c3, trio, rl, rm, rn, true, val, gr)) continue;
                double ge = (m_first ? gr[0] : gr[1]) / orr[q], gn = gr[2] / rn;
                fx += ge * ox[q] + gn * ex; fy += ge * oy[q] + gn * ey; fz += ge * oz[q] + gn * ez;
            }
        }
    }
    e = wave_sum(e);
    if (lane == 0) A.e_atom[m] = e;
    if (want_f) {
        fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
        if (lane == 0) { A.forces[3 * (size_t)m] = fx; A.forces[3 * (size_t)m + 1] = fy; A.forces[3 * (size_t)m + 2] = fz; }
    }
}

// frame energies: deterministic tree sum of the per-atom energies
__global__ void k_frame_energy(const double *e_atom, const int64_t *atom_offsets, double *energies) {
    __shared__ double part[256];
    int f = blockIdx.x;
    double s = 0.0;
    for (int64_t a = atom_offsets[f] + threadIdx.x; a < atom_offsets[f + 1]; a += blockDim.x) s += e_atom[a];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = blockDim.x / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) energies[f] = part[0];
}

// ---------------------------------------------------------------------------------
// normal equations: G (+)= X^T X on the fp64 matrix cores, o (+)= X^T y
// ---------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

// D-fragment layout probe: element v of lane l of a 16x16 f64 accumulator is D[row][col]
__global__ void k_mfma_probe(int *rowcol) {
    int l = threadIdx.x;
    double a = (l / 16 == 0) ? (double)(l % 16 + 1) : 0.0;          // A[i][k]: i = l%16, k = l/16
    double b = (l / 16 == 0) ? (double)(l % 16 + 1) * 100.0 : 0.0;  // B[k][j]: k = l/16, j = l%16
    double4_t acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int v = 0; v < 4; v++) {
        int p = (int)(acc[v] / 100.0 + 0.5);   // (i+1)*(j+1) is not unique; decode with a second product below
        rowcol[(l * 4 + v) * 2] = p;
    }
    // second probe: A[i][0] = 1, B[0][j] = j+1  -> D[i][j] = j+1 gives the column
    a = (l / 16 == 0) ? 1.0 : 0.0;
    b = (l / 16 == 0) ? (double)(l % 16 + 1) : 0.0;
    double4_t acc2 = {0, 0, 0, 0};
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
    for (int v = 0; v < 4; v++) {
        int col = (int)(acc2[v] + 0.5);
        int prod = rowcol[(l * 4 + v) * 2];
        rowcol[(l * 4 + v) * 2] = col > 0 ? prod / col - 1 : -1;    // row
        rowcol[(l * 4 + v) * 2 + 1] = col - 1;                       // col
    }
}

// One wave computes a 32x32 tile of G over a chunk of rows; 4 waves per block share nothing.
// grid = (tile pairs (ti <= tj), row chunks).  A = X^T (16 x 4 per MFMA), B = X (4 x 16).
__global__ void __launch_bounds__(256)
k_gram_mfma(const double *x, int64_t n_rows, int n_feat, int64_t ld, int rows_per_chunk,
            const int *tile_i, const int *tile_j, const int *frag_rowcol, double *gram) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int pair = blockIdx.x * 4 + wave;
    int ti = tile_i[pair], tj = tile_j[pair];
    if (ti < 0) return;
    int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = r0 + rows_per_chunk;
    if (r1 > n_rows) r1 = n_rows;
    int i = lane & 15, k = lane >> 4;
    int ca0 = ti * 32 + i, ca1 = ca0 + 16, cb0 = tj * 32 + i, cb1 = cb0 + 16;
    bool va0 = ca0 < n_feat, va1 = ca1 < n_feat, vb0 = cb0 < n_feat, vb1 = cb1 < n_feat;
    double4_t acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
    for (int64_t r = r0; r < r1; r += 4) {
        int64_t row = r + k;
        bool vr = row < r1;
        const double *xr = x + row * ld;
        double a0 = (vr && va0) ? xr[ca0] : 0.0, a1 = (vr && va1) ? xr[ca1] : 0.0;
        double b0 = (vr && vb0) ? xr[cb0] : 0.0, b1 = (vr && vb1) ? xr[cb1] : 0.0;
        acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc11, 0, 0, 0);
    }
    for (int v = 0; v < 4; v++) {
        int fr = frag_rowcol[(lane * 4 + v) * 2], fc = frag_rowcol[(lane * 4 + v) * 2 + 1];
        int gi0 = ti * 32 + fr, gi1 = gi0 + 16, gj0 = tj * 32 + fc, gj1 = gj0 + 16;
        double vals[4] = {acc00[v], acc01[v], acc10[v], acc11[v]};
        int gis[4] = {gi0, gi0, gi1, gi1}, gjs[4] = {gj0, gj1, gj0, gj1};
        for (int q = 0; q < 4; q++) {
            int gi = gis[q], gj = gjs[q];
            if (gi >= n_feat || gj >= n_feat) continue;
            if (ti == tj && gj < gi) continue;                      // diagonal tile: upper part only
            if (vals[q] != 0.0) unsafeAtomicAdd(gram + (size_t)gi * n_feat + gj, vals[q]);
        }
    }
}

__global__ void k_gram_mirror(double *gram, int n_feat) {
    int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j < n_feat && j < i) gram[(size_t)i * n_feat + j] = gram[(size_t)j * n_feat + i];
}

__global__ void k_ordinate(const double *x, const double *y, int64_t n_rows, int n_feat, int64_t ld,
                           int rows_per_chunk, double *ord) {
    int col = blockIdx.x * blockDim.x + threadIdx.x;
    int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = r0 + rows_per_chunk;
    if (r1 > n_rows) r1 = n_rows;
    if (col >= n_feat) return;
    double s = 0.0;
    for (int64_t r = r0; r < r1; r++) s += x[r * ld + col] * y[r];
    if (s != 0.0) unsafeAtomicAdd(ord + col, s);
}

// ---------------------------------------------------------------------------------
// neighbour index dump (debug / parity): unsorted tuples, the host sorts them
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_debug_pairs(const BasisDev *B, const FrameGeom *geoms, const int *frame_of, CellList cl, const double *pos,
              const signed char *spec, int natoms, long long *counts /*[P+1]*/, long long *tuples /*[cap][3]*/,
              long long cap) {
    int m = blockIdx.x;
    if (m >= natoms) return;
    const FrameGeom g = geoms[frame_of[m]];
    const int sm = spec[m];
    double pm[3] = {pos[3 * (size_t)m], pos[3 * (size_t)m + 1], pos[3 * (size_t)m + 2]};
    for_each_candidate(g, cl, m, [&](bool ok, int slot, int s0, int s1, int s2) {
        if (!ok) return;
        double dx, dy, dz;
        image_delta(g, cl, slot, s0, s1, s2, pm, dx, dy, dz);
        double d = norm3_rn(dx, dy, dz);
        int j = cl.s_atom[slot];
        long long sidx = supercell_index(g, s0, s1, s2, j - g.atom_lo);
        int p = B->pair_of[sm * UF3_MAX_SPECIES + cl.s_spec[slot]];
        const PairDev &pd = B->pairs[p];
        if (d > pd.rmin && d < pd.rmax) {
            atomicAdd((unsigned long long *)&counts[p], 1ULL);
            long long o = (long long)atomicAdd((unsigned long long *)&counts[B->P + 1], 1ULL);
            if (o < cap) { tuples[3 * o] = p; tuples[3 * o + 1] = m - g.atom_lo; tuples[3 * o + 2] = sidx; }
        }
        if (B->T > 0 && d > B->rmin3 && d <= B->rmax3) {
            atomicAdd((unsigned long long *)&counts[B->P], 1ULL);
            long long o = (long long)atomicAdd((unsigned long long *)&counts[B->P + 1], 1ULL);
            if (o < cap) { tuples[3 * o] = B->P; tuples[3 * o + 1] = m - g.atom_lo; tuples[3 * o + 2] = sidx; }
        }
    });
}
