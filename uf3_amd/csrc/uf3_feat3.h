// uf3_feat3.h -- 3-body force (and energy) rows by BOND FACTORISATION: k_featurize3 (round 4).  gfx950 only.
//
// Reference: uf3/representation/angles.py:142-286 (featurize_force_3b / arrange_deriv_3b), :17-139 (energy), :424-632
// (triplets, leg values and derivatives).  Same gather formulation as k_featurize's trio blocks (an atom m collects the
// triplets it centres and those in which it is a neighbour of a centre e), but the sum over the triplets is taken in two
// stages instead of one rank-2 update of the (component, l) x (m, n) window per triplet:
//
//   every term of m's rows has the form   T_f[c][i] * ( sum over the triplets that share the bond f )  P_g[j] Q[n]
//   where f is a bond of m that stays fixed -- (m, a) when m is the centre and a sits on one centre leg, (m, e) when m is a
//   neighbour of the centre e -- and g is the bond on the OTHER centre leg ((m, b), or (e, k)), Q the values of leg n:
//
//   stage 1 (per triplet, 4 multiply-adds per lane):   W_s[j][n] += P_s[j] Q_s[n]          s = x, y, z, plain
//       centre role     P = (u_g,x B'(r_g), u_g,y B'(r_g), u_g,z B'(r_g), B(r_g))[j]     Q = B_n(r_ab)[n]
//       neighbour role  P = B(r_ek)[j]     Q = (a3_x B_n'(r_mk), a3_y B_n', a3_z B_n', B_n(r_mk))[n]      a3 = unit(m -> k)
//   stage 2 (per BOND f, 7 multiply-adds per window row and lane):
//       X_c[i][j][n] += T_f[i][c] W_plain[j][n] + T_f[i][plain] W_c[j][n],     X_e[i][j][n] += T_f[i][plain] W_plain[j][n]
//       T_f[i] = (u_f,x B'(r_f)[i], u_f,y B'(r_f)[i], u_f,z B'(r_f)[i], B(r_f)[i]),  u_f = unit(m -> other end of f)
//
// An atom of the benchmark has 273 triplet records but only ~50 (bond, block) pairs: the expansion over the three force
// components and the window of the fixed leg -- what the matrix cores did per record -- happens once per bond.  Lanes are the
// (j, n) positions of the W window (27 for the default trims) twice over: lanes 0-31 keep the sums (x, y) of a position, lanes
// 32-63 (z, plain).  No matrix cores, no per-record staging of three evaluated legs: a record is the n-leg values (centre
// role) or those, their derivatives times a3 and the values of bond (e, k).  DESIGN.md section 3.6.
//
// Orientation: the fixed bond sits on leg l ("normal": rows i = l, window j = m) or on leg m ("transposed": i = m, j = l); a
// block has one orientation per atom species -- transposed exactly when sa != sb and the atom is of species sb (then it is
// always the SECOND neighbour, and its centre-role triplets are summed with the bond on leg m fixed instead).  Blocks with
// sa == sb must fold symmetrically in (l, m) -- column(l, m, n) = raw(l, m, n) + raw(m, l, n) -- so that which of two equal
// neighbours is "first" (angles.py:456-470, incl. the ghost-centre numbering quirk) does not matter; uf3_basis_create checks
// that and everything else this kernel assumes (Feat3 eligibility) and the other launches remain for the rest.
#pragma once

struct Feat3Leg {
    double t0, tlast, inv_h;
    int nk, row0;              // knots; number of the leg's first window row (one per knot interval from 3 on)
};

struct Feat3Args {
    const BasisDev *B;
    const TrioDev *trios;
    const double *rows;        // window rows [n_rows][18]: t_i, t_i+1 | 4 functions x (c0 c1 c2 c3), see uf3_basis_create
    int n_rows;
    const unsigned short *fsrc; // fold tables: per trio [orientation 0 | 1][column][source 0 | 1] -> index of the source bin in the fold's
                               // dump (row f of the fixed leg, position of (j, n): uf3_basis_create), or of a zero
    const int *trio_fsrc;      // [T] first entry of a trio's table
    int n_fsrc;
    Feat3Leg leg_p, leg_n;     // the centre legs (l and m share knots and window), leg n
    int lo_p, ext_p, lo_n, ext_n;
    int pr_rows;               // window rows (of the summed leg) per round of 32 positions: 32 / ext_n, at most ext_p
    const FrameGeom *geoms;
    const int *frame_of;
    N3Lists n3;
    const double *pos;
    const signed char *spec;
    double *x_e, *x_f;
    int ld;                    // doubles between consecutive force rows (>= F)
    int natoms, atoms_per_block, e_direct;
    const int *sel;            // instance selection on the device (uf3_featurize_dev launches two instances back to back when the
    int sel_mode, sel_cap;     // context's list capacity is above 16): 0 always run; 1 run iff *sel <= sel_cap; 2 iff *sel > sel_cap
    int skip;                  // ablations (-DUF3_ABLATE builds only): 1 stage 1, 2 stage 2, 4 centre walk, 8 neighbour walk, 16 fold + stores, 32 bond tables, 64 leg evaluations of the walks
    // hand-off of the neighbour role's stage-1 sums (round 6, k_feat3_w -> k_featurize3<.., HO = true>): W(e -> m) of every
    // directed bond, written by the CENTRE e into the slot of the CONSUMER m: [(m - m_lo) * n3.cap + index of e in m's list]
    // [partner species][wsz doubles]
    double *wbuf;
    int wsz;                   // doubles per (bond, partner species): window positions x (W_x, W_y, W_z, W_plain), padded to whole
                               // 128-byte lines (27 positions: 112 doubles, 7 lines)
    int m_lo;                  // first atom of the launch (the launch covers m_lo .. natoms - 1; wbuf is indexed from m_lo)
};

// Compile-time shape of a launch: EF = rows of the window on the fixed leg (= ext of the centre legs), NR = rounds of 32 W
// positions per half-wave (ext_p * ext_n <= 32 NR - 1: one position stays free as the fold's zero), the strides that follow.
template <int EF, int NR>
struct F3Cfg {
    static constexpr int PS = 32 * NR;                           // positions per fixed-leg row in the fold's dump
    static constexpr int EFP = (EF + 1) & ~1;                    // dense bond values of a neighbour-role record, padded
    static constexpr int RS_N = EFP + 16 + (NR == 1 ? 2 : 0);    // doubles per neighbour-role record: B(r_ek)[ext_p] | (a3 B_n', B_n) x 4 (| pad: 44 dwords apart, the
                                                                 // eight lanes of a 16-byte store group hit different banks)
    static constexpr int RS_C = NR == 1 ? 10 : (NR == 2 ? 12 : 14);   // ... per centre-role record: B_n over the n window (ext_n < RS_C), zero-padded
    static constexpr int NREC = NR == 1 ? 29 : 32;               // neighbour-role records per engine pass
    static constexpr int DUMP = NR == 1 ? 4 * EF * 32 : 2 * (EF * PS + 2);   // the fold's dump: pairs of all four rows | one component per half at a time
    static constexpr int STAGE0 = 64 * RS_C > NREC * RS_N ? 64 * RS_C : NREC * RS_N;
    static constexpr int STAGE = (STAGE0 > DUMP ? STAGE0 : DUMP) + 4;  // + a quad of zeros
    static constexpr int MIN_WAVES = NR == 1 ? 4 : 2;            // waves per SIMD the registers are bounded for
};

typedef double __attribute__((ext_vector_type(2))) F3Pair;
typedef const __attribute__((address_space(3))) F3Pair *F3LdsPairs;
typedef const __attribute__((address_space(3))) double *F3LdsDoubles;

// four consecutive window functions of a leg at x (t0 < x < tlast): values v, derivatives d; returns the knot interval.
// The row of the guessed interval is fetched whole and again only when the guess was off (non-uniform knots, x on a knot).
#ifndef F3_ROWS_GLOBAL
#define F3_ROWS_GLOBAL 0
#endif
typedef const __attribute__((address_space(1))) F3Pair *F3GlobalPairs;
template <bool DERIV>
__device__ __forceinline__ int f3_eval(const double *rows, const Feat3Leg &lg, double x, double (&v)[4], double (&d)[4]) {
    const int hi = lg.nk - 5;
    int iv = 3 + (int)((x - lg.t0) * lg.inv_h);
    iv = iv < 3 ? 3 : (iv > hi ? hi : iv);
    F3Pair kn, cf[8];
    auto load_row = [&](int i) {
#if F3_ROWS_GLOBAL
        F3GlobalPairs q = (F3GlobalPairs)(const F3Pair *)(rows + (size_t)(lg.row0 + i - 3) * 18);
#else
        F3LdsPairs q = (F3LdsPairs)(const F3Pair *)(rows + (size_t)(lg.row0 + i - 3) * 18);
#endif
        F3Pair t0 = q[0], t[8];
#pragma unroll
        for (int e = 0; e < 8; e++) t[e] = q[1 + e];
        kn = t0;
#pragma unroll
        for (int e = 0; e < 8; e++) cf[e] = t[e];
        __builtin_amdgcn_sched_group_barrier(F3_ROWS_GLOBAL ? 0x020 : 0x100, 9, 0);
    };
    load_row(iv);
    if (__builtin_expect(x > kn.y && iv < hi, 0)) {
        do { ++iv; load_row(iv); } while (x > kn.y && iv < hi);
    } else if (__builtin_expect(x <= kn.x && iv > 3, 0)) {
        do { --iv; load_row(iv); } while (x <= kn.x && iv > 3);
    }
    const double u = x - kn.x;
#pragma unroll
    for (int fq = 0; fq < 4; fq++) {
        const double c0 = cf[2 * fq].x, c1 = cf[2 * fq].y, c2 = cf[2 * fq + 1].x, c3 = cf[2 * fq + 1].y;
        double pv = fma(c3, u, c2), pd = fma(c3, u, pv);
        pv = fma(pv, u, c1); pd = fma(pd, u, pv);
        pv = fma(pv, u, c0);
        v[fq] = pv;
        if (DERIV) d[fq] = pd;
    }
    return iv;
}

// (CAP: the list capacity as a compile-time constant -- 16, what a tuned context settles on for bcc / fcc cells -- or 0 for
// the launch's run-time value: with it every per-wave LDS array sits at a constant offset from two bases, offsets that go into
// the LDS instructions' immediate fields instead of scalar registers the kernel has too few of)
#ifndef F3_WIDE_MINW
#define F3_WIDE_MINW 2     // waves per SIMD the wide-window instances (NR > 1) are bounded for
#endif
// HO: the neighbour role's stage-1 sums come from the hand-off buffer (k_feat3_w below ran over the same atoms before this
// launch): what is left of the role is one 16-byte load per lane and bond, and the flush
template <bool WANT_E, int EF, int NR, int CAP = 0, bool HO = false>
__global__ void __launch_bounds__(WPB * WAVE, (NR == 1 ? 4 : F3_WIDE_MINW))
k_featurize3(Feat3Args A) {
    static_assert(!HO || NR == 1, "the hand-off buffer is laid out for the one-round window");
    typedef F3Cfg<EF, NR> Cfg;
    constexpr int RS_C = Cfg::RS_C, RS_N = Cfg::RS_N, NREC = Cfg::NREC, PS = Cfg::PS, EFP = Cfg::EFP;
    extern __shared__ __align__(16) unsigned char smem[];
    const BasisDev *B = A.B;
    // (cap: what the LDS arrays are laid out for -- the longest list this instance serves; ent_stride: entries per atom in the
    // batch's list array, the context's capacity)
    const int F = load_const(&B->F), S = load_const(&B->S), n_trios = load_const(&B->T), cap = CAP > 0 ? CAP : A.n3.cap;
    const int ent_stride = A.n3.cap;
    if (A.sel_mode) {
        const int seen = load_const(A.sel);                       // the batch's longest 3-body list (the list build behind us)
        if ((seen <= A.sel_cap) != (A.sel_mode == 1)) return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- LDS carve (feat3_lds_bytes on the host) ----------------------------------------------------------------------
    double *erow = (double *)smem;
    const bool e_lds = WANT_E && !A.e_direct;
    const size_t e_d = e_lds ? (size_t)F + (F & 1) : 0;
    double *rows_lds = erow + e_d;                                    // window rows, shared
    const double *rows = F3_ROWS_GLOBAL ? A.rows : rows_lds;
    const size_t rows_d = (size_t)A.n_rows * 18;
    const size_t list_d = 5 * (size_t)cap + ((5 * cap) & 1), tq_d = (size_t)cap * EF * 4;
    constexpr size_t stage_d = Cfg::STAGE;
    const size_t per_wave_d = list_d + tq_d + stage_d;
    const size_t per_wave_i = 2 * (size_t)cap + 2 * ((size_t)cap + 1) + (UF3_MAX_SPECIES + 2) + (size_t)cap * (S + 1) + 2 * NREC + cap;
    double *wd = rows_lds + rows_d + (size_t)wave * per_wave_d;
    int *wi = (int *)(rows_lds + rows_d + (size_t)WPB * per_wave_d) + (size_t)wave * per_wave_i;
    double *ox = wd, *oy = ox + cap, *oz = oy + cap, *orr = oz + cap, *oir = orr + cap;
    double *tq = wd + list_d;                                         // [cap][EF][4]: T_f = (Tx Ty Tz T3) of every own bond
    double *stage = tq + tq_d;
    double *zq = stage + (Cfg::STAGE - 4);                            // a quad of zeros
    int *oparent = wi, *oshift = wi + cap, *noff = wi + 2 * cap, *nbase = noff + cap + 1, *so = nbase + cap + 1;
    int *osbp = so + (UF3_MAX_SPECIES + 2);                           // first window row of every own bond
    int *hdrs = osbp + cap;                       // [NREC + NREC] key | first n slot of the records of a pass
    int *ospoff = hdrs + 2 * NREC;                // [cap][S + 1] (last: the only array whose place depends on S)
    const int sp_stride = S + 1;
    unsigned short *fsrc_l = (unsigned short *)((int *)(rows_lds + rows_d + (size_t)WPB * per_wave_d) + (((size_t)WPB * per_wave_i + 3) & ~(size_t)3));

    for (int q = tid; q < (int)rows_d; q += WPB * WAVE) rows_lds[q] = A.rows[q];
    for (int q = tid; q < A.n_fsrc / 2; q += WPB * WAVE) ((int *)fsrc_l)[q] = ((const int *)A.fsrc)[q];
    if (e_lds) { for (int q = tid; q < F; q += WPB * WAVE) erow[q] = 0.0; }
    for (int q = lane; q < (int)stage_d; q += WAVE) stage[q] = 0.0;
    for (int q = lane; q < (int)tq_d; q += WAVE) tq[q] = 0.0;
    __syncthreads();

    // ---- per-lane constants of the W window: lanes 0-31 hold the sums (x, y) of a position, lanes 32-63 (z, plain); a lane
    // serves NR positions, 32 apart --------------------------------------------------------------------------------------
    const int ext_p = A.ext_p, ext_n = A.ext_n, lo_n = A.lo_n, lo_p = A.lo_p, npos = ext_p * ext_n;
    const int half = lane >> 5;
    int p_lane[NR], n_lane[NR], qn_lane[NR], n32_lane[NR];
    const int pr_rows = A.pr_rows;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        // round r holds the rows r * pr_rows ... of the summed leg: local position (row, n) = lane & 31
        const int pl = (lane & 31) / ext_n, p = r * pr_rows + pl;
        const bool on = pl < pr_rows && p < ext_p;
        p_lane[r] = on ? p : 0;
        n_lane[r] = on ? (lane & 31) - pl * ext_n : -(1 << 20);            // (idle positions always read zeros)
        qn_lane[r] = on ? n_lane[r] : RS_C - 1;                            // (... a centre-role record's zero padding)
        n32_lane[r] = n_lane[r] * 32;
    }
    (void)npos;
    const int sbn_max = ext_n > 4 ? ext_n - 4 : 0, sbp_max = ext_p > 4 ? ext_p - 4 : 0;
    const Feat3Leg leg_p = A.leg_p, leg_n = A.leg_n;
    // rounds that hold rows of a bond whose four functions start at window row sb (all of them when the window has <= 4 rows)
    auto rounds_of = [&](int sb) {
        if (EF <= 4) return (1 << NR) - 1;
        const int r0 = sb / pr_rows, r1 = min(sb + 3, ext_p - 1) / pr_rows;
        return ((2 << r1) - 1) & ~((1 << r0) - 1);
    };

    const int bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int block_first = A.m_lo + bid * A.atoms_per_block;
    const int block_end = min(block_first + A.atoms_per_block, A.natoms);
    int erow_frame = -1;
    for (int m0 = block_first; m0 < block_end; m0 += WPB) {
        const int m = m0 + wave;
        const bool active = m < block_end;
        if (e_lds) {
            int f_first = load_const(A.frame_of + m0);
            if (f_first != erow_frame) {
                __syncthreads();
                if (erow_frame >= 0)
                    for (int q = tid; q < F; q += WPB * WAVE) {
                        double v = erow[q];
                        if (v != 0.0) { unsafeAtomicAdd(A.x_e + (size_t)erow_frame * F + q, v); erow[q] = 0.0; }
                    }
                __syncthreads();
                erow_frame = f_first;
            }
        }
        if (!active) continue;
        const int fr = load_const(A.frame_of + m);
        const int sm = ((const __attribute__((address_space(4))) signed char *)(unsigned long long)A.spec)[m];
        ESink es;
        es.lds = erow; es.glob = WANT_E ? A.x_e + (size_t)fr * F : nullptr; es.direct = !e_lds || (fr != erow_frame);

        // ---- own 3-body list -> LDS; species offsets of the neighbours' lists ----------------------------------------
        const int n_own = load_const(A.n3.cnt + m);
        wave_sync();
        for (int e = lane; e < n_own; e += WAVE) {
            const N3Entry en = A.n3.ent[(size_t)m * ent_stride + e];
            ox[e] = en.dx; oy[e] = en.dy; oz[e] = en.dz; orr[e] = en.r; oir[e] = 1.0 / en.r;
            int s0, s1, s2;
            unpack3(en.shiftc, s0, s1, s2);
            oparent[e] = en.parent; oshift[e] = pack3(-s0, -s1, -s2);      // (m's image as the neighbour's list names it)
        }
        if (lane <= S) so[lane] = A.n3.spoff[(size_t)m * (UF3_MAX_SPECIES + 1) + lane];
        wave_sync();
        if (!HO)
            for (int e = lane; e < n_own; e += WAVE) {
                const int *src = A.n3.spoff + (size_t)oparent[e] * (UF3_MAX_SPECIES + 1);
                for (int sp = 0; sp <= S; sp++) ospoff[e * (S + 1) + sp] = src[sp];
            }
        // ---- T_f of every own bond: rows over the whole window of the centre legs, zero where the bond's four functions are
        // not (and all zero when the bond is outside the legs' range) -------------------------------------------------------
        for (int e = lane; e < n_own; e += WAVE) {
            const double r = orr[e];
            double v[4] = {0, 0, 0, 0}, d[4] = {0, 0, 0, 0};
            int sbp = 0;
            if (r > leg_p.t0 && r < leg_p.tlast && !UF3_SKIP(32)) {
                const int iv = f3_eval<true>(rows, leg_p, r, v, d);
                sbp = max(0, min(sbp_max, iv - 3 - lo_p));
            }
            if (EF > 4) osbp[e] = sbp;
            const double ir = oir[e], ux = ox[e] * ir, uy = oy[e] * ir, uz = oz[e] * ir;
            double *dst = tq + (size_t)e * EF * 4;
            if (EF > 4) {
#pragma unroll
                for (int q = 0; q < EF; q++) { *(double2 *)(dst + 4 * q) = double2{0.0, 0.0}; *(double2 *)(dst + 4 * q + 2) = double2{0.0, 0.0}; }
                dst += 4 * sbp;
            }
#pragma unroll
            for (int q = 0; q < (EF < 4 ? EF : 4); q++) {
                *(double2 *)(dst + 4 * q) = double2{ux * d[q], uy * d[q]};
                *(double2 *)(dst + 4 * q + 2) = double2{uz * d[q], v[q]};
            }
        }
        wave_sync();

        for (int t = 0; t < n_trios; t++) {
            const TrioDev *td = A.trios + t;
            typedef int int8_v __attribute__((ext_vector_type(8)));
            const int8_v hv = *(const __attribute__((address_space(4))) int8_v *)(unsigned long long)&td->head;
            const int t_ncol = hv[2], t_sc = hv[3], t_sa = hv[4], t_sb = hv[5], t_col = hv[6];
            const bool centre = t_sc == sm, nbr = t_sa == sm || t_sb == sm;
            if (!centre && !nbr) { zero_rows(A.x_f, m, A.ld, t_col, t_ncol); continue; }
            const bool tr = t_sa != t_sb && sm == t_sb;              // transposed: the fixed bond sits on leg m

            double xacc[NR][EF][2], ws[NR][2];
#pragma unroll
            for (int r = 0; r < NR; r++) {
                ws[r][0] = ws[r][1] = 0.0;
#pragma unroll
                for (int q = 0; q < EF; q++) { xacc[r][q][0] = xacc[r][q][1] = 0.0; }
            }
            int cur = -1;                                           // the bond whose W is being summed (own-list index)

            // stage 2: the open bond's W into the rows of the window.  Lower half: rows (x, y) += (Tx, Ty) W_plain + T3 (W_x, W_y);
            // upper half: rows (z, energy) += (Tz, T3) W_plain + T3 (W_z, 0) -- the energy row from the centre role only.
            int wmask = 0;                                          // rounds that took records since the last flush
            auto flush = [&](bool is_c) {
                if (cur < 0 || UF3_SKIP(2)) return;
                const double *tc = tq + (size_t)cur * EF * 4;
                // (wide windows: only the rounds that were summed into)
                F3Pair tv[EF];
                double t3[EF];
#pragma unroll
                for (int q = 0; q < EF; q++) {
                    tv[q] = *(F3LdsPairs)(const F3Pair *)(tc + 4 * q + 2 * half); t3[q] = ((F3LdsDoubles)tc)[4 * q + 3];
                }
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    if (EF > 4 && !((wmask >> r) & 1)) continue;
                    const int lo = __double2loint(ws[r][1]), hi = __double2hiint(ws[r][1]);
                    const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
                    const double wpl = __hiloint2double(r1[1], r0[1]);        // W_plain (the upper half's second sum) in every lane
                    const double wpl1 = (half && !is_c) ? 0.0 : wpl, ws1 = half ? 0.0 : ws[r][1];
#pragma unroll
                    for (int q = 0; q < EF; q++) {
                        xacc[r][q][0] = fma(tv[q].x, wpl, fma(t3[q], ws[r][0], xacc[r][q][0]));
                        xacc[r][q][1] = fma(tv[q].y, wpl1, fma(t3[q], ws1, xacc[r][q][1]));
                    }
                    ws[r][0] = ws[r][1] = 0.0;
                }
                wmask = 0;
            };

            // ---- centre role: m centres (f, g), f on the fixed leg.  Items p = fi * nG + gi, one per lane and stage slot (not
            // compacted: a void item is a record of zeros); the records of a fixed bond are a run of slots and the T rows of
            // their partners g a run of the table: stage 1 reads both at constant offsets from the run's first record.
            if (centre && !UF3_SKIP(4)) {
                const int fs = tr ? t_sb : t_sa, gs = tr ? t_sa : t_sb;
                const int f_lo = so[fs], nF = so[fs + 1] - f_lo, g_lo = so[gs], nG = so[gs + 1] - g_lo;
                const bool same = t_sa == t_sb;
                const int total = __builtin_amdgcn_readfirstlane(nF * nG);
                const int nGs = __builtin_amdgcn_readfirstlane(nG), f_los = __builtin_amdgcn_readfirstlane(f_lo),
                          g_los = __builtin_amdgcn_readfirstlane(g_lo);
                const float rcp_ng = __builtin_amdgcn_rcpf((float)max(nG, 1));
                for (int p0 = 0; p0 < total; p0 += WAVE) {
                    const int p = p0 + lane;
                    bool valid = p < total;
                    int fi = (int)(((float)p + 0.5f) * rcp_ng);
                    fi -= (fi * nG > p) ? 1 : 0;
                    fi += ((fi + 1) * nG <= p) ? 1 : 0;
                    const int gi = p - fi * nG;
                    valid = valid && (!same || gi > fi);
                    const int f = f_lo + fi, gg = g_lo + gi;
                    double bn[4] = {0, 0, 0, 0}, dum[4];
                    int sbn = 0;
                    if (valid) {
                        const double rf = orr[f], rg = orr[gg];
                        const double ex = ox[gg] - ox[f], ey = oy[gg] - oy[f], ez = oz[gg] - oz[f];
                        const double rn = norm3_leg(ex, ey, ez);
                        valid = (rf > leg_p.t0) & (rf < leg_p.tlast) & (rg > leg_p.t0) & (rg < leg_p.tlast) &
                                (rn > leg_n.t0) & (rn < leg_n.tlast);
                        if (valid && !UF3_SKIP(64)) {
                            const int iv = f3_eval<false>(rows, leg_n, rn, bn, dum);
                            sbn = max(0, min(sbn_max, iv - 3 - lo_n));
                        }
                    }
                    int smask = (1 << NR) - 1;                      // rounds the partners g of this step have rows in
                    if (EF > 4) {
                        const int rm = valid ? rounds_of(osbp[gg]) : 0;
                        smask = 0;
#pragma unroll
                        for (int r = 0; r < NR; r++) smask |= __ballot((rm >> r) & 1) ? (1 << r) : 0;
                    }
                    {
                        double *rp = stage + (size_t)lane * RS_C;
#pragma unroll
                        for (int u = 0; u < RS_C; u += 2) *(double2 *)(rp + u) = double2{0.0, 0.0};
                        if (valid) {
#pragma unroll
                            for (int u = 0; u < 4; u++) rp[sbn + u] = bn[u];
                        }
                    }
                    wave_sync();
                    // rows fi of the item rectangle that this step touches
                    const int p1 = min(total, p0 + WAVE);
                    int fi_s = p0 / nGs;
                    for (int pr = fi_s * nGs; pr < p1 && !UF3_SKIP(1); pr += nGs, fi_s++) {
                        int s0 = max(pr, p0), s1 = min(pr + nGs, p1);               // items of this row inside the step
                        if (same) s0 = max(s0, pr + fi_s + 1);
                        if (s0 >= s1) continue;
                        const int key = f_los + fi_s;
                        if (key != cur) { flush(true); cur = key; }
                        wmask |= smask;
                        const double *qp = stage + (size_t)(s0 - p0) * RS_C;
                        const double *tp = tq + (size_t)(g_los + (s0 - pr)) * EF * 4 + 2 * half;
                        int cnt = s1 - s0;
                        auto body = [&](auto tag) {
                            constexpr int CNT = decltype(tag)::value;
#pragma unroll
                            for (int r = 0; r < NR; r++) {
                                if (EF > 4 && !((smask >> r) & 1)) continue;
                                double bq[CNT];
                                F3Pair tt[CNT];
#pragma unroll
                                for (int i = 0; i < CNT; i++) {
                                    bq[i] = ((F3LdsDoubles)qp)[i * RS_C + qn_lane[r]];
                                    tt[i] = *(F3LdsPairs)(const F3Pair *)(tp + (size_t)i * EF * 4 + 4 * p_lane[r]);
                                }
#pragma unroll
                                for (int i = 0; i < CNT; i++) { ws[r][0] = fma(tt[i].x, bq[i], ws[r][0]); ws[r][1] = fma(tt[i].y, bq[i], ws[r][1]); }
                            }
                            qp += CNT * RS_C; tp += (size_t)CNT * EF * 4; cnt -= CNT;
                        };
                        while (cnt >= 4) body(std::integral_constant<int, 4>{});
                        if (cnt == 3) body(std::integral_constant<int, 3>{});
                        else if (cnt == 2) body(std::integral_constant<int, 2>{});
                        else if (cnt == 1) body(std::integral_constant<int, 1>{});
                    }
                    wave_sync();
                }
                flush(true);
                cur = -1;
            }
            // ---- neighbour role: m is a neighbour of the centre e (fixed bond (m, e)); k runs over e's list.  The valid items of
            // a step are compacted into the stage in order (runs of a fixed bond are found with one ballot); what a lane needs of
            // a record besides its operands -- the first n slot -- it reads itself (no per-record work on the scalar unit).
            if (HO && nbr && !UF3_SKIP(8)) {
                // W(e -> m) of every own bond to a centre of this block, as that centre's k_feat3_w left it: consecutive bonds are
                // consecutive slots of the buffer; four loads in flight, then their flushes
                const int sx = sm == t_sa ? t_sb : t_sa;
                const int rc_lo = __builtin_amdgcn_readfirstlane(so[t_sc]), ncen = __builtin_amdgcn_readfirstlane(so[t_sc + 1]) - rc_lo;
                const bool lane_on = n_lane[0] >= 0;
                const size_t wstep = (size_t)S * A.wsz;
                const double *wb = A.wbuf + (((size_t)(m - A.m_lo) * ent_stride + rc_lo) * S + sx) * A.wsz + (lane & 31) * 4 + 2 * half;
                for (int e0 = 0; e0 < ncen; e0 += 4) {
                    F3Pair w[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int idx = min(e0 + i, ncen - 1);
                        w[i] = lane_on ? __builtin_nontemporal_load((const F3Pair *)(wb + (size_t)idx * wstep)) : F3Pair{0.0, 0.0};
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (e0 + i < ncen) {
                            cur = rc_lo + e0 + i;
                            ws[0][0] = w[i].x; ws[0][1] = w[i].y;
                            flush(false);
                        }
                }
                cur = -1;
            }
            if (!HO && nbr && !UF3_SKIP(8)) {
                const int sx = sm == t_sa ? t_sb : t_sa;
                const int rc_lo = so[t_sc], ncen = so[t_sc + 1] - rc_lo;
                int total_n = 0;
                for (int e0 = 0; e0 < ncen; e0 += WAVE) {
                    const int e = e0 + lane;
                    int cnt = 0, base = 0;
                    if (e < ncen) {
                        const int *sp = ospoff + (size_t)(rc_lo + e) * sp_stride;
                        base = sp[sx]; cnt = sp[sx + 1] - base;
                    }
                    const int incl = wave_scan_incl(cnt);
                    if (e < ncen) { noff[e] = total_n + incl - cnt; nbase[e] = base; }
                    total_n += __builtin_amdgcn_readlane(incl, WAVE - 1);
                }
                total_n = __builtin_amdgcn_readfirstlane(total_n);
                if (lane == 0) noff[ncen] = total_n;
                wave_sync();
                for (int p0 = 0; p0 < total_n; p0 += WAVE) {
                    const int q = p0 + lane;
                    bool valid = q < total_n;
                    double pv[4] = {0, 0, 0, 0}, bn[4] = {0, 0, 0, 0}, bd[4] = {0, 0, 0, 0}, a3[3] = {0, 0, 0}, dum[4];
                    int sbn = 0, sbp = 0, e = 0;
                    if (valid) {
                        int lo = 0, hi = ncen - 1;
                        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (noff[mid] <= q) lo = mid; else hi = mid - 1; }
                        e = rc_lo + lo;
                        const int kk = nbase[lo] + (q - noff[lo]);
                        const int pc = oparent[e];
                        const double oex = ox[e], oey = oy[e], oez = oz[e], oer = orr[e];
                        const N3Entry ke = A.n3.ent[(size_t)pc * ent_stride + kk];
                        valid = !(ke.parent == m && ke.shiftc == oshift[e]);                  // k is m itself
                        const double ex = oex + ke.dx, ey = oey + ke.dy, ez = oez + ke.dz;   // m -> k
                        const double rn = norm3_leg(ex, ey, ez), rk = ke.r;
                        valid = valid & (oer > leg_p.t0) & (oer < leg_p.tlast) & (rk > leg_p.t0) & (rk < leg_p.tlast) &
                                (rn > leg_n.t0) & (rn < leg_n.tlast);
                        if (valid && !UF3_SKIP(64)) {
                            const double in = fast_rcp(rn);
                            a3[0] = ex * in; a3[1] = ey * in; a3[2] = ez * in;
                            const int ivp = f3_eval<false>(rows, leg_p, rk, pv, dum);
                            sbp = max(0, min(sbp_max, ivp - 3 - lo_p));
                            const int iv = f3_eval<true>(rows, leg_n, rn, bn, bd);
                            sbn = max(0, min(sbn_max, iv - 3 - lo_n));
                        }
                    }
                    const unsigned long long mask = __ballot(valid);
                    const int nv = __popcll(mask), rank = mbcnt(mask);
                    int smask = (1 << NR) - 1;                      // rounds the bonds (e, k) of this step have rows in
                    if (EF > 4) {
                        const int rm = valid ? rounds_of(sbp) : 0;
                        smask = 0;
#pragma unroll
                        for (int r = 0; r < NR; r++) smask |= __ballot((rm >> r) & 1) ? (1 << r) : 0;
                    }
                    for (int sp0 = 0; sp0 < nv; sp0 += NREC) {
                        const int slot = rank - sp0;
                        if (valid && slot >= 0 && slot < NREC) {
                            double *rp = stage + (size_t)slot * RS_N;
                            if (EF > 4) {
#pragma unroll
                                for (int u = 0; u < EFP; u += 2) *(double2 *)(rp + u) = double2{0.0, 0.0};
#pragma unroll
                                for (int u = 0; u < 4; u++) rp[sbp + u] = pv[u];
                            } else {
                                *(double2 *)rp = double2{pv[0], pv[1]};
                                *(double2 *)(rp + 2) = double2{pv[2], pv[3]};
                            }
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                *(double2 *)(rp + EFP + 4 * u) = double2{a3[0] * bd[u], a3[1] * bd[u]};
                                *(double2 *)(rp + EFP + 2 + 4 * u) = double2{a3[2] * bd[u], bn[u]};
                            }
                            hdrs[slot] = e; hdrs[NREC + slot] = sbn * 32;      // (byte offset of the first n slot's quad)
                        }
                        wave_sync();
                        const int r_end = min(NREC, nv - sp0);
                        const int v_key = hdrs[min(lane, NREC - 1)], v_prev = hdrs[min(max(lane - 1, 0), NREC - 1)];
                        unsigned long long gm = __ballot(lane < r_end && (lane == 0 || v_key != v_prev));
                        while (gm && !UF3_SKIP(1)) {
                            const int g0 = __builtin_ctzll(gm);
                            gm &= gm - 1;
                            const int g1 = gm ? __builtin_ctzll(gm) : r_end;
                            const int key = __builtin_amdgcn_readlane(v_key, g0);
                            if (key != cur) { flush(false); cur = key; }
                            wmask |= smask;
                            const double *rp = stage + (size_t)g0 * RS_N;
                            const int *sp = hdrs + NREC + g0;
                            int cnt = g1 - g0;
                            auto body = [&](auto tag) {
                                constexpr int CNT = decltype(tag)::value;
                                int sb[CNT];
#pragma unroll
                                for (int i = 0; i < CNT; i++) sb[i] = ((const __attribute__((address_space(3))) int *)sp)[i];
#pragma unroll
                                for (int r = 0; r < NR; r++) {
                                    if (EF > 4 && !((smask >> r) & 1)) continue;
                                    double bq[CNT];
                                    F3Pair tt[CNT];
#pragma unroll
                                    for (int i = 0; i < CNT; i++) {
                                        const unsigned off = (unsigned)(n32_lane[r] - sb[i]);          // 32 (n - first slot)
                                        const char *qa = (const char *)(rp + i * RS_N + EFP + 2 * half) + off;
                                        qa = off < 128u ? qa : (const char *)zq;
                                        bq[i] = ((F3LdsDoubles)rp)[i * RS_N + p_lane[r]];
                                        tt[i] = *(F3LdsPairs)(const F3Pair *)qa;
                                    }
#pragma unroll
                                    for (int i = 0; i < CNT; i++) { ws[r][0] = fma(bq[i], tt[i].x, ws[r][0]); ws[r][1] = fma(bq[i], tt[i].y, ws[r][1]); }
                                }
                                rp += CNT * RS_N; sp += CNT; cnt -= CNT;
                            };
                            while (cnt >= 4) body(std::integral_constant<int, 4>{});
                            if (cnt == 3) body(std::integral_constant<int, 3>{});
                            else if (cnt == 2) body(std::integral_constant<int, 2>{});
                            else if (cnt == 1) body(std::integral_constant<int, 1>{});
                        }
                        wave_sync();
                    }
                }
                flush(false);
                cur = -1;
            }
            if (UF3_SKIP(16)) continue;
            // ---- fold: the rows of the window -> LDS, the block's columns sum their (one or two) source bins -----------------
            wave_sync();
            const unsigned short *ft = fsrc_l + load_const(A.trio_fsrc + t) + (size_t)(tr ? 2 * t_ncol : 0);
            if (NR == 1) {
                // (pairs: a lane's two rows side by side -- [half][f][32 positions] x (x, y | z, energy))
#pragma unroll
                for (int q = 0; q < EF; q++)
                    *(double2 *)(stage + (size_t)(((half * EF + q) * 32 + (lane & 31)) * 2)) = double2{xacc[0][q][0], xacc[0][q][1]};
                wave_sync();
                for (int col = lane; col < t_ncol; col += WAVE) {
                    const int s0 = ft[2 * col], s1 = ft[2 * col + 1];
                    F3LdsPairs d0 = (F3LdsPairs)(const F3Pair *)stage, d1 = d0 + EF * 32;
                    const F3Pair a0 = d0[s0], a1 = d0[s1], b0 = d1[s0], b1 = d1[s1];
                    double *dst = A.x_f + (size_t)m * 3 * A.ld + t_col + col;
                    __builtin_nontemporal_store(a0.x + a1.x, dst); __builtin_nontemporal_store(a0.y + a1.y, dst + A.ld);
                    __builtin_nontemporal_store(b0.x + b1.x, dst + 2 * (size_t)A.ld);
                    if (WANT_E) es.add(t_col + col, b0.y + b1.y);
                }
                wave_sync();
            } else {
                // one component of each half at a time -- (x | z), then (y | energy): two buffers [f][32 NR positions] of doubles, a
                // zero behind each for the second source a column may not have
                constexpr int BUF = EF * PS + 2;
#pragma unroll
                for (int k = 0; k < 2; k++) {
#pragma unroll
                    for (int r = 0; r < NR; r++)
#pragma unroll
                        for (int q = 0; q < EF; q++) stage[half * BUF + q * PS + r * 32 + (lane & 31)] = xacc[r][q][k];
                    if (lane == 0 || lane == 32) stage[half * BUF + EF * PS] = 0.0;
                    wave_sync();
                    for (int col = lane; col < t_ncol; col += WAVE) {
                        const int s0 = ft[2 * col], s1 = ft[2 * col + 1];
                        F3LdsDoubles da = (F3LdsDoubles)stage, db = da + BUF;
                        const double va = da[s0] + da[s1];
                        __builtin_nontemporal_store(va, A.x_f + ((size_t)m * 3 + k) * A.ld + t_col + col);
                        if (k == 0 || WANT_E) {
                            const double vb = db[s0] + db[s1];
                            if (k == 0) __builtin_nontemporal_store(vb, A.x_f + ((size_t)m * 3 + 2) * A.ld + t_col + col);
                            else es.add(t_col + col, vb);
                        }
                    }
                    wave_sync();
                }
            }
        }
    }
    if (e_lds) {
        __syncthreads();
        if (erow_frame >= 0)
            for (int q = tid; q < F; q += WPB * WAVE) {
                double v = erow[q];
                if (v != 0.0) unsafeAtomicAdd(A.x_e + (size_t)erow_frame * F + q, v);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_feat3_w (round 6) -- the neighbour role's stage-1 sums, computed ONCE, at the centre.
//
// k_featurize3's neighbour role has atom m rebuild, for every bond (m, e) to a centre e, the sums
//     W_s[j][n] = sum over e's neighbours k != m of  B(r_ek)[j] * (a3 B_n'(r_mk), B_n(r_mk))_s[n],     a3 = unit(m -> k)
// by walking e's list -- the same (bond, partner) rectangle that e's own centre role enumerates.  Here the centre e does it:
// its ordered pairs (f, g), f the bond that stays fixed (the future consumer), g the partner, items p = fi * nG + gi of the
// rectangle one per lane, uncompacted (a void item is a record of zeros), a record DENSE over the n window (a3 B_n' | B_n at
// every n slot: 9 quads, zeros outside the four functions of the interval) so that stage 1 reads record and bond table at
// constant offsets -- no clamp arithmetic, no compaction, no headers.  The sums of a fixed bond f leave as ONE 16-byte store per
// lane into the slot of the consumer: [f's atom][index of e in f's list][species of the partners] (Feat3Args::wbuf), found with
// a reverse look-up in the atom's prologue.  k_featurize3<.., HO = true> reads them back and keeps only the flush.
// Reference arithmetic: angles.py:142-286 (the terms of featurize_force_3b that differentiate leg n, and the n-leg factor of
// the terms that differentiate the centre leg of the neighbour), :517-632.
#ifndef F3W_NRECP
#define F3W_NRECP 32       // records per stage pass (a step of 64 items is 64 / F3W_NRECP passes)
#endif
struct F3WCfg {
    static constexpr int RS_W = 38;                 // doubles per record: 9 n slots x (a3x B', a3y B', a3z B', B) | pad: 76 dwords apart, so that
                                                    // the sixteen lanes of a 16-byte store group fall on different banks (72: eight-way conflicts)
    static constexpr int NRECP = F3W_NRECP;
    static constexpr int STAGE = NRECP * RS_W;
};

template <int EF, int CAP = 0>
__global__ void __launch_bounds__(WPB * WAVE, 4)
k_feat3_w(Feat3Args A) {
    static_assert(EF <= 4, "one-round windows only");
    constexpr int RS_W = F3WCfg::RS_W, NRECP = F3WCfg::NRECP;
    extern __shared__ __align__(16) unsigned char smem[];
    const BasisDev *B = A.B;
    const int S = load_const(&B->S), n_trios = load_const(&B->T), cap = CAP > 0 ? CAP : A.n3.cap;
    const int ent_stride = A.n3.cap;
    if (A.sel_mode) {
        const int seen = load_const(A.sel);
        if ((seen <= A.sel_cap) != (A.sel_mode == 1)) return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- LDS carve (feat3w_lds_bytes on the host): window rows (shared) | per wave: list (4 cap) | bond values [cap][4] | stage
    // | ints: parent, shift, reverse index, list length of the neighbour (cap each), species offsets ------------------------------
    double *rows_lds = (double *)smem;
    const double *rows = rows_lds;
    const size_t rows_d = (size_t)A.n_rows * 18;
    const size_t per_wave_d = 8 * (size_t)cap + F3WCfg::STAGE;
    const size_t per_wave_i = 4 * (size_t)cap + (UF3_MAX_SPECIES + 2);
    double *wd = rows_lds + rows_d + (size_t)wave * per_wave_d;
    int *wi = (int *)(rows_lds + rows_d + (size_t)WPB * per_wave_d) + (size_t)wave * per_wave_i;
    double *ox = wd, *oy = ox + cap, *oz = oy + cap, *orr = oz + cap;
    double *tb = wd + 4 * cap;                                        // [cap][4]: B(r)[window row] of every own bond
    double *stage = tb + 4 * cap;
    int *oparent = wi, *oshift = wi + cap, *orev = wi + 2 * cap, *ocnt = wi + 3 * cap, *so = wi + 4 * cap;

    for (int q = tid; q < (int)rows_d; q += WPB * WAVE) rows_lds[q] = A.rows[q];
    for (int q = lane; q < F3WCfg::STAGE; q += WAVE) stage[q] = 0.0;
    __syncthreads();

    const int ext_p = A.ext_p, ext_n = A.ext_n, lo_n = A.lo_n;
    const int half = lane >> 5;
    const int pl = (lane & 31) / ext_n;
    const bool lane_on = pl < ext_p;
    const int p_lane = lane_on ? pl : 0, n_lane = lane_on ? (lane & 31) - pl * ext_n : 0;     // (idle lanes sum what they like; never stored)
    const int sbn_max = ext_n > 4 ? ext_n - 4 : 0;
    const Feat3Leg leg_p = A.leg_p, leg_n = A.leg_n;
    const size_t wslot = (size_t)(lane & 31) * 4 + 2 * half;

    const int bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int block_first = A.m_lo + bid * A.atoms_per_block;
    const int block_end = min(block_first + A.atoms_per_block, A.natoms);
    for (int m0 = block_first; m0 < block_end; m0 += WPB) {
        const int m = m0 + wave;
        if (m >= block_end) continue;
        const int sm = ((const __attribute__((address_space(4))) signed char *)(unsigned long long)A.spec)[m];
        const int n_own = load_const(A.n3.cnt + m);
        wave_sync();
        for (int e = lane; e < n_own; e += WAVE) {
            const N3Entry en = A.n3.ent[(size_t)m * ent_stride + e];
            ox[e] = en.dx; oy[e] = en.dy; oz[e] = en.dz; orr[e] = en.r;
            int s0, s1, s2;
            unpack3(en.shiftc, s0, s1, s2);
            oparent[e] = en.parent; oshift[e] = pack3(-s0, -s1, -s2);      // (this atom's image as the neighbour's list names it)
            ocnt[e] = A.n3.cnt[en.parent];
            orev[e] = -1;
            double v[4] = {0, 0, 0, 0}, d[4];
            if (en.r > leg_p.t0 && en.r < leg_p.tlast) f3_eval<false>(rows, leg_p, en.r, v, d);
            *(double2 *)(tb + 4 * e) = double2{v[0], v[1]};
            *(double2 *)(tb + 4 * e + 2) = double2{v[2], EF > 3 ? v[3] : 0.0};
        }
        if (lane <= S) so[lane] = A.n3.spoff[(size_t)m * (UF3_MAX_SPECIES + 1) + lane];
        wave_sync();
        // ---- where this atom sits in each neighbour's list (the slot its sums go to): lanes = (bond, entry of the neighbour's list)
        for (int it = lane; it < n_own * cap; it += WAVE) {
            const int f = it / cap, q = it - f * cap;
            if (q < ocnt[f]) {
                const int2 ps = *(const int2 *)&A.n3.ent[(size_t)oparent[f] * ent_stride + q].parent;
                if (ps.x == m && ps.y == oshift[f]) orev[f] = q;
            }
        }
        wave_sync();

        for (int t = 0; t < n_trios; t++) {
            const TrioDev *td = A.trios + t;
            typedef int int8_v __attribute__((ext_vector_type(8)));
            const int8_v hv = *(const __attribute__((address_space(4))) int8_v *)(unsigned long long)&td->head;
            const int t_sc = hv[3], t_sa = hv[4], t_sb = hv[5];
            if (t_sc != sm) continue;
            const bool same = t_sa == t_sb;
            for (int o = 0; o < (same ? 1 : 2); o++) {
                // fixed bonds f of species fs (the consumers), partners g of species gs
                const int fs = o ? t_sb : t_sa, gs = o ? t_sa : t_sb;
                const int f_lo = __builtin_amdgcn_readfirstlane(so[fs]), nF = __builtin_amdgcn_readfirstlane(so[fs + 1]) - f_lo;
                const int g_lo = __builtin_amdgcn_readfirstlane(so[gs]), nG = __builtin_amdgcn_readfirstlane(so[gs + 1]) - g_lo;
                if (nF <= 0) continue;
                double ws0 = 0.0, ws1 = 0.0;
                int cur = -1;
                auto store_w = [&]() {
                    if (cur >= 0) {
                        const int rv = orev[cur], pf = oparent[cur];
                        if (rv >= 0 && lane_on && pf >= A.m_lo) {
                            double *dst = A.wbuf + (((size_t)(pf - A.m_lo) * ent_stride + rv) * S + gs) * A.wsz + wslot;
                            *(F3Pair *)dst = F3Pair{ws0, ws1};
                        }
                    }
                    ws0 = ws1 = 0.0;
                };
                if (nG <= 0) {                          // no partners: the consumers read zeros
                    for (int fi = 0; fi < nF; fi++) { cur = f_lo + fi; store_w(); }
                    continue;
                }
                const int total = nF * nG;
                const float rcp_ng = __builtin_amdgcn_rcpf((float)nG);
                for (int p0 = 0; p0 < total; p0 += WAVE) {
                    const int p = p0 + lane;
                    bool valid = p < total;
                    int fi = (int)(((float)p + 0.5f) * rcp_ng);
                    fi -= (fi * nG > p) ? 1 : 0;
                    fi += ((fi + 1) * nG <= p) ? 1 : 0;
                    const int gi = p - fi * nG;
                    valid = valid && (!same || gi != fi);
                    double bn[4] = {0, 0, 0, 0}, bd[4] = {0, 0, 0, 0}, a3[3] = {0, 0, 0};
                    int sbn = 0;
                    if (valid) {
                        const int f = f_lo + fi, gg = g_lo + gi;
                        const double rf = orr[f], rg = orr[gg];
                        const double ex = ox[gg] - ox[f], ey = oy[gg] - oy[f], ez = oz[gg] - oz[f];     // f -> g
                        const double rn = norm3_leg(ex, ey, ez);
                        valid = (rf > leg_p.t0) & (rf < leg_p.tlast) & (rg > leg_p.t0) & (rg < leg_p.tlast) &
                                (rn > leg_n.t0) & (rn < leg_n.tlast);
                        if (valid) {
                            const double in = fast_rcp(rn);
                            a3[0] = ex * in; a3[1] = ey * in; a3[2] = ez * in;
                            const int iv = f3_eval<true>(rows, leg_n, rn, bn, bd);
                            sbn = max(0, min(sbn_max, iv - 3 - lo_n));
                        }
                    }
                    const int p1 = min(total, p0 + WAVE);
                    for (int q0 = p0; q0 < p1; q0 += NRECP) {
                        const int slot = p - q0;
                        const bool mine = valid && slot >= 0 && slot < NRECP;
                        double *rp = stage + (size_t)(slot & (NRECP - 1)) * RS_W + 4 * sbn;
                        if (mine) {
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                *(double2 *)(rp + 4 * u) = double2{a3[0] * bd[u], a3[1] * bd[u]};
                                *(double2 *)(rp + 4 * u + 2) = double2{a3[2] * bd[u], bn[u]};
                            }
                        }
                        wave_sync();
                        const int q1 = min(q0 + NRECP, p1);
                        int fi_s = q0 / nG;
                        for (int pr = fi_s * nG; pr < q1; pr += nG, fi_s++) {
                            const int s0 = max(pr, q0), s1 = min(pr + nG, q1);             // items of this fixed bond inside the pass
                            const int key = f_lo + fi_s;
                            if (key != cur) { store_w(); cur = key; }
                            const double *qp = stage + (size_t)(s0 - q0) * RS_W + 4 * n_lane + 2 * half;
                            const double *tp = tb + (size_t)(g_lo + (s0 - pr)) * 4 + p_lane;
                            int cnt = s1 - s0;
                            auto body = [&](auto tag) {
                                constexpr int CNT = decltype(tag)::value;
                                double bq[CNT];
                                F3Pair tt[CNT];
#pragma unroll
                                for (int i = 0; i < CNT; i++) {
                                    bq[i] = ((F3LdsDoubles)tp)[i * 4];
                                    tt[i] = *(F3LdsPairs)(const F3Pair *)(qp + (size_t)i * RS_W);
                                }
#pragma unroll
                                for (int i = 0; i < CNT; i++) { ws0 = fma(bq[i], tt[i].x, ws0); ws1 = fma(bq[i], tt[i].y, ws1); }
                                qp += CNT * RS_W; tp += CNT * 4; cnt -= CNT;
                            };
                            while (cnt >= 4) body(std::integral_constant<int, 4>{});
                            if (cnt == 3) body(std::integral_constant<int, 3>{});
                            else if (cnt == 2) body(std::integral_constant<int, 2>{});
                            else if (cnt == 1) body(std::integral_constant<int, 1>{});
                        }
                        wave_sync();
                        if (mine) {                     // the record's four quads back to zero: the stage stays clean between passes
#pragma unroll
                            for (int u = 0; u < 8; u++) *(double2 *)(rp + 2 * u) = double2{0.0, 0.0};
                        }
                    }
                }
                store_w();
            }
        }
    }
}
