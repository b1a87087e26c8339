// uf3_kernels.h -- the gfx950 kernels of the UF3 hot path.
//
//   k_frame_bins      atom -> (frame, wrapped fractional bin)            [HBM-trivial]
//   k_bin_fill / k_bin_finish   cell list: counting sort of the atoms by bin (32-B slot records in bin order);
//                     k_prepare_small: the whole stage of an MD-step batch in one workgroup
//   k_build_n3        per-atom 3-body neighbour lists with image shifts, sorted by
//                     (species, reference supercell index)               one wave / atom (evaluator path)
//   k_build_n3_ext    extension lists for batches with atoms far outside their cell (reference's image range)
//   k_featurize<E, F, R, MODE, IMG>   energy row + 3 force rows per atom   one wave / atom, one launch per block family:
//                     MODE 0      one-body + pair columns; also builds the 3-body lists from its candidates
//                     MODE 6 / 7  3-body windows of <= 32 rows on the fp64 matrix cores (7: three waves / SIMD)
//                     MODE 8 / 9  ... of <= 64 / <= 128 rows
//                     MODE 1-5    generic output-stationary 3-body kernels (wider windows)
//   k_featurize3<E, EF, NR>   (uf3_feat3.h) the 3-body force rows of a basis that qualifies (mode bit 12), by bond
//                     factorisation on the vector units: the launch the headline runs; the matrix-core modes above
//                     then serve energy-only calls, batches with atoms far outside their cell and the other bases
//   k_eval<GATHER, VIR, CAP, MD, TAB>   energy + forces (+ virial) of a fitted model     one wave / atom: every triplet once at its
//                     centre + k_eval_collect (whole batch, a block of centres), or gathered at its three atoms (a block of
//                     atoms).  MD: the candidates from the context's persistent lists (k_build_sup + k_sup_reverse, built with
//                     a skin) instead of a cell-list walk, the neighbours' force shares through a stamped inbox
//                     (k_eval_collect_md / k_eval_collect_md_halo).  TAB: the centre legs of the triplets from per-bond tables,
//                     leg n's knot records in LDS: four waves per SIMD
//   k_md_fetch        a small batch's staged block out of the caller's pinned memory (MD steps: no cell-list kernel to do it)
//   k_frame_sum       per-frame sums of the per-atom energies / virial shares; the last kernel of an evaluator call: status
//                     words and the call's sequence number into the pinned block the host polls
//   k_gram_tiled / k_gram_mfma   X^T X on the fp64 matrix cores, X^T y riding along in the diagonal workgroups / waves
//
// Formulation: every atom m GATHERS all pair terms and all triplet terms it takes part in --
// as centre, or as one of the two neighbours of a centre c in N3(m) -- so its three force-feature rows are complete
// when its wave is done and are written exactly once, coalesced, with no global atomics.  A triplet is visited by
// each of its three atoms; translation invariance makes the three visits the three slices of the reference's
// arrange_deriv_3b (angles.py:235-286) output.
#pragma once
#include "uf3_device.h"

// ---------------------------------------------------------------------------------
// cell list
// ---------------------------------------------------------------------------------
__global__ void k_frame_bins(const BasisDev *B, const FrameGeom *geoms, const int64_t *atom_offsets,
                             int n_frames, int natoms, const double *pos, const int32_t *z,
                             int *frame_of, int *atom_bin, int *atom_wrap, signed char *spec,
                             int *sort_key, int *bin_count, int *err_flag) {
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
    bool far_out = false;
    for (int k = 0; k < 3; k++) {
        double f = x * g.inv[k] + y * g.inv[3 + k] + w * g.inv[6 + k];
        far_out |= f < g.win_lo[k] || f > g.win_hi[k];
        if (g.per[k]) {
            double fl = floor(f);
            int b = (int)((f - fl) * g.nb[k]);
            bin[k] = b >= g.nb[k] ? g.nb[k] - 1 : (b < 0 ? 0 : b);
            wrap[k] = (int)fl;
            if (wrap[k] < -250 || wrap[k] > 250) { atomicExch(err_flag, 1); wrap[k] = 0; }
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
    if (far_out) err_flag[4] = 1;                                   // an atom far outside its cell (see TrioWalk::img_check)
    sort_key[a] = g.bin_base + lb;
    atomicAdd(bin_count + g.bin_base + lb, 1);
}

// Counting sort of the atoms by global bin (k_frame_bins counted the bins, an exclusive scan turned the counts into
// bin_start): k_bin_fill drops every atom into its bin in whatever order the atomics grant, k_bin_finish -- one thread per
// bin, a bin holds a handful of atoms -- puts each bin into ascending atom order and writes the slot records.  The result is
// the stable radix sort's (bin, then atom index), with 5 launches instead of rocPRIM's 25 for 320 k atoms.
// exclusive prefix sums of up to 16 384 bin counts in ONE workgroup (1024 threads, a run of consecutive counts each, wave scans on the
// DPP network): the library scan's two launches (look-back state, scan) are two dispatch latencies on a call whose every launch is
// a few microseconds -- a 10 k-atom evaluator call is ten of them in front of the centre pass
__global__ void __launch_bounds__(1024)
k_scan_small(const int *cnt, int *start, int n) {
    __shared__ int wsum[16];
    const int t = threadIdx.x, per = (n + 1023) >> 10, lo = t * per, hi = min(lo + per, n);
    int s = 0;
    for (int q = lo; q < hi; q++) s += cnt[q];
    const int incl = wave_scan_incl(s);
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    if (t < 64) {
        const int v = t < 16 ? wsum[t] : 0;
        const int w = wave_scan_incl(v);
        if (t < 16) wsum[t] = w - v;                    // exclusive over the waves
    }
    __syncthreads();
    int run = wsum[t >> 6] + incl - s;
    for (int q = lo; q < hi; q++) { const int c = cnt[q]; start[q] = run; run += c; }
}

__global__ void k_bin_fill(const int *atom_key, int natoms, const int *bin_start, int *bin_count, int *sorted_val) {
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= natoms) return;
    const int b = atom_key[a];
    const int p = atomicSub(bin_count + b, 1) - 1;            // (the counts are back at zero afterwards)
    sorted_val[bin_start[b] + p] = a;
}

__global__ void k_bin_finish(int nbins, const int *bin_start, int *sorted_val, const double *pos, const int *atom_wrap,
                             const signed char *spec, SlotRec *slots) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbins) return;
    const int s0 = bin_start[b], n = bin_start[b + 1] - s0;
    for (int i = 1; i < n; i++) {                             // insertion sort, ascending atom index
        const int v = sorted_val[s0 + i];
        int j = i - 1;
        while (j >= 0 && sorted_val[s0 + j] > v) { sorted_val[s0 + j + 1] = sorted_val[s0 + j]; --j; }
        sorted_val[s0 + j + 1] = v;
    }
    for (int i = 0; i < n; i++) {
        const int a = sorted_val[s0 + i];
        SlotRec r;
        r.x = pos[3 * (size_t)a]; r.y = pos[3 * (size_t)a + 1]; r.z = pos[3 * (size_t)a + 2];
        r.atom = a;
        int w0, w1, w2;
        unpack3(atom_wrap[a], w0, w1, w2);
        r.ws = pack_ws(w0, w1, w2, spec[a]);
        slots[s0 + i] = r;
    }
}

// The whole cell-list stage of a SMALL batch (<= UF3_SMALL_ATOMS atoms: an MD step) in one workgroup: frame / bin /
// wrap / species per atom, a bitonic sort of (global bin << 32 | atom) in LDS -- the keys are distinct, so the order is
// the stable radix sort's -- then bin starts and slot records.  One launch instead of seven.
#define UF3_SMALL_ATOMS 2048
__global__ void __launch_bounds__(1024)
k_prepare_small(const BasisDev *B, const FrameGeom *geoms, const int64_t *atom_offsets, int n_frames, int natoms,
                int nbins, const double *pos, const int32_t *z, int *frame_of, int *atom_bin, int *atom_wrap,
                signed char *spec, int *bin_start, SlotRec *slots, int *flags, int n_zero_flags,
                const int4 *host_block, int4 *dev_block, int block_int4s) {
    __shared__ unsigned long long keys[UF3_SMALL_ATOMS];
    const int tid = threadIdx.x, nt = blockDim.x;        // (256 threads for <= 256 atoms: cheaper barriers; 1024 otherwise)
    // host_block != null: positions | species | frame geometry | offsets of the batch still sit in the caller's pinned staging
    // block (device-visible host memory): this workgroup fetches them itself -- a few KB over the link -- instead of waiting for
    // a copy engine's transfer in front of it (one dependent step of an MD call less).  pos, z, geoms, atom_offsets point into
    // dev_block, where the later kernels read them.
    if (n_frames == 1 && natoms <= 256 && nt >= natoms) {
        // One cell of an MD step, one atom per thread, every fetch of the kernel in flight at once: the staging block's copy
        // for the later kernels, the thread's own atom and the frame geometry straight from the staging block, the species
        // table from the basis; then one barrier, the bins, a rank sort (every thread counts the keys below its own -- distinct
        // keys: the rank is the slot) and the slot record written from the registers the atom arrived in.  Three dependent
        // memory round trips less than the general path below.
        __shared__ __align__(8) int geom_words[(sizeof(FrameGeom) + 3) / 4];      // (read back as a FrameGeom: doubles inside)
        __shared__ __align__(8) int z2s_words[30];
        const long long to_host = host_block ? (const char *)host_block - (const char *)dev_block : 0;
        const double *hpos = (const double *)((const char *)pos + to_host);
        const int32_t *hz = (const int32_t *)((const char *)z + to_host);
        const int *hgeom = (const int *)((const char *)geoms + to_host);
        double x = 0, y = 0, w = 0;
        int zz = 0;
        if (tid < natoms) { x = hpos[3 * tid]; y = hpos[3 * tid + 1]; w = hpos[3 * tid + 2]; zz = hz[tid]; }
        if (tid < (int)(sizeof(FrameGeom) / 4)) geom_words[tid] = hgeom[tid];
        if (tid < 30) z2s_words[tid] = ((const int *)B->z2s)[tid];
        if (host_block) for (int q = tid; q < block_int4s; q += nt) dev_block[q] = host_block[q];
        if (tid < n_zero_flags) flags[1 + tid] = 0;
        if (tid == 0) { flags[3] = 0; flags[4] = 0; flags[5] = 0; flags[12] = 0; }      // ([5]: md_build's list-length report; [12]: k_frame_sum's count of finished workgroups -- left over by an aborted call otherwise)
        __syncthreads();
        const FrameGeom &g = *(const FrameGeom *)geom_words;
        unsigned long long key = ~0ull;
        bool outside = false;
        int ws = 0;
        if (tid < natoms) {
            frame_of[tid] = 0;
            int sp = (zz >= 0 && zz < 120) ? ((const signed char *)z2s_words)[zz] : -1;
            if (sp < 0) { atomicExch(flags, 2); sp = 0; }
            spec[tid] = (signed char)sp;
            int bin[3], wrap[3];
            for (int k = 0; k < 3; k++) {
                double f = x * g.inv[k] + y * g.inv[3 + k] + w * g.inv[6 + k];
                outside |= f < g.win_lo[k] || f > g.win_hi[k];
                if (g.per[k]) {
                    double fl = floor(f);
                    int b = (int)((f - fl) * g.nb[k]);
                    bin[k] = b >= g.nb[k] ? g.nb[k] - 1 : (b < 0 ? 0 : b);
                    wrap[k] = (int)fl;
                    if (wrap[k] < -250 || wrap[k] > 250) { atomicExch(flags, 1); wrap[k] = 0; }
                } else {
                    long long q = (long long)floor(f / g.binw[k]);
                    int b = (int)(q % g.nb[k]);
                    bin[k] = b < 0 ? b + g.nb[k] : b;
                    wrap[k] = 0;
                }
            }
            const int lb = (bin[0] * g.nb[1] + bin[1]) * g.nb[2] + bin[2];
            atom_bin[tid] = lb;
            atom_wrap[tid] = pack3(wrap[0], wrap[1], wrap[2]);
            ws = pack_ws(wrap[0], wrap[1], wrap[2], sp);
            key = ((unsigned long long)(unsigned)(g.bin_base + lb) << 32) | (unsigned)tid;
        }
        keys[tid] = key;
        __syncthreads();
        if (outside) flags[4] = 1;                                // (after a barrier: thread 0 has zeroed it)
        int rank = 0;
        for (int b = 0; b < natoms; b++) rank += keys[b] < key;
        __shared__ unsigned long long sorted[256];
        if (tid < natoms) {
            sorted[rank] = key;
            SlotRec r;
            r.x = x; r.y = y; r.z = w; r.atom = tid; r.ws = ws;
            slots[rank] = r;
        }
        __syncthreads();
        for (int b = tid; b <= nbins; b += nt) {                  // first slot with bin >= b
            int lo = 0, hi = natoms;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if ((int)(sorted[mid] >> 32) < b) lo = mid + 1; else hi = mid;
            }
            bin_start[b] = lo;
        }
        return;
    }
    if (host_block) {
        for (int q = tid; q < block_int4s; q += nt) dev_block[q] = host_block[q];
        __syncthreads();
    }
    if (tid < n_zero_flags) flags[1 + tid] = 0;                   // (n3_need, cand_need of the launches that follow)
    if (tid == 0) { flags[3] = 0; flags[4] = 0; flags[5] = 0; flags[12] = 0; }  // (extension-list need, "some atom outside its cell", k_frame_sum's workgroup count)
    bool outside = false;
    int n_pow2 = 64;
    while (n_pow2 < natoms) n_pow2 <<= 1;
    for (int a = tid; a < n_pow2; a += nt) {
        unsigned long long key = ~0ull;
        if (a < natoms) {
            int lo = 0, hi = n_frames - 1;                // frame with atom_offsets[f] <= a < atom_offsets[f+1]
            while (lo < hi) {
                int mid = (lo + hi + 1) >> 1;
                if (atom_offsets[mid] <= a) lo = mid; else hi = mid - 1;
            }
            const FrameGeom &g = geoms[lo];
            frame_of[a] = lo;
            int zz = z[a];
            int s = (zz >= 0 && zz < 120) ? B->z2s[zz] : -1;
            if (s < 0) { atomicExch(flags, 2); s = 0; }
            spec[a] = (signed char)s;
            double x = pos[3 * (size_t)a], y = pos[3 * (size_t)a + 1], w = pos[3 * (size_t)a + 2];
            int bin[3], wrap[3];
            for (int k = 0; k < 3; k++) {
                double f = x * g.inv[k] + y * g.inv[3 + k] + w * g.inv[6 + k];
                outside |= f < g.win_lo[k] || f > g.win_hi[k];
                if (g.per[k]) {
                    double fl = floor(f);
                    int b = (int)((f - fl) * g.nb[k]);
                    bin[k] = b >= g.nb[k] ? g.nb[k] - 1 : (b < 0 ? 0 : b);
                    wrap[k] = (int)fl;
                    if (wrap[k] < -250 || wrap[k] > 250) { atomicExch(flags, 1); wrap[k] = 0; }
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
            key = ((unsigned long long)(unsigned)(g.bin_base + lb) << 32) | (unsigned)a;
        }
        keys[a] = key;
    }
    __syncthreads();
    if (outside) flags[4] = 1;                                    // (after the barrier: thread 0 has zeroed it)
    if (natoms <= 256) {
        // an MD-step cell: rank sort -- every thread counts the keys below its own (distinct keys: the rank is the slot) with
        // broadcast reads, one barrier instead of the bitonic network's log^2 n
        __shared__ unsigned long long sorted[256];
        const unsigned long long mine = tid < natoms ? keys[tid] : ~0ull;
        int rank = 0;
        for (int b = 0; b < natoms; b++) rank += keys[b] < mine;
        if (tid < natoms) sorted[rank] = mine;
        __syncthreads();
        if (tid < natoms) keys[tid] = sorted[tid];
        __syncthreads();
    } else
    for (int size = 2; size <= n_pow2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < n_pow2 / 2; t += nt) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;          // the pair (i, i + stride) of this stage
                const bool up = (i & size) == 0;
                const unsigned long long a = keys[i], b = keys[j];
                if ((a > b) == up) { keys[i] = b; keys[j] = a; }
            }
            __syncthreads();
        }
    for (int sidx = tid; sidx < natoms; sidx += nt) {
        const int a = (int)(unsigned)keys[sidx];
        SlotRec r;
        r.x = pos[3 * (size_t)a]; r.y = pos[3 * (size_t)a + 1]; r.z = pos[3 * (size_t)a + 2];
        r.atom = a;
        int w0, w1, w2;
        unpack3(atom_wrap[a], w0, w1, w2);                 // (written by this thread block above; same-block visibility
        r.ws = pack_ws(w0, w1, w2, spec[a]);               //  through the barriers)
        slots[sidx] = r;
    }
    for (int b = tid; b <= nbins; b += nt) {             // first slot with bin >= b
        int lo = 0, hi = natoms;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if ((int)(keys[mid] >> 32) < b) lo = mid + 1; else hi = mid;
        }
        bin_start[b] = lo;
    }
}

// vector from atom m (original position pm) to the image (slot, shift) of a neighbour
__device__ __forceinline__ void image_delta(const FrameGeom &g, const SlotRec &sr, int s0, int s1, int s2,
                                            const double *pm, double &dx, double &dy, double &dz) {
    double off[3];
    for (int k = 0; k < 3; k++) off[k] = s0 * g.cell[k] + s1 * g.cell[3 + k] + s2 * g.cell[6 + k];
    // (p_j + offset) - p_i, as the reference tiles positions first (geometry.py:146-148)
    dx = (sr.x + off[0]) - pm[0];
    dy = (sr.y + off[1]) - pm[1];
    dz = (sr.z + off[2]) - pm[2];
}

// ---------------------------------------------------------------------------------
// 3-body neighbour lists: one wave per atom
// ---------------------------------------------------------------------------------
// atom of workgroup blockIdx.x: first + blockIdx.x; or, when marks are given (which [natoms], n_which = {block begin, block
// end} on the device), the blockIdx.x-th atom outside that block if it is marked: the halo of a block of atoms
__global__ void __launch_bounds__(64)
k_build_n3(const BasisDev *B, const FrameGeom *geoms, const int *frame_of, CellList cl, N3Lists n3,
           const double *pos, int natoms, int *overflow_need, int first, const int *which, const int *n_which) {
    extern __shared__ __align__(16) unsigned char smem[];
    int cap = n3.cap;
    unsigned long long *key = (unsigned long long *)smem;
    double *ex = (double *)(key + cap), *ey = ex + cap, *ez = ey + cap, *er = ez + cap;
    int *eparent = (int *)(er + cap), *eshift = eparent + cap, *esidx = eshift + cap, *espec = esidx + cap;

    int m = first + blockIdx.x;
    if (which) {
        // `which` = halo marks [natoms], n_which = {block begin, block end}: workgroup b takes the b-th atom OUTSIDE the block and
        // leaves unless the block's lists mention it
        const int lo = n_which[0], hi = n_which[1];
        m = (int)blockIdx.x < lo ? (int)blockIdx.x : (int)blockIdx.x + (hi - lo);
        if (m >= natoms || !which[m]) return;
    }
    if (m >= natoms) return;
    int lane = lane_id();
    const FrameGeom g = geoms[frame_of[m]];
    double pm[3] = {pos[3 * (size_t)m], pos[3 * (size_t)m + 1], pos[3 * (size_t)m + 2]};
    double rmin3 = B->rmin3, rmax3 = B->rmax3;
    int count = 0;
    for_each_candidate(g, cl, m, [&](bool ok, const SlotRec &sr, int sj, int s0, int s1, int s2) {
        double dx = 0, dy = 0, dz = 0, d = 0;
        if (ok) {
            image_delta(g, sr, s0, s1, s2, pm, dx, dy, dz);
            d = norm3_rn(dx, dy, dz);
            ok = (d > rmin3) && (d <= rmax3);            // angles.py:340: lower strict, upper inclusive
        }
        unsigned long long mask = __ballot(ok);
        if (ok) {
            int e = count + mbcnt(mask);
            if (e < cap) {
                int j = sr.atom;
                int sidx = supercell_index(g, s0, s1, s2, j - g.atom_lo);
                int sp = sj;
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
    for (int sp = lane; sp <= UF3_MAX_SPECIES; sp += WAVE) {     // entries are species-sorted: offsets per species
        int below = 0;
        for (int f = 0; f < count; f++) below += espec[f] < sp;
        n3.spoff[(size_t)m * (UF3_MAX_SPECIES + 1) + sp] = below;
    }
    size_t base = (size_t)m * cap;
    for (int e = lane; e < count; e += WAVE) {     // rank sort by (species, supercell index)
        unsigned long long k = key[e];
        int rank = 0;
        for (int f = 0; f < count; f++) rank += key[f] < k;
        N3Entry out;
        out.dx = ex[e]; out.dy = ey[e]; out.dz = ez[e]; out.r = er[e];
        out.parent = eparent[e]; out.shiftc = eshift[e]; out.sidx = esidx[e]; out.spec = espec[e];
        n3.ent[base + rank] = out;
    }
}


// Extension lists (see N3Lists): for a batch that holds atoms outside their cell, the 3-body neighbours of every atom
// whose image index lies in (fac, 2 fac] on some axis.  A ghost image c + s_c of the atom takes its neighbours from the
// reference's supercell, i.e. by ABSOLUTE image index |s_c + s_k| <= fac: entries of the atom's own list can fall out of
// that range (TrioWalk::img_check drops them) and entries beyond the atom's own range can fall into it -- these.  Leaves at
// once when no atom of the batch lies outside its cell (flags[4] == 0: every batch in practice).  Sorted by (species,
// image, atom); flags[3] reports the capacity needed.
__global__ void __launch_bounds__(64)
k_build_n3_ext(const BasisDev *B, const FrameGeom *geoms, const int *frame_of, CellList cl, N3Lists n3,
               const double *pos, int natoms, int *flags) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (flags[4] == 0) return;
    const int cap = n3.xcap;
    unsigned long long *key = (unsigned long long *)smem;
    double *ex = (double *)(key + cap), *ey = ex + cap, *ez = ey + cap, *er = ez + cap;
    int *eparent = (int *)(er + cap), *eshift = eparent + cap, *espec = eshift + cap;
    const int lane = lane_id();
    const double rmin3 = B->rmin3, rmax3 = B->rmax3;
    for (int m = blockIdx.x; m < natoms; m += gridDim.x) {
        const FrameGeom g = geoms[frame_of[m]];
        double pm[3] = {pos[3 * (size_t)m], pos[3 * (size_t)m + 1], pos[3 * (size_t)m + 2]};
        int count = 0;
        for_each_candidate(g, cl, m, [&](bool ok, const SlotRec &sr, int sj, int s0, int s1, int s2) {
            double dx = 0, dy = 0, dz = 0, d = 0;
            ok = ok && (abs(s0) > g.fac[0] || abs(s1) > g.fac[1] || abs(s2) > g.fac[2]);
            if (ok) {
                image_delta(g, sr, s0, s1, s2, pm, dx, dy, dz);
                d = norm3_rn(dx, dy, dz);
                ok = (d > rmin3) && (d <= rmax3);
            }
            unsigned long long mask = __ballot(ok);
            if (ok) {
                int e = count + mbcnt(mask);
                if (e < cap) {
                    key[e] = ((unsigned long long)sj << 58) | ((unsigned long long)(unsigned)pack3(s0, s1, s2) << 28) |
                             (unsigned)(sr.atom - g.atom_lo);
                    ex[e] = dx; ey[e] = dy; ez[e] = dz; er[e] = d;
                    eparent[e] = sr.atom; eshift[e] = pack3(s0, s1, s2); espec[e] = sj;
                }
            }
            count += __popcll(mask);
        }, 2);
        __syncthreads();
        if (count > cap) { if (lane == 0) atomicMax(flags + 3, count); count = cap; }
        if (cap > 0) {
            for (int sp = lane; sp <= UF3_MAX_SPECIES; sp += WAVE) {
                int below = 0;
                for (int f = 0; f < count; f++) below += espec[f] < sp;
                n3.xoff[(size_t)m * (UF3_MAX_SPECIES + 1) + sp] = below;
            }
            for (int e = lane; e < count; e += WAVE) {
                unsigned long long k = key[e];
                int rank = 0;
                for (int f = 0; f < count; f++) rank += key[f] < k;
                N3Entry out;
                out.dx = ex[e]; out.dy = ey[e]; out.dz = ez[e]; out.r = er[e];
                out.parent = eparent[e]; out.shiftc = eshift[e]; out.sidx = -1; out.spec = espec[e];
                n3.xent[(size_t)m * cap + rank] = out;
            }
        }
        __syncthreads();
    }
}


// Halo of a block of atoms [lo, hi): the atoms outside the block that appear in a block atom's 3-body list (the gather route
// walks their lists, the centre route's collection pass serves them).  mark[] must be zero on entry.
__global__ void __launch_bounds__(256)
k_mark_halo(N3Lists n3, int lo, int hi, int *mark, int *range) {
    const int m = lo + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) { range[0] = lo; range[1] = hi; }
    if (m >= hi) return;
    const int n = min(n3.cnt[m], n3.cap);
    for (int e = threadIdx.x & 63; e < n; e += 64) {
        const int p = n3.ent[(size_t)m * n3.cap + e].parent;
        if (p < lo || p >= hi) mark[p] = 1;                       // (plain stores of the same value: no order to keep)
    }
}

// three buffers zeroed by one launch (the decomposed evaluator's force rows, list counts and halo marks: three fills and
// their launch gaps were a tenth of a rank's share at world 8)
__global__ void __launch_bounds__(256)
k_zero3(unsigned *p0, size_t n0, unsigned *p1, size_t n1, unsigned *p2, size_t n2) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1 + n2; i += stride) {
        if (i < n0) p0[i] = 0u;
        else if (i < n0 + n1) p1[i - n0] = 0u;
        else p2[i - n0 - n1] = 0u;
    }
}

// ---------------------------------------------------------------------------------
// MD route: persistent superset lists (a Verlet list with a skin).  One entry per neighbour image within r_search + skin of
// the atom at build time: who it is (atom, image shift relative to the ORIGINAL positions, reference supercell index,
// species) -- no geometry, which every step recomputes from the current positions exactly as the cell-list walk would.
// Entries are sorted by (species, supercell index): a step filters them by the true distances and compacts the survivors
// IN THAT ORDER, so the pairs an atom sums and its 3-body list are the same sequence whatever superset they were drawn from --
// results do not depend on when the lists were last rebuilt (calculator.py:124-153 rebuilds everything every call).
// ---------------------------------------------------------------------------------
struct __attribute__((aligned(16))) SupEntry { int parent, shiftc, sidx, spec; };

__global__ void __launch_bounds__(64)
k_build_sup(const BasisDev *B, const FrameGeom *geoms, const int *frame_of, CellList cl, const double *pos, int natoms,
            double r_sup2, SupEntry *ent, int *cnt, int cap, int *overflow_need) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long *key = (unsigned long long *)smem;
    int *eparent = (int *)(key + cap), *eshift = eparent + cap;
    const int m = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    if (m >= natoms) return;
    const int lane = lane_id();
    const FrameGeom g = geoms[frame_of[m]];
    const double pm[3] = {pos[3 * (size_t)m], pos[3 * (size_t)m + 1], pos[3 * (size_t)m + 2]};
    int count = 0;
    for_each_candidate(g, cl, m, [&](bool ok, const SlotRec &sr, int sj, int s0, int s1, int s2) {
        if (ok) {
            double dx, dy, dz;
            image_delta(g, sr, s0, s1, s2, pm, dx, dy, dz);
            ok = norm3_sq_rn(dx, dy, dz) <= r_sup2;
        }
        const unsigned long long mask = __ballot(ok);
        if (ok) {
            const int e = count + mbcnt(mask);
            if (e < cap) {
                key[e] = ((unsigned long long)sj << 32) | (unsigned)supercell_index(g, s0, s1, s2, sr.atom - g.atom_lo);
                eparent[e] = sr.atom; eshift[e] = pack3(s0, s1, s2);
            }
        }
        count += __popcll(mask);
    });
    __syncthreads();
    if (count > cap) { if (lane == 0) atomicMax(overflow_need, count); count = cap; }
    if (lane == 0) cnt[m] = count;
    for (int e = lane; e < count; e += WAVE) {       // rank sort by (species, supercell index): distinct keys
        const unsigned long long k = key[e];
        int rank = 0;
        for (int f = 0; f < count; f++) rank += key[f] < k;
        SupEntry out;
        out.parent = eparent[e]; out.shiftc = eshift[e]; out.sidx = (int)(unsigned)k; out.spec = (int)(k >> 32);
        ent[(size_t)m * cap + rank] = out;
    }
}

// where an atom sits in each of its neighbours' lists: entry q of atom m (neighbour j at image shift s) gets, packed above its
// species, 1 + the index of (m at shift -s) in j's list (0: not there -- the two ends of a pair at the very edge of the build
// radius may round differently).  What a step's centre pass needs to hand a neighbour its share of the triplet forces directly
// (k_eval<MD>: md_inbox) instead of leaving it to be looked for (k_eval_collect).  16 lanes per atom.
__global__ void __launch_bounds__(256)
k_sup_reverse(SupEntry *ent, const int *cnt, int cap, int natoms, const FrameGeom *geoms, const int *frame_of, const signed char *spec,
              const int *overflow_need, int *hard_flag) {
    const int wg = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    // (a build whose verdict the host did not wait for: lists that overflowed their capacity are clipped -- the step behind
    // this launch must be discarded like one whose atoms outran the skin, and the repeat builds with the host looking)
    if (hard_flag && blockIdx.x == 0 && threadIdx.x == 0 && *overflow_need > cap) *hard_flag = 1;
    const int m = wg * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (m >= natoms) return;
    const int n = min(cnt[m], cap);
    const FrameGeom &g = geoms[frame_of[m]];
    const int sm = spec[m], m_local = m - g.atom_lo;
    for (int q = sub; q < n; q += 16) {
        SupEntry *mine = ent + (size_t)m * cap + q;
        const int j = mine->parent;
        int s0, s1, s2;
        unpack3(mine->shiftc, s0, s1, s2);
        // the neighbour's list is sorted by (species, supercell index): m at the opposite shift has a known key there -- a
        // binary search (six probes) instead of a scan of the list (58 entries: 1.3 GB of reads per build at 50 k atoms)
        const int back = pack3(-s0, -s1, -s2), nj = min(cnt[j], cap);
        const long long want = ((long long)sm << 32) | (unsigned)supercell_index(g, -s0, -s1, -s2, m_local);
        const SupEntry *theirs = ent + (size_t)j * cap;
        int lo = 0, hi = nj;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const int2 k = *(const int2 *)&theirs[mid].sidx;                      // (sidx, spec | position bits written by other lanes: the species byte is stable)
            const long long key = ((long long)(k.y & 0xff) << 32) | (unsigned)k.x;
            if (key < want) lo = mid + 1; else hi = mid;
        }
        int hit = -1;
        if (lo < nj) {
            const int2 id = *(const int2 *)&theirs[lo].parent;
            if (id.x == m && id.y == back) hit = lo;
        }
        mine->spec = (mine->spec & 0xff) | ((hit + 1) << 8);
    }
}

// a small batch's staged block (positions | species, in the caller's pinned memory) into device memory, and the status words
// of the launches behind it zeroed: what k_prepare_small does on the way for calls that build a cell list
__global__ void __launch_bounds__(256)
k_md_fetch(const int4 *host_block, int4 *dev_block, int block_int4s, int *flags) {
    for (int q = threadIdx.x; q < block_int4s; q += blockDim.x) dev_block[q] = host_block[q];
    if (threadIdx.x < 6) flags[1 + threadIdx.x] = 0;
    if (threadIdx.x == 0) flags[12] = 0;
}

// Self-test of the BAR staging path (uf3_ctx: bar_self_test): the block the HOST has just stored into, read the two
// ways the kernels read it -- vector loads (a lane per word) and scalar loads through the constant address space (the first words,
// as load_const reads struct fields) -- into this round's slab of `out`, then the round's number into a pinned word the host polls:
// the same no-stream-sync hand-shake as the evaluator's MD steps.  A stale cache line shows as an earlier round's pattern.
__global__ void __launch_bounds__(256)
k_bar_probe(const unsigned *block, int n_words, unsigned *out, unsigned *done_word, unsigned round) {
    for (int q = threadIdx.x; q < n_words; q += blockDim.x) out[q] = block[q];
    if (threadIdx.x == 0) {
        typedef const __attribute__((address_space(4))) unsigned *ConstWords;
        ConstWords sc = (ConstWords)(unsigned long long)block;
        for (int q = 0; q < 16 && q < n_words; q++) out[n_words + q] = sc[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(done_word, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// what an MD step reads besides the lists -- frame geometry | offsets, frame and species of every atom, the positions of the
// build -- into the context's persistent copies: one launch instead of four device-to-device copies
__global__ void __launch_bounds__(256)
k_md_snapshot(const unsigned *geo_src, unsigned *geo_dst, size_t geo_words, const unsigned *fo_src, unsigned *fo_dst, size_t natoms,
              const signed char *sp_src, signed char *sp_dst, const unsigned *pos_src, unsigned *pos_dst) {
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < geo_words; i += step) geo_dst[i] = geo_src[i];
    for (size_t i = i0; i < natoms; i += step) { fo_dst[i] = fo_src[i]; sp_dst[i] = sp_src[i]; }
    for (size_t i = i0; i < 6 * natoms; i += step) pos_dst[i] = pos_src[i];
}

// largest list length of a batch (capacity tuning after the first build of a context)
__global__ void k_max_count(const int *cnt, int n, int *out) {
    int v = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v = max(v, cnt[i]);
    for (int sh = 32; sh > 0; sh >>= 1) v = max(v, __shfl_xor(v, sh));
    if ((threadIdx.x & 63) == 0) atomicMax(out, v);
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
#ifndef WPB
#define WPB 4             // independent waves per workgroup (they share only the energy row)
#endif
#define NSTAGE 32         // records staged in LDS per wave at a time
#define CAND_STRIDE 6     // doubles per 2-body candidate: dx, dy, dz, d | species | int2 {atom, packed image shift}
#define ITEM_STRIDE 38    // doubles per staged triplet record (16-B aligned)
// triplet record (doubles): 0-7 (Bl,B'l)[4], 8-15 (Bm,B'm)[4], 16-23 (Bn,B'n)[4],
//   24-26 A1, 27-29 A2, 30-32 A3, 34-35 int4 {first l, first m, first n, centre flag}, 36-37 zero pair
// pair record (doubles): 0-7 (B,B')[4], 8-10 2*(R_j-R_m)/r, 11 {first basis index, -}

// Ablation switches (experiments only: the library built with -DUF3_ABLATE reads UF3_DEBUG_SKIP; tools/ablate_counters.sh):
// 1 two-body, 2 centre role, 4 neighbour role, 8 MFMA steps, 16 leg evaluation + staging, 32 row stores, 128 the grouped
// fold, 256 its energy adds.  Compiled out otherwise -- eighteen tests of a scalar, six of them per staging pass.
#ifdef UF3_ABLATE
#define UF3_SKIP(bit) (A.skip & (bit))
#else
#define UF3_SKIP(bit) 0
#endif

// optional in-kernel phase timers (experiments only: -DUF3_PHASE_TIMING); lane 0 of every wave adds the
// cycles it spent in a phase to a global table
#ifdef UF3_PHASE_TIMING
__device__ unsigned long long g_phase[16];
struct PhaseClock {
    unsigned long long t;
    __device__ __forceinline__ PhaseClock() : t(__builtin_amdgcn_s_memtime()) {}
    __device__ __forceinline__ void lap(int phase) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // charge outstanding loads to this phase
        unsigned long long n = __builtin_amdgcn_s_memtime();
        if (lane_id() == 0) atomicAdd(&g_phase[phase], n - t);
        t = __builtin_amdgcn_s_memtime();
    }
};
#else
struct PhaseClock { __device__ __forceinline__ void lap(int) {} };
#endif

// Read-only tables (descriptors, knot records ...) never change while a kernel runs.  Through a plain global
// pointer the compiler must use vector loads it cannot hoist or merge (the kernels also store to global memory);
// through the constant address space a wave-uniform address becomes scalar loads into SGPRs.
template <class T>
__device__ __forceinline__ T load_const(const T *p) {
    static_assert(sizeof(T) % sizeof(int) == 0, "word-sized tables only");
    typedef const __attribute__((address_space(4))) int *ConstInts;
    ConstInts src = (ConstInts)(unsigned long long)p;
    T out;
    int *dst = (int *)&out;
#pragma unroll
    for (int q = 0; q < (int)(sizeof(T) / sizeof(int)); q++) dst[q] = src[q];
    return out;
}

struct FeatArgs {
    const BasisDev *B;
    const TrioDev *trios;     // explicit global pointers (no flat loads through the struct)
    const KnotRec *recs;
    const int *colsrc;        // per trio [ncol][nsrc] packed (l | m<<8 | n<<16) raw bins feeding a column, -1 pad
    const int *frag;          // v_mfma_f64_16x16x4 accumulator layout: [lane][4] -> (row, col), from k_mfma_probe
    const int *dsrc;          // dense trios: colsrc entries as offsets (pair * 16 + n bin) into the dumped window, -1 pad
    int n_dsrc, dsrc_lds;     // table length; staged in LDS by the MFMA specialisation when dsrc_lds != 0
    const unsigned short *gsrc;   // grouped windows: per column six double indices into the dumped group tiles, (source 0 | 1) x
                                  // (group 0 | 1 | 2), 15 (a zero) where the combination does not apply; block offset in
                                  // TrioHead::grouped >> 8
    int n_gsrc, gsrc_lds;         // table length (shorts); copied to LDS by the matrix-core launch when gsrc_lds != 0
    const FrameGeom *geoms;
    const int *frame_of;
    CellList cl;
    N3Lists n3;
    const double *pos;
    const signed char *spec;
    double *x_e;        // [n_frames][F] or null
    double *x_f;        // [natoms][3][ld] or null
    int ld;             // doubles between consecutive force rows (>= F; F: dense rows)
    int *cand_need;     // overflow report of the 2-body candidate stage
    const int *outside; // != 0: some atom of the batch lies outside its cell (null: the image-range rule is switched off)
    int *n3_need;       // ... of the 3-body neighbour lists (MODE 0 builds them when build_n3 != 0)
    int *n3_seen;       // or null: the longest list of the batch (k_featurize3's instance is picked by it on the device)
    int build_n3;
    int e_direct;       // energy row too long for LDS: every contribution goes straight to HBM (global atomics)
    int natoms, atoms_per_block;
    int cand_cap;       // 2-body candidates staged per atom
    int n_recs;         // KnotRec count (for the LDS copy)
    int n_pair_recs;    // ... of which belong to the pair blocks (they come first)
    int wrow_base, n_wrows;   // window rows of the grouped layouts: where they start (16-byte units from the first knot record), how many
    int trio_rec_lo;    // first record a trio leg refers to (>= n_pair_recs unless a trio leg shares a pair's knot sequence)
    int n_pair_cols;    // columns of all pair blocks together (they follow the S one-body columns)
    int dense_stage;    // doubles of per-wave stage the MFMA specialisation needs (max over dense trios)
    int dense_nrec;     // records staged per pass by the MFMA specialisation (<= DENSE_NREC)
    int skip;           // profiling ablations (UF3_DEBUG_SKIP): 1 two-body, 2 centre role, 4 neighbour role,
                        // 8 MFMA steps, 16 leg evaluation + staging, 32 row stores (mode 6)
};

__device__ __forceinline__ void lds_add(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS traffic inside one wave is in order; only the compiler has to be held back
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// where a wave adds energy-row contributions: the block's LDS row, or (atoms of a frame other than
// the one the LDS row currently holds) straight to HBM
struct ESink {
    double *lds;     // [F]
    double *glob;    // x_e row of this atom's frame
    bool direct;
    __device__ __forceinline__ void add(int col, double v) const {
        if (v == 0.0) return;
        if (direct) unsafeAtomicAdd(glob + col, v); else lds_add(lds + col, v);
    }
};

// One triplet seen from atom m.  rl, rm, rn: leg lengths in the trio's (l, m, n) order;
// a1/a2/a3: -(d r_leg / d R_m) for the three legs (zero vector when the leg does not move with m).
struct TripletGeom {
    double rl, rm, rn;
    double a1[3], a2[3], a3[3];
    bool centre;
    bool first;      // neighbour role: m is the trio's first neighbour (leg l joins the centre and m)
    int i1, i2;      // own-list entries the unit vectors come from: (aa, bb) centre role, (e, -) neighbour role
};

struct TripletRec {
    double v[3][4], d[3][4];
    int first[3];
};

template <bool WANT_F>
__device__ __forceinline__ bool eval_triplet(const KnotRec *recs, const TrioDev *td, const TripletGeom &t, bool valid,
                                             TripletRec &r) {
    if (!valid) return false;
    // leg masks t[0] <= r <= t[-1] (angles.py:502-508).  r == t[0] selects no basis function
    // (searchsorted - 4 < 0) and at r == t[-1] every selected scipy element evaluates to 0
    // (half-open last interval), so both ends contribute nothing: open interval here.
    if (!((t.rl > td->leg[0].t0) && (t.rl < td->leg[0].tlast) && (t.rm > td->leg[1].t0) && (t.rm < td->leg[1].tlast) &&
          (t.rn > td->leg[2].t0) && (t.rn < td->leg[2].tlast))) return false;
    KnotRec kl, km, kn;
    int il = load_interval(recs, td->leg[0], t.rl, kl);
    int im = load_interval(recs, td->leg[1], t.rm, km);
    int in = load_interval(recs, td->leg[2], t.rn, kn);
    bspline4<WANT_F>(kl, t.rl, r.v[0], r.d[0]);
    bspline4<WANT_F>(km, t.rm, r.v[1], r.d[1]);
    bspline4<WANT_F>(kn, t.rn, r.v[2], r.d[2]);
    r.first[0] = il - 3; r.first[1] = im - 3; r.first[2] = in - 3;
    return true;
}

// Output-stationary accumulation: this lane owns NCH columns (one per 64-column chunk) of the current
// trio block and gathers, from every staged record, the raw bins that feed them (1, 2 or 6 symmetry
// images).  Branch-free on purpose: out-of-block sources read a clamped slot and are masked through
// the n-leg pair, so the independent chains of one record (and of two records per trip) interleave.
struct ColSrc { int l, m, n; };   // one raw bin feeding a column; l = 1 << 20 marks "none"

template <bool WANT_E, bool WANT_F, int NSRC, int NCH>
__device__ __forceinline__ void gather_one(const double *rec, const ColSrc (&src)[NCH][NSRC], double (&acc)[NCH][4]) {
    const int4 mt = *(const int4 *)(rec + 34);
    double2 a01 = {0, 0}, a23 = {0, 0}, a45 = {0, 0}, a67 = {0, 0};
    double a8 = 0.0;
    if (WANT_F) {
        a01 = *(const double2 *)(rec + 24); a23 = *(const double2 *)(rec + 26);
        a45 = *(const double2 *)(rec + 28); a67 = *(const double2 *)(rec + 30);
        a8 = rec[32];
    }
    const bool centre = WANT_E && mt.w != 0;         // wave-uniform: only centre-role records feed the energy row
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
#pragma unroll
        for (int k = 0; k < NSRC; k++) {
            const unsigned a = (unsigned)(src[ch][k].l - mt.x), b = (unsigned)(src[ch][k].m - mt.y),
                           c = (unsigned)(src[ch][k].n - mt.z);
            const bool ok = (a | b | c) < 4u;
            const double2 L = *(const double2 *)(rec + 2 * (a & 3u));
            const double2 M = *(const double2 *)(rec + 8 + 2 * (b & 3u));
            // out-of-block sources read the record's zero pair (slot 33 is never written: kept 0,0 at 36-37)
            const double2 N = *(const double2 *)(rec + (ok ? 16 + 2 * (c & 3u) : 36));
            const double z = L.x * M.x;
            if (centre) acc[ch][3] = fma(z, N.x, acc[ch][3]);
            if (WANT_F) {
                const double p1 = L.y * (M.x * N.x), p2 = M.y * (L.x * N.x), p3 = N.y * z;
                // A1 = (a01.x, a01.y, a23.x)  A2 = (a23.y, a45.x, a45.y)  A3 = (a67.x, a67.y, a8)
                acc[ch][0] = fma(p3, a67.x, fma(p2, a23.y, fma(p1, a01.x, acc[ch][0])));
                acc[ch][1] = fma(p3, a67.y, fma(p2, a45.x, fma(p1, a01.y, acc[ch][1])));
                acc[ch][2] = fma(p3, a8, fma(p2, a45.y, fma(p1, a23.x, acc[ch][2])));
            }
        }
    }
}

template <bool WANT_E, bool WANT_F, int NSRC, int NCH>
__device__ __forceinline__ void gather_records(const double *stage, int n_staged, const ColSrc (&src)[NCH][NSRC],
                                               double (&acc)[NCH][4]) {
    double acc2[NCH][4];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) for (int u = 0; u < 4; u++) acc2[ch][u] = 0.0;
    int q = 0;
    if (NSRC * NCH <= 2)                         // narrow variants: two independent records per trip
        for (; q + 1 < n_staged; q += 2) {
            gather_one<WANT_E, WANT_F, NSRC, NCH>(stage + (size_t)q * ITEM_STRIDE, src, acc);
            gather_one<WANT_E, WANT_F, NSRC, NCH>(stage + (size_t)(q + 1) * ITEM_STRIDE, src, acc2);
        }
    else
        for (; q < n_staged; q++) gather_one<WANT_E, WANT_F, NSRC, NCH>(stage + (size_t)q * ITEM_STRIDE, src, acc);
    if (q < n_staged) gather_one<WANT_E, WANT_F, NSRC, NCH>(stage + (size_t)q * ITEM_STRIDE, src, acc);
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) for (int u = 0; u < 4; u++) acc[ch][u] += acc2[ch][u];
}

// 64 evaluated triplets (one per lane) pass through the wave's NSTAGE-record LDS stage in quarters
template <bool WANT_E, bool WANT_F, int NSRC, int NCH>
__device__ __forceinline__ void stage_and_gather(const TripletGeom &t, const TripletRec &r, bool valid, double *stage,
                                                 const ColSrc (&src)[NCH][NSRC], double (&acc)[NCH][4]) {
    const int lane = lane_id();
    for (int part = 0; part < WAVE / NSTAGE; part++) {
        bool mine = valid && ((lane / NSTAGE) == part);
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
            *(int4 *)(rec + 34) = make_int4(r.first[0], r.first[1], r.first[2], t.centre ? 1 : 0);
            rec[36] = 0.0; rec[37] = 0.0;
        }
        wave_sync();
        gather_records<WANT_E, WANT_F, NSRC, NCH>(stage, __popcll(mask), src, acc);
        wave_sync();
    }
}

// per-wave scratch in LDS
struct WaveLds {
    double *ox, *oy, *oz, *orr, *oir; // own 3-body neighbour list [cap]: vector, length, 1 / length
    int *oparent, *oshift, *osidx;
    int *noff, *nbase;                 // neighbour-role prefix [cap+1] / start index [cap]
    int *so;                           // species offsets in the own list [S+1]
    int *ospoff;                       // species offsets of every own neighbour's list [cap][S+1]
    int sp_stride;                     // S + 1
    double *geo;                       // MFMA specialisation: geometry of the walked triplets, 7 fields of GEO_N
    double *stage;                     // NSTAGE triplet / pair records
    double *cand;                      // 2-body candidates [cand_cap][CAND_STRIDE] (aliases stage)
    double *pstage;                    // 2-body row buffer [4][n_pair_cols] (behind the candidates, inside stage)
};

// Which triplets does atom m contribute to trio block td?  Items [0, cnt_c) are the triplets m centres (own
// neighbours of species sa x sb); items [cnt_c, n_items) the triplets in which m is a neighbour of a centre of
// species sc (one of m's own neighbours e), enumerated through that centre's list.
struct TrioWalk {
    int cnt_c, ra_lo, rb_lo, nb_;
    float rcp_nb;          // 1 / nb_ (item index -> (p / nb_, p % nb_) without an integer division)
    int total_n, rc_lo, ncen, sx;
    int n_items;
    // The reference tiles the positions AS GIVEN with images -fac .. fac per axis (geometry.py:108-149) and takes the third
    // atom of a ghost-centred triplet from that supercell (angles.py:424-514): an image beyond the range does not exist
    // there and the triplet's force terms are absent from its rows.  With every atom inside its cell this never happens
    // (two 3-body legs reach no further than r_cut); a batch with unwrapped atoms (k_frame_bins marks it) checks the
    // third atom's absolute image index.
    int img_check;
};

template <bool WANT_F, bool IMG>
__device__ __forceinline__ void trio_walk_setup(const FeatArgs &A, const WaveLds &w, int sc, int sa, int sb, int sm, TrioWalk &k) {
    const int lane = lane_id();
    // (compiled into the IMG launches only: the ordinary ones leave at once when the batch has such atoms and the host repeats
    // the call with these)
    k.img_check = (IMG && WANT_F && A.outside) ? __builtin_amdgcn_readfirstlane(A.outside[0]) : 0;   // 0 off, 1 range rule, 2 + extension lists
    if (k.img_check && A.n3.xcap > 0) k.img_check = 2;
    k.cnt_c = 0; k.ra_lo = 0; k.rb_lo = 0; k.nb_ = 1;
    if (sm == sc && !UF3_SKIP(2)) {          // m is the centre: own neighbours of species sa x sb
        k.ra_lo = w.so[sa]; k.rb_lo = w.so[sb];
        int na = w.so[sa + 1] - k.ra_lo;
        k.nb_ = w.so[sb + 1] - k.rb_lo;
        k.cnt_c = (sa == sb) ? na * (na - 1) / 2 : na * k.nb_;
        if (k.nb_ < 1) k.nb_ = 1;
    }
    k.total_n = 0; k.rc_lo = 0; k.ncen = 0; k.sx = -1;
    if (WANT_F && !UF3_SKIP(4)) {            // m is a neighbour of a centre of species sc
        if (sm == sa) k.sx = sb; else if (sm == sb) k.sx = sa;
        if (k.sx >= 0) {
            k.rc_lo = w.so[sc];
            k.ncen = w.so[sc + 1] - k.rc_lo;
            for (int e0 = 0; e0 < k.ncen; e0 += WAVE) {           // exclusive scan of |N3_sx(centre e)|
                int e = e0 + lane;
                int cnt = 0, base = 0;
                if (e < k.ncen) {
                    const int *sp = w.ospoff + (size_t)(k.rc_lo + e) * w.sp_stride;
                    base = sp[k.sx]; cnt = sp[k.sx + 1] - base;
                    if (IMG && k.img_check > 1) {     // + the centre's extension entries: base | own count << 8 | first extension << 16
                        const int *xo = A.n3.xoff + (size_t)w.oparent[k.rc_lo + e] * (UF3_MAX_SPECIES + 1);
                        const int xb = xo[k.sx], xn = xo[k.sx + 1] - xb;
                        base |= (cnt << 8) | (xb << 16);
                        cnt += xn;
                    }
                }
                const int incl = wave_scan_incl(cnt);
                if (e < k.ncen) { w.noff[e] = k.total_n + incl - cnt; w.nbase[e] = base; }
                k.total_n += __builtin_amdgcn_readlane(incl, WAVE - 1);
            }
            if (lane == 0) w.noff[k.ncen] = k.total_n;
            wave_sync();
        }
    }
    // everything above is the same in all lanes: keep it in scalar registers
    k.cnt_c = __builtin_amdgcn_readfirstlane(k.cnt_c); k.ra_lo = __builtin_amdgcn_readfirstlane(k.ra_lo);
    k.rb_lo = __builtin_amdgcn_readfirstlane(k.rb_lo); k.nb_ = __builtin_amdgcn_readfirstlane(k.nb_);
    k.total_n = __builtin_amdgcn_readfirstlane(k.total_n); k.rc_lo = __builtin_amdgcn_readfirstlane(k.rc_lo);
    k.ncen = __builtin_amdgcn_readfirstlane(k.ncen); k.sx = __builtin_amdgcn_readfirstlane(k.sx);
    k.n_items = k.cnt_c + k.total_n;
    k.rcp_nb = __builtin_amdgcn_rcpf((float)k.nb_);
}

// geometry of item p of the walk; false when the item is void (p out of range, or the third atom is m itself)
template <bool WANT_F, bool IMG>
__device__ __forceinline__ bool trio_walk_geom(const FeatArgs &A, const FrameGeom &g, const WaveLds &w, int sa, int sb,
                                               const TrioWalk &k, int m, int sm, int p, TripletGeom &tg) {
    const int cap = A.n3.cap;
    const int m_local = m - g.atom_lo;
    bool valid = p < k.n_items;
    tg.centre = p < k.cnt_c;
    tg.first = false;
    tg.i1 = tg.i2 = 0;
    if (valid && tg.centre) {
        int aa, bb;
        if (sa == sb) {
            // pair index -> (aa < bb): bb = floor((1 + sqrt(1 + 8 p)) / 2).  In single precision the root of a non-square
            // 1 + 8 p < 2^15 stays >= 0.004 away from the next integer, far more than the hardware root's error; one
            // branch-free correction each way is kept as a guard.
            bb = (int)((1.0f + __builtin_amdgcn_sqrtf(fmaf(8.0f, (float)p, 1.0f))) * 0.5f);
            bb -= (bb * (bb - 1) / 2 > p) ? 1 : 0;
            bb += ((bb + 1) * bb / 2 <= p) ? 1 : 0;
            aa = p - bb * (bb - 1) / 2;
            aa += k.ra_lo; bb += k.ra_lo;
        } else {
            // p / nb_ by a reciprocal (p < 4096, nb_ <= 64: (p + 0.5) / nb_ is at least 0.5 / nb_ away from an integer)
            const int q = (int)(((float)p + 0.5f) * k.rcp_nb);
            aa = k.ra_lo + q; bb = k.rb_lo + (p - q * k.nb_);
        }
        tg.rl = w.orr[aa]; tg.rm = w.orr[bb];
        tg.i1 = aa; tg.i2 = bb;
        double ex = w.ox[bb] - w.ox[aa], ey = w.oy[bb] - w.oy[aa], ez = w.oz[bb] - w.oz[aa];
        tg.rn = norm3_leg(ex, ey, ez);
        if (WANT_F) {
            double il = w.oir[aa], im = w.oir[bb];
            tg.a1[0] = w.ox[aa] * il; tg.a1[1] = w.oy[aa] * il; tg.a1[2] = w.oz[aa] * il;
            tg.a2[0] = w.ox[bb] * im; tg.a2[1] = w.oy[bb] * im; tg.a2[2] = w.oz[bb] * im;
            tg.a3[0] = tg.a3[1] = tg.a3[2] = 0.0;
        }
    } else if (valid) {
        int q = p - k.cnt_c;
        int lo = 0, hi = k.ncen - 1;                   // centre e with noff[e] <= q < noff[e+1]
        while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (w.noff[mid] <= q) lo = mid; else hi = mid - 1; }
        int e = k.rc_lo + lo, kk = w.nbase[lo] + (q - w.noff[lo]);
        int pc = w.oparent[e];
        const N3Entry *kp = A.n3.ent + (size_t)pc * cap + kk;
        if (IMG && k.img_check > 1) {                   // (packed by trio_walk_setup)
            const int nb = w.nbase[lo], rel = q - w.noff[lo], own = (nb >> 8) & 0xff;
            kp = rel < own ? A.n3.ent + (size_t)pc * cap + (nb & 0xff) + rel
                           : A.n3.xent + (size_t)pc * A.n3.xcap + (nb >> 16) + (rel - own);
        }
        int s0, s1, s2;
        unpack3(w.oshift[e], s0, s1, s2);
        // the centre's own-list fields go out together with the entry's global load (requested after it they would each wait
        // behind it: this chain is the walk's critical path)
        const double oex = w.ox[e], oey = w.oy[e], oez = w.oz[e], oer = w.orr[e], oie = w.oir[e];
        const N3Entry ke = *kp;
        asm volatile("" ::: "memory");
        int kparent = ke.parent, kshift = ke.shiftc;
        valid = !(kparent == m && kshift == pack3(-s0, -s1, -s2));       // k is m itself
        if (IMG && k.img_check) {                                        // k's image as the reference numbers it
            int t0, t1, t2;
            unpack3(kshift, t0, t1, t2);
            valid = valid && abs(s0 + t0) <= g.fac[0] && abs(s1 + t1) <= g.fac[1] && abs(s2 + t2) <= g.fac[2];
        }
        if (valid) {
            int ksidx = ke.sidx;
            int msidx = supercell_index(g, -s0, -s1, -s2, m_local);      // m as numbered from c
            double vx = ke.dx, vy = ke.dy, vz = ke.dz, rk = ke.r;
            double ex = oex + vx, ey = oey + vy, ez = oez + vz;              // m -> k
            tg.rn = norm3_leg(ex, ey, ez);
            bool m_first = neighbour_is_first(g, sm, k.sx, s0, s1, s2, m_local, msidx, ksidx, kshift,
                                              kparent - g.atom_lo);
            tg.i1 = e; tg.i2 = e;
            // 1 / rn: hardware reciprocal + two Newton steps (the unit vector needs no correctly rounded quotient)
            double in = __builtin_amdgcn_rcp(tg.rn);
            in = fma(fma(-tg.rn, in, 1.0), in, in);
            in = fma(fma(-tg.rn, in, 1.0), in, in);
            const double ie = oie;
            double ue[3] = {oex * ie, oey * ie, oez * ie};
            tg.a3[0] = ex * in; tg.a3[1] = ey * in; tg.a3[2] = ez * in;
            tg.first = m_first;
            if (m_first) {
                tg.rl = oer; tg.rm = rk;
                for (int u = 0; u < 3; u++) { tg.a1[u] = ue[u]; tg.a2[u] = 0.0; }
            } else {
                tg.rl = rk; tg.rm = oer;
                for (int u = 0; u < 3; u++) { tg.a1[u] = 0.0; tg.a2[u] = ue[u]; }
            }
        }
    }
    return valid;
}

template <bool WANT_E, bool WANT_F, int NSRC, int NCH, bool IMG>
__device__ __forceinline__ void trio_block(const FeatArgs &A, const BasisDev *B, const KnotRec *recs, const FrameGeom &g,
                                           const WaveLds &w, int m, int sm, int t, const ESink &es) {
    const int lane = lane_id();
    const TrioDev td_copy = load_const(A.trios + t);          // scalar loads (see load_const)
    const TrioDev *td = &td_copy;
    TrioWalk k;
    trio_walk_setup<WANT_F, IMG>(A, w, td->sc, td->sa, td->sb, sm, k);
    const int ncol = td->ncol;
    for (int c0 = 0; c0 < ncol; c0 += NCH * WAVE) {
        ColSrc src[NCH][NSRC];
        double acc[NCH][4];
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) for (int u = 0; u < 4; u++) acc[ch][u] = 0.0;
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            int col = c0 + ch * WAVE + lane;
#pragma unroll
            for (int q = 0; q < NSRC; q++) {
                int sp = col < ncol ? A.colsrc[td->src_off + col * NSRC + q] : -1;
                src[ch][q].l = sp < 0 ? (1 << 20) : (sp & 255);
                src[ch][q].m = (sp >> 8) & 255;
                src[ch][q].n = (sp >> 16) & 255;
            }
        }
        for (int p0 = 0; p0 < k.n_items; p0 += WAVE) {
            TripletGeom tg;
            TripletRec r;
            bool valid = trio_walk_geom<WANT_F, IMG>(A, g, w, td->sa, td->sb, k, m, sm, p0 + lane, tg);
            valid = eval_triplet<WANT_F>(recs, td, tg, valid, r);
            stage_and_gather<WANT_E, WANT_F, NSRC, NCH>(tg, r, valid, w.stage, src, acc);
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            int col = c0 + ch * WAVE + lane;
            if (col < ncol) {
                if (WANT_F) {
                    double *dst = A.x_f + (size_t)m * 3 * A.ld + td->col + col;
                    __builtin_nontemporal_store(acc[ch][0], dst); __builtin_nontemporal_store(acc[ch][1], dst + A.ld);
                    __builtin_nontemporal_store(acc[ch][2], dst + 2 * (size_t)A.ld);
                }
                if (WANT_E) es.add(td->col + col, acc[ch][3]);
            }
        }
    }
}

// ---- dense window on the fp64 matrix cores (MODE 6 - 9) ----------------------------------------------------------
// For one atom and trio block the rows are a rank-(2 x records) sum of outer products over the window of raw bins that
// feed the block's columns (ext_l x ext_m x ext_n bins starting at lo):
//     X_c[l][m][n] = sum over records, two K slots s each, of   A_s[(c, l)] * B_s[(m, n)]
//   class 0  m centres the triplet       s0:  u1_c B'_l | B_m  B_n      s1:  u2_c B_l | B'_m B_n    energy row: B_l | B_m B_n in s0
//   class 1  m neighbour, on leg l       s0:  u_c  B'_l | B_m  B_n      s1:  a3_c B_l | B_m  B'_n
//   class 2  m neighbour, on leg m       s0:  u_c  B_l  | B'_m B_n      s1:  a3_c B_l | B_m  B'_n
// (u: unit vectors of the legs that end at m, a3: unit vector m -> k; these are -(d r_leg / d R_m)).  On v_mfma_f64_16x16x4:
// rows (c, l) with c = x, y, z, energy (4 ext_l rows, RT tiles of 16), columns (m, n) (CT tiles), K = 4 slots = 2
// records per step.  Every operand is ONE product of two 8-byte LDS reads, the same for all classes, because the staging
// pass arranges each record by its class:
//   L pairs (LX0[l], LX1[l]) | M pairs (MB0[m], MB1[m]) | N pairs (NB0[n], NB1[n]) | D pairs (D0[c], D1[c]) c = x,y,z,e | 0 0
//   lane (row (c, l), slot s):  A = L[2l + s] * D[2c + s]   (energy rows: L[2l + 1] * D[6 + s], D[6] = 1 for class 0, else 0)
//   lane (col (m, n), slot s):  B = M[2m + s] * N[2n + s]
// Records are walked 63 at a time (geometry to LDS, sorted by class), staged <= 21 at a time by lanes (record, leg) that
// evaluate one leg each and scatter its four (value, derivative) entries to their place inside the window.  The
// accumulators are the raw window; the symmetry fold into columns happens once per (atom, block) through dsrc.
// Energy-only launches stage two records per staged record (K slot = record parity), rows = l only.
typedef double double4_t __attribute__((ext_vector_type(4)));

struct DenseLayout {
    int oM, oN, oD, oZ, stride, cw;      // in doubles (L pairs at 0); cw = width of a dumped row
};
__host__ __device__ __forceinline__ DenseLayout dense_layout(int ext_l, int ext_m, int ext_n) {
    DenseLayout d;
    d.oM = 2 * ext_l; d.oN = d.oM + 2 * ext_m; d.oD = d.oN + 2 * ext_n; d.oZ = d.oD + 8;
    d.stride = (d.oZ + 2 + 7) & ~7;
    d.cw = 16 * ((ext_m * ext_n + 15) / 16);
    return d;
}
// (row tiles, column tiles) of the four specialisations
__host__ __device__ constexpr int dense_rt(int mode) { return mode == 9 ? 2 : 1; }
__host__ __device__ constexpr int dense_ct(int mode) { return mode == 6 ? 1 : (mode == 7 ? 2 : (mode == 8 ? 4 : 6)); }
// specialisation that serves a window, 0 if none does
__host__ __device__ __forceinline__ int dense_mode_for(int ext_l, int ext_m, int ext_n) {
    const int rt = (4 * ext_l + 15) / 16, ct = (ext_m * ext_n + 15) / 16;
    if (rt == 1) return ct == 1 ? 6 : (ct == 2 ? 7 : (ct <= 4 ? 8 : (ct <= 6 ? 9 : 0)));
    return (rt == 2 && ct <= 6) ? 9 : 0;
}
#define DENSE_NREC 21     // records staged per pass (3 lanes each), upper bound; the launch may use fewer
#define GEO_N 64          // field stride of the walked triplets' geometry: rl | rm | rn | a3x | a3y | a3z | packed ints

template <int RT, int CT>
struct DenseLane {
    int aL[RT], aD[RT];      // row (c, l) of row tile rt, K slot of this lane: L entry, D entry (doubles inside a record)
    int bM[CT], bN[CT];      // column (m, n) of column tile ct: M entry, N entry
};

// one step = two staged records (four K slots); TMASK: the column tiles the records of this step can touch
template <int TMASK, int RT, int CT>
__device__ __forceinline__ void dense_step(const double *rec, const DenseLane<RT, CT> &o, double4_t (&acc)[RT][CT]) {
    // all operand reads go out before anything waits on them
    double la[RT], da[RT], mb[CT], nb[CT];
    constexpr int N_READS = 2 * (RT + __builtin_popcount(TMASK));
#pragma unroll
    for (int rt = 0; rt < RT; rt++) { la[rt] = rec[o.aL[rt]]; da[rt] = rec[o.aD[rt]]; }
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
        if ((TMASK >> ct) & 1) { mb[ct] = rec[o.bM[ct]]; nb[ct] = rec[o.bN[ct]]; }
    __builtin_amdgcn_sched_group_barrier(0x100, N_READS, 0);                              // DS reads
    double a[RT], b[CT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) a[rt] = la[rt] * da[rt];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) if ((TMASK >> ct) & 1) b[ct] = mb[ct] * nb[ct];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
            if ((TMASK >> ct) & 1) acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt], b[ct], acc[rt][ct], 0, 0, 0);
}

// staged records [r0, r1) (both even; two K slots each) into the accumulators, two records per step.  STRIDE != 0: the
// record stride is a compile-time constant, so the unrolled steps address their operands with immediate offsets
// (one address update per operand and trip instead of one per operand and step).
template <int TMASK, int RT, int CT, int STRIDE>
__device__ __forceinline__ void dense_accumulate(const double *stage, int rt_stride, int r0, int r1, const DenseLane<RT, CT> &o,
                                                 double4_t (&acc)[RT][CT]) {
    const int stride = STRIDE ? STRIDE : rt_stride;
    const int half = lane_id() >> 5;                          // which record of the step this lane's K slot belongs to
    const double *rec = stage + (size_t)(r0 + half) * stride;
    int q = r0;
    if (STRIDE && RT * CT <= 4) {      // (a step of a wide window is 8-12 MFMAs: nothing to gain, registers to lose)
        for (; q + 8 <= r1; q += 8, rec += 8 * stride) {
#pragma unroll
            for (int u = 0; u < 4; u++) dense_step<TMASK, RT, CT>(rec + 2 * u * stride, o, acc);
        }
        if (q + 4 <= r1) {
#pragma unroll
            for (int u = 0; u < 2; u++) dense_step<TMASK, RT, CT>(rec + 2 * u * stride, o, acc);
            q += 4; rec += 4 * stride;
        }
    }
    for (; q < r1; q += 2, rec += 2 * stride) dense_step<TMASK, RT, CT>(rec, o, acc);
}

// accumulator window -> LDS, then lanes <-> columns fold the symmetry images and write the rows (the end of a block of the
// matrix-core specialisations that keep the raw window in registers).  As many components (x, y, z, energy) per dump as the
// stage holds; the fold table's entries of a lane's columns are fetched once, before the first dump (they were re-read from
// HBM for every component: a third of the wide windows' vector instructions went into this function).
template <bool WANT_E, bool WANT_F, int RT, int CT>
__device__ __forceinline__ void dense_fold(const FeatArgs &A, const WaveLds &w, const TrioDev *td, const DenseLayout &dl, int m, int F,
                                           const ESink &es, const int (&fragp)[4], const int *dsrc, double4_t (&acc)[RT][CT]) {
    constexpr int MAXIT = RT * CT <= 2 ? 2 : 8;                  // 64-column rounds whose table entries stay in registers
    const int lane = lane_id();
    const int ncol = td->ncol, ext_l = td->ext[0], nsrc = td->nsrc, cw = dl.cw;
    double *dump = w.stage;
    const int c_first = WANT_F ? 0 : 3, c_last = WANT_E ? 3 : 2;
    const int rows_c = ext_l * cw;                               // doubles of one component's rows
    const int cpp = max(1, min(c_last - c_first + 1, A.dense_stage / rows_c));      // components per dump
    const int inv_l = (65536 + ext_l - 1) / ext_l;
    const bool in_regs = nsrc <= 2 && ncol <= MAXIT * WAVE;      // wave-uniform
    int o0[MAXIT], o1[MAXIT];
    if (in_regs) {
        // (the table through the LDS or the global address space explicitly: a generic pointer makes flat loads)
        auto fetch = [&](auto as_tag) {
            typedef decltype(as_tag) Ints;
            Ints tab = (Ints)(dsrc + td->src_off);
#pragma unroll
            for (int it = 0; it < MAXIT; it++) {
                const int col = min(lane + it * WAVE, ncol - 1);
                o0[it] = tab[col * nsrc];
                o1[it] = tab[col * nsrc + (nsrc > 1 ? 1 : 0)];
            }
        };
        if (A.dsrc_lds) fetch((const __attribute__((address_space(3))) int *)nullptr);
        else fetch((const __attribute__((address_space(1))) int *)nullptr);
#pragma unroll
        for (int it = 0; it < MAXIT; it++) {
            const int col = lane + it * WAVE;
            if (col >= ncol) o0[it] = -1;
            if (col >= ncol || nsrc < 2) o1[it] = -1;
        }
    }
    for (int c0 = c_first; c0 <= c_last; c0 += cpp) {
        const int nc = min(cpp, c_last - c0 + 1);
        wave_sync();
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int fr = fragp[v] >> 4, fc = fragp[v] & 15;
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const int row = rt * 16 + fr;
                // rows are (component, l) with force rows wanted, l alone (the energy) without
                const int c = WANT_F ? (row * inv_l) >> 16 : 3, pl = WANT_F ? row - c * ext_l : row;
                const bool ok = c >= c0 && c < c0 + nc && pl < ext_l;
#pragma unroll
                for (int ct = 0; ct < CT; ct++)
                    if (ok && ct * 16 < cw) dump[((c - c0) * ext_l + pl) * cw + ct * 16 + fc] = acc[rt][ct][v];
            }
        }
        wave_sync();
        if (in_regs) {
#pragma unroll
            for (int it = 0; it < MAXIT; it++) {
                if (it * WAVE >= ncol) break;                    // (wave-uniform)
                const int col = lane + it * WAVE;
                if (col >= ncol) continue;
                // both sources of a column are read whether they exist or not and weighted 0 / 1: no branch around a read
                const double w0 = o0[it] >= 0 ? 1.0 : 0.0, w1 = o1[it] >= 0 ? 1.0 : 0.0;
                const double *p0 = dump + max(o0[it], 0), *p1 = dump + max(o1[it], 0);
                double t0[4], t1[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { const int qq = q < nc ? q : 0; t0[q] = p0[qq * rows_c]; t1[q] = p1[qq * rows_c]; }
                asm volatile("" ::: "memory");                   // (all reads requested before the first sum)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (q >= nc) break;
                    const double val = w0 * t0[q] + w1 * t1[q];
                    const int comp = c0 + q;
                    if (comp < 3) { if (!UF3_SKIP(32)) __builtin_nontemporal_store(val, A.x_f + (size_t)m * 3 * A.ld + (size_t)comp * A.ld + td->col + col); }
                    else es.add(td->col + col, val);
                }
            }
        } else {
            for (int col = lane; col < ncol; col += WAVE)
                for (int q = 0; q < nc; q++) {
                    double val = 0.0;
                    for (int sidx = 0; sidx < nsrc; sidx++) {
                        const int off = dsrc[td->src_off + col * nsrc + sidx];
                        if (off >= 0) val += dump[q * rows_c + off];
                    }
                    const int comp = c0 + q;
                    if (comp < 3) { if (!UF3_SKIP(32)) __builtin_nontemporal_store(val, A.x_f + (size_t)m * 3 * A.ld + (size_t)comp * A.ld + td->col + col); }
                    else es.add(td->col + col, val);
                }
        }
    }
}

template <bool WANT_E, bool WANT_F, int MODE, bool IMG, bool RL>
__device__ __forceinline__ void trio_block_mfma(const FeatArgs &A, const BasisDev *B, const KnotRec *recs, const FrameGeom &g,
                                                const WaveLds &w, int m, int sm, int t, const ESink &es,
                                                const int (&fragp)[4], const int *dsrc) {
    constexpr int RT = dense_rt(MODE), CT = dense_ct(MODE);
    const int lane = lane_id();
    // the descriptor through the constant address space: scalar loads into SGPRs (the tables never change while a
    // kernel runs; through a plain global pointer every field read is a vector load the compiler cannot hoist)
    const TrioDev td_copy = load_const(A.trios + t);
    const TrioDev *td = &td_copy;
    PhaseClock pc;
    TrioWalk k;
    trio_walk_setup<WANT_F, IMG>(A, w, td->sc, td->sa, td->sb, sm, k);
    const int F = load_const(&B->F);     // (a scalar load: as B->F it is a vector load and a wait on everything in flight)
    const int lo_l = td->lo[0], lo_m = td->lo[1], lo_n = td->lo[2];
    const int ext_l = td->ext[0], ext_m = td->ext[1], ext_n = td->ext[2];
    const DenseLayout dl = dense_layout(ext_l, ext_m, ext_n);
    const int r16 = lane & 15, slot = (lane >> 4) & 1;
    // per-lane operand offsets (small quotients by multiply-shift, exact for dividends < 4096 and divisors <= 64)
    const int inv_l = (65536 + ext_l - 1) / ext_l, inv_m = (65536 + ext_m - 1) / ext_m;      // wave-uniform
    DenseLane<RT, CT> o;
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        const int row = rt * 16 + r16;
        if (WANT_F) {
            const int c = (row * inv_l) >> 16, pl = row - c * ext_l;
            const bool ok = c < 3 || (c == 3 && WANT_E);
            o.aL[rt] = ok ? 2 * pl + (c == 3 ? 1 : slot) : dl.oZ;
            o.aD[rt] = ok ? dl.oD + 2 * c + slot : dl.oZ;
        } else {
            const bool ok = row < ext_l;
            o.aL[rt] = ok ? 2 * row + slot : dl.oZ;
            o.aD[rt] = ok ? dl.oD + 6 + slot : dl.oZ;
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const int col = ct * 16 + r16;                       // columns n-major: col = n * ext_m + m
        const int pn = (col * inv_m) >> 16, pm = col - pn * ext_m;
        const bool ok = pn < ext_n;
        o.bM[ct] = ok ? dl.oM + 2 * pm + slot : dl.oZ;
        o.bN[ct] = ok ? dl.oN + 2 * pn + slot : dl.oZ;
    }
    // staging role of this lane: leg `leg` of record `li` of the pass (lanes 3*li .. 3*li+2)
    const int li = (lane * 21846) >> 16, leg = lane - 3 * li;
    LegDev lg;
    lg.rec_off = leg == 0 ? td->leg[0].rec_off : (leg == 1 ? td->leg[1].rec_off : td->leg[2].rec_off);
    lg.nk = leg == 0 ? td->leg[0].nk : (leg == 1 ? td->leg[1].nk : td->leg[2].nk);
    lg.t0 = leg == 0 ? td->leg[0].t0 : (leg == 1 ? td->leg[1].t0 : td->leg[2].t0);
    lg.tlast = leg == 0 ? td->leg[0].tlast : (leg == 1 ? td->leg[1].tlast : td->leg[2].tlast);
    lg.inv_h = leg == 0 ? td->leg[0].inv_h : (leg == 1 ? td->leg[1].inv_h : td->leg[2].inv_h);
    const int w_off = leg == 0 ? 0 : (leg == 1 ? dl.oM : dl.oN);
    const int w_ext = leg == 0 ? ext_l : (leg == 1 ? ext_m : ext_n), w_lo = leg == 0 ? lo_l : (leg == 1 ? lo_m : lo_n);
    // Two column tiles (the reference's default trims): a record touches 4 consecutive n bins, i.e. only tile 0, only
    // tile 1 or both (columns are n-major), decided by the knot interval of r_n: r_n <= thr0 / r_n > thr2 / else.  The
    // walk sorts its triplets by that class, and the steps of a pass that hold only records of class 0 (2) skip the MFMA
    // of tile 1 (0).
    constexpr bool SPLIT = WANT_F && CT == 2;
    const double thr0 = td->thr0, thr2 = td->thr2;
    double4_t acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[rt][ct] = double4_t{0, 0, 0, 0};
    pc.lap(1);
    const int nrec = A.dense_nrec, batch = 3 * nrec;
    const double lo_r[3] = {td->leg[0].t0, td->leg[1].t0, td->leg[2].t0};
    const double hi_r[3] = {td->leg[0].tlast, td->leg[1].tlast, td->leg[2].tlast};
    for (int p0 = 0; p0 < k.n_items; p0 += batch) {
        // ---- walk: one triplet per lane, geometry to LDS (sorted by tile class) ---------------------------------
        int n_valid, n_cls0 = 0, n_cls01 = 0;
        {
            TripletGeom tg;
            bool valid = lane < batch && p0 + lane < k.n_items;
            if (valid) valid = trio_walk_geom<WANT_F, IMG>(A, g, w, td->sa, td->sb, k, m, sm, p0 + lane, tg);
            // leg masks t[0] <= r <= t[-1] (angles.py:502-508); both ends contribute nothing (see eval_triplet)
            if (valid)                   // (bounds hoisted; `&`, not `&&`: no branch, no memory access per clause)
                valid = (tg.rl > lo_r[0]) & (tg.rl < hi_r[0]) & (tg.rm > lo_r[1]) & (tg.rm < hi_r[1]) &
                        (tg.rn > lo_r[2]) & (tg.rn < hi_r[2]);
            int rank;
            if (SPLIT) {
                const bool is0 = valid && tg.rn <= thr0, is2 = valid && tg.rn > thr2, is1 = valid && !is0 && !is2;
                const unsigned long long m0 = __ballot(is0), m1 = __ballot(is1), m2 = __ballot(is2);
                n_cls0 = __popcll(m0); n_cls01 = n_cls0 + __popcll(m1);
                n_valid = n_cls01 + __popcll(m2);
                rank = is0 ? mbcnt(m0) : (is1 ? n_cls0 + mbcnt(m1) : n_cls01 + mbcnt(m2));
            } else {
                const unsigned long long mv = __ballot(valid);
                n_valid = __popcll(mv);
                rank = mbcnt(mv);
            }
            if (valid) {
                double *gq = w.geo + rank;
                gq[0] = tg.rl; gq[GEO_N] = tg.rm; gq[2 * GEO_N] = tg.rn;
                if (WANT_F) {
                    gq[3 * GEO_N] = tg.a3[0]; gq[4 * GEO_N] = tg.a3[1]; gq[5 * GEO_N] = tg.a3[2];
                    ((int2 *)(w.geo + 6 * GEO_N))[rank] = make_int2(tg.i1 | (tg.i2 << 16), tg.centre ? 0 : (tg.first ? 1 : 2));
                }
            }
        }
        wave_sync();
        pc.lap(2);
        // ---- staging passes: lane (record li, leg) evaluates one leg and scatters it into the window ---------
        for (int base = 0; base < n_valid; base += nrec) {
            const int n_part = min(nrec, n_valid - base);
            // staged records, padded with an all-zero record to an even count (a step is two records)
            const int n_real = WANT_F ? n_part : (n_part + 1) >> 1, n_staged = n_real + (n_real & 1);
            const bool mine = li < n_part && !UF3_SKIP(16);
            // the records of this pass (and the padding record) start from zero: one contiguous fill by the whole wave.
            // LDS writes of a wave stay in program order, so the fill lands before any lane's scatter below
            if (!UF3_SKIP(16))
                for (int q = 2 * lane; q < n_staged * dl.stride; q += 2 * WAVE) *(double2 *)(w.stage + q) = double2{0.0, 0.0};
            if (mine) {
                const int gi = base + li;
                // (the LDS reads of a pass in two waves, as in trio_block_grouped)
                const double x = w.geo[leg * GEO_N + gi];
                const int2 pk = WANT_F ? ((const int2 *)(w.geo + 6 * GEO_N))[gi] : make_int2(0, 0);
                const double a3 = WANT_F ? w.geo[(3 + leg) * GEO_N + gi] : 0.0;
                __builtin_amdgcn_sched_group_barrier(0x100, WANT_F ? 3 : 1, 0);
                const int i1 = pk.x & 0xffff, i2 = pk.x >> 16;
                const double *oc = w.ox + (size_t)leg * A.n3.cap;               // ox | oy | oz are consecutive [cap] arrays
                const double oc1 = WANT_F ? oc[i1] : 0.0, oi1 = WANT_F ? w.oir[i1] : 0.0, oc2 = WANT_F ? oc[i2] : 0.0,
                             oi2 = WANT_F ? w.oir[i2] : 0.0;
                KnotRec kr;
                double v[4], d[4];
                const int first = load_interval<RL ? 3 : 1>(recs, lg, x, kr) - 3;
                bspline4<WANT_F>(kr, x, v, d);
                if (WANT_F) {
                    double *rec = w.stage + (size_t)li * dl.stride;
                    const int cls = pk.y;
                    // which of (value, derivative) goes to K slot 0 / 1 (table in the header of this section)
                    const bool d0 = leg == 0 ? cls != 2 : (leg == 1 ? cls == 2 : false);
                    const bool d1 = leg == 0 ? false : (leg == 1 ? cls == 0 : cls != 0);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const unsigned ws = (unsigned)(first + q - w_lo);
                        if (ws < (unsigned)w_ext)
                            *(double2 *)(rec + w_off + 2 * ws) = double2{d0 ? d[q] : v[q], d1 ? d[q] : v[q]};
                    }
                    // component `leg` of the two direction vectors: unit vectors of own-list entries, and m -> k
                    const double u1 = oc1 * oi1, u2 = oc2 * oi2;
                    *(double2 *)(rec + dl.oD + 2 * leg) = double2{u1, cls == 0 ? u2 : a3};
                    if (leg == 0 && cls == 0) rec[dl.oD + 6] = 1.0;
                } else {
                    double *rec = w.stage + (size_t)(li >> 1) * dl.stride;
                    const int sl2 = li & 1;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const unsigned ws = (unsigned)(first + q - w_lo);
                        if (ws < (unsigned)w_ext) rec[w_off + 2 * ws + sl2] = v[q];
                    }
                    if (leg == 0) rec[dl.oD + 6 + sl2] = 1.0;
                }
            }
            // steps (two records) of class 0 only: [0, e_a); of class 2 only: [s_c, n_staged) (the padding record is neutral)
            const int e_a = SPLIT ? max(0, min(n_part, n_cls0 - base)) & ~1 : 0;
            const int s_c = SPLIT ? (max(0, min(n_part, n_cls01 - base)) + 1) & ~1 : n_staged;
            wave_sync();
            pc.lap(4);
            if (!UF3_SKIP(8)) {
                constexpr int S0 = MODE == 6 ? 32 : (MODE == 7 ? 40 : (MODE == 8 ? 48 : 56));    // the usual stride of the mode
                constexpr int ALL = (1 << CT) - 1;
                if (dl.stride == S0) {       // wave-uniform
                    if (SPLIT) {
                        dense_accumulate<1, RT, CT, S0>(w.stage, S0, 0, e_a, o, acc);
                        dense_accumulate<ALL, RT, CT, S0>(w.stage, S0, e_a, s_c, o, acc);
                        dense_accumulate<(CT == 2 ? 2 : ALL), RT, CT, S0>(w.stage, S0, s_c, n_staged, o, acc);
                    } else dense_accumulate<ALL, RT, CT, S0>(w.stage, S0, 0, n_staged, o, acc);
                } else if (SPLIT) {
                    dense_accumulate<1, RT, CT, 0>(w.stage, dl.stride, 0, e_a, o, acc);
                    dense_accumulate<ALL, RT, CT, 0>(w.stage, dl.stride, e_a, s_c, o, acc);
                    dense_accumulate<(CT == 2 ? 2 : ALL), RT, CT, 0>(w.stage, dl.stride, s_c, n_staged, o, acc);
                } else dense_accumulate<ALL, RT, CT, 0>(w.stage, dl.stride, 0, n_staged, o, acc);
            }
            wave_sync();
            pc.lap(5);
        }
    }
    dense_fold<WANT_E, WANT_F, RT, CT>(A, w, td, dl, m, F, es, fragp, dsrc, acc);
    wave_sync();
    pc.lap(6);
    // the dump left window values in the stage: stale slots must stay finite (they are multiplied by 0), which they are
}

// ---- banded wide windows (MODE 9 force launches): the untrimmed / higher-resolution blocks -----------------------------
// A window of 6 x 6 x 12 bins is 2 row tiles x 5 column tiles; the dense loop above spends 10 MFMAs on every step although a
// record touches four consecutive n bins, i.e. at most three neighbouring column tiles (columns are n-major).  The leg-n
// intervals are cut into three BANDS (TrioDev::gthr0 / gthr2 again: knot values of leg n), every band's records touch only
// the column tiles band_tile[b] .. band_tile[b] + 2; the walk sorts the triplets by band, every band starts on an even slot
// of the stage, and a step is 2 x 3 = 6 MFMAs into the band's six accumulator tiles.  As with the grouped windows the steps of
// a band are ONE asm statement (three per pass, straight-line between them): in C++ the compiler copies the twelve accumulator
// tiles at every loop boundary and spills ~130 registers; here they stay where they are.
// one step (two records of 64 doubles = 0x400 bytes) at byte offset OFF from the running addresses
#define UF3_BAND_STEP(OFF)                                                                      \
        "ds_read_b64 %[la0], %[va0] offset:" OFF "\n"                                           \
        "ds_read_b64 %[da0], %[vd0] offset:" OFF "\n"                                           \
        "ds_read_b64 %[la1], %[va1] offset:" OFF "\n"                                           \
        "ds_read_b64 %[da1], %[vd1] offset:" OFF "\n"                                           \
        "ds_read_b64 %[mb0], %[vm0] offset:" OFF "\n"                                           \
        "ds_read_b64 %[nb0], %[vn0] offset:" OFF "\n"                                           \
        "ds_read_b64 %[mb1], %[vm1] offset:" OFF "\n"                                           \
        "ds_read_b64 %[nb1], %[vn1] offset:" OFF "\n"                                           \
        "ds_read_b64 %[mb2], %[vm2] offset:" OFF "\n"                                           \
        "ds_read_b64 %[nb2], %[vn2] offset:" OFF "\n"                                           \
        "s_waitcnt lgkmcnt(6)\n"                                                                \
        "v_mul_f64 %[la0], %[la0], %[da0]\n"                                                    \
        "v_mul_f64 %[la1], %[la1], %[da1]\n"                                                    \
        "s_waitcnt lgkmcnt(4)\n"                                                                \
        "v_mul_f64 %[mb0], %[mb0], %[nb0]\n"                                                    \
        "s_waitcnt lgkmcnt(2)\n"                                                                \
        "v_mul_f64 %[mb1], %[mb1], %[nb1]\n"                                                    \
        "s_waitcnt lgkmcnt(0)\n"                                                                \
        "v_mul_f64 %[mb2], %[mb2], %[nb2]\n"                                                    \
        "v_mfma_f64_16x16x4_f64 %[c00], %[la0], %[mb0], %[c00]\n"                               \
        "v_mfma_f64_16x16x4_f64 %[c10], %[la1], %[mb0], %[c10]\n"                               \
        "v_mfma_f64_16x16x4_f64 %[c01], %[la0], %[mb1], %[c01]\n"                               \
        "v_mfma_f64_16x16x4_f64 %[c11], %[la1], %[mb1], %[c11]\n"                               \
        "v_mfma_f64_16x16x4_f64 %[c02], %[la0], %[mb2], %[c02]\n"                               \
        "v_mfma_f64_16x16x4_f64 %[c12], %[la1], %[mb2], %[c12]\n"
#define UF3_BAND_ADVANCE(BYTES)                                                                 \
        "v_add_u32 %[va0], " BYTES ", %[va0]\n"                                                 \
        "v_add_u32 %[vd0], " BYTES ", %[vd0]\n"                                                 \
        "v_add_u32 %[va1], " BYTES ", %[va1]\n"                                                 \
        "v_add_u32 %[vd1], " BYTES ", %[vd1]\n"                                                 \
        "v_add_u32 %[vm0], " BYTES ", %[vm0]\n"                                                 \
        "v_add_u32 %[vn0], " BYTES ", %[vn0]\n"                                                 \
        "v_add_u32 %[vm1], " BYTES ", %[vm1]\n"                                                 \
        "v_add_u32 %[vn1], " BYTES ", %[vn1]\n"                                                 \
        "v_add_u32 %[vm2], " BYTES ", %[vm2]\n"                                                 \
        "v_add_u32 %[vn2], " BYTES ", %[vn2]\n"
__device__ __forceinline__ void banded_steps(unsigned va0, unsigned vd0, unsigned va1, unsigned vd1, unsigned vm0, unsigned vn0,
                                             unsigned vm1, unsigned vn1, unsigned vm2, unsigned vn2, int n_steps,
                                             double4_t &c00, double4_t &c01, double4_t &c02, double4_t &c10, double4_t &c11,
                                             double4_t &c12) {
    double la0, da0, la1, da1, mb0, nb0, mb1, nb1, mb2, nb2;
    // four steps per trip share one advance of the ten operand addresses (the steps reach their records through the
    // instruction's offset field), single steps finish the band
    asm volatile(
        "s_cmp_lt_u32 %[n], 4\n"
        "s_cbranch_scc1 1f\n"
        "0:\n"
        UF3_BAND_STEP("0") UF3_BAND_STEP("0x400") UF3_BAND_STEP("0x800") UF3_BAND_STEP("0xc00")
        "s_sub_u32 %[n], %[n], 4\n"
        UF3_BAND_ADVANCE("0x1000")
        "s_cmp_lt_u32 %[n], 4\n"
        "s_cbranch_scc0 0b\n"
        "1:\n"
        "s_cmp_lt_u32 %[n], 2\n"
        "s_cbranch_scc1 2f\n"
        UF3_BAND_STEP("0") UF3_BAND_STEP("0x400")
        "s_sub_u32 %[n], %[n], 2\n"
        UF3_BAND_ADVANCE("0x800")
        "2:\n"
        "s_cmp_eq_u32 %[n], 0\n"
        "s_cbranch_scc1 3f\n"
        UF3_BAND_STEP("0")
        "3:\n"
        "s_nop 15\n"
        "s_nop 3\n"
        : [c00] "+v"(c00), [c01] "+v"(c01), [c02] "+v"(c02), [c10] "+v"(c10), [c11] "+v"(c11), [c12] "+v"(c12),
          [va0] "+v"(va0), [vd0] "+v"(vd0), [va1] "+v"(va1), [vd1] "+v"(vd1), [vm0] "+v"(vm0), [vn0] "+v"(vn0),
          [vm1] "+v"(vm1), [vn1] "+v"(vn1), [vm2] "+v"(vm2), [vn2] "+v"(vn2), [n] "+s"(n_steps),
          [la0] "=&v"(la0), [da0] "=&v"(da0), [la1] "=&v"(la1), [da1] "=&v"(da1), [mb0] "=&v"(mb0), [nb0] "=&v"(nb0),
          [mb1] "=&v"(mb1), [nb1] "=&v"(nb1), [mb2] "=&v"(mb2), [nb2] "=&v"(nb2)
        :
        : "scc", "memory");
}
#undef UF3_BAND_STEP
#undef UF3_BAND_ADVANCE

// STATIC012: the bands' first column tiles are 0, 1, 2 (what the 6 x 6 x 12 window of an untrimmed block gives): the three asm
// statements of a pass then name their accumulator tiles at compile time and follow each other without a branch between them.
// Chosen per tile at run time (four-way switches around the statements), the compiler copies accumulator tiles at every join --
// ~45 64-bit moves per band and pass, a quarter of the launch's vector instructions.
template <bool WANT_E, bool IMG, bool RL, bool STATIC012>
__device__ __forceinline__ void trio_block_banded(const FeatArgs &A, const BasisDev *B, const KnotRec *recs, const FrameGeom &g,
                                                  const WaveLds &w, int m, int sm, int t, const ESink &es,
                                                  const int (&fragp)[4], const int *dsrc) {
    constexpr int RT = 2, CT = 6, STRIDE = 64;
    constexpr bool WANT_F = true;
    const int lane = lane_id();
    const TrioDev td_copy = load_const(A.trios + t);
    const TrioDev *td = &td_copy;
    PhaseClock pc;
    TrioWalk k;
    trio_walk_setup<WANT_F, IMG>(A, w, td->sc, td->sa, td->sb, sm, k);
    const int F = load_const(&B->F);
    const int lo_l = td->lo[0], lo_m = td->lo[1], lo_n = td->lo[2];
    const int ext_l = td->ext[0], ext_m = td->ext[1], ext_n = td->ext[2];
    const DenseLayout dl = dense_layout(ext_l, ext_m, ext_n);          // (dl.stride == STRIDE: the host sends no other window here)
    const int r16 = lane & 15, slot = (lane >> 4) & 1;
    const int inv_l = (65536 + ext_l - 1) / ext_l, inv_m = (65536 + ext_m - 1) / ext_m;
    // LDS byte addresses of this lane's operands in record (lane >> 5) of the stage
    const unsigned stage_lds = (unsigned)(size_t)(__attribute__((address_space(3))) double *)w.stage + (lane >> 5) * (STRIDE * 8);
    unsigned aL[RT], aD[RT], bM[CT], bN[CT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        const int row = rt * 16 + r16;
        const int c = (row * inv_l) >> 16, pl = row - c * ext_l;
        const bool ok = c < 3 || (c == 3 && WANT_E);
        aL[rt] = stage_lds + 8 * (ok ? 2 * pl + (c == 3 ? 1 : slot) : dl.oZ);
        aD[rt] = stage_lds + 8 * (ok ? dl.oD + 2 * c + slot : dl.oZ);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const int col = ct * 16 + r16;                       // columns n-major: col = n * ext_m + m
        const int pn = (col * inv_m) >> 16, pm = col - pn * ext_m;
        const bool ok = pn < ext_n;
        bM[ct] = stage_lds + 8 * (ok ? dl.oM + 2 * pm + slot : dl.oZ);
        bN[ct] = stage_lds + 8 * (ok ? dl.oN + 2 * pn + slot : dl.oZ);
    }
    const int li = (lane * 21846) >> 16, leg = lane - 3 * li;
    LegDev lg;
    lg.rec_off = leg == 0 ? td->leg[0].rec_off : (leg == 1 ? td->leg[1].rec_off : td->leg[2].rec_off);
    lg.nk = leg == 0 ? td->leg[0].nk : (leg == 1 ? td->leg[1].nk : td->leg[2].nk);
    lg.t0 = leg == 0 ? td->leg[0].t0 : (leg == 1 ? td->leg[1].t0 : td->leg[2].t0);
    lg.tlast = leg == 0 ? td->leg[0].tlast : (leg == 1 ? td->leg[1].tlast : td->leg[2].tlast);
    lg.inv_h = leg == 0 ? td->leg[0].inv_h : (leg == 1 ? td->leg[1].inv_h : td->leg[2].inv_h);
    const int w_off = leg == 0 ? 0 : (leg == 1 ? dl.oM : dl.oN);
    const int w_ext = leg == 0 ? ext_l : (leg == 1 ? ext_m : ext_n), w_lo = leg == 0 ? lo_l : (leg == 1 ? lo_m : lo_n);
    const double thr0 = td->gthr0, thr2 = td->gthr2;
    const int bt0 = td->band_tile[0], bt1 = td->band_tile[1], bt2 = td->band_tile[2];     // wave-uniform
    double4_t acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[rt][ct] = double4_t{0, 0, 0, 0};
    pc.lap(1);
    const int nrec = A.dense_nrec, batch = 3 * nrec;
    const double lo_r[3] = {td->leg[0].t0, td->leg[1].t0, td->leg[2].t0};
    const double hi_r[3] = {td->leg[0].tlast, td->leg[1].tlast, td->leg[2].tlast};
    for (int p0 = 0; p0 < k.n_items; p0 += batch) {
        int n_valid, n_g0, n_g01;
        {
            TripletGeom tg;
            bool valid = lane < batch && p0 + lane < k.n_items;
            if (valid) valid = trio_walk_geom<WANT_F, IMG>(A, g, w, td->sa, td->sb, k, m, sm, p0 + lane, tg);
            if (valid)
                valid = (tg.rl > lo_r[0]) & (tg.rl < hi_r[0]) & (tg.rm > lo_r[1]) & (tg.rm < hi_r[1]) &
                        (tg.rn > lo_r[2]) & (tg.rn < hi_r[2]);
            const bool is0 = valid && tg.rn <= thr0, is2 = valid && tg.rn > thr2, is1 = valid && !is0 && !is2;
            const unsigned long long m0 = __ballot(is0), m1 = __ballot(is1), m2 = __ballot(is2);
            n_g0 = __popcll(m0); n_g01 = n_g0 + __popcll(m1);
            n_valid = n_g01 + __popcll(m2);
            if (valid) {
                const int rank = is0 ? mbcnt(m0) : (is1 ? n_g0 + mbcnt(m1) : n_g01 + mbcnt(m2));
                double *gq = w.geo + rank;
                gq[0] = tg.rl; gq[GEO_N] = tg.rm; gq[2 * GEO_N] = tg.rn;
                gq[3 * GEO_N] = tg.a3[0]; gq[4 * GEO_N] = tg.a3[1]; gq[5 * GEO_N] = tg.a3[2];
                ((int2 *)(w.geo + 6 * GEO_N))[rank] = make_int2(tg.i1 | (tg.i2 << 16), tg.centre ? 0 : (tg.first ? 1 : 2));
            }
        }
        wave_sync();
        pc.lap(2);
        for (int base = 0; base < n_valid; base += nrec) {
            const int n_part = min(nrec, n_valid - base);
            // band boundaries inside this pass (records are sorted by band); every band starts on an even slot
            const int b0 = max(0, min(n_part, n_g0 - base)), b1 = max(0, min(n_part, n_g01 - base));
            const int st0 = (b0 + 1) >> 1, st1 = (b1 - b0 + 1) >> 1, st2 = (n_part - b1 + 1) >> 1;
            const int n_staged = 2 * (st0 + st1 + st2);
            const bool mine = li < n_part && !UF3_SKIP(16);
            if (!UF3_SKIP(16)) {
                if (A.dense_stage == 1280) {                   // (the usual stage, 20 records of 64 doubles: cleared whole, no loop)
#pragma unroll
                    for (int u = 0; u < 10; u++) *(double2 *)(w.stage + 2 * lane + u * 2 * WAVE) = double2{0.0, 0.0};
                } else
                    for (int q = 2 * lane; q < n_staged * STRIDE; q += 2 * WAVE) *(double2 *)(w.stage + q) = double2{0.0, 0.0};
            }
            if (mine) {
                const int gi = base + li;
                const double x = w.geo[leg * GEO_N + gi];
                const int2 pk = ((const int2 *)(w.geo + 6 * GEO_N))[gi];
                const double a3 = w.geo[(3 + leg) * GEO_N + gi];
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                const int i1 = pk.x & 0xffff, i2 = pk.x >> 16;
                const double *oc = w.ox + (size_t)leg * A.n3.cap;
                const double oc1 = oc[i1], oi1 = w.oir[i1], oc2 = oc[i2], oi2 = w.oir[i2];
                KnotRec kr;
                double v[4], d[4];
                const int first = load_interval<RL ? 3 : 1>(recs, lg, x, kr) - 3;
                bspline4<WANT_F>(kr, x, v, d);
                double *rec = w.stage + (size_t)(li + (li >= b0 ? (b0 & 1) : 0) + (li >= b1 ? ((b1 - b0) & 1) : 0)) * STRIDE;
                const int cls = pk.y;
                const bool d0 = leg == 0 ? cls != 2 : (leg == 1 ? cls == 2 : false);
                const bool d1 = leg == 0 ? false : (leg == 1 ? cls == 0 : cls != 0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned ws = (unsigned)(first + q - w_lo);
                    if (ws < (unsigned)w_ext)
                        *(double2 *)(rec + w_off + 2 * ws) = double2{d0 ? d[q] : v[q], d1 ? d[q] : v[q]};
                }
                const double u1 = oc1 * oi1, u2 = oc2 * oi2;
                *(double2 *)(rec + dl.oD + 2 * leg) = double2{u1, cls == 0 ? u2 : a3};
                if (leg == 0 && cls == 0) rec[dl.oD + 6] = 1.0;
            }
            wave_sync();
            pc.lap(4);
            if (!UF3_SKIP(8)) {
                // band b's steps go to the column tiles band_tile[b] .. + 2 (wave-uniform switches: the tiles are registers)
                const unsigned off1 = (unsigned)(2 * st0) * (STRIDE * 8), off2 = off1 + (unsigned)(2 * st1) * (STRIDE * 8);
#define UF3_BAND(BT, OFF, NST)                                                                                                    \
    if ((BT) == 0) banded_steps(aL[0] + (OFF), aD[0] + (OFF), aL[1] + (OFF), aD[1] + (OFF), bM[0] + (OFF), bN[0] + (OFF), bM[1] + (OFF), bN[1] + (OFF), \
                                bM[2] + (OFF), bN[2] + (OFF), NST, acc[0][0], acc[0][1], acc[0][2], acc[1][0], acc[1][1], acc[1][2]);   \
    else if ((BT) == 1) banded_steps(aL[0] + (OFF), aD[0] + (OFF), aL[1] + (OFF), aD[1] + (OFF), bM[1] + (OFF), bN[1] + (OFF), bM[2] + (OFF), bN[2] + (OFF), \
                                     bM[3] + (OFF), bN[3] + (OFF), NST, acc[0][1], acc[0][2], acc[0][3], acc[1][1], acc[1][2], acc[1][3]); \
    else if ((BT) == 2) banded_steps(aL[0] + (OFF), aD[0] + (OFF), aL[1] + (OFF), aD[1] + (OFF), bM[2] + (OFF), bN[2] + (OFF), bM[3] + (OFF), bN[3] + (OFF), \
                                     bM[4] + (OFF), bN[4] + (OFF), NST, acc[0][2], acc[0][3], acc[0][4], acc[1][2], acc[1][3], acc[1][4]); \
    else banded_steps(aL[0] + (OFF), aD[0] + (OFF), aL[1] + (OFF), aD[1] + (OFF), bM[3] + (OFF), bN[3] + (OFF), bM[4] + (OFF), bN[4] + (OFF),       \
                      bM[5] + (OFF), bN[5] + (OFF), NST, acc[0][3], acc[0][4], acc[0][5], acc[1][3], acc[1][4], acc[1][5]);
                // (the records of a walk step are sorted by band: a pass mostly holds one or two of them)
                if (STATIC012) {                       // (a band without steps falls through its statement)
                    banded_steps(aL[0], aD[0], aL[1], aD[1], bM[0], bN[0], bM[1], bN[1], bM[2], bN[2], st0,
                                 acc[0][0], acc[0][1], acc[0][2], acc[1][0], acc[1][1], acc[1][2]);
                    banded_steps(aL[0] + off1, aD[0] + off1, aL[1] + off1, aD[1] + off1, bM[1] + off1, bN[1] + off1, bM[2] + off1, bN[2] + off1,
                                 bM[3] + off1, bN[3] + off1, st1, acc[0][1], acc[0][2], acc[0][3], acc[1][1], acc[1][2], acc[1][3]);
                    banded_steps(aL[0] + off2, aD[0] + off2, aL[1] + off2, aD[1] + off2, bM[2] + off2, bN[2] + off2, bM[3] + off2, bN[3] + off2,
                                 bM[4] + off2, bN[4] + off2, st2, acc[0][2], acc[0][3], acc[0][4], acc[1][2], acc[1][3], acc[1][4]);
                } else {
                    if (st0 > 0) { UF3_BAND(bt0, 0u, st0) }
                    if (st1 > 0) { UF3_BAND(bt1, off1, st1) }
                    if (st2 > 0) { UF3_BAND(bt2, off2, st2) }
                }
#undef UF3_BAND
            }
            pc.lap(5);
            wave_sync();
        }
    }
    dense_fold<WANT_E, WANT_F, RT, CT>(A, w, td, dl, m, F, es, fragp, dsrc, acc);
    wave_sync();
    pc.lap(6);
}

// ---- grouped n windows: the 3 x 3 x 9 blocks of the reference's default trims ------------------------------------
// The 27 columns (n, m) of such a window need two 16-column tiles, but a record touches four consecutive n bins only.
// Cover the n bins by three overlapping groups of five ([0,4], [2,6], [4,8]: 15 columns each = ONE tile): every record
// falls into exactly one group (TrioDev::gthr0 / gthr2), the walk sorts its triplets by group, a step of two records
// of one group is a single MFMA into that group's accumulator, and the step at a group boundary runs once per group
// with the other record's B operand zeroed.  137 steps -> ~165 MFMA per atom instead of 222 (two tiles with skipping),
// a staged record shrinks from 40 to 32 doubles.  The three accumulator tiles are added into the usual dumped window
// (rows (c, l), columns n-major) before the symmetry fold.
// The MFMA steps of one staged pass of the grouped windows, written out by hand: records of group g fill the even-aligned
// slot range that follows group g - 1 (an odd group is padded with a zero record), so one set of four operand addresses
// runs through the stage and group g's steps accumulate into its own tile.  Why assembly: with three accumulator tiles
// alive across three loops the compiler copies whole tiles at every loop boundary (about 100 v_mov per pass, 1400 per
// atom).  Inside one asm statement the tiles stay where they are: per step 4 LDS reads, 2 products, 2 address updates
// (two steps per trip share them through the immediate offset) and the MFMA.
// Spacing: a VALU result needs two issue slots before an MFMA may read it as A / B (the compiler's s_nop 1); MFMAs that
// chain through the accumulator need none; 18 wait states after the last MFMA before other code may read a tile.
#define UF3_GROUP_STEPS(G)                                                                      \
    "s_cmp_lt_u32 %[n" G "], 2\n"                                                               \
    "s_cbranch_scc1 1f\n"                                                                       \
    "0:\n"                                                                                      \
    "ds_read_b64 %[la], %[va]\n"                                                                \
    "ds_read_b64 %[da], %[vd]\n"                                                                \
    "ds_read_b64 %[mb], %[vm]\n"                                                                \
    "ds_read_b64 %[nb], %[vn]\n"                                                                \
    "ds_read_b64 %[la2], %[va] offset:544\n"                                                    \
    "ds_read_b64 %[da2], %[vd] offset:544\n"                                                    \
    "ds_read_b64 %[mb2], %[vm] offset:544\n"                                                    \
    "ds_read_b64 %[nb2], %[vn] offset:544\n"                                                    \
    "s_sub_u32 %[n" G "], %[n" G "], 2\n"                                                       \
    "v_add_u32 %[va], 0x440, %[va]\n"                                                           \
    "v_add_u32 %[vd], 0x440, %[vd]\n"                                                           \
    "v_add_u32 %[vm], 0x440, %[vm]\n"                                                           \
    "v_add_u32 %[vn], 0x440, %[vn]\n"                                                           \
    "s_waitcnt lgkmcnt(4)\n"                                                                    \
    "v_mul_f64 %[la], %[la], %[da]\n"                                                           \
    "v_mul_f64 %[mb], %[mb], %[nb]\n"                                                           \
    "s_waitcnt lgkmcnt(0)\n"                                                                    \
    "v_mul_f64 %[la2], %[la2], %[da2]\n"                                                        \
    "v_mul_f64 %[mb2], %[mb2], %[nb2]\n"                                                        \
    "v_mfma_f64_16x16x4_f64 %[acc" G "], %[la], %[mb], %[acc" G "]\n"                           \
    "s_cmp_lt_u32 %[n" G "], 2\n"                                                               \
    "v_mfma_f64_16x16x4_f64 %[acc" G "], %[la2], %[mb2], %[acc" G "]\n"                         \
    "s_cbranch_scc0 0b\n"                                                                       \
    "1:\n"                                                                                      \
    "s_cmp_eq_u32 %[n" G "], 0\n"                                                               \
    "s_cbranch_scc1 2f\n"                                                                       \
    "ds_read_b64 %[la], %[va]\n"                                                                \
    "ds_read_b64 %[da], %[vd]\n"                                                                \
    "ds_read_b64 %[mb], %[vm]\n"                                                                \
    "ds_read_b64 %[nb], %[vn]\n"                                                                \
    "v_add_u32 %[va], 0x220, %[va]\n"                                                           \
    "v_add_u32 %[vd], 0x220, %[vd]\n"                                                           \
    "v_add_u32 %[vm], 0x220, %[vm]\n"                                                           \
    "v_add_u32 %[vn], 0x220, %[vn]\n"                                                           \
    "s_waitcnt lgkmcnt(2)\n"                                                                    \
    "v_mul_f64 %[la], %[la], %[da]\n"                                                           \
    "s_waitcnt lgkmcnt(0)\n"                                                                    \
    "v_mul_f64 %[mb], %[mb], %[nb]\n"                                                           \
    "s_nop 1\n"                                                                                 \
    "v_mfma_f64_16x16x4_f64 %[acc" G "], %[la], %[mb], %[acc" G "]\n"                           \
    "2:\n"

__device__ __forceinline__ void grouped_pass_steps(unsigned va, unsigned vd, unsigned vm, unsigned vn, int n0, int n1, int n2,
                                                   double4_t &acc0, double4_t &acc1, double4_t &acc2) {
    double la, da, mb, nb, la2, da2, mb2, nb2;
    asm volatile(UF3_GROUP_STEPS("0") UF3_GROUP_STEPS("1") UF3_GROUP_STEPS("2")
                 "s_nop 15\n"
                 "s_nop 3\n"
                 : [acc0] "+v"(acc0), [acc1] "+v"(acc1), [acc2] "+v"(acc2), [va] "+v"(va), [vd] "+v"(vd), [vm] "+v"(vm),
                   [vn] "+v"(vn), [n0] "+s"(n0), [n1] "+s"(n1), [n2] "+s"(n2), [la] "=&v"(la), [da] "=&v"(da), [mb] "=&v"(mb),
                   [nb] "=&v"(nb), [la2] "=&v"(la2), [da2] "=&v"(da2), [mb2] "=&v"(mb2), [nb2] "=&v"(nb2)
                 :
                 : "scc", "memory");
}
#undef UF3_GROUP_STEPS

// What the lanes of a grouped window need that depends only on the window's LAYOUT -- knot sequences of the three legs, window
// box, group thresholds: TrioDev::layout numbers the distinct ones, a basis built with one set of per-interaction settings
// has a single one --, not on the block or the atom.  Computed when a block of another layout comes along (in practice once
// per wave and launch) instead of at every (atom, block): the per-block set-up was 220 vector instructions, a quarter of
// them spills of the descriptor's scalars.
struct GroupedLayout {
    int id;                               // TrioDev::layout this was computed for (-1: none yet)
    int ext_l, ext_n, lo_l, lo_m, lo_n;   // (ext_m == 3)
    double lo_r[3], hi_r[3], gthr0, gthr2;
    // per lane, packed (they stay in registers across blocks and atoms):
    unsigned ops;                         // byte offsets of the four MFMA operands inside a staged record: va | vd << 8 | vm << 16 | vn << 24
    unsigned legpack;                     // staging role (li, leg): number of the leg's first window row (TrioDev::wrow) | knots << 16 | first pair of its window inside a record << 24
    double leg_t0, leg_inv_h;             // ... and its support start / interval guess
    unsigned tiles01, tiles23;            // accumulator element v -> double index inside a dumped group tile (16 bits each, 0xffff: none)
};

template <bool WANT_E>
__device__ __forceinline__ void grouped_layout_setup(const FeatArgs &A, int t, const int (&fragp)[4], GroupedLayout &L) {
    constexpr int GW = 5;
    const int lane = lane_id();
    const TrioDev td_copy = load_const(A.trios + t);
    const TrioDev *td = &td_copy;
    L.id = td->layout;
    L.ext_l = td->ext[0]; L.ext_n = td->ext[2];
    L.lo_l = td->lo[0]; L.lo_m = td->lo[1]; L.lo_n = td->lo[2];
    for (int q = 0; q < 3; q++) { L.lo_r[q] = td->leg[q].t0; L.hi_r[q] = td->leg[q].tlast; }
    L.gthr0 = td->gthr0; L.gthr2 = td->gthr2;
    const int ext_l = L.ext_l, ext_m = 3;
    const int oM = 2 * ext_l, oN = oM + 2 * ext_m, oD = oN + 2 * GW, oZ = oD + 8;       // (oZ + 2 <= 32: ext_l, ext_m <= 3)
    const int r16 = lane & 15, slot = (lane >> 4) & 1;
    const int inv_l = (65536 + ext_l - 1) / ext_l, inv_m = (65536 + ext_m - 1) / ext_m;
    {
        const int c = (r16 * inv_l) >> 16, pl = r16 - c * ext_l;
        const bool ok = c < 3 || (c == 3 && WANT_E);
        const unsigned va = 8 * (ok ? 2 * pl + (c == 3 ? 1 : slot) : oZ), vd = 8 * (ok ? oD + 2 * c + slot : oZ);
        const int nl = (r16 * inv_m) >> 16, pm = r16 - nl * ext_m;         // column = (n - group base) * ext_m + m
        const unsigned vm = 8 * (nl < GW ? oM + 2 * pm + slot : oZ), vn = 8 * (nl < GW ? oN + 2 * nl + slot : oZ);
        L.ops = va | (vd << 8) | (vm << 16) | (vn << 24);                   // (a record is 256 bytes)
    }
    const int li = (lane * 21846) >> 16, leg = lane - 3 * li;
    const int rec_off = leg == 0 ? td->wrow[0] : (leg == 1 ? td->wrow[1] : td->wrow[2]);     // (the leg's window rows)
    const int nk = leg == 0 ? td->leg[0].nk : (leg == 1 ? td->leg[1].nk : td->leg[2].nk);
    L.leg_t0 = leg == 0 ? td->leg[0].t0 : (leg == 1 ? td->leg[1].t0 : td->leg[2].t0);
    L.leg_inv_h = leg == 0 ? td->leg[0].inv_h : (leg == 1 ? td->leg[1].inv_h : td->leg[2].inv_h);
    L.legpack = (unsigned)rec_off | ((unsigned)nk << 16) | ((unsigned)(leg == 0 ? 0 : (leg == 1 ? oM : oN)) << 24);   // (host: rec_off < 65536, nk < 256)
    // dumped tiles: [group][row 4 c + l][16 columns] -- component c always 4 rows on, whatever ext_l, so that the fold reads a
    // column's four components at fixed distances
    unsigned ts[4];
#pragma unroll
    for (int v = 0; v < 4; v++) {
        // (from the probed table itself, not from fragp: layouts change rarely -- never within a basis with one set of 3-body
        // settings --, and in a grouped-only launch nothing else would keep those four registers alive)
        const int row = load_const(A.frag + (lane * 4 + v) * 2), col = load_const(A.frag + (lane * 4 + v) * 2 + 1);
        const int c = (row * inv_l) >> 16, pl = row - c * ext_l;
        // (accumulator rows past the window -- 4 ext_l .. 15 -- are parked in row 3 of the tile when ext_l < 4, a row the fold
        // never reads, so that the dump needs no lane mask; with ext_l == 4 every row is in use)
        ts[v] = row < 4 * ext_l ? (unsigned)((4 * c + pl) * 16 + col) : (ext_l < 4 ? (unsigned)(3 * 16 + col) : 0xffffu);
    }
    L.tiles01 = ts[0] | (ts[1] << 16); L.tiles23 = ts[2] | (ts[3] << 16);
}

// th: the block's dispatch header; gsrc: its fold table (see FeatArgs::gsrc), gsrc_off = th.grouped >> 8
template <bool WANT_E, bool IMG, bool ROWS_LDS>        // ROWS_LDS: recs (the window rows behind them) is an LDS copy
__device__ __forceinline__ void trio_block_grouped(const FeatArgs &A, const BasisDev *B, const KnotRec *recs, const FrameGeom &g,
                                                   const WaveLds &w, int m, int sm, const TrioHead &th, const ESink &es,
                                                   const GroupedLayout &L, const unsigned short *gsrc) {
    constexpr int STRIDE = 34, GW = 5, NG = 3;   // (34, not 32: the same slot of consecutive records falls into different banks)
    constexpr bool WANT_F = true;
    PhaseClock pc;
    const int lane = lane_id();
    TrioWalk k;
    trio_walk_setup<WANT_F, IMG>(A, w, th.sc, th.sa, th.sb, sm, k);
    const int ncol = th.ncol;                             // (the row stride A.ld is a kernel argument, a scalar: read through B it was a
                                                          // vector load once per block and a wait on the row stores of the block before)
    const unsigned short *gsrc_blk = gsrc + (th.grouped >> 8);      // the block's fold table (LDS when it fits)
    const int ext_l = L.ext_l, ext_m = 3;
    const int oM = 2 * ext_l, oN = oM + 2 * ext_m, oD = oN + 2 * GW;
    const int li = (lane * 21846) >> 16, leg = lane - 3 * li;
    double4_t acc[NG][1][1];
#pragma unroll
    for (int q = 0; q < NG; q++) acc[q][0][0] = double4_t{0, 0, 0, 0};
    const int nrec = A.dense_nrec, batch = 3 * nrec;
    // LDS byte addresses of this lane's four operands in record (lane >> 5) of the stage
    const unsigned stage_lds = (unsigned)(size_t)(__attribute__((address_space(3))) double *)w.stage + (lane >> 5) * (STRIDE * 8);
    const unsigned a_va = stage_lds + (L.ops & 0xffu), a_vd = stage_lds + ((L.ops >> 8) & 0xffu),
                   a_vm = stage_lds + ((L.ops >> 16) & 0xffu), a_vn = stage_lds + (L.ops >> 24);
    LegDev lg;                                                  // (what load_interval looks at)
    lg.rec_off = (int)(L.legpack & 0xffffu); lg.nk = (int)((L.legpack >> 16) & 0xffu); lg.t0 = L.leg_t0; lg.inv_h = L.leg_inv_h;
    lg.tlast = 0.0;
    const int w_off = (int)(L.legpack >> 24);
    // the pair at oZ of every record slot -- what the masked operands read -- must be zero: the fold of the block before dumped its
    // tiles over the stage (the only place these pairs are ever written)
    if (lane < A.dense_stage / STRIDE) *(double2 *)(w.stage + lane * STRIDE + (oD + 8)) = double2{0.0, 0.0};
    pc.lap(1);
    for (int p0 = 0; p0 < k.n_items; p0 += batch) {
        int n_valid, n_g0, n_g01;
        {
            TripletGeom tg;
            bool valid = lane < batch && p0 + lane < k.n_items;
            if (valid) valid = trio_walk_geom<WANT_F, IMG>(A, g, w, th.sa, th.sb, k, m, sm, p0 + lane, tg);
            if (valid)
                valid = (tg.rl > L.lo_r[0]) & (tg.rl < L.hi_r[0]) & (tg.rm > L.lo_r[1]) & (tg.rm < L.hi_r[1]) &
                        (tg.rn > L.lo_r[2]) & (tg.rn < L.hi_r[2]);
            const bool is0 = valid && tg.rn <= L.gthr0, is2 = valid && tg.rn > L.gthr2, is1 = valid && !is0 && !is2;
            const unsigned long long m0 = __ballot(is0), m1 = __ballot(is1), m2 = __ballot(is2);
            n_g0 = __popcll(m0); n_g01 = n_g0 + __popcll(m1);
            n_valid = n_g01 + __popcll(m2);
            if (valid) {
                const int rank = is0 ? mbcnt(m0) : (is1 ? n_g0 + mbcnt(m1) : n_g01 + mbcnt(m2));
                double *gq = w.geo + rank;
                gq[0] = tg.rl; gq[GEO_N] = tg.rm; gq[2 * GEO_N] = tg.rn;
                gq[3 * GEO_N] = tg.a3[0]; gq[4 * GEO_N] = tg.a3[1]; gq[5 * GEO_N] = tg.a3[2];
                ((int2 *)(w.geo + 6 * GEO_N))[rank] =
                    make_int2(tg.i1 | (tg.i2 << 16), (tg.centre ? 0 : (tg.first ? 1 : 2)) | ((is0 ? 0 : (is1 ? 1 : 2)) << 2));
            }
        }
        wave_sync();
        pc.lap(2);
        for (int base = 0; base < n_valid; base += nrec) {
            const int n_part = min(nrec, n_valid - base);
            // group boundaries inside this pass (records are sorted by group): [0, b0) group 0, [b0, b1) group 1, rest 2;
            // every group starts on an even slot (its steps never share a record pair with another group's)
            const int b0 = max(0, min(n_part, n_g0 - base)), b1 = max(0, min(n_part, n_g01 - base));
            const int st0 = (b0 + 1) >> 1, st1 = (b1 - b0 + 1) >> 1, st2 = (n_part - b1 + 1) >> 1;
            const bool mine = li < n_part && !UF3_SKIP(16);
            // Nothing is cleared: every slot of every record of the pass is written below (window slots from the leg's window row,
            // zeros included), the pair at oZ is never written, and of a padding record -- the slot after an odd group -- only the
            // l window matters (its A operand must be zero; the rest is multiplied by it, and stale stage contents are finite).
            if (!UF3_SKIP(16) && lane < 9) {
                const int pr = (lane * 21846) >> 16, ch = lane - 3 * pr;            // padding record pr, pair ch of its l window
                const int odd0 = b0 & 1, odd1 = (b1 - b0) & 1, odd2 = (n_part - b1) & 1;
                const int slot = pr == 0 ? b0 : (pr == 1 ? b1 + odd0 : n_part + odd0 + odd1);
                if ((pr == 0 ? odd0 : (pr == 1 ? odd1 : odd2)) && ch < ext_l)
                    *(double2 *)(w.stage + (size_t)slot * STRIDE + 2 * ch) = double2{0.0, 0.0};
            }
            if (mine) {
                const int gi = base + li;
                const double x = w.geo[leg * GEO_N + gi];
                const int2 pk = ((const int2 *)(w.geo + 6 * GEO_N))[gi];
                const double a3 = w.geo[(3 + leg) * GEO_N + gi];
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                const int i1 = pk.x & 0xffff, i2 = pk.x >> 16;
                const double *oc = w.ox + (size_t)leg * A.n3.cap;
                const double oc1 = oc[i1], oi1 = w.oir[i1], oc2 = oc[i2], oi2 = w.oir[i2];
                // the window row of x's knot interval: the whole row at the guessed interval at once, again only when the guess
                // was off (non-uniform knots, x on a boundary)
                const int hi = lg.nk - 5;
                int iv = 3 + (int)((x - lg.t0) * lg.inv_h);
                iv = iv < 3 ? 3 : (iv > hi ? hi : iv);
                const double2 *rows = (const double2 *)recs + A.wrow_base + 9 * (lg.rec_off - 3);   // (rows of nine pairs, see uf3_basis_create)
                double2 kn, cf[8];
                // (through the LDS or the global address space explicitly: a generic pointer makes flat loads)
                typedef double __attribute__((ext_vector_type(2))) Pair;
                typedef const __attribute__((address_space(3))) Pair *LdsPairs;
                typedef const __attribute__((address_space(1))) Pair *GlobalPairs;
                typedef typename std::conditional<ROWS_LDS, LdsPairs, GlobalPairs>::type Pairs;
                auto load_row = [&](int i) {
                    Pairs q = (Pairs)(const Pair *)(rows + 9 * i);
                    Pair t0 = q[0], t[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) t[e] = q[1 + e];
                    kn = double2{t0.x, t0.y};
#pragma unroll
                    for (int e = 0; e < 8; e++) cf[e] = double2{t[e].x, t[e].y};
                    __builtin_amdgcn_sched_group_barrier(0x120, 9, 0);                   // DS or VMEM reads
                };
                load_row(iv);
                if (__builtin_expect(x > kn.y && iv < hi, 0)) {
                    do { ++iv; load_row(iv); } while (x > kn.y && iv < hi);
                } else if (__builtin_expect(x <= kn.x && iv > 3, 0)) {
                    do { --iv; load_row(iv); } while (x <= kn.x && iv > 3);
                }
                const double u = x - kn.x;
                const int cls = pk.y & 3, grp = pk.y >> 2;
                // first window slot of the row's four functions (see uf3_basis_create)
                const int sb = leg == 2 ? max(0, min(1, iv - 3 - L.lo_n - 2 * grp)) : 0;
                double *rec = w.stage + (size_t)(li + (li >= b0 ? (b0 & 1) : 0) + (li >= b1 ? ((b1 - b0) & 1) : 0)) * STRIDE;
                double *win = rec + w_off + 2 * sb;
                // which of (value, derivative) goes to K slot 0 / 1 (table in the header of this section)
                const bool d0 = leg == 0 ? cls != 2 : (leg == 1 ? cls == 2 : false);
                const bool d1 = leg == 0 ? false : (leg == 1 ? cls == 0 : cls != 0);
                const double u1 = oc1 * oi1, u2 = oc2 * oi2;
                const double2 dir = double2{u1, cls == 0 ? u2 : a3};
                // five stores per lane.  Leg n: its four functions and the zero of the fifth slot; legs l and m (three slots): three
                // functions, then leg l the energy flag and both their direction pair; leg n's direction pair is the sixth
#pragma unroll
                for (int fq = 0; fq < 4; fq++) {
                    const double c0 = cf[2 * fq].x, c1 = cf[2 * fq].y, c2 = cf[2 * fq + 1].x, c3 = cf[2 * fq + 1].y;
                    double pv = fma(c3, u, c2), pd = fma(c3, u, pv);            // Horner, value and derivative together
                    pv = fma(pv, u, c1); pd = fma(pd, u, pv);
                    pv = fma(pv, u, c0);
                    double2 val = double2{d0 ? pd : pv, d1 ? pd : pv};
                    double *dst = win + 2 * fq;
                    bool on = fq < ext_l || leg != 0;                            // (a narrower l window ends where m's begins)
                    if (fq == 3) {
                        if (leg == 0) { val = double2{cls == 0 ? 1.0 : 0.0, 0.0}; dst = rec + oD + 6; on = true; }
                        else if (leg == 1) { val = dir; dst = rec + oD + 2; }
                    }
                    if (on) *(double2 *)dst = val;
                }
                if (leg != 1) *(double2 *)(leg == 0 ? rec + oD : rec + w_off + (sb ? 0 : 8)) = leg == 0 ? dir : double2{0.0, 0.0};
                if (leg == 2) *(double2 *)(rec + oD + 4) = dir;
            }
            wave_sync();
            pc.lap(4);
            if (!UF3_SKIP(8)) grouped_pass_steps(a_va, a_vd, a_vm, a_vn, st0, st1, st2, acc[0][0][0], acc[1][0][0], acc[2][0][0]);
            pc.lap(5);
            wave_sync();
        }
    }
    // Fold.  The three group tiles go to LDS side by side with plain stores ([group][row 4 c + l][16 columns]: no zero fill, no
    // read-modify-write of a shared window) and every column of the block then sums its source bins over the one to three
    // groups that hold them.  Where each (source, group) sits inside the tiles is a table made by the host (FeatArgs::gsrc: six
    // addresses per column; a combination that does not apply points at column 15 of tile 0, which no record ever touches and
    // is therefore zero): 24 reads at fixed component distances and their sum, no address arithmetic, no weights.
    double *tiles = w.stage;                                     // 3 x 16 x 16 doubles = the host's minimum stage
    if (UF3_SKIP(128)) return;                                    // (ablation: no fold, no rows)
    // the fold table's entries of this lane's (at most two) columns: six tile addresses each
    // (through the LDS or the global address space explicitly: as a generic pointer it is a flat load, whose results the
    // compiler can only wait for with vmcnt(0) -- i.e. together with the row stores of the round before)
    int3 fs[2];
    if (A.gsrc_lds) {
        typedef const __attribute__((address_space(3))) int *LdsInts;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            LdsInts q = (LdsInts)(const int *)(gsrc_blk + 6 * min(lane + it * WAVE, ncol - 1));
            fs[it] = make_int3(q[0], q[1], q[2]);
        }
    } else {
        typedef const __attribute__((address_space(1))) int *GlobalInts;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            GlobalInts q = (GlobalInts)(const int *)(gsrc_blk + 6 * min(lane + it * WAVE, ncol - 1));
            fs[it] = make_int3(q[0], q[1], q[2]);
        }
    }
    const unsigned ts[4] = {L.tiles01 & 0xffffu, L.tiles01 >> 16, L.tiles23 & 0xffffu, L.tiles23 >> 16};
#pragma unroll
    for (int grp = 0; grp < NG; grp++)
#pragma unroll
        for (int v = 0; v < 4; v++)
            if (ts[v] != 0xffffu) tiles[grp * 256 + ts[v]] = acc[grp][0][0][v];
    wave_sync();
    pc.lap(7);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int col = lane + it * WAVE;
        if (it * WAVE >= ncol) break;                              // (wave-uniform: blocks of <= 64 columns fold in one round)
        if (col >= ncol) continue;
        const unsigned a[6] = {(unsigned)fs[it].x & 0xffffu, (unsigned)fs[it].x >> 16, (unsigned)fs[it].y & 0xffffu,
                               (unsigned)fs[it].y >> 16, (unsigned)fs[it].z & 0xffffu, (unsigned)fs[it].z >> 16};
        double sum[4];
        if (th.nsrc == 1) {                                        // (wave-uniform: no symmetry images -- the second source's
            double tv[3][4];                                       //  three entries all point at the zero)
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const double *tp = tiles + a[q];
                tv[q][0] = tp[0]; tv[q][1] = tp[64]; tv[q][2] = tp[128];
                tv[q][3] = WANT_E ? tp[192] : 0.0;
            }
            __builtin_amdgcn_sched_group_barrier(0x100, WANT_E ? 12 : 9, 0);    // all DS reads first
#pragma unroll
            for (int c = 0; c < 4; c++) sum[c] = (tv[0][c] + tv[1][c]) + tv[2][c];
        } else {
            double tv[6][4];
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const double *tp = tiles + a[q];
                tv[q][0] = tp[0]; tv[q][1] = tp[64]; tv[q][2] = tp[128];
                tv[q][3] = WANT_E ? tp[192] : 0.0;
            }
            __builtin_amdgcn_sched_group_barrier(0x100, WANT_E ? 24 : 18, 0);   // all DS reads first
#pragma unroll
            for (int c = 0; c < 4; c++) sum[c] = ((tv[0][c] + tv[1][c]) + (tv[2][c] + tv[3][c])) + (tv[4][c] + tv[5][c]);
        }
        pc.lap(8);
        if (!UF3_SKIP(32)) {
            // (streaming stores: the rows are not read again by this launch -- they should not push the neighbour lists, which
            // are, out of the L2)
            double *dst = A.x_f + (size_t)m * 3 * A.ld + th.col + col;
            __builtin_nontemporal_store(sum[0], dst); __builtin_nontemporal_store(sum[1], dst + A.ld);
            __builtin_nontemporal_store(sum[2], dst + 2 * (size_t)A.ld);
        }
        pc.lap(9);
        if (WANT_E && !UF3_SKIP(256)) es.add(th.col + col, sum[3]);
    }
    wave_sync();
    pc.lap(6);
}

// 2-body columns of atom m: lanes <-> neighbour images (the candidates collected in LDS).  Each lane evaluates its
// bond once and adds its four basis values / derivatives into a per-wave row buffer in LDS (native ds_add_f64;
// bonds of one shell hit the same four columns, the LDS serialises those); the buffer then leaves as coalesced
// rows.  Columns of pair blocks the atom does not belong to stay zero in the buffer.
template <bool WANT_E, bool WANT_F, bool RL>           // RL: recs is the workgroup's LDS copy
__device__ __forceinline__ void pair_rows(const FeatArgs &A, const BasisDev *B, const KnotRec *recs, const WaveLds &w, int m,
                                          int sm, int n_cand, const ESink &es) {
    const int lane = lane_id(), S = load_const(&B->S);
    const int n2 = A.n_pair_cols;                       // pair columns are [S, S + n2)
    double *row = w.pstage;                             // [4][n2]: energy, fx, fy, fz
    const int pairs_uniform = load_const(&B->pairs_uniform), lead2 = load_const(&B->lead2), trail2 = load_const(&B->trail2);
    for (int q = lane; q < 4 * n2; q += WAVE) row[q] = 0.0;
    wave_sync();
    for (int e0 = 0; e0 < n_cand; e0 += WAVE) {
        const int e = e0 + lane;
        if (e < n_cand) {
            const double *c = w.cand + (size_t)e * CAND_STRIDE;
            const double d = c[3];
            // (one set of 2-body settings: knots, range and size of pair 0 through scalar loads, only the block's first column
            // per lane -- instead of two dependent vector loads per field)
            const int pair_idx = sm * UF3_MAX_SPECIES + (int)c[4];
            LegDev leg;
            double p_rmin, p_rmax;
            int p_nb;
            const int p_col = B->pair_col[pair_idx];
            if (pairs_uniform) {
                leg = load_const(&B->pairs[0].leg); p_rmin = load_const(&B->pairs[0].rmin); p_rmax = load_const(&B->pairs[0].rmax);
                p_nb = load_const(&B->pairs[0].nb);
            } else {
                const PairDev &pd = B->pairs[B->pair_of[pair_idx]];
                leg = pd.leg; p_rmin = pd.rmin; p_rmax = pd.rmax; p_nb = pd.nb;
            }
            if (!(d > p_rmin && d < p_rmax)) continue;                // a 3-body-only neighbour
            KnotRec kr;
            double v[4], dv[4];
            const int first = load_interval<RL ? 3 : 1>(recs, leg, d, kr) - 3;
            bspline4<WANT_F>(kr, d, v, dv);
            const double s = 2.0 / d;                   // both directed images of the bond (distances.py:116-141)
            const double dir[3] = {s * c[0], s * c[1], s * c[2]};
            const int hi = p_nb - trail2, base = p_col - S;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int bf = first + q;
                if (bf >= lead2 && bf < hi) {           // bspline.py:840,880
                    double *dst = row + base + bf;
                    if (WANT_E) lds_add(dst, v[q]);
                    if (WANT_F) { lds_add(dst + n2, dv[q] * dir[0]); lds_add(dst + 2 * n2, dv[q] * dir[1]); lds_add(dst + 3 * n2, dv[q] * dir[2]); }
                }
            }
        }
    }
    wave_sync();
    for (int col = lane; col < n2; col += WAVE) {
        if (WANT_F) {
            double *dst = A.x_f + (size_t)m * 3 * A.ld + S + col;
            __builtin_nontemporal_store(row[n2 + col], dst); __builtin_nontemporal_store(row[2 * n2 + col], dst + A.ld);
            __builtin_nontemporal_store(row[3 * n2 + col], dst + 2 * (size_t)A.ld);
        }
        if (WANT_E) es.add(S + col, row[col]);
    }
    wave_sync();
}

// 3-body neighbour list of atom m from the candidates MODE 0 already holds in LDS (same output as k_build_n3):
// images with r_min3 < d <= r_max3 (angles.py:340), sorted by (species, reference supercell index), padded to cap.
__device__ __forceinline__ void build_n3_list(const FeatArgs &A, const BasisDev *B, const FrameGeom &g, const WaveLds &w,
                                              int m, int n_cand) {
    const int lane = lane_id(), cap = A.n3.cap;
    unsigned long long *key = (unsigned long long *)w.pstage;      // the row buffer is free again
    int *src = (int *)(key + cap);
    const double rmin3 = load_const(&B->rmin3), rmax3 = load_const(&B->rmax3);
    int count = 0;
    for (int e0 = 0; e0 < n_cand; e0 += WAVE) {
        const int e = e0 + lane;
        const double *c = w.cand + (size_t)e * CAND_STRIDE;
        const bool ok = e < n_cand && c[3] > rmin3 && c[3] <= rmax3;
        const unsigned long long mask = __ballot(ok);
        if (ok) {
            const int slot = count + mbcnt(mask);
            if (slot < cap) {
                const int2 ps = *(const int2 *)(c + 5);
                int s0, s1, s2;
                unpack3(ps.y, s0, s1, s2);
                const int sidx = supercell_index(g, s0, s1, s2, ps.x - g.atom_lo);
                key[slot] = ((unsigned long long)(int)c[4] << 32) | (unsigned)sidx;
                src[slot] = e;
            }
        }
        count += __popcll(mask);
    }
    wave_sync();
    // (the batch's longest list: a look first -- after the first few atoms nearly every wave finds its count there already)
    if (A.n3_seen && lane == 0 && count > __builtin_nontemporal_load(A.n3_seen)) atomicMax(A.n3_seen, count);
    if (count > cap) { if (lane == 0) atomicMax(A.n3_need, count); count = cap; }
    if (lane == 0) A.n3.cnt[m] = count;
    if (count <= WAVE) {                 // entries are species-sorted: offsets per species = entries of a lower species (ballots)
        const int my_spec = lane < count ? (int)(key[lane] >> 32) : UF3_MAX_SPECIES;
        int below = count;
        for (int sp = 0; sp <= load_const(&B->S); sp++) {
            const int n_lower = __popcll(__ballot(my_spec < sp));
            if (lane == sp) below = n_lower;
        }
        if (lane <= UF3_MAX_SPECIES) A.n3.spoff[(size_t)m * (UF3_MAX_SPECIES + 1) + lane] = below;
    } else
        for (int sp = lane; sp <= UF3_MAX_SPECIES; sp += WAVE) {
            int below = 0;
            for (int f = 0; f < count; f++) below += (int)(key[f] >> 32) < sp;
            A.n3.spoff[(size_t)m * (UF3_MAX_SPECIES + 1) + sp] = below;
        }
    for (int e = lane; e < count; e += WAVE) {                      // rank sort
        const unsigned long long k = key[e];
        int rank = 0;
        for (int f = 0; f < count; f++) rank += key[f] < k;
        const double *c = w.cand + (size_t)src[e] * CAND_STRIDE;
        const int2 ps = *(const int2 *)(c + 5);
        N3Entry out;
        out.dx = c[0]; out.dy = c[1]; out.dz = c[2]; out.r = c[3];
        out.parent = ps.x; out.shiftc = ps.y; out.sidx = (int)(unsigned)k; out.spec = (int)(k >> 32);
        A.n3.ent[(size_t)m * cap + rank] = out;
    }
    wave_sync();
}

__device__ __forceinline__ void zero_rows(double *x_f, int m, int ld, int col, int n) {
    for (int c = lane_id(); c < n; c += WAVE) {
        double *dst = x_f + (size_t)m * 3 * ld + col + c;
        __builtin_nontemporal_store(0.0, dst); __builtin_nontemporal_store(0.0, dst + ld); __builtin_nontemporal_store(0.0, dst + 2 * (size_t)ld);
    }
}

// MODE selects the column blocks a launch is responsible for, so that each specialisation carries only the
// registers of its own path: 0 = one-body + pair blocks; 1..5 = trio blocks whose (nsrc, 64-column chunks
// per walk) is (1,1), (1,2), (2,1), (2,2), (6,1); 6..9 = trio blocks with a dense window accumulated on the matrix
// cores (trio_block_mfma) in (row tiles, column tiles) = (1,1), (1,2), (1,4), (2,6).  Every block is written by
// exactly one launch.  (6 and 7 are compiled for three waves per SIMD; whether three workgroups fit a CU is the
// launch's LDS footprint.)
__device__ __forceinline__ int trio_mode(const TrioDev *td) {
    if (td->dense) return td->dense;
    const bool wide = td->ncol > WAVE;
    return td->nsrc == 1 ? (wide ? 2 : 1) : (td->nsrc == 2 ? (wide ? 4 : 3) : 5);
}

// MODE_ 10 = MODE 7 for a basis whose mode-7 blocks all stage grouped windows (the launch then carries no code and no
// registers of the ordinary two-tile path); MODE_ 11 = MODE 9 with banded windows only, likewise.
template <bool WANT_E, bool WANT_F, bool RECS_LDS, int MODE_, bool IMG>
__global__ void __launch_bounds__(WPB * WAVE, MODE_ == 0 ? 5 : ((MODE_ == 6 || MODE_ == 7 || MODE_ == 10) ? 3 : 2))
k_featurize(FeatArgs A) {
    constexpr int MODE = MODE_ == 10 ? 7 : (MODE_ == 11 ? 9 : MODE_);
    constexpr bool GROUPED_ONLY = MODE_ == 10 || MODE_ == 11;      // (11: mode 9, banded windows with the band tiles 0, 1, 2 only)
    extern __shared__ __align__(16) unsigned char smem[];
    const BasisDev *B = A.B;
    const int F = load_const(&B->F), S = load_const(&B->S), n_trios = load_const(&B->T), cap = A.n3.cap;
    // a batch with atoms far outside their cell needs the reference's image-range rule in the 3-body force rows (TrioWalk::
    // img_check): the ordinary launches leave it to the IMG ones (the host sees the same flag and repeats the call)
    if (!IMG && WANT_F && MODE != 0 && A.outside && A.outside[0]) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // wave-uniform: LDS pointers and atom indices in SGPRs
    double *erow = (double *)smem;                                         // [F] shared by the block (WANT_E)
    const bool e_lds = WANT_E && !A.e_direct;
    // (the pair launch adds to the one-body and pair columns only: its LDS row ends there)
    const int FE = MODE == 0 ? S + A.n_pair_cols : F;
    const size_t e_d = e_lds ? (size_t)FE + (FE & 1) : 0;
    // LDS carve (must match feat_lds_bytes on the host).  MODE 0 (pairs): candidate list + pair records, pair
    // knot records only; trio modes: own neighbour list + triplet records, all knot records.
    const size_t cand_d = (size_t)A.cand_cap * CAND_STRIDE;
    const size_t pair_buf_d = max(4 * (size_t)A.n_pair_cols, (3 * (size_t)cap + 1) / 2 + 2);   // row buffer / sort keys
    constexpr bool DENSE = MODE >= 6;
    const size_t stage_d = MODE == 0 ? cand_d + pair_buf_d
                           : (DENSE ? (size_t)A.dense_stage : (size_t)NSTAGE * ITEM_STRIDE);
    const size_t list_d = MODE == 0 ? 0 : 5 * (size_t)cap + ((5 * cap) & 1);
    const size_t geo_d = DENSE ? (size_t)7 * GEO_N : 0;
    const size_t per_wave_d = list_d + stage_d + (stage_d & 1) + geo_d;
    const size_t per_wave_i = MODE == 0 ? 0 : 3 * (size_t)cap + 2 * ((size_t)cap + 1) + (UF3_MAX_SPECIES + 2) +
                                                  (size_t)cap * (S + 1);
    double *wd = erow + e_d + (size_t)wave * per_wave_d;
    int *wi = (int *)(erow + e_d + (size_t)WPB * per_wave_d) + (size_t)wave * per_wave_i;
    WaveLds w;
    w.ox = wd; w.oy = w.ox + cap; w.oz = w.oy + cap; w.orr = w.oz + cap; w.oir = w.orr + cap;
    w.stage = wd + list_d;
    w.geo = w.stage + stage_d + (stage_d & 1);
    w.cand = w.stage;
    w.pstage = w.stage + cand_d;
    w.oparent = wi; w.oshift = wi + cap; w.osidx = wi + 2 * cap;
    w.noff = wi + 3 * cap; w.nbase = w.noff + cap + 1; w.so = w.nbase + cap + 1;
    w.ospoff = w.so + (UF3_MAX_SPECIES + 2);
    w.sp_stride = S + 1;

    // knot-interval records (de Boor-Cox coefficients): staged in LDS when they fit
    KnotRec *recs_lds;
    {
        size_t ints_total = ((size_t)WPB * per_wave_i + 3) & ~(size_t)3;   // 16-B alignment
        recs_lds = (KnotRec *)((int *)(erow + e_d + (size_t)WPB * per_wave_d) + ints_total);
    }
    // (pair records come first: the pair launch copies those, a trio launch only the records its legs refer to)
    const int rec_first = MODE == 0 ? 0 : A.trio_rec_lo, rec_end = MODE == 0 ? A.n_pair_recs : A.n_recs;
    if (RECS_LDS) {
        const double *srcp = (const double *)(A.recs + rec_first);
        double *dstp = (double *)recs_lds;
        for (int q = tid; q < (rec_end - rec_first) * 12; q += WPB * WAVE) dstp[q] = srcp[q];
    }
    const KnotRec *recs = RECS_LDS ? recs_lds - rec_first : A.recs;
    int fragp[4] = {0, 0, 0, 0};
    const int *dsrc = A.dsrc;
    const unsigned short *gsrc = A.gsrc;
    if (DENSE) {
        for (int v = 0; v < 4; v++) fragp[v] = A.frag[(lane * 4 + v) * 2] * 16 + A.frag[(lane * 4 + v) * 2 + 1];
        int *dl = (int *)(recs_lds + (RECS_LDS ? rec_end - rec_first : 0));
        if (A.dsrc_lds) {
            for (int q = tid; q < A.n_dsrc; q += WPB * WAVE) dl[q] = A.dsrc[q];
            dsrc = dl;
            dl += A.n_dsrc;
        }
        if (A.gsrc_lds) {                                          // (n_gsrc is even)
            for (int q = tid; q < A.n_gsrc / 2; q += WPB * WAVE) dl[q] = ((const int *)A.gsrc)[q];
            gsrc = (const unsigned short *)dl;
        }
    }
    if (e_lds) { for (int q = tid; q < FE; q += WPB * WAVE) erow[q] = 0.0; }
    if (DENSE) for (int q = lane; q < (int)stage_d; q += WAVE) w.stage[q] = 0.0;   // masked operands read stale slots
    __syncthreads();
    // workgroups go to the 8 XCDs round-robin by their linear id, each XCD has its own L2: give every XCD one contiguous
    // eighth of the atoms (atoms are in cell order, a neighbour's list is then mostly in the same L2: +1 % on the clock).
    // The grid is a multiple of 8; surplus workgroups find no atoms.
    const int bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int block_first = bid * A.atoms_per_block;
    const int block_end = min(block_first + A.atoms_per_block, A.natoms);
    int erow_frame = -1;
    GroupedLayout GL;
    GL.id = -1;
    FrameGeom g;                                                 // geometry of the frame the wave's current atom belongs to
    int g_frame = -1;
    for (int m0 = block_first; m0 < block_end; m0 += WPB) {      // the block's waves take consecutive atoms
        const int m = m0 + wave;
        const bool active = m < block_end;
        if (e_lds) {
            int f_first = load_const(A.frame_of + m0);
            if (f_first != erow_frame) {                          // block-uniform
                __syncthreads();
                if (erow_frame >= 0)
                    for (int q = tid; q < FE; q += WPB * WAVE) {
                        double v = erow[q];
                        if (v != 0.0) { unsafeAtomicAdd(A.x_e + (size_t)erow_frame * F + q, v); erow[q] = 0.0; }
                    }
                __syncthreads();
                erow_frame = f_first;
            }
        }
        if (!active) continue;
        const int fr = load_const(A.frame_of + m);
        if (fr != g_frame) { g = A.geoms[fr]; g_frame = fr; }         // (wave-uniform; consecutive atoms mostly share their frame)
        const int sm = ((const __attribute__((address_space(4))) signed char *)(unsigned long long)A.spec)[m];
        const double pm[3] = {A.pos[3 * (size_t)m], A.pos[3 * (size_t)m + 1], A.pos[3 * (size_t)m + 2]};
        ESink es;
        es.lds = erow; es.glob = WANT_E ? A.x_e + (size_t)fr * F : nullptr; es.direct = !e_lds || (fr != erow_frame);
        // ---- 1-body columns ------------------------------------------------------------------
        if (MODE == 0 && WANT_F) zero_rows(A.x_f, m, A.ld, 0, S);
        if (MODE == 0 && WANT_E && lane == 0) es.add(sm, 1.0);
        // ---- 2-body: neighbour images -> LDS once, then one pass per pair block ------------------
        if (MODE == 0) {
            int n_cand = 0;
            const bool build3 = A.build_n3 != 0;
            // (the range tests on the SQUARED distance, against thresholds that make them the same decisions as on its correctly
            // rounded root -- PairDev::s_lo: the root itself, 20 instructions, is taken once for the candidates kept, below)
            const double s3_lo = load_const(&B->s3_lo), s3_hi = load_const(&B->s3_hi);
            const int pairs_uniform0 = load_const(&B->pairs_uniform);
            const double2 rr0 = double2{load_const(&B->pairs[0].s_lo), load_const(&B->pairs[0].s_hi)};
            if (!UF3_SKIP(1)) for_each_candidate(g, A.cl, m, [&](bool ok, const SlotRec &sr, int sj, int s0, int s1, int s2) {
                double dx = 0, dy = 0, dz = 0, d = 0;
                if (ok) {
                    image_delta(g, sr, s0, s1, s2, pm, dx, dy, dz);
                    d = norm3_sq_rn(dx, dy, dz);
                    // kept if inside its pair's range (distances.py:66, strict both sides) or a 3-body neighbour;
                    // (s_lo, s_hi) in one load, no short-circuit: every clause evaluated would be a memory round trip
                    const double2 rr = pairs_uniform0 ? rr0 : *(const double2 *)&B->pairs[B->pair_of[sm * UF3_MAX_SPECIES + sj]].s_lo;
                    ok = ((d > rr.x) & (d < rr.y)) | (build3 & (d > s3_lo) & (d <= s3_hi));
                }
                unsigned long long mask = __ballot(ok);
                if (ok) {
                    int e = n_cand + mbcnt(mask);
                    if (e < A.cand_cap) {
                        double *c = w.cand + (size_t)e * CAND_STRIDE;
                        *(double2 *)(c) = double2{dx, dy};
                        *(double2 *)(c + 2) = double2{dz, d};
                        c[4] = (double)sj;
                        *(int2 *)(c + 5) = make_int2(sr.atom, pack3(s0, s1, s2));
                    }
                }
                n_cand += __popcll(mask);
            });
            if (n_cand > A.cand_cap) { if (lane == 0) atomicMax(A.cand_need, n_cand); n_cand = A.cand_cap; }
            wave_sync();
            for (int e = lane; e < n_cand; e += WAVE) {                       // squared distance -> distance
                double *c = w.cand + (size_t)e * CAND_STRIDE + 3;
                *c = sqrt(*c);
            }
            wave_sync();
            if (!UF3_SKIP(512)) pair_rows<WANT_E, WANT_F, RECS_LDS>(A, B, recs, w, m, sm, n_cand, es);
            if (A.build_n3 && !UF3_SKIP(1024)) build_n3_list(A, B, g, w, m, n_cand);
        }
        // ---- 3-body ---------------------------------------------------------------------------
        if (MODE != 0 && n_trios > 0) {
            PhaseClock pcl;
            const int n = load_const(A.n3.cnt + m);
            size_t base = (size_t)m * cap;
            wave_sync();
            for (int e = lane; e < n; e += WAVE) {
                const N3Entry en = A.n3.ent[base + e];
                w.ox[e] = en.dx; w.oy[e] = en.dy; w.oz[e] = en.dz; w.orr[e] = en.r; w.oir[e] = 1.0 / en.r;
                w.oparent[e] = en.parent; w.oshift[e] = en.shiftc; w.osidx[e] = en.sidx;
            }
            if (lane <= S) w.so[lane] = A.n3.spoff[(size_t)m * (UF3_MAX_SPECIES + 1) + lane];
            wave_sync();
            if (WANT_F)                                            // neighbour role: the lists of m's neighbours
                for (int q = lane; q < n * (S + 1); q += WAVE) {
                    const int e = q / (S + 1), sp = q - e * (S + 1);
                    w.ospoff[e * (S + 1) + sp] = A.n3.spoff[(size_t)w.oparent[e] * (UF3_MAX_SPECIES + 1) + sp];
                }
            wave_sync();
            pcl.lap(0);
            for (int t = 0; t < n_trios; t++) {
                const TrioDev *td = A.trios + t;
                // one 32-byte scalar load (see TrioHead; as eight ints the compiler sinks every field's load to its branch)
                typedef int int8_v __attribute__((ext_vector_type(8)));
                const int8_v hv = *(const __attribute__((address_space(4))) int8_v *)(unsigned long long)&td->head;
                const TrioHead th = {hv[0], hv[1], hv[2], hv[3], hv[4], hv[5], hv[6], hv[7]};
                const int t_dense = th.dense, t_nsrc = th.nsrc, t_ncol = th.ncol;
                const int t_mode = t_dense ? t_dense : (t_nsrc == 1 ? (t_ncol > WAVE ? 2 : 1) : (t_nsrc == 2 ? (t_ncol > WAVE ? 4 : 3) : 5));
                if (t_mode != MODE) continue;
                const int t_sc = th.sc, t_sa = th.sa, t_sb = th.sb;
                const bool touches = (t_sc == sm) || (WANT_F && (t_sa == sm || t_sb == sm));
                if (!touches) { if (WANT_F && !UF3_SKIP(32)) zero_rows(A.x_f, m, A.ld, th.col, t_ncol); continue; }
                if (MODE == 1) trio_block<WANT_E, WANT_F, 1, 1, IMG>(A, B, recs, g, w, m, sm, t, es);
                else if (MODE == 2) trio_block<WANT_E, WANT_F, 1, 2, IMG>(A, B, recs, g, w, m, sm, t, es);
                else if (MODE == 3) trio_block<WANT_E, WANT_F, 2, 1, IMG>(A, B, recs, g, w, m, sm, t, es);
                else if (MODE == 4) trio_block<WANT_E, WANT_F, 2, 2, IMG>(A, B, recs, g, w, m, sm, t, es);
                else if (MODE == 5) trio_block<WANT_E, WANT_F, 6, 1, IMG>(A, B, recs, g, w, m, sm, t, es);
                else if (MODE == 7 && WANT_F && th.grouped) {
                    if (__builtin_expect((th.grouped & 0xff) - 1 != GL.id, 0)) grouped_layout_setup<WANT_E>(A, t, fragp, GL);
                    trio_block_grouped<WANT_E, IMG, RECS_LDS>(A, B, recs, g, w, m, sm, th, es, GL, gsrc);
                }
                else if (MODE == 9 && WANT_F && th.grouped)
                    // (TrioHead::grouped of a banded trio: 1 | its bands' first tiles << 8, 4 bits each)
                    // MODE_ 11: the host has checked that every banded trio of the basis has the tiles 0, 1, 2
                    trio_block_banded<WANT_E, IMG, RECS_LDS, MODE_ == 11>(A, B, recs, g, w, m, sm, t, es, fragp, dsrc);
                else if (!GROUPED_ONLY)
                    trio_block_mfma<WANT_E, WANT_F, (MODE >= 6 ? MODE : 6), IMG, RECS_LDS>(A, B, recs, g, w, m, sm, t, es, fragp, dsrc);
            }
        }
    }
    if (e_lds) {
        __syncthreads();
        if (erow_frame >= 0)
            for (int q = tid; q < FE; q += WPB * WAVE) {
                double v = erow[q];
                if (v != 0.0) unsafeAtomicAdd(A.x_e + (size_t)erow_frame * F + q, v);
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
    double *virial;               // [natoms][6] dE/d(strain) shares (xx,yy,zz,yz,xz,xy) or null
    double *nbr_f;                // [natoms * cap][3] force each centre's triplets put on its list entries (two-pass route)
    int *n3_need;                 // fused list build: longest list seen when one overflowed `cap`
    int fuse_n3;                  // !GATHER: build the atom's 3-body list from the candidates of the pair walk
    int natoms;
    int atom_lo;                  // first atom of this launch (blocks cover [atom_lo, natoms_end))
    int atom_hi;
    // collection pass of a block of CENTRES (uf3_eval_centres): only the centres [atom_lo, atom_hi) have run, and only the
    // atoms inside the block or marked as its halo have lists
    const int *halo_mark;         // [natoms] != 0: a halo atom of the block; null: every atom collects from every centre
    // MD route (k_eval<..., MD = true>): the candidates come from the context's persistent superset lists instead of a cell-list walk
    const SupEntry *sup_ent;      // [natoms][sup_cap] neighbours within r_search + skin at build time, sorted by (species, supercell index)
    const int *sup_cnt;           // [natoms]
    int sup_cap;
    const double *pos_ref;        // [natoms][3] positions the lists were built from
    const int32_t *z_now;         // [natoms] atomic numbers of THIS call (compared with the species the lists were built for)
    double md_hard2, md_soft2;    // squared displacement limits: (skin / 2)^2 -- beyond it the lists may miss a neighbour -- and the early warning
    double *md_inbox;             // [natoms][sup_cap][4]: force the triplets centred on a neighbour put on this atom | the step's stamp, at the
                                  // atom's own list position of that neighbour (written by the neighbour's centre pass)
    int *md_surv;                 // [natoms][n3.cap] list position (in the superset list) of each entry of the step's 3-body list
    double md_stamp;              // this launch's stamp: inbox entries with another one are left over from earlier steps
    int *md_mark;                 // a block of centres (uf3_eval_centres on the MD route): [natoms] number of the last launch whose centres
    int md_mark_now;              // wrote into the atom's inbox; null: whole batch
    int *md_flags;                // [0] = 1: some atom moved past the hard limit (results invalid, rebuild and repeat); [1] = 1: past the soft one
    // CW instances (k_eval<..., CW = true>, round 6): the trios' coefficients inside the kept-bin window of the centre legs,
    // [T][ext_l][ext_m][dim_n] doubles followed by a run of zeros, copied into the workgroup's LDS (uf3_eval uploads the table when
    // every coefficient outside the window is exactly zero -- the trimmed bins of a fitted model)
    const double *c3w;            // device table, cw_bytes long (a multiple of 1024)
    int cw_bytes;                 // table + zero run, bytes
    int cw_zero;                  // byte offset of the zero run (long enough for any row offset + 32)
    int cw_lo, cw_ext;            // first kept bin and number of kept bins of the centre legs (legs l and m alike)
    int lds_per_wave;             // bytes of the per-wave arrays of a multi-wave workgroup
    // per-workgroup sums of the collection pass (one whole frame on the MD route): [n_wg] energies | [n_wg][6] strain derivatives |
    // int64 {0, n_wg}: k_frame_sum then adds n_wg = natoms / 16 partial sums instead of natoms per-atom values (12 -> 4 us at 50 k atoms)
    double *part_e, *part_v;
    long long *part_off;
    int cw_recs_bytes;            // all knot records of the basis (BasisDev::recs), copied behind the table: the pair splines, the
    int cw_c2;                    // per-bond leg tables and leg n read them from LDS; then the pair coefficients c2 (cw_c2 doubles)
};

#ifndef EVAL_CGROUP
#define EVAL_CGROUP 4     // coefficient rows (32 bytes each) in flight per lane (8 / 16 need a fourth of the registers more:
                          // 2 waves per SIMD, 0.58 against 0.50 ms on the 50 k-atom ternary frame)
#endif
// V and its three leg partials at (rl, rm, rn) from the full coefficient grid of a trio
__device__ __forceinline__ bool trio_value(const BasisDev *B, const double *c3, int trio, double rl, double rm, double rn,
                                           bool want_grad, double &val, double *grad) {
    if (trio < 0) return false;
    // (the per-lane descriptor through the global address space explicitly: B->trios is a pointer read from memory, through it
    // the loads would be flat loads)
    typedef const __attribute__((address_space(1))) TrioDev *GlobalTrio;
    GlobalTrio td = (GlobalTrio)(load_const(&B->trios) + trio);
    const KnotRec *recs = load_const(&B->recs);
    // all descriptor fields in flight together (per-lane trio: vector loads), then a branch-free range test.  When every trio
    // has the same legs and grid (one set of 3-body settings: the usual case) those come from trio 0 through scalar loads and
    // only the offset of the trio's coefficient grid is per lane.
    LegDev l0, l1, l2;
    int dim_m, dim_n;
    const int lut_off = td->lut_off;
    if (load_const(&B->trio_legs_uniform)) {
        const TrioDev *t0 = load_const(&B->trios);
        l0 = load_const(&t0->leg[0]); l1 = load_const(&t0->leg[1]); l2 = load_const(&t0->leg[2]);
        dim_m = load_const(&t0->dim_m); dim_n = load_const(&t0->dim_n);
    } else {
        auto leg_of = [&](int q) {
            LegDev l;
            l.rec_off = td->leg[q].rec_off; l.nk = td->leg[q].nk; l.t0 = td->leg[q].t0; l.tlast = td->leg[q].tlast; l.inv_h = td->leg[q].inv_h;
            return l;
        };
        l0 = leg_of(0); l1 = leg_of(1); l2 = leg_of(2);
        dim_m = td->dim_m; dim_n = td->dim_n;
    }
    if (!((rl > l0.t0) & (rl < l0.tlast) & (rm > l1.t0) & (rm < l1.tlast) & (rn > l2.t0) & (rn < l2.tlast))) return false;
    // Memory round trips, not arithmetic, bound this function (the compiler had serialised it into ~20 dependent loads):
    // the three legs' knot records go out together, then -- the coefficient block only needs the interval indices -- the
    // coefficient rows, EVAL_CGROUP at a time, ahead of the arithmetic that consumes them.
    KnotRec kl, km, kn;
    int il = load_interval_guess<1>(recs, l0, rl, kl), im = load_interval_guess<1>(recs, l1, rm, km), in = load_interval_guess<1>(recs, l2, rn, kn);
    il = load_interval_fix<1>(recs, l0, rl, il, kl); im = load_interval_fix<1>(recs, l1, rm, im, km); in = load_interval_fix<1>(recs, l2, rn, in, kn);
    int mn = dim_m * dim_n;
#ifdef UF3_ABLATE_EVAL_COEF
    const double *c = c3 + lut_off + ((il - 3) * mn + (im - 3) * dim_n + (in - 3)) * 0;      // (experiment: every lane the same rows)
#else
    const double *c = c3 + lut_off + (il - 3) * mn + (im - 3) * dim_n + (in - 3);
#endif
    typedef double coeff4 __attribute__((ext_vector_type(4), aligned(8)));
    coeff4 cc[EVAL_CGROUP];
#pragma unroll
    for (int q = 0; q < EVAL_CGROUP; q++) cc[q] = *(const coeff4 *)(c + (q >> 2) * mn + (q & 3) * dim_n);   // four consecutive n bins per request
    asm volatile("" ::: "memory");
    double vl[4], vm[4], vn[4], dl[4], dm[4], dn[4];
    bspline4<true>(kl, rl, vl, dl);
    bspline4<true>(km, rm, vm, dm);
    bspline4<true>(kn, rn, vn, dn);
    // contracted innermost leg first: s(a, b) = sum_n c v_n, then over b (with v_m and d_m), then over a -- 12 operations per
    // (a, b) instead of 15
    double v = 0, g0 = 0, g1 = 0, g2 = 0;
    double sa = 0, sda = 0, sma = 0;
#pragma unroll
    for (int q0 = 0; q0 < 16; q0 += EVAL_CGROUP) {
        if (q0 > 0) {
#pragma unroll
            for (int q = 0; q < EVAL_CGROUP; q++) cc[q] = *(const coeff4 *)(c + ((q0 + q) >> 2) * mn + ((q0 + q) & 3) * dim_n);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int q = 0; q < EVAL_CGROUP; q++) {
            const int a = (q0 + q) >> 2, b = (q0 + q) & 3;
            const double s = cc[q][0] * vn[0] + cc[q][1] * vn[1] + cc[q][2] * vn[2] + cc[q][3] * vn[3];
            if (b == 0) { sa = vm[0] * s; } else sa += vm[b] * s;
            if (want_grad) {
                const double sd = cc[q][0] * dn[0] + cc[q][1] * dn[1] + cc[q][2] * dn[2] + cc[q][3] * dn[3];
                if (b == 0) { sda = vm[0] * sd; sma = dm[0] * s; } else { sda += vm[b] * sd; sma += dm[b] * s; }
            }
            if (b == 3) {
                v += vl[a] * sa;
                if (want_grad) { g0 += dl[a] * sa; g1 += vl[a] * sma; g2 += vl[a] * sda; }
            }
        }
    }
    val = v; grad[0] = g0; grad[1] = g1; grad[2] = g2;
    return true;
}

// The same value and partials for the centre pass of k_eval when every trio has the legs of trio 0 (the usual case): the two
// CENTRE legs are bonds of the atom's list, evaluated once per bond into an LDS table (interval | four values | four
// derivatives) instead of once per triplet -- 14 evaluations instead of 182 per atom, and ONE knot record live per lane instead
// of three (72 of the kernel's registers) --, leg n at the guessed interval together with the first coefficient rows, the rows
// after them two groups ahead of the arithmetic: two dependent memory round trips per triplet instead of five.  Same knot
// records, same de Boor triangle, same order of the contraction as trio_value: the same bits.
// tl / tm: [8] values | derivatives of legs l / m; il / im their intervals; lut_off from the wave's table
template <int AS>
__device__ __forceinline__ bool trio_value_tab(const KnotRec *recs, const double *c3, int lut_off, const LegDev &l2, int dim_m, int dim_n,
                                               int il, int im, const double *tl, const double *tm, double rn, bool want_grad,
                                               double &val, double *grad) {
    typedef double coeff4 __attribute__((ext_vector_type(4), aligned(8)));
    KnotRec kn;
#ifdef UF3_ABL_KN
    int in_g = 3 + (int)((rn - l2.t0) * l2.inv_h);
    in_g = in_g < 3 ? 3 : (in_g > l2.nk - 5 ? l2.nk - 5 : in_g);
    for (int u = 0; u < 6; u++) { kn.t[u] = l2.t0 + (in_g - 5 + u) * 0.4; kn.r[u] = 2.5 / (1 + (u > 0) + (u > 2)); }
#else
    const int in_g = load_interval_guess<AS>(recs, l2, rn, kn);
#endif
    const int mn = dim_m * dim_n;
    // (rows through ONE per-lane 32-bit byte offset on top of the uniform grid pointer: as 64-bit per-lane pointers the row
    // addresses, hoisted by the compiler, cost two registers each)
    const unsigned off0 = 8u * (unsigned)(lut_off + (il - 3) * mn + (im - 3) * dim_n - 3);
    unsigned off = off0 + 8u * (unsigned)in_g;
    auto row = [&](int a, int q) { return *(const coeff4 *)((const char *)c3 + (off + 8u * (unsigned)(a * mn + q * dim_n))); };
    coeff4 cc[EVAL_CGROUP];
#pragma unroll
    for (int q = 0; q < EVAL_CGROUP; q++) cc[q] = row(q >> 2, q & 3);
    asm volatile("" ::: "memory");
#ifdef UF3_ABL_KN
    const int in = in_g;
#else
    const int in = load_interval_fix<AS>(recs, l2, rn, in_g, kn);
#endif
    if (__builtin_expect(in != in_g, 0)) {             // (non-uniform knots, or rn on an interval boundary)
        off = off0 + 8u * (unsigned)in;
#pragma unroll
        for (int q = 0; q < EVAL_CGROUP; q++) cc[q] = row(q >> 2, q & 3);
        asm volatile("" ::: "memory");
    }
    double vn[4], dn[4];
    bspline4<true>(kn, rn, vn, dn);
    double v = 0, g0 = 0, g1 = 0, g2 = 0;
    double sa = 0, sda = 0, sma = 0;
#pragma unroll
    for (int q0 = 0; q0 < 16; q0 += EVAL_CGROUP) {
#ifdef UF3_ABL_HALF
        if (q0 == 4) {
#else
        if (q0 > 0) {
#endif
#pragma unroll
            for (int q = 0; q < EVAL_CGROUP; q++) cc[q] = row((q0 + q) >> 2, (q0 + q) & 3);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int q = 0; q < EVAL_CGROUP; q++) {
            const int a = (q0 + q) >> 2, b = (q0 + q) & 3;
            const double s = cc[q][0] * vn[0] + cc[q][1] * vn[1] + cc[q][2] * vn[2] + cc[q][3] * vn[3];
            const double vmb = tm[b];
            if (b == 0) { sa = vmb * s; } else sa += vmb * s;
            if (want_grad) {
                const double sd = cc[q][0] * dn[0] + cc[q][1] * dn[1] + cc[q][2] * dn[2] + cc[q][3] * dn[3];
                const double dmb = tm[4 + b];
                if (b == 0) { sda = vmb * sd; sma = dmb * s; } else { sda += vmb * sd; sma += dmb * s; }
            }
            if (b == 3) {
                const double vla = tl[a];
                v += vla * sa;
                if (want_grad) { g0 += tl[4 + a] * sa; g1 += vla * sma; g2 += vla * sda; }
            }
        }
    }
    val = v; grad[0] = g0; grad[1] = g1; grad[2] = g2;
    return true;
}

// trio_value_tab over the kept-bin WINDOW of the centre legs, coefficient rows out of the workgroup's LDS (k_eval<..., CW>).  The
// table holds, per trio, the rows of the window over ALL n bins -- [EXT][EXT][dim_n]; every coefficient outside it is exactly zero
// (checked when the model was uploaded) -- and the per-bond leg tables are dense over the window rows (value | derivative of window
// row w, zero where the bond's four functions do not reach): the contraction runs over EXT x EXT rows instead of the 4 x 4 of the
// interval (9 instead of 16 at the default trims), every row at a constant offset from the lane's (trio, n interval) base -- no
// validity logic.  The terms left out are products with exact zeros and the others keep their order: the same bits as trio_value_tab.
//   cw: the table in LDS; t_base: byte offset of the trio's block; tl / tm: [2 * EXT] window values | derivatives of legs l / m
template <int EXT, int AS = 3>          // AS: where the table lives -- 3: the workgroup's LDS (CW instances), 1: global memory (WIN instances)
__device__ __forceinline__ bool trio_value_tab_cw(const KnotRec *recs, const unsigned char *cw, int t_base, const LegDev &l2, int dim_n,
                                                  const double *tl, const double *tm, double rn, bool want_grad, double &val, double *grad) {
    // (rows start at any n bin: 8-byte alignment -- ds_read2_b64; a 16-byte read at an odd double took ~60 LDS cycles per wave)
    typedef double coeff2 __attribute__((ext_vector_type(2), aligned(8)));
    typedef const __attribute__((address_space(AS))) coeff2 *LdsPairs;
    KnotRec kn;
    const int in = load_interval<3>(recs, l2, rn, kn);
    const int row_b = 8 * dim_n;                                       // bytes between consecutive rows of leg m; EXT of them per row of leg l
    const unsigned char *base = cw + (t_base + 8 * (in - 3));
    double vn[4], dn[4];
    bspline4<true>(kn, rn, vn, dn);
    double v = 0, g0 = 0, g1 = 0, g2 = 0;
    double sa = 0, sda = 0, sma = 0;
#pragma unroll
    for (int a = 0; a < EXT; a++) {
        coeff2 c01[EXT], c23[EXT];
#pragma unroll
        for (int b = 0; b < EXT; b++) {
            LdsPairs q = (LdsPairs)(const coeff2 *)(base + (a * EXT + b) * row_b);
            c01[b] = q[0]; c23[b] = q[1];
        }
#pragma unroll
        for (int b = 0; b < EXT; b++) {
            const double s = c01[b].x * vn[0] + c01[b].y * vn[1] + c23[b].x * vn[2] + c23[b].y * vn[3];
            const double vmb = tm[b];
            if (b == 0) { sa = vmb * s; } else sa += vmb * s;
            if (want_grad) {
                const double sd = c01[b].x * dn[0] + c01[b].y * dn[1] + c23[b].x * dn[2] + c23[b].y * dn[3];
                const double dmb = tm[EXT + b];
                if (b == 0) { sda = vmb * sd; sma = dmb * s; } else { sda += vmb * sd; sma += dmb * s; }
            }
        }
        const double vla = tl[a];
        v += vla * sa;
        if (want_grad) { g0 += tl[EXT + a] * sa; g1 += vla * sma; g2 += vla * sda; }
    }
    val = v; grad[0] = g0; grad[1] = g1; grad[2] = g2;
    return true;
}

// Wave-wide sum through DPP (round 6): __shfl_xor on a double is two ds_bpermute_b32 through the LDS crossbar per step -- twelve LDS
// instructions per sum, ten sums at the end of every k_eval wave -- where the data-parallel primitives move registers directly.
// quad_perm [1,0,3,2] and [2,3,0,1], row_half_mirror, row_mirror: every lane of a row of 16 holds the row's sum; row_bcast:15 into
// rows 1 and 3, row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum, returned to every lane.  A fixed tree: deterministic
// (tools/experiments/dpp_sum_test.hip checks it against a serial sum).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int l2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int h2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(h2, l2);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_take<0xB1, 0xf>(v);
    v += dpp_take<0x4E, 0xf>(v);
    v += dpp_take<0x141, 0xf>(v);
    v += dpp_take<0x140, 0xf>(v);
    v += dpp_take<0x142, 0xa>(v);
    v += dpp_take<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// the sums of three values over each aligned group of 16 lanes (a DPP row), in every lane of the group
__device__ __forceinline__ void row16_sum3(double &a, double &b, double &c) {
    a += dpp_take<0xB1, 0xf>(a); b += dpp_take<0xB1, 0xf>(b); c += dpp_take<0xB1, 0xf>(c);
    a += dpp_take<0x4E, 0xf>(a); b += dpp_take<0x4E, 0xf>(b); c += dpp_take<0x4E, 0xf>(c);
    a += dpp_take<0x141, 0xf>(a); b += dpp_take<0x141, 0xf>(b); c += dpp_take<0x141, 0xf>(c);
    a += dpp_take<0x140, 0xf>(a); b += dpp_take<0x140, 0xf>(b); c += dpp_take<0x140, 0xf>(c);
}

#define EVAL_Q 5          // doubles per queued bond of the evaluator
#define EVAL_TAB_KN 24    // most knot intervals of leg n whose records the TAB instances copy into LDS
#define EVAL_TAB_CAP 36   // longest list whose per-bond leg tables (2 x 64 B + 2 x 4 B per entry) fit over the queue of the pair walk
// GATHER: every atom also walks the triplets it belongs to as a neighbour (each triplet evaluated at its three atoms;
// what a block of atoms of a decomposed frame needs).  !GATHER: each triplet once, at its centre, which also sums the
// force it puts on each of its list entries (nbr_f, in LDS first); k_eval_collect then adds to every atom what its
// neighbours' triplets put on it.  One wave owns a centre and LDS adds of a wave keep their order: deterministic.
// (3 waves per SIMD; bounding it to 4 or 5 spills: 0.51 -> 0.60 / 1.19 ms on the 50 k-atom ternary frame)
#ifndef EVAL_MINW
#define EVAL_MINW 3
#endif
// (CAP: the list capacity as a compile-time constant -- 16, what a tuned context settles on for bcc / fcc cells -- or 0 for
// the launch's run-time value: with it every LDS array of the one-wave workgroup sits at a constant address, in the LDS
// instructions' immediate fields instead of scalar registers, of which the kernel spills 73)
// (MD: the candidates of the pair walk come from the persistent superset lists, see k_build_sup -- no cell list, no sort; only
// with !GATHER.  The survivors are taken in list order, which is the order the fused build below sorts into.)
// (TAB: the centre legs of the triplets from per-bond tables, trio_value_tab -- chosen by the host when every trio has the legs of
// trio 0, T <= 64 and the list capacity is <= EVAL_TAB_CAP; one knot record live per lane instead of three: 109 registers,
// four waves per SIMD.  A template parameter because the two triplet paths in ONE kernel cost the registers of the larger.)
#ifndef EVAL_TAB_MINW
#define EVAL_TAB_MINW 4
#endif
// (CW, round 6 -- VERDICT round 5 item 3: the coefficient rows of the triplets out of LDS.  A workgroup is EVAL_CW_WAVES one-atom
// waves instead of one; its waves copy the window table of ALL trios (EvalArgs::c3w: 19 KB for the ternary notebook basis) with
// global_load_lds, every knot record of the basis and the pair coefficients into LDS once and meet at ONE barrier behind the first
// batch of the list filter; everything else stays per wave.  The forces a centre's triplets put on its list entries are gathered from
// a per-trip stage instead of added atomically, the triplets' strain derivative comes from the gathered forces.  MD route, TAB
// instances, lists of at most 16 entries -- uf3_eval decides.)
#ifndef EVAL_CW_WAVES
#define EVAL_CW_WAVES 8
#endif
#define EVAL_CW_EXT 3       // kept bins of the centre legs the CW instances are compiled for (the default trims: 3 of 9)
// (the per-atom syncs of the body: a workgroup barrier in the one-wave workgroups, a wave barrier in the multi-wave ones, whose
// waves run through the body independently)
#define EVAL_SYNC() do { if (CW) wave_sync(); else __syncthreads(); } while (0)
// (WIN: the window-row contraction -- 3 x 3 coefficient rows per triplet out of the window table EvalArgs::c3w, per-bond leg tables dense
// over the window rows -- without the LDS copies: the table is read through global memory by the one-wave workgroups of the TAB
// instances that CW does not serve: the rebuild-everything route, longer lists, blocks of centres.  CW implies WIN.)
template <bool GATHER, bool VIR, int CAP = 0, bool MD = false, bool TAB = false, bool CW = false, bool WIN = CW>
__global__ void __launch_bounds__(CW ? 64 * EVAL_CW_WAVES : 64, TAB ? EVAL_TAB_MINW : EVAL_MINW)
k_eval(EvalArgs A) {
    static_assert(!CW || (TAB && !GATHER), "the LDS tables belong to the TAB centre pass");
    static_assert(!WIN || (TAB && !GATHER), "the window table belongs to the TAB centre pass");
    static_assert(!CW || WIN, "the LDS table is the window table");
    extern __shared__ __align__(16) unsigned char smem_all[];
    const BasisDev *B = A.B;
    const int cap = CAP > 0 ? CAP : A.n3.cap;
    const KnotRec *recs_g = load_const(&B->recs);
    const int S = load_const(&B->S);
    const int wave = CW ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    // (CW: [window table | zero run][every knot record of the basis][pair coefficients] shared, then the waves' own arrays)
    const size_t cw_shared = CW ? (size_t)A.cw_bytes + A.cw_recs_bytes + (((size_t)A.cw_c2 * 8 + 15) & ~(size_t)15) : 0;
    unsigned char *smem = smem_all + (CW ? cw_shared + (size_t)wave * A.lds_per_wave : 0);
    double *ox = (double *)smem, *oy = ox + cap, *oz = oy + cap, *orr = oz + cap;
    int *oparent = (int *)(orr + cap), *oshift = oparent + cap, *osidx = oshift + cap, *ospec = osidx + cap,
        *ooff = ospec + cap;
    double *queue = (double *)(ooff + cap + 1 + ((cap + 1) & 1));          // [128][EVAL_Q]: dx, dy, dz, d, species
    double *gx = queue + 2 * WAVE * EVAL_Q, *gy = gx + cap, *gz = gy + cap;  // !GATHER: force on the list entries
    // fused list build: entries in walk order (sorted into ox.. afterwards) and their (species, supercell index) keys
    double *ux = gz + cap, *uy = ux + cap, *uz = uy + cap, *ur = uz + cap;
    unsigned long long *ukey = (unsigned long long *)(ur + cap);
    int *uparent = (int *)(ukey + cap), *ushift = uparent + cap;
    const bool fuse = !GATHER && !MD && A.fuse_n3;
    int count3 = 0;
    // (one contiguous eighth of the atoms per XCD, as in k_featurize; the grid is a multiple of 8)
    int m = A.atom_lo + (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * (CW ? EVAL_CW_WAVES : 1) + wave;
    if (CW) {
        // the workgroup's tables: 1 KB per wave-instruction straight into LDS (no registers); landed by the barrier below
        const int n_chunks = A.cw_bytes >> 10;
        for (int ch = wave; ch < n_chunks; ch += EVAL_CW_WAVES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)A.c3w + ((size_t)ch << 10) + (threadIdx.x & 63) * 16),
                                             (__attribute__((address_space(3))) void *)(smem_all + ((size_t)ch << 10)), 16, 0, 0);
        const int4 *src = (const int4 *)recs_g;
        int4 *dst = (int4 *)(smem_all + A.cw_bytes);
        for (int q = threadIdx.x; q < (A.cw_recs_bytes >> 4); q += 64 * EVAL_CW_WAVES) dst[q] = src[q];
        double *c2d = (double *)(smem_all + A.cw_bytes + A.cw_recs_bytes);
        for (int q = threadIdx.x; q < A.cw_c2; q += 64 * EVAL_CW_WAVES) c2d[q] = A.c2[q];
    }
    // (CW: the knot records and pair coefficients the splines below read -- the workgroup's LDS copies once the barrier is passed)
    const KnotRec *recs_l = CW ? (const KnotRec *)(smem_all + A.cw_bytes) : recs_g;
    const double *c2_l = CW ? (const double *)(smem_all + A.cw_bytes + A.cw_recs_bytes) : A.c2;
    if (m >= A.atom_hi) {
        if (CW) __syncthreads();              // (the workgroup's one barrier: every wave arrives once)
        return;
    }
    int lane = lane_id();
    // (MD route: what belongs to the atom is the same for all lanes -- through scalar loads, the frame's cell included: vector
    // loads of them were a chain of dependent round trips in front of the list filter, and the cell sat in 18 vector registers)
    FrameGeom g;
    if (MD) {
        const int fidx = load_const(A.frame_of + m);
        for (int k = 0; k < 9; k++) g.cell[k] = load_const(&A.geoms[fidx].cell[k]);
    } else
        g = A.geoms[A.frame_of[m]];
    const int sm = ((const __attribute__((address_space(4))) signed char *)(unsigned long long)A.spec)[m];
    const bool want_f = A.forces != nullptr, want_v = VIR && A.virial != nullptr;   // (VIR: compiled out of the usual launches)
    double vir[6] = {0, 0, 0, 0, 0, 0};
    double pm[3];
    if (MD) { pm[0] = load_const(A.pos + 3 * (size_t)m); pm[1] = load_const(A.pos + 3 * (size_t)m + 1); pm[2] = load_const(A.pos + 3 * (size_t)m + 2); }
    else { pm[0] = A.pos[3 * (size_t)m]; pm[1] = A.pos[3 * (size_t)m + 1]; pm[2] = A.pos[3 * (size_t)m + 2]; }
    double e = 0.0, fx = 0.0, fy = 0.0, fz = 0.0;
    // TAB instances: what the triplet loop needs of the basis -- the legs of trio 0, the species -> trio and trio -> grid offset
    // tables (one register each), leg n's knot records into LDS -- is requested HERE, ahead of the pair phase that hides it
    KnotRec *kn_lds = CW ? (KnotRec *)recs_l + (load_const(&load_const(&B->trios)->leg[2].rec_off) + 3)
                         : (KnotRec *)(smem + ((((unsigned char *)(ushift + cap) - smem) + 15) & ~(size_t)15));
    int trio_tab = -1, lut_tab = 0;
    if (TAB && !GATHER) {
        // (the legs' descriptors themselves are scalar loads, read again where they are used: held from here they cost ~40 of
        // the scalar registers the kernel is short of)
        const TrioDev *t0 = load_const(&B->trios);
        const int n_rec_off = load_const(&t0->leg[2].rec_off), n_nk = load_const(&t0->leg[2].nk);
        trio_tab = B->trio_of[sm * UF3_MAX_SPECIES * UF3_MAX_SPECIES + lane];
        typedef const __attribute__((address_space(1))) TrioDev *GlobalTrio;
        if (lane < load_const(&B->T)) lut_tab = ((GlobalTrio)t0)[lane].lut_off;
        // leg n's knot records (intervals 3 .. nk - 5) into LDS: 96 bytes per triplet less through the vector memory path,
        // which bounds this kernel (the coefficient rows stay there: 512 bytes per triplet)
        if (!CW) {
            const int4 *src = (const int4 *)(recs_g + n_rec_off + 3);
            int4 *dst = (int4 *)kn_lds;
            for (int q = lane; q < (n_nk - 7) * 6; q += WAVE) dst[q] = src[q];
        }
    }
    PhaseClock pce;                   // (-DUF3_PHASE_TIMING builds only: tools/experiments/eval_phase.py)
    if (lane == 0) e = A.c1[sm];
    // 2-body: bonds inside their pair's range are queued in LDS and evaluated 64 at a time (about one candidate in
    // five survives the range test: evaluating in place would leave most lanes idle in the spline code)
    int queued = 0;
    const int ev_pairs_uniform = load_const(&B->pairs_uniform);
    const double ev_rmin0 = load_const(&B->pairs[0].s_lo), ev_rmax0 = load_const(&B->pairs[0].s_hi);   // (thresholds on the squared distance)
    const double s3_lo = load_const(&B->s3_lo), s3_hi = load_const(&B->s3_hi);
    auto drain = [&](int count) {
        const double *c = queue + (size_t)lane * EVAL_Q;
        if (lane < count) {
            const double dx = c[0], dy = c[1], dz = c[2], d = sqrt(c[3]);        // (queued with the squared distance, see below)
            const int pair_idx = sm * UF3_MAX_SPECIES + (int)c[4];
            KnotRec kr;
            const LegDev leg = ev_pairs_uniform ? load_const(&B->pairs[0].leg) : B->pairs[B->pair_of[pair_idx]].leg;
            const int col = B->pair_col[pair_idx];
            int i = load_interval<CW ? 3 : 1>(recs_l, leg, d, kr);
            double v[4], dv[4];
            bspline4<true>(kr, d, v, dv);
            const double *cf = c2_l + (col - S) + (i - 3);
            double phi = 0, dphi = 0;
            for (int q = 0; q < 4; q++) { phi += cf[q] * v[q]; dphi += cf[q] * dv[q]; }
            e += phi;
            const double t = dphi * fast_rcp(d), s = 2.0 * t;
            fx += s * dx; fy += s * dy; fz += s * dz;
            if (want_v) {   // d/d(strain) of the directed pair sum: phi'(r) r (x) r / r
                vir[0] += t * dx * dx; vir[1] += t * dy * dy; vir[2] += t * dz * dz;
                vir[3] += t * dy * dz; vir[4] += t * dx * dz; vir[5] += t * dx * dy;
            }
        }
    };
    if (MD) {
        // the lists are valid while no atom has moved more than skin / 2 since they were built, and for the species they were built for
        {
            const double ux0 = pm[0] - load_const(A.pos_ref + 3 * (size_t)m), uy0 = pm[1] - load_const(A.pos_ref + 3 * (size_t)m + 1),
                         uz0 = pm[2] - load_const(A.pos_ref + 3 * (size_t)m + 2);
            const double moved = ux0 * ux0 + uy0 * uy0 + uz0 * uz0;
            const int zz = load_const(A.z_now + m);
            const bool other_species = zz < 0 || zz >= 120 || B->z2s[zz] != sm;
            if (lane == 0) {
                if (!(moved <= A.md_hard2) || other_species) A.md_flags[0] = 1;
                if (!(moved <= A.md_soft2)) A.md_flags[1] = 1;
            }
        }
        const int n_sup = min(load_const(A.sup_cnt + m), A.sup_cap);
        const SupEntry *sup = A.sup_ent + (size_t)m * A.sup_cap;
        const size_t base3 = (size_t)m * cap;
        for (int q0 = 0; q0 < n_sup; q0 += WAVE) {
            const int q = q0 + lane;
            bool ok = q < n_sup;
            const SupEntry en = sup[ok ? q : 0];
            SlotRec sr;
            sr.x = A.pos[3 * (size_t)en.parent]; sr.y = A.pos[3 * (size_t)en.parent + 1]; sr.z = A.pos[3 * (size_t)en.parent + 2];
            int s0, s1, s2;
            unpack3(en.shiftc, s0, s1, s2);
            const int sj = en.spec & 0xff, rev1 = en.spec >> 8;       // (1 + this atom's position in the neighbour's list, k_sup_reverse)
            double dx, dy, dz;
            image_delta(g, sr, s0, s1, s2, pm, dx, dy, dz);
            const double d = norm3_sq_rn(dx, dy, dz);
            double rmin = ev_rmin0, rmax = ev_rmax0;
            if (!ev_pairs_uniform) { const PairDev &pd = B->pairs[B->pair_of[sm * UF3_MAX_SPECIES + sj]]; rmin = pd.s_lo; rmax = pd.s_hi; }
            const bool ok3 = ok & (d > s3_lo) & (d <= s3_hi);
            ok = ok & (d > rmin) & (d < rmax);
            // 3-body list: the survivors in list order ARE the (species, supercell index) order
            const unsigned long long mask3 = __ballot(ok3);
            if (ok3) {
                const int slot = count3 + mbcnt(mask3);
                if (slot < cap) {
                    // (no list in memory: the collection pass of this route reads the inbox, not the neighbours' lists)
                    ox[slot] = dx; oy[slot] = dy; oz[slot] = dz; orr[slot] = sqrt(d);
                    oparent[slot] = en.parent; oshift[slot] = en.shiftc; osidx[slot] = en.sidx; ospec[slot] = sj;
                    ooff[slot] = rev1;
                    A.md_surv[base3 + slot] = q;
                    gx[slot] = 0.0; gy[slot] = 0.0; gz[slot] = 0.0;
                }
            }
            count3 += __popcll(mask3);
            const unsigned long long mask = __ballot(ok);
            if (ok) {
                double *c = queue + (size_t)(queued + mbcnt(mask)) * EVAL_Q;
                c[0] = dx; c[1] = dy; c[2] = dz; c[3] = d; c[4] = (double)sj;
            }
            queued += __popcll(mask);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (CW: the workgroup's ONE barrier, behind the first batch's two round trips -- the tables have landed long since)
            if (CW && q0 == 0) __syncthreads();
            if (queued >= WAVE) {
                drain(WAVE);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                queued -= WAVE;                                             // move the tail to the front
                double tail[EVAL_Q];
                if (lane < queued) for (int u = 0; u < EVAL_Q; u++) tail[u] = queue[(size_t)(WAVE + lane) * EVAL_Q + u];
                __builtin_amdgcn_wave_barrier();
                if (lane < queued) for (int u = 0; u < EVAL_Q; u++) queue[(size_t)lane * EVAL_Q + u] = tail[u];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else {
    if (CW) __syncthreads();          // (the walk drains its first batch of pair splines from the LDS copies: the workgroup's one barrier)
    for_each_candidate(g, A.cl, m, [&](bool ok, const SlotRec &sr, int sj, int s0, int s1, int s2) {
        double dx = 0, dy = 0, dz = 0, d = 0;
        bool ok3 = false;
        if (ok) {
            // (d: the SQUARED distance here; the ranges as thresholds on it that give the decisions of the correctly rounded root,
            // PairDev::s_lo -- the root is taken where a kept candidate is used)
            double rmin = ev_rmin0, rmax = ev_rmax0;
            if (!ev_pairs_uniform) { const PairDev &pd = B->pairs[B->pair_of[sm * UF3_MAX_SPECIES + sj]]; rmin = pd.s_lo; rmax = pd.s_hi; }
            image_delta(g, sr, s0, s1, s2, pm, dx, dy, dz);
            d = norm3_sq_rn(dx, dy, dz);
            ok3 = fuse & (d > s3_lo) & (d <= s3_hi);                 // angles.py:340: lower strict, upper inclusive
            ok = (d > rmin) & (d < rmax);
        }
        if (fuse) {
            const unsigned long long mask3 = __ballot(ok3);
            if (ok3) {
                const int slot = count3 + mbcnt(mask3);
                if (slot < cap) {
                    const int sidx = supercell_index(g, s0, s1, s2, sr.atom - g.atom_lo);
                    ux[slot] = dx; uy[slot] = dy; uz[slot] = dz; ur[slot] = d;
                    uparent[slot] = sr.atom; ushift[slot] = pack3(s0, s1, s2);
                    ukey[slot] = ((unsigned long long)sj << 32) | (unsigned)sidx;
                }
            }
            count3 += __popcll(mask3);
        }
        const unsigned long long mask = __ballot(ok);
        if (ok) {
            double *c = queue + (size_t)(queued + mbcnt(mask)) * EVAL_Q;
            c[0] = dx; c[1] = dy; c[2] = dz; c[3] = d; c[4] = (double)sj;
        }
        queued += __popcll(mask);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (queued >= WAVE) {
            drain(WAVE);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            queued -= WAVE;                                             // move the tail to the front
            double tail[EVAL_Q];
            if (lane < queued) for (int q = 0; q < EVAL_Q; q++) tail[q] = queue[(size_t)(WAVE + lane) * EVAL_Q + q];
            __builtin_amdgcn_wave_barrier();
            if (lane < queued) for (int q = 0; q < EVAL_Q; q++) queue[(size_t)lane * EVAL_Q + q] = tail[q];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    });
    }
    pce.lap(9);                       // (MD route: the list filter; else the candidate walk with its full pair batches)
    if (CW && MD && load_const(A.sup_cnt + m) <= 0) __syncthreads();       // (an atom without candidates never entered the loop above)
    drain(queued);
    pce.lap(10);
    if (load_const(&B->T) > 0) {
        size_t base = (size_t)m * cap;
        int n;
        if (fuse) {
            // the list of this atom from the walk above, rank-sorted by (species, supercell index) like k_build_n3's:
            // into LDS for the loops below and into the batch's list array for k_eval_collect
            EVAL_SYNC();
            if (count3 > cap) { if (lane == 0) atomicMax(A.n3_need, count3); count3 = cap; }
            n = count3;
            if (lane == 0) A.n3.cnt[m] = n;
            for (int q = lane; q < n; q += WAVE) {
                const unsigned long long k = ukey[q];
                int rank = 0;
                for (int f = 0; f < n; f++) rank += ukey[f] < k;
                N3Entry en;
                en.dx = ux[q]; en.dy = uy[q]; en.dz = uz[q]; en.r = sqrt(ur[q]);
                en.parent = uparent[q]; en.shiftc = ushift[q]; en.sidx = (int)(unsigned)k; en.spec = (int)(k >> 32);
                A.n3.ent[base + rank] = en;
                ox[rank] = en.dx; oy[rank] = en.dy; oz[rank] = en.dz; orr[rank] = en.r;
                oparent[rank] = en.parent; oshift[rank] = en.shiftc; osidx[rank] = en.sidx; ospec[rank] = en.spec;
                gx[rank] = 0.0; gy[rank] = 0.0; gz[rank] = 0.0;
            }
        } else if (MD) {
            // (written in place by the filter above, already in order)
            EVAL_SYNC();
            if (count3 > cap) { if (lane == 0) atomicMax(A.n3_need, count3); count3 = cap; }
            n = count3;
            if (lane == 0) A.n3.cnt[m] = n;
        } else {
            n = A.n3.cnt[m];
            for (int q = lane; q < n; q += WAVE) {
                const N3Entry en = A.n3.ent[base + q];
                ox[q] = en.dx; oy[q] = en.dy; oz[q] = en.dz; orr[q] = en.r;
                oparent[q] = en.parent; oshift[q] = en.shiftc; osidx[q] = en.sidx; ospec[q] = en.spec;
                if (!GATHER) { gx[q] = 0.0; gy[q] = 0.0; gz[q] = 0.0; }
            }
        }
        EVAL_SYNC();
        pce.lap(11);
        int n_pairs = n * (n - 1) / 2;
        // Centre pass, one set of 3-body legs (trio_value_tab): the centre legs of every bond once, into LDS (over the queue of
        // the pair walk, which is done with); the species -> trio and trio -> grid offset tables in one register each, looked up
        // with lane shuffles instead of two dependent loads
        const bool tab_path = TAB && !GATHER && n_pairs > 0;
        // (the per-bond tables over the queue of the pair walk, which is done with; lists longer than EVAL_TAB_CAP: behind the knot
        // records -- the choice of the instance must not depend on the capacity, or a context's first call, at the estimated
        // capacity, would differ from the later ones in the last bit)
        const int ts = CW ? cap : (cap <= EVAL_TAB_CAP ? EVAL_TAB_CAP : cap);      // (CW: the room behind the tables is the stage of the force gather)
        double *tlv = cap <= EVAL_TAB_CAP ? queue : (double *)(kn_lds + EVAL_TAB_KN), *tmv = tlv;
        int *tli = (int *)(tlv + 16 * ts), *tmi = tli;
        LegDev leg_n, leg_n_lds;
        int tab_dim_m = 0, tab_dim_n = 0;
        if (tab_path) {
            const TrioDev *t0 = load_const(&B->trios);
            const LegDev l0 = load_const(&t0->leg[0]), l1 = load_const(&t0->leg[1]);
            leg_n = load_const(&t0->leg[2]);
            tab_dim_m = load_const(&t0->dim_m); tab_dim_n = load_const(&t0->dim_n);
            leg_n_lds = leg_n;
            leg_n_lds.rec_off = -3;                      // (interval i of the copy: kn_lds[i - 3])
            const bool same01 = l0.rec_off == l1.rec_off && l0.nk == l1.nk;
            if (!same01) { tmv = tlv + 8 * ts; tmi = tli + ts; }
            for (int q = lane; q < n; q += WAVE) {
                const double r = orr[q];
                for (int which = 0; which < (same01 ? 1 : 2); which++) {
                    const LegDev &lg = which ? l1 : l0;
                    int i = -1;
                    double v[4] = {0, 0, 0, 0}, d[4] = {0, 0, 0, 0};
                    if ((r > lg.t0) & (r < lg.tlast)) {
                        KnotRec k;
                        i = load_interval<CW ? 3 : 1>(recs_l, lg, r, k);
                        bspline4<true>(k, r, v, d);
                    }
                    double *dst = (which ? tmv : tlv) + 8 * q;
                    if (WIN) {
                        // dense over the window rows (EVAL_CW_EXT of them): value | derivative of row w, zero where the bond's four
                        // functions (i - 3 .. i) do not reach
                        for (int u = 0; u < 2 * EVAL_CW_EXT; u++) dst[u] = 0.0;
                        const int w0 = i - 3 - A.cw_lo;
                        for (int u = 0; u < 4; u++)
                            if (i >= 0 && (unsigned)(w0 + u) < (unsigned)EVAL_CW_EXT) { dst[w0 + u] = v[u]; dst[EVAL_CW_EXT + w0 + u] = d[u]; }
                    } else
                    for (int u = 0; u < 4; u++) { dst[u] = v[u]; dst[4 + u] = d[u]; }
                    (which ? tmi : tli)[q] = i;
                }
            }
            EVAL_SYNC();
        }
        pce.lap(14);
        // (CW: the stage of the force gather behind the per-bond tables -- 6 x 64 doubles, into the room of gx / gy / gz, which this
        // route does not use: lists of at most 16 entries, see the host's LDS budget -- and entry q's force in lane q)
        double *fstage = (double *)(((size_t)(tli + 2 * ts) + 15) & ~(size_t)15);
        double ex_f[3] = {0.0, 0.0, 0.0};
#pragma unroll 1
        for (int p0 = 0; p0 < n_pairs; p0 += WAVE) {      // (not unrolled: two triplets' loads in flight per lane do not fit the registers)
            const bool act = p0 + lane < n_pairs;
            const int p = act ? p0 + lane : 0;
            // (pair index -> (aa < bb) as in trio_walk_geom: hardware root + one guard each way, no loops)
            int bb = (int)((1.0f + __builtin_amdgcn_sqrtf(fmaf(8.0f, (float)p, 1.0f))) * 0.5f);
            bb -= (bb * (bb - 1) / 2 > p) ? 1 : 0;
            bb += ((bb + 1) * bb / 2 <= p) ? 1 : 0;
            int aa = p - bb * (bb - 1) / 2;
            double rl = orr[aa], rm = orr[bb];
            double rn = norm3_leg(ox[bb] - ox[aa], oy[bb] - oy[aa], oz[bb] - oz[aa]);
            double val, gr[3];
            if (CW) { val = 0.0; gr[0] = 0.0; gr[1] = 0.0; gr[2] = 0.0; }       // (a CW lane without a triplet carries zeros; elsewhere it leaves the trip)
#if defined(UF3_ABLATE_EVAL) && UF3_ABLATE_EVAL == 1
            if (rl > 0) continue;           // (experiment: no triplet values)
#endif
            bool good;
            if (TAB) {
                // (the shuffles in uniform control flow: every lane of the tables takes part)
                const int trio = __shfl(trio_tab, ospec[aa] * UF3_MAX_SPECIES + ospec[bb]);
                const int lut_off = __shfl(lut_tab, max(trio, 0));
                const int il = tli[aa], im = tmi[bb];
                good = act & (trio >= 0) & (il >= 0) & (im >= 0) & (rn > leg_n.t0) & (rn < leg_n.tlast);
                if (!CW && !good) continue;
                if (WIN) {
                    (void)lut_off;
                    if (good)
                        trio_value_tab_cw<EVAL_CW_EXT, CW ? 3 : 1>(kn_lds, CW ? smem_all : (const unsigned char *)A.c3w,
                                                                   max(trio, 0) * (EVAL_CW_EXT * EVAL_CW_EXT * tab_dim_n * 8), leg_n_lds, tab_dim_n,
                                                                   tlv + 8 * aa, tmv + 8 * bb, rn, want_f || want_v, val, gr);
                } else
                trio_value_tab<3>(kn_lds, A.c3, lut_off, leg_n_lds, tab_dim_m, tab_dim_n, il, im, tlv + 8 * aa, tmv + 8 * bb, rn,
                               want_f || want_v, val, gr);
            } else {
                int trio = B->trio_of[(sm * UF3_MAX_SPECIES + ospec[aa]) * UF3_MAX_SPECIES + ospec[bb]];
                if (!act || !trio_value(B, A.c3, trio, rl, rm, rn, want_f || want_v, val, gr)) continue;
                good = true;
            }
            // (CW: a lane without a triplet carries zeros through the force arithmetic below -- every lane takes part in the gather)
            e += val;
            if (want_f || (CW && want_v)) {   // F_m = -dV/dR_m = gl * u_ij + gm * u_ik
                const double a = gr[0] * fast_rcp(rl), b = gr[1] * fast_rcp(rm);
                fx += a * ox[aa] + b * ox[bb]; fy += a * oy[aa] + b * oy[bb]; fz += a * oz[aa] + b * oz[bb];
                if (!GATHER) {   // F_j = -gl u_ij + gn (R_k - R_j) / rn,  F_k = -gm u_ik - gn (R_k - R_j) / rn
                    const double cc = gr[2] * fast_rcp(rn);
                    const double cx = cc * (ox[bb] - ox[aa]), cy = cc * (oy[bb] - oy[aa]), cz = cc * (oz[bb] - oz[aa]);
                    if (CW) {
                        // Six ds_add_f64 per lane onto 14 addresses were 15 % of the kernel (34 of 230 us at 50 k atoms: a wave's adds
                        // to one address serialise).  Instead: every lane leaves its two forces in a stage (six conflict-free 8-byte
                        // stores), and lane q < n gathers entry q's share of this trip -- pair (q, b) sits at lane
                        // hi (hi - 1) / 2 + lo - p0 -- into registers it keeps across the trips, in a fixed order.
                        double *st = fstage + lane;
                        st[0] = cx - a * ox[aa]; st[64] = cy - a * oy[aa]; st[128] = cz - a * oz[aa];
                        st[192] = -cx - b * ox[bb]; st[256] = -cy - b * oy[bb]; st[320] = -cz - b * oz[bb];
                        wave_sync();
                        {
                            // lanes (entry q = lane / 4, part = lane % 4): the partners o = part, part + 4, ... of entry q, four loads in flight
                            const int gq = lane >> 2, gpart = lane & 3;
                            double t0[4], t1[4], t2[4];
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const int o = gpart + 4 * k;
                                const int lo = min(gq, o), hi = max(gq, o);
                                const int rel = hi * (hi - 1) / 2 + lo - p0;
                                const bool take = (gq < n) & (o < n) & (o != gq) & ((unsigned)rel < (unsigned)WAVE);
                                const double *src = fstage + (take ? rel + (gq == lo ? 0 : 192) : 0);
                                t0[k] = src[0]; t1[k] = src[64]; t2[k] = src[128];
                                if (!take) { t0[k] = 0.0; t1[k] = 0.0; t2[k] = 0.0; }
                            }
#pragma unroll
                            for (int k = 0; k < 4; k++) { ex_f[0] += t0[k]; ex_f[1] += t1[k]; ex_f[2] += t2[k]; }
                        }
                        wave_sync();
                    } else {
                    lds_add(gx + aa, cx - a * ox[aa]); lds_add(gy + aa, cy - a * oy[aa]); lds_add(gz + aa, cz - a * oz[aa]);
                    lds_add(gx + bb, -cx - b * ox[bb]); lds_add(gy + bb, -cy - b * oy[bb]); lds_add(gz + bb, -cz - b * oz[bb]);
                    }
                }
            }
            // (CW: the triplets' strain derivative comes out of the gathered forces behind the loop -- with the centre at the origin a
            // triplet's sum over legs of dV/dr r (x) r / r is -(o_a (x) F_a + o_b (x) F_b) -- instead of 30 multiply-adds per triplet)
            if (want_v && !CW) {   // each triplet once (at its centre): sum over legs of dV/dr * r (x) r / r
                double ta = gr[0] * fast_rcp(rl), tb = gr[1] * fast_rcp(rm), tc = gr[2] * fast_rcp(rn);
                double cx = ox[bb] - ox[aa], cy = oy[bb] - oy[aa], cz = oz[bb] - oz[aa];
                vir[0] += ta * ox[aa] * ox[aa] + tb * ox[bb] * ox[bb] + tc * cx * cx;
                vir[1] += ta * oy[aa] * oy[aa] + tb * oy[bb] * oy[bb] + tc * cy * cy;
                vir[2] += ta * oz[aa] * oz[aa] + tb * oz[bb] * oz[bb] + tc * cz * cz;
                vir[3] += ta * oy[aa] * oz[aa] + tb * oy[bb] * oz[bb] + tc * cy * cz;
                vir[4] += ta * ox[aa] * oz[aa] + tb * ox[bb] * oz[bb] + tc * cx * cz;
                vir[5] += ta * ox[aa] * oy[aa] + tb * ox[bb] * oy[bb] + tc * cx * cy;
            }
        }
        pce.lap(12);
        if (CW && (want_f || want_v)) {
            EVAL_SYNC();
#pragma unroll
            for (int u = 0; u < 3; u++) { ex_f[u] += dpp_take<0xB1, 0xf>(ex_f[u]); ex_f[u] += dpp_take<0x4E, 0xf>(ex_f[u]); }
            if (want_v && (lane & 3) == 0 && (lane >> 2) < n) {
                const int q = lane >> 2;
                const double qx = ox[q], qy = oy[q], qz = oz[q];
                vir[0] -= qx * ex_f[0]; vir[1] -= qy * ex_f[1]; vir[2] -= qz * ex_f[2];
                vir[3] -= 0.5 * (qy * ex_f[2] + qz * ex_f[1]); vir[4] -= 0.5 * (qx * ex_f[2] + qz * ex_f[0]); vir[5] -= 0.5 * (qx * ex_f[1] + qy * ex_f[0]);
            }
        }
        if (want_f && !GATHER && MD) {
            // what this centre's triplets put on each neighbour, straight into the neighbour's inbox at ITS list position of this atom
            if (!CW) EVAL_SYNC();
            for (int q = CW ? (lane >> 2) : lane; q < n; q += WAVE) {
                const int rev1 = ooff[q];
                if (rev1 > 0 && (!CW || (lane & 3) == 0)) {
                    typedef double inbox4 __attribute__((ext_vector_type(4)));
                    const inbox4 v = {CW ? ex_f[0] : gx[q], CW ? ex_f[1] : gy[q], CW ? ex_f[2] : gz[q], A.md_stamp};      // (CW: q == lane / 4)
                    *(inbox4 *)(A.md_inbox + 4 * ((size_t)oparent[q] * A.sup_cap + (rev1 - 1))) = v;
                    if (A.md_mark) A.md_mark[oparent[q]] = A.md_mark_now;
                }
            }
        }
        if (want_f && !GATHER && !MD) {
            EVAL_SYNC();
            for (int q = lane; q < n; q += WAVE) {
                double *dst = A.nbr_f + 3 * (base + q);
                dst[0] = gx[q]; dst[1] = gy[q]; dst[2] = gz[q];
            }
        }
        if (want_f && GATHER) {
            int total = 0;
            for (int e0 = 0; e0 < n; e0 += WAVE) {
                int q = e0 + lane;
                int cnt = q < n ? A.n3.cnt[oparent[q]] : 0;
                const int incl = wave_scan_incl(cnt);
                if (q < n) ooff[q] = total + incl - cnt;
                total += __builtin_amdgcn_readlane(incl, WAVE - 1);
            }
            if (lane == 0) ooff[n] = total;
            EVAL_SYNC();
            int m_local = m - g.atom_lo;
            for (int p = lane; p < total; p += WAVE) {
                int lo = 0, hi = n - 1;
                while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (ooff[mid] <= p) lo = mid; else hi = mid - 1; }
                int q = lo, kk = p - ooff[q];
                size_t kb = (size_t)oparent[q] * cap + kk;
                int s0, s1, s2;
                unpack3(oshift[q], s0, s1, s2);
                const N3Entry ke = A.n3.ent[kb];
                if (ke.parent == m && ke.shiftc == pack3(-s0, -s1, -s2)) continue;
                int ksp = ke.spec, ksidx = ke.sidx;
                int msidx = supercell_index(g, -s0, -s1, -s2, m_local);
                double vx = ke.dx, vy = ke.dy, vz = ke.dz, rk = ke.r;
                double ex = ox[q] + vx, ey = oy[q] + vy, ez = oz[q] + vz;
                double rn = norm3_leg(vx - (-ox[q]), vy - (-oy[q]), vz - (-oz[q]));
                bool m_first = neighbour_is_first(g, sm, ksp, s0, s1, s2, m_local, msidx, ksidx, ke.shiftc,
                                                  ke.parent - g.atom_lo);
                int sc = ospec[q];
                double val, gr[3];
                int trio; double rl, rm;
                if (m_first) { rl = orr[q]; rm = rk; trio = B->trio_of[(sc * UF3_MAX_SPECIES + sm) * UF3_MAX_SPECIES + ksp]; }
                else { rl = rk; rm = orr[q]; trio = B->trio_of[(sc * UF3_MAX_SPECIES + ksp) * UF3_MAX_SPECIES + sm]; }
                if (!trio_value(B, A.c3, trio, rl, rm, rn, true, val, gr)) continue;
                double ge = (m_first ? gr[0] : gr[1]) * fast_rcp(orr[q]), gn = gr[2] * fast_rcp(rn);
                fx += ge * ox[q] + gn * ex; fy += ge * oy[q] + gn * ey; fz += ge * oz[q] + gn * ez;
            }
        }
    }
    pce.lap(13);
    e = wave_sum(e);
    if (lane == 0) A.e_atom[m] = e;
    if (want_v) {
        for (int q = 0; q < 6; q++) { double v = wave_sum(vir[q]); if (lane == 0) A.virial[6 * (size_t)m + q] = v; }
    }
    if (want_f) {
        fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
        if (lane == 0) { A.forces[3 * (size_t)m] = fx; A.forces[3 * (size_t)m + 1] = fy; A.forces[3 * (size_t)m + 2] = fz; }
    }
    pce.lap(15);
}

// second pass of the two-pass evaluator: 16 lanes per atom m; for every entry (centre c, image shift s) of m's list the
// lanes look m up in c's list (entry with parent m and shift -s: the lists are symmetric) and take what c's triplets put
// on it; fixed lane order + a fixed shuffle tree: deterministic
// (the 16 lanes of atom m; every lane returns the sum)
__device__ __forceinline__ void eval_collect_atom(const EvalArgs &A, int m, int sub, double &sx, double &sy, double &sz) {
    const int cap = A.n3.cap, n = A.n3.cnt[m];
    const int c_lo = A.halo_mark ? A.atom_lo : 0, c_hi = A.halo_mark ? A.atom_hi : A.natoms;
    const N3Entry *mine = A.n3.ent + (size_t)m * cap;
    sx = 0.0; sy = 0.0; sz = 0.0;
    // lanes <-> own entries; each lane scans its centre's list (independent loads: three dependent round trips per
    // entry instead of three per list position)
    for (int q = sub; q < n; q += 16) {
        const int2 me = *(const int2 *)&mine[q].parent;
        const int c = me.x;
        if (c < c_lo || c >= c_hi) continue;               // (a centre outside the block: another rank's)
        int s0, s1, s2;
        unpack3(me.y, s0, s1, s2);
        const int back = pack3(-s0, -s1, -s2), nc = A.n3.cnt[c];
        const N3Entry *theirs = A.n3.ent + (size_t)c * cap;
        int hit = -1;
        for (int r0 = 0; r0 < nc; r0 += 8) {                // eight entries' keys in flight together (a plain loop waits for each)
            int2 key[8];
#pragma unroll
            for (int u = 0; u < 8; u++) key[u] = *(const int2 *)&theirs[min(r0 + u, nc - 1)].parent;
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (r0 + u < nc && key[u].x == m && key[u].y == back) hit = r0 + u;
        }
        if (hit >= 0) {
            const double *f = A.nbr_f + 3 * ((size_t)c * cap + hit);
            sx += f[0]; sy += f[1]; sz += f[2];
        }
    }
    row16_sum3(sx, sy, sz);
}

__global__ void __launch_bounds__(256)
k_eval_collect(EvalArgs A) {
    // (whole batch: one contiguous eighth of the atoms per XCD, as in k_eval.  A block of centres: the atoms with work are the
    // block and its halo -- one contiguous stretch, which that mapping would hand to one or two XCDs: plain round-robin instead)
    const int wg = A.halo_mark ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    const int m = wg * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (m < A.natoms && (!A.halo_mark || (m >= A.atom_lo && m < A.atom_hi) || A.halo_mark[m])) {
        double sx, sy, sz;
        eval_collect_atom(A, m, sub, sx, sy, sz);
        if (sub == 0) { double *f = A.forces + 3 * (size_t)m; f[0] += sx; f[1] += sy; f[2] += sz; }
    }
}

// collection pass of the MD route: atom m adds what its neighbours' centre passes left in its inbox, over the entries of ITS
// 3-body list in list order (16 lanes, entries strided, one shuffle tree: the order of the additions depends on the step's lists
// only, not on when the superset lists were built).  An entry whose stamp is not this launch's has no counterpart this step.
// The same for a block of CENTRES [atom_lo, atom_hi) of a decomposed frame (uf3_eval_centres on the MD route): every atom of the
// frame checks ITSELF against the lists' reference positions and species (a rank's lists are only good while no atom of the
// frame -- inside its block or not -- has moved skin / 2), and the atoms a centre of the block has written to this step
// (md_mark == the launch's number; the block's own atoms always) add what they find in their inbox under this launch's stamp,
// in the order of their superset lists.  Rows of all other atoms stay zero.
__global__ void __launch_bounds__(256)
k_eval_collect_md_halo(EvalArgs A) {
    const int m = (int)blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (m >= A.natoms) return;
    const bool own = m >= A.atom_lo && m < A.atom_hi;
    if (sub == 0 && !own) {                        // (the block's atoms have checked themselves in the centre pass)
        const double ux = A.pos[3 * (size_t)m] - A.pos_ref[3 * (size_t)m], uy = A.pos[3 * (size_t)m + 1] - A.pos_ref[3 * (size_t)m + 1],
                     uz = A.pos[3 * (size_t)m + 2] - A.pos_ref[3 * (size_t)m + 2];
        const double moved = ux * ux + uy * uy + uz * uz;
        const int zz = A.z_now[m];
        const bool other_species = zz < 0 || zz >= 120 || A.B->z2s[zz] != A.spec[m];
        if (!(moved <= A.md_hard2) || other_species) A.md_flags[0] = 1;
        if (!(moved <= A.md_soft2)) A.md_flags[1] = 1;
    }
    if (!own && A.md_mark[m] != A.md_mark_now) return;
    const int n = min(A.sup_cnt[m], A.sup_cap);
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int q = sub; q < n; q += 16) {
        typedef double inbox4 __attribute__((ext_vector_type(4)));
        const inbox4 v = *(const inbox4 *)(A.md_inbox + 4 * ((size_t)m * A.sup_cap + q));
        if (v[3] == A.md_stamp) { sx += v[0]; sy += v[1]; sz += v[2]; }
    }
    row16_sum3(sx, sy, sz);
    if (sub == 0) { double *f = A.forces + 3 * (size_t)m; f[0] += sx; f[1] += sy; f[2] += sz; }
}

__global__ void __launch_bounds__(256)
k_eval_collect_md(EvalArgs A) {
    const int wg = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    const int m = wg * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const bool active = m < A.natoms;
    if (!active && !A.part_e) return;
    if (active) {
        const int cap = A.n3.cap, n = min(A.n3.cnt[m], cap);
        double sx = 0.0, sy = 0.0, sz = 0.0;
        for (int t = sub; t < n; t += 16) {
            typedef double inbox4 __attribute__((ext_vector_type(4)));
            const int q = A.md_surv[(size_t)m * cap + t];
            const inbox4 v = *(const inbox4 *)(A.md_inbox + 4 * ((size_t)m * A.sup_cap + q));
            if (v[3] == A.md_stamp) { sx += v[0]; sy += v[1]; sz += v[2]; }
        }
        row16_sum3(sx, sy, sz);
        if (sub == 0) { double *f = A.forces + 3 * (size_t)m; f[0] += sx; f[1] += sy; f[2] += sz; }
    }
    if (A.part_e) {
        // this workgroup's sixteen atoms' energies (and strain derivatives), added in atom order: what k_frame_sum sums afterwards
        __shared__ double pe[16][8];
        const int a = threadIdx.x >> 4;
        if (sub < 7) pe[a][sub] = !active ? 0.0 : (sub == 0 ? A.e_atom[m] : (A.virial ? A.virial[6 * (size_t)m + sub - 1] : 0.0));
        __syncthreads();
        if (threadIdx.x < 7 && (size_t)wg * 16 < (size_t)A.natoms) {
            double s_ = 0.0;
            for (int q = 0; q < 16; q++) s_ += pe[q][threadIdx.x];
            if (threadIdx.x == 0) A.part_e[wg] = s_; else if (A.virial) A.part_v[6 * (size_t)wg + threadIdx.x - 1] = s_;
        }
        if (wg == 0 && threadIdx.x == 0) { A.part_off[0] = 0; A.part_off[1] = (A.natoms + 15) / 16; }
    }
}

// per-frame sums of per-atom quantities: blockIdx.y = 0 energy (width 1), 1..6 virial components (width 6, if
// given); deterministic tree
// (flags_dst: the status words of the launches before it ride along behind the results, so that a small batch needs
// one download)
// (mirror: the caller's pinned result block, energies [nf] | virials [nf][6] | forces [n_force] | status words -- device-visible
// host memory, written here directly so that a small batch needs no copy-engine transfer behind its last kernel)
// (one per-frame sum by the first `width_t` threads of the workgroup -- every thread of the workgroup calls; the order of
// the additions depends on width_t only)
__device__ __forceinline__ double frame_sum_tree(const double *src, int width, int64_t a0, int64_t a1, int width_t, double *part) {
    double s = 0.0;
    if ((int)threadIdx.x < width_t) {
        // (eight loads in flight per trip, added in the order a plain loop adds them: the loop was a chain of dependent round
        // trips, 16 us for a 50 k-atom frame)
        int64_t a = a0 + threadIdx.x;
        const int64_t bd = width_t;
        for (; a + 7 * bd < a1; a += 8 * bd) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[(a + u * bd) * width];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; a < a1; a += bd) s += src[a * width];
        part[threadIdx.x] = s;
    }
    __syncthreads();
    for (int w = width_t / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    const double r = part[0];
    __syncthreads();
    return r;
}

__global__ void k_frame_sum(const double *e_atom, const double *v_atom, const int64_t *atom_offsets, double *e_out,
                            double *v_out, const int *flags_src, int *flags_dst, double *mirror, const double *forces,
                            int n_force, int a_lo, int a_hi, unsigned seq, int *tail_count, unsigned *seq_dst) {
    __shared__ double part[1024];
    const int f = blockIdx.x, comp = (int)blockIdx.y - 1;
    if (flags_dst && f == 0 && comp < 0 && threadIdx.x < 4) flags_dst[threadIdx.x] = flags_src[threadIdx.x];
    if (mirror && f == 0 && comp < 0) {
        const int nf = gridDim.x;
        if (forces) for (int q = threadIdx.x; q < n_force; q += blockDim.x) mirror[7 * (size_t)nf + q] = forces[q];
        if (threadIdx.x < 4) ((int *)(mirror + 7 * (size_t)nf + n_force))[threadIdx.x] = flags_src[threadIdx.x];
    }
    const double *src = comp < 0 ? e_atom : v_atom + comp;
    const int width = comp < 0 ? 1 : 6;
    // (a share of a block of atoms: only [a_lo, a_hi) carries values)
    const int64_t a0 = max(atom_offsets[f], (int64_t)a_lo), a1 = min(atom_offsets[f + 1], (int64_t)a_hi);
    const double total = frame_sum_tree(src, width, a0, a1, (int)blockDim.x, part);
    if (threadIdx.x == 0) {
        if (comp < 0) e_out[f] = total; else v_out[(size_t)f * 6 + comp] = total;
        if (mirror) { if (comp < 0) mirror[f] = total; else mirror[gridDim.x + (size_t)f * 6 + comp] = total; }
    }
    // (seq != 0: a launch into the caller's pinned block -- the last store of its last workgroup, behind system-scope fences, is
    // the call's sequence number, which the host entry polls for instead of waiting for the stream's completion signal)
    if (seq) {
        __shared__ int last;
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const int n_blocks = (int)(gridDim.x * gridDim.y);
            last = n_blocks == 1 || atomicAdd(tail_count, 1) == n_blocks - 1;
            if (last) {
                if (n_blocks > 1) { *tail_count = 0; __threadfence_system(); }
                // (seq_dst: a device-resident call -- only the status words and this number go to the host's pinned block)
                __hip_atomic_store(seq_dst ? seq_dst : (unsigned *)(mirror + 7 * (size_t)gridDim.x + n_force) + 4, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// normal equations: G (+)= X^T X on the fp64 matrix cores, o (+)= X^T y
// ---------------------------------------------------------------------------------

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
// (min 4 waves/SIMD: with the default target of 8 the accumulators are shuttled between VGPRs and AGPRs around
// every MFMA group -- 64 v_accvgpr moves per 4 MFMAs)
__global__ void __launch_bounds__(256, 4)
k_gram_mfma(const double *x, const double *y, int64_t n_rows, int n_feat, int64_t ld, int rows_per_chunk, int blocks_xy,
            const int *tile_i, const int *tile_j, const int *frag_rowcol, double *gram, double *ord) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // XCD-aware order: workgroups are dealt to the 8 XCDs round-robin by their linear id, and each XCD has its own
    // L2.  All tile pairs of one row chunk go to the same XCD, back to back, so that the chunk (rows x n_feat
    // doubles, a few MB) is fetched into that L2 once and the other reads of it hit there.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int chunk = xcd + 8 * (slot / blocks_xy), pair_block = slot % blocks_xy;
    int pair = pair_block * 4 + wave;
    int ti = tile_i[pair], tj = tile_j[pair];
    if (ti < 0) return;
    int64_t r0 = (int64_t)chunk * rows_per_chunk, r1 = r0 + rows_per_chunk;
    if (r0 >= n_rows) return;
    if (r1 > n_rows) r1 = n_rows;
    int i = lane & 15, k = lane >> 4;
    int ca0 = ti * 32 + i, ca1 = ca0 + 16, cb0 = tj * 32 + i, cb1 = cb0 + 16;
    const bool va0 = ca0 < n_feat, va1 = ca1 < n_feat, vb0 = cb0 < n_feat, vb1 = cb1 < n_feat;
    // columns past n_feat are read from column 0 and zeroed by a select: the loads stay unconditional (a guarded
    // load is a branch, and four branches per step keep only one step's loads in flight)
    const double *pa0 = x + (va0 ? ca0 : 0), *pa1 = x + (va1 ? ca1 : 0), *pb0 = x + (vb0 ? cb0 : 0), *pb1 = x + (vb1 ? cb1 : 0);
    double4_t acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
    // X^T y rides along in the waves of the diagonal tile pairs: the lane already holds X[row k][column i] of its two
    // 16-column halves; the four k groups of a column meet in a cross-lane sum at the end (other waves: weight 0)
    const bool want_ord = ord && y && ti == tj;
    const double *py = want_ord ? y : x;                            // (any readable address when unused)
    const double yw = want_ord ? 1.0 : 0.0;
    double oy0 = 0.0, oy1 = 0.0;
    int64_t r = r0;
    constexpr int KU = 4;                                           // k-steps per trip: 4 * KU loads in flight
    for (; r + 4 * KU <= r1; r += 4 * KU) {
        double a0[KU], a1[KU], b0[KU], b1[KU], yv[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int64_t o = (r + 4 * u + k) * ld;
            a0[u] = pa0[o]; a1[u] = pa1[o]; b0[u] = pb0[o]; b1[u] = pb1[o]; yv[u] = py[r + 4 * u + k];
        }
        __builtin_amdgcn_sched_group_barrier(0x20, 5 * KU, 0);      // all VMEM reads first
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const double x0 = va0 ? a0[u] : 0.0, x1 = va1 ? a1[u] : 0.0, y0 = vb0 ? b0[u] : 0.0, y1 = vb1 ? b1[u] : 0.0;
            oy0 += x0 * (yv[u] * yw); oy1 += x1 * (yv[u] * yw);
            acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y1, acc11, 0, 0, 0);
        }
    }
    for (; r < r1; r += 4) {                                        // ragged end of the chunk
        const int64_t row = r + k;
        const bool vr = row < r1;
        const int64_t o0 = (vr ? row : r0) * ld;
        double a0 = pa0[o0], a1 = pa1[o0], b0 = pb0[o0], b1 = pb1[o0];
        a0 = (vr && va0) ? a0 : 0.0; a1 = (vr && va1) ? a1 : 0.0; b0 = (vr && vb0) ? b0 : 0.0; b1 = (vr && vb1) ? b1 : 0.0;
        const double yr = py[vr ? row : r0] * yw;
        oy0 += a0 * yr; oy1 += a1 * yr;
        acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc11, 0, 0, 0);
    }
    if (want_ord) {
        oy0 += __shfl_xor(oy0, 16); oy0 += __shfl_xor(oy0, 32);
        oy1 += __shfl_xor(oy1, 16); oy1 += __shfl_xor(oy1, 32);
        if (k == 0) {
            if (va0 && oy0 != 0.0) unsafeAtomicAdd(ord + ca0, oy0);
            if (va1 && oy1 != 0.0) unsafeAtomicAdd(ord + ca1, oy1);
        }
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

// X^T X, LDS-tiled.  G's upper triangle is cut into 64 x 64 patches (4 x 4 MFMA tiles = 128 accumulator registers, one
// wave each): FULL patches above the diagonal, DIAG patches on it (tiles i <= j only).  A workgroup of four waves takes up
// to four patches that touch at most four 64-column ranges (2 x 2 neighbours, four diagonal patches, or leftovers packed
// by the host: GramBlock) over a chunk of rows.  Slabs of GT_KS rows of those ranges go through LDS once per workgroup
// (coalesced 8-byte loads; the next slab is in flight in registers while the current one is multiplied), so a k-step of
// 16 MFMAs costs a wave 8 LDS operand reads and no global traffic -- against one wave per 32 x 32 tile with direct loads
// (k_gram_mfma) a quarter of the cache reads per MFMA and a third of the re-fetch between sibling tiles.  Waves of one
// workgroup carry the same kind of patch wherever the packing allows, so none idles at the slab barriers.  The
// workgroups that hold a range's diagonal patch also accumulate X^T y for it while staging.
#define GT_KS 16
#define GT_LDW 80         // slab row stride (doubles) of one 64-column range: rows k, k + 1 of an operand read fall into
                          // different halves of the banks
struct GramBlock {
    int range[4];         // 64-column ranges staged by the workgroup (unused slots repeat a used one)
    int wa[4], wb[4];     // per wave: slot of its A range, of its B range
    int kind[4];          // per wave: 0 none, 1 FULL, 2 DIAG
    int ord_mask;         // bit q: this workgroup accumulates X^T y for range[q]
    int pad[3];
};

template <int KIND>       // 1: all 16 tiles, 2: tiles (i <= j) of a diagonal patch (A and B are the same columns)
__device__ __forceinline__ void gram_slab_steps(const double *sa, const double *sb, double4_t (&acc)[4][4], int lane) {
    const int lo = (lane >> 4) * GT_LDW + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < GT_KS / 4; kk++) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = sa[lo + kk * 4 * GT_LDW + 16 * u];
#pragma unroll
        for (int u = 0; u < 4; u++) b[u] = KIND == 2 ? a[u] : sb[lo + kk * 4 * GT_LDW + 16 * u];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (KIND == 1 || j >= i) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

// SUB: the product over a SUBSET of the rows and of the columns (uf3_gram_force_rows_dev: the force rows of the atoms of one
// species, on the columns of the blocks that species takes part in).  rowmap lists the rows (seg[0]: first entry, seg[1]: how
// many -- both on the device, nobody waits for the count; the list is readable two slabs past its end), colmap the n_cols
// columns in ascending order; patches, ranges and the triangle test are in subset columns, gram / ord are written at the mapped
// ones.  n_rows is then only the bound the launch was sized for.
template <bool SUB>
__global__ void __launch_bounds__(256, 2)
k_gram_tiled(const double *x, const double *y, int64_t n_rows, int n_feat, int64_t ld, int rows_per_chunk, int blocks_per_chunk,
             const GramBlock *blocks, const int *frag_rowcol, double *gram, double *ord, const int *rowmap, const int *seg,
             const int *colmap, int n_cols) {
    __shared__ double slab[4][GT_KS * GT_LDW];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    // all workgroups of one row chunk run on the same XCD, back to back (one L2 serves their re-reads)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int chunk = xcd + 8 * (slot / blocks_per_chunk);
    const GramBlock gb = load_const(blocks + slot % blocks_per_chunk);
    const int64_t r0 = (int64_t)chunk * rows_per_chunk;
    const int *rowbase = nullptr;
    if (SUB) { n_rows = seg[1]; rowbase = rowmap + seg[0]; } else n_cols = n_feat;
    if (r0 >= n_rows) return;
    const int64_t r1 = min(n_rows, r0 + rows_per_chunk);
    const int kind = wave == 0 ? gb.kind[0] : (wave == 1 ? gb.kind[1] : (wave == 2 ? gb.kind[2] : gb.kind[3]));
    const int wa = wave == 0 ? gb.wa[0] : (wave == 1 ? gb.wa[1] : (wave == 2 ? gb.wa[2] : gb.wa[3]));
    const int wb = wave == 0 ? gb.wb[0] : (wave == 1 ? gb.wb[1] : (wave == 2 ? gb.wb[2] : gb.wb[3]));
    // staging role of this thread: column (t & 63) of every range, rows (t >> 6) + 4 j of the slab
    const int sc = t & 63, sr = t >> 6;
    // Nothing conditional inside the slab loop: at every control-flow merge the compiler copies the 64 accumulator pairs, and
    // a guarded load is a branch.  Columns past n_feat are read from column 0 and multiplied by zero on the way into LDS;
    // the loop runs over whole slabs (the prefetch of the last trip re-reads the last slab), a ragged tail follows it.
    const double *pq[4];
    double mq[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int col = gb.range[q] * 64 + sc;
        pq[q] = x + (col < n_cols ? (SUB ? colmap[col] : col) : 0);
        mq[q] = col < n_cols ? 1.0 : 0.0;
    }
    const bool want_ord = gb.ord_mask != 0 && ord && y;
    double gq[4][GT_KS / 4], yq[GT_KS / 4], ysc = 0.0, oq[4] = {0.0, 0.0, 0.0, 0.0};
    int ridx[GT_KS / 4];                                 // SUB: rows of the NEXT fetch (their indices arrive a slab ahead of the data)
    if (SUB) {
#pragma unroll
        for (int j = 0; j < GT_KS / 4; j++) ridx[j] = rowbase[r0 + sr + 4 * j];
    }
    // (X^T y is accumulated here, when the prefetched values are consumed anyway -- at fetch time it would wait for the
    // loads before the slab's MFMAs instead of after them)
    auto store = [&](bool with_ord) {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int j = 0; j < GT_KS / 4; j++) {
                slab[q][(sr + 4 * j) * GT_LDW + sc] = gq[q][j] * mq[q];
                if (with_ord) oq[q] += gq[q][j] * (yq[j] * ysc);
            }
    };
    double4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = double4_t{0, 0, 0, 0};
    const double *sa = slab[wa], *sb = slab[wb];
    const int n_full = (int)((r1 - r0) / GT_KS);
    const int64_t r_tail = r0 + (int64_t)n_full * GT_KS;
    // one loop per (patch kind, with / without X^T y)
    auto run = [&](auto kind_c, auto ord_c) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr bool ORD = decltype(ord_c)::value;
        auto fetch = [&](int64_t rs, double yscale) {
            ysc = yscale;
#pragma unroll
            for (int j = 0; j < GT_KS / 4; j++) {
                const int64_t row = SUB ? (int64_t)ridx[j] : rs + sr + 4 * j;
                if (ORD) yq[j] = y[row];
#pragma unroll
                for (int q = 0; q < 4; q++) gq[q][j] = pq[q][row * ld];
            }
            if (SUB) {
#pragma unroll
                for (int j = 0; j < GT_KS / 4; j++) ridx[j] = rowbase[rs + GT_KS + sr + 4 * j];
            }
        };
        if (n_full > 0) {
            fetch(r0, 1.0);
            store(ORD);
            __syncthreads();
            for (int s = 0; s < n_full; s++) {
                const bool last = s + 1 >= n_full;
                fetch(r0 + (int64_t)(last ? s : s + 1) * GT_KS, last ? 0.0 : 1.0);
                if (KIND) gram_slab_steps<KIND ? KIND : 1>(sa, sb, acc, lane);
                __syncthreads();
                store(ORD);
                __syncthreads();
            }
        }
        if (r_tail < r1) {                          // ragged end of the last chunk: rows past it are zeros
#pragma unroll
            for (int j = 0; j < GT_KS / 4; j++) {
                const int64_t lrow = r_tail + sr + 4 * j;
#pragma unroll
                for (int q = 0; q < 4; q++) gq[q][j] = 0.0;
                if (lrow < r1) {
                    const int64_t row = SUB ? (int64_t)rowbase[lrow] : lrow;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        gq[q][j] = pq[q][row * ld];
                        if (ORD) oq[q] += gq[q][j] * y[row];
                    }
                }
            }
            store(false);
            __syncthreads();
            if (KIND) gram_slab_steps<KIND ? KIND : 1>(sa, sb, acc, lane);
        }
    };
    using std::integral_constant;
    if (want_ord) {
        if (kind == 1) run(integral_constant<int, 1>{}, integral_constant<bool, true>{});
        else if (kind == 2) run(integral_constant<int, 2>{}, integral_constant<bool, true>{});
        else run(integral_constant<int, 0>{}, integral_constant<bool, true>{});
    } else {
        if (kind == 1) run(integral_constant<int, 1>{}, integral_constant<bool, false>{});
        else if (kind == 2) run(integral_constant<int, 2>{}, integral_constant<bool, false>{});
        else run(integral_constant<int, 0>{}, integral_constant<bool, false>{});
    }
    if (want_ord) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int col = gb.range[q] * 64 + sc;
            if (((gb.ord_mask >> q) & 1) && col < n_cols && oq[q] != 0.0) unsafeAtomicAdd(ord + (SUB ? colmap[col] : col), oq[q]);
        }
    }
    if (kind == 0) return;
    const int a0 = (wa == 0 ? gb.range[0] : (wa == 1 ? gb.range[1] : (wa == 2 ? gb.range[2] : gb.range[3]))) * 64;
    const int b0 = (wb == 0 ? gb.range[0] : (wb == 1 ? gb.range[1] : (wb == 2 ? gb.range[2] : gb.range[3]))) * 64;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (kind == 2 && j < i) continue;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int gi = a0 + 16 * i + frag_rowcol[(lane * 4 + v) * 2];
                const int gj = b0 + 16 * j + frag_rowcol[(lane * 4 + v) * 2 + 1];
                const double val = acc[i][j][v];
                if (gi < n_cols && gj < n_cols && gj >= gi && val != 0.0) {
                    const int mi = SUB ? colmap[gi] : gi, mj = SUB ? colmap[gj] : gj;
                    unsafeAtomicAdd(gram + (size_t)mi * n_feat + mj, val);
                }
            }
        }
}

// Force rows by species (uf3_gram_force_rows_dev).  seg[2 s] / seg[2 s + 1]: first entry and number of entries of species s
// in rows[]; cursor[s]: entries handed out so far.  Two launches: counts (what = 0), then the lists (what = 1; the blocks
// work the segment starts out of the counts themselves).  The three rows of an atom stay adjacent; the order of the atoms within
// a segment is whatever the atomics make it (a sum over the rows does not care).
#define SR_ATOMS 4096      // atoms per workgroup (one global atomic per workgroup, species and pass: ~300 workgroups per million atoms)
__global__ void __launch_bounds__(256)
k_species_rows(const BasisDev *B, const int32_t *z, int64_t n_atoms, int what, int *seg, int *cursor, int *rows) {
    __shared__ int cnt[UF3_MAX_SPECIES], base[UF3_MAX_SPECIES];
    const int tid = threadIdx.x;
    const int64_t a0 = (int64_t)blockIdx.x * SR_ATOMS;
    if (tid < UF3_MAX_SPECIES) cnt[tid] = 0;
    __syncthreads();
    auto species_of = [&](int64_t m) {
        if (m >= n_atoms) return -1;
        const int zz = z[m];
        return (zz >= 0 && zz < 120) ? (int)B->z2s[zz] : -1;
    };
    for (int it = 0; it < SR_ATOMS / 256; it++) {
        const int s = species_of(a0 + it * 256 + tid);
        if (s >= 0) atomicAdd(&cnt[s], 1);
    }
    __syncthreads();
    if (what == 0) {
        if (tid < UF3_MAX_SPECIES && cnt[tid]) atomicAdd(&seg[2 * tid + 1], 3 * cnt[tid]);
        return;
    }
    if (tid < UF3_MAX_SPECIES) {
        int start = 0;
        for (int q = 0; q < tid; q++) start += seg[2 * q + 1];
        if (blockIdx.x == 0) seg[2 * tid] = start;
        base[tid] = start + (cnt[tid] ? atomicAdd(&cursor[tid], 3 * cnt[tid]) : 0);
        cnt[tid] = 0;
    }
    __syncthreads();
    for (int it = 0; it < SR_ATOMS / 256; it++) {
        const int64_t m = a0 + it * 256 + tid;
        const int s = species_of(m);
        if (s >= 0) {
            int *o = rows + base[s] + 3 * atomicAdd(&cnt[s], 1);
            o[0] = (int)(3 * m); o[1] = (int)(3 * m + 1); o[2] = (int)(3 * m + 2);
        }
    }
}

// X^T X for narrow matrices (F <= 80: one-element bases, config 4's F = 73), slabs through LDS.  The direct kernel gives every
// 32 x 32 tile pair its own wave, which reads its 64 columns of every row itself: at F = 73 six waves read the rows six times
// (1.5 ms per 3.84 M rows, the L2 carrying it).  Here a workgroup of four waves takes a chunk of rows, brings slabs of 32 rows
// x F columns into LDS once (coalesced, the next slab in flight in registers while the current one is multiplied) and its
// waves share the <= 15 tile pairs (16 x 16 tiles, upper triangle) of the WHOLE Gram matrix: every row is read once from HBM.
// X^T y rides along in the staging threads.
#define GS_ROWS 32
#define GS_LDW 80
template <int NP>         // tile pairs of this wave (1 .. 4): nothing conditional inside the slab loop
__device__ __forceinline__ void gram_small_steps(const double *slab, int lane, const int (&oa)[4], const int (&ob)[4], double4_t (&acc)[4]) {
    const double *rowp = slab + (lane >> 4) * GS_LDW + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < GS_ROWS / 4; kk++) {
        double a[NP], b[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) { a[q] = rowp[kk * 4 * GS_LDW + oa[q]]; b[q] = rowp[kk * 4 * GS_LDW + ob[q]]; }
#pragma unroll
        for (int q = 0; q < NP; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc[q], 0, 0, 0);
    }
}

__global__ void __launch_bounds__(256, 2)
k_gram_small(const double *x, const double *y, int64_t n_rows, int n_feat, int64_t ld, int64_t rows_per_block,
             const int *frag_rowcol, double *gram, double *ord) {
    __shared__ double slab[GS_ROWS * GS_LDW];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n_rows, r0 + rows_per_block);
    if (r0 >= n_rows) return;
    const int nt = (n_feat + 15) >> 4, n_pairs = nt * (nt + 1) / 2;          // tiles per side, tile pairs (ti <= tj)
    // this wave's tile pairs: wave, wave + 4, ... (at most four of the fifteen); column offsets of their operands in a slab row
    int oa[4], ob[4], np_w = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int p = wave + 4 * q, a = 0;
        oa[q] = ob[q] = 0;
        if (p < n_pairs) {
            while (p >= nt - a) { p -= nt - a; ++a; }                       // row a of the upper triangle holds nt - a pairs
            oa[q] = 16 * a; ob[q] = 16 * (a + p);
            np_w = q + 1;
        }
    }
    // staging role: column sc of rows sr, sr + 2, ... of the slab (128 column slots, the first GS_LDW in use); columns past
    // n_feat are read from column 0 and multiplied by zero, the loads of whole slabs are unconditional
    const int sc = t & 127, sr = t >> 7;
    const bool col_ok = sc < n_feat, stage = sc < GS_LDW;
    const double *px = x + (col_ok ? sc : 0);
    const double mq = col_ok ? 1.0 : 0.0;
    const bool want_ord = ord && y;
    const double *py = want_ord ? y : x;
    const double yw = want_ord ? mq : 0.0;
    double4_t acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = double4_t{0, 0, 0, 0};
    double oy = 0.0, nxt[GS_ROWS / 2], ynx[GS_ROWS / 2];
    const int n_full = (int)((r1 - r0) / GS_ROWS);
    auto fetch = [&](int64_t rs) {
#pragma unroll
        for (int j = 0; j < GS_ROWS / 2; j++) { const int64_t row = rs + sr + 2 * j; nxt[j] = px[row * ld]; ynx[j] = py[row]; }
    };
    auto store = [&](double ysc) {
        if (stage) {
#pragma unroll
            for (int j = 0; j < GS_ROWS / 2; j++) { slab[(sr + 2 * j) * GS_LDW + sc] = nxt[j] * mq; oy += nxt[j] * (ynx[j] * ysc); }
        }
    };
    auto steps = [&]() {
        if (np_w == 4) gram_small_steps<4>(slab, lane, oa, ob, acc);
        else if (np_w == 3) gram_small_steps<3>(slab, lane, oa, ob, acc);
        else if (np_w == 2) gram_small_steps<2>(slab, lane, oa, ob, acc);
        else if (np_w == 1) gram_small_steps<1>(slab, lane, oa, ob, acc);
    };
    if (n_full > 0) {
        fetch(r0);
        for (int sidx = 0; sidx < n_full; sidx++) {
            __syncthreads();                                                // (the previous slab has been read)
            store(yw);
            __syncthreads();
            // (the prefetch of the last trip re-reads the last slab: nothing conditional around the loads)
            fetch(r0 + (int64_t)(sidx + 1 < n_full ? sidx + 1 : sidx) * GS_ROWS);
            steps();
        }
    }
    const int64_t r_tail = r0 + (int64_t)n_full * GS_ROWS;
    if (r_tail < r1) {                                                      // ragged end of the chunk: rows past it are zeros
#pragma unroll
        for (int j = 0; j < GS_ROWS / 2; j++) {
            const int64_t row = r_tail + sr + 2 * j;
            nxt[j] = 0.0; ynx[j] = 0.0;
            if (row < r1) { nxt[j] = px[row * ld]; ynx[j] = py[row]; }
        }
        __syncthreads();
        store(yw);
        __syncthreads();
        steps();
    }
    if (want_ord && col_ok && oy != 0.0) unsafeAtomicAdd(ord + sc, oy);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (q >= np_w) break;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int gi = oa[q] + frag_rowcol[(lane * 4 + v) * 2], gj = ob[q] + frag_rowcol[(lane * 4 + v) * 2 + 1];
            const double val = acc[q][v];
            if (gi < n_feat && gj < n_feat && gj >= gi && val != 0.0) unsafeAtomicAdd(gram + (size_t)gi * n_feat + gj, val);
        }
    }
}

__global__ void k_gram_mirror(double *gram, int n_feat) {
    int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j < n_feat && j < i) gram[(size_t)i * n_feat + j] = gram[(size_t)j * n_feat + i];
}


// ---------------------------------------------------------------------------------
// fit bookkeeping around the Gram pieces (uf3_fit_rows_dev / uf3_fit_pack_dev)
// ---------------------------------------------------------------------------------
// block 0 .. n_frames - 1: one energy row each -- per-atom normalisation in place, the frozen energy of the frame; the
// blocks behind them: chunks of the force targets.  Partial sums meet in moments[] through fp64 atomics.
__global__ void __launch_bounds__(256)
k_fit_rows(int n_frames, int n_feat, double *x_e, const double *counts, const double *y_e, const double *y_f, int64_t n_y_f,
           const int64_t *frozen, const double *c_frozen, int n_frozen, double *moments) {
    __shared__ double part[2][256];
    const int tid = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    if ((int)blockIdx.x < n_frames) {
        const int f = blockIdx.x;
        double *row = x_e + (size_t)f * n_feat;
        const double n_at = counts[f];
        for (int q = tid; q < n_feat; q += 256) row[q] = row[q] / n_at;
        __syncthreads();
        double dot = 0.0;
        for (int q = tid; q < n_frozen; q += 256) dot += row[frozen[q]] * c_frozen[q];
        part[0][tid] = dot;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) { if (tid < w) part[0][tid] += part[0][tid + w]; __syncthreads(); }
        if (tid == 0) { const double yf = y_e[f] - part[0][0]; unsafeAtomicAdd(moments + 1, yf); unsafeAtomicAdd(moments + 2, yf * yf); }
        return;
    }
    if (!y_f) return;
    const int64_t chunk = (int64_t)(blockIdx.x - n_frames) * 256 * 16;
    for (int u = 0; u < 16; u++) {
        const int64_t q = chunk + (int64_t)u * 256 + tid;
        if (q < n_y_f) { const double v = y_f[q]; s1 += v; s2 += v * v; }
    }
    part[0][tid] = s1; part[1][tid] = s2;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) { part[0][tid] += part[0][tid + w]; part[1][tid] += part[1][tid + w]; }
        __syncthreads();
    }
    if (tid == 0) { unsafeAtomicAdd(moments + 4, part[0][0]); unsafeAtomicAdd(moments + 5, part[1][0]); }
}

// packed [G_e | G_f | o_e | o_f | m_e | m_f] over the unfrozen columns; blockIdx.y = 0 energy pieces, 1 force pieces;
// blockIdx.x = row of the packed Gram (the thread block copies the row and folds the frozen columns into the ordinate)
__global__ void __launch_bounds__(256)
k_fit_pack(int n_feat, const double *flat, const int64_t *keep, int n_keep, const int64_t *frozen, const double *c_frozen,
           int n_frozen, double n_e, double n_f, double *packed) {
    __shared__ double part[256];
    const int tid = threadIdx.x, which = blockIdx.y, i = blockIdx.x;
    const size_t F = (size_t)n_feat, K = (size_t)n_keep;
    const double *gram = flat + which * F * F, *ordn = flat + 2 * F * F + which * F;
    double *g_out = packed + which * K * K, *o_out = packed + 2 * K * K + which * K;
    if (i < n_keep) {                                 // (n_keep == 0 -- every column frozen -- launches one block for the moments)
        const double *row = gram + (size_t)keep[i] * F;
        for (int j = tid; j < n_keep; j += 256) g_out[(size_t)i * K + j] = row[keep[j]];
        double dot = 0.0;
        for (int q = tid; q < n_frozen; q += 256) dot += row[frozen[q]] * c_frozen[q];
        part[tid] = dot;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) { if (tid < w) part[tid] += part[tid + w]; __syncthreads(); }
        if (tid == 0) o_out[i] = ordn[keep[i]] - part[0];
    }
    if (i == 0 && tid < 3) {
        const double *m_in = flat + 2 * F * F + 2 * F + 3 * which;
        double *m_out = packed + 2 * K * K + 2 * K + 3 * which;
        m_out[tid] = tid == 0 ? (which == 0 ? n_e : n_f) : m_in[tid];
    }
}

// ---------------------------------------------------------------------------------
// neighbour index dump (debug / parity): unsorted tuples, the host sorts them
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_debug_pairs(const BasisDev *B, const FrameGeom *geoms, const int *frame_of, CellList cl, const double *pos,
              const signed char *spec, int natoms, long long *counts /*[P+1]*/, long long *tuples /*[cap][3]*/,
              long long cap, double *geo /*[cap][4]: d, (R_j - R_i) / d; may be null*/) {
    int m = blockIdx.x;
    if (m >= natoms) return;
    const FrameGeom g = geoms[frame_of[m]];
    const int sm = spec[m];
    double pm[3] = {pos[3 * (size_t)m], pos[3 * (size_t)m + 1], pos[3 * (size_t)m + 2]};
    for_each_candidate(g, cl, m, [&](bool ok, const SlotRec &sr, int sj, int s0, int s1, int s2) {
        if (!ok) return;
        double dx, dy, dz;
        image_delta(g, sr, s0, s1, s2, pm, dx, dy, dz);
        double d = norm3_rn(dx, dy, dz);
        int j = sr.atom;
        long long sidx = supercell_index(g, s0, s1, s2, j - g.atom_lo);
        int p = B->pair_of[sm * UF3_MAX_SPECIES + sj];
        const PairDev &pd = B->pairs[p];
        if (d > pd.rmin && d < pd.rmax) {
            atomicAdd((unsigned long long *)&counts[p], 1ULL);
            long long o = (long long)atomicAdd((unsigned long long *)&counts[B->P + 1], 1ULL);
            if (o < cap) {
                tuples[3 * o] = p; tuples[3 * o + 1] = m - g.atom_lo; tuples[3 * o + 2] = sidx;
                // distances.py:331-364: delta_r / rij, IEEE quotients (nothing else in the library divides this way)
                if (geo) { geo[4 * o] = d; geo[4 * o + 1] = dx / d; geo[4 * o + 2] = dy / d; geo[4 * o + 3] = dz / d; }
            }
        }
        if (B->T > 0 && d > B->rmin3 && d <= B->rmax3) {
            atomicAdd((unsigned long long *)&counts[B->P], 1ULL);
            long long o = (long long)atomicAdd((unsigned long long *)&counts[B->P + 1], 1ULL);
            if (o < cap) { tuples[3 * o] = B->P; tuples[3 * o + 1] = m - g.atom_lo; tuples[3 * o + 2] = sidx; }
        }
    });
}


// ---------------------------------------------------------------------------------
// dense helpers behind the module-level surfaces of uf3.representation.distances (small frames: O(n m) memory like the
// reference's cdist): the distance matrix in scipy's order of operations, and compute_direction_cosines
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_distance_matrix(const double *a, long long na, const double *b, long long nb, double *out) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= na * nb) return;
    const long long i = q / nb, j = q - i * nb;
    out[q] = norm3_rn(a[3 * i] - b[3 * j], a[3 * i + 1] - b[3 * j + 1], a[3 * i + 2] - b[3 * j + 2]);
}

// out [n_atoms][3][n_d] = ((m == j) - (m == i)) * (R_j - R_i)[c] / r_ij   (distances.py:331-364)
__global__ void __launch_bounds__(256)
k_direction_cosines(const double *sup_pos, const long long *i_where, const long long *j_where, const double *rij,
                    long long n_d, long long n_atoms, double *out) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_atoms * 3 * n_d) return;
    const long long idx = q % n_d, c = (q / n_d) % 3, m = q / (3 * n_d);
    const long long i = i_where[idx], j = j_where[idx];
    const double kron = (double)((m == j) - (m == i));
    out[q] = __dmul_rn(kron, sup_pos[3 * j + c] - sup_pos[3 * i + c]) / rij[idx];
}
