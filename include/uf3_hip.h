/*
 * uf3_hip.h -- C ABI of the MI355X-native UF3 hot path (libuf3hip.so).
 *
 * The reference (uf3 v0.4.0) is pure Python and has no FFI; the boundary it
 * offers is its object API.  Each entry point below replaces the arithmetic
 * behind one of those Python surfaces, so that the classes in uf3_amd/ (same
 * names and signatures as the reference's) are thin ctypes shims:
 *
 *   uf3_basis_create      <- uf3/representation/bspline.py:20-88   (BSplineBasis tables)
 *   uf3_featurize[_dev]   <- uf3/representation/process.py:293-506 (evaluate_configuration,
 *                            featurize_{energy,force}_{2B,3B}); distances.py:19-143,
 *                            angles.py:17-232, bspline.py:810-895
 *   uf3_gram[_dev]        <- uf3/regression/least_squares.py:716-760 (X^T X, X^T y)
 *   uf3_eval[_dev]        <- uf3/forcefield/calculator.py:156-343  (energy, forces)
 *   uf3_neighbors_debug   <- distances.py:48-69 / angles.py:289-346 index semantics
 *   uf3_pair_geometry, uf3_distance_matrix, uf3_direction_cosines
 *                         <- the free functions of distances.py:19-143, 212-235, 331-364 and angles.py:289-346
 *
 * Conventions
 *   - every function returns 0 on success, a non-zero UF3_E* code otherwise;
 *     uf3_last_error(ctx) gives the message (ctx may be NULL for create failures);
 *   - the caller owns all buffers; the library allocates only inside the opaque
 *     handles (grow-only device workspace) and frees in *_destroy;
 *   - a ctx is bound to one HIP device and one stream; calls on one ctx must be
 *     serialised by the caller (one ctx per device / per Python thread);
 *   - "_dev" entries take HBM pointers for the bulk arrays and enqueue on the ctx
 *     stream without synchronising; the plain entries take host pointers, copy in
 *     and out, and synchronise before returning.  Frame metadata (offsets, cells,
 *     pbc) is always host memory: it is a few hundred bytes per frame.
 *     uf3_featurize_dev synchronises only while a context is still learning its
 *     neighbour capacities (the first calls); afterwards the kernels' status words
 *     travel to pinned host memory behind the launches and are looked at by the next
 *     call on the context and by uf3_ctx_synchronize, which then report
 *     UF3_ESPECIES / UF3_EINVAL / UF3_ERETRY for the EARLIER call;
 *   - all floating point data is IEEE double, C-contiguous; positions in Angstrom.
 */
#ifndef UF3_HIP_H
#define UF3_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uf3_ctx uf3_ctx;
typedef struct uf3_basis uf3_basis;

enum {
    UF3_OK = 0,
    UF3_EINVAL = 1,     /* bad argument / unsupported basis */
    UF3_ESPECIES = 2,   /* a frame contains an element outside the basis (process.py:321-330) */
    UF3_EHIP = 3,       /* HIP runtime error */
    UF3_ENOMEM = 4,
    UF3_EOVERFLOW = 5,  /* internal capacity exceeded after retries */
    UF3_ERETRY = 6      /* an earlier asynchronous uf3_featurize_dev call ran with neighbour capacities that turned
                           out too small, or met atoms given far outside their periodic cell (their 3-body force
                           rows need the launches that carry the reference's image-range rule): its outputs (and
                           whatever was derived from them) are invalid; the context has adapted -- repeat the work
                           since the last uf3_ctx_synchronize */
};

/* Flat description of a BSplineBasis (host memory, copied by uf3_basis_create). */
typedef struct uf3_basis_spec {
    int32_t n_species;            /* S */
    const int32_t *species_z;     /* [S] ascending atomic numbers */
    int32_t n_pairs;              /* S(S+1)/2 pair blocks in column order */
    const int32_t *pair_z;        /* [P][2], z0 <= z1 */
    const int32_t *pair_nk;       /* [P] knots per pair (4-fold ends included) */
    const double *pair_knots;     /* concatenated */
    const double *pair_rmin;      /* [P] r_min_map */
    const double *pair_rmax;      /* [P] r_max_map */
    const int32_t *pair_col;      /* [P] first column of the block (1-body columns count) */
    int32_t lead2, trail2;        /* trimmed basis functions of every pair block */
    int32_t n_trios;              /* 0 for a 2-body basis */
    const int32_t *trio_z;        /* [T][3] centre, n1 <= n2 */
    const int32_t *trio_nk;       /* [T][3] knots of the l (ij), m (ik), n (jk) legs */
    const double *trio_knots;     /* concatenated l, m, n per trio */
    const int32_t *trio_col;      /* [T] first column of the compressed block */
    const int32_t *trio_ncol;     /* [T] compressed block width */
    const int32_t *trio_lut;      /* concatenated [L*M*N] raw bin -> column within block, or -1
                                     (symmetry fold, template mask and trims already applied) */
    int32_t n_feat;               /* F = total columns, y excluded */
    double r_cut;                 /* BSplineBasis.r_cut: image range of the reference supercell */
} uf3_basis_spec;

/* A batch of frames.  All three arrays live in HOST memory. */
typedef struct uf3_frames {
    int32_t n_frames;
    const int64_t *atom_offsets;  /* [n_frames+1], atom_offsets[0] = 0 */
    const double *cells;          /* [n_frames][3][3], rows = lattice vectors */
    const uint8_t *pbc;           /* [n_frames][3] */
} uf3_frames;

int uf3_ctx_create(int device, uf3_ctx **out);
void uf3_ctx_destroy(uf3_ctx *ctx);
/* enqueue on an existing hipStream_t, e.g. torch's current stream; NULL = HIP's null stream (torch's
 * default).  Until this is called the ctx uses a private non-blocking stream, which is NOT ordered with
 * work the caller enqueues elsewhere: callers of the _dev entries should always set their stream. */
int uf3_ctx_set_stream(uf3_ctx *ctx, void *hip_stream);
/* back to the ctx's private stream (what a temporary user of uf3_ctx_set_stream does when it is done) */
int uf3_ctx_use_own_stream(uf3_ctx *ctx);
int uf3_ctx_synchronize(uf3_ctx *ctx);
const char *uf3_last_error(const uf3_ctx *ctx);
/* Which sources this binary was compiled from: the first 16 hex digits of the sha256 over uf3_hip.hip, uf3_kernels.h,
 * uf3_feat3.h, uf3_device.h and this header, concatenated in that order (the Makefile passes it in; "unknown" for a build
 * outside it).  __graft_entry__.build() rebuilds when it differs from the tree's, smoke() prints it. */
const char *uf3_build_id(void);
/* timing of the dominant kernel: (re)start / read accumulated HIP-event time in ms and launches */
int uf3_ctx_timing_reset(uf3_ctx *ctx, int enable);
int uf3_ctx_timing_read(uf3_ctx *ctx, double *featurize_ms, int64_t *featurize_launches,
                        double *neighbor_ms, double *gram_ms, double *eval_ms);

/* MD route of the evaluator (round 5).  The reference's calculator rebuilds supercell, distances and neighbour pairs on every
 * call (uf3/forcefield/calculator.py:124-153, 183-343).  With skin > 0 (Angstrom; 0 = off, the default) the whole-batch
 * energy + force entries (uf3_eval[_virial][_dev] on a basis with 3-body terms) keep, per atom, every neighbour image within
 * r_cut + skin in device memory -- sorted by (species, reference supercell index), no geometry -- and each call filters that
 * list by the true distances of ITS positions: the surviving pairs and 3-body lists are the reference's, in a fixed order, so
 * results do not depend on when the lists were built.  They are rebuilt when the batch layout (offsets, cells, pbc), the basis
 * or a species changes, and when an atom has moved more than skin / 2 from where they were built (checked on the device in
 * every call; past 0.7 of that the next call rebuilds first, past all of it the call repeats itself on new lists).
 * uf3_ctx_md_stats: list builds | calls served from lists | calls repeated because an atom outran the skin. */
int uf3_ctx_md_skin(uf3_ctx *ctx, double skin);
int uf3_ctx_md_stats(uf3_ctx *ctx, int64_t *builds, int64_t *steps, int64_t *redone);

/*
 * The fit's accumulation with HOST arrays in (round 5; SURVEY 8b's `uf3_gram_accumulate`): what the reference does as
 * BasisFeaturizer.evaluate -> HDF5 tables -> WeightedLinearModel.fit_from_file (uf3/representation/process.py:121-291,
 * uf3/regression/least_squares.py:355-483), without the rows ever leaving the GPU and without anything but this library between
 * the caller's arrays and the normal equations.  uf3_fit_add takes one pointer per frame (positions [N][3], atomic numbers [N] as
 * int64 or int32, force targets [N][3]; total energies per frame) -- no concatenation on the caller's side --, packs chunks of
 * <= max_atoms_per_chunk atoms into pinned staging (two sets, one transfer per chunk on a copy stream beside the previous chunk's
 * kernels), featurizes, normalises the energy rows and targets per atom (least_squares.py:697-700), accumulates
 * [G_e | G_f | o_e | o_f | m_e | m_f] over all F columns on the device and returns without waiting for the GPU.  A neighbour
 * capacity that overflowed in an earlier chunk surfaces as UF3_ERETRY from a later uf3_fit_add or from uf3_fit_pack: uf3_fit_reset
 * and add everything again (capacities only grow).  uf3_fit_pack folds the frozen columns out, optionally sums the packed pieces
 * over the ranks of the context's communicator (uf3_comm_init) and copies the 2 n_keep^2 + 2 n_keep + 6 doubles to the host:
 * what WeightedLinearModel.fit_from_pieces solves.  frozen_idx / frozen_c: the columns fixed by the basis (bspline.py:577-635)
 * and their coefficients; keep: the others.
 */
typedef struct uf3_fit uf3_fit;
int uf3_fit_create(uf3_basis *basis, int with_forces, int64_t max_atoms_per_chunk /* <= 0: 320000 */, const int64_t *frozen_idx,
                   const double *frozen_c, int32_t n_frozen, uf3_fit **out);
void uf3_fit_destroy(uf3_fit *fit);
int uf3_fit_reset(uf3_fit *fit);
int uf3_fit_add(uf3_fit *fit, int32_t n_frames, const int64_t *atom_counts, const double *const *positions, const void *const *z,
                int z_is_int64, const double *cells /*[n_frames][9]*/, const uint8_t *pbc /*[n_frames][3]*/, const double *energies,
                const double *const *forces /* NULL: a fit without forces */);
int uf3_fit_pack(uf3_fit *fit, const int64_t *keep, int32_t n_keep, int allreduce, double *out_host);
int uf3_fit_info(const uf3_fit *fit, int64_t *n_chunks, double *n_energy_rows, double *n_force_rows);
/* the pieces into a device buffer of the caller's (2 F^2 + 2 F + 6 doubles; zeroed by uf3_fit_reset, not here) instead of the
 * accumulator's own: frames given as host arrays and batches already resident in HBM then add up in one place.  NULL: undo. */
int uf3_fit_use_flat(uf3_fit *fit, double *d_flat);
/* a call's first chunk holds this fraction of max_atoms_per_chunk and the following ones double it up to the limit (default
 * 0.125: the GPU starts after a short pack and later packs hide behind the previous chunk's kernels; 1: equal chunks) */
int uf3_fit_first_chunk(uf3_fit *fit, double fraction);

/* RCCL behind the C ABI (round 5; SURVEY 8b's `uf3_gram_allreduce`).  One process per GPU.  The one exchange of the path is
 * the SUM over the ranks of the packed normal-equation pieces [G_e | G_f | o_e | o_f | m_e | m_f] (and, for a decomposed
 * frame, of [forces | energy | strain derivative]); the reference returns per-chunk results to the parent process and adds them
 * there (uf3/representation/process.py:196-254, uf3/regression/least_squares.py:391-412).  Rank 0 draws an id
 * (uf3_comm_unique_id, UF3_COMM_ID_BYTES bytes), the host gets it to the other ranks by any channel it has (MPI, a file,
 * torch.distributed's store), every rank joins with uf3_comm_init -- a collective call --; uf3_allreduce_sum_f64 then sums a
 * device buffer in place over xGMI, on the context's stream, asynchronously.  librccl is opened at run time (the copy already in
 * the process, else UF3_RCCL_PATH, else the system's): the library itself links against nothing but HIP. */
#define UF3_COMM_ID_BYTES 128
int uf3_comm_unique_id(uf3_ctx *ctx, void *id);
int uf3_comm_init(uf3_ctx *ctx, int n_ranks, int rank, const void *id);
int uf3_comm_destroy(uf3_ctx *ctx);
int uf3_comm_info(const uf3_ctx *ctx, int32_t *n_ranks, int32_t *rank);     /* 0 / -1 without a communicator */
int uf3_allreduce_sum_f64(uf3_ctx *ctx, double *d_buf, int64_t n);
int uf3_gram_allreduce(uf3_ctx *ctx, double *d_packed, int64_t n);           /* the same call under SURVEY 8b's name */

int uf3_basis_create(uf3_ctx *ctx, const uf3_basis_spec *spec, uf3_basis **out);
void uf3_basis_destroy(uf3_basis *basis);
/* Diagnostics: which featurizer specialisations the basis uses.  Bit 0: one-body + pair blocks (the launch that also
 * builds the 3-body neighbour lists); bits 1..5: 3-body blocks on the generic output-stationary kernels ((symmetry images,
 * 64-column chunks) = (1,1) (1,2) (2,1) (2,2) (6,1)); bits 6..9: 3-body blocks whose window of non-trimmed bins runs on the
 * fp64 matrix cores, rows (component, l) x columns (n, m) in 16 x 16 tiles: (row tiles, column tiles) = (1,1) (1,2) (1,<=4)
 * (<=2,<=6).  Within bit 7 the 3 x 3 x <=9 windows of the reference's default trims stage grouped n windows, within bit 9
 * the wide windows run banded; both are chosen per block by uf3_basis_create.  Bit 12 (round 4): the
 * basis qualifies for k_featurize3 -- one window layout on all trios, centre legs alike, 3 x <= 9 up to 6 x <= 13 kept bins,
 * symmetric folds for equal neighbour species --, which then writes the 3-body FORCE (and with them energy) rows of every block
 * by bond factorisation on the fp64 vector units (uf3_amd/csrc/uf3_feat3.h); the launches of bits 1..9 remain for energy-only
 * calls, for batches with atoms far outside their cell and for every other basis.  Setting UF3_NO_FEAT3 in the environment
 * before uf3_basis_create switches k_featurize3 off, UF3_NO_MFMA_FEAT keeps every block on the generic kernels (used by the
 * tests to compare the three paths). */
int uf3_basis_featurizer_modes(const uf3_basis *basis, int32_t *mask);

/*
 * Feature rows of a batch of frames (y column excluded).
 *   x_e [n_frames][F]   energy rows: element counts | 2-body | 3-body      (NULL: skip)
 *   x_f [sum N][3][F]   force rows of atom a, component c at ((a*3)+c)*F   (NULL: skip)
 */
int uf3_featurize(uf3_basis *basis, const uf3_frames *frames, const double *pos /*[sumN][3]*/,
                  const int32_t *z /*[sumN]*/, double *x_e, double *x_f);
int uf3_featurize_dev(uf3_basis *basis, const uf3_frames *frames, const double *d_pos,
                      const int32_t *d_z, double *d_x_e, double *d_x_f);
/* The same with the force rows `ld` doubles apart (ld >= F; columns F .. ld of a row are left alone), the layout uf3_gram_dev
 * and uf3_gram_force_rows_dev read through their own `ld`: with ld a multiple of 16 every row starts on a 128-byte line --
 * rows of F = 434 or 1798 doubles do not, and the partial lines at the ends of a row's column segments are written twice
 * (WRITE_SIZE 1.15 x the rows at F = 434, profiles/round5_hbm_counters.json). */
int uf3_featurize_ld_dev(uf3_basis *basis, const uf3_frames *frames, const double *d_positions, const int32_t *d_z,
                         double *d_x_e, double *d_x_f, int64_t ld);

/*
 * Normal-equation pieces of a row block: gram[F][F] (+)= X^T X, ord[F] (+)= X^T y, with
 * X [n_rows][ld] row-major (first F columns used).  accumulate = 0 overwrites.
 */
int uf3_gram(uf3_ctx *ctx, const double *x, const double *y, int64_t n_rows, int32_t n_feat,
             int64_t ld, int accumulate, double *gram, double *ord);
int uf3_gram_dev(uf3_ctx *ctx, const double *d_x, const double *d_y, int64_t n_rows,
                 int32_t n_feat, int64_t ld, int accumulate, double *d_gram, double *d_ord);
/*
 * The same pieces for the FORCE rows of a featurized batch (d_x_f as uf3_featurize_dev wrote it, [3 n_atoms][ld]; d_z the
 * atomic numbers the batch was featurized with).  The rows of an atom of species s are zero outside the blocks s takes part
 * in -- in the reference's dense X^T X (least_squares.py:19-67) those zeros are multiplied like everything else --: with two
 * or more species the rows are listed by species on the device and each list is multiplied on its own columns only.  The
 * result equals uf3_gram_dev's up to the order of summation; one species, narrow matrices and small batches go there.
 */
int uf3_gram_force_rows_dev(uf3_basis *basis, const double *d_x_f, const double *d_y_f, const int32_t *d_z,
                            int64_t n_atoms, int64_t ld, int accumulate, double *d_gram, double *d_ord);

/*
 * The bookkeeping around the Gram pieces of a device-resident fit (what the reference does on the host in
 * dataframe_to_tuples, least_squares.py:666-713, freeze_columns / VarianceRecorder, :19-67, :296-304, :817-890), on the
 * context's stream, nothing synchronises:
 *   uf3_fit_rows_dev   energy rows of a batch divided by their frames' atom counts (per-atom normalisation, :697-700; in
 *                      place), and the target moments: moments[1..2] += (sum, sum of squares) of the FROZEN energies
 *                      y_e - x_e[:, frozen] . c_frozen, moments[4..5] += those of the force targets (y_f may be NULL).
 *                      moments[0] / [3] (the counts) are the caller's.  The energy rows are PACKED: row f starts at
 *                      d_x_e + f * n_feat (no leading dimension, unlike uf3_gram_dev: pass unpadded rows).
 *   uf3_fit_pack_dev   flat [G_e (F x F) | G_f (F x F) | o_e (F) | o_f (F) | m_e (3) | m_f (3)] over all F columns ->
 *                      the same layout over the n_keep unfrozen columns, frozen columns folded out on the Gram level
 *                      (o_keep -= G[keep, frozen] . c_frozen): the additive pieces one rank hands to the all-reduce.
 *                      n_keep == 0 (every column frozen) is allowed: the six moments are then the whole packed buffer.
 * keep / frozen are int64 column indices in HBM, c_frozen the frozen coefficients in HBM.
 */
int uf3_fit_rows_dev(uf3_ctx *ctx, int32_t n_frames, int32_t n_feat, double *d_x_e, const double *d_atom_counts,
                     const double *d_y_e, const double *d_y_f, int64_t n_y_f, const int64_t *d_frozen,
                     const double *d_c_frozen, int32_t n_frozen, double *d_moments /*[6]*/);
int uf3_fit_pack_dev(uf3_ctx *ctx, int32_t n_feat, const double *d_flat, const int64_t *d_keep, int32_t n_keep,
                     const int64_t *d_frozen, const double *d_c_frozen, int32_t n_frozen, double n_energy_rows,
                     double n_force_rows, double *d_packed);

/*
 * Energy and forces of a fitted model on a batch of frames.
 *   c1 [S]; c2 concatenated pair coefficient vectors (nk-4 each, all basis functions);
 *   c3 concatenated full L*M*N grids per trio (BSplineBasis.decompress_3B output).
 *   energies [n_frames]; forces [sum N][3] (NULL: energies only).
 */
int uf3_eval(uf3_basis *basis, const uf3_frames *frames, const double *pos, const int32_t *z,
             const double *c1, const double *c2, const double *c3, double *energies, double *forces);
int uf3_eval_dev(uf3_basis *basis, const uf3_frames *frames, const double *d_pos, const int32_t *d_z,
                 const double *c1, const double *c2, const double *c3 /* host */,
                 double *d_energies, double *d_forces);

/*
 * Same, plus the analytic strain derivative of the energy per frame: virials [n_frames][6] = dE/d(eps) in
 * Voigt order (xx, yy, zz, yz, xz, xy), eV; stress = virial / cell volume.  Row N3 of SURVEY section 8f: the
 * reference only offers finite differences (calculator.py:399-404).
 */
int uf3_eval_virial(uf3_basis *basis, const uf3_frames *frames, const double *pos, const int32_t *z,
                    const double *c1, const double *c2, const double *c3, double *energies, double *forces,
                    double *virials);
int uf3_eval_virial_dev(uf3_basis *basis, const uf3_frames *frames, const double *d_pos, const int32_t *d_z,
                        const double *c1, const double *c2, const double *c3 /* host */,
                        double *d_energies, double *d_forces, double *d_virials);

/*
 * The share of atoms [atom_begin, atom_end) (indices into the concatenated batch) of the same quantities: the
 * one-body, pair and centre-role triplet energies of those atoms, their force rows (the other rows of `forces`
 * are zero on return from the host variant and untouched by the _dev variant) and their share of the strain
 * derivative.  Every atom gathers its own force, so shares over disjoint ranges add up to uf3_eval_virial's
 * results: the spatial decomposition of ONE large frame over GPUs is a range per rank + one sum-reduce (SURVEY
 * section 8f row N4; not in the reference, whose calculator is single-process).  forces / virials may be NULL.
 */
int uf3_eval_atoms(uf3_basis *basis, const uf3_frames *frames, const double *pos, const int32_t *z,
                   const double *c1, const double *c2, const double *c3, int64_t atom_begin, int64_t atom_end,
                   double *energies, double *forces, double *virials);
int uf3_eval_atoms_dev(uf3_basis *basis, const uf3_frames *frames, const double *d_pos, const int32_t *d_z,
                       const double *c1, const double *c2, const double *c3 /* host */, int64_t atom_begin,
                       int64_t atom_end, double *d_energies, double *d_forces, double *d_virials);

/*
 * The share of the CENTRES [atom_begin, atom_end): their one-body, pair and centre-role triplet energies and strain
 * derivative (as above), their pair forces, and what their triplets put on EVERY atom -- each triplet is evaluated once, at
 * its centre; rows of atoms inside the range or in its halo (the atoms its 3-body lists mention) come back non-zero, all
 * others zero.  Shares over disjoint ranges add up to uf3_eval_virial's results like those of uf3_eval_atoms, at a third
 * of the triplet work: what a rank of a decomposed frame computes before the one sum-reduce of
 * [energy | strain derivative | forces] (uf3_amd/parallel.py: sharded_evaluate).  forces / virials may be NULL.
 * Both variants write EVERY row of `forces` (the _dev variant zeroes the whole array on the stream before it adds the
 * shares): the caller does not clear the buffer between calls.
 */
int uf3_eval_centres(uf3_basis *basis, const uf3_frames *frames, const double *pos, const int32_t *z,
                     const double *c1, const double *c2, const double *c3, int64_t atom_begin, int64_t atom_end,
                     double *energies, double *forces, double *virials);
int uf3_eval_centres_dev(uf3_basis *basis, const uf3_frames *frames, const double *d_pos, const int32_t *d_z,
                         const double *c1, const double *c2, const double *c3 /* host */, int64_t atom_begin,
                         int64_t atom_end, double *d_energies, double *d_forces, double *d_virials);

/*
 * Neighbour indices in the reference's supercell numbering (ghost index =
 * image_rank * N + atom, geometry.py:108-149), single frame, host buffers.
 *   pair_ij [P][pair_cap][2]  2-body (i, j) per pair block, row-major sorted; pair_count [P]
 *   n3_ij   [n3_cap][2]       3-body neighbour pairs of real centres;         n3_count [1]
 * Pass caps of 0 (and NULL arrays) to obtain the counts only.
 */
int uf3_neighbors_debug(uf3_basis *basis, const uf3_frames *frame, const double *pos, const int32_t *z,
                        int64_t *pair_count, int64_t *pair_ij, int64_t pair_cap,
                        int64_t *n3_count, int64_t *n3_ij, int64_t n3_cap);
/* The 3-body neighbour lists the LAST featurizer / evaluator call on the basis' context built and consumed (the product path's
 * own lists, not the separate walk behind uf3_neighbors_debug): counts [natoms] and, per atom, the reference supercell index
 * (image_rank * N + atom) of every entry in list order, sidx [natoms][sidx_cap] with sidx_cap >= *cap_out (pass sidx = NULL to
 * learn the capacity first).  Test infrastructure (reference: uf3/representation/angles.py:289-346 identify_ij): valid only
 * straight after a synchronised single-frame call and before anything else runs on the context. */
int uf3_n3_lists_debug(uf3_basis *basis, int64_t natoms, int64_t *cap_out, int32_t *counts, int32_t *sidx, int64_t sidx_cap);

/*
 * The same 2-body pairs with their geometry: pair_geo [P][pair_cap][4] = distance, then (R_j - R_i) / distance -- what
 * distances_by_interaction / derivatives_by_interaction (distances.py:19-143) select out of the dense distance matrix
 * and what compute_direction_cosines (:331-364) divides, as lists, in the order of pair_ij.  Caps of 0: counts only.
 */
int uf3_pair_geometry(uf3_basis *basis, const uf3_frames *frame, const double *pos, const int32_t *z,
                      int64_t *pair_count, int64_t *pair_ij, double *pair_geo, int64_t pair_cap);

/*
 * Dense helpers behind the module-level functions of uf3.representation.distances / angles, for frames small enough
 * for an n x m matrix (the reference's own limit).  Host buffers.
 *   uf3_distance_matrix     out [na][nb] = |a_i - b_j| in scipy cdist's order of operations (get_distance_matrix,
 *                           distances.py:212-235; identify_ij's matrix, angles.py:289-346)
 *   uf3_direction_cosines   out [n_atoms][3][n_d] = ((m == j) - (m == i)) (R_j - R_i) / r_ij (distances.py:331-364)
 */
int uf3_distance_matrix(uf3_ctx *ctx, const double *a, int64_t na, const double *b, int64_t nb, double *out);
int uf3_direction_cosines(uf3_ctx *ctx, const double *sup_pos, int64_t n_sup, const int64_t *i_where,
                          const int64_t *j_where, const double *rij, int64_t n_d, int64_t n_atoms, double *out);

#ifdef __cplusplus
}
#endif
#endif /* UF3_HIP_H */
