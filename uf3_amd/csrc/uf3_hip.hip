// uf3_hip.hip -- C ABI (include/uf3_hip.h) of the MI355X-native UF3 hot path:
// contexts, device-resident basis tables, per-call frame geometry, kernel launches.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <thread>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../../include/uf3_hip.h"
#include "uf3_kernels.h"
#include "uf3_feat3.h"
#include <chrono>
#include <dlfcn.h>

// ------------------------------------------------------------------------------ plumbing
struct Buf {
    void *p = nullptr;
    size_t cap = 0;
    bool fine = false;          // fine-grained device memory (ensure_fine): the host may store into it through the BAR
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        fine = false;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    hipError_t ensure_fine(size_t bytes) {
        if (bytes <= cap && fine) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        fine = false;
        size_t want = std::max(bytes, cap) + bytes / 4 + 256;
        hipError_t e = hipExtMallocWithFlags(&p, want, hipDeviceMallocFinegrained);
        if (e == hipSuccess) { cap = want; fine = true; }
        return e;
    }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *)p; }
};

// grow-only pinned host memory: small transfers go through it so that they are true asynchronous copies (a copy
// from / to pageable caller memory is staged by the runtime and waits)
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = std::max(bytes, (size_t)4096);
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocCoherent);   // (fine-grained: kernels read and write these blocks in flight)
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) hipHostFree(p); p = nullptr; cap = 0; }
};
#define UF3_PIN_LIMIT (512 * 1024)   // bytes: larger transfers go straight from / to the caller's memory
#define UF3_BAR_LIMIT (128 * 1024)   // bytes: largest block the host stores into device memory itself (beyond it the copy engine is quicker)

struct uf3_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;
    int lds_max = 65536;
    int n_cu = 256;
    // grow-only workspace
    Buf geoms, offsets, frame_of, atom_bin, atom_wrap, spec, key_in, key_out, val_in, val_out, sort_tmp,
        bin_start, slots, flags,
        n3_cnt, n3_int, n3_dbl, e_atom, nbr_f, coeff, stage_pos, stage_z, stage_out, stage_out2,
        gram_tiles[UF3_MAX_SPECIES + 1], gram_tij, frag, dbg,
        sp_rows, sp_seg,                // force rows by species (uf3_gram_force_rows_dev): row lists | segment starts, counts, cursors
        halo,                           // marks + index list of the halo atoms of a decomposed frame
        n3x_ent, n3x_off,               // extension lists (batches with atoms outside their cell; see N3Lists)
        bin_cnt,                        // atoms per cell-list bin (counting sort)
        part_sums,                      // per-workgroup energy / strain-derivative sums of the MD collection pass (see EvalArgs)
        f3w;                            // hand-off buffer k_feat3_w -> k_featurize3<HO>: [atoms of a slice][list capacity][S][wsz] doubles
    int n3_cap = 0, cand_cap = 0;
    size_t bin_cnt_clean = 0;        // ints of bin_cnt known to be zero (k_bin_fill leaves the counts it used at zero)
    int n3_last_cap = 0, n3_last_natoms = 0;   // layout of the 3-body lists in the workspace right now (uf3_n3_lists_debug)
    int n3x_cap = 0;                 // capacity of the extension lists (0 until a batch needed them)
    bool img_mode = false;           // a batch with atoms far outside their cell has been seen: 3-body launches with the image-range rule
    int gram_plan_np[UF3_MAX_SPECIES + 1] = {0}, gram_plan_blocks[UF3_MAX_SPECIES + 1] = {0}, gram_plan_next = 0;   // workgroup plans of k_gram_tiled held in
                                                     // gram_tiles[] (each for this many 64-column ranges; 0: none)
    int gram_direct_feat = 0;                        // tile-pair table of k_gram_mfma held in gram_tij (for this n_feat; 0: none)
    bool n3_tuned = false;           // capacity re-sized once to the lists actually seen
    bool cand_tuned = false;         // a featurizer call has completed with the current candidate capacity
    // status words of asynchronous featurizer calls: copied to pinned slots behind the launches, looked at later
    struct Pending { hipEvent_t ev = nullptr; int cap = 0, cand = 0, xcap = 0; bool has3 = false, img = true, img_launch = false, live = false; };
    enum { N_PENDING = 16 };
    Pending pending_chk[N_PENDING];
    PinBuf pin_flags;                // [N_PENDING][8] ints
    int pending_head = 0;
    // A bad verdict on an asynchronous call belongs to whoever synchronises the context next, not to whichever entry happens to
    // poll first: it is remembered here until uf3_ctx_synchronize has reported it (the entries that meet it on the way still
    // return it once, so an asynchronous loop stops early).
    int async_bad = 0;
    std::string async_msg;
    bool frag_ready = false;
    int32_t *d_stage_z = nullptr;       // species of the staged batch (tail of stage_pos)
    // small MD steps without a fetch kernel: with a large BAR the host stores positions | species straight into a (fine-grained)
    // device block -- no launch that reads the caller's pinned block, no dispatch gap behind a 4 us kernel (eval_impl, MD route)
    bool bar_ok = false;
    bool bar_tested = false;            // bar_self_test has run on this context (it decides bar_ok once, at the first small batch)
    Buf stage_bar;                      // the fine-grained block of small batches (upload_frames)
    char *stage_cur = nullptr;          // the block that holds the current host-entry batch: stage_bar or stage_pos
    size_t staged_in_dev = 0;           // bytes of positions | species the host entry has stored into stage_pos already (0: none)
    // environment switches of the featurizer's asynchronous path, re-read at the top of every featurizer call (read_f3_env)
    bool env_no_feat3 = false, env_f3_no_cap16 = false, env_f3_no_select = false, env_debug_lds = false;
    int env_f3_bps = 24;
    unsigned eval_seq = 0;              // sequence number of the last small evaluator call whose tail kernel signals through the pinned block
    bool tail_signalled = false;        // ... and whether the last eval_impl's k_frame_sum signals
    bool pin_in_busy = false;           // a kernel that reads pin_in directly has been launched and not yet waited for
    size_t pin_in_pending = 0;          // small batch: bytes of positions | species waiting in pin_in; the cell-list
                                        // stage appends the frame geometry and sends everything in ONE copy
    PinBuf pin_in, pin_geo, pin_out;    // positions + species | frame geometry + offsets | results
    PinBuf pin_eval;                    // device-resident evaluator calls: status words [4] | sequence number, written by the last kernel
    hipEvent_t pin_in_done = nullptr, pin_geo_done = nullptr;   // the copies out of pin_in / pin_geo have executed
    std::vector<double> coeff_shadow;   // host copy of the model last uploaded by uf3_eval (c1 | c2 | c3)
    Buf coeff_cw;                       // window table of that model's 3-body coefficients (k_eval<..., CW>), when cw_of is its basis
    const void *cw_of = nullptr;        // the basis the table in coeff_cw was built for (null: none -- coefficients outside the window)
    const void *coeff_dev = nullptr;    // ... and where it lives
    // MD route of the evaluator (uf3_ctx_md_skin): persistent superset lists with a skin, see k_build_sup.  Everything a step
    // needs besides the current positions lives in its own buffers -- the workspace above belongs to whichever call ran last
    struct MdState {
        bool cap_tuned = false;         // the list capacity has held in a build the host looked at
        bool verify_next = false;       // the last step was discarded: the next build waits for its own report
        bool flags_clean = false;       // the status words [1..3] are known to be zero (the last call was an MD step that set none)
        double skin = 0.0;              // 0: off
        bool valid = false;             // the lists describe (basis, offsets, cells, pbc) below
        bool stale = false;             // some atom has passed the early-warning displacement: rebuild before the next step
        const uf3_basis *basis = nullptr;
        int natoms = 0, n_frames = 0, cap = 0;
        std::vector<int64_t> offsets;
        std::vector<double> cells;
        std::vector<uint8_t> pbc;
        Buf ent, cnt, pos_ref, geo, frame_of, spec;       // geo: FrameGeom [n_frames] | atom offsets [n_frames + 1]
        Buf inbox, surv, mark;                            // see EvalArgs::md_inbox / md_surv / md_mark
        size_t inbox_zeroed = 0;                          // bytes of inbox known to hold no stamp of a future launch
        size_t geo_bytes = 0;
        long long builds = 0, steps = 0, redone = 0;
    } md;
    // RCCL communicator of this rank (uf3_comm_init): the library is opened at run time (no link dependency), see rccl_api()
    void *comm = nullptr;
    int comm_ranks = 0, comm_rank = -1;
    bool md_step = false;               // the last eval_impl ran on the persistent lists (its status words 2 / 3 are the displacement flags)
    // timing
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double t_feat = 0, t_nbr = 0, t_gram = 0, t_eval = 0;
    long long n_feat_launch = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[4];
};

struct uf3_basis {
    uf3_ctx *ctx = nullptr;
    BasisDev host;              // host mirror (device pointers inside)
    BasisDev *dev = nullptr;
    TrioDev *d_trios = nullptr;
    KnotRec *d_recs = nullptr;
    int *d_lut = nullptr;
    int *d_colsrc = nullptr;
    int *d_dsrc = nullptr;           // colsrc as offsets into the dumped dense window (MFMA specialisation)
    unsigned short *d_gsrc = nullptr; // grouped windows: fold tables (FeatArgs::gsrc)
    size_t n_gsrc = 0;
    bool all_grouped7 = false;       // every mode-7 trio stages grouped windows (its force launches do not touch dsrc)
    bool all_banded9 = false;        // every mode-9 trio runs banded (trio_block_banded) with its bands on the column tiles 0, 1, 2
    size_t n_dsrc = 0;
    int *d_sp_cols = nullptr;        // [S][F]: the columns of the blocks species s takes part in, ascending (uf3_gram_force_rows_dev)
    int sp_ncols[UF3_MAX_SPECIES] = {0};
    std::vector<int> block_bounds;   // column boundaries of interaction blocks (for column windows)
    size_t c2_len = 0, c3_len = 0, n_recs = 0;
    size_t n_pair_recs = 0;
    size_t trio_rec_lo = 0;          // first knot record a trio leg refers to
    size_t wrow_lo = 0;              // where the window rows of the grouped layouts start (they follow the knot records)
    int n_wrows = 0;                 // ... and how many there are
    int dense_stride[16] = {0};      // per featurizer mode: largest staged-record stride (doubles) among its trios
    int dense_stride_f[16] = {0};    // ... when force rows are wanted (grouped 3 x 3 x 9 windows stage 32-double records)
    bool dense_grouped[16] = {false}; // a trio of the mode stages grouped n windows (even-aligned groups: up to two padding records per pass)
    int dense_dump[16] = {0};        // ... smallest stage (doubles) the fold of its widest window needs
    int modes = 1;                   // bit m set: some trio block is handled by featurizer specialisation m
    double r_cut = 0;
    // k_featurize3 (3-body force rows by bond factorisation, uf3_feat3.h): eligibility and tables
    bool feat3_ok = false;
    bool eval_tab_ok = false;        // the evaluator's centre pass may run its TAB instances (see k_eval)
    // k_eval<..., CW>: the coefficients inside the kept-bin window of the centre legs as a table for LDS (EvalArgs::c3w)
    bool eval_cw_ok = false;         // one window on every trio, centre legs alike, table + zero run small enough
    int cw_lo = 0, cw_ext = 0, cw_dim_m = 0, cw_dim_n = 0, cw_zero = 0, cw_bytes = 0;
    std::vector<int> cw_lut_off;     // per trio: where its full grid starts in c3
    std::vector<int> cw_dim_l;
    double *d_f3rows = nullptr;      // window rows of the centre legs and of leg n
    int n_f3rows = 0;
    unsigned short *d_f3src = nullptr;   // fold tables
    int n_f3src = 0;
    int *d_f3off = nullptr;          // [T] a trio's table
    Feat3Leg f3_leg_p, f3_leg_n;
    int f3_lo_p = 0, f3_ext_p = 0, f3_lo_n = 0, f3_ext_n = 0;
    int f3_nr = 0;                   // rounds of 32 window positions of the launch that serves the window (1: default trims; 2, 3: wider)
};

// ------------------------------------------------------------------------------ environment switches
// The UF3_* switches (A/B measurements, tests) are looked at on every call -- tests flip them between two calls on one context --
// but not through ~25 getenv scans per call: the entries of the hot paths call uf3_env_refresh(), which walks `environ` once when
// its fingerprint (the entries' addresses: setenv / unsetenv change them) has moved and keeps the UF3_* entries; uf3_env() then
// answers from those few (none at all in production).
extern char **environ;
struct Uf3EnvCache { unsigned long long print = ~0ull; std::vector<std::pair<std::string, std::string>> vars; };
static thread_local Uf3EnvCache g_env;
static void uf3_env_refresh() {
    unsigned long long fp = 1469598103934665603ull;
    for (char **e = environ; e && *e; e++) fp = (fp ^ (unsigned long long)(uintptr_t)*e) * 1099511628211ull;
    if (fp == g_env.print) return;
    g_env.print = fp;
    g_env.vars.clear();
    for (char **e = environ; e && *e; e++) {
        if (std::strncmp(*e, "UF3_", 4)) continue;
        const char *eq = std::strchr(*e, '=');
        if (eq) g_env.vars.emplace_back(std::string(*e, eq - *e), std::string(eq + 1));
    }
}
static const char *uf3_env(const char *name) {
    if (g_env.print == ~0ull) uf3_env_refresh();
    for (const auto &kv : g_env.vars)
        if (kv.first == name) return kv.second.c_str();
    return nullptr;
}

// host stores into device memory through the BAR are write-combined: drain them before the launch that reads them is queued
static inline void uf3_store_fence() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_sfence();
#else
    __sync_synchronize();
#endif
}

static thread_local std::string g_err;
static int fail(uf3_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    g_err = msg;
    return code;
}
#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return fail(ctx, UF3_EHIP, std::string(#call) + ": " + hipGetErrorString(e__));       \
    } while (0)

extern "C" const char *uf3_last_error(const uf3_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }
#ifndef UF3_BUILD_ID
#define UF3_BUILD_ID "unknown"
#endif
extern "C" const char *uf3_build_id(void) { return UF3_BUILD_ID; }

// the featurizer's switches, re-read on every call (uf3_env is a look-up in the cached UF3_* entries; tests and A/B runs flip
// them between calls on one shared context -- ADVICE round 5)
static void read_f3_env(uf3_ctx *c) {
    c->env_no_feat3 = uf3_env("UF3_NO_FEAT3") != nullptr;
    c->env_f3_no_cap16 = uf3_env("UF3_F3_NO_CAP16") != nullptr;
    c->env_f3_no_select = uf3_env("UF3_F3_NO_SELECT") != nullptr;
    c->env_debug_lds = uf3_env("UF3_DEBUG_LDS") != nullptr;
    c->env_f3_bps = uf3_env("UF3_F3_BPS") ? std::max(1, atoi(uf3_env("UF3_F3_BPS"))) : 24;
}

extern "C" int uf3_ctx_create(int device, uf3_ctx **out) {
    uf3_env_refresh();
    if (!out) return fail(nullptr, UF3_EINVAL, "uf3_ctx_create: null out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, UF3_EHIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, UF3_EINVAL, "uf3_ctx_create: bad device index");
    uf3_ctx *c = new uf3_ctx();
    c->device = device;
    HIPCHK(c, hipSetDevice(device));
    HIPCHK(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIPCHK(c, hipEventCreateWithFlags(&c->pin_in_done, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->pin_geo_done, hipEventDisableTiming));
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, device));
    c->lds_max = (int)prop.sharedMemPerBlock;
    c->n_cu = prop.multiProcessorCount;
    read_f3_env(c);
    { int large = 0; c->bar_ok = hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, device) == hipSuccess && large && !uf3_env("UF3_NO_BAR_STAGE"); }
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        std::string m = std::string("uf3_hip is built for gfx950 only, device is ") + prop.gcnArchName;
        delete c;
        return fail(nullptr, UF3_EHIP, m);
    }
    *out = c;
    return UF3_OK;
}

extern "C" int uf3_comm_destroy(uf3_ctx *c);
extern "C" void uf3_ctx_destroy(uf3_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    Buf *all[] = {&c->geoms, &c->offsets, &c->frame_of, &c->atom_bin, &c->atom_wrap, &c->spec, &c->key_in,
                  &c->key_out, &c->val_in, &c->val_out, &c->sort_tmp, &c->bin_start, &c->slots, &c->flags, &c->n3_cnt, &c->n3_int, &c->n3_dbl, &c->e_atom, &c->nbr_f, &c->coeff,
                  &c->stage_pos, &c->stage_z, &c->stage_out, &c->stage_out2, &c->sp_rows, &c->sp_seg, &c->gram_tij, &c->frag, &c->dbg, &c->halo, &c->n3x_ent, &c->n3x_off,
                  &c->bin_cnt, &c->f3w, &c->coeff_cw, &c->part_sums};
    for (Buf *b : all) b->release();
    for (Buf &b : c->gram_tiles) b.release();
    if (c->comm) uf3_comm_destroy(c);
    { Buf *mdb[] = {&c->md.ent, &c->md.cnt, &c->md.pos_ref, &c->md.geo, &c->md.frame_of, &c->md.spec, &c->md.inbox, &c->md.surv, &c->md.mark}; for (Buf *b : mdb) b->release(); }
    c->stage_bar.release();
    c->pin_in.release(); c->pin_geo.release(); c->pin_out.release(); c->pin_flags.release(); c->pin_eval.release();
    for (auto &pd : c->pending_chk) if (pd.ev) hipEventDestroy(pd.ev);
    if (c->pin_in_done) hipEventDestroy(c->pin_in_done);
    if (c->pin_geo_done) hipEventDestroy(c->pin_geo_done);
    for (auto &v : c->pending) for (auto &p : v) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" int uf3_ctx_set_stream(uf3_ctx *c, void *s) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    c->stream = (hipStream_t)s;     // NULL is HIP's null stream (what torch's default stream is)
    return UF3_OK;
}

extern "C" int uf3_ctx_use_own_stream(uf3_ctx *c) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    c->stream = c->own_stream;
    return UF3_OK;
}

static int check_flags(uf3_ctx *c);
static int poll_pending(uf3_ctx *c, bool wait, bool remember = true);

extern "C" int uf3_ctx_synchronize(uf3_ctx *c) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    poll_pending(c, true);                           // (whatever it finds is remembered in async_bad)
    if (c->async_bad) {
        const int rc = c->async_bad;
        const std::string msg = c->async_msg;
        c->async_bad = 0; c->async_msg.clear();
        return fail(c, rc, msg);
    }
    return check_flags(c);
}

// what the synchronous host entries do at their end: wait for the stream and look at the verdict of THEIR OWN call (verdicts on
// earlier asynchronous calls were drained into async_bad before the call started and stay there for uf3_ctx_synchronize)
static int sync_own_call(uf3_ctx *c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int rc = poll_pending(c, true, false);
    if (rc) return rc;
    return check_flags(c);
}

// ---- timing of kernel classes with HIP events on the launch stream ----------------------
enum { T_FEAT = 0, T_NBR = 1, T_GRAM = 2, T_EVAL = 3 };
struct Timed {
    uf3_ctx *c; int cls; hipEvent_t a = nullptr, b = nullptr;
    Timed(uf3_ctx *c_, int cls_) : c(c_), cls(cls_) {
        if (c->timing) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, c->stream); }
    }
    ~Timed() {
        if (c->timing) { hipEventRecord(b, c->stream); c->pending[cls].push_back({a, b}); if (cls == T_FEAT) c->n_feat_launch++; }
    }
};

extern "C" int uf3_ctx_timing_reset(uf3_ctx *c, int enable) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    hipStreamSynchronize(c->stream);
    for (auto &v : c->pending) { for (auto &p : v) { hipEventDestroy(p.first); hipEventDestroy(p.second); } v.clear(); }
    c->t_feat = c->t_nbr = c->t_gram = c->t_eval = 0;
    c->n_feat_launch = 0;
    c->timing = enable != 0;
    return UF3_OK;
}

extern "C" int uf3_ctx_timing_read(uf3_ctx *c, double *feat_ms, int64_t *feat_launches, double *nbr_ms,
                                   double *gram_ms, double *eval_ms) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double *acc[4] = {&c->t_feat, &c->t_nbr, &c->t_gram, &c->t_eval};
    for (int k = 0; k < 4; k++) {
        for (auto &p : c->pending[k]) {
            float ms = 0;
            hipEventElapsedTime(&ms, p.first, p.second);
            *acc[k] += ms;
            hipEventDestroy(p.first); hipEventDestroy(p.second);
        }
        c->pending[k].clear();
    }
    if (feat_ms) *feat_ms = c->t_feat;
    if (feat_launches) *feat_launches = c->n_feat_launch;
    if (nbr_ms) *nbr_ms = c->t_nbr;
    if (gram_ms) *gram_ms = c->t_gram;
    if (eval_ms) *eval_ms = c->t_eval;
    return UF3_OK;
}

// ------------------------------------------------------------------------------ basis
// Interval records of one knot sequence; identical sequences (common: the same settings on every pair / trio)
// share one run of the table, which keeps it small enough to live in LDS.
static void fill_leg(LegDev &leg, const double *t, int nk, std::vector<KnotRec> &recs) {
    leg.nk = nk;
    leg.t0 = t[0];
    leg.tlast = t[nk - 1];
    leg.inv_h = (leg.tlast > leg.t0) ? (double)(nk - 7) / (leg.tlast - leg.t0) : 0.0;
    std::vector<KnotRec> mine;
    for (int i = 0; i < nk - 1; i++) {
        KnotRec r;
        std::memset(&r, 0, sizeof(r));
        if (i >= 3 && i <= nk - 5) {
            for (int q = 0; q < 6; q++) r.t[q] = t[i - 2 + q];
            auto rc = [](double d) { return d > 0 ? 1.0 / d : 0.0; };
            r.r[0] = rc(t[i + 1] - t[i]);
            r.r[1] = rc(t[i + 1] - t[i - 1]); r.r[2] = rc(t[i + 2] - t[i]);
            r.r[3] = rc(t[i + 1] - t[i - 2]); r.r[4] = rc(t[i + 2] - t[i - 1]); r.r[5] = rc(t[i + 3] - t[i]);
        }
        mine.push_back(r);
    }
    for (size_t off = 0; off + mine.size() <= recs.size(); off++)
        if (std::memcmp(&recs[off], mine.data(), sizeof(KnotRec) * mine.size()) == 0) { leg.rec_off = (int)off; return; }
    leg.rec_off = (int)recs.size();
    recs.insert(recs.end(), mine.begin(), mine.end());
}

// Coefficients, in powers of u = x - t[i], of the cubic B-spline basis function j on the knot interval (t[i], t[i+1]]:
// Cox - de Boor with polynomials instead of numbers (long double; 0 / 0 := 0 at repeated knots).  Zero when the interval is
// outside the function's support.
static void bspline_piece(const double *t, int j, int i, double c[4]) {
    long double cur[4][4];
    for (int k = 0; k < 4; k++) for (int e = 0; e < 4; e++) cur[k][e] = (e == 0 && j + k == i) ? 1.0L : 0.0L;
    for (int d = 1; d <= 3; d++)
        for (int k = 0; k + d <= 3; k++) {
            const int q = j + k;
            const long double den1 = (long double)t[q + d] - t[q], den2 = (long double)t[q + d + 1] - t[q + 1];
            long double out[4] = {0, 0, 0, 0};
            if (den1 != 0) {
                const long double a0 = ((long double)t[i] - t[q]) / den1, a1 = 1.0L / den1;
                for (int e = 0; e < 4; e++) { out[e] += a0 * cur[k][e]; if (e < 3) out[e + 1] += a1 * cur[k][e]; }
            }
            if (den2 != 0) {
                const long double b0 = ((long double)t[q + d + 1] - t[i]) / den2, b1 = -1.0L / den2;
                for (int e = 0; e < 4; e++) { out[e] += b0 * cur[k + 1][e]; if (e < 3) out[e + 1] += b1 * cur[k + 1][e]; }
            }
            for (int e = 0; e < 4; e++) cur[k][e] = out[e];
        }
    for (int e = 0; e < 4; e++) c[e] = (double)cur[0][e];
}

// Range tests on the squared distance.  The kernels form s = (dx dx + dy dy) + dz dz without fused operations and take the
// correctly rounded root, so that "d < r" agrees with the reference's cdist to the last bit; the root is monotonic in s, so
// the same decisions can be taken on s itself against thresholds found here, and the root is left to the few candidates kept.
static double sq_root_le(double r) {          // the largest double s with sqrt(s) <= r   (r >= 0)
    double s = r * r;
    while (std::sqrt(s) <= r) s = std::nextafter(s, HUGE_VAL);
    while (s > 0 && std::sqrt(s) > r) s = std::nextafter(s, -HUGE_VAL);
    return s;
}
static double sq_root_ge(double r) {          // the smallest double s with sqrt(s) >= r
    double s = r * r;
    while (s > 0 && std::sqrt(s) >= r) s = std::nextafter(s, -HUGE_VAL);
    while (std::sqrt(s) < r) s = std::nextafter(s, HUGE_VAL);
    return s;
}

extern "C" int uf3_basis_create(uf3_ctx *c, const uf3_basis_spec *s, uf3_basis **out) {
    uf3_env_refresh();
    if (!c || !s || !out) return fail(c, UF3_EINVAL, "uf3_basis_create: null argument");
    if (s->n_species < 1 || s->n_species > UF3_MAX_SPECIES)
        return fail(c, UF3_EINVAL, "uf3_basis_create: 1..8 species supported");
    if (s->n_pairs != s->n_species * (s->n_species + 1) / 2)
        return fail(c, UF3_EINVAL, "uf3_basis_create: n_pairs must be S(S+1)/2");
    HIPCHK(c, hipSetDevice(c->device));
    uf3_basis *b = new uf3_basis();
    b->ctx = c;
    b->r_cut = s->r_cut;
    BasisDev &h = b->host;
    std::memset(&h, 0, sizeof(h));
    h.S = s->n_species; h.P = s->n_pairs; h.T = s->n_trios; h.F = s->n_feat;
    h.lead2 = s->lead2; h.trail2 = s->trail2;
    std::memset(h.z2s, -1, sizeof(h.z2s));
    for (int i = 0; i < h.S; i++) {
        int z = s->species_z[i];
        if (z < 1 || z >= 120 || (i && z <= s->species_z[i - 1])) { delete b; return fail(c, UF3_EINVAL, "species_z must be ascending atomic numbers"); }
        h.z2s[z] = (signed char)i;
    }
    for (auto &v : h.pair_of) v = -1;
    for (auto &v : h.trio_of) v = -1;
    std::vector<KnotRec> recs;
    std::vector<int> bounds;
    for (int i = 0; i <= h.S; i++) bounds.push_back(i);
    const double *kp = s->pair_knots;
    h.rmax2 = 0;
    for (int p = 0; p < h.P; p++) {
        int za = s->pair_z[2 * p], zb = s->pair_z[2 * p + 1];
        int a = (za > 0 && za < 120) ? h.z2s[za] : -1, bb = (zb > 0 && zb < 120) ? h.z2s[zb] : -1;
        int nk = s->pair_nk[p];
        if (a < 0 || bb < 0 || nk < 8) { delete b; return fail(c, UF3_EINVAL, "bad pair block"); }
        h.pair_of[a * UF3_MAX_SPECIES + bb] = h.pair_of[bb * UF3_MAX_SPECIES + a] = (short)p;
        PairDev &pd = h.pairs[p];
        fill_leg(pd.leg, kp, nk, recs);
        pd.col = s->pair_col[p];
        pd.nb = nk - 4;
        pd.sa = std::min(a, bb); pd.sb = std::max(a, bb);
        pd.rmin = s->pair_rmin[p] > 0 ? s->pair_rmin[p] : 0.0;   // max(r_min, 0), distances.py:60
        pd.rmax = s->pair_rmax[p];
        pd.s_lo = sq_root_le(pd.rmin); pd.s_hi = sq_root_ge(pd.rmax);
        h.rmax2 = std::max(h.rmax2, pd.rmax);
        bounds.push_back(pd.col + pd.nb);
        b->c2_len += (size_t)pd.nb;
        kp += nk;
    }
    b->n_pair_recs = recs.size();
    std::vector<TrioDev> trios(h.T);
    std::vector<const double *> legn_knots(h.T, nullptr);        // knots of leg n of every trio
    std::vector<const double *> leg_knots(3 * (size_t)h.T, nullptr);   // ... and of every leg
    const double *tp = s->trio_knots;
    double lo3 = 1e300, hi3 = -1e300;
    size_t lut_len = 0;
    for (int t = 0; t < h.T; t++) {
        int zc = s->trio_z[3 * t], za = s->trio_z[3 * t + 1], zb = s->trio_z[3 * t + 2];
        int sc = h.z2s[zc], sa = h.z2s[za], sb = h.z2s[zb];
        if (sc < 0 || sa < 0 || sb < 0 || sa > sb) { delete b; return fail(c, UF3_EINVAL, "bad trio block"); }
        h.trio_of[(sc * UF3_MAX_SPECIES + sa) * UF3_MAX_SPECIES + sb] = (short)t;
        h.trio_of[(sc * UF3_MAX_SPECIES + sb) * UF3_MAX_SPECIES + sa] = (short)t;
        TrioDev &td = trios[t];
        for (int d = 0; d < 3; d++) {
            int nk = s->trio_nk[3 * t + d];
            if (nk < 8) { delete b; return fail(c, UF3_EINVAL, "trio knot vector too short"); }
            fill_leg(td.leg[d], tp, nk, recs);
            if (d == 2) legn_knots[t] = tp;
            leg_knots[3 * (size_t)t + d] = tp;
            for (int q = 0; q < nk; q++) {
                lo3 = std::min(lo3, tp[q]);
                if (d < 2) hi3 = std::max(hi3, tp[q]);     // angles.py:322-325: centre legs only
            }
            tp += nk;
        }
        td.dim_l = td.leg[0].nk - 4; td.dim_m = td.leg[1].nk - 4; td.dim_n = td.leg[2].nk - 4;
        td.col = s->trio_col[t];
        td.ncol = s->trio_ncol[t];
        td.sc = sc; td.sa = sa; td.sb = sb;
        td.lut_off = (int)lut_len;
        lut_len += (size_t)td.dim_l * td.dim_m * td.dim_n;
        bounds.push_back(td.col + td.ncol);
    }
    b->c3_len = lut_len;
    h.rmin3 = h.T ? std::max(lo3, 0.0) : 0.0;
    h.rmax3 = h.T ? hi3 : 0.0;
    h.s3_lo = sq_root_le(h.rmin3); h.s3_hi = sq_root_le(h.rmax3);
    h.rsearch = std::max(h.rmax2, h.rmax3);
    std::vector<int> lut(lut_len ? lut_len : 1, -1);
    for (int t = 0; t < h.T; t++) {
        size_t n = (size_t)trios[t].dim_l * trios[t].dim_m * trios[t].dim_n;
        const int32_t *src = s->trio_lut + trios[t].lut_off;
        for (size_t q = 0; q < n; q++) {
            int v = src[q];
            if (v >= trios[t].ncol) { delete b; return fail(c, UF3_EINVAL, "trio_lut entry out of range"); }
            lut[trios[t].lut_off + q] = v < 0 ? -1 : trios[t].col + v;
        }
    }
    // per column: the distinct raw bins (symmetry images) that feed it, for the output-stationary kernel
    std::vector<int> colsrc, dsrc;
    for (int t = 0; t < h.T; t++) {
        TrioDev &td = trios[t];
        if (td.dim_l > 255 || td.dim_m > 255 || td.dim_n > 255) { delete b; return fail(c, UF3_EINVAL, "3-body grid dimension > 255"); }
        std::vector<std::vector<int>> per_col(td.ncol);
        size_t n = (size_t)td.dim_l * td.dim_m * td.dim_n;
        for (size_t q = 0; q < n; q++) {
            int v = s->trio_lut[td.lut_off + q];
            if (v < 0) continue;
            int l = (int)(q / ((size_t)td.dim_m * td.dim_n)), m = (int)((q / td.dim_n) % td.dim_m), nn = (int)(q % td.dim_n);
            per_col[v].push_back(l | (m << 8) | (nn << 16));
        }
        size_t mx = 1;
        for (auto &v : per_col) mx = std::max(mx, v.size());
        td.nsrc = mx <= 1 ? 1 : (mx <= 2 ? 2 : 6);
        if (mx > 6) { delete b; return fail(c, UF3_EINVAL, "a 3-body column is fed by more than 6 raw bins"); }
        td.src_off = (int)colsrc.size();
        // bounding box of the feeding raw bins; small boxes go to the MFMA specialisation (mode 6)
        int lo[3] = {1 << 20, 1 << 20, 1 << 20}, hi[3] = {-1, -1, -1};
        for (auto &v : per_col) for (int sp : v) {
            int idx[3] = {sp & 255, (sp >> 8) & 255, (sp >> 16) & 255};
            for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], idx[a]); hi[a] = std::max(hi[a], idx[a]); }
        }
        td.dense = 0;
        for (int a = 0; a < 3; a++) { td.lo[a] = hi[a] < 0 ? 0 : lo[a]; td.ext[a] = hi[a] < 0 ? 1 : hi[a] - lo[a] + 1; }
        // rows (c, l) x columns (m, n) of the window in 16 x 16 tiles: (1,1) (1,2) (1,<=4) (<=2,<=6) have a matrix-core
        // specialisation (modes 6-9); wider windows stay on the generic kernels
        DenseLayout dl = dense_layout(td.ext[0], td.ext[1], td.ext[2]);
        const int dmode = dense_mode_for(td.ext[0], td.ext[1], td.ext[2]);
        if (dmode && !uf3_env("UF3_NO_MFMA_FEAT") && (dmode <= 7 || !uf3_env("UF3_NO_WIDE_MFMA"))) {
            td.dense = dmode;
            b->dense_stride[dmode] = std::max(b->dense_stride[dmode], dl.stride);
            b->dense_dump[dmode] = std::max(b->dense_dump[dmode], td.ext[0] * dl.cw);
        }
        td.grouped = 0; td.gthr0 = -1e300; td.gthr2 = 1e300;
        if (td.dense == 7 && td.ext[1] == 3 && td.ext[0] <= 3 && td.ext[2] >= 6 && td.ext[2] <= 9 && td.nsrc <= 2 && td.ncol <= 128 &&
            !uf3_env("UF3_NO_NGROUP")) {
            // first window bin f = interval - 3 - lo_n: f <= 1 -> group 0 (bins 0..4), f >= 4 -> group 2 (bins 4..8)
            const double *tn = legn_knots[t];
            const int nk = td.leg[2].nk, i_lo = 3, i_hi = nk - 5;
            const int i0 = td.lo[2] + 4, i2 = td.lo[2] + 7;       // last interval of group 0, first of group 2
            td.grouped = 1;
            if (i0 >= i_lo) td.gthr0 = tn[std::min(i0, i_hi) + 1];
            if (i2 <= i_hi) td.gthr2 = tn[std::max(i2, i_lo)];
        }
        // wide windows (mode 9, two row tiles): cut the intervals of leg n into at most three bands whose records touch at most
        // three neighbouring column tiles (trio_block_banded); windows that do not fit keep the dense loop
        td.banded = 0; td.band_tile[0] = td.band_tile[1] = td.band_tile[2] = 0;
        if (td.dense == 9 && dl.stride == 64 && (4 * td.ext[0] + 15) / 16 == 2 && !uf3_env("UF3_NO_BANDS")) {
            const double *tn = legn_knots[t];
            const int i_lo = 3, i_hi = td.leg[2].nk - 5, n_ct = (td.ext[1] * td.ext[2] + 15) / 16;
            auto tiles_of = [&](int i, int &t0, int &t1) {
                const int f = i - 3 - td.lo[2], n0 = std::max(f, 0), n1 = std::min(f + 3, td.ext[2] - 1);
                if (n0 > n1) return false;
                t0 = (n0 * td.ext[1]) / 16; t1 = ((n1 + 1) * td.ext[1] - 1) / 16;
                return true;
            };
            int band_first[4] = {i_lo, -1, -1, -1}, band_lo[3] = {-1, -1, -1}, nb = 0;
            bool ok = true;
            for (int i = i_lo; i <= i_hi && ok; i++) {
                int t0, t1;
                if (!tiles_of(i, t0, t1)) continue;                       // (an interval outside the window: any band)
                if (band_lo[nb] < 0) band_lo[nb] = t0;
                if (t1 - band_lo[nb] + 1 > 3) {                           // does not fit the open band: start the next one
                    if (++nb > 2) { ok = false; break; }
                    band_first[nb] = i; band_lo[nb] = t0;
                    if (t1 - t0 + 1 > 3) ok = false;
                }
            }
            if (ok) {
                td.banded = 1;
                td.gthr0 = nb >= 1 ? tn[band_first[1]] : 1e300;           // r_n <= t_i: interval < i
                td.gthr2 = nb >= 2 ? tn[band_first[2]] : 1e300;
                for (int q = 0; q < 3; q++) td.band_tile[q] = std::max(0, std::min(band_lo[q] < 0 ? 0 : band_lo[q], std::max(0, n_ct - 3)));
                b->dense_grouped[9] = true;
            }
        }
        if (td.dense) b->dense_stride_f[td.dense] = std::max(b->dense_stride_f[td.dense], td.grouped ? 34 : dl.stride);
        if (td.grouped && (td.leg[0].nk > 255 || td.leg[1].nk > 255 || td.leg[2].nk > 255 || recs.size() > 65535)) td.grouped = 0;   // (GroupedLayout packs them)
        if (td.dense && td.grouped) b->dense_grouped[td.dense] = true;
        td.thr0 = -1e300; td.thr2 = 1e300;
        if (td.dense && dense_ct(td.dense) == 2) {
            // intervals of leg n in order: tile 0 only, ..., tile 1 only (an interval i holds t_i < r <= t_{i+1})
            const double *tn = legn_knots[t];
            auto tiles_of = [&](int i) {
                const int f = i - 3 - td.lo[2], n0 = std::max(f, 0), n1 = std::min(f + 3, td.ext[2] - 1);
                if (n0 > n1) return 0;
                return ((n0 * td.ext[1]) < 16 ? 1 : 0) | (((n1 + 1) * td.ext[1] - 1) >= 16 ? 2 : 0);
            };
            const int i_lo = 3, i_hi = td.leg[2].nk - 5;
            int i0 = i_lo - 1, i2 = i_hi + 1;
            while (i0 + 1 <= i_hi && !(tiles_of(i0 + 1) & 2)) i0++;      // last interval of the leading run without tile 1
            while (i2 - 1 >= i_lo && !(tiles_of(i2 - 1) & 1)) i2--;      // first interval of the trailing run without tile 0
            if (i0 >= i_lo) td.thr0 = tn[i0 + 1];
            if (i2 <= i_hi) td.thr2 = tn[i2];
        }
        b->modes |= 1 << (td.dense ? td.dense : td.nsrc == 1 ? (td.ncol > WAVE ? 2 : 1) : (td.nsrc == 2 ? (td.ncol > WAVE ? 4 : 3) : 5));
        for (auto &v : per_col) for (int k = 0; k < td.nsrc; k++) {
            int sp = k < (int)v.size() ? v[k] : -1;
            colsrc.push_back(sp);
            // offset inside one component's dumped rows: row l (width cw), columns n-major (n * ext_m + m)
            dsrc.push_back(sp < 0 || !td.dense ? -1
                           : ((sp & 255) - td.lo[0]) * dl.cw + (((sp >> 16) & 255) - td.lo[2]) * td.ext[1] + (((sp >> 8) & 255) - td.lo[1]));
        }
    }
    // grouped windows: number the distinct window layouts and write the fold tables -- per column, for (source 0 | 1) x
    // (group 0 | 1 | 2), the double index of the source bin inside the dumped group tiles ([group][row 4 c + l][16 columns],
    // component 0) or 15, an entry no record touches
    std::vector<unsigned short> gsrc;
    {
        std::vector<int> layout_rep;                 // first trio of every layout
        for (int t = 0; t < h.T; t++) {
            TrioDev &td = trios[t];
            td.layout = -1; td.gsrc_off = 0;
            if (!td.grouped) continue;
            auto same_leg = [](const LegDev &a, const LegDev &b2) {
                return a.rec_off == b2.rec_off && a.nk == b2.nk && a.t0 == b2.t0 && a.tlast == b2.tlast && a.inv_h == b2.inv_h;
            };
            for (size_t q = 0; q < layout_rep.size() && td.layout < 0; q++) {
                const TrioDev &o = trios[layout_rep[q]];
                bool same = o.gthr0 == td.gthr0 && o.gthr2 == td.gthr2;
                for (int a = 0; a < 3; a++) same = same && o.lo[a] == td.lo[a] && o.ext[a] == td.ext[a] && same_leg(o.leg[a], td.leg[a]);
                if (same) td.layout = (int)q;
            }
            if (td.layout < 0) { td.layout = (int)layout_rep.size(); layout_rep.push_back(t); }
            if (td.layout > 254) { td.grouped = 0; td.layout = -1; continue; }     // (never in practice: the ordinary two-tile path)
            std::vector<unsigned short> mine;
            for (int col = 0; col < td.ncol; col++)
                for (int q = 0; q < 2; q++) {
                    const int sp = q < td.nsrc ? colsrc[td.src_off + col * td.nsrc + q] : -1;
                    for (int grp = 0; grp < 3; grp++) {
                        unsigned short a = 15;
                        if (sp >= 0) {
                            const int l_rel = (sp & 255) - td.lo[0], m_rel = ((sp >> 8) & 255) - td.lo[1], n_rel = ((sp >> 16) & 255) - td.lo[2];
                            const int nl = n_rel - 2 * grp;
                            if (nl >= 0 && nl < 5) a = (unsigned short)(grp * 256 + l_rel * 16 + nl * td.ext[1] + m_rel);
                        }
                        mine.push_back(a);
                    }
                }
            // (identical tables -- the same symmetry on the same layout -- are stored once: small enough for LDS)
            td.gsrc_off = -1;
            for (size_t off = 0; off + mine.size() <= gsrc.size() && td.gsrc_off < 0; off += 2)
                if (std::equal(mine.begin(), mine.end(), gsrc.begin() + off)) td.gsrc_off = (int)off;
            if (td.gsrc_off < 0) {
                td.gsrc_off = (int)gsrc.size();
                gsrc.insert(gsrc.end(), mine.begin(), mine.end());
                while (gsrc.size() & 1) gsrc.push_back(15);      // (blocks start on a 4-byte boundary)
            }
        }
        // Window rows of the grouped layouts, appended to the knot records (so they travel to LDS with them): for every leg and
        // knot interval i one row of 18 doubles [t_i, t_i+1 | 4 functions x (c0 c1 c2 c3)] -- the polynomial pieces, in powers of
        // x - t_i, of four consecutive basis functions of the leg's window (leg n: the window of the GROUP interval i belongs
        // to), starting at window slot sb: 0 for the legs l and m (three slots: the fourth function is not stored), and
        // clamp(i - 3 - window start, 0, 1) for leg n (five slots: every function alive on the interval is among the four, the
        // fifth slot is zero).  Zeros where a function is past the window or vanishes on the interval.  The staging pass
        // evaluates a record's window slots straight from the row, every slot of every record is written in every pass, and
        // nothing has to be cleared (trio_block_grouped).  TrioDev::wrow: the number of a leg's first row.
        b->wrow_lo = recs.size();
        std::vector<double> wrows;
        int n_rows = 0;
        for (size_t q = 0; q < layout_rep.size(); q++) {
            const int t = layout_rep[q];
            TrioDev &rep = trios[t];
            for (int a = 0; a < 3; a++) {
                const double *tk = leg_knots[3 * (size_t)t + a];
                const int nk = rep.leg[a].nk;
                rep.wrow[a] = n_rows;
                for (int i = 3; i <= nk - 5; i++, n_rows++) {
                    double row[18] = {0};
                    row[0] = tk[i]; row[1] = tk[i + 1];
                    int w_lo = rep.lo[a], w_ext = a == 0 ? rep.ext[0] : 3, sb = 0;
                    if (a == 2) {
                        const int f = i - 3 - rep.lo[2], grp = f <= 1 ? 0 : (f >= 4 ? 2 : 1);
                        w_lo = rep.lo[2] + 2 * grp; w_ext = std::min(5, rep.ext[2] - 2 * grp);
                        sb = std::max(0, std::min(1, f - 2 * grp));
                    }
                    for (int fq = 0; fq < 4 && sb + fq < w_ext; fq++) {
                        const int j = w_lo + sb + fq;
                        if (j >= i - 3 && j <= i && j >= 0 && j <= nk - 5) bspline_piece(tk, j, i, row + 2 + 4 * fq);
                    }
                    wrows.insert(wrows.end(), row, row + 18);
                }
            }
        }
        // (rows of nine 16-byte pairs, one behind the other: lanes that read pair k of different rows hit different banks -- 9 is
        // odd, rows 16 apart share one)
        b->n_wrows = n_rows;
        while (wrows.size() % 12) wrows.push_back(0.0);
        for (size_t q = 0; q < wrows.size(); q += 12) {
            KnotRec kr;
            std::memcpy(&kr, &wrows[q], sizeof kr);
            recs.push_back(kr);
        }
        for (auto &td : trios)
            if (td.grouped && td.layout >= 0) for (int a = 0; a < 3; a++) td.wrow[a] = trios[layout_rep[td.layout]].wrow[a];
        if (n_rows > 65535)                                                 // (GroupedLayout packs the row numbers in 16 bits)
            for (auto &td : trios) if (td.grouped) { delete b; return fail(c, UF3_EINVAL, "too many knot intervals for the grouped windows"); }
    }
    std::sort(bounds.begin(), bounds.end());
    bounds.erase(std::unique(bounds.begin(), bounds.end()), bounds.end());
    if (bounds.back() != h.F) { delete b; return fail(c, UF3_EINVAL, "column blocks do not add up to n_feat"); }
    b->block_bounds = bounds;

    b->n_recs = recs.size();
    b->trio_rec_lo = recs.size();
    for (auto &td : trios) for (int d = 0; d < 3; d++) b->trio_rec_lo = std::min(b->trio_rec_lo, (size_t)td.leg[d].rec_off);
    HIPCHK(c, hipMalloc(&b->d_recs, sizeof(KnotRec) * std::max<size_t>(1, recs.size())));
    HIPCHK(c, hipMemcpy(b->d_recs, recs.data(), sizeof(KnotRec) * recs.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&b->d_lut, sizeof(int) * lut.size()));
    HIPCHK(c, hipMemcpy(b->d_lut, lut.data(), sizeof(int) * lut.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&b->d_colsrc, sizeof(int) * std::max<size_t>(1, colsrc.size())));
    if (!colsrc.empty()) HIPCHK(c, hipMemcpy(b->d_colsrc, colsrc.data(), sizeof(int) * colsrc.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&b->d_dsrc, sizeof(int) * std::max<size_t>(1, dsrc.size())));
    if (!dsrc.empty()) HIPCHK(c, hipMemcpy(b->d_dsrc, dsrc.data(), sizeof(int) * dsrc.size(), hipMemcpyHostToDevice));
    b->n_dsrc = dsrc.size();
    b->n_gsrc = gsrc.size();
    b->all_grouped7 = true;
    for (auto &td : trios) if (td.dense == 7 && !td.grouped) b->all_grouped7 = false;
    b->all_banded9 = true;
    for (auto &td : trios)      // (the banded-only launch names its accumulator tiles at compile time: bands on the tiles 0, 1, 2)
        if (td.dense == 9 && !(td.banded && td.band_tile[0] == 0 && td.band_tile[1] == 1 && td.band_tile[2] == 2)) b->all_banded9 = false;
    HIPCHK(c, hipMalloc(&b->d_gsrc, sizeof(unsigned short) * std::max<size_t>(8, gsrc.size() + 8)));
    if (!gsrc.empty()) HIPCHK(c, hipMemcpy(b->d_gsrc, gsrc.data(), sizeof(unsigned short) * gsrc.size(), hipMemcpyHostToDevice));
    for (auto &td : trios)
        td.head = TrioHead{td.dense, td.nsrc, td.ncol, td.sc, td.sa, td.sb, td.col,
                           td.grouped ? ((td.layout + 1) | (td.gsrc_off << 8))
                                      : (td.banded ? (1 | (td.band_tile[0] << 8) | (td.band_tile[1] << 12) | (td.band_tile[2] << 16)) : 0)};
    HIPCHK(c, hipMalloc(&b->d_trios, sizeof(TrioDev) * std::max<size_t>(1, trios.size())));
    if (!trios.empty())
        HIPCHK(c, hipMemcpy(b->d_trios, trios.data(), sizeof(TrioDev) * trios.size(), hipMemcpyHostToDevice));
    h.trios = b->d_trios; h.recs = b->d_recs; h.lut = b->d_lut;
    h.pairs_uniform = !uf3_env("UF3_NO_UNIFORM_LEGS");
    for (int a = 0; a < UF3_MAX_SPECIES * UF3_MAX_SPECIES; a++) h.pair_col[a] = h.pair_of[a] >= 0 ? h.pairs[h.pair_of[a]].col : 0;
    for (int p2 = 1; p2 < h.P && h.pairs_uniform; p2++) {
        const PairDev &a = h.pairs[0], &q = h.pairs[p2];
        if (a.leg.rec_off != q.leg.rec_off || a.leg.nk != q.leg.nk || a.leg.t0 != q.leg.t0 || a.leg.tlast != q.leg.tlast ||
            a.leg.inv_h != q.leg.inv_h || a.rmin != q.rmin || a.rmax != q.rmax || a.nb != q.nb) h.pairs_uniform = 0;
    }
    h.trio_legs_uniform = h.T > 0 && !uf3_env("UF3_NO_UNIFORM_LEGS");
    for (int t = 1; t < h.T && h.trio_legs_uniform; t++) {
        for (int d = 0; d < 3; d++) {
            const LegDev &a = trios[0].leg[d], &q = trios[t].leg[d];
            if (a.rec_off != q.rec_off || a.nk != q.nk || a.t0 != q.t0 || a.tlast != q.tlast || a.inv_h != q.inv_h) h.trio_legs_uniform = 0;
        }
        if (trios[t].dim_l != trios[0].dim_l || trios[t].dim_m != trios[0].dim_m || trios[t].dim_n != trios[0].dim_n) h.trio_legs_uniform = 0;
    }
    // k_eval<TAB>: one set of 3-body legs, the trio tables in one register each, leg n's knot records in the workgroup's LDS
    b->eval_tab_ok = h.trio_legs_uniform && h.T <= WAVE && trios[0].leg[2].nk - 7 <= EVAL_TAB_KN;
    if (b->eval_tab_ok && h.T > 0) {
        const TrioDev &t0 = trios[0];
        bool ok = t0.lo[0] == t0.lo[1] && t0.ext[0] == t0.ext[1] && t0.ext[0] >= 1;
        for (int t = 0; t < h.T && ok; t++) {
            const TrioDev &td = trios[t];
            ok = td.lo[0] == t0.lo[0] && td.lo[1] == t0.lo[0] && td.ext[0] == t0.ext[0] && td.ext[1] == t0.ext[0] &&
                 td.dim_m == t0.dim_m && td.dim_n == t0.dim_n && td.dim_l == t0.dim_l;
        }
        if (ok) {
            const int ext = t0.ext[0], dn = t0.dim_n;
            const size_t table = (size_t)h.T * ext * ext * dn * 8;
            const size_t zero = ((size_t)(3 * ext + 3) * dn * 8 + 32 + 15) / 16 * 16;        // (any row offset + one row's 32 bytes)
            const size_t total = (table + zero + 1023) / 1024 * 1024;
            ok = total <= 40 * 1024 && ext == EVAL_CW_EXT;        // (UF3_EVAL_NO_CW is looked at per call, where the instance is chosen)
            if (ok) {
                b->cw_lo = t0.lo[0]; b->cw_ext = ext; b->cw_dim_m = t0.dim_m; b->cw_dim_n = dn;
                b->cw_zero = (int)table; b->cw_bytes = (int)total;
                for (int t = 0; t < h.T; t++) { b->cw_lut_off.push_back(trios[t].lut_off); b->cw_dim_l.push_back(trios[t].dim_l); }
            }
        }
        b->eval_cw_ok = ok;
    }
    // ---- k_featurize3 (uf3_feat3.h): one window layout for all trios, centre legs alike, a W window of at most 31 positions;
    // trios with two equal neighbour species must fold symmetrically in (l, m)
    {
        bool ok = h.T > 0 && h.trio_legs_uniform && !uf3_env("UF3_NO_FEAT3");
        auto same_leg = [](const LegDev &a, const LegDev &q) {
            return a.rec_off == q.rec_off && a.nk == q.nk && a.t0 == q.t0 && a.tlast == q.tlast && a.inv_h == q.inv_h;
        };
        if (ok) {
            const TrioDev &t0 = trios[0];
            // the instantiated shapes (rows of the centre-leg window, rounds of 32 positions): (3, 1) (4, 2) (5, 3) (6, 3)
            const int ep = t0.ext[0], en = t0.ext[2], np = ep * en;
            // (a round of 32 positions holds whole rows of the summed leg: 32 / ext_n of them)
            b->f3_nr = ep <= 3 ? 1 : (ep == 4 ? 2 : 3);          // (windows of one or two rows -- the reference's default resolution
                                                                  // keeps 2 x 2 x 7 bins -- run in the three-row launch)
            const int prr = en > 0 && en <= 32 ? std::min(ep, 32 / en) : 0;
            (void)np;
            const bool shape = prr > 0 && prr * b->f3_nr >= ep &&
                               ((ep >= 1 && ep <= 3 && prr * en <= 31 && en <= 9) || (ep == 4 && en <= 11) || ((ep == 5 || ep == 6) && en <= 13));
            ok = t0.lo[0] == t0.lo[1] && t0.ext[0] == t0.ext[1] && same_leg(t0.leg[0], t0.leg[1]) && shape &&
                 t0.leg[0].nk >= 8 && t0.leg[2].nk >= 8 && !(ep > 3 && uf3_env("UF3_NO_FEAT3_WIDE"));
            for (int t = 0; t < h.T && ok; t++) {
                const TrioDev &td = trios[t];
                for (int a = 0; a < 3; a++) ok = ok && td.lo[a] == t0.lo[a] && td.ext[a] == t0.ext[a];
                ok = ok && td.nsrc <= 2 && td.ncol >= 1;
                if (ok && td.sa == td.sb)               // (l, m, n) and (m, l, n) feed the same column
                    for (int col = 0; col < td.ncol && ok; col++)
                        for (int q = 0; q < td.nsrc && ok; q++) {
                            const int sp = colsrc[td.src_off + col * td.nsrc + q];
                            if (sp < 0) continue;
                            const int swapped = ((sp >> 8) & 255) | ((sp & 255) << 8) | (sp & 0xff0000);
                            bool found = false;
                            for (int q2 = 0; q2 < td.nsrc; q2++) found = found || colsrc[td.src_off + col * td.nsrc + q2] == swapped;
                            ok = found;
                        }
            }
        }
        if (ok) {
            const TrioDev &t0 = trios[0];
            b->f3_lo_p = t0.lo[0]; b->f3_ext_p = t0.ext[0]; b->f3_lo_n = t0.lo[2]; b->f3_ext_n = t0.ext[2];
            std::vector<double> rows;
            int n_rows = 0;
            for (int kind = 0; kind < 2; kind++) {
                const int a = kind == 0 ? 0 : 2;
                const double *tk = leg_knots[a];
                const int nk = t0.leg[a].nk, w_lo = t0.lo[a], w_ext = t0.ext[a];
                Feat3Leg &lg = kind == 0 ? b->f3_leg_p : b->f3_leg_n;
                lg.t0 = t0.leg[a].t0; lg.tlast = t0.leg[a].tlast; lg.inv_h = t0.leg[a].inv_h; lg.nk = nk; lg.row0 = n_rows;
                for (int i = 3; i <= nk - 5; i++, n_rows++) {
                    // four consecutive functions of the window, the first at slot clamp(i - 3 - window start, 0, ext - 4); zeros
                    // where a function lies past the window or vanishes on the interval
                    double row[18] = {0};
                    row[0] = tk[i]; row[1] = tk[i + 1];
                    const int sb = std::max(0, std::min(std::max(0, w_ext - 4), i - 3 - w_lo));
                    for (int fq = 0; fq < 4 && sb + fq < w_ext; fq++) {
                        const int j = w_lo + sb + fq;
                        if (j >= i - 3 && j <= i && j >= 0 && j <= nk - 5) bspline_piece(tk, j, i, row + 2 + 4 * fq);
                    }
                    rows.insert(rows.end(), row, row + 18);
                }
            }
            b->n_f3rows = n_rows;
            std::vector<unsigned short> fsrc;
            std::vector<int> foff(h.T, 0);
            for (int t = 0; t < h.T; t++) {
                const TrioDev &td = trios[t];
                std::vector<unsigned short> mine;
                for (int o = 0; o < 2; o++)
                    for (int col = 0; col < td.ncol; col++)
                        for (int q = 0; q < 2; q++) {
                            const int sp = q < td.nsrc ? colsrc[td.src_off + col * td.nsrc + q] : -1;
                            // (a column's missing second source: position 31 of row 0 -- never a window position in the
                            // one-round launch --, or the zero behind the dumped rows)
                            const int ps = 32 * b->f3_nr;
                            unsigned short e = (unsigned short)(b->f3_nr == 1 ? 31 : td.ext[0] * ps);
                            if (sp >= 0) {
                                const int l = (sp & 255) - td.lo[0], m = ((sp >> 8) & 255) - td.lo[1], n = ((sp >> 16) & 255) - td.lo[2];
                                const int prr = std::min(td.ext[0], 32 / td.ext[2]), j = o == 0 ? m : l;
                                e = (unsigned short)((o == 0 ? l : m) * ps + (j / prr) * 32 + (j % prr) * td.ext[2] + n);
                            }
                            mine.push_back(e);
                        }
                int at = -1;
                for (size_t off = 0; off + mine.size() <= fsrc.size() && at < 0; off += 2)
                    if (std::equal(mine.begin(), mine.end(), fsrc.begin() + off)) at = (int)off;
                if (at < 0) { at = (int)fsrc.size(); fsrc.insert(fsrc.end(), mine.begin(), mine.end()); }
                foff[t] = at;
            }
            while (fsrc.size() & 1) fsrc.push_back(31);
            ok = fsrc.size() * 2 <= 16384;
            if (ok) {
                b->n_f3src = (int)fsrc.size();
                HIPCHK(c, hipMalloc(&b->d_f3rows, sizeof(double) * rows.size()));
                HIPCHK(c, hipMemcpy(b->d_f3rows, rows.data(), sizeof(double) * rows.size(), hipMemcpyHostToDevice));
                HIPCHK(c, hipMalloc(&b->d_f3src, sizeof(unsigned short) * fsrc.size()));
                HIPCHK(c, hipMemcpy(b->d_f3src, fsrc.data(), sizeof(unsigned short) * fsrc.size(), hipMemcpyHostToDevice));
                HIPCHK(c, hipMalloc(&b->d_f3off, sizeof(int) * foff.size()));
                HIPCHK(c, hipMemcpy(b->d_f3off, foff.data(), sizeof(int) * foff.size(), hipMemcpyHostToDevice));
            }
        }
        b->feat3_ok = ok;
        if (ok) b->modes |= 1 << 12;
    }
    // the force rows of an atom of species s are zero outside the blocks s takes part in (and in the one-body columns)
    {
        std::vector<int> sp_cols((size_t)h.S * h.F, 0);
        for (int sp = 0; sp < h.S; sp++) {
            std::vector<char> in(h.F, 0);
            for (int p2 = 0; p2 < h.P; p2++)
                if (h.pairs[p2].sa == sp || h.pairs[p2].sb == sp) for (int q = 0; q < h.pairs[p2].nb; q++) in[h.pairs[p2].col + q] = 1;
            for (auto &td : trios)
                if (td.sc == sp || td.sa == sp || td.sb == sp) for (int q = 0; q < td.ncol; q++) in[td.col + q] = 1;
            int n = 0;
            for (int q = 0; q < h.F; q++) if (in[q]) sp_cols[(size_t)sp * h.F + n++] = q;
            b->sp_ncols[sp] = n;
        }
        HIPCHK(c, hipMalloc(&b->d_sp_cols, sizeof(int) * sp_cols.size()));
        HIPCHK(c, hipMemcpy(b->d_sp_cols, sp_cols.data(), sizeof(int) * sp_cols.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMalloc(&b->dev, sizeof(BasisDev)));
    HIPCHK(c, hipMemcpy(b->dev, &h, sizeof(BasisDev), hipMemcpyHostToDevice));
    *out = b;
    return UF3_OK;
}

extern "C" int uf3_basis_featurizer_modes(const uf3_basis *b, int32_t *mask) {
    if (!b || !mask) return UF3_EINVAL;
    *mask = b->modes;
    return UF3_OK;
}

extern "C" void uf3_basis_destroy(uf3_basis *b) {
    if (!b) return;
    hipSetDevice(b->ctx->device);
    hipStreamSynchronize(b->ctx->stream);
    hipFree(b->dev); hipFree(b->d_trios); hipFree(b->d_recs); hipFree(b->d_lut); hipFree(b->d_colsrc); hipFree(b->d_dsrc); hipFree(b->d_gsrc); hipFree(b->d_sp_cols);
    hipFree(b->d_f3rows); hipFree(b->d_f3src); hipFree(b->d_f3off);
    delete b;
}

// ------------------------------------------------------------------------------ frame geometry
static void cross3(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double norm3(const double *a) { return std::sqrt(dot3(a, a)); }

// image range of the reference supercell (geometry.py:54-83), on the cell as given
static void reference_factors(const double *cell, double r_cut, int *fac) {
    bool all_zero = true, zero_vec = false;
    for (int i = 0; i < 9; i++) if (cell[i] != 0.0) all_zero = false;
    for (int i = 0; i < 3; i++) if (norm3(cell + 3 * i) == 0.0) zero_vec = true;
    if (all_zero || zero_vec) { fac[0] = fac[1] = fac[2] = 1; return; }
    double n[3][3];
    cross3(cell + 3, cell + 6, n[0]); cross3(cell, cell + 6, n[1]); cross3(cell, cell + 3, n[2]);
    for (int i = 0; i < 3; i++) {
        double s = dot3(cell + 3 * i, n[i]) / dot3(n[i], n[i]);
        double p[3] = {n[i][0] * s, n[i][1] * s, n[i][2] * s};
        fac[i] = (int)std::ceil(r_cut / norm3(p));
    }
}

static bool invert3(const double *m, double *inv) {
    double c0[3], c1[3], c2[3];
    cross3(m + 3, m + 6, c0); cross3(m + 6, m, c1); cross3(m, m + 3, c2);
    double det = dot3(m, c0);
    if (!(std::fabs(det) > 1e-300)) return false;
    for (int k = 0; k < 3; k++) { inv[3 * k] = c0[k] / det; inv[3 * k + 1] = c1[k] / det; inv[3 * k + 2] = c2[k] / det; }
    return true;
}

// (extra: added to the search radius of the cell list -- the skin of the MD route's superset lists; the reference's image range
// `fac` and the window stay those of r_cut)
static int make_geom(uf3_ctx *c, const uf3_basis *b, const uf3_frames *fr, int f, FrameGeom &g, int &bin_cursor, double extra = 0.0) {
    const double *cell = fr->cells + 9 * (size_t)f;
    const uint8_t *pbc = fr->pbc + 3 * (size_t)f;
    int64_t lo = fr->atom_offsets[f], hi = fr->atom_offsets[f + 1];
    std::memset(&g, 0, sizeof(g));
    g.atom_lo = (int)lo; g.atom_hi = (int)hi;
    int n = (int)(hi - lo), n_per = 0;
    for (int k = 0; k < 3; k++) { g.per[k] = pbc[k] ? 1 : 0; n_per += g.per[k]; }
    std::memcpy(g.cell, cell, sizeof(double) * 9);
    // effective cell: non-periodic axes become unit vectors orthogonal to the periodic ones
    double eff[9];
    std::memcpy(eff, cell, sizeof(eff));
    if (n_per == 0) {
        double id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::memcpy(eff, id, sizeof(eff));
    } else if (n_per < 3) {
        int pa[3], np = 0, na[3], nn = 0;
        for (int k = 0; k < 3; k++) { if (g.per[k]) pa[np++] = k; else na[nn++] = k; }
        if (np == 2) {
            double w[3];
            cross3(eff + 3 * pa[0], eff + 3 * pa[1], w);
            double l = norm3(w);
            if (l == 0) return fail(c, UF3_EINVAL, "periodic lattice vectors are parallel");
            for (int q = 0; q < 3; q++) eff[3 * na[0] + q] = w[q] / l;
        } else {
            const double *u = eff + 3 * pa[0];
            double l = norm3(u);
            if (l == 0) return fail(c, UF3_EINVAL, "periodic lattice vector has zero length");
            double t[3] = {0, 0, 0};
            int mn = 0;
            for (int q = 1; q < 3; q++) if (std::fabs(u[q]) < std::fabs(u[mn])) mn = q;
            t[mn] = 1.0;
            double v[3], w[3];
            cross3(u, t, v);
            double lv = norm3(v);
            for (int q = 0; q < 3; q++) v[q] /= lv;
            cross3(u, v, w);
            double lw = norm3(w);
            for (int q = 0; q < 3; q++) { eff[3 * na[0] + q] = v[q]; eff[3 * na[1] + q] = w[q] / lw; }
        }
    }
    if (!invert3(eff, g.inv)) return fail(c, UF3_EINVAL, "cell is singular along a periodic direction");
    double rs = b->host.rsearch + extra;
    // bins of half the search radius (scan radius 2): 125 bins cover 15.6 r^3 instead of 27 r^3 for 27 full-size bins
    const double bin_frac = uf3_env("UF3_BIN_FRAC") ? atof(uf3_env("UF3_BIN_FRAC")) : 0.5;
    if (n_per) {
        reference_factors(cell, b->r_cut, g.fac);
        for (int k = 0; k < 3; k++) if (!g.per[k]) g.fac[k] = 0;
    }
    // perpendicular heights of the effective cell
    double nrm[3][3];
    cross3(eff + 3, eff + 6, nrm[0]); cross3(eff + 6, eff, nrm[1]); cross3(eff, eff + 3, nrm[2]);
    double vol = std::fabs(dot3(eff, nrm[0]));
    int nb_np = std::max(1, std::min(32, (int)std::cbrt((double)std::max(1, n) / 4.0)));
    long long nbins = 1;
    for (int k = 0; k < 3; k++) {
        double h = vol / norm3(nrm[k]);
        g.cnt[k] = g.per[k] ? 2 * g.fac[k] + 1 : 1;
        if (g.per[k]) {
            int nb = (int)std::floor(h / (rs * bin_frac));
            nb = std::max(1, std::min(nb, 256));
            g.nb[k] = nb;
            g.rad[k] = (int)std::ceil(rs / (h / nb) - 1e-12);
            if (g.rad[k] < 1) g.rad[k] = 1;
            g.binw[k] = 1.0 / nb;
            if (g.fac[k] > 400) return fail(c, UF3_EINVAL, "cell is too small relative to the cutoff");
            // atoms inside a window of fractional width fac + 1 - r_cut / h around the cell (>= 1: the cell itself always
            // is) are at most fac images apart whenever they are within r_cut of each other
            const double w = g.fac[k] + 1.0 - b->r_cut / h - 1e-9;
            g.win_lo[k] = 0.5 - 0.5 * w; g.win_hi[k] = 0.5 + 0.5 * w;
        } else {
            g.win_lo[k] = -1e300; g.win_hi[k] = 1e300;
            g.nb[k] = nb_np;
            g.rad[k] = 1;
            g.binw[k] = rs / h;      // fractional width of a bin of real width rs
        }
        nbins *= g.nb[k];
    }
    long long m = (long long)g.cnt[0] * g.cnt[1] * g.cnt[2] * n;
    if (m >= (1LL << 31)) return fail(c, UF3_EINVAL, "reference supercell index exceeds int32");
    g.bin_base = bin_cursor;
    if (nbins + bin_cursor >= (1LL << 30)) return fail(c, UF3_EINVAL, "too many cell-list bins");
    bin_cursor += (int)nbins;
    return UF3_OK;
}

// ------------------------------------------------------------------------------ shared pipeline
struct Prepared {
    int natoms = 0, n_frames = 0, nbins = 0;
    double max_density = 0;
    CellList cl;
    N3Lists n3;
    const FrameGeom *geoms = nullptr;
    const int *frame_of = nullptr;
    const signed char *spec = nullptr;
    const int64_t *d_offsets = nullptr;
    size_t geo_bytes = 0;       // FrameGeom [n_frames] | atom offsets [n_frames + 1] behind `geoms`, one block
    bool deferred = false;      // the list-capacity / error flags of this build have not been read yet
    bool flags_zeroed = false;  // the cell-list stage has already zeroed the n3 / candidate status words
    bool small_prepared = false; // ... by the one-workgroup kernel, which also zeroes md_build's list-length report
};

static int check_flags(uf3_ctx *c) {
    if (!c->flags.p) return UF3_OK;
    int fl[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpy(fl, c->flags.p, sizeof(fl), hipMemcpyDeviceToHost));
    if (fl[0] == 2) { hipMemset(c->flags.p, 0, sizeof(fl)); return fail(c, UF3_ESPECIES, "frame contains an element outside the basis"); }
    if (fl[0] == 1) { hipMemset(c->flags.p, 0, sizeof(fl)); return fail(c, UF3_EINVAL, "atom too far outside the periodic cell (|wrap| > 250)"); }
    return UF3_OK;
}

// Status words of earlier asynchronous featurizer calls (error flag | 3-body list length needed | candidates needed).
// wait = false: only the slots whose copy has completed are looked at.  Returns the first problem found, once.
static int poll_pending(uf3_ctx *c, bool wait, bool remember) {
    int rc = UF3_OK;
    for (int q = 0; q < uf3_ctx::N_PENDING; q++) {
        uf3_ctx::Pending &p = c->pending_chk[(c->pending_head + q) % uf3_ctx::N_PENDING];
        if (!p.live) continue;
        if (wait) { HIPCHK(c, hipEventSynchronize(p.ev)); }
        else if (hipEventQuery(p.ev) != hipSuccess) continue;
        p.live = false;
        const int *fl = (const int *)c->pin_flags.p + 8 * ((c->pending_head + q) % uf3_ctx::N_PENDING);
        if (fl[0] == 2 && !rc) rc = fail(c, UF3_ESPECIES, "frame contains an element outside the basis (earlier asynchronous call)");
        else if (fl[0] == 1 && !rc) rc = fail(c, UF3_EINVAL, "atom too far outside the periodic cell (|wrap| > 250) (earlier asynchronous call)");
        bool grown = false;
        // (with headroom: a data set ordered by increasing density would otherwise overflow chunk after chunk)
        if (p.has3 && fl[1] > p.cap) { c->n3_cap = std::max(c->n3_cap, (std::max(fl[1] + 8, p.cap + p.cap / 4) + 7) / 8 * 8); grown = true; }
        if (fl[2] > p.cand) { c->cand_cap = std::max(c->cand_cap, (std::max(fl[2] + 16, p.cand + p.cand / 4) + 7) / 8 * 8); grown = true; }
        if (fl[3] > p.xcap) { c->n3x_cap = std::max(c->n3x_cap, std::min(248, (fl[3] + 8 + 7) / 8 * 8)); grown = true; }
        if (fl[4] && !p.img) { c->img_mode = true; grown = true; }      // (its 3-body launches left without writing rows)
        else if (!fl[4] && p.img_launch) c->img_mode = false;            // back to the ordinary launches
        if (grown && !rc)
            rc = fail(c, UF3_ERETRY, "an earlier asynchronous featurizer call overflowed its neighbour capacities: its outputs are "
                                     "invalid; the capacities have been raised, repeat the work since the last synchronisation");
    }
    if (rc && c->flags.p) hipMemsetAsync(c->flags.p, 0, sizeof(int), c->stream);
    if (rc && remember && !c->async_bad) { c->async_bad = rc; c->async_msg = c->err; }
    return rc;
}

static int n3_alloc(uf3_ctx *c, int natoms, int cap, N3Lists &n3) {
    HIPCHK(c, c->n3_cnt.ensure(sizeof(int) * (size_t)natoms * (UF3_MAX_SPECIES + 2)));
    HIPCHK(c, c->n3_dbl.ensure(sizeof(N3Entry) * (size_t)natoms * cap));
    n3.cap = cap;
    c->n3_last_cap = cap; c->n3_last_natoms = natoms;
    n3.cnt = c->n3_cnt.as<int>();
    n3.spoff = n3.cnt + natoms;
    n3.ent = c->n3_dbl.as<N3Entry>();
    return UF3_OK;
}

static int n3_cap_estimate(const uf3_basis *b, double dens) {
    double r = b->host.rmax3;
    double est = dens > 0 ? 4.18879 * r * r * r * dens : 24.0;
    return std::max(16, ((int)(est * 1.8) + 8 + 7) / 8 * 8);
}

// The density estimate is generous: after the first successful build later calls pad to what that batch really
// needed (a batch that needs more trips the overflow path and grows the capacity again).
static int n3_tune(uf3_ctx *c, const N3Lists &n3, int natoms) {
    hipStream_t st = c->stream;
    int *flags = c->flags.as<int>();
    c->n3_tuned = true;
    HIPCHK(c, hipMemsetAsync(flags + 6, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_max_count, dim3(64), dim3(256), 0, st, n3.cnt, natoms, flags + 6);
    int seen = 0;
    HIPCHK(c, hipMemcpyAsync(&seen, flags + 6, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    c->n3_cap = std::max(8, (seen + 7) / 8 * 8);
    return UF3_OK;
}

// 3-body lists of the HALO of the block of atoms [lo, hi), whose own lists exist: the atoms outside the block that those
// lists mention are marked and collected on the device (c->halo: marks [natoms] | indices [natoms] | count), then built
static int build_halo_lists(uf3_basis *b, const Prepared &P, const N3Lists &n3, const double *d_pos, int lo, int hi,
                            bool marks_zeroed = false) {
    uf3_ctx *c = b->ctx;
    hipStream_t st = c->stream;
    const int natoms = P.natoms, nb_ = hi - lo;
    HIPCHK(c, c->halo.ensure(sizeof(int) * ((size_t)natoms + 4)));
    int *mark = c->halo.as<int>(), *range = mark + natoms;
    if (!marks_zeroed) HIPCHK(c, hipMemsetAsync(mark, 0, sizeof(int) * (size_t)natoms, st));
    if (nb_ <= 0 || natoms - nb_ <= 0) return UF3_OK;
    const size_t lds = (size_t)n3.cap * (8 + 32 + 16);
    hipLaunchKernelGGL(k_mark_halo, dim3((nb_ + 3) / 4), dim3(256), 0, st, n3, lo, hi, mark, range);
    // (one workgroup per atom outside the block; the unmarked ones leave at once)
    hipLaunchKernelGGL(k_build_n3, dim3(natoms - nb_), dim3(64), lds, st, b->dev, P.geoms, P.frame_of, P.cl, n3, d_pos,
                       natoms, c->flags.as<int>() + 1, 0, (const int *)mark, (const int *)range);
    HIPCHK(c, hipGetLastError());
    return UF3_OK;
}

// cell list + 3-body neighbour lists for a batch (positions / species already in HBM)
// defer_check: once the list capacity is tuned, do not wait for the build's flags; the caller reads them together
// with its results and repeats the call if the lists overflowed (all kernels are safe on clipped lists)
// n3_lo / n3_hi: build the 3-body lists only for the atoms [n3_lo, n3_hi) and for their halo (the atoms in their
// lists): what a rank of a decomposed frame needs (n3_hi < 0: every atom)
static int prepare(uf3_basis *b, const uf3_frames *fr, const double *d_pos, const int32_t *d_z, bool need_n3,
                   Prepared &P, bool defer_check = false, int64_t n3_lo = 0, int64_t n3_hi = -1, double extra_radius = 0.0) {
    uf3_ctx *c = b->ctx;
    c->md.flags_clean = false;          // (the cell-list stage and what follows it use the status words)
    if (!fr || fr->n_frames < 1 || !fr->atom_offsets || !fr->cells || !fr->pbc)
        return fail(c, UF3_EINVAL, "bad uf3_frames");
    if (fr->atom_offsets[0] != 0) return fail(c, UF3_EINVAL, "atom_offsets[0] must be 0");
    int nf = fr->n_frames;
    int64_t total = fr->atom_offsets[nf];
    if (total < 1 || total >= (1LL << 31) / 8) return fail(c, UF3_EINVAL, "batch must hold 1 .. 2^28 atoms");
    for (int f = 0; f < nf; f++)
        if (fr->atom_offsets[f + 1] < fr->atom_offsets[f]) return fail(c, UF3_EINVAL, "atom_offsets must be non-decreasing");
    int natoms = (int)total;
    std::vector<FrameGeom> geoms(nf);
    int bin_cursor = 0;
    double dens = 0;
    for (int f = 0; f < nf; f++) {
        int rc = make_geom(c, b, fr, f, geoms[f], bin_cursor, extra_radius);
        if (rc) return rc;
        int n = geoms[f].atom_hi - geoms[f].atom_lo;
        double vol = 0;
        if (geoms[f].per[0] && geoms[f].per[1] && geoms[f].per[2]) {
            double cr[3];
            cross3(geoms[f].cell + 3, geoms[f].cell + 6, cr);
            vol = std::fabs(dot3(geoms[f].cell, cr));
        }
        if (vol > 0) dens = std::max(dens, n / vol);
    }
    int nbins = bin_cursor;
    hipStream_t st = c->stream;
    HIPCHK(c, hipSetDevice(c->device));
    // frame geometry | atom offsets: one block, staged in pinned memory (the copy is asynchronous; the event tells
    // the next call when the staging block may be overwritten)
    const size_t geo_bytes = (sizeof(FrameGeom) * nf + 15) / 16 * 16, off_bytes = sizeof(int64_t) * (nf + 1);
    const FrameGeom *d_geoms;
    const int64_t *d_offsets;
    const int4 *host_block = nullptr;           // != null: k_prepare_small fetches the staged block itself
    size_t host_block_bytes = 0;
    if (c->staged_in_dev && d_pos == (const double *)c->stage_cur) {
        // positions | species are in the device block already (upload_frames, through the BAR): geometry | offsets behind them
        const size_t at = c->staged_in_dev;
        c->staged_in_dev = 0;
        std::memcpy(c->stage_cur + at, geoms.data(), sizeof(FrameGeom) * nf);
        std::memcpy(c->stage_cur + at + geo_bytes, fr->atom_offsets, off_bytes);
        uf3_store_fence();
        d_geoms = (const FrameGeom *)((const char *)c->stage_cur + at);
        d_offsets = (const int64_t *)((const char *)c->stage_cur + at + geo_bytes);
    } else if (c->pin_in_pending && d_pos == (const double *)c->stage_cur) {
        // small batch staged by upload_frames: positions | species | geometry | offsets leave pin_in in one piece -- fetched by
        // the cell-list kernel itself when that is the one-workgroup kernel, by one copy otherwise
        const size_t at = c->pin_in_pending;
        std::memcpy((char *)c->pin_in.p + at, geoms.data(), sizeof(FrameGeom) * nf);
        std::memcpy((char *)c->pin_in.p + at + geo_bytes, fr->atom_offsets, off_bytes);
        if (natoms <= UF3_SMALL_ATOMS && !uf3_env("UF3_NO_SMALL_PREPARE") && !uf3_env("UF3_NO_ZERO_COPY")) {
            host_block = (const int4 *)c->pin_in.p;
            host_block_bytes = (at + geo_bytes + off_bytes + 15) / 16 * 16;
        } else {
            HIPCHK(c, hipMemcpyAsync(c->stage_cur, c->pin_in.p, at + geo_bytes + off_bytes, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipEventRecord(c->pin_in_done, st));
        }
        c->pin_in_pending = 0;
        d_geoms = (const FrameGeom *)((const char *)c->stage_cur + at);
        d_offsets = (const int64_t *)((const char *)c->stage_cur + at + geo_bytes);
    } else {
        HIPCHK(c, c->geoms.ensure(geo_bytes + off_bytes));
        HIPCHK(c, hipEventSynchronize(c->pin_geo_done));
        HIPCHK(c, c->pin_geo.ensure(geo_bytes + off_bytes));
        std::memcpy(c->pin_geo.p, geoms.data(), sizeof(FrameGeom) * nf);
        std::memcpy((char *)c->pin_geo.p + geo_bytes, fr->atom_offsets, off_bytes);
        HIPCHK(c, hipMemcpyAsync(c->geoms.p, c->pin_geo.p, geo_bytes + off_bytes, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipEventRecord(c->pin_geo_done, st));
        d_geoms = c->geoms.as<FrameGeom>();
        d_offsets = (const int64_t *)((const char *)c->geoms.p + geo_bytes);
    }
    size_t na = (size_t)natoms;
    HIPCHK(c, c->frame_of.ensure(4 * na)); HIPCHK(c, c->atom_bin.ensure(4 * na)); HIPCHK(c, c->atom_wrap.ensure(4 * na));
    HIPCHK(c, c->spec.ensure(na)); HIPCHK(c, c->key_in.ensure(4 * na)); HIPCHK(c, c->val_out.ensure(4 * na));
    HIPCHK(c, c->bin_start.ensure(4 * ((size_t)nbins + 2)));
    HIPCHK(c, c->slots.ensure(sizeof(SlotRec) * na));
    if (!c->flags.p) { HIPCHK(c, c->flags.ensure(64)); HIPCHK(c, hipMemsetAsync(c->flags.p, 0, 64, st)); }
    int *flags = c->flags.as<int>();

    Timed tm(c, T_NBR);
    int tb = 256, gb = (natoms + tb - 1) / tb;
    if (natoms <= UF3_SMALL_ATOMS && !uf3_env("UF3_NO_SMALL_PREPARE")) {
        // an MD step: the whole cell-list stage in one workgroup (and the status words of the launches that follow zeroed)
        hipLaunchKernelGGL(k_prepare_small, dim3(1), dim3(natoms <= 256 ? 256 : 1024), 0, st, b->dev, d_geoms, d_offsets, nf,
                           natoms, nbins, d_pos, d_z, c->frame_of.as<int>(), c->atom_bin.as<int>(), c->atom_wrap.as<int>(),
                           c->spec.as<signed char>(), c->bin_start.as<int>(), c->slots.as<SlotRec>(), flags, 2,
                           host_block, (int4 *)c->stage_cur, (int)(host_block_bytes / 16));
        P.small_prepared = true;
        if (host_block) c->pin_in_busy = true;     // (until the caller's wait for the stream: upload_frames checks)
        // (host_block: no event behind the kernel -- a record between two launches costs the next kernel ~5 us of dispatch
        // latency, and the only caller on this path, the synchronous evaluator entry, waits for the stream before it returns)
        P.flags_zeroed = true;
    } else {
    // n3 need | candidate need | extension-list need | "some atom outside its cell" (k_frame_bins): one fill for all four (the
    // launches behind this stage then need none of their own; every fill is a small kernel on the stream)
    HIPCHK(c, hipMemsetAsync(flags + 1, 0, 4 * sizeof(int), st));
    P.flags_zeroed = true;
    // counting sort by global bin: counts (k_frame_bins) -> exclusive scan = bin starts -> fill -> per-bin order + slot records
    // (k_bin_fill takes every count back to zero: the buffer is cleared only where it has not been through a fill yet)
    {
        const void *before = c->bin_cnt.p;
        HIPCHK(c, c->bin_cnt.ensure(4 * ((size_t)nbins + 8)));
        if (c->bin_cnt.p != before) c->bin_cnt_clean = 0;
        if ((size_t)nbins + 1 > c->bin_cnt_clean) {
            HIPCHK(c, hipMemsetAsync(c->bin_cnt.p, 0, (4 * ((size_t)nbins + 1) + 15) / 16 * 16, st));      // (whole 16-byte pieces: one fill kernel)
            c->bin_cnt_clean = (size_t)nbins + 1;
        }
    }
    hipLaunchKernelGGL(k_frame_bins, dim3(gb), dim3(tb), 0, st, b->dev, d_geoms,
                       d_offsets, nf, natoms, d_pos, d_z, c->frame_of.as<int>(), c->atom_bin.as<int>(),
                       c->atom_wrap.as<int>(), c->spec.as<signed char>(), c->key_in.as<int>(), c->bin_cnt.as<int>(), flags);
    // (up to 16 384 bins -- frames of about 20 k atoms: one workgroup walking 36 k counts of the 50 k-atom frame took 40 us longer than the
    // library scan's two launches)
    if ((size_t)nbins + 1 <= 16384 && !uf3_env("UF3_NO_SMALL_SCAN"))
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, st, (const int *)c->bin_cnt.as<int>(), c->bin_start.as<int>(), nbins + 1);
    else {
        size_t tmp_bytes = 0;
        HIPCHK(c, rocprim::exclusive_scan(nullptr, tmp_bytes, c->bin_cnt.as<int>(), c->bin_start.as<int>(), 0, (size_t)nbins + 1,
                                           rocprim::plus<int>(), st));
        HIPCHK(c, c->sort_tmp.ensure(tmp_bytes));
        HIPCHK(c, rocprim::exclusive_scan(c->sort_tmp.p, tmp_bytes, c->bin_cnt.as<int>(), c->bin_start.as<int>(), 0, (size_t)nbins + 1,
                                           rocprim::plus<int>(), st));
    }
    hipLaunchKernelGGL(k_bin_fill, dim3(gb), dim3(tb), 0, st, c->key_in.as<int>(), natoms, c->bin_start.as<int>(),
                       c->bin_cnt.as<int>(), c->val_out.as<int>());
    hipLaunchKernelGGL(k_bin_finish, dim3((nbins + tb - 1) / tb), dim3(tb), 0, st, nbins, c->bin_start.as<int>(),
                       c->val_out.as<int>(), d_pos, c->atom_wrap.as<int>(), c->spec.as<signed char>(), c->slots.as<SlotRec>());
    }
    HIPCHK(c, hipGetLastError());

    P.natoms = natoms; P.n_frames = nf; P.nbins = nbins; P.max_density = dens;
    P.geo_bytes = geo_bytes + off_bytes;
    P.geoms = d_geoms;
    P.frame_of = c->frame_of.as<int>();
    P.spec = c->spec.as<signed char>();
    P.d_offsets = d_offsets;
    P.cl.bin_start = c->bin_start.as<int>(); P.cl.slots = c->slots.as<SlotRec>();
    P.cl.atom_bin = c->atom_bin.as<int>(); P.cl.atom_wrap = c->atom_wrap.as<int>();
    std::memset(&P.n3, 0, sizeof(P.n3));
    if (need_n3 && b->host.T > 0) {
        // capacity: remembered from earlier calls, else a density estimate; overflow -> grow and redo
        if (c->n3_cap == 0) c->n3_cap = n3_cap_estimate(b, dens);
        for (int attempt = 0; attempt < 6; attempt++) {
            int cap = c->n3_cap;
            int rc = n3_alloc(c, natoms, cap, P.n3);
            if (rc) return rc;
            HIPCHK(c, hipMemsetAsync(flags + 1, 0, sizeof(int), st));
            size_t lds = (size_t)cap * (8 + 32 + 16);
            if ((int)lds > c->lds_max) return fail(c, UF3_EOVERFLOW, "3-body neighbour list does not fit in LDS");
            const bool block_only = n3_hi >= 0 && !(n3_lo == 0 && n3_hi == natoms) && !uf3_env("UF3_NO_HALO");
            if (!block_only)
                hipLaunchKernelGGL(k_build_n3, dim3(natoms), dim3(64), lds, st, b->dev, P.geoms, P.frame_of, P.cl, P.n3,
                                   d_pos, natoms, flags + 1, 0, (const int *)nullptr, (const int *)nullptr);
            else if (n3_hi > n3_lo) {
                // the block, then the atoms its lists mention (marked and collected on the device), nothing else
                const int nb_ = (int)(n3_hi - n3_lo);
                HIPCHK(c, hipMemsetAsync(P.n3.cnt, 0, sizeof(int) * (size_t)natoms, st));      // lists not built: empty
                hipLaunchKernelGGL(k_build_n3, dim3(nb_), dim3(64), lds, st, b->dev, P.geoms, P.frame_of, P.cl, P.n3,
                                   d_pos, natoms, flags + 1, (int)n3_lo, (const int *)nullptr, (const int *)nullptr);
                int rh = build_halo_lists(b, P, P.n3, d_pos, (int)n3_lo, (int)n3_hi);
                if (rh) return rh;
            }
            HIPCHK(c, hipGetLastError());
            if (defer_check && c->n3_tuned) { P.deferred = true; return UF3_OK; }
            int fl[4] = {0, 0, 0, 0};                                   // error flag | list length needed | .. | ..
            HIPCHK(c, hipMemcpyAsync(fl, flags, sizeof(fl), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            if (fl[0]) return check_flags(c);
            const int need = fl[1];
            if (need <= cap) {
                if (!c->n3_tuned) { int rt = n3_tune(c, P.n3, natoms); if (rt) return rt; }
                return UF3_OK;
            }
            c->n3_tuned = true;
            c->n3_cap = (need + 8 + 7) / 8 * 8;
        }
        return fail(c, UF3_EOVERFLOW, "3-body neighbour capacity did not converge");
    }
    return UF3_OK;
}

// ------------------------------------------------------------------------------ featurize
static int ensure_frag(uf3_ctx *c);

// LDS bytes of one featurizer workgroup; must mirror the carve at the top of k_featurize
static size_t feat_lds_bytes(int F, int S, int cap, int cand_cap, bool want_e, size_t n_recs, int mode, int dense_stage,
                             int dense_nrec, int n_pair_cols) {
    const bool dense = mode >= 6;
    if (mode == 0) F = S + n_pair_cols;            // (the pair launch's energy row: one-body + pair columns)
    size_t e_d = want_e ? (size_t)F + (F & 1) : 0;
    size_t cand_d = (size_t)cand_cap * CAND_STRIDE;
    size_t pair_buf_d = std::max(4 * (size_t)n_pair_cols, (3 * (size_t)cap + 1) / 2 + 2);
    size_t stage_d = mode == 0 ? cand_d + pair_buf_d
                     : (dense ? (size_t)dense_stage : (size_t)NSTAGE * ITEM_STRIDE);
    size_t list_d = mode == 0 ? 0 : 5 * (size_t)cap + ((5 * cap) & 1);
    size_t geo_d = dense ? (size_t)7 * GEO_N : 0;
    size_t per_wave_d = list_d + stage_d + (stage_d & 1) + geo_d;
    size_t per_wave_i = mode == 0 ? 0 : 3 * (size_t)cap + 2 * ((size_t)cap + 1) + (UF3_MAX_SPECIES + 2) +
                                        (size_t)cap * (S + 1);
    size_t ints = ((size_t)WPB * per_wave_i + 3) & ~(size_t)3;
    return (e_d + WPB * per_wave_d) * 8 + ints * 4 + n_recs * sizeof(KnotRec) + 32;
}

// LDS footprint of k_featurize3 at list capacity cap (uf3_feat3.h): grows by ~168 B (3-row windows) to ~264 B (6-row windows)
// per list entry and wave -- faster than the launches it replaces, so uf3_featurize_dev asks BEFORE it chooses the launch
static void feat3_shape(const uf3_basis *b, int &ep, int &stage, int &nrec) {
    ep = std::max(3, b->f3_ext_p);
    switch (ep) {
        case 3: stage = F3Cfg<3, 1>::STAGE; nrec = F3Cfg<3, 1>::NREC; break;
        case 4: stage = F3Cfg<4, 2>::STAGE; nrec = F3Cfg<4, 2>::NREC; break;
        case 5: stage = F3Cfg<5, 3>::STAGE; nrec = F3Cfg<5, 3>::NREC; break;
        default: stage = F3Cfg<6, 3>::STAGE; nrec = F3Cfg<6, 3>::NREC; break;
    }
}
static size_t feat3_lds_bytes(const uf3_basis *b, int cap, bool e_lds) {
    int ep, stage, nrec;
    feat3_shape(b, ep, stage, nrec);
    const int F = b->host.F, S = b->host.S;
    const size_t e_d = e_lds ? (size_t)F + (F & 1) : 0, rows_d = (size_t)b->n_f3rows * 18;
    const size_t list_d = 5 * (size_t)cap + ((5 * cap) & 1), tq_d = (size_t)cap * ep * 4, stage_d = (size_t)stage;
    const size_t per_wave_i = 2 * (size_t)cap + 2 * ((size_t)cap + 1) + (UF3_MAX_SPECIES + 2) + (size_t)cap * (S + 1) + 2 * (size_t)nrec + (size_t)cap;
    const size_t ints = ((size_t)WPB * per_wave_i + 3) & ~(size_t)3;
    return (e_d + rows_d + WPB * (list_d + tq_d + stage_d)) * 8 + ints * 4 + (size_t)b->n_f3src * 2 + 32;
}
// ... of k_feat3_w: window rows | per wave: list, bond values, stage | ints
static size_t feat3w_lds_bytes(const uf3_basis *b, int cap) {
    const size_t rows_d = (size_t)b->n_f3rows * 18;
    const size_t per_wave_d = 8 * (size_t)cap + F3WCfg::STAGE, per_wave_i = 4 * (size_t)cap + (UF3_MAX_SPECIES + 2);
    return (rows_d + WPB * per_wave_d) * 8 + WPB * per_wave_i * 4 + 32;
}
#define UF3_LDS_LIMIT ((size_t)160 * 1024 - 512)

static int featurize_dev_impl(uf3_basis *b, const uf3_frames *fr, const double *d_pos, const int32_t *d_z,
                              double *d_xe, double *d_xf, int64_t ld);

extern "C" int uf3_featurize_dev(uf3_basis *b, const uf3_frames *fr, const double *d_pos, const int32_t *d_z,
                                 double *d_xe, double *d_xf) {
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    return featurize_dev_impl(b, fr, d_pos, d_z, d_xe, d_xf, b->host.F);
}

// force rows `ld` doubles apart (ld >= F; the columns F .. ld of a row are not touched): with ld a multiple of 16 every row
// starts on a 128-byte line -- rows of F = 434 or 1798 doubles do not, and the partial lines at their ends are written twice
extern "C" int uf3_featurize_ld_dev(uf3_basis *b, const uf3_frames *fr, const double *d_pos, const int32_t *d_z,
                                    double *d_xe, double *d_xf, int64_t ld) {
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    if (ld < b->host.F || ld > (1 << 24)) return fail(b->ctx, UF3_EINVAL, "uf3_featurize_ld_dev: ld must be >= the number of features");
    return featurize_dev_impl(b, fr, d_pos, d_z, d_xe, d_xf, ld);
}

static int featurize_dev_impl(uf3_basis *b, const uf3_frames *fr, const double *d_pos, const int32_t *d_z,
                              double *d_xe, double *d_xf, int64_t ld) {
    uf3_ctx *c = b->ctx;
    uf3_env_refresh();
    read_f3_env(c);
    if (!d_pos || !d_z) return fail(c, UF3_EINVAL, "null positions / species");
    if (!d_xe && !d_xf) return UF3_OK;
    // verdicts on earlier asynchronous calls that have arrived: returned to the asynchronous caller (their owner) HERE, so not
    // kept for uf3_ctx_synchronize as well -- a caller that redoes the work would meet the same verdict again after a clean redo
    { int rc0 = poll_pending(c, false, false); if (rc0) return rc0; }
    Prepared P;
    const bool old_n3 = uf3_env("UF3_SEPARATE_N3") != nullptr;     // debugging: lists from k_build_n3 instead
    int rc = prepare(b, fr, d_pos, d_z, old_n3, P);       // cell list only: MODE 0 builds the 3-body lists itself
    if (rc) return rc;
    hipStream_t st = c->stream;
    const int F = b->host.F;
    const bool want_e = d_xe != nullptr, want_f = d_xf != nullptr;
    const bool has3 = b->host.T > 0 && !old_n3;
    if (has3 && c->n3_cap == 0) c->n3_cap = n3_cap_estimate(b, P.max_density);
    int cap = 1;
    auto cand_estimate = [&]() {
        double r = b->host.rsearch;
        double est = P.max_density > 0 ? 4.18879 * r * r * r * P.max_density : 64.0;
        return std::max(32, ((int)(est * 1.6) + 16 + 7) / 8 * 8);
    };
    if (c->cand_cap == 0) c->cand_cap = cand_estimate();
    bool caps_reset = false;
    FeatArgs A;
    A.B = b->dev; A.trios = b->d_trios; A.recs = b->d_recs; A.colsrc = b->d_colsrc;
    A.frag = nullptr;
    A.dense_stage = 0; A.dense_nrec = DENSE_NREC;
    A.dsrc = b->d_dsrc; A.n_dsrc = (int)b->n_dsrc;
    A.gsrc = b->d_gsrc; A.n_gsrc = (int)b->n_gsrc; A.gsrc_lds = 0;
    const int dense_modes = (1 << 6) | (1 << 7) | (1 << 8) | (1 << 9);
    const bool dsrc_ok = (b->modes & dense_modes) && b->n_dsrc * sizeof(int) <= 8192 && !uf3_env("UF3_NO_LDS_DSRC");
    A.dsrc_lds = dsrc_ok;
    if (b->modes & dense_modes) { rc = ensure_frag(c); if (rc) return rc; A.frag = c->frag.as<int>(); }
    A.geoms = P.geoms; A.frame_of = P.frame_of; A.cl = P.cl;
    std::memset(&A.n3, 0, sizeof(A.n3));
    A.n3.cap = 1;
    if (old_n3 && P.n3.cap) A.n3 = P.n3;
    A.build_n3 = has3 ? 1 : 0;
    // the block-shared energy row costs 8 F bytes of LDS: past 48 KB the contributions go straight to HBM instead
    A.e_direct = (want_e && (size_t)F * 8 > 48 * 1024) ? 1 : 0;
    A.n3_need = c->flags.as<int>() + 1;
    A.n3_seen = nullptr;
    A.pos = d_pos; A.spec = P.spec; A.x_e = d_xe; A.x_f = d_xf; A.ld = (int)ld; A.natoms = P.natoms;
    A.cand_need = c->flags.as<int>() + 2;
    // (UF3_KEEP_GHOST_TERMS: keep the force terms of ghost-centred triplets whose third atom the reference's image range
    // does not reach -- rows of unwrapped atoms are then the exact gradient of the energy row instead of the reference's)
    A.outside = uf3_env("UF3_KEEP_GHOST_TERMS") ? nullptr : c->flags.as<int>() + 4;
    { const char *e = uf3_env("UF3_DEBUG_SKIP"); A.skip = e ? atoi(e) : 0; }
    for (int attempt = 0; attempt < 6; attempt++) {
        A.cand_cap = c->cand_cap;
        A.n_recs = (int)b->n_recs;
        A.n_pair_recs = (int)b->n_pair_recs;
        A.trio_rec_lo = (int)b->trio_rec_lo;
        A.wrow_base = (int)(b->wrow_lo * 6); A.n_wrows = b->n_wrows;
        A.n_pair_cols = 0;
        for (int p = 0; p < b->host.P; p++) A.n_pair_cols += b->host.pairs[p].nb;
        if (old_n3) cap = A.n3.cap;
        if (has3) {
            cap = c->n3_cap;
            rc = n3_alloc(c, P.natoms, cap, A.n3);
            if (rc) return rc;
        }
        if (want_e) HIPCHK(c, hipMemsetAsync(d_xe, 0, sizeof(double) * (size_t)P.n_frames * F, st));
        if (!P.flags_zeroed || attempt) HIPCHK(c, hipMemsetAsync(A.n3_need, 0, 2 * sizeof(int), st));       // n3_need, cand_need
        const bool img_launch = c->img_mode && has3 && want_f && A.outside;
        const bool ext_lists = img_launch;
        if (ext_lists) {
            // extension lists for batches with atoms outside their cell (the kernel leaves at once otherwise); storage only
            // once a batch has asked for it (flags[3], through the same retry path as the other capacities)
            const int xcap = c->n3x_cap;
            A.n3.xcap = xcap; A.n3.xent = nullptr; A.n3.xoff = nullptr;
            if (xcap > 0) {
                HIPCHK(c, c->n3x_ent.ensure(sizeof(N3Entry) * (size_t)P.natoms * xcap));
                HIPCHK(c, c->n3x_off.ensure(sizeof(int) * (size_t)P.natoms * (UF3_MAX_SPECIES + 1)));
                A.n3.xent = c->n3x_ent.as<N3Entry>(); A.n3.xoff = c->n3x_off.as<int>();
            }
            hipLaunchKernelGGL(k_build_n3_ext, dim3(std::min(P.natoms, 2048)), dim3(64), (size_t)xcap * (8 + 32 + 12) + 16, st,
                               b->dev, P.geoms, P.frame_of, P.cl, A.n3, d_pos, P.natoms, c->flags.as<int>());
        }
        // 3-body force rows by bond factorisation (k_featurize3) where the basis allows it: one launch for all trio blocks,
        // behind the pair launch that builds the lists
        // (dense or long-range lists, past ~200 entries at the default trims and ~100 on the 6 x 12 windows, do not fit its
        // LDS layout: those calls keep the matrix-core / generic launches, which handled them before k_featurize3 existed)
        const bool feat3 = b->feat3_ok && want_f && (has3 || old_n3) && !img_launch && cap <= 255 && !c->env_no_feat3 &&
                           feat3_lds_bytes(b, cap, want_e && !A.e_direct) <= UF3_LDS_LIMIT;
        // (the pair launch below leaves the batch's longest 3-body list in flags[7] when k_featurize3 can use it: see there)
        A.n3_seen = nullptr;
        if (feat3 && has3 && cap > 16 && !c->env_f3_no_cap16 && !c->env_f3_no_select) {
            A.n3_seen = c->flags.as<int>() + 7;
            HIPCHK(c, hipMemsetAsync(A.n3_seen, 0, sizeof(int), st));
        }
        bool restart = false;
        {
            Timed tm(c, T_FEAT);
            for (int mode = 0; mode <= 9; mode++) {
                if (!(b->modes & (1 << mode))) continue;
                if (feat3 && mode >= 1) continue;
                const bool dense_mode = mode >= 6;
                // knot records go to LDS when the block then still reaches the occupancy its registers allow
                // (10: mode 7 with grouped windows only -- the force launches of a basis whose mode-7 blocks are all grouped)
                const int launch_mode = (mode == 7 && want_f && b->all_grouped7 && !img_launch && !uf3_env("UF3_NO_GROUPED_ONLY")) ? 10
                                        : ((mode == 9 && want_f && b->all_banded9 && !img_launch && !uf3_env("UF3_NO_GROUPED_ONLY")) ? 11 : mode);
                // (a grouped-only launch reads the window rows, not the knot records of the legs)
                A.trio_rec_lo = (int)(launch_mode == 10 ? b->wrow_lo : b->trio_rec_lo);
                size_t n_rec_mode = mode == 0 ? b->n_pair_recs : b->n_recs - (size_t)A.trio_rec_lo;
                const int S = b->host.S;
                const size_t cu_lds = 160 * 1024 - 1024;
                if (dense_mode) A.dsrc_lds = dsrc_ok && !(mode == 7 && want_f && b->all_grouped7);
                const bool gsrc_wanted = mode == 7 && want_f && b->dense_grouped[7] && b->n_gsrc > 0 && b->n_gsrc * 2 <= 4096 && !uf3_env("UF3_NO_LDS_GSRC");
                const size_t gsrc_bytes = gsrc_wanted ? b->n_gsrc * 2 : 0;
                A.gsrc_lds = gsrc_wanted;
                size_t lds_extra = ((dense_mode && A.dsrc_lds) ? sizeof(int) * b->n_dsrc : 0) + gsrc_bytes;
                bool recs_lds = false;
                size_t lds = 0, lds_plain = 0, lds_recs = 0;
                bool found = false;
                if (dense_mode) {
                    const int stride = want_f ? b->dense_stride_f[mode] : b->dense_stride[mode];
                    // (grouped windows fold from three 16 x 16 tiles side by side: 768 doubles)
                    const int dump = want_f && b->dense_grouped[mode] ? std::max(768, b->dense_dump[mode]) : b->dense_dump[mode];
                    const bool grouped = want_f && b->dense_grouped[mode];
                    const int nrec_max = std::max(4, std::min(DENSE_NREC, 1200 / stride));
                    // (+ the padding record of an odd pass; grouped windows: one per odd group, at most 2 + (nr odd))
                    auto stage_for = [&](int nr) { return std::max(dump, (nr + (nr & 1) + (grouped ? 2 : 0)) * stride); };
                    // (21 records = 63 staging lanes = one walk step: a block of <= 63 items is one step and three passes)
                    A.dense_nrec = nrec_max; A.dense_stage = stage_for(nrec_max);
                    if (mode <= 7 && !uf3_env("UF3_NO_OCC3")) {
                        // records per staging pass: as many as the stage allows; fewer (smaller stage) if that lets a third
                        // workgroup onto the CU -- the kernel is latency-bound.  Three workgroups per CU need <= 52 KB each
                        // (LDS is granted in coarse granules: 53 KB did not fit).  Candidates in order of preference: the
                        // knot records / window rows in LDS first, then more records per pass
                        const int tries[3] = {nrec_max, std::min(nrec_max, 20), std::min(nrec_max, 15)};
                        const size_t budget = (WPB == 4 ? 52 : 13 * WPB) * 1024;      // (12 waves per CU: 3 x 4 or 2 x 6)
                        const bool dsrc_allowed = A.dsrc_lds, recs_allowed = !uf3_env("UF3_NO_LDS_RECS") && !(img_launch && mode != 0);
                        for (int q = 0; q < 12 && !found; q++) {
                            // (the records / window rows in LDS first: a staging lane reads eleven 16-byte pieces of them per pass)
                            const int nr = tries[(q % 6) / 2];
                            const bool with_recs = q < 6, with_dsrc = (q & 1) == 0;
                            if ((with_recs && !recs_allowed) || (with_dsrc && !dsrc_allowed)) continue;
                            size_t need = feat_lds_bytes(F, S, cap, A.cand_cap, want_e && !A.e_direct, with_recs ? n_rec_mode : 0, mode,
                                                         stage_for(nr), nr, A.n_pair_cols) + (with_dsrc ? sizeof(int) * b->n_dsrc : 0) + gsrc_bytes;
                            if (need <= budget) {
                                found = true; recs_lds = with_recs; lds = lds_recs = need;
                                A.dense_nrec = nr; A.dense_stage = stage_for(nr); A.dsrc_lds = with_dsrc;
                            }
                        }
                    }
                }
                if (!found) {
                    lds_plain = feat_lds_bytes(F, S, cap, A.cand_cap, want_e && !A.e_direct, 0, mode, A.dense_stage, A.dense_nrec, A.n_pair_cols) + lds_extra;
                    lds_recs = feat_lds_bytes(F, S, cap, A.cand_cap, want_e && !A.e_direct, n_rec_mode, mode, A.dense_stage, A.dense_nrec, A.n_pair_cols) + lds_extra;
                    const size_t lds_target = cu_lds / (mode == 0 ? 4 : 2);
                    recs_lds = lds_recs <= lds_target && !uf3_env("UF3_NO_LDS_RECS") && !(img_launch && mode != 0);
                    lds = recs_lds ? lds_recs : lds_plain;
                }
                if (lds > UF3_LDS_LIMIT && mode == 0 && !c->cand_tuned && c->cand_cap > 64) {
                    // the candidate capacity is still the density ESTIMATE (60 % headroom): a dense frame must not fail on the
                    // guess -- take what fits, the overflow flag reports the true need and the call is repeated with it
                    while (c->cand_cap > 64 && feat_lds_bytes(F, S, cap, c->cand_cap, want_e && !A.e_direct, recs_lds ? n_rec_mode : 0, 0,
                                                              A.dense_stage, A.dense_nrec, A.n_pair_cols) + lds_extra > UF3_LDS_LIMIT)
                        c->cand_cap = (c->cand_cap * 7 / 8 + 7) / 8 * 8;
                    A.cand_cap = c->cand_cap;
                    lds = feat_lds_bytes(F, S, cap, A.cand_cap, want_e && !A.e_direct, recs_lds ? n_rec_mode : 0, 0, A.dense_stage,
                                         A.dense_nrec, A.n_pair_cols) + lds_extra;
                }
                if (lds > UF3_LDS_LIMIT && !caps_reset) {
                    // the capacities are grow-only memories of the densest batch this context has seen: a sparser batch on a wider
                    // basis must not fail on them -- back to this batch's own estimates, once, and the call starts over
                    const int est3 = has3 ? n3_cap_estimate(b, P.max_density) : 0, estc = cand_estimate();
                    if ((has3 && c->n3_cap > est3) || c->cand_cap > estc) {
                        if (has3 && c->n3_cap > est3) { c->n3_cap = est3; c->n3_tuned = false; }
                        if (c->cand_cap > estc) { c->cand_cap = estc; c->cand_tuned = false; }
                        caps_reset = restart = true;
                        break;
                    }
                }
                if (lds > UF3_LDS_LIMIT) return fail(c, UF3_EOVERFLOW, "featurizer LDS footprint exceeds 160 KB (F or neighbour count too large)");
                // blocks of WPB waves, each walking a contiguous run of atoms (keeps the shared energy row on
                // one frame); many more blocks than resident slots (measured: 2 per slot 3400 frames/s, 16-48 per slot
                // 3640, finer again slower): the tail of the launch is short and concurrent blocks work on nearby atoms
                int per_cu = std::max(1, std::min(8, (int)((size_t)(160 * 1024) / lds)));
                int n_blocks = std::min((P.natoms + WPB - 1) / WPB, c->n_cu * per_cu * 24);
                int apb = ((P.natoms + n_blocks - 1) / n_blocks + WPB - 1) / WPB * WPB;
                n_blocks = (P.natoms + apb - 1) / apb;
                A.atoms_per_block = apb;
                if (uf3_env("UF3_DEBUG_LDS"))
                    fprintf(stderr, "uf3 featurize mode %d: lds %zu B (plain %zu, with recs %zu), recs_lds %d, cap %d, cand_cap %d, "
                            "blocks %d x %d atoms, n_recs %zu, dense nrec %d stage %d\n", launch_mode, lds, lds_plain, lds_recs,
                            (int)recs_lds, cap, A.cand_cap, n_blocks, apb, n_rec_mode, A.dense_nrec, A.dense_stage);
#define UF3_GRID(n) (((n) + 7) / 8 * 8)      /* whole rounds over the XCDs (surplus workgroups find no atoms) */
/* (the dynamic-LDS attribute belongs to the kernel instance on a device, not to a context or a thread: one process-wide, \
   mutex-protected high-water mark per instance and device that only grows -- ADVICE round 5) */ \
#define UF3_LAUNCH1(E, Fo, R, M, I)                                                                                   \
    do {                                                                                                            \
        {                                                                                                           \
            static std::mutex mu; static size_t have[64] = {0};                                                     \
            std::lock_guard<std::mutex> lk(mu);                                                                     \
            size_t &hv = have[c->device & 63];                                                                      \
            if (lds > hv) {                                                                                         \
                HIPCHK(c, hipFuncSetAttribute((const void *)k_featurize<E, Fo, R, M, I>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                hv = lds;                                                                                           \
            }                                                                                                       \
        }                                                                                                           \
        hipLaunchKernelGGL((k_featurize<E, Fo, R, M, I>), dim3(UF3_GRID(n_blocks)), dim3(WPB * WAVE), lds, st, A);   \
    } while (0)
#define UF3_LAUNCH(M)                                                                                               \
    do {                                                                                                            \
        if (img_launch && M != 0) {   /* (the image-range variants exist with the knot records in HBM only) */       \
            if (want_e) UF3_LAUNCH1(true, true, false, (M == 0 ? 1 : M), true); else UF3_LAUNCH1(false, true, false, (M == 0 ? 1 : M), true); \
        }                                                                                                           \
        else if (want_e && want_f) { if (recs_lds) UF3_LAUNCH1(true, true, true, M, false); else UF3_LAUNCH1(true, true, false, M, false); }   \
        else if (want_f) { if (recs_lds) UF3_LAUNCH1(false, true, true, M, false); else UF3_LAUNCH1(false, true, false, M, false); }    \
        else { if (recs_lds) UF3_LAUNCH1(true, false, true, M, false); else UF3_LAUNCH1(true, false, false, M, false); }               \
    } while (0)
                switch (launch_mode) {
                    case 0: UF3_LAUNCH(0); break;
                    case 1: UF3_LAUNCH(1); break;
                    case 2: UF3_LAUNCH(2); break;
                    case 3: UF3_LAUNCH(3); break;
                    case 4: UF3_LAUNCH(4); break;
                    case 5: UF3_LAUNCH(5); break;
                    case 6: UF3_LAUNCH(6); break;
                    case 7: UF3_LAUNCH(7); break;
                    case 8: UF3_LAUNCH(8); break;
                    case 10: if (want_e) { if (recs_lds) UF3_LAUNCH1(true, true, true, 10, false); else UF3_LAUNCH1(true, true, false, 10, false); }
                             else { if (recs_lds) UF3_LAUNCH1(false, true, true, 10, false); else UF3_LAUNCH1(false, true, false, 10, false); }
                             break;
                    case 11: if (want_e) { if (recs_lds) UF3_LAUNCH1(true, true, true, 11, false); else UF3_LAUNCH1(true, true, false, 11, false); }
                             else { if (recs_lds) UF3_LAUNCH1(false, true, true, 11, false); else UF3_LAUNCH1(false, true, false, 11, false); }
                             break;
                    default: UF3_LAUNCH(9); break;
                }
#undef UF3_LAUNCH
#undef UF3_LAUNCH1
            }
            if (feat3 && !restart) {
                Feat3Args G;
                G.B = b->dev; G.trios = b->d_trios; G.rows = b->d_f3rows; G.n_rows = b->n_f3rows;
                G.fsrc = b->d_f3src; G.trio_fsrc = b->d_f3off; G.n_fsrc = b->n_f3src;
                G.leg_p = b->f3_leg_p; G.leg_n = b->f3_leg_n;
                G.lo_p = b->f3_lo_p; G.ext_p = b->f3_ext_p; G.lo_n = b->f3_lo_n; G.ext_n = b->f3_ext_n;
                G.pr_rows = std::min(b->f3_ext_p, 32 / b->f3_ext_n);
                G.geoms = P.geoms; G.frame_of = P.frame_of; G.n3 = A.n3; G.pos = d_pos; G.spec = P.spec;
                G.x_e = d_xe; G.x_f = d_xf; G.ld = (int)ld; G.natoms = P.natoms; G.e_direct = A.e_direct; G.skip = A.skip;
                G.sel = A.n3_seen; G.sel_mode = 0; G.sel_cap = 16;
                int ep, stage, nrec;
                feat3_shape(b, ep, stage, nrec);
                (void)stage; (void)nrec;
                const int nr = b->f3_nr;
                G.wbuf = nullptr; G.wsz = 0; G.m_lo = 0;
        /* (the dynamic-LDS attribute belongs to the kernel instance on a device: one process-wide high-water mark each) */   \
#define UF3_F3_ATTR(KERNEL, lds)                                                                                           \
    do {                                                                                                                   \
        static std::mutex mu; static size_t have[64] = {0};                                                                \
        std::lock_guard<std::mutex> lk(mu);                                                                                \
        size_t &hv = have[c->device & 63];                                                                                 \
        if ((lds) > hv) {                                                                                                  \
            HIPCHK(c, hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds)));  \
            hv = (lds);                                                                                                    \
        }                                                                                                                  \
    } while (0)
#define UF3_F3_LAUNCH1(E, EFv, NRv, CAPv, HOv)                                                                              \
    do {                                                                                                                   \
        UF3_F3_ATTR((k_featurize3<E, EFv, NRv, CAPv, HOv>), lds);                                                          \
        hipLaunchKernelGGL((k_featurize3<E, EFv, NRv, CAPv, HOv>), dim3(grid), dim3(WPB * WAVE), lds, st, G);               \
    } while (0)
#define UF3_F3_LAUNCH(E, EFv, NRv)                                                                                          \
    do { if (lcap == 16 && !c->env_f3_no_cap16) UF3_F3_LAUNCH1(E, EFv, NRv, 16, false); else UF3_F3_LAUNCH1(E, EFv, NRv, 0, false); } while (0)
#define UF3_F3W_LAUNCH1(CAPv)                                                                                               \
    do {                                                                                                                   \
        UF3_F3_ATTR((k_feat3_w<3, CAPv>), lds);                                                                            \
        hipLaunchKernelGGL((k_feat3_w<3, CAPv>), dim3(grid), dim3(WPB * WAVE), lds, st, G);                                 \
    } while (0)
                // the launch's grid over the atoms m_lo .. natoms - 1 (G.m_lo, G.natoms)
                auto f3_grid = [&](size_t lds) -> unsigned {
                    const int n_at = G.natoms - G.m_lo;
                    int per_cu = std::max(1, std::min(8, (int)((size_t)(160 * 1024) / lds)));
                    const int bps = c->env_f3_bps;
                    int n_blocks = std::min((n_at + WPB - 1) / WPB, c->n_cu * per_cu * bps);
                    int apb = ((n_at + n_blocks - 1) / n_blocks + WPB - 1) / WPB * WPB;
                    n_blocks = (n_at + apb - 1) / apb;
                    G.atoms_per_block = apb;
                    return (unsigned)((n_blocks + 7) / 8 * 8);
                };
                // one launch laid out for lists of up to `lcap` entries (the batch's list array keeps the context's capacity as
                // its stride); sel_mode: see Feat3Args
                auto launch_f3 = [&](int lcap, int sel_mode, bool ho) -> int {
                    G.sel_mode = sel_mode;
                    const size_t lds = feat3_lds_bytes(b, lcap, want_e && !A.e_direct);      // (<= UF3_LDS_LIMIT: checked where feat3 was decided)
                    const unsigned grid = f3_grid(lds);
                    if (c->env_debug_lds)
                        fprintf(stderr, "uf3 featurize3: lds %zu B, cap %d (lists %d apart), selection %d, hand-off %d, blocks %u x %d atoms, window %d x %d, %d round(s)\n",
                                lds, lcap, cap, sel_mode, (int)ho, grid, G.atoms_per_block, G.ext_p, G.ext_n, nr);
                    switch (ep) {
                        case 3:      // (the default trims: the list capacity as a constant also at 24 and 32 -- fcc and denser cells)
                            if (ho) {        // (the measured experiment of round 6: capacity 16 as a constant, everything else generic)
                                if (lcap == 16 && !c->env_f3_no_cap16) { if (want_e) UF3_F3_LAUNCH1(true, 3, 1, 16, true); else UF3_F3_LAUNCH1(false, 3, 1, 16, true); }
                                else { if (want_e) UF3_F3_LAUNCH1(true, 3, 1, 0, true); else UF3_F3_LAUNCH1(false, 3, 1, 0, true); }
                            }
                            else if ((lcap == 24 || lcap == 32) && !c->env_f3_no_cap16) {
                                if (lcap == 24) { if (want_e) UF3_F3_LAUNCH1(true, 3, 1, 24, false); else UF3_F3_LAUNCH1(false, 3, 1, 24, false); }
                                else { if (want_e) UF3_F3_LAUNCH1(true, 3, 1, 32, false); else UF3_F3_LAUNCH1(false, 3, 1, 32, false); }
                            }
                            else if (want_e) UF3_F3_LAUNCH(true, 3, 1); else UF3_F3_LAUNCH(false, 3, 1);
                            break;
                        case 4: if (want_e) UF3_F3_LAUNCH(true, 4, 2); else UF3_F3_LAUNCH(false, 4, 2); break;
                        case 5: if (want_e) UF3_F3_LAUNCH(true, 5, 3); else UF3_F3_LAUNCH(false, 5, 3); break;
                        default: if (want_e) UF3_F3_LAUNCH(true, 6, 3); else UF3_F3_LAUNCH(false, 6, 3); break;
                    }
                    return UF3_OK;
                };
                // the producer of the hand-off (k_feat3_w) over the same atoms, same instance selection
                auto launch_f3w = [&](int lcap, int sel_mode) -> int {
                    G.sel_mode = sel_mode;
                    const size_t lds = feat3w_lds_bytes(b, lcap);
                    const unsigned grid = f3_grid(lds);
                    if (c->env_debug_lds)
                        fprintf(stderr, "uf3 feat3_w: lds %zu B, cap %d (lists %d apart), selection %d, blocks %u x %d atoms, atoms %d .. %d\n",
                                lds, lcap, cap, sel_mode, grid, G.atoms_per_block, G.m_lo, G.natoms);
                    if (lcap == 16 && !c->env_f3_no_cap16) UF3_F3W_LAUNCH1(16); else UF3_F3W_LAUNCH1(0);
                    return UF3_OK;
                };
                // The lists of this batch were built just now at the context's capacity -- an estimate on a context's first call,
                // what the densest batch so far needed later on -- but the instance that serves most cells is the one laid out for 16
                // entries (4 workgroups per CU, LDS offsets as immediates).  Which one applies is known on the device only: the list
                // build leaves the batch's longest list in flags[7], and two launches follow of which one leaves at once.
                auto launch_pair = [&](bool producer, bool ho) -> int {
                    if (A.n3_seen && cap > 16) {
                        int r1 = producer ? launch_f3w(16, 1) : launch_f3(16, 1, ho);
                        if (r1) return r1;
                        return producer ? launch_f3w(cap, 2) : launch_f3(cap, 2, ho);
                    }
                    return producer ? launch_f3w(cap, 0) : launch_f3(cap, 0, ho);
                };
                // Hand-off (round 6; UF3_F3_HANDOFF=1 only -- it was built, measured and LOST: 4460-4630 against
                // 6410 frames/s): the neighbour role's stage-1 sums are computed once, by the centre (k_feat3_w), and
                // reach the neighbours through HBM: slices of whole frames, producer then consumer, one buffer re-used by every slice
                const int ho_default = 0;
                // (looked at on every call: tests and A/B runs flip it between calls on one context)
                const int ho_env = uf3_env("UF3_F3_HANDOFF") ? atoi(uf3_env("UF3_F3_HANDOFF")) : ho_default;
                const int slice_env = uf3_env("UF3_F3_SLICE") ? std::max(0, atoi(uf3_env("UF3_F3_SLICE"))) : 0;
                const bool handoff = ho_env != 0 && ep == 3 && nr == 1 && feat3w_lds_bytes(b, cap) <= UF3_LDS_LIMIT;
                if (!handoff) {
                    int r1 = launch_pair(false, false);
                    if (r1) return r1;
                } else {
                    const int S_ = b->host.S;
                    G.wsz = (G.ext_p * G.ext_n * 4 + 15) / 16 * 16;
                    const size_t per_atom = (size_t)cap * S_ * G.wsz * sizeof(double);
                    const int64_t slice_target = slice_env > 0 ? slice_env : 320000;
                    const int64_t *off = fr->atom_offsets;
                    // (slices: runs of whole frames of at most slice_target atoms, at least one frame)
                    int64_t widest = 0;
                    for (int f0 = 0; f0 < P.n_frames;) {
                        int f1 = f0 + 1;
                        while (f1 < P.n_frames && off[f1 + 1] - off[f0] <= slice_target) f1++;
                        widest = std::max(widest, off[f1] - off[f0]);
                        f0 = f1;
                    }
                    HIPCHK(c, c->f3w.ensure(per_atom * (size_t)widest));
                    G.wbuf = c->f3w.as<double>();
                    for (int f0 = 0; f0 < P.n_frames;) {
                        int f1 = f0 + 1;
                        while (f1 < P.n_frames && off[f1 + 1] - off[f0] <= slice_target) f1++;
                        G.m_lo = (int)off[f0]; G.natoms = (int)off[f1];
                        if (G.natoms > G.m_lo) {
                            int r1 = launch_pair(true, true);
                            if (r1) return r1;
                            r1 = launch_pair(false, true);
                            if (r1) return r1;
                        }
                        f0 = f1;
                    }
                    G.m_lo = 0; G.natoms = P.natoms;
                }
#undef UF3_F3_LAUNCH
#undef UF3_F3_LAUNCH1
#undef UF3_F3W_LAUNCH1
#undef UF3_F3_ATTR
            }
        }
        HIPCHK(c, hipGetLastError());
        if (restart) continue;
        if ((!has3 || c->n3_tuned) && c->cand_tuned && !old_n3 && !uf3_env("UF3_SYNC_FEATURIZE")) {
            // capacities known from earlier calls: do not wait.  The status words follow the launches into a pinned
            // slot; the next call on this context / uf3_ctx_synchronize looks at them (UF3_ERETRY if they overflowed)
            int slot = -1;
            for (int q = 0; q < uf3_ctx::N_PENDING && slot < 0; q++)
                if (!c->pending_chk[(c->pending_head + q) % uf3_ctx::N_PENDING].live) slot = (c->pending_head + q) % uf3_ctx::N_PENDING;
            if (slot < 0) {        // all slots in flight: wait for the oldest
                int rcw = poll_pending(c, true);
                if (rcw) return rcw;
                slot = c->pending_head;
            }
            HIPCHK(c, c->pin_flags.ensure(sizeof(int) * 8 * uf3_ctx::N_PENDING));
            uf3_ctx::Pending &pd = c->pending_chk[slot];
            if (!pd.ev) HIPCHK(c, hipEventCreateWithFlags(&pd.ev, hipEventDisableTiming));
            HIPCHK(c, hipMemcpyAsync((int *)c->pin_flags.p + 8 * slot, c->flags.p, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipEventRecord(pd.ev, st));
            pd.cap = cap; pd.cand = c->cand_cap; pd.has3 = has3; pd.xcap = ext_lists ? c->n3x_cap : (1 << 30); pd.img = img_launch || !(has3 && want_f && A.outside); pd.img_launch = img_launch; pd.live = true;
            c->pending_head = (slot + 1) % uf3_ctx::N_PENDING;
            return UF3_OK;
        }
        int fl[8] = {0, 0, 0, 0, 0, 0, 0, 0};            // error flag, n3 need, candidate need, extension need, atoms far outside
        HIPCHK(c, hipMemcpyAsync(fl, c->flags.p, sizeof(fl), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (fl[0]) return check_flags(c);
        bool redo = false;
        if (has3 && want_f && A.outside && fl[4] && !c->img_mode) { c->img_mode = true; redo = true; }
        else if (img_launch && !fl[4]) c->img_mode = false;                 // (valid all the same) back to the ordinary launches
        if (ext_lists && fl[3] > c->n3x_cap) {
            if (fl[3] > 248) return fail(c, UF3_EOVERFLOW, "extension neighbour lists of atoms far outside their cell exceed 248 entries");
            c->n3x_cap = (fl[3] + 8 + 7) / 8 * 8; redo = true;
        }
        if (has3 && fl[1] > cap) { c->n3_cap = (fl[1] + 8 + 7) / 8 * 8; c->n3_tuned = true; redo = true; }
        if (fl[2] > c->cand_cap) { c->cand_cap = (fl[2] + 16 + 7) / 8 * 8; redo = true; }
        if (redo) continue;
        if (has3 && !c->n3_tuned) { rc = n3_tune(c, A.n3, P.natoms); if (rc) return rc; }
        c->cand_tuned = true;
        return UF3_OK;
    }
    return fail(c, UF3_EOVERFLOW, "neighbour capacities did not converge");
}

// host-buffer helpers
// BAR staging rests on two things the HIP API does not promise: that host stores into a fine-grained device
// allocation, followed by a store fence, are visible to the NEXT kernel launched -- no stale line of the block left in the GPU's
// L2 or scalar cache from the previous launch -- and that this holds with no stream synchronisation in between (the MD steps wait on
// a polled status word).  So the path is gated on a start-up self-test of exactly that sequence, on this device, with this driver:
// 48 rounds of (host writes a new pattern into the block | fence | k_bar_probe reads it with vector AND scalar loads | host polls the
// pinned word the kernel leaves last), patterns compared at the end.  One mismatch, a time-out or any HIP error: bar_ok = false and
// the small batches take the pinned-block route (k_md_fetch / k_prepare_small fetch them) as before round 5.  Costs < 1 ms, once.
static bool bar_self_test(uf3_ctx *c) {
    const int n_words = 2048, rounds = 48;          // (8 KB: what a 128-atom MD step stages is 3.6 KB)
    Buf blk, out;
    PinBuf word;
    bool ok = blk.ensure_fine(4 * (size_t)n_words) == hipSuccess && out.ensure(4 * (size_t)rounds * (n_words + 16)) == hipSuccess &&
              word.ensure(64) == hipSuccess;
    std::vector<unsigned> got;
    if (ok) {
        volatile unsigned *w = (volatile unsigned *)word.p;
        *w = 0;
        std::vector<unsigned> pat(n_words);
        for (int r = 1; r <= rounds && ok; r++) {
            for (int q = 0; q < n_words; q++) pat[q] = 0x9e3779b9u * (unsigned)(r * 4099 + q) + (unsigned)r;
            std::memcpy(blk.p, pat.data(), 4 * (size_t)n_words);
            uf3_store_fence();
            hipLaunchKernelGGL(k_bar_probe, dim3(1), dim3(256), 0, c->stream, (const unsigned *)blk.p, n_words,
                               out.as<unsigned>() + (size_t)(r - 1) * (n_words + 16), (unsigned *)word.p, (unsigned)r);
            ok = hipGetLastError() == hipSuccess;
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(200);
            while (ok && __atomic_load_n((const unsigned *)word.p, __ATOMIC_ACQUIRE) != (unsigned)r)
                if (std::chrono::steady_clock::now() > t_end) ok = false;
        }
        ok = hipStreamSynchronize(c->stream) == hipSuccess && ok;
        if (ok) {
            got.resize((size_t)rounds * (n_words + 16));
            ok = hipMemcpy(got.data(), out.p, 4 * got.size(), hipMemcpyDeviceToHost) == hipSuccess;
        }
        for (int r = 1; r <= rounds && ok; r++)
            for (int q = 0; q < n_words + 16 && ok; q++) {
                const int src = q < n_words ? q : q - n_words;
                ok = got[(size_t)(r - 1) * (n_words + 16) + q] == 0x9e3779b9u * (unsigned)(r * 4099 + src) + (unsigned)r;
            }
    }
    (void)hipGetLastError();
    blk.release(); out.release(); word.release();
    if (!ok && uf3_env("UF3_DEBUG_LDS")) fprintf(stderr, "uf3: BAR staging self-test failed on device %d: small batches go through the pinned block\n", c->device);
    return ok;
}

static int upload_frames(uf3_ctx *c, const uf3_frames *fr, const double *pos, const int32_t *z, int &natoms,
                         bool defer_small = false) {
    if (!fr || fr->n_frames < 1 || !fr->atom_offsets) return fail(c, UF3_EINVAL, "bad uf3_frames");
    int64_t total = fr->atom_offsets[fr->n_frames];
    if (total < 1 || total >= (1LL << 28)) return fail(c, UF3_EINVAL, "batch must hold 1 .. 2^28 atoms");
    if (!pos || !z) return fail(c, UF3_EINVAL, "null positions / species");
    natoms = (int)total;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bp = 24 * (size_t)natoms, bz = 4 * (size_t)natoms;
    // (room behind positions | species for the frame geometry: see pin_in_pending)
    const size_t geo_room = 96 + sizeof(FrameGeom) * (size_t)fr->n_frames + 8 * ((size_t)fr->n_frames + 1);
    // small batches of the synchronous evaluator entry: with a large BAR the block is a fine-grained device allocation and the
    // host stores into it directly (write-combined, posted; a store fence; the launches' doorbell follows over the same link) --
    // no kernel or copy that reads the caller's memory.  The entry's previous call has been waited for (pin_in_busy says when an
    // error return skipped that wait): nothing reads the block.  Gated on bar_self_test.
    bool bar = defer_small && c->bar_ok && bp + bz + geo_room <= UF3_BAR_LIMIT;
    if (bar && !c->bar_tested) {
        c->bar_tested = true;
        if (!uf3_env("UF3_BAR_NO_SELFTEST")) c->bar_ok = bar = bar_self_test(c);
        if (uf3_env("UF3_BAR_FORCE_FAIL")) c->bar_ok = bar = false;       // (tests: the fall-back route on a box where the test passes)
    }
    // (a block of its own: the big batches' staging block stays ordinary device memory)
    if (bar && c->stage_bar.ensure_fine(bp + bz + geo_room) != hipSuccess) { (void)hipGetLastError(); c->bar_ok = false; bar = false; }
    if (!bar) HIPCHK(c, c->stage_pos.ensure(bp + bz + geo_room));  // positions | species, one block
    c->stage_cur = (char *)(bar ? c->stage_bar.p : c->stage_pos.p);
    c->d_stage_z = (int32_t *)(c->stage_cur + bp);
    c->pin_in_pending = 0;
    c->staged_in_dev = 0;
    if (bar) {
        if (c->pin_in_busy) { HIPCHK(c, hipStreamSynchronize(c->stream)); c->pin_in_busy = false; }
        std::memcpy(c->stage_cur, pos, bp);
        std::memcpy(c->stage_cur + bp, z, bz);
        uf3_store_fence();
        c->staged_in_dev = (bp + bz + 15) / 16 * 16;
        // (kernels that read this block follow; an entry that returns an error behind them never waits for them: the flag stays up
        // until a successful wait clears it, and the next upload waits for the stream before it stores into the block -- above)
        c->pin_in_busy = true;
        return UF3_OK;
    }
    if (bp + bz <= UF3_PIN_LIMIT) {
        // (an entry that returned an error behind a launch reading the staging block directly never waited for it)
        if (c->pin_in_busy) { HIPCHK(c, hipStreamSynchronize(c->stream)); c->pin_in_busy = false; }
        HIPCHK(c, hipEventSynchronize(c->pin_in_done));
        HIPCHK(c, c->pin_in.ensure(bp + bz + geo_room));
        std::memcpy(c->pin_in.p, pos, bp);
        std::memcpy((char *)c->pin_in.p + bp, z, bz);
        if (defer_small && bp + bz + geo_room <= UF3_PIN_LIMIT) { c->pin_in_pending = (bp + bz + 15) / 16 * 16; return UF3_OK; }
        HIPCHK(c, hipMemcpyAsync(c->stage_pos.p, c->pin_in.p, bp + bz, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->pin_in_done, c->stream));
    } else {
        HIPCHK(c, hipMemcpyAsync(c->stage_pos.p, pos, bp, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_stage_z, z, bz, hipMemcpyHostToDevice, c->stream));
    }
    return UF3_OK;
}

extern "C" int uf3_featurize(uf3_basis *b, const uf3_frames *fr, const double *pos, const int32_t *z, double *xe,
                             double *xf) {
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    uf3_ctx *c = b->ctx;
    int natoms = 0;
    poll_pending(c, true);                   // verdicts on earlier asynchronous calls: remembered for uf3_ctx_synchronize, not ours
    int rc = upload_frames(c, fr, pos, z, natoms);
    if (rc) return rc;
    size_t F = (size_t)b->host.F;
    size_t be = xe ? 8 * F * fr->n_frames : 0, bf = xf ? 8 * F * 3 * (size_t)natoms : 0;
    if (be) HIPCHK(c, c->stage_out.ensure(be));
    if (bf) HIPCHK(c, c->stage_out2.ensure(bf));
    for (int attempt = 0; attempt < 6; attempt++) {
        rc = uf3_featurize_dev(b, fr, (const double *)c->stage_cur, c->d_stage_z,
                               be ? c->stage_out.as<double>() : nullptr, bf ? c->stage_out2.as<double>() : nullptr);
        if (rc) return rc;
        if (be) HIPCHK(c, hipMemcpyAsync(xe, c->stage_out.p, be, hipMemcpyDeviceToHost, c->stream));
        if (bf) HIPCHK(c, hipMemcpyAsync(xf, c->stage_out2.p, bf, hipMemcpyDeviceToHost, c->stream));
        rc = sync_own_call(c);
        if (rc != UF3_ERETRY) return rc;     // the lists of this very call overflowed: repeat it with the raised capacities
    }
    return fail(c, UF3_EOVERFLOW, "neighbour capacities did not converge");
}

// ------------------------------------------------------------------------------ eval
// ---- MD route: persistent superset lists with a skin (uf3_ctx_md_skin) ----------------------
// The reference's calculator rebuilds supercell, distances and neighbour pairs on every call (calculator.py:124-153,
// 183-343).  An MD loop repeats that work on almost the same positions: with a skin s the context keeps, per atom, every
// neighbour image within r_cut + s of it (k_build_sup: who, which image, reference supercell index, species -- sorted by
// (species, supercell index), no geometry) and a step filters that list by the TRUE distances of the current positions.  The
// survivors come out in the order the rebuild-every-step route sorts its lists into, so a step's result does not depend on
// when the lists were built.  Valid while no atom has moved more than s / 2 from where the lists were built, the cells, the
// offsets and the species are the same: the kernel checks displacement and species, the host the rest.
static bool md_key_matches(const uf3_ctx::MdState &md, const uf3_basis *b, const uf3_frames *fr) {
    if (!md.valid || md.basis != b || !fr || !fr->atom_offsets || !fr->cells || !fr->pbc || fr->n_frames != md.n_frames) return false;
    const size_t nf = (size_t)fr->n_frames;
    return fr->atom_offsets[nf] == md.natoms && !memcmp(fr->atom_offsets, md.offsets.data(), 8 * (nf + 1)) &&
           !memcmp(fr->cells, md.cells.data(), 72 * nf) && !memcmp(fr->pbc, md.pbc.data(), 3 * nf);
}

// may_defer: the caller's step has a verdict of its own (status words read behind its last kernel) and nothing zeroes them between
// this build and that step -- then a build at a capacity that has held before does not wait for its own report: an overflow
// raises the step's "lists outrun" word (k_sup_reverse), the step is discarded and the repeat builds with the host looking
static int md_build(uf3_basis *b, const uf3_frames *fr, const double *d_pos, const int32_t *d_z, Prepared &P, bool may_defer = false) {
    uf3_ctx *c = b->ctx;
    uf3_ctx::MdState &md = c->md;
    md.valid = false; md.stale = false;
    const bool defer = may_defer && md.cap_tuned && !md.verify_next && md.basis == b && md.cap > 0 && !uf3_env("UF3_MD_SYNC_BUILD");
    int rc = prepare(b, fr, d_pos, d_z, false, P, false, 0, -1, md.skin);      // cell list at r_cut + skin
    if (rc) return rc;
    hipStream_t st = c->stream;
    const int natoms = P.natoms, nf = P.n_frames;
    const double r_sup = b->host.rsearch + md.skin;
    if (md.cap == 0 || md.basis != b) {
        const double est = P.max_density > 0 ? 4.18879 * r_sup * r_sup * r_sup * P.max_density : 64.0;
        md.cap = std::max(32, ((int)(est * 1.3) + 16 + 7) / 8 * 8);
    }
    int *flags = c->flags.as<int>();
    const double r_sup2 = r_sup * r_sup * (1.0 + 1e-12);
    for (int attempt = 0; ; attempt++) {
        const int cap = md.cap;
        HIPCHK(c, md.ent.ensure(sizeof(SupEntry) * (size_t)natoms * cap));
        HIPCHK(c, md.cnt.ensure(sizeof(int) * (size_t)natoms));
        if (!(P.small_prepared && attempt == 0)) HIPCHK(c, hipMemsetAsync(flags + 5, 0, sizeof(int), st));    // (the one-workgroup cell-list kernel has zeroed it)
        const size_t lds = (size_t)cap * 16;
        if ((int)lds > c->lds_max) return fail(c, UF3_EOVERFLOW, "MD neighbour list does not fit in LDS");
        hipLaunchKernelGGL(k_build_sup, dim3((unsigned)((natoms + 7) / 8 * 8)), dim3(64), lds, st, b->dev, P.geoms, P.frame_of, P.cl,
                           d_pos, natoms, r_sup2, md.ent.as<SupEntry>(), md.cnt.as<int>(), cap, flags + 5);
        HIPCHK(c, hipGetLastError());
        if (defer) break;
        int fl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIPCHK(c, hipMemcpyAsync(fl, flags, sizeof(fl), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        c->pin_in_busy = false;
        if (fl[0]) return check_flags(c);
        if (fl[5] <= cap) { md.cap_tuned = true; md.verify_next = false; break; }
        if (attempt >= 5) return fail(c, UF3_EOVERFLOW, "MD neighbour capacity did not converge");
        md.cap = (fl[5] + 8 + 7) / 8 * 8;
    }
    hipLaunchKernelGGL(k_sup_reverse, dim3((unsigned)(((natoms + 15) / 16 + 7) / 8 * 8)), dim3(256), 0, st, md.ent.as<SupEntry>(),
                       (const int *)md.cnt.as<int>(), md.cap, natoms, P.geoms, P.frame_of, P.spec, (const int *)(flags + 5),
                       defer ? flags + 2 : (int *)nullptr);
    HIPCHK(c, hipGetLastError());
    // inbox of every (atom, list position): zeroed when (re)allocated -- stamps start at 1
    {
        const size_t need = 32 * (size_t)natoms * md.cap;
        const void *before = md.inbox.p;
        HIPCHK(c, md.inbox.ensure(need));
        if (md.inbox.p != before || md.inbox_zeroed < need) {
            HIPCHK(c, hipMemsetAsync(md.inbox.p, 0, md.inbox.cap, st));
            md.inbox_zeroed = md.inbox.cap;
        }
    }
    // what a step reads besides the lists: frame geometry | offsets, frame and species of every atom, the positions of the build
    const size_t geo_bytes = (sizeof(FrameGeom) * (size_t)nf + 15) / 16 * 16;
    HIPCHK(c, md.geo.ensure(P.geo_bytes));
    HIPCHK(c, md.frame_of.ensure(4 * (size_t)natoms));
    HIPCHK(c, md.spec.ensure((size_t)natoms));
    HIPCHK(c, md.pos_ref.ensure(24 * (size_t)natoms));
    hipLaunchKernelGGL(k_md_snapshot, dim3((unsigned)std::min<size_t>((6 * (size_t)natoms + 255) / 256, 2048)), dim3(256), 0, st,
                       (const unsigned *)P.geoms, md.geo.as<unsigned>(), P.geo_bytes / 4, (const unsigned *)P.frame_of, md.frame_of.as<unsigned>(),
                       (size_t)natoms, P.spec, md.spec.as<signed char>(), (const unsigned *)d_pos, md.pos_ref.as<unsigned>());
    HIPCHK(c, hipGetLastError());
    md.geo_bytes = geo_bytes;
    md.basis = b; md.natoms = natoms; md.n_frames = nf;
    md.offsets.assign(fr->atom_offsets, fr->atom_offsets + nf + 1);
    md.cells.assign(fr->cells, fr->cells + 9 * (size_t)nf);
    md.pbc.assign(fr->pbc, fr->pbc + 3 * (size_t)nf);
    md.valid = true;
    md.builds++;
    return UF3_OK;
}

// a step on the lists: `P` as the kernels behind it expect it, from the context's persistent copies
static void md_prepared(const uf3_ctx *c, Prepared &P) {
    const uf3_ctx::MdState &md = c->md;
    P = Prepared();
    P.natoms = md.natoms; P.n_frames = md.n_frames;
    P.geoms = md.geo.as<FrameGeom>();
    P.d_offsets = (const int64_t *)((const char *)md.geo.p + md.geo_bytes);
    P.frame_of = md.frame_of.as<int>();
    P.spec = md.spec.as<signed char>();
    P.flags_zeroed = true;
    std::memset(&P.n3, 0, sizeof(P.n3));
    std::memset(&P.cl, 0, sizeof(P.cl));
}

extern "C" int uf3_ctx_md_skin(uf3_ctx *c, double skin) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    if (!(skin >= 0.0) || skin > 4.0) return fail(c, UF3_EINVAL, "uf3_ctx_md_skin: skin must lie in [0, 4] Angstrom");
    if (skin != c->md.skin) { c->md.valid = false; c->md.cap = 0; c->md.cap_tuned = false; }
    c->md.skin = skin;
    return UF3_OK;
}

extern "C" int uf3_ctx_md_stats(uf3_ctx *c, int64_t *builds, int64_t *steps, int64_t *redone) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    if (builds) *builds = c->md.builds;
    if (steps) *steps = c->md.steps;
    if (redone) *redone = c->md.redone;
    return UF3_OK;
}

static int eval_impl(uf3_basis *b, const uf3_frames *fr, const double *d_pos, const int32_t *d_z, const double *c1,
                     const double *c2, const double *c3, double *d_energies, double *d_forces, double *d_virials,
                     int64_t atom_begin = 0, int64_t atom_end = -1, int *deferred_cap = nullptr, int *flags_tail = nullptr,
                     double *mirror = nullptr, bool centre_share = false) {
    uf3_ctx *c = b->ctx;
    uf3_env_refresh();
    if (!d_pos || !d_z || !c1 || !d_energies) return fail(c, UF3_EINVAL, "uf3_eval: null argument");
    if ((b->c2_len && !c2) || (b->c3_len && !c3)) return fail(c, UF3_EINVAL, "uf3_eval: missing coefficients");
    // whole batch with forces and 3-body terms: every triplet once, at its centre, + a collection pass; a block of
    // atoms (its neighbours' centres may lie outside the block): every atom walks the triplets it belongs to.
    // Two-pass route, list capacity known: the centre pass builds each atom's 3-body list from the candidates of
    // its own pair walk (no k_build_n3 launch, one neighbourhood scan less).
    const int64_t total = (fr && fr->atom_offsets && fr->n_frames >= 1) ? fr->atom_offsets[fr->n_frames] : -1;
    if (total >= 0 && (atom_begin < 0 || (atom_end >= 0 && (atom_begin > atom_end || atom_end > total))))
        return fail(c, UF3_EINVAL, "uf3_eval_atoms: atom range outside the batch");
    const bool whole = atom_begin == 0 && (atom_end < 0 || atom_end == total);
    // a block of CENTRES (uf3_eval_centres): the two-pass route on the block -- every triplet once, at its centre inside the
    // block; the collection pass then serves the block and its halo (the atoms the block's lists mention), whose lists are
    // built for that purpose.  Rows of all other atoms stay zero; the shares of disjoint blocks add up to the frame.
    const bool centres = centre_share && !whole && d_forces && b->host.T > 0 && atom_end > atom_begin;
    const bool two_pass = (whole || centres) && d_forces && b->host.T > 0 && !uf3_env("UF3_EVAL_GATHER");
    const bool fuse = two_pass && c->n3_tuned && c->n3_cap > 0 && !uf3_env("UF3_SEPARATE_N3");
    Prepared P;
    int rc;
    hipStream_t st = c->stream;
    // MD route: the candidates of every atom from the context's persistent lists instead of a cell-list walk (see md_build)
    const bool md_step = c->md.skin > 0.0 && fuse && (whole || centres) && !uf3_env("UF3_NO_MD");     // (a block of centres too: round 5)
    c->md_step = md_step;
    const bool was_clean = c->md.flags_clean;      // (true only straight after an MD step of eval_host that set no status word)
    c->md.flags_clean = false;
    if (md_step) {
        HIPCHK(c, hipSetDevice(c->device));
        if (c->md.stale || !md_key_matches(c->md, b, fr)) {
            rc = md_build(b, fr, d_pos, d_z, P, true);
            if (rc) return rc;
        } else if (c->staged_in_dev && d_pos == (const double *)c->stage_cur) {
            // positions | species are in the device block already (upload_frames, through the BAR).  k_md_fetch also zeroed the
            // step's status words: needed only when they are not known to be zero (was_clean: the previous call was an MD step
            // of the host entry that set none -- the host has just read them)
            c->staged_in_dev = 0;
            if (!was_clean)
                hipLaunchKernelGGL(k_md_fetch, dim3(1), dim3(64), 0, st, (const int4 *)nullptr, (int4 *)nullptr, 0, c->flags.as<int>());
        } else if (c->pin_in_pending && d_pos == (const double *)c->stage_cur) {
            // a small batch staged by upload_frames: positions | species are still in the caller's pinned block
            const size_t at = c->pin_in_pending;
            c->pin_in_pending = 0;
            if (c->md.natoms <= UF3_SMALL_ATOMS && !uf3_env("UF3_NO_ZERO_COPY")) {
                hipLaunchKernelGGL(k_md_fetch, dim3(1), dim3(256), 0, st, (const int4 *)c->pin_in.p, (int4 *)c->stage_cur, (int)(at / 16),
                                   c->flags.as<int>());
                c->pin_in_busy = true;
            } else {
                HIPCHK(c, hipMemcpyAsync(c->stage_cur, c->pin_in.p, at, hipMemcpyHostToDevice, st));
                HIPCHK(c, hipEventRecord(c->pin_in_done, st));
                HIPCHK(c, hipMemsetAsync(c->flags.as<int>() + 1, 0, 4 * sizeof(int), st));
            }
        } else
            HIPCHK(c, hipMemsetAsync(c->flags.as<int>() + 1, 0, 4 * sizeof(int), st));
        md_prepared(c, P);
    } else {
        rc = prepare(b, fr, d_pos, d_z, !fuse, P, deferred_cap != nullptr, atom_begin, whole ? -1 : atom_end);
        if (rc) return rc;
    }
    if (deferred_cap) *deferred_cap = P.deferred ? P.n3.cap : 0;
    size_t n1 = (size_t)b->host.S, n2 = b->c2_len, n3 = b->c3_len;
    HIPCHK(c, c->coeff.ensure(8 * (n1 + n2 + n3 + 1)));
    double *dc = c->coeff.as<double>();
    // an MD loop passes the same model every step: upload (and wait for the caller's buffers) only when they changed
    std::vector<double> &sh = c->coeff_shadow;
    const bool same = c->coeff_dev == (const void *)dc && sh.size() == n1 + n2 + n3 && !memcmp(sh.data(), c1, 8 * n1) &&
                      (!n2 || !memcmp(sh.data() + n1, c2, 8 * n2)) && (!n3 || !memcmp(sh.data() + n1 + n2, c3, 8 * n3));
    if (!same) {
        sh.resize(n1 + n2 + n3);
        memcpy(sh.data(), c1, 8 * n1);
        if (n2) memcpy(sh.data() + n1, c2, 8 * n2);
        if (n3) memcpy(sh.data() + n1 + n2, c3, 8 * n3);
        c->coeff_dev = nullptr;
        HIPCHK(c, hipMemcpyAsync(dc, sh.data(), 8 * sh.size(), hipMemcpyHostToDevice, st));
        // the window table of the CW instances: only when every coefficient outside the kept-bin window of the centre legs is
        // exactly zero (bins without a column decompress to zero: any fitted or loaded model; arbitrary grids keep the global rows)
        c->cw_of = nullptr;
        std::vector<double> tab;
        if (b->eval_cw_ok && n3) {
            const double *g3 = sh.data() + n1 + n2;
            const int lo = b->cw_lo, ext = b->cw_ext, dm = b->cw_dim_m, dn = b->cw_dim_n;
            tab.assign((size_t)b->cw_bytes / 8, 0.0);
            bool zero_outside = true;
            for (size_t t = 0; t < b->cw_lut_off.size() && zero_outside; t++) {
                const double *g = g3 + b->cw_lut_off[t];
                for (int l = 0; l < b->cw_dim_l[t] && zero_outside; l++)
                    for (int mm = 0; mm < dm && zero_outside; mm++) {
                        const bool in = l >= lo && l < lo + ext && mm >= lo && mm < lo + ext;
                        const double *row = g + ((size_t)l * dm + mm) * dn;
                        if (in) std::memcpy(tab.data() + ((t * ext + (l - lo)) * ext + (mm - lo)) * dn, row, 8 * (size_t)dn);
                        else for (int q = 0; q < dn; q++) zero_outside = zero_outside && row[q] == 0.0;
                    }
            }
            if (zero_outside) {
                HIPCHK(c, c->coeff_cw.ensure((size_t)b->cw_bytes));
                HIPCHK(c, hipMemcpyAsync(c->coeff_cw.p, tab.data(), (size_t)b->cw_bytes, hipMemcpyHostToDevice, st));
                c->cw_of = b;
            }
        }
        HIPCHK(c, hipStreamSynchronize(st));
        c->coeff_dev = dc;
    }
    HIPCHK(c, c->e_atom.ensure(8 * (size_t)P.natoms * (d_virials ? 7 : 1)));
    if (atom_end < 0) atom_end = P.natoms;
    if (atom_begin < 0 || atom_begin > atom_end || atom_end > P.natoms)
        return fail(c, UF3_EINVAL, "uf3_eval_atoms: atom range outside the batch");
    // (atoms outside the range: the per-frame sums run over the range only)
    EvalArgs A;
    A.B = b->dev; A.geoms = P.geoms; A.frame_of = P.frame_of; A.cl = P.cl; A.n3 = P.n3;
    if (!A.n3.cap) A.n3.cap = 1;
    A.pos = d_pos; A.spec = P.spec; A.c1 = dc; A.c2 = dc + n1; A.c3 = dc + n1 + n2;
    A.e_atom = c->e_atom.as<double>(); A.forces = d_forces; A.natoms = P.natoms;
    A.atom_lo = (int)atom_begin; A.atom_hi = (int)atom_end;
    A.virial = d_virials ? A.e_atom + P.natoms : nullptr;
    A.nbr_f = nullptr; A.n3_need = nullptr; A.fuse_n3 = 0;
    A.halo_mark = nullptr;
    A.c3w = nullptr; A.cw_bytes = 0; A.cw_zero = 0; A.cw_lo = 0; A.cw_ext = 0; A.lds_per_wave = 0; A.cw_recs_bytes = 0; A.cw_c2 = 0;
    A.part_e = nullptr; A.part_v = nullptr; A.part_off = nullptr;
    // (rows of atoms that no centre of the block touches: zero.  A block of centres with the fused list build zeroes rows, list
    // counts and halo marks in ONE launch inside the loop below)
    const bool zero3 = centres && fuse && !md_step && !uf3_env("UF3_NO_HALO");
    if (centre_share && !whole && d_forces && !zero3 && !md_step)
        HIPCHK(c, hipMemsetAsync(d_forces, 0, 24 * (size_t)P.natoms, st));
    {
        Timed tm(c, T_EVAL);
        for (int attempt = 0; ; attempt++) {
            c->tail_signalled = false;
            bool part_sums = false;
            if (fuse) {
                rc = n3_alloc(c, P.natoms, c->n3_cap, A.n3);
                if (rc) return rc;
                A.fuse_n3 = 1;
                A.n3_need = c->flags.as<int>() + 1;
                if (!P.flags_zeroed || attempt) HIPCHK(c, hipMemsetAsync(A.n3_need, 0, (md_step ? 3 : 1) * sizeof(int), st));
            }
            if (md_step) {
                const uf3_ctx::MdState &md = c->md;
                A.fuse_n3 = 0;
                A.geoms = P.geoms; A.frame_of = P.frame_of; A.spec = P.spec;
                A.sup_ent = md.ent.as<SupEntry>(); A.sup_cnt = md.cnt.as<int>(); A.sup_cap = md.cap;
                A.pos_ref = md.pos_ref.as<double>(); A.z_now = d_z;
                const double hard = 0.5 * md.skin * (1.0 - 1e-9), soft = 0.7 * hard;
                A.md_hard2 = hard * hard; A.md_soft2 = soft * soft;
                A.md_flags = c->flags.as<int>() + 2;
                c->md.steps++;
                HIPCHK(c, c->md.surv.ensure(sizeof(int) * (size_t)P.natoms * A.n3.cap));
                A.md_inbox = c->md.inbox.as<double>(); A.md_surv = c->md.surv.as<int>();
                A.md_stamp = (double)c->md.steps;
                A.md_mark = nullptr; A.md_mark_now = 0;
                if (centres) {
                    // marks of the atoms the block's centres write to: zeroed when (re)allocated, launch numbers only grow
                    const void *before = c->md.mark.p;
                    HIPCHK(c, c->md.mark.ensure(sizeof(int) * (size_t)P.natoms));
                    if (c->md.mark.p != before) HIPCHK(c, hipMemsetAsync(c->md.mark.p, 0, c->md.mark.cap, st));
                    A.md_mark = c->md.mark.as<int>();
                    A.md_mark_now = (int)(c->md.steps & 0x7fffffff);
                }
            }
            const size_t cap = (size_t)A.n3.cap;
            // own list (32 + 20 B per entry), queue of bonds, force on the entries (24), walk-order entries + keys (48)
            const size_t lds_plain = cap * 32 + (5 * cap + 2) * 4 + 16 + 2 * WAVE * EVAL_Q * sizeof(double) + cap * 24 + cap * 48;
            // TAB instances (centre legs from per-bond tables, k_eval): + leg n's knot records, + the tables when they do not fit
            // over the queue.  Chosen by the basis alone -- not by the capacity -- unless the longer layout does not fit at all
            const size_t lds_tab = lds_plain + 16 + EVAL_TAB_KN * sizeof(KnotRec) + (cap > EVAL_TAB_CAP ? 136 * cap + 16 : 0);
            const bool tab = two_pass && b->eval_tab_ok && (int)lds_tab <= c->lds_max && !uf3_env("UF3_EVAL_NO_TAB");
            // CW instances (MD route only: measured 197 against 190 M atom-steps/s there, a loss on the plain route): EVAL_CW_WAVES
            // one-atom waves per workgroup around one copy of the window table, every knot record and the pair coefficients
            const size_t cw_per_wave = ((md_step ? lds_plain - cap * 48 : lds_plain) + 15) / 16 * 16;      // (the walk-order arrays: fused builds only)
            const size_t cw_recs = b->n_recs * sizeof(KnotRec), cw_c2b = (n2 * 8 + 15) / 16 * 16;
            const size_t lds_cw = (size_t)b->cw_bytes + cw_recs + cw_c2b + EVAL_CW_WAVES * cw_per_wave;
            const bool cw = tab && md_step && c->cw_of == (const void *)b && cap <= 16 &&      /* (lists of <= 16 entries: the force gather's stage, k_eval) */
                             lds_cw <= UF3_LDS_LIMIT / (EVAL_CW_WAVES > 8 ? 1 : 2) && !uf3_env("UF3_EVAL_NO_CW");
            const size_t lds = cw ? lds_cw : (tab ? lds_tab : lds_plain);
            // WIN instances: the same window table read through global memory by the one-wave TAB instances CW does not serve
            const bool win = tab && !cw && c->cw_of == (const void *)b && !uf3_env("UF3_EVAL_NO_CW");
            if (win) { A.c3w = c->coeff_cw.as<double>(); A.cw_lo = b->cw_lo; A.cw_ext = b->cw_ext; }
            if (cw) {
                A.c3w = c->coeff_cw.as<double>(); A.cw_bytes = b->cw_bytes; A.cw_zero = b->cw_zero; A.cw_lo = b->cw_lo; A.cw_ext = b->cw_ext;
                A.lds_per_wave = (int)cw_per_wave; A.cw_recs_bytes = (int)cw_recs; A.cw_c2 = (int)n2;
            }
            if (!cw && (int)lds > c->lds_max) return fail(c, UF3_EOVERFLOW, "3-body neighbour list does not fit in LDS");
            if (two_pass) {
                if (!md_step) {
                    HIPCHK(c, c->nbr_f.ensure(24 * (size_t)P.natoms * cap));
                    A.nbr_f = c->nbr_f.as<double>();
                }
                // (a block of centres on the MD route: its halo's rows are sums into zeros -- again on every attempt)
                if (md_step && centres) HIPCHK(c, hipMemsetAsync(d_forces, 0, 24 * (size_t)P.natoms, st));
                const int64_t n_centres = atom_end - atom_begin;
                if (zero3) {
                    HIPCHK(c, c->halo.ensure(sizeof(int) * ((size_t)P.natoms + 4)));
                    const size_t words = 8 * (size_t)P.natoms;
                    hipLaunchKernelGGL(k_zero3, dim3((unsigned)std::min<size_t>((words + 255) / 256, 4096)), dim3(256), 0, st,
                                       (unsigned *)d_forces, 6 * (size_t)P.natoms, (unsigned *)A.n3.cnt, (size_t)P.natoms,
                                       (unsigned *)c->halo.p, (size_t)P.natoms);
                }
                else if (centres && fuse && !md_step) HIPCHK(c, hipMemsetAsync(A.n3.cnt, 0, sizeof(int) * (size_t)P.natoms, st));   // (lists not built: empty)
                const dim3 eg((unsigned)((n_centres + 7) / 8 * 8));
                {
                    // instance: strain derivative | list capacity 16 as a constant | candidates from the persistent lists | centre
                    // legs from per-bond tables (one set of 3-body legs, T <= 64, short lists: the usual case)
                    const bool cap16 = cap == 16 && !uf3_env("UF3_EVAL_NO_CAP16");
                    const int inst = (A.virial ? 1 : 0) | (cap16 ? 2 : 0) | (md_step ? 4 : 0) | (tab ? 8 : 0);
#define UF3_EVAL_CW_CASE(I) case I: {                                                                                                  \
        static std::mutex mu; static size_t have[64] = {0};                                                                           \
        { std::lock_guard<std::mutex> lk(mu); size_t &hv = have[c->device & 63];                                                      \
          if (lds > hv) { HIPCHK(c, hipFuncSetAttribute((const void *)k_eval<false, ((I) & 1) != 0, ((I) & 2) ? 16 : 0, ((I) & 4) != 0, true, true>, \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hv = lds; } }         \
        hipLaunchKernelGGL((k_eval<false, ((I) & 1) != 0, ((I) & 2) ? 16 : 0, ((I) & 4) != 0, true, true>), eg_cw, dim3(64 * EVAL_CW_WAVES), lds, st, A); } break;
                    const dim3 eg_cw((unsigned)(((n_centres + EVAL_CW_WAVES - 1) / EVAL_CW_WAVES + 7) / 8 * 8));
                    if (cw) switch (inst & 7) {
                        // (MD route only: with the walk-order arrays of the fused list build the waves' own LDS is 7.1 KB each, and two
                        // 8-wave workgroups + two copies of the tables do not fit a CU -- 168 of 160 KB; not instantiated)
                        UF3_EVAL_CW_CASE(4) UF3_EVAL_CW_CASE(5) UF3_EVAL_CW_CASE(6) UF3_EVAL_CW_CASE(7)
                        default: return fail(c, UF3_EINVAL, "k_eval<CW> outside the MD route");
                    }
#undef UF3_EVAL_CW_CASE
#define UF3_EVAL_WIN_CASE(I) case I: hipLaunchKernelGGL((k_eval<false, ((I) & 1) != 0, ((I) & 2) ? 16 : 0, ((I) & 4) != 0, true, false, true>), eg, dim3(64), lds, st, A); break;
                    else if (win) switch (inst & 7) {
                        UF3_EVAL_WIN_CASE(0) UF3_EVAL_WIN_CASE(1) UF3_EVAL_WIN_CASE(2) UF3_EVAL_WIN_CASE(3)
                        UF3_EVAL_WIN_CASE(4) UF3_EVAL_WIN_CASE(5) UF3_EVAL_WIN_CASE(6) UF3_EVAL_WIN_CASE(7)
                    }
#undef UF3_EVAL_WIN_CASE
#define UF3_EVAL_CASE(I) case I: hipLaunchKernelGGL((k_eval<false, ((I) & 1) != 0, ((I) & 2) ? 16 : 0, ((I) & 4) != 0, ((I) & 8) != 0>), eg, dim3(64), lds, st, A); break;
                    else switch (inst) {
                        UF3_EVAL_CASE(0) UF3_EVAL_CASE(1) UF3_EVAL_CASE(2) UF3_EVAL_CASE(3) UF3_EVAL_CASE(4) UF3_EVAL_CASE(5)
                        UF3_EVAL_CASE(6) UF3_EVAL_CASE(7) UF3_EVAL_CASE(8) UF3_EVAL_CASE(9) UF3_EVAL_CASE(10) UF3_EVAL_CASE(11)
                        UF3_EVAL_CASE(12) UF3_EVAL_CASE(13) UF3_EVAL_CASE(14) UF3_EVAL_CASE(15)
                    }
#undef UF3_EVAL_CASE
                }
                if (fuse && deferred_cap) *deferred_cap = (int)cap;
                if (centres && !md_step) {
                    // the block's lists exist now (this launch built them, or prepare did): the halo's, then the collection
                    // pass over block + halo
                    if (fuse || uf3_env("UF3_NO_HALO")) { int rh = build_halo_lists(b, P, A.n3, d_pos, (int)atom_begin, (int)atom_end, zero3); if (rh) return rh; }
                    A.halo_mark = c->halo.as<int>();
                }
                if (md_step && centres)      // every atom checks itself; the block and the atoms its centres wrote to collect
                    hipLaunchKernelGGL(k_eval_collect_md_halo, dim3((unsigned)((P.natoms + 15) / 16)), dim3(256), 0, st, A);
                else if (md_step) {
                    // one whole frame, large: the collection pass also leaves per-workgroup sums of the atoms' energies (and strain
                    // derivatives), and the frame sum adds those -- natoms / 16 values instead of natoms (12 -> 4 us at 50 k atoms)
                    part_sums = P.n_frames == 1 && whole && P.natoms >= 8192 && !mirror && !uf3_env("UF3_NO_PART_SUMS");
                    if (part_sums) {
                        const size_t n_wg = ((size_t)P.natoms + 15) / 16;
                        HIPCHK(c, c->part_sums.ensure(8 * (7 * n_wg + 2)));
                        A.part_e = c->part_sums.as<double>(); A.part_v = A.part_e + n_wg; A.part_off = (long long *)(A.part_v + 6 * n_wg);
                    }
                    hipLaunchKernelGGL(k_eval_collect_md, dim3((unsigned)(((P.natoms + 15) / 16 + 7) / 8 * 8)), dim3(256), 0, st, A);
                }
                else hipLaunchKernelGGL(k_eval_collect, dim3((unsigned)(((P.natoms + 15) / 16 + 7) / 8 * 8)), dim3(256), 0, st, A);
            } else if (atom_end > atom_begin) {
                if (A.virial) hipLaunchKernelGGL((k_eval<true, true>), dim3((unsigned)((atom_end - atom_begin + 7) / 8 * 8)), dim3(64), lds, st, A);
                else hipLaunchKernelGGL((k_eval<true, false>), dim3((unsigned)((atom_end - atom_begin + 7) / 8 * 8)), dim3(64), lds, st, A);
            }
            // (one workgroup per frame and component: wide for big frames, the loop is a latency chain)
            const int sum_threads = P.natoms / P.n_frames >= 2048 ? 1024 : 256;
            // (into the caller's pinned block: the launch's last workgroup signals the host itself; flags[12] counts workgroups)
            // INVARIANT of the polled completion (eval_host returns without hipStreamSynchronize once the sequence number shows):
            // k_frame_sum is the LAST launch of the call's chain, and its sequence store, behind system-scope fences, the last
            // access of that chain to pin_in / pin_out -- the next call's memcpy into those blocks relies on it.  Anything queued
            // behind this launch (a timing event, a second mirror write) must clear `tail_signalled` so that the host waits
            // for the stream instead.
            unsigned seq = 0;
            if (mirror && !c->timing && !uf3_env("UF3_NO_TAIL_SPIN")) {        // (event timing queues a record behind the chain)
                if (++c->eval_seq == 0) c->eval_seq = 1;
                seq = c->eval_seq;
                c->tail_signalled = true;
            }
            // a device-resident call that looks at its status words before it returns (the fused list build, the MD route): the
            // last kernel puts them, and the call's sequence number behind them, into a pinned block the host polls -- no copy
            // of 16 bytes through the runtime's staging and no wait for the stream's completion signal (~10 us of an MD step)
            int *flags_host = nullptr;
            unsigned *seq_host = nullptr;
            if (!mirror && !flags_tail && fuse && !deferred_cap && !c->timing && !uf3_env("UF3_NO_TAIL_SPIN")) {
                HIPCHK(c, c->pin_eval.ensure(64));
                flags_host = (int *)c->pin_eval.p; seq_host = (unsigned *)c->pin_eval.p + 4;
                if (++c->eval_seq == 0) c->eval_seq = 1;
                seq = c->eval_seq;
            }
            if (part_sums)
                hipLaunchKernelGGL(k_frame_sum, dim3(1, d_virials ? 7 : 1), dim3(1024), 0, st, (const double *)A.part_e,
                                   (const double *)A.part_v, (const int64_t *)A.part_off, d_energies, d_virials, (const int *)c->flags.as<int>(), flags_host ? flags_host : flags_tail,
                                   mirror, (const double *)d_forces, d_forces ? 3 * P.natoms : 0, 0, (P.natoms + 15) / 16, seq, c->flags.as<int>() + 12,
                                   seq_host);
            else
            hipLaunchKernelGGL(k_frame_sum, dim3(P.n_frames, d_virials ? 7 : 1), dim3(sum_threads), 0, st, A.e_atom,
                               A.virial, P.d_offsets, d_energies, d_virials, (const int *)c->flags.as<int>(), flags_host ? flags_host : flags_tail,
                               mirror, (const double *)d_forces, d_forces ? 3 * P.natoms : 0, (int)atom_begin, (int)atom_end, seq, c->flags.as<int>() + 12,
                               seq_host);
            if (fuse && !deferred_cap) {
                // the lists were part of this launch: did they fit?  (Asked after everything is queued -- all kernels
                // are safe on clipped lists -- so that the GPU does not idle while the host looks.)
                int fl[4] = {0, 0, 0, 0};
                bool arrived = false;
                if (seq_host) {
                    const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
                    for (int spin = 0; !arrived; spin++) {
                        arrived = __atomic_load_n(seq_host, __ATOMIC_ACQUIRE) == seq;
                        if (!arrived && (spin & 255) == 255 && std::chrono::steady_clock::now() > t_end) break;
                    }
                    if (!arrived) HIPCHK(c, hipStreamSynchronize(st));
                    std::memcpy(fl, flags_host, sizeof(fl));
                } else {
                    HIPCHK(c, hipMemcpyAsync(fl, c->flags.p, sizeof(fl), hipMemcpyDeviceToHost, st));
                    HIPCHK(c, hipStreamSynchronize(st));
                }
                if (fl[0]) return check_flags(c);
                if (md_step && fl[2]) {
                    // an atom beyond skin / 2 of where the lists were built (or of another species): the results are void --
                    // new lists from these positions, and again
                    if (attempt >= 5) return fail(c, UF3_EOVERFLOW, "MD neighbour lists did not settle");
                    c->md.redone++;
                    rc = md_build(b, fr, d_pos, d_z, P);
                    if (rc) return rc;
                    md_prepared(c, P);
                    continue;
                }
                if (md_step && fl[3]) c->md.stale = true;        // (early warning: rebuild in front of the next step)
                if (fl[1] > (int)cap) {
                    if (attempt >= 5) return fail(c, UF3_EOVERFLOW, "3-body neighbour capacity did not converge");
                    c->n3_cap = (fl[1] + 8 + 7) / 8 * 8;
                    continue;
                }
            }
            break;
        }
    }
    HIPCHK(c, hipGetLastError());
    return UF3_OK;
}

extern "C" int uf3_eval_dev(uf3_basis *b, const uf3_frames *fr, const double *d_pos, const int32_t *d_z,
                            const double *c1, const double *c2, const double *c3, double *d_energies, double *d_forces) {
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    return eval_impl(b, fr, d_pos, d_z, c1, c2, c3, d_energies, d_forces, nullptr);
}

extern "C" int uf3_eval_virial_dev(uf3_basis *b, const uf3_frames *fr, const double *d_pos, const int32_t *d_z,
                                   const double *c1, const double *c2, const double *c3, double *d_energies,
                                   double *d_forces, double *d_virials) {
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    return eval_impl(b, fr, d_pos, d_z, c1, c2, c3, d_energies, d_forces, d_virials);
}

static int eval_host(uf3_basis *b, const uf3_frames *fr, const double *pos, const int32_t *z, const double *c1,
                     const double *c2, const double *c3, double *energies, double *forces, double *virials,
                     int64_t atom_begin = 0, int64_t atom_end = -1, bool centre_share = false) {
    uf3_ctx *c = b->ctx;
    uf3_env_refresh();
    if (!energies) return fail(c, UF3_EINVAL, "uf3_eval: null energies");
    int natoms = 0;
    poll_pending(c, false);                  // (arrived verdicts on asynchronous featurizer calls: for uf3_ctx_synchronize, not for us)
    int rc = upload_frames(c, fr, pos, z, natoms, true);
    if (rc) return rc;
    // results in one block: energies [nf] | virials [nf][6] | forces [natoms][3]
    const size_t nf = (size_t)fr->n_frames, bf = forces ? 24 * (size_t)natoms : 0, total = 8 * nf * 7 + bf;
    HIPCHK(c, c->stage_out.ensure(total + 16));
    double *d_e = c->stage_out.as<double>(), *d_v = d_e + nf, *d_f = d_e + 7 * nf;
    if (forces && (atom_begin != 0 || (atom_end >= 0 && atom_end != natoms)))   // rows of other ranks' atoms: zero
        HIPCHK(c, hipMemsetAsync(d_f, 0, bf, c->stream));
    if (total + 16 <= UF3_PIN_LIMIT) {
        // small batch (an MD step): results and the neighbour stage's status words come back in ONE download and ONE
        // wait -- the lists are built at the remembered capacity and the evaluation runs on them right away; if they
        // overflowed (or a species / wrap error was flagged) the results are discarded and the call repeated / failed
        HIPCHK(c, c->pin_out.ensure(total + 32));
        int *d_flags_tail = (int *)((char *)c->stage_out.p + total);
        for (int attempt = 0; attempt < 6; attempt++) {
            int cap_used = 0;
            // (up to the one-workgroup cell-list limit the last kernel writes the results into the pinned block itself; the
            // mirror's layout has the forces right behind the 7 nf sums, as d_e does)
            const bool zero_copy = natoms <= UF3_SMALL_ATOMS && !uf3_env("UF3_NO_ZERO_COPY");
            *(volatile unsigned *)((char *)c->pin_out.p + total + 16) = 0;
            rc = eval_impl(b, fr, (const double *)c->stage_cur, c->d_stage_z, c1, c2, c3, d_e, forces ? d_f : nullptr,
                           virials ? d_v : nullptr, atom_begin, atom_end, &cap_used, d_flags_tail,
                           zero_copy ? (double *)c->pin_out.p : nullptr, centre_share);
            if (rc) return rc;
            if (!zero_copy) HIPCHK(c, hipMemcpyAsync(c->pin_out.p, d_e, total + 16, hipMemcpyDeviceToHost, c->stream));
            bool arrived = false;
            if (zero_copy && c->tail_signalled) {
                // the tail kernel's last store, behind a system-scope fence, is this call's sequence number: poll the pinned
                // block for it (a few us sooner than the stream's completion signal reaches the host); bounded, then the
                // ordinary wait
                const unsigned *word = (const unsigned *)((const char *)c->pin_out.p + total + 16);
                const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
                for (int spin = 0; !arrived; spin++) {
                    arrived = __atomic_load_n(word, __ATOMIC_ACQUIRE) == c->eval_seq;
                    if (!arrived && (spin & 255) == 255 && std::chrono::steady_clock::now() > t_end) break;
                }
            }
            if (!arrived) HIPCHK(c, hipStreamSynchronize(c->stream));
            c->pin_in_busy = false;
            poll_pending(c, true);           // (remembered, see above)
            {
                const int *fl = (const int *)((const char *)c->pin_out.p + total);
                if (fl[0]) return check_flags(c);
                if (c->md_step && fl[2]) { c->md.valid = false; c->md.verify_next = true; c->md.redone++; continue; }      // (lists outrun, or clipped by a build nobody waited for: rebuilt by the repeat, with the host looking)
                if (c->md_step && fl[3]) c->md.stale = true;
                if (cap_used && fl[1] > cap_used) { c->n3_cap = (fl[1] + 8 + 7) / 8 * 8; continue; }
                c->md.flags_clean = c->md_step && !fl[1] && !fl[2] && !fl[3];
            }
            const double *h = (const double *)c->pin_out.p;
            std::memcpy(energies, h, 8 * nf);
            if (virials) std::memcpy(virials, h + nf, 48 * nf);
            if (forces) std::memcpy(forces, h + 7 * nf, bf);
            return UF3_OK;
        }
        return fail(c, UF3_EOVERFLOW, "3-body neighbour capacity did not converge");
    }
    rc = eval_impl(b, fr, (const double *)c->stage_cur, c->d_stage_z, c1, c2, c3, d_e, forces ? d_f : nullptr,
                   virials ? d_v : nullptr, atom_begin, atom_end, nullptr, nullptr, nullptr, centre_share);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(energies, d_e, 8 * nf, hipMemcpyDeviceToHost, c->stream));
    if (virials) HIPCHK(c, hipMemcpyAsync(virials, d_v, 48 * nf, hipMemcpyDeviceToHost, c->stream));
    if (forces) HIPCHK(c, hipMemcpyAsync(forces, d_f, bf, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    poll_pending(c, true);
    return check_flags(c);
}

extern "C" int uf3_eval(uf3_basis *b, const uf3_frames *fr, const double *pos, const int32_t *z, const double *c1,
                        const double *c2, const double *c3, double *energies, double *forces) {
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    return eval_host(b, fr, pos, z, c1, c2, c3, energies, forces, nullptr);
}

extern "C" int uf3_eval_virial(uf3_basis *b, const uf3_frames *fr, const double *pos, const int32_t *z, const double *c1,
                               const double *c2, const double *c3, double *energies, double *forces, double *virials) {
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    return eval_host(b, fr, pos, z, c1, c2, c3, energies, forces, virials);
}

extern "C" int uf3_eval_atoms_dev(uf3_basis *b, const uf3_frames *fr, const double *d_pos, const int32_t *d_z,
                                  const double *c1, const double *c2, const double *c3, int64_t atom_begin,
                                  int64_t atom_end, double *d_energies, double *d_forces, double *d_virials) {
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    if (atom_end < 0) return fail(b->ctx, UF3_EINVAL, "uf3_eval_atoms: atom range outside the batch");
    return eval_impl(b, fr, d_pos, d_z, c1, c2, c3, d_energies, d_forces, d_virials, atom_begin, atom_end);
}

extern "C" int uf3_eval_atoms(uf3_basis *b, const uf3_frames *fr, const double *pos, const int32_t *z, const double *c1,
                              const double *c2, const double *c3, int64_t atom_begin, int64_t atom_end,
                              double *energies, double *forces, double *virials) {
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    if (atom_end < 0) return fail(b->ctx, UF3_EINVAL, "uf3_eval_atoms: atom range outside the batch");
    return eval_host(b, fr, pos, z, c1, c2, c3, energies, forces, virials, atom_begin, atom_end);
}

extern "C" int uf3_eval_centres_dev(uf3_basis *b, const uf3_frames *fr, const double *d_pos, const int32_t *d_z,
                                    const double *c1, const double *c2, const double *c3, int64_t atom_begin,
                                    int64_t atom_end, double *d_energies, double *d_forces, double *d_virials) {
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    if (atom_end < 0) return fail(b->ctx, UF3_EINVAL, "uf3_eval_centres: atom range outside the batch");
    return eval_impl(b, fr, d_pos, d_z, c1, c2, c3, d_energies, d_forces, d_virials, atom_begin, atom_end, nullptr, nullptr, nullptr, true);
}

extern "C" int uf3_eval_centres(uf3_basis *b, const uf3_frames *fr, const double *pos, const int32_t *z, const double *c1,
                                const double *c2, const double *c3, int64_t atom_begin, int64_t atom_end,
                                double *energies, double *forces, double *virials) {
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    if (atom_end < 0) return fail(b->ctx, UF3_EINVAL, "uf3_eval_centres: atom range outside the batch");
    return eval_host(b, fr, pos, z, c1, c2, c3, energies, forces, virials, atom_begin, atom_end, true);
}

extern "C" int uf3_allreduce_sum_f64(uf3_ctx *c, double *d_buf, int64_t n);
// ------------------------------------------------------------------------------ the fit's accumulation, host arrays in (round 5)
// What pipeline.DeviceFitAccumulator.add_frames does with torch's buffers and streams, inside the library: frames (one pointer
// per frame: positions, atomic numbers, force targets -- no concatenation on the caller's side) are packed chunk by chunk into
// one of two pinned blocks, ONE transfer per chunk runs on a copy stream beside the previous chunk's kernels, the rows of a
// chunk live in HBM between the featurizer and the Gram kernels, and everything additive sits in one flat device buffer
// [G_e | G_f | o_e | o_f | m_e | m_f].  Reference: BasisFeaturizer.evaluate -> HDF5 -> WeightedLinearModel.fit_from_file
// (uf3/representation/process.py:121-291, uf3/regression/least_squares.py:355-483).
struct uf3_fit {
    uf3_basis *b = nullptr;
    uf3_ctx *c = nullptr;
    bool with_forces = true;
    int F = 0;
    int64_t max_atoms = 320000;
    double first_fraction = 0.125;
    int pack_threads = 4;
    Buf flat, xe, xf, frozen_idx, frozen_c;
    double *flat_ext = nullptr;      // the caller's device buffer for the pieces instead of `flat` (uf3_fit_use_flat)
    int n_frozen = 0;
    struct Set { PinBuf host; Buf dev; hipEvent_t copied = nullptr, consumed = nullptr; bool copied_live = false, consumed_live = false; } set[2];
    hipStream_t copy_stream = nullptr;
    double n_e = 0, n_f = 0;
    long long n_chunks = 0;
};

extern "C" void uf3_fit_destroy(uf3_fit *f) {
    if (!f) return;
    hipSetDevice(f->c->device);
    hipStreamSynchronize(f->c->stream);
    if (f->copy_stream) { hipStreamSynchronize(f->copy_stream); hipStreamDestroy(f->copy_stream); }
    for (auto &st : f->set) {
        st.host.release(); st.dev.release();
        if (st.copied) hipEventDestroy(st.copied);
        if (st.consumed) hipEventDestroy(st.consumed);
    }
    f->flat.release(); f->xe.release(); f->xf.release(); f->frozen_idx.release(); f->frozen_c.release();
    delete f;
}

extern "C" int uf3_fit_reset(uf3_fit *f) {
    if (!f) return fail(nullptr, UF3_EINVAL, "null fit");
    uf3_ctx *c = f->c;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(f->flat_ext ? (void *)f->flat_ext : f->flat.p, 0, 8 * (2 * (size_t)f->F * f->F + 2 * (size_t)f->F + 6), c->stream));
    f->n_e = f->n_f = 0; f->n_chunks = 0;
    return UF3_OK;
}

// the pieces into a device buffer of the caller's (2 F^2 + 2 F + 6 doubles, [G_e | G_f | o_e | o_f | m_e | m_f]; not zeroed here:
// uf3_fit_reset does that) -- so that frames given as host arrays and batches already resident in HBM (uf3_featurize_dev +
// uf3_gram*_dev by the caller) add up in one place.  NULL: back to the accumulator's own buffer.
// a call's first chunk holds this fraction of max_atoms_per_chunk and the following ones double it up to the limit (default 0.125:
// the GPU starts sooner; 1: chunks of equal size)
extern "C" int uf3_fit_first_chunk(uf3_fit *f, double fraction) {
    if (!f || !(fraction > 0.0) || fraction > 1.0) return fail(f ? f->c : nullptr, UF3_EINVAL, "uf3_fit_first_chunk: fraction in (0, 1]");
    f->first_fraction = fraction;
    return UF3_OK;
}

extern "C" int uf3_fit_use_flat(uf3_fit *f, double *d_flat) {
    if (!f) return fail(nullptr, UF3_EINVAL, "null fit");
    f->flat_ext = d_flat;
    return UF3_OK;
}

extern "C" int uf3_fit_create(uf3_basis *b, int with_forces, int64_t max_atoms_per_chunk, const int64_t *frozen_idx,
                              const double *frozen_c, int32_t n_frozen, uf3_fit **out) {
    if (!b || !out || n_frozen < 0 || (n_frozen && (!frozen_idx || !frozen_c))) return fail(b ? b->ctx : nullptr, UF3_EINVAL, "uf3_fit_create: bad argument");
    uf3_ctx *c = b->ctx;
    uf3_env_refresh();                      // (UF3_FIT_PACK_THREADS below: not a stale cache)
    HIPCHK(c, hipSetDevice(c->device));
    uf3_fit *f = new uf3_fit();
    f->b = b; f->c = c; f->with_forces = with_forces != 0; f->F = b->host.F;
    if (max_atoms_per_chunk > 0) f->max_atoms = max_atoms_per_chunk;
    f->n_frozen = n_frozen;
    if (const char *e = uf3_env("UF3_FIT_PACK_THREADS")) f->pack_threads = std::max(1, std::min(16, atoi(e)));
    auto bail = [&](int rc) { uf3_fit_destroy(f); return rc; };
    if (f->flat.ensure(8 * (2 * (size_t)f->F * f->F + 2 * (size_t)f->F + 6)) != hipSuccess) return bail(fail(c, UF3_ENOMEM, "uf3_fit_create: out of device memory"));
    if (n_frozen) {
        if (f->frozen_idx.ensure(8 * (size_t)n_frozen) != hipSuccess || f->frozen_c.ensure(8 * (size_t)n_frozen) != hipSuccess)
            return bail(fail(c, UF3_ENOMEM, "uf3_fit_create: out of device memory"));
        if (hipMemcpy(f->frozen_idx.p, frozen_idx, 8 * (size_t)n_frozen, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(f->frozen_c.p, frozen_c, 8 * (size_t)n_frozen, hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(c, UF3_EHIP, "uf3_fit_create: upload of the frozen columns failed"));
    }
    if (hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(c, UF3_EHIP, "uf3_fit_create: no copy stream"));
    for (auto &st : f->set)
        if (hipEventCreateWithFlags(&st.copied, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&st.consumed, hipEventDisableTiming) != hipSuccess)
            return bail(fail(c, UF3_EHIP, "uf3_fit_create: no events"));
    int rc = uf3_fit_reset(f);
    if (rc) return bail(rc);
    *out = f;
    return UF3_OK;
}

// frames: n_frames entries each of atom_counts / pos[f] ([N_f][3]) / z[f] ([N_f], int64 when z_is_int64 else int32) / cells
// ([n_frames][9]) / pbc ([n_frames][3]) / energies (total energy of the frame) / forces[f] ([N_f][3]; null array: no forces)
extern "C" int uf3_fit_add(uf3_fit *f, int32_t n_frames, const int64_t *atom_counts, const double *const *pos, const void *const *z,
                           int z_is_int64, const double *cells, const uint8_t *pbc, const double *energies, const double *const *forces) {
    if (!f) return fail(nullptr, UF3_EINVAL, "null fit");
    uf3_ctx *c = f->c;
    if (n_frames < 0 || (n_frames && (!atom_counts || !pos || !z || !cells || !pbc || !energies))) return fail(c, UF3_EINVAL, "uf3_fit_add: bad argument");
    if (f->with_forces && n_frames && !forces) return fail(c, UF3_EINVAL, "uf3_fit_add: this accumulator was set up with forces: pass them");
    HIPCHK(c, hipSetDevice(c->device));
    const int F = f->F;
    const size_t F2 = (size_t)F * F;
    double *flat = f->flat_ext ? f->flat_ext : f->flat.as<double>(), *gram_e = flat, *gram_f = flat + F2, *ord_e = flat + 2 * F2, *ord_f = ord_e + F, *mom = ord_f + F;
    int start = 0;
    for (int i = 0; i < n_frames; i++)
        if (atom_counts[i] <= 0) return fail(c, UF3_EINVAL, "uf3_fit_add: a frame without atoms (its per-atom energy target is undefined)");
    // chunk sizes grow geometrically from first_fraction of the limit: the GPU starts after a short pack, and the pack of
    // chunk k + 1 (host, about half the GPU's time per frame on a one-species basis) hides behind the kernels of chunk k
    // The chunks of this call, planned ahead (round 6: the plan, not the loop, decides).  The ramp: first_fraction of the limit, then
    // doubling -- the GPU starts after a short pack.  Behind the ramp the remaining frames are spread EVENLY over the fewest chunks
    // of at most the limit (128 frames of 10 k atoms used to end on a 4-frame chunk -- 4 + 8 + 16 + 3 x 32 + 4 -- whose launches
    // fill the GPU for a fraction of their tails).  Every buffer is sized ONCE for the largest chunk (a Buf that grows chunk by
    // chunk frees and allocates -- an implicit device synchronisation -- in the middle of the copy / compute overlap).
    std::vector<std::pair<int, int>> plan;
    {
        double fr_ = f->first_fraction;
        int s0 = 0;
        while (s0 < n_frames && fr_ < 1.0) {
            const int64_t limit = std::max<int64_t>(1, (int64_t)(f->max_atoms * fr_));
            fr_ = std::min(1.0, 2.0 * fr_);
            int s1 = s0;
            int64_t atoms = 0;
            while (s1 < n_frames && (s1 == s0 || atoms + atom_counts[s1] <= limit)) atoms += atom_counts[s1++];
            plan.emplace_back(s0, s1);
            s0 = s1;
        }
        if (s0 < n_frames) {
            int64_t rest = 0;
            for (int i = s0; i < n_frames; i++) rest += atom_counts[i];
            int64_t n_chunks = (rest + f->max_atoms - 1) / f->max_atoms;
            for (;; n_chunks++) {
                // chunk k ends behind the first frame at which the running atom count reaches (k + 1) / n_chunks of the rest
                std::vector<std::pair<int, int>> tail;
                int q0 = s0;
                int64_t run = 0;
                bool fits = true;
                for (int64_t k = 0; k < n_chunks && q0 < n_frames; k++) {
                    const int64_t target = (rest * (k + 1) + n_chunks - 1) / n_chunks;
                    int q1 = q0;
                    int64_t atoms = 0;
                    while (q1 < n_frames && (q1 == q0 || run + atoms < target)) atoms += atom_counts[q1++];
                    if (k == n_chunks - 1) while (q1 < n_frames) atoms += atom_counts[q1++];
                    fits = fits && (atoms <= f->max_atoms || q1 == q0 + 1);
                    run += atoms;
                    tail.emplace_back(q0, q1);
                    q0 = q1;
                }
                if (fits || n_chunks >= n_frames - s0) { plan.insert(plan.end(), tail.begin(), tail.end()); break; }
            }
        }
        int64_t big_atoms = 0, big_block = 0;
        int big_nf = 0;
        for (const auto &ch : plan) {
            int64_t atoms = 0;
            for (int i = ch.first; i < ch.second; i++) atoms += atom_counts[i];
            big_atoms = std::max(big_atoms, atoms); big_nf = std::max(big_nf, ch.second - ch.first);
            big_block = std::max<int64_t>(big_block, 6 * atoms + 2 * (int64_t)(ch.second - ch.first) + (atoms + 1) / 2);
        }
        if (big_atoms >= (1LL << 28)) return fail(c, UF3_EINVAL, "uf3_fit_add: a chunk must hold 1 .. 2^28 atoms");
        for (int q = 0; q < 2 && n_frames; q++) {
            uf3_fit::Set &st = f->set[q];
            if (8 * (size_t)big_block > st.host.cap && st.copied_live) { HIPCHK(c, hipEventSynchronize(st.copied)); st.copied_live = false; }
            HIPCHK(c, st.host.ensure(8 * (size_t)big_block));
            if (8 * (size_t)big_block > st.dev.cap && st.consumed_live) { HIPCHK(c, hipEventSynchronize(st.consumed)); st.consumed_live = false; }
            HIPCHK(c, st.dev.ensure(8 * (size_t)big_block));
        }
        if (n_frames) {
            if (8 * (size_t)big_nf * F > f->xe.cap || (f->with_forces && 24 * (size_t)big_atoms * F > f->xf.cap)) HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, f->xe.ensure(8 * (size_t)big_nf * F));
            if (f->with_forces) HIPCHK(c, f->xf.ensure(24 * (size_t)big_atoms * F));
        }
    }
    for (const auto &chunk : plan) {
        start = chunk.first;
        const int stop = chunk.second;
        int64_t atoms = 0;
        for (int i = start; i < stop; i++) atoms += atom_counts[i];
        const int nf = stop - start;
        if (atoms < 1 || atoms >= (1LL << 28)) return fail(c, UF3_EINVAL, "uf3_fit_add: a chunk must hold 1 .. 2^28 atoms");
        // block layout, host and device alike: positions [3 A] | force targets [3 A] | per-atom energies [nf] | atom counts [nf] | species [A] (int32)
        const size_t A3 = 3 * (size_t)atoms, n_block = 2 * A3 + 2 * (size_t)nf + ((size_t)atoms + 1) / 2;
        uf3_fit::Set &st = f->set[f->n_chunks & 1];
        if (st.copied_live) { HIPCHK(c, hipEventSynchronize(st.copied)); st.copied_live = false; }      // (the pinned block is free again)
        HIPCHK(c, st.host.ensure(8 * n_block));
        if (8 * n_block > st.dev.cap && st.consumed_live) { HIPCHK(c, hipEventSynchronize(st.consumed)); st.consumed_live = false; }
        HIPCHK(c, st.dev.ensure(8 * n_block));
        double *h = (double *)st.host.p, *h_pos = h, *h_yf = h + A3, *h_ye = h + 2 * A3, *h_cnt = h_ye + nf;
        int32_t *h_z = (int32_t *)(h_cnt + nf);
        std::vector<int64_t> offsets(nf + 1, 0);
        for (int i = 0; i < nf; i++) {
            const int64_t n = atom_counts[start + i];
            if (n && (!pos[start + i] || !z[start + i] || (f->with_forces && !forces[start + i]))) return fail(c, UF3_EINVAL, "uf3_fit_add: null frame array");
            offsets[i + 1] = offsets[i] + n;
        }
        auto pack = [&](int i0, int i1) {
            for (int i = i0; i < i1; i++) {
                const int64_t n = atom_counts[start + i], k = offsets[i];
                std::memcpy(h_pos + 3 * k, pos[start + i], 24 * (size_t)n);
                if (z_is_int64) { const int64_t *zz = (const int64_t *)z[start + i]; for (int64_t q = 0; q < n; q++) h_z[k + q] = (int32_t)zz[q]; }
                else std::memcpy(h_z + k, z[start + i], 4 * (size_t)n);
                if (f->with_forces) std::memcpy(h_yf + 3 * k, forces[start + i], 24 * (size_t)n);
                h_cnt[i] = (double)n;
                h_ye[i] = energies[start + i] / (double)n;      // per-atom normalisation of the targets (least_squares.py:697-700)
            }
        };
        // big chunks are packed by a few threads (a single core copies ~8 GB/s into pinned memory: 2 ms per 32 frames of 10 k atoms)
        // (from two frames of 10 k atoms on: one core packs 0.18 ms per such frame, the GPU takes 0.145 on a one-species basis)
        const int n_thr = (atoms >= 20000 && nf >= 2) ? std::min(f->pack_threads, nf) : 1;
        if (n_thr > 1) {
            std::vector<std::thread> thr;
            int done_to = nf / n_thr;                   // frames [0, done_to) are this thread's; a thread that cannot be started leaves its share to it too
            std::vector<std::pair<int, int>> mine;
            for (int t = 1; t < n_thr; t++) {
                const int i0 = (int)((int64_t)nf * t / n_thr), i1 = (int)((int64_t)nf * (t + 1) / n_thr);
                try { thr.emplace_back(pack, i0, i1); } catch (...) { mine.emplace_back(i0, i1); }
            }
            pack(0, done_to);
            for (auto &r : mine) pack(r.first, r.second);
            for (auto &t : thr) t.join();
        } else pack(0, nf);
        if (st.consumed_live) HIPCHK(c, hipStreamWaitEvent(f->copy_stream, st.consumed, 0));            // (the device block is free again)
        HIPCHK(c, hipMemcpyAsync(st.dev.p, st.host.p, 8 * n_block, hipMemcpyHostToDevice, f->copy_stream));
        HIPCHK(c, hipEventRecord(st.copied, f->copy_stream));
        st.copied_live = true;
        HIPCHK(c, hipStreamWaitEvent(c->stream, st.copied, 0));
        double *d = st.dev.as<double>(), *d_pos = d, *d_yf = d + A3, *d_ye = d + 2 * A3, *d_cnt = d_ye + nf;
        const int32_t *d_z = (const int32_t *)(d_cnt + nf);
        // the rows of this chunk (the previous chunk's kernels are ahead of these launches on the same stream)
        HIPCHK(c, f->xe.ensure(8 * (size_t)nf * F));
        if (f->with_forces) HIPCHK(c, f->xf.ensure(8 * A3 * F));
        uf3_frames fr;
        fr.n_frames = nf; fr.atom_offsets = offsets.data(); fr.cells = cells + 9 * (size_t)start; fr.pbc = pbc + 3 * (size_t)start;
        int rc = uf3_featurize_dev(f->b, &fr, d_pos, d_z, f->xe.as<double>(), f->with_forces ? f->xf.as<double>() : nullptr);
        if (rc) return rc;
        rc = uf3_fit_rows_dev(c, nf, F, f->xe.as<double>(), d_cnt, d_ye, f->with_forces ? d_yf : nullptr, f->with_forces ? (int64_t)A3 : 0,
                              f->n_frozen ? f->frozen_idx.as<int64_t>() : nullptr, f->n_frozen ? f->frozen_c.as<double>() : nullptr, f->n_frozen, mom);
        if (rc) return rc;
        rc = uf3_gram_dev(c, f->xe.as<double>(), d_ye, nf, F, F, 1, gram_e, ord_e);
        if (rc) return rc;
        if (f->with_forces) {
            rc = uf3_gram_force_rows_dev(f->b, f->xf.as<double>(), d_yf, d_z, atoms, F, 1, gram_f, ord_f);
            if (rc) return rc;
        }
        HIPCHK(c, hipEventRecord(st.consumed, c->stream));
        st.consumed_live = true;
        f->n_e += nf;
        if (f->with_forces) f->n_f += (double)A3;
        f->n_chunks++;
    }
    return UF3_OK;
}

// the additive pieces of this rank over the n_keep unfrozen columns, frozen columns folded out: [G_e | G_f | o_e | o_f | m_e | m_f]
// (2 n_keep^2 + 2 n_keep + 6 doubles) into host memory; allreduce != 0: summed over the ranks of the context's communicator first
extern "C" int uf3_fit_pack(uf3_fit *f, const int64_t *keep, int32_t n_keep, int allreduce, double *out_host) {
    if (!f || !out_host || n_keep < 0 || n_keep > f->F || (n_keep && !keep)) return fail(f ? f->c : nullptr, UF3_EINVAL, "uf3_fit_pack: bad argument");
    uf3_ctx *c = f->c;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = uf3_ctx_synchronize(c);                  // (verdicts on the asynchronous featurizer calls: UF3_ERETRY = start over)
    if (rc) return rc;
    const size_t n = 2 * (size_t)n_keep * n_keep + 2 * (size_t)n_keep + 6;
    Buf d_keep, d_out;
    HIPCHK(c, d_keep.ensure(8 * (size_t)std::max(n_keep, 1)));
    HIPCHK(c, d_out.ensure(8 * n));
    auto done = [&](int r) { hipStreamSynchronize(c->stream); d_keep.release(); d_out.release(); return r; };
    if (n_keep && hipMemcpyAsync(d_keep.p, keep, 8 * (size_t)n_keep, hipMemcpyHostToDevice, c->stream) != hipSuccess) return done(fail(c, UF3_EHIP, "uf3_fit_pack: upload failed"));
    rc = uf3_fit_pack_dev(c, f->F, f->flat_ext ? f->flat_ext : f->flat.as<double>(), d_keep.as<int64_t>(), n_keep, f->n_frozen ? f->frozen_idx.as<int64_t>() : nullptr,
                          f->n_frozen ? f->frozen_c.as<double>() : nullptr, f->n_frozen, f->n_e, f->n_f, d_out.as<double>());
    if (rc) return done(rc);
    if (allreduce && c->comm) { rc = uf3_allreduce_sum_f64(c, d_out.as<double>(), (int64_t)n); if (rc) return done(rc); }
    if (hipMemcpyAsync(out_host, d_out.p, 8 * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return done(fail(c, UF3_EHIP, "uf3_fit_pack: download failed"));
    if (hipStreamSynchronize(c->stream) != hipSuccess) return done(fail(c, UF3_EHIP, "uf3_fit_pack: stream"));
    return done(UF3_OK);
}

extern "C" int uf3_fit_info(const uf3_fit *f, int64_t *n_chunks, double *n_energy_rows, double *n_force_rows) {
    if (!f) return fail(nullptr, UF3_EINVAL, "null fit");
    if (n_chunks) *n_chunks = f->n_chunks;
    if (n_energy_rows) *n_energy_rows = f->n_e;
    if (n_force_rows) *n_force_rows = f->n_f;
    return UF3_OK;
}

// ------------------------------------------------------------------------------ RCCL behind the C ABI
// One process per GPU; the ONE exchange of the path is the sum of the packed normal-equation pieces (and, for a decomposed
// frame, of [forces | energy | strain derivative]) over the ranks -- what the reference does by returning per-chunk results
// to the parent process and adding them there (uf3/representation/process.py:196-254, uf3/regression/least_squares.py:
// 391-412); SURVEY 8b's `uf3_gram_allreduce`.  librccl is opened at run time: the copy already in the process (a PyTorch-ROCm
// wheel brings its own, next to its own HIP runtime, and two HIP runtimes in one process do not mix), else UF3_RCCL_PATH,
// else the system's.
struct RcclId { char b[128]; };            // ncclUniqueId, passed by value
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};
static RcclApi &rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api;
    tried = true;
    const char *names[] = {uf3_env("UF3_RCCL_PATH"), "librccl.so.1", "librccl.so"};
    for (int pass = 0; pass < 2 && !api.lib; pass++)                 // first: whatever is loaded already
        for (const char *nm : names) {
            if (!nm || api.lib) continue;
            api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
        }
    if (!api.lib) { api.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return api; }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce) { api.why = "librccl lacks the nccl* entry points"; api.lib = nullptr; }
    return api;
}
static int rccl_fail(uf3_ctx *c, const char *what, int rc) {
    RcclApi &r = rccl_api();
    return fail(c, UF3_EHIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}

extern "C" int uf3_comm_unique_id(uf3_ctx *c, void *id128) {
    uf3_env_refresh();
    if (!c || !id128) return fail(c, UF3_EINVAL, "uf3_comm_unique_id: null argument");
    RcclApi &r = rccl_api();
    if (!r.lib) return fail(c, UF3_EHIP, r.why);
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = r.GetUniqueId(id128);
    return rc ? rccl_fail(c, "ncclGetUniqueId", rc) : UF3_OK;
}

extern "C" int uf3_comm_init(uf3_ctx *c, int n_ranks, int rank, const void *id128) {
    uf3_env_refresh();
    if (!c || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, UF3_EINVAL, "uf3_comm_init: bad argument");
    RcclApi &r = rccl_api();
    if (!r.lib) return fail(c, UF3_EHIP, r.why);
    if (c->comm) { int rc0 = uf3_comm_destroy(c); if (rc0) return rc0; }
    HIPCHK(c, hipSetDevice(c->device));
    RcclId id;
    std::memcpy(id.b, id128, sizeof(id.b));
    const int rc = r.CommInitRank(&c->comm, n_ranks, id, rank);
    if (rc) { c->comm = nullptr; return rccl_fail(c, "ncclCommInitRank", rc); }
    c->comm_ranks = n_ranks; c->comm_rank = rank;
    return UF3_OK;
}

extern "C" int uf3_comm_destroy(uf3_ctx *c) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    if (!c->comm) return UF3_OK;
    RcclApi &r = rccl_api();
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    const int rc = r.CommDestroy ? r.CommDestroy(c->comm) : 0;
    c->comm = nullptr; c->comm_ranks = 0; c->comm_rank = -1;
    return rc ? rccl_fail(c, "ncclCommDestroy", rc) : UF3_OK;
}

extern "C" int uf3_comm_info(const uf3_ctx *c, int32_t *n_ranks, int32_t *rank) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    if (n_ranks) *n_ranks = c->comm ? c->comm_ranks : 0;
    if (rank) *rank = c->comm ? c->comm_rank : -1;
    return UF3_OK;
}

// in place, on the context's stream, asynchronous like every *_dev entry
extern "C" int uf3_allreduce_sum_f64(uf3_ctx *c, double *d_buf, int64_t n) {
    if (!c || (n > 0 && !d_buf) || n < 0) return fail(c, UF3_EINVAL, "uf3_allreduce_sum_f64: bad argument");
    if (!c->comm) return fail(c, UF3_EINVAL, "uf3_allreduce_sum_f64: no communicator (uf3_comm_init)");
    if (n == 0) return UF3_OK;
    RcclApi &r = rccl_api();
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = r.AllReduce(d_buf, d_buf, (size_t)n, /* ncclFloat64 */ 8, /* ncclSum */ 0, c->comm, c->stream);
    return rc ? rccl_fail(c, "ncclAllReduce", rc) : UF3_OK;
}

extern "C" int uf3_gram_allreduce(uf3_ctx *c, double *d_packed, int64_t n) { return uf3_allreduce_sum_f64(c, d_packed, n); }

// ------------------------------------------------------------------------------ gram
static int ensure_frag(uf3_ctx *c) {
    if (c->frag_ready) return UF3_OK;
    HIPCHK(c, c->frag.ensure(sizeof(int) * 64 * 4 * 2));
    hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, c->stream, c->frag.as<int>());
    int tab[512];
    HIPCHK(c, hipMemcpyAsync(tab, c->frag.p, sizeof(tab), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    bool seen[256] = {false};
    for (int q = 0; q < 256; q++) {
        int r = tab[2 * q], cc = tab[2 * q + 1];
        if (r < 0 || r > 15 || cc < 0 || cc > 15 || seen[r * 16 + cc])
            return fail(c, UF3_EHIP, "unexpected v_mfma_f64_16x16x4 fragment layout");
        seen[r * 16 + cc] = true;
    }
    c->frag_ready = true;
    return UF3_OK;
}

// k_gram_tiled over all rows and columns (rowmap == nullptr), or over the rows rowmap[seg[0] .. seg[0] + seg[1]) -- seg on the
// device -- and the n_cols columns of colmap: n_rows is then the bound the grid covers, n_rows_plan what a segment is expected
// to hold (the chunks are cut for it; workgroups past the segment leave at once).  Accumulates the upper triangle.
static int launch_gram_tiled(uf3_ctx *c, const double *dx, const double *dy, int64_t n_rows, int64_t n_rows_plan, int n_feat,
                             int64_t ld, double *d_gram, double *d_ord, const int *rowmap, const int *seg, const int *colmap,
                             int n_cols) {
    hipStream_t st = c->stream;
    const int np = (n_cols + 63) / 64;
    // LDS-tiled kernel: patches of 64 x 64 packed into workgroups (at most four patches on at most four column ranges)
    int ps = -1;
    // (one slot per species' column subset + the full matrix: a basis of UF3_MAX_SPECIES species never evicts)
    const int n_slots = UF3_MAX_SPECIES + 1;
    for (int q = 0; q < n_slots; q++) if (c->gram_plan_np[q] == np) ps = q;
    if (ps < 0) {
        ps = c->gram_plan_next;
        c->gram_plan_next = (ps + 1) % n_slots;
        std::vector<GramBlock> plan;
        auto fresh = [&]() { GramBlock g; memset(&g, 0, sizeof g); for (int q = 0; q < 4; q++) g.range[q] = -1; return g; };
        auto slot_of = [&](GramBlock &g, int r, bool add) {
            for (int q = 0; q < 4; q++) if (g.range[q] == r) return q;
            if (add) for (int q = 0; q < 4; q++) if (g.range[q] < 0) { g.range[q] = r; return q; }
            return -1;
        };
        auto n_waves = [&](const GramBlock &g) { int n = 0; for (int w = 0; w < 4; w++) n += g.kind[w] != 0; return n; };
        auto fits = [&](const GramBlock &g, int pa, int pb) {
            if (n_waves(g) >= 4) return false;
            int free_slots = 0, need = 0;
            bool has_a = false, has_b = false;
            for (int q = 0; q < 4; q++) { free_slots += g.range[q] < 0; has_a |= g.range[q] == pa; has_b |= g.range[q] == pb; }
            need = (has_a ? 0 : 1) + ((has_b || pb == pa) ? 0 : 1);
            return need <= free_slots;
        };
        auto add_patch = [&](GramBlock &g, int pa, int pb) {
            const int w = n_waves(g);
            g.wa[w] = slot_of(g, pa, true); g.wb[w] = slot_of(g, pb, true);
            g.kind[w] = pa == pb ? 2 : 1;
            if (pa == pb) g.ord_mask |= 1 << g.wa[w];
        };
        // diagonal patches four at a time (equal work per wave)
        for (int p0 = 0; p0 < np; p0 += 4) {
            GramBlock g = fresh();
            for (int p = p0; p < std::min(np, p0 + 4); p++) add_patch(g, p, p);
            plan.push_back(g);
        }
        // off-diagonal patches four to a workgroup on at most four ranges: greedy (to the patch that opened the workgroup add
        // the ones that need no new range, then one, then two), over the natural order and a few shuffles of it; the fewest
        // workgroups win (every workgroup costs the time of a full patch, however many of its waves have one)
        std::vector<std::pair<int, int>> pairs;
        for (int pa = 0; pa < np; pa++) for (int pb = pa + 1; pb < np; pb++) pairs.push_back({pa, pb});
        std::vector<GramBlock> best;
        unsigned lcg = 12345u;
        for (int attempt = 0; attempt < 64; attempt++) {
            std::vector<std::pair<int, int>> left = pairs;
            if (attempt)
                for (size_t q = left.size(); q > 1; q--) { lcg = lcg * 1664525u + 1013904223u; std::swap(left[q - 1], left[(lcg >> 8) % q]); }
            std::vector<GramBlock> blocks;
            while (!left.empty()) {
                GramBlock g = fresh();
                add_patch(g, left[0].first, left[0].second);
                left.erase(left.begin());
                while (n_waves(g) < 4 && !left.empty()) {
                    int pick = -1;
                    for (int need = 0; need <= 2 && pick < 0; need++)
                        for (size_t q = 0; q < left.size() && pick < 0; q++) {
                            const int n_new = (slot_of(g, left[q].first, false) < 0) + (slot_of(g, left[q].second, false) < 0);
                            if (n_new == need && fits(g, left[q].first, left[q].second)) pick = (int)q;
                        }
                    if (pick < 0) break;
                    add_patch(g, left[pick].first, left[pick].second);
                    left.erase(left.begin() + pick);
                }
                blocks.push_back(g);
            }
            if (best.empty() || blocks.size() < best.size()) best = blocks;
            if (best.size() * 4 < pairs.size() + 4) break;                  // (cannot get better)
        }
        for (auto &g : best) plan.push_back(g);
        for (auto &g : plan)
            for (int q = 0; q < 4; q++) if (g.range[q] < 0) g.range[q] = g.range[0];
        c->gram_plan_np[ps] = 0;
        HIPCHK(c, hipDeviceSynchronize());                             // (a launch -- on this or an earlier stream of the context -- may still be reading the plan this one replaces)
        HIPCHK(c, c->gram_tiles[ps].ensure(sizeof(GramBlock) * plan.size()));
        HIPCHK(c, hipMemcpyAsync(c->gram_tiles[ps].p, plan.data(), sizeof(GramBlock) * plan.size(), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));
        c->gram_plan_np[ps] = np; c->gram_plan_blocks[ps] = (int)plan.size();
    }
    // row chunks: about two rounds of workgroups over the chip's resident slots (two per CU)
    const int bpc = c->gram_plan_blocks[ps];
    // (measured flat between two and twelve workgroups per CU)
    const int want_chunks = std::max(8, (c->n_cu * 4 + bpc - 1) / bpc / 8 * 8);
    int64_t rpc = (n_rows_plan + want_chunks - 1) / want_chunks;
    rpc = std::max<int64_t>(256, (rpc + GT_KS - 1) / GT_KS * GT_KS);
    const int chunks = (int)((n_rows + rpc - 1) / rpc), chunks8 = (chunks + 7) / 8 * 8;
    if (rowmap)
        hipLaunchKernelGGL(k_gram_tiled<true>, dim3(bpc * chunks8), dim3(256), 0, st, dx, (d_ord && dy) ? dy : nullptr, n_rows, n_feat,
                           ld, (int)rpc, bpc, (const GramBlock *)c->gram_tiles[ps].p, c->frag.as<int>(), d_gram, d_ord, rowmap, seg, colmap,
                           n_cols);
    else
        hipLaunchKernelGGL(k_gram_tiled<false>, dim3(bpc * chunks8), dim3(256), 0, st, dx, (d_ord && dy) ? dy : nullptr, n_rows, n_feat,
                           ld, (int)rpc, bpc, (const GramBlock *)c->gram_tiles[ps].p, c->frag.as<int>(), d_gram, d_ord,
                           (const int *)nullptr, (const int *)nullptr, (const int *)nullptr, n_feat);
    HIPCHK(c, hipGetLastError());
    return UF3_OK;
}

extern "C" int uf3_gram_dev(uf3_ctx *c, const double *dx, const double *dy, int64_t n_rows, int32_t n_feat, int64_t ld,
                            int accumulate, double *d_gram, double *d_ord) {
    uf3_env_refresh();
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    if (!dx || !d_gram || n_feat < 1 || ld < n_feat || n_rows < 0) return fail(c, UF3_EINVAL, "uf3_gram: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_frag(c);
    if (rc) return rc;
    hipStream_t st = c->stream;
    if (!accumulate) {
        HIPCHK(c, hipMemsetAsync(d_gram, 0, 8 * (size_t)n_feat * n_feat, st));
        if (d_ord) HIPCHK(c, hipMemsetAsync(d_ord, 0, 8 * (size_t)n_feat, st));
    }
    if (n_rows == 0) return UF3_OK;
    // (up to two column ranges the patches are mostly padding: one wave per 32 x 32 tile with direct loads is faster there --
    // measured 0.62 against 0.81 ms at F = 73, 960 k rows)
    // (and below ~64 k rows the row chunks get too short for the slab pipeline: 0.24 against 0.21 ms at 30 001 x 425)
    if (n_feat > 128 && n_rows >= 65536 && !uf3_env("UF3_GRAM_DIRECT")) {
        // LDS-tiled kernel: patches of 64 x 64 packed into workgroups (at most four patches on at most four column ranges)
        Timed tm(c, T_GRAM);
        rc = launch_gram_tiled(c, dx, dy, n_rows, n_rows, n_feat, ld, d_gram, d_ord, nullptr, nullptr, nullptr, n_feat);
        if (rc) return rc;
        hipLaunchKernelGGL(k_gram_mirror, dim3((n_feat + 255) / 256, n_feat), dim3(256), 0, st, d_gram, n_feat);
        HIPCHK(c, hipGetLastError());
        return UF3_OK;
    }
    if (n_feat <= GS_LDW && n_rows >= 16384 && !uf3_env("UF3_GRAM_DIRECT")) {
        // narrow matrices: slabs of 32 rows through LDS, every row read once (k_gram_small); about eight workgroups per CU
        const int64_t want_blocks = (int64_t)c->n_cu * 8;
        int64_t rpb = (n_rows + want_blocks - 1) / want_blocks;
        rpb = std::max<int64_t>(4 * GS_ROWS, (rpb + GS_ROWS - 1) / GS_ROWS * GS_ROWS);
        const int64_t blocks = (n_rows + rpb - 1) / rpb;
        Timed tm(c, T_GRAM);
        hipLaunchKernelGGL(k_gram_small, dim3((unsigned)blocks), dim3(256), 0, st, dx, (d_ord && dy) ? dy : nullptr, n_rows, n_feat, ld,
                           rpb, c->frag.as<int>(), d_gram, d_ord);
        hipLaunchKernelGGL(k_gram_mirror, dim3((n_feat + 255) / 256, n_feat), dim3(256), 0, st, d_gram, n_feat);
        HIPCHK(c, hipGetLastError());
        return UF3_OK;
    }
    // tile-pair table of the direct kernel: built and uploaded when n_feat changes (its own buffer and key: a fit that sends
    // its energy rows here and its force rows to the tiled kernel keeps both plans)
    const int nt = (n_feat + 31) / 32;
    size_t np = (size_t)nt * (nt + 1) / 2;
    np = (np + 3) / 4 * 4;
    if (c->gram_direct_feat != n_feat) {
        std::vector<int> tij;
        for (int i = 0; i < nt; i++) for (int j = i; j < nt; j++) tij.push_back(i);
        while (tij.size() < np) tij.push_back(-1);
        for (int i = 0; i < nt; i++) for (int j = i; j < nt; j++) tij.push_back(j);
        while (tij.size() < 2 * np) tij.push_back(-1);
        c->gram_direct_feat = 0;
        HIPCHK(c, c->gram_tij.ensure(8 * np));
        HIPCHK(c, hipMemcpyAsync(c->gram_tij.p, tij.data(), 8 * np, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));                           // (tij is a local)
        c->gram_direct_feat = n_feat;
    }
    int *d_ti = c->gram_tij.as<int>(), *d_tj = d_ti + np;
    // enough row chunks to fill the chip, each a multiple of 4 rows
    int blocks_xy = (int)(np / 4);
    int want_chunks = std::max(1, (c->n_cu * 8 + blocks_xy - 1) / blocks_xy);
    int64_t rpc = (n_rows + want_chunks - 1) / want_chunks;
    rpc = std::max<int64_t>(64, (rpc + 3) / 4 * 4);
    int chunks = (int)((n_rows + rpc - 1) / rpc);
    {
        Timed tm(c, T_GRAM);
        const int chunks8 = (chunks + 7) / 8 * 8;                  // whole rounds over the 8 XCDs
        hipLaunchKernelGGL(k_gram_mfma, dim3(blocks_xy * chunks8), dim3(256), 0, st, dx, (d_ord && dy) ? dy : nullptr, n_rows, n_feat, ld,
                           (int)rpc, blocks_xy, d_ti, d_tj, c->frag.as<int>(), d_gram, d_ord);
        hipLaunchKernelGGL(k_gram_mirror, dim3((n_feat + 255) / 256, n_feat), dim3(256), 0, st, d_gram, n_feat);
    }
    HIPCHK(c, hipGetLastError());
    return UF3_OK;
}

extern "C" int uf3_gram_force_rows_dev(uf3_basis *b, const double *d_x_f, const double *d_y_f, const int32_t *d_z, int64_t n_atoms,
                                       int64_t ld, int accumulate, double *d_gram, double *d_ord) {
    uf3_env_refresh();
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    uf3_ctx *c = b->ctx;
    const int F = b->host.F, S = b->host.S;
    if (!d_x_f || !d_gram || !d_z || n_atoms < 0 || ld < F || 3 * n_atoms > INT32_MAX)
        return fail(c, UF3_EINVAL, "uf3_gram_force_rows_dev: bad argument");
    int widest = 0;
    for (int sp = 0; sp < S; sp++) widest = std::max(widest, b->sp_ncols[sp]);
    // One species, a narrow matrix or short segments: the plain product (same result; the zero columns are multiplied).  Also
    // when no species leaves out at least one 64-column range's worth of columns.
    if (S < 2 || F <= 128 || 3 * n_atoms / S < 65536 || (widest + 63) / 64 >= (F + 63) / 64 || uf3_env("UF3_GRAM_DENSE"))
        return uf3_gram_dev(c, d_x_f, d_y_f, 3 * n_atoms, F, ld, accumulate, d_gram, d_ord);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_frag(c);
    if (rc) return rc;
    hipStream_t st = c->stream;
    if (!accumulate) {
        HIPCHK(c, hipMemsetAsync(d_gram, 0, 8 * (size_t)F * F, st));
        if (d_ord) HIPCHK(c, hipMemsetAsync(d_ord, 0, 8 * (size_t)F, st));
    }
    if (n_atoms == 0) return UF3_OK;
    // row lists by species (readable two slabs past the end: the kernel fetches row indices a slab ahead)
    const size_t n_list = 3 * (size_t)n_atoms + 2 * GT_KS + 8;
    HIPCHK(c, c->sp_rows.ensure(sizeof(int) * n_list));
    HIPCHK(c, c->sp_seg.ensure(sizeof(int) * 3 * UF3_MAX_SPECIES));
    int *seg = c->sp_seg.as<int>(), *cursor = seg + 2 * UF3_MAX_SPECIES, *rows = c->sp_rows.as<int>();
    Timed tm(c, T_GRAM);
    HIPCHK(c, hipMemsetAsync(seg, 0, sizeof(int) * 3 * UF3_MAX_SPECIES, st));
    // (the whole list: k_species_rows skips atoms whose species is outside the basis, so entries between the end of the segments
    // and 3 n_atoms would otherwise be uninitialised row indices for the kernel's one-slab-ahead prefetch; 12 bytes per atom)
    HIPCHK(c, hipMemsetAsync(rows, 0, sizeof(int) * n_list, st));
    const unsigned nblk = (unsigned)((n_atoms + SR_ATOMS - 1) / SR_ATOMS);
    hipLaunchKernelGGL(k_species_rows, dim3(nblk), dim3(256), 0, st, (const BasisDev *)b->dev, d_z, n_atoms, 0, seg, cursor, rows);
    hipLaunchKernelGGL(k_species_rows, dim3(nblk), dim3(256), 0, st, (const BasisDev *)b->dev, d_z, n_atoms, 1, seg, cursor, rows);
    for (int sp = 0; sp < S; sp++) {
        if (b->sp_ncols[sp] == 0) continue;
        rc = launch_gram_tiled(c, d_x_f, d_y_f, 3 * n_atoms, 3 * n_atoms / S, F, ld, d_gram, d_ord, rows, seg + 2 * sp,
                               b->d_sp_cols + (size_t)sp * F, b->sp_ncols[sp]);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_gram_mirror, dim3((F + 255) / 256, F), dim3(256), 0, st, d_gram, F);
    HIPCHK(c, hipGetLastError());
    return UF3_OK;
}

extern "C" int uf3_gram(uf3_ctx *c, const double *x, const double *y, int64_t n_rows, int32_t n_feat, int64_t ld,
                        int accumulate, double *gram, double *ord) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    if (!x || !gram || n_feat < 1 || ld < n_feat || n_rows < 0) return fail(c, UF3_EINVAL, "uf3_gram: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    size_t bx = 8 * (size_t)n_rows * ld, bg = 8 * (size_t)n_feat * n_feat, bo = 8 * (size_t)n_feat;
    HIPCHK(c, c->stage_out2.ensure(bx + 8 * (size_t)n_rows + 64));
    HIPCHK(c, c->stage_out.ensure(bg + bo));
    double *dxp = c->stage_out2.as<double>(), *dyp = dxp + (size_t)n_rows * ld;
    double *dg = c->stage_out.as<double>(), *dord = dg + (size_t)n_feat * n_feat;
    if (n_rows) HIPCHK(c, hipMemcpyAsync(dxp, x, bx, hipMemcpyHostToDevice, c->stream));
    if (y && n_rows) HIPCHK(c, hipMemcpyAsync(dyp, y, 8 * (size_t)n_rows, hipMemcpyHostToDevice, c->stream));
    if (accumulate) {
        HIPCHK(c, hipMemcpyAsync(dg, gram, bg, hipMemcpyHostToDevice, c->stream));
        if (ord) HIPCHK(c, hipMemcpyAsync(dord, ord, bo, hipMemcpyHostToDevice, c->stream));
    }
    int rc = uf3_gram_dev(c, dxp, y ? dyp : nullptr, n_rows, n_feat, ld, accumulate, dg, (ord && y) ? dord : nullptr);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(gram, dg, bg, hipMemcpyDeviceToHost, c->stream));
    if (ord && y) HIPCHK(c, hipMemcpyAsync(ord, dord, bo, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return UF3_OK;
}

// ------------------------------------------------------------------------------ fit bookkeeping
extern "C" int uf3_fit_rows_dev(uf3_ctx *c, int32_t n_frames, int32_t n_feat, double *d_x_e, const double *d_counts,
                                const double *d_y_e, const double *d_y_f, int64_t n_y_f, const int64_t *d_frozen,
                                const double *d_c_frozen, int32_t n_frozen, double *d_moments) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    if (n_frames < 0 || n_feat < 1 || !d_x_e || !d_counts || !d_y_e || !d_moments || n_y_f < 0 || n_frozen < 0 ||
        (n_frozen && (!d_frozen || !d_c_frozen)))
        return fail(c, UF3_EINVAL, "uf3_fit_rows_dev: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t f_blocks = d_y_f ? (n_y_f + 4095) / 4096 : 0;
    if (n_frames + f_blocks == 0) return UF3_OK;
    hipLaunchKernelGGL(k_fit_rows, dim3((unsigned)(n_frames + f_blocks)), dim3(256), 0, c->stream, n_frames, n_feat, d_x_e, d_counts,
                       d_y_e, d_y_f, n_y_f, d_frozen, d_c_frozen, n_frozen, d_moments);
    HIPCHK(c, hipGetLastError());
    return UF3_OK;
}

extern "C" int uf3_fit_pack_dev(uf3_ctx *c, int32_t n_feat, const double *d_flat, const int64_t *d_keep, int32_t n_keep,
                                const int64_t *d_frozen, const double *d_c_frozen, int32_t n_frozen, double n_energy_rows,
                                double n_force_rows, double *d_packed) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    if (n_feat < 1 || n_keep < 0 || n_keep > n_feat || !d_flat || (n_keep && !d_keep) || !d_packed || n_frozen < 0 ||
        (n_frozen && (!d_frozen || !d_c_frozen)))
        return fail(c, UF3_EINVAL, "uf3_fit_pack_dev: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    // (n_keep == 0, every column frozen: the six moments are the whole packed buffer)
    hipLaunchKernelGGL(k_fit_pack, dim3((unsigned)std::max(n_keep, 1), 2), dim3(256), 0, c->stream, n_feat, d_flat, d_keep, n_keep, d_frozen,
                       d_c_frozen, n_frozen, n_energy_rows, n_force_rows, d_packed);
    HIPCHK(c, hipGetLastError());
    return UF3_OK;
}

// ------------------------------------------------------------------------------ neighbour debug
static int neighbors_impl(uf3_basis *b, const uf3_frames *fr, const double *pos, const int32_t *z,
                          int64_t *pair_count, int64_t *pair_ij, double *pair_geo, int64_t pair_cap, int64_t *n3_count,
                          int64_t *n3_ij, int64_t n3_cap) {
    uf3_env_refresh();
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    uf3_ctx *c = b->ctx;
    if (!fr || fr->n_frames != 1) return fail(c, UF3_EINVAL, "the neighbour queries take exactly one frame");
    int natoms = 0;
    int rc = upload_frames(c, fr, pos, z, natoms);
    if (rc) return rc;
    Prepared P;
    rc = prepare(b, fr, (const double *)c->stage_cur, c->d_stage_z, false, P);
    if (rc) return rc;
    int np = b->host.P;
    std::vector<long long> counts(np + 2, 0);
    std::vector<long long> tuples;
    std::vector<double> geo;
    long long cap = 0;
    for (int pass = 0; pass < 2; pass++) {
        const size_t head = 8 * (size_t)(np + 2), ncap = (size_t)std::max<long long>(1, cap);
        HIPCHK(c, c->dbg.ensure(head + 24 * ncap + (pair_geo ? 32 * ncap : 0)));
        long long *d_counts = c->dbg.as<long long>(), *d_tuples = d_counts + np + 2;
        double *d_geo = pair_geo ? (double *)(d_tuples + 3 * ncap) : nullptr;
        HIPCHK(c, hipMemsetAsync(d_counts, 0, head, c->stream));
        hipLaunchKernelGGL(k_debug_pairs, dim3(natoms), dim3(64), 0, c->stream, b->dev, P.geoms, P.frame_of, P.cl,
                           (const double *)c->stage_cur, P.spec, natoms, d_counts, d_tuples, cap, d_geo);
        HIPCHK(c, hipMemcpyAsync(counts.data(), d_counts, head, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (pass == 0) { cap = counts[np + 1]; if (cap == 0) break; continue; }
        tuples.resize(3 * (size_t)cap);
        HIPCHK(c, hipMemcpy(tuples.data(), d_tuples, 24 * (size_t)cap, hipMemcpyDeviceToHost));
        if (pair_geo) {
            geo.resize(4 * (size_t)cap);
            HIPCHK(c, hipMemcpy(geo.data(), d_geo, 32 * (size_t)cap, hipMemcpyDeviceToHost));
        }
    }
    rc = check_flags(c);
    if (rc) return rc;
    if (pair_count) for (int p = 0; p < np; p++) pair_count[p] = counts[p];
    if (n3_count) *n3_count = counts[np];
    struct T3 { long long p, i, j, at; };
    std::vector<T3> v((size_t)cap);
    for (long long q = 0; q < cap; q++) v[q] = {tuples[3 * q], tuples[3 * q + 1], tuples[3 * q + 2], q};
    std::sort(v.begin(), v.end(), [](const T3 &a, const T3 &bb) {
        return a.p != bb.p ? a.p < bb.p : (a.i != bb.i ? a.i < bb.i : a.j < bb.j);
    });
    std::vector<long long> cur(np + 1, 0);
    for (const T3 &t : v) {
        if (t.p < np) {
            if (cur[t.p] < pair_cap) {
                const long long o = t.p * pair_cap + cur[t.p];
                if (pair_ij) { pair_ij[2 * o] = t.i; pair_ij[2 * o + 1] = t.j; }
                if (pair_geo) for (int k = 0; k < 4; k++) pair_geo[4 * o + k] = geo[4 * (size_t)t.at + k];
            }
        } else if (n3_ij && cur[np] < n3_cap) { n3_ij[2 * cur[np]] = t.i; n3_ij[2 * cur[np] + 1] = t.j; }
        cur[t.p]++;
    }
    return UF3_OK;
}

// The 3-body neighbour lists the LAST featurizer / evaluator call on this context built and consumed (MODE 0's build_n3_list or
// k_build_n3 -- not the separate walk of k_debug_pairs): per atom the entry count and, per entry, the neighbour's reference
// supercell index (image_rank * N + atom) in list order (species, then supercell index).  Test infrastructure: valid only straight
// after a synchronised call, before anything else runs on the context.
extern "C" int uf3_n3_lists_debug(uf3_basis *b, int64_t natoms, int64_t *cap_out, int32_t *counts, int32_t *sidx, int64_t sidx_cap) {
    if (!b) return fail(nullptr, UF3_EINVAL, "null basis");
    uf3_ctx *c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->n3_last_cap <= 0 || natoms != c->n3_last_natoms || !c->n3_cnt.p || !c->n3_dbl.p)
        return fail(c, UF3_EINVAL, "uf3_n3_lists_debug: no 3-body lists of a batch of this size in the workspace");
    const int cap = c->n3_last_cap;
    if (cap_out) *cap_out = cap;
    if (counts) HIPCHK(c, hipMemcpy(counts, c->n3_cnt.p, sizeof(int) * (size_t)natoms, hipMemcpyDeviceToHost));
    if (sidx) {
        if (sidx_cap < cap) return fail(c, UF3_EINVAL, "uf3_n3_lists_debug: sidx_cap below the list capacity");
        std::vector<N3Entry> ent((size_t)natoms * cap);
        HIPCHK(c, hipMemcpy(ent.data(), c->n3_dbl.p, sizeof(N3Entry) * ent.size(), hipMemcpyDeviceToHost));
        for (int64_t a = 0; a < natoms; a++)
            for (int q = 0; q < cap; q++) sidx[a * sidx_cap + q] = ent[(size_t)a * cap + q].sidx;
    }
    return UF3_OK;
}

extern "C" int uf3_neighbors_debug(uf3_basis *b, const uf3_frames *fr, const double *pos, const int32_t *z,
                                   int64_t *pair_count, int64_t *pair_ij, int64_t pair_cap, int64_t *n3_count,
                                   int64_t *n3_ij, int64_t n3_cap) {
    return neighbors_impl(b, fr, pos, z, pair_count, pair_ij, nullptr, pair_cap, n3_count, n3_ij, n3_cap);
}

extern "C" int uf3_pair_geometry(uf3_basis *b, const uf3_frames *fr, const double *pos, const int32_t *z,
                                 int64_t *pair_count, int64_t *pair_ij, double *pair_geo, int64_t pair_cap) {
    if (pair_cap > 0 && !pair_geo) return fail(b ? b->ctx : nullptr, UF3_EINVAL, "uf3_pair_geometry: null pair_geo");
    return neighbors_impl(b, fr, pos, z, pair_count, pair_ij, pair_cap > 0 ? pair_geo : nullptr, pair_cap, nullptr, nullptr, 0);
}

// ------------------------------------------------------------------------------ dense helpers (small frames)
extern "C" int uf3_distance_matrix(uf3_ctx *c, const double *a, int64_t na, const double *bb, int64_t nb, double *out) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    if (na < 0 || nb < 0 || (na && !a) || (nb && !bb)) return fail(c, UF3_EINVAL, "uf3_distance_matrix: bad argument");
    const size_t n = (size_t)na * (size_t)nb;
    if (!n) return UF3_OK;
    if (!out) return fail(c, UF3_EINVAL, "uf3_distance_matrix: null out");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->dbg.ensure(24 * (size_t)(na + nb) + 8 * n));
    double *d_a = c->dbg.as<double>(), *d_b = d_a + 3 * na, *d_out = d_b + 3 * nb;
    HIPCHK(c, hipMemcpyAsync(d_a, a, 24 * (size_t)na, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_b, bb, 24 * (size_t)nb, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_distance_matrix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_a, (long long)na, d_b,
                       (long long)nb, d_out);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, d_out, 8 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return UF3_OK;
}

extern "C" int uf3_direction_cosines(uf3_ctx *c, const double *sup_pos, int64_t n_sup, const int64_t *i_where,
                                     const int64_t *j_where, const double *rij, int64_t n_d, int64_t n_atoms, double *out) {
    if (!c) return fail(nullptr, UF3_EINVAL, "null ctx");
    if (n_sup < 0 || n_d < 0 || n_atoms < 0) return fail(c, UF3_EINVAL, "uf3_direction_cosines: bad argument");
    const size_t n = (size_t)n_atoms * 3 * (size_t)n_d;
    if (!n) return UF3_OK;
    if (!sup_pos || !i_where || !j_where || !rij || !out) return fail(c, UF3_EINVAL, "uf3_direction_cosines: null argument");
    for (int64_t q = 0; q < n_d; q++)
        if (i_where[q] < 0 || i_where[q] >= n_sup || j_where[q] < 0 || j_where[q] >= n_sup)
            return fail(c, UF3_EINVAL, "uf3_direction_cosines: index outside the supercell");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->dbg.ensure(24 * (size_t)n_sup + 24 * (size_t)n_d + 8 * n));
    double *d_pos = c->dbg.as<double>();
    long long *d_i = (long long *)(d_pos + 3 * n_sup), *d_j = d_i + n_d;
    double *d_r = (double *)(d_j + n_d), *d_out = d_r + n_d;
    HIPCHK(c, hipMemcpyAsync(d_pos, sup_pos, 24 * (size_t)n_sup, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_i, i_where, 8 * (size_t)n_d, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_j, j_where, 8 * (size_t)n_d, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_r, rij, 8 * (size_t)n_d, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_direction_cosines, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_pos, d_i, d_j, d_r,
                       (long long)n_d, (long long)n_atoms, d_out);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, d_out, 8 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return UF3_OK;
}

#ifdef UF3_PHASE_TIMING
// experiments only: read (and clear) the in-kernel phase timers
extern "C" int uf3_debug_phase(unsigned long long *out16) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16);
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z));
    return 0;
}
#endif
