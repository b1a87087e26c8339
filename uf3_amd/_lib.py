"""
ctypes binding of libuf3hip.so (include/uf3_hip.h) -- the only compute path.

There is deliberately no CPU fallback: if the HIP library is missing or no
MI355X is visible, every call that needs arithmetic raises ``HipUnavailable``.
"""
import ctypes as C
import sys
import importlib.util
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UF3_LIB_PATH", os.path.join(_HERE, "csrc", "libuf3hip.so"))

EXPORTS = ["uf3_ctx_create", "uf3_ctx_destroy", "uf3_ctx_set_stream", "uf3_ctx_synchronize",
           "uf3_last_error", "uf3_build_id", "uf3_ctx_timing_reset", "uf3_ctx_timing_read",
           "uf3_ctx_use_own_stream", "uf3_basis_create", "uf3_basis_destroy", "uf3_basis_featurizer_modes",
           "uf3_featurize", "uf3_featurize_dev", "uf3_gram", "uf3_gram_dev",
           "uf3_eval", "uf3_eval_dev", "uf3_eval_virial", "uf3_eval_virial_dev", "uf3_eval_atoms", "uf3_eval_atoms_dev",
           "uf3_eval_centres", "uf3_eval_centres_dev",
           "uf3_neighbors_debug", "uf3_n3_lists_debug", "uf3_fit_rows_dev", "uf3_fit_pack_dev", "uf3_gram_force_rows_dev",
           "uf3_pair_geometry", "uf3_distance_matrix", "uf3_direction_cosines",
           "uf3_ctx_md_skin", "uf3_ctx_md_stats",
           "uf3_featurize_ld_dev", "uf3_fit_create", "uf3_fit_destroy", "uf3_fit_reset", "uf3_fit_add", "uf3_fit_pack", "uf3_fit_info", "uf3_fit_use_flat", "uf3_fit_first_chunk",
           "uf3_comm_unique_id", "uf3_comm_init", "uf3_comm_destroy", "uf3_comm_info", "uf3_allreduce_sum_f64", "uf3_gram_allreduce"]


SOURCES = ("uf3_hip.hip", "uf3_kernels.h", "uf3_feat3.h", "uf3_device.h", os.path.join("..", "..", "include", "uf3_hip.h"))


def source_build_id(csrc_dir=None):
    """What ``uf3_build_id()`` of a library compiled from the tree's sources returns (the Makefile's recipe)."""
    import hashlib
    csrc_dir = csrc_dir or os.path.join(_HERE, "csrc")
    h = hashlib.sha256()
    for name in SOURCES:
        with open(os.path.join(csrc_dir, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_id():
    """The source hash the loaded library was compiled from."""
    return load().uf3_build_id().decode()


class HipUnavailable(RuntimeError):
    pass


class UF3Error(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libuf3hip error {code}: {message}")
        self.code = code


class SpeciesError(UF3Error):
    """A frame contains an element outside the basis (UF3_ESPECIES)."""


class RetryError(UF3Error):
    """UF3_ERETRY: an earlier asynchronous ``uf3_featurize_dev`` call overflowed its neighbour capacities (or met atoms
    far outside their cell, whose 3-body force rows need the launches with the reference's image-range rule); its outputs
    are invalid, the context has adapted: repeat the work since the last synchronisation."""


class BasisSpec(C.Structure):
    _fields_ = [("n_species", C.c_int32), ("species_z", C.c_void_p),
                ("n_pairs", C.c_int32), ("pair_z", C.c_void_p), ("pair_nk", C.c_void_p),
                ("pair_knots", C.c_void_p), ("pair_rmin", C.c_void_p), ("pair_rmax", C.c_void_p),
                ("pair_col", C.c_void_p), ("lead2", C.c_int32), ("trail2", C.c_int32),
                ("n_trios", C.c_int32), ("trio_z", C.c_void_p), ("trio_nk", C.c_void_p),
                ("trio_knots", C.c_void_p), ("trio_col", C.c_void_p), ("trio_ncol", C.c_void_p),
                ("trio_lut", C.c_void_p), ("n_feat", C.c_int32), ("r_cut", C.c_double)]


class Frames(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("atom_offsets", C.c_void_p), ("cells", C.c_void_p),
                ("pbc", C.c_void_p)]


_lib = None
_lock = threading.Lock()


def _share_hip_runtime_with_torch():
    """One HIP runtime per process, whichever of torch / uf3_amd is imported first.

    PyTorch-ROCm wheels bundle their own ``libamdhip64.so``.  If this library (linked against the system ROCm) has
    initialised the system runtime before torch is imported, torch's libraries bind to the already loaded system
    runtime and then find no device (``No HIP GPUs are available``).  With torch imported first everything binds to the
    bundled runtime, which works -- so when a torch wheel with a bundled runtime is installed, that runtime is
    loaded here first.  torch itself is not imported."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen libuf3hip.so (built in-tree by ``__graft_entry__.build()``)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipUnavailable(
                f"{LIB_PATH} not found: build it with `make -C uf3_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "uf3_amd has no CPU fallback.")
        _share_hip_runtime_with_torch()
        lib = C.CDLL(LIB_PATH)
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        lib.uf3_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        lib.uf3_ctx_destroy.argtypes = [vp]
        lib.uf3_ctx_destroy.restype = None
        lib.uf3_ctx_set_stream.argtypes = [vp, vp]
        lib.uf3_ctx_use_own_stream.argtypes = [vp]
        lib.uf3_ctx_synchronize.argtypes = [vp]
        lib.uf3_last_error.argtypes = [vp]
        lib.uf3_last_error.restype = C.c_char_p
        lib.uf3_build_id.argtypes = []
        lib.uf3_build_id.restype = C.c_char_p
        lib.uf3_ctx_timing_reset.argtypes = [vp, C.c_int]
        lib.uf3_ctx_timing_read.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64), C.POINTER(dbl),
                                            C.POINTER(dbl), C.POINTER(dbl)]
        lib.uf3_ctx_md_skin.argtypes = [vp, dbl]
        lib.uf3_fit_create.argtypes = [vp, C.c_int, i64, vp, vp, i32, C.POINTER(vp)]
        lib.uf3_fit_destroy.argtypes = [vp]
        lib.uf3_fit_destroy.restype = None
        lib.uf3_fit_reset.argtypes = [vp]
        lib.uf3_fit_add.argtypes = [vp, i32, vp, vp, vp, C.c_int, vp, vp, vp, vp]
        lib.uf3_fit_pack.argtypes = [vp, vp, i32, C.c_int, vp]
        lib.uf3_fit_info.argtypes = [vp, C.POINTER(i64), C.POINTER(dbl), C.POINTER(dbl)]
        lib.uf3_fit_use_flat.argtypes = [vp, vp]
        lib.uf3_fit_first_chunk.argtypes = [vp, dbl]
        lib.uf3_comm_unique_id.argtypes = [vp, vp]
        lib.uf3_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
        lib.uf3_comm_destroy.argtypes = [vp]
        lib.uf3_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
        lib.uf3_allreduce_sum_f64.argtypes = [vp, vp, i64]
        lib.uf3_gram_allreduce.argtypes = [vp, vp, i64]
        lib.uf3_ctx_md_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
        lib.uf3_basis_create.argtypes = [vp, C.POINTER(BasisSpec), C.POINTER(vp)]
        lib.uf3_basis_destroy.argtypes = [vp]
        lib.uf3_basis_destroy.restype = None
        lib.uf3_basis_featurizer_modes.argtypes = [vp, vp]
        for name in ("uf3_featurize", "uf3_featurize_dev"):
            getattr(lib, name).argtypes = [vp, C.POINTER(Frames), vp, vp, vp, vp]
        lib.uf3_featurize_ld_dev.argtypes = [vp, C.POINTER(Frames), vp, vp, vp, vp, i64]
        for name in ("uf3_gram", "uf3_gram_dev"):
            getattr(lib, name).argtypes = [vp, vp, vp, i64, i32, i64, C.c_int, vp, vp]
        for name in ("uf3_eval", "uf3_eval_dev"):
            getattr(lib, name).argtypes = [vp, C.POINTER(Frames), vp, vp, vp, vp, vp, vp, vp]
        for name in ("uf3_eval_virial", "uf3_eval_virial_dev"):
            getattr(lib, name).argtypes = [vp, C.POINTER(Frames), vp, vp, vp, vp, vp, vp, vp, vp]
        for name in ("uf3_eval_atoms", "uf3_eval_atoms_dev", "uf3_eval_centres", "uf3_eval_centres_dev"):
            getattr(lib, name).argtypes = [vp, C.POINTER(Frames), vp, vp, vp, vp, vp, i64, i64, vp, vp, vp]
        lib.uf3_neighbors_debug.argtypes = [vp, C.POINTER(Frames), vp, vp, vp, vp, i64, vp, vp, i64]
        lib.uf3_n3_lists_debug.argtypes = [vp, i64, vp, vp, vp, i64]
        lib.uf3_gram_force_rows_dev.argtypes = [vp, vp, vp, vp, i64, i64, i32, vp, vp]
        lib.uf3_pair_geometry.argtypes = [vp, C.POINTER(Frames), vp, vp, vp, vp, vp, i64]
        lib.uf3_distance_matrix.argtypes = [vp, vp, i64, vp, i64, vp]
        lib.uf3_direction_cosines.argtypes = [vp, vp, i64, vp, vp, vp, i64, i64, vp]
        lib.uf3_fit_rows_dev.argtypes = [vp, i32, i32, vp, vp, vp, vp, i64, vp, vp, i32, vp]
        lib.uf3_fit_pack_dev.argtypes = [vp, i32, vp, vp, i32, vp, vp, i32, dbl, dbl, vp]
        _lib = lib
        return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _addr(a):
    """Address of an array's buffer as a plain int (what a ``c_void_p`` argument takes): a fifth of the cost of
    ``a.ctypes.data_as`` -- on an MD-step call of the evaluator the nine pointer arguments were a third of the host time."""
    return None if a is None else a.__array_interface__["data"][0]


class Context:
    """One HIP device + stream + grow-only workspace.  Not thread-safe: one per thread/device."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.uf3_ctx_create(int(device), C.byref(h))
        if rc:
            msg = self.lib.uf3_last_error(None).decode()
            raise HipUnavailable(f"cannot create a HIP context on device {device}: {msg}")
        self.handle = h
        self.device = int(device)
        self._pid = os.getpid()

    def check(self, rc):
        if rc:
            msg = self.lib.uf3_last_error(self.handle).decode()
            raise {2: SpeciesError, 6: RetryError}.get(rc, UF3Error)(rc, msg)

    def set_stream(self, stream_ptr):
        """Bind the context to a HIP stream (0 / None: the null stream); returns the stream bound before, so that
        temporary users can put it back (``None`` before the first call = the context's own stream)."""
        prev = getattr(self, "_stream", None)
        self.check(self.lib.uf3_ctx_set_stream(self.handle, C.c_void_p(stream_ptr or 0)))
        self._stream = stream_ptr or 0
        return prev

    def use_own_stream(self):
        self.check(self.lib.uf3_ctx_use_own_stream(self.handle))
        self._stream = None

    def restore_stream(self, prev):
        if prev is None:
            self.use_own_stream()
        else:
            self.set_stream(prev)

    def synchronize(self):
        self.check(self.lib.uf3_ctx_synchronize(self.handle))

    def timing_reset(self, enable=True):
        self.check(self.lib.uf3_ctx_timing_reset(self.handle, int(enable)))

    def timing_read(self):
        f, n, nb, g, e = C.c_double(), C.c_int64(), C.c_double(), C.c_double(), C.c_double()
        self.check(self.lib.uf3_ctx_timing_read(self.handle, C.byref(f), C.byref(n), C.byref(nb),
                                                C.byref(g), C.byref(e)))
        return dict(featurize_ms=f.value, featurize_launches=n.value, neighbor_ms=nb.value,
                    gram_ms=g.value, eval_ms=e.value)

    def md_skin(self, skin):
        """MD route of the evaluator: keep per-atom superset neighbour lists out to ``r_cut + skin`` (Angstrom) on the device
        and reuse them while no atom has moved more than ``skin / 2``; 0 switches it off (every call rebuilds, as the
        reference's calculator does).  Results do not depend on when the lists were built."""
        skin = float(skin)
        if getattr(self, "_md_skin", 0.0) != skin:
            self.check(self.lib.uf3_ctx_md_skin(self.handle, skin))
            self._md_skin = skin

    # ---- RCCL through the library itself (uf3_comm_*): no torch.distributed in the data path ----------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        self.check(self.lib.uf3_comm_unique_id(self.handle, buf))
        return buf.raw

    def comm_init(self, n_ranks, rank, unique_id):
        """Collective: every rank calls it with the id rank 0 drew (``comm_unique_id``) -- carried over by whatever the host
        has (``parallel.native_comm`` uses torch.distributed's store, a file works as well)."""
        self.check(self.lib.uf3_comm_init(self.handle, int(n_ranks), int(rank), C.c_char_p(bytes(unique_id))))

    def comm_destroy(self):
        self.check(self.lib.uf3_comm_destroy(self.handle))

    def comm_info(self):
        n, r = C.c_int32(), C.c_int32()
        self.check(self.lib.uf3_comm_info(self.handle, C.byref(n), C.byref(r)))
        return n.value, r.value

    def allreduce_sum(self, device_ptr, n):
        """Sum ``n`` doubles at ``device_ptr`` over the ranks, in place, on the context's stream (asynchronous)."""
        self.check(self.lib.uf3_allreduce_sum_f64(self.handle, C.c_void_p(device_ptr), int(n)))

    def md_stats(self):
        b, s, r = C.c_int64(), C.c_int64(), C.c_int64()
        self.check(self.lib.uf3_ctx_md_stats(self.handle, C.byref(b), C.byref(s), C.byref(r)))
        return dict(builds=b.value, steps=s.value, redone=r.value)

    def __del__(self):
        try:
            if getattr(self, "handle", None) and os.getpid() == self._pid:
                self.lib.uf3_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


_contexts = {}


def get_context(device=None):
    """Per-process, per-device shared context (device defaults to LOCAL_RANK or 0)."""
    if device is None:
        device = int(os.environ.get("UF3_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    key = (os.getpid(), int(device))
    if key not in _contexts:
        _contexts[key] = Context(device)
    return _contexts[key]


def flatten_basis(basis):
    """BSplineBasis -> (BasisSpec, keep-alive arrays).  Pure host bookkeeping."""
    from uf3_amd.data.composition import atomic_numbers as Z
    cs = basis.chemical_system
    els = list(cs.element_list)
    sizes, offsets = basis.get_interaction_partitions()
    pairs = list(cs.interactions_map[2])
    trios = list(cs.interactions_map.get(3, [])) if cs.degree > 2 else []
    keep = {}
    keep["species_z"] = np.array([Z[e] for e in els], dtype=np.int32)
    keep["pair_z"] = np.array([[Z[a], Z[b]] for a, b in pairs], dtype=np.int32).reshape(-1, 2)
    pk = [np.asarray(basis.knots_map[p], dtype=np.float64) for p in pairs]
    keep["pair_nk"] = np.array([len(k) for k in pk], dtype=np.int32)
    keep["pair_knots"] = np.ascontiguousarray(np.concatenate(pk))
    keep["pair_rmin"] = np.array([float(basis.r_min_map[p]) for p in pairs])
    keep["pair_rmax"] = np.array([float(basis.r_max_map[p]) for p in pairs])
    keep["pair_col"] = np.array([int(offsets[p]) for p in pairs], dtype=np.int32)
    tk = [[np.asarray(k, dtype=np.float64) for k in basis.knots_map[t]] for t in trios]
    keep["trio_z"] = np.array([[Z[a], Z[b], Z[c]] for a, b, c in trios], dtype=np.int32).reshape(-1, 3)
    keep["trio_nk"] = np.array([[len(k) for k in ks] for ks in tk], dtype=np.int32).reshape(-1, 3)
    keep["trio_knots"] = (np.ascontiguousarray(np.concatenate([k for ks in tk for k in ks]))
                          if trios else np.zeros(1))
    keep["trio_col"] = np.array([int(offsets[t]) for t in trios], dtype=np.int32)
    keep["trio_ncol"] = np.array([int(sizes[t]) for t in trios], dtype=np.int32)
    luts = []
    for t in trios:
        lut, w = basis.column_sources(t)
        hit = lut >= 0
        if not np.allclose(w[hit], 1.0, rtol=0, atol=1e-14):
            raise ValueError(f"symmetry weights of {t} do not fold to 1; unsupported template")
        luts.append(lut.astype(np.int32))
    keep["trio_lut"] = np.ascontiguousarray(np.concatenate(luts)) if trios else np.zeros(1, np.int32)
    s = BasisSpec()
    s.n_species, s.n_pairs, s.n_trios = len(els), len(pairs), len(trios)
    for name in ("species_z", "pair_z", "pair_nk", "pair_knots", "pair_rmin", "pair_rmax", "pair_col",
                 "trio_z", "trio_nk", "trio_knots", "trio_col", "trio_ncol", "trio_lut"):
        setattr(s, name, _p(keep[name]))
    s.lead2, s.trail2 = int(basis.leading_trim[2]), int(basis.trailing_trim[2])
    s.n_feat = int(np.sum(basis.get_feature_partition_sizes()))
    s.r_cut = float(basis.r_cut)
    keep["pairs"], keep["trios"] = pairs, trios
    return s, keep


class DeviceBasis:
    """Device-resident tables of one BSplineBasis on one context."""

    def __init__(self, basis, ctx=None):
        self.ctx = ctx or get_context()
        self.spec, self._keep = flatten_basis(basis)
        self.n_feat = self.spec.n_feat
        self.n_species = self.spec.n_species
        self.pairs, self.trios = self._keep["pairs"], self._keep["trios"]
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.uf3_basis_create(self.ctx.handle, C.byref(self.spec), C.byref(h)))
        self.handle = h
        self._pid = os.getpid()

    @property
    def featurizer_modes(self):
        """Bit mask of the featurizer specialisations in use (include/uf3_hip.h: uf3_basis_featurizer_modes)."""
        mask = C.c_int32(0)
        self.ctx.check(self.ctx.lib.uf3_basis_featurizer_modes(self.handle, C.byref(mask)))
        return mask.value

    def __del__(self):
        try:
            if getattr(self, "handle", None) and os.getpid() == self._pid and self.ctx.handle:
                self.ctx.lib.uf3_basis_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class RawDeviceBasis:
    """Device tables from a hand-made description instead of a ``BSplineBasis``: what the module-level functions of
    ``uf3_amd.representation.distances`` / ``angles`` need -- pair ranges without a spline space, or 3-body knot sets
    whose raw L x M x N bins are the output columns (no symmetry fold, no template).

    species_z   ascending atomic numbers
    pairs       {(za, zb): (r_min, r_max)} for any subset of the S(S+1)/2 pairs (the others get an empty range)
    trios       list of ((zc, za, zb), [l_knots, m_knots, n_knots], lut) -- lut: int32 [L*M*N] raw bin -> column of the
                block or -1 -- in column order behind the one-body and pair columns
    r_cut       image range of the reference supercell to reproduce
    """

    def __init__(self, species_z, pairs=None, trios=(), r_cut=None, ctx=None):
        self.ctx = ctx or get_context()
        zs = [int(z) for z in species_z]
        assert zs == sorted(set(zs)) and zs, "species_z: ascending atomic numbers"
        pairs = {tuple(sorted((int(a), int(b)))): v for (a, b), v in (pairs or {}).items()}
        keep = {}
        keep["species_z"] = np.array(zs, dtype=np.int32)
        all_pairs = [(zs[i], zs[j]) for i in range(len(zs)) for j in range(i, len(zs))]
        self.pairs = all_pairs
        ranges = [pairs.get(p, (0.0, 0.0)) for p in all_pairs]
        keep["pair_z"] = np.array(all_pairs, dtype=np.int32).reshape(-1, 2)
        keep["pair_nk"] = np.full(len(all_pairs), 8, dtype=np.int32)
        # (a spline space is not needed for neighbour queries: the shortest legal knot vector over the range)
        keep["pair_knots"] = np.ascontiguousarray(np.concatenate(
            [np.repeat([max(float(lo), 0.0), max(float(hi), max(float(lo), 0.0) + 1.0)], 4) for lo, hi in ranges]))
        keep["pair_rmin"] = np.array([float(lo) for lo, _ in ranges])
        keep["pair_rmax"] = np.array([float(hi) for _, hi in ranges])
        col = len(zs)
        keep["pair_col"] = np.arange(col, col + 4 * len(all_pairs), 4, dtype=np.int32)
        col += 4 * len(all_pairs)
        trio_col, trio_ncol, luts, knots, nk = [], [], [], [], []
        for zt, ks, lut in trios:
            ks = [np.asarray(k, dtype=np.float64) for k in ks]
            lut = np.ascontiguousarray(lut, dtype=np.int32).ravel()
            assert lut.size == int(np.prod([len(k) - 4 for k in ks]))
            ncol = int(lut.max()) + 1 if lut.size and lut.max() >= 0 else 1
            trio_col.append(col)
            trio_ncol.append(ncol)
            col += ncol
            luts.append(lut)
            knots.extend(ks)
            nk.append([len(k) for k in ks])
        self.trio_col, self.trio_ncol = trio_col, trio_ncol
        keep["trio_z"] = np.array([list(zt) for zt, _, _ in trios], dtype=np.int32).reshape(-1, 3)
        keep["trio_nk"] = np.array(nk, dtype=np.int32).reshape(-1, 3)
        keep["trio_knots"] = np.ascontiguousarray(np.concatenate(knots)) if knots else np.zeros(1)
        keep["trio_col"] = np.array(trio_col, dtype=np.int32)
        keep["trio_ncol"] = np.array(trio_ncol, dtype=np.int32)
        keep["trio_lut"] = np.ascontiguousarray(np.concatenate(luts)) if luts else np.zeros(1, np.int32)
        s = BasisSpec()
        s.n_species, s.n_pairs, s.n_trios = len(zs), len(all_pairs), len(trios)
        for name in ("species_z", "pair_z", "pair_nk", "pair_knots", "pair_rmin", "pair_rmax", "pair_col",
                     "trio_z", "trio_nk", "trio_knots", "trio_col", "trio_ncol", "trio_lut"):
            setattr(s, name, _p(keep[name]))
        s.lead2, s.trail2 = 0, 0
        s.n_feat = col
        if r_cut is None:
            r_cut = max([float(hi) for _, hi in ranges] + [float(k[-1]) for k in knots] + [1e-6])
        s.r_cut = float(r_cut)
        self.spec, self._keep = s, keep
        self.n_feat, self.n_species = col, len(zs)
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.uf3_basis_create(self.ctx.handle, C.byref(s), C.byref(h)))
        self.handle = h
        self._pid = os.getpid()

    def __del__(self):
        try:
            if getattr(self, "handle", None) and os.getpid() == self._pid and self.ctx.handle:
                self.ctx.lib.uf3_basis_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


# Device tables are cached OUTSIDE the basis objects (ctypes handles cannot be pickled or deep-copied, and the
# reference's objects travel through multiprocessing / joblib).  Weak keys: a table set dies with its basis.  A basis
# bumps ``_tables_version`` whenever its knots / trims / templates change (BSplineBasis.update_knots,
# update_basis_functions), which retires the tables built for the old state.
_device_tables = weakref.WeakKeyDictionary()


def device_basis(basis, ctx=None):
    """Cached DeviceBasis of a BSplineBasis (re-created after fork, unpickling or a change of the basis)."""
    ctx = ctx or get_context()
    per_basis = _device_tables.get(basis)
    if per_basis is None:
        per_basis = _device_tables[basis] = {}
    key = (os.getpid(), ctx.device)
    version = getattr(basis, "_tables_version", 0)
    hit = per_basis.get(key)
    if hit is None or hit[0] != version:
        hit = per_basis[key] = (version, DeviceBasis(basis, ctx))
    return hit[1]


def drop_device_basis(basis):
    """Forget the device tables of a basis (they are rebuilt on the next use)."""
    _device_tables.pop(basis, None)


class FrameBatch:
    """Host-side frame metadata + concatenated positions / species of a list of Atoms."""

    def __init__(self, atoms_list, periodic=None):
        self.n_frames = len(atoms_list)
        if self.n_frames == 1:          # an MD step: no concatenations
            a = atoms_list[0]
            self.pos = np.ascontiguousarray(a.get_positions(), dtype=np.float64).reshape(-1, 3)
            self.z = np.ascontiguousarray(a.get_atomic_numbers(), dtype=np.int32)
            self.offsets = np.array([0, len(self.z)], dtype=np.int64)
            self.cells = np.ascontiguousarray(a.get_cell(), dtype=np.float64).reshape(1, 3, 3)
            self.pbc = np.zeros((1, 3), dtype=np.uint8)
            if periodic is not False:
                self.pbc[0, :] = a.get_pbc() if hasattr(a, "get_pbc") else a.pbc
        else:
            n = [len(a) for a in atoms_list]
            self.offsets = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
            self.pos = np.ascontiguousarray(np.concatenate(
                [np.asarray(a.get_positions(), dtype=np.float64).reshape(-1, 3) for a in atoms_list]))
            self.z = np.ascontiguousarray(np.concatenate(
                [np.asarray(a.get_atomic_numbers(), dtype=np.int32) for a in atoms_list]))
            self.cells = np.ascontiguousarray(np.array(
                [np.array(a.get_cell(), dtype=np.float64).reshape(3, 3) for a in atoms_list]))
            pbc = np.zeros((self.n_frames, 3), dtype=np.uint8)
            for k, a in enumerate(atoms_list):
                pbc[k, :] = np.asarray(a.get_pbc() if hasattr(a, "get_pbc") else a.pbc)
            if periodic is False:
                pbc[:] = 0
            self.pbc = pbc
        self.n_atoms = int(self.offsets[-1])
        self.struct = make_frames(self.offsets, self.cells, self.pbc)

    def refresh(self, atoms):
        """The same single frame again (an MD step): positions, species, cell and boundary flags copied into the existing
        arrays -- no allocations, the C struct stays.  False when the atom count differs (build a new batch)."""
        if self.n_frames != 1 or len(atoms) != self.n_atoms:
            return False
        pos = getattr(atoms, "positions", None)
        np.copyto(self.pos, pos if pos is not None else atoms.get_positions())
        num = getattr(atoms, "numbers", None)
        np.copyto(self.z, num if num is not None else atoms.get_atomic_numbers(), casting="unsafe")
        np.copyto(self.cells[0], atoms.get_cell())
        self.pbc[0, :] = atoms.get_pbc() if hasattr(atoms, "get_pbc") else atoms.pbc
        return True


def make_frames(offsets, cells, pbc):
    f = Frames()
    f.n_frames = len(offsets) - 1
    f.atom_offsets, f.cells, f.pbc = _addr(offsets), _addr(cells), _addr(pbc)
    f._keep = (offsets, cells, pbc)
    return f
