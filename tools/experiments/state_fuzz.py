"""State fuzz: ONE process, the shared per-device context, a few bases and frames of different density / size / periodicity,
and a long seeded random sequence of calls -- featurize batches, evaluate (MD route on / off, atoms nudged between calls, species
or cell changed now and then), fit accumulation -- each checked against the oracle.  Looks for results that depend on what the
context did before (capacities, persistent lists, staging blocks, cached coefficients).
    python tools/experiments/state_fuzz.py [n_ops] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from uf3_amd import synthetic, pipeline, _lib
from uf3_amd.data.atoms import Atoms
from uf3_amd.representation import process
from uf3_amd.regression import least_squares as ls
from uf3_amd.forcefield import calculator

n_ops = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
TOL = 1e-9


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max())) if a.size else 0.0


bases = {"W": synthetic.notebook_basis(['W']), "WMo": synthetic.notebook_basis(['Mo', 'W']),
         "WMo_lead0": synthetic.notebook_basis(['Mo', 'W'], lead3=0)}
zs = {"W": [74], "WMo": [42, 74], "WMo_lead0": [42, 74]}


def make_frame(kind, numbers, seed):
    if kind == "bcc":
        return synthetic.lattice_frame("bcc", (4, 4, 4), 3.165, numbers, seed=seed)
    if kind == "bcc_dense":
        return synthetic.lattice_frame("bcc", (5, 5, 5), 2.45, numbers, seed=seed, rattle=0.03, strain=0.0)
    if kind == "fcc":
        return synthetic.lattice_frame("fcc", (3, 3, 4), 4.05, numbers, seed=seed, rattle=0.05)
    if kind == "slab":
        a = synthetic.lattice_frame("bcc", (4, 4, 3), 3.2, numbers, seed=seed)
        cell = np.array(a.cell, float); cell[2, 2] += 12.0
        return Atoms(numbers=a.numbers, positions=a.positions, cell=cell, pbc=[True, True, False])
    if kind == "cluster":
        a = synthetic.lattice_frame("bcc", (3, 3, 3), 3.1, numbers, seed=seed)
        return Atoms(numbers=a.numbers, positions=a.positions, cell=np.eye(3) * 40.0, pbc=[False, False, False])
    raise ValueError(kind)


kinds = ["bcc", "bcc_dense", "fcc", "slab", "cluster"]
frames = {(b, k): make_frame(k, zs[b], 100 + 7 * i) for b in bases for i, k in enumerate(kinds)}
obs = {b: O.OracleBasis(bases[b]) for b in bases}
fzs = {b: process.BasisFeaturizer(bases[b]) for b in bases}
row_cache = {}


def oracle_rows(b, k):
    if (b, k) not in row_cache:
        row_cache[(b, k)] = O.featurize(obs[b], frames[(b, k)])
    return row_cache[(b, k)]


models, calcs = {}, {}
for b in bases:
    coeff = np.random.default_rng(5).normal(0, 0.05, bases[b].n_feats)
    coeff[bases[b].col_idx] = 0.0
    m = ls.WeightedLinearModel(bases[b]); m.coefficients = coeff
    models[b] = m
    calcs[(b, 0.0)] = calculator.UFCalculator(m, md_skin=0.0)
    calcs[(b, 0.5)] = calculator.UFCalculator(m, md_skin=0.5)
    calcs[(b, None)] = calculator.UFCalculator(m)
walkers = {}          # (basis, kind) -> current Atoms of an "MD" run
worst, bad = 0.0, 0
for op in range(n_ops):
    b = list(bases)[int(rng.integers(len(bases)))]
    what = rng.choice(["featurize", "evaluate", "evaluate", "evaluate", "fit"])
    if what == "featurize":
        ks = [kinds[int(i)] for i in rng.integers(0, len(kinds), int(rng.integers(1, 4)))]
        x_e, x_f, off = fzs[b].featurize_frames([frames[(b, k)] for k in ks])
        err = 0.0
        for i, k in enumerate(ks):
            ref = oracle_rows(b, k)
            err = max(err, rel(x_e[i], ref["xe"]), rel(x_f[off[i]:off[i + 1]], ref["xf"]))
        desc = f"featurize {b} {ks}"
    elif what == "evaluate":
        k = kinds[int(rng.integers(len(kinds)))]
        skin = [0.0, 0.5, None][int(rng.integers(3))]
        at = walkers.get((b, k), frames[(b, k)])
        r = rng.random()
        if r < 0.6:        # a small MD-like move
            at = Atoms(numbers=at.numbers, positions=at.positions + rng.normal(0, 0.02, at.positions.shape), cell=at.cell, pbc=at.pbc)
        elif r < 0.7:      # a large move: lists must be rebuilt
            at = Atoms(numbers=at.numbers, positions=at.positions + rng.normal(0, 0.25, at.positions.shape), cell=at.cell, pbc=at.pbc)
        elif r < 0.8 and len(zs[b]) > 1:    # species shuffled
            at = Atoms(numbers=rng.permutation(at.numbers), positions=at.positions, cell=at.cell, pbc=at.pbc)
        elif r < 0.9 and any(at.pbc):       # a strained cell
            s = np.eye(3) + rng.normal(0, 0.01, (3, 3))
            at = Atoms(numbers=at.numbers, positions=at.positions @ s, cell=np.array(at.cell) @ s, pbc=at.pbc)
        walkers[(b, k)] = at
        virial = bool(rng.random() < 0.3) and all(at.pbc)
        out = calcs[(b, skin)].evaluate_frames([at], virial=virial)
        e_ref, f_ref = O.evaluate(obs[b], at, models[b].coefficients)
        err = max(rel(out[0][0], e_ref), rel(out[1], f_ref))
        desc = f"evaluate {b} {k} skin {skin} virial {virial}"
    else:
        ks = [kinds[int(i)] for i in rng.integers(0, len(kinds), int(rng.integers(2, 5)))]
        acc = pipeline.DeviceFitAccumulator(models[b], fzs[b], max_atoms_per_chunk=int(rng.choice([150, 400, 100000])))
        fr = [frames[(b, k)] for k in ks]
        en = rng.normal(-8.0, 1.0, len(fr)) * np.array([len(f) for f in fr])
        fo = [rng.normal(0, 0.5, (len(f), 3)) for f in fr]
        for attempt in range(4):
            # (asynchronous calls: a batch denser than anything the context has seen overflows its capacities, the verdict
            # arrives with the next synchronisation and the work is repeated -- what pipeline.fit_frames does)
            try:
                acc.add_frames(fr, en, fo)
                p = acc.pieces()
                break
            except _lib.RetryError:
                acc.reset()
        xe = np.array([oracle_rows(b, k)["xe"] for k in ks])
        xf = np.concatenate([oracle_rows(b, k)["xf"].reshape(-1, bases[b].n_feats) for k in ks])
        keep = np.setdiff1d(np.arange(bases[b].n_feats), np.asarray(bases[b].col_idx, dtype=int))
        n_at = xe[:, :len(zs[b])].sum(axis=1)
        xe_n = xe / n_at[:, None]
        ge = (xe_n.T @ xe_n)[np.ix_(keep, keep)]
        gf = (xf.T @ xf)[np.ix_(keep, keep)]
        err = max(rel(p["gram_e"], ge), rel(p["gram_f"], gf))
        desc = f"fit {b} {ks}"
    worst = max(worst, err)
    if not err < TOL:
        bad += 1
        print(f"op {op:4d}: {desc}: {err:.2e}   <-- MISMATCH", flush=True)
    elif op % 20 == 0:
        print(f"op {op:4d}: {desc}: {err:.1e}", flush=True)
print(f"{n_ops} operations, worst {worst:.2e}, mismatches {bad}")
sys.exit(1 if bad else 0)
