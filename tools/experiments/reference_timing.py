"""Build container only (needs /root/reference and the stand-ins of tests/golden/_standins): wall time of the
reference's own featurizer next to the CPU restatement (oracle/uf3_oracle.c) on identical small inputs, one process.
numba is absent here, so the reference's jitted loops run interpreted: its 3-body times are pessimistic."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "_standins"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
warnings.simplefilter("ignore")
import numpy as np
import ase
from uf3.data import composition as rc
from uf3.representation import bspline as rb, process as rp
from oracle import oracle as O
from uf3_amd import synthetic

def ref_geom(a):
    return ase.Atoms(numbers=a.get_atomic_numbers(), positions=a.get_positions(), cell=np.array(a.get_cell()), pbc=a.get_pbc())

def timed(f, *args):
    t0 = time.perf_counter(); out = f(*args); return time.perf_counter() - t0, out

mine = synthetic.notebook_basis(['W'])
kw = dict(r_min_map=mine.r_min_map, r_max_map=mine.r_max_map, resolution_map=mine.resolution_map,
          leading_trim=mine.leading_trim, trailing_trim=mine.trailing_trim)
ref_basis = rb.BSplineBasis(rc.ChemicalSystem(['W'], 3), **kw)
fz = rp.BasisFeaturizer(ref_basis)
ob = O.OracleBasis(mine)
import json
cases = []
for reps, forces in (((2, 2, 2), True), ((3, 3, 3), True), ((4, 4, 4), False)):
    a = synthetic.lattice_frame("bcc", reps, 3.165, [74], seed=5)
    g = ref_geom(a)
    n = len(a)
    f_in = np.zeros((3, n)) if forces else None
    t_ref, rows = timed(lambda: fz.evaluate_configuration(g, name="x", energy=0.0, forces=f_in))
    t_or, ref = timed(O.featurize, ob, a)
    xe = np.array(rows[("x", "energy")][1:])
    err = np.abs(xe - ref["xe"]).max() / np.abs(ref["xe"]).max()
    cases.append(dict(atoms=n, rows="energy + force rows" if forces else "energy row only (the restatement computes both)",
                      reference_s=round(t_ref, 3), restatement_s=round(t_or, 6), ratio=round(t_ref / t_or, 1),
                      energy_rows_agree_to=float(err)))
    print(f"{n:4d} atoms, {'energy + force rows' if forces else 'energy row only  '}: reference {t_ref:8.2f} s, "
          f"restatement {t_or * 1e3:7.2f} ms (computes both), ratio {t_ref / t_or:8.0f}, energy rows agree to {err:.1e}")

full = [c for c in cases if c["rows"].startswith("energy + force")]
out = dict(what="wall time of the reference's BasisFeaturizer.evaluate_configuration (uf3 v0.4.0, imported from /root/reference behind the "
                "stand-ins of tests/golden/_standins) next to oracle/uf3_oracle.c on identical rattled bcc-W cells, notebook 2+3-body basis "
                "(F = 73), one process each, build container",
           caveat="numba absent: the reference's three jitted loops ran interpreted, so its 3-body times are pessimistic; the reference "
                  "cannot run 10k-atom frames at all (583 GB M x M distance matrix, BASELINE.md section 3): the ratio is an ESTIMATE "
                  "taken at the largest size with force rows it finishes here",
           host=dict(cpu_count=os.cpu_count(), python=sys.version.split()[0], numpy=np.__version__),
           cases=cases,
           reference_to_port_ratio=full[-1]["ratio"], ratio_taken_at_atoms=full[-1]["atoms"])
path = os.path.join(ROOT, "profiles", "reference_calibration.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
