"""Hand-off launch pair (k_feat3_w -> k_featurize3<HO>) against the one-kernel k_featurize3 and the oracle.
   gpurun -- 'python tools/experiments/handoff_check.py'"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from uf3_amd import synthetic
from uf3_amd.representation import process
from oracle import oracle as O


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def rows(fz, frames, ho):
    os.environ["UF3_F3_HANDOFF"] = str(ho)
    try:
        x_e, x_f, _ = fz.featurize_frames(frames)
    finally:
        os.environ.pop("UF3_F3_HANDOFF", None)
    return x_e, x_f


cases = []
fr, ba = synthetic.config_c2(); cases.append(("c2 W 1024", ba, [fr]))
fr, ba = synthetic.config_c3(); cases.append(("c3 NeXe 4096 (wide window: hand-off not taken)", ba, [fr]))
wmo = synthetic.notebook_basis(['Mo', 'W'])
cases.append(("WMo 2000 x 3 frames", wmo, [synthetic.lattice_frame("bcc", (10, 10, 10), 3.165, [42, 74], s) for s in (1, 2, 3)]))
tern = synthetic.notebook_basis(['Mo', 'Nb', 'W'])
cases.append(("MoNbW 1024 + 432 atoms", tern, [synthetic.lattice_frame("bcc", (8, 8, 8), 3.2, [41, 42, 74], 5),
                                               synthetic.lattice_frame("bcc", (6, 6, 6), 3.2, [41, 42, 74], 6)]))
cases.append(("W fcc 864 (18 neighbours: capacity 24)", synthetic.notebook_basis(['W']), [synthetic.lattice_frame("fcc", (6, 6, 6), 3.9, [74], 7)]))
for name, basis, frames in cases:
    fz = process.BasisFeaturizer(basis)
    e0, f0 = rows(fz, frames, 0)
    e1, f1 = rows(fz, frames, 1)
    ob = O.OracleBasis(basis)
    ref = O.featurize(ob, frames[0])
    n0 = len(frames[0])
    print(f"{name}: handoff vs one-kernel: energy {rel(e1, e0):.2e} force {rel(f1, f0):.2e} | "
          f"vs oracle (frame 0): one-kernel {rel(f0[:n0], ref['xf']):.2e} handoff {rel(f1[:n0], ref['xf']):.2e} "
          f"energy {rel(e1[0], ref['xe']):.2e}")
