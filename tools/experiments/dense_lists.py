import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from oracle import oracle as O
from uf3_amd import synthetic, _lib
from uf3_amd.representation import process
from _util import rel_err, worst_elementwise
for els, nums in ((['W'], [74]), (['Mo', 'W'], [42, 74])):
    for lead in (3, 0):
        basis = synthetic.notebook_basis(els, lead3=lead)
        for a, reps in ((1.75, (4, 4, 4)), (1.6, (5, 4, 4))):
            atoms = synthetic.lattice_frame("bcc", reps, a, nums, 5, rattle=0.05)
            fz = process.BasisFeaturizer(basis)
            x_e, x_f, _ = fz.featurize_frames([atoms])
            ref = O.featurize(O.OracleBasis(basis), atoms)
            pairs, n3 = fz.neighbor_indices(atoms)
            per = np.bincount(n3[:, 0]).max() if len(n3) else 0
            print(els, lead, a, "max 3-body neighbours", per, "modes", hex(fz._dev()[1].featurizer_modes),
                  "rows", rel_err(x_e[0], ref["xe"]), rel_err(x_f, ref["xf"].reshape(x_f.shape)),
                  worst_elementwise(x_f, ref["xf"].reshape(x_f.shape)))
