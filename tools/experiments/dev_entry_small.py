import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from uf3_amd import synthetic, _lib
from uf3_amd.forcefield import calculator
from uf3_amd.regression import least_squares as ls
from uf3_amd.representation import process
basis = synthetic.notebook_basis(['Mo', 'W'])
model = ls.WeightedLinearModel(basis)
coeff = np.random.default_rng(4).normal(0, 0.05, basis.n_feats); coeff[basis.col_idx] = 0.0
model.coefficients = coeff
calc = calculator.UFCalculator(model)
ctx = _lib.get_context(0); db = _lib.device_basis(basis, ctx)
dev = torch.device("cuda:0")
for reps in ((4, 4, 4), (2, 2, 2), (3, 2, 2)):
    atoms = synthetic.lattice_frame("bcc", reps, 3.165, [42, 74], seed=5)
    e, f, _ = calc.evaluate_frames([atoms]); e, f, _ = calc.evaluate_frames([atoms])
    batch = _lib.FrameBatch([atoms])
    d_pos = torch.from_numpy(batch.pos).to(dev); d_z = torch.from_numpy(batch.z).to(dev)
    d_e = torch.empty((1,), dtype=torch.float64, device=dev); d_f = torch.empty((batch.n_atoms, 3), dtype=torch.float64, device=dev)
    for _ in range(2):
        ctx.check(ctx.lib.uf3_eval_dev(db.handle, C.byref(batch.struct), C.c_void_p(d_pos.data_ptr()), C.c_void_p(d_z.data_ptr()),
                                       _lib._p(calc._c1), _lib._p(calc._c2), _lib._p(calc._c3), C.c_void_p(d_e.data_ptr()), C.c_void_p(d_f.data_ptr())))
    ctx.check(ctx.lib.uf3_ctx_synchronize(ctx.handle)) if hasattr(ctx, "handle") else torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(reps, batch.n_atoms, "energy equal", d_e.item() == e[0], "forces equal", np.array_equal(d_f.cpu().numpy(), f))
    fz = process.BasisFeaturizer(basis, device=0)
    xe, xf, _ = fz.featurize_frames([atoms])
    d_xe = torch.zeros((1, basis.n_feats), dtype=torch.float64, device=dev); d_xf = torch.zeros((3 * batch.n_atoms, basis.n_feats), dtype=torch.float64, device=dev)
    for _ in range(2):
        d_xe.zero_()
        ctx.check(ctx.lib.uf3_featurize_dev(db.handle, C.byref(batch.struct), C.c_void_p(d_pos.data_ptr()), C.c_void_p(d_z.data_ptr()),
                                            C.c_void_p(d_xe.data_ptr()), C.c_void_p(d_xf.data_ptr())))
    torch.cuda.synchronize()
    print("   rows equal", np.array_equal(d_xf.cpu().numpy().reshape(xf.shape), xf), "energy row close", np.abs(d_xe.cpu().numpy().reshape(xe.shape) - xe).max())
