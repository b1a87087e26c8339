"""Kernel timeline of one rank's share of the decomposed 50 k-atom ternary frame (uf3_eval_centres_dev at world W):
    rocprofv3 --kernel-trace -d gpurun_out/dectr -o t --output-format csv -- python tools/experiments/decomposed_trace.py 8
then `python tools/experiments/decomposed_trace.py --report gpurun_out/dectr` prints the last call's launches."""
import sys, os, glob, csv, ctypes as C
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    last = len(names) - 1 - names[::-1].index(next(n for n in names[::-1] if "k_frame_sum" in n))
    first = last
    while first > 0 and "k_frame_bins" not in names[first] and "k_prepare_small" not in names[first]:
        first -= 1
    t0 = int(rows[first]["Start_Timestamp"]); prev = None
    for r in rows[first:last + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f'{r["Kernel_Name"][:48]:48s} start {(s - t0) / 1e3:8.2f} us  dur {(e - s) / 1e3:7.2f} us  gap {((s - prev) / 1e3) if prev else 0:6.2f}')
        prev = e
    print(f"total {(prev - t0) / 1e3:.1f} us")
    sys.exit(0)
import numpy as np, torch
from uf3_amd import synthetic, _lib
from uf3_amd.forcefield import calculator
from uf3_amd.regression import least_squares as ls
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
ctx = _lib.get_context(0)
atoms = synthetic.lattice_frame("bcc", (25, 25, 40), 3.165, [23, 42, 74], 4000)
basis = synthetic.notebook_basis(['V', 'Mo', 'W'])
model = ls.WeightedLinearModel(basis)
coeff = np.random.default_rng(11).normal(0, 0.05, basis.n_feats); coeff[basis.col_idx] = 0.0
model.coefficients = coeff
calc = calculator.UFCalculator(model)
db = _lib.device_basis(basis, ctx)
batch = _lib.FrameBatch([atoms])
d_pos = torch.from_numpy(batch.pos).to(dev); d_z = torch.from_numpy(batch.z).to(dev)
d_e = torch.empty((1,), dtype=torch.float64, device=dev); d_f = torch.empty((batch.n_atoms, 3), dtype=torch.float64, device=dev)
d_v = torch.empty((6,), dtype=torch.float64, device=dev)
lo, hi = 0, (batch.n_atoms + world - 1) // world
for _ in range(20):
    ctx.check(ctx.lib.uf3_eval_centres_dev(db.handle, C.byref(batch.struct), C.c_void_p(d_pos.data_ptr()), C.c_void_p(d_z.data_ptr()),
                                           _lib._p(calc._c1), _lib._p(calc._c2), _lib._p(calc._c3), lo, hi, C.c_void_p(d_e.data_ptr()),
                                           C.c_void_p(d_f.data_ptr()), C.c_void_p(d_v.data_ptr())))
torch.cuda.synchronize()
print("world", world, "block", hi - lo, "energy share", d_e.item())
