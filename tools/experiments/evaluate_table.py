"""BasisFeaturizer.evaluate on a table of small frames: batched (the product) against the frame-by-frame loop over
evaluate_configuration that the reference's evaluate performs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pandas as pd
from uf3_amd import synthetic
from uf3_amd.representation import process

basis = synthetic.notebook_basis(['W'])
fz = process.BasisFeaturizer(basis)
rng = np.random.default_rng(0)
n_frames = 256
geoms = [synthetic.lattice_frame("bcc", (4, 4, 4), 3.165, [74], seed=k) for k in range(n_frames)]
f = {c: [rng.normal(0, 1, len(g)).tolist() for g in geoms] for c in ('fx', 'fy', 'fz')}
df = pd.DataFrame(dict(geometry=geoms, energy=rng.normal(-1000, 1, n_frames), **f))
fz.evaluate(df.iloc[:8], progress=False)
t0 = time.perf_counter(); out = fz.evaluate(df, progress=False); t1 = time.perf_counter()
t2 = time.perf_counter()
m = {}
for name, row in df.iloc[:64].iterrows():
    m.update(fz.evaluate_configuration(row["geometry"], name, row["energy"], [row["fx"], row["fy"], row["fz"]], "energy"))
ref = fz.arrange_features_dataframe(m)
t3 = time.perf_counter()
print(f"{n_frames} frames x 128 atoms, F={len(out.columns) - 1}: batched evaluate {n_frames / (t1 - t0):.0f} frames/s; "
      f"frame-by-frame loop {64 / (t3 - t2):.0f} frames/s; same rows: {np.allclose(out.iloc[:len(ref)].to_numpy(), ref.to_numpy(), rtol=1e-12, atol=1e-12)}")
