"""
``UFCalculator``: energy and forces of a fitted UF3 model on the MI355X.

Surface follows the reference's ``uf3/forcefield/calculator.py`` (:40-153, helpers
:490-573): ``UFCalculator(model)``, ASE's ``calculate(atoms, properties,
system_changes)`` protocol with ``results['energy'|'free_energy'|'forces'|'stress']``,
``get_potential_energy`` / ``get_forces``.  The evaluation is the featurizer's
traversal contracted with the coefficients on the fly (``uf3_eval`` in
``libuf3hip.so``): pair splines from the pair coefficient vectors, trio splines from
the decompressed L x M x N grids (``decompress_3B``), as the reference builds its
``ndsplines`` objects.  Stress is analytic (the strain derivative is accumulated in the same
pass, ``uf3_eval_virial``; the reference differentiates the energy numerically,
``calculator.py:399-404``).  ASE is optional: the class derives from ``ase``'s Calculator when
ASE is importable and is otherwise a duck-typed stand-alone.
"""
import ctypes as C

import os

import numpy as np

from uf3_amd import _lib

try:  # pragma: no cover - ASE is not installed on the build / GPU boxes
    from ase.calculators.calculator import Calculator as _Base, all_changes
except Exception:  # noqa: BLE001
    all_changes = ['positions', 'numbers', 'cell', 'pbc', 'initial_charges', 'initial_magmoms']

    class _Base:
        def __init__(self, **kwargs):
            self.results = {}
            self.atoms = None

        def calculate(self, atoms=None, properties=None, system_changes=all_changes):
            if atoms is not None:
                self.atoms = atoms.copy() if hasattr(atoms, "copy") else atoms

        def get_property(self, name, atoms=None, allow_calculation=True):
            self.calculate(atoms, [name], all_changes)
            return self.results[name]

        def get_potential_energy(self, atoms=None, force_consistent=False):
            return self.get_property('energy', atoms)

        def get_forces(self, atoms=None):
            return self.get_property('forces', atoms)

        def get_stress(self, atoms=None):
            return self.get_property('stress', atoms)


def coefficients_by_interaction(element_list, interactions_map, partition_sizes, coefficients):
    parts = np.array_split(coefficients, np.cumsum(partition_sizes)[:-1])
    n = len(element_list)
    solutions = {el: v for el, v in zip(element_list, parts[:n])}
    for idx, key in enumerate(list(interactions_map[2]) + list(interactions_map.get(3, []))):
        solutions[key] = parts[n + idx]
    return solutions


AUTO_MD_SKIN = 0.5        # Angstrom: the skin of ``md_skin="auto"``


class UFCalculator(_Base):
    implemented_properties = ['energy', 'forces', 'stress']

    def __init__(self, model, device=None, md_skin=None, **kwargs):
        """``md_skin`` (Angstrom): keep the device's neighbour lists between calls out to ``r_cut + md_skin`` and rebuild them
        only when an atom has moved more than ``md_skin / 2`` (``_lib.Context.md_skin``) -- what an MD or relaxation loop
        wants; 0 rebuilds everything on every call like the reference (``calculator.py:124-153``).  Default (``None``): the
        environment's ``UF3_MD_SKIN`` if set, else ``"auto"`` -- a skin of 0.5 Angstrom from the second consecutive call on
        the same single frame (atom count, cell, boundary conditions) on, nothing kept for one-off calls on changing structures.
        Results do not depend on the setting beyond rounding (1e-13), nor on when the lists were built."""
        super().__init__(**kwargs)
        self.bspline_config = model.bspline_config
        self.model = model
        self.device = device
        if md_skin is None:
            md_skin = os.environ.get("UF3_MD_SKIN", "auto")
        self.md_auto = isinstance(md_skin, str) and md_skin.strip().lower() == "auto"
        self.md_skin = AUTO_MD_SKIN if self.md_auto else float(md_skin)
        self._last_batch = None
        self._last_layout = None
        basis = self.bspline_config
        self.solutions = coefficients_by_interaction(basis.element_list, basis.interactions_map,
                                                     basis.partition_sizes, model.coefficients)
        self.pair_potentials = {p: np.asarray(self.solutions[p], dtype=float)
                                for p in basis.interactions_map[2]}
        self.trio_potentials = {}
        if basis.degree > 2:
            self.trio_potentials = {t: basis.decompress_3B(np.asarray(self.solutions[t], dtype=float), t)
                                    for t in basis.interactions_map[3]}
        self._c1 = np.ascontiguousarray([float(np.ravel(self.solutions[el])[0]) for el in basis.element_list])
        self._c2 = np.ascontiguousarray(np.concatenate([self.pair_potentials[p] for p in basis.interactions_map[2]]))
        self._c3 = (np.ascontiguousarray(np.concatenate([self.trio_potentials[t].ravel()
                                                         for t in basis.interactions_map[3]]))
                    if self.trio_potentials else np.zeros(1))
        self._pc = (_lib._addr(self._c1), _lib._addr(self._c2), _lib._addr(self._c3))     # (the arrays live as long as self)

    degree = property(lambda self: self.bspline_config.degree)
    element_list = property(lambda self: self.bspline_config.element_list)
    interactions_map = property(lambda self: self.bspline_config.interactions_map)
    r_min_map = property(lambda self: self.bspline_config.r_min_map)
    r_max_map = property(lambda self: self.bspline_config.r_max_map)
    r_cut = property(lambda self: self.bspline_config.r_cut)
    partition_sizes = property(lambda self: self.bspline_config.partition_sizes)
    coefficients = property(lambda self: self.model.coefficients)
    chemical_system = property(lambda self: self.bspline_config.chemical_system)

    def __repr__(self):
        return "\n".join(["UFCalculator:", repr(self.model)])

    def evaluate_frames(self, atoms_list, forces=True, virial=False):
        """Energies [n_frames], forces [sum N, 3] (and dE/d(strain) [n_frames, 6]) of a batch of frames."""
        ctx = _lib.get_context(self.device)
        db = _lib.device_basis(self.bspline_config, ctx)
        # (an MD loop hands over the same single frame again and again: its batch is kept and refreshed in place)
        batch = self._last_batch
        if len(atoms_list) != 1 or batch is None or not batch.refresh(atoms_list[0]):
            batch = _lib.FrameBatch(atoms_list)
            self._last_batch = batch if len(atoms_list) == 1 else None
        skin = self.md_skin
        if self.md_auto:        # lists are worth keeping once the same single frame comes back
            layout = (batch.n_atoms, batch.cells.tobytes(), batch.pbc.tobytes()) if batch.n_frames == 1 else None
            if layout is None or layout != self._last_layout:
                skin = 0.0
            self._last_layout = layout
        if getattr(ctx, "_md_skin", 0.0) != skin:
            ctx.md_skin(skin)
        e = np.empty(batch.n_frames)
        f = np.empty((batch.n_atoms, 3)) if forces else None
        addr = _lib._addr                     # (plain ints: the host side of an MD-step call is as long as its kernels)
        args = (db.handle, C.byref(batch.struct), addr(batch.pos), addr(batch.z), self._pc[0], self._pc[1], self._pc[2],
                addr(e), addr(f))
        if virial:
            v = np.empty((batch.n_frames, 6))
            rc = ctx.lib.uf3_eval_virial(*args, addr(v))
            if rc:
                ctx.check(rc)
            return e, f, batch.offsets, v
        rc = ctx.lib.uf3_eval(*args)
        if rc:
            ctx.check(rc)
        return e, f, batch.offsets

    def evaluate_atom_range(self, atoms, atom_begin, atom_end, forces=True, virial=False):
        """
        Share of atoms [atom_begin, atom_end) of one frame: (energy share, forces [N, 3] with the other
        atoms' rows zero, dE/d(strain) share [6] or None).  Shares over disjoint ranges covering the frame
        add up to ``evaluate_frames`` (``uf3_eval_atoms``; used by ``parallel.sharded_evaluate``).
        """
        ctx = _lib.get_context(self.device)
        db = _lib.device_basis(self.bspline_config, ctx)
        batch = _lib.FrameBatch([atoms])
        e = np.empty(1)
        f = np.empty((batch.n_atoms, 3)) if forces else None
        v = np.empty((1, 6)) if virial else None
        addr = _lib._addr
        ctx.check(ctx.lib.uf3_eval_atoms(db.handle, C.byref(batch.struct), addr(batch.pos), addr(batch.z), self._pc[0], self._pc[1],
                                         self._pc[2], int(atom_begin), int(atom_end), addr(e), addr(f), addr(v)))
        return float(e[0]), f, (v[0] if virial else None)

    def evaluate_centre_range(self, atoms, atom_begin, atom_end, forces=True, virial=False):
        """
        Share of the CENTRES [atom_begin, atom_end) of one frame: (energy share, forces [N, 3], dE/d(strain) share [6] or
        None).  Every triplet is evaluated once, at its centre inside the range, so the force array also carries what those
        triplets put on atoms outside it (the range's halo); all other rows are zero.  Shares over disjoint ranges covering
        the frame add up to ``evaluate_frames`` at a third of ``evaluate_atom_range``'s triplet work (``uf3_eval_centres``;
        what ``parallel.sharded_evaluate`` reduces).
        """
        ctx = _lib.get_context(self.device)
        skin = 0.0 if self.md_auto else self.md_skin            # (blocks of centres keep lists only when asked to)
        if getattr(ctx, "_md_skin", 0.0) != skin:
            ctx.md_skin(skin)                     # (with a skin the ranks' whole-frame lists live across steps)
        db = _lib.device_basis(self.bspline_config, ctx)
        batch = _lib.FrameBatch([atoms])
        e = np.empty(1)
        f = np.empty((batch.n_atoms, 3)) if forces else None
        v = np.empty((1, 6)) if virial else None
        addr = _lib._addr
        ctx.check(ctx.lib.uf3_eval_centres(db.handle, C.byref(batch.struct), addr(batch.pos), addr(batch.z), self._pc[0], self._pc[1],
                                           self._pc[2], int(atom_begin), int(atom_end), addr(e), addr(f), addr(v)))
        return float(e[0]), f, (v[0] if virial else None)

    def calculate(self, atoms=None, properties=None, system_changes=tuple(all_changes)):
        if properties is None:
            properties = self.implemented_properties
        _Base.calculate(self, atoms, properties, system_changes)
        want_f = 'forces' in properties
        if ('energy' in properties) or ('free_energy' in properties) or want_f:
            e, f, _ = self.evaluate_frames([atoms], forces=want_f)
            self.results['energy'] = float(e[0])
            self.results['free_energy'] = self.results['energy']
            if want_f:
                self.results['forces'] = f
        if 'stress' in properties:
            self.results['stress'] = self._get_stress(atoms)

    def _get_potential_energy(self, atoms=None, force_consistent=None):
        return float(self.evaluate_frames([atoms], forces=False)[0][0])

    def _get_forces(self, atoms=None):
        return self.evaluate_frames([atoms], forces=True)[1]

    def _get_stress(self, atoms=None, numerical=False, d=1e-6):
        """
        Stress in Voigt order (xx, yy, zz, yz, xz, xy), eV/A^3.  Default: analytic virial accumulated in
        the same neighbour traversal as the forces.  ``numerical=True`` reproduces the
        reference's route (central differences of the energy under strain, calculator.py:399-404), the 12
        strained copies going to the GPU as one batch.
        """
        cell = np.array(atoms.get_cell(), dtype=float).reshape(3, 3)
        vol = abs(np.linalg.det(cell))
        if not numerical:
            return self.evaluate_frames([atoms], forces=False, virial=True)[3][0] / vol
        pos = np.asarray(atoms.get_positions(), dtype=float)
        from uf3_amd.data.atoms import Atoms
        frames, pairs = [], [(0, 0), (1, 1), (2, 2), (1, 2), (0, 2), (0, 1)]
        for i, j in pairs:
            for sign in (+1, -1):
                eps = np.eye(3)
                if i == j:
                    eps[i, i] += sign * d
                else:
                    eps[i, j] += 0.5 * sign * d
                    eps[j, i] += 0.5 * sign * d
                frames.append(Atoms(numbers=atoms.get_atomic_numbers(), positions=pos @ eps, cell=cell @ eps,
                                    pbc=atoms.get_pbc()))
        e = self.evaluate_frames(frames, forces=False)[0]
        return np.array([(e[2 * k] - e[2 * k + 1]) / (2 * d * vol) for k in range(6)])

    def relax_fmax(self, geom, fmax=0.05, relax_cell=True, verbose=False, timeout=60.0, max_steps=2000, dt=0.1):
        """
        Minimise the maximum force (reference: calculator.py:406-436, which drives ASE's BFGSLineSearch on an
        ExpCellFilter).  ASE is not a dependency here: the same objective -- atomic forces, plus for fully
        periodic frames with ``relax_cell`` the strain derivative (analytic virial) -- is minimised with FIRE
        (Bitzek et al., PRL 97, 170201).  Returns a relaxed copy; warns and returns the last iterate on timeout.
        """
        import time
        import warnings
        from uf3_amd.data.atoms import Atoms
        numbers = np.asarray(geom.get_atomic_numbers())
        pos = np.array(geom.get_positions(), dtype=float)
        cell0 = np.array(geom.get_cell(), dtype=float).reshape(3, 3)
        pbc = np.asarray(geom.get_pbc(), dtype=bool)
        with_cell = bool(relax_cell and np.all(pbc))
        n = len(numbers)
        # generalised coordinates: scaled positions * cell0 (so they are lengths) and, optionally, the deformation
        # gradient D (cell = cell0 @ D), weighted by n like ASE's cell filters
        frac = pos @ np.linalg.inv(cell0) if with_cell else None
        D = np.eye(3)
        x_vel = np.zeros((n + (3 if with_cell else 0), 3))
        alpha0, f_inc, f_dec, f_alpha, n_min, dt_max = 0.1, 1.1, 0.5, 0.99, 5, 10 * dt
        alpha, n_pos, t0 = alpha0, 0, time.time()

        def evaluate(pos_, cell_):
            frame = Atoms(numbers=numbers, positions=pos_, cell=cell_, pbc=pbc)
            if with_cell:
                e, f, _, v = self.evaluate_frames([frame], forces=True, virial=True)
                vv = v[0]
                sigma_v = np.array([[vv[0], vv[5], vv[4]], [vv[5], vv[1], vv[3]], [vv[4], vv[3], vv[2]]])   # dE/d(eps)
                return float(e[0]), f, sigma_v
            e, f, _ = self.evaluate_frames([frame], forces=True)
            return float(e[0]), f, None

        cell = cell0.copy()
        for step in range(max_steps):
            e, f, dE_deps = evaluate(pos, cell)
            if with_cell:
                # positions move with the cell: q = frac @ cell0, x = q @ D; dE/dq = -f @ D^T; dE/dD = D^-T dE/deps
                g_cell = -(np.linalg.inv(D).T @ dE_deps) / max(n, 1)
                force = np.vstack([f @ D.T, g_cell])
                crit = max(np.abs(f).max(), np.abs(dE_deps).max() / max(n, 1))
            else:
                force, crit = f, np.abs(f).max()
            if verbose:
                print(f"relax_fmax step {step}: E = {e:.8f}  max|F| = {np.abs(f).max():.5f}")
            if crit < fmax:
                break
            if (time.time() - t0) > timeout:
                warnings.warn("Relaxation timed out.", RuntimeWarning)
                break
            power = float(np.sum(force * x_vel))
            if power > 0:
                fn, vn = np.linalg.norm(force), np.linalg.norm(x_vel)
                x_vel = (1 - alpha) * x_vel + (alpha * vn / fn) * force if fn > 0 else x_vel
                n_pos += 1
                if n_pos > n_min:
                    dt, alpha = min(dt * f_inc, dt_max), alpha * f_alpha
            else:
                x_vel[:] = 0.0
                n_pos, dt, alpha = 0, dt * f_dec, alpha0
            x_vel = x_vel + dt * force
            step_x = dt * x_vel
            big = np.abs(step_x).max()
            if big > 0.2:                                   # trust radius (Angstrom / unit strain)
                step_x *= 0.2 / big
            if with_cell:
                q = frac @ cell0 + step_x[:n]
                frac = q @ np.linalg.inv(cell0)
                D = D + step_x[n:] / max(n, 1)
                cell = cell0 @ D
                pos = frac @ cell
            else:
                pos = pos + step_x
        return Atoms(numbers=numbers, positions=pos, cell=cell, pbc=pbc)

