"""
Species bookkeeping for the UF3 hot path: element ordering, the list of pair /
trio interactions (= feature-column blocks) and their integer hashes.

Mirrors the public surface of the reference's ``uf3/data/composition.py``
(``ChemicalSystem``: :28-164, ``sort_interaction_symbols``: :191-202,
``get_element_combinations``: :214-249, Szudzik hashes: :252-308) so that
``BSplineBasis`` / ``BasisFeaturizer`` users can switch imports only.
Own implementation; integer / string logic only.
"""
import itertools
import re
import numpy as np

chemical_symbols = (
    "X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As "
    "Se Br Kr Rb Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd "
    "Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am "
    "Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl Mc Lv Ts Og").split()
atomic_numbers = {s: z for z, s in enumerate(chemical_symbols)}
# the reference orders elements by a table that is literally the atomic number
reference_X = {s: z for s, z in atomic_numbers.items() if z > 0}


def symbols2numbers(symbols):
    """'H2O' | ['H', 'O'] | [1, 8] -> list of atomic numbers."""
    if isinstance(symbols, str):
        out = []
        for el, n in re.findall(r"([A-Z][a-z]?)(\d*)", symbols):
            out.extend([atomic_numbers[el]] * (int(n) if n else 1))
        return out
    return [int(s) if isinstance(s, (int, np.integer)) else atomic_numbers[s] for s in symbols]


def _symbol(el):
    return chemical_symbols[int(el)] if isinstance(el, (int, np.integer)) else str(el)


def sort_elements(symbols):
    return sorted(symbols, key=lambda el: atomic_numbers[el])


def sort_interaction_symbols(symbols, fix_first=True):
    """Z-sort an interaction tuple; for >=3 bodies the first (centre) stays put."""
    symbols = list(symbols)
    if len(symbols) >= 3 and fix_first:
        return tuple([symbols[0]] + sort_elements(symbols[1:]))
    return tuple(sort_elements(symbols))


def sort_interaction_map(imap):
    return {sort_interaction_symbols(k): v for k, v in imap.items()}


def szudzik_pair(pairs):
    xy = np.asarray(pairs)
    x, y = xy[..., 0], xy[..., 1]
    return np.where(x > y, x * x + y, y * y + x + y)


def get_szudzik_hash(array):
    """Left fold of Szudzik's pairing function over the columns of an (n, d) int array."""
    array = np.asarray(array)
    h = array[:, 0]
    for c in range(1, array.shape[1]):
        h = szudzik_pair(np.stack([h, array[:, c]], axis=-1))
    return h


def szudzik_unpair(hash_list):
    """Inverse of ``szudzik_pair`` (composition.py:272-290): with s = floor(sqrt(h)) and t = h - s^2,
    (x, y) = (s, t) where t < s and (t - s, s) otherwise.  Returns an (n, 2) float array like the reference."""
    h = np.asarray(hash_list)
    s = np.sqrt(h).astype(int)
    t = h - s * s
    first = t < s
    return np.stack([np.where(first, s, t - s), np.where(first, t, s)], axis=1).astype(float)


def unpack_szudzik_hash(hash_list, n_iter):
    """Undo the left fold of ``get_szudzik_hash``: n_iter columns per hash (composition.py:311-328)."""
    rest = np.asarray(hash_list)
    columns = []
    for _ in range(n_iter - 1):
        pair = szudzik_unpair(rest)
        columns.append(pair[:, 1])
        rest = pair[:, 0]
    columns.append(rest)
    return np.stack(columns[::-1], axis=1)


def symbols_to_hash(symbols):
    return get_szudzik_hash(np.array([[atomic_numbers[el] for el in symbols]]))[0]


def hash_to_symbols(hash_, n=2):
    return tuple(chemical_symbols[int(z)] for z in unpack_szudzik_hash([hash_], n)[0])


def hash_gather(values, hashes):
    """{hash: the values carrying it}, hashes ascending (composition.py:350-359)."""
    values, hashes = np.asarray(values), np.asarray(hashes)
    return {int(h): values[hashes == h] for h in np.unique(hashes)}


def get_element_combinations(element_list, n=3):
    """(centre, n1<=n2, ...) tuples, centre-major, each unique tuple once."""
    elements = sort_elements([_symbol(e) for e in element_list])
    seen, out = set(), []
    for combo in itertools.product(elements, repeat=n):
        key = sort_interaction_symbols(combo)
        if key not in seen:
            seen.add(key)
            out.append(key)
    return out


class ChemicalSystem:
    """Elements (ascending Z), degree, and the ordered interaction lists."""

    def __init__(self, element_list, degree=2):
        self.degree = int(degree)
        self.element_list = tuple(sort_elements({_symbol(e) for e in element_list}))
        self.numbers = [atomic_numbers[el] for el in self.element_list]
        self.interactions_map = self.get_interactions_map()
        self.interactions = self.get_interactions_list()
        self.interaction_hashes = self.get_interaction_hashes()

    @staticmethod
    def from_config(config):
        return ChemicalSystem.from_dict(config)

    @staticmethod
    def from_dict(config):
        return ChemicalSystem(config["element_list"], config["degree"])

    def as_dict(self):
        return dict(element_list=self.element_list, degree=self.degree)

    def __repr__(self):
        lines = ["ChemicalSystem:",
                 f"    Elements: {self.element_list}",
                 f"    Degree: {self.degree}",
                 f"    Pairs: {self.interactions_map[2]}"]
        if self.degree > 2:
            lines.append(f"    Trios: {self.interactions_map[3]}")
        return "\n".join(lines)

    def get_composition_tuple(self, geometry):
        numbers = np.asarray(geometry.get_atomic_numbers())
        return np.array([np.count_nonzero(numbers == z) for z in self.numbers], dtype=int)

    def get_interactions_map(self):
        zkey = lambda c: [atomic_numbers[x] for x in c]  # noqa: E731
        imap = {1: self.element_list}
        pairs = [sort_interaction_symbols(c) for c in
                 itertools.combinations_with_replacement(self.element_list, 2)]
        imap[2] = sorted(pairs, key=zkey)
        for d in range(3, self.degree + 1):
            imap[d] = sorted(get_element_combinations(self.element_list, d), key=zkey)
        return imap

    def get_interactions_list(self):
        out = list(self.element_list)
        for d in range(2, self.degree + 1):
            out.extend(self.interactions_map[d])
        return out

    def get_interaction_hashes(self):
        hashes = {}
        for d in range(2, self.degree + 1):
            numbers = np.array([symbols2numbers(t) for t in self.interactions_map[d]])
            numbers[:, 1:] = np.sort(numbers[:, 1:], axis=1)
            hashes[d] = get_szudzik_hash(numbers)
        return hashes


def interactions_to_numbers(interactions):
    if isinstance(interactions, tuple):
        return tuple(symbols2numbers(interactions))
    if isinstance(interactions, list):
        return [interactions_to_numbers(i) for i in interactions]
    if isinstance(interactions, dict):
        return {k: interactions_to_numbers(v) for k, v in interactions.items()}
    if isinstance(interactions, str):
        return atomic_numbers[interactions]
    raise ValueError(interactions)
