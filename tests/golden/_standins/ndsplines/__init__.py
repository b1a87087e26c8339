"""NDSpline stand-in on top of scipy (golden capture only)."""
import numpy as np
from scipy import interpolate


class NDSpline:
    def __init__(self, knots, coefficients, degrees, periodic=False, extrapolate=True):
        self.knots = [np.asarray(k, dtype=float) for k in knots]
        self.coefficients = np.asarray(coefficients, dtype=float)
        self.ndim = len(self.knots)
        self.k = int(degrees) if np.ndim(degrees) == 0 else int(degrees[0])
        self.extrapolate = extrapolate
        if self.ndim == 1:
            self._s = interpolate.BSpline(self.knots[0], self.coefficients.reshape(-1),
                                          self.k, extrapolate=extrapolate)
        else:
            self._s = interpolate.NdBSpline(tuple(self.knots), self.coefficients,
                                            self.k, extrapolate=extrapolate)

    def __call__(self, x, nus=0):
        x = np.asarray(x, dtype=float)
        if self.ndim == 1:
            nu = int(np.ravel(nus)[0]) if np.ndim(nus) else int(nus)
            v = self._s(x.reshape(-1), nu=nu)
        else:
            nu = tuple(int(n) for n in np.ravel(nus)) if np.ndim(nus) else (int(nus),) * self.ndim
            v = self._s(x.reshape(-1, self.ndim), nu=nu)
        v = np.asarray(v, dtype=float)
        v[np.isnan(v)] = 0.0
        return v
