"""Problem description shared by the product bindings, the tests and bench.py.

Mirrors the argument list of the reference's ntg() (ntg.h:72-99): one Spec holds everything
that is common to a batch (grid, spline orders, linear constraint rows, active variables,
problem family); per-problem data (lowerb/upperb, initial guess) are arrays of shape [batch, .].
"""
from __future__ import annotations
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple
import numpy as np

AV = Tuple[int, int]  # (output, deriv)  -- reference av.h:22-26

FAM_KINCAR = 0
FAM_VANDERPOL = 1
FAM_TESTFAM = 2
FAM_OBSTACLE = 3
FAM_QUADROTOR = 4
FAM_MANIP = 5


def linspace_c(d0: float, d1: float, n: int) -> np.ndarray:
    """Bit-exact restatement of the reference's cumulative-add linspace (ntg.c:374-389)."""
    v = np.empty(n, dtype=np.float64)
    if d0 == d1:
        v[:] = d0
        return v
    h = (d1 - d0) / (n - 1)
    v[0] = d0
    for i in range(1, n):
        v[i] = v[i - 1] + h
    return v


@dataclass
class Spec:
    nout: int
    bps: np.ndarray                      # [nbps]
    kninterv: List[int]
    knots: List[np.ndarray]              # knots[o] has kninterv[o]+1 entries
    order: List[int]
    mult: List[int]
    maxderiv: List[int]
    family: int
    lic: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))   # [nlic, nz]
    ltc: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    lfc: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    nnlic: int = 0
    nnltc: int = 0
    nnlfc: int = 0
    icav: Sequence[AV] = ()
    tcav: Sequence[AV] = ()
    fcav: Sequence[AV] = ()
    nicf: int = 0
    nucf: int = 0
    nfcf: int = 0
    icostav: Sequence[AV] = ()
    tcostav: Sequence[AV] = ()
    fcostav: Sequence[AV] = ()
    lin_ineq: Sequence[int] = ()         # optional [nlic+nltc+nlfc]: 1 = linear row is an inequality row in every problem
    name: str = ""

    # ---- derived sizes (colloc.c:34-52, ntg.c:155-157) ----
    @property
    def nbps(self) -> int:
        return len(self.bps)

    @property
    def ncoef(self) -> List[int]:
        return [l * (k - m) + m for l, k, m in zip(self.kninterv, self.order, self.mult)]

    @property
    def nC(self) -> int:
        return sum(self.ncoef)

    @property
    def nz(self) -> int:
        return sum(self.maxderiv)

    @property
    def nlic(self) -> int:
        return self.lic.shape[0]

    @property
    def nltc(self) -> int:
        return self.ltc.shape[0]

    @property
    def nlfc(self) -> int:
        return self.lfc.shape[0]

    @property
    def nclin(self) -> int:
        return self.nlic + self.nltc * self.nbps + self.nlfc

    @property
    def ncnln(self) -> int:
        return self.nnlic + self.nnltc * self.nbps + self.nnlfc

    @property
    def nbounds(self) -> int:
        return self.nlic + self.nltc + self.nlfc + self.nnlic + self.nnltc + self.nnlfc

    @property
    def sumk(self) -> int:
        return sum(self.order)

    def eval_bytes(self) -> int:
        """Algorithmic bytes of one funobj+funcon evaluation."""
        nnzJ = (self.nnlic + self.nnlfc) * self.sumk + self.nnltc * self.nbps * self.sumk
        return 8 * (self.nC + 1 + self.nC + self.ncnln + nnzJ)
