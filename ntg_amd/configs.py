"""Workload definitions (SURVEY.md §8 size table, §8d synthetic inputs).

  A   vanderpol as shipped      (examples/vanderpol.c:17-21,160-169)
  K0  kincar as shipped         (examples/kincar.c:133-137,319-339)
  B   kincar 2 outputs, order 6, mult 3, 20 intervals, P = 5l+1 = 101
  M   headline: 6 flat outputs (three cars stacked), same splines
  T   testfam: every callback slot populated (exercises cost.c / constraints.c row orders)
  O   kincar + circular obstacle (one nonlinear trajectory inequality)
  D   quadrotor: 4 outputs, order 8, mult 4, 40 intervals, maxderiv 5, P = 201, 2 nonlinear trajectory rows
  E   manipulator: 12 outputs, order 6, mult 3, 60 intervals, P = 301, 4 nonlinear trajectory inequalities
"""
from __future__ import annotations
import numpy as np
from .spec import Spec, linspace_c, FAM_KINCAR, FAM_VANDERPOL, FAM_TESTFAM, FAM_OBSTACLE, FAM_QUADROTOR, FAM_MANIP

SEED = 20261003
WHEELBASE = 3.0  # kincar.c:43


def kincar_flat_forward(x, u, b: float = WHEELBASE) -> np.ndarray:
    """State/input -> flat flag, the map of examples/kincar.c:46-65 (returns [2,3])."""
    z = np.zeros((2, 3))
    z[0, 0] = x[0]
    z[1, 0] = x[1]
    z[0, 1] = u[0] * np.cos(x[2])
    z[1, 1] = u[0] * np.sin(x[2])
    thdot = (u[0] / b) * np.tan(u[1])
    z[0, 2] = -u[0] * thdot * np.sin(x[2])
    z[1, 2] = u[0] * thdot * np.cos(x[2])
    return z


def _kincar_spec(ncars: int, order: int, mult: int, ninterv: int, nbps: int, T: float, name: str) -> Spec:
    nout = 2 * ncars
    nz = 3 * nout
    eye = np.eye(nz)
    return Spec(
        nout=nout, bps=linspace_c(0.0, T, nbps), kninterv=[ninterv] * nout,
        knots=[linspace_c(0.0, T, ninterv + 1) for _ in range(nout)],
        order=[order] * nout, mult=[mult] * nout, maxderiv=[3] * nout, family=FAM_KINCAR,
        lic=eye.copy(), lfc=eye.copy(), ltc=np.zeros((0, nz)),
        nucf=1, tcostav=[(o, 2) for o in range(nout)], name=name)


def config_K0() -> Spec:
    return _kincar_spec(1, 5, 3, 2, 20, 5.0, "K0:kincar-shipped")


def config_B() -> Spec:
    return _kincar_spec(1, 6, 3, 20, 101, 5.0, "B:kincar-2out-k6-l20")


def config_M() -> Spec:
    return _kincar_spec(3, 6, 3, 20, 101, 5.0, "M:kincar-6out-k6-l20")


def config_A() -> Spec:
    lic = np.zeros((2, 3)); lic[0, 0] = 1.0; lic[1, 1] = 1.0       # vanderpol.c:160-164
    lfc = np.zeros((1, 3)); lfc[0, 0] = -1.0; lfc[0, 1] = 1.0      # vanderpol.c:167-169
    return Spec(nout=1, bps=linspace_c(0.0, 5.0, 20), kninterv=[2], knots=[linspace_c(0.0, 5.0, 3)],
                order=[5], mult=[3], maxderiv=[3], family=FAM_VANDERPOL, lic=lic, lfc=lfc,
                ltc=np.zeros((0, 3)), nucf=1, tcostav=[(0, 0), (0, 1), (0, 2)], name="A:vanderpol-shipped")


def bounds_A():
    b = np.array([1.0, 0.0, 1.0])
    return b.copy(), b.copy()


def bounds_K0_shipped():
    """Lane change of kincar.c:319-320."""
    zi = kincar_flat_forward([0.0, -2.0, 0.0], [8.0, 0.0])
    zf = kincar_flat_forward([40.0, 2.0, 0.0], [8.0, 0.0])
    b = np.concatenate([zi.ravel(), zf.ravel()])
    return b.copy(), b.copy()


def config_T(nout: int = 3, order: int = 5, mult: int = 3, ninterv: int = 4, nbps: int = 17) -> Spec:
    """testfam: linear rows of all three kinds + all six callback slots."""
    nz = 3 * nout
    rng = np.random.default_rng(7)
    lic = np.round(rng.uniform(-1, 1, (2, nz)), 3)
    ltc = np.round(rng.uniform(-1, 1, (1, nz)), 3)
    lfc = np.round(rng.uniform(-1, 1, (2, nz)), 3)
    L = nout - 1
    # different spline spec on the last output to exercise the per-output tables
    orders = [order] * nout; mults = [mult] * nout; nints = [ninterv] * nout
    orders[L] = order + 1; nints[L] = ninterv + 1
    return Spec(
        nout=nout, bps=linspace_c(0.0, 2.0, nbps), kninterv=nints,
        knots=[linspace_c(0.0, 2.0, l + 1) for l in nints], order=orders, mult=mults,
        maxderiv=[3] * nout, family=FAM_TESTFAM, lic=lic, ltc=ltc, lfc=lfc,
        nnlic=1, nnltc=2, nnlfc=1,
        icav=[(0, 0), (L, 1)], tcav=[(0, 0), (0, 1), (L, 0), (L, 2)], fcav=[(0, 0), (0, 2), (L, 1)],
        nicf=1, nucf=1, nfcf=1,
        icostav=[(o, d) for o in range(nout) for d in (0, 1)],
        tcostav=[(o, d) for o in range(nout) for d in (0, 1, 2)],
        fcostav=[(o, d) for o in range(nout) for d in (0, 1, 2)], name="T:testfam")


def kincar_random_bounds(ncars: int, batch: int, seed: int = SEED):
    """Per-problem equality bounds for the kincar family: draws in the order
    x0,y0,th0,v0,d0,xf,yf,thf,vf,df per car, PCG64 stream `seed`.  Returns lower, upper
    of shape [batch, 12*ncars] laid out [lic rows (output-major, deriv-minor); lfc rows]."""
    rng = np.random.default_rng(seed)
    nout = 2 * ncars
    b = np.empty((batch, 6 * nout))
    for p in range(batch):
        for c in range(ncars):
            x0 = rng.uniform(-5, 5); y0 = rng.uniform(-3, 3); th0 = rng.uniform(-0.3, 0.3)
            v0 = rng.uniform(4, 12); d0 = rng.uniform(-0.1, 0.1)
            xf = x0 + rng.uniform(30, 50); yf = rng.uniform(-3, 3); thf = rng.uniform(-0.3, 0.3)
            vf = rng.uniform(4, 12); df = rng.uniform(-0.1, 0.1)
            zi = kincar_flat_forward([x0, y0, th0], [v0, d0])
            zf = kincar_flat_forward([xf, yf, thf], [vf, df])
            b[p, 6 * c:6 * c + 6] = zi.ravel()
            b[p, 3 * nout + 6 * c:3 * nout + 6 * c + 6] = zf.ravel()
    return b.copy(), b.copy()


INF_BOUND = 1e20  # NPSOL's "infinite bound" 


def config_O(ninterv: int = 20, order: int = 6) -> Spec:
    """kincar (2 outputs) + circular-obstacle trajectory constraint (family 3): nonlinear inequality
    (x-20)^2 + (y-0.5)^2 >= r^2 at every breakpoint, r^2 given through the bounds."""
    s = _kincar_spec(1, order, 3, ninterv, 5 * ninterv + 1, 5.0, f"O:kincar-obstacle-k{order}-l{ninterv}")
    s.family = FAM_OBSTACLE
    s.nnltc = 1
    s.tcav = [(0, 0), (1, 0)]
    return s


def obstacle_bounds(batch: int, radius: float = 3.0, seed: int = SEED):
    """kincar bounds of kincar_random_bounds + [r^2, +inf) for the obstacle row."""
    lo, up = kincar_random_bounds(1, batch, seed)
    lo = np.concatenate([lo, np.full((batch, 1), radius * radius)], axis=1)
    up = np.concatenate([up, np.full((batch, 1), INF_BOUND)], axis=1)
    return lo, up


QUAD_G = 9.81


def config_D(ninterv: int = 40, order: int = 8, mult: int = 4, T: float = 5.0) -> Spec:
    """Quadrotor flat outputs (x, y, z, yaw), family 4: snap^2 + yaw-acceleration^2 running cost, the whole flag
    pinned at both ends (rest to rest), thrust^2 and speed^2 bounded at every breakpoint."""
    nout, d = 4, 5
    nz = nout * d
    nbps = 5 * ninterv + 1
    eye = np.eye(nz)
    return Spec(
        nout=nout, bps=linspace_c(0.0, T, nbps), kninterv=[ninterv] * nout,
        knots=[linspace_c(0.0, T, ninterv + 1) for _ in range(nout)],
        order=[order] * nout, mult=[mult] * nout, maxderiv=[d] * nout, family=FAM_QUADROTOR,
        lic=eye.copy(), lfc=eye.copy(), ltc=np.zeros((0, nz)),
        nnltc=2, tcav=[(0, 1), (1, 1), (2, 1), (0, 2), (1, 2), (2, 2)],
        nucf=1, tcostav=[(0, 4), (1, 4), (2, 4), (3, 2)], name=f"D:quadrotor-4out-k{order}-l{ninterv}")


def quadrotor_bounds(batch: int, vmax: float = 6.0, seed: int = SEED):
    """Rest-to-rest flights: start p0, yaw0 -> p0 + U(8,22) m along a random direction, yaw0 + U(-1,1).
    Rows: [lic (20); lfc (20); thrust^2 in [(0.5 g)^2, (1.8 g)^2]; speed^2 in (-inf, vmax^2]]."""
    rng = np.random.default_rng(seed)
    lo = np.zeros((batch, 42)); up = np.zeros((batch, 42))
    for p in range(batch):
        p0 = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(2, 6)])
        yaw0 = rng.uniform(-0.5, 0.5)
        dirv = rng.normal(size=3); dirv[2] *= 0.3; dirv /= np.linalg.norm(dirv)
        pf = p0 + dirv * rng.uniform(8, 22)
        yawf = yaw0 + rng.uniform(-1, 1)
        zi = np.zeros((4, 5)); zf = np.zeros((4, 5))
        zi[:3, 0] = p0; zi[3, 0] = yaw0; zf[:3, 0] = pf; zf[3, 0] = yawf
        lo[p, :20] = zi.ravel(); lo[p, 20:40] = zf.ravel()
        up[p, :40] = lo[p, :40]
        lo[p, 40] = (0.5 * QUAD_G) ** 2; up[p, 40] = (1.8 * QUAD_G) ** 2
        lo[p, 41] = -INF_BOUND; up[p, 41] = vmax * vmax
    return lo, up


def config_E(ninterv: int = 60, order: int = 6, mult: int = 3, narms: int = 4, T: float = 5.0) -> Spec:
    """narms planar 3-link arms (family 5): joint-acceleration^2 running cost, the whole flag pinned at both ends,
    one tip-height ceiling (nonlinear inequality) per arm at every breakpoint."""
    nout = 3 * narms
    nz = 3 * nout
    nbps = 5 * ninterv + 1
    eye = np.eye(nz)
    return Spec(
        nout=nout, bps=linspace_c(0.0, T, nbps), kninterv=[ninterv] * nout,
        knots=[linspace_c(0.0, T, ninterv + 1) for _ in range(nout)],
        order=[order] * nout, mult=[mult] * nout, maxderiv=[3] * nout, family=FAM_MANIP,
        lic=eye.copy(), lfc=eye.copy(), ltc=np.zeros((0, nz)),
        nnltc=narms, tcav=[(o, 0) for o in range(nout)],
        nucf=1, tcostav=[(o, 2) for o in range(nout)], name=f"E:manipulator-{nout}out-k{order}-l{ninterv}")


def manipulator_bounds(batch: int, narms: int = 4, ceiling: float = 2.5, seed: int = SEED):
    """Rest-to-rest swings of every arm from low on the right to low on the left (shoulder 0.1-0.5 -> 2.6-3.0 rad,
    elbow and wrist bent 0.2-0.6 rad): the straight joint-space path lifts the tip to ~2.9, above the ceiling.
    Rows: [lic (9 narms); lfc (9 narms); tip height of arm j in (-inf, ceiling]]."""
    rng = np.random.default_rng(seed)
    nout = 3 * narms
    nb = 6 * nout + narms
    lo = np.zeros((batch, nb)); up = np.zeros((batch, nb))
    for p in range(batch):
        zi = np.zeros((nout, 3)); zf = np.zeros((nout, 3))
        for j in range(narms):
            zi[3 * j, 0] = rng.uniform(0.1, 0.5); zf[3 * j, 0] = rng.uniform(2.6, 3.0)
            zi[3 * j + 1, 0] = rng.uniform(0.2, 0.6); zf[3 * j + 1, 0] = rng.uniform(0.2, 0.6)
            zi[3 * j + 2, 0] = rng.uniform(0.2, 0.6); zf[3 * j + 2, 0] = rng.uniform(0.2, 0.6)
        lo[p, :3 * nout] = zi.ravel(); lo[p, 3 * nout:6 * nout] = zf.ravel()
        up[p, :6 * nout] = lo[p, :6 * nout]
        lo[p, 6 * nout:] = -INF_BOUND; up[p, 6 * nout:] = ceiling
    return lo, up
