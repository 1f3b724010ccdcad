"""ctypes binding of libntg_amd.so (include/ntg_amd.h).  Host plumbing only: it marshals a Spec
into the C ABI and passes torch device pointers; all numerics run in the library's HIP kernels.
There is no fallback: if the library is missing or no GPU is visible, calls raise."""
from __future__ import annotations
import ctypes as C
import os
from typing import Optional
import numpy as np

from .spec import Spec

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NTG_AMD_LIB") or os.path.join(_HERE, "libntg_amd.so")   # NTG_AMD_LIB: a tuning build of the same library (tests/tools_variants.py)
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


class NtgError(RuntimeError):
    pass


class _AV(C.Structure):
    _fields_ = [("output", C.c_int), ("deriv", C.c_int)]


class _Spec(C.Structure):
    _fields_ = [("nout", C.c_int), ("nbps", C.c_int), ("bps", dp), ("kninterv", ip),
                ("knots", C.POINTER(dp)), ("order", ip), ("mult", ip), ("maxderiv", ip),
                ("family", C.c_int),
                ("nlic", C.c_int), ("nltc", C.c_int), ("nlfc", C.c_int),
                ("lic", dp), ("ltc", dp), ("lfc", dp),
                ("nnlic", C.c_int), ("nnltc", C.c_int), ("nnlfc", C.c_int),
                ("nicav", C.c_int), ("ntcav", C.c_int), ("nfcav", C.c_int),
                ("icav", C.POINTER(_AV)), ("tcav", C.POINTER(_AV)), ("fcav", C.POINTER(_AV)),
                ("nicf", C.c_int), ("nucf", C.c_int), ("nfcf", C.c_int),
                ("nicostav", C.c_int), ("ntcostav", C.c_int), ("nfcostav", C.c_int),
                ("icostav", C.POINTER(_AV)), ("tcostav", C.POINTER(_AV)), ("fcostav", C.POINTER(_AV)),
                ("lin_ineq", ip)]


class SolveOpts(C.Structure):
    _fields_ = [("itlim", C.c_int), ("opttol", C.c_double), ("steplimit", C.c_double),
                ("ls_mu", C.c_double), ("ls_eta", C.c_double), ("ls_maxfev", C.c_int),
                ("hessian", C.c_int), ("fixed_iters", C.c_int), ("block_threads", C.c_int), ("qn_memory", C.c_int), ("warm_start", C.c_int)]


_lib = None


def lib():
    """Load libntg_amd.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NtgError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        L.ntg_last_error.restype = C.c_char_p
        L.ntg_solve_kernel_name.restype = C.c_char_p
        L.ntg_batch_solve_kernel.restype = C.c_char_p
        L.ntg_batch_solve_kernel.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ntg_batch_workspace_bytes.restype = C.c_longlong
        L.ntg_batch_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ntg_plan_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.ntg_plan_destroy.argtypes = [C.c_void_p]
        L.ntg_plan_dims.argtypes = [C.c_void_p] + [ip] * 7
        L.ntg_plan_tables.argtypes = [C.c_void_p, dp, ip, dp]
        L.ntg_batch_bounds.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.ntg_batch_eval.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.ntg_batch_solve.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_longlong, C.c_void_p]
        L.ntg_basis_batch.argtypes = [C.c_int] * 6 + [C.c_void_p] * 5
        L.ntg_batch_mpc_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_longlong, C.c_void_p]
        L.ntg_batch_mpc_shift_multipliers.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
        L.ntg_batch_interp.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ntg_batch_interp_strided.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]
        L.ntg_plan_set_grids.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ntg_plan_clear_grids.argtypes = [C.c_void_p]
        L.ntg_plan_clear_grids.restype = None
        L.ntg_batch_kincar_reverse.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.ntg_batch_mpc_shift.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _check_tensor(t, dev, dtype=None):
    """the C ABI takes raw device pointers: a tensor of the wrong dtype / device / layout would corrupt memory silently"""
    import torch
    if t is None:
        return
    want = dtype if dtype is not None else torch.float64
    if not (t.is_cuda and t.device == dev and t.dtype == want and t.is_contiguous()):
        raise NtgError(f"tensor must be a contiguous {want} tensor on {dev}, got {t.dtype} on {t.device} (contiguous: {t.is_contiguous()})")


def _check(rc):
    if rc != 0:
        raise NtgError(f"libntg_amd error {rc}: {lib().ntg_last_error().decode()}")


def default_opts(**kw) -> SolveOpts:
    o = SolveOpts()
    lib().ntg_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Plan:
    """Device-resident, batch-shared part of a problem (ntg_plan)."""

    def __init__(self, spec: Spec, device: int = 0):
        self.spec = spec
        self.device = device
        self.grid_batch = 0   # batch of the per-problem grids in force (set_grids / clear_grids)
        k = self._keep = {}
        k["bps"] = np.ascontiguousarray(spec.bps, dtype=np.float64)
        for nm in ("kninterv", "order", "mult", "maxderiv"):
            k[nm] = np.asarray(getattr(spec, nm), dtype=np.int32)
        k["knots"] = [np.ascontiguousarray(x, dtype=np.float64) for x in spec.knots]
        k["kp"] = (dp * spec.nout)(*[x.ctypes.data_as(dp) for x in k["knots"]])
        for nm in ("lic", "ltc", "lfc"):
            k[nm] = np.ascontiguousarray(getattr(spec, nm), dtype=np.float64).reshape(-1)

        def avs(lst):
            arr = (_AV * max(len(lst), 1))()
            for j, (o, d) in enumerate(lst):
                arr[j].output, arr[j].deriv = o, d
            return arr
        for nm in ("icav", "tcav", "fcav", "icostav", "tcostav", "fcostav"):
            k[nm] = avs(list(getattr(spec, nm)))
        s = _Spec()
        s.nout, s.nbps = spec.nout, spec.nbps
        s.bps = k["bps"].ctypes.data_as(dp)
        s.kninterv = k["kninterv"].ctypes.data_as(ip); s.order = k["order"].ctypes.data_as(ip)
        s.mult = k["mult"].ctypes.data_as(ip); s.maxderiv = k["maxderiv"].ctypes.data_as(ip)
        s.knots = k["kp"]; s.family = spec.family
        s.nlic, s.nltc, s.nlfc = spec.nlic, spec.nltc, spec.nlfc
        s.lic, s.ltc, s.lfc = (k[nm].ctypes.data_as(dp) for nm in ("lic", "ltc", "lfc"))
        s.nnlic, s.nnltc, s.nnlfc = spec.nnlic, spec.nnltc, spec.nnlfc
        s.nicav, s.ntcav, s.nfcav = len(spec.icav), len(spec.tcav), len(spec.fcav)
        s.icav, s.tcav, s.fcav = k["icav"], k["tcav"], k["fcav"]
        s.nicf, s.nucf, s.nfcf = spec.nicf, spec.nucf, spec.nfcf
        s.nicostav, s.ntcostav, s.nfcostav = len(spec.icostav), len(spec.tcostav), len(spec.fcostav)
        s.icostav, s.tcostav, s.fcostav = k["icostav"], k["tcostav"], k["fcostav"]
        if len(spec.lin_ineq):
            assert len(spec.lin_ineq) == spec.nlic + spec.nltc + spec.nlfc
            k["lin_ineq"] = np.asarray(spec.lin_ineq, dtype=np.int32)
            s.lin_ineq = k["lin_ineq"].ctypes.data_as(ip)
        self._cspec = s
        h = C.c_void_p()
        _check(lib().ntg_plan_create(C.byref(s), device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().ntg_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup tables (host copies) ----
    def tables(self):
        sp = self.spec
        nblk = sum(sp.nbps * k * d for k, d in zip(sp.order, sp.maxderiv))
        blk = np.zeros(nblk); off = np.zeros((sp.nout, sp.nbps), dtype=np.int32)
        A = np.zeros((sp.nC, max(sp.nclin, 1)))
        _check(lib().ntg_plan_tables(self.h, blk.ctypes.data_as(dp), off.ctypes.data_as(ip), A.ctypes.data_as(dp)))
        return dict(blk=blk, off=off, A=(A.T.copy() if sp.nclin else np.zeros((0, sp.nC))))

    # ---- batched entry points; tensors are torch CUDA(HIP) float64/int32, contiguous ----
    def bounds(self, lower, upper):
        import torch
        sp = self.spec
        batch = lower.shape[0]
        ntot = sp.nC + sp.nclin + sp.ncnln
        bl = torch.empty((batch, ntot), dtype=torch.float64, device=lower.device); bu = torch.empty_like(bl)
        _check(lib().ntg_batch_bounds(self.h, batch, _ptr(lower), _ptr(upper), _ptr(bl), _ptr(bu), self._stream()))
        return bl, bu

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def eval(self, x, mode: int = 2, want_dense_jac: bool = False, out=None):
        """funobj + funcon for a batch.  `out` (a dict returned by an earlier call with the same shapes) reuses its buffers."""
        import torch
        sp = self.spec
        assert x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()
        batch = x.shape[0]
        dev = x.device
        if out is not None:
            f, g, c, jb = out["f"], out["g"], out.get("c"), out.get("jband")
            cj = out.get("_cj")   # the [batch][nC][ncnln] buffer behind an earlier dense Jacobian, reused
            if want_dense_jac and sp.ncnln and cj is None:
                cj = torch.empty((batch, sp.nC, sp.ncnln), dtype=torch.float64, device=dev)
            for t in (f, g, c, jb):
                _check_tensor(t, dev)
        else:
            f = torch.empty(batch, dtype=torch.float64, device=dev)
            g = torch.empty((batch, sp.nC), dtype=torch.float64, device=dev)
            c = jb = cj = None
            if sp.ncnln:
                c = torch.zeros((batch, sp.ncnln), dtype=torch.float64, device=dev)
                jb = torch.zeros((batch, sp.ncnln, sp.sumk), dtype=torch.float64, device=dev)
                if want_dense_jac:
                    cj = torch.empty((batch, sp.nC, sp.ncnln), dtype=torch.float64, device=dev)
        _check(lib().ntg_batch_eval(self.h, batch, _ptr(x), mode, _ptr(f), _ptr(g), _ptr(c), _ptr(jb), _ptr(cj), self._stream()))
        out = dict(f=f, g=g)
        if sp.ncnln:
            out.update(c=c, jband=jb)
            if want_dense_jac:
                out["cJac"] = cj.transpose(1, 2).contiguous()
                out["_cj"] = cj
        return out

    def interp(self, x, times):
        """Flat flag of every problem at the given times: x [batch, nC], times [ntimes] -> [batch, ntimes, nz].
        After set_grids: times [batch, ntimes], every problem at its own times on its own knots."""
        import torch
        assert x.is_cuda and x.dtype == torch.float64 and x.is_contiguous() and times.is_cuda and times.dtype == torch.float64
        batch = x.shape[0]
        if times.dim() == 1:
            stride = 0                       # one time vector for the batch (with per-problem grids: every problem on its own knots)
        elif times.dim() == 2 and self.grid_batch and times.shape[0] == batch:
            stride = times.shape[1]
        elif times.dim() == 2:
            raise NtgError("per-problem times [batch, ntimes] need per-problem grids of that batch (set_grids); pass a 1-D time vector")
        else:
            raise NtgError("times must be [ntimes] or, after set_grids, [batch, ntimes]")
        ntimes = times.shape[-1]
        z = torch.empty((batch, ntimes, self.spec.nz), dtype=torch.float64, device=x.device)
        _check(lib().ntg_batch_interp_strided(self.h, batch, _ptr(x), ntimes, _ptr(times.contiguous()), stride, _ptr(z), self._stream()))
        return z

    def set_grids(self, knots, bps, with_precond: bool = True):
        """Per-problem grids: knots [batch, ninterv+1], bps [batch, nbps] (device, float64).  See ntg_plan_set_grids."""
        _check_tensor(knots, knots.device); _check_tensor(bps, bps.device)
        if knots.shape[0] != bps.shape[0] or knots.shape[1] != self.spec.kninterv[0] + 1 or bps.shape[1] != self.spec.nbps:
            raise NtgError("knots must be [batch, ninterv+1] and bps [batch, nbps]")
        _check(lib().ntg_plan_set_grids(self.h, knots.shape[0], _ptr(knots), _ptr(bps), int(with_precond), self._stream()))
        self.grid_batch = int(knots.shape[0])

    def clear_grids(self):
        lib().ntg_plan_clear_grids(self.h)
        self.grid_batch = 0

    def kincar_reverse(self, z, wheelbase: float = 3.0, reverse_gear: bool = False):
        """Flat flag -> (x, y, theta, v, delta) per car: z [batch, ntimes, nz] (from interp) -> [batch, ntimes, ncars, 5]."""
        import torch
        _check_tensor(z, z.device)
        out = torch.empty((z.shape[0], z.shape[1], self.spec.nout // 2, 5), dtype=torch.float64, device=z.device)
        _check(lib().ntg_batch_kincar_reverse(self.h, z.shape[0], z.shape[1], _ptr(z), C.c_double(wheelbase), int(reverse_gear), _ptr(out), self._stream()))
        return out

    def mpc_shift(self, x, lower, upper, shift_bp: int, shift_knots: int):
        """Receding-horizon step in place: re-pin initial bounds to the solution's flag at breakpoint
        shift_bp, shift the coefficients by shift_knots knot intervals."""
        _check(lib().ntg_batch_mpc_shift(self.h, x.shape[0], shift_bp, shift_knots, _ptr(x), _ptr(lower), _ptr(upper), self._stream()))

    def mpc_shift_multipliers(self, batch: int, shift_bp: int, opts: SolveOpts, work):
        """The multipliers' share of the receding-horizon step: estimates in `work` move shift_bp breakpoints towards the start of the horizon."""
        _check(lib().ntg_batch_mpc_shift_multipliers(self.h, batch, shift_bp, C.byref(opts), _ptr(work), work.numel() * work.element_size(), self._stream()))

    def mpc_run(self, x, lower, upper, nsteps: int, shift_bp: int, shift_knots: int, opts: Optional[SolveOpts] = None, work=None):
        """nsteps x (solve, shift) inside the library (hipGraph replay).  Returns (inform of the last step, #not converged)."""
        import torch
        o = opts if opts is not None else default_opts()
        batch = x.shape[0]
        need = self.workspace_bytes(batch, o)
        if work is None:
            work = torch.empty(need, dtype=torch.uint8, device=x.device)
        inform = torch.empty(batch, dtype=torch.int32, device=x.device)
        bad = torch.zeros(1, dtype=torch.int32, device=x.device)
        torch.cuda.current_stream().synchronize()
        _check(lib().ntg_batch_mpc_run(self.h, batch, nsteps, shift_bp, shift_knots, _ptr(x), _ptr(lower), _ptr(upper), C.byref(o),
                                       _ptr(inform), _ptr(bad), _ptr(work), work.numel() * work.element_size(), self._stream()))
        return inform, bad

    def solve_kernel(self, batch: int, opts: Optional[SolveOpts] = None) -> str:
        """name of the kernel solve() launches for this batch and these options"""
        return lib().ntg_batch_solve_kernel(self.h, batch, C.byref(opts) if opts is not None else None).decode()

    def workspace_bytes(self, batch: int, opts: Optional[SolveOpts] = None) -> int:
        return int(lib().ntg_batch_workspace_bytes(self.h, batch, C.byref(opts) if opts is not None else None))

    def solve(self, lower, upper, x, opts: Optional[SolveOpts] = None, work=None, out=None, want_lambda: bool = False):
        """In place on x.  Returns dict(objective, inform, iters, nfev[, clambda])."""
        import torch
        sp = self.spec
        assert x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()
        assert lower.is_contiguous() and upper.is_contiguous()
        batch = x.shape[0]
        dev = x.device
        o = opts if opts is not None else default_opts()
        need = self.workspace_bytes(batch, o)
        if work is None:
            work = torch.empty(need, dtype=torch.uint8, device=dev)
        assert work.numel() * work.element_size() >= need
        _check_tensor(lower, dev); _check_tensor(upper, dev)
        if lower.shape != (batch, sp.nbounds) or upper.shape != (batch, sp.nbounds):
            raise NtgError(f"bounds must be [{batch}, {sp.nbounds}]")
        if out is not None:
            _check_tensor(out["objective"], dev)
            for k in ("inform", "iters", "nfev"):
                _check_tensor(out[k], dev, torch.int32)
            _check_tensor(out.get("clambda"), dev)
        if out is None:
            out = dict(objective=torch.empty(batch, dtype=torch.float64, device=dev),
                       inform=torch.empty(batch, dtype=torch.int32, device=dev),
                       iters=torch.empty(batch, dtype=torch.int32, device=dev),
                       nfev=torch.empty(batch, dtype=torch.int32, device=dev))
            if want_lambda:
                out["clambda"] = torch.empty((batch, sp.nC + sp.nclin + sp.ncnln), dtype=torch.float64, device=dev)
        _check(lib().ntg_batch_solve(self.h, batch, _ptr(lower), _ptr(upper), _ptr(x), C.byref(o),
                                     _ptr(out["objective"]), _ptr(out["inform"]), _ptr(out["iters"]), _ptr(out["nfev"]),
                                     _ptr(out.get("clambda")), _ptr(work), work.numel() * work.element_size(), self._stream()))
        return out


def basis_batch(knots, bps, order: int, mult: int, maxderiv: int):
    """bsplvd at every collocation point of many grids: knots [G, l+1], bps [G, P] (torch, device)."""
    import torch
    G, l1 = knots.shape
    P = bps.shape[1]
    blk = torch.empty((G, P, order, maxderiv), dtype=torch.float64, device=knots.device)
    off = torch.empty((G, P), dtype=torch.int32, device=knots.device)
    _check(lib().ntg_basis_batch(G, l1 - 1, order, mult, maxderiv, P, _ptr(knots), _ptr(bps), _ptr(blk), _ptr(off),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return blk, off
