"""Tensor-level wrappers over the C ABI (include/omlm.h).  No autograd, no fallbacks.

dtype code convention (include/omlm.h): 0 = fp32 ("bf16x3" split for GEMM/attention operands), 1 = bf16, 2 = fp16.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional, Sequence

import torch

from . import hip
from .hip import call, ptr, stream_ptr

F32, BF16, F16 = 0, 1, 2
_CODES = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
H16 = (torch.bfloat16, torch.float16)            # the 16-bit GEMM / attention operand types (precision "bf16" / "fp16")


def dcode(dtype: torch.dtype) -> int:
    try:
        return _CODES[dtype]
    except KeyError:
        raise TypeError(f"unsupported operand dtype {dtype}") from None


def tdtype(code: int) -> torch.dtype:
    return {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[code]


def _chk(t: torch.Tensor, name: str):
    hip.require_gpu(t, name)
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


# ---- bf16x3 through the bf16 tile kernels: fp32 operands are split once into bf16 hi / lo planes (omlm_split_planes) and the
# three products run as ONE GEMM with a 3x longer k-loop (omlm_gemm_planes).  The register-staged fp32 kernel measured 169
# TFLOP/s fp32-equivalent on the trunk shapes; the plane form runs at a third of the bf16 kernels' ~1 PFLOP/s.
# Planes are cached ONLY inside a plane scope (planes_begin .. planes_end: one forward + its backward, opened by the engine), so an
# activation feeding its forward, input-gradient and weight-gradient GEMMs is split once -- and nothing survives the scope: the
# kernels (and the fused optimizer) write tensors through raw pointers without bumping torch's version counter, so a cache that
# outlives the step would hand out planes of last step's weights (a captured HIP graph would additionally have lost the split
# launches: they ran during the eager warm-up).  Every scope starts empty, i.e. the split kernels of a captured micro-step are part
# of its graph and read the current weights on every replay.  Outside a scope (direct ops.gemm calls) nothing is cached.
# OMLM_X3_PLANES=0 keeps every fp32 GEMM on the register-staged kernel.
_X3_PLANES = os.environ.get("OMLM_X3_PLANES", "1") == "1"
_X3_MIN_MACS = 1 << 26
_PLANES = {}
_PLANE_SCOPE = [0]                 # > 0 while the engine sequences a forward / backward


def planes_begin():
    """Open a plane scope (engine.run_forward): the cache starts empty."""
    _PLANES.clear()
    _PLANE_SCOPE[0] = 1


def planes_end():
    """Close the scope (end of the backward, or of a forward that saves nothing): every cached plane is dropped."""
    _PLANES.clear()
    _PLANE_SCOPE[0] = 0


def operand_planes(t: torch.Tensor, rows: int, ld: int):
    """(planes buffer, byte distance hi -> lo) for the fp32 region rows x ld at t's data pointer."""
    n = int(rows) * int(ld)
    key = id(t)
    scoped = _PLANE_SCOPE[0] > 0
    if scoped:
        ent = _PLANES.get(key)
        if ent is not None and ent[0]() is t and ent[1] == t._version and ent[2] == (t.data_ptr(), n):
            return ent[3], ent[4]
    pe = (n + 7) // 8 * 8
    buf = torch.empty(2 * pe, dtype=torch.bfloat16, device=t.device)
    call("omlm_split_planes", ptr(t), ptr(buf), n, pe, stream_ptr())
    if scoped:
        try:
            ref = weakref.ref(t, lambda _r, k=key: _PLANES.pop(k, None))
            _PLANES[key] = (ref, t._version, (t.data_ptr(), n), buf, pe * 2)
        except TypeError:
            pass
    return buf, pe * 2


_TAIL_WS = {}                  # (device index, stream) -> workspace of the GEMM tails' deterministic split-K: a per-call argument of the library
_TAIL_WS_BYTES = int(os.environ.get("OMLM_GEMM_TAIL_WS_MB", "64")) << 20


def tail_workspace(device: torch.device):
    """(pointer, bytes) of the CURRENT stream's split-K scratch on `device` (include/omlm.h: the buffer belongs to the call, one per stream that
    launches GEMMs concurrently).  A stream that is capturing gets its own buffer from the graph's pool: it lives as long as the graph that
    replays its launches."""
    if _TAIL_WS_BYTES <= 0:
        return None, 0
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(device).cuda_stream)
    buf = _TAIL_WS.get(key)
    if buf is None:
        buf = _TAIL_WS[key] = torch.empty(_TAIL_WS_BYTES // 4, device=device)
    return ptr(buf), _TAIL_WS_BYTES


def gemm(A: torch.Tensor, B: torch.Tensor, C_: torch.Tensor, *, M: int, N: int, K: int,
         lda: Optional[int] = None, ldb: Optional[int] = None, ldc: Optional[int] = None,
         a_kmajor: bool = False, b_kmajor: bool = False, Cin: Optional[torch.Tensor] = None,
         ldcin: Optional[int] = None, a_map=None, b_map=None, c_map=None, alpha: float = 1.0,
         a_rows: Optional[int] = None, b_rows: Optional[int] = None, planes: bool = True):
    """C[m,n] = alpha * sum_k A(m,k) B(n,k) (+ Cin).  A, B share dtype (bf16 or fp32); C is fp32 or bf16.
    planes=False keeps fp32 operands on the register-staged kernel (no hi/lo plane buffers, no plane cache)."""
    hip.require_gpu(A, "A")
    assert A.dtype == B.dtype, (A.dtype, B.dtype)
    lda = lda if lda is not None else A.shape[-1]
    ldb = ldb if ldb is not None else B.shape[-1]
    ldc = ldc if ldc is not None else C_.shape[-1]
    if Cin is not None:
        assert Cin.dtype == torch.float32
        ldcin = ldcin if ldcin is not None else Cin.shape[-1]
    a_rows = a_rows if a_rows is not None else A.numel() // lda
    b_rows = b_rows if b_rows is not None else B.numel() // ldb
    ws, wsb = tail_workspace(A.device) if (A.dtype in H16 or A.dtype == torch.float32) else (None, 0)
    if (A.dtype == torch.float32 and planes and _X3_PLANES and M * N * K >= _X3_MIN_MACS and not (a_kmajor and a_map is not None)
            and not (b_kmajor and b_map is not None) and (a_kmajor and b_kmajor or K % 8 == 0)):
        pa, sa = operand_planes(A, a_rows, lda)
        pb, sb = operand_planes(B, b_rows, ldb)
        call("omlm_gemm_planes", ptr(pa), sa, ptr(pb), sb, ptr(C_), ptr(Cin), ptr(a_map), ptr(b_map), ptr(c_map),
             a_rows, b_rows, M, N, K, lda, ldb, ldc, ldcin or 0, int(a_kmajor), int(b_kmajor), dcode(C_.dtype), float(alpha),
             ws, wsb, stream_ptr())
        return
    call("omlm_gemm", ptr(A), ptr(B), ptr(C_), ptr(Cin), ptr(a_map), ptr(b_map), ptr(c_map),
         a_rows, b_rows, M, N, K, lda, ldb, ldc, ldcin or 0, int(a_kmajor), int(b_kmajor),
         dcode(A.dtype), dcode(C_.dtype), float(alpha), ws, wsb, stream_ptr())


def gemm_planes16(A, A_lo, B, B_lo, C_, C_lo=None, *, M: int, N: int, K: int, Cin: Optional[torch.Tensor] = None, a_map=None, c_map=None,
                  ldc: Optional[int] = None, a_rows: Optional[int] = None, b_rows: Optional[int] = None):
    """C = A B^T (+ Cin) with both operands as hi/lo planes of ONE 16-bit type (A [M, K], B [N, K] row-major; omlm_gemm_planes16): three
    products, fp32 accumulation.  C_lo None: C is fp32.  C_lo given: C / C_lo receive the result as planes of the operand type.
    a_map / c_map: row maps as in gemm (physical row of logical row m in A's planes / in C and Cin)."""
    hip.require_gpu(A, "A")
    assert A.dtype in H16 and A_lo.dtype == A.dtype and B.dtype == A.dtype and B_lo.dtype == A.dtype
    assert A_lo.shape == A.shape and B_lo.shape == B.shape
    if C_lo is None:
        assert C_.dtype == torch.float32 and (Cin is None or Cin.dtype == torch.float32)
    else:
        assert C_.dtype == A.dtype and C_lo.dtype == A.dtype and C_lo.shape == C_.shape and Cin is None
    lda, ldb = A.shape[-1], B.shape[-1]
    ldc = ldc if ldc is not None else C_.shape[-1]
    ws, wsb = tail_workspace(A.device)
    call("omlm_gemm_planes16", ptr(A), ptr(A_lo), ptr(B), ptr(B_lo), ptr(C_), ptr(C_lo), ptr(Cin), ptr(a_map), ptr(c_map),
         a_rows if a_rows is not None else A.numel() // lda, b_rows if b_rows is not None else B.numel() // ldb,
         M, N, K, lda, ldb, ldc, Cin.shape[-1] if Cin is not None else 0, dcode(A.dtype), ws, wsb, stream_ptr())


class Fp8Planes:
    """The fp8 companions of a half operand [rows, ld] for omlm_gemm_mx16: planes [2, rows padded to 256, 2 ld] bytes (hi8, lo8; element k of a row
    at byte k, zero up to the next multiple of 128) and one E8M0 scale byte per row.  Written by layernorm_fwd_mx / ffmid_fwd_mx (which also
    write their rows' zero tails: zero=False is enough for them) / QuantRowsGroup (weights: persistent, zero-filled once)."""
    __slots__ = ("planes", "scale", "rows", "ld")

    def __init__(self, rows: int, ld: int, device, zero: bool = True):
        self.rows, self.ld = int(rows), int(ld)
        rp = (self.rows + 255) // 256 * 256
        # bytes [K, ceil128(K)) of every real row must be zeros (an fp8 NaN byte there would poison the row's outputs); pad rows and their
        # scales only ever reach output rows >= M, which no epilogue stores
        self.planes = (torch.zeros if zero else torch.empty)(2, rp, 2 * self.ld, dtype=torch.uint8, device=device)
        self.scale = (torch.full((rp,), 127, dtype=torch.uint8, device=device) if zero else torch.empty(rp, dtype=torch.uint8, device=device))

    @property
    def stride(self) -> int:
        return self.planes.shape[1] * self.planes.shape[2]


def gemm_mx16(A, A8: Fp8Planes, B, B8: Fp8Planes, C_, C_lo=None, *, M: int, N: int, K: int, Cin: Optional[torch.Tensor] = None):
    """C = A B^T (+ Cin): the half product of the hi planes plus the two correction products on fp8 at twice the matrix rate (omlm_gemm_mx16).
    C_lo: plane output -- a half tensor, or a uint8 tensor of C's shape for the lo plane as bf8 (e5m2) bytes."""
    hip.require_gpu(A, "A")
    assert A.dtype == torch.float16 and B.dtype == torch.float16
    lda, ldb = A.shape[-1], B.shape[-1]
    assert A8.ld == lda and B8.ld == ldb and A8.rows >= M and B8.rows >= N
    if C_lo is None:
        assert C_.dtype == torch.float32 and (Cin is None or Cin.dtype == torch.float32)
    else:
        assert C_.dtype == A.dtype and C_lo.dtype in (A.dtype, torch.uint8) and C_lo.shape == C_.shape and Cin is None
    ws, wsb = tail_workspace(A.device)
    call("omlm_gemm_mx16", ptr(A), ptr(A8.planes), A8.stride, ptr(A8.scale), ptr(B), ptr(B8.planes), B8.stride, ptr(B8.scale),
         ptr(C_), ptr(C_lo), int(C_lo is not None and C_lo.dtype == torch.uint8), ptr(Cin), A.numel() // lda, B.numel() // ldb, M, N, K, lda, ldb, C_.shape[-1],
         Cin.shape[-1] if Cin is not None else 0, ws, wsb, stream_ptr())


class _WgradDesc(C.Structure):
    """omlm_gemm_wgrad_desc (include/omlm.h)"""
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("c_map", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int)]


class WgradGroup:
    """Collects the weight-gradient contractions dW[M,N] += dY[K,M]^T X[K,N] of a backward pass (16-bit operands of ONE type, rows =
    tokens) and issues them as ONE grouped launch (omlm_gemm_wgrad_group).  The operands are kept alive until :meth:`flush`."""

    def __init__(self):
        self.items = []
        self.dtype = None

    def add(self, dY: torch.Tensor, X: torch.Tensor, dW: torch.Tensor, *, M: int, N: int, K: int, c_map=None):
        hip.require_gpu(dY, "dY")
        assert dY.dtype in H16 and X.dtype == dY.dtype and dW.dtype == torch.float32
        assert self.dtype in (None, dY.dtype), "one operand type per group"
        self.dtype = dY.dtype
        self.items.append((dY, X, dW, c_map, M, N, K, dY.shape[-1], X.shape[-1], dW.shape[-1]))

    def flush(self, splits: int = 0):
        n = len(self.items)
        if n == 0:
            return
        arr = (_WgradDesc * n)()
        for d, (dY, X, dW, c_map, M, N, K, lda, ldb, ldc) in zip(arr, self.items):
            d.A, d.B, d.C, d.c_map = ptr(dY), ptr(X), ptr(dW), ptr(c_map)
            d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, lda, ldb, ldc
        call("omlm_gemm_wgrad_group", C.cast(arr, C.c_void_p), n, int(splits), dcode(self.dtype), stream_ptr())
        self.items = []


class _CastDesc(C.Structure):
    """omlm_cast_pad_desc (include/omlm.h)"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("R", C.c_int), ("C", C.c_int), ("ld_src", C.c_int), ("ld_dst", C.c_int),
                ("transpose", C.c_int), ("lo", C.c_int)]


class CastPadGroup:
    """Collects weight re-packs (cast_pad / transpose_cast problems with ONE output type) and issues them as one launch."""

    def __init__(self):
        self.items, self.dtype = [], None

    def add(self, src, dst, R, C_, ld_src, ld_dst, transpose=False, lo=False):
        """lo: dst receives the lo plane rne16(v - rne16(v)) of the cast (the hi/lo weight planes of precision "fp16ff")."""
        hip.require_gpu(src, "src")
        assert src.dtype == torch.float32 and self.dtype in (None, dst.dtype), "fp32 sources, one output type per group"
        self.dtype = dst.dtype
        self.items.append((src, dst, int(R), int(C_), int(ld_src), int(ld_dst), int(bool(transpose)), int(bool(lo))))

    def flush(self):
        n = len(self.items)
        if n == 0:
            return
        if os.environ.get("OMLM_PACK_GROUP", "1") == "0":            # A/B lever: one launch per re-pack
            assert not any(it[7] for it in self.items), "lo planes exist in the grouped launch only"
            for src, dst, R, C_, ld_src, ld_dst, tr, _ in self.items:
                (transpose_cast if tr else cast_pad)(src, dst, R, C_, ld_src, ld_dst)
            self.items = []
            return
        arr = (_CastDesc * n)()
        for d, (src, dst, R, C_, ld_src, ld_dst, tr, lo) in zip(arr, self.items):
            d.src, d.dst, d.R, d.C, d.ld_src, d.ld_dst, d.transpose, d.lo = ptr(src), ptr(dst), R, C_, ld_src, ld_dst, tr, lo
        call("omlm_cast_pad_group", C.cast(arr, C.c_void_p), n, dcode(self.dtype), stream_ptr())
        self.items = []


def layernorm_fwd(x, gamma, y, xcast, mean, rstd, eps=1e-5):
    M, D = x.shape
    call("omlm_layernorm_fwd", ptr(x), ptr(gamma), ptr(y), ptr(xcast), ptr(mean), ptr(rstd),
         M, D, y.shape[-1], float(eps), dcode(y.dtype), stream_ptr())


def layernorm_fwd_planes(x, gamma, y, y_lo, mean, rstd, eps=1e-5):
    """layernorm_fwd with the result as hi/lo planes of y's 16-bit type: y = rne16(v), y_lo = rne16(v - y) (omlm_layernorm_fwd_planes)."""
    M, D = x.shape
    assert y.dtype in H16 and y_lo.dtype == y.dtype and y_lo.shape == y.shape
    call("omlm_layernorm_fwd_planes", ptr(x), ptr(gamma), ptr(y), ptr(y_lo), ptr(mean), ptr(rstd),
         M, D, y.shape[-1], float(eps), dcode(y.dtype), stream_ptr())


def layernorm_fwd_mx(x, gamma, y, P: "Fp8Planes", mean, rstd, eps=1e-5):
    """layernorm_fwd with y as omlm_gemm_mx16's A operand: the half hi plane y (bit for bit layernorm_fwd's) + its fp8 planes and row scales in P."""
    M, D = x.shape
    assert y.dtype == torch.float16 and P.ld == y.shape[-1] and P.rows >= M
    call("omlm_layernorm_fwd_mx", ptr(x), ptr(gamma), ptr(y), ptr(P.planes), P.stride, ptr(P.scale), ptr(mean), ptr(rstd),
         M, D, y.shape[-1], float(eps), stream_ptr())


class _QuantDesc(C.Structure):
    """omlm_quant_rows_desc (include/omlm.h)"""
    _fields_ = [("src", C.c_void_p), ("dst8", C.c_void_p), ("lo_stride", C.c_longlong), ("scale8", C.c_void_p),
                ("R", C.c_int), ("C", C.c_int), ("ld_src", C.c_int), ("ld8", C.c_int)]


class QuantRowsGroup:
    """Collects the fp8 re-packs of fp32 weights (rows [row0, row0 + R) of an Fp8Planes from src [R, C]) and issues them as one launch."""

    def __init__(self):
        self.items = []

    def add(self, src, P: "Fp8Planes", row0: int, R: int, C_: int, ld_src: int):
        hip.require_gpu(src, "src")
        assert src.dtype == torch.float32 and row0 + R <= P.rows and C_ <= 2 * P.ld
        self.items.append((src, P, int(row0), int(R), int(C_), int(ld_src)))

    def flush(self):
        n = len(self.items)
        if n == 0:
            return
        arr = (_QuantDesc * n)()
        for d, (src, P, row0, R, C_, ld_src) in zip(arr, self.items):
            pitch = 2 * P.ld
            d.src, d.dst8, d.lo_stride, d.scale8 = ptr(src), P.planes.data_ptr() + row0 * pitch, P.stride, P.scale.data_ptr() + row0
            d.R, d.C, d.ld_src, d.ld8 = R, C_, ld_src, pitch
        call("omlm_quant_rows_mx", C.cast(arr, C.c_void_p), n, stream_ptr())
        self.items = []


_LN_WS = {}


def _ln_workspace(D: int, device) -> torch.Tensor:
    key = (D, str(device))
    if key not in _LN_WS:
        _LN_WS[key] = torch.empty(int(hip.lib().omlm_layernorm_bwd_workspace_bytes(D)) // 4, device=device)
    return _LN_WS[key]


class _ColsumDesc(C.Structure):
    """omlm_colsum_desc (include/omlm.h)"""
    _fields_ = [("part", C.c_void_p), ("out", C.c_void_p), ("P", C.c_int), ("C", C.c_int), ("ldp", C.c_int)]


class ColsumGroup:
    """Collects column sums out[c] += sum_p part[p, c] (the d(gamma) partial rows of the LayerNorm backwards of one backward pass) and issues
    them as ONE launch (omlm_colsum_group).  The partial buffers are kept alive until :meth:`flush`."""

    def __init__(self):
        self.items = []

    def add(self, part: torch.Tensor, out: torch.Tensor, P: int, C_: int, ldp: int):
        self.items.append((part, out, int(P), int(C_), int(ldp)))

    def flush(self):
        n = len(self.items)
        if n == 0:
            return
        arr = (_ColsumDesc * n)()
        for d, (part, out, P, C_, ldp) in zip(arr, self.items):
            d.part, d.out, d.P, d.C, d.ldp = ptr(part), ptr(out), P, C_, ldp
        call("omlm_colsum_group", C.cast(arr, C.c_void_p), n, stream_ptr())
        self.items = []


def layernorm_bwd(dy, x, gamma, mean, rstd, dres, dx, dxcast, dgamma, dx_scale=1.0, dres2=None, defer: Optional["ColsumGroup"] = None):
    """dres2 (optional): a second residual-gradient term in the cast type (dxcast's dtype; with dxcast None its own dtype names it).
    defer: the d(gamma) partial rows stay in a workspace of this call's own and their column sum joins the group's one launch."""
    M, D = x.shape
    code = dcode(dxcast.dtype) if dxcast is not None else (dcode(dres2.dtype) if dres2 is not None else F32)
    assert dres2 is None or dcode(dres2.dtype) == code, "dres2 must have the cast type"
    if dgamma is not None and defer is not None:
        rows = min(M, 2048)
        ws = torch.empty(rows * D, device=x.device)     # private, alive until the group's flush: one partial row per workgroup (min(M, 2048) of them)
        call("omlm_layernorm_bwd2", ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dres2), ptr(dx),
             ptr(dxcast), None, ptr(ws), M, D, float(dx_scale), code, dcode(dy.dtype), stream_ptr())
        defer.add(ws, dgamma, rows, D, D)
        return
    ws = _ln_workspace(D, x.device) if dgamma is not None else None       # stream-ordered reuse: one backward at a time
    call("omlm_layernorm_bwd2", ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dres2), ptr(dx),
         ptr(dxcast), ptr(dgamma), ptr(ws), M, D, float(dx_scale), code, dcode(dy.dtype), stream_ptr())


def qk_norm_fwd(q_raw, kv_raw, q_scale, k_scale, q, k, v, H):
    call("omlm_qk_norm_fwd", ptr(q_raw), ptr(kv_raw), ptr(q_scale), ptr(k_scale), ptr(q), ptr(k), ptr(v),
         q_raw.shape[0], H, dcode(q.dtype), stream_ptr())


def qk_norm_bwd(dq, dk, dv, q_raw, kv_raw, q_scale, k_scale, dq_raw, dkv_raw, dq_scale, dk_scale, H):
    call("omlm_qk_norm_bwd", ptr(dq), ptr(dk), ptr(dv), ptr(q_raw), ptr(kv_raw), ptr(q_scale), ptr(k_scale),
         ptr(dq_raw), ptr(dkv_raw), ptr(dq_scale), ptr(dk_scale), q_raw.shape[0], H, dcode(dq_raw.dtype), stream_ptr())


_DBIAS_WS = {}
_DBIAS_RETIRED = []            # outgrown workspaces stay alive (captured graphs may still point at them)


def gemm_qknorm(A, B, C_, scale, norm_out, groups, *, M, N, K, C2=None, c2_col0=0):
    """C = per-head l2-normalised, scaled A B^T in the 16-bit operand type (omlm_gemm_qknorm); norm_out [M, >= groups] fp32."""
    hip.require_gpu(A, "A")
    assert A.dtype in H16 and B.dtype == A.dtype and C_.dtype == A.dtype and norm_out.dtype == torch.float32 and scale.dtype == torch.float32
    assert C2 is None or C2.dtype == A.dtype
    call("omlm_gemm_qknorm", ptr(A), ptr(B), ptr(C_), ptr(C2), int(c2_col0), C2.shape[-1] if C2 is not None else 0, ptr(scale), ptr(norm_out),
         norm_out.shape[-1] if norm_out.dim() > 1 else 1, int(groups), A.numel() // A.shape[-1], B.numel() // B.shape[-1], M, N, K,
         A.shape[-1], B.shape[-1], C_.shape[-1], dcode(A.dtype), stream_ptr())


def qk_norm_bwd2(dq, dk, dv, q, k, qn, kn, q_scale, k_scale, dq_raw, dkv_raw, dq_scale, dk_scale, H):
    call("omlm_qk_norm_bwd2", ptr(dq), ptr(dk), ptr(dv), ptr(q), ptr(k), ptr(qn), ptr(kn), ptr(q_scale), ptr(k_scale),
         ptr(dq_raw), ptr(dkv_raw), ptr(dq_scale), ptr(dk_scale), q.shape[0], H, dcode(dq_raw.dtype), stream_ptr())


class AttnBias:
    """Rel-pos bias table of one attention layer in the two layouts the kernels read: `table` [N, ld] fp32 (row = i - j,
    column = head) and `tableT`, its transposed / zero-padded / log2(e)-scaled form (omlm_attn_bias_prepare).

    q_scale / k_scale (the layer's learned per-dim scales) or qk_bound (an explicit bound on |q . k|, e.g. 1.0 for unit vectors)
    let the 16-bit forward take its exponentials against a fixed reference point instead of a running maximum; without either the
    online softmax runs.  half=True (fp16 attention operands): the reference point sits 15 octaves lower, so the probability numerators
    span half's normal range instead of sitting at its bottom (the kernels fall back to the online softmax if the scales grow too wide)."""

    def __init__(self, table: Optional[torch.Tensor], N: int, H: int, device=None, q_scale=None, k_scale=None,
                 qk_bound: float = 0.0, scale: float = 8.0, half: bool = False, _tableT: Optional[torch.Tensor] = None):
        self.table, self.N, self.H = table, N, H
        dev = table.device if table is not None else device
        if _tableT is not None:                         # (AttnBias.group: the table was written by the grouped launch)
            self.tableT = _tableT
            return
        self.tableT = torch.empty(int(hip.lib().omlm_attn_bias_table_floats(N, H)), device=dev)
        call("omlm_attn_bias_prepare", ptr(table), ptr(self.tableT), N, H, table.shape[-1] if table is not None else 0,
             ptr(q_scale), ptr(k_scale), float(qk_bound), float(scale), 15 if half else 0, stream_ptr())

    @staticmethod
    def group(table: Optional[torch.Tensor], N: int, H: int, device, q_scales, k_scales, scale: float = 8.0, half: bool = False):
        """One AttnBias per layer (q_scales[l], k_scales[l]) over the same rel-pos table, written by ONE launch
        (omlm_attn_bias_prepare_group) instead of one small launch per layer."""
        L = len(q_scales)
        dev = table.device if table is not None else device
        nfl = int(hip.lib().omlm_attn_bias_table_floats(N, H))
        buf = torch.empty(L, nfl, device=dev)
        outs = (C.c_void_p * L)(*[buf[l].data_ptr() for l in range(L)])
        qs = (C.c_void_p * L)(*[t.data_ptr() for t in q_scales])
        ks = (C.c_void_p * L)(*[t.data_ptr() for t in k_scales])
        for t in list(q_scales) + list(k_scales):
            hip.require_gpu(t, "scale")
        call("omlm_attn_bias_prepare_group", ptr(table), C.cast(outs, C.c_void_p), L, N, H, table.shape[-1] if table is not None else 0,
             C.cast(qs, C.c_void_p), C.cast(ks, C.c_void_p), 0.0, float(scale), 15 if half else 0, stream_ptr())
        return [AttnBias(table, N, H, dev, _tableT=buf[l]) for l in range(L)]

    def dbias_workspace(self, B: int, N: int, H: int) -> torch.Tensor:
        """Scratch for the backward's d(bias) partial rows (omlm_mqa_attn_bwd_workspace_bytes): ONE buffer per device, shared by every
        layer and every step (the reduction kernel of a layer consumes it before the next layer's dQ kernel writes it: stream order;
        like _ln_workspace it assumes one backward at a time per device).  It is quadratic in N (B H ceil(N/32)^2 128 bytes: 40 MB at
        B = 32, N = 1116), so one copy per layer was 24 x 53 MB for the musiclm_large leg."""
        n = int(hip.lib().omlm_mqa_attn_bwd_workspace_bytes(B, N, H)) // 4
        key = str(self.tableT.device)
        ws = _DBIAS_WS.get(key)
        if ws is None or ws.numel() < n:
            if ws is not None:
                # a captured HIP graph (graph.py) holds the OLD buffer's address in its kernel nodes: a regrown workspace must not free
                # it, or later replays would write d(bias) partials into memory the allocator has handed to someone else
                _DBIAS_RETIRED.append(ws)
            ws = _DBIAS_WS[key] = torch.empty(n, device=self.tableT.device, dtype=torch.float32)
        return ws


def _attn_bias(bias, N, H, device) -> "AttnBias":
    return bias if isinstance(bias, AttnBias) else AttnBias(bias, N, H, device)


def attn_fwd(q, k, v, bias, keymask, out, lse, B, N, H, scale):
    """bias: an AttnBias (built once per forward by the engine), a raw [N, ld] table, or None."""
    ab = _attn_bias(bias, N, H, q.device)
    call("omlm_mqa_attn_fwd", ptr(q), ptr(k), ptr(v), ptr(ab.table), ptr(ab.tableT), ptr(keymask), ptr(out), ptr(lse),
         B, N, H, float(scale), ab.table.shape[-1] if ab.table is not None else 0, dcode(q.dtype), stream_ptr())


def attn_bwd(q, k, v, bias, keymask, out, dout, lse, delta, dq, dk, dv, dbias, B, N, H, scale, workspace=True):
    """bias: the AttnBias the forward used (its tableT carries the reference point lse is relative to), a raw table, or None.
    workspace=False: d(bias) by device-scope atomics straight into the table (the C ABI's null-workspace form; slower)."""
    ab = _attn_bias(bias, N, H, q.device)
    bias = ab.table
    ws = ab.dbias_workspace(B, N, H) if dbias is not None and workspace else None
    call("omlm_mqa_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(bias), ptr(ab.tableT), ptr(keymask), ptr(out), ptr(dout), ptr(lse),
         ptr(delta), ptr(dq), ptr(dk), ptr(dv), ptr(dbias), ptr(ws), B, N, H, float(scale),
         bias.shape[-1] if bias is not None else 0, dcode(q.dtype), stream_ptr())


def pack_conv_taps(convw: torch.Tensor, F: int, Fp: int) -> torch.Tensor:
    """Reference ds_conv.weight viewed [2F, 3] -> tap-major, padded [3, 2*Fp] in h1's column layout (tiny re-pack)."""
    out = torch.zeros(3, 2 * Fp, device=convw.device, dtype=torch.float32)
    out[:, :F] = convw[:F].t()
    out[:, Fp:Fp + F] = convw[F:].t()
    return out


def pad_vector(v: torch.Tensor, n: int) -> torch.Tensor:
    out = torch.zeros(n, device=v.device, dtype=torch.float32)
    out[: v.numel()] = v
    return out


def ffmid_fwd(h1, convw, gamma, h2, mean, rstd, nseq, F, Fp, p, seed, eps=1e-5, seed_dev=None, drop_bits=None, gh=None):
    """convw: packed taps [3, 2*Fp] (pack_conv_taps); gamma: padded [Fp] (pad_vector)."""
    assert convw.shape == (3, 2 * Fp) and gamma.numel() == Fp
    assert convw.dtype == h1.dtype and gamma.dtype == h1.dtype, "taps / gamma travel in the operand dtype of h1"
    call("omlm_ffmid_fwd", ptr(h1), ptr(convw), ptr(gamma), ptr(h2), ptr(mean), ptr(rstd),
         h1.shape[0], nseq, F, Fp, float(eps), float(p), int(seed), ptr(seed_dev), ptr(drop_bits), ptr(gh), dcode(h1.dtype),
         stream_ptr())


def ffmid_fwd_planes(h1, h1_lo, convw, convw_lo, gamma, gamma_lo, h2, h2_lo, mean, rstd, nseq, F, Fp, p, seed, eps=1e-5, seed_dev=None,
                     drop_bits=None, gh=None):
    """ffmid_fwd on hi/lo planes (precision "fp16ff"): h1, taps and gamma are read as hi + lo, h2 leaves as planes (omlm_ffmid_fwd_planes)."""
    assert convw.shape == (3, 2 * Fp) and gamma.numel() == Fp and h1.dtype in H16
    for t in (h1_lo, convw, convw_lo, gamma, gamma_lo, h2, h2_lo):
        assert t.dtype == h1.dtype, "every plane travels in the operand dtype of h1"
    assert h1_lo.shape == h1.shape and h2_lo.shape == h2.shape and convw_lo.shape == convw.shape
    call("omlm_ffmid_fwd_planes", ptr(h1), ptr(h1_lo), ptr(convw), ptr(convw_lo), ptr(gamma), ptr(gamma_lo), ptr(h2), ptr(h2_lo), ptr(mean),
         ptr(rstd), h1.shape[0], nseq, F, Fp, float(eps), float(p), int(seed), ptr(seed_dev), ptr(drop_bits), ptr(gh), dcode(h1.dtype),
         stream_ptr())


def ffmid_fwd_mx(h1, h1_lo, convw, convw_lo, gamma, gamma_lo, h2, P: "Fp8Planes", mean, rstd, nseq, F, Fp, p, seed, eps=1e-5, seed_dev=None,
                 drop_bits=None, gh=None):
    """ffmid_fwd_planes with h2 leaving as omlm_gemm_mx16's A operand: the half hi plane h2 + its fp8 planes and row scales in P.
    h1_lo: the lo plane of h1 as bf8 bytes (uint8, h1's shape: gemm_mx16's plane output)."""
    assert convw.shape == (3, 2 * Fp) and gamma.numel() == Fp and h1.dtype == torch.float16 and h1_lo.dtype == torch.uint8
    for t in (convw, convw_lo, gamma, gamma_lo, h2):
        assert t.dtype == h1.dtype, "every plane travels in the operand dtype of h1"
    assert h1_lo.shape == h1.shape and convw_lo.shape == convw.shape and P.ld == Fp and P.rows >= h1.shape[0]
    call("omlm_ffmid_fwd_mx", ptr(h1), ptr(h1_lo), ptr(convw), ptr(convw_lo), ptr(gamma), ptr(gamma_lo), ptr(h2), ptr(P.planes), P.stride,
         ptr(P.scale), ptr(mean), ptr(rstd), h1.shape[0], nseq, F, Fp, float(eps), float(p), int(seed), ptr(seed_dev), ptr(drop_bits), ptr(gh),
         stream_ptr())


def ffmid_set_impl(impl: int):
    """0: wave-per-row kernels; 1 (default): column-strip kernels (csrc/ffmid2.hip) wherever their preconditions hold."""
    call("omlm_ffmid_set_impl", int(impl))


def ffmid_bwd_workspace_floats(F, Fp) -> int:
    return int(hip.lib().omlm_ffmid_bwd_workspace_bytes(F, Fp)) // 4


def ffmid_bwd(dh2, h1, convw, gamma, mean, rstd, du_tmp, dh1, dgamma, dconv, workspace, nseq, F, Fp, p, seed, seed_dev=None,
              drop_bits=None, gh=None):
    assert convw.shape == (3, 2 * Fp) and gamma.numel() == Fp
    assert convw.dtype == h1.dtype and gamma.dtype == h1.dtype, "taps / gamma travel in the operand dtype of h1"
    call("omlm_ffmid_bwd", ptr(dh2), ptr(h1), ptr(convw), ptr(gamma), ptr(mean), ptr(rstd), ptr(du_tmp), ptr(dh1),
         ptr(dgamma), ptr(dconv), ptr(workspace), h1.shape[0], nseq, F, Fp, float(p), int(seed), ptr(seed_dev), ptr(drop_bits),
         ptr(gh), dcode(h1.dtype), stream_ptr())


def _ptr_array(tensors: Sequence[Optional[torch.Tensor]]):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = ptr(t) if t is not None else None
    return arr


_INDEX_ERR = {}


def index_error_flag(device) -> torch.Tensor:
    """Per-device int32 flag the gather / cross-entropy kernels OR into when they meet an index past its table (they skip the
    row instead of reading out of bounds).  Reading it synchronises: see :func:`raise_on_index_error`."""
    key = str(device)
    if key not in _INDEX_ERR:
        _INDEX_ERR[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _INDEX_ERR[key]


def raise_on_index_error(device):
    """Raise IndexError if any kernel since the last call met an out-of-range token id / position / label (torch's embedding and
    cross entropy raise a device assert in the same situation).  Synchronises the device: call it where the host already waits
    (the trainer does after every step's loss read-back)."""
    flag = _INDEX_ERR.get(str(device))
    if flag is None:
        return
    v = int(flag.item())
    if v:
        flag.zero_()
        what = [n for b, n in ((1, "token id >= embedding rows"), (2, "position >= absolute position rows"), (4, "label >= vocabulary")) if v & b]
        raise IndexError("open_musiclm_amd: " + ", ".join(what) + " (rows were skipped; check codebook sizes / the token store)")


def _rows_array(tensors):
    if tensors is None:
        return None
    return (C.c_longlong * len(tensors))(*[int(t.shape[0]) if t is not None else 0 for t in tensors])


def embed_fwd(ids, seg, posidx, tables, starts, pos_tables, out):
    B, N = ids.shape
    D = out.shape[-1]
    pos_arr = _ptr_array(pos_tables) if pos_tables is not None else None
    call("omlm_embed_gather_fwd", ptr(ids), ptr(seg), ptr(posidx), _ptr_array(tables), _ptr_array(starts),
         pos_arr, len(tables), ptr(out), B, N, D, _rows_array(tables), _rows_array(pos_tables), ptr(index_error_flag(out.device)),
         stream_ptr())


def embed_bwd(ids, seg, posidx, dtables, dstarts, dpos_tables, dx, alpha):
    B, N = ids.shape
    D = dx.shape[-1]
    pos_arr = _ptr_array(dpos_tables) if dpos_tables is not None else None
    call("omlm_embed_gather_bwd", ptr(ids), ptr(seg), ptr(posidx), _ptr_array(dtables), _ptr_array(dstarts),
         pos_arr, len(dtables), ptr(dx), B, N, D, float(alpha), _rows_array(dtables), _rows_array(dpos_tables),
         ptr(index_error_flag(dx.device)), stream_ptr())


def prepare_train_batch(raw_ids, eos_ids, quantizers, codebooks, pad_id, scores, n_drop, want_labels):
    """One launch for the wrapper's id / label / key-mask construction (omlm_prepare_train_batch).  raw_ids: list of int64 [B, len_s]
    contiguous; returns (ids32 [B, N], keymask uint8 [B, N], labels: list of int32 [B, len_s + 1] or None, lens: the per-sequence token
    counts the engine's layout wants -- len_s + 1 for conditioning sequences, len_last for the predicted one)."""
    n = len(raw_ids)
    B, dev = raw_ids[0].shape[0], raw_ids[0].device
    for t in raw_ids:
        hip.require_gpu(t, "token ids")
        assert t.dtype == torch.int64 and t.dim() == 2 and t.is_contiguous() and t.shape[0] == B
    lens = [int(t.shape[1]) for t in raw_ids]
    N = sum(l + 1 for l in lens) + (n - 1)
    ids32 = torch.empty(B, N, dtype=torch.int32, device=dev)
    keymask = torch.empty(B, N, dtype=torch.uint8, device=dev)
    labels = [torch.empty(B, l + 1, dtype=torch.int32, device=dev) if w else None for l, w in zip(lens, want_labels)]
    arr_i = lambda v: (C.c_int * n)(*[int(x) for x in v])
    call("omlm_prepare_train_batch", _ptr_array(raw_ids), _ptr_array(labels), arr_i(lens), arr_i(eos_ids), arr_i(quantizers), arr_i(codebooks),
         n, B, int(pad_id), ptr(scores), int(n_drop), ptr(ids32), ptr(keymask), N, stream_ptr())
    return ids32, keymask, labels, [l + 1 for l in lens[:-1]] + [lens[-1]]


def ce_fwd(logits, labels, row_lse, nll_sum, V):
    R, ld = logits.shape
    call("omlm_cross_entropy_fwd", ptr(logits), ptr(labels), ptr(row_lse), ptr(nll_sum), R, V, ld,
         ptr(index_error_flag(logits.device)), stream_ptr())


def ce_bwd(logits, labels, row_lse, gscale, coef, dlogits, V):
    R, ld = logits.shape
    call("omlm_cross_entropy_bwd", ptr(logits), ptr(labels), ptr(row_lse), ptr(gscale), float(coef), ptr(dlogits),
         R, V, ld, dlogits.shape[-1], dcode(dlogits.dtype), stream_ptr())


def sumsq_accumulate(g, out, partials=None):
    """out[0] += sum g^2; with `partials` (>= 2048 floats) the reduction order is fixed (bit-reproducible)."""
    call("omlm_sumsq_accumulate", ptr(g), g.numel(), ptr(out), ptr(partials), stream_ptr())


def adamw_clip_step(p, g, m, v, p16, *, lr, beta1, beta2, eps, wd, step, gscale, gnorm_sq, max_norm,
                    decoupled, zero_grad, ls_state=None):
    """p16: optional 16-bit shadow of p (bf16 or fp16: the operand type of the model's precision), written by the same kernel.
    ls_state: the device-side loss-scale block of precision "fp16" (engine.loss_scale_state)."""
    call("omlm_adamw_clip_step", ptr(p), ptr(g), ptr(m), ptr(v), ptr(p16), p.numel(), float(lr), float(beta1),
         float(beta2), float(eps), float(wd), int(step), float(gscale), ptr(gnorm_sq), float(max_norm or 0.0),
         int(decoupled), int(zero_grad), dcode(p16.dtype) if p16 is not None else BF16, ptr(ls_state), stream_ptr())


def loss_scale_update(ls_state, gnorm_sq, *, growth=2.0, backoff=0.5, interval=2000, scale_min=1.0, scale_max=65536.0):
    call("omlm_loss_scale_update", ptr(ls_state), ptr(gnorm_sq), float(growth), float(backoff), int(interval), float(scale_min),
         float(scale_max), stream_ptr())


def cast_pad(src, dst, R, C_, ld_src, ld_dst):
    call("omlm_cast_pad", ptr(src), ptr(dst), R, C_, ld_src, ld_dst, dcode(dst.dtype), stream_ptr())


def transpose_cast(src, dst, R, C_, ld_src, ld_dst):
    """dst[c, r] = cast(src[r, c]) for r < R, c < C_ (fp32 source, dst fp32 or bf16)."""
    call("omlm_transpose_cast", ptr(src), ptr(dst), R, C_, ld_src, ld_dst, dcode(dst.dtype), stream_ptr())


def colsum_accumulate(part, out, P, C_, ldp):
    call("omlm_colsum_accumulate", ptr(part), ptr(out), P, C_, ldp, stream_ptr())


def relpos_first_fwd(w0, b0, pre, z, n, Hd):
    call("omlm_relpos_first_fwd", ptr(w0), ptr(b0), ptr(pre), ptr(z), n, Hd, stream_ptr())


def relpos_first_bwd(ds, dw0, n, Hd):
    call("omlm_relpos_first_bwd", ptr(ds), ptr(dw0), n, Hd, stream_ptr())


def bias_silu_fwd(a, b, pre, z, R, C_):
    call("omlm_bias_silu_fwd", ptr(a), ptr(b), ptr(pre), ptr(z), R, C_, stream_ptr())


def silu_bwd(dz, pre, ds, total):
    call("omlm_silu_bwd", ptr(dz), ptr(pre), ptr(ds), total, stream_ptr())


def bias_add(a, b, out, R, C_, ld):
    call("omlm_bias_add", ptr(a), ptr(b), ptr(out), R, C_, ld, stream_ptr())


def relpos_mlp_fwd(w0, b0, W1, b1, W2, b2, W3, b3, saves, table, n, Hd, H, ldb):
    """The whole rel-pos MLP as one launch (omlm_relpos_mlp_fwd).  saves: None or [pre0, z0, pre1, z1, pre2, z2] ([n, Hd] fp32 each)."""
    sv = saves if saves is not None else [None] * 6
    call("omlm_relpos_mlp_fwd", ptr(w0), ptr(b0), ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(W3), ptr(b3), *[ptr(t) for t in sv], ptr(table),
         int(n), int(Hd), int(H), int(ldb), stream_ptr())


def relpos_mlp_bwd(dtable, W1, W2, W3, saves, scratch, grads, n, Hd, H, ldb):
    """Backward of the fused MLP: saves = [pre0, z0, pre1, z1, pre2, z2]; grads = [gw0, gb0, gW1, gb1, gW2, gb2, gW3, gb3] (accumulated into)."""
    call("omlm_relpos_mlp_bwd", ptr(dtable), ptr(W1), ptr(W2), ptr(W3), *[ptr(t) for t in saves], ptr(scratch), *[ptr(g) for g in grads],
         int(n), int(Hd), int(H), int(ldb), stream_ptr())


def rvq_encode(x, codebooks_T, indices, residual_out, n, D, C_, nstage, idx_stride=None):
    """Residual-VQ chain in the library's distance form (-cdist, first maximum; csrc/optim_misc.hip FORM_CDIST).
    indices: int32 [n, nstage] (or, with nstage == 1, any int32 view whose rows are idx_stride elements apart)."""
    if idx_stride is None or idx_stride == nstage:
        call("omlm_rvq_encode", ptr(x), ptr(codebooks_T), ptr(indices), ptr(residual_out), n, D, C_, nstage, stream_ptr())
    else:
        assert nstage == 1
        call("omlm_rvq_encode_strided", ptr(x), ptr(codebooks_T), ptr(indices), int(idx_stride), ptr(residual_out), n, D, C_,
             stream_ptr())


def nearest_centroid(x, centroids_T, indices, n, D, C_):
    """k-means assign (sklearn MiniBatchKMeans.predict, hf_hubert_kmeans.py:87): squared-difference form; indices int32 [n]."""
    call("omlm_nearest_centroid", ptr(x), ptr(centroids_T), ptr(indices), n, D, C_, stream_ptr())


def vq_accumulate(x, indices, idx_stride, counts, sums, n, D, K):
    call("omlm_vq_accumulate", ptr(x), ptr(indices), int(idx_stride), ptr(counts), ptr(sums), n, D, K, stream_ptr())


def vq_kmeans_update(means, means_T, counts, sums, K, D):
    call("omlm_vq_kmeans_update", ptr(means), ptr(means_T), ptr(counts), ptr(sums), K, D, stream_ptr())


def vq_ema_update(cluster_size, embed_avg, embed, embed_T, counts, sums, total_scratch, K, D, decay, eps):
    call("omlm_vq_ema_update", ptr(cluster_size), ptr(embed_avg), ptr(embed), ptr(embed_T), ptr(counts), ptr(sums),
         ptr(total_scratch), K, D, float(decay), float(eps), stream_ptr())


def sample_topk_gumbel(logits, uniform, out, V, k, temperature, forbid_last):
    B, ld = logits.shape
    call("omlm_sample_topk_gumbel", ptr(logits), ptr(uniform), ptr(out), B, V, ld, int(k), float(temperature),
         int(forbid_last), stream_ptr())


def probe_tr16(out):
    call("omlm_probe_tr16", ptr(out), stream_ptr())
