"""Thin tensor-level wrappers over the C ABI (include/cxrmate_hip.h). torch is used for device memory and streams only:
every function here allocates outputs with torch.empty/zeros and hands raw device pointers to libcxrmate_hip.so."""
from __future__ import annotations

import ctypes as _ct

import contextlib
import gc

import torch

from ._lib import LIB, CxrError

BF16 = torch.bfloat16


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise CxrError("cxrmate_amd kernels need CUDA(HIP) tensors; there is no CPU path")
    return t.data_ptr()


_DEV_INDEX = None


_SIDE_RAW = None        # raw handle of the weight-gradient stream while wgrad_flush issues collected launches (see below)


def _s():
    """Raw hipStream_t of torch's current stream (fast path: no Stream object, no device queries -- this runs ~1500x per step)."""
    global _DEV_INDEX
    if _SIDE_RAW is not None:
        return _SIDE_RAW
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_DEV_INDEX)


def _chk(t, dtype=None):
    assert t.is_contiguous() or t.stride(-1) == 1, "innermost dimension must be contiguous"
    if dtype is not None:
        assert t.dtype == dtype, (t.dtype, dtype)


# ------------------------------------------------------------------------------------------------ GEMM family
GEMM_PROFILE = None     # bench.py: set to a list to collect (flops, start_event, end_event) per GEMM launch on the current stream


def gemm_nt(a, w, bias=None, residual=None, act=0, aux=None, out=None, out_f32=False, accumulate=False, alpha=1.0, drop=None, row_scale=None):
    """out[M,N] = epi(alpha * a[M,K] @ w[N,K]^T). a, w bf16 2-D (row stride arbitrary, unit column stride).
    Train mode: drop = (p, seed, site, rows_per_b, t0) drops the dense output before the residual; row_scale = (scale fp32 [M / rows], rows,
    after_residual) multiplies by a per-image DropPath factor."""
    _chk(a, BF16); _chk(w, BF16)
    M, K = a.shape
    N, K2 = w.shape
    assert K == K2, (a.shape, w.shape)
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else BF16)
    assert out.shape == (M, N) and out.dtype == (torch.float32 if out_f32 else BF16)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    if residual is not None:
        _chk(residual, BF16); assert residual.shape == (M, N)
    if aux is not None:
        _chk(aux, BF16); assert aux.shape == (M, N)
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    LIB.call("cxr_gemm_nt_bf16", _p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), _p(bias), _p(residual),
             residual.stride(0) if residual is not None else 0, _p(aux), aux.stride(0) if aux is not None else 0,
             M, N, K, float(alpha), int(act), int(out_f32), int(accumulate),
             *((float(drop[0]), _p(drop[1]), int(drop[2]), int(drop[3]), int(drop[4])) if drop is not None and drop[0] > 0 else (0.0, None, 0, 1, 0)),
             *((_p(row_scale[0]), int(row_scale[1]), int(bool(row_scale[2]))) if row_scale is not None else (None, 1, 0)), _s())
    if prof is not None:
        e1.record()
        nbytes = 2.0 * (M * K + N * K) + M * N * ((4 if out_f32 else 2) + (2 if residual is not None else 0) + (2 if aux is not None else 0))
        prof.append((2.0 * M * N * K, e0, e1, (M, N, K), nbytes))
    return out


_GEMM_GROUP = __import__("os").environ.get("CXR_GEMM_GROUP", "1") != "0"


class _GemmNtDesc(_ct.Structure):
    """Mirror of `cxr_gemm_nt_desc` (include/cxrmate_hip.h)."""
    _fields_ = [("A", _ct.c_void_p), ("lda", _ct.c_long), ("W", _ct.c_void_p), ("ldw", _ct.c_long), ("C", _ct.c_void_p), ("ldc", _ct.c_long),
                ("bias", _ct.c_void_p), ("residual", _ct.c_void_p), ("ldr", _ct.c_long), ("aux", _ct.c_void_p), ("ldaux", _ct.c_long),
                ("M", _ct.c_int), ("N", _ct.c_int), ("K", _ct.c_int), ("alpha", _ct.c_float), ("act", _ct.c_int), ("out_f32", _ct.c_int),
                ("accumulate", _ct.c_int), ("drop_p", _ct.c_float), ("drop_seed", _ct.c_void_p), ("drop_site", _ct.c_uint),
                ("drop_rows_per_b", _ct.c_int), ("drop_t0", _ct.c_int), ("row_scale", _ct.c_void_p), ("rs_rows", _ct.c_int), ("rs_after", _ct.c_int)]


def gemm_nt_group(problems):
    """Up to three gemm_nt problems with equal N and K in one launch. problems: [(a [M_i,K], w [N,K], bias | None)] -> [out_i [M_i,N] bf16]."""
    assert 1 <= len(problems) <= 3
    if GEMM_PROFILE is not None or not _GEMM_GROUP:       # per-GEMM profiling wants one launch per problem; CXR_GEMM_GROUP=0: A/B switch
        return [gemm_nt(a, w, bias=b) for a, w, b in problems]
    arr = (_GemmNtDesc * len(problems))()
    outs = []
    for d, (a, w, bias) in zip(arr, problems):
        _chk(a, BF16); _chk(w, BF16)
        M, K = a.shape
        N, K2 = w.shape
        assert K == K2, (a.shape, w.shape)
        out = torch.empty((M, N), device=a.device, dtype=BF16)
        if bias is not None:
            assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
            d.bias = _p(bias)
        d.A, d.lda, d.W, d.ldw, d.C, d.ldc = _p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0)
        d.M, d.N, d.K, d.alpha, d.drop_rows_per_b, d.rs_rows = M, N, K, 1.0, 1, 1
        outs.append(out)
    LIB.call("cxr_gemm_nt_group_bf16", _ct.addressof(arr), len(problems), _s())
    return outs


def transpose(x, pad_cols_to=1, out=None):
    """x [R,C] bf16 -> [C, R(padded)] (padding columns are zero)."""
    _chk(x, BF16)
    R, C = x.shape
    Rp = ((R + pad_cols_to - 1) // pad_cols_to) * pad_cols_to
    if out is None:
        out = (torch.zeros if Rp != R else torch.empty)((C, Rp), device=x.device, dtype=BF16)
    LIB.call("cxr_transpose_bf16", _p(x), x.stride(0), _p(out), out.stride(0), R, C, _s())
    return out


class BatchedTranspose:
    """W^T of many bf16 matrices in ONE launch. Sources and destinations are persistent buffers, so the device-side table is built once;
    `pad_cols_to` pads the transposed row length (padding columns stay zero)."""

    def __init__(self, sources, pad_cols_to=None):
        pad_cols_to = pad_cols_to or [1] * len(sources)
        self.outs, rows, tiles = [], [], 0
        for x, pad in zip(sources, pad_cols_to):
            _chk(x, BF16)
            R, C = x.shape
            Rp = ((R + pad - 1) // pad) * pad
            out = (torch.zeros if Rp != R else torch.empty)((C, Rp), device=x.device, dtype=BF16)
            tx, ty = (C + 63) // 64, (R + 63) // 64
            rows.append([x.data_ptr(), out.data_ptr(), x.stride(0), out.stride(0), R, C, tiles, tx])
            tiles += tx * ty
            self.outs.append(out)
        self.sources = list(sources)                   # keep the views alive: the table holds raw pointers
        self.total = tiles
        self.table = torch.tensor(rows, dtype=torch.int64).to(sources[0].device)

    def run(self):
        LIB.call("cxr_transpose_batched_bf16", _p(self.table), len(self.outs), self.total, _s())
        return self.outs


def colsum_into(x, out):
    """out[c] += sum_r x[r,c]  (fp32)."""
    _chk(x, BF16)
    LIB.call("cxr_colsum_bf16", _p(x), x.stride(0), _p(out), x.shape[0], x.shape[1], _s())


_TN_WS = {}             # (device index, raw stream handle) -> fp32 scratch of the deterministic split-K weight-gradient GEMM (launches of one
                        # stream are serial); 16 M floats (64 MB) cover every shape of this model: an output beyond it (splits * I_pad * J_pad floats, more than
                        # 192 splits) falls back to fp32 atomics inside cxr_gemm_tn_bf16 -- reported once per shape, never silent
_TN_WS_FLOATS = 16 << 20
_TN_FALLBACK_SEEN = set()


def _tn_ws(stream, device):
    key = (device.index, stream)
    ws = _TN_WS.get(key)
    if ws is None:
        ws = _TN_WS[key] = torch.empty(_TN_WS_FLOATS, dtype=torch.float32, device=device)
    return ws


def _tn_note_fallback(R, I, J):
    """Warn when a shape cannot use the deterministic partial-tile path (the launch plan comes from the library: cxr_gemm_tn_plan)."""
    import warnings
    splits, need = _ct.c_int(0), _ct.c_long(0)
    LIB.call("cxr_gemm_tn_plan", int(R), int(I), int(J), _ct.byref(splits), _ct.byref(need))
    if splits.value > 1 and (need.value > _TN_WS_FLOATS or splits.value > 192) and (I, J) not in _TN_FALLBACK_SEEN:
        _TN_FALLBACK_SEEN.add((I, J))
        warnings.warn(f"weight-gradient GEMM {I}x{J} over {R} rows: {splits.value} splits need {need.value} scratch floats (> {_TN_WS_FLOATS}) -> fp32 "
                      "atomics, the sum order (last bits of the gradient) is not reproducible for this shape")


class _TnPending(_ct.Structure):
    """cxr_tn_pending (include/cxrmate_hip.h): a weight gradient whose sum over the token splits is still to be added to C / dbias"""
    _fields_ = [("ws", _ct.c_void_p), ("wsb", _ct.c_void_p), ("C", _ct.c_void_p), ("dbias", _ct.c_void_p), ("ldc", _ct.c_long),
                ("I", _ct.c_int), ("J", _ct.c_int), ("Ip", _ct.c_int), ("Jp", _ct.c_int), ("splits", _ct.c_int), ("reserved", _ct.c_int)]


# Deferred split sums of the weight-gradient stream (CXR_TN_DEFER=0: every GEMM followed by its own reduce launch, as before round 4). Each deferred
# GEMM gets scratch of its own from a bump allocator over a few large buffers (288 GB of HBM: the ~2 GB of partial tiles a step leaves behind are not a
# constraint); wgrad_reduce() adds all pending sums with one launch per 40 and hands the scratch back.
_TN_DEFER = __import__("os").environ.get("CXR_TN_DEFER", "1") != "0"
_TN_CHUNK = 64 << 20                                       # floats per arena buffer (256 MB)
_TN_DEFER_FLOATS = int(__import__("os").environ.get("CXR_TN_DEFER_MB", "64")) << 18      # pending partial tiles that trigger a reduce launch (MB -> floats)
_TN_ARENA = {}                                            # device index -> [buffers, index of the current one, floats used in it]
_TN_PENDING = []                                          # (_TnPending, tensors kept alive)


def _tn_arena_alloc(device, n):
    ar = _TN_ARENA.get(device.index)
    if ar is None:
        ar = _TN_ARENA[device.index] = [[], 0, 0]
    bufs = ar[0]
    n = (n + 63) // 64 * 64
    while True:
        if ar[1] < len(bufs) and ar[2] + n <= bufs[ar[1]].numel():
            out = bufs[ar[1]][ar[2]: ar[2] + n]
            ar[2] += n
            return out
        if ar[1] < len(bufs):
            ar[1] += 1
            ar[2] = 0
            continue
        bufs.append(torch.empty(max(n, _TN_CHUNK), dtype=torch.float32, device=device))


_TN_PENDING_PTRS = set()


def wgrad_reduce(side=None):
    """Add every pending split sum of the weight-gradient stream to its gradient (one launch per 40 sums, on that stream, behind the GEMMs that left
    the partial tiles). Called at every point where something may READ a weight gradient: the joins, the fork points of the gradient all-reduce and of
    the early optimiser update, other side-stream work that consumes a GEMM's result, the end of a wgrad_overlap context."""
    if not _TN_PENDING:
        return
    side = WGRAD_STREAM if side is None else side
    assert side is not None, "pending weight-gradient sums outlived their stream"
    arr = (_TnPending * len(_TN_PENDING))(*[d for d, _ in _TN_PENDING])
    LIB.call("cxr_gemm_tn_reduce_batch", _ct.addressof(arr), len(_TN_PENDING), side.cuda_stream)
    _TN_PENDING.clear()
    _TN_PENDING_PTRS.clear()
    for ar in _TN_ARENA.values():
        ar[1] = ar[2] = 0                                  # the same stream orders the next GEMM's partial tiles behind this reduce


def gemm_tn(p, q, out, dbias=None, alpha=1.0):
    """out[I,J] (fp32) += alpha * p[R,I]^T @ q[R,J];  dbias[I] += colsum(p). p, q bf16 row-major (row stride free). Issued from the weight-gradient
    stream's deferred launches (wgrad_flush) the sum over the token splits stays pending until wgrad_reduce()."""
    _chk(p, BF16); _chk(q, BF16)
    R, I = p.shape
    R2, J = q.shape
    assert R == R2 and out.dtype == torch.float32 and out.shape == (I, J) and out.stride(1) == 1
    prof = GEMM_PROFILE
    if prof is not None:
        pstream = WGRAD_STREAM if _SIDE_RAW is not None else None           # the stream the kernel is launched on
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(pstream)
    stream = _s()
    # The batched reduce adds every pending split sum with a plain read-modify-write of its destination: two entries of one batch that name the
    # same C (a weight used twice in one backward, tied / shared parameters), or a one-split launch adding into a C whose sum is still pending,
    # would race. A destination that is already pending is settled first (stream order then serialises the two, as before the deferral).
    if _TN_PENDING_PTRS and (out.data_ptr() in _TN_PENDING_PTRS or (dbias is not None and dbias.data_ptr() in _TN_PENDING_PTRS)):
        wgrad_reduce()
    if _TN_DEFER and _SIDE_RAW is not None and WGRAD_STREAM is not None:
        splits, need = _ct.c_int(0), _ct.c_long(0)
        LIB.call("cxr_gemm_tn_plan", int(R), int(I), int(J), _ct.byref(splits), _ct.byref(need))
        ws = _tn_arena_alloc(p.device, need.value) if (splits.value > 1 and need.value > 0) else None
        pend = _TnPending()
        LIB.call("cxr_gemm_tn_partial_bf16", _p(p), p.stride(0), _p(q), q.stride(0), _p(out), out.stride(0), _p(dbias), R, I, J, float(alpha), _p(ws),
                 0 if ws is None else ws.numel(), _ct.byref(pend), stream)
        if pend.splits > 1:
            _TN_PENDING.append((pend, (out, dbias)))
            _TN_PENDING_PTRS.add(out.data_ptr())
            if dbias is not None:
                _TN_PENDING_PTRS.add(dbias.data_ptr())
            # ... but not for long: the partial tiles should still be in the 256-MB Infinity Cache when the reduce reads them (with every sum of a
            # step pending, ~2 GB of partial tiles went out to HBM and came back: the step got 0.16 ms SLOWER than with one reduce launch per GEMM)
            ar = _TN_ARENA[p.device.index]
            if len(_TN_PENDING) >= 40 or ar[1] > 0 or ar[2] >= _TN_DEFER_FLOATS:
                wgrad_reduce()
    else:
        ws = _tn_ws(stream, p.device)
        if (I, J) not in _TN_FALLBACK_SEEN and I * J > (1 << 20):
            _tn_note_fallback(R, I, J)
        LIB.call("cxr_gemm_tn_bf16", _p(p), p.stride(0), _p(q), q.stride(0), _p(out), out.stride(0), _p(dbias), R, I, J, float(alpha), _p(ws), ws.numel(),
                 stream)
    if prof is not None:
        e1.record(pstream)
        prof.append((2.0 * R * I * J, e0, e1, ("tn", I, J, R), 2.0 * R * (I + J) + 4.0 * I * J))
    return out


# Weight-gradient work (split-K TN GEMMs, depthwise-conv tap sums) only feeds the optimiser, never the dX chain: when WGRAD_STREAM is
# set (training.py) it is issued on that side stream so that these atomics-/latency-bound kernels overlap the MFMA-bound dX GEMMs and
# attention kernels of the main stream. The side stream waits for the main stream's current position (operands are ready), operands
# are pinned with record_stream, and the step joins the side stream before the all-reduce / optimiser (wgrad_join).
WGRAD_STREAM = None


class _on_wgrad_stream:
    def __init__(self, *tensors):
        self.tensors = tensors

    def __enter__(self):
        self.side = WGRAD_STREAM
        if self.side is not None:
            wgrad_flush()                                  # (launches collected by _side_defer come first on the side stream)
            wgrad_reduce()                                 # (... and what follows on that stream may read a weight gradient)
            self.side.wait_stream(torch.cuda.current_stream())
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.side is not None:
            self.ctx.__exit__(*exc)
            for t in self.tensors:
                if t is not None:
                    t.record_stream(self.side)
        return False


_WGRAD_BATCH = max(1, int(__import__("os").environ.get("CXR_WGRAD_BATCH", "4")))
_SIDE_DEFERRED = []     # launches waiting for the next fork point (closures issuing C-ABI launches on the side stream)
_SIDE_PENDING = []      # their operands, kept alive until the next wgrad_join()


def _side_defer(fn, *tensors):
    """Issue `fn` (C-ABI launches only, nothing allocated) on the weight-gradient stream. Ordering against the main stream costs an event record
    on the MAIN stream -- a barrier packet that keeps the next main-stream kernel from starting under the tail of the previous one: ~9 us each,
    145 of them = 1.3 ms of a 41-ms step when every weight-gradient launch had its own. The launches are therefore collected and issued in
    batches of CXR_WGRAD_BATCH behind ONE event (wgrad_flush); every synchronisation with the side stream flushes first. Operands are kept alive
    until the next wgrad_join() instead of record_stream (after the join nothing on the side stream can still read them); as before, nothing may
    overwrite an operand in place before that join -- the side stream runs arbitrarily far behind the main stream."""
    if WGRAD_STREAM is None:
        fn()
        return
    _SIDE_DEFERRED.append(fn)
    _SIDE_PENDING.append(tensors)
    if len(_SIDE_DEFERRED) >= _WGRAD_BATCH:
        wgrad_flush()


def wgrad_flush():
    """Issue the collected weight-gradient launches: one event on the main stream (their operands are complete), the side stream waits for it."""
    global _SIDE_RAW
    if not _SIDE_DEFERRED:
        return
    side = WGRAD_STREAM
    assert side is not None, "deferred weight-gradient launches outlived their stream (wgrad_overlap exits through wgrad_flush)"
    ev = torch.cuda.Event()
    ev.record()
    side.wait_event(ev)
    _SIDE_RAW = side.cuda_stream
    try:
        for fn in _SIDE_DEFERRED:
            fn()
    finally:
        _SIDE_RAW = None
        _SIDE_DEFERRED.clear()


def wgrad_discard():
    """Drop collected launches without issuing them (error path of training.wgrad_overlap)."""
    _SIDE_DEFERRED.clear()
    _SIDE_PENDING.clear()
    _TN_PENDING.clear()
    _TN_PENDING_PTRS.clear()
    for ar in _TN_ARENA.values():
        ar[1] = ar[2] = 0


def wgrad_join(stream=None):
    """stream: the weight-gradient stream to join when called outside the wgrad_overlap context that launched on it (deferred joins)"""
    wgrad_flush()
    stream = WGRAD_STREAM if stream is None else stream
    wgrad_reduce(stream)
    if stream is not None:
        torch.cuda.current_stream().wait_stream(stream)
    _SIDE_PENDING.clear()
    gemm_exclusive(True)                                   # nothing runs beside the main stream any more


_GEMM_EXCLUSIVE = True


def gemm_exclusive(on: bool):
    """on: NT GEMM launches may use the persistent one-workgroup-per-CU kernels (csrc/gemm_ws.hip, gemm_pk.hip: 144 KB of LDS each). The backward
    passes turn it off while weight-gradient kernels run beside the main stream (wgrad_begin) -- such a workgroup cannot share a CU with them
    and the kernels' static work partition would wait for the last CU to free up (measured: 43.9 -> 45.8 ms per step)."""
    global _GEMM_EXCLUSIVE
    if bool(on) != _GEMM_EXCLUSIVE:
        _GEMM_EXCLUSIVE = bool(on)
        LIB.call("cxr_gemm_set_exclusive", int(bool(on)))


_EXCL_ALWAYS = __import__("os").environ.get("CXR_GEMM_EXCL_ALWAYS") == "1"      # lab switch: persistent GEMM kernels also beside the weight-gradient stream


def wgrad_begin():
    """Start of a backward pass: from here on weight-gradient kernels may be running on the side stream (until wgrad_join)."""
    if WGRAD_STREAM is not None and not _EXCL_ALWAYS:
        gemm_exclusive(False)


def wgrad_mark():
    """Event at the side stream's current position (None without a side stream): lets a later consumer wait for exactly the work issued so far
    (the W^T transposes of a forward pass) instead of joining the whole weight-gradient backlog."""
    if WGRAD_STREAM is None:
        return None
    wgrad_flush()
    wgrad_reduce()
    ev = torch.cuda.Event()
    ev.record(WGRAD_STREAM)
    return ev


def wgrad_wait(ev):
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)


_WGRAD_SKIP = __import__("os").environ.get("CXR_WGRAD_SKIP") == "1"       # TIMING EXPERIMENT ONLY (wrong gradients): no weight-gradient GEMM is launched
if _WGRAD_SKIP:
    if __import__("os").environ.get("CXR_DEBUG_TIMING") != "1":
        raise RuntimeError("CXR_WGRAD_SKIP=1 drops every Linear weight gradient (a timing experiment): it is only honoured together with "
                           "CXR_DEBUG_TIMING=1")
    __import__("warnings").warn("CXR_WGRAD_SKIP=1: NO Linear weight gradient is computed in this process -- timing experiment, training results are wrong",
                                RuntimeWarning, stacklevel=1)


def linear_bwd_weight(dy, x, dw, db=None):
    """dw[N,K] += dy[M,N]^T @ x[M,K]; db[N] += colsum(dy)   (one split-K TN kernel; no transposes)."""
    if _WGRAD_SKIP:
        return
    _side_defer(lambda: gemm_tn(dy, x, dw, dbias=db), dy, x, dw, db)


def linear_bwd_input(dy, w_t, **kw):
    """dx[M,K] = dy[M,N] @ w[N,K]   given w_t = w^T [K,N] (bf16)."""
    return gemm_nt(dy, w_t, **kw)


# ------------------------------------------------------------------------------------------------ attention
@contextlib.contextmanager
def graph_capture(graph, pool=None):
    """torch.cuda.graph with Python's cyclic garbage collector held off for the duration of the capture. A collection that starts INSIDE a capture
    (thousands of ctypes launches allocate enough objects to trigger one) destroys whatever garbage earlier work left behind -- old CUDAGraph
    objects, tensors with cross-stream uses -- and their destructors issue HIP calls that are illegal while a stream is capturing in global mode:
    the process aborts (seen in the GPU test suite: 'Fatal Python error: Aborted ... Garbage-collecting' under _DecodeSession.step).
    torch.cuda.graph itself collects once BEFORE the capture begins; nothing new becomes collectable garbage that matters during it."""
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, pool=pool):
            yield
    finally:
        if was:
            gc.enable()


def attention_config(fwd=0, bwd=0):
    """kernel generation of attention / attention_bwd (1 = rounds 1-2, 2 = round 3; 0 keeps the setting): A/B measurements and parity tests"""
    LIB.call("cxr_attn_config", int(fwd), int(bwd))


def attention(q, k, v, heads, scale, kpm=None, causal=False, causal_shift=None, need_lse=False, out=None, drop=None):
    """q [B,Tq,H*64], k/v [B,Tk,H*64] bf16 (last dim contiguous; batch/row strides free). -> out [B,Tq,H*64], lse [B,H,Tq] | None"""
    B, Tq, D = q.shape
    Tk = k.shape[1]
    assert D == heads * 64 and k.shape[2] == D and v.shape[2] == D
    for t in (q, k, v):
        assert t.dtype == BF16 and t.stride(2) == 1
    if out is None:
        out = torch.empty((B, Tq, D), device=q.device, dtype=BF16)
    lse = torch.empty((B, heads, Tq), device=q.device, dtype=torch.float32) if need_lse else None
    if kpm is not None:
        assert kpm.dtype == torch.uint8 and kpm.shape == (B, Tk) and kpm.stride(1) == 1
    if causal_shift is None:
        causal_shift = Tk - Tq
    LIB.call("cxr_attn_fwd_bf16", _p(q), _p(k), _p(v), _p(out), _p(lse), _p(kpm), q.stride(0), q.stride(1), k.stride(0), k.stride(1),
             v.stride(0), v.stride(1), out.stride(0), out.stride(1), kpm.stride(0) if kpm is not None else 0, B, heads, Tq, Tk,
             float(scale), int(causal), int(causal_shift), *_drop_args(drop), _s())
    return out, lse


def attention_q8(q, k, v, heads, scale, out_scale, kpm=None, causal=False, causal_shift=None):
    """attention whose only consumer is an e4m3 GEMM with input scale `out_scale`: -> e4m3 [B, Tq, D] holding bf16(context) / out_scale."""
    B, Tq, D = q.shape
    Tk = k.shape[1]
    for t in (q, k, v):
        assert t.dtype == BF16 and t.stride(2) == 1
    out8 = torch.empty((B, Tq, D), device=q.device, dtype=FP8)
    if causal_shift is None:
        causal_shift = Tk - Tq
    LIB.call("cxr_attn_fwd_q8_bf16", _p(q), _p(k), _p(v), _p(out8), out8.stride(0), out8.stride(1), 1.0 / float(out_scale), _p(kpm), q.stride(0), q.stride(1),
             k.stride(0), k.stride(1), v.stride(0), v.stride(1), kpm.stride(0) if kpm is not None else 0, B, heads, Tq, Tk, float(scale), int(causal),
             int(causal_shift), _s())
    return out8


def _drop_args(drop):
    """drop = None | (p, seed int32/uint32 device tensor [1], site, t0) -> the four trailing dropout arguments of the attention entry points"""
    if drop is None or drop[0] <= 0.0:
        return 0.0, None, 0, 0
    p, seed, site, t0 = drop
    return float(p), _p(seed), int(site), int(t0)


def attention_bwd(q, k, v, o, do, lse, heads, scale, kpm=None, causal=False, causal_shift=None, drop=None, dk_out=None, dv_out=None, dq_out=None):
    """dq_out / dk_out / dv_out: optional pre-allocated [B,T,D] views (column slices of one wider matrix); dk_out and dv_out must have EQUAL strides."""
    B, Tq, D = q.shape
    Tk = k.shape[1]
    dq = dq_out if dq_out is not None else torch.empty((B, Tq, D), device=q.device, dtype=BF16)
    dk = dk_out if dk_out is not None else torch.empty((B, Tk, D), device=q.device, dtype=BF16)
    dv = dv_out if dv_out is not None else torch.empty((B, Tk, D), device=q.device, dtype=BF16)
    assert dk.shape == (B, Tk, D) and dv.shape == (B, Tk, D) and dk.stride() == dv.stride() and dk.stride(2) == 1 and dk.dtype == BF16 and dv.dtype == BF16
    assert dq.shape == (B, Tq, D) and dq.stride(2) == 1 and dq.dtype == BF16
    delta = torch.empty((B, heads, Tq), device=q.device, dtype=torch.float32)
    for t in (q, k, v, o, do):
        assert t.dtype == BF16 and t.stride(2) == 1
    if causal_shift is None:
        causal_shift = Tk - Tq
    do = do if do.stride() == o.stride() else do.contiguous()
    ks, qs = not dk.is_contiguous(), not dq.is_contiguous()
    LIB.call("cxr_attn_bwd_bf16", _p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), _p(kpm),
             q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), o.stride(0), o.stride(1),
             kpm.stride(0) if kpm is not None else 0, B, heads, Tq, Tk, float(scale), int(causal), int(causal_shift), *_drop_args(drop),
             dk.stride(0) if ks else 0, dk.stride(1) if ks else 0, dq.stride(0) if qs else 0, dq.stride(1) if qs else 0, _s())
    return dq, dk, dv


# ------------------------------------------------------------------------------------------------ dropout / DropPath (train mode)
def dropout_add(y, resid, p, seed, site, rows_per_b, t0=0, row_scale=None, out=None):
    """out = resid + f*y (bf16 2-D): f = keep/(1-p) from the counter-based hash of (seed, site, row // rows_per_b, t0 + row % rows_per_b, col),
    or f = row_scale[row // rows_per_b]. resid None -> out = f*y (the backward of the same dropout applied to a gradient)."""
    _chk(y, BF16)
    R, C = y.shape
    if out is None:
        out = torch.empty((R, C), device=y.device, dtype=BF16)
    LIB.call("cxr_dropout_add_bf16", _p(y), y.stride(0), _p(resid), resid.stride(0) if resid is not None else 0, _p(out), out.stride(0), R, C,
             float(p), _p(seed), int(site), int(rows_per_b), int(t0), _p(row_scale), _s())
    return out


def dropout_mask(R, C, p, seed, site, rows_per_b, t0=0, factor=False):
    """Keep mask (uint8 [R,C]) or fp32 factor keep/(1-p) of one dropout site."""
    out = torch.empty((R, C), device=seed.device, dtype=torch.float32 if factor else torch.uint8)
    LIB.call("cxr_dropout_mask", None if factor else _p(out), _p(out) if factor else None, R, C, float(p), _p(seed), int(site), int(rows_per_b),
             int(t0), _s())
    return out


def dropout_site_factors(nsites, Bn, p, seed, site0):
    """DropPath factors [nsites, Bn] fp32 of the consecutive sites site0 .. site0 + nsites - 1 (one launch)."""
    out = torch.empty((nsites, Bn), device=seed.device, dtype=torch.float32)
    LIB.call("cxr_dropout_site_factors", _p(out), int(nsites), int(Bn), float(p), _p(seed), int(site0), _s())
    return out


class TapsLayout:
    """Raw depthwise taps [C,9] (views of the fp32 parameter buffer) -> persistent [9,C] copies, all projections in ONE launch per weight version."""

    def __init__(self, sources):
        rows, blocks = [], 0
        self.outs = []
        for w in sources:
            C = w.shape[0]
            assert w.dtype == torch.float32 and w.is_contiguous() and w.numel() == 9 * C
            out = torch.empty((9, C), device=w.device, dtype=torch.float32)
            rows.append([w.data_ptr(), out.data_ptr(), C, blocks])
            blocks += (9 * C + 255) // 256
            self.outs.append(out)
        self.sources, self.blocks = list(sources), blocks
        self.sig = tuple(w.data_ptr() for w in sources)
        self.table = torch.tensor(rows, dtype=torch.int64).to(sources[0].device)

    def run(self):
        LIB.call("cxr_dwproj_taps_layout", _p(self.table), len(self.outs), self.blocks, _s())
        return self.outs


# ------------------------------------------------------------------------------------------------ normalisation
def layernorm(x, gamma, beta, eps, need_stats=False, out=None):
    """x [rows, C] bf16 (row stride free) -> y, stats[rows,2] | None"""
    _chk(x, BF16)
    rows, C = x.shape
    if out is None:
        out = torch.empty((rows, C), device=x.device, dtype=BF16)
    stats = torch.empty((rows, 2), device=x.device, dtype=torch.float32) if need_stats else None
    LIB.call("cxr_layernorm_fwd_bf16", _p(x), x.stride(0), _p(gamma), _p(beta), _p(out), out.stride(0), _p(stats), rows, C, float(eps), _s())
    return out, stats


def layernorm_q8(x, gamma, beta, eps, scale):
    """LayerNorm whose only consumer is an e4m3 GEMM with input scale `scale`: x [rows, C] bf16 -> e4m3 [rows, C] holding LN(x) / scale (no bf16
    output, no separate quantisation pass)."""
    _chk(x, BF16)
    rows, C = x.shape
    out8 = torch.empty((rows, C), device=x.device, dtype=FP8)
    LIB.call("cxr_layernorm_q8_bf16", _p(x), x.stride(0), _p(gamma), _p(beta), None, 0, _p(out8), out8.stride(0), 1.0 / float(scale), None, rows, C,
             float(eps), _s())
    return out8


def layernorm_bwd(x, dy, gamma, stats, dgamma, dbeta, add=None, out=None, drop=None, row_scale=None, main=True):
    """-> dx, or (dx, dx2) with dx2 = f * dx when drop = (p, seed, site, rows_per_b, t0) (dropout mask re-applied) or row_scale =
    (scale fp32 [rows / rows_per_b], rows_per_b) (DropPath factor) is given. main=False (with drop / row_scale): only dx2 is written -> (None, dx2)."""
    _chk(x, BF16); _chk(dy, BF16)
    rows, C = x.shape
    assert main or drop is not None or row_scale is not None
    if out is None and main:
        out = torch.empty((rows, C), device=x.device, dtype=BF16)
    second = drop is not None or row_scale is not None
    out2 = torch.empty((rows, C), device=x.device, dtype=BF16) if second else None
    ws = None
    if dgamma is not None:
        nb = LIB.load().cxr_layernorm_bwd_grid(rows, C)
        ws = torch.empty((nb, 2, C), device=x.device, dtype=torch.float32)
    side = WGRAD_STREAM is not None and dgamma is not None       # row sum of the (dgamma, dbeta) partials off the critical path
    LIB.call("cxr_layernorm_bwd_bf16", _p(x), x.stride(0), _p(dy), dy.stride(0), _p(gamma), _p(stats), _p(add),
             add.stride(0) if add is not None else 0, _p(out), out.stride(0) if out is not None else C, None if side else _p(dgamma), None if side else _p(dbeta), _p(ws), rows, C, _p(out2),
             out2.stride(0) if second else 0, float(drop[0]) if drop is not None else 0.0, _p(drop[1]) if drop is not None else None,
             int(drop[2]) if drop is not None else 0, int(drop[3]) if drop is not None else (int(row_scale[1]) if row_scale is not None else 1),
             int(drop[4]) if drop is not None else 0, _p(row_scale[0]) if row_scale is not None else None, _s())
    if side:
        _side_defer(lambda: LIB.call("cxr_layernorm_bwd_reduce", _p(ws), rows, C, _p(dgamma), _p(dbeta), _s()), ws, dgamma, dbeta)
    return (out, out2) if second else out


# ------------------------------------------------------------------------------------------------ convolutional pieces
def im2col_pixels(px, ks, stride, pad, kpad):
    Bn, Cin, H, W = px.shape
    assert px.dtype == torch.float32 and px.is_contiguous()
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    col = torch.empty((Bn * Ho * Wo, kpad), device=px.device, dtype=BF16)
    LIB.call("cxr_im2col_nchw_f32", _p(px), _p(col), Bn, Cin, H, W, ks, stride, pad, Ho, Wo, kpad, _s())
    return col, Ho, Wo


def patch_embed_pack(w, out=None):
    """w [64, 3, 7, 7] fp32 -> the packed bf16 operand [64, 192] of patch_embed_s1 (K order (c, ky) x 8 taps, zero padded)"""
    assert w.dtype == torch.float32 and tuple(w.shape) == (64, 3, 7, 7) and w.is_contiguous()
    if out is None:
        out = torch.empty((64, 192), device=w.device, dtype=BF16)
    LIB.call("cxr_patch_embed_pack_f32", _p(w), _p(out), _s())
    return out


def patch_embed_s1(px, wpk, bias, gamma, beta, eps, need_e=False, out=None):
    """px [Bn, 3, H, W] fp32 -> (y [Bn*Ho*Wo, 64] bf16 = LayerNorm(conv7x7s4p2(px) + bias), e | None, stats | None, Ho, Wo): the CvT stage-1 patch
    embedding in one launch (csrc/conv.hip patch_embed_s1_kernel)."""
    Bn, Cin, H, W = px.shape
    assert px.dtype == torch.float32 and px.is_contiguous() and Cin == 3 and H % 4 == 0 and W % 4 == 0 and W <= 384
    Ho, Wo = H // 4, W // 4
    rows = Bn * Ho * Wo
    y = out if out is not None else torch.empty((rows, 64), device=px.device, dtype=BF16)
    assert y.shape == (rows, 64) and y.is_contiguous()
    e = torch.empty((rows, 64), device=px.device, dtype=BF16) if need_e else None
    stats = torch.empty((rows, 2), device=px.device, dtype=torch.float32) if need_e else None
    LIB.call("cxr_patch_embed_s1_f32", _p(px), _p(wpk), _p(bias), _p(gamma), _p(beta), float(eps), _p(e), _p(y), _p(stats), Bn, H, W, _s())
    return y, e, stats, Ho, Wo


def im2col_tokens(x, H, W, stride, pad):
    """x [Bn, H*W, C] bf16 (batch/row strides free) -> col [Bn*Ho*Wo, 9*C]"""
    Bn, L, C = x.shape
    assert L == H * W and x.stride(2) == 1
    Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    col = torch.empty((Bn * Ho * Wo, 9 * C), device=x.device, dtype=BF16)
    LIB.call("cxr_im2col_tok_bf16", _p(x), x.stride(0), x.stride(1), _p(col), Bn, C, H, W, stride, pad, Ho, Wo, _s())
    return col, Ho, Wo


def gemm_nt_conv(x, H, W, stride, pad, w, bias=None, ksz=3):
    """Implicit-GEMM convolution of a token-major activation: x [Bn, H*W, Cin] bf16 (batch / row strides free), w [N, ksz*ksz*Cin] in (ky, kx, c) order
    -> (out [Bn*Ho*Wo, N] bf16 = conv(x) + bias, Ho, Wo). The im2col matrix is never materialised (csrc/gemm.hip ConvA)."""
    Bn, L, C = x.shape
    assert L == H * W and x.stride(2) == 1 and x.dtype == BF16 and w.dtype == BF16 and w.shape[1] == ksz * ksz * C
    Ho, Wo = (H + 2 * pad - ksz) // stride + 1, (W + 2 * pad - ksz) // stride + 1
    N = w.shape[0]
    out = torch.empty((Bn * Ho * Wo, N), device=x.device, dtype=BF16)
    LIB.call("cxr_gemm_nt_conv_bf16", _p(x), x.stride(0), x.stride(1), Bn, H, W, C, int(ksz), int(stride), int(pad), _p(w), w.stride(0), _p(out), out.stride(0),
             _p(bias), N, _s())
    return out, Ho, Wo


def col2im_tokens(dcol, Bn, C, H, W, stride, pad):
    Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    dx = torch.empty((Bn, H * W, C), device=dcol.device, dtype=BF16)
    LIB.call("cxr_col2im_tok_bf16", _p(dcol), _p(dx), dx.stride(0), dx.stride(1), Bn, C, H, W, stride, pad, Ho, Wo, _s())
    return dx


def bn_fold(w, g, b, mean, var, eps):
    """w [C,1,3,3] fp32 etc -> wf [9,C], sh [C] fp32"""
    C = w.shape[0]
    wf = torch.empty((9, C), device=w.device, dtype=torch.float32)
    sh = torch.empty((C,), device=w.device, dtype=torch.float32)
    LIB.call("cxr_bn_fold", _p(w), _p(g), _p(b), _p(mean), _p(var), float(eps), _p(wf), _p(sh), C, _s())
    return wf, sh


def bn_fold_bwd(w, g, mean, var, eps, G, S, dw, dg, db):
    LIB.call("cxr_bn_fold_bwd", _p(w), _p(g), _p(mean), _p(var), float(eps), _p(G), _p(S), _p(dw), _p(dg), _p(db), w.shape[0], _s())


def dwconv_bn(x, H, W, stride, tok0, fold0, fold1=None):
    """x [Bn, tok0+H*W, C] -> y0 (and y1) [Bn, tok0+Ho*Wo, C]"""
    Bn, L, C = x.shape
    assert L == tok0 + H * W and x.dtype == BF16 and x.stride(2) == 1
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y0 = torch.empty((Bn, tok0 + Ho * Wo, C), device=x.device, dtype=BF16)
    y1 = torch.empty_like(y0) if fold1 is not None else None
    LIB.call("cxr_dwconv_bn_fwd_bf16", _p(x), x.stride(0), x.stride(1), _p(fold0[0]), _p(fold0[1]),
             _p(fold1[0]) if fold1 is not None else None, _p(fold1[1]) if fold1 is not None else None, _p(y0), _p(y1),
             y0.stride(0), y0.stride(1), Bn, C, H, W, stride, tok0, _s())
    return y0, y1


def dwconv_bn_bwd_dx(projs, Bn, C, H, W, tok0):
    """projs: list of (dy [Bn, tok0+Ho*Wo, C], wf [9,C], stride) -> dx [Bn, tok0+H*W, C]"""
    dx = torch.empty((Bn, tok0 + H * W, C), device=projs[0][0].device, dtype=BF16)
    args = []
    for i in range(3):
        if i < len(projs):
            dy, wf, st = projs[i]
            args += [_p(dy), _p(wf), dy.stride(0), dy.stride(1), st]
        else:
            args += [None, None, 0, 0, 1]
    LIB.call("cxr_dwconv_bn_bwd_dx_bf16", *args, len(projs), _p(dx), dx.stride(0), dx.stride(1), Bn, C, H, W, tok0, _s())
    return dx


_DW_WS = {}


def _dw_ws(C, device):
    """Scratch for the two-stage per-channel reductions of the depthwise-conv kernels, one per (stream, C): partial rows live only
    between the two kernels of one C-ABI call."""
    key = (_s(), C, device)
    ws = _DW_WS.get(key)
    if ws is None:
        ws = _DW_WS[key] = torch.empty(LIB.load().cxr_dwconv_ws_floats(C), device=device, dtype=torch.float32)
    return ws


def dwconv_bn_bwd_w(x, dy, H, W, stride, tok0, out=None):
    """-> tap sums G [9,C], S [C] (views of `out` fp32 [10, C], overwritten)."""
    Bn, _, C = x.shape
    if out is None:
        out = torch.empty((10, C), device=x.device, dtype=torch.float32)
    LIB.call("cxr_dwconv_bn_bwd_w_bf16", _p(x), x.stride(0), x.stride(1), _p(dy), dy.stride(0), dy.stride(1), _p(out), _p(_dw_ws(C, x.device)),
             Bn, C, H, W, stride, tok0, _s())
    return out[:9], out[9]


def dwconv_stats(x, H, W, stride, tok0, wr0, wr1=None, dy0=None, dy1=None, out=None):
    """Per-channel reductions over the raw depthwise conv outputs c of one or two projections -> fp32 [nproj, 2, C] (train-mode BatchNorm):
    (sum c, sum c^2) in the forward; with dy0 (dy1) [Bn, tok0+Ho*Wo, C]: (sum dy, sum dy*c) for the backward."""
    Bn, L, C = x.shape
    n = 2 if wr1 is not None else 1
    stats = out if out is not None else torch.empty((n, 2, C), device=x.device, dtype=torch.float32)
    if dy1 is not None:
        assert dy1.stride() == dy0.stride()
    LIB.call("cxr_dwconv_stats_bf16", _p(x), x.stride(0), x.stride(1), _p(wr0), _p(wr1), _p(dy0), _p(dy1),
             dy0.stride(0) if dy0 is not None else 0, dy0.stride(1) if dy0 is not None else 0, _p(stats), _p(_dw_ws(C, x.device)), Bn, C, H, W,
             stride, tok0, _s())
    return stats


def dwconv_bn_train_fwd_stats(x, H, W, stride, tok0, eps, momentum, projs):
    """Train-mode BatchNorm forward bookkeeping of one or two depthwise projections of x in ONE call. projs: list of dicts with wt (raw taps
    [9,C]), w ([C,1,3,3] parameter), g, b, run_mean, run_var. -> per projection ((wf, sh), mean, rstd); count."""
    Bn, L, C = x.shape
    outs, args = [], []
    for p in projs:
        wf = torch.empty((9, C), device=x.device, dtype=torch.float32)
        aux = torch.empty((3, C), device=x.device, dtype=torch.float32)                 # sh, mean, rstd
        outs.append(((wf, aux[0]), aux[1], aux[2]))
        args += [_p(p["wt"]), _p(p["w"]), _p(p["g"]), _p(p["b"]), _p(p["run_mean"]), _p(p["run_var"]), _p(aux[1]), _p(aux[2]), _p(wf), _p(aux[0])]
    if len(projs) == 1:
        args += [None] * 10
    LIB.call("cxr_dwconv_bn_train_fwd_stats_bf16", _p(x), x.stride(0), x.stride(1), Bn, C, H, W, int(stride), int(tok0), float(eps), float(momentum),
             _p(_dw_ws(C, x.device)), *args, _s())
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    return outs, Bn * Ho * Wo


def dwconv_bn_train_bwd_stats(x, H, W, stride, tok0, projs):
    """Train-mode BatchNorm backward bookkeeping in ONE call. projs: list of dicts with wt, dy, g, mean, rstd, dg, db. -> coef [3,C] per projection."""
    Bn, L, C = x.shape
    coefs, args = [], []
    for p in projs:
        coef = torch.empty((3, C), device=x.device, dtype=torch.float32)
        coefs.append(coef)
        args += [_p(p["wt"]), _p(p["dy"]), _p(p["g"]), _p(p["mean"]), _p(p["rstd"]), _p(p["dg"]), _p(p["db"]), _p(coef)]
    if len(projs) == 1:
        args += [None] * 8
    else:
        assert projs[0]["dy"].stride() == projs[1]["dy"].stride()
    dy = projs[0]["dy"]
    LIB.call("cxr_dwconv_bn_train_bwd_stats_bf16", _p(x), x.stride(0), x.stride(1), Bn, C, H, W, int(stride), int(tok0), _p(_dw_ws(C, x.device)),
             *args, dy.stride(0), dy.stride(1), _s())
    return coefs


def bn_train_finalize(stats, count, w, g, b, eps, momentum, run_mean, run_var):
    """-> (wf [9,C], sh [C]) folded with the batch statistics, mean [C], rstd [C]; running stats updated in place."""
    C = w.shape[0]
    wf = torch.empty((9, C), device=w.device, dtype=torch.float32)
    sh = torch.empty((C,), device=w.device, dtype=torch.float32)
    mr = torch.empty((2, C), device=w.device, dtype=torch.float32)
    LIB.call("cxr_bn_train_finalize", _p(stats), int(count), _p(w), _p(g), _p(b), float(eps), float(momentum), _p(run_mean), _p(run_var),
             _p(mr[0]), _p(mr[1]), _p(wf), _p(sh), C, _s())
    return (wf, sh), mr[0], mr[1]


def bn_train_bwd_coef(g, mean, rstd, SD, count, dg, db):
    C = g.shape[0]
    coef = torch.empty((3, C), device=g.device, dtype=torch.float32)
    LIB.call("cxr_bn_train_bwd_coef", _p(g), _p(mean), _p(rstd), _p(SD), int(count), _p(dg), _p(db), _p(coef), C, _s())
    return coef


def dwconv_bn_train_dc_(x, wr, coef, dy, H, W, stride, tok0):
    Bn, _, C = x.shape
    LIB.call("cxr_dwconv_bn_train_dc_bf16", _p(x), x.stride(0), x.stride(1), _p(wr), _p(coef), _p(dy), dy.stride(0), dy.stride(1), Bn, C, H, W,
             stride, tok0, _s())
    return dy


def tap_grad_accum(G, dw):
    LIB.call("cxr_tap_grad_accum", _p(G), _p(dw), G.shape[1], _s())


# ---- fused query / key / value convolutional projections (csrc/dwproj.hip): one LDS-staged pass per layer and step of the BatchNorm algebra
class DwProj(_ct.Structure):
    """Mirror of `cxr_dwproj` (include/cxrmate_hip.h): one depthwise projection of the shared activation; device pointers as integers."""
    _fields_ = [("stride", _ct.c_int), ("taps", _ct.c_void_p), ("shift", _ct.c_void_p), ("y", _ct.c_void_p), ("y_bs", _ct.c_long), ("y_rs", _ct.c_long),
                ("w", _ct.c_void_p), ("gamma", _ct.c_void_p), ("beta", _ct.c_void_p), ("run_mean", _ct.c_void_p), ("run_var", _ct.c_void_p),
                ("mean", _ct.c_void_p), ("rstd", _ct.c_void_p), ("taps_out", _ct.c_void_p), ("shift_out", _ct.c_void_p),
                ("dgamma", _ct.c_void_p), ("dbeta", _ct.c_void_p), ("coef", _ct.c_void_p), ("GS", _ct.c_void_p), ("dw", _ct.c_void_p),
                ("yf", _ct.c_void_p), ("yf_bs", _ct.c_long), ("yf_rs", _ct.c_long)]


_DWP_PTRS = ("taps", "shift", "w", "gamma", "beta", "run_mean", "run_var", "mean", "rstd", "taps_out", "shift_out", "dgamma", "dbeta", "coef", "GS", "dw")
_DWP_WS = {}


def _dwproj_ws(Bn, C, H, W, device):
    key = (_s(), Bn, C, H, W, device)
    ws = _DWP_WS.get(key)
    if ws is None:
        n = LIB.load().cxr_dwproj_ws_floats(Bn, C, H, W)
        if n <= 0:
            raise CxrError(f"cxr_dwproj_ws_floats({Bn}, {C}, {H}, {W}) -> {n}")
        ws = _DWP_WS[key] = torch.empty(n, device=device, dtype=torch.float32)
    return ws


def _dwproj_array(projs):
    """projs: list of dicts {stride, <tensor fields of cxr_dwproj>, y: bf16 [Bn, L, C]} -> (ctypes array, keep-alive list)"""
    arr = (DwProj * len(projs))()
    for d, p in zip(arr, projs):
        d.stride = int(p["stride"])
        for k in _DWP_PTRS:
            t = p.get(k)
            if t is not None:
                assert t.dtype == torch.float32 and t.is_contiguous(), k
                setattr(d, k, _p(t))
        y = p.get("y")
        if y is not None:
            assert y.dtype in (BF16, FP8) and y.stride(2) == 1
            d.y, d.y_bs, d.y_rs = _p(y), y.stride(0), y.stride(1)
        yf = p.get("yf")
        if yf is not None:
            assert yf.dtype == BF16 and yf.stride(2) == 1
            d.yf, d.yf_bs, d.yf_rs = _p(yf), yf.stride(0), yf.stride(1)
    return arr


def _dwproj_geo(x, H, W, tok0):
    Bn, L, C = x.shape
    assert L == tok0 + H * W and x.dtype == BF16 and x.stride(2) == 1, (x.shape, H, W, tok0)
    return Bn, C


def dwproj_apply(x, H, W, tok0, projs):
    """projs: [{stride, taps (folded [9,C]), shift [C]}] -> list of y [Bn, tok0+Ho*Wo, C]; class rows copied through."""
    Bn, C = _dwproj_geo(x, H, W, tok0)
    ys = []
    for p in projs:
        Ho, Wo = (H - 1) // p["stride"] + 1, (W - 1) // p["stride"] + 1
        ys.append(torch.empty((Bn, tok0 + Ho * Wo, C), device=x.device, dtype=BF16))
    arr = _dwproj_array([dict(p, y=y) for p, y in zip(projs, ys)])
    LIB.call("cxr_dwproj_apply_bf16", _p(x), x.stride(0), x.stride(1), Bn, C, H, W, int(tok0), _ct.addressof(arr), len(projs), _s())
    return ys


def dwproj_apply_q8(x, H, W, tok0, projs, scales):
    """dwproj_apply with e4m3 outputs y / scales[q] (the consumers are e4m3 GEMMs with these input scales)."""
    Bn, C = _dwproj_geo(x, H, W, tok0)
    ys = []
    for p in projs:
        Ho, Wo = (H - 1) // p["stride"] + 1, (W - 1) // p["stride"] + 1
        ys.append(torch.empty((Bn, tok0 + Ho * Wo, C), device=x.device, dtype=FP8))
    arr = _dwproj_array([dict(p, y=y) for p, y in zip(projs, ys)])
    inv = (_ct.c_float * len(projs))(*[1.0 / float(s_) for s_ in scales])
    LIB.call("cxr_dwproj_apply_q8", _p(x), x.stride(0), x.stride(1), Bn, C, H, W, int(tok0), _ct.addressof(arr), len(projs), inv, _s())
    return ys


def dwproj_bn_train_stats(x, H, W, tok0, eps, momentum, projs):
    """Train-mode BatchNorm statistics of all projections in one pass. projs: [{stride, taps (raw [9,C]), w [C,9], gamma, beta, run_mean, run_var}]
    (running statistics moved in place) -> per projection dict(taps=folded, shift=, mean=, rstd=, count=, stride=)."""
    Bn, C = _dwproj_geo(x, H, W, tok0)
    outs = torch.empty((len(projs), 12, C), device=x.device, dtype=torch.float32)         # folded taps [9,C], shift, mean, rstd
    full = []
    for i, p in enumerate(projs):
        full.append(dict(p, taps_out=outs[i, :9], shift_out=outs[i, 9], mean=outs[i, 10], rstd=outs[i, 11]))
    arr = _dwproj_array(full)
    LIB.call("cxr_dwproj_bn_train_stats_bf16", _p(x), x.stride(0), x.stride(1), Bn, C, H, W, int(tok0), float(eps), float(momentum), _ct.addressof(arr),
             len(projs), _p(_dwproj_ws(Bn, C, H, W, x.device)), _s())
    res = []
    for i, p in enumerate(projs):
        Ho, Wo = (H - 1) // p["stride"] + 1, (W - 1) // p["stride"] + 1
        res.append(dict(stride=p["stride"], taps=outs[i, :9], shift=outs[i, 9], mean=outs[i, 10], rstd=outs[i, 11], count=Bn * Ho * Wo))
    return res


def dwproj_bn_train_bwd_stats(x, H, W, tok0, projs):
    """projs: [{stride, taps (raw), y = dL/d(BN out), gamma, mean, rstd, dgamma, dbeta (accumulated)[, yf = the projection's forward output, beta]}]
    -> coef list ([3,C] each: a, kb, kc). With yf + beta the pass streams (dy, yf) instead of recomputing the convolution (see csrc/dwproj.hip)."""
    Bn, C = _dwproj_geo(x, H, W, tok0)
    coefs = torch.empty((len(projs), 3, C), device=x.device, dtype=torch.float32)
    arr = _dwproj_array([dict(p, coef=coefs[i]) for i, p in enumerate(projs)])
    LIB.call("cxr_dwproj_bn_train_bwd_stats_bf16", _p(x), x.stride(0), x.stride(1), Bn, C, H, W, int(tok0), _ct.addressof(arr), len(projs),
             _p(_dwproj_ws(Bn, C, H, W, x.device)), _s())
    return [coefs[i] for i in range(len(projs))]


def dwproj_dc_taps_(x, H, W, tok0, projs, need_GS=False):
    """projs: [{stride, taps (raw), y (rewritten in place as dc when coef is given), coef or None, dw [C,9] (+= tap sums) or None}]
    -> GS [nproj, 10, C] (9 tap sums, then sum dc) when need_GS (or when a projection has no dw)."""
    Bn, C = _dwproj_geo(x, H, W, tok0)
    GS = None
    if need_GS or any(p.get("dw") is None for p in projs):
        GS = torch.empty((len(projs), 10, C), device=x.device, dtype=torch.float32)
    arr = _dwproj_array([dict(p, GS=None if GS is None else GS[i]) for i, p in enumerate(projs)])
    LIB.call("cxr_dwproj_dc_taps_bf16", _p(x), x.stride(0), x.stride(1), Bn, C, H, W, int(tok0), _ct.addressof(arr), len(projs),
             _p(_dwproj_ws(Bn, C, H, W, x.device)), _s())
    return GS


def dwproj_dx(projs, Bn, C, H, W, tok0):
    """projs: [{stride, taps, y = dc}] -> dx [Bn, tok0+H*W, C]"""
    dx = torch.empty((Bn, tok0 + H * W, C), device=projs[0]["y"].device, dtype=BF16)
    arr = _dwproj_array(projs)
    LIB.call("cxr_dwproj_dx_bf16", _p(dx), dx.stride(0), dx.stride(1), Bn, C, H, W, int(tok0), _ct.addressof(arr), len(projs), _s())
    return dx



# ------------------------------------------------------------------------------------------------ embeddings / integer ops
def bert_embed(ids, tt, pid, word, typ, posw, gamma, beta, eps, T, pos_offset=0, need_sum=False, drop=None, out_dal=False):
    """out_dal: the output rows in the decode activation layout (dal_rows(R) x C elements, see dec_gemm)."""
    R = ids.numel()
    C = word.shape[1]
    out = torch.empty((dal_rows(R) if out_dal else R, C), device=ids.device, dtype=BF16)
    ssum = torch.empty((R, C), device=ids.device, dtype=BF16) if need_sum else None
    stats = torch.empty((R, 2), device=ids.device, dtype=torch.float32) if need_sum else None
    LIB.call("cxr_bert_embed_fwd", _p(ids), _p(tt), _p(pid), _p(word), _p(typ), _p(posw), _p(gamma), _p(beta), float(eps), _p(ssum), _p(out),
             _p(stats), R, T, pos_offset, C, *((float(drop[0]), _p(drop[1]), int(drop[2])) if drop is not None and drop[0] > 0 else (0.0, None, 0)),
             int(bool(out_dal)), _s())
    return out, ssum, stats


def bert_embed_bwd(dsum, ids, tt, pid, dword, dtype_, dpos, T, pos_offset, padding_idx):
    # embedding-table gradients only feed the optimiser: weight-gradient stream
    _side_defer(lambda: LIB.call("cxr_bert_embed_bwd", _p(dsum), _p(ids), _p(tt), _p(pid), _p(dword), _p(dtype_), _p(dpos), ids.numel(), T, pos_offset,
                                 int(padding_idx), dsum.shape[1], _s()), dsum, ids, tt, pid, dword, dtype_, dpos)


_SPECIAL_CACHE = {}


def special_tensors(special, sections, device):
    """Device copies of (special ids, section ids), cached: generation asks for the same handful every step (no per-step H2D copy)."""
    sections = tuple(sections) if sections is not None else tuple(range(len(special) + 1))
    key = (tuple(special), sections, str(device))
    hit = _SPECIAL_CACHE.get(key)
    if hit is None:
        hit = (torch.tensor(list(special), dtype=torch.int64, device=device), torch.tensor(list(sections), dtype=torch.int64, device=device))
        _SPECIAL_CACHE[key] = hit
    return hit


def token_type_ids(ids, special, sections, past=False, out=None):
    """Device version of token_ids_to_token_type_ids[_past]; ids int64 [B,T] (row stride free)."""
    assert ids.dtype == torch.int64 and ids.stride(1) == 1
    B, T = ids.shape
    sp, se = special_tensors(special, sections, ids.device)
    if out is None:
        out = torch.empty((B, 1 if past else T), dtype=torch.int64, device=ids.device)
    LIB.call("cxr_token_type_ids", _p(ids), ids.stride(0), B, T, _p(sp), _p(se), len(special), _p(out), out.stride(0), int(past), _s())
    return out


def decode_step_inputs(ids, strip, cur, special0, special1, sections, half_rows, mask_token_id, new_id, tt, pos, mask, tt_hist, pos_hist):
    """One launch for the per-step input assembly of a cached decode (see cxr_decode_step_inputs). ids int64 [rows, Lmax]; outputs are the
    caller's persistent buffers: new_id / tt / pos [rows, 1] int64, mask uint8 [rows, Lmax] | None, histories [rows, Lmax] int64 | None."""
    rows = ids.shape[0]
    sp0, se = special_tensors(special0, sections, ids.device)
    sp1, _ = special_tensors(special1, sections, ids.device)
    LIB.call("cxr_decode_step_inputs", _p(ids), ids.stride(0), rows, int(strip), int(cur), _p(sp0), len(special0), _p(sp1), len(special1), _p(se),
             int(half_rows), int(mask_token_id if mask is not None else -1), _p(new_id), _p(tt), _p(pos), _p(mask),
             mask.stride(0) if mask is not None else 0, _p(tt_hist), _p(pos_hist), tt_hist.stride(0) if tt_hist is not None else 0, _s())


def decode_step_embed(ids, strip, cur, special0, special1, sections, half_rows, mask_token_id, new_id, tt, pos, mask, tt_hist, pos_hist, word, typ,
                      posw, gamma, beta, eps, drop=None, out_dal=True):
    """decode_step_inputs + the BERT embeddings of the new token in one launch -> embedding output bf16 (decode activation layout)."""
    rows = ids.shape[0]
    sp0, se = special_tensors(special0, sections, ids.device)
    sp1, _ = special_tensors(special1, sections, ids.device)
    out = torch.empty((dal_rows(rows) if out_dal else rows, word.shape[1]), device=ids.device, dtype=BF16)
    LIB.call("cxr_decode_step_embed", _p(ids), ids.stride(0), rows, int(strip), int(cur), _p(sp0), len(special0), _p(sp1), len(special1), _p(se),
             int(half_rows), int(mask_token_id if mask is not None else -1), _p(new_id), _p(tt), _p(pos), _p(mask),
             mask.stride(0) if mask is not None else 0, _p(tt_hist), _p(pos_hist), tt_hist.stride(0) if tt_hist is not None else 0,
             _p(word), _p(typ), _p(posw), _p(gamma), _p(beta), float(eps), _p(out), int(bool(out_dal)),
             *((float(drop[0]), _p(drop[1]), int(drop[2])) if drop is not None and drop[0] > 0 else (0.0, None, 0)), _s())
    return out


def mask_position_ids(ids, mask_token_id):
    assert ids.dtype == torch.int64 and ids.stride(1) == 1
    B, T = ids.shape
    mask = torch.empty((B, T), dtype=torch.uint8, device=ids.device)
    pos = torch.empty((B, T), dtype=torch.int64, device=ids.device)
    LIB.call("cxr_mask_position_ids", _p(ids), ids.stride(0), B, T, int(mask_token_id), _p(mask), mask.stride(0), _p(pos), pos.stride(0), _s())
    return mask, pos


def image_mask(pixel_values, tokens):
    """pixel_values [B,N,3,H,W] fp32 -> uint8 [B, N*tokens]"""
    B, N = pixel_values.shape[:2]
    assert pixel_values.dtype == torch.float32 and pixel_values.is_contiguous()
    out = torch.empty((B, N * tokens), dtype=torch.uint8, device=pixel_values.device)
    LIB.call("cxr_image_mask", _p(pixel_values), pixel_values.stride(1), B * N, tokens, _p(out), _s())
    return out


# ------------------------------------------------------------------------------------------------ losses / selection
def ce_weights(labels, ignore_index, mode=0, reward=None, T=0):
    R = labels.numel()
    w = torch.empty((R,), dtype=torch.float32, device=labels.device)
    LIB.call("cxr_ce_weights", _p(labels), R, int(ignore_index), mode, _p(reward), T, _p(w), _s())
    return w


def softmax_ce(logits, labels, ignore_index, row_w, thr=None, need_grad=True):
    """logits fp32 or bf16 [R,V]; -> loss scalar tensor (fp32, [1]), row_loss [R], dlogits bf16 [R,V] | None"""
    R, V = logits.shape
    assert logits.dtype in (torch.float32, BF16) and logits.stride(1) == 1
    if logits.dtype == BF16 and (logits.stride(0) % 8 or V > 32768 or logits.data_ptr() % 16):
        logits = logits.float()              # the bf16 kernel keeps a whole row in registers and reads 16-byte vectors
    row_loss = torch.empty((R,), dtype=torch.float32, device=logits.device)
    Vp = ((V + 63) // 64) * 64            # zero-padded so that V can be the K dimension of the LM-head backward GEMMs
    dl = torch.empty((R, Vp), dtype=BF16, device=logits.device) if need_grad else None
    LIB.call("cxr_softmax_ce", _p(logits), logits.stride(0), _p(labels), int(ignore_index), _p(thr), _p(row_w), _p(row_loss), _p(dl),
             dl.stride(0) if dl is not None else 0, R, V, int(logits.dtype == BF16), _s())
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
    LIB.call("cxr_ce_reduce", _p(row_loss), _p(row_w), R, _p(loss), _s())
    return loss, row_loss, (dl[:, :V] if dl is not None else None)


def topk_threshold(logits, k, top_p=1.0, temperature=1.0):
    """Per-row value threshold of TopKLogitsWarper(k) followed by TopPLogitsWarper(top_p) at `temperature` (logits are the RAW scores)."""
    R, V = logits.shape
    thr = torch.empty((R,), dtype=torch.float32, device=logits.device)
    LIB.call("cxr_topk_threshold", _p(logits), logits.stride(0), R, V, int(k), float(top_p), float(temperature), _p(thr), _s())
    return thr


def select_token(logits, mode=0, temperature=1.0, top_k=0, u=None, unfinished=None, eos=-1, pad=0, need_margin=False, out=None, top_p=1.0,
                 n_sample=-1):
    """out: int64 [R] (element stride free: a column view of the running id buffer works). mode 1, n_sample < R: rows [0, n_sample) are sampled,
    rows [n_sample, R) take the argmax in the same launch."""
    R, V = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    nxt = out if out is not None else torch.empty((R,), dtype=torch.int64, device=logits.device)
    assert nxt.dim() == 1 and nxt.shape[0] == R
    margin = torch.empty((R,), dtype=torch.float32, device=logits.device) if need_margin else None
    LIB.call("cxr_select_token", _p(logits), logits.stride(0), R, V, mode, float(temperature), int(top_k), float(top_p), _p(u), _p(nxt),
             nxt.stride(0) if R > 1 else 1, _p(unfinished), int(eos), int(pad), _p(margin), int(n_sample), _s())
    return nxt, margin


def log_softmax_rows_(x, add_row=None):
    LIB.call("cxr_log_softmax_rows", _p(x), x.stride(0), x.shape[0], x.shape[1], _p(add_row), _s())
    return x


# ------------------------------------------------------------------------------------------------ optimiser / plumbing
def adamw_step(p, g, m, v, p16, lr, b1, b2, eps, wd, step, gscale=1.0, step_dev=None):
    """step >= 1: host step count. step == 0 with step_dev (int32 device scalar): the kernel reads the count itself (graph replay)."""
    LIB.call("cxr_adamw_step", _p(p), _p(g), _p(m), _p(v), _p(p16), p.numel(), float(lr), float(b1), float(b2), float(eps), float(wd),
             int(step), _p(step_dev), float(gscale), _s())


def increment_(counter):
    LIB.call("cxr_increment_i32", _p(counter), _s())


def cast_to_bf16(src, dst=None):
    if dst is None:
        dst = torch.empty(src.shape, dtype=BF16, device=src.device)
    LIB.call("cxr_cast_f32_to_bf16", _p(src), _p(dst), src.numel(), _s())
    return dst


def cast_to_f32(src, dst=None):
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    LIB.call("cxr_cast_bf16_to_f32", _p(src), _p(dst), src.numel(), _s())
    return dst


def add(a, b, out=None):
    rows, C = a.shape
    if out is None:
        out = torch.empty((rows, C), dtype=BF16, device=a.device)
    LIB.call("cxr_add_bf16", _p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), rows, C, _s())
    return out


def copy_rows(src, dst):
    """src, dst [B, rows, C] bf16 views (arbitrary batch/row strides)."""
    B, rows, C = src.shape
    assert dst.shape == src.shape and src.stride(2) == 1 and dst.stride(2) == 1
    LIB.call("cxr_copy_rows_bf16", _p(src), src.stride(0), src.stride(1), _p(dst), dst.stride(0), dst.stride(1), B, rows, C, _s())
    return dst


def bcast_row(row_f32, dst):
    """dst [B, L, C] bf16: dst[:, 0, :] = row"""
    LIB.call("cxr_bcast_row_f32_bf16", _p(row_f32), _p(dst), dst.stride(0), dst.shape[0], dst.shape[2], _s())


def sum_row0_into(src, out_f32):
    LIB.call("cxr_sum_row0_bf16_f32", _p(src), src.stride(0), _p(out_f32), src.shape[0], src.shape[2], _s())


def gather_batch(src, idx, rows, dst):
    """dst[b, :rows] = src[idx[b], :rows]  for [B, Tmax, C] bf16 caches."""
    B, _, C = dst.shape
    LIB.call("cxr_gather_batch_bf16", _p(src), src.stride(0), src.stride(1), _p(dst), dst.stride(0), dst.stride(1), _p(idx), B, rows, C, _s())
    return dst


# ------------------------------------------------------------------------------------------------ fp8 (e4m3) linear layers of the frozen encoder
FP8 = torch.float8_e4m3fn
FP8_MAX = 448.0


def quantize_fp8(x, scale, out=None):
    """bf16 [M,K] -> e4m3 [M,K] holding x / scale (saturating)."""
    _chk(x, BF16)
    M, K = x.shape
    if out is None:
        out = torch.empty((M, K), dtype=FP8, device=x.device)
    LIB.call("cxr_quantize_fp8", _p(x), x.stride(0), _p(out), out.stride(0), M, K, 1.0 / float(scale), _s())
    return out


def gemm_nt_fp8(a8, w8, scale, bias=None, residual=None, act=0, out_scale=None, want_bf16=True, row_scale=None):
    """epi(scale * a8[M,K] @ w8[N,K]^T) -> (bf16 [M,N] | None, e4m3 [M,N] holding value / out_scale | None)."""
    assert a8.dtype == FP8 and w8.dtype == FP8 and a8.stride(1) == 1 and w8.stride(1) == 1
    M, K = a8.shape
    N = w8.shape[0]
    c = torch.empty((M, N), dtype=BF16, device=a8.device) if want_bf16 else None
    c8 = torch.empty((M, N), dtype=FP8, device=a8.device) if out_scale is not None else None
    LIB.call("cxr_gemm_nt_fp8", _p(a8), a8.stride(0), _p(w8), w8.stride(0), _p(c), N, _p(c8), N, 1.0 / float(out_scale) if out_scale is not None else 0.0,
             M, N, K, float(scale), _p(bias), _p(residual), residual.stride(0) if residual is not None else 0, int(act),
             *((_p(row_scale[0]), int(row_scale[1]), int(bool(row_scale[2]))) if row_scale is not None else (None, 1, 0)), _s())
    return c, c8


def gather_batch_multi(srcs, idx, rows, dsts):
    """dsts[i][b, :rows] = srcs[i][idx[b], :rows] for up to 16 [B, Tmax, C] bf16 caches of one geometry, one launch."""
    n = len(srcs)
    B, _, C = dsts[0].shape
    pin = (_ct.c_void_p * n)(*[s.data_ptr() for s in srcs])
    pout = (_ct.c_void_p * n)(*[d.data_ptr() for d in dsts])
    LIB.call("cxr_gather_batch_multi_bf16", pin, pout, n, dsts[0].stride(0), dsts[0].stride(1), _p(idx), B, rows, C, _s())


def beam_ws(rows, V, nb, device):
    """Scan workspace of beam_step for `rows` = beams * studies rows over a vocabulary of V."""
    return torch.empty(rows * ((V + 4095) // 4096) * (2 + 4 * nb), dtype=torch.float32, device=device)


def beam_step(logits, running, sequences, run_scores, beam_scores, finished, unsat, allhit, beam_idx, cur, max_length, eos, div, ws=None):
    """One device-side beam-search step (csrc/decode.hip beam_scan_kernel + beam_step_kernel; include/cxrmate_hip.h cxr_beam_step). Beam-major rows."""
    nb, B, L = running.shape
    R, V = logits.shape
    assert R == nb * B and logits.dtype == torch.float32 and running.is_contiguous() and sequences.is_contiguous()
    if ws is None:
        ws = beam_ws(R, V, nb, logits.device)
    assert ws.numel() >= R * ((V + 4095) // 4096) * (2 + 4 * nb)
    LIB.call("cxr_beam_step", _p(logits), logits.stride(0), _p(running), _p(sequences), _p(run_scores), _p(beam_scores), _p(finished), _p(unsat),
             _p(allhit), _p(beam_idx), _p(ws), B, nb, V, L, int(cur), int(max_length), int(eos), float(div), _s())


def topk_rows(x, k):
    """x fp32 [R, n] -> (values [R,k], indices [R,k]) in descending order."""
    R, n = x.shape
    vals = torch.empty((R, k), dtype=torch.float32, device=x.device)
    inds = torch.empty((R, k), dtype=torch.int64, device=x.device)
    LIB.call("cxr_topk_rows", _p(x), x.stride(0), R, n, int(k), _p(vals), _p(inds), _s())
    return vals, inds


def gelu_bwd(dy, u):
    assert dy.is_contiguous() and u.is_contiguous() and dy.shape == u.shape
    dx = torch.empty_like(dy)
    LIB.call("cxr_gelu_bwd_bf16", _p(dy), _p(u), _p(dx), dy.numel(), _s())
    return dx


def cosine_rows(a, b, eps=1e-8):
    """a, b fp32 [R, C] -> [R]"""
    R, C = a.shape
    out = torch.empty((R,), dtype=torch.float32, device=a.device)
    LIB.call("cxr_cosine_rows_f32", _p(a), a.stride(0), _p(b), b.stride(0), _p(out), R, C, float(eps), _s())
    return out


def _ln_args(ln):
    """ln = None | (gamma, beta, eps, stats_out | None) -> the four A-operand LayerNorm arguments of the skinny GEMM entry points"""
    if ln is None:
        return None, None, 0.0, None
    return _p(ln[0]), _p(ln[1]), float(ln[2]), _p(ln[3])


def gemm_skinny(a, w, bias=None, residual=None, act=0, out=None, out_f32=False, drop=None, ln_a=None, ln_r=None):
    """Decode-step linear: a [M<=64, K] @ w[N, K]^T (+bias, GELU, dropout, +residual). Weight-streaming kernel (no LDS staging).
    drop = (p, seed, site, t): train-mode dropout of the dense output, row m = sequence m at absolute position t.
    ln_a = (gamma, beta, eps, stats_out): `a` is a raw pre-LayerNorm sum, normalised inside the kernel (row statistics published to stats_out
    fp32 [M,2]); ln_r = (stats, gamma, beta): the residual is LayerNorm(residual) with published statistics."""
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else BF16)
    LIB.call("cxr_gemm_skinny_bf16", _p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), _p(bias), _p(residual),
             residual.stride(0) if residual is not None else 0, M, N, K, int(act), int(out_f32), *_ln_args(ln_a),
             *((_p(ln_r[0]), _p(ln_r[1]), _p(ln_r[2])) if ln_r is not None else (None, None, None)), *_drop_args(drop), _s())
    return out


def pack_mask_bits(kpm, out=None):
    """key-padding mask uint8 [B, T] (1 = attend) -> bit words int32 [B, ceil(T/32)] for attention_decode(kpm_bits=...)"""
    B, T = kpm.shape
    if out is None:
        out = torch.empty((B, (T + 31) // 32), device=kpm.device, dtype=torch.int32)
    LIB.call("cxr_pack_mask_bits", _p(kpm), kpm.stride(0), B, T, _p(out), out.shape[1], _s())
    return out


def attention_decode(q, k, v, heads, scale, kpm=None, out=None, drop=None, head_major=False, wg_keys=0, out_dal=False, kpm_bits=None):
    """q [B,1,H*64] (or [B,H*64]); k, v [B or B/2,Tk,H*64] views (batch/row strides free) -> [B, H*64]. With B/2 K/V rows, query rows
    b and b + B/2 share K/V row b."""
    B = q.shape[0]
    share = B // k.shape[0]
    assert k.shape[0] * share == B and share in (1, 2, 4)
    if head_major:                         # k, v [Bkv, H, Tk, 64] contiguous: every (b, h) K/V stream is one contiguous block
        Tk = k.shape[2]
        k_bs, k_rs, v_bs, v_rs, hs = k.stride(0), k.stride(2), v.stride(0), v.stride(2), k.stride(1)
    else:
        Tk = k.shape[1]
        k_bs, k_rs, v_bs, v_rs, hs = k.stride(0), k.stride(1), v.stride(0), v.stride(1), 64
    key = ("attn_decode_ws", _s(), B * heads, q.device)
    ws = _DW_WS.get(key)
    if ws is None:
        ws = _DW_WS[key] = torch.empty(B * heads * 8 * 66, device=q.device, dtype=torch.float32)
    D = heads * 64
    if out is None:
        out = torch.empty((dal_rows(B) if out_dal else B, D), device=q.device, dtype=BF16)
    if kpm_bits is not None:                                    # bit words from pack_mask_bits (row stride in bytes)
        mask, mask_bs, bits = kpm_bits, kpm_bits.stride(0) * 4, 1
    else:
        mask, mask_bs, bits = kpm, (kpm.stride(0) if kpm is not None else 0), 0
    LIB.call("cxr_attn_decode_bf16", _p(q), _p(k), _p(v), _p(out), _p(mask), q.stride(0), k_bs, k_rs, v_bs, v_rs,
             out.stride(0), mask_bs, B, heads, Tk, float(scale), share, _p(ws), int(hs), *_drop_args(drop),
             int(wg_keys), int(bool(out_dal)), bits, _s())
    return out


def pack_cross_kv(k, v, heads, out=None):
    """k, v [Bkv, Tk, H*64] bf16 (equal strides) -> (Kp, Vp): fragment-ordered copies for attention_cross_mfma (one launch, once per decode)."""
    Bkv, Tk, D = k.shape
    assert v.shape == k.shape and k.stride() == v.stride() and k.stride(2) == 1 and Tk % 32 == 0 and D == heads * 64
    kp, vp = out if out is not None else (torch.empty(Bkv * Tk * D, device=k.device, dtype=BF16), torch.empty(Bkv * Tk * D, device=k.device, dtype=BF16))
    LIB.call("cxr_pack_cross_kv_bf16", _p(k), _p(v), k.stride(0), k.stride(1), _p(kp), _p(vp), Bkv, heads, Tk, _s())
    return kp, vp


def attention_cross_mfma_ok(B, Bkv, Tk):
    return Tk % 32 == 0 and Tk <= 8 * 1152 and B % Bkv == 0 and B // Bkv <= 4


def attention_cross_mfma(q, packed, Bkv, Tk, heads, scale, kpm_bits=None, out=None, drop=None, out_dal=False):
    """Cross-attention of a cached decode step on the matrix cores (csrc/decode.hip attn_cross_mfma_kernel): q [B, H*64]; packed = pack_cross_kv(k, v)
    of the Bkv studies; kpm_bits uint32 [Bkv, words] | None. Query rows b + g*Bkv share K/V row b. -> [B, H*64] (or DAL)."""
    B = q.shape[0]
    D = heads * 64
    assert attention_cross_mfma_ok(B, Bkv, Tk) and packed[0].numel() == Bkv * Tk * D
    if out is None:
        out = torch.empty((dal_rows(B) if out_dal else B, D), device=q.device, dtype=BF16)
    key = ("attn_decode_ws", _s(), B * heads, q.device)
    ws = _DW_WS.get(key)
    if ws is None:
        ws = _DW_WS[key] = torch.empty(B * heads * 8 * 66, device=q.device, dtype=torch.float32)
    LIB.call("cxr_attn_cross_mfma_bf16", _p(q), _p(packed[0]), _p(packed[1]), _p(out), _p(kpm_bits), q.stride(0), out.stride(0),
             kpm_bits.stride(0) if kpm_bits is not None else 0, B, heads, Tk, float(scale), B // Bkv, *_drop_args(drop), int(bool(out_dal)), _p(ws), _s())
    return out


def attention_cross_mfma_q(x_dal, M, stats, eps, wp, bc, packed, Bkv, Tk, heads, scale, kpm_bits=None, out=None, drop=None, out_dal=True):
    """attention_cross_mfma with the query projection inside the kernel: x_dal = raw hidden rows (decode activation layout, M rows x 768), stats = its
    producer's partial row statistics [tiles, M, 2], (wp, bc) = the packed query Linear with the LayerNorm folded in (dec_pack_weight)."""
    D = heads * 64
    assert D == 768 and M % Bkv == 0 and M // Bkv <= 4 and Tk % 32 == 0 and Tk <= 1920 and packed[0].numel() == Bkv * Tk * D
    if out is None:
        out = torch.empty((dal_rows(M) if out_dal else M, D), device=x_dal.device, dtype=BF16)
    LIB.call("cxr_attn_cross_mfma_q_bf16", _p(x_dal), x_dal.shape[0] // 16, M, _p(stats), stats.shape[0], float(eps), _p(wp), _p(bc), _p(packed[0]), _p(packed[1]),
             _p(out), _p(kpm_bits), out.stride(0), kpm_bits.stride(0) if kpm_bits is not None else 0, M, heads, Tk, float(scale), M // Bkv, *_drop_args(drop),
             int(bool(out_dal)), _s())
    return out


# ------------------------------------------------------------------------------------------------ decode-step linear layers (csrc/decode_gemm.hip)
def dal_rows(M):
    """rows of the buffer that holds M rows in the decode activation layout (16-row tiles; 3 tiles are stored as 4)"""
    t = (M + 15) // 16
    return 16 * (4 if t == 3 else t)


class _DecProb(_ct.Structure):
    """Mirror of `cxr_dec_gemm_prob` (include/cxrmate_hip.h)."""
    _fields_ = [("Wp", _ct.c_void_p), ("bc", _ct.c_void_p), ("C", _ct.c_void_p), ("ldc", _ct.c_long), ("N", _ct.c_int), ("c_dal", _ct.c_int),
                ("fold", _ct.c_int), ("no_bias", _ct.c_int), ("lr_Ap", _ct.c_void_p), ("lr_B", _ct.c_void_p), ("lr_site", _ct.c_uint)]


class _DecDesc(_ct.Structure):
    """Mirror of `cxr_dec_gemm_desc`."""
    _fields_ = [("A", _ct.c_void_p), ("M", _ct.c_int), ("K", _ct.c_int), ("nprob", _ct.c_int), ("act", _ct.c_int), ("out_f32", _ct.c_int),
                ("nc_hint", _ct.c_int), ("mt_hint", _ct.c_int), ("p", _DecProb * 3), ("stats", _ct.c_void_p), ("stats_tiles", _ct.c_int), ("eps", _ct.c_float),
                ("residual", _ct.c_void_p), ("ldr", _ct.c_long), ("rgb", _ct.c_void_p), ("out_stats", _ct.c_void_p),
                ("drop_p", _ct.c_float), ("drop_seed", _ct.c_void_p), ("drop_site", _ct.c_uint), ("drop_t", _ct.c_int),
                ("lr_p", _ct.c_float), ("lr_seed", _ct.c_void_p), ("lr_scale", _ct.c_float), ("lr_t", _ct.c_int)]


def dec_pack_weight(w, gamma=None, beta=None, bias=None, out=None):
    """w bf16 [N, K] row-major -> (Wp bf16 [N*K] in MFMA-fragment order, bc fp32 [N, 2] = (bias', colsum)); gamma / beta: the LayerNorm folded
    into the weights (W' = W diag(gamma), bias' = bias + W beta). `out` = (Wp, bc) refreshes existing buffers in place (graph-captured pointers)."""
    N, K = w.shape
    assert w.dtype == BF16 and w.stride(1) == 1
    wp, bc = out if out is not None else (torch.empty(((N + 15) // 16) * 16 * K, device=w.device, dtype=BF16),
                                          torch.empty((N, 2), device=w.device, dtype=torch.float32))
    LIB.call("cxr_dec_pack_weight_bf16", _p(w), w.stride(0), _p(gamma), _p(beta), _p(bias), _p(wp), _p(bc), N, K, _s())
    return wp, bc


def dec_pack_lora(a, gamma=None, beta=None, out=None):
    """LoRA A bf16 [8, K] -> fragment-ordered B operand [A diag(gamma); A diag(beta)] (bf16 [16*K])"""
    r, K = a.shape
    assert r == 8 and a.is_contiguous()
    o = out if out is not None else torch.empty(16 * K, device=a.device, dtype=BF16)
    LIB.call("cxr_dec_pack_lora_bf16", _p(a), _p(gamma), _p(beta), _p(o), K, _s())
    return o


def dec_to_dal(x, want_stats=False, want_out=True):
    """row-major bf16 [M, K] -> (decode-activation-layout copy | None, per-16-column-tile row statistics fp32 [K/16, M, 2] | None)"""
    M, K = x.shape
    out = torch.empty((dal_rows(M), K), device=x.device, dtype=BF16) if want_out else None
    st = torch.empty((K // 16, M, 2), device=x.device, dtype=torch.float32) if want_stats else None
    LIB.call("cxr_dec_to_dal_bf16", _p(x), x.stride(0), M, K, _p(out), _p(st), _s())
    return out, st


def dec_from_dal(x, M, K):
    out = torch.empty((M, K), device=x.device, dtype=BF16)
    LIB.call("cxr_dec_from_dal_bf16", _p(x), M, K, _p(out), out.stride(0), _s())
    return out


def dec_gemm(a, M, K, probs, act=0, out_f32=False, stats=None, eps=0.0, residual=None, rgb=None, out_stats=False, drop=None, lora=None, nc_hint=0,
             mt_hint=0):
    """Decode-step linear layer(s) on `a` (bf16, decode activation layout of [M, K]). probs: 1-3 dicts {wp, bc, N, fold=False, out=None |
    row-major tensor view [M, N] (e.g. KV-cache rows), lora=(Ap, B, site) | None}; a problem without `out` gets a fresh DAL buffer.
    stats = (partials fp32 [tiles, M, 2]) of the LayerNorm input (of `a` for folded problems, of `residual` with rgb = fp32 [N, 2] (gamma, beta));
    residual: DAL buffer added to problem 0; out_stats: also return problem 0's output partials; drop = (p, seed, site, t);
    lora = (p, seed, scale, t). Returns ([outputs], out_stats | None)."""
    d = _DecDesc()
    d.A, d.M, d.K, d.nprob, d.act, d.out_f32, d.nc_hint, d.mt_hint = _p(a), M, K, len(probs), int(act), int(out_f32), int(nc_hint), int(mt_hint)
    outs = []
    for i, pr in enumerate(probs):
        q = d.p[i]
        N = pr["N"]
        out = pr.get("out")
        if out is None:
            if out_f32:
                out = torch.empty((M, N), device=a.device, dtype=torch.float32)
                q.c_dal, q.ldc = 0, N
            else:
                out = torch.empty((dal_rows(M), N), device=a.device, dtype=BF16)
                q.c_dal, q.ldc = 1, 0
        else:
            assert out.shape[0] == M and out.shape[1] == N and out.stride(1) == 1
            q.c_dal, q.ldc = 0, out.stride(0)
        outs.append(out)
        q.Wp, q.bc, q.C, q.N, q.fold, q.no_bias = _p(pr["wp"]), _p(pr["bc"]), _p(out), N, int(bool(pr.get("fold"))), 0
        lo = pr.get("lora")
        if lo is not None:
            q.lr_Ap, q.lr_B, q.lr_site = _p(lo[0]), _p(lo[1]), int(lo[2])
    if stats is not None:
        d.stats, d.stats_tiles, d.eps = _p(stats), stats.shape[0], float(eps)
    if residual is not None:
        d.residual, d.ldr = _p(residual), 0
        if rgb is not None:
            d.rgb = _p(rgb)
    ost = None
    if out_stats:
        ost = torch.empty((probs[0]["N"] // 16, M, 2), device=a.device, dtype=torch.float32)
        d.out_stats = _p(ost)
    if drop is not None and drop[0] > 0:
        d.drop_p, d.drop_seed, d.drop_site, d.drop_t = float(drop[0]), _p(drop[1]), int(drop[2]), int(drop[3])
    if lora is not None:
        d.lr_p, d.lr_seed, d.lr_scale, d.lr_t = float(lora[0]), _p(lora[1]), float(lora[2]), int(lora[3])
    LIB.call("cxr_dec_gemm_bf16", _ct.addressof(d), _s())
    return outs, ost


def gemm_skinny3(a, w0, b0, c0, w1, b1, c1, w2, b2, c2, ln_a=None, lora0=None, lora1=None, lora_in=None):
    """c_i = a @ w_i^T + b_i for three equally-shaped projections in ONE launch (outputs may be strided KV-cache rows).
    lora0 / lora1 = (t fp32 [M,8], B bf16 [N,8]): rank-8 term of problem 0 / 1 with a precomputed down-projection t; or
    lora_in = dict(A0, B0, A1, B1, p, seed, site0, site1, tpos, scale) (K = 768): the down-projections are computed inside the kernel."""
    M, K = a.shape
    N = w0.shape[0]
    assert w0.stride(0) == w1.stride(0) == w2.stride(0)
    li = lora_in
    if li is not None:
        assert K == 768 and li["A0"].is_contiguous() and li["A1"].is_contiguous()
        extra = (None, _p(li["B0"]), None, _p(li["B1"]), _p(li["A0"]), _p(li["A1"]), float(li["p"]), _p(li["seed"]), int(li["site0"]), int(li["site1"]),
                 int(li["tpos"]), float(li["scale"]))
    else:
        extra = (*((_p(lora0[0]), _p(lora0[1])) if lora0 else (None, None)), *((_p(lora1[0]), _p(lora1[1])) if lora1 else (None, None)),
                 None, None, 0.0, None, 0, 0, 0, 0.0)
    LIB.call("cxr_gemm_skinny3_bf16", _p(a), a.stride(0), _p(w0), _p(b0), _p(c0), c0.stride(0), _p(w1), _p(b1), _p(c1), c1.stride(0),
             _p(w2), _p(b2), _p(c2), c2.stride(0), w0.stride(0), M, N, K, *_ln_args(ln_a), *extra, _s())


# ------------------------------------------------------------------------------------------------ LoRA branch (train mode)
def lora_down(x, W0, t0=None, drop0=None, W1=None, drop1=None, rows_per_b=1, tpos0=0, seed=None, ln=None, scale=1.0, w_is_b=False):
    """t_i [M,8] fp32 = scale * dropout_i(x) @ A_i^T  (w_is_b: W_i is a lora_B [N,8] and the contraction runs over its rows: dy @ B).
    drop_i = (p, site); ln = (gamma, beta, eps) normalises the raw rows first. One launch for both problems."""
    M, K = x.shape
    outs = [torch.empty((M, 8), device=x.device, dtype=torch.float32) for _ in range(2 if W1 is not None else 1)]
    st = (lambda W: (1, W.stride(0))) if w_is_b else (lambda W: (W.stride(0), 1))
    p0, s0 = drop0 if drop0 is not None else (0.0, 0)
    p1, s1 = drop1 if drop1 is not None else (0.0, 0)
    LIB.call("cxr_lora_down_bf16", _p(x), x.stride(0), M, K, _p(W0), *st(W0), _p(outs[0]), float(p0), int(s0),
             _p(W1), *(st(W1) if W1 is not None else (0, 0)), _p(outs[1]) if W1 is not None else None, float(p1), int(s1), _p(seed),
             int(rows_per_b), int(tpos0), _p(ln[0]) if ln else None, _p(ln[1]) if ln else None, float(ln[2]) if ln else 0.0, float(scale), _s())
    return outs if W1 is not None else outs[0]


def lora_up_add_(y, t, W, w_is_b, drop=None, rows_per_b=1, tpos0=0, seed=None):
    """y[m,n] += f(m,n) * sum_r t[m,r] W(r,n) in place; W = lora_B [N,8] (w_is_b) or lora_A [8,N]; drop = (p, site) re-applies a mask."""
    M, N = y.shape
    rs, cs = (1, W.stride(0)) if w_is_b else (W.stride(0), 1)
    p, site = drop if drop is not None else (0.0, 0)
    LIB.call("cxr_lora_up_add_bf16", _p(y), y.stride(0), M, N, _p(t), _p(W), rs, cs, float(p), _p(seed), int(site), int(rows_per_b), int(tpos0), _s())
    return y


class _LoraDownDesc(_ct.Structure):
    """Mirror of `cxr_lora_down_desc` (include/cxrmate_hip.h)."""
    _fields_ = [("x", _ct.c_void_p), ("ldx", _ct.c_long), ("W", _ct.c_void_p), ("w_rs", _ct.c_long), ("w_cs", _ct.c_long), ("t", _ct.c_void_p),
                ("p", _ct.c_float), ("site", _ct.c_uint)]


class _LoraUpDesc(_ct.Structure):
    """Mirror of `cxr_lora_up_desc`."""
    _fields_ = [("y", _ct.c_void_p), ("ldy", _ct.c_long), ("t", _ct.c_void_p), ("W", _ct.c_void_p), ("w_rs", _ct.c_long), ("w_cs", _ct.c_long),
                ("p", _ct.c_float), ("site", _ct.c_uint)]


class _LoraOuterDesc(_ct.Structure):
    """Mirror of `cxr_lora_outer_desc`."""
    _fields_ = [("a", _ct.c_void_p), ("lda", _ct.c_long), ("t", _ct.c_void_p), ("G", _ct.c_void_p), ("g_ks", _ct.c_long), ("g_rs", _ct.c_long),
                ("p", _ct.c_float), ("site", _ct.c_uint)]


def _lora_w_strides(W, w_is_b):
    return (1, W.stride(0)) if w_is_b else (W.stride(0), 1)


def lora_down_multi(probs, rows_per_b=1, tpos0=0, seed=None, scale=1.0):
    """Teacher-forced pass: t_i [M,8] fp32 = scale * dropout_i(x_i) @ W_i^T for up to two problems in one launch (matrix cores).
    probs: [dict(x=[M,K] bf16, W=, w_is_b=False, drop=(p, site) | None)] -> [t_i]"""
    M, K = probs[0]["x"].shape
    arr = (_LoraDownDesc * len(probs))()
    outs = []
    for d, pr in zip(arr, probs):
        x, W = pr["x"], pr["W"]
        assert x.shape == (M, K) and x.dtype == BF16 and x.stride(1) == 1 and W.dtype == BF16
        t = torch.empty((M, 8), device=x.device, dtype=torch.float32)
        p, site = pr.get("drop") or (0.0, 0)
        rs, cs = _lora_w_strides(W, pr.get("w_is_b", False))
        d.x, d.ldx, d.W, d.w_rs, d.w_cs, d.t, d.p, d.site = _p(x), x.stride(0), _p(W), rs, cs, _p(t), float(p), int(site)
        outs.append(t)
    LIB.call("cxr_lora_down_multi_bf16", _ct.addressof(arr), len(probs), M, K, _p(seed), int(rows_per_b), int(tpos0), float(scale), _s())
    return outs


def lora_up_add_multi_(probs, rows_per_b=1, tpos0=0, seed=None):
    """y_i[m,n] += f_i(m,n) * sum_r t_i[m,r] W_i(r,n) for up to two problems in one launch; two problems naming the same y are added in one
    read-modify-write. probs: [dict(y=[M,N] bf16 (row stride free), t=, W=, w_is_b=, drop=(p, site) | None)]"""
    M, N = probs[0]["y"].shape
    arr = (_LoraUpDesc * len(probs))()
    for d, pr in zip(arr, probs):
        y, t, W = pr["y"], pr["t"], pr["W"]
        assert y.shape == (M, N) and y.dtype == BF16 and y.stride(1) == 1 and t.shape == (M, 8) and t.is_contiguous() and t.dtype == torch.float32
        p, site = pr.get("drop") or (0.0, 0)
        rs, cs = _lora_w_strides(W, pr.get("w_is_b", False))
        d.y, d.ldy, d.t, d.W, d.w_rs, d.w_cs, d.p, d.site = _p(y), y.stride(0), _p(t), _p(W), rs, cs, float(p), int(site)
    LIB.call("cxr_lora_up_add_multi_bf16", _ct.addressof(arr), len(probs), M, N, _p(seed), int(rows_per_b), int(tpos0), _s())


def lora_outer_multi_into(probs, scale=1.0, rows_per_b=1, tpos0=0, seed=None):
    """G_i[k*g_ks + r*g_rs] += scale * sum_m f_i(m,k) a_i[m,k] t_i[m,r] for up to four problems in one launch.
    probs: [dict(a=[M,K] bf16, t=[M,8] fp32, G= fp32, g_ks=, g_rs=, drop=(p, site) | None)]"""
    M, K = probs[0]["a"].shape
    arr = (_LoraOuterDesc * len(probs))()
    for d, pr in zip(arr, probs):
        a, t, G = pr["a"], pr["t"], pr["G"]
        assert a.shape == (M, K) and a.dtype == BF16 and a.stride(1) == 1 and t.shape == (M, 8) and t.is_contiguous() and G.dtype == torch.float32
        p, site = pr.get("drop") or (0.0, 0)
        d.a, d.lda, d.t, d.G, d.g_ks, d.g_rs, d.p, d.site = _p(a), a.stride(0), _p(t), _p(G), int(pr["g_ks"]), int(pr["g_rs"]), float(p), int(site)
    LIB.call("cxr_lora_outer_multi_bf16", _ct.addressof(arr), len(probs), M, K, float(scale), _p(seed), int(rows_per_b), int(tpos0), _s())


def lora_outer_into(a, t, G, g_ks, g_rs, scale=1.0, drop=None, rows_per_b=1, tpos0=0, seed=None):
    """G[k*g_ks + r*g_rs] += scale * sum_m f(m,k) a[m,k] t[m,r]  (fp32 G; strides in elements)."""
    M, K = a.shape
    p, site = drop if drop is not None else (0.0, 0)
    LIB.call("cxr_lora_outer_bf16", _p(a), a.stride(0), M, K, _p(t), _p(G), int(g_ks), int(g_rs), float(scale), float(p), _p(seed), int(site),
             int(rows_per_b), int(tpos0), _s())
