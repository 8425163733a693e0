"""Point-major building blocks for the per-point heads: activations live as (B, N, C) so that one point
is one contiguous row (what both the row gathers and the per-point GEMMs want).  GroupNorm(+ReLU) runs
through csrc/gn.hip.  The 1x1 convolutions (M4:556-603,644-699,713) are bf16 GEMMs: csrc/gemm.hip (hand-written MFMA
kernels: forward with fused bias + GroupNorm statistics, weight gradient through the hardware transpose read) where
that kernel is at least as fast as the library -- output widths <= 128 and the 256->256 layers whose statistics pass
it absorbs -- and torch.nn.functional.linear (hipBLASLt) for the wide layers, where csrc/gemm.hip reaches 0.6-0.8x of
the library's speed (tools/gemm_bench.py).  GCANET_GEMM=own|lib forces one or the other everywhere."""
import os

import torch

from . import _lib


def _run(name, like, *args):
    with _lib.on_device(like):
        _lib.call(name, *args, _lib.stream_of(like))


class ZeroArena:
    """One pre-zeroed device allocation for the small accumulators the library is asked to zero ("zeroed by the call":
    GroupNorm sums, weight-gradient tiles, dgamma/dbeta, ...): ~35 fills of ~4.5 us each per training step become ONE
    fill of the bytes handed out in the previous step.  Registered with the library (gcn_zero_arena_register), which
    then skips the fill of any span inside it.  Contract: call begin_step() once per step before the first forward and
    use zeroed_scratch() buffers only within the step that took them."""
    live = None

    def __init__(self, device, nbytes=96 << 20):
        self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self.off = 0
        self.dirty = 0
        with torch.cuda.device(device):
            _lib.call("gcn_zero_arena_register", _lib.ptr(self.buf), nbytes)
        ZeroArena.live = self

    def begin_step(self):
        if self.dirty:
            self.buf[:self.dirty].zero_()
        self.off = 0
        self.dirty = 0

    def take(self, nbytes):
        off = (self.off + 255) & ~255
        if off + nbytes > self.buf.numel():
            return None
        self.off = off + nbytes
        self.dirty = max(self.dirty, self.off)
        return self.buf[off:off + nbytes]

    def close(self):
        with torch.cuda.device(self.buf.device):
            _lib.call("gcn_zero_arena_register", None, 0)
        if ZeroArena.live is self:
            ZeroArena.live = None


def zeroed_scratch(nbytes, device):
    """uint8 scratch for an accumulator the library zeroes: from the live ZeroArena (already zero, the library skips its
    fill) or a plain allocation (the library fills it)."""
    a = ZeroArena.live
    if a is not None and a.buf.device == torch.device(device):
        t = a.take(nbytes)
        if t is not None:
            return t
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def zeroed_like(shape, dtype, device):
    n = 1
    for d in shape:
        n *= d
    return zeroed_scratch(n * torch.empty((), dtype=dtype).element_size(), device).view(dtype).view(*shape)


def _acc_buffers(n_f64, C, device, tail_bytes=0):
    """(S (n_f64,) f64, dgamma (C,) f32, dbeta (C,) f32) carved back to back from ONE allocation, so that the library
    zeroes them with a single fill (csrc/common.h:zero_spans) instead of three 4.5-us launches.  tail_bytes > 0: a
    fourth return value, a uint8 scratch view that starts right behind dbeta (its zeroed header joins the same fill)."""
    head = n_f64 * 8 + 2 * C * 4
    raw = torch.empty(head + tail_bytes, dtype=torch.uint8, device=device) if tail_bytes else zeroed_scratch(head, device)
    S = raw[:n_f64 * 8].view(torch.float64)
    dg = raw[n_f64 * 8:n_f64 * 8 + C * 4].view(torch.float32)
    db = raw[n_f64 * 8 + C * 4:head].view(torch.float32)
    return (S, dg, db, raw[head:]) if tail_bytes else (S, dg, db)


class GroupNormReLUFunction(torch.autograd.Function):
    """y = [ReLU](GroupNorm(x)) for x (B,N,C) f32 or bf16 (output dtype = input dtype)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu, gsum=None):
        _lib.require_cuda(x)
        assert x.dim() == 3 and x.dtype in (torch.float32, torch.bfloat16)
        x = x.contiguous()
        B, N, C = x.shape
        dt = 1 if x.dtype == torch.bfloat16 else 0
        ga, be = gamma.float().contiguous(), beta.float().contiguous()
        y = torch.empty_like(x)
        mean_rstd = torch.empty(B, groups, 2, dtype=torch.float32, device=x.device)
        if gsum is not None:          # statistics came out of the producing GEMM's epilogue (csrc/gemm.hip)
            _run("gcn_gn_apply", x, _lib.ptr(x), dt, _lib.ptr(gsum), _lib.ptr(ga), _lib.ptr(be), B, N, C, groups, float(eps),
                 int(relu), _lib.ptr(y), _lib.ptr(mean_rstd))
        else:
            ws = zeroed_like((B, groups, 2), torch.float64, x.device)
            _run("gcn_gn_fwd", x, _lib.ptr(x), dt, _lib.ptr(ga), _lib.ptr(be), B, N, C, groups, float(eps), int(relu),
                 _lib.ptr(y), _lib.ptr(mean_rstd), _lib.ptr(ws))
        ctx.save_for_backward(x, ga, be, mean_rstd)
        ctx.cfg = (groups, relu, dt)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ga, be, mean_rstd = ctx.saved_tensors
        groups, relu, dt = ctx.cfg
        B, N, C = x.shape
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = torch.empty(_lib.lib().gcn_gn_bwd_ws_bytes(B, N, C, groups), dtype=torch.uint8, device=x.device)
        _run("gcn_gn_bwd", x, _lib.ptr(dy), _lib.ptr(x), dt, _lib.ptr(ga), _lib.ptr(be), _lib.ptr(mean_rstd), B, N, C,
             groups, int(relu), _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws))
        return dx, dgamma, dbeta, None, None, None, None


def group_norm_relu(x, gn, relu=True):
    """x (B,N,C) point-major; gn: an nn.GroupNorm holding (num_groups, weight, bias, eps)."""
    return GroupNormReLUFunction.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu)


def tall_skinny_tn(a, b, chunks=32, out_dtype=None):
    """a^T @ b for a (M,P), b (M,Q) with M >> P,Q (weight-gradient shape).  The library picks a tile
    config with a few dozen workgroups for the direct call (each looping over all M rows: 150-350 us at
    M=65536 on MI355X); splitting the long reduction into `chunks` batched GEMMs + a sum fills the chip.
    out_dtype: the partial sums are added up directly into that type (no separate cast kernel)."""
    M = a.shape[0]
    if M % chunks != 0 or M // chunks < 64:
        r = a.t() @ b
        return r if out_dtype is None else r.to(out_dtype)
    part = torch.bmm(a.view(chunks, M // chunks, -1).transpose(1, 2), b.view(chunks, M // chunks, -1))
    return part.sum(0, dtype=out_dtype)


class CastCache:
    """bf16 (autocast-dtype) copies of a model's floating-point parameters, refreshed by ONE multi-tensor copy per
    step instead of one cast kernel per layer and call (a training step had ~40 of them, 5 us each).  LinearPMFunction
    looks a weight up by identity and version; anything not registered (weight slices, ...) is cast as before."""
    _live = None

    def __init__(self, module, dtype=torch.bfloat16, pad_k=None):
        """pad_k: {parameter: Kpad} -- 1x1-conv weights whose GEMM input is zero-padded to Kpad columns."""
        self.params = [p for p in module.parameters() if p.dtype == torch.float32 and p.dim() >= 1]
        self.dtype = dtype
        self.version = {}
        self.by_id = {id(p): i for i, p in enumerate(self.params)}
        pad_k = {id(p): k for p, k in (pad_k or {}).items()}
        # 1x1-conv weights (Cout, Cin, 1[, 1]) live inside a zeroed (Cout rounded up to 32, Cin rounded up to 16 or
        # pad_k) bf16 image, the operand layout of csrc/gemm.hip; `copies` are views of it, so the one multi-tensor copy
        # per step fills the padded images as well
        self.padded, self.copies = {}, []
        for p in self.params:
            if p.dim() >= 2 and p[0].numel() == p.shape[1]:
                n, k = p.shape[0], p.shape[1]
                kp = max(pad_k.get(id(p), 0), (k + 15) // 16 * 16)
                img = torch.zeros((n + 31) // 32 * 32, kp, dtype=dtype, device=p.device)
                self.padded[id(p)] = img
                self.copies.append(img[:n, :k].view(p.shape) if (kp == k) else img[:n, :k].unflatten(1, p.shape[1:]))
            else:
                self.copies.append(torch.empty_like(p, dtype=dtype))
        self._segs = None

    def _segment_table(self, chunk=16384):
        """Device table for gcn_multi_cast_bf16: one record per `chunk` elements of every parameter."""
        recs = []
        for p, c in zip(self.params, self.copies):
            if not (p.is_cuda and p.is_contiguous() and self.dtype == torch.bfloat16):
                return None
            img = self.padded.get(id(p))
            cols = p[0].numel() if p.dim() >= 2 else p.numel()
            pitch = img.shape[1] if img is not None else cols
            n = p.numel()
            for first in range(0, n, chunk):
                recs.append([p.data_ptr(), c.data_ptr(), cols, pitch, first, min(chunk, n - first)])
        dev = self.params[0].device
        return torch.tensor(recs, dtype=torch.int64).to(dev), [p.data_ptr() for p in self.params]

    def refresh(self):
        """Call once per step after the optimizer update (or before the first forward)."""
        with torch.no_grad():
            if self._segs is None and self.params and self.params[0].is_cuda:
                self._segs = self._segment_table() or False
            if self._segs and self._segs[1] == [p.data_ptr() for p in self.params]:
                tab = self._segs[0]
                with _lib.on_device(tab):
                    _lib.call("gcn_multi_cast_bf16", _lib.ptr(tab), tab.shape[0], _lib.stream_of(tab))
            else:                                   # CPU tensors, another dtype, or storage that moved
                torch._foreach_copy_(self.copies, self.params)
        self.version = {id(p): p._version for p in self.params}
        CastCache._live = self

    @staticmethod
    def lookup(t, dtype):
        """A current low-precision copy of parameter (or flatten(1) view of a parameter) `t`, or None."""
        c = CastCache._live
        if c is None or c.dtype != dtype:
            return None
        base = t._base if t._base is not None else t
        i = c.by_id.get(id(base))
        if (i is None or c.version.get(id(base)) != base._version or base.numel() != t.numel() or not t.is_contiguous()
                or t.storage_offset() != base.storage_offset()):      # only whole-tensor reshapes of the parameter
            return None
        cp = c.copies[i]
        return cp.view(t.shape) if cp.is_contiguous() else cp.reshape(t.shape)

    @staticmethod
    def lookup_padded(t, kx):
        """The zero-padded (Np, kx) image of 1x1-conv weight `t` (a flatten(1) view of a registered parameter), or None."""
        c = CastCache._live
        if c is None:
            return None
        base = t._base if t._base is not None else t
        img = c.padded.get(id(base))
        if img is None or c.version.get(id(base)) != base._version or img.shape[1] != kx or base.numel() != t.numel():
            return None
        return img


def _own_gemm(M, N, K, fused_gn):
    """Policy: does csrc/gemm.hip serve this (M rows, N outputs, K inputs) layer?  (tools/gemm_bench.py, M = 65536)"""
    mode = os.environ.get("GCANET_GEMM", "auto")
    if mode == "lib" or K % 16 != 0 or M < 128:
        return False
    if mode == "own":
        return True
    lim = int(os.environ.get("GCANET_GEMM_FUSED_K", "256"))
    return N <= 128 or (fused_gn and N <= 256 and K <= lim)


def gemm_own(x2, wq, bias, N, out_f32=False, gn=None, rows_per_cloud=0):
    """out (M,N) = x2 (M,K) bf16 @ wq (Np,K)^T bf16 + bias through csrc/gemm.hip.  gn = number of groups: also returns
    the (M/rows_per_cloud, gn, 2) f64 sums / sums of squares of the f32 results (for gcn_gn_apply)."""
    M, K = x2.shape
    out = torch.empty(M, N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x2.device)
    gsum = ws = None
    if gn:
        gsum = torch.empty(M // rows_per_cloud, gn, 2, dtype=torch.float64, device=x2.device)
        ws = torch.empty(_lib.lib().gcn_gemm_stats_ws_bytes(M, N), dtype=torch.uint8, device=x2.device)
    with _lib.on_device(x2):
        _lib.call("gcn_gemm_bf16", _lib.ptr(x2), _lib.ptr(wq), _lib.ptr(bias), _lib.ptr(out), int(out_f32), M, N, wq.shape[0], K,
                  _lib.ptr(gsum), _lib.ptr(ws), rows_per_cloud if gn else 0, gn or 0, _lib.stream_of(x2))
    return out, gsum


def _own_wgrad(dy2, x2):
    """Does csrc/gemm.hip's weight-gradient kernel serve dy2 (M,N)^T @ x2 (M,K)?  (bf16 rows, 16-byte aligned)"""
    if os.environ.get("GCANET_GEMM", "auto") == "lib":
        return False
    return (dy2.is_cuda and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and dy2.shape[1] % 8 == 0
            and x2.shape[1] % 8 == 0 and dy2.shape[0] >= 512 and dy2.is_contiguous() and x2.is_contiguous())


def _wgrad_own(dy2, x2, with_bias):
    """(dW (N,K) f32, db (N,) f32 | None) = (dy2^T @ x2, column sums of dy2) in ONE pass over the rows (csrc/gemm.hip):
    the row slices' partial tiles are folded in a fixed order by the same call, the bias gradient rides along."""
    M, N = dy2.shape
    K = x2.shape[1]
    raw = torch.empty(N * K + (N if with_bias else 0), dtype=torch.float32, device=dy2.device)
    dw = raw[:N * K].view(N, K)
    db = raw[N * K:] if with_bias else None
    # partial tiles of the row slices: scratch per call (free with the caching allocator; inside a HIP-graph capture it
    # belongs to the graph's pool -- a process-global buffer that a later, larger call replaces leaves replays a stale address)
    ws = torch.empty(_lib.lib().gcn_gemm_wgrad_ws_bytes(M, N, K), dtype=torch.uint8, device=dy2.device)
    with _lib.on_device(dy2):
        _lib.call("gcn_gemm_wgrad_bf16", _lib.ptr(dy2), _lib.ptr(x2), M, N, K, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws),
                  _lib.stream_of(dy2))
    return dw, db


def _wgrad_narrow(dy2, x2, with_bias):
    """(dW (N,K) f32, db (N,) f32 | None) for a layer with at most 32 outputs -- or, operands swapped, at most 32 inputs --
    in one streaming pass (csrc/gemm.hip: gcn_wgrad_narrow); None when the shape is not served."""
    if os.environ.get("GCANET_GEMM", "auto") == "lib" or os.environ.get("GCANET_NARROW", "1") == "0" \
            or not (dy2.is_cuda and dy2.is_contiguous() and x2.is_contiguous()):
        return None
    ok = (torch.float32, torch.bfloat16)
    if dy2.dtype not in ok or x2.dtype not in ok:
        return None
    M, N = dy2.shape
    K = x2.shape[1]
    L = _lib.lib()
    swap = False
    # where it wins (tools/wgrad_narrow_bench.py, M = 65536): 3x256 16 us vs 55 (library: split-K bmm + partial sum +
    # column sum), 30x30 21 vs 34-45, 3x128 f32 11 vs 52, 10x256 25 vs 35; it is a VALU kernel with 4 N accumulators per
    # lane -- beyond 16 outputs only two waves fit a SIMD and 22x256 takes 48 us against the library's 35 (split-K bmm +
    # partial-sum pass + column sum).  It is served here all the same since round 3: the library path's two reductions
    # zero their semaphores with hipMemsetAsync, i.e. put memset NODES into the captured step (csrc/common.h: fill_dev
    # says why this library issues none), and it is three launches against one.
    small = lambda n, k: n <= 4 or n * k <= 1024 or (n <= 32 and k <= 256)
    if not (N <= 32 and small(N, K) and L.gcn_wgrad_narrow_supported(M, N, K)):
        if with_bias or not (K <= 32 and small(K, N) and L.gcn_wgrad_narrow_supported(M, K, N)):
            return None
        swap = True                                    # narrow INPUT: dW^T = X^T . dY from the same kernel
        dy2, x2, N, K = x2, dy2, K, N
    raw = torch.empty(N * K + (N if with_bias else 0), dtype=torch.float32, device=dy2.device)
    dw = raw[:N * K].view(N, K)
    db = raw[N * K:] if with_bias else None
    ws = torch.empty(L.gcn_wgrad_narrow_ws_bytes(M, N, K), dtype=torch.uint8, device=dy2.device)
    with _lib.on_device(dy2):
        _lib.call("gcn_wgrad_narrow", _lib.ptr(dy2), int(dy2.dtype == torch.bfloat16), _lib.ptr(x2),
                  int(x2.dtype == torch.bfloat16), M, N, K, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws), _lib.stream_of(dy2))
    return (dw.t() if swap else dw), db


class LinearPMFunction(torch.autograd.Function):
    """y = x @ W^T + b on point-major rows with a split-K weight gradient (see tall_skinny_tn).
    Runs in the autocast dtype (bf16 under torch.autocast, as the plain F.linear would).  With gn_groups > 0 the
    second output is the GroupNorm statistics of y (csrc/gemm.hip epilogue) for group_norm_relu(..., gsum=)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gn_groups=0):
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
        N, K = weight.shape
        Kx = x.shape[-1]                  # may exceed K: trailing zero columns the caller added to reach K % 16 == 0
        M = x.numel() // Kx
        rows = x.shape[-2] if x.dim() >= 2 else M
        fused = gn_groups > 0 and (N // max(gn_groups, 1)) % 32 == 0 and rows % 128 == 0
        ctx.own = bool(x.is_cuda and dt == torch.bfloat16 and _own_gemm(M, N, Kx, fused))
        if ctx.own:
            xc = x.to(dt).contiguous()
            wq = CastCache.lookup_padded(weight, Kx)
            if wq is None:                # not registered (or stale): pack here
                wq = torch.zeros((N + 31) // 32 * 32, Kx, dtype=dt, device=x.device)
                wq[:N, :K] = weight
            y, gsum = gemm_own(xc.view(M, Kx), wq, None if bias is None else bias.float(), N, gn=gn_groups if fused else None,
                               rows_per_cloud=rows)
            ctx.save_for_backward(xc, wq)
            ctx.has_bias, ctx.in_dtypes, ctx.nk = bias is not None, (x.dtype, weight.dtype), (N, K)
            ctx.mark_non_differentiable(*([gsum] if gsum is not None else []))
            y = y.view(*x.shape[:-1], N)
            return (y, gsum) if gn_groups > 0 else y
        if Kx != K:
            x = x[..., :K]
        xc = x.to(dt)
        wc = (CastCache.lookup(weight, dt) if weight.dtype != dt else None)
        wc = weight.to(dt) if wc is None else wc
        bc = None
        if bias is not None:
            bc = CastCache.lookup(bias, dt) if bias.dtype != dt else None
            bc = bias.to(dt) if bc is None else bc
        ctx.save_for_backward(xc, wc)
        ctx.has_bias = bias is not None
        ctx.in_dtypes = (x.dtype, weight.dtype)
        ctx.kx = Kx
        with torch.autocast("cuda", enabled=False):
            y = torch.nn.functional.linear(xc, wc, bc)
        return (y, None) if gn_groups > 0 else y

    @staticmethod
    def backward(ctx, dy, _gsum=None):
        dx, dw, db = LinearPMFunction.backward_parts(ctx, dy)
        return dx, dw, db, None

    @staticmethod
    def backward_parts(ctx, dy, parts=None):
        """(dx, dW, db).  parts = [(first column, width, dtype) | None, ...]: dx becomes a LIST with one contiguous
        gradient per listed column block of the input (None entries stay None) -- CatLinearPMFunction."""
        xc, wc = ctx.saved_tensors
        dy = dy.to(xc.dtype).contiguous()
        rows = dy.reshape(-1, dy.shape[-1])

        def dgrad(wm, full_dtype):
            if parts is None:
                return (dy @ wm).to(full_dtype)
            return [None if pt is None else (dy @ wm[:, pt[0]:pt[0] + pt[1]]).to(pt[2]) for pt in parts]

        with torch.autocast("cuda", enabled=False):
            db = None
            if ctx.own:
                N, K = ctx.nk
                Kx = xc.shape[-1]
                dx = dgrad(wc[:N], ctx.in_dtypes[0])                      # (.., Kx); the zero-weight padding columns get 0
                nar = None if N % 8 == 0 else _wgrad_narrow(rows, xc.reshape(-1, Kx), ctx.has_bias)
                if N % 8 == 0:                                            # csrc/gemm.hip: transpose-read weight gradient
                    dwf, db = _wgrad_own(rows, xc.reshape(-1, Kx), ctx.has_bias)
                    dw = dwf[:, :K].to(ctx.in_dtypes[1])
                elif nar is not None:                                     # narrow outputs (10, 22, 3, 30): one streaming pass
                    dw, db = nar[0][:, :K].to(ctx.in_dtypes[1]), nar[1]
                else:
                    dw = tall_skinny_tn(rows, xc.reshape(-1, Kx), out_dtype=ctx.in_dtypes[1])[:, :K]
            else:
                dx = dgrad(wc, ctx.in_dtypes[0])
                if parts is None and ctx.kx != wc.shape[1]:
                    dx = torch.nn.functional.pad(dx, (0, ctx.kx - wc.shape[1]))
                x2 = xc.reshape(-1, xc.shape[-1])
                nar = None
                if _own_wgrad(rows, x2):
                    dwf, db = _wgrad_own(rows, x2, ctx.has_bias)
                    dw = dwf.to(ctx.in_dtypes[1])
                else:
                    nar = _wgrad_narrow(rows, x2, ctx.has_bias)
                    if nar is not None:
                        dw, db = nar[0].to(ctx.in_dtypes[1]), nar[1]
                    else:
                        dw = tall_skinny_tn(rows, x2, out_dtype=ctx.in_dtypes[1])
            if ctx.has_bias and db is None:
                db = rows.sum(0, dtype=torch.float32)                     # f32 accumulation, no f32 copy of dy
        return dx, dw, db


def linear_pm(x, weight, bias=None):
    return LinearPMFunction.apply(x, weight, bias)


class CatLinearPMFunction(torch.autograd.Function):
    """LinearPMFunction on cat(parts, dim=-1) that owns the concatenation: the input gradient is computed per part,
    dy @ W[:, columns of the part], into CONTIGUOUS tensors and only for the parts that need one.  Autograd's own cat
    backward hands out column slices of one wide gradient: every consumer downstream then adds strided rows (22 us per
    (65536, 256) bf16 add against 11 contiguous), and the columns of parts without a gradient (xyz, zero padding) are
    computed for nothing -- 39 us for the 272-wide input of conv3, of which 256 columns are wanted."""

    @staticmethod
    def forward(ctx, weight, bias, gn_groups, *parts):
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else parts[0].dtype
        x = torch.cat([p.to(dt) for p in parts], dim=-1)
        out = LinearPMFunction.forward(ctx, x, weight, bias, gn_groups)
        off, ctx.parts = 0, []
        for i, p in enumerate(parts):
            ctx.parts.append((off, p.shape[-1], p.dtype) if ctx.needs_input_grad[3 + i] else None)
            off += p.shape[-1]
        return out

    @staticmethod
    def backward(ctx, dy, _gsum=None):
        dxs, dw, db = LinearPMFunction.backward_parts(ctx, dy, ctx.parts)
        return (dw, db, None, *dxs)


def cat_conv1x1_gn_relu(parts, conv, gn, relu=True):
    """conv1x1_gn_relu(torch.cat(parts, dim=-1), conv, gn) with per-part input gradients (CatLinearPMFunction)."""
    y, gsum = CatLinearPMFunction.apply(conv.weight.flatten(1), conv.bias, gn.num_groups, *parts)
    return GroupNormReLUFunction.apply(y, gn.weight, gn.bias, gn.num_groups, gn.eps, relu, gsum)


_ONES = {}


class AddRowBroadcastFunction(torch.autograd.Function):
    """a (B,N,C) + b (B,C) broadcast over the N rows.  Backward: the gradient of b is the column sum of the incoming
    gradient, taken as a (1 x N) . (N x C) product per cloud on the matrix cores -- torch's own reduction for this shape
    runs in two stages whose semaphores are zeroed by hipMemsetAsync: a memset NODE in the captured step (see
    csrc/common.h: fill_dev for why the step holds none)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.b_dtype = b.dtype
        return a + b.to(a.dtype).unsqueeze(1)

    @staticmethod
    def backward(ctx, g):
        B, N, _ = g.shape
        key = (B, N, g.dtype, g.device)
        if key not in _ONES:
            _ONES[key] = torch.ones(B, 1, N, dtype=g.dtype, device=g.device)
        gb = torch.bmm(_ONES[key], g.contiguous()).squeeze(1)
        return g, gb.to(ctx.b_dtype)


def add_row_broadcast(a, b):
    return AddRowBroadcastFunction.apply(a, b)


def conv1x1(x, conv):
    """Conv1d(kernel 1) applied to point-major x (B,N,Cin) as a GEMM with the SAME parameter tensor."""
    return linear_pm(x, conv.weight.flatten(1), conv.bias)      # (Cout,Cin,1) -> (Cout,Cin) view


def conv1x1_gn_relu(x, conv, gn, relu=True):
    """group_norm_relu(conv1x1(x, conv), gn) with the GroupNorm statistics taken from the GEMM's epilogue where
    csrc/gemm.hip serves the layer (no separate statistics pass over the conv output)."""
    y, gsum = LinearPMFunction.apply(x, conv.weight.flatten(1), conv.bias, gn.num_groups)
    return GroupNormReLUFunction.apply(y, gn.weight, gn.bias, gn.num_groups, gn.eps, relu, gsum)


class GlobalMaxPoolFunction(torch.autograd.Function):
    """x (B,N,C) -> max over points (B,C)  (M4:513 `x.max(dim=2)` on the channel-major tensor).  Backward routes the
    gradient to the arg-max row only: one zero fill + a (B,C)-element scatter instead of torch's dense
    `grad * (x == max) / count` passes over (B,N,C)."""

    @staticmethod
    def forward(ctx, x):
        vals, arg = x.max(dim=1)
        ctx.save_for_backward(arg)
        ctx.shape = x.shape
        return vals

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        g = torch.zeros(ctx.shape, dtype=dout.dtype, device=dout.device)
        g.scatter_(1, arg.unsqueeze(1), dout.unsqueeze(1))
        return g


def global_max_pool(x):
    return GlobalMaxPoolFunction.apply(x)


class GroupNormReLUMaxFunction(torch.autograd.Function):
    """max over points of [ReLU](GroupNorm(x)) for x (B,N,C) -> (B,C), the (B,N,C) activation never written
    (csrc/gn.hip: gn_apply_max_kernel).  Backward: the gradient reaches one row per (sample, channel), so the GroupNorm
    input gradient is a per-(sample, group) affine map of x plus B*C corrections (gcn_gn_max_bwd) -- no dense dy."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        _lib.require_cuda(x)
        assert x.dim() == 3 and x.dtype in (torch.float32, torch.bfloat16)
        x = x.contiguous()
        B, N, C = x.shape
        dt = 1 if x.dtype == torch.bfloat16 else 0
        ga, be = gamma.float().contiguous(), beta.float().contiguous()
        vals = torch.empty(B, C, dtype=torch.float32, device=x.device)
        arg = torch.empty(B, C, dtype=torch.int64, device=x.device)
        mean_rstd = torch.empty(B, groups, 2, dtype=torch.float32, device=x.device)
        ws = zeroed_like((B, groups, 2), torch.float64, x.device)
        best = zeroed_like((B, C), torch.int64, x.device)
        _run("gcn_gn_max_fwd", x, _lib.ptr(x), dt, _lib.ptr(ga), _lib.ptr(be), B, N, C, groups, float(eps), int(relu),
             _lib.ptr(vals), _lib.ptr(arg), _lib.ptr(mean_rstd), _lib.ptr(ws), _lib.ptr(best))
        ctx.save_for_backward(x, ga, be, mean_rstd, arg)
        ctx.cfg = (groups, relu, dt)
        return vals.to(x.dtype)

    @staticmethod
    def backward(ctx, dout):
        x, ga, be, mean_rstd, arg = ctx.saved_tensors
        groups, relu, dt = ctx.cfg
        B, N, C = x.shape
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = torch.empty(_lib.lib().gcn_gn_max_bwd_ws_floats(B, C, groups), dtype=torch.float32, device=x.device)
        _run("gcn_gn_max_bwd", x, _lib.ptr(x), dt, _lib.ptr(ga), _lib.ptr(be), _lib.ptr(mean_rstd),
             _lib.ptr(dout.float().contiguous()), _lib.ptr(arg), B, N, C, groups, int(relu), _lib.ptr(dx), _lib.ptr(dgamma),
             _lib.ptr(dbeta), _lib.ptr(ws))
        return dx, dgamma, dbeta, None, None, None


def group_norm_relu_max(x, gn, relu=True):
    """max over dim 1 of group_norm_relu(x, gn): (B,N,C) -> (B,C)."""
    return GroupNormReLUMaxFunction.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu)


class ParamNormaliseFunction(torch.autograd.Function):
    """(..., 22) parameter rows: unit-normalise the direction triples 4:7, 8:11, 15:18 (M4:664-676)."""

    @staticmethod
    def forward(ctx, p):
        _lib.require_cuda(p)
        assert p.shape[-1] == 22
        pc = p.float().contiguous()
        out = torch.empty_like(pc)
        _run("gcn_param_normalise_fwd", pc, _lib.ptr(pc), pc.numel() // 22, _lib.ptr(out))
        ctx.save_for_backward(pc)
        ctx.in_dtype = p.dtype
        return out

    @staticmethod
    def backward(ctx, go):
        (pc,) = ctx.saved_tensors
        go = go.float().contiguous()
        gi = torch.empty_like(pc)
        _run("gcn_param_normalise_bwd", pc, _lib.ptr(pc), _lib.ptr(go), pc.numel() // 22, _lib.ptr(gi))
        return gi.to(ctx.in_dtype)


def param_normalise(p):
    return ParamNormaliseFunction.apply(p)


class RowNormaliseFunction(torch.autograd.Function):
    """x / x.norm(dim=-1, keepdim=True) on (..., C) rows, one kernel each way (csrc/heads.hip)."""

    @staticmethod
    def forward(ctx, x):
        _lib.require_cuda(x)
        xc = x.float().contiguous()
        y = torch.empty_like(xc)
        _run("gcn_row_normalise_fwd", xc, _lib.ptr(xc), xc.numel() // xc.shape[-1], xc.shape[-1], _lib.ptr(y))
        ctx.save_for_backward(xc)
        ctx.in_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, go):
        (xc,) = ctx.saved_tensors
        go = go.float().contiguous()
        gi = torch.empty_like(xc)
        _run("gcn_row_normalise_bwd", xc, _lib.ptr(xc), _lib.ptr(go), xc.numel() // xc.shape[-1], xc.shape[-1], _lib.ptr(gi))
        return gi.to(ctx.in_dtype)


def row_normalise(x):
    return RowNormaliseFunction.apply(x)
