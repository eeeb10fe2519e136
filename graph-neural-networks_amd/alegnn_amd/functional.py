"""Autograd boundary of the HIP path: ``LSIGF`` with the reference's signature (alegnn/utils/graphML.py:83).

    y = LSIGF(h, S, x, b)      h [F,E,K,G], S [E,N,N] (dense tensor, or SparseGSO / scipy sparse), x [B,G,N], b [F,1]|[F,N]|None

Forward and backward are two C-ABI calls (gf_lsigf_forward / gf_lsigf_backward, include/gfhip.h) on the current
torch HIP stream.  torch is used for device memory, streams and autograd bookkeeping only -- the arithmetic is in
libgfhip.so.  S gets no gradient (it is not a Parameter in the reference either, graphML.py:2099).
"""
from __future__ import annotations

import os

import torch

from . import _lib
from .gso import EdgePattern, SparseGSO


def _ptr(t):
    return None if t is None else t.data_ptr()


def _require_f32_cuda(name, t):
    if t.device.type != "cuda":
        raise RuntimeError(f"alegnn_amd.LSIGF: `{name}` is on '{t.device}'. This implementation runs on MI355X (HIP) only "
                           "and has no CPU fallback; move the module and data to 'cuda'.")
    if t.dtype != torch.float32:
        raise TypeError(f"alegnn_amd.LSIGF: `{name}` has dtype {t.dtype}; the gfx950 kernels compute in float32 "
                        "(cast the module / data with .float()).")


_PAD_WIDTHS = os.environ.get("GFHIP_PAD_WIDTHS", "1") != "0"


def _padded_width(w):
    return w if (not _PAD_WIDTHS or w % 8 == 0 or w > 128) else 8 * ((w + 7) // 8)


class _LSIGFFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h, bias, gso: SparseGSO, relu=False):
        L = _lib.lib()
        B, G, Nin = x.shape
        F_, E, K, G2 = h.shape
        N = gso.N
        T = 1 + E * (K - 1)
        x = x.contiguous()
        h = h.contiguous()
        bias_c = None if bias is None else bias.contiguous()
        with torch.cuda.device(x.device):
            plans = gso.plans(x.device)
            Z = torch.empty((T, B, N, G), dtype=torch.float32, device=x.device)
            y = torch.empty((B, F_, Nin), dtype=torch.float32, device=x.device)
            stream = torch.cuda.current_stream().cuda_stream
            fwd = L.gf_lsigf_forward_relu if relu else L.gf_lsigf_forward
            _lib.check(fwd(plans, E, x.data_ptr(), h.data_ptr(), _ptr(bias_c), Z.data_ptr(), y.data_ptr(),
                           B, G, F_, K, Nin, stream), "gf_lsigf_forward")
        ctx.gso = gso
        ctx.dims = (B, G, F_, E, K, Nin, N, T)
        ctx.has_bias = bias is not None
        need_taps = ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2])  # dh / db read Z
        ctx.relu = bool(relu)
        ctx.save_for_backward(h, Z if need_taps else None, y if relu else None)   # the ReLU mask is (y > 0)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        h, Z, y_act = ctx.saved_tensors
        B, G, F_, E, K, Nin, N, T = ctx.dims
        need_dx, need_dh, need_db = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dy = dy.contiguous()
        dev = dy.device
        with torch.cuda.device(dev):
            plans = ctx.gso.plans(dev)
            P = torch.empty((T if need_dx else 1, B, N, F_), dtype=torch.float32, device=dev)
            dx = torch.empty((B, G, Nin), dtype=torch.float32, device=dev) if need_dx else None
            dh = torch.empty_like(h) if need_dh else None
            db = torch.empty((F_, 1), dtype=torch.float32, device=dev) if need_db else None
            ws = None
            ws_bytes = 0
            if need_dh or need_db:
                ws_bytes = L.gf_grad_taps_workspace_bytes(B, N, G, F_, E, K)
                ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream().cuda_stream
            if ctx.relu:
                _lib.check(L.gf_lsigf_backward_relu(plans, E, dy.data_ptr(), y_act.data_ptr(), _ptr(Z), h.data_ptr(), P.data_ptr(),
                                                    _ptr(dx), _ptr(dh), _ptr(db), _ptr(ws), ws_bytes, B, G, F_, K, Nin, stream),
                           "gf_lsigf_backward_relu")
            else:
                _lib.check(L.gf_lsigf_backward(plans, E, dy.data_ptr(), _ptr(Z), h.data_ptr(), P.data_ptr(), _ptr(dx), _ptr(dh),
                                               _ptr(db), _ptr(ws), ws_bytes, B, G, F_, K, Nin, stream), "gf_lsigf_backward")
        return dx, dh, db, None, None


def LSIGF(h, S, x, b=None, activation=None):
    """Linear shift-invariant graph filter, reference signature and semantics (graphML.py:83-176):

        y[b,f,n] = sum_{e,k,g} h[f,e,k,g] * (x_g S_e^k)[b,n] + b[f]

    ``S`` may be the reference's dense ``[E,N,N]`` tensor or anything ``SparseGSO.from_any`` accepts.
    ``x`` may have fewer nodes than S (Nin < N): it is zero-padded and the output keeps the first Nin nodes,
    which is what GraphFilter.forward does around its LSIGF call (graphML.py:2131-2143).
    ``activation='relu'`` (superset) fuses the ReLU that follows the filter in SelectionGNN (architectures.py:286-289) into the
    contraction's epilogue and its mask into the backward pass; it needs the per-feature bias form ``b [F,1]`` (or None).
    """
    gso = SparseGSO.from_any(S)
    assert h.dim() == 4 and x.dim() == 3
    F_, E, K, G = h.shape
    assert gso.E == E                                  # graphML.py:135
    assert x.shape[1] == G                             # graphML.py:139
    assert x.shape[2] <= gso.N                         # graphML.py:140 (== N); < N only via GraphFilter padding
    _require_f32_cuda("x", x)
    _require_f32_cuda("h", h)
    fused_bias = None
    late_bias = None
    if b is not None:
        _require_f32_cuda("b", b)
        assert b.dim() == 2 and b.shape[0] == F_
        if b.shape[1] == 1:
            fused_bias = b
        else:                                          # per-node bias [F,N] (graphML.py:110-112): broadcast add, plumbing
            late_bias = b
    assert activation in (None, "relu")
    relu = activation == "relu"
    # Feature counts that are not multiples of 8 (the 1-feature input of every example architecture, a 5-class output) are
    # zero-padded to the next multiple of 8: the padded channels carry zeros through the filter (zero taps, zero signals) and
    # are cut off again, autograd of pad / slice returns the exact gradients -- and the layer runs on the MFMA / 16-byte paths
    # instead of the scalar generic kernels (config-3 first layer, G = 1: 0.51 -> 0.2 ms).  GFHIP_PAD_WIDTHS=0 disables it.
    Gp, Fp = _padded_width(G), _padded_width(F_)
    if Gp != G or Fp != F_:
        if Gp != G:
            x = torch.nn.functional.pad(x, (0, 0, 0, Gp - G))
        h = torch.nn.functional.pad(h, (0, Gp - G, 0, 0, 0, 0, 0, Fp - F_))
        if fused_bias is not None and Fp != F_:
            fused_bias = torch.nn.functional.pad(fused_bias, (0, 0, 0, Fp - F_))
    if relu and late_bias is not None:                         # the activation must see the bias: apply both outside
        y = _LSIGFFunction.apply(x, h, None, gso, False)
        return torch.relu((y[:, :F_] if Fp != F_ else y) + late_bias[:, : x.shape[2]])
    y = _LSIGFFunction.apply(x, h, fused_bias, gso, relu)
    if Fp != F_:
        y = y[:, :F_].contiguous()
    if late_bias is not None:
        y = y + late_bias[:, : y.shape[2]]
    return y


_HANDOVER = os.environ.get("GFHIP_LAYER_HANDOVER", "1") != "0"


class _LSIGFChainFunction(torch.autograd.Function):
    """Consecutive ReLU graph-filter layers on ONE graph with the signals handed over in the internal layout -- column panels, or
    node-major rows on graphs beyond the LDS panel limit -- (gf_lsigf_forward_ex / gf_lsigf_backward_ex): layer l's contraction writes relu(y_l) straight into tap 0 of layer l+1's stack and,
    in the backward, layer l+1 writes its dx -- masked by relu'(y_l) -- straight into tap 0 of layer l's adjoint stack.  Per inner
    boundary one reference-layout tensor, one pack pass forward and one backward disappear (the reference permutes at every layer,
    graphML.py:170-171; SelectionGNN strings the blocks together, architectures.py:286-294).  Arithmetic and summation orders are
    those of the separate layers: outputs and gradients are bitwise the same."""

    @staticmethod
    def forward(ctx, x, gso, relu_last, grad_mode, *params):
        L = _lib.lib()
        nl = len(params) // 2
        hs = [params[2 * l].contiguous() for l in range(nl)]
        bs = [None if params[2 * l + 1] is None else params[2 * l + 1].contiguous() for l in range(nl)]
        B, G0, N = x.shape
        E = hs[0].shape[1]
        x = x.contiguous()
        dev = x.device
        # The backward needs every layer's tap stack; a forward nobody differentiates (evaluation under no_grad, frozen inputs and
        # parameters) needs only two at a time: layer l's and the one layer l writes its output into.  needs_input_grad mirrors
        # requires_grad whatever the grad mode, and inside forward() grad mode is always off: the caller samples it (grad_mode).
        keep = bool(grad_mode) and any(ctx.needs_input_grad)
        stacks = [None] * nl

        def stack_of(l):
            if stacks[l] is None:
                _, _, K, G = hs[l].shape
                stacks[l] = torch.empty((1 + E * (K - 1), B * G // 4, N, 4), dtype=torch.float32, device=dev)
            return stacks[l]

        with torch.cuda.device(dev):
            plans = gso.plans(dev)
            st = torch.cuda.current_stream().cuda_stream
            y = torch.empty((B, hs[-1].shape[0], N), dtype=torch.float32, device=dev)
            for l in range(nl):
                F_, _, K, G = hs[l].shape
                last = l == nl - 1
                flags = (1 if (not last or relu_last) else 0) | (2 if l > 0 else 0) | (0 if last else 4)
                out = y if last else stack_of(l + 1)
                _lib.check(L.gf_lsigf_forward_ex(plans, E, x.data_ptr() if l == 0 else None, hs[l].data_ptr(), _ptr(bs[l]), stack_of(l).data_ptr(),
                                                 out.data_ptr(), B, G, F_, K, N, flags, st), "gf_lsigf_forward_ex")
                if not keep:
                    stacks[l] = None                   # (stream-ordered allocator: the next layer's stack may reuse the block)
        ctx.gso, ctx.nl, ctx.relu_last, ctx.B, ctx.N, ctx.E = gso, nl, bool(relu_last), B, N, E
        ctx.has_bias = [b is not None for b in bs]
        if keep:
            ctx.save_for_backward(y if relu_last else None, *hs, *stacks)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        nl, B, N, E = ctx.nl, ctx.B, ctx.N, ctx.E
        saved = ctx.saved_tensors
        y_act, hs, stacks = saved[0], saved[1:1 + nl], saved[1 + nl:1 + 2 * nl]
        dy = dy.contiguous()
        dev = dy.device
        grads = [None] * (2 * nl)
        dx = None
        with torch.cuda.device(dev):
            plans = ctx.gso.plans(dev)
            st = torch.cuda.current_stream().cuda_stream
            Ps = [torch.empty((1 + E * (hs[l].shape[2] - 1), B * hs[l].shape[0] // 4, N, 4), dtype=torch.float32, device=dev) for l in range(nl)]
            need_x = ctx.needs_input_grad[0]
            for l in range(nl - 1, -1, -1):
                F_, _, K, G = hs[l].shape
                last, first = l == nl - 1, l == 0
                need_dh = ctx.needs_input_grad[4 + 2 * l]                 # (inputs: x, gso, relu_last, grad_mode, then h_l, b_l per layer)
                need_db = ctx.has_bias[l] and ctx.needs_input_grad[5 + 2 * l]
                dh = torch.empty_like(hs[l]) if need_dh else None
                db = torch.empty((F_, 1), dtype=torch.float32, device=dev) if need_db else None
                ws, ws_bytes = None, 0
                if need_dh or need_db:
                    ws_bytes = L.gf_grad_taps_workspace_bytes(B, N, G, F_, E, K)
                    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
                if first:
                    dx = torch.empty((B, G, N), dtype=torch.float32, device=dev) if need_x else None
                    out, mask, oflag = dx, None, 0
                else:                                   # hand the gradient to the layer below, masked by ITS ReLU: relu(y_{l-1}) is tap 0 here
                    out, mask, oflag = Ps[l - 1], stacks[l], 4
                flags = (0 if last else 2) | oflag
                _lib.check(L.gf_lsigf_backward_ex(plans, E, dy.data_ptr() if last else None, _ptr(y_act) if last else None, stacks[l].data_ptr(),
                                                  hs[l].data_ptr(), Ps[l].data_ptr(), _ptr(out), _ptr(dh), _ptr(db), _ptr(ws), ws_bytes, B, G, F_, K, N,
                                                  flags, _ptr(mask), st), "gf_lsigf_backward_ex")
                grads[2 * l], grads[2 * l + 1] = dh, db
        return (dx, None, None, None, *grads)


def lsigf_chain_supported(gso, x, layers):
    """True when `layers` = [(h, b, relu), ...] (consecutive GraphFilter layers on `gso`) can hand their signals over in the internal
    layout: every layer on the SAME pipeline (column panels, or node-major rows with widths that are multiples of 4) with per-feature
    bias, ReLU after every layer but possibly the last, Nin == N."""
    if not _HANDOVER or len(layers) < 2 or x.dim() != 3 or x.device.type != "cuda" or x.shape[2] != gso.N:
        return False
    L = _lib.lib()
    plans = gso.plans(x.device)
    pipes = set()
    for l, (h, b, relu) in enumerate(layers):
        F_, E, K, G = h.shape
        if E != gso.E or (b is not None and (b.dim() != 2 or b.shape[1] != 1)) or (l < len(layers) - 1 and not relu):
            return False
        if l > 0 and layers[l - 1][0].shape[0] != G:
            return False
        Gp, Fp = _padded_width(G), _padded_width(F_)
        pipe = L.gf_lsigf_pipeline(plans, E, Gp, Fp, K)
        if pipe not in (1, 2) or (pipe == 1 and (Gp % 4 or Fp % 4)):
            return False
        pipes.add(pipe)
    return len(pipes) == 1


def LSIGF_chain(layers, S, x):
    """relu(LSIGF(h_L, S, ... relu(LSIGF(h_1, S, x, b_1)) ..., b_L)) for consecutive filter layers on one graph, with the intermediate
    signals kept in the internal layout (see _LSIGFChainFunction).  layers = [(h [F,E,K,G], b [F,1]|None, relu: bool), ...]; the caller
    checks ``lsigf_chain_supported`` first.  Feature counts that are not multiples of 8 are zero-padded exactly as LSIGF does."""
    gso = SparseGSO.from_any(S)
    _require_f32_cuda("x", x)
    params = []
    G0 = layers[0][0].shape[3]
    Gp = _padded_width(G0)
    if Gp != G0:
        x = torch.nn.functional.pad(x, (0, 0, 0, Gp - G0))
    for (h, b, _) in layers:
        _require_f32_cuda("h", h)
        F_, E, K, G = h.shape
        Gp, Fp = _padded_width(G), _padded_width(F_)
        if Gp != G or Fp != F_:
            h = torch.nn.functional.pad(h, (0, Gp - G, 0, 0, 0, 0, 0, Fp - F_))
            if b is not None and Fp != F_:
                b = torch.nn.functional.pad(b, (0, 0, 0, Fp - F_))
        params += [h, b]
    y = _LSIGFChainFunction.apply(x, gso, bool(layers[-1][2]), torch.is_grad_enabled(), *params)
    F_last = layers[-1][0].shape[0]
    return y[:, :F_last].contiguous() if y.shape[1] != F_last else y


class _NVGFFunction(torch.autograd.Function):
    """Node-variant filter: gf_nvgf_forward / gf_nvgf_backward.  h is the expanded bank [F,E,K,G,N]; the bias is added by the
    kernel when it is per-feature ([F,1]), its gradient is a plain reduction of dy (left to autograd of the caller's add when
    the bias is per-node)."""

    @staticmethod
    def forward(ctx, x, h, bias, gso: SparseGSO):
        L = _lib.lib()
        B, G, Nin = x.shape
        F_, E, K, G2, N = h.shape
        T = 1 + E * (K - 1)
        x = x.contiguous()
        h = h.contiguous()
        bias_c = None if bias is None else bias.contiguous()
        with torch.cuda.device(x.device):
            plans = gso.plans(x.device)
            Z = torch.empty((T, B, N, G), dtype=torch.float32, device=x.device)
            y = torch.empty((B, F_, Nin), dtype=torch.float32, device=x.device)
            n = L.gf_nvgf_scratch_floats(B, N, G, F_, E, K, 0)
            scratch = torch.empty(n, dtype=torch.float32, device=x.device)
            _lib.check(L.gf_nvgf_forward(plans, E, x.data_ptr(), h.data_ptr(), _ptr(bias_c), Z.data_ptr(), y.data_ptr(),
                                         scratch.data_ptr(), n, B, G, F_, K, Nin, torch.cuda.current_stream().cuda_stream),
                       "gf_nvgf_forward")
        ctx.gso = gso
        ctx.dims = (B, G, F_, E, K, Nin, N)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(h, Z if ctx.needs_input_grad[1] else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        h, Z = ctx.saved_tensors
        B, G, F_, E, K, Nin, N = ctx.dims
        need_dx, need_dh = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dy = dy.contiguous()
        dev = dy.device
        dx = dh = None
        if need_dx or need_dh:
            with torch.cuda.device(dev):
                plans = ctx.gso.plans(dev)
                dx = torch.empty((B, G, Nin), dtype=torch.float32, device=dev) if need_dx else None
                dh = torch.empty_like(h) if need_dh else None
                n = L.gf_nvgf_scratch_floats(B, N, G, F_, E, K, 1)
                scratch = torch.empty(n, dtype=torch.float32, device=dev)
                _lib.check(L.gf_nvgf_backward(plans, E, dy.data_ptr(), _ptr(Z) if need_dh else dy.data_ptr(), h.data_ptr(),
                                              _ptr(dx), _ptr(dh), scratch.data_ptr(), n, B, G, F_, K, Nin,
                                              torch.cuda.current_stream().cuda_stream), "gf_nvgf_backward")
        db = dy.sum(dim=(0, 2)).unsqueeze(1) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dh, db, None


class _ExpandTaps(torch.autograd.Function):
    """h = weight[..., copyNodes] (NodeVariantGF.forward, graphML.py:2485).  Backward folds dh over the nodes that share a tap
    node with gf_nvgf_fold_taps (fixed order) instead of autograd's atomic index_add."""

    @staticmethod
    def forward(ctx, weight, copyNodes, grp_ptr, grp_idx):
        ctx.save_for_backward(grp_ptr, grp_idx)
        ctx.M = weight.shape[-1]
        return torch.index_select(weight, weight.dim() - 1, copyNodes)

    @staticmethod
    def backward(ctx, dh):
        grp_ptr, grp_idx = ctx.saved_tensors
        dh = dh.contiguous()
        _require_f32_cuda("dh", dh)
        N = dh.shape[-1]
        R = dh.numel() // N
        out = torch.empty(dh.shape[:-1] + (ctx.M,), dtype=torch.float32, device=dh.device)
        with torch.cuda.device(dh.device):
            _lib.check(_lib.lib().gf_nvgf_fold_taps(dh.data_ptr(), grp_ptr.data_ptr(), grp_idx.data_ptr(), out.data_ptr(), R, N, ctx.M,
                                                    torch.cuda.current_stream().cuda_stream), "gf_nvgf_fold_taps")
        return out, None, None, None


def expand_node_taps(weight, copyNodes, grp_ptr, grp_idx):
    return _ExpandTaps.apply(weight, copyNodes, grp_ptr, grp_idx)


def NVGF(h, S, x, b=None):
    """Node-variant graph filter, reference signature and semantics (graphML.py:293-387):

        y[b,f,n] = sum_{e,k,g} h[f,e,k,g,n] * (x_g S_e^k)[b,n] + b[f,n]

    h [F,E,K,G,N], S dense [E,N,N] or anything SparseGSO.from_any accepts, x [B,G,Nin<=N] (zero-padded, the output keeps Nin
    nodes: NodeVariantGF.forward's wrapper, :2487-2497), b [F,1] or [F,N] or None."""
    gso = SparseGSO.from_any(S)
    assert h.dim() == 5 and x.dim() == 3
    F_, E, K, G, N = h.shape
    assert gso.E == E and gso.N == N                                # graphML.py:346-347
    assert x.shape[1] == G and x.shape[2] <= N                      # :350-351
    _require_f32_cuda("x", x)
    _require_f32_cuda("h", h)
    fused = late = None
    if b is not None:
        _require_f32_cuda("b", b)
        assert b.dim() == 2 and b.shape[0] == F_
        fused, late = (b, None) if b.shape[1] == 1 else (None, b)
    y = _NVGFFunction.apply(x, h, fused, gso)
    if late is not None:
        y = y + late[:, : y.shape[2]]
    return y


class _EVGFFunction(torch.autograd.Function):
    """One edge feature of EVGF with per-edge storage: two C-ABI calls (gf_evgf_forward / gf_evgf_backward)."""

    @staticmethod
    def forward(ctx, x, wdiag, wedge, bias, pattern: EdgePattern):
        L = _lib.lib()
        B, G, Nin = x.shape
        F_, G2, N = wdiag.shape
        K = wedge.shape[1] + 1
        x, wdiag, wedge = x.contiguous(), wdiag.contiguous(), wedge.contiguous()
        bias_c = None if bias is None else bias.contiguous()
        dev = x.device
        with torch.cuda.device(dev):
            plan = pattern.plan(dev)
            V = torch.empty((K, F_ * G, N, B), dtype=torch.float32, device=dev)
            scratch = torch.empty(L.gf_evgf_scratch_floats(B, G, F_, N, 0), dtype=torch.float32, device=dev)
            y = torch.empty((B, F_, Nin), dtype=torch.float32, device=dev)
            _lib.check(L.gf_evgf_forward(plan, x.data_ptr(), wdiag.data_ptr(), wedge.data_ptr() if K > 1 else None, _ptr(bias_c),
                                         V.data_ptr(), scratch.data_ptr(), y.data_ptr(), B, G, F_, K, Nin,
                                         torch.cuda.current_stream().cuda_stream), "gf_evgf_forward")
        ctx.pattern = pattern
        ctx.dims = (B, G, F_, K, Nin, N)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, wdiag, wedge, V)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, wdiag, wedge, V = ctx.saved_tensors
        B, G, F_, K, Nin, N = ctx.dims
        need_dx, need_dd, need_de = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2] and K > 1
        need_db = ctx.has_bias and ctx.needs_input_grad[3]
        dy = dy.contiguous()
        dev = dy.device
        with torch.cuda.device(dev):
            plan = ctx.pattern.plan(dev)
            scratch = torch.empty(L.gf_evgf_scratch_floats(B, G, F_, N, 1), dtype=torch.float32, device=dev)
            dx = torch.empty_like(x) if need_dx else None
            dd = torch.empty_like(wdiag) if need_dd else None
            de = torch.empty_like(wedge) if need_de else None
            db = torch.empty((F_, 1), dtype=torch.float32, device=dev) if need_db else None
            _lib.check(L.gf_evgf_backward(plan, dy.data_ptr(), x.data_ptr(), wdiag.data_ptr(), wedge.data_ptr() if K > 1 else None,
                                          V.data_ptr(), scratch.data_ptr(), _ptr(dx), _ptr(dd), _ptr(de), _ptr(db), B, G, F_, K, Nin,
                                          torch.cuda.current_stream().cuda_stream), "gf_evgf_backward")
        if ctx.needs_input_grad[2] and K == 1:
            de = torch.zeros_like(wedge)
        return dx, dd, de, db, None


def EVGF_edges(pattern: EdgePattern, wdiag, wedge, x, b=None):
    """Edge-variant graph filter for one edge feature with per-edge storage (reference: EVGF, graphML.py:389-488, fed
    with Phi = weightEV * pattern, graphML.py:2676):

        v_0 = diag(wdiag[f,g]) x_g ;  v_k = Phi_k^{fg} v_{k-1}  (Phi_k^{fg} = wedge[f,k-1,g,:] on `pattern`) ;  y_f = sum_{g,k} v_k + b_f

    wdiag [F,G,N], wedge [F,K-1,G,nnzp], x [B,G,Nin<=N] (zero-padded), b [F,1]|None  ->  y [B,F,Nin]."""
    assert wdiag.dim() == 3 and wedge.dim() == 4 and x.dim() == 3
    F_, G, N = wdiag.shape
    assert N == pattern.N
    assert wedge.shape[0] == F_ and wedge.shape[2] == G and wedge.shape[3] == pattern.nnzp
    assert x.shape[1] == G                                     # graphML.py:444
    assert x.shape[2] <= N                                     # graphML.py:445 (== N); < N via EdgeVariantGF padding
    for name, t in (("x", x), ("wdiag", wdiag), ("wedge", wedge)):
        _require_f32_cuda(name, t)
    if b is not None:
        _require_f32_cuda("b", b)
        assert b.dim() == 2 and b.shape[0] == F_ and b.shape[1] == 1
    return _EVGFFunction.apply(x, wdiag, wedge, b, pattern)


class _MaxPoolLocalFunction(torch.autograd.Function):
    """MaxPoolLocal forward / backward through gf_maxpool_forward / gf_maxpool_backward (include/gfhip.h)."""

    @staticmethod
    def forward(ctx, x, nbh, rev_ptr, rev_i, rev_p):
        L = _lib.lib()
        B, F_, Nin = x.shape
        Nout, M = nbh.shape
        x = x.contiguous()
        with torch.cuda.device(x.device):
            v = torch.empty((B, F_, Nout), dtype=torch.float32, device=x.device)
            arg = torch.empty((B, F_, Nout), dtype=torch.int32, device=x.device)
            _lib.check(L.gf_maxpool_forward(x.data_ptr(), nbh.data_ptr(), v.data_ptr(), arg.data_ptr(), B, F_, Nin, Nout, M,
                                            torch.cuda.current_stream().cuda_stream), "gf_maxpool_forward")
        ctx.save_for_backward(arg, rev_ptr, rev_i, rev_p)
        ctx.dims = (B, F_, Nin, Nout)
        return v

    @staticmethod
    def backward(ctx, dv):
        L = _lib.lib()
        arg, rev_ptr, rev_i, rev_p = ctx.saved_tensors
        B, F_, Nin, Nout = ctx.dims
        dv = dv.contiguous()
        with torch.cuda.device(dv.device):
            dx = torch.empty((B, F_, Nin), dtype=torch.float32, device=dv.device)
            _lib.check(L.gf_maxpool_backward(dv.data_ptr(), arg.data_ptr(), rev_ptr.data_ptr(), rev_i.data_ptr(), rev_p.data_ptr(),
                                             dx.data_ptr(), B, F_, Nin, Nout, torch.cuda.current_stream().cuda_stream),
                       "gf_maxpool_backward")
        return dx, None, None, None, None


def max_pool_local(x, nbh, rev_ptr, rev_i, rev_p):
    """v[b,f,i] = max_{j in nbh[i]} x[b,f,j] (reference MaxPoolLocal.forward, graphML.py:1996-2021) on the HIP path."""
    _require_f32_cuda("x", x)
    return _MaxPoolLocalFunction.apply(x, nbh, rev_ptr, rev_i, rev_p)
