"""Fused per-Gaussian MLP heads on the CDNA4 matrix cores (csrc/mlp.hip).

``FusedHead`` wraps a WeightMLP / DeformMLP host mirror (riggs_amd.skeleton): same parameters (fp32 masters, the ones
the optimizer and the checkpoints see), but forward and backward run as ONE HIP launch each with 16-bit operands and
fp32 accumulation; the parameter gradients — (256 x N)·(N x K) products over all Gaussians — are one streaming launch per MLP
(csrc/mlp_wgrad.hip: ``riggs_mlp_wgrad``, every layer's product and bias sum, split partials summed by a second launch).  The operand format is a choice: ``"fp16"`` (default: 11 significand bits —
parameter gradients within ~2 % of the fp32 mirror; the incoming gradient is scaled by a power of two on the device so
that half precision's narrow range is never the limit, and the parameter gradients are scaled back) or ``"bf16"``
(8 bits, fp32's range, 3-11 % on the gradients).  The reference computes these MLPs in fp32
(skeleton_utils/network_utils.py:6-112), so the path is OPT-IN (``SkeletonWarp.use_fused_heads(True)``).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

ROWS = 64
FORMATS = {"fp16": torch.float16, "bf16": torch.bfloat16}
DEFAULT_FORMAT = "fp16"


def _fmt_dtype(fmt):
    if fmt not in FORMATS:
        raise ValueError("operand format must be 'fp16' or 'bf16'")
    return FORMATS[fmt]


def layout_probe() -> torch.Tensor:
    """D = A·B from the kernel's self-test (A = [I_16; 0], asymmetric B): rows 0..15 must equal B."""
    out = torch.zeros(32, 32, device="cuda")
    L.check(L.lib().riggs_mlp_layout_probe(out.data_ptr(), L.stream_ptr()), "riggs_mlp_layout_probe")
    return out


def expected_probe() -> torch.Tensor:
    k = torch.arange(16, dtype=torch.float32)[:, None]
    n = torch.arange(32, dtype=torch.float32)[None, :]
    b = torch.where(k < 8, 32 * k + n, -(32 * (k - 8) + n + 1))
    return torch.cat([b, torch.zeros(16, 32)], 0)


class Packed:
    """bf16 copies of the weights in the kernels' layouts.  The buffers are allocated once; ``repack()`` refills them from
    the fp32 masters with one launch (``riggs_mlp_pack``) whenever those changed."""

    def __init__(self, linears, head, in_ch: int, skip: int, fmt: str = None, tail_ch: int = 0):
        """``tail_ch`` > 0: the masters' input is ``in_ch + tail_ch`` wide and its last ``tail_ch`` values are THE SAME FOR EVERY ROW
        (DeformMLP's pose, skeleton_warp.py:152): those columns are not packed — ``set_tail(vector)`` folds what they contribute into
        the biases of the two layers that read the input (fp32), and ``param_grads`` writes their gradient as (bias gradient) x
        tail.  ``in_ch`` is then the per-row part alone."""
        dev = head.weight.device
        self.tail_ch = int(tail_ch)
        self.bias_eff = torch.empty(2, 256, device=dev) if self.tail_ch else None
        self._tail_set = False  # (forward() refuses to read bias_eff before the first set_tail)
        self.fmt = fmt or DEFAULT_FORMAT
        self.dtype, self.fp16 = _fmt_dtype(self.fmt), int(self.fmt == "fp16")
        self.linears, self.head = list(linears), head
        self.in_ch, self.in_pad, self.skip, self.depth = in_ch, (in_ch + 63) & ~63, skip, len(self.linears)
        self.out_ch = head.weight.shape[0]
        bf = dict(dtype=self.dtype, device=dev)
        kpad = [self.in_pad if l == 0 else (self.in_pad + 256 if l == skip + 1 else 256) for l in range(self.depth)]
        self.w = [torch.empty(256, k, **bf) for k in kpad]
        self.wt = [None] + [torch.empty(256, 256, **bf) for _ in range(1, self.depth)]
        self.w_out = torch.empty(32, 256, **bf)
        self.w_out_t_bf16 = torch.empty(256, 32, **bf)
        self._wp = (C.c_void_p * self.depth)(*[t.data_ptr() for t in self.w])
        self._wtp = (C.c_void_p * self.depth)(*[(t.data_ptr() if t is not None else None) for t in self.wt])
        self.repack()

    def rows(self, l: int) -> torch.Tensor:
        """Layer ``l``'s packed weights back in row-major (256, K_pad) order (tests, debugging): the kernels' copy is
        FRAGMENT-MAJOR — [neuron tile T][K-step s][lane][8 values] with value j of lane = W[32 T + (lane & 31)][16 s + 8 (lane >> 5) + j]
        (csrc/mlp.hip: one contiguous 1 KB run per wave load)."""
        w = self.w[l]
        S = w.shape[1] // 16
        return w.reshape(8, S, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(256, S * 16)

    def repack(self):
        ws = [L.require_cuda_f32("weight", lin.weight.detach()) for lin in self.linears]
        wo = L.require_cuda_f32("head weight", self.head.weight.detach())
        src = (C.c_void_p * self.depth)(*[t.data_ptr() for t in ws])
        L.check(L.lib().riggs_mlp_pack_tail(self.in_ch, self.tail_ch, self.out_ch, self.depth, self.skip, src, wo.data_ptr(), self._wp,
                                            self._wtp, self.w_out.data_ptr(), self.w_out_t_bf16.data_ptr(), self.fp16, L.stream_ptr()),
                "riggs_mlp_pack_tail")
        self._w_masters = ws
        self.b = [L.require_cuda_f32("bias", lin.bias.detach()) for lin in self.linears]  # (views of the masters)
        self.b_out = L.require_cuda_f32("head bias", self.head.bias.detach())
        bp = [t.data_ptr() for t in self.b]
        if self.tail_ch:  # (the two layers that read the input take their bias from bias_eff: set_tail)
            bp[0], bp[self.skip + 1] = self.bias_eff[0].data_ptr(), self.bias_eff[1].data_ptr()
        self._bp = (C.c_void_p * self.depth)(*bp)

    def set_tail(self, tail: torch.Tensor) -> torch.Tensor:
        """Fold this frame's constant input tail into the biases (``riggs_mlp_tail_bias``: one launch); returns the vector as the
        kernels read it (what ``param_grads`` needs again)."""
        tail = L.require_cuda_f32("tail", tail.detach().reshape(-1), (self.tail_ch,))
        s = self.skip + 1
        L.check(L.lib().riggs_mlp_tail_bias(self.in_ch, self.tail_ch, self._w_masters[0].data_ptr(), self.b[0].data_ptr(),
                                            self._w_masters[s].data_ptr(), self.b[s].data_ptr(), tail.data_ptr(),
                                            self.bias_eff.data_ptr(), L.stream_ptr()), "riggs_mlp_tail_bias")
        self._tail_set = True
        return tail


def embed_bf16(p: Packed, x_emb: torch.Tensor) -> torch.Tensor:
    """(N, in_ch) fp32 -> (N rounded up to 128, in_pad) bf16, zero padded: the kernels' A operand of the first and the
    skip layer, and the right-hand side of their weight gradients."""
    N = x_emb.shape[0]
    x_emb = L.require_cuda_f32("x_emb", x_emb, (N, p.in_ch))
    xb = torch.zeros((N + 127) // 128 * 128, p.in_pad, dtype=p.dtype, device=x_emb.device)
    xb[:N, :p.in_ch] = x_emb
    return xb


def embed_positions_bf16(x: torch.Tensor, multires: int, tail: torch.Tensor = None, fmt: str = None) -> torch.Tensor:
    """[x, sin(2^k x), cos(2^k x) ..., tail] per row as the kernels' padded 16-bit operand (``fmt``), in one launch (``tail``:
    a vector appended to every row — DeformMLP's pose)."""
    dtype = _fmt_dtype(fmt or DEFAULT_FORMAT)
    N = x.shape[0]
    x = L.require_cuda_f32("x", x, (N, 3))
    n_tail = 0 if tail is None else tail.numel()
    in_pad = (3 * (1 + 2 * multires) + n_tail + 63) & ~63
    if tail is not None:
        tail = L.require_cuda_f32("tail", tail.reshape(-1))
    xb = torch.empty((N + 127) // 128 * 128, in_pad, dtype=dtype, device=x.device)
    L.check(L.lib().riggs_mlp_embed(N, multires, n_tail, x.data_ptr(), L.ptr(tail), xb.data_ptr(), int(dtype == torch.float16),
                                    L.stream_ptr()), "riggs_mlp_embed")
    return xb


def forward(p: Packed, x_emb: torch.Tensor, want_acts: bool, xb: torch.Tensor = None, n_dev: torch.Tensor = None,
            sigmoid: bool = False, res_base: torch.Tensor = None, res_mask: torch.Tensor = None):
    """``n_dev`` (device int32 scalar, optional): only the first ``min(n_dev, N)`` rows exist (include/riggs_hip.h:
    n_rows_dev) — the compacted rows of the row-sparse backward.  ``sigmoid`` / ``res_base`` / ``res_mask``: the output epilogue
    (``struct riggs_mlp_epilogue``) — with ``res_base`` the third return value is ``res_base + out * res_mask``."""
    N = x_emb.shape[0]
    if p.tail_ch and not p._tail_set:
        raise ValueError("a head with a constant input tail: Packed.set_tail(vector) comes before its first forward")
    if xb is None:
        xb = embed_bf16(p, x_emb)
    out = torch.empty(N, p.out_ch, device=x_emb.device)
    acts = masks = None
    if want_acts:
        rows = L.lib().riggs_mlp_rows_per_workgroup()
        acts = torch.empty(p.depth, N, 256, dtype=p.dtype, device=x_emb.device)
        masks = torch.empty(p.depth, (N + rows - 1) // rows, 256, 4, dtype=torch.int32, device=x_emb.device)
    epi, res_out = None, None
    if sigmoid or res_base is not None:
        epi = L.MlpEpilogue()
        epi.sigmoid = int(bool(sigmoid))
        if res_base is not None:
            res_base = L.require_cuda_f32("res_base", res_base, (N, p.out_ch))
            if res_mask is not None:
                res_mask = L.require_cuda_f32("res_mask", res_mask.reshape(-1), (N,))
            res_out = torch.empty_like(res_base)
            epi.res_base, epi.res_mask, epi.res_out = res_base.data_ptr(), L.ptr(res_mask), res_out.data_ptr()
        epi = C.byref(epi)
    L.check(L.lib().riggs_mlp_forward(N, p.in_ch, p.out_ch, p.depth, p.skip, p._wp, p._bp, p.w_out.data_ptr(),
                                      p.b_out.data_ptr(), xb.data_ptr(), L.ptr(acts), L.ptr(masks), out.data_ptr(), L.ptr(n_dev),
                                      epi, p.fp16, L.stream_ptr()), "riggs_mlp_forward")
    if res_base is not None:
        return out, ((acts, masks) if want_acts else None), res_out
    return out, (acts, masks) if want_acts else None


_ZERO_WORD = {}


def _aligned(t: torch.Tensor) -> torch.Tensor:
    """``t`` at a 16-byte aligned address (the kernels read 16-byte pieces): a contiguous VIEW at an odd float offset — e.g. a
    ``narrow`` of a ``torch.cat`` backward — is copied."""
    return t if t.data_ptr() % 16 == 0 else t.clone()


def grad_scale(g_out: torch.Tensor) -> torch.Tensor:
    """The power of two (a device scalar: no host synchronisation) that lifts max|g_out| to ~2^10: with fp16 operands the
    gradients of a per-pixel-averaged loss (1e-6 .. 1e-9) would otherwise sit in half precision's subnormals.  Two launches
    (``riggs_mlp_grad_scale``; as torch ops it was seven and a copy of the tensor)."""
    if g_out.numel() == 0:
        return torch.ones(1, device=g_out.device)
    g = _aligned(L.require_cuda_f32("g_out", g_out))
    key = (g.device, L.stream_ptr())  # (one accumulator word per stream: two heads' backwards on two streams must not share it)
    word = _ZERO_WORD.get(key)
    if word is None:
        word = _ZERO_WORD[key] = torch.zeros(1, dtype=torch.int32, device=g.device)
    scale = torch.empty(1, device=g.device)
    L.check(L.lib().riggs_mlp_grad_scale(g.numel(), g.data_ptr(), scale.data_ptr(), word.data_ptr(), L.stream_ptr()), "riggs_mlp_grad_scale")
    return scale


_L2_SCRATCH = {}


def l2_grad_scale(g_out: torch.Tensor, out: torch.Tensor, coef: torch.Tensor, mean_sq: torch.Tensor = None):
    """``riggs_mlp_l2_grad_scale``: ``g_eff = g_out + coef * out`` (the L2 regulariser on the MLP's output folded into its
    cotangent; ``coef`` a device scalar), the fp16 gradient scale of ``g_eff`` and — into ``mean_sq`` when given — mean(out^2).
    The two launches ``grad_scale`` makes anyway.  Returns ``(g_eff, scale)``."""
    g = _aligned(L.require_cuda_f32("g_out", g_out))
    o = _aligned(L.require_cuda_f32("out", out, tuple(g.shape)))
    key = (g.device, L.stream_ptr())
    sc = _L2_SCRATCH.get(key)
    if sc is None:
        sc = _L2_SCRATCH[key] = (torch.zeros(1, dtype=torch.int32, device=g.device), torch.empty(512, device=g.device))
    g_eff = torch.empty_like(g)
    scale = torch.empty(1, device=g.device)
    L.check(L.lib().riggs_mlp_l2_grad_scale(g.numel(), g.data_ptr(), o.data_ptr(), coef.data_ptr(), g_eff.data_ptr(), scale.data_ptr(),
                                            sc[0].data_ptr(), sc[1].data_ptr(), L.ptr(mean_sq), L.stream_ptr()), "riggs_mlp_l2_grad_scale")
    return g_eff, scale


def cotangent(g: torch.Tensor, g_rows: torch.Tensor, row_mask: torch.Tensor, sigmoid_out: torch.Tensor, l2_out: torch.Tensor,
              l2_coef: torch.Tensor, mean_sq: torch.Tensor = None):
    """``riggs_mlp_cotangent``: ``g_eff = (g + g_rows * row_mask[row]) * [s (1 - s)] + l2_coef * l2_out`` for an (N, out_ch) head —
    every piece optional (``g`` or ``g_rows`` must be there) — and the fp16 gradient scale of ``g_eff``; the two launches
    ``grad_scale`` makes anyway, whatever is folded in.  Returns ``(g_eff, scale)``."""
    ref = g if g is not None else g_rows
    N, out_ch = ref.shape
    tens = {}
    for name, t in (("g", g), ("g_rows", g_rows), ("sigmoid_out", sigmoid_out), ("l2_out", l2_out)):
        tens[name] = None if t is None else _aligned(L.require_cuda_f32(name, t, (N, out_ch)))
    if row_mask is not None:
        row_mask = L.require_cuda_f32("row_mask", row_mask.reshape(-1), (N,))
    key = (ref.device, L.stream_ptr())
    sc = _L2_SCRATCH.get(key)
    if sc is None:
        sc = _L2_SCRATCH[key] = (torch.zeros(1, dtype=torch.int32, device=ref.device), torch.empty(512, device=ref.device))
    g_eff = torch.empty(N, out_ch, device=ref.device)
    scale = torch.empty(1, device=ref.device)
    L.check(L.lib().riggs_mlp_cotangent(N, out_ch, L.ptr(tens["g"]), L.ptr(tens["g_rows"]), L.ptr(row_mask), L.ptr(tens["sigmoid_out"]),
                                        L.ptr(tens["l2_out"]), L.ptr(l2_coef) if l2_out is not None else None, g_eff.data_ptr(),
                                        scale.data_ptr(), sc[0].data_ptr(), sc[1].data_ptr(), L.ptr(mean_sq), L.stream_ptr()),
            "riggs_mlp_cotangent")
    return g_eff, scale


def live_rows(p: Packed, g_out: torch.Tensor, xb: torch.Tensor, sigmoid_out: torch.Tensor = None, want_scale: bool = False):
    """The rows of ``g_out`` (N, out_ch) that hold a non-zero, compacted in ascending order (``riggs_mlp_live_rows``: two launches,
    no atomics, no host synchronisation): ``(idx (N) int32, count (1) int32 on the device, xb_live, g_live[, scale])`` — the
    gathered rows of the padded 16-bit operand ``xb`` and of ``g_out``; only the first ``count`` rows of each are defined.
    ``sigmoid_out``: the head's sigmoid-ed output — ``g_out * s (1 - s)`` is what is tested and gathered.  ``want_scale``: the fp16
    gradient scale of that cotangent comes out of the same two launches."""
    N = g_out.shape[0]
    dev = g_out.device
    g_out = L.require_cuda_f32("g_out", g_out, (N, p.out_ch))
    if sigmoid_out is not None:
        sigmoid_out = L.require_cuda_f32("sigmoid_out", sigmoid_out, (N, p.out_ch))
    idx = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    xl = torch.empty_like(xb)
    gl = torch.empty_like(g_out)
    scale = torch.empty(1, device=dev) if want_scale else None
    ws = torch.empty(int(L.lib().riggs_mlp_live_rows_workspace_bytes(N)) // 8 + 1, dtype=torch.int64, device=dev)
    L.check(L.lib().riggs_mlp_live_rows(N, p.out_ch, p.in_ch, g_out.data_ptr(), L.ptr(sigmoid_out), xb.data_ptr(), ws.data_ptr(),
                                        idx.data_ptr(), count.data_ptr(), xl.data_ptr(), gl.data_ptr(), L.ptr(scale), L.stream_ptr()),
            "riggs_mlp_live_rows")
    if want_scale:
        return idx, count, xl, gl, scale
    return idx, count, xl, gl


def backward_data(p: Packed, g_out: torch.Tensor, masks: torch.Tensor, scale: torch.Tensor = None, bias_sums: bool = True,
                  n_dev: torch.Tensor = None):
    """dL/d(pre-activation) of every hidden layer in the 16-bit format (depth, N, 256), and (``bias_sums``) the bias gradients
    (depth, 256) — both times ``scale`` when given (``grad_scale``).  Without ``bias_sums`` the second value is None:
    ``param_grads`` sums the columns beside its products."""
    N = g_out.shape[0]
    g_out = L.require_cuda_f32("g_out", g_out, (N, p.out_ch))
    dpre = torch.empty(p.depth, N, 256, dtype=p.dtype, device=g_out.device)
    rows = L.lib().riggs_mlp_rows_per_workgroup()
    db_part = torch.empty((N + rows - 1) // rows, p.depth, 256, device=g_out.device) if bias_sums else None
    L.check(L.lib().riggs_mlp_backward(N, p.out_ch, p.depth, p.skip, p._wtp, p.w_out_t_bf16.data_ptr(), g_out.data_ptr(),
                                       L.ptr(scale), masks.data_ptr(), dpre.data_ptr(), L.ptr(db_part), L.ptr(n_dev), p.fp16,
                                       L.stream_ptr()), "riggs_mlp_backward")
    if bias_sums and n_dev is not None:
        raise ValueError("bias_sums with a device-side row count: the workgroups past the count leave their slices unwritten")
    return dpre, (db_part.sum(0) if bias_sums else None)


def param_grads(p: Packed, xb: torch.Tensor, acts: torch.Tensor, dpre: torch.Tensor, g_out: torch.Tensor, scale: torch.Tensor = None,
                n_dev: torch.Tensor = None, tail: torch.Tensor = None):
    """Every parameter gradient of the MLP — [dW_0, db_0, ..., dW_{D-1}, db_{D-1}, dW_out, db_out], fp32, the masters' shapes,
    ``scale`` taken out again — from the operands the two passes left in memory (``riggs_mlp_wgrad``: three launches)."""
    N = g_out.shape[0]
    dev = g_out.device
    g_out = L.require_cuda_f32("g_out", g_out, (N, p.out_ch))
    in_true = p.in_ch + p.tail_ch
    if p.tail_ch and tail is None:
        raise ValueError("a head with a constant input tail needs the tail of its forward for the parameter gradients")
    k_true = [in_true if l == 0 else (in_true + 256 if l == p.skip + 1 else 256) for l in range(p.depth)]
    if N == 0:
        out = []
        for k in k_true:
            out += [torch.zeros(256, k, device=dev), torch.zeros(256, device=dev)]
        return out + [torch.zeros(p.out_ch, 256, device=dev), torch.zeros(p.out_ch, device=dev)]
    gw = [torch.empty(256, k, device=dev) for k in k_true]
    gb = torch.empty(p.depth, 256, device=dev)
    gwo, gbo = torch.empty(p.out_ch, 256, device=dev), torch.empty(p.out_ch, device=dev)
    nbytes = int(L.lib().riggs_mlp_wgrad_workspace_bytes(N, p.in_ch, p.depth, p.skip))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    gwp = (C.c_void_p * p.depth)(*[t.data_ptr() for t in gw])
    gbp = (C.c_void_p * p.depth)(*[gb[l].data_ptr() for l in range(p.depth)])
    L.check(L.lib().riggs_mlp_wgrad_tail(N, p.in_ch, p.tail_ch, L.ptr(tail) if p.tail_ch else None, p.out_ch, p.depth, p.skip,
                                         xb.data_ptr(), acts.data_ptr(), dpre.data_ptr(), g_out.data_ptr(), L.ptr(scale), ws.data_ptr(),
                                         nbytes, gwp, gbp, gwo.data_ptr(), gbo.data_ptr(), L.ptr(n_dev), p.fp16, L.stream_ptr()),
            "riggs_mlp_wgrad_tail")
    out = []
    for l in range(p.depth):
        out += [gw[l], gb[l]]
    return out + [gwo, gbo]


# ---- the same gradients through the library (torch.bmm = hipBLASLt): what ``param_grads`` replaced; kept as the comparison
# the tests and tools/mlp_kernel_time.py run beside it, never called by the product path
def _wgrad(d: torch.Tensor, a: torch.Tensor, splits: int = 128) -> torch.Tensor:
    """d^T · a for d (N, M), a (N, K) in bf16 with fp32 output.  The reduction runs over N = 3e5 rows: one GEMM with
    K_gemm = N leaves the library without parallelism (0.63 ms at 256 x 256 outputs); a batched GEMM over `splits` row
    blocks plus a sum is 8x faster (0.08 ms)."""
    N = d.shape[0]
    n = N // splits
    if n < 64:
        return torch.mm(d.t(), a, out_dtype=torch.float32)
    main = n * splits
    out = torch.bmm(d[:main].view(splits, n, -1).transpose(1, 2), a[:main].view(splits, n, -1), out_dtype=torch.float32).sum(0)
    if main < N:
        out = out + torch.mm(d[main:].t(), a[main:], out_dtype=torch.float32)
    return out


def _wgrad_multi(d: torch.Tensor, a: torch.Tensor, splits: int = 128) -> torch.Tensor:
    """``_wgrad`` for a stack of layers: d (L, N, M), a (L, N, K) -> (L, M, K), one batched GEMM over L x splits blocks.

    The (layer, split) axes only merge into one batch axis without a copy when the split count divides N, so the
    divisor of N nearest ``splits`` is used; without one in [splits/2, 2*splits] the layers go one at a time."""
    Lr, N = d.shape[0], d.shape[1]
    best = 0
    for c in range(splits // 2, 2 * splits + 1):
        if N % c == 0 and abs(c - splits) < abs(best - splits):
            best = c
    if best == 0 or N // best < 64:
        return torch.stack([_wgrad(d[i], a[i], splits) for i in range(Lr)])
    n = N // best
    dm = d.view(Lr * best, n, d.shape[2])
    am = a.view(Lr * best, n, a.shape[2])
    return torch.bmm(dm.transpose(1, 2), am, out_dtype=torch.float32).view(Lr, best, d.shape[2], a.shape[2]).sum(1)


_ONES = {}


def _colsum(d: torch.Tensor) -> torch.Tensor:
    """Column sums of a tall bf16 matrix in fp32, as a (split-K) GEMM with a block of ones: torch's column reduction of a
    (3e5, 256) bf16 tensor takes 0.09 ms and of a (3e5, 23) fp32 one 0.6 ms; this is ~0.03 ms."""
    key = (d.device, d.shape[0], d.dtype)
    ones = _ONES.get(key)
    if ones is None:
        ones = _ONES[key] = torch.ones(d.shape[0], 8, dtype=d.dtype, device=d.device)
    return _wgrad(d, ones)[:, 0].contiguous()


def library_param_grads(p: Packed, xb: torch.Tensor, acts: torch.Tensor, dpre: torch.Tensor, db: torch.Tensor, g_out: torch.Tensor,
                        scale: torch.Tensor = None):
    """``param_grads`` through library GEMMs (comparison only; ``db`` from ``backward_data(..., bias_sums=True)``)."""
    # every layer's product with the previous layer's activations has the same shape — the skip layer's hidden block
    # included — so layers 1 .. depth - 1 are ONE batched split-K GEMM; the two products with the embedding stay single calls
    if p.tail_ch:
        raise ValueError("library_param_grads: the comparison path knows no constant tail")
    gws = [None] * p.depth
    ls = p.skip + 1
    xb = xb[:g_out.shape[0]]
    g_run = _wgrad_multi(dpre[1:p.depth], acts[0:p.depth - 1])
    for l in range(1, p.depth):
        gws[l] = g_run[l - 1]
    gws[0] = _wgrad(dpre[0], xb)[:, :p.in_ch]
    gws[ls] = torch.cat([_wgrad(dpre[ls], xb)[:, :p.in_ch], gws[ls]], 1)
    grads = []
    for l in range(p.depth):
        grads += [gws[l], db[l]]
    gob = torch.nn.functional.pad(g_out if scale is None else g_out * scale, (0, 32 - p.out_ch)).to(p.dtype)
    grads += [_wgrad(gob, acts[p.depth - 1])[:p.out_ch], _colsum(gob)[:p.out_ch]]
    if scale is not None:  # every product above carries the factor once: take it out again (a power of two: exact)
        grads = [g.contiguous() for g in grads]
        torch._foreach_mul_(grads, torch.reciprocal(scale).reshape(()))
    return grads


class _FusedMLP(torch.autograd.Function):
    """out = MLP(x_emb) with the fused kernels; gradients for the (fp32 master) parameters only.  With ``res = (base, mask)`` the
    function returns ``(out, base + out * mask)`` — the residual join of the forward's epilogue — and hands ``base`` its cotangent
    straight through."""

    @staticmethod
    def forward(ctx, x_emb, head, l2, res_base, res_mask, *params):
        p = head._packed()
        ctx.tail = p.set_tail(head._tail) if p.tail_ch else None  # (this frame's pose -> the two biases; kept for the backward)
        ctx.l2 = l2  # None, or (coef, mean_sq): an L2 regulariser on the output folded into the backward (cotangent)
        n_rows = head._n_rows if x_emb.dtype == p.dtype else x_emb.shape[0]
        xb = x_emb if x_emb.dtype == p.dtype else embed_bf16(p, x_emb)
        ctx.head, ctx.p, ctx.n, ctx.sparse, ctx.sig = head, p, n_rows, bool(head.sparse_rows), bool(head.out_sigmoid)
        ctx.res = res_base is not None
        ctx.res_mask = res_mask.detach() if res_mask is not None else None
        ctx.set_materialize_grads(False)
        if ctx.sparse:
            # row-sparse backward: nothing is stored here — the backward repeats the forward for the rows that carry a gradient
            r = forward(p, xb[:n_rows], False, xb, sigmoid=ctx.sig, res_base=res_base, res_mask=res_mask)
            ctx.save_for_backward(xb)
        else:
            r = forward(p, xb[:n_rows], True, xb, sigmoid=ctx.sig, res_base=res_base, res_mask=res_mask)
            acts, masks = r[1]
            ctx.save_for_backward(xb, acts, masks)
        out = r[0]
        ctx.out = out if (l2 is not None or ctx.sig) else None  # (detached inside a Function: no reference cycle)
        if ctx.res:
            return out, r[2]
        return out

    @staticmethod
    def backward(ctx, g_out, g_res=None):
        p = ctx.p
        n_lead = 5
        if g_out is None and g_res is None:
            return (None,) * (n_lead + 2 * p.depth + 2)
        g_out = None if g_out is None else _aligned(g_out.contiguous())
        g_res = None if g_res is None else g_res.contiguous()
        sig = ctx.out if ctx.sig else None
        plain = g_res is None and ctx.l2 is None  # the cotangent is g_out (x the sigmoid's factor)
        if ctx.sparse and plain:
            (xb,) = ctx.saved_tensors
            _, count, xl, gl, scale = live_rows(p, g_out, xb, sig, want_scale=True)
            if not p.fp16:
                scale = None
            if p.tail_ch:
                p.set_tail(ctx.tail)  # (the repeated forward reads the biases of THIS call's frame)
            _, (acts, masks) = forward(p, xl[:ctx.n], True, xl, n_dev=count)
            dpre, _ = backward_data(p, gl, masks, scale, bias_sums=False, n_dev=count)
            grads = param_grads(p, xl, acts, dpre, gl, scale, n_dev=count, tail=ctx.tail)
            ctx.head.last_live_count = count
        else:
            if plain and sig is None:
                scale = grad_scale(g_out) if p.fp16 else None
            else:
                l2o, l2c, msq = (ctx.out, ctx.l2[0], ctx.l2[1]) if ctx.l2 is not None else (None, None, None)
                g_out, scale = cotangent(g_out, g_res, ctx.res_mask if g_res is not None else None, sig, l2o, l2c, msq)
                if not p.fp16:
                    scale = None
            if ctx.sparse:
                (xb,) = ctx.saved_tensors
                _, count, xl, gl = live_rows(p, g_out, xb)
                if p.tail_ch:
                    p.set_tail(ctx.tail)
                _, (acts, masks) = forward(p, xl[:ctx.n], True, xl, n_dev=count)
                dpre, _ = backward_data(p, gl, masks, scale, bias_sums=False, n_dev=count)
                grads = param_grads(p, xl, acts, dpre, gl, scale, n_dev=count, tail=ctx.tail)
                ctx.head.last_live_count = count
            else:
                xb, acts, masks = ctx.saved_tensors
                dpre, _ = backward_data(p, g_out, masks, scale, bias_sums=False)
                grads = param_grads(p, xb, acts, dpre, g_out, scale, tail=ctx.tail)
        return (None, None, None, g_res if ctx.res else None, None) + tuple(grads)


class FusedHead:
    """Runs ``net`` (a WeightMLP or DeformMLP host mirror) through the fused kernels.  ``net`` keeps owning the fp32
    parameters; the bf16 copies are rebuilt when a parameter's version counter changes (optimizer step, load)."""

    def __init__(self, linears, head_linear, in_ch: int, skip: int, fmt: str = None, sparse_rows: bool = False,
                 out_sigmoid: bool = False, tail_ch: int = 0):
        """``sparse_rows``: the backward runs on the rows whose cotangent is non-zero (``live_rows``) and the forward stores no
        activations — exact (a zero row contributes zero to every parameter gradient), and the right choice for a head whose
        cotangent reaches only the Gaussians the render touched (the WeightMLP: 10-30 % of the rows); a head with a dense
        cotangent (the DeformMLP under its L2 regulariser, train_rig.py:446-454) would pay a second forward for nothing."""
        self.fmt = fmt or DEFAULT_FORMAT
        self.sparse_rows = bool(sparse_rows)
        self.out_sigmoid = bool(out_sigmoid)  # the head's value goes through a sigmoid inside the forward launch (WeightMLP)
        self.last_live_count = None  # device int32 (1,): the live rows of the last row-sparse backward
        self.linears, self.head_linear, self.in_ch, self.skip = list(linears), head_linear, in_ch, skip
        self.tail_ch = int(tail_ch)  # the input's last tail_ch values are one vector for all rows (``__call__(..., tail=)``): Packed
        self._tail = None
        self._pk, self._ver, self._ptrs = None, None, None

    def params(self):
        ps = []
        for lin in self.linears:
            ps += [lin.weight, lin.bias]
        return ps + [self.head_linear.weight, self.head_linear.bias]

    def _packed(self) -> Packed:
        ptrs = tuple(q.data_ptr() for q in self.params())
        ver = tuple(q._version for q in self.params())
        if self._pk is None or ptrs != self._ptrs:
            self._pk, self._ptrs, self._ver = Packed(self.linears, self.head_linear, self.in_ch, self.skip, self.fmt, self.tail_ch), ptrs, ver
            fresh = True
        else:
            fresh = False
        if torch.cuda.is_current_stream_capturing():
            # inside a hipGraph capture the conversion must be PART of the graph: a replay sees new fp32 masters
            # (the captured optimizer step) without this Python running again
            self._pk.repack()
            self._ver = None
        elif not fresh and ver != self._ver:
            self._pk.repack()
            self._ver = ver
        return self._pk

    def __call__(self, x_emb: torch.Tensor, n_rows: int = None, l2=None, res=None, tail: torch.Tensor = None):
        """``x_emb``: (N, in_ch) fp32, or the padded bf16 operand of ``embed_positions_bf16`` together with ``n_rows`` = N.
        ``l2``: None, or ``(coef, mean_sq)`` — device scalars: the backward adds ``coef * output`` to the incoming cotangent
        (d/d output of ``lambda * mean(output^2)`` for ``coef = 2 lambda / output.numel()``) and writes mean(output^2) into
        ``mean_sq`` (may be None); no launch beyond the gradient scale's two.
        ``res``: None, or ``(base, mask)`` — the call returns ``(output, base + output * mask)`` (``mask``: (N, 1) / (N) without a
        gradient of its own, or None = 1), joined by the forward launch's epilogue; the backward sends the second value's cotangent
        through to ``base`` and, times ``mask``, into the MLP."""
        self._n_rows = n_rows
        if self.tail_ch and tail is None:
            raise ValueError("this head was built with a constant input tail: pass it (tail=)")
        if tail is not None and not self.tail_ch:
            raise ValueError("tail= needs a head built with tail_ch (the input's constant part is then left out of x_emb)")
        self._tail = tail
        if res is not None:
            return _FusedMLP.apply(x_emb.contiguous(), self, l2, res[0], res[1], *self.params())
        return _FusedMLP.apply(x_emb.contiguous(), self, l2, None, None, *self.params())
