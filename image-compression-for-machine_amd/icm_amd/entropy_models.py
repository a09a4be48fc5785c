"""compressai/entropy_models mirror: EntropyModel / EntropyBottleneck / GaussianConditional
(entropy_models.py:66-666).  Likelihoods run on the fused HIP kernels; ``update()`` builds the quantised CDF tables
from device-computed pmfs (``icm_eb_pmf_table`` / ``icm_gc_pmf_table`` + host ``icm_pmf_to_quantized_cdf``);
``compress()`` / ``decompress()`` quantise on the device (``icm_quantize`` / ``icm_gc_build_indexes`` /
``icm_dequantize``) and code the symbols with the host rANS coder (``icm_amd.ans``), as the reference does."""
from __future__ import annotations

import ctypes as C
from typing import Any, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from ._lib import check, ptr
from .layers import run_module
from .ops import LowerBound


class EntropyModel(nn.Module):
    """entropy_models.py:70-170 (quantize / buffers); the coder proxy is not instantiated."""

    def __init__(self, likelihood_bound: float = 1e-9, entropy_coder: Optional[str] = None,
                 entropy_coder_precision: int = 16):
        super().__init__()
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        self._lik_bound = float(likelihood_bound)
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        self._injected_noise = None

    offset = property(lambda self: self._offset)
    quantized_cdf = property(lambda self: self._quantized_cdf)
    cdf_length = property(lambda self: self._cdf_length)

    def inject_noise(self, noise: Optional[torch.Tensor]):
        """testing hook: use this U(-1/2,1/2) sample instead of drawing one in train mode"""
        self._injected_noise = noise

    def _noise_like(self, x):
        if self._injected_noise is not None:
            return self._injected_noise.to(x.device, torch.float32).contiguous()
        return torch.rand(x.shape, dtype=torch.float32, device=x.device) - 0.5  # RNG plumbing (entropy_models.py:131-135)

    # ---- quantisation (entropy_models.py:126-170)
    @staticmethod
    def _mean_strides(x, means):
        """(tensor, m_bs, m_cs, m_ps) addressing ``means`` for every element of x [N,C,*]: full-size means, or means that
        broadcast over the spatial dims / batch (per-channel medians: entropy_models.py:500-504)"""
        if means is None:
            return None, 0, 0, 0
        N, Cc = x.shape[0], x.shape[1]
        HW = x[0, 0].numel()
        m = means.to(torch.float32)
        if m.shape == x.shape:
            m = m.contiguous()
            return m, Cc * HW, HW, 1
        if m.dim() == x.dim() and m.shape[1] == Cc and all(d == 1 for d in m.shape[2:]) and m.shape[0] in (1, N):
            m = m.reshape(m.shape[0], Cc).contiguous()
            return m, (Cc if m.shape[0] == N and N > 1 else 0), 1, 0
        m = m.expand(x.shape).contiguous()
        return m, Cc * HW, HW, 1

    def quantize(self, inputs, mode, means=None):
        if mode not in ("noise", "dequantize", "symbols"):
            raise ValueError(f'Invalid quantization mode: "{mode}"')
        x = inputs.to(torch.float32).contiguous()
        if mode == "noise":
            out = x.clone()
            noise = self._noise_like(x)
            check(L.lib().icm_add_grad(ptr(noise), 0, ptr(out), out.numel(), 1, L.stream()), "add noise")
            return out
        if x.dim() < 2:
            x = x.reshape(1, -1)
            means = None if means is None else means.reshape(1, -1)
        N, Cc = x.shape[0], x.shape[1]
        HW = max(1, x[0, 0].numel())
        m, mbs, mcs, mps = self._mean_strides(x, means)
        sym = torch.empty(x.shape, dtype=torch.int32, device=x.device) if mode == "symbols" else None
        deq = torch.empty_like(x) if mode == "dequantize" else None
        check(L.lib().icm_quantize(ptr(x), Cc * HW, ptr(m), mbs, mcs, mps, ptr_any(sym), ptr(deq), N, Cc, HW, L.stream()),
              "quantize")
        return (sym if mode == "symbols" else deq).reshape(inputs.shape)

    def _quantize(self, inputs, mode, means=None):
        import warnings
        warnings.warn("_quantize is deprecated. Use quantize instead.")
        return self.quantize(inputs, mode, means)

    @staticmethod
    def dequantize(inputs, means=None):
        """float(symbols) + means (entropy_models.py:159-166)"""
        sym = inputs.to(torch.int32).contiguous()
        if sym.dim() < 2:
            out = sym.float()
            return out + means if means is not None else out
        N, Cc = sym.shape[0], sym.shape[1]
        HW = max(1, sym[0, 0].numel())
        m, mbs, mcs, mps = EntropyModel._mean_strides(sym, means)
        out = torch.empty(sym.shape, dtype=torch.float32, device=sym.device)
        check(L.lib().icm_dequantize(ptr_any(sym), ptr(m), mbs, mcs, mps, ptr(out), Cc * HW, N, Cc, HW, L.stream()),
              "dequantize")
        return out

    @classmethod
    def _dequantize(cls, inputs, means=None):
        import warnings
        warnings.warn("_dequantize. Use dequantize instead.")
        return cls.dequantize(inputs, means)

    # ---- CDF tables (entropy_models.py:172-199)
    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        """quantised CDF rows from per-row pmfs (host: ``icm_pmf_to_quantized_cdf``), zero-padded to max_length + 2"""
        from .ans import pmf_to_quantized_cdf
        pm = pmf.detach().cpu().float().numpy()
        tm = tail_mass.detach().cpu().float().numpy().reshape(len(pm), -1)
        pl = [int(v) for v in pmf_length.detach().cpu().reshape(-1).tolist()]
        cdf = torch.zeros((len(pl), int(max_length) + 2), dtype=torch.int32)
        for i, n in enumerate(pl):
            row = pmf_to_quantized_cdf(np.concatenate((pm[i, :n], tm[i, :1])), self.entropy_coder_precision)
            cdf[i, :len(row)] = torch.tensor(row, dtype=torch.int32)
        return cdf.to(pmf.device)

    def _check_cdf_size(self):
        if self._quantized_cdf.numel() == 0:
            raise ValueError("Uninitialized CDFs. Run update() first")
        if len(self._quantized_cdf.size()) != 2:
            raise ValueError(f"Invalid CDF size {self._quantized_cdf.size()}")

    def _check_offsets_size(self):
        if self._offset.numel() == 0:
            raise ValueError("Uninitialized offsets. Run update() first")
        if len(self._offset.size()) != 1:
            raise ValueError(f"Invalid offsets size {self._offset.size()}")

    def _check_cdf_length(self):
        if self._cdf_length.numel() == 0:
            raise ValueError("Uninitialized CDF lengths. Run update() first")
        if len(self._cdf_length.size()) != 1:
            raise ValueError(f"Invalid offsets size {self._cdf_length.size()}")

    def _tables(self):
        """host copies of the coder tables, cached until the buffers change"""
        from .ans import _Tables
        # keyed on a counter that update() / load_state_dict() bump, plus the buffers' identity: the caching allocator can
        # hand a freed block back, so (data_ptr, _version, shape) alone would serve stale tables after update(force=True)
        key = (getattr(self, "_tab_gen", 0), self._quantized_cdf.data_ptr(), self._quantized_cdf._version,
               tuple(self._quantized_cdf.shape), self._offset.data_ptr(), self._cdf_length.data_ptr())
        if getattr(self, "_tab_key", None) != key:
            self._tab = _Tables(self._quantized_cdf.detach().cpu().numpy(), self._cdf_length.detach().cpu().numpy(),
                                self._offset.detach().cpu().numpy())
            self._tab_key = key
        return self._tab

    # ---- coding (entropy_models.py:200-290)
    def compress(self, inputs, indexes, means=None, flag=1):
        symbols = self.quantize(inputs, "symbols", means)
        if len(inputs.size()) < 2:
            raise ValueError("Invalid `inputs` size. Expected a tensor with at least 2 dimensions.")
        if inputs.size() != indexes.size():
            raise ValueError("`inputs` and `indexes` should have the same size.")
        self._check_cdf_size()
        self._check_cdf_length()
        self._check_offsets_size()
        from .ans import _encode
        t = self._tables()
        sym = symbols.detach().cpu().numpy().astype(np.int32)
        idx = indexes.detach().cpu().numpy().astype(np.int32)
        return [_encode(np.ascontiguousarray(sym[i].reshape(-1)), np.ascontiguousarray(idx[i].reshape(-1)), t)
                for i in range(sym.shape[0])]

    def decompress(self, strings, indexes, means=None, flag=1):
        if not isinstance(strings, (tuple, list)):
            raise ValueError("Invalid `strings` parameter type.")
        if not len(strings) == indexes.size(0):
            raise ValueError("Invalid strings or indexes parameters")
        if len(indexes.size()) < 2:
            raise ValueError("Invalid `indexes` size. Expected a tensor with at least 2 dimensions.")
        self._check_cdf_size()
        self._check_cdf_length()
        self._check_offsets_size()
        if means is not None:
            if means.size()[:2] != indexes.size()[:2]:
                raise ValueError("Invalid means or indexes parameters")
            if means.size() != indexes.size():
                for i in range(2, len(indexes.size())):
                    if means.size(i) != 1:
                        raise ValueError("Invalid means parameters")
        from .ans import RansDecoder
        t = self._tables()
        idx = indexes.detach().cpu().numpy().astype(np.int32)
        out = np.empty(idx.shape, dtype=np.int32)
        dec = RansDecoder()
        for i, sbytes in enumerate(strings):
            dec.set_stream(sbytes)
            out[i] = dec.decode_stream_np(np.ascontiguousarray(idx[i].reshape(-1)), t).reshape(idx[i].shape)
        sym = torch.from_numpy(out).to(self._quantized_cdf.device)
        return self.dequantize(sym, means)

    def _invalidate_tables(self):
        """drop the host copies of the coder tables (called by every update() and by load_state_dict)"""
        self._tab_gen = getattr(self, "_tab_gen", 0) + 1
        self._tab_key = None
        self._tab = None

    def _load_from_state_dict(self, *args, **kwargs):
        self._invalidate_tables()
        return super()._load_from_state_dict(*args, **kwargs)

    def update(self, *a, **k):
        raise NotImplementedError()


def ptr_any(t):
    return 0 if t is None else t.data_ptr()


class EntropyBottleneck(EntropyModel):
    """entropy_models.py:293-489.  forward(x) -> (x_tilde, likelihood)."""

    def __init__(self, channels: int, *args: Any, tail_mass: float = 1e-9, init_scale: float = 10,
                 filters: Tuple[int, ...] = (3, 3, 3, 3), **kwargs: Any):
        super().__init__(*args, **kwargs)
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        if self.filters != (3, 3, 3, 3):
            raise NotImplementedError("icm EntropyBottleneck: the kernel is specialised for filters=(3,3,3,3)")
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        filters = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        channels = self.channels
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / filters[i + 1]))
            matrix = torch.Tensor(channels, filters[i + 1], filters[i])
            matrix.data.fill_(init)
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(matrix))
            bias = torch.Tensor(channels, filters[i + 1], 1)
            nn.init.uniform_(bias, -0.5, 0.5)
            self.register_parameter(f"_bias{i:d}", nn.Parameter(bias))
            if i < len(self.filters):
                factor = torch.Tensor(channels, filters[i + 1], 1)
                nn.init.zeros_(factor)
                self.register_parameter(f"_factor{i:d}", nn.Parameter(factor))
        self.quantiles = nn.Parameter(torch.Tensor(channels, 1, 3))
        init = torch.Tensor([-self.init_scale, 0, self.init_scale])
        self.quantiles.data = init.repeat(self.quantiles.size(0), 1, 1)
        target = np.log(2 / self.tail_mass - 1)
        self._target = float(target)
        self.register_buffer("target", torch.Tensor([-target, 0, target]))

    def _get_medians(self):
        return self.quantiles[:, :, 1:2]

    def forward(self, x, training: Optional[bool] = None):
        if training is None:
            training = self.training
        noise = self._noise_like(x) if training else None
        lb = self._lik_bound if self.use_likelihood_bound else 0.0

        def f(tape, P, t):
            zt, lik = E.eb_likelihood(tape, t, {"e." + k: p for k, p in P.items()}, "e", noise, lb, want_zt=True)
            return (zt, lik)
        return run_module(self, f, x.contiguous())

    def loss(self):
        """aux loss (entropy_models.py:395-398): gradient flows to ``quantiles`` only."""
        return _EbAux.apply(self, self.quantiles)

    def update(self, force: bool = False) -> bool:
        """entropy_models.py:354-393: per-channel offsets and quantised CDFs of the learned density"""
        if self._offset.numel() > 0 and not force:
            return False
        offset, pmf, tail, pmf_length, max_length = self._pmf_tables()
        self._offset = offset
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail, pmf_length, max_length)
        self._cdf_length = pmf_length + 2
        self._invalidate_tables()
        return True

    def _pmf_tables(self):
        """device side of update(): (offset [C] int32, pmf [C, L], tail_mass [C, 1], pmf_length [C] int32, L)"""
        dev = self.quantiles.device
        Cc = self.channels
        st = L.stream()
        q = self.quantiles.detach().contiguous()
        minima = torch.empty(Cc, dtype=torch.int32, device=dev)
        maxima = torch.empty(Cc, dtype=torch.int32, device=dev)
        check(L.lib().icm_eb_table_bounds(ptr(q), Cc, minima.data_ptr(), maxima.data_ptr(), st), "eb_table_bounds")
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max().item())
        P = {"e." + k: p.detach().contiguous() for k, p in self.named_parameters()}
        prm = E._eb_params(P, "e")
        pmf = torch.empty((Cc, max_length), dtype=torch.float32, device=dev)
        tail = torch.empty(Cc, dtype=torch.float32, device=dev)
        check(L.lib().icm_eb_pmf_table(C.byref(prm), minima.data_ptr(), Cc, max_length, ptr(pmf), ptr(tail), st),
              "eb_pmf_table")
        return -minima, pmf, tail.reshape(Cc, 1), pmf_length, max_length

    @staticmethod
    def _build_indexes(size):
        """channel index of every element (entropy_models.py:491-502)"""
        N, Cc = size[0], size[1]
        view = [1] * len(size)
        view[1] = -1
        return torch.arange(Cc, dtype=torch.int32).view(*view).repeat(N, 1, *size[2:])

    @staticmethod
    def _extend_ndims(tensor, n):
        return tensor.reshape(-1, *([1] * n)) if n > 0 else tensor.reshape(-1)

    def compress(self, x):
        indexes = self._build_indexes(x.size())
        spatial_dims = len(x.size()) - 2
        medians = self._extend_ndims(self._get_medians().detach(), spatial_dims)
        medians = medians.expand(x.size(0), *([-1] * (spatial_dims + 1)))
        return super().compress(x, indexes, medians, 0)

    def decompress(self, strings, size):
        output_size = (len(strings), self._quantized_cdf.size(0), *size)
        indexes = self._build_indexes(output_size)
        medians = self._extend_ndims(self._get_medians().detach(), len(size))
        medians = medians.expand(len(strings), *([-1] * (len(size) + 1)))
        return super().decompress(strings, indexes, medians, 0)


class _EbAux(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, quantiles):
        P = {"e." + k: p.detach() for k, p in mod.named_parameters()}
        prm = E._eb_params(P, "e")
        loss = torch.empty(1, dtype=torch.float32, device=quantiles.device)
        dq = torch.empty_like(quantiles)
        check(L.lib().icm_eb_aux_loss(C.byref(prm), ptr(loss), ptr(dq), mod.channels, mod._target, L.stream()), "eb_aux")
        ctx.save_for_backward(dq)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dq,) = ctx.saved_tensors
        return None, dq * g


class _QuantizeOut(torch.autograd.Function):
    """EntropyModel.quantize in its two differentiable modes (entropy_models.py:126-150) on the HIP kernels:
    noise given -> inputs + noise (identity gradient to inputs); else round(inputs - means) + means, whose gradient is
    zero for inputs (torch.round) and one for means (added back after the round)."""

    @staticmethod
    def forward(ctx, mod, inputs, means, noise):
        ctx.noise_mode = noise is not None
        ctx.bshape = means.shape
        x = inputs.to(torch.float32).contiguous()
        if noise is not None:
            out = x.clone()
            check(L.lib().icm_add_grad(ptr(noise.contiguous()), 0, ptr(out), out.numel(), 1, L.stream()), "add noise")
            return out
        with torch.no_grad():
            return mod.quantize(x, "dequantize", means)

    @staticmethod
    def backward(ctx, g):
        if ctx.noise_mode:
            return None, g, None, None
        gm = g
        if tuple(ctx.bshape) != tuple(g.shape):   # broadcast means: sum the gradient over the broadcast dims
            gm = g.sum_to_size(ctx.bshape)
        return None, None, gm, None


class GaussianConditional(EntropyModel):
    """entropy_models.py:525-659.  forward(inputs, scales, means) -> (outputs, likelihood)."""

    def __init__(self, scale_table: Optional[Union[List, Tuple]], *args: Any, scale_bound: float = 0.11,
                 tail_mass: float = 1e-9, **kwargs: Any):
        super().__init__(*args, **kwargs)
        if not isinstance(scale_table, (type(None), list, tuple)):
            raise ValueError(f'Invalid type for scale_table "{type(scale_table)}"')
        if isinstance(scale_table, (list, tuple)) and len(scale_table) < 1:
            raise ValueError(f'Invalid scale_table length "{len(scale_table)}"')
        if scale_table and (scale_table != sorted(scale_table) or any(s <= 0 for s in scale_table)):
            raise ValueError(f'Invalid scale_table "({scale_table})"')
        self.tail_mass = float(tail_mass)
        if scale_bound is None and scale_table:
            scale_bound = scale_table[0]
        if scale_bound <= 0:
            raise ValueError("Invalid parameters")
        self._scale_bound = float(scale_bound)
        self.lower_bound_scale = LowerBound(scale_bound)
        self.register_buffer("scale_table", torch.Tensor(tuple(float(s) for s in scale_table)) if scale_table
                             else torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]) if scale_bound is not None else None)

    @staticmethod
    def _prepare_scale_table(scale_table):
        return torch.Tensor(tuple(float(s) for s in scale_table))

    @staticmethod
    def _standardized_quantile(quantile):
        import scipy.stats
        return scipy.stats.norm.ppf(quantile)

    def update_scale_table(self, scale_table, force=False):
        """entropy_models.py:585-596"""
        if self._offset.numel() > 0 and not force:
            return False
        device = self.scale_table.device
        self.scale_table = self._prepare_scale_table(scale_table).to(device)
        self.update()
        return True

    def update(self):
        """entropy_models.py:598-624: one quantised Gaussian CDF per entry of the scale table"""
        offset, pmf, tail, pmf_length, max_length = self._pmf_tables()
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail, pmf_length, max_length)
        self._offset = offset
        self._cdf_length = pmf_length + 2
        self._invalidate_tables()

    def _pmf_tables(self):
        """device side of update(): (offset [ns] int32, pmf [ns, L], tail_mass [ns, 1], pmf_length [ns] int32, L)"""
        dev = self.scale_table.device
        ns = int(self.scale_table.numel())
        if ns == 0:
            raise ValueError("empty scale_table: call update_scale_table() first")
        st = L.stream()
        multiplier = float(-self._standardized_quantile(self.tail_mass / 2))
        table = self.scale_table.detach().to(torch.float32).contiguous()
        centers = torch.empty(ns, dtype=torch.int32, device=dev)
        check(L.lib().icm_gc_table_centers(ptr(table), ns, multiplier, centers.data_ptr(), st), "gc_table_centers")
        pmf_length = 2 * centers + 1
        max_length = int(pmf_length.max().item())
        pmf = torch.empty((ns, max_length), dtype=torch.float32, device=dev)
        tail = torch.empty(ns, dtype=torch.float32, device=dev)
        check(L.lib().icm_gc_pmf_table(ptr(table), centers.data_ptr(), ns, max_length, ptr(pmf), ptr(tail), st),
              "gc_pmf_table")
        return -centers, pmf, tail.reshape(ns, 1), pmf_length, max_length

    def build_indexes(self, scales):
        """index of the first table entry >= max(scale, bound) (entropy_models.py:661-666)"""
        sc = scales.to(torch.float32).contiguous()
        N, Cc = sc.shape[0], sc.shape[1]
        HW = max(1, sc[0, 0].numel())
        idx = torch.empty(sc.shape, dtype=torch.int32, device=sc.device)
        table = self.scale_table.detach().to(sc.device, torch.float32).contiguous()
        check(L.lib().icm_gc_build_indexes(ptr(sc), Cc * HW, ptr(table), int(table.numel()), self._scale_bound,
                                           idx.data_ptr(), N, Cc, HW, L.stream()), "gc_build_indexes")
        return idx

    def _likelihood(self, inputs, scales, means=None):
        """entropy_models.py:626-643 (no likelihood bound, no quantisation): the fused kernel in eval mode on integer-
        offset inputs reproduces it; exposed for API parity"""
        if means is None:
            means = torch.zeros_like(inputs)
        lik = E.new(inputs)
        tape = E.Tape(need_grad=False)
        # the kernel quantises round(x - mu) + mu: callers of _likelihood pass already-quantised inputs
        E.gc_likelihood_ste(tape, inputs.contiguous(), means.contiguous(), scales.contiguous(), None, lik, None, None,
                            self._scale_bound, 0.0)
        return lik

    def forward(self, inputs, scales, means=None, training: Optional[bool] = None):
        if training is None:
            training = self.training
        if means is None:
            means = torch.zeros_like(inputs)
        noise = self._noise_like(inputs) if training else None
        lb = self._lik_bound if self.use_likelihood_bound else 0.0
        sb = self._scale_bound

        def f(tape, y, sc, mu):
            lik = E.new(y)
            E.gc_likelihood_ste(tape, y, mu, sc, noise, lik, None, None, sb, lb)
            return (lik,)
        (lik,) = E.tape_function(f, [inputs.contiguous(), scales.contiguous(), means.contiguous()])
        # entropy_models.py:126-150,655: outputs = quantize(inputs, "noise" | "dequantize", means) WITH its gradient:
        # d/d inputs = 1 (noise) or 0 (round), d/d means = 0 (noise) or 1 (dequantize adds the means back after the round)
        outputs = _QuantizeOut.apply(self, inputs, means, noise)
        return outputs, lik
