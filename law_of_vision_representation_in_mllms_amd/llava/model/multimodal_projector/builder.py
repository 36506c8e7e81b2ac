"""mm_projector on MI355X — same factory name / config keys / errors as
llava/model/multimodal_projector/builder.py:34-59 (`linear`, `mlp{N}x_gelu`, `identity`).

The returned module keeps nn.Linear-compatible parameter names (`0.weight`, `0.bias`, `2.weight`, ... for the MLP), so the
reference's `mm_projector.load_state_dict(get_w(weights, 'mm_projector'))` (llava_arch.py:183-189) works unchanged.
Like an nn.Sequential, forward computes in the dtype of its parameters: bf16 / fp16 parameters (what LLaVA has after
`model.to(bfloat16)`) run the bf16 MFMA GEMM with fused bias + exact-erf GELU epilogue; fp32 parameters (a freshly built module)
run the exact-fp32 MFMA path (csrc/f32ops.hip) - the reference-precision mode used for end-to-end 1e-4 score parity.
"""
import re

import torch
import torch.nn as nn

from .... import _lib, engine


class IdentityMap(nn.Module):
    def forward(self, x, *args, **kwargs):
        return x

    @property
    def config(self):
        return {"mm_projector_type": 'identity'}


class HipMLPProjector(nn.Sequential):
    """Linear -> [GELU -> Linear]*(depth-1); parameters live in ordinary nn.Linear children (indices 0, 2, 4, ...)."""

    def __init__(self, in_features, hidden, depth):
        mods = [nn.Linear(in_features, hidden)]
        for _ in range(1, depth):
            mods.append(nn.GELU())
            mods.append(nn.Linear(hidden, hidden))
        super().__init__(*mods)
        self._packed = None

    def _pack(self, device, dtype):
        lin = [m for m in self if isinstance(m, nn.Linear)]
        key = tuple((m.weight.data_ptr(), m.weight._version) for m in lin) + (str(device), str(dtype))
        if self._packed is None or self._packed[0] != key:
            ws = [m.weight.detach().to(device=device, dtype=dtype).contiguous() for m in lin]
            bs = [None if m.bias is None else m.bias.detach().to(device=device, dtype=torch.float32).contiguous() for m in lin]
            self._packed = (key, ws, bs)
        return self._packed[1], self._packed[2]

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            _lib.require_gpu()
            x = x.cuda()
        fp32 = next(self.parameters()).dtype == torch.float32
        ws, bs = self._pack(x.device, torch.float32 if fp32 else torch.bfloat16)
        shp = x.shape
        mult = 4 if fp32 else 64
        for w in ws:
            if w.shape[0] % mult or w.shape[1] % mult:
                raise ValueError(f"projector widths must be multiples of {mult} for the {'fp32' if fp32 else 'bf16'} MFMA GEMM")
        h = x.reshape(-1, shp[-1]).to(torch.float32 if fp32 else torch.bfloat16).contiguous()
        n = len(ws)
        for i, (w, b) in enumerate(zip(ws, bs)):
            last = i == n - 1
            if fp32:
                h = engine.gemm_f32(h, w, b, _lib.EPI_BIAS if last else _lib.EPI_ACT, act="none" if last else "gelu")
            else:
                h = engine.gemm(h, w, b, _lib.EPI_BIAS if last else _lib.EPI_ACT, act="none" if last else "gelu")
        return h.reshape(*shp[:-1], ws[-1].shape[0]).to(x.dtype)


def build_vision_projector(config, delay_load=False, **kwargs):
    projector_type = getattr(config, 'mm_projector_type', 'linear')

    if projector_type == 'linear':
        return HipMLPProjector(config.mm_hidden_size, config.hidden_size, 1)

    mlp_gelu_match = re.match(r'^mlp(\d+)x_gelu$', projector_type)
    if mlp_gelu_match:
        return HipMLPProjector(config.mm_hidden_size, config.hidden_size, int(mlp_gelu_match.group(1)))

    if projector_type == 'identity':
        return IdentityMap()

    if re.match(r'^perceiver(\d+)x$', projector_type):
        raise NotImplementedError("perceiver resampler projector is outside the scoring hot path (SURVEY.md §2.1)")

    raise ValueError(f'Unknown projector type: {projector_type}')
