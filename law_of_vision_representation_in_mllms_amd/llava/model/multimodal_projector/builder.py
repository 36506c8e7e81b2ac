"""mm_projector on MI355X — same factory name / config keys / errors as
llava/model/multimodal_projector/builder.py:34-59 (`linear`, `mlp{N}x_gelu`, `identity`).

The returned module keeps nn.Linear-compatible parameter names (`0.weight`, `0.bias`, `2.weight`, ... for the MLP), so the
reference's `mm_projector.load_state_dict(get_w(weights, 'mm_projector'))` (llava_arch.py:183-189) works unchanged;
forward runs the bf16 MFMA GEMM with fused bias + exact-erf GELU epilogue.
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

    def _pack(self, device):
        lin = [m for m in self if isinstance(m, nn.Linear)]
        key = tuple((m.weight.data_ptr(), m.weight._version) for m in lin) + (str(device),)
        if self._packed is None or self._packed[0] != key:
            ws = [m.weight.detach().to(device=device, dtype=torch.bfloat16).contiguous() for m in lin]
            bs = [None if m.bias is None else m.bias.detach().to(device=device, dtype=torch.float32).contiguous() for m in lin]
            self._packed = (key, ws, bs)
        return self._packed[1], self._packed[2]

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            _lib.require_gpu()
            x = x.cuda()
        ws, bs = self._pack(x.device)
        shp = x.shape
        for w in ws:
            if w.shape[0] % 64 or w.shape[1] % 64:
                raise ValueError("projector widths must be multiples of 64 for the MFMA GEMM")
        h = x.reshape(-1, shp[-1]).to(torch.bfloat16).contiguous()
        n = len(ws)
        for i, (w, b) in enumerate(zip(ws, bs)):
            last = i == n - 1
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
