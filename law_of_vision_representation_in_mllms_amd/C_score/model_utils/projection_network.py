"""Post-processors of the C score on MI355X - drop-ins for C_score/model_utils/projection_network.py.

DummyAggregationNetwork (:7-13) is the zero-shot identity (`batch * 1.0`) every score of the paper uses.  AggregationNetwork (:15-125) is
GeoAware-SC's supervised post-processor: the feature map's channel groups (`feature_dims`) each go through ONE detectron2-style
BottleneckBlock (model_utils/resnet.py:174-286: 1x1 conv -> GN -> ReLU -> 3x3 conv -> GN -> ReLU -> 1x1 conv -> GN, projection shortcut
1x1 conv -> GN, add, ReLU; no conv biases; GN with `num_norm_groups` groups), the results are mixed with softmax(mixing_weights).
Parameter names follow the reference's state_dict (`bottleneck_layers.{l}.0.{conv1,conv2,conv3,shortcut}.{weight,norm.weight,norm.bias}`,
`mixing_weights`, `logit_scale`, `self_logit_scale`), so `load_pretrained_weights` / `load_state_dict` of a GeoAware-SC checkpoint work.
forward runs in fp32 (the reference's dtype) on the exact-fp32 MFMA path: visrep_gemm_f32 for the 1x1 convolutions, visrep_im2col3x3_f32 +
visrep_gemm_f32 for the 3x3 one, visrep_groupnorm_f32 for GroupNorm fused with the ReLU / shortcut add / mixing weight.  Inference only
(the training half of pck_train.py is not built): dropout, the pose / position embeddings and `last_layer` (all None / off in the
reference's evaluation) are not modelled.
"""
import numpy as np
import torch
import torch.nn as nn


class DummyAggregationNetwork(nn.Module):
    def forward(self, batch):
        return batch * 1.0


class _ConvGN(nn.Module):
    """bias-free Conv2d followed by GroupNorm (`.weight`, `.norm.weight`, `.norm.bias` as in resnet.py's Conv2d wrapper)."""

    def __init__(self, cin, cout, k, groups):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")      # fvcore c2_msra_fill
        self.norm = nn.GroupNorm(groups, cout)
        self.k = k


class _Bottleneck(nn.Module):
    def __init__(self, cin, cout, mid, groups, kernel_size):
        super().__init__()
        self.shortcut = _ConvGN(cin, cout, 1, groups) if cin != cout else None
        self.conv1 = _ConvGN(cin, mid, kernel_size[0], groups)
        self.conv2 = _ConvGN(mid, mid, kernel_size[1], groups)
        self.conv3 = _ConvGN(mid, cout, kernel_size[2], groups)


class AggregationNetwork(nn.Module):
    def __init__(self, device="cuda", feature_dims=[640, 1280, 1280, 768], projection_dim=384, num_norm_groups=32, save_timestep=[1],
                 kernel_size=[1, 3, 1], contrastive_temp=10, feat_map_dropout=0.0):
        super().__init__()
        if list(kernel_size) != [1, 3, 1]:
            raise NotImplementedError("AggregationNetwork: kernel_size [1, 3, 1] (the reference's default) is what is built")
        self.feature_dims = list(feature_dims)
        self.projection_dim, self.num_norm_groups = projection_dim, num_norm_groups
        self.save_timestep = save_timestep
        self.feat_map_dropout = feat_map_dropout
        self.device = device
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.self_logit_scale = nn.Parameter(torch.ones([]) * np.log(contrastive_temp))
        self.bottleneck_layers = nn.ModuleList(
            nn.Sequential(_Bottleneck(d, projection_dim, projection_dim // 4, num_norm_groups, kernel_size)) for d in self.feature_dims)
        self.mixing_weights = nn.Parameter(torch.ones(len(self.feature_dims) * len(save_timestep)))
        self.mixing_weights_names = [f"timestep-{save_timestep}_layer-{l + 1}" for l in range(len(self.feature_dims)) for _ in save_timestep]
        self._packed = None

    def load_pretrained_weights(self, pretrained_dict):
        """projection_network.py:70-87: the first four mixing weights from the checkpoint, every other matching key as it is."""
        custom = self.state_dict()
        if 'mixing_weights' in custom and 'mixing_weights' in pretrained_dict:
            if custom['mixing_weights'].shape != pretrained_dict['mixing_weights'].shape:
                custom['mixing_weights'][:4] = pretrained_dict['mixing_weights'][:4]
                custom['mixing_weights'][4] = torch.zeros_like(custom['mixing_weights'][4])
            else:
                custom['mixing_weights'][:4] = pretrained_dict['mixing_weights'][:4]
        custom.update({k: v for k, v in pretrained_dict.items() if k in custom and k != 'mixing_weights'})
        self.load_state_dict(custom, strict=False)

    def _pack(self, dev):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) + (str(dev),)
        if self._packed is None or self._packed[0] != key:
            f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
            blocks = []
            for seq in self.bottleneck_layers:
                b = seq[0]
                ent = {}
                for name in ("shortcut", "conv1", "conv2", "conv3"):
                    m = getattr(b, name)
                    if m is None:
                        ent[name] = None
                        continue
                    w = m.weight.detach()
                    w2 = w.reshape(w.shape[0], -1) if m.k == 1 else w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)   # 3x3: column order (ky, kx, c)
                    ent[name] = (f(w2), f(m.norm.weight), f(m.norm.bias), float(m.norm.eps))
                blocks.append(ent)
            self._packed = (key, blocks)
        return self._packed[1]

    @torch.no_grad()
    def forward(self, batch, pose=None):
        """batch [B, sum(feature_dims) * len(save_timestep), H, W] -> [B, projection_dim, H, W] fp32 (projection_network.py:89-125)."""
        from ... import _lib, engine
        lib = _lib.require_gpu()
        dev = batch.device if batch.is_cuda else torch.device("cuda")
        B, Ctot, H, W = batch.shape
        HW, G, D = H * W, self.num_norm_groups, self.projection_dim
        tokens = batch.to(device=dev, dtype=torch.float32).permute(0, 2, 3, 1).reshape(B * HW, Ctot).contiguous()   # channels-last
        mix = torch.softmax(self.mixing_weights.detach().float(), dim=0).tolist()
        blocks = self._pack(dev)
        out = torch.zeros(B * HW, D, dtype=torch.float32, device=dev)
        sp = _lib.stream_ptr

        def gn(x, p, y, relu, resid=None, alpha=1.0, accumulate=False):
            _lib.check(lib.visrep_groupnorm_f32(_lib.ptr(x), _lib.ptr(p[1]), _lib.ptr(p[2]), _lib.ptr(resid), _lib.ptr(y), B, HW, x.shape[1], G, p[3],
                                                int(relu), float(alpha), int(accumulate), sp()), "visrep_groupnorm_f32")
            return y
        start = 0
        for i, w in enumerate(mix):
            l = i % len(self.feature_dims)
            blk, cin = blocks[l], self.feature_dims[l]
            if cin % 4 or (D // 4) % 4:
                raise ValueError("AggregationNetwork: channel counts must be multiples of 4")
            feats = tokens[:, start:start + cin]                                     # a column slice: ld = Ctot, rows 16-byte aligned
            start += cin
            h1 = engine.gemm_f32(feats, blk["conv1"][0])
            gn(h1, blk["conv1"], h1, True)
            cols = torch.empty(B * HW, 9 * h1.shape[1], dtype=torch.float32, device=dev)
            _lib.check(lib.visrep_im2col3x3_f32(_lib.ptr(h1), _lib.ptr(cols), B, H, W, h1.shape[1], sp()), "visrep_im2col3x3_f32")
            h2 = engine.gemm_f32(cols, blk["conv2"][0])
            gn(h2, blk["conv2"], h2, True)
            h3 = engine.gemm_f32(h2, blk["conv3"][0])
            if blk["shortcut"] is not None:
                sc = engine.gemm_f32(feats, blk["shortcut"][0])
                gn(sc, blk["shortcut"], sc, False)
            else:
                sc = feats.contiguous()
            gn(h3, blk["conv3"], out, True, resid=sc, alpha=w, accumulate=True)       # out += w * relu(GN(conv3) + shortcut)
        return out.view(B, H, W, D).permute(0, 3, 1, 2).contiguous()
