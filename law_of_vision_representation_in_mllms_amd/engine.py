"""Device-side tower engine: packed ViT weights resident in HBM + the composed HIP forward (visrep_vit_forward).

PyTorch is used for device memory and streams only; every FLOP of the forward runs in libvisrep_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib
from .vit_weights import ViTSpec


def _round_up(v: int, a: int) -> int:
    return (v + a - 1) // a * a


_SCRATCH = {}


def ensure_scratch(device, nbytes: int = 64 << 20) -> None:
    """One split-K scratch PER DEVICE for visrep_gemm_bf16 (visrep_set_scratch registers it for the current device): few-tile
    problems - the 128x128 tail launches of the ViT GEMMs, the low-resolution convolutions of the diffusion towers - split their K
    loop over idle CUs.  S * M * N * 4 bytes <= 33.6 MB by construction of the split rule (S * tiles <= 2 * 256 block slots,
    128x128 tiles).  The planes are not keyed by stream: one stream per device runs GEMMs at a time (every engine here does)."""
    device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _SCRATCH:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        with torch.cuda.device(idx):
            _lib.check(_lib.require_gpu().visrep_set_scratch(_lib.ptr(buf), nbytes), "visrep_set_scratch")
        _SCRATCH[idx] = buf


class VitEngine:
    """One ViT tower on one GPU.

    forward(pixels, n_layers) returns hidden_states[n_layers] as a bf16 tensor [B, tokens, d]
    (index 0 = embeddings (+pre-LN for CLIP), index i = output of encoder layer i — the HF
    `output_hidden_states=True` convention the reference towers index with `select_layer`).
    """

    def __init__(self, spec: ViTSpec, weights: dict, device: Optional[torch.device] = None, fuse_ln: Optional[bool] = None):
        """fuse_ln (default on; VISREP_FUSE_LN=0 turns it off): fold every block's LayerNorm into the GEMMs that consume it -
        gamma into the weight rows, beta into the bias, the per-row mean / rstd into the GEMM epilogue (ln_rt / ln_s) - so the
        normalised activations are never written to HBM, and take the row statistics from the epilogue of the residual GEMM that
        produced the rows (visrep_gemm_bf16_resid_stats) - so LayerNorm never reads the residual stream either; only layer 0's
        first LayerNorm runs the read-only statistics pass.  Measured on ViT-L/14-336, batch 256: the two LayerNorm passes per
        layer (19.7 ms of kernel time per 4 forwards) are replaced by 15.2 ms of heavier epilogues + partial-sum reduction:
        +0.8 % kernel time, +1.4 % on the bench line (profiles/round1_fold_stats.md)."""
        self.lib = _lib.require_gpu()
        import os
        self.fuse_ln = (os.environ.get("VISREP_FUSE_LN", "1") != "0") if fuse_ln is None else bool(fuse_ln)
        self.spec = spec
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if spec.d != spec.heads * 64:
            raise ValueError("HIP attention kernel supports head_dim 64 only")
        if spec.d % 128 or spec.mlp % 128:
            raise ValueError("d and mlp must be multiples of 128")
        ensure_scratch(self.device)
        self.kpad = _round_up(3 * spec.patch * spec.patch, 64)
        self._keep = []          # device tensors referenced by raw pointers
        dev = self.device

        def mat(t):              # bf16 [out, in]
            x = t.detach().to(device=dev, dtype=torch.bfloat16).contiguous()
            self._keep.append(x)
            return x

        def vec(t):              # fp32 vector
            if t is None:
                return None
            x = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            self._keep.append(x)
            return x

        pw = torch.zeros(spec.d, self.kpad, dtype=torch.float32)
        pw[:, : 3 * spec.patch * spec.patch] = weights["patch_w"].float()
        if weights["pos"].shape[0] != spec.tokens:
            raise ValueError(f"position embedding has {weights['pos'].shape[0]} rows, spec wants {spec.tokens}")
        self._patch_w = mat(pw)
        self._vecs = {k: vec(weights.get(k)) for k in ("patch_b", "cls", "pos", "pre_ln_g", "pre_ln_b")}
        n = len(weights["layers"])
        self.n_layers = n
        self._layers = (_lib.VitLayer * max(n, 1))()
        # softmax scale in the exponent's base, folded into the Q rows of the Q|K|V projection (attn_fwd PS): q' = c * q with
        # c = head_dim^-0.5 * log2(e), so that 2^(q'.k) = e^(q.k / sqrt(head_dim)).  VISREP_Q_PRESCALE=0 keeps the scale in the kernel.
        # VISREP_Q_PRESCALE=2 (default) additionally lets towers with a CLS token and 64 n patches run the image-aligned attention kernel.
        self._q_mode = int(os.environ.get("VISREP_Q_PRESCALE", "2"))
        self.q_prescaled = self._q_mode != 0
        qc = torch.ones(3 * spec.d)
        if self.q_prescaled:
            qc[: spec.d] = 0.125 * 1.4426950408889634

        def fold(Wm, bias, g, b, rowscale=None):
            """Linear(LayerNorm(x)) = rstd * (x (g o W)^T - mean * s) + (W b + bias): (g o W in bf16, s over the ROUNDED rows, b').
            rowscale (the Q rows' softmax scale) multiplies the fp32 product before its single rounding to bf16."""
            Wb = Wm.detach().float().to(torch.bfloat16).float()               # the weights the unfused path multiplies with
            rs = torch.ones(Wb.shape[0]) if rowscale is None else rowscale
            Wf = (Wb * g.detach().float()[None] * rs[:, None]).to(torch.bfloat16)
            shift = Wb @ b.detach().float()                                  # beta pushed through the linear map
            return Wf, Wf.float().sum(1), (shift if bias is None else shift + bias.detach().float()) * rs

        for i, L in enumerate(weights["layers"]):
            ent = self._layers[i]
            L = dict(L)
            extra = {"sqkv": None, "s1": None}
            if self.fuse_ln:
                L["wqkv"], extra["sqkv"], L["bqkv"] = fold(L["wqkv"], L.get("bqkv"), L["ln1_g"], L["ln1_b"], qc)
                L["w1"], extra["s1"], L["b1"] = fold(L["w1"], L.get("b1"), L["ln2_g"], L["ln2_b"])
            elif self.q_prescaled:                                             # unfused path: scale the (bf16-rounded) rows, round once more
                L["wqkv"] = L["wqkv"].detach().float().to(torch.bfloat16).float() * qc[:, None]
                if L.get("bqkv") is not None:
                    L["bqkv"] = L["bqkv"].detach().float() * qc
            for k in ("wqkv", "wo", "w1", "w2"):
                setattr(ent, k, mat(L[k]).data_ptr())
            for k in ("ln1_g", "ln1_b", "bqkv", "bo", "ls1", "ln2_g", "ln2_b", "b1", "b2", "ls2"):
                v = vec(L.get(k))
                setattr(ent, k, 0 if v is None else v.data_ptr())
            for k, t in extra.items():
                v = vec(t)
                setattr(ent, k, 0 if v is None else v.data_ptr())
        self._w = _lib.VitWeights()
        self._w.patch_w = self._patch_w.data_ptr()
        for k, v in self._vecs.items():
            setattr(self._w, k, 0 if v is None else v.data_ptr())
        self._w.layers = C.cast(self._layers, C.POINTER(_lib.VitLayer))
        self._cfg = _lib.VitConfig(spec.image_size, spec.patch, spec.d, spec.heads, spec.mlp, n, spec.tokens,
                                   int(spec.has_cls), int(spec.pre_ln), _lib.ACT[spec.act], self.kpad, float(spec.eps), self._q_mode)
        self._ws: Dict[int, torch.Tensor] = {}
        self._pinned = set()                 # batch sizes whose workspace a captured HIP graph refers to

    def workspace(self, B: int) -> torch.Tensor:
        ws = self._ws.get(B)
        if ws is None:
            nbytes = self.lib.visrep_vit_workspace_bytes(C.byref(self._cfg), B)
            # a few batch sizes stay resident; a workspace that a HIP graph captured (the image-variation featurizer records this engine's
            # forward inside its own graph) is PINNED: a replayed graph keeps using the pointer it saw, so it is never evicted
            for old in [b for b in self._ws if b not in self._pinned][: max(0, len(self._ws) - len(self._pinned) - 5)]:
                self._ws.pop(old)
            ws = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[B] = ws
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            self._pinned.add(B)
        return ws

    @torch.no_grad()
    def forward(self, pixels: torch.Tensor, n_layers: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        s = self.spec
        if pixels.dim() != 4 or pixels.shape[1] != 3 or pixels.shape[2] != s.image_size or pixels.shape[3] != s.image_size:
            raise ValueError(f"Input image size {tuple(pixels.shape)} doesn't match tower ({s.image_size}*{s.image_size}).")
        n_layers = self.n_layers if n_layers is None else n_layers
        if not 0 <= n_layers <= self.n_layers:
            raise ValueError("n_layers out of range")
        px = pixels.to(self.device)
        if px.dtype not in (torch.float32, torch.bfloat16):
            px = px.float()
        px = px.contiguous()
        B = px.shape[0]
        if out is None:
            out = torch.empty(B, s.tokens, s.d, dtype=torch.bfloat16, device=self.device)
        ws = self.workspace(B)
        with torch.cuda.device(self.device):
            rc = self.lib.visrep_vit_forward(C.byref(self._cfg), C.byref(self._w), _lib.ptr(px),
                                             _lib.F32 if px.dtype == torch.float32 else _lib.BF16,
                                             _lib.ptr(out), B, n_layers, _lib.ptr(ws), _lib.stream_ptr())
        _lib.check(rc, "visrep_vit_forward")
        return out


class VitEngineF32:
    """The same tower in reference precision: fp32 weights, activations and accumulation (csrc/f32ops.hip, visrep_vit_forward_f32).

    The reference's C-score CLIP / OpenCLIP / DINOv2 towers run in fp32 (C_score/extract_feature.py:36-45,49-50: no dtype cast,
    fp32 pixels); this engine is what makes images -> tower -> A / C score comparable with that chain at the 1e-4 bar.  About
    1/6 of the bf16 engine's throughput on the split-bf16 route (1/10 on the exact-fp32 MFMA route, which takes any head width that is a
    multiple of 4).
    forward() returns fp32 [B, tokens, d]; images are processed in chunks so that the fp32 score matrices stay below `max_ws_bytes`."""

    def __init__(self, spec: ViTSpec, weights: dict, device: Optional[torch.device] = None, max_ws_bytes: int = 8 << 30, gemm: str = "auto",
                 products: Optional[int] = None):
        """gemm: 'split' = projections and attention as split-bf16 products on the bf16 matrix pipe (fp32 values as bf16 planes, fp32
        accumulation, visrep_vit_forward_f32_split); 'native' = exact-fp32 MFMA (visrep_vit_forward_f32); 'auto' = split where the tower's
        shapes allow it (d, mlp % 256 == 0, head width 64), VISREP_F32_GEMM in the environment overrides.
        products (split route): 6 = three planes, every plane pair >= 2^-24 of the result (fp32-equivalent, ~1e-7 per product sum);
        4 / 3 = two planes (16 significand bits), all four pairs / without (mid, mid): ~4e-6 per product sum at half the matrix work.
        None = DEFAULT_SPLIT_PRODUCTS = 6, the fp32-equivalent set (3 holds A <= 1e-4 and exact PCK hits on the full-size SYNTHETIC towers,
        profiles/round4_precision.md, and is the sweep's explicit opt-in); VISREP_F32_PRODUCTS in the environment overrides."""
        self.lib = _lib.require_gpu()
        self.spec = spec
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if spec.d % spec.heads or (spec.d // spec.heads) % 4 or spec.mlp % 4:
            raise ValueError("fp32 engine: d / heads and mlp must be multiples of 4")
        self.kpad = _round_up(3 * spec.patch * spec.patch, 4)
        self.max_ws_bytes = int(max_ws_bytes)
        self._keep = []
        dev = self.device

        def f32(t):
            if t is None:
                return None
            x = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            self._keep.append(x)
            return x

        pw = torch.zeros(spec.d, self.kpad, dtype=torch.float32)
        pw[:, : 3 * spec.patch * spec.patch] = weights["patch_w"].float()
        if weights["pos"].shape[0] != spec.tokens:
            raise ValueError(f"position embedding has {weights['pos'].shape[0]} rows, spec wants {spec.tokens}")
        self._patch_w = f32(pw)
        self._vecs = {k: f32(weights.get(k)) for k in ("patch_b", "cls", "pos", "pre_ln_g", "pre_ln_b")}
        n = len(weights["layers"])
        self.n_layers = n
        self._layers = (_lib.VitLayer * max(n, 1))()
        for i, L in enumerate(weights["layers"]):
            ent = self._layers[i]
            for k in ("wqkv", "wo", "w1", "w2", "ln1_g", "ln1_b", "bqkv", "bo", "ls1", "ln2_g", "ln2_b", "b1", "b2", "ls2"):
                v = f32(L.get(k))
                setattr(ent, k, 0 if v is None else v.data_ptr())
            ent.sqkv = 0
            ent.s1 = 0
        self._w = _lib.VitWeights()
        self._w.patch_w = self._patch_w.data_ptr()
        for k, v in self._vecs.items():
            setattr(self._w, k, 0 if v is None else v.data_ptr())
        self._w.layers = C.cast(self._layers, C.POINTER(_lib.VitLayer))
        self._cfg = _lib.VitConfig(spec.image_size, spec.patch, spec.d, spec.heads, spec.mlp, n, spec.tokens,
                                   int(spec.has_cls), int(spec.pre_ln), _lib.ACT[spec.act], self.kpad, float(spec.eps))
        self._ws: Dict[int, torch.Tensor] = {}
        self._pinned = set()                 # batch sizes whose workspace a captured HIP graph refers to
        import os
        gemm = os.environ.get("VISREP_F32_GEMM", gemm)
        if gemm not in ("auto", "split", "native"):
            raise ValueError(f"gemm must be 'auto', 'split' or 'native', got {gemm!r}")
        can = bool(self.lib.visrep_vit_f32_split_supported(C.byref(self._cfg)))
        if gemm == "split" and not can:
            raise ValueError("split-bf16 route needs d and mlp to be multiples of 256 and head width 64")
        self.gemm = "split" if (gemm != "native" and can) else "native"
        products = int(os.environ.get("VISREP_F32_PRODUCTS", products if products is not None else DEFAULT_SPLIT_PRODUCTS))
        if products not in (3, 4, 6):
            raise ValueError(f"products must be 3, 4 or 6, got {products}")
        self.products = products if self.gemm == "split" else None
        self._wsplit = None
        if self.gemm == "split":
            npl = split_planes(products)
            # the four projection matrices of every layer as bf16 planes [N, npl K] (hi | mid [| lo]), split on the device once
            self._split_layers = (_lib.VitLayer * max(n, 1))()
            with torch.cuda.device(self.device):
                for i in range(n):
                    for k in ("wqkv", "wo", "w1", "w2"):
                        wf = self._keep_by_ptr(getattr(self._layers[i], k))
                        planes = torch.empty(wf.shape[0], npl * wf.shape[1], dtype=torch.bfloat16, device=dev)
                        _lib.check(self.lib.visrep_split_bf16_planes(_lib.ptr(wf), wf.stride(0), wf.shape[0], wf.shape[1], npl, _lib.ptr(planes),
                                                                     _lib.stream_ptr()), "visrep_split_bf16_planes")
                        self._keep.append(planes)
                        setattr(self._split_layers[i], k, planes.data_ptr())
            self._wsplit = _lib.VitWeights()
            self._wsplit.layers = C.cast(self._split_layers, C.POINTER(_lib.VitLayer))

    def _keep_by_ptr(self, ptr: int) -> torch.Tensor:
        for t in self._keep:
            if t.data_ptr() == ptr:
                return t
        raise KeyError(ptr)

    def chunk(self) -> int:
        """Images per library call.  Bounded by the workspace budget, then picked for the GEMMs' tile rounds (best_chunk): the persistent
        256 x 256 kernel runs floor(tiles / CUs) rounds + a tail launch for a remainder of up to CUs / 4 tiles (else one more round), and the
        N = d projections have the fewest tiles per row block - 64 images of a 577-token tower are 2.25 rounds of them (3 run), 113 images
        3.98 (4 run)."""
        per = self.lib.visrep_vit_f32_workspace_bytes(C.byref(self._cfg), 1)
        cap = max(1, min(128, self.max_ws_bytes // max(per, 1)))
        import os
        forced = int(os.environ.get("VISREP_F32_CHUNK", "0"))              # A/B knob (tools/)
        return min(forced, cap) if forced > 0 else best_chunk(self.spec.tokens, self.spec.d, cap)

    def workspace(self, B: int) -> torch.Tensor:
        ws = self._ws.get(B)
        if ws is None:
            for old in [b for b in self._ws if b not in self._pinned][: max(0, len(self._ws) - len(self._pinned) - 5)]:
                self._ws.pop(old)
            ws = torch.empty(self.lib.visrep_vit_f32_workspace_bytes(C.byref(self._cfg), B), dtype=torch.uint8, device=self.device)
            self._ws[B] = ws
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            self._pinned.add(B)                 # a captured graph refers to this pointer: never evicted
        return ws

    @torch.no_grad()
    def forward(self, pixels: torch.Tensor, n_layers: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        s = self.spec
        if pixels.dim() != 4 or pixels.shape[1] != 3 or pixels.shape[2] != s.image_size or pixels.shape[3] != s.image_size:
            raise ValueError(f"Input image size {tuple(pixels.shape)} doesn't match tower ({s.image_size}*{s.image_size}).")
        n_layers = self.n_layers if n_layers is None else n_layers
        if not 0 <= n_layers <= self.n_layers:
            raise ValueError("n_layers out of range")
        px = pixels.to(device=self.device, dtype=torch.float32).contiguous()
        B = px.shape[0]
        if out is None:
            out = torch.empty(B, s.tokens, s.d, dtype=torch.float32, device=self.device)
        step = self.chunk()
        with torch.cuda.device(self.device):
            for b0 in range(0, B, step):
                nb = min(step, B - b0)
                ws = self.workspace(nb)
                if self.gemm == "split":
                    rc = self.lib.visrep_vit_forward_f32_split(C.byref(self._cfg), C.byref(self._w), C.byref(self._wsplit), self.products,
                                                               _lib.ptr(px[b0:b0 + nb]), _lib.ptr(out[b0:b0 + nb]), nb, n_layers, _lib.ptr(ws),
                                                               _lib.stream_ptr())
                else:
                    rc = self.lib.visrep_vit_forward_f32(C.byref(self._cfg), C.byref(self._w), _lib.ptr(px[b0:b0 + nb]), _lib.ptr(out[b0:b0 + nb]),
                                                         nb, n_layers, _lib.ptr(ws), _lib.stream_ptr())
                _lib.check(rc, "visrep_vit_forward_f32")
        return out


def best_chunk(tokens: int, d: int, cap: int, cus: Optional[int] = None) -> int:
    """Images per launch (<= cap, >= cap / 2) for which a [count * tokens, d] GEMM wastes the least of the persistent 256 x 256 kernel's tile
    rounds.  The cost model is the GEMM dispatcher's own rule (csrc/gemm_bf16.hip, visrep_gemm_dispatch): T tiles on `cus` CUs run as
    floor(T / cus) full rounds; a remainder goes to the 128 x 128 tail launch when 0 < 4 * rem <= cus and the full rounds end on a row
    boundary - priced here at 0.45 of a round plus twice its share of one (profiles/round4_gemm.md section 5: a tail pair of launches is
    14-18 us beside a 31-us K = 1024 round, and that kernel runs at about half the rate) - and costs a whole extra round otherwise.
    Returns the LARGEST count whose efficiency (tiles / (cus * rounds paid)) is within 1 % of the best in the range.
    cus = None asks the library for the current device's CU count (visrep_device_cu_count; 256 when no device is visible).  On MI355X:
    best_chunk(577, 1024, 128) = 113 (3.98 rounds, 4 paid), best_chunk(257, 1024, 128) = 127 (exactly 2 rounds, no tail launch)."""
    if cus is None:
        cus = 256                                               # no device visible (host-side planning / tests): the MI355X count
        if torch.cuda.is_available():
            cus = int(_lib.load().visrep_device_cu_count())
    ntn = max(1, d // 256)

    def eff(c):
        tiles = -(-c * tokens // 256) * ntn
        rounds, rem = divmod(tiles, cus)
        if rem == 0:
            paid = rounds
        elif rounds >= 1 and rem * 4 <= cus and (rounds * cus) % ntn == 0:
            paid = rounds + 0.45 + 2.0 * rem / cus
        else:
            paid = rounds + 1
        return tiles / (cus * paid)
    lo = max(cap // 2, 1)
    effs = {c: eff(c) for c in range(lo, cap + 1)}
    top = max(effs.values())
    return max(c for c, e in effs.items() if e >= top - 0.01)


class VitEngineCPU:
    """The same tower on HOST cores (visrep_vit_forward_cpu, csrc/host_twins.hip): plain C++ fp32 on host threads.  BASELINE configs[0]
    ("vision_tower feature-extract on 32 COCO images, CPU float32 - plumbing, no GPU") and SURVEY §8b's `*_cpu` twins.  Built ONLY when the
    caller asks for device "cpu" explicitly (make_engine(..., device="cpu"); a tower config with device="cpu") - the GPU engines never fall
    back to it.  forward() takes and returns CPU fp32 tensors."""

    def __init__(self, spec: ViTSpec, weights: dict, threads: Optional[int] = None):
        self.lib = _lib.load()
        self.spec = spec
        self.device = torch.device("cpu")
        self.threads = int(threads) if threads else 0
        self._keep = []

        def f32(t):
            if t is None:
                return None
            x = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
            self._keep.append(x)
            return x
        if weights["pos"].shape[0] != spec.tokens:
            raise ValueError(f"position embedding has {weights['pos'].shape[0]} rows, spec wants {spec.tokens}")
        self.kpad = 3 * spec.patch * spec.patch
        self._patch_w = f32(weights["patch_w"].reshape(spec.d, -1)[:, : self.kpad])
        self._vecs = {k: f32(weights.get(k)) for k in ("patch_b", "cls", "pos", "pre_ln_g", "pre_ln_b")}
        n = len(weights["layers"])
        self.n_layers = n
        self._layers = (_lib.VitLayer * max(n, 1))()
        for i, L in enumerate(weights["layers"]):
            ent = self._layers[i]
            for k in ("wqkv", "wo", "w1", "w2", "ln1_g", "ln1_b", "bqkv", "bo", "ls1", "ln2_g", "ln2_b", "b1", "b2", "ls2"):
                v = f32(L.get(k))
                setattr(ent, k, 0 if v is None else v.data_ptr())
            ent.sqkv = 0
            ent.s1 = 0
        self._w = _lib.VitWeights()
        self._w.patch_w = self._patch_w.data_ptr()
        for k, v in self._vecs.items():
            setattr(self._w, k, 0 if v is None else v.data_ptr())
        self._w.layers = C.cast(self._layers, C.POINTER(_lib.VitLayer))
        self._cfg = _lib.VitConfig(spec.image_size, spec.patch, spec.d, spec.heads, spec.mlp, n, spec.tokens,
                                   int(spec.has_cls), int(spec.pre_ln), _lib.ACT[spec.act], self.kpad, float(spec.eps))

    @torch.no_grad()
    def forward(self, pixels: torch.Tensor, n_layers: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        s = self.spec
        if pixels.dim() != 4 or pixels.shape[1] != 3 or pixels.shape[2] != s.image_size or pixels.shape[3] != s.image_size:
            raise ValueError(f"Input image size {tuple(pixels.shape)} doesn't match tower ({s.image_size}*{s.image_size}).")
        n_layers = self.n_layers if n_layers is None else n_layers
        if not 0 <= n_layers <= self.n_layers:
            raise ValueError("n_layers out of range")
        px = pixels.detach().to(device="cpu", dtype=torch.float32).contiguous()
        B = px.shape[0]
        if out is None:
            out = torch.empty(B, s.tokens, s.d, dtype=torch.float32)
        rc = self.lib.visrep_vit_forward_cpu(C.byref(self._cfg), C.byref(self._w), _lib.ptr(px), _lib.ptr(out), B, n_layers, self.threads)
        _lib.check(rc, "visrep_vit_forward_cpu")
        return out


def make_engine(spec: ViTSpec, weights: dict, device=None, precision: str = "bf16", products: Optional[int] = None):
    """precision 'bf16' = the MFMA throughput engine; 'fp32' = the reference-precision engine (products: its split-bf16 product set,
    None = DEFAULT_SPLIT_PRODUCTS).  device "cpu" - and only an explicit "cpu" - builds the host twin (fp32, whatever `precision` says)."""
    if device is not None and torch.device(device).type == "cpu":
        return VitEngineCPU(spec, weights)
    if precision in ("bf16", torch.bfloat16):
        return VitEngine(spec, weights, device)
    if precision in ("fp32", "float32", torch.float32):
        return VitEngineF32(spec, weights, device, products=products)
    raise ValueError(f"precision must be 'bf16' or 'fp32', got {precision!r}")


# Plane-pair products of the split-bf16 route (VitEngineF32; see its __init__).  The DEFAULT is the fp32-EQUIVALENT set: six products over
# three-plane operands (24 significand bits per operand, every dropped term < 2^-24 of the result) - what a drop-in for the reference's fp32
# towers (C_score/extract_feature.py:36-45) has to be on checkpoints nobody has validated a cheaper set on.  3 = (hi,hi) (hi,mid) (mid,hi)
# over two-plane operands (16 significand bits, the mid x mid term dropped) is an explicit OPT-IN for throughput runs (the sweep passes
# tower_products=3 and says so in its dtype field): measured at full size on SYNTHETIC N(0, 0.02) weights (profiles/round4_precision.md) -
# CLIP-L/14-336 features 8.6e-6 rel-L2 of the fp32 oracle (six products: 2.5e-6, the bf16 engine: 1.2e-2), A score 1e-7 relative, 0 of 2,400
# PCK hits flipped - at 1.8x the six-product tower's rate; real checkpoints with massive-activation outlier channels were never available
# offline, which is why it is not the default.
DEFAULT_SPLIT_PRODUCTS = 6
THROUGHPUT_SPLIT_PRODUCTS = 3


def split_planes(products: int) -> int:
    """bf16 planes per fp32 value a product set reads: three for the six-product (fp32-equivalent) set, two for 3 / 4 products."""
    return 3 if products == 6 else 2


def split_bf16_planes(x: torch.Tensor, nplanes: int = 3) -> torch.Tensor:
    """fp32 [rows, K] -> bf16 planes [rows, nplanes K] = hi | mid [| lo] (x = hi + mid + lo to 24 bits; hi + mid to 16): the operand format
    of gemm_f32_split."""
    lib = _lib.require_gpu()
    rows, K = x.shape
    planes = torch.empty(rows, nplanes * K, dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.visrep_split_bf16_planes(_lib.ptr(x), x.stride(0), rows, K, nplanes, _lib.ptr(planes), _lib.stream_ptr()), "visrep_split_bf16_planes")
    return planes


def split_bf16x3(x: torch.Tensor) -> torch.Tensor:
    return split_bf16_planes(x, 3)


def gemm_f32_split(a_planes: torch.Tensor, w_planes: torch.Tensor, bias: Optional[torch.Tensor] = None, act: str = "none",
                   resid: Optional[torch.Tensor] = None, ls: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                   planes_out: bool = False, products: int = 6):
    """fp32 act(A W^T + bias) (+ LayerScale, residual) on the bf16 matrix pipe from bf16 planes (split_bf16_planes with
    split_planes(products) planes); returns the fp32 result, or its own planes [M, npl N] when planes_out."""
    lib = _lib.require_gpu()
    npl = split_planes(products)
    M, Kp = a_planes.shape
    N, K = w_planes.shape[0], Kp // npl
    if w_planes.shape[1] != Kp or Kp % npl:
        raise ValueError("gemm_f32_split: operand plane widths differ")
    po = torch.empty(M, npl * N, dtype=torch.bfloat16, device=a_planes.device) if planes_out else None
    if out is None and not planes_out:
        out = torch.empty(M, N, dtype=torch.float32, device=a_planes.device)
    rc = lib.visrep_gemm_f32_split(_lib.ptr(a_planes), _lib.ptr(w_planes), M, N, K, products, _lib.ptr(bias), _lib.ACT[act], _lib.ptr(resid),
                                   _lib.ptr(ls), _lib.ptr(out), out.stride(0) if out is not None else N, _lib.ptr(po), _lib.stream_ptr())
    _lib.check(rc, "visrep_gemm_f32_split")
    return po if planes_out else out


def gemm_f32(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = _lib.EPI_BIAS, act: str = "none",
             resid: Optional[torch.Tensor] = None, ls: Optional[torch.Tensor] = None, w_kn: bool = False, alpha: float = 1.0,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 C = epilogue(alpha * A @ W^T + bias) (w_kn: A @ W) on the exact-fp32 MFMA path."""
    lib = _lib.require_gpu()
    M, K = a.shape
    N = w.shape[1] if w_kn else w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    rc = lib.visrep_gemm_f32(_lib.ptr(a), a.stride(0), _lib.ptr(w), w.stride(0), int(w_kn), _lib.ptr(bias), _lib.ptr(out), out.stride(0), M, N, K,
                             epilogue, _lib.ACT[act], _lib.ptr(resid), _lib.ptr(ls), float(alpha), 1, 1, None, _lib.stream_ptr())
    _lib.check(rc, "visrep_gemm_f32")
    return out


# ------------------------------------------------------------------------------------------------ thin op wrappers
def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = _lib.EPI_BIAS,
         act: str = "none", resid: Optional[torch.Tensor] = None, ls: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = epilogue(A @ W^T): A [M,K] bf16, W [N,K] bf16 (nn.Linear layout), fp32 bias/ls.  EPI_VT is exposed via linear_vt()."""
    lib = _lib.require_gpu()
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32 if epilogue == _lib.EPI_F32 else torch.bfloat16, device=a.device)
    rc = lib.visrep_gemm_bf16(_lib.ptr(a), a.stride(0), _lib.ptr(w), w.stride(0), _lib.ptr(bias), _lib.ptr(out), out.stride(0),
                              M, N, K, epilogue, _lib.ACT[act], _lib.ptr(resid), _lib.ptr(ls), _lib.stream_ptr())
    _lib.check(rc, "visrep_gemm_bf16")
    return out


def linear_vt(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: Optional[torch.Tensor] = None,
              col_offset: int = 0) -> torch.Tensor:
    """V projection in the attention kernel's V^T layout: returns [N, ldvt] bf16 (see csrc/attention.hip).

    out / col_offset: write the tokens of `a` as columns [col_offset, col_offset + M) of an existing (zero-initialised) V^T
    buffer - how a joint key sequence is assembled from several token groups (SD3: image tokens, then prompt tokens);
    col_offset must be a multiple of 16 (the perm16 granule)."""
    lib = _lib.require_gpu()
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        ldvt = _round_up(M, 64) + 64
        out = torch.zeros(N, ldvt, dtype=torch.bfloat16, device=a.device)
    if col_offset % 16 or col_offset + M > out.shape[1]:
        raise ValueError("linear_vt: col_offset must be a multiple of 16 and fit the buffer")
    rc = lib.visrep_gemm_bf16(_lib.ptr(a), a.stride(0), _lib.ptr(w), w.stride(0), _lib.ptr(bias), C.c_void_p(out.data_ptr() + 2 * col_offset),
                              out.stride(0), M, N, K, _lib.EPI_VT, 0, None, None, _lib.stream_ptr())
    _lib.check(rc, "visrep_gemm_bf16(VT)")
    return out


def gemm_rows(a: torch.Tensor, period: int, stride: int, first: int, rows: int, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
              epilogue: int = _lib.EPI_BIAS, act: str = "none", ln_rt: Optional[torch.Tensor] = None, ln_s: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GEMM over a periodic subset of a's rows (visrep_gemm_bf16_rows): logical row r = physical (r // period) * stride + r % period + first.
    EPI_VT returns / fills a V^T buffer [N, ld] (columns = logical rows, perm16), the others [rows, N]."""
    lib = _lib.require_gpu()
    N, K = w.shape[0], a.shape[1]
    if (rows - 1) // period * stride + (rows - 1) % period + first >= a.shape[0]:
        raise ValueError("gemm_rows: the row map reaches past a")
    if out is None:
        if epilogue == _lib.EPI_VT:
            out = torch.zeros(N, _round_up(rows, 64) + 64, dtype=torch.bfloat16, device=a.device)
        else:
            out = torch.empty(rows, N, dtype=torch.float32 if epilogue == _lib.EPI_F32 else torch.bfloat16, device=a.device)
    rc = lib.visrep_gemm_bf16_rows(_lib.ptr(a), a.stride(0), period, stride, first, _lib.ptr(w), w.stride(0), _lib.ptr(bias), _lib.ptr(ln_rt),
                                   _lib.ptr(ln_s), _lib.ptr(out), out.stride(0), rows, N, K, epilogue, _lib.ACT[act], _lib.stream_ptr())
    _lib.check(rc, "visrep_gemm_bf16_rows")
    return out


def layernorm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    lib = _lib.require_gpu()
    y = torch.empty_like(x)
    rc = lib.visrep_layernorm(_lib.ptr(x), x.stride(0), _lib.ptr(g), _lib.ptr(b), _lib.ptr(y), y.stride(0), x.shape[0], x.shape[1],
                              eps, _lib.stream_ptr())
    _lib.check(rc, "visrep_layernorm")
    return y


def mhsa(qk: torch.Tensor, vt: torch.Tensor, B: int, T: int, H: int, scale: float) -> torch.Tensor:
    lib = _lib.require_gpu()
    out = torch.empty(B * T, H * 64, dtype=torch.bfloat16, device=qk.device)
    rc = lib.visrep_mhsa_fwd(_lib.ptr(qk), qk.stride(0), _lib.ptr(vt), vt.stride(0), _lib.ptr(out), out.stride(0), B, T, H, 64,
                             scale, _lib.stream_ptr())
    _lib.check(rc, "visrep_mhsa_fwd")
    return out


def mhsa_cls_supported(T: int) -> bool:
    return bool(_lib.load().visrep_mhsa_cls_supported(int(T)))


def mhsa_cls(qk: torch.Tensor, vt: torch.Tensor, vcls: torch.Tensor, B: int, T: int, H: int) -> torch.Tensor:
    """Image-aligned self-attention (visrep_mhsa_cls_fwd): qk [B T, 2 H 64] with pre-scaled Q; vt = V^T of the patch tokens (column
    b (T - 1) + t - 1, as linear_vt over the patch rows writes it); vcls = V rows of the CLS tokens [B, H 64]."""
    lib = _lib.require_gpu()
    out = torch.empty(B * T, H * 64, dtype=torch.bfloat16, device=qk.device)
    rc = lib.visrep_mhsa_cls_fwd(_lib.ptr(qk), qk.stride(0), _lib.ptr(vt), vt.stride(0), _lib.ptr(vcls), vcls.stride(0), _lib.ptr(out),
                                 out.stride(0), B, T, H, 64, _lib.stream_ptr())
    _lib.check(rc, "visrep_mhsa_cls_fwd")
    return out
