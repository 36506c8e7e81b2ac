"""C-score feature extraction on MI355X — drop-in for C_score/extract_feature.py.

Same entry points: extract_features(image_path) -> Tensor[1, C, h, w] and process_images(input_dir, output_dir) writing
<output_dir>/<class>/<image>_<suffix>.pt (extract_feature.py:54-106,110-130).  The reference is a module-level script with
hard-coded `feature`/paths (all ten `feature` values are supported: the four ViT towers and DIFT1.5 / DIFT2.1 / DIFTXL / IMDIFT /
DiTDIFT / SD3DIFT) and an internally inconsistent DINOv2 branch (224-px input reshaped as 24x24); here
`configure(feature, img_size, suffix)` sets them, grid = img_size // patch, and nothing runs at import time.
Pre-processing is the reference's own (NOT the HF processors): PIL resize((s, s)) then (x/255 - 0.5) * 2 (l.65-67).
"""
import os
from types import SimpleNamespace

import numpy as np
import torch
from PIL import Image

# Define the input and output paths (extract_feature.py:16-17)
input_path = './data/SPair-71k/JPEGImages'
output_path = './spair_feature/dino336'

feature = "DINOv2"
_DEFAULT_SIZE = {"CLIP": 224, "OPENCLIP": 224, "DINOv2": 224, "SigLIP": 224}
_TOWER_ID = {"CLIP": 'openai/clip-vit-large-patch14', "OPENCLIP": 'laion/CLIP-ViT-L-14-laion2B-s32B-b82K',
             "DINOv2": 'facebook/dinov2-large', "SigLIP": 'google/siglip-base-patch16-224'}
# diffusion featurizers (extract_feature.py:23-34) and the input side each one is fed at (:56-63)
_DIFT = {"DIFT2.1": ("sd", 'stabilityai/stable-diffusion-2-1', 768), "DIFT1.5": ("sd", 'runwayml/stable-diffusion-v1-5', 768),
         "DIFTXL": ("sd", 'stabilityai/stable-diffusion-xl-base-1.0', 512), "IMDIFT": ("imsd", None, 768),
         "DiTDIFT": ("dit", None, 512), "SD3DIFT": ("sd3", None, 512)}
_state = SimpleNamespace(dift=None, img_size=None, suffix="dino336", batch=64, kind="vit", device_preprocess=False, flip=False)


# The precision each ViT tower runs in HERE follows the reference script: CLIP / OPENCLIP / DINOv2 are built with no dtype cast
# and fed fp32 pixels (extract_feature.py:36-45,80-87) -> fp32 engine; SigLIP is cast `.to(torch.bfloat16)` (l.46-47) -> bf16 engine.
_REF_PRECISION = {"CLIP": "fp32", "OPENCLIP": "fp32", "DINOv2": "fp32", "SigLIP": "bf16"}


class args_c:
    def __init__(self, img_size=None, synthetic_weights=False, tower_precision=None):
        self.mm_vision_select_layer = -2
        self.vit_img_size = img_size
        self.synthetic_weights = synthetic_weights
        self.tower_precision = tower_precision


def configure(feature_name="DINOv2", img_size=None, suffix=None, synthetic_weights=False, batch=64, precision=None):
    """Build the tower once (the reference does this at import, extract_feature.py:23-50).

    precision: None = the reference's own dtype for that tower (_REF_PRECISION: fp32 for CLIP / OPENCLIP / DINOv2, bf16 for SigLIP);
    'bf16' selects the MFMA throughput engine for any tower (~20x faster, features within ~1e-2 of the fp32 ones)."""
    global feature
    from ..llava.model.multimodal_encoder.clip_encoder import CLIPVisionTower
    from ..llava.model.multimodal_encoder.dinov2_encoder import DinoV2VisionTower
    from ..llava.model.multimodal_encoder.siglip_encoder import SigLipVisionTower
    if feature_name in _DIFT:
        return _configure_dift(feature_name, img_size, suffix, synthetic_weights, batch)
    if feature_name not in _TOWER_ID:
        raise KeyError(f"unknown feature {feature_name!r}")
    feature = feature_name
    _state.kind = "vit"
    size = img_size or _DEFAULT_SIZE[feature_name]
    a = args_c(img_size=size, synthetic_weights=synthetic_weights, tower_precision=precision or _REF_PRECISION[feature_name])
    cls = {"CLIP": CLIPVisionTower, "OPENCLIP": CLIPVisionTower, "DINOv2": DinoV2VisionTower, "SigLIP": SigLipVisionTower}[feature_name]
    _state.dift = cls(vision_tower=_TOWER_ID[feature_name], args=a)
    _state.img_size = size
    _state.suffix = suffix or {"CLIP": "clip", "OPENCLIP": "openclip", "DINOv2": f"dino{size}" if size != 224 else "dino", "SigLIP": "siglip"}[feature_name]
    _state.batch = batch
    return _state.dift


def _configure_dift(feature_name, img_size, suffix, synthetic_weights, batch):
    global feature
    from ..llava.model.multimodal_encoder.diffLVLM.src.models.dift_dit import DiTFeaturizer
    from ..llava.model.multimodal_encoder.diffLVLM.src.models.dift_imsd import IMSDFeaturizer
    from ..llava.model.multimodal_encoder.diffLVLM.src.models.dift_sd import SDFeaturizer
    from ..llava.model.multimodal_encoder.diffLVLM.src.models.dift_sd3 import SD3Featurizer
    kind, sd_id, size = _DIFT[feature_name]
    syn = True if synthetic_weights else None
    if kind == "sd":
        _state.dift = SDFeaturizer(sd_id=sd_id, synthetic=syn)
    else:
        _state.dift = {"imsd": IMSDFeaturizer, "dit": DiTFeaturizer, "sd3": SD3Featurizer}[kind](synthetic=syn)
    feature = feature_name
    _state.kind, _state.img_size, _state.batch = kind, img_size or size, min(batch, 8)
    _state.suffix = suffix or feature_name.lower()
    return _state.dift


def _dift_maps(px, post_noise=None, ddim_noise=None):
    """[B, 3, s, s] -> [B, C, h, w] as extract_feature.py:68-103 returns per image (prompt '', ensemble 1, default t / block)."""
    f = _state.dift.forward(px.to(torch.bfloat16), prompt='', ensemble_size=1, post_noise=post_noise, ddim_noise=ddim_noise)
    if f.dim() == 3:
        f = f.unsqueeze(0)
    if _state.kind == "dit":
        f = f.permute(0, 1, 3, 2)              # extract_feature.py:95: permute(0, 1, 3, 2).view(...) - the map is stored transposed
    return f


def _load_pixels(image_path, size):
    """extract_feature.py:65-67: RGB, resize((s, s)) (PIL bicubic), PILToTensor, (x / 255 - 0.5) * 2 in float32.  The arithmetic is
    done in numpy: the same correctly-rounded float32 operations, without torch's intra-op thread pool waking up for a 150k-element
    tensor (on a 256-core host that cost more than the JPEG decode: 29 ms per image serial, profiles/round1_pipeline.md)."""
    img = _maybe_flip(Image.open(image_path).convert('RGB')).resize((size, size))
    a = np.asarray(img).transpose(2, 0, 1).astype(np.float32)               # PILToTensor layout: [3, H, W]
    return torch.from_numpy((a / np.float32(255.0) - np.float32(0.5)) * np.float32(2.0))


def _load_pixels_worker(image_path, size):
    """_load_pixels for the decode pool, with the arithmetic as torch ops: inside a pool thread they run inline (no intra-op
    fan-out) and release the GIL, whereas the numpy steps of _load_pixels make eight threads queue on the GIL.  Measured on a
    256-core host, 1024 JPEGs -> DINOv2-L maps -> files: torch-in-pool 690-790 images/s, numpy-in-pool 180-210, and on the calling
    thread torch 34 vs numpy 200-310 (profiles/round1_pipeline.md).  Same bits either way (tests/test_host_preprocess.py)."""
    img = _maybe_flip(Image.open(image_path).convert('RGB')).resize((size, size))
    a = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1)           # PILToTensor: uint8 [3, H, W]
    return (a / 255.0 - 0.5) * 2


def _load_pixels_device(image_path, size, device="cuda"):
    """Same values as _load_pixels, but only the JPEG decode runs on the host: Pillow-exact bicubic resize and the
    (x / 255 - 0.5) * 2 arithmetic run on the GPU (device_preprocess)."""
    return _finish_on_device(_decode_rgb(image_path, size), size, device)


def _decode_rgb(image_path, size=None):
    """JPEG -> uint8 [H, W, 3] on the host (the part that runs on the decode pool)."""
    return np.array(_maybe_flip(Image.open(image_path).convert('RGB')))


def _maybe_flip(img):
    """ADAPT_FLIP's second feature set (pck_train.py:33-37 `<img>_<model>_flip.pt`): the same extraction on the mirrored image
    (pck_train.py:112 `img1.transpose(Image.FLIP_LEFT_RIGHT)`)."""
    return img.transpose(Image.FLIP_LEFT_RIGHT) if getattr(_state, "flip", False) else img


def _finish_on_device(a, size, device="cuda"):
    from .. import device_preprocess as DP
    dev = torch.from_numpy(a).to(device)
    return DP.to_tensor(DP.resize_u8(dev, (size, size)), (0, 0, size, size), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))


def _prefetched_device_decode(chunks, device="cuda"):
    """The all-device input path: host threads only parse and Huffman-decode the JPEG files of chunk i + 1 while the GPU
    reconstructs (IDCT, chroma upsampling, colour conversion: device_jpeg), resizes (Pillow-exact bicubic) and normalises chunk i.
    Bit-identical to _load_pixels (tests/test_gpu_jpeg.py); PNG / progressive / CMYK files go through PIL inside the decoder."""
    from .. import device_jpeg as DJ
    from .. import device_preprocess as DP
    dec = getattr(_state, "jpeg_decoder", None)
    if dec is None:
        dec = _state.jpeg_decoder = DJ.DeviceJpegDecoder(device)
    size = _state.img_size
    pending = dec.submit([p for p, _ in chunks[0]]) if chunks else None
    for i, chunk in enumerate(chunks):
        nxt = dec.submit([p for p, _ in chunks[i + 1]]) if i + 1 < len(chunks) else None
        imgs = dec.finish(pending)
        yield chunk, DP.preprocess_batch(imgs, [(size, size)] * len(imgs), [(0, 0, size, size)] * len(imgs), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5),
                                         flip=getattr(_state, "flip", False))       # flip: Image.FLIP_LEFT_RIGHT, folded into the fetch
        pending = nxt


def _to_maps(f):
    # tokens [B, N, C] -> [B, C, g, g]   (extract_feature.py:82,86,90 with g = sqrt(N))
    B, N, C = f.shape
    g = int(round(N ** 0.5))
    return f.permute(0, 2, 1).reshape(B, C, g, g)


def extract_features(image_path):
    if _state.dift is None:
        configure(feature)
    px = _load_pixels(image_path, _state.img_size).unsqueeze(0)
    if _state.kind != "vit":
        return _dift_maps(px)
    return _to_maps(_state.dift.forward(px))


def _prefetched(chunks, load, workers, finish=None):
    """Yield (chunk, pixel batch) with `load` (JPEG decode, + resize on the host path) of chunk i+1 running on a thread pool while
    the GPU works on chunk i (PIL releases the GIL while decoding; the reference decodes, runs and saves one image at a time).
    `finish` (device resize + normalise of one decoded image) runs on the calling thread, which owns the HIP stream."""
    from concurrent.futures import ThreadPoolExecutor
    done = (lambda x: x) if finish is None else (lambda x: finish(x, _state.img_size))
    if workers <= 1 or not chunks:
        for chunk in chunks:
            yield chunk, torch.stack([done(load(p, _state.img_size)) for p, _ in chunk])
        return
    with ThreadPoolExecutor(max_workers=workers) as pool:
        submit = lambda chunk: [pool.submit(load, p, _state.img_size) for p, _ in chunk]
        pending = submit(chunks[0])
        for i, chunk in enumerate(chunks):
            nxt = submit(chunks[i + 1]) if i + 1 < len(chunks) else None
            yield chunk, torch.stack([done(f.result()) for f in pending])
            pending = nxt


def process_images(input_dir, output_dir, workers=8, flip=False):
    """Walks input_dir like the reference; towers run in batches (the reference runs batch 1), files are identical.
    flip=True writes the mirrored images' maps as `<image>_<suffix>_flip.pt` (what pck_train's ADAPT_FLIP reads)."""
    if _state.dift is None:
        configure(feature)
    _state.flip = bool(flip)
    try:
        return _process_images(input_dir, output_dir, workers)
    finally:
        _state.flip = False


def _process_images(input_dir, output_dir, workers):
    todo = []
    for root, _, files in os.walk(input_dir):
        for file in sorted(files):
            if file.endswith(('.jpg', '.jpeg', '.png')):
                class_name = os.path.basename(root)
                image_name = os.path.splitext(file)[0]
                suffix = f'{_state.suffix}{"_flip" if getattr(_state, "flip", False) else ""}'
                todo.append((os.path.join(root, file), os.path.join(output_dir, class_name, f'{image_name}_{suffix}.pt')))
    d = torch.distributed
    if d.is_available() and d.is_initialized():
        todo = todo[d.get_rank()::d.get_world_size()]                        # image-sharded across ranks, no collective
    on_device = getattr(_state, "device_preprocess", False)           # decode on the pool, resize + normalise on the GPU
    chunks = [todo[s:s + _state.batch] for s in range(0, len(todo), _state.batch)]
    from concurrent.futures import ThreadPoolExecutor

    def save(m, out):
        # the copy out of the batch tensor (torch.save of a view would pickle the whole batch's storage) is made HERE, on the writer thread:
        # 64 one-megabyte clones on the calling thread each wake torch's intra-op pool and re-take the GIL against the writers - measured
        # 37 ms per clone once the calling thread is no longer parked in pool futures (profiles/round3_pipeline.md)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        torch.save(m.unsqueeze(0).clone(), out)
    # the 1-MB-per-image pickles are written by a small pool behind the GPU (they were most of the loop's wall-clock:
    # profiles/round1_pipeline.md); at most two batches of maps are in flight
    with ThreadPoolExecutor(max_workers=max(1, min(workers, 8))) as writers:
        inflight = []
        if on_device and getattr(_state, "device_decode", True):
            stream = _prefetched_device_decode(chunks)
        elif on_device:
            stream = _prefetched(chunks, _decode_rgb, workers, _finish_on_device)
        else:
            stream = _prefetched(chunks, _load_pixels_worker if workers > 1 else _load_pixels, workers)
        for chunk, px in stream:
            maps = (_dift_maps(px) if _state.kind != "vit" else _to_maps(_state.dift.forward(px))).cpu()
            batch = [writers.submit(save, m, out) for (_, out), m in zip(chunk, maps)]
            for (_, out) in chunk:
                print(f'Saved features to {out}')
            inflight.append(batch)
            if len(inflight) > 2:
                for f in inflight.pop(0):
                    f.result()
        for batch in inflight:
            for f in batch:
                f.result()                                             # re-raises a failed write


if __name__ == "__main__":
    from .. import dist_env
    _owned = dist_env.init_from_env()                                 # under torchrun: images sharded rank::world
    try:
        configure(feature)
        process_images(input_path, output_path)
    finally:
        dist_env.finalize(_owned)
