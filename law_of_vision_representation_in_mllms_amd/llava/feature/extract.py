"""Data-parallel feature dump of a bare vision tower — drop-in for llava/feature/extract.py (the reference's only multi-GPU
use of a tower; scripts/v1_5/feature/extract.sh launches it on 8 GPUs).

Reference behaviour (extract.py:17-23, 193-231): a 7-entry registry id -> builder, the tower wrapped in DDP, a
DistributedSampler(shuffle=False) over the LLaVA conversation json, batch 1, `model(images.bf16)` and
`torch.save(out[0].squeeze().cpu(), <root>/<image's parent dir>/<stem>.pt)` for every image whose file does not exist
yet, then a barrier.  Tokenizer, conversation pre-processing and DDP contribute nothing to the files (a frozen tower has no
gradients to reduce), so they are not rebuilt.

Here: one process per GPU, entries `rank::world` of the json (the sampler's order without its wrap-around padding, which only
re-visits existing files), JPEG decode + expand2square + the tower's processor on a thread pool one batch ahead of the GPU,
the tower's HIP forward on batches of `per_device_train_batch_size` images (EVERY image of a batch is written — the
reference drops all but the first when the batch is larger than 1), one barrier at the end, no data-path collective.
The output root is `--feature_dir` (the reference hard-wires a private NAS path); the other flag names are the reference's,
unknown training flags of extract.sh are ignored.
"""
import argparse
import json
import os
from concurrent.futures import ThreadPoolExecutor

import torch
import torch.distributed as dist
from PIL import Image

from ..mm_utils import expand2square
from ..model.multimodal_encoder.builder import build_diffusion_vision_tower, build_dinov2_vision_tower, build_vision_tower

build_function_mapping = {
    'openai/clip-vit-large-patch14-336': build_vision_tower,
    'stabilityai/stable-diffusion-2-1': build_diffusion_vision_tower,
    'stabilityai/stable-diffusion-1-5': build_diffusion_vision_tower,
    'runwayml/stable-diffusion-v1-5': build_diffusion_vision_tower,
    'lambdalabs/sd-image-variations-diffusers': build_diffusion_vision_tower,
    'facebook/dinov2-large': build_dinov2_vision_tower,
    'stabilityai/stable-diffusion-xl-base-1.0': build_diffusion_vision_tower,
}


def cleanup():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def list_images(data_args):
    """[(image path, feature path suffix)] of the json entries that have an 'image' (extract.py:109-124, 226-229)."""
    with open(data_args.data_path) as f:
        entries = json.load(f)
    if data_args.image_folder is None:
        raise ValueError("image_folder is required (the reference's fallback table of private NAS folders is not reproduced)")
    out = []
    for e in entries:
        if 'image' not in e:
            continue
        path = os.path.join(data_args.image_folder, e['image'])
        parts = path.split('/')
        out.append((path, os.path.join(parts[-2], parts[-1].split('.')[0] + '.pt')))
    return out


def load_image(path, processor, image_aspect_ratio):
    """extract.py:121-147: RGB, optional pad to square with the processor's mean colour, the tower's own processor."""
    image = Image.open(path).convert('RGB')
    if image_aspect_ratio == 'pad':
        image = expand2square(image, tuple(int(x * 255) for x in processor.image_mean))
    return processor.preprocess(image, return_tensors='pt')['pixel_values'][0]


def _device_input_stream(chunks, processor, image_aspect_ratio, device):
    """The all-device input path: host threads parse + Huffman-decode the files of chunk i + 1 while the GPU
    reconstructs, pads, resizes, crops and normalises chunk i (device_jpeg + device_preprocess; bit-identical to load_image)."""
    from ... import device_jpeg as DJ
    from ... import device_preprocess as DP
    dec = DJ.DeviceJpegDecoder(device)
    # ViT towers: the device twin of their SimpleImageProcessor; diffusion towers: DiffImageProcessor.device_twin (resize-only, [-1, 1])
    dproc = processor.device_twin(device) if hasattr(processor, "device_twin") else DP.DevicePreprocessor.like(processor, device)
    bg = tuple(int(x * 255) for x in dproc.image_mean)
    pending = dec.submit([p for p, _ in chunks[0]]) if chunks else None
    for i, chunk in enumerate(chunks):
        nxt = dec.submit([p for p, _ in chunks[i + 1]]) if i + 1 < len(chunks) else None
        imgs = dec.finish(pending)
        yield chunk, dproc.preprocess_padded(imgs, bg if image_aspect_ratio == 'pad' else None)
        pending = nxt


def inference(model_args, data_args, training_args, model=None, workers=8, device_decode=None):
    """Returns the number of feature files this rank wrote.  device_decode (default: VISREP_DEVICE_DECODE=1): JPEG reconstruction,
    padding, resize and normalisation on the GPU instead of PIL + the CPU processor on the thread pool."""
    rank, world = _rank_world()
    if model is None:
        model = build_function_mapping[model_args.vision_tower](model_args)       # KeyError for ids outside the registry, as the reference
    data_args.image_processor = processor = model.image_processor
    root = training_args.feature_dir
    todo = [(p, os.path.join(root, rel)) for p, rel in list_images(data_args)[rank::world]]
    todo = [(p, o) for p, o in todo if not os.path.exists(o)]
    bs = max(1, int(training_args.per_device_train_batch_size))
    chunks = [todo[i:i + bs] for i in range(0, len(todo), bs)]
    written = 0
    if device_decode is None:
        device_decode = os.environ.get("VISREP_DEVICE_DECODE") == "1"
    if device_decode and not (hasattr(processor, "resize_to") or hasattr(processor, "device_twin")):
        raise ValueError("device_decode needs an image processor with a device twin (SimpleImageProcessor geometry or DiffImageProcessor)")

    def host_stream(pool):
        submit = lambda chunk: [pool.submit(load_image, p, processor, data_args.image_aspect_ratio) for p, _ in chunk]
        pending = submit(chunks[0]) if chunks else None
        for i, chunk in enumerate(chunks):
            nxt = submit(chunks[i + 1]) if i + 1 < len(chunks) else None
            yield chunk, torch.stack([f.result() for f in pending])
            pending = nxt

    def save(feat, out_path):
        # extract.py:211-214 `torch.save(feat.squeeze().cpu().clone(), path)`; the clone out of the batch tensor is made on the writer thread
        os.makedirs(os.path.dirname(out_path), exist_ok=True)
        torch.save(feat.squeeze().clone(), out_path)

    # one download per batch, files written by a small pool behind the GPU (at most two batches of features in flight)
    with torch.no_grad(), ThreadPoolExecutor(max_workers=workers) as pool, ThreadPoolExecutor(max_workers=max(1, min(workers, 8))) as writers:
        stream = _device_input_stream(chunks, processor, data_args.image_aspect_ratio, model.device) if device_decode else host_stream(pool)
        inflight = []
        for chunk, images in stream:
            images = images.to(dtype=torch.bfloat16)
            outputs = model(images).cpu()
            inflight.append([writers.submit(save, feat, out_path) for (_, out_path), feat in zip(chunk, torch.split(outputs, 1))])
            written += len(chunk)
            if len(inflight) > 2:
                for f in inflight.pop(0):
                    f.result()
        for batch in inflight:
            for f in batch:
                f.result()                                             # re-raises a failed write
    return written


def build_parser():
    p = argparse.ArgumentParser()
    # ModelArguments (llava/train/train.py:71-87)
    p.add_argument('--vision_tower', type=str, required=True)
    p.add_argument('--mm_vision_select_layer', type=int, default=-1)
    p.add_argument('--mm_vision_select_feature', type=str, default='patch')
    p.add_argument('--up_ft_index', type=int, default=0)
    p.add_argument('--t', type=int, default=1)
    p.add_argument('--prompt', type=str, default='')
    p.add_argument('--ensemble_size', type=int, default=1)
    p.add_argument('--img_size', type=int, default=768)
    # DataArguments (:91-98)
    p.add_argument('--data_path', type=str, required=True)
    p.add_argument('--image_folder', type=str, default=None)
    p.add_argument('--image_aspect_ratio', type=str, default='square')
    # TrainingArguments
    p.add_argument('--per_device_train_batch_size', type=int, default=1)
    p.add_argument('--feature_dir', type=str, required=True)
    p.add_argument('--local_rank', type=int, default=0)
    return p


def main(argv=None):
    args, _ignored = build_parser().parse_known_args(argv)
    from ... import dist_env
    owned = dist_env.init_from_env()                                  # under torchrun / deepspeed launchers: one process per GPU
    try:
        return inference(args, args, args)
    finally:
        dist_env.finalize(owned)                                      # barrier + teardown (extract.py:259-260)


if __name__ == "__main__":
    main()
