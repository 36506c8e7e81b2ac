"""llava/feature/extract.py (data-parallel tower feature dump) on CPU: registry, json walk, pad-to-square + processor, file
naming, skip-existing, every image of a batch written, and the rank::world image sharding on 2 gloo ranks.  The tower is a
stand-in module (the real towers need the HIP library); test_gpu_dropin.py runs a real one."""
import json
import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.multiprocessing as mp
from PIL import Image

from law_of_vision_representation_in_mllms_amd import vit_weights as VW
from law_of_vision_representation_in_mllms_amd.llava.feature import extract as FX
from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.image_processing import default_image_processor


class StubTower(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.image_processor = default_image_processor(VW.SPECS["openai/clip-vit-large-patch14"])
        self.seen = []

    def forward(self, images):
        assert images.dtype == torch.bfloat16 and images.shape[1:] == (3, 224, 224)
        self.seen.append(images.shape[0])
        # [B, 4, 3]: per-quadrant channel means, so the features identify the image
        q = images.float().view(-1, 3, 2, 112, 2, 112).mean(dim=(3, 5)).reshape(-1, 3, 4).permute(0, 2, 1)
        return q.to(torch.bfloat16)


def make_data(tmp, n=7):
    rs = np.random.RandomState(3)
    entries = []
    for i in range(n):
        sub = "coco" if i % 2 else "vg"
        os.makedirs(f"{tmp}/imgs/{sub}", exist_ok=True)
        w, h = int(rs.randint(60, 200)), int(rs.randint(60, 200))
        Image.fromarray(rs.randint(0, 255, (h, w, 3), dtype=np.uint8)).save(f"{tmp}/imgs/{sub}/im{i}.jpg")
        entries.append({"id": i, "image": f"{sub}/im{i}.jpg", "conversations": []})
    entries.insert(2, {"id": "text-only", "conversations": []})
    with open(f"{tmp}/data.json", "w") as f:
        json.dump(entries, f)
    return SimpleNamespace(vision_tower="stub", data_path=f"{tmp}/data.json", image_folder=f"{tmp}/imgs", image_aspect_ratio="pad",
                           per_device_train_batch_size=3, feature_dir=f"{tmp}/feats")


def test_registry_is_the_references():
    assert sorted(FX.build_function_mapping) == sorted([
        'openai/clip-vit-large-patch14-336', 'stabilityai/stable-diffusion-2-1', 'stabilityai/stable-diffusion-1-5',
        'runwayml/stable-diffusion-v1-5', 'lambdalabs/sd-image-variations-diffusers', 'facebook/dinov2-large',
        'stabilityai/stable-diffusion-xl-base-1.0'])
    a = FX.build_parser().parse_known_args(["--vision_tower", "facebook/dinov2-large", "--data_path", "d.json", "--feature_dir", "o",
                                            "--bf16", "True", "--deepspeed", "zero2.json"])[0]
    assert (a.img_size, a.t, a.up_ft_index, a.mm_vision_select_layer, a.image_aspect_ratio) == (768, 1, 0, -1, "square")


def test_dump_writes_one_file_per_image_and_skips_existing(tmp_path):
    args = make_data(str(tmp_path))
    tower = StubTower()
    n = FX.inference(args, args, args, model=tower)
    assert n == 7 and tower.seen == [3, 3, 1]
    files = sorted(os.path.relpath(os.path.join(r, f), args.feature_dir) for r, _, fs in os.walk(args.feature_dir) for f in fs)
    assert files == sorted(f"{'coco' if i % 2 else 'vg'}/im{i}.pt" for i in range(7))
    f3 = torch.load(f"{args.feature_dir}/coco/im3.pt")
    assert f3.shape == (4, 3) and f3.dtype == torch.bfloat16
    want = tower(FX.load_image(f"{tmp_path}/imgs/coco/im3.jpg", tower.image_processor, "pad")[None].to(torch.bfloat16))[0]
    assert torch.equal(f3, want)
    os.remove(f"{args.feature_dir}/vg/im4.pt")
    tower.seen.clear()
    assert FX.inference(args, args, args, model=tower) == 1 and tower.seen == [1]        # only the missing file is recomputed
    # 'pad' changes the pixels of a non-square image (mean-colour bars), 'square' squashes
    a = FX.load_image(f"{tmp_path}/imgs/coco/im3.jpg", tower.image_processor, "pad")
    b = FX.load_image(f"{tmp_path}/imgs/coco/im3.jpg", tower.image_processor, "square")
    assert a.shape == b.shape == (3, 224, 224) and not torch.equal(a, b)


def _worker(rank, world, tmp, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = SimpleNamespace(vision_tower="stub", data_path=f"{tmp}/data.json", image_folder=f"{tmp}/imgs", image_aspect_ratio="pad",
                           per_device_train_batch_size=2, feature_dir=f"{tmp}/feats")
    n = FX.inference(args, args, args, model=StubTower())
    dist.barrier()
    q.put((rank, n))
    dist.destroy_process_group()


def test_two_ranks_split_the_images(tmp_path):
    args = make_data(str(tmp_path))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, str(tmp_path), port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {0: 4, 1: 3}                                      # entries 0,2,4,6 and 1,3,5 of the image list
    n_files = sum(len(fs) for _, _, fs in os.walk(args.feature_dir))
    assert n_files == 7
    single = str(tmp_path / "single")
    args.feature_dir = single
    FX.inference(args, args, args, model=StubTower())
    for i in range(7):
        rel = f"{'coco' if i % 2 else 'vg'}/im{i}.pt"
        assert torch.equal(torch.load(f"{single}/{rel}"), torch.load(f"{tmp_path}/feats/{rel}"))
