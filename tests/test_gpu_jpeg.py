"""Device JPEG decode on the MI355X: files -> host Huffman decode -> visrep_jpeg_reconstruct must equal PIL's
Image.open(...).convert('RGB') bit for bit (the oracle chain of tests/test_host_jpeg.py is the same arithmetic on the CPU); the
C-feature and LLaVA feature-dump loaders on the all-device input path produce the same files as on the PIL path."""
import io
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(__file__))
from test_host_jpeg import CASES, encode, photo  # noqa: E402

from law_of_vision_representation_in_mllms_amd import device_jpeg as DJ  # noqa: E402
from oracle import jpeg as OJ  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_batch_of_ragged_files_equals_pil_bit_for_bit():
    files = [encode(photo(w, h, w * 1000 + h), **kw) for w, h, kw in CASES]
    files.append(encode(photo(77, 50, 5, grey=True), quality=80))
    dec = DJ.DeviceJpegDecoder(DEV, threads=4)
    got = dec.decode(files)
    assert dec.stats["device"] == len(files) and dec.stats["pil"] == 0
    for data, g in zip(files, got):
        want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        assert g.dtype == torch.uint8 and tuple(g.shape) == want.shape
        assert np.array_equal(g.cpu().numpy(), want)
    # the oracle (what the CPU suite pins against PIL) agrees too
    info, coef, qtab = DJ.entropy_decode(files[2])
    assert np.array_equal(got[2].cpu().numpy(), OJ.reconstruct(info, coef, qtab))
    # a second batch through the same decoder, one image
    assert np.array_equal(dec.decode(files[:1])[0].cpu().numpy(), np.asarray(Image.open(io.BytesIO(files[0])).convert("RGB")))


def test_files_the_decoder_does_not_take_go_through_pil_and_are_counted(tmp_path):
    prog = encode(photo(64, 64, 7), quality=75, progressive=True)
    png = io.BytesIO()
    photo(40, 30, 3).save(png, "PNG")
    good = encode(photo(64, 48, 1), quality=75)
    path = tmp_path / "x.jpg"
    path.write_bytes(good)
    dec = DJ.DeviceJpegDecoder(DEV, threads=2)
    out = dec.decode([prog, png.getvalue(), str(path)])
    assert dec.stats["device"] == 1 and dec.stats["pil"] == 2 and any("progressive" in k for k in dec.stats["pil_reasons"])
    for data, g in zip((prog, png.getvalue(), good), out):
        assert np.array_equal(g.cpu().numpy(), np.asarray(Image.open(io.BytesIO(data)).convert("RGB")))


def _write_tree(root, n=7):
    os.makedirs(root / "JPEGImages" / "cat")
    sizes = [(500, 375), (375, 500), (333, 251), (640, 480), (97, 61), (256, 256), (300, 301)]
    for i in range(n):
        w, h = sizes[i % len(sizes)]
        photo(w, h, 40 + i).save(root / "JPEGImages" / "cat" / f"im{i}.jpg", quality=[75, 90, 60][i % 3], subsampling=[2, 1, 0][i % 3])
    return str(root / "JPEGImages")


def test_extract_feature_device_decode_path_writes_the_same_files(tmp_path, monkeypatch):
    from law_of_vision_representation_in_mllms_amd.C_score import extract_feature as EF
    src = _write_tree(tmp_path)
    EF.configure("DINOv2", img_size=224, synthetic_weights=True, batch=4)
    # pixels: all-device path vs the reference order (PIL decode + PIL resize + float arithmetic), bit-identical
    chunks = [[(os.path.join(src, "cat", f"im{i}.jpg"), None) for i in range(7)]]
    _, px = next(EF._prefetched_device_decode(chunks, DEV))
    for i in range(7):
        assert torch.equal(px[i].cpu(), EF._load_pixels(chunks[0][i][0], 224)), i
    EF._state.flip = True
    try:
        _, pxf = next(EF._prefetched_device_decode(chunks, DEV))
        assert torch.equal(pxf[3].cpu(), EF._load_pixels(chunks[0][3][0], 224))
    finally:
        EF._state.flip = False
    outs = {}
    for tag, devpre in (("pil", False), ("dev", True)):
        EF._state.device_preprocess = devpre
        EF.process_images(src, str(tmp_path / tag), workers=2)
        outs[tag] = {f: torch.load(tmp_path / tag / "cat" / f) for f in sorted(os.listdir(tmp_path / tag / "cat"))}
    EF._state.device_preprocess = False
    assert len(outs["pil"]) == 7 and sorted(outs["pil"]) == sorted(outs["dev"])
    for f in outs["pil"]:
        assert torch.equal(outs["pil"][f], outs["dev"][f]), f


def test_llava_feature_dump_device_decode_equals_host_path(tmp_path, monkeypatch):
    import json
    from law_of_vision_representation_in_mllms_amd.llava.feature import extract as FE
    from law_of_vision_representation_in_mllms_amd.llava.model.llava_arch import build_function_mapping
    src = _write_tree(tmp_path, 5)
    entries = [{"image": f"cat/im{i}.jpg"} for i in range(5)] + [{"text": "no image"}]
    (tmp_path / "data.json").write_text(json.dumps(entries))
    monkeypatch.setenv("VISREP_SYNTHETIC_WEIGHTS", "1")
    tid = 'openai/clip-vit-large-patch14'
    margs = SimpleNamespace(vision_tower=tid, mm_vision_tower=tid, mm_vision_select_layer=-2, mm_vision_select_feature='patch', device=DEV)
    model = build_function_mapping[tid](margs)
    for aspect in ("pad", "square"):
        files = {}
        for tag, dd in (("host", False), ("dev", True)):
            dargs = SimpleNamespace(data_path=str(tmp_path / "data.json"), image_folder=src, image_aspect_ratio=aspect)
            targs = SimpleNamespace(feature_dir=str(tmp_path / f"{aspect}_{tag}"), per_device_train_batch_size=3)
            assert FE.inference(margs, dargs, targs, model=model, workers=2, device_decode=dd) == 5
            files[tag] = {f: torch.load(tmp_path / f"{aspect}_{tag}" / "cat" / f) for f in sorted(os.listdir(tmp_path / f"{aspect}_{tag}" / "cat"))}
        for f in files["host"]:
            assert torch.equal(files["host"][f], files["dev"][f]), (aspect, f)
