"""The restated Pillow resampling tables (device_preprocess.pil_coeffs) against `PIL.Image.resize` itself, bit for bit, on the CPU;
the device kernels consume exactly these tables (GPU parity: tests/test_gpu_dropin.py)."""
import numpy as np
import pytest
from PIL import Image

from law_of_vision_representation_in_mllms_amd import device_preprocess as DP


@pytest.mark.parametrize("w,h,ow,oh", [(640, 480, 224, 224), (500, 375, 448, 336), (333, 500, 336, 504), (100, 80, 224, 224), (768, 768, 224, 224),
                                       (17, 9, 5, 31), (224, 300, 224, 224), (1024, 683, 336, 336)])
def test_fixed_point_bicubic_equals_pil(w, h, ow, oh):
    rs = np.random.RandomState(w + h)
    a = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
    a[: h // 3] = rs.randint(0, 2, (h // 3, w, 3)) * 255            # hard edges: exercise the clamp of over/undershoot
    want = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BICUBIC))
    got = DP.resample_reference(a, (ow, oh))
    assert got.shape == want.shape and np.array_equal(got, want)


def test_coeff_tables_shape_and_sum():
    b, k, ks = DP.pil_coeffs(640, 224)
    assert b.shape == (224, 2) and k.shape == (224, ks) and ks == 13
    assert np.all(np.abs(k.sum(1) - (1 << 22)) <= ks)             # rows sum to 1.0 in 22-bit fixed point, up to rounding
    assert b[0, 0] == 0 and b[-1].sum() == 640


def test_c_path_loader_is_the_reference_expression(tmp_path):
    """extract_feature.py:65-67: resize((s, s)) -> PILToTensor -> (x / 255 - 0.5) * 2, bit for bit (the loader does it in numpy)."""
    import numpy as np
    import torch
    from PIL import Image
    from law_of_vision_representation_in_mllms_amd.C_score import extract_feature as EF
    rs = np.random.RandomState(4)
    Image.fromarray(rs.randint(0, 256, (90, 130, 3), dtype=np.uint8)).save(tmp_path / "a.jpg")
    img = Image.open(tmp_path / "a.jpg").convert('RGB').resize((56, 56))
    want = (torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1) / 255.0 - 0.5) * 2
    for load in (EF._load_pixels, EF._load_pixels_worker):                # calling-thread and decode-pool forms of the same expression
        got = load(str(tmp_path / "a.jpg"), 56)
        assert got.dtype == torch.float32 and got.shape == (3, 56, 56) and torch.equal(got, want)


def test_lanczos_bilinear_tables_and_geoaware_loader_match_reference():
    """Pillow's LANCZOS / BILINEAR resampling and the GeoAware-SC loader (utils_correspondence.resize: LANCZOS long side -> target,
    zero or edge padding) restated on the host, bit for bit against outputs of PIL / the reference function (georesize.npz)."""
    import os
    import numpy as np
    from law_of_vision_representation_in_mllms_amd import device_preprocess as DP
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "georesize.npz"))
    T = int(z["target"])
    for tag in ("land", "port", "square", "wide", "up"):
        a = z[f"{tag}.in"]
        assert np.array_equal(DP.resample_reference(a, (37, 29), "lanczos"), z[f"{tag}.lanczos"]), tag
        assert np.array_equal(DP.resample_reference(a, (37, 29), "bilinear"), z[f"{tag}.bilinear"]), tag
        for edge in (False, True):
            assert np.array_equal(DP.geoaware_resize_reference(a, T, edge), z[f"{tag}.edge{int(edge)}"]), (tag, edge)
    assert DP.geoaware_geometry(500, 375, 840) == ((840, 630), (105, 0)) and DP.geoaware_geometry(375, 500, 840) == ((630, 840), (0, 105))


def test_mm_utils_process_images_modes():
    """llava/mm_utils.py:64-95 semantics: 'pad' squares on a mean-colour canvas (centred on the short axis) and stacks; other modes take
    images[0] only - one [1, 3, H, W] batch per processor for a processor list, else the bare [3, H, W] tensor."""
    from types import SimpleNamespace
    import numpy as np
    import torch
    from PIL import Image
    from law_of_vision_representation_in_mllms_amd.llava import mm_utils as MU
    wide = Image.fromarray(np.full((4, 10, 3), 200, np.uint8))
    sq = MU.expand2square(wide, (1, 2, 3))
    a = np.asarray(sq)
    assert sq.size == (10, 10) and (a[3:7] == 200).all() and (a[:3] == (1, 2, 3)).all() and (a[7:] == (1, 2, 3)).all()
    tall = Image.fromarray(np.full((9, 4, 3), 50, np.uint8))
    b = np.asarray(MU.expand2square(tall, (0, 0, 0)))
    assert b.shape == (9, 9, 3) and (b[:, 2:6] == 50).all() and (b[:, :2] == 0).all() and (b[:, 6:] == 0).all()
    assert MU.expand2square(sq, (0, 0, 0)) is sq

    class Proc:
        image_mean = [0.5, 0.25, 0.0]

        def preprocess(self, img, return_tensors="pt"):
            return {"pixel_values": torch.from_numpy(np.asarray(img, np.float32)).permute(2, 0, 1)[None]}
    p = Proc()
    out = MU.process_images([wide, tall.resize((10, 3))], p, SimpleNamespace(image_aspect_ratio="pad"))
    assert out.shape == (2, 3, 10, 10) and out[0, :, 0, 0].tolist() == [127.0, 63.0, 0.0]
    ragged = MU.process_images([wide, tall], p, SimpleNamespace(image_aspect_ratio="pad"))
    assert isinstance(ragged, list) and ragged[0].shape == (3, 10, 10) and ragged[1].shape == (3, 9, 9)
    assert MU.process_images([wide, tall], p, SimpleNamespace()).shape == (3, 4, 10)
    lst = MU.process_images([wide, tall], [p, p], SimpleNamespace(image_aspect_ratio="square"))
    assert len(lst) == 2 and lst[0].shape == (1, 3, 4, 10)
