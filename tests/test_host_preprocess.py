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
