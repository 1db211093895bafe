"""DLPack / array-interface interop of the Python surface, modelled on the reference's
test/python/integration/test_dlpack_torch.py (import through an object that offers ONLY the DLPack protocol; export to torch)."""
import numpy as np
import pytest
import torch

from nvimagecodec_amd import api


class _DLPackOnly:
    """an object with nothing but the DLPack protocol (the reference test builds the same)"""

    def __init__(self, t):
        self.__dlpack__ = t.__dlpack__
        self.__dlpack_device__ = t.__dlpack_device__


@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.int16])
def test_dlpack_import_and_export_host(dtype):
    host = np.random.default_rng(1).integers(0, 128, (64, 48, 3)).astype(dtype)
    t = torch.from_numpy(host)
    for make in (api.as_image, api.from_dlpack):
        img = make(_DLPackOnly(t))
        assert img.shape == host.shape and img.dtype == np.dtype(dtype) and img.ndim == 3
        assert img.buffer_kind == api.ImageBufferKind.STRIDED_HOST
        assert (host == torch.from_dlpack(img).numpy()).all()
    cap = torch.utils.dlpack.to_dlpack(t)            # a bare PyCapsule
    img = api.from_dlpack(cap)
    assert (host == np.asarray(img.cpu()._array)).all()
    with pytest.raises(TypeError):
        api.from_dlpack(host.tolist())
    # zero copy: writing through the image is seen by the producer
    img = api.from_dlpack(_DLPackOnly(t))
    img._array[0, 0, 0] = 77
    assert t[0, 0, 0] == 77


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.int16])
def test_dlpack_import_and_export_device(dtype):
    host = np.random.default_rng(2).integers(0, 128, (640, 480, 3)).astype(dtype)
    dev = torch.as_tensor(host, device="cuda")
    img = api.as_image(_DLPackOnly(dev))
    assert img.shape == host.shape and img.dtype == np.dtype(dtype) and img.buffer_kind == api.ImageBufferKind.STRIDED_DEVICE
    assert (host == torch.from_dlpack(img).cpu().numpy()).all()
    img2 = api.from_dlpack(torch.utils.dlpack.to_dlpack(dev))
    assert img2.as_tensor().data_ptr() == dev.data_ptr()           # zero copy
    cap = api.as_image(dev).to_dlpack()
    assert (host == torch.from_dlpack(cap).cpu().numpy()).all()
    assert api.as_image(dev).__cuda_array_interface__["shape"] == host.shape


@pytest.mark.gpu
def test_decoded_image_round_trips_through_dlpack_into_the_encoder():
    import oracle
    from nvimagecodec_amd.synth import synth_image
    rgb = synth_image(131, 77, seed=4)
    jpeg = oracle.encode(rgb, "420", 90)
    with api.Decoder() as dec, api.Encoder() as enc:
        img = dec.decode(jpeg)
        again = api.from_dlpack(_DLPackOnly(torch.from_dlpack(img)))
        out = enc.encode(again, "jpeg", api.EncodeParams(quality=90, chroma_subsampling=api.ChromaSubsampling.CSS_420))
        assert out == oracle.encode(oracle.decode(jpeg), "420", 90)
