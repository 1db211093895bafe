"""The C-ABI boundary (no GPU needed): libhipjpeg_ext.so loads, exports every function declared in include/*.h, the
ctypes mirror has the reference's struct layout, and the extension entry point fills the descriptor the way
nvImageCodec's plugin framework expects (reference src/plugin_framework.cpp:309-351)."""
import ctypes as C
import os
import re

from conftest import ROOT
from nvimagecodec_amd import _native
from nvimagecodec_amd import abi as A


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(nvimgcodec[A-Z]\w+|hipjpeg[A-Z]\w+)\s*\(", text))
    # drop typedef'd function-pointer types and macros
    return {n for n in names if not n.endswith("_t") and not n.endswith("Func_t") and n != "hipjpegHandle"}


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_libraries_export_every_declared_symbol_and_nothing_of_each_others():
    """The extension module (what a real nvImageCodec loads) exports nvimgcodecExtensionModuleEntry + the hipjpeg* C-ABI and NO other
    nvimgcodec* name -- loaded next to a real libnvimgcodec it must not interpose the application API.  The application-side
    API of the test harness lives in libhipjpeg_host.so."""
    ext, host = _exported(_native.LIB_PATH), _exported(_native.HOST_LIB_PATH)
    hip_decl, nv_decl = _declared_functions("hipjpeg.h"), _declared_functions("nvimgcodec_abi.h")
    assert "nvimgcodecExtensionModuleEntry" in nv_decl and "hipjpegDecodeBatch" in hip_decl
    harness_only = {"hipjpegTestDoubleReports"}  # counted by the harness's future objects
    missing = [n for n in sorted(hip_decl - harness_only) if n not in ext]
    assert not missing, f"declared in include/hipjpeg.h but not exported by the extension: {missing}"
    assert [n for n in ext if re.match(r"nvimgcodec[A-Z]", n)] == ["nvimgcodecExtensionModuleEntry"]
    missing = [n for n in sorted(nv_decl - {"nvimgcodecExtensionModuleEntry"}) if n not in host]
    assert not missing, f"declared in include/nvimgcodec_abi.h but not exported by the host harness: {missing}"
    assert "nvimgcodecExtensionModuleEntry" not in host and harness_only <= host
    assert not [n for n in host if n.startswith("hipjpeg") and n not in harness_only]
    _native.load()
    _native.load_host()


def test_ctypes_mirror_matches_reference_layout():
    for cls, size in A.EXPECTED_SIZES.items():
        assert C.sizeof(cls) == size, cls.__name__
    assert A.ImageInfo.buffer.offset == 2208 and A.ImageInfo.cuda_stream.offset == 2232 and A.ImageInfo.plane_info.offset == 416
    assert A.DecoderDesc.canDecode.offset == 72 and A.DecoderDesc.decode.offset == 80
    assert A.FrameworkDesc.registerDecoder.offset == 80 and A.ExecutionParams.device_id.offset == 56


def test_extension_module_entry_contract():
    lib = A.bind(_native.load())
    assert lib.nvimgcodecExtensionModuleEntry(None) != A.STATUS_SUCCESS
    wrong = A.init(A.ExtensionDesc, A.ST_DECODER_DESC)
    assert lib.nvimgcodecExtensionModuleEntry(C.byref(wrong)) == A.STATUS_INVALID_PARAMETER
    d = A.init(A.ExtensionDesc, A.ST_EXTENSION_DESC)
    assert lib.nvimgcodecExtensionModuleEntry(C.byref(d)) == A.STATUS_SUCCESS
    assert d.id == b"hipjpeg_ext" and d.ext_api_version == 200 and d.struct_size == 64
    assert bool(d.create) and bool(d.destroy)


def test_extension_registers_decoder_into_a_foreign_framework():
    """Play the framework: hand our extension a FrameworkDesc made of Python callbacks and watch what it registers."""
    lib = A.bind(_native.load())
    d = A.init(A.ExtensionDesc, A.ST_EXTENSION_DESC)
    assert lib.nvimgcodecExtensionModuleEntry(C.byref(d)) == A.STATUS_SUCCESS
    seen = {}

    def reg_dec(inst, desc, prio):
        seen["dec"] = (desc.contents.id, desc.contents.codec, desc.contents.backend_kind, prio, desc.contents.struct_type, desc.contents.struct_size)
        return 0

    def unreg_dec(inst, desc):
        seen["unreg"] = desc.contents.id
        return 0

    cbs = dict(log=A.LogFn(lambda *a: 0), registerEncoder=A.RegisterEncoderFn(lambda i, d_, p: seen.setdefault("enc", (d_.contents.id, p)) and 0),
               unregisterEncoder=A.UnregisterEncoderFn(lambda i, d_: 0), registerDecoder=A.RegisterDecoderFn(reg_dec),
               unregisterDecoder=A.UnregisterDecoderFn(unreg_dec), registerParser=A.RegisterParserFn(lambda *a: 0),
               unregisterParser=A.UnregisterParserFn(lambda *a: 0))
    fw = A.init(A.FrameworkDesc, A.ST_FRAMEWORK_DESC, id=b"fake-framework", version=200, ext_api_version=200, cudart_version=0, **cbs)
    ext = C.c_void_p()
    assert d.create(d.instance, C.byref(ext), C.byref(fw)) == A.STATUS_SUCCESS
    assert seen["dec"][0] == b"hipjpeg_decoder" and seen["dec"][1] == b"jpeg"
    assert seen["dec"][2] == A.BACKEND_KIND_HYBRID_CPU_GPU and seen["dec"][3] == float(A.PRIORITY_HIGH)
    assert seen["dec"][4] == A.ST_DECODER_DESC and seen["dec"][5] == 88
    assert d.destroy(ext) == A.STATUS_SUCCESS
    assert seen["unreg"] == b"hipjpeg_decoder"   # destroy unregisters first (libjpeg_turbo_ext.cpp:34)
