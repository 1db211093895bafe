"""A scriptable fake decoder plugin that talks to the framework through the real C function tables (ctypes callbacks).
Same idea as the reference's MockDecoderPlugin (test/api/can_decode_test.cpp:45-145)."""
import ctypes as C

from nvimagecodec_amd import abi as A


class FakeDecoderPlugin:
    """canDecode returns `can_status` (an int, or a callable(index, cs_info, img_info) -> int); decode() fills the
    output buffer with `fill` and reports `decode_status` through imageReady."""

    def __init__(self, plugin_id, backend_kind=A.BACKEND_KIND_CPU_ONLY, priority=A.PRIORITY_NORMAL, can_status=A.PS_SUCCESS,
                 decode_status=A.PS_SUCCESS, fill=0x5A, create_status=A.STATUS_SUCCESS):
        self.plugin_id = plugin_id.encode()
        self.priority = priority
        self.can_status = can_status
        self.decode_status = decode_status
        self.fill = fill
        self.create_status = create_status
        self.log = []          # ("create"|"canDecode"|"decode"|"destroy", batch_size)
        self.seen_options = None
        self.seen_device = None
        self._cbs = [A.DecoderCreateFn(self._create), A.DecoderDestroyFn(self._destroy), A.CanDecodeFn(self._can), A.DecodeFn(self._decode)]
        self.desc = A.init(A.DecoderDesc, A.ST_DECODER_DESC, id=self.plugin_id, codec=b"jpeg", backend_kind=backend_kind,
                           create=self._cbs[0], destroy=self._cbs[1], canDecode=self._cbs[2], decode=self._cbs[3])
        self._ext_cbs = [A.ExtensionCreateFn(self._ext_create), A.ExtensionDestroyFn(self._ext_destroy)]
        self.ext_desc = A.init(A.ExtensionDesc, A.ST_EXTENSION_DESC, id=self.plugin_id + b"_ext", version=100, ext_api_version=200,
                               create=self._ext_cbs[0], destroy=self._ext_cbs[1])
        self.framework = None

    # ---- extension
    def _ext_create(self, instance, out_ext, fw):
        self.framework = fw.contents
        st = self.framework.registerDecoder(self.framework.instance, C.byref(self.desc), C.c_float(self.priority))
        out_ext[0] = 0x1234
        return st

    def _ext_destroy(self, ext):
        self.framework.unregisterDecoder(self.framework.instance, C.byref(self.desc))
        return A.STATUS_SUCCESS

    # ---- decoder
    def _create(self, instance, out_decoder, exec_params, options):
        self.log.append(("create", 0))
        self.seen_options = options
        self.seen_device = exec_params.contents.device_id
        if self.create_status != A.STATUS_SUCCESS:
            return self.create_status
        out_decoder[0] = 0xBEEF
        return A.STATUS_SUCCESS

    def _destroy(self, decoder):
        self.log.append(("destroy", 0))
        return A.STATUS_SUCCESS

    def _infos(self, cs, im):
        ci = A.init(A.ImageInfo, A.ST_IMAGE_INFO)
        cs.contents.getImageInfo(cs.contents.instance, C.byref(ci))
        ii = A.init(A.ImageInfo, A.ST_IMAGE_INFO)
        im.contents.getImageInfo(im.contents.instance, C.byref(ii))
        return ci, ii

    def _can(self, decoder, status, code_streams, images, n, params):
        self.log.append(("canDecode", n))
        for i in range(n):
            if callable(self.can_status):
                ci, ii = self._infos(code_streams[i], images[i])
                status[i] = self.can_status(i, ci, ii)
            else:
                status[i] = self.can_status
        return A.STATUS_SUCCESS

    def _decode(self, decoder, code_streams, images, n, params):
        self.log.append(("decode", n))
        for i in range(n):
            ci, ii = self._infos(code_streams[i], images[i])
            st = self.decode_status(i) if callable(self.decode_status) else self.decode_status
            if st == A.PS_SUCCESS and ii.buffer_kind == A.BUFFER_KIND_STRIDED_HOST:
                size = sum(ii.plane_info[p].row_stride * ii.plane_info[p].height for p in range(ii.num_planes))
                C.memset(ii.buffer, self.fill, size)
            images[i].contents.imageReady(images[i].contents.instance, st)
        return A.STATUS_SUCCESS

    def count(self, what):
        return sum(1 for w, _ in self.log if w == what)
