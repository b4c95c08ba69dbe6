"""ORBextractor: host-side mirror of the reference class (include/ORBextractor.h:52-168)
over the C ABI.  Same constructor arguments, same getters, `__call__` = operator()."""
import ctypes as C

import numpy as np

from ._lib import (GFO_STAGE_MAX, KEYPOINT_DTYPE, Params, StageTime, check, load_library, ptr)


class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) -- ORBextractor.h:81-82.

    extractor(image) -> (keypoints, descriptors): keypoints is a structured array laid out like
    cv::KeyPoint, descriptors an (N, 32) uint8 array (operator(), ORBextractor.h:89-91).
    """

    def __init__(self, nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, device=0, max_batch=1, combining=False):
        """combining=True opts the per-frame entry points (`__call__` on one image, `extract_stereo`) into the frame combiner
        (gfo_ctx_set_combining): frames that several threads submit at once through several such extractors share one
        device batch.  Same results, bit for bit; the pyramid / debug hooks of a combined call are not available."""
        self._L = load_library()
        self._ctx = C.c_void_p()
        prm = Params(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_batch)
        check(self._L, None, self._L.gfo_ctx_create(C.byref(prm), device, C.byref(self._ctx)))
        self.nfeatures, self.scaleFactor, self.nlevels = nfeatures, scaleFactor, nlevels
        self.iniThFAST, self.minThFAST = iniThFAST, minThFAST
        s = np.zeros(nlevels, np.float32); si = np.zeros(nlevels, np.float32)
        g = np.zeros(nlevels, np.float32); gi = np.zeros(nlevels, np.float32)
        q = np.zeros(nlevels, np.int32)
        check(self._L, self._ctx, self._L.gfo_ctx_tables(self._ctx, ptr(s), ptr(si), ptr(g), ptr(gi), ptr(q)))
        self._tables = (s, si, g, gi, q)
        self._last_shape = None
        if combining:
            check(self._L, self._ctx, self._L.gfo_ctx_set_combining(self._ctx, 1))

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._L.gfo_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # getters, ORBextractor.h:93-119
    def GetLevels(self): return self.nlevels
    def GetScaleFactor(self): return self.scaleFactor
    def GetInitThres(self): return self.iniThFAST
    def GetMinThres(self): return self.minThFAST
    def GetScaleFactors(self): return self._tables[0].copy()
    def GetInverseScaleFactors(self): return self._tables[1].copy()
    def GetScaleSigmaSquares(self): return self._tables[2].copy()
    def GetInverseScaleSigmaSquares(self): return self._tables[3].copy()
    @property
    def mnFeaturesPerLevel(self): return self._tables[4].copy()

    @property
    def handle(self):
        return self._ctx

    @property
    def ctx_id(self):
        return int(self._L.gfo_ctx_id(self._ctx))

    def set_combining(self, on=True):
        check(self._L, self._ctx, self._L.gfo_ctx_set_combining(self._ctx, 1 if on else 0))

    def combiner_stats(self):
        """(device batches, requests) the frame combiner's engine of this extractor has served"""
        b, r = C.c_int64(), C.c_int64()
        check(self._L, self._ctx, self._L.gfo_combiner_stats(self._ctx, C.byref(b), C.byref(r)))
        return b.value, r.value

    def combiner_counters(self):
        """gfo_combiner_counters as a dict (frame combiner engine + this context's stereo rig)"""
        v = (C.c_int64 * 8)()
        check(self._L, self._ctx, self._L.gfo_combiner_counters(self._ctx, v, 8))
        names = ("batches", "requests", "batches_redone", "slots_prepared", "engine_broken", "rig_frames", "rig_served", "rig_alone")
        return dict(zip(names, (int(x) for x in v)))

    def pair_with(self, right, params):
        """gfo_ctx_pair: this extractor and `right` are the left / right extractor of one stereo rig (None dissolves it)"""
        from ._lib import StereoParamsC
        p = StereoParamsC(*params) if params is not None else None
        rc = self._L.gfo_ctx_pair(self._ctx, right._ctx if right is not None else None, C.byref(p) if p is not None else None)
        check(self._L, self._ctx, rc)

    def max_keypoints(self):
        return self._L.gfo_ctx_max_keypoints(self._ctx)

    def set_stream(self, hip_stream):
        check(self._L, self._ctx, self._L.gfo_ctx_set_stream(self._ctx, C.c_void_p(hip_stream)))

    def synchronize(self):
        check(self._L, self._ctx, self._L.gfo_ctx_synchronize(self._ctx))

    STAGE_PYRAMID, STAGE_FAST, STAGE_SELECT, STAGE_DESCRIPTORS = 1, 2, 3, 4

    def chain_after(self, other, stage=1):
        """gfo_ctx_chain: every extraction submitted to this extractor starts on the device only after the extraction most
        recently submitted to `other` has finished `stage` (None removes the edge).  For applications that alternate
        batches between extractors; changes no result."""
        h = other.handle if other is not None else None
        check(self._L, self._ctx, self._L.gfo_ctx_chain(self._ctx, h, int(stage)))

    # operator()
    def __call__(self, image, mask=None):
        if image is None or image.size == 0:
            return np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
        kps, descs = self.extract_batch([image])
        return kps[0], descs[0]

    def extract_batch(self, images):
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        h, w = imgs[0].shape
        assert all(im.shape == (h, w) for im in imgs), "one batch = one image size"
        n = len(imgs)
        cap = max(self.max_keypoints(), self.nfeatures + 64)
        while True:
            kp = np.zeros((n, cap), KEYPOINT_DTYPE)
            desc = np.zeros((n, cap, 32), np.uint8)
            cnt = np.zeros(n, np.int32)
            arr = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
            rc = self._L.gfo_extract_batch(self._ctx, arr, n, w, h, w, ptr(kp), ptr(desc), cap, ptr(cnt))
            if rc == -3:  # capacity: the geometry is planned now, retry with the exact bound
                cap = max(int(cnt.max()), self.max_keypoints())
                continue
            check(self._L, self._ctx, rc)
            break
        self._last_shape = (h, w)
        return [kp[i, :cnt[i]].copy() for i in range(n)], [desc[i, :cnt[i]].copy() for i in range(n)]

    def extract_stereo(self, left, right, params):
        """One stereo frame in one submission (gfo_extract_stereo): the reference's stereo Frame constructor body --
        ExtractORB x2 (Frame.cc:84-87) + ComputeStereoMatches_Undistorted (:1167-1316).
        Returns (kp_l, desc_l, kp_r, desc_r, nmatched, mvuRight, mvDepth, best_dist, best_idx_r)."""
        from ._lib import StereoParamsC
        left = np.ascontiguousarray(left, np.uint8)
        right = np.ascontiguousarray(right, np.uint8)
        h, w = left.shape
        assert right.shape == (h, w)
        cap = max(self.max_keypoints(), self.nfeatures + 64)
        p = StereoParamsC(*params)
        while True:
            kl = np.zeros(cap, KEYPOINT_DTYPE); kr = np.zeros(cap, KEYPOINT_DTYPE)
            dl = np.zeros((cap, 32), np.uint8); dr = np.zeros((cap, 32), np.uint8)
            u = np.zeros(cap, np.float32); dp = np.zeros(cap, np.float32)
            bd = np.zeros(cap, np.int32); bi = np.zeros(cap, np.int32)
            nl, nr, nm = C.c_int(), C.c_int(), C.c_int()
            rc = self._L.gfo_extract_stereo(self._ctx, ptr(left), ptr(right), w, h, w, C.byref(p), ptr(kl), ptr(dl), ptr(kr), ptr(dr), cap,
                                            C.byref(nl), C.byref(nr), ptr(u), ptr(dp), ptr(bd), ptr(bi), C.byref(nm))
            if rc == -3:
                cap = max(nl.value, nr.value, self.max_keypoints())
                continue
            check(self._L, self._ctx, rc)
            break
        self._last_shape = (h, w)
        a, b = nl.value, nr.value
        return kl[:a].copy(), dl[:a].copy(), kr[:b].copy(), dr[:b].copy(), nm.value, u[:a].copy(), dp[:a].copy(), bd[:a].copy(), bi[:a].copy()

    # device-resident path (bench / chained stereo)
    def extract_batch_device(self, dev_ptr, nimg, w, h, pitch=None, img_stride=None):
        pitch = pitch or w
        img_stride = img_stride or pitch * h
        check(self._L, self._ctx, self._L.gfo_extract_batch_device(self._ctx, C.c_void_p(dev_ptr), nimg, w, h, pitch, img_stride))
        self._last_shape = (h, w)
        self._last_n = nimg

    def batch_counts(self, nimg, per_level=False):
        n = np.zeros(nimg, np.int32)
        pl = np.zeros((nimg, self.nlevels), np.int32) if per_level else None
        check(self._L, self._ctx, self._L.gfo_batch_counts(self._ctx, ptr(n), ptr(pl)))
        return (n, pl) if per_level else n

    def batch_fetch(self, image):
        cap = self.max_keypoints()
        kp = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        check(self._L, self._ctx, self._L.gfo_batch_fetch(self._ctx, image, ptr(kp), ptr(desc), cap, C.byref(n)))
        return kp[:n.value].copy(), desc[:n.value].copy()

    def batch_deliver(self, host_ptr=None, host_bytes=0):
        """gfo_batch_deliver: queue the copy of everything the last batch produced (counts, keypoints, descriptors, stereo
        outputs) into ONE pinned host block; returns its layout.  host_ptr=None: layout only (`bytes` = size to allocate).
        Asynchronous: deliver_wait() blocks until the block is complete."""
        from ._lib import DeliveryC
        lay = DeliveryC()
        check(self._L, self._ctx, self._L.gfo_batch_deliver(self._ctx, C.c_void_p(host_ptr) if host_ptr else None, host_bytes, C.byref(lay)))
        return lay

    def deliver_wait(self):
        check(self._L, self._ctx, self._L.gfo_deliver_wait(self._ctx))

    @staticmethod
    def delivered_views(block, lay):
        """numpy views into a delivered block (a uint8 array over the pinned memory): counts, keypoints [nimg][stride],
        descriptors [nimg][stride][32] and, when the batch was stereo-matched, (u_right, depth, best_dist, best_idx, nmatched)"""
        n, ks, npair = lay.nimg, lay.kp_stride, lay.nimg // 2
        def arr(off, dtype, shape):
            cnt = int(np.prod(shape))
            return block[off:off + cnt * np.dtype(dtype).itemsize].view(dtype).reshape(shape)
        out = {"flags": arr(lay.off_flags, np.int32, (4,)), "counts": arr(lay.off_counts, np.int32, (n,)),
               "kp": arr(lay.off_kp, KEYPOINT_DTYPE, (n, ks)), "desc": arr(lay.off_desc, np.uint8, (n, ks, 32))}
        if lay.stereo:
            out.update(u_right=arr(lay.off_u_right, np.float32, (npair, ks)), depth=arr(lay.off_depth, np.float32, (npair, ks)),
                       best_dist=arr(lay.off_best_dist, np.int32, (npair, ks)), best_idx=arr(lay.off_best_idx, np.int32, (npair, ks)),
                       nmatched=arr(lay.off_nmatched, np.int32, (npair,)))
        return out

    # ComputePyramid / mvImagePyramid, ORBextractor.h:127-132
    def ComputePyramid(self, image):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = image.shape
        check(self._L, self._ctx, self._L.gfo_compute_pyramid(self._ctx, ptr(image), w, h, w))
        self._last_shape = (h, w)

    def pyramid_level(self, level, image=0, border=0):
        h0, w0 = self._last_shape
        out = np.zeros((h0 + 2 * border, w0 + 2 * border), np.uint8)
        w, h = C.c_int(), C.c_int()
        check(self._L, self._ctx, self._L.gfo_pyramid_level(self._ctx, image, level, border, ptr(out), out.shape[1], C.byref(w), C.byref(h)))
        return out[:h.value + 2 * border, :w.value + 2 * border].copy()

    @property
    def mvImagePyramid(self):
        """The reference's public member: levels with their 19-px reflect frame removed (the ROI
        view the reference stores, ORBextractor.cc:1184)."""
        return [self.pyramid_level(l) for l in range(self.nlevels)]

    # inspection hooks used by the parity tests
    def debug_blurred_level(self, level, image=0):
        lv = self.pyramid_level(level, image)
        out = np.zeros_like(lv)
        check(self._L, self._ctx, self._L.gfo_debug_blurred_level(self._ctx, image, level, ptr(out), out.shape[1]))
        return out

    def debug_level_candidates(self, level, image=0):
        n = C.c_int()
        check(self._L, self._ctx, self._L.gfo_debug_level_candidates(self._ctx, image, level, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.int32)
        check(self._L, self._ctx, self._L.gfo_debug_level_candidates(self._ctx, image, level, ptr(out), n.value, C.byref(n)))
        return out[:n.value]

    # measurement hooks
    def profile_enable(self, on=True):
        check(self._L, self._ctx, self._L.gfo_profile_enable(self._ctx, 1 if on else 0))

    def profile_read(self, reset=True):
        arr = (StageTime * GFO_STAGE_MAX)()
        n = C.c_int()
        check(self._L, self._ctx, self._L.gfo_profile_read(self._ctx, arr, GFO_STAGE_MAX, C.byref(n), 1 if reset else 0))
        return {arr[i].name.decode(): (arr[i].ms, arr[i].launches) for i in range(n.value)}
