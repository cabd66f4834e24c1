"""ORBVocabulary / LineVocabulary (include/ORBVocabulary.h:30-34 = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) on the GPU.

    voc = ORBVocabulary(); voc.loadFromTextFile("ORBvoc.txt")            # src/System.cc:68-77
    bow, fv = voc.transform(descriptors, 4)                               # Frame::ComputeBoW, src/Frame.cc:585-597

bow : {word id: value} (DBoW2::BowVector), fv : {node id: [feature indices]} (DBoW2::FeatureVector); both in ascending key order.
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import check, lib, ptr

L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = range(6)
TF_IDF, TF, IDF, BINARY = range(4)


class ORBVocabulary:
    def __init__(self, context=None):
        self._h = C.c_void_p()
        self._context = context

    def __del__(self):
        self.clear()

    def clear(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().olf_voc_destroy(self._h)
            self._h = C.c_void_p()

    def _ctx(self):
        if self._context is None:
            self._context = _lib.Context(_lib.default_params(), 640, 480, 1)
        return self._context

    def loadFromTextFile(self, filename):
        """bool TemplatedVocabulary::loadFromTextFile(const std::string&), Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1425"""
        self.clear()
        h = C.c_void_p()
        rc = lib().olf_voc_load_text(str(filename).encode(), C.byref(h))
        if rc == _lib.OLF_ERR_INVALID:
            return False                     # the reference prints a message and returns false on a malformed header
        check(rc, "olf_voc_load_text")
        self._h = h
        return True

    @classmethod
    def from_arrays(cls, k, L, parent, is_leaf, desc, weight, scoring=L1_NORM, weighting=TF_IDF, context=None):
        self = cls(context)
        parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(len(parent), 32); weight = np.ascontiguousarray(weight, np.float64)
        check(lib().olf_voc_create(int(k), int(L), int(scoring), int(weighting), len(parent), ptr(parent), ptr(is_leaf), ptr(desc), ptr(weight),
                                   C.byref(self._h)), "olf_voc_create")
        return self

    def info(self):
        v = [C.c_int() for _ in range(6)]
        check(lib().olf_voc_info(self._h, *[C.byref(x) for x in v]), "olf_voc_info")
        return dict(zip(("k", "L", "scoring", "weighting", "n_nodes", "n_words"), (x.value for x in v)))

    def size(self):
        return self.info()["n_words"] if self._h.value else 0

    def empty(self):
        return self.size() == 0

    def transform(self, features, levelsup=4):
        """void transform(const vector<TDescriptor>&, BowVector&, FeatureVector&, int levelsup), TemplatedVocabulary.h:1127-1195"""
        d = np.ascontiguousarray(features, np.uint8).reshape(-1, 32)
        n = len(d)
        if not self._h.value or n == 0:
            return {}, {}
        ids, vals = np.zeros(n, np.int32), np.zeros(n, np.float64)
        nodes, offs, idx = np.zeros(n, np.int32), np.zeros(n + 1, np.int32), np.zeros(n, np.int32)
        nb, nf = C.c_int(), C.c_int()
        check(lib().olf_bow_transform(self._ctx().handle, self._h, ptr(d), n, int(levelsup), ptr(ids), ptr(vals), C.byref(nb), ptr(nodes), ptr(offs),
                                      ptr(idx), C.byref(nf)), "olf_bow_transform")
        bow = {int(ids[i]): float(vals[i]) for i in range(nb.value)}
        fv = {int(nodes[a]): idx[offs[a]:offs[a + 1]].tolist() for a in range(nf.value)}
        return bow, fv


LineVocabulary = ORBVocabulary
