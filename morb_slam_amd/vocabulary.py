"""DBoW2 vocabulary for the BoW descent (Frame::ComputeBoW): text-format loader (TemplatedVocabulary::loadFromTextFile,
Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1420) through the C ABI, and a writer of the same format
(saveToTextFile :1424-1460) used by the tests — ORBvoc.txt itself is a missing blob of the reference checkout."""
import ctypes as C

import numpy as np

from .capi import check, lib, ptr


class Vocabulary:
    def __init__(self, k, L, nodeDesc, firstChild, childCount, wordId, weight):
        self.k, self.L = k, L
        self.nodeDesc, self.firstChild, self.childCount, self.wordId, self.weight = nodeDesc, firstChild, childCount, wordId, weight

    @property
    def nNodes(self):
        return len(self.firstChild)

    @staticmethod
    def load_text(path):
        Lb = lib()
        h = C.c_void_p()
        check(Lb.morb_vocabulary_load_text(str(path).encode(), C.byref(h)))
        try:
            k, L, n, nw = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            check(Lb.morb_vocabulary_info(h, C.byref(k), C.byref(L), C.byref(n), C.byref(nw)))
            a = (np.zeros((n.value, 32), np.uint8), np.zeros(n.value, np.int32), np.zeros(n.value, np.int32), np.zeros(n.value, np.int32),
                 np.zeros(n.value, np.float32))
            check(Lb.morb_vocabulary_arrays(h, *[ptr(x) for x in a]))
            w64 = np.zeros(n.value, np.float64); sc, wt = C.c_int(), C.c_int()
            check(Lb.morb_vocabulary_weights(h, ptr(w64), C.byref(sc), C.byref(wt)))
        finally:
            Lb.morb_vocabulary_destroy(h)
        v = Vocabulary(k.value, L.value, *a)
        v.weight64, v.scoring, v.weighting = w64, sc.value, wt.value   # WordValue is double in DBoW2
        return v


def save_text(path, k, L, parent, is_leaf, desc, weight, scoring=0, weighting=0):
    """Nodes 1 .. n in id order: 'parent isLeaf d0 .. d31 weight' (node 0, the root, is implicit)."""
    with open(path, "w") as f:
        f.write(f"{k} {L} {scoring} {weighting}\n")
        for i in range(len(parent)):
            f.write(f"{int(parent[i])} {int(is_leaf[i])} " + " ".join(str(int(b)) for b in desc[i]) + f" {float(weight[i]):.6g}\n")
