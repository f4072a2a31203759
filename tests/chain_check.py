"""Checker for morb_slam_amd.frontend.StereoFrontEnd (TEST INFRASTRUCTURE: uses the CPU oracle; imported by tests/ and by
bench.py's post-region self-check only — never by the product path).

verify_frames compares sampled stereo frames of one buffer set with the oracle, field by field:
keypoint records + descriptors of both images (ORBextractor.cc:1006-1086), mvuRight / mvDepth bit patterns
(Frame.cc:889-1047), BoW word + node ids (TemplatedVocabulary.h:1218-1259), the SearchByBoW table and count
(ORBmatcher.cc:218-395)."""
import numpy as np

import oracle_lib as O


def verify_frames(fe, S, frames, host_images):
    """fe: StereoFrontEnd on one rank (no exchange); S: one of its buffer sets after a step; frames: local frame indices;
    host_images: uint8 array [2 B, H, W] = what fe.images holds.  Returns the number of frames verified; AssertionError on the
    first difference."""
    from morb_slam_amd.capi import KP_DTYPE
    assert fe.exch is None, "multi-rank chains are checked by tests/test_parallel_gpu.py"
    fe.sync()
    kps, desc, cnt, mono = (t.cpu().numpy() for t in S.out)
    uR, dep = (t.cpu().numpy() for t in S.st_out)
    word, node = (t.cpu().numpy() for t in S.bow_out)
    mt, nm = (t.cpu().numpy() for t in S.match_out)
    nfeat = fe.exts[0].nfeatures
    memo = {}

    def ora(img):
        if img not in memo:
            o = O.OracleExtractor(nfeat, 1.2, 8, 20, 7)
            mo, ko, do = o(host_images[img])
            memo[img] = (o, mo, ko, do)
        return memo[img]
    vd, vf = fe.voc_host
    for f in frames:
        for img in (2 * f, 2 * f + 1):
            _, mo, ko, do = ora(img)
            assert cnt[img] == len(ko) and mono[img] == mo, f"frame {f} image {img}: {cnt[img]} keypoints vs oracle {len(ko)}"
            assert kps[img, :cnt[img]].reshape(-1).view(KP_DTYPE).tobytes() == ko.tobytes(), f"frame {f} image {img}: keypoint records differ"
            assert np.array_equal(desc[img, :cnt[img]], do), f"frame {f} image {img}: descriptors differ"
        ol, _, kl, dl = ora(2 * f)
        orr, _, kr, dr = ora(2 * f + 1)
        ue, de = O.stereo_matches(ol, orr, kl, dl, kr, dr, np.float32(fe.mbf), np.float32(fe.mb))
        n = len(kl)
        assert uR[f, :n].tobytes() == ue.tobytes(), f"frame {f}: mvuRight differs"
        assert dep[f, :n].tobytes() == de.tobytes(), f"frame {f}: mvDepth differs"
        we, ne = O.bow_transform(dl, vd, vf, fe.VK, fe.VL, fe.VUP)
        assert np.array_equal(word[2 * f, :n], we) and np.array_equal(node[2 * f, :n], ne), f"frame {f}: BoW word / node ids differ"
        # SearchByBoW(previous frame as the reference keyframe, this frame)
        p = int(np.nonzero(fe.f_host == 2 * f)[0][0])
        a = int(fe.kf_host[p])
        _, _, ka, da = ora(a)
        _, na = O.bow_transform(da, vd, vf, fe.VK, fe.VL, fe.VUP)
        cnt_e, me = O.search_by_bow(da, ka["angle"], fe.has_mp_host[a, :len(ka)], na, dl, kl["angle"], ne, 0.7, True)
        assert nm[p] == cnt_e, f"frame {f}: SearchByBoW count {nm[p]} vs oracle {cnt_e}"
        assert np.array_equal(mt[p, :n], me), f"frame {f}: SearchByBoW table differs"
    return len(frames)
