"""Host half of the turn-to-turn reuse (llava_mi355x/reuse.py): row identities, longest common prefix, the two LRU caches.  CPU only — the package's
__init__ needs the built library (it loads; no GPU call is made)."""
import numpy as np


def _reuse():
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lmx_reuse", os.path.join(root, "llava-plus-codebase_amd", "llava_mi355x", "reuse.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


class _Holder:
    def __init__(self): self.closed = False
    def close(self): self.closed = True


def test_row_keys_identify_text_and_image_rows():
    R = _reuse()
    P = 4
    # gather plan of "t5 t9 <image0> t7 <image1>": token ids, then -2-k for feature row k
    src = np.array([5, 9, -2, -3, -4, -5, 7, -6, -7, -8, -9], dtype=np.int32)
    ha, hb = (0x1234567890abcdef, 0x0fedcba987654321), (0x1111111111111111, 0x2222222222222222)
    k1 = R.row_keys(src, [ha, hb], P)
    assert k1.dtype == np.int64 and (k1 >= 0).all()
    assert list(k1[[0, 1, 6]]) == [5, 9, 7]                       # text rows are their token ids
    assert len(set(k1.tolist())) == len(k1)                       # every image row has its own key
    assert ((k1[2:6] >> 62) & 1).all() and (k1[[0, 1, 6]] < (1 << 31)).all()
    # same images -> same keys; another image in slot 0 changes exactly its four rows
    assert np.array_equal(k1, R.row_keys(src, [ha, hb], P))
    k2 = R.row_keys(src, [hb, hb], P)
    assert (k2[2:6] != k1[2:6]).all() and np.array_equal(np.delete(k2, [2, 3, 4, 5]), np.delete(k1, [2, 3, 4, 5]))
    # the same image in two slots: patch p of slot 0 and patch p of slot 1 are the same feature row -> the same key
    assert np.array_equal(k2[2:6], k2[7:11])
    # padding rows (-1) cannot be keyed; a plan pointing past the hashed images neither
    assert R.row_keys(np.array([5, -1, 7], dtype=np.int32), [ha], P) is None
    assert R.row_keys(src, [ha], P) is None


def test_common_prefix():
    R = _reuse()
    a = np.array([1, 2, 3, 4, 5], dtype=np.int64)
    assert R.common_prefix(a, a) == 5
    assert R.common_prefix(a, a[:3]) == 3
    assert R.common_prefix(a, np.array([1, 2, 9, 4, 5], dtype=np.int64)) == 2
    assert R.common_prefix(a, np.array([7], dtype=np.int64)) == 0
    assert R.common_prefix(a, np.array([], dtype=np.int64)) == 0


def test_prefix_cache_takes_the_longest_match_and_evicts_lru():
    R = _reuse()
    pc = R.PrefixCache(capacity=2, min_rows=3)
    h = [_Holder() for _ in range(4)]
    pc.put(np.arange(10), h[0])                                   # 0..9
    pc.put(np.concatenate([np.arange(6), [99, 98]]), h[1])        # 0..5, 99, 98
    got, n = pc.take(np.concatenate([np.arange(6), [99, 7, 7]]))
    assert got is h[1] and n == 7 and len(pc) == 1                # longest match wins and leaves the cache
    got, n = pc.take(np.array([0, 1, 55]))
    assert got is None and n == 0 and len(pc) == 1                # 2 common rows < min_rows: the entry stays
    pc.put(np.arange(4), h[2]); pc.put(np.arange(5), h[3])        # capacity 2: the oldest entry (h[0]) is closed
    assert h[0].closed and not h[2].closed and len(pc) == 2
    pc.clear()
    assert h[2].closed and h[3].closed and len(pc) == 0
    assert pc.hits == 1 and pc.misses == 1


def test_prefix_cache_leaves_an_entry_to_its_own_conversation():
    """A request that shares only the system prompt with a finished conversation (40 of 700 rows) does not take that conversation's sequence: the turn that
    continues it still finds all of it."""
    R = _reuse()
    pc = R.PrefixCache(capacity=4, min_rows=32)
    h = _Holder()
    conv = np.concatenate([np.arange(40), 1000 + np.arange(660)])                 # system prompt, then image rows + question + answer
    pc.put(conv, h)
    other = np.concatenate([np.arange(40), 5000 + np.arange(600)])                # another conversation: same system prompt, another image
    got, n = pc.take(other)
    assert got is None and n == 0 and len(pc) == 1 and not h.closed
    got, n = pc.take(np.concatenate([conv, 9000 + np.arange(80)]))                # the second turn of the first conversation
    assert got is h and n == 700 and len(pc) == 0
    # the same image asked another question: most of the entry is reused
    pc.put(conv, h)
    got, n = pc.take(np.concatenate([conv[:620], 7000 + np.arange(30)]))
    assert got is h and n == 620


def test_image_feature_cache_lru():
    R = _reuse()
    c = R.ImageFeatureCache(2)
    c.put((1, 1), "a"); c.put((2, 2), "b")
    assert c.get((1, 1)) == "a"                                   # refreshes (1, 1)
    c.put((3, 3), "c")                                            # evicts (2, 2)
    assert c.get((2, 2)) is None and c.get((1, 1)) == "a" and c.get((3, 3)) == "c"
    assert c.hits == 3 and c.misses == 1
