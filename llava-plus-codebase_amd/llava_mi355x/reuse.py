"""Reuse across the turns of one conversation.

The LLaVA-Plus tool loop (llava/serve/gradio_web_server_llava_plus.py:498-637) answers a question in two `generate` calls: the second prompt is the first
one — system text, the SAME image, the question — followed by the model's own first answer and the tool's output.  The reference worker
(llava/serve/model_worker.py:122-192) re-encodes the image and re-prefills every row.  Two host-side caches in front of the engine avoid that:

* ImageFeatureCache — `encode_images` keyed by a 128-bit content hash of each image's pixel tensor (lmx_op_hash128 on the device, 16 bytes read back): pixels
  already seen return the rows the tower + projector produced for them (the same bits: they ARE that earlier output).
* PrefixCache — finished requests donate their sequence (KV cache) together with the identity of every row it holds: token id for a text row,
  (image hash, patch index) for an image row, then the ids generated on top.  A new request takes the entry with the longest common row prefix, the engine
  forgets everything behind that prefix (lmx_seq_truncate) and prefills only the rows that follow — arithmetically a chunked prefill whose first chunk ran
  earlier.  Entries are taken, not shared: a sequence serves one request at a time.

Both are off unless switched on (`LlavaLlamaForCausalLM.enable_reuse`, LLAVA_MI355X_REUSE=1 through load_pretrained_model); bench.py never enables them
(a cached tower output inside its timed region would be work skipped)."""
import threading
from collections import OrderedDict
from typing import List, Optional, Tuple

import numpy as np

_IMG_FLAG = np.uint64(1) << np.uint64(62)
_MASK62 = (np.uint64(1) << np.uint64(62)) - np.uint64(1)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def row_keys(src_row: np.ndarray, image_hashes: List[Tuple[int, int]], tokens_per_image: int) -> Optional[np.ndarray]:
    """Identity of every spliced prompt row.  src_row: one row of lmx_splice_plan's gather plan (>= 0 token id, -1 zero row, -2-k row k of the flattened
    image-feature matrix); image_hashes[i] = (h1, h2) of the i-th encoded image (feature rows [i P, (i + 1) P)).  Returns int64 keys — token ids stay
    themselves, an image row is bit 62 | mix(hash, patch) — or None when the row holds padding (no reuse for padded batches)."""
    src = np.asarray(src_row, dtype=np.int64)
    if (src == -1).any():
        return None
    keys = src.copy()
    img = src <= -2
    if img.any():
        if tokens_per_image <= 0:
            return None
        k = (-2 - src[img]).astype(np.int64)
        which, patch = k // tokens_per_image, k % tokens_per_image
        if which.max() >= len(image_hashes):
            return None
        h1 = np.asarray([image_hashes[i][0] for i in range(len(image_hashes))], dtype=np.uint64)
        h2 = np.asarray([image_hashes[i][1] for i in range(len(image_hashes))], dtype=np.uint64)
        with np.errstate(over="ignore"):
            mixed = (h1[which] ^ ((patch.astype(np.uint64) + np.uint64(1)) * _GOLD) ^ (h2[which] >> np.uint64(7))) & _MASK62
        keys[img] = (mixed | _IMG_FLAG).astype(np.int64)
    return keys


def common_prefix(a: np.ndarray, b: np.ndarray) -> int:
    m = min(len(a), len(b))
    if m == 0:
        return 0
    neq = np.nonzero(a[:m] != b[:m])[0]
    return int(neq[0]) if len(neq) else m


class ImageFeatureCache:
    """LRU: (h1, h2) -> device tensor [tokens_per_image, hidden] (a private copy of the rows encode_images produced)."""

    def __init__(self, capacity: int):
        self.capacity = int(capacity)
        self._d: "OrderedDict[Tuple[int, int], object]" = OrderedDict()
        self._lock = threading.Lock()
        self.hits = 0
        self.misses = 0

    def get(self, key):
        with self._lock:
            v = self._d.get(key)
            if v is not None:
                self._d.move_to_end(key)
                self.hits += 1
            else:
                self.misses += 1
            return v

    def put(self, key, value) -> None:
        with self._lock:
            self._d[key] = value
            self._d.move_to_end(key)
            while len(self._d) > self.capacity:
                self._d.popitem(last=False)

    def clear(self) -> None:
        with self._lock:
            self._d.clear()


class PrefixCache:
    """LRU of (row keys, sequence holder).  take() removes the entry it returns; put() may evict (and close) the oldest."""

    def __init__(self, capacity: int, min_rows: int = 32):
        self.capacity = int(capacity)
        self.min_rows = int(min_rows)
        self._entries: List[Tuple[np.ndarray, object]] = []       # most recently used last
        self._lock = threading.Lock()
        self.hits = 0
        self.misses = 0
        self.rows_reused = 0

    def take(self, keys: np.ndarray):
        """(holder, common prefix length) of the entry sharing the longest row prefix with `keys`, else (None, 0).  A match counts from min_rows on AND only when
        it covers at least half of the entry: an entry is the context of ONE conversation, and a new conversation that merely shares the system prompt with it
        (a few dozen rows of several hundred) must not take it away from the turn that will reuse all of it."""
        with self._lock:
            best, best_n = -1, 0
            for i, (k, _) in enumerate(self._entries):
                n = common_prefix(k, keys)
                if n > best_n and 2 * n >= len(k):
                    best, best_n = i, n
            if best < 0 or best_n < self.min_rows:
                self.misses += 1
                return None, 0
            _, holder = self._entries.pop(best)
            self.hits += 1
            return holder, best_n

    def put(self, keys: np.ndarray, holder) -> None:
        evicted = []
        with self._lock:
            self._entries.append((np.asarray(keys, dtype=np.int64), holder))
            while len(self._entries) > self.capacity:
                evicted.append(self._entries.pop(0)[1])
        for h in evicted:
            h.close()

    def clear(self) -> None:
        with self._lock:
            old, self._entries = self._entries, []
        for _, h in old:
            h.close()

    def __len__(self):
        with self._lock:
            return len(self._entries)
