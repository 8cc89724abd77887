"""Host-side counterparts of the reference's ``crnn/util.py``.

``resizeNormalize`` (util.py:6-13) and the arg-max / greedy CTC collapse that
``strLabelConverter`` (util.py:16-22) performs run inside the HIP library; what
is left on the host is the table lookup from label positions to characters.
"""
import numpy as np


def labels_to_text(labels, alphabet, alphabet_codepoints=None):
    """Collapsed label positions (class index - 1) -> str, i.e. the ''.join of util.py:21-22."""
    if len(labels) == 0:
        return ''
    if alphabet_codepoints is not None:
        return alphabet_codepoints[np.asarray(labels)].astype("<u4").tobytes().decode("utf-32-le")
    return ''.join(alphabet[int(k)] for k in labels)


def strLabelConverter(res, alphabet):
    """Same contract as the reference function (util.py:16-22): raw per-timestep class indices
    -> text.  Kept for callers that hold raw arg-max sequences (e.g. the ABI's out_raw)."""
    raw = []
    prev = None
    for i, k in enumerate(res):
        k = int(k)
        if k != 0 and not (i > 0 and prev == k):
            raw.append(alphabet[k - 1])
        prev = k
    return ''.join(raw)
