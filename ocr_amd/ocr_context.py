"""Batched CRNN branch for the reference's OCR context.

`obj_det/ocr_context.py` `TextOcrModel.batch_ocr(images, use_paddle_first=False)` loops the
CRNN one crop at a time (`ocr_context.py:302-310`), and `_ocr` calls
`self._chinese_ocr_instance(img)` for single crops (`:119,:123,:139`).  This adapter keeps those
call shapes and their error conventions ("" for a None image or any per-image failure) but runs
the crops as width-bucketed device batches through `CRNN.predict_lines`.

Usage inside the reference (one line in TextOcrModel.__init__ / batch_ocr):
    self._crnn_batch = CrnnOcrContext(settings.crnn)
    ...
    else:   # use_paddle_first == False
        return self._crnn_batch.batch_ocr(images)
"""
import numpy as np


class CrnnOcrContext(object):
    def __init__(self, crnn, max_batch=256):
        self.crnn = crnn
        self.max_batch = max_batch
        self._chinese_ocr_instance = crnn.ocr   # same attribute name as TextOcrModel (:28)

    def ocr(self, img):
        """Single BGR crop -> text; "" on any failure (the try/except of ocr_context.py:133-144)."""
        try:
            return self.crnn.ocr(img) if img is not None else ""
        except Exception:
            return ""

    @staticmethod
    def _usable(img):
        if img is None:
            return False
        a = np.asarray(img)
        return a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3 and a.size > 0

    def batch_ocr(self, images, use_paddle_first=False, batch_size=None):
        """List of BGR crops -> list of texts, position by position (ocr_context.py:271-314, CRNN branch)."""
        if not images:
            return []
        results = [""] * len(images)
        good = [i for i, im in enumerate(images) if self._usable(im)]
        if not good:
            return results
        try:
            texts = self.crnn.predict_lines([images[i] for i in good],
                                            max_batch=batch_size or self.max_batch)
            for i, t in zip(good, texts):
                results[i] = t
        except Exception:
            # one bad crop (e.g. resized width 0) must not lose the others: isolate per image
            for i in good:
                results[i] = self.ocr(images[i])
        return results

    def page_ocr(self, page, boxes):
        """Texts of the crops ``page[y1:y2, x1:x2]`` for boxes = [(x1, y1, x2, y2), ...]: what the
        detectors do crop by crop (vat_detect.py:291,318 -> ocr_context.py:119) with ONE upload of the page;
        the gray conversion and the resize read the rectangles from the page in device memory.
        "" for a degenerate box, as the per-crop try/except of the reference gives."""
        if page is None or len(boxes) == 0:
            return [""] * len(boxes)
        a = np.asarray(page)
        results = [""] * len(boxes)
        good = []
        for i, b in enumerate(boxes):
            x1, y1, x2, y2 = [int(v) for v in b]
            if x1 < 0 or y1 < 0 or min(x2, a.shape[1]) - x1 <= 0 or min(y2, a.shape[0]) - y1 <= 0:
                continue
            good.append(i)
        if not good:
            return results
        try:
            texts = self.crnn.ocr_page(a, [boxes[i] for i in good], max_batch=self.max_batch)
            for i, t in zip(good, texts):
                results[i] = t
        except Exception:
            for i in good:
                x1, y1, x2, y2 = [int(v) for v in boxes[i]]
                results[i] = self.ocr(a[y1:y2, x1:x2])
        return results
