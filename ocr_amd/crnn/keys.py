"""Alphabets of the CRNN recogniser.

Mirror of the reference's ``crnn/keys.py:3-4`` (``alphabetChinese`` 5529 code
points, ``alphabetEnglish`` 64 characters).  The character tables are *data*
the trained weights are defined against (class k <-> alphabet[k-1], class 0 is
the CTC blank, ``crnn/util.py:16-22``), so they are kept as UTF-8 data files
under ``ocr_amd/data/`` (written by ``tools/extract_alphabets.py``) and loaded
here.
"""
import os

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def _load(name):
    with open(os.path.join(_DATA, name), "r", encoding="utf-8", newline="") as f:
        return f.read()


alphabetChinese = _load("alphabet_chinese.txt")
alphabetEnglish = _load("alphabet_english.txt")

assert len(alphabetChinese) == 5529, len(alphabetChinese)
assert len(alphabetEnglish) == 64, len(alphabetEnglish)
