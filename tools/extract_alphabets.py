"""Write ocr_amd/data/alphabet_{chinese,english}.txt from the reference's
crnn/keys.py:3-4.  Run in the build container only (needs /root/reference);
the data files it writes are committed."""
import os
import sys

sys.path.insert(0, "/root/reference")
from crnn import keys  # noqa: E402

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ocr_amd", "data")
os.makedirs(out, exist_ok=True)
for name, s in (("alphabet_chinese.txt", keys.alphabetChinese), ("alphabet_english.txt", keys.alphabetEnglish)):
    with open(os.path.join(out, name), "w", encoding="utf-8", newline="") as f:
        f.write(s)
    print(name, len(s))
