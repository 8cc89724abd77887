"""Reference logits for the hardest inputs of the suite: uniform-noise lines at different widths.

Same rules as make_golden.py (whose helpers it uses): runs only in the build container, imports the
reference's crnn/network_torch.py as it lies in /root/reference (nothing copied), records inputs and
the reference's outputs as plain numpy data in tests/golden/noise_logits.npz:

  w<W>/gray        uint8 [32, W]     the line
  w<W>/en_logits   f32 [T, 65]       reference forward(), 65-class head, all classes
  w<W>/zh_cols     int32 [T, 257]    per timestep: the 16 classes with the largest reference logits,
                                     then the 241 classes 0, 23, 46, ... (a fixed stride through all 5530)
  w<W>/zh_logits   f32 [T, 257]      reference forward(), 5530-class head, at those classes
  w<W>/zh_idx      int32 [T]         reference arg-max

Usage:  python tests/golden/make_golden_noise.py
"""
import os

import numpy as np

import make_golden as mg   # imports the reference (see its header)

WIDTHS = [24, 57, 100, 141, 200, 233, 280, 281, 301, 330]


def main():
    zh = mg.build_ref(5530, mg.ref_keys.alphabetChinese)
    en = mg.build_ref(65, mg.ref_keys.alphabetEnglish)
    arrays = {}
    stride = np.arange(0, 5530, 23, dtype=np.int32)
    for k, w in enumerate(WIDTHS):
        ln = mg.synth_lines(7000 + k, 1, w)[0]
        _, lz = mg.ref_logits(zh, ln)
        _, le = mg.ref_logits(en, ln)
        top = np.argsort(-lz, axis=1, kind="stable")[:, :16].astype(np.int32)
        cols = np.concatenate([top, np.broadcast_to(stride, (lz.shape[0], len(stride)))], 1)
        arrays["w%d/gray" % w] = ln
        arrays["w%d/en_logits" % w] = le.astype(np.float32)
        arrays["w%d/zh_cols" % w] = cols
        arrays["w%d/zh_logits" % w] = np.take_along_axis(lz, cols, 1).astype(np.float32)
        arrays["w%d/zh_idx" % w] = lz.argmax(1).astype(np.int32)
    out = os.path.join(mg.HERE, "noise_logits.npz")
    np.savez_compressed(out, **arrays)
    print("wrote", len(arrays), "arrays;", os.path.getsize(out) / 1e6, "MB")


if __name__ == "__main__":
    main()
