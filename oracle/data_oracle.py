"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's noise helpers on the loader side of the hot path
(SURVEY.md 8(f) row 3).  Only tests/ and oracle/gen_golden.py import this.  Pinned by tests/golden/salt_pepper.npz, which holds the
outputs of the reference's own `add_salt_peper` / `add_salt_peper_3D` (code/data_harvard.py:24-48) and the draws they consumed."""
import numpy as np


def salt_pepper_count(amount, H, W):
    """Points per polarity: ceil(amount * H * W * 0.5) (data_harvard.py:27,30 for a 2-D slice, :39,43 for an HWC image)."""
    return int(np.ceil(amount * H * W * 0.5))


def salt_pepper_hwc(image, salt_r, salt_c, pep_r, pep_c):
    """add_salt_peper (data_harvard.py:35-48) on an HWC image, or add_salt_peper_3D (:24-33) on a 2-D slice, given the
    coordinate draws: salt points -> 1 on every channel, THEN pepper points -> 0 (a pixel drawn by both ends at 0).  The
    reference draws coordinates with np.random.randint(0, size - 1): the last row / column is never hit."""
    out = np.copy(image)
    out[salt_r, salt_c] = 1.0
    out[pep_r, pep_c] = 0.0
    return out
