#!/usr/bin/env python3
"""Writes tests/golden/inferno_256.npy: the 256 RGB8 entries of the Inferno gradient the reference paints
with (ui/src/lib.rs:113-115, colorous::INFERNO).  colorous' table is the one of d3-scale-chromatic, which is
matplotlib's `inferno` listed colormap rounded to 8 bits per channel (round(255 * x)); matplotlib ships in
this image, so the table is regenerated from that published data rather than typed in.  Spot checks against
the first and last entries of d3's string ("000004", "010005", ... "fcffa4") are in test_colormap.py."""
import os

import numpy as np
from matplotlib import colormaps

data = np.asarray(colormaps["inferno"].colors, np.float64)
assert data.shape == (256, 3)
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "inferno_256.npy"), np.rint(255.0 * data).astype(np.uint8))
