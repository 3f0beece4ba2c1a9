"""Developer tool: one line per raw stamp file of tools/fused_wave_stamps.py (FLOW2D_STAMPS_OUT): launch span and the
lifetimes of the older (wave slot 0) and the younger (slot 1) wave of a SIMD.  usage: python tools/stamps_summary.py files..."""
import sys

import numpy as np

for path in sys.argv[1:]:
    st = np.load(path)
    t0 = st[:, 0].astype(np.int64) * 10
    t1 = st[:, 1].astype(np.int64) * 10
    life = (t1 - t0) / 1e3
    slot = st[:, 3].astype(np.int64) & 15
    print("%-60s span %6.1f us   slot 0: median %6.1f p90 %6.1f   slot 1: median %6.1f p90 %6.1f   mean %6.1f" %
          (path.split("/")[-1], (t1.max() - t0.min()) / 1e3, np.median(life[slot == 0]), np.percentile(life[slot == 0], 90),
           np.median(life[slot == 1]), np.percentile(life[slot == 1], 90), life.mean()))
