"""MI355X-native tree-cover inference hot path (drop-in for the per-tile numeric
core of wri/sentinel-tree-cover's src/download_and_predict_job.py).

The directory name follows the build contract (`sentinel-tree-cover_amd/`); since a
hyphen is not a legal identifier, import it through the `ttc` shim at the repo
root (`import ttc`) or `importlib.import_module("sentinel-tree-cover_amd")`.

All arithmetic runs in hand-written HIP kernels inside `libttc_hip.so`
(csrc/, C-ABI in include/ttc.h).  Importing this package does NOT load the
library; the first compute call does, and fails loudly if it is missing.
"""
__all__ = ["synth"]
