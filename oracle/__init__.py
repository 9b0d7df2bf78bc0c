"""CPU oracle for the tree-cover inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`sentinel-tree-cover_amd/`) may import from here; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do, and only
as the checker / reported baseline.

* `restate_numpy`  -- numpy/float64 restatement of the reference's numpy/scipy
  stages (codecs, indices, date regrid, Whittaker, medians, window grid,
  window assembly, normalisation, post-masks, Gaussian overlap mosaic,
  bilinear 20 m->10 m, DSen2 tiling driver).  PINNED against golden vectors
  captured from the imported reference (tools/gen_golden.py -> tests/golden/).
* `restate_model`  -- torch-CPU restatement of the ConvGRU/U-Net graph
  (src/train/src/model.py + src/train/train-model.py) and of the DSen2-lite
  graph decoded from models-release/supres-40k-swir/superresolve_graph.pb.
  PARITY UNPINNED w.r.t. the true TF graphs: TensorFlow is not installable
  here and the ConvGRU/U-Net weights are absent from the checkout
  (SURVEY.md F3/F4).  DSen2 uses the real published weights.
"""
