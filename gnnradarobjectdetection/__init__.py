"""Import-path shim: ``gnnradarobjectdetection.graph_constructor`` and ``gnnradarobjectdetection.gnn`` resolve to the
MI355X implementation in ``radargnn_amd`` so that code written against the reference package
(``from gnnradarobjectdetection.gnn.gnn_models import DetNetBasic`` ...) runs unchanged.  Only the hot-path
sub-packages exist here; the reference's pre/post-processors are out of scope."""

# a whole-module pickle written by the reference's trainer (gnn/trainer.py:342-354) names torch_geometric classes: make those
# paths resolve to the HIP classes when torch_geometric itself is not installed (radargnn_amd/checkpoint.py)
from radargnn_amd.checkpoint import install_reference_pickle_shims as _install

_install()
