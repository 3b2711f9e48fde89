"""Import-path shim: ``gnnradarobjectdetection.graph_constructor`` and ``gnnradarobjectdetection.gnn`` resolve to the
MI355X implementation in ``radargnn_amd`` so that code written against the reference package
(``from gnnradarobjectdetection.gnn.gnn_models import DetNetBasic`` ...) runs unchanged.  Only the hot-path
sub-packages exist here; the reference's pre/post-processors are out of scope.

A whole-module pickle written by the reference's trainer (gnn/trainer.py:342-354) names torch_geometric classes.  Reading one with the
reference's own ``torch.load`` (evaluate.py:46-52) needs those paths importable; where torch_geometric is not installed,
``enable_reference_pickles()`` (or RGNN_REFERENCE_PICKLES=1 in the environment of an unchanged script) registers stand-in modules
for them -- an explicit opt-in, because ``import torch_geometric`` succeeds with the stand-in while they are installed
(radargnn_amd/checkpoint.py; ``radargnn_amd.checkpoint.load_reference_model`` needs none of this)."""
import os as _os


def enable_reference_pickles() -> bool:
    from radargnn_amd.checkpoint import install_reference_pickle_shims
    return install_reference_pickle_shims()


def disable_reference_pickles() -> None:
    from radargnn_amd.checkpoint import remove_reference_pickle_shims
    remove_reference_pickle_shims()


if _os.environ.get("RGNN_REFERENCE_PICKLES"):
    enable_reference_pickles()
