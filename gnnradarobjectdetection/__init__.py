"""Import-path shim: ``gnnradarobjectdetection.graph_constructor`` and ``gnnradarobjectdetection.gnn`` resolve to the
MI355X implementation in ``radargnn_amd`` so that code written against the reference package
(``from gnnradarobjectdetection.gnn.gnn_models import DetNetBasic`` ...) runs unchanged.  Only the hot-path
sub-packages exist here; the reference's pre/post-processors are out of scope."""
