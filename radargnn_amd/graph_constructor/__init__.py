"""Host-side mirror of ``gnnradarobjectdetection.graph_constructor`` (numpy in, numpy out) running the search and
the feature extraction on the MI355X, plus the 25-line caller ``build_geometric_graph``."""
from .configs import GraphConstructionConfiguration  # noqa: F401
from .graph import (GeometricGraph, Graph, build_geometric_graph, create_graph_tensors,  # noqa: F401
                    nearest_neighbor_index, nearest_neighbor_points)
from .features import get_En_equivariant_point_pair_metrics  # noqa: F401
