from radargnn_amd.graph_constructor.graph import GeometricGraph, Graph  # noqa: F401
