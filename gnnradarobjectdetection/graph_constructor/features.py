from radargnn_amd.graph_constructor.features import get_En_equivariant_point_pair_metrics  # noqa: F401
