from radargnn_amd.gnn.configs import GNNArchitectureConfig  # noqa: F401
