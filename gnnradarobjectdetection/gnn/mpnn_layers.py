from radargnn_amd.gnn.linear import Linear  # noqa: F401
from radargnn_amd.gnn.mpnn_layers import MPNNConv, RadarPointGNNConv  # noqa: F401
